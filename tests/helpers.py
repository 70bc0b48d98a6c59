"""Shared helpers for the tests (loading golden fixtures, building oracle inputs)."""
import os
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def golden_model_case(name):
    """-> (cfg namespace, state_dict of torch tensors, batch dict, raw npz)."""
    from oracle import ref_model as R
    z = load_npz(name)
    cfgd = {str(k): float(v) for k, v in zip(z["cfg/keys"], z["cfg/vals"])}
    cfg = R.default_cfg(**{k: (int(v) if k != "dropout" else v) for k, v in cfgd.items()})
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    for k in [k for k in z.files if k.startswith("pe_head/")]:
        sd[k[8:]] = R.positional_encoding(5000, cfg.d_model).unsqueeze(0)
    batch = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in/")}
    return cfg, sd, batch, z


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def free_port():
    """A TCP port that was free a moment ago (bound to port 0, then released): fixed rendezvous ports collide on a shared box."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def mfma_golden_case():
    """-> (cfg namespace, state_dict of torch tensors incl. PE buffers, batch dict, raw npz, golden_inputs module): the d_model = 512
    case whose weights / batch are rebuilt from numpy seeds (oracle/golden_inputs.py) and whose expected values come from the
    reference (tests/golden/model_mfma_d512.npz, written by `python oracle/gen_golden.py mfma`)."""
    from oracle import golden_inputs as GI
    from oracle import ref_model as R
    case = GI.MFMA_CASE
    cfg = R.default_cfg(**case["cfg"])
    sd = {k: torch.from_numpy(v) for k, v in GI.mfma_state_dict(case).items()}
    for k in ("encoder.positional_encoding.pe", "decoder.positional_encoding.pe"):
        sd[k] = R.positional_encoding(5000, cfg.d_model).unsqueeze(0)
    batch = {k: torch.from_numpy(v) for k, v in GI.mfma_batch(case).items()}
    return cfg, sd, batch, load_npz(case["name"] + ".npz"), GI
