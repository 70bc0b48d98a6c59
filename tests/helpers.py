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
