"""Every documented environment switch of the engine against the default setting: ONE training step (no optimizer) of the same model on the
same batch, every gradient tensor compared with the default's (worst |g - g0| / max|g0| over all tensors, and the tensor).  The switches
change the ORDER of work and of a few bf16 / fp32 sums, never the mathematics: a difference above ~1e-2 is a bug in a rarely used path.
LAYERS=2 python tools/switch_matrix.py   (B=32 T=500 V=4232 LAYERS=6 TO=17 for the headline shapes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import Models, kernels as K
from asr_chinese_e2e_amd.data_handler import Vocab, synthetic_pack
LAYERS, TO = int(os.environ.get("LAYERS", "2")), int(os.environ.get("TO", "9"))
B, T, V = int(os.environ.get("B", "4")), int(os.environ.get("T", "96")), int(os.environ.get("V", "60"))
DROPOUT = float(os.environ.get("DROPOUT", "0"))
SETTINGS = [{}, {"ASR_WGRAD_OVERLAP": "0"}, {"ASR_DETERMINISTIC": "1"}, {"ASR_WGRAD_GROUP": "0"}, {"ASR_WGRAD_GROUP": "layer"}, {"ASR_WGRAD_GROUP": "block"},
            {"ASR_WGRAD_GROUP": "pair"}, {"ASR_WGRAD_GROUP": "pair_ffn"}, {"ASR_WGRAD_GROUP": "pair_attn"}, {"ASR_KV_GROUPS": "1"}, {"ASR_KV_GROUPS": "2"},
            {"ASR_KV_GROUPS": "6"}, {"ASR_WGRAD_DEFER": ""}, {"ASR_WGRAD_DEFER": "fc,w2"}, {"ASR_ARMED_FORK": "0"}, {"ASR_DEC_EXEC": "0"},
            {"ASR_DEC_CU_LIMIT": "0"}, {"ASR_DEC_CU_LIMIT": "96"}, {"ASR_WGRAD_GROUP": "block", "ASR_WGRAD_DEFER": "w2", "ASR_FUSE_RELU_BWD": "0"}]
pack = synthetic_pack(B, T, 80, V, seed=27, ragged=True, Lmin=max(2, TO - 3), Lmax=TO, device="cuda", dtype=torch.bfloat16)
names, ref, ref_loss, bad = None, None, None, 0
for cfg_name in (("joint", "ctc") if os.environ.get("MODEL", "both") == "both" else (os.environ["MODEL"],)):
    ref = None
    for env in SETTINGS:
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        K.set_deterministic(env.get("ASR_DETERMINISTIC") == "1")
        try:
            torch.manual_seed(5)
            M = Models.TransformerCTC if cfg_name == "ctc" else Models.TransformerOffical
            cfg = M.get_default_config()()
            cfg.fn_build(dict(n_mels=80, lfr_m=1, layer_num=LAYERS, dropout=DROPOUT, ctc_weight=1.0 if cfg_name == "ctc" else 0.3, dtype="bf16"))
            m = M(cfg, Vocab.synthetic(V)).cuda()
            m._ensure_engine(torch.device("cuda", 0))
            for _ in range(2):      # the second step runs with warm caches (decoder buffers, workspaces): that is the one compared
                m._step_seed = 1234
                m.zero_flat_grads()
                loss = m.train_step(pack)[0]
            torch.cuda.synchronize()
            flat = m._flat
            g = flat.g.clone()
            if ref is None:
                ref, ref_loss, names = g, loss.clone(), list(flat.index)
                scale = torch.stack([flat.view(ref, n).abs().max() for n in names]).clamp_min(1e-30)
                print(f"{cfg_name}: default setting, loss {float(loss[0]):.6f}")
                continue
            d = (g - ref).abs()
            rel = torch.stack([flat.view(d, n).max() for n in names]) / scale
            rel = torch.where(torch.tensor([n.endswith("w_ks.bias") for n in names], device="cuda"), torch.zeros_like(rel), rel)      # analytically zero gradients: round-off only
            w = int(rel.argmax())
            flag = "" if float(rel[w]) < 1e-2 and abs(float(loss[0]) - float(ref_loss[0])) <= 1e-4 * abs(float(ref_loss[0])) else "   <-- LOOK"
            bad += bool(flag)
            print(f"  {' '.join(f'{k}={v}' for k, v in env.items()):70s} loss diff {abs(float(loss[0]) - float(ref_loss[0])):.2e}  worst gradient {float(rel[w]):.2e} ({names[w]}){flag}", flush=True)
        finally:
            for k, v in saved.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
            K.set_deterministic(False)
print("settings to look at:", bad)
