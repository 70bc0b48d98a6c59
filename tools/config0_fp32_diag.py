"""Diagnostic for tests/test_model_gpu.py::test_baseline_config0_matches_oracle[fp32]: per-tensor error statistics of the fp32 path
against the oracle (fraction of elements outside the test's tolerance, largest error relative to the tensor's and the global maximum)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_model as R
from tests.test_model_gpu import build, oracle_case, to_pack

over = dict(layer_num=2, use_decoder=False, ctc_weight=1.0)
cfg, sd, batch = oracle_case(4, 100, 80, 50, 12, over, seed=21)
ref = R.RefTrainer(sd, cfg, warmup=4000).iterate(batch)
model = build(cfg, 50, "TransformerCTC", dtype="fp32").cuda()
model.load_state_dict({k: v for k, v in sd.items()})
model._ensure_engine("cuda")
model.zero_flat_grads()
loss, _ = model.train_step(to_pack(batch))
print("loss rel", abs(float(loss[0]) - float(ref["loss"])) / abs(float(ref["loss"])))
gmax = max(float(g.abs().max()) for g in ref["grads"].values())
for n, p in model.named_parameters():
    g = ref["grads"][n]
    d = (p.grad.cpu() - g).abs()
    frac = float((d > 1e-3 * g.abs() + 2e-5 * max(gmax, 1.0)).float().mean())
    print(f"{n:50s} frac_off {frac:.5f}  max_err/tensor_max {float(d.max()) / (float(g.abs().max()) + 1e-30):.2e}  max_err/gmax {float(d.max()) / gmax:.2e}  tensor_max/gmax {float(g.abs().max()) / gmax:.2e}")
