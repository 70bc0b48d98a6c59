"""Time of the step's batched weight transposes (asr_transpose_batched_bf16 over every matrix of the joint model) and a spot check of three copies."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from asr_chinese_e2e_amd import Models
from asr_chinese_e2e_amd.data_handler import Vocab
M = Models.TransformerOffical
cfg = M.get_default_config()(); cfg.fn_build(dict(n_mels=80, lfr_m=1, dropout=0.0, ctc_weight=0.3))
model = M(cfg, Vocab.synthetic(4232)).cuda()
eng = model._ensure_engine("cuda")
from asr_chinese_e2e_amd import kernels as K
f = eng.flat
def run(): K.transpose_batched(f.lp, f.lpT, eng._tr_tiles)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(f"transposes of the joint model: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us for {eng._tr_tiles.shape[0]} tiles")
# check one matrix
l = eng.enc[0][0].qkv
print("qkv copy ok:", torch.equal(l.wlpT, l.wlp.t()), " head ok:", torch.equal(eng.ctc_lo.wlpT, eng.ctc_lo.wlp.t()), " kv_all ok:", torch.equal(eng.kv_all.wlpT, eng.kv_all.wlp.t()))
