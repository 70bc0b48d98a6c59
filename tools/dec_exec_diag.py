"""Where the native decoder sequencer (ASR_DEC_EXEC=1) and the per-kernel Python path (=0) differ, in deterministic mode:
per parameter tensor (flat order) the largest gradient difference, for exec-vs-exec, python-vs-python and exec-vs-python."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import kernels as K
from tests.test_model_gpu import build, oracle_case, to_pack

over = dict(d_model=512, hidden_size=64, num_head=8, ff_size=1024, layer_num=2, ctc_weight=0.3, dropout=0.0)
cfg, sd, batch = oracle_case(4, 136, 80, 56, 12, over, seed=9)
sd["decoder.tgt_word_emb.weight"] = sd["decoder.tgt_word_emb.weight"] * 0.05
sd["decoder.tgt_word_prj.weight"] = sd["decoder.tgt_word_emb.weight"]
pack = to_pack(batch)
K.set_deterministic(True)
def run(mode):
    os.environ["ASR_DEC_EXEC"] = mode
    model = build(cfg, 56, "TransformerOffical", dtype="bf16").cuda()
    model.load_state_dict(sd); model.train()
    model._ensure_engine("cuda"); model.zero_flat_grads()
    loss, _ = model.train_step(pack)
    torch.cuda.synchronize()
    return model, loss.clone(), model._flat.g.clone()
runs = {k: run(k[0]) for k in ("1a", "1b", "0a", "0b")}
def cmp(a, b):
    ma, la, ga = runs[a]; mb, lb, gb = runs[b]
    print(f"== {a} vs {b}: loss equal {torch.equal(la, lb)}  max |dg| {float((ga - gb).abs().max()):.3e}")
    for n, (off, shape) in ma._flat.index.items():
        k = 1
        for s_ in shape: k *= s_
        d = float((ga[off:off + k] - gb[off:off + k]).abs().max())
        if d > 0:
            print(f"     {n:55s} {d:.3e} (max |g| {float(ga[off:off + k].abs().max()):.3e})")
cmp("1a", "1b"); cmp("0a", "0b"); cmp("1a", "0a")
