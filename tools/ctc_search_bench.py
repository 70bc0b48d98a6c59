"""CTC prefix beam search / joint rescoring throughput at the BASELINE shape (B=32, T=500, V=4232), random weights."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import Models
from asr_chinese_e2e_amd.data_handler import Vocab, synthetic_pack
M = Models.TransformerOffical
cfg = M.get_default_config()(); cfg.fn_build(dict(n_mels=80, lfr_m=1, dropout=0.0, ctc_weight=0.3, cer_in_iterate=False))
model = M(cfg, Vocab.synthetic(4232)).cuda(); model.eval()
pack = synthetic_pack(32, 500, 80, 4232, device="cuda", dtype=torch.bfloat16)
def t(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
a = t(lambda: model.ctc_prefix_beam_search(pack, beam_size=5, nbest=1))
b = t(lambda: model.beam_search(pack, beam_size=5, nbest=1, decode_max_len=32, ctc_weight=0.3))
c = t(lambda: model.ctc_greedy_search(pack))
print(f"ctc prefix beam search (beam 5, top-10 classes per frame): {a:8.1f} ms per batch of 32 = {32e3 / a:7.1f} utt/s")
print(f"attention beam 5 + CTC rescoring (lambda 0.3):            {b:8.1f} ms per batch of 32 = {32e3 / b:7.1f} utt/s")
print(f"ctc greedy:                                                {c:8.1f} ms per batch of 32 = {32e3 / c:7.1f} utt/s")
