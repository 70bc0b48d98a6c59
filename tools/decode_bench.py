"""Decoding throughput at BASELINE size: attention beam search and greedy CTC on
one MI355X, next to the CPU oracle (restated reference algorithm) on a bounded sample.
python tools/decode_bench.py [--beam 5] [--max-len 32] [--batch 32] [--frames 500]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import Models
from asr_chinese_e2e_amd.data_handler import Vocab, synthetic_pack

ap = argparse.ArgumentParser()
ap.add_argument("--beam", type=int, default=5); ap.add_argument("--max-len", type=int, default=32)
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--frames", type=int, default=500)
ap.add_argument("--vocab", type=int, default=4232); ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--cpu-utts", type=int, default=1)
a = ap.parse_args()
M = Models.TransformerOffical
cfg = M.get_default_config()(); cfg.fn_build(dict(n_mels=80, lfr_m=1, dropout=0.0, ctc_weight=0.3))
torch.manual_seed(0)
model = M(cfg, Vocab.synthetic(a.vocab)).cuda(); model.eval()
pack = synthetic_pack(a.batch, a.frames, 80, a.vocab, device="cuda", dtype=torch.bfloat16)
model.beam_search(pack, a.beam, 1, a.max_len); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    hyps = model.beam_search(pack, a.beam, 1, a.max_len)
torch.cuda.synchronize(); t_beam = (time.perf_counter() - t0) / a.reps
steps = max(len(h[0]["yseq"]) for h in hyps) - 1
model.ctc_greedy_search(pack); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    model.ctc_greedy_search(pack)
torch.cuda.synchronize(); t_ctc = (time.perf_counter() - t0) / a.reps
out = {"workload": f"beam search beam={a.beam} max_len={a.max_len}, B={a.batch}, T={a.frames}, V={a.vocab}, 6+6 layers, bf16, random weights, incl. encoder",
       "beam_utt_per_s": a.batch / t_beam, "beam_ms_per_batch": 1e3 * t_beam, "decode_steps": steps,
       "ctc_greedy_utt_per_s": a.batch / t_ctc, "ctc_greedy_ms_per_batch": 1e3 * t_ctc}
if a.cpu_utts > 0:
    from oracle import ref_model as R
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    ocfg = R.default_cfg(n_mels=80, lfr_m=1)
    wave = pack.wave[: a.cpu_utts].float().cpu(); wl = pack.wave_len[: a.cpu_utts].cpu()
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    t0 = time.perf_counter()
    enc = R.encoder_forward(sd, ocfg, wave, wl)
    for b in range(a.cpu_utts):
        R.beam_search(sd, ocfg, enc[b, : int(wl[b])], a.beam, 1, a.max_len)
    t_cpu = time.perf_counter() - t0
    out["cpu_oracle_beam_utt_per_s"] = a.cpu_utts / t_cpu
    out["cpu_sample"] = f"{a.cpu_utts} utterance(s), {torch.get_num_threads()} threads, reference algorithm (whole decoder re-run per hypothesis and step)"
print(json.dumps(out))
