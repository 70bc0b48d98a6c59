#!/usr/bin/env python3
"""bench.py - utterances/s of the full training step (fwd + CTC loss + bwd + clip + Noam/Adam).

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by the driver as  python -m torch.distributed.run --nproc-per-node N ... bench.py
  (one rank per GPU over RCCL).  W untimed warm-up steps, then EXACTLY K timed steps bracketed by
  barrier + torch.cuda.synchronize() on both sides, MAX over ranks, rank 0 prints ONE JSON line.

Workload at N=1 = BASELINE.json configs[1]: 6-layer Transformer encoder + CTC-only, bf16,
batch 32, T=500, F=80, V=4232 (AISHELL-1 char vocab size), synthetic N(0,1) features, random-init
weights.  --config joint runs configs[2] (encoder-decoder, lambda=0.3) instead.
Weak scaling: every rank processes its own 32-utterance batch.

Extra objects on the JSON line:
  roofline      dominant kernel family (MFMA GEMMs), algorithmic FLOP / HIP-event time of its
                launches inside the timed region, against the dense bf16 MFMA peak (2.5 PFLOP/s).
  cpu_baseline  the CPU oracle (oracle/ref_model.RefTrainer: op-for-op port of the reference's
                TransformerOffical.iterate + CTC) timed on this host's cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16, MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--config", default="ctc", choices=["ctc", "joint"])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=500)
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--vocab", type=int, default=4232)
    ap.add_argument("--window", type=int, default=-1)
    ap.add_argument("--dropout", type=float, default=0.0)
    ap.add_argument("--graph", action="store_true", help="replay one captured hipGraph per step instead of eager launches "
                    "(measured SLOWER on MI355X/ROCm 7: 5.21 vs 4.88 ms CTC-only, 8.81 vs 8.42 ms joint - the step is GPU-bound)")
    ap.add_argument("--cer", action="store_true", help="joint config: character error rate of the greedy ids in every training step, as the "
                    "reference's iterate does (on the device: asr_cer)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--cpu-sample-batch", type=int, default=8)
    return ap.parse_args()


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
            if q != "max":
                n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return min(n, 64)


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(args):
    """Oracle ('port') timed on the host cores: same model/shape, a bounded sample (batch 8)."""
    from oracle import ref_model as R
    from asr_chinese_e2e_amd.data_handler import synthetic_pack
    torch.set_num_threads(host_cores())
    cores = torch.get_num_threads()
    log(f"cpu baseline on {cores} threads (affinity {len(os.sched_getaffinity(0))})")
    joint = args.config == "joint"
    cfg = R.default_cfg(n_mels=80, lfr_m=1, layer_num=args.layers, use_decoder=joint, ctc_weight=0.3 if joint else 1.0)
    sd = R.init_state_dict(cfg, args.vocab, seed=0)
    tr = R.RefTrainer(sd, cfg, warmup=4000, id2token=[str(i) for i in range(args.vocab)])
    B = args.cpu_sample_batch
    pack = synthetic_pack(B, args.frames, 80, args.vocab, seed=1234)
    batch = {k: pack[k] for k in ("wave", "wave_len", "tgt_for_input", "tgt_len")}
    tw = time.time()
    tr.iterate(batch, loop_masks=True, with_cer=joint)            # warm-up
    log(f"cpu warm-up step {time.time() - tw:.1f} s")
    n, t0 = 0, time.time()
    while n < 2 or (time.time() - t0 < 10.0 and n < 6):
        tr.iterate(batch, loop_masks=True, with_cer=joint)
        n += 1
        log(f"cpu step {n}: {time.time() - t0:.1f} s")
    dt = (time.time() - t0) / n
    return {"value": B / dt, "unit": "utterances/s", "cores": cores, "kind": "port",
            "sample": f"{n} steps of batch {B} x T={args.frames} (same {args.layers}-layer model, fp32, torch CPU, "
                      f"python-loop masks as the reference), {dt:.2f} s/step"}


def pmc_traffic(family):
    """HBM bytes per launch of a kernel family from the committed PMC summary (None if absent)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "round1_pmc_traffic.json")
    pat = {"gemm_nt": "gemm_nt", "gemm_tn": "gemm_tn", "lib_gemm_dgrad": "Cijk"}[family]
    try:
        table = json.load(open(path))
    except (OSError, ValueError):
        return None
    rows = [v for k, v in table.items() if pat in k]
    n = sum(r["launches"] for r in rows)
    return sum(r["hbm_bytes_per_launch"] * r["launches"] for r in rows) / n if n else None


def main():
    args = parse()
    # stdout carries exactly ONE line, the JSON result: libraries that write to file descriptor 1 (RCCL prints a
    # version banner when its first communicator is created) are sent to stderr for the whole run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the measured path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from asr_chinese_e2e_amd import Models, kernels as K
    from asr_chinese_e2e_amd.data_handler import Vocab, synthetic_pack
    from asr_chinese_e2e_amd.Trainer import FusedAdam, NoamOpt
    use_dp = world > 1 or os.environ.get("ASR_FORCE_DP") == "1"    # ASR_FORCE_DP: exercise the RCCL path with one rank
    if use_dp:
        from asr_chinese_e2e_amd import dist as D
        D.init(os.environ.get("ASR_DIST_BACKEND", "nccl"))   # "gloo": rehearse N ranks on a one-GPU box (LOCAL_RANK=0 for all)

    joint = args.config == "joint"
    Model = Models.TransformerOffical if joint else Models.TransformerCTC
    cfg = Model.get_default_config()()
    cfg.fn_build(dict(n_mels=80, lfr_m=1, dropout=args.dropout, layer_num=args.layers, ctc_weight=0.3 if joint else 1.0, dtype="bf16",
                      attn_window=args.window, cer_in_iterate=args.cer, warm_up=4000))
    torch.manual_seed(0)
    model = Model(cfg, Vocab.synthetic(args.vocab)).to(dev)
    adam = FusedAdam(model.parameters(), lr=3e-4, betas=(0.9, 0.98), eps=1e-9)
    opt = NoamOpt(cfg.d_model, 1, cfg.warm_up, adam)
    pack = synthetic_pack(args.batch, args.frames, 80, args.vocab, seed=1234 + rank, device=dev, dtype=torch.bfloat16)
    runner = model
    graphed = False
    if use_dp:
        runner = D.DataParallel(model, dev)
    elif args.graph and args.dropout == 0.0:
        from asr_chinese_e2e_amd.graph import GraphedModel
        runner = GraphedModel(model)          # whole step as one hipGraph (single process, no dropout)
        graphed = True

    def barrier():
        if use_dp:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    last = None
    log(f"rank {rank}/{world}: model on {dev}, warm-up {args.warmup} steps")
    for _ in range(args.warmup):
        last, _ = runner.iterate(pack, optimizer=opt, is_train=True)
    barrier()
    log("warm-up done")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last, _ = runner.iterate(pack, optimizer=opt, is_train=True)
    barrier()
    dt = time.perf_counter() - t0
    log(f"timed region done: {1e3 * dt / args.steps:.2f} ms/step")
    # Roofline leg: the SAME steps once more with HIP events around every launch of the GEMM
    # families (events are recorded on the stream the kernels are launched on).  It is a separate
    # pass because ~80 event pairs per step cost ~25 % wall time on ROCm; `value` above is from the
    # un-instrumented region.  rocprofv3 (profiles/) cross-checks the per-kernel durations.
    timer = None
    if not args.no_kernel_timer:
        timer = K.LaunchTimer(["gemm_nt", "gemm_tn", "lib_gemm_dgrad"])
        K.TIMER = timer
        n_inst = min(args.steps, 10)
        model._engine.overlap_wgrad = False   # standalone kernel durations (concurrent streams inflate them)
        inst_runner = model if graphed else runner          # events cannot be read back from a captured graph
        for _ in range(n_inst):
            inst_runner.iterate(pack, optimizer=opt, is_train=True)
        barrier()
        K.TIMER = None
        model._engine.overlap_wgrad = True
    if use_dp:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    loss = float(last.loss)

    if rank == 0:
        utt = world * args.batch * args.steps / dt
        out = {
            "metric": "training throughput (utterances/s; frames/s = x T), AISHELL-1-shaped 80-mel T=500",
            "value": utt, "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "frames_per_s": utt * args.frames, "final_loss": loss,
            "config": {"workload": ("configs[2]: joint CTC/attention (lambda=0.3) encoder-decoder" if joint else
                                    "configs[1]: 6-layer Transformer encoder + CTC-only") +
                                   f", bf16, per-GPU batch {args.batch}, T={args.frames}, F=80, V={args.vocab}, "
                                   f"{args.layers} layers, d_model 512, 8x64 heads, ff 1024, dropout {args.dropout}, "
                                   "fwd+loss+bwd+clip+Noam/Adam per step" + (", one hipGraph per step" if graphed else ", eager launches"),
                       "global_batch": world * args.batch, "seq_len": args.frames, "parallelism": f"dp{world}"},
        }
        if timer is not None:
            s = timer.summary()
            fam = {k: v for k, v in s.items()}
            dom = max(fam, key=lambda k: fam[k]["total_ms"]) if fam else None
            if dom:
                a = fam[dom]["work_per_s"] / 1e12
                out["roofline"] = {"bound": "mfma", "kernel": {"gemm_nt": "gemm_nt_persist_kernel (asr_gemm_nt_bf16)", "gemm_tn": "gemm_tn_dma_kernel (asr_gemm_tn_bf16)",
                                                               "lib_gemm_dgrad": "hipBLASLt GEMM (input gradients not on the own kernel)"}[dom],
                                   "achieved": a, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": a / MFMA_BF16_PEAK_TFLOPS,
                                   "traffic": pmc_traffic(dom), "traffic_source": "profiles/round1_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE passes of this command (separate runs), bytes per launch averaged over the family's launches, FETCH_SIZE x2 (gfx950 correction)",
                                   "avg_launch_us": fam[dom]["avg_us"], "launches": fam[dom]["launches"],
                                   "flop_per_launch": fam[dom]["work_per_launch"],
                                   "share_of_step": fam[dom]["total_ms"] / n_inst / (1e3 * dt / args.steps)}
                out["kernel_families"] = {k: {"tflops": v["work_per_s"] / 1e12, "avg_us": v["avg_us"], "launches": v["launches"],
                                              "share_of_step": v["total_ms"] / n_inst / (1e3 * dt / args.steps)} for k, v in fam.items()}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if use_dp:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
