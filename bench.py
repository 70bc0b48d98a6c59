#!/usr/bin/env python3
"""bench.py - utterances/s of the full training step (fwd + loss + bwd + grad all-reduce + clip + Noam/Adam).

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1: either the driver launches  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
  (one rank per GPU over RCCL; RANK / LOCAL_RANK / WORLD_SIZE in the environment), or - when WORLD_SIZE is NOT set -
  this script starts those N ranks itself as a child `torch.distributed.run` BEFORE it touches the GPU (the parent
  never initialises HIP and never execs) and passes the child's single JSON line through.
  W untimed warm-up steps, then EXACTLY K timed steps bracketed by barrier + torch.cuda.synchronize() on both
  sides, MAX over ranks, rank 0 prints ONE JSON line.  `n_gpus` is the number of ranks the process group formed.

Workload - ONE workload at every N (BASELINE.json's metric is the 1 / 2 / 4 / 8-GPU series of one per-GPU workload):
  BASELINE.json configs[2] (N = 1) = configs[3] per GPU (N > 1): joint CTC/attention (lambda = 0.3) encoder-decoder, bf16,
         32 utterances per GPU, T=500, F=80, V=4232, synthetic N(0,1) features, random-init weights; at N > 1 weak scaling with the
         bucketed RCCL all-reduce overlapped with backward.  --config ctc|joint overrides.
  Extra keys at N = 1: configs[1] (6-layer encoder + CTC-only: `ctc_ms_per_step`, `ctc_utterances_per_s`), the reference's dropout 0.1
         recipe on both models, configs[4] per GPU (long-form), training from waveforms - each timed the same way.

Extra objects on the JSON line:
  roofline      the GEMM family with the most time IN THE STEP (weight-gradient stream live beside the main stream, as in the timed
                region): algorithmic FLOP / HIP-event time of its launches, against the dense bf16 MFMA peak (2.5 PFLOP/s);
                `frac_standalone` = the same launches with the overlap off; `roofline_other_gemm_family` = the other family.
  kernels       per kernel family, HIP-event timed on the stream it is launched on: launches, us, algorithmic
                bytes and FLOP per launch (SURVEY.md 8(d) figures), HBM GB/s and fraction of 8 TB/s, TFLOP/s and
                fraction of the MFMA peak for the matrix kernels.
  cpu_baseline  the CPU oracle (oracle/ref_model.RefTrainer: op-for-op port of the reference's
                TransformerOffical.iterate + CTC) timed on this host's cores on a bounded sample.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16, MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBS = 8000.0
PMC_TRAFFIC_FILES = [os.path.join(ROOT, "profiles", n) for n in ("round5_pmc_traffic.json", "round4_c_pmc_traffic.json")]      # first that exists
PMC_TRAFFIC_FILE = next((f for f in PMC_TRAFFIC_FILES if os.path.isfile(f)), PMC_TRAFFIC_FILES[0])


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--config", default=None, choices=["ctc", "joint"], help="default: joint (configs[2] on one GPU = configs[3] per GPU on several)")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=500)
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--vocab", type=int, default=4232)
    ap.add_argument("--window", type=int, default=-1)
    ap.add_argument("--dropout", type=float, default=0.0)
    ap.add_argument("--graph", action="store_true", help="replay one captured hipGraph per step instead of eager launches "
                    "(measured SLOWER than the multi-stream eager step on MI355X/ROCm 7, DESIGN.md section 4 'Streams': the step is GPU-bound)")
    ap.add_argument("--no-cer", action="store_true", help="skip the character error rate of the greedy ids that every training step computes by default, "
                    "as the reference's iterate does (transformer_official.py:83-94; the trainer reads it every step, trainer11.py:73-75; here on the "
                    "device: asr_cer of the decoder's argmax ids, or of the greedy CTC path for the CTC-only model)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra single-GPU measurements (joint model, dropout 0.1)")
    ap.add_argument("--cpu-sample-batch", type=int, default=32)
    ap.add_argument("--plumbing", action="store_true", help="tests only: form the process group, exchange one tensor, print the line "
                    "skeleton - no GPU work (drives the N-rank launcher on a CPU-only machine)")
    return ap.parse_args(argv)


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


def launch_ranks(args):
    """WORLD_SIZE unset and --gpus N > 1: run the N ranks as a child torch.distributed.run.  The parent has not
    touched the GPU (no HIP call, not even torch.cuda.is_available()) and does not exec: it waits and relays."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    log(f"WORLD_SIZE not set: starting {args.gpus} ranks: {' '.join(cmd)}")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    lines = [l for l in p.stdout.decode("utf-8", "replace").splitlines() if l.startswith("{")]
    if p.returncode != 0 or not lines:
        sys.stderr.write(p.stdout.decode("utf-8", "replace"))
        raise SystemExit(p.returncode or 1)
    print(lines[-1], flush=True)
    raise SystemExit(0)


def cpu_baseline(args, config, warm=3, timed=5):
    """Oracle ('port') timed on the host cores: same model/shape (3 warm-up + >= 5 timed steps)."""
    import torch
    from oracle import ref_model as R
    from asr_chinese_e2e_amd.data_handler import synthetic_pack
    torch.set_num_threads(host_cores())
    cores = torch.get_num_threads()
    log(f"cpu baseline on {cores} threads (affinity {len(os.sched_getaffinity(0))})")
    joint = config == "joint"
    cfg = R.default_cfg(n_mels=80, lfr_m=1, layer_num=args.layers, use_decoder=joint, ctc_weight=0.3 if joint else 1.0)
    sd = R.init_state_dict(cfg, args.vocab, seed=0)
    tr = R.RefTrainer(sd, cfg, warmup=4000, id2token=[str(i) for i in range(args.vocab)])
    B = args.cpu_sample_batch
    pack = synthetic_pack(B, args.frames, 80, args.vocab, seed=1234)
    batch = {k: pack[k] for k in ("wave", "wave_len", "tgt_for_input", "tgt_len")}
    tw = time.time()
    n_warm = 0
    while n_warm < warm and (n_warm < 1 or time.time() - tw < 20.0):   # `warm` warm-up steps unless they alone exceed ~20 s
        tr.iterate(batch, loop_masks=True, with_cer=joint)
        n_warm += 1
    log(f"cpu warm-up: {n_warm} steps, {time.time() - tw:.1f} s")
    n, t0 = 0, time.time()
    while n < timed or (time.time() - t0 < 10.0 and n < timed + 3):
        tr.iterate(batch, loop_masks=True, with_cer=joint)
        n += 1
        log(f"cpu step {n}: {time.time() - t0:.1f} s")
    dt = (time.time() - t0) / n
    return {"value": B / dt, "unit": "utterances/s", "cores": cores, "kind": "port", "workload": config,
            "sample": f"{n_warm} warm-up + {n} timed steps of batch {B} x T={args.frames} (same {args.layers}-layer {'joint CTC/attention' if joint else 'encoder + CTC'} "
                      f"model, fp32, torch CPU, python-loop masks{' and host CER' if joint else ''} as the reference), {dt:.2f} s/step"}


def pmc_traffic(family):
    """HBM bytes per launch of a kernel family from the committed PMC summary (None if absent)."""
    pat = {"gemm_nt": "gemm_nt", "gemm_tn": "gemm_tn"}.get(family, family)
    try:
        table = json.load(open(PMC_TRAFFIC_FILE))
    except (OSError, ValueError):
        return None
    rows = [v for k, v in table.items() if pat in k]
    n = sum(r["launches"] for r in rows)
    return sum(r["hbm_bytes_per_launch"] * r["launches"] for r in rows) / n if n else None


def profiled_in_step_us(family, workload):
    """(launches per step, average us per launch, file) of a GEMM family in the committed rocprofv3 --kernel-trace summary of THIS command
    (profiles/round5_[ctc_]kernel_stats_per_step.csv, written by tools/profile_round.sh): the cross-check of the live HIP-event figure.  Events on a
    stream also see what the launch waits for in front of its first workgroup, so they read 10 - 20 % above the trace's kernel durations."""
    import csv
    path = os.path.join(ROOT, "profiles", "round5_ctc_kernel_stats_per_step.csv" if workload == "ctc" else "round5_kernel_stats_per_step.csv")
    try:
        rows = [r for r in csv.DictReader(open(path)) if r["Name"].startswith(family + "_")]
    except OSError:
        return None
    n = sum(float(r["CallsPerStep"]) for r in rows)
    tot = sum(float(r["TotalUsPerStep"]) for r in rows)
    return (n, tot / n, os.path.relpath(path, ROOT)) if n else None


MATRIX_FAMILIES = ("gemm_nt", "gemm_tn", "sdpa_fwd", "sdpa_bwd")
KERNEL_NAMES = {"gemm_nt": "gemm_nt_spec_kernel (asr_gemm_nt_bf16)", "gemm_tn": "gemm_tn_dma_kernel (asr_gemm_tn_bf16) + gemm_tn_grouped_kernel (the decoder layers' grouped launches)",
                "sdpa_fwd": "sdpa_fwd_fused_bf16_kernel (asr_sdpa_fwd)",
                "sdpa_bwd": "sdpa_bwd_fused_bf16_kernel (asr_sdpa_bwd)", "add_ln_fwd": "add_ln_fwd_kernel", "add_ln_bwd": "add_ln_bwd_kernel",
                "ctc": "ctc_lse_gather_rows + ctc_alpha_beta + ctc_label_fix (asr_ctc_fwd_bwd)", "xent": "xent_kernel", "adam": "adam_kernel (asr_adam_step)",
                "grad_sumsq": "sumsq kernels (asr_grad_sumsq)"}


class Run:
    """One model + optimizer + batch on this rank, and the timed loop over it."""

    def __init__(self, args, config, dropout, rank, dev, use_dp, batch=None, frames=None, window=None):
        import torch
        from asr_chinese_e2e_amd import Models
        from asr_chinese_e2e_amd.data_handler import Vocab, synthetic_pack
        from asr_chinese_e2e_amd.Trainer import FusedAdam, NoamOpt
        self.torch, self.args, self.joint, self.use_dp = torch, args, config == "joint", use_dp
        batch, frames, window = batch or args.batch, frames or args.frames, args.window if window is None else window
        Model = Models.TransformerOffical if self.joint else Models.TransformerCTC
        cfg = Model.get_default_config()()
        cfg.fn_build(dict(n_mels=80, lfr_m=1, dropout=dropout, layer_num=args.layers, ctc_weight=0.3 if self.joint else 1.0, dtype="bf16",
                          attn_window=window, cer_in_iterate=not args.no_cer, warm_up=4000))
        torch.manual_seed(0)
        self.model = Model(cfg, Vocab.synthetic(args.vocab)).to(dev)
        self.opt = NoamOpt(cfg.d_model, 1, cfg.warm_up, FusedAdam(self.model.parameters(), lr=3e-4, betas=(0.9, 0.98), eps=1e-9))
        self.pack = synthetic_pack(batch, frames, 80, args.vocab, seed=1234 + rank, device=dev, dtype=torch.bfloat16)
        self.runner, self.graphed, self.dp = self.model, False, None
        if use_dp:
            from asr_chinese_e2e_amd import dist as D
            self.runner = self.dp = D.DataParallel(self.model, dev)
            self.dp.bucketer.measure_exposed = True      # two timing events per step around the final wait for the communication stream
        elif args.graph and dropout == 0.0:
            from asr_chinese_e2e_amd.graph import GraphedModel
            self.runner = GraphedModel(self.model)          # whole step as one hipGraph (single process, no dropout)
            self.graphed = True
        self.last = None

    def barrier(self):
        if self.use_dp:
            self.torch.distributed.barrier()
        self.torch.cuda.synchronize()

    def steps(self, n, runner=None):
        runner = runner or self.runner
        for _ in range(n):
            self.last, _ = runner.iterate(self.pack, optimizer=self.opt, is_train=True)

    def timed(self, warmup, steps, strict=True):
        """`warmup` untimed steps, then exactly `steps` timed ones between barrier + synchronize.  The caching allocator must not
        grow inside the timed region (a hipMalloc synchronises the device): the engine keeps the operands of its side-stream kernels
        alive itself instead of Tensor.record_stream, so the allocation sequence repeats from the second step on (DESIGN.md section 5,
        round 4) - counted here, and with `strict` a non-zero count ends the run instead of reporting a number measured across it."""
        torch = self.torch
        # Building a model leaves a few hundred MB of freed host memory at the top of the C heap (parameter initialisation on the host), and glibc
        # gives it back to the kernel at some LATER free(): one ~30-ms pause of the launching thread at a random step - seen as one fifth of a
        # timed region running 40 % slow in 4 of 9 runs, with no collector pass in it (counted below); with malloc_trim(0) here: 0 of 8 runs
        # (BENCH_PREP=trim), with collect + trim + freeze: 0 of 18.  So: collect what the construction left, trim the heap, and keep the
        # survivors out of the collector's way (a full pass over this process takes 80 ms) - all BEFORE the warm-up steps: an 80-ms host pause
        # right in front of the timed region left the GPU idle and the first fifth 5 % slow.  BENCH_PREP=none restores the old state.
        import ctypes
        import gc
        prep = os.environ.get("BENCH_PREP", "all")
        if prep == "all":
            gc.collect()
        if prep in ("all", "trim"):
            try:
                ctypes.CDLL("libc.so.6").malloc_trim(0)
            except (OSError, AttributeError):
                pass
        if prep == "all":
            gc.freeze()
        self.steps(warmup)
        self.barrier()
        if self.dp is not None:
            self.dp.bucketer.exposed_ms()      # drop the warm-up steps' samples
        # five stream events inside the timed region (no host synchronisation): per-fifth step times, so that a one-off stall of the box
        # (seen twice in round 4: 30 ms inside one 50-step region, gone in the next run) shows in the line instead of hiding in the mean
        nchunk = min(5, steps)
        bounds = [steps * (i + 1) // nchunk for i in range(nchunk)]
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(nchunk + 1)]
        passes = []      # (generation, ms) of every collector pass that starts inside the timed region

        def on_gc(phase, info, _t=[0.0]):
            if phase == "start":
                _t[0] = time.perf_counter()
            else:
                passes.append((info["generation"], round(1e3 * (time.perf_counter() - _t[0]), 2)))
        gc.callbacks.append(on_gc)
        a0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
        t0 = time.perf_counter()
        evs[0].record()
        done = 0
        for i, upto in enumerate(bounds):
            self.steps(upto - done)
            done = upto
            evs[i + 1].record()
        self.barrier()
        dt = time.perf_counter() - t0
        self.chunk_ms = [round(evs[i].elapsed_time(evs[i + 1]) / max(1, bounds[i] - (bounds[i - 1] if i else 0)), 4) for i in range(nchunk)]
        gc.callbacks.remove(on_gc)
        self.gc_passes = [p for p in passes if p[0] >= 1 or p[1] >= 1.0]      # young-generation passes of microseconds are not worth a line
        gc.unfreeze()
        self.alloc_growth = torch.cuda.memory_stats().get("num_device_alloc", 0) - a0
        self.peak_gb = torch.cuda.max_memory_allocated() / 1e9      # peak of live tensors so far in this process (the engine keeps side-stream operands alive until the step's join)
        if self.alloc_growth:
            log(f"{self.alloc_growth} device allocation(s) INSIDE the timed region ({'joint' if self.joint else 'ctc'} model, {warmup} warm-up steps)")
            if strict:
                raise SystemExit("bench: the caching allocator grew inside a timed region - the number would include hipMalloc synchronisations")
        return dt

    def from_waveforms(self, nbatch=24, epochs=2):
        """ms/step when the batches come from WAVEFORMS instead of one resident batch: `nbatch` batches of args.batch synthetic 5-s utterances
        in host memory -> BucketedWaveLoader (pinned staging, two copies, log-mel / normalisation / SpecAugment on the loader's stream, a helper
        thread two batches ahead) -> iterate.  One warm-up epoch, then `epochs` timed ones.  Never `value`: the contract's inputs are resident."""
        import numpy as np
        torch = self.torch
        from asr_chinese_e2e_amd.data_handler import AudioParser, BucketedWaveLoader, WaveDataset
        rng = np.random.RandomState(0)
        B, S = self.args.batch, 16000 * 5
        items = [((rng.randn(S) * 0.1).astype(np.float32), [int(t) for t in rng.randint(4, self.args.vocab, size=16)]) for _ in range(B * nbatch)]
        dev = next(self.model.parameters()).device
        loader = BucketedWaveLoader(WaveDataset(items, self.model.vocab), B, parser=AudioParser(n_mels=80, lfr_m=1, lfr_n=1, device=dev), augment=True, shuffle=True,
                                    seed=1, dtype=torch.bfloat16, device=dev)

        def epoch():
            for pack in loader:
                self.model.iterate(pack, optimizer=self.opt, is_train=True)
        epoch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(epochs):
            epoch()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / (epochs * nbatch)

    def streams_independent(self):
        """True when main, weight-gradient, auxiliary and communication stream pairwise run side by side (engine.streams_conflict)."""
        from asr_chinese_e2e_amd import engine as E
        eng = self.model._engine
        ss = [self.torch.cuda.current_stream(), eng.side, eng.ctc_stream] + ([self.dp.bucketer.comm_stream] if self.dp is not None else [])
        self.torch.cuda.synchronize()
        return not any(E.streams_conflict(a, b) for i, a in enumerate(ss) for b in ss[i + 1:])

    def replica_checksum(self):
        """(min, max) over the ranks of the sum of all parameters after the timed steps: equal unless the replicas diverged
        (the bf16 wire format rounds, but every rank receives the SAME reduced bucket and applies the same update)."""
        torch = self.torch
        cs = self.model._flat.p.double().sum().reshape(1)
        lo, hi = cs.clone(), cs.clone()
        if torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
            torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        return float(lo), float(hi)

    def kernel_pass(self, n_inst, in_step=False):
        """The SAME steps once more with HIP events around every launch of the listed kernel families (recorded on
        the stream the kernel is launched on).  A separate pass: ~100 event pairs per step cost ~25 % wall time on
        ROCm.  in_step = False: the weight-gradient overlap is off, so the durations are stand-alone (what the roofline fractions are
        priced on); in_step = True: both streams live as in the timed region - a kernel's duration then includes what it loses to the
        kernels running beside it (the weight gradients take 51 us there, 40 alone)."""
        from asr_chinese_e2e_amd import kernels as K
        torch = self.torch
        timer = K.LaunchTimer(list(KERNEL_NAMES))
        K.TIMER = timer
        if not in_step:
            self.model._engine.overlap_wgrad = False
        # every GEMM of the step is an own kernel (round 3): count what still reaches the library through torch during these steps
        self.lib_gemm_calls = 0
        saved = {}

        def counting(fn):
            def wrapped(*a, **k):
                self.lib_gemm_calls += 1
                return fn(*a, **k)
            return wrapped
        import torch.nn.functional as F
        for owner, name in ((torch, "mm"), (torch, "addmm"), (torch, "matmul"), (torch, "bmm"), (torch, "baddbmm"), (torch, "einsum"), (torch, "mv"), (F, "linear"),
                            (torch.Tensor, "addmm_"), (torch.Tensor, "addmm"), (torch.Tensor, "matmul"), (torch.Tensor, "mm"), (torch.Tensor, "bmm"),
                            (torch.Tensor, "__matmul__"), (torch.Tensor, "__rmatmul__")):
            saved[(owner, name)] = getattr(owner, name)
            setattr(owner, name, counting(saved[(owner, name)]))
        try:
            self.steps(n_inst, runner=self.model if self.graphed else self.runner)   # events cannot be read back from a captured graph
            self.barrier()
        finally:
            for (owner, name), fn in saved.items():
                setattr(owner, name, fn)
            K.TIMER = None
            self.model._engine.overlap_wgrad = not self.model._engine.deterministic
        self.lib_gemm_calls_per_step = self.lib_gemm_calls / max(n_inst, 1)
        if self.lib_gemm_calls:
            raise SystemExit(f"bench: {self.lib_gemm_calls} library GEMM call(s) in {n_inst} steps - every projection is expected on an own kernel")
        return timer.summary()


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        launch_ranks(args)                      # never returns
    # stdout carries exactly ONE line, the JSON result: libraries that write to file descriptor 1 (RCCL prints a
    # version banner when its first communicator is created) are sent to stderr for the whole run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if env_world is not None and args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or unset WORLD_SIZE and let bench.py start them)")
    config = args.config or "joint"      # ONE workload at every N: the N = 8 value divided by the N = 1 value is the scaling of configs[2] -> configs[3]
    import torch

    if args.plumbing:                           # launcher test on a CPU-only machine
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        formed = dist.get_world_size()
        dist.barrier()
        if rank == 0:
            os.write(result_fd, (json.dumps({"metric": "plumbing", "value": 0.0, "unit": "utterances/s", "n_gpus": formed, "steps": args.steps,
                                             "warmup": args.warmup, "sum_of_ranks": float(t), "config": {"workload": config,
                                             "global_batch": formed * args.batch, "parallelism": f"dp{formed}"}}) + "\n").encode())
        dist.destroy_process_group()
        return

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the measured path)")
    backend = os.environ.get("ASR_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_idx = local if backend == "nccl" else local % max(ndev, 1)   # "gloo": rehearse N ranks on a one-GPU box (they share the card)
    torch.cuda.set_device(dev_idx)
    dev = torch.device("cuda", dev_idx)
    use_dp = world > 1 or os.environ.get("ASR_FORCE_DP") == "1"    # ASR_FORCE_DP: exercise the RCCL path with one rank
    if use_dp:
        from asr_chinese_e2e_amd import dist as D
        os.environ["LOCAL_RANK"] = str(dev_idx)
        # The first N > 1 run on xGMI is also the first execution of the RCCL path anywhere (the build box has one GPU): if the
        # process group or its first collective fails, say why on stderr and exit non-zero - this process has initialised the
        # GPU, so it never re-execs or falls back to another backend.
        try:
            D.init(backend)
            world = torch.distributed.get_world_size()      # the ranks the process group actually formed
        except Exception as e:      # noqa: BLE001 - reported verbatim, then the run ends
            log(f"rank {rank}: process group FAILED over backend {backend}: {type(e).__name__}: {e}")
            log("environment: " + ", ".join(f"{k}={os.environ.get(k)}" for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                                                  "HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG")))
            raise SystemExit(3)

    # The model (and with it the engine's streams) is built BEFORE the first collective: the process group takes its internal stream from
    # torch's pool at that moment, and dist.DataParallel first leaves the pool in front of a stream that runs beside the compute streams
    # (engine.steer_stream_pool).  The first collectives - the wrapper's parameter broadcast, then a one-element all-reduce - are checked here.
    try:
        run = Run(args, config, args.dropout, rank, dev, use_dp)
        if use_dp:
            probe = torch.ones(1, device=dev)
            torch.distributed.all_reduce(probe)
            torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise RuntimeError(f"first all-reduce summed {probe.item()} over a group of {world}")
    except Exception as e:      # noqa: BLE001
        if not use_dp:
            raise
        log(f"rank {rank}: first collective FAILED over backend {backend}: {type(e).__name__}: {e}")
        log("environment: " + ", ".join(f"{k}={os.environ.get(k)}" for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                                              "HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG")))
        raise SystemExit(3)

    log(f"rank {rank}/{world}: {config} model on {dev}, warm-up {args.warmup} steps")
    dt = run.timed(args.warmup, args.steps, strict=(world == 1))      # N > 1 has never run on xGMI: count and report, do not end the run
    alloc_growth = run.alloc_growth
    headline_peak = run.peak_gb
    headline_chunks = run.chunk_ms
    headline_gc = run.gc_passes
    log(f"timed region done: {1e3 * dt / args.steps:.2f} ms/step")
    summary, n_inst = None, min(args.steps, 10)
    lib_gemm_per_step = None
    in_step = None
    if not args.no_kernel_timer:
        summary = run.kernel_pass(n_inst)
        lib_gemm_per_step = run.lib_gemm_calls_per_step
        in_step = run.kernel_pass(n_inst, in_step=True)
    if use_dp:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    loss = float(run.last.loss)
    joint, graphed = run.joint, run.graphed
    wire = None
    if run.dp is not None:
        # comm_exposed_ms: time per step the compute stream waited for the communication stream AFTER backward had finished (HIP
        # events around GradBucketer.finish()'s wait; mean over the timed steps, MAX over ranks) = all-reduce time backward did not hide
        ex = torch.tensor([run.dp.bucketer.exposed_ms(reset=False) or 0.0], device=dev, dtype=torch.float64)
        if world > 1:
            torch.distributed.all_reduce(ex, op=torch.distributed.ReduceOp.MAX)
        lo, hi = run.replica_checksum()
        esz = 2 if run.dp.bucketer.wire is not None else 4
        wire = {"bytes_per_step": run.dp.bucketer.bytes_on_wire, "buckets": len(run.dp.bucketer.buckets),
                "bucket_bytes": [int((e - s) * esz) for s, e in run.dp.bucketer.buckets],
                "dtype": "bf16" if run.dp.bucketer.wire is not None else "fp32", "backend": backend, "comm_exposed_ms": float(ex),
                "comm_exposed_note": "time the compute stream waited for the communication stream after backward; only meaningful under the "
                                     "RCCL backend (gloo blocks the host instead)",
                # replicas must stay identical: every rank receives the same reduced buckets and applies the same fused update
                "replica_param_checksum_min": lo, "replica_param_checksum_max": hi, "replicas_identical": lo == hi,
                # the step's streams were chosen by measurement (engine.pick_stream): each runs beside the others (no shared hardware queue / pipe)
                "streams_independent": run.streams_independent(), "pool_draws_before_first_collective": run.dp.pool_draws,
                "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                "nccl_env": {k: os.environ.get(k) for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS", "RCCL_MSCCL_ENABLE")
                             if os.environ.get(k) is not None}}
        if lo != hi:
            log(f"REPLICAS DIVERGED: parameter checksum min {lo!r} max {hi!r} over {world} ranks")

    extras = {}
    if world == 1 and not use_dp and not args.no_extras and rank == 0:
        # the other single-GPU configurations, timed the same way (shorter).
        # The blocks of the finished Run go back to the caching allocator and the next Run takes them from there: NO
        # torch.cuda.empty_cache() in between.  Round 3 had one, and the driver's numbers for these keys were 12 - 38 % above the
        # 200 / 30 runs: after the hipFree of ~3 GB every following configuration ran 6 - 10 steps at 2 - 2.5x the step time
        # (Tensor.record_stream in the engine kept freed blocks unusable until an event poll; the pool had to regrow by hipMalloc)
        # and, with that fixed, still showed one 10 - 20 ms stall a few steps in (tools/alloc_diag.py; absent when nothing was freed).
        import gc
        del run
        gc.collect()
        es, ew = max(10, min(args.steps, 50)), max(10, min(args.warmup, 15))
        extras_alloc, extras_chunks, extras_gc, extras_peak = {}, {}, {}, {}

        def extra(key, cfg_name, dropout, batch=None, frames=None, window=None, kernels=False):
            torch.cuda.reset_peak_memory_stats()
            r = Run(args, cfg_name, dropout, rank, dev, False, batch=batch, frames=frames, window=window)
            d = r.timed(ew, es)
            extras[f"{key}_ms_per_step"] = 1e3 * d / es
            extras[f"{key}_utterances_per_s"] = (batch or args.batch) * es / d
            if kernels and not args.no_kernel_timer:
                ks = r.kernel_pass(min(es, 5))
                extras[f"{key}_kernels"] = {k: {"avg_us": v["avg_us"], "launches_per_step": v["launches"] / min(es, 5)}
                                            for k, v in ks.items() if k in ("sdpa_fwd", "sdpa_bwd", "ctc", "gemm_nt", "gemm_tn")}
            extras_alloc[key], extras_chunks[key], extras_gc[key] = r.alloc_growth, r.chunk_ms, r.gc_passes
            extras_peak[key] = round(r.peak_gb, 2)
            del r
            gc.collect()

        other = "joint" if config == "ctc" else "ctc"
        extra(other, other, args.dropout)      # configs[1] (6-layer encoder + CTC-only) when the headline is the joint model
        if args.dropout == 0.0:      # the reference's recipe (dropout 0.1, transformer_official.py:115-122) on both models
            extra("dropout_0.1", config, 0.1)
            extra(f"{other}_dropout_0.1", other, 0.1)
            extras["dropout_0.1_config"] = f"dropout_0.1_*: the headline ({config}) model with dropout 0.1; {other}_dropout_0.1_*: the {other} model"
        # BASELINE.json configs[4] per GPU: long-form utterances (T = 2000 frames, +-50-frame attention band, batch 8), joint model
        extra("long_form", "joint", args.dropout, batch=8, frames=2000, window=50, kernels=True)
        extras["long_form_config"] = "configs[4] per GPU: joint CTC/attention, B=8, T=2000, attention band +-50 frames, bf16"
        # the configurations fed from waveforms through the loader (host staging + front end included): never `value`
        for name in (config, other):
            r5 = Run(args, name, args.dropout, rank, dev, False)
            extras[f"{name}_from_waveforms_ms_per_step"] = r5.from_waveforms()
            del r5
            gc.collect()
        extras["from_waveforms_config"] = "24 batches per epoch of synthetic 5-s utterances in host memory, SpecAugment on, BucketedWaveLoader (helper thread, 2 ahead); 1 warm-up + 2 timed epochs"
        extras["extras_protocol"] = f"{ew} warm-up + {es} timed steps each, caching allocator kept between configurations"
        extras["allocator_growth_in_timed_regions"] = dict(extras_alloc, headline=alloc_growth)
        extras["ms_per_step_by_fifth_of_each_timed_region"] = dict(extras_chunks, headline=headline_chunks)
        extras["collector_passes_in_timed_regions"] = dict(extras_gc, headline=headline_gc)      # (generation, ms) each; [] = none
        # live-tensor peak (GB) while each configuration ran (torch.cuda.max_memory_allocated, reset per configuration): includes the operands of side-stream
        # kernels that the engine holds until the step's join instead of marking them with Tensor.record_stream
        extras["peak_memory_allocated_gb"] = dict(extras_peak, headline=round(headline_peak, 2))
        log(f"extras: {extras}")

    if rank == 0:
        utt = world * args.batch * args.steps / dt
        step_ms = 1e3 * dt / args.steps
        out = {
            "metric": "training throughput (utterances/s; frames/s = x T), AISHELL-1-shaped 80-mel T=500",
            "value": utt, "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic", "ranks_formed": world, "allocator_growth_in_timed_region": alloc_growth,
            "frames_per_s": utt * args.frames, "final_loss": loss,
            "ms_per_step_by_fifth": headline_chunks,      # rank 0's stream events inside the timed region: a stall of the box shows as one outlier
            "config": {"workload": (("configs[3]: data-parallel " if world > 1 else "configs[2]: ") + "joint CTC/attention (lambda=0.3) encoder-decoder" if joint else
                                    "configs[1]: 6-layer Transformer encoder + CTC-only") +
                                   f", bf16, per-GPU batch {args.batch}, T={args.frames}, F=80, V={args.vocab}, "
                                   f"{args.layers} layers, d_model 512, 8x64 heads, ff 1024, dropout {args.dropout}, "
                                   "fwd+loss+" + ("" if args.no_cer else "CER+") + "bwd+" + ("bucketed gradient all-reduce+" if use_dp else "") + "clip+Noam/Adam per step" +
                                   (", one hipGraph per step" if graphed else ", eager launches"),
                       "global_batch": world * args.batch, "seq_len": args.frames, "parallelism": f"dp{world}",
                       # north_star's "CTC loss matching reference to 1e-4 rel" is a property of the loss KERNELS (tests/test_kernels_gpu.py::test_ctc,
                       # incl. T = 2000) and of the fp32 parity mode end to end (8.6e-9); the bf16 model this line times carries the rounding of
                       # its bf16 encoder activations into the loss: gate 1e-3 rel, measured 2.5e-4 .. 4.9e-4 up to T = 500 and
                       # 9.3e-4 at T = 2000 (profiles/round4_ctc_parity_diag.txt)
                       "parity": {"ctc_loss_rel_tolerance_kernels_and_fp32_mode": 1e-4, "bf16_model_loss_rel_gate": 1e-3,
                                  "bf16_model_loss_rel_measured": "2.5e-4 .. 9.3e-4",
                                  "this_workload_at_full_size_vs_oracle": "tests/test_model_gpu.py::test_full_size_step_matches_oracle: loss 5e-6 (joint) / 1.3e-4 (ctc), "
                                  "every significant gradient tensor cosine >= 0.9993 / 0.9985; >= 0.99987 against the oracle on bf16-rounded weight matrices"}},
        }
        if wire is not None:
            out["all_reduce"] = wire
        out.update(extras)
        if summary:
            # counted by kernel_pass over its instrumented steps (torch.mm / addmm / matmul / bmm / baddbmm / einsum / F.linear / the @
            # operator are wrapped there; a non-zero count ends the run): the value measured, not a literal
            out["library_gemm_calls_per_step"] = lib_gemm_per_step
            # roofline: priced on the condition of the TIMED region - the in-step pass (weight-gradient / auxiliary streams live beside the main
            # stream); the stand-alone pass (overlap off) of the same launches is `frac_standalone`.  Both GEMM families are on the line.
            src = in_step or summary
            fam = {k: v for k, v in src.items() if k in ("gemm_nt", "gemm_tn")}
            order = sorted(fam, key=lambda k: -fam[k]["total_ms"])

            def roof(k):
                a = fam[k]["work_per_s"] / 1e12
                alone = summary[k]["work_per_s"] / 1e12 if k in summary else None
                prof = profiled_in_step_us(k, config)
                prof_obj = None
                if prof:      # the same algorithmic FLOP over the committed trace's in-step kernel durations
                    prof_tf = fam[k]["work_per_launch"] * fam[k]["launches"] / n_inst / (prof[0] * prof[1] * 1e-6) / 1e12
                    prof_obj = {"file": prof[2], "launches_per_step": prof[0], "avg_launch_us": prof[1], "frac": prof_tf / MFMA_BF16_PEAK_TFLOPS}
                return {"bound": "mfma", "kernel": KERNEL_NAMES[k], "achieved": a, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": a / MFMA_BF16_PEAK_TFLOPS,
                        "committed_rocprofv3_trace": prof_obj,
                        "condition": "in the step: HIP events on the launch stream with the weight-gradient / auxiliary streams live, as in the timed region" if in_step else
                                     "stand-alone (weight-gradient overlap off)",
                        "frac_standalone": None if alone is None else alone / MFMA_BF16_PEAK_TFLOPS,
                        "traffic": pmc_traffic(k), "traffic_source": os.path.relpath(PMC_TRAFFIC_FILE, ROOT) + ": rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE passes of this command (separate runs), bytes per launch averaged over the family's launches, FETCH_SIZE x2 (gfx950 correction)",
                        "avg_launch_us": fam[k]["avg_us"], "avg_launch_us_standalone": summary[k]["avg_us"] if k in summary else None,
                        "launches": fam[k]["launches"], "flop_per_launch": fam[k]["work_per_launch"],
                        "share_of_step": fam[k]["total_ms"] / n_inst / step_ms}
            if order:
                out["roofline"] = roof(order[0])
            if len(order) > 1:
                out["roofline_other_gemm_family"] = roof(order[1])
            table = {}
            for k, v in summary.items():
                row = {"kernel": KERNEL_NAMES[k], "launches_per_step": v["launches"] / n_inst, "avg_us": v["avg_us"],
                       "share_of_step": v["total_ms"] / n_inst / step_ms}
                if v.get("bytes_per_launch"):
                    gbs = v["bytes_per_s"] / 1e9
                    row.update(bytes_per_launch=v["bytes_per_launch"], hbm_gbs=gbs, hbm_frac=gbs / HBM_PEAK_GBS)
                if k in MATRIX_FAMILIES:
                    tf = v["work_per_s"] / 1e12
                    row.update(flop_per_launch=v["work_per_launch"], tflops=tf, mfma_frac=tf / MFMA_BF16_PEAK_TFLOPS)
                if in_step and k in in_step:
                    row["in_step_avg_us"] = in_step[k]["avg_us"]
                table[k] = row
            out["kernels"] = table
            out["kernels_note"] = ("HIP events on the launch stream.  avg_us and the fractions: weight-gradient overlap off (stand-alone durations); "
                                   "in_step_avg_us: a second instrumented pass with both streams live, as in the timed region (includes what a kernel loses to "
                                   "the kernels beside it); bytes / FLOP per launch are the algorithmic figures of SURVEY.md 8(d) / DESIGN.md section 4; "
                                   "hbm_frac against 8 TB/s, mfma_frac against 2.5 PFLOP/s dense bf16")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, config)      # the SAME workload as `value`
            if not args.no_extras:      # and the other model of the extra keys, on a smaller sample
                other = "joint" if config == "ctc" else "ctc"
                out[f"cpu_baseline_{other}"] = cpu_baseline(args, other, warm=2, timed=3)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if use_dp:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
