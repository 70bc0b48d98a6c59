"""Which HIP streams share a hardware queue?  A stream whose kernels sit BEHIND another stream's kernels in one in-order hardware queue cannot
overtake them: launch a long spin kernel on stream A, then a tiny kernel on stream B and wait for B - if that takes as long as the spin, B shares
A's queue.  Prints, for the null stream and N fresh torch pool streams (+ one low-priority stream of the library), who is blocked behind whom.
python tools/queue_probe.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_chinese_e2e_amd import engine as E

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
buf = torch.zeros(64, device=dev)
spin_cycles = int(os.environ.get("SPIN", "4000000"))


def blocked(a, b):
    """ms until a tiny kernel on b completes while a long spin runs on a."""
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        torch.cuda._sleep(spin_cycles)
    ev = torch.cuda.Event()
    with torch.cuda.stream(b):
        buf.add_(1.0)
        ev.record()
    t0 = time.perf_counter()
    ev.synchronize()
    dt = 1e3 * (time.perf_counter() - t0)
    torch.cuda.synchronize()
    return dt


null = torch.cuda.default_stream()
t0 = time.perf_counter(); torch.cuda._sleep(spin_cycles); torch.cuda.synchronize(); spin_ms = 1e3 * (time.perf_counter() - t0)
t0 = time.perf_counter(); torch.cuda._sleep(spin_cycles); torch.cuda.synchronize(); spin_ms = 1e3 * (time.perf_counter() - t0)
print(f"spin of {spin_cycles} cycles = {spin_ms:.2f} ms")
streams = [("null", null)]
for i in range(n):
    streams.append((f"pool{i}", torch.cuda.Stream()))
streams.append(("low", E._side_stream(dev)))
names = [s[0] for s in streams]
print("rows: stream running the spin; columns: stream of the tiny kernel; X = the tiny kernel waited for the spin (same hardware queue)")
print("        " + " ".join(f"{x:>6s}" for x in names))
for na, a in streams:
    row = []
    for nb, b in streams:
        if a is b:
            row.append("     -")
            continue
        dt = blocked(a, b)
        row.append("     X" if dt > 0.5 * spin_ms else "     .")
    print(f"{na:>6s}  " + " ".join(row), flush=True)

# ---- second test: two busy queues served by the SAME command-processor pipe?  A chain of N short spin kernels (one workgroup each, ~20 us) goes to
# each of two streams at the same time: on separate pipes the two chains run side by side (time of one chain), otherwise they alternate.
SP = int(os.environ.get("SPIN2", "50000"))


def chains(ss, n=100):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for s in ss:
            with torch.cuda.stream(s):
                torch.cuda._sleep(SP)
    th = 1e3 * (time.perf_counter() - t0)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0), th


base, th = min(chains([null]) for _ in range(3))
print(f"\n100 spin kernels of {SP} cycles on the null stream alone: {base:.2f} ms (host enqueue {th:.2f}).  Below: 100 on EACH of two streams at once, ms")
print("        " + " ".join(f"{x:>6s}" for x in names))
for na, a in streams:
    row = []
    for nb, b in streams:
        if a is b:
            row.append("     -")
            continue
        row.append(f"{min(chains([a, b])[0] for _ in range(2)):6.2f}")
    print(f"{na:>6s}  " + " ".join(row), flush=True)
