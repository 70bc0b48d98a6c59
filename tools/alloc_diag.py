#!/usr/bin/env python3
"""Per-step wall time and caching-allocator activity of the bench's Runs, in the order bench.py builds them
(ctc -> joint -> dropout 0.1 -> long-form, torch.cuda.empty_cache() between): which steps still call hipMalloc / grow segments.
usage: python tools/alloc_diag.py [steps]   (on the GPU box)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench


def stats():
    s = torch.cuda.memory_stats()
    return (s.get("num_device_alloc", 0), s.get("num_device_free", 0), s.get("segment.all.current", 0), s.get("reserved_bytes.all.current", 0) >> 20,
            s.get("allocated_bytes.all.peak", 0) >> 20, s.get("num_alloc_retries", 0))


def trace(tag, run, n):
    print(f"== {tag}: step  ms   device_allocs device_frees segments reserved_MiB peak_alloc_MiB retries", flush=True)
    prev = stats()
    for i in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run.steps(1)
        torch.cuda.synchronize()
        dt = 1e3 * (time.perf_counter() - t0)
        cur = stats()
        flag = " <-- allocator grew" if cur[0] != prev[0] else ""
        if i >= 14 and cur[0] == prev[0]:
            prev = cur
            continue
        print(f"   {i:3d} {dt:8.3f}  {cur[0]:6d} {cur[1]:6d} {cur[2]:5d} {cur[3]:7d} {cur[4]:7d} {cur[5]:3d}{flag}", flush=True)
        prev = cur
    # un-synchronised steady-state time, as bench.py measures it
    t0 = time.perf_counter()
    run.steps(20)
    torch.cuda.synchronize()
    print(f"   steady state (20 steps, no per-step sync): {1e3 * (time.perf_counter() - t0) / 20:.3f} ms/step", flush=True)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    args = bench.parse([])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    for tag, kw in (("ctc", dict(config="ctc", dropout=0.0)), ("joint", dict(config="joint", dropout=0.0)), ("ctc dropout 0.1", dict(config="ctc", dropout=0.1)),
                    ("long-form", dict(config="joint", dropout=0.0, batch=8, frames=2000, window=50))):
        run = bench.Run(args, kw["config"], kw["dropout"], 0, dev, False, batch=kw.get("batch"), frames=kw.get("frames"), window=kw.get("window"))
        if float(os.environ.get("DIAG_SLEEP", "0")) > 0:
            torch.cuda.synchronize()
            time.sleep(float(os.environ["DIAG_SLEEP"]))
        trace(tag, run, n)
        del run
        if os.environ.get("DIAG_EMPTY", "1") == "1":
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
