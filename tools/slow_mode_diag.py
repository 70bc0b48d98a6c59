#!/usr/bin/env python3
"""Does a configuration's step time change while it runs?  Builds the bench's Runs in bench.py's order (no empty_cache between them) and times
each in chunks of 10 un-synchronised steps: python tools/slow_mode_diag.py [chunks]   (on the GPU box)"""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench


def main():
    chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    args = bench.parse([])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    for tag, kw in (("ctc", dict(config="ctc", dropout=0.0)), ("joint", dict(config="joint", dropout=0.0)), ("ctc dropout 0.1", dict(config="ctc", dropout=0.1)),
                    ("long-form", dict(config="joint", dropout=0.0, batch=8, frames=2000, window=50)), ("ctc dropout 0.1 again", dict(config="ctc", dropout=0.1))):
        run = bench.Run(args, kw["config"], kw["dropout"], 0, dev, False, batch=kw.get("batch"), frames=kw.get("frames"), window=kw.get("window"))
        run.steps(15)
        torch.cuda.synchronize()
        out = []
        a0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
        for _ in range(chunks):
            t0 = time.perf_counter()
            run.steps(10)
            torch.cuda.synchronize()
            out.append(1e2 * (time.perf_counter() - t0))
        grew = torch.cuda.memory_stats().get("num_device_alloc", 0) - a0
        print(f"{tag:24s} ms/step per chunk of 10: " + " ".join(f"{x:.3f}" for x in out) + f"   device allocations meanwhile: {grew}", flush=True)
        del run
        gc.collect()


if __name__ == "__main__":
    main()
