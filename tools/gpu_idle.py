"""GPU idle time per training step from a rocprofv3 --kernel-trace csv: span of a step (adam to adam), union of the
busy intervals over all streams, and the gaps by size - tells a launch-bound step from a kernel-bound one.
python tools/gpu_idle.py kernel_trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
lo, hi = ends[len(ends) // 2], ends[-1]
seg = rows[lo + 1: hi + 1]
nsteps = len(ends) - 1 - len(ends) // 2
t0, t1 = int(rows[lo]["End_Timestamp"]), int(rows[hi]["End_Timestamp"])
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
busy, cur_s, cur_e, gaps = 0, iv[0][0], iv[0][1], []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = t1 - t0
print(f"steps {nsteps}: span {span / nsteps / 1e3:.1f} us/step, busy (union over streams) {busy / nsteps / 1e3:.1f} us/step, idle {(span - busy) / nsteps / 1e3:.1f} us/step")
print(f"kernels/step {len(seg) / nsteps:.0f}, sum of durations {sum(e - s for s, e in iv) / nsteps / 1e3:.1f} us/step")
for lo_, hi_ in ((0, 2), (2, 5), (5, 10), (10, 20), (20, 1e9)):
    g = [x for x in gaps if lo_ * 1e3 <= x < hi_ * 1e3]
    print(f"  gaps {lo_:>3}-{hi_ if hi_ < 1e8 else 'inf':>3} us: {len(g) / nsteps:6.1f} per step, {sum(g) / nsteps / 1e3:7.1f} us per step")
