"""Per-kernel totals per step from a rocprofv3 --kernel-trace csv (steps delimited by the adam kernel):
python tools/trace_stats.py kernel_trace.csv [out.csv]  -> name, calls/step, avg us, total us/step, share"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
seg = rows[ends[len(ends) // 2] + 1: ends[-1] + 1]
nsteps = len(ends) - 1 - len(ends) // 2
agg = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0][:70]
    agg[n][0] += 1
    agg[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
tot = sum(v[1] for v in agg.values())
lines = [("Name", "CallsPerStep", "AverageUs", "TotalUsPerStep", "Percentage")]
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append((n, f"{c / nsteps:.1f}", f"{t / c:.2f}", f"{t / nsteps:.1f}", f"{100 * t / tot:.2f}"))
for l in lines[:40]:
    print(f"{l[0]:72s} {l[1]:>8s} {l[2]:>9s} {l[3]:>10s} {l[4]:>7s}")
print(f"sum of kernel time per step: {tot / nsteps:.1f} us over {nsteps} steps")
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as f:
        w = csv.writer(f)
        w.writerows(lines)
