"""Summarise a rocprofv3 kernel_stats.csv per training step: python tools/prof_summary.py FILE NSTEPS"""
import csv, sys
f, n = sys.argv[1], float(sys.argv[2])
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"GPU busy {tot / n / 1e6:.3f} ms/step over {n:.0f} steps")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    print(f"{name[:64]:64s} calls/step {int(r['Calls']) / n:6.1f}  avg {float(r['AverageNs']) / 1e3:8.1f} us  {int(r['TotalDurationNs']) / n / 1e6:6.3f} ms/step  {float(r['Percentage']):5.1f}%")
