"""rocprofv3 results .db (rocpd) -> kernel stats CSV with the columns of `rocprofv3 --stats` (kernel_stats.csv):
python tools/db_to_stats_csv.py FILE.db OUT.csv"""
import csv, math, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [x for x in t if "kernel_dispatch" in x][0]
ks = [x for x in t if "kernel_symbol" in x][0]
rows = {}
for name, dur in c.execute(f"select s.kernel_name, d.end-d.start from {kd} d join {ks} s on d.kernel_id=s.id"):
    rows.setdefault(name, []).append(dur)
tot = sum(sum(v) for v in rows.values())
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        n, s = len(v), sum(v)
        mean = s / n
        sd = math.sqrt(sum((x - mean) ** 2 for x in v) / (n - 1)) if n > 1 else 0.0
        w.writerow([name, n, s, round(mean, 3), round(100.0 * s / tot, 4), min(v), max(v), round(sd, 3)])
