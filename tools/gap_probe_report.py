import sqlite3, statistics, sys
c = sqlite3.connect(sys.argv[1])
t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [x for x in t if "kernel_dispatch" in x][0]; ks = [x for x in t if "kernel_symbol" in x][0]
rows = list(c.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
def kind(n):
    if n.startswith("Cijk"): return "lib"
    if "gemm_nt" in n: return "ownNT"
    if "relu" in n: return "ownRelu"
    if "elementwise" in n or "vectorized" in n: return "torchEW"
    return "other"
pairs = {}
for p, q in zip(rows[:-1], rows[1:]):
    pairs.setdefault((kind(p[2]), kind(q[2])), []).append((q[0] - p[1]) / 1e3)
for k, v in sorted(pairs.items()):
    if len(v) >= 20:
        v.sort()
        print(k, len(v), "median %.2f  p10 %.2f  p90 %.2f" % (statistics.median(v), v[len(v) // 10], v[9 * len(v) // 10]))
