"""Per-step timeline of a rocprofv3 --kernel-trace csv: busy time per queue, their union, idle gaps, and the kernels in flight.
python tools/timeline.py kernel_trace.csv [first_step last_step]   (steps are delimited by the adam kernel)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (len(ends) - 4, len(ends) - 2)
for s in range(lo, hi):
    seg = rows[ends[s] + 1: ends[s + 1] + 1]
    t0 = int(seg[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in seg)
    perq = collections.defaultdict(list)
    for r in seg:
        perq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    def union(iv):
        iv = sorted(iv); tot = 0; cs, ce = iv[0][0], iv[0][1]
        for a, b, *_ in iv[1:]:
            if a > ce: tot += ce - cs; cs, ce = a, b
            else: ce = max(ce, b)
        return tot + ce - cs
    allv = [x for v in perq.values() for x in v]
    print(f"step {s}: span {1e-3 * (t1 - t0):8.1f} us, {len(seg)} kernels; union busy {1e-3 * union(allv):8.1f} us")
    for qid, v in perq.items():
        print(f"   queue {qid}: {len(v):4d} kernels, busy {1e-3 * union(v):8.1f} us, sum {1e-3 * sum(b - a for a, b, _ in v):8.1f} us")
    if len(sys.argv) > 4:   # dump
        for r in seg:
            print(f"      {1e-3 * (int(r['Start_Timestamp']) - t0):8.1f} {1e-3 * (int(r['End_Timestamp']) - int(r['Start_Timestamp'])):7.1f} q{r['Queue_Id']} {short(r['Kernel_Name'])}")
