"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), per launch.
Units and gfx950 correction follow MI355X_MICROARCH.md (HBM section): counter values are KiB;
FETCH_SIZE reports exactly half the bytes of a wide coalesced streaming read on gfx950, so the
read side is doubled (marked 'corrected'); WRITE_SIZE is exact for 16-B-per-lane stores and atomics.
usage: python tools/pmc_summary.py FETCH_csv WRITE_csv out.json"""
import collections, csv, json, sys


def load(path):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
        key = name.split("(")[0][:60]
        agg[key][0] += float(r["Counter_Value"])
        agg[key][1] += 1
    return agg


f, w = load(sys.argv[1]), load(sys.argv[2])
out = {}
for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, [0, 1])[0] + w.get(k, [0, 1])[0])):
    fr, fn = f.get(k, [0.0, 1])
    wr, wn = w.get(k, [0.0, 1])
    rd = 2.0 * fr * 1024 / max(fn, 1)
    wrb = wr * 1024 / max(wn, 1)
    out[k] = {"launches": fn, "read_bytes_per_launch_corrected": rd, "write_bytes_per_launch": wrb, "hbm_bytes_per_launch": rd + wrb}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in list(out.items())[:22]:
    print(f"{k:60s} n={v['launches']:4d} read {v['read_bytes_per_launch_corrected'] / 1e6:8.2f} MB  write {v['write_bytes_per_launch'] / 1e6:8.2f} MB")
