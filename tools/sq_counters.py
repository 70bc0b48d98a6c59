"""Per-kernel SQ counter summary from rocprofv3 --pmc csv files (any number of passes):
python tools/sq_counters.py out.json pass1_counter_collection.csv [pass2 ...]
Prints, per kernel, counter sums per launch.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves,
SQ_VALU_MFMA_BUSY_CYCLES cycles summed over SIMDs (MI355X_MICROARCH.md)."""
import collections, csv, json, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
        key = name.split("(")[0][:48]
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[key][r["Counter_Name"]] += 1
out = {k: {c: v / cnt[k][c] for c, v in d.items()} for k, d in agg.items()}
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, d in out.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:32s} {v:16.0f}")
