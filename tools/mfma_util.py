"""MFMA pipe utilisation per kernel from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE).
SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over the SIMDs (32 per v_mfma_f32_32x32x16_bf16);
GRBM_GUI_ACTIVE is the sum over the 8 XCDs of the cycles the kernel was resident (MI355X_MICROARCH.md), so
utilisation = BUSY / (GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs).   usage: python tools/mfma_util.py counters.csv out.json"""
import collections, csv, json, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    key = name.split("(")[0][:60]
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[key] += 1
out = {}
for k, v in agg.items():
    busy, act = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("GRBM_GUI_ACTIVE", 0.0)
    if busy > 0 and act > 0:
        out[k] = {"launches": cnt[k], "mfma_busy_cycles_per_launch": busy / max(cnt[k], 1), "resident_cycles_per_launch": act / 8 / max(cnt[k], 1),
                  "mfma_utilisation": busy / (act / 8 * 256 * 4)}
json.dump(dict(sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_launch"] * kv[1]["launches"])), open(sys.argv[2], "w"), indent=1)
for k, v in list(out.items())[:14]:
    print(f"{k:60s} n={v['launches']:4d} util {100 * v['mfma_utilisation']:5.1f} %")
