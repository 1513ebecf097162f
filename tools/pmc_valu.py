"""profiles/*_pmc_valu.json from one rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES pass
(counter_collection csv):  python tools/pmc_valu.py pass.csv frames_per_launch "about text" > out.json
SQ_INSTS_VALU = wave64 VALU instructions issued.  valu_busy = the fraction of SIMD-cycles in which the VALU executes while
the kernel runs = SQ_ACTIVE_INST_VALU (quad-cycles, summed over the 1024 SIMDs; MI355X_MICROARCH.md constants table) x 4 /
(SQ_BUSY_CYCLES (cycles, summed over the 32 SQ instances = 8 XCD x 4 SE) / 32 x 1024) = ACTIVE / (8 x BUSY)."""
import collections, csv, json, sys

tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); n[k] += 1
out = {"meta": {"_about": sys.argv[3], "frames_per_launch": int(sys.argv[2]), "w": 1241, "h": 376}, "kernels": {}}
for k in sorted(tot):
    v = tot[k]; m = max(n[k], 1)
    out["kernels"][k] = {"launches": int(n[k]), "SQ_INSTS_VALU_avg_per_launch": round(v.get("SQ_INSTS_VALU", 0) / m, 1),
                         "SQ_ACTIVE_INST_VALU_avg_per_launch": round(v.get("SQ_ACTIVE_INST_VALU", 0) / m, 1),
                         "SQ_WAVE_CYCLES_avg_per_launch": round(v.get("SQ_WAVE_CYCLES", 0) / m, 1),
                         "SQ_BUSY_CYCLES_avg_per_launch": round(v.get("SQ_BUSY_CYCLES", 0) / m, 1)}
    if v.get("SQ_BUSY_CYCLES"):
        out["kernels"][k]["valu_busy"] = round(v.get("SQ_ACTIVE_INST_VALU", 0) / (8.0 * v["SQ_BUSY_CYCLES"]), 4)
print(json.dumps(out, indent=1))
