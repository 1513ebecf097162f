"""profiles/*_pmc_valu.json from one rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES pass
(counter_collection csv):  python tools/pmc_valu.py pass.csv frames_per_launch "about text" > out.json
SQ_INSTS_VALU = wave64 VALU instructions issued.  valu_issue_frac = the share of the full-rate issue slots the kernel uses
while it runs = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x SQ_BUSY_CYCLES / 32) = INSTS / (16 x BUSY): a wave64 VALU instruction
issues over 2 cycles at full rate (MI355X_MICROARCH.md), SQ_BUSY_CYCLES is in cycles of the clock the chip actually ran at,
summed over the 32 SQ instances (8 XCD x 4 SE).  Half-rate instructions (v_min/v_max/v_cmp/v_perm/SDWA/VOP3 integer) take two
slots each, so a kernel made of them saturates at 0.5.  (SQ_ACTIVE_INST_VALU turned out to be ~1 quad-cycle per instruction
whatever its rate, i.e. the same number as SQ_INSTS_VALU: it is kept in the file but not used as a busy fraction.)"""
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
        out["kernels"][k]["valu_issue_frac"] = round(v.get("SQ_INSTS_VALU", 0) / (16.0 * v["SQ_BUSY_CYCLES"]), 4)
print(json.dumps(out, indent=1))
