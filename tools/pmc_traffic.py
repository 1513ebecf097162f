"""profiles/*_pmc_hbm_traffic.json from the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter_collection csv):
python tools/pmc_traffic.py fetch.csv write.csv frames_per_launch "about text" > out.json"""
import collections, csv, json, sys


def per_kernel(path, counter):
    tot = collections.defaultdict(float); n = collections.Counter(); seen = set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        tot[k] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); n[k] += 1
    return tot, n


f, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
w, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"meta": {"_about": sys.argv[4], "frames_per_launch": int(sys.argv[3]), "w": 1241, "h": 376}, "kernels": {}}
for k in sorted(set(f) | set(w)):
    out["kernels"][k] = {"FETCH_SIZE_KB_avg_per_launch": round(f[k] / max(nf[k], 1), 1), "WRITE_SIZE_KB_avg_per_launch": round(w[k] / max(nw[k], 1), 1),
                         "launches": int(max(nf[k], nw[k]))}
print(json.dumps(out, indent=1))
