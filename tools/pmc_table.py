import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set(); names = []
for r in rows:
    k = r['Kernel_Name'].split('(')[0][:28]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] not in names: names.append(r['Counter_Name'])
    if r['Dispatch_Id'] not in seen: seen.add(r['Dispatch_Id']); cnt[k] += 1
print('kernel'.ljust(28), 'disp', *[n.replace('SQ_', '')[:13].rjust(13) for n in names])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', kv[1].get(names[0], 0))):
    print(k.ljust(28), str(cnt[k]).rjust(4), *[('%.4g' % v[n]).rjust(13) for n in names])
