"""rocprofv3 --pmc counter_collection.csv  ->  one row per kernel (template arguments kept): dispatches and the SUM of
every counter over them.  The committed evidence behind profiles/README.md (the raw per-dispatch CSVs are tens of MB).
    python tools/pmc_table.py <rocprof output dir> <out.csv> [steps]"""
import collections, csv, glob, re, sys

d, out = sys.argv[1], sys.argv[2]
files = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
names = set()
for f in files:
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('dsir::(anonymous namespace)::', '').replace('void ', '')
        n = re.sub(r'\(.*', '', n)
        acc[n][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[n][r['Counter_Name']] += 1
        names.add(r['Counter_Name'])
names = sorted(names)
key = 'GRBM_GUI_ACTIVE' if 'GRBM_GUI_ACTIVE' in names else names[0]
rows = sorted(acc.items(), key=lambda kv: -kv[1].get(key, 0.0))
with open(out, 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'dispatches'] + names)
    for n, c in rows:
        w.writerow([n, max(cnt[n].values())] + ['%.6g' % c.get(k, 0.0) for k in names])
    w.writerow(['TOTAL', sum(max(v.values()) for v in cnt.values())] + ['%.6g' % sum(c.get(k, 0.0) for c in acc.values()) for k in names])
print(open(out).read()[:3000])
