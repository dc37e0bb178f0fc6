"""Summarise a rocprofv3 --pmc counter_collection csv: per kernel, mean of each counter."""
import csv, glob, re, sys, collections
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ''
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].replace('dsir::(anonymous namespace)::', '').replace('void ', '')
    n = re.sub(r'\(.*', '', n)
    if pat and pat not in n: continue
    acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n, cs in acc.items():
    print(n, ' '.join('%s=%.4g(n=%d)' % (c, sum(v) / len(v), len(v)) for c, v in sorted(cs.items())))
