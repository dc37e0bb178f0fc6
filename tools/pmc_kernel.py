"""Per-kernel averages of every counter in a rocprofv3 --pmc counter_collection csv (kernels filtered by substring)."""
import csv, glob, sys, collections
d, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name'] and int(r['Grid_Size']) > 100000:
            acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for n, c in acc.items():
    print(n)
    for k, v in sorted(c.items()):
        print(f"   {k:32s} n={len(v):4d} avg={sum(v) / len(v):16.1f}")
