"""Aggregate a rocprofv3 --pmc counter_collection csv over ALL dispatches: per kernel and in total, the cycles each SIMD
would need for the VALU instructions (4 cycles per wave64 instruction) and the cycles its MFMA pipe was busy."""
import csv, glob, re, sys, collections
d = sys.argv[1]
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].replace('dsir::(anonymous namespace)::', '').replace('void ', '')
    n = re.sub(r'\(.*', '', n)
    acc[n][r['Counter_Name']] += float(r['Counter_Value'])
tot = collections.defaultdict(float)
rows = []
for n, c in acc.items():
    valu = c.get('SQ_INSTS_VALU', 0) * 4 / 1024 / 1e6          # M cycles per SIMD
    mfma = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / 1e6
    gui = c.get('GRBM_GUI_ACTIVE', 0) / 8 / 1e6
    rows.append((gui, n, valu, mfma))
    tot['valu'] += valu; tot['mfma'] += mfma; tot['gui'] += gui
rows.sort(reverse=True)
print(f"{'kernel':50s} {'Mcyc':>8s} {'VALU':>8s} {'MFMA':>8s}")
for gui, n, valu, mfma in rows[:25]:
    print(f"{n[:50]:50s} {gui:8.1f} {valu:8.1f} {mfma:8.1f}")
print(f"{'TOTAL':50s} {tot['gui']:8.1f} {tot['valu']:8.1f} {tot['mfma']:8.1f}")
