#!/bin/bash
# Evidence run on the GPU box: kernel trace + HBM-side PMC passes (separate runs) of the align training step, 8 pairs, eager launches
# (every operator its own dispatch).   tools/profile_train.sh <tag>   -> gpurun_out/<tag>_train_*
set -o pipefail
tag=${1:-r03}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cmd="tools/bench_train.py --pairs 8 --steps 3 --eager"
rm -rf /tmp/pt_trace; rocprofv3 --kernel-trace --stats -d /tmp/pt_trace --output-format csv -- python3 $cmd > $out/${tag}_train_trace.log 2>&1 || { tail -5 $out/${tag}_train_trace.log; exit 1; }
cp "$(find /tmp/pt_trace -name '*kernel_stats.csv' | head -1)" $out/${tag}_train_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pt_$c
  rocprofv3 --pmc $c GRBM_GUI_ACTIVE -d /tmp/pt_$c --output-format csv -- python3 $cmd > $out/${tag}_train_$c.log 2>&1 || { tail -5 $out/${tag}_train_$c.log; exit 1; }
  python3 tools/pmc_table.py /tmp/pt_$c $out/${tag}_train_pmc_$c.csv > /dev/null
done
python3 - $out/${tag}_train_kernel_stats.csv $out/${tag}_train_pmc_FETCH_SIZE.csv $out/${tag}_train_pmc_WRITE_SIZE.csv $out/${tag}_train_hbm.txt <<'PY'
import csv, re, sys
stats = {}
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r'\(.*', '', r['Name'].replace('dsir::(anonymous namespace)::', '').replace('void ', ''))
    stats[n] = (int(r['Calls']), float(r['TotalDurationNs']))
f = {r['kernel']: float(r['FETCH_SIZE']) for r in csv.DictReader(open(sys.argv[2])) if r['kernel'] != 'TOTAL'}
w = {r['kernel']: float(r['WRITE_SIZE']) for r in csv.DictReader(open(sys.argv[3])) if r['kernel'] != 'TOTAL'}
rows = []
for n, (calls, ns) in stats.items():
    if n in f:
        b = f[n] * 2 * 1024 + w.get(n, 0.0) * 1024          # FETCH_SIZE doubled (gfx950 note of MI355X_MICROARCH.md), KiB -> bytes
        rows.append((ns, n, calls, b))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
with open(sys.argv[4], 'w') as o:
    o.write("kernel, calls, avg us, share of kernel time, HBM-side MB per call, GB/s over its own duration (peak 8000)\n")
    for ns, n, calls, b in rows[:24]:
        o.write(f"{n[:60]:60s} {calls:6d} {ns / calls / 1e3:8.1f} {100 * ns / tot:6.2f} % {b / calls / 1e6:9.1f} {b / ns:8.0f}\n")
print(open(sys.argv[4]).read())
PY
