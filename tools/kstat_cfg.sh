#!/bin/bash
# tools/kstat_cfg.sh TAG <bench args...>: rocprofv3 kernel stats of bench.py --timed-only at another configuration; top rows
tag=$1; shift
export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag --output-format csv -- python3 bench.py --timed-only --steps 3 --warmup 1 "$@" > gpurun_out/${tag}_cfg.json 2> gpurun_out/${tag}_cfg.err || { tail -5 gpurun_out/${tag}_cfg.err; exit 1; }
f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in rows[:16]:
    n = r['Name'].replace('dsir::(anonymous namespace)::', '').replace('void ', '')
    print(f"{n[:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/tot*100:5.2f} %")
PY
