#!/bin/bash
# tools/kstat_single.sh TAG [extra env assignments...]: rocprofv3 kernel stats of one engine registering 128 pairs (the launches of the
# roofline pass): gpurun_out/TAG_kernel_stats_single_engine.csv + the top rows
tag=$1; shift
export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
env "$@" rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag --output-format csv -- python3 bench.py --pairs 128 --streams 1 --steps 5 --warmup 1 --timed-only > gpurun_out/${tag}_single.json 2> gpurun_out/${tag}_single.err || { tail -5 gpurun_out/${tag}_single.err; exit 1; }
f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats_single_engine.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in rows[:28]:
    n = r['Name'].replace('dsir::(anonymous namespace)::', '').replace('void ', '')
    print(f"{n[:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/tot*100:5.2f} %")
PY
