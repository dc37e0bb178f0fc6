#!/bin/bash
# round-3: batch-1 latency - where it goes now, and the screened arg-min at one pair
out=gpurun_out
export TMPDIR=/tmp
for v in 200000000 20000000; do
  export DSIR_SCREEN_MIN_WORK=$v
  python3 bench.py --pairs 1 --streams 1 --steps 20 --warmup 3 --no-cpu-baseline --no-companion > $out/r3_b1_$v.json 2> $out/r3_b1_$v.err
  python3 - $out/r3_b1_$v.json "DSIR_SCREEN_MIN_WORK=$v" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(sys.argv[2], "batch1 ms", j.get("batch1_latency", {}).get("ms_per_pair"), "eager pairs/s", j["value"])
PY
done
unset DSIR_SCREEN_MIN_WORK
rm -rf /tmp/prof_b; rocprofv3 --kernel-trace --stats -d /tmp/prof_b --output-format csv -- python3 bench.py --pairs 1 --streams 1 --steps 20 --warmup 2 --timed-only > $out/r3_b1_trace.json 2> $out/r3_b1_trace.err
cp "$(find /tmp/prof_b -name '*kernel_stats.csv' | head -1)" $out/r3_b1_kernel_stats.csv
head -30 $out/r3_b1_kernel_stats.csv | cut -d, -f1-4 | sed 's/dsir::(anonymous namespace):://' | cut -c1-120
