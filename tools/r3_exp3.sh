#!/bin/bash
# round-3: XCD-aware gathers (pooling, score, grid KNN) - bit-identity + kernel averages + bench
out=gpurun_out
export TMPDIR=/tmp
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "knn or stages or determinism or teacher_forced or pool or engine_pool" > $out/r3_xcd_tests.log 2>&1; echo "tests rc=$?"; tail -2 $out/r3_xcd_tests.log
rm -rf /tmp/prof_a; rocprofv3 --kernel-trace --stats -d /tmp/prof_a --output-format csv -- python3 bench.py --pairs 128 --streams 1 --steps 5 --warmup 1 --timed-only > $out/r3_xcd_trace.json 2> $out/r3_xcd_trace.err
cp "$(find /tmp/prof_a -name '*kernel_stats.csv' | head -1)" $out/r3_xcd_kernel_stats_single.csv
grep -E "gather_max|score_point|grid_knn_kernel|agg_chain" $out/r3_xcd_kernel_stats_single.csv | cut -d, -f1-4 | cut -c1-150
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-companion --no-latency > $out/r3_xcd_bench.json 2> $out/r3_xcd_bench.err
python3 - $out/r3_xcd_bench.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print("pairs/s", j["value"], "ms/step", j["ms_per_step"])
PY
