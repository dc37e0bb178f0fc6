#!/bin/bash
# round-3: fp16-split head - parity + same-box A/B
out=gpurun_out
python3 -m pytest tests/test_gpu_parity.py tests/test_pipelines.py tests/test_gpu_bench_config.py -m gpu -x -q -s > $out/r3_headh_tests.log 2>&1; echo "tests rc=$?"; grep -E "head-split|passed|failed" $out/r3_headh_tests.log | tail -6
export TMPDIR=/tmp
rm -rf /tmp/prof_a; rocprofv3 --kernel-trace --stats -d /tmp/prof_a --output-format csv -- python3 bench.py --pairs 128 --streams 1 --steps 5 --warmup 1 --timed-only > $out/r3_headh_trace.json 2> $out/r3_headh_trace.err
cp "$(find /tmp/prof_a -name '*kernel_stats.csv' | head -1)" $out/r3_headh_kernel_stats_single.csv
grep -E "head_mlp" $out/r3_headh_kernel_stats_single.csv | cut -d, -f1-5 | cut -c40-200
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-companion --no-latency > $out/r3_headh_bench.json 2> $out/r3_headh_bench.err
python3 - $out/r3_headh_bench.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print("pairs/s", j["value"], "ms/step", j["ms_per_step"])
PY
