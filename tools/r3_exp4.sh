#!/bin/bash
# round-3: cloud-batched deep GEMM (pw_tile_cl4_kernel) - bit identity + same-box A/B
out=gpurun_out
python3 -m pytest tests/test_gpu_bench_config.py -m gpu -x -q -k "cloud_batched" > $out/r3_cl4_tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/r3_cl4_tests.log
for v in 16 0 16 0; do
  export DSIR_TILE_CL4_MIN=$v
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-companion --no-latency > $out/r3_cl4_$v.json 2> $out/r3_cl4_$v.err
  python3 - $out/r3_cl4_$v.json "DSIR_TILE_CL4_MIN=$v" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(sys.argv[2], "pairs/s", j["value"], "ms/step", j["ms_per_step"])
PY
done
unset DSIR_TILE_CL4_MIN
export TMPDIR=/tmp
rm -rf /tmp/prof_a; rocprofv3 --kernel-trace --stats -d /tmp/prof_a --output-format csv -- python3 bench.py --pairs 128 --streams 1 --steps 5 --warmup 1 --timed-only > $out/r3_cl4_trace.json 2> $out/r3_cl4_trace.err
cp "$(find /tmp/prof_a -name '*kernel_stats.csv' | head -1)" $out/r3_cl4_kernel_stats_single.csv
grep -E "pw_tile_cl4|pw_tile_small" $out/r3_cl4_kernel_stats_single.csv | cut -d, -f1-5 | cut -c40-200
