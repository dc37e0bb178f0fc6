#!/bin/bash
# round-3: LDS-tiled GEMM family (pw_tile / pw_tile_small) on the fp16 pipe - parity + A/B vs DSIR_TILE_F32
out=gpurun_out
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_config.py tests/test_pipelines.py tests/test_gpu_large_configs.py -m gpu -x -q > $out/r3_tileh_tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/r3_tileh_tests.log
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export DSIR_TILE_F32=1; else unset DSIR_TILE_F32; fi
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-companion > $out/r3_tileh_f32_$v.json 2> $out/r3_tileh_f32_$v.err
  python3 - $out/r3_tileh_f32_$v.json "DSIR_TILE_F32=$v" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(sys.argv[2], "pairs/s", j["value"], "ms/step", j["ms_per_step"], "batch1 ms", j.get("batch1_latency", {}).get("ms_per_pair"))
PY
done
unset DSIR_TILE_F32
export TMPDIR=/tmp
rm -rf /tmp/prof_a; rocprofv3 --kernel-trace --stats -d /tmp/prof_a --output-format csv -- python3 bench.py --pairs 128 --streams 1 --steps 5 --warmup 1 --timed-only > $out/r3_tileh_trace.json 2> $out/r3_tileh_trace.err
cp "$(find /tmp/prof_a -name '*kernel_stats.csv' | head -1)" $out/r3_tileh_kernel_stats_single.csv
grep -E "pw_tile" $out/r3_tileh_kernel_stats_single.csv | cut -d, -f1-5 | cut -c40-200
