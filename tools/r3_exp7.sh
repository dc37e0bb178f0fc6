#!/bin/bash
# round-3: split-K deep GEMM - parity tests, batch-1 latency and throughput A/B
out=gpurun_out
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_config.py -m gpu -x -q > $out/r3_splitk_tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/r3_splitk_tests.log
for v in 1 2 0 1 2 0; do
  export DSIR_TILE_SPLITK=$v
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-companion > $out/r3_splitk_$v.json 2> $out/r3_splitk_$v.err
  python3 - $out/r3_splitk_$v.json "DSIR_TILE_SPLITK=$v" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(sys.argv[2], "pairs/s", j["value"], "ms/step", j["ms_per_step"], "batch1 ms", j.get("batch1_latency", {}).get("ms_per_pair"))
PY
done
