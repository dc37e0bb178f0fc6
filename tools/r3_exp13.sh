#!/bin/bash
# round-3: decoder layer 160 -> 32 through the split pw_tile kernel instead of pw_gemm
out=gpurun_out
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stages or teacher or free_running or ragged" > $out/r3_cout_tests.log 2>&1; echo "tests rc=$?"; tail -2 $out/r3_cout_tests.log
for v in 32 64 32 64; do
  export DSIR_TILE_MIN_COUT=$v
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-companion > $out/r3_cout_$v.json 2> $out/r3_cout_$v.err
  python3 - $out/r3_cout_$v.json "DSIR_TILE_MIN_COUT=$v" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(sys.argv[2], "pairs/s", j["value"], "ms/step", j["ms_per_step"], "batch1 ms", j.get("batch1_latency", {}).get("ms_per_pair"))
PY
done
