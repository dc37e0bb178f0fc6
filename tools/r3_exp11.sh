#!/bin/bash
# round-3: is caching the enc half of the attention scores still worth its HBM reads now that the contraction is cheap?
out=gpurun_out
for v in 0 256 999; do
  export DSIR_S2_MIN_D=$v
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-companion > $out/r3_s2_$v.json 2> $out/r3_s2_$v.err
  python3 - $out/r3_s2_$v.json "DSIR_S2_MIN_D=$v" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(sys.argv[2], "pairs/s", j["value"], "ms/step", j["ms_per_step"], "batch1 ms", j.get("batch1_latency", {}).get("ms_per_pair"))
PY
done
