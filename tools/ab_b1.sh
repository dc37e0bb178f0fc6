#!/bin/bash
# tools/ab_b1.sh VAR "v1 v2 ...": batch-1 latency (bench.py's hipGraph replay of one pair) per value of an environment switch
var=$1; vals=$2
export DSIR_TUNING=1
for v in $vals; do
  export $var=$v
  python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-companion --pairs 8 --streams 1 2>/dev/null > gpurun_out/b1_$v.json
  python3 - gpurun_out/b1_$v.json "$var=$v" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "batch1 ms", j["batch1_latency"]["ms_per_pair"])
PY
done
