#!/bin/bash
# round-3: batch geometry again, after the fp16-split layers
out=gpurun_out
for cfg in "256 2" "384 3" "256 4" "512 4" "192 2" "384 2"; do
  set -- $cfg
  python3 bench.py --pairs $1 --streams $2 --steps 8 --warmup 2 --no-cpu-baseline --no-companion --no-latency > $out/r3_geo2_$1_$2.json 2> $out/r3_geo2_$1_$2.err
  python3 - $out/r3_geo2_$1_$2.json "$cfg" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(sys.argv[2], "pairs/s", j["value"], "ms/step", j["ms_per_step"])
PY
done
