#!/bin/bash
# tools/ab_args.sh "args1" "args2" ...: bench.py once per argument set; prints value / latency
mkdir -p gpurun_out
i=0
for a in "$@"; do
  i=$((i+1))
  python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-companion --no-latency $a > gpurun_out/aba_$i.json 2> gpurun_out/aba_$i.err || { tail -3 gpurun_out/aba_$i.err; exit 1; }
  python3 - gpurun_out/aba_$i.json "$a" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(sys.argv[2], "-> pairs/s", j["value"], "model_only", j.get("model_only", {}).get("value"), "ms/step", j["ms_per_step"])
PY
done
