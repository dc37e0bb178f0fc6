#!/bin/bash
# tools/ab_env.sh VAR "v1 v2 ..." [bench args]: bench.py once per value of an environment switch; prints value / latency
var=$1; vals=$2; shift 2
mkdir -p gpurun_out
export DSIR_TUNING=1   # the library reads DSIR_* switches only behind this gate (include/dsir.h)
for v in $vals; do
  export $var=$v
  python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-companion "$@" > gpurun_out/ab_${var}_$v.json 2> gpurun_out/ab_${var}_$v.err || { tail -3 gpurun_out/ab_${var}_$v.err; exit 1; }
  python3 - gpurun_out/ab_${var}_$v.json "$var=$v" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(sys.argv[2], "pairs/s", j["value"], "model_only", j.get("model_only", {}).get("value"), "batch1 ms", j.get("batch1_latency", {}).get("ms_per_pair"),
      "kernel ms", j["roofline"].get("avg_launch_ms"), "op ms", j["roofline"].get("operation", {}).get("avg_ms"))
PY
done
