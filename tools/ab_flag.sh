#!/bin/bash
# tools/ab_flag.sh FLAG [rounds] [bench args]: bench.py with and without a DSIR_* switch (a flag is ON when it is set at all), alternating
flag=$1; n=${2:-2}; shift 2
mkdir -p gpurun_out
show() { python3 - "$1" "$2" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "pairs/s", j["value"], "model_only", j.get("model_only", {}).get("value"), "batch1 ms", j.get("batch1_latency", {}).get("ms_per_pair"))
PY
}
for i in $(seq 1 $n); do
  python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-companion "$@" > gpurun_out/abf_off_$i.json 2> gpurun_out/abf_off_$i.err || { tail -3 gpurun_out/abf_off_$i.err; exit 1; }
  show gpurun_out/abf_off_$i.json "default"
  env DSIR_TUNING=1 $flag=1 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-companion "$@" > gpurun_out/abf_on_$i.json 2> gpurun_out/abf_on_$i.err || { tail -3 gpurun_out/abf_on_$i.err; exit 1; }
  show gpurun_out/abf_on_$i.json "$flag=1"
done
