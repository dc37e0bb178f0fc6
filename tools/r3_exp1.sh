#!/bin/bash
# round-3 experiment: batch geometry of the bench + timeline statistics
export TMPDIR=/tmp
out=gpurun_out
for cfg in "256 2" "512 2" "384 3" "512 4" "384 2"; do
  set -- $cfg
  python3 bench.py --pairs $1 --streams $2 --steps 8 --warmup 2 --no-cpu-baseline --no-companion --no-latency > $out/r3_geo_$1_$2.json 2> $out/r3_geo_$1_$2.err
  python3 - $out/r3_geo_$1_$2.json "$cfg" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(sys.argv[2], "pairs/s", j["value"], "ms/step", j["ms_per_step"], "kernel ms", j["roofline"].get("avg_launch_ms"))
PY
done
rm -rf /tmp/prof_t; rocprofv3 --kernel-trace -d /tmp/prof_t --output-format csv -- python3 bench.py --timed-only --steps 3 --warmup 1 > $out/r3_trace.json 2> $out/r3_trace.err
python3 tools/trace_gaps.py /tmp/prof_t $out/r3_trace_gaps.json > /dev/null
head -c 1500 $out/r3_trace_gaps.json
