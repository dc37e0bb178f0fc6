#!/bin/bash
# round-3: pruned search - ref splits of the long ranges (item granularity vs entries per row)
out=gpurun_out
for sp in 2 4 8; do
  export DSIR_SCREEN_SPLITS=$sp
  python3 bench.py --points 16384 --feat-len 4 --shape kitti --pairs 32 --steps 8 --warmup 2 --no-cpu-baseline --no-companion --no-latency > $out/r3_sp_c3_$sp.json 2> $out/r3_sp_c3_$sp.err
  python3 bench.py --points 65536 --partial-overlap --pairs 4 --steps 4 --warmup 1 --no-cpu-baseline --no-companion --no-latency > $out/r3_sp_c5_$sp.json 2> $out/r3_sp_c5_$sp.err
  python3 - $out/r3_sp_c3_$sp.json $out/r3_sp_c5_$sp.json "splits=$sp" <<'PY'
import json, sys
a = json.load(open(sys.argv[1])); b = json.load(open(sys.argv[2]))
print(sys.argv[3], "C3 pairs/s", a["value"], "kernel ms", a["roofline"].get("avg_launch_ms"), "| C5 pairs/s", b["value"], "kernel ms", b["roofline"].get("avg_launch_ms"))
PY
done
