#!/bin/bash
# round-3: pruned search - 256-row blocks (finer tile lists, less reuse of a staged ref tile) vs 512-row blocks
out=gpurun_out
for rt in 4 2; do
  export DSIR_SCREEN_RT=$rt
  python3 bench.py --points 16384 --feat-len 4 --shape kitti --pairs 32 --steps 8 --warmup 2 --no-cpu-baseline --no-companion --no-latency > $out/r3_rt_c3_$rt.json 2> $out/r3_rt_c3_$rt.err
  python3 bench.py --points 65536 --partial-overlap --pairs 4 --steps 4 --warmup 1 --no-cpu-baseline --no-companion --no-latency > $out/r3_rt_c5_$rt.json 2> $out/r3_rt_c5_$rt.err
  python3 - $out/r3_rt_c3_$rt.json $out/r3_rt_c5_$rt.json "RT=$rt" <<'PY'
import json, sys
a = json.load(open(sys.argv[1])); b = json.load(open(sys.argv[2]))
print(sys.argv[3], "C3 pairs/s", a["value"], "kernel ms", a["roofline"].get("avg_launch_ms"), "| C5 pairs/s", b["value"], "kernel ms", b["roofline"].get("avg_launch_ms"))
PY
done
