#!/bin/bash
# round-3: is a persistent grid itself slower for the dense screening? (DSIR_SCREEN_PERSIST = workgroups per CU)
out=gpurun_out
export DSIR_PRUNE_MIN_K=0
for ps in 0 1; do
  export DSIR_SCREEN_PERSIST=$ps
  python3 bench.py --points 16384 --feat-len 4 --shape kitti --pairs 32 --steps 8 --warmup 2 --no-cpu-baseline --no-companion --no-latency > $out/r3_ps_c3_$ps.json 2> $out/r3_ps_c3_$ps.err
  python3 bench.py --points 65536 --partial-overlap --pairs 4 --steps 4 --warmup 1 --no-cpu-baseline --no-companion --no-latency > $out/r3_ps_c5_$ps.json 2> $out/r3_ps_c5_$ps.err
  python3 - $out/r3_ps_c3_$ps.json $out/r3_ps_c5_$ps.json "persist=$ps" <<'PY'
import json, sys
a = json.load(open(sys.argv[1])); b = json.load(open(sys.argv[2]))
print(sys.argv[3], "C3 pairs/s", a["value"], "kernel ms", a["roofline"].get("avg_launch_ms"), "| C5 pairs/s", b["value"], "kernel ms", b["roofline"].get("avg_launch_ms"))
PY
done
