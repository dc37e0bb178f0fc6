#!/bin/bash
# round-3: rows per launch from which the pruned search runs (DSIR_PRUNE_MIN_ROWS): single-pair latency at 16384 / 65536 points, and 4 x 16384
out=gpurun_out
for mr in 65536 16384 1; do
  export DSIR_PRUNE_MIN_ROWS=$mr
  python3 bench.py --points 16384 --feat-len 4 --shape kitti --pairs 1 --streams 1 --steps 20 --warmup 3 --timed-only > $out/r3_e30_a_$mr.json 2> $out/r3_e30_a_$mr.err
  python3 bench.py --points 16384 --feat-len 4 --shape kitti --pairs 2 --streams 1 --steps 20 --warmup 3 --timed-only > $out/r3_e30_b_$mr.json 2> $out/r3_e30_b_$mr.err
  python3 bench.py --points 32768 --pairs 1 --streams 1 --steps 10 --warmup 2 --timed-only > $out/r3_e30_c_$mr.json 2> $out/r3_e30_c_$mr.err
  python3 - $mr <<'PY'
import json, sys
mr = sys.argv[1]
v = [json.load(open(f"gpurun_out/r3_e30_{k}_{mr}.json"))["ms_per_step"] for k in "abc"]
print("min rows", mr, "1 x 16384:", v[0], "ms | 2 x 16384:", v[1], "ms | 1 x 32768:", v[2], "ms")
PY
done
