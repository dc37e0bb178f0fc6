#!/bin/bash
# round-3: interpolation search through the next level's grid (grid_nn1_kernel): parity + A/B (DSIR_NO_NN1_GRID)
out=gpurun_out
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "knn or interp or pyramid" > $out/r3_e23_tests.log 2>&1; echo "tests rc=$?"; tail -2 $out/r3_e23_tests.log
for v in grid brute; do
  if [ $v = brute ]; then export DSIR_NO_NN1_GRID=1; else unset DSIR_NO_NN1_GRID; fi
  python3 bench.py --points 16384 --feat-len 4 --shape kitti --pairs 32 --steps 8 --warmup 2 --no-cpu-baseline --no-companion --no-latency > $out/r3_e23_c3_$v.json 2> $out/r3_e23_c3_$v.err
  python3 bench.py --points 65536 --partial-overlap --pairs 4 --steps 4 --warmup 1 --no-cpu-baseline --no-companion --no-latency > $out/r3_e23_c5_$v.json 2> $out/r3_e23_c5_$v.err
  python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-companion --no-latency > $out/r3_e23_c2_$v.json 2> $out/r3_e23_c2_$v.err
  python3 - $v <<'PY'
import json, sys
v = sys.argv[1]
print(v, " ".join(f"{c} {json.load(open(f'gpurun_out/r3_e23_{c}_{v}.json'))['value']}" for c in ("c2", "c3", "c5")))
PY
done
