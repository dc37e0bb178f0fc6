#!/bin/bash
# round-3: rows per cloud up to which the deep-level GEMMs take the 32/64-row tiles (DSIR_TILE_SMALL_M): large clouds have 1024 - 4096-row levels
out=gpurun_out
for sm in 320 1100 4200; do
  export DSIR_TILE_SMALL_M=$sm
  python3 bench.py --steps 8 --warmup 2 --timed-only > $out/r3_e31_c2_$sm.json 2> $out/r3_e31_c2_$sm.err
  python3 bench.py --points 16384 --feat-len 4 --shape kitti --pairs 32 --steps 8 --warmup 2 --timed-only > $out/r3_e31_c3_$sm.json 2> $out/r3_e31_c3_$sm.err
  python3 bench.py --points 65536 --partial-overlap --pairs 4 --steps 4 --warmup 1 --timed-only > $out/r3_e31_c5_$sm.json 2> $out/r3_e31_c5_$sm.err
  python3 bench.py --points 65536 --partial-overlap --pairs 1 --streams 1 --steps 8 --warmup 2 --timed-only > $out/r3_e31_b5_$sm.json 2> $out/r3_e31_b5_$sm.err
  python3 - $sm <<'PY'
import json, sys
sm = sys.argv[1]
v = {c: json.load(open(f"gpurun_out/r3_e31_{c}_{sm}.json")) for c in ("c2", "c3", "c5", "b5")}
print("small_m", sm, "C2", v["c2"]["value"], "C3", v["c3"]["value"], "C5", v["c5"]["value"], "| 1 x 65536:", v["b5"]["ms_per_step"], "ms")
PY
done
