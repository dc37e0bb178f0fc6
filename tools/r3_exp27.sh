#!/bin/bash
# round-3: workgroups per cloud of the GroupNorm row-stream layers (DSIR_STREAM_MIN_BLOCKS): batch-1 latency vs throughput
out=gpurun_out
for fb in 16 32 64 128; do
  export DSIR_STREAM_MIN_BLOCKS=$fb
  python3 bench.py --pairs 1 --streams 1 --steps 40 --warmup 5 --timed-only > $out/r3_e27_b1_$fb.json 2> $out/r3_e27_b1_$fb.err
  python3 bench.py --steps 8 --warmup 2 --timed-only > $out/r3_e27_c2_$fb.json 2> $out/r3_e27_c2_$fb.err
  python3 - $fb <<'PY'
import json, sys
fb = sys.argv[1]
a = json.load(open(f"gpurun_out/r3_e27_b1_{fb}.json")); b = json.load(open(f"gpurun_out/r3_e27_c2_{fb}.json"))
print("min blocks", fb, "batch-1 ms/pair", a["ms_per_step"], "| C2 pairs/s", b["value"])
PY
done
