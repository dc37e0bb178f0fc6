#!/bin/bash
# round-3: the reworked pruned search on the 5000-point configuration (DSIR_PRUNE_MIN_K=4096) and with 256-row blocks on the large ones
out=gpurun_out
DSIR_PRUNE_MIN_K=4096 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-companion --no-latency > $out/r3_e29_c2_prune.json 2> $out/r3_e29_c2_prune.err
python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-companion --no-latency > $out/r3_e29_c2.json 2> $out/r3_e29_c2.err
python3 - <<'PY'
import json
a = json.load(open("gpurun_out/r3_e29_c2_prune.json")); b = json.load(open("gpurun_out/r3_e29_c2.json"))
print("C2 pruned", a["value"], "kernel ms", a["roofline"]["avg_launch_ms"], (a["roofline"].get("pruned") or {}).get("executed_share_of_dense_flops"), "| unpruned", b["value"], b["roofline"]["avg_launch_ms"])
PY
for rt in 4 2; do
  export DSIR_SCREEN_RT=$rt
  python3 bench.py --points 16384 --feat-len 4 --shape kitti --pairs 32 --steps 8 --warmup 2 --no-cpu-baseline --no-companion --no-latency > $out/r3_e29_c3_$rt.json 2> $out/r3_e29_c3_$rt.err
  python3 bench.py --points 65536 --partial-overlap --pairs 4 --steps 4 --warmup 1 --no-cpu-baseline --no-companion --no-latency > $out/r3_e29_c5_$rt.json 2> $out/r3_e29_c5_$rt.err
  python3 - $rt <<'PY'
import json, sys
rt = sys.argv[1]
a = json.load(open(f"gpurun_out/r3_e29_c3_{rt}.json")); b = json.load(open(f"gpurun_out/r3_e29_c5_{rt}.json"))
print("RT", rt, "C3", a["value"], a["roofline"]["pruned"]["executed_share_of_dense_flops"], "| C5", b["value"], b["roofline"]["pruned"]["executed_share_of_dense_flops"])
PY
done
