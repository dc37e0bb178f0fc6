#!/bin/bash
# round-3: pruned search - exactness (screened vs exhaustive bits, fp64 arg-min) and effect on C2 / C3 / C5
out=gpurun_out
python3 -m pytest tests/test_gpu_large_configs.py tests/test_gpu_bench_config.py -m gpu -x -q -s > $out/r3_prune_tests.log 2>&1; echo "tests rc=$?"; grep -E "\[prune\]|passed|failed|Error" $out/r3_prune_tests.log | tail -8
for v in 8192 0; do
  export DSIR_PRUNE_MIN_K=$v
  python3 bench.py --points 16384 --feat-len 4 --shape kitti --pairs 32 --steps 8 --warmup 2 --no-cpu-baseline --no-companion --no-latency > $out/r3_prune_c3_$v.json 2> $out/r3_prune_c3_$v.err
  python3 bench.py --points 65536 --partial-overlap --pairs 4 --steps 4 --warmup 1 --no-cpu-baseline --no-companion --no-latency > $out/r3_prune_c5_$v.json 2> $out/r3_prune_c5_$v.err
  python3 - $out/r3_prune_c3_$v.json $out/r3_prune_c5_$v.json "DSIR_PRUNE_MIN_K=$v" <<'PY'
import json, sys
a = json.load(open(sys.argv[1])); b = json.load(open(sys.argv[2]))
print(sys.argv[3], "C3 pairs/s", a["value"], "kernel ms", a["roofline"].get("avg_launch_ms"), "| C5 pairs/s", b["value"], "kernel ms", b["roofline"].get("avg_launch_ms"))
PY
done
unset DSIR_PRUNE_MIN_K
python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-companion --no-latency > $out/r3_prune_c2.json 2> $out/r3_prune_c2.err
python3 - $out/r3_prune_c2.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print("C2 pairs/s", j["value"], "kernel ms", j["roofline"].get("avg_launch_ms"))
PY
