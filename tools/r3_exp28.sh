#!/bin/bash
# round-3: mlp_skip of every RandLA block on a side stream (fork / join, captured into the hipGraph): batch-1 latency, bits
out=gpurun_out
for f in 0 2 8; do
  export DSIR_FORK_MAX_CLOUDS=$f
  python3 bench.py --pairs 1 --streams 1 --steps 40 --warmup 5 --timed-only > $out/r3_e28_b1_$f.json 2> $out/r3_e28_b1_$f.err || { tail -3 $out/r3_e28_b1_$f.err; exit 1; }
  python3 bench.py --pairs 4 --streams 1 --steps 20 --warmup 3 --timed-only > $out/r3_e28_b4_$f.json 2> $out/r3_e28_b4_$f.err || { tail -3 $out/r3_e28_b4_$f.err; exit 1; }
  python3 - $f <<'PY'
import json, sys
f = sys.argv[1]
a = json.load(open(f"gpurun_out/r3_e28_b1_{f}.json")); b = json.load(open(f"gpurun_out/r3_e28_b4_{f}.json"))
print("fork max clouds", f, "batch-1 ms/pair", a["ms_per_step"], "| 4 pairs ms/step", b["ms_per_step"])
PY
done
export DSIR_FORK_MAX_CLOUDS=2
python3 -m pytest tests/test_gpu_bench_config.py tests/test_gpu_parity.py -m gpu -x -q > $out/r3_e28_tests.log 2>&1; echo "tests with fork rc=$?"; tail -2 $out/r3_e28_tests.log
