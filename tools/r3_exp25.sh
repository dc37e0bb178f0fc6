#!/bin/bash
# round-3: training step after the launch-geometry changes: gradient parity tests + step time (8 / 32 pairs)
out=gpurun_out
python3 -m pytest tests/test_train.py -m gpu -x -q > $out/r3_e25_tests.log 2>&1; echo "tests rc=$?"; tail -2 $out/r3_e25_tests.log
python3 tools/bench_train.py --pairs 8 --steps 10 2>/dev/null | tail -1
python3 tools/bench_train.py --pairs 32 --steps 5 2>/dev/null | tail -1
