#!/bin/bash
# round-3: training step with the position-encoding branch shared across the registration iterations
out=gpurun_out
export TMPDIR=/tmp
python3 -m pytest tests/test_train.py tests/test_align_loss.py -m gpu -x -q > $out/r3_train_tests.log 2>&1; echo "tests rc=$?"; tail -4 $out/r3_train_tests.log
python3 tools/bench_train.py --pairs 8 --steps 5 2>/dev/null | tail -1
python3 tools/bench_train.py --pairs 32 --steps 4 2>/dev/null | tail -1
python3 tools/bench_train.py --pairs 8 --steps 4 --full 2>/dev/null | tail -1
rm -rf /tmp/prof_c; rocprofv3 --kernel-trace --stats -d /tmp/prof_c --output-format csv -- python3 tools/bench_train.py --pairs 8 --steps 3 --eager > $out/r3_train_trace.log 2>&1
cp "$(find /tmp/prof_c -name '*kernel_stats.csv' | head -1)" $out/r3_train_kernel_stats_p8.csv
head -25 $out/r3_train_kernel_stats_p8.csv | cut -d, -f1-5 | sed 's/dsir::(anonymous namespace):://' | cut -c1-110
