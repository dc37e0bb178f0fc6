#!/bin/bash
# round-3: fp16-split aggregation chain - parity tests + same-box A/B of the bench
out=gpurun_out
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "agg_chain_split or stages_vs_oracle or free_running or teacher_forced" > $out/r3_aggh_tests.log 2>&1; echo "tests rc=$?"; grep -E "agg-split|passed|failed" $out/r3_aggh_tests.log | tail -8
for v in 0 1; do
  if [ $v = 1 ]; then export DSIR_AGG_F32=1; else unset DSIR_AGG_F32; fi
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-companion --no-latency > $out/r3_aggh_f32_$v.json 2> $out/r3_aggh_f32_$v.err
  python3 - $out/r3_aggh_f32_$v.json "DSIR_AGG_F32=$v" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(sys.argv[2], "pairs/s", j["value"], "ms/step", j["ms_per_step"])
PY
done
unset DSIR_AGG_F32
export TMPDIR=/tmp
rm -rf /tmp/prof_a; rocprofv3 --kernel-trace --stats -d /tmp/prof_a --output-format csv -- python3 bench.py --pairs 128 --streams 1 --steps 5 --warmup 1 --timed-only > $out/r3_aggh_trace.json 2> $out/r3_aggh_trace.err
cp "$(find /tmp/prof_a -name '*kernel_stats.csv' | head -1)" $out/r3_aggh_kernel_stats_single.csv
head -8 $out/r3_aggh_kernel_stats_single.csv | cut -c1-160
