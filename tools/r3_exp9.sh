#!/bin/bash
# round-3: attentive-pooling score contraction on the fp16 pipe (pw_stream KQ 8 / 16) - parity + kernel averages + bench
out=gpurun_out
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_config.py tests/test_pipelines.py -m gpu -x -q > $out/r3_atth_tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/r3_atth_tests.log
export TMPDIR=/tmp
rm -rf /tmp/prof_a; rocprofv3 --kernel-trace --stats -d /tmp/prof_a --output-format csv -- python3 bench.py --pairs 128 --streams 1 --steps 5 --warmup 1 --timed-only > $out/r3_atth_trace.json 2> $out/r3_atth_trace.err
cp "$(find /tmp/prof_a -name '*kernel_stats.csv' | head -1)" $out/r3_atth_kernel_stats_single.csv
python3 - <<'PY'
import csv
a={r['Name']:r for r in csv.DictReader(open('gpurun_out/r3_headh_kernel_stats_single.csv'))}
b={r['Name']:r for r in csv.DictReader(open('gpurun_out/r3_atth_kernel_stats_single.csv'))}
ta=sum(float(r['TotalDurationNs']) for r in a.values()); tb=sum(float(r['TotalDurationNs']) for r in b.values())
print("total kernel ms", round(ta/1e6,1), "->", round(tb/1e6,1))
for k in a:
    if k in b and ("5, 0" in k or "4, 0, 0" in k) and "pw_stream" in k:
        print(f"{float(a[k]['AverageNs'])/1e3:8.1f} -> {float(b[k]['AverageNs'])/1e3:8.1f} us  {k[40:90]}")
PY
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-companion --no-latency > $out/r3_atth_bench.json 2> $out/r3_atth_bench.err
python3 - $out/r3_atth_bench.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print("pairs/s", j["value"], "ms/step", j["ms_per_step"])
PY
