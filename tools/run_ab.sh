export TMPDIR=/tmp; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_config.py tests/test_serve.py -x -q > gpurun_out/t_agg.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/t_agg.log
bash tools/kstat_single.sh new 2>&1 | head -8
for i in 1 2; do
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-companion --timed-only 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new value', j['value'], j['ms_per_step'])"
(cd _old && python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-companion --timed-only 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r04 value', j['value'], j['ms_per_step'])")
done
python3 tools/b1_timeline.py run 5000 1 2>&1 | grep graph
(cd _old && cp ../tools/b1_timeline.py /tmp/b1t.py && sed -i 's/print("graph", eng.graph_stats(), /print("graph", /' /tmp/b1t.py && PYTHONPATH=. python3 /tmp/b1t.py run 5000 1 2>&1 | grep graph)
