export TMPDIR=/tmp; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "pyramid or knn or kabsch or register_with_supplied" > gpurun_out/t_b8.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/t_b8.log
rm -rf /tmp/b8t && rocprofv3 --kernel-trace -d /tmp/b8t --output-format csv -- python3 tools/b1_timeline.py run 5000 8 > gpurun_out/b8_run.txt 2>&1; tail -1 gpurun_out/b8_run.txt
python3 tools/b1_timeline.py report /tmp/b8t gpurun_out/b8_timeline_new.txt | tail -1
grep -n " 0.0 us" -A 12 gpurun_out/b8_timeline_new.txt
export GPU_MAX_HW_QUEUES=8
for cfg in "1 1 0" "1 8 0" "2 4 0" "2 8 0"; do timeout -k 10 120 python3 tools/thread_replay.py $cfg 2>&1 | grep "engines"; done
unset GPU_MAX_HW_QUEUES
for i in 1 2; do
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-companion --timed-only 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new value', j['value'], j['ms_per_step'])"
(cd _old && python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-companion --timed-only 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old value', j['value'], j['ms_per_step'])")
done
