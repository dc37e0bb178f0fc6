export TMPDIR=/tmp; mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=4
for cfg in "3 2 0" "3 3 0" "3 4 0" "3 5 0" "2 5 0" "2 6 0" "1 9 0" "1 12 0"; do
timeout -k 10 120 python3 tools/thread_replay.py $cfg 2>&1 | grep "engines" | tee -a gpurun_out/thread_replay4.txt
done
