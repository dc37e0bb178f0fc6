export TMPDIR=/tmp; mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
for w in "" 1; do for cfg in "1 1 0" "2 1 0" "1 4 0" "2 4 0" "1 8 0" "2 8 0" "2 2 0" "3 2 0" "3 1 0"; do
WALK=$w timeout -k 10 120 python3 tools/thread_replay.py $cfg 2>&1 | grep "engines\|host time" | tee -a gpurun_out/thread_replay2.txt
done; done
