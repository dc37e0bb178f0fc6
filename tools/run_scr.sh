export TMPDIR=/tmp; mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
for b in 2 4 6 8 12 16; do
echo "screened (default, >= 1e8): $(timeout -k 10 120 python3 tools/thread_replay.py 1 $b 0 2>&1 | grep engines)"
echo "exhaustive:                 $(DSIR_TUNING=1 DSIR_NO_SCREEN=1 timeout -k 10 120 python3 tools/thread_replay.py 1 $b 0 2>&1 | grep engines)"
done
