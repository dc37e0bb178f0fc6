export TMPDIR=/tmp; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_walk.py -x -q > gpurun_out/t_walk.log 2>&1; echo "walk/fork tests rc=$?"; tail -4 gpurun_out/t_walk.log
export DSIR_TUNING=1
for m in 0 1 2 4 8 9 11 15; do
echo "mask $m: $(DSIR_FORK_MASK=$m python3 tools/b1_timeline.py run 5000 1 2>&1 | grep -o 'ms_per_registration [0-9.]*')  8 pairs: $(DSIR_FORK_MASK=$m python3 tools/b1_timeline.py run 5000 8 2>&1 | grep -o 'ms_per_registration [0-9.]*')"
done
