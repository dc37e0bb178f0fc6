export TMPDIR=/tmp; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_walk.py -x -q > gpurun_out/t_walk.log 2>&1; echo "walk tests rc=$?"; tail -2 gpurun_out/t_walk.log
export DSIR_TUNING=1 DSIR_WALK_TRACE=1
for cfg in "0 32" "1 64"; do set -- $cfg; echo "== flags $1 wpc $2"; DSIR_WALK_FLAGS=$1 DSIR_WALK_WPC=$2 python3 tools/walk_trace.py 5000 1 gpurun_out/walk_trace_f$1_w$2.txt | sed -n 2,23p; done
unset DSIR_WALK_TRACE
for cfg in "0 32" "1 64" "1 128"; do set -- $cfg; echo "== b1 flags $1 wpc $2"; DSIR_WALK_FLAGS=$1 DSIR_WALK_WPC=$2 python3 tools/b1_timeline.py run 5000 1 2>&1 | grep graph; done
DSIR_NO_WALK=1 python3 tools/b1_timeline.py run 5000 1 2>&1 | grep graph
