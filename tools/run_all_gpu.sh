export TMPDIR=/tmp; mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1; echo "all rc=$?"; tail -3 gpurun_out/t_all.log
