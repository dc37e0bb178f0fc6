export TMPDIR=/tmp; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_train.py tests/test_train_loop.py tests/test_align_loss.py -x -q -m gpu > gpurun_out/t_train.log 2>&1; echo "train tests rc=$?"; tail -8 gpurun_out/t_train.log
python3 tools/bench_train.py > gpurun_out/bench_train.log 2>&1; tail -12 gpurun_out/bench_train.log
