export TMPDIR=/tmp; mkdir -p gpurun_out
python3 tools/k8_sweep.py 8 2>&1 | grep -v Warning | tee gpurun_out/k8_sweep.txt
GPU_MAX_HW_QUEUES=8 python3 tools/k8_sweep.py 8 2>&1 | grep -v Warning | tee -a gpurun_out/k8_sweep.txt
GPU_MAX_HW_QUEUES=8 python3 tools/k8_sweep.py 16 2>&1 | grep -v Warning | tee -a gpurun_out/k8_sweep.txt
