export TMPDIR=/tmp; mkdir -p gpurun_out
run() { python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-companion --timed-only "$@" 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"; }
for i in 1 2; do
echo "default HWQ streams2: $(run)"
echo "HWQ=8 streams2: $(GPU_MAX_HW_QUEUES=8 run)"
echo "HWQ=16 streams2: $(GPU_MAX_HW_QUEUES=16 run)"
done
echo "HWQ=8 streams1 pairs256: $(GPU_MAX_HW_QUEUES=8 run --streams 1)"
echo "default streams1 pairs256: $(run --streams 1)"
echo "HWQ=8 streams3 : $(GPU_MAX_HW_QUEUES=8 run --streams 3 --pairs 255)"
echo "HWQ=8 streams4 : $(GPU_MAX_HW_QUEUES=8 run --streams 4)"
echo "HWQ=2 streams2: $(GPU_MAX_HW_QUEUES=2 run)"
