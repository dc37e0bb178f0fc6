export TMPDIR=/tmp; mkdir -p gpurun_out
show() { python3 -c "
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); cs=j['concurrent_single_pairs']
print(sys.argv[2], 'value', j['value'], 'b1', j['batch1_latency']['ms_per_pair'], {k:v['pairs_per_s'] for k,v in cs.items() if isinstance(v,dict)})" $1 "$2"; }
for q in 4 8; do
GPU_MAX_HW_QUEUES=$q python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-companion > gpurun_out/hwq_$q.json 2>/dev/null; show gpurun_out/hwq_$q.json "bench HWQ=$q"
GPU_MAX_HW_QUEUES=$q python3 tools/k8_sweep.py 8 2>&1 | grep "engines [12] "
GPU_MAX_HW_QUEUES=$q python3 tools/k8_sweep.py 16 2>&1 | grep "engines [12] "
done
