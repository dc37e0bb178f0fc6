export TMPDIR=/tmp; mkdir -p gpurun_out
python3 bench.py > gpurun_out/bench_r05a.json 2> gpurun_out/bench_r05a.err; echo "bench rc=$?"; tail -3 gpurun_out/bench_r05a.err
python3 - <<'PY'
import json
j=json.loads(open('gpurun_out/bench_r05a.json').read().strip().splitlines()[-1])
print('value', j['value'], 'ms/step', j['ms_per_step'])
r=j['roofline']; print('frac', r.get('frac'), 'frac_executed', r.get('frac_executed'), 'whole_step', r.get('whole_step'))
print('batch1', j.get('batch1_latency'))
print('K8', (j.get('concurrent_single_pairs') or {}).get('K8'))
print('large', j.get('large_configs_n1'))
print('train', j.get('training_step'))
print('parity', (j.get('parity_check') or {}).get('ok'))
PY
