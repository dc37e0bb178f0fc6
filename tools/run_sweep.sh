export TMPDIR=/tmp; mkdir -p gpurun_out
for cfg in "256 2" "384 3" "384 2" "512 2" "512 4" "768 3"; do set -- $cfg
  python3 bench.py --pairs $1 --streams $2 --steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-companion --timed-only 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pairs $1 streams $2', j['value'], j['ms_per_step'])"
done
