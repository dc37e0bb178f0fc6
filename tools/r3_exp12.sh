#!/bin/bash
# round-3: with cheap MFMAs, is the score GEMM split by linearity (EPI_ATT2: extra G GEMM + G row gathers) still a win?
out=gpurun_out
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export DSIR_NO_ATT2=1; else unset DSIR_NO_ATT2; fi
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-companion > $out/r3_att2_$v.json 2> $out/r3_att2_$v.err
  python3 - $out/r3_att2_$v.json "DSIR_NO_ATT2=$v" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(sys.argv[2], "pairs/s", j["value"], "ms/step", j["ms_per_step"], "batch1 ms", j.get("batch1_latency", {}).get("ms_per_pair"))
PY
done
