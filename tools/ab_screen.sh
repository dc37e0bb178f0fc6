mkdir -p gpurun_out
for v in ${VARIANTS:-"" bc128 w4 w4b}; do
  if [ -z "$v" ]; then unset DSIR_LIB; else export DSIR_LIB=$PWD/deepsir_amd/libdsir_$v.so; fi
  echo "== variant ${v:-base}"
  python3 - <<'PY'
import numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
from deepsir_amd.arch import NetConfig
from deepsir_amd.engine import Engine
eng = Engine(NetConfig(), 0, max_points=5000, max_pairs=4)
g = torch.Generator().manual_seed(1)
a = torch.nn.functional.normalize(torch.randn(4, 5000, 64, generator=g), dim=2).cuda()
b = torch.nn.functional.normalize(torch.randn(4, 4999, 64, generator=g), dim=2).cuda()
ex = eng.nn_match(a, b); sc, st = eng.nn_match_screened(a, b)
print("equal:", bool(torch.equal(ex, sc)), st)
PY
  bash tools/kstat.sh v$v screen_kernel -- python3 tools/microbench.py match_screened --clouds 64 --reps 20 | grep screen_kernel
done
