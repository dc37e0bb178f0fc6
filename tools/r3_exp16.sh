#!/bin/bash
# round-3: pruned search, second build (MFMA bound pass, dense kernel separate) - exactness, C2/C3/C5 A/B, kernel breakdown
out=gpurun_out
bash tools/r3_exp15.sh || exit 1
export DSIR_PRUNE_MIN_K=8192
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/r3_prune_prof -o c3 -- python3 $GRAFT_REPO_ROOT/bench.py --points 16384 --feat-len 4 --shape kitti --pairs 32 --steps 4 --warmup 1 --no-cpu-baseline --no-companion --no-latency > $GRAFT_REPO_ROOT/$out/r3_prune_prof_c3.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r3_prune_prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
PY
