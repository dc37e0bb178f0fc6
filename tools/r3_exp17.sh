#!/bin/bash
# round-3: pruned search - kernel breakdown of C3 / C5 with the pruned search on
out=$GRAFT_REPO_ROOT/gpurun_out
export DSIR_PRUNE_MIN_K=8192
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/r3_prune_prof -o c3 -- python3 $GRAFT_REPO_ROOT/bench.py --points 16384 --feat-len 4 --shape kitti --pairs 32 --steps 4 --warmup 1 --no-cpu-baseline --no-companion --no-latency > $out/r3_prune_prof_c3.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/r3_prune_prof -o c5 -- python3 $GRAFT_REPO_ROOT/bench.py --points 65536 --partial-overlap --pairs 4 --steps 3 --warmup 1 --no-cpu-baseline --no-companion --no-latency > $out/r3_prune_prof_c5.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
for f in sorted(glob.glob("gpurun_out/r3_prune_prof/**/*kernel_stats.csv", recursive=True)):
    print(f)
    rows = list(csv.DictReader(open(f)))
    for r in rows[:16]:
        print(f"{r['Name'][:80]:80s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
PY
