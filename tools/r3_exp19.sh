#!/bin/bash
# round-3: pruned search - the mechanism's own cost: every tile kept (DSIR_PRUNE_KEEP_ALL) vs the dense kernel
out=$GRAFT_REPO_ROOT/gpurun_out
export DSIR_PRUNE_KEEP_ALL=1
export DSIR_PRUNE_ID_ROWS=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/r3_prune_prof -o c3all -- python3 $GRAFT_REPO_ROOT/bench.py --points 16384 --feat-len 4 --shape kitti --pairs 32 --steps 4 --warmup 1 --no-cpu-baseline --no-companion --no-latency > $out/r3_prune_prof_c3all.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/r3_prune_prof -o c5all -- python3 $GRAFT_REPO_ROOT/bench.py --points 65536 --partial-overlap --pairs 4 --steps 3 --warmup 1 --no-cpu-baseline --no-companion --no-latency > $out/r3_prune_prof_c5all.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
for f in sorted(glob.glob("gpurun_out/r3_prune_prof/c?all_kernel_stats.csv")):
    print(f)
    for r in list(csv.DictReader(open(f)))[:40]:
        if "screen_kernel" in r["Name"] or "tile_bound" in r["Name"] or "Sort" in r["Name"] or "sort" in r["Name"]:
            print(f"{r['Name'][:80]:80s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
PY
