#!/bin/bash
# round-3: kernel breakdown of the training step (8 pairs, graph replay)
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/r3_train_prof -o t8 -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py --pairs 8 --steps 5 > $out/r3_train_prof.log 2>&1 || { tail -5 $out/r3_train_prof.log; exit 1; }
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/r3_train_prof.log
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r3_train_prof/t8_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows[:28]:
    print(f"{r['Name'][:78]:78s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:8.1f} pct {float(r['Percentage']):6.2f}")
PY
