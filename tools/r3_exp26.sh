#!/bin/bash
# round-3: the inference path read by GRID: launches with few workgroups and long durations (one engine, 128 pairs)
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pg; rocprofv3 --kernel-trace --output-format csv -d /tmp/pg -o g -- python3 $GRAFT_REPO_ROOT/bench.py ${BENCH_ARGS:---pairs 128 --streams 1 --steps 3 --warmup 1 --timed-only} > $out/r3_grid.log 2>&1 || { tail -5 $out/r3_grid.log; exit 1; }
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, collections, glob, re
f = glob.glob("/tmp/pg/**/*kernel_trace.csv", recursive=True)[0]
g = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = re.sub(r'\(.*', '', r['Kernel_Name'].replace('dsir::(anonymous namespace)::', '').replace('void ', ''))[:48]
    wg = (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))) * (int(r['Grid_Size_Y']) // max(1, int(r['Workgroup_Size_Y']))) * (int(r['Grid_Size_Z']) // max(1, int(r['Workgroup_Size_Z'])))
    g[(n, wg, int(r['Workgroup_Size_X']))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in g.values())
rows = sorted(g.items(), key=lambda kv: -sum(kv[1]))
with open("gpurun_out/r3_grid_table.txt", "w") as o:
    o.write(f"total kernel ms {tot/1e3:.1f}\n")
    for (n, wg, ws), v in rows[:70]:
        o.write(f"{n:48s} wgs {wg:7d} x{ws:5d} calls {len(v):5d} avg_us {sum(v)/len(v):8.1f} share {100*sum(v)/tot:5.2f} %\n")
print(open("gpurun_out/r3_grid_table.txt").read())
PY
