"""Timeline statistics of a rocprofv3 --kernel-trace CSV: is the GPU ever idle, and how much do the engines' kernels overlap?

    python tools/trace_gaps.py <dir with *kernel_trace.csv> [out.json]

Prints: wall span, union of busy intervals (at least one kernel running), sum of kernel durations (> union when kernels of
the two engine streams overlap), idle share, histogram of the idle gaps, and the kernels that most often run ALONE for long
(no overlap partner) - the ones whose own occupancy decides the step time."""
import csv, glob, json, os, sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
assert files, f"no *kernel_trace.csv under {d}"
ev = []
for f in files:
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
ev.sort()
t0, t1 = ev[0][0], max(e[1] for e in ev)
# skip the set-up phase: statistics over the last 60 % of the span (the timed steps)
cut = t0 + int(0.4 * (t1 - t0))
ev = [e for e in ev if e[0] >= cut]
t0 = ev[0][0]
span = t1 - t0
busy = 0
gaps = []
cur_s, cur_e = ev[0][0], ev[0][1]
for s, e, _ in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
total = sum(e - s for s, e, _ in ev)
# time each kernel runs alone (sweep line)
pts = []
for i, (s, e, _) in enumerate(ev):
    pts.append((s, 1, i)); pts.append((e, -1, i))
pts.sort()
active = set()
alone = defaultdict(int)
dur = defaultdict(int)
prev = pts[0][0]
for t, k, i in pts:
    if len(active) == 1:
        alone[ev[next(iter(active))][2]] += t - prev
    prev = t
    if k == 1:
        active.add(i)
    else:
        active.discard(i)
for s, e, n in ev:
    dur[n] += e - s
hist = {"<2us": 0, "2-5us": 0, "5-20us": 0, "20-100us": 0, ">100us": 0}
for g in gaps:
    hist["<2us" if g < 2000 else "2-5us" if g < 5000 else "5-20us" if g < 20000 else "20-100us" if g < 100000 else ">100us"] += g
out = {"kernels": len(ev), "span_ms": span / 1e6, "busy_union_ms": busy / 1e6, "sum_kernel_ms": total / 1e6,
       "idle_share": 1 - busy / span, "overlap_factor": total / busy,
       "idle_ms_by_gap_size": {k: v / 1e6 for k, v in hist.items()}, "gaps": len(gaps),
       "alone_top": [{"kernel": n[:90], "alone_ms": a / 1e6, "total_ms": dur[n] / 1e6}
                     for n, a in sorted(alone.items(), key=lambda x: -x[1])[:25]]}
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
