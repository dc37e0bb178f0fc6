"""FETCH_SIZE / WRITE_SIZE tables of tools/profile_bench.sh (bench.py --timed-only: every dispatch belongs to one of the
warmup + steps identical steps)  ->  profiles/step_hbm.json, read by bench.py as roofline.hbm_gbps.
The number of steps is COUNTED from the table (score_point_kernel runs once per engine call; engine calls / streams = steps); the
<calls> argument is only cross-checked.  Rounds 3 and 4 (until this check existed) passed 5 for a command line that runs 4 steps -
bench.py had lost its separate allocation call in round 3 - and under-reported the traffic per pair by a factor 4/5: round 3's
432 MB per pair was 540, round 4's 362 was 452 (profiles/README.md, "HBM per pair: the corrected divisor").
Units and the gfx950 correction as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: both counters in KiB,
FETCH_SIZE x 2 (a wide coalesced read is tallied at half its bytes), WRITE_SIZE exact; Infinity-Cache hits are counted.
    python tools/step_hbm.py <tag>_pmc_fetch.csv <tag>_pmc_write.csv <calls> <pairs> <points> <streams> <iters> out.json"""
import csv, json, sys
fetch, write, calls, pairs, points, streams, iters, out = sys.argv[1], sys.argv[2], *map(int, sys.argv[3:8]), sys.argv[8]
def total(path, col):
    return [float(r[col]) for r in csv.DictReader(open(path)) if r['kernel'] == 'TOTAL'][0]
fs, ws = total(fetch, 'FETCH_SIZE'), total(write, 'WRITE_SIZE')
engine_calls = [float(r['dispatches']) for r in csv.DictReader(open(fetch)) if r['kernel'].startswith('score_point_kernel')]
if engine_calls:
    counted = int(round(engine_calls[0] / streams))
    if counted != calls:
        print(f"step_hbm: {calls} steps claimed, {counted} counted from the table (score_point_kernel x {int(engine_calls[0])} on {streams} streams): using {counted}", file=sys.stderr)
    calls = counted
per_step = (2.0 * fs + ws) * 1024.0 / calls
res = {"pairs": pairs, "points": points, "streams": streams, "iters": iters, "steps_profiled": calls,
       "fetch_size_kib_raw_total": fs, "write_size_kib_total": ws, "hbm_bytes_per_step": per_step,
       "hbm_bytes_per_pair": per_step / pairs,
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --timed-only`, summed over every "
                 "dispatch, / steps (counted from the table: score_point_kernel dispatches / streams); FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md; Infinity-Cache hits included"}
json.dump(res, open(out, "w"), indent=1)
print(res)
