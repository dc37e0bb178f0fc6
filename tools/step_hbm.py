"""FETCH_SIZE / WRITE_SIZE tables of tools/profile_bench.sh (bench.py --timed-only: every dispatch belongs to one of the
1 + warmup + steps identical register calls)  ->  profiles/step_hbm.json, read by bench.py as roofline.hbm_gbps.
Units and the gfx950 correction as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: both counters in KiB,
FETCH_SIZE x 2 (a wide coalesced read is tallied at half its bytes), WRITE_SIZE exact; Infinity-Cache hits are counted.
    python tools/step_hbm.py <tag>_pmc_fetch.csv <tag>_pmc_write.csv <calls> <pairs> <points> <streams> <iters> out.json"""
import csv, json, sys
fetch, write, calls, pairs, points, streams, iters, out = sys.argv[1], sys.argv[2], *map(int, sys.argv[3:8]), sys.argv[8]
def total(path, col):
    return [float(r[col]) for r in csv.DictReader(open(path)) if r['kernel'] == 'TOTAL'][0]
fs, ws = total(fetch, 'FETCH_SIZE'), total(write, 'WRITE_SIZE')
per_step = (2.0 * fs + ws) * 1024.0 / calls
res = {"pairs": pairs, "points": points, "streams": streams, "iters": iters, "register_calls_profiled": calls,
       "fetch_size_kib_raw_total": fs, "write_size_kib_total": ws, "hbm_bytes_per_step": per_step,
       "hbm_bytes_per_pair": per_step / pairs,
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --timed-only`, summed over every "
                 "dispatch, / register calls; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md; Infinity-Cache hits included"}
json.dump(res, open(out, "w"), indent=1)
print(res)
