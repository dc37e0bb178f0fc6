"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over `tools/microbench.py match` into
profiles/nn_match_pmc.json, the per-launch HBM traffic of nn_match_kernel that bench.py reports as
roofline.traffic.  Correction per /opt/skills/guides/MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE
counts a wide coalesced read at half its bytes (double it); WRITE_SIZE is exact; both are in KiB."""
import csv, glob, json, sys

fetch_dir, write_dir, pairs, points, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]


def mean_counter(d, name):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
         if "nn_match_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(v) / len(v), len(v)


fs, n1 = mean_counter(fetch_dir, "FETCH_SIZE")
ws, n2 = mean_counter(write_dir, "WRITE_SIZE")
res = {"kernel": "nn_match_kernel", "pairs": pairs, "points": points, "fetch_size_kib_raw": fs, "write_size_kib": ws,
       "launches_sampled": [n1, n2], "hbm_bytes_per_launch": (2.0 * fs + ws) * 1024.0,
       "algorithmic_bytes_per_launch": pairs * (2 * points * 64 * 4 + 2 * points * 4 + points * 8),
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); FETCH_SIZE doubled per the gfx950 note"}
json.dump(res, open(out, "w"), indent=1)
print(res)
