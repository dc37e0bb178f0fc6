"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over `tools/microbench.py match` into
profiles/nn_match_pmc.json, the per-launch HBM traffic of nn_match_kernel that bench.py reports as
roofline.traffic.  Correction per /opt/skills/guides/MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE
counts a wide coalesced read at half its bytes (double it); WRITE_SIZE is exact; both are in KiB."""
import csv, glob, json, sys

fetch_dir, write_dir, pairs, points, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
# kernels that make up one arg-min launch: the exhaustive kernel, or the screened path's passes (argv[6] = "screened")
NAMES = ("screen_kernel", "exact_pick_kernel", "nn_match_kernel", "unpack_listed_kernel") if len(sys.argv) > 6 and sys.argv[6] == "screened" else ("nn_match_kernel",)


def mean_counter(d, name):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = {}
    for r in csv.DictReader(open(f)):
        for n in NAMES:
            if n in r["Kernel_Name"] and r["Counter_Name"] == name:
                per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))   # one entry per template instance
    total = sum(sum(v) / len(v) for v in per.values())       # one of each kernel per launch
    return total, min(len(v) for v in per.values())


fs, n1 = mean_counter(fetch_dir, "FETCH_SIZE")
ws, n2 = mean_counter(write_dir, "WRITE_SIZE")
res = {"kernel": "+".join(NAMES), "pairs": pairs, "points": points, "fetch_size_kib_raw": fs, "write_size_kib": ws,
       "launches_sampled": [n1, n2], "hbm_bytes_per_launch": (2.0 * fs + ws) * 1024.0,
       "algorithmic_bytes_per_launch": pairs * (2 * points * 64 * 4 + 2 * points * 4 + points * 8),
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); FETCH_SIZE doubled per the gfx950 note"}
json.dump(res, open(out, "w"), indent=1)
print(res)
