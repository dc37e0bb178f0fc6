"""Top kernels of the step with their rooflines: one JSON table from the committed rocprofv3 summaries.

    python tools/kernel_table.py <kernel_stats.csv> <pmc_issue.csv> <pmc_fetch.csv> <pmc_write.csv> <out.json> [top]

Inputs (tools/kstat_single.sh, tools/profile_bench.sh --pairs 128 --streams 1: ONE engine, kernels serialised):
  kernel_stats : rocprofv3 --kernel-trace --stats (durations)
  pmc_*        : tools/pmc_table.py tables (sum of each counter over a kernel's dispatches)
Per kernel: share of the summed kernel time, average duration, HBM-side bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE in KB
(FETCH_SIZE doubled: the gfx950 note of MI355X_MICROARCH.md; Infinity-Cache hits are counted), TB/s against 8 TB/s, matrix-pipe
busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs ... ) and vector-ALU issue = 4 cycles per
SQ_INSTS_VALU over the SIMD-cycles of the launch.  bench.py copies the table into roofline.kernels."""
import csv
import json
import re
import sys

stats_f, issue_f, fetch_f, write_f, out_f = sys.argv[1:6]
top = int(sys.argv[6]) if len(sys.argv) > 6 else 8


def norm(n):
    n = n.replace("dsir::(anonymous namespace)::", "").replace("dsir::", "").replace("void ", "")
    n = re.sub(r"\(.*", "", n)
    m = re.match(r"_ZN4dsir12_GLOBAL__N_1\d+([a-z_0-9]+?)I", n)
    return (m.group(1) + "<mangled>") if m else n.strip()


def table(f):
    return {norm(r["kernel"]): r for r in csv.DictReader(open(f)) if r["kernel"] != "TOTAL"}


st = {}
for r in csv.DictReader(open(stats_f)):
    st[norm(r["Name"])] = r
iss, fe, wr = table(issue_f), table(fetch_f), table(write_f)
total_ns = sum(float(r["TotalDurationNs"]) for r in st.values())
rows = sorted(st.items(), key=lambda kv: -float(kv[1]["TotalDurationNs"]))[:top]
out = []
for name, r in rows:
    avg_us = float(r["AverageNs"]) / 1e3
    e = {"kernel": name, "calls": int(r["Calls"]), "time_share": round(float(r["TotalDurationNs"]) / total_ns, 4), "avg_us": round(avg_us, 1)}
    if name in fe and name in wr:
        nf, nw = int(fe[name]["dispatches"]), int(wr[name]["dispatches"])
        byts = (2.0 * float(fe[name]["FETCH_SIZE"]) / nf + float(wr[name]["WRITE_SIZE"]) / nw) * 1e3      # counters are in KB
        e["hbm_mb_per_launch"] = round(byts / 1e6, 1)
        e["hbm_tb_s"] = round(byts / (avg_us * 1e-6) / 1e12, 2)
        e["hbm_frac_of_8tb_s"] = round(byts / (avg_us * 1e-6) / 8e12, 3)
    if name in iss:
        n = int(iss[name]["dispatches"])
        cyc = float(iss[name]["GRBM_GUI_ACTIVE"]) / n / 8.0            # GRBM_GUI_ACTIVE sums the 8 XCDs
        e["valu_issue_busy"] = round(float(iss[name]["SQ_INSTS_VALU"]) / n * 4.0 / 1024.0 / cyc, 3)
        e["mfma_pipe_busy"] = round(float(iss[name]["SQ_VALU_MFMA_BUSY_CYCLES"]) / n / 1024.0 / cyc, 3)
    out.append(e)
res = {"kernels": out, "total_kernel_ms": round(total_ns / 1e6, 2),
       "note": "one engine, 128 pairs x 5000 points per call, kernels serialised (tools/kstat_single.sh + tools/profile_bench.sh --pairs 128 "
               "--streams 1); time_share of the summed kernel time; HBM-side bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction, "
               "Infinity-Cache hits counted); busy figures = issue cycles / SIMD-cycles of the launch"}
json.dump(res, open(out_f, "w"), indent=1)
for e in out:
    print(e)
