#!/bin/bash
# Evidence run on the GPU box (called through gpurun): kernel trace + the PMC passes of ONE bench.py command line.
#   tools/profile_bench.sh <tag> [bench.py arguments...]
# Counters are collected in their own passes (never combined with tracing domains).  Summaries land in
# gpurun_out/<tag>_*; copy what is to be judged into profiles/.
set -o pipefail
tag=$1; shift
args="--timed-only --steps 3 --warmup 1 $*"   # 4 identical steps (1 warm-up + 3 timed): tools/step_hbm.py counts them from the table
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
run() { local name=$1; shift; rm -rf /tmp/prof_$name; rocprofv3 "$@" -d /tmp/prof_$name --output-format csv -- python3 bench.py $args > $out/${tag}_$name.json 2> $out/${tag}_$name.err; }
run trace --kernel-trace --stats || { echo "trace run failed"; tail -5 $out/${tag}_trace.err; exit 1; }
cp "$(find /tmp/prof_trace -name '*kernel_stats.csv' | head -1)" $out/${tag}_kernel_stats.csv
echo "kernel stats: $(wc -l < $out/${tag}_kernel_stats.csv) rows"
run pmc_issue --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE || { echo "pmc issue run failed"; tail -5 $out/${tag}_pmc_issue.err; exit 1; }
python3 tools/pmc_table.py /tmp/prof_pmc_issue $out/${tag}_pmc_issue.csv > /dev/null
run pmc_fetch --pmc FETCH_SIZE GRBM_GUI_ACTIVE || { echo "pmc fetch run failed"; exit 1; }
python3 tools/pmc_table.py /tmp/prof_pmc_fetch $out/${tag}_pmc_fetch.csv > /dev/null
run pmc_write --pmc WRITE_SIZE GRBM_GUI_ACTIVE || { echo "pmc write run failed"; exit 1; }
python3 tools/pmc_table.py /tmp/prof_pmc_write $out/${tag}_pmc_write.csv > /dev/null
tail -2 $out/${tag}_pmc_fetch.csv; tail -1 $out/${tag}_pmc_write.csv
echo done
