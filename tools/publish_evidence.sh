#!/bin/bash
# tools/publish_evidence.sh <tag>: copy what tools/collect_evidence.sh <tag> left under gpurun_out/ (scratch) into profiles/ (tracked)
tag=${1:-r05}
cd "$(dirname "$0")/.."
for f in bench_default.json bench_c1.json bench_c3.json bench_c5.json kernel_stats.csv kernel_stats_single_engine.csv batch1_kernel_stats.csv \
         pmc_fetch.csv pmc_issue.csv pmc_write.csv kernel_table.json trace_gaps.json nn_match_pmc_FETCH_SIZE.csv nn_match_pmc_WRITE_SIZE.csv \
         step_hbm.json nn_match_pmc.json; do
  [ -f gpurun_out/${tag}_$f ] && cp gpurun_out/${tag}_$f profiles/${tag}_$f || echo "missing gpurun_out/${tag}_$f"
done
for f in pmc_fetch.csv pmc_issue.csv pmc_write.csv; do [ -f gpurun_out/${tag}s_$f ] && cp gpurun_out/${tag}s_$f profiles/${tag}s_$f; done
cp gpurun_out/${tag}_kernel_table.json profiles/kernel_table.json  # read by bench.py (roofline.kernels): the latest collection's table
cp gpurun_out/${tag}_step_hbm.json profiles/step_hbm.json          # read by bench.py (roofline.hbm_gbps)
cp gpurun_out/${tag}_nn_match_pmc.json profiles/nn_match_pmc.json  # read by bench.py (roofline.traffic)
