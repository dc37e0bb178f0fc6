#!/bin/bash
# One call on the GPU box (through gpurun): every number profiles/README.md quotes for the current build.
#   bash tools/collect_evidence.sh <tag>          -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
set -o pipefail
tag=${1:-r05}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
echo "== default bench"; python3 bench.py > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.err || { tail -5 $out/${tag}_bench_default.err; exit 1; }
echo "== kernel trace + PMC passes of the bench command"; bash tools/profile_bench.sh $tag || exit 1
python3 tools/step_hbm.py $out/${tag}_pmc_fetch.csv $out/${tag}_pmc_write.csv 4 256 5000 2 5 $out/${tag}_step_hbm.json > /dev/null || exit 1
python3 tools/trace_gaps.py /tmp/prof_trace $out/${tag}_trace_gaps.json > /dev/null || echo "(trace_gaps skipped)"
echo "== one engine (the launches of the roofline pass): kernel trace, its own PMC passes, the per-kernel roofline table"
bash tools/kstat_single.sh $tag > $out/${tag}_single_top.txt 2>&1 || { tail -5 $out/${tag}_single_top.txt; exit 1; }
mkdir -p $out/single && bash tools/profile_bench.sh ${tag}s --pairs 128 --streams 1 > $out/${tag}s_profile.log 2>&1 || { tail -5 $out/${tag}s_profile.log; exit 1; }
python3 tools/kernel_table.py $out/${tag}_kernel_stats_single_engine.csv $out/${tag}s_pmc_issue.csv $out/${tag}s_pmc_fetch.csv $out/${tag}s_pmc_write.csv $out/${tag}_kernel_table.json 8 > /dev/null || exit 1
echo "== one pair in flight"
rm -rf /tmp/prof_b1; rocprofv3 --kernel-trace --stats -d /tmp/prof_b1 --output-format csv -- python3 bench.py --pairs 1 --streams 1 --steps 20 --warmup 2 --timed-only > $out/${tag}_b1.json 2> $out/${tag}_b1.err || exit 1
cp "$(find /tmp/prof_b1 -name '*kernel_stats.csv' | head -1)" $out/${tag}_batch1_kernel_stats.csv
echo "== HBM traffic of one screened arg-min launch (128 pairs)"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/nm_$c
  rocprofv3 --pmc $c -d /tmp/nm_$c --output-format csv -- python3 tools/microbench.py match_screened --clouds 128 --reps 5 > $out/${tag}_nm_$c.log 2>&1 || { tail -3 $out/${tag}_nm_$c.log; exit 1; }
  python3 tools/pmc_table.py /tmp/nm_$c $out/${tag}_nn_match_pmc_$c.csv > /dev/null
done
python3 tools/pmc_traffic.py /tmp/nm_FETCH_SIZE /tmp/nm_WRITE_SIZE 128 5000 $out/${tag}_nn_match_pmc.json screened > /dev/null || exit 1
echo "== other BASELINE configurations"
python3 bench.py --points 2048 --pairs 512 --steps 10 > $out/${tag}_bench_c1.json 2> $out/${tag}_bench_c1.err || { tail -3 $out/${tag}_bench_c1.err; exit 1; }
python3 bench.py --points 16384 --feat-len 4 --shape kitti --pairs 128 --streams 4 --steps 6 > $out/${tag}_bench_c3.json 2> $out/${tag}_bench_c3.err || { tail -3 $out/${tag}_bench_c3.err; exit 1; }
python3 bench.py --points 65536 --partial-overlap --pairs 16 --steps 4 --warmup 1 --no-cpu-baseline > $out/${tag}_bench_c5.json 2> $out/${tag}_bench_c5.err || { tail -3 $out/${tag}_bench_c5.err; exit 1; }
for f in default c1 c3 c5; do python3 - $out/${tag}_bench_$f.json $f <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(sys.argv[2], j["value"], "pairs/s", "batch1", j.get("batch1_latency", {}).get("ms_per_pair"), "frac", j["roofline"].get("frac"),
      "parity", (j.get("parity_check") or {}).get("ok"), "undecided", (j.get("screening") or {}).get("undecided_row_rate"))
PY
done
echo done
