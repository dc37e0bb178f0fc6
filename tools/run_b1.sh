export TMPDIR=/tmp; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "few_clouds or pyramid or knn or kabsch or register_with_supplied" > gpurun_out/t_b1.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/t_b1.log
rm -rf /tmp/b1t && rocprofv3 --kernel-trace -d /tmp/b1t --output-format csv -- python3 tools/b1_timeline.py run 5000 1 > gpurun_out/b1_run.txt 2>&1; tail -3 gpurun_out/b1_run.txt
python3 tools/b1_timeline.py report /tmp/b1t gpurun_out/b1_timeline_new.txt | tail -5
head -64 gpurun_out/b1_timeline_new.txt
python3 tools/b1_timeline.py run 5000 1 2>&1 | grep -i "graph\|ms"
DSIR_TUNING=1 DSIR_NO_PYRAMID_MERGE=1 DSIR_KABSCH_STREAM=1 python3 tools/b1_timeline.py run 5000 1 2>&1 | grep -i "graph\|ms"
