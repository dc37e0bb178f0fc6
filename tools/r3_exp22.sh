#!/bin/bash
# round-3: class entries tested against the best exact distance (exact_pick_kernel), kabsch passes unrolled: screening parity + effect
out=gpurun_out
python3 -m pytest tests/test_gpu_large_configs.py tests/test_gpu_bench_config.py tests/test_gpu_parity.py tests/test_gpu_screen_bound.py -m gpu -x -q > $out/r3_e22_tests.log 2>&1; echo "tests rc=$?"; tail -2 $out/r3_e22_tests.log
python3 bench.py --points 16384 --feat-len 4 --shape kitti --pairs 32 --steps 8 --warmup 2 --no-cpu-baseline --no-companion > $out/r3_e22_c3.json 2> $out/r3_e22_c3.err
python3 bench.py --points 65536 --partial-overlap --pairs 4 --steps 4 --warmup 1 --no-cpu-baseline --no-companion > $out/r3_e22_c5.json 2> $out/r3_e22_c5.err
python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-companion > $out/r3_e22_c2.json 2> $out/r3_e22_c2.err
python3 - <<'PY'
import json
for c in ("c2", "c3", "c5"):
    j = json.load(open(f"gpurun_out/r3_e22_{c}.json"))
    print(c, "pairs/s", j["value"], "kernel ms", j["roofline"].get("avg_launch_ms"), "op ms", j["roofline"]["operation"]["avg_ms"], "undecided", j["screening"]["undecided_row_rate"],
          "batch1", (j.get("batch1_latency") or {}).get("ms_per_pair"))
PY
