"""Ordered timeline of ONE single-pair registration replayed from its hipGraph.

    rocprofv3 --kernel-trace -d /tmp/b1t --output-format csv -- python3 tools/b1_timeline.py run [points] [pairs]
    python3 tools/b1_timeline.py report /tmp/b1t gpurun_out/b1_timeline.txt

`run` replays the registration a few times (the last replay is the one reported) and prints the graph's launch census
(dsir_graph_stats).  `report` takes the kernel trace, keeps the last replay (the kernels after the last long idle gap), and writes
one line per launch: start offset, duration, gap to the previous kernel's end, name - plus per-phase sums (KNN pyramid, feature
extractor, per-iteration matching / inlier model / pose) split at the schedule's marker kernels."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(points=5000, pairs=1):
    os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
    import time
    import torch
    import deepsir_amd  # noqa: F401
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine
    from deepsir_amd.synth import make_batch
    from deepsir_amd.weights import generate_state_dict
    cfg = NetConfig(feat_len=3)
    eng = Engine(cfg, 0, max_points=points, max_pairs=pairs)
    eng.load_state_dict(generate_state_dict(cfg, 0))
    b = make_batch(points, [10_000 + i for i in range(pairs)], 3)
    src, ref = torch.from_numpy(b["points_src"]).cuda(), torch.from_numpy(b["points_ref"]).cuda()
    out = {"transforms": torch.empty((pairs, 5, 3, 4), device="cuda")}
    eng.enable_graph(True)
    for _ in range(3):
        eng.register(src, ref, 5, want_aux=False, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        eng.register(src, ref, 5, want_aux=False, out=out, sync=False)
    eng.sync()
    ms = (time.perf_counter() - t0) / n * 1e3
    time.sleep(0.05)            # an idle gap in front of the replay the report keeps
    eng.register(src, ref, 5, want_aux=False, out=out)
    print("graph", eng.graph_stats(), "ms_per_registration", round(ms, 4), "pairs", pairs, flush=True)


def report(d, out_path):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    assert files, d
    ev = []
    for f in files:
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    ev.sort()
    cut = 0
    for i in range(1, len(ev)):
        if ev[i][0] - ev[i - 1][1] > 20_000_000:      # 20 ms of nothing: the sleep before the last replay
            cut = i
    ev = ev[cut:]
    t0 = ev[0][0]
    lines, prev_end = [], t0
    tot_k = tot_gap = 0
    by = {}
    for s, e, n in ev:
        short = n.replace("dsir::(anonymous namespace)::", "").replace("void ", "")
        short = short.split("(")[0] if "<" not in short.split("(")[0] else short[: short.index(">") + 1]
        lines.append(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.2f}  gap {(s - prev_end) / 1e3:6.2f}  {short}")
        tot_k += e - s
        tot_gap += max(0, s - prev_end)
        a = by.setdefault(short, [0, 0])
        a[0] += 1; a[1] += e - s
        prev_end = max(prev_end, e)
    span = prev_end - t0
    head = [f"launches {len(ev)}  span {span / 1e3:.1f} us  sum of kernel durations {tot_k / 1e3:.1f} us  sum of gaps {tot_gap / 1e3:.1f} us", ""]
    head.append("per kernel: launches, total us, mean us")
    for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        head.append(f"  {c:4d} {t / 1e3:9.1f} {t / c / 1e3:8.2f}  {k}")
    head.append("")
    open(out_path, "w").write("\n".join(head + lines) + "\n")
    print("\n".join(head[:40]))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(*(int(x) for x in sys.argv[2:4]))
    else:
        report(sys.argv[2], sys.argv[3])
