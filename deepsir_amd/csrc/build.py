"""Build libdsir.so (hipcc, gfx950 only).  Used by __graft_entry__.build()."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
SOURCES = ["engine.hip", "pw_gemm.hip", "pw_stream.hip", "pw_tile.hip", "walk.hip", "att_pool.hip", "lse_uv.hip", "head_mlp.hip", "head_mlp_h.hip", "agg_chain.hip", "agg_chain_h.hip", "select.hip", "misc.hip", "knn.hip", "knn_grid.hip", "score.hip", "nn_match.hip", "nn_screen.hip", "nn_prune.hip", "kabsch.hip", "icp.hip", "finetune.hip", "align_loss.hip", "train_ops.hip", "metrics.hip", "preprocess.hip"]
HEADERS = ["kernels.h", "device_utils.h", "svd3.h", "pw_tile_body.h", "att_pool_body.h", "misc_body.h", os.path.join(ROOT, "include", "dsir.h"), os.path.join(ROOT, "include", "dsir_train.h")]
OUT = os.path.join(PKG, "libdsir.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
         "-I" + HERE, "-Wall", "-Wno-unused-function", "-Wno-unused-value"]


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [h if os.path.isabs(h) else os.path.join(HERE, h) for h in HEADERS]
    jobs = []
    for s in SOURCES:
        src = os.path.join(HERE, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or not _newer(obj, [src] + hdrs):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for msg in ex.map(run, jobs):
            if verbose and msg:
                print(msg)
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or not os.path.exists(OUT):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    # the plain C++ host over the C ABI (examples/cabi_register.cpp): links against libdsir.so only
    ex_src = os.path.join(ROOT, "examples", "cabi_register.cpp")
    ex_bin = os.path.join(ROOT, "examples", "cabi_register")
    if os.path.exists(ex_src) and (force or not _newer(ex_bin, [ex_src, OUT, os.path.join(ROOT, "include", "dsir.h"), os.path.join(ROOT, "include", "dsir_train.h")])):
        run([hipcc, "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), ex_src, "-o", ex_bin,
             "-L" + PKG, "-ldsir", "-Wl,-rpath," + PKG, "-Wl,-rpath,$ORIGIN/../deepsir_amd"])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
