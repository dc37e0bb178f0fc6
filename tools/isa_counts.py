"""Static instruction mix of the attentive-pooling kernels: compiles csrc/att_pool.hip for gfx950 and
counts, per kernel, the instructions of each class in the emitted ISA (whole kernel: prologue, unit loop, epilogue).

    python3 tools/isa_counts.py > profiles/r05_isa_counts.txt"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "deepsir_amd", "csrc", sys.argv[1] if len(sys.argv) > 1 else "att_pool.hip")
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.dirname(src), "-Wno-unused-value", "--cuda-device-only", "-S", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    s = open(out).read()
print(f"# static ISA instruction counts per kernel of {os.path.basename(src)} (gfx950, hipcc -O3): whole kernel")
for key in re.findall(r"\n(_ZN4dsir\S+):\s+; @", s):
    start = s.index("\n" + key + ":") + len(key) + 2
    body = s[start:s.index(".Lfunc_end", start)]
    lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith((".", ";"))]
    c = collections.Counter(l.split()[0] for l in lines)
    cls = lambda pred: sum(v for k, v in c.items() if pred(k))
    name = subprocess.run(["c++filt", key], capture_output=True, text=True).stdout.strip().replace("dsir::(anonymous namespace)::", "")
    print(f"{name[:60]:60s} total {len(lines):5d}  VALU {cls(lambda k: k.startswith('v_') and not k.startswith('v_mfma')):5d}  MFMA {cls(lambda k: k.startswith('v_mfma')):4d}  "
          f"LDS {cls(lambda k: k.startswith('ds_')):4d}  VMEM {cls(lambda k: k.startswith(('global_', 'buffer_'))):4d}  SALU+wait {cls(lambda k: k.startswith('s_')):4d}  "
          f"exp {c['v_exp_f32_e32']:3d}  cvt_pk_f16 {c['v_cvt_pk_f16_f32']:3d}  cndmask {cls(lambda k: k.startswith('v_cndmask')):3d}  max {cls(lambda k: k.startswith('v_max_f32')):3d}")
