"""Write the two binary inputs of examples/cabi_register (see its header): a state-dict and a batch of pairs."""
import struct
import sys

import numpy as np


def write_weights(path, sd):
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(sd)))
        for k, v in sd.items():
            name = k.encode()
            f.write(struct.pack("<i", len(name)) + name)
            a = np.asarray(v)
            if k.endswith("num_batches_tracked"):
                f.write(struct.pack("<i", -1))
                continue
            a = np.ascontiguousarray(a, dtype=np.float32)
            f.write(struct.pack("<i", a.ndim))
            f.write(struct.pack("<%dq" % a.ndim, *a.shape))
            f.write(a.tobytes())


def write_pairs(path, src, ref):
    src, ref = np.ascontiguousarray(src, np.float32), np.ascontiguousarray(ref, np.float32)
    assert src.shape == ref.shape and src.ndim == 3
    with open(path, "wb") as f:
        f.write(struct.pack("<iii", *src.shape))
        f.write(src.tobytes())
        f.write(ref.tobytes())


if __name__ == "__main__":
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.synth import make_batch
    from deepsir_amd.weights import generate_state_dict
    out = sys.argv[1] if len(sys.argv) > 1 else "."
    cfg = NetConfig()
    write_weights(os.path.join(out, "weights.bin"), generate_state_dict(cfg, 0))
    b = make_batch(2048, [1, 2], 3)
    write_pairs(os.path.join(out, "pairs.bin"), b["points_src"], b["points_ref"])
    print("wrote", out)
