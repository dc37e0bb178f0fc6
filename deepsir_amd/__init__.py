"""MI355X-native registration engine behind the reference's ``network.model.Network`` interface."""
import os as _os

# hipGraph replay (include/dsir.h, dsir_enable_graph) on this ROCm: the runtime's "graph packet capture" fast path replays a
# captured registration WRONGLY from its third launch on when the host has waited between launches (wrong poses, then GPU
# memory faults: profiles/README.md, round 4; tools/graph_replay_check.py reproduces it).  The ROCclr flag below switches that
# path off at no measurable cost (3.69 ms per single-pair replay either way).  The HIP runtime reads it when it INITIALISES,
# so it has to be in the environment before the first GPU call of the process: importing this package before touching the GPU
# is enough; ``graph_replay_safe()`` tells whether that held, and ``Engine.enable_graph`` refuses to capture when it did not.
_FLAG = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
_preset = _os.environ.get(_FLAG)
_os.environ.setdefault(_FLAG, "0")
# (Streams -> hardware queues: the runtime hands its GPU_MAX_HW_QUEUES queues - four by default - to streams in creation order, and two
# streams on one queue do not overlap.  Nothing is set here: with eight queues three or more engines in flight run four times SLOWER than
# one after the other, with four they overlap; what had put the serving scheduler's two engines on one queue in round 5 was a pair of idle
# auxiliary streams per context, created on demand since - csrc/engine.hip, ensure_aux_streams; profiles/r05_serving_queues.txt.)


def graph_replay_safe() -> bool:
    """True when the flag above reached the HIP runtime: it was ALREADY in the environment when this package was imported, or
    the process had not opened the GPU driver yet at that moment (``_gpu_untouched``)."""
    return _SAFE and _os.environ.get(_FLAG) == "0"


def _gpu_untouched() -> bool:
    """Has this process initialised the HIP runtime yet?  torch's own state does not say: ``torch.cuda.is_available()`` goes
    through hipGetDeviceCount, which initialises the runtime (and reads the ROCclr flags) while ``torch.cuda.is_initialized()``
    stays False, and any other library can initialise HIP without torch.  What every initialisation does is open the
    compute driver node: a process with no descriptor on /dev/kfd has not started the runtime.  Where /proc cannot be read the
    answer is the conservative one (touched)."""
    try:
        for fd in _os.listdir("/proc/self/fd"):
            try:
                if _os.readlink("/proc/self/fd/" + fd) == "/dev/kfd":
                    return False
            except OSError:
                continue
    except OSError:
        return False
    import sys
    t = sys.modules.get("torch")
    if t is None:
        return True
    try:
        return not t.cuda.is_initialized()
    except Exception:
        return False


_SAFE = (_preset == "0") or _gpu_untouched()
