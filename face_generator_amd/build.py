"""Build libfacegen_hip.so (gfx950) in-tree with hipcc.  `python -m face_generator_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["api.hip", "net.hip", "conv_ops.hip", "igemm.hip", "wino.hip", "wino_wgrad.hip", "pointwise.hip", "thin.hip", "step.hip"]
LIB = os.path.join(HERE, "libfacegen_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
         "-DFG_BUILDING"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, measure=False):
    """measure=True builds libfacegen_hip_measure.so (objects csrc/*.m.o) with -DFG_MEASURE: the s_memtime trace kernels, their
    wrong-result DBG variants and the FG_*_TRACE / FG_DEBUG_NOSTORE / FG_THIN_TPW|NT|DBG switches exist ONLY there
    (FACEGEN_HIP_LIB points the binding at it; scripts/gpu.sh trace-* modes do).  The default library has none of them."""
    if measure:
        return _build(force, verbose, ".m.o", LIB.replace(".so", "_measure.so"), FLAGS + ["-DFG_MEASURE"])
    return _build(force, verbose, ".o", LIB, FLAGS)


def _build(force, verbose, osuf, LIB, FLAGS):
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "facegen_hip.h"))
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", osuf))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, measure="--measure" in sys.argv)
