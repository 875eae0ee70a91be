"""Build csrc/libdsr_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["conv_gemm.hip", "conv_gemm_persist.hip", "conv_wgrad.hip", "conv_wgrad_tile.hip", "conv_smalln.hip", "conv_c64.hip", "conv_cin8.hip", "conv_first_bwd.hip", "conv_api.hip", "pointwise.hip", "linear.hip", "resample.hip"]
SO = os.path.join(CSRC, "libdsr_hip.so")


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "dsr_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return SO
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Werror=return-type", "-o", SO] + srcs
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    return SO
