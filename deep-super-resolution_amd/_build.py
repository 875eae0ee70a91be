"""Build csrc/libdsr_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

Each .hip file is compiled to its own object (in parallel, re-done only when the file or a header is newer) and
the objects are linked into one shared library; nothing is fetched or installed."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
SOURCES = ["conv_gemm.hip", "conv_gemm_persist.hip", "conv_wgrad.hip", "conv_wgrad_tile.hip", "conv_smalln.hip",
           "conv_c64.hip", "conv_halo64.hip", "conv_cin8.hip", "conv_rgb9.hip", "conv_dgrad_s2.hip", "conv_first_bwd.hip", "conv_first2.hip", "conv_api.hip", "pointwise.hip", "linear.hip",
           "resample.hip", "metrics.hip", "data.hip"]
SO = os.path.join(CSRC, "libdsr_hip.so")
# -Werror=return-type: a C-ABI entry point that flows off its end without `return` is undefined behaviour (hipcc -O3
# emits no `ret`, the call runs into the next function) -- that was the round-1 host segfault in dsr_pw_bn_eval_affine.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Werror=return-type"]


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "dsr_hip.h"))
    return hs


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _obj(src):
    return os.path.join(OBJ, src[:-4] + ".o")


def _obj_stale(src, hdr_time):
    o = _obj(src)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    return os.path.getmtime(os.path.join(CSRC, src)) > t or hdr_time > t


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, s) for s in _sources()] + _headers()
    return any(os.path.getmtime(d) > t for d in deps)


def compile_cmd(src):
    return ["hipcc"] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", _obj(src)]


def build(force=False, verbose=False):
    if not force and not _stale():
        return SO
    os.makedirs(OBJ, exist_ok=True)
    hdr_time = max(os.path.getmtime(h) for h in _headers())
    todo = [s for s in _sources() if force or _obj_stale(s, hdr_time)]

    def one(src):
        cmd = compile_cmd(src)
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r

    workers = max(1, min(len(todo), (os.cpu_count() or 2)))
    if todo:
        with ThreadPoolExecutor(workers) as ex:
            for src, r in ex.map(one, todo):
                if r.returncode != 0:
                    raise RuntimeError(f"hipcc failed on {src}:\n" + r.stdout + r.stderr)
    link = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + [_obj(s) for s in _sources()]
    if verbose:
        print(" ".join(link), flush=True)
    r = subprocess.run(link, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + r.stdout + r.stderr)
    return SO
