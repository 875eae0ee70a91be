#!/usr/bin/env python3
"""Per-shape timing of the conv kernels through the C ABI (GPU box only; development aid, not part of the product).
    python tools/microbench_conv.py [filter]
Prints algorithmic TFLOP/s (2*M*Cout*KH*KW*Cin) for forward / dgrad / wgrad of the layer shapes of BASELINE config 3."""
import ctypes as C
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "deep-super-resolution_amd"
L = importlib.import_module(PKG + "._lib")
F = importlib.import_module(PKG + ".functional")
lib = L.lib()

SHAPES = [  # name, N, H, W, Cin, Cout, k, stride, pad
    ("G.trunk 64->64 @128", 32, 128, 128, 64, 64, 3, 1, 1),
    ("G.ps0 64->256 @128", 32, 128, 128, 64, 256, 3, 1, 1),
    ("G.ps1 64->256 @256", 32, 256, 256, 64, 256, 3, 1, 1),
    ("G.conv1 9x9 3->64 @128", 32, 128, 128, 3, 64, 9, 1, 4),
    ("G.conv3 9x9 64->3 @512", 32, 512, 512, 64, 3, 9, 1, 4),
    ("D.conv 3->64 @512", 32, 512, 512, 3, 64, 3, 1, 1),
    ("D.b0 64->64 s2 @512", 32, 512, 512, 64, 64, 3, 2, 1),
    ("D.b1 64->128 @256", 32, 256, 256, 64, 128, 3, 1, 1),
    ("D.b2 128->128 s2 @256", 32, 256, 256, 128, 128, 3, 2, 1),
    ("D.b3 128->256 @128", 32, 128, 128, 128, 256, 3, 1, 1),
    ("D.b4 256->256 s2 @128", 32, 128, 128, 256, 256, 3, 2, 1),
    ("D.b5 256->512 @64", 32, 64, 64, 256, 512, 3, 1, 1),
    ("D.b6 512->512 s2 @64", 32, 64, 64, 512, 512, 3, 2, 1),
    ("V.3->64 @224", 32, 224, 224, 3, 64, 3, 1, 1),
    ("V.64->64 @224", 32, 224, 224, 64, 64, 3, 1, 1),
    ("V.64->128 @112", 32, 112, 112, 64, 128, 3, 1, 1),
    ("V.128->128 @112", 32, 112, 112, 128, 128, 3, 1, 1),
    ("V.128->256 @56", 32, 56, 56, 128, 256, 3, 1, 1),
    ("V.256->256 @56", 32, 56, 56, 256, 256, 3, 1, 1),
    ("V.256->512 @28", 32, 28, 28, 256, 512, 3, 1, 1),
    ("V.512->512 @28", 32, 28, 28, 512, 512, 3, 1, 1),
    ("V.512->512 @14", 32, 14, 14, 512, 512, 3, 1, 1),
]


def r8(c):
    return (c + 7) // 8 * 8


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    print(f"{'layer':28s} {'GFLOP':>8s} | {'fwd ms':>8s} {'TF':>6s} | {'dgrad ms':>8s} {'TF':>6s} | {'wgrad ms':>8s} {'TF':>6s}")
    for name, n, h, w, cin, cout, k, s, p in SHAPES:
        if flt and flt not in name:
            continue
        d = L.ConvDesc(L.BF16, n, h, w, cin, cout, k, k, s, p, 0)
        oh, ow = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        fl = 2.0 * n * oh * ow * cout * k * k * cin
        x = (torch.rand(n, h, w, r8(cin), device=dev) - 0.5).to(torch.bfloat16)
        wt = (torch.rand(cout, cin, k, k, device=dev) - 0.5) * 0.1
        bias = torch.zeros(cout, device=dev)
        wf = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 0), dtype=torch.bfloat16, device=dev)
        wd = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 1), dtype=torch.bfloat16, device=dev)
        L.check(lib.dsr_conv_pack_weight(C.byref(d), wt.data_ptr(), wf.data_ptr(), wd.data_ptr(), st))
        y = torch.empty(n, oh, ow, r8(cout), dtype=torch.bfloat16, device=dev)
        rows = lib.dsr_conv_stats_rows(C.byref(d))
        part = torch.empty((rows + 64) * 2 * r8(cout), dtype=torch.float32, device=dev)
        nostats = os.environ.get("MB_NOSTATS", "0") == "1"      # VGG-style launches: no BatchNorm statistics epilogue
        ep = L.Epilogue(0, 0.0, None, bias.data_ptr(), part.data_ptr() if (cout > 16 and cin > 8 and not nostats) else None, 0, None)   # first layers carry no BatchNorm
        tf = timeit(lambda: L.check(lib.dsr_conv_fwd(C.byref(d), x.data_ptr(), wf.data_ptr(), C.byref(ep), y.data_ptr(), st)))
        dy = (torch.rand_like(y.float()) - 0.5).to(torch.bfloat16)
        dx = torch.empty_like(x)
        wsz = lib.dsr_conv_dgrad_workspace(C.byref(d))
        ws = torch.empty(max(wsz, 16), dtype=torch.uint8, device=dev)
        td = timeit(lambda: L.check(lib.dsr_conv_dgrad(C.byref(d), dy.data_ptr(), wd.data_ptr(), dx.data_ptr(), ws.data_ptr(), wsz, st)))
        dw = torch.empty_like(wt)
        wsz2 = lib.dsr_conv_wgrad_workspace(C.byref(d))
        ws2 = torch.empty(wsz2, dtype=torch.uint8, device=dev)
        tw = timeit(lambda: L.check(lib.dsr_conv_wgrad(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws2.data_ptr(), wsz2, st)))
        names = [lib.dsr_conv_kernel_name(C.byref(d), op, C.byref(ep) if op == 0 else None).decode().replace("conv_", "").replace("_kernel", "")
                 for op in (0, 1, 2)]
        print(f"{name:28s} {fl/1e9:8.1f} | {tf*1e3:8.3f} {fl/tf/1e12:6.0f} | {td*1e3:8.3f} {fl/td/1e12:6.0f} | {tw*1e3:8.3f} {fl/tw/1e12:6.0f}"
              f" | {' / '.join(names)}", flush=True)


if __name__ == "__main__":
    main()
