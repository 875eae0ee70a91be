#!/usr/bin/env python3
"""Timing of the discriminator's first two convolutions at config 3 (32 x 512 x 512): one fused launch (dsr_conv_first2_fwd, with
and without writing the first layer's activation) against the two launches it replaces (GPU box only; development aid)."""
import ctypes as C
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = importlib.import_module("deep-super-resolution_amd._lib")
lib = L.lib()


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n, h, w in ((32, 512, 512), (2, 512, 512), (4, 64, 64)):
        d0 = L.ConvDesc(L.BF16, n, h, w, 3, 64, 3, 3, 1, 1, 0)
        d1 = L.ConvDesc(L.BF16, n, h, w, 64, 64, 3, 3, 2, 1, 0)
        x = torch.zeros(n, h, w, 8, device=dev, dtype=torch.bfloat16)
        x[..., :3] = (torch.rand(n, h, w, 3, device=dev) - 0.5).to(torch.bfloat16)
        w0 = (torch.rand(64, 3, 3, 3, device=dev) - 0.5) * 0.5
        w1 = (torch.rand(64, 64, 3, 3, device=dev) - 0.5) * 0.1
        b0 = torch.zeros(64, device=dev)
        b1 = torch.zeros(64, device=dev)

        def pack(d, wt):
            wf = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 0), dtype=torch.bfloat16, device=dev)
            wd = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 1), dtype=torch.bfloat16, device=dev)
            L.check(lib.dsr_conv_pack_weight(C.byref(d), wt.data_ptr(), wf.data_ptr(), wd.data_ptr(), st))
            return wf
        wf0, wf1 = pack(d0, w0), pack(d1, w1)
        a0 = torch.empty(n, h, w, 64, device=dev, dtype=torch.bfloat16)
        y1 = torch.empty(n, h // 2, w // 2, 64, device=dev, dtype=torch.bfloat16)
        rows = max(lib.dsr_conv_stats_rows(C.byref(d1)), lib.dsr_conv_first2_stats_rows(C.byref(d0)))
        part = torch.empty((rows + 64) * 2 * 64, device=dev)
        ep0 = L.Epilogue(L.ACT_LEAKY, 0.2, None, b0.data_ptr(), None, 0, None)
        ep1 = L.Epilogue(L.ACT_NONE, 0.0, None, b1.data_ptr(), part.data_ptr(), 0, None)
        t_a = timeit(lambda: L.check(lib.dsr_conv_fwd(C.byref(d0), x.data_ptr(), wf0.data_ptr(), C.byref(ep0), a0.data_ptr(), st)))
        t_b = timeit(lambda: L.check(lib.dsr_conv_fwd(C.byref(d1), a0.data_ptr(), wf1.data_ptr(), C.byref(ep1), y1.data_ptr(), st)))
        t_keep = timeit(lambda: L.check(lib.dsr_conv_first2_fwd(C.byref(d0), C.byref(d1), x.data_ptr(), wf0.data_ptr(), b0.data_ptr(), 0.2,
                                                                wf1.data_ptr(), b1.data_ptr(), a0.data_ptr(), y1.data_ptr(), part.data_ptr(), st)))
        t_drop = timeit(lambda: L.check(lib.dsr_conv_first2_fwd(C.byref(d0), C.byref(d1), x.data_ptr(), wf0.data_ptr(), b0.data_ptr(), 0.2,
                                                                wf1.data_ptr(), b1.data_ptr(), None, y1.data_ptr(), part.data_ptr(), st)))
        print(f"{n} x {h} x {w}: first layer {t_a*1e3:7.1f} us + second layer {t_b*1e3:7.1f} us = {(t_a+t_b)*1e3:7.1f} us | fused, activation kept "
              f"{t_keep*1e3:7.1f} us | fused, not kept {t_drop*1e3:7.1f} us", flush=True)


def backward():
    """dsr_conv_dgrad_first_bwd (one launch, the 64-channel gradient between the layers never written) against the two launches
    it replaces: dsr_conv_dgrad of the stride-2 layer, then dsr_conv_first_bwd_recompute."""
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    n, h, w = 32, 512, 512
    g = torch.Generator(device="cpu").manual_seed(5)
    xg = torch.zeros(n, h, w, 8, dtype=torch.bfloat16, device=dev)
    xg[..., :3] = (torch.rand(n, h, w, 3, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
    w0 = ((torch.rand(64, 3, 3, 3, generator=g) - 0.5) * 0.6).to(dev)
    b0 = ((torch.rand(64, generator=g) - 0.5) * 0.2).to(dev)
    w1 = ((torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.1).to(dev)
    dy = (torch.rand(n, h // 2, w // 2, 64, generator=g) - 0.5).to(torch.bfloat16).to(dev)
    d0 = L.ConvDesc(L.BF16, n, h, w, 3, 64, 3, 3, 1, 1, 0)
    d1 = L.ConvDesc(L.BF16, n, h, w, 64, 64, 3, 3, 2, 1, 0)
    wf1 = torch.empty(lib.dsr_conv_packed_elems(C.byref(d1), 0), dtype=torch.bfloat16, device=dev)
    wd1 = torch.empty(lib.dsr_conv_packed_elems(C.byref(d1), 1), dtype=torch.bfloat16, device=dev)
    L.check(lib.dsr_conv_pack_weight(C.byref(d1), w1.data_ptr(), wf1.data_ptr(), wd1.data_ptr(), st))
    dw = torch.empty(64, 3, 3, 3, dtype=torch.float32, device=dev)
    db = torch.empty(64, dtype=torch.float32, device=dev)
    da0 = torch.empty(n, h, w, 64, dtype=torch.bfloat16, device=dev)
    wsz = lib.dsr_conv_dgrad_first_bwd_workspace(C.byref(d1))
    ws = torch.empty(wsz, dtype=torch.uint8, device=dev)
    wsz2 = lib.dsr_conv_dgrad_workspace(C.byref(d1))
    ws2 = torch.empty(max(wsz2, 16), dtype=torch.uint8, device=dev)
    wsz3 = lib.dsr_conv_first_bwd_workspace(C.byref(d0))
    ws3 = torch.empty(wsz3, dtype=torch.uint8, device=dev)

    def fused():
        L.check(lib.dsr_conv_dgrad_first_bwd(C.byref(d0), C.byref(d1), dy.data_ptr(), wd1.data_ptr(), xg.data_ptr(), w0.data_ptr(),
                                             b0.data_ptr(), L.ACT_LEAKY, 0.2, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), wsz, st))

    def dgrad():
        L.check(lib.dsr_conv_dgrad(C.byref(d1), dy.data_ptr(), wd1.data_ptr(), da0.data_ptr(), ws2.data_ptr(), wsz2, st))

    def first_bwd():
        L.check(lib.dsr_conv_first_bwd_recompute(C.byref(d0), xg.data_ptr(), da0.data_ptr(), w0.data_ptr(), b0.data_ptr(), L.ACT_LEAKY,
                                                 0.2, dw.data_ptr(), db.data_ptr(), ws3.data_ptr(), wsz3, st))

    def both():
        dgrad()
        first_bwd()

    print(f"backward of the first two layers, batch {n}, {h}x{w}:")
    for name, fn in (("one launch  (dsr_conv_dgrad_first_bwd)", fused), ("dsr_conv_dgrad (stride 2)", dgrad),
                     ("dsr_conv_first_bwd_recompute", first_bwd), ("two launches", both)):
        print(f"  {name:42s} {timeit(fn):7.3f} ms")


if __name__ == "__main__":
    if "backward" in sys.argv[1:]:
        backward()
    else:
        main()
