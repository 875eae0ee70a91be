#!/usr/bin/env python3
"""dsr_conv_dgrad_bn (input gradient of a stride-2 layer + the BatchNorm-backward sums of the layer in front, one launch)
against the two launches it replaces (dsr_conv_dgrad, dsr_pw_bn_act_bwd_reduce) at the discriminator's config-3 shapes
(GPU box only; development aid)."""
import ctypes as C
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = importlib.import_module("deep-super-resolution_amd._lib")
F = importlib.import_module("deep-super-resolution_amd.functional")
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
    n = 32
    for name, h, cin, cout in (("D.b2 (sums of b1)", 256, 128, 128), ("D.b4 (sums of b3)", 128, 256, 256), ("D.b6 (sums of b5)", 64, 512, 512)):
        w = h
        d = L.ConvDesc(L.BF16, n, h, w, cin, cout, 3, 3, 2, 1, 0)
        assert lib.dsr_conv_dgrad_bn_supported(C.byref(d)) == 1
        dy = (torch.rand(n, h // 2, w // 2, cout, device=dev) - 0.5).to(torch.bfloat16)
        y = torch.randn(n, h, w, cin, device=dev).to(torch.bfloat16)
        wt = (torch.rand(cout, cin, 3, 3, device=dev) - 0.5) * 0.1
        wf = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 0), dtype=torch.bfloat16, device=dev)
        wd = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 1), dtype=torch.bfloat16, device=dev)
        L.check(lib.dsr_conv_pack_weight(C.byref(d), wt.data_ptr(), wf.data_ptr(), wd.data_ptr(), st))
        sc, sh = torch.rand(cin, device=dev) + 0.5, torch.rand(cin, device=dev) - 0.5
        dx = torch.empty(n, h, w, cin, dtype=torch.bfloat16, device=dev)
        scr = lib.dsr_pw_scratch_rows()
        rows = lib.dsr_conv_dgrad_bn_rows(C.byref(d))
        part = torch.empty((rows + scr) * 3 * cin, dtype=torch.float32, device=dev)
        wsz = lib.dsr_conv_dgrad_workspace(C.byref(d))
        ws = torch.empty(max(wsz, 16), dtype=torch.uint8, device=dev)
        p = n * h * w
        blocks, rpb = F._bn_bwd_blocks(p, F.ACT_LEAKY)
        part2 = torch.empty((blocks + scr) * 3 * cin, dtype=torch.float32, device=dev)

        def fused():
            L.check(lib.dsr_conv_dgrad_bn(C.byref(d), dy.data_ptr(), wd.data_ptr(), dx.data_ptr(), y.data_ptr(), sc.data_ptr(),
                                          sh.data_ptr(), F.ACT_LEAKY, 0.2, part.data_ptr(), st))

        def dgrad():
            L.check(lib.dsr_conv_dgrad(C.byref(d), dy.data_ptr(), wd.data_ptr(), dx.data_ptr(), ws.data_ptr(), wsz, st))

        def reduce():
            L.check(lib.dsr_pw_bn_act_bwd_reduce(L.BF16, dx.data_ptr(), y.data_ptr(), sc.data_ptr(), sh.data_ptr(), sc.data_ptr(),
                                                 sh.data_ptr(), p, cin, blocks, rpb, F.ACT_LEAKY, 0.2, None, part2.data_ptr(), st))

        def both():
            dgrad()
            reduce()
        tf, td, tr, tb = timeit(fused), timeit(dgrad), timeit(reduce), timeit(both)
        print(f"{name:20s} {cin:3d}->{cout:3d} @{h}: one launch {tf:6.3f} ms | dgrad {td:6.3f} + reduce {tr:6.3f} = {tb:6.3f} ms (back to back)", flush=True)


if __name__ == "__main__":
    main()
