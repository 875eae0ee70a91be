#!/usr/bin/env python3
"""Does an MFMA-bound convolution overlap with an HBM-bound stream kernel on a second HIP stream?  (GPU box, development aid)
Times N launches of a conv (C ABI) alone, M launches of the Adam kernel on a 268 M-element tensor alone, then both at once on
two streams; prints the three wall times.  Perfect overlap: both = max(conv, adam); none: both = conv + adam."""
import ctypes as C
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "deep-super-resolution_amd"
L = importlib.import_module(PKG + "._lib")
lib = L.lib()
dev = torch.device("cuda:0")


def conv_case(n, h, w, cin, cout, stride=1):
    d = L.ConvDesc(L.BF16, n, h, w, cin, cout, 3, 3, stride, 1, 0)
    oh, ow = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
    x = (torch.rand(n, h, w, cin, device=dev) - 0.5).to(torch.bfloat16)
    wt = (torch.rand(cout, cin, 3, 3, device=dev) - 0.5) * 0.1
    wf = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 0), dtype=torch.bfloat16, device=dev)
    wd = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 1), dtype=torch.bfloat16, device=dev)
    st0 = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.dsr_conv_pack_weight(C.byref(d), wt.data_ptr(), wf.data_ptr(), wd.data_ptr(), st0))
    y = torch.empty(n, oh, ow, cout, dtype=torch.bfloat16, device=dev)
    ep = L.Epilogue(3, 0.0, None, None, None, 0, None)
    fl = 2.0 * n * oh * ow * cout * 9 * cin
    return (lambda st: L.check(lib.dsr_conv_fwd(C.byref(d), x.data_ptr(), wf.data_ptr(), C.byref(ep), y.data_ptr(), st))), fl, (x, wf, y)


def main():
    nel = 268 * 1024 * 1024
    p, g, m, v = (torch.rand(nel, device=dev) for _ in range(4))
    step = torch.ones(1, dtype=torch.int32, device=dev)
    adam = lambda st: L.check(lib.dsr_pw_adam(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), nel, 1e-4, 0.9, 0.999, 1e-8,
                                              step.data_ptr(), 1.0, None, st))
    y_bn = (torch.rand(32 * 256 * 256, 64, device=dev) - 0.5).to(torch.bfloat16)
    o_bn = torch.empty_like(y_bn)
    sc = torch.rand(64, device=dev)
    bn = lambda st: L.check(lib.dsr_pw_bn_act_fwd(0, y_bn.data_ptr(), sc.data_ptr(), sc.data_ptr(), None, o_bn.data_ptr(),
                                                  32 * 256 * 256, 64, 1, 0.2, None, st))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    h1, h2 = C.c_void_p(s1.cuda_stream), C.c_void_p(s2.cuda_stream)

    def wall(fa, na, fb, nb):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(max(na, nb)):          # interleaved issue so that both queues stay fed
            if i < na:
                fa(h1)
            if i < nb:
                fb(h2)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    for name, args in (("V.256->256 @56", (32, 56, 56, 256, 256)), ("D.b5 256->512 @64", (32, 64, 64, 256, 512)),
                       ("G.trunk 64->64 @128 (c64)", (32, 128, 128, 64, 64)), ("D.b2 128->128 s2 @256", (32, 256, 256, 128, 128, 2))):
        conv, fl, keep = conv_case(*args)
        for hb_name, hb, nb in (("adam 268M", adam, 6), ("bn_act_fwd 537MB", bn, 60)):
            nc = 80
            wall(conv, 5, hb, 2)
            tc = wall(conv, nc, hb, 0)
            th = wall(conv, 0, hb, nb)
            tb = wall(conv, nc, hb, nb)
            print(f"{name:28s} + {hb_name:18s}: conv {tc:7.2f} ms ({fl*nc/tc/1e9:6.0f} TF)  hbm {th:7.2f} ms  both {tb:7.2f} ms  "
                  f"(sum {tc+th:7.2f}, max {max(tc,th):7.2f}) overlap gain {100*(tc+th-tb)/(tc+th):5.1f} %", flush=True)


if __name__ == "__main__":
    main()
