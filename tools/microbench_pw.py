#!/usr/bin/env python3
"""Effective HBM rate of the pointwise BatchNorm / activation kernels through the C ABI at config-3 tensor sizes
(GPU box only; development aid).   python tools/microbench_pw.py
Bytes counted: every operand tensor once (algorithmic), i.e. 2-3 "units" of P x Cp x 2 bytes per launch."""
import ctypes as C
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "deep-super-resolution_amd"
L = importlib.import_module(PKG + "._lib")
lib = L.lib()

SHAPES = [("G.trunk 32x128x128x64", 32 * 128 * 128, 64), ("D.b0 32x256x256x64", 32 * 256 * 256, 64),
          ("D.b1 32x256x256x128", 32 * 256 * 256, 128), ("D.b3 32x128x128x256", 32 * 128 * 128, 256),
          ("D.b5 32x64x64x512", 32 * 64 * 64, 512)]


def timeit(fn, reps=10):
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
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    print(f"{'tensor':26s} | {'bn_act_fwd':>12s} {'+res':>8s} | {'bwd_reduce':>10s} | {'bwd_apply':>9s} | {'act_bwd':>8s}   (TB/s algorithmic)")
    for name, p, c in SHAPES:
        y = (torch.rand(p, c, device=dev) - 0.5).to(torch.bfloat16)
        res = (torch.rand(p, c, device=dev) - 0.5).to(torch.bfloat16)
        dout = (torch.rand(p, c, device=dev) - 0.5).to(torch.bfloat16)
        out = torch.empty_like(y)
        f32 = lambda: torch.rand(c, device=dev) + 0.5
        scale, shift, mean, rstd, c1, c2 = f32(), f32(), f32(), f32(), f32() * 1e-3, f32() * 1e-3
        unit = p * c * 2
        t_f = timeit(lambda: L.check(lib.dsr_pw_bn_act_fwd(0, ptr(y), ptr(scale), ptr(shift), None, ptr(out), p, c, 1, 0.2, None, st)))
        t_fr = timeit(lambda: L.check(lib.dsr_pw_bn_act_fwd(0, ptr(y), ptr(scale), ptr(shift), ptr(res), ptr(out), p, c, 1, 0.2, None, st)))
        rpb = C.c_int()
        blocks = lib.dsr_pw_reduce_blocks(p, C.byref(rpb))
        part = torch.empty((blocks + 64) * 3 * c, dtype=torch.float32, device=dev)
        t_r = timeit(lambda: L.check(lib.dsr_pw_bn_act_bwd_reduce(0, ptr(dout), ptr(y), ptr(scale), ptr(shift), ptr(mean), ptr(rstd),
                                                                 p, c, blocks, rpb.value, 1, 0.2, None, ptr(part), st)))
        t_a = timeit(lambda: L.check(lib.dsr_pw_bn_act_bwd_apply(0, ptr(dout), ptr(y), ptr(scale), ptr(shift), ptr(mean), ptr(rstd),
                                                                ptr(c1), ptr(c2), ptr(out), p, c, 1, 0.2, None, 1, st)))
        n = 32
        hw = p // n
        h = int(hw ** 0.5)
        part2 = torch.empty((blocks + 64) * 2 * c, dtype=torch.float32, device=dev)
        t_b = timeit(lambda: L.check(lib.dsr_pw_act_bwd(0, ptr(dout), ptr(y), ptr(out), n, h, h, c, c, 0, 1, 0.2, None, blocks, rpb.value,
                                                       ptr(part2), st)))
        print(f"{name:26s} | {2*unit/t_f/1e12:12.2f} {3*unit/t_fr/1e12:8.2f} | {2*unit/t_r/1e12:10.2f} | {3*unit/t_a/1e12:9.2f} | {3*unit/t_b/1e12:8.2f}",
              flush=True)


if __name__ == "__main__":
    main()
