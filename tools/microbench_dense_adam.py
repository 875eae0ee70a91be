#!/usr/bin/env python3
"""Timing of the config-3 dense1 update (O = 1024, K = 512*32*32, 64 samples) through the C ABI (GPU box only; development aid):
dsr_linear_wgrad + dsr_pw_adam against the fused dsr_linear_wgrad_adam, with the HBM bytes each form moves."""
import ctypes as C
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = importlib.import_module("deep-super-resolution_amd._lib")
lib = L.lib()


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
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    o, k, bp = 1024, 512 * 32 * 32, 64
    dyt = (torch.rand(o, bp, device=dev) - 0.5).to(torch.bfloat16)
    xt = (torch.rand(k, bp, device=dev) - 0.5).to(torch.bfloat16)
    p = torch.rand(o, k, device=dev)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    sh = torch.empty(o, k, dtype=torch.bfloat16, device=dev)
    step = torch.ones(1, dtype=torch.int32, device=dev)
    dw = torch.empty(o, k, device=dev)
    n = o * k
    t_w = timeit(lambda: L.check(lib.dsr_linear_wgrad(L.BF16, dyt.data_ptr(), xt.data_ptr(), dw.data_ptr(), bp, o, k, st)))
    t_a = timeit(lambda: L.check(lib.dsr_pw_adam(p.data_ptr(), dw.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-4, 0.9, 0.999,
                                                 1e-8, step.data_ptr(), 1.0, sh.data_ptr(), st)))
    print(f"dsr_linear_wgrad      {t_w*1e3:7.3f} ms  {4*n/t_w/1e12:5.2f} TB/s (4 B/param written)")
    print(f"dsr_pw_adam           {t_a*1e3:7.3f} ms  {30*n/t_a/1e12:5.2f} TB/s (16 read + 14 written B/param)")
    for kpb in os.environ.get("KPB", "1,2,4,8").split(","):
        os.environ["DSR_WGRAD_ADAM_KPB"] = kpb
        t_f = timeit(lambda: L.check(lib.dsr_linear_wgrad_adam(L.BF16, dyt.data_ptr(), xt.data_ptr(), bp, o, k, 1, 1.0, p.data_ptr(),
                                                               m.data_ptr(), v.data_ptr(), sh.data_ptr(), step.data_ptr(), 1e-4, 0.9,
                                                               0.999, 1e-8, 1.0, st)))
        print(f"dsr_linear_wgrad_adam {t_f*1e3:7.3f} ms  {26*n/t_f/1e12:5.2f} TB/s (12 read + 14 written B/param)  kpb={kpb}   "
              f"two launches / fused = {(t_w+t_a)/t_f:.2f}")


if __name__ == "__main__":
    main()
