#!/usr/bin/env python3
"""dsr_conv_dgrad_ps (9x9 tail input gradient + the PixelShuffle-PReLU backward in its epilogue, one launch) against the two
launches it replaces (dsr_conv_dgrad, dsr_pw_act_bwd with pixshuf = 1) at config 3 (32 x 512 x 512) (GPU box only; development aid)."""
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
    n, H, W = 32, 512, 512
    d = L.ConvDesc(L.BF16, n, H, W, 64, 3, 9, 9, 1, 4, 0)
    wt = (torch.rand(3, 64, 9, 9, device=dev) - 0.5) * 0.05
    wf = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 0), dtype=torch.bfloat16, device=dev)
    wd = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 1), dtype=torch.bfloat16, device=dev)
    L.check(lib.dsr_conv_pack_weight(C.byref(d), wt.data_ptr(), wf.data_ptr(), wd.data_ptr(), st))
    dy3 = torch.zeros(n, H, W, 8, dtype=torch.bfloat16, device=dev)
    dy3[..., :3] = (torch.rand(n, H, W, 3, device=dev) - 0.5).to(torch.bfloat16)
    out = torch.randn(n, H, W, 64, device=dev).to(torch.bfloat16)
    dA = torch.empty(n, H, W, 64, dtype=torch.bfloat16, device=dev)
    dyu = torch.empty(n, H // 2, W // 2, 256, dtype=torch.bfloat16, device=dev)
    prelu = torch.full((1,), 0.25, device=dev)
    wsz = lib.dsr_conv_dgrad_workspace(C.byref(d))
    ws = torch.empty(max(wsz, 16), dtype=torch.uint8, device=dev)
    scr = lib.dsr_pw_scratch_rows()
    p = n * (H // 2) * (W // 2)
    blocks, rpb = F._reduce_blocks(p)
    part = torch.empty((blocks + scr) * 2 * 256, dtype=torch.float32, device=dev)
    rows = lib.dsr_conv_dgrad_ps_rows(C.byref(d))
    part2 = torch.empty((rows + scr) * 2 * 256, dtype=torch.float32, device=dev)

    def dgrad():
        L.check(lib.dsr_conv_dgrad(C.byref(d), dy3.data_ptr(), wd.data_ptr(), dA.data_ptr(), ws.data_ptr(), wsz, st))

    def act_bwd():
        L.check(lib.dsr_pw_act_bwd(L.BF16, dA.data_ptr(), out.data_ptr(), dyu.data_ptr(), n, H // 2, W // 2, 256, 64, 1, F.ACT_PRELU, 0.0,
                                   prelu.data_ptr(), blocks, rpb, part.data_ptr(), st))

    def both():
        dgrad()
        act_bwd()

    def fused():
        L.check(lib.dsr_conv_dgrad_ps(C.byref(d), dy3.data_ptr(), wd.data_ptr(), out.data_ptr(), prelu.data_ptr(), dyu.data_ptr(),
                                      part2.data_ptr(), st))
    print(f"tail input gradient + PixelShuffle-PReLU backward, batch {n}, {H}x{W}:")
    for name, fn in (("one launch  (dsr_conv_dgrad_ps)", fused), ("dsr_conv_dgrad (9x9 tail)", dgrad), ("dsr_pw_act_bwd (pixel shuffle)", act_bwd),
                     ("two launches", both)):
        print(f"  {name:36s} {timeit(fn):7.3f} ms", flush=True)


if __name__ == "__main__":
    main()
