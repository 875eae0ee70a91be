"""Time the dense-head GEMMs (discriminator.py:41: 64 x 524288 x 1024 at the config-3 shape) through the C ABI."""
import ctypes as C
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

L = importlib.import_module("deep-super-resolution_amd._lib")


def main():
    lib = L.lib()
    dev = torch.device("cuda:0")
    b, k, o = 64, 524288, 1024
    ptr = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    x = (torch.randn(b, k, device=dev) * 0.1).bfloat16()
    w = (torch.randn(o, k, device=dev) * 0.01).bfloat16()
    dy = (torch.randn(b, o, device=dev) * 0.1).bfloat16()
    bias = torch.zeros(o, device=dev)
    wsz = lib.dsr_linear_fwd_workspace(b, k, o)
    ws = torch.empty(wsz, dtype=torch.uint8, device=dev)
    out = torch.empty((b, o), dtype=torch.float32, device=dev)
    dx = torch.empty((b, k), dtype=torch.bfloat16, device=dev)
    xt = x.t().contiguous()
    dyt = dy.t().contiguous()
    dw = torch.empty((o, k), dtype=torch.float32, device=dev)
    calls = {
        "fwd": (lambda: lib.dsr_linear_fwd(0, ptr(x), ptr(w), ptr(bias), 1, 0.2, ptr(out), b, k, o, ptr(ws), wsz, st), w.numel() * 2),
        "dgrad": (lambda: lib.dsr_linear_dgrad(0, ptr(dy), ptr(w), ptr(dx), b, o, k, st), w.numel() * 2),
        "wgrad": (lambda: lib.dsr_linear_wgrad(0, ptr(dyt), ptr(xt), ptr(dw), 64, o, k, st), w.numel() * 4),
    }
    for name, (fn, nbytes) in calls.items():
        for _ in range(2):
            L.check(fn())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"{name:6s} {ms * 1e3:8.1f} us  {nbytes / ms / 1e9:6.2f} TB/s (weight-side bytes only)", flush=True)
    ref = torch.nn.functional.leaky_relu(x.float() @ w.float().t(), 0.2)
    print("fwd max err", (out - ref).abs().max().item(), "ref max", ref.abs().max().item())
    print("dgrad rel", ((dx.float() - dy.float() @ w.float()).abs().max() / (dy.float() @ w.float()).abs().max()).item())
    print("wgrad max err", (dw - dy.float().t() @ x.float()).abs().max().item())


if __name__ == "__main__":
    main()
