#!/usr/bin/env python3
"""Timing of the data-side path on DIV2K-sized images (GPU box only; development aid): halving a 2040x1356 image with the
Pillow-exact bicubic kernels, Gaussian noise on the device, and cutting a batch-32 config-3 patch pair out of an HBM-resident bank."""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
D = importlib.import_module("deep-super-resolution_amd.utils.degradation")
DS = importlib.import_module("deep-super-resolution_amd.dataset")


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(0)
    hr = torch.from_numpy(rng.randint(0, 256, (1356, 2040, 3), dtype=np.uint8)).to(dev)
    t = timeit(lambda: D.downsample(hr, 2))
    print(f"downsample 2040x1356 -> 1020x678 (two passes)   {t*1e6:8.1f} us   {(hr.numel() + hr.numel()//2 + hr.numel()//4)/t/1e9:6.1f} GB/s moved")
    half = D.downsample(hr, 2)
    t = timeit(lambda: D.add_gaussian_noise(half, 0.05, rng="device"))
    print(f"gaussian noise on 1020x678 (device rng)         {t*1e6:8.1f} us")
    # a bank of 100 pre-shrunk pairs (LR 170x255 after two halvings of the x8 file, HR 4x that), config-3 patches 128 -> 512
    pairs = []
    for i in range(100):
        lr = torch.from_numpy(rng.randint(0, 256, (170, 255, 3), dtype=np.uint8)).to(dev)
        pairs.append((lr, D.resize(lr, 255 * 4, 170 * 4)))
    bank = DS.PatchBank(pairs, 4, (128, 128), rng=np.random.RandomState(1))
    t = timeit(lambda: bank.sample(32))
    print(f"PatchBank.sample(32): 32 x (3x128x128, 3x512x512)  {t*1e6:8.1f} us per batch  ({t*1e3/40.2*100:5.2f} % of a 40.2 ms step)")


if __name__ == "__main__":
    main()
