#!/usr/bin/env python3
"""Timeline of the two-stream config-3 step (steps.TRACE marks, ms from the start of the step; GPU box only; development aid)."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

steps = importlib.import_module("deep-super-resolution_amd.steps")


def main():
    dev = torch.device("cuda:0")
    os.environ["DSR_GAN_GRAPH"] = "0"
    step, _ = bench.build_step("gan_x4", dev, 1)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    for rep in range(2):
        steps.TRACE = []
        step()
        torch.cuda.synchronize()
        tr = steps.TRACE
        steps.TRACE = None
        t0 = tr[0][1]
        print(f"--- step {rep}")
        for label, ev in tr:
            print(f"{t0.elapsed_time(ev):8.2f} ms  {label}")


if __name__ == "__main__":
    main()
