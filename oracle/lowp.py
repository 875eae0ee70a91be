"""Oracle (test infrastructure): a "16-bit storage" model of the fp32 oracle.

The HIP path stores activations and activation-gradients in bf16 (GAN training) or fp16 (DIP, inference) and
accumulates in fp32.  Networks on this path are stacks of train-mode BatchNorms over small populations, which
amplify storage rounding far beyond one ulp, so "how far may the HIP result be from the fp32 oracle?" has no
fixed answer.  This context manager answers it per test: inside it, every convolution of the oracle rounds its
input, weight and output (and, in backward, the gradients flowing through those points) to the given 16-bit
type -- an idealised restatement of the product's storage policy with the oracle's own arithmetic.  Tests then
require   |hip - fp32 oracle|  <=  c * |16-bit-storage oracle - fp32 oracle| + eps.

What is modelled, exactly (pinned by tests/test_oracle_golden.py::test_lowp_storage_rounds_exactly_the_stated_points):
  * F.conv2d: input, weight and output rounded to `dtype` (round-to-nearest-even, the cast torch does); in backward the
    gradients arriving at those three points are rounded the same way; the bias and the accumulation stay fp32;
  * F.linear (the discriminator's dense head): input and weight rounded (the product feeds the dense head 16-bit activations
    and a 16-bit shadow of the fp32 weights); its OUTPUT stays fp32, as the product's does.
What is NOT modelled: the 16-bit stores after BatchNorm / activation passes (their outputs are rounded once more where the
next convolution reads them -- that rounding IS modelled, as the next conv's input rounding), BatchNorm statistics (fp32 on
both sides), Adam (fp32 on both sides).  The floor is therefore a LOWER bound of what an implementation with the product's
storage policy can reach, which is the direction the tests need.
"""
import contextlib

import torch
import torch.nn.functional as F


class _Round(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.dtype = dtype
        return x.to(dtype).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).to(g.dtype), None


@contextlib.contextmanager
def storage(dtype):
    orig_conv, orig_lin = F.conv2d, F.linear

    def conv(inp, w, b=None, *a, **kw):
        return _Round.apply(orig_conv(_Round.apply(inp, dtype), _Round.apply(w, dtype), b, *a, **kw), dtype)

    def linear(inp, w, b=None):
        return orig_lin(_Round.apply(inp, dtype), _Round.apply(w, dtype), b)

    F.conv2d, F.linear = conv, linear
    try:
        yield
    finally:
        F.conv2d, F.linear = orig_conv, orig_lin
