"""Deterministic closed-form parameter / input filler (test infrastructure).

Golden fixtures only store *outputs*; the parameters and inputs that produced them
are regenerated from (key name, shape) by this exact-integer hash, so fixtures stay
KB-sized and the GPU box (which has no /root/reference) can rebuild the very same
state_dicts.  splitmix64 on uint64 is bit-exact on every platform.
"""
import zlib

import numpy as np
import torch

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)


def hash_uniform(n, seed):
    """n float64 values in [-1, 1), a pure function of (n, seed)."""
    with np.errstate(over="ignore"):
        x = np.arange(1, n + 1, dtype=np.uint64) * _G + np.uint64(seed) * _M2
        x ^= x >> np.uint64(30)
        x *= _M1
        x ^= x >> np.uint64(27)
        x *= _M2
        x ^= x >> np.uint64(31)
    return (x >> np.uint64(11)).astype(np.float64) / float(1 << 53) * 2.0 - 1.0


def _seed_of(name, salt=0):
    return (zlib.crc32(name.encode()) + 7919 * salt) & 0x7FFFFFFF


def tensor(name, shape, scale=1.0, offset=0.0, salt=0, dtype=torch.float32):
    n = int(np.prod(shape)) if len(shape) else 1
    v = hash_uniform(n, _seed_of(name, salt)) * scale + offset
    return torch.from_numpy(v.reshape(shape)).to(dtype)


def fill_state_dict(reference_sd, salt=0):
    """Return a new state_dict with the same keys/shapes/dtypes, filled closed-form.

    Rules by key suffix (shapes come from the module being filled):
      *.weight of rank>=2  : uniform(-a, a), a = sqrt(3 / fan_in)   (unit-gain-ish)
      *.weight of rank 1, size 1 (PReLU)      : 0.25 + 0.05*u
      *.weight of rank 1 (BatchNorm gamma)    : 1 + 0.2*u
      *.bias                                  : 0.1*u
      *.running_mean                          : 0.1*u
      *.running_var                           : 1 + 0.3*u   (>0)
      *.num_batches_tracked                   : 0
    """
    out = {}
    for k, v in reference_sd.items():
        shape = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros(shape, dtype=v.dtype)
        elif k.endswith("running_mean"):
            out[k] = tensor(k, shape, 0.1, 0.0, salt, v.dtype)
        elif k.endswith("running_var"):
            out[k] = tensor(k, shape, 0.3, 1.0, salt, v.dtype)
        elif k.endswith("bias"):
            out[k] = tensor(k, shape, 0.1, 0.0, salt, v.dtype)
        elif k.endswith("weight") and len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            out[k] = tensor(k, shape, float(np.sqrt(3.0 / fan_in)), 0.0, salt, v.dtype)
        elif k.endswith("weight") and shape == (1,):
            out[k] = tensor(k, shape, 0.05, 0.25, salt, v.dtype)
        elif k.endswith("weight"):
            out[k] = tensor(k, shape, 0.2, 1.0, salt, v.dtype)
        else:
            raise KeyError(f"filler: no rule for {k} {shape}")
    return out
