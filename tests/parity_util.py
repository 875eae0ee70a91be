"""Shared gradient-parity rule of the GPU model tests.

These networks stack train-mode BatchNorms over small populations, which amplify 16-bit storage rounding far beyond one
ulp, so "how close must a bf16-storage gradient be to the fp32 oracle's?" is answered against a measured floor: the
oracle itself re-run with bf16 (or fp16) conv storage (oracle/lowp.py) -- the product's storage policy restated with the
oracle's own arithmetic.  Rule per gradient tensor (HIP vs fp32 oracle, floor = storage-model oracle vs fp32 oracle):

    1 - cos_hip <= F * (1 - cos_floor) + 0.01        and        cos_hip >= 0.93
        F = 1.5 for tensors of >= 4096 elements, 2.0 for smaller ones
    |norm ratio - 1| <= 0.03 for tensors of >= 4096 elements, 0.15 for smaller ones
    (the small ones are 64 ... 512-element BatchNorm / bias vectors: both the HIP deviation and the floor are then single draws
    of a noisy quantity -- measured on the HIP path up to 1 - cos = 0.037 where the storage-model oracle's draw is 0.017, and
    norm ratios up to 0.116 where it shows 0.09; different summation orders of the same kernels move these by +-0.01)

Not compared: conv biases in front of a train-mode BatchNorm (analytically zero gradient; the reference holds ~1e-9
rounding noise there) and one-element PReLU slopes (near-cancelling sums of +/- terms over a whole
activation map: the storage-model oracle itself flips the sign of some) -- those get an absolute bound tied to the
largest slope gradient, see prelu_ok().
"""


def cos(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-300))


def pre_bn_bias(k):
    return k.endswith("bias") and (("residual_blocks" in k and ".conv" in k) or k == "conv2.bias" or
                                   ("convblocks" in k and ".conv1." in k))


def grad_table(named_grads, ref_grads):
    """{key: (cosine, norm ratio)} against the oracle for every comparable tensor."""
    out = {}
    for k, g in named_grads:
        r = ref_grads.get(k)
        if g is None or r is None or pre_bn_bias(k) or r.numel() == 1 or float(r.abs().max()) == 0.0:
            continue
        out[k] = (cos(g.cpu(), r), float(g.double().norm().cpu() / r.double().norm()))
    return out


def compare_grads(hip, ref, sim, tag=""):
    """hip / ref / sim: {key: gradient}.  Returns (bad, table): violations of the rule above and, per tensor,
    (cos_hip, cos_floor, ratio_hip, ratio_floor, numel)."""
    th, ts = grad_table(hip.items(), ref), grad_table(sim.items(), ref)
    bad, table = [], {}
    for k, (c, r) in th.items():
        cf, rf = ts[k]
        n = ref[k].numel()
        table[tag + k] = (round(c, 4), round(cf, 4), round(r, 4), round(rf, 4), n)
        big = n >= 4096
        if (1 - c) > (1.5 if big else 2.0) * (1 - cf) + 0.01 or c < 0.93 or abs(r - 1) > (0.03 if big else 0.15):
            bad.append((tag + k,) + table[tag + k])
    return bad, table


def prelu_ok(hip, ref, sim):
    """One-element PReLU slope gradients: |hip - ref| <= 2 |sim - ref| + 5 % of the largest slope gradient."""
    ks = [k for k in ref if k.endswith("prelu1.weight")]
    scal = max(abs(float(ref[k])) for k in ks)
    return [(k, float(hip[k]), float(ref[k]), float(sim[k])) for k in ks
            if abs(float(hip[k]) - float(ref[k])) > 2.0 * abs(float(sim[k]) - float(ref[k])) + 0.05 * scal]
