"""Generator inference (eval_GAN.py:44,94: ``gan_G.eval()`` then ``gan_G(LR_image)``), whole-image or tiled.

In eval mode every BatchNorm is a fixed affine map, so the generator is a pure convolution stack with a finite
receptive field: 4 (9x9 head) + 2*blocks + 1 (3x3 trunk) LR pixels, plus 1, 1/2, 1/4 ... for the shuffle convs and
4/factor for the 9x9 tail -- 40 LR pixels for the 16-block x8 model.  A tile computed with that much
halo is therefore bit-identical to the same region of the whole-image result; tiling only bounds the activation
footprint (64 channels at 8x resolution) when images are large.  BASELINE config 5 runs this in fp16.
"""
import torch


def receptive_halo(gen):
    blocks = len(gen.residual_blocks)
    return 4 + 2 * blocks + 1 + 2 + 1        # head + trunk + conv2 + shuffle convs/tail (rounded up)


@torch.no_grad()
def super_resolve(gen, lr, tile=None, halo=None, dtype=torch.float16):
    """lr: fp32 NCHW [N,3,h,w] on the GPU -> fp32 [N,3,h*f,w*f].  tile=None runs the whole image at once."""
    was_training = gen.training
    old = gen.compute_dtype
    gen.eval()
    gen.compute_dtype = dtype
    try:
        if tile is None:
            return gen(lr)
        n, _, h, w = lr.shape
        halo = receptive_halo(gen) if halo is None else halo
        f = 2 ** len(gen.pixel_shuffle_blocks)
        out = torch.empty((n, 3, h * f, w * f), dtype=torch.float32, device=lr.device)
        for y0 in range(0, h, tile):
            for x0 in range(0, w, tile):
                y1, x1 = min(y0 + tile, h), min(x0 + tile, w)
                ya, xa = max(y0 - halo, 0), max(x0 - halo, 0)
                yb, xb = min(y1 + halo, h), min(x1 + halo, w)
                sr = gen(lr[:, :, ya:yb, xa:xb].contiguous())
                out[:, :, y0 * f:y1 * f, x0 * f:x1 * f] = sr[:, :, (y0 - ya) * f:(y1 - ya) * f, (x0 - xa) * f:(x1 - xa) * f]
        return out
    finally:
        gen.compute_dtype = old
        gen.train(was_training)
