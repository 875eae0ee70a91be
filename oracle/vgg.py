"""Oracle (test infrastructure): VGG19 feature trunk + preprocessing of the content loss.

Follows /root/reference/utils/GAN.py:
  :19-57   layer list of torchvision vgg19().features
  :72      cut at index 36  -> 16 conv3x3+ReLU, max-pools after conv 2, 4, 8, 12 (the 5th pool is cut)
  :77-78   frozen weights
  :82-83   VGG19_Weights.IMAGENET1K_V1.transforms() on *both* images
  :86-90   MSE of the two feature maps
torchvision is NOT installed here and its pretrained weights cannot be fetched:
  * the weights are deterministic stand-ins (oracle.filler) -- same architecture;
  * transforms() is restated from torchvision's published ImageClassification preset
    (resize shorter side to 256 with bilinear+antialias, centre-crop 224, normalise with
    mean (0.485,0.456,0.406) / std (0.229,0.224,0.225)).  PARITY UNPINNED for these two.
"""
import torch
import torch.nn.functional as F

# (cin, cout) per conv; 'M' = MaxPool2d(2,2).  features[:36] of vgg19.
CFG = [(3, 64), (64, 64), "M", (64, 128), (128, 128), "M",
       (128, 256), (256, 256), (256, 256), (256, 256), "M",
       (256, 512), (512, 512), (512, 512), (512, 512), "M",
       (512, 512), (512, 512), (512, 512), (512, 512)]
# index inside torchvision's features Sequential for each conv (for state_dict key names)
CONV_INDEX = [0, 2, 5, 7, 10, 12, 14, 16, 19, 21, 23, 25, 28, 30, 32, 34]
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def vgg_shapes():
    s = {}
    convs = [c for c in CFG if c != "M"]
    for idx, (ci, co) in zip(CONV_INDEX, convs):
        s[f"{idx}.weight"] = (co, ci, 3, 3)
        s[f"{idx}.bias"] = (co,)
    return s


def preprocess(img, resize=256, crop=224):
    """ImageClassification preset for a float NCHW batch."""
    n, c, h, w = img.shape
    if h <= w:
        nh, nw = resize, int(resize * w / h)
    else:
        nh, nw = int(resize * h / w), resize
    x = F.interpolate(img, size=(nh, nw), mode="bilinear", align_corners=False, antialias=True)
    top = int(round((nh - crop) / 2.0))
    left = int(round((nw - crop) / 2.0))
    x = x[:, :, top:top + crop, left:left + crop]
    mean = torch.tensor(MEAN, dtype=x.dtype)[None, :, None, None]
    std = torch.tensor(STD, dtype=x.dtype)[None, :, None, None]
    return (x - mean) / std


def features36(sd, x):
    it = iter(CONV_INDEX)
    for c in CFG:
        if c == "M":
            x = F.max_pool2d(x, 2, 2)
        else:
            i = next(it)
            x = F.relu(F.conv2d(x, sd[f"{i}.weight"], sd[f"{i}.bias"], padding=1))
    return x


def vgg_loss(sd, img1, img2, resize=256, crop=224):
    f1 = features36(sd, preprocess(img1, resize, crop))
    f2 = features36(sd, preprocess(img2, resize, crop))
    return ((f1 - f2) ** 2).mean()
