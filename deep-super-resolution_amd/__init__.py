"""deep-super-resolution_amd: MI355X-native hot path of LewisClifton/Deep-Super-Resolution.

The directory name is not a Python identifier; import it with
    importlib.import_module("deep-super-resolution_amd")
(``from __graft_entry__ import pkg`` does that), or call ``install_dropin()`` to make the
reference's own import lines (``from models.GAN.generator import Generator`` ...) resolve here.
"""
import importlib
import sys

__all__ = ["install_dropin", "load"]


def load(sub):
    """importlib.import_module of a submodule of this package, e.g. load('models.GAN.generator')."""
    return importlib.import_module(__name__ + "." + sub)


def install_dropin():
    """Alias this package's mirrors under the reference's module paths so that train_GAN.py / DIP.py /
    eval_GAN.py import lines pick up the HIP-backed classes unchanged."""
    for name in ("models", "models.GAN", "models.GAN.generator", "models.GAN.discriminator", "models.DIP",
                 "utils", "utils.downsampler", "utils.GAN", "utils.DIP"):
        try:
            sys.modules[name] = load(name)
        except ModuleNotFoundError:
            pass
