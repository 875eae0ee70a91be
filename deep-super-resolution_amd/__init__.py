"""deep-super-resolution_amd: MI355X-native hot path of LewisClifton/Deep-Super-Resolution.

The directory name is not a Python identifier; import it with
    importlib.import_module("deep-super-resolution_amd")
(``from __graft_entry__ import pkg`` does that), or call ``install_dropin()`` to make the
reference's own import lines (``from models.GAN.generator import Generator`` ...) resolve here.
"""
import importlib
import sys
import types

__all__ = ["install_dropin", "load"]

# reference module path -> the mirror in this package.  Only LEAF modules are aliased: the reference's own `utils` and
# `models` packages keep serving everything this package does not mirror (utils.common), so `from utils.common import *`
# (train_GAN.py:15, eval_GAN.py:14, DIP.py:15) keeps working.  `dataset` is a top-level module of the reference.
MIRRORS = ("models.GAN.generator", "models.GAN.discriminator", "models.DIP", "models.DIP.skip", "models.DIP.utils",
           "utils.downsampler", "utils.GAN", "utils.DIP", "utils.degradation")
TOP_LEVEL_MIRRORS = ("dataset",)


def load(sub):
    """importlib.import_module of a submodule of this package, e.g. load('models.GAN.generator')."""
    return importlib.import_module(__name__ + "." + sub)


def _parent(name):
    """The reference's own package `name` when it is importable (its directory is on sys.path), else an empty
    stand-in package, so that the aliased leaves below are importable either way."""
    if name in MIRRORS:                  # models.DIP is a mirrored package: its submodules hang off the mirror
        return load(name)
    mod = sys.modules.get(name)
    if mod is not None and not (getattr(mod, "__name__", "") or "").startswith(__name__):
        return mod
    sys.modules.pop(name, None)          # (an alias of this package's own parent left by an older install_dropin)
    try:
        return importlib.import_module(name)
    except ImportError:
        mod = types.ModuleType(name)
        mod.__path__ = []                # a package with nothing of its own in it
        sys.modules[name] = mod
        if "." in name:
            setattr(_parent(name.rsplit(".", 1)[0]), name.rsplit(".", 1)[1], mod)
        return mod


def install_dropin():
    """Alias this package's mirrors under the reference's module paths so that the import lines of train_GAN.py /
    DIP.py / eval_GAN.py pick up the HIP-backed classes unchanged.  Returns the list of aliased names.

    With the reference's directory on sys.path its `utils` / `models` packages stay the reference's own: only the leaf
    modules listed in MIRRORS are replaced, and `utils.DIP` re-exports `utils.common` like the reference's does
    (utils/DIP.py:3 `from .common import *`)."""
    done = []
    for name in MIRRORS:
        ours = load(name)
        pkg_name, leaf = name.rsplit(".", 1)
        parent = _parent(pkg_name)
        sys.modules[name] = ours
        setattr(parent, leaf, ours)      # `import utils.GAN as g` / `from utils import GAN` read the attribute
        done.append(name)
    for name in TOP_LEVEL_MIRRORS:       # dataset.py (its own version needs torchvision, which is absent here)
        sys.modules[name] = load(name)
        done.append(name)
    try:                                 # utils/DIP.py:3 -- the reference's helpers ride along when they exist
        common = importlib.import_module("utils.common")
    except ImportError:
        common = None
    if common is not None:               # (utils/degradation.py:2 does the same `from utils.common import *`)
        public = getattr(common, "__all__", [k for k in vars(common) if not k.startswith("_")])
        for mod in (sys.modules["utils.DIP"], sys.modules["utils.degradation"]):
            for k in public:
                if not hasattr(mod, k):
                    setattr(mod, k, getattr(common, k))
    return done
