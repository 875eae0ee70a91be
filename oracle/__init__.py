"""CPU oracle for the super-resolution hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain fp32 (torch CPU / numpy) restatement of the reference
algorithms on the hot path of LewisClifton/Deep-Super-Resolution
(SURVEY.md section 8a).  Every function cites the reference file:line it follows.

It is *not* part of the product: only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker
(or as the timed CPU baseline) -- never as a fallback of the HIP path.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference's own
modules (models/GAN, models/DIP, utils/downsampler, utils/DIP) in the build
container and stores their outputs as small ``.npz`` fixtures under ``tests/golden``;
``tests/test_oracle_golden.py`` checks this restatement against those fixtures.
Pieces of the path that live in third-party packages that are absent here
(torchvision's VGG19 weights + ``ImageClassification`` preset, torchmetrics PSNR) are
restated from their published definitions and marked "parity unpinned" in place.
"""
from . import filler, gan, dip, downsampler, losses, lowp, vgg, recipes  # noqa: F401
