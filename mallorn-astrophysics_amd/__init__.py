"""lcfe -- MI355X-native light-curve feature-extraction engine.

Drop-in for the per-object feature extractors of MALLORN-astrophysics ``src/features/``:
the ``extract_*_features(lightcurves, ...) -> DataFrame`` functions keep the reference's
signatures; underneath, light curves are CSR-packed and handed to hand-written HIP kernels
(gfx950) through the C-ABI declared in ``include/lcfe.h``.  There is no CPU fallback: the
extractors raise if ``liblcfe.so`` is missing or no GPU is present.
"""
__version__ = "0.1.0"
