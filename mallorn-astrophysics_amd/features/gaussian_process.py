"""Mirror of ``src/features/gaussian_process.py`` (per-band scikit-learn GP) backed by the HIP kernel of
``csrc/gp1d.hpp``: scikit-learn is not needed at run time."""
from ._frame import run_extractor

LSST_BANDS = ["u", "g", "r", "i", "z", "y"]


def extract_gp_features(lightcurves, metadata=None, object_ids=None, verbose=True):
    """gaussian_process.py:251-289: 21 columns per object (length scale, amplitude, noise and log marginal
    likelihood of a C * RBF + White GP per band g, r, i, z; cross-band ratios and means), ``object_id``
    last.  ``metadata`` is accepted for signature compatibility (the reference does not read it)."""
    return run_extractor("gp1d", lightcurves, object_ids, id_last=True)
