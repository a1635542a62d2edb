"""Mirror of ``src/features/multiband_gp.py`` backed by the HIP GP kernel (george is not needed)."""
from ._frame import run_extractor

LSST_BANDS = ["u", "g", "r", "i", "z", "y"]
BAND_WAVELENGTHS = {"u": 3670, "g": 4825, "r": 6222, "i": 7545, "z": 8691, "y": 9710}


def extract_multiband_gp_features(lightcurves, metadata=None, object_ids=None, verbose=True):
    """multiband_gp.py:347-385: 27 columns per object, ``object_id`` last.  ``metadata`` is accepted
    for signature compatibility (the reference does not read it either)."""
    return run_extractor("gp2d", lightcurves, object_ids, id_last=True)
