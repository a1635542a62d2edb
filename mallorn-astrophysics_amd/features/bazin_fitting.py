"""Mirror of ``src/features/bazin_fitting.py`` backed by the HIP TRF kernel."""
from typing import List, Optional

from ._frame import run_extractor

LSST_BANDS = ["u", "g", "r", "i", "z", "y"]


def extract_bazin_features(lightcurves, object_ids: Optional[List[str]] = None):
    """bazin_fitting.py:254-288: 52 Bazin columns per object, ``object_id`` last."""
    return run_extractor("bazin", lightcurves, object_ids, id_last=True)
