"""Mirror of ``src/features/statistical.py`` backed by the HIP statistics kernel."""
from typing import List, Optional

from ..columns import STAT_INT_COLUMNS
from ._frame import run_extractor

LSST_BANDS = ["u", "g", "r", "i", "z", "y"]


def extract_statistical_features(lightcurves, object_ids: Optional[List[str]] = None, bands: List[str] = LSST_BANDS):
    """statistical.py:135-226: one row per object that has rows, 123 feature columns
    (``*_n_obs`` and ``peak_band`` int64), ``object_id`` first."""
    if list(bands) != LSST_BANDS:
        raise ValueError("the HIP statistics kernel is built for the six LSST bands u,g,r,i,z,y")
    return run_extractor("stat", lightcurves, object_ids, id_last=False, int_columns=STAT_INT_COLUMNS)


def add_metadata_features(features, metadata):
    """statistical.py:229-253 (host side: a left merge and two scalar columns)."""
    result = features.merge(metadata[["object_id", "Z", "EBV"]], on="object_id", how="left")
    result["luminosity_distance"] = result["Z"] * 4280
    result["time_dilation"] = 1 + result["Z"]
    return result
