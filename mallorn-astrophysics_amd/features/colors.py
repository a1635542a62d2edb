"""Mirror of ``src/features/colors.py`` backed by the HIP kernel."""
from ._frame import run_extractor


def extract_color_features(lightcurves, object_ids=None):
    """colors.py:347-380: 83 columns per object (``peak_mjd`` first), ``object_id`` last."""
    return run_extractor("color", lightcurves, object_ids, id_last=True)
