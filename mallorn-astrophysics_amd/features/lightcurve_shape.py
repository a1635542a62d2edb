"""Mirror of ``src/features/lightcurve_shape.py`` backed by the HIP kernel."""
from ._frame import run_extractor


def extract_shape_features(lightcurves, object_ids=None):
    """lightcurve_shape.py:335-368: 65 columns per object, ``object_id`` last."""
    return run_extractor("shape", lightcurves, object_ids, id_last=True)
