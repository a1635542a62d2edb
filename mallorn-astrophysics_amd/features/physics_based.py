"""Mirror of ``src/features/physics_based.py`` backed by the HIP kernel."""
from ._frame import run_extractor


def extract_physics_features(lightcurves, metadata, object_ids=None):
    """physics_based.py:461-502: 32 columns per object (redshift ``Z`` from ``metadata``; ids
    missing from it, or NaN, count as z = 0), ``object_id`` last."""
    return run_extractor("physics", lightcurves, object_ids, metadata=metadata, id_last=True)
