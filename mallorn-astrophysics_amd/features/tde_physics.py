"""Mirror of ``src/features/tde_physics.py`` backed by the HIP kernel."""
from ._frame import run_extractor


def extract_tde_physics_features(lightcurves, object_ids=None):
    """tde_physics.py:377-411: 25 columns per object, ``object_id`` last."""
    return run_extractor("tde", lightcurves, object_ids, id_last=True)
