"""Batched form of the v55 decline-model features.

The reference computes them per object inside its scripts
(``scripts/train_v55_powerlaw.py:147-202`` / ``scripts/visualize_and_powerlaw.py:148-255``,
``extract_powerlaw_features(obj_id, lc_data)``); here all objects go to the device in one call.
"""
from ._frame import run_extractor


def extract_powerlaw_features(lightcurves, object_ids=None):
    """27 R^2 columns ``{g,r,i}_{model}_r2``; ``object_id`` first (as the reference's dict)."""
    return run_extractor("powerlaw", lightcurves, object_ids, id_last=False)
