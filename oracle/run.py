"""Batch driver of the oracle: CSR arrays in, [n_obj, ncols] float64 out."""
import numpy as np

from . import bazin, color, gp1d, gp2d, physics, powerlaw, research, shape, stat, tde
from .common import iter_objects

_MODS = {"stat": stat, "bazin": bazin, "powerlaw": powerlaw, "tde": tde, "color": color,
         "shape": shape, "physics": physics, "gp2d": gp2d,
         "gp1d": gp1d,           # per-band scikit-learn GP
         "research": research}   # v115 research features
NCOLS = {k: m.NCOL for k, m in _MODS.items()}


def extract(name, csr, z=None, lo=0, hi=None):
    """Run oracle feature set ``name`` over objects [lo, hi) of a CSR batch."""
    mod = _MODS[name]
    n_obj = len(csr["offsets"]) - 1
    hi = n_obj if hi is None else hi
    out = np.empty((hi - lo, mod.NCOL))
    for i, o in enumerate(iter_objects(csr, z)):
        if i < lo:
            continue
        if i >= hi:
            break
        out[i - lo] = mod.extract_one(o)
    return out
