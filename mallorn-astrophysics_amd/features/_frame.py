"""Shared DataFrame plumbing of the extractors: pack -> C-ABI -> frame."""
import numpy as np

from ..columns import COLUMNS
from ..engine import extract_csr
from ..packing import pack_lightcurves


# longest light curve (rows) a feature set's largest kernel tier takes; longer objects (bands, for the per-band GP)
# come back as NaN with status -100 where the reference would compute values (INTEGRATION.md "Limits")
SET_LIMITS = {"bazin": 2048, "powerlaw": 1024, "gp2d": 767, "gp1d": 767, "research": 1024}


def _extract_and_warn(set_name, csr, z, kept):
    """Run one set and warn -- with the count and the first ids -- about objects beyond its tier limit: their NaN
    rows are otherwise indistinguishable from "the fit failed" NaNs."""
    import warnings

    from .. import _lib
    lib = _lib.load()
    n = np.diff(csr["offsets"])
    if lib.lcfe_nstatus(1 << _lib_set_index(set_name)):
        out, status = extract_csr(set_name, csr, z=z, return_status=True)
        over = np.flatnonzero((status == -100).any(axis=1))
    else:
        out = extract_csr(set_name, csr, z=z)
        over = np.flatnonzero(n > lib.lcfe_max_points())
    if over.size:
        limit = SET_LIMITS.get(set_name, int(lib.lcfe_max_points()))
        what = "a band longer than 767 valid points or more than 767 rows" if set_name == "gp1d" else f"more than {limit} rows"
        warnings.warn(f"lcfe[{set_name}]: {over.size} object(s) with {what} are beyond the largest kernel tier and "
                      f"got NaN (status -100), e.g. {[kept[i] for i in over[:5]]}", RuntimeWarning, stacklevel=3)
    return out


def _lib_set_index(set_name):
    from ..columns import SET_NAMES
    return SET_NAMES.index(set_name)


def run_extractor(set_name, lightcurves, object_ids=None, metadata=None, id_last=True, int_columns=()):
    import pandas as pd

    csr, kept = pack_lightcurves(lightcurves, object_ids)
    z = None
    if metadata is not None:
        # physics_based.py:481,496: z_lookup.get(obj_id, nan)
        zmap = dict(zip(metadata["object_id"], metadata["Z"]))
        z = np.array([zmap.get(i, np.nan) for i in kept], dtype=np.float64)
    out = _extract_and_warn(set_name, csr, z, kept)
    df = pd.DataFrame(out, columns=COLUMNS[set_name])
    for c in int_columns:
        df[c] = df[c].astype(np.int64)
    if id_last:
        df["object_id"] = kept
    else:
        df.insert(0, "object_id", kept)
    return df
