"""Shared DataFrame plumbing of the extractors: pack -> C-ABI -> frames.

``extract_all`` is the engine-speed form of the drop-in boundary: ONE ``pack_lightcurves`` of the long frame and ONE
``lcfe_extract(mask)`` for every requested feature set (the sets then run side by side on the engine's streams and
the CSR batch crosses PCIe once), cut into the per-extractor frames the reference's functions return.  The
``extract_*_features`` mirrors are one-set calls of it.  ``ngpu > 1`` (or ``LCFE_NGPU``) shards the batch over the
GPUs of the node through ``dist.extract_multi_gpu`` (one child process per GPU, started before any GPU call of theirs).
"""
import os
import warnings

import numpy as np

from ..columns import COLUMNS, SET_NAMES, STAT_INT_COLUMNS
from ..engine import extract_csr, mask_of, sets_of
from ..packing import pack_lightcurves

# frame conventions of the reference's extractors: where ``object_id`` sits and which columns are int64
#   statistical.py:214-226 (id first, *_n_obs / peak_band int64); train_v55_powerlaw.py:196-202 (dict with the id first);
#   every other extractor appends the id last (colors.py:377, bazin_fitting.py:283, multiband_gp.py:381, ...)
ID_FIRST = {"stat", "powerlaw"}
INT_COLUMNS = {"stat": STAT_INT_COLUMNS}
# sets that read the redshift column of the metadata (physics_based.py:481, research_features.py:552-559)
NEEDS_Z = {"physics", "research"}


def _limit_message(set_name, csr, rows, lib):
    """What the objects `rows` of a set ran into (status -100): the message names the limit that was hit, so that a
    limit-NaN can be told from a failed fit.  The LDS tiers end at 2048 rows (1024 for the object-level fits and the
    research set, 767 for the 2-D GP); longer light curves run in the long-object tier (working set in global scratch),
    whose limits are the ones reported here."""
    n = np.diff(csr["offsets"])[rows]
    max_rows = int(lib.lcfe_max_points())
    over_rows = int((n > max_rows).sum())
    parts = [f"{over_rows} with more than {max_rows} rows"] if over_rows else []
    rest = len(rows) - over_rows
    if rest:
        if set_name == "gp2d":
            parts.append(f"{rest} with more than {int(lib.lcfe_gp2d_max_points())} valid points")
        elif set_name == "gp1d":
            parts.append(f"{rest} with a band of more than 767 valid points or more than 767 rows")
        elif set_name == "research":
            parts.append(f"{rest} whose r band spans more than 65536 days (the Mexican-hat grid of the long-object tier)")
        else:
            parts.append(f"{rest} beyond a tier limit")
    return ", ".join(parts)


def _warn_limits(names, csr, kept, status, lib):
    """RuntimeWarning -- with the count, the cause and the first ids -- for the objects a set returned NaN for because
    they are beyond a kernel limit: such rows are otherwise indistinguishable from "the fit failed" NaNs."""
    n = np.diff(csr["offsets"])
    st0 = 0
    for name in names:
        nst = int(lib.lcfe_nstatus(1 << SET_NAMES.index(name)))
        if nst:
            over = np.flatnonzero((status[:, st0:st0 + nst] == -100).any(axis=1))
            st0 += nst
        else:
            over = np.flatnonzero(n > lib.lcfe_max_points())
        if over.size:
            warnings.warn(f"lcfe[{name}]: {over.size} object(s) beyond a kernel limit got NaN (status -100): "
                          f"{_limit_message(name, csr, over, lib)}; e.g. {[kept[i] for i in over[:5]]}",
                          RuntimeWarning, stacklevel=4)


def frame_of(set_name, block, kept):
    """The DataFrame an extractor of the reference returns for `set_name` from its block of the feature matrix."""
    import pandas as pd

    df = pd.DataFrame(block, columns=COLUMNS[set_name])
    for c in INT_COLUMNS.get(set_name, ()):
        df[c] = df[c].astype(np.int64)
    if set_name in ID_FIRST:
        df.insert(0, "object_id", kept)
    else:
        df["object_id"] = kept
    return df


def redshifts(metadata, kept):
    """z per kept object: ``z_lookup.get(obj_id, nan)`` (physics_based.py:481,496)."""
    zmap = dict(zip(metadata["object_id"], metadata["Z"]))
    return np.array([zmap.get(i, np.nan) for i in kept], dtype=np.float64)


def default_ngpu():
    try:
        return max(1, int(os.environ.get("LCFE_NGPU", "1")))
    except ValueError:
        return 1


def extract_all(lightcurves=None, metadata=None, object_ids=None, sets=None, csr=None, ngpu=None, return_matrix=False):
    """Every requested feature set from ONE pack and ONE engine call.

    ``lightcurves``: the long frame (``object_id, Time (MJD), Flux, Flux_err, Filter``), or pass ``csr=(csr, ids)``
    as ``utils.data_loader.load_lightcurves_csr`` / ``packing.pack_lightcurves`` return it (``object_ids`` then selects
    and orders objects of that batch).  ``sets``: names (default: all ten).  Returns ``{set name: DataFrame}`` with the
    reference's conventions per extractor; with ``return_matrix`` also the raw ``(matrix, status, kept_ids)``."""
    from .. import _lib
    from ..packing import select_objects

    lib = _lib.load()
    names = sets_of(mask_of(list(SET_NAMES) if sets is None else sets))         # canonical (column) order
    mask = mask_of(names)
    if csr is not None:
        batch, ids = csr
        batch, kept = (batch, list(ids)) if object_ids is None else select_objects(batch, ids, object_ids)
    else:
        batch, kept = pack_lightcurves(lightcurves, object_ids)
    z = None
    if metadata is not None and (NEEDS_Z & set(names)):
        z = redshifts(metadata, kept)
    ngpu = default_ngpu() if ngpu is None else int(ngpu)
    if ngpu > 1:
        from ..dist import extract_multi_gpu
        out, status = extract_multi_gpu(mask, batch, z, ngpu)
    else:
        res = extract_csr(mask, batch, z=z, return_status=True)
        out, status = res
    _warn_limits(names, batch, kept, status, lib)
    frames, col0 = {}, 0
    for name in names:
        ncol = len(COLUMNS[name])
        frames[name] = frame_of(name, out[:, col0:col0 + ncol], kept)
        col0 += ncol
    if return_matrix:
        return frames, (out, status, kept)
    return frames


def run_extractor(set_name, lightcurves, object_ids=None, metadata=None, id_last=True, int_columns=()):
    """One extractor of the reference = a one-set call of ``extract_all`` (``id_last`` / ``int_columns`` are implied by
    the set and kept in the signature for the mirrors that spell them out)."""
    return extract_all(lightcurves, metadata=metadata, object_ids=object_ids, sets=[set_name])[set_name]
