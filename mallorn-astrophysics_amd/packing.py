"""DataFrame -> CSR packer: the host side of the drop-in boundary.

The reference groups the long light-curve frame with ``groupby('object_id')`` and loops over
ids in Python (``src/features/statistical.py:155-165``).  Here the same frame is turned once
into five flat arrays -- ``offsets`` int64[n_obj+1], ``t``/``flux``/``err`` float64[total],
``band`` uint8[total] -- with the rows of every object kept in *file order* (several features
depend on it: ``statistical.py:186-190``, ``lightcurve_shape.py:287-306``, ``colors.py:43``).
Objects that have no rows are dropped, as the reference ``continue``s over them
(``statistical.py:163-165``).
"""
from __future__ import annotations

import numpy as np

BANDS = "ugrizy"
UNKNOWN_BAND = 255

COL_ID, COL_T, COL_F, COL_E, COL_B = "object_id", "Time (MJD)", "Flux", "Flux_err", "Filter"


def band_codes(filters) -> np.ndarray:
    """Map Filter strings to 0..5 (u,g,r,i,z,y); anything else -> 255."""
    import pandas as pd

    cat = pd.Categorical(filters, categories=list(BANDS))
    codes = np.asarray(cat.codes)
    return np.where(codes < 0, UNKNOWN_BAND, codes).astype(np.uint8)


def pack_lightcurves(lightcurves, object_ids=None):
    """Return ``(csr, kept_ids)`` for the objects of ``object_ids`` that have >= 1 row.

    ``csr`` is a dict with keys offsets, t, flux, err, band.  ``kept_ids`` preserves the
    order of ``object_ids`` (default: order of first appearance, like ``Series.unique()``).
    """
    import pandas as pd

    ids_col = lightcurves[COL_ID]
    if object_ids is None:
        object_ids = ids_col.unique()
    object_ids = np.asarray(list(object_ids), dtype=object)
    # position of each requested id; rows of unrequested ids are dropped
    codes = pd.Categorical(ids_col, categories=pd.unique(object_ids)).codes
    uniq = pd.unique(object_ids)
    keep = codes >= 0
    rows = np.flatnonzero(keep)
    codes = codes[keep]
    order = np.argsort(codes, kind="stable")          # stable: file order inside each object
    rows = rows[order]
    counts_u = np.bincount(codes, minlength=len(uniq))
    # duplicated ids in object_ids produce duplicated output rows, as the reference loop does
    pos_in_uniq = pd.Categorical(object_ids, categories=uniq).codes
    starts_u = np.concatenate([[0], np.cumsum(counts_u)])
    has_rows = counts_u[pos_in_uniq] > 0
    kept_ids = object_ids[has_rows]
    sel = pos_in_uniq[has_rows]
    if len(uniq) == len(object_ids) and has_rows.all():
        gather = rows
        n = counts_u
    else:
        n = counts_u[sel]
        gather = np.concatenate([rows[starts_u[u]:starts_u[u + 1]] for u in sel]) if len(sel) else rows[:0]
    offsets = np.zeros(len(n) + 1, np.int64)
    np.cumsum(n, out=offsets[1:])
    t = np.ascontiguousarray(lightcurves[COL_T].to_numpy(dtype=np.float64)[gather])
    f = np.ascontiguousarray(lightcurves[COL_F].to_numpy(dtype=np.float64)[gather])
    e = np.ascontiguousarray(lightcurves[COL_E].to_numpy(dtype=np.float64)[gather])
    b = np.ascontiguousarray(band_codes(lightcurves[COL_B])[gather])
    return {"offsets": offsets, "t": t, "flux": f, "err": e, "band": b}, list(kept_ids)


def select_objects(csr, ids, object_ids):
    """Objects ``object_ids`` of an existing CSR batch (``ids[k]`` names object k), in the order of ``object_ids`` --
    what ``pack_lightcurves(frame, object_ids)`` returns for the frame the batch was packed from: ids without rows
    (or unknown) are dropped, a repeated id repeats its object.  Returns ``(csr, kept_ids)``; no copy when the
    selection is the whole batch in its own order."""
    ids = list(ids)
    object_ids = list(object_ids)
    if object_ids == ids:
        return csr, ids
    # ids are matched by their text: the C++ reader returns the id column as strings, a metadata frame read by pandas
    # may hold the same ids as integers; the kept ids are the caller's objects
    pos = {}
    for k, i in enumerate(ids):
        pos.setdefault(str(i), k)                        # first occurrence, as Categorical codes do
    off = np.asarray(csr["offsets"], np.int64)
    hit = [(pos[str(i)], i) for i in object_ids if str(i) in pos and off[pos[str(i)] + 1] > off[pos[str(i)]]]
    sel = [k for k, _ in hit]
    kept = [i for _, i in hit]
    sel = np.asarray(sel, np.int64)
    n = off[sel + 1] - off[sel] if len(sel) else np.zeros(0, np.int64)
    offsets = np.zeros(len(sel) + 1, np.int64)
    np.cumsum(n, out=offsets[1:])
    # row gather: for object j rows off[sel[j]] .. off[sel[j] + 1]
    gather = (np.repeat(off[sel] - offsets[:-1], n) + np.arange(offsets[-1])) if len(sel) else np.zeros(0, np.int64)
    out = {"offsets": offsets}
    for k in ("t", "flux", "err", "band"):
        out[k] = np.ascontiguousarray(np.asarray(csr[k])[gather])
    return out, kept


def check_csr(csr):
    """Host-side shape checks done before any kernel launch (a malformed CSR must never reach
    the device: an out-of-bounds read can reset the GPU)."""
    off = np.asarray(csr["offsets"])
    if off.dtype != np.int64 or off.ndim != 1 or off.size < 1:
        raise ValueError("offsets must be int64[n_obj+1]")
    if off[0] != 0 or np.any(np.diff(off) < 0):
        raise ValueError("offsets must start at 0 and be non-decreasing")
    total = int(off[-1])
    for k, dt in (("t", np.float64), ("flux", np.float64), ("err", np.float64), ("band", np.uint8)):
        a = csr[k]
        if a.dtype != dt or a.ndim != 1 or a.size != total or not a.flags.c_contiguous:
            raise ValueError(f"{k} must be C-contiguous {np.dtype(dt).name}[{total}]")
    return off.size - 1, total
