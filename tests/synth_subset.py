"""Take a subset of the objects of a CSR batch."""
import numpy as np


def take(csr, rows):
    off = csr["offsets"]
    idx = np.concatenate([np.arange(off[i], off[i + 1]) for i in rows]) if len(rows) else np.zeros(0, np.int64)
    n = np.array([off[i + 1] - off[i] for i in rows], np.int64)
    out = {k: np.ascontiguousarray(csr[k][idx]) for k in ("t", "flux", "err", "band")}
    out["offsets"] = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
    for k in ("z", "ebv"):
        if k in csr:
            out[k] = np.asarray(csr[k])[list(rows)]
    return out
