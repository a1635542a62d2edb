"""Shared DataFrame plumbing of the extractors: pack -> C-ABI -> frame."""
import numpy as np

from ..columns import COLUMNS
from ..engine import extract_csr
from ..packing import pack_lightcurves


def run_extractor(set_name, lightcurves, object_ids=None, metadata=None, id_last=True, int_columns=()):
    import pandas as pd

    csr, kept = pack_lightcurves(lightcurves, object_ids)
    z = None
    if metadata is not None:
        # physics_based.py:481,496: z_lookup.get(obj_id, nan)
        zmap = dict(zip(metadata["object_id"], metadata["Z"]))
        z = np.array([zmap.get(i, np.nan) for i in kept], dtype=np.float64)
    out = extract_csr(set_name, csr, z=z)
    df = pd.DataFrame(out, columns=COLUMNS[set_name])
    for c in int_columns:
        df[c] = df[c].astype(np.int64)
    if id_last:
        df["object_id"] = kept
    else:
        df.insert(0, "object_id", kept)
    return df
