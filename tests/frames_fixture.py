"""Loader of tests/golden/golden_frames.npz (made by tests/golden/make_frames_golden.py from the reference)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_frames.npz")


def load():
    g = np.load(GOLDEN)
    return {k: g[k] for k in g.files}


def frame(g, split, name):
    """Rebuild a cache frame (column order, object_id position and int64 columns as the reference left them)."""
    import pandas as pd
    cols = [str(c) for c in g[f"{split}_{name}_cols"]]
    feat = [c for c in cols if c != "object_id"]
    df = pd.DataFrame(g[f"{split}_{name}_val"], columns=feat)
    for c in g[f"{split}_{name}_int"]:
        df[str(c)] = df[str(c)].astype(np.int64)
    df["object_id"] = [str(i) for i in g[f"{split}_{name}_ids"]]
    return df[cols]


def selection(g):
    import pandas as pd
    imp = [str(c) for c in g["sel_importance"]]
    return {"importance_df": pd.DataFrame({"feature": imp, "importance": np.linspace(1.0, 0.0, len(imp))}),
            "high_corr_df": pd.DataFrame({"feature_1": [str(c) for c in g["sel_corr_1"]],
                                          "feature_2": [str(c) for c in g["sel_corr_2"]]})}


def csr(g, split):
    return {k: g[f"{split}_{k}"] for k in ("offsets", "t", "flux", "err", "band", "z", "ebv")}
