"""Frame assembly: the cached feature frames -> the model matrices the train_v*.py scripts build.

Host side of SURVEY.md §8 row a18 (pandas, no GPU work): what the consumers of the caches do with them
before handing ``X`` to XGBoost.  Every statement below restates the reference's own pandas statement (cited
per line), so that name collisions, column order and dtypes come out as pandas makes them there:

* ``assemble_v34a``  -- ``scripts/train_v34a_bazin.py:55-123``: 120 selected base columns + 25 TDE + 27 GP2D
  + 52 Bazin = the 224-column frame.
* ``assemble_v92d``  -- ``non_successful_tests/scripts/train_v92_focal_adversarial.py:65-103``: all four caches
  merged, filtered by v34a's saved ``feature_names`` minus the two shift features, ``nan_to_num(posinf=1e10)``.

Two traps of the reference that are reproduced, not repaired:

1. ``train_v34a_bazin.py:75`` merges the TDE cache into ``train_base`` BEFORE the selection is applied, so the
   colour feature ``temp_stability`` (``colors.py:340``) and the TDE feature of the same name
   (``tde_physics.py:265``) become ``temp_stability_x`` / ``temp_stability_y`` there; a selection list that
   names ``temp_stability`` raises ``KeyError`` in the reference and here.
2. ``train_v92_focal_adversarial.py:86-88`` merges all four caches, so ``temp_stability`` and ``r_bazin_t0``
   (``physics_based.py:430`` vs ``bazin_fitting.py:159``) both exist only with ``_x`` / ``_y`` suffixes in
   ``train_all``; the filter ``f in train_all.columns`` (:97) then silently drops the plain names that v34a's
   ``feature_names`` holds.  With the reference's own statements the v92d matrix therefore has 224 - 2 shift
   features - 2 suffixed names = 220 columns whenever both plain names are in the v34a list (BASELINE.json's
   "222" counts only the two shift features).
"""
from __future__ import annotations

import numpy as np

SHIFT_FEATURES = ["all_rise_time", "all_asymmetry"]          # train_v92_focal_adversarial.py:95


def select_features(selection, n: int = 120):
    """``selected_120`` of train_v34a_bazin.py:59-68: walk ``high_corr_df`` in row order, drop ``feature_2`` of
    every pair whose ``feature_1`` has not been dropped, keep the first ``n`` rows of ``importance_df`` that
    survive."""
    importance_df = selection["importance_df"]
    high_corr_df = selection["high_corr_df"]
    corr_to_drop = set()
    for _, row in high_corr_df.iterrows():                                        # :63-65
        if row["feature_1"] not in corr_to_drop:
            corr_to_drop.add(row["feature_2"])
    clean_features = importance_df[~importance_df["feature"].isin(corr_to_drop)]  # :66
    return clean_features.head(n)["feature"].tolist()                             # :67


def assemble_v34a(base, tde, gp2d, bazin, selection, n_selected: int = 120):
    """The v34a frame of one data split.  Returns ``(X, feature_names, frame)``: ``X`` float matrix
    ``[n_obj, 224]``, names in column order, and the merged DataFrame (``object_id`` first).

    base   ``features_v4_cache.pkl['train_features' | 'test_features']``  (train_v34a_bazin.py:55-57)
    tde    ``tde_physics_cache.pkl['train' | 'test']``                      (:70-73)
    gp2d   ``multiband_gp_cache.pkl['train' | 'test']``                     (:78-82)
    bazin  output of ``extract_bazin_features`` (the script recomputes it inline, :99-107)
    selection  ``selected_features.pkl``: dict with ``importance_df`` and ``high_corr_df`` (:59-61)
    """
    selected = select_features(selection, n_selected)
    gp2d_cols = [c for c in gp2d.columns if c != "object_id"]                     # :82
    base = base.merge(tde, on="object_id", how="left")                            # :75  (collision -> _x/_y)
    v21 = base[["object_id"] + selected].copy()                                   # :84
    v21 = v21.merge(tde, on="object_id", how="left")                              # :85
    v21 = v21.merge(gp2d[["object_id"] + gp2d_cols], on="object_id", how="left")  # :86
    combined = v21.merge(bazin, on="object_id", how="left")                       # :118
    X = combined.drop(columns=["object_id"]).values                               # :121
    feature_names = [c for c in combined.columns if c != "object_id"]             # :123
    return X, feature_names, combined


def assemble_v92d(base, tde, gp2d, bazin, v34a_features):
    """The v92d matrix of one data split (train_v92_focal_adversarial.py:86-103).  Returns
    ``(X, available_features, frame)``."""
    all_ = base.merge(tde, on="object_id", how="left")                            # :86
    all_ = all_.merge(gp2d, on="object_id", how="left")                           # :87
    all_ = all_.merge(bazin, on="object_id", how="left")                          # :88
    available = [f for f in v34a_features if f in all_.columns and f not in SHIFT_FEATURES]   # :96
    X = all_[available].values                                                    # :99
    X = np.nan_to_num(X, nan=np.nan, posinf=1e10, neginf=-1e10)                   # :102
    return X, available, all_


def load_caches(processed_dir, split: str = "train"):
    """Read the four cache pickles written by scripts/precompute_features.py / cache_bazin_features.py
    (files this package wrote itself) and return ``(base, tde, gp2d, bazin)`` of one split."""
    import pickle
    from pathlib import Path

    d = Path(processed_dir)

    def rd(name):
        with open(d / name, "rb") as f:
            return pickle.load(f)

    base = rd("features_v4_cache.pkl")[f"{split}_features"]
    return base, rd("tde_physics_cache.pkl")[split], rd("multiband_gp_cache.pkl")[split], \
        rd("bazin_features_cache.pkl")[split]
