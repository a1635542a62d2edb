"""Golden fixture of the frame assembly: the 224-column v34a frame and the v92d
matrix, built by the REFERENCE's own statements from REFERENCE extractor outputs.

Run in the build container only (needs the read-only reference checkout):

    python tests/golden/make_frames_golden.py [/root/reference]

What runs, all from the checkout at generation time (nothing of it is stored):
  * ``src/features/{statistical,colors,lightcurve_shape,physics_based,tde_physics,bazin_fitting}.py`` imported
    unchanged build the cache frames exactly as ``scripts/train_v4_physics.py:60-80`` (statistics + metadata +
    colours + shapes + physics, three left merges), ``train_v7_tde_physics.py:79-99`` and
    ``cache_bazin_features.py:31-45`` do;
  * the 2-D GP cache comes from ``oracle/gp2d.py`` (george is not installed: PARITY UNPINNED for those 27
    columns, as everywhere else);
  * lines 55-123 of ``scripts/train_v34a_bazin.py`` and lines 65-103 of
    ``non_successful_tests/scripts/train_v92_focal_adversarial.py`` are exec'd on those caches in a scratch
    directory, with a synthetic ``selected_features.pkl`` (``importance_df`` / ``high_corr_df``: the real one is
    git-ignored model output, SURVEY.md finding 4).

Stored (arrays and names only): the synthetic inputs, the selection lists, the cache frames, and the
reference's ``feature_names`` / ``X_train`` / ``X_test`` for v34a and v92d -> ``tests/golden/golden_frames.npz``.
"""
import os
import pickle
import sys
import tempfile
import warnings
from pathlib import Path

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.path.insert(0, os.path.join(REF, "src"))

from mallorn_astrophysics_amd import synth  # noqa: E402
from mallorn_astrophysics_amd.columns import COLUMNS  # noqa: E402


def datasets():
    train = synth.concat([synth.make_lightcurves(40, seed=341, n_median=100.0), synth.edge_cases()])
    test = synth.make_lightcurves(24, seed=342, n_median=100.0)
    return train, test


def synthetic_selection(base_cols, seed=7):
    """A ``selected_features.pkl`` of the reference's shape (``feature_selection.py:353-356``): an importance
    table over the base columns and a high-correlation pair list.  ``temp_stability`` is ranked last: the
    reference script raises KeyError when the selection names it (see frames.py, trap 1)."""
    import pandas as pd
    rng = np.random.default_rng(seed)
    feats = [c for c in base_cols if c not in ("object_id", "peak_mjd", "r_bazin_t0", "temp_stability")]
    # the two "shift" features v92d removes are confirmed members of the real selection (WRITEUP.md:40-59): rank them first
    shift = ["all_rise_time", "all_asymmetry"]
    feats = shift + list(rng.permutation([c for c in feats if c not in shift])) + ["temp_stability"]
    importance_df = pd.DataFrame({"feature": feats, "importance": np.linspace(1.0, 0.0, len(feats))})
    pairs = rng.choice(np.arange(2, len(feats) - 1), size=(40, 2))
    pairs = pairs[pairs[:, 0] != pairs[:, 1]]
    high_corr_df = pd.DataFrame({"feature_1": [feats[a] for a, _ in pairs], "feature_2": [feats[b] for _, b in pairs],
                                 "correlation": rng.uniform(0.95, 1.0, len(pairs))})
    return {"importance_df": importance_df, "high_corr_df": high_corr_df}


def main():
    import pandas as pd
    from features import bazin_fitting, colors, lightcurve_shape, physics_based, statistical, tde_physics
    import oracle

    warnings.simplefilter("ignore")
    train, test = datasets()
    store = {}
    frames = {}
    for split, lc in (("train", train), ("test", test)):
        ids = synth.object_ids(len(lc["offsets"]) - 1, prefix=split)
        df, meta = synth.to_dataframe(lc, ids)
        for k in ("offsets", "t", "flux", "err", "band", "z", "ebv"):
            store[f"{split}_{k}"] = lc[k]
        # train_v4_physics.py:60-80
        stat = statistical.add_metadata_features(statistical.extract_statistical_features(df, ids), meta)
        base = stat.merge(colors.extract_color_features(df, ids), on="object_id", how="left")
        base = base.merge(lightcurve_shape.extract_shape_features(df, ids), on="object_id", how="left")
        base = base.merge(physics_based.extract_physics_features(df, meta, ids), on="object_id", how="left")
        tde = tde_physics.extract_tde_physics_features(df, ids)
        bz = bazin_fitting.extract_bazin_features(df, ids)
        # the 2-D GP cache: oracle (george absent), object_id last as multiband_gp.py:379-385 leaves it
        kept = [i for i, n in zip(ids, np.diff(lc["offsets"])) if n > 0]
        gp = pd.DataFrame(oracle.extract("gp2d", lc), columns=COLUMNS["gp2d"])
        gp["object_id"] = ids
        gp = gp[gp["object_id"].isin(kept)].reset_index(drop=True)
        frames[split] = dict(base=base, tde=tde, gp2d=gp, bazin=bz, ids=ids)
        for name, fr in (("base", base), ("tde", tde), ("gp2d", gp), ("bazin", bz)):
            cols = [c for c in fr.columns if c != "object_id"]
            store[f"{split}_{name}_cols"] = np.array(list(fr.columns), dtype=str)       # incl. object_id position
            store[f"{split}_{name}_ids"] = np.array(fr["object_id"].tolist(), dtype=str)
            store[f"{split}_{name}_val"] = fr[cols].to_numpy(np.float64)
            store[f"{split}_{name}_int"] = np.array([c for c in cols if fr[c].dtype == np.int64], dtype=str)
        print(split, "base", base.shape, "tde", tde.shape, "gp2d", gp.shape, "bazin", bz.shape, flush=True)

    selection = synthetic_selection(list(frames["train"]["base"].columns))
    store["sel_importance"] = np.array(selection["importance_df"]["feature"].tolist(), dtype=str)
    store["sel_corr_1"] = np.array(selection["high_corr_df"]["feature_1"].tolist(), dtype=str)
    store["sel_corr_2"] = np.array(selection["high_corr_df"]["feature_2"].tolist(), dtype=str)

    with tempfile.TemporaryDirectory() as tmp:
        base_path = Path(tmp)
        proc = base_path / "data/processed"
        proc.mkdir(parents=True)
        pd.to_pickle({"train_features": frames["train"]["base"], "test_features": frames["test"]["base"]},
                     proc / "features_v4_cache.pkl")
        pd.to_pickle({"train": frames["train"]["tde"], "test": frames["test"]["tde"]}, proc / "tde_physics_cache.pkl")
        pd.to_pickle({"train": frames["train"]["gp2d"], "test": frames["test"]["gp2d"]}, proc / "multiband_gp_cache.pkl")
        pd.to_pickle({"train": frames["train"]["bazin"], "test": frames["test"]["bazin"]}, proc / "bazin_features_cache.pkl")
        pd.to_pickle(selection, proc / "selected_features.pkl")

        # ---- train_v34a_bazin.py:55-123, executed from the checkout
        src = open(os.path.join(REF, "scripts", "train_v34a_bazin.py")).read().splitlines()
        train_df, _ = synth.to_dataframe(train, frames["train"]["ids"])
        test_df, _ = synth.to_dataframe(test, frames["test"]["ids"])
        ns = {"pd": pd, "np": np, "pickle": pickle, "base_path": base_path, "train_lc": train_df, "test_lc": test_df,
              "train_ids": frames["train"]["ids"], "test_ids": frames["test"]["ids"]}
        exec(compile("\n".join(src[54:123]), "train_v34a_bazin.py[55:123]", "exec"), ns)
        names34 = ns["feature_names"]
        assert len(names34) == 224, len(names34)
        store["v34a_names"] = np.array(names34, dtype=str)
        store["v34a_X_train"] = np.asarray(ns["X_train"], np.float64)
        store["v34a_X_test"] = np.asarray(ns["X_test"], np.float64)
        print("v34a", ns["X_train"].shape, ns["X_test"].shape, flush=True)

        # ---- train_v92_focal_adversarial.py:65-103
        with open(proc / "v34a_artifacts.pkl", "wb") as f:
            pickle.dump({"feature_names": names34}, f)
        src = open(os.path.join(REF, "non_successful_tests", "scripts", "train_v92_focal_adversarial.py")).read().splitlines()
        ns2 = {"pd": pd, "np": np, "pickle": pickle, "base_path": base_path}
        exec(compile("\n".join(src[64:103]), "train_v92_focal_adversarial.py[65:103]", "exec"), ns2)
        store["v92d_names"] = np.array(ns2["available_features"], dtype=str)
        store["v92d_X_train"] = np.asarray(ns2["X_train"], np.float64)
        store["v92d_X_test"] = np.asarray(ns2["X_test"], np.float64)
        print("v92d", ns2["X_train"].shape, ns2["X_test"].shape, flush=True)

    np.savez_compressed(os.path.join(HERE, "golden_frames.npz"), **store)


if __name__ == "__main__":
    main()
