"""Frame assembly -- host logic, runs without a GPU.

The fixture holds the cache frames the REFERENCE extractors produced for a synthetic train / test split, a
synthetic ``selected_features.pkl``, and the ``feature_names`` / ``X`` matrices the reference's own statements
(``train_v34a_bazin.py:55-123``, ``train_v92_focal_adversarial.py:65-103``, exec'd from the checkout by
tests/golden/make_frames_golden.py) built from them.  ``frames.assemble_*`` must reproduce them exactly: same
names, same order, same values bit for bit."""
import numpy as np
import pytest

import frames_fixture as fx
from mallorn_astrophysics_amd import frames
from mallorn_astrophysics_amd.columns import COLUMNS, STAT_INT_COLUMNS
from mallorn_astrophysics_amd.features.statistical import add_metadata_features


def same(a, b):
    return a.shape == b.shape and np.array_equal(np.nan_to_num(a, nan=-7.25e300), np.nan_to_num(b, nan=-7.25e300))


@pytest.fixture(scope="module")
def g():
    return fx.load()


@pytest.mark.parametrize("split", ["train", "test"])
def test_v34a_frame_equals_reference(g, split):
    X, names, frame = frames.assemble_v34a(fx.frame(g, split, "base"), fx.frame(g, split, "tde"),
                                           fx.frame(g, split, "gp2d"), fx.frame(g, split, "bazin"), fx.selection(g))
    assert names == [str(c) for c in g["v34a_names"]]
    assert len(names) == 224 and frame.columns[0] == "object_id"
    assert X.dtype == np.float64 and same(X, g[f"v34a_X_{split}"])
    # 120 selected + 25 TDE + 27 GP2D + 52 Bazin, in that order
    assert names[120:145] == COLUMNS["tde"] and names[145:172] == COLUMNS["gp2d"] and names[172:] == COLUMNS["bazin"]


@pytest.mark.parametrize("split", ["train", "test"])
def test_v92d_matrix_equals_reference(g, split):
    X, names, _ = frames.assemble_v92d(fx.frame(g, split, "base"), fx.frame(g, split, "tde"),
                                       fx.frame(g, split, "gp2d"), fx.frame(g, split, "bazin"),
                                       [str(c) for c in g["v34a_names"]])
    assert names == [str(c) for c in g["v92d_names"]]
    assert same(X, g[f"v92d_X_{split}"])
    # the reference's filter drops the two shift features AND the two names its merges suffixed (_x/_y)
    dropped = [c for c in g["v34a_names"] if c not in names]
    assert dropped == ["all_rise_time", "all_asymmetry", "temp_stability", "r_bazin_t0"]
    assert not np.isposinf(X).any() and not np.isneginf(X).any()


def test_selection_naming_temp_stability_raises_like_the_reference(g):
    sel = fx.selection(g)
    imp = sel["importance_df"]
    sel["importance_df"] = imp.iloc[::-1].reset_index(drop=True)          # temp_stability (ranked last) first
    with pytest.raises(KeyError):
        frames.assemble_v34a(fx.frame(g, "train", "base"), fx.frame(g, "train", "tde"), fx.frame(g, "train", "gp2d"),
                             fx.frame(g, "train", "bazin"), sel)


def test_nan_to_num_of_v92d_clamps_infinities(g):
    base = fx.frame(g, "train", "base").copy()
    col = [c for c in g["v92d_names"] if c in base.columns][0]
    base.loc[0, col] = np.inf
    base.loc[1, col] = -np.inf
    X, names, _ = frames.assemble_v92d(base, fx.frame(g, "train", "tde"), fx.frame(g, "train", "gp2d"),
                                       fx.frame(g, "train", "bazin"), [str(c) for c in g["v34a_names"]])
    j = names.index(col)
    assert X[0, j] == 1e10 and X[1, j] == -1e10


@pytest.mark.parametrize("split", ["train", "test"])
def test_add_metadata_features_values(g, split):
    """statistical.py:229-253 against the reference's own output (the first 127 columns of the base frame)."""
    import pandas as pd
    base = fx.frame(g, split, "base")
    stat = base[["object_id"] + COLUMNS["stat"]].copy()
    ids = [str(i) for i in g[f"{split}_base_ids"]]
    # metadata rows in a different order, plus an object without light-curve rows: the merge is a LEFT join on features
    meta = pd.DataFrame({"object_id": ids[::-1] + ["absent"], "Z": list(g[f"{split}_z"][::-1]) + [0.5],
                         "EBV": list(g[f"{split}_ebv"][::-1]) + [0.1], "target": 0})
    out = add_metadata_features(stat, meta)
    want = base[["object_id"] + COLUMNS["stat"] + ["Z", "EBV", "luminosity_distance", "time_dilation"]]
    assert list(out.columns) == list(want.columns)
    assert list(out["object_id"]) == ids
    for c in STAT_INT_COLUMNS:
        assert out[c].dtype == np.int64
    assert same(out.drop(columns=["object_id"]).to_numpy(float), want.drop(columns=["object_id"]).to_numpy(float))
