import numpy as np

from mallorn_astrophysics_amd import synth
from mallorn_astrophysics_amd.packing import pack_lightcurves


def test_pack_roundtrip_and_order():
    lc = synth.make_lightcurves(20, seed=3)
    ids = synth.object_ids(20)
    df, _ = synth.to_dataframe(lc, ids)
    csr, kept = pack_lightcurves(df)
    assert kept == ids
    for k in ("offsets", "t", "flux", "err", "band"):
        assert np.array_equal(csr[k], lc[k]), k
    # shuffled rows: objects interleaved in the frame, file order inside an object preserved
    rng = np.random.default_rng(0)
    perm = rng.permutation(len(df))
    sh = df.iloc[perm].reset_index(drop=True)
    csr2, kept2 = pack_lightcurves(sh, ids)
    assert kept2 == ids
    for i in range(20):
        s, e = csr2["offsets"][i], csr2["offsets"][i + 1]
        rows = sh[sh.object_id == ids[i]]
        assert np.array_equal(csr2["t"][s:e], rows["Time (MJD)"].to_numpy())
        assert np.array_equal(csr2["flux"][s:e], rows["Flux"].to_numpy())


def test_pack_missing_and_subset_ids():
    lc = synth.make_lightcurves(6, seed=4)
    ids = synth.object_ids(6)
    df, _ = synth.to_dataframe(lc, ids)
    want = [ids[4], "zzz_absent", ids[1]]
    csr, kept = pack_lightcurves(df, want)
    assert kept == [ids[4], ids[1]]          # statistical.py:163-165: absent ids are skipped
    assert len(csr["offsets"]) == 3
    n4 = lc["offsets"][5] - lc["offsets"][4]
    assert csr["offsets"][1] == n4
    assert np.array_equal(csr["t"][:n4], lc["t"][lc["offsets"][4]:lc["offsets"][5]])


def test_unknown_filter_maps_to_255():
    lc = synth.edge_cases()
    df, _ = synth.to_dataframe(lc)
    csr, _ = pack_lightcurves(df)
    assert np.array_equal(csr["band"], lc["band"])
    assert (csr["band"] == 255).sum() == 2


def test_device_batch_rejects_short_z_before_touching_the_gpu():
    """engine.DeviceBatch: a z vector that does not have one entry per object must never reach a kernel."""
    import numpy as np
    import pytest
    from mallorn_astrophysics_amd import synth
    from mallorn_astrophysics_amd.engine import DeviceBatch
    lc = synth.make_lightcurves(5, seed=3)
    with pytest.raises(ValueError):
        DeviceBatch(lc, z=lc["z"][:4], device=0)
    with pytest.raises(ValueError):
        DeviceBatch(lc, z=np.zeros((5, 1)), device=0)
