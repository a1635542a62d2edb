"""Parity tests proper: the HIP path through the C-ABI against the golden fixtures (real
reference outputs) and against the oracle on seeded synthetic inputs.  Need an MI355X."""
import os

import numpy as np
import pytest

import oracle
import parity
from conftest import load_golden
from mallorn_astrophysics_amd import synth
from mallorn_astrophysics_amd.columns import COLUMNS, SET_NAMES, STAT_INT_COLUMNS
from mallorn_astrophysics_amd.engine import extract_csr

pytestmark = pytest.mark.gpu

# tolerance written per set: BASELINE.md asks <= 1e-4 relative on fitted parameters and bit-exact
# counts; closed-form statistics are held much tighter.
TOL = {"stat": dict(rtol=1e-9, atol=1e-12), "tde": dict(rtol=1e-8, atol=1e-8), "color": dict(rtol=1e-9, atol=1e-10),
       "shape": dict(rtol=1e-8, atol=1e-10), "physics": dict(rtol=1e-9, atol=1e-10),
       # v115 research features: closed-form least squares instead of np.polyfit, a direct sum instead of scipy's
       # (possibly FFT-based) convolution; near-zero slopes of constant light curves need the absolute floor
       "research": dict(rtol=1e-8, atol=1e-9)}
INT = {"stat": STAT_INT_COLUMNS}
# band fits (of 288 + 72 cross-band blocks) of the gp1d fixture that may miss 1e-4 although the reference is stable there
GP1D_MAX_STABLE_BAD = 12
SETS = list(TOL)


@pytest.mark.parametrize("name", SETS)
def test_golden(name, golden_inputs):
    ref = load_golden(name)
    got = extract_csr(name, golden_inputs, z=golden_inputs["z"])
    bad = parity.compare(got, ref, COLUMNS[name], int_cols=INT.get(name, ()), label=name, **TOL[name])
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("name", SETS)
def test_seeded_vs_oracle(name):
    lc = synth.make_lightcurves(300, seed=1234)
    got = extract_csr(name, lc, z=lc["z"])
    ref = oracle.extract(name, lc, lc["z"])
    bad = parity.compare(got, ref, COLUMNS[name], int_cols=INT.get(name, ()), label=name, **TOL[name])
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("name", SETS)
def test_lds_tiers_and_long_objects(name):
    """Objects of 100..2048 points exercise every LDS tier; longer ones (the reference has no cap: statistical.py:41-132,
    research_features.py:163-430) take the long-object tier, whose working set lives in global scratch -- up to 16384
    rows; beyond that a NaN row (documented)."""
    rng = np.random.default_rng(5)
    objs = []
    for n in (100, 128, 129, 256, 257, 500, 512, 513, 1000, 1024, 1500, 2048, 2049, 3000, 5000, 16385):
        t = np.sort(59000 + rng.uniform(0, 800, n))
        f = 30 * np.exp(-0.5 * ((t - 59300) / 40) ** 2) + rng.normal(0, 1, n)
        objs.append((t, f, np.full(n, 1.0), rng.choice(6, n)))
    lc = synth.from_objects(objs)
    got = extract_csr(name, lc, z=lc["z"])
    n_ok = len(objs) - 1
    sub = {k: (v[:lc["offsets"][n_ok]] if k in ("t", "flux", "err", "band") else v) for k, v in lc.items()}
    sub["offsets"] = lc["offsets"][:n_ok + 1]
    ref = oracle.extract(name, sub, lc["z"][:n_ok])
    bad = parity.compare(got[:n_ok], ref, COLUMNS[name], int_cols=INT.get(name, ()), label=name, **TOL[name])
    assert not bad, "\n".join(bad)
    assert np.isnan(got[n_ok:]).all()


def test_research_long_r_band_span_takes_the_long_tier():
    """An r band spanning more than 4096 days does not fit the Mexican-hat grid of the LDS tiers (32 KiB); the light
    curve is handed to the long-object tier (64 Ki-day grid in global scratch) instead of coming back as NaN with
    status -100 (research_features.py:352-430 has no limit)."""
    rng = np.random.default_rng(8)
    objs = []
    for n, span in ((150, 300.0), (200, 5000.0), (900, 9000.0), (60, 70000.0)):
        t = np.sort(50000 + rng.uniform(0, span, n))
        b = rng.choice(6, n, p=[.05, .1, .4, .2, .15, .1])
        f = 20 + 5 * np.sin((t - 50000) / 37.0) + rng.normal(0, 1, n)
        objs.append((t, f, np.full(n, 0.8), b))
    lc = synth.from_objects(objs)
    got, st = extract_csr("research", lc, z=lc["z"], return_status=True)
    assert (st[:3, 0] == 0).all() and st[3, 0] == -100        # 70000 days: beyond the long tier's grid as well
    keep = synth.from_objects(objs[:3])
    ref = oracle.extract("research", keep, lc["z"][:3])
    bad = parity.compare(got[:3], ref, COLUMNS["research"], label="research", **TOL["research"])
    assert not bad, "\n".join(bad)


def test_statistics_band_shapes():
    """Band layouts that take different routes through the lean statistics kernel: bands of up to 32
    / 64 rows (8-lane groups, 4 / 8 values per lane), longer bands (full-wave passes sized by the
    band), empty bands, a NaN flux inside a long band, and objects handed to the general kernel
    (unknown band code, rows out of time order) in every lean tier."""
    rng = np.random.default_rng(21)
    objs = []

    def add(counts, nan_at=None, unknown=False, shuffle=False):
        n = int(sum(counts))
        t = np.sort(59000 + rng.uniform(0, 600, n))
        b = rng.permutation(np.repeat(np.arange(6), counts)).astype(np.int64)
        f = 25 * np.exp(-0.5 * ((t - 59250) / 60) ** 2) * (1 + 0.2 * b) + rng.normal(0, 2, n)
        e = rng.uniform(0.5, 2.0, n)
        if nan_at is not None:
            f[np.flatnonzero(b == nan_at)[3]] = np.nan
        if unknown:
            b[rng.choice(n, 3, replace=False)] = 255
        if shuffle:
            p = rng.permutation(n)
            t, f, e, b = t[p], f[p], e[p], b[p]
        objs.append((t, f, e, b))

    add([10, 20, 32, 31, 5, 0])            # all bands <= 32
    add([0, 33, 64, 12, 0, 1])             # bands <= 64
    add([3, 65, 20, 20, 10, 2])            # one band on the full wave, tier 128
    add([0, 0, 128, 0, 0, 0])              # single band, n = 128
    add([5, 129, 70, 30, 10, 12])          # bands of 129 and 70 rows, tier 256
    add([0, 0, 0, 256, 0, 0])              # single band, n = 256
    add([40, 257, 90, 60, 33, 32])         # band > 256 rows, tier 512
    add([85, 85, 85, 85, 86, 86])          # n = 512, every band on the full wave
    add([3, 65, 20, 20, 10, 2], nan_at=1)  # NaN inside a long band
    add([10, 20, 30, 31, 5, 4], nan_at=2)
    add([10, 20, 30, 31, 5, 4], unknown=True)      # general kernel, CAP 128
    add([20, 60, 70, 40, 5, 5], shuffle=True)      # general kernel, CAP 256
    add([50, 100, 100, 80, 40, 30], unknown=True, shuffle=True)   # general kernel, CAP 512
    lc = synth.from_objects(objs)
    got = extract_csr("stat", lc)
    ref = oracle.extract("stat", lc)
    bad = parity.compare(got, ref, COLUMNS["stat"], int_cols=STAT_INT_COLUMNS, rtol=1e-9, atol=1e-12)
    assert not bad, "\n".join(bad)


def lane_route_objects():
    """The hand-made light curves of test_statistics_lane_routes (also fed to the bounds-checked debug build)."""
    rng = np.random.default_rng(77)
    objs = []
    _lane_route_fill(objs, rng)
    return objs, rng


def test_statistics_lane_routes():
    """The eight- and four-light-curves-per-wavefront statistics kernels (stat_lanes.hpp, stat_lanes16.hpp): band lengths
    on both sides of every routing threshold of the plan kernels (16- / 32-row lanes, 8 / 16 lanes per light curve, r and i
    over two / four lanes, 128 / 256 / 512 rows), bands of
    0 / 1 / 2 rows, odd and even r / i halves, negative and zero times, NaN and inf fluxes in either half, rows out of
    time order and unknown band codes inside lane-eligible light curves, and batch lengths that do not fill the last
    group of eight -- each object against the oracle, and the batch in a different order."""
    objs, rng = lane_route_objects()
    lc = synth.from_objects(objs)
    got = extract_csr("stat", lc)
    ref = oracle.extract("stat", lc)
    bad = parity.compare(got, ref, COLUMNS["stat"], int_cols=STAT_INT_COLUMNS, rtol=1e-9, atol=1e-12)
    assert not bad, "\n".join(bad)
    # the same light curves in another order and another batch length (3 light curves in the last group of 8 -> 5)
    order = rng.permutation(len(objs))[:-2]
    lc2 = synth.from_objects([objs[i] for i in order])
    got2 = extract_csr("stat", lc2)
    key = lambda m: np.nan_to_num(m, nan=-7.25e300)
    assert np.array_equal(key(got2), key(got[order])), "a light curve's statistics depend on its batch"


def _lane_route_fill(objs, rng):
    def add(counts, t0=59000.0, nan_band=None, inf_band=None, shuffle=False, unknown=False, dup_t=False):
        n = int(sum(counts))
        t = np.sort(t0 + rng.uniform(0, 500, n))
        if dup_t and n > 3:
            t[2] = t[1]
        b = rng.permutation(np.repeat(np.arange(6), counts)).astype(np.int64)
        f = 20 * np.exp(-0.5 * ((t - t0 - 200) / 50) ** 2) * (1 + 0.3 * b) + rng.normal(0, 1.5, n)
        e = rng.uniform(0.3, 2.0, n)
        e[rng.random(n) < 0.05] = 0.0                     # rows the SNR mean leaves out
        if nan_band is not None:
            f[np.flatnonzero(b == nan_band)[-2]] = np.nan  # second half of a split band
        if inf_band is not None:
            f[np.flatnonzero(b == inf_band)[0]] = np.inf
        if unknown:
            b[rng.choice(n, 2, replace=False)] = 200
        if shuffle:
            p = rng.permutation(n)
            t, f, e, b = t[p], f[p], e[p], b[p]
        objs.append((t, f, e, b))

    for c in ([16, 16, 32, 32, 16, 16], [17, 16, 32, 32, 16, 16], [16, 16, 33, 32, 16, 16], [16, 16, 32, 32, 16, 0],
              [32, 32, 64, 64, 32, 32], [33, 32, 64, 64, 32, 31], [32, 32, 65, 64, 32, 31], [32, 32, 63, 63, 32, 32],
              [0, 0, 1, 0, 0, 0], [1, 1, 1, 1, 1, 1], [2, 0, 2, 3, 0, 1], [0, 0, 0, 0, 0, 5], [5, 0, 0, 0, 0, 0],
              [3, 9, 31, 30, 20, 7], [10, 20, 41, 40, 30, 12], [4, 4, 64, 4, 4, 4], [4, 4, 4, 64, 4, 48],
              [8, 8, 40, 40, 16, 16], [20, 20, 20, 20, 24, 24], [30, 30, 60, 60, 30, 30], [12, 12, 48, 47, 5, 5]):
        add(c)
    # 16 lanes per light curve (stat_lanes16.hpp): bands of up to 64 rows (r, i: 128) in up to 512 rows, on both sides
    # of the thresholds, four-part splits with uneven parts, parts that stay empty
    for c in ([64, 64, 128, 128, 64, 64], [65, 64, 128, 128, 64, 63], [64, 64, 129, 127, 64, 64], [33, 10, 70, 75, 40, 12],
              [10, 10, 101, 99, 10, 10], [1, 0, 97, 3, 33, 2], [40, 0, 0, 0, 0, 41], [0, 0, 5, 0, 0, 33], [60, 60, 60, 60, 8, 8],
              [50, 50, 100, 100, 50, 50], [34, 34, 66, 66, 34, 22]):
        add(c)
    add([33, 10, 70, 75, 40, 12], nan_band=2)
    add([33, 10, 70, 75, 40, 12], nan_band=4, inf_band=3)
    add([50, 50, 100, 100, 50, 50], nan_band=3)
    add([50, 50, 100, 100, 50, 50], t0=-100.0, dup_t=True)
    add([50, 50, 100, 100, 50, 50], shuffle=True)         # general kernel
    add([10, 10, 30, 30, 10, 10], t0=-250.0)              # times around zero: slots behind a lane's rows hold 0.0
    add([10, 10, 30, 30, 10, 10], t0=0.0)
    add([6, 6, 21, 20, 6, 6], nan_band=2)
    add([6, 6, 20, 21, 6, 6], nan_band=3)
    add([6, 6, 20, 21, 6, 6], inf_band=3)
    add([6, 6, 20, 21, 6, 6], nan_band=0, inf_band=2)
    add([6, 6, 20, 21, 6, 6], dup_t=True)
    add([6, 6, 20, 21, 6, 6], shuffle=True)               # general kernel
    add([6, 6, 20, 21, 6, 6], unknown=True)               # general kernel
    add([30, 30, 60, 60, 30, 30], shuffle=True)
    add([30, 30, 60, 60, 30, 30], unknown=True)


_DEBUG_CHILD = """
import ctypes, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
from mallorn_astrophysics_amd import synth, _lib
from mallorn_astrophysics_amd.engine import extract_csr
from test_gpu_parity import lane_route_objects
lib = _lib.load()
objs, _ = lane_route_objects()
a = extract_csr("stat", synth.from_objects(objs))
b = extract_csr("stat", synth.make_lightcurves(6000, seed=4242))
if hasattr(lib, "lcfe_debug_lanes_check"):
    buf = (ctypes.c_uint * 4)()
    assert lib.lcfe_debug_lanes_check(buf) == 0
    print("LANES_CHECK", list(buf))
np.savez({out!r}, a=a, b=b)
"""


def test_statistics_lanes_debug_build(tmp_path):
    """SURVEY.md section 5, sanitizer row (the GPU AddressSanitizer is not available on this pool): the -DLCFE_DEBUG build
    of the library (`make -C mallorn-astrophysics_amd/csrc debug`, built by __graft_entry__.build()) runs the statistics
    lanes kernels -- 98 % of the survey's light curves, device-only code the host simulation cannot cover -- with every
    LDS index range-checked and canary words around their buffers.  Fed with the hand-made routing cases and 6,000 seeded
    light curves it must count no out-of-range index and no damaged canary, and return what the release build returns,
    bit for bit."""
    import subprocess
    import sys as _sys
    from conftest import ROOT
    dbg = os.path.join(ROOT, "mallorn-astrophysics_amd", "csrc", "build", "liblcfe_debug.so")
    if not os.path.exists(dbg):
        pytest.skip("debug library not built (make -C mallorn-astrophysics_amd/csrc debug)")
    outs = {}
    for tag, env in (("debug", dict(os.environ, LCFE_LIB_PATH=dbg)), ("release", {k: v for k, v in os.environ.items() if k != "LCFE_LIB_PATH"})):
        path = str(tmp_path / f"{tag}.npz")
        code = _DEBUG_CHILD.format(root=ROOT, tests=os.path.join(ROOT, "tests"), out=path)
        r = subprocess.run([_sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        outs[tag] = (np.load(path), r.stdout)
    line = [ln for ln in outs["debug"][1].splitlines() if ln.startswith("LANES_CHECK")]
    assert line, "the debug library does not export lcfe_debug_lanes_check"
    counts = eval(line[0].split(" ", 1)[1])
    assert counts[0] == 0 and counts[1] == 0, f"out-of-range LDS indices / damaged canaries: {counts}"
    for k in ("a", "b"):
        d, r = outs["debug"][0][k], outs["release"][0][k]
        diff = np.argwhere(~((d == r) | (np.isnan(d) & np.isnan(r))))
        if len(diff):
            with np.errstate(all="ignore"):
                rel = np.abs(d - r) / np.maximum(np.abs(r), 1e-300)
            print(f"set {k}: {len(diff)} entries differ between the debug and the release build, max rel {np.nanmax(rel):.3e}; first:",
                  [(int(i), COLUMNS["stat"][j], float(d[i, j]), float(r[i, j])) for i, j in diff[:6]],
                  "objects:", {int(i): int((diff[:, 0] == i).sum()) for i in np.unique(diff[:, 0])})
        assert len(diff) == 0, f"{k}: debug and release builds disagree in {len(diff)} entries"


_PLAN_CHILD = """
import sys
import numpy as np
sys.path.insert(0, {root!r})
from mallorn_astrophysics_amd import synth
from mallorn_astrophysics_amd.engine import extract_csr
lc = synth.make_lightcurves(1500, seed=77, n_max=700, n_median=150.0)
out, st = extract_csr(["bazin", "gp2d"], lc, z=lc["z"], return_status=True)
np.savez({out!r}, out=out, st=st)
"""


def test_gp_launch_plan_and_ticket_order_do_not_change_results(tmp_path):
    """The 2-D GP tiers serve their tickets longest light curve first (gp_sort_kernel) and are spread over the engine's
    streams by a launch plan (LCFE_GP_PLAN: an experiment knob, read once per process).  Objects are independent, so the
    outputs and status words must be the same bit for bit whatever the plan: default, every tier on one stream with
    the sets serialised, a plan with the tiers in another order and workgroup caps, three GP streams.  Light curves of up
    to 700 rows: all six tiers take part."""
    import subprocess
    import sys as _sys
    from conftest import ROOT
    base = {k: v for k, v in os.environ.items() if not k.startswith("LCFE_GP_") and k != "LCFE_SERIAL"}
    envs = {"default": base, "serial": dict(base, LCFE_SERIAL="1"),
            "plan": dict(base, LCFE_GP_PLAN="2:64,5,0|4:96,3,1"), "three": dict(base, LCFE_GP_STREAMS="3", LCFE_GP_PLAN="4|2,0,5|3,1")}
    res = {}
    for tag, env in envs.items():
        path = str(tmp_path / f"{tag}.npz")
        r = subprocess.run([_sys.executable, "-c", _PLAN_CHILD.format(root=ROOT, out=path)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        res[tag] = np.load(path)
    ref = res["default"]
    assert (ref["st"][:, 12 + 3] >= 512).any() and (ref["st"][:, 12 + 3] < 64).any(), "the sample must reach the first and the last tier"
    for tag in ("serial", "plan", "three"):
        o, r = res[tag]["out"], ref["out"]
        assert ((o == r) | (np.isnan(o) & np.isnan(r))).all(), f"{tag}: outputs differ from the default plan"
        assert np.array_equal(res[tag]["st"], ref["st"]), f"{tag}: status words differ from the default plan"


def test_special_values_fuzz():
    """NaN, +-inf, +-0, huge and tiny fluxes, NaN / inf / zero / negative errors, duplicated time stamps,
    constant bands and heavy ties, injected into seeded light curves: every streaming set must still agree
    with the oracle (tools/fuzz_special_values.py exits non-zero on any mismatch)."""
    import subprocess
    import sys as _sys
    from conftest import ROOT
    r = subprocess.run([_sys.executable, os.path.join(ROOT, "tools", "fuzz_special_values.py"), "240", "5"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_unsorted_rows_match(golden_inputs):
    """Shuffling the rows of every object must not change the statistics (sort paths on device)."""
    rng = np.random.default_rng(11)
    lc = synth.make_lightcurves(64, seed=77)
    sh = {k: v.copy() for k, v in lc.items()}
    for i in range(64):
        s, e = lc["offsets"][i], lc["offsets"][i + 1]
        p = rng.permutation(e - s) + s
        for k in ("t", "flux", "err", "band"):
            sh[k][s:e] = lc[k][p]
    a = extract_csr("stat", lc)
    b = extract_csr("stat", sh)
    ref = oracle.extract("stat", sh)
    bad = parity.compare(b, ref, COLUMNS["stat"], int_cols=STAT_INT_COLUMNS, rtol=1e-9, atol=1e-12)
    assert not bad, "\n".join(bad)
    bad = parity.compare(b, a, COLUMNS["stat"], int_cols=STAT_INT_COLUMNS, rtol=1e-9, atol=1e-12)
    assert not bad, "\n".join(bad)


def test_empty_batch_and_single_object():
    lc = synth.make_lightcurves(1, seed=2)
    got = extract_csr("stat", lc)
    assert got.shape == (1, 123)
    empty = {"offsets": np.zeros(1, np.int64), "t": np.zeros(0), "flux": np.zeros(0), "err": np.zeros(0),
             "band": np.zeros(0, np.uint8)}
    assert extract_csr("stat", empty).shape == (0, 123)


def test_dataframe_boundary_statistical(golden_inputs):
    """extract_statistical_features(DataFrame) -> DataFrame, as the reference's callers use it."""
    from mallorn_astrophysics_amd.features.statistical import extract_statistical_features
    ids = synth.object_ids(len(golden_inputs["offsets"]) - 1)
    df, meta = synth.to_dataframe(golden_inputs, ids)
    want = [ids[5], "absent_id", ids[0], ids[260]]
    out = extract_statistical_features(df, want)
    assert list(out["object_id"]) == [ids[5], ids[0], ids[260]]
    assert list(out.columns) == ["object_id"] + COLUMNS["stat"]
    for c in STAT_INT_COLUMNS:
        assert out[c].dtype == np.int64
    ref = load_golden("stat")[[5, 0, 260]]
    bad = parity.compare(out[COLUMNS["stat"]].to_numpy(float), ref, COLUMNS["stat"], int_cols=STAT_INT_COLUMNS,
                         rtol=1e-9, atol=1e-12)
    assert not bad, "\n".join(bad)


def test_device_resident_path_matches_host_path():
    import torch
    from mallorn_astrophysics_amd.engine import DeviceBatch
    lc = synth.make_lightcurves(500, seed=9)
    host = extract_csr("stat", lc)
    db = DeviceBatch(lc, z=lc["z"], device=0)
    out, _ = db.run("stat")
    torch.cuda.synchronize()
    assert np.array_equal(np.nan_to_num(out.cpu().numpy(), nan=-7.0), np.nan_to_num(host, nan=-7.0))


def test_understated_max_len_gives_nan_rows_not_garbage():
    """lcfe_extract_device trusts the caller's max_len to pick the tiers it launches; objects longer
    than that bound must come back as NaN rows with status -100, never as uninitialised memory."""
    import torch
    from mallorn_astrophysics_amd.engine import DeviceBatch
    rng = np.random.default_rng(2)
    objs = []
    for n in (40, 100, 128, 200, 300, 600):
        t = np.sort(59000 + rng.uniform(0, 400, n))
        objs.append((t, rng.normal(10, 3, n), np.full(n, 1.0), rng.choice(6, n)))
    lc = synth.from_objects(objs)
    full = extract_csr(["stat", "bazin", "gp2d"], lc)
    db = DeviceBatch(lc, device=0)
    db.max_len = 128                                   # a lie: three objects are longer
    ncol = full.shape[1]
    out = torch.full((6, ncol), 12345.0, dtype=torch.float64, device=db.device)
    out, st = db.run(["stat", "bazin", "gp2d"], out=out)
    torch.cuda.synchronize()
    got, st = out.cpu().numpy(), st.cpu().numpy()
    assert np.isnan(got[3:]).all()
    assert (st[3:] == -100).all()
    # the bound selects every tier up to the first capacity >= max_len, so the short objects are untouched
    assert np.array_equal(np.nan_to_num(got[:3], nan=-7.0), np.nan_to_num(full[:3], nan=-7.0))


@pytest.mark.parametrize("name", ["bazin", "powerlaw"])
def test_fits_golden(name, golden_inputs):
    """Bounded TRF fits on the device against the real reference (stability-aware rule)."""
    from conftest import check_fit_parity
    got, st = extract_csr(name, golden_inputs, z=golden_inputs["z"], return_status=True)
    check_fit_parity(got, name, COLUMNS[name])
    # the converged cost (chi^2 / R^2) and, for Bazin, status + nfev against golden_bazin.npz['nfev']
    from conftest import check_cost_parity
    check_cost_parity(got, st, name, golden_inputs)


def test_bazin_known_answer(golden_inputs):
    """SURVEY.md 8c known-answer fit (fixture object 260, r band)."""
    got = extract_csr("bazin", golden_inputs)[260]
    c = COLUMNS["bazin"]
    want = {"r_bazin_A": 38.47790522191582, "r_bazin_t0": 59011.29707849894,
            "r_bazin_tau_rise": 2.806862745741297, "r_bazin_tau_fall": 33.40674353526996,
            "r_bazin_B": 0.42573133782897704, "r_bazin_fit_chi2": 1.165819932699636,
            "r_bazin_rise_fall_ratio": 0.0840208402461331, "r_bazin_peak_flux": 38.9036365597448}
    for k, v in want.items():
        assert abs(got[c.index(k)] - v) <= 1e-4 * abs(v), (k, got[c.index(k)], v)
    pl = extract_csr("powerlaw", golden_inputs)[260]
    pc = COLUMNS["powerlaw"]
    for k, v in {"r_powerlaw_5_3_r2": 0.6489175767591367, "r_exponential_r2": 0.9947432001615515,
                 "r_linear_r2": 0.9038136480982363}.items():
        assert abs(pl[pc.index(k)] - v) <= 1e-4 * abs(v), (k, pl[pc.index(k)], v)


def test_multi_set_call_concatenates_columns(golden_inputs):
    a = extract_csr(["stat", "bazin"], golden_inputs)
    s = extract_csr("stat", golden_inputs)
    b = extract_csr("bazin", golden_inputs)
    assert a.shape[1] == 123 + 52
    assert np.array_equal(np.nan_to_num(a[:, :123], nan=-7), np.nan_to_num(s, nan=-7))
    assert np.array_equal(np.nan_to_num(a[:, 123:], nan=-7), np.nan_to_num(b, nan=-7))


def test_dataframe_boundary_all_extractors(golden_inputs):
    """Every drop-in extractor: signature, column order, object_id placement, values."""
    from mallorn_astrophysics_amd.features import (bazin_fitting, colors, lightcurve_shape, physics_based,
                                                   powerlaw, tde_physics)
    ids = synth.object_ids(len(golden_inputs["offsets"]) - 1)
    df, meta = synth.to_dataframe(golden_inputs, ids)
    want = ids[:12] + ids[-17:]
    rows = list(range(12)) + list(range(len(ids) - 17, len(ids)))
    from mallorn_astrophysics_amd.features import research_features
    cases = [("research", research_features.extract_research_features(df, want, meta, verbose=False)),
             ("tde", tde_physics.extract_tde_physics_features(df, want)),
             ("color", colors.extract_color_features(df, want)),
             ("shape", lightcurve_shape.extract_shape_features(df, want)),
             ("physics", physics_based.extract_physics_features(df, meta, want))]
    for name, out in cases:
        assert list(out.columns) == COLUMNS[name] + ["object_id"], name
        assert list(out["object_id"]) == want
        ref = load_golden(name)[rows]
        bad = parity.compare(out[COLUMNS[name]].to_numpy(float), ref, COLUMNS[name], label=name, **TOL[name])
        assert not bad, "\n".join(bad)
    bz = bazin_fitting.extract_bazin_features(df, want)
    assert list(bz.columns) == COLUMNS["bazin"] + ["object_id"]
    pw = powerlaw.extract_powerlaw_features(df, want)
    assert list(pw.columns) == ["object_id"] + COLUMNS["powerlaw"]


def test_gp2d_vs_oracle(golden_inputs):
    """2-D GP on the device vs the scipy-driven oracle.  PARITY UNPINNED: the reference uses george,
    which is absent; the oracle restates george's published algorithm (oracle/gp2d.py)."""
    from conftest import check_fit_parity, load_gp_oracle_fixture
    ref, probes = load_gp_oracle_fixture()
    got, st = extract_csr("gp2d", golden_inputs, return_status=True)
    # "stable" is judged from only two probe runs here, and exp(mean) (gp2d_amplitude) amplifies tiny
    # differences of the optimiser's end point: tolerate 2.5 % of the stable objects
    check_fit_parity(got, "gp2d", COLUMNS["gp2d"], ref=ref, probes=probes, max_stable_bad=6)
    assert (st[:, 3] == np.array([((golden_inputs["band"][a:b] < 6) & ~np.isnan(golden_inputs["flux"][a:b])
                                   & ~np.isnan(golden_inputs["err"][a:b]) & (golden_inputs["err"][a:b] > 0)).sum()
                                  for a, b in zip(golden_inputs["offsets"][:-1], golden_inputs["offsets"][1:])])).all()


def test_gp2d_long_objects_use_global_tier():
    """Gram-matrix tiers by length: LDS (60, 130), global scratch (191, 400, 600), and -- multiband_gp.py:66 only asks
    for N >= 10 -- the long-object tier for light curves of more than 767 rows (800, 1200: Gram matrix AND working set in
    global scratch).  2100 valid points are beyond it: NaN row, status -100."""
    rng = np.random.default_rng(3)
    objs = []
    for n in (60, 130, 191, 400, 600, 800, 1200, 2100):
        t = np.sort(59000 + rng.uniform(0, 300, n))
        b = rng.choice(6, n)
        f = 30 * np.exp(-0.5 * ((t - 59100) / 30) ** 2) * (1 + 0.1 * b) + rng.normal(0, 1, n)
        objs.append((t, f, np.full(n, 1.0), b))
    lc = synth.from_objects(objs)
    got, st = extract_csr("gp2d", lc, return_status=True)
    assert np.isnan(got[7]).all() and st[7, 0] == -100
    assert not np.isnan(got[:7, :3]).any()
    assert (st[:7, 3] == [60, 130, 191, 400, 600, 800, 1200]).all()
    keep = synth.from_objects(objs[:7])
    ref = oracle.extract("gp2d", keep)
    bad = parity.compare(got[:7], ref, COLUMNS["gp2d"], rtol=1e-4, atol=1e-9)
    assert len(bad) <= 2, "\n".join(bad)


def test_gp1d_vs_reference_golden(golden_inputs):
    """Per-band scikit-learn GP (set gp1d): device kernel against the outputs of the REAL reference module
    (tests/golden/golden_gp1d.npz), under the rule of the bounded fits (conftest.check_fit_parity): the fixture holds
    two probe runs of the real module with every flux moved by one ulp; a band fit the reference reproduces under both
    probes is "stable" and must match within 1e-4 (identical NaN mask), and over all fits the share within 1e-4 must
    reach the reference's own self-agreement minus 5 points.  The reference reproduces itself on 100 % of these fits,
    so the rule is strict here; the handful of listed exceptions are fits whose likelihood is flat in one direction
    (scipy stops at a 2e-9 relative decrease of the objective: the end point along such a direction is decided by the
    rounding of the Cholesky factorisation, which LAPACK and the device kernel do not share).  No value may be off by
    more than 2 %."""
    import os
    from conftest import ROOT, check_fit_parity
    from synth_subset import take
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_gp1d.npz"))
    sub = take(golden_inputs, g["pick"])
    got, st = extract_csr("gp1d", sub, return_status=True)
    ref = g["out"]
    assert (np.isnan(got) == np.isnan(ref)).all()
    summ = check_fit_parity(got, "gp1d", COLUMNS["gp1d"], max_stable_bad=GP1D_MAX_STABLE_BAD, ref=ref, probes=[g["out_p1"], g["out_p2"]])
    both = ~np.isnan(ref)
    rel = np.abs(got - ref)[both] / np.maximum(np.abs(ref[both]), 1e-8)
    print("gp1d share within 1e-4:", float((rel <= 1e-4).mean()), "max", float(rel.max()))
    assert rel.max() <= 0.02, rel.max()
    assert (st >= 0).all()


def test_gp1d_dataframe_boundary_and_long_bands(golden_inputs):
    """extract_gp_features(DataFrame) -> DataFrame with the reference's columns; a band with more than 159
    valid points takes the global-scratch matrix."""
    from mallorn_astrophysics_amd.features.gaussian_process import extract_gp_features
    from synth_subset import take
    sub = take(golden_inputs, np.arange(6))
    ids = synth.object_ids(6)
    df, meta = synth.to_dataframe(sub, ids)
    out = extract_gp_features(df, meta, ids, verbose=False)
    assert list(out.columns) == COLUMNS["gp1d"] + ["object_id"] and list(out["object_id"]) == ids
    rng = np.random.default_rng(8)
    t = np.sort(59000 + rng.uniform(0, 300, 260))
    b = np.r_[np.full(200, 2), np.full(30, 1), np.full(30, 3)]           # 200 rows in r
    f = 20 * np.exp(-0.5 * ((t - 59100) / 40) ** 2) + rng.normal(0, 1, 260)
    lc = synth.from_objects([(t, f, np.full(260, 1.0), b[rng.permutation(260)])])
    got, st = extract_csr("gp1d", lc, return_status=True)
    # the 200-point r band is beyond the 159-point LDS matrix: it is fitted on a matrix in global scratch (the
    # reference has no cap, gaussian_process.py:53-66) and must agree with scikit-learn like every other band
    assert (st[0] >= 0).all() and not np.isnan(got[0, 0:12]).any()
    ref = oracle.extract("gp1d", lc)
    both = ~np.isnan(ref[0])
    assert np.array_equal(np.isnan(got[0]), np.isnan(ref[0]))
    rel = np.abs(got[0] - ref[0])[both] / np.maximum(np.abs(ref[0][both]), 1e-8)
    assert rel.max() <= 0.02 and (rel <= 1e-4).mean() >= 0.8, (rel.max(), (rel <= 1e-4).mean())


def test_entry_point_scripts_write_the_caches(tmp_path):
    """scripts/precompute_features.py and scripts/cache_bazin_features.py on a synthetic data set
    laid out like the competition data; the pickles must have the layout the train_v*.py scripts read."""
    import pickle
    import subprocess
    import sys as _sys
    from conftest import ROOT
    from mallorn_astrophysics_amd.utils.data_loader import write_synthetic_dataset
    write_synthetic_dataset(tmp_path, n_train=24, n_test=30, seed=3)
    env = dict(os.environ, LCFE_DATA_ROOT=str(tmp_path))
    for script in ("precompute_features.py", "cache_bazin_features.py"):
        r = subprocess.run([_sys.executable, os.path.join(ROOT, "scripts", script)], env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
    proc = tmp_path / "data" / "processed"
    v4 = pickle.load(open(proc / "features_v4_cache.pkl", "rb"))
    assert set(v4) == {"train_features", "test_features"}
    assert v4["train_features"].shape == (24, 1 + 123 + 4 + 83 + 65 + 32)
    assert len(v4["test_features"]) == 30
    for name, ncol in (("tde_physics_cache.pkl", 25), ("multiband_gp_cache.pkl", 27), ("bazin_features_cache.pkl", 52),
                       ("enhanced_colors_cache.pkl", 83), ("gp_features_cache.pkl", 21),
                       ("research_features_cache.pkl", 40)):
        c = pickle.load(open(proc / name, "rb"))
        assert set(c) == {"train", "test"} and c["train"].shape == (24, ncol + 1), name
    pw = pickle.load(open(proc / "powerlaw_features.pkl", "rb"))
    assert pw.shape == (24, 28)
    # second run: every cache is skipped
    r = subprocess.run([_sys.executable, os.path.join(ROOT, "scripts", "precompute_features.py")], env=env,
                       capture_output=True, text=True)
    assert r.stdout.count("already cached") == 8


def test_full_size_properties():
    """BASELINE config sizes (10,178 objects): properties that do not need the oracle --
    run-to-run bit identity, independence of the rows from batch order and batch composition."""
    n = 10178
    lc = synth.make_lightcurves(n, seed=10178)
    sets = ["stat", "bazin", "powerlaw", "tde", "color", "shape", "physics", "gp2d"]
    a = extract_csr(sets, lc, z=lc["z"])
    b = extract_csr(sets, lc, z=lc["z"])
    assert a.shape == (n, 434)
    key = lambda m: np.nan_to_num(m, nan=-7.25e300)
    assert np.array_equal(key(a), key(b)), "two runs on the same input differ"
    # reversed object order -> the same rows, reversed
    import synth_subset
    rev = synth_subset.take(lc, list(range(n - 1, -1, -1)))
    c = extract_csr(sets, rev, z=rev["z"])
    assert np.array_equal(key(c[::-1]), key(a)), "rows depend on the position of the object in the batch"
    # a subset computed alone equals its rows in the full batch
    rows = list(range(0, n, 37))
    sub = synth_subset.take(lc, rows)
    d = extract_csr(sets, sub, z=sub["z"])
    assert np.array_equal(key(d), key(a[rows])), "rows depend on the other objects of the batch"
    # integer columns are integers, counts add up
    cols = COLUMNS["stat"]
    nobs = a[:, [cols.index(f"{p}_n_obs") for p in "ugrizy"]]
    assert np.array_equal(nobs, np.round(nobs))
    assert np.array_equal(nobs.sum(1), a[:, cols.index("all_n_obs")])
    assert np.array_equal(a[:, cols.index("all_n_obs")], np.diff(lc["offsets"]).astype(float))


def test_device_batch_validates_out_and_status_buffers():
    """DeviceBatch.run: wrong shape / dtype / device of a caller-supplied buffer raises instead of reaching a kernel."""
    import torch
    from mallorn_astrophysics_amd.engine import DeviceBatch
    lc = synth.make_lightcurves(20, seed=4)
    db = DeviceBatch(lc, z=lc["z"], device=0)
    with pytest.raises(ValueError):
        db.run("stat", out=torch.empty((20, 100), dtype=torch.float64, device=db.device))
    with pytest.raises(ValueError):
        db.run("stat", out=torch.empty((20, 123), dtype=torch.float32, device=db.device))
    with pytest.raises(ValueError):
        db.run("stat", out=torch.empty((20, 123), dtype=torch.float64))            # host tensor
    with pytest.raises(ValueError):
        db.run("bazin", status=torch.zeros((20, 11), dtype=torch.int32, device=db.device))
    with pytest.raises(ValueError):
        db.run("bazin", status=torch.zeros((20, 12), dtype=torch.int64, device=db.device))


def test_over_long_objects_are_reported_not_silent():
    """The DataFrame wrappers warn (count, cause and ids) about objects beyond the long-object tier (16384 rows): their
    NaN rows would otherwise look like failed fits (the reference has no limit at all)."""
    from mallorn_astrophysics_amd.features.bazin_fitting import extract_bazin_features
    rng = np.random.default_rng(12)
    objs = []
    for n in (60, 2100, 16500):              # 2100 rows: long-object tier; 16500: beyond it
        t = np.sort(59000 + rng.uniform(0, 400, n))
        objs.append((t, rng.normal(10, 3, n), np.full(n, 1.0), rng.choice(6, n)))
    lc = synth.from_objects(objs)
    df, _ = synth.to_dataframe(lc, ["short", "long", "huge"])
    with pytest.warns(RuntimeWarning, match="huge"):
        out = extract_bazin_features(df, ["short", "long", "huge"])
    vals = out[COLUMNS["bazin"]].to_numpy(float)
    assert np.isnan(vals[2]).all() and not np.isnan(vals[1]).all()


def test_bazin_long_light_curves_vs_oracle():
    """Light curves of more than 1024 rows: the fit-by-fit path only needs one BAND (up to 256 rows) in LDS; a light
    curve with a longer band, or of more than 2048 rows, is fitted by the long-object tier (object-level kernel, working
    set in global scratch) -- bazin_fitting.py:76-93 has no cap."""
    rng = np.random.default_rng(31)
    objs = []
    for n in (1100, 1530, 2046, 2600):       # <= 184, <= 256 and 341 / 434 rows per band: the last two exceed the 256-row fit tier -> long-object tier
        t = np.sort(59000 + rng.uniform(0, 600, n))
        b = rng.permutation(np.repeat(np.arange(6), n // 6 + 1)[:n])
        f = 40 * np.exp(-(t - 59200) / 60) / (1 + np.exp(-(t - 59200) / 8)) * (1 + 0.1 * b) + 3 + rng.normal(0, 1.5, n)
        objs.append((t, f, np.full(n, 1.5), b))
    lc = synth.from_objects(objs)
    got, st = extract_csr("bazin", lc, return_status=True)
    ref = oracle.extract("bazin", lc)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    both = ~np.isnan(ref)
    rel = np.abs(got - ref)[both] / np.maximum(np.abs(ref[both]), 1e-9)
    assert (rel <= 1e-4).mean() >= 0.9, (rel <= 1e-4).mean()
    assert (st[:, 0::2] > 0).all()
    # the decline fits of the same light curves (train_v55_powerlaw.py:161-163 has no cap either)
    gotp, stp = extract_csr("powerlaw", lc, return_status=True)
    refp = oracle.extract("powerlaw", lc)
    assert np.array_equal(np.isnan(gotp), np.isnan(refp))
    bothp = ~np.isnan(refp)
    relp = np.abs(gotp - refp)[bothp] / np.maximum(np.abs(refp[bothp]), 1e-9)
    assert (relp <= 1e-4).mean() >= 0.9, (relp <= 1e-4).mean()
    assert (stp != -100).all()


def test_all_ten_sets_in_one_call_equal_the_single_set_calls(golden_inputs):
    """One call with every feature set (all workspace regions in use at once: GP scratch slabs, Bazin and decline fit
    lists, the per-band GP's slab) must give, column block by column block, what the single-set calls give."""
    names = list(SET_NAMES)
    both = extract_csr(names, golden_inputs, z=golden_inputs["z"])
    col = 0
    for name in names:
        one = extract_csr(name, golden_inputs, z=golden_inputs["z"])
        blk = both[:, col:col + one.shape[1]]
        col += one.shape[1]
        assert np.array_equal(np.nan_to_num(blk, nan=-7.0), np.nan_to_num(one, nan=-7.0)), name
    assert col == both.shape[1] == sum(len(COLUMNS[n]) for n in names)


def test_sets_do_not_depend_on_the_statistics_set_being_in_the_mask():
    """The statistics set routes its light curves through lists of its own: the tier lists the other sets read are never
    appended to (round-2 advice: a 128-row-tier light curve the lanes kernels could not take used to be appended to the
    256-row tier's list, so every other set processed it twice whenever `stat` was in the mask, and the per-tier fit
    lists -- sized for one pass -- could overflow).  Worst case for the fit lists: every object has six fit-able bands
    of 5..32 rows (all fits in tier 0), a few light curves of up to 128 rows carry one 40-row band (not lane-eligible
    at 8 lanes), one is longer than 128 rows.  All sets in ONE call must equal the single-set calls bit for bit, and
    the streaming sets the oracle."""
    rng = np.random.default_rng(2026)
    objs = []

    def add(counts):
        n = int(sum(counts))
        t = np.sort(59000 + rng.uniform(0, 600, n))
        b = rng.permutation(np.repeat(np.arange(6), counts)).astype(np.int64)
        f = 25 * np.exp(-0.5 * ((t - 59250) / 45) ** 2) * (1 + 0.2 * b) + rng.normal(0, 1.0, n)
        objs.append((t, f, rng.uniform(0.4, 1.5, n), b))

    for _ in range(300):
        add(rng.integers(5, 22, 6))                       # six fit-able bands, <= 128 rows in all
    for _ in range(6):
        c = rng.integers(5, 16, 6)
        c[rng.integers(0, 6)] = 40                        # one 40-row band inside a <= 128-row light curve
        add(c)
    add([30, 30, 30, 30, 30, 30])                         # 180 rows: the 256-row tier exists in this batch
    order = rng.permutation(len(objs))
    lc = synth.from_objects([objs[i] for i in order])
    assert np.diff(lc["offsets"]).max() > 128
    names = ["stat", "bazin", "powerlaw", "tde", "color", "shape", "physics", "research"]
    got, status = extract_csr(names, lc, z=lc["z"], return_status=True)
    key = lambda m: np.nan_to_num(m, nan=-7.25e300)
    col0 = st0 = 0
    for name in names:
        one = extract_csr(name, lc, z=lc["z"], return_status=True)
        one, st = one if isinstance(one, tuple) else (one, None)
        ncol = one.shape[1]
        assert np.array_equal(key(got[:, col0:col0 + ncol]), key(one)), f"{name}: depends on the other sets of the mask"
        if st is not None and st.shape[1]:
            assert np.array_equal(status[:, st0:st0 + st.shape[1]], st), name
            st0 += st.shape[1]
        if name in TOL:
            ref = oracle.extract(name, lc, lc["z"])
            bad = parity.compare(one, ref, COLUMNS[name], int_cols=INT.get(name, ()), label=name, **TOL[name])
            assert not bad, "\n".join(bad)
        col0 += ncol
    # every band of 5+ rows was fitted exactly once: no Bazin column group is left unwritten
    baz = got[:, 123:123 + 48].reshape(len(order), 6, 8)
    stb, nfev = status[:, 0:12:2], status[:, 1:12:2]
    # (a fit that fails in scipy's prologue -- start point outside the bounds -- has a negative status and no evaluation;
    #  a fit that was queued but never run would show the zeros the status buffer starts with)
    assert ((nfev > 0) | (stb < 0)).all(), "a Bazin band fit was queued but never run"
    assert np.isfinite(baz[stb > 0]).all()


def test_extract_all_is_one_pack_one_engine_call_and_equals_the_extractors(golden_inputs):
    """``features.extract_all``: every requested set from ONE pack_lightcurves and ONE lcfe_extract(mask); its frames
    must equal, value for value and dtype for dtype, what the single extractors return (SURVEY.md §8b conventions:
    id first for statistics / power-law, last elsewhere; int64 count columns)."""
    from unittest import mock

    from mallorn_astrophysics_amd import features, packing
    from mallorn_astrophysics_amd.features import _frame
    from mallorn_astrophysics_amd.features.statistical import extract_statistical_features
    from mallorn_astrophysics_amd.features.colors import extract_color_features
    from mallorn_astrophysics_amd.features.bazin_fitting import extract_bazin_features
    from mallorn_astrophysics_amd.features.physics_based import extract_physics_features
    from mallorn_astrophysics_amd.features.powerlaw import extract_powerlaw_features
    lc = synth.make_lightcurves(120, seed=99)
    ids = synth.object_ids(120, prefix="obj")
    df, meta = synth.to_dataframe(lc, ids)
    want = ids[::-1][:100] + ["absent"]
    calls = {"pack": 0, "extract": 0}
    real_pack, real_extract = _frame.pack_lightcurves, _frame.extract_csr

    def counting_pack(*a, **k):
        calls["pack"] += 1
        return real_pack(*a, **k)

    def counting_extract(*a, **k):
        calls["extract"] += 1
        return real_extract(*a, **k)

    names = ["stat", "bazin", "powerlaw", "color", "physics"]
    with mock.patch.object(_frame, "pack_lightcurves", counting_pack), mock.patch.object(_frame, "extract_csr", counting_extract):
        frames = features.extract_all(df, metadata=meta, object_ids=want, sets=names)
    assert calls == {"pack": 1, "extract": 1}
    assert list(frames) == names
    singles = {"stat": extract_statistical_features(df, want), "bazin": extract_bazin_features(df, want),
               "powerlaw": extract_powerlaw_features(df, want), "color": extract_color_features(df, want),
               "physics": extract_physics_features(df, meta, want)}
    for name in names:
        a, b = frames[name], singles[name]
        assert list(a.columns) == list(b.columns) and list(a.dtypes) == list(b.dtypes), name
        assert a["object_id"].tolist() == want[:100], name
        for c in a.columns:
            if c == "object_id":
                continue
            assert np.array_equal(np.nan_to_num(a[c].to_numpy(float), nan=-7.0), np.nan_to_num(b[c].to_numpy(float), nan=-7.0)), (name, c)
    # the same batch handed over as a CSR (what the C++ CSV reader returns): no pack at all
    csr0, ids0 = packing.pack_lightcurves(df)
    frames2 = features.extract_all(csr=(csr0, ids0), metadata=meta, object_ids=want, sets=names)
    for name in names:
        assert frames2[name].equals(frames[name]), name


def test_multi_gpu_launcher_two_ranks_match_one_process():
    """``dist.extract_multi_gpu``: child ranks started with torch.distributed.run, batch handed over through
    memory-mapped files, cost-balanced contiguous shards, one gather.  On this one-GPU box the two ranks share the card
    and gather through gloo (host tensors); on a node with a GPU per rank the backend is nccl = RCCL.  The matrix and
    the status words must equal the single-process call bit for bit."""
    from mallorn_astrophysics_amd.dist import extract_multi_gpu
    lc = synth.make_lightcurves(400, seed=31)
    names = ["stat", "bazin", "powerlaw", "tde", "color", "shape", "physics", "gp2d"]
    one, st_one = extract_csr(names, lc, z=lc["z"], return_status=True)
    two, st_two = extract_multi_gpu(names, lc, lc["z"], ngpu=2, backend="gloo", timeout_s=600)
    key = lambda m: np.nan_to_num(m, nan=-7.25e300)
    assert two.shape == one.shape and np.array_equal(key(two), key(one))
    assert np.array_equal(st_two, st_one)
    # a mask without status words
    two, st_two = extract_multi_gpu(["stat", "color"], lc, None, ngpu=2, backend="gloo", timeout_s=600)
    assert st_two is None and np.array_equal(key(two), key(extract_csr(["stat", "color"], lc)))
