"""Generate the golden fixtures by running the REAL reference extractors.

Run in the build container only (needs the read-only reference checkout):

    python tests/golden/make_golden.py [/root/reference]

It imports ``src/features/{statistical,colors,lightcurve_shape,physics_based,tde_physics,
bazin_fitting}.py`` unchanged, and executes lines 106-202 of
``scripts/train_v55_powerlaw.py`` (the self-contained decline-model block; the script itself
loads competition data at import and cannot be imported) read from the checkout at run time.
Only arrays (synthetic inputs + the reference's outputs) and column names are written:

    tests/golden/golden_inputs.npz   CSR arrays + Z of the fixture objects
    tests/golden/golden_<set>.npz    reference output matrix [n_obj, ncols] (+ per-fit nfev)
    tests/golden/columns.json        column order as emitted by the reference

The 2-D GP (``multiband_gp.py``) needs ``george``, which is not installed: no GP fixture is
produced and GP parity is documented as unpinned (oracle/gp2d.py).
"""
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.path.insert(0, os.path.join(REF, "src"))

from mallorn_astrophysics_amd import synth  # noqa: E402
from mallorn_astrophysics_amd.columns import COLUMNS  # noqa: E402


def fixture_set():
    a = synth.make_lightcurves(200, seed=34)
    b = synth.make_lightcurves(60, seed=35, t_span=300.0, n_median=160.0)
    c = synth.edge_cases()
    return synth.concat([a, b, c])


def main():
    import pandas as pd
    from features import (bazin_fitting, colors, lightcurve_shape, physics_based, statistical,
                          tde_physics)

    lc = fixture_set()
    ids = synth.object_ids(len(lc["offsets"]) - 1)
    df, meta = synth.to_dataframe(lc, ids)
    np.savez_compressed(os.path.join(HERE, "golden_inputs.npz"), offsets=lc["offsets"], t=lc["t"],
                        flux=lc["flux"], err=lc["err"], band=lc["band"], z=lc["z"])
    warnings.simplefilter("ignore")
    cols_seen = {}

    def save(name, frame, **extra):
        assert list(frame["object_id"]) == ids, name
        cols = [c for c in frame.columns if c != "object_id"]
        cols_seen[name] = cols
        assert set(cols) == set(COLUMNS[name]), (name, set(cols) ^ set(COLUMNS[name]))
        mat = frame[COLUMNS[name]].to_numpy(dtype=np.float64)
        np.savez_compressed(os.path.join(HERE, f"golden_{name}.npz"), out=mat, **extra)
        print(name, mat.shape, "nan frac", np.isnan(mat).mean().round(3), flush=True)

    save("stat", statistical.extract_statistical_features(df, ids))
    save("color", colors.extract_color_features(df, ids))
    save("shape", lightcurve_shape.extract_shape_features(df, ids))
    save("physics", physics_based.extract_physics_features(df, meta, ids))
    save("tde", tde_physics.extract_tde_physics_features(df, ids))

    # Bazin: also record nfev per fit by wrapping curve_fit (diagnostic only)
    nfev_log = []
    real_cf = bazin_fitting.curve_fit

    def cf(*a, **k):
        try:
            r = real_cf(*a, full_output=True, **k)
        except Exception:
            nfev_log.append(-1)
            raise
        nfev_log.append(r[2]["nfev"])
        return r[0], r[1]
    bazin_fitting.curve_fit = cf
    save("bazin", bazin_fitting.extract_bazin_features(df, ids), nfev=np.array(nfev_log))
    bazin_fitting.curve_fit = real_cf

    # power-law block: executed from the reference script text, never copied
    src = open(os.path.join(REF, "scripts", "train_v55_powerlaw.py")).read().splitlines()
    block = "\n".join(src[105:202])
    ns = {"np": np, "pd": pd}
    exec(compile(block, "train_v55_powerlaw.py[106:202]", "exec"), ns)
    grouped = {k: g for k, g in df.groupby("object_id")}
    rows = [ns["extract_powerlaw_features"](i, grouped[i]) for i in ids]
    save("powerlaw", pd.DataFrame(rows))

    with open(os.path.join(HERE, "columns.json"), "w") as f:
        json.dump(cols_seen, f, indent=0)


if __name__ == "__main__":
    main()
