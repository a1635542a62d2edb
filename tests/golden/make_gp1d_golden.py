"""Golden fixture of the per-band scikit-learn GP ("next" row): runs the REAL
reference module ``src/features/gaussian_process.py`` (imported unchanged from the read-only checkout)
on a subset of the golden inputs and stores inputs' object indices + the 21 output columns.

    python tests/golden/make_gp1d_golden.py [/root/reference]
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.path.insert(0, os.path.join(REF, "src"))

from mallorn_astrophysics_amd import synth  # noqa: E402
from synth_subset import take as take_objects  # noqa: E402


def main():
    from features import gaussian_process as ref
    import oracle.gp1d as orc

    g = np.load(os.path.join(HERE, "golden_inputs.npz"))
    lc = {k: g[k] for k in g.files}
    n = len(lc["offsets"]) - 1
    pick = np.r_[0:40, 200:215, n - 17:n]                 # seeded objects, long ones, the edge cases
    sub = take_objects(lc, pick)
    ids = synth.object_ids(len(pick))
    df, meta = synth.to_dataframe(sub, ids)
    warnings.simplefilter("ignore")
    frame = ref.extract_gp_features(df, meta, ids, verbose=False)
    assert list(frame["object_id"]) == ids
    cols = [c for c in frame.columns if c != "object_id"]
    assert cols == orc.COLUMNS, (cols, orc.COLUMNS)
    out = frame[orc.COLUMNS].to_numpy(np.float64)
    np.savez_compressed(os.path.join(HERE, "golden_gp1d.npz"), pick=pick, out=out)
    print("gp1d", out.shape, "nan frac", np.isnan(out).mean().round(3))


if __name__ == "__main__":
    main()
