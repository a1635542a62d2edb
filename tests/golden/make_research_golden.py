"""Golden fixture of the v115 research features: runs the REAL reference module
``src/features/research_features.py`` (imported unchanged from the read-only checkout) on the golden inputs.

    python tests/golden/make_research_golden.py [/root/reference]

Stores the 40 output columns (reference order) -> tests/golden/golden_research.npz.  Objects whose r band (g band when r has
fewer than 3 rows) holds only NaN fluxes make the reference raise (``Series.idxmax`` of an all-NaN column,
research_features.py:275-279); the fixture set has none.
"""
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


def main():
    from features import research_features as ref

    g = np.load(os.path.join(HERE, "golden_inputs.npz"))
    lc = {k: g[k] for k in g.files}
    ids = synth.object_ids(len(lc["offsets"]) - 1)
    df, meta = synth.to_dataframe(lc, ids)
    warnings.simplefilter("ignore")
    frame = ref.extract_research_features(df, ids, meta, verbose=False)
    assert list(frame["object_id"]) == ids
    cols = [c for c in frame.columns if c != "object_id"]
    assert cols == COLUMNS["research"], (cols, COLUMNS["research"])
    assert list(frame.columns)[-1] == "object_id"
    out = frame[cols].to_numpy(np.float64)
    np.savez_compressed(os.path.join(HERE, "golden_research.npz"), out=out)
    print("research", out.shape, "nan frac", np.isnan(out).mean().round(3))


if __name__ == "__main__":
    main()
