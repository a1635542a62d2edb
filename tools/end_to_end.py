"""End-to-end example on one GPU: competition-layout CSV files -> C++ ingest -> all feature sets on
the MI355X -> one DataFrame, with the wall time of every stage.  Usage: end_to_end.py [n_objects]"""
import os
import sys
import tempfile
import time

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mallorn_astrophysics_amd.columns import COLUMNS, SET_NAMES          # noqa: E402
from mallorn_astrophysics_amd.engine import extract_csr                   # noqa: E402
from mallorn_astrophysics_amd.packing import pack_lightcurves             # noqa: E402
from mallorn_astrophysics_amd.utils.data_loader import load_lightcurves, load_lightcurves_csr, write_synthetic_dataset  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as root:
    t0 = time.perf_counter()
    write_synthetic_dataset(root, n_train=8, n_test=n, seed=5, n_splits=4)
    raw = os.path.join(root, "data", "raw")
    print(f"wrote a synthetic test split of {n} objects in {time.perf_counter() - t0:.1f} s", flush=True)

    t0 = time.perf_counter()
    csr, ids = load_lightcurves_csr("test", raw)
    t_ingest = time.perf_counter() - t0
    sets = [s for s in SET_NAMES if s != "physics"]          # physics needs the metadata redshift column
    extract_csr(["stat"], csr)                                # first call: library + context start-up
    t0 = time.perf_counter()
    out = extract_csr(sets, csr)
    t_gpu = time.perf_counter() - t0
    cols = [c for s in sets for c in COLUMNS[s]]
    frame = pd.DataFrame(out, columns=cols)
    frame.insert(0, "object_id", ids)
    print(f"C++ ingest {t_ingest:.2f} s ({len(ids) / t_ingest:,.0f} light curves/s), GPU features {t_gpu:.2f} s "
          f"({len(ids) / t_gpu:,.0f} light curves/s incl. PCIe), frame {frame.shape}", flush=True)

    t0 = time.perf_counter()
    ref_csr, ref_ids = pack_lightcurves(load_lightcurves("test", raw))
    t_pandas = time.perf_counter() - t0
    same = all(np.array_equal(csr[k].view(np.uint8), ref_csr[k].view(np.uint8)) for k in ref_csr) and ids == [str(i) for i in ref_ids]
    print(f"pandas read_csv + concat + pack {t_pandas:.2f} s ({len(ids) / t_pandas:,.0f} light curves/s); identical CSR: {same}")
    assert same
