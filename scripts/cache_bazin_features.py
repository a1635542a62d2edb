"""Bazin feature cache on the MI355X.

Entry point with the role of the reference's ``scripts/cache_bazin_features.py`` (:16-45): it leaves
``data/processed/bazin_features_cache.pkl`` = ``{'train': frame, 'test': frame}`` for the ``train_v34a``
family of scripts, with the six-band fits done by liblcfe instead of a Python loop over
``scipy.optimize.curve_fit``.  Per split the CSV files go through the C++ reader straight into the CSR batch
(``--pandas``: ``pd.read_csv`` as the reference) and ONE engine call; ``--ngpu N`` / ``LCFE_NGPU`` shards a split over
N GPUs of the node (child ranks started before any GPU call, ``dist.extract_multi_gpu``).
"""
import argparse
import os
import sys
import time
from pathlib import Path

import pandas as pd

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mallorn_astrophysics_amd.features import extract_all  # noqa: E402
from mallorn_astrophysics_amd.utils.data_loader import (get_base_path, load_lightcurves, load_lightcurves_csr,  # noqa: E402
                                                         load_metadata)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--ngpu", type=int, default=int(os.environ.get("LCFE_NGPU", "1")))
    ap.add_argument("--pandas", action="store_true", help="read the CSV files with pandas instead of the C++ reader")
    a = ap.parse_args(argv)
    meta = dict(zip(("train", "test"), load_metadata()))
    frames = {}
    for split in ("train", "test"):
        ids = meta[split]["object_id"].tolist()
        t0 = time.perf_counter()
        src = dict(lightcurves=load_lightcurves(split)) if a.pandas else dict(csr=load_lightcurves_csr(split))
        t1 = time.perf_counter()
        frames[split] = extract_all(object_ids=ids, sets=["bazin"], ngpu=a.ngpu, **src)["bazin"]
        dt = time.perf_counter() - t1
        print(f"[bazin cache] {split}: {frames[split].shape[1] - 1} columns x {len(frames[split])} objects: read "
              f"{t1 - t0:.2f} s, fits {dt:.2f} s ({len(frames[split]) / max(dt, 1e-9):,.0f} light curves/s)", flush=True)
    target = get_base_path() / "data" / "processed" / "bazin_features_cache.pkl"
    target.parent.mkdir(parents=True, exist_ok=True)
    pd.to_pickle(frames, target)
    print(f"[bazin cache] wrote {target} ({target.stat().st_size / 1024:.1f} KiB)")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
