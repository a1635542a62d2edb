"""Bazin feature cache on the MI355X.

Entry point with the role of the reference's ``scripts/cache_bazin_features.py``: it leaves
``data/processed/bazin_features_cache.pkl`` = ``{'train': frame, 'test': frame}`` for the ``train_v34a``
family of scripts, with the six-band fits done by liblcfe instead of a Python loop over
``scipy.optimize.curve_fit``.
"""
import sys
import time
from pathlib import Path

import pandas as pd

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mallorn_astrophysics_amd.features.bazin_fitting import extract_bazin_features  # noqa: E402
from mallorn_astrophysics_amd.utils.data_loader import get_base_path, load_all_data  # noqa: E402


def main():
    data = load_all_data()
    frames = {}
    for split in ("train", "test"):
        ids = data[f"{split}_meta"]["object_id"].tolist()
        t0 = time.perf_counter()
        frames[split] = extract_bazin_features(data[f"{split}_lc"], ids)
        print(f"[bazin cache] {split}: {frames[split].shape[1] - 1} columns x {len(frames[split])} objects "
              f"in {time.perf_counter() - t0:.2f} s", flush=True)
    target = get_base_path() / "data" / "processed" / "bazin_features_cache.pkl"
    target.parent.mkdir(parents=True, exist_ok=True)
    pd.to_pickle(frames, target)
    print(f"[bazin cache] wrote {target} ({target.stat().st_size / 1024:.1f} KiB)")


if __name__ == "__main__":
    main()
