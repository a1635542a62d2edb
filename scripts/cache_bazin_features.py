"""Precompute and cache Bazin parametric features for reuse.

Same entry point, prints and cache file as the reference's ``scripts/cache_bazin_features.py``
(``data/processed/bazin_features_cache.pkl`` = ``{'train': df, 'test': df}``), with the fits
running on the MI355X through liblcfe instead of a Python loop over ``scipy.curve_fit``.
"""
import sys
from pathlib import Path

import pandas as pd

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mallorn_astrophysics_amd.utils.data_loader import get_base_path, load_all_data  # noqa: E402
from mallorn_astrophysics_amd.features.bazin_fitting import extract_bazin_features  # noqa: E402

base_path = get_base_path()

print("Caching Bazin Features for v34a...")
print("=" * 80)

data = load_all_data()
train_ids = data['train_meta']['object_id'].tolist()
test_ids = data['test_meta']['object_id'].tolist()

print("\nExtracting Bazin features for training set...")
train_bazin = extract_bazin_features(data['train_lc'], train_ids)
print(f"   Extracted {len(train_bazin.columns)-1} features for {len(train_bazin)} objects")

print("\nExtracting Bazin features for test set...")
test_bazin = extract_bazin_features(data['test_lc'], test_ids)
print(f"   Extracted {len(test_bazin.columns)-1} features for {len(test_bazin)} objects")

cache_path = base_path / 'data/processed/bazin_features_cache.pkl'
cache_path.parent.mkdir(parents=True, exist_ok=True)
pd.to_pickle({'train': train_bazin, 'test': test_bazin}, cache_path)

print(f"\nBazin features cached to: {cache_path}")
print(f"Cache size: {cache_path.stat().st_size / 1024:.1f} KB")
print("=" * 80)
