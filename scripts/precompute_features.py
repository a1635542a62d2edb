"""Pre-compute the cached feature frames the train_v*.py scripts read.

Same entry point as the reference's ``scripts/precompute_features.py`` (colour cache first,
"already cached" skip logic, ``{'train', 'test'}`` pickles).  The reference script only writes
``enhanced_colors_cache.pkl`` and the per-band sklearn GP cache; the other frames of the v34a / v55
feature matrix are built inline by other scripts (SURVEY.md finding 5).  This version writes all
of them in one go, every frame computed on the MI355X:

    enhanced_colors_cache.pkl   colors.extract_color_features            (precompute_features.py:40-56)
    features_v4_cache.pkl       statistics + metadata + colours + shape + physics merged
                                {'train_features', 'test_features'}      (train_v4_physics.py:53-109)
    tde_physics_cache.pkl       tde_physics.extract_tde_physics_features (train_v7_tde_physics.py:79-99)
    multiband_gp_cache.pkl      multiband_gp.extract_multiband_gp_features (train_v19_multiband_gp.py:92-110)
    bazin_features_cache.pkl    bazin_fitting.extract_bazin_features     (cache_bazin_features.py:39-45)
    powerlaw_features.pkl       decline-model R^2, train frame only      (visualize_and_powerlaw.py:373-375)

    gp_features_cache.pkl       gaussian_process.extract_gp_features (per-band scikit-learn GP,
                                precompute_features.py:58-76) -- the cache the reference script itself writes
    research_features_cache.pkl research_features.extract_research_features (train_v113_research_lgbm.py:120-140)
"""
import pickle
import sys
from pathlib import Path

sys.stdout.reconfigure(line_buffering=True)
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from mallorn_astrophysics_amd.utils.data_loader import get_base_path, load_all_data  # noqa: E402
from mallorn_astrophysics_amd.features.statistical import extract_statistical_features, add_metadata_features  # noqa: E402
from mallorn_astrophysics_amd.features.colors import extract_color_features  # noqa: E402
from mallorn_astrophysics_amd.features.lightcurve_shape import extract_shape_features  # noqa: E402
from mallorn_astrophysics_amd.features.physics_based import extract_physics_features  # noqa: E402
from mallorn_astrophysics_amd.features.tde_physics import extract_tde_physics_features  # noqa: E402
from mallorn_astrophysics_amd.features.multiband_gp import extract_multiband_gp_features  # noqa: E402
from mallorn_astrophysics_amd.features.gaussian_process import extract_gp_features  # noqa: E402
from mallorn_astrophysics_amd.features.bazin_fitting import extract_bazin_features  # noqa: E402
from mallorn_astrophysics_amd.features.powerlaw import extract_powerlaw_features  # noqa: E402
from mallorn_astrophysics_amd.features.research_features import extract_research_features  # noqa: E402

print("=" * 60, flush=True)
print("Pre-computing feature caches (MI355X)", flush=True)
print("=" * 60, flush=True)

base_path = get_base_path()
proc = base_path / 'data/processed'
proc.mkdir(parents=True, exist_ok=True)

print("\n1. Loading data...", flush=True)
data = load_all_data()
train_lc, test_lc = data['train_lc'], data['test_lc']
train_meta, test_meta = data['train_meta'], data['test_meta']
train_ids = train_meta['object_id'].tolist()
test_ids = test_meta['object_id'].tolist()
print(f"   Train: {len(train_ids)} objects", flush=True)
print(f"   Test: {len(test_ids)} objects", flush=True)


def cached(name, build, keys=('train', 'test')):
    path = proc / name
    if path.exists():
        print(f"   {name}: already cached!", flush=True)
        return
    frames = build()
    with open(path, 'wb') as f:
        pickle.dump(dict(zip(keys, frames)) if keys else frames, f)
    n = len(frames[0].columns) - 1 if keys else len(frames.columns) - 1
    print(f"   Saved {name} ({n} features)", flush=True)


print("\n2. Computing enhanced color features...", flush=True)
cached('enhanced_colors_cache.pkl', lambda: (extract_color_features(train_lc, train_ids),
                                             extract_color_features(test_lc, test_ids)))


def base_frame(lc, meta, ids):
    # train_v4_physics.py:60-80: statistics (+ metadata) merged with colours, shapes and physics
    f = add_metadata_features(extract_statistical_features(lc, ids), meta)
    for other in (extract_color_features(lc, ids), extract_shape_features(lc, ids),
                  extract_physics_features(lc, meta, ids)):
        f = f.merge(other, on='object_id', how='left')
    return f


print("\n3. Computing the base frame (statistics + colours + shapes + physics)...", flush=True)
cached('features_v4_cache.pkl', lambda: (base_frame(train_lc, train_meta, train_ids),
                                         base_frame(test_lc, test_meta, test_ids)),
       keys=('train_features', 'test_features'))

print("\n4. Computing TDE physics features...", flush=True)
cached('tde_physics_cache.pkl', lambda: (extract_tde_physics_features(train_lc, train_ids),
                                         extract_tde_physics_features(test_lc, test_ids)))

print("\n5. Computing multi-band GP features...", flush=True)
cached('multiband_gp_cache.pkl', lambda: (extract_multiband_gp_features(train_lc, train_meta, train_ids),
                                          extract_multiband_gp_features(test_lc, test_meta, test_ids)))

print("\n5b. Computing per-band GP length-scale features...", flush=True)
cached('gp_features_cache.pkl', lambda: (extract_gp_features(train_lc, train_meta, train_ids, verbose=False),
                                         extract_gp_features(test_lc, test_meta, test_ids, verbose=False)))

print("\n6. Computing Bazin features...", flush=True)
cached('bazin_features_cache.pkl', lambda: (extract_bazin_features(train_lc, train_ids),
                                            extract_bazin_features(test_lc, test_ids)))

print("\n7. Computing power-law decline features (train frame, as the reference stores it)...", flush=True)
cached('powerlaw_features.pkl', lambda: extract_powerlaw_features(train_lc, train_ids), keys=None)

print("\n8. Computing v115 research features...", flush=True)
cached('research_features_cache.pkl', lambda: (extract_research_features(train_lc, train_ids, train_meta, verbose=False),
                                               extract_research_features(test_lc, test_ids, test_meta, verbose=False)))

print("\n" + "=" * 60, flush=True)
print("DONE! Features are now cached.", flush=True)
print("=" * 60, flush=True)
