"""Pre-compute the cached feature frames the train_v*.py scripts read.

Same entry point as the reference's ``scripts/precompute_features.py`` (colour cache first, "already cached" skip
logic, ``{'train', 'test'}`` pickles).  The reference script only writes ``enhanced_colors_cache.pkl`` and the per-band
sklearn GP cache; the other frames of the v34a / v55 feature matrix are built inline by other scripts (SURVEY.md
finding 5).  This version writes all of them, every frame computed on the MI355X:

    enhanced_colors_cache.pkl   colors.extract_color_features            (precompute_features.py:40-56)
    features_v4_cache.pkl       statistics + metadata + colours + shape + physics merged
                                {'train_features', 'test_features'}      (train_v4_physics.py:53-109)
    tde_physics_cache.pkl       tde_physics.extract_tde_physics_features (train_v7_tde_physics.py:79-99)
    multiband_gp_cache.pkl      multiband_gp.extract_multiband_gp_features (train_v19_multiband_gp.py:92-110)
    gp_features_cache.pkl       gaussian_process.extract_gp_features (per-band scikit-learn GP,
                                precompute_features.py:58-76) -- the cache the reference script itself writes
    bazin_features_cache.pkl    bazin_fitting.extract_bazin_features     (cache_bazin_features.py:39-45)
    powerlaw_features.pkl       decline-model R^2, train frame only      (visualize_and_powerlaw.py:373-375)
    research_features_cache.pkl research_features.extract_research_features (train_v113_research_lgbm.py:120-140)

Engine speed: per split the light curves are read ONCE (the C++ CSV reader straight into the CSR batch; ``--pandas``
reads them through ``pd.read_csv`` as the reference does), packed ONCE, and every feature set a missing cache needs is
computed by ONE engine call (``features.extract_all``): the sets run side by side on the engine's streams and the
batch crosses PCIe once.  ``--ngpu N`` (or ``LCFE_NGPU=N``) shards each split over N GPUs of the node: N child ranks are
started before any GPU call, objects are cut into cost-balanced contiguous shards and one RCCL gather brings the rows
back (``dist.extract_multi_gpu``); this process itself never touches a GPU then.
"""
import argparse
import os
import pickle
import sys
import time
from pathlib import Path

sys.stdout.reconfigure(line_buffering=True)
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from mallorn_astrophysics_amd.utils.data_loader import (get_base_path, load_lightcurves, load_lightcurves_csr,  # noqa: E402
                                                         load_metadata)
from mallorn_astrophysics_amd.features import extract_all  # noqa: E402
from mallorn_astrophysics_amd.features.statistical import add_metadata_features  # noqa: E402

# cache file -> (feature sets it is made of, pickle keys; None = bare train frame)
CACHES = {
    'enhanced_colors_cache.pkl': (('color',), ('train', 'test')),
    'features_v4_cache.pkl': (('stat', 'color', 'shape', 'physics'), ('train_features', 'test_features')),
    'tde_physics_cache.pkl': (('tde',), ('train', 'test')),
    'multiband_gp_cache.pkl': (('gp2d',), ('train', 'test')),
    'gp_features_cache.pkl': (('gp1d',), ('train', 'test')),
    'bazin_features_cache.pkl': (('bazin',), ('train', 'test')),
    'powerlaw_features.pkl': (('powerlaw',), None),
    'research_features_cache.pkl': (('research',), ('train', 'test')),
}


def base_frame(frames, meta):
    # train_v4_physics.py:60-80: statistics (+ metadata) merged with colours, shapes and physics
    f = add_metadata_features(frames['stat'], meta)
    for other in (frames['color'], frames['shape'], frames['physics']):
        f = f.merge(other, on='object_id', how='left')
    return f


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('--ngpu', type=int, default=int(os.environ.get('LCFE_NGPU', '1')),
                    help='GPUs of this node to shard every split over (child ranks, one per GPU)')
    ap.add_argument('--pandas', action='store_true', help='read the CSV files with pandas instead of the C++ reader')
    a = ap.parse_args(argv)

    print("=" * 60, flush=True)
    print(f"Pre-computing feature caches (MI355X, {a.ngpu} GPU{'s' if a.ngpu > 1 else ''})", flush=True)
    print("=" * 60, flush=True)
    proc = get_base_path() / 'data/processed'
    proc.mkdir(parents=True, exist_ok=True)

    missing = [name for name in CACHES if not (proc / name).exists()]
    for name in CACHES:
        if name not in missing:
            print(f"   {name}: already cached!", flush=True)
    if not missing:
        print("DONE! Features are now cached.", flush=True)
        return 0

    print("\n1. Loading data...", flush=True)
    meta = dict(zip(('train', 'test'), load_metadata()))
    print(f"   Train: {len(meta['train'])} objects", flush=True)
    print(f"   Test: {len(meta['test'])} objects", flush=True)

    frames = {}
    for split in ('train', 'test'):
        names = sorted({s for c in missing for s in CACHES[c][0] if split == 'train' or CACHES[c][1] is not None})
        if not names:
            continue
        ids = meta[split]['object_id'].tolist()
        t0 = time.perf_counter()
        if a.pandas:
            src = dict(lightcurves=load_lightcurves(split))
        else:
            src = dict(csr=load_lightcurves_csr(split))
        t1 = time.perf_counter()
        frames[split] = extract_all(metadata=meta[split], object_ids=ids, sets=names, ngpu=a.ngpu, **src)
        t2 = time.perf_counter()
        n = len(next(iter(frames[split].values())))
        print(f"\n2. {split}: {n} light curves, sets {'+'.join(names)}: read {t1 - t0:.2f} s, "
              f"features {t2 - t1:.2f} s ({n / max(t2 - t1, 1e-9):,.0f} light curves/s through the entry point)", flush=True)

    print("\n3. Writing the caches...", flush=True)
    for name in missing:
        sets, keys = CACHES[name]
        if name == 'features_v4_cache.pkl':
            value = {k: base_frame(frames[s], meta[s]) for k, s in zip(keys, ('train', 'test'))}
            ncol = len(value[keys[0]].columns) - 1
        elif keys is None:
            value = frames['train'][sets[0]]
            ncol = len(value.columns) - 1
        else:
            value = {k: frames[k][sets[0]] for k in keys}
            ncol = len(value[keys[0]].columns) - 1
        with open(proc / name, 'wb') as f:
            pickle.dump(value, f)
        print(f"   Saved {name} ({ncol} features)", flush=True)

    print("\n" + "=" * 60, flush=True)
    print("DONE! Features are now cached.", flush=True)
    print("=" * 60, flush=True)
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
