"""Competition-data loading with the reference's function names and return shapes
(``src/utils/data_loader.py:20-88``).  The data root defaults to ``<repo>/data/raw`` and can be
redirected with ``LCFE_DATA_ROOT`` (tests and demos write synthetic CSVs there)."""
import os
from pathlib import Path
from typing import Optional, Tuple

import pandas as pd

LSST_BANDS = ["u", "g", "r", "i", "z", "y"]
BAND_WAVELENGTHS = {"u": 367.0, "g": 482.5, "r": 622.2, "i": 754.5, "z": 869.1, "y": 971.0}


def get_base_path() -> Path:
    root = os.environ.get("LCFE_DATA_ROOT")
    return Path(root) if root else Path(__file__).resolve().parents[2]


def get_data_path() -> Path:
    return get_base_path() / "data" / "raw"


def load_metadata(data_path: Optional[Path] = None) -> Tuple[pd.DataFrame, pd.DataFrame]:
    data_path = get_data_path() if data_path is None else Path(data_path)
    return pd.read_csv(data_path / "train_log.csv"), pd.read_csv(data_path / "test_log.csv")


def load_lightcurves(split: str = "train", data_path: Optional[Path] = None) -> pd.DataFrame:
    data_path = get_data_path() if data_path is None else Path(data_path)
    frames = []
    for i in range(1, 21):
        p = data_path / f"split_{i:02d}" / f"{split}_full_lightcurves.csv"
        if p.exists():
            frames.append(pd.read_csv(p))
    if not frames:
        raise FileNotFoundError(f"No {split} lightcurve files found")
    return pd.concat(frames, ignore_index=True)


def load_lightcurves_csr(split: str = "train", data_path: Optional[Path] = None, n_threads: int = 0):
    """The split files of ``load_lightcurves`` straight into the CSR batch the GPU path takes, without the
    DataFrame: ``(csr, object_ids)`` equal to ``pack_lightcurves(load_lightcurves(split))`` bit for bit
    (multi-threaded C++ reader, ``utils/ingest.py`` / ``include/lcfe_ingest.h``)."""
    from .ingest import read_lightcurves_csr

    data_path = get_data_path() if data_path is None else Path(data_path)
    paths = [data_path / f"split_{i:02d}" / f"{split}_full_lightcurves.csv" for i in range(1, 21)]
    paths = [p for p in paths if p.exists()]
    if not paths:
        raise FileNotFoundError(f"No {split} lightcurve files found")
    return read_lightcurves_csr(paths, n_threads=n_threads)


def load_all_data(data_path: Optional[Path] = None) -> dict:
    data_path = get_data_path() if data_path is None else Path(data_path)
    train_meta, test_meta = load_metadata(data_path)
    return {"train_meta": train_meta, "test_meta": test_meta,
            "train_lc": load_lightcurves("train", data_path), "test_lc": load_lightcurves("test", data_path)}


def write_synthetic_dataset(root, n_train=40, n_test=60, seed=0, n_splits=3):
    """Write a synthetic data set in the competition's directory layout (for demos and tests)."""
    import numpy as np

    from .. import synth

    raw = Path(root) / "data" / "raw"
    raw.mkdir(parents=True, exist_ok=True)
    for split, n, sd in (("train", n_train, seed), ("test", n_test, seed + 1)):
        lc = synth.make_lightcurves(n, seed=sd)
        ids = synth.object_ids(n, prefix=split)
        df, meta = synth.to_dataframe(lc, ids)
        if split == "train":
            meta["target"] = (np.arange(n) % 7 == 0).astype(int)
        meta.to_csv(raw / f"{split}_log.csv", index=False)
        part = np.array_split(np.arange(n), n_splits)
        for k, idx in enumerate(part, 1):
            d = raw / f"split_{k:02d}"
            d.mkdir(exist_ok=True)
            keep = df["object_id"].isin([ids[i] for i in idx])
            df[keep].to_csv(d / f"{split}_full_lightcurves.csv", index=False)
    (Path(root) / "data" / "processed").mkdir(parents=True, exist_ok=True)
