"""CSV files -> CSR batch through liblcfe_ingest.so (include/lcfe_ingest.h): the fast host ingest.

``read_lightcurves_csr(paths)`` returns what ``pack_lightcurves(pd.concat(map(pd.read_csv, paths)))``
returns -- the same arrays bit for bit (pandas' own float conversion is restated in the library),
objects in order of first appearance, rows of an object in file order -- without building the
DataFrame (reference: src/utils/data_loader.py:36-62 + the groupby of statistical.py:155).
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_LIB = None


def _load():
    global _LIB
    if _LIB is None:
        path = os.environ.get("LCFE_INGEST_LIB_PATH") or os.path.join(
            os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc", "liblcfe_ingest.so")
        if not os.path.exists(path):
            raise RuntimeError(f"liblcfe_ingest.so not found at {path}: run `make -C mallorn-astrophysics_amd/csrc`")
        lib = ctypes.CDLL(path)
        lib.lcfe_csv_open.restype = ctypes.c_void_p
        lib.lcfe_csv_open.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.c_int, ctypes.c_int]
        lib.lcfe_csv_close.argtypes = [ctypes.c_void_p]
        for name in ("lcfe_csv_n_objects", "lcfe_csv_n_rows", "lcfe_csv_id_bytes"):
            getattr(lib, name).restype = ctypes.c_int64
            getattr(lib, name).argtypes = [ctypes.c_void_p]
        lib.lcfe_csv_fill.restype = ctypes.c_int
        lib.lcfe_csv_fill.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 7
        lib.lcfe_csv_parse_double.restype = ctypes.c_int
        lib.lcfe_csv_parse_double.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_double)]
        lib.lcfe_ingest_last_error.restype = ctypes.c_char_p
        _LIB = lib
    return _LIB


def parse_double(text: str) -> float:
    """One CSV field through the library's number conversion (raises ValueError if not a number)."""
    lib = _load()
    raw = text.encode()
    out = ctypes.c_double()
    if lib.lcfe_csv_parse_double(raw, len(raw), ctypes.byref(out)):
        raise ValueError(f"not a number: {text!r}")
    return out.value


def read_lightcurves_csr(paths, n_threads: int = 0):
    """Parse light-curve CSV files -> ``(csr, ids)``; ``csr`` as ``packing.pack_lightcurves`` returns it."""
    lib = _load()
    paths = [os.fspath(p) for p in paths]
    arr = (ctypes.c_char_p * len(paths))(*[p.encode() for p in paths])
    h = lib.lcfe_csv_open(arr, len(paths), int(n_threads))
    if not h:
        raise RuntimeError(lib.lcfe_ingest_last_error().decode())
    try:
        n_obj, n_rows, nb = lib.lcfe_csv_n_objects(h), lib.lcfe_csv_n_rows(h), lib.lcfe_csv_id_bytes(h)
        csr = {"offsets": np.zeros(n_obj + 1, np.int64), "t": np.empty(n_rows), "flux": np.empty(n_rows),
               "err": np.empty(n_rows), "band": np.empty(n_rows, np.uint8)}
        id_off = np.zeros(n_obj + 1, np.int64)
        id_bytes = np.empty(max(nb, 1), np.uint8)
        ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        if lib.lcfe_csv_fill(h, ptr(csr["offsets"]), ptr(csr["t"]), ptr(csr["flux"]), ptr(csr["err"]), ptr(csr["band"]),
                             ptr(id_off), ptr(id_bytes)):
            raise RuntimeError(lib.lcfe_ingest_last_error().decode())
    finally:
        lib.lcfe_csv_close(h)
    raw = id_bytes.tobytes()
    ids = [raw[id_off[k]:id_off[k + 1]].decode() for k in range(n_obj)]
    return csr, ids
