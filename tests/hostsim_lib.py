"""ctypes loader of tests/hostsim/libhostsim.so (built on demand with make)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        d = os.path.join(HERE, "hostsim")
        # HOSTSIM_SANITIZE=1: the AddressSanitizer/UBSan build (run pytest with LD_PRELOAD=libasan.so)
        name = "libhostsim_asan.so" if os.environ.get("HOSTSIM_SANITIZE") == "1" else "libhostsim.so"
        subprocess.run(["make", "-s", "-C", d, name], check=True)
        _lib = ctypes.CDLL(os.path.join(d, name))
        _lib.hostsim_extract.restype = ctypes.c_int
    return _lib


def extract(set_id, csr, z=None, ncol=None, nstatus=0):
    L = lib()
    n_obj = len(csr["offsets"]) - 1
    out = np.full((n_obj, ncol), np.nan)
    st = np.zeros((n_obj, max(nstatus, 1)), np.int32)
    p = lambda a, t: a.ctypes.data_as(ctypes.POINTER(t))
    zz = None if z is None else np.ascontiguousarray(z, np.float64)
    rc = L.hostsim_extract(
        ctypes.c_int(set_id), ctypes.c_int64(n_obj), p(np.ascontiguousarray(csr["offsets"], np.int64), ctypes.c_int64),
        p(csr["t"], ctypes.c_double), p(csr["flux"], ctypes.c_double), p(csr["err"], ctypes.c_double),
        p(csr["band"], ctypes.c_uint8), None if zz is None else p(zz, ctypes.c_double),
        p(out, ctypes.c_double), p(st, ctypes.c_int32))
    assert rc == 0
    return out, st


def gp1d(csr):
    """Per-band 1-D GP templates (csrc/gp1d.hpp) on the host -> (out[n_obj, 21], status[n_obj, 4])."""
    L = lib()
    n_obj = len(csr["offsets"]) - 1
    out = np.full((n_obj, 21), np.nan)
    st = np.zeros((n_obj, 4), np.int32)
    p = lambda a, t: a.ctypes.data_as(ctypes.POINTER(t))
    L.hostsim_gp1d.restype = ctypes.c_int
    rc = L.hostsim_gp1d(ctypes.c_int64(n_obj), p(np.ascontiguousarray(csr["offsets"], np.int64), ctypes.c_int64),
                        p(csr["t"], ctypes.c_double), p(csr["flux"], ctypes.c_double), p(csr["err"], ctypes.c_double),
                        p(csr["band"], ctypes.c_uint8), p(out, ctypes.c_double), p(st, ctypes.c_int32))
    assert rc == 0
    return out, st
