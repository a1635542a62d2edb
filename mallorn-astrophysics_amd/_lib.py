"""ctypes binding of liblcfe.so (include/lcfe.h).  Fails loudly: there is no CPU fallback."""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LCFE_LIB_PATH") or os.path.join(_HERE, "csrc", "liblcfe.so")   # override: instrumented builds
NUM_SETS = 10

c_i64p = ctypes.POINTER(ctypes.c_int64)
c_f64p = ctypes.POINTER(ctypes.c_double)
c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_i32p = ctypes.POINTER(ctypes.c_int32)


class LcfeStats(ctypes.Structure):
    _fields_ = [("kernel_ms", ctypes.c_double * NUM_SETS), ("h2d_ms", ctypes.c_double),
                ("d2h_ms", ctypes.c_double), ("bytes_in", ctypes.c_int64), ("bytes_out", ctypes.c_int64),
                ("launches", ctypes.c_int32 * NUM_SETS), ("reserved", ctypes.c_int32)]


class LcfeError(RuntimeError):
    """Infrastructure failure (HIP error, bad argument) -- never a numerical fit failure."""


_lib = None

# The engine runs its feature sets on four streams (the caller's + three of its own).  HIP maps streams onto four
# hardware queues by default; in a process that holds more streams (RCCL's, a framework's) two of the engine's streams
# then share a queue and their kernels run one after the other (measured: 1.98 s instead of 1.81 s per benchmark pass
# with an RCCL communicator alive).  The variable is read when the HIP runtime starts, so it is set at import --
# harmless if the runtime is already up, and an explicit setting of the user's wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own ``libamdhip64.so.7``; ``liblcfe.so`` links the same
    SONAME from /opt/rocm.  Whichever is mapped first serves the whole process, and torch cannot
    see the GPU through a runtime it was not built with.  When torch is installed (it is the
    allocator / stream / RCCL plumbing of the device-resident path) map its runtime first, without
    paying for ``import torch``."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load liblcfe.so (built by ``__graft_entry__.build()`` / ``make -C csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LcfeError(f"{LIB_PATH} is missing: build it with `make -C {os.path.dirname(LIB_PATH)}` "
                        "(hipcc --offload-arch=gfx950); lcfe has no CPU fallback")
    _preload_torch_hip_runtime()
    lib = ctypes.CDLL(LIB_PATH)
    lib.lcfe_version.restype = ctypes.c_int
    lib.lcfe_device_count.restype = ctypes.c_int
    lib.lcfe_last_error.restype = ctypes.c_char_p
    lib.lcfe_ncols.restype = ctypes.c_int64
    lib.lcfe_ncols.argtypes = [ctypes.c_int]
    lib.lcfe_nstatus.restype = ctypes.c_int64
    lib.lcfe_nstatus.argtypes = [ctypes.c_int]
    lib.lcfe_colname.restype = ctypes.c_char_p
    lib.lcfe_colname.argtypes = [ctypes.c_int, ctypes.c_int64]
    lib.lcfe_max_points.restype = ctypes.c_int64
    lib.lcfe_gp2d_max_points.restype = ctypes.c_int64
    lib.lcfe_implemented_mask.restype = ctypes.c_int
    lib.lcfe_workspace_bytes.restype = ctypes.c_size_t
    lib.lcfe_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64]
    lib.lcfe_workspace_bytes_for.restype = ctypes.c_size_t
    lib.lcfe_workspace_bytes_for.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
    lib.lcfe_extract.restype = ctypes.c_int
    lib.lcfe_extract.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int64, c_i64p, c_f64p, c_f64p, c_f64p,
                                 c_u8p, c_f64p, c_f64p, c_i32p, ctypes.POINTER(LcfeStats)]
    lib.lcfe_extract_device.restype = ctypes.c_int
    lib.lcfe_extract_device.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
                                        ctypes.c_int64, ctypes.c_int64] + [ctypes.c_void_p] * 9 + \
                                       [ctypes.c_size_t, ctypes.POINTER(LcfeStats)]
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise LcfeError(f"{what}: {load().lcfe_last_error().decode()}")


def stats_to_dict(st: LcfeStats):
    return {"kernel_ms": list(st.kernel_ms), "h2d_ms": st.h2d_ms, "d2h_ms": st.d2h_ms,
            "bytes_in": st.bytes_in, "bytes_out": st.bytes_out, "launches": list(st.launches)}
