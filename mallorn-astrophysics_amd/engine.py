"""Host side of the C-ABI: CSR arrays in, feature matrices out.

``extract_csr`` is the host-buffer path the drop-in ``extract_*_features`` wrappers use;
``DeviceBatch`` keeps a CSR batch resident in HBM (torch tensors are only the allocator /
stream / RCCL plumbing) for the benchmark and the multi-GPU path.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from .columns import COLUMNS, SET_NAMES
from .packing import check_csr


def mask_of(sets) -> int:
    if isinstance(sets, int):
        return sets
    if isinstance(sets, str):
        sets = [sets]
    m = 0
    for s in sets:
        m |= 1 << SET_NAMES.index(s)
    return m


def sets_of(mask: int):
    return [SET_NAMES[i] for i in range(len(SET_NAMES)) if mask >> i & 1]


def columns_of(mask: int):
    cols = []
    for s in sets_of(mask):
        cols += COLUMNS[s]
    return cols


def _ptr(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


def extract_csr(sets, csr, z=None, device=-1, return_status=False, return_prof=False):
    """Run feature sets over a CSR batch (host numpy arrays) -> float64[n_obj, ncols]."""
    lib = _lib.load()
    mask = mask_of(sets)
    n_obj, total = check_csr(csr)
    ncol = lib.lcfe_ncols(mask)
    nst = lib.lcfe_nstatus(mask)
    out = np.full((n_obj, ncol), np.nan)
    status = np.zeros((n_obj, nst), np.int32) if nst else None
    zz = None
    if z is not None:
        zz = np.ascontiguousarray(z, np.float64)
        if zz.shape != (n_obj,):
            raise ValueError("z must have one entry per object")
    prof = _lib.LcfeStats()
    if n_obj:
        rc = lib.lcfe_extract(mask, device, n_obj, _ptr(csr["offsets"], _lib.c_i64p), _ptr(csr["t"], _lib.c_f64p),
                              _ptr(csr["flux"], _lib.c_f64p), _ptr(csr["err"], _lib.c_f64p),
                              _ptr(csr["band"], _lib.c_u8p), _ptr(zz, _lib.c_f64p), _ptr(out, _lib.c_f64p),
                              _ptr(status, _lib.c_i32p), ctypes.byref(prof))
        _lib.check(rc, "lcfe_extract")
    res = [out]
    if return_status:
        res.append(status)
    if return_prof:
        res.append(_lib.stats_to_dict(prof))
    return res[0] if len(res) == 1 else tuple(res)


class DeviceBatch:
    """A CSR batch resident in HBM on one GPU (torch owns the allocations and the stream)."""

    def __init__(self, csr, z=None, device=None):
        import torch

        self.torch = torch
        n_obj, total = check_csr(csr)
        if z is not None:
            z = np.ascontiguousarray(z, np.float64)
            if z.shape != (n_obj,):                 # the physics kernels read z[i] for every object
                raise ValueError(f"z must have one entry per object: shape {z.shape}, expected ({n_obj},)")
        self.n_obj, self.n_points = n_obj, total
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.max_len = int(np.diff(csr["offsets"]).max()) if n_obj else 0
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        self.offsets = to(csr["offsets"])
        self.t, self.flux, self.err, self.band = to(csr["t"]), to(csr["flux"]), to(csr["err"]), to(csr["band"])
        self.z = None if z is None else to(z)
        self._ws = None

    def run(self, sets, out=None, status=None, prof=False):
        """Enqueue the kernels of ``sets`` on torch's current stream; returns (out, status[, prof])."""
        torch = self.torch
        lib = _lib.load()
        mask = mask_of(sets)
        ncol, nst = lib.lcfe_ncols(mask), lib.lcfe_nstatus(mask)
        if out is None:
            out = torch.empty((self.n_obj, ncol), dtype=torch.float64, device=self.device)
        if status is None and nst:
            status = torch.zeros((self.n_obj, nst), dtype=torch.int32, device=self.device)
        # a malformed buffer must never reach a kernel (an out-of-bounds access can reset the GPU)
        if not (out.is_contiguous() and tuple(out.shape) == (self.n_obj, ncol) and out.dtype == torch.float64
                and out.device == self.device):
            raise ValueError(f"out must be a contiguous float64 [{self.n_obj}, {ncol}] tensor on {self.device}")
        if nst and not (status.is_contiguous() and tuple(status.shape) == (self.n_obj, nst)
                        and status.dtype == torch.int32 and status.device == self.device):
            raise ValueError(f"status must be a contiguous int32 [{self.n_obj}, {nst}] tensor on {self.device}")
        wsb = lib.lcfe_workspace_bytes_for(mask, self.n_obj, self.n_points, self.max_len)
        if self._ws is None or self._ws.numel() < wsb:
            self._ws = torch.empty(int(wsb), dtype=torch.uint8, device=self.device)
        st = _lib.LcfeStats()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        p = lambda x: None if x is None else ctypes.c_void_p(x.data_ptr())
        rc = lib.lcfe_extract_device(mask, self.device.index, ctypes.c_void_p(stream), self.n_obj, self.n_points,
                                     self.max_len, p(self.offsets), p(self.t), p(self.flux), p(self.err),
                                     p(self.band), p(self.z), p(out), p(status), p(self._ws), wsb,
                                     ctypes.byref(st) if prof else None)
        _lib.check(rc, "lcfe_extract_device")
        if prof:
            return out, status, _lib.stats_to_dict(st)
        return out, status
