"""Helpers shared by the oracle modules: per-object views of a CSR batch."""
import numpy as np

BANDS = "ugrizy"


class Obj:
    """One object's rows in file order, with cached per-band (optionally time-sorted) views."""

    __slots__ = ("t", "f", "e", "b", "z", "_bands", "_sorted")

    def __init__(self, t, f, e, b, z=np.nan):
        self.t, self.f, self.e, self.b, self.z = t, f, e, b, z
        self._bands = {}
        self._sorted = {}

    def band(self, k):
        """(t, f, e) of band ``k`` (0..5) in file order == ``obj_lc[obj_lc.Filter == band]``."""
        v = self._bands.get(k)
        if v is None:
            m = self.b == k
            v = (self.t[m], self.f[m], self.e[m])
            self._bands[k] = v
        return v

    def band_sorted(self, k):
        """Band ``k`` sorted by time == ``.sort_values('Time (MJD)')``.

        pandas' default quicksort leaves the order of equal time stamps undefined; the build
        rule (SURVEY.md §8a traps) is a stable sort, i.e. ties keep file order.
        """
        v = self._sorted.get(k)
        if v is None:
            t, f, e = self.band(k)
            o = np.argsort(t, kind="stable")
            v = (t[o], f[o], e[o])
            self._sorted[k] = v
        return v


def iter_objects(csr, z=None):
    off = csr["offsets"]
    t, f, e, b = csr["t"], csr["flux"], csr["err"], csr["band"]
    for i in range(len(off) - 1):
        s, q = off[i], off[i + 1]
        yield Obj(t[s:q], f[s:q], e[s:q], b[s:q], np.nan if z is None else z[i])


def polyfit1(x, y):
    """``np.polyfit(x, y, 1)`` -- kept as the numpy call so the oracle has numpy's arithmetic."""
    return np.polyfit(x, y, 1)
