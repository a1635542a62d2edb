"""CPU oracle: a numpy/scipy restatement of the reference's feature extractors.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this directory; it is
used by ``tests/``, by ``__graft_entry__.smoke()`` and by ``bench.py``'s ``cpu_baseline``
leg, and only as the checker / the timed CPU baseline.

Each function cites the reference ``file:line`` it follows (paths relative to the
MALLORN-astrophysics checkout).  The restatement works on the CSR arrays the device consumes
(per object: ``t, flux, err`` float64 and ``band`` uint8 codes 0..5 = u,g,r,i,z,y in file
order) instead of pandas frames.

Pinning: ``tests/golden/make_golden.py`` ran the *real* reference modules (imported from
the read-only checkout, in the build container) on the synthetic fixture set and committed
inputs + outputs as ``tests/golden/golden_*.npz``; ``tests/test_oracle_golden.py`` checks
every oracle function against those vectors.  The bounded curve fits call
``scipy.optimize.curve_fit`` with the reference's exact arguments -- scipy is the
third-party library the reference's arithmetic lives in and is present on the GPU box.
The 2-D GP is the exception: the reference uses ``george`` (not installed, not vendored), so
``oracle/gp2d.py`` restates george's published algorithm and is **parity unpinned**.
"""
from .run import extract, NCOLS  # noqa: F401
