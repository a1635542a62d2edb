"""Importable alias for the ``mallorn-astrophysics_amd/`` package directory.

The package directory carries the repository's name (with a hyphen, which Python cannot
import).  This stub points ``__path__`` at that directory and runs its ``__init__``, so
``import mallorn_astrophysics_amd.features.statistical`` resolves into
``mallorn-astrophysics_amd/features/statistical.py``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "mallorn-astrophysics_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f
