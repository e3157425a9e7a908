"""Importable alias for the package directory ``rumi-slam_amd/`` (a hyphen cannot be imported).

All code lives in ``rumi-slam_amd/``; this module only points ``__path__`` there and runs its
``__init__``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "rumi-slam_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f
