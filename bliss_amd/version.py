"""ref python/bliss/version.py:8 — `version()` returns bl_version()."""
from . import _lib


def version():
    return _lib.load().bl_version()
