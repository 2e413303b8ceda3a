"""MI355X-native IAF-WaveNet student generation path (andabi/parallel-wavenet-vocoder).

The directory name carries a hyphen (fixed by the project layout); import it as ``pwv_amd``
(the shim package at the repo root points here).
"""
from . import _lib  # noqa: F401
from .hparam import hparam  # noqa: F401

__all__ = ['hparam']
