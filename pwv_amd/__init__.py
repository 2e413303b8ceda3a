"""Importable name for the package directory ``parallel-wavenet-vocoder_amd/`` (a hyphen is not
legal in a Python module name).  ``import pwv_amd.modules`` resolves to
``parallel-wavenet-vocoder_amd/modules.py``."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'parallel-wavenet-vocoder_amd')
__path__.insert(0, _real)
with open(_os.path.join(_real, '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(_real, '__init__.py'), 'exec'))
