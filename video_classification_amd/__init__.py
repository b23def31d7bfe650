"""Importable alias of the product package.

The package directory is ``video-classification_amd/`` (the repo's naming contract); a hyphen is not a legal
Python identifier, so this shim points ``video_classification_amd`` at that directory and runs its __init__."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "video-classification_amd")
__path__ = [_real]
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
del _os, _f, _real
