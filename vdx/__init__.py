"""`vdx` — canonical import name of the product package whose sources live in
`decentralised-verification-and-distributed-execution-of-large-scale-video-diffusion-models_amd/`
(a directory name mandated by the build layout that is not a valid Python identifier).
`vdx.__path__` points at that directory, so `vdx.ops`, `vdx.unet3d`, ... are its modules."""
import os as _os

_REAL = _os.path.join(
    _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
    "decentralised-verification-and-distributed-execution-of-large-scale-video-diffusion-models_amd")
__path__ = [_REAL]

from . import _lib  # noqa: E402,F401

__all__ = ["_lib"]
