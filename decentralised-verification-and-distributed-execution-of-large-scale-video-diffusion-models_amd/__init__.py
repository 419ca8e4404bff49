"""MI355X-native hybrid FSDP + frame-chunked denoising path for latent video diffusion.

Drop-in for the hot path of `Distribution/strategies/fsdp_chunked_coherent.py` of the reference
(diffusers `UNet3DConditionModel` / `DDIMScheduler` call surface) — hand-written HIP kernels for
gfx950 behind a C-ABI (`include/vdx.h`, `libvdx_hip.so`), Python host code on PyTorch-ROCm.

This directory's name is fixed by the build contract and is not a Python identifier.  The
canonical import name is `vdx` (repo-root package whose `__path__` is this directory); importing
the directory name through importlib hands back that same package object, so there is exactly
one copy of every submodule.
"""
import os as _os
import sys as _sys

if __name__ != "vdx":
    _root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    if _root not in _sys.path:
        _sys.path.insert(0, _root)
    import vdx as _vdx
    _sys.modules[__name__] = _vdx
