"""Import shims that let the reference's strategy scripts run UNCHANGED on this stack:

    python -m vdx.compat.run Distribution/strategies/fsdp_chunked_coherent.py --mode hybrid_ctx ...

`install()` registers, under the names the reference imports at module scope
(`fsdp_chunked_coherent.py:15-16,22`), stand-ins for three packages this image does not have:

  diffusers  -> vdx.compat.diffusers_shim   DiffusionPipeline.from_pretrained(...) returning an object with assignable
                                            .unet .text_encoder .vae .tokenizer .scheduler built from the HIP modules
  pynvml     -> vdx.compat.pynvml_shim      nvmlInit / nvmlDeviceGetHandleByIndex / nvmlDeviceGetMemoryInfo(h).used
  cv2        -> vdx.compat.cv2_shim         cvtColor, calcOpticalFlowFarneback, remap, VideoWriter(_fourcc), constants

A real installation of any of the three wins: a shim is only registered when the import fails.
"""
from __future__ import annotations

import importlib
import sys

_SHIMS = {"diffusers": "diffusers_shim", "pynvml": "pynvml_shim", "cv2": "cv2_shim"}


def install(force: bool = False):
    """Register the shims in sys.modules; returns the list of names that were shimmed."""
    done = []
    for name, mod in _SHIMS.items():
        if not force:
            try:
                importlib.import_module(name)
                continue
            except ImportError:
                pass
        sys.modules[name] = importlib.import_module(f"{__name__}.{mod}")
        done.append(name)
    return done
