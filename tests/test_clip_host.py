"""CPU suite: CLIPTextModel weight ingest against the real dependency's module tree (transformers is installed in
this image: the text tower is the one operator family whose reference implementation can be imported)."""
import importlib.util
import os
import sys

import pytest
import torch

import vdx  # noqa: F401
from vdx._lib import VdxError
from vdx.clip_text import CLIPTextConfig, CLIPTextModel

transformers = pytest.importorskip("transformers")


def test_clip_ingest_covers_every_key_of_the_real_module():
    cfg = transformers.CLIPTextConfig(vocab_size=49408, hidden_size=1024, intermediate_size=4096, num_hidden_layers=23,
                                      num_attention_heads=16, max_position_embeddings=77, hidden_act="gelu",
                                      projection_dim=1024)
    with torch.device("meta"):
        ref = transformers.CLIPTextModel(cfg)
    sd = dict(ref.state_dict())
    n_ref = sum(v.numel() for v in sd.values() if v.dtype.is_floating_point)
    assert n_ref == 340_387_840                       # SD-2.x text tower as shipped (23 of OpenCLIP ViT-H's 24 layers)
    m = CLIPTextModel(CLIPTextConfig.sd2()).load_transformers_state_dict(sd, device="meta")
    assert m.num_parameters() == n_ref - 23 * 1024    # the value bias is folded behind the output projection
    # older transformers releases prefix the keys with `text_model.` — same table
    CLIPTextModel(CLIPTextConfig.sd2()).load_transformers_state_dict({"text_model." + k: v for k, v in sd.items()}, device="meta")
    sd["encoder.layers.0.bogus.weight"] = torch.empty(1, device="meta")
    with pytest.raises(VdxError):
        CLIPTextModel(CLIPTextConfig.sd2()).load_transformers_state_dict(sd, device="meta")


def test_clip_has_no_cpu_path():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    ref, ids = mg.clip_tiny()
    m = CLIPTextModel(CLIPTextConfig(vocab_size=1000, hidden_size=128, intermediate_size=512, num_hidden_layers=3,
                                     num_attention_heads=2)).load_transformers_state_dict(ref.state_dict())
    with pytest.raises(VdxError):
        m(ids)
