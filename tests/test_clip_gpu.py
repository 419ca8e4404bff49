"""-m gpu: CLIPTextModel (fsdp_chunked_coherent.py:96-103) on libvdx_hip.so against the REAL
dependency, `transformers.CLIPTextModel` in fp32 on the CPU with seeded weights: committed golden output (tiny
widths), the full SD-2.x text tower live, and the causal-mask property."""
import importlib.util
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
transformers = pytest.importorskip("transformers")
HERE = os.path.dirname(os.path.abspath(__file__))


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def _make_golden():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return mg


def test_flash_attention_causal_matches_torch(gpu):
    import vdx  # noqa: F401
    from vdx import ops
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(2)
    for n_seq, s, skv, heads in ((2, 128, 77, 2), (1, 320, 320, 1), (1, 640, 640, 2)):
        inner = heads * 64
        q = torch.randn(n_seq * s, inner, generator=g).half()
        k = torch.randn(n_seq * s, inner, generator=g).half()
        v = torch.randn(n_seq * s, inner, generator=g).half()
        qh, kh, vh = (t.float().view(n_seq, s, heads, 64).transpose(1, 2) for t in (q, k, v))
        mask = torch.ones(s, s, dtype=torch.bool).tril()
        mask[:, skv:] = False
        ref = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=mask).transpose(1, 2).reshape(n_seq * s, inner)
        out = ops.flash_attn(q.to(gpu), k.to(gpu), v.t().contiguous().to(gpu), n_seq=n_seq, sq=s, skv=skv, skv_pad=s,
                             heads=heads, seq_per_kv=1, scale=0.125, causal=True).float().cpu()
        err = (out - ref).abs()
        assert torch.isfinite(out).all() and err.max() <= 4e-3 * ref.abs().max() + 4e-3 * 1.0, (s, err.max())


def test_gelu_matches_torch(gpu):
    import vdx  # noqa: F401
    from vdx import ops
    x = torch.linspace(-8, 8, 4096).half()
    out = ops.gelu(x.to(gpu)).float().cpu()
    ref = torch.nn.functional.gelu(x.float())
    assert (out - ref).abs().max() <= 2e-3


def test_clip_tiny_matches_golden_from_transformers(gpu):
    import vdx  # noqa: F401
    from vdx.clip_text import CLIPTextConfig, CLIPTextModel
    ref, ids = _make_golden().clip_tiny()
    gold = np.load(os.path.join(HERE, "golden", "clip_tiny.npz"))
    m = CLIPTextModel(CLIPTextConfig(vocab_size=1000, hidden_size=128, intermediate_size=512, num_hidden_layers=3,
                                     num_attention_heads=2)).load_transformers_state_dict(ref.state_dict(), device=gpu)
    out = m(ids.to(gpu))
    assert out[0].shape == (2, 77, 128) and out[0].dtype == torch.float16 and out.last_hidden_state is out[0]
    err, floor = rel_l2(out[0].float().cpu(), torch.from_numpy(gold["out"])), float(gold["floor"])
    print(f"clip_tiny: rel-L2 {err:.3e} (fp16-CPU floor {floor:.3e})")
    assert err <= 2 * floor
    # causal: tokens after position 30 cannot influence positions <= 30
    ids2 = ids.clone()
    ids2[:, 31:] = 7
    out2 = m(ids2.to(gpu))[0]
    assert torch.equal(out2[:, :31], out[0][:, :31]) and not torch.equal(out2[:, 31:], out[0][:, 31:])


def test_clip_full_tower_matches_transformers_live(gpu):
    """The SD-2.x text tower at full size (1024 wide, 16 heads, 23 layers, 340 M parameters): transformers fp32 on
    the host vs the HIP path on the same seeded weights and token ids."""
    import vdx  # noqa: F401
    from vdx.clip_text import CLIPTextConfig, CLIPTextModel
    cfg = transformers.CLIPTextConfig(vocab_size=49408, hidden_size=1024, intermediate_size=4096, num_hidden_layers=23,
                                      num_attention_heads=16, max_position_embeddings=77, hidden_act="gelu",
                                      projection_dim=1024)
    torch.manual_seed(123)
    ref = transformers.CLIPTextModel(cfg).eval()
    with torch.no_grad():
        sd = {k: (v * (3.0 if (v.dim() == 2 and "embedding" not in k) else 1.0)).half().float() if v.dtype.is_floating_point else v
              for k, v in ref.state_dict().items()}
        ref.load_state_dict(sd)
    ids = torch.randint(0, 49408, (2, 77), generator=torch.Generator().manual_seed(9))
    ids[1, 1:] = 49407                                  # the empty prompt: BOS + EOS padding
    with torch.no_grad():
        want = ref(ids)[0]
    m = CLIPTextModel(CLIPTextConfig.sd2()).load_transformers_state_dict(ref.state_dict(), device=gpu)
    got = m(ids.to(gpu))[0]
    err = rel_l2(got.float().cpu(), want)
    print(f"clip full tower: rel-L2 {err:.3e}, out std {float(want.std()):.3f}")
    assert got.shape == (2, 77, 1024) and torch.isfinite(got).all() and err < 6e-3
    assert torch.equal(got, m(ids.to(gpu))[0])          # deterministic
