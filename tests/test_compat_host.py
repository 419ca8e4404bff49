"""CPU suite: the import shims for the three packages the reference imports and this image lacks (diffusers, pynvml,
cv2 — `Distribution/strategies/fsdp_chunked_coherent.py:15-16,22`), and the two result-section functions built on the
cv2 surface (`flow_err` :236-246, the mp4 :250-253)."""
import struct
import sys

import numpy as np
import pytest
import torch

import vdx  # noqa: F401
from vdx import metrics
from vdx.compat import cv2_shim as cv2


def smooth_image(h, w, seed):
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    img = ndimage.gaussian_filter(rng.standard_normal((h + 40, w + 40)), 4.0)
    img = (img - img.min()) / (img.max() - img.min())
    return img


def test_install_registers_only_missing_modules():
    from vdx import compat
    had = {m: m in sys.modules for m in ("diffusers", "pynvml", "cv2")}
    done = compat.install()
    try:
        for name in done:
            assert sys.modules[name].__name__.startswith("vdx.compat.")
        import diffusers
        import pynvml
        import cv2 as c
        assert hasattr(diffusers, "DiffusionPipeline") and hasattr(pynvml, "nvmlDeviceGetMemoryInfo")
        assert all(hasattr(c, n) for n in ("cvtColor", "calcOpticalFlowFarneback", "remap", "VideoWriter",
                                           "VideoWriter_fourcc", "COLOR_BGR2GRAY", "COLOR_RGB2BGR", "INTER_LINEAR"))
    finally:
        for name in done:
            if not had[name]:
                del sys.modules[name]


def test_cvtcolor_matches_opencv_fixed_point_weights():
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 77]]], np.uint8)      # B, G, R order
    assert cv2.cvtColor(px, cv2.COLOR_BGR2GRAY).tolist() == [[29, 150, 76, 142]]            # 0.114 / 0.587 / 0.299
    assert cv2.cvtColor(px, cv2.COLOR_RGB2GRAY).tolist() == [[76, 150, 29, (10 * 4899 + 200 * 9617 + 77 * 1868 + 8192) >> 14]]
    assert np.array_equal(cv2.cvtColor(px, cv2.COLOR_RGB2BGR), px[..., ::-1])


def test_remap_identity_shift_and_border():
    img = (np.arange(6 * 8).reshape(6, 8) * 3).astype(np.uint8)
    gx, gy = np.meshgrid(np.arange(8, dtype=np.float32), np.arange(6, dtype=np.float32))
    assert np.array_equal(cv2.remap(img, gx, gy, cv2.INTER_LINEAR), img)
    half = cv2.remap(img, gx + 0.5, gy, cv2.INTER_LINEAR)
    assert half[2, 3] == round((int(img[2, 3]) + int(img[2, 4])) / 2 + 1e-9) or abs(int(half[2, 3]) - (int(img[2, 3]) + int(img[2, 4])) / 2) <= 0.5
    out = cv2.remap(img, gx + 100, gy, cv2.INTER_LINEAR)
    assert out.max() == 0                                                                      # constant-0 border


@pytest.mark.parametrize("dx,dy", [(2.0, -1.5), (-3.0, 0.5), (0.0, 0.0)])
def test_farneback_recovers_a_translation(dx, dy):
    h, w = 96, 128
    big = smooth_image(h, w, 3) * 255
    gx, gy = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    a = cv2._sample(big, gx + 20, gy + 20)
    b = cv2._sample(big, gx + 20 - dx, gy + 20 - dy)            # b(x) = a(x - d): content moves by +d
    flow = cv2.calcOpticalFlowFarneback(a, b, None, 0.5, 3, 15, 3, 5, 1.2, 0)
    assert flow.shape == (h, w, 2) and flow.dtype == np.float32
    core = flow[16:-16, 16:-16]
    assert abs(core[..., 0].mean() - dx) < 0.15 and abs(core[..., 1].mean() - dy) < 0.15
    assert core[..., 0].std() < 0.3 and core[..., 1].std() < 0.3


def test_flow_err_follows_the_reference_loop():
    h, w = 64, 96
    big = smooth_image(h, w, 5) * 255
    gx, gy = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    def frame(shift):
        g = np.clip(cv2._sample(big, gx + 20 - shift, gy + 20), 0, 255).astype(np.uint8)
        return np.repeat(g[..., None], 3, axis=2)
    frames = [frame(1.5 * i) for i in range(6)]
    ranges = [(0, 3), (2, 5), (4, 6)]
    fe = metrics.flow_warp_error(frames, ranges)
    # the reference samples prev at x + flow (:241-243), i.e. it moves prev AGAINST the motion: for content moving by
    # +d the warped frame is prev(x + d) while next is prev(x - d).  Expected value with the true flow d = (1.5, 0):
    want = []
    for e in (3, 5):
        prev = frames[e - 1][..., 0].astype(np.float64)
        warped = np.rint(cv2.remap(prev, (gx + 1.5).astype(np.float32), gy.astype(np.float32), cv2.INTER_LINEAR))
        want.append(np.mean(np.abs(warped[:, :-2] - frames[e][:, :-2, 0].astype(np.float64))))
    assert fe is not None and abs(fe - np.mean(want)) < 0.15 * np.mean(want)
    assert metrics.flow_warp_error([frames[0]] * 6, ranges) == 0.0
    assert metrics.flow_warp_error(frames, [(0, 6)]) is None and metrics.flow_warp_error(frames[:1], [(0, 1)]) is None


def test_videowriter_writes_a_parseable_mp4_with_decodable_jpeg_samples(tmp_path):
    from PIL import Image
    import io
    path = str(tmp_path / "out.mp4")
    frames = [np.full((48, 64, 3), (10 * i, 100, 200 - 10 * i), np.uint8) for i in range(5)]      # RGB
    metrics.write_video(frames, path, 8)
    data = open(path, "rb").read()
    pos, boxes = 0, {}
    while pos < len(data):
        size, kind = struct.unpack(">I4s", data[pos:pos + 8])
        boxes[kind] = (pos, size)
        pos += size
    assert list(boxes) == [b"ftyp", b"mdat", b"moov"] and pos == len(data)
    moov = data[boxes[b"moov"][0]:boxes[b"moov"][0] + boxes[b"moov"][1]]
    assert b"mp4v" in moov and b"esds" in moov and b"stsz" in moov and b"vide" in moov
    i = moov.index(b"stsz")
    n = struct.unpack(">I", moov[i + 12:i + 16])[0]
    sizes = struct.unpack(f">{n}I", moov[i + 16:i + 16 + 4 * n])
    assert n == 5 and sum(sizes) == boxes[b"mdat"][1] - 8
    off = boxes[b"mdat"][0] + 8
    for k, s in enumerate(sizes):
        im = np.asarray(Image.open(io.BytesIO(data[off:off + s])).convert("RGB"))
        assert im.shape == (48, 64, 3) and abs(int(im[10, 10, 0]) - 10 * k) <= 3 and abs(int(im[10, 10, 2]) - (200 - 10 * k)) <= 3
        off += s


def test_hash_tokenizer_and_scheduler_surface():
    from vdx.compat import diffusers_shim as d
    tok = d.HashTokenizer()
    ids = tok(["a red panda eating bamboo", ""], padding="max_length", max_length=tok.model_max_length, truncation=True,
              return_tensors="pt").input_ids
    assert ids.shape == (2, 77) and ids.dtype == torch.int64
    assert ids[0, 0] == 49406 and ids[0, 6] == 49407 and (ids[1, 1:] == 49407).all() and 1000 <= int(ids[0, 1]) < 49000
    again = tok(["a red panda eating bamboo"]).input_ids
    assert torch.equal(again[0], ids[0])
    s = d.DDIMScheduler()
    s.set_timesteps(50, device="cpu")
    assert s.init_noise_sigma == 1.0 and [int(t) for t in s.timesteps[:2]] == [981, 961]


def test_from_pretrained_reads_a_local_checkpoint_directory(tmp_path):
    """`DiffusionPipeline.from_pretrained(<dir>)` (`fsdp_chunked_coherent.py:55-61`) on a diffusers-layout directory
    written here at narrow widths: the widths come from the directory's own config.json files (not from the Zeroscope
    defaults), the .safetensors weights are ingested strictly (a missing or an unexpected key raises), 1x1 projections
    stored as Conv2d weights are accepted, scheduler_config.json reaches the scheduler, and configurations the kernels
    are not built for are refused instead of loaded wrong."""
    import json
    import os
    from safetensors.torch import load_file, save_file
    import ckpt_dir
    from vdx._lib import VdxError
    from vdx.compat import diffusers_shim as d
    root = str(tmp_path / "ckpt")
    usd = ckpt_dir.write(root)
    pipe = d.DiffusionPipeline.from_pretrained(root, torch_dtype=torch.float16, low_cpu_mem_usage=True, use_safetensors=False,
                                               device_map=None)
    assert not pipe.synthetic_weights
    assert pipe.unet.cfg.block_out_channels == ckpt_dir.UNET_CH and pipe.unet.cfg.cross_attention_dim == ckpt_dir.CROSS
    assert pipe.unet.cfg.transformer_in_heads == 8 and pipe.unet.config.in_channels == 4
    assert pipe.vae.cfg.block_out_channels == ckpt_dir.VAE_CH and pipe.vae.config.scaling_factor == 0.18215
    assert pipe.text_encoder.cfg.hidden_size == 128 and pipe.text_encoder.cfg.num_hidden_layers == 2
    assert pipe.scheduler.config.steps_offset == 1 and pipe.scheduler.config.beta_end == 0.012
    # the conv-shaped projection was packed like the Linear one
    k = "down_blocks.0.attentions.0.proj_in.weight"
    assert usd[k].dim() == 4 and torch.equal(pipe.unet.W[k], usd[k].reshape(usd[k].shape[0], -1))
    # strict: unexpected and missing keys
    f = os.path.join(root, "unet", "diffusion_pytorch_model.safetensors")
    sd = load_file(f)
    save_file(dict(sd, **{"bogus.weight": torch.zeros(2)}), f)
    with pytest.raises(VdxError, match="unexpected"):
        d.DiffusionPipeline.from_pretrained(root)
    sd.pop("mid_block.resnets.1.conv2.bias")
    save_file(sd, f)
    with pytest.raises(VdxError, match="missing key"):
        d.DiffusionPipeline.from_pretrained(root)
    # refused configurations
    ckpt_dir.write(root)
    cfgf = os.path.join(root, "unet", "config.json")
    cfg = json.load(open(cfgf))
    json.dump(dict(cfg, attention_head_dim=8), open(cfgf, "w"))
    with pytest.raises(ValueError, match="attention_head_dim"):
        d.DiffusionPipeline.from_pretrained(root)
    json.dump(cfg, open(cfgf, "w"))
    tcf = os.path.join(root, "text_encoder", "config.json")
    json.dump(dict(json.load(open(tcf)), hidden_act="quick_gelu"), open(tcf, "w"))
    with pytest.raises(ValueError, match="hidden_act"):
        d.DiffusionPipeline.from_pretrained(root)
