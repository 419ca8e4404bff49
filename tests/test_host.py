"""CPU suite (-m "not gpu"): host logic of the product against the oracle, and the C-ABI library
(loads, exports every symbol include/vdx.h declares — no compute calls without a GPU)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

import vdx  # noqa: F401
from vdx import _lib, packing
from vdx.planner import PlannerError, plan
from vdx.scheduler import DDIMScheduler

from oracle.ddim_ref import DDIMSchedulerRef
from oracle.pipeline_ref import PlannerHang, my_ranges, plan_chunks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "vdx.h")).read()
    declared = set(re.findall(r"\b(vdx_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 15
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/vdx.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.load().vdx_version() >= 1


def test_gemm_args_struct_matches_header_field_order():
    hdr = open(os.path.join(ROOT, "include", "vdx.h")).read()
    body = re.search(r"typedef struct vdx_gemm_args \{(.*?)\} vdx_gemm_args;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        decl = re.sub(r"^(const\s+)?(void\s*\*|float\s*\*|int32_t|size_t)\s*", "", decl)
        names += [n.strip().lstrip("*") for n in decl.split(",")]
    assert names == [f[0] for f in _lib.GemmArgs._fields_]


@pytest.mark.parametrize("rule", ["coherent", "third"])
def test_planner_matches_oracle_sweep(rule):
    n = 0
    for T in (8, 16, 24, 32, 48, 64, 96, 100):
        for world in (1, 2, 3, 4, 6, 8):
            for cs in (0, 6, 8, 16, 24):
                for ov in (0, 2, 4):
                    for no_chunk in (False, True):
                        try:
                            want = plan_chunks(T, world, cs, ov, no_chunk, rule)
                        except PlannerHang:
                            with pytest.raises(PlannerError):
                                plan(T, world, cs, ov, no_chunk, rule)
                            continue
                        got = plan(T, world, cs, ov, no_chunk, rule)
                        assert (got.chunk, got.overlap, list(got.ranges)) == want, (T, world, cs, ov, no_chunk)
                        for r in range(world):
                            assert got.for_rank(r) == my_ranges(want[2], world, r)
                        assert len(got.ranges) % world == 0
                        n += 1
    assert n > 500


def test_scheduler_tables_match_oracle():
    a, b = DDIMScheduler(), DDIMSchedulerRef()
    assert torch.equal(a.alphas_cumprod, b.alphas_cumprod)
    for steps in (10, 50, 3):
        a.set_timesteps(steps)
        b.set_timesteps(steps)
        assert a.timesteps.tolist() == b.timesteps.tolist() == a._host_timesteps
        for t in a._host_timesteps:
            assert a.coefficients(t) == tuple(float(c) for c in b.coefficients(t))
    assert a.init_noise_sigma == 1.0
    x = torch.zeros(3)
    assert a.scale_model_input(x, 5) is x


def test_scheduler_step_has_no_cpu_path():
    s = DDIMScheduler()
    s.set_timesteps(10)
    with pytest.raises(_lib.VdxError):
        s.step(torch.zeros(1, 4, 2, 8, 8, dtype=torch.float16), 901, torch.zeros(1, 4, 2, 8, 8, dtype=torch.float16))


def test_packing_layouts():
    # gathered convolutions: K = (c // 64) * T * 64 + tap * 64 + c % 64
    w = torch.arange(2 * 128 * 9, dtype=torch.float32).reshape(2, 128, 3, 3)
    p = packing.pack_conv3x3(w)
    c, ky, kx = 70, 1, 2
    assert p.shape == (2, 9 * 128) and p[1, (c // 64) * 9 * 64 + (ky * 3 + kx) * 64 + c % 64] == w[1, c, ky, kx]
    w3 = torch.arange(2 * 128 * 3, dtype=torch.float32).reshape(2, 128, 3, 1, 1)
    p = packing.pack_tconv3(w3)
    assert p.shape == (2, 3 * 128) and p[1, (c // 64) * 3 * 64 + 2 * 64 + c % 64] == w3[1, c, 2, 0, 0]
    wi = torch.arange(2 * 4 * 9, dtype=torch.float32).reshape(2, 4, 3, 3)
    assert packing.pack_conv_in(wi)[1, (1 * 3 + 2) * 4 + 3] == wi[1, 3, 1, 2]
    w = torch.arange(16 * 2, dtype=torch.float32).reshape(16, 2)
    b = torch.arange(16, dtype=torch.float32)
    wp, bp = packing.pack_geglu(w, b)
    # rows: value 0-3, gate 0-3, value 4-7, gate 4-7
    assert bp.tolist() == [0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15]
    assert torch.equal(wp[:, 0] / 2, bp)
    assert packing.pad_rows(torch.ones(4, 3)).shape == (64, 3)


def test_unet_weight_ingest_covers_every_diffusers_key():
    """Packing consumes every key of the diffusers-shaped table exactly once (meta tensors: no memory)."""
    from vdx.unet3d import UNet3DConditionModel, UNet3DConfig
    from oracle.unet3d_ref import UNet3DConditionModelRef, UNet3DConfig as RefCfg
    with torch.device("meta"):
        ref = UNet3DConditionModelRef(RefCfg.zeroscope())
    sd = dict(ref.state_dict())
    m = UNet3DConditionModel(UNet3DConfig.zeroscope())
    m.load_diffusers_state_dict(sd, device="meta")
    # + conv_out rows padded 4 -> 64, conv_in K padded 36 -> 64
    # + the K7 images (a second, re-laid-out copy of the temporal q|k|v / to_out weights): width 320 (5 level-0 temporal
    #   transformers, 2 attentions each) as the blobs of csrc/tattn2.hip — 100 units of 8 KB + two fp32 vectors — and
    #   width 512 (transformer_in) as the stage images of csrc/tattn_fused.hip (to_out padded to the q|k|v stage size)
    k7 = 10 * (100 * 4096 + 2 * 2 * 320) + 2 * (3 * 512 * 512 + 2 * 16 * 12288)
    # + the K8 blobs (csrc/ff_fused.hip): the ten level-0 feed-forwards (5 spatial + 5 temporal transformers) as 300 units
    #   of 8 KB + the fp32 biases (2 x 1280 + 320)
    k8 = 10 * (300 * 4096 + 2 * (2 * 1280 + 320))
    # + the K5 blobs (csrc/xattn.hip): the five level-0 cross-attentions as 50 units of 4096 halfs + the two fp32 bias vectors
    k5 = 5 * (50 * 4096 + 2 * 2 * 320)
    # + K8's tail blobs (proj_out behind the feed-forward): the ten level-0 transformers, 25 units + the fp32 bias vector each
    k8p = 10 * (25 * 4096 + 2 * 320)
    assert m.num_parameters() == 1_411_233_860 + 60 * 2880 + 60 + 320 * 28 + k7 + k8 + k5 + k8p
    assert m.config.in_channels == 4
    sd["bogus.weight"] = torch.empty(1, device="meta")
    with pytest.raises(_lib.VdxError):
        UNet3DConditionModel(UNet3DConfig.zeroscope()).load_diffusers_state_dict(sd, device="meta")


GLOO_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
import vdx
from vdx.pipeline import gather_chunks
from vdx.planner import plan
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
cp = plan(24, world, 0, 4)
C, H, W = 4, 3, 5
mine = []
for (s, e) in cp.for_rank(rank):
    t = torch.arange(s, e, dtype=torch.float16).view(1, 1, e - s, 1, 1).expand(1, C, e - s, H, W).contiguous()
    mine.append(t * (rank + 1))
out = gather_chunks(mine, cp, rank, world)
# reference order: rank-major (for lst in gathered: for s,e,latc in lst)
want = [r for rr in range(world) for r in cp.for_rank(rr)]
assert [(s, e) for s, e, _ in out] == want, (out, want)
for i, (s, e, t) in enumerate(out):
    owner = [rr for rr in range(world) if (s, e) in cp.for_rank(rr)][0]
    ref = torch.arange(s, e, dtype=torch.float16).view(1, 1, e - s, 1, 1).expand(1, C, e - s, H, W) * (owner + 1)
    assert t.shape == (1, C, e - s, H, W) and torch.equal(t, ref)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_chunk_gather_two_ranks_gloo(tmp_path):
    script = tmp_path / "gather.py"
    script.write_text(GLOO_SCRIPT.format(root=ROOT))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


def test_state_dict_spec_matches_oracle_module_tree():
    from vdx.unet3d import UNet3DConfig
    from vdx.weights import state_dict_spec, synthetic_state_dict
    from oracle.unet3d_ref import UNet3DConditionModelRef, UNet3DConfig as RefCfg
    for ref_cfg, cfg in ((RefCfg.zeroscope(), UNet3DConfig.zeroscope()),
                         (RefCfg.tiny(), UNet3DConfig(block_out_channels=(64, 128, 128, 128),
                                                      cross_attention_dim=128, transformer_in_heads=2))):
        with torch.device("meta"):
            ref = UNet3DConditionModelRef(ref_cfg)
        want = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
        assert state_dict_spec(cfg) == want
    sd = synthetic_state_dict(cfg, seed=3)
    assert set(sd) == set(want) and all(v.dtype == torch.float16 for v in sd.values())
    assert abs(float(sd["conv_norm_out.weight"].float().mean()) - 1.0) < 0.05
    assert abs(float(sd["mid_block.temp_convs.0.conv2.0.weight"].float().mean()) - 1.0) < 0.05


def test_gemm_plan_is_host_only_and_splits_the_mostly_idle_last_round():
    """vdx_gemm_plan launches nothing (runs here without a GPU): the kernel family per shape and the row at which a
    product is cut into whole rounds of 256 big tiles + a tail (level-1 width-640 shapes at 24 and 16 frames)."""
    import ctypes as C
    from vdx import _lib
    lib = _lib.load()

    def plan(M, N, K, mode=0, taps=1, **kw):
        g = _lib.GemmArgs()
        g.a = g.w = g.out = 1 << 20            # never dereferenced on the host
        g.M, g.N, g.K, g.mode, g.c1 = M, N, K, mode, K // taps
        g.lda, g.ldo = K // taps, N
        for k, v in kw.items():
            setattr(g, k, v)
        v, s = C.c_int32(-1), C.c_int32(-1)
        assert lib.vdx_gemm_plan(C.byref(g), C.byref(v), C.byref(s)) == 0, lib.vdx_last_error()
        return v.value, s.value

    assert plan(110592, 640, 1920, mode=2, taps=3, frames=24, hw=2304) == (2, 98304)     # 3.375 rounds -> 3 + tail
    assert plan(73728, 640, 1920, mode=2, taps=3, frames=16, hw=2304) == (2, 65536)       # 2.25 rounds -> 2 + tail
    assert plan(73728, 640, 1920, mode=2, taps=3, frames=16, hw=2304, row_begin=65536) == (1, 0)   # the tail: 128x128
    assert plan(442368, 320, 960, mode=2, taps=3, frames=24, hw=9216) == (2, 0)           # 6.75 rounds: not worth it
    assert plan(18432, 1280, 1280) == (1, 16384)      # 16-frame level 2: 1.125 rounds of big tiles -> one whole round + a small-tile tail
    assert plan(18432, 1280, 1280, row_begin=16384) == (1, 0)
    assert plan(6912, 1280, 1280) == (8, 0)                                               # level 3
    assert plan(442368, 320, 320) == (7, 0)                                               # weights-stationary
    assert plan(442368, 320, 320, row_end=1024)[0] != 7                                   # ... whole products only
    g = _lib.GemmArgs()
    v, s = C.c_int32(), C.c_int32()
    assert lib.vdx_gemm_plan(C.byref(g), C.byref(v), C.byref(s)) != 0 and b"null" in lib.vdx_last_error()
    for bad in (dict(row_begin=4096), dict(row_begin=100, row_end=50), dict(row_end=5000), dict(row_begin=-1)):
        g = _lib.GemmArgs()
        g.a = g.w = g.out = 1 << 20
        g.M, g.N, g.K, g.c1, g.lda, g.ldo = 4096, 320, 1280, 1280, 1280, 320
        for k, val in bad.items():
            setattr(g, k, val)
        assert lib.vdx_gemm_plan(C.byref(g), C.byref(v), C.byref(s)) != 0 and b"rows [" in lib.vdx_last_error(), bad


def test_fused_conv_predicates_are_host_only_and_state_their_limits():
    """K1 / K3's `supported` / `preferred` entry points answer without a GPU (the UNet asks them per layer): widths in 64-channel
    slices, 320-column tiles, K1's per-picture scale / shift table bounded by the LDS left beside the tiles (C1 + C2 <= 1536),
    K3's frame chunk a divisor of the clip (16 / 12 / 8)."""
    lib = _lib.load()
    assert lib.vdx_conv3x3_gn_supported(320, 0, 320) == 1 and lib.vdx_conv3x3_gn_supported(640, 320, 320) == 1
    assert lib.vdx_conv3x3_gn_supported(960, 576, 320) == 1 and lib.vdx_conv3x3_gn_supported(1280, 0, 1280) == 1
    assert lib.vdx_conv3x3_gn_supported(1280, 1280, 1280) == 0          # 2560 input channels: the table does not fit
    assert lib.vdx_conv3x3_gn_supported(320, 0, 4) == 0 and lib.vdx_conv3x3_gn_supported(100, 0, 320) == 0
    assert lib.vdx_conv3x3_gn_supported(320, 40, 320) == 0
    assert lib.vdx_tconv_gn_supported(320, 320, 24) == 1 and lib.vdx_tconv_gn_supported(1280, 1280, 16) == 1
    assert lib.vdx_tconv_gn_supported(320, 320, 20) == 0 and lib.vdx_tconv_gn_supported(320, 256, 16) == 0


def test_split_k_is_not_taken_by_any_product_of_the_xl_forward():
    """whether a product runs its tail as split-K (a different summation order) depends on its row count, so a
    sample's bits could depend on the window length.  Pinned here: for every GEMM the XL UNet calls with allow_ksplit
    (3x3 convolutions, temporal convolutions, feed-forward output projections) at 24-, 16- and 12-frame windows,
    vdx_gemm_plan_ksplit answers "no" — the planner's 42 MB workspace limit and its cost model keep split-K out of the
    UNet; a change of its threshold that lets one in fails here before it moves a golden."""
    import ctypes as C
    from vdx import _lib
    lib = _lib.load()
    widths, B = (320, 640, 1280, 1280), 2

    def ks(M, N, K, mode, taps, hh=0, ww=0, F_=0, hw=0):
        g = _lib.GemmArgs()
        g.a = g.w = g.out = 1 << 20
        g.M, g.N, g.K, g.mode, g.c1 = M, N, K, mode, K // taps
        g.lda, g.ldo = K // taps, N
        if mode == 1:
            g.h_in = g.h_out = hh
            g.w_in = g.w_out = ww
            g.stride = 1
        if mode == 2:
            g.frames, g.hw = F_, hw
        s_, k_, w_ = C.c_int32(-1), C.c_int32(-1), C.c_size_t(1)
        assert lib.vdx_gemm_plan_ksplit(C.byref(g), C.byref(s_), C.byref(k_), C.byref(w_)) == 0, lib.vdx_last_error()
        return k_.value
    n = 0
    for F_ in (24, 16, 12):
        for lvl, Cw in enumerate(widths):
            hh, ww = 72 >> lvl, 128 >> lvl
            S = hh * ww
            M = B * F_ * S
            cins = {Cw}
            if lvl < 3:
                cins |= {Cw + widths[lvl + 1], 2 * Cw, Cw + (widths[lvl - 1] if lvl else Cw)}    # up-block concats, previous level in
            if lvl:
                cins.add(widths[lvl - 1])
            for cin in sorted(cins):
                assert ks(M, Cw, 9 * cin, 1, 9, hh, ww) == 0, (F_, lvl, cin)          # resnet conv1 / conv2
                n += 1
            assert ks(M, Cw, 3 * Cw, 2, 3, F_=F_, hw=S) == 0, (F_, lvl)               # temporal convolution
            assert ks(M, Cw, 4 * Cw, 0, 1) == 0, (F_, lvl)                            # feed-forward output projection
            n += 2
    assert n >= 60


def test_split_k_plan_never_takes_the_upsample_to_size_gather():
    """the split-K kernels are the VAR = 1 instantiation whose gather shifts by `upsample` (0 | 1); with
    upsample = 2 (nearest-to-size) they would read the wrong source pixels, in bounds and silently.  vdx_gemm_plan_ksplit
    must answer "no" for every such convolution (the advisor's case: 48 images 8x22 -> 16x43, N = 640, used to return
    ksplit = 8 at split_row 32768) and keep answering as before for the same shape without the size map."""
    import ctypes as C
    from vdx import _lib
    lib = _lib.load()

    def plan_ks(n_img, h_in, w_in, h_out, w_out, N, cin, ups):
        g = _lib.GemmArgs()
        g.a = g.w = g.out = 1 << 20            # never dereferenced on the host
        g.M, g.N, g.K, g.mode, g.c1 = n_img * h_out * w_out, N, 9 * cin, 1, cin
        g.lda, g.ldo = cin, N
        g.h_in, g.w_in, g.h_out, g.w_out, g.stride, g.upsample = h_in, w_in, h_out, w_out, 1, ups
        s_, k_, w_ = C.c_int32(-1), C.c_int32(-1), C.c_size_t(1)
        assert lib.vdx_gemm_plan_ksplit(C.byref(g), C.byref(s_), C.byref(k_), C.byref(w_)) == 0, lib.vdx_last_error()
        return s_.value, k_.value, w_.value

    assert plan_ks(48, 8, 22, 16, 43, 640, 640, 2) == (0, 0, 0)
    n = 0
    for hl, wl in ((16, 43), (17, 33), (33, 129), (21, 21), (50, 100), (129, 16)):
        for F_ in (8, 12, 16, 24):
            for B in (1, 2):
                for lvl, C_ in ((2, 1280), (1, 640)):        # the two upsamplers whose N fills 320-wide tiles on many rows
                    ho, wo = -(-hl // (1 << lvl)), -(-wl // (1 << lvl))
                    hi, wi = -(-ho // 2), -(-wo // 2)
                    assert plan_ks(B * F_, hi, wi, ho, wo, C_, C_, 2)[1] == 0
                    n += 1
    assert n == 96
    # the x2 form of a shape of the same size still plans a split (the check above is not vacuous)
    assert plan_ks(48, 8, 22, 16, 44, 640, 640, 1)[1] > 1


@pytest.mark.parametrize("Fr,rot", [(24, 0), (16, 3), (12, 1), (8, 4), (48, 2)])
def test_k7b_packing_matches_kernel_indexing(Fr, rot):
    """packing.pack_k7b against a lane-level walk through csrc/tattn2.hip's own addressing (tests/k7b_emulator.py: the
    unit / tile / slot arithmetic, the MFMA lane maps, the accumulator-as-operand hand-offs and the permuted k index of
    the output projection), compared with the fp32 statement of the sub-block on one 48-row group."""
    import numpy as np
    import torch.nn.functional as Fn
    from k7b_emulator import Wave, p0
    from vdx import packing
    inner, heads = 320, 5
    G = 48 // Fr
    g = torch.Generator().manual_seed(7 + Fr)
    h = lambda x: x.half().float()    # noqa: E731
    t = h(torch.randn(48, inner, generator=g) * 1.5 + 0.3)          # rows in group order: row = pixel * Fr + frame
    gamma, beta = h(1 + 0.2 * torch.randn(inner, generator=g)), h(0.1 * torch.randn(inner, generator=g))
    wq, wk, wv, wo = (h(torch.randn(inner, inner, generator=g) * s_) for s_ in (0.09, 0.09, 0.06, 0.05))
    bo = h(0.1 * torch.randn(inner, generator=g))
    ln = Fn.layer_norm(t, (inner,), gamma, beta, 1e-5)
    seq = lambda x: x.reshape(G, Fr, heads, 64).permute(0, 2, 1, 3)   # noqa: E731
    a = torch.softmax(seq(ln @ wq.t()) @ seq(ln @ wk.t()).transpose(-1, -2) * 0.125, -1) @ seq(ln @ wv.t())
    ref = (t + a.permute(0, 2, 1, 3).reshape(48, inner) @ wo.t() + bo).numpy()
    blob = packing.pack_k7b(wq, wk, wv, wo, gamma, beta, bo, 0.125)
    assert blob.dtype == torch.float16 and blob.numel() == 100 * 4096 + 2 * 2 * inner
    out = Wave(blob.numpy(), inner, Fr, rot).run(p0(t.numpy(), 1e-5), t.numpy())
    assert np.abs(out - ref).max() <= 3e-3 * np.abs(ref).max() + 3e-3


def test_k7_counted_waits_match_the_emitted_isa(tmp_path):
    """The same check for csrc/tattn2.hip (K7, second design): all 65 steps of its five frame specialisations — DMA pieces,
    plain loads, stores and the wait of every step as tools/k7b_check_waits.py derives them, against the emitted ISA."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    pkg = os.path.join(ROOT, "decentralised-verification-and-distributed-execution-of-large-scale-video-diffusion-models_amd")
    out = tmp_path / "tattn2.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "--cuda-device-only", "-S",
                    "-I", os.path.join(ROOT, "include"), "-I", os.path.join(pkg, "csrc"), os.path.join(pkg, "csrc", "tattn2.hip"), "-o", str(out)],
                   check=True, timeout=900)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "k7b_check_waits.py"), str(out)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.count(" 0 mismatching") == 5, r.stdout + r.stderr


def test_k8_counted_waits_match_the_emitted_isa(tmp_path):
    """csrc/ff_fused.hip waits with `s_waitcnt vmcnt(N)`, N derived from a model of every vector-memory instruction a wave
    issues per chunk.  The model is only right while the compiler emits exactly the loads the source counts, where it
    counts them (hipcc once merged 24 identical bias loads into 8: the wait then let a weight unit be read before it
    had landed).  tools/k8_check_waits.py re-derives the schedule and compares it with the ISA of this very source."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    pkg = os.path.join(ROOT, "decentralised-verification-and-distributed-execution-of-large-scale-video-diffusion-models_amd")
    out = tmp_path / "ff_fused.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "--cuda-device-only", "-S",
                    "-I", os.path.join(ROOT, "include"), "-I", os.path.join(pkg, "csrc"), os.path.join(pkg, "csrc", "ff_fused.hip"), "-o", str(out)],
                   check=True, timeout=600)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "k8_check_waits.py"), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_lab_build_is_refused_as_the_product(tmp_path):
    """`VDX_LIB_PATH` must not slip a stamps / ablation library (timing-only code paths, some with wrong
    results) into the product.  A library whose conv_fused unit is compiled with -DK1_ABL_NONORM reports it through
    `vdx_build_flags()`; `_lib.load()` raises unless VDX_ALLOW_LAB_BUILD=1 (what the lab tools set)."""
    import glob
    import subprocess
    import sys
    csrc = os.path.join(ROOT, "decentralised-verification-and-distributed-execution-of-large-scale-video-diffusion-models_amd", "csrc")
    objs = [o for o in glob.glob(os.path.join(csrc, "build", "*.o")) if os.path.basename(o) != "conv_fused.o"]
    if len(objs) < 10:
        pytest.skip("object files of the product build are not here (built elsewhere)")
    lab_o, lab_so = str(tmp_path / "conv_fused_lab.o"), str(tmp_path / "libvdx_hip_lab.so")
    hipcc = "/opt/rocm/bin/hipcc"
    subprocess.run([hipcc, "-O1", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-DK1_ABL_NONORM", "-c",
                    os.path.join(csrc, "conv_fused.hip"), "-o", lab_o], check=True, capture_output=True)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", lab_o] + objs + ["-ldl", "-o", lab_so], check=True, capture_output=True)
    code = "import sys; sys.path.insert(0, %r); import vdx; from vdx import _lib; l = _lib.load(); print('flags', l.vdx_build_flags())" % ROOT
    env = {k: v for k, v in os.environ.items() if k != "VDX_ALLOW_LAB_BUILD"}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(env, VDX_LIB_PATH=lab_so))
    assert r.returncode != 0 and "lab macros" in r.stderr and "= 64" in r.stderr, r.stderr[-1500:]
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(env, VDX_LIB_PATH=lab_so, VDX_ALLOW_LAB_BUILD="1"))
    assert r.returncode == 0 and "flags 64" in r.stdout, r.stderr[-1500:]
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "flags 0" in r.stdout, r.stderr[-1500:]


def test_k1_k3_choice_depends_on_one_samples_shape_only():
    """the fused GroupNorm + convolution kernels (K1, K3) and the apply pass + conv GEMM differ by one
    rounding of the normalised value, so WHICH of them runs must not depend on the batch, the window count or the chip —
    or a sample's bits would.  Pinned: at every level of the XL UNet the choice is the same for B in {1, 2} and any number of
    images, it is "fused" at level 0 for F in {8, 12, 16, 24} and "not fused" at levels 1-3 and at the small latents of the
    reference's other callers; and the C entry points really ignore the batch argument."""
    lib = _lib.load()
    widths = (320, 640, 1280, 1280)
    for lvl, Cw in enumerate(widths):
        hh, ww = 72 >> lvl, 128 >> lvl
        for F_ in (8, 12, 16, 24):
            k1 = {lib.vdx_conv3x3_gn_preferred(Cw, 0, Cw, B * F_, hh, ww) for B in (1, 2, 3)}
            k1 |= {lib.vdx_conv3x3_gn_preferred(Cw, 0, Cw, n, hh, ww) for n in (1, 7, 11, 48, 1000)}
            k3 = {lib.vdx_tconv_gn_preferred(Cw, Cw, B, F_, hh * ww) for B in (1, 2, 3, 16)}
            assert k1 == {1 if lvl == 0 else 0}, (lvl, F_, k1)
            assert k3 == {1 if lvl == 0 else 0}, (lvl, F_, k3)
        # the up blocks' concat inputs at level 0 (320 + 320, 320 + 640 channels)
        if lvl == 0:
            assert {lib.vdx_conv3x3_gn_preferred(320, c2, 320, n, hh, ww) for c2 in (320, 640) for n in (8, 24, 48)} == {1}
    for h_, w_ in ((32, 32), (40, 72), (16, 16), (16, 32)):               # cfg1 / InferNet caller latents: never the fused pair
        assert lib.vdx_conv3x3_gn_preferred(320, 0, 320, 48, h_, w_) == 0
        assert lib.vdx_tconv_gn_preferred(320, 320, 2, 24, h_ * w_) == (1 if ((h_ * w_ + 15) // 16) * 2 >= 512 else 0)


@pytest.mark.parametrize("kv_len,rot,item", [(77, 0, 0), (77, 3, 1), (5, 1, 1), (80, 4, 0), (33, 2, 1)])
def test_k5_packing_matches_kernel_indexing(kv_len, rot, item):
    """packing.pack_k5 / pack_k5_kv against a lane-level walk through csrc/xattn.hip's own addressing (tests/k5_emulator.py:
    the q stream and the output projection of tattn2's unit format, the per-prompt key / value fragment blobs, the MFMA lane
    maps, the masked key slots), compared with the fp32 statement of the cross-attention sub-block on one 48-row
    group of batch item `item`."""
    import numpy as np
    import torch.nn.functional as Fn
    from k5_emulator import Wave5
    from k7b_emulator import p0
    from vdx import packing
    inner, heads, cross, pad, n_items = 320, 5, 96, 128, 2
    g = torch.Generator().manual_seed(11 + kv_len + rot)
    h = lambda x: x.half().float()    # noqa: E731
    t = h(torch.randn(48, inner, generator=g) * 1.5 + 0.3)
    gamma, beta = h(1 + 0.2 * torch.randn(inner, generator=g)), h(0.1 * torch.randn(inner, generator=g))
    wq, wo = h(torch.randn(inner, inner, generator=g) * 0.09), h(torch.randn(inner, inner, generator=g) * 0.05)
    wk, wv = h(torch.randn(inner, cross, generator=g) * 0.12), h(torch.randn(inner, cross, generator=g) * 0.09)
    bo = h(0.1 * torch.randn(inner, generator=g))
    ehs = h(torch.randn(n_items, kv_len, cross, generator=g))
    ehs_pad = torch.zeros(n_items, pad, cross)
    ehs_pad[:, :kv_len] = ehs
    k_rows = h(ehs_pad.reshape(-1, cross) @ wk.t())                            # what the un-fused path's K GEMM leaves (fp16)
    vt = h(wv @ ehs_pad.reshape(-1, cross).t())                                # and its V^T GEMM
    ln = Fn.layer_norm(t, (inner,), gamma, beta, 1e-5)
    q = (ln @ wq.t()).reshape(48, heads, 64).permute(1, 0, 2)
    kk_ = k_rows.reshape(n_items, pad, heads, 64)[item, :kv_len].permute(1, 0, 2)
    vv_ = vt.t().reshape(n_items, pad, heads, 64)[item, :kv_len].permute(1, 0, 2)
    a = torch.softmax(q @ kk_.transpose(-1, -2) * 0.125, -1) @ vv_
    ref = (t + a.permute(1, 0, 2).reshape(48, inner) @ wo.t() + bo).numpy()
    blob = packing.pack_k5(wq, wo, gamma, beta, bo, 0.125)
    kvb = packing.pack_k5_kv(k_rows.half(), vt.half(), n_items, pad)
    lib = _lib.load()
    assert blob.dtype == torch.float16 and blob.numel() * 2 == lib.vdx_cross_attn_block_pack_bytes(inner)
    assert kvb.numel() * 2 == n_items * lib.vdx_cross_attn_block_kv_bytes(inner)
    out = Wave5(blob.numpy(), kvb.numpy(), item, rot, kv_len).run(p0(t.numpy(), 1e-5), t.numpy())
    assert np.abs(out - ref).max() <= 3e-3 * np.abs(ref).max() + 3e-3


def test_k5_counted_waits_match_the_emitted_isa(tmp_path):
    """The same check as for K7 / K8 on csrc/xattn.hip (K5): all 45 steps — DMA pieces, plain loads, stores and the counted
    `s_waitcnt vmcnt(N)` of every step as tools/k5_check_waits.py derives them from the schedule, against the emitted ISA."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    pkg = os.path.join(ROOT, "decentralised-verification-and-distributed-execution-of-large-scale-video-diffusion-models_amd")
    out = tmp_path / "xattn.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "--cuda-device-only", "-S",
                    "-I", os.path.join(ROOT, "include"), "-I", os.path.join(pkg, "csrc"), os.path.join(pkg, "csrc", "xattn.hip"), "-o", str(out)],
                   check=True, timeout=900)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "k5_check_waits.py"), str(out)], capture_output=True, text=True)
    assert r.returncode == 0 and " 0 mismatching" in r.stdout, r.stdout + r.stderr


# ---------------------------------------------------------------------------------------------------------------
# the job's front end (vdx/pipeline.py: build_arg_parser / config_from_args / emu_*): the reference's argparse and its
# network-emulation sleeps on the build's own driver
# ---------------------------------------------------------------------------------------------------------------
def test_front_end_has_the_reference_flags_and_defaults():
    """`fsdp_chunked_coherent.py:281-300`, flag for flag: name, type and default — `full_experiments_ZeroscopeXL.sh:25-92`
    drives the job through exactly these."""
    import vdx  # noqa: F401
    from vdx.pipeline import DiffuserConfig, build_arg_parser, config_from_args
    ref = {"model_id": "cerspense/zeroscope_v2_XL", "prompt": "a rocket in space, 4k", "num_frames": 32, "steps": 50,
           "guidance_scale": 7.5, "chunk_size": 0, "overlap": 4, "fps": 8, "height": 576, "width": 1024, "device": "cuda",
           "mode": "hybrid_ctx", "context_weight": 0.35, "emu_bw_mbps": 0, "emu_rtt_ms": 0, "emu_jitter_ms": 0,
           "out_csv": "results.csv"}
    p = build_arg_parser()
    a = p.parse_args([])
    for k, v in ref.items():
        assert getattr(a, k) == v and type(getattr(a, k)) is type(v), k
    mode = [x for x in p._actions if x.dest == "mode"][0]
    assert list(mode.choices) == ["fsdp", "chunk", "hybrid", "hybrid_ctx"]
    for flag, typ in (("--num_frames", int), ("--steps", int), ("--guidance_scale", float), ("--emu_bw_mbps", float),
                      ("--emu_rtt_ms", float), ("--emu_jitter_ms", float), ("--context_weight", float), ("--fps", int)):
        assert [x for x in p._actions if flag in x.option_strings][0].type is typ, flag
    cfg = config_from_args(p.parse_args("--mode hybrid --num_frames 48 --emu_bw_mbps 100 --emu_rtt_ms 20 --emu_jitter_ms 2".split()))
    assert isinstance(cfg, DiffuserConfig) and (cfg.use_fsdp, cfg.no_chunking, cfg.use_ctx) == (True, False, False)    # :302-304
    assert (cfg.num_frames, cfg.emu_bw_mbps, cfg.emu_rtt_ms, cfg.emu_jitter_ms) == (48, 100.0, 20.0, 2.0)
    assert config_from_args(p.parse_args(["--mode", "fsdp"])).no_chunking and not config_from_args(p.parse_args(["--mode", "chunk"])).use_fsdp


def test_network_emulation_sleeps_follow_the_reference():
    """:195-199 — `sleep(payload_bytes / (emu_bw_mbps * 1e6 / 8))`, then `sleep(max(0, gauss(rtt, jitter)) / 1000)`; :257-258 —
    `sleep(emu_rtt_ms / 1000)` before the reduction.  payload_bytes is the reference's formula (frames x channels x 2)."""
    import random
    import vdx  # noqa: F401
    from vdx.pipeline import DiffuserConfig, emu_gather_delay_s, emu_reduce_delay_s
    off = DiffuserConfig()
    assert emu_gather_delay_s(10 ** 9, off) == 0.0 and emu_reduce_delay_s(off) == 0.0
    payload = 16 * 4 * 2                                     # one 16-frame chunk: (e - s) * in_channels * 2  (:194)
    bw = DiffuserConfig(emu_bw_mbps=1.0)
    assert emu_gather_delay_s(payload, bw) == payload / (1.0 * 1e6 / 8) == 0.001024
    rtt = DiffuserConfig(emu_rtt_ms=20.0, emu_jitter_ms=0.0)
    assert abs(emu_gather_delay_s(payload, rtt) - 0.020) < 1e-12 and emu_reduce_delay_s(rtt) == 0.020
    both = DiffuserConfig(emu_bw_mbps=8.0, emu_rtt_ms=10.0, emu_jitter_ms=3.0)
    want = payload / 1e6 + max(0.0, random.Random(5).gauss(10.0, 3.0) / 1000.0)
    assert abs(emu_gather_delay_s(payload, both, random.Random(5)) - want) < 1e-12
    neg = DiffuserConfig(emu_rtt_ms=0.001, emu_jitter_ms=50.0)          # a negative draw sleeps 0, never a negative time
    assert min(emu_gather_delay_s(0, neg, random.Random(s)) for s in range(50)) == 0.0


def test_ws_counted_waits_match_the_emitted_isa():
    """csrc/gemm_ws.hip's `Ws::sync` waits with `s_waitcnt vmcnt(n)`, n = the DMA batches, epilogue stores and (round 6: residual
    rows requested one step ahead) loads a wave has issued since the chunk it is about to read.  tools/ws_check_waits.py compares
    the function totals of the emitted ISA with what the program structure issues, for every instantiation, and refuses any
    scratch / buffer operation (scratch traffic counts in vmcnt: a spill would silently break the count)."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ws_check_waits.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ws_check_waits: ok" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("ok  ") >= 15 and "ring=4" in r.stdout
