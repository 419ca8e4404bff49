"""-m gpu: every HIP kernel of libvdx_hip.so against a plain PyTorch fp32 CPU reference of the
same operator (the primitives diffusers composes), called through the C-ABI via vdx.ops.
Orchestration kernels (ctx injection, CFG+DDIM, blend) are compared BIT-EXACT with the oracle
(oracle/pipeline_ref.py, oracle/ddim_ref.py) which restates fsdp_chunked_coherent.py on CPU fp16.
Tolerance for fp16-output contractions: |err| <= 3e-3 * max|ref| + 3e-3 * |ref|  (one fp16
rounding of an fp32-accumulated result is 4.9e-4 relative)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    import vdx  # noqa: F401
    from vdx import ops, packing
    return ops, packing


def h(x):  # fp16-rounded fp32 CPU tensor
    return x.half().float()


def close(out, ref, tol=3e-3):
    out = out.float().cpu()
    assert out.shape == ref.shape, (out.shape, ref.shape)
    assert torch.isfinite(out).all()
    scale = ref.abs().max().item() + 1e-6
    err = (out - ref).abs()
    bound = tol * scale + tol * ref.abs()
    bad = (err > bound)
    assert not bad.any(), f"max err {err.max().item():.4g} (scale {scale:.4g}), {int(bad.sum())} / {bad.numel()} bad"


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(2, 64, 64), (100, 128, 320), (300, 320, 128), (257, 640, 64), (128, 192, 1024)])
@pytest.mark.parametrize("flags", ["none", "bias", "bias+res", "bias2", "all"])
def test_gemm_plain(gpu, M, N, K, flags):
    ops, _ = _ops()
    g = torch.Generator().manual_seed(M * 7 + N + K)
    a = h(torch.randn(M, K, generator=g))
    w = h(torch.randn(N, K, generator=g) / math.sqrt(K))
    bias = h(torch.randn(N, generator=g)) if "bias" in flags or flags == "all" else None
    res = h(torch.randn(M, N, generator=g)) if "res" in flags or flags == "all" else None
    rpb = 37
    b2 = h(torch.randn((M + rpb - 1) // rpb, N, generator=g)) if flags in ("bias2", "all") else None
    ref = a @ w.t()
    if bias is not None:
        ref = ref + bias
    if b2 is not None:
        ref = ref + b2.repeat_interleave(rpb, 0)[:M]
    if res is not None:
        ref = ref + res
    d = lambda t: None if t is None else t.half().to(gpu)
    out = ops.gemm(d(a), d(w), M=M, bias=d(bias), bias2=d(b2), rows_per_bias2=rpb if b2 is not None else 0,
                   residual=d(res))
    close(out, ref)


@pytest.mark.parametrize("variant", [0, 2, 3, 4, 6, 8])
def test_gemm_320_wide_kernels_all_modes(gpu, variant):
    """The 320-wide kernels in every gather mode, with M tails: variant 2 = 256x320 two-stage,
    3 = 256x320 four-stage ring (counted vmcnt), 4 = 128x320 two blocks per CU, 0 = automatic choice."""
    ops, packing = _ops()
    g = torch.Generator().manual_seed(77)
    d = lambda t: t.half().to(gpu)
    # plain, two sources, bias + residual: M = 12300 rows, N = 1280
    M, c1, c2, N = 12300, 64, 64, 1280
    a1, a2 = h(torch.randn(M, c1, generator=g)), h(torch.randn(M, c2, generator=g))
    w = h(torch.randn(N, c1 + c2, generator=g) / 11)
    b, res = h(torch.randn(N, generator=g)), h(torch.randn(M, N, generator=g))
    assert "320" in ops.gemm_kernel_name(M, N, c1 + c2, 0, False) or "<2, 2" in ops.gemm_kernel_name(M, N, c1 + c2, 0, False)
    close(ops.gemm(d(a1), d(w), M=M, a2=d(a2), bias=d(b), residual=d(res), variant=variant),
          torch.cat([a1, a2], 1) @ w.t() + b + res)
    # GEGLU: C = 320 -> N = 2560
    M, C = 6200, 320
    x = h(torch.randn(M, C, generator=g))
    w = h(torch.randn(8 * C, C, generator=g) / math.sqrt(C))
    bb = h(torch.randn(8 * C, generator=g) * 0.1)
    a_, g_ = (x @ w.t() + bb).chunk(2, dim=-1)
    wp, bp = packing.pack_geglu(w.half(), bb.half())
    close(ops.gemm(d(x), wp.to(gpu), M=M, bias=bp.to(gpu), geglu=True, variant=variant), a_ * F.gelu(g_))
    # conv3x3 with temb bias: 3 images of 63x65, 64 -> 1280, and a long-K one (K = 9*256 = 2304)
    for cin, cout in ((64, 1280), (256, 640)):
        n, hh, ww = 3, 63, 65
        x = h(torch.randn(n, cin, hh, ww, generator=g))
        w = h(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin))
        temb = h(torch.randn(n, cout, generator=g))
        ref = packing.nchw_to_rows(F.conv2d(x, w, None, padding=1) + temb[:, :, None, None])
        close(ops.gemm(d(packing.nchw_to_rows(x)), d(packing.pack_conv3x3(w)), M=n * hh * ww, mode=ops.CONV3X3,
                       bias2=d(temb), rows_per_bias2=hh * ww, conv=(n, hh, ww, hh, ww, 1, False), variant=variant), ref)
    # temporal conv: B=2, F=6, HW=1030, 64 -> 1280
    B, Fr, HW, C, Co = 2, 6, 1030, 64, 1280
    x5 = h(torch.randn(B, C, Fr, HW, 1, generator=g))
    w = h(torch.randn(Co, C, 3, 1, 1, generator=g) / 14)
    ref = F.conv3d(x5, w, None, padding=(1, 0, 0))[..., 0].permute(0, 2, 3, 1).reshape(B * Fr * HW, Co)
    rows = x5[..., 0].permute(0, 2, 3, 1).reshape(B * Fr * HW, C).contiguous()
    close(ops.gemm(d(rows), d(packing.pack_tconv3(w)), M=B * Fr * HW, mode=ops.TCONV3, tconv=(Fr, HW), variant=variant), ref)


@pytest.mark.parametrize("split,v_head,v_tail", [(4096, 2, 1), (5000, 2, 8), (256, 1, 2), (12288, 8, 3)])
def test_gemm_row_ranges_are_bit_identical_to_the_whole_product(gpu, split, v_head, v_tail):
    """vdx_gemm_args.row_begin / row_end: one product covered by two calls on different tile shapes (any split row,
    not only tile multiples) carries the bits of the single call, in every gather mode."""
    ops, packing = _ops()
    g = torch.Generator().manual_seed(split)
    d = lambda t: t.half().to(gpu)

    def both(**kw):
        whole = ops.gemm(variant=2, **kw)
        out = torch.full_like(whole, float("nan"))
        ops.gemm(variant=v_head, row_end=split, out=out, **kw)
        assert torch.isnan(out[split:].float()).all()              # rows past row_end untouched
        ops.gemm(variant=v_tail, row_begin=split, out=out, **kw)
        assert torch.equal(out, whole)

    M, c1, c2, N = 12300 + 77, 64, 128, 640
    both(a=d(torch.randn(M, c1, generator=g)), w=d(torch.randn(N, c1 + c2, generator=g) / 13), M=M,
         a2=d(torch.randn(M, c2, generator=g)), bias=d(torch.randn(N, generator=g)), residual=d(torch.randn(M, N, generator=g)))
    n, hh, ww, cin, cout = 3, 65, 64, 128, 320                       # 12480 rows; temb row per image
    x = h(torch.randn(n, cin, hh, ww, generator=g))
    wc = h(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin))
    both(a=d(packing.nchw_to_rows(x)), w=d(packing.pack_conv3x3(wc)), M=n * hh * ww, mode=ops.CONV3X3,
         bias2=d(torch.randn(n, cout, generator=g)), rows_per_bias2=hh * ww, conv=(n, hh, ww, hh, ww, 1, False))
    B, Fr, HW, C, Co = 2, 6, 1031, 64, 640                           # 12372 rows
    rows = h(torch.randn(B * Fr * HW, C, generator=g))
    wt = h(torch.randn(Co, C, 3, 1, 1, generator=g) / 14)
    both(a=d(rows), w=d(packing.pack_tconv3(wt)), M=B * Fr * HW, mode=ops.TCONV3, tconv=(Fr, HW),
         bias=d(torch.randn(Co, generator=g)))


@pytest.mark.parametrize("mode,ks", [("conv", 8), ("conv", 3), ("tconv", 4), ("plain", 2), ("plain2", 5)])
def test_gemm_split_k_tail(gpu, mode, ks):
    """vdx_gemm_args.ksplit: the rows of a call as `ks` K slices per 256x320 tile (fp32 slabs) + the fixed-order reduction
    that runs the epilogue (bias, time-embedding rows, residual), for the conv / temporal-conv gathers and plain rows,
    with row tails and a row range that starts inside the product; against fp32, against the unsplit kernel (same
    values up to the fp32 summation order) and bitwise against itself."""
    ops, packing = _ops()
    g = torch.Generator().manual_seed(ks * 7 + len(mode))
    if mode == "conv":
        n, cin, cout, hh, ww = 5, 128, 320, 24, 20
        x = h(torch.randn(n, cin, hh, ww, generator=g))
        w = h(torch.randn(cout, cin, 3, 3, generator=g) / 30)
        b = h(torch.randn(cout, generator=g) * 0.1)
        temb = h(torch.randn(n, cout, generator=g) * 0.2)
        res = h(torch.randn(n * hh * ww, cout, generator=g))
        ref = packing.nchw_to_rows(F.conv2d(x, w, b, padding=1) + temb[:, :, None, None]) + res
        kw = dict(a=packing.nchw_to_rows(x).half().to(gpu), w=packing.pack_conv3x3(w.half()).to(gpu), M=n * hh * ww,
                  mode=ops.CONV3X3, bias=b.half().to(gpu), bias2=temb.half().to(gpu), rows_per_bias2=hh * ww,
                  residual=res.half().to(gpu), conv=(n, hh, ww, hh, ww, 1, False))
    elif mode == "tconv":
        B, Fr, S, C = 2, 5, 130, 320
        x5 = h(torch.randn(B, C, Fr, S, 1, generator=g))
        w = h(torch.randn(C, C, 3, 1, 1, generator=g) / 30)
        b = h(torch.randn(C, generator=g) * 0.1)
        y = F.conv3d(x5, w, b, padding=(1, 0, 0))
        ref = y.permute(0, 2, 3, 4, 1).reshape(B * Fr * S, C)
        kw = dict(a=x5.permute(0, 2, 3, 4, 1).reshape(B * Fr * S, C).half().contiguous().to(gpu), w=packing.pack_tconv3(w.half()).to(gpu),
                  M=B * Fr * S, mode=ops.TCONV3, bias=b.half().to(gpu), tconv=(Fr, S))
    else:
        M, N, K = (1500, 640, 2560) if mode == "plain" else (700, 1280, 5120)
        a = h(torch.randn(M, K, generator=g))
        w = h(torch.randn(N, K, generator=g) / 40)
        b = h(torch.randn(N, generator=g) * 0.1)
        res = h(torch.randn(M, N, generator=g))
        ref = a @ w.t() + b + res
        kw = dict(a=a.half().to(gpu), w=w.half().to(gpu), M=M, bias=b.half().to(gpu), residual=res.half().to(gpu))
    M = kw["M"]
    plain = ops.gemm(variant=2, **kw)
    out = ops.gemm(ksplit=ks, **kw)
    close(out, ref)
    close(out, plain.float().cpu(), tol=2e-3)
    assert torch.equal(out, ops.gemm(ksplit=ks, **kw))
    # a row range that starts inside the product: rows before it stay untouched, rows in it equal the whole split-K product
    rb = 256 * (M // 512)
    part = torch.full_like(out, float("nan"))
    ops.gemm(ksplit=ks, row_begin=rb, out=part, **kw)
    assert torch.isnan(part[:rb].float()).all() and torch.equal(part[rb:], out[rb:])


def test_gemm_split_k_plan(gpu):
    """The planners on the shapes they were fitted for.  vdx_gemm_plan: the level-2 convolution of a 16-frame window
    (18 432 rows, N = 1280: 288 big tiles = one round + 32) is cut into whole rounds of big tiles + a small-tile tail
    although the product as a whole would prefer small tiles.  vdx_gemm_plan_ksplit: consistent answers (a split-K tail
    only with >= 4 K tiles per slice and at most 128 slabs of workspace; none for products that fill their rounds, for
    short K, for what the weights-stationary kernels take); ops.gemm(allow_ksplit=True) runs whatever is planned."""
    ops, _ = _ops()
    import ctypes as C
    from vdx import _lib
    lib = _lib.load()

    def args(M, N, K, mode=0, taps=1):
        g = _lib.GemmArgs()
        dummy = torch.zeros(16, dtype=torch.float16, device=gpu).data_ptr()
        g.a, g.w, g.out = dummy, dummy, dummy
        g.M, g.N, g.K, g.mode, g.c1 = M, N, K, mode, K // taps
        g.lda, g.ldo = K // taps, N
        g.h_in = g.h_out = 16
        g.w_in = g.w_out = 16
        g.stride = 1
        g.frames, g.hw = 16, M // (2 * 16) if mode == 2 else 1
        return g

    def plan_ks(*a, **k):
        g = args(*a, **k)
        s_, k_, w_ = C.c_int32(0), C.c_int32(0), C.c_size_t(0)
        _lib.check(lib.vdx_gemm_plan_ksplit(C.byref(g), C.byref(s_), C.byref(k_), C.byref(w_)), "plan")
        return s_.value, k_.value, w_.value

    def plan(*a, **k):
        g = args(*a, **k)
        v_, s_ = C.c_int32(0), C.c_int32(0)
        _lib.check(lib.vdx_gemm_plan(C.byref(g), C.byref(v_), C.byref(s_)), "plan")
        return v_.value, s_.value
    assert plan(18432, 1280, 11520, mode=1, taps=9) == (1, 16384)           # small tiles as a whole, big + small when split
    assert plan(73728, 640, 5760, mode=1, taps=9) == (2, 65536)
    assert plan(65536, 320, 2880, mode=1, taps=9) == (2, 0)                 # whole rounds
    for shape in ((18432, 1280, 11520, 1, 9), (73728, 640, 5760, 1, 9), (65536, 320, 2880, 1, 9), (18432, 1280, 320, 0, 1),
                  (18432, 1280, 23040, 1, 9)):
        M, N, K, mode, taps = shape
        split, ks, ws = plan_ks(M, N, K, mode=mode, taps=taps)
        if ks:
            t = -(-(M - split) // 256) * -(-N // 320)
            assert ks >= 2 and (K // 64) // ks >= 4 and t * ks <= 128 and ws == t * ks * 327680 and split % 256 == 0
        else:
            assert (split, ws) == (0, 0)
    assert plan_ks(65536, 320, 2880, mode=1, taps=9)[1] == 0 and plan_ks(18432, 1280, 320)[1] == 0
    M, N, K = 18432, 1280, 5120
    g = torch.Generator().manual_seed(1)
    a = h(torch.randn(M, K, generator=g)).half().to(gpu)
    w = (h(torch.randn(N, K, generator=g)) / 60).half().to(gpu)
    rec = []
    ops.PROFILE = rec
    try:
        out = ops.gemm(a, w, M=M, allow_ksplit=True)
    finally:
        ops.PROFILE = None
    assert len(rec) == 2 and rec[0][4][0] == 16384 and rec[1][4][0] == M - 16384
    close(out, a.float().cpu() @ w.float().cpu().t())


def test_gemm_automatic_tail_split(gpu):
    """A 16-frame level-1 temporal conv (73 728 x 640, 2.25 rounds of 256x320 tiles): the automatic path runs two whole
    rounds + a 128x128 tail (vdx_gemm_plan) and returns the bits of the single 256x320 launch; bench.py's per-launch
    records name both kernels."""
    ops, packing = _ops()
    g = torch.Generator().manual_seed(3)
    B, Fr, HW, C = 2, 16, 2304, 640
    M = B * Fr * HW
    rows = torch.randn(M, C, generator=g).half().to(gpu)
    w = packing.pack_tconv3(h(torch.randn(C, C, 3, 1, 1, generator=g) / 40)).half().to(gpu)
    one = ops.gemm(rows, w, M=M, mode=ops.TCONV3, tconv=(Fr, HW), variant=2)
    ops.PROFILE = rec = []
    try:
        auto = ops.gemm(rows, w, M=M, mode=ops.TCONV3, tconv=(Fr, HW))
    finally:
        ops.PROFILE = None
    assert torch.equal(auto, one)
    assert [(r[0].split("<")[1][:8], r[4][0]) for r in rec] == [("256, 320", 65536), ("128, 128", 8192)]


@pytest.mark.parametrize("M,N,K,flags", [
    (64, 320, 320, "bias"), (192, 640, 320, "none"), (4160, 960, 320, "bias"), (1984, 320, 320, "bias+res"),
    (16448, 320, 320, "bias+res"), (704, 1600, 320, "geglu"), (20480, 1920, 320, "none"),      # 10 waves, 320-wide panels
    (16448, 2560, 320, "geglu"), (704, 2560, 320, "geglu"), (1088, 512, 320, "bias"),           # 8 waves, pipelined epilogue
    (320, 1536, 512, "none"), (16416, 512, 512, "bias+res"), (2080, 4096, 512, "geglu"), (992, 768, 512, "bias+res"),
    (160, 640, 640, "bias+res"), (16416, 1920, 640, "none"), (2080, 5120, 640, "geglu"), (96, 1280, 640, "bias")])
def test_gemm_weights_stationary(gpu, M, N, K, flags):
    """gemm_ws.hip (short-K Linear layers, K in {320, 512, 640}): pinned by variant 7 on small M, picked
    automatically from M >= 16384; odd/even/uneven chunk counts per row group, partial last panels, strided
    A / residual / out views."""
    ops, packing = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    d = lambda t: t.half().to(gpu)
    x = h(torch.randn(M, K, generator=g))
    w = h(torch.randn(N, K, generator=g) / math.sqrt(K))
    b = h(torch.randn(N, generator=g) * 0.3)
    variant = 0 if M >= 16384 else 7
    assert "gemm_ws_kernel" in ops.gemm_kernel_name(M, N, K, ops.PLAIN, flags == "geglu", variant)
    abuf = torch.zeros(M, K + 64, dtype=torch.float16, device=gpu)     # lda > K
    abuf[:, 32:32 + K] = d(x)
    a_view = abuf[:, 32:32 + K]
    if flags == "geglu":
        a_, g_ = (x @ w.t() + b).chunk(2, dim=-1)
        wp, bp = packing.pack_geglu(w.half(), b.half())
        obuf = torch.full((M, N // 2 + 8), 7.0, dtype=torch.float16, device=gpu)
        out = ops.gemm(a_view, wp.to(gpu), M=M, bias=bp.to(gpu), geglu=True, variant=variant, out=obuf[:, :N // 2])
        close(out, a_ * F.gelu(g_))
        assert (obuf[:, N // 2:] == 7.0).all()
        return
    ref = x @ w.t()
    bias = res = None
    if "bias" in flags:
        bias = d(b)
        ref = ref + b
    if "res" in flags:
        r = h(torch.randn(M, N, generator=g))
        rbuf = torch.zeros(M, N + 16, dtype=torch.float16, device=gpu)
        rbuf[:, 8:8 + N] = d(r)
        res = rbuf[:, 8:8 + N]
        ref = ref + r
    obuf = torch.full((M, N + 8), 7.0, dtype=torch.float16, device=gpu)
    out = ops.gemm(a_view, d(w), M=M, bias=bias, residual=res, variant=variant, out=obuf[:, :N])
    close(out, ref)
    assert (obuf[:, N:] == 7.0).all()
    # the tiled kernel computes the same products in the same K order: identical bits
    assert torch.equal(out, ops.gemm(a_view, d(w), M=M, bias=bias, residual=res, variant=2))


@pytest.mark.parametrize("variant", [2, 4, 8])
@pytest.mark.parametrize("M,N", [(3000, 1600), (700, 2880), (5000, 4160)])
def test_gemm_wide_n_panel_order(gpu, variant, M, N):
    """More than four 320-wide column tiles: tiles are walked in column panels of 4 (gemm_tile_of); 5, 9 and 13 tiles
    exercise a last panel of width 1; M tails on top."""
    ops, _ = _ops()
    g = torch.Generator().manual_seed(M + N)
    K = 128
    a = h(torch.randn(M, K, generator=g))
    w = h(torch.randn(N, K, generator=g) / math.sqrt(K))
    b = h(torch.randn(N, generator=g))
    out = ops.gemm(a.half().to(gpu), w.half().to(gpu), M=M, bias=b.half().to(gpu), variant=variant)
    close(out, a @ w.t() + b)


def test_gemm_two_sources_and_strided_views(gpu):
    ops, _ = _ops()
    g = torch.Generator().manual_seed(5)
    M, c1, c2, N = 200, 128, 64, 128
    a1, a2 = h(torch.randn(M, c1, generator=g)), h(torch.randn(M, c2, generator=g))
    w = h(torch.randn(N, c1 + c2, generator=g) / 14)
    ref = torch.cat([a1, a2], 1) @ w.t()
    # sources are column slices of wider buffers (lda > c)
    buf1 = torch.zeros(M, 256, dtype=torch.float16, device=gpu)
    buf1[:, 64:64 + c1] = a1.half().to(gpu)
    buf2 = torch.zeros(M + 3, 64, dtype=torch.float16, device=gpu)
    buf2[:M] = a2.half().to(gpu)
    outbuf = torch.zeros(M, 512, dtype=torch.float16, device=gpu)
    ops.gemm(buf1[:, 64:64 + c1], w.half().to(gpu), M=M, a2=buf2, out=outbuf[:, 128:128 + N])
    close(outbuf[:, 128:128 + N], ref)
    assert outbuf[:, :128].abs().max() == 0 and outbuf[:, 256:].abs().max() == 0


def test_gemm_swapped_gives_transposed_product(gpu):
    """V^T = Wv . X^T : the same kernel with weights as `a` and activations as `w`."""
    ops, _ = _ops()
    g = torch.Generator().manual_seed(6)
    tokens, C = 192, 128
    x = h(torch.randn(tokens, C, generator=g))
    wv = h(torch.randn(C, C, generator=g) / 11)
    vt = ops.gemm(wv.half().to(gpu), x.half().to(gpu), M=C)
    close(vt, (x @ wv.t()).t().contiguous())


@pytest.mark.parametrize("M,C", [(70, 64), (260, 320)])
def test_gemm_geglu(gpu, M, C):
    ops, packing = _ops()
    g = torch.Generator().manual_seed(M + C)
    x = h(torch.randn(M, C, generator=g))
    w = h(torch.randn(8 * C, C, generator=g) / math.sqrt(C))
    b = h(torch.randn(8 * C, generator=g) * 0.1)
    a_, g_ = (x @ w.t() + b).chunk(2, dim=-1)
    ref = a_ * F.gelu(g_)
    wp, bp = packing.pack_geglu(w.half(), b.half())
    out = ops.gemm(x.half().to(gpu), wp.to(gpu), M=M, bias=bp.to(gpu), geglu=True)
    assert out.shape == (M, 4 * C)
    close(out, ref)


@pytest.mark.parametrize("cin,cout,n,hh,ww,stride,ups", [
    (64, 64, 3, 6, 10, 1, False), (128, 320, 2, 9, 7, 1, False), (64, 128, 2, 8, 12, 2, False),
    (64, 64, 2, 7, 9, 2, False), (64, 128, 2, 5, 6, 1, True)])
def test_conv3x3(gpu, cin, cout, n, hh, ww, stride, ups):
    ops, packing = _ops()
    g = torch.Generator().manual_seed(cin + cout + hh)
    x = h(torch.randn(n, cin, hh, ww, generator=g))
    w = h(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin))
    b = h(torch.randn(cout, generator=g) * 0.1)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    ref4 = F.conv2d(xin, w, b, stride=stride, padding=1)
    ho, wo = ref4.shape[2:]
    temb = h(torch.randn(n, cout, generator=g))
    res = h(torch.randn(n * ho * wo, cout, generator=g))
    ref = packing.nchw_to_rows(ref4 + temb[:, :, None, None]) + res
    out = ops.gemm(packing.nchw_to_rows(x).half().to(gpu), packing.pack_conv3x3(w).half().to(gpu),
                   M=n * ho * wo, mode=ops.CONV3X3, bias=b.half().to(gpu),
                   bias2=temb.half().to(gpu), rows_per_bias2=ho * wo, residual=res.half().to(gpu),
                   conv=(n, hh, ww, ho, wo, stride, ups))
    close(out, ref)


@pytest.mark.parametrize("B,Fr,HW,C,Co", [(2, 5, 12, 64, 64), (1, 24, 9, 128, 128), (2, 1, 10, 64, 128)])
def test_tconv3(gpu, B, Fr, HW, C, Co):
    ops, packing = _ops()
    g = torch.Generator().manual_seed(Fr + C)
    x5 = h(torch.randn(B, C, Fr, HW, 1, generator=g))
    w = h(torch.randn(Co, C, 3, 1, 1, generator=g) / math.sqrt(3 * C))
    b = h(torch.randn(Co, generator=g) * 0.1)
    ref5 = F.conv3d(x5, w, b, padding=(1, 0, 0))                       # (B,Co,F,HW,1)
    rows = x5[..., 0].permute(0, 2, 3, 1).reshape(B * Fr * HW, C).contiguous()
    ref = ref5[..., 0].permute(0, 2, 3, 1).reshape(B * Fr * HW, Co).contiguous()
    res = h(torch.randn(B * Fr * HW, Co, generator=g))
    out = ops.gemm(rows.half().to(gpu), packing.pack_tconv3(w).half().to(gpu), M=B * Fr * HW, mode=ops.TCONV3,
                   bias=b.half().to(gpu), residual=res.half().to(gpu), tconv=(Fr, HW))
    close(out, ref + res)


@pytest.mark.parametrize("n,hh,ww,c1,c2,cout,temb,resid", [
    (3, 12, 32, 64, 0, 320, True, False),       # two patch rows, one patch column; conv1 form (time-embedding row)
    (2, 7, 64, 128, 0, 320, False, True),       # ragged patch rows (7 = 6 + 1), two patch columns; conv2 form (residual)
    (2, 6, 40, 64, 64, 320, True, False),       # two sources (skip concat), ragged width (40 = 32 + 8)
    (1, 18, 32, 192, 128, 640, True, True),     # five channel slices across the two sources, two column tiles
    (4, 5, 9, 64, 0, 320, False, False),        # an image smaller than one patch
])
def test_conv3x3_gn_fused(gpu, n, hh, ww, c1, c2, cout, temb, resid):
    """K1 (csrc/conv_fused.hip): statistics pass + the 3x3 convolution that normalises its staged image patch in LDS, against
    the fp32 statement of ResnetBlock2D's conv(SiLU(GroupNorm(cat(x, skip)))) + time-embedding row + residual (every op
    output rounded to fp16 as the reference's fp16 modules do) AND against the un-fused kernels."""
    ops, packing = _ops()
    g = torch.Generator().manual_seed(n * 100 + hh + ww + c1 + c2)
    C_ = c1 + c2
    x = h(torch.randn(n, C_, hh, ww, generator=g) * 1.5 + 0.4 * torch.randn(1, C_, 1, 1, generator=g))
    gamma, beta = h(1 + 0.2 * torch.randn(C_, generator=g)), h(0.3 * torch.randn(C_, generator=g))
    w = h(torch.randn(cout, C_, 3, 3, generator=g) / math.sqrt(9 * C_))
    b = h(torch.randn(cout, generator=g) * 0.1)
    frames = 2 if n % 2 == 0 else 1                    # images per time-embedding row
    te = h(torch.randn(n // frames, cout, generator=g) * 0.3) if temb else None
    M = n * hh * ww
    res = h(torch.randn(M, cout, generator=g)) if resid else None
    act = h(F.silu(h(F.group_norm(x, 32, gamma, beta, 1e-5))))
    ref4 = F.conv2d(act, w, b, padding=1)
    if temb:
        ref4 = ref4 + te.repeat_interleave(frames, 0)[:, :, None, None]
    ref = packing.nchw_to_rows(ref4)
    if resid:
        ref = ref + res
    dv = lambda t: None if t is None else t.half().to(gpu)   # noqa: E731
    rows = packing.nchw_to_rows(x).half().to(gpu)
    xa = rows[:, :c1].contiguous()
    xb = rows[:, c1:].contiguous() if c2 else None
    wp = packing.pack_conv3x3(w.half()).to(gpu)
    assert ops.conv3x3_gn_supported(c1, c2, cout)
    kw = dict(x2=xb, bias=dv(b), bias2=dv(te), rows_per_bias2=frames * hh * ww, residual=dv(res), groups=32, n_img=n, h=hh, wd=ww, eps=1e-5)
    out = ops.conv3x3_gn(xa, dv(gamma), dv(beta), wp, **kw)
    close(out, ref, tol=4e-3)
    nrm = ops.groupnorm(xa, dv(gamma), dv(beta), groups=32, n_samples=n, rows_per_sample=hh * ww, eps=1e-5, silu_act=True, x2=xb)
    unf = ops.gemm(nrm, wp, M=M, mode=ops.CONV3X3, bias=dv(b), bias2=dv(te), rows_per_bias2=frames * hh * ww, residual=dv(res),
                   conv=(n, hh, ww, hh, ww, 1, False))
    close(out, unf.float().cpu(), tol=2e-3)
    assert torch.equal(out, ops.conv3x3_gn(xa, dv(gamma), dv(beta), wp, **kw))
    # a sample's bits do not depend on the batch it is computed in (tiles never span images; statistics per image)
    if n > 1 and not temb:
        S = hh * ww
        one = ops.conv3x3_gn(xa[S:2 * S], dv(gamma), dv(beta), wp, x2=None if xb is None else xb[S:2 * S], bias=dv(b),
                             residual=None if res is None else dv(res)[S:2 * S], groups=32, n_img=1, h=hh, wd=ww, eps=1e-5,
                             partition_samples=n)
        both = ops.conv3x3_gn(xa, dv(gamma), dv(beta), wp, x2=xb, bias=dv(b), residual=dv(res), groups=32, n_img=n, h=hh, wd=ww, eps=1e-5,
                              partition_samples=n)
        assert torch.equal(one, both[S:2 * S])


def _tconv_gn_ref(x5, gamma, beta, w, b, res, eps=1e-5):
    """fp32 statement of one link of TemporalConvLayer's chain: GroupNorm(32) over (C/32, F, h, w) jointly,
    SiLU, Conv3d (3,1,1) with zero padding in time — every op output rounded to fp16 as the reference's fp16 modules do."""
    n = h(F.group_norm(x5, 32, gamma, beta, eps))
    a = h(F.silu(n))
    y = F.conv3d(a, w, b, padding=(1, 0, 0))
    B, Co, Fr, S, _ = y.shape
    y = y[..., 0].permute(0, 2, 3, 1).reshape(B * Fr * S, Co)
    return y + res if res is not None else y


@pytest.mark.parametrize("B,Fr,S,C,Co,resid", [
    (2, 24, 16, 64, 320, True),        # FT = 12: two frame chunks with halo frames, level-0 like
    (1, 16, 40, 128, 320, False),      # FT = 16: the whole clip in one tile; 40 pixels = 2.5 pixel blocks (tail rows)
    (2, 12, 9, 64, 640, True),         # FT = 12, fewer pixels than one block, two column tiles
    (1, 8, 33, 192, 320, True),        # FT = 8, three channel slices
    (2, 48, 16, 64, 320, False),       # FT = 16: three chunks
])
def test_tconv_gn_fused(gpu, B, Fr, S, C, Co, resid):
    """K3 (csrc/tconv_fused.hip): statistics pass + the temporal convolution that normalises its staged image in LDS,
    against the fp32 reference AND against the un-fused kernels (GroupNorm apply pass + TCONV3 GEMM)."""
    ops, packing = _ops()
    g = torch.Generator().manual_seed(B * 1000 + Fr * 10 + C)
    x5 = h(torch.randn(B, C, Fr, S, 1, generator=g) * 1.5 + 0.3 * torch.randn(1, C, 1, 1, 1, generator=g))
    gamma, beta = h(1 + 0.2 * torch.randn(C, generator=g)), h(0.3 * torch.randn(C, generator=g))
    w = h(torch.randn(Co, C, 3, 1, 1, generator=g) / math.sqrt(3 * C))
    b = h(torch.randn(Co, generator=g) * 0.1)
    M = B * Fr * S
    res = h(torch.randn(M, Co, generator=g)) if resid else None
    ref = _tconv_gn_ref(x5, gamma, beta, w, b, res)
    rows = x5[..., 0].permute(0, 2, 3, 1).reshape(M, C).contiguous().half().to(gpu)
    dv = lambda t: None if t is None else t.half().to(gpu)   # noqa: E731
    wp = packing.pack_tconv3(w).half().to(gpu)
    assert ops.tconv_gn_supported(C, Co, Fr)
    out = ops.tconv_gn(rows, dv(gamma), dv(beta), wp, bias=dv(b), residual=dv(res), groups=32, B=B, F=Fr, S=S, eps=1e-5)
    close(out, ref, tol=4e-3)
    n = ops.groupnorm(rows, dv(gamma), dv(beta), groups=32, n_samples=B, rows_per_sample=Fr * S, eps=1e-5, silu_act=True)
    unfused = ops.gemm(n, wp, M=M, mode=ops.TCONV3, bias=dv(b), residual=dv(res), tconv=(Fr, S))
    close(out, unfused.float().cpu(), tol=2e-3)
    assert torch.equal(out, ops.tconv_gn(rows, dv(gamma), dv(beta), wp, bias=dv(b), residual=dv(res), groups=32, B=B, F=Fr, S=S, eps=1e-5))


def test_tconv_gn_fused_large_mean_and_batch_invariance(gpu):
    """K3 with channels whose mean is far from 0 (the statistics pass is the cancellation-free one) and a sample's bits
    whatever batch it is computed in (partition_samples pins the statistics' slab partition)."""
    ops, packing = _ops()
    g = torch.Generator().manual_seed(77)
    B, Fr, S, C, Co = 2, 16, 48, 64, 320
    x5 = h(torch.randn(B, C, Fr, S, 1, generator=g) * 0.7 + 40.0 * torch.randn(1, C, 1, 1, 1, generator=g))
    gamma, beta = h(1 + 0.2 * torch.randn(C, generator=g)), h(0.3 * torch.randn(C, generator=g))
    w = h(torch.randn(Co, C, 3, 1, 1, generator=g) / math.sqrt(3 * C))
    M = B * Fr * S
    ref = _tconv_gn_ref(x5, gamma, beta, w, None, None)
    rows = x5[..., 0].permute(0, 2, 3, 1).reshape(M, C).contiguous().half().to(gpu)
    dv = lambda t: t.half().to(gpu)   # noqa: E731
    wp = packing.pack_tconv3(w).half().to(gpu)
    both = ops.tconv_gn(rows, dv(gamma), dv(beta), wp, groups=32, B=B, F=Fr, S=S, eps=1e-5, partition_samples=B)
    close(both, ref, tol=6e-3)
    one = ops.tconv_gn(rows[M // 2:], dv(gamma), dv(beta), wp, groups=32, B=1, F=Fr, S=S, eps=1e-5, partition_samples=B)
    assert torch.equal(one, both[M // 2:])
    assert not ops.tconv_gn_supported(C, Co, 1) and not ops.tconv_gn_supported(C, Co, 20) and not ops.tconv_gn_supported(C, 256, 16)
    with pytest.raises(Exception, match="not supported"):
        ops.tconv_gn(rows[:S], dv(gamma), dv(beta), wp, groups=32, B=1, F=1, S=S, eps=1e-5)


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ns,rps,c1,c2,G,silu", [
    (6, 35, 64, 0, 32, True), (3, 300, 320, 0, 32, True), (2, 130, 64, 128, 32, True),
    (4, 16, 128, 0, 32, False), (2, 513, 1280, 640, 32, True)])
def test_groupnorm(gpu, ns, rps, c1, c2, G, silu):
    ops, _ = _ops()
    g = torch.Generator().manual_seed(ns + rps + c1)
    C = c1 + c2
    x = h(torch.randn(ns * rps, C, generator=g) * 2 + torch.randn(1, C, generator=g))
    gamma, beta = h(1 + 0.2 * torch.randn(C, generator=g)), h(0.3 * torch.randn(C, generator=g))
    eps = 1e-5
    x3 = x.reshape(ns, rps, C).permute(0, 2, 1)                      # (N, C, L)
    ref = F.group_norm(x3, G, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 1).reshape(ns * rps, C)
    xd = x.half().to(gpu)
    x1 = xd[:, :c1].contiguous()
    x2 = xd[:, c1:].contiguous() if c2 else None
    out = ops.groupnorm(x1, gamma.half().to(gpu), beta.half().to(gpu), groups=G, n_samples=ns,
                        rows_per_sample=rps, eps=eps, silu_act=silu, x2=x2)
    close(out, ref, tol=4e-3)


@pytest.mark.parametrize("ns,rps,C,mean,std", [(3, 300, 320, 60.0, 1.0), (2, 20000, 64, 1500.0, 4.0), (1, 9216 * 4, 320, -700.0, 2.0)])
def test_groupnorm_large_mean(gpu, ns, rps, C, mean, std):
    """Groups whose mean is tens to hundreds of standard deviations away from 0 (large-mean channels of trained
    checkpoints): E[x^2] - mean^2 from fp32 sums loses the variance there; the kernel's shifted sums merged with
    Chan's formula must not.  Reference = fp64 GroupNorm of the same fp16 data."""
    ops, _ = _ops()
    g = torch.Generator().manual_seed(int(abs(mean)) + rps)
    x = h(torch.randn(ns * rps, C, generator=g) * std + mean + 0.1 * std * torch.randn(1, C, generator=g))
    gamma, beta = h(1 + 0.2 * torch.randn(C, generator=g)), h(0.3 * torch.randn(C, generator=g))
    x3 = x.double().reshape(ns, rps, C).permute(0, 2, 1)
    ref = F.group_norm(x3, 32, gamma.double(), beta.double(), 1e-5).permute(0, 2, 1).reshape(ns * rps, C).float()
    out = ops.groupnorm(x.half().to(gpu), gamma.half().to(gpu), beta.half().to(gpu), groups=32, n_samples=ns,
                        rows_per_sample=rps, eps=1e-5, silu_act=False)
    close(out, ref, tol=4e-3)


@pytest.mark.parametrize("M,C", [(37, 64), (100, 320), (9, 1280), (5, 512), (4099, 320), (131, 640), (64, 1280),
                                 (70001, 320), (40003, 640), (20011, 1280),      # a wave walks several row groups
                                 (1000, 1024), (777, 768), (3, 2048), (50, 96), (33, 32)])
def test_layernorm(gpu, M, C):
    ops, _ = _ops()
    g = torch.Generator().manual_seed(M + C)
    x = h(torch.randn(M, C, generator=g) * 3 + 1)
    gamma, beta = h(1 + 0.2 * torch.randn(C, generator=g)), h(0.3 * torch.randn(C, generator=g))
    ref = F.layer_norm(x, (C,), gamma, beta, 1e-5)
    out = ops.layernorm(x.half().to(gpu), gamma.half().to(gpu), beta.half().to(gpu), M=M)
    close(out, ref, tol=4e-3)


# ---------------------------------------------------------------------------------------------
def _attn_ref(q, k, v, heads):
    n, s, _ = q.shape
    qh = q.view(n, s, heads, 64).transpose(1, 2)
    kh = k.view(n, -1, heads, 64).transpose(1, 2)
    vh = v.view(n, -1, heads, 64).transpose(1, 2)
    o = F.scaled_dot_product_attention(qh, kh, vh)
    return o.transpose(1, 2).reshape(n * s, heads * 64)


@pytest.mark.parametrize("n_seq,s,heads", [(3, 144, 2), (2, 64, 1), (2, 200, 3), (1, 576, 2), (1, 4096, 2), (5, 45, 2), (3, 4, 1),
                                           (2, 2304, 3), (4, 67, 2)])
def test_flash_self_attention_v_rows(gpu, n_seq, s, heads):
    """q, k, v as the three column blocks of one [rows][3*inner] matrix (`vdx_flash_attn_rows_f16`: V staged as rows and
    transposed by the LDS read).  Token counts that are no multiple of 8 (45 = the 40x72 latent's mid block, 4, 67)
    need no padded copy in this layout: the next image's rows / the zero page stand behind a masked key."""
    ops, _ = _ops()
    g = torch.Generator().manual_seed(n_seq + s)
    inner = heads * 64
    M = n_seq * s
    qkv = h(torch.randn(M, 3 * inner, generator=g))
    q, k, v = qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:]
    ref = _attn_ref(q.reshape(n_seq, s, inner), k.reshape(n_seq, s, inner), v.reshape(n_seq, s, inner), heads)
    qkv_d = qkv.half().to(gpu)
    out = ops.flash_attn(qkv_d[:, :inner], qkv_d[:, inner:2 * inner], qkv_d[:, 2 * inner:], n_seq=n_seq, sq=s, skv=s,
                         skv_pad=s, heads=heads, seq_per_kv=1, scale=0.125, v_rows=True)
    close(out, ref, tol=4e-3)
    # bit-identical to the V^T form: same products in the same order, only the way V reaches the operand differs
    Mp = ops.round_up(M, 64)
    if s % 8 == 0:
        vt = torch.zeros(inner, Mp, dtype=torch.float16, device=gpu)
        vt[:, :M] = qkv_d[:, 2 * inner:].t()
        out_t = ops.flash_attn(qkv_d[:, :inner], qkv_d[:, inner:2 * inner], vt, n_seq=n_seq, sq=s, skv=s, skv_pad=s,
                               heads=heads, seq_per_kv=1, scale=0.125)
        assert torch.equal(out, out_t)


def test_flash_v_rows_last_rows_of_the_buffer(gpu):
    """The last image's last tile reaches past the buffer's rows: those come from the zero page, never from memory
    behind the allocation (the V rows sit at the very end of a caching-allocator block here; NaNs behind a masked
    key would poison the sums)."""
    ops, _ = _ops()
    g = torch.Generator().manual_seed(77)
    s, heads, n_seq = 100, 1, 2
    big = torch.full((n_seq * s + 64, 192), float("nan"), dtype=torch.float16, device=gpu)
    qkv = h(torch.randn(n_seq * s, 192, generator=g))
    big[:n_seq * s] = qkv.half().to(gpu)
    view = big[:n_seq * s]
    ref = _attn_ref(qkv[:, :64].reshape(n_seq, s, 64), qkv[:, 64:128].reshape(n_seq, s, 64), qkv[:, 128:].reshape(n_seq, s, 64), heads)
    out = ops.flash_attn(view[:, :64], view[:, 64:128], view[:, 128:], n_seq=n_seq, sq=s, skv=s, skv_pad=s, heads=heads,
                         seq_per_kv=1, scale=0.125, v_rows=True)
    assert bool(torch.isfinite(out.float()).all())
    close(out, ref, tol=4e-3)


@pytest.mark.parametrize("n_seq,s,heads", [(3, 144, 2), (2, 64, 1), (2, 200, 3), (1, 576, 2), (1, 4096, 2)])
def test_flash_self_attention(gpu, n_seq, s, heads):
    ops, _ = _ops()
    g = torch.Generator().manual_seed(n_seq + s)
    inner = heads * 64
    M = n_seq * s
    Mp = ops.round_up(M, 64)
    qkv = h(torch.randn(M, 3 * inner, generator=g))
    q, k, v = qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:]
    ref = _attn_ref(q.reshape(n_seq, s, inner), k.reshape(n_seq, s, inner), v.reshape(n_seq, s, inner), heads)
    qkv_d = qkv.half().to(gpu)
    vt = torch.zeros(inner, Mp, dtype=torch.float16, device=gpu)
    vt[:, :M] = v.t().half().to(gpu)
    out = ops.flash_attn(qkv_d[:, :inner], qkv_d[:, inner:2 * inner], vt, n_seq=n_seq, sq=s, skv=s, skv_pad=s,
                         heads=heads, seq_per_kv=1, scale=0.125)
    close(out, ref, tol=4e-3)


def test_flash_attention_large_scores_online_rescale(gpu):
    """Forces the running max to jump at a late tile (one key row spiked against every query)."""
    ops, _ = _ops()
    g = torch.Generator().manual_seed(11)
    s, heads = 320, 1
    q = h(torch.randn(s, 64, generator=g))
    k = h(torch.randn(s, 64, generator=g))
    v = h(torch.randn(s, 64, generator=g))
    k[300] = h(q.mean(0) * 40 + 3)
    ref = _attn_ref(q[None], k[None], v[None], heads)
    vt = v.t().contiguous().half().to(gpu)
    out = ops.flash_attn(q.half().to(gpu), k.half().to(gpu), vt, n_seq=1, sq=s, skv=s, skv_pad=s, heads=heads,
                         seq_per_kv=1, scale=0.125)
    close(out, ref, tol=4e-3)


@pytest.mark.parametrize("case", ["all_very_negative", "all_very_positive", "late_spike_up", "drift_down_then_up",
                                  "all_minus_150_nat", "tile0_spike_plus_150_nat", "falling_60_nat_per_tile",
                                  "rising_60_nat_per_tile", "first_tile_low_then_window", "all_plus_2500_nat",
                                  "all_minus_2500_nat", "rising_700_nat_per_tile"])
@pytest.mark.parametrize("s", [320, 640])
def test_flash_attention_lazy_offset_branches(gpu, case, s):
    """The softmax offset is lazy (kept at 0 while row maxima of the scaled scores stay in (-4, 10]).
    These inputs FORCE every branch of that logic — rows that start far below the window, far above
    it, jump late, or drift — on both the 32- and 64-queries-per-wave kernels (s = 320 / 640); a full
    fp32 reference checks every output element (a rescale bug is silent: no NaN)."""
    ops, _ = _ops()
    g = torch.Generator().manual_seed(len(case) + s)
    q = h(torch.randn(s, 64, generator=g))
    k = h(torch.randn(s, 64, generator=g))
    v = h(torch.randn(s, 64, generator=g))
    u = q.mean(0) / q.mean(0).norm()
    if case == "all_very_negative":
        q = h(q + 6 * u)
        k = h(k - 6 * u)                       # every score ~ -36*... strongly negative
    elif case == "all_very_positive":
        q = h(q + 9 * u)
        k = h(k + 9 * u)                       # scaled scores ~ +14: above the window from tile 0
    elif case == "late_spike_up":
        k[s - 20] = h(q.mean(0) * 40 + 3)      # one key far above the window in the last tile
    elif case in ("all_minus_150_nat", "tile0_spike_plus_150_nat", "falling_60_nat_per_tile",
                  "rising_60_nat_per_tile", "first_tile_low_then_window", "all_plus_2500_nat", "all_minus_2500_nat",
                  "rising_700_nat_per_tile"):
        # Channel 0 carries an exact per-key shift of the scaled score: q[:,0] = 8 (4 on odd rows) and
        # scale 1/8 make the score of key j move by b[j] (b[j]/2) nat.  Softmax is shift-invariant, so the
        # fp32 reference stays finite on every one of these; a kernel whose offset can move DOWN by more
        # than 128 exp2-units (or that rescales 0 by 2^big) returns NaN here (round-1 advisor finding).
        tile = torch.arange(s) // 64
        b = {"all_minus_150_nat": torch.full((s,), -150.0),
             "tile0_spike_plus_150_nat": torch.where(torch.arange(s) == 5, 150.0, 0.0),
             "falling_60_nat_per_tile": -60.0 * tile,
             "rising_60_nat_per_tile": 60.0 * tile,
             "first_tile_low_then_window": torch.where(tile == 0, -40.0, 0.0),
             # offsets beyond 2048 exp2-units, where the fp16 the offset is carried in has a spacing of 2 and 4: the
             # re-centred running maximum then lands within +-1 / +-2 of zero, inside the window (include/vdx.h states
             # the supported range: |score| <= 5000 nat)
             "all_plus_2500_nat": torch.full((s,), 2500.0), "all_minus_2500_nat": torch.full((s,), -2500.0),
             "rising_700_nat_per_tile": 700.0 * tile}[case]
        q[:, 0] = torch.where(torch.arange(s) % 2 == 0, 8.0, 4.0)
        k[:, 0] = b
    else:
        ramp = torch.linspace(-8, 8, s)[:, None]
        q = h(q + 4 * u)
        k = h(k + ramp * u)                    # maxima drift from below the window to above it
    ref = _attn_ref(q[None], k[None], v[None], 1)
    out = ops.flash_attn(q.half().to(gpu), k.half().to(gpu), v.t().contiguous().half().to(gpu), n_seq=1, sq=s,
                         skv=s, skv_pad=s, heads=1, seq_per_kv=1, scale=0.125)
    assert bool(torch.isfinite(out.float()).all()), "flash attention produced non-finite values"
    close(out, ref, tol=4e-3)
    out_r = ops.flash_attn(q.half().to(gpu), k.half().to(gpu), v.half().to(gpu), n_seq=1, sq=s, skv=s, skv_pad=s, heads=1,
                           seq_per_kv=1, scale=0.125, v_rows=True)
    assert torch.equal(out_r, out)


@pytest.mark.parametrize("B,Fr,s,heads,skv", [(2, 3, 144, 2, 77), (1, 4, 64, 1, 77), (2, 2, 100, 2, 5)])
def test_flash_cross_attention(gpu, B, Fr, s, heads, skv):
    ops, _ = _ops()
    g = torch.Generator().manual_seed(B + Fr + s)
    inner = heads * 64
    skv_pad = 128
    q = h(torch.randn(B * Fr, s, inner, generator=g))
    k = h(torch.randn(B, skv, inner, generator=g))
    v = h(torch.randn(B, skv, inner, generator=g))
    ref = _attn_ref(q, k.repeat_interleave(Fr, 0), v.repeat_interleave(Fr, 0), heads)
    kd = torch.zeros(B * skv_pad, inner, dtype=torch.float16, device=gpu)
    vt = torch.zeros(inner, B * skv_pad, dtype=torch.float16, device=gpu)
    for b in range(B):
        kd[b * skv_pad:b * skv_pad + skv] = k[b].half().to(gpu)
        vt[:, b * skv_pad:b * skv_pad + skv] = v[b].t().half().to(gpu)
    out = ops.flash_attn(q.reshape(-1, inner).half().to(gpu), kd, vt, n_seq=B * Fr, sq=s, skv=skv, skv_pad=skv_pad,
                         heads=heads, seq_per_kv=Fr, scale=0.125)
    close(out, ref, tol=4e-3)
    vd = torch.zeros(B * skv_pad, inner, dtype=torch.float16, device=gpu)
    for b in range(B):
        vd[b * skv_pad:b * skv_pad + skv] = v[b].half().to(gpu)
    out_r = ops.flash_attn(q.reshape(-1, inner).half().to(gpu), kd, vd, n_seq=B * Fr, sq=s, skv=skv, skv_pad=skv_pad,
                           heads=heads, seq_per_kv=Fr, scale=0.125, v_rows=True)
    assert torch.equal(out_r, out)


@pytest.mark.parametrize("B,Fr,HW,heads", [(2, 5, 7, 2), (1, 24, 12, 5), (2, 32, 3, 1), (1, 1, 9, 2), (2, 16, 130, 8),
                                           (1, 33, 5, 2), (2, 48, 6, 1), (1, 96, 3, 2), (1, 128, 2, 1)])
def test_temporal_attention(gpu, B, Fr, HW, heads):
    ops, _ = _ops()
    g = torch.Generator().manual_seed(B + Fr + HW)
    inner = heads * 64
    qkv = h(torch.randn(B * Fr * HW, 3 * inner, generator=g))
    # sequences over frames for every (b, pixel)
    t = qkv.reshape(B, Fr, HW, 3 * inner).permute(0, 2, 1, 3).reshape(B * HW, Fr, 3 * inner)
    o = _attn_ref(t[..., :inner], t[..., inner:2 * inner], t[..., 2 * inner:], heads)
    ref = o.reshape(B, HW, Fr, inner).permute(0, 2, 1, 3).reshape(B * Fr * HW, inner)
    out = ops.temporal_attn(qkv.half().to(gpu), B=B, F=Fr, HW=HW, heads=heads, scale=0.125)
    close(out, ref, tol=4e-3)


def _temporal_block_ref(t, gamma, beta, wq, wk, wv, wo, bo, B, Fr, HW, heads):
    """fp32 statement of `s = s + attn(LN(s))` of TransformerTemporalModel on rows [(b*F+f)*HW+p][inner]."""
    inner = t.shape[1]
    ln = F.layer_norm(t, (inner,), gamma, beta, 1e-5)
    q, k, v = ln @ wq.t(), ln @ wk.t(), ln @ wv.t()
    seq = lambda x: x.reshape(B, Fr, HW, inner).permute(0, 2, 1, 3).reshape(B * HW, Fr, inner)   # noqa: E731
    o = _attn_ref(seq(q), seq(k), seq(v), heads)
    o = o.reshape(B, HW, Fr, inner).permute(0, 2, 1, 3).reshape(B * Fr * HW, inner)
    return t + o @ wo.t() + bo


@pytest.mark.parametrize("inner,B,Fr,HW", [(320, 2, 24, 20), (320, 1, 16, 7), (320, 2, 12, 9), (320, 1, 8, 5),
                                           (320, 1, 24, 1), (320, 2, 6, 33), (320, 1, 48, 3), (320, 1, 1, 100),
                                           (512, 2, 24, 10), (512, 1, 16, 6), (512, 2, 12, 13), (512, 1, 3, 50)])
def test_temporal_attn_block_fused(gpu, inner, B, Fr, HW):
    """K7 (csrc/tattn_fused.hip): LayerNorm -> q|k|v -> attention over the frames of each pixel -> to_out + bias +
    residual in one kernel, against the fp32 reference and against the un-fused kernels it replaces.  Shapes cover
    F in {24, 16, 12} (the BASELINE chunks) and other divisors of 48, pixel counts that do not fill the last row
    group / the last block, and both widths the kernel is built for."""
    ops, _ = _ops()
    from vdx import packing
    heads = inner // 64
    g = torch.Generator().manual_seed(inner + Fr + HW)
    M = B * Fr * HW
    t = h(torch.randn(M, inner, generator=g) * 1.5 + 0.3)
    gamma, beta = h(1 + 0.2 * torch.randn(inner, generator=g)), h(0.1 * torch.randn(inner, generator=g))
    wq, wk, wv, wo = (h(torch.randn(inner, inner, generator=g) * s_) for s_ in (0.09, 0.09, 0.06, 0.05))
    bo = h(0.1 * torch.randn(inner, generator=g))
    ref = _temporal_block_ref(t, gamma, beta, wq, wk, wv, wo, bo, B, Fr, HW, heads)
    d = lambda x: x.half().to(gpu)   # noqa: E731
    assert ops.temporal_attn_block_supported(inner, Fr)
    out = ops.temporal_attn_block(d(t), d(gamma), d(beta), d(packing.pack_k7_qkv(wq, wk, wv)), d(packing.pack_k7_out(wo)),
                                  d(bo), B=B, F=Fr, HW=HW, scale=0.125)
    close(out, ref, tol=4e-3)
    # the un-fused chain (LayerNorm, q|k|v GEMM, attention core, output GEMM with residual)
    ln = ops.layernorm(d(t), d(gamma), d(beta), M=M)
    qkv = ops.gemm(ln, d(torch.cat([wq, wk, wv], 0)), M=M)
    o = ops.temporal_attn(qkv, B=B, F=Fr, HW=HW, heads=heads, scale=0.125)
    unf = ops.gemm(o, d(wo), M=M, bias=d(bo), residual=d(t))
    close(out, unf.float().cpu(), tol=3e-3)
    # a second launch gives the same bits (no dependence on block scheduling)
    out2 = ops.temporal_attn_block(d(t), d(gamma), d(beta), d(packing.pack_k7_qkv(wq, wk, wv)), d(packing.pack_k7_out(wo)),
                                   d(bo), B=B, F=Fr, HW=HW, scale=0.125)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("B,Fr,HW", [(2, 24, 20), (1, 16, 7), (2, 12, 9), (1, 8, 5), (1, 24, 1), (2, 6, 33), (1, 48, 3),
                                     (1, 1, 100), (2, 24, 301), (1, 16, 515), (2, 3, 64), (1, 2, 777),
                                     (2, 24, 1100), (1, 12, 4100), (3, 16, 2000)])
def test_temporal_attn_block2_fused(gpu, B, Fr, HW):
    """K7, second design (csrc/tattn2.hip, inner 320): LayerNorm (folded into the packed weights) -> q|k -> scores ->
    v -> P.V -> to_out + bias + residual in one kernel, against the fp32 statement of the sub-block and
    against the un-fused kernels.  F in {24, 16, 12} (the BASELINE chunks) and other divisors of 48; pixel counts that do
    not fill the last row group / the last block, counts that give many blocks (every rotation of the head order), and
    counts of more than 256 tiles (a workgroup then walks several tiles: the next tile's rows are fetched and normalised
    behind the output projection of the current one, the last round is ragged)."""
    ops, _ = _ops()
    from vdx import packing
    inner, heads = 320, 5
    g = torch.Generator().manual_seed(inner + Fr + HW)
    M = B * Fr * HW
    t = h(torch.randn(M, inner, generator=g) * 1.5 + 0.3)
    gamma, beta = h(1 + 0.2 * torch.randn(inner, generator=g)), h(0.1 * torch.randn(inner, generator=g))
    wq, wk, wv, wo = (h(torch.randn(inner, inner, generator=g) * s_) for s_ in (0.09, 0.09, 0.06, 0.05))
    bo = h(0.1 * torch.randn(inner, generator=g))
    ref = _temporal_block_ref(t, gamma, beta, wq, wk, wv, wo, bo, B, Fr, HW, heads)
    d = lambda x: x.half().to(gpu)   # noqa: E731
    assert ops.temporal_attn_block2_supported(inner, Fr)
    blob = packing.pack_k7b(wq, wk, wv, wo, gamma, beta, bo, 0.125).to(gpu)
    out = ops.temporal_attn_block2(d(t), blob, B=B, F=Fr, HW=HW)
    close(out, ref, tol=4e-3)
    ln = ops.layernorm(d(t), d(gamma), d(beta), M=M)
    qkv = ops.gemm(ln, d(torch.cat([wq, wk, wv], 0)), M=M)
    o = ops.temporal_attn(qkv, B=B, F=Fr, HW=HW, heads=heads, scale=0.125)
    unf = ops.gemm(o, d(wo), M=M, bias=d(bo), residual=d(t))
    close(out, unf.float().cpu(), tol=4e-3)
    # a second launch gives the same bits (no dependence on block scheduling); a strided input / output too
    out2 = ops.temporal_attn_block2(d(t), blob, B=B, F=Fr, HW=HW)
    assert torch.equal(out, out2)
    wide_in = torch.zeros(M, inner + 64, dtype=torch.float16, device=gpu)
    wide_in[:, :inner] = d(t)
    wide_out = torch.full((M, inner + 8), 7.0, dtype=torch.float16, device=gpu)
    ops.temporal_attn_block2(wide_in[:, :inner], blob, B=B, F=Fr, HW=HW, out=wide_out[:, :inner])
    assert torch.equal(wide_out[:, :inner], out) and bool((wide_out[:, inner:] == 7.0).all())


def _cross_block_ref(t, gamma, beta, wq, wk, wv, wo, bo, ehs, n_items, rows, heads):
    """fp32 statement of `t + attn2(norm2(t), encoder_hidden_states)` per batch item."""
    inner = t.shape[1]
    ln = F.layer_norm(t, (inner,), gamma, beta, 1e-5)
    q = (ln @ wq.t()).reshape(n_items, rows, heads, 64).permute(0, 2, 1, 3)
    k = (ehs @ wk.t()).reshape(n_items, -1, heads, 64).permute(0, 2, 1, 3)
    v = (ehs @ wv.t()).reshape(n_items, -1, heads, 64).permute(0, 2, 1, 3)
    p = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(n_items * rows, inner)
    return t + o @ wo.t() + bo


@pytest.mark.parametrize("n_items,rows,kv_len", [(2, 2 * 9216, 77), (1, 1000, 77), (3, 192, 5), (2, 50, 80), (1, 300 * 192, 77), (2, 4 * 9216 + 64, 33)])
def test_cross_attn_block_fused(gpu, n_items, rows, kv_len):
    """K5 (csrc/xattn.hip, inner 320): LayerNorm (folded into the packed weights) -> q -> scores against the item's text
    keys -> softmax -> P.V -> to_out + bias + residual in one kernel, against the fp32 statement of the sub-block (SURVEY
    A.5) and against the un-fused kernels (LayerNorm, q GEMM, flash attention on the padded text, output GEMM).  Texts of
    77 tokens (CLIP), of 5, of 33 and of the full 80 slots; row counts that fill no tile, exactly one, many (a workgroup walks
    several: the next tile's rows are fetched behind the projection), items whose last tile is ragged."""
    ops, packing = _ops()
    inner, heads, cross, pad = 320, 5, 128, 128
    g = torch.Generator().manual_seed(n_items + rows + kv_len)
    M = n_items * rows
    t = h(torch.randn(M, inner, generator=g) * 1.5 + 0.3)
    gamma, beta = h(1 + 0.2 * torch.randn(inner, generator=g)), h(0.1 * torch.randn(inner, generator=g))
    wq, wo = h(torch.randn(inner, inner, generator=g) * 0.09), h(torch.randn(inner, inner, generator=g) * 0.05)
    wk, wv = h(torch.randn(inner, cross, generator=g) * 0.12), h(torch.randn(inner, cross, generator=g) * 0.09)
    bo = h(0.1 * torch.randn(inner, generator=g))
    ehs = h(torch.randn(n_items, kv_len, cross, generator=g))
    d = lambda x: x.half().to(gpu)   # noqa: E731
    # the text keys / values as the un-fused path keeps them: k rows of the zero-padded text, V^T
    ehs_pad = torch.zeros(n_items * pad, cross)
    ehs_pad.view(n_items, pad, cross)[:, :kv_len] = ehs
    k_rows = ops.gemm(d(ehs_pad), d(wk), M=n_items * pad)
    vt = ops.gemm(d(wv), d(ehs_pad), M=inner)
    ref = _cross_block_ref(t, gamma, beta, wq, wk, wv, wo, bo, ehs, n_items, rows, heads)
    assert ops.cross_attn_block_supported(inner, kv_len) and not ops.cross_attn_block_supported(inner, 81) and not ops.cross_attn_block_supported(640, 77)
    blob = packing.pack_k5(wq, wo, gamma, beta, bo, 0.125).to(gpu)
    kvb = packing.pack_k5_kv(k_rows, vt, n_items, pad)
    out = ops.cross_attn_block(d(t), blob, kvb, kv_len=kv_len, n_items=n_items, rows_per_item=rows)
    close(out, ref, tol=5e-3)
    ln = ops.layernorm(d(t), d(gamma), d(beta), M=M)
    q = ops.gemm(ln, d(wq), M=M)
    o = ops.flash_attn(q, k_rows, vt, n_seq=n_items, sq=rows, skv=kv_len, skv_pad=pad, heads=heads, seq_per_kv=1, scale=0.125)
    unf = ops.gemm(o, d(wo), M=M, bias=d(bo), residual=d(t))
    close(out, unf.float().cpu(), tol=5e-3)
    out2 = ops.cross_attn_block(d(t), blob, kvb, kv_len=kv_len, n_items=n_items, rows_per_item=rows)
    assert torch.equal(out, out2)
    # an item's rows carry the same bits wherever the item sits in the batch (tiles are aligned to items; the head order
    # is a function of the tile's position inside its item)
    if n_items > 1:
        last = ops.cross_attn_block(d(t)[(n_items - 1) * rows:], blob, kvb[n_items - 1:].contiguous(), kv_len=kv_len, n_items=1, rows_per_item=rows)
        assert torch.equal(last, out[(n_items - 1) * rows:])
    wide_in = torch.zeros(M, inner + 64, dtype=torch.float16, device=gpu)
    wide_in[:, :inner] = d(t)
    wide_out = torch.full((M, inner + 8), 7.0, dtype=torch.float16, device=gpu)
    ops.cross_attn_block(wide_in[:, :inner], blob, kvb, kv_len=kv_len, n_items=n_items, rows_per_item=rows, out=wide_out[:, :inner])
    assert torch.equal(wide_out[:, :inner], out) and bool((wide_out[:, inner:] == 7.0).all())


def test_cross_attn_block_run_to_run_bits_at_full_size(gpu):
    """Nine tiles per workgroup, every wave's counted waits under load, the key / value units of two items streaming through
    the ring beside the weights: the bits must repeat (a wait that counts more vector-memory instructions than the ISA holds
    would let a unit be read before it has landed — the failure shows as a run-to-run difference at full size only)."""
    ops, packing = _ops()
    inner, pad, kv_len, n_items, rows = 320, 128, 77, 2, 24 * 9216
    g = torch.Generator(device=gpu).manual_seed(9)
    r = lambda *sh, k=1.0: (torch.randn(*sh, device=gpu, generator=g) * k).half()      # noqa: E731
    blob = packing.pack_k5(r(inner, inner, k=0.09), r(inner, inner, k=0.05), r(inner, k=0.2) + 1, r(inner, k=0.1), r(inner, k=0.1), 0.125)
    kvb = packing.pack_k5_kv(r(n_items * pad, inner), r(inner, n_items * pad), n_items, pad)
    t = r(n_items * rows, inner, k=1.5)
    first = ops.cross_attn_block(t, blob, kvb, kv_len=kv_len, n_items=n_items, rows_per_item=rows).clone()
    assert bool(torch.isfinite(first.float()).all())
    for _ in range(6):
        assert torch.equal(ops.cross_attn_block(t, blob, kvb, kv_len=kv_len, n_items=n_items, rows_per_item=rows), first)


@pytest.mark.parametrize("Fr,HW", [(24, 1100), (16, 37), (12, 700)])
def test_temporal_attn_block2_batch_invariant(gpu, Fr, HW):
    """A sample's result has the same bits wherever it sits in the batch and whatever else is in it: tiles are aligned
    to batch items and the order in which a tile sums the heads is a function of its position inside the item (the
    persistent grid, the block that happens to own a tile and the batch size do not enter)."""
    ops, _ = _ops()
    from vdx import packing
    inner = 320
    g = torch.Generator().manual_seed(Fr + HW)
    t1 = h(torch.randn(Fr * HW, inner, generator=g) * 1.5 + 0.3).half()
    t2 = h(torch.randn(Fr * HW, inner, generator=g)).half()
    gamma, beta = h(1 + 0.2 * torch.randn(inner, generator=g)), h(0.1 * torch.randn(inner, generator=g))
    wq, wk, wv, wo = (h(torch.randn(inner, inner, generator=g) * s_) for s_ in (0.09, 0.09, 0.06, 0.05))
    bo = h(0.1 * torch.randn(inner, generator=g))
    blob = packing.pack_k7b(wq, wk, wv, wo, gamma, beta, bo, 0.125).to(gpu)
    alone = ops.temporal_attn_block2(t1.to(gpu), blob, B=1, F=Fr, HW=HW)
    three = ops.temporal_attn_block2(torch.cat([t2, t1, t1]).to(gpu), blob, B=3, F=Fr, HW=HW)
    M = Fr * HW
    assert torch.equal(three[M:2 * M], alone) and torch.equal(three[2 * M:], alone)


def test_temporal_attn_block2_large_mean(gpu):
    """Rows whose mean is far from zero (|mean| = 40 sigma): the statistics are two-pass fp32, the centred row is
    formed as x * rstd - mean * rstd in fp32 before the fp16 rounding."""
    ops, _ = _ops()
    from vdx import packing
    inner, heads, B, Fr, HW = 320, 5, 1, 24, 16
    g = torch.Generator().manual_seed(5)
    M = B * Fr * HW
    t = h(torch.randn(M, inner, generator=g) * 0.5 + 20.0)
    gamma, beta = h(1 + 0.2 * torch.randn(inner, generator=g)), h(0.1 * torch.randn(inner, generator=g))
    wq, wk, wv, wo = (h(torch.randn(inner, inner, generator=g) * s_) for s_ in (0.09, 0.09, 0.06, 0.05))
    bo = h(0.1 * torch.randn(inner, generator=g))
    ref = _temporal_block_ref(t, gamma, beta, wq, wk, wv, wo, bo, B, Fr, HW, heads)
    blob = packing.pack_k7b(wq, wk, wv, wo, gamma, beta, bo, 0.125).to(gpu)
    out = ops.temporal_attn_block2(t.half().to(gpu), blob, B=B, F=Fr, HW=HW)
    close(out, ref, tol=4e-3)


def test_temporal_attn_block_rejects_unsupported(gpu):
    ops, _ = _ops()
    from vdx._lib import VdxError
    assert not ops.temporal_attn_block_supported(320, 5) and not ops.temporal_attn_block_supported(640, 24)
    assert not ops.temporal_attn_block2_supported(512, 24) and not ops.temporal_attn_block2_supported(320, 5)
    with pytest.raises(VdxError):
        ops.temporal_attn_block2(torch.zeros(20, 320, dtype=torch.float16, device=gpu), torch.zeros(16, dtype=torch.float16, device=gpu), B=1, F=4, HW=5)
    t = torch.zeros(5 * 4, 320, dtype=torch.float16, device=gpu)
    v = torch.zeros(320, dtype=torch.float16, device=gpu)
    w = torch.zeros(16, dtype=torch.float16, device=gpu)
    with pytest.raises(VdxError):
        ops.temporal_attn_block(t, v, v, w, w, v, B=1, F=5, HW=4, scale=0.125)


# ---------------------------------------------------------------------------------------------
def test_conv_in_and_output_permute(gpu):
    ops, packing = _ops()
    g = torch.Generator().manual_seed(3)
    B, Cin, Fr, H, W, Cout = 2, 4, 3, 6, 10, 64
    x = h(torch.randn(B, Cin, Fr, H, W, generator=g))
    w = h(torch.randn(Cout, Cin, 3, 3, generator=g) / 6)
    b = h(torch.randn(Cout, generator=g) * 0.1)
    x4 = x.permute(0, 2, 1, 3, 4).reshape(B * Fr, Cin, H, W)
    ref = packing.nchw_to_rows(F.conv2d(x4, w, b, padding=1))
    out = ops.conv_in(x.half().to(gpu), packing.pack_conv_in(w.half()).to(gpu), b.half().to(gpu))
    close(out, ref)
    # rows -> (B,C,F,H,W) takes the first C columns
    back = ops.rows_to_ncfhw(out, B, 4, Fr, H, W)
    want = out[:, :4].float().cpu().reshape(B, Fr, H, W, 4).permute(0, 4, 1, 2, 3)
    assert torch.equal(back.float().cpu(), want)


def test_silu(gpu):
    ops, _ = _ops()
    x = h(torch.randn(2, 1280))
    close(ops.silu(x.half().to(gpu)), F.silu(x))


# ---- orchestration kernels: bit-exact against the oracle -----------------------------------------
def test_cfg_input_bit_exact(gpu):
    ops, _ = _ops()
    from oracle.pipeline_ref import base_noise, global_context
    T, C, H, W = 6, 4, 8, 8
    lat = base_noise(T, C, H, W)
    ctx = global_context(T, C, H, W)
    want = torch.cat([lat] * 2) + 0.35 * ctx.repeat(1, 1, T, 1, 1)     # fsdp_chunked_coherent.py:133-137
    got = ops.cfg_input(lat.to(gpu), ctx.to(gpu), 0.35)
    assert torch.equal(got.cpu(), want)
    got = ops.cfg_input(lat.to(gpu), None, 0.35)
    assert torch.equal(got.cpu(), torch.cat([lat] * 2))


@pytest.mark.parametrize("steps", [10, 50])
def test_cfg_ddim_step_bit_exact(gpu, steps):
    ops, _ = _ops()
    from oracle.ddim_ref import DDIMSchedulerRef
    s = DDIMSchedulerRef()
    s.set_timesteps(steps)
    g = torch.Generator().manual_seed(steps)
    lat = torch.randn(1, 4, 5, 8, 8, generator=g).half()
    for t in s.timesteps[[0, 1, steps // 2, steps - 1]]:
        noise = torch.randn(2, 4, 5, 8, 8, generator=g).half()
        u, c = noise.chunk(2)
        guided = u + 7.5 * (c - u)                                      # :141 (CPU and GPU rules agree)
        want = s.step(guided, t, lat).prev_sample                     # :142, torch-CPU evaluation
        want_gpu = s.step_gpu_rules(guided, t, lat).prev_sample       # :142, torch-GPU type rules
        got = ops.cfg_ddim_step(noise.to(gpu), lat.to(gpu), 7.5, s.coefficients(int(t))).cpu()
        assert want.dtype == torch.float16
        assert torch.equal(got, want_gpu), f"t={int(t)}: {(got.float() - want_gpu.float()).abs().max()}"
        assert torch.equal(ops.ddim_step(guided.to(gpu), lat.to(gpu), s.coefficients(int(t))).cpu(), want_gpu)
        # vs the CPU evaluation: the two rule differences move intermediates by <= 1 fp16 ulp; the
        # final sum can cancel, so the bound is absolute: 4 ulps of the largest intermediate (x0)
        x0 = s.step(guided, t, lat).pred_original_sample
        scale = max(float(want.float().abs().max()), float(x0.float().abs().max()))
        assert float((got.float() - want.float()).abs().max()) <= 4 * 2.0 ** -10 * scale
        lat = want


@pytest.mark.parametrize("hin,win,hout,wout", [(5, 8, 9, 15), (9, 15, 18, 30), (3, 3, 5, 6), (4, 4, 8, 8), (7, 2, 7, 3)])
def test_conv3x3_upsample_to_size(gpu, hin, win, hout, wout):
    """conv3x3 over a source nearest-upsampled to an explicit size (vdx_gemm_args.upsample = 2) against
    F.interpolate(size=, mode="nearest") + F.conv2d."""
    ops, packing = _ops()
    g = torch.Generator().manual_seed(hin * 31 + wout)
    n, cin, cout = 3, 64, 128
    x = h(torch.randn(n, cin, hin, win, generator=g))
    w = h(torch.randn(cout, cin, 3, 3, generator=g) / 24)
    b = h(torch.randn(cout, generator=g) * 0.1)
    ref = packing.nchw_to_rows(F.conv2d(F.interpolate(x, size=(hout, wout), mode="nearest"), w, b, padding=1))
    out = ops.gemm(packing.nchw_to_rows(x).half().to(gpu), packing.pack_conv3x3(w.half()).to(gpu), M=n * hout * wout,
                   mode=ops.CONV3X3, bias=b.half().to(gpu), conv=(n, hin, win, hout, wout, 1, 2))
    close(out, ref)


def test_split_k_refuses_upsample_to_size_and_short_workspaces(gpu):
    """A pinned ksplit on the nearest-to-size gather is an error (no split-K instantiation carries the size map: ADVICE r3),
    and a workspace smaller than tiles x slices x 327 680 bytes is refused instead of overrun.  The automatic path
    (allow_ksplit) on a to-size convolution large enough for the planner to consider a split gives the un-split result."""
    ops, packing = _ops()
    from vdx import _lib
    g = torch.Generator().manual_seed(11)
    n, cin, cout, hin, win, hout, wout = 2, 64, 320, 5, 8, 9, 15
    x = h(torch.randn(n, cin, hin, win, generator=g))
    w = h(torch.randn(cout, cin, 3, 3, generator=g) / 24)
    kw = dict(M=n * hout * wout, mode=ops.CONV3X3, conv=(n, hin, win, hout, wout, 1, 2))
    xr, wp = packing.nchw_to_rows(x).half().to(gpu), packing.pack_conv3x3(w.half()).to(gpu)
    with pytest.raises(_lib.VdxError, match="upsample"):
        ops.gemm(xr, wp, ksplit=2, **kw)
    ref = packing.nchw_to_rows(F.conv2d(F.interpolate(x, size=(hout, wout), mode="nearest"), w, padding=1))
    close(ops.gemm(xr, wp, allow_ksplit=True, **kw), ref)
    # the advisor's shape through the automatic path: 48 images 8x22 -> 16x43, N = 640 (the planner used to split it)
    n, cin, cout, hin, win, hout, wout = 48, 640, 640, 8, 22, 16, 43
    x = h(torch.randn(n, cin, hin, win, generator=g))
    w = h(torch.randn(cout, cin, 3, 3, generator=g) / 76)
    xr, wp = packing.nchw_to_rows(x).half().to(gpu), packing.pack_conv3x3(w.half()).to(gpu)
    kw = dict(M=n * hout * wout, mode=ops.CONV3X3, conv=(n, hin, win, hout, wout, 1, 2))
    ref = packing.nchw_to_rows(F.conv2d(F.interpolate(x, size=(hout, wout), mode="nearest"), w, padding=1))
    out = ops.gemm(xr, wp, allow_ksplit=True, **kw)
    close(out, ref)
    assert torch.equal(out, ops.gemm(xr, wp, **kw))
    # short workspace: the C entry point checks the size it is told
    import ctypes as C
    M, N, K = 2048, 320, 1280
    a = torch.zeros(M, K, dtype=torch.float16, device=gpu)
    wz = torch.zeros(N, K, dtype=torch.float16, device=gpu)
    o = torch.empty(M, N, dtype=torch.float16, device=gpu)
    ws = torch.empty(8 * 2 * 327680 // 4 - 4, dtype=torch.float32, device=gpu)
    ga = _lib.GemmArgs()
    ga.a, ga.w, ga.out, ga.M, ga.N, ga.K, ga.c1, ga.lda, ga.ldo = a.data_ptr(), wz.data_ptr(), o.data_ptr(), M, N, K, K, K, N
    ga.ksplit, ga.workspace, ga.workspace_bytes = 2, ws.data_ptr(), ws.numel() * 4
    lib = _lib.load()
    assert lib.vdx_gemm_f16(C.byref(ga), None) != 0 and b"workspace" in lib.vdx_last_error()


def test_scheduler_step_uses_the_timesteps_value(gpu):
    """`DDIMScheduler.step` takes the timestep by VALUE (diffusers semantics), also out of sequence: elements of
    `scheduler.timesteps` in any order (recognised by their storage address: no device read), copies of them and a
    foreign device tensor (read with one sync), against host integers."""
    _ops()
    from vdx.scheduler import DDIMScheduler
    s = DDIMScheduler()
    s.set_timesteps(50, device=gpu)
    g = torch.Generator().manual_seed(3)
    eps = torch.randn(1, 4, 3, 8, 8, generator=g).half().to(gpu)
    lat = torch.randn(1, 4, 3, 8, 8, generator=g).half().to(gpu)
    host = [int(v) for v in s.timesteps.cpu()]
    want = {t: s.step(eps, t, lat).prev_sample for t in set(host)}
    for i in (0, 5, 2, 49, 5):                               # out of sequence, repeated
        assert torch.equal(s.step(eps, s.timesteps[i], lat).prev_sample, want[host[i]])
    for k, t in enumerate(s.timesteps[3:6]):                 # iteration over a slice: views of the same storage
        assert torch.equal(s.step(eps, t, lat).prev_sample, want[host[3 + k]])
    picked = s.timesteps[[0, 5, 2]]                          # advanced indexing copies: another storage
    for k, i in enumerate((0, 5, 2)):
        assert torch.equal(s.step(eps, picked[k], lat).prev_sample, want[host[i]])
    assert torch.equal(s.step(eps, torch.tensor(host[7], device=gpu), lat).prev_sample, want[host[7]])
    assert torch.equal(s.step_cfg(torch.cat([eps, eps]), s.timesteps[9], lat, 7.5),
                       s.step_cfg(torch.cat([eps, eps]), host[9], lat, 7.5))
    # a warm-up call with the first timestep does not shift the loop that follows (there is no cursor)
    s.step(eps, s.timesteps[0], lat)
    for i, t in enumerate(s.timesteps):
        if i in (0, 1, 48, 49):
            assert torch.equal(s.step(eps, t, lat).prev_sample, want[host[i]])


def test_blend_bit_exact(gpu):
    ops, _ = _ops()
    from oracle.pipeline_ref import plan_chunks, ramp_blend
    T, C, H, W = 24, 4, 6, 6
    cs, ov, ranges = plan_chunks(T, 2, 0, 4)
    g = torch.Generator().manual_seed(9)
    chunks = [(s, e, torch.randn(1, C, e - s, H, W, generator=g).half()) for s, e in ranges]
    like = torch.zeros(1, C, T, H, W, dtype=torch.float16)
    want = ramp_blend(chunks, T, ov, like)
    full = torch.zeros(1, C, T, H, W, dtype=torch.float16, device=gpu)
    weight = torch.zeros(T, dtype=torch.float32, device=gpu)
    ramp = torch.linspace(0, 1, ov)
    for s, e, lat in chunks:
        w = torch.ones(e - s)
        k = min(ov, e - s)
        w[:k] = ramp[:k]
        w[-k:] = torch.flip(ramp[:k], [0])
        ops.blend_accumulate(full, weight, lat.to(gpu), w.to(gpu), s, e)
    got = ops.blend_finalize(full, weight)
    assert torch.equal(got.cpu(), want)


# ---------------------------------------------------------------------------------------------
def _ff_ref(t, gamma, beta, w1, b1, w2, b2):
    """fp32 statement of `t + ff(norm3(t))` of BasicTransformerBlock (GEGLU, erf GELU)."""
    inner = t.shape[1]
    ln = F.layer_norm(t, (inner,), gamma, beta, 1e-5)
    pr = ln @ w1.t() + b1
    val, gate = pr.chunk(2, dim=1)
    return t + (val * F.gelu(gate)) @ w2.t() + b2


@pytest.mark.parametrize("M", [192, 1000, 64 * 1024 + 5, 7])
def test_ff_block_fused(gpu, M):
    """K8 against the fp32 statement and against the un-fused chain it replaces (LayerNorm, GEGLU GEMM, GEMM + residual)."""
    ops, _ = _ops()
    from vdx import packing
    g = torch.Generator().manual_seed(M)
    inner = 320
    t = h(torch.randn(M, inner, generator=g) * 1.5 + 0.3)
    gamma, beta = h(1 + 0.2 * torch.randn(inner, generator=g)), h(0.1 * torch.randn(inner, generator=g))
    w1, b1 = h(torch.randn(8 * inner, inner, generator=g) * 0.06), h(torch.randn(8 * inner, generator=g) * 0.1)
    w2, b2 = h(torch.randn(inner, 4 * inner, generator=g) * 0.03), h(torch.randn(inner, generator=g) * 0.1)
    ref = _ff_ref(t, gamma, beta, w1, b1, w2, b2)
    blob = packing.pack_k8(w1.to(gpu), b1.to(gpu), w2.to(gpu), b2.to(gpu), gamma.to(gpu), beta.to(gpu))
    td = t.half().to(gpu)
    out = ops.ff_block(td, blob, M=M)
    close(out, ref, tol=4e-3)
    # the un-fused chain on the same inputs
    wp, bp = packing.pack_geglu(w1.half().to(gpu), b1.half().to(gpu))
    ln = ops.layernorm(td, gamma.half().to(gpu), beta.half().to(gpu), M=M)
    gg = ops.gemm(ln, wp, M=M, bias=bp, geglu=True)
    un = ops.gemm(gg, packing.pack_conv1x1(w2.half().to(gpu)), M=M, bias=b2.half().to(gpu), residual=td)
    close(out, un.float().cpu(), tol=4e-3)


@pytest.mark.parametrize("M,half_x", [(192, False), (1000, False), (2 * 9216, True), (300 * 192 + 40, False), (2 * 24 * 9216, True), (24 * 9216, False)])
def test_ff_block_with_proj_out_fused(gpu, M, half_x):
    """K8 with the transformer's proj_out + residual behind it (PO, round 5): out = x + W_p (t + ff(norm3(t))) + b_p in one
    kernel, against the fp32 statement and against the kernels it replaces (K8, then the proj_out GEMM with its residual).
    One tile, ragged tiles, many tiles per workgroup (the next tile's rows are fetched and normalised behind the tail: the
    bits must repeat run to run), and `half_x`: both halves of the rows pair with the same rows of x (the shared prefix)."""
    ops, _ = _ops()
    from vdx import packing
    g = torch.Generator(device=gpu).manual_seed(M)
    inner = 320
    r = lambda *sh, k=1.0: (torch.randn(*sh, device=gpu, generator=g) * k).half()      # noqa: E731
    t = r(M, inner, k=1.5) + 0.3
    xrows = M // 2 if half_x else M
    x = r(xrows, inner)
    gamma, beta = r(inner, k=0.2) + 1, r(inner, k=0.1)
    w1, b1, w2, b2 = r(8 * inner, inner, k=0.06), r(8 * inner, k=0.1), r(inner, 4 * inner, k=0.03), r(inner, k=0.1)
    wp, bp = r(inner, inner, k=0.05), r(inner, k=0.1)
    blob = packing.pack_k8(w1, b1, w2, b2, gamma, beta)
    tail = packing.pack_k8_proj(wp, bp)
    out = ops.ff_block(t, blob, M=M, proj=(tail, x, xrows))
    # the kernels it replaces
    y = ops.ff_block(t, blob, M=M)
    xx = torch.cat([x, x]) if half_x else x
    un = ops.gemm(y, wp, M=M, bias=bp, residual=xx)
    close(out, un.float().cpu(), tol=4e-3)
    if M <= 20000:
        f = lambda v: v.float().cpu()       # noqa: E731
        ref = _ff_ref(f(t), f(gamma), f(beta), f(w1), f(b1), f(w2), f(b2)) @ f(wp).t() + f(bp) + f(xx)
        close(out, ref, tol=5e-3)
    for _ in range(3):
        assert torch.equal(ops.ff_block(t, blob, M=M, proj=(tail, x, xrows)), out)
    # rows are independent: a row range alone gives the same bits (tiles are 192 rows: a range that starts on a tile boundary)
    if M > 384 and not half_x:
        part = ops.ff_block(t[192:], blob, M=M - 192, proj=(tail, x[192:], M - 192))
        assert torch.equal(part, out[192:])
    # the plain kernel is unchanged by the shared template
    assert torch.equal(ops.ff_block(t, blob, M=M), y)


def test_ff_block_large_mean_and_outliers(gpu):
    """Rows with a large common offset (LayerNorm statistics) and gate values far outside the polynomial's interval."""
    ops, _ = _ops()
    from vdx import packing
    g = torch.Generator().manual_seed(5)
    inner, M = 320, 777
    t = h(torch.randn(M, inner, generator=g) * 0.5 + 40.0)
    t[::7] = h(torch.randn((M + 6) // 7, inner, generator=g) * 30.0)
    gamma, beta = h(1 + 0.2 * torch.randn(inner, generator=g)), h(0.1 * torch.randn(inner, generator=g))
    w1, b1 = h(torch.randn(8 * inner, inner, generator=g) * 0.3), h(torch.randn(8 * inner, generator=g) * 2.0)
    w2, b2 = h(torch.randn(inner, 4 * inner, generator=g) * 0.01), h(torch.randn(inner, generator=g) * 0.1)
    ref = _ff_ref(t, gamma, beta, w1, b1, w2, b2)
    blob = packing.pack_k8(w1.to(gpu), b1.to(gpu), w2.to(gpu), b2.to(gpu), gamma.to(gpu), beta.to(gpu))
    out = ops.ff_block(t.half().to(gpu), blob, M=M)
    assert bool(torch.isfinite(out.float()).all())
    close(out, ref, tol=6e-3)


def test_ff_block_run_to_run_bits_at_full_size(gpu):
    """Nine rounds of tiles per workgroup, every wave's counted waits under load: the bits must repeat (a wait that counts
    more vector-memory instructions than the ISA holds lets a weight unit be read before it has landed — seen once as a
    run-to-run difference of the whole forward, never at small sizes)."""
    ops, _ = _ops()
    from vdx import packing
    g = torch.Generator(device=gpu).manual_seed(3)
    inner, M = 320, 2 * 24 * 72 * 128
    r = lambda *sh, k=1.0: (torch.randn(*sh, device=gpu, generator=g) * k).half()      # noqa: E731
    blob = packing.pack_k8(r(8 * inner, inner, k=0.06), r(8 * inner, k=0.1), r(inner, 4 * inner, k=0.03), r(inner, k=0.1),
                           r(inner, k=0.2) + 1, r(inner, k=0.1))
    t = r(M, inner)
    first = ops.ff_block(t, blob, M=M).clone()
    for _ in range(6):
        assert torch.equal(ops.ff_block(t, blob, M=M), first)


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C,N,n_samples,rps", [(320, 320, 3, 128), (320, 512, 2, 192), (640, 640, 5, 64), (320, 320, 48, 576), (640, 640, 2, 4608)])
def test_groupnorm_folded_into_linear(gpu, C, N, n_samples, rps):
    """`norm` -> `proj_in` of the transformer blocks as per-sample weights + bias on the raw rows
    (vdx_groupnorm_fold_linear_f16 + vdx_gemm_args.wset_rows) against the fp32 statement and the un-fused pair."""
    ops, _ = _ops()
    g = torch.Generator().manual_seed(C + N + n_samples)
    M = n_samples * rps
    x = h(torch.randn(n_samples, rps, C, generator=g) * (0.5 + torch.rand(n_samples, 1, 1, generator=g) * 2) + torch.randn(n_samples, 1, C, generator=g))
    gamma, beta = h(1 + 0.2 * torch.randn(C, generator=g)), h(0.2 * torch.randn(C, generator=g))
    w, b = h(torch.randn(N, C, generator=g) * 0.05), h(torch.randn(N, generator=g) * 0.1)
    ref = F.linear(F.group_norm(x.permute(0, 2, 1), 32, gamma, beta, 1e-6).permute(0, 2, 1), w, b).reshape(M, N)
    xd = x.reshape(M, C).half().to(gpu)
    out = ops.groupnorm_linear(xd, gamma.half().to(gpu), beta.half().to(gpu), w.half().to(gpu), b.half().to(gpu), groups=32,
                               n_samples=n_samples, rows_per_sample=rps, eps=1e-6)
    close(out, ref, tol=4e-3)
    n = ops.groupnorm(xd, gamma.half().to(gpu), beta.half().to(gpu), groups=32, n_samples=n_samples, rows_per_sample=rps, eps=1e-6,
                      silu_act=False)
    un = ops.gemm(n, w.half().to(gpu), M=M, bias=b.half().to(gpu))
    close(out, un.float().cpu(), tol=4e-3)


def test_groupnorm_folded_into_linear_large_mean(gpu):
    """Group means 100 standard deviations out: the fold keeps the mean term in fp32 and forms it with the rounded weights
    the GEMM uses, so the rounding of the per-sample weights multiplies deviations only."""
    ops, _ = _ops()
    g = torch.Generator().manual_seed(9)
    C, N, n_samples, rps = 320, 320, 4, 256
    M = n_samples * rps
    x = h(torch.randn(n_samples, rps, C, generator=g) * 0.5 + 50.0 * torch.randn(n_samples, 1, C, generator=g).sign())
    gamma, beta = h(1 + 0.2 * torch.randn(C, generator=g)), h(0.2 * torch.randn(C, generator=g))
    w, b = h(torch.randn(N, C, generator=g) * 0.05), h(torch.randn(N, generator=g) * 0.1)
    ref = F.linear(F.group_norm(x.permute(0, 2, 1), 32, gamma, beta, 1e-6).permute(0, 2, 1), w, b).reshape(M, N)
    out = ops.groupnorm_linear(x.reshape(M, C).half().to(gpu), gamma.half().to(gpu), beta.half().to(gpu), w.half().to(gpu),
                               b.half().to(gpu), groups=32, n_samples=n_samples, rows_per_sample=rps, eps=1e-6)
    close(out, ref, tol=6e-3)


def _wide_affine(C, g):
    """gamma spanning 1e-3 .. 30 (log-uniform, both signs), beta up to +-5: the large-gamma analogue of the *_large_mean
    cases (real norm3 / norm1 channels have |gamma| far above 1 or near 0; folding gamma into fp16 weights
    rounds W * gamma once where the reference rounds the normalised activation first)."""
    mag = torch.exp(torch.empty(C).uniform_(math.log(1e-3), math.log(30.0), generator=g))
    sign = torch.where(torch.rand(C, generator=g) < 0.25, -1.0, 1.0)
    return h(mag * sign), h(torch.empty(C).uniform_(-5.0, 5.0, generator=g))


def test_fused_kernels_with_wide_gamma_and_beta(gpu):
    """K7 (second design), K8 and the GroupNorm fold with the norm's affine folded into fp16 weights, gamma in 1e-3 .. 30 and
    beta in +-5, against the fp32 statements.  The tolerance is relative to the output's scale (parity with a TRAINED
    checkpoint stays unpinned: the reference holds no fixture)."""
    ops, _ = _ops()
    from vdx import packing
    g = torch.Generator().manual_seed(123)
    inner, heads, B, Fr, HW = 320, 5, 1, 24, 40
    M = B * Fr * HW
    gamma, beta = _wide_affine(inner, g)
    t = h(torch.randn(M, inner, generator=g) * 1.5 + 0.3)
    # K7b: the weights are scaled so that the scores stay in a trained network's range (|gamma| up to 30 enters q AND k)
    wq, wk, wv, wo = (h(torch.randn(inner, inner, generator=g) * s_) for s_ in (0.006, 0.006, 0.01, 0.05))
    bo = h(0.1 * torch.randn(inner, generator=g))
    ref = _temporal_block_ref(t, gamma, beta, wq, wk, wv, wo, bo, B, Fr, HW, heads)
    blob = packing.pack_k7b(wq, wk, wv, wo, gamma, beta, bo, 0.125).to(gpu)
    out = ops.temporal_attn_block2(t.half().to(gpu), blob, B=B, F=Fr, HW=HW)
    close(out, ref, tol=6e-3)
    # K8
    w1, b1 = h(torch.randn(8 * inner, inner, generator=g) * 0.01), h(torch.randn(8 * inner, generator=g) * 0.1)
    w2, b2 = h(torch.randn(inner, 4 * inner, generator=g) * 0.03), h(torch.randn(inner, generator=g) * 0.1)
    ref = _ff_ref(t, gamma, beta, w1, b1, w2, b2)
    blob = packing.pack_k8(w1.to(gpu), b1.to(gpu), w2.to(gpu), b2.to(gpu), gamma.to(gpu), beta.to(gpu))
    out = ops.ff_block(t.half().to(gpu), blob, M=M)
    close(out, ref, tol=6e-3)
    # GroupNorm -> proj_in fold
    C, N, n_samples, rps = 320, 320, 3, 320
    x = h(torch.randn(n_samples, rps, C, generator=g) * 1.3 + torch.randn(n_samples, 1, C, generator=g))
    w, b = h(torch.randn(N, C, generator=g) * 0.02), h(torch.randn(N, generator=g) * 0.1)
    ref = F.linear(F.group_norm(x.permute(0, 2, 1), 32, gamma, beta, 1e-6).permute(0, 2, 1), w, b).reshape(n_samples * rps, N)
    out = ops.groupnorm_linear(x.reshape(-1, C).half().to(gpu), gamma.half().to(gpu), beta.half().to(gpu), w.half().to(gpu),
                               b.half().to(gpu), groups=32, n_samples=n_samples, rows_per_sample=rps, eps=1e-6)
    close(out, ref, tol=6e-3)
    # K1 / K3 apply the affine in fp32 (no fold): the same affine through them
    xi = h(torch.randn(2, C, 6, 32, generator=g) * 1.5)
    wc = h(torch.randn(320, C, 3, 3, generator=g) / math.sqrt(9 * C) / 8)
    ref = packing.nchw_to_rows(F.conv2d(h(F.silu(h(F.group_norm(xi, 32, gamma, beta, 1e-5)))), wc, padding=1))
    out = ops.conv3x3_gn(packing.nchw_to_rows(xi).half().to(gpu), gamma.half().to(gpu), beta.half().to(gpu),
                         packing.pack_conv3x3(wc.half()).to(gpu), groups=32, n_img=2, h=6, wd=32, eps=1e-5)
    close(out, ref, tol=6e-3)
    S, Co = 32, 320
    x5 = h(torch.randn(1, C, 16, S, 1, generator=g) * 1.5)
    wt = h(torch.randn(Co, C, 3, 1, 1, generator=g) / math.sqrt(3 * C) / 8)
    ref = _tconv_gn_ref(x5, gamma, beta, wt, None, None)
    rows = x5[..., 0].permute(0, 2, 3, 1).reshape(16 * S, C).contiguous().half().to(gpu)
    out = ops.tconv_gn(rows, gamma.half().to(gpu), beta.half().to(gpu), packing.pack_tconv3(wt).half().to(gpu), groups=32, B=1,
                       F=16, S=S, eps=1e-5)
    close(out, ref, tol=6e-3)


def test_ff_block_strided_rows(gpu):
    """t and out as column blocks of wider matrices (leading dimensions larger than the width): same bits as contiguous rows."""
    ops, _ = _ops()
    from vdx import packing
    g = torch.Generator(device=gpu).manual_seed(8)
    inner, M = 320, 1111
    r = lambda *sh, k=1.0: (torch.randn(*sh, device=gpu, generator=g) * k).half()      # noqa: E731
    blob = packing.pack_k8(r(8 * inner, inner, k=0.06), r(8 * inner, k=0.1), r(inner, 4 * inner, k=0.03), r(inner, k=0.1),
                           r(inner, k=0.2) + 1, r(inner, k=0.1))
    wide = r(M, 3 * inner)
    t = wide[:, inner:2 * inner]
    ref = ops.ff_block(t.contiguous(), blob, M=M)
    dst = torch.zeros(M, 2 * inner + 64, dtype=torch.float16, device=gpu)
    out = ops.ff_block(t, blob, M=M, out=dst[:, 64:64 + inner])
    assert torch.equal(out, ref) and float(dst[:, :64].abs().max()) == 0.0 and float(dst[:, 64 + inner:].abs().max()) == 0.0
