"""Host-side op wrappers: torch device tensors -> pointers/sizes -> libvdx_hip.so.

Every wrapper validates shapes/strides/alignment on the host before a kernel is enqueued
(a bad shape must raise here, never fault on the device).  Kernels run on torch's current
HIP stream.  All tensors are fp16 CUDA(=HIP) tensors whose last dimension is contiguous;
activations are channels-last row matrices [pixels][channels].
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import GemmArgs, VdxError

PLAIN, CONV3X3, TCONV3 = 0, 1, 2
EPI_GEGLU = 1


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor], name: str, dtype=torch.float16) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise VdxError(f"{name}: expected a GPU tensor (the denoising path has no CPU fallback)")
    if t.dtype != dtype:
        raise VdxError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.dim() >= 1 and t.stride(-1) != 1:
        raise VdxError(f"{name}: last dimension must be contiguous")
    if t.data_ptr() % 16 != 0:
        raise VdxError(f"{name}: pointer not 16-byte aligned")
    return t.data_ptr()


def _rows(t: torch.Tensor, name: str):
    if t.dim() != 2:
        raise VdxError(f"{name}: expected a 2-D row matrix, got shape {tuple(t.shape)}")
    return t.shape[0], t.shape[1], t.stride(0)


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# --------------------------------------------------------------------------------------------
_KSPLIT_WS = {}     # (device, stream) -> fp32 workspace of the split-K tails (grown on demand; the slice kernel and its
                    # reduction are ordered by ONE stream, so two streams must not share slabs)


def _ksplit_workspace(device, nbytes):
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _KSPLIT_WS.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = _KSPLIT_WS[key] = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
    return ws


def groupnorm_linear_supported(C: int, N: int, rows_per_sample: int) -> bool:
    """Shapes `groupnorm_linear` folds (the weights-stationary GEMM families with a weight set per sample)."""
    return C in (320, 640) and N % 32 == 0 and rows_per_sample % 64 == 0


def groupnorm_linear(x, gamma, beta, w, bias, *, groups, n_samples, rows_per_sample, eps, partition_samples=0, out=None):
    """Linear(GroupNorm(x)) without the normalised tensor (`norm` -> `proj_in` of the transformer blocks): the statistics
    pass, then one weight matrix + fp32 bias per sample (`vdx_groupnorm_fold_linear_f16`), then the weights-stationary GEMM
    on the raw rows with `wset_rows = rows_per_sample`."""
    lib = _lib.load()
    r, Cc, ldx = _rows(x, "x")
    M = n_samples * rows_per_sample
    N, K = w.shape
    if r < M or K != Cc or gamma.numel() != Cc or beta.numel() != Cc:
        raise VdxError(f"groupnorm_linear: x [{r}][{Cc}], w [{N}][{K}], gamma {gamma.numel()}: shapes do not match")
    if not groupnorm_linear_supported(Cc, N, rows_per_sample) or not w.is_contiguous():
        raise VdxError(f"groupnorm_linear: C={Cc}, N={N}, rows_per_sample={rows_per_sample} not supported")
    need = lib.vdx_groupnorm_workspace_part(n_samples, rows_per_sample, Cc, groups, partition_samples)
    key = (x.device.index, torch.cuda.current_stream(x.device).cuda_stream)
    ws = _gn_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=x.device)
        _gn_ws[key] = ws
    w_s = torch.empty((n_samples * N, K), dtype=torch.float16, device=x.device)
    b_s = torch.empty((n_samples, N), dtype=torch.float32, device=x.device)
    _lib.check(lib.vdx_groupnorm_fold_linear_f16(_p(x, "x"), Cc, ldx, _p(gamma, "gamma"), _p(beta, "beta"), float(eps), groups,
                                                 n_samples, rows_per_sample, ws.data_ptr(), partition_samples, _p(w, "w"),
                                                 _p(bias, "bias"), N, w_s.data_ptr(), b_s.data_ptr(), _stream()),
               "vdx_groupnorm_fold_linear_f16")
    return gemm(x, w_s, M=M, wset_rows=rows_per_sample, wset_bias=b_s, out=out)


def gemm(a, w, *, M, mode=PLAIN, a2=None, bias=None, bias2=None, rows_per_bias2=0, residual=None,
         out=None, geglu=False, conv=None, tconv=None, variant=0, row_begin=0, row_end=0, allow_ksplit=False, ksplit=0,
         wset_rows=0, wset_bias=None):
    """out[M][N] = epi(gather(a|a2)[M][K] @ w[N][K]^T).  See include/vdx.h `vdx_gemm_args`.
    `allow_ksplit`: the tail of the product (less than half a round of big tiles) may run as K slices + a fixed-order
    reduction (vdx_gemm_plan_ksplit) — faster on the 16-frame windows, not bit-identical to the unsplit order (the
    callers that rely on row-split bit-identity do not pass it).  `ksplit`: pin it for rows [row_begin, row_end).
    `wset_rows` / `wset_bias`: one weight set per `wset_rows` rows, w = [M / wset_rows][N][K] and an fp32 bias per set
    (a GroupNorm folded into the Linear: `groupnorm_linear`)."""
    lib = _lib.load()
    ar, c1, lda = _rows(a, "a")
    N, K = w.shape
    if wset_rows:
        if M % wset_rows or N % (M // wset_rows) or wset_bias is None or wset_bias.dtype != torch.float32:
            raise VdxError("gemm: wset_rows needs M % wset_rows == 0, w = [sets * N][K] and an fp32 wset_bias [sets][N]")
        N //= M // wset_rows
        if wset_bias.numel() != (M // wset_rows) * N or not wset_bias.is_contiguous():
            raise VdxError("gemm: wset_bias must be contiguous [sets][N]")
    if not w.is_contiguous():
        raise VdxError("w: must be contiguous [N][K]")
    c2 = 0
    g = GemmArgs()
    if a2 is not None:
        a2r, c2, lda2 = _rows(a2, "a2")
        g.lda2 = lda2
    taps = {PLAIN: 1, CONV3X3: 9, TCONV3: 3}[mode]
    if K != taps * (c1 + c2):
        raise VdxError(f"gemm: K={K} != {taps}*(c1={c1}+c2={c2})")
    n_out = N // 2 if geglu else N
    # rows each operand must provide
    if mode == PLAIN:
        need_rows = M
    elif mode == CONV3X3:
        n_img, h_in, w_in, h_out, w_out, stride, ups = conv
        if M != n_img * h_out * w_out:
            raise VdxError(f"gemm: M={M} != n_img*h_out*w_out={n_img * h_out * w_out}")
        need_rows = n_img * h_in * w_in
        g.h_in, g.w_in, g.h_out, g.w_out, g.stride, g.upsample = h_in, w_in, h_out, w_out, stride, int(ups)
    else:
        frames, hw = tconv
        if M % (frames * hw) != 0:
            raise VdxError(f"gemm: M={M} is not a whole number of (frames={frames} x hw={hw}) clips")
        need_rows = M
        g.frames, g.hw = frames, hw
    if ar < need_rows or (a2 is not None and a2r < need_rows):
        raise VdxError(f"gemm: source has {ar} rows, kernel would read {need_rows}")
    if out is None:
        out = torch.empty((M, n_out), dtype=torch.float16, device=a.device)
    orow, ocol, ldo = _rows(out, "out")
    if orow < M or ocol < n_out:
        raise VdxError(f"gemm: out {tuple(out.shape)} smaller than [{M}][{n_out}]")
    if bias is not None and bias.numel() != N:
        raise VdxError(f"gemm: bias has {bias.numel()} elements, N={N}")
    if bias2 is not None:
        b2r, b2c, ldb2 = _rows(bias2, "bias2")
        if rows_per_bias2 <= 0 or b2r * rows_per_bias2 < M or b2c < N:
            raise VdxError("gemm: bias2 does not cover M rows / N columns")
        g.rows_per_bias2, g.ldb2 = rows_per_bias2, ldb2
    if residual is not None:
        rr, rc, ldr = _rows(residual, "residual")
        if rr < M or rc < N:
            raise VdxError(f"gemm: residual {tuple(residual.shape)} smaller than [{M}][{N}]")
        g.ldr = ldr
    g.a, g.a2, g.w = _p(a, "a"), _p(a2, "a2"), _p(w, "w")
    g.bias, g.bias2, g.residual, g.out = _p(bias, "bias"), _p(bias2, "bias2"), _p(residual, "residual"), _p(out, "out")
    g.M, g.N, g.K, g.mode, g.c1, g.c2 = M, N, K, mode, c1, c2
    g.lda, g.ldo = lda, ldo
    g.epilogue = (EPI_GEGLU if geglu else 0) | ((variant & 15) << 8)   # variant: kernel override (tests/tuning)
    if wset_rows:
        g.wset_rows, g.wset_bias = wset_rows, wset_bias.data_ptr()
    # One product, up to two launches: whole rounds of 256 big tiles, then the rest on whatever tile suits it
    # (vdx_gemm_plan; the bits do not depend on the split).  A pinned variant or an explicit row range is left alone.
    spans = [(row_begin, row_end, ksplit)]
    if variant == 0 and row_begin == 0 and row_end == 0 and ksplit == 0 and not wset_rows:
        key = (M, N, K, mode, geglu, a2 is not None, bias2 is not None, allow_ksplit, int(g.upsample))      # everything the plan depends on
        plan_ = _PLAN_CACHE.get(key)
        if plan_ is None:
            v_, split_, ks_, wsb_ = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_size_t(0)
            if allow_ksplit:
                _lib.check(lib.vdx_gemm_plan_ksplit(C.byref(g), C.byref(split_), C.byref(ks_), C.byref(wsb_)), "vdx_gemm_plan_ksplit")
            if ks_.value == 0:
                _lib.check(lib.vdx_gemm_plan(C.byref(g), C.byref(v_), C.byref(split_)), "vdx_gemm_plan")
            plan_ = _PLAN_CACHE[key] = (split_.value, ks_.value, wsb_.value)
        split, ks, wsb = plan_
        if split:
            spans = [(0, split, 0), (split, 0, ks)]
    elif ksplit > 1:
        nt = -(-((row_end or M) - row_begin) // 256) * -(-N // 320)
        wsb = nt * ksplit * 327680
    for rb, re_, ks in spans:
        g.row_begin, g.row_end, g.ksplit, g.workspace, g.workspace_bytes = rb, re_, ks, None, 0
        if ks > 1:
            ws_ = _ksplit_workspace(out.device, wsb)
            g.workspace, g.workspace_bytes = ws_.data_ptr(), ws_.numel() * 4
        if PROFILE is not None:
            rows = (re_ or M) - rb
            name = gemm_kernel_name(rows, N, K, mode, geglu, 7 if wset_rows else variant, single_source=a2 is None and bias2 is None,
                                    residual=residual is not None, whole=(rb == 0 and re_ in (0, M)), wset=bool(wset_rows))
            if ks > 1:
                name = f"gemm_kernel<256, 320, 4, 2, {mode}, false, {'true' if mode else 'false'}, 1> split-K + reduce"
        if PROFILE is None or not _profiled(name):
            _lib.check(lib.vdx_gemm_f16(C.byref(g), _stream()), "vdx_gemm_f16")
            continue
        # bench.py instrumentation: HIP events on the launch stream around this one kernel
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        _lib.check(lib.vdx_gemm_f16(C.byref(g), _stream()), "vdx_gemm_f16")
        ev1.record()
        PROFILE.append((name, 2.0 * rows * N * K, ev0, ev1, (rows, N, K)))
    return out


_PLAN_CACHE = {}
PROFILE = None   # set to a list by bench.py to collect (kernel name, algorithmic FLOPs, start, end)
PROFILE_ONLY = None   # with PROFILE set: only launches whose kernel name contains one of these strings get events (None: all)


def _profiled(name: str) -> bool:
    return PROFILE_ONLY is None or any(s_ in name for s_ in PROFILE_ONLY)


WS_MIN_ROWS = 16384   # gemm.hip: smallest M the weights-stationary K=320 kernel is picked for


def gemm_kernel_name(M: int, N: int, K: int, mode: int, geglu: bool, variant: int = 0, single_source: bool = True,
                     residual: bool = False, whole: bool = True, wset: bool = False) -> str:
    """Name of the instantiation vdx_gemm_f16 launches (as rocprofv3 prints it) for M rows (`whole`: the call covers
    the whole product — the weights-stationary kernels take no row ranges)."""
    v = variant
    if whole and single_source and mode == PLAIN and N % 32 == 0 and (v == 7 or (v == 0 and M >= WS_MIN_ROWS)):
        fam = None   # mirrors vdx_gemm_ws_family (gemm_ws.hip): (K, waves, chunk rows, pipelined)
        if K == 320 and M % 64 == 0:
            fam = (320, 10, 64, False) if (N % 320 == 0 and not (geglu and N % 256 == 0)) else (320, 8, 64, True)
        elif ((K == 512 and N % 256 == 0) or K == 640) and M % 32 == 0:
            fam = (K, 8, 32, True)
        if fam:
            b = lambda x: "true" if x else "false"
            return f"gemm_ws_kernel<{fam[0]}, {fam[1]}, {fam[2]}, {b(geglu)}, {b(residual and not geglu)}, {b(fam[3])}, {b(wset)}>"
    if v == 0:
        nt320 = (N + 319) // 320
        fits = nt320 * 320 * 4 <= N * 5 and M >= 1024
        if not fits:
            v = 1 if N > 64 else 5
        else:   # mirrors pick_tile (gemm.hip): rounds of 256 tiles x relative tile time
            t256 = ((M + 255) // 256) * nt320
            t128 = ((M + 127) // 128) * nt320
            t1 = ((M + 127) // 128) * ((N + 127) // 128)
            c2, c8, c1 = 10 * ((t256 + 255) // 256), 8 * ((t128 + 255) // 256), 3 * ((t1 + 255) // 256)
            v = 2 if (c2 <= c8 and c2 <= c1) else (8 if c8 <= c1 else 1)
    tail = f"{0 if geglu else mode}, {'true' if geglu else 'false'}"
    split = {1: ", false, 0>", 9: ", false, 0>", 2: ", true, 0>" if (mode != PLAIN and not geglu) else ", false, 0>", 5: ", false, 0>",
             6: ", false, 0>"}.get(v, ">")   # gemm_kernel's SPLIT flag and VAR (0: the product kernels)
    return {1: "gemm_kernel<128, 128, 4, 2, ", 9: "gemm_kernel<128, 128, 2, 2, ", 2: "gemm_kernel<256, 320, 4, 2, ", 3: "gemm_ring_kernel<4, 64, 4, ",
            4: "gemm_ring_kernel<2, 64, 2, ", 8: "gemm_ring_kernel<4, 32, 4, ", 5: "gemm_kernel<256, 64, 4, 1, ", 6: "gemm_kernel<256, 320, 4, 2, "}[v] + tail + split


def conv_in(x, w, bias, out=None):
    """x (B,Cin,F,H,W) fp16 -> rows [B*F*H*W][Cout]; w [Cout][Kpad] = pack_conv3x3 zero-padded in K
    to a multiple of 64.  im2col gather (HBM-light: Cin = 4) + the MFMA GEMM."""
    lib = _lib.load()
    B, Cin, F, H, W = x.shape
    if not x.is_contiguous():
        raise VdxError("conv_in: x must be contiguous (B,C,F,H,W)")
    Cout, Kpad = w.shape
    if Kpad % 64 != 0 or Kpad < 9 * Cin or not w.is_contiguous():
        raise VdxError("conv_in: w must be contiguous [Cout][Kpad], Kpad a multiple of 64 >= 9*Cin")
    M = B * F * H * W
    cols = torch.empty((M, Kpad), dtype=torch.float16, device=x.device)
    _lib.check(lib.vdx_im2col_in_f16(_p(x, "x"), _p(cols, "cols"), B, Cin, F, H, W, Kpad, _stream()),
               "vdx_im2col_in_f16")
    return gemm(cols, w, M=M, bias=bias, out=out)


def rows_to_ncfhw(rows, B, C, F, H, W, out=None):
    lib = _lib.load()
    r, c, ld = _rows(rows, "rows")
    if r < B * F * H * W or c < C:
        raise VdxError("rows_to_ncfhw: rows too small")
    if out is None:
        out = torch.empty((B, C, F, H, W), dtype=torch.float16, device=rows.device)
    if tuple(out.shape) != (B, C, F, H, W) or not out.is_contiguous():
        raise VdxError("rows_to_ncfhw: bad out")
    _lib.check(lib.vdx_rows_to_ncfhw_f16(_p(rows, "rows"), ld, _p(out, "out"), B, C, F, H, W, _stream()),
               "vdx_rows_to_ncfhw_f16")
    return out


def silu(x, out=None):
    lib = _lib.load()
    if not x.is_contiguous():
        raise VdxError("silu: x must be contiguous")
    if out is None:
        out = torch.empty_like(x)
    _lib.check(lib.vdx_silu_f16(_p(x, "x"), _p(out, "out"), x.numel(), _stream()), "vdx_silu_f16")
    return out


def timestep_embedding(t_dev, B, dim, out=None):
    """Sinusoidal embedding of ONE fp32 timestep held in device memory -> fp16 [B][dim] (include/vdx.h)."""
    lib = _lib.load()
    if t_dev.dtype != torch.float32 or t_dev.numel() != 1:
        raise VdxError("timestep_embedding: t must be one fp32 value on the GPU")
    if out is None:
        out = torch.empty((B, dim), dtype=torch.float16, device=t_dev.device)
    _lib.check(lib.vdx_timestep_embedding_f16(_p(t_dev, "t", torch.float32), _p(out, "out"), B, dim, _stream()),
               "vdx_timestep_embedding_f16")
    return out


def gelu(x, out=None):
    lib = _lib.load()
    if not x.is_contiguous():
        raise VdxError("gelu: x must be contiguous")
    if out is None:
        out = torch.empty_like(x)
    _lib.check(lib.vdx_gelu_f16(_p(x, "x"), _p(out, "out"), x.numel(), _stream()), "vdx_gelu_f16")
    return out


# --------------------------------------------------------------------------------------------
_gn_ws: dict = {}


def groupnorm(x, gamma, beta, *, groups, n_samples, rows_per_sample, eps, silu_act, x2=None, out=None, partition_samples=0):
    """GroupNorm (+SiLU) over rows [n_samples*rows_per_sample][C]; x2 = second concat source.
    partition_samples: reduce the statistics with the slab partition of a batch of that many samples (bit-stable
    results for a sample whatever batch it is normalised in; include/vdx.h)."""
    lib = _lib.load()
    r, c1, ldx = _rows(x, "x")
    c2, ldx2 = 0, 0
    M = n_samples * rows_per_sample
    if r < M:
        raise VdxError(f"groupnorm: x has {r} rows, need {M}")
    if x2 is not None:
        r2, c2, ldx2 = _rows(x2, "x2")
        if r2 < M:
            raise VdxError(f"groupnorm: x2 has {r2} rows, need {M}")
    Cc = c1 + c2
    if gamma.numel() != Cc or beta.numel() != Cc:
        raise VdxError(f"groupnorm: gamma/beta size {gamma.numel()} != C={Cc}")
    if out is None:
        out = torch.empty((M, Cc), dtype=torch.float16, device=x.device)
    orow, ocol, ldy = _rows(out, "out")
    if orow < M or ocol < Cc:
        raise VdxError("groupnorm: out too small")
    need = lib.vdx_groupnorm_workspace_part(n_samples, rows_per_sample, Cc, groups, partition_samples)
    key = (x.device.index, torch.cuda.current_stream(x.device).cuda_stream)
    ws = _gn_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=x.device)
        _gn_ws[key] = ws
    _lib.check(lib.vdx_groupnorm_part_f16(_p(x, "x"), c1, ldx, _p(x2, "x2"), c2, ldx2, _p(gamma, "gamma"),
                                          _p(beta, "beta"), float(eps), groups, n_samples, rows_per_sample,
                                          int(bool(silu_act)), _p(out, "out"), ldy, ws.data_ptr(), partition_samples,
                                          _stream()), "vdx_groupnorm_f16")
    return out


def conv3x3_gn_preferred(c1: int, c2: int, N: int, n_img: int, h: int, w: int) -> bool:
    """K1 supported AND expected to be faster than the apply pass + conv GEMM (level 0 of the XL UNet)."""
    return bool(_lib.load().vdx_conv3x3_gn_preferred(c1, c2, N, n_img, h, w))


def conv3x3_gn_supported(c1: int, c2: int, N: int) -> bool:
    return bool(_lib.load().vdx_conv3x3_gn_supported(c1, c2, N))


def conv3x3_gn(x, gamma, beta, w, *, x2=None, bias=None, bias2=None, rows_per_bias2=0, residual=None, groups, n_img, h, wd, eps,
               partition_samples=0, out=None):
    """Conv2d 3x3 (pad 1, stride 1) of SiLU(GroupNorm4d(cat(x, x2))) — conv1 / conv2 of ResnetBlock2D — without the
    normalised tensor: the statistics pass (`vdx_groupnorm_stats_f16`, one sample per image) leaves a scale / shift pair per
    (image, channel); K1 (`vdx_conv3x3_gn_f16`, csrc/conv_fused.hip) applies them, and the SiLU, to its staged image patch
    in LDS.  x (and x2: the skip tensor of the up blocks): raw rows [n_img*h*wd][c]; w: packed weights [N][9*(c1+c2)]."""
    lib = _lib.load()
    r, c1, ldx = _rows(x, "x")
    S = h * wd
    M = n_img * S
    c2 = ldx2 = 0
    if x2 is not None:
        r2, c2, ldx2 = _rows(x2, "x2")
        if r2 < M:
            raise VdxError(f"conv3x3_gn: x2 has {r2} rows, need {M}")
    Cc = c1 + c2
    N, K = w.shape
    if r < M or K != 9 * Cc or gamma.numel() != Cc or beta.numel() != Cc or not w.is_contiguous():
        raise VdxError(f"conv3x3_gn: x [{r}][{c1}] (+{c2}), w [{N}][{K}], gamma {gamma.numel()}, M = {M}: shapes do not match")
    if not conv3x3_gn_supported(c1, c2, N):
        raise VdxError(f"conv3x3_gn: c1={c1}, c2={c2}, N={N} not supported")
    if bias is not None and bias.numel() != N:
        raise VdxError(f"conv3x3_gn: bias has {bias.numel()} elements, N={N}")
    ldb2 = 0
    if bias2 is not None:
        b2r, b2c, ldb2 = _rows(bias2, "bias2")
        if rows_per_bias2 <= 0 or b2r * rows_per_bias2 < M or b2c < N:
            raise VdxError("conv3x3_gn: bias2 does not cover M rows / N columns")
    ldr = 0
    if residual is not None:
        rr, rc, ldr = _rows(residual, "residual")
        if rr < M or rc < N:
            raise VdxError(f"conv3x3_gn: residual {tuple(residual.shape)} smaller than [{M}][{N}]")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float16, device=x.device)
    orow, ocol, ldo = _rows(out, "out")
    if orow < M or ocol < N:
        raise VdxError("conv3x3_gn: out too small")
    need = lib.vdx_groupnorm_workspace_part(n_img, S, Cc, groups, partition_samples)
    key = (x.device.index, torch.cuda.current_stream(x.device).cuda_stream)
    ws = _gn_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=x.device)
        _gn_ws[key] = ws
    off = C.c_size_t(0)
    _lib.check(lib.vdx_groupnorm_stats_f16(_p(x, "x"), c1, ldx, _p(x2, "x2"), c2, ldx2, _p(gamma, "gamma"), _p(beta, "beta"), float(eps),
                                           groups, n_img, S, ws.data_ptr(), partition_samples, C.byref(off), _stream()),
               "vdx_groupnorm_stats_f16")
    name = "conv3x3_gn_kernel"
    timed = PROFILE is not None and _profiled(name)       # bench.py --profile-all: HIP events around this launch
    if timed:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    _lib.check(lib.vdx_conv3x3_gn_f16(_p(x, "x"), ldx, _p(x2, "x2"), ldx2, c1, c2, ws.data_ptr() + off.value, _p(w, "w"), _p(bias, "bias"),
                                      _p(bias2, "bias2"), rows_per_bias2, ldb2, _p(residual, "residual"), ldr, _p(out, "out"), ldo,
                                      n_img, h, wd, N, _stream()), "vdx_conv3x3_gn_f16")
    if timed:
        ev1.record()
        PROFILE.append((name, 2.0 * M * N * K, ev0, ev1, (M, N, K)))
    return out


def tconv_gn_supported(C: int, N: int, F: int) -> bool:
    """Shapes `tconv_gn` (K3, csrc/tconv_fused.hip) takes: C % 64 == 0, N % 320 == 0, F % 8 == 0."""
    return bool(_lib.load().vdx_tconv_gn_supported(C, N, F))


def tconv_gn_preferred(C: int, N: int, B: int, F: int, S: int) -> bool:
    """K3 supported AND expected to be faster than the apply pass + TCONV3 GEMM (level 0 of the XL UNet)."""
    return bool(_lib.load().vdx_tconv_gn_preferred(C, N, B, F, S))


def tconv_gn(x, gamma, beta, w, *, bias=None, residual=None, groups, B, F, S, eps, partition_samples=0, out=None):
    """Conv3d (3,1,1) of SiLU(GroupNorm5d(x)) — one link of TemporalConvLayer's chain — without the normalised tensor:
    the statistics pass (`vdx_groupnorm_stats_f16`, n_samples = B, rows_per_sample = F*S) leaves a scale / shift pair per
    (sample, channel); K3 (`vdx_tconv_gn_f16`) applies them, and the SiLU, to its staged image in LDS.  x: raw rows
    [B*F*S][C]; w: packed temporal weights [N][3*C] (packing.pack_tconv)."""
    lib = _lib.load()
    r, Cc, ldx = _rows(x, "x")
    M = B * F * S
    N, K = w.shape
    if r < M or K != 3 * Cc or gamma.numel() != Cc or beta.numel() != Cc or not w.is_contiguous():
        raise VdxError(f"tconv_gn: x [{r}][{Cc}], w [{N}][{K}], gamma {gamma.numel()}, M = {M}: shapes do not match")
    if not tconv_gn_supported(Cc, N, F):
        raise VdxError(f"tconv_gn: C={Cc}, N={N}, F={F} not supported")
    if bias is not None and bias.numel() != N:
        raise VdxError(f"tconv_gn: bias has {bias.numel()} elements, N={N}")
    ldr = 0
    if residual is not None:
        rr, rc, ldr = _rows(residual, "residual")
        if rr < M or rc < N:
            raise VdxError(f"tconv_gn: residual {tuple(residual.shape)} smaller than [{M}][{N}]")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float16, device=x.device)
    orow, ocol, ldo = _rows(out, "out")
    if orow < M or ocol < N:
        raise VdxError("tconv_gn: out too small")
    need = lib.vdx_groupnorm_workspace_part(B, F * S, Cc, groups, partition_samples)
    key = (x.device.index, torch.cuda.current_stream(x.device).cuda_stream)
    ws = _gn_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=x.device)
        _gn_ws[key] = ws
    off = C.c_size_t(0)
    _lib.check(lib.vdx_groupnorm_stats_f16(_p(x, "x"), Cc, ldx, None, 0, 0, _p(gamma, "gamma"), _p(beta, "beta"), float(eps), groups, B,
                                           F * S, ws.data_ptr(), partition_samples, C.byref(off), _stream()), "vdx_groupnorm_stats_f16")
    name = f"tconv_gn_kernel<{16 if F % 16 == 0 else 12 if F % 12 == 0 else 8}>"
    timed = PROFILE is not None and _profiled(name)       # bench.py --profile-all: HIP events around this launch
    if timed:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    _lib.check(lib.vdx_tconv_gn_f16(_p(x, "x"), ldx, ws.data_ptr() + off.value, _p(w, "w"), _p(bias, "bias"), _p(residual, "residual"),
                                    ldr, _p(out, "out"), ldo, B, F, S, Cc, N, _stream()), "vdx_tconv_gn_f16")
    if timed:
        ev1.record()
        PROFILE.append((name, 2.0 * M * N * K, ev0, ev1, (M, N, K)))
    return out


def layernorm(x, gamma, beta, *, M, eps=1e-5, out=None):
    lib = _lib.load()
    r, Cc, ldx = _rows(x, "x")
    if r < M:
        raise VdxError("layernorm: x too small")
    if gamma.numel() != Cc or beta.numel() != Cc:
        raise VdxError("layernorm: gamma/beta size")
    if out is None:
        out = torch.empty((M, Cc), dtype=torch.float16, device=x.device)
    orow, ocol, ldy = _rows(out, "out")
    if orow < M or ocol < Cc:
        raise VdxError("layernorm: out too small")
    _lib.check(lib.vdx_layernorm_f16(_p(x, "x"), ldx, _p(gamma, "gamma"), _p(beta, "beta"), float(eps), M, Cc,
                                     _p(out, "out"), ldy, _stream()), "vdx_layernorm_f16")
    return out


def softmax_rows(x, *, rows, cols, scale):
    """In-place softmax(scale * x[r, :cols]) per row (fp32 inside): AutoencoderKL mid-block attention."""
    lib = _lib.load()
    r, c, ld = _rows(x, "x")
    if r < rows or c < cols:
        raise VdxError("softmax_rows: x too small")
    _lib.check(lib.vdx_softmax_rows_f16(_p(x, "x"), ld, rows, cols, float(scale), _stream()), "vdx_softmax_rows_f16")
    return x


def rows_to_u8_frames(rows, n, H, W):
    """Decoder output rows [n*H*W][ld] (RGB first) -> uint8 (n,H,W,3): fsdp_chunked_coherent.py:224-225."""
    lib = _lib.load()
    r, c, ld = _rows(rows, "rows")
    if r < n * H * W or c < 3:
        raise VdxError("rows_to_u8_frames: rows too small")
    out = torch.empty((n, H, W, 3), dtype=torch.uint8, device=rows.device)
    _lib.check(lib.vdx_rows_to_u8_frames(_p(rows, "rows"), ld, n * H * W, out.data_ptr(), _stream()),
               "vdx_rows_to_u8_frames")
    return out


# --------------------------------------------------------------------------------------------
def flash_attn(q, k, vt, *, n_seq, sq, skv, skv_pad, heads, seq_per_kv, scale, out=None, causal=False, v_rows=False):
    """q rows [n_seq*sq][>=heads*64]; k rows [n_kv*skv_pad][>=heads*64]; vt [heads*64][>= n_kv*skv_pad] — or, with
    `v_rows`, V as rows like k (the third column block of a q|k|v projection: `vdx_flash_attn_rows_f16`)."""
    lib = _lib.load()
    qr, qc, ldq = _rows(q, "q")
    kr, kc, ldk = _rows(k, "k")
    vr, vc, ldvt = _rows(vt, "v" if v_rows else "vt")
    inner = heads * 64
    n_kv = n_seq // seq_per_kv
    if qr < n_seq * sq or qc < inner:
        raise VdxError("flash_attn: q too small")
    if kr < n_kv * skv_pad or kc < inner:
        raise VdxError("flash_attn: k too small")
    if v_rows and (vr < n_kv * skv_pad or vc < inner):
        raise VdxError("flash_attn: v too small")
    if not v_rows and (vr < inner or vc < n_kv * skv_pad):
        raise VdxError("flash_attn: vt too small")
    if out is None:
        out = torch.empty((n_seq * sq, inner), dtype=torch.float16, device=q.device)
    orow, ocol, ldo = _rows(out, "out")
    if orow < n_seq * sq or ocol < inner:
        raise VdxError("flash_attn: out too small")
    rec = False
    if PROFILE is not None:        # bench.py instrumentation (as in gemm): HIP events on the launch stream
        two = sq >= 512 and skv >= 256                      # flash.hip: 64 queries per wave
        name = f"flash_attn_kernel<{2 if two else 1}, {'true' if causal else 'false'}, {'true' if v_rows else 'false'}>"
        rec = _profiled(name)
    if rec:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record()
    fn = lib.vdx_flash_attn_rows_f16 if v_rows else lib.vdx_flash_attn_f16
    _lib.check(fn(_p(q, "q"), ldq, _p(k, "k"), ldk, _p(vt, "vt"), ldvt, _p(out, "out"), ldo,
                  n_seq, sq, skv, skv_pad, heads, seq_per_kv, float(scale), int(bool(causal)), _stream()),
               "vdx_flash_attn_rows_f16" if v_rows else "vdx_flash_attn_f16")
    if rec:
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record()
        PROFILE.append((name, 4.0 * n_seq * heads * sq * skv * 64, ev0, ev1, (n_seq * sq, skv, heads * 64)))
    return out


def temporal_attn(qkv, *, B, F, HW, heads, scale, out=None):
    lib = _lib.load()
    r, c, ld = _rows(qkv, "qkv")
    inner = heads * 64
    M = B * F * HW
    if r < M or c < 3 * inner:
        raise VdxError("temporal_attn: qkv too small")
    if out is None:
        out = torch.empty((M, inner), dtype=torch.float16, device=qkv.device)
    orow, ocol, ldo = _rows(out, "out")
    if orow < M or ocol < inner:
        raise VdxError("temporal_attn: out too small")
    _lib.check(lib.vdx_temporal_attn_f16(_p(qkv, "qkv"), ld, _p(out, "out"), ldo, B, F, HW, heads, float(scale),
                                         _stream()), "vdx_temporal_attn_f16")
    return out


def temporal_attn_block2_supported(inner: int, F: int) -> bool:
    return bool(_lib.load().vdx_temporal_attn_block2_supported(inner, F))


def temporal_attn_block2(t, packed, *, B, F, HW, eps=1e-5, out=None):
    """K7, second design (csrc/tattn2.hip): t + to_out(attention_over_frames(LayerNorm(t))) in one kernel; LayerNorm's
    affine, the softmax scale and the biases are inside `packed` (packing.pack_k7b)."""
    lib = _lib.load()
    r, inner, ldt = _rows(t, "t")
    M = B * F * HW
    if r < M:
        raise VdxError(f"temporal_attn_block2: t has {r} rows, need {M}")
    if not lib.vdx_temporal_attn_block2_supported(inner, F):
        raise VdxError(f"temporal_attn_block2: inner={inner}, F={F} not supported by the fused kernel")
    if packed.dtype != torch.float16 or packed.numel() * 2 != lib.vdx_temporal_attn_block2_pack_bytes(inner) \
            or not packed.is_contiguous():
        raise VdxError("temporal_attn_block2: packed blob does not match the kernel's layout (packing.pack_k7b)")
    if out is None:
        out = torch.empty((M, inner), dtype=torch.float16, device=t.device)
    orow, ocol, ldo = _rows(out, "out")
    if orow < M or ocol < inner:
        raise VdxError("temporal_attn_block2: out too small")
    if out.data_ptr() == t.data_ptr():
        raise VdxError("temporal_attn_block2: out may not alias t")
    _lib.check(lib.vdx_temporal_attn_block2_f16(_p(t, "t"), ldt, _p(packed, "packed"), float(eps), _p(out, "out"), ldo,
                                                B, F, HW, inner, _stream()), "vdx_temporal_attn_block2_f16")
    return out


def ff_block_supported(inner: int) -> bool:
    return bool(_lib.load().vdx_ff_block_supported(inner))


def ff_block(t, packed, *, M, eps=1e-5, out=None, proj=None):
    """K8 (csrc/ff_fused.hip): t + ff(LayerNorm(t)) — GEGLU feed-forward of a transformer block — in one kernel;
    LayerNorm's affine and the biases are inside `packed` (packing.pack_k8).
    `proj` = (tail blob of packing.pack_k8_proj, x, xrows): the transformer's proj_out and its residual run behind the
    feed-forward in the same kernel — out[r] = x[r % xrows] + W_p . (t[r] + ff(LayerNorm(t[r]))) + b_p, xrows = M or M / 2."""
    lib = _lib.load()
    r, inner, ldt = _rows(t, "t")
    if r < M:
        raise VdxError(f"ff_block: t has {r} rows, need {M}")
    if not lib.vdx_ff_block_supported(inner):
        raise VdxError(f"ff_block: inner={inner} not supported by the fused kernel")
    if packed.dtype != torch.float16 or packed.numel() * 2 != lib.vdx_ff_block_pack_bytes(inner) or not packed.is_contiguous():
        raise VdxError("ff_block: packed blob does not match the kernel's layout (packing.pack_k8)")
    if out is None:
        out = torch.empty((M, inner), dtype=torch.float16, device=t.device)
    orow, ocol, ldo = _rows(out, "out")
    if orow < M or ocol < inner:
        raise VdxError("ff_block: out too small")
    if out.data_ptr() == t.data_ptr():
        raise VdxError("ff_block: out may not alias t")
    name = f"ff_fused_kernel<{inner}, {'true' if proj is not None else 'false'}>"
    rec = PROFILE is not None and _profiled(name)
    if rec:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record()
    if proj is None:
        _lib.check(lib.vdx_ff_block_f16(_p(t, "t"), ldt, _p(packed, "packed"), float(eps), _p(out, "out"), ldo, M, inner, _stream()),
                   "vdx_ff_block_f16")
    else:
        blob, x, xrows = proj
        xr, xc, ldx = _rows(x, "x")
        if xc < inner or xr < xrows or xrows not in (M, M // 2) or (xrows != M and 2 * xrows != M):
            raise VdxError(f"ff_block: x {tuple(x.shape)} / xrows {xrows} do not pair with M = {M} rows")
        if blob.dtype != torch.float16 or blob.numel() * 2 != lib.vdx_ff_block_proj_pack_bytes(inner) or not blob.is_contiguous():
            raise VdxError("ff_block: proj blob does not match the kernel's layout (packing.pack_k8_proj)")
        if out.data_ptr() == x.data_ptr():
            raise VdxError("ff_block: out may not alias x")
        _lib.check(lib.vdx_ff_block_proj_f16(_p(t, "t"), ldt, _p(packed, "packed"), float(eps), _p(x, "x"), ldx, int(xrows), _p(blob, "proj"),
                                             _p(out, "out"), ldo, M, inner, _stream()), "vdx_ff_block_proj_f16")
    if rec:
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record()
        PROFILE.append((name, 2.0 * M * inner * (12 + (1 if proj is not None else 0)) * inner, ev0, ev1, (M, inner, 12 * inner)))
    return out


def cross_attn_block_supported(inner: int, kv_len: int) -> bool:
    return bool(_lib.load().vdx_cross_attn_block_supported(inner, kv_len))


def cross_attn_block(t, packed, kv_packed, *, kv_len, n_items, rows_per_item, eps=1e-5, out=None):
    """K5 (csrc/xattn.hip): t + to_out(softmax(q K^T) V), q = LayerNorm(t).W_q^T — the cross-attention sub-block of a spatial
    transformer — in one kernel.  `packed`: packing.pack_k5 (LayerNorm's affine, the scale, the biases inside);
    `kv_packed`: packing.pack_k5_kv of the text keys / values, [n_items][heads][3 units]; rows [n_items*rows_per_item][inner]."""
    lib = _lib.load()
    r, inner, ldt = _rows(t, "t")
    M = n_items * rows_per_item
    if r < M:
        raise VdxError(f"cross_attn_block: t has {r} rows, need {M}")
    if not lib.vdx_cross_attn_block_supported(inner, kv_len):
        raise VdxError(f"cross_attn_block: inner={inner}, kv_len={kv_len} not supported by the fused kernel")
    if packed.dtype != torch.float16 or packed.numel() * 2 != lib.vdx_cross_attn_block_pack_bytes(inner) or not packed.is_contiguous():
        raise VdxError("cross_attn_block: packed blob does not match the kernel's layout (packing.pack_k5)")
    if kv_packed.dtype != torch.float16 or kv_packed.numel() * 2 != n_items * lib.vdx_cross_attn_block_kv_bytes(inner) \
            or not kv_packed.is_contiguous():
        raise VdxError("cross_attn_block: key / value blob does not match the kernel's layout (packing.pack_k5_kv)")
    if out is None:
        out = torch.empty((M, inner), dtype=torch.float16, device=t.device)
    orow, ocol, ldo = _rows(out, "out")
    if orow < M or ocol < inner:
        raise VdxError("cross_attn_block: out too small")
    if out.data_ptr() == t.data_ptr():
        raise VdxError("cross_attn_block: out may not alias t")
    rec = PROFILE is not None and _profiled(f"xattn_kernel<{inner}>")
    if rec:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record()
    _lib.check(lib.vdx_cross_attn_block_f16(_p(t, "t"), ldt, _p(packed, "packed"), _p(kv_packed, "kv_packed"), int(kv_len), float(eps),
                                            _p(out, "out"), ldo, n_items, rows_per_item, inner, _stream()), "vdx_cross_attn_block_f16")
    if rec:
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record()
        PROFILE.append((f"xattn_kernel<{inner}>", 2.0 * M * (2 * inner * inner + 2 * kv_len * inner), ev0, ev1, (M, inner, 2 * inner + 2 * kv_len)))
    return out


def temporal_attn_block_supported(inner: int, F: int) -> bool:
    return bool(_lib.load().vdx_temporal_attn_block_supported(inner, F))


def temporal_attn_block(t, gamma, beta, wqkv_packed, wo_packed, bo, *, B, F, HW, scale, eps=1e-5, out=None):
    """K7: t + to_out(attention_over_frames(LayerNorm(t))) in one kernel (include/vdx.h)."""
    lib = _lib.load()
    r, inner, ldt = _rows(t, "t")
    M = B * F * HW
    if r < M:
        raise VdxError(f"temporal_attn_block: t has {r} rows, need {M}")
    if not lib.vdx_temporal_attn_block_supported(inner, F):
        raise VdxError(f"temporal_attn_block: inner={inner}, F={F} not supported by the fused kernel")
    if gamma.numel() != inner or beta.numel() != inner or bo.numel() != inner:
        raise VdxError("temporal_attn_block: gamma/beta/bias size")
    if wqkv_packed.numel() * 2 != lib.vdx_temporal_attn_block_wqkv_bytes(inner) or \
            wo_packed.numel() * 2 != lib.vdx_temporal_attn_block_wo_bytes(inner):
        raise VdxError("temporal_attn_block: packed weight size does not match the kernel's stage layout")
    if not (wqkv_packed.is_contiguous() and wo_packed.is_contiguous()):
        raise VdxError("temporal_attn_block: packed weights must be contiguous")
    if out is None:
        out = torch.empty((M, inner), dtype=torch.float16, device=t.device)
    orow, ocol, ldo = _rows(out, "out")
    if orow < M or ocol < inner:
        raise VdxError("temporal_attn_block: out too small")
    if out.data_ptr() == t.data_ptr():
        raise VdxError("temporal_attn_block: out may not alias t")
    _lib.check(lib.vdx_temporal_attn_block_f16(_p(t, "t"), ldt, _p(gamma, "gamma"), _p(beta, "beta"), float(eps),
                                               _p(wqkv_packed, "wqkv"), _p(wo_packed, "wo"), _p(bo, "bo"),
                                               _p(out, "out"), ldo, B, F, HW, inner, float(scale), _stream()),
               "vdx_temporal_attn_block_f16")
    return out


# --------------------------------------------------------------------------------------------
def cfg_input(lat, ctx, weight, out=None):
    """fsdp_chunked_coherent.py:133-137: cat([lat]*2) (+ weight * ctx.repeat(F)).
    The result is TAGGED as a known duplicate (`is_cfg_duplicate`): the UNet then computes its text-independent blocks once.
    The tag is tied to torch's version counter, which torch operations bump and this library's kernels do NOT: never hand the
    result to a vdx op as its `out=` (nothing in this package does)."""
    lib = _lib.load()
    b, Cc, F, H, W = lat.shape
    if b != 1 or not lat.is_contiguous():
        raise VdxError("cfg_input: lat must be contiguous (1,C,F,H,W)")
    if ctx is not None and (tuple(ctx.shape) != (1, Cc, 1, H, W) or not ctx.is_contiguous()):
        raise VdxError("cfg_input: ctx must be contiguous (1,C,1,H,W)")
    if out is None:
        out = torch.empty((2, Cc, F, H, W), dtype=torch.float16, device=lat.device)
    _lib.check(lib.vdx_cfg_input_f16(_p(lat, "lat"), _p(ctx, "ctx"), float(weight), _p(out, "out"), Cc, F, H * W,
                                     _stream()), "vdx_cfg_input_f16")
    out._vdx_cfg_dup = out._version        # both batch items hold the same values until somebody writes the tensor
    return out


def is_cfg_duplicate(x) -> bool:
    """True for a tensor `cfg_input` produced and nobody has written since (torch's version counter): its two batch items
    are known to be equal, which lets the UNet compute the text-independent blocks once (unet3d.forward)."""
    tag = getattr(x, "_vdx_cfg_dup", None)
    try:
        return tag is not None and tag == x._version
    except Exception:       # inference-mode tensors do not track versions
        return False


def cfg_ddim_step(eps2, lat, guidance, coeffs, out=None):
    """fsdp_chunked_coherent.py:141-142.  coeffs = (sqrt(1-a_t), sqrt(a_t), sqrt(a_prev), sqrt(1-a_prev))."""
    lib = _lib.load()
    if eps2.shape[0] != 2 or tuple(eps2.shape[1:]) != tuple(lat.shape[1:]) or lat.shape[0] != 1:
        raise VdxError("cfg_ddim_step: eps2 must be (2,...) matching lat (1,...)")
    if not (eps2.is_contiguous() and lat.is_contiguous()):
        raise VdxError("cfg_ddim_step: tensors must be contiguous")
    if out is None:
        out = torch.empty_like(lat)
    s1, sa, sp, s1p = (float(c) for c in coeffs)
    _lib.check(lib.vdx_cfg_ddim_step_f16(_p(eps2, "eps2"), _p(lat, "lat"), _p(out, "out"), float(guidance),
                                         s1, sa, sp, s1p, lat.numel(), _stream()), "vdx_cfg_ddim_step_f16")
    return out


def ddim_step(eps, lat, coeffs, out=None):
    """`scheduler.step(eps, t, lat).prev_sample` (fsdp_chunked_coherent.py:142) without the CFG combine."""
    lib = _lib.load()
    if tuple(eps.shape) != tuple(lat.shape) or not (eps.is_contiguous() and lat.is_contiguous()):
        raise VdxError("ddim_step: eps and lat must be contiguous and of equal shape")
    if out is None:
        out = torch.empty_like(lat)
    s1, sa, sp, s1p = (float(c) for c in coeffs)
    _lib.check(lib.vdx_ddim_step_f16(_p(eps, "eps"), _p(lat, "lat"), _p(out, "out"), s1, sa, sp, s1p,
                                     lat.numel(), _stream()), "vdx_ddim_step_f16")
    return out


def blend_accumulate(full, weight, chunk, w, s, e):
    lib = _lib.load()
    _, Cc, T, H, W = full.shape
    if tuple(chunk.shape) != (1, Cc, e - s, H, W) or not (chunk.is_contiguous() and full.is_contiguous()):
        raise VdxError("blend_accumulate: chunk shape does not match range")
    if weight.numel() != T or w.numel() != e - s:
        raise VdxError("blend_accumulate: weight vectors")
    _lib.check(lib.vdx_blend_accumulate_f16(_p(full, "full"), _p(weight, "weight", torch.float32),
                                            _p(chunk, "chunk"), _p(w, "w", torch.float32), Cc, T, H * W, s, e,
                                            _stream()), "vdx_blend_accumulate_f16")


def blend_finalize(full, weight):
    lib = _lib.load()
    _, Cc, T, H, W = full.shape
    out = torch.empty(full.shape, dtype=torch.float32, device=full.device)
    _lib.check(lib.vdx_blend_finalize_f32(_p(full, "full"), _p(weight, "weight", torch.float32),
                                          _p(out, "out", torch.float32), Cc, T, H * W, _stream()),
               "vdx_blend_finalize_f32")
    return out


# --------------------------------------------------------------------------------------------
# Persistent-grid reserve and box probes (include/vdx.h, last section): not on the denoising path
def set_reserved_cus(n: int) -> int:
    """Leave `n` compute units free in every persistent grid (weights-stationary GEMMs, K5 / K7 / K8) for the channel kernels
    of a collective that runs beside the step (vdx/shard.py sets it for world > 1).  Results do not depend on it.
    Returns the CU count the persistent grids now fill."""
    lib = _lib.load()
    _lib.check(lib.vdx_set_reserved_cus(int(n)), "vdx_set_reserved_cus")
    _PLAN_CACHE.clear()          # the tiled GEMM's split rows depend on the reserve (vdx_gemm_plan: rounds of the unreserved CUs)
    return lib.vdx_persistent_grid_cus()


def reserved_cus() -> int:
    return _lib.load().vdx_reserved_cus()


def probe_mfma(device, iters: int = 4000, repeats: int = 3) -> float:
    """Sustained TFLOP/s of a FIXED dense fp16 MFMA stream on this part, in this process (`vdx_probe_mfma_f16`): what the
    box gives the instruction every matrix kernel of the library is made of.  Median of `repeats` timed launches after one
    warm-up launch (~20 ms each at the default `iters`)."""
    lib = _lib.load()
    n = 2 * torch.cuda.get_device_properties(device).multi_processor_count * 256
    scratch = torch.empty(n, dtype=torch.float32, device=device)
    flops = C.c_double(0.0)
    times = []
    for r in range(repeats + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.vdx_probe_mfma_f16(_p(scratch, "scratch", torch.float32), n, iters, C.byref(flops), _stream()), "vdx_probe_mfma_f16")
        e1.record()
        e1.synchronize()
        if r:
            times.append(e0.elapsed_time(e1))
    times.sort()
    return flops.value / (times[len(times) // 2] * 1e-3) / 1e12


def occupancy_hog(blocks: int, lds_bytes: int, micros: int, stream=None) -> None:
    """Enqueue `blocks` workgroups that hold `lds_bytes` of a CU's LDS each for `micros` us and touch no memory — a stand-in
    for the CUs a collective's channel kernels hold (one-GPU rehearsal of the distributed path, bench.py --hog)."""
    st = stream.cuda_stream if stream is not None else _stream()
    _lib.check(_lib.load().vdx_probe_occupancy_hog(int(blocks), int(lds_bytes), int(micros), st), "vdx_probe_occupancy_hog")
