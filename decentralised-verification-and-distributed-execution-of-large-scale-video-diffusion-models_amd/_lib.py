"""ctypes binding of libvdx_hip.so (C-ABI declared in include/vdx.h).

The product path has no CPU fallback: if the library is missing or a symbol is absent,
loading raises.  Nothing here imports `oracle/`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VDX_LIB_PATH") or os.path.join(_HERE, "libvdx_hip.so")   # override: dev/diagnostic builds


class GemmArgs(C.Structure):
    """Mirror of `vdx_gemm_args` (include/vdx.h)."""
    _fields_ = [
        ("a", C.c_void_p), ("a2", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p),
        ("bias2", C.c_void_p), ("residual", C.c_void_p), ("out", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("mode", C.c_int32),
        ("c1", C.c_int32), ("c2", C.c_int32),
        ("lda", C.c_int32), ("lda2", C.c_int32), ("ldo", C.c_int32), ("ldr", C.c_int32),
        ("h_in", C.c_int32), ("w_in", C.c_int32), ("h_out", C.c_int32), ("w_out", C.c_int32),
        ("stride", C.c_int32), ("upsample", C.c_int32), ("frames", C.c_int32), ("hw", C.c_int32),
        ("rows_per_bias2", C.c_int32), ("ldb2", C.c_int32), ("epilogue", C.c_int32),
        ("row_begin", C.c_int32), ("row_end", C.c_int32),
        ("ksplit", C.c_int32), ("wset_rows", C.c_int32), ("workspace", C.c_void_p), ("wset_bias", C.c_void_p),
        ("workspace_bytes", C.c_size_t),
    ]


_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes): every symbol include/vdx.h declares
SIGNATURES = {
    "vdx_last_error": (C.c_char_p, []),
    "vdx_version": (_i, []),
    "vdx_build_flags": (_i, []),
    "vdx_gemm_f16": (_i, [C.POINTER(GemmArgs), _vp]),
    "vdx_gemm_plan": (_i, [C.POINTER(GemmArgs), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "vdx_gemm_plan_ksplit": (_i, [C.POINTER(GemmArgs), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_size_t)]),
    "vdx_softmax_rows_f16": (_i, [_vp, _i, _i, _i, _f, _vp]),
    "vdx_rows_to_u8_frames": (_i, [_vp, _i, _sz, _vp, _vp]),
    "vdx_im2col_in_f16": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "vdx_rows_to_ncfhw_f16": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "vdx_silu_f16": (_i, [_vp, _vp, _sz, _vp]),
    "vdx_groupnorm_workspace": (_sz, [_i, _i, _i, _i]),
    "vdx_groupnorm_f16": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _f, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "vdx_groupnorm_workspace_part": (_sz, [_i, _i, _i, _i, _i]),
    "vdx_groupnorm_part_f16": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _f, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    "vdx_groupnorm_fold_linear_f16": (_i, [_vp, _i, _i, _vp, _vp, _f, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "vdx_groupnorm_stats_f16": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _f, _i, _i, _i, _vp, _i, C.POINTER(_sz), _vp]),
    "vdx_conv3x3_gn_supported": (_i, [_i, _i, _i]),
    "vdx_conv3x3_gn_preferred": (_i, [_i, _i, _i, _i, _i, _i]),
    "vdx_conv3x3_gn_f16": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "vdx_tconv_gn_supported": (_i, [_i, _i, _i]),
    "vdx_tconv_gn_preferred": (_i, [_i, _i, _i, _i, _i]),
    "vdx_tconv_gn_f16": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "vdx_layernorm_f16": (_i, [_vp, _i, _vp, _vp, _f, _i, _i, _vp, _i, _vp]),
    "vdx_flash_attn_f16": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "vdx_flash_attn_rows_f16": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "vdx_timestep_embedding_f16": (_i, [_vp, _vp, _i, _i, _vp]),
    "vdx_gelu_f16": (_i, [_vp, _vp, _sz, _vp]),
    "vdx_temporal_attn_f16": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "vdx_ff_block_supported": (_i, [_i]),
    "vdx_ff_block_pack_bytes": (_sz, [_i]),
    "vdx_ff_block_f16": (_i, [_vp, _i, _vp, _f, _vp, _i, _i, _i, _vp]),
    "vdx_ff_block_proj_pack_bytes": (_sz, [_i]),
    "vdx_ff_block_proj_f16": (_i, [_vp, _i, _vp, _f, _vp, _i, _i, _vp, _vp, _i, _i, _i, _vp]),
    "vdx_cross_attn_block_supported": (_i, [_i, _i]),
    "vdx_cross_attn_block_pack_bytes": (_sz, [_i]),
    "vdx_cross_attn_block_kv_bytes": (_sz, [_i]),
    "vdx_cross_attn_block_f16": (_i, [_vp, _i, _vp, _vp, _i, _f, _vp, _i, _i, _i, _i, _vp]),
    "vdx_temporal_attn_block_supported": (_i, [_i, _i]),
    "vdx_temporal_attn_block_wqkv_bytes": (_sz, [_i]),
    "vdx_temporal_attn_block_wo_bytes": (_sz, [_i]),
    "vdx_temporal_attn_block_f16": (_i, [_vp, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "vdx_temporal_attn_block2_supported": (_i, [_i, _i]),
    "vdx_temporal_attn_block2_pack_bytes": (_sz, [_i]),
    "vdx_temporal_attn_block2_f16": (_i, [_vp, _i, _vp, _f, _vp, _i, _i, _i, _i, _i, _vp]),
    "vdx_comm_unique_id": (_i, [_vp]),
    "vdx_comm_init": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "vdx_comm_destroy": (_i, [_vp]),
    "vdx_allgather_shard": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "vdx_halo_exchange": (_i, [_vp, _vp, _sz, _i, _vp, _sz, _i, _vp]),
    "vdx_ipc_export": (_i, [_vp, _vp, C.POINTER(_sz)]),
    "vdx_ipc_open": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "vdx_ipc_close": (_i, [_vp, _sz]),
    "vdx_peer_gather": (_i, [_vp, C.POINTER(_vp), _i, _sz, _vp]),
    "vdx_set_reserved_cus": (_i, [_i]),
    "vdx_reserved_cus": (_i, []),
    "vdx_persistent_grid_cus": (_i, []),
    "vdx_probe_mfma_f16": (_i, [_vp, _sz, _i, C.POINTER(C.c_double), _vp]),
    "vdx_probe_occupancy_hog": (_i, [_i, _i, _i, _vp]),
    "vdx_cfg_input_f16": (_i, [_vp, _vp, _f, _vp, _i, _i, _i, _vp]),
    "vdx_cfg_ddim_step_f16": (_i, [_vp, _vp, _vp, _f, _f, _f, _f, _f, _sz, _vp]),
    "vdx_ddim_step_f16": (_i, [_vp, _vp, _vp, _f, _f, _f, _f, _sz, _vp]),
    "vdx_blend_accumulate_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "vdx_blend_finalize_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
}

_lib = None


def source_sha() -> str:
    """sha256 over the kernel sources (csrc/*.hip, *.h, include/vdx.h): the identity of the build a profile was
    taken on.  `.git` does not travel to the GPU box, so profiles/ records this instead of a commit id."""
    import glob
    import hashlib
    hsh = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "vdx.h"))
    for f in files:
        hsh.update(os.path.basename(f).encode())
        hsh.update(open(f, "rb").read())
    return hsh.hexdigest()[:16]


class VdxError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the HIP library; raise loudly when it is not built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VdxError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the denoising path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    flags = lib.vdx_build_flags()
    if flags and os.environ.get("VDX_ALLOW_LAB_BUILD") != "1":
        # VDX_LIB_PATH must not slip a stamps / ablation build (timing-only code paths, some with wrong results) into the product
        raise VdxError(f"{LIB_PATH} was built with lab macros (vdx_build_flags() = {flags}): not a parity build. "
                       "Lab tools set VDX_ALLOW_LAB_BUILD=1; the product path never does.")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().vdx_last_error()
        raise VdxError(f"{what}: {msg.decode() if msg else 'error'} (rc={rc})")
