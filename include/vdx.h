/* vdx.h — C-ABI of libvdx_hip.so: the MI355X (gfx950) kernels behind the
 * diffusers `UNet3DConditionModel` / `DDIMScheduler` call surface that the
 * reference's `Distribution/strategies/fsdp_chunked_coherent.py` uses.
 *
 * The reference has no native code; every entry point below
 * replaces a group of torch/diffusers operator calls reached from
 *   fsdp_chunked_coherent.py:140   noise = self.unet(x, t, encoder_hidden_states=emb).sample
 *   fsdp_chunked_coherent.py:133-137,141-142   ctx injection, CFG combine, scheduler.step
 *   fsdp_chunked_coherent.py:204-217   linear-ramp overlap blend
 * Each declaration cites the operator(s) it stands in for.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *     (the host side allocates through PyTorch-ROCm); no allocation, no sync inside;
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued, not waited for;
 *   - return 0 on success, negative on error; `vdx_last_error()` gives the message
 *     (thread-local);
 *   - activations are fp16, channels-last: a (B,C,F,H,W) tensor of diffusers is held as the
 *     row-major matrix [B*F*H*W rows][C] ("rows" = latent pixels of one frame);
 *   - all contractions accumulate in fp32 on the matrix cores (v_mfma_f32_16x16x32_f16).
 */
#ifndef VDX_H
#define VDX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vdx_stream_t;

const char* vdx_last_error(void);
int vdx_version(void);
/* 0 for the product build.  Non-zero: some translation unit was compiled with a lab macro (phase stamps, ablations — timing
 * only, some variants compute wrong results); bit = unit (1 gemm, 2 gemm_ws, 4 tattn_fused, 8 tattn2, 16 flash, 32 ff_fused,
 * 64 conv_fused, 128 xattn).  The Python binding refuses such a library unless VDX_ALLOW_LAB_BUILD=1 (the lab tools set it). */
int vdx_build_flags(void);

/* ------------------------------------------------------------------------------------------
 * GEMM / implicit-GEMM family:  out[M][N] = epilogue( A_gathered[M][K] * W[N][K]^T )
 * Replaces torch.nn.Linear / Conv2d(3x3,1x1) / Conv3d((3,1,1)) as composed by diffusers
 * ResnetBlock2D, TemporalConvLayer, Transformer2DModel, TransformerTemporalModel,
 * Downsample2D, Upsample2D, all reached from
 * fsdp_chunked_coherent.py:140.
 * ---------------------------------------------------------------------------------------- */
enum { VDX_GEMM_PLAIN = 0, VDX_GEMM_CONV3X3 = 1, VDX_GEMM_TCONV3 = 2 };
enum { VDX_EPI_GEGLU = 1 };

typedef struct vdx_gemm_args {
    const void* a;        /* fp16 source 0, row stride lda (elements)                           */
    const void* a2;       /* fp16 source 1 (channel concat after source 0) or NULL              */
    const void* w;        /* fp16 [N][K], K contiguous; gathers: K = (c/64)*T*64 + tap*64 + c%64,
                             T = 9 (tap = ky*3+kx) or 3 (tap = kt)                                 */
    const void* bias;     /* fp16 [N] or NULL                                                    */
    const void* bias2;    /* fp16 [M / rows_per_bias2][N] or NULL (time-embedding projection)    */
    const void* residual; /* fp16 [M][ldr] or NULL, added after bias                             */
    void* out;            /* fp16 [M][ldo]   (GEGLU: [M][N/2])                                   */
    int32_t M, N, K;      /* N % 64 == 0, K % 64 == 0                                            */
    int32_t mode;         /* VDX_GEMM_*                                                          */
    int32_t c1, c2;       /* channels of source 0 / 1; per-tap K = c1 + c2; both % 64 == 0       */
    int32_t lda, lda2, ldo, ldr;
    int32_t h_in, w_in;   /* conv3x3: source image size (rows of `a` = n*h_in*w_in)              */
    int32_t h_out, w_out; /* conv3x3: output image size (M = n*h_out*w_out)                      */
    int32_t stride;       /* conv3x3: 1 or 2 (pad 1)                                             */
    int32_t upsample;     /* conv3x3: 1 = source is nearest-x2 upsampled on the fly; 2 = nearest-upsampled to
                           * (h_out, w_out) — diffusers' `upsample_size` path for latents not divisible by 8 */
    int32_t frames, hw;   /* tconv3: M = b*frames*hw, taps step `hw` rows, zero pad in time      */
    int32_t rows_per_bias2, ldb2; /* bias2 row = m / rows_per_bias2, row stride ldb2 elements        */
    int32_t epilogue;     /* VDX_EPI_* flags                                                     */
    int32_t row_begin, row_end; /* only output rows [row_begin, row_end) are computed (row_end 0 = M); every pointer
                             still addresses row 0 and M stays the whole product's row count.  Lets a caller cover one
                             product with two calls that use different tile shapes (vdx_gemm_plan); the results are
                             bit-identical however the rows are split                                 */
    int32_t ksplit;       /* > 1: the rows of this call are computed on 256x320 tiles as `ksplit` slices of K per tile
                             (fp32 partial slabs in `workspace`) + a fixed-order reduction that runs the epilogue.  Fills
                             the chip when the call has far fewer than 256 tiles (the tail of a product).  Changes the
                             summation order of these rows (NOT bit-identical to ksplit = 0; deterministic)          */
    int32_t wset_rows;    /* > 0: ONE WEIGHT SET PER `wset_rows` ROWS — rows [s*wset_rows, (s+1)*wset_rows) use w + s*N*K and
                             wset_bias + s*N (a GroupNorm folded into this Linear: vdx_groupnorm_fold_linear_f16).  Plain
                             mode on the weights-stationary kernels only (K = 320 / 640, M and wset_rows multiples of 64),
                             no bias / bias2 / residual / GEGLU                                                    */
    void* workspace;      /* ksplit > 1: >= tiles * ksplit * 327 680 bytes (vdx_gemm_plan_ksplit), 16-byte aligned    */
    const float* wset_bias; /* wset_rows > 0: fp32 [M / wset_rows][N], the initial accumulators                          */
    size_t workspace_bytes; /* ksplit > 1: the size of `workspace`; the call is refused when the slabs would not fit        */
} vdx_gemm_args;

int vdx_gemm_f16(const vdx_gemm_args* a, vdx_stream_t stream);

/* What vdx_gemm_f16 would do with `a` (host-only, launches nothing): *variant = the kernel family it picks for rows
 * [row_begin, row_end) (1 = 128x128, 2 = 256x320, 8 = 128x320 ring, 5 = 256x64, 7 = weights-stationary), *split_row = a row at
 * which splitting the product into two calls ([row_begin, split_row) and [split_row, row_end)) is expected to be faster
 * (whole rounds of 256 big tiles + a tail of small ones instead of a mostly idle last round), or 0.                  */
int vdx_gemm_plan(const vdx_gemm_args* a, int32_t* variant, int32_t* split_row);
/* The same question with a split-K tail allowed (whole products only): rows [0, *split_row) as one ordinary call, rows
 * [*split_row, M) as one call with ksplit = *ksplit and a workspace of *workspace_bytes; *ksplit = 0 when vdx_gemm_plan's
 * answer is at least as good.  Pays on the 16-frame windows of BASELINE cfg4 / cfg5, whose row counts leave 1/8 - 1/2 of a
 * round of big tiles (level 2: 288 tiles on 256 CUs).                                                              */
int vdx_gemm_plan_ksplit(const vdx_gemm_args* a, int32_t* split_row, int32_t* ksplit, size_t* workspace_bytes);

/* conv_in gather: (B,Cin,F,H,W) fp16 latent -> im2col rows [B*F*H*W][Kpad], K = (ky*3+kx)*Cin + ci,
 * zero padded to Kpad (multiple of 64); conv_in = this + vdx_gemm_f16 with w [Cout][Kpad]
 * (UNet3DConditionModel.conv_in after the permute/reshape).                         */
int vdx_im2col_in_f16(const void* x_ncfhw, void* out_rows, int B, int Cin, int F, int H, int W,
                      int Kpad, vdx_stream_t stream);

/* channels-last rows [B*F*H*W][ld] (first C columns) -> (B,C,F,H,W) fp16 (UNet output permute) */
int vdx_rows_to_ncfhw_f16(const void* rows, int ld, void* out, int B, int C, int F, int H, int W,
                          vdx_stream_t stream);

/* y = x * sigmoid(x), n elements (TimestepEmbedding act / ResnetBlock2D.nonlinearity(temb)) */
int vdx_silu_f16(const void* x, void* y, size_t n, vdx_stream_t stream);

/* Sinusoidal timestep embedding of UNet3DConditionModel (`Timesteps(320, flip_sin_to_cos=True, shift 0)`):
 * out[b][0:dim/2] = cos(t*f_j), out[b][dim/2:] = sin(t*f_j), f_j = exp(-ln(1e4) j / (dim/2)); fp32 math, fp16 store.
 * `t_device` points at ONE fp32 timestep in device memory (all B batch items share it, fsdp_chunked_coherent.py:140). */
int vdx_timestep_embedding_f16(const float* t_device, void* out, int B, int dim, vdx_stream_t stream);

/* y = gelu(x) (exact, erf), n elements: CLIPMLP's activation between fc1 and fc2 (hidden_act "gelu") */
int vdx_gelu_f16(const void* x, void* y, size_t n, vdx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Normalisation (torch.nn.GroupNorm on 4-D and 5-D inputs, torch.nn.LayerNorm).
 * GroupNorm over `rows_per_sample` rows x (C/G) channels per (sample, group):
 *   4-D GN: rows_per_sample = H*W (sample = one frame); 5-D GN: rows_per_sample = F*H*W.
 * Two sources (x | x2) cover the skip concatenation in up blocks.
 * ---------------------------------------------------------------------------------------- */
size_t vdx_groupnorm_workspace(int n_samples, int rows_per_sample, int C, int G);
/* y[M][C] = act( (x - mean) * rstd * gamma + beta ), act = SiLU if silu != 0                   */
int vdx_groupnorm_f16(const void* x, int c1, int ldx, const void* x2, int c2, int ldx2,
                      const void* gamma, const void* beta, float eps, int G,
                      int n_samples, int rows_per_sample, int silu,
                      void* y, int ldy, void* workspace, vdx_stream_t stream);
/* The same with the row-slab partition of the statistics fixed as for `partition_samples` samples (0 = n_samples): a
 * sample's result then has the same bits alone, in part of a batch or in the whole batch (callers that split a batch
 * to save memory, or decode frames one by one like fsdp_chunked_coherent.py:219-225, pass the full batch size). */
size_t vdx_groupnorm_workspace_part(int n_samples, int rows_per_sample, int C, int G, int partition_samples);
/* GroupNorm folded into the Linear that follows it without an activation (Transformer2DModel / TransformerTemporalModel:
 * `norm` -> `proj_in`): the statistics pass of vdx_groupnorm_part_f16, then per sample s
 *   w_out[s] = fp16(w diag(scale_s))  [N][C],   bias_out[s] = bias + w.beta - w_out[s].mean_s  (fp32 [N])
 * so that  vdx_gemm_f16(x, w_out, wset_rows = rows_per_sample, wset_bias = bias_out)  ==  Linear(GroupNorm(x)) without
 * the normalised tensor ever being written.  w_out: n_samples*N*C fp16, bias_out: n_samples*N fp32.              */
int vdx_groupnorm_fold_linear_f16(const void* x, int C, int ldx, const void* gamma, const void* beta, float eps, int G,
                                  int n_samples, int rows_per_sample, void* workspace, int partition_samples,
                                  const void* w, const void* bias, int N, void* w_out, void* bias_out, vdx_stream_t stream);
int vdx_groupnorm_part_f16(const void* x, int c1, int ldx, const void* x2, int c2, int ldx2,
                           const void* gamma, const void* beta, float eps, int G,
                           int n_samples, int rows_per_sample, int silu,
                           void* y, int ldy, void* workspace, int partition_samples, vdx_stream_t stream);
/* The statistics pass alone: leaves scale[s][c] = rstd_s,g * gamma_c and shift[s][c] = beta_c - mean_s,g * scale as
 * [n_samples][C][2] fp32 at byte *scale_shift_offset of `workspace` (vdx_groupnorm_workspace_part bytes), for a consumer that
 * applies the normalisation itself (vdx_tconv_gn_f16).                                                               */
int vdx_groupnorm_stats_f16(const void* x, int c1, int ldx, const void* x2, int c2, int ldx2, const void* gamma,
                            const void* beta, float eps, int G, int n_samples, int rows_per_sample, void* workspace,
                            int partition_samples, size_t* scale_shift_offset, vdx_stream_t stream);
/* K3 — TemporalConvLayer's Sequential(GroupNorm, SiLU, Conv3d (3,1,1)) with the normalisation applied INSIDE the convolution
 * (SURVEY.md §2.3 row TemporalConvLayer, App. A.4; reached four times per layer from fsdp_chunked_coherent.py:140):
 *   out[(b*F + f)*S + p][n] = bias[n] + residual + sum_kt sum_c w[n][(c/64)*192 + kt*64 + c%64] * silu(x[(b*F + f+kt-1)*S + p][c] * scale[b][c] + shift[b][c])
 * with zero rows for frames outside [0, F) (the padding applies to the normalised tensor).  x: raw rows [B*F*S][ldx];
 * scale_shift: [B][C][2] fp32 from vdx_groupnorm_stats_f16 (n_samples = B, rows_per_sample = F*S); w: the packed temporal
 * weights of vdx_gemm_f16's VDX_GEMM_TCONV3 mode ([N][3*C]).  Supported: C % 64 == 0, N % 320 == 0, F % 8 == 0
 * (vdx_tconv_gn_supported); other shapes take vdx_groupnorm_f16 + vdx_gemm_f16.                                        */
/* K1 — ResnetBlock2D's conv(SiLU(GroupNorm(x))) with the normalisation applied INSIDE the 3x3 convolution (SURVEY.md §2.3 row
 * ResnetBlock2D, App. A.3; conv1 and conv2 of every ResNet block, reached from fsdp_chunked_coherent.py:140):
 *   out[n*h*w + y*w + x][o] = bias[o] + bias2[(n*h*w) / rows_per_bias2][o] + residual + sum_{ky,kx,c} w[o][(c/64)*576 + (ky*3+kx)*64 + c%64]
 *                             * silu(cat(a, a2)[n*h*w + (y+ky-1)*w + (x+kx-1)][c] * scale[n][c] + shift[n][c])
 * with zero for pixels outside the image (the padding applies to the normalised tensor).  a / a2: raw rows (a2 / c2 = the skip
 * tensor of the up blocks, or NULL / 0); scale_shift: [n_img][c1 + c2][2] fp32 from vdx_groupnorm_stats_f16 (n_samples = n_img,
 * rows_per_sample = h*w); w: the packed weights of vdx_gemm_f16's VDX_GEMM_CONV3X3 mode.  Stride 1, no upsampling.
 * Supported: c1 % 64 == 0, c2 % 64 == 0, N % 320 == 0; vdx_conv3x3_gn_preferred: expected to beat vdx_groupnorm_f16 + vdx_gemm_f16
 * (one column tile, image width a multiple of 32, the chip filled twice over: level 0 of the XL UNet).                */
int vdx_conv3x3_gn_supported(int c1, int c2, int N);
int vdx_conv3x3_gn_preferred(int c1, int c2, int N, int n_img, int h, int w);
int vdx_conv3x3_gn_f16(const void* a, int lda, const void* a2, int lda2, int c1, int c2, const float* scale_shift,
                       const void* w, const void* bias, const void* bias2, int rows_per_bias2, int ldb2,
                       const void* residual, int ldr, void* out, int ldo, int n_img, int h, int w_px, int N,
                       vdx_stream_t stream);
int vdx_tconv_gn_supported(int C, int N, int F);
/* 1 when K3 is expected to beat vdx_groupnorm_f16 + vdx_gemm_f16 on this shape (one column tile, the chip filled twice over) */
int vdx_tconv_gn_preferred(int C, int N, int B, int F, int S);
int vdx_tconv_gn_f16(const void* x, int ldx, const float* scale_shift, const void* w, const void* bias,
                     const void* residual, int ldr, void* out, int ldo, int B, int F, int S, int C, int N,
                     vdx_stream_t stream);
/* In-place softmax over the first `cols` columns of each of `rows` rows of x (row stride ld), logits scaled by
 * `scale` in fp32: the probabilities of AutoencoderKL's mid-block attention (one 512-channel head over h*w
 * tokens; diffusers Attention with upcast softmax), reached from fsdp_chunked_coherent.py:223.            */
int vdx_softmax_rows_f16(void* x, int ld, int rows, int cols, float scale, vdx_stream_t stream);

int vdx_layernorm_f16(const void* x, int ldx, const void* gamma, const void* beta, float eps,
                      int M, int C, void* y, int ldy, vdx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Attention cores (diffusers `Attention` with plain softmax, scale = d^-0.5, head dim 64).
 * ---------------------------------------------------------------------------------------- */
/* Flash-style attention over contiguous sequences (spatial self-attention and text
 * cross-attention of Transformer2DModel).
 *   q  : fp16 rows [n_seq*sq][ldq], head h at columns h*64..h*64+63
 *   k  : fp16 rows [n_kv*skv_pad][ldk]
 *   vt : fp16 [heads*64][ldvt]  V transposed: vt[h*64+d][kvb*skv_pad + key]
 *   kv batch of sequence s is s / seq_per_kv (cross-attn: all frames of a sample share text).
 *   causal != 0: query i sees keys <= i only (CLIPTextModel's self-attention, fsdp_chunked_coherent.py:102).
 *   out: fp16 rows [n_seq*sq][ldo].
 *   Supported score range: |q.k * scale| <= 5000 nat.  The kernel's lazy softmax offset is carried as an fp16 MFMA
 *   operand; up to that magnitude its spacing (<= 4 exp2-units) keeps a re-centred row inside the window in which
 *   no further move is needed; it is clamped at +-60000 exp2-units, beyond ~11000 nat rows can stay mis-centred.  */
int vdx_flash_attn_f16(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt,
                       void* out, int ldo, int n_seq, int sq, int skv, int skv_pad, int heads,
                       int seq_per_kv, float scale, int causal, vdx_stream_t stream);
/* The same with V as ROWS: v fp16 [n_kv*skv_pad][ldv], head h at columns h*64.. like k — q, k and v can then be the
 * three column blocks of ONE projection's output (spatial self-attention of Transformer2DModel: no transposed V
 * product).  The V tile is staged like K and transposed by the LDS read (ds_read_b64_tr_b16).  skv_pad >= skv, any
 * value (a V^T row of 8-key chunks is what asks for a multiple of 8 above); rows skv..skv_pad-1 must hold finite values. */
int vdx_flash_attn_rows_f16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv,
                            void* out, int ldo, int n_seq, int sq, int skv, int skv_pad, int heads,
                            int seq_per_kv, float scale, int causal, vdx_stream_t stream);

/* K8 — the feed-forward sub-block of BasicTransformerBlock (`t = t + ff(norm3(t))`, GEGLU with the
 * erf GELU, both in Transformer2DModel and TransformerTemporalModel) as ONE kernel: LayerNorm -> [val | gate] projection ->
 * val * gelu(gate) -> output projection (+bias) + residual; the [rows][4*inner] intermediate never leaves the CU
 * (csrc/ff_fused.hip).  Built for inner 320 (level 0).
 *   t, out : fp16 rows [M][ld], `inner` columns used; out may not alias t
 *   packed : vdx/packing.py pack_k8 (LayerNorm's affine folded into the first projection), vdx_ff_block_pack_bytes bytes */
int vdx_ff_block_supported(int inner);
size_t vdx_ff_block_pack_bytes(int inner);
int vdx_ff_block_f16(const void* t, int ldt, const void* packed, float eps, void* out, int ldo, int M, int inner,
                     vdx_stream_t stream);
/* The same kernel with the transformer's `proj_out` and its residual behind the feed-forward (the sub-block's output never
 * reaches HBM): out[r] = x[r % xrows] + W_p . (t[r] + ff(LayerNorm(t[r]))) + b_p.  x: the transformer's input rows, xrows = M, or
 * M / 2 when both halves of the batch pair with the same rows of x (the CFG-shared prefix); proj_packed: vdx/packing.py
 * pack_k8_proj (25 weight units, fp32 b_p; vdx_ff_block_proj_pack_bytes bytes).  out may alias neither t nor x. */
size_t vdx_ff_block_proj_pack_bytes(int inner);
int vdx_ff_block_proj_f16(const void* t, int ldt, const void* packed, float eps, const void* x, int ldx, int xrows,
                          const void* proj_packed, void* out, int ldo, int M, int inner, vdx_stream_t stream);

/* Temporal self-attention of TransformerTemporalModel: sequences run over the
 * F frames of one latent pixel.  qkv: fp16 rows [B*F*HW][ldqkv] = [q | k | v] each heads*64
 * wide, row = (b*F + f)*HW + p.  out rows likewise, [ldo] wide.  F <= 128.                     */
int vdx_temporal_attn_f16(const void* qkv, int ldqkv, void* out, int ldo, int B, int F, int HW,
                          int heads, float scale, vdx_stream_t stream);

/* K7 — one attention sub-block of TransformerTemporalModel (`s = s + attn(LN(s))`, both attn1 and the
 * "double self-attention" attn2) as ONE kernel: LayerNorm -> q|k|v -> softmax over the F frames of each latent pixel
 * -> P.V -> to_out.0 (+bias) + residual.  Rows are read once and written once (csrc/tattn_fused.hip).
 *   t, out : fp16 rows [B*F*HW][ld], row = (b*F + f)*HW + p, `inner` columns used; out may not alias t
 *   gamma, beta : LayerNorm affine [inner];  bo : to_out.0 bias [inner]
 *   wqkv_packed / wo_packed : weight stage images (vdx/packing.py pack_k7_qkv / pack_k7_out), sizes given by
 *   vdx_temporal_attn_block_wqkv_bytes / _wo_bytes.
 * Supported: inner 320 or 512, F a divisor of 48 (vdx_temporal_attn_block_supported); callers use the separate
 * LayerNorm / GEMM / vdx_temporal_attn_f16 kernels otherwise.                                              */
int vdx_temporal_attn_block_supported(int inner, int F);
size_t vdx_temporal_attn_block_wqkv_bytes(int inner);
size_t vdx_temporal_attn_block_wo_bytes(int inner);
int vdx_temporal_attn_block_f16(const void* t, int ldt, const void* gamma, const void* beta, float eps,
                                const void* wqkv_packed, const void* wo_packed, const void* bo,
                                void* out, int ldo, int B, int F, int HW, int inner, float scale,
                                vdx_stream_t stream);

/* K5 (csrc/xattn.hip) — the cross-attention sub-block of diffusers' BasicTransformerBlock in a spatial transformer
 * (`t = t + attn2(norm2(t), encoder_hidden_states)`; call site fsdp_chunked_coherent.py:140) as one kernel:
 *   out[r] = t[r] + W_o . softmax_k( c . (W_q LN(t[r]))_h . K[item(r)]_h[k] ) V[item(r)]_h + b_o      per head h, k < kv_len
 * t / out: [n_items * rows_per_item][inner] fp16 rows, item(r) = r / rows_per_item; `packed`: vdx/packing.py pack_k5 (q units
 * with LayerNorm's affine and the scale folded in, output-projection units, fp32 q bias, fp32 b_o;
 * vdx_cross_attn_block_pack_bytes bytes); `kv_packed`: the text keys / values of every item in MFMA-fragment order
 * (pack_k5_kv: n_items * vdx_cross_attn_block_kv_bytes bytes), 80 key slots of which the first kv_len are used.
 * out may not alias t.  Supported: inner 320, 1 <= kv_len <= 80.                                                          */
int vdx_cross_attn_block_supported(int inner, int kv_len);
size_t vdx_cross_attn_block_pack_bytes(int inner);
size_t vdx_cross_attn_block_kv_bytes(int inner);
int vdx_cross_attn_block_f16(const void* t, int ldt, const void* packed, const void* kv_packed, int kv_len, float eps,
                             void* out, int ldo, int n_items, int rows_per_item, int inner, vdx_stream_t stream);

/* K7, second design (csrc/tattn2.hip) — the same sub-block (`s = s + attn(LN(s))`; call site
 * fsdp_chunked_coherent.py:140) with LayerNorm's affine, the softmax scale and all biases folded into ONE packed blob
 * (vdx/packing.py pack_k7b: q|k|v units, output-projection units with the permuted k index, fp32 q bias, fp32 output
 * bias; vdx_temporal_attn_block2_pack_bytes bytes).  t / out as above.  Supported: inner 320, F a divisor of 48.   */
int vdx_temporal_attn_block2_supported(int inner, int F);
size_t vdx_temporal_attn_block2_pack_bytes(int inner);
int vdx_temporal_attn_block2_f16(const void* t, int ldt, const void* packed, float eps, void* out, int ldo,
                                 int B, int F, int HW, int inner, vdx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Orchestration ops the reference owns (fsdp_chunked_coherent.py).
 * ---------------------------------------------------------------------------------------- */
/* :133-137  x = cat[lat,lat] (+ w * ctx.repeat(F));  lat (1,C,F,H,W), ctx (1,C,1,H,W) or NULL  */
int vdx_cfg_input_f16(const void* lat, const void* ctx, float weight, void* x2, int C, int F,
                      int HW, vdx_stream_t stream);
/* :141-142  lat' = DDIM.step(u + gs*(c-u), t, lat)  with the four scheduler coefficients;
 * eps2 (2,C,F,H,W) = [u; c].  Rounds to fp16 after every tensor op, like torch.              */
int vdx_cfg_ddim_step_f16(const void* eps2, const void* lat, void* lat_out, float guidance,
                          float sqrt_one_minus_at, float sqrt_at, float sqrt_aprev,
                          float sqrt_one_minus_aprev, size_t n, vdx_stream_t stream);
/* :142 alone: lat' = DDIM.step(eps, t, lat) (no CFG combine) — what `scheduler.step` of the
 * unchanged reference script binds to.                                                        */
int vdx_ddim_step_f16(const void* eps, const void* lat, void* lat_out, float sqrt_one_minus_at,
                      float sqrt_at, float sqrt_aprev, float sqrt_one_minus_aprev, size_t n,
                      vdx_stream_t stream);
/* :204-217 one chunk's contribution: full[s:e] += lat*w (fp16 accumulator), weight[s:e] += w;
 * w = fp32 device vector of length e-s (the linear ramps, built by the host exactly as :207-213) */
int vdx_blend_accumulate_f16(void* full, float* weight, const void* chunk, const float* w, int C,
                             int T, int HW, int s, int e, vdx_stream_t stream);
/* :217 lat = full / clamp(weight, 1e-6) -> fp32                                                */
int vdx_blend_finalize_f32(const void* full, const float* weight, float* out, int C, int T,
                           int HW, vdx_stream_t stream);

/* Decoded frames -> uint8 HWC exactly as fsdp_chunked_coherent.py:224-225 maps them
 * ((sample*0.5+0.5).clamp(0,1), *255, .byte(); fp16 rounding after every op, truncation at the end).
 * rows: the decoder's channels-last output rows [n_pixels][ld], RGB in the first 3 columns.            */
int vdx_rows_to_u8_frames(const void* rows, int ld, size_t n_pixels, void* out_u8, vdx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Exchange steps on RCCL (resolved at run time: no link-time dependency).  One communicator per process / GPU.
 *   vdx_comm_unique_id : rank 0 fills a 128-byte id and shares it with the other ranks out of band
 *   vdx_comm_init      : collective; every rank passes the same id
 *   vdx_allgather_shard: full[r*shard_bytes ..] = rank r's shard — the per-unit parameter gather that stands in for
 *                        FSDP's flat-parameter all-gather (fsdp_chunked_coherent.py:63-88), once per shard unit per step
 *   vdx_halo_exchange  : send `send_bytes` to rank `send_to` and receive `recv_bytes` from rank `recv_from` as one group
 *                        (either side may be 0 bytes) — the overlap frames of the post-loop exchange (:190-202)
 * Both enqueue on `side_stream` and return; the caller orders them against its compute stream with events. */
typedef struct vdx_comm vdx_comm;
/* Peer-mapped shards — the same parameter gather WITHOUT a collective and without compute units: weights
 * are read-only after load, so every rank exports the allocation holding its shards once, maps the others', and pulls.
 *   vdx_ipc_export : 64-byte HIP IPC handle of the allocation `dev_ptr` lies in + its byte offset inside it
 *   vdx_ipc_open   : map a peer's export (another process; same or another GPU of the node) -> device pointer here
 *   vdx_ipc_close  : unmap (pointer and offset as returned / passed above)
 *   vdx_peer_gather: full[r*shard_bytes ..] = srcs[r][0 .. shard_bytes) for r < world, as device-to-device copies on
 *                    `side_stream` (copy engines; srcs[own rank] = the local shard)                              */
int vdx_ipc_export(const void* dev_ptr, void* handle64, size_t* offset_bytes);
int vdx_ipc_open(const void* handle64, size_t offset_bytes, void** dev_ptr);
int vdx_ipc_close(void* dev_ptr, size_t offset_bytes);
int vdx_peer_gather(void* full, const void* const* srcs, int world, size_t shard_bytes, vdx_stream_t side_stream);
int vdx_comm_unique_id(void* id128);
int vdx_comm_init(const void* id128, int rank, int world, vdx_comm** out);
int vdx_comm_destroy(vdx_comm* comm);
int vdx_allgather_shard(vdx_comm* comm, const void* shard, void* full, size_t shard_bytes, vdx_stream_t side_stream);
int vdx_halo_exchange(vdx_comm* comm, const void* send_buf, size_t send_bytes, int send_to, void* recv_buf,
                      size_t recv_bytes, int recv_from, vdx_stream_t side_stream);

/* ------------------------------------------------------------------------------------------
 * Exact-fit grids and the CUs a collective holds.  The weights-stationary GEMMs and the fused sub-block kernels (K5, K7,
 * K8) launch one workgroup per compute unit, and vdx_gemm_plan covers a product with whole rounds of one 256x320 tile per
 * CU + a tail.  While a parameter gather (fsdp_chunked_coherent.py:63-88's FSDP all-gather; here RCCL on a side stream)
 * runs beside the step, each of its channel kernels holds a CU, and an exact-fit launch then runs a whole round more.
 * `vdx_set_reserved_cus(r)` makes every persistent grid and every such main launch leave r CUs free per round (rounded so
 * the grid stays a multiple of 8 = the XCD count; the tail launch takes the rows left over); results do not depend on r,
 * bit for bit.  Process-wide, not thread-safe against concurrent launches; a caller that caches vdx_gemm_plan's answers
 * must drop them when r changes (vdx/ops.py does).  Default 0; vdx/shard.py sets VDX_RESERVED_CUS for a world > 1.       */
int vdx_set_reserved_cus(int n);
int vdx_reserved_cus(void);
int vdx_persistent_grid_cus(void);

/* Box probes — not on the denoising path; bench.py's `box` object and the one-GPU rehearsal of the distributed path.
 *   vdx_probe_mfma_f16      : a fixed dense v_mfma_f32_32x32x16_f16 stream (2 waves / SIMD, per-lane operands) on every CU;
 *                             *flops = its FLOP count; time it with events around the call -> the part's sustained MFMA rate
 *   vdx_probe_occupancy_hog : `blocks` workgroups x 256 threads holding `lds_bytes` of LDS each for `micros` us, touching
 *                             no memory: stands in for the CUs a collective's channel kernels hold                        */
int vdx_probe_mfma_f16(float* out, size_t out_floats, int iters, double* flops, vdx_stream_t stream);
int vdx_probe_occupancy_hog(int blocks, int lds_bytes, int micros, vdx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VDX_H */
