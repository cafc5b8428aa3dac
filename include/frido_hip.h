/*
 * frido_hip.h — C ABI of libfrido_hip.so, the MI355X (gfx950) hot path of the Frido sampler.
 *
 * The reference (davidhalladay/Frido) is 100 % Python/PyTorch and has no FFI of its own
 * (SURVEY.md §2b, §8b): every device op there is an ATen call made from the files cited below.
 * Each entry point here replaces one cluster of those call sites.  Conventions:
 *   - extern "C", plain pointers + sizes, no torch types; all pointers are DEVICE pointers
 *     owned by the caller (PyTorch-ROCm tensors, or hipMalloc from C), 16-byte aligned;
 *   - every launcher returns 0 on success, a negative FRIDO_E* code on bad arguments or a HIP
 *     error (never throws, never exits); kernels are enqueued on the caller's hipStream_t and
 *     are hipGraph-capturable (no allocation, no synchronisation inside);
 *   - activations are NHWC fp32 ("f32") or NHWC 16-bit "operand" tensors.  An operand tensor is
 *     a matrix [rows][K] of 16-bit elements with K contiguous (`frido_bf16` in the signatures is
 *     the 16-bit storage type).  One-plane mode (nsplit == 1): bf16.  Two-plane mode (nsplit == 2,
 *     the "bf16x3" precision keyword): a second plane holding the rounding residual lives `lo`
 *     elements after the first, products are accumulated as hi*hi + hi*lo + lo*hi in fp32 on the
 *     MFMA pipe; both planes are fp16 (frido_x3_plane_format() == 1) or, in a -DFRIDO_X3_F16=0 build, bf16 (≈2^-17
 *     relative, fp32's range).  fp16 pairs: representation error max(2^-22 |v|, 2^-25) -- relative down to |v| = 2^-3, an
 *     absolute floor below (the lo plane is an fp16 subnormal there); values beyond +-65504 SATURATE at that bound when a
 *     kernel of this library (or frido_amd.engine.pack_matrix) produces the operand -- a host packing operands itself must
 *     clamp the same way, an inf plane makes the three-pass product NaN.
 */
#ifndef FRIDO_HIP_H
#define FRIDO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* frido_stream_t;   /* a hipStream_t */
typedef uint16_t frido_bf16;

/* Bumped on every incompatible change of a descriptor struct, a workspace layout or an operand element format; a host
 * compiled against another value must not call into the library (frido_abi_version() returns the library's).
 * 1: rounds 1-2.  2: FridoGemm.sk_mode / gn_part inserted mid-struct, FridoAttnSmall / FridoSoftmax / FridoGnStats grew, the
 * split-K workspace starts with a 64-KiB ticket header that the CALLER zeroes once (frido_gemm_workspace_bytes), two-plane
 * operands became fp16 pairs (frido_x3_plane_format() == 1).  3 (r04): FridoGemm grew at its END (out_u8 / ldu8 / u8_mode, the fused
 * GroupNorm-apply input gn_*), two-plane operand producers saturate at +-65504.  4 (r04): FridoGemm.sk_mode 2 + FridoGnApply.sk_* at the
 * struct's END (a split-K GEMM's reduction finished by the GroupNorm launch that consumes its output).  5 (r05): FridoAttnSmall.skip_act_store
 * at the struct's END; frido_status_flags(&word, clear) (sticky saturation / non-finite flags; clear = 1 resets them). */
#define FRIDO_ABI_VERSION 6
#define FRIDO_SPLITK_HEADER_BYTES 65536     /* the ticket header at the start of a split-K workspace; partial sums follow: [splitk][M][N] f32 */

#define FRIDO_OK 0
#define FRIDO_EINVAL (-1)
#define FRIDO_EHIP (-2)
#define FRIDO_EUNSUPPORTED (-3)

enum { FRIDO_ACT_NONE = 0, FRIDO_ACT_RELU = 1, FRIDO_ACT_SILU = 2, FRIDO_ACT_GELU = 3 /* exact erf GELU */,
       FRIDO_ACT_QUICKGELU = 4 /* x * sigmoid(1.702 x): the MLP activation of OpenAI CLIP's text tower */ };

/* ------------------------------------------------------------------------------------------
 * frido_gemm — implicit-GEMM on the MFMA pipe:  for z < batch:
 *     C[z][m][n] = act(alpha * sum_k A[z][m][k] * B[z][n][k] + bias[n] + rowvec[r(m)][n]) + residual[z][m][n]
 * Dense mode (conv == 0): A is an operand matrix [M][K] (lda).  Replaces nn.Linear / 1x1 conv /
 *   einsum call sites: frido/modules/attention.py:161-192 (to_q/k/v/out, QK^T, PV), :47-64 (FF),
 *   :265-280 (proj_in/out); pyunet.py:248 (skip_connection), :225-231 (emb_layers), :560-565
 *   (time_embed); taming/modules/diffusionmodules/model.py:148-192 (AttnBlock q/k/v/proj, bmm);
 *   taming/models/msvqgan.py:75 (post_quant_conv).
 * Conv mode (conv == 1): A is an NHWC operand image [Bimg][Hs][Ws][Cin]; row m = (b, oy, ox),
 *   k = (ky*kw + kx)*Cin + c; logical input pixel (oy*stride+ky-pad, ox*stride+kx-pad) inside
 *   [0,Hl)x[0,Wl) maps to source pixel ((iy >> up_shift) << dn_shift, ...), zero outside.
 *   Replaces F.conv2d call sites: pyunet.py:208-239 (ResBlock 3x3), :110,119-121 (Upsample =
 *   up_shift 1), :152-156 (Downsample, stride 2 pad 1), :575-600,797-803 (heads);
 *   spade_norm.py:52-58 (SPADE convs on the nearest-resized map = dn_shift);
 *   taming/.../model.py:38-75 (Up/Downsample, asymmetric pad = pad 0 + bounds), :85-112.
 * K must be a multiple of 32 (operands are zero-padded by their producers); M, N arbitrary.
 */
typedef struct FridoGemm {
    int32_t M, N, K, batch;
    int32_t nsplit;             /* 1 = bf16, 2 = bf16x3 */
    int32_t conv;               /* 0 dense, 1 conv */
    const frido_bf16* A; int64_t a_lo; int64_t a_bs; int32_t lda;
    int32_t Hs, Ws, Cin;        /* conv: source image dims, Cin % 32 == 0 */
    int32_t Hl, Wl;             /* conv: logical (resized) input dims */
    int32_t Ho, Wo;             /* conv: output dims; M = Bimg*Ho*Wo */
    int32_t kh, kw, stride, pad, up_shift, dn_shift;
    /* asymmetric padding and output-row interleave, used by the 2x2 PHASE convolutions that replace "nearest x2 upsample ->
       conv3x3" (pyunet.py:110-121, taming model.py:49-53): output pixel (2y+a, 2x+b) only sees a 2x2 source window with
       summed taps, so four K = 4 Cin GEMMs do 4/9 of the work of one K = 9 Cin GEMM on the upsampled plane.
       padx: left padding (pad is the top padding); up2_phase = 0: rows are written in order; 1 + 2a + b: row m = (img, y, x)
       of this GEMM goes to row (img, 2y + a, 2x + b) of a [Bimg][2 Ho][2 Wo] output (Ho, Wo powers of two); 5: all four phases in one launch -- batch = 4, phase = batch
       index, B holds the four phase weight matrices b_bs apart, pad / padx are ignored (1 - a, 1 - b). */
    int32_t padx, up2_phase;
    /* optional second A operand appended along K: k in [K, K + K2) reads the dense matrix A2[m][k - K] (lda2).  Used to
       fold the ResBlock's 1x1 skip conv (pyunet.py:248,300; taming model.py:131-135) into the second 3x3 conv. */
    const frido_bf16* A2; int64_t a2_lo; int32_t lda2, K2;
    const frido_bf16* B; int64_t b_lo; int64_t b_bs; int32_t ldb;
    float alpha;
    const float* bias;          /* [N] per output column */
    const float* row_bias;      /* [M] per output row (transposed projections) */
    const float* rowvec; int32_t rows_per_vec; int32_t ldv;   /* row index m / rows_per_vec + *rowvec_step */
    const int32_t* rowvec_step;
    int32_t act;
    int32_t geglu;              /* 1: weight rows are interleaved in 16-row blocks [a | gate]; the epilogue writes
                                   (a) * gelu_erf(gate) as an operand [M][N/2] (attention.py:42-44), no f32 output */
    const float* residual; int64_t res_bs; int32_t ldr;
    float* out_f32; int64_t of_bs; int32_t ldo;
    int32_t res_bf16, out_bf16; /* 1: `residual` / `out_f32` point at bf16 activations (bf16 residual stream) */
    frido_bf16* out_op; int64_t oo_lo; int64_t oo_bs; int32_t ldoo;
    int32_t batch_inner;        /* > 1: blockIdx.y = zo * batch_inner + zi (e.g. batch x heads); the *_bs strides apply
                                   to zo and the *_bs2 strides to zi */
    int64_t a_bs2, b_bs2, of_bs2, oo_bs2;
    int32_t splitk;             /* > 1: K is split over gridDim.z; partial sums go to the workspace `ws` and are added in slice
                                   order 0 .. splitk-1 (bit-identical run to run), batch == 1.  Size: frido_gemm_workspace_bytes. */
    float* ws;                  /* [64 KiB of arrival tickets -- ZERO before the first launch that uses the buffer, left zero by
                                   every launch][partial sums] */
    int32_t sk_mode;            /* who adds the partial sums: 0 = a second kernel (splitk_reduce: [splitk][M][N] partials);
                                   1 = the LAST workgroup of each output tile to arrive (ticket per tile, agent-scope release /
                                   acquire around it; partials stored fragment-major per tile), which then runs the GEMM's own
                                   epilogue -- no second launch, and gn_part stays available.  Ring kernels only (the bf16
                                   patch-staged 3x3 kernel always uses 0); at most 16384 output tiles.
                                   2 (r04) = NOBODY in this launch: the [splitk][M][N] partials stay in `ws` and the epilogue
                                   (alpha, bias, rowvec, residual, out_f32) is the CONSUMER's job -- a frido_gn_fused launch with
                                   FridoGnApply.sk_* set, issued before anything else touches the workspace.  Ring kernels only;
                                   f32 output [M][N] (ldo == N), no activation / row bias / GEGLU / operand output. */
    float* gn_part;             /* optional (bf16x3 f32-stream outputs): per-channel partial {sum, sum of squares} of the STORED values
                                   over every 32-row block, gn_part[((m / 32) * N + n) * 2 + {0, 1}], written by the store-from-
                                   registers epilogue; frido_gn_stats (p1 / p2) turns them into GroupNorm statistics without
                                   re-reading the tensor (pyunet.py:262-300: every GroupNorm input is a conv / linear output).
                                   Needs: nsplit 2, f32 output only, no activation / row bias / GEGLU / two-kernel split-K /
                                   batching / upsample phases, N % 8 == 0, M % 32 == 0 (rejected otherwise) */
    int32_t tile;               /* 0 auto, 1 = 128x128, 2 = 128x192, 3 = 64x64, 4 = 128x64, 5 = 64x192, 6 = 64x128 (BK 32);
                                   7 = 256x128 (8 waves), 8 = 256x256 (8 waves, bf16 mode);
                                   11..16 = the same shapes with BK 64 (bf16 mode, K and Cin multiples of 64), 17 = 256x128 with BK 64;
                                   18 = 128x192 on eight waves, 19 = 256x192 on eight waves (bf16x3 mode);
                                   20 = 256x192 / 21 = 128x192 with the GroupNorm-apply fused in (gn_* below, bf16x3 mode);
                                   40 (r06) = fused GroupNorm [+ SiLU] + 3x3 conv with N = 3 or 4 output channels on the f32 VALU (gn_* + w_f32
                                   below; no MFMA, no operand planes: the denoiser's eps head);
                                   31 / 33 / 34 / 35 / 36 (r06) = tiles 1 / 3 / 4 / 5 / 6 with K split over the TWO wave groups of one 8-wave
                                   workgroup (partial tiles added through LDS, acc0 + acc1): dense bf16x3 GEMMs with an even number of
                                   32-deep k-tiles and splitk <= 1 -- for launches of fewer workgroups than the chip has CU slots */
    int32_t flags;              /* A/B switches (0 = defaults): bit 0 = do not stage the bf16 residual tile through LDS in the
                                   epilogue, bit 1 = do not hoist a launch-wide timestep vector into the bias, bit 4 = do not take the
                                   streamlined epilogues (bit 5 / 6: only the split-K / GEGLU one); TIMING EXPERIMENTS ONLY
                                   (results are garbage): bit 2 = skip the whole epilogue, bit 3 = skip only its stores;
                                   bits 8..15 / 16..23 / 24..25 (r05 experiment, honoured only by -DFRIDO_STAGGER_RT=1 builds of igemm.hip,
                                   results unchanged): start delay in quarter microseconds of the workgroups with dispatch ids 256..511 of a
                                   two-per-CU tile / smallest grid it applies to in units of 64 workgroups (0 = 768) / which workgroups wait; bit 26: the
                                   one-workgroup-per-CU kernels too (odd XCDs of the first 256 workgroups wait);
                                   bit 27 (r06): the ring kernel walks its output tiles in COLUMN PANELS (P = 8 columns x all rows, P halved while the weight panel
                                   exceeds 3 MB) instead of row-major, so that an XCD's concurrent workgroups share an L2-resident weight panel;
                                   a pure re-ordering of independent tiles, results unchanged */
    /* optional uint8 image output (r04: the output path of scripts/sample_diffusion.py fused into the decoder's last conv --
       the all-gather and the NPZ / PNG writers then move uint8): out_u8[row * ldu8 + n] for n < N, NHWC.  u8_mode 1 =
       custom_to_np (sample_diffusion.py:115-121): ((x + 1) * 127.5) clamped to [0, 255], truncated; 2 = custom_to_pil
       (:103-113): x clamped to [-1, 1], (x + 1) / 2, times 255, truncated -- every step a separate fp32 rounding, like the
       torch / numpy expressions.  x is the value the f32 output would hold; out_f32 / out_op may be null.  Element-wise epilogue
       (no split-K, no GroupNorm partial sums). */
    uint8_t* out_u8; int32_t ldu8, u8_mode;
    /* optional FUSED GroupNorm-apply input (r04; tiles 20 / 21, conv mode 3x3 stride 1 pad 1 on 16..64-pixel-wide planes, nsplit 2):
       the A operand is not read from memory (A may be null) but produced inside the kernel from the f32 residual stream --
           a = act( (x - mean) * rstd * gn_weight[c] + gn_bias[c]  [* (1 + gn_gamma) + gn_beta] )
       i.e. frido_gn_apply (pyunet.py:262-300 in_layers / out_layers; spade_norm.py:44-60) folded into the convolution that
       consumes it: x = the virtual channel concat of gn_x1 [rows][gn_C1] and gn_x2 [rows][gn_C2] (gn_C1 + gn_C2 == Cin,
       multiples of 32), gn_partials = frido_gn_stats' output [Bimg][gn_nsplit_px][gn_groups][2] doubles, gn_gamma / gn_beta
       optional f32 maps [rows][Cin], gn_act = FRIDO_ACT_NONE / FRIDO_ACT_SILU.  Per-element arithmetic is frido_gn_apply's.
       With K2 > 0 the appended K range reads the RAW f32 rows raw_x1 [rows][raw_C1] | raw_x2 [rows][raw_C2]
       (raw_C1 + raw_C2 == K2), split into hi / lo planes in the kernel, instead of the operand A2 (the fused 1x1 skip conv).
       M and H*W multiples of the tile's 256 / 128 rows, N a multiple of 192, no split-K. */
    const float* gn_x1; const float* gn_x2; int32_t gn_C1, gn_C2;
    const double* gn_partials; int32_t gn_nsplit_px, gn_groups; float gn_eps;
    const float* gn_weight; const float* gn_bias; const float* gn_gamma; const float* gn_beta;
    int32_t gn_act;
    const float* raw_x1; const float* raw_x2; int32_t raw_C1, raw_C2;
    /* (r06, tile 40) f32 weights of the fused GroupNorm + 3x3 conv with a TINY output width (N <= 4: the denoiser's output head,
       pyunet.py:775-803): [Cin / 32][9 taps][N][32 channels] floats; B is unused.  Grown at the struct's END (FridoOp union: 512 B). */
    const float* w_f32;
} FridoGemm;

/* GroupNorm statistics (32 groups, biased variance, fp32) over a virtual channel concat of two
 * NHWC f32 tensors: nn.GroupNorm call sites util.py:214-216 (eps 1e-5), attention.py:76-77 and
 * taming/.../model.py:34-35 (eps 1e-6); the concat is pyunet.py:939.  Each (b, pixel-split) workgroup writes
 * per-group partial {sum, sumsq} doubles to partials[b][s][g][2]; frido_gn_apply combines them in a fixed order.
 * (A last-arriver finalise inside this kernel was measured: the agent-scope release fence per workgroup costs 3x
 * what it saves — MI355X_MICROARCH.md price list, `buffer_wbl2` with dirty L2 — so the combine lives in apply.) */
typedef struct FridoGnStats {
    const float* x1; int32_t C1; const float* x2; int32_t C2;
    int32_t B, HW, groups, nsplit_px;
    double* partials;
    int32_t x_bf16;             /* 1: x1 / x2 are bf16 */
    const float* p1; const float* p2;   /* both non-null (p2 only when C2 > 0): do not read x1 / x2 -- sum the producers' per-channel
                                   partial sums instead (FridoGemm.gn_part: [B * HW / 32][C1 or C2][2]); nsplit_px must be 1 */
} FridoGnStats;

/* GroupNorm apply (+ SPADE modulation + SiLU) -> operand tensor.
 *   y = (x - mean) * rstd * weight[c] + bias[c];  if gamma: y = y * (1 + gamma) + beta  (spade_norm.py:60)
 *   if act == SILU: y = y * sigmoid(y)   (pyunet.py:210,234; taming model.py:28-31)
 * Optionally also writes the un-normalised x as an operand (`raw_op`, input of a 1x1 skip conv,
 * pyunet.py:248,300) and/or y as f32 (`out_f32`). */
typedef struct FridoGnApply {
    const float* x1; int32_t C1; const float* x2; int32_t C2;
    int32_t B, HW, groups, nsplit_px;
    const double* partials; float eps;
    const float* weight; const float* bias;
    const float* gamma; const float* beta;
    int32_t act; int32_t nsplit;
    frido_bf16* out_op; int64_t out_lo;
    frido_bf16* raw_op; int64_t raw_lo;
    float* out_f32;
    int32_t x_bf16;             /* 1: x1 / x2 are bf16 */
    int32_t gb_bf16;            /* 1: gamma / beta are bf16 */
    /* optional (r04, ABI 4), frido_gn_fused with f32 input only: x1 is not READ but produced here, from the raw split-K partial sums
       of the GEMM that would have written it (FridoGemm.sk_mode 2 leaves them in its workspace and launches no reduction):
         x1[m][c] = sk_alpha * sum_z sk_ws[z][m][c] (z = 0 .. sk_n - 1 in order) + sk_bias[c] + sk_rowvec[m / sk_rows_per_vec + *sk_rowvec_step][c]
                    + sk_residual[m][c]
       -- splitk_reduce's arithmetic bit for bit -- with m = b * HW + pixel, rows C1 wide.  The value is also stored to sk_out[m][c]
       (row stride C1) unless sk_out is NULL (nobody else reads the tensor).  sk_ws = workspace + FRIDO_SPLITK_HEADER_BYTES.
       One launch and one round trip of the tensor less per small-plane convolution (pyunet.py:262-300: conv -> GroupNorm). */
    const float* sk_ws; int32_t sk_n; float sk_alpha;
    const float* sk_bias; const float* sk_rowvec; const int32_t* sk_rowvec_step; int32_t sk_rows_per_vec, sk_ldv;
    const float* sk_residual; int32_t sk_ldr;
    float* sk_out;
} FridoGnApply;

/* LayerNorm over the last dim (eps 1e-5, affine): attention.py:203-205 -> operand tensor. */
typedef struct FridoLayerNorm {
    const float* x; int32_t rows, C; float eps;
    const float* weight; const float* bias;
    int32_t nsplit; frido_bf16* out_op; int64_t out_lo;
    int32_t x_bf16;
    float* out_f32;             /* optional f32 copy of the result (final norm of the cond-stage encoder) */
} FridoLayerNorm;

/* Row softmax (attention.py:188, taming model.py:181): x[rows][N] f32 (ld) -> operand [rows][Npad]
 * with columns >= N written as zero. */
typedef struct FridoSoftmax {
    const float* x; int32_t rows, N, ld, Npad;
    int32_t nsplit; frido_bf16* out_op; int64_t out_lo;
    int32_t causal_nq;          /* > 0: causal mask -- row r only sees keys 0 .. (r mod causal_nq) (CLIP text tower), the rest
                                   of the row is written as zero */
} FridoSoftmax;

/* Fused attention core for SHORT key sequences (Nk <= 128): O = softmax(alpha * Q K^T) V in one launch, the score
 * matrix never leaving the chip.  Covers every cross-attention of the U-Net (Nk = context tokens: 26, 92 or 1) and its
 * self-attention on the 8x8 plane (attention.py:170-193); longer sequences take the GEMM -> softmax -> GEMM path.
 * Q rows [B*Nq] (row stride ldq), K rows [B][Nk] (per-sample stride k_bs elements, row stride ldk), VT = V transposed
 * [B][dv][ldvt] with zero columns beyond Nk (the layout PV's B operand has on the three-kernel path), O operand rows
 * [B*Nq] (ldo).  Nq % 16 == 0, d % 32 == 0, dv % 16 == 0, ldvt = Nk rounded up to 32.
 * With `out_act` set the result leaves as the residual stream instead of an operand:
 * out_act[row][c] = O[row][c] + bias[c] + residual[row][c] (f32, or bf16 when act_bf16) -- the form used when the
 * attention output projection has been folded into V (single head: W_o (P V) = P (V W_o^T)).  With BOTH out_act and out_op set
 * (short-key kernel only) the stream values are additionally written as an operand (hi / lo planes, ldo). */
typedef struct FridoAttnSmall {
    const frido_bf16* Q; int64_t q_lo; int32_t ldq;
    const frido_bf16* K; int64_t k_lo; int64_t k_bs; int32_t ldk;
    const frido_bf16* VT; int64_t vt_lo; int64_t vt_bs; int32_t ldvt;
    frido_bf16* out_op; int64_t out_lo; int32_t ldo;
    int32_t B, Nq, Nk, d, dv, nsplit; float alpha;
    void* out_act; const void* residual; const float* bias; int32_t ld_act, ldr, act_bf16;
    /* optional (r03), with the f32 residual-stream output (out_act, act_bf16 = 0) in bf16x3 mode: the LayerNorm of the row just
       written, (x - mean) * rstd * ln_w + ln_b over the dv channels, as a hi / lo operand [B * Nq][ld_ln] -- the norm2 / norm3 of a
       transformer block (attention.py:222-227) without a launch of their own and without re-reading the stream from HBM.  The
       workgroup must own whole rows: frido_attn_flash with d = 256 or 384; frido_attn_small with B * Nq / 16 >= 256 (rejected otherwise). */
    frido_bf16* ln_op; int64_t ln_lo; int32_t ld_ln;
    const float* ln_w; const float* ln_b; float ln_eps;
    /* (r05, frido_attn_small only) 1: the f32 stream rows are NOT stored -- out_act stays non-null and selects the stream form
       (O + bias + residual), whose values leave only as the operand copy (out_op) and / or the fused LayerNorm (ln_op).  The
       transformer block's h3 = attn2(norm2(h2)) + h2 (attention.py:226) is read by nothing else once FF2 + proj_out are one GEMM. */
    int32_t skip_act_store;
} FridoAttnSmall;

/* GEGLU gate (attention.py:42-44): x[rows][2H] f32 -> operand [rows][H] = x[:, :H] * gelu_erf(x[:, H:]). */
typedef struct FridoGeglu {
    const float* x; int32_t rows, H;
    int32_t nsplit; frido_bf16* out_op; int64_t out_lo;
} FridoGeglu;

/* f32 -> operand conversion with channel slice and zero padding.  src is NCHW (nchw = 1, channel
 * stride HW) or NHWC/row-major (nchw = 0) with Csrc channels; channels [c0, c0+Cuse) go to operand
 * columns [0, Cuse), columns up to Cpad are zero.  `scale` multiplies the value (1.0 = exact copy). */
typedef struct FridoPack {
    const float* src; int32_t B, HW, Csrc, c0, Cuse, Cpad, nchw;
    float scale;
    int32_t nsplit; frido_bf16* out_op; int64_t out_lo;
} FridoPack;

/* f32 layout change: NHWC [B][HW][Csrc] columns [c0, c0+Cuse) -> NCHW dst[B][Cdst][HW] at channel d0,
 * or the reverse direction (to_nchw = 0: NCHW src -> NHWC dst), or an NHWC -> NHWC column-block copy (to_nchw = 2). */
typedef struct FridoRelayout {
    const float* src; float* dst;
    int32_t B, HW, Csrc, c0, Cuse, Cdst, d0, to_nchw;
} FridoRelayout;

/* VectorQuantizer2 lookup (taming/modules/vqvae/quantize.py:272-294) fused with the per-scale
 * 1/scale_factor of decode_first_stage (frido.py:832-838): z = x[..., c0:c0+e] * inv_scale;
 * idx = argmin_j (|z|^2 + |e_j|^2 - 2 z.e_j) (lowest index on ties); writes z + (e_idx - z) into
 * zq[..., q0:q0+e] (NHWC f32, Cq channels) and idx (int64). */
typedef struct FridoVq {
    const float* x; int32_t npix, Cx, c0, e;
    float inv_scale;
    const float* codebook; int32_t n_codes;
    float* zq; int32_t Cq, q0;
    int64_t* idx;
    const int64_t* force_idx;   /* test hook: use these code indices instead of the argmin (the decoder is then compared on
                                   the reference's own codes, independent of VQ decision boundaries) */
} FridoVq;

/* One DDIM / PLMS state update (ddim.py:232-273, plms.py:247-303) on the NHWC f32 latent state
 * x[B][HW][Cx] for the active stage channels [start, start+nch):
 *   e      = CFG mix e_u + s (e_c - e_u) if eps_uncond else eps_cond         (ddim.py:211-226)
 *   PLMS:  e = sum_k ab[k] * hist[k] (Adams-Bashforth coefficients from coef row) when hist != null
 *   x0     = (x - sqrt(1-a_t) e) / sqrt(a_t);  x' = sqrt(a_prev) x0 + sqrt(1-a_prev-sigma^2) e + sigma*noise*T
 * Coefficients are read from coef[*step][8] = {a_t, a_prev, sigma, sqrt(1-a_t), ab0..ab3};
 * noise is either a tape (noise[*step * noise_stride + ...], NHWC [B][HW][nch_noise] with the
 * active channels at `noise_c0`) or Philox4x32-10 keyed by (seed, global sample index, *step).
 * Frozen channels [0,start) are passed through (x0 = x, x' = x0). */
typedef struct FridoSamplerStep {
    float* x; int32_t B, HW, Cx, start, nch;
    const float* eps_cond; const float* eps_uncond; float cfg_scale;   /* [B*HW][nch] */
    float* eps_out;              /* optional: the (CFG-mixed) eps, [B*HW][nch] (PLMS history slot) */
    const float* hist1; const float* hist2; const float* hist3;   /* PLMS older eps or null */
    const float* coef; const int32_t* step; int32_t coef_row_offset;
    const float* noise; int64_t noise_stride; int32_t noise_C, noise_c0;
    uint64_t seed; int64_t sample0; int32_t rng_stream;
    const int64_t* rng_dev;      /* optional device {seed, sample0}: overrides the two fields above (graph replay) */
    float temperature;
    float* x_out;                /* where x' goes (may alias x) */
    float* pred_x0;              /* optional [B][HW][Cx] */
    int32_t write_x;             /* 0: only eps_out/pred_x0 */
    /* graph-replay forms (one captured step body serves every step / guidance scale):
       cfg_dev: optional device float overriding cfg_scale;
       hist_ring: PLMS eps history as a 4-slot ring [4][hist_stride] indexed by the DEVICE step counter instead of the
       explicit hist1..3 / eps_out pointers.  hist_mode 1 (regular step, plms.py:175-177,285-301): the (CFG-mixed) eps of
       step i goes to slot i & 3 and is combined with the min(i, 3) older slots (i-1) & 3, ...; hist_mode 3 (second half of
       the Heun-style first step, plms.py:285-289): nothing is stored, eps is combined with slot i & 3. */
    const float* cfg_dev;
    float* hist_ring; int64_t hist_stride; int32_t hist_mode;
} FridoSamplerStep;

/* Stage hand-off (ddim.py:177-185): channels [c0,c1) of x[B][H][W][Cx] replaced by their
 * 2^levels x 2^levels block mean (avg_pool2d applied `levels` times, then nearest expand). */
typedef struct FridoHandoff {
    float* x; int32_t B, H, W, Cx, c0, c1, levels;
} FridoHandoff;

/* Gaussian fill: dst[i] ~ N(0,1), Philox keyed by (seed, sample index = sample0 + i / per_sample, stream). */
typedef struct FridoRandn {
    float* dst; int64_t n; int64_t per_sample; uint64_t seed; int64_t sample0; int32_t rng_stream;
} FridoRandn;

/* Sinusoidal timestep embedding (frido/modules/diffusionmodules/util.py:151-171):
 * out[i][0:half] = cos(t_i * f_k), out[i][half:2*half] = sin(t_i * f_k), f_k = exp(-ln(max_period) * k / half),
 * all in fp32 like the reference; t is int64 (DDPM indices). */
typedef struct FridoTimeEmb { const int64_t* t; int32_t n, dim; float max_period; float* out; } FridoTimeEmb;

/* ConvTranspose2d(k=4, stride=2, padding=1) on small-channel f32 NHWC maps (taming/models/msvqgan.py:81-83, the
 * coarse-to-fine `upsample` of the MS-VQGAN encoder): src [B][h][w][Cin] -> dst [B][2h][2w][Cout] (+bias);
 * weight in nn.ConvTranspose2d layout [Cin][Cout][4][4]. */
typedef struct FridoConvT { const float* src; float* dst; const float* weight; const float* bias;
                            int32_t B, h, w, Cin, Cout; } FridoConvT;

/* Nearest 2^up_shift up-sampling of channels [c0, c0+Cuse) of an NHWC f32 map [B][h][w][Csrc], scaled, written into
 * channels [d0, ...) of an NCHW f32 tensor [B][Cdst][h<<up][w<<up] (msvqgan.py:364-374 + frido.py:654-662). */
typedef struct FridoPlace { const float* src; float* dst; int32_t B, h, w, Csrc, c0, Cuse, Cdst, d0, up_shift;
                            float scale; } FridoPlace;

/* Token + absolute position embedding (frido/modules/x_transformer.py:25-36,620-622):
 * out[r][:] = tok[tokens[r]][:] + pos[r % n][:]   (f32, D % 4 == 0). */
typedef struct FridoEmbed { const int64_t* tokens; const float* tok; const float* pos; float* out;
                            int32_t rows, n, D, vocab; } FridoEmbed;      /* pos == NULL: plain row gather out[r] = tok[tokens[r]] */

/* Row L2 normalisation out[r] = x[r] / ||x[r]||_2 (FrozenCLIPTextEmbedder.forward, frido/modules/encoders/modules.py:213-214). */
typedef struct FridoL2Norm { const float* x; float* out; int32_t rows, C; } FridoL2Norm;

/* Output conversion of scripts/sample_diffusion.py:115-121 (custom_to_np): NHWC f32 in [-1, 1] -> uint8 NHWC,
 * ((x + 1) * 127.5) clamped to [0, 255] and truncated. */
typedef struct FridoToU8 { const float* src; uint8_t* dst; int64_t n; } FridoToU8;

/* step counter update: *step += delta (one thread). */
typedef struct FridoStepAdd { int32_t* step; int32_t delta; } FridoStepAdd;

/* device-to-device copy of n bytes (16-byte aligned, n % 16 == 0): a graph-capturable memcpy node for the sampler state
 * save / restore of the PLMS first step (plms.py:285-289 evaluates eps at x_prev and then steps from x again). */
typedef struct FridoCopy { const void* src; void* dst; int64_t n; } FridoCopy;

/* fill a device buffer with a 32-bit pattern. */
typedef struct FridoFill { uint32_t* dst; int64_t n; uint32_t value; } FridoFill;

/* Cross-stream ordering inside a program (native executor only): everything enqueued so far on stream `from` happens
 * before anything enqueued later on stream `to`.  Stream 0 is the caller's stream, stream 1 a side stream the library owns
 * per caller stream; an op runs on the stream named by FridoOp.stream.  Used to run the two independent projections of an
 * attention block (frido/modules/attention.py:175-177: to_q / to_v of the same input) concurrently; both directions are
 * captured into the hipGraph as parallel branches. */
typedef struct FridoSync { int32_t from, to; } FridoSync;

enum FridoOpKind {
    FRIDO_OP_GEMM = 1, FRIDO_OP_GN_STATS, FRIDO_OP_GN_APPLY, FRIDO_OP_LAYERNORM, FRIDO_OP_SOFTMAX,
    FRIDO_OP_GEGLU, FRIDO_OP_PACK, FRIDO_OP_RELAYOUT, FRIDO_OP_VQ, FRIDO_OP_SAMPLER_STEP,
    FRIDO_OP_HANDOFF, FRIDO_OP_RANDN, FRIDO_OP_STEP_ADD, FRIDO_OP_FILL, FRIDO_OP_TIME_EMB, FRIDO_OP_CONVT, FRIDO_OP_PLACE, FRIDO_OP_EMBED, FRIDO_OP_TO_U8, FRIDO_OP_ATTN_SMALL, FRIDO_OP_GN_FUSED, FRIDO_OP_COPY, FRIDO_OP_ATTN_FLASH, FRIDO_OP_SYNC, FRIDO_OP_L2NORM, FRIDO_OP__COUNT
};

/* A program is an array of tagged ops executed in order on one stream by the native executor. */
typedef struct FridoOp {
    int32_t kind; int32_t stream;   /* 0 = the caller's stream, 1 = the library's side stream (see FridoSync) */
    union {
        FridoGemm gemm; FridoGnStats gn_stats; FridoGnApply gn_apply; FridoLayerNorm layernorm;
        FridoSoftmax softmax; FridoGeglu geglu; FridoPack pack; FridoRelayout relayout; FridoVq vq;
        FridoSamplerStep sampler_step; FridoHandoff handoff; FridoRandn randn; FridoStepAdd step_add;
        FridoFill fill; FridoTimeEmb time_emb; FridoConvT convt; FridoPlace place; FridoEmbed embed; FridoToU8 to_u8; FridoAttnSmall attn_small; FridoCopy copy; FridoSync sync; FridoL2Norm l2norm;
        char _size[512];
    } u;
} FridoOp;

/* ---- single-op launchers ---- */
int frido_gemm(const FridoGemm* d, frido_stream_t s);
/* Bytes of the split-K workspace `ws` a descriptor needs (0 when splitk <= 1): the 64-KiB ticket header + splitk * M' * N'
 * floats (M', N' padded to whole tiles for sk_mode 1).  The caller owns the buffer and ZEROES it once when it allocates it
 * (one per stream is enough: launches on a stream are ordered); a host that is not the bundled Python runtime sizes it with
 * this call instead of reading frido_amd/tune.py. */
int64_t frido_gemm_workspace_bytes(const FridoGemm* d);
int frido_gn_stats(const FridoGnStats* d, frido_stream_t s);
int frido_gn_apply(const FridoGnApply* d, frido_stream_t s);
int frido_layernorm(const FridoLayerNorm* d, frido_stream_t s);
int frido_softmax(const FridoSoftmax* d, frido_stream_t s);
int frido_geglu(const FridoGeglu* d, frido_stream_t s);
int frido_pack(const FridoPack* d, frido_stream_t s);
int frido_relayout(const FridoRelayout* d, frido_stream_t s);
int frido_vq(const FridoVq* d, frido_stream_t s);
int frido_sampler_step(const FridoSamplerStep* d, frido_stream_t s);
int frido_handoff(const FridoHandoff* d, frido_stream_t s);
int frido_randn(const FridoRandn* d, frido_stream_t s);
int frido_step_add(const FridoStepAdd* d, frido_stream_t s);
int frido_fill(const FridoFill* d, frido_stream_t s);
int frido_time_emb(const FridoTimeEmb* d, frido_stream_t s);
int frido_convt(const FridoConvT* d, frido_stream_t s);
int frido_place(const FridoPlace* d, frido_stream_t s);
int frido_embed(const FridoEmbed* d, frido_stream_t s);
int frido_to_u8(const FridoToU8* d, frido_stream_t s);
int frido_attn_small(const FridoAttnSmall* d, frido_stream_t s);
int frido_copy(const FridoCopy* d, frido_stream_t s);
int frido_l2norm(const FridoL2Norm* d, frido_stream_t s);
/* Flash-style attention core for LONG key sequences on the same FridoAttnSmall descriptor (any Nk; d = dv in
 * {128, 256, 384, 512, 576}; Nq arbitrary): online softmax over 32-key tiles, the [Nq][Nk] score matrix is never formed.
 * Replaces the QK^T GEMM -> f32 scores -> softmax -> PV GEMM chain of frido/modules/attention.py:170-193 on the 32x32 /
 * 64x64 planes and of the VQGAN AttnBlock (taming/modules/diffusionmodules/model.py:168-192: 4096 keys at 256^2, 16384 at
 * 512^2).  frido_attn_flash_supported(d) tells whether a head dimension is instantiated. */
int frido_attn_flash(const FridoAttnSmall* d, frido_stream_t s);
int frido_attn_flash_supported(int32_t d);
/* (r05) head widths for which a two-plane frido_attn_flash launch may carry FridoAttnSmall.ln_op (its workgroups own whole rows):
 * 256 and 384 always, 512 on the d-split 8-wave form (the default; FRIDO_FLASH_DSPLIT=0 selects the 4-wave form). */
int frido_attn_flash_ln_supported(int32_t d);
/* One-launch GroupNorm (statistics + apply) on a FridoGnApply descriptor whose `partials` is unused; bf16 stream only.
 * frido_gn_fused_chunk returns the channel-chunk width it would use (and the workgroup size), or 0 if the descriptor does
 * not qualify -- then gn_stats + gn_apply is the path. */
int frido_gn_fused(const FridoGnApply* d, frido_stream_t s);
int frido_gn_fused_chunk(const FridoGnApply* d, int* nthreads);

/* ---- native executor: run / capture a whole program ---- */
int frido_run(const FridoOp* ops, int32_t n, frido_stream_t s);
/* Same, with a HIP event recorded on `s` around every op; ms[i] = device time of op i (synchronises at the end). */
int frido_run_timed(const FridoOp* ops, int32_t n, frido_stream_t s, float* ms);
/* Capture `ops` into a hipGraph on stream `s` (which must be a non-default stream) and
 * instantiate it.  Returns an opaque handle in *out. */
int frido_graph_capture(const FridoOp* ops, int32_t n, frido_stream_t s, void** out);
int frido_graph_launch(void* graph, frido_stream_t s);
int frido_graph_destroy(void* graph);

/* ---- timing on the launch stream (HIP events) ---- */
int frido_event_create(void** ev);
int frido_event_record(void* ev, frido_stream_t s);
int frido_event_elapsed_ms(void* start, void* stop, float* ms);   /* synchronises on `stop` */
int frido_event_destroy(void* ev);

/* one-time initialisation (kernel attributes); safe to call repeatedly, must precede graph capture */
int frido_init(void);

/* ---- introspection ---- */
int frido_abi_version(void);
/* Element format of the planes of a TWO-plane (nsplit = 2, "bf16x3" precision) operand: 0 = bf16 hi + bf16 lo (rounds 1-2),
 * 1 = fp16 hi + fp16 lo (r03 default: 22 mantissa bits at the same bytes and MFMA passes).  One-plane operands are always bf16.
 * A host that packs weights / operands itself (hi = round(v), lo = round(v - hi) in this format) asks here. */
int frido_x3_plane_format(void);
int frido_sizeof_op(void);             /* sizeof(FridoOp): checked by the ctypes mirror */
int frido_sizeof_desc(int32_t kind);   /* sizeof of the descriptor struct of that op kind */
const char* frido_last_error(void);
/* Sticky numerics status of the CURRENT device (r05): kernels cannot return errors, so operand producers and normalisation kernels
 * OR bits into a device word when they meet a value the arithmetic cannot represent, and this call reads it back (it synchronises
 * the device: call it after a sampling pass, not inside a captured region).  clear != 0 resets the word after reading.
 *   FRIDO_STATUS_SATURATED: a two-plane fp16 operand (frido_x3_plane_format() == 1) was clamped at +-65504 -- the fp32-class error
 *     bound of the "bf16x3" arithmetic no longer holds for that tensor; run the model on the bf16-pair build of the library.
 *   FRIDO_STATUS_NONFINITE: a GroupNorm / LayerNorm / softmax statistic was NaN or infinite (a NaN / inf reached the stream).
 * The reference has no counterpart (torch propagates NaN / inf through F.conv2d etc.); this is how they stay visible here. */
#define FRIDO_STATUS_SATURATED 1u
#define FRIDO_STATUS_NONFINITE 2u
int frido_status_flags(uint32_t* flags, int32_t clear);
/* (r06, ABI 6) The same word in STREAM ORDER: one small kernel on `stream` ORs (clear != 0: and resets) the per-file words, the result
 * comes back through a pinned host word after ONE hipStreamSynchronize(stream) -- no device drain, no blocking symbol copies.  This is
 * what the host side calls after a sampling pass / a decode to decide whether the model must move to the bf16-pair planes
 * (frido_amd/models.py auto plane selection); not capturable (it synchronises the stream). */
int frido_status_poll(frido_stream_t stream, uint32_t* flags, int32_t clear);
/* diagnostic: the status word of ONE source file of the library (index = link order: igemm, convgn, norm, misc, attn, flash, runtime);
 * returns -1 past the last one.  tools/find_saturation.py uses it to name the kernel family that raised a bit. */
int frido_status_word_of(int32_t idx, uint32_t* word);
int frido_device_info(int32_t* cu_count, int32_t* gcn_arch_is_gfx950, int64_t* hbm_bytes);

#ifdef __cplusplus
}
#endif
#endif /* FRIDO_HIP_H */
