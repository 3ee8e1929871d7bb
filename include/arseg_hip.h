/*
 * arseg_hip.h -- C ABI of libarseg_hip.so: the MI355X (gfx950) kernels of AR-Seg's LR-branch
 * inference hot path.  This is the drop-in boundary: plain pointers and
 * sizes, no torch / C++ types.  Each entry point cites the reference interface it replaces
 * (paths relative to the AR-Seg repository).
 *
 * Conventions
 *   - All data pointers are DEVICE pointers unless the name ends in `_host`.
 *   - The caller owns every buffer (inputs, outputs, workspace).  Nothing is allocated here.
 *   - Every function only ENQUEUES work on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream) and never synchronises the device.
 *   - Return value: 0 = ok; < 0 = ARSEG_E* (argument / shape problem, nothing was launched);
 *     > 0 = a hipError_t from the launch.  Nothing throws or aborts.
 *   - "NHWC" tensors are [N][H][W][ld] floats with `ld >= C` the per-pixel channel stride, so a
 *     channel slice of a wider tensor can be read or written in place (concat without copies).
 *   - No hidden global state; the library is thread-compatible (one stream per caller thread).
 */
#ifndef ARSEG_HIP_H
#define ARSEG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARSEG_ABI_VERSION 5

enum arseg_status {
    ARSEG_OK = 0,
    ARSEG_EINVAL = -1,        /* null pointer, non-positive size, misaligned pointer / stride */
    ARSEG_EUNSUPPORTED = -2,  /* shape outside what the kernels are built for (see each function) */
    ARSEG_EWORKSPACE = -3     /* workspace too small; query the *_workspace_bytes function */
};

enum arseg_act { ARSEG_ACT_NONE = 0, ARSEG_ACT_RELU = 1, ARSEG_ACT_PRELU = 2, ARSEG_ACT_SIGMOID = 3 };
enum arseg_layout { ARSEG_NCHW = 0, ARSEG_NHWC = 1, ARSEG_C8 = 2 /* [N][C/8][H][W][8], the CReFF kernel's layout */ };
enum arseg_flow_dtype { ARSEG_FLOW_F32 = 0, ARSEG_FLOW_F64 = 1 };
enum arseg_resize_mode { ARSEG_NEAREST = 0, ARSEG_BILINEAR = 1 };
enum arseg_reduce_op { ARSEG_REDUCE_MEAN = 0, ARSEG_REDUCE_MAX = 1 };
enum arseg_dtype { ARSEG_DT_F32 = 0, ARSEG_DT_F16 = 1, ARSEG_DT_BF16 = 2 };   /* storage element type of the 16-bit entry points */
enum arseg_creff_warp_impl { ARSEG_CREFF_WARP_AUTO = 0, ARSEG_CREFF_WARP_TILES = 1 /* creff_rr.hip */, ARSEG_CREFF_WARP_ROLL = 2 /* creff_roll.hip */ };
enum arseg_creff_impl { ARSEG_CREFF_AUTO = 0, ARSEG_CREFF_MFMA = 1 /* split-fp16 matrix-core kernel */, ARSEG_CREFF_VALU = 2 /* fp32 VALU kernel */ };

typedef void *arseg_stream_t; /* hipStream_t */

int arseg_version(void);
const char *arseg_status_string(int status);

/* ---------------------------------------------------------------------------------------------
 * The `localAttention` pair (third-party CUDA extension the reference imports at
 * model/attention.py:7-11; call sites model/attention.py:18 and :38).
 *   similar  : s[n,y,x,dy*kW+dx] = sum_c q[n,c,y,x] * k[n,c,y+dy-kH/2,x+dx-kW/2]   (0 outside)
 *   weighting: o[n,c,y,x]        = sum_i v[n,c,y+dy_i-kH/2,x+dx_i-kW/2] * w[n,y,x,i] (0 outside)
 * q,k,v,o: NCHW contiguous fp32; s,w: [N,H,W,kH*kW].  kH,kW odd, kH*kW <= 121.
 * ------------------------------------------------------------------------------------------- */
int arseg_local_similar_fwd(const float *q, const float *k, float *s, int N, int C, int H, int W, int kH, int kW,
                            arseg_stream_t stream);
int arseg_local_weighting_fwd(const float *v, const float *w, float *o, int N, int C, int H, int W, int kH, int kW,
                              arseg_stream_t stream);
/* The same pair on NHWC (torch.channels_last) features: element (n,c,y,x) at ((n*H + y)*W + x)*ld + c, ld >= C shared by the two inputs
 * (similar) / by v and o (weighting); s, w keep [N,H,W,kH*kW].  No layout change between the NHWC backbone tensors and the op. */
int arseg_local_similar_nhwc_fwd(const float *q, const float *k, int ld, float *s, int N, int C, int H, int W, int kH, int kW,
                                 arseg_stream_t stream);
int arseg_local_weighting_nhwc_fwd(const float *v, const float *w, int ld, float *o, int N, int C, int H, int W, int kH, int kW,
                                   arseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * warpFeature(feature, flow)                                            evaluation.py:61-87
 * feature: [N,C,H,W] in `layout` (NCHW or NHWC with ld == C); out: same layout, or C8 when
 * out_layout == ARSEG_C8 (NHWC input only); flow: [N,H,W,2] (dx,dy) in feature pixels, fp32 or fp64.  Sampling = grid_sample(bilinear, zeros, align_corners=False) of the grid
 * normalised with 2*g/(W-1)-1, reproduced operation by operation (fp64 grid -> fp32 cast).
 * NHWC requires C % 4 == 0.
 * ------------------------------------------------------------------------------------------- */
int arseg_warp_fwd(const float *feature, const void *flow, int flow_dtype, float *out, int N, int C, int H, int W,
                   int layout, int out_layout, arseg_stream_t stream);

/* Motion-vector resize block                                            evaluation.py:176-180
 * mv_q: int16 quarter-pel [N,H,W,2] exactly as stored on disk (dataset/camvid.py:624-626,
 * dataset/cityscapes.py:282-285); out: fp64 [N,Hp,Wp,2] = bilinear(align_corners=True) of
 * (mv_q/4) * Hp/H (both components scaled by Hp/H as the reference does). */
int arseg_mv_resize_fwd(const int16_t *mv_q, double *out, int N, int H, int W, int Hp, int Wp, arseg_stream_t stream);

/* The same block for a float flow field [N,H,W,2] in pixels (fp32 or fp64; the reference's DataLoader hands over fp64 = int16/4):
 * out fp64 [N,Hp,Wp,2] = bilinear(align_corners=True) of flow * Hp/H, computed in fp64. */
int arseg_flow_resize_fwd(const void *flow, int flow_dtype, double *out, int N, int H, int W, int Hp, int Wp, arseg_stream_t stream);

/* The two steps above fused (fast path): warp an NHWC feature straight from the int16 MV map;
 * out_layout = ARSEG_NHWC or ARSEG_C8. */
int arseg_warp_mvq_fwd(const float *feature, const int16_t *mv_q, float *out, int N, int C, int Hp, int Wp, int H,
                       int W, int out_layout, arseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * CReFF: MyAttention.forward(hr_feat, lr_feat) [+ final 1x1 classifier]
 *                                   model/attention.py:184-213; model/pspnet.py:219-231;
 *                                   model/bisenet.py:565-575
 * One fused kernel: bilinear(align_corners=True) upsample of lr, the three depthwise 3x3 convs
 * (+bias), the kH x kW local QK^T with zero-score padding taps, softmax over all kH*kW taps,
 * PV, residual add; optionally the 1x1 classifier (+ log-softmax over classes) on the result.
 *   hr   : C8 [N,C/8,Hp,Wp,8]  (already warped; arseg_warp*_fwd can write it directly)
 *   lr   : NHWC [N,hp,wp,C] (ld == C)
 *   wq/wk/wv : depthwise weights packed [9][C] (tap-major), bq/bk/bv : [C]
 *   p_out: C8 [N,C/8,Hp,Wp,8]   (channel-blocked so that the kernel's 8-channel chunks are contiguous)
 *   logits: NCHW [N,n_cls,Hp,Wp] or NULL (then wf/bf are ignored); wf: [n_cls][C], bf: [n_cls];
 *   log_softmax != 0 applies LogSoftmax over the class dimension (PSPNet head).
 * Supported: C % 8 == 0, kH == kW in {3,5,7}, n_cls <= 32, N*C*Hp*Wp*4 < 2 GiB.
 * ------------------------------------------------------------------------------------------- */
int arseg_creff_fwd(const float *hr, const float *lr, const float *wq, const float *bq, const float *wk,
                    const float *bk, const float *wv, const float *bv, float *p_out, const float *wf, const float *bf,
                    int n_cls, float *logits, int log_softmax, int N, int C, int Hp, int Wp, int hp, int wp, int kH,
                    int kW, arseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * warpFeature + MyAttention.forward + final 1x1 classifier in ONE kernel (the non-keyframe tail of
 * EvalAlterRes, evaluation.py:176-193, for the 64-channel full-resolution PSPNet feature):
 *                                   evaluation.py:61-87,176-180; model/attention.py:184-213;
 *                                   model/pspnet.py:219-231
 * The keyframe feature is read UN-warped and sampled with the frame's motion vectors while a tile
 * is staged, so the warped tensor never exists in memory.
 *   ref_nhwc_host : HOST array of N device pointers; entry i = keyframe feature of frame i,
 *                   NHWC [Hp][Wp][C] (ld == C).  Frames of one GOP pass the same pointer.
 *   mv_q : int16 quarter-pel [N,H,W,2] as in arseg_warp_mvq_fwd (resized to (Hp,Wp) in fp64 in-kernel)
 *   lr, wq..bv, wf, bf, logits, log_softmax : as arseg_creff_fwd
 *   p_out : the fused feature, layout p_layout = ARSEG_C8 [N,C/8,Hp,Wp,8] or ARSEG_NHWC [N,Hp,Wp,C]
 * Supported: C == 64, kH == kW == 7, N <= 32, n_cls <= 32, C*Hp*Wp*4 < 2 GiB per frame (the rolling
 * kernel addresses every frame through its own buffer descriptor; launches that fall to the tile kernel --
 * 17-32 classes, impl = TILES -- need N*C*Hp*Wp*4 < 2 GiB); anything else returns ARSEG_EUNSUPPORTED and the
 * caller uses arseg_warp_mvq_fwd + arseg_creff_fwd or splits the batch.
 * ------------------------------------------------------------------------------------------- */
int arseg_creff_warp_fwd(const float *const *ref_nhwc_host, const int16_t *mv_q, int H, int W, const float *lr,
                         const float *wq, const float *bq, const float *wk, const float *bk, const float *wv,
                         const float *bv, float *p_out, int p_layout, const float *wf, const float *bf, int n_cls,
                         float *logits, int log_softmax, int N, int C, int Hp, int Wp, int hp, int wp, int kH, int kW,
                         arseg_stream_t stream);

/* The same with the kernel choice made explicit (measurements, tests): impl = enum arseg_creff_warp_impl -- AUTO / ROLL: the rolling
 * kernel (creff_roll.hip: a workgroup walks down a 16-column strip, key / value records of the 7 x 7 windows in LDS rings, producer and
 * consumer waves of different rows overlap); TILES: the 16 x 16 tile kernel of rounds 2-3 (creff_rr.hip).  seg_rows = rows of a strip
 * segment (> 0: the rolling kernel works on fixed segments of that many rows, rounded up to even, and raised if a workgroup would get more
 * than 64 of them; 0 = default: whole strips dealt to the workgroups, the remainder cut into equal runs of row pairs); max_wgs = upper bound on its persistent
 * workgroups (0 = one per compute unit; fewer leave compute units to kernels of other streams).  No environment variables are read. */
int arseg_creff_warp_fwd_ex(const float *const *ref_nhwc_host, const int16_t *mv_q, int H, int W, const float *lr,
                            const float *wq, const float *bq, const float *wk, const float *bk, const float *wv,
                            const float *bv, float *p_out, int p_layout, const float *wf, const float *bf, int n_cls,
                            float *logits, int log_softmax, int N, int C, int Hp, int Wp, int hp, int wp, int kH, int kW,
                            int impl, int seg_rows, int max_wgs, arseg_stream_t stream);

/* Which kernel arseg_creff_warp_fwd_ex runs for a launch of this shape and these knobs -- a pure query, nothing is launched: returns
 * ARSEG_CREFF_WARP_ROLL or ARSEG_CREFF_WARP_TILES (> 0), ARSEG_EUNSUPPORTED for shapes the fused entry point does not cover (or impl = ROLL on a
 * launch the rolling kernel does not admit), ARSEG_EINVAL for bad arguments.  n_cls = 0: no head.  The rule: the rolling kernel serves every
 * launch it admits (C == 64, 7 x 7, no head or <= 16 classes, a schedule that fits its 64-entry piece table); the tile kernel serves 17-32-class
 * heads, oversized schedules and impl = TILES.  (bench.py labels its roofline line with this; tests enforce the table.) */
int arseg_creff_warp_select(int N, int C, int Hp, int Wp, int hp, int wp, int kH, int kW, int n_cls, int impl, int seg_rows, int max_wgs);

/* The same with the kernel choice made explicit (measurements, tests): impl = enum arseg_creff_impl (AUTO: the matrix-core kernel
 * for C >= 128, the VALU kernel otherwise); mfma_tile_rows = 0 (by launch size), 8 or 16.  No environment variables are read. */
int arseg_creff_fwd_ex(const float *hr, const float *lr, const float *wq, const float *bq, const float *wk,
                       const float *bk, const float *wv, const float *bv, float *p_out, const float *wf, const float *bf,
                       int n_cls, float *logits, int log_softmax, int N, int C, int Hp, int Wp, int hp, int wp, int kH,
                       int kW, int impl, int mfma_tile_rows, arseg_stream_t stream);

/* Layout changes to / from C8 at the API boundary (layout = ARSEG_NCHW or ARSEG_NHWC; ld = NHWC channel stride). */
int arseg_to_c8_fwd(const float *in, int layout, int in_ld, float *out, int N, int C, int HW, arseg_stream_t stream);
int arseg_from_c8_fwd(const float *in, float *out, int layout, int out_ld, int N, int C, int HW, arseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * conv2d (+ folded BatchNorm / bias, + residual add, + activation) as an implicit GEMM on the
 * matrix cores: desc.math selects the back end -- ARSEG_MATH_F16X3 (default of the Python side: every fp32 operand split
 * into hi + lo fp16, three v_mfma_f32_32x32x16_f16 per product, fp32 accumulation; operand range below) or ARSEG_MATH_F32
 * (v_mfma_f32_32x32x2_f32).  Replaces every nn.Conv2d/BatchNorm2d/ReLU/PReLU stack of
 * model/extractors.py:35-66,108-158, model/pspnet.py:14-46 and model/bisenet.py:31-60,162-399.
 *   out[n,oy,ox,co] = act( scale[co] * sum_{r,s,ci} in[n, oy*stride-pad+r*dil, ox*stride-pad+s*dil, ci]
 *                                        * w[co][(r*S+s)*Cin+ci]  + bias[co] + residual[n,oy,ox,co] )
 * in: NHWC (in_ld), Cin % 4 == 0 (pad RGB to 4), Cin a power of two unless R*S == 1.
 * w_packed: [Cout][Kpad] from arseg_pack_conv_weight_host.  scale/bias: [Cout] or NULL (1 / 0).
 * residual: NHWC (res_ld) or NULL.  tile_cfg / split_k: 0 = choose automatically.
 * ------------------------------------------------------------------------------------------- */
typedef struct arseg_conv_desc {
    int N, H, W, Cin, in_ld;
    int Cout, out_ld, res_ld;
    int R, S, stride, pad, dil;
    int act;           /* enum arseg_act */
    float prelu_slope; /* single shared slope (nn.PReLU() default, model/pspnet.py:40) */
    int tile_cfg;      /* 0 auto; 1..4 = 128x128, 128x64, 64x64, 64x128 (K step 32) with a double-buffered LDS tile; 5..8 = the
                          same tiles single-buffered (half the LDS, more workgroups per CU); 9..12 = single-buffered, K step 64;
                          13..16 = patch-resident kernel for 3x3 stride-1 pad==dil convs under ARSEG_MATH_F16X3 (Cin % 32 == 0): the
                          input patch of a 128- (13, 14) or 256-pixel (15, 16) tile stays in LDS for all nine taps, BN = 64 / 128;
                          ARSEG_EUNSUPPORTED for other shapes; 17..19 = 256x128, 128x256, 256x256 tiles on 8 / 16 waves (F16X3 only):
                          more MFMA work per byte fetched from L2 / Infinity Cache, for wide GEMMs that fill the chip; 20..22 (r6) = the
                          patch-resident kernel with BN = 64 on squarer pixel tiles (20: 256 pixels as 8 x 32, 21: 16 x 16, 22: 128 pixels as
                          8 x 16: less halo per output than the 4 x 64 / 2 x 64 tiles 13..16 take on a wide map); refused on narrower maps */
    int split_k;       /* 0 auto, >= 1 explicit */
    /* batched mode (used by the Winograd path): `batch` independent problems of identical shape, problem b reads
       in + b*in_batch_stride, w_packed + b*w_batch_stride and writes out + b*out_batch_stride (strides in floats);
       batch <= 1 = a single problem.  No residual and no split-K in batched mode. */
    int batch;
    long long in_batch_stride, w_batch_stride, out_batch_stride;
    int math;          /* enum arseg_math: which MFMA back end evaluates the fp32 GEMM (selects the w_packed format too) */
    int upsample2x;    /* 1: `in` is the LOW-resolution tensor [N, H/2, W/2, in_ld] and the conv runs on its x2 bilinear
                          (align_corners=False) upsample, which is never materialised (PSPUpsample, model/pspnet.py:43-46): H, W stay
                          the conv's input size, both even, dil == 1.  Patch-resident plans only (tile_cfg 13..16; the 16-bit conv
                          ignores it); ARSEG_EUNSUPPORTED otherwise -- the Winograd route has its own fused form
                          (arseg_wino43_input_fwd upsample2x). */
    void *range_flag;  /* ARSEG_MATH_F16X3 only, may be NULL: a caller-owned, 4-byte aligned device word.  Bit 0 is set (atomic OR, never
                          cleared by the library) when an activation this conv multiplies exceeds range_limit in magnitude, i.e. when
                          the hi/lo fp16 pair starts to lose bits (65504) or clamps (131008): the caller reads the word once per
                          batch of launches and repeats the batch with ARSEG_MATH_F32 if it is set.  The Winograd route passes the
                          same word to its batched GEMM, which watches the TRANSFORMED activations it actually multiplies.  No host
                          synchronisation, one v_max3 per 2 activations in 1 / (Cout / tile) of the workgroups. */
    float range_limit; /* <= 0: 65504 */
} arseg_conv_desc;

/* ARSEG_MATH_F32:   v_mfma_f32_32x32x2_f32 on the fp32 operands; w_packed from arseg_pack_conv_weight_host.
 * ARSEG_MATH_F16X3: fp32 emulated on the fp16 matrix cores: x = hi + lo (two fp16, 22 significant bits),
 *                   a.b = a_hi.b_hi + a_hi.b_lo + a_lo.b_hi with fp32 accumulation (error ~2^-21 relative per product,
 *                   activations must satisfy |x| <= 131008 = 2 x 65504: the hi/lo pair clamps beyond, full 22-bit precision below 65504; |x| below the fp16 normal range carries an absolute error <= 6e-8).  w_packed from arseg_split_weight_f16x3_host, and `scale` must
 *                   carry that function's per-channel chan_mul_inv factor.
 * ARSEG_MATH_F16:   reduced precision: plain fp16 operands (activations rounded to nearest, the hi halves of the same split
 *                   weights), one fp16 MFMA per product, fp32 accumulation; ~1e-3 relative error per conv.  Not used by default. */
enum arseg_math { ARSEG_MATH_F32 = 0, ARSEG_MATH_F16X3 = 1, ARSEG_MATH_F16 = 2 };

int arseg_conv_out_hw(const arseg_conv_desc *d, int *Ho, int *Wo);
size_t arseg_conv2d_workspace_bytes(const arseg_conv_desc *d);
int arseg_conv2d_fwd(const arseg_conv_desc *d, const float *in, const float *w_packed, const float *scale,
                     const float *bias, const float *residual, float *out, void *workspace, size_t workspace_bytes,
                     arseg_stream_t stream);

/* Plan selection ("find", what MIOpen calls miopenFindConvolutionForwardAlgorithm; the reference gets it implicitly from
 * torch.backends.cudnn.benchmark, train.py / evaluation.py): runs every launch plan the shape admits -- all tile_cfg it supports x
 * split-K {1,2,3,4,6,8}, plus the built-in heuristic (0,0) -- `reps` times each (<= 0: 3) on the caller's buffers, times them with
 * HIP events and returns the fastest as (*tile_cfg, *split_k) for arseg_conv_desc, its time in *best_us (may be NULL).
 * d->tile_cfg / d->split_k are ignored.  `workspace` should hold arseg_conv2d_find_workspace_bytes(d) bytes (candidates needing more
 * than workspace_bytes are skipped).  The ONLY entry point that synchronises the stream; `out` holds a valid result afterwards.
 * Returns the error of the heuristic plan if no candidate could be launched. */
size_t arseg_conv2d_find_workspace_bytes(const arseg_conv_desc *d);
int arseg_conv2d_find(const arseg_conv_desc *d, const float *in, const float *w_packed, const float *scale, const float *bias,
                      const float *residual, float *out, void *workspace, size_t workspace_bytes, int reps, int *tile_cfg,
                      int *split_k, float *best_us, arseg_stream_t stream);

/* Winograd F(4x4,3x3) path for 3x3 stride-1 convs with pad == dil (model/extractors.py:30-32 conv3x3, model/pspnet.py:38):
 *   V[36][T][Cin]  = arseg_wino43_input_fwd(in NHWC)          T = arseg_wino43_tiles(N,H,W,dil)
 *   M[36][T][Cout] = 36 GEMMs V[k] x U[k]^T                   arseg_conv2d_fwd in batched mode (batch = 36, 1x1)
 *   out NHWC       = arseg_wino43_output_fwd(M) with the usual scale / bias / residual / activation epilogue
 * U = arseg_wino43_pack_weight_host(w OIHW) -> [36][Cout][Cin] (Cin % 32 == 0 so that it is a valid packed 1x1 weight).
 * Operand range under ARSEG_MATH_F16X3: the GEMM operands are the TRANSFORMED activations V = B^T d B, up to 100x (typically ~10x) the
 * activations, so unscaled the split-fp16 range (|V| <= 131008) is reached for |x| >~ 1.3e3 in the worst case (beyond it V clamps): pass
 * v_scale = 2^-4 / m_scale = 2^4 (what the Python layer does) to move the limit to |x| ~ 2e4 worst case; the price is the absolute
 * floor of the low fp16 half (subnormal step 2^-24) rising to 2^-20 in units of V, still below Winograd's own fp32 rounding for O(1) data.
 * 2.25x..4x fewer MACs than the direct form; fp32 rounding error ~1e-5 relative instead of ~1e-6. */
long long arseg_wino43_tiles(int N, int H, int W, int dil);
/* upsample2x != 0: `in` is the low-resolution tensor [N,H/2,W/2,C] and the x2 bilinear (align_corners=False) upsample of
 * PSPUpsample (model/pspnet.py:45) is applied on the fly (H, W = upsampled size, even; dil == 1). */
/* v_scale / m_scale: V is stored multiplied by v_scale (> 0), M is multiplied by m_scale before the epilogue; with powers of two and
 * m_scale = 1 / v_scale the result is unchanged bit for bit while the GEMM operands shrink -- under ARSEG_MATH_F16X3 the transformed
 * activations (up to 100x, typically ~10x the input) otherwise leave the split-fp16 range for |x| >~ 1.3e3.  1.0f = no scaling. */
int arseg_wino43_input_fwd(const float *in, int in_ld, float *V, int N, int H, int W, int C, int dil, int upsample2x, float v_scale,
                           arseg_stream_t stream);
/* The same transform with V written in the split-row operand format of arseg_gemm_x3_fwd (C % 32 == 0, V_split 16-byte aligned, the
 * same T*C*4 bytes per frequency): the 36 GEMMs then run as ONE arseg_gemm_x3_fwd(V_split, U_f16x3, M, T, Cout, C, ..., batch = 36).
 * range_flag / range_limit: as in arseg_conv_desc -- the transform is where the fp32 operands of that GEMM are last seen. */
int arseg_wino43_input_split_fwd(const float *in, int in_ld, void *V_split, int N, int H, int W, int C, int dil, int upsample2x,
                                 float v_scale, void *range_flag, float range_limit, arseg_stream_t stream);
int arseg_wino43_output_fwd(const float *M, const float *scale, const float *bias, const float *residual, int res_ld, float *out,
                            int out_ld, int N, int H, int W, int Cout, int dil, int act, float prelu_slope, float m_scale,
                            arseg_stream_t stream);

/* The per-image pyramid operand of arseg_gemm_x3_cat_fwd for the folded PSP bottleneck (model/pspnet.py:14-31) in one pass:
 * out[n][co][k] (split rows, K = 64) = t[n][k][co] * unscale[co] for k < rows, 0 beyond.  t fp32 [N, rows, Cout], rows <= 64.
 * range_flag / range_limit: as in arseg_conv_desc -- this pass is where the fp32 values of that GEMM operand are last seen (un-scaled
 * pyramid terms can leave the split-fp16 range when the folded scale of a channel is small); NULL: no watch. */
int arseg_psp_w2_split_fwd(const float *t, const float *unscale, void *out, int N, int rows, int Cout, void *range_flag, float range_limit,
                           arseg_stream_t stream);

/* (Part of the nn.Conv2d replacement above: the GEMM inside conv3x3 on the Winograd route, /root/reference/model/extractors.py:30-32,
 * the 1x1 bottleneck of PSPModule, model/pspnet.py:26, and the low-resolution tap GEMM of PSPUpsample, model/pspnet.py:38-46.)
 * Batched GEMM on operands that are already split into fp16 (hi, lo) pairs ("split rows": a row of K values, K % 32 == 0, is K/32
 * groups of 128 bytes = 32 hi halves then 32 lo halves; the f16x3 weight format of arseg_split_weight_f16x3_host, now also for the
 * activations):   out[b][m][n] = act(scale[n] * sum_k x[b][m][k] * w[b][n][k] + bias[n] + residual[m][n])   (scale / bias / residual may
 * be NULL; residual fp32 [M][res_ld], batch == 1 only)
 * x_split [batch][M][K], w_split [batch][N][K] in split rows (batch strides in BYTES, multiples of 16), out fp32 [batch][M][out_ld]
 * (stride in floats), N % 4 == 0.  Operands reach LDS by LDS-DMA, no register staging (csrc/gemm_x3.hip).  tile_cfg 0..11
 * (M x N per workgroup): 0 = 256x256 with 8 waves, 1 = 256x256 with 16, 2 / 5 = 128x256 with 8 / 16, 3 = 128x128,
 * 4 = 256x128, 6 = 256x256 with 16 waves in two groups half a K step apart; 7 = 256x64, 8 = 128x64 with 4 waves, 9 = 128x64, 10 = 64x128, 11 = 128x128 with 32x64 wave
 * tiles (round 5: narrow tiles for 64- / 128-channel outputs; anything else: ARSEG_EINVAL).  out_split != 0: `out` is written as split rows too (N % 32 == 0, out_ld == N) -- it is the next GEMM's x_split, e.g. the
 * PSP bottleneck feeding the tap-decomposed up_1 conv.  The same fp32-grade arithmetic as ARSEG_MATH_F16X3 (three fp16 MFMAs per product).
 * arseg_split_rows_fwd converts fp32 rows [rows][in_ld] (times `mul`, a power of two keeps it exact) to split rows [rows][K].
 * range_flag / range_limit (may be NULL / <= 0: 65504): the operand range word of arseg_conv_desc, set by whoever WRITES split rows from
 * fp32 values (the split pass, a GEMM with out_split) -- the consuming GEMM never sees the fp32 operand. */
int arseg_split_rows_fwd(const float *in, long long in_ld, void *out_split, long long rows, int K, float mul, void *range_flag,
                         float range_limit, arseg_stream_t stream);
int arseg_gemm_x3_fwd(const void *x_split, const void *w_split, float *out, int M, int N, int K, int out_ld, int batch,
                      long long x_batch_stride, long long w_batch_stride, long long out_batch_stride, const float *scale,
                      const float *bias, const float *residual, int res_ld, int act, float prelu_slope, int out_split, int tile_cfg,
                      void *range_flag, float range_limit, arseg_stream_t stream);
/* The same GEMM over a K-concatenated operand pair: out = act(scale * (x . w^T + x2 . w2^T) + bias), x2 [batch][M][K2], w2 [batch][N][K2] split rows
 * with their own batch strides (0 = shared by the batch).  The folded PSP pyramid uses it (model/pspnet.py:14-31): x2 = the bilinear
 * interpolation matrix of the pooled rows (shared), w2 = the per-image pyramid terms, so that sum_s upsample(...) is two more K steps of the
 * bottleneck GEMM instead of a 92 MB tensor written by one kernel and read back as a residual by the next. */
int arseg_gemm_x3_cat_fwd(const void *x_split, const void *w_split, const void *x2_split, const void *w2_split, float *out, int M, int N, int K,
                          int K2, int out_ld, int batch, long long x_batch_stride, long long w_batch_stride, long long x2_batch_stride,
                          long long w2_batch_stride, long long out_batch_stride, const float *scale, const float *bias, int act,
                          float prelu_slope, int out_split, int tile_cfg, void *range_flag, float range_limit, arseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * 1x1 stride-1 convolution of the 16-bit storage path as a plain GEMM of the LDS-DMA kernel (csrc/gemm_x3.hip) -- the 1x1 ConvBNReLU layers of
 * /root/reference/model/bisenet.py:162-186,335-340,387-399: out[m][co] = act(scale[co] * x[m] . w[co] + bias[co] + residual[m][co]),
 * x [M][K] / w [Cout][K] fp16 or bf16 (dtype: enum arseg_dtype; K % 64 == 0, dense rows), out / residual 16-bit with row strides out_ld / res_ld
 * (elements; a channel slice of a wider tensor is fine), scale / bias fp32 (NULL: 1 / 0), Cout % 4 == 0, 16-byte aligned pointers.
 * tile_cfg 0-11 (0-6 as arseg_gemm_x3_fwd, 7-11 narrow tiles for Cout = 64 / 128); none is chosen for the caller.
 * (Round 5's implicit-3x3 entry points on zero-bordered rows, arseg_conv3x3_rows_fwd / arseg_pad_rows_fwd, were removed in ABI v5: never selected.)
 * ------------------------------------------------------------------------------------------- */
int arseg_gemm_rows16_fwd(const void *x_rows, const void *w_rows, void *out, int dtype, long long M, int K, int Cout, int out_ld, const float *scale,
                          const float *bias, const void *residual, int res_ld, int act, float prelu_slope, int tile_cfg, arseg_stream_t stream);

/* conv3x3 (pad 1, stride 1) of a x2 bilinear (align_corners=False) upsample -- PSPUpsample, /root/reference/model/pspnet.py:43-46 --
 * by tap decomposition: since a 1x1 conv commutes with a per-channel resize, conv3x3(Up(x)) = sum_t shift_t(Up(W_t x)).  The caller
 * runs ONE 1x1 conv at low resolution with the nine taps stacked along the output channels (weights [9*Cout][Cin], row t*Cout + co =
 * W[co][.][t/3][t%3]; arseg_conv2d_fwd, no epilogue) into z = [N,h,w,9*Cout] (row stride z_ld), and this entry point samples the nine
 * planes at the shifted positions of the never-materialised upsampled image (zero outside it = the conv's padding), sums them and applies
 * out = act(scale * sum + bias) into out = [N,2h,2w,Cout].  Same multiply count as Winograd F(4x4,3x3) on the upsampled image, without
 * its transformed operand (9x the low-resolution input) and without its rounding amplification.  Cout % 4 == 0, 16-byte aligned. */
int arseg_upconv3x3_tap_gather_fwd(const float *z, int z_ld, const float *scale, const float *bias, float *out, int out_ld, int N, int h,
                                   int w, int Cout, int act, float prelu_slope, arseg_stream_t stream);
/* The same gather with its output written as split rows [N, 2h, 2w, Cout] (Cout % 32 == 0; the operand format of arseg_gemm_x3_fwd): the next
 * layer is again a tap-decomposed PSPUpsample whose low-resolution GEMM stages them by LDS-DMA (up_1 -> up_2).  range_flag / range_limit as in
 * arseg_split_rows_fwd. */
int arseg_upconv3x3_tap_gather_split_fwd(const float *z, int z_ld, const float *scale, const float *bias, void *out_split, int N, int h, int w,
                                         int Cout, int act, float prelu_slope, void *range_flag, float range_limit, arseg_stream_t stream);
int arseg_wino43_pack_weight_host(const float *w_oihw, int Cout, int Cin, float *out_host);


/* Host-side weight preparation (the "weight packer"; CPU pointers).
 * arseg_packed_k: padded GEMM depth for a conv (multiple of 32).
 * arseg_pack_conv_weight_host: OIHW [Cout][Cin][R][S] -> [Cout][Kpad], k = (r*S+s)*Cin_pad + ci, zero padded.
 * arseg_fold_bn_host: scale = gamma/sqrt(var+eps), bias = beta + (conv_bias - mean)*scale  (conv_bias may be NULL).
 * arseg_pack_dw3x3_host: depthwise [C][1][3][3] -> [9][C]. */
int arseg_packed_k(int Cin_pad, int R, int S);
int arseg_pack_conv_weight_host(const float *w_oihw, int Cout, int Cin, int R, int S, int Cin_pad, float *out_host);
/* arseg_split_weight_f16x3_host: [Cout][Kpad] fp32 (from arseg_pack_conv_weight_host / arseg_wino43_pack_weight_host rows)
 * -> ARSEG_MATH_F16X3 operand format (same byte count: per 32-k tile 32 hi halves then 32 lo halves), each row multiplied
 * by a power of two; chan_mul_inv[Cout] receives the inverse factors (multiply them into the epilogue scale); NULL = no scaling. */
int arseg_split_weight_f16x3_host(const float *w_packed_host, int Cout, int Kpad, void *out_host, float *chan_mul_inv);
int arseg_fold_bn_host(const float *gamma, const float *beta, const float *mean, const float *var, float eps,
                       const float *conv_bias, int C, float *scale_out, float *bias_out);
int arseg_pack_dw3x3_host(const float *w, int C, float *out_host);

/* ---------------------------------------------------------------------------------------------
 * Small NHWC layers of the backbones.
 * ------------------------------------------------------------------------------------------- */
/* nn.MaxPool2d(3, stride 2, padding 1)             model/extractors.py:116, model/bisenet.py:77 */
int arseg_maxpool3x3s2_fwd(const float *in, float *out, int N, int H, int W, int C, arseg_stream_t stream);
/* nn.AdaptiveAvgPool2d((oh,ow))                                            model/pspnet.py:23
 * out[n][bin][c] at out + n*out_n_stride + bin*out_ld + c (0 = dense defaults: out_ld = C, out_n_stride = oh*ow*C),
 * so several pyramid levels can be pooled straight into one block-structured matrix. */
int arseg_adaptive_avgpool_fwd(const float *in, int in_ld, float *out, int out_ld, long long out_n_stride, int N, int H, int W,
                               int C, int oh, int ow, arseg_stream_t stream);
/* The same into one column block of a block-structured matrix [N][rows][n_blocks*C] (the folded PSP pyramid, model/pspnet.py:14-31):
 * level `block` writes its pooled map into columns [block*C, (block+1)*C) of its oh*ow rows and ZEROS into the other blocks of those
 * rows -- no fill launch.  `out` = the level's first row in image 0 (column 0 of the matrix), out_n_stride = elements between images. */
int arseg_adaptive_avgpool_blockrow_fwd(const float *in, int in_ld, float *out, long long out_n_stride, int N, int H, int W, int C,
                                        int oh, int ow, int n_blocks, int block, arseg_stream_t stream);
/* The whole pooled matrix of the folded pyramid in one pass over the map: out [N][rows][n_sizes * C], rows = sum sizes[i]^2, identical to n_sizes
 * calls of arseg_adaptive_avgpool_blockrow_fwd up to the order of summation (every bin is a union of cells of the grid spanned by all bin
 * edges; the cells are summed once, 1 / 4 of the reads).  sizes[i] <= 6, n_sizes <= 4; workspace = arseg_psp_pool_matrix_workspace_bytes. */
size_t arseg_psp_pool_matrix_workspace_bytes(int N, int H, int W, int C, int n_sizes, const int *sizes);
int arseg_psp_pool_matrix_fwd(const float *in, int in_ld, float *out, void *workspace, size_t workspace_bytes, int N, int H, int W, int C,
                              int n_sizes, const int *sizes, arseg_stream_t stream);
/* PSPModule priors (model/pspnet.py:27-30), folded: with t[n][off_s + i][c] the per-level maps AFTER the stage conv and
 * the level's slice of the bottleneck conv (both 1x1, i.e. linear and commuting with bilinear upsampling), this writes
 * out[n,y,x,c] = sum_s upsample_bilinear(align_corners=False)(t_s[n])(y,x,c); off_s = sum_{j<s} sizes[j]^2 (host array). */
int arseg_psp_prior_sum_fwd(const float *t, float *out, int N, int H, int W, int C, int n_sizes, const int *sizes_host,
                            arseg_stream_t stream);
/* torch.mean(x,(2,3)) / F.adaptive_max_pool2d(x,1): out [N][C]   model/bisenet.py:252,292,390; pspnet.py:94 */
int arseg_global_reduce_fwd(const float *in, int in_ld, float *out, int N, int H, int W, int C, int op,
                            arseg_stream_t stream);
/* The same reduction with a caller-owned workspace (arseg_global_reduce_workspace_bytes, 0 = not needed): a large map of few images (the keyframe's
 * auxiliary head) is reduced in two deterministic stages -- row bands into the workspace, then the bands in order -- so that the whole chip reads it. */
size_t arseg_global_reduce_workspace_bytes(int N, int H, int W, int C);
int arseg_global_reduce_ws_fwd(const float *in, int in_ld, float *out, void *workspace, size_t workspace_bytes, int N, int H, int W, int C, int op,
                               arseg_stream_t stream);
/* F.interpolate / F.upsample / nn.Upsample: nearest or bilinear, align_corners on/off, either layout.
 * NHWC: in_ld/out_ld channel strides (C % 4 == 0); NCHW: planes contiguous, ld arguments ignored.
 * model/pspnet.py:29,45,97; model/bisenet.py:215,284,298,442; evaluation.py:117,188,201 */
int arseg_resize_fwd(const float *in, float *out, int N, int C, int Hin, int Win, int Hout, int Wout, int mode,
                     int align_corners, int layout, int in_ld, int out_ld, arseg_stream_t stream);
/* out[n,y,x,c] = x[n,y,x,c] * scale[n,c] + (add_full ? add_full[n,y,x,c] : 0) + (add_vec ? add_vec[n,c] : 0)
 * ARM: feat*atten (+avg)  model/bisenet.py:258,295;  FFM: feat*atten + feat  model/bisenet.py:397-398 */
int arseg_scale_add_fwd(const float *x, const float *scale, const float *add_full, const float *add_vec, float *out,
                        int N, int HW, int C, arseg_stream_t stream);
/* final 1x1 classifier on an NHWC feature, NCHW logits out (+ optional LogSoftmax over classes)
 * model/pspnet.py:66-67,96-98; model/bisenet.py:211,448 */
int arseg_head_fwd(const float *p, int p_ld, const float *wf, const float *bf, float *logits, int N, int HW, int C,
                   int n_cls, int log_softmax, arseg_stream_t stream);
/* decoded frame NCHW [N,3,H,W] -> NHWC4 [N,h,w,4] (4th channel 0), bilinear align_corners=True when (h,w) != (H,W)
 * evaluation.py:115-117,186-188 fused with the layout change the conv engine wants */
int arseg_frame_to_nhwc4_fwd(const float *img, float *out, int N, int H, int W, int h, int w, arseg_stream_t stream);
/* Decoded uint8 HWC frame(s) [N,H,W,3] (device) -> normalised NHWC4 at (h,w): ToTensor + Normalize(mean, std)
 * (dataset/camvid.py:503-506, dataset/cityscapes.py:208-214) + the evaluator's bilinear align_corners=True downscale
 * (evaluation.py:186-188) in one pass.  mean3 / std3: host pointers to 3 floats. */
int arseg_frame_u8_to_nhwc4_fwd(const uint8_t *img_hwc, float *out, int N, int H, int W, int h, int w, const float *mean3,
                                const float *std3, arseg_stream_t stream);

/* mergeMotion (pre-process/generate_compressed_dataset_camvid.py:6-56): chains the codec's per-frame motion fields back to
 * the keyframe.  flows: int16 [n_frames+1][H][W][3] = (mv_x, mv_y quarter-pel, reference index), entries <= frame_start unused;
 * out: int16 [n_frames+1][H][W][2], frame f > 0 = accumulated quarter-pel motion of frame f to the keyframe (what the
 * datasets' .bin files hold), frame 0 = -1 as in the reference.  workspace: arseg_merge_motion_workspace_bytes() bytes. */
size_t arseg_merge_motion_workspace_bytes(int n_frames, int H, int W);
int arseg_merge_motion_fwd(const int16_t *flows, int16_t *out, void *workspace, size_t workspace_bytes, int n_frames, int frame_start,
                           int H, int W, arseg_stream_t stream);
/* layout changes at the API boundary */
int arseg_nchw_to_nhwc_fwd(const float *in, float *out, int N, int C, int HW, int out_ld, arseg_stream_t stream);
int arseg_nhwc_to_nchw_fwd(const float *in, int in_ld, float *out, int N, int C, int HW, arseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * 16-bit storage path (BASELINE configs[2]: BiSeNet-18 bf16, configs[4]: BiSeNet-18 0.3x fp16; model/bisenet.py:438-461,546-575).
 * Activations and weights are NHWC fp16 or bf16 (dtype = ARSEG_DT_F16 | ARSEG_DT_BF16), every product is ONE
 * v_mfma_f32_32x32x16_{f16,bf16}, accumulation and the epilogue (folded BN scale / bias, residual, activation) are fp32, results are
 * rounded to 16 bits once, at the store.  Cin % 8 == 0 (frames are ingested as NHWC8), in_ld / out_ld / res_ld % 8 == 0.
 *   arseg_pack_conv_weight16_host: OIHW fp32 -> [Cout][Kpad16] 16-bit with k = (r*S+s)*Cin_pad + ci, Kpad16 = arseg_packed_k16(...)
 *   arseg_conv2d16_fwd: desc as arseg_conv2d_fwd (tile_cfg: 0 auto; 1 / 2 = 64- / 128-channel tile with K step 32; 3 / 4 = the same with
 *                       K step 64; 5..8 = patch-resident kernel for 3x3 stride-1 pad == dil convs with Cin % 64 == 0 (the input patch of a
 *                       128- (5, 6) / 256-pixel (7, 8) tile stays in LDS for all nine taps, 64 / 128 output channels; no split-K;
 *                       ARSEG_EUNSUPPORTED for other shapes; 10..13 = the same kernel on squarer pixel tiles -- 10: 256 pixels as 8 x 32, 64 channels;
 *                       11: 16 x 16, 64 channels; 12: 8 x 32, 128 channels; 13: 128 pixels as 8 x 16, 64 channels -- refused on maps whose
 *                       default tile is already that narrow); 9 = stem kernel (7x7 stride 2 pad 3, NHWC8 -> 64 channels, no residual: all weights
 *                       resident in LDS, one staged input patch per 8x32 output tile); split_k: 0 = automatic -- K slices for launches whose tiles do not fill the chip, e.g. the 16x32-map
 *                       layers of BiSeNet-18 --, >= 1 explicit; deterministic: fp32 partial sums in `workspace`
 *                       (arseg_conv2d16_workspace_bytes(desc) bytes, 0 without split-K), summed in slice order by a second kernel that
 *                       applies the epilogue; split-K needs Cout % 8 == 0, otherwise one slice.  batch unused)
 * ------------------------------------------------------------------------------------------- */
int arseg_packed_k16(int Cin_pad, int R, int S);
int arseg_pack_conv_weight16_host(const float *w_oihw_host, int Cout, int Cin, int R, int S, int Cin_pad, int dtype, void *out_host);
size_t arseg_conv2d16_workspace_bytes(const arseg_conv_desc *d);
int arseg_conv2d16_fwd(const arseg_conv_desc *d, int dtype, const void *in, const void *w_packed16, const float *scale,
                       const float *bias, const void *residual, void *out, void *workspace, size_t workspace_bytes,
                       arseg_stream_t stream);
/* The small layers on 16-bit NHWC tensors (C, ld % 8 == 0), same arithmetic as their fp32 counterparts above, fp32 inside:
 *   frame ingest: NCHW fp32 RGB -> NHWC8 (channels 3..7 zero) + bilinear align_corners=True downscale      evaluation.py:186-188
 *   maxpool 3x3 s2 p1; torch.mean(x,(2,3)) -> [N][C]; resize (nearest | bilinear, align_corners on / off); x*scale[n,c] (+add_full) (+add_vec[n,c])
 *   head: 1x1 classifier (fp32 weights) on a 16-bit feature -> fp32 NCHW logits (+ LogSoftmax)
 *   cast: fp32 <-> 16-bit element conversion (count % 8 == 0)
 *   warp_mvq16: arseg_warp_mvq_fwd on a 16-bit keyframe feature, fp32 C8 out (the CReFF kernels' input layout) */
int arseg_frame_to_nhwc8_16_fwd(const float *img, void *out, int dtype, int N, int H, int W, int h, int w, arseg_stream_t stream);
int arseg_maxpool3x3s2_16_fwd(const void *in, void *out, int dtype, int N, int H, int W, int C, arseg_stream_t stream);
size_t arseg_global_mean16_workspace_bytes(int N, int H, int W, int C);      /* fp32 partial sums of pixel slices */
int arseg_global_mean16_fwd(const void *in, int in_ld, void *out, int dtype, int N, int H, int W, int C, void *workspace,
                            size_t workspace_bytes, arseg_stream_t stream);
int arseg_resize16_fwd(const void *in, void *out, int dtype, int N, int C, int Hin, int Win, int Hout, int Wout, int mode,
                       int align_corners, int in_ld, int out_ld, arseg_stream_t stream);
int arseg_scale_add16_fwd(const void *x, const void *scale, const void *add_full, const void *add_vec, void *out, int dtype, int N,
                          int HW, int C, arseg_stream_t stream);
int arseg_head16_fwd(const void *p, int p_ld, int dtype, const float *wf, const float *bf, float *logits, int N, int HW, int C,
                     int n_cls, int log_softmax, arseg_stream_t stream);
int arseg_cast_fwd(const void *in, int in_dtype, void *out, int out_dtype, long long count, arseg_stream_t stream);
int arseg_warp_mvq16_fwd(const void *feature, int dtype, const int16_t *mv_q, float *out_c8, int N, int C, int Hp, int Wp, int H,
                         int W, arseg_stream_t stream);
/* The same with an explicit element stride between the frames' features: 0 = all N frames (the non-keyframes of one GOP) sample the same
 * keyframe feature -- one launch per GOP instead of one per frame. */
int arseg_warp_mvq16_shared_fwd(const void *feature, long long feat_n_stride, int dtype, const int16_t *mv_q, float *out_c8, int N, int C,
                                int Hp, int Wp, int H, int W, arseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Evaluator tail                                                       evaluation.py:201-213
 * logits NCHW [N,n_cls,h,w] -> bilinear resize to HxW -> argmax -> pred int32 [N,H,W]
 * and hist[label*n_cls+pred] += 1 for label != ignore_label (hist: int64 [n_cls*n_cls], accumulated).
 * pred or hist/label may be NULL.
 *   align_corners != 0: the evaluator's F.interpolate(logits, label_size, align_corners=True) (evaluation.py:201);
 *   align_corners == 0: nn.Upsample(scale_factor, bilinear, align_corners=False) = BiSeNetOutput.up (model/bisenet.py:215-216),
 *                       i.e. head -> x8 upsample -> argmax without materialising the [n_cls,H,W] logits (the evaluator's own resize is
 *                       then the identity: labels have the frame's size).
 * argmax follows torch.argmax (first maximum wins; NaN counts as maximum).  The reference applies softmax first (evaluation.py:203):
 * monotone, so the result is the same except for top-two logits closer than fp32 exp() can separate (< 3e-8 apart, |logit| < 0.25).
 * ------------------------------------------------------------------------------------------- */
int arseg_argmax_confusion_fwd(const float *logits, const int64_t *label, int32_t *pred, int64_t *hist, int N,
                               int n_cls, int h, int w, int H, int W, int ignore_label, int align_corners, arseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Measurement aids (no reference counterpart; BASELINE.md section 3: roofline fractions are reported against the datasheet peaks AND
 * against on-box micro-benchmarks).  bench.py times each with HIP events and prints `peaks_measured`.
 *   arseg_peak_stream_copy: dst[0 .. n_bytes) = src[0 .. n_bytes) with 16-byte accesses (n_bytes % 16 == 0, both 16-byte aligned):
 *                           2 x n_bytes of HBM traffic per launch.
 *   arseg_peak_mfma_f16:    a full-chip launch whose every wave issues iters x 8 independent v_mfma_f32_32x32x16_f16 and touches no
 *                           memory (scratch: >= 4 bytes, never written in practice); *flops_out (may be NULL) = FLOPs issued per launch.
 * ------------------------------------------------------------------------------------------- */
int arseg_peak_stream_copy(const void *src, void *dst, size_t n_bytes, arseg_stream_t stream);
int arseg_peak_mfma_f16(float *scratch, int iters, double *flops_out, arseg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The symbol names SURVEY.md section 8(b) lists for this boundary, as aliases of the entry points above (identical arguments):
 *   arseg_creff_fused_fwd     = arseg_creff_warp_fwd      (warp + CReFF + final 1x1 in one launch)
 *   arseg_conv2d_bn_act_fwd   = arseg_conv2d_fwd          arseg_pack_weights      = arseg_pack_conv_weight_host
 *   arseg_maxpool3x3s2        = arseg_maxpool3x3s2_fwd    arseg_adaptive_avgpool  = arseg_adaptive_avgpool_fwd
 *   arseg_global_reduce       = arseg_global_reduce_fwd   arseg_resize            = arseg_resize_fwd
 *   arseg_scale_add           = arseg_scale_add_fwd
 * ------------------------------------------------------------------------------------------- */
int arseg_creff_fused_fwd(const float *const *ref_nhwc_host, const int16_t *mv_q, int H, int W, const float *lr, const float *wq,
                          const float *bq, const float *wk, const float *bk, const float *wv, const float *bv, float *p_out, int p_layout,
                          const float *wf, const float *bf, int n_cls, float *logits, int log_softmax, int N, int C, int Hp, int Wp, int hp,
                          int wp, int kH, int kW, arseg_stream_t stream);
int arseg_conv2d_bn_act_fwd(const arseg_conv_desc *d, const float *in, const float *w_packed, const float *scale, const float *bias,
                            const float *residual, float *out, void *workspace, size_t workspace_bytes, arseg_stream_t stream);
int arseg_pack_weights(const float *w_oihw_host, int Cout, int Cin, int R, int S, int Cin_pad, float *out_host);
int arseg_maxpool3x3s2(const float *in, float *out, int N, int H, int W, int C, arseg_stream_t stream);
int arseg_adaptive_avgpool(const float *in, int in_ld, float *out, int out_ld, long long out_n_stride, int N, int H, int W, int C, int oh,
                           int ow, arseg_stream_t stream);
int arseg_global_reduce(const float *in, int in_ld, float *out, int N, int H, int W, int C, int op, arseg_stream_t stream);
int arseg_resize(const float *in, float *out, int N, int C, int Hin, int Win, int Hout, int Wout, int mode, int align_corners, int layout,
                 int in_ld, int out_ld, arseg_stream_t stream);
int arseg_scale_add(const float *x, const float *scale, const float *add_full, const float *add_vec, float *out, int N, int HW, int C,
                    arseg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ARSEG_HIP_H */
