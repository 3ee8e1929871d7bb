// conv3x3(pad 1) of a x2 bilinear upsample (PSPUpsample, /root/reference/model/pspnet.py:34-46) without the upsampled tensor and
// without a transform of the wide input:
//
//   conv3x3(Up(x)) = sum_t shift_t( W_t . Up(x) ) = sum_t shift_t( Up(W_t . x) )          (a 1x1 conv commutes with a per-channel resize)
//
// so the nine taps W_t are applied as ONE 1x1 conv at LOW resolution (a plain GEMM [N*h*w, Cin] x [Cin, 9*Cout]: 2.25 multiplies per output
// pixel and (ci, co) pair -- the same count as Winograd F(4x4,3x3), but its A operand is the low-resolution tensor read once instead
// of a transformed tensor nine times its size), and this kernel finishes the job: for every output pixel the nine tap planes are
// sampled at the shifted position of the (never materialised) upsampled image -- zero outside it, which is the conv's padding;
// ATen's clamped bilinear taps inside it -- summed, and sent through the folded BN + activation epilogue.
//
// Exact x2, align_corners=False: upsampled row 2y is .25 L[y-1] + .75 L[y], row 2y+1 is .75 L[y] + .25 L[y+1] with the row index clamped
// to the image (ATen clamps the source coordinate at 0 and the second tap at h-1; the clamped blend equals it to an ulp).  All four
// outputs of the 2x2 block of low-resolution pixel (y, x) depend on the 3x3 low-resolution neighbourhood only, separably:
//   H_ky(r, X) = sum_kx UpX(z_{ky,kx}(r, .))(X + kx - 1)        per low-resolution row r, both X = 2x, 2x+1
//   out(Y, X)  = sum_ky UpY(H_ky(., X))(Y + ky - 1)             both Y = 2y, 2y+1
// A thread owns a low-resolution column x and four channels and walks a strip of rows keeping H of three rows in registers: 27 16-byte
// loads per low-resolution pixel, the 3x column re-use is left to L1/L2.  Memory-bound: z is read once from HBM, out written once.
#include "arseg_common.h"

namespace {

__device__ __forceinline__ float up_act(float v, int act, float slope) {
    switch (act) {
        case ARSEG_ACT_RELU: return fmaxf(v, 0.0f);
        case ARSEG_ACT_PRELU: return v >= 0.0f ? v : v * slope;
        case ARSEG_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
        default: return v;
    }
}

__global__ __launch_bounds__(256) void up2_tap_gather_kernel(const float *__restrict__ z, int z_ld, const float *__restrict__ scale,
                                                             const float *__restrict__ bias, float *__restrict__ out, int out_ld, int N, int h,
                                                             int w, int C, int rs, int act, float slope) {
    const int Cv = C >> 2, strips = (h + rs - 1) / rs;
    const long long total = (long long)N * strips * w * Cv;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % Cv) * 4;
    long long q = idx / Cv;
    const int x = (int)(q % w); q /= w;
    const int s = (int)(q % strips);
    const int n = (int)(q / strips);
    const int col[3] = {max(x - 1, 0), x, min(x + 1, w - 1)};
    const bool vL = x >= 1, vR = x + 1 < w;
    const float *zn = z + (size_t)n * h * w * z_ld + c;

    auto hrow = [&](int r, f32x4 (&H)[3][2]) {
        const float *zr = zn + (size_t)r * w * z_ld;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            f32x4 a[3][3];                                   // [kx][column x-1, x, x+1]
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int j = 0; j < 3; ++j) a[kx][j] = *reinterpret_cast<const f32x4 *>(zr + (size_t)col[j] * z_ld + (ky * 3 + kx) * C);
            // X = 2x: upsampled columns 2x-1 (outside the image for x == 0), 2x, 2x+1
            f32x4 e = (0.25f * a[1][0] + 0.75f * a[1][1]) + (0.75f * a[2][1] + 0.25f * a[2][2]);
            const f32x4 el = 0.75f * a[0][0] + 0.25f * a[0][1];
            H[ky][0] = vL ? e + el : e;
            // X = 2x+1: upsampled columns 2x, 2x+1, 2x+2 (outside for x == w-1)
            f32x4 o = (0.25f * a[0][0] + 0.75f * a[0][1]) + (0.75f * a[1][1] + 0.25f * a[1][2]);
            const f32x4 orr = 0.25f * a[2][1] + 0.75f * a[2][2];
            H[ky][1] = vR ? o + orr : o;
        }
    };

    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
    if (scale) sc = *reinterpret_cast<const f32x4 *>(scale + c);
    if (bias) bi = *reinterpret_cast<const f32x4 *>(bias + c);
    const int y0 = s * rs, y1 = min(y0 + rs, h);
    f32x4 Hm[3][2], H0[3][2], Hp[3][2];
    hrow(max(y0 - 1, 0), Hm);
    hrow(y0, H0);
    float *on = out + (size_t)n * (2 * h) * (2 * w) * out_ld + c;
    for (int y = y0; y < y1; ++y) {
        hrow(min(y + 1, h - 1), Hp);
        const bool vT = y >= 1, vB = y + 1 < h;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            // Y = 2y: upsampled rows 2y-1 (outside for y == 0), 2y, 2y+1;   Y = 2y+1: rows 2y, 2y+1, 2y+2 (outside for y == h-1)
            f32x4 o0 = (0.25f * Hm[1][b] + 0.75f * H0[1][b]) + (0.75f * H0[2][b] + 0.25f * Hp[2][b]);
            const f32x4 t0 = 0.75f * Hm[0][b] + 0.25f * H0[0][b];
            o0 = vT ? o0 + t0 : o0;
            f32x4 o1 = (0.25f * Hm[0][b] + 0.75f * H0[0][b]) + (0.75f * H0[1][b] + 0.25f * Hp[1][b]);
            const f32x4 t1 = 0.25f * H0[2][b] + 0.75f * Hp[2][b];
            o1 = vB ? o1 + t1 : o1;
            o0 = o0 * sc + bi;
            o1 = o1 * sc + bi;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o0[e] = up_act(o0[e], act, slope); o1[e] = up_act(o1[e], act, slope); }
            *reinterpret_cast<f32x4 *>(on + ((size_t)(2 * y) * (2 * w) + 2 * x + b) * out_ld) = o0;
            *reinterpret_cast<f32x4 *>(on + ((size_t)(2 * y + 1) * (2 * w) + 2 * x + b) * out_ld) = o1;
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int b = 0; b < 2; ++b) { Hm[ky][b] = H0[ky][b]; H0[ky][b] = Hp[ky][b]; }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same decomposition with the tap planes kept on chip, for 64 input channels (PSPUpsample `up_3`, 64 -> 64 on the largest map): at
// K = 64 the tap GEMM is memory-bound if z goes through HBM (9 x Cout fp32 values per low-resolution pixel written and re-read).  Here a
// workgroup owns 8 x 14 low-resolution pixels (16 x 28 outputs): it stages the 10 x 16 pixel patch (+1 halo, coordinates clamped to the
// image = ATen's clamped taps) split into hi/lo fp16 halves once, and then per group of 16 output channels
//   1. computes z[160 pixels][9 taps x 16 channels] on the matrix cores (v_mfma_f32_16x16x32_f16, the f16x3 scheme: a_lo.b_hi + a_hi.b_lo +
//      a_hi.b_hi, fp32 accumulate; one patch row = one 16-row M block; the weight fragments come straight from L2 in MFMA operand
//      order -- the packed split weights of PackedConv.taps() are that order) into LDS,
//   2. runs the gather of up2_tap_gather_kernel on the LDS-resident planes and stores the 16-channel slices of the 2 x 2 output blocks.
// Executed multiplies: 2.25 x (160 / 112) per output pixel and channel pair instead of 9.
// Status (round 2): correct (tests/test_gpu_ops.py::test_conv2d_fused_upsample) and on par with the patch-resident direct kernel (371 vs
// 373 us for up_3 of an 11-frame batch, 129 vs 139 us for the keyframe), not ahead of it: with the planes filling LDS there is one wave
// per SIMD, so the phases of a tile run in turn -- per tile (52 k cycles; -DFT_TIMING stamps) staging 13 %, MFMAs + plane stores 37 %
// (25 cycles per 16x16x32 MFMA with all four SIMDs issuing, against 17 for one), gather 39 % (VALU-bound: ~1000 instructions per thread
// and channel group), barriers 7 %.  It is therefore an opt-in route (ops: ARSEG_CONV_UP2_FUSED=1); what it needs is producer / consumer
// waves on double-buffered 8-channel planes so that the matrix pipe and the VALU overlap.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int FT_IH = 8, FT_IW = 14, FT_PH = FT_IH + 2, FT_PW = FT_IW + 2, FT_NPX = FT_PH * FT_PW;      // 160 patch pixels
constexpr int FT_ROWA = 272;                      // bytes per patch pixel: 2 x [32 hi | 32 lo] halves + 16 pad
constexpr int FT_ZLD = 144;                       // floats per z row: 9 taps x 16 channels; 144 = 16 mod 64: the gather's ds_read_b128 lane groups
                                                  // (4 pixels x 4 channel quads) hit 64 distinct banks, the MFMA phase's b128 stores are issue-bound anyway
constexpr int FT_A_BYTES = FT_NPX * FT_ROWA, FT_Z_BYTES = FT_NPX * FT_ZLD * 4;      // 43,520 + 94,720
constexpr int FT_MAXCO = 128;                     // per-row factors, scale and bias of up to this many output channels sit in LDS
constexpr int FT_SMEM = FT_A_BYTES + FT_Z_BYTES + (9 + 2) * FT_MAXCO * 4;

struct FusedParams {
    const float *in; const void *w9; const float *s9, *scale, *bias; float *out;
    int in_ld, out_ld, N, h, w, Cout, tiles_x, tiles_y, act; float slope;
};

#ifdef FT_TIMING
__device__ unsigned long long g_ft_dbg[16];
#define FT_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (blockIdx.x == 7 && threadIdx.x == 0) atomicAdd(&g_ft_dbg[i], t_ - tprev_); tprev_ = t_; } while (0)
#else
#define FT_STAMP(i)
#endif
__global__ __launch_bounds__(256) void upconv_fused64_kernel(const FusedParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
#ifdef FT_TIMING
    unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
#endif
    unsigned char *As = fsm;
    float *Zs = reinterpret_cast<float *>(fsm + FT_A_BYTES);
    float *S9 = reinterpret_cast<float *>(fsm + FT_A_BYTES + FT_Z_BYTES), *Sc = S9 + 9 * FT_MAXCO, *Bi = Sc + FT_MAXCO;
    // (one workgroup per CU: every global load whose result is needed at once is an exposed round trip -- constants go to LDS up front)
    for (int i = threadIdx.x; i < 9 * p.Cout; i += 256) S9[i] = p.s9[i];
    for (int i = threadIdx.x; i < p.Cout; i += 256) { Sc[i] = p.scale ? p.scale[i] : 1.0f; Bi[i] = p.bias ? p.bias[i] : 0.0f; }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x;
    const int tx = bid % p.tiles_x; bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int n = bid / p.tiles_y;
    const int iy0 = ty * FT_IH, ix0 = tx * FT_IW;
    const float *inb = p.in + (size_t)n * p.h * p.w * p.in_ld;

    // ---- patch -> LDS, split.  item = (patch pixel, 16-byte piece of 4 channels): 160 x 16 items, 10 per thread
#pragma unroll
    for (int it = 0; it < 10; ++it) {
        const int i = tid + it * 256, px = i >> 4, piece = i & 15, pr = px / FT_PW, pc = px - pr * FT_PW;
        const int gy = min(max(iy0 - 1 + pr, 0), p.h - 1), gx = min(max(ix0 - 1 + pc, 0), p.w - 1);
        const f32x4 v = *reinterpret_cast<const f32x4 *>(inb + ((size_t)gy * p.w + gx) * p.in_ld + piece * 4);
        unsigned h01, h23, l01, l23;
        arseg_split_f16(v, h01, h23, l01, l23);
        unsigned char *row = As + px * FT_ROWA + (piece >> 3) * 128 + (piece & 7) * 8;
        *reinterpret_cast<uint2 *>(row) = uint2{h01, h23};
        *reinterpret_cast<uint2 *>(row + 64) = uint2{l01, l23};
    }
    __syncthreads();
    FT_STAMP(0);

    // ---- MFMA roles: wave = (mh, ng); M blocks (= patch rows) 5*mh .. 5*mh+4, taps ng ? 5..8 : 0..4
    const int mh = wave >> 1, ng = wave & 1, t_lo = ng ? 5 : 0, t_n = ng ? 4 : 5;
    const int l16 = lane & 15, lq = lane >> 4;
    h16x8 ah[5][2], al[5][2];
#pragma unroll
    for (int b = 0; b < 5; ++b)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned char *a = As + ((5 * mh + b) * 16 + l16) * FT_ROWA + ks * 128 + lq * 16;
            ah[b][ks] = *reinterpret_cast<const h16x8 *>(a);
            al[b][ks] = *reinterpret_cast<const h16x8 *>(a + 64);
        }
    const unsigned char *w9 = reinterpret_cast<const unsigned char *>(p.w9);      // [9*Cout rows][2 K tiles][32 hi | 32 lo halves] = 256 bytes per row

    // ---- gather roles: thread = (co quad cg, inner column gx_, row pair rg): 4 x 14 x 4 = 224 threads
    const int cg = tid & 3, gq = tid >> 2, gx_ = gq % FT_IW, rg = gq / FT_IW;          // rg 0..3 active (gq < 56)
    const int x = ix0 + gx_, pcx = gx_ + 1;                                            // image column, patch column
    const bool g_on = gq < 4 * FT_IW && x < p.w;
    // taps that fall outside the upsampled image carry coefficient 0 (the planes hold finite values of in-image pixels there)
    const float cL75 = x >= 1 ? 0.75f : 0.0f, cL25 = x >= 1 ? 0.25f : 0.0f, cR75 = x + 1 < p.w ? 0.75f : 0.0f, cR25 = x + 1 < p.w ? 0.25f : 0.0f;
    const float a_slope = p.act == ARSEG_ACT_PRELU ? p.slope : 1.0f, a_lo = p.act == ARSEG_ACT_RELU ? 0.0f : -INFINITY;
    const bool a_sig = p.act == ARSEG_ACT_SIGMOID;

    // weight fragments of one channel group: [tap of this wave][K step] x {hi, lo}; the next group's are requested before the gather
    u32x4 bcur[5][4], bnxt[5][4];
    auto load_b = [&](int co0, u32x4 (&B)[5][4]) {
#pragma unroll
        for (int ti = 0; ti < 5; ++ti) {
            const int t = min(t_lo + ti, 8);
            const unsigned char *r = w9 + (size_t)(t * p.Cout + co0 + l16) * 256 + lq * 16;
            B[ti][0] = *reinterpret_cast<const u32x4 *>(r);
            B[ti][1] = *reinterpret_cast<const u32x4 *>(r + 64);
            B[ti][2] = *reinterpret_cast<const u32x4 *>(r + 128);
            B[ti][3] = *reinterpret_cast<const u32x4 *>(r + 192);
        }
    };
    load_b(0, bcur);
    FT_STAMP(1);

    for (int co0 = 0; co0 < p.Cout; co0 += 16) {
        // ---- 1. z = patch x taps for 16 output channels.  Weights are the MFMA's A operand (rows = channels), the patch its B operand
        // (columns = pixels): a lane's four results are four consecutive channels of one pixel = one 16-byte store.  The stores of a
        // tap are issued under the MFMAs of the next one.
        f32x4 pend[5];
        int pend_t = -1;
        auto flush = [&]() {
            const f32x4 s9 = *reinterpret_cast<const f32x4 *>(S9 + pend_t * p.Cout + co0 + lq * 4);
#pragma unroll
            for (int b = 0; b < 5; ++b) *reinterpret_cast<f32x4 *>(Zs + ((5 * mh + b) * 16 + l16) * FT_ZLD + pend_t * 16 + lq * 4) = pend[b] * s9;
        };
#pragma unroll
        for (int ti = 0; ti < 5; ++ti) {
            if (ti < t_n) {
                const h16x8 fbh0 = __builtin_bit_cast(h16x8, bcur[ti][0]), fbl0 = __builtin_bit_cast(h16x8, bcur[ti][1]);
                const h16x8 fbh1 = __builtin_bit_cast(h16x8, bcur[ti][2]), fbl1 = __builtin_bit_cast(h16x8, bcur[ti][3]);
                f32x4 acc[5];
#pragma unroll
                for (int b = 0; b < 5; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int b = 0; b < 5; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbh0, al[b][0], acc[b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < 5; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbl0, ah[b][0], acc[b], 0, 0, 0);
                if (ti > 0) flush();
#pragma unroll
                for (int b = 0; b < 5; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbh0, ah[b][0], acc[b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < 5; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbh1, al[b][1], acc[b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < 5; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbl1, ah[b][1], acc[b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < 5; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbh1, ah[b][1], acc[b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < 5; ++b) pend[b] = acc[b];
                pend_t = t_lo + ti;
            }
        }
        flush();
        FT_STAMP(2);
        __syncthreads();
        // The next group's weight fragments are requested here and consumed BEFORE the gather's global stores are issued: stores count
        // in vmcnt too, and with one wave per SIMD a later wait for the fragments would also wait for the stores' acknowledgements.
        if (co0 + 16 < p.Cout) load_b(co0 + 16, bnxt);
        FT_STAMP(3);

        // ---- 2. gather from the LDS planes (the arithmetic of up2_tap_gather_kernel as FMA chains; patch rows / columns hold the clamped pixels)
        f32x4 gres[2][2][2];
        if (g_on) {
            const float *zc = Zs + cg * 4;
            auto hrow = [&](int pr, f32x4 (&H)[3][2]) {
                const float *zr = zc + (pr * FT_PW + pcx - 1) * FT_ZLD;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    f32x4 a[3][3];
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int j = 0; j < 3; ++j) a[kx][j] = *reinterpret_cast<const f32x4 *>(zr + j * FT_ZLD + (ky * 3 + kx) * 16);
                    f32x4 e = 0.25f * a[1][0];
                    e = 0.75f * a[1][1] + e; e = 0.75f * a[2][1] + e; e = 0.25f * a[2][2] + e; e = cL75 * a[0][0] + e; e = cL25 * a[0][1] + e;
                    H[ky][0] = e;
                    f32x4 o = 0.25f * a[0][0];
                    o = 0.75f * a[0][1] + o; o = 0.75f * a[1][1] + o; o = 0.25f * a[1][2] + o; o = cR25 * a[2][1] + o; o = cR75 * a[2][2] + o;
                    H[ky][1] = o;
                }
            };
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(Sc + co0 + cg * 4), bi = *reinterpret_cast<const f32x4 *>(Bi + co0 + cg * 4);
            f32x4 Hm[3][2], H0[3][2], Hp[3][2];
            const int r0 = 2 * rg;                       // first inner row of this thread; patch row r0 + 1
            hrow(r0, Hm);
            hrow(r0 + 1, H0);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int y = iy0 + r0 + k;
                hrow(r0 + k + 2, Hp);
                if (y < p.h) {
                    const float cT75 = y >= 1 ? 0.75f : 0.0f, cT25 = y >= 1 ? 0.25f : 0.0f, cB75 = y + 1 < p.h ? 0.75f : 0.0f, cB25 = y + 1 < p.h ? 0.25f : 0.0f;
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        f32x4 o0 = 0.25f * Hm[1][b];
                        o0 = 0.75f * H0[1][b] + o0; o0 = 0.75f * H0[2][b] + o0; o0 = 0.25f * Hp[2][b] + o0; o0 = cT75 * Hm[0][b] + o0; o0 = cT25 * H0[0][b] + o0;
                        f32x4 o1 = 0.25f * Hm[0][b];
                        o1 = 0.75f * H0[0][b] + o1; o1 = 0.75f * H0[1][b] + o1; o1 = 0.25f * Hp[1][b] + o1; o1 = cB25 * H0[2][b] + o1; o1 = cB75 * Hp[2][b] + o1;
                        o0 = o0 * sc + bi;
                        o1 = o1 * sc + bi;
                        if (a_sig) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { o0[e] = 1.0f / (1.0f + __expf(-o0[e])); o1[e] = 1.0f / (1.0f + __expf(-o1[e])); }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                o0[e] = fmaxf(o0[e] >= 0.0f ? o0[e] : o0[e] * a_slope, a_lo);
                                o1[e] = fmaxf(o1[e] >= 0.0f ? o1[e] : o1[e] * a_slope, a_lo);
                            }
                        }
                        gres[k][b][0] = o0;
                        gres[k][b][1] = o1;
                    }
                }
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int b = 0; b < 2; ++b) { Hm[ky][b] = H0[ky][b]; H0[ky][b] = Hp[ky][b]; }
            }
        }
#pragma unroll
        for (int ti = 0; ti < 5; ++ti)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bcur[ti][q] = bnxt[ti][q];
                asm volatile("" : "+v"(bcur[ti][q]));          // the fragments have landed here, ahead of the stores
            }
        if (g_on) {
            const int r0 = 2 * rg;
            float *on = p.out + (size_t)n * (2 * p.h) * (2 * p.w) * p.out_ld + co0 + cg * 4;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int y = iy0 + r0 + k;
                if (y < p.h) {
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        *reinterpret_cast<f32x4 *>(on + ((size_t)(2 * y) * (2 * p.w) + 2 * x + b) * p.out_ld) = gres[k][b][0];
                        *reinterpret_cast<f32x4 *>(on + ((size_t)(2 * y + 1) * (2 * p.w) + 2 * x + b) * p.out_ld) = gres[k][b][1];
                    }
                }
            }
        }
        FT_STAMP(4);
        __syncthreads();          // the next channel group overwrites the planes
        FT_STAMP(5);
    }
}

}  // namespace

#ifdef FT_TIMING
extern "C" int arseg__ft_dbg(unsigned long long *host16, int reset) {
    if (reset) { unsigned long long z[16] = {}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ft_dbg), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_ft_dbg), 16 * sizeof(unsigned long long));
}
#endif

extern "C" int arseg_upconv3x3_fused_fwd(const float *in, int in_ld, const void *w9_h3, const float *s9, const float *scale, const float *bias,
                                         float *out, int out_ld, int N, int h, int w, int Cin, int Cout, int act, float prelu_slope,
                                         arseg_stream_t stream) {
    ARSEG_CHECK_PTR(in); ARSEG_CHECK_PTR(w9_h3); ARSEG_CHECK_PTR(s9); ARSEG_CHECK_PTR(out);
    ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(h); ARSEG_CHECK_POS(w); ARSEG_CHECK_POS(Cout);
    if (Cin != 64 || (Cout & 15) || Cout > FT_MAXCO) return ARSEG_EUNSUPPORTED;
    if (in_ld < Cin || (in_ld & 3) || out_ld < Cout || (out_ld & 3)) return ARSEG_EINVAL;
    if (!ARSEG_ALIGNED16(in) || !ARSEG_ALIGNED16(out) || !ARSEG_ALIGNED16(w9_h3) || (scale && !ARSEG_ALIGNED16(scale)) || (bias && !ARSEG_ALIGNED16(bias)))
        return ARSEG_EINVAL;
    FusedParams p;
    p.in = in; p.w9 = w9_h3; p.s9 = s9; p.scale = scale; p.bias = bias; p.out = out;
    p.in_ld = in_ld; p.out_ld = out_ld; p.N = N; p.h = h; p.w = w; p.Cout = Cout; p.act = act; p.slope = prelu_slope;
    p.tiles_x = (w + FT_IW - 1) / FT_IW; p.tiles_y = (h + FT_IH - 1) / FT_IH;
    const long long grid = (long long)N * p.tiles_x * p.tiles_y;
    if (grid >= (1ll << 31)) return ARSEG_EUNSUPPORTED;
    static ArsegSmemAttr attr;
    if (int e = arseg_allow_smem(attr, reinterpret_cast<const void *>(upconv_fused64_kernel), FT_SMEM)) return e;
    hipLaunchKernelGGL(upconv_fused64_kernel, dim3((unsigned)grid), dim3(256), FT_SMEM, arseg_stream(stream), p);
    return arseg_launch_status();
}

extern "C" int arseg_upconv3x3_tap_gather_fwd(const float *z, int z_ld, const float *scale, const float *bias, float *out, int out_ld, int N,
                                              int h, int w, int Cout, int act, float prelu_slope, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(z); ARSEG_CHECK_PTR(out); ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(h); ARSEG_CHECK_POS(w); ARSEG_CHECK_POS(Cout);
    if ((Cout & 3) || z_ld < 9 * Cout || (z_ld & 3) || out_ld < Cout || (out_ld & 3)) return ARSEG_EINVAL;
    if (!ARSEG_ALIGNED16(z) || !ARSEG_ALIGNED16(out) || (scale && !ARSEG_ALIGNED16(scale)) || (bias && !ARSEG_ALIGNED16(bias))) return ARSEG_EINVAL;
    // strip length: the longest (fewest re-computed boundary rows: (rs + 2) / rs) that still gives every SIMD a couple of waves
    const long long per_strip = (long long)N * w * (Cout / 4);
    int rs = 16;
    while (rs > 2 && per_strip * ((h + rs - 1) / rs) < 2048ll * 64) rs >>= 1;
    const long long total = per_strip * ((h + rs - 1) / rs);
    if ((total + 255) / 256 >= (1ll << 31)) return ARSEG_EUNSUPPORTED;
    hipLaunchKernelGGL(up2_tap_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, arseg_stream(stream), z, z_ld, scale, bias,
                       out, out_ld, N, h, w, Cout, rs, act, prelu_slope);
    return arseg_launch_status();
}
