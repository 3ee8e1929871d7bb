// conv2d (+ folded BatchNorm / bias, + residual, + activation) on 16-bit tensors: the bf16 / fp16 storage configurations of
// BASELINE.json (configs[2]: BiSeNet-18 bf16, configs[4]: BiSeNet-18 0.3x fp16).  Replaces the nn.Conv2d / BatchNorm2d / ReLU stacks of
// model/bisenet.py:31-60,162-399 (and every other conv of the path) when the network runs with 16-bit activations and weights.
//
// Implicit GEMM, one v_mfma_f32_32x32x16_{f16,bf16} per 16-deep product, fp32 accumulation, fp32 epilogue, one rounding to 16 bits at
// the store.  The GEMM is computed transposed, D[co][pixel] = sum_k W[co][k] * X[pixel][k] (A = weights, B = activations): both
// operands are then K-contiguous rows in memory ([Cout][Kpad] weights, NHWC pixels), i.e. exactly the MFMA A / B fragment (a lane's 8
// consecutive k), read from LDS with one ds_read_b128.
//   * tile: CO_T (64 | 128) output channels x 128 pixels, 4 waves, K step 32, double-buffered LDS, register-staged prefetch of the next
//     K step under the MFMAs of the current one, one barrier per K step;
//   * im2col by address arithmetic through buffer descriptors: a tap outside the image / a row past M / the K padding get an
//     out-of-range offset and load zeros (no branches); Cin is a multiple of 8 (a 16-byte piece never straddles a filter tap), RGB frames are
//     ingested as NHWC8;
//   * epilogue through LDS: the fp32 accumulators are transposed into [pixel][co] rows so that scale / bias / residual / activation
//     run on 8 consecutive channels and the store is a coalesced 16-byte vector of a full NHWC row.
#include "arseg_common.h"
#ifndef PATCH_ABL
#define PATCH_ABL 0         // dev builds (tools/bench_patch16.py): 1 = no epilogue, 2 = one K step instead of nchunk * 9, 4 = fragments read once (no LDS reads in the loop)
#endif
#ifndef STEM_ABL
#define STEM_ABL 0          // dev builds (tools/bench_stem16.py): 1 = no output stores, 2 = one K step instead of 25, 4 = the patch is loaded once
#endif

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct Conv16Params {
    const uint16_t *in, *w, *res;
    const float *scale, *bias;
    uint16_t *out;
    int N, H, W, Cin, in_ld, Ho, Wo, Cout, out_ld, res_ld;
    int R, S, stride, pad, dil;
    int K, Kpad, M, act;
    float slope;
    int tiles_co, tiles_px;
    unsigned in_bytes, w_bytes;
    int patch_tw, patch_l2tw, tiles_m;      // patch-resident 3x3 kernel: tile width (a power of two), its log2, pixel tiles
    int nsplit, kt_per_split;        // split-K: blockIdx.y = K slice, fp32 partial sums to `ws` [nsplit][M][Cout], epilogue in the reduce kernel
    float *ws;
    int log2Cin, inv_S;              // (r6) Cin = 1 << log2Cin (or -1), 65536 / S + 1: the K -> (tap, channel) decode of the implicit GEMM without divisions
};

constexpr int PIX_T = 128, KPAD = 64;                  // Kpad16 is a multiple of 64 (both K steps divide it)
constexpr unsigned OOB = 0x80000000u;

template <bool BF>
__device__ __forceinline__ f32x16 mfma16(const u32x4 a, const u32x4 b, const f32x16 c) {
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}

// activation as arithmetic on two uniform parameters (no switch per element: the epilogues apply it to 32-64 values per lane, and the four-way
// branch per value made them as long as the MFMAs of a short-K tile): none / relu / prelu = max(v >= 0 ? v : v * s, lo) with (s, lo) = (1, -inf) /
// (1, 0) / (slope, -inf); sigmoid keeps its (uniform) branch
__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    if (act == ARSEG_ACT_SIGMOID) return 1.0f / (1.0f + __expf(-v));
    const float s = act == ARSEG_ACT_PRELU ? slope : 1.0f, lo = act == ARSEG_ACT_RELU ? 0.0f : -INFINITY;
    return fmaxf(v >= 0.0f ? v : v * s, lo);
}

template <bool BF, int CO_T, int BK>      // BK: K step (32 | 64 halves); LDS rows carry 8 halves of padding (ds_read_b128 of 32 rows conflict free)
__global__ __launch_bounds__(256) void conv16_kernel(const Conv16Params p) {
    constexpr int LDK = BK + 8, CPR = BK / 8, RPP = 256 / CPR;      // 16-byte pieces per row, rows staged per pass of the workgroup
    constexpr int RB = PIX_T / RPP;                                 // pixel rows staged per thread
    constexpr int WPX = CO_T == 128 ? 64 : 32;          // pixels per wave (CO_T = 128: 2 x 2 waves of 64 x 64; 64: 1 x 4 waves of 64 x 32)
    constexpr int TPX = WPX / 32;                       // 32-pixel MFMA tiles per wave
    constexpr int RA = CO_T / RPP;                      // weight rows staged per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *As = reinterpret_cast<uint16_t *>(smem);                       // [2][CO_T][LDK]
    uint16_t *Bs = As + 2 * CO_T * LDK;                                      // [2][PIX_T][LDK]
    float *Ot = reinterpret_cast<float *>(smem);                             // epilogue: [64 or 128 pixels][CO_T + 4] fp32 (aliases the staging)
    constexpr int OLD = CO_T + 4;

    // XCD-aware (bijective) remap of the linear block id: XCD x gets a contiguous chunk of tiles
    const int nblk = p.tiles_co * p.tiles_px;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_px = bid / p.tiles_co, tile_co = bid - tile_px * p.tiles_co;      // co fastest: neighbours share the pixel rows
    const int px0 = tile_px * PIX_T, co0 = tile_co * CO_T;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(p.in), 0, (int)p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(p.w), 0, (int)p.w_bytes, 0x00020000);

    const int tid = threadIdx.x;
    const int chunk = tid % CPR, row0 = tid / CPR;       // 16-byte piece of the K step; rows row0 + RPP * i

    // the output pixels whose rows this thread stages (constant over the K loop)
    int iy0[RB], ix0[RB], rowoff[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int m = px0 + row0 + RPP * i;
        if (m < p.M) {
            const int hw = p.Ho * p.Wo;
            const int n = m / hw, rem = m - n * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            iy0[i] = oy * p.stride - p.pad;
            ix0[i] = ox * p.stride - p.pad;
            rowoff[i] = ((n * p.H + iy0[i]) * p.W + ix0[i]) * p.in_ld * 2;      // bytes; may be negative
        } else {
            iy0[i] = -(1 << 28); ix0[i] = 0; rowoff[i] = 0;                     // every tap fails the bounds test
        }
    }
    unsigned woff[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int co = co0 + row0 + RPP * i;
        woff[i] = co < p.Cout ? (unsigned)(co * p.Kpad) * 2u + chunk * 16u : OOB;
    }

    struct Regs { u32x4 a[RA], b[RB]; };
    auto load = [&](int kt, Regs &r) {
        // k -> (tap, channel) -> (filter row, column).  (r6) Two integer divisions by run-time values per thread and K step (~50 VALU) stood beside 4-8
        // MFMAs per wave and step; with Cin a power of two >= BK the tap is UNIFORM over the workgroup (a K step never straddles a tap) and
        // comes from scalar shifts, the filter row from a multiply-high (exact for taps < 64)
        const int k = kt * BK + chunk * 8;
        int tap, ci;
        if (p.log2Cin >= 0 && (1 << p.log2Cin) >= BK) {
            const int kb = kt * BK;
            tap = kb >> p.log2Cin; ci = (kb & ((1 << p.log2Cin) - 1)) + chunk * 8;
        } else {
            tap = k / p.Cin; ci = k - tap * p.Cin;
        }
        const int fr = (tap * p.inv_S) >> 16, fs = tap - fr * p.S;
        const int dy = fr * p.dil, dx = fs * p.dil;
        const int tapoff = ((dy * p.W + dx) * p.in_ld + ci) * 2;
        const bool kok = k < p.K;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const bool ok = kok && (unsigned)(iy0[i] + dy) < (unsigned)p.H && (unsigned)(ix0[i] + dx) < (unsigned)p.W;
            r.b[i] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, ok ? (unsigned)(rowoff[i] + tapoff) : OOB, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < RA; ++i) r.a[i] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, woff[i] + (unsigned)kt * (BK * 2u), 0, 0);
    };
    auto store = [&](int buf, const Regs &r) {
#pragma unroll
        for (int i = 0; i < RA; ++i) *reinterpret_cast<u32x4 *>(As + (buf * CO_T + row0 + RPP * i) * LDK + chunk * 8) = r.a[i];
#pragma unroll
        for (int i = 0; i < RB; ++i) *reinterpret_cast<u32x4 *>(Bs + (buf * PIX_T + row0 + RPP * i) * LDK + chunk * 8) = r.b[i];
    };

    const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wco0 = CO_T == 128 ? (wave >> 1) * 64 : 0;
    const int wpx0 = CO_T == 128 ? (wave & 1) * 64 : wave * 32;
    f32x16 acc[2][TPX];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TPX; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int kt0 = blockIdx.y * p.kt_per_split, ktiles = min(p.Kpad / BK, kt0 + p.kt_per_split);      // this K slice (all of K without split-K)
    Regs rg;
    load(kt0, rg);
    store(kt0 & 1, rg);
    __syncthreads();
    for (int kt = kt0; kt < ktiles; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < ktiles) load(kt + 1, rg);
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            u32x4 a[2], b[TPX];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const u32x4 *>(As + (buf * CO_T + wco0 + 32 * i + li) * LDK + kk * 16 + lh * 8);
#pragma unroll
            for (int j = 0; j < TPX; ++j) b[j] = *reinterpret_cast<const u32x4 *>(Bs + (buf * PIX_T + wpx0 + 32 * j + li) * LDK + kk * 16 + lh * 8);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TPX; ++j) acc[i][j] = mfma16<BF>(a[i], b[j], acc[i][j]);
        }
        if (kt + 1 < ktiles) store(buf ^ 1, rg);
        __syncthreads();
    }

    // ---- epilogue: D[co][px] -> LDS [px][co] fp32 (CO_T = 128: one 64-pixel half at a time) -> scale, bias, residual, activation -> 16 bit
    constexpr int NPASS = CO_T == 128 ? 2 : 1, PPX = PIX_T / NPASS;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        if (NPASS == 1 || (wave & 1) == ps) {
            const int pbase = NPASS == 1 ? wpx0 : 0;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TPX; ++j)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int co = wco0 + 32 * i + 8 * q4 + 4 * lh, px = pbase + 32 * j + li;
                        *reinterpret_cast<f32x4 *>(Ot + px * OLD + co) = f32x4{acc[i][j][4 * q4], acc[i][j][4 * q4 + 1], acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3]};
                    }
        }
        __syncthreads();
        constexpr int ITEMS = PPX * (CO_T / 8);          // 8-channel vectors of this pass
        for (int it = tid; it < ITEMS; it += 256) {
            const int px = it / (CO_T / 8), c8 = it - px * (CO_T / 8);
            const int m = px0 + ps * PPX + px, co = co0 + c8 * 8;
            if (p.nsplit > 1) {                       // split-K: raw fp32 partial sums, the epilogue runs in conv16_splitk_reduce_kernel
                if (m < p.M && co < p.Cout) {
                    float *dst = p.ws + ((size_t)blockIdx.y * p.M + m) * p.Cout + co;
                    *reinterpret_cast<f32x4 *>(dst) = *reinterpret_cast<const f32x4 *>(Ot + px * OLD + c8 * 8);
                    *reinterpret_cast<f32x4 *>(dst + 4) = *reinterpret_cast<const f32x4 *>(Ot + px * OLD + c8 * 8 + 4);
                }
                continue;
            }
            if (m < p.M && co < p.Cout) {
                float v[8];
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(Ot + px * OLD + c8 * 8), v1 = *reinterpret_cast<const f32x4 *>(Ot + px * OLD + c8 * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = v0[e]; v[4 + e] = v1[e]; }
                const int nco = min(8, p.Cout - co);
                u32x4 rres = {0, 0, 0, 0};
                float sc[8], bi[8];
                if (nco == 8) {          // the common case: whole 8-channel vectors of scale / bias / residual
                    const f32x4 s0 = p.scale ? *reinterpret_cast<const f32x4 *>(p.scale + co) : f32x4{1.f, 1.f, 1.f, 1.f};
                    const f32x4 s1 = p.scale ? *reinterpret_cast<const f32x4 *>(p.scale + co + 4) : f32x4{1.f, 1.f, 1.f, 1.f};
                    const f32x4 b0 = p.bias ? *reinterpret_cast<const f32x4 *>(p.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
                    const f32x4 b1 = p.bias ? *reinterpret_cast<const f32x4 *>(p.bias + co + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { sc[e] = s0[e]; sc[4 + e] = s1[e]; bi[e] = b0[e]; bi[4 + e] = b1[e]; }
                    if (p.res) rres = *reinterpret_cast<const u32x4 *>(p.res + (size_t)m * p.res_ld + co);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int c = min(co + e, p.Cout - 1);
                        sc[e] = p.scale ? p.scale[c] : 1.0f; bi[e] = p.bias ? p.bias[c] : 0.0f;
                    }
                }
                uint16_t o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float x = v[e] * sc[e] + bi[e];
                    if (p.res) {
                        const uint16_t rb = nco == 8 ? (uint16_t)((e & 1) ? rres[e >> 1] >> 16 : rres[e >> 1] & 0xffffu) : p.res[(size_t)m * p.res_ld + min(co + e, p.Cout - 1)];
                        x += arseg_h2f<BF>(rb);
                    }
                    o[e] = arseg_f2h<BF>(act_apply(x, p.act, p.slope));
                }
                uint16_t *dst = p.out + (size_t)m * p.out_ld + co;
                if (nco == 8) *reinterpret_cast<u32x4 *>(dst) = u32x4{o[0] | ((unsigned)o[1] << 16), o[2] | ((unsigned)o[3] << 16), o[4] | ((unsigned)o[5] << 16), o[6] | ((unsigned)o[7] << 16)};
                else
                    for (int e = 0; e < nco; ++e) dst[e] = o[e];
            }
        }
        if (ps + 1 < NPASS) __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Patch-resident kernel for 3x3 stride-1 pad == dil convs with Cin % 64 == 0 (the 16-bit twin of conv3x3_patch_kernel in
// conv_igemm.hip): a workgroup owns a TH x TW pixel tile of ONE image and BN output channels and, per 64-channel chunk, stages the
// (TH+2d) x (TW+2d) input patch once (128 bytes per pixel); the nine taps are A-fragment address offsets into it.  Only the weight tile
// (BN x 64 halves) streams per tap, requested two taps ahead (two register stages + LDS double buffer).  The implicit-GEMM kernel
// above re-fetches the activation slice of every tap from L2 and has 16 MFMAs per wave between barriers; here the activations are
// read from L2 once instead of nine times.  D[pixel][co] (A = pixels, B = weights), C/D layout: co = lane & 31.
template <bool BF, int BN, int WM>      // WM wave rows of 64 output pixels each: BM = 64*WM pixels, 2*WM waves
__global__ __launch_bounds__(128 * WM, (BN == 64 ? 4 : 2)) void conv16_patch_kernel(const Conv16Params p) {
    constexpr int ROWB = 144;                          // bytes per LDS row: 64 halves + pad (conflict-free b128 reads)
    constexpr int NT = 128 * WM, BM = 64 * WM;
    constexpr int TN = BN / 64, TM = 2, RB = (BN * 8 + NT - 1) / NT, RPB = NT / 8;
    constexpr int MAXI = WM == 2 ? 9 : 7;              // patch items (pixel, 16-byte piece) per thread: <= 288 / 448 pixels
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int TW = p.patch_tw, log2TW = p.patch_l2tw;
    const int TH = BM >> log2TW, d = p.dil, PW = TW + 2 * d, PH = TH + 2 * d, npx = PW * PH;
    unsigned char *Ps = smem;                                                  // [npx][ROWB]
    unsigned char *Bs = Ps + ((npx * ROWB + 255) & ~255);                        // [2][BN][ROWB]

    const int tiles_x = (p.Wo + TW - 1) >> log2TW, tiles_y = (p.Ho + TH - 1) / TH;
    const int nblk = p.tiles_m * p.tiles_co;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid / p.tiles_co, tile_n = bid - tile_m * p.tiles_co;
    const int img = tile_m / (tiles_x * tiles_y), trem = tile_m - img * (tiles_x * tiles_y);
    const int ty0 = (trem / tiles_x) * TH, tx0 = (trem - (trem / tiles_x) * tiles_x) * TW, n0 = tile_n * BN;

    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(p.in), 0, (int)p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(p.w), 0, (int)p.w_bytes, 0x00020000);
    const int tid = threadIdx.x;

    unsigned poff[MAXI];                               // item i = (pixel i/8, piece i%8): global byte offset without the chunk term, or OOB
#pragma unroll
    for (int it = 0; it < MAXI; ++it) {
        const int i = tid + it * NT, px = i >> 3, py = px / PW, pxx = px - py * PW;
        const int gy = ty0 - d + py, gx = tx0 - d + pxx;
        const bool ok = px < npx && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        poff[it] = ok ? (unsigned)((((img * p.H + gy) * p.W + gx) * p.in_ld) * 2 + (i & 7) * 16) : OOB;
    }
    const int chunk = tid & 7, row0 = tid >> 3;
    unsigned woff[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int n = n0 + row0 + RPB * i;
        woff[i] = (n < p.Cout && row0 + RPB * i < BN) ? (unsigned)(n * p.Kpad + chunk * 8) * 2u : OOB;
    }
    const int nchunk = p.Cin >> 6;

    struct BRegs { u32x4 v[RB]; };
    u32x4 rp[MAXI];
    auto load_patch = [&](int ck) {
#pragma unroll
        for (int it = 0; it < MAXI; ++it) rp[it] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, poff[it] + (unsigned)(ck * 128), 0, 0);
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int it = 0; it < MAXI; ++it) {
            const int i = tid + it * NT, px = i >> 3;
            if (px < npx) *reinterpret_cast<u32x4 *>(Ps + px * ROWB + (i & 7) * 16) = rp[it];
        }
    };
    auto load_b = [&](int ck, int tap, BRegs &r) {
        const unsigned kt = (unsigned)(tap * nchunk + ck) * 128u;
#pragma unroll
        for (int i = 0; i < RB; ++i) r.v[i] = __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, woff[i] + kt, 0, 0);
    };
    auto store_b = [&](int buf, const BRegs &r) {
#pragma unroll
        for (int i = 0; i < RB; ++i)
            if (row0 + RPB * i < BN) *reinterpret_cast<u32x4 *>(Bs + (buf * BN + row0 + RPB * i) * ROWB + chunk * 16) = r.v[i];
    };

    const int wave = tid >> 6, lane = tid & 63, wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    int arow[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int m = wm * 64 + tm * 32 + li, ty = m >> log2TW, tx = m & (TW - 1);
        arow[tm] = (ty * PW + tx) * ROWB + lh * 16;
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const int nsteps = (PATCH_ABL & 2) ? 1 : nchunk * 9;                     // K steps s = (chunk ck, tap): s = 9*ck + tap
    BRegs r0, r1;
    load_patch(0);
    load_b(0, 0, r0);
    store_patch();
    store_b(0, r0);
    if (nsteps > 1) load_b(0, 1, r0);
    __syncthreads();
    int cur = 0, ck = 0, tap = 0;              // of the step being computed
    int ck2 = 0, tap2 = 2;                     // of the step two ahead
    auto step = [&](int s_, BRegs &ld, BRegs &st) {
        if (s_ + 2 < nsteps) load_b(ck2, tap2, ld);
        if (tap == 0 && ck + 1 < nchunk) load_patch(ck + 1);
        const int r3 = (tap * 11) >> 5, s3 = tap - r3 * 3;                 // tap = 3*r3 + s3
        const int toff = (r3 * d * PW + s3 * d) * ROWB;
        const unsigned char *bh = Bs + (cur * BN + wn * (BN / 2) + li) * ROWB + lh * 16;
        u32x4 fa[4][TM], fb[4][TN];                  // all fragments of the tap first, then the MFMAs back to back
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int t = 0; t < TM; ++t) fa[ks][t] = *reinterpret_cast<const u32x4 *>(Ps + arow[t] + toff + ks * 32);
#pragma unroll
            for (int t = 0; t < TN; ++t) fb[ks][t] = *reinterpret_cast<const u32x4 *>(bh + t * 32 * ROWB + ks * 32);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma16<BF>(fa[ks][tm], fb[ks][tn], acc[tm][tn]);
        if (s_ + 1 < nsteps) store_b(cur ^ 1, st);
        __syncthreads();
        cur ^= 1;
        if (tap == 8 && ck + 1 < nchunk) {     // everybody is past the last tap of this chunk: the patch may be replaced
            store_patch();
            __syncthreads();
        }
        if (++tap == 9) { tap = 0; ++ck; }
        if (++tap2 == 9) { tap2 = 0; ++ck2; }
    };
#pragma unroll 1
    for (int s_ = 0; s_ < nsteps; s_ += 2) {
        step(s_, r1, r0);
        if (s_ + 1 < nsteps) step(s_ + 1, r0, r1);
    }

    if ((PATCH_ABL & 1) && acc[0][0][0] != 12345.678f) return;
    // epilogue through LDS (the patch and the weight tiles are dead: the last step ended with a barrier): 128 pixels at a time the fp32
    // accumulators are laid out [pixel][co] so that scale / bias / residual / activation run on 8 consecutive channels and the store is a
    // 16-byte vector of an NHWC row.  C/D layout of the 32x32 MFMA: col = lane&31 (co), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel).
    float *Ot = reinterpret_cast<float *>(smem);
    constexpr int OLD = BN + 4;
#pragma unroll 1
    for (int half = 0; half < BM / 128; ++half) {
        if ((wm >> 1) == half) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int pl = (wm & 1) * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        Ot[pl * OLD + wn * (BN / 2) + tn * 32 + li] = acc[tm][tn][r];
                    }
        }
        __syncthreads();
        for (int it = tid; it < 128 * (BN / 8); it += NT) {
            const int pl = it / (BN / 8), c8 = it - pl * (BN / 8);
            const int ml = half * 128 + pl, oy = ty0 + (ml >> log2TW), ox = tx0 + (ml & (TW - 1)), co = n0 + c8 * 8;
            if (oy >= p.Ho || ox >= p.Wo || co >= p.Cout) continue;
            const size_t m = ((size_t)img * p.Ho + oy) * p.Wo + ox;
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(Ot + pl * OLD + c8 * 8), v1 = *reinterpret_cast<const f32x4 *>(Ot + pl * OLD + c8 * 8 + 4);
            const int nco = min(8, p.Cout - co);
            uint16_t o[8];
            if (nco == 8) {
                const f32x4 s0 = p.scale ? *reinterpret_cast<const f32x4 *>(p.scale + co) : f32x4{1.f, 1.f, 1.f, 1.f};
                const f32x4 s1 = p.scale ? *reinterpret_cast<const f32x4 *>(p.scale + co + 4) : f32x4{1.f, 1.f, 1.f, 1.f};
                const f32x4 b0 = p.bias ? *reinterpret_cast<const f32x4 *>(p.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
                const f32x4 b1 = p.bias ? *reinterpret_cast<const f32x4 *>(p.bias + co + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                u32x4 rres = {0, 0, 0, 0};
                if (p.res) rres = *reinterpret_cast<const u32x4 *>(p.res + m * p.res_ld + co);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float x = (e < 4 ? v0[e] : v1[e - 4]) * (e < 4 ? s0[e] : s1[e - 4]) + (e < 4 ? b0[e] : b1[e - 4]);
                    if (p.res) x += arseg_h2f<BF>((uint16_t)((e & 1) ? rres[e >> 1] >> 16 : rres[e >> 1] & 0xffffu));
                    o[e] = arseg_f2h<BF>(act_apply(x, p.act, p.slope));
                }
                *reinterpret_cast<u32x4 *>(p.out + m * p.out_ld + co) =
                    u32x4{o[0] | ((unsigned)o[1] << 16), o[2] | ((unsigned)o[3] << 16), o[4] | ((unsigned)o[5] << 16), o[6] | ((unsigned)o[7] << 16)};
            } else {
                for (int e = 0; e < nco; ++e) {
                    float x = (e < 4 ? v0[e] : v1[e - 4]) * (p.scale ? p.scale[co + e] : 1.0f) + (p.bias ? p.bias[co + e] : 0.0f);
                    if (p.res) x += arseg_h2f<BF>(p.res[m * p.res_ld + co + e]);
                    p.out[m * p.out_ld + co + e] = arseg_f2h<BF>(act_apply(x, p.act, p.slope));
                }
            }
        }
        __syncthreads();
    }
}

template <bool BF, int BN, int WM>
int launch_patch16(const Conv16Params &p, hipStream_t st) {
    const int th = 64 * WM / p.patch_tw, npx = (th + 2 * p.dil) * (p.patch_tw + 2 * p.dil);
    const size_t stage = (size_t)((npx * 144 + 255) & ~255) + (size_t)2 * BN * 144, epi = (size_t)128 * (BN + 4) * 4;
    const size_t smem = stage > epi ? stage : epi;
    static ArsegSmemAttr attr;
    if (int e = arseg_allow_smem(attr, reinterpret_cast<const void *>(conv16_patch_kernel<BF, BN, WM>), smem)) return e;
    hipLaunchKernelGGL((conv16_patch_kernel<BF, BN, WM>), dim3(p.tiles_m * p.tiles_co), dim3(128 * WM), smem, st, p);
    return arseg_launch_status();
}
template <bool BF>
int launch_patch16_cfg(const Conv16Params &p, int cfg, hipStream_t st) {
    switch (cfg) {
        case 5: return launch_patch16<BF, 64, 2>(p, st);
        case 6: return launch_patch16<BF, 128, 2>(p, st);
        case 7: case 10: case 11: return launch_patch16<BF, 64, 4>(p, st);
        case 13: return launch_patch16<BF, 64, 2>(p, st);
        default: return launch_patch16<BF, 128, 4>(p, st);      // 8, 12
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Stem kernel: 7x7 stride-2 pad-3 conv of an NHWC8 frame to 64 channels (+ folded BN + activation) -- conv1 of the ResNet-18 context
// path and of the spatial path (model/bisenet.py:31-60,109-116 via extractors.py:114).  The implicit-GEMM kernel spends 200 us per
// 11 frames on it (K = 49 taps x 8 channels, every activation fetched ~12 times from L2, 64-channel tiles).  Here a persistent workgroup
// keeps ALL weights in LDS (64 x 49 taps x 16 bytes) and per 8 x 32 output tile stages the 21 x 69 pixel input patch once; one MFMA K
// step of 16 = two taps (lane half lh picks the tap), fragments are single ds_read_b128s: weights [co][tap] rows padded to an odd
// multiple of 16 bytes, the patch split into even / odd column planes so that the stride-2 pixel walk of a wave is contiguous.
// D[co][pixel] (A = weights): a lane owns one output pixel and 4 consecutive channels per accumulator group -> 8-byte stores.
template <bool BF>
__global__ __launch_bounds__(256, 2) void conv16_stem_kernel(const Conv16Params p) {
    constexpr int TH = 8, TW = 32, PH = 2 * TH + 5, PW = 2 * TW + 5, PLN = (PW + 1) / 2, PROW = 2 * PLN;     // 21 x 69 patch, 35-piece planes
    constexpr int WROW = 51;                                  // weight row stride in 16-byte pieces (49 taps + pad: odd)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4 *Wl = reinterpret_cast<u32x4 *>(smem);                        // [64][WROW]
    u32x4 *Pl = Wl + 64 * WROW;                                         // [PH][2][PLN]
    f32x4 *SBl = reinterpret_cast<f32x4 *>(Pl + PH * PROW);              // [16] scale | [16] bias (global loads in the epilogue would be
                                                                        // 16 exposed round trips per tile)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(p.in), 0, (int)p.in_bytes, 0x00020000);
    for (int i = tid; i < 64 * 50; i += 256) {                // weights once per workgroup (tap 49 = the zero K padding)
        const int co = i / 50, t = i - co * 50;
        Wl[co * WROW + t] = *reinterpret_cast<const u32x4 *>(p.w + (size_t)co * p.Kpad + t * 8);
    }
    if (tid < 128) reinterpret_cast<float *>(SBl)[tid] = tid < 64 ? (p.scale ? p.scale[tid] : 1.0f) : (p.bias ? p.bias[tid - 64] : 0.0f);
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH, ntiles = p.N * tiles_x * tiles_y;
    constexpr int NIT = (PH * PW + 255) / 256;                // patch items (pixel = one 16-byte piece) per thread
    u32x4 pre[NIT];                                           // the NEXT tile's patch travels in registers under the MFMAs of the current one
    auto patch_load = [&](int tile_) {
        const int img_ = tile_ / (tiles_x * tiles_y), tr_ = tile_ - img_ * (tiles_x * tiles_y);
        const int py0 = 2 * (tr_ / tiles_x) * TH - 3, px0 = 2 * (tr_ - (tr_ / tiles_x) * tiles_x) * TW - 3;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256, row = i / PW, c = i - row * PW, gy = py0 + row, gx = px0 + c;
            const bool ok = i < PH * PW && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            pre[it] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, ok ? (unsigned)(((img_ * p.H + gy) * p.W + gx) * p.in_ld) * 2u : OOB, 0, 0);
        }
    };
    auto patch_store = [&]() {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256, row = i / PW, c = i - row * PW;
            if (i < PH * PW) Pl[row * PROW + (c & 1) * PLN + (c >> 1)] = pre[it];
        }
    };
    if ((int)blockIdx.x < ntiles) patch_load(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int img = tile / (tiles_x * tiles_y), trem = tile - img * (tiles_x * tiles_y);
        const int ty0 = (trem / tiles_x) * TH, tx0 = (trem - (trem / tiles_x) * tiles_x) * TW;
        __syncthreads();                                      // the previous tile's patch is no longer read (and the weights are in place)
        patch_store();
        __syncthreads();
        if (!(STEM_ABL & 4) && tile + (int)gridDim.x < ntiles) patch_load(tile + gridDim.x);
        f32x16 acc[2][2];                                     // [co tile][pixel row of the wave]
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
#pragma unroll 5
        for (int j = 0; j < ((STEM_ABL & 2) ? 1 : 25); ++j) {
            const int t = 2 * j + lh, tc = min(t, 48), r = (tc * 37) >> 8, sx = tc - 7 * r;      // tap (r, sx); t == 49 multiplies zero weights
            u32x4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = Wl[(32 * i + li) * WROW + t];
#pragma unroll
            for (int i = 0; i < 2; ++i) b[i] = Pl[(2 * (2 * wave + i) + r) * PROW + (sx & 1) * PLN + li + (sx >> 1)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int k = 0; k < 2; ++k) acc[i][k] = mfma16<BF>(a[i], b[k], acc[i][k]);
        }
        // epilogue: lane = pixel (column li of output row 2*wave + k), accumulator rows = channels (r&3) + 8*(r>>2) + 4*lh (+32 i).
        // (r6, tools/bench_stem16.py with -DSTEM_ABL: of 124 us per 11-frame batch the MFMA loop was 40 and this epilogue 59 -- its ARITHMETIC, not its
        // 8-byte stores: a software bf16 rounding (~8 VALU per value) and a four-way activation switch per value.  With v_cvt_pk_bf16_f32 and the
        // branch-free activation: 106-108 us.  Measured and not kept: the same rows laid out in LDS and stored as contiguous 16-byte pieces (108.4 against
        // 109-111 us: the stores were never the cost), barriers that leave the stores in flight (no change).)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int oy = ty0 + 2 * wave + k, ox = tx0 + li;
            if (oy >= p.Ho || ox >= p.Wo) continue;
            if ((STEM_ABL & 1) && acc[0][k][0] != 12345.678f) continue;
            uint16_t *dst = p.out + (((size_t)img * p.Ho + oy) * p.Wo + ox) * p.out_ld;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int co = 32 * i + 8 * q4 + 4 * lh;
                    const f32x4 sc = SBl[co >> 2], bi = SBl[16 + (co >> 2)];
                    uint16_t o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = arseg_f2h<BF>(act_apply(acc[i][k][4 * q4 + e] * sc[e] + bi[e], p.act, p.slope));
                    *reinterpret_cast<u32x2 *>(dst + co) = u32x2{o[0] | ((unsigned)o[1] << 16), o[2] | ((unsigned)o[3] << 16)};
                }
        }
    }
}

template <bool BF>
int launch_stem16(const Conv16Params &p, hipStream_t st) {
    const size_t smem = (size_t)(64 * 51 + 21 * 70 + 32) * 16;
    static ArsegSmemAttr attr;
    if (int e = arseg_allow_smem(attr, reinterpret_cast<const void *>(conv16_stem_kernel<BF>), smem)) return e;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const long long ntiles = (long long)p.N * arseg_cdiv(p.Ho, 8) * arseg_cdiv(p.Wo, 32);
    const int grid = (int)(ntiles < 2 * cus ? ntiles : 2 * cus);
    hipLaunchKernelGGL((conv16_stem_kernel<BF>), dim3(grid), dim3(256), smem, st, p);
    return arseg_launch_status();
}

// sums the split-K partials in slice order and applies the epilogue (scale, bias, residual, activation, one rounding to 16 bits);
// 8 channels per thread (Cout % 8 == 0 with split-K)
template <bool BF>
__global__ __launch_bounds__(256) void conv16_splitk_reduce_kernel(const Conv16Params p) {
    const int c8n = p.Cout >> 3;
    const long long total = (long long)p.M * c8n;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(idx / c8n), co = (int)(idx - (long long)m * c8n) * 8;
        const float *src = p.ws + (size_t)m * p.Cout + co;
        f32x4 a = *reinterpret_cast<const f32x4 *>(src), b = *reinterpret_cast<const f32x4 *>(src + 4);
        for (int z = 1; z < p.nsplit; ++z) {
            a += *reinterpret_cast<const f32x4 *>(src + (size_t)z * p.M * p.Cout);
            b += *reinterpret_cast<const f32x4 *>(src + (size_t)z * p.M * p.Cout + 4);
        }
        u32x4 rres = {0, 0, 0, 0};
        if (p.res) rres = *reinterpret_cast<const u32x4 *>(p.res + (size_t)m * p.res_ld + co);
        uint16_t o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = (e < 4 ? a[e] : b[e - 4]) * (p.scale ? p.scale[co + e] : 1.0f) + (p.bias ? p.bias[co + e] : 0.0f);
            if (p.res) x += arseg_h2f<BF>((uint16_t)((e & 1) ? rres[e >> 1] >> 16 : rres[e >> 1] & 0xffffu));
            o[e] = arseg_f2h<BF>(act_apply(x, p.act, p.slope));
        }
        *reinterpret_cast<u32x4 *>(p.out + (size_t)m * p.out_ld + co) =
            u32x4{o[0] | ((unsigned)o[1] << 16), o[2] | ((unsigned)o[3] << 16), o[4] | ((unsigned)o[5] << 16), o[6] | ((unsigned)o[7] << 16)};
    }
}

template <bool BF, int CO_T, int BK>
int launch(const Conv16Params &p, hipStream_t st) {
    const size_t stage = (size_t)2 * (CO_T + PIX_T) * (BK + 8) * 2, epi = (size_t)(CO_T == 128 ? 64 : 128) * (CO_T + 4) * 4;
    const size_t smem = stage > epi ? stage : epi;
    static ArsegSmemAttr attr;
    if (int e = arseg_allow_smem(attr, reinterpret_cast<const void *>(conv16_kernel<BF, CO_T, BK>), smem)) return e;
    Conv16Params q = p;
    q.kt_per_split = arseg_cdiv(p.Kpad / BK, p.nsplit);
    q.nsplit = arseg_cdiv(p.Kpad / BK, q.kt_per_split);            // no empty slices
    hipLaunchKernelGGL((conv16_kernel<BF, CO_T, BK>), dim3(p.tiles_co * p.tiles_px, q.nsplit), dim3(256), smem, st, q);
    if (int e = arseg_launch_status()) return e;
    if (q.nsplit > 1) {
        long long blocks = ((long long)p.M * (p.Cout >> 3) + 255) / 256;
        hipLaunchKernelGGL((conv16_splitk_reduce_kernel<BF>), dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, st, q);
        return arseg_launch_status();
    }
    return ARSEG_OK;
}
template <bool BF>
int launch_cfg(const Conv16Params &p, bool wide, bool deep, hipStream_t st) {
    if (wide) return deep ? launch<BF, 128, 64>(p, st) : launch<BF, 128, 32>(p, st);
    return deep ? launch<BF, 64, 64>(p, st) : launch<BF, 64, 32>(p, st);
}

}  // namespace

namespace {
// split-K slices of a launch: explicit (desc.split_k >= 1) or, with 0, chosen so that a launch whose tiles do not fill the chip and
// whose K loop is long gets ~2 workgroups per CU (the 16x32-map layers of BiSeNet-18: 176 tiles, K = 4608)
int conv16_nsplit(const arseg_conv_desc *d, long long M, int Kpad, int co_t) {
    if ((d->Cout & 7) || d->tile_cfg >= 5) return 1;          // (the patch-resident and stem plans have no split-K)
    const int kt64 = Kpad / 64;
    int ns = d->split_k;
    if (ns <= 0) {
        const long long tiles = (long long)arseg_cdiv(d->Cout, co_t) * arseg_cdiv(M, PIX_T);
        ns = 1;
        while (tiles * ns < 384 && ns < 8 && kt64 / (ns * 2) >= 6) ns *= 2;
    }
    if (ns > kt64) ns = kt64;
    return ns < 1 ? 1 : ns;
}
bool conv16_wide(const arseg_conv_desc *d, long long M) {
    const int cfg = d->tile_cfg;
    if (cfg == 0) return d->Cout > 64 && (long long)arseg_cdiv(d->Cout, 128) * arseg_cdiv(M, PIX_T) >= 512;
    return cfg == 2 || cfg == 4;
}
}  // namespace

extern "C" size_t arseg_conv2d16_workspace_bytes(const arseg_conv_desc *d) {
    int Ho, Wo;
    if (!d || arseg_conv_out_hw(d, &Ho, &Wo) != ARSEG_OK) return 0;
    const long long M = (long long)d->N * Ho * Wo;
    const int Kpad = (d->R * d->S * d->Cin + KPAD - 1) / KPAD * KPAD;
    const int ns = conv16_nsplit(d, M, Kpad, conv16_wide(d, M) ? 128 : 64);
    return ns > 1 ? (size_t)ns * M * d->Cout * sizeof(float) : 0;
}

extern "C" int arseg_conv2d16_fwd(const arseg_conv_desc *d, int dtype, const void *in, const void *w_packed16, const float *scale,
                                  const float *bias, const void *residual, void *out, void *workspace, size_t workspace_bytes,
                                  arseg_stream_t stream) {
    if (!d) return ARSEG_EINVAL;
    ARSEG_CHECK_PTR(in); ARSEG_CHECK_PTR(w_packed16); ARSEG_CHECK_PTR(out);
    ARSEG_CHECK_POS(d->N); ARSEG_CHECK_POS(d->H); ARSEG_CHECK_POS(d->W); ARSEG_CHECK_POS(d->Cin); ARSEG_CHECK_POS(d->Cout);
    ARSEG_CHECK_POS(d->R); ARSEG_CHECK_POS(d->S); ARSEG_CHECK_POS(d->stride); ARSEG_CHECK_POS(d->dil);
    if (dtype != ARSEG_DT_F16 && dtype != ARSEG_DT_BF16) return ARSEG_EINVAL;
    if ((d->Cin & 7) || (d->in_ld & 7) || d->in_ld < d->Cin) return ARSEG_EINVAL;
    if ((d->out_ld & 7) || d->out_ld < d->Cout || (residual && ((d->res_ld & 7) || d->res_ld < d->Cout))) return ARSEG_EINVAL;
    if (!ARSEG_ALIGNED16(in) || !ARSEG_ALIGNED16(w_packed16) || !ARSEG_ALIGNED16(out) || (residual && !ARSEG_ALIGNED16(residual))) return ARSEG_EINVAL;
    if (d->batch > 1) return ARSEG_EUNSUPPORTED;
    int Ho, Wo;
    if (int e = arseg_conv_out_hw(d, &Ho, &Wo)) return e;
    Conv16Params p;
    p.in = (const uint16_t *)in; p.w = (const uint16_t *)w_packed16; p.res = (const uint16_t *)residual; p.scale = scale; p.bias = bias;
    p.out = (uint16_t *)out;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.in_ld = d->in_ld; p.Ho = Ho; p.Wo = Wo; p.Cout = d->Cout; p.out_ld = d->out_ld;
    p.res_ld = d->res_ld; p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
    p.K = d->R * d->S * d->Cin; p.Kpad = (p.K + KPAD - 1) / KPAD * KPAD; p.act = d->act; p.slope = d->prelu_slope;
    p.log2Cin = -1;
    for (int b = 3; b < 16; ++b) if (d->Cin == (1 << b)) p.log2Cin = b;
    p.inv_S = 65536 / d->S + 1;
    if (d->R * d->S > 64) return ARSEG_EUNSUPPORTED;          // (the multiply-high filter-row decode is exact for taps < 64: up to 7 x 7 and 8 x 8)
    const long long M = (long long)d->N * Ho * Wo;
    const size_t in_bytes = (size_t)d->N * d->H * d->W * d->in_ld * 2, w_bytes = (size_t)d->Cout * p.Kpad * 2;
    if (M >= (1ll << 31) || in_bytes >= (1ull << 31) || w_bytes >= (1ull << 31)) return ARSEG_EUNSUPPORTED;       // 32-bit buffer offsets
    p.M = (int)M; p.in_bytes = (unsigned)in_bytes; p.w_bytes = (unsigned)w_bytes;
    // tile_cfg: 0 auto; 1 / 2 = 64- / 128-channel tile with K step 32; 3 / 4 = the same with K step 64
    const int cfg = d->tile_cfg;
    if (cfg < 0 || cfg > 13) return ARSEG_EINVAL;
    if (cfg == 9) {          // stem kernel: 7x7 stride-2 pad-3, NHWC8 -> 64 channels, no residual
        if (d->R != 7 || d->S != 7 || d->stride != 2 || d->pad != 3 || d->dil != 1 || d->Cin != 8 || d->Cout != 64 || residual || d->split_k > 1)
            return ARSEG_EUNSUPPORTED;
        p.patch_tw = 0; p.patch_l2tw = 0; p.tiles_m = 0; p.tiles_co = 1; p.tiles_px = 0; p.nsplit = 1; p.kt_per_split = 0; p.ws = nullptr;
        hipStream_t st = arseg_stream(stream);
        return dtype == ARSEG_DT_BF16 ? launch_stem16<true>(p, st) : launch_stem16<false>(p, st);
    }
    if (cfg >= 5) {          // patch-resident 3x3 kernel: 5 / 6 = 128-pixel tiles with 64 / 128 output channels, 7 / 8 = 256-pixel tiles;
        // (r6) 10 / 11 / 12 / 13 = squarer tiles (less halo: 8 x 32 = 340 patch pixels, 16 x 16 = 324, against 4 x 64 = 396 for the same 256 outputs):
        // 10 = 256 pixels as 8 x 32, 64 channels; 11 = 16 x 16, 64 channels; 12 = 8 x 32, 128 channels; 13 = 128 pixels as 8 x 16, 64 channels
        if (d->R != 3 || d->S != 3 || d->stride != 1 || d->pad != d->dil || (d->Cin & 63) || d->split_k > 1) return ARSEG_EUNSUPPORTED;
        const int bm = (cfg == 5 || cfg == 6 || cfg == 13) ? 128 : 256;
        int tw = Wo >= 48 ? 64 : (Wo >= 24 ? 32 : 16);
        if (cfg == 10 || cfg == 12) { if (tw <= 32) return ARSEG_EUNSUPPORTED; tw = 32; }        // (the same tile as 7 / 8 on a narrower map: not a new plan)
        if (cfg == 11 || cfg == 13) { if (tw <= 16) return ARSEG_EUNSUPPORTED; tw = 16; }
        const int th = bm / tw;
        if ((th + 2 * d->dil) * (tw + 2 * d->dil) > (bm == 128 ? 288 : 448)) return ARSEG_EUNSUPPORTED;
        p.patch_tw = tw; p.patch_l2tw = tw == 64 ? 6 : (tw == 32 ? 5 : 4);
        p.tiles_m = d->N * arseg_cdiv(Ho, th) * arseg_cdiv(Wo, tw);
        p.tiles_co = arseg_cdiv(d->Cout, (cfg == 6 || cfg == 8 || cfg == 12) ? 128 : 64); p.tiles_px = 0;
        p.nsplit = 1; p.kt_per_split = 0; p.ws = nullptr;
        hipStream_t st = arseg_stream(stream);
        return dtype == ARSEG_DT_BF16 ? launch_patch16_cfg<true>(p, cfg, st) : launch_patch16_cfg<false>(p, cfg, st);
    }
    p.patch_tw = 0; p.patch_l2tw = 0; p.tiles_m = 0;
    // 128-channel tiles when they still give every CU a few workgroups, K step 64 (half the barriers, 74 KB of LDS) for long K loops
    const bool wide = conv16_wide(d, M), deep = cfg == 0 ? p.K >= 512 : cfg >= 3;
    const int co_t = wide ? 128 : 64;
    if (d->split_k < 0) return ARSEG_EINVAL;
    p.nsplit = conv16_nsplit(d, M, p.Kpad, co_t);
    p.kt_per_split = 0; p.ws = reinterpret_cast<float *>(workspace);
    if (p.nsplit > 1) {
        if (!workspace || workspace_bytes < (size_t)p.nsplit * M * d->Cout * sizeof(float)) return ARSEG_EWORKSPACE;
        if (!ARSEG_ALIGNED16(workspace)) return ARSEG_EINVAL;
    }
    p.tiles_co = arseg_cdiv(d->Cout, co_t); p.tiles_px = arseg_cdiv(M, PIX_T);
    hipStream_t st = arseg_stream(stream);
    return dtype == ARSEG_DT_BF16 ? launch_cfg<true>(p, wide, deep, st) : launch_cfg<false>(p, wide, deep, st);
}
