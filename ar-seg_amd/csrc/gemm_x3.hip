// Batched GEMM on operands that are ALREADY split into fp16 (hi, lo) pairs in HBM (the f16x3 arithmetic of conv_igemm.hip:
// a.b = a_hi.b_hi + a_hi.b_lo + a_lo.b_hi on v_mfma_f32_16x16x32_f16, fp32 accumulate).
//
//   C[b][m][n] = act(scale[n] * sum_k X[b][m][k] * W[b][n][k] + bias[n])
//
// Operand layout ("split rows", the LDS row layout of conv_igemm.hip's f16x3 back end moved to HBM): a row of K values is K/32
// groups of 128 bytes, each = 32 hi halves then 32 lo halves.  The same 4 bytes per value as fp32, so a producer that writes its
// output this way (arseg_wino43_input_fwd with split = 1, arseg_split_rows_fwd) costs the consumer nothing -- and the consumer no
// longer stages through registers: there is no conversion left to do on the way into LDS.
//
// Why a second GEMM kernel.  conv_igemm_kernel register-stages its operands (global -> VGPR -> split -> ds_write), two barriers or
// two register stages per 32-deep K step: MFMA pipe 25-38 % busy.  Here
//   * operands go HBM/L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction), no VGPRs, no VALU;
//   * the LDS image is the linear image the DMA writes (8 rows x 128 B per instruction); bank conflicts of the fragment reads are
//     removed by permuting the SOURCE 16-byte slots of a row (slot ^ ((row >> 1) & 7)) and applying the same XOR on the read: the
//     four 16-lane groups of a ds_read_b128 then touch 16 distinct slots of the 256-byte bank row;
//   * tiles of 128 x 128 ... 256 x 256 per workgroup of 8 or 16 waves (tile_cfg, timed per shape), one 128-byte group per K step, double
//     buffered, ONE barrier per K step; tile_cfg 6 runs the 16 waves of a 256 x 256 tile as two groups half a K step apart (see STAG below):
//     while one group issues MFMAs the other reads fragments and requests the next step (-12 % on the largest GEMM);
//   * D is computed transposed (weights as the MFMA's A operand) so that a lane holds 4 consecutive output channels: 16-byte stores.
#include "arseg_common.h"

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct GX3Params {
    const unsigned char *X, *W;
    const unsigned char *X2, *W2;     // K concatenation: K steps >= K / 32 read these (K2 values per row), e.g. [features | interpolation matrix] x
    int K2;                           //   [bottleneck weights ; per-image pyramid terms]: out = x.w^T + x2.w2^T  (0: none)
    unsigned x2_bytes, w2_bytes;
    long long x2_bs, w2_bs;
    float *C;
    const float *scale, *bias, *res;
    int M, N, K, ldc, res_ld, out_split;
    unsigned x_bytes, w_bytes;        // extent of one problem's operands (buffer descriptors)
    long long x_bs, w_bs, c_bs;       // batch strides: bytes, bytes, floats
    int tiles_m, tiles_n, batch;
    int act;
    float slope;
    unsigned *range_flag;             // out_split: set when a value written as a split row exceeds range_limit (arseg_conv_desc.range_flag)
    float range_limit;
};

__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p; }
__device__ __forceinline__ u32x4 make_rsrc(const void *base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu)),
                 (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
}
// LDS[lds_base + lane*16 .. +15] <- buffer[voff .. +15]
__device__ __forceinline__ void dma16_buf(const u32x4 rsrc, unsigned voff, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_base), "v"(voff), "s"(rsrc) : "memory");
}

// NWM x NWN waves; 16-row fragments per wave: WTM along M (activation rows), WTN along N (weight rows)
template <int NWM, int NWN, int WTM, int WTN, int ABL = 0, bool STAG = false>
__global__ __launch_bounds__(64 * NWM * NWN) void gemm_x3_kernel(const GX3Params p) {
    constexpr int NW = NWM * NWN, BM = 16 * NWM * WTM, BN = 16 * NWN * WTN;
    constexpr int XB = BM * 128, WB = BN * 128, STAGE = XB + WB;
    constexpr int XI = BM / (8 * NW), WI = BN / (8 * NW);  // DMA instructions per wave and K step (8 rows each)
    static_assert(XI >= 1 && WI >= 1 && XI * 8 * NW == BM && WI * 8 * NW == BN, "tile rows must divide over the waves in 8-row pieces");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    const int per = p.tiles_m * p.tiles_n, nblk = per * p.batch;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int b = bid / per, rem = bid - b * per;
    const int tile_m = rem / p.tiles_n, tile_n = rem - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const u32x4 x_rsrc = make_rsrc(p.X + (size_t)b * p.x_bs, p.x_bytes);
    const u32x4 w_rsrc = make_rsrc(p.W + (size_t)b * p.w_bs, p.w_bytes);
    const unsigned ld = (unsigned)p.K * 4u, ld2 = (unsigned)p.K2 * 4u;
    const u32x4 x2_rsrc = make_rsrc(p.K2 ? p.X2 + (size_t)b * p.x2_bs : p.X, p.K2 ? p.x2_bytes : 0u);
    const u32x4 w2_rsrc = make_rsrc(p.K2 ? p.W2 + (size_t)b * p.w2_bs : p.W, p.K2 ? p.w2_bytes : 0u);
    const int k1 = p.K >> 5;

    // DMA: lane (r8, pos) of instruction j fills LDS row wave*rows + 8j + r8, 16-byte position pos, with source slot pos ^ g(row)
    const int r8 = lane >> 3, pos = lane & 7;
    unsigned xsrc[XI], wsrc[WI], xsrc2[XI], wsrc2[WI];
#pragma unroll
    for (int j = 0; j < XI; ++j) {
        const int row = wave * (BM / NW) + j * 8 + r8;
        const unsigned gr = (unsigned)min(m0 + row, p.M - 1), sl = (unsigned)((pos ^ ((row >> 1) & 7)) << 4);
        xsrc[j] = gr * ld + sl;
        xsrc2[j] = gr * ld2 + sl;
    }
#pragma unroll
    for (int j = 0; j < WI; ++j) {
        const int row = wave * (BN / NW) + j * 8 + r8;
        const unsigned gr = (unsigned)min(n0 + row, p.N - 1), sl = (unsigned)((pos ^ ((row >> 1) & 7)) << 4);
        wsrc[j] = gr * ld + sl;
        wsrc2[j] = gr * ld2 + sl;
    }
    const unsigned lds0 = lds_addr(smem);
    auto issue_x = [&](int kt, int st) {
        const bool second = kt >= k1;                  // (uniform) the K steps of the concatenated second operand pair
        const unsigned kb = (unsigned)(second ? kt - k1 : kt) * 128u, base = lds0 + (unsigned)st * STAGE;
        const u32x4 rs = second ? x2_rsrc : x_rsrc;
#pragma unroll
        for (int j = 0; j < XI; ++j) dma16_buf(rs, (second ? xsrc2[j] : xsrc[j]) + kb, base + (unsigned)(wave * (BM / NW) + j * 8) * 128u);
    };
    auto issue_w = [&](int kt, int st) {
        const bool second = kt >= k1;
        const unsigned kb = (unsigned)(second ? kt - k1 : kt) * 128u, base = lds0 + (unsigned)st * STAGE;
        const u32x4 rs = second ? w2_rsrc : w_rsrc;
#pragma unroll
        for (int j = 0; j < WI; ++j) dma16_buf(rs, (second ? wsrc2[j] : wsrc[j]) + kb, base + XB + (unsigned)(wave * (BN / NW) + j * 8) * 128u);
    };

    // fragment reads: lane (i16, kq) reads row i16 of a 16-row fragment, source slot kq (hi) / 4 + kq (lo) -> position slot ^ (i16 >> 1)
    const int i16 = lane & 15, kq = lane >> 4;
    const unsigned xo = (unsigned)(wm * (BM / NWM) + i16) * 128u + (unsigned)((kq ^ (i16 >> 1)) << 4);
    const unsigned wo = XB + (unsigned)(wn * (BN / NWN) + i16) * 128u + (unsigned)((kq ^ (i16 >> 1)) << 4);

    f32x4 acc[WTM][WTN];
#pragma unroll
    for (int a = 0; a < WTM; ++a)
#pragma unroll
        for (int c = 0; c < WTN; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int PRIO = (ABL & 4) ? 1 : 0;
    constexpr bool NOREAD = (ABL & 2) != 0, NODMA = (ABL & 1) != 0;
    constexpr int HM = WTM >= 4 ? WTM / 2 : WTM;         // activation fragments held at a time
    auto mfmas = [&](int h, const h16x8 (&wh)[WTN], const h16x8 (&wl)[WTN], const h16x8 (&xh)[HM], const h16x8 (&xl)[HM]) {
#pragma unroll
        for (int a = 0; a < HM; ++a)
#pragma unroll
            for (int c = 0; c < WTN; ++c) acc[h * HM + a][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[c], xh[a], acc[h * HM + a][c], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < HM; ++a)
#pragma unroll
            for (int c = 0; c < WTN; ++c) acc[h * HM + a][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c], xl[a], acc[h * HM + a][c], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < HM; ++a)
#pragma unroll
            for (int c = 0; c < WTN; ++c) acc[h * HM + a][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c], xh[a], acc[h * HM + a][c], 0, 0, 0);
    };
    h16x8 pwh[WTN], pwl[WTN], pxh[HM], pxl[HM];           // (ablation builds: fragments read once)
    auto compute = [&](int st, int kt_next) {       // kt_next >= 0: the next K step's operands are requested into the other buffer on the way
        const unsigned char *base = smem + st * STAGE;
        if constexpr (NOREAD) {
#pragma unroll
            for (int h = 0; h < WTM / HM; ++h) {
                if (kt_next >= 0) { if (h == 0) issue_x(kt_next, st ^ 1); else if (h == WTM / HM - 1) issue_w(kt_next, st ^ 1); }
                mfmas(h, pwh, pwl, pxh, pxl);
            }
            return;
        }
        h16x8 wh[WTN], wl[WTN];
#pragma unroll
        for (int c = 0; c < WTN; ++c) {
            wh[c] = *reinterpret_cast<const h16x8 *>(base + wo + c * 2048);
            wl[c] = *reinterpret_cast<const h16x8 *>(base + (wo ^ 64u) + c * 2048);
        }
#pragma unroll
        for (int h = 0; h < WTM / HM; ++h) {
            h16x8 xh[HM], xl[HM];
#pragma unroll
            for (int a = 0; a < HM; ++a) {
                xh[a] = *reinterpret_cast<const h16x8 *>(base + xo + (h * HM + a) * 2048);
                xl[a] = *reinterpret_cast<const h16x8 *>(base + (xo ^ 64u) + (h * HM + a) * 2048);
            }
            // the DMA requests go out behind the fragment reads of a half, in front of its MFMAs: a request costs the issuing wave 60-180
            // cycles during which the other waves of the SIMD own the matrix pipe
            if (kt_next >= 0) {
                if (h == 0) issue_x(kt_next, st ^ 1);
                if (h == WTM / HM - 1) issue_w(kt_next, st ^ 1);
            }
            __builtin_amdgcn_s_setprio(PRIO);
            mfmas(h, wh, wl, xh, xl);
            __builtin_amdgcn_s_setprio(0);
        }
    };

    const int nk = k1 + (p.K2 >> 5);
    issue_x(0, 0);
    issue_w(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (NOREAD) {
#pragma unroll
        for (int c = 0; c < WTN; ++c) {
            pwh[c] = *reinterpret_cast<const h16x8 *>(smem + wo + c * 2048);
            pwl[c] = *reinterpret_cast<const h16x8 *>(smem + (wo ^ 64u) + c * 2048);
        }
#pragma unroll
        for (int a = 0; a < HM; ++a) {
            pxh[a] = *reinterpret_cast<const h16x8 *>(smem + xo + a * 2048);
            pxl[a] = *reinterpret_cast<const h16x8 *>(smem + (xo ^ 64u) + a * 2048);
        }
    }
    if constexpr (STAG) {
        // Two wave groups half a K step apart (each SIMD hosts two waves of either group): a K step is read(half 0) | MFMA(half 0) | read(half 1) |
        // MFMA(half 1), separated by barriers, and group 1 starts one barrier late -- while one group issues MFMAs the other reads fragments and
        // requests the next K step, instead of all 16 waves doing the same thing at the same time.  Hand-over rules: fragment reads are retired
        // (lgkmcnt) before the barrier that ends their segment, so a stage may be refilled by whoever passes that barrier; the requests of step
        // kt + 1 go out in read(half 0) of step kt and are retired (vmcnt) by each wave before the last barrier the EARLIER group passes in step kt.
        static_assert(WTM / HM == 2, "two halves");
        const int grp = (wave >> 2) & 1;
        auto bar = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        if (grp) bar();
#pragma unroll 1
        for (int kt = 0; kt < nk; ++kt) {
            const unsigned char *base = smem + (kt & 1) * STAGE;
            h16x8 wh[WTN], wl[WTN], xh[HM], xl[HM];
#pragma unroll
            for (int c = 0; c < WTN; ++c) {
                wh[c] = *reinterpret_cast<const h16x8 *>(base + wo + c * 2048);
                wl[c] = *reinterpret_cast<const h16x8 *>(base + (wo ^ 64u) + c * 2048);
            }
#pragma unroll
            for (int a = 0; a < HM; ++a) {
                xh[a] = *reinterpret_cast<const h16x8 *>(base + xo + a * 2048);
                xl[a] = *reinterpret_cast<const h16x8 *>(base + (xo ^ 64u) + a * 2048);
            }
            if (kt + 1 < nk) {
                issue_x(kt + 1, (kt & 1) ^ 1);
                issue_w(kt + 1, (kt & 1) ^ 1);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            bar();
            __builtin_amdgcn_s_setprio(1);
            mfmas(0, wh, wl, xh, xl);
            __builtin_amdgcn_s_setprio(0);
            bar();
#pragma unroll
            for (int a = 0; a < HM; ++a) {
                xh[a] = *reinterpret_cast<const h16x8 *>(base + xo + (HM + a) * 2048);
                xl[a] = *reinterpret_cast<const h16x8 *>(base + (xo ^ 64u) + (HM + a) * 2048);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (grp) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bar();
            __builtin_amdgcn_s_setprio(1);
            mfmas(1, wh, wl, xh, xl);
            __builtin_amdgcn_s_setprio(0);
            if (!grp) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bar();
        }
        if (!grp) bar();
    } else {
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        // (the other buffer is free: everybody finished reading it before the last barrier)
        compute(NOREAD ? 0 : cur, (kt + 1 < nk && !NODMA) ? kt + 1 : -1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    }

    // epilogue: D[n][m]: lane (i16, kq) of fragment (a, c) holds channels n..n+3 (n = 16c + 4kq) of row m = 16a + i16
    float *__restrict__ C = p.C + (size_t)b * p.c_bs;
    const bool prelu = p.act == ARSEG_ACT_PRELU, relu = p.act == ARSEG_ACT_RELU, sigm = p.act == ARSEG_ACT_SIGMOID;
    float vmax = 0.f;
#pragma unroll
    for (int c = 0; c < WTN; ++c) {
        const int n = n0 + wn * (BN / NWN) + c * 16 + 4 * kq;
        if (n >= p.N) continue;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
        if (p.scale) sc = *reinterpret_cast<const f32x4 *>(p.scale + n);
        if (p.bias) bi = *reinterpret_cast<const f32x4 *>(p.bias + n);
#pragma unroll
        for (int a = 0; a < WTM; ++a) {
            const int m = m0 + wm * (BM / NWM) + a * 16 + i16;
            if (m >= p.M) continue;
            f32x4 v = acc[a][c] * sc + bi;
            if (p.res) v += *reinterpret_cast<const f32x4 *>(p.res + (size_t)m * p.res_ld + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (relu) v[e] = fmaxf(v[e], 0.f);
                else if (prelu) v[e] = v[e] >= 0.f ? v[e] : v[e] * p.slope;
                else if (sigm) v[e] = 1.0f / (1.0f + __expf(-v[e]));
            }
            if (p.out_split) {            // the output is the next GEMM's activation operand: written as split rows (ldc == N, N % 32 == 0)
                unsigned h01, h23, l01, l23;
                arseg_split_f16(v, h01, h23, l01, l23);
                vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                unsigned char *o = reinterpret_cast<unsigned char *>(C) + ((size_t)m * p.N) * 4 + (n >> 5) * 128 + (n & 31) * 2;
                *reinterpret_cast<uint2 *>(o) = uint2{h01, h23};
                *reinterpret_cast<uint2 *>(o + 64) = uint2{l01, l23};
            } else {
                // M / z / a 1x1 conv's output: tens of MB that the next launch streams once -- nontemporal stores (`nt`) keep them out of the
                // 4 MB L2 of the XCD: up_2's tap GEMM (207 MB out, K = 256) 138 -> 107 us alone, +0.5 % on the step
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(C + (size_t)m * p.ldc + n));
            }
        }
    }
    if (p.range_flag && vmax > p.range_limit) atomicOr(p.range_flag, 1u);
}

// fp32 rows -> split rows (the operand layout above); 4 values per thread
__global__ __launch_bounds__(256) void split_rows_kernel(const float *__restrict__ in, long long in_ld, unsigned char *__restrict__ out, long long rows, int K, float mul,
                                                         unsigned *range_flag, float range_limit) {
    const int k4 = K >> 2;
    float vmax = 0.f;
    const long long total = rows * k4;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long r = idx / k4;
        const int c = (int)(idx - r * k4) * 4;
        f32x4 v = *reinterpret_cast<const f32x4 *>(in + r * in_ld + c);
        v *= mul;
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        unsigned h01, h23, l01, l23;
        arseg_split_f16(v, h01, h23, l01, l23);
        unsigned char *o = out + r * (long long)K * 4 + (c >> 5) * 128 + (c & 31) * 2;
        *reinterpret_cast<uint2 *>(o) = uint2{h01, h23};
        *reinterpret_cast<uint2 *>(o + 64) = uint2{l01, l23};
    }
    if (range_flag && vmax > range_limit) atomicOr(range_flag, 1u);
}

// The per-image second weight operand of arseg_gemm_x3_cat_fwd for the folded PSP bottleneck (model/pspnet.py:14-31), straight from the pyramid
// terms: w2[n][co][k] = t[n][k][co] * unscale[co] for k < rows, 0 for rows <= k < 64, written as split rows (one thread = 4 consecutive k).
__global__ __launch_bounds__(256) void psp_w2_split_kernel(const float *__restrict__ t, const float *__restrict__ unscale, unsigned char *__restrict__ out,
                                                           int N, int rows, int Cout, unsigned *range_flag, float range_limit) {
    const long long total = (long long)N * Cout * 16;
    float vmax = 0.f;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(idx & 15) * 4;
        const long long nc = idx >> 4;
        const int co = (int)(nc % Cout);
        const long long n = nc / Cout;
        const float f = unscale[co];
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = k + j < rows ? t[(n * rows + k + j) * Cout + co] * f : 0.f;
        // (NaN compares false in fmaxf's favour of the other operand: the !(.. <= ..) form below catches it as the other watchers do not need to --
        // a NaN here can only come from t, which the GEMM that produced t already multiplied)
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        unsigned h01, h23, l01, l23;
        arseg_split_f16(v, h01, h23, l01, l23);
        unsigned char *o = out + nc * 256 + (k >> 5) * 128 + (k & 31) * 2;
        *reinterpret_cast<uint2 *>(o) = uint2{h01, h23};
        *reinterpret_cast<uint2 *>(o + 64) = uint2{l01, l23};
    }
    if (range_flag && vmax > range_limit) atomicOr(range_flag, 1u);
}

template <int NWM, int NWN, int WTM, int WTN, int ABL = 0, bool STAG = false>
int launch_x3(GX3Params &p, hipStream_t hs) {
    constexpr int BM = 16 * NWM * WTM, BN = 16 * NWN * WTN;
    p.tiles_m = arseg_cdiv(p.M, BM); p.tiles_n = arseg_cdiv(p.N, BN);
    if ((long long)p.tiles_m * p.tiles_n * p.batch >= (1ll << 31)) return ARSEG_EUNSUPPORTED;
    const size_t smem = (size_t)2 * (BM + BN) * 128;
    static ArsegSmemAttr attr;
    if (int e = arseg_allow_smem(attr, reinterpret_cast<const void *>(gemm_x3_kernel<NWM, NWN, WTM, WTN, ABL, STAG>), smem)) return e;
    hipLaunchKernelGGL((gemm_x3_kernel<NWM, NWN, WTM, WTN, ABL, STAG>), dim3(p.tiles_m * p.tiles_n * p.batch), dim3(64 * NWM * NWN), smem, hs, p);
    return arseg_launch_status();
}

template <int ABL>
int launch_cfg(GX3Params &p, int cfg, hipStream_t hs) {
    switch (cfg) {
        case 0: return launch_x3<2, 4, 8, 4, ABL>(p, hs);      // 256 x 256,  8 waves
        case 1: return launch_x3<4, 4, 4, 4, ABL>(p, hs);      // 256 x 256, 16 waves
        case 2: return launch_x3<2, 4, 4, 4, ABL>(p, hs);      // 128 x 256,  8 waves
        case 3: return launch_x3<2, 4, 4, 2, ABL>(p, hs);      // 128 x 128,  8 waves
        case 4: return launch_x3<4, 4, 4, 2, ABL>(p, hs);      // 256 x 128, 16 waves
        case 5: return launch_x3<4, 4, 2, 4, ABL>(p, hs);      // 128 x 256, 16 waves
        case 6: return launch_x3<4, 4, 4, 4, ABL, true>(p, hs);   // 256 x 256, 16 waves in two staggered groups
        default: return ARSEG_EINVAL;
    }
}

}  // namespace

extern "C" int arseg_psp_w2_split_fwd(const float *t, const float *unscale, void *out, int N, int rows, int Cout, void *range_flag,
                                      float range_limit, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(t); ARSEG_CHECK_PTR(unscale); ARSEG_CHECK_PTR(out);
    ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(rows); ARSEG_CHECK_POS(Cout);
    if (rows > 64) return ARSEG_EUNSUPPORTED;
    if (!ARSEG_ALIGNED16(out) || (reinterpret_cast<uintptr_t>(range_flag) & 3)) return ARSEG_EINVAL;
    const long long total = (long long)N * Cout * 16;
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(psp_w2_split_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, arseg_stream(stream), t, unscale,
                       reinterpret_cast<unsigned char *>(out), N, rows, Cout, reinterpret_cast<unsigned *>(range_flag), range_limit > 0.0f ? range_limit : 65504.0f);
    return arseg_launch_status();
}

extern "C" int arseg_split_rows_fwd(const float *in, long long in_ld, void *out, long long rows, int K, float mul, void *range_flag,
                                    float range_limit, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(in); ARSEG_CHECK_PTR(out);
    if (reinterpret_cast<uintptr_t>(range_flag) & 3) return ARSEG_EINVAL;
    if (rows <= 0 || K <= 0 || (K & 31) || in_ld < K || (in_ld & 3) || !ARSEG_ALIGNED16(in) || !ARSEG_ALIGNED16(out)) return ARSEG_EINVAL;
    long long blocks = (rows * (K >> 2) + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, arseg_stream(stream), in, in_ld, reinterpret_cast<unsigned char *>(out), rows, K, mul,
                       reinterpret_cast<unsigned *>(range_flag), range_limit > 0.0f ? range_limit : 65504.0f);
    return arseg_launch_status();
}

static int gemm_x3(const void *x_split, const void *w_split, const void *x2_split, const void *w2_split, float *out, int M, int N, int K, int K2,
                   int out_ld, int batch, long long x_batch_stride, long long w_batch_stride, long long x2_batch_stride, long long w2_batch_stride,
                   long long out_batch_stride, const float *scale, const float *bias, const float *residual, int res_ld, int act,
                   float prelu_slope, int out_split, int tile_cfg, void *range_flag, float range_limit, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(x_split); ARSEG_CHECK_PTR(w_split); ARSEG_CHECK_PTR(out);
    if (M <= 0 || N <= 0 || K <= 0 || (K & 31) || (N & 3) || out_ld < N || (out_ld & 3) || batch <= 0) return ARSEG_EINVAL;
    if (!ARSEG_ALIGNED16(x_split) || !ARSEG_ALIGNED16(w_split) || !ARSEG_ALIGNED16(out) || (x_batch_stride & 15) || (w_batch_stride & 15) || (out_batch_stride & 3))
        return ARSEG_EINVAL;
    if (K2 < 0 || (K2 & 31) || (K2 && (!x2_split || !w2_split || !ARSEG_ALIGNED16(x2_split) || !ARSEG_ALIGNED16(w2_split) || (x2_batch_stride & 15) || (w2_batch_stride & 15))))
        return ARSEG_EINVAL;
    if ((scale && !ARSEG_ALIGNED16(scale)) || (bias && !ARSEG_ALIGNED16(bias))) return ARSEG_EINVAL;
    if (residual && (batch > 1 || res_ld < N || (res_ld & 3) || !ARSEG_ALIGNED16(residual))) return ARSEG_EINVAL;
    if (out_split && ((N & 31) || out_ld != N)) return ARSEG_EINVAL;
    const int Kmax = K > K2 ? K : K2;
    if ((long long)M * Kmax * 4 >= (1ll << 32) || (long long)N * Kmax * 4 >= (1ll << 32)) return ARSEG_EUNSUPPORTED;      // 32-bit buffer offsets
    GX3Params p;
    p.X = reinterpret_cast<const unsigned char *>(x_split); p.W = reinterpret_cast<const unsigned char *>(w_split); p.C = out;
    p.X2 = reinterpret_cast<const unsigned char *>(x2_split); p.W2 = reinterpret_cast<const unsigned char *>(w2_split); p.K2 = K2;
    p.x2_bytes = (unsigned)((long long)M * K2 * 4); p.w2_bytes = (unsigned)((long long)N * K2 * 4);
    p.x2_bs = batch > 1 ? x2_batch_stride : 0; p.w2_bs = batch > 1 ? w2_batch_stride : 0;
    p.scale = scale; p.bias = bias; p.res = residual; p.res_ld = res_ld; p.out_split = out_split ? 1 : 0; p.M = M; p.N = N; p.K = K; p.ldc = out_ld;
    p.x_bytes = (unsigned)((long long)M * K * 4); p.w_bytes = (unsigned)((long long)N * K * 4);
    p.x_bs = batch > 1 ? x_batch_stride : 0; p.w_bs = batch > 1 ? w_batch_stride : 0; p.c_bs = batch > 1 ? out_batch_stride : 0;
    p.batch = batch; p.act = act; p.slope = prelu_slope;
    if (reinterpret_cast<uintptr_t>(range_flag) & 3) return ARSEG_EINVAL;
    p.range_flag = out_split ? reinterpret_cast<unsigned *>(range_flag) : nullptr; p.range_limit = range_limit > 0.0f ? range_limit : 65504.0f;
    hipStream_t hs = arseg_stream(stream);
    const int abl = tile_cfg >> 3;
    tile_cfg &= 7;
    switch (abl) {
        case 0: return launch_cfg<0>(p, tile_cfg, hs);
        case 1: return launch_cfg<1>(p, tile_cfg, hs);
        case 2: return launch_cfg<2>(p, tile_cfg, hs);
        case 3: return launch_cfg<3>(p, tile_cfg, hs);
        case 4: return launch_cfg<4>(p, tile_cfg, hs);
        default: return ARSEG_EINVAL;
    }
}

extern "C" int arseg_gemm_x3_fwd(const void *x_split, const void *w_split, float *out, int M, int N, int K, int out_ld, int batch,
                                 long long x_batch_stride, long long w_batch_stride, long long out_batch_stride, const float *scale,
                                 const float *bias, const float *residual, int res_ld, int act, float prelu_slope, int out_split, int tile_cfg,
                                 void *range_flag, float range_limit, arseg_stream_t stream) {
    return gemm_x3(x_split, w_split, nullptr, nullptr, out, M, N, K, 0, out_ld, batch, x_batch_stride, w_batch_stride, 0, 0, out_batch_stride, scale,
                   bias, residual, res_ld, act, prelu_slope, out_split, tile_cfg, range_flag, range_limit, stream);
}

extern "C" int arseg_gemm_x3_cat_fwd(const void *x_split, const void *w_split, const void *x2_split, const void *w2_split, float *out, int M, int N,
                                     int K, int K2, int out_ld, int batch, long long x_batch_stride, long long w_batch_stride,
                                     long long x2_batch_stride, long long w2_batch_stride, long long out_batch_stride, const float *scale,
                                     const float *bias, int act, float prelu_slope, int out_split, int tile_cfg, void *range_flag,
                                     float range_limit, arseg_stream_t stream) {
    if (K2 <= 0) return ARSEG_EINVAL;
    return gemm_x3(x_split, w_split, x2_split, w2_split, out, M, N, K, K2, out_ld, batch, x_batch_stride, w_batch_stride, x2_batch_stride, w2_batch_stride,
                   out_batch_stride, scale, bias, nullptr, 0, act, prelu_slope, out_split, tile_cfg, range_flag, range_limit, stream);
}
