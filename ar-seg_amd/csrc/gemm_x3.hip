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
//
// (r5) Two generalisations of the same kernel, both without a byte of im2col or a register-staged operand:
//   * IMPLICIT 3x3 (taps == 9).  The activation operand is a zero-bordered ("padded") NHWC image [N][H + 2d][W + 2d][Cin] in the row format, and
//     GEMM row m IS padded pixel m: tap t of a 3x3 conv with padding == dilation == d reads row m + ((t / 3 - 1) * (W + 2d) + (t % 3 - 1)) * d.
//     A K step is (tap, 128-byte channel group): the tap is one scalar added to every lane's DMA source offset, the padding is the image's own
//     zero border, rows in front of / behind the buffer are the buffer descriptor's out-of-range zero.  Border rows of the output are garbage and
//     are either dropped (out_mode NHWC: the rows are compacted to the unpadded image) or written as zeros (out_mode PADDED: the output is the next
//     3x3 conv's operand, same geometry).  Cost of the border: (H + 2d)(W + 2d) / HW rows computed -- 1.05 at 64 x 128, 1.10 at 32 x 64, d = 1.
//   * 16-BIT ROWS (FMT 1 = fp16, 2 = bf16): a row is K values of 2 bytes, a K step is still 128 bytes = 64 values = two v_mfma_f32_16x16x32 per
//     fragment pair instead of the three of the hi/lo emulation; the same DMA image, the same swizzle, the same fragment reads (slot kq = k 8kq..,
//     slot 4 + kq = k 32 + 8kq..).  The 16-bit storage path's 1x1 and 3x3 stride-1 convs (BASELINE configs[2] / [4]) run on it.
#include "arseg_common.h"

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
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
    // implicit 3x3 (taps == 9, K = 9 * Cin): see the header comment.  taps == 1: a plain GEMM.
    int taps, kg;                     // kg = 128-byte K steps per tap
    int pH, pW, pad, dil, iH, iW;     // padded image extent, border width (= the conv's padding), dilation, unpadded extent
    int out_mode;                     // 0: row m -> row m; 1: padded row -> unpadded NHWC row (border rows dropped); 2: padded row kept, border rows written as zeros
    const unsigned char *res_rows;    // residual in the ACTIVATION's row format at the same row index m (out_mode 2 chains: the block input); else p.res (plain, at the output row)
    unsigned ldx;                     // bytes per activation row (K / taps values)
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

typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
template <int FMT>
__device__ __forceinline__ f32x4 mfma16(const h16x8 a, const h16x8 b, const f32x4 c) {
    if constexpr (FMT == 2) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b16x8, a), __builtin_bit_cast(b16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// NWM x NWN waves; 16-row fragments per wave: WTM along M (activation rows), WTN along N (weight rows); FMT 0: split rows (f16x3), 1: fp16 rows, 2: bf16 rows
template <int NWM, int NWN, int WTM, int WTN, int ABL = 0, bool STAG = false, int FMT = 0>
__global__ __launch_bounds__(64 * NWM * NWN) void gemm_x3_kernel(const GX3Params p) {
    constexpr unsigned EB = FMT == 0 ? 4u : 2u;           // bytes per operand value
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
    const unsigned ld = (unsigned)p.K * EB, ld2 = (unsigned)p.K2 * EB, ldx = p.ldx;
    const u32x4 x2_rsrc = make_rsrc(p.K2 ? p.X2 + (size_t)b * p.x2_bs : p.X, p.K2 ? p.x2_bytes : 0u);
    const u32x4 w2_rsrc = make_rsrc(p.K2 ? p.W2 + (size_t)b * p.w2_bs : p.W, p.K2 ? p.w2_bytes : 0u);
    const int k1 = (int)(ld >> 7);

    // DMA: lane (r8, pos) of instruction j fills LDS row wave*rows + 8j + r8, 16-byte position pos, with source slot pos ^ g(row)
    const int r8 = lane >> 3, pos = lane & 7;
    unsigned xsrc[XI], wsrc[WI], xsrc2[XI], wsrc2[WI];
#pragma unroll
    for (int j = 0; j < XI; ++j) {
        const int row = wave * (BM / NW) + j * 8 + r8;
        const unsigned gr = (unsigned)min(m0 + row, p.M - 1), sl = (unsigned)((pos ^ ((row >> 1) & 7)) << 4);
        xsrc[j] = gr * ldx + sl;
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
    // implicit 3x3: K step kt = (tap, group) -> byte offset of the step inside the activation rows = tap row offset * ldx + group * 128
    // (may be "negative": the 32-bit sum then lies beyond the buffer and the descriptor returns zeros -- only border rows read there)
    const int kg = p.kg, taps = p.taps, pWd = p.pW * p.dil, dil = p.dil;
    auto xstep = [&](int kt) -> unsigned {
        if (taps == 1) return (unsigned)kt * 128u;
        const int tap = kt / kg, g = kt - tap * kg, ty = tap / 3, tx = tap - 3 * ty;
        return (unsigned)((ty - 1) * pWd + (tx - 1) * dil) * ldx + (unsigned)g * 128u;
    };
    // (two explicit arms: a `second ? xsrc2[j] : xsrc[j]` select made hipcc index the offset arrays through scratch -- a scratch load and a
    // vmcnt(0) behind every DMA request)
    auto issue_x = [&](int kt, int st) {
        const unsigned base = lds0 + (unsigned)st * STAGE;
        if (kt >= k1) {                                // (uniform) the K steps of the concatenated second operand pair
            const unsigned kb = (unsigned)(kt - k1) * 128u;
#pragma unroll
            for (int j = 0; j < XI; ++j) dma16_buf(x2_rsrc, xsrc2[j] + kb, base + (unsigned)(wave * (BM / NW) + j * 8) * 128u);
        } else {
            const unsigned kb = xstep(kt);
#pragma unroll
            for (int j = 0; j < XI; ++j) dma16_buf(x_rsrc, xsrc[j] + kb, base + (unsigned)(wave * (BM / NW) + j * 8) * 128u);
        }
    };
    auto issue_w = [&](int kt, int st) {
        const unsigned base = lds0 + (unsigned)st * STAGE;
        if (kt >= k1) {
            const unsigned kb = (unsigned)(kt - k1) * 128u;
#pragma unroll
            for (int j = 0; j < WI; ++j) dma16_buf(w2_rsrc, wsrc2[j] + kb, base + XB + (unsigned)(wave * (BN / NW) + j * 8) * 128u);
        } else {
            const unsigned kb = (unsigned)kt * 128u;
#pragma unroll
            for (int j = 0; j < WI; ++j) dma16_buf(w_rsrc, wsrc[j] + kb, base + XB + (unsigned)(wave * (BN / NW) + j * 8) * 128u);
        }
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
        if constexpr (FMT != 0) {      // 16-bit rows: "h" = k 0..31, "l" = k 32..63 of the step, one product each
#pragma unroll
            for (int a = 0; a < HM; ++a)
#pragma unroll
                for (int c = 0; c < WTN; ++c) acc[h * HM + a][c] = mfma16<FMT>(wh[c], xh[a], acc[h * HM + a][c]);
#pragma unroll
            for (int a = 0; a < HM; ++a)
#pragma unroll
                for (int c = 0; c < WTN; ++c) acc[h * HM + a][c] = mfma16<FMT>(wl[c], xl[a], acc[h * HM + a][c]);
            return;
        }
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

    const int nk = k1 + (int)(ld2 >> 7);
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
    const bool prelu = p.act == ARSEG_ACT_PRELU, relu = p.act == ARSEG_ACT_RELU, sigm = p.act == ARSEG_ACT_SIGMOID;
    float vmax = 0.f;
    // this lane's rows: where each goes (orow < 0: nowhere) and whether it is a pixel of the image (implicit 3x3: border rows of the padded
    // geometry are computed like any other and then dropped or zeroed)
    int orow[WTM], urow[WTM];         // output row; row of the unpadded NHWC image (a plain residual lives there), -1: none
    unsigned border = 0u;             // bit a: row a is a border pixel of the padded geometry
#pragma unroll
    for (int a = 0; a < WTM; ++a) {
        const int m = m0 + wm * (BM / NWM) + a * 16 + i16;
        orow[a] = urow[a] = m < p.M ? m : -1;
        if (p.out_mode != 0 && m < p.M) {
            const unsigned per = (unsigned)(p.pH * p.pW), img = (unsigned)m / per, r = (unsigned)m - img * per, yp = r / (unsigned)p.pW, xp = r - yp * (unsigned)p.pW;
            const int y = (int)yp - p.pad, x = (int)xp - p.pad;
            const bool in = (unsigned)y < (unsigned)p.iH && (unsigned)x < (unsigned)p.iW;
            if (!in) border |= 1u << a;
            urow[a] = in ? ((int)img * p.iH + y) * p.iW + x : -1;
            if (p.out_mode == 1) orow[a] = urow[a];
        }
    }
    unsigned char *__restrict__ Cb = reinterpret_cast<unsigned char *>(p.C) + (size_t)b * p.c_bs * (FMT == 0 ? 4 : 2);
#pragma unroll
    for (int c = 0; c < WTN; ++c) {
        const int n = n0 + wn * (BN / NWN) + c * 16 + 4 * kq;
        if (n >= p.N) continue;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
        if (p.scale) sc = *reinterpret_cast<const f32x4 *>(p.scale + n);
        if (p.bias) bi = *reinterpret_cast<const f32x4 *>(p.bias + n);
#pragma unroll
        for (int a = 0; a < WTM; ++a) {
            if (orow[a] < 0) continue;
            const int m = m0 + wm * (BM / NWM) + a * 16 + i16;
            f32x4 v = acc[a][c] * sc + bi;
            if constexpr (FMT == 0) {
                if (p.res && urow[a] >= 0) v += *reinterpret_cast<const f32x4 *>(p.res + (size_t)urow[a] * p.res_ld + n);
                if (p.res_rows) {          // split rows [M][N]: 4 hi halves + 4 lo halves of this lane's channels
                    const unsigned char *r = p.res_rows + ((size_t)m * p.N) * 4 + (n >> 5) * 128 + (n & 31) * 2;
                    const uint2 h = *reinterpret_cast<const uint2 *>(r), l = *reinterpret_cast<const uint2 *>(r + 64);
                    const h16x2 h0 = __builtin_bit_cast(h16x2, h.x), h1 = __builtin_bit_cast(h16x2, h.y), l0 = __builtin_bit_cast(h16x2, l.x), l1 = __builtin_bit_cast(h16x2, l.y);
                    v[0] += (float)h0[0] + (float)l0[0]; v[1] += (float)h0[1] + (float)l0[1];
                    v[2] += (float)h1[0] + (float)l1[0]; v[3] += (float)h1[1] + (float)l1[1];
                }
            } else {
                const uint16_t *r16 = p.res_rows ? reinterpret_cast<const uint16_t *>(p.res_rows) + (size_t)m * p.N + n
                                                 : (p.res && urow[a] >= 0 ? reinterpret_cast<const uint16_t *>(p.res) + (size_t)urow[a] * p.res_ld + n : nullptr);
                if (r16) {
                    const uint2 rv = *reinterpret_cast<const uint2 *>(r16);
                    v[0] += arseg_h2f<FMT == 2>((uint16_t)(rv.x & 0xffffu)); v[1] += arseg_h2f<FMT == 2>((uint16_t)(rv.x >> 16));
                    v[2] += arseg_h2f<FMT == 2>((uint16_t)(rv.y & 0xffffu)); v[3] += arseg_h2f<FMT == 2>((uint16_t)(rv.y >> 16));
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (relu) v[e] = fmaxf(v[e], 0.f);
                else if (prelu) v[e] = v[e] >= 0.f ? v[e] : v[e] * p.slope;
                else if (sigm) v[e] = 1.0f / (1.0f + __expf(-v[e]));
            }
            if ((border >> a) & 1u) v = f32x4{0.f, 0.f, 0.f, 0.f};          // (out_mode 2) the zero border of the next conv's operand
            if constexpr (FMT != 0) {
                const uint2 o = {(unsigned)arseg_f2h<FMT == 2>(v[0]) | ((unsigned)arseg_f2h<FMT == 2>(v[1]) << 16),
                                 (unsigned)arseg_f2h<FMT == 2>(v[2]) | ((unsigned)arseg_f2h<FMT == 2>(v[3]) << 16)};
                *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(Cb) + (size_t)orow[a] * p.ldc + n) = o;
            } else if (p.out_split) {            // the output is the next GEMM's activation operand: written as split rows (ldc == N, N % 32 == 0)
                unsigned h01, h23, l01, l23;
                arseg_split_f16(v, h01, h23, l01, l23);
                vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                unsigned char *o = Cb + ((size_t)orow[a] * p.N) * 4 + (n >> 5) * 128 + (n & 31) * 2;
                *reinterpret_cast<uint2 *>(o) = uint2{h01, h23};
                *reinterpret_cast<uint2 *>(o + 64) = uint2{l01, l23};
            } else {
                // M / z / a 1x1 conv's output: tens of MB that the next launch streams once -- nontemporal stores (`nt`) keep them out of the
                // 4 MB L2 of the XCD: up_2's tap GEMM (207 MB out, K = 256) 138 -> 107 us alone, +0.5 % on the step
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(Cb) + (size_t)orow[a] * p.ldc + n));
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

template <int NWM, int NWN, int WTM, int WTN, int ABL = 0, bool STAG = false, int FMT = 0>
int launch_x3(GX3Params &p, hipStream_t hs) {
    constexpr int BM = 16 * NWM * WTM, BN = 16 * NWN * WTN;
    p.tiles_m = arseg_cdiv(p.M, BM); p.tiles_n = arseg_cdiv(p.N, BN);
    if ((long long)p.tiles_m * p.tiles_n * p.batch >= (1ll << 31)) return ARSEG_EUNSUPPORTED;
    const size_t smem = (size_t)2 * (BM + BN) * 128;
    static ArsegSmemAttr attr;
    if (int e = arseg_allow_smem(attr, reinterpret_cast<const void *>(gemm_x3_kernel<NWM, NWN, WTM, WTN, ABL, STAG, FMT>), smem)) return e;
    hipLaunchKernelGGL((gemm_x3_kernel<NWM, NWN, WTM, WTN, ABL, STAG, FMT>), dim3(p.tiles_m * p.tiles_n * p.batch), dim3(64 * NWM * NWN), smem, hs, p);
    return arseg_launch_status();
}

// tile_cfg -> tile shape.  0-6: the GEMM shapes of rounds 3-4; 7-10 (r5): narrow / short tiles for the implicit 3x3 convs of the 64- and 128-channel
// layers (a 64-wide N tile cannot feed 16 waves: 8 rows of weights per DMA instruction)
template <int ABL, int FMT>
int launch_cfg(GX3Params &p, int cfg, hipStream_t hs) {
    switch (cfg) {
        case 0: return launch_x3<2, 4, 8, 4, ABL, false, FMT>(p, hs);      // 256 x 256,  8 waves
        case 1: return launch_x3<4, 4, 4, 4, ABL, false, FMT>(p, hs);      // 256 x 256, 16 waves
        case 2: return launch_x3<2, 4, 4, 4, ABL, false, FMT>(p, hs);      // 128 x 256,  8 waves
        case 3: return launch_x3<2, 4, 4, 2, ABL, false, FMT>(p, hs);      // 128 x 128,  8 waves
        case 4: return launch_x3<4, 4, 4, 2, ABL, false, FMT>(p, hs);      // 256 x 128, 16 waves
        case 5: return launch_x3<4, 4, 2, 4, ABL, false, FMT>(p, hs);      // 128 x 256, 16 waves
        case 6: return launch_x3<4, 4, 4, 4, ABL, true, FMT>(p, hs);       // 256 x 256, 16 waves in two staggered groups
        default: break;
    }
    if constexpr (ABL == 0) {
        switch (cfg) {
            case 7: return launch_x3<4, 2, 4, 2, 0, false, FMT>(p, hs);    // 256 x  64,  8 waves
            case 8: return launch_x3<4, 1, 2, 4, 0, false, FMT>(p, hs);    // 128 x  64,  4 waves (three workgroups per compute unit)
            case 9: return launch_x3<4, 2, 2, 2, 0, false, FMT>(p, hs);    // 128 x  64,  8 waves
            case 10: return launch_x3<2, 4, 2, 2, 0, false, FMT>(p, hs);   //  64 x 128,  8 waves
            case 11: return launch_x3<4, 2, 2, 4, 0, false, FMT>(p, hs);   // 128 x 128,  8 waves, 32 x 64 wave tiles
            default: break;
        }
    }
    return ARSEG_EINVAL;
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
    p.taps = 1; p.kg = K >> 5; p.pH = p.pW = p.iH = p.iW = 1; p.pad = 0; p.dil = 1; p.out_mode = 0; p.res_rows = nullptr; p.ldx = (unsigned)K * 4u;
    hipStream_t hs = arseg_stream(stream);
#ifdef ARSEG_GX3_ABLATE      // dev builds (tools/bench_gemm_x3.py): tile_cfg = 16 * ablation + tile; wrong results, same instruction stream otherwise
    switch (tile_cfg >> 4) {
        case 1: return launch_cfg<1, 0>(p, tile_cfg & 15, hs);
        case 2: return launch_cfg<2, 0>(p, tile_cfg & 15, hs);
        case 3: return launch_cfg<3, 0>(p, tile_cfg & 15, hs);
        case 4: return launch_cfg<4, 0>(p, tile_cfg & 15, hs);
        default: break;
    }
#endif
    return launch_cfg<0, 0>(p, tile_cfg, hs);
}

// Implicit 3x3 conv / plain GEMM on rows of any of the three formats (fmt = enum arseg_rows_fmt); see arseg_conv3x3_rows_fwd in the header.
// (round 6) The implicit-3x3 entry points arseg_conv3x3_rows_fwd / arseg_pad_rows_fwd of round 5 are gone from the ABI: the route measured parity with
// the patch-resident kernels and was never selected.  conv_rows keeps its generality (taps = 9 on zero-bordered rows) because the kernel's K-step
// addressing is shared with the plain GEMMs; the only caller left is arseg_gemm_rows16_fwd (taps = 1).
enum { ARSEG_ROWS_X3 = 0, ARSEG_ROWS_F16 = 1, ARSEG_ROWS_BF16 = 2 };
enum { ARSEG_ROWS_OUT_NHWC = 0, ARSEG_ROWS_OUT_PADDED = 1 };
static int conv_rows(const void *x_rows, const void *w_rows, void *out, int fmt, int taps, int N, int H, int W, int Cin, int Cout, int dil, int out_mode, int out_ld,
                     const float *scale, const float *bias, const void *residual, int res_mode, int res_ld, int act, float prelu_slope, int tile_cfg,
                     void *range_flag, float range_limit, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(x_rows); ARSEG_CHECK_PTR(w_rows); ARSEG_CHECK_PTR(out);
    ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(H); ARSEG_CHECK_POS(W); ARSEG_CHECK_POS(Cin); ARSEG_CHECK_POS(Cout);
    if (fmt != ARSEG_ROWS_X3 && fmt != ARSEG_ROWS_F16 && fmt != ARSEG_ROWS_BF16) return ARSEG_EINVAL;
    if ((taps != 1 && taps != 9) || dil < 1 || dil > 8) return ARSEG_EINVAL;
    const int eb = fmt == ARSEG_ROWS_X3 ? 4 : 2, kgran = 128 / eb;          // values per 128-byte K step
    if (Cin % kgran) return ARSEG_EUNSUPPORTED;
    if ((Cout & 3) || !ARSEG_ALIGNED16(x_rows) || !ARSEG_ALIGNED16(w_rows) || !ARSEG_ALIGNED16(out)) return ARSEG_EINVAL;
    if ((scale && !ARSEG_ALIGNED16(scale)) || (bias && !ARSEG_ALIGNED16(bias))) return ARSEG_EINVAL;
    if (out_mode != ARSEG_ROWS_OUT_NHWC && out_mode != ARSEG_ROWS_OUT_PADDED) return ARSEG_EINVAL;
    if (res_mode != ARSEG_ROWS_OUT_NHWC && res_mode != ARSEG_ROWS_OUT_PADDED) return ARSEG_EINVAL;
    if (taps == 1 && (out_mode != ARSEG_ROWS_OUT_NHWC || (residual && res_mode != ARSEG_ROWS_OUT_NHWC))) return ARSEG_EINVAL;      // a 1x1 conv has no border
    const int pad = taps == 9 ? dil : 0, pH = H + 2 * pad, pW = W + 2 * pad;
    const long long M = (long long)N * pH * pW, K = (long long)taps * Cin;
    if (M * Cin * eb >= (1ll << 31) || (long long)Cout * K * eb >= (1ll << 32) || M >= (1ll << 31)) return ARSEG_EUNSUPPORTED;      // 32-bit buffer offsets, "negative" tap offsets
    const bool padded_out = out_mode == ARSEG_ROWS_OUT_PADDED;
    // padded output = the next conv's operand in the same format: split rows need whole 32-channel groups, rows are dense
    if (padded_out && ((fmt == ARSEG_ROWS_X3 && (Cout & 31)) || (fmt != ARSEG_ROWS_X3 && (Cout & 7)) || out_ld != Cout)) return ARSEG_EINVAL;
    if (!padded_out && (out_ld < Cout || (out_ld & 3))) return ARSEG_EINVAL;
    if (residual) {
        if (!ARSEG_ALIGNED16(residual)) return ARSEG_EINVAL;
        if (res_mode == ARSEG_ROWS_OUT_PADDED ? (res_ld != Cout || (fmt == ARSEG_ROWS_X3 && (Cout & 31))) : (res_ld < Cout || (res_ld & 3))) return ARSEG_EINVAL;
    }
    if (reinterpret_cast<uintptr_t>(range_flag) & 3) return ARSEG_EINVAL;
    GX3Params p;
    p.X = reinterpret_cast<const unsigned char *>(x_rows); p.W = reinterpret_cast<const unsigned char *>(w_rows); p.C = reinterpret_cast<float *>(out);
    p.X2 = p.W2 = nullptr; p.K2 = 0; p.x2_bytes = p.w2_bytes = 0; p.x2_bs = p.w2_bs = 0;
    p.scale = scale; p.bias = bias;
    p.res = residual && res_mode == ARSEG_ROWS_OUT_NHWC ? reinterpret_cast<const float *>(residual) : nullptr;
    p.res_rows = residual && res_mode == ARSEG_ROWS_OUT_PADDED ? reinterpret_cast<const unsigned char *>(residual) : nullptr;
    p.res_ld = res_ld; p.out_split = padded_out && fmt == ARSEG_ROWS_X3 ? 1 : 0;
    p.M = (int)M; p.N = Cout; p.K = (int)K; p.ldc = out_ld;
    p.x_bytes = (unsigned)(M * Cin * eb); p.w_bytes = (unsigned)((long long)Cout * K * eb);
    p.x_bs = p.w_bs = p.c_bs = 0; p.batch = 1; p.act = act; p.slope = prelu_slope;
    p.range_flag = p.out_split ? reinterpret_cast<unsigned *>(range_flag) : nullptr; p.range_limit = range_limit > 0.0f ? range_limit : 65504.0f;
    p.taps = taps; p.kg = Cin / kgran; p.pH = pH; p.pW = pW; p.pad = pad; p.dil = dil; p.iH = H; p.iW = W;
    p.out_mode = taps == 1 ? 0 : (padded_out ? 2 : 1); p.ldx = (unsigned)Cin * eb;
    hipStream_t hs = arseg_stream(stream);
    switch (fmt) {
        case ARSEG_ROWS_X3: return launch_cfg<0, 0>(p, tile_cfg, hs);
        case ARSEG_ROWS_F16: return launch_cfg<0, 1>(p, tile_cfg, hs);
        default: return launch_cfg<0, 2>(p, tile_cfg, hs);
    }
}

extern "C" int arseg_gemm_rows16_fwd(const void *x_rows, const void *w_rows, void *out, int dtype, long long M, int K, int Cout, int out_ld, const float *scale,
                                     const float *bias, const void *residual, int res_ld, int act, float prelu_slope, int tile_cfg, arseg_stream_t stream) {
    if (dtype != ARSEG_DT_F16 && dtype != ARSEG_DT_BF16) return ARSEG_EINVAL;
    if (M <= 0 || M >= (1ll << 31)) return ARSEG_EINVAL;
    return conv_rows(x_rows, w_rows, out, dtype == ARSEG_DT_BF16 ? ARSEG_ROWS_BF16 : ARSEG_ROWS_F16, 1, 1, (int)M, 1, K, Cout, 1, ARSEG_ROWS_OUT_NHWC, out_ld, scale, bias,
                     residual, ARSEG_ROWS_OUT_NHWC, res_ld, act, prelu_slope, tile_cfg, nullptr, 0.0f, stream);
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
