// CReFF for the 64-channel full-resolution PSPNet feature with the motion-vector warp fused in -- one kernel per non-keyframe
// batch:  warpFeature (evaluation.py:61-87, on the int16 MV map of evaluation.py:176-180)  ->  MyAttention.forward
// (model/attention.py:184-213)  ->  final 1x1 classifier + LogSoftmax (model/pspnet.py:225-229).
//
// Why a third kernel.  creff.hip / creff_mfma.hip read an already warped feature (a separate gather kernel writes and
// re-reads 2 x 134 MB per 512x1024 frame) and stage the keyframe tile twice, once for the key pass and once for the value pass,
// each time with its halo.  Here a 16x16 tile's warped keyframe region (24x24 pixels x 64 channels = 147 KB) is gathered ONCE
// from the un-warped NHWC feature and then lives in the REGISTERS of the workgroup (52 VGPRs per lane); LDS holds one thing
// at a time: the staged gather, then the query tile, then all key records, then all value records.
//
// Roles.  A workgroup is 16 waves.  Wave w plays two parts:
//   * "walker" for channel group w (4 channels): its lanes are 4 DPP rows = {x span 0..15, x span 8..23} x {upper, lower half of the
//     region}; a lane keeps its pixel column (13 rows) in registers, gets the x-1 / x+1 neighbours with row_shr:1 / row_shl:1
//     and runs the depthwise 3x3 key (later: value) convolution down the column with wave-uniform weights (scalar registers);
//     the result is written to LDS as split-fp16 records {4 hi | 4 lo};
//   * "consumer" of an 8x2 query patch (pc = w&1, pr = w>>1): Q.K^T, softmax, P.V and the classifier on
//     v_mfma_f32_16x16x32_f16 with the hi/lo split packed along K exactly as in creff_mfma.hip (two MFMAs per fp32-grade product).
//     The 7x7 windows of a patch cover 8 rows x 14 columns of keys = 112 keys, flattened into 7 MFMA row blocks of 16
//     (49 of a query's 112 slots are real; the others are masked before the softmax).
// Phases (one __syncthreads between each): taps -> gather + bilinear warp into LDS -> column registers -> lr_up tile -> query conv
// (registers) -> key records -> Q.K^T + softmax -> value records -> P.V + residual + classifier + stores.
//
// Arithmetic contract: as creff.hip (zero-padded unfold: keys / values outside the image are 0 and still take softmax mass).
#include "creff_params.h"
#include "warp_math.h"

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TX = 16, TY = 16, NT = 1024, CH = 64;
constexpr int R4W = 24, R4N = 576;                   // warped keyframe region: tile + 3 (window) + 1 (conv) halo
constexpr int R3W = 22;                              // key / value records: tile + 3 halo (22 x 22 = 484 per channel group)
constexpr int HPL = 577;                             // plane stride (float4) of the staged region: 577*4 % 32 == 4 -> the 8 lanes of a
                                                     // ds_write_b128 group (8 channel groups of one pixel) cover all 32 banks
constexpr int KPLK = 496, KPLV = 500;                // record plane strides: keys (ds_read_b128) a multiple of 16 records, values
                                                     // (transpose read) 16 banks apart
constexpr int LW = 18, LN = 324, LPL = 329;          // lr_up tile (+1 halo) per channel group
constexpr int BIG_BYTES = 16 * HPL * 16;             // 147,712: staged region / lr_up tile / key records / value records
constexpr int TAPW_OFF = BIG_BYTES;                  // [576] {ex, wx, ey, wy} with the tap validity folded in (0 = tap outside)
constexpr int TAPO_OFF = TAPW_OFF + R4N * 16;        // [576] pixel index of the NW tap | dx << 30 | dy << 31 (clamped taps)
constexpr int WFS_OFF = TAPW_OFF;                    // classifier records alias the tap tables (dead after the gather)
constexpr int TB_OFF = TAPO_OFF + R4N * 4;           // [18 rows + 18 cols] bilinear taps of the lr_up tile
constexpr int WDQ_OFF = TB_OFF + 2 * LW * 16;        // [4 chunks][9 taps + bias][4 groups] query conv weights
constexpr int SMEM_BYTES = WDQ_OFF + 4 * 10 * 4 * 16;   // 162,368 <= 163,840
constexpr int MAXN = 32;
constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr float LOG2E = 1.44269504088896340736f;

struct RRParams {
    const float *ref[MAXN];       // un-warped keyframe feature of each frame, NHWC [Hp][Wp][64]
    const int16_t *mv;            // [N][H][W][2] quarter-pel
    const float *lr, *wq, *bq, *wk, *bk, *wv, *bv, *wf, *bf;
    float *p_out, *logits;
    int N, Hp, Wp, hp, wp, H, W, n_cls, log_softmax, p_layout;
    unsigned p_bytes, l_bytes;
    float sy, sx;
};

__device__ __forceinline__ void split4(const f32x4 v, u32x2 &hi, u32x2 &lo) {
    float h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = __uint_as_float(__float_as_uint(v[j]) & 0xFFFFE000u);   // 11 significant bits: exact in fp16
        l[j] = v[j] - h[j];
    }
    hi.x = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(h[0], h[1]));
    hi.y = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(h[2], h[3]));
    lo.x = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(l[0], l[1]));
    lo.y = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(l[2], l[3]));
}
__device__ __forceinline__ h16x8 pack8(const u32x2 a, const u32x2 b) { return __builtin_bit_cast(h16x8, u32x4{a.x, a.y, b.x, b.y}); }
__device__ __forceinline__ u32x2 lds_tr16(const unsigned char *p) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)p));
}
template <int CTRL>
__device__ __forceinline__ f32x4 dpp4(const f32x4 v) {     // row_shr:1 = 0x111 (lane i <- lane i-1), row_shl:1 = 0x101 (lane i <- lane i+1)
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[j]), CTRL, 0xF, 0xF, true));
    return r;
}
__device__ __forceinline__ f32x4 fma4(const f32x4 a, const f32x4 b, f32x4 c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = fmaf(a[j], b[j], c[j]);
    return c;
}
// record index (row-major in the 22 x 22 record region) of flat window key f (0..111) of the patch at (pc, pr)
__device__ __forceinline__ int key_rec(int f, int pc, int pr) {
    const int ky = (f * 147) >> 11, kx = f - 14 * ky;        // f / 14 for f < 112
    return (2 * pr + ky) * R3W + 8 * pc + kx;
}

template <int NB>      // NB: classifier row blocks of 16 classes (0: no head)
__global__ __launch_bounds__(NT) void creff_rr_kernel(const RRParams p) {
    constexpr int NBA = NB > 0 ? NB : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4 *BIGf = reinterpret_cast<f32x4 *>(smem);
    u32x4 *BIGu = reinterpret_cast<u32x4 *>(smem);
    f32x4 *TapW = reinterpret_cast<f32x4 *>(smem + TAPW_OFF);
    unsigned *TapO = reinterpret_cast<unsigned *>(smem + TAPO_OFF);
    f32x4 *Wfs = reinterpret_cast<f32x4 *>(smem + WFS_OFF);        // [4 chunks][4 groups][NBA*16] {4 hi | 4 lo}
    f32x4 *Tb = reinterpret_cast<f32x4 *>(smem + TB_OFF);
    f32x4 *WdQ = reinterpret_cast<f32x4 *>(smem + WDQ_OFF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane & 15, g = lane >> 4, qy = q >> 3, qx = q & 7;
    const int pc = wave & 1, pr = wave >> 1;
    // XCD-aware tile order (workgroup ids go round-robin to the 8 XCDs; give each XCD a contiguous run of tiles so that the
    // halos neighbouring tiles share are fetched into one L2)
    int n, ty0, tx0;
    {
        const int tiles_x = gridDim.x, per_img = gridDim.x * gridDim.y, nblk = per_img * gridDim.z;
        int bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int qn = nblk >> 3, rn = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
        n = bid / per_img;
        const int rem = bid - n * per_img;
        ty0 = (rem / tiles_x) * TY; tx0 = (rem - (rem / tiles_x) * tiles_x) * TX;
    }
    const int Hp = p.Hp, Wp = p.Wp;
    const int yq = 2 * pr + qy, xq = 8 * pc + qx;               // this lane's query pixel, tile relative
    const int gyq = ty0 + yq, gxq = tx0 + xq;

    // ------------------------------------------------------------------ phase 0: sampling taps of the region, lr_up tables, query weights
    if (tid < R4N) {
        const int yr = tid / R4W, xr = tid - yr * R4W;
        const int gy = ty0 - 4 + yr, gx = tx0 - 4 + xr;
        f32x4 w = {0.f, 0.f, 0.f, 0.f};
        unsigned o = 0;
        if ((unsigned)gy < (unsigned)Hp && (unsigned)gx < (unsigned)Wp) {      // outside the image the region is zero (conv padding)
            double fx, fy;
            const int16_t *mvn = p.mv + (size_t)n * p.H * p.W * 2;
            if (Hp == p.H && Wp == p.W) {          // identity resize (PSPNet): (q/4 * Hp) / H == q/4 exactly
                const int16_t *m = mvn + ((size_t)gy * p.W + gx) * 2;
                fx = (double)m[0] / 4.0; fy = (double)m[1] / 4.0;
            } else {
                mv_at(mvn, p.H, p.W, Hp, Wp, gy, gx, fx, fy);
            }
            float ngx, ngy;
            norm_grid<double>(gx, gy, fx, fy, Hp, Wp, ngx, ngy);
            const Taps t = make_taps(ngx, ngy, Hp, Wp);
            const int xa = min(max(t.x0, 0), Wp - 1), xc = min(max(t.x0 + 1, 0), Wp - 1);
            const int ya = min(max(t.y0, 0), Hp - 1), yc = min(max(t.y0 + 1, 0), Hp - 1);
            o = (unsigned)(ya * Wp + xa) | ((unsigned)(xc - xa) << 30) | ((unsigned)(yc - ya) << 31);
            w = f32x4{t.vx0 ? t.ex : 0.f, t.vx1 ? t.wx : 0.f, t.vy0 ? t.ey : 0.f, t.vy1 ? t.wy : 0.f};
        }
        TapW[tid] = w; TapO[tid] = o;
    } else if (tid < R4N + 2 * LW) {
        // bilinear(align_corners=True) taps of the lr_up tile rows / columns (tile coordinate -1 .. 16):
        // {lr index of tap 0 (rows: * wp), of tap 1, weight of tap 1, inside the image}
        const int e = tid - R4N;
        const bool row = e < LW;
        const int rel = row ? e : e - LW;
        const int gc = (row ? ty0 : tx0) - 1 + rel, lim = row ? Hp : Wp;
        int i0, i1; float l1;
        arseg_src_index(row ? p.sy : p.sx, min(max(gc, 0), lim - 1), true, row ? p.hp : p.wp, i0, i1, l1);
        l1 = fminf(fmaxf(l1, 0.f), 1.f);
        Tb[e] = f32x4{__int_as_float(row ? i0 * p.wp : i0), __int_as_float(row ? i1 * p.wp : i1), l1, (unsigned)gc < (unsigned)lim ? 1.0f : 0.0f};
    } else if (tid >= 640 && tid < 640 + 160) {
        const int e = tid - 640, gg = e & 3, tp = (e >> 2) % 10, c = e / 40;
        const float *src = tp < 9 ? p.wq + (size_t)tp * CH + c * 16 + gg * 4 : p.bq + c * 16 + gg * 4;
        WdQ[e] = *reinterpret_cast<const f32x4 *>(src);
    }
    __syncthreads();

    // ------------------------------------------------------------------ phase 1: gather + bilinear warp of the region into LDS
    // unit = (region pixel, channel group): 16 lanes read one whole 256-byte pixel per tap
    {
        const float *img = p.ref[n];
        const int g16 = tid & 15, pl = tid >> 4;
        const unsigned row_off = (unsigned)Wp * CH;
#pragma unroll 1
        for (int k0 = 0; k0 < 9; k0 += 3) {
            f32x4 v[3][4], w[3];
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                const int pix = pl + 64 * (k0 + kk);
                const unsigned o = TapO[pix];
                w[kk] = TapW[pix];
                const float *a = img + (size_t)(o & 0x3FFFFFFFu) * CH + g16 * 4;
                const unsigned dxo = (o & 0x40000000u) ? CH : 0u, dyo = (o & 0x80000000u) ? row_off : 0u;
                v[kk][0] = *reinterpret_cast<const f32x4 *>(a);
                v[kk][1] = *reinterpret_cast<const f32x4 *>(a + dxo);
                v[kk][2] = *reinterpret_cast<const f32x4 *>(a + dyo);
                v[kk][3] = *reinterpret_cast<const f32x4 *>(a + dyo + dxo);
            }
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                const int pix = pl + 64 * (k0 + kk);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc += v[kk][0] * (w[kk][0] * w[kk][2]);      // same order as warp_mvq_nhwc_kernel
                acc += v[kk][1] * (w[kk][1] * w[kk][2]);
                acc += v[kk][2] * (w[kk][0] * w[kk][3]);
                acc += v[kk][3] * (w[kk][1] * w[kk][3]);
                BIGf[g16 * HPL + pix] = acc;
            }
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ phase 2: walker columns into registers
    // lane = DPP row r4 (span = r4 & 1, half = r4 >> 1) x 16 columns; channel group = wave
    const int r4 = lane >> 4, xi = lane & 15;
    const int xr = 8 * (r4 & 1) + xi;                          // region column of this lane
    const int rb = 11 * (r4 >> 1);                             // first region row of its 13-row column
    f32x4 h[13];
#pragma unroll
    for (int j = 0; j < 13; ++j) h[j] = BIGf[wave * HPL + (rb + j) * R4W + xr];
    // record column / validity of this lane's conv outputs: span 0 writes record columns 0..10, span 1 columns 11..21
    const int xk = xr - 1;
    const bool wr_ok = (r4 & 1) ? (xi >= 4 && xi <= 14) : (xi >= 1 && xi <= 11);
    const bool col_in = (unsigned)(tx0 - 3 + xk) < (unsigned)Wp;
    __syncthreads();

    // ------------------------------------------------------------------ phase 3: lr_up tile (+1 halo), all 64 channels
    {
        const float *lrn = p.lr + (size_t)n * p.hp * p.wp * CH;
        const int g16 = tid & 15, pl = tid >> 4;
#pragma unroll 1
        for (int k0 = 0; k0 < 6; k0 += 2) {
            f32x4 v[2][4]; f32x4 tyv[2], txv[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int px = min(pl + 64 * (k0 + kk), LN - 1);
                const int r = px / LW, c = px - r * LW;
                tyv[kk] = Tb[r]; txv[kk] = Tb[LW + c];
                const int r0 = __float_as_int(tyv[kk][0]), r1 = __float_as_int(tyv[kk][1]), x0 = __float_as_int(txv[kk][0]), x1 = __float_as_int(txv[kk][1]);
                const float *b = lrn + g16 * 4;
                v[kk][0] = *reinterpret_cast<const f32x4 *>(b + (size_t)(r0 + x0) * CH);
                v[kk][1] = *reinterpret_cast<const f32x4 *>(b + (size_t)(r0 + x1) * CH);
                v[kk][2] = *reinterpret_cast<const f32x4 *>(b + (size_t)(r1 + x0) * CH);
                v[kk][3] = *reinterpret_cast<const f32x4 *>(b + (size_t)(r1 + x1) * CH);
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int px = pl + 64 * (k0 + kk);
                const float ly = tyv[kk][2], lx = txv[kk][2];
                f32x4 o = (1.f - ly) * ((1.f - lx) * v[kk][0] + lx * v[kk][1]) + ly * ((1.f - lx) * v[kk][2] + lx * v[kk][3]);
                if (tyv[kk][3] * txv[kk][3] == 0.f) o = f32x4{0.f, 0.f, 0.f, 0.f};      // conv zero padding outside the image
                if (px < LN) BIGf[g16 * LPL + px] = o;
            }
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ phase 4: query conv, lane local (channels 16c + 4g .. +3 of the lane's own query)
    u32x2 qh[4], ql[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const f32x4 *w = WdQ + c * 40;
        const f32x4 *ls = BIGf + (4 * c + g) * LPL + yq * LW + xq;
        f32x4 qv = w[9 * 4 + g];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) qv = fma4(w[(dy * 3 + dx) * 4 + g], ls[dy * LW + dx], qv);
        split4(qv, qh[c], ql[c]);
        // pin the result here: LLVM otherwise sinks the whole FMA chain to its use in phase 6 and keeps the 76 loaded vectors alive
        asm volatile("" : "+v"(qh[c].x), "+v"(qh[c].y), "+v"(ql[c].x), "+v"(ql[c].y));
        __builtin_amdgcn_sched_barrier(0);      // one chunk at a time: hoisting all 36 tile reads would spill the walker columns
    }
    __syncthreads();

    // ------------------------------------------------------------------ phases 5 / 7: key (value) records of the whole region
    auto conv_records = [&](const float *wt, const float *bs, int kpl) {
        // wave-uniform weights in scalar registers.  Issued through asm: behind the barriers hipcc no longer proves the weight
        // memory unclobbered and would fetch the 40 values into VGPRs with vector loads.
        const float *wg = wt + 4 * wave, *bg = bs + 4 * wave;
        u32x4 ws[9], bsv;
        asm volatile("s_load_dwordx4 %0, %10, 0x0\n\ts_load_dwordx4 %1, %10, 0x100\n\ts_load_dwordx4 %2, %10, 0x200\n\t"
                     "s_load_dwordx4 %3, %10, 0x300\n\ts_load_dwordx4 %4, %10, 0x400\n\ts_load_dwordx4 %5, %10, 0x500\n\t"
                     "s_load_dwordx4 %6, %10, 0x600\n\ts_load_dwordx4 %7, %10, 0x700\n\ts_load_dwordx4 %8, %10, 0x800\n\t"
                     "s_load_dwordx4 %9, %11, 0x0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(ws[0]), "=&s"(ws[1]), "=&s"(ws[2]), "=&s"(ws[3]), "=&s"(ws[4]), "=&s"(ws[5]), "=&s"(ws[6]), "=&s"(ws[7]),
                       "=&s"(ws[8]), "=&s"(bsv)
                     : "s"(wg), "s"(bg) : "memory");
        f32x4 w[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) w[t] = __builtin_bit_cast(f32x4, ws[t]);
        const f32x4 bias = __builtin_bit_cast(f32x4, bsv);
        f32x4 l0 = dpp4<0x111>(h[0]), r0 = dpp4<0x101>(h[0]);
        f32x4 l1 = dpp4<0x111>(h[1]), r1 = dpp4<0x101>(h[1]);
        u32x4 *dst = BIGu + wave * kpl + xk;
#pragma unroll
        for (int j = 0; j < 11; ++j) {
            const f32x4 l2 = dpp4<0x111>(h[j + 2]), r2 = dpp4<0x101>(h[j + 2]);
            f32x4 acc = bias;
            acc = fma4(w[0], l0, acc); acc = fma4(w[1], h[j], acc); acc = fma4(w[2], r0, acc);
            acc = fma4(w[3], l1, acc); acc = fma4(w[4], h[j + 1], acc); acc = fma4(w[5], r1, acc);
            acc = fma4(w[6], l2, acc); acc = fma4(w[7], h[j + 2], acc); acc = fma4(w[8], r2, acc);
            const int yk = rb + j;
            const bool in = col_in && (unsigned)(ty0 - 3 + yk) < (unsigned)Hp;
            if (!in) acc = f32x4{0.f, 0.f, 0.f, 0.f};         // the unfold's zero padding
            u32x2 hi, lo;
            split4(acc, hi, lo);
            if (wr_ok) dst[yk * R3W] = u32x4{hi.x, hi.y, lo.x, lo.y};
            l0 = l1; r0 = r1; l1 = l2; r1 = r2;
            __builtin_amdgcn_sched_barrier(0);      // row by row: hoisting the neighbour moves of all 13 rows would spill the columns
        }
    };
    conv_records(p.wk, p.bk, KPLK);
    __syncthreads();

    // ------------------------------------------------------------------ phase 6: scores S[b][i] = q . key(16b + 4g + i), then softmax
    f32x4 S[7];
#pragma unroll
    for (int b = 0; b < 7; ++b) S[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        int krec[7];
#pragma unroll
        for (int b = 0; b < 7; ++b) krec[b] = key_rec(16 * b + q, pc, pr);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const u32x4 *ka = BIGu + (4 * c + g) * KPLK;
            const h16x8 b1 = pack8(qh[c], ql[c]), b2 = pack8(ql[c], qh[c]);
#pragma unroll
            for (int b = 0; b < 7; ++b) {
                const h16x8 a = __builtin_bit_cast(h16x8, ka[krec[b]]);
                S[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b1, S[b], 0, 0, 0);
                S[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b2, S[b], 0, 0, 0);
            }
        }
    }
    // softmax over the 49 taps (padding taps included); slot (b, i) is key f = 16b + 4g + i = (ky, kx): real iff both
    // ky - qy and kx - qx lie in [0, 6]
    float inv;
    u32x4 P[7];
    {
        float m = -INFINITY;
#pragma unroll
        for (int b = 0; b < 7; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = 16 * b + 4 * g + i, ky = (f * 147) >> 11, kx = f - 14 * ky;
                const bool ok = (unsigned)(ky - qy) <= 6u && (unsigned)(kx - qx) <= 6u;
                S[b][i] = ok ? S[b][i] : -INFINITY;
                m = fmaxf(m, S[b][i]);
            }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        const float ml = m * LOG2E;
        float z = 0.f;
#pragma unroll
        for (int b = 0; b < 7; ++b) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                S[b][i] = __builtin_amdgcn_exp2f(fmaf(S[b][i], LOG2E, -ml));     // masked slots: exp2(-inf) = 0
                z += S[b][i];
            }
            u32x2 hi, lo;
            split4(S[b], hi, lo);
            P[b] = u32x4{hi.x, hi.y, lo.x, lo.y};
        }
        z += __shfl_xor(z, 16);
        z += __shfl_xor(z, 32);
        inv = 1.0f / z;                            // applied to the weighted sum instead of the 112 weights
    }
    __syncthreads();

    // ------------------------------------------------------------------ phase 7: value records (the key records are dead)
    conv_records(p.wv, p.bv, KPLV);
    // classifier records [chunk][group][class] {4 hi | 4 lo} into the (dead) tap tables
    if (NB > 0 && tid < 4 * 4 * NBA * 16) {
        const int cls = tid % (NBA * 16), gg = (tid / (NBA * 16)) & 3, c = tid / (4 * NBA * 16);
        f32x4 wv4 = {0.f, 0.f, 0.f, 0.f};
        if (cls < p.n_cls) wv4 = *reinterpret_cast<const f32x4 *>(p.wf + (size_t)cls * CH + c * 16 + gg * 4);
        u32x2 hi, lo;
        split4(wv4, hi, lo);
        Wfs[tid] = __builtin_bit_cast(f32x4, u32x4{hi.x, hi.y, lo.x, lo.y});
    }
    __syncthreads();

    // ------------------------------------------------------------------ phase 8: P.V, residual, classifier, stores
    f32x4 lg[NBA];
#pragma unroll
    for (int nb = 0; nb < NBA; ++nb) lg[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool inq = gyq < Hp && gxq < Wp;
    const __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.p_out, 0, (int)p.p_bytes, 0x00020000);
    {
        // residual term lr_up(own pixel): bilinear taps (table rows clamp into the image)
        const f32x4 tyv = Tb[yq + 1], txv = Tb[LW + xq + 1];
        const int r0 = __float_as_int(tyv[0]), r1 = __float_as_int(tyv[1]), x0 = __float_as_int(txv[0]), x1 = __float_as_int(txv[1]);
        const float ly = tyv[2], lx = txv[2];
        const float *lrn = p.lr + (size_t)n * p.hp * p.wp * CH + 4 * g;
        int vrec[7];
#pragma unroll
        for (int b = 0; b < 7; ++b) vrec[b] = key_rec(16 * b + 4 * g + (q >> 2), pc, pr);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float *b0 = lrn + 16 * c;
            const f32x4 a00 = *reinterpret_cast<const f32x4 *>(b0 + (size_t)(r0 + x0) * CH), a01 = *reinterpret_cast<const f32x4 *>(b0 + (size_t)(r0 + x1) * CH);
            const f32x4 a10 = *reinterpret_cast<const f32x4 *>(b0 + (size_t)(r1 + x0) * CH), a11 = *reinterpret_cast<const f32x4 *>(b0 + (size_t)(r1 + x1) * CH);
            const unsigned char *va = reinterpret_cast<const unsigned char *>(BIGu + (4 * c + (q & 3)) * KPLV);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 7; ++b) {
                const u32x2 vh = lds_tr16(va + vrec[b] * 16), vl = lds_tr16(va + vrec[b] * 16 + 8);
                const h16x8 pb = __builtin_bit_cast(h16x8, P[b]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(pack8(vh, vl), pb, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(pack8(vl, vh), pb, acc, 0, 0, 0);
            }
            const f32x4 lrc = (1.f - ly) * ((1.f - lx) * a00 + lx * a01) + ly * ((1.f - lx) * a10 + lx * a11);
            const f32x4 o = lrc + acc * inv;              // p[query][16c + 4g .. +3]
            unsigned off;
            if (p.p_layout == ARSEG_C8)
                off = (unsigned)(((((size_t)n * 8 + 2 * c + (g >> 1)) * Hp + gyq) * Wp + gxq) * 8 + (g & 1) * 4) * 4u;
            else
                off = (unsigned)((((size_t)n * Hp + gyq) * Wp + gxq) * CH + 16 * c + 4 * g) * 4u;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), p_rsrc, inq ? off : OOB, 0, 0);
            if (NB > 0) {
                u32x2 oh, ol;
                split4(o, oh, ol);
                const h16x8 o1 = pack8(oh, ol), o2 = pack8(ol, oh);
#pragma unroll
                for (int nb = 0; nb < NBA; ++nb) {
                    const h16x8 wa = __builtin_bit_cast(h16x8, Wfs[(c * 4 + g) * NBA * 16 + nb * 16 + q]);
                    lg[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, o1, lg[nb], 0, 0, 0);
                    lg[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, o2, lg[nb], 0, 0, 0);
                }
            }
        }
    }

    // ------------------------------------------------------------------ logits: lg[nb][i] = class 16nb + 4g + i of query q
    if (NB > 0) {
        const __amdgpu_buffer_rsrc_t l_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.logits, 0, (int)p.l_bytes, 0x00020000);
        float m = -INFINITY;
#pragma unroll
        for (int nb = 0; nb < NBA; ++nb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cls = nb * 16 + 4 * g + i;
                lg[nb][i] += p.bf[min(cls, p.n_cls - 1)];
                m = fmaxf(m, cls < p.n_cls ? lg[nb][i] : -INFINITY);
            }
        if (p.log_softmax) {
            m = fmaxf(m, __shfl_xor(m, 16));
            m = fmaxf(m, __shfl_xor(m, 32));
            float z = 0.f;
#pragma unroll
            for (int nb = 0; nb < NBA; ++nb)
#pragma unroll
                for (int i = 0; i < 4; ++i) z += nb * 16 + 4 * g + i < p.n_cls ? expf(lg[nb][i] - m) : 0.f;
            z += __shfl_xor(z, 16);
            z += __shfl_xor(z, 32);
            const float lse = m + logf(z);
#pragma unroll
            for (int nb = 0; nb < NBA; ++nb) lg[nb] -= lse;
        }
#pragma unroll
        for (int nb = 0; nb < NBA; ++nb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cls = nb * 16 + 4 * g + i;
                const unsigned off = (unsigned)(((((size_t)n * p.n_cls + cls) * Hp + gyq) * Wp + gxq) * sizeof(float));
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(lg[nb][i]), l_rsrc, (inq && cls < p.n_cls) ? off : OOB, 0, 0);
            }
    }
}

template <int NB>
int launch(const RRParams &p, hipStream_t st) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(creff_rr_kernel<NB>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return (int)e;
    dim3 grid(arseg_cdiv(p.Wp, TX), arseg_cdiv(p.Hp, TY), p.N);
    hipLaunchKernelGGL((creff_rr_kernel<NB>), grid, dim3(NT), SMEM_BYTES, st, p);
    return arseg_launch_status();
}

}  // namespace

extern "C" int arseg_creff_warp_fwd(const float *const *ref_nhwc_host, const int16_t *mv_q, int H, int W, const float *lr,
                                    const float *wq, const float *bq, const float *wk, const float *bk, const float *wv,
                                    const float *bv, float *p_out, int p_layout, const float *wf, const float *bf, int n_cls,
                                    float *logits, int log_softmax, int N, int C, int Hp, int Wp, int hp, int wp, int kH, int kW,
                                    arseg_stream_t stream) {
    ARSEG_CHECK_PTR(ref_nhwc_host); ARSEG_CHECK_PTR(mv_q); ARSEG_CHECK_PTR(lr); ARSEG_CHECK_PTR(wq); ARSEG_CHECK_PTR(bq); ARSEG_CHECK_PTR(wk);
    ARSEG_CHECK_PTR(bk); ARSEG_CHECK_PTR(wv); ARSEG_CHECK_PTR(bv); ARSEG_CHECK_PTR(p_out);
    ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(C); ARSEG_CHECK_POS(Hp); ARSEG_CHECK_POS(Wp); ARSEG_CHECK_POS(hp); ARSEG_CHECK_POS(wp);
    ARSEG_CHECK_POS(H); ARSEG_CHECK_POS(W);
    if (C != CH || kH != 7 || kW != 7 || N > MAXN) return ARSEG_EUNSUPPORTED;
    if (p_layout != ARSEG_C8 && p_layout != ARSEG_NHWC) return ARSEG_EINVAL;
    if ((size_t)N * C * Hp * Wp * sizeof(float) >= (1ull << 31)) return ARSEG_EUNSUPPORTED;   // 32-bit buffer offsets
    if ((size_t)Hp * Wp >= (1u << 30)) return ARSEG_EUNSUPPORTED;                              // tap index packing
    if (!ARSEG_ALIGNED16(lr) || !ARSEG_ALIGNED16(p_out) || !ARSEG_ALIGNED16(wq) || !ARSEG_ALIGNED16(wk) || !ARSEG_ALIGNED16(wv) ||
        !ARSEG_ALIGNED16(bq) || !ARSEG_ALIGNED16(bk) || !ARSEG_ALIGNED16(bv))
        return ARSEG_EINVAL;
    const bool head = logits != nullptr;
    if (head) {
        if (!wf || !bf || n_cls <= 0) return ARSEG_EINVAL;
        if (n_cls > 32) return ARSEG_EUNSUPPORTED;
        if (!ARSEG_ALIGNED16(wf)) return ARSEG_EINVAL;
    }
    RRParams p;
    for (int i = 0; i < N; ++i) {
        if (!ref_nhwc_host[i] || !ARSEG_ALIGNED16(ref_nhwc_host[i])) return ARSEG_EINVAL;
        p.ref[i] = ref_nhwc_host[i];
    }
    for (int i = N; i < MAXN; ++i) p.ref[i] = nullptr;
    p.mv = mv_q; p.lr = lr; p.wq = wq; p.bq = bq; p.wk = wk; p.bk = bk; p.wv = wv; p.bv = bv; p.wf = wf; p.bf = bf;
    p.p_out = p_out; p.logits = logits;
    p.N = N; p.Hp = Hp; p.Wp = Wp; p.hp = hp; p.wp = wp; p.H = H; p.W = W; p.n_cls = head ? n_cls : 0; p.log_softmax = log_softmax;
    p.p_layout = p_layout;
    p.p_bytes = (unsigned)((size_t)N * C * Hp * Wp * sizeof(float)); p.l_bytes = head ? (unsigned)((size_t)N * n_cls * Hp * Wp * sizeof(float)) : 0u;
    p.sy = arseg_resize_scale(hp, Hp, true); p.sx = arseg_resize_scale(wp, Wp, true);
    hipStream_t st = arseg_stream(stream);
    if (!head) return launch<0>(p, st);
    return n_cls <= 16 ? launch<1>(p, st) : launch<2>(p, st);
}
