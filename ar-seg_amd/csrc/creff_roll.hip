// CReFF for the 64-channel full-resolution PSPNet feature, ROLLING form -- one kernel per non-keyframe batch:
// warpFeature (evaluation.py:61-87, on the int16 MV map of evaluation.py:176-180)  ->  MyAttention.forward
// (model/attention.py:184-213)  ->  final 1x1 classifier + LogSoftmax (model/pspnet.py:225-229).
//
// Why a fourth kernel.  creff_rr.hip works on 16 x 16 tiles whose working set IS the compute unit (147 KB of LDS, key records OR value
// records, never both), so its nine phases run one after the other and the VALU, the texture path, the LDS and the matrix pipe are busy
// in turn (profiles/r03_creff_ablation.json).  Here a workgroup walks DOWN a 16-pixel-wide strip two rows at a time and keeps only what
// the 7 x 7 windows of the current row pair need: 8 rows of key records and 10 rows of value records in two LDS rings (101 KB).  The
// halo shrinks from 2.25x (24 x 24 region per 16 x 16 tile) to 1.5x (gather) / 1.375x (records), and -- the point -- the stages of
// DIFFERENT row pairs run at the same time on different waves:
//
//   producer waves (6)                                             consumer waves (2, one per 8-column query patch)
//   H1(t): taps of gather t+1 | blend gather t -> warp stage       H1(t): Q.K^T + softmax of step s = t - 5   (key ring, Q records)
//          lr_up rows of step t-4 -> lr_up stage
//   ---------------------------------------------------------------- barrier A
//   H2(t): key + value conv k = t-1 -> rings | query conv s = t-4   H2(t): P.V + residual + classifier + stores of step s
//          requests of gather t+1 (registers) | MVs of gather t+2
//   ---------------------------------------------------------------- barrier B
//
// so the depthwise convolutions and the gather of rows further down run under the MFMAs of the rows being finished.  A producer lane
// owns one (column, channel group) of the strip for the whole segment and keeps a 2-row window of its 3-column neighbourhood in
// registers (the third row of a 3 x 3 stencil is the staged one): every warped / upsampled value is written to LDS once and read three times.
//
// The consumer arithmetic (hi/lo split-fp16 operands on v_mfma_f32_16x16x32_f16, 8 x 14 key window per 2 x 8 query patch flattened
// into 7 blocks of 16, softmax over all 49 taps incl. padding taps) is that of creff_rr.hip; only the record addresses differ (rings).
//
// Arithmetic contract: as creff.hip (zero-padded unfold: keys / values outside the image are 0 and still take softmax mass).
#include "creff_params.h"
#ifndef ROLL_PNT
#define ROLL_PNT 2            // cache policy of the p / logits stores: 2 = nontemporal
#endif
#include "warp_math.h"

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int CH = 64, NT = 512, NCONS = 2, NPL = NT - 64 * NCONS;      // 384 producer lanes
constexpr int SW = 16;                               // query columns of a strip (two 8-column patches)
constexpr int RW = SW + 6;                           // key / value record columns (22)
constexpr int GW = SW + 8;                           // warped keyframe columns (24): + 1 for the depthwise convs
constexpr int LW = SW + 2;                           // lr_up columns (18)
constexpr int KSLOT = 8, VSLOT = 10;                 // ring rows: keys are dead after H1, values after H2 of the step that read them last
constexpr int KPL = KSLOT * RW, VPL = VSLOT * RW;    // records per channel-group plane (176: a multiple of 16 -- see creff_rr.hip; 220)
constexpr int WPL = GW + 1, LPL = LW + 1;            // plane pitch of the two stages in f32x4 (25 * 16 B = 16 mod 128, 19 * 16 B = 48 mod 128:
                                                     // the 8 lanes of a ds_write_b128 group -- 8 channel groups of one pixel -- cover all 32 banks)
constexpr int NGP = 2 * GW, NLP = 2 * LW;            // staged pixels per iteration: 48 warped, 36 lr_up
constexpr int K_OFF = 0;
constexpr int V_OFF = K_OFF + 16 * KPL * 16;         //  45,056
constexpr int WS_OFF = V_OFF + 16 * VPL * 16;        // 101,376  warp stage [2 rows][16 groups][WPL]
constexpr int LS_OFF = WS_OFF + 2 * 16 * WPL * 16;   // 114,176  lr_up stage [2 rows][16 groups][LPL]
constexpr int Q_OFF = LS_OFF + 2 * 16 * LPL * 16;    // 123,904  query records [2 patches][4 chunks][4 groups][16 queries] {4 hi | 4 lo}
constexpr int TW_OFF = Q_OFF + 2 * 4 * 4 * 16 * 16;  // 132,096  [2][48] {ex, wx, ey, wy} with the tap validity folded in
constexpr int TO_OFF = TW_OFF + 2 * NGP * 16;        // 133,632  [2][48] pixel index of the NW tap | dx << 30 | dy << 31 (clamped taps)
constexpr int WD_OFF = TO_OFF + 2 * NGP * 4;         // 134,016  depthwise weights [key | value | query][16 groups][9 taps + bias]
constexpr int WF_OFF = WD_OFF + 3 * 160 * 16;        // 141,696  classifier records [4 chunks][4 groups][32] {4 hi | 4 lo}
constexpr int BF_OFF = WF_OFF + 4 * 4 * 32 * 16;     // 149,888  classifier bias [32]
constexpr int SMEM_BYTES = BF_OFF + 32 * 4;          // 150,016 <= 163,840
constexpr int MAXN = 32;
constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr float LOG2E = 1.44269504088896340736f;

struct RollParams {
    const float *ref[MAXN];       // un-warped keyframe feature of each frame, NHWC [Hp][Wp][64]
    const int16_t *mv;            // [N][H][W][2] quarter-pel
    const float *lr, *wq, *bq, *wk, *bk, *wv, *bv, *wf, *bf;
    float *p_out, *logits;
    int N, Hp, Wp, hp, wp, H, W, n_cls, log_softmax, p_layout, nstrips, nseg, seg_rows;
    unsigned p_bytes, l_bytes, lr_bytes, ref_bytes;
    float sy, sx;
};

__device__ __forceinline__ void split4(const f32x4 v, u32x2 &hi, u32x2 &lo) {
    unsigned h01, h23, l01, l23;
    arseg_split_f16(v, h01, h23, l01, l23);
    hi = u32x2{h01, h23}; lo = u32x2{l01, l23};
}
__device__ __forceinline__ h16x8 pack8(const u32x2 a, const u32x2 b) { return __builtin_bit_cast(h16x8, u32x4{a.x, a.y, b.x, b.y}); }
__device__ __forceinline__ u32x2 lds_tr16(const unsigned char *p) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)p));
}
// a * b + c on packed pairs (v_pk_fma_f32: 2 FMAs per issue slot)
__device__ __forceinline__ f32x4 fma4(const f32x4 a, const f32x4 b, const f32x4 c) {
    const f32x2 lo = __builtin_elementwise_fma(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1), __builtin_shufflevector(c, c, 0, 1));
    const f32x2 hi = __builtin_elementwise_fma(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3), __builtin_shufflevector(c, c, 2, 3));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
// reductions over the 4 DPP rows of a wave (lanes l, l^16, l^32, l^48) on the VALU (see creff_rr.hip)
__device__ __forceinline__ float rows_max(float x) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows_sum(float x) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ double uniform_f64(double x) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// Workgroup barrier that orders LDS traffic only: global loads requested before it stay in flight across it (the gather of the next
// row pair travels under the convolutions of this one).
__device__ __forceinline__ void wg_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// 3 x 3 depthwise stencil on 4 channels: rows a (above), b, c (below), each {left, centre, right}; weights w[0..8] + bias w[9]
// (accumulation order of creff_rr.hip / creff.hip)
__device__ __forceinline__ f32x4 stencil(const f32x4 *w, const f32x4 (&a)[3], const f32x4 (&b)[3], const f32x4 (&c)[3]) {
    f32x4 acc = w[9];
    acc = fma4(w[0], a[0], acc); acc = fma4(w[1], a[1], acc); acc = fma4(w[2], a[2], acc);
    acc = fma4(w[3], b[0], acc); acc = fma4(w[4], b[1], acc); acc = fma4(w[5], b[2], acc);
    acc = fma4(w[6], c[0], acc); acc = fma4(w[7], c[1], acc); acc = fma4(w[8], c[2], acc);
    return acc;
}

template <int NB>      // NB: classifier row blocks of 16 classes (0: no head)
__global__ __launch_bounds__(NT) void creff_roll_kernel(const RollParams p) {
    constexpr int NBA = NB > 0 ? NB : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4 *Kr = reinterpret_cast<u32x4 *>(smem + K_OFF);
    u32x4 *Vr = reinterpret_cast<u32x4 *>(smem + V_OFF);
    f32x4 *Ws = reinterpret_cast<f32x4 *>(smem + WS_OFF);
    f32x4 *Ls = reinterpret_cast<f32x4 *>(smem + LS_OFF);
    u32x4 *Qr = reinterpret_cast<u32x4 *>(smem + Q_OFF);
    f32x4 *TapW = reinterpret_cast<f32x4 *>(smem + TW_OFF);
    unsigned *TapO = reinterpret_cast<unsigned *>(smem + TO_OFF);
    f32x4 *Wd = reinterpret_cast<f32x4 *>(smem + WD_OFF);
    f32x4 *Wfs = reinterpret_cast<f32x4 *>(smem + WF_OFF);        // [4 chunks][4 groups][NBA*16]
    float *Bfs = reinterpret_cast<float *>(smem + BF_OFF);

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Hp = p.Hp, Wp = p.Wp;

    // ------------------------------------------------------------------ once per launch: weight tables
    for (int e = tid; e < 3 * 160; e += NT) {
        const int which = e / 160, r = e - which * 160, cg = r / 10, tp = r - cg * 10;
        const float *w = which == 0 ? p.wk : which == 1 ? p.wv : p.wq, *b = which == 0 ? p.bk : which == 1 ? p.bv : p.bq;
        Wd[e] = *reinterpret_cast<const f32x4 *>(tp < 9 ? w + tp * CH + cg * 4 : b + cg * 4);
    }
    if (NB > 0) {
        for (int e = tid; e < 4 * 4 * NBA * 16; e += NT) {
            const int cls = e % (NBA * 16), gg = (e / (NBA * 16)) & 3, c = e / (4 * NBA * 16);
            f32x4 wv4 = {0.f, 0.f, 0.f, 0.f};
            if (cls < p.n_cls) wv4 = *reinterpret_cast<const f32x4 *>(p.wf + (size_t)cls * CH + c * 16 + gg * 4);
            u32x2 hi, lo;
            split4(wv4, hi, lo);
            Wfs[e] = __builtin_bit_cast(f32x4, u32x4{hi.x, hi.y, lo.x, lo.y});
        }
        if (tid < NBA * 16) Bfs[tid] = tid < p.n_cls ? p.bf[tid] : 0.f;
    }
    __syncthreads();

    // Persistent workgroups, XCD-aware order (as creff_rr.hip): XCD x owns a contiguous run of units, its workgroups take neighbouring
    // strips of one segment row at the same time, so the halo columns they share are fetched into one L2 once.
    const int per_img = p.nstrips * p.nseg, nunits = per_img * p.N;
    const int nx = min(8, (int)gridDim.x);
    const int xcd = blockIdx.x % nx, slot = blockIdx.x / nx, nslot = ((int)gridDim.x - xcd + nx - 1) / nx;
    const int u_lo = (int)((long long)nunits * xcd / nx), u_hi = (int)((long long)nunits * (xcd + 1) / nx);

    if (wave < NCONS) {
        // ====================================================================================== consumer: one 2 x 8 query patch per step
        const int lane = tid & 63, q = lane & 15, g = lane >> 4, pc = wave;
        // window key f = 16b + k0 = 14 ky + kx of the 8 x 14 patch window (k0 = this lane's key of block b): ky = b + (2b + k0 >= 14)
        int kky[7], kkx[7], vky[7], vkx[7];
        f32x4 maskv[7];                              // 0 where slot (b, i) = key 16b + 4g + i is a real window tap of query q, -inf elsewhere:
        {                                            // the scores are accumulated ON TOP of it (no select per slot and step)
            const int vk0 = 4 * g + (q >> 2), qy = q >> 3, qx = q & 7;
#pragma unroll
            for (int b = 0; b < 7; ++b) {
                int e = 2 * b + q, up = e >= 14;
                kky[b] = b + up; kkx[b] = e - 14 * up + 8 * pc;
                e = 2 * b + vk0; up = e >= 14;
                vky[b] = b + up; vkx[b] = e - 14 * up + 8 * pc;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int f = 16 * b + 4 * g + i, ky = f / 14, kx = f - 14 * ky;
                    maskv[b][i] = ((unsigned)(ky - qy) <= 6u && (unsigned)(kx - qx) <= 6u) ? 0.f : -INFINITY;
                }
            }
        }
        for (int unit = u_lo + slot; unit < u_hi; unit += nslot) {
            const int n = unit / per_img, rem = unit - n * per_img;
            const int seg = rem / p.nstrips, strip = rem - seg * p.nstrips;
            const int x0 = strip * SW, ys = seg * p.seg_rows;
            const int S = (min(p.seg_rows, Hp - ys) + 1) >> 1;
            const __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.p_out, 0, (int)p.p_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t lr_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.lr), 0, (int)p.lr_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t l_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.logits, 0, (int)p.l_bytes, 0x00020000);
            for (int t = -2; t <= S + 4; ++t) {
                const int s = t - 5;
                float inv = 0.f;
                u32x4 P[7];
                // ---------------------------------------------------------------- H1: scores S[b][i] = q . key(16b + 4g + i), softmax
                if (s >= 0) {
                    u32x2 qh[4], ql[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const u32x4 v = Qr[((pc * 4 + c) * 4 + g) * 16 + q];
                        qh[c] = u32x2{v.x, v.y}; ql[c] = u32x2{v.z, v.w};
                    }
                    const int b8 = (2 * s) & 7;
                    int krec[7];
#pragma unroll
                    for (int b = 0; b < 7; ++b) krec[b] = ((kky[b] + b8) & 7) * RW + kkx[b];
                    f32x4 Sc[7];
#pragma unroll
                    for (int b = 0; b < 7; ++b) Sc[b] = maskv[b];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const u32x4 *ka = Kr + (4 * c + g) * KPL;
                        const h16x8 b1 = pack8(qh[c], ql[c]), b2 = pack8(ql[c], qh[c]);
#pragma unroll
                        for (int b = 0; b < 7; ++b) {
                            const h16x8 a = __builtin_bit_cast(h16x8, ka[krec[b]]);
                            Sc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b1, Sc[b], 0, 0, 0);
                            Sc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b2, Sc[b], 0, 0, 0);
                        }
                    }
                    float m = -INFINITY;
#pragma unroll
                    for (int b = 0; b < 7; ++b)
#pragma unroll
                        for (int i = 0; i < 4; ++i) m = fmaxf(m, Sc[b][i]);
                    m = rows_max(m);
                    const float ml = m * LOG2E;
                    float z = 0.f;
#pragma unroll
                    for (int b = 0; b < 7; ++b) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            Sc[b][i] = __builtin_amdgcn_exp2f(fmaf(Sc[b][i], LOG2E, -ml));      // masked slots: exp2(-inf) = 0
                            z += Sc[b][i];
                        }
                        u32x2 hi, lo;
                        split4(Sc[b], hi, lo);
                        P[b] = u32x4{hi.x, hi.y, lo.x, lo.y};
                    }
                    z = rows_sum(z);
                    inv = 1.0f / z;                            // applied to the weighted sum instead of the 112 weights
                }
                wg_sync();
                // ---------------------------------------------------------------- H2: P.V, residual, classifier, stores
                if (s >= 0) {
                    const int gyq = ys + 2 * s + (q >> 3), gxq = x0 + 8 * pc + (q & 7);       // this lane's query pixel
                    const bool inq = gyq < Hp && gxq < Wp;
                    const unsigned pix = (unsigned)(gyq * Wp + gxq), plane = (unsigned)(Hp * Wp);
                    const unsigned p_off0 = p.p_layout == ARSEG_C8 ? (((unsigned)n * 8u + (unsigned)(g >> 1)) * plane + pix) * 32u + (unsigned)(g & 1) * 16u
                                                                   : ((unsigned)n * plane + pix) * (CH * 4u) + 16u * g;
                    const unsigned p_step = p.p_layout == ARSEG_C8 ? 2u * plane * 32u : 64u;      // chunk c: + c * p_step
                    const unsigned l_off0 = ((unsigned)n * (unsigned)p.n_cls * plane + pix) * 4u;
                    f32x4 lg[NBA];
#pragma unroll
                    for (int nb = 0; nb < NBA; ++nb) lg[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
                    // residual term lr_up(own pixel): bilinear(align_corners=True) taps, coordinates clamped into the image
                    int y0, y1, xx0, xx1; float ly, lx;
                    arseg_src_index(p.sy, min(gyq, Hp - 1), true, p.hp, y0, y1, ly);
                    arseg_src_index(p.sx, min(gxq, Wp - 1), true, p.wp, xx0, xx1, lx);
                    ly = fminf(fmaxf(ly, 0.f), 1.f); lx = fminf(fmaxf(lx, 0.f), 1.f);
                    const unsigned lrb = (unsigned)n * (unsigned)(p.hp * p.wp) * (CH * 4u) + 16u * g;
                    const unsigned o00 = lrb + (unsigned)(y0 * p.wp + xx0) * (CH * 4u), o01 = lrb + (unsigned)(y0 * p.wp + xx1) * (CH * 4u);
                    const unsigned o10 = lrb + (unsigned)(y1 * p.wp + xx0) * (CH * 4u), o11 = lrb + (unsigned)(y1 * p.wp + xx1) * (CH * 4u);
                    auto lr_tap = [&](unsigned o, int c) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lr_rsrc, o, 64 * c, 0)); };
                    // byte offset of the value record of (block b, this lane's key) in a channel-group plane
                    const int b2s = 2 * s, b10 = b2s - 10 * (b2s / 10);
                    unsigned vrec[7];
#pragma unroll
                    for (int b = 0; b < 7; ++b) {
                        const unsigned r0 = (unsigned)(vky[b] + b10);
                        vrec[b] = (min(r0, r0 - 10u) * RW + (unsigned)vkx[b]) * 16u;
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const f32x4 a00 = lr_tap(o00, c), a01 = lr_tap(o01, c), a10 = lr_tap(o10, c), a11 = lr_tap(o11, c);
                        const unsigned char *va = reinterpret_cast<const unsigned char *>(Vr + (4 * c + (q & 3)) * VPL);
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int b = 0; b < 7; ++b) {
                            const u32x2 vh = lds_tr16(va + vrec[b]), vl = lds_tr16(va + vrec[b] + 8);
                            const h16x8 pb = __builtin_bit_cast(h16x8, P[b]);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(pack8(vh, vl), pb, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(pack8(vl, vh), pb, acc, 0, 0, 0);
                        }
                        const f32x4 lrc = (1.f - ly) * ((1.f - lx) * a00 + lx * a01) + ly * ((1.f - lx) * a10 + lx * a11);
                        const f32x4 o = lrc + acc * inv;       // p[query][16c + 4g .. +3]: store, then this chunk's share of the classifier
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), p_rsrc, inq ? p_off0 + (unsigned)c * p_step : OOB, 0, ROLL_PNT);
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
                    // logits: lg[nb][i] = class 16nb + 4g + i of query q
                    if (NB > 0) {
                        float m = -INFINITY;
#pragma unroll
                        for (int nb = 0; nb < NBA; ++nb)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int cls = nb * 16 + 4 * g + i;
                                lg[nb][i] += Bfs[cls];
                                m = fmaxf(m, cls < p.n_cls ? lg[nb][i] : -INFINITY);
                            }
                        if (p.log_softmax) {
                            m = rows_max(m);
                            float z = 0.f;
#pragma unroll
                            for (int nb = 0; nb < NBA; ++nb)
#pragma unroll
                                for (int i = 0; i < 4; ++i) z += nb * 16 + 4 * g + i < p.n_cls ? __expf(lg[nb][i] - m) : 0.f;
                            z = rows_sum(z);
                            const float lse = m + __logf(z);
#pragma unroll
                            for (int nb = 0; nb < NBA; ++nb) lg[nb] -= lse;
                        }
#pragma unroll
                        for (int nb = 0; nb < NBA; ++nb)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int cls = nb * 16 + 4 * g + i;
                                const unsigned off = l_off0 + (unsigned)cls * plane * 4u;
                                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(lg[nb][i]), l_rsrc, (inq && cls < p.n_cls) ? off : OOB, 0, ROLL_PNT);
                            }
                    }
                }
                wg_sync();
            }
        }
    } else {
        // ====================================================================================== producers
        const int pl = tid - 64 * NCONS;                   // 0 .. 383
        const bool kv_ok = pl < 16 * RW;                   // key / value lane: (channel group, record column), column fastest
        const int kcg = min(pl / RW, 15), kx = pl - RW * (pl / RW);
        const bool q_ok = pl < 16 * SW;                    // query lane: (channel group, query column)
        const int qcg = (pl >> 4) & 15, qx = pl & 15;
        const int tl = pl - (NPL - 64);                    // tap lane: the last producer wave's lanes 0 .. 47, one per staged warp pixel
        const bool tap_lane = tl >= 0 && tl < NGP;
        const bool mv_ident = Hp == p.H && Wp == p.W;
        const double g_dW = uniform_f64((double)max(Wp - 1, 1)), g_dH = uniform_f64((double)max(Hp - 1, 1));
        const double g_rW = uniform_f64(1.0 / g_dW), g_rH = uniform_f64(1.0 / g_dH);      // grid normalisation: extents and their reciprocals
        const f32x4 *wK = Wd + kcg * 10, *wV = Wd + 160 + kcg * 10, *wQ = Wd + 320 + qcg * 10;

        for (int unit = u_lo + slot; unit < u_hi; unit += nslot) {
            const int n = unit / per_img, rem = unit - n * per_img;
            const int seg = rem / p.nstrips, strip = rem - seg * p.nstrips;
            const int x0 = strip * SW, ys = seg * p.seg_rows;
            const int S = (min(p.seg_rows, Hp - ys) + 1) >> 1;
            const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.ref[n]), 0, (int)p.ref_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t lr_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.lr), 0, (int)p.lr_bytes, 0x00020000);
            const unsigned lr_img = (unsigned)n * (unsigned)(p.hp * p.wp) * (CH * 4u);
            f32x4 kvw[2][3], qw[2][3], gv[2][4];
            unsigned mvv = 0u;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) { kvw[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; qw[i][j] = kvw[i][j]; }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) gv[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

            for (int t = -2; t <= S + 4; ++t) {
                // ================================================================ H1
                // ---- lr_up rows ys + 2t - 7, ys + 2t - 6 (+1 halo column each side): request the bilinear taps first
                const bool l_on = t >= 3 && t <= S + 3;
                f32x4 lv[2][4];
                float lwy0[2], lwy1[2], lwx0[2], lwx1[2];
                int ldst[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int v = pl + NPL * i, px = v >> 4, cg = v & 15;
                    const bool on = l_on && px < NLP;
                    const int rr = px >= LW ? 1 : 0, cc = px - LW * rr;
                    const int gy = ys + 2 * t - 7 + rr, gx = x0 - 1 + cc;
                    int i0, i1, j0, j1; float l, m;
                    arseg_src_index(p.sy, min(max(gy, 0), Hp - 1), true, p.hp, i0, i1, l);
                    arseg_src_index(p.sx, min(max(gx, 0), Wp - 1), true, p.wp, j0, j1, m);
                    l = fminf(fmaxf(l, 0.f), 1.f); m = fminf(fmaxf(m, 0.f), 1.f);
                    const float iny = (unsigned)gy < (unsigned)Hp ? 1.f : 0.f, inx = (unsigned)gx < (unsigned)Wp ? 1.f : 0.f;
                    lwy0[i] = (1.f - l) * iny; lwy1[i] = l * iny; lwx0[i] = (1.f - m) * inx; lwx1[i] = m * inx;
                    ldst[i] = on ? (rr * 16 + cg) * LPL + cc : -1;
                    const unsigned b = lr_img + 16u * cg;
                    const unsigned o00 = b + (unsigned)(i0 * p.wp + j0) * (CH * 4u), o01 = b + (unsigned)(i0 * p.wp + j1) * (CH * 4u);
                    const unsigned o10 = b + (unsigned)(i1 * p.wp + j0) * (CH * 4u), o11 = b + (unsigned)(i1 * p.wp + j1) * (CH * 4u);
                    lv[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lr_rsrc, on ? o00 : OOB, 0, 0));
                    lv[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lr_rsrc, on ? o01 : OOB, 0, 0));
                    lv[i][2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lr_rsrc, on ? o10 : OOB, 0, 0));
                    lv[i][3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lr_rsrc, on ? o11 : OOB, 0, 0));
                }
                // ---- sampling taps of gather t + 1 (warp rows ys - 4 + 2(t+1), +1), one lane per pixel; the motion vector was requested in H2(t-1)
                if (tap_lane && t >= -1 && t <= S + 2) {
                    const int rr = tl >= GW ? 1 : 0, cc = tl - GW * rr;
                    const int gy = ys - 2 + 2 * t + rr, gx = x0 - 4 + cc;
                    f32x4 w = {0.f, 0.f, 0.f, 0.f};
                    unsigned o = 0;
                    if ((unsigned)gy < (unsigned)Hp && (unsigned)gx < (unsigned)Wp) {      // outside the image the warped feature is zero (conv padding)
                        double fx, fy;
                        if (mv_ident) {                        // identity resize (PSPNet): (q/4 * Hp) / H == q/4 exactly
                            fx = (double)(short)(mvv & 0xFFFFu) / 4.0; fy = (double)(short)(mvv >> 16) / 4.0;
                        } else {
                            mv_at(p.mv + (size_t)n * p.H * p.W * 2, p.H, p.W, Hp, Wp, gy, gx, fx, fy);
                        }
                        float ngx, ngy;
                        norm_grid_rcp(gx, gy, fx, fy, g_dW, g_dH, g_rW, g_rH, ngx, ngy);
                        const Taps tp = make_taps(ngx, ngy, Hp, Wp);
                        const int xa = min(max(tp.x0, 0), Wp - 1), xc = min(max(tp.x0 + 1, 0), Wp - 1);
                        const int ya = min(max(tp.y0, 0), Hp - 1), yc = min(max(tp.y0 + 1, 0), Hp - 1);
                        o = (unsigned)(ya * Wp + xa) | ((unsigned)(xc - xa) << 30) | ((unsigned)(yc - ya) << 31);
                        w = f32x4{tp.vx0 ? tp.ex : 0.f, tp.vx1 ? tp.wx : 0.f, tp.vy0 ? tp.ey : 0.f, tp.vy1 ? tp.wy : 0.f};
                    }
                    const int e = ((t + 1) & 1) * NGP + tl;
                    TapW[e] = w; TapO[e] = o;
                }
                // ---- gather t: blend the four taps requested in H2(t-1), stage
                if (t >= 0 && t <= S + 3) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int v = pl + NPL * i, px = v >> 4, cg = v & 15;
                        const f32x4 w = TapW[(t & 1) * NGP + px];
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                        acc += gv[i][0] * (w[0] * w[2]);      // same order as warp_mvq_nhwc_kernel
                        acc += gv[i][1] * (w[1] * w[2]);
                        acc += gv[i][2] * (w[0] * w[3]);
                        acc += gv[i][3] * (w[1] * w[3]);
                        const int rr = px >= GW ? 1 : 0, cc = px - GW * rr;
                        Ws[(rr * 16 + cg) * WPL + cc] = acc;
                    }
                }
                // ---- lr_up: interpolate, stage
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    if (ldst[i] >= 0)
                        Ls[ldst[i]] = lwy0[i] * (lwx0[i] * lv[i][0] + lwx1[i] * lv[i][1]) + lwy1[i] * (lwx0[i] * lv[i][2] + lwx1[i] * lv[i][3]);
                wg_sync();
                // ================================================================ H2
                // ---- key + value records of rows rho = 2k, 2k + 1 (k = t - 1; image rows ys - 3 + rho)
                if (kv_ok && t >= 0 && t <= S + 3) {
                    f32x4 n0[3], n1[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) { n0[j] = Ws[kcg * WPL + kx + j]; n1[j] = Ws[(16 + kcg) * WPL + kx + j]; }
                    if (t >= 1) {
                        const int k = t - 1, r0 = ys - 3 + 2 * k;
                        const bool col_in = (unsigned)(x0 - 3 + kx) < (unsigned)Wp;
                        f32x4 wt[10];
#pragma unroll
                        for (int j = 0; j < 10; ++j) wt[j] = wK[j];
                        f32x4 ka = stencil(wt, kvw[0], kvw[1], n0), kb = stencil(wt, kvw[1], n0, n1);
#pragma unroll
                        for (int j = 0; j < 10; ++j) wt[j] = wV[j];
                        f32x4 va = stencil(wt, kvw[0], kvw[1], n0), vb = stencil(wt, kvw[1], n0, n1);
                        // the unfold's zero padding: records outside the image are zero
                        const bool in_a = col_in && (unsigned)r0 < (unsigned)Hp, in_b = col_in && (unsigned)(r0 + 1) < (unsigned)Hp;
                        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                        ka = in_a ? ka : zero; va = in_a ? va : zero; kb = in_b ? kb : zero; vb = in_b ? vb : zero;
                        u32x2 hi, lo;
                        const int ks = (2 * k) & 7, vs = 2 * k - 10 * ((2 * k) / 10);
                        split4(ka, hi, lo); Kr[kcg * KPL + ks * RW + kx] = u32x4{hi.x, hi.y, lo.x, lo.y};
                        split4(kb, hi, lo); Kr[kcg * KPL + (ks + 1) * RW + kx] = u32x4{hi.x, hi.y, lo.x, lo.y};
                        split4(va, hi, lo); Vr[kcg * VPL + vs * RW + kx] = u32x4{hi.x, hi.y, lo.x, lo.y};
                        split4(vb, hi, lo); Vr[kcg * VPL + (vs + 1) * RW + kx] = u32x4{hi.x, hi.y, lo.x, lo.y};
                    }
#pragma unroll
                    for (int j = 0; j < 3; ++j) { kvw[0][j] = n0[j]; kvw[1][j] = n1[j]; }
                }
                // ---- query records of step s = t - 4 (query rows ys + 2s, + 1)
                if (q_ok && l_on) {
                    f32x4 m0[3], m1[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) { m0[j] = Ls[qcg * LPL + qx + j]; m1[j] = Ls[(16 + qcg) * LPL + qx + j]; }
                    if (t >= 4) {
                        f32x4 wt[10];
#pragma unroll
                        for (int j = 0; j < 10; ++j) wt[j] = wQ[j];
                        const f32x4 qa = stencil(wt, qw[0], qw[1], m0), qb = stencil(wt, qw[1], m0, m1);
                        u32x4 *dst = Qr + (((qx >> 3) * 4 + (qcg >> 2)) * 4 + (qcg & 3)) * 16 + (qx & 7);
                        u32x2 hi, lo;
                        split4(qa, hi, lo); dst[0] = u32x4{hi.x, hi.y, lo.x, lo.y};
                        split4(qb, hi, lo); dst[8] = u32x4{hi.x, hi.y, lo.x, lo.y};
                    }
#pragma unroll
                    for (int j = 0; j < 3; ++j) { qw[0][j] = m0[j]; qw[1][j] = m1[j]; }
                }
                // ---- requests of gather t + 1 (held in registers across the barrier: the first touch of a keyframe row comes from HBM)
                if (t >= -1 && t <= S + 2) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int v = pl + NPL * i, px = v >> 4, cg = v & 15;
                        const unsigned o = TapO[((t + 1) & 1) * NGP + px];
                        const unsigned a = (o & 0x3FFFFFFFu) * (CH * 4u) + 16u * cg;
                        const unsigned dxo = (o & 0x40000000u) ? CH * 4u : 0u, dyo = (o & 0x80000000u) ? (unsigned)Wp * (CH * 4u) : 0u;
                        gv[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, a, 0, 0));
                        gv[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, a + dxo, 0, 0));
                        gv[i][2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, a + dyo, 0, 0));
                        gv[i][3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, a + dyo + dxo, 0, 0));
                    }
                }
                // ---- motion vectors of gather t + 2 (identity-resize case: one int16 pair per pixel)
                if (tap_lane && mv_ident && t <= S + 1) {
                    const int rr = tl >= GW ? 1 : 0, cc = tl - GW * rr;
                    const int gy = ys + 2 * t + rr, gx = x0 - 4 + cc;
                    mvv = 0u;
                    if ((unsigned)gy < (unsigned)Hp && (unsigned)gx < (unsigned)Wp)
                        mvv = *reinterpret_cast<const unsigned *>(p.mv + ((size_t)n * p.H * p.W + (size_t)gy * p.W + gx) * 2);
                }
                wg_sync();
            }
        }
    }
}

template <int NB>
int launch(const RollParams &p, hipStream_t st) {
    static ArsegSmemAttr attr;
    if (int e = arseg_allow_smem(attr, reinterpret_cast<const void *>(creff_roll_kernel<NB>), SMEM_BYTES)) return e;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const long long nunits = (long long)p.nstrips * p.nseg * p.N;
    const int grid = (int)(nunits < cus ? nunits : cus);
    hipLaunchKernelGGL((creff_roll_kernel<NB>), dim3(grid), dim3(NT), SMEM_BYTES, st, p);
    return arseg_launch_status();
}

}  // namespace

// creff_rr.hip's entry point dispatches here (same argument checks there).
int arseg_creff_roll_launch(const float *const *ref_nhwc_host, const int16_t *mv_q, int H, int W, const float *lr, const float *wq,
                            const float *bq, const float *wk, const float *bk, const float *wv, const float *bv, float *p_out,
                            int p_layout, const float *wf, const float *bf, int n_cls, float *logits, int log_softmax, int N, int Hp,
                            int Wp, int hp, int wp, int seg_rows, hipStream_t st) {
    const bool head = logits != nullptr;
    RollParams p;
    for (int i = 0; i < N; ++i) p.ref[i] = ref_nhwc_host[i];
    for (int i = N; i < MAXN; ++i) p.ref[i] = nullptr;
    p.mv = mv_q; p.lr = lr; p.wq = wq; p.bq = bq; p.wk = wk; p.bk = bk; p.wv = wv; p.bv = bv; p.wf = wf; p.bf = bf;
    p.p_out = p_out; p.logits = logits;
    p.N = N; p.Hp = Hp; p.Wp = Wp; p.hp = hp; p.wp = wp; p.H = H; p.W = W; p.n_cls = head ? n_cls : 0; p.log_softmax = log_softmax;
    p.p_layout = p_layout;
    p.nstrips = arseg_cdiv(Wp, SW);
    if (seg_rows <= 0) seg_rows = Hp >= 256 ? 128 : Hp;      // segments of a strip: enough units to fill 256 CUs evenly, few enough that the
    seg_rows = (seg_rows + 1) & ~1;                           // 7 fill iterations of a segment stay small beside its rows / 2 steps
    p.seg_rows = seg_rows; p.nseg = arseg_cdiv(Hp, seg_rows);
    p.p_bytes = (unsigned)((size_t)N * CH * Hp * Wp * sizeof(float)); p.l_bytes = head ? (unsigned)((size_t)N * n_cls * Hp * Wp * sizeof(float)) : 0u;
    p.lr_bytes = (unsigned)((size_t)N * CH * hp * wp * sizeof(float));
    p.ref_bytes = (unsigned)((size_t)CH * Hp * Wp * sizeof(float));
    p.sy = arseg_resize_scale(hp, Hp, true); p.sx = arseg_resize_scale(wp, Wp, true);
    if (!head) return launch<0>(p, st);
    return n_cls <= 16 ? launch<1>(p, st) : launch<2>(p, st);
}
