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
// cache policy of the p / logits stores: 2 = nontemporal (1.7 GB per 11-frame launch that nobody re-reads soon; keeps the XCD's L2 for the
// overlapping gather of the keyframe feature: 242.4-243.9 -> 241.4 us per frame)
#ifndef RR_PNT
#define RR_PNT 2
#endif
#include "warp_math.h"

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TX = 16, TY = 16, NT = 1024, CH = 64;
constexpr int LW = 18, LN = 324, LPL = 329;          // lr_up tile (+1 halo) per channel group
constexpr int R4W = 24, R4N = 576;                   // warped keyframe region: tile + 3 (window) + 1 (conv) halo
constexpr int R3W = 22;                              // key / value records: tile + 3 halo (22 x 22 = 484 per channel group)
constexpr int HPL = 577;                             // plane stride (float4) of the staged region: 577*4 % 32 == 4 -> the 8 lanes of a
                                                     // ds_write_b128 group (8 channel groups of one pixel) cover all 32 banks
constexpr int KPLK = 496, KPLV = 500;                // record plane strides: keys (ds_read_b128) a multiple of 16 records, values
                                                     // (transpose read) 16 banks apart
constexpr int BIG_BYTES = 16 * HPL * 16;             // 147,712: staged region / lr_up tile / key records / value records
constexpr int TAPW_OFF = BIG_BYTES;                  // [576] {ex, wx, ey, wy} with the tap validity folded in (0 = tap outside)
constexpr int TAPO_OFF = TAPW_OFF + R4N * 16;        // [576] pixel index of the NW tap | dx << 30 | dy << 31 (clamped taps)
constexpr int WFS_OFF = 16 * KPLV * 16;              // classifier records: behind the value records, inside BIG (128,000 + 8 KB <= 147,712)
constexpr int WDQ_OFF = TAPO_OFF + R4N * 4;          // [4 chunks][9 taps + bias][4 groups] query conv weights
constexpr int LRT_OFF = WDQ_OFF + 4 * 10 * 4 * 16;      // [18 rows | 18 columns] of the lr_up tile: {tap offset 0, tap offset 1, weight 0, weight 1}
constexpr int SMEM_BYTES = LRT_OFF + 2 * LW * 16;       // 162,368 <= 163,840
constexpr int MAXN = 32;
constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr float LOG2E = 1.44269504088896340736f;
#ifndef RR_G1
#define RR_G1 2       // gather units requested ahead of the query conv
#endif
#ifndef RR_PEARLY
#define RR_PEARLY 0   // 1: the single-pixel round is requested ahead of the query conv as well
#endif
#ifndef RR_GB
#define RR_GB 5       // gather units per later batch
#endif

struct RRParams {
    const float *ref[MAXN];       // un-warped keyframe feature of each frame, NHWC [Hp][Wp][64]
    const int16_t *mv;            // [N][H][W][2] quarter-pel
    const float *lr, *wq, *bq, *wk, *bk, *wv, *bv, *wf, *bf;
    float *p_out, *logits;
    int N, Hp, Wp, hp, wp, H, W, n_cls, log_softmax, p_layout, tiles_x, tiles_y;
    unsigned p_bytes, l_bytes, lr_bytes;
    float sy, sx;
    unsigned long long *dbg;
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
// LDS[lds_base + lane * 16] <- 16 bytes at g (LDS-DMA: no staging registers).  Inline asm: hipcc serialises the builtin (waterfall loop
// over M0 with a vmcnt(0) per load); the loads are invisible to its s_waitcnt bookkeeping, so the publishing barrier is preceded
// by an explicit s_waitcnt vmcnt(0).
__device__ __forceinline__ void dma16_glb(const void *g, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_base), "v"(g) : "memory");
}
__device__ __forceinline__ void dma4_glb(const void *g, unsigned lds_base) {        // LDS[lds_base + lane * 4] <- 4 bytes at g
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_base), "v"(g) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p; }
template <int CTRL>
__device__ __forceinline__ f32x4 dpp4(const f32x4 v) {     // row_shr:1 = 0x111 (lane i <- lane i-1), row_shl:1 = 0x101 (lane i <- lane i+1)
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[j]), CTRL, 0xF, 0xF, true));
    return r;
}
// a * b + c on packed pairs: v_pk_fma_f32 issues 2 FMAs in 4.2 cycles per wave, v_fmac_f32 one in 3.0 (measured on MI355X, 4 waves
// per SIMD) -- the depthwise convolutions are bound by exactly this
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 fma4(const f32x4 a, const f32x4 b, const f32x4 c) {
    const f32x2 lo = __builtin_elementwise_fma(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1), __builtin_shufflevector(c, c, 0, 1));
    const f32x2 hi = __builtin_elementwise_fma(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3), __builtin_shufflevector(c, c, 2, 3));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
// Record index (row-major in the 22 x 22 record region) of flat window key f = 16b + k0 (k0 in 0..15) of the patch at (pc, pr).
// The window is 8 rows x 14 columns: f = 14 ky + kx.  With e = 2b + k0 (< 28): ky = b + (e >= 14), kx = e - 14 (e >= 14), so
// rec = (2pr + ky) * 22 + 8pc + kx = [44 pr + 8 pc + k0] + 24 b + 8 (k0 >= 14 - 2b): one compare-select per block.
// reductions over the 4 DPP rows of a wave (lanes l, l^16, l^32, l^48) on the VALU: v_permlane16_swap exchanges the odd rows of its
// first operand with the even rows of the second, v_permlane32_swap the upper half of the first with the lower half of the second --
// fed two copies of x they return the pair (x, partner's x) in every lane.  (__shfl_xor is a ds_bpermute: an LDS round trip that all
// 16 lock-stepped waves of the workgroup wait for.)
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
// a wave-uniform double pinned to scalar registers (uniform fp64 values are computed on the VALU; left in VGPRs across the tile loop
// they are spilled to scratch and reloaded -- a memory round trip -- in the phase that uses them)
__device__ __forceinline__ double uniform_f64(double x) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    unsigned lo, hi;                                 // (asm: the builtin is sunk to the use and the VGPR pair stays live)
    asm volatile("s_nop 1\n\tv_readfirstlane_b32 %0, %2\n\tv_readfirstlane_b32 %1, %3" : "=s"(lo), "=s"(hi) : "v"((unsigned)u), "v"((unsigned)(u >> 32)));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int key_rec(int b, int k0, int base0) { return base0 + 24 * b + (k0 >= 14 - 2 * b ? 8 : 0); }

#ifdef RR_TIMING
// dev builds only: a wave adds the shader-clock ticks since its previous stamp to its dbg row (scalar registers, one atomic)
// (every wave: dbg[16 * wave + i]; tools/time_phases.py reports wave 0, the mean and the slowest wave of each phase)
#define RR_STAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        if ((tid0 & 63) == 0 && p.dbg) atomicAdd(p.dbg + 16 * (tid0 >> 6) + (i), now_ - tprev_); tprev_ = now_; } while (0)
#else
#define RR_STAMP(i) do { } while (0)
#endif
#ifdef RR_ABLATE
#define RR_ON(bit) (!((RR_ABLATE >> (bit)) & 1) || p.N > 1000000)
#else
#define RR_ON(bit) true
#endif

template <int NB>      // NB: classifier row blocks of 16 classes (0: no head)
__global__ __launch_bounds__(NT) void creff_rr_kernel(const RRParams p) {
    constexpr int NBA = NB > 0 ? NB : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4 *BIGf = reinterpret_cast<f32x4 *>(smem);
    u32x4 *BIGu = reinterpret_cast<u32x4 *>(smem);
    f32x4 *TapW = reinterpret_cast<f32x4 *>(smem + TAPW_OFF);
    unsigned *TapO = reinterpret_cast<unsigned *>(smem + TAPO_OFF);
    f32x4 *Wfs = reinterpret_cast<f32x4 *>(smem + WFS_OFF);        // [4 chunks][4 groups][NBA*16] {4 hi | 4 lo}
    f32x4 *WdQ = reinterpret_cast<f32x4 *>(smem + WDQ_OFF);
    u32x4 *LrT = reinterpret_cast<u32x4 *>(smem + LRT_OFF);
    float *Bfs = reinterpret_cast<float *>(smem + WFS_OFF + 4 * 4 * NBA * 16 * 16);      // [NBA*16] classifier bias (0 beyond n_cls)

    const int tid0 = threadIdx.x;
    // Persistent workgroups (one per CU: 158 KB of LDS): each walks its share of the tiles, so the 16-wave launch latency is paid
    // once.  XCD-aware order: workgroup ids go round-robin to the 8 XCDs (private L2s); XCD x owns a contiguous run of tiles and
    // its workgroups take neighbouring tiles at the same time, so the halos they share are fetched into one L2 once.
    const int per_img = p.tiles_x * p.tiles_y, ntiles = per_img * p.N;
    const int nx = min(8, (int)gridDim.x);                  // (fewer than 8 workgroups: one tile run per workgroup)
    const int xcd = blockIdx.x % nx, slot = blockIdx.x / nx, nslot = ((int)gridDim.x - xcd + nx - 1) / nx;
    const int t_lo = (int)((long long)ntiles * xcd / nx), t_hi = (int)((long long)ntiles * (xcd + 1) / nx);
    if (tid0 >= 640 && tid0 < 640 + 160) {           // query conv weights: the same for every tile
        const int e = tid0 - 640, gg = e & 3, tp = (e >> 2) % 10, c = e / 40;
        const float *src = tp < 9 ? p.wq + (size_t)tp * CH + c * 16 + gg * 4 : p.bq + c * 16 + gg * 4;
        WdQ[e] = *reinterpret_cast<const f32x4 *>(src);
    }
    // which of a lane's 28 score slots are real window taps: slot (b, i) is key f = 16b + 4g + i = (ky, kx) of the 8 x 14 patch
    // window; real iff ky - qy and kx - qx both lie in [0, 6].  A per-lane constant: one bit per slot.
    unsigned okmask = 0;
    {
        const int lane = tid0 & 63, q = lane & 15, g = lane >> 4, qy = q >> 3, qx = q & 7;
        for (int s = 0; s < 28; ++s) {
            const int f = 16 * (s >> 2) + 4 * g + (s & 3), ky = f / 14, kx = f - 14 * ky;
            if ((unsigned)(ky - qy) <= 6u && (unsigned)(kx - qx) <= 6u) okmask |= 1u << s;
        }
    }
    // The motion vector of a lane's region pixel (lanes < 576; identity-resize case) is requested one tile ahead, by LDS-DMA into the
    // lane's own slot of the tap-offset table (dead between the gather and the next tile's tap arithmetic): a register would be
    // live across the whole tile and hipcc spills it (scratch traffic + a vmcnt(0) at the request).
    const bool mv_ident = p.Hp == p.H && p.Wp == p.W;
    const double g_dW = uniform_f64((double)max(p.Wp - 1, 1)), g_dH = uniform_f64((double)max(p.Hp - 1, 1)), g_rW = uniform_f64(1.0 / g_dW), g_rH = uniform_f64(1.0 / g_dH);      // grid normalisation: extents and their reciprocals
#define RR_TID(t) int t = tid0; asm volatile("" : "+v"(t))
    auto mv_fetch = [&](int tile_) {
        const int n_ = tile_ / per_img, tr_ = tile_ - n_ * per_img;
        RR_TID(tq);
        const int yr = tq / R4W, xr = tq - yr * R4W;
        const int gy = (tr_ / p.tiles_x) * TY - 4 + yr, gx = (tr_ - (tr_ / p.tiles_x) * p.tiles_x) * TX - 4 + xr;
        if (mv_ident && tq < R4N && (unsigned)gy < (unsigned)p.Hp && (unsigned)gx < (unsigned)p.Wp)
            dma4_glb(p.mv + ((size_t)n_ * p.H * p.W + (size_t)gy * p.W + gx) * 2, lds_addr(TapO) + (unsigned)__builtin_amdgcn_readfirstlane(tq >> 6) * 256u);
    };
    if (t_lo + slot < t_hi) mv_fetch(t_lo + slot);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int tile = t_lo + slot; tile < t_hi; tile += nslot) {
    // Everything derived from the thread id is recomputed per phase from an opaque copy: left alone, LLVM hoists the per-lane
    // constants of all phases (record indices, masks, addresses) to the top of the tile loop where they occupy ~100 registers.
#ifdef RR_TIMING
    unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
#endif
    const int n = tile / per_img, trem = tile - n * per_img;
    const int ty0 = (trem / p.tiles_x) * TY, tx0 = (trem - (trem / p.tiles_x) * p.tiles_x) * TX;
    const int Hp = p.Hp, Wp = p.Wp;

    // ------------------------------------------------------------------ phase 0a: sampling taps; raw lr window under the tile into LDS
    // The window of the low-resolution feature under the tile (+1 halo, +1 for the second bilinear tap; block uniform) is staged as
    // whole 256-byte pixels (~110 coalesced pixel loads instead of 324 x 4 scattered 16-byte taps per channel group); the bilinear taps
    // of the lr_up tile are then read from LDS.  The tap arithmetic of the warp (fp64, one lane per region pixel) runs while the
    // window loads are in flight; the motion vector it starts from was requested during the previous tile.
    int wy_lo, wx_lo, wy_n, wx_n;
    {
        int a0, a1, b0, b1; float l;
        arseg_src_index(p.sy, max(ty0 - 1, 0), true, p.hp, a0, a1, l);
        arseg_src_index(p.sy, min(ty0 + TY, Hp - 1), true, p.hp, b0, b1, l);
        wy_lo = a0; wy_n = b1 - a0 + 1;
        arseg_src_index(p.sx, max(tx0 - 1, 0), true, p.wp, a0, a1, l);
        arseg_src_index(p.sx, min(tx0 + TX, Wp - 1), true, p.wp, b0, b1, l);
        wx_lo = a0; wx_n = b1 - a0 + 1;
    }
    const int wnpx = wy_n * wx_n;
    f32x4 *LwA = BIGf + 16 * LPL;                                             // [window pixel][16 channel groups], behind the lr_up tile
    const bool win_lds = wnpx <= 256 && 16 * LPL * 16 + wnpx * 256 <= BIG_BYTES;      // (always at the 0.5x scale: <= 11 x 11 pixels)
    {
        RR_TID(t);
        const float *lrn = p.lr + (size_t)n * p.hp * p.wp * CH;
        const int g16 = t & 15, pl = t >> 4;
        const unsigned mv_cur = t < R4N ? TapO[t] : 0u;
        // window pixels by LDS-DMA: a wave's 64 lanes = 4 pixels x 16 channel groups = 1 KB contiguous in the pixel-major image
        if (win_lds && RR_ON(2)) {
            const unsigned wbase = lds_addr(LwA) + (unsigned)__builtin_amdgcn_readfirstlane(t >> 6) * 1024u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int px = pl + 64 * k;
                if (px < wnpx) {
                    const int r = px / wx_n, c = px - r * wx_n;
                    dma16_glb(lrn + ((size_t)(wy_lo + r) * p.wp + wx_lo + c) * CH + g16 * 4, wbase + k * 16384u);
                }
            }
        }
        if (!RR_ON(0)) {
            if (t < R4N) { TapW[t] = f32x4{0.5f, 0.5f, 0.5f, 0.5f}; TapO[t] = (unsigned)t; }
        } else if (t < R4N) {
            const int yr = t / R4W, xr = t - yr * R4W;
            const int gy = ty0 - 4 + yr, gx = tx0 - 4 + xr;
            f32x4 w = {0.f, 0.f, 0.f, 0.f};
            unsigned o = 0;
            if ((unsigned)gy < (unsigned)Hp && (unsigned)gx < (unsigned)Wp) {      // outside the image the region is zero (conv padding)
                double fx, fy;
                if (mv_ident) {                        // identity resize (PSPNet): (q/4 * Hp) / H == q/4 exactly
                    fx = (double)(short)(mv_cur & 0xFFFFu) / 4.0; fy = (double)(short)(mv_cur >> 16) / 4.0;
                } else {
                    int H_ = p.H, W_ = p.W, Hp_ = Hp, Wp_ = Wp;   // opaque: keeps the fp64 scale factors of this (rare) path from being hoisted
                    asm volatile("" : "+s"(H_), "+s"(W_), "+s"(Hp_), "+s"(Wp_));      // out of the tile loop, where they would sit in spilled registers
                    mv_at(p.mv + (size_t)n * p.H * p.W * 2, H_, W_, Hp_, Wp_, gy, gx, fx, fy);
                }
                float ngx, ngy;
                norm_grid_rcp(gx, gy, fx, fy, g_dW, g_dH, g_rW, g_rH, ngx, ngy);
                int Hq = Hp, Wq = Wp;                  // opaque: (float)W etc. are converted here, not hoisted into (spilled) loop-invariant VGPRs
                asm volatile("" : "+s"(Hq), "+s"(Wq));
                const Taps tp = make_taps(ngx, ngy, Hq, Wq);
                const int xa = min(max(tp.x0, 0), Wp - 1), xc = min(max(tp.x0 + 1, 0), Wp - 1);
                const int ya = min(max(tp.y0, 0), Hp - 1), yc = min(max(tp.y0 + 1, 0), Hp - 1);
                o = (unsigned)(ya * Wp + xa) | ((unsigned)(xc - xa) << 30) | ((unsigned)(yc - ya) << 31);
                w = f32x4{tp.vx0 ? tp.ex : 0.f, tp.vx1 ? tp.wx : 0.f, tp.vy0 ? tp.ey : 0.f, tp.vy1 ? tp.wy : 0.f};
            }
            TapW[t] = w; TapO[t] = o;
        } else if (t < R4N + 2 * LW) {
            // bilinear (align_corners=True) taps of the lr_up tile, once per tile row / column: offsets relative to the staged window (or
            // pixel offsets into the lr image without it); the zero padding outside the image is folded into the weights
            const int i = t - R4N, isx = i >= LW, j = i - (isx ? LW : 0);
            const int gq = (isx ? tx0 : ty0) - 1 + j, lim = isx ? Wp : Hp;
            int i0, i1; float l;
            arseg_src_index(isx ? p.sx : p.sy, min(max(gq, 0), lim - 1), true, isx ? p.wp : p.hp, i0, i1, l);
            l = fminf(fmaxf(l, 0.f), 1.f);
            const float in = (unsigned)gq < (unsigned)lim ? 1.f : 0.f;
            const int mul = 16 * (isx ? 1 : (win_lds ? wx_n : p.wp)), lo = win_lds ? (isx ? wx_lo : wy_lo) : 0;     // in 16-byte units
            LrT[i] = u32x4{(unsigned)((i0 - lo) * mul), (unsigned)((i1 - lo) * mul), __float_as_uint((1.f - l) * in), __float_as_uint(l * in)};
        }
        RR_STAMP(12);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the window has landed (and the motion vector requested ahead of it)
    }
    __syncthreads();
    RR_STAMP(9);

    // ------------------------------------------------------------------ phases 1 + 2: query conv | gather + bilinear warp of the region into LDS
    // Gather unit = (region pixel, channel group): 16 lanes read one whole 256-byte pixel per tap.  The staged region overwrites the
    // lr_up tile, so it can only be WRITTEN after the query conv -- but the taps of the first G1 units are REQUESTED before it and
    // travel while the conv runs (their 16 G1 registers are free here: neither the columns nor the softmax weights are live yet).
    // Gather unit = a 2 x 2 block of region pixels x 16 channel groups (16 lanes).  When the four pixels sample one rigid 3 x 3 source
    // neighbourhood (equal motion vectors -- codec MVs are block constant -- and no clamping at the image border) the block costs 9
    // pixel loads instead of 16: the gather is bound by the bytes the texture path returns (64 B / clk / CU).  Other blocks take
    // the per-pixel path at commit time.
    constexpr int G1 = RR_G1, GU = 2, BW = R4W / 2;       // 144 blocks: two rounds of 64; the pixels of the last 16 blocks one by one (64 units)
    const float *g_img = p.ref[n];
    const unsigned g_row_off = (unsigned)Wp * CH;
    auto g_prep = [&](int k, const float *&a) -> bool {
        RR_TID(t);
        const int g16 = t & 15, u = (t >> 4) + 64 * k;
        const int by = u / BW, p00 = 2 * by * R4W + 2 * (u - by * BW);
        const unsigned o00 = TapO[p00], o01 = TapO[p00 + 1], o10 = TapO[p00 + R4W], o11 = TapO[p00 + R4W + 1];
        const bool rigid = (o00 >> 30) == 3u && o01 == o00 + 1u && o10 == o00 + (unsigned)Wp && o11 == o10 + 1u;
        // (the loads are unconditional: other blocks read the 3 x 3 pixels at the image origin and drop them -- a branch around the
        // loads makes hipcc spill their destination registers)
        a = g_img + (size_t)(rigid ? o00 & 0x3FFFFFFFu : 0u) * CH + g16 * 4;
        return rigid;
    };
    auto g_load = [&](const float *a, int j) { return *reinterpret_cast<const f32x4 *>(a + (j / 3) * g_row_off + (j % 3) * CH); };
    auto g_issue = [&](int k, f32x4 (&v)[9]) -> bool {
        const float *a;
        const bool rigid = g_prep(k, a);
#pragma unroll
        for (int j = 0; j < 9; ++j) v[j] = g_load(a, j);
        return rigid;
    };
    auto g_blend = [&](const f32x4 a, const f32x4 b, const f32x4 c, const f32x4 d, const f32x4 w) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc += a * (w[0] * w[2]);      // same order as warp_mvq_nhwc_kernel
        acc += b * (w[1] * w[2]);
        acc += c * (w[0] * w[3]);
        acc += d * (w[1] * w[3]);
        return acc;
    };
    auto g_commit = [&](int k, const f32x4 (&v)[9], bool rigid) {
        RR_TID(t);
        const int g16 = t & 15, u = (t >> 4) + 64 * k;
        const int by = u / BW, p00 = 2 * by * R4W + 2 * (u - by * BW);
        f32x4 *dst = BIGf + g16 * HPL + p00;
        if (rigid) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    dst[i * R4W + j] = g_blend(v[3 * i + j], v[3 * i + j + 1], v[3 * i + 3 + j], v[3 * i + 4 + j], TapW[p00 + i * R4W + j]);
        } else {
#pragma unroll 1
            for (int q = 0; q < 4; ++q) {                    // rare: one pixel at a time keeps the register footprint of the common path
                const int pq = p00 + (q >> 1) * R4W + (q & 1);
                const unsigned o = TapO[pq];
                const float *a = g_img + (size_t)(o & 0x3FFFFFFFu) * CH + g16 * 4;
                const unsigned dxo = (o & 0x40000000u) ? CH : 0u, dyo = (o & 0x80000000u) ? g_row_off : 0u;
                const f32x4 x0 = *reinterpret_cast<const f32x4 *>(a), x1 = *reinterpret_cast<const f32x4 *>(a + dxo);
                const f32x4 x2 = *reinterpret_cast<const f32x4 *>(a + dyo), x3 = *reinterpret_cast<const f32x4 *>(a + dyo + dxo);
                BIGf[g16 * HPL + pq] = g_blend(x0, x1, x2, x3, TapW[pq]);
            }
        }
    };
    // the 64 pixels of blocks 128..143, one per 16-lane group (general path: clamped taps, always valid -- every lane has work in every
    // round, and nothing is defined under a branch: hipcc spills registers that are)
    auto p_pix = [&]() { RR_TID(t); const int u = 2 * 64 + (t >> 6), q = (t >> 4) & 3, by = u / BW; return 2 * by * R4W + 2 * (u - by * BW) + (q >> 1) * R4W + (q & 1); };
    auto p_issue = [&](f32x4 (&x)[4]) {
        RR_TID(t);
        const unsigned o = TapO[p_pix()];
        const float *a = g_img + (size_t)(o & 0x3FFFFFFFu) * CH + (t & 15) * 4;
        const unsigned dxo = (o & 0x40000000u) ? CH : 0u, dyo = (o & 0x80000000u) ? g_row_off : 0u;
        x[0] = *reinterpret_cast<const f32x4 *>(a); x[1] = *reinterpret_cast<const f32x4 *>(a + dxo);
        x[2] = *reinterpret_cast<const f32x4 *>(a + dyo); x[3] = *reinterpret_cast<const f32x4 *>(a + dyo + dxo);
    };
    auto p_commit = [&](const f32x4 (&x)[4]) {
        RR_TID(t);
        const int pq = p_pix();
        BIGf[(t & 15) * HPL + pq] = g_blend(x[0], x[1], x[2], x[3], TapW[pq]);
    };
    // The requests of the first G1 rounds are spread over the lr_up units below: issued in one burst they fill the texture path's
    // queue and the waves sit in the issue stage until it drains, with nothing overlapped.
    f32x4 gv[G1 > 0 ? G1 : 1][9];
    bool grigid[G1 > 0 ? G1 : 1];
    const float *ga[G1 > 0 ? G1 : 1];
#pragma unroll
    for (int k = 0; k < G1; ++k) grigid[k] = g_prep(k, ga[k]);

    // ------------------------------------------------------------------ phase 0b: lr_up tile (+1 halo, all 64 channels) into LDS
    {
        RR_TID(t);
        const int g16 = t & 15, pl = t >> 4;
        const f32x4 *lrg = reinterpret_cast<const f32x4 *>(p.lr + (size_t)n * p.hp * p.wp * CH) + g16;
        const f32x4 *wb = LwA + g16;
        constexpr int NL = G1 > 0 ? 9 : 0;       // round 0 here (a wave can post ~9 requests before the texture queue stalls it); round 1 ahead of the query conv
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (RR_ON(1)) {
#pragma unroll
                for (int j = k * NL / 6; j < (k + 1) * NL / 6; ++j) gv[0][j] = g_load(ga[0], j);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int px = pl + 64 * k;
            if (win_lds && RR_ON(2) && (k < 5 || px < LN)) {
                const int r = px / LW, c = px - r * LW;
                const u32x4 rt = LrT[r], ct = LrT[LW + c];
                const float wy0 = __uint_as_float(rt.z), wy1 = __uint_as_float(rt.w), wx0 = __uint_as_float(ct.z), wx1 = __uint_as_float(ct.w);
                const f32x4 v0 = wb[rt.x + ct.x], v1 = wb[rt.x + ct.y], v2 = wb[rt.y + ct.x], v3 = wb[rt.y + ct.y];
                BIGf[g16 * LPL + px] = wy0 * (wx0 * v0 + wx1 * v1) + wy1 * (wx0 * v2 + wx1 * v3);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // without the staged window (other scales): taps from global memory.  A separate loop: with the LDS and the global taps merged
        // into one, hipcc waits for vmcnt(0) -- i.e. for the gather requests in flight -- before every interpolation.
        if (!win_lds && RR_ON(2)) {
            for (int px = pl; px < LN; px += 64) {
                const int r = px / LW, c = px - r * LW;
                const u32x4 rt = LrT[r], ct = LrT[LW + c];
                const float wy0 = __uint_as_float(rt.z), wy1 = __uint_as_float(rt.w), wx0 = __uint_as_float(ct.z), wx1 = __uint_as_float(ct.w);
                const f32x4 v0 = lrg[rt.x + ct.x], v1 = lrg[rt.x + ct.y], v2 = lrg[rt.y + ct.x], v3 = lrg[rt.y + ct.y];
                BIGf[g16 * LPL + px] = wy0 * (wx0 * v0 + wx1 * v1) + wy1 * (wx0 * v2 + wx1 * v3);
            }
        }
    }
    __syncthreads();
    RR_STAMP(1);

    // ------------------------------------------------------------------ phase 1: query conv, lane local (channels 16c + 4g .. +3 of the lane's own query)
    if (RR_ON(1)) {
#pragma unroll
        for (int k = 1; k < G1; ++k)
#pragma unroll
            for (int j = 0; j < 9; ++j) gv[k][j] = g_load(ga[k], j);
    }
#if RR_PEARLY
    f32x4 gx[4];
    if (RR_ON(1)) p_issue(gx);
#endif
    u32x2 qh[4], ql[4];
    {
        RR_TID(t);
        const int lane = t & 63, wave = t >> 6, q = lane & 15, g = lane >> 4;
        const int yq = 2 * (wave >> 1) + (q >> 3), xq = 8 * (wave & 1) + (q & 7);
#pragma unroll
        for (int c = 0; c < 4; ++c) { qh[c] = u32x2{(unsigned)t, 0u}; ql[c] = qh[c]; }
        if (RR_ON(3))
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
            // pin the result here: LLVM otherwise sinks the whole FMA chain to its use in the score phase and keeps the 76 loaded vectors alive
            asm volatile("" : "+v"(qh[c].x), "+v"(qh[c].y), "+v"(ql[c].x), "+v"(ql[c].y));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    RR_STAMP(2);

    if (RR_ON(1)) {
#pragma unroll
        for (int k = 0; k < G1; ++k) g_commit(k, gv[k], grigid[k]);
        __builtin_amdgcn_sched_barrier(0);
        {       // the remaining rounds in flight together
            f32x4 v[GU - G1 > 0 ? GU - G1 : 1][9], x[4];
            bool rg[GU - G1 > 0 ? GU - G1 : 1];
#pragma unroll
            for (int k = G1; k < GU; ++k) rg[k - G1] = g_issue(k, v[k - G1]);
#if RR_PEARLY
            p_commit(gx);
#else
            p_issue(x);
#endif
#pragma unroll
            for (int k = G1; k < GU; ++k) g_commit(k, v[k - G1], rg[k - G1]);
#if !RR_PEARLY
            p_commit(x);
#endif
        }
    }
    __syncthreads();
    RR_STAMP(3);

    // ------------------------------------------------------------------ phase 3: walker columns into registers
    // lane = DPP row r4 (span = r4 & 1, half = r4 >> 1) x 16 columns; channel group = wave
    f32x4 h[13];
    {
        RR_TID(t);
        const int lane = t & 63, wave = t >> 6, r4 = lane >> 4, xi = lane & 15;
        const int xr = 8 * (r4 & 1) + xi, rb = 11 * (r4 >> 1);
#pragma unroll
        for (int j = 0; j < 13; ++j) h[j] = BIGf[wave * HPL + (RR_ON(2) ? (rb + j) * R4W + xr : 0)];
        if (tile + nslot < t_hi) mv_fetch(tile + nslot);       // the tap tables are dead
    }
    __syncthreads();
    RR_STAMP(4);

    // ------------------------------------------------------------------ phases 4 / 6: key (value) records of the whole region
    // Interior tiles (every record of the 22 x 22 region inside the image: all but the frame's border tiles) need no padding mask -- a
    // compare, a scalar and and four selects per row: the row loop stores unmasked records and border tiles (a tile-uniform branch) zero
    // their out-of-image records afterwards.
    const bool rec_interior = ty0 >= 3 && tx0 >= 3 && ty0 + TY + 3 <= Hp && tx0 + TX + 3 <= Wp;
    auto conv_records = [&](const float *wt, const float *bs, int kpl) {
        RR_TID(t);
        const int lane = t & 63, r4 = lane >> 4, xi = lane & 15;
        const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
        const int xk = 8 * (r4 & 1) + xi - 1, rb = 11 * (r4 >> 1);       // record column, first record row of this lane
        // span 0 writes record columns 0..10, span 1 columns 11..21
        const bool wr_ok = (r4 & 1) ? (xi >= 4 && xi <= 14) : (xi >= 1 && xi <= 11);
        const bool col_in = (unsigned)(tx0 - 3 + xk) < (unsigned)Wp;
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
        for (int tp = 0; tp < 9; ++tp) w[tp] = __builtin_bit_cast(f32x4, ws[tp]);
        const f32x4 bias = __builtin_bit_cast(f32x4, bsv);
        f32x4 l0 = dpp4<0x111>(h[0]), r0 = dpp4<0x101>(h[0]);
        f32x4 l1 = dpp4<0x111>(h[1]), r1 = dpp4<0x101>(h[1]);
        u32x4 *dst = BIGu + wave * kpl + rb * R3W + xk;
        int gy = ty0 - 3 + rb;
#pragma unroll
        for (int j = 0; j < 11; ++j) {
            const f32x4 l2 = dpp4<0x111>(h[j + 2]), r2 = dpp4<0x101>(h[j + 2]);
            f32x4 acc = bias;
            acc = fma4(w[0], l0, acc); acc = fma4(w[1], h[j], acc); acc = fma4(w[2], r0, acc);
            acc = fma4(w[3], l1, acc); acc = fma4(w[4], h[j + 1], acc); acc = fma4(w[5], r1, acc);
            acc = fma4(w[6], l2, acc); acc = fma4(w[7], h[j + 2], acc); acc = fma4(w[8], r2, acc);
            u32x2 hi, lo;
            split4(acc, hi, lo);
            if (wr_ok) dst[j * R3W] = u32x4{hi.x, hi.y, lo.x, lo.y};
            l0 = l1; r0 = r1; l1 = l2; r1 = r2;
            __builtin_amdgcn_sched_barrier(0);      // row by row: hoisting the neighbour moves of all 13 rows would spill the columns
        }
        if (!rec_interior) {                        // the unfold's zero padding: records outside the image are zero
            unsigned zz = 0u;
            asm volatile("" : "+v"(zz));            // (defined here: hipcc otherwise keeps a zero vector in four registers across the whole tile loop)
#pragma unroll 1
            for (int j = 0; j < 11; ++j)
                if (wr_ok && !(col_in && (unsigned)(gy + j) < (unsigned)Hp)) dst[j * R3W] = u32x4{zz, zz, zz, zz};
        }
    };
    if (RR_ON(4)) conv_records(p.wk, p.bk, KPLK);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the next tile's motion vectors have landed long ago
    __syncthreads();
    RR_STAMP(5);

    // ------------------------------------------------------------------ phase 5: scores S[b][i] = q . key(16b + 4g + i), then softmax
    float inv;
    u32x4 P[7];
    {
        RR_TID(t);
        const int lane = t & 63, q = lane & 15, g = lane >> 4;
        const int wave = __builtin_amdgcn_readfirstlane(t >> 6), pc = wave & 1, pr = wave >> 1;
        f32x4 S[7];
#pragma unroll
        for (int b = 0; b < 7; ++b) S[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (RR_ON(5)) {
            int krec[7];
#pragma unroll
            for (int b = 0; b < 7; ++b) krec[b] = key_rec(b, q, 2 * pr * R3W + 8 * pc + q);
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
            // softmax over the 49 taps (padding taps included)
        RR_STAMP(10);
        float m = -INFINITY;
#pragma unroll
        for (int b = 0; b < 7; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                S[b][i] = (okmask >> (4 * b + i)) & 1u ? S[b][i] : -INFINITY;
                m = fmaxf(m, S[b][i]);
            }
        m = rows_max(m);
        const float ml = m * LOG2E;
        float z = 0.f;
#pragma unroll
        for (int b = 0; b < 7; ++b) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (RR_ON(11)) S[b][i] = __builtin_amdgcn_exp2f(fmaf(S[b][i], LOG2E, -ml));     // masked slots: exp2(-inf) = 0
                z += S[b][i];
            }
            u32x2 hi, lo;
            split4(S[b], hi, lo);
            P[b] = u32x4{hi.x, hi.y, lo.x, lo.y};
        }
        z = rows_sum(z);
        inv = 1.0f / z;                            // applied to the weighted sum instead of the 112 weights
#pragma unroll
        for (int b = 0; b < 7; ++b) asm volatile("" : "+v"(P[b]));      // finish the softmax here, not behind the value conv
    }
    __syncthreads();
    RR_STAMP(6);

    // ------------------------------------------------------------------ phase 6: value records (the key records are dead)
    if (RR_ON(6)) conv_records(p.wv, p.bv, KPLV);
    // classifier records [chunk][group][class] {4 hi | 4 lo}, behind the value records
    {
        RR_TID(t);
        if (NB > 0 && t < 4 * 4 * NBA * 16) {
            const int cls = t % (NBA * 16), gg = (t / (NBA * 16)) & 3, c = t / (4 * NBA * 16);
            f32x4 wv4 = {0.f, 0.f, 0.f, 0.f};
            if (cls < p.n_cls) wv4 = *reinterpret_cast<const f32x4 *>(p.wf + (size_t)cls * CH + c * 16 + gg * 4);
            u32x2 hi, lo;
            split4(wv4, hi, lo);
            Wfs[t] = __builtin_bit_cast(f32x4, u32x4{hi.x, hi.y, lo.x, lo.y});
        }
        if (NB > 0 && t >= 512 && t < 512 + NBA * 16) Bfs[t - 512] = t - 512 < p.n_cls ? p.bf[t - 512] : 0.f;
    }
    __syncthreads();
    RR_STAMP(7);

    // ------------------------------------------------------------------ phase 7: P.V, residual, classifier, stores
    {
        RR_TID(t);
        const int lane = t & 63, q = lane & 15, g = lane >> 4;
        const int wave = __builtin_amdgcn_readfirstlane(t >> 6), pc = wave & 1, pr = wave >> 1;
        const int gyq = ty0 + 2 * pr + (q >> 3), gxq = tx0 + 8 * pc + (q & 7);       // this lane's query pixel
        const bool inq = gyq < Hp && gxq < Wp;
        // byte offsets in 32 bits, once per tile (the entry point bounds both tensors below 2 GiB): per chunk / class only a uniform term is added
        const unsigned pix = (unsigned)(gyq * Wp + gxq), plane = (unsigned)(Hp * Wp);
        const unsigned p_off0 = p.p_layout == ARSEG_C8 ? (((unsigned)n * 8u + (unsigned)(g >> 1)) * plane + pix) * 32u + (unsigned)(g & 1) * 16u
                                                       : ((unsigned)n * plane + pix) * (CH * 4u) + 16u * g;
        const unsigned p_step = p.p_layout == ARSEG_C8 ? 2u * plane * 32u : 64u;      // chunk c: + c * p_step
        const unsigned l_off0 = ((unsigned)n * (unsigned)p.n_cls * plane + pix) * 4u;
        f32x4 lg[NBA];
#pragma unroll
        for (int nb = 0; nb < NBA; ++nb) lg[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        const __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.p_out, 0, (int)p.p_bytes, 0x00020000);
        if (RR_ON(7)) {
            // residual term lr_up(own pixel): bilinear(align_corners=True) taps, coordinates clamped into the image
            int y0, y1, x0, x1; float ly, lx;
            arseg_src_index(p.sy, min(gyq, Hp - 1), true, p.hp, y0, y1, ly);
            arseg_src_index(p.sx, min(gxq, Wp - 1), true, p.wp, x0, x1, lx);
            ly = fminf(fmaxf(ly, 0.f), 1.f); lx = fminf(fmaxf(lx, 0.f), 1.f);
            // 32-bit byte offsets into the lr tensor (buffer loads: the 64-bit pointers of four taps cost 8 registers in the phase with the
            // highest register pressure of the kernel)
            const __amdgpu_buffer_rsrc_t lr_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.lr), 0, (int)p.lr_bytes, 0x00020000);
            const unsigned lrb = (unsigned)n * (unsigned)(p.hp * p.wp) * (CH * 4u) + 16u * g;
            const unsigned o00 = lrb + (unsigned)(y0 * p.wp + x0) * (CH * 4u), o01 = lrb + (unsigned)(y0 * p.wp + x1) * (CH * 4u);
            const unsigned o10 = lrb + (unsigned)(y1 * p.wp + x0) * (CH * 4u), o11 = lrb + (unsigned)(y1 * p.wp + x1) * (CH * 4u);
            auto lr_tap = [&](unsigned o, int c) { return RR_ON(8) ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lr_rsrc, o, 64 * c, 0)) : f32x4{(float)o, 0.f, 0.f, 0.f}; };
            // byte offset of the value record of (block b, this lane's key) in a channel-group plane: vbase + 384 b (+128 where bit b of
            // vsel is set) -- see key_rec; one register pair instead of seven addresses
            const int vk0 = 4 * g + (q >> 2);
            const unsigned vbase = (unsigned)(2 * pr * R3W + 8 * pc + vk0) * 16u;
            unsigned vsel = 0;
#pragma unroll
            for (int b = 0; b < 7; ++b) vsel |= (vk0 >= 14 - 2 * b ? 1u : 0u) << b;
            auto vrec = [&](int b) { return vbase + 384u * b + (((vsel >> b) & 1u) << 7); };
            auto epilogue = [&](int c, const f32x4 o) {       // o = p[query][16c + 4g .. +3]: store, then this chunk's share of the classifier
                const unsigned off = p_off0 + (unsigned)c * p_step;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), p_rsrc, (inq && RR_ON(9)) ? off : OOB, 0, RR_PNT);
                if (NB > 0 && RR_ON(13)) {
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
            };
            // the four residual taps of a chunk are requested at the top of the chunk and blended BEHIND its 14 MFMAs (requested one chunk
            // ahead and blended at the top, chunk 0 waited for its taps with nothing to cover the L2 round trip)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 a00 = lr_tap(o00, c), a01 = lr_tap(o01, c), a10 = lr_tap(o10, c), a11 = lr_tap(o11, c);
                const unsigned char *va = reinterpret_cast<const unsigned char *>(BIGu + (4 * c + (q & 3)) * KPLV);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int b = 0; b < 7; ++b) {
                    const u32x2 vh = lds_tr16(va + vrec(b)), vl = lds_tr16(va + vrec(b) + 8);
                    const h16x8 pb = __builtin_bit_cast(h16x8, P[b]);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(pack8(vh, vl), pb, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(pack8(vl, vh), pb, acc, 0, 0, 0);
                }
                const f32x4 lrc = (1.f - ly) * ((1.f - lx) * a00 + lx * a01) + ly * ((1.f - lx) * a10 + lx * a11);
                epilogue(c, lrc + acc * inv);
            }
        }

        // logits: lg[nb][i] = class 16nb + 4g + i of query q
        RR_STAMP(11);
        if (NB > 0) {
            const __amdgpu_buffer_rsrc_t l_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.logits, 0, (int)p.l_bytes, 0x00020000);
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
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(lg[nb][i]), l_rsrc, (inq && cls < p.n_cls && RR_ON(14)) ? off : OOB, 0, RR_PNT);
                }
        }
    }
    RR_STAMP(13);
    __syncthreads();          // the next tile's tables / staging overwrite LDS this tile still reads
    RR_STAMP(8);
#ifdef RR_TIMING
    if (tid0 == 0 && p.dbg) atomicAdd(p.dbg + 15, 1ull);
#endif
  }
}

template <int NB>
int launch(const RRParams &p, hipStream_t st) {
    static ArsegSmemAttr attr;
    if (int e = arseg_allow_smem(attr, reinterpret_cast<const void *>(creff_rr_kernel<NB>), SMEM_BYTES)) return e;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const long long ntiles = (long long)p.tiles_x * p.tiles_y * p.N;
    const int grid = (int)(ntiles < cus ? ntiles : cus);
    hipLaunchKernelGGL((creff_rr_kernel<NB>), dim3(grid), dim3(NT), SMEM_BYTES, st, p);
    return arseg_launch_status();
}

#ifdef RR_TIMING
unsigned long long *g_rr_dbg = nullptr;
#endif
}  // namespace

#ifdef RR_TIMING
extern "C" void arseg__rr_set_dbg(void *ptr) { g_rr_dbg = (unsigned long long *)ptr; }
#endif

extern "C" int arseg_creff_warp_fwd_ex(const float *const *ref_nhwc_host, const int16_t *mv_q, int H, int W, const float *lr,
                                       const float *wq, const float *bq, const float *wk, const float *bk, const float *wv,
                                       const float *bv, float *p_out, int p_layout, const float *wf, const float *bf, int n_cls,
                                       float *logits, int log_softmax, int N, int C, int Hp, int Wp, int hp, int wp, int kH, int kW,
                                       int impl, int seg_rows, int max_wgs, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(ref_nhwc_host); ARSEG_CHECK_PTR(mv_q); ARSEG_CHECK_PTR(lr); ARSEG_CHECK_PTR(wq); ARSEG_CHECK_PTR(bq); ARSEG_CHECK_PTR(wk);
    ARSEG_CHECK_PTR(bk); ARSEG_CHECK_PTR(wv); ARSEG_CHECK_PTR(bv); ARSEG_CHECK_PTR(p_out);
    ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(C); ARSEG_CHECK_POS(Hp); ARSEG_CHECK_POS(Wp); ARSEG_CHECK_POS(hp); ARSEG_CHECK_POS(wp);
    ARSEG_CHECK_POS(H); ARSEG_CHECK_POS(W);
    if (C != CH || kH != 7 || kW != 7 || N > MAXN) return ARSEG_EUNSUPPORTED;
    if (p_layout != ARSEG_C8 && p_layout != ARSEG_NHWC) return ARSEG_EINVAL;
    if ((size_t)C * Hp * Wp * sizeof(float) >= (1ull << 31) || (size_t)C * hp * wp * sizeof(float) >= (1ull << 31))
        return ARSEG_EUNSUPPORTED;                                                             // 32-bit buffer offsets within a frame
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
    if (impl != ARSEG_CREFF_WARP_AUTO && impl != ARSEG_CREFF_WARP_TILES && impl != ARSEG_CREFF_WARP_ROLL) return ARSEG_EINVAL;
    if (seg_rows < 0 || max_wgs < 0) return ARSEG_EINVAL;
    for (int i = 0; i < N; ++i)
        if (!ref_nhwc_host[i] || !ARSEG_ALIGNED16(ref_nhwc_host[i])) return ARSEG_EINVAL;
    // ONE dispatch rule (arseg_creff_warp_select states it, tests/test_gpu_ops.py::test_creff_dispatch_table enforces it): the rolling kernel
    // (creff_roll.hip) serves every launch it admits -- no head or a head of <= 16 classes, a schedule that fits its piece table -- the
    // 16 x 16 tile kernel below the rest (17-32 classes, oversized schedules) and impl = TILES
    if (impl != ARSEG_CREFF_WARP_TILES) {
        const int e = arseg_creff_roll_launch(ref_nhwc_host, mv_q, H, W, lr, wq, bq, wk, bk, wv, bv, p_out, p_layout, wf, bf, n_cls, logits,
                                              log_softmax, N, Hp, Wp, hp, wp, seg_rows, max_wgs, false, arseg_stream(stream));
        if (e != ARSEG_EUNSUPPORTED || impl == ARSEG_CREFF_WARP_ROLL) return e;
    }
    // the tile kernel addresses the whole batch through one descriptor per tensor (the rolling kernel: one per frame)
    if ((size_t)N * C * Hp * Wp * sizeof(float) >= (1ull << 31) || (size_t)N * C * hp * wp * sizeof(float) >= (1ull << 31)) return ARSEG_EUNSUPPORTED;
    RRParams p;
    for (int i = 0; i < N; ++i) p.ref[i] = ref_nhwc_host[i];
    for (int i = N; i < MAXN; ++i) p.ref[i] = nullptr;
    p.mv = mv_q; p.lr = lr; p.wq = wq; p.bq = bq; p.wk = wk; p.bk = bk; p.wv = wv; p.bv = bv; p.wf = wf; p.bf = bf;
    p.p_out = p_out; p.logits = logits;
    p.N = N; p.Hp = Hp; p.Wp = Wp; p.hp = hp; p.wp = wp; p.H = H; p.W = W; p.n_cls = head ? n_cls : 0; p.log_softmax = log_softmax;
    p.p_layout = p_layout; p.tiles_x = arseg_cdiv(Wp, TX); p.tiles_y = arseg_cdiv(Hp, TY);
    p.p_bytes = (unsigned)((size_t)N * C * Hp * Wp * sizeof(float)); p.l_bytes = head ? (unsigned)((size_t)N * n_cls * Hp * Wp * sizeof(float)) : 0u;
    p.lr_bytes = (unsigned)((size_t)N * C * hp * wp * sizeof(float));
    p.sy = arseg_resize_scale(hp, Hp, true); p.sx = arseg_resize_scale(wp, Wp, true);
    p.dbg = nullptr;
#ifdef RR_TIMING
    p.dbg = g_rr_dbg;
#endif
    hipStream_t st = arseg_stream(stream);
    if (!head) return launch<0>(p, st);
    return n_cls <= 16 ? launch<1>(p, st) : launch<2>(p, st);
}

extern "C" int arseg_creff_warp_select(int N, int C, int Hp, int Wp, int hp, int wp, int kH, int kW, int n_cls, int impl, int seg_rows, int max_wgs) {
    if (N <= 0 || C <= 0 || Hp <= 0 || Wp <= 0 || hp <= 0 || wp <= 0 || n_cls < 0 || seg_rows < 0 || max_wgs < 0) return ARSEG_EINVAL;
    if (impl != ARSEG_CREFF_WARP_AUTO && impl != ARSEG_CREFF_WARP_TILES && impl != ARSEG_CREFF_WARP_ROLL) return ARSEG_EINVAL;
    if (C != CH || kH != 7 || kW != 7 || N > MAXN || n_cls > 32) return ARSEG_EUNSUPPORTED;
    if ((size_t)C * Hp * Wp * sizeof(float) >= (1ull << 31) || (size_t)C * hp * wp * sizeof(float) >= (1ull << 31) || (size_t)Hp * Wp >= (1u << 30))
        return ARSEG_EUNSUPPORTED;
    // (the tile kernel: 32-bit offsets over the whole batch; the rolling kernel: over one frame)
    const bool tiles_fit = (size_t)N * C * Hp * Wp * sizeof(float) < (1ull << 31) && (size_t)N * C * hp * wp * sizeof(float) < (1ull << 31);
    if (impl == ARSEG_CREFF_WARP_TILES) return tiles_fit ? ARSEG_CREFF_WARP_TILES : ARSEG_EUNSUPPORTED;
    const int e = arseg_creff_roll_launch(nullptr, nullptr, 1, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ARSEG_NHWC, nullptr,
                                          nullptr, n_cls, nullptr, 0, N, Hp, Wp, hp, wp, seg_rows, max_wgs, true, nullptr);
    if (e == ARSEG_OK) return ARSEG_CREFF_WARP_ROLL;
    if (e != ARSEG_EUNSUPPORTED) return e;
    return (impl == ARSEG_CREFF_WARP_ROLL || !tiles_fit) ? ARSEG_EUNSUPPORTED : ARSEG_CREFF_WARP_TILES;
}

extern "C" int arseg_creff_warp_fwd(const float *const *ref_nhwc_host, const int16_t *mv_q, int H, int W, const float *lr,
                                    const float *wq, const float *bq, const float *wk, const float *bk, const float *wv,
                                    const float *bv, float *p_out, int p_layout, const float *wf, const float *bf, int n_cls,
                                    float *logits, int log_softmax, int N, int C, int Hp, int Wp, int hp, int wp, int kH, int kW,
                                    arseg_stream_t stream) {
    return arseg_creff_warp_fwd_ex(ref_nhwc_host, mv_q, H, W, lr, wq, bq, wk, bk, wv, bv, p_out, p_layout, wf, bf, n_cls, logits, log_softmax,
                                   N, C, Hp, Wp, hp, wp, kH, kW, ARSEG_CREFF_WARP_AUTO, 0, 0, stream);
}
