// CReFF for the 64-channel full-resolution PSPNet feature, ROLLING form -- one kernel per non-keyframe batch:
// warpFeature (evaluation.py:61-87, on the int16 MV map of evaluation.py:176-180)  ->  MyAttention.forward
// (model/attention.py:184-213)  ->  final 1x1 classifier + LogSoftmax (model/pspnet.py:225-229).
//
// Why a fourth kernel.  creff_rr.hip works on 16 x 16 tiles whose working set IS the compute unit (147 KB of LDS, key records OR value
// records, never both), so its nine phases run one after the other and the VALU, the texture path, the LDS and the matrix pipe are busy
// in turn (profiles/r03_creff_ablation.json).  Here a workgroup walks DOWN a 16-pixel-wide strip two rows at a time and keeps only what
// the 7 x 7 windows of the current row pair need: 8 rows of key records and 8 rows of value records in two LDS rings (2 x 45 KB).  The
// halo shrinks from 2.25x (24 x 24 region per 16 x 16 tile) to 1.5x (gather) / 1.375x (records), and -- the point -- the stages of
// DIFFERENT row pairs run at the same time on different waves (16 waves: 4 consumers, one per SIMD, and 12 producers):
//
//   producer waves 4..15                                           consumer waves 0..3 = (8-column query patch pc, key half kh)
//   H1(t): taps of gather t | blend -> warp stage                  H1(t): [kh 0: merge + residual + classifier + stores of step s-1]
//          lr_up rows of step t-4 -> lr_up stage                          Q.K^T + softmax of step s = t - 5 over the wave's key blocks
//          value conv k = t-2 (rows in registers) -> value ring
//   ---------------------------------------------------------------- barrier A
//   H2(t): key conv k = t-1 -> key ring | query conv s = t-4        H2(t): P.V over the wave's key blocks  [kh 1: partials -> LDS]
//          residual record of step t-5 | tap table of gather t+2           [kh 0: log-softmax + logits stores of step s-1]
//          + touch of its lines | MVs of gather t+3
//   ---------------------------------------------------------------- barrier B
//
// so the depthwise convolutions and the gather of rows further down run under the MFMAs of the rows being finished.  A producer lane
// owns one (column, channel group) of the strip for the whole segment and keeps a 2-row window of its 3-column neighbourhood in
// registers (the third row of a 3 x 3 stencil is the staged one): every warped / upsampled value is written to LDS once and read three times.
//
// The two consumer waves of a patch split its 7 key blocks (8 x 14 window keys, flattened into blocks of 16) 3 + 4: each runs Q.K^T,
// a softmax relative to ITS OWN maximum and P.V over its blocks; the halves are merged flash-style one half step later by the kh-0 wave
// (o = (o0 a0 + o1 a1) / (z0 a0 + z1 a1), a = exp(m_half - max)) -- no barrier beyond the two of the iteration.  Otherwise the consumer
// arithmetic (hi/lo split-fp16 operands on v_mfma_f32_16x16x32_f16, softmax over all 49 taps incl. padding taps) is that of creff_rr.hip.
//
// Arithmetic contract: as creff.hip (zero-padded unfold: keys / values outside the image are 0 and still take softmax mass).
#include "creff_params.h"
#ifndef ROLL_PNT
#define ROLL_PNT 2            // cache policy of the p / logits stores: 2 = nontemporal
#endif
#ifndef ROLL_LRTAB
#define ROLL_LRTAB 0          // 1: the row taps of the lr_up rows come from a table the tap wave writes (measured slower: an LDS round trip in
#endif                        // front of the lr loads, and the tap wave is the longest of H2)
#ifndef ROLL_SPLIT6
#define ROLL_SPLIT6 0          // 1: lo halves by v_fma_mixlo / mixhi_f16 (6 instead of 8 instructions per 4 values): measured 2 % SLOWER (5685 vs 5572 cycles per step)
#endif
#ifndef ROLL_LPRIO
#define ROLL_LPRIO 0           // 1: raised issue priority while a producer wave requests its taps (and, for key/value waves, runs the value conv):
#endif                         // zero-sum -- the producers' H1 drops 2600 -> 1900 cycles, the merge waves' rises 2380 -> 3280: a SIMD's H1 is issue-bound
#ifndef ROLL_CPRIO
#define ROLL_CPRIO 2          // issue priority of the consumer waves (producers: 0)
#endif
#ifndef ROLL_KSLOT5
#define ROLL_KSLOT5 0         // 1: key ring row r lives in ring slot (5 r) & 7 instead of r & 7 (see kslot below): removes the bank conflicts of the key-block
                              // reads across a window-row wrap -- measured (r6, same box, tools/ab_roll.sh): 0.1626-0.1627 vs 0.1617-0.1621 ms per frame,
                              // i.e. 0.4 % SLOWER: those conflicts are not on the critical path of either half step
#endif
#include "warp_math.h"

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int CH = 64, NT = 1024, NCONS = 4;         // 16 waves: 4 consumers + 768 producer lanes, one gather unit (pixel, channel group) each
constexpr int SW = 16;                               // query columns of a strip (two 8-column patches)
constexpr int RW = SW + 6;                           // key / value record columns (22)
constexpr int GW = SW + 8;                           // warped keyframe columns (24): + 1 for the depthwise convs
constexpr int LW = SW + 2;                           // lr_up columns (18)
constexpr int KSLOT = 8, VSLOT = 8;                  // ring rows = the 8 rows under the windows of a row pair: key rows are written in H2 and read
                                                     // in H1, value rows written in H1 and read in H2 -- nobody reads a ring while it is written
constexpr int KPL = KSLOT * RW;                      // key records per channel-group plane: 176, a multiple of 16 (ds_read_b128 of one key
                                                     // block: the 4 groups g land in one 256-byte bank row side by side -- see creff_rr.hip)
constexpr int VHALF = VSLOT * RW * 8;                // value records: per channel group a plane of the hi halves (8 bytes per record) and one of
constexpr int VPL = (2 * VHALF + 64) / 16;           // the lo halves, groups 2880 bytes = 64 mod 256 apart -- a transpose read takes 8 bytes of 8
                                                     // consecutive keys from each of 4 groups: 4 x 64 bytes side by side in the 256-byte bank row
constexpr int WPL = GW + 1, LPL = LW + 1;            // plane pitch of the two stages in f32x4 (25 * 16 B = 16 mod 128, 19 * 16 B = 48 mod 128:
                                                     // the 8 lanes of a ds_write_b128 group -- 8 channel groups of one pixel -- cover all 32 banks)
constexpr int NGP = 2 * GW, NLP = 2 * LW;            // staged pixels per iteration: 48 warped, 36 lr_up
constexpr int KV_LANES = 16 * RW;                   // key / value lanes: producer lanes 0 .. 351 (waves 4 .. 9)
constexpr int K_OFF = 0;
constexpr int V_OFF = K_OFF + 16 * KPL * 16;         //  45,056
constexpr int WS_OFF = V_OFF + 16 * VPL * 16;        //  90,112  warp stage [2 rows][16 groups][WPL]
constexpr int LS_OFF = WS_OFF + 2 * 16 * WPL * 16;   // 102,912  lr_up stage [2 rows][16 groups][LPL]
constexpr int Q_OFF = LS_OFF + 2 * 16 * LPL * 16;    // 112,640  query records [2 patches][4 chunks][4 groups][16 queries] {4 hi | 4 lo}
constexpr int R_OFF = Q_OFF + 2 * 16 * 16 * 16;      // 120,832  residual records lr_up(query) [2 patches][16 groups][16 queries] fp32 x 4
constexpr int XB_OFF = R_OFF + 2 * 16 * 16 * 16;     // 129,024  partials of the kh-1 waves [2 patches][4 chunks + {m, z}][64 lanes]
constexpr int TW_OFF = XB_OFF + 2 * 5 * 64 * 16;     // 139,264  [48] {ex, wx, ey, wy} with the tap validity folded in
constexpr int TO_OFF = TW_OFF + NGP * 16;            //          [48] byte offsets of the four (clamped) taps of a pixel: NW, NE, SW, SE
constexpr int LR_OFF = TO_OFF + NGP * 16;            //          [2] lr taps of the two lr_up rows of an iteration {row 0 bytes, row 1 bytes, w0, w1}
constexpr int WD_OFF = LR_OFF + 2 * 16;              //          depthwise weights [key | value | query][16 groups][9 taps + bias]
constexpr int WF_OFF = WD_OFF + 3 * 160 * 16;        // 147,904  classifier records [4 chunks][4 groups][32] {4 hi | 4 lo}
constexpr int BF_OFF = WF_OFF + 4 * 4 * 32 * 16;     // 156,096  classifier bias [32]
constexpr int SC_OFF = BF_OFF + 32 * 4;              //          this workgroup's list of pieces (struct PieceTab)
constexpr int SMEM_BYTES = SC_OFF + 16 + 64 * 16;    // 158,896 <= 163,840
constexpr int T_FIRST = -3;                         // first iteration of a segment (MV request of gather 0); the last is S + 5
constexpr int MAXN = 32;
constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr float LOG2E = 1.44269504088896340736f;

// Ring slot of key row r.  A consumer's ds_read_b128 of one key block takes 16 consecutive window keys e = 2b + q, which wrap from window
// row b to row b + 1 at e = 14: with the rows in consecutive slots (pitch RW = 22 records) the lanes behind the wrap sit 22 - 14 = 8 records
// = half a 256-byte bank row away from where a contiguous run would put them and collide with the lanes 8 records further on (2-way
// conflicts on up to half of the lanes: profiles/r04_v1_pmc_creff_roll.json, SQ_LDS_BANK_CONFLICT = 26 % of the LDS-active cycles).  With
// consecutive rows 5 slots apart (5 is coprime to 8: still a permutation of the ring) the wrap lands 5 * 22 - 14 = 96 = 0 mod 16 records
// (or -3 * 22 - 14 = -80) further: the 16 lanes cover the 64 banks exactly once.
__device__ __forceinline__ int kslot(int r) { return ROLL_KSLOT5 ? (5 * r) & 7 : r & 7; }

struct RollParams {
    const float *ref[MAXN];       // un-warped keyframe feature of each frame, NHWC [Hp][Wp][64]
    const int16_t *mv;            // [N][H][W][2] quarter-pel
    const float *lr, *wq, *bq, *wk, *bk, *wv, *bv, *wf, *bf;
    float *p_out, *logits;
    int N, Hp, Wp, hp, wp, H, W, n_cls, log_softmax, p_layout, nstrips, nseg, seg_rows, balanced;
    unsigned p_bytes, l_bytes, lr_bytes, ref_bytes;      // of ONE frame: every frame has its own buffer descriptor
    float sy, sx;
    unsigned long long *dbg;
};

// fp32 -> two fp16 (x = hi + lo, 22 bits together; see arseg_split_f16) in 6 instructions per 4 values: the lo halves are written by
// v_fma_mixlo / mixhi_f16 straight into their packed register (arseg_split_f16: fp32 differences + a packing convert, 8 instructions).
// The lo half is rounded to nearest instead of toward zero; beyond |x| = 131008 it becomes Inf instead of clamping -- far outside the
// features this kernel sees (the tile kernel measured this form 2 % slower; here the VALU issue count is what bounds a half step).
__device__ __forceinline__ void split4(const f32x4 v, u32x2 &hi, u32x2 &lo) {
#if ROLL_SPLIT6
    unsigned h01, h23, l01, l23;
    asm("v_cvt_pkrtz_f16_f32 %0, %4, %5\n\tv_cvt_pkrtz_f16_f32 %1, %6, %7\n\t"
        "v_fma_mixlo_f16 %2, %0, -1.0, %4 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %2, %0, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %3, %1, -1.0, %6 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %3, %1, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(h01), "=&v"(h23), "=&v"(l01), "=&v"(l23) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
#else
    unsigned h01, h23, l01, l23;
    arseg_split_f16(v, h01, h23, l01, l23);
#endif
    hi = u32x2{h01, h23}; lo = u32x2{l01, l23};
}
__device__ __forceinline__ u32x4 split4r(const f32x4 v) {      // the record form {4 hi | 4 lo}
    u32x2 hi, lo;
    split4(v, hi, lo);
    return u32x4{hi.x, hi.y, lo.x, lo.y};
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
// a wave-uniform double pinned to scalar registers (left to the compiler, uniform fp64 values live in VGPR pairs across the whole
// kernel and are spilled; asm: the builtin is folded back into the vector value)
__device__ __forceinline__ double uniform_f64(double x) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    unsigned lo, hi;
    asm volatile("s_nop 1\n\tv_readfirstlane_b32 %0, %2\n\tv_readfirstlane_b32 %1, %3" : "=s"(lo), "=s"(hi) : "v"((unsigned)u), "v"((unsigned)(u >> 32)));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// Workgroup barrier that orders LDS traffic only: global loads requested before it stay in flight across it (the gather of the next
// row pair travels under the convolutions of this one).
__device__ __forceinline__ void wg_sync() {
#ifdef ROLL_FULLSYNC
    __syncthreads();
    return;
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// 3 x 3 depthwise stencil on 4 channels, two output rows at once: rows r0..r3, each {left, centre, right}; out_a = rows r0..r2, out_b =
// rows r1..r3; weights w[0..8] + bias w[9] read from LDS one tap at a time (accumulation order of creff_rr.hip / creff.hip)
__device__ __forceinline__ void stencil2(const f32x4 *w, const f32x4 (&r0)[3], const f32x4 (&r1)[3], const f32x4 (&r2)[3],
                                         const f32x4 (&r3)[3], f32x4 &oa, f32x4 &ob) {
    f32x4 a = w[9], b = a;
#pragma unroll
    for (int j = 0; j < 3; ++j) { const f32x4 wj = w[j]; a = fma4(wj, r0[j], a); b = fma4(wj, r1[j], b); }
#pragma unroll
    for (int j = 0; j < 3; ++j) { const f32x4 wj = w[3 + j]; a = fma4(wj, r1[j], a); b = fma4(wj, r2[j], b); }
#pragma unroll
    for (int j = 0; j < 3; ++j) { const f32x4 wj = w[6 + j]; a = fma4(wj, r2[j], a); b = fma4(wj, r3[j], b); }
    oa = a; ob = b;
}

// the same with the ten weight vectors already in registers (requested ahead of the barrier in front of the stencil: constants, so
// their LDS round trip need not sit on the critical path of the half step that uses them)
__device__ __forceinline__ void stencil2r(const f32x4 (&w)[10], const f32x4 (&r0)[3], const f32x4 (&r1)[3], const f32x4 (&r2)[3],
                                          const f32x4 (&r3)[3], f32x4 &oa, f32x4 &ob) {
    f32x4 a = w[9], b = a;
#pragma unroll
    for (int j = 0; j < 3; ++j) { a = fma4(w[j], r0[j], a); b = fma4(w[j], r1[j], b); }
#pragma unroll
    for (int j = 0; j < 3; ++j) { a = fma4(w[3 + j], r1[j], a); b = fma4(w[3 + j], r2[j], b); }
#pragma unroll
    for (int j = 0; j < 3; ++j) { a = fma4(w[6 + j], r2[j], a); b = fma4(w[6 + j], r3[j], b); }
    oa = a; ob = b;
}

// half and half: taps 0..4 in registers (requested before the barrier), taps 5..8 + bias from LDS -- the first taps read only the window rows
// the lane holds, so the stencil starts without waiting for anything it requested in this half step
__device__ __forceinline__ void stencil2h(const f32x4 (&wa)[5], const f32x4 *w, const f32x4 (&r0)[3], const f32x4 (&r1)[3], const f32x4 (&r2)[3],
                                          const f32x4 (&r3)[3], f32x4 &oa, f32x4 &ob) {
    f32x4 a = w[9], b = a;
#pragma unroll
    for (int j = 0; j < 3; ++j) { a = fma4(wa[j], r0[j], a); b = fma4(wa[j], r1[j], b); }
#pragma unroll
    for (int j = 0; j < 2; ++j) { a = fma4(wa[3 + j], r1[j], a); b = fma4(wa[3 + j], r2[j], b); }
    { const f32x4 wj = w[5]; a = fma4(wj, r1[2], a); b = fma4(wj, r2[2], b); }
#pragma unroll
    for (int j = 0; j < 3; ++j) { const f32x4 wj = w[6 + j]; a = fma4(wj, r2[j], a); b = fma4(wj, r3[j], b); }
    oa = a; ob = b;
}

#ifdef ROLL_TIMING
// dev builds only: every wave accumulates the shader-clock ticks of its four segments per iteration (H1 work, wait at barrier A, H2 work,
// wait at barrier B) and adds them to dbg[8 * wave + i] when it is done (tools/time_roll.py)
#define RT_DECL unsigned long long tacc_[4] = {0ull, 0ull, 0ull, 0ull}, tprev_ = __builtin_amdgcn_s_memtime(), nsteps_ = 0
#define RT(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc_[i] += now_ - tprev_; tprev_ = now_; } while (0)
#define RT_STEPS(n) nsteps_ += (n)
#define RT_FLUSH() do { if ((tid & 63) == 0 && p.dbg) { for (int i_ = 0; i_ < 4; ++i_) atomicAdd(p.dbg + 8 * wave + i_, tacc_[i_]); \
        atomicAdd(p.dbg + 8 * wave + 4, nsteps_); } } while (0)
#elif defined(ROLL_MARK)
// census builds only: comment markers in the listing at the segment boundaries (tools/roll_census.py --marks)
#define RT_DECL do { } while (0)
#define RT(i) asm volatile("; ROLLMARK " #i)
#define RT_STEPS(n) do { } while (0)
#define RT_FLUSH() do { } while (0)
#else
#define RT_DECL do { } while (0)
#define RT(i) do { } while (0)
#define RT_STEPS(n) do { } while (0)
#define RT_FLUSH() do { } while (0)
#endif

struct Smem {
    u32x4 *Kr, *Vr, *Qr;
    f32x4 *Ws, *Ls, *Rr, *Xb, *TapW, *Wd, *Wfs;
    u32x4 *TapO, *LrRow;
    float *Bfs;
};
// This workgroup's pieces of work; a piece = `S` steps (row pairs) of one 16-column strip of one frame from row `ys` on, and costs S + 9
// iterations (the rings fill and drain).  Thread 0 writes the list to LDS once (at most MAXPIECES entries {frame, strip, ys, S}); every
// role walks it.
//   seg_rows given: fixed segments -- units (frame, segment, strip); XCD x owns a contiguous run of them, its workgroups take neighbouring
//   strips of one segment row at the same time.
//   default: the strips of all frames, frame-major, are dealt to the XCDs in contiguous runs; the G workgroups of an XCD walk G neighbouring
//   strips top to bottom at the same time (the halo columns neighbours share are fetched into that L2 once), as often as that goes, and
//   the strips that are left are cut into G equal runs of steps -- every workgroup ends within one fill of every other.  (Fixed 128-row
//   segments: 11 units x 73 iterations per workgroup for the 11-frame headline launch; this: 2 x 265 + 192 + 9 or 18 = 731 .. 740.)
constexpr int MAXPIECES = 64;
struct PieceTab { int count, pad[3]; int4 e[MAXPIECES]; };
__device__ void build_pieces(const RollParams &p, PieceTab *tab) {      // (thread 0)
    const int nx = min(8, (int)gridDim.x);
    const int xcd = blockIdx.x % nx, slot = blockIdx.x / nx;
    const int G = ((int)gridDim.x - xcd + nx - 1) / nx;      // workgroups of this XCD
    const int T = p.nstrips * p.nseg * p.N, a = (int)((long long)T * xcd / nx), b = (int)((long long)T * (xcd + 1) / nx);
    const int npass = p.balanced ? (b - a) / G : (a + slot < b ? (b - a - slot + G - 1) / G : 0);
    int c = 0;
    for (int k = 0; k < npass && c < MAXPIECES; ++k) {
        const int lin = a + slot + k * G, per_img = p.nstrips * p.nseg, n = lin / per_img, rem = lin - n * per_img, seg = rem / p.nstrips;
        const int ys = seg * p.seg_rows;
        tab->e[c++] = int4{n, rem - seg * p.nstrips, ys, (min(p.seg_rows, p.Hp - ys) + 1) >> 1};
    }
    if (p.balanced) {                                         // (nseg == 1: a unit is a whole strip)
        const int SH = (p.Hp + 1) >> 1, s_rem = a + npass * G, tot = (b - s_rem) * SH, rpw = (tot + G - 1) / G;
        int cur = min(slot * rpw, tot);
        const int lin1 = min((slot + 1) * rpw, tot);
        while (cur < lin1 && c < MAXPIECES) {
            const int sidx = cur / SH, y = cur - sidx * SH, lin = s_rem + sidx, S = min(SH - y, lin1 - cur), n = lin / p.nstrips;
            tab->e[c++] = int4{n, lin - n * p.nstrips, 2 * y, S};
            cur += S;
        }
    }
    tab->count = c;
}

// ============================================================================================== consumer: wave (patch pc, key half KH)
// KH 0: key blocks 0..2 (+ merge / epilogue of the previous step), KH 1: key blocks 3..6
template <int NB, int KH>
__device__ __forceinline__ void consumer(const RollParams &p, const Smem &sm, const PieceTab *sc, const int tid, const int wave) {
    constexpr int NBA = NB > 0 ? NB : 1;
    constexpr int B0 = KH ? 3 : 0, NBK = KH ? 4 : 3;
    const int Hp = p.Hp, Wp = p.Wp;
    const int lane = tid & 63, q = lane & 15, g = lane >> 4, pc = wave & 1;
    // window key f = 16b + k0 = 14 ky + kx of the 8 x 14 patch window (k0 = this lane's key of block b): ky = b + (2b + k0 >= 14)
    int kky[NBK], kkx[NBK], vky[NBK], vkx[NBK];
    f32x4 maskv[NBK];                            // 0 where slot (b, i) = key 16b + 4g + i is a real window tap of query q, -inf elsewhere:
    {                                            // the scores are accumulated ON TOP of it (no select per slot and step)
        const int vk0 = 4 * g + (q >> 2), qy = q >> 3, qx = q & 7;
#pragma unroll
        for (int j = 0; j < NBK; ++j) {
            const int b = B0 + j;
            int e = 2 * b + q, up = e >= 14;
            kky[j] = b + up; kkx[j] = e - 14 * up + 8 * pc;
            e = 2 * b + vk0; up = e >= 14;
            vky[j] = b + up; vkx[j] = e - 14 * up + 8 * pc;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = 16 * b + 4 * g + i, ky = f / 14, kx = f - 14 * ky;
                maskv[j][i] = ((unsigned)(ky - qy) <= 6u && (unsigned)(kx - qx) <= 6u) ? 0.f : -INFINITY;
            }
        }
    }
    f32x4 bias[NBA];                             // classifier bias of this lane's classes 16nb + 4g .. + 3 (-inf beyond n_cls: such a class
#pragma unroll                                   // drops out of the log-softmax by itself)
    for (int nb = 0; nb < NBA; ++nb) bias[nb] = KH == 0 && NB > 0 ? *reinterpret_cast<const f32x4 *>(sm.Bfs + nb * 16 + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
    RT_DECL;
    const int npieces = __builtin_amdgcn_readfirstlane(sc->count);
    for (int piece = 0; piece < npieces; ++piece) {
        const int4 pe = sc->e[piece];      // (read by all lanes, the same entry: readfirstlane keeps it -- and the descriptors built from it -- scalar)
        const int n = __builtin_amdgcn_readfirstlane(pe.x), strip = __builtin_amdgcn_readfirstlane(pe.y);
        const int ys = __builtin_amdgcn_readfirstlane(pe.z), S = __builtin_amdgcn_readfirstlane(pe.w);
        const int x0 = strip * SW;
        RT_STEPS(S);
        // one descriptor PER FRAME (scalar arithmetic per piece): the 32-bit buffer offsets then only have to span a frame, not the batch
        const __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.p_out + (size_t)n * (size_t)(CH * Hp) * Wp, 0, (int)p.p_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t l_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.logits + (size_t)n * (size_t)(p.n_cls * Hp) * Wp, 0, (int)p.l_bytes, 0x00020000);
        // Store offsets of this lane's query pixel, split into a per-lane part that is constant down the strip (voffset; OOB for lanes
        // whose column or class lies outside) and a wave-uniform part that moves with the step (soffset: scalar arithmetic only).
        const int qy = q >> 3, gxq = x0 + 8 * pc + (q & 7);
        const bool col_ok = gxq < Wp, c8 = p.p_layout == ARSEG_C8;
        const unsigned plane = (unsigned)(Hp * Wp);
        const unsigned vp = !col_ok ? OOB : c8 ? ((unsigned)(g >> 1) * plane + (unsigned)(qy * Wp + gxq)) * 32u + (unsigned)(g & 1) * 16u
                                               : (unsigned)(qy * Wp + gxq) * (CH * 4u) + 16u * g;
        const unsigned p_cstep = c8 ? 2u * plane * 32u : 64u;                                    // chunk c: + c * p_cstep
        const unsigned p_unit = (unsigned)(ys * Wp) * (c8 ? 32u : CH * 4u);
        const unsigned p_rstep = (unsigned)(2 * Wp) * (c8 ? 32u : CH * 4u);                     // step s: + s * p_rstep
        unsigned vl[NBA][4];
#pragma unroll
        for (int nb = 0; nb < NBA; ++nb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cls = nb * 16 + 4 * g + i;
                vl[nb][i] = col_ok && cls < p.n_cls ? ((unsigned)cls * plane + (unsigned)(qy * Wp + gxq)) * 4u : OOB;
            }
        const unsigned l_unit = (unsigned)(ys * Wp) * 4u, l_rstep = (unsigned)(2 * Wp) * 4u;
        f32x4 Oh[4];                                 // this half's un-normalised P.V (KH 0: carried to the merge in the next H1)
        float mh = 0.f, zh = 1.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) Oh[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int t = T_FIRST; t <= S + 5; ++t) {
            const int s = t - 5;
            u32x4 P[NBK], P2[NBK];
            f32x4 lgc[NBA];                              // logits of step s - 1 before the log-softmax: finished in H2 (this wave's H1 is its longer half)
            // ---------------------------------------------------------------- H1 (KH 0 first): merge the halves of step s - 1, residual, classifier, stores
            if (KH == 0 && s >= 1) {
                // rows of the step: ys + 2(s-1), + 1; the second one may lie below the image (odd height): those lanes store nothing
                const bool rowok = qy == 0 || ys + 2 * (s - 1) + 1 < Hp;
                const unsigned p_s = p_unit + (unsigned)(s - 1) * p_rstep;
                const f32x4 *xb = sm.Xb + pc * 5 * 64 + lane;
                const f32x4 mz = xb[4 * 64];
                // mh / mz[0] are the ROUNDED exponent offsets m * log2(e) each half subtracted from its scores: the same numbers here, so
                // that their rounding cancels between the halves exactly as it cancels inside one softmax
                const float M = fmaxf(mh, mz[0]);
                const float a0 = __builtin_amdgcn_exp2f(mh - M), a1 = __builtin_amdgcn_exp2f(mz[0] - M);
                const float inv = 1.0f / (zh * a0 + mz[1] * a1);
                const float s0 = a0 * inv, s1 = a1 * inv;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4 o = sm.Rr[(pc * 16 + 4 * c + g) * 16 + q] + (Oh[c] * s0 + xb[c * 64] * s1);      // p[query][16c + 4g .. +3]
                    if (rowok) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), p_rsrc, vp, p_s + (unsigned)c * p_cstep, ROLL_PNT);
                    if (NB > 0) {
                        const u32x4 os = split4r(o);
                        const h16x8 o1 = __builtin_bit_cast(h16x8, os), o2 = __builtin_bit_cast(h16x8, u32x4{os.z, os.w, os.x, os.y});
#pragma unroll
                        for (int nb = 0; nb < NBA; ++nb) {
                            const h16x8 wa = __builtin_bit_cast(h16x8, sm.Wfs[(c * 4 + g) * NBA * 16 + nb * 16 + q]);
                            lgc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, o1, c == 0 ? bias[nb] : lgc[nb], 0, 0, 0);
                            lgc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, o2, lgc[nb], 0, 0, 0);
                        }
                    }
                }
            }
            // ---------------------------------------------------------------- H1: scores S[b][i] = q . key(16b + 4g + i) over this half's blocks, softmax
            if (s >= 0 && s < S) {
                const int b8 = (2 * s) & 7;
                int krec[NBK];
#pragma unroll
                for (int j = 0; j < NBK; ++j) krec[j] = kslot(kky[j] + b8) * RW + kkx[j];
                f32x4 Sc[NBK];
#pragma unroll
                for (int j = 0; j < NBK; ++j) Sc[j] = maskv[j];
                // operands one chunk ahead of the MFMAs that use them (left alone, hipcc requests a key record right in front of its two
                // MFMAs: one LDS round trip per block and chunk on the critical path of the wave)
                u32x4 qv[4], ka[2][NBK];
#pragma unroll
                for (int c = 0; c < 4; ++c) qv[c] = sm.Qr[((pc * 4 + c) * 4 + g) * 16 + q];
#pragma unroll
                for (int j = 0; j < NBK; ++j) ka[0][j] = sm.Kr[g * KPL + krec[j]];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c < 3) {
#pragma unroll
                        for (int j = 0; j < NBK; ++j) ka[(c + 1) & 1][j] = sm.Kr[(4 * (c + 1) + g) * KPL + krec[j]];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const h16x8 b1 = __builtin_bit_cast(h16x8, qv[c]), b2 = __builtin_bit_cast(h16x8, u32x4{qv[c].z, qv[c].w, qv[c].x, qv[c].y});
#pragma unroll
                    for (int j = 0; j < NBK; ++j) Sc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, ka[c & 1][j]), b1, Sc[j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < NBK; ++j) Sc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, ka[c & 1][j]), b2, Sc[j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                float m = -INFINITY;
#pragma unroll
                for (int j = 0; j < NBK; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) m = fmaxf(m, Sc[j][i]);
                m = rows_max(m);                           // finite: both halves hold real taps of every query
                const float ml = m * LOG2E;
                float z = 0.f;
#pragma unroll
                for (int j = 0; j < NBK; ++j) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        Sc[j][i] = __builtin_amdgcn_exp2f(fmaf(Sc[j][i], LOG2E, -ml));      // masked slots: exp2(-inf) = 0
                        z += Sc[j][i];
                    }
                    P[j] = split4r(Sc[j]);
                    P2[j] = u32x4{P[j].z, P[j].w, P[j].x, P[j].y};      // {lo | hi}: the second MFMA of a product swaps THIS operand, once per step
                }
                mh = ml; zh = rows_sum(z);
            }
            RT(0);
            wg_sync();
            RT(1);
            // ---------------------------------------------------------------- H2 (KH 0 first): log-softmax + logits stores of step s - 1
            if (KH == 0 && NB > 0 && s >= 1) {
                const bool rowok = qy == 0 || ys + 2 * (s - 1) + 1 < Hp;
                const unsigned l_s = l_unit + (unsigned)(s - 1) * l_rstep;
                f32x4 lg[NBA];
#pragma unroll
                for (int nb = 0; nb < NBA; ++nb) lg[nb] = lgc[nb];
                // logits: lg[nb][i] = class 16nb + 4g + i of query q (-inf for classes beyond n_cls)
                if (p.log_softmax) {
                    float m = -INFINITY;
#pragma unroll
                    for (int nb = 0; nb < NBA; ++nb)
#pragma unroll
                        for (int i = 0; i < 4; ++i) m = fmaxf(m, lg[nb][i]);
                    m = rows_max(m);
                    float z = 0.f;
#pragma unroll
                    for (int nb = 0; nb < NBA; ++nb)
#pragma unroll
                        for (int i = 0; i < 4; ++i) z += __expf(lg[nb][i] - m);
                    z = rows_sum(z);
                    const float lse = m + __logf(z);
#pragma unroll
                    for (int nb = 0; nb < NBA; ++nb) lg[nb] -= lse;
                }
                if (rowok) {
#pragma unroll
                    for (int nb = 0; nb < NBA; ++nb)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(lg[nb][i]), l_rsrc, vl[nb][i], l_s, ROLL_PNT);
                }
            }
            // ---------------------------------------------------------------- H2: P.V over this half's blocks (un-normalised)
            if (s >= 0 && s < S) {
                // byte offset of the value record of (block b, this lane's key) in a channel-group plane
                const int b8 = (2 * s) & 7;
                unsigned vrec[NBK];
#pragma unroll
                for (int j = 0; j < NBK; ++j) vrec[j] = (unsigned)((((vky[j] + b8) & 7) * RW + vkx[j]) * 8);
                // value operands one chunk ahead, as the key records above
                u32x2 vh[2][NBK], vl[2][NBK];
                {
                    const unsigned char *va = reinterpret_cast<const unsigned char *>(sm.Vr + (q & 3) * VPL);
#pragma unroll
                    for (int j = 0; j < NBK; ++j) { vh[0][j] = lds_tr16(va + vrec[j]); vl[0][j] = lds_tr16(va + VHALF + vrec[j]); }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c < 3) {
                        const unsigned char *va = reinterpret_cast<const unsigned char *>(sm.Vr + (4 * (c + 1) + (q & 3)) * VPL);
#pragma unroll
                        for (int j = 0; j < NBK; ++j) { vh[(c + 1) & 1][j] = lds_tr16(va + vrec[j]); vl[(c + 1) & 1][j] = lds_tr16(va + VHALF + vrec[j]); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < NBK; ++j) {
                        const h16x8 va8 = pack8(vh[c & 1][j], vl[c & 1][j]);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(va8, __builtin_bit_cast(h16x8, P[j]), acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(va8, __builtin_bit_cast(h16x8, P2[j]), acc, 0, 0, 0);
                    }
                    Oh[c] = acc;
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (KH == 1) {
                    f32x4 *xb = sm.Xb + pc * 5 * 64 + lane;
#pragma unroll
                    for (int c = 0; c < 4; ++c) xb[c * 64] = Oh[c];
                    xb[4 * 64] = f32x4{mh, zh, 0.f, 0.f};
                }
            }
            RT(2);
            wg_sync();
            RT(3);
        }
    }
    RT_FLUSH();
}

// ============================================================================================== producers (waves 4..15)
// Every producer lane owns one gather unit (staged warp pixel, channel group) and -- lanes 0..575 -- one lr_up unit; on top of that
// a wave has ONE role (a template parameter: the roles' registers never coexist):
//   ROLE_KV  (waves 4..9)           lane = (channel group, record column): key + value depthwise convs, records into the rings
//   ROLE_Q   (waves 10, 11, 14, 15) lane = (channel group, query column): query depthwise conv, query + residual records
//   ROLE_AUX (waves 12, 13)         wave 13: sampling taps of the gather two iterations ahead (one lane per staged pixel)
// Gather pipeline (j = index of a pair of warp rows):  MVs of j requested in H2(j-3)  ->  fp64 grid arithmetic in H1(j-2)  ->  tap table in
// H2(j-2)  ->  the four taps requested in H1(j-1), in flight for a whole iteration (the first touch of a keyframe row comes from HBM)  ->
// blended and staged in H1(j)  ->  consumed by the convs in H2(j).
enum { ROLE_KV = 0, ROLE_Q = 1, ROLE_AUX = 2 };
template <int ROLE>
__device__ __forceinline__ void producer(const RollParams &p, const Smem &sm, const PieceTab *sc, const int tid, const int wave) {
    const int Hp = p.Hp, Wp = p.Wp;
    const int pl = tid - 64 * NCONS;                   // 0 .. 767
    const int gpx = pl >> 4, gcg = pl & 15;            // gather unit: staged warp pixel (row gpx / 24, column gpx % 24), channel group
    const int grr = gpx >= GW ? 1 : 0, gcc = gpx - GW * grr;
    // lr_up units (staged lr_up pixel, channel group): 576 of them -- two on every lane of the four query waves (units ql, ql + 256), one
    // on every lane of wave 12 (512 + lane); the channel group of a unit is the lane's gather channel group (all offsets are multiples of 16)
    constexpr int NLU = ROLE == ROLE_Q ? 2 : ROLE == ROLE_AUX ? 1 : 0;
    const bool kv_ok = pl < KV_LANES;                  // key / value lane: (channel group, record column), column fastest
    const int kcg = min(pl / RW, 15), kx = pl - RW * (pl / RW);
    // query lane: (channel group, query column); the four query waves are 10, 11, 14, 15 (see the role table in the kernel)
    const int ql = ((wave < 12 ? wave - 10 : wave - 12) & 3) * 64 + (tid & 63), qcg = (ql >> 4) & 15, qx = ql & 15;
    const int tl = tid & 63;                           // tap lane: wave 13's lanes 0 .. 47, one per staged warp pixel
    const bool tap_lane = ROLE == ROLE_AUX && wave == 13 && tl < NGP;
    const int trr = tl >= GW ? 1 : 0, tcc = tl - GW * trr;
    const bool mv_ident = Hp == p.H && Wp == p.W;
    double g_dW = 0.0, g_dH = 0.0, g_rW = 0.0, g_rH = 0.0;
    if (ROLE == ROLE_AUX) {                            // grid normalisation: extents and their reciprocals
        g_dW = uniform_f64((double)max(Wp - 1, 1)); g_dH = uniform_f64((double)max(Hp - 1, 1));
        g_rW = uniform_f64(1.0 / g_dW); g_rH = uniform_f64(1.0 / g_dH);
    }
    RT_DECL;

    const int npieces = __builtin_amdgcn_readfirstlane(sc->count);
    for (int piece = 0; piece < npieces; ++piece) {
        const int4 pe = sc->e[piece];      // (read by all lanes, the same entry: readfirstlane keeps it -- and the descriptors built from it -- scalar)
        const int n = __builtin_amdgcn_readfirstlane(pe.x), strip = __builtin_amdgcn_readfirstlane(pe.y);
        const int ys = __builtin_amdgcn_readfirstlane(pe.z), S = __builtin_amdgcn_readfirstlane(pe.w);
        const int x0 = strip * SW;
        RT_STEPS(S);
        const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.ref[n]), 0, (int)p.ref_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t lr_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.lr) + (size_t)n * (size_t)(p.hp * p.wp) * CH, 0, (int)p.lr_bytes, 0x00020000);
        const unsigned lr_img = 16u * gcg;           // (the frame is in the descriptor's base)
        const bool x_inner = x0 >= 3 && x0 + SW + 3 <= Wp;      // every record column of the strip inside the image
        // lr_up column taps of this lane's units: the same for every row of the strip
        bool l_lane[NLU > 0 ? NLU : 1];
        int lrr[NLU > 0 ? NLU : 1], lcc[NLU > 0 ? NLU : 1];
        unsigned lx0[NLU > 0 ? NLU : 1], lx1[NLU > 0 ? NLU : 1];
        float lwx0[NLU > 0 ? NLU : 1], lwx1[NLU > 0 ? NLU : 1];
#pragma unroll
        for (int i = 0; i < NLU; ++i) {
            const int px = (ROLE == ROLE_Q ? ql + 256 * i : 512 + (tid & 63)) >> 4;
            l_lane[i] = px < NLP && (ROLE == ROLE_Q || wave == 12);
            lrr[i] = px >= LW ? 1 : 0; lcc[i] = px - LW * lrr[i];
            const int gx = x0 - 1 + lcc[i];
            int j0, j1; float m;
            arseg_src_index(p.sx, min(max(gx, 0), Wp - 1), true, p.wp, j0, j1, m);
            m = fminf(fmaxf(m, 0.f), 1.f);
            const float inx = (unsigned)gx < (unsigned)Wp ? 1.f : 0.f;
            lwx0[i] = (1.f - m) * inx; lwx1[i] = m * inx;
            lx0[i] = (unsigned)j0 * (CH * 4u); lx1[i] = (unsigned)j1 * (CH * 4u);
            if (!l_lane[i]) { lx0[i] = OOB - lr_img; lx1[i] = OOB - lr_img; }      // (wave 13: no unit -- with the row offset still beyond the buffer)
        }
        // ROLE_KV: the four warp rows under the two record rows being produced (rows 0, 1: the window; 2, 3: staged this iteration);
        // ROLE_Q: rows 0, 1 = the window of lr_up rows
        f32x4 row[ROLE == ROLE_KV ? 4 : 2][3], gw = {0.f, 0.f, 0.f, 0.f}, savA = {0.f, 0.f, 0.f, 0.f};
        f32x4 wreg[10];                              // the depthwise weights of the NEXT half step's stencil (ROLE_KV: key in H2 / value in H1)
#pragma unroll
        for (int j = 0; j < 10; ++j) wreg[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned mvv = 0u, pft[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        u32x4 go = {0u, 0u, 0u, 0u};
        float ngx = 0.f, ngy = 0.f;
#pragma unroll
        for (int i = 0; i < (ROLE == ROLE_KV ? 4 : 2); ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) row[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int t = T_FIRST; t <= S + 5; ++t) {
            // ================================================================ H1
            f32x4 gv[4];
#if ROLL_LPRIO
            __builtin_amdgcn_s_setprio(3);      // every wave's requests leave before anybody's arithmetic (the youngest waves of a SIMD otherwise issue theirs last)
#endif
            if (ROLE == ROLE_KV) {      // two of the four gather taps travel under the value conv (all four: 16 more registers than the conv leaves)
#pragma unroll
                for (int k = 0; k < 2; ++k) gv[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, go[k] + 16u * gcg, 0, 0));
            }
            if (ROLE == ROLE_KV) {
                // ---- value records of rows rho = 2k, 2k + 1 (k = t - 2) from the four warp rows the key conv of H2(t-1) used (weights requested at
                // the end of H2(t-1)); then the window moves on
                if (kv_ok && t >= 1 && t <= S + 4) {
                    if (t >= 2) {
                        const int k = t - 2, r0 = ys - 3 + 2 * k;
                        const bool col_in = (unsigned)(x0 - 3 + kx) < (unsigned)Wp;
                        const bool in_a = col_in && (unsigned)r0 < (unsigned)Hp, in_b = col_in && (unsigned)(r0 + 1) < (unsigned)Hp;
                        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                        f32x4 a, b;
                        stencil2r(wreg, row[0], row[1], row[2], row[3], a, b);
                        if (!(x_inner && r0 >= 0 && r0 + 1 < Hp)) { a = in_a ? a : zero; b = in_b ? b : zero; }      // (wave uniform)
                        u32x2 *vdst = reinterpret_cast<u32x2 *>(reinterpret_cast<unsigned char *>(sm.Vr) + kcg * (VPL * 16)) + ((2 * k) & 7) * RW + kx;
                        const u32x4 ra = split4r(a), rb = split4r(b);
                        vdst[0] = u32x2{ra.x, ra.y}; vdst[VHALF / 8] = u32x2{ra.z, ra.w};
                        vdst[RW] = u32x2{rb.x, rb.y}; vdst[VHALF / 8 + RW] = u32x2{rb.z, rb.w};
                    }
#pragma unroll
                    for (int j = 0; j < 3; ++j) { row[0][j] = row[2][j]; row[1][j] = row[3][j]; }
                }
            }
            // ---- gather t: the four taps of this lane's pixel (tap offsets read in H1(t-1); the lines were touched by the tap wave in H2(t-2),
            // so these come from the L2: no load of a compute wave is in flight across a barrier -- hipcc's s_waitcnt bookkeeping otherwise
            // makes the LDS reads of H2 wait for them)
#ifdef ROLL_NOGATHER
            const bool g_on = false;
#else
            const bool g_on = t >= 0 && t <= S + 3;
#endif
#pragma unroll
            for (int k = ROLE == ROLE_KV ? 2 : 0; k < 4; ++k)      // (outside [0, S+3] the offsets are stale but valid: the values are not used)
                gv[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, go[k] + 16u * gcg, 0, 0));
            // ---- lr_up rows ys + 2t - 7, ys + 2t - 6 (+1 halo column each side): the bilinear taps (lines touched by wave 12 in H2(t-2))
#ifdef ROLL_NOLR
            const bool l_on = false;
#else
            const bool l_on = t >= 3 && t <= S + 3;
#endif
            f32x4 lv[NLU > 0 ? NLU : 1][4];
            float lwy0[NLU > 0 ? NLU : 1], lwy1[NLU > 0 ? NLU : 1];
            {                                                        // (unconditional: hipcc spills registers that loads define under a branch;
#pragma unroll                                                       //  outside [3, S+3] the table rows are stale, the values unused)
                for (int i = 0; i < NLU; ++i) {
#if ROLL_LRTAB
                    const u32x4 rt = sm.LrRow[lrr[i]];              // the row taps of the iteration (tap wave, H2(t-1))
                    lwy0[i] = __uint_as_float(rt.z); lwy1[i] = __uint_as_float(rt.w);
                    const unsigned r0 = lr_img + rt.x, r1 = lr_img + rt.y;
#else
                    const int gy = ys + 2 * t - 7 + lrr[i];
                    int i0, i1; float l;
                    arseg_src_index(p.sy, min(max(gy, 0), Hp - 1), true, p.hp, i0, i1, l);
                    l = fminf(fmaxf(l, 0.f), 1.f);
                    const float iny = (unsigned)gy < (unsigned)Hp ? 1.f : 0.f;
                    lwy0[i] = (1.f - l) * iny; lwy1[i] = l * iny;
                    const unsigned r0 = lr_img + (unsigned)(i0 * p.wp) * (CH * 4u), r1 = lr_img + (unsigned)(i1 * p.wp) * (CH * 4u);
#endif
                    lv[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lr_rsrc, r0 + lx0[i], 0, 0));
                    lv[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lr_rsrc, r0 + lx1[i], 0, 0));
                    lv[i][2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lr_rsrc, r1 + lx0[i], 0, 0));
                    lv[i][3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lr_rsrc, r1 + lx1[i], 0, 0));
                }
            }
#if ROLL_LPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            // ---- sampling position of gather t + 2 (warp rows ys - 4 + 2(t+2), +1), fp64 like the reference; the MV was requested in H2(t-1)
            if (ROLE == ROLE_AUX && tap_lane && t >= -2 && t <= S + 1) {
                const int gy = ys + 2 * t + trr, gx = x0 - 4 + tcc;
                double fx, fy;
                if (mv_ident) {                        // identity resize (PSPNet): (q/4 * Hp) / H == q/4 exactly
                    fx = (double)(short)(mvv & 0xFFFFu) / 4.0; fy = (double)(short)(mvv >> 16) / 4.0;
                } else {
                    mv_at(p.mv + (size_t)n * p.H * p.W * 2, p.H, p.W, Hp, Wp, min(max(gy, 0), Hp - 1), min(max(gx, 0), Wp - 1), fx, fy);
                }
                norm_grid_rcp(gx, gy, fx, fy, g_dW, g_dH, g_rW, g_rH, ngx, ngy);
            }
            // ---- lr_up: interpolate, stage
#pragma unroll
            for (int i = 0; i < NLU; ++i)
                if (l_on && l_lane[i])
                    sm.Ls[(lrr[i] * 16 + gcg) * LPL + lcc[i]] = lwy0[i] * (lwx0[i] * lv[i][0] + lwx1[i] * lv[i][1]) + lwy1[i] * (lwx0[i] * lv[i][2] + lwx1[i] * lv[i][3]);
            // ---- gather t: blend, stage; then the tap offsets / weights of gather t + 1 (table of H2(t-1))
            if (g_on) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc += gv[0] * (gw[0] * gw[2]);      // same order as warp_mvq_nhwc_kernel
                acc += gv[1] * (gw[1] * gw[2]);
                acc += gv[2] * (gw[0] * gw[3]);
                acc += gv[3] * (gw[1] * gw[3]);
                sm.Ws[(grr * 16 + gcg) * WPL + gcc] = acc;
            }
            if (t >= -1 && t <= S + 2) { go = sm.TapO[gpx]; gw = sm.TapW[gpx]; }
            if (ROLE == ROLE_KV) {
#pragma unroll
                for (int j = 0; j < 10; ++j) wreg[j] = sm.Wd[kcg * 10 + j];
            }
            f32x4 wqa[5];
            if (ROLE == ROLE_Q) {
#pragma unroll
                for (int j = 0; j < 5; ++j) wqa[j] = sm.Wd[320 + qcg * 10 + j];
            }
            RT(0);
            wg_sync();
            RT(1);
            // ================================================================ H2
            if (ROLE == ROLE_KV) {
                // ---- key records of rows rho = 2k, 2k + 1 (k = t - 1; image rows ys - 3 + rho); the value records follow in H1(t+1)
                if (kv_ok && t >= 0 && t <= S + 3) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) { row[2][j] = sm.Ws[kcg * WPL + kx + j]; row[3][j] = sm.Ws[(16 + kcg) * WPL + kx + j]; }
                    if (t >= 1) {
                        const int k = t - 1, r0 = ys - 3 + 2 * k;
                        const bool col_in = (unsigned)(x0 - 3 + kx) < (unsigned)Wp;
                        // the unfold's zero padding: records outside the image are zero
                        const bool in_a = col_in && (unsigned)r0 < (unsigned)Hp, in_b = col_in && (unsigned)(r0 + 1) < (unsigned)Hp;
                        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                        f32x4 a, b;
                        stencil2r(wreg, row[0], row[1], row[2], row[3], a, b);
                        if (!(x_inner && r0 >= 0 && r0 + 1 < Hp)) { a = in_a ? a : zero; b = in_b ? b : zero; }      // (wave uniform)
                        sm.Kr[kcg * KPL + kslot(2 * k) * RW + kx] = split4r(a);
                        sm.Kr[kcg * KPL + kslot(2 * k + 1) * RW + kx] = split4r(b);
                    }
                }
#pragma unroll
                for (int j = 0; j < 10; ++j) wreg[j] = sm.Wd[160 + kcg * 10 + j];
            } else if (ROLE == ROLE_Q) {
                // ---- residual records of step t - 5: lr_up at the query pixels (row A saved in H2(t-1), row B still in the window)
                if (t >= 5 && t <= S + 4) {
                    f32x4 *dst = sm.Rr + ((qx >> 3) * 16 + qcg) * 16 + (qx & 7);
                    dst[0] = savA; dst[8] = row[0][1];
                }
                // ---- query records of step s = t - 4 (query rows ys + 2s, + 1)
                if (l_on) {
                    f32x4 m0[3], m1[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) { m0[j] = sm.Ls[qcg * LPL + qx + j]; m1[j] = sm.Ls[(16 + qcg) * LPL + qx + j]; }
                    if (t >= 4) {
                        f32x4 a, b;
                        stencil2h(wqa, sm.Wd + 320 + qcg * 10, row[0], row[1], m0, m1, a, b);
                        u32x4 *dst = sm.Qr + (((qx >> 3) * 4 + (qcg >> 2)) * 4 + (qcg & 3)) * 16 + (qx & 7);
                        dst[0] = split4r(a); dst[8] = split4r(b);
                        savA = row[1][1];
                    }
#pragma unroll
                    for (int j = 0; j < 3; ++j) { row[0][j] = m0[j]; row[1][j] = m1[j]; }
                }
            } else {
                // ---- motion vectors of gather t + 3 (identity-resize case: one int16 pair per pixel).  FIRST: behind the touch loads below,
                // hipcc puts an s_waitcnt vmcnt(0) in front of this request (registers of loads it cannot prove finished) and the wave --
                // the longest of H2 -- sits out the HBM latency of the lines it has just touched
                if (tap_lane && mv_ident && t <= S) {
                    const int gy = ys + 2 * t + 2 + trr, gx = x0 - 4 + tcc;
                    mvv = 0u;
                    if ((unsigned)gy < (unsigned)Hp && (unsigned)gx < (unsigned)Wp)
                        mvv = *reinterpret_cast<const unsigned *>(p.mv + ((size_t)n * p.H * p.W + (size_t)gy * p.W + gx) * 2);
                }
                // ---- tap table of gather t + 2 from the sampling positions of H1(t)
                if (tap_lane && t >= -2 && t <= S + 1) {
                    const int gy = ys + 2 * t + trr, gx = x0 - 4 + tcc;
                    f32x4 w = {0.f, 0.f, 0.f, 0.f};
                    u32x4 o = {0u, 0u, 0u, 0u};
                    if ((unsigned)gy < (unsigned)Hp && (unsigned)gx < (unsigned)Wp) {      // outside the image the warped feature is zero (conv padding)
                        const Taps tp = make_taps(ngx, ngy, Hp, Wp);
                        const int xa = min(max(tp.x0, 0), Wp - 1), xc = min(max(tp.x0 + 1, 0), Wp - 1);
                        const int ya = min(max(tp.y0, 0), Hp - 1), yc = min(max(tp.y0 + 1, 0), Hp - 1);
                        o = u32x4{(unsigned)(ya * Wp + xa), (unsigned)(ya * Wp + xc), (unsigned)(yc * Wp + xa), (unsigned)(yc * Wp + xc)} * (CH * 4u);
                        w = f32x4{tp.vx0 ? tp.ex : 0.f, tp.vx1 ? tp.wx : 0.f, tp.vy0 ? tp.ey : 0.f, tp.vy1 ? tp.wy : 0.f};
                    }
                    sm.TapW[tl] = w; sm.TapO[tl] = o;
                    // touch the eight 128-byte lines of the four tap pixels: the gather lanes read them one and a half iterations from now
                    asm volatile("" :: "v"(pft[0]), "v"(pft[1]), "v"(pft[2]), "v"(pft[3]), "v"(pft[4]), "v"(pft[5]), "v"(pft[6]), "v"(pft[7]));
#pragma unroll
                    for (int i = 0; i < 8; ++i) pft[i] = __builtin_amdgcn_raw_buffer_load_b32(g_rsrc, o[i >> 1] + ((i & 1) ? 128u : 0u), 0, 0);
                }
                // ---- lr row taps of iteration t + 1 (lr_up rows ys + 2t - 5, ys + 2t - 4): lanes 48, 49 of the tap wave
                if (ROLL_LRTAB && wave == 13 && (tl == NGP || tl == NGP + 1) && t >= 2 && t <= S + 2) {
                    const int gy = ys + 2 * t - 5 + (tl - NGP);
                    int i0, i1; float l;
                    arseg_src_index(p.sy, min(max(gy, 0), Hp - 1), true, p.hp, i0, i1, l);
                    l = fminf(fmaxf(l, 0.f), 1.f);
                    const float iny = (unsigned)gy < (unsigned)Hp ? 1.f : 0.f;
                    sm.LrRow[tl - NGP] = u32x4{(unsigned)(i0 * p.wp) * (CH * 4u), (unsigned)(i1 * p.wp) * (CH * 4u), __float_as_uint((1.f - l) * iny), __float_as_uint(l * iny)};
                }
                // ---- wave 12 touches the lr pixels under the lr_up rows of iteration t + 2 (rows ys + 2t - 3, ys + 2t - 2): lane = (lr row 0..3 from
                // the first tap row, lr column 0..15 from the first tap column of the strip), both lines of the pixel
                if (wave == 12 && t >= 1 && t <= S + 1) {
                    asm volatile("" :: "v"(pft[0]), "v"(pft[1]));
                    const int tq = tid & 63;
                    int i0, i1, ie0, ie1, j0, j1, je0, je1; float l;
                    arseg_src_index(p.sy, min(max(ys + 2 * t - 3, 0), Hp - 1), true, p.hp, i0, i1, l);
                    arseg_src_index(p.sy, min(max(ys + 2 * t - 2, 0), Hp - 1), true, p.hp, ie0, ie1, l);
                    arseg_src_index(p.sx, min(max(x0 - 1, 0), Wp - 1), true, p.wp, j0, j1, l);
                    arseg_src_index(p.sx, min(max(x0 + SW, 0), Wp - 1), true, p.wp, je0, je1, l);
                    const int r = i0 + (tq >> 4), c = j0 + (tq & 15);
                    const unsigned off = r <= ie1 && c <= je1 ? (unsigned)(r * p.wp + c) * (CH * 4u) : OOB;
                    pft[0] = __builtin_amdgcn_raw_buffer_load_b32(lr_rsrc, off, 0, 0);
                    pft[1] = __builtin_amdgcn_raw_buffer_load_b32(lr_rsrc, off == OOB ? OOB : off + 128u, 0, 0);
                }
            }
            RT(2);
            wg_sync();
            RT(3);
        }
    }
    RT_FLUSH();
}

template <int NB>      // NB: classifier row blocks of 16 classes (0: no head)
__global__ __launch_bounds__(NT) void creff_roll_kernel(const RollParams p) {
    constexpr int NBA = NB > 0 ? NB : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Smem sm;
    sm.Kr = reinterpret_cast<u32x4 *>(smem + K_OFF); sm.Vr = reinterpret_cast<u32x4 *>(smem + V_OFF);
    sm.Ws = reinterpret_cast<f32x4 *>(smem + WS_OFF); sm.Ls = reinterpret_cast<f32x4 *>(smem + LS_OFF);
    sm.Qr = reinterpret_cast<u32x4 *>(smem + Q_OFF); sm.Rr = reinterpret_cast<f32x4 *>(smem + R_OFF);
    sm.Xb = reinterpret_cast<f32x4 *>(smem + XB_OFF); sm.TapW = reinterpret_cast<f32x4 *>(smem + TW_OFF);
    sm.TapO = reinterpret_cast<u32x4 *>(smem + TO_OFF); sm.LrRow = reinterpret_cast<u32x4 *>(smem + LR_OFF);
    sm.Wd = reinterpret_cast<f32x4 *>(smem + WD_OFF);
    sm.Wfs = reinterpret_cast<f32x4 *>(smem + WF_OFF);        // [4 chunks][4 groups][NBA*16]
    sm.Bfs = reinterpret_cast<float *>(smem + BF_OFF);

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ------------------------------------------------------------------ once per launch: weight tables
    for (int e = tid; e < 3 * 160; e += NT) {
        const int which = e / 160, r = e - which * 160, cg = r / 10, tp = r - cg * 10;
        const float *w = which == 0 ? p.wk : which == 1 ? p.wv : p.wq, *b = which == 0 ? p.bk : which == 1 ? p.bv : p.bq;
        sm.Wd[e] = *reinterpret_cast<const f32x4 *>(tp < 9 ? w + tp * CH + cg * 4 : b + cg * 4);
    }
    if (NB > 0) {
        for (int e = tid; e < 4 * 4 * NBA * 16; e += NT) {
            const int cls = e % (NBA * 16), gg = (e / (NBA * 16)) & 3, c = e / (4 * NBA * 16);
            f32x4 wv4 = {0.f, 0.f, 0.f, 0.f};
            if (cls < p.n_cls) wv4 = *reinterpret_cast<const f32x4 *>(p.wf + (size_t)cls * CH + c * 16 + gg * 4);
            sm.Wfs[e] = __builtin_bit_cast(f32x4, split4r(wv4));
        }
        if (tid < NBA * 16) sm.Bfs[tid] = tid < p.n_cls ? p.bf[tid] : -INFINITY;
    }
    __syncthreads();

    // Persistent workgroups, XCD-aware order (as creff_rr.hip): this workgroup's list of pieces (struct PieceTab)
    PieceTab *sc = reinterpret_cast<PieceTab *>(smem + SC_OFF);
    if (tid == 0) build_pieces(p, sc);
    __syncthreads();

#ifdef ROLL_ONLY      // dev builds only: one role alone, to read ITS register count off -Rpass-analysis=kernel-resource-usage (0 / 1: consumers, 2: key/value, 3: aux, 4: query)
    if (ROLL_ONLY == 0) consumer<NB, 0>(p, sm, sc, tid, wave);
    if (ROLL_ONLY == 1) consumer<NB, 1>(p, sm, sc, tid, wave);
    if (ROLL_ONLY == 2) producer<ROLE_KV>(p, sm, sc, tid, wave);
    if (ROLL_ONLY == 3) producer<ROLE_AUX>(p, sm, sc, tid, wave);
    if (ROLL_ONLY == 4) producer<ROLE_Q>(p, sm, sc, tid, wave);
    return;
#endif
    if (wave < NCONS) {
        __builtin_amdgcn_s_setprio(ROLL_CPRIO);       // the consumers are the critical path of both halves of an iteration
        if (wave < 2) consumer<NB, 0>(p, sm, sc, tid, wave);
        else consumer<NB, 1>(p, sm, sc, tid, wave);
    } else if (wave < 10) {
        // Roles by SIMD (a workgroup's waves go to the four SIMDs cyclically, so waves w and w + 4 share one): every SIMD hosts one
        // consumer and about the same producer VALU work --  SIMD a: 0 C | 4 KV | 8 KV | 12 AUX     SIMD b: 1 C | 5 KV | 9 KV | 13 AUX (taps)
        //                                                 SIMD c: 2 C | 6 KV | 10 Q | 14 Q       SIMD d: 3 C | 7 KV | 11 Q | 15 Q
        producer<ROLE_KV>(p, sm, sc, tid, wave);
    } else if (wave == 12 || wave == 13) {
        producer<ROLE_AUX>(p, sm, sc, tid, wave);
    } else {
        producer<ROLE_Q>(p, sm, sc, tid, wave);
    }
}

template <int NB>
int launch(const RollParams &p, int max_wgs, hipStream_t st) {
    static ArsegSmemAttr attr;
    if (int e = arseg_allow_smem(attr, reinterpret_cast<const void *>(creff_roll_kernel<NB>), SMEM_BYTES)) return e;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    // workgroups: one per CU; fewer when the launch is small -- mode 0: one per unit, mode 1: at least 16 steps each (a piece costs 9 iterations of fill)
    const long long nunits = p.balanced ? ((long long)p.nstrips * p.N * ((p.Hp + 1) >> 1) + 15) / 16 : (long long)p.nstrips * p.nseg * p.N;
    if (max_wgs > 0 && max_wgs < cus) cus = max_wgs;      // leave compute units to the kernels of other streams (the workgroups are persistent)
    const int grid = (int)(nunits < cus ? nunits : cus);
    hipLaunchKernelGGL((creff_roll_kernel<NB>), dim3(grid), dim3(NT), SMEM_BYTES, st, p);
    return arseg_launch_status();
}

#ifdef ROLL_TIMING
unsigned long long *g_roll_dbg = nullptr;
#endif
}  // namespace

#ifdef ROLL_TIMING
extern "C" void arseg__roll_set_dbg(void *ptr) { g_roll_dbg = (unsigned long long *)ptr; }
#endif

// creff_rr.hip's entry point dispatches here (same argument checks there).
int arseg_creff_roll_launch(const float *const *ref_nhwc_host, const int16_t *mv_q, int H, int W, const float *lr, const float *wq,
                            const float *bq, const float *wk, const float *bk, const float *wv, const float *bv, float *p_out,
                            int p_layout, const float *wf, const float *bf, int n_cls, float *logits, int log_softmax, int N, int Hp,
                            int Wp, int hp, int wp, int seg_rows, int max_wgs, bool dry_run, hipStream_t st) {
    const bool head = dry_run ? n_cls > 0 : logits != nullptr;
    // one head instantiation: up to 16 classes.  A 17-32-class head (the NB = 2 form of rounds 4) spilled in the merge wave and was never
    // selected by AUTO; such heads run on the tile kernel (creff_rr.hip) -- VERDICT r4 item 6.
    if (head && n_cls > 16) return ARSEG_EUNSUPPORTED;
    RollParams p;
    for (int i = 0; i < MAXN; ++i) p.ref[i] = (!dry_run && i < N) ? ref_nhwc_host[i] : nullptr;
    p.mv = mv_q; p.lr = lr; p.wq = wq; p.bq = bq; p.wk = wk; p.bk = bk; p.wv = wv; p.bv = bv; p.wf = wf; p.bf = bf;
    p.p_out = p_out; p.logits = logits;
    p.N = N; p.Hp = Hp; p.Wp = Wp; p.hp = hp; p.wp = wp; p.H = H; p.W = W; p.n_cls = head ? n_cls : 0; p.log_softmax = log_softmax;
    p.p_layout = p_layout;
    p.nstrips = arseg_cdiv(Wp, SW);
    p.balanced = seg_rows <= 0;                               // default: strips dealt in balanced runs of steps (struct PieceTab); seg_rows > 0: fixed segments
    if (seg_rows <= 0) seg_rows = Hp;
    seg_rows = (seg_rows + 1) & ~1;
    if (!p.balanced) {                                        // a workgroup's list holds MAXPIECES pieces: longer segments if the launch has more
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        if (max_wgs > 0 && max_wgs < cus) cus = max_wgs;
        for (;;) {
            const long long units = (long long)p.nstrips * arseg_cdiv(Hp, seg_rows) * N, wgs = units < cus ? units : cus;
            // (an XCD's share of the units over its share of the workgroups, both rounded against us)
            const long long nx = wgs < 8 ? wgs : 8, share = (units + nx - 1) / nx, g = wgs / nx;
            if ((share + g - 1) / g <= MAXPIECES) break;
            //  whole-height segments and the list still does not fit: the pieces beyond MAXPIECES would never be computed
            if (seg_rows >= Hp) return ARSEG_EUNSUPPORTED;
            seg_rows *= 2;
        }
    }
    p.seg_rows = seg_rows; p.nseg = arseg_cdiv(Hp, seg_rows);
    p.p_bytes = (unsigned)((size_t)CH * Hp * Wp * sizeof(float)); p.l_bytes = head ? (unsigned)((size_t)n_cls * Hp * Wp * sizeof(float)) : 0u;
    p.lr_bytes = (unsigned)((size_t)CH * hp * wp * sizeof(float));
    p.ref_bytes = (unsigned)((size_t)CH * Hp * Wp * sizeof(float));
    p.sy = arseg_resize_scale(hp, Hp, true); p.sx = arseg_resize_scale(wp, Wp, true);
    p.dbg = nullptr;
#ifdef ROLL_TIMING
    p.dbg = g_roll_dbg;
#endif
    if (p.balanced) {      // a workgroup's list of pieces must hold its whole-strip passes + the pieces of its remainder run
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        if (max_wgs > 0 && max_wgs < cus) cus = max_wgs;
        const long long T = (long long)p.nstrips * N, nunits = (T * ((Hp + 1) >> 1) + 15) / 16, wgs = nunits < cus ? nunits : cus;
        const long long nx = wgs < 8 ? wgs : 8, share = (T + nx - 1) / nx, g = wgs / nx;
        if (share / g + 3 > MAXPIECES) return ARSEG_EUNSUPPORTED;
    }
    if (dry_run) return ARSEG_OK;
    return head ? launch<1>(p, max_wgs, st) : launch<0>(p, max_wgs, st);
}
