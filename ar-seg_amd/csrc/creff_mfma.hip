// CReFF with the two contractions of the local attention (Q.K^T over channels, P.V over the window) on the gfx950 matrix
// cores.  Same arithmetic contract as creff.hip (reference: MyAttention.forward, model/attention.py:184-213, followed by
// the frozen 1x1 classifier, model/pspnet.py:225-229 / model/bisenet.py:571-572); that file holds the fp32 VALU version,
// which remains the fallback for window sizes other than 7x7 and for shapes this kernel does not cover.
//
// fp32 on fp16 MFMAs.  Every fp32 operand x is split into hi + lo (two fp16, 22 significant bits together).  The K index
// of v_mfma_f32_16x16x32_f16 is packed as [4 hi | 4 lo] per lane, so MFMA(A={a_hi,a_lo}, B={b_hi,b_lo}) yields
// a_hi.b_hi + a_lo.b_lo and MFMA(A, B'={b_lo,b_hi}) the two cross terms: two instructions give the full fp32-grade product.
//
// Geometry.  A workgroup (16 waves) owns a 16x16 pixel tile; a wave owns an 8x2 patch of queries.  The 7x7 windows of a
// patch lie inside 8 rows x 14 columns of keys; padded to 8 x 16 = 128 key slots this is 8 MFMA row blocks, block b =
// window row b, slot kx = window column (49 of the 128 scores per query are real, the rest are masked before the softmax).
//   scores   D[key kx][query] (block b) += K[row b][kx][16 ch] . Q[query][16 ch]      A = K from LDS, B = Q in registers
//   softmax  in registers: a query's 128 slots live in 4 lanes x 32 registers (two xor-shuffles for max and sum)
//   values   D[ch][query] += V[row b][kx][ch] . P[kx][query]    A = V through ds_read_b64_tr_b16 (LDS transpose read),
//            B = P: the score registers are already in B-operand order, nothing moves between lanes
//   head     D[class][query] += Wf[class][16 ch] . p[query][16 ch]   B = the fused feature, again in D order
// Channels are processed in chunks of 16: the warped keyframe chunk (+halo) is staged in LDS, the depthwise 3x3 key /
// value convolution runs on the VALU and writes its result as split fp16 records ([group of 4 ch][px]{4 hi | 4 lo}, zero outside the
// image = the unfold's padding, model/attention.py:56-58), the query convolution is lane-local (each lane convolves the
// 4 channels of its own query that its B operand needs).
#include "creff_params.h"

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));

constexpr int TX = 16;
constexpr int G = 4;                                  // float4 groups per 16-channel chunk
constexpr int RW = 24, RWC = 22;                      // key/value region: record columns, computed columns
constexpr int HWD = 26, LWD = 18;                     // columns of the staged hr chunk (+3 window +1 conv halo) / of the lr_up tile
constexpr int LWCAP = 512;                            // raw lr window capacity (float4): 128 low-resolution pixels
// Tile height TY (16: one 16-wave workgroup per CU with double-buffered hr staging; 8: 8 waves, single-buffered hr staging,
// half the LDS so that two workgroups share a CU and cover each other's barriers and DMA latency).
template <int TY> struct Geo {
    static constexpr int NT = 64 * TY, HBUF = TY == 16 ? 2 : 1;
    static constexpr int RH = TY + 6, HH = TY + 8, LH = TY + 2;
    static constexpr int KPLK = RH * RW, KPLV = RH * RW + 4;   // plane stride of the key / value records: keys are read with ds_read_b128
                                                               // (planes on the same banks), values with the transpose read (16 banks apart)
    static constexpr int HPL = HH * HWD, LPL = LH * LWD + 4;
};
constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr float LOG2E = 1.44269504088896340736f;

__device__ __forceinline__ void split4(const f32x4 v, u32x2 &hi, u32x2 &lo) {
    unsigned h01, h23, l01, l23;
    arseg_split_f16(v, h01, h23, l01, l23);
    hi = u32x2{h01, h23}; lo = u32x2{l01, l23};
}
// {hi, lo, hi}: dwords 0..3 are the operand {hi,lo}, dwords 2..5 the swapped operand {lo,hi} -- no register copies
// reductions over the 4 DPP rows of a wave on the VALU (see creff_rr.hip): __shfl_xor is a ds_bpermute (LDS round trip + an address VGPR)
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
__device__ __forceinline__ u32x6 pack6(const u32x2 hi, const u32x2 lo) { return u32x6{hi.x, hi.y, lo.x, lo.y, hi.x, hi.y}; }
__device__ __forceinline__ h16x8 op_a(const u32x6 v) { return __builtin_bit_cast(h16x8, __builtin_shufflevector(v, v, 0, 1, 2, 3)); }
__device__ __forceinline__ h16x8 op_b(const u32x6 v) { return __builtin_bit_cast(h16x8, __builtin_shufflevector(v, v, 2, 3, 4, 5)); }
__device__ __forceinline__ h16x8 pack8(const u32x2 a, const u32x2 b) { return __builtin_bit_cast(h16x8, u32x4{a.x, a.y, b.x, b.y}); }
__device__ __forceinline__ u32x2 lds_tr16(const unsigned char *p) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)p));
}

// Asynchronous memory traffic is issued through inline asm on purpose.  hipcc (ROCm 7.2) serialises the LDS-DMA builtins
// (a waterfall loop over the M0 base with an s_waitcnt vmcnt(0) in front of every load) and, on gfx9, drains every counter it
// knows about in front of each s_barrier -- so builtin stores would expose the full write latency at the next barrier.
// Loads: waited for explicitly (s_waitcnt vmcnt(0)) before the barrier that publishes their LDS image.  Stores: fire and
// forget (their data registers are read at issue).
__device__ __forceinline__ u32x4 make_rsrc(const void *base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;      // wave uniform: pin the descriptor to SGPRs
    return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu)),
                 (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
}
__device__ __forceinline__ void dma16_buf(const u32x4 rsrc, unsigned voff, unsigned lds_base) {   // LDS[lds_base + lane*16] <- buffer
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_base), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void dma16_glb(const void *g, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_base), "v"(g) : "memory");
}
__device__ __forceinline__ void store16_buf(const u32x4 v, const u32x4 rsrc, unsigned voff) {
    // s_nop: a VMEM store of more than 64 bits needs two wait states (gfx940+) before its data VGPRs may be overwritten (the
    // compiler pads this hazard for its own stores, not inside asm)
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void store4_buf(unsigned v, const u32x4 rsrc, unsigned voff) {
    asm volatile("buffer_store_dword %0, %1, %2, 0 offen" ::"v"(v), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p; }

#ifdef MFMA_TIMING
// dev builds only (tools/time_mfma.py): every wave adds the shader-clock ticks since its previous stamp to its row of a device-global table
__device__ unsigned long long g_mfma_dbg[16 * 16];
#define MF_STAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc_[i] += now_ - tprev_; tprev_ = now_; } while (0)
#else
#define MF_STAMP(i) do { } while (0)
#endif

template <int NB, int TY>      // NB: classifier row blocks of 16 classes (0: no head)
__global__ __launch_bounds__(64 * TY, 4) void creff_mfma_kernel(const CreffParams p) {
#ifdef MFMA_TIMING
    unsigned long long tacc_[15] = {};          // (wave uniform: scalar registers; flushed once per tile at the end of the kernel)
    unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
#endif
    constexpr int NBA = NB > 0 ? NB : 1;
    typedef Geo<TY> GE;
    constexpr int NT = GE::NT, HBUF = GE::HBUF, RH = GE::RH, HH = GE::HH, LH = GE::LH, KPLK = GE::KPLK, KPLV = GE::KPLV, HPL = GE::HPL, LPL = GE::LPL;
    extern __shared__ __attribute__((aligned(16))) f32x4 smem4[];
    f32x4 *Hs = smem4;                          // [2][G][HPL]   (double buffered, filled by LDS-DMA)
    f32x4 *Ls = Hs + HBUF * G * HPL;            // [G][LPL]
    f32x4 *Lw = Ls + G * LPL;                   // [2] raw lr window [G][px]
    f32x4 *Wd = Lw + 2 * LWCAP;                 // [2][3 convs][9 taps + bias][G]
    f32x4 *Wfs = Wd + 2 * 3 * 10 * G;           // [2][G][NBA*16]: classifier slice of a chunk, {4 hi | 4 lo} halves per entry
    f32x4 *Tb = Wfs + 2 * G * NBA * 16;         // bilinear tables: [LH] rows then [LWD] columns of the lr_up tile
    u32x4 *Kl = reinterpret_cast<u32x4 *>(Tb + LH + LWD);   // key / value records [G][RH*RW]: {4 hi halves | 4 lo halves}

    const int tid = threadIdx.x, tid_ = tid, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 15, g = lane >> 4, qy = q >> 3, qx = q & 7;
    const int pc = wave & 1, pr = wave >> 1;
    // XCD-aware tile order: consecutive workgroup ids go round-robin to the 8 XCDs (private L2s); give each XCD a
    // contiguous run of tiles so that the halos neighbouring tiles share are fetched into one L2 only
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
    const int CB = p.C >> 4;
    const int yq = 2 * pr + qy, xq = 8 * pc + qx;               // this lane's query pixel, tile relative
    const int gyq = ty0 + yq, gxq = tx0 + xq;

    // window of the low-resolution feature under the tile (+1 halo), block uniform (see creff.hip)
    int ly_lo, ly_n, lx_lo, lx_n;
    {
        int a0, a1, b0, b1; float l;
        arseg_src_index(p.sy, max(ty0 - 1, 0), true, p.hp, a0, a1, l);
        arseg_src_index(p.sy, min(ty0 + TY, p.Hp - 1), true, p.hp, b0, b1, l);
        ly_lo = a0; ly_n = b1 - a0 + 1;
        arseg_src_index(p.sx, max(tx0 - 1, 0), true, p.wp, a0, a1, l);
        arseg_src_index(p.sx, min(tx0 + TX, p.Wp - 1), true, p.wp, b0, b1, l);
        lx_lo = a0; lx_n = b1 - a0 + 1;
    }
    // bilinear(align_corners=True) taps of the tile rows / columns (tile coordinate -1 .. 16), once per tile:
    // {offset of tap 0 in the window, offset of tap 1, weight of tap 1, inside the image}
    if (tid < LH + LWD) {
        const bool row = tid < LH;
        const int rel = row ? tid : tid - LH;
        const int gc = (row ? ty0 : tx0) - 1 + rel, lim = row ? p.Hp : p.Wp;
        int i0, i1; float l1;
        arseg_src_index(row ? p.sy : p.sx, min(max(gc, 0), lim - 1), true, row ? p.hp : p.wp, i0, i1, l1);
        l1 = fminf(fmaxf(l1, 0.f), 1.f);
        const int o0 = row ? (i0 - ly_lo) * lx_n : i0 - lx_lo, o1 = row ? (i1 - ly_lo) * lx_n : i1 - lx_lo;
        Tb[tid] = f32x4{__int_as_float(o0), __int_as_float(o1), l1, (unsigned)gc < (unsigned)lim ? 1.0f : 0.0f};
    }
    // bilinear sample of the staged lr window at tile row r / column c (table indices), channel group gg
    auto lr_up = [&](const f32x4 *win, int r, int c, int gg) {
        const f32x4 ty = Tb[r], tx = Tb[LH + c];
        const f32x4 *b = win + gg * (ly_n * lx_n);      // one plane per channel group
        const int r0 = __float_as_int(ty[0]), r1 = __float_as_int(ty[1]), x0 = __float_as_int(tx[0]), x1 = __float_as_int(tx[1]);
        const f32x4 a = b[r0 + x0], bb = b[r0 + x1], cc = b[r1 + x0], d = b[r1 + x1];
        const float ly = ty[2], lx2 = tx[2];
        return (1.f - ly) * ((1.f - lx2) * a + lx2 * bb) + ly * ((1.f - lx2) * cc + lx2 * d);
    };

    // Per-thread work items of the staging / convolution rounds do not depend on the channel chunk: decode them once.
    // Hs layout (r5): [row][channel group][column] -- a staged row is G * HWD = 104 consecutive slots, i.e. two LDS-DMA instructions whose source
    // offsets are a lane constant plus scalars (issue_hr).  Key / value conv items: one thread = channel group, TWO vertically adjacent record rows,
    // one record column -- the 4 x 3 input window is read once for both outputs and so are the ten weight vectors (22 LDS reads per two outputs
    // instead of 38: the conv was bound by its LDS reads, profiles/r05_creff_mfma_phases_before.json)
    constexpr int HROW = G * HWD;
    static_assert(RH % 2 == 0, "record rows come in pairs");
    constexpr int K_TOT = G * (RH / 2) * RWC, K_NI = (K_TOT + NT - 1) / NT;
    constexpr int L_TOT = G * LH * LWD, L_NI = (L_TOT + NT - 1) / NT;
    constexpr unsigned BAD = 0x80000000u;             // beyond num_records even after the chunk offset is added
    int kcode[K_NI];          // one register per item: row | column << 8 | group << 16 | upper output inside the image << 20 | lower << 21
#pragma unroll
    for (int it = 0; it < K_NI; ++it) {
        const int i = min(tid + it * NT, K_TOT - 1);   // surplus lanes of the last round redo the last item
        const int gg = i / ((RH / 2) * RWC), rem = i - gg * ((RH / 2) * RWC), rp = rem / RWC, c = rem - rp * RWC, r = 2 * rp;
        const bool cin = (unsigned)(tx0 - 3 + c) < (unsigned)p.Wp;
        const int in0 = cin && (unsigned)(ty0 - 3 + r) < (unsigned)p.Hp, in1 = cin && (unsigned)(ty0 - 2 + r) < (unsigned)p.Hp;
        kcode[it] = r | (c << 8) | (gg << 16) | (in0 << 20) | (in1 << 21);
    }
    int lrc_[L_NI];                                    // lr_up tile items: row | col << 8 | group << 16, -1 = none
#pragma unroll
    for (int it = 0; it < L_NI; ++it) {
        const int i = tid + it * NT;                   // group slowest: consecutive lanes write consecutive LDS slots
        const int gg = i / (LH * LWD), px = i - gg * (LH * LWD), r = px / LWD, c = px - r * LWD;
        lrc_[it] = i < L_TOT ? (r | (c << 8) | (gg << 16)) : -1;
    }
    const u32x4 h_rsrc = make_rsrc(p.hr + (size_t)n * p.C * p.Hp * p.Wp, (unsigned)((size_t)p.C * p.Hp * p.Wp * sizeof(float)));
    const unsigned h_chunk = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(2 * p.Hp * p.Wp * 8) * 4u));      // bytes between 16-channel chunks (C8 layout); pinned to an SGPR (as a VGPR it was spilled and reloaded between the DMA requests)

    // Staging is asynchronous: the next chunk's hr region, lr window and depthwise weights go global -> LDS directly
    // (LDS-DMA, no staging registers) into the other half of double buffers while the current chunk is convolved and
    // multiplied.  With one 16-wave workgroup per CU nothing else would hide the memory latency.  Out-of-image hr elements
    // carry an out-of-range buffer offset and arrive as zeros (the conv's zero padding).
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int lw_tot = G * ly_n * lx_n;
    // exact for dividends below 2^16 (here < 2048): floor(2^32 / d) + 1
    const unsigned m_npx = (unsigned)__builtin_amdgcn_readfirstlane((int)(0xFFFFFFFFu / (unsigned)(ly_n * lx_n) + 1u));
    const unsigned m_lxn = (unsigned)__builtin_amdgcn_readfirstlane((int)(0xFFFFFFFFu / (unsigned)lx_n + 1u));
    f32x4 pf_w;
    // (measured and dropped in r5: the next chunk's requests spread over the iteration -- one behind barrier A, the rest between the VALU phases --
    // instead of a burst behind the barrier: the address arithmetic then lives across the convs, 100+ bytes of scratch, 72.7 -> 90.4 us per frame)
    auto issue_hr = [&](int k, int buf) {
        // the offsets are decoded again per chunk from an opaque thread id: kept in registers across the chunk loops they were spilled,
        // and every reload from scratch waited (vmcnt(0)) for the DMA issued just before it -- three serialised round trips per chunk
        // (r5) one instruction = 64 (or the last 40) consecutive slots of one staged row: slot s = (group, column) is a LANE constant, the row and
        // the chunk are scalars -- ~10 VALU per instruction where the per-item decode of rounds 1-4 took ~25 for each of three (the DMA issue was
        // 16 % of pass 1 and the pole wave's longest phase).  Lanes beyond the row's 104 slots are masked off (they would zero the next row).
        // (the lane id comes from mbcnt, two instructions: any copy of it that lives across the chunk loops ends up in scratch, and a scratch reload
        // between two DMA requests waits for the first one)
        int l_ = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); asm volatile("" : "+v"(l_));
        constexpr int PER_W = (HH * 2) / (NT / 64);          // instructions per wave: 3 (16-row tiles) / 4 (8-row tiles)
        static_assert(PER_W * (NT / 64) == HH * 2, "rows x 2 must divide over the waves");
#pragma unroll
        for (int j = 0; j < PER_W; ++j) {
            const int ins = wave_u * PER_W + j, r = ins >> 1, h = ins & 1;      // (scalar)
            const int sl = 64 * h + l_, gg = sl / HWD, c = sl - gg * HWD;
            const int gy = ty0 - 4 + r, gx = tx0 - 4 + c;
            const bool ok = (unsigned)gy < (unsigned)p.Hp && (unsigned)gx < (unsigned)p.Wp;
            const unsigned ho = ok ? (unsigned)((((gg >> 1) * p.Hp + gy) * p.Wp + gx) * 8 + (gg & 1) * 4) * 4u : BAD;
            if (sl < HROW) dma16_buf(h_rsrc, ho + (unsigned)__builtin_amdgcn_readfirstlane((int)(k * h_chunk)), lds_addr(Hs + (HBUF == 2 ? buf : 0) * G * HPL + r * HROW + 64 * h));
        }
    };
    auto issue = [&](int k, int buf, bool head) {
        if (HBUF == 2) issue_hr(k, buf);           // double buffered: nobody reads the other half now
        // the source addresses are recomputed per chunk from an opaque copy of the thread id: as loop invariants hipcc keeps three 64-bit
        // per-lane pointers live across both chunk loops, spills them and reloads them from scratch in every iteration (a vmcnt wait
        // that also waits for the LDS-DMA just issued)
        // (r5) the lr window is requested by the upper half of a 16-wave workgroup and the weights by waves 6 / 7: with everything on the lowest waves,
        // wave 0 issued twice the requests of the mean wave and every barrier waited for it (tools/time_mfma.py)
        constexpr int LW0 = NT == 1024 ? 512 : 0, WD0 = NT == 1024 ? 384 : 0;
        int tid = tid_ - LW0; asm volatile("" : "+v"(tid));
        const int wave_l = wave_u - LW0 / 64;
        if (wave_l >= 0 && wave_l * 64 < lw_tot) {
            const int i = min(tid, lw_tot - 1);          // surplus lanes of the last wave repeat the last item (stay inside Lw)
            // divisions by the (uniform) window extents as multiply-high with SGPR constants: hipcc's own expansion hoists its VGPR reciprocals
            // out of the chunk loops, where they were spilled and reloaded behind the DMA requests (r5)
            const int npx = ly_n * lx_n, gg = (int)__umulhi((unsigned)i, m_npx), px = i - gg * npx, r = (int)__umulhi((unsigned)px, m_lxn), c = px - r * lx_n;
            const float *src = p.lr + ((size_t)n * p.hp * p.wp + (size_t)(ly_lo + r) * p.wp + lx_lo + c) * p.C + k * 16 + gg * 4;
            if (tid < lw_tot) dma16_glb(src, lds_addr(Lw + buf * LWCAP + wave_l * 64));
        }
        tid += LW0 - WD0;
        const int wave_w = wave_u - WD0 / 64;
        if (wave_w >= 0 && wave_w * 64 < 3 * 10 * G) {                  // depthwise weights [9][C] + biases of the chunk
            const int t = min(tid, 3 * 10 * G - 1);
            const int gg = t & 3, tp = (t >> 2) % 10, cv = t / (G * 10);
            const float *w = cv == 0 ? p.wq : (cv == 1 ? p.wk : p.wv);
            const float *bb = cv == 0 ? p.bq : (cv == 1 ? p.bk : p.bv);
            const int c = k * 16 + gg * 4;
            const float *src = tp < 9 ? w + (size_t)tp * p.C + c : bb + c;
            if (tid < 3 * 10 * G) dma16_glb(src, lds_addr(Wd + buf * 3 * 10 * G + wave_w * 64));
        }
        (void)head;
    };
    // classifier slice [class][16 ch] of chunk k, entry = (group, class): requested right in front of a chunk's MFMA loop and committed behind it (r5: as
    // part of issue() the four registers were live through the whole iteration, across the convs that need every register they can get)
    auto load_head = [&](int k) {
        int tid = tid_; asm volatile("" : "+v"(tid));          // (opaque: the class / group decode and the pointer are not worth a spilled register pair)
        if (NB > 0 && tid < G * NBA * 16) {
            const int cls = tid % (NBA * 16), gg = tid / (NBA * 16);
            pf_w = f32x4{0.f, 0.f, 0.f, 0.f};
            if (cls < p.n_cls) pf_w = *reinterpret_cast<const f32x4 *>(p.wf + (size_t)cls * p.C + k * 16 + gg * 4);
        }
    };
    auto commit_head = [&](int buf) {
        int tid = tid_; asm volatile("" : "+v"(tid));
        if (NB > 0 && tid < G * NBA * 16) {
            u32x2 hi, lo;
            split4(pf_w, hi, lo);
            Wfs[buf * G * NBA * 16 + tid] = __builtin_bit_cast(f32x4, u32x4{hi.x, hi.y, lo.x, lo.y});
        }
    };
    // key (cv=1) or value (cv=2) records of the region: bias + dw3x3(Hs), zero outside the image, split to fp16 hi/lo
    auto conv_kv = [&](int cv, int buf) {
        const f32x4 *w = Wd + (buf * 3 + cv) * 10 * G, *hs = Hs + (HBUF == 2 ? buf : 0) * G * HPL;
#pragma unroll
        for (int it = 0; it < K_NI; ++it) {
            const int code = kcode[it], kr = code & 255, kc = (code >> 8) & 255, kgg = (code >> 16) & 15;
            const f32x4 *h = hs + (kr * G + kgg) * HWD + kc;
            const f32x4 *wg = w + kgg;
            f32x4 a0 = wg[9 * G], a1 = a0;
            f32x4 u0 = h[0], u1 = h[1], u2 = h[2];          // window row t (upper output's tap row t)
#pragma unroll
            for (int t = 0; t < 3; ++t) {            // tap row t: weights read once, applied to window row t (upper) and t + 1 (lower output)
                const f32x4 v0 = h[(t + 1) * HROW], v1 = h[(t + 1) * HROW + 1], v2 = h[(t + 1) * HROW + 2];
                { const f32x4 w0 = wg[(t * 3) * G]; a0 += w0 * u0; a1 += w0 * v0; }
                __builtin_amdgcn_sched_barrier(0);
                { const f32x4 w1 = wg[(t * 3 + 1) * G]; a0 += w1 * u1; a1 += w1 * v1; }
                __builtin_amdgcn_sched_barrier(0);
                { const f32x4 w2 = wg[(t * 3 + 2) * G]; a0 += w2 * u2; a1 += w2 * v2; }
                u0 = v0; u1 = v1; u2 = v2;
                __builtin_amdgcn_sched_barrier(0);      // keep the live set at two window rows + one weight row (all twelve loads hoisted: spills)
            }
            if (!(code & (1 << 20))) a0 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!(code & (1 << 21))) a1 = f32x4{0.f, 0.f, 0.f, 0.f};
            u32x2 hi, lo;
            u32x4 *dst = Kl + kr * RW + kc + kgg * (cv == 1 ? KPLK : KPLV);
            split4(a0, hi, lo);
            dst[0] = u32x4{hi.x, hi.y, lo.x, lo.y};
            split4(a1, hi, lo);
            dst[RW] = u32x4{hi.x, hi.y, lo.x, lo.y};
        }
    };

    // the two record columns that only pad the windows to 16 slots are never computed: zero them once
    if (tid < G * RH * 2) {
        const int gg = tid / (RH * 2), e = ((tid % (RH * 2)) >> 1) * RW + RWC + (tid & 1);
        Kl[gg * KPLK + e] = u32x4{0, 0, 0, 0};
        Kl[gg * KPLV + e] = u32x4{0, 0, 0, 0};       // (the two layouts overlap; both sets of pad columns stay zero)
    }

    const u32x4 *ka = Kl + g * KPLK + (2 * pr) * RW + 8 * pc + q;             // A operand of the scores, row block 0
    // transpose read: lane i of a 16-lane group supplies the 8-byte piece (pixel 4g + i/4, channel group i%4) and receives
    // channel i of the group's 4 pixels (verified on gfx950: out[i][j] = halfword i%4 of the piece of lane 4j + i/4)

    f32x4 S[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) S[b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ------------------------------------------------------------------ pass 1: scores
    issue(0, 0, false);
    if (HBUF == 1) issue_hr(0, 0);
    for (int k = 0; k < CB; ++k) {
        const int buf = k & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's share of chunk k has landed
        MF_STAMP(0);
        __syncthreads();                         // ... everybody's has; everybody is done with the other buffers
        MF_STAMP(1);
        issue(k + 1 < CB ? k + 1 : 0, buf ^ 1, k + 1 == CB);      // after the last chunk: chunk 0 again, for pass 2
        MF_STAMP(2);
#pragma unroll
        for (int it = 0; it < L_NI; ++it) {
            const int code = lrc_[it];
            if (code >= 0) {
                const int r = code & 255, c = (code >> 8) & 255, gg = code >> 16;
                f32x4 v = lr_up(Lw + buf * LWCAP, r, c, gg);
                if (Tb[r][3] * Tb[LH + c][3] == 0.f) v = f32x4{0.f, 0.f, 0.f, 0.f};      // conv zero padding outside the image
                Ls[gg * LPL + r * LWD + c] = v;
            }
        }
        MF_STAMP(3);
        conv_kv(1, buf);
        MF_STAMP(4);
        __syncthreads();
        MF_STAMP(5);
        if (HBUF == 1) issue_hr(k + 1 < CB ? k + 1 : 0, 0);      // single hr buffer: free now that the key records are built
        // query conv, lane local: channels 4g..4g+3 of this lane's own pixel
        const f32x4 *w = Wd + (buf * 3 + 0) * 10 * G;
        f32x4 qv = w[9 * G + g];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) qv += w[(dy * 3 + dx) * G + g] * Ls[g * LPL + (yq + dy) * LWD + xq + dx];
        u32x2 qh, ql;
        split4(qv, qh, ql);
        const u32x6 q6 = pack6(qh, ql);
        MF_STAMP(6);
        if (k + 1 == CB) load_head(0);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const h16x8 a = __builtin_bit_cast(h16x8, ka[b * RW]);
            S[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, op_a(q6), S[b], 0, 0, 0);
            S[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, op_b(q6), S[b], 0, 0, 0);
        }
        if (k + 1 == CB) commit_head(buf ^ 1);
        MF_STAMP(7);
    }

    // ------------------------------------------------------------------ softmax over the 49 taps (padding taps included)
    // S[b][i] = score(query q, key row b, key column 4g+i); tap (b - qy, 4g+i - qx) is real iff both are in [0,6]
    float inv;
    u32x4 P4[8];          // {hi | lo}: the swapped operand {lo | hi} is rebuilt per use (4 moves) -- as 6-register {hi, lo, hi} aliases the 8
                          // blocks cost 48 VGPRs and the kernel spilled into its chunk loops (scratch reloads wait on vmcnt with the DMA)
    {
        bool colok[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) colok[i] = (unsigned)(4 * g + i - qx) <= 6u;
        float m = -INFINITY;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool rowok = (unsigned)(b - qy) <= 6u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                S[b][i] = (rowok && colok[i]) ? S[b][i] : -INFINITY;
                m = fmaxf(m, S[b][i]);
            }
        }
        m = rows_max(m);
        const float ml = m * LOG2E;
        float z = 0.f;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                S[b][i] = __builtin_amdgcn_exp2f(fmaf(S[b][i], LOG2E, -ml));     // masked slots: exp2(-inf) = 0
                z += S[b][i];
            }
            u32x2 hi, lo;
            split4(S[b], hi, lo);
            P4[b] = u32x4{hi.x, hi.y, lo.x, lo.y};
        }
        z = rows_sum(z);
        inv = 1.0f / z;                            // applied to the weighted sum instead of the 128 weights
    }

    MF_STAMP(8);
    f32x4 lg[NBA];
#pragma unroll
    for (int nb = 0; nb < NBA; ++nb) lg[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool inq = gyq < p.Hp && gxq < p.Wp;
    const u32x4 p_rsrc = make_rsrc(p.p_out, p.p_bytes), l_rsrc = make_rsrc(p.logits, p.l_bytes);
    // p (C8 layout) offset of this lane's query in chunk k: ((((n C/8 + 2k + g/2) Hp + gyq) Wp + gxq) 8 + (g&1) 4) floats
    const unsigned p_kstep = 2u * (unsigned)p.Hp * (unsigned)p.Wp * 32u;

    // ------------------------------------------------------------------ pass 2: weighted values, residual, head
    for (int k = 0; k < CB; ++k) {
        const int buf = (k + CB) & 1;            // continues the alternation of pass 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MF_STAMP(9);
        __syncthreads();
        MF_STAMP(10);
        if (k + 1 < CB) issue(k + 1, buf ^ 1, true);
        conv_kv(2, buf);
        int t2 = tid_; asm volatile("" : "+v"(t2));
        const int q2 = t2 & 15, g2 = (t2 >> 4) & 3, w2 = t2 >> 6, pc2 = w2 & 1, pr2 = w2 >> 1;
        const f32x4 lrc = lr_up(Lw + buf * LWCAP, 2 * pr2 + (q2 >> 3) + 1, 8 * pc2 + (q2 & 7) + 1, g2);      // residual term, channels 4g..4g+3 (table rows clamp into the image)
        MF_STAMP(11);
        __syncthreads();
        MF_STAMP(12);
        if (HBUF == 1 && k + 1 < CB) issue_hr(k + 1, 0);
        if (k + 1 < CB) load_head(k + 1);
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        // (r5) the transpose-read pointer and the store offset are rebuilt per chunk from an opaque thread id: as loop invariants they were hoisted,
        // spilled, and their reloads sat behind the DMA requests of the next chunk
        const unsigned char *va = reinterpret_cast<const unsigned char *>(Kl + (q2 & 3) * KPLV + (2 * pr2) * RW + 8 * pc2 + 4 * g2 + (q2 >> 2));
        const int gy2 = ty0 + 2 * pr2 + (q2 >> 3), gx2 = tx0 + 8 * pc2 + (q2 & 7);
        const unsigned p_off0 = (((((unsigned)n * (unsigned)(p.C >> 3) + (unsigned)(g2 >> 1)) * p.Hp + gy2) * p.Wp + gx2) * 8u + (g2 & 1) * 4u) * 4u;
        const bool inq = gy2 < p.Hp && gx2 < p.Wp;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const h16x8 a = pack8(lds_tr16(va + b * RW * 16), lds_tr16(va + b * RW * 16 + 8));
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h16x8, P4[b]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h16x8, u32x4{P4[b].z, P4[b].w, P4[b].x, P4[b].y}), acc, 0, 0, 0);
        }
        const f32x4 o = lrc + acc * inv;              // p[query][16k + 4g .. +3]
        const unsigned off = p_off0 + (unsigned)k * p_kstep;       // 32-bit: the 64-bit form kept three hoisted partial products in spilled registers
        store16_buf(__builtin_bit_cast(u32x4, o), p_rsrc, inq ? off : OOB);
        if (NB > 0) {
            u32x2 oh, ol;
            split4(o, oh, ol);
            const u32x6 o6 = pack6(oh, ol);
#pragma unroll
            for (int nb = 0; nb < NBA; ++nb) {
                const h16x8 wa = __builtin_bit_cast(h16x8, Wfs[(buf * G + g2) * NBA * 16 + nb * 16 + q2]);
                lg[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, op_a(o6), lg[nb], 0, 0, 0);
                lg[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, op_b(o6), lg[nb], 0, 0, 0);
            }
        }
        if (k + 1 < CB) commit_head(buf ^ 1);
        MF_STAMP(13);
    }

    // ------------------------------------------------------------------ logits: lg[nb][i] = class 16nb + 4g + i of query q
    if (NB > 0) {
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
            m = rows_max(m);
            float z = 0.f;
#pragma unroll
            for (int nb = 0; nb < NBA; ++nb)
#pragma unroll
                for (int i = 0; i < 4; ++i) z += nb * 16 + 4 * g + i < p.n_cls ? expf(lg[nb][i] - m) : 0.f;
            z = rows_sum(z);
            const float lse = m + logf(z);
#pragma unroll
            for (int nb = 0; nb < NBA; ++nb) lg[nb] -= lse;
        }
#pragma unroll
        for (int nb = 0; nb < NBA; ++nb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cls = nb * 16 + 4 * g + i;
                const unsigned off = (unsigned)(((((size_t)n * p.n_cls + cls) * p.Hp + gyq) * p.Wp + gxq) * sizeof(float));
                store4_buf(__float_as_uint(lg[nb][i]), l_rsrc, (inq && cls < p.n_cls) ? off : OOB);
            }
    }
    MF_STAMP(14);
#ifdef MFMA_TIMING
    if ((threadIdx.x & 63) == 0)
        for (int i = 0; i < 15; ++i) atomicAdd(&g_mfma_dbg[16 * (threadIdx.x >> 6) + i], tacc_[i]);
    if (threadIdx.x == 0) atomicAdd(&g_mfma_dbg[15], 1ull);
#endif
}

template <int NB, int TY>
int launch(const CreffParams &p, hipStream_t st) {
    constexpr int NBA = NB > 0 ? NB : 1;
    typedef Geo<TY> GE;
    const size_t smem = ((size_t)GE::HBUF * G * GE::HPL + (size_t)G * GE::LPL + 2 * LWCAP + 2 * 3 * 10 * G + 2 * G * NBA * 16) * sizeof(f32x4) +
                        (GE::LH + LWD) * sizeof(f32x4) + (size_t)G * GE::KPLV * sizeof(u32x4);
    static ArsegSmemAttr attr;
    if (int e = arseg_allow_smem(attr, reinterpret_cast<const void *>(creff_mfma_kernel<NB, TY>), smem)) return e;
    dim3 grid(arseg_cdiv(p.Wp, TX), arseg_cdiv(p.Hp, TY), p.N);
    hipLaunchKernelGGL((creff_mfma_kernel<NB, TY>), grid, dim3(GE::NT), smem, st, p);
    return arseg_launch_status();
}

}  // namespace

#ifdef MFMA_TIMING
extern "C" void arseg__mfma_dbg_read(unsigned long long *host, int reset) {
    (void)hipDeviceSynchronize();
    if (host) (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_mfma_dbg), sizeof(unsigned long long) * 256);
    if (reset) { static unsigned long long z[256]; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mfma_dbg), z, sizeof(z)); }
}
#endif

int arseg_creff_mfma_launch(const CreffParams &p, hipStream_t st) {
    if (p.C & 15) return ARSEG_EUNSUPPORTED;
    //  the lr-window decode divides by multiply-high reciprocals floor(2^32 / d) + 1, which wrap to 0 for d == 1: an LR feature one
    // pixel wide or high (a 1-wide window) runs on the VALU kernel instead (the dispatcher's fallback for EUNSUPPORTED)
    if (p.hp < 2 || p.wp < 2) return ARSEG_EUNSUPPORTED;
    // the raw lr window under a tile (+1 halo, +1 for the second bilinear tap) must fit its LDS slot
    const int wy = (int)((16 + 1) * p.sy) + 3, wx = (int)((TX + 1) * p.sx) + 3;
    if (G * wy * wx > LWCAP) return ARSEG_EUNSUPPORTED;
    if (p.n_cls > 32) return ARSEG_EUNSUPPORTED;
    // 16-row tiles (one 16-wave workgroup per CU) are faster per frame once the launch fills the chip (batched frames:
    // 0.089 vs 0.104 ms per BiSeNet frame); small launches -- a single 128x256 map is 128 such tiles -- do better with
    // 8-row tiles, twice the workgroups, two per CU (109 vs 151 us).  p.mfma_tile_rows = 8 | 16 pins one (tests, measurements).
    const int ty_env = p.mfma_tile_rows;
    const long long tiles16 = (long long)arseg_cdiv(p.Wp, TX) * arseg_cdiv(p.Hp, 16) * p.N;
    const bool ty8 = ty_env == 8 || (ty_env != 16 && tiles16 < 512);
    if (p.n_cls == 0) return ty8 ? launch<0, 8>(p, st) : launch<0, 16>(p, st);
    if (p.n_cls <= 16) return ty8 ? launch<1, 8>(p, st) : launch<1, 16>(p, st);
    return ty8 ? launch<2, 8>(p, st) : launch<2, 16>(p, st);
}
