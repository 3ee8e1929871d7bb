// CReFF -- cross-resolution feature fusion -- as ONE kernel.
//
// Reference: MyAttention.forward (model/attention.py:184-213) followed by the frozen 1x1 classifier
// (model/pspnet.py:225-229, model/bisenet.py:571-572).  The reference runs seven separate ops
// (interpolate, three depthwise convs, localAttention.similar_forward, softmax,
// localAttention.weighting_forward, add) with the [P,49] score tensor and four [C,P] intermediates
// round-tripping HBM.  Here one workgroup owns a TH x 32 pixel tile and keeps everything on chip.
//
// Data layout: the (already warped) keyframe feature `hr` and the fused output `p` use the
// channel-blocked layout C8 = [N][C/8][H][W][8]: the kernel walks the channels in chunks of 8, and
// in C8 a chunk of a tile row is one contiguous run (32 B per pixel), so every staged byte of a
// fetched line is used.  (Fetching 32-byte pieces out of 256-byte NHWC pixels would waste 3/4 of
// each 128-byte line per chunk.)  `lr` is small (1/4 of the pixels) and is gathered from NHWC.
//
// Per 8-channel chunk, pass 1 (scores):
//   Hs <- hr tile + (R+1) halo           (LDS, zero outside the image = conv zero padding)
//   Ls <- bilinear(align_corners=True) upsample of lr on tile + 1 halo      (LDS)
//   Ks <- bias + depthwise3x3(Hs) on tile + R halo, ZERO outside the image  (the unfold's padding:
//         padded taps score 0 and still take part in the softmax -- model/attention.py:56-58,203)
//   q  <- bias + depthwise3x3(Ls) at the thread's two pixels                (registers)
//   S[2][KS*KS] += q . Ks window                                            (registers)
// softmax over the KS*KS taps in registers, then pass 2 per chunk:
//   Vs <- value conv of the re-staged hr chunk, A = sum_i W_i * V_i, p = lr_up + A -> global (C8),
//   logits[k] += wf[k][chunk] . p                                            (registers)
// and finally (optional) log-softmax over classes and an NCHW store of the logits.
//
// Thread <-> work: lanes run along x (32 per tile row), each thread owns two vertically adjacent
// pixels so that every K/V vector read from LDS feeds up to two queries (7 FMA4 per 8 reads instead
// of 4): the loop is VALU-bound, not LDS-bound.  LDS tiles are [channel group of 4][row][col] float4,
// so a wave's ds_read_b128 covers 32 consecutive 16-byte slots = conflict free.
// fp32 throughout (VALU); the banded QK^T wastes >75% of an fp32 MFMA tile, which runs at the VALU
// rate anyway (MI355X_MICROARCH: v_mfma_f32_* = 64 FLOP/clk/SIMD), so MFMA would be slower here.
#include "creff_params.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Output stores are issued through inline asm: on gfx9 hipcc drains every counter it knows about in front of each s_barrier,
// so a builtin store exposes the full write latency at the next barrier of the chunk loop.
__device__ __forceinline__ u32x4 make_rsrc(const void *base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu)),
                 (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
}
__device__ __forceinline__ void store16_buf(const u32x4 v, const u32x4 rsrc, unsigned voff) {
    // s_nop: a VMEM store of more than 64 bits needs two wait states (gfx940+) before its data VGPRs may be overwritten
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void store4_buf(unsigned v, const u32x4 rsrc, unsigned voff) {
    asm volatile("buffer_store_dword %0, %1, %2, 0 offen" ::"v"(v), "v"(voff), "s"(rsrc) : "memory");
}

constexpr int TW = 32;
constexpr int G = 2;       // float4 groups per 8-channel chunk
constexpr int PAD = 4;     // float4 pad between the two group planes (bank spread for paired writes)

__device__ __forceinline__ float dot4(const f32x4 a, const f32x4 b, float acc) {
    acc = fmaf(a[0], b[0], acc); acc = fmaf(a[1], b[1], acc); acc = fmaf(a[2], b[2], acc); acc = fmaf(a[3], b[3], acc);
    return acc;
}

// acc + w * v with plain scalar FMAs: written as `w * v` on the vector type, hipcc materialises a
// {w,w} register pair per weight for v_pk_fma_f32 and hoists all of them out of the chunk loop
// (2 x 98 extra VGPRs -> spills).
__device__ __forceinline__ f32x4 axpy4(float w, const f32x4 v, f32x4 acc) {
    acc[0] = fmaf(w, v[0], acc[0]); acc[1] = fmaf(w, v[1], acc[1]); acc[2] = fmaf(w, v[2], acc[2]); acc[3] = fmaf(w, v[3], acc[3]);
    return acc;
}

// bilinear(align_corners=True) sample of the NHWC lr feature at HR pixel (gy,gx), channels [c, c+4)
__device__ __forceinline__ f32x4 lr_up_at(const CreffParams &p, int n, int gy, int gx, int c) {
    int y0, y1, x0, x1; float ly, lx;
    arseg_src_index(p.sy, gy, true, p.hp, y0, y1, ly);
    arseg_src_index(p.sx, gx, true, p.wp, x0, x1, lx);
    ly = fminf(fmaxf(ly, 0.f), 1.f); lx = fminf(fmaxf(lx, 0.f), 1.f);
    const float *b = p.lr + (size_t)n * p.hp * p.wp * p.C + c;
    const f32x4 a = *reinterpret_cast<const f32x4 *>(b + ((size_t)y0 * p.wp + x0) * p.C);
    const f32x4 bb = *reinterpret_cast<const f32x4 *>(b + ((size_t)y0 * p.wp + x1) * p.C);
    const f32x4 cc = *reinterpret_cast<const f32x4 *>(b + ((size_t)y1 * p.wp + x0) * p.C);
    const f32x4 d = *reinterpret_cast<const f32x4 *>(b + ((size_t)y1 * p.wp + x1) * p.C);
    return (1.f - ly) * ((1.f - lx) * a + lx * bb) + ly * ((1.f - lx) * cc + lx * d);
}

template <int KS, int NC, int TH>
__global__ __launch_bounds__(16 * TH, 2) void creff_kernel(const CreffParams p) {
    constexpr int R = KS / 2, T = KS * KS, NT = 16 * TH;
    constexpr int HH = TH + 2 * R + 2, HWD = TW + 2 * R + 2, HPL = HH * HWD + PAD;   // hr tile (+R+1 halo)
    constexpr int KH = TH + 2 * R, KWD = TW + 2 * R, KPL = KH * KWD + PAD;           // K / V tile (+R halo)
    constexpr int LH = TH + 2, LWD = TW + 2, LPL = LH * LWD + PAD;                   // lr_up tile (+1 halo)
    extern __shared__ __attribute__((aligned(16))) f32x4 smem4[];
    f32x4 *Hs = smem4, *Ks = Hs + G * HPL, *Ls = Ks + G * KPL, *Wd = Ls + G * LPL;   // Wd: [3][10][G] (9 taps + bias)
    f32x4 *Wf = Wd + 3 * 10 * G;                                                  // classifier [n_cls][C/4], staged once

    const int tid = threadIdx.x, lx = tid & 31, yp = tid >> 5;
    const int n = blockIdx.z, ty0 = blockIdx.y * TH, tx0 = blockIdx.x * TW;
    const int CB = p.C >> 3;
    const int px = tx0 + lx, py0 = ty0 + 2 * yp;

    // Window of the low-resolution feature this tile's upsampled pixels touch (align_corners=True source rows/cols of
    // the tile + 1 halo), block-uniform and the same for every chunk.  Per chunk it is staged into LDS with a handful of
    // contiguous loads and the bilinear taps are read from there; gathering the taps straight from global memory costs
    // 16-20 scattered 16-byte loads per thread and chunk (a quarter of the kernel's time).
    int ly_lo, ly_n, lx_lo, lx_n;
    {
        int a0, a1, b0, b1; float l;
        arseg_src_index(p.sy, max(ty0 - 1, 0), true, p.hp, a0, a1, l);
        arseg_src_index(p.sy, min(ty0 + TH, p.Hp - 1), true, p.hp, b0, b1, l);
        ly_lo = a0; ly_n = b1 - a0 + 1;
        arseg_src_index(p.sx, max(tx0 - 1, 0), true, p.wp, a0, a1, l);
        arseg_src_index(p.sx, min(tx0 + TW, p.Wp - 1), true, p.wp, b0, b1, l);
        lx_lo = a0; lx_n = b1 - a0 + 1;
    }
    // pass 1 parks the window in the (not yet written) K tile, pass 2 in the (no longer used) lr_up tile
    const bool lr_lds = G * ly_n * lx_n <= (G * LPL < G * KPL ? G * LPL : G * KPL);
    auto stage_lr = [&](f32x4 *dst, int cb, int tid) {
        const float *src = p.lr + (size_t)n * p.hp * p.wp * p.C + cb * 8;
        for (int i = tid; i < G * ly_n * lx_n; i += NT) {
            const int g = i & 1, pc = i >> 1, r = pc / lx_n, c = pc - r * lx_n;
            dst[i] = *reinterpret_cast<const f32x4 *>(src + ((size_t)(ly_lo + r) * p.wp + lx_lo + c) * p.C + g * 4);
        }
    };
    // bilinear(align_corners=True) sample of the staged window at HR pixel (gy,gx), channel group g
    auto lr_up_lds = [&](const f32x4 *win, int gy, int gx, int g) {
        int y0, y1, x0, x1; float ly, lx2;
        arseg_src_index(p.sy, gy, true, p.hp, y0, y1, ly);
        arseg_src_index(p.sx, gx, true, p.wp, x0, x1, lx2);
        ly = fminf(fmaxf(ly, 0.f), 1.f); lx2 = fminf(fmaxf(lx2, 0.f), 1.f);
        const f32x4 *b = win + g;
        const int r0 = (y0 - ly_lo) * lx_n - lx_lo, r1 = (y1 - ly_lo) * lx_n - lx_lo;
        const f32x4 a = b[(r0 + x0) * 2], bb = b[(r0 + x1) * 2], cc = b[(r1 + x0) * 2], d = b[(r1 + x1) * 2];
        return (1.f - ly) * ((1.f - lx2) * a + lx2 * bb) + ly * ((1.f - lx2) * cc + lx2 * d);
    };

    // The staging loops have compile-time trip counts and are fully unrolled so that every global load of a
    // phase is in flight at once (with a `tid`-bounded loop hipcc issues them one at a time and the kernel
    // becomes latency bound at 2 waves per SIMD).
    auto stage_hr = [&](int cb, int tid) {
        const float *src = p.hr + ((size_t)n * CB + cb) * p.Hp * p.Wp * 8;
        constexpr int TOT = G * HH * HWD, NI = (TOT + NT - 1) / NT;
        f32x4 v[NI];
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int i = tid + it * NT;
            const int g = i & 1, pc = i >> 1, r = pc / HWD, c = pc - r * HWD;
            const int gy = ty0 - (R + 1) + r, gx = tx0 - (R + 1) + c;
            v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (i < TOT && (unsigned)gy < (unsigned)p.Hp && (unsigned)gx < (unsigned)p.Wp)
                v[it] = *reinterpret_cast<const f32x4 *>(src + ((size_t)gy * p.Wp + gx) * 8 + g * 4);
        }
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int i = tid + it * NT;
            if (i < TOT) Hs[(i & 1) * HPL + (i >> 1)] = v[it];
        }
    };
    auto stage_dw = [&](int cb, int tid) {   // depthwise weights + biases of this chunk: Wd[conv][tap(9)=bias][g]
        for (int i = tid; i < 3 * 10 * G; i += NT) {
            const int g = i % G, t = (i / G) % 10, cv = i / (G * 10);
            const float *w = cv == 0 ? p.wq : (cv == 1 ? p.wk : p.wv);
            const float *b = cv == 0 ? p.bq : (cv == 1 ? p.bk : p.bv);
            const int c = cb * 8 + g * 4;
            Wd[i] = t < 9 ? *reinterpret_cast<const f32x4 *>(w + (size_t)t * p.C + c) : *reinterpret_cast<const f32x4 *>(b + c);
        }
    };
    // K or V tile = bias + dw3x3(Hs), zero outside the image (select, no branch: the Hs halo is zero filled, so the reads
    // are always in range).  A thread owns one (column, channel group) and walks down a segment of rows with a 3-row
    // sliding window in registers: 3 LDS reads per output instead of 9, the 9 taps + bias are read once per segment
    // (the LDS pipeline, not the VALU, is this kernel's busiest unit).
    auto conv_tile = [&](int cv, int tid) {
        constexpr int LANES = G * KWD;                       // (column, group) pairs of one tile row
        if constexpr (NT >= LANES) {
            constexpr int NSEG = NT / LANES, ROWS = (KH + NSEG - 1) / NSEG;
            const int seg = tid / LANES, rem = tid - seg * LANES, g = rem / KWD, c = rem - g * KWD;
            if (seg < NSEG) {
                f32x4 wt[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) wt[k] = Wd[(cv * 10 + k) * G + g];
                const f32x4 bias = Wd[(cv * 10 + 9) * G + g];
                const int rb = seg * ROWS;
                const f32x4 *hp = Hs + g * HPL + rb * HWD + c;
                const int gx = tx0 - R + c;
                const bool colok = (unsigned)gx < (unsigned)p.Wp;
                f32x4 h[3][3];
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) h[dy][dx] = hp[dy * HWD + dx];
#pragma unroll
                for (int i = 0; i < ROWS; ++i) {
                    const int r = rb + i;
                    if (r < KH) {                             // (uniform per segment except in the last one)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) h[(i + 2) % 3][dx] = hp[(i + 2) * HWD + dx];
                        f32x4 acc = bias;
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                            for (int dx = 0; dx < 3; ++dx) acc += wt[dy * 3 + dx] * h[(i + dy) % 3][dx];
                        const int gy = ty0 - R + r;
                        if (!(colok && (unsigned)gy < (unsigned)p.Hp)) acc = f32x4{0.f, 0.f, 0.f, 0.f};
                        Ks[g * KPL + r * KWD + c] = acc;
                    }
                }
            }
        } else {
            constexpr int TOT = G * KH * KWD, NI = (TOT + NT - 1) / NT;
#pragma unroll 2
            for (int it = 0; it < NI; ++it) {
                const int i = min(tid + it * NT, TOT - 1);          // the surplus lanes of the last round redo the last item
                const int g = i / (KH * KWD), pc = i - g * (KH * KWD), r = pc / KWD, c = pc - r * KWD;
                const int gy = ty0 - R + r, gx = tx0 - R + c;
                f32x4 acc = Wd[(cv * 10 + 9) * G + g];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) acc += Wd[(cv * 10 + dy * 3 + dx) * G + g] * Hs[g * HPL + (r + dy) * HWD + c + dx];
                if (!((unsigned)gy < (unsigned)p.Hp && (unsigned)gx < (unsigned)p.Wp)) acc = f32x4{0.f, 0.f, 0.f, 0.f};
                Ks[g * KPL + pc] = acc;
            }
        }
    };

    if (NC > 0)
        for (int i = tid; i < p.n_cls * (p.C >> 2); i += NT) Wf[i] = *reinterpret_cast<const f32x4 *>(p.wf + (size_t)i * 4);

    float S0[T], S1[T];
#pragma unroll
    for (int i = 0; i < T; ++i) { S0[i] = 0.f; S1[i] = 0.f; }

    // ------------------------------------------------------------------ pass 1: scores
    // Barriers: Hs/Ls/Wd are last read before the second barrier of a chunk and rewritten at the top of
    // the next one; Ks is rewritten only after the next chunk's first barrier, which every thread
    // reaches after finishing its score loop -- two barriers per chunk are enough.
    for (int cb = 0; cb < CB; ++cb) {
        // The staging index math does not depend on cb; left alone, LLVM hoists all of it out of the chunk
        // loop (hundreds of live values -> spills).  An opaque copy of the thread id per iteration keeps
        // it inside the loop where its registers die immediately.
        int t = tid;
        asm volatile("" : "+v"(t));
        stage_hr(cb, t);
        stage_dw(cb, t);
        if (lr_lds) {
            __syncthreads();             // the previous chunk's score walk is done with the K tile
            stage_lr(Ks, cb, t);
            __syncthreads();
        }
        {
            constexpr int TOT = G * LH * LWD, NI = (TOT + NT - 1) / NT;
            f32x4 v[NI];
#pragma unroll
            for (int it = 0; it < NI; ++it) {
                const int i = t + it * NT;
                const int g = i & 1, pc = i >> 1, r = pc / LWD, c = pc - r * LWD;
                const int gy = ty0 - 1 + r, gx = tx0 - 1 + c;
                const bool ok = i < TOT && (unsigned)gy < (unsigned)p.Hp && (unsigned)gx < (unsigned)p.Wp;
                // clamped coordinates keep the taps in range; lanes outside the image are zeroed afterwards
                const int cy = min(max(gy, 0), p.Hp - 1), cx = min(max(gx, 0), p.Wp - 1);
                v[it] = lr_lds ? lr_up_lds(Ks, cy, cx, g) : lr_up_at(p, n, cy, cx, cb * 8 + g * 4);
                if (!ok) v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int it = 0; it < NI; ++it) {
                const int i = t + it * NT;
                if (i < TOT) Ls[(i & 1) * LPL + (i >> 1)] = v[it];
            }
        }
        __syncthreads();
        conv_tile(1, t);
        f32x4 q[2][G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            q[0][g] = Wd[(0 * 10 + 9) * G + g];
            q[1][g] = q[0][g];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const f32x4 v = Ls[g * LPL + (2 * yp + r) * LWD + lx + dx];
                    if (r < 3) q[0][g] += Wd[(0 * 10 + r * 3 + dx) * G + g] * v;
                    if (r >= 1) q[1][g] += Wd[(0 * 10 + (r - 1) * 3 + dx) * G + g] * v;
                }
        }
        __syncthreads();
        // Row-pipelined window walk.  hipcc otherwise hoists all (KS+1)*KS*G LDS reads of the unrolled
        // walk to the top (they do not depend on the FMAs) and spills ~500 VGPRs; tying each row's LDS
        // offset to a value produced by the FMAs two rows earlier keeps exactly two rows in flight.
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const f32x4 *base = Ks + g * KPL + (2 * yp) * KWD + lx;
            f32x4 kv[2][KS];
#pragma unroll
            for (int dx = 0; dx < KS; ++dx) kv[0][dx] = base[dx];
            float tie = 0.f;
#pragma unroll
            for (int r = 0; r < KS + 1; ++r) {
                if (r < KS) {
                    int off = (r + 1) * KWD;
                    asm volatile("" : "+v"(off) : "v"(tie));
#pragma unroll
                    for (int dx = 0; dx < KS; ++dx) kv[(r + 1) & 1][dx] = base[off + dx];
                }
                // component-major: consecutive FMAs hit different accumulators (a dot4 per tap would be a
                // 4-deep dependent chain issued back to back, stalling the in-order VALU)
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
                    for (int dx = 0; dx < KS; ++dx) {
                        if (r < KS) S0[r * KS + dx] = fmaf(q[0][g][c4], kv[r & 1][dx][c4], S0[r * KS + dx]);
                        if (r >= 1) S1[(r - 1) * KS + dx] = fmaf(q[1][g][c4], kv[r & 1][dx][c4], S1[(r - 1) * KS + dx]);
                    }
                tie = r < KS ? S0[r * KS + KS - 1] : S1[(r - 1) * KS + KS - 1];
            }
        }
    }

    // ------------------------------------------------------------------ softmax over all taps (padding taps included)
    {
        float m0 = S0[0], m1 = S1[0];
#pragma unroll
        for (int i = 1; i < T; ++i) { m0 = fmaxf(m0, S0[i]); m1 = fmaxf(m1, S1[i]); }
        float z0 = 0.f, z1 = 0.f;
#pragma unroll
        for (int i = 0; i < T; ++i) { S0[i] = __expf(S0[i] - m0); z0 += S0[i]; S1[i] = __expf(S1[i] - m1); z1 += S1[i]; }
        const float r0 = 1.0f / z0, r1 = 1.0f / z1;
#pragma unroll
        for (int i = 0; i < T; ++i) { S0[i] *= r0; S1[i] *= r1; }
    }

    constexpr int NCA = NC > 0 ? NC : 1;
    float lg[2][NCA];
    if (NC > 0) {
#pragma unroll
        for (int k = 0; k < NCA; ++k) { lg[0][k] = p.bf[min(k, p.n_cls - 1)]; lg[1][k] = lg[0][k]; }   // rows >= n_cls: dummies, never stored
    }
    const bool in0 = px < p.Wp && py0 < p.Hp, in1 = px < p.Wp && py0 + 1 < p.Hp;
    // Bounds-checked buffer stores: lanes outside the image get an offset past num_records and the hardware
    // drops the store -- no divergent branch around the stores (a branch makes LLVM sink the PV FMA chain
    // into it and spill hundreds of VGPRs).
    const u32x4 p_rsrc = make_rsrc(p.p_out, p.p_bytes), l_rsrc = make_rsrc(p.logits, p.l_bytes);

    // ------------------------------------------------------------------ pass 2: weighted values, residual, head
    for (int cb = 0; cb < CB; ++cb) {
        int t = tid;
        asm volatile("" : "+v"(t));
        stage_hr(cb, t);
        stage_dw(cb, t);
        // residual term for this thread's two pixels: from the staged window (parked in the lr_up tile, unused in this
        // pass) after the barrier, or -- fallback -- gathered from global memory now so the latency hides under the conv
        f32x4 lrv[2][G];
        if (lr_lds) stage_lr(Ls, cb, t);
        else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < G; ++g) lrv[j][g] = lr_up_at(p, n, min(py0 + j, p.Hp - 1), min(px, p.Wp - 1), cb * 8 + g * 4);
        }
        __syncthreads();
        if (lr_lds) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < G; ++g) lrv[j][g] = lr_up_lds(Ls, min(py0 + j, p.Hp - 1), min(px, p.Wp - 1), g);
        }
        conv_tile(2, t);
        __syncthreads();
        f32x4 a[2][G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            a[0][g] = f32x4{0.f, 0.f, 0.f, 0.f}; a[1][g] = a[0][g];
            const f32x4 *base = Ks + g * KPL + (2 * yp) * KWD + lx;
            f32x4 vv[2][KS];
#pragma unroll
            for (int dx = 0; dx < KS; ++dx) vv[0][dx] = base[dx];
            float tie = 0.f;
#pragma unroll
            for (int r = 0; r < KS + 1; ++r) {
                if (r < KS) {
                    int off = (r + 1) * KWD;
                    asm volatile("" : "+v"(off) : "v"(tie));
#pragma unroll
                    for (int dx = 0; dx < KS; ++dx) vv[(r + 1) & 1][dx] = base[off + dx];
                }
#pragma unroll
                for (int dx = 0; dx < KS; ++dx) {
                    if (r < KS) a[0][g] = axpy4(S0[r * KS + dx], vv[r & 1][dx], a[0][g]);
                    if (r >= 1) a[1][g] = axpy4(S1[(r - 1) * KS + dx], vv[r & 1][dx], a[1][g]);
                }
                tie = r < KS ? a[0][g][3] : a[1][g][3];
            }
        }
        // Everything below is computed unconditionally (coordinates clamped into the image) and pinned with
        // an opaque asm before the bounds test: otherwise LLVM sinks the whole PV FMA chain into the two
        // `if (inside)` blocks, which keeps all window vectors live across them (hundreds of spills).
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gy = min(py0 + j, p.Hp - 1), gx = min(px, p.Wp - 1);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const f32x4 o = lrv[j][g] + a[j][g];
                const unsigned off = (unsigned)((((((size_t)n * CB + cb) * p.Hp + gy) * p.Wp + gx) * 8 + g * 4) * sizeof(float));
                store16_buf(__builtin_bit_cast(u32x4, o), p_rsrc, (j == 0 ? in0 : in1) ? off : 0xFFFFFFF0u);
                if (NC > 0) {
#pragma unroll
                    for (int k = 0; k < NCA; ++k) {   // branch-free: class rows past n_cls re-read the last row (LDS broadcast)
                        const f32x4 w = Wf[min(k, p.n_cls - 1) * (p.C >> 2) + cb * 2 + g];
                        lg[j][k] = dot4(o, w, lg[j][k]);
                    }
                }
            }
        }
    }

    if (NC > 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (p.log_softmax) {
                float m = -INFINITY;
#pragma unroll
                for (int k = 0; k < NCA; ++k) m = fmaxf(m, k < p.n_cls ? lg[j][k] : -INFINITY);
                float z = 0.f;
#pragma unroll
                for (int k = 0; k < NCA; ++k) z += k < p.n_cls ? expf(lg[j][k] - m) : 0.f;
                const float lse = m + logf(z);
#pragma unroll
                for (int k = 0; k < NCA; ++k) lg[j][k] -= lse;
            }
#pragma unroll
            for (int k = 0; k < NCA; ++k) {
                const unsigned off = (unsigned)(((((size_t)n * p.n_cls + k) * p.Hp + py0 + j) * p.Wp + px) * sizeof(float));
                store4_buf(__float_as_uint(lg[j][k]), l_rsrc, ((j == 0 ? in0 : in1) && k < p.n_cls) ? off : 0xFFFFFFF0u);
            }
        }
    }
}

template <int KS, int NC, int TH>
int launch_creff(const CreffParams &p, hipStream_t st) {
    constexpr int R = KS / 2;
    constexpr size_t fl4 = (size_t)G * ((TH + 2 * R + 2) * (TW + 2 * R + 2) + PAD) + (size_t)G * ((TH + 2 * R) * (TW + 2 * R) + PAD) +
                           (size_t)G * ((TH + 2) * (TW + 2) + PAD) + 3 * 10 * G;
    const size_t smem = (fl4 + (NC > 0 ? (size_t)p.n_cls * (p.C >> 2) : 0)) * sizeof(f32x4);
    if (smem > 160 * 1024) return ARSEG_EUNSUPPORTED;
    static ArsegSmemAttr attr;
    if (int e = arseg_allow_smem(attr, reinterpret_cast<const void *>(creff_kernel<KS, NC, TH>), smem)) return e;
    dim3 grid(arseg_cdiv(p.Wp, TW), arseg_cdiv(p.Hp, TH), p.N);
    hipLaunchKernelGGL((creff_kernel<KS, NC, TH>), grid, dim3(16 * TH), smem, st, p);
    return arseg_launch_status();
}

template <int KS, int NC>
int dispatch_th(const CreffParams &p, hipStream_t st) {
    // pick the tallest tile that still yields >= 256 workgroups (one per CU)
    const long long tiles_x = arseg_cdiv(p.Wp, TW);
    if (tiles_x * arseg_cdiv(p.Hp, 16) * p.N >= 256) return launch_creff<KS, NC, 16>(p, st);
    if (KS == 7) {
        if (tiles_x * arseg_cdiv(p.Hp, 8) * p.N >= 256) return launch_creff<KS, NC, 8>(p, st);
        return launch_creff<KS, NC, 4>(p, st);
    }
    return launch_creff<KS, NC, 16>(p, st);
}

// ---------------------------------------------------------------------------------------------
// layout conversion to / from the channel-blocked C8 layout
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nchw_c8_kernel(const float *__restrict__ in, float *__restrict__ out, int N, int C, int HW, int to_c8) {
    const int CB = C >> 3;
    const long long total = (long long)N * CB * HW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long hw = idx % HW, ncb = idx / HW;     // ncb = n*CB + cb ; NCHW plane index = ncb*8 + j
        if (to_c8) {
            f32x4 a, b;
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[j] = in[(ncb * 8 + j) * HW + hw]; b[j] = in[(ncb * 8 + 4 + j) * HW + hw]; }
            *reinterpret_cast<f32x4 *>(out + idx * 8) = a;
            *reinterpret_cast<f32x4 *>(out + idx * 8 + 4) = b;
        } else {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(in + idx * 8), b = *reinterpret_cast<const f32x4 *>(in + idx * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { out[(ncb * 8 + j) * HW + hw] = a[j]; out[(ncb * 8 + 4 + j) * HW + hw] = b[j]; }
        }
    }
}

__global__ __launch_bounds__(256) void nhwc_c8_kernel(const float *__restrict__ in, float *__restrict__ out, int N, int C, int HW, int ld, int to_c8) {
    const int c4n = C >> 2, CB = C >> 3;
    const long long total = (long long)N * HW * c4n;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % c4n);
        const long long pix = idx / c4n, n = pix / HW, hw = pix - n * HW;
        const size_t a_nhwc = (size_t)pix * ld + c4 * 4;
        const size_t a_c8 = (((size_t)n * CB + (c4 >> 1)) * HW + hw) * 8 + (c4 & 1) * 4;
        if (to_c8) *reinterpret_cast<f32x4 *>(out + a_c8) = *reinterpret_cast<const f32x4 *>(in + a_nhwc);
        else *reinterpret_cast<f32x4 *>(out + a_nhwc) = *reinterpret_cast<const f32x4 *>(in + a_c8);
    }
}

}  // namespace

extern "C" int arseg_creff_fwd_ex(const float *hr, const float *lr, const float *wq, const float *bq, const float *wk,
                                  const float *bk, const float *wv, const float *bv, float *p_out, const float *wf,
                                  const float *bf, int n_cls, float *logits, int log_softmax, int N, int C, int Hp, int Wp,
                                  int hp, int wp, int kH, int kW, int impl, int mfma_tile_rows, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(hr); ARSEG_CHECK_PTR(lr); ARSEG_CHECK_PTR(wq); ARSEG_CHECK_PTR(bq); ARSEG_CHECK_PTR(wk); ARSEG_CHECK_PTR(bk);
    ARSEG_CHECK_PTR(wv); ARSEG_CHECK_PTR(bv); ARSEG_CHECK_PTR(p_out);
    ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(C); ARSEG_CHECK_POS(Hp); ARSEG_CHECK_POS(Wp); ARSEG_CHECK_POS(hp); ARSEG_CHECK_POS(wp);
    if (C & 7) return ARSEG_EUNSUPPORTED;
    if (kH != kW || (kH != 3 && kH != 5 && kH != 7)) return ARSEG_EUNSUPPORTED;
    if (impl < ARSEG_CREFF_AUTO || impl > ARSEG_CREFF_VALU || (mfma_tile_rows != 0 && mfma_tile_rows != 8 && mfma_tile_rows != 16)) return ARSEG_EINVAL;
    if (N > 65535) return ARSEG_EUNSUPPORTED;
    if ((size_t)N * C * Hp * Wp * sizeof(float) >= (1ull << 31)) return ARSEG_EUNSUPPORTED;   // 32-bit buffer offsets
    if (!ARSEG_ALIGNED16(hr) || !ARSEG_ALIGNED16(lr) || !ARSEG_ALIGNED16(p_out) || !ARSEG_ALIGNED16(wq) || !ARSEG_ALIGNED16(wk) ||
        !ARSEG_ALIGNED16(wv) || !ARSEG_ALIGNED16(bq) || !ARSEG_ALIGNED16(bk) || !ARSEG_ALIGNED16(bv))
        return ARSEG_EINVAL;
    const bool head = logits != nullptr;
    if (head) {
        if (!wf || !bf || n_cls <= 0) return ARSEG_EINVAL;
        if (n_cls > 32) return ARSEG_EUNSUPPORTED;
        if (!ARSEG_ALIGNED16(wf)) return ARSEG_EINVAL;
    }
    CreffParams p;
    p.hr = hr; p.lr = lr; p.wq = wq; p.bq = bq; p.wk = wk; p.bk = bk; p.wv = wv; p.bv = bv; p.wf = wf; p.bf = bf;
    p.p_out = p_out; p.logits = logits;
    p.N = N; p.C = C; p.Hp = Hp; p.Wp = Wp; p.hp = hp; p.wp = wp; p.n_cls = head ? n_cls : 0; p.log_softmax = log_softmax;
    p.p_bytes = (unsigned)((size_t)N * C * Hp * Wp * sizeof(float)); p.l_bytes = head ? (unsigned)((size_t)N * n_cls * Hp * Wp * sizeof(float)) : 0u;
    p.sy = arseg_resize_scale(hp, Hp, true); p.sx = arseg_resize_scale(wp, Wp, true);
    p.mfma_tile_rows = mfma_tile_rows;
    hipStream_t st = arseg_stream(stream);
    if (kH == 7) {
        // Two implementations.  The matrix-core kernel (creff_mfma.hip: 16-wave workgroups, 16x16 tiles) wins on wide features
        // and small maps (BiSeNet, C=256 at 1/8 resolution: 151 us vs 652 us per frame on MI355X); on the 64-channel
        // full-resolution PSPNet feature the fp32 VALU kernel below is still ahead (363 us vs 388 us). 
        // `impl` pins one of them (A/B measurements, tests).
        if (impl == ARSEG_CREFF_MFMA || (impl == ARSEG_CREFF_AUTO && C >= 128)) {
            const int st_m = arseg_creff_mfma_launch(p, st);
            if (st_m != ARSEG_EUNSUPPORTED) return st_m;
        }
        if (!head) return dispatch_th<7, 0>(p, st);
        if (n_cls <= 12) return dispatch_th<7, 12>(p, st);
        if (n_cls <= 19) return dispatch_th<7, 19>(p, st);
        return dispatch_th<7, 32>(p, st);
    }
    if (kH == 5) return head ? dispatch_th<5, 32>(p, st) : dispatch_th<5, 0>(p, st);
    return head ? dispatch_th<3, 32>(p, st) : dispatch_th<3, 0>(p, st);
}

extern "C" int arseg_creff_fwd(const float *hr, const float *lr, const float *wq, const float *bq, const float *wk,
                               const float *bk, const float *wv, const float *bv, float *p_out, const float *wf,
                               const float *bf, int n_cls, float *logits, int log_softmax, int N, int C, int Hp, int Wp,
                               int hp, int wp, int kH, int kW, arseg_stream_t stream) {
    return arseg_creff_fwd_ex(hr, lr, wq, bq, wk, bk, wv, bv, p_out, wf, bf, n_cls, logits, log_softmax, N, C, Hp, Wp, hp, wp, kH, kW,
                              ARSEG_CREFF_AUTO, 0, stream);
}

extern "C" int arseg_to_c8_fwd(const float *in, int layout, int in_ld, float *out, int N, int C, int HW, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(in); ARSEG_CHECK_PTR(out); ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(C); ARSEG_CHECK_POS(HW);
    if ((C & 7) || !ARSEG_ALIGNED16(out)) return ARSEG_EINVAL;
    long long b;
    if (layout == ARSEG_NCHW) {
        b = ((long long)N * (C >> 3) * HW + 255) / 256;
        hipLaunchKernelGGL(nchw_c8_kernel, dim3((int)(b > 16384 ? 16384 : b)), dim3(256), 0, arseg_stream(stream), in, out, N, C, HW, 1);
    } else if (layout == ARSEG_NHWC) {
        if ((in_ld & 3) || in_ld < C || !ARSEG_ALIGNED16(in)) return ARSEG_EINVAL;
        b = ((long long)N * HW * (C >> 2) + 255) / 256;
        hipLaunchKernelGGL(nhwc_c8_kernel, dim3((int)(b > 16384 ? 16384 : b)), dim3(256), 0, arseg_stream(stream), in, out, N, C, HW, in_ld, 1);
    } else return ARSEG_EINVAL;
    return arseg_launch_status();
}

extern "C" int arseg_from_c8_fwd(const float *in, float *out, int layout, int out_ld, int N, int C, int HW, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(in); ARSEG_CHECK_PTR(out); ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(C); ARSEG_CHECK_POS(HW);
    if ((C & 7) || !ARSEG_ALIGNED16(in)) return ARSEG_EINVAL;
    long long b;
    if (layout == ARSEG_NCHW) {
        b = ((long long)N * (C >> 3) * HW + 255) / 256;
        hipLaunchKernelGGL(nchw_c8_kernel, dim3((int)(b > 16384 ? 16384 : b)), dim3(256), 0, arseg_stream(stream), in, out, N, C, HW, 0);
    } else if (layout == ARSEG_NHWC) {
        if ((out_ld & 3) || out_ld < C || !ARSEG_ALIGNED16(out)) return ARSEG_EINVAL;
        b = ((long long)N * HW * (C >> 2) + 255) / 256;
        hipLaunchKernelGGL(nhwc_c8_kernel, dim3((int)(b > 16384 ? 16384 : b)), dim3(256), 0, arseg_stream(stream), in, out, N, C, HW, out_ld, 0);
    } else return ARSEG_EINVAL;
    return arseg_launch_status();
}
