// 3x3 stride-1 convolution 64 -> 64 channels (+ folded BN / bias, residual, activation) as a FUSED Winograd F(2x2,3x3): input transform, the sixteen
// 64 x 64 GEMMs and the output transform of an 8 x 8 pixel block all happen on chip -- the transformed tensors V and M never exist in memory.
//   reference layers: extractors.BasicBlock of layer1 (/root/reference/model/extractors.py:35-66, 64 -> 64 at 1/4 resolution) and PSPUpsample up_3
//   (/root/reference/model/pspnet.py:34-46, 64 -> 64 at full resolution behind a x2 bilinear upsample) -- the layers where the un-fused F(4x4,3x3)
//   route loses to the direct conv (their transforms would move 6x the bytes of a 40 us conv) and the direct conv pays nine products per tap.
//
// Why this shape fits the machine.  F(2x2,3x3) has 4 x 4 = 16 frequency positions, a workgroup has 16 waves: wave p owns position p for the whole
// launch and keeps ITS 64 x 64 weight slice U_p = (G g G^T)_p -- split into fp16 hi / lo like every f16x3 operand -- in 64 VGPRs as the MFMA A operand.
// No weight is ever re-read from LDS or L2 (the patch-resident direct kernel streams 147 KB of weights per tile).  Per 8 x 8 block (16 Winograd tiles):
//   phase 1 (all waves)  the 10 x 10 x 64 input patch (LDS, fetched by LDS-DMA one block ahead) -> V = B^T d B for the 16 tiles, split to hi / lo halves, into LDS
//                        in MFMA B-operand order [position][tile][k]; thread = (tile, channel pair, half of the positions), v_pk_add_f32 on the pair
//   phase 2 (wave p)     D_p[cout][tile] = U_p . V_p^T: 4 row blocks x 2 k blocks x 3 products = 24 v_mfma_f32_16x16x32_f16; M_p written over V_p (same 4 KB)
//   phase 3 (all waves)  Y = A^T M A per (tile, cout), y * scale + bias (+ residual), activation, 2 x 2 pixels stored; thread = (tile, cout)
// 2.25x fewer products than the direct conv and 24 MFMAs per wave and block instead of 54 + operand streaming; the price is the transforms' VALU work
// (~90 + ~50 instructions per thread and block), which is what bounds the kernel.  Numerics: F(2x2,3x3)'s matrices hold 0, +-1, +-1/2 -- the transform
// is exact in fp32 up to one rounding per add, measured error against fp64 like the direct f16x3 conv (tests/test_gpu_ops.py::test_conv3x3_wino2).
//
// upsample2x != 0: the conv runs on the x2 bilinear (align_corners=False) upsample of the low-resolution input, which is never materialised: the patch is
// blended from the 6 x 6 low-resolution pixels under it by the VALU instead of being fetched by DMA (PSPUpsample).
#include "arseg_common.h"

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

constexpr int C = 64, NT = 512, NW = NT / 64, BS = 8;     // channels (in == out), threads (8 waves: two frequency positions each, 256 registers per lane), block edge in pixels
constexpr int PW = BS + 2, PPX = PW * PW;             // patch: 10 x 10 pixels
constexpr int PATCH_B = PPX * C * 4;                  // 25600 bytes
constexpr int TROW = 272;                             // bytes per tile row of a position's V / M slab (256 + 16: the 16 tiles of a ds_read_b128 spread over the banks)
constexpr int VP = 16 * TROW;                         // one position
constexpr int VM_OFF = 2 * PATCH_B, LO_OFF = VM_OFF + 16 * VP;
constexpr int LW = 6, LO_B = LW * LW * C * 4;                 // upsample variant: the 6 x 6 low-resolution pixels under a patch, double buffered
constexpr int SMEM_BYTES = LO_OFF, SMEM_BYTES_UP2 = LO_OFF + 2 * LO_B;
constexpr unsigned BAD = 0x80000000u;

struct W2Params {
    const float *x;            // NHWC fp32 [N][H][W][in_ld] (upsample2x: [N][H/2][W/2][in_ld])
    const unsigned char *u;    // split rows [16 positions * 64 cout][64 cin]
    const float *scale, *bias, *res;
    float *out;
    int N, H, W, in_ld, out_ld, res_ld, act, up2;
    float slope;
    unsigned x_bytes, out_bytes;
    unsigned *range_flag;
    float range_limit;
    int nbx, nby, nblk;
};

__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p; }
__device__ __forceinline__ u32x4 make_rsrc(const void *base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu)),
                 (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
}
// workgroup barrier that orders LDS traffic only: hipcc's __syncthreads() also drains vmcnt, i.e. waits for the output stores of the block before
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void store4_buf(float v, const u32x4 rsrc, unsigned voff) {      // fire and forget (invisible to hipcc's vmcnt bookkeeping)
    asm volatile("buffer_store_dword %0, %1, %2, 0 offen" ::"v"(v), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void dma16_buf(const u32x4 rsrc, unsigned voff, unsigned lds_base) {   // LDS[lds_base + lane*16] <- buffer[voff]
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_base), "v"(voff), "s"(rsrc) : "memory");
}

#ifdef W2_TIMING      // dev builds (tools/time_wino2.py): per-wave shader-clock ticks per phase, accumulated in scalar registers, flushed once per workgroup
__device__ unsigned long long g_w2_dbg[16 * 8];
#define W2_STAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc_[i] += now_ - tprev_; tprev_ = now_; } while (0)
#else
#define W2_STAMP(i) do { } while (0)
#endif

// swizzle key of patch pixel (py, px): window pixel (i, j) of tile (ty, tx) is patch pixel (2 ty + i, 2 tx + j), so the key is the tile index plus a
// constant per window position -- the 16 tiles of one read get 16 different keys
__device__ __forceinline__ int swz(int py, int px) { return (4 * (py >> 1) + (px >> 1)) & 15; }

template <bool UP2>
__global__ __launch_bounds__(NT, 2) void wino2_kernel(const W2Params p) {
#ifdef W2_TIMING
    unsigned long long tacc_[7] = {};
    unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
#endif
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;

    // ---- once per launch: this wave's weight slice U_p as MFMA A operands (row = cout, 8 consecutive k per lane), epilogue constants
    h16x8 uh[2][4][2], ul[2][4][2];            // [position of the wave][cout block][k block]: 128 VGPRs
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const unsigned char *r = p.u + ((size_t)((2 * wave + pp) * C + nb * 16 + i16) * 2 + kb) * 128 + kq * 16;
                uh[pp][nb][kb] = *reinterpret_cast<const h16x8 *>(r);
                ul[pp][nb][kb] = *reinterpret_cast<const h16x8 *>(r + 64);
            }
    const float sc = p.scale ? p.scale[lane] : 1.0f, bi = p.bias ? p.bias[lane] : 0.0f;
    const u32x4 x_rsrc = make_rsrc(p.x, p.x_bytes), o_rsrc = make_rsrc(p.out, p.out_bytes);
    const unsigned lds0 = lds_addr(smem);
    const int per_img = p.nbx * p.nby;
    const int Hs = UP2 ? p.H >> 1 : p.H, Ws = UP2 ? p.W >> 1 : p.W;      // extent of the tensor in memory

    // the input patch of block `blk` -> patch buffer pb.  Plain: 25 LDS-DMA instructions of 4 pixels (1 KiB) each, instruction i by wave i & 15.
    auto fetch_patch = [&](int blk, int pb) {
        const int n = blk / per_img, rem = blk - n * per_img, by0 = (rem / p.nbx) * BS, bx0 = (rem - (rem / p.nbx) * p.nbx) * BS;
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            const int ins = wave + NW * rep;
            if (ins < (PPX + 3) / 4) {
                const int px = ins * 4 + (lane >> 4), py = px / PW, pxx = px - py * PW;
                const int gy = by0 - 1 + py, gx = bx0 - 1 + pxx;
                const bool ok = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W && px < PPX;
                // the 16-byte slot a lane fetches is XORed with a pixel key (swz): the transform reads one slot of SIXTEEN pixels (one per tile) with a
                // ds_read_b128 -- un-swizzled they would all sit on the same four banks
                const unsigned off = ok ? (unsigned)(((n * p.H + gy) * p.W + gx) * p.in_ld) * 4u + (unsigned)((lane & 15) ^ swz(py, pxx)) * 16u : BAD;
                dma16_buf(x_rsrc, off, lds0 + (unsigned)pb * PATCH_B + (unsigned)ins * 1024u);
            }
        }
    };
    // x2 bilinear (align_corners=False) variant.  Patch pixel (gy, gx) of the upsampled image = blend of low-resolution rows y0 = floor(gy / 2 - .25), y0 + 1
    // (ATen: src = (dst + .5) / 2 - .5 clamped at 0, the upper neighbour clamped to the last row), zero outside the upsampled image (the conv's padding).
    // The 6 x 6 low-resolution pixels under the patch (rows by0 / 2 - 1 .. by0 / 2 + 4) are fetched by LDS-DMA one block ahead (9 instructions of 4 pixels)
    // and the 10 x 10 patch is blended from LDS -- the first version blended from global memory inside the MFMA phase: 3 k cycles of load latency per block.
    auto fetch_low = [&](int blk, int lb) {
        const int n = blk / per_img, rem = blk - n * per_img, ly0 = (rem / p.nbx) * (BS / 2) - 1, lx0 = (rem - (rem / p.nbx) * p.nbx) * (BS / 2) - 1;
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            const int ins = wave + NW * rep;
            if (ins < (LW * LW) / 4) {
                const int px = ins * 4 + (lane >> 4), py = px / LW, pxx = px - py * LW;
                const int gy = ly0 + py, gx = lx0 + pxx;
                const bool ok = (unsigned)gy < (unsigned)Hs && (unsigned)gx < (unsigned)Ws;
                const unsigned off = ok ? (unsigned)(((n * Hs + gy) * Ws + gx) * p.in_ld) * 4u + (unsigned)(lane & 15) * 16u : BAD;
                dma16_buf(x_rsrc, off, lds0 + (unsigned)LO_OFF + (unsigned)lb * LO_B + (unsigned)ins * 1024u);
            }
        }
    };
    auto build_patch_up2 = [&](int blk, int pb, int lb) {
        const int rem = blk % per_img, by0 = (rem / p.nbx) * BS, bx0 = (rem - (rem / p.nbx) * p.nbx) * BS;
        const int ly0 = by0 / 2 - 1, lx0 = bx0 / 2 - 1;
        float *P = reinterpret_cast<float *>(smem + pb * PATCH_B);
        const float *L = reinterpret_cast<const float *>(smem + LO_OFF + lb * LO_B);
        for (int it = tid; it < PPX * (C / 4); it += NT) {
            const int px = it >> 4, c4 = (it & 15) * 4, py = px / PW, pxx = px - py * PW;
            const int gy = by0 - 1 + py, gx = bx0 - 1 + pxx;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) {
                const float sy = fmaxf(0.5f * ((float)gy + 0.5f) - 0.5f, 0.0f), sx = fmaxf(0.5f * ((float)gx + 0.5f) - 0.5f, 0.0f);
                const int y0 = (int)sy, x0 = (int)sx, y1 = min(y0 + 1, Hs - 1), x1 = min(x0 + 1, Ws - 1);
                const float ly = sy - (float)y0, lx = sx - (float)x0;
                const float *b = L + c4;
                const f32x4 a00 = *reinterpret_cast<const f32x4 *>(b + ((y0 - ly0) * LW + x0 - lx0) * C), a01 = *reinterpret_cast<const f32x4 *>(b + ((y0 - ly0) * LW + x1 - lx0) * C);
                const f32x4 a10 = *reinterpret_cast<const f32x4 *>(b + ((y1 - ly0) * LW + x0 - lx0) * C), a11 = *reinterpret_cast<const f32x4 *>(b + ((y1 - ly0) * LW + x1 - lx0) * C);
                // ATen's order of operations for upsample_bilinear2d: (1-ly) * ((1-lx) a00 + lx a01) + ly * ((1-lx) a10 + lx a11)
                v = (1.0f - ly) * ((1.0f - lx) * a00 + lx * a01) + ly * ((1.0f - lx) * a10 + lx * a11);
            }
            *reinterpret_cast<f32x4 *>(P + px * C + (((c4 >> 2) ^ swz(py, pxx)) << 2)) = v;
        }
    };

    // phase-A role: wave w owns positions (xi, nu0) and (xi, nu0 + 1), xi = w >> 1, nu0 = 2 (w & 1).  B^T has two +-1 entries per row, so a position is a
    // signed sum of 2 x 2 window pixels; the two positions of a wave share their window rows and one of their three window columns:
    //   nu 0, 1 (columns 0, 1, 2):  V0 = eA - eC, V1 = eB + eC          nu 2, 3 (columns 1, 2, 3):  V0 = eB - eA, V1 = eA - eC        eX = si1 d[ri1][X] + si2 d[ri2][X]
    const int xi = wave >> 1, odd = wave & 1;
    const int ri1 = xi == 0 ? 0 : 1, ri2 = xi == 3 ? 3 : 2, cA = odd;
    const float si1 = xi == 2 ? -1.f : 1.f, si2 = (xi == 0 || xi == 3) ? -1.f : 1.f;
    u16x2 vm16 = {0, 0};

    int blk = blockIdx.x, pb = 0;
    if (blk < p.nblk) {
        if constexpr (UP2) fetch_low(blk, 0); else fetch_patch(blk, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    if constexpr (UP2) {
        if (blk < p.nblk) build_patch_up2(blk, 0, 0);
        lds_barrier();
    }
    W2_STAMP(0);
    for (; blk < p.nblk; blk += gridDim.x, pb ^= 1) {
        const int n = blk / per_img, rem = blk - n * per_img, by0 = (rem / p.nbx) * BS, bx0 = (rem - (rem / p.nbx) * p.nbx) * BS;
        // the next block's patch goes into the other buffer (read last in the previous block's phase 1) while this block is transformed and multiplied:
        // requested first, a whole phase 1 + 2 ahead of its wait
        const int nblk_next = blk + gridDim.x;
        if (nblk_next < p.nblk) {
            if constexpr (UP2) fetch_low(nblk_next, pb ^ 1); else fetch_patch(nblk_next, pb ^ 1);
        }
        // ------------------------------------------------------------ phase A (wave p = position p): V_p of the 16 tiles straight into MFMA B operands, then
        // D_p[cout][tile] = U_p . V_p^T.  No V in LDS and no barrier between transform and product: V_p[tile][c] = s11 d[i1][j1] + s12 d[i1][j2] + s21 d[i2][j1]
        // + s22 d[i2][j2] (B^T has two +-1 entries per row), lane (i16 = tile, kq) builds the 8 channels of its k slice for both k blocks -- the waves
        // drift apart, one wave's transform (VALU, LDS) runs beside another's MFMAs.  (v1 of this kernel transformed all positions in a phase of its own, wrote V
        // to LDS and read it back behind a barrier: 8.5 k cycles per block, VALU phases and the MFMA phase strictly one after the other.)
        {
            const int ty = i16 >> 2, tx = i16 & 3;
            // LDS byte address of (pixel, slot kq * 2) for the six window pixels; slot sl = kb * 8 + kq * 2 + q differs from it in disjoint bits, so the other
            // three reads of a pixel are an XOR with a constant
            auto base = [&](int r, int cc) {
                const int py = 2 * ty + r, px = 2 * tx + cc;
                return lds0 + (unsigned)pb * PATCH_B + (unsigned)(py * PW + px) * (C * 4) + (unsigned)(((kq * 2) ^ swz(py, px)) << 4);
            };
            const unsigned a1A = base(ri1, cA), a1B = base(ri1, cA + 1), a1C = base(ri1, cA + 2), a2A = base(ri2, cA), a2B = base(ri2, cA + 1), a2C = base(ri2, cA + 2);
            auto ld = [](unsigned a) { return *reinterpret_cast<const f32x4 __attribute__((address_space(3))) *>((size_t)a); };
            f32x4 acc[2][4];
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[pp][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                unsigned hh[2][4], ll[2][4];
#pragma unroll
                for (int q = 0; q < 2; ++q) {          // two 4-channel slots of the lane's 8 channels
                    const unsigned x = (unsigned)(kb * 128 + q * 16);
                    const f32x4 eA = si1 * ld(a1A ^ x) + si2 * ld(a2A ^ x), eB = si1 * ld(a1B ^ x) + si2 * ld(a2B ^ x), eC = si1 * ld(a1C ^ x) + si2 * ld(a2C ^ x);
                    const f32x4 v0 = odd ? eB - eA : eA - eC, v1 = odd ? eA - eC : eB + eC;
                    arseg_split_f16(v0, hh[0][2 * q], hh[0][2 * q + 1], ll[0][2 * q], ll[0][2 * q + 1]);
                    arseg_split_f16(v1, hh[1][2 * q], hh[1][2 * q + 1], ll[1][2 * q], ll[1][2 * q + 1]);
                    // operand range watch on the hi halves (round toward zero: |v| >= 65504 <=> |hi| = 65504 or inf), one packed 16-bit max per two values
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) {
                        vm16 = __builtin_elementwise_max(vm16, __builtin_bit_cast(u16x2, hh[pp][2 * q] & 0x7fff7fffu));
                        vm16 = __builtin_elementwise_max(vm16, __builtin_bit_cast(u16x2, hh[pp][2 * q + 1] & 0x7fff7fffu));
                    }
                }
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const h16x8 bh = __builtin_bit_cast(h16x8, u32x4{hh[pp][0], hh[pp][1], hh[pp][2], hh[pp][3]}), bl = __builtin_bit_cast(h16x8, u32x4{ll[pp][0], ll[pp][1], ll[pp][2], ll[pp][3]});
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) acc[pp][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ul[pp][nb][kb], bh, acc[pp][nb], 0, 0, 0);
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) acc[pp][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(uh[pp][nb][kb], bl, acc[pp][nb], 0, 0, 0);
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) acc[pp][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(uh[pp][nb][kb], bh, acc[pp][nb], 0, 0, 0);
                }
            }
            // lane (i16 = tile, kq) holds couts 16 nb + 4 kq .. + 3 of its tile
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                unsigned char *Mp = smem + VM_OFF + (2 * wave + pp) * VP + i16 * TROW;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) *reinterpret_cast<f32x4 *>(Mp + (nb * 16 + kq * 4) * 4) = acc[pp][nb];
            }
        }
        W2_STAMP(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the prefetched patch has landed (before this block's stores join the queue)
        lds_barrier();
        W2_STAMP(4);
        // ------------------------------------------------------------ phase B: output transform + epilogue, thread = (tile, cout), two tiles per wave
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int t3 = 2 * wave + tt, ty3 = t3 >> 2, tx3 = t3 & 3;
            const unsigned char *Mb = smem + VM_OFF + t3 * TROW + lane * 4;
            float m[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) m[q] = *reinterpret_cast<const float *>(Mb + q * VP);
            float s0[4], s1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0[j] = m[0 * 4 + j] + m[1 * 4 + j] + m[2 * 4 + j];
                s1[j] = m[1 * 4 + j] - m[2 * 4 + j] - m[3 * 4 + j];
            }
            const float y[2][2] = {{s0[0] + s0[1] + s0[2], s0[1] - s0[2] - s0[3]}, {s1[0] + s1[1] + s1[2], s1[1] - s1[2] - s1[3]}};
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int gy = by0 + 2 * ty3 + i, gx = bx0 + 2 * tx3 + j;
                    if (gy < p.H && gx < p.W) {
                        const unsigned px = (unsigned)((n * p.H + gy) * p.W + gx);
                        float v = y[i][j] * sc + bi;
                        if (p.res) v += p.res[(size_t)px * p.res_ld + lane];
                        if (p.act == ARSEG_ACT_RELU) v = fmaxf(v, 0.f);
                        else if (p.act == ARSEG_ACT_PRELU) v = v >= 0.f ? v : v * p.slope;
                        else if (p.act == ARSEG_ACT_SIGMOID) v = 1.0f / (1.0f + __expf(-v));
                        store4_buf(v, o_rsrc, (px * (unsigned)p.out_ld + (unsigned)lane) * 4u);
                    }
                }
        }
        if constexpr (UP2) { if (nblk_next < p.nblk) build_patch_up2(nblk_next, pb ^ 1, pb ^ 1); }      // from the staged low-resolution pixels (landed before barrier 2)
        W2_STAMP(5);
        lds_barrier();            // the next block's phase 1 overwrites the M slabs
        W2_STAMP(6);
    }
    if (p.range_flag && (vm16[0] >= 0x7bffu || vm16[1] >= 0x7bffu)) atomicOr(p.range_flag, 1u);      // 0x7bff = 65504
#ifdef W2_TIMING
    if (lane == 0)
        for (int i = 0; i < 7; ++i) atomicAdd(&g_w2_dbg[8 * (2 * wave) + i], tacc_[i]), atomicAdd(&g_w2_dbg[8 * (2 * wave + 1) + i], tacc_[i]);
    if (tid == 0) atomicAdd(&g_w2_dbg[7], 1ull);
#endif
}

}  // namespace

// Weights: OIHW [64][64][3][3] fp32 -> U[p = 4 xi + nu][cout][cin] = (G g G^T)[xi][nu], G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]] (host, fp64 arithmetic).
// The caller scales rows per output channel and splits them with arseg_split_weight_f16x3_host([16 * 64 rows][64]) -- the layout the kernel loads.
#ifdef W2_TIMING
extern "C" void arseg__w2_dbg_read(unsigned long long *host, int reset) {
    (void)hipDeviceSynchronize();
    if (host) (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_w2_dbg), sizeof(unsigned long long) * 128);
    if (reset) { static unsigned long long z[128]; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_w2_dbg), z, sizeof(z)); }
}
#endif

extern "C" int arseg_wino2_pack_weight_host(const float *w, int Cout, int Cin, float *out) {
    if (!w || !out || Cout <= 0 || Cin <= 0) return ARSEG_EINVAL;
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci) {
            const float *g = w + ((size_t)co * Cin + ci) * 9;
            double tmp[4][3];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 3; ++j) tmp[i][j] = G[i][0] * g[0 * 3 + j] + G[i][1] * g[1 * 3 + j] + G[i][2] * g[2 * 3 + j];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j)
                    out[((size_t)(i * 4 + j) * Cout + co) * Cin + ci] = (float)(tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2]);
        }
    return ARSEG_OK;
}

extern "C" int arseg_conv3x3_wino2_fwd(const float *in, int in_ld, const void *u_split, const float *scale, const float *bias, const float *residual,
                                       int res_ld, float *out, int out_ld, int N, int H, int W, int Cin, int Cout, int act, float prelu_slope,
                                       int upsample2x, void *range_flag, float range_limit, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(in); ARSEG_CHECK_PTR(u_split); ARSEG_CHECK_PTR(out);
    ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(H); ARSEG_CHECK_POS(W);
    if (Cin != C || Cout != C) return ARSEG_EUNSUPPORTED;
    if (in_ld < C || (in_ld & 3) || out_ld < C || (residual && res_ld < C) || !ARSEG_ALIGNED16(in) || !ARSEG_ALIGNED16(u_split)) return ARSEG_EINVAL;
    if (upsample2x && ((H & 1) || (W & 1))) return ARSEG_EINVAL;
    if (reinterpret_cast<uintptr_t>(range_flag) & 3) return ARSEG_EINVAL;
    const long long in_px = (long long)N * (upsample2x ? (H / 2) * (long long)(W / 2) : (long long)H * W);
    if (in_px * in_ld * 4 >= (1ll << 31)) return ARSEG_EUNSUPPORTED;          // 32-bit buffer offsets of the patch DMA
    W2Params p;
    p.x = in; p.u = reinterpret_cast<const unsigned char *>(u_split); p.scale = scale; p.bias = bias; p.res = residual; p.out = out;
    p.N = N; p.H = H; p.W = W; p.in_ld = in_ld; p.out_ld = out_ld; p.res_ld = res_ld; p.act = act; p.up2 = upsample2x ? 1 : 0; p.slope = prelu_slope;
    p.x_bytes = (unsigned)(in_px * in_ld * 4);
    if ((long long)N * H * W * out_ld * 4 >= (1ll << 32)) return ARSEG_EUNSUPPORTED;      // 32-bit store offsets
    p.out_bytes = (unsigned)((long long)N * H * W * out_ld * 4);
    p.range_flag = reinterpret_cast<unsigned *>(range_flag); p.range_limit = range_limit > 0.0f ? range_limit : 65504.0f;
    p.nbx = arseg_cdiv(W, BS); p.nby = arseg_cdiv(H, BS);
    const long long nblk = (long long)p.nbx * p.nby * N;
    if (nblk >= (1ll << 31)) return ARSEG_EUNSUPPORTED;
    p.nblk = (int)nblk;
    static ArsegSmemAttr attr0, attr1;
    if (int e = upsample2x ? arseg_allow_smem(attr1, reinterpret_cast<const void *>(wino2_kernel<true>), SMEM_BYTES_UP2)
                           : arseg_allow_smem(attr0, reinterpret_cast<const void *>(wino2_kernel<false>), SMEM_BYTES)) return e;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const int grid = (int)(nblk < cus ? nblk : cus);
    if (upsample2x) hipLaunchKernelGGL(wino2_kernel<true>, dim3(grid), dim3(NT), SMEM_BYTES_UP2, arseg_stream(stream), p);
    else hipLaunchKernelGGL(wino2_kernel<false>, dim3(grid), dim3(NT), SMEM_BYTES, arseg_stream(stream), p);
    return arseg_launch_status();
}
