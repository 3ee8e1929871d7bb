// Winograd F(4x4, 3x3) transforms for the 3x3 stride-1 convolutions of the backbones (pad == dilation, any dilation).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      (Lavin & Gray, "Fast Algorithms for Convolutional Neural Networks")
//
// One 6x6 input patch yields a 4x4 output tile with 36 multiplies per (cin,cout) pair instead of 144: the 36 element-wise
// products over all tiles are 36 independent [T x Cin] x [Cin x Cout] GEMMs, which run on the fp32-MFMA implicit-GEMM
// kernel in batched mode (conv_igemm.hip).  This file holds the two memory-bound transforms around them and the host-side
// weight transform.  Dilated convs (layer3/4 of the dilated ResNet, model/extractors.py:139-142) are handled exactly by
// the polyphase view: a dilation-d conv is d*d independent dense 3x3 convs on the sub-images {(y,x): y%d==sy, x%d==sx}.
//
// Layouts: V = [36][T][C], M = [36][T][Cout], T = N*d*d*tilesY*tilesX, tile t = (((n*d+sy)*d+sx)*tilesY+ty)*tilesX+tx,
// tilesY = ceil(ceil(H/d)/4).  Threads run along channels (coalesced 4-byte accesses, 36 registers of patch per thread).
// fp32 throughout; the transform constants grow the rounding error to ~1e-5 relative (vs 1e-6 for the direct form).
#include "arseg_common.h"
#ifndef WINO_NT_LOADS
#define WINO_NT_LOADS 1
#endif

namespace {

struct WinoGeom { int N, H, W, d, tilesY, tilesX, T; };

__host__ __device__ inline WinoGeom make_geom(int N, int H, int W, int d) {
    WinoGeom g;
    g.N = N; g.H = H; g.W = W; g.d = d;
    g.tilesY = ((H + d - 1) / d + 3) / 4;
    g.tilesX = ((W + d - 1) / d + 3) / 4;
    g.T = N * d * d * g.tilesY * g.tilesX;
    return g;
}

template <typename F>
__device__ __forceinline__ void bt6(const F x[6], F r[6]) {
    r[0] = 4.f * x[0] - 5.f * x[2] + x[4];
    r[1] = -4.f * x[1] - 4.f * x[2] + x[3] + x[4];
    r[2] = 4.f * x[1] - 4.f * x[2] - x[3] + x[4];
    r[3] = -2.f * x[1] - x[2] + 2.f * x[3] + x[4];
    r[4] = 2.f * x[1] - x[2] - 2.f * x[3] + x[4];
    r[5] = 4.f * x[1] - 5.f * x[3] + x[5];
}

template <typename F>
__device__ __forceinline__ void at6(const F m[6], F y[4]) {
    y[0] = m[0] + m[1] + m[2] + m[3] + m[4];
    y[1] = m[1] - m[2] + 2.f * m[3] - 2.f * m[4];
    y[2] = m[1] + m[2] + 4.f * m[3] + 4.f * m[4];
    y[3] = m[1] - m[2] + 8.f * m[3] - 8.f * m[4] + m[5];
}

__device__ __forceinline__ void tile_origin(const WinoGeom &g, int t, int &n, int &y0, int &x0) {
    const int tx = t % g.tilesX; t /= g.tilesX;
    const int ty = t % g.tilesY; t /= g.tilesY;
    const int sx = t % g.d; t /= g.d;
    const int sy = t % g.d;
    n = t / g.d;
    y0 = sy + g.d * 4 * ty;     // image row of the tile's first output
    x0 = sx + g.d * 4 * tx;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// One transformed value (pair) of frequency k, tile t, channel c.  SPLIT: V is written in the "split rows" operand format of
// arseg_gemm_x3_fwd (csrc/gemm_x3.hip: per 32 channels 32 hi halves then 32 lo halves, the same 4 bytes per value) so that the batched
// GEMM can stage it by LDS-DMA; the running |V| maximum feeds the operand range word (the GEMM no longer sees fp32 values to watch).
template <bool SPLIT, typename F>
__device__ __forceinline__ void put_v(float *__restrict__ V, size_t row, int C, int c, const F v, float &vmax) {
    if constexpr (SPLIT) {
        static_assert(sizeof(F) == 8, "split rows are written two channels at a time");
        unsigned h, l;
        arseg_split_f16_pair(v[0], v[1], h, l);
        vmax = fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1])));
        unsigned char *o = reinterpret_cast<unsigned char *>(V) + (row * C) * 4 + (c >> 5) * 128 + (c & 31) * 2;
        *reinterpret_cast<unsigned *>(o) = h;
        *reinterpret_cast<unsigned *>(o + 64) = l;
    } else {
        *reinterpret_cast<F *>(V + row * C + c) = v;
    }
}

// F = float (any C) or f32x2 (C even, 8-byte aligned rows): channels per thread
template <typename F, bool SPLIT = false>
__global__ __launch_bounds__(256) void wino43_input_kernel(const float *__restrict__ in, int in_ld, float *__restrict__ V, int C,
                                                           WinoGeom g, float vscale, unsigned *range_flag, float range_limit) {
    constexpr int VW = sizeof(F) / sizeof(float);
    const int Cv = C / VW, total = g.T * Cv;
    float vmax = 0.f;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int t = idx / Cv, c = (idx - t * Cv) * VW;
        int n, y0, x0;
        tile_origin(g, t, n, y0, x0);
        const float *base = in + (size_t)n * g.H * g.W * in_ld + c;
        F d[6][6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int y = y0 + g.d * (i - 1);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int x = x0 + g.d * (j - 1);
                d[i][j] = F(0.f);
                if ((unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W) d[i][j] = *reinterpret_cast<const F *>(base + ((size_t)y * g.W + x) * in_ld);
            }
        }
        F tmp[6][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {            // B^T d : transform every column
            F col[6], r[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) col[i] = d[i][j];
            bt6(col, r);
#pragma unroll
            for (int i = 0; i < 6; ++i) tmp[i][j] = r[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {            // (B^T d) B : transform every row
            F r[6];
            bt6(tmp[i], r);
#pragma unroll
            for (int j = 0; j < 6; ++j) put_v<SPLIT>(V, (size_t)(i * 6 + j) * g.T + t, C, c, F(r[j] * vscale), vmax);
        }
    }
    if (SPLIT && range_flag && vmax > range_limit) atomicOr(range_flag, 1u);
}

// Input transform with the x2 bilinear upsample of PSPUpsample (F.upsample default = align_corners=False,
// model/pspnet.py:45) fused in: `in` is the LOW-resolution tensor [N,H/2,W/2,C]; the 6x6 patch of the (never
// materialised) upsampled image is built from a 4x4 low-resolution neighbourhood.  With scale exactly 2 the source
// offsets are the constants 0.25 / 0.75, and clamping the neighbourhood loads to the image edge reproduces ATen's
// border handling exactly (src < 0 -> 0; i1 = min(i0+1, h-1)).  Dilation 1 only.
template <typename F, bool SPLIT = false>
__global__ __launch_bounds__(256) void wino43_input_up2_kernel(const float *__restrict__ in, int in_ld, float *__restrict__ V, int C,
                                                               WinoGeom g, float vscale, unsigned *range_flag, float range_limit) {
    constexpr int VW = sizeof(F) / sizeof(float);
    const int Cv = C / VW, total = g.T * Cv, h = g.H >> 1, w = g.W >> 1;
    float vmax = 0.f;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int t = idx / Cv, c = (idx - t * Cv) * VW;
        int n, y0, x0;
        tile_origin(g, t, n, y0, x0);                     // multiples of 4 (d == 1)
        const float *base = in + (size_t)n * h * w * in_ld + c;
        const int by = (y0 >> 1) - 1, bx = (x0 >> 1) - 1;
        F L[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = min(max(by + i, 0), h - 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xx = min(max(bx + j, 0), w - 1);
                L[i][j] = *reinterpret_cast<const F *>(base + ((size_t)yy * w + xx) * in_ld);
            }
        }
        F rows[6][4];                                     // vertical interpolation: upsampled rows y0-1 .. y0+4
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int k = i >> 1;                         // rows (y0-1,y0) use L[0],L[1]; (y0+1,y0+2) L[1],L[2]; (y0+3,y0+4) L[2],L[3]
            const float l = (i & 1) ? 0.75f : 0.25f;
            const bool ok = (unsigned)(y0 - 1 + i) < (unsigned)g.H;
#pragma unroll
            for (int j = 0; j < 4; ++j) rows[i][j] = ok ? (1.f - l) * L[k][j] + l * L[k + 1][j] : F(0.f);
        }
        F d[6][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int k = j >> 1;
            const float l = (j & 1) ? 0.75f : 0.25f;
            const bool ok = (unsigned)(x0 - 1 + j) < (unsigned)g.W;
#pragma unroll
            for (int i = 0; i < 6; ++i) d[i][j] = ok ? (1.f - l) * rows[i][k] + l * rows[i][k + 1] : F(0.f);
        }
        F tmp[6][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            F col[6], r[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) col[i] = d[i][j];
            bt6(col, r);
#pragma unroll
            for (int i = 0; i < 6; ++i) tmp[i][j] = r[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            F r[6];
            bt6(tmp[i], r);
#pragma unroll
            for (int j = 0; j < 6; ++j) put_v<SPLIT>(V, (size_t)(i * 6 + j) * g.T + t, C, c, F(r[j] * vscale), vmax);
        }
    }
    if (SPLIT && range_flag && vmax > range_limit) atomicOr(range_flag, 1u);
}

__device__ __forceinline__ float wino_act(float v, int act, float slope) {
    switch (act) {
        case ARSEG_ACT_RELU: return fmaxf(v, 0.0f);
        case ARSEG_ACT_PRELU: return v >= 0.0f ? v : v * slope;
        case ARSEG_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
        default: return v;
    }
}

template <typename F>
__global__ __launch_bounds__(256) void wino43_output_kernel(const float *__restrict__ M, const float *__restrict__ scale,
                                                            const float *__restrict__ bias, const float *__restrict__ res, int res_ld,
                                                            float *__restrict__ out, int out_ld, int C, int act, float slope, WinoGeom g, float mscale) {
    constexpr int VW = sizeof(F) / sizeof(float);
    const int Cv = C / VW, total = g.T * Cv;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int t = idx / Cv, c = (idx - t * Cv) * VW;
        int n, y0, x0;
        tile_origin(g, t, n, y0, x0);
        F m[6][6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
#if WINO_NT_LOADS
                m[i][j] = __builtin_nontemporal_load(reinterpret_cast<const F *>(M + ((size_t)(i * 6 + j) * g.T + t) * C + c));
#else
                m[i][j] = *reinterpret_cast<const F *>(M + ((size_t)(i * 6 + j) * g.T + t) * C + c);
#endif
            }
        F tmp[4][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {            // A^T m : columns
            F col[6], y[4];
#pragma unroll
            for (int i = 0; i < 6; ++i) col[i] = m[i][j];
            at6(col, y);
#pragma unroll
            for (int i = 0; i < 4; ++i) tmp[i][j] = y[i];
        }
        F sc = F(1.f), bi = F(0.f);
        if (scale) sc = *reinterpret_cast<const F *>(scale + c);
        if (bias) bi = *reinterpret_cast<const F *>(bias + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) {            // (A^T m) A : rows, then the conv epilogue
            F y[4];
            at6(tmp[i], y);
            const int oy = y0 + g.d * i;
            if (oy >= g.H) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ox = x0 + g.d * j;
                if (ox >= g.W) continue;
                const size_t pix = ((size_t)n * g.H + oy) * g.W + ox;
                F v = (y[j] * mscale) * sc + bi;
                if (res) v += *reinterpret_cast<const F *>(res + pix * res_ld + c);
                float *vp = reinterpret_cast<float *>(&v);
#pragma unroll
                for (int e = 0; e < VW; ++e) vp[e] = wino_act(vp[e], act, slope);
                *reinterpret_cast<F *>(out + pix * out_ld + c) = v;
            }
        }
    }
}

inline int grid_for(long long total) {
    long long b = (total + 255) / 256;
    return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" long long arseg_wino43_tiles(int N, int H, int W, int dil) {
    if (N <= 0 || H <= 0 || W <= 0 || dil <= 0) return 0;
    const long long t = (long long)N * dil * dil * (((H + dil - 1) / dil + 3) / 4) * (((W + dil - 1) / dil + 3) / 4);
    return t;
}

static int wino43_input(const float *in, int in_ld, float *V, int N, int H, int W, int C, int dil, int upsample2x, float v_scale, bool split,
                        void *range_flag, float range_limit, arseg_stream_t stream) {
    if (!(v_scale > 0.0f)) return ARSEG_EINVAL;
    ARSEG_CHECK_PTR(in); ARSEG_CHECK_PTR(V); ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(H); ARSEG_CHECK_POS(W); ARSEG_CHECK_POS(C); ARSEG_CHECK_POS(dil);
    if (in_ld < C) return ARSEG_EINVAL;
    if (upsample2x && (dil != 1 || (H & 1) || (W & 1))) return ARSEG_EUNSUPPORTED;
    const long long T = arseg_wino43_tiles(N, H, W, dil);
    if (T * C >= (1ll << 31)) return ARSEG_EUNSUPPORTED;
    const WinoGeom g = make_geom(N, H, W, dil);
    const bool vec2 = !(C & 1) && !(in_ld & 1) && !(reinterpret_cast<uintptr_t>(in) & 7) && !(reinterpret_cast<uintptr_t>(V) & 7);
    unsigned *rf = reinterpret_cast<unsigned *>(range_flag);
    const float rl = range_limit > 0.0f ? range_limit : 65504.0f;
    hipStream_t hs = arseg_stream(stream);
    if (split) {
        if (!vec2 || (C & 31) || !ARSEG_ALIGNED16(V) || (reinterpret_cast<uintptr_t>(range_flag) & 3)) return ARSEG_EINVAL;
        if (upsample2x) hipLaunchKernelGGL((wino43_input_up2_kernel<f32x2, true>), dim3(grid_for((long long)g.T * C / 2)), dim3(256), 0, hs, in, in_ld, V, C, g, v_scale, rf, rl);
        else hipLaunchKernelGGL((wino43_input_kernel<f32x2, true>), dim3(grid_for((long long)g.T * C / 2)), dim3(256), 0, hs, in, in_ld, V, C, g, v_scale, rf, rl);
        return arseg_launch_status();
    }
    if (upsample2x) {
        if (vec2) hipLaunchKernelGGL((wino43_input_up2_kernel<f32x2>), dim3(grid_for((long long)g.T * C / 2)), dim3(256), 0, hs, in, in_ld, V, C, g, v_scale, rf, rl);
        else hipLaunchKernelGGL((wino43_input_up2_kernel<float>), dim3(grid_for((long long)g.T * C)), dim3(256), 0, hs, in, in_ld, V, C, g, v_scale, rf, rl);
        return arseg_launch_status();
    }
    if (vec2)
        hipLaunchKernelGGL((wino43_input_kernel<f32x2>), dim3(grid_for((long long)g.T * C / 2)), dim3(256), 0, hs, in, in_ld, V, C, g, v_scale, rf, rl);
    else
        hipLaunchKernelGGL((wino43_input_kernel<float>), dim3(grid_for((long long)g.T * C)), dim3(256), 0, hs, in, in_ld, V, C, g, v_scale, rf, rl);
    return arseg_launch_status();
}

extern "C" int arseg_wino43_input_fwd(const float *in, int in_ld, float *V, int N, int H, int W, int C, int dil, int upsample2x,
                                      float v_scale, arseg_stream_t stream) {
    return wino43_input(in, in_ld, V, N, H, W, C, dil, upsample2x, v_scale, false, nullptr, 0.0f, stream);
}

extern "C" int arseg_wino43_input_split_fwd(const float *in, int in_ld, void *V_split, int N, int H, int W, int C, int dil, int upsample2x,
                                            float v_scale, void *range_flag, float range_limit, arseg_stream_t stream) {
    return wino43_input(in, in_ld, reinterpret_cast<float *>(V_split), N, H, W, C, dil, upsample2x, v_scale, true, range_flag, range_limit, stream);
}

extern "C" int arseg_wino43_output_fwd(const float *M, const float *scale, const float *bias, const float *residual, int res_ld, float *out,
                                       int out_ld, int N, int H, int W, int Cout, int dil, int act, float prelu_slope, float m_scale, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(M); ARSEG_CHECK_PTR(out); ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(H); ARSEG_CHECK_POS(W); ARSEG_CHECK_POS(Cout); ARSEG_CHECK_POS(dil);
    if (out_ld < Cout || (residual && res_ld < Cout)) return ARSEG_EINVAL;
    const long long T = arseg_wino43_tiles(N, H, W, dil);
    if (T * Cout >= (1ll << 31)) return ARSEG_EUNSUPPORTED;
    const WinoGeom g = make_geom(N, H, W, dil);
    auto al8 = [](const void *p) { return !(reinterpret_cast<uintptr_t>(p) & 7); };
    if (!(Cout & 1) && !(out_ld & 1) && !(res_ld & 1) && al8(M) && al8(out) && al8(scale) && al8(bias) && al8(residual))
        hipLaunchKernelGGL(wino43_output_kernel<f32x2>, dim3(grid_for((long long)g.T * Cout / 2)), dim3(256), 0, arseg_stream(stream), M, scale,
                           bias, residual, res_ld, out, out_ld, Cout, act, prelu_slope, g, m_scale);
    else
        hipLaunchKernelGGL(wino43_output_kernel<float>, dim3(grid_for((long long)g.T * Cout)), dim3(256), 0, arseg_stream(stream), M, scale,
                           bias, residual, res_ld, out, out_ld, Cout, act, prelu_slope, g, m_scale);
    return arseg_launch_status();
}

// U[k][co][ci] = (G g G^T)[k], k = i*6 + j; computed in double, stored fp32.  out: [36][Cout][Cin].
extern "C" int arseg_wino43_pack_weight_host(const float *w, int Cout, int Cin, float *out) {
    if (!w || !out || Cout <= 0 || Cin <= 0) return ARSEG_EINVAL;
    static const double G[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci) {
            const float *g = w + ((size_t)co * Cin + ci) * 9;
            double tmp[6][3];
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 3; ++j) tmp[i][j] = G[i][0] * g[0 * 3 + j] + G[i][1] * g[1 * 3 + j] + G[i][2] * g[2 * 3 + j];
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j) {
                    const double u = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
                    out[((size_t)(i * 6 + j) * Cout + co) * Cin + ci] = (float)u;
                }
        }
    return ARSEG_OK;
}
