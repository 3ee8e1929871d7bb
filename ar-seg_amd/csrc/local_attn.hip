// Stand-alone replacements for the two forward functions of the third-party `localAttention`
// CUDA extension the reference imports (model/attention.py:7-11; wrappers :13-53):
//   similar_forward(x_ori, x_loc, kH, kW)    -> [N,H,W,kH*kW]
//   weighting_forward(x_ori, x_weight, kH, kW) -> [N,C,H,W]
// NCHW fp32 like the original.  Each 256-thread block owns an 8x32 pixel tile; the (tile + halo)
// window of 4 channel planes of K / V is staged in LDS per step so every global element is read
// once per tile (the original re-reads each K/V element kH*kW times), the kH*kW scores / weights
// of a pixel live in registers.  Windows 3x3, 5x5, 7x7 are compiled; other odd sizes take a
// generic one-thread-per-(pixel,tap) path.  The fused CReFF kernel (creff.hip) is the fast path;
// this pair exists so that code written against `localAttention` keeps working.
#include "arseg_common.h"

namespace {

constexpr int TH = 8, TW = 32, CH = 4;

// element (n, c, y, x) of a feature tensor: NCHW planes, or NHWC rows with channel stride ld
template <bool NHWC>
__device__ __forceinline__ size_t feat_idx(int n, int c, int y, int x, int C, int H, int W, int ld) {
    return NHWC ? (((size_t)n * H + y) * W + x) * ld + c : ((size_t)n * C + c) * ((size_t)H * W) + (size_t)y * W + x;
}

template <int KS, bool NHWC>
__global__ __launch_bounds__(256) void similar_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                      float *__restrict__ s, int C, int H, int W, int ld) {
    constexpr int R = KS / 2, LH = TH + KS - 1, LW = TW + KS - 1;
    __shared__ float kt[CH][LH][LW + 1];
    const int n = blockIdx.z, y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int y = y0 + ty, x = x0 + tx;
    const bool inside = y < H && x < W;
    float acc[KS * KS];
#pragma unroll
    for (int i = 0; i < KS * KS; ++i) acc[i] = 0.f;
    for (int c0 = 0; c0 < C; c0 += CH) {
        __syncthreads();
        for (int i = threadIdx.x; i < CH * LH * LW; i += 256) {
            const int cc = i / (LH * LW), r = (i / LW) % LH, col = i % LW;
            const int yy = y0 + r - R, xx = x0 + col - R, c = c0 + cc;
            float v = 0.f;
            if (c < C && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) v = k[feat_idx<NHWC>(n, c, yy, xx, C, H, W, ld)];
            kt[cc][r][col] = v;
        }
        __syncthreads();
        if (inside) {
#pragma unroll
            for (int cc = 0; cc < CH; ++cc) {
                if (c0 + cc >= C) break;
                const float qv = q[feat_idx<NHWC>(n, c0 + cc, y, x, C, H, W, ld)];
#pragma unroll
                for (int dy = 0; dy < KS; ++dy)
#pragma unroll
                    for (int dx = 0; dx < KS; ++dx) acc[dy * KS + dx] += qv * kt[cc][ty + dy][tx + dx];
            }
        }
    }
    if (inside) {
        float *o = s + (((size_t)n * H + y) * W + x) * (KS * KS);
#pragma unroll
        for (int i = 0; i < KS * KS; ++i) o[i] = acc[i];
    }
}

template <int KS, bool NHWC>
__global__ __launch_bounds__(256) void weighting_kernel(const float *__restrict__ v, const float *__restrict__ w,
                                                        float *__restrict__ o, int C, int H, int W, int ld) {
    constexpr int R = KS / 2, LH = TH + KS - 1, LW = TW + KS - 1;
    __shared__ float vt[CH][LH][LW + 1];
    const int n = blockIdx.z, y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int y = y0 + ty, x = x0 + tx;
    const bool inside = y < H && x < W;
    float wt[KS * KS];
    if (inside) {
        const float *wp = w + (((size_t)n * H + y) * W + x) * (KS * KS);
#pragma unroll
        for (int i = 0; i < KS * KS; ++i) wt[i] = wp[i];
    }
    for (int c0 = 0; c0 < C; c0 += CH) {
        __syncthreads();
        for (int i = threadIdx.x; i < CH * LH * LW; i += 256) {
            const int cc = i / (LH * LW), r = (i / LW) % LH, col = i % LW;
            const int yy = y0 + r - R, xx = x0 + col - R, c = c0 + cc;
            float val = 0.f;
            if (c < C && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) val = v[feat_idx<NHWC>(n, c, yy, xx, C, H, W, ld)];
            vt[cc][r][col] = val;
        }
        __syncthreads();
        if (inside) {
#pragma unroll
            for (int cc = 0; cc < CH; ++cc) {
                if (c0 + cc >= C) break;
                float acc = 0.f;
#pragma unroll
                for (int dy = 0; dy < KS; ++dy)
#pragma unroll
                    for (int dx = 0; dx < KS; ++dx) acc += vt[cc][ty + dy][tx + dx] * wt[dy * KS + dx];
                o[feat_idx<NHWC>(n, c0 + cc, y, x, C, H, W, ld)] = acc;
            }
        }
    }
}

// generic window sizes: one thread per (pixel, tap) / per output element, straight from global memory
template <bool NHWC>
__global__ __launch_bounds__(256) void similar_generic_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                              float *__restrict__ s, int N, int C, int H, int W, int kH, int kW, int ld) {
    const int T = kH * kW;
    const long long total = (long long)N * H * W * T;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(idx % T);
        const long long pix = idx / T;
        const int x = (int)(pix % W), y = (int)((pix / W) % H), n = (int)(pix / ((long long)W * H));
        const int yy = y + t / kW - kH / 2, xx = x + t % kW - kW / 2;
        float acc = 0.f;
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
            for (int c = 0; c < C; ++c) acc += q[feat_idx<NHWC>(n, c, y, x, C, H, W, ld)] * k[feat_idx<NHWC>(n, c, yy, xx, C, H, W, ld)];
        }
        s[idx] = acc;
    }
}

template <bool NHWC>
__global__ __launch_bounds__(256) void weighting_generic_kernel(const float *__restrict__ v, const float *__restrict__ w,
                                                                float *__restrict__ o, int N, int C, int H, int W, int kH, int kW, int ld) {
    const long long total = (long long)N * C * H * W;
    const int T = kH * kW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        // (n, c, y, x) with x fastest (NCHW) or c fastest (NHWC): consecutive threads touch consecutive output elements either way
        int x, y, c, n;
        if (NHWC) { c = (int)(idx % C); x = (int)((idx / C) % W); y = (int)((idx / ((long long)C * W)) % H); n = (int)(idx / ((long long)C * W * H)); }
        else { x = (int)(idx % W); y = (int)((idx / W) % H); c = (int)((idx / ((long long)W * H)) % C); n = (int)(idx / ((long long)W * H * C)); }
        const float *wp = w + (((size_t)n * H + y) * W + x) * T;
        float acc = 0.f;
        for (int dy = 0; dy < kH; ++dy)
            for (int dx = 0; dx < kW; ++dx) {
                const int yy = y + dy - kH / 2, xx = x + dx - kW / 2;
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) acc += v[feat_idx<NHWC>(n, c, yy, xx, C, H, W, ld)] * wp[dy * kW + dx];
            }
        o[feat_idx<NHWC>(n, c, y, x, C, H, W, ld)] = acc;
    }
}

int check(const void *a, const void *b, const void *c, int N, int C, int H, int W, int kH, int kW) {
    if (!a || !b || !c) return ARSEG_EINVAL;
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || kH <= 0 || kW <= 0) return ARSEG_EINVAL;
    if (!(kH & 1) || !(kW & 1)) return ARSEG_EUNSUPPORTED;
    if (kH * kW > 121 || N > 65535) return ARSEG_EUNSUPPORTED;
    return ARSEG_OK;
}

}  // namespace

namespace {
template <bool NHWC>
int similar_launch(const float *q, const float *k, float *s, int N, int C, int H, int W, int kH, int kW, int ld, hipStream_t hs) {
    dim3 grid(arseg_cdiv(W, TW), arseg_cdiv(H, TH), N);
    if (kH == kW && kH == 7) hipLaunchKernelGGL((similar_kernel<7, NHWC>), grid, dim3(256), 0, hs, q, k, s, C, H, W, ld);
    else if (kH == kW && kH == 5) hipLaunchKernelGGL((similar_kernel<5, NHWC>), grid, dim3(256), 0, hs, q, k, s, C, H, W, ld);
    else if (kH == kW && kH == 3) hipLaunchKernelGGL((similar_kernel<3, NHWC>), grid, dim3(256), 0, hs, q, k, s, C, H, W, ld);
    else {
        long long b = ((long long)N * H * W * kH * kW + 255) / 256;
        hipLaunchKernelGGL(similar_generic_kernel<NHWC>, dim3((int)(b > 16384 ? 16384 : b)), dim3(256), 0, hs, q, k, s, N, C, H, W, kH, kW, ld);
    }
    return arseg_launch_status();
}
template <bool NHWC>
int weighting_launch(const float *v, const float *w, float *o, int N, int C, int H, int W, int kH, int kW, int ld, hipStream_t hs) {
    dim3 grid(arseg_cdiv(W, TW), arseg_cdiv(H, TH), N);
    if (kH == kW && kH == 7) hipLaunchKernelGGL((weighting_kernel<7, NHWC>), grid, dim3(256), 0, hs, v, w, o, C, H, W, ld);
    else if (kH == kW && kH == 5) hipLaunchKernelGGL((weighting_kernel<5, NHWC>), grid, dim3(256), 0, hs, v, w, o, C, H, W, ld);
    else if (kH == kW && kH == 3) hipLaunchKernelGGL((weighting_kernel<3, NHWC>), grid, dim3(256), 0, hs, v, w, o, C, H, W, ld);
    else {
        long long b = ((long long)N * C * H * W + 255) / 256;
        hipLaunchKernelGGL(weighting_generic_kernel<NHWC>, dim3((int)(b > 16384 ? 16384 : b)), dim3(256), 0, hs, v, w, o, N, C, H, W, kH, kW, ld);
    }
    return arseg_launch_status();
}
}  // namespace

extern "C" int arseg_local_similar_fwd(const float *q, const float *k, float *s, int N, int C, int H, int W, int kH, int kW,
                                       arseg_stream_t stream) {
    int st = check(q, k, s, N, C, H, W, kH, kW);
    if (st != ARSEG_OK) return st;
    return similar_launch<false>(q, k, s, N, C, H, W, kH, kW, 0, arseg_stream(stream));
}

extern "C" int arseg_local_weighting_fwd(const float *v, const float *w, float *o, int N, int C, int H, int W, int kH, int kW,
                                         arseg_stream_t stream) {
    int st = check(v, w, o, N, C, H, W, kH, kW);
    if (st != ARSEG_OK) return st;
    return weighting_launch<false>(v, w, o, N, C, H, W, kH, kW, 0, arseg_stream(stream));
}

// The same pair on NHWC (channels_last) features with channel stride ld >= C (both inputs / the output share it): no layout change
// between the backbone's NHWC tensors and the op.  Scores and weights keep the [N,H,W,kH*kW] layout.
extern "C" int arseg_local_similar_nhwc_fwd(const float *q, const float *k, int ld, float *s, int N, int C, int H, int W, int kH, int kW,
                                            arseg_stream_t stream) {
    int st = check(q, k, s, N, C, H, W, kH, kW);
    if (st != ARSEG_OK) return st;
    if (ld < C) return ARSEG_EINVAL;
    return similar_launch<true>(q, k, s, N, C, H, W, kH, kW, ld, arseg_stream(stream));
}

extern "C" int arseg_local_weighting_nhwc_fwd(const float *v, const float *w, int ld, float *o, int N, int C, int H, int W, int kH, int kW,
                                              arseg_stream_t stream) {
    int st = check(v, w, o, N, C, H, W, kH, kW);
    if (st != ARSEG_OK) return st;
    if (ld < C) return ARSEG_EINVAL;
    return weighting_launch<true>(v, w, o, N, C, H, W, kH, kW, ld, arseg_stream(stream));
}
