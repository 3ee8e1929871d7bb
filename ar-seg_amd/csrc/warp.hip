// Motion-vector guided warp of the keyframe feature (evaluation.py:61-87) and the MV resize block
// (evaluation.py:176-180).  Pure gather, HBM/L2 bound: one thread per (pixel, 4-channel group) for
// NHWC (16-byte coalesced loads along channels), one thread per pixel looping channels for NCHW.
//
// The sampling position is reproduced operation by operation:
//   vgrid = (float)x + flow                (fp64 when flow is fp64: float32 + float64 promotes)
//   g     = 2.0 * vgrid / max(W-1,1) - 1.0 (evaluation.py:80-81), cast to fp32 (:83)
//   ix    = ((g + 1) * W - 1) / 2          (grid_sample, align_corners=False default, fp32)
//   4-tap bilinear, taps outside the image contribute zero (padding_mode='zeros').
// Note zero motion is NOT the identity: ix = x*W/(W-1) - 0.5.
#include "warp_math.h"

namespace {

__device__ __forceinline__ f32x4 gather4(const float *img, const Taps &t, int W, int ld, int c) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float *p = img + ((long long)t.y0 * W + t.x0) * ld + c;
    if (t.vy0 && t.vx0) acc += *reinterpret_cast<const f32x4 *>(p) * t.wnw;
    if (t.vy0 && t.vx1) acc += *reinterpret_cast<const f32x4 *>(p + ld) * t.wne;
    if (t.vy1 && t.vx0) acc += *reinterpret_cast<const f32x4 *>(p + (size_t)W * ld) * t.wsw;
    if (t.vy1 && t.vx1) acc += *reinterpret_cast<const f32x4 *>(p + (size_t)W * ld + ld) * t.wse;
    return acc;
}

// output address of channels [c, c+4) of pixel `pix` (= (n*H + y)*W + x): NHWC or channel-blocked C8
__device__ __forceinline__ size_t out_addr(long long pix, int c, int C, long long HW, int c8) {
    if (!c8) return (size_t)pix * C + c;
    const long long n = pix / HW, hw = pix - n * HW;
    return (((size_t)n * (C >> 3) + (c >> 3)) * HW + hw) * 8 + (c & 4);
}

template <typename FT>
__global__ __launch_bounds__(256) void warp_nhwc_kernel(const float *__restrict__ feat, const FT *__restrict__ flow,
                                                        float *__restrict__ out, int N, int C, int H, int W, int c8) {
    const int c4n = C >> 2;
    const long long total = (long long)N * H * W * c4n;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) * 4;
        const long long pix = idx / c4n;
        const int x = (int)(pix % W), y = (int)((pix / W) % H), n = (int)(pix / ((long long)W * H));
        float gx, gy;
        norm_grid<FT>(x, y, flow[pix * 2], flow[pix * 2 + 1], H, W, gx, gy);
        const Taps t = make_taps(gx, gy, H, W);
        *reinterpret_cast<f32x4 *>(out + out_addr(pix, c, C, (long long)H * W, c8)) = gather4(feat + (size_t)n * H * W * C, t, W, C, c);
    }
}

template <typename FT>
__global__ __launch_bounds__(256) void warp_nchw_kernel(const float *__restrict__ feat, const FT *__restrict__ flow,
                                                        float *__restrict__ out, int N, int C, int H, int W) {
    const long long total = (long long)N * H * W;
    const size_t plane = (size_t)H * W;
    for (long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(pix % W), y = (int)((pix / W) % H), n = (int)(pix / ((long long)W * H));
        float gx, gy;
        norm_grid<FT>(x, y, flow[pix * 2], flow[pix * 2 + 1], H, W, gx, gy);
        const Taps t = make_taps(gx, gy, H, W);
        const float *src = feat + (size_t)n * C * plane + ((long long)t.y0 * W + t.x0);
        float *dst = out + (size_t)n * C * plane + (size_t)y * W + x;
        for (int c = 0; c < C; ++c, src += plane, dst += plane) {
            float acc = 0.f;
            if (t.vy0 && t.vx0) acc += src[0] * t.wnw;
            if (t.vy0 && t.vx1) acc += src[1] * t.wne;
            if (t.vy1 && t.vx0) acc += src[W] * t.wsw;
            if (t.vy1 && t.vx1) acc += src[W + 1] * t.wse;
            *dst = acc;
        }
    }
}

// the same block for a float flow field in pixels (any values, fp32 or fp64 input; the reference's tensor is fp64): both components
// scaled by Hp/H, bilinear(align_corners=True), arithmetic in fp64
template <typename FT>
__global__ __launch_bounds__(256) void flow_resize_kernel(const FT *__restrict__ flow, double *__restrict__ out, int N, int H, int W, int Hp, int Wp) {
    const long long total = (long long)N * Hp * Wp;
    const double sy = Hp > 1 ? (double)(H - 1) / (double)(Hp - 1) : 0.0, sx = Wp > 1 ? (double)(W - 1) / (double)(Wp - 1) : 0.0;
    for (long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(pix % Wp), y = (int)((pix / Wp) % Hp), n = (int)(pix / ((long long)Wp * Hp));
        const FT *f = flow + (size_t)n * H * W * 2;
        const double ry = sy * y, rx = sx * x;
        int y0 = min((int)ry, H - 1), x0 = min((int)rx, W - 1);
        const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
        const double ly = fmin(fmax(ry - y0, 0.0), 1.0), lx = fmin(fmax(rx - x0, 0.0), 1.0);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            auto val = [&](int yy, int xx) { return (double)f[((size_t)yy * W + xx) * 2 + k] * (double)Hp / (double)H; };   // flow * Hp / H, left to right
            out[pix * 2 + k] = (1.0 - ly) * ((1.0 - lx) * val(y0, x0) + lx * val(y0, x1)) + ly * ((1.0 - lx) * val(y1, x0) + lx * val(y1, x1));
        }
    }
}

__global__ __launch_bounds__(256) void mv_resize_kernel(const int16_t *__restrict__ mv, double *__restrict__ out, int N,
                                                        int H, int W, int Hp, int Wp) {
    const long long total = (long long)N * Hp * Wp;
    for (long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(pix % Wp), y = (int)((pix / Wp) % Hp), n = (int)(pix / ((long long)Wp * Hp));
        double fx, fy;
        mv_at(mv + (size_t)n * H * W * 2, H, W, Hp, Wp, y, x, fx, fy);
        out[pix * 2] = fx; out[pix * 2 + 1] = fy;
    }
}

// grid = (ceil(Wp/64), Hp, N); a block covers 64 pixels of one row.  Phase 1: one lane per pixel does the fp64 coordinate
// arithmetic (two divisions) and the tap setup and parks the result in LDS -- done by 16 lanes per pixel it would occupy
// whole waves with 4 useful lanes.  Phase 2: 16 lanes per pixel walk the channel vectors, four pixels groups per thread;
// the four taps of a pixel are read as whole contiguous pixels (C*4 bytes each), branch-free: taps outside the image get
// weight 0 and a clamped (in-range) address.
__global__ __launch_bounds__(256) void warp_mvq_nhwc_kernel(const float *__restrict__ feat, const int16_t *__restrict__ mv,
                                                            float *__restrict__ out, int N, int C, int Hp, int Wp, int H, int W, int c8) {
    __shared__ int s_off[4][64];
    __shared__ float s_w[4][64];
    const int tid = threadIdx.x, y = blockIdx.y, n = blockIdx.z, xb = blockIdx.x * 64;
    if (tid < 64) {
        const int x = min(xb + tid, Wp - 1);                                  // clamped: surplus lanes redo the last pixel
        double fx, fy;
        if (Hp == H && Wp == W) {              // identity resize (PSPNet): (q/4 * Hp) / H == q/4 exactly
            const int16_t *m = mv + ((size_t)n * H * W + (size_t)y * W + x) * 2;
            fx = (double)m[0] / 4.0; fy = (double)m[1] / 4.0;
        } else {
            mv_at(mv + (size_t)n * H * W * 2, H, W, Hp, Wp, y, x, fx, fy);
        }
        float gx, gy;
        norm_grid<double>(x, y, fx, fy, Hp, Wp, gx, gy);
        const Taps t = make_taps(gx, gy, Hp, Wp);
        const int xa = min(max(t.x0, 0), Wp - 1), xc = min(max(t.x0 + 1, 0), Wp - 1);
        const int ya = min(max(t.y0, 0), Hp - 1), yc = min(max(t.y0 + 1, 0), Hp - 1);
        s_off[0][tid] = ya * Wp + xa; s_off[1][tid] = ya * Wp + xc; s_off[2][tid] = yc * Wp + xa; s_off[3][tid] = yc * Wp + xc;
        s_w[0][tid] = t.vy0 && t.vx0 ? t.wnw : 0.f; s_w[1][tid] = t.vy0 && t.vx1 ? t.wne : 0.f;
        s_w[2][tid] = t.vy1 && t.vx0 ? t.wsw : 0.f; s_w[3][tid] = t.vy1 && t.vx1 ? t.wse : 0.f;
    }
    __syncthreads();
    const int sub = tid & 15;
    const float *img = feat + (size_t)n * Hp * Wp * C;
    const int hw = Hp * Wp;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int pl = it * 16 + (tid >> 4);
        const int x = min(xb + pl, Wp - 1), pix = y * Wp + x;
        const int o0 = s_off[0][pl], o1 = s_off[1][pl], o2 = s_off[2][pl], o3 = s_off[3][pl];
        const float w0 = s_w[0][pl], w1 = s_w[1][pl], w2 = s_w[2][pl], w3 = s_w[3][pl];
        for (int c = sub * 4; c < C; c += 64) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc += *reinterpret_cast<const f32x4 *>(img + (size_t)o0 * C + c) * w0;
            acc += *reinterpret_cast<const f32x4 *>(img + (size_t)o1 * C + c) * w1;
            acc += *reinterpret_cast<const f32x4 *>(img + (size_t)o2 * C + c) * w2;
            acc += *reinterpret_cast<const f32x4 *>(img + (size_t)o3 * C + c) * w3;
            const size_t o = c8 ? ((((size_t)n * (C >> 3) + (c >> 3)) * hw + pix) * 8 + (c & 4)) : (((size_t)n * hw + pix) * C + c);
            *reinterpret_cast<f32x4 *>(out + o) = acc;      // (duplicate lanes of a clamped pixel store identical values)
        }
    }
}

inline int grid_for(long long total) {
    long long b = (total + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int arseg_warp_fwd(const float *feature, const void *flow, int flow_dtype, float *out, int N, int C, int H,
                              int W, int layout, int out_layout, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(feature); ARSEG_CHECK_PTR(flow); ARSEG_CHECK_PTR(out);
    ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(C); ARSEG_CHECK_POS(H); ARSEG_CHECK_POS(W);
    if (flow_dtype != ARSEG_FLOW_F32 && flow_dtype != ARSEG_FLOW_F64) return ARSEG_EINVAL;
    hipStream_t st = arseg_stream(stream);
    if (layout == ARSEG_NHWC) {
        if ((C & 3) || !ARSEG_ALIGNED16(feature) || !ARSEG_ALIGNED16(out)) return ARSEG_EINVAL;
        if (out_layout != ARSEG_NHWC && out_layout != ARSEG_C8) return ARSEG_EINVAL;
        const int c8 = out_layout == ARSEG_C8;
        if (c8 && (C & 7)) return ARSEG_EINVAL;
        const int g = grid_for((long long)N * H * W * (C >> 2));
        if (flow_dtype == ARSEG_FLOW_F64)
            hipLaunchKernelGGL(warp_nhwc_kernel<double>, dim3(g), dim3(256), 0, st, feature, (const double *)flow, out, N, C, H, W, c8);
        else
            hipLaunchKernelGGL(warp_nhwc_kernel<float>, dim3(g), dim3(256), 0, st, feature, (const float *)flow, out, N, C, H, W, c8);
    } else if (layout == ARSEG_NCHW) {
        if (out_layout != ARSEG_NCHW) return ARSEG_EINVAL;
        const int g = grid_for((long long)N * H * W);
        if (flow_dtype == ARSEG_FLOW_F64)
            hipLaunchKernelGGL(warp_nchw_kernel<double>, dim3(g), dim3(256), 0, st, feature, (const double *)flow, out, N, C, H, W);
        else
            hipLaunchKernelGGL(warp_nchw_kernel<float>, dim3(g), dim3(256), 0, st, feature, (const float *)flow, out, N, C, H, W);
    } else return ARSEG_EINVAL;
    return arseg_launch_status();
}

extern "C" int arseg_mv_resize_fwd(const int16_t *mv_q, double *out, int N, int H, int W, int Hp, int Wp,
                                   arseg_stream_t stream) {
    ARSEG_CHECK_PTR(mv_q); ARSEG_CHECK_PTR(out);
    ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(H); ARSEG_CHECK_POS(W); ARSEG_CHECK_POS(Hp); ARSEG_CHECK_POS(Wp);
    hipLaunchKernelGGL(mv_resize_kernel, dim3(grid_for((long long)N * Hp * Wp)), dim3(256), 0, arseg_stream(stream), mv_q,
                       out, N, H, W, Hp, Wp);
    return arseg_launch_status();
}

extern "C" int arseg_flow_resize_fwd(const void *flow, int flow_dtype, double *out, int N, int H, int W, int Hp, int Wp,
                                     arseg_stream_t stream) {
    ARSEG_CHECK_PTR(flow); ARSEG_CHECK_PTR(out);
    ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(H); ARSEG_CHECK_POS(W); ARSEG_CHECK_POS(Hp); ARSEG_CHECK_POS(Wp);
    const int g = grid_for((long long)N * Hp * Wp);
    if (flow_dtype == ARSEG_FLOW_F64)
        hipLaunchKernelGGL(flow_resize_kernel<double>, dim3(g), dim3(256), 0, arseg_stream(stream), (const double *)flow, out, N, H, W, Hp, Wp);
    else if (flow_dtype == ARSEG_FLOW_F32)
        hipLaunchKernelGGL(flow_resize_kernel<float>, dim3(g), dim3(256), 0, arseg_stream(stream), (const float *)flow, out, N, H, W, Hp, Wp);
    else return ARSEG_EINVAL;
    return arseg_launch_status();
}

extern "C" int arseg_warp_mvq_fwd(const float *feature, const int16_t *mv_q, float *out, int N, int C, int Hp, int Wp,
                                  int H, int W, int out_layout, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(feature); ARSEG_CHECK_PTR(mv_q); ARSEG_CHECK_PTR(out);
    ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(C); ARSEG_CHECK_POS(H); ARSEG_CHECK_POS(W); ARSEG_CHECK_POS(Hp); ARSEG_CHECK_POS(Wp);
    if ((C & 3) || !ARSEG_ALIGNED16(feature) || !ARSEG_ALIGNED16(out)) return ARSEG_EINVAL;
    if (out_layout != ARSEG_NHWC && out_layout != ARSEG_C8) return ARSEG_EINVAL;
    if (out_layout == ARSEG_C8 && (C & 7)) return ARSEG_EINVAL;
    if (Hp > 65535 || N > 65535) return ARSEG_EUNSUPPORTED;
    hipLaunchKernelGGL(warp_mvq_nhwc_kernel, dim3(arseg_cdiv(Wp, 64), Hp, N), dim3(256), 0, arseg_stream(stream), feature, mv_q, out, N, C,
                       Hp, Wp, H, W, out_layout == ARSEG_C8 ? 1 : 0);
    return arseg_launch_status();
}
