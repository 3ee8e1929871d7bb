// Sampling arithmetic of warpFeature (evaluation.py:61-87) and of the MV resize block (evaluation.py:176-180),
// shared by warp.hip (stand-alone warp kernels) and creff_rr.hip (warp fused into the CReFF tile staging).
//
// The sampling position is reproduced operation by operation:
//   vgrid = (float)x + flow                (fp64 when flow is fp64: float32 + float64 promotes)
//   g     = 2.0 * vgrid / max(W-1,1) - 1.0 (evaluation.py:80-81), cast to fp32 (:83)
//   ix    = ((g + 1) * W - 1) / 2          (grid_sample, align_corners=False default, fp32)
//   4-tap bilinear, taps outside the image contribute zero (padding_mode='zeros').
// Note zero motion is NOT the identity: ix = x*W/(W-1) - 0.5.
#pragma once
#include "arseg_common.h"

struct Taps { int x0, y0; float ex, wx, ey, wy; float wnw, wne, wsw, wse; bool vx0, vx1, vy0, vy1; };

__device__ __forceinline__ Taps make_taps(float gx, float gy, int H, int W) {
    const float ix = ((gx + 1.0f) * (float)W - 1.0f) * 0.5f;
    const float iy = ((gy + 1.0f) * (float)H - 1.0f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    Taps t;
    // clamp far-out-of-range positions before the int conversion (all taps invalid there anyway)
    t.x0 = (int)fminf(fmaxf(fx, -2.0f), (float)W);
    t.y0 = (int)fminf(fmaxf(fy, -2.0f), (float)H);
    t.ex = fx + 1.0f - ix; t.ey = fy + 1.0f - iy;           // distances to the east / south taps
    t.wx = ix - fx; t.wy = iy - fy;
    t.wnw = t.ex * t.ey; t.wne = t.wx * t.ey; t.wsw = t.ex * t.wy; t.wse = t.wx * t.wy;
    t.vx0 = (unsigned)t.x0 < (unsigned)W; t.vx1 = (unsigned)(t.x0 + 1) < (unsigned)W;
    t.vy0 = (unsigned)t.y0 < (unsigned)H; t.vy1 = (unsigned)(t.y0 + 1) < (unsigned)H;
    return t;
}

template <typename FT>
__device__ __forceinline__ void norm_grid(int x, int y, FT fx, FT fy, int H, int W, float &gx, float &gy) {
    const FT vx = (FT)(float)x + fx, vy = (FT)(float)y + fy;
    gx = (float)((FT)2.0 * vx / (FT)max(W - 1, 1) - (FT)1.0);
    gy = (float)((FT)2.0 * vy / (FT)max(H - 1, 1) - (FT)1.0);
}

// The same with the two divisions by the (uniform) extents replaced by a*y + one correction step, y = RN(1/b) computed once per
// kernel with a true division: q0 = RN(a*y), r = a - b*q0 (exact in an fma), q = RN(q0 + r*y) is the correctly rounded quotient a/b
// (Markstein) -- the values match norm_grid<double>, the 2 x 12 slow fp64 instructions of the division expansion become 2 x 3.
__device__ __forceinline__ double div_by_rcp(double a, double b, double y) {
    const double q0 = a * y;
    return __builtin_fma(__builtin_fma(-q0, b, a), y, q0);
}
__device__ __forceinline__ void norm_grid_rcp(int x, int y, double fx, double fy, double dW, double dH, double rW, double rH, float &gx, float &gy) {
    const double vx = (double)(float)x + fx, vy = (double)(float)y + fy;
    gx = (float)(div_by_rcp(2.0 * vx, dW, rW) - 1.0);
    gy = (float)(div_by_rcp(2.0 * vy, dH, rH) - 1.0);
}

// (mv_q/4) * Hp/H resampled to (Hp,Wp) with align_corners=True, in fp64 like the reference
__device__ __forceinline__ void mv_at(const int16_t *__restrict__ mv, int H, int W, int Hp, int Wp, int y, int x,
                                      double &fx, double &fy) {
    const double sy = Hp > 1 ? (double)(H - 1) / (double)(Hp - 1) : 0.0;
    const double sx = Wp > 1 ? (double)(W - 1) / (double)(Wp - 1) : 0.0;
    const double ry = sy * y, rx = sx * x;
    int y0 = (int)ry, x0 = (int)rx;
    y0 = min(y0, H - 1); x0 = min(x0, W - 1);
    const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
    const double ly = fmin(fmax(ry - y0, 0.0), 1.0), lx = fmin(fmax(rx - x0, 0.0), 1.0);
    double v[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        auto val = [&](int yy, int xx) { return (double)mv[((size_t)yy * W + xx) * 2 + k] / 4.0 * (double)Hp / (double)H; };
        v[k] = (1.0 - ly) * ((1.0 - lx) * val(y0, x0) + lx * val(y0, x1)) + ly * ((1.0 - lx) * val(y1, x0) + lx * val(y1, x1));
    }
    fx = v[0]; fy = v[1];
}
