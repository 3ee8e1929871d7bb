// The small NHWC layers around the conv engine: pooling, global reductions, resizes (every
// align_corners / nearest flavour the reference uses), ARM/FFM channel scaling, the 1x1 classifier
// head, frame ingest (downscale + NCHW -> NHWC4), layout changes and the evaluator tail.
// All of them are HBM/L2-bound element-wise or small-window kernels: 16-byte channel vectors per
// lane (coalesced along C), grid-stride loops, no MFMA.
#include "arseg_common.h"

namespace {

inline int grid_for(long long total, int cap = 8192) {
    long long b = (total + 255) / 256;
    return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

// ------------------------------------------------------------------ MaxPool2d(3, 2, 1)
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float *__restrict__ in, float *__restrict__ out, int N,
                                                           int H, int W, int C, int Ho, int Wo) {
    const int c4n = C >> 2;
    const long long total = (long long)N * Ho * Wo * c4n;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) * 4;
        const long long pix = idx / c4n;
        const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), n = (int)(pix / ((long long)Wo * Ho));
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(in + (((size_t)n * H + iy) * W + ix) * C + c);
                m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
            }
        }
        *reinterpret_cast<f32x4 *>(out + pix * C + c) = m;
    }
}

// ------------------------------------------------------------------ adaptive avg pool / global mean / global max
// block = (bin, 16-channel chunk, n); 256 threads = 64 pixel lanes x 4 channel vectors; fixed-order LDS tree over the
// pixel lanes (deterministic).  Small chunks keep many blocks in flight for the whole-image bins (1x1 pyramid level,
// global mean / max), which are pure streaming reads.
template <bool IS_MAX, int PL = 64, int CV = 4>      // PL pixel lanes x CV channel vectors (16 bytes each): 256 threads, or 1024 for launches with few blocks
__global__ __launch_bounds__(CV * PL) void window_reduce_kernel(const float *__restrict__ in, int in_ld, float *__restrict__ out,
                                                               int out_ld, long long out_n_stride, int H, int W, int C, int oh, int ow,
                                                               int zb_n = 0, int zb_self = 0) {
    __shared__ f32x4 red[PL][CV + 1];
    const int bin = blockIdx.x, by = bin / ow, bx = bin - by * ow, n = blockIdx.z;
    const int y0 = (by * H) / oh, y1 = ((by + 1) * H + oh - 1) / oh;
    const int x0 = (bx * W) / ow, x1 = ((bx + 1) * W + ow - 1) / ow;
    const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
    const int c = blockIdx.y * (CV * 4) + cv * 4;
    const int ww = x1 - x0, cnt = (y1 - y0) * ww;
    f32x4 acc = IS_MAX ? f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY} : f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        for (int i = pl; i < cnt; i += PL) {
            const int yy = y0 + i / ww, xx = x0 + i % ww;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(in + (((size_t)n * H + yy) * W + xx) * in_ld + c);
            if (IS_MAX) { acc[0] = fmaxf(acc[0], v[0]); acc[1] = fmaxf(acc[1], v[1]); acc[2] = fmaxf(acc[2], v[2]); acc[3] = fmaxf(acc[3], v[3]); }
            else acc += v;
        }
    }
    red[pl][cv] = acc;
    __syncthreads();
#pragma unroll
    for (int s = PL / 2; s >= 1; s >>= 1) {
        if (pl < s) {
            const f32x4 o = red[pl + s][cv];
            f32x4 t = red[pl][cv];
            if (IS_MAX) { t[0] = fmaxf(t[0], o[0]); t[1] = fmaxf(t[1], o[1]); t[2] = fmaxf(t[2], o[2]); t[3] = fmaxf(t[3], o[3]); }
            else t += o;
            red[pl][cv] = t;
        }
        __syncthreads();
    }
    if (pl == 0 && c < C) {
        f32x4 t = red[0][cv];
        if (!IS_MAX) t = t / (float)cnt;
        float *o = out + (size_t)n * out_n_stride + (size_t)bin * out_ld + c;
        *reinterpret_cast<f32x4 *>(o) = t;
        // block-row mode (folded PSP pyramid): the row holds zb_n column blocks of C channels, this level owns block zb_self (`out` points at
        // its first column) -- the same channels of the sibling blocks are zero, written here instead of by a fill launch
        for (int j = 0; j < zb_n; ++j)
            if (j != zb_self) *reinterpret_cast<f32x4 *>(o + (j - zb_self) * C) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// ------------------------------------------------------------------ global mean / max of large maps in two deterministic stages
// One workgroup per 16 channels and image (window_reduce_kernel with a 1 x 1 output) leaves the chip idle for a single keyframe: 16 workgroups
// read an 8 MB map (32 us in a GOP step).  Stage 1: the image is cut into `parts` row bands, a workgroup reduces one band x 64 channels into
// the workspace [N][parts][C]; stage 2 combines the parts in order.
template <bool IS_MAX>
__global__ __launch_bounds__(256) void global_parts_kernel(const float *__restrict__ in, int in_ld, float *__restrict__ ws, int HW, int C, int parts) {
    __shared__ f32x4 red[16][17];
    const int part = blockIdx.x, n = blockIdx.z, cv = threadIdx.x & 15, pl = threadIdx.x >> 4, c = blockIdx.y * 64 + cv * 4;
    const int per = (HW + parts - 1) / parts, p0 = part * per, p1 = min(p0 + per, HW);
    f32x4 acc = IS_MAX ? f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY} : f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < C)
        for (int i = p0 + pl; i < p1; i += 16) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(in + ((size_t)n * HW + i) * in_ld + c);
            if (IS_MAX) { acc[0] = fmaxf(acc[0], v[0]); acc[1] = fmaxf(acc[1], v[1]); acc[2] = fmaxf(acc[2], v[2]); acc[3] = fmaxf(acc[3], v[3]); }
            else acc += v;
        }
    red[pl][cv] = acc;
    __syncthreads();
#pragma unroll
    for (int s = 8; s >= 1; s >>= 1) {
        if (pl < s) {
            const f32x4 o = red[pl + s][cv];
            f32x4 t = red[pl][cv];
            if (IS_MAX) { t[0] = fmaxf(t[0], o[0]); t[1] = fmaxf(t[1], o[1]); t[2] = fmaxf(t[2], o[2]); t[3] = fmaxf(t[3], o[3]); }
            else t += o;
            red[pl][cv] = t;
        }
        __syncthreads();
    }
    if (pl == 0 && c < C) *reinterpret_cast<f32x4 *>(ws + ((size_t)n * parts + part) * C + c) = red[0][cv];
}
template <bool IS_MAX>
__global__ __launch_bounds__(256) void global_combine_kernel(const float *__restrict__ ws, float *__restrict__ out, int HW, int C, int parts) {
    const int n = blockIdx.y, c = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= C) return;
    f32x4 acc = *reinterpret_cast<const f32x4 *>(ws + (size_t)n * parts * C + c);
    for (int q = 1; q < parts; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(ws + ((size_t)n * parts + q) * C + c);
        if (IS_MAX) { acc[0] = fmaxf(acc[0], v[0]); acc[1] = fmaxf(acc[1], v[1]); acc[2] = fmaxf(acc[2], v[2]); acc[3] = fmaxf(acc[3], v[3]); }
        else acc += v;
    }
    if (!IS_MAX) acc = acc / (float)HW;
    *reinterpret_cast<f32x4 *>(out + (size_t)n * C + c) = acc;
}

// ------------------------------------------------------------------ pyramid pooling in one pass over the map (PSPModule, model/pspnet.py:14-31)
// The adaptive-average bins of the pyramid levels overlap within a level (H not divisible by s) and across levels, but every bin is a union of
// cells of the grid spanned by ALL bin edges of all levels (<= 25 edges per axis).  Stage 1 sums every cell (each pixel is read exactly once: the
// four per-level launches read the map four times, 184 MB instead of 46 MB for the 11-frame LR batch), stage 2 adds the cells of a bin, divides
// by its pixel count and writes the block-structured row (its level's column block, zeros in the sibling blocks).
struct PoolGrid { int ny, nx, nlev, rows; int ey[26], ex[26]; int size[4], off[4]; };

__global__ __launch_bounds__(256) void psp_cells_kernel(const float *__restrict__ in, int in_ld, float *__restrict__ cells, int H, int W, int C, PoolGrid g) {
    __shared__ f32x4 red[16][17];
    const int cell = blockIdx.x, cy = cell / g.nx, cx = cell - cy * g.nx, n = blockIdx.z;
    const int y0 = g.ey[cy], y1 = g.ey[cy + 1], x0 = g.ex[cx], x1 = g.ex[cx + 1];
    const int cv = threadIdx.x & 15, pl = threadIdx.x >> 4, c = blockIdx.y * 64 + cv * 4;
    const int ww = x1 - x0, cnt = (y1 - y0) * ww;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (c < C)
        for (int i = pl; i < cnt; i += 16) {
            const int yy = y0 + i / ww, xx = x0 + i % ww;
            acc += *reinterpret_cast<const f32x4 *>(in + (((size_t)n * H + yy) * W + xx) * in_ld + c);
        }
    red[pl][cv] = acc;
    __syncthreads();
#pragma unroll
    for (int s = 8; s >= 1; s >>= 1) {
        if (pl < s) red[pl][cv] += red[pl + s][cv];
        __syncthreads();
    }
    if (pl == 0 && c < C) *reinterpret_cast<f32x4 *>(cells + ((size_t)n * g.ny * g.nx + cell) * C + c) = red[0][cv];
}

__global__ __launch_bounds__(256) void psp_bins_kernel(const float *__restrict__ cells, float *__restrict__ out, int H, int W, int C, PoolGrid g) {
    const int row = blockIdx.x, n = blockIdx.y;
    int lev = 0;
    while (lev + 1 < g.nlev && row >= g.off[lev + 1]) ++lev;
    const int s = g.size[lev], b = row - g.off[lev], by = b / s, bx = b - by * s;
    const int y0 = (by * H) / s, y1 = ((by + 1) * H + s - 1) / s, x0 = (bx * W) / s, x1 = ((bx + 1) * W + s - 1) / s;
    const float inv = 1.0f / (float)((y1 - y0) * (x1 - x0));
    const float *cn = cells + (size_t)n * g.ny * g.nx * C;
    float *o = out + ((size_t)n * g.rows + row) * ((size_t)g.nlev * C);
    // the cells of a bin are a rectangle of the cell grid (both are cut at the same edges): no tests inside the loops, four loads in flight
    int cy0 = 0, cy1 = g.ny, cx0 = 0, cx1 = g.nx;
    while (g.ey[cy0] < y0) ++cy0;
    while (g.ey[cy1] > y1) --cy1;
    while (g.ex[cx0] < x0) ++cx0;
    while (g.ex[cx1] > x1) --cx1;
    const int nc = (cy1 - cy0) * (cx1 - cx0), cw = cx1 - cx0;
    for (int c = threadIdx.x * 4; c < C; c += 1024) {
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
        auto cell = [&](int i) { return *reinterpret_cast<const f32x4 *>(cn + (size_t)((cy0 + i / cw) * g.nx + cx0 + i % cw) * C + c); };
        int i = 0;
        for (; i + 4 <= nc; i += 4) {
            const f32x4 v0 = cell(i), v1 = cell(i + 1), v2 = cell(i + 2), v3 = cell(i + 3);
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        }
        for (; i < nc; ++i) a0 += cell(i);
        const f32x4 acc = (a0 + a1) + (a2 + a3);
        for (int j = 0; j < g.nlev; ++j) *reinterpret_cast<f32x4 *>(o + (size_t)j * C + c) = j == lev ? acc * inv : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// ------------------------------------------------------------------ resize
__global__ __launch_bounds__(256) void resize_nhwc_kernel(const float *__restrict__ in, float *__restrict__ out, int N, int C,
                                                          int Hin, int Win, int Hout, int Wout, int mode, int align, int in_ld,
                                                          int out_ld) {
    const int c4n = C >> 2;
    const long long total = (long long)N * Hout * Wout * c4n;
    const float sy = arseg_resize_scale(Hin, Hout, align), sx = arseg_resize_scale(Win, Wout, align);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) * 4;
        const long long pix = idx / c4n;
        const int ox = (int)(pix % Wout), oy = (int)((pix / Wout) % Hout), n = (int)(pix / ((long long)Wout * Hout));
        const float *base = in + (size_t)n * Hin * Win * in_ld + c;
        f32x4 v;
        if (mode == ARSEG_NEAREST) {
            const int iy = min((int)floorf((float)oy * ((float)Hin / (float)Hout)), Hin - 1);
            const int ix = min((int)floorf((float)ox * ((float)Win / (float)Wout)), Win - 1);
            v = *reinterpret_cast<const f32x4 *>(base + ((size_t)iy * Win + ix) * in_ld);
        } else {
            int y0, y1, x0, x1; float ly, lx;
            arseg_src_index(sy, oy, align, Hin, y0, y1, ly);
            arseg_src_index(sx, ox, align, Win, x0, x1, lx);
            ly = fminf(fmaxf(ly, 0.f), 1.f); lx = fminf(fmaxf(lx, 0.f), 1.f);
            const f32x4 a = *reinterpret_cast<const f32x4 *>(base + ((size_t)y0 * Win + x0) * in_ld);
            const f32x4 b = *reinterpret_cast<const f32x4 *>(base + ((size_t)y0 * Win + x1) * in_ld);
            const f32x4 cc = *reinterpret_cast<const f32x4 *>(base + ((size_t)y1 * Win + x0) * in_ld);
            const f32x4 d = *reinterpret_cast<const f32x4 *>(base + ((size_t)y1 * Win + x1) * in_ld);
            v = (1.f - ly) * ((1.f - lx) * a + lx * b) + ly * ((1.f - lx) * cc + lx * d);
        }
        *reinterpret_cast<f32x4 *>(out + pix * out_ld + c) = v;
    }
}

// Exact x2 bilinear upsample with align_corners=False (F.upsample's default: PSPUpsample, model/pspnet.py:45): the source
// offsets are the constants 0.25 / 0.75, so one thread turns a 3x3 low-resolution neighbourhood (clamped at the border, which
// reproduces ATen's src < 0 -> 0 and i1 = min(i0+1, h-1)) into the 2x2 outputs of its pixel: no per-element index divisions,
// every loaded vector feeds four outputs.  grid = (ceil(Win*C/4 / 256), Hin, N).
__global__ __launch_bounds__(256) void upsample2x_nhwc_kernel(const float *__restrict__ in, float *__restrict__ out, int C, int Hin, int Win,
                                                              int in_ld, int out_ld) {
    const int c4n = C >> 2, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Win * c4n) return;
    const int j = t / c4n, c = (t - j * c4n) * 4, i = blockIdx.y, n = blockIdx.z;
    const float *base = in + (size_t)n * Hin * Win * in_ld + c;
    f32x4 L[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int y = min(max(i - 1 + a, 0), Hin - 1);
#pragma unroll
        for (int b = 0; b < 3; ++b) L[a][b] = *reinterpret_cast<const f32x4 *>(base + ((size_t)y * Win + min(max(j - 1 + b, 0), Win - 1)) * in_ld);
    }
    f32x4 r[2][3];                                     // the two output rows, still at the three low-resolution columns
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        r[0][b] = (i > 0) ? 0.25f * L[0][b] + 0.75f * L[1][b] : L[1][b];            // src = i - 0.25: (1-l) L[i-1] + l L[i], l = 0.75
        r[1][b] = 0.75f * L[1][b] + 0.25f * L[2][b];                                // src = i + 0.25 (L[2] is L[1] again on the last row)
    }
    const int Wout = 2 * Win;
    float *o = out + ((size_t)n * 2 * Hin + 2 * i) * Wout * out_ld + (size_t)(2 * j) * out_ld + c;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const f32x4 left = (j > 0) ? 0.25f * r[a][0] + 0.75f * r[a][1] : r[a][1];
        const f32x4 right = 0.75f * r[a][1] + 0.25f * r[a][2];
        *reinterpret_cast<f32x4 *>(o + (size_t)a * Wout * out_ld) = left;
        *reinterpret_cast<f32x4 *>(o + (size_t)a * Wout * out_ld + out_ld) = right;
    }
}

__global__ __launch_bounds__(256) void resize_nchw_kernel(const float *__restrict__ in, float *__restrict__ out, int NC, int Hin,
                                                          int Win, int Hout, int Wout, int mode, int align) {
    const long long total = (long long)NC * Hout * Wout;
    const float sy = arseg_resize_scale(Hin, Hout, align), sx = arseg_resize_scale(Win, Wout, align);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % Wout), oy = (int)((idx / Wout) % Hout);
        const long long pc = idx / ((long long)Wout * Hout);
        const float *base = in + (size_t)pc * Hin * Win;
        float v;
        if (mode == ARSEG_NEAREST) {
            const int iy = min((int)floorf((float)oy * ((float)Hin / (float)Hout)), Hin - 1);
            const int ix = min((int)floorf((float)ox * ((float)Win / (float)Wout)), Win - 1);
            v = base[(size_t)iy * Win + ix];
        } else {
            int y0, y1, x0, x1; float ly, lx;
            arseg_src_index(sy, oy, align, Hin, y0, y1, ly);
            arseg_src_index(sx, ox, align, Win, x0, x1, lx);
            ly = fminf(fmaxf(ly, 0.f), 1.f); lx = fminf(fmaxf(lx, 0.f), 1.f);
            v = (1.f - ly) * ((1.f - lx) * base[(size_t)y0 * Win + x0] + lx * base[(size_t)y0 * Win + x1]) +
                ly * ((1.f - lx) * base[(size_t)y1 * Win + x0] + lx * base[(size_t)y1 * Win + x1]);
        }
        out[idx] = v;
    }
}

// Bilinear NCHW resize, 4 consecutive x outputs per thread (one 16-byte store; the row taps and the y blend are shared): the
// x8 / x16 upsamples of BiSeNetOutput (model/bisenet.py:215-216) write 0.4 GB of logits per 11-frame batch and are store bound.
__global__ __launch_bounds__(256) void resize_nchw_bilinear_x4_kernel(const float *__restrict__ in, float *__restrict__ out, int NC, int Hin, int Win,
                                                                      int Hout, int Wout, int align) {
    const int w4 = Wout >> 2;
    const long long total = (long long)NC * Hout * w4;
    const float sy = arseg_resize_scale(Hin, Hout, align), sx = arseg_resize_scale(Win, Wout, align);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int xq = (int)(idx % w4), oy = (int)((idx / w4) % Hout);
        const long long pc = idx / ((long long)w4 * Hout);
        const float *base = in + (size_t)pc * Hin * Win;
        int y0, y1; float ly;
        arseg_src_index(sy, oy, align, Hin, y0, y1, ly);
        ly = fminf(fmaxf(ly, 0.f), 1.f);
        const float *r0 = base + (size_t)y0 * Win, *r1 = base + (size_t)y1 * Win;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int x0, x1; float lx;
            arseg_src_index(sx, 4 * xq + e, align, Win, x0, x1, lx);
            lx = fminf(fmaxf(lx, 0.f), 1.f);
            v[e] = (1.f - ly) * ((1.f - lx) * r0[x0] + lx * r0[x1]) + ly * ((1.f - lx) * r1[x0] + lx * r1[x1]);
        }
        *reinterpret_cast<f32x4 *>(out + ((size_t)pc * Hout + oy) * Wout + 4 * xq) = v;
    }
}

// The same resize for an exact x8 / x16 horizontal factor with align_corners=False (the BiSeNetOutput upsamples): the S outputs
// x in [S*j + S/2, S*j + 3S/2) share their two source columns (j, j+1; clamped at the borders, where the run is half as long), so a thread
// takes one run of one output row: 4 loads and S/4 16-byte stores instead of 16 loads and the index arithmetic per store.  Each value is
// the expression of the kernel above on the same taps and the same lx (computed per element the same way): identical results.
template <int S>
__global__ __launch_bounds__(256) void resize_nchw_bilinear_runs_kernel(const float *__restrict__ in, float *__restrict__ out, int NC, int Hin,
                                                                        int Win, int Hout) {
    const int Wout = S * Win, nrun = Win + 1;
    const long long total = (long long)NC * Hout * nrun;
    const float sy = arseg_resize_scale(Hin, Hout, false), sx = arseg_resize_scale(Win, Wout, false);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int jr = (int)(idx % nrun) - 1, oy = (int)((idx / nrun) % Hout);
        const long long pc = idx / ((long long)nrun * Hout);
        const float *base = in + (size_t)pc * Hin * Win;
        int y0, y1; float ly;
        arseg_src_index(sy, oy, false, Hin, y0, y1, ly);
        ly = fminf(fmaxf(ly, 0.f), 1.f);
        const int x0 = max(jr, 0), x1 = min(jr + 1, Win - 1);
        const float *r0 = base + (size_t)y0 * Win, *r1 = base + (size_t)y1 * Win;
        const float a0 = r0[x0], a1 = r0[x1], b0 = r1[x0], b1 = r1[x1];
        const int xs = max(S * jr + S / 2, 0), xe = min(S * jr + 3 * S / 2, Wout);
        float *o = out + ((size_t)pc * Hout + oy) * Wout;
        for (int x = xs; x < xe; x += 4) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float src = fmaxf(sx * ((float)(x + e) + 0.5f) - 0.5f, 0.0f);
                float lx = src - (float)x0;
                lx = fminf(fmaxf(lx, 0.f), 1.f);
                v[e] = (1.f - ly) * ((1.f - lx) * a0 + lx * a1) + ly * ((1.f - lx) * b0 + lx * b1);
            }
            *reinterpret_cast<f32x4 *>(o + x) = v;
        }
    }
}

// ------------------------------------------------------------------ pyramid priors: sum of bilinear upsamples
// out[n,y,x,c] = sum_s bilinear(align_corners=False)( t[n, off_s .. off_s + size_s^2, c] reshaped [size_s,size_s] )(y,x)
struct PriorSizes { int n; int size[4]; int off[4]; int rows; };

// One thread per (image row, channel quad) walks x: the taps of a level change only every W / size pixels, so they are reloaded when the
// source column advances (~32 loads per 64-pixel row instead of 16 per pixel: the per-pixel form was bound by the load-instruction rate).
// The blend expression per pixel is unchanged.
__global__ __launch_bounds__(256) void psp_prior_sum_kernel(const float *__restrict__ t, float *__restrict__ out, int N, int H, int W,
                                                            int C, PriorSizes ps) {
    const int c4n = C >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)N * H * c4n) return;
    const int c = (int)(idx % c4n) * 4, y = (int)((idx / c4n) % H), n = (int)(idx / ((long long)c4n * H));
    f32x4 a[4], b[4], cc[4], d[4];
    float ly[4], sxs[4];
    int y0[4], y1[4], cx0[4];
    const float *base[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int sz = s < ps.n ? ps.size[s] : 1;
        base[s] = t + ((size_t)n * ps.rows + (s < ps.n ? ps.off[s] : 0)) * C + c;
        arseg_src_index(arseg_resize_scale(sz, H, false), y, false, sz, y0[s], y1[s], ly[s]);
        ly[s] = fminf(fmaxf(ly[s], 0.f), 1.f);
        sxs[s] = arseg_resize_scale(sz, W, false);
        cx0[s] = -1;
        a[s] = b[s] = cc[s] = d[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float *o = out + ((size_t)(n * H + y) * W) * C + c;
    for (int x = 0; x < W; ++x) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s >= ps.n) continue;
            const int sz = ps.size[s];
            int x0, x1; float lx;
            arseg_src_index(sxs[s], x, false, sz, x0, x1, lx);
            lx = fminf(fmaxf(lx, 0.f), 1.f);
            if (x0 != cx0[s]) {                      // (uniform across the wave: lanes differ in the channel only)
                cx0[s] = x0;
                a[s] = *reinterpret_cast<const f32x4 *>(base[s] + (size_t)(y0[s] * sz + x0) * C);
                b[s] = *reinterpret_cast<const f32x4 *>(base[s] + (size_t)(y0[s] * sz + x1) * C);
                cc[s] = *reinterpret_cast<const f32x4 *>(base[s] + (size_t)(y1[s] * sz + x0) * C);
                d[s] = *reinterpret_cast<const f32x4 *>(base[s] + (size_t)(y1[s] * sz + x1) * C);
            }
            acc += (1.f - ly[s]) * ((1.f - lx) * a[s] + lx * b[s]) + ly[s] * ((1.f - lx) * cc[s] + lx * d[s]);
        }
        *reinterpret_cast<f32x4 *>(o + (size_t)x * C) = acc;
    }
}

// per-pixel form (one thread per output vector): used when there are too few (row, channel quad) pairs to fill the chip (single images)
__global__ __launch_bounds__(256) void psp_prior_sum_px_kernel(const float *__restrict__ t, float *__restrict__ out, int N, int H, int W,
                                                               int C, PriorSizes ps) {
    const int c4n = C >> 2;
    const long long total = (long long)N * H * W * c4n;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) * 4;
        const long long pix = idx / c4n;
        const int x = (int)(pix % W), y = (int)((pix / W) % H), n = (int)(pix / ((long long)W * H));
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < ps.n; ++s) {
            const int sz = ps.size[s];
            const float *base = t + ((size_t)n * ps.rows + ps.off[s]) * C + c;
            int y0, y1, x0, x1; float ly, lx;
            arseg_src_index(arseg_resize_scale(sz, H, false), y, false, sz, y0, y1, ly);
            arseg_src_index(arseg_resize_scale(sz, W, false), x, false, sz, x0, x1, lx);
            ly = fminf(fmaxf(ly, 0.f), 1.f); lx = fminf(fmaxf(lx, 0.f), 1.f);
            const f32x4 a = *reinterpret_cast<const f32x4 *>(base + (size_t)(y0 * sz + x0) * C);
            const f32x4 b = *reinterpret_cast<const f32x4 *>(base + (size_t)(y0 * sz + x1) * C);
            const f32x4 cc = *reinterpret_cast<const f32x4 *>(base + (size_t)(y1 * sz + x0) * C);
            const f32x4 d = *reinterpret_cast<const f32x4 *>(base + (size_t)(y1 * sz + x1) * C);
            acc += (1.f - ly) * ((1.f - lx) * a + lx * b) + ly * ((1.f - lx) * cc + lx * d);
        }
        *reinterpret_cast<f32x4 *>(out + pix * C + c) = acc;
    }
}

// ------------------------------------------------------------------ ARM / FFM channel scaling
__global__ __launch_bounds__(256) void scale_add_kernel(const float *__restrict__ x, const float *__restrict__ scale,
                                                        const float *__restrict__ add_full, const float *__restrict__ add_vec,
                                                        float *__restrict__ out, int N, int HW, int C) {
    const int c4n = C >> 2;
    const long long total = (long long)N * HW * c4n;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) * 4;
        const long long pix = idx / c4n;
        const int n = (int)(pix / HW);
        f32x4 v = *reinterpret_cast<const f32x4 *>(x + pix * C + c) * *reinterpret_cast<const f32x4 *>(scale + (size_t)n * C + c);
        if (add_full) v += *reinterpret_cast<const f32x4 *>(add_full + pix * C + c);
        if (add_vec) v += *reinterpret_cast<const f32x4 *>(add_vec + (size_t)n * C + c);
        *reinterpret_cast<f32x4 *>(out + pix * C + c) = v;
    }
}

// ------------------------------------------------------------------ 1x1 classifier head (NHWC in, NCHW out)
template <int NC>
__global__ __launch_bounds__(256) void head_kernel(const float *__restrict__ p, int p_ld, const float *__restrict__ wf,
                                                   const float *__restrict__ bf, float *__restrict__ logits, int N, int HW, int C,
                                                   int n_cls, int log_softmax) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];   // [n_cls][C]
    for (int i = threadIdx.x; i < n_cls * C; i += blockDim.x) wsm[i] = wf[i];
    __syncthreads();
    const long long total = (long long)N * HW;
    for (long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (long long)gridDim.x * blockDim.x) {
        float acc[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) acc[k] = (k < n_cls) ? bf[k] : 0.f;
        const float *pp = p + pix * p_ld;
        for (int c = 0; c < C; c += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(pp + c);
#pragma unroll
            for (int k = 0; k < NC; ++k)
                if (k < n_cls) {
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(wsm + k * C + c);
                    acc[k] += v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3];
                }
        }
        if (log_softmax) {
            float m = -INFINITY;
#pragma unroll
            for (int k = 0; k < NC; ++k) if (k < n_cls) m = fmaxf(m, acc[k]);
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < NC; ++k) if (k < n_cls) s += expf(acc[k] - m);
            const float lse = m + logf(s);
#pragma unroll
            for (int k = 0; k < NC; ++k) acc[k] -= lse;
        }
        const int n = (int)(pix / HW);
        const long long hw = pix - (long long)n * HW;
#pragma unroll
        for (int k = 0; k < NC; ++k)
            if (k < n_cls) logits[((size_t)n * n_cls + k) * HW + hw] = acc[k];
    }
}

// The same head on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation): D[class][pixel] = W . P^T per
// 32-pixel tile of a wave.  The pixel tile is staged through LDS with coalesced 16-byte loads (the one-thread-per-pixel kernel above walks
// 256-byte rows with a lane stride of 256 bytes and re-reads the weights from LDS per pixel: 108 us for the 512x1024x64 keyframe feature
// against ~35 us of memory time).  K order: lane half lh covers channels [lh*C/2, (lh+1)*C/2) so a lane's operands are contiguous (one
// ds_read_b128 per 4 MFMAs and operand).  A lane owns one pixel and 16 of the 32 class rows; LogSoftmax needs one cross-half exchange.
template <int NPC>      // 16-byte pieces of a 32-pixel tile per lane: C / 8 (8 at C = 64, 16 at C = 128)
__global__ __launch_bounds__(256, NPC == 8 ? 3 : 2) void head_mfma_kernel(const float *__restrict__ p, int p_ld, const float *__restrict__ wf,
                                                        const float *__restrict__ bf, float *__restrict__ logits, int N, int HW, int C,
                                                        int n_cls, int log_softmax) {
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    const int PS = C + 4;                                   // row stride (floats): 16-byte aligned, rows 4 slots apart (conflict-free b128 reads)
    float *Wl = hsm;                                        // [32][PS] (rows >= n_cls zero)
    float *Bl = Wl + 32 * PS;                               // [32] bias
    float *Pl = Bl + 32 + (threadIdx.x >> 6) * 32 * PS;     // this wave's pixel tile [32][PS]
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5, c4n = C >> 2;
    for (int i = tid; i < 32 * c4n; i += 256) {
        const int r = i / c4n, c = (i - r * c4n) * 4;
        *reinterpret_cast<f32x4 *>(Wl + r * PS + c) = r < n_cls ? *reinterpret_cast<const f32x4 *>(wf + (size_t)r * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (tid < 32) Bl[tid] = tid < n_cls ? bf[tid] : 0.f;
    __syncthreads();
    const long long total = (long long)N * HW, ntile = (total + 31) / 32;
    const int npc = (32 * c4n + 63) / 64;                   // 16-byte pieces of a tile per lane (8 at C = 64)
    // The next tile of the wave is requested before the current one is multiplied (registers, npc <= 16 pieces per lane) and written to LDS
    // behind it: a wave used to load, wait, multiply and store in turn (62 us for the 512 x 1024 keyframe, 2.6 TB/s).
    f32x4 nx[NPC];
    auto fetch = [&](long long tile) {
        const long long px0 = tile * 32;
#pragma unroll
        for (int j = 0; j < NPC; ++j) {
            const int i = lane + 64 * j, r = i / c4n, c = (i - r * c4n) * 4;
            nx[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (j < npc && i < 32 * c4n && px0 + r < total) nx[j] = *reinterpret_cast<const f32x4 *>(p + (size_t)(px0 + r) * p_ld + c);
        }
    };
    const long long tstep = (long long)gridDim.x * 4;
    long long tile = (long long)blockIdx.x * 4 + (tid >> 6);
    if (tile < ntile) fetch(tile);
    for (; tile < ntile; tile += tstep) {
        const long long px0 = tile * 32;
#pragma unroll
        for (int j = 0; j < NPC; ++j) {                     // stage the tile: consecutive lanes, consecutive 16-byte pieces
            const int i = lane + 64 * j, r = i / c4n, c = (i - r * c4n) * 4;
            if (j < npc && i < 32 * c4n) *reinterpret_cast<f32x4 *>(Pl + r * PS + c) = nx[j];
        }
        if (tile + tstep < ntile) fetch(tile + tstep);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const float *wa = Wl + li * PS + lh * (C >> 1), *pb = Pl + li * PS + lh * (C >> 1);
        for (int k = 0; k < (C >> 1); k += 4) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(wa + k), b = *reinterpret_cast<const f32x4 *>(pb + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
        }
        // lane = pixel li; acc[r] = class (r&3) + 8*(r>>2) + 4*lh
        float m = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cls = (r & 3) + 8 * (r >> 2) + 4 * lh;
            acc[r] += Bl[cls];
            m = fmaxf(m, cls < n_cls ? acc[r] : -INFINITY);
        }
        if (log_softmax) {
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
            m = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            float z = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) z += (r & 3) + 8 * (r >> 2) + 4 * lh < n_cls ? expf(acc[r] - m) : 0.f;
            auto sz = __builtin_amdgcn_permlane32_swap(__float_as_uint(z), __float_as_uint(z), false, false);
            const float lse = m + logf(__uint_as_float(sz[0]) + __uint_as_float(sz[1]));
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] -= lse;
        }
        const long long pix = px0 + li;
        if (pix < total) {
            const int n = (int)(pix / HW);
            const long long hw = pix - (long long)n * HW;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cls = (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (cls < n_cls) logits[((size_t)n * n_cls + cls) * HW + hw] = acc[r];
            }
        }
    }
}

// ------------------------------------------------------------------ frame ingest: NCHW RGB -> NHWC4 (+ downscale)
__global__ __launch_bounds__(256) void frame_to_nhwc4_kernel(const float *__restrict__ img, float *__restrict__ out, int N, int H,
                                                             int W, int h, int w) {
    const long long total = (long long)N * h * w;
    const float sy = arseg_resize_scale(H, h, true), sx = arseg_resize_scale(W, w, true);
    const bool same = (h == H && w == W);
    for (long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(pix % w), oy = (int)((pix / w) % h), n = (int)(pix / ((long long)w * h));
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const float *base = img + (size_t)n * 3 * H * W;
        if (same) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = base[(size_t)c * H * W + (size_t)oy * W + ox];
        } else {
            int y0, y1, x0, x1; float ly, lx;
            arseg_src_index(sy, oy, true, H, y0, y1, ly);
            arseg_src_index(sx, ox, true, W, x0, x1, lx);
            ly = fminf(fmaxf(ly, 0.f), 1.f); lx = fminf(fmaxf(lx, 0.f), 1.f);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float *b = base + (size_t)c * H * W;
                v[c] = (1.f - ly) * ((1.f - lx) * b[(size_t)y0 * W + x0] + lx * b[(size_t)y0 * W + x1]) +
                       ly * ((1.f - lx) * b[(size_t)y1 * W + x0] + lx * b[(size_t)y1 * W + x1]);
            }
        }
        *reinterpret_cast<f32x4 *>(out + pix * 4) = v;
    }
}

// The same for the downscaling case with coalesced reads: one workgroup = 256 consecutive output pixels of one output row.  The two source
// rows under it (x span of the 256 pixels, three planes) are staged in LDS with 16-byte loads, the four taps of a pixel come from there
// (the kernel above reads 12 strided scalars per pixel: 0.4 TB/s).  Same blend, same operation order: bit-identical results.
constexpr int F4_SPAN = 1056;                    // staged floats per row and plane: 255 * sx + 4 (+3 alignment) <= F4_SPAN  <=>  sx <= 4.1
__global__ __launch_bounds__(256) void frame_to_nhwc4_rows_kernel(const float *__restrict__ img, float *__restrict__ out, int N, int H, int W, int h, int w,
                                                                  int segs) {
    __shared__ __attribute__((aligned(16))) float st[6][F4_SPAN];
    const float sy = arseg_resize_scale(H, h, true), sx = arseg_resize_scale(W, w, true);
    const int seg = blockIdx.x % segs, oy = (blockIdx.x / segs) % h, n = blockIdx.x / (segs * h);
    const int ox0 = seg * 256, ox1 = min(ox0 + 255, w - 1);
    int y0, y1, xa, xb, xe0, xe1; float ly, lt;
    arseg_src_index(sy, oy, true, H, y0, y1, ly);
    arseg_src_index(sx, ox0, true, W, xa, xb, lt);
    arseg_src_index(sx, ox1, true, W, xe0, xe1, lt);
    ly = fminf(fmaxf(ly, 0.f), 1.f);
    const int xs4 = xa & ~3, nch = (xe1 - xs4) / 4 + 1;        // 16-byte chunks per row and plane (W % 4 == 0: the last one stays inside the row)
    const float *base = img + (size_t)n * 3 * H * W;
    for (int i = threadIdx.x; i < 6 * nch; i += 256) {
        const int r = i / nch, ch = i - r * nch, c = r >> 1, y = (r & 1) ? y1 : y0;
        *reinterpret_cast<f32x4 *>(&st[r][4 * ch]) = *reinterpret_cast<const f32x4 *>(base + ((size_t)c * H + y) * W + xs4 + 4 * ch);
    }
    __syncthreads();
    const int ox = ox0 + threadIdx.x;
    if (ox < w) {
        int x0, x1; float lx;
        arseg_src_index(sx, ox, true, W, x0, x1, lx);
        lx = fminf(fmaxf(lx, 0.f), 1.f);
        x0 -= xs4; x1 -= xs4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 3; ++c)
            v[c] = (1.f - ly) * ((1.f - lx) * st[2 * c][x0] + lx * st[2 * c][x1]) + ly * ((1.f - lx) * st[2 * c + 1][x0] + lx * st[2 * c + 1][x1]);
        *reinterpret_cast<f32x4 *>(out + (((size_t)n * h + oy) * w + ox) * 4) = v;
    }
}

// decoded uint8 HWC frame -> normalised NHWC4 (+ downscale): ToTensor (x/255), Normalize ((x-mean)/std) -- dataset/camvid.py:
// 503-506, dataset/cityscapes.py:208-214 -- and the evaluator's bilinear(align_corners=True) downscale (evaluation.py:186-188)
// in one pass, each tap normalised with the reference's fp32 operation order before it is blended.
__global__ __launch_bounds__(256) void frame_u8_to_nhwc4_kernel(const uint8_t *__restrict__ img, float *__restrict__ out, int N, int H,
                                                                int W, int h, int w, float m0, float m1, float m2, float s0, float s1, float s2) {
    const long long total = (long long)N * h * w;
    const float sy = arseg_resize_scale(H, h, true), sx = arseg_resize_scale(W, w, true);
    const bool same = (h == H && w == W);
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
    for (long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(pix % w), oy = (int)((pix / w) % h), n = (int)(pix / ((long long)w * h));
        const uint8_t *base = img + (size_t)n * H * W * 3;
        auto px = [&](int y, int x, int c) { return ((float)base[((size_t)y * W + x) * 3 + c] / 255.0f - mean[c]) / sd[c]; };
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (same) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = px(oy, ox, c);
        } else {
            int y0, y1, x0, x1; float ly, lx;
            arseg_src_index(sy, oy, true, H, y0, y1, ly);
            arseg_src_index(sx, ox, true, W, x0, x1, lx);
            ly = fminf(fmaxf(ly, 0.f), 1.f); lx = fminf(fmaxf(lx, 0.f), 1.f);
#pragma unroll
            for (int c = 0; c < 3; ++c)
                v[c] = (1.f - ly) * ((1.f - lx) * px(y0, x0, c) + lx * px(y0, x1, c)) + ly * ((1.f - lx) * px(y1, x0, c) + lx * px(y1, x1, c));
        }
        *reinterpret_cast<f32x4 *>(out + pix * 4) = v;
    }
}

// ------------------------------------------------------------------ mergeMotion: chain per-frame codec MVs to the keyframe
// (pre-process/generate_compressed_dataset_camvid.py:6-56).  Frames are sequential (a pixel links to the parent of its
// target in an earlier frame), pixels independent: one launch per frame, an integer gather from the earlier frames' links.
// dp[f][y][x] = int4 (x, y, frame, -) of the linked position, frame == -1: no link yet (the keyframe).
__device__ __forceinline__ int round_half_even_div4(int v) {        // np.round(v / 4) for integer v
    const int b = v >> 2, r = v & 3;                                  // v = 4b + r, floor division
    return r < 2 ? b : (r > 2 ? b + 1 : b + (b & 1));                 // .5 -> the even neighbour
}
__global__ __launch_bounds__(256) void merge_motion_step_kernel(const int16_t *__restrict__ flow, int4 *__restrict__ dp, int f1, int H, int W) {
    const int hw = H * W;
    for (int pix = blockIdx.x * blockDim.x + threadIdx.x; pix < hw; pix += gridDim.x * blockDim.x) {
        const int y = pix / W, x = pix - y * W;
        int mx = flow[(size_t)pix * 3], my = flow[(size_t)pix * 3 + 1], ref = flow[(size_t)pix * 3 + 2];
        if (ref < 0 || ref >= 3) { mx = 0; my = 0; ref = 0; }           // intra block: zero motion, previous frame (:20-22)
        const int j2 = min(max(y + round_half_even_div4(my), 0), H - 1), k2 = min(max(x + round_half_even_div4(mx), 0), W - 1);
        const int f2 = max(0, f1 - ref - 1);
        const int4 parent = dp[(size_t)f2 * hw + (size_t)j2 * W + k2];
        dp[(size_t)f1 * hw + pix] = parent.z != -1 ? parent : make_int4(k2, j2, f2, 0);
    }
}
__global__ __launch_bounds__(256) void merge_motion_out_kernel(const int4 *__restrict__ dp, int16_t *__restrict__ out, int F1, int H, int W) {
    const long long total = (long long)F1 * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int pix = (int)(i % ((long long)H * W)), f = (int)(i / ((long long)H * W)), y = pix / W, x = pix - y * W;
        const int4 d = dp[i];
        // frame 0 keeps the initial -1 (the reference converts frames 1.. only, :53-54); astype(np.short) wraps like the cast
        out[i * 2] = (int16_t)(f == 0 ? d.x : (d.x - x) * 4);
        out[i * 2 + 1] = (int16_t)(f == 0 ? d.y : (d.y - y) * 4);
    }
}
__global__ __launch_bounds__(256) void fill_int4_kernel(int4 *__restrict__ p, long long n, int v) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = make_int4(v, v, v, v);
}

// ------------------------------------------------------------------ layout changes (LDS-tiled 32x32 transposes)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float *__restrict__ in, float *__restrict__ out, int C, int HW,
                                                           int out_ld) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, hw0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, hw = hw0 + tx;
        tile[j][tx] = (c < C && hw < HW) ? in[((size_t)n * C + c) * HW + hw] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int hw = hw0 + j, c = c0 + tx;
        if (c < C && hw < HW) out[((size_t)n * HW + hw) * out_ld + c] = tile[tx][j];
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float *__restrict__ in, int in_ld, float *__restrict__ out, int C,
                                                           int HW) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, hw0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int hw = hw0 + j, c = c0 + tx;
        tile[j][tx] = (c < C && hw < HW) ? in[((size_t)n * HW + hw) * in_ld + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, hw = hw0 + tx;
        if (c < C && hw < HW) out[((size_t)n * C + c) * HW + hw] = tile[tx][j];
    }
}

// ------------------------------------------------------------------ evaluator tail
__global__ __launch_bounds__(256) void argmax_confusion_kernel(const float *__restrict__ logits, const int64_t *__restrict__ label,
                                                               int32_t *__restrict__ pred, unsigned long long *__restrict__ hist,
                                                               int N, int n_cls, int h, int w, int H, int W, int ignore_label, int align) {
    __shared__ unsigned int lh[1024];
    for (int i = threadIdx.x; i < n_cls * n_cls; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    const long long total = (long long)N * H * W;
    const float sy = arseg_resize_scale(h, H, align != 0), sx = arseg_resize_scale(w, W, align != 0);
    const bool same = (h == H && w == W);
    for (long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(pix % W), oy = (int)((pix / W) % H), n = (int)(pix / ((long long)W * H));
        int y0 = oy, y1 = oy, x0 = ox, x1 = ox; float ly = 0.f, lx = 0.f;
        if (!same) {
            arseg_src_index(sy, oy, align != 0, h, y0, y1, ly);
            arseg_src_index(sx, ox, align != 0, w, x0, x1, lx);
            ly = fminf(fmaxf(ly, 0.f), 1.f); lx = fminf(fmaxf(lx, 0.f), 1.f);
        }
        // torch.argmax semantics: the first maximum wins, a NaN counts as the maximum (the first NaN wins)
        float best = -INFINITY; int bi = 0; bool best_nan = false;
        for (int k = 0; k < n_cls; ++k) {
            const float *b = logits + ((size_t)n * n_cls + k) * h * w;
            float v;
            if (same) v = b[(size_t)oy * w + ox];
            else v = (1.f - ly) * ((1.f - lx) * b[(size_t)y0 * w + x0] + lx * b[(size_t)y0 * w + x1]) +
                     ly * ((1.f - lx) * b[(size_t)y1 * w + x0] + lx * b[(size_t)y1 * w + x1]);
            const bool isn = v != v, take = !best_nan & ((v > best) | isn);          // branch free
            best = take ? v : best; bi = take ? k : bi; best_nan = best_nan | (take & isn);
        }
        if (pred) pred[pix] = bi;
        if (hist && label) {
            const long long lab = label[pix];
            if (lab != ignore_label && lab >= 0 && lab < n_cls) atomicAdd(&lh[(int)lab * n_cls + bi], 1u);
        }
    }
    __syncthreads();
    if (hist)
        for (int i = threadIdx.x; i < n_cls * n_cls; i += blockDim.x)
            if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
}

// The same tail for an exact x S bilinear upsample with align_corners=False, S a power of two (BiSeNetOutput's nn.Upsample(x8),
// model/bisenet.py:215-216): the S output pixels x = S*j + S/2 .. S*j + 3S/2 - 1 of a row all interpolate between the low-resolution
// columns j and j+1 (src = j + (r + 0.5) / S), so one thread takes such a run, loads its 4 taps once per class and evaluates the S
// pixels from registers -- 4 loads per class and run instead of 4 S; the per-pixel kernel above issues 76 scattered loads per output
// pixel at 19 classes and is bound by the texture path's instruction rate (77 us per 1024x2048 frame; this one: memory-side trivial).
// Same taps and weights (arseg_src_index) and the same argmax semantics; the blend is regrouped (see below).
template <int S>
__global__ __launch_bounds__(256) void argmax_confusion_up_kernel(const float *__restrict__ logits, const int64_t *__restrict__ label,
                                                                  int32_t *__restrict__ pred, unsigned long long *__restrict__ hist,
                                                                  int N, int n_cls, int h, int w, int ignore_label) {
    __shared__ unsigned int lh[1024];
    for (int i = threadIdx.x; i < n_cls * n_cls; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    const int H = S * h, W = S * w, runs = w + 1;
    const long long total = (long long)N * H * runs;
    const bool pred_vec = (reinterpret_cast<uintptr_t>(pred) & 15u) == 0;      // 16 / 8-byte label stores need an aligned base (rows are multiples of S)
    const float sc = arseg_resize_scale(h, H, false);          // = 1 / S exactly
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(idx % runs) - 1, oy = (int)((idx / runs) % H), n = (int)(idx / ((long long)runs * H));
        int y0, y1; float ly;
        arseg_src_index(sc, oy, false, h, y0, y1, ly);
        ly = fminf(fmaxf(ly, 0.f), 1.f);
        const int x0 = max(j, 0), x1 = min(x0 + 1, w - 1), xs = S * j + S / 2;       // first output column of the run (may be negative for j = -1)
        float lx[S];
#pragma unroll
        for (int r = 0; r < S; ++r) {
            int a, b;
            arseg_src_index(sc, min(max(xs + r, 0), W - 1), false, w, a, b, lx[r]);
            lx[r] = fminf(fmaxf(lx[r], 0.f), 1.f);
        }
        float best[S]; int bi[S]; bool bn[S];
#pragma unroll
        for (int r = 0; r < S; ++r) { best[r] = -INFINITY; bi[r] = 0; bn[r] = false; }
        const float *b = logits + (size_t)n * n_cls * h * w;
        const size_t o00 = (size_t)y0 * w + x0, o01 = (size_t)y0 * w + x1, o10 = (size_t)y1 * w + x0, o11 = (size_t)y1 * w + x1, cs = (size_t)h * w;
        for (int k0 = 0; k0 < n_cls; k0 += 4) {          // four classes' taps in flight (a class at a time is bound by the load latency)
            float t[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float *bk = b + (size_t)min(k0 + u, n_cls - 1) * cs;
                if (x1 > x0) {          // the two taps of a row are neighbours: one 8-byte load (the kernel is bound by the number of load instructions)
                    // (a 4-byte aligned pair type: the address is odd in floats for every other run -- gfx950 global loads take any dword
                    // address, and the reduced alignment makes that a defined access instead of a misaligned float2)
                    typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
                    const f32x2_a4 a01 = *reinterpret_cast<const f32x2_a4 *>(bk + o00), a11 = *reinterpret_cast<const f32x2_a4 *>(bk + o10);
                    t[u][0] = a01.x; t[u][1] = a01.y; t[u][2] = a11.x; t[u][3] = a11.y;
                } else {
                    t[u][0] = bk[o00]; t[u][1] = bk[o01]; t[u][2] = bk[o10]; t[u][3] = bk[o11];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (k0 + u >= n_cls) break;
                // the bilinear blend is linear in lx along the run: v(r) = a + lx[r] * b -- one FMA per pixel and class (the expanded form,
                // 6 operations, made this kernel VALU bound); same value up to fp32 rounding of the regrouped sum
                const float a = (1.f - ly) * t[u][0] + ly * t[u][2];
                const float b = (1.f - ly) * (t[u][1] - t[u][0]) + ly * (t[u][3] - t[u][2]);
#pragma unroll
                for (int r = 0; r < S; ++r) {          // branch free (the short-circuit form compiles to a divergent branch per pixel and class)
                    const float v = fmaf(lx[r], b, a);
                    const bool isn = v != v, take = !bn[r] & ((v > best[r]) | isn);
                    best[r] = take ? v : best[r];
                    bi[r] = take ? k0 + u : bi[r];
                    bn[r] = bn[r] | (take & isn);
                }
            }
        }
        const long long row = ((long long)n * H + oy) * W;
        const bool whole = xs >= 0 && xs + S <= W;              // interior run: S consecutive labels, S/2 * 4 bytes aligned (W = S w, xs = S j + S/2)
        if (pred && whole && S >= 4 && pred_vec) {               // vector stores (scalar ones: 4 bytes per lane at a 4 S byte stride)
            if constexpr (S == 8) {
                *reinterpret_cast<int4 *>(pred + row + xs) = int4{bi[0], bi[1], bi[2], bi[3]};
                *reinterpret_cast<int4 *>(pred + row + xs + 4) = int4{bi[S > 4 ? 4 : 0], bi[S > 5 ? 5 : 0], bi[S > 6 ? 6 : 0], bi[S > 7 ? 7 : 0]};
            } else {
                *reinterpret_cast<int2 *>(pred + row + xs) = int2{bi[0], bi[1]};
                *reinterpret_cast<int2 *>(pred + row + xs + 2) = int2{bi[S > 2 ? 2 : 0], bi[S > 3 ? 3 : 0]};
            }
        }
#pragma unroll
        for (int r = 0; r < S; ++r) {
            const int ox = xs + r;
            if (ox < 0 || ox >= W) continue;
            if (pred && !(whole && S >= 4 && pred_vec)) pred[row + ox] = bi[r];
            if (hist && label) {
                const long long lab = label[row + ox];
                if (lab != ignore_label && lab >= 0 && lab < n_cls) atomicAdd(&lh[(int)lab * n_cls + bi[r]], 1u);
            }
        }
    }
    __syncthreads();
    if (hist)
        for (int i = threadIdx.x; i < n_cls * n_cls; i += blockDim.x)
            if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
}

}  // namespace

extern "C" int arseg_maxpool3x3s2_fwd(const float *in, float *out, int N, int H, int W, int C, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(in); ARSEG_CHECK_PTR(out); ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(H); ARSEG_CHECK_POS(W); ARSEG_CHECK_POS(C);
    if (C & 3) return ARSEG_EINVAL;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for((long long)N * Ho * Wo * (C >> 2))), dim3(256), 0, arseg_stream(stream),
                       in, out, N, H, W, C, Ho, Wo);
    return arseg_launch_status();
}

extern "C" int arseg_adaptive_avgpool_fwd(const float *in, int in_ld, float *out, int out_ld, long long out_n_stride, int N, int H,
                                          int W, int C, int oh, int ow, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(in); ARSEG_CHECK_PTR(out); ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(H); ARSEG_CHECK_POS(W); ARSEG_CHECK_POS(C);
    ARSEG_CHECK_POS(oh); ARSEG_CHECK_POS(ow);
    if ((C & 3) || (in_ld & 3) || in_ld < C) return ARSEG_EINVAL;
    if (out_ld == 0) out_ld = C;
    if (out_n_stride == 0) out_n_stride = (long long)oh * ow * out_ld;
    if ((out_ld & 3) || out_ld < C || (out_n_stride & 3) || !ARSEG_ALIGNED16(out)) return ARSEG_EINVAL;
    const dim3 grid(oh * ow, arseg_cdiv(C, 16), N);
    if ((long long)grid.x * grid.y * grid.z < 256)          // few, large windows (1x1 / 2x2 pyramid levels): 4x the pixel lanes per block
        hipLaunchKernelGGL((window_reduce_kernel<false, 256>), grid, dim3(1024), 0, arseg_stream(stream), in, in_ld, out, out_ld, out_n_stride, H, W, C, oh, ow);
    else
        hipLaunchKernelGGL((window_reduce_kernel<false, 64>), grid, dim3(256), 0, arseg_stream(stream), in, in_ld, out, out_ld, out_n_stride, H, W, C, oh, ow);
    return arseg_launch_status();
}

// adaptive average pooling into one column block of a block-structured matrix [N][rows][n_blocks * C] (the folded PSP pyramid,
// model/pspnet.py:14-31): level `block` writes its pooled map into columns [block*C, (block+1)*C) of its rows and zeros into the same rows of
// the other blocks.  `out` = first row of the level in image 0 (column 0 of the matrix); out_n_stride = elements between images.
extern "C" int arseg_adaptive_avgpool_blockrow_fwd(const float *in, int in_ld, float *out, long long out_n_stride, int N, int H, int W, int C,
                                                   int oh, int ow, int n_blocks, int block, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(in); ARSEG_CHECK_PTR(out); ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(H); ARSEG_CHECK_POS(W); ARSEG_CHECK_POS(C);
    ARSEG_CHECK_POS(oh); ARSEG_CHECK_POS(ow); ARSEG_CHECK_POS(n_blocks);
    if (block < 0 || block >= n_blocks) return ARSEG_EINVAL;
    const int out_ld = n_blocks * C;
    if ((C & 3) || (in_ld & 3) || in_ld < C || (out_n_stride & 3) || out_n_stride < (long long)oh * ow * out_ld || !ARSEG_ALIGNED16(out)) return ARSEG_EINVAL;
    float *o = out + (size_t)block * C;
    if (!(C & 63) && (long long)oh * ow * (C / 64) * N >= 64) {
        // 64 channels per workgroup: a pixel's share is 256 contiguous bytes (with 16 channels every pixel row of the map is fetched in 64-byte
        // pieces by 32 different workgroups: 16-20 us per pyramid level of the 11-frame LR batch against 10-11.5).  Launches that would
        // have fewer than 64 such workgroups (the 1x1 / 2x2 levels of a single keyframe) keep the narrow form: they need the parallelism more
        const dim3 grid(oh * ow, C / 64, N);
        if ((long long)grid.x * grid.y * grid.z < 512)
            hipLaunchKernelGGL((window_reduce_kernel<false, 64, 16>), grid, dim3(1024), 0, arseg_stream(stream), in, in_ld, o, out_ld, out_n_stride, H, W, C, oh, ow, n_blocks, block);
        else
            hipLaunchKernelGGL((window_reduce_kernel<false, 16, 16>), grid, dim3(256), 0, arseg_stream(stream), in, in_ld, o, out_ld, out_n_stride, H, W, C, oh, ow, n_blocks, block);
        return arseg_launch_status();
    }
    const dim3 grid(oh * ow, arseg_cdiv(C, 16), N);
    if ((long long)grid.x * grid.y * grid.z < 256)
        hipLaunchKernelGGL((window_reduce_kernel<false, 256>), grid, dim3(1024), 0, arseg_stream(stream), in, in_ld, o, out_ld, out_n_stride, H, W, C, oh, ow, n_blocks, block);
    else
        hipLaunchKernelGGL((window_reduce_kernel<false, 64>), grid, dim3(256), 0, arseg_stream(stream), in, in_ld, o, out_ld, out_n_stride, H, W, C, oh, ow, n_blocks, block);
    return arseg_launch_status();
}

static int pool_grid(int H, int W, int n_sizes, const int *sizes, PoolGrid *g) {
    if (n_sizes < 1 || n_sizes > 4) return ARSEG_EINVAL;
    g->nlev = n_sizes; g->rows = 0;
    for (int i = 0; i < 4; ++i) { g->size[i] = 1; g->off[i] = 0; }
    int ey[64], ex[64], ny = 0, nx = 0;
    for (int i = 0; i < n_sizes; ++i) {
        const int s = sizes[i];
        if (s <= 0 || s > 6) return ARSEG_EUNSUPPORTED;
        g->size[i] = s; g->off[i] = g->rows; g->rows += s * s;
        for (int b = 0; b < s; ++b) {
            ey[ny++] = (b * H) / s; ey[ny++] = ((b + 1) * H + s - 1) / s;
            ex[nx++] = (b * W) / s; ex[nx++] = ((b + 1) * W + s - 1) / s;
        }
    }
    auto uniq = [](int *a, int n) {
        for (int i = 1; i < n; ++i) { const int v = a[i]; int j = i - 1; while (j >= 0 && a[j] > v) { a[j + 1] = a[j]; --j; } a[j + 1] = v; }
        int m = 0;
        for (int i = 0; i < n; ++i) if (m == 0 || a[m - 1] != a[i]) a[m++] = a[i];
        return m;
    };
    ny = uniq(ey, ny); nx = uniq(ex, nx);
    if (ny > 26 || nx > 26) return ARSEG_EUNSUPPORTED;
    for (int i = 0; i < ny; ++i) g->ey[i] = ey[i];
    for (int i = 0; i < nx; ++i) g->ex[i] = ex[i];
    g->ny = ny - 1; g->nx = nx - 1;
    return ARSEG_OK;
}

// The folded pyramid's pooled matrix [N][rows][n_sizes * C] (rows = sum s^2: level i's adaptive average pool in columns [i*C, (i+1)*C) of its
// rows, zeros elsewhere) in one pass over the map; workspace = the cell sums.
extern "C" size_t arseg_psp_pool_matrix_workspace_bytes(int N, int H, int W, int C, int n_sizes, const int *sizes) {
    PoolGrid g;
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || !sizes || pool_grid(H, W, n_sizes, sizes, &g) != ARSEG_OK) return 0;
    return (size_t)N * g.ny * g.nx * C * sizeof(float);
}

extern "C" int arseg_psp_pool_matrix_fwd(const float *in, int in_ld, float *out, void *workspace, size_t workspace_bytes, int N, int H, int W, int C,
                                         int n_sizes, const int *sizes, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(in); ARSEG_CHECK_PTR(out); ARSEG_CHECK_PTR(workspace); ARSEG_CHECK_PTR(sizes);
    ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(H); ARSEG_CHECK_POS(W); ARSEG_CHECK_POS(C);
    if ((C & 3) || (in_ld & 3) || in_ld < C || !ARSEG_ALIGNED16(in) || !ARSEG_ALIGNED16(out) || !ARSEG_ALIGNED16(workspace) || N > 65535) return ARSEG_EINVAL;
    PoolGrid g;
    if (int e = pool_grid(H, W, n_sizes, sizes, &g)) return e;
    if (workspace_bytes < (size_t)N * g.ny * g.nx * C * sizeof(float)) return ARSEG_EWORKSPACE;
    hipStream_t hs = arseg_stream(stream);
    hipLaunchKernelGGL(psp_cells_kernel, dim3(g.ny * g.nx, arseg_cdiv(C, 64), N), dim3(256), 0, hs, in, in_ld, reinterpret_cast<float *>(workspace), H, W, C, g);
    hipLaunchKernelGGL(psp_bins_kernel, dim3(g.rows, N), dim3(256), 0, hs, reinterpret_cast<const float *>(workspace), out, H, W, C, g);
    return arseg_launch_status();
}

extern "C" int arseg_psp_prior_sum_fwd(const float *t, float *out, int N, int H, int W, int C, int n_sizes, const int *sizes,
                                       arseg_stream_t stream) {
    ARSEG_CHECK_PTR(t); ARSEG_CHECK_PTR(out); ARSEG_CHECK_PTR(sizes);
    ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(H); ARSEG_CHECK_POS(W); ARSEG_CHECK_POS(C);
    if (n_sizes < 1 || n_sizes > 4 || (C & 3) || !ARSEG_ALIGNED16(t) || !ARSEG_ALIGNED16(out)) return ARSEG_EINVAL;
    PriorSizes ps;
    ps.n = n_sizes; ps.rows = 0;
    for (int i = 0; i < 4; ++i) { ps.size[i] = 1; ps.off[i] = 0; }
    for (int i = 0; i < n_sizes; ++i) {
        if (sizes[i] <= 0) return ARSEG_EINVAL;
        ps.size[i] = sizes[i]; ps.off[i] = ps.rows; ps.rows += sizes[i] * sizes[i];
    }
    if ((long long)N * H * (C >> 2) >= 49152)
        hipLaunchKernelGGL(psp_prior_sum_kernel, dim3(arseg_cdiv((long long)N * H * (C >> 2), 256)), dim3(256), 0, arseg_stream(stream), t, out, N,
                           H, W, C, ps);
    else
        hipLaunchKernelGGL(psp_prior_sum_px_kernel, dim3(grid_for((long long)N * H * W * (C >> 2))), dim3(256), 0, arseg_stream(stream), t, out, N,
                           H, W, C, ps);
    return arseg_launch_status();
}

extern "C" int arseg_global_reduce_fwd(const float *in, int in_ld, float *out, int N, int H, int W, int C, int op,
                                       arseg_stream_t stream) {
    ARSEG_CHECK_PTR(in); ARSEG_CHECK_PTR(out); ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(H); ARSEG_CHECK_POS(W); ARSEG_CHECK_POS(C);
    if ((C & 3) || (in_ld & 3) || in_ld < C) return ARSEG_EINVAL;
    dim3 grid(1, arseg_cdiv(C, 16), N);
    const bool big = (long long)grid.y * grid.z < 256;
    if (op == ARSEG_REDUCE_MEAN) {
        if (big) hipLaunchKernelGGL((window_reduce_kernel<false, 256>), grid, dim3(1024), 0, arseg_stream(stream), in, in_ld, out, C, (long long)C, H, W, C, 1, 1);
        else hipLaunchKernelGGL((window_reduce_kernel<false, 64>), grid, dim3(256), 0, arseg_stream(stream), in, in_ld, out, C, (long long)C, H, W, C, 1, 1);
    } else if (op == ARSEG_REDUCE_MAX) {
        if (big) hipLaunchKernelGGL((window_reduce_kernel<true, 256>), grid, dim3(1024), 0, arseg_stream(stream), in, in_ld, out, C, (long long)C, H, W, C, 1, 1);
        else hipLaunchKernelGGL((window_reduce_kernel<true, 64>), grid, dim3(256), 0, arseg_stream(stream), in, in_ld, out, C, (long long)C, H, W, C, 1, 1);
    } else return ARSEG_EINVAL;
    return arseg_launch_status();
}

static int global_parts(int N, int H, int W, int C) {          // row bands per image of the two-stage reduce (0: the one-launch form is fine)
    const long long HW = (long long)H * W, blocks = (long long)N * arseg_cdiv(C, 64);
    if ((long long)N * arseg_cdiv(C, 16) >= 64 || HW < 2048 || HW >= (1ll << 31)) return 0;      // (the one-launch form already has enough workgroups)
    long long parts = 768 / blocks;
    if (parts > HW / 256) parts = HW / 256;
    return parts >= 2 ? (int)parts : 0;
}
extern "C" size_t arseg_global_reduce_workspace_bytes(int N, int H, int W, int C) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    return (size_t)N * global_parts(N, H, W, C) * C * sizeof(float);
}
// arseg_global_reduce_fwd with a workspace (arseg_global_reduce_workspace_bytes; 0 bytes = not needed): large maps of few images are reduced in
// two deterministic stages so that the whole chip reads them.
extern "C" int arseg_global_reduce_ws_fwd(const float *in, int in_ld, float *out, void *workspace, size_t workspace_bytes, int N, int H, int W, int C,
                                          int op, arseg_stream_t stream) {
    const int parts = (N > 0 && H > 0 && W > 0 && C > 0) ? global_parts(N, H, W, C) : 0;
    if (!parts || !workspace || N > 65535) return arseg_global_reduce_fwd(in, in_ld, out, N, H, W, C, op, stream);
    ARSEG_CHECK_PTR(in); ARSEG_CHECK_PTR(out);
    if ((C & 3) || (in_ld & 3) || in_ld < C || !ARSEG_ALIGNED16(in) || !ARSEG_ALIGNED16(out) || !ARSEG_ALIGNED16(workspace)) return ARSEG_EINVAL;
    if (workspace_bytes < (size_t)N * parts * C * sizeof(float)) return ARSEG_EWORKSPACE;
    if (op != ARSEG_REDUCE_MEAN && op != ARSEG_REDUCE_MAX) return ARSEG_EINVAL;
    hipStream_t hs = arseg_stream(stream);
    float *ws = reinterpret_cast<float *>(workspace);
    const dim3 g1(parts, arseg_cdiv(C, 64), N), g2(arseg_cdiv(C, 1024), N);
    if (op == ARSEG_REDUCE_MAX) {
        hipLaunchKernelGGL(global_parts_kernel<true>, g1, dim3(256), 0, hs, in, in_ld, ws, H * W, C, parts);
        hipLaunchKernelGGL(global_combine_kernel<true>, g2, dim3(256), 0, hs, ws, out, H * W, C, parts);
    } else {
        hipLaunchKernelGGL(global_parts_kernel<false>, g1, dim3(256), 0, hs, in, in_ld, ws, H * W, C, parts);
        hipLaunchKernelGGL(global_combine_kernel<false>, g2, dim3(256), 0, hs, ws, out, H * W, C, parts);
    }
    return arseg_launch_status();
}


extern "C" int arseg_resize_fwd(const float *in, float *out, int N, int C, int Hin, int Win, int Hout, int Wout, int mode,
                                int align_corners, int layout, int in_ld, int out_ld, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(in); ARSEG_CHECK_PTR(out); ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(C); ARSEG_CHECK_POS(Hin); ARSEG_CHECK_POS(Win);
    ARSEG_CHECK_POS(Hout); ARSEG_CHECK_POS(Wout);
    if (mode != ARSEG_NEAREST && mode != ARSEG_BILINEAR) return ARSEG_EINVAL;
    if (layout == ARSEG_NHWC) {
        if ((C & 3) || (in_ld & 3) || (out_ld & 3) || in_ld < C || out_ld < C) return ARSEG_EINVAL;
        if (mode == ARSEG_BILINEAR && !align_corners && Hout == 2 * Hin && Wout == 2 * Win && Hin <= 65535 && N <= 65535) {
            hipLaunchKernelGGL(upsample2x_nhwc_kernel, dim3(arseg_cdiv((long long)Win * (C >> 2), 256), Hin, N), dim3(256), 0, arseg_stream(stream),
                               in, out, C, Hin, Win, in_ld, out_ld);
            return arseg_launch_status();
        }
        hipLaunchKernelGGL(resize_nhwc_kernel, dim3(grid_for((long long)N * Hout * Wout * (C >> 2))), dim3(256), 0,
                           arseg_stream(stream), in, out, N, C, Hin, Win, Hout, Wout, mode, align_corners ? 1 : 0, in_ld, out_ld);
    } else if (layout == ARSEG_NCHW) {
        if (mode == ARSEG_BILINEAR && !align_corners && (Wout == 8 * Win || Wout == 16 * Win) && ARSEG_ALIGNED16(out)) {
            const int g = grid_for((long long)N * C * Hout * (Win + 1), 65536);
            if (Wout == 8 * Win)
                hipLaunchKernelGGL(resize_nchw_bilinear_runs_kernel<8>, dim3(g), dim3(256), 0, arseg_stream(stream), in, out, N * C, Hin, Win, Hout);
            else
                hipLaunchKernelGGL(resize_nchw_bilinear_runs_kernel<16>, dim3(g), dim3(256), 0, arseg_stream(stream), in, out, N * C, Hin, Win, Hout);
        } else if (mode == ARSEG_BILINEAR && (Wout & 3) == 0 && ARSEG_ALIGNED16(out))
            hipLaunchKernelGGL(resize_nchw_bilinear_x4_kernel, dim3(grid_for((long long)N * C * Hout * (Wout >> 2), 16384)), dim3(256), 0,
                               arseg_stream(stream), in, out, N * C, Hin, Win, Hout, Wout, align_corners ? 1 : 0);
        else
            hipLaunchKernelGGL(resize_nchw_kernel, dim3(grid_for((long long)N * C * Hout * Wout, 16384)), dim3(256), 0,
                               arseg_stream(stream), in, out, N * C, Hin, Win, Hout, Wout, mode, align_corners ? 1 : 0);
    } else return ARSEG_EINVAL;
    return arseg_launch_status();
}

extern "C" int arseg_scale_add_fwd(const float *x, const float *scale, const float *add_full, const float *add_vec, float *out,
                                   int N, int HW, int C, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(x); ARSEG_CHECK_PTR(scale); ARSEG_CHECK_PTR(out); ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(HW); ARSEG_CHECK_POS(C);
    if (C & 3) return ARSEG_EINVAL;
    hipLaunchKernelGGL(scale_add_kernel, dim3(grid_for((long long)N * HW * (C >> 2))), dim3(256), 0, arseg_stream(stream), x, scale,
                       add_full, add_vec, out, N, HW, C);
    return arseg_launch_status();
}

extern "C" int arseg_head_fwd(const float *p, int p_ld, const float *wf, const float *bf, float *logits, int N, int HW, int C,
                              int n_cls, int log_softmax, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(p); ARSEG_CHECK_PTR(wf); ARSEG_CHECK_PTR(bf); ARSEG_CHECK_PTR(logits);
    ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(HW); ARSEG_CHECK_POS(C); ARSEG_CHECK_POS(n_cls);
    if ((C & 3) || (p_ld & 3) || p_ld < C) return ARSEG_EINVAL;
    if (n_cls > 32 || (size_t)n_cls * C * sizeof(float) > 60 * 1024) return ARSEG_EUNSUPPORTED;
    hipStream_t st = arseg_stream(stream);
    if (!(C & 7) && C <= 128 && ARSEG_ALIGNED16(p) && ARSEG_ALIGNED16(wf)) {           // fp32 matrix-core kernel: 4 waves x 32-pixel tiles per workgroup
        const size_t sm = (size_t)(5 * 32 * (C + 4) + 32) * sizeof(float);
        static ArsegSmemAttr attr;
        const long long ntile = ((long long)N * HW + 31) / 32;
        long long gb = (ntile + 3) / 4;
        const dim3 grid((unsigned)(gb > 2048 ? 2048 : gb));
        if (C <= 64) {
            if (int e = arseg_allow_smem(attr, reinterpret_cast<const void *>(head_mfma_kernel<8>), sm)) return e;
            hipLaunchKernelGGL(head_mfma_kernel<8>, grid, dim3(256), sm, st, p, p_ld, wf, bf, logits, N, HW, C, n_cls, log_softmax);
        } else {
            static ArsegSmemAttr attr16;
            if (int e = arseg_allow_smem(attr16, reinterpret_cast<const void *>(head_mfma_kernel<16>), sm)) return e;
            hipLaunchKernelGGL(head_mfma_kernel<16>, grid, dim3(256), sm, st, p, p_ld, wf, bf, logits, N, HW, C, n_cls, log_softmax);
        }
        return arseg_launch_status();
    }
    const size_t smem = (size_t)n_cls * C * sizeof(float);
    const int g = grid_for((long long)N * HW, 2048);
    if (n_cls <= 12) hipLaunchKernelGGL(head_kernel<12>, dim3(g), dim3(256), smem, st, p, p_ld, wf, bf, logits, N, HW, C, n_cls, log_softmax);
    else if (n_cls <= 19) hipLaunchKernelGGL(head_kernel<19>, dim3(g), dim3(256), smem, st, p, p_ld, wf, bf, logits, N, HW, C, n_cls, log_softmax);
    else hipLaunchKernelGGL(head_kernel<32>, dim3(g), dim3(256), smem, st, p, p_ld, wf, bf, logits, N, HW, C, n_cls, log_softmax);
    return arseg_launch_status();
}

extern "C" int arseg_frame_to_nhwc4_fwd(const float *img, float *out, int N, int H, int W, int h, int w, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(img); ARSEG_CHECK_PTR(out); ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(H); ARSEG_CHECK_POS(W); ARSEG_CHECK_POS(h); ARSEG_CHECK_POS(w);
    const float sx = arseg_resize_scale(W, w, true);
    const int segs = arseg_cdiv(w, 256);
    if (!(h == H && w == W) && W % 4 == 0 && ARSEG_ALIGNED16(img) && 255.f * sx + 8.f <= (float)F4_SPAN && (long long)N * h * segs < (1ll << 31)) {
        hipLaunchKernelGGL(frame_to_nhwc4_rows_kernel, dim3((unsigned)(N * h * segs)), dim3(256), 0, arseg_stream(stream), img, out, N, H, W, h, w, segs);
        return arseg_launch_status();
    }
    hipLaunchKernelGGL(frame_to_nhwc4_kernel, dim3(grid_for((long long)N * h * w)), dim3(256), 0, arseg_stream(stream), img, out, N, H, W, h, w);
    return arseg_launch_status();
}

extern "C" int arseg_frame_u8_to_nhwc4_fwd(const uint8_t *img_hwc, float *out, int N, int H, int W, int h, int w, const float *mean3,
                                           const float *std3, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(img_hwc); ARSEG_CHECK_PTR(out); ARSEG_CHECK_PTR(mean3); ARSEG_CHECK_PTR(std3);
    ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(H); ARSEG_CHECK_POS(W); ARSEG_CHECK_POS(h); ARSEG_CHECK_POS(w);
    if (std3[0] == 0.f || std3[1] == 0.f || std3[2] == 0.f) return ARSEG_EINVAL;
    hipLaunchKernelGGL(frame_u8_to_nhwc4_kernel, dim3(grid_for((long long)N * h * w)), dim3(256), 0, arseg_stream(stream), img_hwc, out, N, H,
                       W, h, w, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    return arseg_launch_status();
}

extern "C" size_t arseg_merge_motion_workspace_bytes(int n_frames, int H, int W) {
    return n_frames < 0 || H <= 0 || W <= 0 ? 0 : (size_t)(n_frames + 1) * H * W * sizeof(int4);
}

extern "C" int arseg_merge_motion_fwd(const int16_t *flows, int16_t *out, void *workspace, size_t workspace_bytes, int n_frames, int frame_start,
                                      int H, int W, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(flows); ARSEG_CHECK_PTR(out); ARSEG_CHECK_PTR(workspace); ARSEG_CHECK_POS(n_frames); ARSEG_CHECK_POS(H); ARSEG_CHECK_POS(W);
    if (frame_start < 0 || frame_start >= n_frames) return ARSEG_EINVAL;
    if (workspace_bytes < arseg_merge_motion_workspace_bytes(n_frames, H, W)) return ARSEG_EWORKSPACE;
    if (!ARSEG_ALIGNED16(workspace) || (long long)H * W > (1ll << 30)) return ARSEG_EINVAL;
    hipStream_t st = arseg_stream(stream);
    int4 *dp = reinterpret_cast<int4 *>(workspace);
    const long long n = (long long)(n_frames + 1) * H * W;
    hipLaunchKernelGGL(fill_int4_kernel, dim3(grid_for(n)), dim3(256), 0, st, dp, n, -1);
    for (int f1 = frame_start + 1; f1 <= n_frames; ++f1)
        hipLaunchKernelGGL(merge_motion_step_kernel, dim3(grid_for((long long)H * W)), dim3(256), 0, st, flows + (size_t)f1 * H * W * 3, dp, f1, H, W);
    hipLaunchKernelGGL(merge_motion_out_kernel, dim3(grid_for(n)), dim3(256), 0, st, dp, out, n_frames + 1, H, W);
    return arseg_launch_status();
}

extern "C" int arseg_nchw_to_nhwc_fwd(const float *in, float *out, int N, int C, int HW, int out_ld, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(in); ARSEG_CHECK_PTR(out); ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(C); ARSEG_CHECK_POS(HW);
    if (out_ld < C) return ARSEG_EINVAL;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(arseg_cdiv(HW, 32), arseg_cdiv(C, 32), N), dim3(256), 0, arseg_stream(stream), in, out, C, HW, out_ld);
    return arseg_launch_status();
}

extern "C" int arseg_nhwc_to_nchw_fwd(const float *in, int in_ld, float *out, int N, int C, int HW, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(in); ARSEG_CHECK_PTR(out); ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(C); ARSEG_CHECK_POS(HW);
    if (in_ld < C) return ARSEG_EINVAL;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(arseg_cdiv(HW, 32), arseg_cdiv(C, 32), N), dim3(256), 0, arseg_stream(stream), in, in_ld, out, C, HW);
    return arseg_launch_status();
}

extern "C" int arseg_argmax_confusion_fwd(const float *logits, const int64_t *label, int32_t *pred, int64_t *hist, int N, int n_cls,
                                          int h, int w, int H, int W, int ignore_label, int align_corners, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(logits); ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(n_cls); ARSEG_CHECK_POS(h); ARSEG_CHECK_POS(w); ARSEG_CHECK_POS(H); ARSEG_CHECK_POS(W);
    if (n_cls > 32) return ARSEG_EUNSUPPORTED;
    if (!pred && !(hist && label)) return ARSEG_EINVAL;
    unsigned long long *hh = reinterpret_cast<unsigned long long *>(hist);
    const int S = H / h;
    if (!align_corners && S * h == H && S * w == W && (S == 2 || S == 4 || S == 8)) {          // exact power-of-two upsample: one thread per run of S pixels
        const int g = grid_for((long long)N * H * (w + 1), 4096);
        hipStream_t st = arseg_stream(stream);
        if (S == 8) hipLaunchKernelGGL(argmax_confusion_up_kernel<8>, dim3(g), dim3(256), 0, st, logits, label, pred, hh, N, n_cls, h, w, ignore_label);
        else if (S == 4) hipLaunchKernelGGL(argmax_confusion_up_kernel<4>, dim3(g), dim3(256), 0, st, logits, label, pred, hh, N, n_cls, h, w, ignore_label);
        else hipLaunchKernelGGL(argmax_confusion_up_kernel<2>, dim3(g), dim3(256), 0, st, logits, label, pred, hh, N, n_cls, h, w, ignore_label);
        return arseg_launch_status();
    }
    hipLaunchKernelGGL(argmax_confusion_kernel, dim3(grid_for((long long)N * H * W, 1024)), dim3(256), 0, arseg_stream(stream), logits,
                       label, pred, hh, N, n_cls, h, w, H, W, ignore_label, align_corners);
    return arseg_launch_status();
}
