// Small NHWC layers of the 16-bit storage path (BASELINE configs[2] / configs[4]: BiSeNet-18 with bf16 / fp16 tensors; the layers of
// model/bisenet.py:77,215,252-258,284-298,390-398 and the frame ingest of evaluation.py:186-188).  Same arithmetic as layers.hip with
// 16-bit loads / stores: every kernel reads 8 channels (16 bytes) per thread, computes in fp32 and rounds once (to nearest even) at the
// store.  dtype = ARSEG_DT_F16 | ARSEG_DT_BF16.
#include "arseg_common.h"
#include "warp_math.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool BF>
__device__ __forceinline__ void unpack8(const u32x4 v, float (&f)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[2 * e] = arseg_h2f<BF>((uint16_t)(v[e] & 0xffffu)); f[2 * e + 1] = arseg_h2f<BF>((uint16_t)(v[e] >> 16)); }
}
template <bool BF>
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
    u32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (unsigned)arseg_f2h<BF>(f[2 * e]) | ((unsigned)arseg_f2h<BF>(f[2 * e + 1]) << 16);
    return v;
}
__device__ __forceinline__ u32x4 ld8(const uint16_t *p) { return *reinterpret_cast<const u32x4 *>(p); }
__device__ __forceinline__ void st8(uint16_t *p, const u32x4 v) { *reinterpret_cast<u32x4 *>(p) = v; }

inline int grid_for(long long total, int cap = 8192) {
    long long b = (total + 255) / 256;
    return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

// ------------------------------------------------------------------ frame ingest: NCHW fp32 RGB -> NHWC8 16-bit (+ bilinear align_corners=True downscale)
template <bool BF>
__global__ __launch_bounds__(256) void frame_to_nhwc8_kernel(const float *__restrict__ img, uint16_t *__restrict__ out, int N, int H, int W, int h, int w) {
    const long long total = (long long)N * h * w;
    const float sy = arseg_resize_scale(H, h, true), sx = arseg_resize_scale(W, w, true);
    const bool same = (h == H && w == W);
    for (long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(pix % w), oy = (int)((pix / w) % h), n = (int)(pix / ((long long)w * h));
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
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
        st8(out + pix * 8, pack8<BF>(v));
    }
}

// The same for the downscaling case with coalesced reads (the 16-bit twin of frame_to_nhwc4_rows_kernel, csrc/layers.hip): one workgroup =
// 256 consecutive output pixels of one output row; the two source rows under it (x span of the 256 pixels, three planes) are staged in LDS
// with 16-byte loads and the four taps of a pixel come from there -- the kernel above reads 12 strided scalars per pixel.  Same blend,
// same operation order: bit-identical results.
constexpr int F8_SPAN = 1056;                    // staged floats per row and plane: 255 * sx + 4 (+3 alignment) <= F8_SPAN  <=>  sx <= 4.1
template <bool BF>
__global__ __launch_bounds__(256) void frame_to_nhwc8_rows_kernel(const float *__restrict__ img, uint16_t *__restrict__ out, int N, int H, int W, int h, int w,
                                                                  int segs) {
    __shared__ __attribute__((aligned(16))) float st[6][F8_SPAN];
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
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 3; ++c)
            v[c] = (1.f - ly) * ((1.f - lx) * st[2 * c][x0] + lx * st[2 * c][x1]) + ly * ((1.f - lx) * st[2 * c + 1][x0] + lx * st[2 * c + 1][x1]);
        st8(out + (((size_t)n * h + oy) * w + ox) * 8, pack8<BF>(v));
    }
}

// ------------------------------------------------------------------ nn.MaxPool2d(3, stride 2, padding 1)
template <bool BF>
__global__ __launch_bounds__(256) void maxpool16_kernel(const uint16_t *__restrict__ in, uint16_t *__restrict__ out, int N, int H, int W, int C, int Ho, int Wo) {
    const int c8n = C >> 3;
    const long long total = (long long)N * Ho * Wo * c8n;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c8n) * 8;
        const long long pix = idx / c8n;
        const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), n = (int)(pix / ((long long)Wo * Ho));
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = 2 * oy - 1 + dy;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = 2 * ox - 1 + dx;
                if ((unsigned)ix >= (unsigned)W) continue;
                float f[8];
                unpack8<BF>(ld8(in + (((size_t)n * H + iy) * W + ix) * C + c), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], f[e]);
            }
        }
        st8(out + pix * C + c, pack8<BF>(m));
    }
}

// ------------------------------------------------------------------ torch.mean(x, (2,3)) in two steps: fp32 partial sums of pixel slices, then the mean
// step 1: grid (C/256 blocks of 32 channel vectors, N, S slices); block = 32 channel vectors x 8 pixel lanes; part[n][s][c] fp32
template <bool BF>
__global__ __launch_bounds__(256) void global_sum16_kernel(const uint16_t *__restrict__ in, int in_ld, float *__restrict__ part, int HW, int C, int S) {
    __shared__ float red[8][32][8];
    const int n = blockIdx.y, sl = blockIdx.z, cv = blockIdx.x * 32 + (threadIdx.x & 31), pl = threadIdx.x >> 5;
    const int per = (HW + S - 1) / S, p0 = sl * per, p1 = min(p0 + per, HW);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (cv * 8 < C) {
        const uint16_t *base = in + (size_t)n * HW * in_ld + cv * 8;
        for (int px = p0 + pl; px < p1; px += 8) {
            float f[8];
            unpack8<BF>(ld8(base + (size_t)px * in_ld), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += f[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[pl][threadIdx.x & 31][e] = acc[e];
    __syncthreads();
    if (pl == 0 && cv * 8 < C) {
        float *o = part + ((size_t)n * S + sl) * C + cv * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x & 31][e];
            o[e] = t;
        }
    }
}
// step 2: out[n][c] = sum_s part[n][s][c] / HW.  8 lanes per (n, c) take every 8th slice, their sums are combined in lane order
// (a fixed order: deterministic); one thread per (n, c) walking all S slices was a chain of S dependent-latency loads (60 us at S = 1024).
template <bool BF>
__global__ __launch_bounds__(256) void global_mean_fin16_kernel(const float *__restrict__ part, uint16_t *__restrict__ out, int N, int C, int S, float inv) {
    __shared__ float red[8][32];
    const int i = blockIdx.x * 32 + (threadIdx.x & 31), k = threadIdx.x >> 5;
    float t = 0.f;
    if (i < N * C) {
        const int n = i / C, c = i - n * C;
        for (int s = k; s < S; s += 8) t += part[((size_t)n * S + s) * C + c];
    }
    red[k][threadIdx.x & 31] = t;
    __syncthreads();
    if (k == 0 && i < N * C) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) a += red[j][threadIdx.x];
        out[i] = arseg_f2h<BF>(a * inv);
    }
}

// ------------------------------------------------------------------ resize (nearest / bilinear, align_corners on / off)
template <bool BF>
__global__ __launch_bounds__(256) void resize16_kernel(const uint16_t *__restrict__ in, uint16_t *__restrict__ out, int N, int C, int Hin, int Win,
                                                       int Hout, int Wout, int mode, int align, int in_ld, int out_ld) {
    const int c8n = C >> 3;
    const long long total = (long long)N * Hout * Wout * c8n;
    const float sy = arseg_resize_scale(Hin, Hout, align), sx = arseg_resize_scale(Win, Wout, align);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c8n) * 8;
        const long long pix = idx / c8n;
        const int ox = (int)(pix % Wout), oy = (int)((pix / Wout) % Hout), n = (int)(pix / ((long long)Wout * Hout));
        const uint16_t *base = in + (size_t)n * Hin * Win * in_ld + c;
        if (mode == ARSEG_NEAREST) {
            const int iy = min((int)floorf((float)oy * ((float)Hin / (float)Hout)), Hin - 1);
            const int ix = min((int)floorf((float)ox * ((float)Win / (float)Wout)), Win - 1);
            st8(out + pix * out_ld + c, ld8(base + ((size_t)iy * Win + ix) * in_ld));
        } else {
            int y0, y1, x0, x1; float ly, lx;
            arseg_src_index(sy, oy, align, Hin, y0, y1, ly);
            arseg_src_index(sx, ox, align, Win, x0, x1, lx);
            ly = fminf(fmaxf(ly, 0.f), 1.f); lx = fminf(fmaxf(lx, 0.f), 1.f);
            float a[8], b[8], cc[8], d[8], v[8];
            unpack8<BF>(ld8(base + ((size_t)y0 * Win + x0) * in_ld), a);
            unpack8<BF>(ld8(base + ((size_t)y0 * Win + x1) * in_ld), b);
            unpack8<BF>(ld8(base + ((size_t)y1 * Win + x0) * in_ld), cc);
            unpack8<BF>(ld8(base + ((size_t)y1 * Win + x1) * in_ld), d);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (1.f - ly) * ((1.f - lx) * a[e] + lx * b[e]) + ly * ((1.f - lx) * cc[e] + lx * d[e]);
            st8(out + pix * out_ld + c, pack8<BF>(v));
        }
    }
}

// ------------------------------------------------------------------ ARM / FFM channel scaling: x * scale[n,c] (+ add_full) (+ add_vec[n,c])
template <bool BF>
__global__ __launch_bounds__(256) void scale_add16_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ scale, const uint16_t *__restrict__ add_full,
                                                          const uint16_t *__restrict__ add_vec, uint16_t *__restrict__ out, int N, int HW, int C) {
    const int c8n = C >> 3;
    const long long total = (long long)N * HW * c8n;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c8n) * 8;
        const long long pix = idx / c8n;
        const int n = (int)(pix / HW);
        float v[8], s[8], t[8];
        unpack8<BF>(ld8(x + pix * C + c), v);
        unpack8<BF>(ld8(scale + (size_t)n * C + c), s);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= s[e];
        if (add_full) {
            unpack8<BF>(ld8(add_full + pix * C + c), t);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += t[e];
        }
        if (add_vec) {
            unpack8<BF>(ld8(add_vec + (size_t)n * C + c), t);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += t[e];
        }
        st8(out + pix * C + c, pack8<BF>(v));
    }
}

// ------------------------------------------------------------------ 1x1 classifier head: NHWC 16-bit feature, fp32 weights, NCHW fp32 logits
template <bool BF, int NC>
__global__ __launch_bounds__(256) void head16_kernel(const uint16_t *__restrict__ p, int p_ld, const float *__restrict__ wf, const float *__restrict__ bf,
                                                     float *__restrict__ logits, int N, int HW, int C, int n_cls, int log_softmax) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];   // [n_cls][C]
    for (int i = threadIdx.x; i < n_cls * C; i += blockDim.x) wsm[i] = wf[i];
    __syncthreads();
    const long long total = (long long)N * HW;
    for (long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (long long)gridDim.x * blockDim.x) {
        float acc[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) acc[k] = (k < n_cls) ? bf[k] : 0.f;
        const uint16_t *pp = p + pix * p_ld;
        for (int c = 0; c < C; c += 8) {
            float v[8];
            unpack8<BF>(ld8(pp + c), v);
#pragma unroll
            for (int k = 0; k < NC; ++k)
                if (k < n_cls) {
                    const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wsm + k * C + c), w1 = *reinterpret_cast<const f32x4 *>(wsm + k * C + c + 4);
                    acc[k] += v[0] * w0[0] + v[1] * w0[1] + v[2] * w0[2] + v[3] * w0[3] + v[4] * w1[0] + v[5] * w1[1] + v[6] * w1[2] + v[7] * w1[3];
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

// The same head on the fp32 matrix cores (see head_mfma_kernel in layers.hip): D[class][pixel] per 32-pixel tile of a wave, the 16-bit
// pixel tile converted to fp32 while it is staged through LDS in 64-channel chunks (C % 64 == 0), fp32 weights resident in LDS.  The
// one-thread-per-pixel kernel above re-reads every weight from LDS per pixel (1200 ds_read_b128 per pixel at C = 256, 19 classes).
template <bool BF>
__global__ __launch_bounds__(256) void head16_mfma_kernel(const uint16_t *__restrict__ p, int p_ld, const float *__restrict__ wf, const float *__restrict__ bf,
                                                          float *__restrict__ logits, int N, int HW, int C, int n_cls, int log_softmax) {
    constexpr int KC = 64, PS = KC + 4;
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    const int WS = C + 4;
    float *Wl = hsm;                                        // [32][WS] (rows >= n_cls zero)
    float *Bl = Wl + 32 * WS;                               // [32] bias
    float *Pl = Bl + 32 + (threadIdx.x >> 6) * 32 * PS;     // this wave's pixel tile chunk [32][PS]
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5, c4n = C >> 2;
    for (int i = tid; i < 32 * c4n; i += 256) {
        const int r = i / c4n, c = (i - r * c4n) * 4;
        *reinterpret_cast<f32x4 *>(Wl + r * WS + c) = r < n_cls ? *reinterpret_cast<const f32x4 *>(wf + (size_t)r * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (tid < 32) Bl[tid] = tid < n_cls ? bf[tid] : 0.f;
    __syncthreads();
    const long long total = (long long)N * HW, ntile = (total + 31) / 32;
    for (long long tile = (long long)blockIdx.x * 4 + (tid >> 6); tile < ntile; tile += (long long)gridDim.x * 4) {
        const long long px0 = tile * 32;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        for (int kc = 0; kc < C; kc += KC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {                   // 32 pixels x 8 pieces of 8 halves: consecutive lanes, consecutive 16-byte pieces
                const int i = lane + 64 * j, r = i >> 3, c = (i & 7) * 8;
                float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (px0 + r < total) unpack8<BF>(ld8(p + (size_t)(px0 + r) * p_ld + kc + c), f);
                *reinterpret_cast<f32x4 *>(Pl + r * PS + c) = f32x4{f[0], f[1], f[2], f[3]};
                *reinterpret_cast<f32x4 *>(Pl + r * PS + c + 4) = f32x4{f[4], f[5], f[6], f[7]};
            }
            const float *wa = Wl + li * WS + kc + lh * (KC / 2), *pb = Pl + li * PS + lh * (KC / 2);
#pragma unroll
            for (int k = 0; k < KC / 2; k += 4) {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(wa + k), b = *reinterpret_cast<const f32x4 *>(pb + k);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
            }
        }
        float m = -INFINITY;                                // lane = pixel li; acc[r] = class (r&3) + 8*(r>>2) + 4*lh
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

// ------------------------------------------------------------------ element type conversion (fp32 <-> 16-bit), 8 elements per thread
template <bool BF>
__global__ __launch_bounds__(256) void cast_to32_kernel(const uint16_t *__restrict__ in, float *__restrict__ out, long long n8) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        float f[8];
        unpack8<BF>(ld8(in + i * 8), f);
        *reinterpret_cast<f32x4 *>(out + i * 8) = f32x4{f[0], f[1], f[2], f[3]};
        *reinterpret_cast<f32x4 *>(out + i * 8 + 4) = f32x4{f[4], f[5], f[6], f[7]};
    }
}
template <bool BF>
__global__ __launch_bounds__(256) void cast_to16_kernel(const float *__restrict__ in, uint16_t *__restrict__ out, long long n8) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(in + i * 8), b = *reinterpret_cast<const f32x4 *>(in + i * 8 + 4);
        const float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        st8(out + i * 8, pack8<BF>(f));
    }
}

// ------------------------------------------------------------------ MV resize + warp of a 16-bit NHWC keyframe feature -> fp32 C8 (the CReFF kernels' input)
// Same structure as warp_mvq_nhwc_kernel (warp.hip): one lane per pixel does the fp64 coordinate arithmetic, then 8 lanes per pixel
// move 8-channel vectors; the blend is fp32 on the converted taps.
template <bool BF>
__global__ __launch_bounds__(256) void warp_mvq16_kernel(const uint16_t *__restrict__ feat, const int16_t *__restrict__ mv, float *__restrict__ out, int N, int C,
                                                         int Hp, int Wp, int H, int W, long long feat_n_stride) {
    __shared__ int s_off[4][64];
    __shared__ float s_w[4][64];
    const int tid = threadIdx.x, y = blockIdx.y, n = blockIdx.z, xb = blockIdx.x * 64;
    if (tid < 64) {
        const int x = min(xb + tid, Wp - 1);
        double fx, fy;
        if (Hp == H && Wp == W) {
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
    const int sub = tid & 7;
    const uint16_t *img = feat + (size_t)n * feat_n_stride;       // (stride 0: the frames of a GOP sample one keyframe feature)
    const int hw = Hp * Wp;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int pl = it * 32 + (tid >> 3);
        const int x = min(xb + pl, Wp - 1), pix = y * Wp + x;
        const int o0 = s_off[0][pl], o1 = s_off[1][pl], o2 = s_off[2][pl], o3 = s_off[3][pl];
        const float w0 = s_w[0][pl], w1 = s_w[1][pl], w2 = s_w[2][pl], w3 = s_w[3][pl];
        for (int c = sub * 8; c < C; c += 64) {
            float a[8], b[8], cc[8], d[8];
            unpack8<BF>(ld8(img + (size_t)o0 * C + c), a);
            unpack8<BF>(ld8(img + (size_t)o1 * C + c), b);
            unpack8<BF>(ld8(img + (size_t)o2 * C + c), cc);
            unpack8<BF>(ld8(img + (size_t)o3 * C + c), d);
            f32x4 r0, r1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                r0[e] = a[e] * w0 + b[e] * w1 + cc[e] * w2 + d[e] * w3;
                r1[e] = a[4 + e] * w0 + b[4 + e] * w1 + cc[4 + e] * w2 + d[4 + e] * w3;
            }
            float *o = out + (((size_t)n * (C >> 3) + (c >> 3)) * hw + pix) * 8;      // C8: [N][C/8][H][W][8]
            *reinterpret_cast<f32x4 *>(o) = r0;
            *reinterpret_cast<f32x4 *>(o + 4) = r1;
        }
    }
}

#define DISPATCH_BF(dtype, CALL_T, CALL_F) do { if ((dtype) == ARSEG_DT_BF16) { CALL_T; } else if ((dtype) == ARSEG_DT_F16) { CALL_F; } else return ARSEG_EINVAL; } while (0)

}  // namespace

extern "C" int arseg_frame_to_nhwc8_16_fwd(const float *img, void *out, int dtype, int N, int H, int W, int h, int w, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(img); ARSEG_CHECK_PTR(out); ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(H); ARSEG_CHECK_POS(W); ARSEG_CHECK_POS(h); ARSEG_CHECK_POS(w);
    if (!ARSEG_ALIGNED16(out)) return ARSEG_EINVAL;
    const int segs = arseg_cdiv(w, 256);
    const float sx = arseg_resize_scale(W, w, true);
    if (!(h == H && w == W) && W % 4 == 0 && ARSEG_ALIGNED16(img) && 255.f * sx + 8.f <= (float)F8_SPAN && (long long)N * h * segs < (1ll << 31)) {
        const unsigned gr = (unsigned)(N * h * segs);
        DISPATCH_BF(dtype, hipLaunchKernelGGL(frame_to_nhwc8_rows_kernel<true>, dim3(gr), dim3(256), 0, arseg_stream(stream), img, (uint16_t *)out, N, H, W, h, w, segs),
                    hipLaunchKernelGGL(frame_to_nhwc8_rows_kernel<false>, dim3(gr), dim3(256), 0, arseg_stream(stream), img, (uint16_t *)out, N, H, W, h, w, segs));
        return arseg_launch_status();
    }
    const int g = grid_for((long long)N * h * w);
    DISPATCH_BF(dtype, hipLaunchKernelGGL(frame_to_nhwc8_kernel<true>, dim3(g), dim3(256), 0, arseg_stream(stream), img, (uint16_t *)out, N, H, W, h, w),
                hipLaunchKernelGGL(frame_to_nhwc8_kernel<false>, dim3(g), dim3(256), 0, arseg_stream(stream), img, (uint16_t *)out, N, H, W, h, w));
    return arseg_launch_status();
}

extern "C" int arseg_maxpool3x3s2_16_fwd(const void *in, void *out, int dtype, int N, int H, int W, int C, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(in); ARSEG_CHECK_PTR(out); ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(H); ARSEG_CHECK_POS(W); ARSEG_CHECK_POS(C);
    if ((C & 7) || !ARSEG_ALIGNED16(in) || !ARSEG_ALIGNED16(out)) return ARSEG_EINVAL;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int g = grid_for((long long)N * Ho * Wo * (C >> 3));
    DISPATCH_BF(dtype, hipLaunchKernelGGL(maxpool16_kernel<true>, dim3(g), dim3(256), 0, arseg_stream(stream), (const uint16_t *)in, (uint16_t *)out, N, H, W, C, Ho, Wo),
                hipLaunchKernelGGL(maxpool16_kernel<false>, dim3(g), dim3(256), 0, arseg_stream(stream), (const uint16_t *)in, (uint16_t *)out, N, H, W, C, Ho, Wo));
    return arseg_launch_status();
}

static int mean16_slices(int N, int HW, int C) {
    long long blocks = (long long)arseg_cdiv(C, 256) * N;
    int S = (int)(512 / (blocks < 1 ? 1 : blocks));           // aim for ~512 workgroups
    S = S < 1 ? 1 : S;
    const int maxS = (HW + 63) / 64;                           // at least 64 pixels per slice
    return S > maxS ? maxS : S;
}

extern "C" size_t arseg_global_mean16_workspace_bytes(int N, int H, int W, int C) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    return (size_t)N * mean16_slices(N, H * W, C) * C * sizeof(float);
}

extern "C" int arseg_global_mean16_fwd(const void *in, int in_ld, void *out, int dtype, int N, int H, int W, int C, void *workspace,
                                       size_t workspace_bytes, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(in); ARSEG_CHECK_PTR(out); ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(H); ARSEG_CHECK_POS(W); ARSEG_CHECK_POS(C);
    if ((C & 7) || (in_ld & 7) || in_ld < C || !ARSEG_ALIGNED16(in) || N > 65535) return ARSEG_EINVAL;
    const int S = mean16_slices(N, H * W, C);
    if (!workspace || workspace_bytes < (size_t)N * S * C * sizeof(float)) return ARSEG_EWORKSPACE;
    const dim3 grid(arseg_cdiv(C, 256), N, S);
    float *part = (float *)workspace;
    const int gf = arseg_cdiv((long long)N * C, 32);
    hipStream_t st = arseg_stream(stream);
    const float inv = 1.0f / (float)(H * W);
    if (dtype == ARSEG_DT_BF16) {
        hipLaunchKernelGGL(global_sum16_kernel<true>, grid, dim3(256), 0, st, (const uint16_t *)in, in_ld, part, H * W, C, S);
        hipLaunchKernelGGL(global_mean_fin16_kernel<true>, dim3(gf), dim3(256), 0, st, part, (uint16_t *)out, N, C, S, inv);
    } else if (dtype == ARSEG_DT_F16) {
        hipLaunchKernelGGL(global_sum16_kernel<false>, grid, dim3(256), 0, st, (const uint16_t *)in, in_ld, part, H * W, C, S);
        hipLaunchKernelGGL(global_mean_fin16_kernel<false>, dim3(gf), dim3(256), 0, st, part, (uint16_t *)out, N, C, S, inv);
    } else return ARSEG_EINVAL;
    return arseg_launch_status();
}

extern "C" int arseg_resize16_fwd(const void *in, void *out, int dtype, int N, int C, int Hin, int Win, int Hout, int Wout, int mode, int align_corners,
                                  int in_ld, int out_ld, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(in); ARSEG_CHECK_PTR(out); ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(C); ARSEG_CHECK_POS(Hin); ARSEG_CHECK_POS(Win);
    ARSEG_CHECK_POS(Hout); ARSEG_CHECK_POS(Wout);
    if (mode != ARSEG_NEAREST && mode != ARSEG_BILINEAR) return ARSEG_EINVAL;
    if ((C & 7) || (in_ld & 7) || (out_ld & 7) || in_ld < C || out_ld < C || !ARSEG_ALIGNED16(in) || !ARSEG_ALIGNED16(out)) return ARSEG_EINVAL;
    const int g = grid_for((long long)N * Hout * Wout * (C >> 3));
    DISPATCH_BF(dtype, hipLaunchKernelGGL(resize16_kernel<true>, dim3(g), dim3(256), 0, arseg_stream(stream), (const uint16_t *)in, (uint16_t *)out, N, C, Hin, Win, Hout, Wout, mode, align_corners, in_ld, out_ld),
                hipLaunchKernelGGL(resize16_kernel<false>, dim3(g), dim3(256), 0, arseg_stream(stream), (const uint16_t *)in, (uint16_t *)out, N, C, Hin, Win, Hout, Wout, mode, align_corners, in_ld, out_ld));
    return arseg_launch_status();
}

extern "C" int arseg_scale_add16_fwd(const void *x, const void *scale, const void *add_full, const void *add_vec, void *out, int dtype, int N, int HW, int C,
                                     arseg_stream_t stream) {
    ARSEG_CHECK_PTR(x); ARSEG_CHECK_PTR(scale); ARSEG_CHECK_PTR(out); ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(HW); ARSEG_CHECK_POS(C);
    if ((C & 7) || !ARSEG_ALIGNED16(x) || !ARSEG_ALIGNED16(scale) || !ARSEG_ALIGNED16(out) || (add_full && !ARSEG_ALIGNED16(add_full)) || (add_vec && !ARSEG_ALIGNED16(add_vec)))
        return ARSEG_EINVAL;
    const int g = grid_for((long long)N * HW * (C >> 3));
    DISPATCH_BF(dtype, hipLaunchKernelGGL(scale_add16_kernel<true>, dim3(g), dim3(256), 0, arseg_stream(stream), (const uint16_t *)x, (const uint16_t *)scale, (const uint16_t *)add_full, (const uint16_t *)add_vec, (uint16_t *)out, N, HW, C),
                hipLaunchKernelGGL(scale_add16_kernel<false>, dim3(g), dim3(256), 0, arseg_stream(stream), (const uint16_t *)x, (const uint16_t *)scale, (const uint16_t *)add_full, (const uint16_t *)add_vec, (uint16_t *)out, N, HW, C));
    return arseg_launch_status();
}

extern "C" int arseg_head16_fwd(const void *p, int p_ld, int dtype, const float *wf, const float *bf, float *logits, int N, int HW, int C, int n_cls,
                                int log_softmax, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(p); ARSEG_CHECK_PTR(wf); ARSEG_CHECK_PTR(bf); ARSEG_CHECK_PTR(logits); ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(HW); ARSEG_CHECK_POS(C); ARSEG_CHECK_POS(n_cls);
    if ((C & 7) || (p_ld & 7) || p_ld < C || !ARSEG_ALIGNED16(p)) return ARSEG_EINVAL;
    if (n_cls > 32) return ARSEG_EUNSUPPORTED;
    hipStream_t st = arseg_stream(stream);
    const uint16_t *pp = (const uint16_t *)p;
    if (!(C & 63) && C <= 512 && ARSEG_ALIGNED16(wf)) {          // fp32 matrix-core kernel, 64-channel chunks
        const size_t sm = (size_t)(32 * (C + 4) + 32 + 4 * 32 * 68) * sizeof(float);
        static ArsegSmemAttr attr_bf, attr_h;
        const long long ntile = ((long long)N * HW + 31) / 32;
        const long long gb = (ntile + 3) / 4;
        const dim3 grid((unsigned)(gb > 2048 ? 2048 : gb));
        if (dtype == ARSEG_DT_BF16) {
            if (int e = arseg_allow_smem(attr_bf, reinterpret_cast<const void *>(head16_mfma_kernel<true>), sm)) return e;
            hipLaunchKernelGGL(head16_mfma_kernel<true>, grid, dim3(256), sm, st, pp, p_ld, wf, bf, logits, N, HW, C, n_cls, log_softmax);
        } else if (dtype == ARSEG_DT_F16) {
            if (int e = arseg_allow_smem(attr_h, reinterpret_cast<const void *>(head16_mfma_kernel<false>), sm)) return e;
            hipLaunchKernelGGL(head16_mfma_kernel<false>, grid, dim3(256), sm, st, pp, p_ld, wf, bf, logits, N, HW, C, n_cls, log_softmax);
        } else return ARSEG_EINVAL;
        return arseg_launch_status();
    }
    const size_t smem = (size_t)n_cls * C * sizeof(float);
    if (smem > 64 * 1024) return ARSEG_EUNSUPPORTED;
    const int g = grid_for((long long)N * HW, 2048);
#define HEAD(BF_) do { if (n_cls <= 12) hipLaunchKernelGGL((head16_kernel<BF_, 12>), dim3(g), dim3(256), smem, st, pp, p_ld, wf, bf, logits, N, HW, C, n_cls, log_softmax); \
                       else if (n_cls <= 19) hipLaunchKernelGGL((head16_kernel<BF_, 19>), dim3(g), dim3(256), smem, st, pp, p_ld, wf, bf, logits, N, HW, C, n_cls, log_softmax); \
                       else hipLaunchKernelGGL((head16_kernel<BF_, 32>), dim3(g), dim3(256), smem, st, pp, p_ld, wf, bf, logits, N, HW, C, n_cls, log_softmax); } while (0)
    DISPATCH_BF(dtype, HEAD(true), HEAD(false));
#undef HEAD
    return arseg_launch_status();
}

extern "C" int arseg_cast_fwd(const void *in, int in_dtype, void *out, int out_dtype, long long count, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(in); ARSEG_CHECK_PTR(out);
    if (count <= 0 || (count & 7) || !ARSEG_ALIGNED16(in) || !ARSEG_ALIGNED16(out)) return ARSEG_EINVAL;
    const int g = grid_for(count / 8);
    hipStream_t st = arseg_stream(stream);
    if (out_dtype == ARSEG_DT_F32) {
        DISPATCH_BF(in_dtype, hipLaunchKernelGGL(cast_to32_kernel<true>, dim3(g), dim3(256), 0, st, (const uint16_t *)in, (float *)out, count / 8),
                    hipLaunchKernelGGL(cast_to32_kernel<false>, dim3(g), dim3(256), 0, st, (const uint16_t *)in, (float *)out, count / 8));
    } else if (in_dtype == ARSEG_DT_F32) {
        DISPATCH_BF(out_dtype, hipLaunchKernelGGL(cast_to16_kernel<true>, dim3(g), dim3(256), 0, st, (const float *)in, (uint16_t *)out, count / 8),
                    hipLaunchKernelGGL(cast_to16_kernel<false>, dim3(g), dim3(256), 0, st, (const float *)in, (uint16_t *)out, count / 8));
    } else return ARSEG_EINVAL;
    return arseg_launch_status();
}

extern "C" int arseg_warp_mvq16_fwd(const void *feature, int dtype, const int16_t *mv_q, float *out_c8, int N, int C, int Hp, int Wp, int H, int W,
                                    arseg_stream_t stream) {
    return arseg_warp_mvq16_shared_fwd(feature, (long long)Hp * Wp * C, dtype, mv_q, out_c8, N, C, Hp, Wp, H, W, stream);
}

extern "C" int arseg_warp_mvq16_shared_fwd(const void *feature, long long feat_n_stride, int dtype, const int16_t *mv_q, float *out_c8, int N, int C,
                                           int Hp, int Wp, int H, int W, arseg_stream_t stream) {
    if (feat_n_stride < 0 || (feat_n_stride & 7)) return ARSEG_EINVAL;
    ARSEG_CHECK_PTR(feature); ARSEG_CHECK_PTR(mv_q); ARSEG_CHECK_PTR(out_c8);
    ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(C); ARSEG_CHECK_POS(H); ARSEG_CHECK_POS(W); ARSEG_CHECK_POS(Hp); ARSEG_CHECK_POS(Wp);
    if ((C & 7) || !ARSEG_ALIGNED16(feature) || !ARSEG_ALIGNED16(out_c8)) return ARSEG_EINVAL;
    if (Hp > 65535 || N > 65535) return ARSEG_EUNSUPPORTED;
    const dim3 grid(arseg_cdiv(Wp, 64), Hp, N);
    DISPATCH_BF(dtype, hipLaunchKernelGGL(warp_mvq16_kernel<true>, grid, dim3(256), 0, arseg_stream(stream), (const uint16_t *)feature, mv_q, out_c8, N, C, Hp, Wp, H, W, feat_n_stride),
                hipLaunchKernelGGL(warp_mvq16_kernel<false>, grid, dim3(256), 0, arseg_stream(stream), (const uint16_t *)feature, mv_q, out_c8, N, C, Hp, Wp, H, W, feat_n_stride));
    return arseg_launch_status();
}
