// conv3x3(pad 1) of a x2 bilinear upsample (PSPUpsample, /root/reference/model/pspnet.py:34-46) without the upsampled tensor and
// without a transform of the wide input:
//
//   conv3x3(Up(x)) = sum_t shift_t( W_t . Up(x) ) = sum_t shift_t( Up(W_t . x) )          (a 1x1 conv commutes with a per-channel resize)
//
// so the nine taps W_t are applied as ONE 1x1 conv at LOW resolution (a plain GEMM [N*h*w, Cin] x [Cin, 9*Cout]: 2.25 multiplies per output
// pixel and (ci, co) pair -- the same count as Winograd F(4x4,3x3), but its A operand is the low-resolution tensor read once instead
// of a transformed tensor nine times its size), and this kernel finishes the job: for every output pixel the nine tap planes are
// sampled at the shifted position of the (never materialised) upsampled image -- zero outside it, which is the conv's padding;
// ATen's clamped bilinear taps inside it -- summed, and sent through the folded BN + activation epilogue.
//
// Exact x2, align_corners=False: upsampled row 2y is .25 L[y-1] + .75 L[y], row 2y+1 is .75 L[y] + .25 L[y+1] with the row index clamped
// to the image (ATen clamps the source coordinate at 0 and the second tap at h-1; the clamped blend equals it to an ulp).  All four
// outputs of the 2x2 block of low-resolution pixel (y, x) depend on the 3x3 low-resolution neighbourhood only, separably:
//   H_ky(r, X) = sum_kx UpX(z_{ky,kx}(r, .))(X + kx - 1)        per low-resolution row r, both X = 2x, 2x+1
//   out(Y, X)  = sum_ky UpY(H_ky(., X))(Y + ky - 1)             both Y = 2y, 2y+1
// A thread owns a low-resolution column x and four channels and walks a strip of rows keeping H of three rows in registers: 27 16-byte
// loads per low-resolution pixel, the 3x column re-use is left to L1/L2.  Memory-bound: z is read once from HBM, out written once.
#include "arseg_common.h"

namespace {

__device__ __forceinline__ float up_act(float v, int act, float slope) {
    switch (act) {
        case ARSEG_ACT_RELU: return fmaxf(v, 0.0f);
        case ARSEG_ACT_PRELU: return v >= 0.0f ? v : v * slope;
        case ARSEG_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
        default: return v;
    }
}

// SPLIT: `out` is written as split rows (the operand format of arseg_gemm_x3_fwd, csrc/gemm_x3.hip; out_ld == C, C % 32 == 0) -- the output is
// the next tap GEMM's activation operand (up_1 -> up_2) -- and the running |out| maximum feeds the operand range word.
template <bool SPLIT>
__global__ __launch_bounds__(256) void up2_tap_gather_kernel(const float *__restrict__ z, int z_ld, const float *__restrict__ scale,
                                                             const float *__restrict__ bias, float *__restrict__ out, int out_ld, int N, int h,
                                                             int w, int C, int rs, int act, float slope, unsigned *range_flag, float range_limit) {
    const int Cv = C >> 2, strips = (h + rs - 1) / rs;
    const long long total = (long long)N * strips * w * Cv;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % Cv) * 4;
    long long q = idx / Cv;
    const int x = (int)(q % w); q /= w;
    const int s = (int)(q % strips);
    const int n = (int)(q / strips);
    const int col[3] = {max(x - 1, 0), x, min(x + 1, w - 1)};
    const bool vL = x >= 1, vR = x + 1 < w;
    const float *zn = z + (size_t)n * h * w * z_ld + c;

    auto hrow = [&](int r, f32x4 (&H)[3][2]) {
        const float *zr = zn + (size_t)r * w * z_ld;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            f32x4 a[3][3];                                   // [kx][column x-1, x, x+1]
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int j = 0; j < 3; ++j) a[kx][j] = *reinterpret_cast<const f32x4 *>(zr + (size_t)col[j] * z_ld + (ky * 3 + kx) * C);
            // X = 2x: upsampled columns 2x-1 (outside the image for x == 0), 2x, 2x+1
            f32x4 e = (0.25f * a[1][0] + 0.75f * a[1][1]) + (0.75f * a[2][1] + 0.25f * a[2][2]);
            const f32x4 el = 0.75f * a[0][0] + 0.25f * a[0][1];
            H[ky][0] = vL ? e + el : e;
            // X = 2x+1: upsampled columns 2x, 2x+1, 2x+2 (outside for x == w-1)
            f32x4 o = (0.25f * a[0][0] + 0.75f * a[0][1]) + (0.75f * a[1][1] + 0.25f * a[1][2]);
            const f32x4 orr = 0.25f * a[2][1] + 0.75f * a[2][2];
            H[ky][1] = vR ? o + orr : o;
        }
    };

    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
    if (scale) sc = *reinterpret_cast<const f32x4 *>(scale + c);
    if (bias) bi = *reinterpret_cast<const f32x4 *>(bias + c);
    const int y0 = s * rs, y1 = min(y0 + rs, h);
    f32x4 Hm[3][2], H0[3][2], Hp[3][2];
    hrow(max(y0 - 1, 0), Hm);
    hrow(y0, H0);
    float *on = out + (size_t)n * (2 * h) * (2 * w) * out_ld + (SPLIT ? 0 : c);
    float vmax = 0.f;
    auto put = [&](size_t pix, const f32x4 v) {
        if constexpr (SPLIT) {
            unsigned h01, h23, l01, l23;
            arseg_split_f16(v, h01, h23, l01, l23);
            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
            unsigned char *o = reinterpret_cast<unsigned char *>(on + pix * out_ld) + (c >> 5) * 128 + (c & 31) * 2;
            *reinterpret_cast<uint2 *>(o) = uint2{h01, h23};
            *reinterpret_cast<uint2 *>(o + 64) = uint2{l01, l23};
        } else {
            *reinterpret_cast<f32x4 *>(on + pix * out_ld) = v;
        }
    };
    for (int y = y0; y < y1; ++y) {
        hrow(min(y + 1, h - 1), Hp);
        const bool vT = y >= 1, vB = y + 1 < h;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            // Y = 2y: upsampled rows 2y-1 (outside for y == 0), 2y, 2y+1;   Y = 2y+1: rows 2y, 2y+1, 2y+2 (outside for y == h-1)
            f32x4 o0 = (0.25f * Hm[1][b] + 0.75f * H0[1][b]) + (0.75f * H0[2][b] + 0.25f * Hp[2][b]);
            const f32x4 t0 = 0.75f * Hm[0][b] + 0.25f * H0[0][b];
            o0 = vT ? o0 + t0 : o0;
            f32x4 o1 = (0.25f * Hm[0][b] + 0.75f * H0[0][b]) + (0.75f * H0[1][b] + 0.25f * Hp[1][b]);
            const f32x4 t1 = 0.25f * H0[2][b] + 0.75f * Hp[2][b];
            o1 = vB ? o1 + t1 : o1;
            o0 = o0 * sc + bi;
            o1 = o1 * sc + bi;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o0[e] = up_act(o0[e], act, slope); o1[e] = up_act(o1[e], act, slope); }
            put((size_t)(2 * y) * (2 * w) + 2 * x + b, o0);
            put((size_t)(2 * y + 1) * (2 * w) + 2 * x + b, o1);
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int b = 0; b < 2; ++b) { Hm[ky][b] = H0[ky][b]; H0[ky][b] = Hp[ky][b]; }
    }
    if (SPLIT && range_flag && vmax > range_limit) atomicOr(range_flag, 1u);
}

}  // namespace

static int tap_gather(const float *z, int z_ld, const float *scale, const float *bias, float *out, int out_ld, int N, int h, int w, int Cout, int act,
                      float prelu_slope, bool split, void *range_flag, float range_limit, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(z); ARSEG_CHECK_PTR(out); ARSEG_CHECK_POS(N); ARSEG_CHECK_POS(h); ARSEG_CHECK_POS(w); ARSEG_CHECK_POS(Cout);
    if ((Cout & 3) || z_ld < 9 * Cout || (z_ld & 3) || out_ld < Cout || (out_ld & 3)) return ARSEG_EINVAL;
    if (!ARSEG_ALIGNED16(z) || !ARSEG_ALIGNED16(out) || (scale && !ARSEG_ALIGNED16(scale)) || (bias && !ARSEG_ALIGNED16(bias))) return ARSEG_EINVAL;
    if (split && ((Cout & 31) || out_ld != Cout || (reinterpret_cast<uintptr_t>(range_flag) & 3))) return ARSEG_EINVAL;
    // strip length: the longest (fewest re-computed boundary rows: (rs + 2) / rs) that still gives every SIMD a couple of waves
    const long long per_strip = (long long)N * w * (Cout / 4);
    int rs = 16;
    while (rs > 2 && per_strip * ((h + rs - 1) / rs) < 2048ll * 64) rs >>= 1;
    const long long total = per_strip * ((h + rs - 1) / rs);
    if ((total + 255) / 256 >= (1ll << 31)) return ARSEG_EUNSUPPORTED;
    unsigned *rf = reinterpret_cast<unsigned *>(range_flag);
    const float rl = range_limit > 0.0f ? range_limit : 65504.0f;
    if (split)
        hipLaunchKernelGGL(up2_tap_gather_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, arseg_stream(stream), z, z_ld, scale, bias,
                           out, out_ld, N, h, w, Cout, rs, act, prelu_slope, rf, rl);
    else
        hipLaunchKernelGGL(up2_tap_gather_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, arseg_stream(stream), z, z_ld, scale, bias,
                           out, out_ld, N, h, w, Cout, rs, act, prelu_slope, rf, rl);
    return arseg_launch_status();
}

extern "C" int arseg_upconv3x3_tap_gather_fwd(const float *z, int z_ld, const float *scale, const float *bias, float *out, int out_ld, int N,
                                              int h, int w, int Cout, int act, float prelu_slope, arseg_stream_t stream) {
    return tap_gather(z, z_ld, scale, bias, out, out_ld, N, h, w, Cout, act, prelu_slope, false, nullptr, 0.0f, stream);
}

extern "C" int arseg_upconv3x3_tap_gather_split_fwd(const float *z, int z_ld, const float *scale, const float *bias, void *out_split, int N, int h,
                                                    int w, int Cout, int act, float prelu_slope, void *range_flag, float range_limit,
                                                    arseg_stream_t stream) {
    return tap_gather(z, z_ld, scale, bias, reinterpret_cast<float *>(out_split), Cout, N, h, w, Cout, act, prelu_slope, true, range_flag, range_limit, stream);
}
