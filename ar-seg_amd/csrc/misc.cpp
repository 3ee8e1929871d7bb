// Version / status strings of libarseg_hip.so (host only).
#include "arseg_hip.h"

extern "C" int arseg_version(void) { return ARSEG_ABI_VERSION; }

extern "C" const char *arseg_status_string(int status) {
    switch (status) {
        case ARSEG_OK: return "ok";
        case ARSEG_EINVAL: return "invalid argument (null pointer, non-positive size, misaligned pointer or stride)";
        case ARSEG_EUNSUPPORTED: return "shape not supported by the compiled kernels";
        case ARSEG_EWORKSPACE: return "workspace missing or too small";
        default: return status > 0 ? "HIP runtime error (value is the hipError_t)" : "unknown arseg status";
    }
}

// ---------------------------------------------------------------------------------------------
// host-side weight packer of the 16-bit storage path (conv16.hip): OIHW fp32 -> [Cout][Kpad16] fp16 / bf16, round to nearest even
// ---------------------------------------------------------------------------------------------
#include <stdint.h>
#include <string.h>

static uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

static uint16_t f32_to_f16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const uint32_t a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (a > 0x7f800000u ? 0x200u : 0u));      // inf / NaN
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                         // rounds to >= 65520: inf
    if (a < 0x33000001u) return (uint16_t)sign;                                                      // < 2^-25: zero
    int32_t e = (int32_t)(a >> 23) - 127;
    uint32_t man = (a & 0x7fffffu) | 0x800000u;                                                      // 24-bit significand
    int shift = e >= -14 ? 13 : 13 + (-14 - e);                                                      // subnormal halves lose more bits
    uint32_t half = man >> shift, rem = man & ((1u << shift) - 1u), mid = 1u << (shift - 1);
    if (rem > mid || (rem == mid && (half & 1u))) ++half;
    if (e >= -14) return (uint16_t)(sign | (uint32_t)(((e + 15) << 10) + (half - 0x400u)));        // half in [0x400, 0x800]: carry bumps the exponent
    return (uint16_t)(sign | half);
}

extern "C" int arseg_packed_k16(int Cin_pad, int R, int S) { return (R * S * Cin_pad + 63) / 64 * 64; }

extern "C" int arseg_pack_conv_weight16_host(const float *w, int Cout, int Cin, int R, int S, int Cin_pad, int dtype, void *out_host) {
    if (!w || !out_host || Cout <= 0 || Cin <= 0 || R <= 0 || S <= 0 || Cin_pad < Cin || (Cin_pad & 7)) return ARSEG_EINVAL;
    if (dtype != ARSEG_DT_F16 && dtype != ARSEG_DT_BF16) return ARSEG_EINVAL;
    const int Kpad = arseg_packed_k16(Cin_pad, R, S);
    uint16_t *out = (uint16_t *)out_host;
    for (int co = 0; co < Cout; ++co) {
        uint16_t *o = out + (size_t)co * Kpad;
        for (int k = 0; k < Kpad; ++k) o[k] = 0;
        for (int ci = 0; ci < Cin; ++ci)
            for (int r = 0; r < R; ++r)
                for (int s = 0; s < S; ++s) {
                    const float v = w[(((size_t)co * Cin + ci) * R + r) * S + s];
                    o[(r * S + s) * Cin_pad + ci] = dtype == ARSEG_DT_BF16 ? f32_to_bf16_rne(v) : f32_to_f16_rne(v);
                }
    }
    return ARSEG_OK;
}

// ---- the symbol names of SURVEY.md section 8(b), as aliases (include/arseg_hip.h)
extern "C" int arseg_creff_fused_fwd(const float *const *ref_nhwc_host, const int16_t *mv_q, int H, int W, const float *lr, const float *wq,
                                     const float *bq, const float *wk, const float *bk, const float *wv, const float *bv, float *p_out,
                                     int p_layout, const float *wf, const float *bf, int n_cls, float *logits, int log_softmax, int N, int C,
                                     int Hp, int Wp, int hp, int wp, int kH, int kW, arseg_stream_t stream) {
    return arseg_creff_warp_fwd(ref_nhwc_host, mv_q, H, W, lr, wq, bq, wk, bk, wv, bv, p_out, p_layout, wf, bf, n_cls, logits, log_softmax, N, C, Hp,
                                Wp, hp, wp, kH, kW, stream);
}
extern "C" int arseg_conv2d_bn_act_fwd(const arseg_conv_desc *d, const float *in, const float *w_packed, const float *scale, const float *bias,
                                       const float *residual, float *out, void *workspace, size_t workspace_bytes, arseg_stream_t stream) {
    return arseg_conv2d_fwd(d, in, w_packed, scale, bias, residual, out, workspace, workspace_bytes, stream);
}
extern "C" int arseg_pack_weights(const float *w, int Cout, int Cin, int R, int S, int Cin_pad, float *out_host) {
    return arseg_pack_conv_weight_host(w, Cout, Cin, R, S, Cin_pad, out_host);
}
extern "C" int arseg_maxpool3x3s2(const float *in, float *out, int N, int H, int W, int C, arseg_stream_t stream) {
    return arseg_maxpool3x3s2_fwd(in, out, N, H, W, C, stream);
}
extern "C" int arseg_adaptive_avgpool(const float *in, int in_ld, float *out, int out_ld, long long out_n_stride, int N, int H, int W, int C,
                                      int oh, int ow, arseg_stream_t stream) {
    return arseg_adaptive_avgpool_fwd(in, in_ld, out, out_ld, out_n_stride, N, H, W, C, oh, ow, stream);
}
extern "C" int arseg_global_reduce(const float *in, int in_ld, float *out, int N, int H, int W, int C, int op, arseg_stream_t stream) {
    return arseg_global_reduce_fwd(in, in_ld, out, N, H, W, C, op, stream);
}
extern "C" int arseg_resize(const float *in, float *out, int N, int C, int Hin, int Win, int Hout, int Wout, int mode, int align_corners,
                            int layout, int in_ld, int out_ld, arseg_stream_t stream) {
    return arseg_resize_fwd(in, out, N, C, Hin, Win, Hout, Wout, mode, align_corners, layout, in_ld, out_ld, stream);
}
extern "C" int arseg_scale_add(const float *x, const float *scale, const float *add_full, const float *add_vec, float *out, int N, int HW,
                               int C, arseg_stream_t stream) {
    return arseg_scale_add_fwd(x, scale, add_full, add_vec, out, N, HW, C, stream);
}
