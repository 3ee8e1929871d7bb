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
