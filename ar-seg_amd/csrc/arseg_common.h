// Shared helpers for the gfx950 kernels of libarseg_hip.so (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "arseg_hip.h"

#define ARSEG_CHECK_PTR(p) do { if ((p) == nullptr) return ARSEG_EINVAL; } while (0)
#define ARSEG_CHECK_POS(v) do { if ((v) <= 0) return ARSEG_EINVAL; } while (0)
#define ARSEG_ALIGNED16(p) ((reinterpret_cast<uintptr_t>(p) & 15u) == 0)

static inline int arseg_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ARSEG_OK : (int)e;
}
// Every entry point converts its stream handle right before it launches: that is also where a stale error of the calling thread is dropped
// (hipGetLastError is per thread and sticky: an unrelated earlier HIP call of the host program -- e.g. a device probe that returned
// hipErrorNoDevice before the runtime was initialised -- would otherwise be reported as this launch's status).
static inline hipStream_t arseg_stream(arseg_stream_t s) {
    (void)hipGetLastError();
    return reinterpret_cast<hipStream_t>(s);
}
static inline int arseg_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE function attribute: remember, per device, the largest request that has
// been granted (a process may drive several GPUs, e.g. under nn.DataParallel).  Grow-only; a race only repeats the same call.
struct ArsegSmemAttr { size_t granted[32] = {}; };
static inline int arseg_allow_smem(ArsegSmemAttr &a, const void *kernel, size_t smem) {
    int dev = -1;
    const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 32;
    if (known && smem <= a.granted[dev]) return ARSEG_OK;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    if (known) a.granted[dev] = smem;
    return ARSEG_OK;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef __HIPCC__
// 16-bit storage element <-> fp32 (BF = true: bfloat16, false: IEEE fp16); conversions to 16 bits round to nearest even
template <bool BF>
__device__ __forceinline__ float arseg_h2f(uint16_t v) {
    if constexpr (BF) return __uint_as_float((unsigned)v << 16);
    else return (float)__builtin_bit_cast(_Float16, v);
}
template <bool BF>
__device__ __forceinline__ uint16_t arseg_f2h(float x) {
    // (r6) bf16: the hardware conversion of gfx950 (v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN) instead of the integer sequence of
    // rounds 2-5 (and / compare / add / shift: ~8 VALU per value -- the epilogue arithmetic of the bf16 stem kernel was as long as its MFMAs)
    if constexpr (BF) return __builtin_bit_cast(uint16_t, (__bf16)x);
    else return __builtin_bit_cast(uint16_t, (_Float16)x);
}
// two values -> one packed dword (low half = a): one v_cvt_pk_bf16_f32 / v_cvt_pkrtz-free v_cvt_pk for fp16
template <bool BF>
__device__ __forceinline__ unsigned arseg_f2h_pair(float a, float b) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    if constexpr (BF) {
        typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_{a, b}, bf16x2_));
    } else {
        typedef _Float16 h16x2_ __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_{a, b}, h16x2_));
    }
}

// fp32 -> two fp16: x = hi + lo with hi = fp16(x) rounded toward zero (11 significant bits) and lo = fp16(x - hi), 22 bits together.
// lo is formed from the fp16 value actually stored (v_fma_mix_f32 reads the packed half directly), so
//   * |x| up to 131008 = 2 x 65504 is still represented (hi saturates at 65504, lo takes the rest; full 22 bits below 65504);
//     beyond that the pair clamps -- the documented range of the split-fp16 back ends (include/arseg_hip.h);
//   * |x| below the fp16 normal range degrades gracefully: the absolute error is at most 2^-24 (6e-8).
// 8 VALU instructions per 4 values (2 cvt_pkrtz + 4 fma_mix + 2 cvt_pkrtz); the mask / subtract / convert form took 12.
__device__ __forceinline__ void arseg_split_f16(const f32x4 v, unsigned &h01, unsigned &h23, unsigned &l01, unsigned &l23) {
    float l0, l1, l2, l3;
    asm("v_cvt_pkrtz_f16_f32 %0, %6, %7\n\tv_cvt_pkrtz_f16_f32 %1, %8, %9\n\t"
        "v_fma_mix_f32 %2, %0, -1.0, %6 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %3, %0, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %4, %1, -1.0, %8 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %5, %1, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(h01), "=&v"(h23), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
    l01 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(l0, l1));
    l23 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(l2, l3));
}
// two values: (a, b) -> packed hi halves, packed lo halves (4 instructions)
__device__ __forceinline__ void arseg_split_f16_pair(float a, float b, unsigned &h, unsigned &l) {
    float l0, l1;
    asm("v_cvt_pkrtz_f16_f32 %0, %3, %4\n\tv_fma_mix_f32 %1, %0, -1.0, %3 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %2, %0, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(h), "=&v"(l0), "=&v"(l1) : "v"(a), "v"(b));
    l = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(l0, l1));
}
#endif

// Bilinear source coordinate exactly as ATen computes it for fp32 tensors
// (area_pixel_compute_source_index): align_corners -> scale = (in-1)/(out-1) (0 when out == 1),
// src = scale*dst; otherwise scale = in/out, src = max(scale*(dst+0.5)-0.5, 0).
__host__ __device__ inline float arseg_resize_scale(int in, int out, bool align) {
    if (align) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.0f;
    return (float)in / (float)out;
}
__device__ inline void arseg_src_index(float scale, int dst, bool align, int in, int &i0, int &i1, float &l1) {
    float src = align ? scale * (float)dst : fmaxf(scale * ((float)dst + 0.5f) - 0.5f, 0.0f);
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - (float)i0;
}
