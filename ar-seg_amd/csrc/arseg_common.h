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
static inline hipStream_t arseg_stream(arseg_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
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
