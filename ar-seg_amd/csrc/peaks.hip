// On-box roofline denominators ("datasheet AND on-box measured stream-copy / MFMA micro-benchmarks; fractions
// reported against both").  Two measurement aids, not part of the hot path and with no counterpart in the reference:
//   arseg_peak_stream_copy : HBM -> HBM copy of n_bytes with 16-byte accesses (read + write = 2 n_bytes of traffic per launch)
//   arseg_peak_mfma_f16    : every wave of a full-chip launch issues `iters` x 8 independent v_mfma_f32_32x32x16_f16 (no memory traffic)
// bench.py times each with HIP events once per run and prints `peaks_measured`.
#include "arseg_common.h"

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void peak_copy_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n16) {
    // grid-stride, four 16-byte loads in flight per thread, consecutive lanes on consecutive addresses
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
        const u32x4 c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
        __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + stride);
        __builtin_nontemporal_store(c, dst + i + 2 * stride); __builtin_nontemporal_store(d, dst + i + 3 * stride);
    }
    for (; i < n16; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

__global__ __launch_bounds__(256) void peak_mfma_kernel(float *out, int iters) {
    // 8 independent accumulator tiles per wave (128 accumulator registers): back-to-back issue without a dependent MFMA in the pipe
    h16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (float)((threadIdx.x + i) & 7)); b[i] = (_Float16)(0.002f * (float)((threadIdx.x * 3 + i) & 7)); }
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t)
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < 8; ++t)
        for (int i = 0; i < 16; ++i) s += acc[t][i];
    if (s == 12345.678f) out[0] = s;      // (never true: keeps the loop alive without a store per thread)
}

}  // namespace

extern "C" int arseg_peak_stream_copy(const void *src, void *dst, size_t n_bytes, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(src); ARSEG_CHECK_PTR(dst);
    if (n_bytes < 16 || (n_bytes & 15) || !ARSEG_ALIGNED16(src) || !ARSEG_ALIGNED16(dst)) return ARSEG_EINVAL;
    hipStream_t st = arseg_stream(stream);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    hipLaunchKernelGGL(peak_copy_kernel, dim3(cus * 16), dim3(256), 0, st, (const u32x4 *)src, (u32x4 *)dst, n_bytes / 16);
    return arseg_launch_status();
}

// flops issued by one launch = waves x iters x 8 x (2 * 32 * 32 * 16); *flops_out receives that number (may be NULL)
extern "C" int arseg_peak_mfma_f16(float *scratch, int iters, double *flops_out, arseg_stream_t stream) {
    ARSEG_CHECK_PTR(scratch); ARSEG_CHECK_POS(iters);
    hipStream_t st = arseg_stream(stream);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const int wgs = cus * 4;                     // 4 workgroups x 4 waves per CU: four waves per SIMD
    if (flops_out) *flops_out = (double)wgs * 4.0 * (double)iters * 8.0 * (2.0 * 32 * 32 * 16);
    hipLaunchKernelGGL(peak_mfma_kernel, dim3(wgs), dim3(256), 0, st, scratch, iters);
    return arseg_launch_status();
}
