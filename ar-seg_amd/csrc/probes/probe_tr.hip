// Hardware probe (not part of libarseg_hip.so): pins the two gfx950 behaviours creff_mfma.hip relies on.
//   1. ds_read_b64_tr_b16 (LDS transpose read): within a 16-lane group, lane i supplies the address of the 8-byte piece
//      (row 4g + i/4, 16-bit columns 4*(i%4)..+3) and receives column i of the group's four rows:
//      out[i][j] = halfword (i % 4) of the piece supplied by lane 4j + i/4.  Any row stride works (80 bytes here).
//   2. v_mfma_f32_16x16x32_f16 operand layout: A[l%16][8*(l/16)+j], B[8*(l/16)+j][l%16], D[4*(l/16)+r][l%16].
// Build and run on an MI355X:  hipcc --offload-arch=gfx950 -O2 probe_tr.hip -o probe_tr && ./probe_tr   (exit code 0 = as expected)
// tests/test_gpu_ops.py::test_hw_probe_transpose_read_and_mfma_layout does exactly that.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void tr_probe(float *out) {
    __shared__ __attribute__((aligned(16))) _Float16 img[64 * 40];
    const int l = threadIdx.x;
    for (int i = l; i < 64 * 40; i += 64) img[i] = (_Float16)((i / 40) * 32 + (i % 40));   // value = row*32 + col, row stride 40 halves
    __syncthreads();
    const int g = l >> 4, i = l & 15;
    const _Float16 *addr = img + (4 * g + (i >> 2)) * 40 + (i & 3) * 4;
    const h16x4 v = __builtin_bit_cast(h16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)addr));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (float)v[j];
}

__global__ void mfma_probe(float *out) {
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    h16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(float)i; b[j] = (_Float16)((8 * g + j == 5) ? (float)(i + 1) : 0.0f); }
    f32x4 c = {0, 0, 0, 0};                                  // A[row i][k] = i, B[k][col n] = (k == 5) * (n + 1)  ->  D[r][c] = r * (c + 1)
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = c[j];
}

int main() {
    float *d, h[256];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 2;
    int bad = 0;
    hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) bad += h[l * 4 + j] != (float)((4 * (l >> 4) + j) * 32 + (l & 15));        // row 4g+j, column i
    printf("transpose read: %d mismatches\n", bad);
    int bad2 = 0;
    hipLaunchKernelGGL(mfma_probe, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) bad2 += h[l * 4 + j] != (float)((4 * (l >> 4) + j) * ((l & 15) + 1));
    printf("mfma 16x16x32 f16 layout: %d mismatches\n", bad2);
    return (bad || bad2) ? 1 : 0;
}
