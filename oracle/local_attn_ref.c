/* ORACLE -- TEST INFRASTRUCTURE ONLY (never linked into or called from the product path).
 *
 * Plain-C restatement of the two forward functions of the third-party `localAttention`
 * extension (github.com/zzd1992/Image-Local-Attention, pinned only as `@master` in the
 * reference's requirements.txt:7 / environment.yml:62 and NOT vendored under /root/reference).
 * Call sites in the reference: model/attention.py:18 (similar_forward) and :38
 * (weighting_forward), used on the hot path at model/attention.py:199,207.
 *
 * Contract restated (from the reference's in-tree unfold restatements, model/attention.py:55-64
 * layout comments and :75-85 `f_weighting_cpu`, and the published algorithm of the op):
 *   similar  : S[n][y][x][dy*kW+dx] = sum_c Q[n][c][y][x] * K[n][c][y+dy-kH/2][x+dx-kW/2]
 *   weighting: O[n][c][y][x]        = sum_i V[n][c][y+dy_i-kH/2][x+dx_i-kW/2] * W[n][y][x][i]
 * Taps that fall outside the image contribute 0 (zero padding), fp32 accumulation in
 * channel / tap order, no 1/sqrt(C) scaling.  Inputs NCHW contiguous, S/W are [N,H,W,kH*kW].
 * `weighting` is pinned against f_weighting_cpu (tests/golden); `similar` is parity-unpinned
 * with respect to the absent CUDA source (see oracle/cpu_ref.py header).
 */
#include <stddef.h>

void oracle_local_similar(const float *q, const float *k, float *s,
                          int N, int C, int H, int W, int kH, int kW)
{
    const int rH = kH / 2, rW = kW / 2, T = kH * kW;
    const size_t plane = (size_t)H * W;
    for (int n = 0; n < N; ++n)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float *so = s + (((size_t)n * H + y) * W + x) * T;
                for (int dy = 0; dy < kH; ++dy)
                    for (int dx = 0; dx < kW; ++dx) {
                        const int yy = y + dy - rH, xx = x + dx - rW;
                        float acc = 0.0f;
                        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                            const float *qp = q + (size_t)n * C * plane + (size_t)y * W + x;
                            const float *kp = k + (size_t)n * C * plane + (size_t)yy * W + xx;
                            for (int c = 0; c < C; ++c)
                                acc += qp[c * plane] * kp[c * plane];
                        }
                        so[dy * kW + dx] = acc;
                    }
            }
}

void oracle_local_weighting(const float *v, const float *w, float *o,
                            int N, int C, int H, int W, int kH, int kW)
{
    const int rH = kH / 2, rW = kW / 2, T = kH * kW;
    const size_t plane = (size_t)H * W;
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            const float *vp = v + ((size_t)n * C + c) * plane;
            float *op = o + ((size_t)n * C + c) * plane;
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    const float *wp = w + (((size_t)n * H + y) * W + x) * T;
                    float acc = 0.0f;
                    for (int dy = 0; dy < kH; ++dy)
                        for (int dx = 0; dx < kW; ++dx) {
                            const int yy = y + dy - rH, xx = x + dx - rW;
                            if (yy >= 0 && yy < H && xx >= 0 && xx < W)
                                acc += vp[(size_t)yy * W + xx] * wp[dy * kW + dx];
                        }
                    op[(size_t)y * W + x] = acc;
                }
        }
}

/* ---------------------------------------------------------------------------------------------
 * The same two functions for the timed CPU leg of bench.py (the CPU path runs on every host core): rows of the
 * image in parallel (OpenMP), the inner loops over x so that they vectorise.  Same products, same accumulation order per output
 * (channels ascending for `similar`, taps in row-major order for `weighting`), contraction off (Makefile: -ffp-contract=off): the
 * results equal the scalar functions above bit for bit (tests/test_oracle_golden.py::test_c_oracle_mt_equals_scalar).
 * ------------------------------------------------------------------------------------------- */
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_local_similar_mt(const float *q, const float *k, float *s,
                             int N, int C, int H, int W, int kH, int kW, int nthreads)
{
    const int rH = kH / 2, rW = kW / 2, T = kH * kW;
    const size_t plane = (size_t)H * W;
    /* (a num_threads clause, not omp_set_num_threads: the OpenMP runtime is shared with torch, whose thread count must not change) */
    const int nt = nthreads > 0 ? nthreads : oracle_max_threads();
#pragma omp parallel num_threads(nt)
    {
        float *acc = (float *)malloc((size_t)T * W * sizeof(float));      /* one row of scores, tap-major */
#pragma omp for collapse(2) schedule(static)
        for (int n = 0; n < N; ++n)
            for (int y = 0; y < H; ++y) {
                memset(acc, 0, (size_t)T * W * sizeof(float));
                for (int c = 0; c < C; ++c) {
                    const float *qr = q + ((size_t)n * C + c) * plane + (size_t)y * W;
                    for (int dy = 0; dy < kH; ++dy) {
                        const int yy = y + dy - rH;
                        if (yy < 0 || yy >= H) continue;
                        const float *kr = k + ((size_t)n * C + c) * plane + (size_t)yy * W;
                        for (int dx = 0; dx < kW; ++dx) {
                            float *a = acc + (size_t)(dy * kW + dx) * W;
                            const int off = dx - rW, x0 = off < 0 ? -off : 0, x1 = off > 0 ? W - off : W;
                            for (int x = x0; x < x1; ++x) a[x] += qr[x] * kr[x + off];
                        }
                    }
                }
                float *so = s + (((size_t)n * H + y) * W) * T;
                for (int t = 0; t < T; ++t)
                    for (int x = 0; x < W; ++x) so[(size_t)x * T + t] = acc[(size_t)t * W + x];
            }
        free(acc);
    }
}

void oracle_local_weighting_mt(const float *v, const float *w, float *o,
                               int N, int C, int H, int W, int kH, int kW, int nthreads)
{
    const int rH = kH / 2, rW = kW / 2, T = kH * kW;
    const size_t plane = (size_t)H * W;
    /* (a num_threads clause, not omp_set_num_threads: the OpenMP runtime is shared with torch, whose thread count must not change) */
    const int nt = nthreads > 0 ? nthreads : oracle_max_threads();
#pragma omp parallel num_threads(nt)
    {
        float *wt = (float *)malloc((size_t)T * W * sizeof(float));       /* one row of weights, tap-major */
#pragma omp for collapse(2) schedule(static)
        for (int n = 0; n < N; ++n)
            for (int y = 0; y < H; ++y) {
                const float *wr = w + (((size_t)n * H + y) * W) * T;
                for (int t = 0; t < T; ++t)
                    for (int x = 0; x < W; ++x) wt[(size_t)t * W + x] = wr[(size_t)x * T + t];
                for (int c = 0; c < C; ++c) {
                    float *orow = o + ((size_t)n * C + c) * plane + (size_t)y * W;
                    for (int x = 0; x < W; ++x) orow[x] = 0.0f;
                    for (int dy = 0; dy < kH; ++dy) {
                        const int yy = y + dy - rH;
                        if (yy < 0 || yy >= H) continue;
                        const float *vr = v + ((size_t)n * C + c) * plane + (size_t)yy * W;
                        for (int dx = 0; dx < kW; ++dx) {
                            const float *a = wt + (size_t)(dy * kW + dx) * W;
                            const int off = dx - rW, x0 = off < 0 ? -off : 0, x1 = off > 0 ? W - off : W;
                            for (int x = x0; x < x1; ++x) orow[x] += vr[x + off] * a[x];
                        }
                    }
                }
            }
        free(wt);
    }
}
