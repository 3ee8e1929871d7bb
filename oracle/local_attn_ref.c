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
