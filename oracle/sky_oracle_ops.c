/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Plain-C (OpenMP) restatement of the dense tensor operators the reference's
 * forward pass relies on, in the reference's own layout (NCHW, fp32).  The
 * graph wiring that calls these lives in oracle/skyeye_oracle.py, one function
 * per reference module with file:line citations.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Semantics follow the PyTorch defaults the reference uses:
 *   conv2d     cross-correlation, zero padding, dilation 1, groups 1
 *              (skyeye/core/models/blocks.py:28-31 -> nn.Conv2d)
 *   bn_act     BatchNorm2d in eval mode, eps 1e-5, then SiLU or identity
 *              (blocks.py:32-37)
 *   maxpool2d  kernel k, stride 1, padding k/2, implicit -inf padding
 *              (blocks.py:142-144)
 *   upsample   F.interpolate(mode='nearest'): src = floor(dst*in/out)
 *              (skyeye/core/models/detector.py:214,218)
 *   bilinear   F.interpolate(mode='bilinear', align_corners=False)
 *              (skyeye/core/models/attention.py:211-212)
 *
 * Pinned against tests/golden/blocks.npz and detectors_*.npz, which were
 * produced by the reference classes themselves (tests/golden/make_golden.py).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define CB 8    /* output channels per work item */
#define BAND 8  /* output rows per work item     */

int sky_oracle_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void sky_oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* y[b,co,oy,ox] = bias[co] + sum_{ci,ky,kx} w[co,ci,ky,kx] * x[b,ci,oy*s+ky-p,ox*s+kx-p] */
void sky_oracle_conv2d(const float* x, const float* w, const float* bias, float* y,
                       int B, int Cin, int H, int W, int Cout, int K, int stride, int pad)
{
    const int Ho = (H + 2 * pad - K) / stride + 1;
    const int Wo = (W + 2 * pad - K) / stride + 1;
    const int ncob = (Cout + CB - 1) / CB;
    const int nband = (Ho + BAND - 1) / BAND;
    const long items = (long)B * ncob * nband;
#pragma omp parallel for schedule(dynamic, 1)
    for (long it = 0; it < items; ++it) {
        const int band = (int)(it % nband);
        const int cob = (int)((it / nband) % ncob);
        const int b = (int)(it / ((long)nband * ncob));
        const int co0 = cob * CB, co1 = co0 + CB < Cout ? co0 + CB : Cout;
        const int oy0 = band * BAND, oy1 = oy0 + BAND < Ho ? oy0 + BAND : Ho;
        for (int co = co0; co < co1; ++co)
            for (int oy = oy0; oy < oy1; ++oy) {
                float* yr = y + (((size_t)b * Cout + co) * Ho + oy) * Wo;
                const float bv = bias ? bias[co] : 0.0f;
                for (int ox = 0; ox < Wo; ++ox) yr[ox] = bv;
            }
        for (int ci = 0; ci < Cin; ++ci)
            for (int ky = 0; ky < K; ++ky)
                for (int kx = 0; kx < K; ++kx) {
                    /* valid ox range: 0 <= ox*stride + kx - pad < W */
                    int ox0 = 0, ox1 = Wo;
                    while (ox0 < Wo && ox0 * stride + kx - pad < 0) ++ox0;
                    while (ox1 > ox0 && (ox1 - 1) * stride + kx - pad >= W) --ox1;
                    for (int oy = oy0; oy < oy1; ++oy) {
                        const int iy = oy * stride + ky - pad;
                        if (iy < 0 || iy >= H) continue;
                        const float* xr = x + (((size_t)b * Cin + ci) * H + iy) * W + (kx - pad);
                        for (int co = co0; co < co1; ++co) {
                            const float wv = w[(((size_t)co * Cin + ci) * K + ky) * K + kx];
                            float* yr = y + (((size_t)b * Cout + co) * Ho + oy) * Wo;
                            if (stride == 1) {
                                for (int ox = ox0; ox < ox1; ++ox) yr[ox] += wv * xr[ox];
                            } else {
                                for (int ox = ox0; ox < ox1; ++ox) yr[ox] += wv * xr[ox * stride];
                            }
                        }
                    }
                }
    }
}

/* in-place eval BatchNorm (+ optional SiLU).  act: 0 identity, 1 SiLU */
void sky_oracle_bn_act(float* x, const float* gamma, const float* beta, const float* mean, const float* var,
                       float eps, int B, int C, int HW, int act)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            const float inv = 1.0f / sqrtf(var[c] + eps);
            const float a = gamma[c] * inv;
            const float s = beta[c] - mean[c] * a;
            float* p = x + ((size_t)b * C + c) * HW;
            if (act) {
                for (int i = 0; i < HW; ++i) {
                    const float v = p[i] * a + s;
                    p[i] = v / (1.0f + expf(-v));
                }
            } else {
                for (int i = 0; i < HW; ++i) p[i] = p[i] * a + s;
            }
        }
}

/* elementwise helpers (in place).  kind: 0 SiLU, 1 sigmoid, 2 ReLU */
void sky_oracle_act(float* x, size_t n, int kind)
{
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) {
        const float v = x[i];
        x[i] = kind == 0 ? v / (1.0f + expf(-v)) : kind == 1 ? 1.0f / (1.0f + expf(-v)) : (v > 0.0f ? v : 0.0f);
    }
}

void sky_oracle_add(float* y, const float* a, const float* b, size_t n)
{
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) y[i] = a[i] + b[i];
}

/* MaxPool2d(k, stride=1, padding=k/2) with -inf padding */
void sky_oracle_maxpool2d(const float* x, float* y, int B, int C, int H, int W, int k)
{
    const int p = k / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            const float* xp = x + ((size_t)b * C + c) * H * W;
            float* yp = y + ((size_t)b * C + c) * H * W;
            for (int oy = 0; oy < H; ++oy)
                for (int ox = 0; ox < W; ++ox) {
                    float m = -INFINITY;
                    for (int dy = -p; dy <= p; ++dy) {
                        const int iy = oy + dy;
                        if (iy < 0 || iy >= H) continue;
                        for (int dx = -p; dx <= p; ++dx) {
                            const int ix = ox + dx;
                            if (ix < 0 || ix >= W) continue;
                            const float v = xp[iy * W + ix];
                            m = v > m ? v : m;
                        }
                    }
                    yp[oy * W + ox] = m;
                }
        }
}

/* F.interpolate(size=(Ho,Wo), mode='nearest') */
void sky_oracle_upsample_nearest(const float* x, float* y, int B, int C, int H, int W, int Ho, int Wo)
{
    const float sh = (float)H / (float)Ho, sw = (float)W / (float)Wo;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            const float* xp = x + ((size_t)b * C + c) * H * W;
            float* yp = y + ((size_t)b * C + c) * Ho * Wo;
            for (int oy = 0; oy < Ho; ++oy) {
                int iy = (int)floorf((float)oy * sh);
                if (iy > H - 1) iy = H - 1;
                for (int ox = 0; ox < Wo; ++ox) {
                    int ix = (int)floorf((float)ox * sw);
                    if (ix > W - 1) ix = W - 1;
                    yp[oy * Wo + ox] = xp[iy * W + ix];
                }
            }
        }
}

/* F.interpolate(size=(Ho,Wo), mode='bilinear', align_corners=False) */
void sky_oracle_bilinear(const float* x, float* y, int B, int C, int H, int W, int Ho, int Wo)
{
    const float sh = (float)H / (float)Ho, sw = (float)W / (float)Wo;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            const float* xp = x + ((size_t)b * C + c) * H * W;
            float* yp = y + ((size_t)b * C + c) * Ho * Wo;
            for (int oy = 0; oy < Ho; ++oy) {
                float fy = ((float)oy + 0.5f) * sh - 0.5f;
                if (fy < 0.0f) fy = 0.0f;
                int y0 = (int)fy;
                int y1 = y0 + (y0 < H - 1 ? 1 : 0);
                const float ly = fy - (float)y0, hy = 1.0f - ly;
                for (int ox = 0; ox < Wo; ++ox) {
                    float fx = ((float)ox + 0.5f) * sw - 0.5f;
                    if (fx < 0.0f) fx = 0.0f;
                    int x0 = (int)fx;
                    int x1 = x0 + (x0 < W - 1 ? 1 : 0);
                    const float lx = fx - (float)x0, hx = 1.0f - lx;
                    yp[oy * Wo + ox] = hy * (hx * xp[y0 * W + x0] + lx * xp[y0 * W + x1]) +
                                       ly * (hx * xp[y1 * W + x0] + lx * xp[y1 * W + x1]);
                }
            }
        }
}

/* y[m,n] = bias[n] + sum_k x[m,k] * w[n,k]   (nn.Linear) */
void sky_oracle_linear(const float* x, const float* w, const float* bias, float* y, int M, int K, int N)
{
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            float acc = bias ? bias[n] : 0.0f;
            const float* xr = x + (size_t)m * K;
            const float* wr = w + (size_t)n * K;
            for (int k = 0; k < K; ++k) acc += xr[k] * wr[k];
            y[(size_t)m * N + n] = acc;
        }
}
