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

/* y[b,co,oy,ox] = bias[co] + sum_{ci,ky,kx} w[co,ci,ky,kx] * x[b,ci,oy*s+ky-p,ox*s+kx-p]
 *
 * Direct convolution with a register-tiled micro-kernel: the image is copied once into a zero-padded buffer (no
 * border tests in the hot loop), each work item owns CB = 4 output channels x one output row x TW = 16 output
 * columns held in 8 AVX2 accumulators across the whole (ci, ky, kx) loop, so the inner loop is
 * 2 input loads (+ a de-interleave for stride 2) + 4 weight broadcasts + 8 FMAs.  Accumulation order per output:
 * ci, then ky, then kx, ascending. */
#include <immintrin.h>

#define TW 16

static inline void load_row16(const float* p, int stride, __m256* lo, __m256* hi)
{
    if (stride == 1) {
        *lo = _mm256_loadu_ps(p);
        *hi = _mm256_loadu_ps(p + 8);
    } else { /* stride 2: take the even elements of 32 consecutive floats */
        const __m256 a = _mm256_loadu_ps(p), b = _mm256_loadu_ps(p + 8), c = _mm256_loadu_ps(p + 16), d = _mm256_loadu_ps(p + 24);
        const __m256 ab = _mm256_shuffle_ps(a, b, 0x88), cd = _mm256_shuffle_ps(c, d, 0x88);   /* a0 a2 b0 b2 | a4 a6 b4 b6 */
        *lo = _mm256_castpd_ps(_mm256_permute4x64_pd(_mm256_castps_pd(ab), 0xD8));
        *hi = _mm256_castpd_ps(_mm256_permute4x64_pd(_mm256_castps_pd(cd), 0xD8));
    }
}

void sky_oracle_conv2d(const float* x, const float* w, const float* bias, float* y,
                       int B, int Cin, int H, int W, int Cout, int K, int stride, int pad)
{
    const int Ho = (H + 2 * pad - K) / stride + 1;
    const int Wo = (W + 2 * pad - K) / stride + 1;
    /* padded input: every tile may read up to TW*stride + K columns past its first column */
    const int Wt = (Wo + TW - 1) / TW * TW;
    const int Hp = (Ho - 1) * stride + K, Wp = (Wt - 1) * stride + K + 32;
    float* xp = (float*)calloc((size_t)B * Cin * Hp * Wp, sizeof(float));
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int ci = 0; ci < Cin; ++ci)
            for (int iy = 0; iy < H; ++iy) {
                const int py = iy + pad;
                if (py >= Hp) continue;
                float* dst = xp + (((size_t)b * Cin + ci) * Hp + py) * Wp + pad;
                const float* src = x + (((size_t)b * Cin + ci) * H + iy) * W;
                const int n = W < Wp - pad ? W : Wp - pad;
                memcpy(dst, src, (size_t)n * sizeof(float));
            }
    const int ncob = (Cout + 3) / 4;
    const long items = (long)B * ncob * Ho;
#pragma omp parallel for schedule(dynamic, 4)
    for (long it = 0; it < items; ++it) {
        const int oy = (int)(it % Ho);
        const int cob = (int)((it / Ho) % ncob);
        const int b = (int)(it / ((long)Ho * ncob));
        const int co0 = cob * 4;
        const int nco = Cout - co0 < 4 ? Cout - co0 : 4;
        for (int ox0 = 0; ox0 < Wo; ox0 += TW) {
            __m256 acc[4][2];
            for (int c = 0; c < 4; ++c) {
                const float bv = (bias && c < nco) ? bias[co0 + c] : 0.0f;
                acc[c][0] = acc[c][1] = _mm256_set1_ps(bv);
            }
            for (int ci = 0; ci < Cin; ++ci) {
                const float* xc = xp + (((size_t)b * Cin + ci) * Hp + (size_t)oy * stride) * Wp + (size_t)ox0 * stride;
                const float* wc = w + ((size_t)co0 * Cin + ci) * K * K;
                for (int ky = 0; ky < K; ++ky)
                    for (int kx = 0; kx < K; ++kx) {
                        __m256 lo, hi;
                        load_row16(xc + (size_t)ky * Wp + kx, stride, &lo, &hi);
                        for (int c = 0; c < nco; ++c) {
                            const __m256 wv = _mm256_broadcast_ss(wc + (size_t)c * Cin * K * K + ky * K + kx);
                            acc[c][0] = _mm256_fmadd_ps(wv, lo, acc[c][0]);
                            acc[c][1] = _mm256_fmadd_ps(wv, hi, acc[c][1]);
                        }
                    }
            }
            const int n = Wo - ox0 < TW ? Wo - ox0 : TW;
            for (int c = 0; c < nco; ++c) {
                float tmp[TW];
                _mm256_storeu_ps(tmp, acc[c][0]);
                _mm256_storeu_ps(tmp + 8, acc[c][1]);
                memcpy(y + (((size_t)b * Cout + co0 + c) * Ho + oy) * Wo + ox0, tmp, (size_t)n * sizeof(float));
            }
        }
    }
    free(xp);
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
