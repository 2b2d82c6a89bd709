/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement of the test-time-augmentation front end:
 *   scale_img            reference skyeye/utils/torch_utils.py:262-288
 *                        (F.interpolate(size=(int(h*ratio), int(w*ratio)), mode='bilinear', align_corners=False), then
 *                         F.pad(..., value=0.447) to a multiple of the grid size), with the flip of the caller
 *                        (x.flip(3) / x.flip(2) ahead of scale_img, the YOLOv5 _forward_augment convention the
 *                        reference's `augment=` flag, validate.py:245 / detect.py:140, stands for).
 *
 * Bilinear arithmetic = ATen's: scale = in / out (fp32); source = scale * (dst + 0.5) - 0.5 evaluated as ONE fused
 * multiply-add (what both the AVX2/AVX-512 CPU build and the GPU build of ATen compile that expression to -- checked here
 * against F.interpolate: without the fusion the result is off by up to 7e-6 at 1280 px, with it by at most 2 ulp), clamped
 * at 0; lambda = source - floor(source); out = hy * (hx * p00 + lx * p01) + ly * (hx * p10 + lx * p11) with every
 * product and sum rounded separately (this file is compiled with -ffp-contract=off).  Pinned by tests/golden/tta.npz
 * (the reference's own scale_img run in the build container) to 5e-7 absolute on [0, 1] inputs: the CPU kernel of ATen
 * sums the four taps in another order.
 */
#include <math.h>
#include <stddef.h>

static inline void tap(int o, float scale, int n_in, int* i0, int* i1, float* l0, float* l1)
{
    float f = fmaf(scale, (float)o + 0.5f, -0.5f);
    if (f < 0.0f) f = 0.0f;
    int i = (int)f;
    if (i > n_in - 1) i = n_in - 1;
    float l = f - (float)i;
    if (l < 0.0f) l = 0.0f;
    if (l > 1.0f) l = 1.0f;
    *i0 = i;
    *i1 = i + (i < n_in - 1 ? 1 : 0);
    *l1 = l;
    *l0 = 1.0f - l;
}

/* x [B*C, H, W] -> y [B*C, PH, PW]; rows/cols >= (sh, sw) hold `pad`.  flip: 0 none, 2 = rows reversed, 3 = columns
 * reversed (applied to the SOURCE, like x.flip(fi) ahead of scale_img).  sh == H && sw == W copies (ratio == 1.0). */
void sky_oracle_scale_img(const float* x, float* y, int planes, int H, int W, int sh, int sw, int PH, int PW, int flip, float pad)
{
    const float rh = (float)H / (float)sh, rw = (float)W / (float)sw;
    const int same = (sh == H && sw == W);
#pragma omp parallel for schedule(static)
    for (int p = 0; p < planes; ++p) {
        const float* xp = x + (size_t)p * H * W;
        float* yp = y + (size_t)p * PH * PW;
        for (int oy = 0; oy < PH; ++oy)
            for (int ox = 0; ox < PW; ++ox) {
                float v = pad;
                if (oy < sh && ox < sw) {
                    int y0, y1, x0, x1;
                    float hy, ly, hx, lx;
                    if (same) {
                        y0 = y1 = oy; x0 = x1 = ox; hy = hx = 1.0f; ly = lx = 0.0f;
                    } else {
                        tap(oy, rh, H, &y0, &y1, &hy, &ly);
                        tap(ox, rw, W, &x0, &x1, &hx, &lx);
                    }
                    if (flip == 2) { y0 = H - 1 - y0; y1 = H - 1 - y1; }
                    if (flip == 3) { x0 = W - 1 - x0; x1 = W - 1 - x1; }
                    if (same)
                        v = xp[(size_t)y0 * W + x0];
                    else
                        v = hy * (hx * xp[(size_t)y0 * W + x0] + lx * xp[(size_t)y0 * W + x1]) +
                            ly * (hx * xp[(size_t)y1 * W + x0] + lx * xp[(size_t)y1 * W + x1]);
                }
                yp[(size_t)oy * PW + ox] = v;
            }
    }
}
