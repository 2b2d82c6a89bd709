// Test-time augmentation + tiling front end around the batch path:
//   scale_img_kernel       reference skyeye/utils/torch_utils.py:262-288 (scale_img) fused with the caller's flip
//   map_detections_kernel  de-scale / un-flip (YOLOv5 _descale_pred, the body the reference's `augment=` flag stands for,
//                          validate.py:245, detect.py:140) and tile offsets, writing straight into the concatenated tensor
//   tile_gather_kernel     overlapping tiles of a large uint8 frame -> the engine's uint8 [n, 3, th, tw] input
// All three are HBM streaming kernels (a few bytes of arithmetic per element); compiled with -ffp-contract=off so that
// every product / sum rounds once like ATen's fp32 tensor ops, with the ONE fused multiply-add ATen has (source index).
#include "sky_kernels.h"

namespace sky {

static inline int tta_grid(long blocks) { return (int)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks)); }

// ATen area_pixel_compute_source_index (align_corners = False) + guard_index_and_lambda.
__device__ __forceinline__ void bilinear_tap(int o, float scale, int n_in, int& i0, int& i1, float& l0, float& l1)
{
    float f = __builtin_fmaf(scale, (float)o + 0.5f, -0.5f);
    f = f < 0.0f ? 0.0f : f;
    int i = (int)f;
    i = i > n_in - 1 ? n_in - 1 : i;
    float l = f - (float)i;
    l = l < 0.0f ? 0.0f : (l > 1.0f ? 1.0f : l);
    i0 = i;
    i1 = i + (i < n_in - 1 ? 1 : 0);
    l1 = l;
    l0 = 1.0f - l;
}

__device__ __forceinline__ float src_value(const float* p, long i) { return p[i]; }
__device__ __forceinline__ float src_value(const unsigned char* p, long i) { return (float)p[i] / 255.0f; }   // validate.py:236-238

// One thread = 4 consecutive output columns of one row of one plane (16-byte stores when PW % 4 == 0).
template <typename SRC>
__global__ void scale_img_kernel(const SRC* __restrict__ src, int planes, int H, int W, float* __restrict__ dst, int sh, int sw, int PH,
                                 int PW, int flip, float pad, float rh, float rw)
{
    const int qw = (PW + 3) >> 2;
    const long total = (long)planes * PH * qw;
    const bool same = (sh == H && sw == W);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int q = (int)(i % qw);
        const long r = i / qw;
        const int oy = (int)(r % PH);
        const long p = r / PH;
        const SRC* xp = src + p * (long)H * W;
        int y0 = oy, y1 = oy;
        float hy = 1.0f, ly = 0.0f;
        if (!same && oy < sh) bilinear_tap(oy, rh, H, y0, y1, hy, ly);
        if (flip == 2) { y0 = H - 1 - y0; y1 = H - 1 - y1; }
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ox = q * 4 + j;
            v[j] = pad;
            if (oy < sh && ox < sw) {
                int x0 = ox, x1 = ox;
                float hx = 1.0f, lx = 0.0f;
                if (!same) bilinear_tap(ox, rw, W, x0, x1, hx, lx);
                if (flip == 3) { x0 = W - 1 - x0; x1 = W - 1 - x1; }
                if (same)
                    v[j] = src_value(xp, (long)y0 * W + x0);
                else
                    v[j] = hy * (hx * src_value(xp, (long)y0 * W + x0) + lx * src_value(xp, (long)y0 * W + x1)) +
                           ly * (hx * src_value(xp, (long)y1 * W + x0) + lx * src_value(xp, (long)y1 * W + x1));
            }
        }
        float* out = dst + (p * PH + oy) * (long)PW + q * 4;
        if ((PW & 3) == 0) {
            *reinterpret_cast<float4*>(out) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (q * 4 + j < PW) out[j] = v[j];
        }
    }
}

hipError_t launch_scale_img(const void* src, int src_u8, int planes, int H, int W, float* dst, int sh, int sw, int PH, int PW, int flip, float pad,
                            hipStream_t s)
{
    const long total = (long)planes * PH * ((PW + 3) >> 2);
    const float rh = (float)H / (float)sh, rw = (float)W / (float)sw;
    const int grid = tta_grid((total + 255) / 256);
    if (src_u8)
        hipLaunchKernelGGL(scale_img_kernel<unsigned char>, dim3(grid), dim3(256), 0, s, (const unsigned char*)src, planes, H, W, dst, sh, sw, PH, PW,
                           flip, pad, rh, rw);
    else
        hipLaunchKernelGGL(scale_img_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)src, planes, H, W, dst, sh, sw, PH, PW, flip, pad, rh,
                           rw);
    return hipGetLastError();
}

// src [B, N, no] rows [row0, row0 + rows) -> dst[b / tpi][dst_row0 + (b % tpi) * rows + r][:]; columns 0..3 (cx, cy, w, h) are divided by
// `scale`, mirrored (flip 3: cx = img_w - cx, flip 2: cy = img_h - cy) and shifted by the tile origin (origins[b] = (y, x)).
__global__ void map_detections_kernel(const float* __restrict__ src, int B, int N, int no, int row0, int rows, float scale, int flip, float img_h,
                                      float img_w, const int* __restrict__ origins, int tpi, float* __restrict__ dst, long dst_rows, long dst_row0)
{
    const long total = (long)B * rows * no;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % no);
        const long t = i / no;
        const int r = (int)(t % rows);
        const int b = (int)(t / rows);
        float v = src[((long)b * N + row0 + r) * no + c];
        if (c < 4) {
            v = v / scale;
            if (c == 0) {
                if (flip == 3) v = img_w - v;
                if (origins) v = v + (float)origins[2 * b + 1];
            } else if (c == 1) {
                if (flip == 2) v = img_h - v;
                if (origins) v = v + (float)origins[2 * b];
            }
        }
        dst[((long)(b / tpi) * dst_rows + dst_row0 + (long)(b % tpi) * rows + r) * no + c] = v;
    }
}

hipError_t launch_map_detections(const float* src, int B, int N, int no, int row0, int rows, float scale, int flip, float img_h, float img_w,
                                 const int* origins, int tpi, float* dst, long dst_rows, long dst_row0, hipStream_t s)
{
    const long total = (long)B * rows * no;
    hipLaunchKernelGGL(map_detections_kernel, dim3(tta_grid((total + 255) / 256)), dim3(256), 0, s, src, B, N, no, row0, rows, scale, flip, img_h, img_w,
                       origins, tpi, dst, dst_rows, dst_row0);
    return hipGetLastError();
}

// Survivors of a per-tile NMS (corner rows x1, y1, x2, y2, ...) moved into the frame of the whole image, in place: rows
// [0, counts[t]) of tile t get x += origin_x, y += origin_y; the zero rows past the count stay zero.
__global__ void offset_boxes_kernel(float* __restrict__ rows, const int* __restrict__ counts, int T, int R, int cols, const int* __restrict__ origins)
{
    const long total = (long)T * R * 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i & 3);
        const long q = i >> 2;
        const int r = (int)(q % R), t = (int)(q / R);
        if (r >= counts[t]) continue;
        float* p = rows + ((long)t * R + r) * cols + c;
        *p = *p + (float)origins[2 * t + ((c & 1) ? 0 : 1)];
    }
}

hipError_t launch_offset_boxes(float* rows, const int* counts, int T, int R, int cols, const int* origins, hipStream_t s)
{
    const long total = (long)T * R * 4;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(offset_boxes_kernel, dim3(tta_grid((total + 255) / 256)), dim3(256), 0, s, rows, counts, T, R, cols, origins);
    return hipGetLastError();
}

// One thread = 4 consecutive output bytes of one row of one channel of one tile.
__global__ void tile_gather_kernel(const unsigned char* __restrict__ src, int H0, int W0, int src_chw, const int* __restrict__ origins, int n,
                                   unsigned char* __restrict__ dst, int th, int tw, int pad, int rev)
{
    const int qw = (tw + 3) >> 2;
    const long total = (long)n * 3 * th * qw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int q = (int)(i % qw);
        long r = i / qw;
        const int y = (int)(r % th);
        r /= th;
        const int c = (int)(r % 3);
        const int t = (int)(r / 3);
        const int sy = origins[2 * t] + y, sx0 = origins[2 * t + 1] + q * 4;
        const int sc = rev ? 2 - c : c;
        unsigned char v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int sx = sx0 + j;
            int u = pad;
            if (sy >= 0 && sy < H0 && sx >= 0 && sx < W0)
                u = src_chw ? src[((long)sc * H0 + sy) * W0 + sx] : src[((long)sy * W0 + sx) * 3 + sc];
            v[j] = (unsigned char)u;
        }
        unsigned char* out = dst + (((long)t * 3 + c) * th + y) * (long)tw + q * 4;
        if ((tw & 3) == 0) {
            *reinterpret_cast<unsigned int*>(out) = (unsigned)v[0] | ((unsigned)v[1] << 8) | ((unsigned)v[2] << 16) | ((unsigned)v[3] << 24);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (q * 4 + j < tw) out[j] = v[j];
        }
    }
}

hipError_t launch_tile_gather(const unsigned char* src, int H0, int W0, int src_chw, const int* origins, int n, unsigned char* dst, int th, int tw,
                              int pad, int rev, hipStream_t s)
{
    const long total = (long)n * 3 * th * ((tw + 3) >> 2);
    hipLaunchKernelGGL(tile_gather_kernel, dim3(tta_grid((total + 255) / 256)), dim3(256), 0, s, src, H0, W0, src_chw, origins, n, dst, th, tw, pad, rev);
    return hipGetLastError();
}

}  // namespace sky
