// Attention-side kernels for gfx950: LayerNorm, the softmax(QK^T)V core shared by TransformerLayer and
// WindowedSelfAttention, and the column-softmax closed form of CrossLayerAttention.
//
//   TransformerLayer       reference attention.py:244-309 (pre-LN nn.MultiheadAttention + FFN, eval mode)
//   WindowedSelfAttention  reference attention.py:312-399 (q*scale, + relative position bias, + optional mask)
//   CrossLayerAttention    reference attention.py:133-241: the region_size^2 "patches" are identical bilinear
//                          resamples (:208-215) and the softmax runs over dim=3 = image rows (:172,232), so
//                          out[b,c,y,x] = R^2 * softmax_y(sum_{c' in head} Q*K / sqrt(Cq))[b,head(c),y,x] * V[b,c,y,x]
//                          (SURVEY App. B.9).
// All projections (QKV, out_proj, FFN, 1x1 convs with bias) run on the streaming convolution kernel: in NHWC a token
// IS a pixel row.  These kernels are first, correct versions (fp32 VALU math, one query per lane); the MFMA
// flash-style core is listed as next in DESIGN.md.
#include "sky_kernels.h"

#include <hip/hip_bf16.h>
#include <math.h>

namespace sky {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<__bf16>(const __bf16* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void st1(T* p, float v);
template <> __device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st1<__bf16>(__bf16* p, float v) { *p = (__bf16)v; }

static inline int cap_grid(long blocks) { return (int)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks)); }

// ------------------------------------------------------------------------------------------------ LayerNorm
// y = (x - mean) / sqrt(var + eps) * gamma + beta over the last (channel) axis; one wave per token.
template <typename T>
__global__ void layernorm_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, const float* __restrict__ g,
                                 const float* __restrict__ b, long tokens, int C, float eps)
{
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwave = ((long)gridDim.x * blockDim.x) >> 6;
    for (long t = wave; t < tokens; t += nwave) {
        float s = 0.0f;
        for (int c = lane; c < C; c += 64) s += ld1(x + t * ldx + c);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / (float)C;
        float v = 0.0f;
        for (int c = lane; c < C; c += 64) { const float d = ld1(x + t * ldx + c) - mean; v += d * d; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        const float inv = 1.0f / sqrtf(v / (float)C + eps);
        for (int c = lane; c < C; c += 64) st1(y + t * ldy + c, (ld1(x + t * ldx + c) - mean) * inv * g[c] + b[c]);
    }
}

hipError_t launch_layernorm(int dtype, const void* x, int ldx, void* y, int ldy, const float* g, const float* b, long tokens, int C,
                            hipStream_t s)
{
    const int grid = cap_grid((tokens + 3) / 4);
    if (dtype == 0) hipLaunchKernelGGL(layernorm_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, ldx, (float*)y, ldy, g, b, tokens, C, 1e-5f);
    else hipLaunchKernelGGL(layernorm_kernel<__bf16>, dim3(grid), dim3(256), 0, s, (const __bf16*)x, ldx, (__bf16*)y, ldy, g, b, tokens, C, 1e-5f);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ attention core
// qkv: [G, N, 3C] (group g = image or window), head h uses channels [h*D, (h+1)*D) of each of the q | k | v thirds.
// out[g, n, h*D + e] = sum_j softmax_j((q_n * scale) . k_j + bias[h, n, j] + mask[g % nW, n, j]) v_j[e]
// One lane per query, keys streamed through LDS 64 at a time, online softmax in registers.
// ws > 0: the groups are the ws x ws windows of a [B, mh, mw] feature map that stays in NHWC order (no window_partition
// copy, attention.py:358-366 expects the caller to have made one): group g = (image, window row, window column), token n =
// (n / ws, n % ws) inside the window; token_row() maps it to the pixel row of the map.
__device__ __forceinline__ long token_row(int g, int n, int N, int ws, int mh, int mw)
{
    if (ws <= 0) return (long)g * N + n;
    const int nwx = mw / ws, nwy = mh / ws;
    const int wx = g % nwx, t = g / nwx, wy = t % nwy, b = t / nwy;
    return ((long)b * mh + wy * ws + n / ws) * mw + wx * ws + n % ws;
}

template <typename T, int D>
__global__ void __launch_bounds__(64) attention_kernel(const T* __restrict__ qkv, int ldq, T* __restrict__ out, int ldo, int N, int C,
                                                       float scale, const float* __restrict__ bias, const float* __restrict__ mask, int nW,
                                                       int ws, int mh, int mw)
{
    __shared__ float ks[64][D + 1];
    __shared__ float vs[64][D + 1];
    const int lane = threadIdx.x;
    const int h = blockIdx.y, g = blockIdx.z;
    const int n = blockIdx.x * 64 + lane;
    const bool qok = n < N;
    const T* base = qkv;
    const long qrow = token_row(g, qok ? n : 0, N, ws, mh, mw);
    float q[D], o[D];
#pragma unroll
    for (int e = 0; e < D; ++e) {
        q[e] = qok ? ld1(base + qrow * ldq + h * D + e) * scale : 0.0f;
        o[e] = 0.0f;
    }
    float m = -INFINITY, l = 0.0f;
    for (int j0 = 0; j0 < N; j0 += 64) {
        __syncthreads();
        const int j = j0 + lane;
        const long jrow = token_row(g, j < N ? j : 0, N, ws, mh, mw);
        for (int e = 0; e < D; ++e) {
            ks[lane][e] = j < N ? ld1(base + jrow * ldq + C + h * D + e) : 0.0f;
            vs[lane][e] = j < N ? ld1(base + jrow * ldq + 2 * C + h * D + e) : 0.0f;
        }
        __syncthreads();
        const int jn = (N - j0) < 64 ? (N - j0) : 64;
        for (int jj = 0; jj < jn; ++jj) {
            float s = 0.0f;
#pragma unroll
            for (int e = 0; e < D; ++e) s += q[e] * ks[jj][e];
            if (qok) {
                if (bias) s += bias[((long)h * N + n) * N + j0 + jj];
                if (mask) s += mask[((long)(g % nW) * N + n) * N + j0 + jj];
            }
            const float mn = s > m ? s : m;
            const float corr = expf(m - mn), p = expf(s - mn);
            l = l * corr + p;
#pragma unroll
            for (int e = 0; e < D; ++e) o[e] = o[e] * corr + p * vs[jj][e];
            m = mn;
        }
    }
    if (qok) {
        const float inv = 1.0f / l;
#pragma unroll
        for (int e = 0; e < D; ++e) st1(out + qrow * ldo + h * D + e, o[e] * inv);
    }
}

template <typename T>
static hipError_t attention_t(const void* qkv, int ldq, void* out, int ldo, int G, int N, int C, int heads, float scale, const float* bias,
                              const float* mask, int nW, int ws, int mh, int mw, hipStream_t s)
{
    const int D = C / heads;
    const dim3 grid((N + 63) / 64, heads, G);
#define SKY_ATT(DD) hipLaunchKernelGGL((attention_kernel<T, DD>), grid, dim3(64), 0, s, (const T*)qkv, ldq, (T*)out, ldo, N, C, scale, bias, mask, nW, ws, mh, mw)
    switch (D) {
        case 8: SKY_ATT(8); break;
        case 16: SKY_ATT(16); break;
        case 32: SKY_ATT(32); break;
        case 64: SKY_ATT(64); break;
        case 128: SKY_ATT(128); break;
        default: return hipErrorInvalidValue;
    }
#undef SKY_ATT
    return hipGetLastError();
}

hipError_t launch_attention(int dtype, const void* qkv, int ldq, void* out, int ldo, int G, int N, int C, int heads, float scale,
                            const float* bias, const float* mask, int nW, int ws, int mh, int mw, hipStream_t s)
{
    return dtype == 0 ? attention_t<float>(qkv, ldq, out, ldo, G, N, C, heads, scale, bias, mask, nW, ws, mh, mw, s)
                      : attention_t<__bf16>(qkv, ldq, out, ldo, G, N, C, heads, scale, bias, mask, nW, ws, mh, mw, s);
}

// ------------------------------------------------------------------------------------------------ CrossLayerAttention
// F.interpolate(mode='bilinear', align_corners=False) source taps for destination index d (attention.py:211-212)
__device__ __forceinline__ void bil_tap(int d, int in, int out, int& i0, int& i1, float& l1)
{
    float f = ((float)d + 0.5f) * ((float)in / (float)out) - 0.5f;
    f = f < 0.0f ? 0.0f : f;
    i0 = (int)f;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = f - (float)i0;
}

// scores[b, y, x, head] = scale * sum_{c in head} Q[b,y,x,c] * bilinear(K)[b,y,x,c]
template <typename T>
__global__ void cla_scores_kernel(const T* __restrict__ q, int ldq, const T* __restrict__ k, int ldk, float* __restrict__ sc, int B, int H,
                                  int W, int h, int w, int C, int heads, float scale)
{
    const int d = C / heads;
    const long total = (long)B * H * W * heads;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int hd = (int)(i % heads);
        const long p = i / heads;
        const int x = (int)(p % W);
        const long t = p / W;
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        int y0, y1, x0, x1;
        float ly, lx;
        bil_tap(y, h, H, y0, y1, ly);
        bil_tap(x, w, W, x0, x1, lx);
        const float hy = 1.0f - ly, hx = 1.0f - lx;
        const T* k00 = k + (((long)b * h + y0) * w + x0) * ldk + hd * d;
        const T* k01 = k + (((long)b * h + y0) * w + x1) * ldk + hd * d;
        const T* k10 = k + (((long)b * h + y1) * w + x0) * ldk + hd * d;
        const T* k11 = k + (((long)b * h + y1) * w + x1) * ldk + hd * d;
        const T* qq = q + p * ldq + hd * d;
        float s = 0.0f;
        for (int c = 0; c < d; ++c) {
            const float kv = hy * (hx * ld1(k00 + c) + lx * ld1(k01 + c)) + ly * (hx * ld1(k10 + c) + lx * ld1(k11 + c));
            s += ld1(qq + c) * kv;
        }
        sc[i] = s * scale;
    }
}

// softmax over y (image rows) for every (b, x, head), in place
__global__ void cla_colsoftmax_kernel(float* __restrict__ sc, int B, int H, int W, int heads)
{
    const long total = (long)B * W * heads;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int hd = (int)(i % heads);
        const long t = i / heads;
        const int x = (int)(t % W);
        const int b = (int)(t / W);
        float m = -INFINITY;
        for (int y = 0; y < H; ++y) { const float v = sc[(((long)b * H + y) * W + x) * heads + hd]; m = v > m ? v : m; }
        float l = 0.0f;
        for (int y = 0; y < H; ++y) l += expf(sc[(((long)b * H + y) * W + x) * heads + hd] - m);
        for (int y = 0; y < H; ++y) {
            float* p = sc + (((long)b * H + y) * W + x) * heads + hd;
            *p = expf(*p - m) / l;
        }
    }
}

// out[b,y,x,c] = R2 * a[b,y,x,head(c)] * bilinear(V)[b,y,x,c]
template <typename T>
__global__ void cla_apply_kernel(const float* __restrict__ a, const T* __restrict__ v, int ldv, T* __restrict__ out, int ldo, int B, int H, int W,
                                 int h, int w, int C, int heads, float r2)
{
    const int d = C / heads;
    const long total = (long)B * H * W * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long p = i / C;
        const int x = (int)(p % W);
        const long t = p / W;
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        int y0, y1, x0, x1;
        float ly, lx;
        bil_tap(y, h, H, y0, y1, ly);
        bil_tap(x, w, W, x0, x1, lx);
        const float hy = 1.0f - ly, hx = 1.0f - lx;
        const float vv = hy * (hx * ld1(v + (((long)b * h + y0) * w + x0) * ldv + c) + lx * ld1(v + (((long)b * h + y0) * w + x1) * ldv + c)) +
                         ly * (hx * ld1(v + (((long)b * h + y1) * w + x0) * ldv + c) + lx * ld1(v + (((long)b * h + y1) * w + x1) * ldv + c));
        st1(out + p * ldo + c, a[p * heads + c / d] * vv * r2);
    }
}

hipError_t launch_cla(int dtype, const void* q, int ldq, const void* kv, int ldkv, int v_off, float* scores, void* out, int ldo, int B, int H,
                      int W, int h, int w, int C, int heads, float scale, float r2, hipStream_t s)
{
    const long n1 = (long)B * H * W * heads, n2 = (long)B * W * heads, n3 = (long)B * H * W * C;
    if (dtype == 0) {
        const float* k = (const float*)kv;
        hipLaunchKernelGGL(cla_scores_kernel<float>, dim3(cap_grid((n1 + 255) / 256)), dim3(256), 0, s, (const float*)q, ldq, k, ldkv, scores, B, H, W, h, w, C, heads, scale);
        hipLaunchKernelGGL(cla_colsoftmax_kernel, dim3(cap_grid((n2 + 63) / 64)), dim3(64), 0, s, scores, B, H, W, heads);
        hipLaunchKernelGGL(cla_apply_kernel<float>, dim3(cap_grid((n3 + 255) / 256)), dim3(256), 0, s, scores, k + v_off, ldkv, (float*)out, ldo, B, H, W, h, w, C, heads, r2);
    } else {
        const __bf16* k = (const __bf16*)kv;
        hipLaunchKernelGGL(cla_scores_kernel<__bf16>, dim3(cap_grid((n1 + 255) / 256)), dim3(256), 0, s, (const __bf16*)q, ldq, k, ldkv, scores, B, H, W, h, w, C, heads, scale);
        hipLaunchKernelGGL(cla_colsoftmax_kernel, dim3(cap_grid((n2 + 63) / 64)), dim3(64), 0, s, scores, B, H, W, heads);
        hipLaunchKernelGGL(cla_apply_kernel<__bf16>, dim3(cap_grid((n3 + 255) / 256)), dim3(256), 0, s, scores, k + v_off, ldkv, (__bf16*)out, ldo, B, H, W, h, w, C, heads, r2);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ token export
// engine NHWC T (pitch ld) -> caller [tokens, C] fp32
template <typename T>
__global__ void export_tokens_kernel(const T* __restrict__ src, int ld, float* __restrict__ dst, long tokens, int C)
{
    const long total = tokens * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long t = i / C;
        const int c = (int)(i - t * C);
        dst[i] = ld1(src + t * ld + c);
    }
}

hipError_t launch_export_tokens(int dtype, const void* src, int ld, float* dst, long tokens, int C, hipStream_t s)
{
    const int grid = cap_grid((tokens * C + 255) / 256);
    if (dtype == 0) hipLaunchKernelGGL(export_tokens_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)src, ld, dst, tokens, C);
    else hipLaunchKernelGGL(export_tokens_kernel<__bf16>, dim3(grid), dim3(256), 0, s, (const __bf16*)src, ld, dst, tokens, C);
    return hipGetLastError();
}

}  // namespace sky
