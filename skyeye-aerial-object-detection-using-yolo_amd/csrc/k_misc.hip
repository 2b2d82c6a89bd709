// Layout glue, pooling, upsampling and the CBAM reductions for gfx950.  All of these are HBM-bound
// byte movers: 16 bytes per lane, channel-contiguous NHWC, grid-stride loops capped near 8 blocks per CU.
#include "sky_kernels.h"

#include "conv_frag.h"

#include <math.h>

namespace sky {

static inline int cap_grid(long blocks) { return (int)(blocks < 1 ? 1 : (blocks > 2048 * 4 ? 2048 * 4 : blocks)); }

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<__bf16>(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f32<__bf16>(float v) { return (__bf16)v; }
// fp8 (OCP e4m3fn): the stored byte; the per-tensor scale is applied by the caller (kernel arguments in_scale / out_inv_scale)
template <> __device__ __forceinline__ float to_f32<fp8_t>(fp8_t v) { return __builtin_amdgcn_cvt_f32_fp8((int)v.v, 0); }
template <> __device__ __forceinline__ fp8_t from_f32<fp8_t>(float v) { return fp8_t{(unsigned char)(fp8_pack2(v, 0.0f, 0u, false) & 255u)}; }

// 16-byte vector of T as floats
template <typename T> struct Vec;
template <> struct Vec<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const float* p, float* v) {
        const f32x4_t r = *reinterpret_cast<const f32x4_t*>(p);
        v[0] = r[0]; v[1] = r[1]; v[2] = r[2]; v[3] = r[3];
    }
    static __device__ __forceinline__ void store(float* p, const float* v) {
        *reinterpret_cast<f32x4_t*>(p) = f32x4_t{v[0], v[1], v[2], v[3]};
    }
};
template <> struct Vec<__bf16> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const __bf16* p, float* v) {
        const u32x4_t r = *reinterpret_cast<const u32x4_t*>(p);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = __uint_as_float(r[e] << 16);
            v[2 * e + 1] = __uint_as_float(r[e] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store(__bf16* p, const float* v) {
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
        }
        *reinterpret_cast<u32x4_t*>(p) = o;
    }
};

template <> struct Vec<fp8_t> {
    static constexpr int N = 16;
    static __device__ __forceinline__ void load(const fp8_t* p, float* v) {
        const u32x4_t r = *reinterpret_cast<const u32x4_t*>(p);
#pragma unroll
        for (int e = 0; e < 4; ++e) fp8_unpack4(r[e], v + 4 * e);
    }
    static __device__ __forceinline__ void store(fp8_t* p, const float* v) {
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fp8_pack4(v[4 * e], v[4 * e + 1], v[4 * e + 2], v[4 * e + 3]);
        *reinterpret_cast<u32x4_t*>(p) = o;
    }
};

// ------------------------------------------------------------------------------------------------ import
// dst[b, y, x, c] (NHWC, pitch ld, C padded with zeros up to Cpad).  With s2d the destination pixel (y, x)
// gathers the 2x2 source block: channel p*C + c with p = 0 TL, 1 BL, 2 TR, 3 BR (blocks.py:176-181).
template <typename T, typename S>
__global__ void import_kernel(const S* __restrict__ src, int src_nhwc, T* __restrict__ dst, int B, int C, int H, int W,
                              int Cpad, int ld, int s2d, int scale255, float oscale)
{
    const int Ho = s2d ? H / 2 : H, Wo = s2d ? W / 2 : W;
    const int Cs = s2d ? 4 * C : C;
    const long total = (long)B * Ho * Wo * Cpad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cpad);
        const long p = i / Cpad;
        const int x = (int)(p % Wo);
        const long t = p / Wo;
        const int y = (int)(t % Ho);
        const int b = (int)(t / Ho);
        float v = 0.0f;
        if (c < Cs) {
            int sc = c, sy = y, sx = x;
            if (s2d) {
                const int patch = c / C;
                sc = c - patch * C;
                sy = 2 * y + (patch & 1);
                sx = 2 * x + (patch >> 1);
            }
            const long si = src_nhwc ? (((long)b * H + sy) * W + sx) * C + sc : (((long)b * C + sc) * H + sy) * W + sx;
            v = (float)src[si];
            if (scale255) v = v / 255.0f;
        }
        if (sizeof(T) == 1) v = v * oscale;
        dst[p * ld + c] = from_f32<T>(v);
    }
}

// Fast path of the detector's stem: NCHW 3-channel image (u8 or fp32) -> space-to-depth NHWC pixel of 12 channels
// (+ zero padding to Cpad).  One thread per output pixel: 6 two-element loads (rows 2y, 2y+1 of each plane; consecutive
// threads read consecutive element pairs), whole 16-byte stores.
template <typename T, typename S>
__global__ void import_s2d3_kernel(const S* __restrict__ src, T* __restrict__ dst, int B, int H, int W, int Cpad, int ld, int scale255, float oscale)
{
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)B * Ho * Wo;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        const int x = (int)(p % Wo);
        const long t = p / Wo;
        const int y = (int)(t % Ho);
        const int b = (int)(t / Ho);
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const S* r0 = src + (((long)b * 3 + c) * H + 2 * y) * W + 2 * x;
            const S* r1 = r0 + W;
            // patch order TL, BL, TR, BR (blocks.py:176-181): channel = patch * 3 + c
            float tl = (float)r0[0], tr = (float)r0[1], bl = (float)r1[0], br = (float)r1[1];
            if (scale255) { tl = tl / 255.0f; tr = tr / 255.0f; bl = bl / 255.0f; br = br / 255.0f; }
            v[0 + c] = tl; v[3 + c] = bl; v[6 + c] = tr; v[9 + c] = br;
        }
        if (sizeof(T) == 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = v[e] * oscale;
        }
        T* o = dst + p * ld;
        constexpr int N = Vec<T>::N;
        for (int g = 0; g < Cpad / N; ++g) Vec<T>::store(o + g * N, v + g * N);
    }
}

hipError_t launch_import(int dtype, const void* src, int src_u8, int src_nhwc, void* dst, int B, int C, int H, int W,
                         int Cpad, int ld, int s2d, int scale255, hipStream_t s, float out_inv_scale)
{
    if (s2d && !src_nhwc && C == 3 && Cpad <= 16 && (W % 2 == 0)) {
        const long total = (long)B * (H / 2) * (W / 2);
        const int grid = cap_grid((total + 255) / 256);
#define SKY_IMPORT3(T, S) \
    hipLaunchKernelGGL((import_s2d3_kernel<T, S>), dim3(grid), dim3(256), 0, s, (const S*)src, (T*)dst, B, H, W, Cpad, ld, scale255, out_inv_scale)
        if (dtype == 0) { if (src_u8) SKY_IMPORT3(float, unsigned char); else SKY_IMPORT3(float, float); }
        else if (dtype == 1) { if (src_u8) SKY_IMPORT3(__bf16, unsigned char); else SKY_IMPORT3(__bf16, float); }
        else            { if (src_u8) SKY_IMPORT3(fp8_t, unsigned char); else SKY_IMPORT3(fp8_t, float); }
#undef SKY_IMPORT3
        return hipGetLastError();
    }
    const long total = (long)B * (s2d ? H / 2 : H) * (s2d ? W / 2 : W) * Cpad;
    const int grid = cap_grid((total + 255) / 256);
#define SKY_IMPORT(T, S) \
    hipLaunchKernelGGL((import_kernel<T, S>), dim3(grid), dim3(256), 0, s, (const S*)src, src_nhwc, (T*)dst, B, C, H, W, Cpad, ld, s2d, scale255, out_inv_scale)
    if (dtype == 0) { if (src_u8) SKY_IMPORT(float, unsigned char); else SKY_IMPORT(float, float); }
    else if (dtype == 1) { if (src_u8) SKY_IMPORT(__bf16, unsigned char); else SKY_IMPORT(__bf16, float); }
    else            { if (src_u8) SKY_IMPORT(fp8_t, unsigned char); else SKY_IMPORT(fp8_t, float); }
#undef SKY_IMPORT
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ export
template <typename T>
__global__ void export_kernel(const T* __restrict__ src, int ld, float* __restrict__ dst, int B, int C, int H, int W, float iscale)
{
    // tile transpose through LDS: 32 pixels x 32 channels
    __shared__ float tile[32][33];
    const long HW = (long)H * W;
    const int ptiles = (int)((HW + 31) / 32), ctiles = (C + 31) / 32;
    const long ntile = (long)B * ptiles * ctiles;
    for (long tIdx = blockIdx.x; tIdx < ntile; tIdx += gridDim.x) {
        const int ct = (int)(tIdx % ctiles);
        const long r = tIdx / ctiles;
        const int pt = (int)(r % ptiles);
        const int b = (int)(r / ptiles);
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0..7
        for (int k = ty; k < 32; k += 8) {
            const long p = (long)pt * 32 + k;
            const int c = ct * 32 + tx;
            tile[k][tx] = (p < HW && c < C) ? (sizeof(T) == 1 ? to_f32<T>(src[((long)b * HW + p) * ld + c]) * iscale : to_f32<T>(src[((long)b * HW + p) * ld + c])) : 0.0f;
        }
        __syncthreads();
        for (int k = ty; k < 32; k += 8) {
            const int c = ct * 32 + k;
            const long p = (long)pt * 32 + tx;
            if (p < HW && c < C) dst[((long)b * C + c) * HW + p] = tile[tx][k];
        }
        __syncthreads();
    }
}

hipError_t launch_export(int dtype, const void* src, int ld, float* dst, int B, int C, int H, int W, hipStream_t s, float in_scale)
{
    const long ntile = (long)B * (((long)H * W + 31) / 32) * ((C + 31) / 32);
    const int grid = cap_grid(ntile);
    if (dtype == 0) hipLaunchKernelGGL(export_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)src, ld, dst, B, C, H, W, 1.0f);
    else if (dtype == 1) hipLaunchKernelGGL(export_kernel<__bf16>, dim3(grid), dim3(256), 0, s, (const __bf16*)src, ld, dst, B, C, H, W, 1.0f);
    else hipLaunchKernelGGL(export_kernel<fp8_t>, dim3(grid), dim3(256), 0, s, (const fp8_t*)src, ld, dst, B, C, H, W, in_scale);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ max pool 5x5
// Separable: a thread owns one (image, column, 16-byte channel vector) and walks SEG output rows downwards keeping the
// horizontal 5-max of the last five input rows in registers: 5 loads per row instead of 25 per output.  Out-of-image
// taps are clamped to the edge, which a max cannot see (the edge pixel is already in the window) -- the same result as
// MaxPool2d's -inf padding, and every load is unconditional.
template <typename T>
__global__ void maxpool5_kernel(const T* __restrict__ src, int lds_, T* __restrict__ dst, int ldd, int B, int H, int W, int C)
{
    constexpr int N = Vec<T>::N;
    constexpr int SEG = 8;
    const int cg = C / N;
    const int nseg = (H + SEG - 1) / SEG;
    const long total = (long)B * nseg * W * cg;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        long t = i / cg;
        const int x = (int)(t % W);
        t /= W;
        const int seg = (int)(t % nseg);
        const int b = (int)(t / nseg);
        const T* img = src + (long)b * H * W * lds_ + g * N;
        int xs[5];
#pragma unroll
        for (int d = 0; d < 5; ++d) {
            int xx = x + d - 2;
            xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
            xs[d] = xx * lds_;
        }
        auto rowmax = [&](int yy, float* out) {
            yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
            const T* row = img + (long)yy * W * lds_;
            float v[5][N];
#pragma unroll
            for (int d = 0; d < 5; ++d) Vec<T>::load(row + xs[d], v[d]);
#pragma unroll
            for (int e = 0; e < N; ++e) {
                float m = v[0][e];
#pragma unroll
                for (int d = 1; d < 5; ++d) m = v[d][e] > m ? v[d][e] : m;
                out[e] = m;
            }
        };
        const int y0 = seg * SEG, y1 = y0 + SEG < H ? y0 + SEG : H;
        float r[5][N];
#pragma unroll
        for (int j = 0; j < 4; ++j) rowmax(y0 - 2 + j, r[j]);
        for (int y = y0; y < y1; ++y) {
            rowmax(y + 2, r[4]);
            float m[N];
#pragma unroll
            for (int e = 0; e < N; ++e) {
                float mm = r[0][e];
#pragma unroll
                for (int j = 1; j < 5; ++j) mm = r[j][e] > mm ? r[j][e] : mm;
                m[e] = mm;
            }
            Vec<T>::store(dst + (((long)b * H + y) * W + x) * ldd + g * N, m);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < N; ++e) r[j][e] = r[j + 1][e];
        }
    }
}

// The whole SPP pyramid (5, 9 = 5o5, 13 = 5o5o5; blocks.py:142-147) of a small map in ONE launch: a workgroup owns one image
// and two 16-byte channel vectors (32 contiguous bytes per pixel), keeps that plane in LDS and runs the three cascaded
// separable 5-max passes on it (row pass -> registers -> LDS, column pass -> registers -> global + LDS); x is read once.
// max is exact, so the result equals three maxpool5 launches (up to the sign of a zero that ties with the other zero).  Workgroups that share the cache lines of a
// pixel (same image, neighbouring channel vectors) are placed on the same XCD (blockIdx % 8).
// In LDS the plane holds ORDER KEYS: fp32 values as they are (v_max_f32), bf16 pairs mapped to unsigned 16-bit integers with the
// same order (sign bit flipped for positives, all bits for negatives) so that one v_pk_max_u16 takes the max of two channels.
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
template <typename T> struct PoolKey;
template <> struct PoolKey<float> {
    static __device__ __forceinline__ u32x4_t to_key(const u32x4_t& v) { return v; }
    static __device__ __forceinline__ u32x4_t from_key(const u32x4_t& k) { return k; }
    static __device__ __forceinline__ u32x4_t vmax(const u32x4_t& m, const u32x4_t& v)
    {
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __float_as_uint(fmaxf(__uint_as_float(m[e]), __uint_as_float(v[e])));
        return o;
    }
};
template <> struct PoolKey<__bf16> {
    static __device__ __forceinline__ unsigned flip(unsigned x, unsigned sign_halves)
    {   // per 16-bit half: sign_halves has bit 0 / bit 16 set where ALL bits are flipped, elsewhere only the sign bit
        return x ^ (((sign_halves << 16) - sign_halves) | 0x80008000u);
    }
    static __device__ __forceinline__ u32x4_t to_key(const u32x4_t& v)
    {
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = flip(v[e], (v[e] >> 15) & 0x00010001u);            // negative: flip everything
        return o;
    }
    static __device__ __forceinline__ u32x4_t from_key(const u32x4_t& k)
    {
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = flip(k[e], (~k[e] >> 15) & 0x00010001u);          // key below 0x8000: was negative
        return o;
    }
    static __device__ __forceinline__ u32x4_t vmax(const u32x4_t& m, const u32x4_t& v)
    {
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned a = m[e], b = v[e];     // scalars first: __builtin_bit_cast of a vector ELEMENT reference is miscompiled
            const u16x2_t r = __builtin_elementwise_max(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b));
            o[e] = __builtin_bit_cast(unsigned, r);
        }
        return o;
    }
};

static constexpr int SPP_CG = 2;       // 16-byte channel vectors per workgroup
static constexpr int SPP_ITEMS = 16;   // (pixel, vector) items per thread at most: H * W * SPP_CG <= 256 * 16

template <typename T>
__global__ void __launch_bounds__(256) spp_pyramid_kernel(const T* __restrict__ src, int lds_, T* __restrict__ dst, int ldd, int B, int H, int W,
                                                          int C, int level_stride)
{
    constexpr int N = Vec<T>::N;
    extern __shared__ __attribute__((aligned(16))) char spp_smem[];
    u32x4_t* tile = reinterpret_cast<u32x4_t*>(spp_smem);
    const int ncg = C / (N * SPP_CG);
    const int q = blockIdx.x >> 3;
    const int g = q % ncg, b = (blockIdx.x & 7) + 8 * (q / ncg);
    if (b >= B) return;                                            // uniform per workgroup
    const int npx = H * W, tot = npx * SPP_CG, tid = threadIdx.x;
    const T* in = src + (long)b * npx * lds_ + g * SPP_CG * N;
    T* out = dst + (long)b * npx * ldd + g * SPP_CG * N;
    u32x4_t r[SPP_ITEMS];
    int py[SPP_ITEMS], px[SPP_ITEMS];
#pragma unroll
    for (int k = 0; k < SPP_ITEMS; ++k) {
        const int idx = tid + k * 256;
        const int p = idx / SPP_CG;
        py[k] = p / W;
        px[k] = p - py[k] * W;
        if (idx < tot) tile[idx] = PoolKey<T>::to_key(*reinterpret_cast<const u32x4_t*>(in + (long)p * lds_ + (idx % SPP_CG) * N));
    }
    __syncthreads();
    for (int level = 0; level < 3; ++level) {
#pragma unroll
        for (int k = 0; k < SPP_ITEMS; ++k) {                      // row pass
            const int idx = tid + k * 256;
            if (idx < tot) {
                const int v = idx % SPP_CG, rowb = py[k] * W;
                u32x4_t m = tile[idx];
#pragma unroll
                for (int d = -2; d <= 2; ++d) {
                    if (d == 0) continue;
                    int xx = px[k] + d;
                    xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
                    m = PoolKey<T>::vmax(m, tile[(rowb + xx) * SPP_CG + v]);
                }
                r[k] = m;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SPP_ITEMS; ++k)
            if (tid + k * 256 < tot) tile[tid + k * 256] = r[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SPP_ITEMS; ++k) {                      // column pass
            const int idx = tid + k * 256;
            if (idx < tot) {
                const int v = idx % SPP_CG;
                u32x4_t m = r[k];
#pragma unroll
                for (int d = -2; d <= 2; ++d) {
                    if (d == 0) continue;
                    int yy = py[k] + d;
                    yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
                    m = PoolKey<T>::vmax(m, tile[(yy * W + px[k]) * SPP_CG + v]);
                }
                r[k] = m;
                *reinterpret_cast<u32x4_t*>(out + (long)level * level_stride + (long)(idx / SPP_CG) * ldd + v * N) = PoolKey<T>::from_key(m);
            }
        }
        if (level == 2) break;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SPP_ITEMS; ++k)
            if (tid + k * 256 < tot) tile[tid + k * 256] = r[k];
        __syncthreads();
    }
}

// dst = the first pooled slice; slices 2 and 3 follow at `level_stride` elements.  hipErrorNotSupported when the plane does not
// fit (the caller then issues three launch_maxpool5).
hipError_t launch_spp_pyramid(int dtype, const void* src, int lds_, void* dst, int ldd, int B, int H, int W, int C, int level_stride, hipStream_t s)
{
    if (dtype == 2) return hipErrorNotSupported;       // fp8 maps take the three separable launches (byte order keys not built)
    const int N = dtype == 0 ? 4 : 8;
    if (C % (N * SPP_CG) != 0 || (long)H * W * SPP_CG > 256L * SPP_ITEMS) return hipErrorNotSupported;
    const size_t lds = (size_t)H * W * SPP_CG * 16;
    const int grid = 8 * (C / (N * SPP_CG)) * ((B + 7) / 8);
    if (dtype == 0)
        hipLaunchKernelGGL(spp_pyramid_kernel<float>, dim3(grid), dim3(256), lds, s, (const float*)src, lds_, (float*)dst, ldd, B, H, W, C, level_stride);
    else
        hipLaunchKernelGGL(spp_pyramid_kernel<__bf16>, dim3(grid), dim3(256), lds, s, (const __bf16*)src, lds_, (__bf16*)dst, ldd, B, H, W, C, level_stride);
    return hipGetLastError();
}

hipError_t launch_maxpool5(int dtype, const void* src, int lds_, void* dst, int ldd, int B, int H, int W, int C, hipStream_t s)
{
    const int N = 16 / dtype_size(dtype);
    const long total = (long)B * ((H + 7) / 8) * W * (C / N);
    const int grid = cap_grid((total + 255) / 256);
    if (dtype == 0) hipLaunchKernelGGL(maxpool5_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)src, lds_, (float*)dst, ldd, B, H, W, C);
    else if (dtype == 1) hipLaunchKernelGGL(maxpool5_kernel<__bf16>, dim3(grid), dim3(256), 0, s, (const __bf16*)src, lds_, (__bf16*)dst, ldd, B, H, W, C);
    else hipLaunchKernelGGL(maxpool5_kernel<fp8_t>, dim3(grid), dim3(256), 0, s, (const fp8_t*)src, lds_, (fp8_t*)dst, ldd, B, H, W, C);   // max is monotone in the stored value
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ nearest upsample
template <typename T>
__global__ void upsample_kernel(const T* __restrict__ src, int lds_, T* __restrict__ dst, int ldd, int B, int H, int W, int C,
                                int Ho, int Wo)
{
    constexpr int N = Vec<T>::N;
    const int cg = C / N;
    const float sh = (float)H / (float)Ho, sw = (float)W / (float)Wo;
    const long total = (long)B * Ho * Wo * cg;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        const long p = i / cg;
        const int x = (int)(p % Wo);
        const long t = p / Wo;
        const int y = (int)(t % Ho);
        const int b = (int)(t / Ho);
        int sy = (int)floorf((float)y * sh), sx = (int)floorf((float)x * sw);
        sy = sy > H - 1 ? H - 1 : sy;
        sx = sx > W - 1 ? W - 1 : sx;
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(src + (((long)b * H + sy) * W + sx) * lds_ + g * N);
        *reinterpret_cast<u32x4_t*>(dst + p * ldd + g * N) = v;
    }
}

hipError_t launch_upsample(int dtype, const void* src, int lds_, void* dst, int ldd, int B, int H, int W, int C, int Ho, int Wo,
                           hipStream_t s)
{
    const int N = 16 / dtype_size(dtype);
    const long total = (long)B * Ho * Wo * (C / N);
    const int grid = cap_grid((total + 255) / 256);
    if (dtype == 0) hipLaunchKernelGGL(upsample_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)src, lds_, (float*)dst, ldd, B, H, W, C, Ho, Wo);
    else if (dtype == 1) hipLaunchKernelGGL(upsample_kernel<__bf16>, dim3(grid), dim3(256), 0, s, (const __bf16*)src, lds_, (__bf16*)dst, ldd, B, H, W, C, Ho, Wo);
    else hipLaunchKernelGGL(upsample_kernel<fp8_t>, dim3(grid), dim3(256), 0, s, (const fp8_t*)src, lds_, (fp8_t*)dst, ldd, B, H, W, C, Ho, Wo);   // bytes move; source and destination share a scale
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ decode
struct AnchorPack { float wh[16]; };

template <bool FAST>
__global__ void decode_kernel(const float* __restrict__ raw, float* __restrict__ det, int B, int na, int gh, int gw, int no,
                              long det_rows, long det_off, float stride_px, AnchorPack an)
{
#pragma clang fp contract(off)
    const long total = (long)B * na * gh * gw * no;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int o = (int)(i % no);
        const long cell = i / no;
        const int x = (int)(cell % gw);
        long t = cell / gw;
        const int y = (int)(t % gh);
        t /= gh;
        const int a = (int)(t % na);
        const int b = (int)(t / na);
        const float s = head_sigmoid<FAST>(raw[i]);
        float d;
        if (o == 0) d = (s * 2.0f - 0.5f + (float)x) * stride_px;
        else if (o == 1) d = (s * 2.0f - 0.5f + (float)y) * stride_px;
        else if (o == 2 || o == 3) { const float t2 = s * 2.0f; d = (t2 * t2) * an.wh[a * 2 + (o - 2)]; }
        else d = s;
        det[((long)b * det_rows + det_off + ((long)a * gh + y) * gw + x) * no + o] = d;
    }
}

hipError_t launch_decode(int dtype, const float* raw, float* det, int B, int na, int gh, int gw, int no, long det_rows, long det_off,
                         float stride_px, const float* anchor_wh, hipStream_t s)
{
    AnchorPack an;
    for (int i = 0; i < 16; ++i) an.wh[i] = i < na * 2 ? anchor_wh[i] : 0.0f;
    const long total = (long)B * na * gh * gw * no;
    if (dtype == 0) hipLaunchKernelGGL(decode_kernel<false>, dim3(cap_grid((total + 255) / 256)), dim3(256), 0, s, raw, det, B, na, gh, gw, no, det_rows,
                                       det_off, stride_px, an);
    else hipLaunchKernelGGL(decode_kernel<true>, dim3(cap_grid((total + 255) / 256)), dim3(256), 0, s, raw, det, B, na, gh, gw, no, det_rows, det_off,
                            stride_px, an);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ CBAM
// ChannelAttention pooling (attention.py:50,54): per (b, c) sum and max over H*W, split into `nchunk` pixel
// ranges so the result does not depend on atomics (deterministic): part[b][chunk][0][c] = sum, [1][c] = max.
template <typename T>
__global__ void ca_reduce_kernel(const T* __restrict__ x, int ld, int HW, int C, int nchunk, float* __restrict__ part, float iscale)
{
    // 16-byte channel vectors; the block's 256 threads are (pixel lane, channel group): each pixel lane strides the
    // chunk's pixels, the lanes are combined through LDS in a fixed order (deterministic, no atomics).
    constexpr int N = Vec<T>::N;
    extern __shared__ float red[];                    // [2][npl][cpb * N]
    const int b = blockIdx.y, chunk = blockIdx.x;
    const long per = ((long)HW + nchunk - 1) / nchunk;
    const long p0 = chunk * per, p1 = (p0 + per < HW) ? p0 + per : HW;
    const int CG = C / N;
    const int cpb = CG < (int)blockDim.x ? CG : (int)blockDim.x;     // channel groups per pass
    const int npl = blockDim.x / cpb;                               // pixel lanes
    const int cgl = threadIdx.x % cpb, pl = threadIdx.x / cpb;
    float* o = part + ((long)(b * nchunk + chunk) * 2) * C;
    for (int cg0 = 0; cg0 < CG; cg0 += cpb) {
        const int cg = cg0 + cgl;
        float sv[N], mv[N];
#pragma unroll
        for (int e = 0; e < N; ++e) { sv[e] = 0.0f; mv[e] = -INFINITY; }
        if (pl < npl && cg < CG)
            for (long p = p0 + pl; p < p1; p += npl) {
                float v[N];
                Vec<T>::load(x + ((long)b * HW + p) * ld + cg * N, v);
                if (sizeof(T) == 1) {
#pragma unroll
                    for (int e = 0; e < N; ++e) v[e] = v[e] * iscale;
                }
#pragma unroll
                for (int e = 0; e < N; ++e) { sv[e] += v[e]; mv[e] = v[e] > mv[e] ? v[e] : mv[e]; }
            }
        if (pl < npl) {
#pragma unroll
            for (int e = 0; e < N; ++e) {
                red[(pl * cpb + cgl) * N + e] = sv[e];
                red[(npl * cpb + pl * cpb + cgl) * N + e] = mv[e];
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < cpb * N; c += blockDim.x) {
            if (cg0 * N + c < C) {
                float s2 = 0.0f, m2 = -INFINITY;
                for (int k = 0; k < npl; ++k) {
                    s2 += red[k * cpb * N + c];
                    const float mk = red[(npl + k) * cpb * N + c];
                    m2 = mk > m2 ? mk : m2;
                }
                o[cg0 * N + c] = s2;
                o[C + cg0 * N + c] = m2;
            }
        }
        __syncthreads();
    }
}

hipError_t launch_ca_reduce(int dtype, const void* x, int ld, int B, int HW, int C, int nchunk, float* part, hipStream_t s, float in_scale)
{
    const int n = 16 / dtype_size(dtype);
    const int cg = C / n, cpb = cg < 256 ? cg : 256, npl = 256 / cpb;
    const size_t lds = (size_t)2 * npl * cpb * n * sizeof(float);
    if (dtype == 0) hipLaunchKernelGGL(ca_reduce_kernel<float>, dim3(nchunk, B), dim3(256), lds, s, (const float*)x, ld, HW, C, nchunk, part, 1.0f);
    else if (dtype == 1) hipLaunchKernelGGL(ca_reduce_kernel<__bf16>, dim3(nchunk, B), dim3(256), lds, s, (const __bf16*)x, ld, HW, C, nchunk, part, 1.0f);
    else hipLaunchKernelGGL(ca_reduce_kernel<fp8_t>, dim3(nchunk, B), dim3(256), lds, s, (const fp8_t*)x, ld, HW, C, nchunk, part, in_scale);
    return hipGetLastError();
}

// shared MLP: Linear(C->R, no bias) -> ReLU -> Linear(R->C, no bias); att = sigmoid(mlp(avg) + mlp(max))
// (attention.py:29-34, 49-58).  One block per image; C <= 2048, R <= 128.
__global__ void ca_mlp_kernel(const float* __restrict__ part, int HW, int C, int nchunk, int R, const float* __restrict__ w0,
                              const float* __restrict__ w2, float* __restrict__ att)
{
    extern __shared__ float sm[];
    float* avg = sm;            // [C]
    float* mx = sm + C;         // [C]
    float* ha = sm + 2 * C;     // [R]
    float* hm = ha + R;         // [R]
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.0f, m = -INFINITY;
        for (int k = 0; k < nchunk; ++k) {
            const float* o = part + ((long)(b * nchunk + k) * 2) * C;
            s += o[c];
            m = o[C + c] > m ? o[C + c] : m;
        }
        avg[c] = s / (float)HW;
        mx[c] = m;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        float sa = 0.0f, sx = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float w = w0[(long)r * C + c];
            sa += w * avg[c];
            sx += w * mx[c];
        }
        ha[r] = sa > 0.0f ? sa : 0.0f;
        hm[r] = sx > 0.0f ? sx : 0.0f;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float oa = 0.0f, om = 0.0f;
        for (int r = 0; r < R; ++r) {
            const float w = w2[(long)c * R + r];
            oa += w * ha[r];
            om += w * hm[r];
        }
        const float z = oa + om;
        att[(long)b * C + c] = 1.0f / (1.0f + expf(-z));
    }
}

hipError_t launch_ca_mlp(const float* part, int B, int HW, int C, int nchunk, int R, const float* w0, const float* w2, float* att,
                         hipStream_t s)
{
    hipLaunchKernelGGL(ca_mlp_kernel, dim3(B), dim3(256), (2 * C + 2 * R) * sizeof(float), s, part, HW, C, nchunk, R, w0, w2, att);
    return hipGetLastError();
}

// SpatialAttention statistics (attention.py:91-92): mean and max over channels of x (optionally pre-scaled by the
// channel gate, which is CombinedAttention's x1 = x * att).  One wave per pixel, lanes stride the channels.
template <typename T>
__global__ void sa_stats_kernel(const T* __restrict__ x, int ld, const float* __restrict__ att, int B, int HW, int C, int lp,
                                float* __restrict__ stats, float iscale)
{
    // lp lanes (a power of two <= 64) share one pixel, each owning 16-byte channel vectors lane, lane + lp, ...
    constexpr int N = Vec<T>::N;
    const int lane = threadIdx.x & 63;
    const int sub = lane & (lp - 1);
    const int ppw = 64 / lp;                                        // pixels per wave
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwave = ((long)gridDim.x * blockDim.x) >> 6;
    const long total = (long)B * HW;
    const int CG = C / N;
    for (long p0 = wave * ppw; p0 < total; p0 += nwave * ppw) {
        const long p = p0 + lane / lp;
        float s = 0.0f, m = -INFINITY;
        if (p < total) {
            const int b = (int)((unsigned)p / (unsigned)HW);
            const float* ab = att ? att + (long)b * C : nullptr;
            for (int g = sub; g < CG; g += lp) {
                float v[N];
                Vec<T>::load(x + p * ld + g * N, v);
                if (sizeof(T) == 1) {
#pragma unroll
                    for (int e = 0; e < N; ++e) v[e] = v[e] * iscale;
                }
                if (ab) {
#pragma unroll
                    for (int q = 0; q < N / 4; ++q) {
                        const f32x4_t a4 = *reinterpret_cast<const f32x4_t*>(ab + g * N + q * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[q * 4 + e] = v[q * 4 + e] * a4[e];
                    }
                }
#pragma unroll
                for (int e = 0; e < N; ++e) {
                    s += v[e];
                    m = v[e] > m ? v[e] : m;
                }
            }
        }
        for (int o = lp >> 1; o > 0; o >>= 1) {
            s += __shfl_xor(s, o);
            const float mo = __shfl_xor(m, o);
            m = mo > m ? mo : m;
        }
        if (sub == 0 && p < total) {
            stats[p * 2] = s / (float)C;
            stats[p * 2 + 1] = m;
        }
    }
}

hipError_t launch_sa_stats(int dtype, const void* x, int ld, const float* att, int B, int HW, int C, float* stats, hipStream_t s, float in_scale)
{
    const long total = (long)B * HW;
    const int cg = C / (16 / dtype_size(dtype));
    int lp = 1;
    while (lp < cg && lp < 64) lp <<= 1;
    const int grid = cap_grid((total * lp + 255) / 256);
    if (dtype == 0) hipLaunchKernelGGL(sa_stats_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, ld, att, B, HW, C, lp, stats, 1.0f);
    else if (dtype == 1) hipLaunchKernelGGL(sa_stats_kernel<__bf16>, dim3(grid), dim3(256), 0, s, (const __bf16*)x, ld, att, B, HW, C, lp, stats, 1.0f);
    else hipLaunchKernelGGL(sa_stats_kernel<fp8_t>, dim3(grid), dim3(256), 0, s, (const fp8_t*)x, ld, att, B, HW, C, lp, stats, in_scale);
    return hipGetLastError();
}

// gate[b, y, x] = sigmoid(conv7x7(stats)) with weights w[0][ch][ky][kx], ch 0 = mean, 1 = max (attention.py:79,95-96)
// One workgroup per 16 x 16 tile of an image: the 22 x 22 x 2 window of the statistics goes to LDS once (zeros outside the image: a
// skipped tap and a tap times zero leave the same sum), the 98 taps of a pixel then run from LDS in the order (ch, ky, kx).
__global__ void __launch_bounds__(256) sa_gate_kernel(const float* __restrict__ stats, const float* __restrict__ w, int B, int H, int W,
                                                      float* __restrict__ gate)
{
    __shared__ float ws[98];
    __shared__ float st[2][22][23];
    const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16;
    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, t2 = tile / tiles_x, ty = t2 % tiles_y, b = t2 / tiles_y;
    if (threadIdx.x < 98) ws[threadIdx.x] = w[threadIdx.x];
    for (int i = threadIdx.x; i < 22 * 22; i += 256) {
        const int ly = i / 22, lx = i - ly * 22;
        const int yy = ty * 16 + ly - 3, xx = tx * 16 + lx - 3;
        const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        const float* p = stats + (((long)b * H + yy) * W + xx) * 2;
        st[0][ly][lx] = ok ? p[0] : 0.0f;
        st[1][ly][lx] = ok ? p[1] : 0.0f;
    }
    __syncthreads();
    const int ly = threadIdx.x >> 4, lx = threadIdx.x & 15;
    const int y = ty * 16 + ly, x = tx * 16 + lx;
    if (y >= H || x >= W) return;
    float acc = 0.0f;
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const int yy = y + ky - 3, xx = x + kx - 3;
                // (same sum as skipping the taps outside the image: they add an exact zero)
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) acc += ws[(ch * 7 + ky) * 7 + kx] * st[ch][ly + ky][lx + kx];
            }
    gate[((long)b * H + y) * W + x] = 1.0f / (1.0f + expf(-acc));
}

hipError_t launch_sa_gate(const float* stats, const float* w, int B, int H, int W, float* gate, hipStream_t s)
{
    const long tiles = (long)B * ((H + 15) / 16) * ((W + 15) / 16);
    if (tiles > 2147483647L) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sa_gate_kernel, dim3((unsigned)tiles), dim3(256), 0, s, stats, w, B, H, W, gate);
    return hipGetLastError();
}

// out[b, p, c] = (x[b, p, c] * att[b, c]) * gate[b, p]   (attention.py:60,98)
template <typename T>
__global__ void scale_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ att, const float* __restrict__ gate,
                             T* __restrict__ out, int ldo, int B, int HW, int C, float ratio)
{
    constexpr int N = Vec<T>::N;
    const int cg = C / N;
    const long total = (long)B * HW * cg;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        const long p = i / cg;
        const int b = (int)(p / HW);
        float v[N];
        Vec<T>::load(x + p * ldx + g * N, v);
        if (att) {
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = v[e] * att[(long)b * C + g * N + e];
        }
        if (gate) {
            const float gt = gate[p];
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = v[e] * gt;
        }
        if (sizeof(T) == 1) {          // fp8: stored values carry per-tensor scales: x * (in_scale / out_scale)
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = v[e] * ratio;
        }
        Vec<T>::store(out + p * ldo + g * N, v);
    }
}

hipError_t launch_scale(int dtype, const void* x, int ldx, const float* att, const float* gate, void* out, int ldo, int B, int HW,
                        int C, hipStream_t s, float in_scale, float out_inv_scale)
{
    const int N = 16 / dtype_size(dtype);
    const long total = (long)B * HW * (C / N);
    const int grid = cap_grid((total + 255) / 256);
    if (dtype == 0) hipLaunchKernelGGL(scale_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, ldx, att, gate, (float*)out, ldo, B, HW, C, 1.0f);
    else if (dtype == 1) hipLaunchKernelGGL(scale_kernel<__bf16>, dim3(grid), dim3(256), 0, s, (const __bf16*)x, ldx, att, gate, (__bf16*)out, ldo, B, HW, C, 1.0f);
    else hipLaunchKernelGGL(scale_kernel<fp8_t>, dim3(grid), dim3(256), 0, s, (const fp8_t*)x, ldx, att, gate, (fp8_t*)out, ldo, B, HW, C, in_scale * out_inv_scale);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ calibration of the fp8 engine
// max |x| over an NHWC view, combined into *amax (bits of a non-negative float: unsigned integer order = float order).
template <typename T>
__global__ void amax_kernel(const T* __restrict__ x, int ld, long pixels, int C, unsigned int* __restrict__ amax)
{
    constexpr int N = Vec<T>::N;
    const int cg = C / N;
    const long total = pixels * cg;
    float m = 0.0f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float v[N];
        Vec<T>::load(x + (i / cg) * ld + (i % cg) * N, v);
#pragma unroll
        for (int e = 0; e < N; ++e) {
            const float a = fabsf(v[e]);
            m = a > m ? a : m;                      // NaN never wins
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(amax, __float_as_uint(m));
}

hipError_t launch_amax(int dtype, const void* x, int ld, long pixels, int C, unsigned int* amax, hipStream_t s)
{
    const long total = pixels * (C / (16 / dtype_size(dtype)));
    const int grid = cap_grid((total + 255) / 256);
    if (dtype == 0) hipLaunchKernelGGL(amax_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, ld, pixels, C, amax);
    else if (dtype == 1) hipLaunchKernelGGL(amax_kernel<__bf16>, dim3(grid), dim3(256), 0, s, (const __bf16*)x, ld, pixels, C, amax);
    else return hipErrorNotSupported;
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ letterbox
// letterbox (reference core/data/augmentation.py:442-496): resize the uint8 HWC frame to (nh, nw) with cv2.resize
// INTER_LINEAR, then constant border (top, left, rest) to (H1, W1).  The resize restates OpenCV's fixed-point 8-bit
// bilinear (imgproc resize.cpp, INTER_RESIZE_COEF_BITS = 11): source coordinate (d + 0.5) * scale - 0.5, taps clamped at the
// edges, coefficients rounded to 1/2048, horizontal pass in int, vertical pass
// ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2.  cv2 is not installed in the build container and the
// reference holds no image fixtures: PARITY UNPINNED against OpenCV, pinned against oracle/.  One thread per output pixel
// (3 channels); dst is HWC (the reference's return layout) or CHW with optional channel reversal (detect.py:133:
// transpose((2, 0, 1))[::-1]), so the frame can go straight into the engine.
__device__ __forceinline__ void lb_tap(int d, double scale, int n, int& s0, int& a0, int& a1)
{
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int si = (int)floorf(f);
    f -= (float)si;
    if (si < 0) { si = 0; f = 0.0f; }
    if (si >= n - 1) { si = n - 1; f = 0.0f; }
    s0 = si;
    a0 = (int)rintf((1.0f - f) * 2048.0f);
    a1 = (int)rintf(f * 2048.0f);
}

__global__ void letterbox_kernel(const unsigned char* __restrict__ src, int H0, int W0, unsigned char* __restrict__ dst, int H1, int W1, int nh,
                                 int nw, int top, int left, int pad, int chw, int rev)
{
    const long total = (long)H1 * W1;
    const double sx = (double)W0 / (double)nw, sy = (double)H0 / (double)nh;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W1), y = (int)(i / W1);
        const int rx = x - left, ry = y - top;
        int v[3] = {pad, pad, pad};
        if (rx >= 0 && rx < nw && ry >= 0 && ry < nh) {
            if (nw == W0 && nh == H0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c] = src[((long)ry * W0 + rx) * 3 + c];
            } else {
                int x0, ax0, ax1, y0, by0, by1;
                lb_tap(rx, sx, W0, x0, ax0, ax1);
                lb_tap(ry, sy, H0, y0, by0, by1);
                const int x1 = x0 < W0 - 1 ? x0 + 1 : x0, y1 = y0 < H0 - 1 ? y0 + 1 : y0;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int r0 = src[((long)y0 * W0 + x0) * 3 + c] * ax0 + src[((long)y0 * W0 + x1) * 3 + c] * ax1;
                    const int r1 = src[((long)y1 * W0 + x0) * 3 + c] * ax0 + src[((long)y1 * W0 + x1) * 3 + c] * ax1;
                    int o = (((by0 * (r0 >> 4)) >> 16) + ((by1 * (r1 >> 4)) >> 16) + 2) >> 2;
                    v[c] = o < 0 ? 0 : (o > 255 ? 255 : o);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int cc = rev ? 2 - c : c;
            if (chw) dst[((long)cc * H1 + y) * W1 + x] = (unsigned char)v[c];
            else dst[((long)y * W1 + x) * 3 + cc] = (unsigned char)v[c];
        }
    }
}

hipError_t launch_letterbox(const unsigned char* src, int H0, int W0, unsigned char* dst, int H1, int W1, int nh, int nw, int top, int left,
                            int pad, int chw, int rev, hipStream_t s)
{
    const long total = (long)H1 * W1;
    hipLaunchKernelGGL(letterbox_kernel, dim3(cap_grid((total + 255) / 256)), dim3(256), 0, s, src, H0, W0, dst, H1, W1, nh, nw, top, left, pad, chw, rev);
    return hipGetLastError();
}

}  // namespace sky
