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
// IS a pixel row.  attention_kernel is the general core (fp32 VALU math, one query per lane: exact mode, windows, bias,
// mask); attention_mfma_kernel is the flash-style MFMA core of the bf16 engine for plain attention over all tokens.
#include "sky_kernels.h"

#include "conv_frag.h"

#include <hip/hip_bf16.h>
#include <math.h>
#include <stdlib.h>

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

// bf16 tokens of NV * 512 channels: a lane keeps its NV 16-byte vectors (8 consecutive channels each) in registers -- one read of the
// token instead of three, 16-byte accesses instead of 2-byte ones.  Two-pass variance on the registers, as above.
template <int NV>
__global__ void __launch_bounds__(256) layernorm_vec_kernel(const __bf16* __restrict__ x, int ldx, __bf16* __restrict__ y, int ldy,
                                                            const float* __restrict__ g, const float* __restrict__ b, long tokens, float eps)
{
    constexpr int C = NV * 512;
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwave = ((long)gridDim.x * blockDim.x) >> 6;
    float gv[NV][8], bv[NV][8];
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            gv[k][e] = g[k * 512 + lane * 8 + e];
            bv[k][e] = b[k * 512 + lane * 8 + e];
        }
    for (long t = wave; t < tokens; t += nwave) {
        float f[NV][8];
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const u32x4_t r = *reinterpret_cast<const u32x4_t*>(x + t * ldx + k * 512 + lane * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[k][2 * e] = __uint_as_float(r[e] << 16);
                f[k][2 * e + 1] = __uint_as_float(r[e] & 0xffff0000u);
                s += f[k][2 * e] + f[k][2 * e + 1];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / (float)C;
        float v = 0.0f;
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = f[k][e] - mean; v += d * d; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        const float inv = 1.0f / sqrtf(v / (float)C + eps);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            u32x4_t o4;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                o4[e] = pack_bf16x2((f[k][2 * e] - mean) * inv * gv[k][2 * e] + bv[k][2 * e], (f[k][2 * e + 1] - mean) * inv * gv[k][2 * e + 1] + bv[k][2 * e + 1]);
            *reinterpret_cast<u32x4_t*>(y + t * ldy + k * 512 + lane * 8) = o4;
        }
    }
}

hipError_t launch_layernorm(int dtype, const void* x, int ldx, void* y, int ldy, const float* g, const float* b, long tokens, int C,
                            hipStream_t s)
{
    const int grid = cap_grid((tokens + 3) / 4);
    if (dtype == 1 && (C == 512 || C == 1024) && ldx % 8 == 0 && ldy % 8 == 0) {
        if (C == 512) hipLaunchKernelGGL(layernorm_vec_kernel<1>, dim3(grid), dim3(256), 0, s, (const __bf16*)x, ldx, (__bf16*)y, ldy, g, b, tokens, 1e-5f);
        else hipLaunchKernelGGL(layernorm_vec_kernel<2>, dim3(grid), dim3(256), 0, s, (const __bf16*)x, ldx, (__bf16*)y, ldy, g, b, tokens, 1e-5f);
        return hipGetLastError();
    }
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

// ------------------------------------------------------------------------------------------------ MFMA flash core
// softmax(Q K^T * scale) V for the bf16 engine without bias / mask (TransformerLayer, attention.py:282-309), on the matrix
// cores.  A workgroup (4 waves) owns 64 queries of one (image, head); wave w owns 16 of them.  Keys are walked 64 at a time:
//   * S^T = K Q^T with v_mfma_f32_16x16x32_bf16: A = K rows from LDS (XOR-swizzled 16-byte chunks), B = Q straight from
//     global in fragment order (kept in registers for the whole kernel).  In the D layout lane (n, g) then holds, for QUERY
//     n = lane & 15, the keys 4g..4g+3 of each 16-key tile: the softmax reductions over keys are in-lane plus two
//     cross-lane-group steps, and the running maximum / sum / rescale factor of a query live in the lanes that own it;
//   * O^T = V^T P^T: the B operand of K-step ks (32 keys) is lane (n, g)'s OWN eight probabilities of tiles 2ks, 2ks+1
//     packed to bf16 -- no data movement --; the matching A operand (channel row, the lane's own 8 keys) comes out of the row-major
//     V tile through ds_read_b64_tr_b16 (4 keys x 16 channels per 16-lane group): V is staged with 16-byte writes;
//   * O^T's D layout gives lane (n, g) the channels e = 16*tile + 4g..4g+3 of its query: 8-byte stores.
// Online softmax in fp32 (exp on v_exp_f32); P and V enter the second product as bf16.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_att_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_att_t;

// WIN: the groups are ws x ws windows of a [B, mh, mw] map addressed in place (token_row) and the relative position bias
// [heads, N, N] (+ optional mask [nW, N, N]) is added to the scaled scores (WindowedSelfAttention, attention.py:377-392).
// The next block's K / V chunks are requested (into registers) before the current block's arithmetic and written to LDS behind the next barrier
// (P5 TransformerLayer of the head-attention variant, 1600 tokens: 0.389 -> 0.355 ms).
// QF: 16-query fragments per wave (a workgroup owns 64 * QF queries); per query the arithmetic does not depend on it.  QF = 2 (every K / V^T
// fragment read feeds two MFMAs, the tiles are fetched half as often) measured 0.369 ms against 0.355 for QF = 1 -- 144 VGPRs take half the
// waves away -- and is not instantiated (experiments/README.md).
template <int D, bool WIN, int QF = 1>
__global__ void __launch_bounds__(256) attention_mfma_kernel(const __bf16* __restrict__ qkv, int ldq, __bf16* __restrict__ out, int ldo, int N,
                                                             int C, float scale, const float* __restrict__ bias, const float* __restrict__ mask,
                                                             int nW, int ws, int mh, int mw)
{
    constexpr int KT = D / 32;                  // K-steps of the first product
    constexpr int OT = D / 16;                  // 16-channel tiles of the output
    constexpr int KROW = D * 2;                 // bytes of one K row in LDS
    constexpr int CM = (D / 8 < 8 ? D / 8 : 8) - 1;   // chunk-swizzle mask inside one K row (rows of 32 channels have 4 chunks)
    constexpr int NCH = 64 * (D / 8) / 256;     // 16-byte chunks of a 64-key K (or V) tile per thread
    __shared__ __attribute__((aligned(16))) char kl[64 * KROW];       // K tile [64 keys][D], chunks swizzled by (key >> 1) & 7
    __shared__ __attribute__((aligned(16))) char vl[64 * KROW];       // V tile [64 keys][D], row-major; 32-byte channel pairs swizzled by the key
                                                                      // so that the transposed reads below are conflict-free
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    // byte offset of 16-byte chunk c of key row `key` in the V tile: the 32-byte pair index is XORed with a function of the key such
    // that the 8 key rows a 32-lane half reads in one transposed read fall on 8 distinct groups of 8 banks ((key * D/16 + pair) mod 8)
    auto vchunk = [](int c, int key) -> int {
        constexpr int PP = D / 16;                                     // pairs per row: 2, 4, 8
        const int f = PP == 2 ? (key >> 2) & 1 : PP == 4 ? (key >> 1) & 3 : key & 7;
        return ((((c >> 1) ^ f) << 1) | (c & 1)) << 4;
    };
    const int h = blockIdx.y, b = blockIdx.z;                          // b: image, or window when WIN
    int q[QF];
    bool qok[QF];
    long qrow[QF];
    u32x4_t qf[QF][KT];                                                // B operand of S^T: Q[q][ks*32 + g*8 .. +8]
    f32x4_t o[QF][OT];
    float m[QF], l[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        q[f] = blockIdx.x * (64 * QF) + wave * (16 * QF) + f * 16 + n;   // this lane's queries
        qok[f] = q[f] < N;
        qrow[f] = WIN ? token_row(b, qok[f] ? q[f] : 0, N, ws, mh, mw) : (long)b * N + q[f];
#pragma unroll
        for (int ks = 0; ks < KT; ++ks)
            qf[f][ks] = qok[f] ? *reinterpret_cast<const u32x4_t*>(qkv + qrow[f] * ldq + h * D + ks * 32 + g * 8) : u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
        for (int t = 0; t < OT; ++t) o[f][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        m[f] = -INFINITY;
        l[f] = 0.0f;
    }
    const float sl2 = scale * 1.4426950408889634f;                     // exp(x * scale) = exp2(x * scale * log2 e)

    // K / V chunks of one 64-key block: thread -> chunks idx = tid + i * 256, key = idx / (D / 8), c = idx % (D / 8)
    u32x4_t kreg[NCH], vreg[NCH];
    auto fetch = [&](int j0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int idx = tid + i * 256;
            const int key = idx / (D / 8), c = idx - key * (D / 8);
            const int j = j0 + key;
            kreg[i] = u32x4_t{0u, 0u, 0u, 0u};
            vreg[i] = kreg[i];
            if (j < N) {
                const long jrow = WIN ? token_row(b, j, N, ws, mh, mw) : (long)b * N + j;
                kreg[i] = *reinterpret_cast<const u32x4_t*>(qkv + jrow * ldq + C + h * D + c * 8);
                vreg[i] = *reinterpret_cast<const u32x4_t*>(qkv + jrow * ldq + 2 * C + h * D + c * 8);
            }
        }
    };
    fetch(0);
    for (int j0 = 0; j0 < N; j0 += 64) {
        __syncthreads();                                               // previous tiles are consumed
        // ---- stage K (row-major, swizzled) and V^T (permuted key slots) ----
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int idx = tid + i * 256;
            const int key = idx / (D / 8), c = idx - key * (D / 8);    // 16-byte chunk c (8 channels) of key row `key`
            *reinterpret_cast<u32x4_t*>(kl + key * KROW + (((c & ~CM) | ((c ^ (key >> 1)) & CM)) << 4)) = kreg[i];
            *reinterpret_cast<u32x4_t*>(vl + key * KROW + vchunk(c, key)) = vreg[i];
        }
        __syncthreads();
        if (j0 + 64 < N) fetch(j0 + 64);                               // flies under this block's arithmetic
        // ---- S^T tiles: keys 16T + 4g + r of this 64-key block x query n ----
        f32x4_t sT[QF][4];
#pragma unroll
        for (int T = 0; T < 4; ++T) {
#pragma unroll
            for (int f = 0; f < QF; ++f) sT[f][T] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) {
                const int key = T * 16 + n;                            // A operand row of this lane
                const int c = ks * 4 + g;
                const u32x4_t kf = *reinterpret_cast<const u32x4_t*>(kl + key * KROW + (((c & ~CM) | ((c ^ (key >> 1)) & CM)) << 4));
#pragma unroll
                for (int f = 0; f < QF; ++f)
                    sT[f][T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_att_t, kf), __builtin_bit_cast(bf16x8_att_t, qf[f][ks]), sT[f][T], 0, 0, 0);
            }
        }
        // ---- online softmax for query n (log2 domain) ----
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            float bm = -INFINITY;
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = j0 + T * 16 + 4 * g + r;
                    float add = 0.0f;                                  // bias[h][query][key] (+ mask[window % nW][query][key])
                    if (WIN && qok[f] && key < N) {
                        if (bias) add = bias[((long)h * N + q[f]) * N + key];
                        if (mask) add += mask[((long)(b % nW) * N + q[f]) * N + key];
                    }
                    sT[f][T][r] = key < N ? __builtin_fmaf(sT[f][T][r], sl2, add * 1.4426950408889634f) : -INFINITY;
                    bm = sT[f][T][r] > bm ? sT[f][T][r] : bm;
                }
            bm = fmaxf(bm, __shfl_xor(bm, 16));
            bm = fmaxf(bm, __shfl_xor(bm, 32));
            const float mn = bm > m[f] ? bm : m[f];
            const float corr = __builtin_amdgcn_exp2f(m[f] - mn);      // m = -inf on the first block: exp2(-inf) = 0
            float bs = 0.0f;
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sT[f][T][r] = __builtin_amdgcn_exp2f(sT[f][T][r] - mn);
                    bs += sT[f][T][r];
                }
            bs += __shfl_xor(bs, 16);
            bs += __shfl_xor(bs, 32);
            l[f] = l[f] * corr + bs;
            m[f] = mn;
#pragma unroll
            for (int t = 0; t < OT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[f][t][r] *= corr;
        }
        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4_t pf[QF];                                            // this lane's probabilities of tiles 2ks, 2ks+1 as 8 bf16
#pragma unroll
            for (int f = 0; f < QF; ++f)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4_t& src = sT[f][2 * ks + (e >> 1)];
                    pf[f][e] = pack_bf16x2(src[(e & 1) * 2], src[(e & 1) * 2 + 1]);
                }
#pragma unroll
            for (int t = 0; t < OT; ++t) {
                // A operand (channel t*16 + n, the 8 keys of this lane's K-group): two transposed reads of 4 keys x 16 channels.  Lane
                // 4q + p of a 16-lane group supplies key row q, channels 4p .. 4p+3 of the block; lane i receives channel i.
                u32x4_t vf;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int krow = (2 * ks + e) * 16 + 4 * g + (n >> 2);
                    const char* ap = vl + krow * KROW + vchunk(t * 2 + ((n & 3) >> 1), krow) + 8 * (n & 1);
                    const s16x4_att_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_att_t*)ap);
                    const uint2 u = __builtin_bit_cast(uint2, r);
                    vf[2 * e] = u.x;
                    vf[2 * e + 1] = u.y;
                }
#pragma unroll
                for (int f = 0; f < QF; ++f)
                    o[f][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_att_t, vf), __builtin_bit_cast(bf16x8_att_t, pf[f]), o[f][t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int f = 0; f < QF; ++f)
        if (qok[f]) {
            const float inv = 1.0f / l[f];
#pragma unroll
            for (int t = 0; t < OT; ++t) {
                uint2 w;
                w.x = pack_bf16x2(o[f][t][0] * inv, o[f][t][1] * inv);
                w.y = pack_bf16x2(o[f][t][2] * inv, o[f][t][3] * inv);
                *reinterpret_cast<uint2*>(out + qrow[f] * ldo + h * D + t * 16 + 4 * g) = w;
            }
        }
}

// ------------------------------------------------------------------------------------------------ 8 x 8 window core
// WindowedSelfAttention over 64-token windows with 32-channel heads (attention.py:377-392; every windowed layer of the
// detector: heads = C / 32), bf16 engine.  The general flash kernel above spends a workgroup, two barriers and a scattered
// V^T staging on 64 x 64 x 32 scores; here ONE WAVE owns a (window, head) pair and walks windows persistently:
//   * the relative position bias of the wave's head (64 values per lane, x log2 e) is loaded once and stays in registers;
//   * Q and K fragments come straight from global in operand order (16 bytes per lane), no LDS;
//   * V is written row-major into a wave-private 4 KB LDS image with 16-byte stores and read back transposed with
//     ds_read_b64_tr_b16 (4 keys x 16 channels per 16-lane group): the A operand of O^T = V^T P^T in the key order of
//     this lane's own probabilities, as in the general kernel;
//   * no workgroup barrier anywhere; the next window's loads are requested before the current window's arithmetic.
// The four waves of a workgroup take four neighbouring heads of the same windows, so their 64-byte head slices share
// cache lines.  Arithmetic and its order are those of attention_mfma_kernel<32, true> with one key block: results are
// bit-identical (tests/test_gpu_attention_mfma.py).
__global__ void __launch_bounds__(256) window_attention_kernel(const __bf16* __restrict__ qkv, int ldq, __bf16* __restrict__ out, int ldo, int G,
                                                               int C, float scale, const float* __restrict__ bias, int ws, int mh, int mw)
{
    constexpr int D = 32, N = 64;
    __shared__ __attribute__((aligned(16))) char vlds[4][N * 64];      // per wave: V [64 keys][32 channels], 32-byte halves
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;        // swapped on keys with bit 2 set (conflict-free tr reads)
    const int n = lane & 15, g = lane >> 4;
    const int h = blockIdx.y * 4 + wave;
    char* vl = vlds[wave];
    constexpr float LOG2E = 1.4426950408889634f;

    float bl[4][4][4];                                                 // bias[h][query qt*16+n][key T*16+4g+r] * log2 e
#pragma unroll
    for (int qt = 0; qt < 4; ++qt)
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            f32x4_t v = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (bias) v = *reinterpret_cast<const f32x4_t*>(bias + ((long)h * N + qt * 16 + n) * N + T * 16 + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) bl[qt][T][r] = v[r] * LOG2E;
        }
    const float sl2 = scale * LOG2E;

    // V staging: lane -> key (lane >> 2) + 16 i, 16-byte chunk lane & 3
    const int vkey = lane >> 2, vc = lane & 3;
    // transposed reads: lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3 of the 4-key x 16-channel block
    const int tq = n >> 2, tp = n & 3;

    // token (i*16 + n) of a window sits at window_base + toff[i] pixel rows; the base is wave-uniform
    int toff[4], voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = i * 16 + n, tv = i * 16 + vkey;
        toff[i] = ws > 0 ? (t / ws) * mw + t % ws : t;
        voff[i] = ws > 0 ? (tv / ws) * mw + tv % ws : tv;
    }
    const int nwx = ws > 0 ? mw / ws : 1, nwy = ws > 0 ? mh / ws : 1;
    auto window_base = [&](int w_) -> int {
        if (ws <= 0) return w_ * N;
        const int wx = w_ % nwx, t = w_ / nwx, wy = t % nwy, b = t / nwy;
        return (b * mh + wy * ws) * mw + wx * ws;
    };
    u32x4_t qf[4], kf[4], vv[4];
    int row[4];
    auto fetch = [&](int w_, u32x4_t (&q_)[4], u32x4_t (&k_)[4], u32x4_t (&v_)[4], int (&row_)[4]) {
        const int base = window_base(w_);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            row_[i] = base + toff[i];
            const __bf16* p = qkv + (long)row_[i] * ldq + h * D + g * 8;
            q_[i] = *reinterpret_cast<const u32x4_t*>(p);
            k_[i] = *reinterpret_cast<const u32x4_t*>(p + C);
            v_[i] = *reinterpret_cast<const u32x4_t*>(qkv + (long)(base + voff[i]) * ldq + 2 * C + h * D + vc * 8);
        }
    };
    int w = blockIdx.x;
    if (w < G) fetch(w, qf, kf, vv, row);
    for (; w < G; w += gridDim.x) {
        // ---- V image of this window, then its transposed fragments ----
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = i * 16 + vkey;
            *reinterpret_cast<u32x4_t*>(vl + key * 64 + ((vc ^ (((key >> 2) & 1) << 1)) << 4)) = vv[i];
        }
        __builtin_amdgcn_wave_barrier();
        u32x4_t vf[2][2];                                              // [K-step of 32 keys][16-channel tile]
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int krow = (2 * ks + e) * 16 + 4 * g + tq;   // keys (2ks + e)*16 + 4g .. +3: elements 4e .. 4e+3
                    const int c = t * 2 + (tp >> 1);
                    const char* a = vl + krow * 64 + ((c ^ ((g & 1) << 1)) << 4) + 8 * (tp & 1);
                    const s16x4_att_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_att_t*)a);
                    const uint2 u = __builtin_bit_cast(uint2, r);
                    vf[ks][t][2 * e] = u.x;
                    vf[ks][t][2 * e + 1] = u.y;
                }
        __builtin_amdgcn_wave_barrier();
        // ---- next window's operands ----
        u32x4_t nq[4], nk[4], nv[4];
        int nrow[4];
        const int wn = w + gridDim.x;
        if (wn < G) fetch(wn, nq, nk, nv, nrow);
        // ---- four 16-query tiles ----
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            f32x4_t sT[4];
#pragma unroll
            for (int T = 0; T < 4; ++T)
                sT[T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_att_t, kf[T]), __builtin_bit_cast(bf16x8_att_t, qf[qt]),
                                                                f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            float bm = -INFINITY;
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sT[T][r] = __builtin_fmaf(sT[T][r], sl2, bl[qt][T][r]);
                    bm = sT[T][r] > bm ? sT[T][r] : bm;
                }
            bm = fmaxf(bm, __shfl_xor(bm, 16));
            bm = fmaxf(bm, __shfl_xor(bm, 32));
            const float mn = bm > -INFINITY ? bm : -INFINITY;
            float bs = 0.0f;
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sT[T][r] = __builtin_amdgcn_exp2f(sT[T][r] - mn);
                    bs += sT[T][r];
                }
            bs += __shfl_xor(bs, 16);
            bs += __shfl_xor(bs, 32);
            f32x4_t o[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4_t pf;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4_t& src = sT[2 * ks + (e >> 1)];
                    pf[e] = pack_bf16x2(src[(e & 1) * 2], src[(e & 1) * 2 + 1]);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_att_t, vf[ks][t]), __builtin_bit_cast(bf16x8_att_t, pf), o[t], 0, 0, 0);
            }
            const float inv = 1.0f / bs;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                uint2 wv;
                wv.x = pack_bf16x2(o[t][0] * inv, o[t][1] * inv);
                wv.y = pack_bf16x2(o[t][2] * inv, o[t][3] * inv);
                *reinterpret_cast<uint2*>(out + (long)row[qt] * ldo + h * D + t * 16 + 4 * g) = wv;
            }
        }
        if (wn < G) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { qf[i] = nq[i]; kf[i] = nk[i]; vv[i] = nv[i]; row[i] = nrow[i]; }
        }
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
                            const float* bias, const float* mask, int nW, int ws, int mh, int mw, hipStream_t s, unsigned opts)
{
    const int D = heads > 0 ? C / heads : 0;
    const bool no_mfma = (opts & OPT_ATTN_VALU) != 0;                     // A/B switch, fixed at plan time
    if (dtype != 0 && !no_mfma && !(opts & OPT_NO_WINATTN) && (ws > 0 || bias) && !mask && N == 64 && D == 32 && heads % 4 == 0 && ldq % 8 == 0 &&
        ldo % 4 == 0 && C % 8 == 0 && (ws <= 0 || (ws == 8 && mh % 8 == 0 && mw % 8 == 0))) {
        const int gy = heads / 4;
        const int gx = G < 512 / gy ? G : (512 / gy > 0 ? 512 / gy : 1);   // two persistent workgroups per CU over all head groups
        hipLaunchKernelGGL(window_attention_kernel, dim3(gx, gy), dim3(256), 0, s, (const __bf16*)qkv, ldq, (__bf16*)out, ldo, G, C, scale, bias, ws, mh,
                           mw);
        return hipGetLastError();
    }
    if (dtype != 0 && !no_mfma && N >= 64 && (D == 32 || D == 64 || D == 128) && ldq % 8 == 0 && ldo % 4 == 0 && C % 8 == 0) {
        const bool win = ws > 0 || bias || mask;
        const dim3 grid((N + 63) / 64, heads, G);
#define SKY_FLASH(DD, WW) hipLaunchKernelGGL((attention_mfma_kernel<DD, WW, 1>), grid, dim3(256), 0, s, (const __bf16*)qkv, ldq, (__bf16*)out, ldo, N, C, \
                                             scale, bias, mask, nW > 0 ? nW : 1, ws, mh, mw)
        if (D == 32) { if (win) SKY_FLASH(32, true); else SKY_FLASH(32, false); }
        else if (D == 64) { if (win) SKY_FLASH(64, true); else SKY_FLASH(64, false); }
        else { if (win) SKY_FLASH(128, true); else SKY_FLASH(128, false); }
#undef SKY_FLASH
        return hipGetLastError();
    }
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
