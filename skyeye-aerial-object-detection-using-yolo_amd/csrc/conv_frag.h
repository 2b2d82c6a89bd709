// MFMA fragment helpers shared by the convolution kernels (k_conv_stream.hip, k_conv_halo.hip).
#pragma once
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>

namespace sky {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

template <typename T>
struct S1;
template <>
struct S1<__bf16> {
    static __device__ __forceinline__ void mma(const u32x4_t& wf, const u32x4_t& pf, f32x4_t& acc)
    {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf), __builtin_bit_cast(bf16x8_t, pf), acc, 0, 0, 0);
    }
    // bf16 output keeps 8 mantissa bits: v_exp_f32 / v_rcp_f32 (1 ulp each) are far inside that
    static __device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
};
template <>
struct S1<float> {
    static __device__ __forceinline__ void mma(const u32x4_t& wf, const u32x4_t& pf, f32x4_t& acc)
    {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wf[j]), __uint_as_float(pf[j]), acc, 0, 0, 0);
    }
    static __device__ __forceinline__ float silu(float v) { return v / (1.0f + expf(-v)); }
};

// ---- fused second 1x1 convolution (ConvArgs::f2_*): C2 -> C2 channels, C2 = 32 or 64 ------------------------------
// The epilogue of a convolution packs, per pixel fragment i and 32-channel group s, 8 consecutive output channels
// of pixel `lane & 15` into 16 bytes (bf16; 2 x 16 bytes in fp32) -- which is exactly the MFMA B operand (pixel = column,
// K-group = lane >> 4) of K-step s of a following 1x1 convolution.  So that convolution runs from registers:
// out2 = act2(W2 * out[:, 0:C2] + b2), W2 staged once per workgroup in LDS with the usual row permutation.
template <typename T>
struct FuseGeom {
    static constexpr int VB = 8 * (int)sizeof(T);      // bytes of one lane's 8-channel vector
    static constexpr int H = VB / 16;                  // 16-byte pieces of it (bf16 1, fp32 2)
};

template <typename T, int C2>
__device__ __forceinline__ void fuse_stage(const void* w2, int kpad2, const float* bias2, char* w2lds, float* b2lds, int tid, int nthreads)
{
    constexpr int ROWB = C2 * (int)sizeof(T), CH = ROWB / 16;
    for (int idx = tid; idx < C2 * CH; idx += nthreads) {
        const int row = idx / CH, c = idx - row * CH;
        const int j = row >> 4, r = row & 15;
        const int ch = (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3);
        *reinterpret_cast<u32x4_t*>(w2lds + row * ROWB + c * 16) =
            *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(w2) + (long)ch * kpad2 * (long)sizeof(T) + c * 16);
    }
    for (int i = tid; i < C2; i += nthreads) b2lds[i] = bias2[i];
}

// acc2[j][i] += W2 fragment j x captured operand of pixel fragment i, over the C2/32 K-steps
template <typename T, int C2, int MF>
__device__ __forceinline__ void fuse_gemm(const u32x4_t (&bop)[MF][C2 / 32][FuseGeom<T>::H], const char* w2lds, f32x4_t (&acc2)[C2 / 16][MF],
                                          int fr, int fq)
{
    constexpr int ROWB = C2 * (int)sizeof(T);
#pragma unroll
    for (int s = 0; s < C2 / 32; ++s)
#pragma unroll
        for (int h = 0; h < FuseGeom<T>::H; ++h)
#pragma unroll
            for (int j = 0; j < C2 / 16; ++j) {
                const u32x4_t wf = *reinterpret_cast<const u32x4_t*>(w2lds + (j * 16 + fr) * ROWB + (s * 32 + fq * 8) * (int)sizeof(T) + h * 16);
#pragma unroll
                for (int i = 0; i < MF; ++i) S1<T>::mma(wf, bop[i][s][h], acc2[j][i]);
            }
}

}  // namespace sky
