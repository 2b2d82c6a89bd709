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

}  // namespace sky
