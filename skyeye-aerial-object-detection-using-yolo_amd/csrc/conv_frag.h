// MFMA fragment helpers shared by the convolution kernels (k_conv_stream.hip, k_conv_halo.hip).
#pragma once
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>

namespace sky {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

// two floats -> two bf16 in one dword (low half = lo): ONE v_cvt_pk_bf16_f32 (round to nearest even, NaN stays NaN).  The scalar
// form `(__bf16)lo | (__bf16)hi << 16` costs four instructions per pair (two single conversions, a shift, an SDWA or).
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi)
{
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}

// One OCP e4m3fn value (the fp8 of gfx950; tools/fp8_probe.hip checks the conversions and operand layouts used below).
// real value = stored value * (per-tensor scale of the buffer it lives in); weights carry a per-output-channel scale.
struct fp8_t { unsigned char v; };
static constexpr float FP8_MAX = 448.0f;

// two floats -> two e4m3 bytes in the low (hi = false) or high half of `old`.  v_cvt_pk_fp8_f32 rounds to nearest even and turns
// anything above 464 into NaN: saturate first (v_med3_f32).
__device__ __forceinline__ unsigned int fp8_pack2(float a, float b, unsigned int old, bool hi)
{
    a = __builtin_fminf(__builtin_fmaxf(a, -FP8_MAX), FP8_MAX);
    b = __builtin_fminf(__builtin_fmaxf(b, -FP8_MAX), FP8_MAX);
    return hi ? (unsigned int)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)old, true) : (unsigned int)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)old, false);
}
__device__ __forceinline__ unsigned int fp8_pack4(float a, float b, float c, float d)
{
    return fp8_pack2(c, d, fp8_pack2(a, b, 0u, false), true);
}
__device__ __forceinline__ void fp8_unpack4(unsigned int w, float* v)
{
    const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true);
    v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
}

// bf16 engine: the bias of a convolution is the initial value of its accumulators (k_conv_halo.hip: acc_start explains); every kernel
// that can compute a bf16 layer follows this switch, so that all of them round a sum in the same order
template <typename T>
struct BiasInAcc { static constexpr bool value = sizeof(T) == 2; };

template <typename T>
struct S1;
template <>
struct S1<__bf16> {
    static __device__ __forceinline__ void mma(const u32x4_t& wf, const u32x4_t& pf, f32x4_t& acc)
    {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf), __builtin_bit_cast(bf16x8_t, pf), acc, 0, 0, 0);
    }
    // bf16 output keeps 8 mantissa bits: v_exp_f32 / v_rcp_f32 (1 ulp each) are far inside that.  The RAW exp2 instruction, not
    // __expf: the library form wraps v_exp_f32 in a denormal-range guard (compare, two selects, a scale: 4 more VALU instructions per
    // value), and the activation, not the MFMA, is the longer pipe of the small-K layers (45 of ~63 VALU cycles per output value of a
    // 64-channel 3x3).  Below 2^-126 the raw result flushes to 0 -> 1 + 0, above 2^127 it is +inf -> rcp = 0 -> v * 0: both right.
    // EXP2 DOMAIN (round 4): the bf16 engine packs the weights and the bias of every SiLU convolution times log2(e) (engine.cpp:
    // pack_conv), so the accumulator + bias is v' = v log2 e and
    //     SiLU(v) = v / (1 + e^-v) = v' / (log2 e (1 + 2^-v')) = v' * rcp(fma(exp2(-v'), log2 e, log2 e)):
    // exp (8 issue cycles) + fma (4) + rcp (8) + mul (4) instead of mul + exp + add + rcp + mul -- the negation is an input modifier.
    // gate(v') = sigmoid(v) / log2 e: SiLU(v) = v' * gate(v'); kernels that write the activation's last multiplication together
    // with a residual add as one fma (k_csp_stage.hip) take the gate alone
    static __device__ __forceinline__ float gate(float v)
    {
#ifdef SKY_PRE_C
        // experiment (tools/pre_scale_ab.sh): the pre-activation held times an arbitrary c instead of log2 e -- one more multiplication,
        // another rounding realisation of every weight; measures how much the reduced-precision agreement rates move with that alone
        return __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_amdgcn_exp2f(v * (float)(-1.4426950408889634 / (SKY_PRE_C))), (float)(SKY_PRE_C), (float)(SKY_PRE_C)));
#else
        return __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_amdgcn_exp2f(-v), 1.4426950408889634f, 1.4426950408889634f));
#endif
    }
    static __device__ __forceinline__ float silu(float v) { return v * gate(v); }
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

template <>
struct S1<fp8_t> {
    // a 16-byte fragment holds 16 K-elements of this lane's K-group: two 16x16x32 instructions (low / high 8 bytes); any pairing
    // of bytes to k is right as long as the weight and the pixel operand use the same one, and they are read the same way
    static __device__ __forceinline__ void mma(const u32x4_t& wf, const u32x4_t& pf, f32x4_t& acc)
    {
        const long wl = (long)wf[0] | ((long)wf[1] << 32), wh = (long)wf[2] | ((long)wf[3] << 32);
        const long pl = (long)pf[0] | ((long)pf[1] << 32), ph = (long)pf[2] | ((long)pf[3] << 32);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wl, pl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wh, ph, acc, 0, 0, 0);
    }
    static __device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f)); }
};
// fp8 only: the two 16-byte fragments of a 128-byte K chunk (64-byte K-steps 0 and 1) in ONE block-scaled instruction with both
// block scales 2^0 (E8M0 0x7f): 4x the K of the bf16 form at twice its cycles, i.e. the 5 PFLOP/s rate (MI355X_MICROARCH.md).
__device__ __forceinline__ void fp8_mma128(const u32x4_t& w0, const u32x4_t& w1, const u32x4_t& p0, const u32x4_t& p1, f32x4_t& acc)
{
    i32x8_t a, b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        a[e] = (int)w0[e]; a[4 + e] = (int)w1[e];
        b[e] = (int)p0[e]; b[4 + e] = (int)p1[e];
    }
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

// ---- XCD-aware tile order of a persistent tile kernel ------------------------------------------------------------------------
// Workgroup b of a launch runs on XCD b % 8 (MI355X_MICROARCH.md), each with its own L2.  With tile = b, b + G, ... horizontally adjacent
// tiles -- which share their halo columns and, for byte-granular frames, the cache lines their rows start and end in -- sit on eight different
// XCDs, and every L2 fetches the shared lines again.  Here XCD x owns the CONTIGUOUS tile range [x n / 8, (x + 1) n / 8) and its G / 8
// workgroups walk it side by side: neighbours meet in one L2.  Same tiles, same arithmetic; only who computes which tile changes.
__device__ __forceinline__ void tile_walk(int ntile, int& first, int& step, int& end)
{
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    if (G & 7) { first = b; step = G; end = ntile; return; }
    const int x = b & 7;
    const int lo = (int)(((long)x * ntile) >> 3);
    end = (int)(((long)(x + 1) * ntile) >> 3);
    step = G >> 3;
    first = lo + (b >> 3);
}

// ---- 8 consecutive output channels of one pixel in the OUTPUT element type TO (epilogues of every convolution kernel) ----
// raw_t: the bytes as loaded / stored; add(): residual += ; pack(): fp32 -> bytes (fp8: times 1 / out scale, saturated)
template <typename TO> struct Out8;
template <> struct Out8<float> {
    static constexpr int NB = 32;
    struct raw_t { u32x4_t a, b; };
    static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, int voff, int imm)
    {
        return raw_t{__builtin_amdgcn_raw_buffer_load_b128(r, voff, imm, 0), __builtin_amdgcn_raw_buffer_load_b128(r, voff, imm + 16, 0)};
    }
    static __device__ __forceinline__ raw_t load(const void* p) { return raw_t{*reinterpret_cast<const u32x4_t*>(p), *(reinterpret_cast<const u32x4_t*>(p) + 1)}; }
    static __device__ __forceinline__ void add(const raw_t& r, float* v, float)
    {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += __uint_as_float(r.a[e]); v[4 + e] += __uint_as_float(r.b[e]); }
    }
    static __device__ __forceinline__ raw_t pack(const float* v, float)
    {
        raw_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o.a[e] = __float_as_uint(v[e]); o.b[e] = __float_as_uint(v[4 + e]); }
        return o;
    }
    static __device__ __forceinline__ void store(const raw_t& o, __amdgpu_buffer_rsrc_t r, int voff)
    {
        __builtin_amdgcn_raw_buffer_store_b128(o.a, r, voff, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(o.b, r, voff + 16, 0, 0);
    }
    static __device__ __forceinline__ void store(const raw_t& o, void* p) { *reinterpret_cast<u32x4_t*>(p) = o.a; *(reinterpret_cast<u32x4_t*>(p) + 1) = o.b; }
    static __device__ __forceinline__ u32x4_t last(const raw_t& o) { return o.b; }
};
template <> struct Out8<__bf16> {
    static constexpr int NB = 16;
    struct raw_t { u32x4_t a; };
    static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, int voff, int imm) { return raw_t{__builtin_amdgcn_raw_buffer_load_b128(r, voff, imm, 0)}; }
    static __device__ __forceinline__ raw_t load(const void* p) { return raw_t{*reinterpret_cast<const u32x4_t*>(p)}; }
    static __device__ __forceinline__ void add(const raw_t& r, float* v, float)
    {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(r.a[e] << 16); v[2 * e + 1] += __uint_as_float(r.a[e] & 0xffff0000u); }
    }
    static __device__ __forceinline__ raw_t pack(const float* v, float)
    {
        raw_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o.a[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
        return o;
    }
    static __device__ __forceinline__ void store(const raw_t& o, __amdgpu_buffer_rsrc_t r, int voff) { __builtin_amdgcn_raw_buffer_store_b128(o.a, r, voff, 0, 0); }
    static __device__ __forceinline__ void store(const raw_t& o, void* p) { *reinterpret_cast<u32x4_t*>(p) = o.a; }
    static __device__ __forceinline__ u32x4_t last(const raw_t& o) { return o.a; }
};
template <> struct Out8<fp8_t> {
    static constexpr int NB = 8;
    struct raw_t { u32x2_t a; };
    static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, int voff, int imm) { return raw_t{__builtin_amdgcn_raw_buffer_load_b64(r, voff, imm, 0)}; }
    static __device__ __forceinline__ raw_t load(const void* p) { return raw_t{*reinterpret_cast<const u32x2_t*>(p)}; }
    static __device__ __forceinline__ void add(const raw_t& r, float* v, float scale)
    {
        float t[8];
        fp8_unpack4(r.a[0], t);
        fp8_unpack4(r.a[1], t + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += t[e] * scale;
    }
    static __device__ __forceinline__ raw_t pack(const float* v, float inv_scale)
    {
        raw_t o;
        o.a[0] = fp8_pack4(v[0] * inv_scale, v[1] * inv_scale, v[2] * inv_scale, v[3] * inv_scale);
        o.a[1] = fp8_pack4(v[4] * inv_scale, v[5] * inv_scale, v[6] * inv_scale, v[7] * inv_scale);
        return o;
    }
    static __device__ __forceinline__ void store(const raw_t& o, __amdgpu_buffer_rsrc_t r, int voff) { __builtin_amdgcn_raw_buffer_store_b64(o.a, r, voff, 0, 0); }
    static __device__ __forceinline__ void store(const raw_t& o, void* p) { *reinterpret_cast<u32x2_t*>(p) = o.a; }
    static __device__ __forceinline__ u32x4_t last(const raw_t& o) { return u32x4_t{o.a[0], o.a[1], 0u, 0u}; }
};

// ---- fused second 1x1 convolution (ConvArgs::f2_*): C2 -> C2 channels, C2 = 32 or 64 ------------------------------
// The epilogue of a convolution packs, per pixel fragment i and 32-channel group s, 8 consecutive output channels
// of pixel `lane & 15` into 16 bytes (bf16; 2 x 16 bytes in fp32) -- which is exactly the MFMA B operand (pixel = column,
// K-group = lane >> 4) of K-step s of a following 1x1 convolution.  So that convolution runs from registers:
// out2 = act2(W2 * out[:, 0:C2] + b2), W2 staged once per workgroup in LDS with the usual row permutation.
template <typename T>
struct FuseGeom {
    static constexpr int VB = 8 * (int)sizeof(T);      // bytes of one lane's 8-channel vector
    static constexpr int H = VB / 16 ? VB / 16 : 1;    // 16-byte pieces of it (bf16 1, fp32 2; fp8 is never fused)
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
