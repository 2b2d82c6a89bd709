// Streaming 1x1 convolution for gfx950 (MI355X): the HBM-bound half of the SkyEye graph.
//
// 48 of the 75 ConvolutionBlocks of skyeye_s (reference blocks.py:10-41; CSP cv1/cv2/cv3, bottleneck cv1, the neck's
// lateral convs) are 1x1: a GEMM D[cout][pixel] = W[cout][cin] * P[pixel][cin] with K = Cin <= 1024 and a huge M.
// Their arithmetic intensity (<= 128 FLOP/B at Cin = Cout = 128) is below the MFMA/HBM ridge, so the kernel is
// built as a byte streamer, not as a tiled GEMM:
//   * the whole weight tile [N_blk][Cin] (+ bias) is loaded into LDS ONCE per workgroup and stays resident;
//     workgroups are persistent (one per CU) and walk a contiguous range of pixel tiles
//   * every wave owns its pixels: the MFMA B operand (pixels) is loaded straight from global memory into
//     registers in fragment order (16 B per lane), one 256-byte-of-K slab ahead of the MFMAs; pixels are read once
//     and shared with no other wave, so an LDS round trip would be pure overhead
//     (cdna_hip_programming.md 5, "glds vs register staging", GEMV row) -- and there is NO barrier in the loop
//   * weight rows are permuted when read from LDS so that a lane ends up with 8 CONSECUTIVE output channels of its
//     pixel per pair of accumulator fragments: the epilogue (bias, SiLU, residual, bf16 pack) stores 16-byte
//     channel vectors straight from registers -- no LDS staging of the output either
//   * masked loads (pixels past M, K tail) read a zero block instead of branching
// Also used for the nearest-2x-upsampled lateral convs of FeatureNeck (detector.py:210-219).
#include "sky_kernels.h"

#include <hip/hip_bf16.h>

namespace sky {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

template <typename T>
struct S1 {
    static __device__ __forceinline__ void mma(const u32x4_t& wf, const u32x4_t& pf, f32x4_t& acc);
    static __device__ __forceinline__ float silu(float v);
};
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

static constexpr int S1_WAVES = 8;

// MF: 16-pixel fragments per wave tile, NF: 16-channel fragments (N_blk = 16*NF output channels per workgroup)
template <typename T, int MF, int NF>
__global__ void __launch_bounds__(S1_WAVES * 64) conv1x1_stream_kernel(const ConvArgs a)
{
    static_assert(NF % 2 == 0, "pairs of fragments form one 8-channel vector");
    constexpr int NB = NF * 16;
    constexpr int TPX = MF * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int Kb = a.Cin * (int)sizeof(T);      // bytes of K per pixel
    const int ksteps = Kb >> 6;                 // 64-byte K-steps (one MFMA group each)
    const int pitch = Kb + 16;                  // LDS row pitch of the resident weight tile
    const int n0 = blockIdx.y * NB;
    const char* __restrict__ in = reinterpret_cast<const char*>(a.in);
    const char* __restrict__ zero = reinterpret_cast<const char*>(a.zero);
    float* lbias = reinterpret_cast<float*>(smem + NB * pitch);

    // ---- resident weights + bias ----
    {
        const char* wsrc = reinterpret_cast<const char*>(a.w);
        const int cpr = Kb >> 4;                // 16-byte chunks per row
        for (int idx = tid; idx < NB * cpr; idx += S1_WAVES * 64) {
            const int row = idx / cpr, c = idx - row * cpr;
            *reinterpret_cast<u32x4_t*>(smem + row * pitch + c * 16) =
                *reinterpret_cast<const u32x4_t*>(wsrc + (long)(n0 + row) * a.Kpad * (long)sizeof(T) + c * 16);
        }
        for (int i = tid; i < NB; i += S1_WAVES * 64) lbias[i] = a.bias[n0 + i];
    }
    __syncthreads();

    // ---- this wave's tiles: contiguous range per workgroup, waves interleaved inside it ----
    const int ntiles = (a.M + TPX - 1) / TPX;
    const int per = (ntiles + gridDim.x - 1) / gridDim.x;
    const int t_end = min(ntiles, (int)(blockIdx.x + 1) * per);
    int t = blockIdx.x * per + wave;
    if (t >= t_end) return;
    const int nslab = (ksteps + 3) >> 2;
    int sl = 0;

    // LDS row of fragment j, MFMA row r: channel (j>>1)*32 + (r>>2)*8 + (j&1)*4 + (r&3)
    const int wrow0 = (fr >> 2) * 8 + (fr & 3);

    f32x4_t acc[NF][MF];
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int i = 0; i < MF; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    u32x4_t pA[MF][4], pB[MF][4];

    auto load_slab = [&](u32x4_t (&dst)[MF][4], int tt, int ss) {
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int m = tt * TPX + i * 16 + fr;
            const long rowoff = (long)m * a.ldi * (long)sizeof(T);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int ks = ss * 4 + kk;
                const bool ok = (m < a.M) && (ks < ksteps);
                const char* src = ok ? in + rowoff + ks * 64 + fq * 16 : zero;
                dst[i][kk] = *reinterpret_cast<const u32x4_t*>(src);
            }
        }
    };
    auto compute = [&](const u32x4_t (&cur)[MF][4], int ss) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ks = ss * 4 + kk;
            if (ks < ksteps) {
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    const int row = (j >> 1) * 32 + (j & 1) * 4 + wrow0;
                    const u32x4_t wf = *reinterpret_cast<const u32x4_t*>(smem + row * pitch + ks * 64 + fq * 16);
#pragma unroll
                    for (int i = 0; i < MF; ++i) S1<T>::mma(wf, cur[i][kk], acc[j][i]);
                }
            }
        }
    };
    auto epilogue = [&](int tt) {
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int m = tt * TPX + i * 16 + fr;
            if (m < a.M) {
                long p0 = m;
                int rep = 1;
                long step_y = 0;
                if (a.up2) {
                    const int x = m % a.Wo;
                    const int q = m / a.Wo;
                    const int y = q % a.Ho;
                    const int b = q / a.Ho;
                    p0 = ((long)(b * 2 * a.Ho + 2 * y)) * (2 * a.Wo) + 2 * x;
                    rep = 4;
                    step_y = 2 * a.Wo;
                }
#pragma unroll
                for (int s = 0; s < NF / 2; ++s) {
                    const int nl = s * 32 + fq * 8;
                    const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(lbias + nl);
                    const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(lbias + nl + 4);
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[2 * s][i][e] + b0[e];
                        v[4 + e] = acc[2 * s + 1][i][e] + b1[e];
                    }
                    if (a.act == ACT_SILU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = S1<T>::silu(v[e]);
                    } else if (a.act == ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
                    }
                    const int n = n0 + nl;
                    if (a.res) {
                        if (sizeof(T) == 2) {
                            const u32x4_t r = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const unsigned short*>(a.res) + (long)m * a.ldr + n);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v[2 * e] += __uint_as_float(r[e] << 16);
                                v[2 * e + 1] += __uint_as_float(r[e] & 0xffff0000u);
                            }
                        } else {
                            const float* rp = reinterpret_cast<const float*>(a.res) + (long)m * a.ldr + n;
                            const f32x4_t r0 = *reinterpret_cast<const f32x4_t*>(rp), r1 = *reinterpret_cast<const f32x4_t*>(rp + 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
                        }
                    }
                    for (int r = 0; r < rep; ++r) {
                        const long p = p0 + (r & 1) + (r >> 1) * step_y;
                        if (sizeof(T) == 2) {
                            u32x4_t o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const __bf16 lo = (__bf16)v[2 * e], hi = (__bf16)v[2 * e + 1];
                                o[e] = (unsigned int)__builtin_bit_cast(unsigned short, lo) | ((unsigned int)__builtin_bit_cast(unsigned short, hi) << 16);
                            }
                            *reinterpret_cast<u32x4_t*>(reinterpret_cast<unsigned short*>(a.out) + p * a.ldo + n) = o;
                        } else {
                            float* op = reinterpret_cast<float*>(a.out) + p * a.ldo + n;
                            *reinterpret_cast<f32x4_t*>(op) = f32x4_t{v[0], v[1], v[2], v[3]};
                            *reinterpret_cast<f32x4_t*>(op + 4) = f32x4_t{v[4], v[5], v[6], v[7]};
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < NF; ++j) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
    };
    // one pipeline step: prefetch the next slab into `nxt`, consume `cur`; false when the wave is done
    auto step = [&](u32x4_t (&cur)[MF][4], u32x4_t (&nxt)[MF][4]) -> bool {
        int nsl = sl + 1, nt = t;
        if (nsl == nslab) { nsl = 0; nt = t + S1_WAVES; }
        const bool more = nt < t_end;
        if (more) load_slab(nxt, nt, nsl);
        compute(cur, sl);
        if (sl == nslab - 1) epilogue(t);
        t = nt;
        sl = nsl;
        return more;
    };

    load_slab(pA, t, 0);
    for (;;) {
        if (!step(pA, pB)) break;
        if (!step(pB, pA)) break;
    }
}

// ------------------------------------------------------------------------------------------------ host
static int s1_pick_nf(int dtype, const ConvArgs& a)
{
    const int esz = dtype == 0 ? 4 : 2;
    const long Kb = (long)a.Cin * esz;
    if (a.ks != 1 || a.stride != 1 || a.head || a.out_f32) return 0;
    if (Kb % 64 != 0) return 0;
    for (int nf : {8, 4, 2}) {
        const int nb = nf * 16;
        if (a.Cout % nb != 0) continue;
        if ((long)nb * (Kb + 16) + nb * 4 > 144 * 1024) continue;
        return nf;
    }
    return 0;
}

template <typename T, int MF, int NF>
static hipError_t s1_launch(const ConvArgs& a, hipStream_t s, int n_cu)
{
    constexpr int NB = NF * 16, TPX = MF * 16;
    const size_t lds = (size_t)NB * ((size_t)a.Cin * sizeof(T) + 16) + NB * 4;
    static size_t attr_lds = 0;
    auto kern = conv1x1_stream_kernel<T, MF, NF>;
    if (lds > attr_lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_lds = lds;
    }
    const int ntiles = (a.M + TPX - 1) / TPX;
    int gx = (ntiles + S1_WAVES - 1) / S1_WAVES;
    if (gx > n_cu) gx = n_cu;
    hipLaunchKernelGGL(kern, dim3(gx, a.Cout / NB), dim3(S1_WAVES * 64), lds, s, a);
    return hipGetLastError();
}

// returns hipErrorNotSupported when the shape is not covered (caller falls back to the implicit-GEMM kernel)
hipError_t launch_conv1x1_stream(int dtype, const ConvArgs& a, hipStream_t s)
{
    const int nf = s1_pick_nf(dtype, a);
    if (nf == 0) return hipErrorNotSupported;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    if (dtype == 0) {
        switch (nf) {
            case 8: return s1_launch<float, 2, 8>(a, s, n_cu);
            case 4: return s1_launch<float, 4, 4>(a, s, n_cu);
            default: return s1_launch<float, 4, 2>(a, s, n_cu);
        }
    }
    switch (nf) {
        case 8: return s1_launch<__bf16, 2, 8>(a, s, n_cu);
        case 4: return s1_launch<__bf16, 4, 4>(a, s, n_cu);
        default: return s1_launch<__bf16, 4, 2>(a, s, n_cu);
    }
}

}  // namespace sky
