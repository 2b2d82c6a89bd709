// Streaming convolution for gfx950 (MI355X): every 1x1 convolution and every 3x3 convolution of the SkyEye graph
// except the three detection levels (ConvolutionBlock, reference blocks.py:10-41; the residual add of BottleneckBlock,
// blocks.py:88-90; the nearest-2x upsample + concat of FeatureNeck, detector.py:210-219, as epilogue variants).
//
// GEMM view: D[cout][pixel] = sum_k W[cout][k] * P[pixel][k], k = (tap, cin).  Most of these layers sit below the
// MFMA/HBM ridge (1x1 with Cin = Cout = 128 has 128 FLOP/B), so the kernel is organised as a byte streamer:
//   * PIXELS NEVER TOUCH LDS.  Every wave owns its pixels: the MFMA B operand is fetched from global memory straight
//     into registers in fragment order (16 bytes per lane through a raw buffer descriptor; a masked lane -- pixel past
//     M, filter tap outside the image, K tail -- uses offset 0xffffffff and the hardware range check returns zeros),
//     one 256-byte slab of K ahead of the MFMAs.  A pixel tile is shared with no other wave, so an LDS round trip
//     would be pure overhead (cdna_hip_programming.md 5, "glds vs register staging", GEMV row).
//   * WEIGHTS LIVE IN LDS as [slab][cout][256 B], 16-byte chunks XOR-swizzled by the row so that the permuted fragment
//     reads below are bank-conflict free (checked against the ds_read_b128 lane groups of MI355X_MICROARCH.md).
//     RING = false: the whole [N_blk][K] tile is loaded once per persistent workgroup and stays resident, no barrier
//     in the loop.  RING = true (K*N_blk too large): the eight waves walk K in lockstep and the 256-byte weight slabs
//     stream through a two-stage ring, one barrier per slab.
//   * weight rows are permuted when read (fragment j, MFMA row r -> channel (j>>1)*32 + (r>>2)*8 + (j&1)*4 + (r&3)) so
//     that a lane ends up with 8 CONSECUTIVE output channels of its pixel per pair of accumulator fragments: the
//     epilogue (bias, SiLU, residual, bf16 pack) stores 16-byte channel vectors straight from registers.
//   * when a filter tap spans a multiple of 64 bytes (UTAP) the tap of a K-step is wave-uniform and its address
//     arithmetic stays on the scalar unit.
#include "sky_kernels.h"

#include "conv_frag.h"

#include <stdlib.h>

#include <type_traits>

namespace sky {

static constexpr int SW = 8;        // waves per workgroup
static constexpr int SLAB = 256;    // bytes of K per slab (4 MFMA K-steps of 64 bytes)

__device__ __forceinline__ int wswz(int row) { return (row & 3) | (((row >> 3) & 3) << 2); }

// MF: 16-pixel fragments per wave tile, NF: 16-channel fragments (N_blk = 16*NF output channels per workgroup)
// KT: 64-byte K-steps in the LAST slab of K (1, 2 or 4) -- a template parameter so that both loop bodies are branch-free
// ONE: K fits one slab (1x1 convolutions with <= 256 bytes of input channels): only the KT real K-steps are fetched
// FC > 0: a following FC -> FC 1x1 convolution of this convolution's first FC output channels runs in the epilogue
// (ConvArgs::f2_*, conv_frag.h): the packed output vectors are its MFMA B operands
template <typename T, int KS, int MF, int NF, bool RING, bool UTAP, int KT, bool ONE = false, int FC = 0, typename TO = T>
__global__ void __launch_bounds__(SW * 64) conv_stream_kernel(const ConvArgs a)
{
    static_assert(NF % 2 == 0, "pairs of fragments form one 8-channel vector");
    static_assert(FC == 0 || (std::is_same<T, TO>::value && sizeof(T) >= 2), "the fused 1x1 takes the packed output as its operand");
    constexpr int NB = NF * 16;
    constexpr bool QS = sizeof(T) == 1;            // fp8 operands: per-channel multiplier behind the bias
    constexpr int NBS = QS ? 2 * NB : NB;
    constexpr int TPX = MF * 16;
    constexpr int BUF = NB * SLAB;                 // one slab of weights: [NB rows][256 B]
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;      // MFMA operand role of this lane: pixel column fr, K group fq
    // LOADER role: with SHUF, 4 adjacent lanes fetch the 64 contiguous bytes of one pixel (coalesced quads for the
    // texture addresser) and ds_bpermute moves them into operand order; without it every lane loads its own operand.
    constexpr bool SHUF = !RING && NF >= 4;
    const int lpix = SHUF ? lane >> 2 : fr, lch = SHUF ? lane & 3 : fq;
    const int Cb = a.Cin * (int)sizeof(T);         // bytes of one filter tap per pixel
    const int cpt = Cb >> 4;                       // 16-byte chunks per tap
    const int Kb = KS * KS * Cb;                   // bytes of K per output pixel
    const int nslab = (Kb + SLAB - 1) / SLAB;
    const int n0 = blockIdx.y * NB;
    const char* __restrict__ in = reinterpret_cast<const char*>(a.in);
    const char* __restrict__ wsrc = reinterpret_cast<const char*>(a.w);
    const long wpitch = (long)a.Kpad * (long)sizeof(T);
    float* lbias = reinterpret_cast<float*>(smem + (RING ? 2 : nslab) * BUF);
    char* const w2lds = reinterpret_cast<char*>(lbias + NBS);
    float* const b2lds = reinterpret_cast<float*>(w2lds + FC * FC * (int)sizeof(T));

    // ---- weight staging ----
    constexpr int WCH = (NB * 16 + SW * 64 - 1) / (SW * 64);    // 16-byte chunks per thread per slab
    u32x4_t wreg[WCH];
    auto load_w = [&](int ss) {
#pragma unroll
        for (int k = 0; k < WCH; ++k) {
            const int idx = tid + k * (SW * 64);
            const int row = idx >> 4, c = idx & 15;
            if (NB * 16 % (SW * 64) == 0 || idx < NB * 16)
                wreg[k] = *reinterpret_cast<const u32x4_t*>(wsrc + (long)(n0 + row) * wpitch + (long)ss * SLAB + c * 16);
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int k = 0; k < WCH; ++k) {
            const int idx = tid + k * (SW * 64);
            const int row = idx >> 4, c = idx & 15;
            if (NB * 16 % (SW * 64) == 0 || idx < NB * 16)
                *reinterpret_cast<u32x4_t*>(smem + buf * BUF + row * SLAB + ((c ^ wswz(row)) << 4)) = wreg[k];
        }
    };
    for (int i = tid; i < NB; i += SW * 64) lbias[i] = a.bias[n0 + i];
    if (QS)
        for (int i = tid; i < NB; i += SW * 64) lbias[NB + i] = a.mult ? a.mult[n0 + i] : 1.0f;
    if constexpr (FC > 0) fuse_stage<T, FC ? FC : 32>(a.f2_w, a.f2_Kpad, a.f2_bias, w2lds, b2lds, tid, SW * 64);

    // ---- tile schedule ----
    const int ntiles = (a.M + TPX - 1) / TPX;
    const int per = (ntiles + gridDim.x - 1) / gridDim.x;
    const int t_begin = blockIdx.x * per;
    const int t_end = min(ntiles, t_begin + per);
    if (t_begin >= t_end) return;                                   // uniform per workgroup
    int t = t_begin + wave;                                         // then t += SW
    int sl = 0;

    if (!RING) {                                                    // resident weights: all slabs once
        for (int ss = 0; ss < nslab; ++ss) {
            load_w(ss);
            store_w(ss);
        }
        __syncthreads();
        if (t >= t_end) return;                                     // no barrier after this point
    }

    // LDS row of fragment j, MFMA row r: channel (j>>1)*32 + (r>>2)*8 + (j&1)*4 + (r&3)
    const int wrow0 = (fr >> 2) * 8 + (fr & 3);
    const int wsw0 = wswz(wrow0);            // rows of different j differ only in bits 2 and 5+, which wswz ignores

    f32x4_t acc[NF][MF];
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int i = 0; i < MF; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // bf16: the accumulators of a tile start from the bias instead of zero (k_conv_halo.hip: acc_start explains)
    auto acc_bias = [&](int i) {
#pragma unroll
        for (int j = 0; j < NF; ++j)
            acc[j][i] = BiasInAcc<T>::value ? *reinterpret_cast<const f32x4_t*>(lbias + (j >> 1) * 32 + fq * 8 + (j & 1) * 4) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    };

    u32x4_t pA[MF][4], pB[MF][4];

    // load-side pixel state of the tile being fetched (decoded once per tile): 32-bit byte offset of the pixel's
    // top-left tap and a 9-bit "tap is inside the image" mask
    int lvo[MF], okm[MF];
    // second input (ConvArgs::in2, 1x1 only): byte offset of this lane's pixel in it, and the slabs of K it provides
    const bool dual = KS == 1 && a.in2 != nullptr;
    const int nsplit = dual ? (a.in2_cin * (int)sizeof(T)) / SLAB : 0;
    int lvo2[MF];
    auto decode_tile = [&](int tt) {
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int m = tt * TPX + i * 16 + lpix;
            if (KS == 1) lvo2[i] = 0;
            if (m >= a.M || tt >= t_end) { lvo[i] = 0; okm[i] = 0; continue; }
            if (KS == 1) {
                lvo[i] = m * a.ldi * (int)sizeof(T);
                okm[i] = 1;
                if (dual) {
                    int m2 = m;
                    if (a.in2_up2) {
                        const int x = m % a.Wo;
                        const int q = m / a.Wo;
                        const int y = q % a.Ho;
                        const int b = q / a.Ho;
                        m2 = (b * (a.Ho >> 1) + (y >> 1)) * (a.Wo >> 1) + (x >> 1);
                    }
                    lvo2[i] = m2 * a.ldi2 * (int)sizeof(T);
                }
            } else {
                const int ox = m % a.Wo;
                const int q = m / a.Wo;
                const int oy = q % a.Ho;
                const int b = q / a.Ho;
                const int iy = oy * a.stride - a.pad, ix = ox * a.stride - a.pad;
                lvo[i] = ((b * a.H + iy) * a.W + ix) * a.ldi * (int)sizeof(T);
                int mask = 0;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
                        if ((unsigned)(iy + ky) < (unsigned)a.H && (unsigned)(ix + kx) < (unsigned)a.W) mask |= 1 << (ky * 3 + kx);
                okm[i] = mask;
            }
        }
    };
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in), 0, (int)a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc2 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(dual ? a.in2 : a.in), 0, (int)(dual ? a.in2_bytes : a.in_bytes), 0x00020000);
    auto load_slab = [&](u32x4_t (&dst)[MF][4], int ss) {
        if (KS == 1 && dual) {
            // whole slabs belong to one input: the first nsplit to in2 (read at the up2-mapped pixel), the rest to `in`
            const bool from2 = ss < nsplit;
            const int sb = from2 ? ss : ss - nsplit;
            const int cpt1 = from2 ? (a.in2_cin * (int)sizeof(T)) >> 4 : cpt - ((a.in2_cin * (int)sizeof(T)) >> 4);
#pragma unroll
            for (int kk = 0; kk < (ONE ? KT : 4); ++kk) {
                const int c0 = (sb * 4 + kk) * 4;
                const int c = UTAP ? c0 : c0 + lch;
                const bool kok = c < cpt1;
                const int koff = c * 16 + (UTAP ? lch * 16 : 0);
#pragma unroll
                for (int i = 0; i < MF; ++i) {
                    const bool ok = kok && okm[i];
                    dst[i][kk] = from2 ? __builtin_amdgcn_raw_buffer_load_b128(rsrc2, ok ? lvo2[i] + koff : -1, 0, 0)
                                       : __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? lvo[i] + koff : -1, 0, 0);
                }
            }
            return;
        }
#pragma unroll
        for (int kk = 0; kk < (ONE ? KT : 4); ++kk) {
            const int c0 = (ss * 4 + kk) * 4;                 // first 16-byte chunk of this K-step (wave-uniform)
            const int c = UTAP ? c0 : c0 + lch;
            int tap = 0, w = c;
            if (KS != 1) {
                tap = a.cpt_shift >= 0 ? (c >> a.cpt_shift) : (c / cpt);
                w = c - tap * cpt;
            }
            const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
            const bool kok = KS == 1 ? (c < cpt) : (tap < KS * KS);
            const int koff = (ky * a.W + kx) * a.ldi * (int)sizeof(T) + w * 16 + (UTAP ? lch * 16 : 0);
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const bool ok = kok && ((okm[i] >> tap) & 1);
                dst[i][kk] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? lvo[i] + koff : -1, 0, 0);
            }
        }
    };
    // Loads are issued in LOADER order (4 adjacent lanes = 64 contiguous bytes of one pixel, so the texture addresser
    // sees 16 coalesced quads per wave load instead of 64 single-lane accesses); the MFMA B operand wants lane
    // fq*16+fr to hold pixel fr / chunk fq.  One ds_bpermute per dword moves the data (LDS crossbar, no LDS memory).
    const int perm_addr = (fr * 4 + fq) * 4;
    auto unshuffle = [&](u32x4_t (&buf)[MF][4]) {
        if (!SHUF) return;
#pragma unroll
        for (int kk = 0; kk < (ONE ? KT : 4); ++kk)
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) buf[i][kk][e] = (unsigned)__builtin_amdgcn_ds_bpermute(perm_addr, (int)buf[i][kk][e]);
    };
    // NKK MFMA K-steps of one slab; `wb` = LDS offset of the slab's weights.
    // Software pipeline over G = NKK * NF/2 groups (K-step kk, fragment pair sp): the two weight fragments of group
    // g+2 are read from LDS before the 2*MF MFMAs of group g issue, so LDS latency hides behind two groups of matrix
    // work while only three pairs (24 VGPRs) are live.  sched_barrier pins that order.
    auto compute_n = [&](const u32x4_t (&cur)[MF][4], int wb, auto nkk_tag) {
        constexpr int NKK = decltype(nkk_tag)::value;
        const char* base = smem + wb + wrow0 * SLAB;
        if constexpr (sizeof(T) == 1 && NKK % 2 == 0) {
            // fp8: two 64-byte K-steps per 16x16x128 instruction (fp8_mma128): half the MFMA count at the same K.  Pipeline over
            // (K-step pair kp, weight fragment j): the two pieces of fragment j + 1 are read before the MF MFMAs of fragment j.
            constexpr int G8 = (NKK / 2) * NF;
            u32x4_t w2[2][2];
#pragma unroll
            for (int g = 0; g < G8 + 1; ++g) {
                if (g < G8) {
                    const int kp = g / NF, j = g % NF;
                    const char* wp = base + ((j >> 1) * 32 + (j & 1) * 4) * SLAB;
                    w2[g & 1][0] = *reinterpret_cast<const u32x4_t*>(wp + ((((2 * kp) * 4 + fq) ^ wsw0) << 4));
                    w2[g & 1][1] = *reinterpret_cast<const u32x4_t*>(wp + ((((2 * kp + 1) * 4 + fq) ^ wsw0) << 4));
                }
                if (g >= 1) {
                    const int q = g - 1, kp = q / NF, j = q % NF;
#pragma unroll
                    for (int i = 0; i < MF; ++i) fp8_mma128(w2[q & 1][0], w2[q & 1][1], cur[i][2 * kp], cur[i][2 * kp + 1], acc[j][i]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
        constexpr int G = NKK * (NF / 2);
        u32x4_t wq[3][2];
#pragma unroll
        for (int g = 0; g < G + 2; ++g) {
            if (g < G) {
                const int kk = g / (NF / 2), sp = g % (NF / 2);
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    wq[g % 3][h] = *reinterpret_cast<const u32x4_t*>(base + (sp * 32 + h * 4) * SLAB + (((kk * 4 + fq) ^ wsw0) << 4));
            }
            if (g >= 2) {
                const int gg = g - 2, kk = gg / (NF / 2), sp = gg % (NF / 2);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < MF; ++i) S1<T>::mma(wq[gg % 3][h], cur[i][kk], acc[2 * sp + h][i]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto compute = [&](const u32x4_t (&cur)[MF][4], int wb, bool last) {
        if (KT == 4 || (!ONE && !last)) compute_n(cur, wb, std::integral_constant<int, 4>());
        else compute_n(cur, wb, std::integral_constant<int, KT>());
    };
    auto epilogue = [&](int tt) {
        constexpr int C2 = FC ? FC : 32;
        u32x4_t bop[MF][C2 / 32][FuseGeom<T>::H];
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int m = tt * TPX + i * 16 + fr;
            if (FC > 0) {
#pragma unroll
                for (int s2 = 0; s2 < C2 / 32; ++s2)
#pragma unroll
                    for (int h = 0; h < FuseGeom<T>::H; ++h) bop[i][s2][h] = u32x4_t{0u, 0u, 0u, 0u};
            }
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
                    if (QS) {
                        const f32x4_t m0 = *reinterpret_cast<const f32x4_t*>(lbias + NB + nl);
                        const f32x4_t m1 = *reinterpret_cast<const f32x4_t*>(lbias + NB + nl + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = acc[2 * s][i][e] * m0[e] + b0[e];
                            v[4 + e] = acc[2 * s + 1][i][e] * m1[e] + b1[e];
                        }
                    } else if (BiasInAcc<T>::value) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {          // the bias is the accumulators' initial value
                            v[e] = acc[2 * s][i][e];
                            v[4 + e] = acc[2 * s + 1][i][e];
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = acc[2 * s][i][e] + b0[e];
                            v[4 + e] = acc[2 * s + 1][i][e] + b1[e];
                        }
                    }
                    if (a.act == ACT_SILU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = S1<T>::silu(v[e]);
                    } else if (a.act == ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
                    }
                    if (a.res) {
                        const char* rp = reinterpret_cast<const char*>(a.res) + ((long)m * a.ldr + n0 + nl) * (long)sizeof(TO);
                        Out8<TO>::add(Out8<TO>::load(rp), v, a.res_scale);
                    }
                    const int n = n0 + nl;
                    const typename Out8<TO>::raw_t o = Out8<TO>::pack(v, a.out_inv_scale);
                    for (int r = 0; r < rep; ++r) {
                        const long p = p0 + (r & 1) + (r >> 1) * step_y;
                        Out8<TO>::store(o, reinterpret_cast<char*>(a.out) + (p * a.ldo + n) * (long)sizeof(TO));
                    }
                    if constexpr (FC > 0) {
                        if (s < FC / 32) {
                            if constexpr (sizeof(T) == 2) bop[i][s][0] = o.a;
                            else { bop[i][s][0] = o.a; bop[i][s][FuseGeom<T>::H - 1] = o.b; }
                        }
                    }
                }
            }
            acc_bias(i);                                       // zeros, or the bias again (bf16)
        }
        if constexpr (FC > 0) {       // the fused 1x1: out2 = act2(W2 * packed[:, 0:FC] + b2)
            f32x4_t acc2[C2 / 16][MF];
#pragma unroll
            for (int j = 0; j < C2 / 16; ++j)
#pragma unroll
                for (int i = 0; i < MF; ++i)
                    acc2[j][i] = BiasInAcc<T>::value ? *reinterpret_cast<const f32x4_t*>(b2lds + (j >> 1) * 32 + fq * 8 + (j & 1) * 4) : f32x4_t{0.f, 0.f, 0.f, 0.f};
            fuse_gemm<T, C2, MF>(bop, w2lds, acc2, fr, fq);
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const int m = tt * TPX + i * 16 + fr;
                if (m < a.M) {
#pragma unroll
                    for (int s = 0; s < C2 / 32; ++s) {
                        const int nl = s * 32 + fq * 8;
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = BiasInAcc<T>::value ? acc2[2 * s][i][e] : acc2[2 * s][i][e] + b2lds[nl + e];
                            v[4 + e] = BiasInAcc<T>::value ? acc2[2 * s + 1][i][e] : acc2[2 * s + 1][i][e] + b2lds[nl + 4 + e];
                        }
                        if (a.f2_act == ACT_SILU) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = S1<T>::silu(v[e]);
                        } else if (a.f2_act == ACT_RELU) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
                        }
                        if (sizeof(T) == 2) {
                            u32x4_t o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                o[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
                            }
                            *reinterpret_cast<u32x4_t*>(reinterpret_cast<unsigned short*>(a.f2_out) + (long)m * a.f2_ldo + nl) = o;
                        } else {
                            float* op = reinterpret_cast<float*>(a.f2_out) + (long)m * a.f2_ldo + nl;
                            *reinterpret_cast<f32x4_t*>(op) = f32x4_t{v[0], v[1], v[2], v[3]};
                            *reinterpret_cast<f32x4_t*>(op + 4) = f32x4_t{v[4], v[5], v[6], v[7]};
                        }
                    }
                }
            }
        }
    };

    // ---- pipeline ----
    int it = 0;
    int total = 0;
    if (RING) {
        const int rounds = (t_end - t_begin + SW - 1) / SW;
        total = rounds * nslab;
        load_w(0);
    }
    decode_tile(t);
    load_slab(pA, 0);
    if (RING) {
        store_w(0);
        __syncthreads();
    }
    if constexpr (BiasInAcc<T>::value) {                            // lbias is visible behind the barriers above
#pragma unroll
        for (int i = 0; i < MF; ++i) acc_bias(i);
    }
    // one step: prefetch the next pixel slab (own registers) and, with RING, the next weight slab (other LDS stage);
    // consume `cur`.  Returns false when this wave (RING: the workgroup) is done.
    auto step = [&](u32x4_t (&cur)[MF][4], u32x4_t (&nxt)[MF][4]) -> bool {
        int nsl = sl + 1, nt = t;
        if (nsl == nslab) { nsl = 0; nt = t + SW; }
        const bool more = RING ? (it + 1 < total) : (nt < t_end);
        // The prefetch is issued UNCONDITIONALLY (past the end it is fully masked / re-reads a valid weight slab).
        // Under `if (more)` hipcc must assume the path without the new loads and then waits for the CURRENT slab with
        // vmcnt(7..0) instead of vmcnt(19..12): every slab would stall on the loads issued a few cycles earlier.
        if (RING) load_w(nsl);
        if (nsl == 0) decode_tile(nt);
        load_slab(nxt, nsl);
        unshuffle(cur);
        compute(cur, RING ? (it & 1) * BUF : sl * BUF, sl == nslab - 1);
        if (sl == nslab - 1 && t < t_end) epilogue(t);
        if (RING) {
            if (more) store_w((it + 1) & 1);
            __syncthreads();
        }
        t = nt;
        sl = nsl;
        ++it;
        return more;
    };
    for (;;) {
        if (!step(pA, pB)) break;
        if (!step(pB, pA)) break;
    }
}

// ------------------------------------------------------------------------------------------------ host
static constexpr size_t LDS_BUDGET = 150 * 1024;

struct StreamPlan {
    int nf = 0;        // 0: not supported
    bool ring = false;
};

static StreamPlan stream_plan(int dtype, const ConvArgs& a)
{
    StreamPlan p;
    const int esz = dtype_size(dtype);
    const long Cb = (long)a.Cin * esz;
    if ((a.ks != 1 && a.ks != 3) || a.head || a.out_f32) return p;       // detection levels stay on the implicit-GEMM kernel
    if (a.ks == 1 && a.stride != 1) return p;
    if (Cb % 16 != 0 || a.in_bytes == 0) return p;
    const long Kb = (long)a.ks * a.ks * Cb;
    const long nslab = (Kb + SLAB - 1) / SLAB;
    if ((long)a.Kpad * esz < nslab * SLAB) return p;                       // packed rows must cover whole slabs
    const bool ring_ok = !(a.opts & OPT_NO_RING) && a.Cout % 128 == 0;
    int fallback = 0;
    for (int nf : {8, 4, 2}) {                                             // resident weights
        const int nb = nf * 16;
        if (a.Cout % nb != 0) continue;
        if ((size_t)nb * nslab * SLAB + nb * 4 > LDS_BUDGET) continue;
        // every N tile re-reads the pixels: resident weights only pay when one or two tiles cover Cout
        if (a.Cout / nb > 2) { if (a.ks == 1 && !fallback) fallback = nf; continue; }
        p.nf = nf;
        return p;
    }
    // K of one or two slabs (<= 256 bf16 channels): an N tile's weights are loaded once per workgroup either way, and the resident form has no
    // barrier per slab -- 128->384 @160x160 244 -> 216 us, 256->512 @80x80 119 -> 96 us (tools/conv_micro.py); beyond that the ring
    if (ring_ok && !(fallback && nslab <= 2)) {                            // weights stream through a two-stage ring
        p.nf = 8;
        p.ring = true;
    } else {
        p.nf = fallback;
    }
    return p;
}

template <typename T, int KS, int MF, int NF, bool RING, bool UTAP, int KT, bool ONE = false, int FC = 0, typename TO = T>
static hipError_t stream_launch(const ConvArgs& a0, hipStream_t s, int n_cu)
{
    constexpr int NB = NF * 16, TPX = MF * 16;
    ConvArgs a = a0;
    const int cpt = a.Cin * (int)sizeof(T) / 16;
    a.cpt_shift = -1;
    for (int sh = 0; sh < 16; ++sh)
        if ((1 << sh) == cpt) a.cpt_shift = sh;
    const size_t Kb = (size_t)KS * KS * a.Cin * sizeof(T);
    const size_t nslab = (Kb + SLAB - 1) / SLAB;
    const size_t lds = (size_t)NB * SLAB * (RING ? 2 : nslab) + (sizeof(T) == 1 ? 2 : 1) * NB * 4 + (FC ? FC * FC * sizeof(T) + FC * 4 : 0);
    static size_t attr[16] = {0};
    auto kern = conv_stream_kernel<T, KS, MF, NF, RING, UTAP, KT, ONE, FC, TO>;
    {
        const hipError_t e = ensure_lds_attr(reinterpret_cast<const void*>(kern), lds, a.device, attr);
        if (e != hipSuccess) return e;
    }
    const int ntiles = (a.M + TPX - 1) / TPX;
    int gx = (ntiles + SW - 1) / SW;
    // one workgroup per CU in total (the kernel's registers / LDS allow no second one): with several N tiles the pixel
    // range is split over n_cu / gy persistent workgroups each, instead of gy rounds of short-lived ones that would each
    // pay the prologue (bias, first weight slab, first pixel slab) again
    const int gy = a.Cout / NB;
    const int cap = (a.opts & OPT_OLDGRID) ? n_cu : (n_cu / gy > 0 ? n_cu / gy : 1);
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL(kern, dim3(gx, a.Cout / NB), dim3(SW * 64), lds, s, a);
    return hipGetLastError();
}

// the fused forms: resident weights, one N tile covering all of Cout (64 -> fused 32, 128 -> fused 64)
template <typename T, int KT>
static hipError_t stream_dispatch_fused(const StreamPlan& p, const ConvArgs& a, hipStream_t s, int n_cu)
{
    const bool one = KT < 4 && (size_t)a.Cin * sizeof(T) <= 256;
    if (p.nf == 4) {
        if (one) return KT < 4 ? stream_launch<T, 1, 2, 4, false, true, (KT < 4 ? KT : 1), true, 32>(a, s, n_cu) : hipErrorNotSupported;
        return stream_launch<T, 1, 2, 4, false, true, KT, false, 32>(a, s, n_cu);
    }
    if (one) return KT < 4 ? stream_launch<T, 1, 2, 8, false, true, (KT < 4 ? KT : 1), true, 64>(a, s, n_cu) : hipErrorNotSupported;
    return KT == 4 ? stream_launch<T, 1, 2, 8, false, true, 4, false, 64>(a, s, n_cu) : hipErrorNotSupported;
}

template <typename T, int KS, int KT, typename TO = T>
static hipError_t stream_dispatch(const StreamPlan& p, bool utap, const ConvArgs& a, hipStream_t s, int n_cu, int* fused = nullptr)
{
    if constexpr (sizeof(T) >= 2 && std::is_same<T, TO>::value) {
        if (KS == 1 && utap && !p.ring && a.f2_w && a.f2_koff == 0 && a.f2_cin == a.f2_cout && p.nf * 16 == a.Cout &&
            ((p.nf == 4 && a.f2_cin == 32) || (p.nf == 8 && a.f2_cin == 64)) && !a.up2 && !a.res) {
            const size_t extra = (size_t)a.f2_cin * a.f2_cin * sizeof(T) + a.f2_cin * 4;
            const size_t Kb = (size_t)a.Cin * sizeof(T), nslab = (Kb + SLAB - 1) / SLAB;
            if ((size_t)a.Cout * SLAB * nslab + a.Cout * 4 + extra <= LDS_BUDGET) {
                const hipError_t e = stream_dispatch_fused<T, KT>(p, a, s, n_cu);
                if (e != hipErrorNotSupported) {
                    if (e == hipSuccess && fused) *fused = 1;
                    return e;
                }
            }
        }
    }
    if (!utap) {   // narrow inputs (the stem): per-lane tap, only built for the 32-channel tile
        if constexpr (KS == 3) { if (!p.ring && p.nf == 2) return stream_launch<T, 3, 4, 2, false, false, KT, false, 0, TO>(a, s, n_cu); }
        if constexpr (sizeof(T) == 1 && KS == 1 && KT == 1) {
            // 1x1 over 32 fp8 channels (32 bytes per pixel: half a K-step; the other half is masked per lane): the fp8 engine's
            // stage-1 bottleneck cv1, which would otherwise fall to the tile kernel
            if (!p.ring && (size_t)a.Cin * sizeof(T) <= 64) {
                switch (p.nf) {
                    case 8: return stream_launch<T, 1, 2, 8, false, false, 1, true, 0, TO>(a, s, n_cu);
                    case 4: return stream_launch<T, 1, 2, 4, false, false, 1, true, 0, TO>(a, s, n_cu);
                    default: return stream_launch<T, 1, 4, 2, false, false, 1, true, 0, TO>(a, s, n_cu);
                }
            }
        }
        return hipErrorNotSupported;
    }
    if (p.ring) return KT == 4 ? stream_launch<T, KS, 2, 8, true, true, 4, false, 0, TO>(a, s, n_cu) : hipErrorNotSupported;
    if (KS == 1 && KT < 4 && (size_t)a.Cin * sizeof(T) <= 256) {   // one slab of K: fetch only its real K-steps
        switch (p.nf) {
            case 8: return stream_launch<T, 1, 2, 8, false, true, KT, true, 0, TO>(a, s, n_cu);
            case 4: return stream_launch<T, 1, 2, 4, false, true, KT, true, 0, TO>(a, s, n_cu);
            default: return stream_launch<T, 1, 4, 2, false, true, KT, true, 0, TO>(a, s, n_cu);
        }
    }
    switch (p.nf) {
        case 8:
            if (KT == 4) return stream_launch<T, KS, 2, 8, false, true, 4, false, 0, TO>(a, s, n_cu);
            [[fallthrough]];   // two loop bodies + 128 channels spill: use two 64-channel tiles instead
        case 4: return stream_launch<T, KS, 2, 4, false, true, KT, false, 0, TO>(a, s, n_cu);
        default: return stream_launch<T, KS, 4, 2, false, true, KT, false, 0, TO>(a, s, n_cu);
    }
}

template <typename T, typename TO = T>
static hipError_t stream_dispatch_t(const StreamPlan& p, const ConvArgs& a, hipStream_t s, int n_cu, int* fused)
{
    const int Cb = a.Cin * (int)sizeof(T);
    const bool utap = ((Cb / 16) & 3) == 0;
    const int Kb = a.ks * a.ks * Cb;
    const int nslab = (Kb + SLAB - 1) / SLAB;
    const int kt = ((Kb - (nslab - 1) * SLAB) + 63) >> 6;
    if (a.ks == 1) {
        switch (kt) {
            case 4: return stream_dispatch<T, 1, 4, TO>(p, utap, a, s, n_cu, fused);
            case 2: return stream_dispatch<T, 1, 2, TO>(p, utap, a, s, n_cu, fused);
            case 1: return stream_dispatch<T, 1, 1, TO>(p, utap, a, s, n_cu, fused);
            default: return hipErrorNotSupported;
        }
    }
    if constexpr (!std::is_same<T, TO>::value) return hipErrorNotSupported;      // mixed types: 1x1 only
    else {
        switch (kt) {
            case 4: return stream_dispatch<T, 3, 4>(p, utap, a, s, n_cu);
            case 2: return stream_dispatch<T, 3, 2>(p, utap, a, s, n_cu);
            case 1: return stream_dispatch<T, 3, 1>(p, utap, a, s, n_cu);
            default: return hipErrorNotSupported;
        }
    }
}

// returns hipErrorNotSupported when the shape is not covered (caller falls back to the implicit-GEMM kernel)
// second input (ConvArgs::in2): whole 256-byte slabs of K from it, 1x1, the up2 mapping needs even output sides
static bool in2_ok(int dtype, const ConvArgs& a)
{
    const int esz = dtype_size(dtype);
    if (a.ks != 1 || a.stride != 1 || a.head || a.src_mode || a.f2_w || a.up2) return false;
    if (a.in2_cin <= 0 || a.in2_cin >= a.Cin || ((long)a.in2_cin * esz) % SLAB != 0 || ((long)(a.Cin - a.in2_cin) * esz) % 16 != 0) return false;
    if (a.in2_bytes == 0 || a.in_bytes == 0 || a.ldi2 % (16 / esz) != 0) return false;
    if (a.in2_up2 && ((a.Ho | a.Wo) & 1)) return false;
    return !(a.opts & (OPT_NO_STREAM | OPT_NO_IN2));
}

bool conv_accepts_in2(int dtype, const ConvArgs& a)
{
    if (!in2_ok(dtype, a)) return false;
    if (a.out_dt >= 0 && a.out_dt != dtype && !(dtype == 2 && a.out_dt == 1)) return false;
    return stream_plan(dtype, a).nf != 0;
}

hipError_t launch_conv_stream(int dtype, const ConvArgs& a, hipStream_t s, int* variant, int* fused)
{
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
    if (a.src_mode) return hipErrorNotSupported;
    if (a.in2 && !in2_ok(dtype, a)) return hipErrorNotSupported;
    const bool mixed = a.out_dt >= 0 && a.out_dt != dtype;
    // mixed types: fp8 operands with a bf16 output (the fp8 engine's last neck convolutions, 1x1, no residual) run here; the rest
    // (its bf16 stem on shapes the narrow halo kernel leaves) on the tile kernel
    if (mixed && !(dtype == 2 && a.out_dt == 1 && a.ks == 1 && !a.f2_w)) return hipErrorNotSupported;
    const StreamPlan p = stream_plan(dtype, a);
    if (p.nf == 0) return hipErrorNotSupported;
    const hipError_t e = mixed ? stream_dispatch_t<fp8_t, __bf16>(p, a, s, n_cu, fused)
                         : dtype == 0 ? stream_dispatch_t<float>(p, a, s, n_cu, fused)
                         : dtype == 1 ? stream_dispatch_t<__bf16>(p, a, s, n_cu, fused) : stream_dispatch_t<fp8_t>(p, a, s, n_cu, fused);
    if (e == hipSuccess && variant) *variant = (p.ring ? 3000 : 2000) + p.nf * 16;
    return e;
}

}  // namespace sky
