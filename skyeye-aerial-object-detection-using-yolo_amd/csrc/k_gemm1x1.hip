// 1x1 convolutions with many input channels as a GEMM for gfx950 (MI355X): ConvolutionBlock(k = 1) of the CSP / SPP / neck layers
// (reference blocks.py:10-41, 105-123, 133-147; detector.py:196-229) with Cin a multiple of 64 (>= 192) and Cout a multiple of 256; chosen
// by default for Cin >= 512 and Cout >= 512 (the P5 stage of skyeye_s, most 1x1 layers of skyeye_l).
//
// conv_stream_kernel keeps a wave's pixels in registers and re-reads every weight fragment from LDS per 32 pixels: at K >= 256 the
// vector-memory path and the LDS reads per MFMA hold it at 450-600 TFLOP/s.  Here the layer is the GEMM
//     D[channel][pixel] = sum_k W[channel][k] * P[pixel][k]            (both operands K-contiguous: NHWC pixels, packed weight rows)
// with a 256-pixel x 256-channel tile per workgroup of eight waves (4 x 2, a wave owns 64 pixels x 128 channels = 32 accumulator
// fragments: 12 fragment reads per 32 MFMAs, and half the operand bytes per FLOP of a 128 x 256 tile), one workgroup per CU.
//   * K advances 64 channels (128 B per row -- a whole cache line per row piece -- two MFMA K-steps) per step.  A step's operands -- 256
//     pixel rows + 256 weight rows x 128 B = 64 KB -- come global -> LDS by LDS-DMA (buffer_load ... lds, 16 B per lane, 8 pieces per wave)
//     into a TWO-stage ring, one step ahead of the MFMAs; the ring runs on across tiles (the next tile's first step flies under this
//     tile's last MFMAs and epilogue).  One raw s_barrier per step; waits are counted (s_waitcnt vmcnt(N): pieces and epilogue stores
//     retire in issue order).
//   * the 16-byte chunk c of row r sits in slot c ^ ((r >> 1) & 7) (the bottleneck kernel's weight-slab swizzle): conflict-free for the
//     ds_read_b128 lane groups of MI355X_MICROARCH.md (tools/lds_bank_check.py).  The swizzle and the weight-row permutation ((fragment
//     j, MFMA row r) -> channel (j>>1)*32 + (r>>2)*8 + (j&1)*4 + (r&3): a lane ends up with 8 consecutive channels per fragment pair) are
//     applied by the DMA's per-lane SOURCE address; the destination of a piece is 1 KB contiguous.
//   * second input (ConvArgs::in2, the neck's concat read in place): the first in2_cin channels of K come from the small lateral map at the
//     up2-mapped pixel, as in conv_stream_kernel.
// Where the time goes (B = 16 @40x40, one round of 200 tiles, ablation builds `make exp G1_SKIP=..`): 1.47 us per 64-channel step with
// everything, 1.3 without the DMA, 1.1 without the MFMAs, 0.76 with neither (fragment reads + barrier) -- the K loop runs the MFMA pipe
// at ~65 %; the epilogue (3.8 us per tile: 128 values per lane through SiLU) is not overlapped.  It replaces the weight-ring form of
// conv_stream_kernel (Cout >= 512): 512->512 @40x40 B = 32 58.5 -> 47.5 us, 768->512 84.6 -> 63.6, 1024->512 107 -> 78.5.
// MFMA instruction, operand roles and K order are conv_stream_kernel's: the outputs are bit-identical to it (tests/test_gpu_gemm1x1.py).
#include "sky_kernels.h"

#include "conv_frag.h"

namespace sky {

// experiments build only (make exp G1_SKIP=<bits>): 1 no MFMAs, 2 no LDS-DMA after the prologue, 4 no epilogue arithmetic / stores, 8 no fragment reads
#if defined(SKY_EXPERIMENTS) && defined(G1_SKIP)
#define G1_OFF(bit) ((G1_SKIP) & (bit))
#else
#define G1_OFF(bit) 0
#endif

namespace g1 {
constexpr int MAXC = 2048;                                     // bias of every output channel stays in LDS
// NWM x NWN waves, a wave owns 64 pixels x NFJ * 16 channels; ROWB bytes of K per row and step; NST ring stages
template <int NWM_, int NWN_, int NFJ_, int ROWB_, int NST_>
struct Geo {
    static constexpr int NWM = NWM_, NWN = NWN_, NFJ = NFJ_, ROWB = ROWB_, NST = NST_;
    static constexpr int NW = NWM * NWN, NT = NW * 64;
    static constexpr int TM = NWM * 64, TN = NWN * NFJ * 16;  // pixels x channels per workgroup tile
    static constexpr int STG = (TM + TN) * ROWB;
    static constexpr int LDS_BYTES = NST * STG + MAXC * 4;
    static constexpr int KSS = ROWB / 64;                      // MFMA K-steps per step
    static constexpr int CH = ROWB / 16;                       // 16-byte chunks per row
    static constexpr int RPP = 1024 / ROWB;                    // rows per DMA piece (one wave instruction = 1 KB)
    static constexpr int NPP = TM / RPP / NW;                  // pixel pieces per wave and step
    static constexpr int NPW = TN / RPP / NW;                  // weight pieces per wave and step
    static constexpr int NP = NPP + NPW;
    static constexpr int NSTORE = 4 * (NFJ / 2);               // buffer stores per lane and epilogue
    static constexpr int LEAD = NST - 1;                       // steps between a piece's issue and its use
    static_assert(ROWB == 64 || ROWB == 128, "row swizzles below");
    static_assert(TM % (RPP * NW) == 0 && TN % (RPP * NW) == 0 && (NPP % 2 == 0 || RPP == 16), "whole pieces per wave");
    static_assert((LEAD - 1) * NP + NSTORE < 64, "vmcnt is a 6-bit counter");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};
// the 16-byte chunk c of a row whose index within its 16-row fragment is r16 sits in slot c ^ swz(r16): conflict-free ds_read_b128 fragment reads
// (lane (fr, fq) reads row fr, chunk ks * 4 + fq) for the lane groups of MI355X_MICROARCH.md
template <int ROWB> __device__ __forceinline__ int swz(int r16) { return ROWB == 64 ? ((r16 >> 3) & 1) * 3 : (r16 >> 1) & 7; }
}  // namespace g1

__device__ __forceinline__ void g1_dma16(__amdgpu_buffer_rsrc_t rsrc, char* dst, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
}
template <int N>
__device__ __forceinline__ void g1_wait_vm()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void g1_barrier()
{
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// `younger` steps of NP pieces (0 .. LEAD - 1) and, with `st`, NSTORE stores are younger in the queue than the pieces waited for
template <typename GEO, int Y = 0>
__device__ __forceinline__ void g1_wait(int younger, bool st)
{
    if constexpr (Y < GEO::LEAD) {
        if (younger == Y) {
            if (st) g1_wait_vm<Y * GEO::NP + GEO::NSTORE>(); else g1_wait_vm<Y * GEO::NP>();
        } else {
            g1_wait<GEO, Y + 1>(younger, st);
        }
    }
}

// RES: a residual (ConvArgs::res, the output's element type) is added behind the activation (TransformerLayer's x + proj(..) / x + ffn(..),
// reference attention.py:244-309); its 16-byte vectors are loaded one pixel fragment ahead, the first four under the tile's last MFMAs
// T = __bf16, or fp8_t (round 4; config 5's engine): the same 128-byte rows hold 128 channels, both 64-byte halves of a row go into ONE
// 16x16x128 block-scaled instruction (fp8_mma128: the pairing of K-steps conv_stream_kernel's fp8 path uses, so the two stay bit-identical),
// the epilogue is the fp8 engine's acc * (input scale x weight scale) + bias, SiLU, * 1 / out scale, e4m3 (8-byte stores); no residual form.
// TO = __bf16 with T = fp8_t: the fp8 engine's last neck convolutions (the maps the detection levels read stay bf16, engine.cpp: neck).
template <typename GEO, bool RES, typename T = __bf16, typename TO = T>
__global__ void __launch_bounds__(GEO::NT, 2) gemm1x1_kernel(const ConvArgs a)
{
    using namespace g1;
    constexpr int NT = GEO::NT, TM = GEO::TM, TN = GEO::TN, ROWB = GEO::ROWB, STG = GEO::STG, NST = GEO::NST, NFJ = GEO::NFJ;
    constexpr int CH = GEO::CH, RPP = GEO::RPP, NPP = GEO::NPP, NPW = GEO::NPW, KSS = GEO::KSS, LEAD = GEO::LEAD;
    constexpr int ESZ = (int)sizeof(T), OSZ = (int)sizeof(TO);
    constexpr bool F8 = ESZ == 1;
    static_assert(!F8 || (!RES && ROWB == 128), "fp8: 128-byte rows (one 16x16x128 instruction per row), no residual form");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    float* const lbias = reinterpret_cast<float*>(smem + NST * STG);
    float* const lmult = lbias + MAXC;                          // fp8: per-channel multipliers behind the biases

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int wm = wave / GEO::NWN, wn = wave % GEO::NWN;      // the wave's 64-pixel / NFJ * 16-channel part of the tile
    const int kshift = (ROWB == 64 ? 5 : 6) + (F8 ? 1 : 0);   // log2 of the channels per step
    const int nk = a.Cin >> kshift;                            // steps per tile
    const bool dual = a.in2 != nullptr;
    const int k2 = dual ? a.in2_cin >> kshift : 0;             // steps of K that come from in2
    const int gy = a.Cout / TN;
    const int mtiles = (a.M + TM - 1) / TM;
    const int nitems = mtiles * gy;
    const int wpitch = a.Kpad * ESZ;

    // workgroups that share an XCD (blockIdx % 8) take neighbouring items: the N tiles of one pixel tile meet in that XCD's L2
    const int G = gridDim.x;
    const int L = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (L >= nitems) return;                                   // (uniform; before any barrier)
    const int my_items = (nitems - L + G - 1) / G;
    const int total = my_items * nk;

    for (int i = tid; i < a.Cout; i += NT) lbias[i] = a.bias[i];
    if (F8)
        for (int i = tid; i < a.Cout; i += NT) lmult[i] = a.mult ? a.mult[i] : 1.0f;

    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, (int)a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t irsrc2 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(dual ? a.in2 : a.in), 0, (int)(dual ? a.in2_bytes : a.in_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, (int)((long)a.Cout * wpitch), 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)a.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(RES ? a.res : a.in), 0, (int)(RES ? a.res_bytes : a.in_bytes), 0x00020000);

    // ---- DMA side: lane l of a piece writes row (l / CH), slot (l % CH) of its RPP rows and fetches chunk slot ^ swz(row) ----
    const int drow = lane / CH, dslot = lane % CH;
    int voffw[NPW];                                            // weights: rows (wave * NPW + q) * RPP + drow of the tile's TN
#pragma unroll
    for (int q = 0; q < NPW; ++q) {
        const int R = (wave * NPW + q) * RPP + drow;           // LDS weight row = (wave column, fragment j, MFMA row r)
        const int j = (R >> 4) % NFJ, r = R & 15;
        const int ch = (R / (NFJ * 16)) * (NFJ * 16) + (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3);
        voffw[q] = ch * wpitch + (dslot ^ swz<ROWB>(r)) * 16;
    }
    int vp[NPP], vp2[NPP];                                     // pixels: rows (wave * NPP + pp) * RPP + drow of the tile
    int i_item = L, i_k = 0, i_wsoff = 0;
    auto set_item = [&]() {
        const int mt = i_item / gy, nt = i_item - mt * gy;
        i_wsoff = nt * TN * wpitch;
#pragma unroll
        for (int pp = 0; pp < NPP; ++pp) {
            const int row = (wave * NPP + pp) * RPP + drow;
            const int m = mt * TM + row;
            const int cb = (dslot ^ swz<ROWB>(row & 15)) * 16;
            const bool ok = m < a.M;
            vp[pp] = ok ? m * a.ldi * ESZ + cb : -1;
            vp2[pp] = -1;
            if (dual && ok) {
                int m2 = m;
                if (a.in2_up2) {
                    const int x = m % a.Wo;
                    const int q = m / a.Wo;
                    const int y = q % a.Ho;
                    const int b = q / a.Ho;
                    m2 = (b * (a.Ho >> 1) + (y >> 1)) * (a.Wo >> 1) + (x >> 1);
                }
                vp2[pp] = m2 * a.ldi2 * ESZ + cb;
            }
        }
    };
    auto issue = [&](int stage) {
        char* const sb = smem + stage * STG;
        const bool from2 = i_k < k2;                           // (uniform)
        const int kb = (from2 ? i_k : i_k - k2) * ROWB;
        if (from2) {
#pragma unroll
            for (int pp = 0; pp < NPP; ++pp) g1_dma16(irsrc2, sb + (wave * NPP + pp) * 1024, vp2[pp] < 0 ? -1 : vp2[pp] + kb, 0);
        } else {
#pragma unroll
            for (int pp = 0; pp < NPP; ++pp) g1_dma16(irsrc, sb + (wave * NPP + pp) * 1024, vp[pp] < 0 ? -1 : vp[pp] + kb, 0);
        }
        const int ws = i_wsoff + i_k * ROWB;
#pragma unroll
        for (int q = 0; q < NPW; ++q) g1_dma16(wrsrc, sb + TM * ROWB + (wave * NPW + q) * 1024, voffw[q], ws);
        if (++i_k == nk) {
            i_k = 0;
            i_item += G;
            set_item();                                        // (past the last item: m >= M everywhere, the pieces are never issued)
        }
    };

    // ---- MFMA side ----
    const int lane_off = fr * ROWB + ((fq ^ swz<ROWB>(fr)) << 4);       // K-step ks of a 128-byte row: ^ (ks << 6)
    const int pbase = lane_off + wm * (64 * ROWB);
    const int wbase = lane_off + TM * ROWB + wn * (NFJ * 16 * ROWB);
    f32x4_t acc[NFJ][4];
#pragma unroll
    for (int j = 0; j < NFJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    set_item();
#pragma unroll
    for (int p = 0; p < LEAD; ++p)
        if (p < total) issue(p);
    __syncthreads();                                           // bias visible (drains the first pieces too: once)

    int c_item = L, c_k = 0;
    // bf16: the accumulators of an item start from the bias of ITS N tile (k_conv_halo.hip: acc_start explains); fp8: from zero
    auto acc_bias = [&](int item) {
        const int nb0 = (item % gy) * TN + wn * (NFJ * 16) + fq * 8;
#pragma unroll
        for (int j = 0; j < NFJ; ++j) {
            const f32x4_t b = F8 ? f32x4_t{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4_t*>(lbias + nb0 + (j >> 1) * 32 + (j & 1) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = b;
        }
    };
    acc_bias(c_item);
    int after_epi = 0;                                         // steps since an epilogue whose stores are still counted (0: none)
    int stage = 0;
    for (int g = 0; g < total; ++g) {
        // pieces of step g landed (mine; the barrier makes it everyone's).  Younger in the queue: the pieces of the next LEAD - 1 steps
        // and, for LEAD steps after an epilogue, its stores.
        {
            const int rem = total - 1 - g;
            g1_wait<GEO>(rem < LEAD - 1 ? rem : LEAD - 1, after_epi != 0);
            if (after_epi && ++after_epi == LEAD + 1) after_epi = 0;
        }
        g1_barrier();
        if (g + LEAD < total && !G1_OFF(2)) issue(stage == 0 ? NST - 1 : stage - 1);   // stage (g + LEAD) % NST: read last in step g - 1, before this barrier
        const char* const sb = smem + stage * STG;
        Out8<__bf16>::raw_t rres[2][NFJ / 2];
        if (RES && c_k == nk - 1) {                            // residual vectors of pixel fragment 0: under this step's MFMAs
            const int mt = c_item / gy, nt = c_item - mt * gy;
            const int m = mt * TM + wm * 64 + fr;
            const int rb = m < a.M ? (m * a.ldr + nt * TN + wn * (NFJ * 16) + fq * 8) * 2 : -1;
#pragma unroll
            for (int sg = 0; sg < NFJ / 2; ++sg) rres[0][sg] = Out8<__bf16>::load(rrsrc, rb < 0 ? -1 : rb + sg * 64, 0);
        }
        if constexpr (F8) {
            // fp8: the two 64-byte halves of every 128-byte row in one 16x16x128 instruction; weight fragments in two halves of NFJ / 2 so that
            // pixel + weight operands stay at 64 registers beside the 128 accumulator registers
            u32x4_t pf0[4], pf1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pf0[i] = *reinterpret_cast<const u32x4_t*>(sb + (pbase + i * (16 * ROWB)));
                pf1[i] = *reinterpret_cast<const u32x4_t*>(sb + ((pbase + i * (16 * ROWB)) ^ 64));
            }
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {
                u32x4_t wf0[NFJ / 2], wf1[NFJ / 2];
#pragma unroll
                for (int jj = 0; jj < NFJ / 2; ++jj) {
                    const int j = jh * (NFJ / 2) + jj;
                    wf0[jj] = *reinterpret_cast<const u32x4_t*>(sb + (wbase + j * (16 * ROWB)));
                    wf1[jj] = *reinterpret_cast<const u32x4_t*>(sb + ((wbase + j * (16 * ROWB)) ^ 64));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int jj = 0; jj < NFJ / 2; ++jj)
#pragma unroll
                    for (int i = 0; i < 4; ++i) fp8_mma128(wf0[jj], wf1[jj], pf0[i], pf1[i], acc[jh * (NFJ / 2) + jj][i]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < KSS; ++ks) {
            // all fragment reads of a K-step go out before its first MFMA (counted lgkmcnt waits follow from the order): hipcc otherwise
            // funnels the weight fragments through one register quad, read -> wait -> 4 MFMAs, per fragment
            u32x4_t pf[4], wf[NFJ];
#pragma unroll
            for (int i = 0; i < 4; ++i) pf[i] = G1_OFF(8) ? u32x4_t{(unsigned)lane, 1u, 2u, 3u} : *reinterpret_cast<const u32x4_t*>(sb + ((pbase + i * (16 * ROWB)) ^ (ks << 6)));
#pragma unroll
            for (int j = 0; j < NFJ; ++j) wf[j] = G1_OFF(8) ? u32x4_t{(unsigned)fr, 5u, 6u, 7u} : *reinterpret_cast<const u32x4_t*>(sb + ((wbase + j * (16 * ROWB)) ^ (ks << 6)));
            __builtin_amdgcn_sched_barrier(0);
            if (!G1_OFF(1)) {
#pragma unroll
                for (int j = 0; j < NFJ; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) S1<__bf16>::mma(wf[j], pf[i], acc[j][i]);
            } else {
#pragma unroll
                for (int j = 0; j < NFJ; ++j) acc[j][0][0] += __uint_as_float(wf[j][0] ^ pf[j & 3][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        stage = stage == NST - 1 ? 0 : stage + 1;
        if (++c_k == nk) {
            // ---- epilogue: bias, activation, bf16, 16-byte channel vectors straight from registers ----
            const int mt = c_item / gy, nt = c_item - mt * gy;
            const int n0 = nt * TN + wn * (NFJ * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = mt * TM + wm * 64 + i * 16 + fr;
                const int obase = m < a.M ? (m * a.ldo + n0 + fq * 8) * OSZ : -1;
                if (RES && i < 3) {
                    const int rb = m + 16 < a.M ? ((m + 16) * a.ldr + n0 + fq * 8) * 2 : -1;
#pragma unroll
                    for (int sg = 0; sg < NFJ / 2; ++sg) rres[(i + 1) & 1][sg] = Out8<__bf16>::load(rrsrc, rb < 0 ? -1 : rb + sg * 64, 0);
                }
#pragma unroll
                for (int sg = 0; sg < NFJ / 2; ++sg) {
                    float v[8];
                    if constexpr (F8) {
                        const int nl = n0 + sg * 32 + fq * 8;
                        const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(lbias + nl), b1 = *reinterpret_cast<const f32x4_t*>(lbias + nl + 4);
                        const f32x4_t m0 = *reinterpret_cast<const f32x4_t*>(lmult + nl), m1 = *reinterpret_cast<const f32x4_t*>(lmult + nl + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {              // conv_stream_kernel's fp8 epilogue, operation by operation
                            v[e] = acc[2 * sg][i][e] * m0[e] + b0[e];
                            v[4 + e] = acc[2 * sg + 1][i][e] * m1[e] + b1[e];
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {              // (the bias was the accumulators' initial value)
                            v[e] = acc[2 * sg][i][e];
                            v[4 + e] = acc[2 * sg + 1][i][e];
                        }
                    }
                    if (a.act == ACT_SILU && !G1_OFF(4)) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = S1<T>::silu(v[e]);
                    } else if (a.act == ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
                    }
                    if (RES) Out8<__bf16>::add(rres[i & 1][sg], v, a.res_scale);
                    const typename Out8<TO>::raw_t o = Out8<TO>::pack(v, a.out_inv_scale);
                    Out8<TO>::store(o, orsrc, obase < 0 ? -1 : obase + sg * 32 * OSZ);      // (always issued: the waits count it)
                }
            }
            c_k = 0;
            c_item += G;
            acc_bias(c_item);                                  // (past the last item: a valid N tile's bias, never used)
            after_epi = 1;
        }
    }
}

// The shipped shape: C = 256 x 256 tile, 128-byte rows, 2 stages (132 KB), one workgroup of 8 waves per CU.  Measured against it on the P5 layers
// (experiments/README.md): A = 128 x 256 tile, 64-byte rows, 3 stages, two workgroups of 4 waves per CU -- every row piece is half a cache line and
// the L2 -> LDS path runs at half its rate; B = 128 x 128 tile, 128-byte rows, 2 stages, two workgroups per CU -- twice the operand bytes per FLOP.
using GeoC = g1::Geo<4, 2, 8, 128, 2>;
#ifdef SKY_EXPERIMENTS
using GeoA = g1::Geo<2, 2, 8, 64, 3>;
using GeoB = g1::Geo<2, 2, 4, 128, 2>;
static int g1_shape(const ConvArgs&)
{
    static const int forced = [] { const char* v = getenv("SKY_GEMM1X1_SHAPE"); return v ? atoi(v) : 2; }();
    return forced;
}
#else
static int g1_shape(const ConvArgs&) { return 2; }
#endif

// conv_stream_kernel's large-K cases this kernel takes over
bool gemm1x1_ok(int dtype, const ConvArgs& a)
{
    using namespace g1;
    if ((dtype != 1 && dtype != 2) || (a.out_dt >= 0 && a.out_dt != dtype && !(dtype == 2 && a.out_dt == 1))) return false;
    if (a.ks != 1 || a.stride != 1 || a.head || a.up2 || a.src_mode || a.f2_w || a.c1_w || a.out_f32) return false;
    const int shape = dtype == 2 ? 2 : g1_shape(a);
    const int esz = dtype == 2 ? 1 : 2;
    if (dtype == 2 && (a.res || a.ldi % 16 != 0 || (a.in2 && a.ldi2 % 16 != 0))) return false;      // fp8: 16-byte row pieces, no residual form
    const int tn = shape == 1 ? 128 : 256, kstep = (shape == 0 ? 32 : 64) * (2 / esz);
    if (a.Cout % tn != 0 || a.Cout > MAXC || a.Cin % kstep != 0 || a.Cin < 3 * kstep || a.Kpad < a.Cin) return false;
    if (a.ldi % 8 != 0 || a.ldo % 8 != 0 || a.in_bytes == 0 || a.out_bytes == 0 || a.M <= 0) return false;
    if ((reinterpret_cast<size_t>(a.in) | reinterpret_cast<size_t>(a.out) | reinterpret_cast<size_t>(a.w)) & 15) return false;
    if ((double)a.Cout * a.Kpad * esz >= 2147483000.0) return false;
    if (a.res && (a.res_bytes == 0 || a.ldr % 8 != 0 || (reinterpret_cast<size_t>(a.res) & 15))) return false;
    if (a.in2) {
        if (a.in2_cin <= 0 || a.in2_cin >= a.Cin || a.in2_cin % kstep != 0 || a.in2_bytes == 0 || a.ldi2 % 8 != 0) return false;
        if (a.in2_up2 && ((a.Ho | a.Wo) & 1)) return false;
        if (reinterpret_cast<size_t>(a.in2) & 15) return false;
    }
    if (a.opts & (OPT_NO_STREAM | OPT_NO_GEMM1X1)) return false;
    if (a.opts & OPT_GEMM1X1_FORCE) return true;
    // where it pays (measured, tools/conv_micro.py): the layers conv_stream_kernel runs with its weight ring (Cout >= 512, K >= 512: the P5
    // stage), from half a tile per CU up; at Cout = 256 the resident-weight streamer reads the pixels once and is as fast or faster
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
    const long items = (long)((a.M + 255) / 256) * (a.Cout / 256);
    return shape == 2 && (a.Cin >= 512 || a.Cout >= 768) && a.Cin >= 256 && a.Cout >= 512 && 2 * items >= n_cu;
}

template <typename GEO, bool RES = false, typename T = __bf16, typename TO = T>
static hipError_t g1_launch(const ConvArgs& a, hipStream_t s, int per_cu)
{
    constexpr size_t LDS = GEO::LDS_BYTES + (sizeof(T) == 1 ? g1::MAXC * 4 : 0);          // fp8: the multipliers behind the biases
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
    const long items = (long)((a.M + GEO::TM - 1) / GEO::TM) * (a.Cout / GEO::TN);
    const long slots = (long)per_cu * n_cu;
    int gx = (int)(items < slots ? items : slots);
    if (gx >= 8) gx &= ~7;                                     // whole XCD rounds (the item order in the kernel)
    static size_t attr[16] = {0};
    {
        const hipError_t e = ensure_lds_attr(reinterpret_cast<const void*>(gemm1x1_kernel<GEO, RES, T, TO>), LDS, a.device, attr);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((gemm1x1_kernel<GEO, RES, T, TO>), dim3(gx), dim3(GEO::NT), LDS, s, a);
    return hipGetLastError();
}

hipError_t launch_gemm1x1(int dtype, const ConvArgs& a, hipStream_t s, int* variant)
{
    if (!gemm1x1_ok(dtype, a)) return hipErrorNotSupported;
    const int shape = g1_shape(a);
    if (variant) *variant = 3256;
    if (dtype == 2) return a.out_dt == 1 ? g1_launch<GeoC, false, fp8_t, __bf16>(a, s, 1) : g1_launch<GeoC, false, fp8_t>(a, s, 1);
#ifdef SKY_EXPERIMENTS
    if (shape == 0 && !a.res) return g1_launch<GeoA>(a, s, 2);
    if (shape == 1 && !a.res) return g1_launch<GeoB>(a, s, 2);
#endif
    (void)shape;
    return a.res ? g1_launch<GeoC, true>(a, s, 1) : g1_launch<GeoC, false>(a, s, 1);
}

}  // namespace sky
