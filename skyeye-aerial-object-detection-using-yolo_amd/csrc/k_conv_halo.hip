// Halo-tile 3x3 convolution for gfx950 (MI355X): the stride-1, pad-1 3x3 ConvolutionBlocks with wide inputs
// (reference blocks.py:10-41; cv2 of BottleneckBlock with its residual add, blocks.py:88-90).
//
// The streaming kernel (k_conv_stream.hip) fetches every filter tap of every pixel from global memory: nine trips
// through the texture addresser per input byte, which is what bounds it (PMC: GRBM_TA_BUSY ~ 90 %, MFMA ~ 20 %).
// Here a workgroup owns a 16 x 16 tile of output pixels and brings the 18 x 18 input tile (tile + 1 pixel halo) into
// LDS ONCE per 128 bytes of input channels; all nine taps are then LDS reads.
//
//   * 4 waves per workgroup, 2 workgroups per CU (LDS 76.5 KB each): while one workgroup waits for its halo the other
//     owns the matrix pipes.  Wave w computes rows 4w..4w+3 of the tile (four 16-pixel MFMA fragments) against all
//     N_blk = 64 or 128 output channels: 16 or 32 MFMAs per pair of 4 + N_blk/16 fragment reads.
//   * HALO LAYOUT  [k-group plane f = 0..3][pixel p = hy*18 + hx, 352 slots][2 x 16 B]: the B operand of
//     v_mfma_*_16x16x32 wants lane (fr, fq) to hold 16 bytes of K-group fq of pixel fr; planes are 11264 B apart
//     (a multiple of 256) and the two 16-byte slots of a pixel are swapped on every other run of 8 pixels, so the 16
//     lanes of every ds_read_b128 lane group hit 16 distinct bank quads for ANY tap offset (tools/lds_bank_check.py).
//     The tile is written by LDS-DMA (buffer_load ... lds, 16 B per lane, no VGPR round trip): one instruction fills
//     1 KB = 32 consecutive pixels of one plane; the per-lane GLOBAL address does the de-interleave, and pixels outside
//     the image use offset 0xffffffff -> the range check writes zeros (= the conv's zero padding).
//   * WEIGHTS stream through a two-stage LDS ring, one slab = [N_blk rows][128 B] = one tap x 128 B of input channels,
//     also by LDS-DMA, 16-byte chunks XOR-swizzled by (row >> 1) & 7 (conflict-free fragment reads).  Rows are
//     permuted exactly like the streaming kernel's (fragment j, MFMA row r -> channel (j>>1)*32 + (r>>2)*8 + (j&1)*4 +
//     (r&3)) so a lane ends up with 8 consecutive output channels per fragment pair -> 16-byte stores from registers.
//   * one barrier per tap (64-byte K-steps x 2); the next slab's DMA is issued right after it.
#include "sky_kernels.h"

#ifndef SKY_HALO_VGPR
#define SKY_HALO_VGPR 256      // VGPR budget of conv_halo_kernel (experiments: -DSKY_HALO_VGPR=N)
#endif

#include "conv_frag.h"

// Kernel experiments (stage switches, clock stamps, LDS padding) exist only in builds with -DSKY_EXPERIMENTS: the shipped
// library has no code path that skips work, whatever SKY_CONV_DBG says (tests/test_gpu_conv_halo.py checks that).
#ifdef SKY_EXPERIMENTS
#define SKY_DBG(a) ((a).dbg)
#else
#define SKY_DBG(a) 0
#endif

#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

namespace sky {

static constexpr int HWV = 4;                 // waves per workgroup

static constexpr int HPIX = 352;              // pixel slots per plane (324 used), 11 DMA instructions of 32 pixels
static constexpr int HPL = HPIX * 32;         // bytes per plane: 11264 = 44 * 256
static constexpr int HALO_BYTES = 4 * HPL;    // 45056
static constexpr int HDMA = HPIX / 32;        // DMA instructions per plane

// Device functions rather than inline builtins: LDS-DMA / s_waitcnt builtins reached from the __global__ template body
// do not type-check in the host pass and hipcc then silently drops the kernel's host stub.
__device__ __forceinline__ void wait_vmcnt0() { __builtin_amdgcn_s_waitcnt(0x0F70); }
// experiment (SKY_CONV_DBG & 256): shader-clock stamps of one wave's second tile, kept in LDS, dumped at kernel end
__device__ __forceinline__ void dbg_stamp(const ConvArgs& a, unsigned long long* st, int nth_tile, int slot)
{
    if ((SKY_DBG(a) & 256) && nth_tile == 1 && threadIdx.x == 0) st[slot] = __builtin_amdgcn_s_memtime();
}
// 16 bytes per lane: global (buffer rsrc, per-lane byte offset voff + uniform soff) -> LDS at `dst` + lane * 16
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, char* dst, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
}

// BIAS IN THE ACCUMULATOR (bf16 engine, round 4): the accumulators of a tile start as the bias of their output channels instead of zero
// -- the first MFMA of every chain takes them as its C operand -- so the epilogue has one VALU addition per value less (of ~5.5
// instructions per SiLU value).  The sum is then rounded as bias + k0 + k1 + ... instead of k0 + k1 + ... + bias: EVERY bf16 kernel that
// can compute a layer does it (conv_frag.h: BiasInAcc), fused and layer-by-layer forms stay bit-identical.  fp32 (exact mode) and fp8
// (acc * multiplier + bias) keep the addition.  Fragment j of a lane: channels (j >> 1) * 32 + fq * 8 + (j & 1) * 4 .. + 3 of the N tile.
template <typename T, int NF>
__device__ __forceinline__ void acc_start(f32x4_t (&acc)[NF][4], const float* lbias, int fq)
{
    if (BiasInAcc<T>::value) asm volatile("" : "+v"(fq));      // (opaque per tile: as loop invariants the NF bias vectors would occupy 4 NF registers for the whole kernel)
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        f32x4_t b = {0.f, 0.f, 0.f, 0.f};
        if (BiasInAcc<T>::value) b = *reinterpret_cast<const f32x4_t*>(lbias + (j >> 1) * 32 + fq * 8 + (j & 1) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = b;
    }
}

// Pixel `idx` (0..255, row-major) of a th x tw tile: tile-local row / column; row = -1 past the end of the tile.
// SQ: the 16 x 16 tile, known at compile time (one fragment per tile row: constants fold into the LDS offsets).
template <bool SQ>
__device__ __forceinline__ void tile_pixel(const ConvArgs& a, int idx, int& ty, int& tx)
{
    if (SQ) {
        ty = idx >> 4;
        tx = idx & 15;
        return;
    }
    ty = (int)(((unsigned)idx * a.magic_w) >> 16);     // idx / tile_w (exact for idx < 512, tile_w <= 64)
    tx = idx - ty * a.tile_w;
    if (ty >= a.tile_h) ty = -1;
}

// Epilogue of one wave's 4 fragments (tile pixels idx0 + 16*i, this lane's column of each) x 16*NF channels: bias, activation,
// residual, pack, 16-byte stores straight from the accumulators (a lane owns 8 consecutive channels per fragment pair,
// first channel nlane + 32*s).  Per image row the residual vectors are requested first and the activation math of the
// row runs under their latency; buffer descriptors give 32-bit offsets and let masked lanes (pixels past the image
// edge) use offset -1: loads return zeros, stores are dropped.  Clears the accumulators.
// FC > 0: the packed vectors of channels [0, FC) are also kept in `bop` (the B operands of a fused 1x1, conv_frag.h).
template <typename T, int NF, int ACT, bool SQ, int FC = 0, typename TO = T>
__device__ __forceinline__ void tile_epilogue(const ConvArgs& a, f32x4_t (&acc)[NF][4], const float* lbias, __amdgpu_buffer_rsrc_t orsrc,
                                              __amdgpu_buffer_rsrc_t rrsrc, int bimg, int y0, int x0, int idx0, int nlane,
                                              u32x4_t (*bop)[FC / 32 ? FC / 32 : 1][FuseGeom<T>::H] = nullptr,
                                              const typename Out8<TO>::raw_t (*pre)[NF / 2] = nullptr)
{
    static_assert(FC == 0 || (std::is_same<T, TO>::value && sizeof(T) >= 2), "the fused 1x1 takes the packed output as its operand");
    constexpr int VB = Out8<TO>::NB;                // bytes of one 8-channel vector in the OUTPUT type
    constexpr bool QS = sizeof(T) == 1;            // fp8 operands: per-channel multiplier (input scale x weight scale)
    const bool has_res = a.res != nullptr;
    const int nl0 = nlane % (NF * 16);            // channel within the workgroup's N tile (bias in LDS)
    const float* lmult = lbias + NF * 16;         // [NB] multipliers behind the bias (staged only when QS)
    // STORE-DATA HAZARD (found in r01_k, tests/test_gpu_determinism.py): the channel-group constant of a store must go into the
    // VECTOR offset (folded by hipcc into the instruction's immediate), never into `soffset`.  A constant above 64 is not an inline
    // operand, lands in an SGPR, and LLVM's hazard recognizer assumes that a MUBUF store with a REGISTER soffset needs no wait state
    // before its data VGPRs are overwritten; it then scheduled the accumulator clear (v_mov v4, 0) directly behind
    // buffer_store_dwordx4 v[4:7] of the last vector of a tile.  On gfx950 with two workgroups per CU contending for the vector
    // memory path the store read lanes 12..15 of its first dword after that clear: zeros (bias bits in general) in the output.
    // `pin` additionally keeps the previous store's registers live across the next bias / residual loads.
    u32x4_t pin = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int ty, tx;
        tile_pixel<SQ>(a, idx0 + i * 16, ty, tx);
        const int oy = y0 + ty, ox = x0 + tx;
        const bool ok = ty >= 0 && oy < a.Ho && ox < a.Wo;
        const int m = (bimg * a.Ho + oy) * a.Wo + ox;
        // masked lanes: offset 0x80000000 stays out of range after the per-group constant is added (see the store below)
        const int ooff = ok ? (m * a.ldo + nlane) * (int)sizeof(TO) : (int)0x80000000;
        typename Out8<TO>::raw_t rv[NF / 2];
        if (pre) {                       // residual vectors already in registers (fused cv1: read from the LDS tile of x)
#pragma unroll
            for (int s = 0; s < NF / 2; ++s) rv[s] = pre[i][s];
        } else if (has_res) {
            const int roff = ok ? (m * a.ldr + nlane) * (int)sizeof(TO) : -1;
#pragma unroll
            for (int s = 0; s < NF / 2; ++s) rv[s] = Out8<TO>::load(rrsrc, roff, s * 4 * VB);
        }
#pragma unroll
        for (int s = 0; s < NF / 2; ++s) {
            const int nl = s * 32 + nl0;
            const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(lbias + nl);
            const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(lbias + nl + 4);
            float v[8];
            if (QS) {
                const f32x4_t m0 = *reinterpret_cast<const f32x4_t*>(lmult + nl);
                const f32x4_t m1 = *reinterpret_cast<const f32x4_t*>(lmult + nl + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[2 * s][i][e] * m0[e] + b0[e];
                    v[4 + e] = acc[2 * s + 1][i][e] * m1[e] + b1[e];
                }
            } else if (BiasInAcc<T>::value) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {                  // the bias is the accumulators' initial value (acc_start)
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
            // (bf16: the next tile's acc_start overwrites the zeros with the bias.  The bias is NOT stored here: as loop-carried values the 16 x
            // NF accumulator registers would stay live across the epilogue and the tile change, and hipcc spills 340 - 460 bytes per lane)
            acc[2 * s][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            acc[2 * s + 1][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (ACT == ACT_SILU) v[e] = S1<T>::silu(v[e]);
                if (ACT == ACT_RELU) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
            }
            if (has_res || pre) Out8<TO>::add(rv[s], v, a.res_scale);
            asm volatile("; previous store data live until here" ::"v"(pin));      // bias + residual of this group are in
            const typename Out8<TO>::raw_t o = Out8<TO>::pack(v, a.out_inv_scale);
            Out8<TO>::store(o, orsrc, ooff + s * 4 * VB);
            pin = Out8<TO>::last(o);
            if constexpr (FC > 0) {
                if (s < FC / 32) {
                    if constexpr (sizeof(T) == 2) bop[i][s][0] = o.a;
                    else { bop[i][s][0] = o.a; bop[i][s][FuseGeom<T>::H - 1] = o.b; }
                }
            }
        }
    }
}

// Step q of the 9 taps of one channel chunk.  Stride 1: taps in order, one halo tile.  Stride 2 (S2): the input pixel
// of output (y, x), tap (ky, kx) is (2y + ky - 1, 2x + kx - 1): its row parity is fixed by ky (even for ky = 1, odd for
// ky = 0 and 2), likewise its column.  So the input splits into four parity phases; each phase is staged as a
// (th+1) x (tw+1) halo tile of "cells" (cell (cy, cx) of phase (dy, dx) = pixel (2cy + dy, 2cx + dx)) and serves the
// 4, 2, 2 or 1 taps of that parity as ordinary unit-stride taps with cell offsets -1 (k = 0) or 0 (k = 1, 2).
struct TapStep { int tap, dy, dx, newhalo; };
template <bool S2>
__device__ __forceinline__ TapStep tap_step(int q)
{
    if (!S2) return TapStep{q, 0, 0, q == 0};
    // taps {0,2,6,8} odd/odd, {1,7} odd rows/even cols, {3,5} even rows/odd cols, {4} even/even
    const int tap = (int)((0x453718620ull >> (4 * q)) & 15);
    const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
    return TapStep{tap, ky != 1, kx != 1, q == 0 || q == 4 || q == 6 || q == 8};
}

// CV1 (bf16, one 128-byte chunk of input channels, i.e. BottleneckBlock(64, 64)): `in` is the bottleneck's INPUT x and the 1x1
// convolution cv1 (blocks.py:88: cv2(cv1(x))) runs on the halo tile before the taps: u = SiLU(W1 x + b1) is computed for the 18 x 18
// halo pixels with W1 resident in LDS and written back over x IN PLACE in the same bank-conflict-free layout (zeros outside the
// image: the 3x3's padding applies to u); the residual x of the tile's own pixels is read from the LDS tile into registers first.
// One launch and ~2/3 of the HBM traffic of the cv1 + 3x3 pair go away.  The output must not alias x (neighbouring tiles still read it).
template <typename T, int NF, bool SQ, bool S2, int FC = 0, typename TO = T, bool CV1 = false>
__global__ void __launch_bounds__(HWV * 64, 2) __attribute__((amdgpu_num_vgpr(SKY_HALO_VGPR))) conv_halo_kernel(const ConvArgs a)
{
    static_assert(!CV1 || (std::is_same<T, __bf16>::value && std::is_same<TO, __bf16>::value && !S2 && FC == 0), "fused cv1: bf16, stride 1");
    constexpr int NB = NF * 16;
    constexpr int NBS = sizeof(T) == 1 ? 2 * NB : NB;      // floats staged behind the weight ring: bias (+ fp8 multipliers)
    constexpr int WSLAB = NB * 128;               // bytes of one weight slab
    constexpr int WDMA = NB / 8 / HWV;            // weight DMA instructions per wave per slab (8 rows each)
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const halo = smem + (((SKY_DBG(a) >> 16) & 0xff) << 10);      // experiment: SKY_CONV_DBG bits 16..23 = KB of padding in front
    char* const wring = halo + HALO_BYTES;
    constexpr int NSLABS = CV1 ? 3 : 2;                                           // ring stages (+ the resident W1 slab of the fused cv1)
    char* const w1lds = wring + 2 * WSLAB;                                        // CV1: W1 as one weight slab [NB rows][128 B]
    float* const lbias = reinterpret_cast<float*>(halo + HALO_BYTES + NSLABS * WSLAB);
    char* const w2lds = halo + HALO_BYTES + NSLABS * WSLAB + NBS * 4;             // fused 1x1 (FC > 0): [FC rows][FC * sizeof(T)]
    float* const b2lds = reinterpret_cast<float*>(w2lds + FC * FC * (int)sizeof(T));
    unsigned long long* const stamps = reinterpret_cast<unsigned long long*>(b2lds + (FC ? FC : 0));   // 64 x 8 B, experiments only
    float* const b1lds = reinterpret_cast<float*>(stamps + 64);                   // CV1: b1 [NB] (behind the 512 bytes of experiment stamps)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (wave-uniform: scalar LDS-DMA targets, no v_readfirstlane -> m0 chain per piece)
    const int fr = lane & 15, fq = lane >> 4;
    const int Cb = a.Cin * (int)sizeof(T);
    const int nchunk = Cb >> 7;
#ifdef SKY_EXPERIMENTS
    if ((a.dbg & 256) && tid < 64) stamps[tid] = 0ull;
#endif
    const int n0 = blockIdx.y * NB;
    const int tile_w = SQ ? 16 : a.tile_w, tile_h = SQ ? 16 : a.tile_h;
    const int tiles_x = (a.Wo + tile_w - 1) / tile_w, tiles_y = (a.Ho + tile_h - 1) / tile_h;
    const int hpw = tile_w + 2, hpix = hpw * (tile_h + 2);
    const int ntile = a.B * tiles_y * tiles_x;
    const int pix_b = a.ldi * (int)sizeof(T);     // bytes between input pixels
    const int wpitch = a.Kpad * (int)sizeof(T);

    for (int i = tid; i < NB; i += HWV * 64) lbias[i] = a.bias[n0 + i];
    if (sizeof(T) == 1)
        for (int i = tid; i < NB; i += HWV * 64) lbias[NB + i] = a.mult ? a.mult[n0 + i] : 1.0f;
    if constexpr (FC > 0) fuse_stage<T, FC ? FC : 32>(a.f2_w, a.f2_Kpad, a.f2_bias, w2lds, b2lds, tid, HWV * 64);

    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, (int)a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, (int)((long)a.Cout * wpitch), 0x00020000);

    // ---- per-lane constants ----
    // weight DMA: instruction q of this wave fills LDS rows (wave*WDMA + q)*8 .. +7; lane -> row, swizzled chunk
    int wrel[WDMA];
#pragma unroll
    for (int q = 0; q < WDMA; ++q) {
        const int row = (wave * WDMA + q) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int j = row >> 4, r = row & 15;
        const int ch = (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3);
        wrel[q] = (n0 + ch) * wpitch + c * 16;
    }
    // fragment reads
    const int arow = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);               // weight fragment, K-step 0 (K-step 1: ^ 64)
    int pbi[4];                                                               // pixel fragments, tap (0,0)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int ty, tx;
        tile_pixel<SQ>(a, wave * 64 + i * 16 + fr, ty, tx);
        pbi[i] = fq * HPL + (ty < 0 ? 0 : ty * hpw + tx) * 32;               // past the tile: any valid slot (never stored)
    }

    // halo DMA: this wave fills plane `wave`; in instruction b lane -> pixel slot p = b*32 + (lane >> 1), 16-byte slot
    // lane & 1 (K-step (lane & 1) ^ (p >> 3 & 1) of the chunk).  Recomputed per call: once per 9 taps, and it keeps
    // 22 VGPRs free for the fragment pipeline.
    auto issue_halo = [&](int bimg, int y0, int x0, int chunk, int dy, int dx) {
        // halo slot (hy, hx) = input pixel (y0 - 1 + hy, x0 - 1 + hx), or with S2 phase pixel (2(y0-1+hy)+dy, 2(x0-1+hx)+dx)
        const int ry0 = S2 ? 2 * (y0 - 1) + dy : y0 - 1, rx0 = S2 ? 2 * (x0 - 1) + dx : x0 - 1;
        const int base = ((bimg * a.H + ry0) * a.W + rx0) * pix_b + chunk * 128 + wave * 16;
#pragma unroll
        for (int b = 0; b < HDMA; ++b) {
            const int p = b * 32 + (lane >> 1);
            const int hy = SQ ? (p * 3641) >> 16 : (int)(((unsigned)p * a.magic_h) >> 16), hx = p - hy * hpw;       // p / hpw
            const int kk = (lane & 1) ^ ((p >> 3) & 1);
            const int sy = S2 ? 2 * hy : hy, sx = S2 ? 2 * hx : hx;
            const bool ok = p < hpix && (unsigned)(ry0 + sy) < (unsigned)a.H && (unsigned)(rx0 + sx) < (unsigned)a.W;
            lds_dma16(irsrc, halo + wave * HPL + b * 1024, ok ? base + (sy * a.W + sx) * pix_b + kk * 64 : -1, 0);
        }
    };
    auto issue_w = [&](int tap, int chunk, int buf) {
        const int kb = tap * Cb + chunk * 128;
#pragma unroll
        for (int q = 0; q < WDMA; ++q)
            lds_dma16(wrsrc, wring + buf * WSLAB + (wave * WDMA + q) * 1024, wrel[q], kb);
    };

    f32x4_t acc[NF][4];
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // Exact mode (fp32, 64-channel tiles): TWO-LEVEL summation.  One accumulator chain over K = 9 taps x Cin is a chain of K / 4 fp32
    // MFMA steps whose rounding error grows like sqrt(K): against float64 a 3x3 256 -> 256 layer measured 5.4e-7 relative rms, 3.3 x
    // the reference's oneDNN convolution (1.65e-7; tools/fp32_layer_error.py).  Every three taps (24 MFMA steps)
    // the partial sum moves into a second set of accumulators, so no chain is longer than max(24, 3 x chunks) additions.
    constexpr bool TWO_LEVEL = std::is_same<T, float>::value && NF == 4;
    f32x4_t tot[TWO_LEVEL ? NF : 1][4];
    if constexpr (TWO_LEVEL) {
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) tot[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    // ---- fused cv1 (CV1) ----
    typename Out8<TO>::raw_t resv[4][NF / 2];          // residual x of this lane's output vectors, taken from the LDS tile
    if constexpr (CV1) {
        // W1 [NB rows][128 B] goes into LDS once per workgroup, laid out like a ring slab (same row permutation and swizzle)
        const __amdgpu_buffer_rsrc_t w1rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.c1_w), 0, (int)((long)NB * a.c1_Kpad * (int)sizeof(T)), 0x00020000);
#pragma unroll
        for (int q = 0; q < WDMA; ++q) {
            const int row = (wave * WDMA + q) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            const int j = row >> 4, r = row & 15;
            const int ch = (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3);
            lds_dma16(w1rsrc, w1lds + (wave * WDMA + q) * 1024, ch * a.c1_Kpad * (int)sizeof(T) + c * 16, 0);
        }
        for (int i = tid; i < NB; i += HWV * 64) b1lds[i] = a.c1_bias[i];
    }
    auto cv1_phase = [&](int bimg, int y0, int x0, int nth_tile) {
        if constexpr (CV1) {
            dbg_stamp(a, stamps, nth_tile, 20);
            if (SKY_DBG(a) & 4096) return;            // experiments: bits 512 no SiLU, 1024 no cv1 MFMAs, 2048 no write-back, 4096 no cv1 phase
            // (1) the residual vectors of this lane's 4 x NF/2 output vectors: piece (plane fq, K-step s) of the centre pixel
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int A = (SQ ? pbi[0] + i * (18 * 32) : pbi[i]) + (hpw + 1) * 32;        // tap (1, 1)
                const int A0 = A + (((A >> 8) & 1) << 4);
#pragma unroll
                for (int s = 0; s < NF / 2; ++s) resv[i][s].a = *reinterpret_cast<const u32x4_t*>(halo + (s ? A0 ^ 16 : A0));
            }
            __syncthreads();                       // the centre pixels of a wave's output rows lie in slots other waves rewrite below
            dbg_stamp(a, stamps, nth_tile, 21);
            // (2) u = SiLU(W1 x + b1) on the 22 pixel fragments of the halo tile (352 slots), 6 per wave in two passes of 3
            const char* wb = w1lds + arow;
            const int k1 = 64 - 2 * (arow & 64);
            // the tile's own accumulators hold cv2's bias here (acc_start / the epilogue): three of the four pixel columns serve as cv1's
            // and start from cv1's bias; pass 0 leaves them at cv1's bias again, pass 1 at cv2's
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const f32x4_t b = *reinterpret_cast<const f32x4_t*>(b1lds + (j >> 1) * 32 + fq * 8 + (j & 1) * 4);
#pragma unroll
                for (int i = 0; i < 3; ++i) acc[j][i] = b;
            }
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                auto& au = acc;
                u32x4_t xf[3][2];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int p = (wave * 6 + pass * 3 + i) * 16 + fr;                       // slots past 351 do not exist: clamp, never stored
                    const int pc = p < HPIX ? p : HPIX - 1;
                    const int A = fq * HPL + pc * 32 + (((pc >> 3) & 1) << 4);
                    xf[i][0] = *reinterpret_cast<const u32x4_t*>(halo + A);
                    xf[i][1] = *reinterpret_cast<const u32x4_t*>(halo + (A ^ 16));
                }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < NF; ++j) {
                        const u32x4_t wf = *reinterpret_cast<const u32x4_t*>(wb + j * 2048 + (kk ? k1 : 0));
#pragma unroll
                        for (int i = 0; i < 3; ++i)
                            if (!(SKY_DBG(a) & 1024)) S1<T>::mma(wf, xf[i][kk], au[j][i]);
                    }
                dbg_stamp(a, stamps, nth_tile, 22 + 2 * pass);
                // (3) back into the tile, in place: this wave's pixels are read by nobody else before the barrier below
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int p = (wave * 6 + pass * 3 + i) * 16 + fr;
                    const int hy = SQ ? (p * 3641) >> 16 : (int)(((unsigned)p * a.magic_h) >> 16), hx = p - hy * hpw;
                    const bool inside = p < hpix && (unsigned)(y0 - 1 + hy) < (unsigned)a.H && (unsigned)(x0 - 1 + hx) < (unsigned)a.W;
#pragma unroll
                    for (int s = 0; s < NF / 2; ++s) {
                        const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(b1lds + s * 32 + fq * 8);
                        const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(b1lds + s * 32 + fq * 8 + 4);
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = au[2 * s][i][e];                         // (cv1's bias was the accumulators' initial value)
                            v[4 + e] = au[2 * s + 1][i][e];
                            if (!(SKY_DBG(a) & 512)) { v[e] = S1<T>::silu(v[e]); v[4 + e] = S1<T>::silu(v[4 + e]); }
                        }
                        au[2 * s][i] = pass == 0 ? b0 : *reinterpret_cast<const f32x4_t*>(lbias + s * 32 + fq * 8);
                        au[2 * s + 1][i] = pass == 0 ? b1 : *reinterpret_cast<const f32x4_t*>(lbias + s * 32 + fq * 8 + 4);
                        typename Out8<T>::raw_t o = Out8<T>::pack(v, 1.0f);
                        if (!inside) o.a = u32x4_t{0u, 0u, 0u, 0u};
                        if (p < HPIX && !(SKY_DBG(a) & 2048)) *reinterpret_cast<u32x4_t*>(halo + fq * HPL + p * 32 + ((s ^ ((p >> 3) & 1)) << 4)) = o.a;
                    }
                }
            }
            dbg_stamp(a, stamps, nth_tile, 25);
            __syncthreads();                       // u is complete
            dbg_stamp(a, stamps, nth_tile, 26);
        }
    };

    // One tap = 2 K-steps x NF/2 groups; group (kk, sp) = two weight fragments (8 consecutive channels per lane) x the
    // four pixel fragments of K-step kk.  Software pipeline: the weight pair of group g+2 is read from LDS before the 8
    // MFMAs of group g issue (three pairs live), the pixel fragments of K-step 1 are requested right after the first
    // weight pair; sched_barrier pins that order.
    auto compute_tap = [&](int tap, int buf) {
        const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
        const int toff = S2 ? ((ky != 0) * hpw + (kx != 0)) * 32 : (ky * hpw + kx) * 32;
        const char* wb = wring + buf * WSLAB + arow;
        int pa[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int A = (SQ ? pbi[0] + i * (18 * 32) : pbi[i]) + toff;
            pa[i] = A + (((A >> 8) & 1) << 4);
        }
        u32x4_t pf[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pf[0][i] = *reinterpret_cast<const u32x4_t*>(halo + pa[i]);
        if constexpr (sizeof(T) == 1) {
            // fp8: both 64-byte K-steps of the chunk go into ONE 16x16x128 instruction per (weight fragment, pixel fragment).
            // Pipeline over the NF/2 fragment pairs: the four weight pieces of pair sp + 1 are read before the 8 MFMAs of pair sp.
#pragma unroll
            for (int i = 0; i < 4; ++i) pf[1][i] = *reinterpret_cast<const u32x4_t*>(halo + (pa[i] ^ 16));
            // Pipeline over the NF weight fragments: the two 16-byte pieces (K-steps 0 / 1) of fragment j + 1 are read before the 4
            // MFMAs of fragment j issue (two stages of 2 x 4 VGPRs).
            u32x4_t w2[2][2];                        // [stage][kk]
            const int k1 = 64 - 2 * (arow & 64);
#pragma unroll
            for (int j = 0; j < NF + 1; ++j) {
                if (j < NF) {
                    w2[j & 1][0] = *reinterpret_cast<const u32x4_t*>(wb + j * 2048);
                    w2[j & 1][1] = *reinterpret_cast<const u32x4_t*>(wb + j * 2048 + k1);
                }
                if (j >= 1) {
                    const int q = j - 1;
#pragma unroll
                    for (int i = 0; i < 4; ++i) fp8_mma128(w2[q & 1][0], w2[q & 1][1], pf[0][i], pf[1][i], acc[q][i]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
        constexpr int HG = NF / 2, G = 2 * HG;
        u32x4_t wq[3][2];
#pragma unroll
        for (int g = 0; g < G + 2; ++g) {
            if (g < G) {
                const int kk = g / HG, sp = g % HG;
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    wq[g % 3][h] = *reinterpret_cast<const u32x4_t*>(wb + (2 * sp + h) * 2048 + (kk ? 64 - 2 * (arow & 64) : 0));
            }
            if (g == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) pf[1][i] = *reinterpret_cast<const u32x4_t*>(halo + (pa[i] ^ 16));
            }
            if (g >= 2) {
                const int gg = g - 2, kk = gg / HG, sp = gg % HG;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) S1<T>::mma(wq[gg % 3][h], pf[kk][i], acc[2 * sp + h][i]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)a.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res ? a.res : a.out), 0, (int)(a.res ? a.res_bytes : a.out_bytes), 0x00020000);
    auto epilogue_act = [&](int bimg, int y0, int x0) {
        if constexpr (FC > 0) {
            // this convolution's epilogue keeps the packed first FC channels, the fused 1x1 runs from them, second epilogue
            constexpr int C2 = FC ? FC : 32;
            u32x4_t bop[4][C2 / 32][FuseGeom<T>::H];
            if (a.act == ACT_SILU) tile_epilogue<T, NF, ACT_SILU, SQ, C2>(a, acc, lbias, orsrc, rrsrc, bimg, y0, x0, wave * 64 + fr, n0 + fq * 8, bop);
            else if (a.act == ACT_RELU) tile_epilogue<T, NF, ACT_RELU, SQ, C2>(a, acc, lbias, orsrc, rrsrc, bimg, y0, x0, wave * 64 + fr, n0 + fq * 8, bop);
            else tile_epilogue<T, NF, ACT_NONE, SQ, C2>(a, acc, lbias, orsrc, rrsrc, bimg, y0, x0, wave * 64 + fr, n0 + fq * 8, bop);
            f32x4_t acc2[C2 / 16][4];
            acc_start<T, C2 / 16>(acc2, b2lds, fq);
            fuse_gemm<T, C2, 4>(bop, w2lds, acc2, fr, fq);
            ConvArgs a2 = a;
            a2.res = nullptr;
            a2.ldo = a.f2_ldo;
            const __amdgpu_buffer_rsrc_t o2 = __builtin_amdgcn_make_buffer_rsrc(a.f2_out, 0, (int)a.f2_out_bytes, 0x00020000);
            if (a.f2_act == ACT_SILU) tile_epilogue<T, C2 / 16, ACT_SILU, SQ>(a2, acc2, b2lds, o2, o2, bimg, y0, x0, wave * 64 + fr, fq * 8);
            else if (a.f2_act == ACT_RELU) tile_epilogue<T, C2 / 16, ACT_RELU, SQ>(a2, acc2, b2lds, o2, o2, bimg, y0, x0, wave * 64 + fr, fq * 8);
            else tile_epilogue<T, C2 / 16, ACT_NONE, SQ>(a2, acc2, b2lds, o2, o2, bimg, y0, x0, wave * 64 + fr, fq * 8);
            return;
        }
        const typename Out8<TO>::raw_t (*pre)[NF / 2] = CV1 && a.c1_res ? resv : nullptr;
        if (a.act == ACT_SILU) tile_epilogue<T, NF, ACT_SILU, SQ, 0, TO>(a, acc, lbias, orsrc, rrsrc, bimg, y0, x0, wave * 64 + fr, n0 + fq * 8, nullptr, pre);
        else if (a.act == ACT_RELU) tile_epilogue<T, NF, ACT_RELU, SQ, 0, TO>(a, acc, lbias, orsrc, rrsrc, bimg, y0, x0, wave * 64 + fr, n0 + fq * 8, nullptr, pre);
        else tile_epilogue<T, NF, ACT_NONE, SQ, 0, TO>(a, acc, lbias, orsrc, rrsrc, bimg, y0, x0, wave * 64 + fr, n0 + fq * 8, nullptr, pre);
    };
    auto decode_tile = [&](int tile, int& bimg, int& y0, int& x0) {
        const int tx = tile % tiles_x;
        const int q = tile / tiles_x;
        bimg = q / tiles_y;
        y0 = (q - bimg * tiles_y) * tile_h;
        x0 = tx * tile_w;
    };

    // ---- persistent tile loop (everything below is uniform per workgroup) ----
    // A tile is the linear sequence g = chunk * 9 + q of (halo phase, tap) steps; weight slab g + 1 is requested while
    // step g computes, a new halo tile (new chunk or, with S2, new parity phase) after a barrier at its first step.
    int tile, tstep, tend;                                   // XCD-aware tile order (conv_frag.h: tile_walk)
    tile_walk(ntile, tile, tstep, tend);
    if (tile >= tend) return;
    int bimg, y0, x0;
    decode_tile(tile, bimg, y0, x0);
    const int G = nchunk * 9;
#ifdef SKY_EXPERIMENTS
    // experiment: phase offset between the two workgroups that share a CU (second half of the grid): bits 24..30 of SKY_CONV_DBG x 1024 cycles
    {
        // which workgroups are delayed: the second half of the grid, or (bit 30) the ones whose waves sit in an odd wave slot of their SIMD
        const unsigned slot = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11));      // HW_REG_HW_ID, wave_id
        const bool late = (a.dbg & (1 << 30)) ? (slot & 1u) != 0 : (int)blockIdx.x >= (int)gridDim.x / 2;
        if (late)
            for (int k = ((a.dbg >> 24) & 0x3f); k > 0; --k) __builtin_amdgcn_s_sleep(16);
    }
#endif
    const TapStep first = tap_step<S2>(0);
    __syncthreads();                                   // bias staged
    const TapStep second = tap_step<S2>(1);
    if (!(SKY_DBG(a) & 2)) issue_halo(bimg, y0, x0, 0, first.dy, first.dx);
    if (!(SKY_DBG(a) & 4)) { issue_w(first.tap, 0, 0); issue_w(second.tap, 0, 1); }
    int nth = 0;
    for (;;) {
        int chunk = 0, q = 0;
        if constexpr (BiasInAcc<T>::value) acc_start<T, NF>(acc, lbias, fq);      // this tile's accumulators start from the bias
        dbg_stamp(a, stamps, nth, 0);
        if ((SKY_DBG(a) & 256) && nth == 1 && threadIdx.x == 0) stamps[60] = __builtin_amdgcn_s_memrealtime();   // 100 MHz reference clock
        if ((SKY_DBG(a) & 256) && nth == 0 && threadIdx.x == 0) { stamps[56] = __builtin_amdgcn_s_memtime(); stamps[57] = __builtin_amdgcn_s_memrealtime(); }
        __builtin_amdgcn_s_setprio(1);                 // tap phase: this wave's MFMAs ahead of the other workgroup's epilogue arithmetic
        for (int g = 0; g < G; ++g) {
            const TapStep st = tap_step<S2>(q);
            if (st.newhalo && g > 0) {
                __syncthreads();                       // every wave is done with the halo tile
                if (!(SKY_DBG(a) & 2)) issue_halo(bimg, y0, x0, chunk, st.dy, st.dx);
            }
            wait_vmcnt0();                              // this wave's DMA (slab g, the halo) has landed (and its older stores)
            __syncthreads();                           // ... and everybody else's; compute(g - 1) is over everywhere
            if (CV1 && g == 0) cv1_phase(bimg, y0, x0, nth);
            int nq = q + 1, nchk = chunk;
            if (nq == 9) { nq = 0; ++nchk; }
            dbg_stamp(a, stamps, nth, 1 + 2 * g);
            // slabs 0 AND 1 of a tile are requested ahead of it (prologue / before the previous epilogue): behind that
            // epilogue's stores the first tap's DMA issue would stall for the length of the store drain (clock stamps)
            if (g > 0 && g + 1 < G && !(SKY_DBG(a) & 4)) issue_w(tap_step<S2>(nq).tap, nchk, (g + 1) & 1);
            if (!(SKY_DBG(a) & 1)) compute_tap(st.tap, g & 1);
            dbg_stamp(a, stamps, nth, 2 + 2 * g);
            if constexpr (TWO_LEVEL) {
                if (q % 3 == 2) {                       // (uniform) partial sum of three taps -> second level; nine taps per chunk: every chunk ends flushed
#pragma unroll
                    for (int j = 0; j < NF; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) { tot[j][i] += acc[j][i]; acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
                }
            }
            q = nq;
            chunk = nchk;
        }
        if constexpr (TWO_LEVEL) {
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc[j][i] = tot[j][i]; tot[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
        }
        // the next tile's first halo tile and weight slab are requested BEFORE this tile's epilogue: their latency runs
        // under the activation math and the stores
        const int next = tile + tstep;
        int nb = 0, ny0 = 0, nx0 = 0;
        if (next < tend) {
            decode_tile(next, nb, ny0, nx0);
            __syncthreads();
            dbg_stamp(a, stamps, nth, 40);
            if (!(SKY_DBG(a) & 2)) issue_halo(nb, ny0, nx0, 0, first.dy, first.dx);
            if (!(SKY_DBG(a) & 4)) { issue_w(first.tap, 0, 0); issue_w(second.tap, 0, 1); }
        }
        dbg_stamp(a, stamps, nth, 41);
        __builtin_amdgcn_s_setprio(0);
        if (!(SKY_DBG(a) & 8)) epilogue_act(bimg, y0, x0);
        dbg_stamp(a, stamps, nth, 42);
        if ((SKY_DBG(a) & 256) && nth == 1 && threadIdx.x == 0) stamps[61] = __builtin_amdgcn_s_memrealtime();
        if ((SKY_DBG(a) & 256) && nth == 0 && threadIdx.x == 0) { stamps[58] = __builtin_amdgcn_s_memtime(); stamps[59] = __builtin_amdgcn_s_memrealtime(); }
        if ((SKY_DBG(a) & 256) && nth == 1) {
            wait_vmcnt0();
            dbg_stamp(a, stamps, nth, 43);
        }
        ++nth;
        if (next >= tend) break;
        tile = next; bimg = nb; y0 = ny0; x0 = nx0;
    }
    if ((SKY_DBG(a) & 256) && a.raw && nth >= 2 && threadIdx.x == 0 && blockIdx.y == 0) {
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.raw) + (size_t)blockIdx.x * 64;
        for (int k = 0; k < 44; ++k) dst[k] = stamps[k];
        for (int k = 56; k < 62; ++k) dst[k] = stamps[k];
    }
}

// ------------------------------------------------------------------------------------------------ narrow inputs
// Same halo idea for 32 or 64 bytes of input channels per pixel (the FocusBlock convolution of the stem, blocks.py:
// 152-182; the 3x3 of the stage-1 bottlenecks): all of K = 9 taps x CB bytes fits in LDS next to TWO halo tiles, so
//   * the weights are loaded once per persistent workgroup as [slab][row][256 B] with the streaming kernel's swizzle
//     and row permutation; there is no weight ring and no barrier inside a tile;
//   * the halo tile of tile t+1 is fetched by LDS-DMA into the other buffer while tile t computes: one barrier per tile;
//   * halo layout [16-byte chunk plane c][pixel slot, 384][16 B]: a 64-byte K-step covers 4 chunks = CB/16 chunks of
//     64/CB taps, so with CB = 32 lanes of K-groups 2, 3 read the NEXT tap's pixel -- only a different LDS address.
//     16 consecutive pixels of a plane are 256 contiguous bytes and planes are 6144 B apart: conflict-free for any tap.
//   * 2 to 4 workgroups per CU (40 to 96 KB of LDS) hide the per-tile latencies; the layers are HBM-bound.
static constexpr int SPX = 384;               // pixel slots per plane (324 used): 6 DMA instructions of 64 pixels
static constexpr int SPL = SPX * 16;          // bytes per plane = 24 * 256

// SRC: 0 = the input is an NHWC tensor of T (LDS-DMA); 1 / 2 = the input is the caller's raw [B, 3, 2H, 2W] frame batch
// (uint8 / float32, NCHW) and FocusBlock's space-to-depth (blocks.py:176-181: patches TL, BL, TR, BR -> channel patch*3 + c),
// the /255 of the uint8 contract (validate.py:238, true division) and the conversion to T happen while the halo tile is
// built: the stem reads 4.9 MB per frame instead of the 13 MB intermediate an import kernel would write and it read back.
template <typename T, int CB, int NF, bool SQ, int SRC = 0, typename TO = T>
__global__ void __launch_bounds__(HWV * 64) conv_halo_small_kernel(const ConvArgs a)
{
    constexpr int NB = NF * 16;
    constexpr int NBS = sizeof(T) == 1 ? 2 * NB : NB;      // bias (+ fp8 multipliers)
    constexpr int NPL = CB / 16;                  // planes
    constexpr int HB = NPL * SPL;                 // one halo buffer
    constexpr int KBYTES = 9 * CB;
    constexpr int NSLAB = (KBYTES + 255) / 256;
    constexpr int NKS = (KBYTES + 63) / 64;       // 64-byte K-steps
    constexpr int WBUF = NB * 256;                // one slab of weights
    constexpr int NPIECE = NPL * 6 / HWV;         // halo DMA instructions per wave per tile
    constexpr int RAWP = 40;                      // bytes per staged raw row: 2 * 18 halo columns + 4 (16 x 16 tiles only)
    constexpr int RAW_ROWS = 3 * 36;              // colours x raw rows of the 18-row halo
    constexpr int RAW_BYTES = SRC == 1 ? RAW_ROWS * RAWP : 0;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const wlds = smem;
    char* const hlds = smem + NSLAB * WBUF;
    float* const lbias = reinterpret_cast<float*>(smem + NSLAB * WBUF + 2 * HB);
    // SRC == 1 (uint8 frames, 16 x 16 tiles): the raw bytes of a tile (3 colours x 36 rows x 40 bytes, rows starting 2 pixels
    // left of the halo so that they are 4-byte aligned) are staged in LDS by dword loads, then turned into the channel planes through a
    // 256-entry table of (T)(i / 255.0f) -- the exact values the import kernel would have written
    unsigned char* const rawl = reinterpret_cast<unsigned char*>(lbias + NBS);
    T* const lut = reinterpret_cast<T*>(rawl + RAW_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (wave-uniform: scalar LDS-DMA targets, no v_readfirstlane -> m0 chain per piece)
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.y * NB;
    const int tile_w = SQ ? 16 : a.tile_w, tile_h = SQ ? 16 : a.tile_h;
    const int tiles_x = (a.W + tile_w - 1) / tile_w, tiles_y = (a.H + tile_h - 1) / tile_h;
    const int hpw = tile_w + 2, hpix = hpw * (tile_h + 2);
    const int ntile = a.B * tiles_y * tiles_x;
    const int pix_b = a.ldi * (int)sizeof(T);
    int tile, tstep, tend;                                   // XCD-aware tile order (conv_frag.h: tile_walk)
    tile_walk(ntile, tile, tstep, tend);
    if (tile >= tend) return;

    // ---- weights and bias: once per workgroup ----
    {
        const char* wsrc = reinterpret_cast<const char*>(a.w);
        const long wpitch = (long)a.Kpad * (long)sizeof(T);
        for (int idx = tid; idx < NSLAB * NB * 16; idx += HWV * 64) {
            const int ss = idx / (NB * 16), rc = idx - ss * (NB * 16);
            const int row = rc >> 4, c = rc & 15;
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(wsrc + (long)(n0 + row) * wpitch + ss * 256 + c * 16);
            const int swz = (row & 3) | (((row >> 3) & 3) << 2);
            *reinterpret_cast<u32x4_t*>(wlds + ss * WBUF + row * 256 + ((c ^ swz) << 4)) = v;
        }
        for (int i = tid; i < NB; i += HWV * 64) lbias[i] = a.bias[n0 + i];
        if (sizeof(T) == 1)
            for (int i = tid; i < NB; i += HWV * 64) lbias[NB + i] = a.mult ? a.mult[n0 + i] : 1.0f;
        if constexpr (SRC == 1) lut[tid] = (T)((float)tid / 255.0f);          // 256 threads = 256 byte values
    }

    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, (int)a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)a.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res ? a.res : a.out), 0, (int)(a.res ? a.res_bytes : a.out_bytes), 0x00020000);

    // per-lane K-step constants: LDS offset of (chunk plane, tap shift) for this lane's 16 bytes of K-step ks
    int toff[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int kb = ks * 64 + fq * 16;
        int tap = kb / CB;
        const int c = (kb - tap * CB) >> 4;
        if (tap > 8) tap = 8;                       // K padding: the weights there are zero, the pixel must only be finite
        const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
        toff[ks] = c * SPL + (ky * hpw + kx) * 16;
    }
    int pbi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int ty, tx;
        tile_pixel<SQ>(a, wave * 64 + i * 16 + fr, ty, tx);
        pbi[i] = (ty < 0 ? 0 : ty * hpw + tx) * 16;
    }
    const int wrow0 = (fr >> 2) * 8 + (fr & 3);
    const int wsw0 = (wrow0 & 3) | (((wrow0 >> 3) & 3) << 2);
    const char* const wfrag = wlds + wrow0 * 256;

    auto decode_tile = [&](int t, int& bimg, int& y0, int& x0) {
        const int tx = t % tiles_x;
        const int q = t / tiles_x;
        bimg = q / tiles_y;
        y0 = (q - bimg * tiles_y) * tile_h;
        x0 = tx * tile_w;
    };
    // raw-frame source: item = (halo pixel p, 16-byte chunk c of its 16 channels); thread t owns items t, t + 256, ...
    constexpr int EPC = 16 / (int)sizeof(T);       // channels per chunk
    constexpr int NITEM = (NPL * SPX + HWV * 64 - 1) / (HWV * 64);
    // SRC == 2 (float frames, the reference's own input contract): element-wise loads, kept simple (not the bench path)
    float rawv[SRC == 2 ? NITEM : 1][SRC == 2 ? EPC : 1];
    // SRC == 1: dword d of the tile's raw block = (row r = colour * rrows + ry, dword cd of the row)
    const int rrows = 2 * (tile_h + 2), rdw = (2 * (tile_w + 2) + 4) / 4;     // raw rows per colour, dwords per row
    constexpr int NDW = (RAW_ROWS * (RAWP / 4) + HWV * 64 - 1) / (HWV * 64);   // upper bound of dwords per thread
    unsigned int rawd[SRC == 1 ? NDW : 1];
    auto load_raw = [&](int t) {
        int bimg, y0, x0;
        decode_tile(t, bimg, y0, x0);
        const int Hr = 2 * a.H, Wr = 2 * a.W;
        if constexpr (SRC == 1) {
            const int ntot = 3 * rrows * rdw;
            const int ry0 = 2 * (y0 - 1), rx0 = 2 * (x0 - 1) - 2;            // rx0 is a multiple of 4 (tile_w even)
#pragma unroll
            for (int k = 0; k < NDW; ++k) {
                const int d = tid + k * (HWV * 64);
                unsigned int v = 0u;
                if (d < ntot) {
                    const int r = d / rdw, cd = d - r * rdw;
                    const int col = r / rrows, ry = r - col * rrows;
                    const int y = ry0 + ry, x = rx0 + 4 * cd;
                    if ((unsigned)y < (unsigned)Hr && x >= 0 && x + 3 < Wr)
                        v = *reinterpret_cast<const unsigned int*>(reinterpret_cast<const unsigned char*>(a.in) + (((long)bimg * 3 + col) * Hr + y) * Wr + x);
                }
                rawd[k] = v;
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < NITEM; ++k) {
            const int item = tid + k * (HWV * 64);
            const int c = item / SPX, p = item - c * SPX;
            const int hy = SQ ? (p * 3641) >> 16 : (int)(((unsigned)p * a.magic_h) >> 16), hx = p - hy * hpw;
            const int sy = y0 - 1 + hy, sx = x0 - 1 + hx;
            const bool ok = c < NPL && p < hpix && (unsigned)sy < (unsigned)a.H && (unsigned)sx < (unsigned)a.W;
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const int ch = c * EPC + e;           // channel of the space-to-depth map: patch * 3 + colour
                const int patch = ch / 3, col = ch - patch * 3;
                float v = 0.0f;
                if (ok && ch < 12) v = reinterpret_cast<const float*>(a.in)[(((long)bimg * 3 + col) * Hr + 2 * sy + (patch & 1)) * Wr + 2 * sx + (patch >> 1)];
                rawv[SRC == 2 ? k : 0][SRC == 2 ? e : 0] = v;
            }
        }
    };
    auto store_raw = [&](int buf) {                 // registers -> LDS planes of halo buffer `buf`
        if constexpr (SRC == 1) {
            const int ntot = 3 * rrows * rdw;
#pragma unroll
            for (int k = 0; k < NDW; ++k) {
                const int d = tid + k * (HWV * 64);
                if (d < ntot) {
                    const int r = d / rdw, cd = d - r * rdw;
                    *reinterpret_cast<unsigned int*>(rawl + r * RAWP + cd * 4) = rawd[k];
                }
            }
            __syncthreads();                        // the raw block is complete
#pragma unroll
            for (int k = 0; k < NITEM; ++k) {
                const int item = tid + k * (HWV * 64);
                const int c = item / SPX, p = item - c * SPX;
                if (c < NPL && p < hpix) {
                    const int hy = SQ ? (p * 3641) >> 16 : (int)(((unsigned)p * a.magic_h) >> 16), hx = p - hy * hpw;
                    T vals[EPC];
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const int ch = c * EPC + e;
                        const int patch = ch / 3, col = ch - patch * 3;
                        vals[e] = ch < 12 ? lut[rawl[(col * rrows + 2 * hy + (patch & 1)) * RAWP + 2 * hx + (patch >> 1) + 2]] : (T)0.0f;
                    }
                    *reinterpret_cast<u32x4_t*>(hlds + buf * HB + c * SPL + p * 16) = *reinterpret_cast<const u32x4_t*>(vals);
                }
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < NITEM; ++k) {
            const int item = tid + k * (HWV * 64);
            const int c = item / SPX, p = item - c * SPX;
            if (c < NPL) {
                u32x4_t o;
                if (sizeof(T) == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = pack_bf16x2(rawv[SRC == 2 ? k : 0][SRC == 2 ? (2 * e) % EPC : 0], rawv[SRC == 2 ? k : 0][SRC == 2 ? (2 * e + 1) % EPC : 0]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = __float_as_uint(rawv[SRC == 2 ? k : 0][SRC == 2 ? e % EPC : 0]);
                }
                *reinterpret_cast<u32x4_t*>(hlds + buf * HB + c * SPL + p * 16) = o;
            }
        }
    };
    auto issue_halo = [&](int t, int buf) {
        if (SRC) { load_raw(t); return; }
        int bimg, y0, x0;
        decode_tile(t, bimg, y0, x0);
        const int base = ((bimg * a.H + (y0 - 1)) * a.W + (x0 - 1)) * pix_b;
#pragma unroll
        for (int j = 0; j < NPIECE; ++j) {
            const int q = wave + HWV * j;           // piece: plane q / 6, 64-pixel block q % 6
            const int c = q / 6, b = q - c * 6;
            const int p = b * 64 + lane;
            const int hy = SQ ? (p * 3641) >> 16 : (int)(((unsigned)p * a.magic_h) >> 16), hx = p - hy * hpw;
            const bool ok = p < hpix && (unsigned)(y0 - 1 + hy) < (unsigned)a.H && (unsigned)(x0 - 1 + hx) < (unsigned)a.W;
            lds_dma16(irsrc, hlds + buf * HB + c * SPL + b * 1024, ok ? base + (hy * a.W + hx) * pix_b + c * 16 : -1, 0);
        }
    };

    f32x4_t acc[NF][4];
    acc_start<float, NF>(acc, lbias, fq);                          // (zeros)

    issue_halo(tile, 0);
    if (SRC) store_raw(0);
    int it = 0;
    for (;;) {
        wait_vmcnt0();                 // this wave's halo pieces have landed (and the previous tile's stores)
        __syncthreads();               // everybody's pieces have landed; everybody is done reading the other buffer (and lbias is visible)
        if constexpr (BiasInAcc<T>::value) acc_start<T, NF>(acc, lbias, fq);      // this tile's accumulators start from the bias
        const int next = tile + tstep;
        if (next < tend) issue_halo(next, (it + 1) & 1);      // SRC: the raw loads fly under the MFMAs; converted below
        const char* hb = hlds + (it & 1) * HB;
        if constexpr (sizeof(T) == 1) {
            // fp8: K-steps in pairs through the 16x16x128 instruction; an odd last K-step through the 16x16x32 pair
#pragma unroll
            for (int kp = 0; kp < NKS / 2; ++kp) {
                u32x4_t pf[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        pf[h][i] = *reinterpret_cast<const u32x4_t*>(hb + (SQ ? pbi[0] + i * (18 * 16) : pbi[i]) + toff[2 * kp + h]);
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    u32x4_t wf[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int ks = 2 * kp + h;
                        wf[h] = *reinterpret_cast<const u32x4_t*>(wfrag + (ks >> 2) * WBUF + ((j >> 1) * 32 + (j & 1) * 4) * 256 + ((((ks & 3) * 4 + fq) ^ wsw0) << 4));
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) fp8_mma128(wf[0], wf[1], pf[0][i], pf[1][i], acc[j][i]);
                }
            }
        }
#pragma unroll
        for (int ks = (sizeof(T) == 1 ? NKS / 2 * 2 : 0); ks < NKS; ++ks) {
            u32x4_t pf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) pf[i] = *reinterpret_cast<const u32x4_t*>(hb + (SQ ? pbi[0] + i * (18 * 16) : pbi[i]) + toff[ks]);
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const u32x4_t wf = *reinterpret_cast<const u32x4_t*>(wfrag + (ks >> 2) * WBUF + ((j >> 1) * 32 + (j & 1) * 4) * 256 +
                                                                      ((((ks & 3) * 4 + fq) ^ wsw0) << 4));
#pragma unroll
                for (int i = 0; i < 4; ++i) S1<T>::mma(wf, pf[i], acc[j][i]);
            }
        }
        int bimg, y0, x0;
        decode_tile(tile, bimg, y0, x0);
        if (a.act == ACT_SILU) tile_epilogue<T, NF, ACT_SILU, SQ, 0, TO>(a, acc, lbias, orsrc, rrsrc, bimg, y0, x0, wave * 64 + fr, n0 + fq * 8);
        else if (a.act == ACT_RELU) tile_epilogue<T, NF, ACT_RELU, SQ, 0, TO>(a, acc, lbias, orsrc, rrsrc, bimg, y0, x0, wave * 64 + fr, n0 + fq * 8);
        else tile_epilogue<T, NF, ACT_NONE, SQ, 0, TO>(a, acc, lbias, orsrc, rrsrc, bimg, y0, x0, wave * 64 + fr, n0 + fq * 8);
        if (next >= tend) break;
        if (SRC) store_raw((it + 1) & 1);    // the other buffer: nobody reads it before the barrier at the loop top
        tile = next;
        ++it;
    }
}

// ---- narrow inputs, stride 2 (64 bytes of channels per pixel: the 32 -> 64 convolution after the stem) ----
// The stride-2 scheme of conv_halo_kernel on the narrow kernel's machinery: the four parity phases of the input are
// staged one after the other as unit-stride cell tiles into the two halo buffers in turn (phase p + 1 is fetched while the
// 4 / 2 / 2 / 1 taps of phase p compute), all of K resident in LDS, one barrier per phase, epilogue after the fourth.
template <typename T, int NF, bool SQ, typename TO = T>
__global__ void __launch_bounds__(HWV * 64) conv_halo_small_s2_kernel(const ConvArgs a)
{
    constexpr int NB = NF * 16, NPL = 4, HB = NPL * SPL, NSLAB = 3, WBUF = NB * 256, NPIECE = NPL * 6 / HWV;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const wlds = smem;
    char* const hlds = smem + NSLAB * WBUF;
    float* const lbias = reinterpret_cast<float*>(smem + NSLAB * WBUF + 2 * HB);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (wave-uniform: scalar LDS-DMA targets, no v_readfirstlane -> m0 chain per piece)
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.y * NB;
    const int tile_w = SQ ? 16 : a.tile_w, tile_h = SQ ? 16 : a.tile_h;
    const int tiles_x = (a.Wo + tile_w - 1) / tile_w, tiles_y = (a.Ho + tile_h - 1) / tile_h;
    const int hpw = tile_w + 2, hpix = hpw * (tile_h + 2);
    const int ntile = a.B * tiles_y * tiles_x;
    const int pix_b = a.ldi * (int)sizeof(T);
    int tile, tstep, tend;                                   // XCD-aware tile order (conv_frag.h: tile_walk)
    tile_walk(ntile, tile, tstep, tend);
    if (tile >= tend) return;
    {
        const char* wsrc = reinterpret_cast<const char*>(a.w);
        const long wpitch = (long)a.Kpad * (long)sizeof(T);
        for (int idx = tid; idx < NSLAB * NB * 16; idx += HWV * 64) {
            const int ss = idx / (NB * 16), rc = idx - ss * (NB * 16);
            const int row = rc >> 4, c = rc & 15;
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(wsrc + (long)(n0 + row) * wpitch + ss * 256 + c * 16);
            const int swz = (row & 3) | (((row >> 3) & 3) << 2);
            *reinterpret_cast<u32x4_t*>(wlds + ss * WBUF + row * 256 + ((c ^ swz) << 4)) = v;
        }
        for (int i = tid; i < NB; i += HWV * 64) lbias[i] = a.bias[n0 + i];
        if (sizeof(T) == 1)
            for (int i = tid; i < NB; i += HWV * 64) lbias[NB + i] = a.mult ? a.mult[n0 + i] : 1.0f;
    }
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, (int)a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)a.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res ? a.res : a.out), 0, (int)(a.res ? a.res_bytes : a.out_bytes), 0x00020000);
    int pbi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int ty, tx;
        tile_pixel<SQ>(a, wave * 64 + i * 16 + fr, ty, tx);
        pbi[i] = fq * SPL + (ty < 0 ? 0 : ty * hpw + tx) * 16;      // K-group fq = 16-byte chunk fq of the pixel's 64 bytes
    }
    const int wrow0 = (fr >> 2) * 8 + (fr & 3);
    const int wsw0 = (wrow0 & 3) | (((wrow0 >> 3) & 3) << 2);
    const char* const wfrag = wlds + wrow0 * 256;
    auto decode_tile = [&](int t, int& bimg, int& y0, int& x0) {
        const int tx = t % tiles_x;
        const int q = t / tiles_x;
        bimg = q / tiles_y;
        y0 = (q - bimg * tiles_y) * tile_h;
        x0 = tx * tile_w;
    };
    // phase ph (order of tap_step<true>: odd/odd, odd rows/even cols, even rows/odd cols, even/even) of tile t
    auto issue_halo = [&](int t, int ph, int buf) {
        int bimg, y0, x0;
        decode_tile(t, bimg, y0, x0);
        const int dy = ph < 2, dx = (ph & 1) == 0;
        const int ry0 = 2 * (y0 - 1) + dy, rx0 = 2 * (x0 - 1) + dx;
        const int base = ((bimg * a.H + ry0) * a.W + rx0) * pix_b;
#pragma unroll
        for (int j = 0; j < NPIECE; ++j) {
            const int q = wave + HWV * j;
            const int c = q / 6, b = q - c * 6;
            const int p = b * 64 + lane;
            const int hy = SQ ? (p * 3641) >> 16 : (int)(((unsigned)p * a.magic_h) >> 16), hx = p - hy * hpw;
            const bool ok = p < hpix && (unsigned)(ry0 + 2 * hy) < (unsigned)a.H && (unsigned)(rx0 + 2 * hx) < (unsigned)a.W;
            lds_dma16(irsrc, hlds + buf * HB + c * SPL + b * 1024, ok ? base + (2 * hy * a.W + 2 * hx) * pix_b + c * 16 : -1, 0);
        }
    };
    f32x4_t acc[NF][4];
    acc_start<float, NF>(acc, lbias, fq);                          // (zeros)

    issue_halo(tile, 0, 0);
    int u = 0;                                    // (tile, phase) counter: halo buffer u & 1
    for (;;) {
        const int next = tile + tstep;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            wait_vmcnt0();
            __syncthreads();
            if constexpr (BiasInAcc<T>::value) {
                if (ph == 0) acc_start<T, NF>(acc, lbias, fq);   // this tile's accumulators start from the bias (lbias is visible behind the barrier)
            }
            if (ph < 3) issue_halo(tile, ph + 1, (u + 1) & 1);
            else if (next < tend) issue_halo(next, 0, (u + 1) & 1);
            const char* hb = hlds + (u & 1) * HB;
            constexpr int Q0[4] = {0, 4, 6, 8}, QN[4] = {4, 2, 2, 1};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < QN[ph]) {
                    const int tap = tap_step<true>(Q0[ph] + k).tap;           // compile-time after unrolling
                    const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
                    const int toff = ((ky != 0) * hpw + (kx != 0)) * 16;
                    u32x4_t pf[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) pf[i] = *reinterpret_cast<const u32x4_t*>(hb + (SQ ? pbi[0] + i * (18 * 16) : pbi[i]) + toff);
#pragma unroll
                    for (int j = 0; j < NF; ++j) {
                        const u32x4_t wf = *reinterpret_cast<const u32x4_t*>(wfrag + (tap >> 2) * WBUF + ((j >> 1) * 32 + (j & 1) * 4) * 256 +
                                                                              ((((tap & 3) * 4 + fq) ^ wsw0) << 4));
#pragma unroll
                        for (int i = 0; i < 4; ++i) S1<T>::mma(wf, pf[i], acc[j][i]);
                    }
                }
            }
            ++u;
        }
        int bimg, y0, x0;
        decode_tile(tile, bimg, y0, x0);
        if (a.act == ACT_SILU) tile_epilogue<T, NF, ACT_SILU, SQ, 0, TO>(a, acc, lbias, orsrc, rrsrc, bimg, y0, x0, wave * 64 + fr, n0 + fq * 8);
        else if (a.act == ACT_RELU) tile_epilogue<T, NF, ACT_RELU, SQ, 0, TO>(a, acc, lbias, orsrc, rrsrc, bimg, y0, x0, wave * 64 + fr, n0 + fq * 8);
        else tile_epilogue<T, NF, ACT_NONE, SQ, 0, TO>(a, acc, lbias, orsrc, rrsrc, bimg, y0, x0, wave * 64 + fr, n0 + fq * 8);
        if (next >= tend) break;
        tile = next;
    }
}

// ------------------------------------------------------------------------------------------------ host
// Tile shape: th x tw <= 256 output pixels whose (th+2) x (tw+2) halo fits `slots` pixel slots, chosen to waste the
// least matrix work on this image size (16 x 16 when the sides divide; a 40-wide map gets 6 x 40).  Widths that are
// not multiples of 16 wrap MFMA fragments across tile rows: a few 2-way LDS bank conflicts, still far cheaper than
// idle lanes.  Returns the covered fraction of the tile grid.
static double pick_tile(ConvArgs& a, int slots)
{
    double best = -1.0;
    for (int tw = 8; tw <= 64; ++tw) {
        int th = 256 / tw;
        while (th > 1 && (th + 2) * (tw + 2) > slots) --th;
        if ((th + 2) * (tw + 2) > slots) continue;
        const double cover = (double)a.Ho * a.Wo / ((double)((a.Ho + th - 1) / th) * ((a.Wo + tw - 1) / tw) * 256.0);
        const double score = cover - (tw % 16 ? 0.03 : 0.0);
        if (score > best + 1e-9) { best = score; a.tile_w = tw; a.tile_h = th; }
    }
    a.magic_w = 65536u / (unsigned)a.tile_w + 1u;
    a.magic_h = 65536u / (unsigned)(a.tile_w + 2) + 1u;
    return (double)a.Ho * a.Wo / ((double)((a.Ho + a.tile_h - 1) / a.tile_h) * ((a.Wo + a.tile_w - 1) / a.tile_w) * 256.0);
}

template <typename T, int NF, bool SQ, bool S2, int FC = 0, typename TO = T, bool CV1 = false>
static hipError_t halo_launch(const ConvArgs& a0, hipStream_t s, int n_cu)
{
    ConvArgs a = a0;
    constexpr int NB = NF * 16;
#ifdef SKY_EXPERIMENTS
    const char* dbg = getenv("SKY_CONV_DBG");
    a.dbg = dbg ? atoi(dbg) : 0;
#else
    a.dbg = 0;
#endif
    size_t lds = HALO_BYTES + (CV1 ? 3 : 2) * NB * 128 + (sizeof(T) == 1 ? 2 : 1) * NB * 4 + (FC ? FC * FC * sizeof(T) + FC * 4 : 0) + (CV1 ? 512 + NB * 4 : 0);
#ifdef SKY_EXPERIMENTS
    lds += ((a.dbg & 256) ? 512 : 0) + ((size_t)((a.dbg >> 16) & 0xff) << 10);
#endif
    // OPT_NF8_SOLO (A/B): 128-channel tiles alone on a CU -- an LDS request above half of the CU's 160 KB excludes a second one
    const bool solo = NF == 8 && (a.opts & OPT_NF8_SOLO);
    if (solo && lds < 84 * 1024) lds = 84 * 1024;
    auto kern = conv_halo_kernel<T, NF, SQ, S2, FC, TO, CV1>;
    static size_t attr[16] = {0};
    {   // the attribute is set to the largest size this kernel can ever ask for (solo / padded variants included)
        const hipError_t e = ensure_lds_attr(reinterpret_cast<const void*>(kern), lds > 84 * 1024 ? lds : 84 * 1024, a.device, attr);
        if (e != hipSuccess) return e;
    }
    const int ntile = a.B * ((a.Ho + a.tile_h - 1) / a.tile_h) * ((a.Wo + a.tile_w - 1) / a.tile_w);
    // two workgroups per CU in total: with several N tiles the tile range is split over fewer, longer-lived workgroups
    const int gy = a.Cout / NB;
    const int per_cu = solo ? 1 : 2;
    const int slots = (a.opts & OPT_OLDGRID) ? per_cu * n_cu : (per_cu * n_cu / gy > 0 ? per_cu * n_cu / gy : 1);
    int gx = ntile < slots ? ntile : slots;
#ifdef SKY_EXPERIMENTS
    if (a.dbg & 64) {
        int nb = -1;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), HWV * 64, lds) != hipSuccess) nb = -1;
        fprintf(stderr, "conv_halo<NF=%d>: %d workgroups per CU with %zu B of LDS\n", NF, nb, lds);
    }
    if ((a.dbg & 128) && gx > n_cu) gx = n_cu;
    static unsigned long long* stamps = nullptr;
    if (a.dbg & 256) {
        if (!stamps && hipMalloc(&stamps, 1024 * 64 * 8) != hipSuccess) return hipErrorOutOfMemory;
        if (hipMemsetAsync(stamps, 0, 1024 * 64 * 8, s) != hipSuccess) return hipErrorUnknown;
        a.raw = reinterpret_cast<float*>(stamps);
    }
#endif
    kern<<<dim3(gx, a.Cout / NB), dim3(HWV * 64), lds, s>>>(a);
#ifdef SKY_EXPERIMENTS
    if (a.dbg & 256) {   // experiment: mean timeline of every workgroup's second tile (shader clocks since its start)
        static int once = 0;
        if (once++ == 2 && hipStreamSynchronize(s) == hipSuccess) {
            static unsigned long long h[1024 * 64];
            if (hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
                double sum[44] = {0};
                int cnt = 0;
                for (int w = 0; w < gx; ++w) {
                    if (!h[w * 64] || !h[w * 64 + 42]) continue;
                    ++cnt;
                    for (int k = 0; k < 44; ++k) sum[k] += h[w * 64 + k] ? (double)(h[w * 64 + k] - h[w * 64]) : 0.0;
                }
                fprintf(stderr, "halo timeline over %d workgroups (NF=%d S2=%d FC=%d), mean shader clocks since tile start:\n", cnt, NF, (int)S2, FC);
                for (int k = 0; k < 44; ++k)
                    if (sum[k] > 0) fprintf(stderr, "  [%2d] %9.0f\n", k, sum[k] / (cnt ? cnt : 1));
                double rt = 0;                 // shader clock held during the tile: cycles / (100 MHz ticks) x 100 MHz
                for (int w = 0; w < gx; ++w)
                    if (h[w * 64] && h[w * 64 + 42] && h[w * 64 + 61] > h[w * 64 + 60]) rt += (double)(h[w * 64 + 61] - h[w * 64 + 60]);
                if (rt > 0) fprintf(stderr, "  shader clock over the tile: %.0f MHz\n", sum[42] / rt * 100.0);
                double c0 = 0, r0 = 0;
                int n0 = 0;
                for (int w = 0; w < gx; ++w)
                    if (h[w * 64 + 58] > h[w * 64 + 56] && h[w * 64 + 59] > h[w * 64 + 57]) { c0 += (double)(h[w * 64 + 58] - h[w * 64 + 56]); r0 += (double)(h[w * 64 + 59] - h[w * 64 + 57]); ++n0; }
                if (n0) fprintf(stderr, "  FIRST tile (every CU holds two workgroups): %.0f cycles, %.2f us, shader clock %.0f MHz\n", c0 / n0, r0 / n0 / 100.0, c0 / r0 * 100.0);
            }
        }
    }
#endif
    return hipGetLastError();
}

template <typename T, int CB, int NF, bool SQ, int SRC = 0, typename TO = T>
static hipError_t halo_small_launch(const ConvArgs& a, hipStream_t s, int n_cu)
{
    constexpr int NB = NF * 16, NSLAB = (9 * CB + 255) / 256;
    const size_t lds = (size_t)NSLAB * NB * 256 + 2 * (CB / 16) * SPL + (sizeof(T) == 1 ? 2 : 1) * NB * 4 + (SRC == 1 ? 3 * 36 * 40 + 256 * sizeof(T) : 0);
    auto kern = conv_halo_small_kernel<T, CB, NF, SQ, SRC, TO>;
    static size_t attr[16] = {0};
    {
        const hipError_t e = ensure_lds_attr(reinterpret_cast<const void*>(kern), lds, a.device, attr);
        if (e != hipSuccess) return e;
    }
    const int per_cu = (int)(160 * 1024 / lds) < 4 ? (int)(160 * 1024 / lds) : 4;
    const int ntile = a.B * ((a.H + a.tile_h - 1) / a.tile_h) * ((a.W + a.tile_w - 1) / a.tile_w);
    const int gy = a.Cout / NB;
    const int slots = per_cu * n_cu / gy > 0 ? per_cu * n_cu / gy : 1;
    const int gx = ntile < slots ? ntile : slots;
    kern<<<dim3(gx, a.Cout / NB), dim3(HWV * 64), lds, s>>>(a);
    return hipGetLastError();
}

template <typename T, int NF, bool SQ, typename TO = T>
static hipError_t halo_small_s2_launch(const ConvArgs& a, hipStream_t s, int n_cu)
{
    constexpr int NB = NF * 16;
    const size_t lds = (size_t)3 * NB * 256 + 2 * 4 * SPL + (sizeof(T) == 1 ? 2 : 1) * NB * 4;
    auto kern = conv_halo_small_s2_kernel<T, NF, SQ, TO>;
    static size_t attr[16] = {0};
    {
        const hipError_t e = ensure_lds_attr(reinterpret_cast<const void*>(kern), lds, a.device, attr);
        if (e != hipSuccess) return e;
    }
    const int per_cu = (int)(160 * 1024 / lds) < 4 ? (int)(160 * 1024 / lds) : 4;
    const int ntile = a.B * ((a.Ho + a.tile_h - 1) / a.tile_h) * ((a.Wo + a.tile_w - 1) / a.tile_w);
    const int gy = a.Cout / NB;
    const int slots = per_cu * n_cu / gy > 0 ? per_cu * n_cu / gy : 1;
    const int gx = ntile < slots ? ntile : slots;
    kern<<<dim3(gx, gy), dim3(HWV * 64), lds, s>>>(a);
    return hipGetLastError();
}

template <typename T, typename TO>
static hipError_t halo_small_dispatch(int cb, int nb, const ConvArgs& a, hipStream_t s, int n_cu)
{
    const bool sq = a.tile_w == 16 && a.tile_h == 16;
    if (a.src_mode) {   // raw frames (FocusBlock import fused): 16 channels = 32 B (bf16) / 64 B (fp32); source element type by mode
        if constexpr (sizeof(T) >= 2) {
            constexpr int CBR = 16 * (int)sizeof(T);
            if (cb != CBR) return hipErrorNotSupported;
#define SKY_RAW(NFF, SQQ) (a.src_mode == 1 ? halo_small_launch<T, CBR, NFF, SQQ, 1, TO>(a, s, n_cu) : halo_small_launch<T, CBR, NFF, SQQ, 2, TO>(a, s, n_cu))
            if (nb == 32) return sq ? SKY_RAW(2, true) : SKY_RAW(2, false);
            return sq ? SKY_RAW(4, true) : SKY_RAW(4, false);
#undef SKY_RAW
        } else {
            return hipErrorNotSupported;      // the stem computes in bf16 also in the fp8 engine
        }
    }
    if (cb == 32) {
        if (nb == 32) return sq ? halo_small_launch<T, 32, 2, true, 0, TO>(a, s, n_cu) : halo_small_launch<T, 32, 2, false, 0, TO>(a, s, n_cu);
        return sq ? halo_small_launch<T, 32, 4, true, 0, TO>(a, s, n_cu) : halo_small_launch<T, 32, 4, false, 0, TO>(a, s, n_cu);
    }
    if (nb == 32) return sq ? halo_small_launch<T, 64, 2, true, 0, TO>(a, s, n_cu) : halo_small_launch<T, 64, 2, false, 0, TO>(a, s, n_cu);
    return sq ? halo_small_launch<T, 64, 4, true, 0, TO>(a, s, n_cu) : halo_small_launch<T, 64, 4, false, 0, TO>(a, s, n_cu);
}

// supported (compute type, output type) pairs of the halo kernels: equal types, and bf16 -> fp8 on the narrow kernels (the
// stem of the fp8 engine)
static bool halo_types_ok(int dtype, const ConvArgs& a, bool narrow)
{
    const int odt = a.out_dt < 0 ? dtype : a.out_dt;
    return odt == dtype || (dtype == 1 && odt == 2 && (narrow || a.stride == 2));     // bf16 -> fp8: narrow kernels, stride-2 tile kernel
}

// the checks that route a convolution to the narrow-input kernel (the only one with a raw-frame loader)
static bool halo_small_ok(int dtype, ConvArgs& a)
{
    const int esz = dtype_size(dtype);
    if (a.ks != 3 || a.stride != 1 || a.pad != 1 || a.head || a.out_f32 || a.up2) return false;
    if (!halo_types_ok(dtype, a, true)) return false;
    if ((!a.src_mode && a.in_bytes == 0) || a.out_bytes == 0 || (a.res && a.res_bytes == 0)) return false;
    const long cb = (long)a.Cin * esz;
    if (!((cb == 32 || cb == 64) && (a.Cout == 32 || a.Cout % 64 == 0))) return false;
    if ((long)a.Kpad * esz < 9L * a.Cin * esz || (long)a.Cout * a.Kpad * esz >= (1L << 31)) return false;
    if (a.opts & OPT_HALO_OFF) return false;
    const double cover = pick_tile(a, SPX);
    return (a.opts & OPT_HALO_FORCE) || cover >= 0.75;
}

bool conv_accepts_raw(int dtype, const ConvArgs& a0)
{
    if (a0.opts & OPT_NO_FUSED_IMPORT) return false;       // A/B switch
    if (dtype == 2) return false;
    ConvArgs a = a0;
    a.src_mode = 1;
    // the raw loader stages 16 x 16 tiles; dword loads want an even map width (raw width a multiple of 4)
    return a.Cin == 16 && !a.res && halo_small_ok(dtype, a) && a.tile_w == 16 && a.tile_h == 16 && a.W % 2 == 0;
}

// the fused cv1 + 3x3 form (conv_halo_kernel CV1): bf16, 64 -> 64 -> 64 channels (one 128-byte chunk, one 64-channel N tile)
static bool cv1_shape_ok(int dtype, ConvArgs& a)
{
    if (dtype == 2)                                          // fp8 engine: the 128- and 64-channel bottleneck kernels (k_bneck_w8.hip, k_bneck_w64f8.hip)
        return (a.Cin == 128 && a.Cout == 128 && bneck128w8_shape_ok(a)) || (a.Cin == 64 && a.Cout == 64 && bneck64w8_shape_ok(a));
    if (dtype != 1 || (a.out_dt >= 0 && a.out_dt != 1)) return false;
    if (a.ks != 3 || a.stride != 1 || a.pad != 1 || a.head || a.out_f32 || a.up2 || a.src_mode || a.f2_w) return false;
    if (a.Cin == 128 && a.Cout == 128) return bneck128w_shape_ok(a) || bneck128_shape_ok(a);      // the 128-channel bottleneck kernels (k_bneck_w.hip, k_bneck.hip)
    if (a.Cin != 64 || a.Cout != 64 || a.c1_Kpad < 64) return false;
    if (bneck64w_shape_ok(a)) return true;                   // three workgroups per CU on 8 x 16 tiles (k_bneck_w64.hip)
    if (a.in_bytes == 0 || a.out_bytes == 0) return false;
    if ((long)a.Kpad * 2 < 9L * 128 || (a.opts & (OPT_HALO_OFF | OPT_NO_FUSE_CV1))) return false;
    const double cover = pick_tile(a, HPIX);
    return (a.opts & OPT_HALO_FORCE) || cover >= 0.75;
}

bool conv_accepts_cv1(int dtype, const ConvArgs& a0)
{
    ConvArgs a = a0;
    return cv1_shape_ok(dtype, a);
}

// returns hipErrorNotSupported when the shape is not covered / not worth it (caller falls back to the streaming kernel)
hipError_t launch_conv_halo(int dtype, const ConvArgs& a0, hipStream_t s, int* variant, int* fused)
{
    ConvArgs a = a0;
    if (a.c1_w) {       // planned as a fused bottleneck: there is no other kernel for this op
        if (!cv1_shape_ok(dtype, a)) return hipErrorInvalidValue;
        if (dtype == 2) {
            const hipError_t e8 = a.Cin == 64 ? launch_bneck64w8(a, s) : launch_bneck128w8(a, s);
            if (e8 == hipSuccess && variant) *variant = a.Cin == 64 ? 7066 : 7257;
            return e8 == hipErrorNotSupported ? hipErrorInvalidValue : e8;
        }
        if (a.Cin == 128) {
            if (bneck128w_shape_ok(a)) {                      // two 4-wave workgroups per CU (round 4)
                const hipError_t ew = launch_bneck128w(a, s);
                if (ew == hipSuccess && variant) *variant = 7256;
                return ew;
            }
            const hipError_t e0 = launch_bneck128(a, s);
            if (e0 == hipSuccess && variant) *variant = 7128;
            return e0;
        }
        if (bneck64w_shape_ok(a)) {
            const hipError_t e6 = launch_bneck64w(a, s);
            if (e6 == hipSuccess && variant) *variant = 7065;
            return e6;
        }
        const int n_cu1 = a.n_cu > 0 ? a.n_cu : 256;
        const bool sq1 = a.tile_w == 16 && a.tile_h == 16;
        const hipError_t e1 = sq1 ? halo_launch<__bf16, 4, true, false, 0, __bf16, true>(a, s, n_cu1) : halo_launch<__bf16, 4, false, false, 0, __bf16, true>(a, s, n_cu1);
        if (e1 == hipSuccess && variant) *variant = 7064;
        return e1;
    }
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
    const int esz = dtype_size(dtype);
    const int odt = a.out_dt < 0 ? dtype : a.out_dt;
    if (a.ks != 3 || (a.stride != 1 && a.stride != 2) || a.pad != 1 || a.head || a.out_f32 || a.up2) return hipErrorNotSupported;
    if ((!a.src_mode && a.in_bytes == 0) || a.out_bytes == 0 || (a.res && a.res_bytes == 0)) return hipErrorNotSupported;
    const long cb = (long)a.Cin * esz;
    const bool small = a.stride == 1 && (cb == 32 || cb == 64) && (a.Cout == 32 || a.Cout % 64 == 0);
    const bool small_s2 = a.stride == 2 && cb == 64 && (a.Cout == 32 || a.Cout % 64 == 0) && !a.src_mode && !a.f2_w;
    if (!halo_types_ok(dtype, a, small || small_s2)) return hipErrorNotSupported;
    if (small_s2) {
        if ((long)a.Kpad * esz < 9L * a.Cin * esz || (long)a.Cout * a.Kpad * esz >= (1L << 31)) return hipErrorNotSupported;
        // measured slower than the streaming kernel on its one layer of skyeye_s (32 -> 64 @640 -> 320: 0.385 vs 0.336 ms, the
        // layer is HBM-bound and four DMA phases per tile cost more than they save): only used when forced (tests)
        if (!(a.opts & OPT_HALO_FORCE)) return hipErrorNotSupported;
        pick_tile(a, SPX);
        const bool sq2 = a.tile_w == 16 && a.tile_h == 16;
        hipError_t e2;
#define SKY_S2(T, TO) (a.Cout == 32 ? (sq2 ? halo_small_s2_launch<T, 2, true, TO>(a, s, n_cu) : halo_small_s2_launch<T, 2, false, TO>(a, s, n_cu)) \
                                    : (sq2 ? halo_small_s2_launch<T, 4, true, TO>(a, s, n_cu) : halo_small_s2_launch<T, 4, false, TO>(a, s, n_cu)))
        if (dtype == 0) e2 = SKY_S2(float, float);
        else if (dtype == 1 && odt == 1) e2 = SKY_S2(__bf16, __bf16);
        else if (dtype == 1) e2 = SKY_S2(__bf16, fp8_t);
        else e2 = SKY_S2(fp8_t, fp8_t);
#undef SKY_S2
        if (e2 == hipSuccess && variant) *variant = 6000 + (a.Cout == 32 ? 32 : 64);
        return e2;
    }
    if (!small && a.stride == 1 && conv3x3_deep_ok(dtype, a)) {      // wide inputs: the deep-pipelined kernel (k_conv3x3_deep.hip)
        const hipError_t ed = launch_conv3x3_deep(dtype, a, s);
        if (ed == hipSuccess && variant) *variant = 4600 + 128;
        return ed;
    }
    if (!small && (a.src_mode || cb % 128 != 0 || a.Cout % 64 != 0)) return hipErrorNotSupported;
    if ((long)a.Kpad * esz < 9L * a.Cin * esz) return hipErrorNotSupported;
    if ((long)a.Cout * a.Kpad * esz >= (1L << 31)) return hipErrorNotSupported;
    if (a.opts & OPT_HALO_OFF) return hipErrorNotSupported;
    {   // bisection switch: SKY_HALO_SKIP bit 0 = stride-1 kernel, 1 = stride-2 kernel, 2 = narrow kernels, 3 = 128-channel tiles, 4 = 64-channel tiles
        const int skip = (int)(a.opts >> OPT_SKIP_SHIFT) & 31;
        if (small ? (skip & 4) : (a.stride == 2 ? (skip & 2) : (skip & 1))) return hipErrorNotSupported;
        if (!small && ((a.Cout % 128 == 0) ? (skip & 8) : (skip & 16))) return hipErrorNotSupported;
    }
    if (!small && a.stride == 2) {      // A/B switch: SKY_HALO_S2=0 sends stride-2 layers to the streaming kernel
        if (a.opts & OPT_S2_OFF) return hipErrorNotSupported;
    }
    const double cover = pick_tile(a, small ? SPX : HPIX);
    // partially filled tiles waste matrix work: keep the streaming kernel when less than 3/4 of the tile grid is image
    if (!(a.opts & OPT_HALO_FORCE) && cover < 0.75) return hipErrorNotSupported;
    if (small) {
        const int nb = a.Cout == 32 ? 32 : 64;
        hipError_t e;
        if (dtype == 0) e = halo_small_dispatch<float, float>((int)cb, nb, a, s, n_cu);
        else if (dtype == 1 && odt == 1) e = halo_small_dispatch<__bf16, __bf16>((int)cb, nb, a, s, n_cu);
        else if (dtype == 1) e = halo_small_dispatch<__bf16, fp8_t>((int)cb, nb, a, s, n_cu);
        else e = halo_small_dispatch<fp8_t, fp8_t>((int)cb, nb, a, s, n_cu);
        if (e == hipSuccess && variant) *variant = 5000 + nb;
        return e;
    }
    // 128-channel tiles where Cout allows (two workgroups per CU).  A/B switches: SKY_HALO_NF8=off -> 64-channel tiles everywhere,
    // SKY_HALO_NF8=solo -> 128-channel tiles alone on a CU (3x3 128->128 @80x80: 74.5 us default, 82 us off, 96 us solo)
    // fp32 (exact mode): 64-channel tiles, whose second accumulator set (two-level summation, conv_halo_kernel) fits the register file
    const int nb = (a.Cout % 128 == 0 && !(a.opts & OPT_NF8_OFF) && dtype != 0) ? 128 : 64;
    const bool sq = a.tile_w == 16 && a.tile_h == 16;
    hipError_t e;
#define SKY_HALO(T, S2V) (nb == 128 ? (sq ? halo_launch<T, 8, true, S2V>(a, s, n_cu) : halo_launch<T, 8, false, S2V>(a, s, n_cu)) \
                                    : (sq ? halo_launch<T, 4, true, S2V>(a, s, n_cu) : halo_launch<T, 4, false, S2V>(a, s, n_cu)))
    if (a.stride == 2 && dtype == 1 && odt == 2) {      // the fp8 engine's stride-2 convolutions behind its bf16 neck outputs
        e = nb == 128 ? (sq ? halo_launch<__bf16, 8, true, true, 0, fp8_t>(a, s, n_cu) : halo_launch<__bf16, 8, false, true, 0, fp8_t>(a, s, n_cu))
                      : (sq ? halo_launch<__bf16, 4, true, true, 0, fp8_t>(a, s, n_cu) : halo_launch<__bf16, 4, false, true, 0, fp8_t>(a, s, n_cu));
    } else if (a.stride == 2) {
        e = dtype == 0 ? SKY_HALO(float, true) : dtype == 1 ? SKY_HALO(__bf16, true) : SKY_HALO(fp8_t, true);
    } else if (dtype != 2 && a.f2_w && nb == 64 && a.Cout == 64 && a.f2_cin == 64 && a.f2_cout == 64 && a.f2_koff == 0 && a.f2_out_bytes) {
        // the 1x1 convolution that follows (the next bottleneck's cv1) runs in this kernel's epilogue
        if (dtype == 0) e = sq ? halo_launch<float, 4, true, false, 64>(a, s, n_cu) : halo_launch<float, 4, false, false, 64>(a, s, n_cu);
        else e = sq ? halo_launch<__bf16, 4, true, false, 64>(a, s, n_cu) : halo_launch<__bf16, 4, false, false, 64>(a, s, n_cu);
        if (e == hipSuccess && fused) *fused = 1;
    } else {
        e = dtype == 0 ? SKY_HALO(float, false) : dtype == 1 ? SKY_HALO(__bf16, false) : SKY_HALO(fp8_t, false);
    }
#undef SKY_HALO
    if (e == hipSuccess && variant) *variant = (a.stride == 2 ? 6000 : 4000) + nb;
    return e;
}

}  // namespace sky
