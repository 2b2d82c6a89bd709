// BottleneckBlock(128, 128) as ONE kernel (bf16 engine): x + cv2_3x3(cv1_1x1(x))      reference blocks.py:69-90
// (the bottlenecks of every CSP block with 128 hidden channels: stage 3 and the 80 x 80 neck blocks of skyeye_s, stage 2 and the
// 160 x 160 neck block of skyeye_l).  Layer by layer a bottleneck is two launches -- the 1x1 on the streaming kernel, the 3x3 +
// residual on the halo-tile kernel -- that move the 128-channel map five times; the halo-tile kernel itself runs the "two-stage
// ring, vmcnt(0) + barrier per tap" structure that tops out at about a third of the matrix rate.
//
// Here ONE workgroup of 8 waves per CU owns a 16 x 16 output tile and everything between x and the output stays in LDS / registers:
//   * the 18 x 18 x 128-channel tile of x (tile + 1 pixel halo) arrives by LDS-DMA in the halo kernels' layout, one image per
//     128-byte channel chunk: [chunk][K-group plane f][pixel slot, 352][2 x 16 B] (k_conv_halo.hip; conflict-free B fragments
//     for any tap offset);
//   * cv1: u = SiLU(W1 x + b1) on the 21 pixel fragments of the halo tile (three per wave, all 128 output channels), written back
//     over x IN PLACE (zeros outside the image: the 3x3's padding applies to u); the residual vectors of a wave's own output
//     pixels are read from the x tile first and stay in registers;
//   * the 18 taps x chunks of the 3x3: wave (pixel group pg, channel half hc) owns tile rows 4 pg .. 4 pg + 3 x 64 output
//     channels (4 x 4 accumulator fragments);
//   * ALL weights -- the two 128-byte K chunks of W1, then the 18 (chunk, tap) slabs of W2, [128 rows][128 B] each -- stream
//     through a FOUR-stage LDS ring by LDS-DMA.  A step = one slab = 32 MFMAs per wave.  The ring runs continuously across
//     tiles, two slabs are always in flight across the step barrier: the waits are COUNTED (s_waitcnt vmcnt(N), never 0 in the
//     steady state) and the barrier is the raw s_barrier -- __syncthreads() would drain the DMA queue (cdna_hip_programming.md 5,
//     "Pipelining across barriers").  At the barrier of step s every wave's pieces of slab s + 1 have landed, so the first
//     fragments of the next step can be requested before its barrier;
//   * the next tile's x is requested after the last tap, its latency runs under the epilogue (bias, SiLU, + x, 16-byte stores).
// K order (chunk, tap, 64-byte K-step), every bf16 rounding point and the activation arithmetic are those of the two-launch form:
// the result is bit-identical to it (tests/test_gpu_bneck128.py).  +27 % work on cv1 (18 x 18 for 16 x 16 pixels).
#include "sky_kernels.h"

#include "conv_frag.h"

// Stage switches for timing experiments exist only in builds with -DSKY_EXPERIMENTS (make -C csrc exp; SKY_BK_DBG=<bits>): 1 no cv1
// activation, 2 no epilogue activation, 4 no tap MFMAs, 8 no cv1 MFMAs, 16 no x DMA, 32 no weight DMA, 64 no stores.  The shipped
// library has no code path that skips work.
#ifdef SKY_EXPERIMENTS
#include <stdio.h>
#include <stdlib.h>
#define BK_DBG(a) ((a).dbg)
#else
#define BK_DBG(a) 0
#endif

namespace sky {

namespace bk {
constexpr int NW = 8, NT = NW * 64;
constexpr int TS = 16, HWD = TS + 2, NHP = HWD * HWD;         // 324 halo pixels
constexpr int C = 128, NCH = 2;                               // channels, 128-byte chunks of them
constexpr int XPIX = 352, PL = XPIX * 32, CHB = 4 * PL;       // pixel slots per plane, bytes per plane (44 * 256), per chunk image
constexpr int XDMA = XPIX / 32;                               // DMA pieces per plane (11)
constexpr int TILE_BYTES = NCH * CHB;                         // 90 112
constexpr int SLAB = C * 128;                                 // one weight slab [128 rows][128 B]
constexpr int NST = 4;                                        // ring stages
constexpr int NFR = 21;                                       // halo pixel fragments that hold real pixels (slots 0 .. 335)
constexpr int NSTEP = NCH + 9 * NCH;                          // steps per tile: cv1 chunks, then (chunk, tap)
constexpr int LDS_BYTES = TILE_BYTES + NST * SLAB + 2 * C * 4;
static_assert(NSTEP % NST == 0, "the ring stage of a step must not depend on the tile");
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
static_assert(PL % 256 == 0 && CHB % 256 == 0, "planes a multiple of 256 B apart: conflict-free fragment reads");
}  // namespace bk

__device__ __forceinline__ void bk_dma16(__amdgpu_buffer_rsrc_t rsrc, char* dst, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
}
// counted wait for this wave's vector-memory queue (LDS-DMA pieces and stores count together, in issue order) + its LDS queue
// the raw barrier (no queue drain); the empty asm keeps the compiler from moving LDS accesses across it
__device__ __forceinline__ void bk_barrier()
{
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
template <int N>
__device__ __forceinline__ void bk_wait_vm()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}
// the count as a value (folds after unrolling: s_waitcnt takes an immediate)
__device__ __forceinline__ void bk_wait_n(int n)
{
    switch (n) {
    case 0: bk_wait_vm<0>(); break;
    case 2: bk_wait_vm<2>(); break;
    case 3: bk_wait_vm<3>(); break;
    case 4: bk_wait_vm<4>(); break;
    case 5: bk_wait_vm<5>(); break;
    case 6: bk_wait_vm<6>(); break;
    default: bk_wait_vm<0>(); break;
    }
}
__device__ __forceinline__ void bk_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// experiments (SKY_BK_DBG & 256): shader-clock stamps of the workgroup's second tile (waves 0 and 7), dumped through ConvArgs::raw
#ifdef SKY_EXPERIMENTS
#define BK_STAMP(slot) do { if ((a.dbg & 256) && nth == 1 && lane == 0 && (wave == 0 || wave == 7)) stamps[(wave ? 32 : 0) + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BK_STAMP(slot) do { } while (0)
#endif

__global__ void __launch_bounds__(bk::NT) bneck128_kernel(const ConvArgs a)
{
    using namespace bk;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const xt = smem;                                    // x tile, then u in place
    char* const ring = smem + TILE_BYTES;
    float* const lb1 = reinterpret_cast<float*>(ring + NST * SLAB);      // cv1 bias [128]
#ifdef SKY_EXPERIMENTS
    unsigned long long* const stamps = reinterpret_cast<unsigned long long*>(lb1 + 2 * C);      // 64 x 8 B behind the biases (experiments build only)
    int nth = 0;
#endif

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int hc = wave & 1, pg = wave >> 1;                  // channel half, pixel group (tile rows 4 pg .. 4 pg + 3)
    const int tiles_x = (a.W + TS - 1) / TS, tiles_y = (a.H + TS - 1) / TS;
    const int ntile = a.B * tiles_y * tiles_x;
    int tile, tstep, tend;                                   // XCD-aware tile order (conv_frag.h: tile_walk)
    tile_walk(ntile, tile, tstep, tend);
    if (tile >= tend) return;
    const int pix_b = a.ldi * 2;
    const int w1pitch = a.c1_Kpad * 2, w2pitch = a.Kpad * 2;

    float* const lb2 = lb1 + C;                                          // cv2 bias [128]
    for (int i = tid; i < C; i += NT) { lb1[i] = a.c1_bias[i]; lb2[i] = a.bias[i]; }

    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, (int)a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)a.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.c1_w), 0, (int)((long)C * w1pitch), 0x00020000);
    const __amdgpu_buffer_rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, (int)((long)C * w2pitch), 0x00020000);

    // weight DMA: a slab is 16 pieces of 8 rows; this wave issues pieces 2 wave, 2 wave + 1.  lane -> LDS row, stored chunk
    // lane & 7 = source chunk (lane & 7) ^ ((row >> 1) & 7); (fragment j, MFMA row r) -> channel (j>>1)*32 + (r>>2)*8 + (j&1)*4 + (r&3)
    int wrel1[2], wrel2[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = (wave * 2 + q) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int j = row >> 4, r = row & 15;
        const int ch = (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3);
        wrel2[q] = ch * w2pitch + c * 16;
        // W1 slabs are ROW halves, [64 rows][256 B] = all of K for 64 output channels (slab hh: fragments 4 hh .. 4 hh + 3), so that
        // the second half's MFMAs can run beside the first half's activation: a piece is 4 rows, lane -> row, stored chunk lane & 15 =
        // source chunk (lane & 15) ^ (row & 15)
        const int row1 = (wave * 2 + q) * 4 + (lane >> 4);
        const int c1 = (lane & 15) ^ (row1 & 15);
        const int j1 = row1 >> 4, r1 = row1 & 15;
        const int ch1 = (j1 >> 1) * 32 + (r1 >> 2) * 8 + (j1 & 1) * 4 + (r1 & 3);
        wrel1[q] = ch1 * w1pitch + c1 * 16;
    }
    // slab of in-tile step s (0, 1: W1 K chunks; 2 ..: W2 (chunk, tap)) into ring stage s & 3
    auto issue_slab = [&](int s) {
        char* const dst = ring + (s & (NST - 1)) * SLAB + wave * 2048;
        if (BK_DBG(a) & 32) return;
        if (s < NCH) {
#pragma unroll
            for (int q = 0; q < 2; ++q) bk_dma16(w1rsrc, dst + q * 1024, wrel1[q], s * 64 * w1pitch);
        } else {
            const int g = s - NCH, chunk = g / 9, tap = g - chunk * 9;
#pragma unroll
            for (int q = 0; q < 2; ++q) bk_dma16(w2rsrc, dst + q * 1024, wrel2[q], tap * (C * 2) + chunk * 128);
        }
    };
    auto decode_tile = [&](int t, int& bimg, int& y0, int& x0) {
        const int tx = t % tiles_x;
        const int q = t / tiles_x;
        bimg = q / tiles_y;
        y0 = (q - bimg * tiles_y) * TS;
        x0 = tx * TS;
    };
    // x tile DMA, one chunk image at a time: a wave fills pieces [b0, b1) of plane wave & 3 (11 pieces of 32 pixel slots per plane); in
    // piece b lane -> pixel slot p = b*32 + (lane >> 1), 16-byte half lane & 1 = K-step (lane & 1) ^ (p >> 3 & 1); outside the
    // image: offset -1 -> the range check writes zeros
    auto issue_x = [&](int bimg, int y0, int x0, int chunk, int b0, int b1) {
        if (BK_DBG(a) & 16) return;
        const int base = ((bimg * a.H + y0 - 1) * a.W + x0 - 1) * pix_b + chunk * 128 + (wave & 3) * 16;
        char* const dst = xt + chunk * CHB + (wave & 3) * PL;
        // (opaque: everything below that depends on the lane alone is a tile-loop invariant hipcc would keep in ~25 registers)
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int b = b0; b < b1; ++b) {
            const int p = b * 32 + (ln >> 1);
            const int hy = (p * 3641) >> 16, hx = p - hy * HWD;            // p / 18
            const int kk = (ln & 1) ^ ((p >> 3) & 1);
            const bool ok = p < NHP && (unsigned)(y0 - 1 + hy) < (unsigned)a.H && (unsigned)(x0 - 1 + hx) < (unsigned)a.W;
            bk_dma16(irsrc, dst + b * 1024, ok ? base + (hy * a.W + hx) * pix_b + kk * 64 : -1, 0);
        }
    };

    // fragment addresses
    const int arow = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);       // weight fragment: row fr of a fragment, K-step 0 (K-step 1: ^ 64)

    int bimg, y0, x0;
    decode_tile(tile, bimg, y0, x0);
    __builtin_amdgcn_s_waitcnt(0xC07F);                       // lgkmcnt(0): the bias writes above
    issue_slab(0);
    issue_slab(1);
    issue_slab(2);
    issue_x(bimg, y0, x0, wave >> 2, 0, XDMA);                // first tile: waves 0..3 chunk 0, waves 4..7 chunk 1
    bool first = true;

    for (;;) {
        const int next = tile + tstep;
        const bool has_next = next < tend;
        Out8<__bf16>::raw_t resv[4][2];                       // residual x of this lane's 4 x 2 output vectors
        BK_STAMP(0);
        u32x4_t wq01[2][2][2];                                // [step parity][channel pair][fragment]: weight pairs of K-step 0, requested a step ahead

        // ---------------- steps 0, 1: cv1 on this wave's halo fragments wave, wave + 8, wave + 16 ----------------
        // step 0: output channels 0 .. 63 (slab 0 = those rows of W1, all of K), step 1: channels 64 .. 127 with the activation of the
        // first half between its MFMAs, then the activation of the second half.  (All eight waves are in the same phase: arithmetic
        // that is not interleaved with matrix work of the same wave overlaps with nothing.)
        {
            int frq = fr, fqq = fq;                           // (opaque per tile, see issue_x)
            asm volatile("" : "+v"(frq), "+v"(fqq));
            // both chunks of x are needed from the start: behind the last x piece this wave issued six stores (epilogue)
            if (first) bk_wait_vm<0>(); else bk_wait_vm<6>();
            bk_barrier();
            BK_STAMP(1);
            issue_slab(3);
            // residual: piece (plane fq, K-step sp) of chunk hc of the centre pixel of this lane's 4 output pixels; nobody rewrites the
            // tile before the barrier of step 1
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pc = (4 * pg + i + 1) * HWD + 1 + frq;
#pragma unroll
                for (int sp = 0; sp < 2; ++sp)
                    resv[i][sp].a = *reinterpret_cast<const u32x4_t*>(xt + hc * CHB + fqq * PL + pc * 32 + ((sp ^ ((pc >> 3) & 1)) << 4));
            }
            u32x4_t xf[3][4];                                 // [fragment][64-byte K-step: chunk * 2 + kk]
            int pfr[3];
            bool inside[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int p = (wave + 8 * i) * 16 + frq;      // (the third fragment of waves 5..7 starts at slot 336+: clamp, never stored)
                const int pc = p < XPIX ? p : XPIX - 1;
                const int A = fqq * PL + pc * 32 + (((pc >> 3) & 1) << 4);
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    xf[i][2 * c] = *reinterpret_cast<const u32x4_t*>(xt + c * CHB + A);
                    xf[i][2 * c + 1] = *reinterpret_cast<const u32x4_t*>(xt + c * CHB + (A ^ 16));
                }
                const int hy = (p * 3641) >> 16, hx = p - hy * HWD;
                inside[i] = p < NHP && (unsigned)(y0 - 1 + hy) < (unsigned)a.H && (unsigned)(x0 - 1 + hx) < (unsigned)a.W;
                pfr[i] = fqq * PL + p * 32 + (((p >> 3) & 1) << 4);
            }
            const int a1row = frq * 256;                      // W1 fragment: row fr of a fragment; 16-byte chunk (K-step * 4 + fq) ^ fr
            auto w1frag = [&](int hh, int jj, int ks) -> u32x4_t {
                return *reinterpret_cast<const u32x4_t*>(ring + hh * SLAB + jj * 4096 + a1row + (((ks * 4 + fqq) ^ frq) << 4));
            };
            // 16 (fragment jj, K-step ks) groups of 3 MFMAs per half; the weight fragment of group g + 2 is read before the MFMAs of group g
            auto half_mma = [&](int hh, f32x4_t (&au)[4][3], auto&& between) {
                u32x4_t wq[3];
#pragma unroll
                for (int gq = 0; gq < 16 + 2; ++gq) {
                    if (gq < 16) wq[gq % 3] = w1frag(hh, gq >> 2, gq & 3);
                    if (gq >= 2) {
                        const int gg = gq - 2;
#pragma unroll
                        for (int i = 0; i < 3; ++i)
                            if (!(BK_DBG(a) & 8)) S1<__bf16>::mma(wq[gg % 3], xf[i][gg & 3], au[gg >> 2][i]);
                        between(gg);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            // u = SiLU(. + b1) -> bf16 -> back into the tile, in place (zeros outside the image: the 3x3's padding applies to u): 32-channel
            // group sg of fragment i = chunk sg >> 1, K-step sg & 1
            auto act_store = [&](const f32x4_t (&au)[4][3], int hh, int i, int sq) {
                if (wave + 8 * i >= NFR) return;              // (uniform) waves 5..7 own two real fragments; their third is computed and dropped
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = au[2 * sq][i][e];                     // (b1 was the accumulators' initial value)
                    v[4 + e] = au[2 * sq + 1][i][e];
                    if (!(BK_DBG(a) & 1)) { v[e] = S1<__bf16>::silu(v[e]); v[4 + e] = S1<__bf16>::silu(v[4 + e]); }
                }
                Out8<__bf16>::raw_t o = Out8<__bf16>::pack(v, 1.0f);
                if (!inside[i]) o.a = u32x4_t{0u, 0u, 0u, 0u};
                *reinterpret_cast<u32x4_t*>(xt + hh * CHB + (sq ? pfr[i] ^ 16 : pfr[i])) = o.a;
            };
            // the accumulators start from the bias (k_conv_halo.hip: acc_start): fragment jj of half hh = channels (2 hh + (jj >> 1)) * 32 + fq * 8 + (jj & 1) * 4 ..
            f32x4_t au0[4][3], au1[4][3];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4_t c0 = *reinterpret_cast<const f32x4_t*>(lb1 + (j >> 1) * 32 + fqq * 8 + (j & 1) * 4);
                const f32x4_t c1 = *reinterpret_cast<const f32x4_t*>(lb1 + (2 + (j >> 1)) * 32 + fqq * 8 + (j & 1) * 4);
#pragma unroll
                for (int i = 0; i < 3; ++i) { au0[j][i] = c0; au1[j][i] = c1; }
            }
            half_mma(0, au0, [](int) {});
            // step 1
            if (first) bk_wait_vm<2>(); else bk_wait_vm<8>();
            bk_barrier();
            BK_STAMP(2);
            issue_slab(4);
            // the six (fragment, 32-channel group) slices of the first half's activation go between the MFMA groups of the second half
            half_mma(1, au1, [&](int gg) {
                if (gg == 1) act_store(au0, 0, 0, 0);
                if (gg == 3) act_store(au0, 0, 0, 1);
                if (gg == 5) act_store(au0, 0, 1, 0);
                if (gg == 7) act_store(au0, 0, 1, 1);
                if (gg == 9) act_store(au0, 0, 2, 0);
                if (gg == 11) act_store(au0, 0, 2, 1);
            });
            BK_STAMP(22);
            // the first tap's K-step 0 weight pairs (slab 2 is complete and visible since the barrier of step 1): their latency runs
            // under the activation arithmetic below
#pragma unroll
            for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    wq01[NCH & 1][sp][h] = *reinterpret_cast<const u32x4_t*>(ring + (NCH & (NST - 1)) * SLAB + (4 * hc + 2 * sp + h) * 2048 + arow);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int sq = 0; sq < 2; ++sq) act_store(au1, 1, i, sq);
            BK_STAMP(23);
        }

        // ---------------- steps 2 .. 19: the 3x3 over u, (chunk, tap) by (chunk, tap) ----------------
        // A step = 4 groups (K-step kk, channel pair sp) of 8 MFMAs: two weight fragments x the four pixel fragments of K-step kk.
        // All eight waves run the steps in lockstep, so a step that read its fragments first and multiplied afterwards would leave
        // the matrix pipes idle for the whole LDS phase.  The fragments of the first two groups of step s + 1 (K-step 0 pixels, both
        // weight pairs of K-step 0) are therefore requested during the MFMAs of step s -- slab s + 1 is complete and visible since
        // the barrier of step s --, and the K-step 1 fragments of step s during its own first two groups.
        f32x4_t acc[4][4];                                    // start from cv2's bias: fragment j = channels 64 hc + (j >> 1) * 32 + fq * 8 + (j & 1) * 4 ..
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4_t c = *reinterpret_cast<const f32x4_t*>(lb2 + 64 * hc + (j >> 1) * 32 + fq * 8 + (j & 1) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = c;
        }
        u32x4_t pf0[2][4];                                    // [step parity]: pixel fragments of K-step 0
        u32x4_t pf1[4], wq2[2], wq3[2];                       // K-step 1 pixel fragments, weight pairs of K-step 1
        int pb[13];                                           // (per tile: 40 instructions, and the registers are free during cv1)
        {
            int frt = fr;
            asm volatile("" : "+v"(frt));
#pragma unroll
            for (int c = 0; c < 13; ++c) pb[c] = fq * PL + ((4 * pg) * HWD + frt) * 32 + (((frt + 8 * pg + c) & 8) << 1);
        }
        int nb = 0, ny0 = 0, nx0 = 0;
        if (has_next) decode_tile(next, nb, ny0, nx0);
        // pixel fragment of tile row 4 pg + r (r = i + ky: 0 .. 5), column shift kx: slot = (4 pg + r) * 18 + fr + kx, address = plane +
        // slot * 32 + 16 * bit 3 of the slot (K-step 0; K-step 1 is the other 16-byte half: ^ 16).  18 = 16 + 2, so that bit is bit 3
        // of fr + 8 pg + c with c = 2 r + kx in 0 .. 12: thirteen per-lane bases pb[c] (computed once per kernel) cover every tap of
        // every step and everything else is an immediate offset of the ds_read.
        // pixel fragment i of in-tile step st, K-step kk (the ^ 16 is applied to the per-lane base, the rest stays an immediate)
        auto tap_frag = [&](int st, int i, int kk) -> u32x4_t {
            const int g = st - NCH, chunk = g / 9, tap = g - chunk * 9;
            const int ky = tap / 3, kx = tap - ky * 3;
            int q = pb[2 * (i + ky) + kx];
            if (kk) {
                asm volatile("" : "+v"(q));                   // (opaque: else hipcc keeps the 13 ^ 16 variants of pb[] in registers as well)
                q ^= 16;
            }
            return *reinterpret_cast<const u32x4_t*>(xt + q + (chunk * CHB + ((i + ky) * HWD + kx) * 32));
        };
        auto wfrag = [&](int st, int kk, int sp, int h) -> u32x4_t {
            return *reinterpret_cast<const u32x4_t*>(ring + (st & (NST - 1)) * SLAB + (4 * hc + 2 * sp + h) * 2048 + (kk ? arow ^ 64 : arow));
        };
        // The DMA requests of a step (its weight slab, then the x pieces of the next tile) do not go out at the head of the step,
        // where all eight waves would issue theirs at once (measured ~150 cycles per piece there), and not at the same point in
        // both waves of a SIMD (waves w and w + 4): waves 4..7 issue behind the first MFMA group, waves 0..3 behind the second, so
        // that one wave's issue stalls run under the other's matrix work.
        auto step_dma = [&](int st) {
            constexpr int XS0d = NCH + 9;
            if (st + 3 < NSTEP) issue_slab(st + 3);
            else if (has_next) issue_slab(st + 3 - NSTEP);
            const int k = st - XS0d;
            const int nx = k < 0 ? 0 : (k * 2 + 2 <= XDMA ? 2 : (k * 2 < XDMA ? XDMA - k * 2 : 0));
            if (nx > 0 && has_next && wave < 4) issue_x(nb, ny0, nx0, 0, k * 2, k * 2 + nx);
        };
#pragma unroll
        for (int s = NCH; s < NSTEP; ++s) {
            const int p = s & 1;
            // Chunk 0 of u is dead after step XS0 - 1 (its nine taps are done): waves 0..3 (which own the planes of chunk 0) request the
            // NEXT tile's x chunk 0 into it, NX(s) pieces per step BEHIND that step's weight slab, so that the pieces are younger than
            // the slab the next wait is for.  Wait of step s: slab s + 1 (requested in step s - 2) must have landed; younger than it
            // are the x pieces of step s - 2, slab s + 2 and the x pieces of step s - 1.
            constexpr int XS0 = NCH + 9;
            auto NX = [](int st) { return st < XS0 ? 0 : (st - XS0) * 2 + 2 <= XDMA ? 2 : (st - XS0) * 2 < XDMA ? XDMA - (st - XS0) * 2 : 0; };
            if (!has_next) { if (s == NSTEP - 2) bk_wait_vm<0>(); else bk_wait_vm<2>(); }
            else if (wave < 4) bk_wait_n(NX(s - 2) + 2 + NX(s - 1));
            else bk_wait_vm<2>();
            bk_barrier();
            BK_STAMP(1 + s);
            if (s == NCH) {                                   // u was not complete before this barrier: no pixel prefetch for the first tap
#pragma unroll
                for (int i = 0; i < 4; ++i) pf0[p][i] = tap_frag(s, i, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) pf1[i] = tap_frag(s, i, 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) wq2[h] = wfrag(s, 1, 0, h);
            __builtin_amdgcn_sched_barrier(0);
            if (s == 6) BK_STAMP(28);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (!(BK_DBG(a) & 4)) S1<__bf16>::mma(wq01[p][0][h], pf0[p][i], acc[h][i]);
            if (s == 6) BK_STAMP(29);
            if (wave >= 4) step_dma(s);
#pragma unroll
            for (int h = 0; h < 2; ++h) wq3[h] = wfrag(s, 1, 1, h);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (!(BK_DBG(a) & 4)) S1<__bf16>::mma(wq01[p][1][h], pf0[p][i], acc[2 + h][i]);
            if (s == 6) BK_STAMP(30);
            if (wave < 4) step_dma(s);
            if (s + 1 < NSTEP) {                        // K-step 0 pixels of the next tap
#pragma unroll
                for (int i = 0; i < 4; ++i) pf0[p ^ 1][i] = tap_frag(s + 1, i, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (!(BK_DBG(a) & 4)) S1<__bf16>::mma(wq2[h], pf1[i], acc[h][i]);
            if (s == 6) BK_STAMP(31);
            if (s + 1 < NSTEP) {                        // K-step 0 weight pairs of the next slab
#pragma unroll
                for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                    for (int h = 0; h < 2; ++h) wq01[p ^ 1][sp][h] = wfrag(s + 1, 0, sp, h);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (!(BK_DBG(a) & 4)) S1<__bf16>::mma(wq3[h], pf1[i], acc[2 + h][i]);
            __builtin_amdgcn_sched_barrier(0);
        }

        // every wave is done with chunk 1 of u before the next tile's x chunk 1 lands on it (waves 4..7 own its planes)
        BK_STAMP(24);
        bk_barrier();
        BK_STAMP(25);
        BK_STAMP(26);
        // ---------------- epilogue: bias, SiLU, + x, bf16, 16-byte stores ----------------
        // The next tile's x chunk 1 (44 pieces: waves 0..3 six each, waves 4..7 five) goes out in front of the first two output
        // rows: every piece is followed by at least six of this wave's stores, which the counted wait of step 1 relies on.
        int fre = fr, fqe = fq;
        asm volatile("" : "+v"(fre), "+v"(fqe));
        const bool colok = x0 + fre < a.W;
        const int off0 = (((bimg * a.H + y0 + 4 * pg) * a.W + x0 + fre) * a.ldo + 64 * hc + 8 * fqe) * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (has_next && i < 2) {
                if (wave < 4) issue_x(nb, ny0, nx0, 1, i * 3, i * 3 + 3);
                else issue_x(nb, ny0, nx0, 1, 6 + i * 3, i == 0 ? 9 : XDMA);
            }
            const bool ok = colok && y0 + 4 * pg + i < a.H;
            // masked lanes: offset 0x80000000 stays out of range after the immediate is added (and the constant goes into the vector
            // offset / immediate, never into soffset: DESIGN.md section 3, store-data hazard)
            const int ooff = ok ? off0 + i * a.W * a.ldo * 2 : (int)0x80000000;
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                // this lane's channels of fragment pair sp: 64 hc + 32 sp + 8 fq .. + 7
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xx = e < 4 ? acc[2 * sp][i][e] : acc[2 * sp + 1][i][e - 4];      // (b2 was the accumulators' initial value)
                    v[e] = (BK_DBG(a) & 2) ? xx : S1<__bf16>::silu(xx);
                    if (a.c1_res) {
                        // multiply and residual add round separately, as in the 128-channel halo-tile kernel's epilogue (its residual
                        // add sits behind a branch, so hipcc does not contract it; the narrow kernels' epilogue is one fma)
#pragma clang fp contract(off)
                        const unsigned rw = resv[i][sp].a[e >> 1];
                        const float res = (e & 1) ? __uint_as_float(rw & 0xffff0000u) : __uint_as_float(rw << 16);
                        v[e] = v[e] + res;
                    }
                }
                if (!(BK_DBG(a) & 64)) Out8<__bf16>::store(Out8<__bf16>::pack(v, 1.0f), orsrc, ooff + sp * 64);
            }
        }
        BK_STAMP(27);
#ifdef SKY_EXPERIMENTS
        if ((a.dbg & 256) && nth == 1 && a.raw && lane == 0 && (wave == 0 || wave == 7)) {
            __builtin_amdgcn_s_waitcnt(0xC07F);
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.raw) + (size_t)blockIdx.x * 64 + (wave ? 32 : 0);
            for (int k = 0; k < 32; ++k) dst[k] = stamps[(wave ? 32 : 0) + k];
        }
        ++nth;
#endif
        if (!has_next) break;
        tile = next; bimg = nb; y0 = ny0; x0 = nx0;
        first = false;
    }
}

// plan-time question (c1_w may not be set yet): would this cv1 + 3x3 pair run on the kernel?
bool bneck128_shape_ok(const ConvArgs& a)
{
    const int th = (a.H + bk::TS - 1) / bk::TS, tw = (a.W + bk::TS - 1) / bk::TS;
    const double cover = (double)a.H * a.W / ((double)th * tw * 256.0);
    if (!(a.opts & OPT_HALO_FORCE) && cover < 0.75) return false;          // partially filled tiles waste matrix work
    return a.ks == 3 && a.stride == 1 && a.pad == 1 && a.Cin == bk::C && a.Cout == bk::C && a.c1_Kpad >= bk::C &&
           a.Kpad >= 9 * bk::C && a.ldi % 8 == 0 && a.ldo % 8 == 0 && a.in_bytes != 0 && a.H >= 1 && a.W >= 1 && a.act == ACT_SILU &&
           a.out_bytes != 0 && !a.head && !a.up2 && !a.out_f32 && !a.src_mode && !a.f2_w && !a.res && !(a.opts & (OPT_HALO_OFF | OPT_NO_FUSE_CV1 | OPT_NO_BNECK128));
}

hipError_t launch_bneck128(const ConvArgs& a0, hipStream_t s)
{
    if (!a0.c1_w || !bneck128_shape_ok(a0)) return hipErrorNotSupported;
    ConvArgs a = a0;
#ifdef SKY_EXPERIMENTS
    const char* dbg = getenv("SKY_BK_DBG");
    a.dbg = dbg ? atoi(dbg) : 0;
#else
    a.dbg = 0;
#endif
    static size_t attr[16] = {0};
    {
        const hipError_t e = ensure_lds_attr(reinterpret_cast<const void*>(bneck128_kernel), bk::LDS_BYTES, a.device, attr);
        if (e != hipSuccess) return e;
    }
    const int ntile = a.B * ((a.H + bk::TS - 1) / bk::TS) * ((a.W + bk::TS - 1) / bk::TS);
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
    const int gx = ntile < n_cu ? ntile : n_cu;
    size_t lds = bk::LDS_BYTES;
#ifdef SKY_EXPERIMENTS
    static unsigned long long* stamps = nullptr;
    if (a.dbg & 256) {
        lds += 512;
        if (!stamps && hipMalloc(&stamps, 1024 * 64 * 8) != hipSuccess) return hipErrorOutOfMemory;
        if (hipMemsetAsync(stamps, 0, 1024 * 64 * 8, s) != hipSuccess) return hipErrorUnknown;
        a.raw = reinterpret_cast<float*>(stamps);
        const hipError_t e2 = ensure_lds_attr(reinterpret_cast<const void*>(bneck128_kernel), lds, a.device, attr);
        if (e2 != hipSuccess) return e2;
    }
#endif
    hipLaunchKernelGGL(bneck128_kernel, dim3(gx), dim3(bk::NT), lds, s, a);
#ifdef SKY_EXPERIMENTS
    if (a.dbg & 256) {   // mean timeline of every workgroup's second tile (shader clocks since its start), waves 0 and 7
        static int once = 0;
        if (once++ == 3 && hipStreamSynchronize(s) == hipSuccess) {
            static unsigned long long h[1024 * 64];
            if (hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
                for (int wv = 0; wv < 2; ++wv) {
                    double sum[32] = {0};
                    int cnt = 0;
                    for (int w = 0; w < gx; ++w) {
                        const unsigned long long* r = h + w * 64 + wv * 32;
                        if (!r[0] || !r[27]) continue;
                        ++cnt;
                        for (int k = 0; k < 32; ++k) sum[k] += r[k] ? (double)(r[k] - r[0]) : 0.0;
                    }
                    fprintf(stderr, "bneck128 timeline, wave %d, %d workgroups, mean shader clocks since the tile's start:\n", wv ? 7 : 0, cnt);
                    for (int k = 0; k < 32; ++k) fprintf(stderr, "  [%2d] %9.0f\n", k, sum[k] / (cnt ? cnt : 1));
                }
            }
        }
    }
#endif
    return hipGetLastError();
}

}  // namespace sky
