// BottleneckBlock(128, 128) as ONE kernel, TWO workgroups per CU (round 4): x + cv2_3x3(cv1_1x1(x))      reference blocks.py:69-90
//
// k_bneck.hip holds one 8-wave workgroup per CU (157 KB of LDS): its eight waves move through cv1 (VALU-heavy), the 18 tap steps
// (matrix-heavy) and the epilogue (VALU-heavy) TOGETHER, so activation arithmetic never runs beside matrix work -- the counters of
// round 3 say VALU and MFMA co-execute in 5 % of the matrix-busy cycles.  Here a workgroup is FOUR waves and owns an 8 x 16 output
// tile with 80 896 bytes of LDS: two workgroups share a CU, run out of phase, and one's cv1 / epilogue runs under the other's taps.
//   * x tile: 10 x 18 halo pixels x 128 channels by LDS-DMA in the halo kernels' layout, one image per 128-byte channel chunk:
//     [chunk][K-group plane f][pixel slot, 184][2 x 16 B] (planes 23 x 256 B apart: conflict-free B fragments for any tap offset);
//     wave w fills plane w of both chunks (6 pieces of 32 slots per plane; the last piece is 24 slots: lanes 48..63 masked);
//   * cv1: u = SiLU(W1 x + b1) on the 12 pixel fragments of the halo tile (three per wave, all 128 output channels), written back
//     over x IN PLACE (zeros outside the image); the residual vectors of a wave's own output pixels are read from the x tile first;
//   * the 18 (chunk, tap) steps of the 3x3: wave (pixel group pg, channel half hc) owns tile rows 4 pg .. 4 pg + 3 x 64 output
//     channels (4 x 4 accumulator fragments) -- the wave tile of k_bneck.hip;
//   * weights: the two ROW halves of W1 ([64 rows][256 B]), then the 18 slabs of W2 ([128 rows][128 B]), 16 KB each, through a
//     TWO-stage ring by LDS-DMA: slab s + 1 is requested at the head of step s and waited for (vmcnt(0) + barrier) at its end --
//     the halo-tile kernel's structure; what hides the wait is the other workgroup, not a deeper ring;
//   * the next tile's x is requested behind the last tap, in front of the epilogue.
// K order (chunk, tap, 64-byte K-step), every bf16 rounding point and the activation arithmetic are those of the two-launch form and
// of k_bneck.hip: the three are bit-identical (tests/test_gpu_bneck128.py).  +41 % work on cv1 (10 x 18 for 8 x 16 pixels).
#include "sky_kernels.h"

#include "conv_frag.h"

// compile-time experiment bits (tools/bneckw_ab.sh; the shipped value is the default below): 1 s_setprio 1 over the tap steps,
// 2 counted wait at the head of a tile (the epilogue's 16 stores stay in flight), 4 the next tile's x chunk 0 requested behind the taps
// of chunk 0.  TIMING-ONLY bits (wrong results; what a wait costs): 8 the weight slabs are requested for the first tile only, 16 no s_barrier at the
// head of a tap step (the waits stay)
#ifndef BW_V
#define BW_V 0
#endif

namespace sky {

namespace bw {
constexpr int NW = 4, NT = NW * 64;
constexpr int TH = 8, TW = 16, HWD = TW + 2, HRW = TH + 2, NHP = HWD * HRW;      // 180 halo pixels
constexpr int C = 128, NCH = 2;
constexpr int XPIX = 184, PL = XPIX * 32, CHB = 4 * PL;       // pixel slots per plane, bytes per plane (23 * 256), per chunk image
constexpr int XDMA = (XPIX + 31) / 32;                        // DMA pieces per plane (6, the last one 24 slots)
constexpr int XLAST = (XPIX - (XDMA - 1) * 32) * 2;           // active lanes of the last piece (48)
constexpr int TILE_BYTES = NCH * CHB;                         // 47 104
constexpr int SLAB = C * 128;                                 // one weight slab, 16 KB
constexpr int NST = 2;
constexpr int NFR = (NHP + 15) / 16;                          // halo pixel fragments (12)
constexpr int NSTEP = NCH + 9 * NCH;                          // 20: cv1 row halves, then (chunk, tap)
constexpr int LDS_BYTES = TILE_BYTES + NST * SLAB + 2 * C * 4;
static_assert(NFR == 3 * NW, "three halo fragments per wave");
static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
static_assert(PL % 256 == 0 && CHB % 256 == 0, "planes a multiple of 256 B apart: conflict-free fragment reads");
static_assert(NSTEP % NST == 0, "the ring stage of a step must not depend on the tile");
}  // namespace bw

__device__ __forceinline__ void bw_dma16(__amdgpu_buffer_rsrc_t rsrc, char* dst, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
}
// this wave's DMA pieces (and older stores) have landed, its LDS queue is empty; then the raw barrier
__device__ __forceinline__ void bw_wait_barrier()
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__global__ void __launch_bounds__(bw::NT, 2) bneck128w_kernel(const ConvArgs a)
{
    using namespace bw;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const xt = smem;                                    // x tile, then u in place
    char* const ring = smem + TILE_BYTES;
    float* const lb1 = reinterpret_cast<float*>(ring + NST * SLAB);      // cv1 bias [128]
    float* const lb2 = lb1 + C;                                          // cv2 bias [128]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int hc = wave & 1, pg = wave >> 1;                  // channel half, pixel group (tile rows 4 pg .. 4 pg + 3)
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int ntile = a.B * tiles_y * tiles_x;
    int tile, tstep, tend;                                   // XCD-aware tile order (conv_frag.h: tile_walk)
    tile_walk(ntile, tile, tstep, tend);
    if (tile >= tend) return;
    const int pix_b = a.ldi * 2;
    const int w1pitch = a.c1_Kpad * 2, w2pitch = a.Kpad * 2;

    for (int i = tid; i < C; i += NT) { lb1[i] = a.c1_bias[i]; lb2[i] = a.bias[i]; }

    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, (int)a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)a.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.c1_w), 0, (int)((long)C * w1pitch), 0x00020000);
    const __amdgpu_buffer_rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, (int)((long)C * w2pitch), 0x00020000);

    // weight DMA: a slab is 16 pieces of 1 KB; this wave issues pieces 4 wave .. 4 wave + 3.
    //   W2 slab [128 rows][128 B]: piece = 8 rows; lane -> row, stored chunk lane & 7 = source chunk (lane & 7) ^ ((row >> 1) & 7);
    //   W1 slab hh = rows 64 hh .. 64 hh + 63 x all of K, [64 rows][256 B]: piece = 4 rows; stored chunk lane & 15 = source chunk (lane & 15) ^ (row & 15);
    //   (fragment j, MFMA row r) -> channel (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3), as everywhere
    int wrel1[4], wrel2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = (wave * 4 + q) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int j = row >> 4, r = row & 15;
        const int ch = (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3);
        wrel2[q] = ch * w2pitch + c * 16;
        const int row1 = (wave * 4 + q) * 4 + (lane >> 4);
        const int c1 = (lane & 15) ^ (row1 & 15);
        const int j1 = row1 >> 4, r1 = row1 & 15;
        const int ch1 = (j1 >> 1) * 32 + (r1 >> 2) * 8 + (j1 & 1) * 4 + (r1 & 3);
        wrel1[q] = ch1 * w1pitch + c1 * 16;
    }
    // slab of in-tile step s (0, 1: W1 row halves; 2 ..: W2 (chunk, tap)) into ring stage s & 1
    auto issue_slab = [&](int s) {
        char* const dst = ring + (s & (NST - 1)) * SLAB + wave * 4096;
        if (s < NCH) {
#pragma unroll
            for (int q = 0; q < 4; ++q) bw_dma16(w1rsrc, dst + q * 1024, wrel1[q], s * 64 * w1pitch);
        } else {
            const int g = s - NCH, chunk = g / 9, tap = g - chunk * 9;
#pragma unroll
            for (int q = 0; q < 4; ++q) bw_dma16(w2rsrc, dst + q * 1024, wrel2[q], tap * (C * 2) + chunk * 128);
        }
    };
    auto decode_tile = [&](int t, int& bimg, int& y0, int& x0) {
        const int tx = t % tiles_x;
        const int q = t / tiles_x;
        bimg = q / tiles_y;
        y0 = (q - bimg * tiles_y) * TH;
        x0 = tx * TW;
    };
    // x tile DMA: this wave fills plane `wave` of both chunk images; in piece b lane -> pixel slot p = b * 32 + (lane >> 1), 16-byte
    // half lane & 1 = K-step (lane & 1) ^ (p >> 3 & 1); outside the image: offset -1 -> the range check writes zeros
    auto issue_x = [&](int bimg, int y0, int x0, int cmask) {
        const int base = ((bimg * a.H + y0 - 1) * a.W + x0 - 1) * pix_b + wave * 16;
        int ln = lane;
        asm volatile("" : "+v"(ln));                          // (opaque: the per-lane part is recomputed per tile, not kept in registers)
#pragma unroll
        for (int b = 0; b < XDMA; ++b) {
            const int p = b * 32 + (ln >> 1);
            const int hy = (p * 3641) >> 16, hx = p - hy * HWD;            // p / 18
            const int kk = (ln & 1) ^ ((p >> 3) & 1);
            const bool ok = p < NHP && (unsigned)(y0 - 1 + hy) < (unsigned)a.H && (unsigned)(x0 - 1 + hx) < (unsigned)a.W;
            const int off = ok ? base + (hy * a.W + hx) * pix_b + kk * 64 : -1;
            if (b < XDMA - 1 || ln < XLAST) {                 // the last piece ends at slot 183: the plane behind it starts there
#pragma unroll
                for (int c = 0; c < NCH; ++c)
                    if (cmask & (1 << c)) bw_dma16(irsrc, xt + c * CHB + wave * PL + b * 1024, off, c * 128);
            }
        }
    };

    const int arow = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);       // W2 fragment: row fr of a fragment, K-step 0 (K-step 1: ^ 64)

    int bimg, y0, x0;
    decode_tile(tile, bimg, y0, x0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the bias writes above
    issue_slab(0);
    issue_x(bimg, y0, x0, 3);
    bool first = true;

    for (;;) {
        const int next = tile + tstep;
        const bool has_next = next < tend;
        Out8<__bf16>::raw_t resv[4][2];                       // residual x of this lane's 4 x 2 output vectors

        // ---------------- steps 0, 1: cv1 on this wave's halo fragments wave, wave + 4, wave + 8 ----------------
        {
            int frq = fr, fqq = fq;                           // (opaque per tile)
            asm volatile("" : "+v"(frq), "+v"(fqq));
            // x tile and slab 0 have landed (BW_V & 2: the epilogue's 16 stores, younger than both, may still be in flight)
            if ((BW_V & 2) && !first) {
                asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            } else {
                bw_wait_barrier();
            }
            issue_slab(1);
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
                const int p = (wave + NW * i) * 16 + frq;     // slots 180 .. 191 of the last fragment hold no pixel: clamp, never stored
                const int pc = p < XPIX ? p : XPIX - 1;
                const int A = fqq * PL + pc * 32 + (((pc >> 3) & 1) << 4);
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    xf[i][2 * c] = *reinterpret_cast<const u32x4_t*>(xt + c * CHB + A);
                    xf[i][2 * c + 1] = *reinterpret_cast<const u32x4_t*>(xt + c * CHB + (A ^ 16));
                }
                const int hy = (p * 3641) >> 16, hx = p - hy * HWD;
                inside[i] = p < NHP && (unsigned)(y0 - 1 + hy) < (unsigned)a.H && (unsigned)(x0 - 1 + hx) < (unsigned)a.W;
                pfr[i] = p < NHP ? fqq * PL + p * 32 + (((p >> 3) & 1) << 4) : -1;
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
                        for (int i = 0; i < 3; ++i) S1<__bf16>::mma(wq[gg % 3], xf[i][gg & 3], au[gg >> 2][i]);
                        between(gg);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            // u = SiLU(. + b1) -> bf16 -> back into the tile, in place: 32-channel group sg of fragment i = chunk sg >> 1, K-step sg & 1
            auto act_store = [&](const f32x4_t (&au)[4][3], int hh, int i, int sq) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {                    // (b1 was the accumulators' initial value)
                    v[e] = S1<__bf16>::silu(au[2 * sq][i][e]);
                    v[4 + e] = S1<__bf16>::silu(au[2 * sq + 1][i][e]);
                }
                Out8<__bf16>::raw_t o = Out8<__bf16>::pack(v, 1.0f);
                if (!inside[i]) o.a = u32x4_t{0u, 0u, 0u, 0u};
                if (pfr[i] >= 0) *reinterpret_cast<u32x4_t*>(xt + hh * CHB + (sq ? pfr[i] ^ 16 : pfr[i])) = o.a;
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
            // every wave has its x fragments and residual vectors in registers behind this barrier: the tile may be rewritten
            bw_wait_barrier();
            issue_slab(2);
            half_mma(1, au1, [&](int gg) {
                if (gg == 1) act_store(au0, 0, 0, 0);
                if (gg == 3) act_store(au0, 0, 0, 1);
                if (gg == 5) act_store(au0, 0, 1, 0);
                if (gg == 7) act_store(au0, 0, 1, 1);
                if (gg == 9) act_store(au0, 0, 2, 0);
                if (gg == 11) act_store(au0, 0, 2, 1);
            });
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int sq = 0; sq < 2; ++sq) act_store(au1, 1, i, sq);
        }

        // ---------------- steps 2 .. 19: the 3x3 over u, (chunk, tap) by (chunk, tap) ----------------
        f32x4_t acc[4][4];                                    // start from cv2's bias: fragment j = channels 64 hc + (j >> 1) * 32 + fq * 8 + (j & 1) * 4 ..
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4_t c = *reinterpret_cast<const f32x4_t*>(lb2 + 64 * hc + (j >> 1) * 32 + fq * 8 + (j & 1) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = c;
        }
        // pixel fragment of tile row 4 pg + r (r = i + ky: 0 .. 5), column shift kx: slot = (4 pg + r) * 18 + fr + kx, address = plane +
        // slot * 32 + 16 * bit 3 of the slot; 18 = 16 + 2, so that bit is bit 3 of fr + 8 pg + c with c = 2 r + kx in 0 .. 12: thirteen
        // per-lane bases cover every tap, everything else is an immediate offset of the ds_read
        int pb[13];
        {
            int frt = fr;
            asm volatile("" : "+v"(frt));
#pragma unroll
            for (int c = 0; c < 13; ++c) pb[c] = fq * PL + ((4 * pg) * HWD + frt) * 32 + (((frt + 8 * pg + c) & 8) << 1);
        }
        int nb = 0, ny0 = 0, nx0 = 0;
        if (has_next) decode_tile(next, nb, ny0, nx0);
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
        u32x4_t pf0[2][4];                                    // [step parity]: pixel fragments of K-step 0 (u is static: requested a step ahead)
        if (BW_V & 1) __builtin_amdgcn_s_setprio(1);          // tap steps: this wave's MFMAs ahead of the other workgroup's cv1 / epilogue arithmetic
#pragma unroll
        for (int s = NCH; s < NSTEP; ++s) {
            const int p = s & 1;
            if ((BW_V & 16) && s > NCH) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else bw_wait_barrier();                           // slab s has landed everywhere, step s - 1 is over everywhere (s = 2: u is complete)
            if (!(BW_V & 8) || first) {
                if (s + 1 < NSTEP) issue_slab(s + 1);
                else if (has_next) issue_slab(0);
            }
            if ((BW_V & 4) && s == NCH + 9 && has_next) issue_x(nb, ny0, nx0, 1);      // chunk 0 of u is dead behind this barrier
            u32x4_t pf1[4], wq[4][2];                         // K-step 1 pixels; weight pairs of the four (K-step, channel pair) groups
            if (s == NCH) {
#pragma unroll
                for (int i = 0; i < 4; ++i) pf0[p][i] = tap_frag(s, i, 0);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) { wq[0][h] = wfrag(s, 0, 0, h); wq[1][h] = wfrag(s, 0, 1, h); }
#pragma unroll
            for (int i = 0; i < 4; ++i) pf1[i] = tap_frag(s, i, 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i) S1<__bf16>::mma(wq[0][h], pf0[p][i], acc[h][i]);
#pragma unroll
            for (int h = 0; h < 2; ++h) wq[2][h] = wfrag(s, 1, 0, h);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i) S1<__bf16>::mma(wq[1][h], pf0[p][i], acc[2 + h][i]);
#pragma unroll
            for (int h = 0; h < 2; ++h) wq[3][h] = wfrag(s, 1, 1, h);
            if (s + 1 < NSTEP) {                              // K-step 0 pixels of the next tap
#pragma unroll
                for (int i = 0; i < 4; ++i) pf0[p ^ 1][i] = tap_frag(s + 1, i, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i) S1<__bf16>::mma(wq[2][h], pf1[i], acc[h][i]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i) S1<__bf16>::mma(wq[3][h], pf1[i], acc[2 + h][i]);
            __builtin_amdgcn_sched_barrier(0);
        }

        // every wave is done with u before the next tile's x lands on it
        bw_wait_barrier();
        if (BW_V & 1) __builtin_amdgcn_s_setprio(0);
        if (has_next) issue_x(nb, ny0, nx0, (BW_V & 4) ? 2 : 3);
        // ---------------- epilogue: bias, SiLU, + x, bf16, 16-byte stores ----------------
        int fre = fr, fqe = fq;
        asm volatile("" : "+v"(fre), "+v"(fqe));
        const bool colok = x0 + fre < a.W;
        const int off0 = (((bimg * a.H + y0 + 4 * pg) * a.W + x0 + fre) * a.ldo + 64 * hc + 8 * fqe) * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = colok && y0 + 4 * pg + i < a.H;
            // masked lanes: offset 0x80000000 stays out of range after the immediate is added (the constant goes into the vector offset /
            // immediate, never into soffset: DESIGN.md section 3, store-data hazard)
            const int ooff = ok ? off0 + i * a.W * a.ldo * 2 : (int)0x80000000;
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xx = e < 4 ? acc[2 * sp][i][e] : acc[2 * sp + 1][i][e - 4];      // (b2 was the accumulators' initial value)
                    v[e] = S1<__bf16>::silu(xx);
                    if (a.c1_res) {
                        // multiply and residual add round separately, as in the 128-channel halo-tile kernel's epilogue and k_bneck.hip
#pragma clang fp contract(off)
                        const unsigned rw = resv[i][sp].a[e >> 1];
                        const float res = (e & 1) ? __uint_as_float(rw & 0xffff0000u) : __uint_as_float(rw << 16);
                        v[e] = v[e] + res;
                    }
                }
                Out8<__bf16>::store(Out8<__bf16>::pack(v, 1.0f), orsrc, ooff + sp * 64);
            }
        }
        if (!has_next) break;
        tile = next; bimg = nb; y0 = ny0; x0 = nx0;
        first = false;
    }
}

// plan-time question (c1_w may not be set yet): would this cv1 + 3x3 pair run on the kernel?
bool bneck128w_shape_ok(const ConvArgs& a)
{
    const int th = (a.H + bw::TH - 1) / bw::TH, tw = (a.W + bw::TW - 1) / bw::TW;
    const double cover = (double)a.H * a.W / ((double)th * tw * (bw::TH * bw::TW));
    if (!(a.opts & OPT_HALO_FORCE) && cover < 0.75) return false;          // partially filled tiles waste matrix work
    return a.ks == 3 && a.stride == 1 && a.pad == 1 && a.Cin == bw::C && a.Cout == bw::C && a.c1_Kpad >= bw::C &&
           a.Kpad >= 9 * bw::C && a.ldi % 8 == 0 && a.ldo % 8 == 0 && a.in_bytes != 0 && a.H >= 1 && a.W >= 1 && a.act == ACT_SILU &&
           a.out_bytes != 0 && !a.head && !a.up2 && !a.out_f32 && !a.src_mode && !a.f2_w && !a.res &&
           !(a.opts & (OPT_HALO_OFF | OPT_NO_FUSE_CV1 | OPT_NO_BNECK128 | OPT_BNECK128_SOLO));
}

hipError_t launch_bneck128w(const ConvArgs& a0, hipStream_t s)
{
    if (!a0.c1_w || !bneck128w_shape_ok(a0)) return hipErrorNotSupported;
    ConvArgs a = a0;
    a.dbg = 0;
    static size_t attr[16] = {0};
    {
        const hipError_t e = ensure_lds_attr(reinterpret_cast<const void*>(bneck128w_kernel), bw::LDS_BYTES, a.device, attr);
        if (e != hipSuccess) return e;
    }
    const int ntile = a.B * ((a.H + bw::TH - 1) / bw::TH) * ((a.W + bw::TW - 1) / bw::TW);
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
    const int slots = 2 * n_cu;                               // two workgroups per CU
    const int gx = ntile < slots ? ntile : slots;
    hipLaunchKernelGGL(bneck128w_kernel, dim3(gx), dim3(bw::NT), bw::LDS_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace sky
