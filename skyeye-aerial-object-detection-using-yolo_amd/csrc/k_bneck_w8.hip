// BottleneckBlock(128, 128) of the fp8 engine as ONE kernel: x + cv2_3x3(cv1_1x1(x))      reference blocks.py:69-90
//
// k_bneck_w.hip's structure (four waves, an 8 x 16 output tile, two workgroups per CU, weights through a two-stage ring by LDS-DMA) in
// e4m3: 128 channels are ONE 128-byte chunk, so
//   * x tile: 10 x 18 halo pixels x 128 bytes in the halo kernels' layout [K-group plane f][pixel slot, 184][2 x 16 B] = 23 552 bytes;
//   * cv1: ONE slab W1 [128 rows][128 B] (step 0), u = e4m3(SiLU(acc * m1 + b1) / s_u) on the 12 pixel fragments of the halo tile (three
//     per wave, all 128 channels: 24 block-scaled 16x16x128 instructions), written back over x IN PLACE as 8-byte pieces (zeros outside
//     the image); the residual bytes of a wave's own output pixels are read from the x tile first;
//   * the 9 taps (steps 1 .. 9): one slab W2 [128 rows][128 B] each; wave (pixel group pg, channel half hc) owns tile rows 4 pg .. 4 pg + 3
//     x 64 output channels: 16 instructions of K = 128 per tap;
//   * epilogue: acc * m2 + b2, SiLU, + x * s_x, / s_out, e4m3, 8-byte stores.
// Every K pairing (fp8_mma128: the two 64-byte K-steps of a chunk in one instruction), every quantisation point and the epilogue
// arithmetic are those of the two-launch form (conv_stream_kernel<fp8> for cv1, conv_halo_kernel<fp8, 8> for the 3x3 + residual):
// bit-identical (tests/test_gpu_bneck128.py).  s_u is the scale of the hidden tensor, which exists in the plan as a scale carrier only
// (engine.cpp: bottleneck); the calibration twin materialises it.
#include "sky_kernels.h"

#include "conv_frag.h"

namespace sky {

namespace bw8 {
constexpr int NW = 4, NT = NW * 64, C = 128;
constexpr int SLAB = C * 128;                                 // one weight slab, 16 KB
constexpr int NST = 2;
constexpr int NSTEP = 1 + 9;                                  // W1, then the taps
static_assert(NSTEP % NST == 0, "the ring stage of a step must not depend on the tile");
// TH = 8: an 8 x 16 tile, waves = 2 pixel groups x 2 channel halves (64 pixels x 64 channels each); TH = 16: a 16 x 16 tile, wave = tile rows 4 w ..
// 4 w + 3 x all 128 channels (the halo-tile kernel's wave tile: 0.75 instead of 1 fragment read per instruction, 27 instead of 41 % more cv1 work,
// half the barriers per pixel) -- measured SLOWER (launch_bneck128w8), kept behind SKY_BNECK128=solo
template <int TH_> struct Geo {
    static constexpr int TH = TH_, TW = 16, HWD = TW + 2, HRW = TH + 2, NHP = HWD * HRW;      // 180 / 324 halo pixels
    static constexpr int NPG = TH / 4, NHC = NW / NPG;        // pixel groups x channel parts = waves
    static constexpr int CW = C / NHC, NFJ = CW / 16;         // channels and channel fragments per wave
    static constexpr int XPIX = TH == 8 ? 184 : 352, PL = XPIX * 32;      // pixel slots per plane, bytes per plane (23 / 44 x 256)
    static constexpr int XDMA = (XPIX + 31) / 32;             // DMA pieces per plane (6, the last one 24 slots; 11)
    static constexpr int XLAST = (XPIX - (XDMA - 1) * 32) * 2;      // active lanes of the last piece (48; 64)
    static constexpr int TILE_BYTES = 4 * PL;                 // 23 552 / 45 056
    static constexpr int NFR = (NHP + 15) / 16;               // halo pixel fragments (12 / 21)
    static constexpr int NPASS = (NFR + 3 * NW - 1) / (3 * NW);      // cv1 passes of three fragments per wave (1 / 2)
    static constexpr int LDS_BYTES = TILE_BYTES + NST * SLAB + 4 * C * 4;
    static_assert(TH == 8 || TH == 16, "tile heights");
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
    static_assert(PL % 256 == 0, "planes a multiple of 256 B apart: conflict-free fragment reads");
};
}  // namespace bw8

__device__ __forceinline__ void bw8_dma16(__amdgpu_buffer_rsrc_t rsrc, char* dst, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void bw8_wait_barrier()
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int TH_>
__global__ void __launch_bounds__(bw8::NT, 2) bneck128w8_kernel(const ConvArgs a)
{
    using namespace bw8;
    using G = Geo<TH_>;
    constexpr int TH = G::TH, TW = G::TW, HWD = G::HWD, NHP = G::NHP, NFJ = G::NFJ, CW = G::CW, XPIX = G::XPIX, PL = G::PL, XDMA = G::XDMA, XLAST = G::XLAST;
    constexpr int TILE_BYTES = G::TILE_BYTES, NPASS = G::NPASS;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const xt = smem;                                    // x tile, then u in place
    char* const ring = smem + TILE_BYTES;
    float* const lb1 = reinterpret_cast<float*>(ring + NST * SLAB);      // cv1 bias [128], multipliers [128]
    float* const lm1 = lb1 + C;
    float* const lb2 = lm1 + C;                                          // cv2 bias [128], multipliers [128]
    float* const lm2 = lb2 + C;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int hc = wave % G::NHC, pg = wave / G::NHC;         // channel part, pixel group (tile rows 4 pg .. 4 pg + 3)
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int ntile = a.B * tiles_y * tiles_x;
    int tile, tstep, tend;                                   // XCD-aware tile order (conv_frag.h: tile_walk)
    tile_walk(ntile, tile, tstep, tend);
    if (tile >= tend) return;
    const int pix_b = a.ldi;
    const int w1pitch = a.c1_Kpad, w2pitch = a.Kpad;

    for (int i = tid; i < C; i += NT) {
        lb1[i] = a.c1_bias[i]; lm1[i] = a.c1_mult ? a.c1_mult[i] : 1.0f;
        lb2[i] = a.bias[i]; lm2[i] = a.mult ? a.mult[i] : 1.0f;
    }

    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, (int)a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)a.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.c1_w), 0, (int)((long)C * w1pitch), 0x00020000);
    const __amdgpu_buffer_rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, (int)((long)C * w2pitch), 0x00020000);

    // weight DMA: a slab [128 rows][128 B] is 16 pieces of 1 KB = 8 rows; this wave issues pieces 4 wave .. 4 wave + 3; lane -> row,
    // stored chunk lane & 7 = source chunk (lane & 7) ^ ((row >> 1) & 7); (fragment j, MFMA row r) -> channel (j >> 1) * 32 + (r >> 2) * 8 +
    // (j & 1) * 4 + (r & 3), as everywhere
    int wrel1[4], wrel2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = (wave * 4 + q) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int j = row >> 4, r = row & 15;
        const int ch = (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3);
        wrel1[q] = ch * w1pitch + c * 16;
        wrel2[q] = ch * w2pitch + c * 16;
    }
    // slab of in-tile step s (0: W1; 1 ..: W2 tap s - 1) into ring stage s & 1
    auto issue_slab = [&](int s) {
        char* const dst = ring + (s & (NST - 1)) * SLAB + wave * 4096;
        if (s == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) bw8_dma16(w1rsrc, dst + q * 1024, wrel1[q], 0);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) bw8_dma16(w2rsrc, dst + q * 1024, wrel2[q], (s - 1) * C);
        }
    };
    auto decode_tile = [&](int t, int& bimg, int& y0, int& x0) {
        const int tx = t % tiles_x;
        const int q = t / tiles_x;
        bimg = q / tiles_y;
        y0 = (q - bimg * tiles_y) * TH;
        x0 = tx * TW;
    };
    // x tile DMA: this wave fills plane `wave`; in piece b lane -> pixel slot p = b * 32 + (lane >> 1), 16-byte half lane & 1 = K-step
    // (lane & 1) ^ (p >> 3 & 1); outside the image: offset -1 -> the range check writes zeros
    auto issue_x = [&](int bimg, int y0, int x0) {
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
            if (b < XDMA - 1 || ln < XLAST) bw8_dma16(irsrc, xt + wave * PL + b * 1024, off, 0);       // the last piece ends at slot 183
        }
    };

    const int arow = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);       // weight fragment: row fr of a fragment, K-step 0 (K-step 1: ^ 64)
    // 8 consecutive channels c0 .. c0 + 7 (c0 a multiple of 8) of pixel slot p: K-step c0 >> 6, plane (c0 & 63) >> 4, half (c0 >> 3) & 1 of the 16-byte piece
    auto piece8 = [&](int p, int c0) -> int {
        return ((c0 & 63) >> 4) * PL + p * 32 + ((((c0 >> 6) & 1) ^ ((p >> 3) & 1)) << 4) + ((c0 >> 3) & 1) * 8;
    };

    int bimg, y0, x0;
    decode_tile(tile, bimg, y0, x0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the bias / multiplier writes above
    issue_slab(0);
    issue_x(bimg, y0, x0);

    for (;;) {
        const int next = tile + tstep;
        const bool has_next = next < tend;
        constexpr bool RES_LDS = TH == 8;                     // 16 x 16 tiles: 32 more registers across the taps would spill; the epilogue reads x again (L2)
        Out8<fp8_t>::raw_t resv[4][NFJ / 2];                  // residual x of this lane's 4 x NFJ / 2 output vectors

        // ---------------- step 0: cv1 on this wave's halo fragments wave, wave + 4, wave + 8 ----------------
        {
            int frq = fr, fqq = fq;                           // (opaque per tile)
            asm volatile("" : "+v"(frq), "+v"(fqq));
            bw8_wait_barrier();                               // x tile and W1 have landed
            issue_slab(1);
#pragma unroll
            for (int i = 0; i < (RES_LDS ? 4 : 0); ++i) {
                const int pc = (4 * pg + i + 1) * HWD + 1 + frq;
#pragma unroll
                for (int sp = 0; sp < NFJ / 2; ++sp) resv[i][sp].a = *reinterpret_cast<const u32x2_t*>(xt + piece8(pc, CW * hc + 32 * sp + 8 * fqq));
            }
            // x fragments of ALL passes first (u is written over them in place), then per pass: 24 instructions, results pinned
            u32x4_t xf[NPASS][3][2];                          // [pass][fragment][64-byte K-step]
            int pst[NPASS][3];
            bool inside[NPASS][3];
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int p = ((ps * 3 + i) * NW + wave) * 16 + frq;      // slots past the last pixel hold nothing: clamp, never stored
                    const int pc = p < XPIX ? p : XPIX - 1;
                    const int A = fqq * PL + pc * 32 + (((pc >> 3) & 1) << 4);
                    xf[ps][i][0] = *reinterpret_cast<const u32x4_t*>(xt + A);
                    xf[ps][i][1] = *reinterpret_cast<const u32x4_t*>(xt + (A ^ 16));
                    const int hy = (p * 3641) >> 16, hx = p - hy * HWD;
                    inside[ps][i] = p < NHP && (unsigned)(y0 - 1 + hy) < (unsigned)a.H && (unsigned)(x0 - 1 + hx) < (unsigned)a.W;
                    pst[ps][i] = p < NHP ? p : -1;
                }
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                f32x4_t au[8][3];
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int i = 0; i < 3; ++i) au[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                // the two pieces of fragment j + 1 are read before the 3 instructions of fragment j
                u32x4_t w2[2][2];
#pragma unroll
                for (int g = 0; g < 8 + 1; ++g) {
                    if (g < 8) {
                        w2[g & 1][0] = *reinterpret_cast<const u32x4_t*>(ring + g * 2048 + arow);
                        w2[g & 1][1] = *reinterpret_cast<const u32x4_t*>(ring + g * 2048 + (arow ^ 64));
                    }
                    if (g >= 1) {
                        const int q = g - 1;
#pragma unroll
                        for (int i = 0; i < 3; ++i) fp8_mma128(w2[q & 1][0], w2[q & 1][1], xf[ps][i][0], xf[ps][i][1], au[q][i]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // (the results are pinned here: hipcc otherwise sinks every MFMA chain into the block of the epilogue piece that reads it and
                // carries the fragments there through scratch)
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(au[j][i]));
                // every wave has its x fragments and residual bytes in registers behind this barrier: the tile may be rewritten
                if (ps == 0) bw8_wait_barrier();
                // u = e4m3(SiLU(acc * m1 + b1) / s_u), back into the tile in place: 32-channel group s of fragment i
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int nl = s * 32 + fqq * 8;
                        const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(lb1 + nl), b1 = *reinterpret_cast<const f32x4_t*>(lb1 + nl + 4);
                        const f32x4_t m0 = *reinterpret_cast<const f32x4_t*>(lm1 + nl), m1 = *reinterpret_cast<const f32x4_t*>(lm1 + nl + 4);
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {            // conv_stream_kernel's fp8 epilogue, operation by operation
                            v[e] = au[2 * s][i][e] * m0[e] + b0[e];
                            v[4 + e] = au[2 * s + 1][i][e] * m1[e] + b1[e];
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = S1<fp8_t>::silu(v[e]);
                        Out8<fp8_t>::raw_t o = Out8<fp8_t>::pack(v, a.c1_out_inv_scale);
                        if (!inside[ps][i]) o.a = u32x2_t{0u, 0u};
                        if (pst[ps][i] >= 0) *reinterpret_cast<u32x2_t*>(xt + piece8(pst[ps][i], nl)) = o.a;
                    }
            }
        }

        // ---------------- steps 1 .. 9: the 3x3 over u, tap by tap ----------------
        f32x4_t acc[NFJ][4];
#pragma unroll
        for (int j = 0; j < NFJ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // pixel fragment of tile row 4 pg + r (r = i + ky: 0 .. 5), column shift kx: slot = (4 pg + r) * 18 + fr + kx, address = plane +
        // slot * 32 + 16 * bit 3 of the slot; 18 = 16 + 2, so that bit is bit 3 of fr + 8 pg + c with c = 2 r + kx in 0 .. 12: thirteen
        // per-lane bases cover every tap, everything else is an immediate offset of the ds_read (k_bneck_w.hip)
        int pb[13];
        {
            int frt = fr;
            asm volatile("" : "+v"(frt));
#pragma unroll
            for (int c = 0; c < 13; ++c) pb[c] = fq * PL + ((4 * pg) * HWD + frt) * 32 + (((frt + 8 * pg + c) & 8) << 1);
        }
        int nb = 0, ny0 = 0, nx0 = 0;
        if (has_next) decode_tile(next, nb, ny0, nx0);
        auto tap_frag = [&](int tap, int i, int kk) -> u32x4_t {
            const int ky = tap / 3, kx = tap - ky * 3;
            int q = pb[2 * (i + ky) + kx];
            if (kk) {
                asm volatile("" : "+v"(q));                   // (opaque: else hipcc keeps the 13 ^ 16 variants of pb[] in registers as well)
                q ^= 16;
            }
            return *reinterpret_cast<const u32x4_t*>(xt + q + ((i + ky) * HWD + kx) * 32);
        };
        auto wfrag = [&](int st, int kk, int j) -> u32x4_t {
            return *reinterpret_cast<const u32x4_t*>(ring + (st & (NST - 1)) * SLAB + (NFJ * hc + j) * 2048 + (kk ? arow ^ 64 : arow));
        };
#pragma unroll
        for (int s = 1; s < NSTEP; ++s) {
            bw8_wait_barrier();                               // slab s has landed everywhere, step s - 1 is over everywhere (s = 1: u is complete)
            if (s + 1 < NSTEP) issue_slab(s + 1);
            else if (has_next) issue_slab(0);
            u32x4_t pf[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) { pf[i][0] = tap_frag(s - 1, i, 0); pf[i][1] = tap_frag(s - 1, i, 1); }
            // the two pieces of weight fragment j + 1 are read before the 4 instructions of fragment j (the halo-tile kernel's fp8 pipeline)
            u32x4_t wq[2][2];
#pragma unroll
            for (int j = 0; j < NFJ + 1; ++j) {
                if (j < NFJ) { wq[j & 1][0] = wfrag(s, 0, j); wq[j & 1][1] = wfrag(s, 1, j); }
                if (j >= 1) {
                    const int q = j - 1;
#pragma unroll
                    for (int i = 0; i < 4; ++i) fp8_mma128(wq[q & 1][0], wq[q & 1][1], pf[i][0], pf[i][1], acc[q][i]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < NFJ; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(acc[j][i]));      // (pinned per step, as above)
            __builtin_amdgcn_sched_barrier(0);
        }

        // every wave is done with u before the next tile's x lands on it
        bw8_wait_barrier();
        if (has_next) issue_x(nb, ny0, nx0);
        // ---------------- epilogue: acc * m2 + b2, SiLU, + x, e4m3, 8-byte stores ----------------
        int fre = fr, fqe = fq;
        asm volatile("" : "+v"(fre), "+v"(fqe));
        const bool colok = x0 + fre < a.W;
        const int off0 = ((bimg * a.H + y0 + 4 * pg) * a.W + x0 + fre) * a.ldo + CW * hc + 8 * fqe;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = colok && y0 + 4 * pg + i < a.H;
            // masked lanes: offset 0x80000000 stays out of range after the immediate is added (the constant goes into the vector offset /
            // immediate, never into soffset: DESIGN.md section 3, store-data hazard)
            const int ooff = ok ? off0 + i * a.W * a.ldo : (int)0x80000000;
            if (!RES_LDS && a.c1_res) {
                const int roff = ok ? ((bimg * a.H + y0 + 4 * pg + i) * a.W + x0 + fre) * a.ldi + CW * hc + 8 * fqe : -1;
#pragma unroll
                for (int sp = 0; sp < NFJ / 2; ++sp) resv[i][sp] = Out8<fp8_t>::load(irsrc, roff, sp * 32);
            }
#pragma unroll
            for (int sp = 0; sp < NFJ / 2; ++sp) {
                const int nl = CW * hc + sp * 32 + fqe * 8;
                const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(lb2 + nl), b1 = *reinterpret_cast<const f32x4_t*>(lb2 + nl + 4);
                const f32x4_t m0 = *reinterpret_cast<const f32x4_t*>(lm2 + nl), m1 = *reinterpret_cast<const f32x4_t*>(lm2 + nl + 4);
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {                    // tile_epilogue's fp8 arithmetic (k_conv_halo.hip), operation by operation
                    v[e] = acc[2 * sp][i][e] * m0[e] + b0[e];
                    v[4 + e] = acc[2 * sp + 1][i][e] * m1[e] + b1[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = S1<fp8_t>::silu(v[e]);
                if (a.c1_res) Out8<fp8_t>::add(resv[i][sp], v, a.res_scale);
                Out8<fp8_t>::store(Out8<fp8_t>::pack(v, a.out_inv_scale), orsrc, ooff + sp * 32);
            }
        }
        if (!has_next) break;
        tile = next; bimg = nb; y0 = ny0; x0 = nx0;
    }
}

// plan-time question (c1_w may not be set yet): would this cv1 + 3x3 pair of the fp8 engine run on the kernel?
static double bw8_cover(const ConvArgs& a, int th_)
{
    const int th = (a.H + th_ - 1) / th_, tw = (a.W + 15) / 16;
    return (double)a.H * a.W / ((double)th * tw * (th_ * 16));
}
bool bneck128w8_shape_ok(const ConvArgs& a)
{
    if (!(a.opts & OPT_HALO_FORCE) && bw8_cover(a, 8) < 0.75) return false;          // partially filled tiles waste matrix work
    return a.ks == 3 && a.stride == 1 && a.pad == 1 && a.Cin == bw8::C && a.Cout == bw8::C && a.c1_Kpad >= bw8::C &&
           a.Kpad >= 9 * bw8::C && a.ldi % 16 == 0 && a.ldo % 8 == 0 && a.in_bytes != 0 && a.H >= 1 && a.W >= 1 && a.act == ACT_SILU &&
           a.out_bytes != 0 && !a.head && !a.up2 && !a.out_f32 && !a.src_mode && !a.f2_w && !a.res && (a.out_dt < 0 || a.out_dt == 2) &&
           !(a.opts & (OPT_HALO_OFF | OPT_NO_FUSE_CV1 | OPT_NO_BNECK128));
}

template <int TH>
static hipError_t bw8_launch(const ConvArgs& a, hipStream_t s)
{
    using G = bw8::Geo<TH>;
    static size_t attr[16] = {0};
    {
        const hipError_t e = ensure_lds_attr(reinterpret_cast<const void*>(bneck128w8_kernel<TH>), G::LDS_BYTES, a.device, attr);
        if (e != hipSuccess) return e;
    }
    const int ntile = a.B * ((a.H + TH - 1) / TH) * ((a.W + 15) / 16);
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
    const int slots = 2 * n_cu;                               // two workgroups per CU
    const int gx = ntile < slots ? ntile : slots;
    hipLaunchKernelGGL(bneck128w8_kernel<TH>, dim3(gx), dim3(bw8::NT), G::LDS_BYTES, s, a);
    return hipGetLastError();
}

hipError_t launch_bneck128w8(const ConvArgs& a0, hipStream_t s)
{
    if (!a0.c1_w || !bneck128w8_shape_ok(a0)) return hipErrorNotSupported;
    ConvArgs a = a0;
    a.dbg = 0;
    // 8 x 16 tiles.  Measured against them (skyeye_l fp8 @1536, the twelve 192 x 192 bottlenecks, same process): 16 x 16 tiles 0.330 ms, 8 x 16 tiles
    // 0.270 -- fewer fragment reads per instruction and less cv1 work do not make up for two small workgroups running out of phase.  SKY_BNECK128=solo
    // selects the 16 x 16 form where it covers the map (A/B switch; both forms are bit-identical to the two launches).
    const bool big = (a.opts & OPT_BNECK128_SOLO) && bw8_cover(a, 16) >= 0.75;
    return big ? bw8_launch<16>(a, s) : bw8_launch<8>(a, s);
}

}  // namespace sky
