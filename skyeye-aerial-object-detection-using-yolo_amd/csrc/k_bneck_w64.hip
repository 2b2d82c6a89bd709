// BottleneckBlock(64, 64) of the bf16 engine as ONE kernel with THREE workgroups per CU: x + cv2_3x3(cv1_1x1(x))      reference blocks.py:69-90
//
// The halo-tile kernel's fused form (conv_halo_kernel<.., CV1>: a 16 x 16 tile, 62 KB of LDS, two workgroups per CU) in k_bneck_w8.hip's geometry
// -- 64 bf16 channels are ONE 128-byte chunk -- with 8 x 16 tiles and 40 448 bytes of LDS: three 4-wave workgroups share a CU (three waves per SIMD),
// each in another phase of its tile (x / W1 landing, cv1, taps, epilogue), so the waits of one are filled by two others instead of one.  (Four fit
// -- BW64_WG -- and measure the same, 0.0965 against 0.0974 ms; their 128-register budget spills five registers, 135 are used at three.)
//   * x tile: 10 x 18 halo pixels x 128 bytes in the halo kernels' layout [K-group plane f][pixel slot, 184][2 x 16 B] = 23 552 bytes;
//   * cv1: ONE slab W1 [64 rows][128 B] (step 0), u = SiLU(W1 x + b1) on the 12 pixel fragments of the halo tile (three per wave, all 64
//     channels), written back over x IN PLACE (zeros outside the image); the residual vectors of a wave's own output pixels are read first;
//   * the 9 taps (steps 1 .. 9): one slab W2 [64 rows][128 B] = 8 KB each through a two-stage ring; wave (pixel group pg, channel half hc) owns tile
//     rows 4 pg .. 4 pg + 3 x 32 output channels (4 x 2 accumulator fragments): 16 MFMAs per tap;
//   * epilogue: SiLU (the bias was the accumulators' initial value), + x, bf16, 16-byte stores.
// K order (tap, 64-byte K-step), every bf16 rounding point and the activation arithmetic are those of the halo-tile kernel's fused form and of
// the two-launch form: the three are bit-identical (tests/test_gpu_fused_cv1.py).  +41 % work on cv1 (10 x 18 for 8 x 16 pixels) against +27 %.
#include "sky_kernels.h"

#include "conv_frag.h"

namespace sky {

namespace bw64 {
constexpr int NW = 4, NT = NW * 64;
constexpr int TH = 8, TW = 16, HWD = TW + 2, HRW = TH + 2, NHP = HWD * HRW;      // 180 halo pixels
constexpr int C = 64;
constexpr int XPIX = 184, PL = XPIX * 32;                     // pixel slots per plane, bytes per plane (23 * 256)
constexpr int XDMA = (XPIX + 31) / 32;                        // DMA pieces per plane (6, the last one 24 slots)
constexpr int XLAST = (XPIX - (XDMA - 1) * 32) * 2;           // active lanes of the last piece (48)
constexpr int TILE_BYTES = 4 * PL;                            // 23 552
constexpr int SLAB = C * 128;                                 // one weight slab, 8 KB
constexpr int NST = 2;
constexpr int NFR = (NHP + 15) / 16;                          // halo pixel fragments (12)
constexpr int NSTEP = 1 + 9;                                  // W1, then the taps
constexpr int LDS_BYTES = TILE_BYTES + NST * SLAB + 2 * C * 4;
#ifndef BW64_V
#define BW64_V 0                                              // compile-time experiment bits: 1 the next tap's pixel fragments requested a step ahead (+32 registers)
#endif
#ifndef BW64_WG
#define BW64_WG 3                                             // workgroups per CU the kernel is compiled for (register budget 512 / BW64_WG per lane)
#endif
static_assert(NFR == 3 * NW, "three halo fragments per wave");
static_assert(BW64_WG * LDS_BYTES <= 160 * 1024, "workgroups per CU");
static_assert(PL % 256 == 0, "planes a multiple of 256 B apart: conflict-free fragment reads");
static_assert(NSTEP % NST == 0, "the ring stage of a step must not depend on the tile");
}  // namespace bw64

__device__ __forceinline__ void bw64_dma16(__amdgpu_buffer_rsrc_t rsrc, char* dst, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void bw64_wait_barrier()
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__global__ void __launch_bounds__(bw64::NT, BW64_WG) bneck64w_kernel(const ConvArgs a)
{
    using namespace bw64;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const xt = smem;                                    // x tile, then u in place
    char* const ring = smem + TILE_BYTES;
    float* const lb1 = reinterpret_cast<float*>(ring + NST * SLAB);      // cv1 bias [64]
    float* const lb2 = lb1 + C;                                          // cv2 bias [64]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int hc = wave & 1, pg = wave >> 1;                  // channel half (32 channels), pixel group (tile rows 4 pg .. 4 pg + 3)
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

    // weight DMA: a slab [64 rows][128 B] is 8 pieces of 1 KB = 8 rows; this wave issues pieces 2 wave, 2 wave + 1; lane -> row, stored chunk
    // lane & 7 = source chunk (lane & 7) ^ ((row >> 1) & 7); (fragment j, MFMA row r) -> channel (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3)
    int wrel1[2], wrel2[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = (wave * 2 + q) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int j = row >> 4, r = row & 15;
        const int ch = (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3);
        wrel1[q] = ch * w1pitch + c * 16;
        wrel2[q] = ch * w2pitch + c * 16;
    }
    // slab of in-tile step s (0: W1; 1 ..: W2 tap s - 1) into ring stage s & 1
    auto issue_slab = [&](int s) {
        char* const dst = ring + (s & (NST - 1)) * SLAB + wave * 2048;
        if (s == 0) {
#pragma unroll
            for (int q = 0; q < 2; ++q) bw64_dma16(w1rsrc, dst + q * 1024, wrel1[q], 0);
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) bw64_dma16(w2rsrc, dst + q * 1024, wrel2[q], (s - 1) * (C * 2));
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
            if (b < XDMA - 1 || ln < XLAST) bw64_dma16(irsrc, xt + wave * PL + b * 1024, off, 0);      // the last piece ends at slot 183
        }
    };

    const int arow = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);       // weight fragment: row fr of a fragment, K-step 0 (K-step 1: ^ 64)

    int bimg, y0, x0;
    decode_tile(tile, bimg, y0, x0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the bias writes above
    issue_slab(0);
    issue_x(bimg, y0, x0);

    for (;;) {
        const int next = tile + tstep;
        const bool has_next = next < tend;
        Out8<__bf16>::raw_t resv[4];                          // residual x of this lane's 4 output vectors (8 channels each)

        // ---------------- step 0: cv1 on this wave's halo fragments wave, wave + 4, wave + 8 ----------------
        {
            int frq = fr, fqq = fq;                           // (opaque per tile)
            asm volatile("" : "+v"(frq), "+v"(fqq));
            bw64_wait_barrier();                              // x tile and W1 have landed
            issue_slab(1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {                     // channels 32 hc + 8 fq ..: K-step hc of plane fq
                const int pc = (4 * pg + i + 1) * HWD + 1 + frq;
                resv[i].a = *reinterpret_cast<const u32x4_t*>(xt + fqq * PL + pc * 32 + ((hc ^ ((pc >> 3) & 1)) << 4));
            }
            u32x4_t xf[3][2];                                 // [fragment][64-byte K-step]
            int pfr[3];
            bool inside[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int p = (wave + NW * i) * 16 + frq;     // slots 180 .. 191 of the last fragment hold no pixel: clamp, never stored
                const int pc = p < XPIX ? p : XPIX - 1;
                const int A = fqq * PL + pc * 32 + (((pc >> 3) & 1) << 4);
                xf[i][0] = *reinterpret_cast<const u32x4_t*>(xt + A);
                xf[i][1] = *reinterpret_cast<const u32x4_t*>(xt + (A ^ 16));
                const int hy = (p * 3641) >> 16, hx = p - hy * HWD;
                inside[i] = p < NHP && (unsigned)(y0 - 1 + hy) < (unsigned)a.H && (unsigned)(x0 - 1 + hx) < (unsigned)a.W;
                pfr[i] = p < NHP ? fqq * PL + p * 32 + (((p >> 3) & 1) << 4) : -1;
            }
            // the accumulators start from the bias (k_conv_halo.hip: acc_start): fragment j = channels (j >> 1) * 32 + fq * 8 + (j & 1) * 4 ..
            f32x4_t au[4][3];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4_t c0 = *reinterpret_cast<const f32x4_t*>(lb1 + (j >> 1) * 32 + fqq * 8 + (j & 1) * 4);
#pragma unroll
                for (int i = 0; i < 3; ++i) au[j][i] = c0;
            }
            // K-step 0 of every fragment, then K-step 1 (the halo-tile kernel's order); the weight fragment of group g + 1 is read before the MFMAs of group g
            u32x4_t wq[2];
#pragma unroll
            for (int g = 0; g < 8 + 1; ++g) {
                if (g < 8) wq[g & 1] = *reinterpret_cast<const u32x4_t*>(ring + (g & 3) * 2048 + ((g >> 2) ? arow ^ 64 : arow));
                if (g >= 1) {
                    const int q = g - 1;
#pragma unroll
                    for (int i = 0; i < 3; ++i) S1<__bf16>::mma(wq[q & 1], xf[i][q >> 2], au[q & 3][i]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(au[j][i]));       // (pinned: k_bneck_w8.hip explains)
            // every wave has its x fragments and residual vectors in registers behind this barrier: the tile may be rewritten
            bw64_wait_barrier();
            // u = SiLU(.) -> bf16 -> back into the tile, in place: 32-channel group s of fragment i = K-step s of plane fq
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {                // (b1 was the accumulators' initial value)
                        v[e] = S1<__bf16>::silu(au[2 * s][i][e]);
                        v[4 + e] = S1<__bf16>::silu(au[2 * s + 1][i][e]);
                    }
                    Out8<__bf16>::raw_t o = Out8<__bf16>::pack(v, 1.0f);
                    if (!inside[i]) o.a = u32x4_t{0u, 0u, 0u, 0u};
                    if (pfr[i] >= 0) *reinterpret_cast<u32x4_t*>(xt + (s ? pfr[i] ^ 16 : pfr[i])) = o.a;
                }
        }

        // ---------------- steps 1 .. 9: the 3x3 over u, tap by tap ----------------
        f32x4_t acc[2][4];                                    // start from cv2's bias: fragment j = channels 32 hc + fq * 8 + j * 4 ..
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x4_t c = *reinterpret_cast<const f32x4_t*>(lb2 + 32 * hc + fq * 8 + j * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = c;
        }
        // thirteen per-lane bases cover every tap, everything else is an immediate offset of the ds_read (k_bneck_w.hip)
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
            return *reinterpret_cast<const u32x4_t*>(ring + (st & (NST - 1)) * SLAB + (2 * hc + j) * 2048 + (kk ? arow ^ 64 : arow));
        };
#if BW64_V & 1
        u32x4_t pfb[2][2][4];                                 // [step parity][K-step][pixel fragment]: u is static, the next tap's pixels are requested a step ahead
#endif
#pragma unroll
        for (int s = 1; s < NSTEP; ++s) {
            bw64_wait_barrier();                              // slab s has landed everywhere, step s - 1 is over everywhere (s = 1: u is complete)
            if (s + 1 < NSTEP) issue_slab(s + 1);
            else if (has_next) issue_slab(0);
            u32x4_t wq[2][2];                                 // [K-step][channel fragment]
#if BW64_V & 1
            auto& pf = pfb[s & 1];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 0; j < 2; ++j) wq[kk][j] = wfrag(s, kk, j);
            if (s == 1) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i) pf[kk][i] = tap_frag(0, i, kk);
            }
            if (s + 1 < NSTEP) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i) pfb[(s & 1) ^ 1][kk][i] = tap_frag(s, i, kk);
            }
#else
            u32x4_t pf[2][4];                                 // [K-step][pixel fragment]
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int j = 0; j < 2; ++j) wq[kk][j] = wfrag(s, kk, j);
#pragma unroll
                for (int i = 0; i < 4; ++i) pf[kk][i] = tap_frag(s - 1, i, kk);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) S1<__bf16>::mma(wq[kk][j], pf[kk][i], acc[j][i]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(acc[j][i]));      // (pinned per step)
            __builtin_amdgcn_sched_barrier(0);
        }

        // every wave is done with u before the next tile's x lands on it
        bw64_wait_barrier();
        if (has_next) issue_x(nb, ny0, nx0);
        // ---------------- epilogue: SiLU, + x, bf16, 16-byte stores ----------------
        int fre = fr, fqe = fq;
        asm volatile("" : "+v"(fre), "+v"(fqe));
        const bool colok = x0 + fre < a.W;
        const int off0 = (((bimg * a.H + y0 + 4 * pg) * a.W + x0 + fre) * a.ldo + 32 * hc + 8 * fqe) * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = colok && y0 + 4 * pg + i < a.H;
            // masked lanes: offset 0x80000000 stays out of range (the constant goes into the vector offset / immediate, never into soffset:
            // DESIGN.md section 3, store-data hazard)
            const int ooff = ok ? off0 + i * a.W * a.ldo * 2 : (int)0x80000000;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xx = e < 4 ? acc[0][i][e] : acc[1][i][e - 4];      // (b2 was the accumulators' initial value)
                v[e] = S1<__bf16>::silu(xx);
                if (a.c1_res) {
                    // multiply and residual add round separately, as in the halo-tile kernel's epilogue
#pragma clang fp contract(off)
                    const unsigned rw = resv[i].a[e >> 1];
                    const float res = (e & 1) ? __uint_as_float(rw & 0xffff0000u) : __uint_as_float(rw << 16);
                    v[e] = v[e] + res;
                }
            }
            Out8<__bf16>::store(Out8<__bf16>::pack(v, 1.0f), orsrc, ooff);
        }
        if (!has_next) break;
        tile = next; bimg = nb; y0 = ny0; x0 = nx0;
    }
}

// plan-time question (c1_w may not be set yet): would this cv1 + 3x3 pair run on the kernel?
bool bneck64w_shape_ok(const ConvArgs& a)
{
    const int th = (a.H + bw64::TH - 1) / bw64::TH, tw = (a.W + bw64::TW - 1) / bw64::TW;
    const double cover = (double)a.H * a.W / ((double)th * tw * (bw64::TH * bw64::TW));
    if (!(a.opts & OPT_HALO_FORCE) && cover < 0.75) return false;          // partially filled tiles waste matrix work
    return a.ks == 3 && a.stride == 1 && a.pad == 1 && a.Cin == bw64::C && a.Cout == bw64::C && a.c1_Kpad >= bw64::C &&
           a.Kpad >= 9 * bw64::C && a.ldi % 8 == 0 && a.ldo % 8 == 0 && a.in_bytes != 0 && a.H >= 1 && a.W >= 1 && a.act == ACT_SILU &&
           a.out_bytes != 0 && !a.head && !a.up2 && !a.out_f32 && !a.src_mode && !a.f2_w && !a.res &&
           !(a.opts & (OPT_HALO_OFF | OPT_NO_FUSE_CV1 | OPT_NO_BNECK64W));
}

hipError_t launch_bneck64w(const ConvArgs& a0, hipStream_t s)
{
    if (!a0.c1_w || !bneck64w_shape_ok(a0)) return hipErrorNotSupported;
    ConvArgs a = a0;
    a.dbg = 0;
    static size_t attr[16] = {0};
    {
        const hipError_t e = ensure_lds_attr(reinterpret_cast<const void*>(bneck64w_kernel), bw64::LDS_BYTES, a.device, attr);
        if (e != hipSuccess) return e;
    }
    const int ntile = a.B * ((a.H + bw64::TH - 1) / bw64::TH) * ((a.W + bw64::TW - 1) / bw64::TW);
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
    const int slots = BW64_WG * n_cu;
    const int gx = ntile < slots ? ntile : slots;
    hipLaunchKernelGGL(bneck64w_kernel, dim3(gx), dim3(bw64::NT), bw64::LDS_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace sky
