// BottleneckBlock(64, 64) of the fp8 engine as ONE kernel: x + cv2_3x3(cv1_1x1(x))      reference blocks.py:69-90
//
// k_bneck_w64.hip's structure (four waves, an 8 x 16 output tile, weights through a two-stage ring by LDS-DMA) for 64-BYTE pixels (64 e4m3 channels):
//   * x tile: 10 x 18 halo pixels in the narrow halo kernel's layout [16-byte plane f][pixel slot, 184][16 B] = 11 776 bytes;
//   * cv1: W1 [64 rows][64 B of K] (step 0; one 64-byte K-step = two 16x16x32 fp8 instructions per fragment pair, the streaming kernel's form),
//     u = e4m3(SiLU(acc * m1 + b1) / s_u) written back over x IN PLACE in 8-byte pieces (zeros outside the image);
//   * the nine taps as the narrow halo kernel pairs them: taps (0, 1), (2, 3), (4, 5), (6, 7) each in ONE 16x16x128 block-scaled instruction per
//     fragment pair (K = tap a's 64 bytes | tap b's 64 bytes: one slab [64 rows][128 B] per pair, steps 1 .. 4), tap 8 through the 16x16x32 pair (step 5);
//   * epilogue: acc * m2 + b2, SiLU, + x * s_x, / s_out, e4m3, 8-byte stores.
// 29 KB of LDS: four workgroups per CU.  Bit-identical to the two-launch form on the fused plan's buffers (conv_stream_kernel<fp8> for cv1,
// conv_halo_small_kernel<fp8, 64> for the 3x3 + residual; SKY_BNECK128=pair, tests/test_gpu_bneck128.py).
#include "sky_kernels.h"

#include "conv_frag.h"

namespace sky {

namespace bw6f {
constexpr int NW = 4, NT = NW * 64;
constexpr int TH = 8, TW = 16, HWD = TW + 2, HRW = TH + 2, NHP = HWD * HRW;      // 180 halo pixels
constexpr int C = 64;                                         // channels = bytes per pixel
constexpr int XPIX = 184, PL = XPIX * 16;                     // pixel slots per plane, bytes per 16-byte plane
constexpr int XDMA = (XPIX + 63) / 64;                        // DMA pieces per plane (3, the last one 56 slots)
constexpr int XLAST = XPIX - (XDMA - 1) * 64;                 // active lanes of the last piece (56)
constexpr int TILE_BYTES = 4 * PL;                            // 11 776
constexpr int SLAB = C * 128;                                 // one weight slab [64 rows][128 B], 8 KB
constexpr int NST = 2;
constexpr int NFR = (NHP + 15) / 16;                          // halo pixel fragments (12)
constexpr int NSTEP = 1 + 4 + 1;                              // W1, four tap pairs, tap 8
constexpr int LDS_BYTES = TILE_BYTES + NST * SLAB + 4 * C * 4;
constexpr int WG = 4;                                         // workgroups per CU the kernel is compiled for
static_assert(NFR == 3 * NW, "three halo fragments per wave");
static_assert(WG * LDS_BYTES <= 160 * 1024, "workgroups per CU");
static_assert(NSTEP % NST == 0, "the ring stage of a step must not depend on the tile");
}  // namespace bw6f

__device__ __forceinline__ void bw6f_dma16(__amdgpu_buffer_rsrc_t rsrc, char* dst, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void bw6f_wait_barrier()
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__global__ void __launch_bounds__(bw6f::NT, bw6f::WG) bneck64w8_kernel(const ConvArgs a)
{
    using namespace bw6f;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const xt = smem;                                    // x tile, then u in place
    char* const ring = smem + TILE_BYTES;
    float* const lb1 = reinterpret_cast<float*>(ring + NST * SLAB);      // cv1 bias [64], multipliers [64]
    float* const lm1 = lb1 + C;
    float* const lb2 = lm1 + C;                                          // cv2 bias [64], multipliers [64]
    float* const lm2 = lb2 + C;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int hc = wave & 1, pg = wave >> 1;                  // channel half (32 channels), pixel group (tile rows 4 pg .. 4 pg + 3)
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

    // weight DMA: a slab [64 rows][128 B] is 8 pieces of 1 KB = 8 rows; this wave issues pieces 2 wave, 2 wave + 1; lane -> row, stored chunk
    // lane & 7 = source chunk (lane & 7) ^ ((row >> 1) & 7); (fragment j, MFMA row r) -> channel (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3).
    // W1 has 64 bytes of K: the second half of its rows is the packing's zero padding (Kpad = 128); tap 8's second half is padding too.
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
    // slab of in-tile step s (0: W1; 1 .. 4: taps 2 s - 2, 2 s - 1; 5: tap 8) into ring stage s & 1
    auto issue_slab = [&](int s) {
        char* const dst = ring + (s & (NST - 1)) * SLAB + wave * 2048;
        if (s == 0) {
#pragma unroll
            for (int q = 0; q < 2; ++q) bw6f_dma16(w1rsrc, dst + q * 1024, wrel1[q], 0);
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) bw6f_dma16(w2rsrc, dst + q * 1024, wrel2[q], (s - 1) * 128);
        }
    };
    auto decode_tile = [&](int t, int& bimg, int& y0, int& x0) {
        const int tx = t % tiles_x;
        const int q = t / tiles_x;
        bimg = q / tiles_y;
        y0 = (q - bimg * tiles_y) * TH;
        x0 = tx * TW;
    };
    // x tile DMA: this wave fills plane `wave` (bytes 16 wave .. 16 wave + 15 of every pixel); in piece b lane -> pixel slot b * 64 + lane; outside the
    // image: offset -1 -> the range check writes zeros
    auto issue_x = [&](int bimg, int y0, int x0) {
        const int base = ((bimg * a.H + y0 - 1) * a.W + x0 - 1) * pix_b + wave * 16;
        int ln = lane;
        asm volatile("" : "+v"(ln));                          // (opaque: the per-lane part is recomputed per tile, not kept in registers)
#pragma unroll
        for (int b = 0; b < XDMA; ++b) {
            const int p = b * 64 + ln;
            const int hy = (p * 3641) >> 16, hx = p - hy * HWD;            // p / 18
            const bool ok = p < NHP && (unsigned)(y0 - 1 + hy) < (unsigned)a.H && (unsigned)(x0 - 1 + hx) < (unsigned)a.W;
            const int off = ok ? base + (hy * a.W + hx) * pix_b : -1;
            if (b < XDMA - 1 || ln < XLAST) bw6f_dma16(irsrc, xt + wave * PL + b * 1024, off, 0);      // the last piece ends at slot 183
        }
    };

    const int arow = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);       // weight fragment: row fr of a fragment, first 64 bytes of K (second: ^ 64)
    // 8 consecutive channels c0 .. c0 + 7 (c0 a multiple of 8) of pixel slot p: plane c0 >> 4, half (c0 >> 3) & 1 of its 16 bytes
    auto piece8 = [&](int p, int c0) -> int { return (c0 >> 4) * PL + p * 16 + ((c0 >> 3) & 1) * 8; };

    int bimg, y0, x0;
    decode_tile(tile, bimg, y0, x0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the bias / multiplier writes above
    issue_slab(0);
    issue_x(bimg, y0, x0);

    for (;;) {
        const int next = tile + tstep;
        const bool has_next = next < tend;
        Out8<fp8_t>::raw_t resv[4];                           // residual x of this lane's 4 output vectors (8 channels each)

        // ---------------- step 0: cv1 on this wave's halo fragments wave, wave + 4, wave + 8 ----------------
        {
            int frq = fr, fqq = fq;                           // (opaque per tile)
            asm volatile("" : "+v"(frq), "+v"(fqq));
            bw6f_wait_barrier();                              // x tile and W1 have landed
            issue_slab(1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pc = (4 * pg + i + 1) * HWD + 1 + frq;
                resv[i].a = *reinterpret_cast<const u32x2_t*>(xt + piece8(pc, 32 * hc + 8 * fqq));
            }
            u32x4_t xf[3];                                    // this lane's 16 bytes (K-group fq) of its pixel of every fragment
            int pst[3];
            bool inside[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int p = (wave + NW * i) * 16 + frq;     // slots 180 .. 191 of the last fragment hold no pixel: clamp, never stored
                const int pc = p < XPIX ? p : XPIX - 1;
                xf[i] = *reinterpret_cast<const u32x4_t*>(xt + fqq * PL + pc * 16);
                const int hy = (p * 3641) >> 16, hx = p - hy * HWD;
                inside[i] = p < NHP && (unsigned)(y0 - 1 + hy) < (unsigned)a.H && (unsigned)(x0 - 1 + hx) < (unsigned)a.W;
                pst[i] = p < NHP ? p : -1;
            }
            f32x4_t au[4][3];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i) au[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            u32x4_t wq[2];                                    // the weight fragment of group g + 1 is read before the MFMAs of group g
#pragma unroll
            for (int g = 0; g < 4 + 1; ++g) {
                if (g < 4) wq[g & 1] = *reinterpret_cast<const u32x4_t*>(ring + g * 2048 + arow);
                if (g >= 1) {
                    const int q = g - 1;
#pragma unroll
                    for (int i = 0; i < 3; ++i) S1<fp8_t>::mma(wq[q & 1], xf[i], au[q][i]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(au[j][i]));       // (pinned: k_bneck_w8.hip explains)
            // every wave has its x fragments and residual bytes in registers behind this barrier: the tile may be rewritten
            bw6f_wait_barrier();
            // u = e4m3(SiLU(acc * m1 + b1) / s_u), back into the tile in place: 32-channel group s of fragment i
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int nl = s * 32 + fqq * 8;
                    const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(lb1 + nl), b1 = *reinterpret_cast<const f32x4_t*>(lb1 + nl + 4);
                    const f32x4_t m0 = *reinterpret_cast<const f32x4_t*>(lm1 + nl), m1 = *reinterpret_cast<const f32x4_t*>(lm1 + nl + 4);
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {                // conv_stream_kernel's fp8 epilogue, operation by operation
                        v[e] = au[2 * s][i][e] * m0[e] + b0[e];
                        v[4 + e] = au[2 * s + 1][i][e] * m1[e] + b1[e];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = S1<fp8_t>::silu(v[e]);
                    Out8<fp8_t>::raw_t o = Out8<fp8_t>::pack(v, a.c1_out_inv_scale);
                    if (!inside[i]) o.a = u32x2_t{0u, 0u};
                    if (pst[i] >= 0) *reinterpret_cast<u32x2_t*>(xt + piece8(pst[i], nl)) = o.a;
                }
        }

        // ---------------- steps 1 .. 5: the 3x3 over u ----------------
        f32x4_t acc[2][4];                                    // fragment j = channels 32 hc + fq * 8 + j * 4 ..
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        int pb;                                               // pixel (tile row 4 pg, column fr) of plane fq: everything else is an immediate offset
        {
            int frt = fr;
            asm volatile("" : "+v"(frt));
            pb = fq * PL + ((4 * pg) * HWD + frt) * 16;
        }
        int nb = 0, ny0 = 0, nx0 = 0;
        if (has_next) decode_tile(next, nb, ny0, nx0);
        auto tap_frag = [&](int tap, int i) -> u32x4_t {
            const int ky = tap / 3, kx = tap - ky * 3;
            return *reinterpret_cast<const u32x4_t*>(xt + pb + ((i + ky) * HWD + kx) * 16);
        };
        auto wfrag = [&](int st, int kk, int j) -> u32x4_t {
            return *reinterpret_cast<const u32x4_t*>(ring + (st & (NST - 1)) * SLAB + (2 * hc + j) * 2048 + (kk ? arow ^ 64 : arow));
        };
#pragma unroll
        for (int s = 1; s < NSTEP; ++s) {
            bw6f_wait_barrier();                              // slab s has landed everywhere, step s - 1 is over everywhere (s = 1: u is complete)
            if (s + 1 < NSTEP) issue_slab(s + 1);
            else if (has_next) issue_slab(0);
            if (s < NSTEP - 1) {                              // taps 2 s - 2 and 2 s - 1 in one instruction (the narrow halo kernel's pairing of K-steps)
                u32x4_t pf[2][4], wq[2][2];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) wq[kk][j] = wfrag(s, kk, j);
#pragma unroll
                    for (int i = 0; i < 4; ++i) pf[kk][i] = tap_frag(2 * s - 2 + kk, i);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) fp8_mma128(wq[0][j], wq[1][j], pf[0][i], pf[1][i], acc[j][i]);
            } else {                                          // tap 8: the 16x16x32 pair
                u32x4_t pf[4], wq[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) wq[j] = wfrag(s, 0, j);
#pragma unroll
                for (int i = 0; i < 4; ++i) pf[i] = tap_frag(8, i);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) S1<fp8_t>::mma(wq[j], pf[i], acc[j][i]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(acc[j][i]));      // (pinned per step)
            __builtin_amdgcn_sched_barrier(0);
        }

        // every wave is done with u before the next tile's x lands on it
        bw6f_wait_barrier();
        if (has_next) issue_x(nb, ny0, nx0);
        // ---------------- epilogue: acc * m2 + b2, SiLU, + x, e4m3, 8-byte stores ----------------
        int fre = fr, fqe = fq;
        asm volatile("" : "+v"(fre), "+v"(fqe));
        const bool colok = x0 + fre < a.W;
        const int off0 = ((bimg * a.H + y0 + 4 * pg) * a.W + x0 + fre) * a.ldo + 32 * hc + 8 * fqe;
        const int nl = 32 * hc + fqe * 8;
        const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(lb2 + nl), b1 = *reinterpret_cast<const f32x4_t*>(lb2 + nl + 4);
        const f32x4_t m0 = *reinterpret_cast<const f32x4_t*>(lm2 + nl), m1 = *reinterpret_cast<const f32x4_t*>(lm2 + nl + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = colok && y0 + 4 * pg + i < a.H;
            // masked lanes: offset 0x80000000 stays out of range (the constant goes into the vector offset / immediate, never into soffset:
            // DESIGN.md section 3, store-data hazard)
            const int ooff = ok ? off0 + i * a.W * a.ldo : (int)0x80000000;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {                        // tile_epilogue's fp8 arithmetic (k_conv_halo.hip), operation by operation
                v[e] = acc[0][i][e] * m0[e] + b0[e];
                v[4 + e] = acc[1][i][e] * m1[e] + b1[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = S1<fp8_t>::silu(v[e]);
            if (a.c1_res) Out8<fp8_t>::add(resv[i], v, a.res_scale);
            Out8<fp8_t>::store(Out8<fp8_t>::pack(v, a.out_inv_scale), orsrc, ooff);
        }
        if (!has_next) break;
        tile = next; bimg = nb; y0 = ny0; x0 = nx0;
    }
}

// plan-time question (c1_w may not be set yet): would this cv1 + 3x3 pair of the fp8 engine run on the kernel?
bool bneck64w8_shape_ok(const ConvArgs& a)
{
    const int th = (a.H + bw6f::TH - 1) / bw6f::TH, tw = (a.W + bw6f::TW - 1) / bw6f::TW;
    const double cover = (double)a.H * a.W / ((double)th * tw * (bw6f::TH * bw6f::TW));
    if (!(a.opts & OPT_HALO_FORCE) && cover < 0.75) return false;          // partially filled tiles waste matrix work
    return a.ks == 3 && a.stride == 1 && a.pad == 1 && a.Cin == bw6f::C && a.Cout == bw6f::C && a.c1_Kpad >= 128 &&
           a.Kpad >= 9 * bw6f::C + 64 && a.ldi % 16 == 0 && a.ldo % 8 == 0 && a.in_bytes != 0 && a.H >= 1 && a.W >= 1 && a.act == ACT_SILU &&
           a.out_bytes != 0 && !a.head && !a.up2 && !a.out_f32 && !a.src_mode && !a.f2_w && !a.res && (a.out_dt < 0 || a.out_dt == 2) &&
           !(a.opts & (OPT_HALO_OFF | OPT_NO_FUSE_CV1 | OPT_NO_BNECK64W));
}

hipError_t launch_bneck64w8(const ConvArgs& a0, hipStream_t s)
{
    if (!a0.c1_w || !bneck64w8_shape_ok(a0)) return hipErrorNotSupported;
    ConvArgs a = a0;
    a.dbg = 0;
    static size_t attr[16] = {0};
    {
        const hipError_t e = ensure_lds_attr(reinterpret_cast<const void*>(bneck64w8_kernel), bw6f::LDS_BYTES, a.device, attr);
        if (e != hipSuccess) return e;
    }
    const int ntile = a.B * ((a.H + bw6f::TH - 1) / bw6f::TH) * ((a.W + bw6f::TW - 1) / bw6f::TW);
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
    const int slots = bw6f::WG * n_cu;
    const int gx = ntile < slots ? ntile : slots;
    hipLaunchKernelGGL(bneck64w8_kernel, dim3(gx), dim3(bw6f::NT), bw6f::LDS_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace sky
