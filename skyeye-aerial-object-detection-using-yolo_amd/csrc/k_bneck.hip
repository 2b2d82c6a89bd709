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
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
    else static_assert(N == 0, "add the count");
}

__global__ void __launch_bounds__(bk::NT) bneck128_kernel(const ConvArgs a)
{
    using namespace bk;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const xt = smem;                                    // x tile, then u in place
    char* const ring = smem + TILE_BYTES;
    float* const lb1 = reinterpret_cast<float*>(ring + NST * SLAB);      // cv1 bias [128]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int hc = wave & 1, pg = wave >> 1;                  // channel half, pixel group (tile rows 4 pg .. 4 pg + 3)
    const int tiles_x = (a.W + TS - 1) / TS, tiles_y = (a.H + TS - 1) / TS;
    const int ntile = a.B * tiles_y * tiles_x;
    int tile = blockIdx.x;
    if (tile >= ntile) return;
    const int pix_b = a.ldi * 2;
    const int w1pitch = a.c1_Kpad * 2, w2pitch = a.Kpad * 2;

    float* const lb2 = lb1 + C;                                          // cv2 bias [128]
    for (int i = tid; i < C; i += NT) { lb1[i] = a.c1_bias[i]; lb2[i] = a.bias[i]; }

    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, (int)a.in_bytes, 0x00020000);
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
        wrel1[q] = ch * w1pitch + c * 16;
        wrel2[q] = ch * w2pitch + c * 16;
    }
    // slab of in-tile step s (0, 1: W1 K chunks; 2 ..: W2 (chunk, tap)) into ring stage s & 3
    auto issue_slab = [&](int s) {
        char* const dst = ring + (s & (NST - 1)) * SLAB + wave * 2048;
        if (s < NCH) {
#pragma unroll
            for (int q = 0; q < 2; ++q) bk_dma16(w1rsrc, dst + q * 1024, wrel1[q], s * 128);
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
    // x tile DMA: wave w fills plane w & 3 of chunk w >> 2 (11 pieces); in piece b lane -> pixel slot p = b*32 + (lane >> 1),
    // 16-byte half lane & 1 = K-step (lane & 1) ^ (p >> 3 & 1); outside the image: offset -1 -> the range check writes zeros
    auto issue_x = [&](int bimg, int y0, int x0) {
        const int base = ((bimg * a.H + y0 - 1) * a.W + x0 - 1) * pix_b + (wave >> 2) * 128 + (wave & 3) * 16;
        char* const dst = xt + (wave >> 2) * CHB + (wave & 3) * PL;
        // (opaque: everything below that depends on the lane alone is a tile-loop invariant hipcc would keep in ~25 registers)
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int b = 0; b < XDMA; ++b) {
            const int p = b * 32 + (ln >> 1);
            const int hy = (p * 3641) >> 16, hx = p - hy * HWD;            // p / 18
            const int kk = (ln & 1) ^ ((p >> 3) & 1);
            const bool ok = p < NHP && (unsigned)(y0 - 1 + hy) < (unsigned)a.H && (unsigned)(x0 - 1 + hx) < (unsigned)a.W;
            bk_dma16(irsrc, dst + b * 1024, ok ? base + (hy * a.W + hx) * pix_b + kk * 64 : -1, 0);
        }
    };

    // fragment addresses
    const int arow = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);       // weight fragment: row fr of a fragment, K-step 0 (K-step 1: ^ 64)
    const int prow0 = fq * PL + ((4 * pg) * HWD + fr) * 32;          // pixel fragments of the taps: tile row 4 pg, tap (0, 0)

    int bimg, y0, x0;
    decode_tile(tile, bimg, y0, x0);
    __builtin_amdgcn_s_waitcnt(0xC07F);                       // lgkmcnt(0): the bias writes above
    issue_slab(0);
    issue_slab(1);
    issue_slab(2);
    issue_x(bimg, y0, x0);
    bool first = true;

    for (;;) {
        const int next = tile + gridDim.x;
        const bool has_next = next < ntile;
        Out8<__bf16>::raw_t resv[4][2];                       // residual x of this lane's 4 x 2 output vectors

        // ---------------- steps 0, 1: cv1 on this wave's halo fragments wave, wave + 8, wave + 16 ----------------
        {
            int frq = fr, fqq = fq;                           // (opaque per tile, see issue_x)
            asm volatile("" : "+v"(frq), "+v"(fqq));
            f32x4_t au[8][3];
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i) au[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NCH; ++s) {
                if (s == 0) { if (first) bk_wait_vm<0>(); else bk_wait_vm<8>(); }
                else { if (first) bk_wait_vm<2>(); else bk_wait_vm<10>(); }
                bk_barrier();
                issue_slab(s + 3);
                if (s == 0) {
                    // residual: piece (plane fq, K-step sp) of chunk hc of the centre pixel of this lane's 4 output pixels
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int pc = (4 * pg + i + 1) * HWD + 1 + frq;
#pragma unroll
                        for (int sp = 0; sp < 2; ++sp)
                            resv[i][sp].a = *reinterpret_cast<const u32x4_t*>(xt + hc * CHB + fqq * PL + pc * 32 + ((sp ^ ((pc >> 3) & 1)) << 4));
                    }
                }
                u32x4_t xf[3][2];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int p = (wave + 8 * i) * 16 + frq;  // (the third fragment of waves 5..7 would start at slot 336+: clamp, never stored)
                    const int pc = p < XPIX ? p : XPIX - 1;
                    const int A = s * CHB + fqq * PL + pc * 32 + (((pc >> 3) & 1) << 4);
                    xf[i][0] = *reinterpret_cast<const u32x4_t*>(xt + A);
                    xf[i][1] = *reinterpret_cast<const u32x4_t*>(xt + (A ^ 16));
                }
                const char* wb = ring + (s & (NST - 1)) * SLAB + arow;
                // 16 (K-step, weight fragment) groups of 3 MFMAs; the weight fragment of group g + 2 is read before the MFMAs of group g
                u32x4_t wq[3];
#pragma unroll
                for (int gq = 0; gq < 16 + 2; ++gq) {
                    if (gq < 16) wq[gq % 3] = *reinterpret_cast<const u32x4_t*>(wb + (gq & 7) * 2048 + ((gq >> 3) ? 64 - 2 * (arow & 64) : 0));
                    if (gq >= 2) {
                        const int gg = gq - 2;
#pragma unroll
                        for (int i = 0; i < 3; ++i) S1<__bf16>::mma(wq[gg % 3], xf[i][gg >> 3], au[gg & 7][i]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // u = SiLU(. + b1) -> bf16 -> back into the tile, in place: this wave's slots are read by nobody else before the next barrier
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (wave + 8 * i >= NFR) break;               // (uniform) waves 5..7 own two real fragments; their third is computed and dropped
                const int p = (wave + 8 * i) * 16 + frq;
                const int hy = (p * 3641) >> 16, hx = p - hy * HWD;
                const bool inside = p < NHP && (unsigned)(y0 - 1 + hy) < (unsigned)a.H && (unsigned)(x0 - 1 + hx) < (unsigned)a.W;
#pragma unroll
                for (int sg = 0; sg < 4; ++sg) {              // 32-channel group sg: chunk sg >> 1, K-step sg & 1
                    const f32x4_t c0 = *reinterpret_cast<const f32x4_t*>(lb1 + sg * 32 + fqq * 8);
                    const f32x4_t c1 = *reinterpret_cast<const f32x4_t*>(lb1 + sg * 32 + fqq * 8 + 4);
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = S1<__bf16>::silu(au[2 * sg][i][e] + c0[e]);
                        v[4 + e] = S1<__bf16>::silu(au[2 * sg + 1][i][e] + c1[e]);
                    }
                    Out8<__bf16>::raw_t o = Out8<__bf16>::pack(v, 1.0f);
                    if (!inside) o.a = u32x4_t{0u, 0u, 0u, 0u};          // the 3x3's zero padding
                    *reinterpret_cast<u32x4_t*>(xt + (sg >> 1) * CHB + fqq * PL + p * 32 + (((sg & 1) ^ ((p >> 3) & 1)) << 4)) = o.a;
                }
            }
        }

        // ---------------- steps 2 .. 19: the 3x3 over u, (chunk, tap) by (chunk, tap) ----------------
        f32x4_t acc[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = NCH; s < NSTEP; ++s) {
            if (s == NSTEP - 2 && !has_next) bk_wait_vm<0>(); else bk_wait_vm<2>();
            bk_barrier();
            if (s + 3 < NSTEP) issue_slab(s + 3);
            else if (has_next) issue_slab(s + 3 - NSTEP);
            const int g = s - NCH, chunk = g / 9, tap = g - chunk * 9;
            const int ky = tap / 3, kx = tap - ky * 3;
            const char* wb = ring + (s & (NST - 1)) * SLAB + (4 * hc) * 2048 + arow;
            // (opaque per step: the 72 fragment addresses of a tile are loop invariants hipcc would otherwise keep in registers)
            int pr = prow0;
            asm volatile("" : "+v"(pr));
            int pa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int A = chunk * CHB + pr + ((i + ky) * HWD + kx) * 32;
                pa[i] = A + (((A >> 8) & 1) << 4);
            }
            u32x4_t pf[2][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) pf[0][i] = *reinterpret_cast<const u32x4_t*>(xt + pa[i]);
            // groups (kk, sp): two weight fragments x four pixel fragments; the weight pair of group g + 2 is read before the MFMAs of group g
            u32x4_t wq[3][2];
#pragma unroll
            for (int gq = 0; gq < 4 + 2; ++gq) {
                if (gq < 4) {
                    const int kk = gq >> 1, sp = gq & 1;
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        wq[gq % 3][h] = *reinterpret_cast<const u32x4_t*>(wb + (2 * sp + h) * 2048 + (kk ? 64 - 2 * (arow & 64) : 0));
                }
                if (gq == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) pf[1][i] = *reinterpret_cast<const u32x4_t*>(xt + (pa[i] ^ 16));
                }
                if (gq >= 2) {
                    const int gg = gq - 2, kk = gg >> 1, sp = gg & 1;
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int i = 0; i < 4; ++i) S1<__bf16>::mma(wq[gg % 3][h], pf[kk][i], acc[2 * sp + h][i]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // every wave is done with u before the next tile's x lands on it
        int nb = 0, ny0 = 0, nx0 = 0;
        bk_barrier();
        if (has_next) {
            decode_tile(next, nb, ny0, nx0);
            issue_x(nb, ny0, nx0);
        }
        // ---------------- epilogue: bias, SiLU, + x, bf16, 16-byte stores ----------------
        int fre = fr, fqe = fq;
        asm volatile("" : "+v"(fre), "+v"(fqe));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int oy = y0 + 4 * pg + i, ox = x0 + fre;
            const bool ok = oy < a.H && ox < a.W;
            char* const op = reinterpret_cast<char*>(a.out) + (((long)bimg * a.H + oy) * a.W + ox) * (long)a.ldo * 2 + (64 * hc + 8 * fqe) * 2;
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                // this lane's channels of fragment pair sp: 64 hc + 32 sp + 8 fq .. + 7
                const f32x4_t c0 = *reinterpret_cast<const f32x4_t*>(lb2 + 64 * hc + 32 * sp + 8 * fqe);
                const f32x4_t c1 = *reinterpret_cast<const f32x4_t*>(lb2 + 64 * hc + 32 * sp + 8 * fqe + 4);
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xx = e < 4 ? acc[2 * sp][i][e] + c0[e] : acc[2 * sp + 1][i][e - 4] + c1[e - 4];
                    v[e] = S1<__bf16>::silu(xx);
                    if (a.c1_res) {
                        // multiply and residual add round separately, as in the 128-channel halo-tile kernel's epilogue (its residual
                        // add sits behind a branch, so hipcc does not contract it; the narrow kernels' epilogue is one fma)
#pragma clang fp contract(off)
                        const unsigned rw = resv[i][sp].a[e >> 1];
                        const float res = (e & 1) ? __uint_as_float(rw & 0xffff0000u) : __uint_as_float(rw << 16);
                        v[e] = v[e] + res;
                    }
                }
                if (ok) Out8<__bf16>::store(Out8<__bf16>::pack(v, 1.0f), op + sp * 64);
            }
        }
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
           !a.head && !a.up2 && !a.out_f32 && !a.src_mode && !a.f2_w && !a.res && !(a.opts & (OPT_HALO_OFF | OPT_NO_FUSE_CV1 | OPT_NO_BNECK128));
}

hipError_t launch_bneck128(const ConvArgs& a, hipStream_t s)
{
    if (!a.c1_w || !bneck128_shape_ok(a)) return hipErrorNotSupported;
    static size_t attr[16] = {0};
    {
        const hipError_t e = ensure_lds_attr(reinterpret_cast<const void*>(bneck128_kernel), bk::LDS_BYTES, a.device, attr);
        if (e != hipSuccess) return e;
    }
    const int ntile = a.B * ((a.H + bk::TS - 1) / bk::TS) * ((a.W + bk::TS - 1) / bk::TS);
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
    const int gx = ntile < n_cu ? ntile : n_cu;
    hipLaunchKernelGGL(bneck128_kernel, dim3(gx), dim3(bk::NT), bk::LDS_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace sky
