// Deep-pipelined 3x3 convolution (stride 1, pad 1) for wide inputs, bf16 engine: cv2 of the BottleneckBlocks with 256 and more
// channels (reference blocks.py:69-90: x + cv2(cv1(x)); ConvolutionBlock blocks.py:10-41) -- the 80 x 80 bottlenecks of skyeye_l
// (256 channels, K = 2 304), which carry a quarter of that graph's time on the halo-tile kernel.
//
// The halo-tile kernel (k_conv_halo.hip) runs a two-stage weight ring with s_waitcnt vmcnt(0) + barrier per tap in two workgroups
// per CU: the structure the guide calls the ~900 TFLOP/s ceiling.  For these layers the matrix work per output value (K / 32 MFMAs
// per 4 values) is several times the activation arithmetic, so one workgroup of 8 waves per CU with the tap machinery of the
// bottleneck kernel (k_bneck.hip) pays:
//   * a work item = a 16 x 16 output tile x 128 output channels; wave (pixel group pg, channel half hc) owns tile rows 4 pg ..
//     4 pg + 3 x 64 channels (4 x 4 accumulator fragments); a step = one (chunk, tap) weight slab [128 rows][128 B] = 32 MFMAs per
//     wave;
//   * the 18 x 18 halo tile of one 128-byte channel chunk lives in one of TWO LDS images (the halo kernels' layout: conflict-free B
//     fragments for any tap); chunk c + 1 (or the next work item's chunk 0) arrives by LDS-DMA during the first three taps of
//     chunk c, two pieces per wave and step;
//   * the weight slabs stream through a FOUR-stage ring that runs continuously across chunks and work items; the waits are counted
//     (s_waitcnt vmcnt(N): behind the slab the wait is for, this wave has issued N younger pieces / loads / stores), the barrier is the
//     raw s_barrier; at the barrier of step s slab s + 1 is complete, so the first fragments of step s + 1 are requested under the
//     MFMAs of step s -- also across the epilogue into the next work item;
//   * the residual vectors are requested four taps before the end of the item; bias, SiLU, + residual (uncontracted, as in the
//     halo-tile kernel), bf16, 16-byte stores.
// K order (chunk, tap, 64-byte K-step) and every rounding are those of the halo-tile kernel with 128-channel tiles: the result is
// bit-identical to it (tests/test_gpu_conv_deep.py).
#include "sky_kernels.h"

#include "conv_frag.h"

namespace sky {

namespace dp {
constexpr int NW = 8, NT = NW * 64;
constexpr int TS = 16, HWD = TS + 2, NHP = HWD * HWD;         // 324 halo pixels
constexpr int XPIX = 352, PL = XPIX * 32, CHB = 4 * PL;       // pixel slots per plane, bytes per plane (44 * 256), per chunk image
constexpr int XDMA = XPIX / 32;                               // DMA pieces per plane (11)
constexpr int NB = 128;                                       // output channels of a work item
constexpr int SLAB = NB * 128;                                // one weight slab [128 rows][128 B]
constexpr int NST = 4;                                        // ring stages
constexpr int GC = 4, GSTEP = 9 * GC;                         // chunks per pass of the unrolled body, its steps (36)
constexpr int MAXC = 1024;                                    // bias staged for up to this many output channels
constexpr int LDS_BYTES = 2 * CHB + NST * SLAB + MAXC * 4;
static_assert(GSTEP % NST == 0 && GC % 2 == 0, "ring stage and halo image of a step must not depend on the pass");
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
}  // namespace dp

__device__ __forceinline__ void dp_dma16(__amdgpu_buffer_rsrc_t rsrc, char* dst, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
}
// the raw barrier (no queue drain); the empty asm keeps the compiler from moving LDS accesses across it
__device__ __forceinline__ void dp_barrier()
{
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
template <int N>
__device__ __forceinline__ void dp_wait_vm()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}
// the count as a value (folds after unrolling where it is a constant; a uniform switch otherwise)
__device__ __forceinline__ void dp_wait_n(int n)
{
    switch (n) {
    case 2: dp_wait_vm<2>(); break;
    case 3: dp_wait_vm<3>(); break;
    case 4: dp_wait_vm<4>(); break;
    case 5: dp_wait_vm<5>(); break;
    case 6: dp_wait_vm<6>(); break;
    case 10: dp_wait_vm<10>(); break;
    case 12: dp_wait_vm<12>(); break;
    default: dp_wait_vm<0>(); break;
    }
}

__global__ void __launch_bounds__(dp::NT) conv3x3_deep_kernel(const ConvArgs a)
{
    using namespace dp;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const himg = smem;                                  // two halo images (chunk parity)
    char* const ring = smem + 2 * CHB;
    float* const lbias = reinterpret_cast<float*>(ring + NST * SLAB);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int hc = wave & 1, pg = wave >> 1;                  // channel half, pixel group (tile rows 4 pg .. 4 pg + 3)
    const int tiles_x = (a.W + TS - 1) / TS, tiles_y = (a.H + TS - 1) / TS;
    const int gy = a.Cout / NB;
    const int nitem = a.B * tiles_y * tiles_x * gy;
    int item, tstep, tend;                                    // XCD-aware item order (conv_frag.h: tile_walk): the N tiles of a pixel tile stay neighbours
    tile_walk(nitem, item, tstep, tend);
    if (item >= tend) return;
    const int pix_b = a.ldi * 2;
    const int Cb = a.Cin * 2;                                 // bytes of one tap of K
    const int NG = (Cb >> 7) / GC;                            // passes of the unrolled body (Cin = 256: 1)
    const int wpitch = a.Kpad * 2;
    const bool has_res = a.res != nullptr;

    for (int i = tid; i < a.Cout; i += NT) lbias[i] = a.bias[i];

    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, (int)a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)a.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(has_res ? a.res : a.out), 0, (int)(has_res ? a.res_bytes : a.out_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, (int)((long)a.Cout * wpitch), 0x00020000);

    // weight DMA: a slab is 16 pieces of 8 rows; this wave issues pieces 2 wave, 2 wave + 1.  lane -> LDS row, stored chunk
    // lane & 7 = source chunk (lane & 7) ^ ((row >> 1) & 7); (fragment j, MFMA row r) -> channel (j>>1)*32 + (r>>2)*8 + (j&1)*4 + (r&3)
    int wrel[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = (wave * 2 + q) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int j = row >> 4, r = row & 15;
        const int ch = (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3);
        wrel[q] = ch * wpitch + c * 16;
    }
    // slab (chunk, tap) of output channels n0 .. n0 + 127 into ring stage `stage`
    auto issue_slab = [&](int n0, int chunk, int tap, int stage) {
        char* const dst = ring + stage * SLAB + wave * 2048;
        const int soff = n0 * wpitch + tap * Cb + chunk * 128;
#pragma unroll
        for (int q = 0; q < 2; ++q) dp_dma16(wrsrc, dst + q * 1024, wrel[q], soff);
    };
    auto decode_item = [&](int it, int& bimg, int& y0, int& x0, int& n0) {
        const int t = it / gy;
        n0 = (it - t * gy) * NB;
        const int tx = t % tiles_x;
        const int q = t / tiles_x;
        bimg = q / tiles_y;
        y0 = (q - bimg * tiles_y) * TS;
        x0 = tx * TS;
    };
    // halo DMA of one chunk: a wave fills pieces [b0, b1) of plane wave & 3 of image `img`; in piece b lane -> pixel slot
    // p = b*32 + (lane >> 1), 16-byte half lane & 1 = K-step (lane & 1) ^ (p >> 3 & 1); outside the image: offset -1 -> zeros
    auto issue_halo = [&](int bimg, int y0, int x0, int chunk, int img, int b0, int b1) {
        const int base = ((bimg * a.H + y0 - 1) * a.W + x0 - 1) * pix_b + chunk * 128 + (wave & 3) * 16;
        char* const dst = himg + img * CHB + (wave & 3) * PL;
        int ln = lane;                                        // (opaque: the per-lane slot arithmetic below is loop invariant, ~25 registers)
        asm volatile("" : "+v"(ln));
        for (int b = b0; b < b1; ++b) {                       // (uniform bounds: one to three pieces)
            const int p = b * 32 + (ln >> 1);
            const int hy = (p * 3641) >> 16, hx = p - hy * HWD;            // p / 18
            const int kk = (ln & 1) ^ ((p >> 3) & 1);
            const bool ok = p < NHP && (unsigned)(y0 - 1 + hy) < (unsigned)a.H && (unsigned)(x0 - 1 + hx) < (unsigned)a.W;
            dp_dma16(irsrc, dst + b * 1024, ok ? base + (hy * a.W + hx) * pix_b + kk * 64 : -1, 0);
        }
    };
    // pieces of the next chunk's halo this wave issues at tap q of a chunk (behind that step's slab): 44 pieces per chunk = 6 for each
    // of waves 0..3 (pieces 0 .. 5 of their plane), 5 for waves 4..7 (pieces 6 .. 10), two per step
    auto nhalo = [&](int q) { return q < 2 ? 2 : q == 2 ? (wave < 4 ? 2 : 1) : 0; };

    // fragment addresses: weight fragment row fr, K-step 0 (K-step 1: ^ 64); pixel fragment of tile row 4 pg + r (r = i + ky), column
    // shift kx: slot = (4 pg + r) * 18 + fr + kx, address = plane + slot * 32 + 16 * bit 3 of the slot (K-step 1: ^ 16); 18 = 16 + 2, so
    // that bit is bit 3 of fr + 8 pg + c with c = 2 r + kx: thirteen per-lane bases, the rest is an immediate
    const int arow = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);
    int pb[13];
#pragma unroll
    for (int c = 0; c < 13; ++c) pb[c] = fq * PL + ((4 * pg) * HWD + fr) * 32 + (((fr + 8 * pg + c) & 8) << 1);
    auto tap_frag = [&](int k, int i, int kk) -> u32x4_t {   // step k of the pass: chunk image (k / 9) & 1, tap k % 9
        const int cc = k / 9, tap = k - cc * 9;
        const int ky = tap / 3, kx = tap - ky * 3;
        int q = pb[2 * (i + ky) + kx];
        if (kk) {
            asm volatile("" : "+v"(q));                       // (opaque: else hipcc keeps the 13 ^ 16 variants of pb[] in registers as well)
            q ^= 16;
        }
        return *reinterpret_cast<const u32x4_t*>(himg + q + ((cc & 1) * CHB + ((i + ky) * HWD + kx) * 32));
    };
    auto wfrag = [&](int k, int kk, int sp, int h) -> u32x4_t {
        return *reinterpret_cast<const u32x4_t*>(ring + (k & (NST - 1)) * SLAB + (4 * hc + 2 * sp + h) * 2048 + (kk ? arow ^ 64 : arow));
    };

    int bimg, y0, x0, n0;
    decode_item(item, bimg, y0, x0, n0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the bias writes above
    issue_slab(n0, 0, 0, 0);
    issue_slab(n0, 0, 1, 1);
    issue_slab(n0, 0, 2, 2);
    issue_halo(bimg, y0, x0, 0, 0, wave < 4 ? 0 : 6, wave < 4 ? 6 : XDMA);
    bool first = true;

    u32x4_t pf0[2][4], wq01[2][2][2];                         // [step parity]: K-step 0 pixel fragments / weight pairs, requested a step ahead
    for (;;) {
        const int next = item + tstep;
        const bool has_next = next < tend;
        int nb = 0, ny0 = 0, nx0 = 0, nn0 = 0;
        if (has_next) decode_item(next, nb, ny0, nx0, nn0);
        Out8<__bf16>::raw_t resv[4][2];
        f32x4_t acc[4][4];                                    // start from the bias (k_conv_halo.hip: acc_start): fragment j = channels n0 + 64 hc + (j >> 1) * 32 + fq * 8 + (j & 1) * 4 ..
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4_t c = *reinterpret_cast<const f32x4_t*>(lbias + n0 + 64 * hc + (j >> 1) * 32 + fq * 8 + (j & 1) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = c;
        }

        for (int cg = 0; cg < NG; ++cg) {
            const bool last_pass = cg == NG - 1;
#pragma unroll
            for (int k = 0; k < GSTEP; ++k) {
                const int p = k & 1;
                const int cc = k / 9, q = k - cc * 9;
                // is there a chunk behind the one of step k (this item's next chunk or the next item's first)?  (uniform)
                auto more_at = [&](int kk_) { return !(last_pass && kk_ / 9 == GC - 1) || has_next; };
                // wait: slab k + 1 (requested in step k - 2) must have landed; behind it this wave issued the halo pieces of step k - 2,
                // slab k + 2, the halo pieces of step k - 1 -- plus the epilogue's eight stores in the first two steps of an item and the
                // eight residual loads behind step GSTEP - 5 of the last pass
                int n = 2;
                if (k >= 2) n += more_at(k - 2) ? nhalo((k - 2) % 9) : 0;
                if (k >= 1) n += more_at(k - 1) ? nhalo((k - 1) % 9) : 0;
                if (k < 2 && cg == 0 && !first) n += 8;
                if ((k == GSTEP - 4 || k == GSTEP - 3) && last_pass && has_res) n += 8;
                if (k >= GSTEP - 2 && last_pass && !has_next) n = 0;      // nothing behind the last slabs
                if (k == 0 && cg == 0 && first) n = 0;
                dp_wait_n(n);
                dp_barrier();
                if (k == 0 && cg == 0 && first) {             // nothing was requested a step ahead of the very first step
#pragma unroll
                    for (int i = 0; i < 4; ++i) pf0[p][i] = tap_frag(k, i, 0);
#pragma unroll
                    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                        for (int h = 0; h < 2; ++h) wq01[p][sp][h] = wfrag(k, 0, sp, h);
                }
                // the DMA requests of step k: slab k + 3 (this pass, the next pass or the next item), then halo pieces of the next chunk
                auto step_dma = [&]() {
                    const int k3 = k + 3;
                    if (k3 < GSTEP) issue_slab(n0, cg * GC + k3 / 9, k3 % 9, k3 & (NST - 1));
                    else if (!last_pass) issue_slab(n0, (cg + 1) * GC, k3 - GSTEP, k3 & (NST - 1));
                    else if (has_next) issue_slab(nn0, 0, k3 - GSTEP, k3 & (NST - 1));
                    if (q < 3 && more_at(k)) {
                        const int lo = (wave < 4 ? 0 : 6) + 2 * q, hi = lo + nhalo(q);
                        const bool same = !(last_pass && cc == GC - 1);                          // the next chunk belongs to this item
                        issue_halo(same ? bimg : nb, same ? y0 : ny0, same ? x0 : nx0, same ? cg * GC + cc + 1 : 0, (cc + 1) & 1, lo, hi);
                    }
                };
                u32x4_t pf1[4], wq2[2], wq3[2];
#pragma unroll
                for (int i = 0; i < 4; ++i) pf1[i] = tap_frag(k, i, 1);
#pragma unroll
                for (int h = 0; h < 2; ++h) wq2[h] = wfrag(k, 1, 0, h);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) S1<__bf16>::mma(wq01[p][0][h], pf0[p][i], acc[h][i]);
                if (wave >= 4) step_dma();                    // (the two waves of a SIMD issue theirs at different points of the step)
#pragma unroll
                for (int h = 0; h < 2; ++h) wq3[h] = wfrag(k, 1, 1, h);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) S1<__bf16>::mma(wq01[p][1][h], pf0[p][i], acc[2 + h][i]);
                if (wave < 4) step_dma();
                if (k == GSTEP - 5 && last_pass && has_res) {
                    // residual vectors of this lane's 4 x 2 output vectors: in flight during the last four taps
                    int fre = fr, fqe = fq;
                    asm volatile("" : "+v"(fre), "+v"(fqe));
                    const bool colok = x0 + fre < a.W;
                    const int roff0 = (((bimg * a.H + y0 + 4 * pg) * a.W + x0 + fre) * a.ldr + n0 + 64 * hc + 8 * fqe) * 2;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int roff = (colok && y0 + 4 * pg + i < a.H) ? roff0 + i * a.W * a.ldr * 2 : -1;
#pragma unroll
                        for (int sp = 0; sp < 2; ++sp) resv[i][sp] = Out8<__bf16>::load(rrsrc, roff, sp * 64);
                    }
                }
                // K-step 0 fragments of the next step (its slab is complete and visible since this step's barrier; the halo image of the
                // next chunk / item has been complete for several steps)
                const bool nxt = k + 1 < GSTEP || !last_pass || has_next;
                if (nxt) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) pf0[p ^ 1][i] = tap_frag((k + 1) % GSTEP, i, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) S1<__bf16>::mma(wq2[h], pf1[i], acc[h][i]);
                if (nxt) {
#pragma unroll
                    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                        for (int h = 0; h < 2; ++h) wq01[p ^ 1][sp][h] = wfrag((k + 1) % GSTEP, 0, sp, h);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) S1<__bf16>::mma(wq3[h], pf1[i], acc[2 + h][i]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---------------- epilogue: bias, SiLU, + residual, bf16, 16-byte stores (the ring and the halo images are not touched) ----------------
        {
            int fre = fr, fqe = fq;
            asm volatile("" : "+v"(fre), "+v"(fqe));
            const bool colok = x0 + fre < a.W;
            const int off0 = (((bimg * a.H + y0 + 4 * pg) * a.W + x0 + fre) * a.ldo + n0 + 64 * hc + 8 * fqe) * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = colok && y0 + 4 * pg + i < a.H;
                // masked lanes: offset 0x80000000 stays out of range after the immediate is added (constants go into the vector offset /
                // immediate, never into soffset: DESIGN.md section 3, store-data hazard)
                const int ooff = ok ? off0 + i * a.W * a.ldo * 2 : (int)0x80000000;
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float xx = e < 4 ? acc[2 * sp][i][e] : acc[2 * sp + 1][i][e - 4];      // (the bias was the accumulators' initial value)
                        v[e] = a.act == ACT_SILU ? S1<__bf16>::silu(xx) : xx;
                        if (has_res) {
                            // multiply and residual add round separately, as in the halo-tile kernel's 128-channel epilogue
#pragma clang fp contract(off)
                            const unsigned rw = resv[i][sp].a[e >> 1];
                            const float res = (e & 1) ? __uint_as_float(rw & 0xffff0000u) : __uint_as_float(rw << 16);
                            v[e] = v[e] + res;
                        }
                    }
                    Out8<__bf16>::store(Out8<__bf16>::pack(v, 1.0f), orsrc, ooff + sp * 64);
                }
            }
        }
        if (!has_next) break;
        item = next; bimg = nb; y0 = ny0; x0 = nx0; n0 = nn0;
        first = false;
    }
}

// would this convolution run on the kernel?  bf16, 3x3 stride 1, Cin a multiple of 256 (passes of four 128-byte chunks), Cout a
// multiple of 128, maps that fill their 16 x 16 tiles
bool conv3x3_deep_ok(int dtype, const ConvArgs& a)
{
    if (dtype != 1 || (a.out_dt >= 0 && a.out_dt != 1)) return false;
    if (a.ks != 3 || a.stride != 1 || a.pad != 1 || a.head || a.up2 || a.out_f32 || a.src_mode || a.f2_w || a.c1_w) return false;
    if (a.Cin % 256 != 0 || a.Cin < 256 || a.Cout % dp::NB != 0 || a.Cout > dp::MAXC || (a.act != ACT_SILU && a.act != ACT_NONE)) return false;
    if (a.ldi % 8 != 0 || a.ldo % 8 != 0 || (a.res && a.ldr % 8 != 0)) return false;
    if (a.in_bytes == 0 || a.out_bytes == 0 || (a.res && a.res_bytes == 0)) return false;
    if ((long)a.Kpad < 9L * a.Cin || (long)a.Cout * a.Kpad * 2 >= (1L << 31)) return false;
    if (a.opts & (OPT_HALO_OFF | OPT_NO_DEEP3X3)) return false;
    const int th = (a.H + dp::TS - 1) / dp::TS, tw = (a.W + dp::TS - 1) / dp::TS;
    const double cover = (double)a.H * a.W / ((double)th * tw * 256.0);
    return (a.opts & OPT_HALO_FORCE) || cover >= 0.75;
}

hipError_t launch_conv3x3_deep(int dtype, const ConvArgs& a, hipStream_t s)
{
    if (!conv3x3_deep_ok(dtype, a)) return hipErrorNotSupported;
    static size_t attr[16] = {0};
    {
        const hipError_t e = ensure_lds_attr(reinterpret_cast<const void*>(conv3x3_deep_kernel), dp::LDS_BYTES, a.device, attr);
        if (e != hipSuccess) return e;
    }
    const int nitem = a.B * ((a.H + dp::TS - 1) / dp::TS) * ((a.W + dp::TS - 1) / dp::TS) * (a.Cout / dp::NB);
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
    const int gx = nitem < n_cu ? nitem : n_cu;
    hipLaunchKernelGGL(conv3x3_deep_kernel, dim3(gx), dim3(dp::NT), dp::LDS_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace sky
