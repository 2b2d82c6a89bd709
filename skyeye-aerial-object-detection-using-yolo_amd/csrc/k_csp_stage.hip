// The first CSP stage of the backbone as ONE kernel (bf16 engine):
//     CSPBlock(64, 64, n = 1)   blocks.py:93-123   cv3(cat(bottleneck(cv1(x)), cv2(x)))        (backbone.py:51)
//     BottleneckBlock(32, 32)   blocks.py:69-90    y1 + cv2_3x3(cv1_1x1(y1))
// Run layer by layer the stage is four launches that move the 64-channel 320 x 320 map (419 MB per 32 frames of 1280 x 1280)
// seven times and two 32-channel maps four times: 2.7 GB for 0.84 GB of compulsory traffic, every launch on the load / store
// path.  Here a workgroup owns a 16 x 16 tile of the output and keeps everything between x and cv3's output in LDS / registers:
//   1. the 18 x 18 x 64-channel tile of x (tile + 1 pixel halo) by LDS-DMA, in the halo kernels' plane layout (k_conv_halo.hip);
//   2. on the 21 pixel fragments of the halo tile: y1 = SiLU(W1 x + b1) (K = 64), packed to bf16 -- which IS the B operand of the
//      bottleneck's 1x1 (conv_frag.h: a lane holds 8 consecutive channels of its pixel) --, u = SiLU(Wb1 y1 + bb1) from those
//      registers; y1 and u (zero outside the image: the 3x3's padding) go to LDS tiles [16-byte channel group][pixel];
//      on the 16 tile rows: y2 = SiLU(W2 x + b2) -> LDS;
//   3. barrier; the next tile's x is requested (its latency runs under step 4);
//   4. per wave two tile rows: the 3x3 over u (one 64-byte K-step per tap), bias, SiLU, + y1 (LDS), packed -> K-step 0 of cv3 in
//      registers; K-step 1 = y2 from LDS; cv3, bias, SiLU, 16-byte stores.
// One workgroup of 8 waves per CU, 139 KB of LDS; the four weight sets stay resident as MFMA A fragments ([K-step][row in
// fragment order][64 B], sd-style swizzle).  K order, rounding points (every intermediate is rounded to bf16 exactly where the
// layer-by-layer form stores it) and activation arithmetic equal the four-launch form: the result is bit-identical to it
// (tests/test_gpu_csp_stage.py).  +27 % work on y1 / u (18 x 18 for 16 x 16 pixels).
#include "sky_kernels.h"

#include "conv_frag.h"

namespace sky {

namespace cs {
constexpr int NW = 8, NT = NW * 64;
constexpr int TS = 16, HW = TS + 2, NHP = HW * HW;       // 324 halo pixels
constexpr int C = 64, HD = 32;                           // stage channels, hidden channels
constexpr int XPIX = 352, XPL = XPIX * 32, X_BYTES = 4 * XPL;      // x tile: the 64-channel halo layout, 45 056 B
constexpr int XDMA = XPIX / 32;                                   // DMA instructions per plane (11)
constexpr int HSL = 336, HPLN = HSL * 16, H_BYTES = 4 * HPLN;     // y1 / u tiles: [4 channel groups][336 slots][16 B], 21 504 B
constexpr int CPLN = 256 * 16, C_BYTES = 4 * CPLN;                // y2 tile: centre pixels only, 16 384 B
constexpr int NFR = (NHP + 15) / 16;                              // halo pixel fragments (21)
constexpr int W12_BYTES = 2 * C * 64, WB1_BYTES = 1 * HD * 64, WB2_BYTES = 9 * HD * 64, W3_BYTES = 2 * C * 64;
constexpr int LDS_BYTES = X_BYTES + 2 * H_BYTES + C_BYTES + W12_BYTES + WB1_BYTES + WB2_BYTES + W3_BYTES + (C + HD + HD + C) * 4;
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
static_assert(HPLN % 256 == 0 && CPLN % 256 == 0 && XPL % 256 == 0, "planes a multiple of 256 B apart: conflict-free fragment reads");
}  // namespace cs

__device__ __forceinline__ void cs_wait_vmcnt0() { __builtin_amdgcn_s_waitcnt(0x0F70); }
__device__ __forceinline__ void cs_lds_dma16(__amdgpu_buffer_rsrc_t rsrc, char* dst, int voff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, voff, 0, 0, 0);
}

// weights [rows][kpad] bf16 (engine packing) -> LDS [K-step][row' = fragment * 16 + MFMA row][64 B], chunk c of a row stored at
// c ^ (((row' & 15) >> 3) << 1); (fragment j, MFMA row r) -> channel (j>>1)*32 + (r>>2)*8 + (j&1)*4 + (r&3), the epilogues' usual one
template <int ROWS, int KSTEPS>
__device__ __forceinline__ void cs_stage_weights(const void* w, int kpad, char* lds, int tid)
{
    const char* src = reinterpret_cast<const char*>(w);
    for (int idx = tid; idx < KSTEPS * ROWS * 4; idx += cs::NT) {
        const int ks = idx / (ROWS * 4), rc = idx - ks * (ROWS * 4);
        const int rowp = rc >> 2, c = rc & 3;
        const int j = rowp >> 4, r = rowp & 15;
        const int ch = (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3);
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(src + (long)ch * kpad * 2 + ks * 64 + c * 16);
        *reinterpret_cast<u32x4_t*>(lds + ks * (ROWS * 64) + rowp * 64 + ((c ^ (((r >> 3) & 1) << 1)) << 4)) = v;
    }
}

__global__ void __launch_bounds__(cs::NT) csp_stage_kernel(const CspStageArgs a)
{
    using namespace cs;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const xt = smem;
    char* const y1t = xt + X_BYTES;
    char* const ut = y1t + H_BYTES;
    char* const y2t = ut + H_BYTES;
    char* const w12 = y2t + C_BYTES;
    char* const wb1 = w12 + W12_BYTES;
    char* const wb2 = wb1 + WB1_BYTES;
    char* const w3 = wb2 + WB2_BYTES;
    float* const b12 = reinterpret_cast<float*>(w3 + W3_BYTES);
    float* const bb1 = b12 + C;
    float* const bb2 = bb1 + HD;
    float* const b3 = bb2 + HD;

    // (wave-uniform: LDS-DMA targets, branches.  SKY_AB_WAVE_VECTOR / SKY_AB_SETPRIO: one-off A/B builds of experiments/README.md)
#ifdef SKY_AB_WAVE_VECTOR
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#else
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int fr = lane & 15, fq = lane >> 4;
    const int tiles_x = (a.W + TS - 1) / TS, tiles_y = (a.H + TS - 1) / TS;
    const int ntile = a.B * tiles_y * tiles_x;
    int tile, tstep, tend;                                   // XCD-aware tile order (conv_frag.h: tile_walk)
    tile_walk(ntile, tile, tstep, tend);
    if (tile >= tend) return;
    const int pix_b = a.ldi * 2;

    cs_stage_weights<C, 2>(a.w12, a.kpad12, w12, tid);
    cs_stage_weights<HD, 1>(a.wb1, a.kpadb1, wb1, tid);
    cs_stage_weights<HD, 9>(a.wb2, a.kpadb2, wb2, tid);
    cs_stage_weights<C, 2>(a.w3, a.kpad3, w3, tid);
    for (int i = tid; i < C; i += NT) { b12[i] = a.b12[i]; b3[i] = a.b3[i]; }
    for (int i = tid; i < HD; i += NT) { bb1[i] = a.bb1[i]; bb2[i] = a.bb2[i]; }

    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, (int)a.in_bytes, 0x00020000);
    auto decode_tile = [&](int t, int& bimg, int& y0, int& x0) {
        const int tx = t % tiles_x;
        const int q = t / tiles_x;
        bimg = q / tiles_y;
        y0 = (q - bimg * tiles_y) * TS;
        x0 = tx * TS;
    };
    // x tile DMA: wave w fills plane w & 3, instructions [0, 6) (waves 0..3) or [6, 11) (waves 4..7); in instruction b lane ->
    // pixel slot p = b*32 + (lane >> 1), 16-byte half lane & 1 = K-step (lane & 1) ^ (p >> 3 & 1); outside the image: zeros
    auto issue_x = [&](int bimg, int y0, int x0) {
        const int pl = wave & 3;
        const int base = ((bimg * a.H + y0 - 1) * a.W + x0 - 1) * pix_b + pl * 16;
        const int b0 = wave < 4 ? 0 : 6, b1 = wave < 4 ? 6 : XDMA;
        for (int b = b0; b < b1; ++b) {
            const int p = b * 32 + (lane >> 1);
            const int hy = (p * 3641) >> 16, hx = p - hy * HW;            // p / 18
            const int kk = (lane & 1) ^ ((p >> 3) & 1);
            const bool ok = p < NHP && (unsigned)(y0 - 1 + hy) < (unsigned)a.H && (unsigned)(x0 - 1 + hx) < (unsigned)a.W;
            cs_lds_dma16(irsrc, xt + pl * XPL + b * 1024, ok ? base + (hy * a.W + hx) * pix_b + kk * 64 : -1);
        }
    };
    const int aswz = ((fq ^ (((fr >> 3) & 1) << 1)) << 4);          // weight fragment: row fr of a fragment, chunk fq
    auto wfrag = [&](const char* w, int rows, int ks, int j) -> u32x4_t {
        return *reinterpret_cast<const u32x4_t*>(w + ks * (rows * 64) + (j * 16 + fr) * 64 + aswz);
    };
    auto xfrag = [&](int p, int kk) -> u32x4_t {                     // x tile: pixel slot p, K-step kk, this lane's K-group
        return *reinterpret_cast<const u32x4_t*>(xt + fq * XPL + p * 32 + ((kk ^ ((p >> 3) & 1)) << 4));
    };
    // 8 channels fq*8 .. +7 (+ 32 s) of this lane's pixel: fragments 2s, 2s + 1 -> bias, SiLU, bf16
    // (the bias is the accumulators' initial value -- k_conv_halo.hip: acc_start -- : bias_pair() starts a fragment pair)
    auto act_pack = [&](const f32x4_t& a0, const f32x4_t& a1, float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = S1<__bf16>::silu(a0[e]);
            v[4 + e] = S1<__bf16>::silu(a1[e]);
        }
    };
    auto bias_pair = [&](const float* bias, f32x4_t& a0, f32x4_t& a1) {
        a0 = *reinterpret_cast<const f32x4_t*>(bias + fq * 8);
        a1 = *reinterpret_cast<const f32x4_t*>(bias + fq * 8 + 4);
    };

    int bimg, y0, x0;
    decode_tile(tile, bimg, y0, x0);
    __syncthreads();                                              // weights and biases staged
#ifdef SKY_AB_SETPRIO
    if (__builtin_amdgcn_readfirstlane(tid >> 6) >= 4) __builtin_amdgcn_s_setprio(1);      // static priority for the younger half
#endif
    issue_x(bimg, y0, x0);
    for (;;) {
        cs_wait_vmcnt0();
        __syncthreads();                                          // x tile landed; every wave is done with the previous tile's LDS tiles
        // ---- step 2a: y1 and u on the halo fragments wave, wave + 8, wave + 16 ----
#pragma unroll 1
        for (int f = wave; f < NFR; f += NW) {
            const int p = f * 16 + fr;
            const int pc = p < NHP ? p : NHP - 1;                 // past the tile: any valid pixel, never stored
            const u32x4_t x0f = xfrag(pc, 0), x1f = xfrag(pc, 1);
            f32x4_t acc0, acc1;
            bias_pair(b12, acc0, acc1);
            S1<__bf16>::mma(wfrag(w12, C, 0, 0), x0f, acc0);
            S1<__bf16>::mma(wfrag(w12, C, 0, 1), x0f, acc1);
            S1<__bf16>::mma(wfrag(w12, C, 1, 0), x1f, acc0);
            S1<__bf16>::mma(wfrag(w12, C, 1, 1), x1f, acc1);
            float v[8];
            act_pack(acc0, acc1, v);
            const Out8<__bf16>::raw_t y1v = Out8<__bf16>::pack(v, 1.0f);
            f32x4_t au0, au1;
            bias_pair(bb1, au0, au1);
            S1<__bf16>::mma(wfrag(wb1, HD, 0, 0), y1v.a, au0);
            S1<__bf16>::mma(wfrag(wb1, HD, 0, 1), y1v.a, au1);
            act_pack(au0, au1, v);
            Out8<__bf16>::raw_t uv = Out8<__bf16>::pack(v, 1.0f);
            const int hy = (pc * 3641) >> 16, hx = pc - hy * HW;
            const bool inside = (unsigned)(y0 - 1 + hy) < (unsigned)a.H && (unsigned)(x0 - 1 + hx) < (unsigned)a.W;
            if (!inside) uv.a = u32x4_t{0u, 0u, 0u, 0u};          // the 3x3's zero padding
            if (p < NHP) {
                *reinterpret_cast<u32x4_t*>(y1t + fq * HPLN + p * 16) = y1v.a;
                *reinterpret_cast<u32x4_t*>(ut + fq * HPLN + p * 16) = uv.a;
            }
        }
        // ---- step 2b: y2 on tile rows 2 wave, 2 wave + 1 ----
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = 2 * wave + i;
            const int p = (r + 1) * HW + 1 + fr;
            const u32x4_t x0f = xfrag(p, 0), x1f = xfrag(p, 1);
            f32x4_t acc0, acc1;
            bias_pair(b12 + HD, acc0, acc1);
            S1<__bf16>::mma(wfrag(w12, C, 0, 2), x0f, acc0);
            S1<__bf16>::mma(wfrag(w12, C, 0, 3), x0f, acc1);
            S1<__bf16>::mma(wfrag(w12, C, 1, 2), x1f, acc0);
            S1<__bf16>::mma(wfrag(w12, C, 1, 3), x1f, acc1);
            float v[8];
            act_pack(acc0, acc1, v);
            *reinterpret_cast<u32x4_t*>(y2t + fq * CPLN + (r * 16 + fr) * 16) = Out8<__bf16>::pack(v, 1.0f).a;
        }
        __syncthreads();                                          // u, y1, y2 complete; x is dead
        const int next = tile + tstep;
        int nb = 0, ny0 = 0, nx0 = 0;
        if (next < tend) {
            decode_tile(next, nb, ny0, nx0);
            issue_x(nb, ny0, nx0);                                // flies under step 4
        }
        // ---- step 4: 3x3 over u, residual, cv3 on tile rows 2 wave, 2 wave + 1 ----
        {
            f32x4_t acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) bias_pair(bb2, acc[0][i], acc[1][i]);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
                u32x4_t pf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) pf[i] = *reinterpret_cast<const u32x4_t*>(ut + fq * HPLN + ((2 * wave + i + ky) * HW + fr + kx) * 16);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const u32x4_t wf = wfrag(wb2, HD, tap, j);
#pragma unroll
                    for (int i = 0; i < 2; ++i) S1<__bf16>::mma(wf, pf[i], acc[j][i]);
                }
            }
            f32x4_t acc3[4][2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i) bias_pair(b3 + s * 32, acc3[2 * s][i], acc3[2 * s + 1][i]);
            u32x4_t vv[2], y2v[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = 2 * wave + i;
                float v[8];
                if (a.shortcut) {
                    // hipcc contracts the activation's last multiply with the residual add of the layer-by-layer epilogue into one fma
                    // (tile_epilogue, k_conv_halo.hip); written out here so that both forms round alike
                    const Out8<__bf16>::raw_t rv = Out8<__bf16>::load(y1t + fq * HPLN + ((r + 1) * HW + 1 + fr) * 16);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float xx = e < 4 ? acc[0][i][e] : acc[1][i][e - 4];
                        const float t = S1<__bf16>::gate(xx);
                        const float res = (e & 1) ? __uint_as_float(rv.a[e >> 1] & 0xffff0000u) : __uint_as_float(rv.a[e >> 1] << 16);
                        v[e] = __builtin_fmaf(xx, t, res);
                    }
                } else {
                    act_pack(acc[0][i], acc[1][i], v);
                }
                vv[i] = Out8<__bf16>::pack(v, 1.0f).a;
                y2v[i] = *reinterpret_cast<const u32x4_t*>(y2t + fq * CPLN + (r * 16 + fr) * 16);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4_t wf = wfrag(w3, C, 0, j);
#pragma unroll
                for (int i = 0; i < 2; ++i) S1<__bf16>::mma(wf, vv[i], acc3[j][i]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4_t wf = wfrag(w3, C, 1, j);
#pragma unroll
                for (int i = 0; i < 2; ++i) S1<__bf16>::mma(wf, y2v[i], acc3[j][i]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int oy = y0 + 2 * wave + i, ox = x0 + fr;
                if (oy < a.H && ox < a.W) {
                    char* op = reinterpret_cast<char*>(a.out) + (((long)bimg * a.H + oy) * a.W + ox) * (long)a.ldo * 2;
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        float v[8];
                        act_pack(acc3[2 * s][i], acc3[2 * s + 1][i], v);
                        Out8<__bf16>::store(Out8<__bf16>::pack(v, 1.0f), op + (s * 32 + fq * 8) * 2);
                    }
                }
            }
        }
        if (next >= tend) break;
        tile = next; bimg = nb; y0 = ny0; x0 = nx0;
    }
}

bool csp_stage_supported(const CspStageArgs& a)
{
    return a.c == cs::C && a.hidden == cs::HD && a.ldi % 8 == 0 && a.ldo % 8 == 0 && a.in_bytes != 0 && a.kpad12 >= cs::C && a.kpadb1 >= cs::HD &&
           a.kpadb2 >= 9 * cs::HD && a.kpad3 >= cs::C && a.H >= 1 && a.W >= 1 && !(a.opts & OPT_NO_CSP_STAGE);
}

hipError_t launch_csp_stage(const CspStageArgs& a, hipStream_t s)
{
    if (!csp_stage_supported(a)) return hipErrorNotSupported;
    static size_t attr[16] = {0};
    {
        const hipError_t e = ensure_lds_attr(reinterpret_cast<const void*>(csp_stage_kernel), cs::LDS_BYTES, a.device, attr);
        if (e != hipSuccess) return e;
    }
    const int ntile = a.B * ((a.H + cs::TS - 1) / cs::TS) * ((a.W + cs::TS - 1) / cs::TS);
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
    const int gx = ntile < n_cu ? ntile : n_cu;
    hipLaunchKernelGGL(csp_stage_kernel, dim3(gx), dim3(cs::NT), cs::LDS_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace sky
