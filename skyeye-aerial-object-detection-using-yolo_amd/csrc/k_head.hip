// Detection level as a byte streamer (bf16 engine): DetectionHead.forward (reference detector.py:61-86: 1x1 convolution with bias,
// channel n = anchor * no + o -> raw[b, anchor, y, x, o]) + process_detections (detector.py:88-145: sigmoid, xy = (2s - 0.5 + grid) *
// stride, wh = (2s)^2 * anchor_grid) for na * no <= 48 output channels (the reference's 3 x (nc + 5) = 45).
//
// The layer is HBM-bound: 256 .. 2048 bytes of input and na * no * 4 bytes of fp32 output per pixel, ~1 MFMA per 64 bytes.  The
// general tile kernel (k_conv.hip) stages the pixels through LDS with a barrier per 128 bytes of K and runs short-lived workgroups;
// here, as in the streaming convolution, a wave owns its pixels:
//   * the 48 weight rows (zero past na * no) stay resident in LDS as MFMA A fragments [K-step][row][64 B];
//   * a wave walks 64 consecutive pixels at a time: the B operands come straight from global memory in fragment order (16 bytes per
//     lane, buffer descriptor: pixels past M read zeros), 3 x 4 accumulator fragments;
//   * epilogue per 16-pixel fragment: bias, decode, the na * no values of each pixel go through a wave-private LDS tile so that the
//     stores run in OUTPUT order -- for one anchor the 16 pixels are one contiguous run of 16 * no floats in det (and in raw) --, as
//     16-byte stores when the runs are aligned.
// K order (64-byte steps, K-group = 16-byte chunk) and every rounding of the epilogue are those of the tile kernel's detection
// epilogue: results are bit-identical to it (tests/test_gpu_head_stream.py).
#include "sky_kernels.h"

#include "conv_frag.h"

namespace sky {

namespace hd {
constexpr int NW = 8, NT = NW * 64;
constexpr int NR = 48;                 // weight rows / output channels held (3 MFMA fragments)
constexpr int SP = 49;                 // floats per pixel of the staging tile (odd pitch: the 16 pixel columns hit 16 banks)
}  // namespace hd

// MF: 16-pixel fragments per wave step (4 where there are pixels enough to fill the device with 64-pixel steps, else 2 or 1)
template <int MF>
__global__ void __launch_bounds__(hd::NT) head_stream_kernel(const ConvArgs a)
{
    using namespace hd;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int KS = (a.Cin * 2) >> 6;                               // 64-byte K-steps
    char* const wl = smem;                                         // [KS][48][64 B]
    float* const lbias = reinterpret_cast<float*>(wl + KS * NR * 64);
    float* const stage = lbias + NR;                               // per wave: det tile [16][SP], raw tile [16][SP]
    int* const meta = reinterpret_cast<int*>(stage + NW * 2 * 16 * SP);      // per wave: [16] raw cell base, [16] det row base (-1: past M)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int nout = a.na * a.no;
    const int HoWo = a.Ho * a.Wo;

    // weights [rows >= 48][Kpad] bf16 -> LDS, chunk c of a row at c ^ (((row & 15) >> 3) << 1); rows are channels (no permutation)
    for (int idx = tid; idx < KS * NR * 4; idx += NT) {
        const int ks = idx / (NR * 4), rc = idx - ks * (NR * 4);
        const int row = rc >> 2, c = rc & 3;
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (row < nout) v = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(a.w) + (long)row * a.Kpad * 2 + ks * 64 + c * 16);
        *reinterpret_cast<u32x4_t*>(wl + ks * (NR * 64) + row * 64 + ((c ^ (((row & 15) >> 3) << 1)) << 4)) = v;
    }
    for (int i = tid; i < NR; i += NT) lbias[i] = i < nout ? a.bias[i] : 0.0f;
    __syncthreads();

    // this lane's 12 output channels: fragment j, rows 4 fq .. 4 fq + 3 -> channel, anchor, output index, anchor size
    int ch_o[3][4];
    float ch_aw[3][4], ch_b[3][4];
    bool ch_ok[3][4];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = j * 16 + 4 * fq + e;
            const int an = c / a.no, o = c - an * a.no;
            ch_ok[j][e] = c < nout;
            ch_o[j][e] = o;
            ch_aw[j][e] = ch_ok[j][e] && (o == 2 || o == 3) ? a.anchor_wh[an * 2 + (o - 2)] : 0.0f;
            ch_b[j][e] = lbias[c];
        }
    const int aswz = ((fq ^ (((fr >> 3) & 1) << 1)) << 4);
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, (int)a.in_bytes, 0x00020000);
    float* const sdet = stage + wave * (2 * 16 * SP);
    float* const sraw = sdet + 16 * SP;
    int* const mrow = meta + wave * 32;
    const int pix_b = a.ldi * 2;
    const int per = 16 * a.no;                                     // floats of one anchor's run of a fragment
    const unsigned magic = (unsigned)(0x100000000ull / (unsigned)a.no) + 1u;      // r / no == umulhi(r, magic) for r < 2^16
    const bool vec_ok = (per & 3) == 0 && (((long)HoWo * a.no) & 3) == 0 && (((long)a.det_off * a.no) & 3) == 0 &&
                        (((long)a.det_rows * a.no) & 3) == 0 && ((reinterpret_cast<size_t>(a.raw) | reinterpret_cast<size_t>(a.det)) & 15) == 0;

    const int ngroups = (a.M + MF * 16 - 1) / (MF * 16);
    for (int g = blockIdx.x * NW + wave; g < ngroups; g += gridDim.x * NW) {
        const int m0 = g * (MF * 16);
        int voff[MF];
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int m = m0 + i * 16 + fr;
            voff[i] = m < a.M ? m * pix_b + fq * 16 : -1;          // (the immediate K offset keeps -1 out of range: in_bytes < 2^31)
        }
        f32x4_t acc[3][MF];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < MF; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // (runtime trip count and a convergent body: hipcc does not partially unroll this loop; the MF loads of a step are in flight together)
#pragma unroll 1
        for (int ks = 0; ks < KS; ++ks) {
            u32x4_t pf[MF];
#pragma unroll
            for (int i = 0; i < MF; ++i) pf[i] = __builtin_amdgcn_raw_buffer_load_b128(irsrc, voff[i] < 0 ? -1 : voff[i] + ks * 64, 0, 0);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const u32x4_t wf = *reinterpret_cast<const u32x4_t*>(wl + ks * (NR * 64) + (j * 16 + fr) * 64 + aswz);
#pragma unroll
                for (int i = 0; i < MF; ++i) S1<__bf16>::mma(wf, pf[i], acc[j][i]);
            }
        }
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int mf0 = m0 + i * 16;
            if (mf0 >= a.M) continue;                              // (uniform)
            // this lane's pixel: grid position, output rows
            const int m = mf0 + fr;
            const bool pok = m < a.M;
            const int x = m % a.Wo;
            const int t = m / a.Wo;
            const int y = t % a.Ho;
            const int b = t / a.Ho;
            const float gxf = (float)x, gyf = (float)y;
            if (fq == 0) {
                mrow[fr] = pok ? m + b * (a.na - 1) * HoWo : -1;                                   // + anchor * HoWo = raw cell
                mrow[16 + fr] = pok ? m + (int)(b * (a.det_rows - HoWo) + a.det_off) : -1;           // + anchor * HoWo = det row
            }
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma clang fp contract(off)
                    const int c = j * 16 + 4 * fq + e;
                    const float v = acc[j][i][e] + ch_b[j][e];
                    const float s = head_sigmoid<true>(v);
                    const int o = ch_o[j][e];
                    float d;
                    if (o == 0) d = (s * 2.0f - 0.5f + gxf) * a.stride_px;
                    else if (o == 1) d = (s * 2.0f - 0.5f + gyf) * a.stride_px;
                    else if (o == 2 || o == 3) { const float t2 = s * 2.0f; d = (t2 * t2) * ch_aw[j][e]; }
                    else d = s;
                    sdet[fr * SP + c] = d;
                    sraw[fr * SP + c] = v;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // out in OUTPUT order: anchor by anchor the fragment's 16 pixels are one run of 16 * no floats
            for (int an = 0; an < a.na; ++an) {
                const int cbase = an * a.no;
                if (vec_ok && (((long)mf0 * a.no) & 3) == 0) {
                    for (int r4 = lane; r4 < per / 4; r4 += 64) {
                        const int r0 = r4 * 4;
                        const int p0 = (int)__umulhi((unsigned)r0, magic), p3 = (int)__umulhi((unsigned)(r0 + 3), magic);
                        const int c0 = mrow[p0], c3 = mrow[p3];
                        const int o0 = r0 - p0 * a.no;
                        if (c0 >= 0 && c3 >= 0 && c3 - c0 == p3 - p0) {                 // contiguous in raw and in det
                            f32x4_t dv, rv;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int r = r0 + e;
                                const int p = (int)__umulhi((unsigned)r, magic);
                                const int idx = p * SP + cbase + (r - p * a.no);
                                dv[e] = sdet[idx];
                                rv[e] = sraw[idx];
                            }
                            *reinterpret_cast<f32x4_t*>(a.det + (long)(mrow[16 + p0] + an * HoWo) * a.no + o0) = dv;
                            if (a.raw) *reinterpret_cast<f32x4_t*>(a.raw + (long)(c0 + an * HoWo) * a.no + o0) = rv;
                        } else {
                            for (int e = 0; e < 4; ++e) {
                                const int r = r0 + e;
                                const int p = (int)__umulhi((unsigned)r, magic);
                                const int o = r - p * a.no;
                                if (mrow[p] < 0) continue;
                                a.det[(long)(mrow[16 + p] + an * HoWo) * a.no + o] = sdet[p * SP + cbase + o];
                                if (a.raw) a.raw[(long)(mrow[p] + an * HoWo) * a.no + o] = sraw[p * SP + cbase + o];
                            }
                        }
                    }
                } else {
                    for (int r = lane; r < per; r += 64) {
                        const int p = (int)__umulhi((unsigned)r, magic);
                        const int o = r - p * a.no;
                        if (mrow[p] < 0) continue;
                        a.det[(long)(mrow[16 + p] + an * HoWo) * a.no + o] = sdet[p * SP + cbase + o];
                        if (a.raw) a.raw[(long)(mrow[p] + an * HoWo) * a.no + o] = sraw[p * SP + cbase + o];
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                       // the tile is rewritten by the next fragment
        }
    }
}

// ---- CSP cv3 (1x1, 128 -> 128, BatchNorm folded, SiLU) + the detection level that reads its output, one kernel -------------------------
// fpn_conv3.cv3 (blocks.py:119-123) produces the 160 x 160 map P3 that detection level 0 (detector.py:61-145) reads: as two launches the
// 210 MB map is written and read back by an HBM-bound streamer each.  Here a wave computes cv3 for its 32 pixels (B operands straight
// from global, W3 resident in LDS as A fragments), stores the bf16 map (the stride-2 convolution of the neck reads it too) and keeps
// the packed vectors -- which ARE the B operands of the detection convolution -- for the 48-row GEMM, decode and output-order stores of
// head_stream_kernel.  Both GEMMs run the MFMA instruction and K order of the kernels they replace (conv_stream_kernel with 128
// resident channels, head_stream_kernel): cv3's map, the raw level and the decoded rows are bit-identical to the two launches
// (tests/test_gpu_head_stream.py).
namespace hd2 {
constexpr int C = 128, KS4 = 4, MF = 2;
}

__global__ void __launch_bounds__(hd::NT) cv3_head_kernel(const ConvArgs a)
{
    using namespace hd;
    using hd2::C;
    using hd2::KS4;
    using hd2::MF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const w3l = smem;                                        // cv3: [4][128 rows in fragment order][64 B]
    char* const wl = w3l + KS4 * C * 64;                           // detection level: [4][48][64 B]
    float* const lb3 = reinterpret_cast<float*>(wl + KS4 * NR * 64);
    float* const lbias = lb3 + C;
    float* const stage = lbias + NR;
    int* const meta = reinterpret_cast<int*>(stage + NW * 2 * 16 * SP);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int nout = a.na * a.no;
    const int HoWo = a.Ho * a.Wo;

    // cv3 weights [128][Kpad] -> LDS row' = fragment j * 16 + MFMA row r <- channel (j>>1)*32 + (r>>2)*8 + (j&1)*4 + (r&3) (a lane then owns 8
    // consecutive channels per fragment pair), chunk c of a row at c ^ (((r >> 3) & 1) << 1)
    for (int idx = tid; idx < KS4 * C * 4; idx += NT) {
        const int ks = idx / (C * 4), rc = idx - ks * (C * 4);
        const int rowp = rc >> 2, c = rc & 3;
        const int j = rowp >> 4, r = rowp & 15;
        const int ch = (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3);
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(a.w) + (long)ch * a.Kpad * 2 + ks * 64 + c * 16);
        *reinterpret_cast<u32x4_t*>(w3l + ks * (C * 64) + rowp * 64 + ((c ^ (((r >> 3) & 1) << 1)) << 4)) = v;
    }
    for (int idx = tid; idx < KS4 * NR * 4; idx += NT) {
        const int ks = idx / (NR * 4), rc = idx - ks * (NR * 4);
        const int row = rc >> 2, c = rc & 3;
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (row < nout) v = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(a.f2_w) + (long)row * a.f2_Kpad * 2 + ks * 64 + c * 16);
        *reinterpret_cast<u32x4_t*>(wl + ks * (NR * 64) + row * 64 + ((c ^ (((row & 15) >> 3) << 1)) << 4)) = v;
    }
    for (int i = tid; i < C; i += NT) lb3[i] = a.bias[i];
    for (int i = tid; i < NR; i += NT) lbias[i] = i < nout ? a.f2_bias[i] : 0.0f;
    __syncthreads();

    int ch_o[3][4];
    float ch_aw[3][4], ch_b[3][4];
    bool ch_ok[3][4];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = j * 16 + 4 * fq + e;
            const int an = c / a.no, o = c - an * a.no;
            ch_ok[j][e] = c < nout;
            ch_o[j][e] = o;
            ch_aw[j][e] = ch_ok[j][e] && (o == 2 || o == 3) ? a.anchor_wh[an * 2 + (o - 2)] : 0.0f;
            ch_b[j][e] = lbias[c];
        }
    const int aswz = ((fq ^ (((fr >> 3) & 1) << 1)) << 4);
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, (int)a.in_bytes, 0x00020000);
    float* const sdet = stage + wave * (2 * 16 * SP);
    float* const sraw = sdet + 16 * SP;
    int* const mrow = meta + wave * 32;
    const int pix_b = a.ldi * 2;
    const int per = 16 * a.no;
    const unsigned magic = (unsigned)(0x100000000ull / (unsigned)a.no) + 1u;
    const bool vec_ok = (per & 3) == 0 && (((long)HoWo * a.no) & 3) == 0 && (((long)a.det_off * a.no) & 3) == 0 &&
                        (((long)a.det_rows * a.no) & 3) == 0 && ((reinterpret_cast<size_t>(a.raw) | reinterpret_cast<size_t>(a.det)) & 15) == 0;

    const int ngroups = (a.M + MF * 16 - 1) / (MF * 16);
    // a wave's pixel vectors (B operands of cv3: MF fragments x 4 K-steps) are loaded one group ahead: the loads of group g + 1 are issued
    // right after the MFMAs of group g released the registers and fly under its two epilogues
    u32x4_t pf[KS4][MF];
    auto load_group = [&](int gg) {
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int m = gg * (MF * 16) + i * 16 + fr;
            const int vo = m < a.M ? m * pix_b + fq * 16 : -1;
#pragma unroll
            for (int ks = 0; ks < KS4; ++ks) pf[ks][i] = __builtin_amdgcn_raw_buffer_load_b128(irsrc, vo < 0 ? -1 : vo + ks * 64, 0, 0);
        }
    };
    const int g_first = blockIdx.x * NW + wave, g_step = gridDim.x * NW;
    if (g_first < ngroups) load_group(g_first);
    for (int g = g_first; g < ngroups; g += g_step) {
        const int m0 = g * (MF * 16);
        // ---- cv3: 128 -> 128 over 4 K-steps ----
        f32x4_t acc1[8][MF];                                      // start from cv3's bias (k_conv_halo.hip: acc_start): fragment j = channels (j >> 1) * 32 + fq * 8 + (j & 1) * 4 ..
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4_t c = *reinterpret_cast<const f32x4_t*>(lb3 + (j >> 1) * 32 + fq * 8 + (j & 1) * 4);
#pragma unroll
            for (int i = 0; i < MF; ++i) acc1[j][i] = c;
        }
#pragma unroll
        for (int ks = 0; ks < KS4; ++ks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const u32x4_t wf = *reinterpret_cast<const u32x4_t*>(w3l + ks * (C * 64) + (j * 16 + fr) * 64 + aswz);
#pragma unroll
                for (int i = 0; i < MF; ++i) S1<__bf16>::mma(wf, pf[ks][i], acc1[j][i]);
            }
        }
        if (g + g_step < ngroups) load_group(g + g_step);
        // bias, SiLU, bf16: the map goes to memory, the packed vectors stay as the B operands (K-step s, K-group fq) of the level
        u32x4_t bop[MF][4];
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int m = m0 + i * 16 + fr;
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) {
                const int nl = sg * 32 + fq * 8;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {                      // (cv3's bias was the accumulators' initial value)
                    v[e] = acc1[2 * sg][i][e];
                    v[4 + e] = acc1[2 * sg + 1][i][e];
                }
                if (a.act == ACT_SILU) {                       // (ACT_NONE: the head-attention variant's output projection, attention.py:312-399)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = S1<__bf16>::silu(v[e]);
                }
                const Out8<__bf16>::raw_t o = Out8<__bf16>::pack(v, 1.0f);
                if (m < a.M) Out8<__bf16>::store(o, reinterpret_cast<char*>(a.out) + ((long)m * a.ldo + nl) * 2);
                bop[i][sg] = m < a.M ? o.a : u32x4_t{0u, 0u, 0u, 0u};        // (the level reads zeros past M, like head_stream_kernel's range check)
            }
        }
        // ---- detection level: 128 -> na * no over the same 4 K-steps ----
        f32x4_t acc[3][MF];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < MF; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS4; ++ks)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const u32x4_t wf = *reinterpret_cast<const u32x4_t*>(wl + ks * (NR * 64) + (j * 16 + fr) * 64 + aswz);
#pragma unroll
                for (int i = 0; i < MF; ++i) S1<__bf16>::mma(wf, bop[i][ks], acc[j][i]);
            }
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int mf0 = m0 + i * 16;
            if (mf0 >= a.M) continue;                              // (uniform)
            const int m = mf0 + fr;
            const bool pok = m < a.M;
            const int x = m % a.Wo;
            const int t = m / a.Wo;
            const int y = t % a.Ho;
            const int b = t / a.Ho;
            const float gxf = (float)x, gyf = (float)y;
            if (fq == 0) {
                mrow[fr] = pok ? m + b * (a.na - 1) * HoWo : -1;
                mrow[16 + fr] = pok ? m + (int)(b * (a.det_rows - HoWo) + a.det_off) : -1;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma clang fp contract(off)
                    const int c = j * 16 + 4 * fq + e;
                    const float v = acc[j][i][e] + ch_b[j][e];
                    const float s = head_sigmoid<true>(v);
                    const int o = ch_o[j][e];
                    float d;
                    if (o == 0) d = (s * 2.0f - 0.5f + gxf) * a.stride_px;
                    else if (o == 1) d = (s * 2.0f - 0.5f + gyf) * a.stride_px;
                    else if (o == 2 || o == 3) { const float t2 = s * 2.0f; d = (t2 * t2) * ch_aw[j][e]; }
                    else d = s;
                    sdet[fr * SP + c] = d;
                    sraw[fr * SP + c] = v;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int an = 0; an < a.na; ++an) {
                const int cbase = an * a.no;
                if (vec_ok && (((long)mf0 * a.no) & 3) == 0) {
                    for (int r4 = lane; r4 < per / 4; r4 += 64) {
                        const int r0 = r4 * 4;
                        const int p0 = (int)__umulhi((unsigned)r0, magic), p3 = (int)__umulhi((unsigned)(r0 + 3), magic);
                        const int c0 = mrow[p0], c3 = mrow[p3];
                        const int o0 = r0 - p0 * a.no;
                        if (c0 >= 0 && c3 >= 0 && c3 - c0 == p3 - p0) {
                            f32x4_t dv, rv;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int r = r0 + e;
                                const int p = (int)__umulhi((unsigned)r, magic);
                                const int idx = p * SP + cbase + (r - p * a.no);
                                dv[e] = sdet[idx];
                                rv[e] = sraw[idx];
                            }
                            *reinterpret_cast<f32x4_t*>(a.det + (long)(mrow[16 + p0] + an * HoWo) * a.no + o0) = dv;
                            if (a.raw) *reinterpret_cast<f32x4_t*>(a.raw + (long)(c0 + an * HoWo) * a.no + o0) = rv;
                        } else {
                            for (int e = 0; e < 4; ++e) {
                                const int r = r0 + e;
                                const int p = (int)__umulhi((unsigned)r, magic);
                                const int o = r - p * a.no;
                                if (mrow[p] < 0) continue;
                                a.det[(long)(mrow[16 + p] + an * HoWo) * a.no + o] = sdet[p * SP + cbase + o];
                                if (a.raw) a.raw[(long)(mrow[p] + an * HoWo) * a.no + o] = sraw[p * SP + cbase + o];
                            }
                        }
                    }
                } else {
                    for (int r = lane; r < per; r += 64) {
                        const int p = (int)__umulhi((unsigned)r, magic);
                        const int o = r - p * a.no;
                        if (mrow[p] < 0) continue;
                        a.det[(long)(mrow[16 + p] + an * HoWo) * a.no + o] = sdet[p * SP + cbase + o];
                        if (a.raw) a.raw[(long)(mrow[p] + an * HoWo) * a.no + o] = sraw[p * SP + cbase + o];
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

static size_t cv3_head_lds_bytes()
{
    using namespace hd;
    return (size_t)hd2::KS4 * hd2::C * 64 + (size_t)hd2::KS4 * NR * 64 + (hd2::C + NR) * 4 + (size_t)NW * 2 * 16 * SP * 4 + NW * 32 * 4;
}

// `a`: the cv3 convolution (in, w, bias, out, ...) with the detection level in f2_w / f2_bias / f2_Kpad and the head fields (na, no, det, raw,
// det_rows, det_off, stride_px, anchor_wh); M pixels
bool cv3_head_supported(int dtype, const ConvArgs& a)
{
    return dtype == 1 && (a.out_dt < 0 || a.out_dt == 1) && a.ks == 1 && a.stride == 1 && a.Cin == hd2::C && a.Cout == hd2::C && (a.act == ACT_SILU || a.act == ACT_NONE) && !a.res &&
           !a.up2 && !a.src_mode && !a.in2 && !a.c1_w && a.f2_w && a.f2_bias && a.f2_Kpad >= hd2::C && a.Kpad >= hd2::C && a.na >= 1 && a.no >= 1 &&
           a.na * a.no <= hd::NR && a.na <= 8 && a.ldi % 8 == 0 && a.ldo % 8 == 0 && a.in_bytes != 0 && a.det != nullptr && a.M > 0 &&
           a.no < 65536 / 16 && !(a.opts & (OPT_NO_STREAM | OPT_NO_HEAD_STREAM | OPT_NO_CV3_HEAD));
}

hipError_t launch_cv3_head(int dtype, const ConvArgs& a, hipStream_t s)
{
    if (!cv3_head_supported(dtype, a)) return hipErrorNotSupported;
    const size_t lds = cv3_head_lds_bytes();
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
    const int ngroups = (a.M + hd2::MF * 16 - 1) / (hd2::MF * 16);
    // small levels: one 32-pixel step per wave slot does not fill the device -- the two launches do better
    if (ngroups < n_cu * hd::NW && !(a.opts & OPT_HEAD_STREAM_FORCE)) return hipErrorNotSupported;
    int gx = (ngroups + hd::NW - 1) / hd::NW;
    if (gx > n_cu) gx = n_cu;
    static size_t attr[16] = {0};
    {
        const hipError_t e = ensure_lds_attr(reinterpret_cast<const void*>(cv3_head_kernel), lds, a.device, attr);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(cv3_head_kernel, dim3(gx), dim3(hd::NT), lds, s, a);
    return hipGetLastError();
}

static size_t head_lds_bytes(int cin)
{
    using namespace hd;
    const int KS = (cin * 2) >> 6;
    return (size_t)KS * NR * 64 + NR * 4 + (size_t)NW * 2 * 16 * SP * 4 + NW * 32 * 4;
}

bool head_stream_supported(int dtype, const ConvArgs& a)
{
    const int odt = a.out_dt < 0 ? dtype : a.out_dt;
    (void)odt;
    return dtype == 1 && a.head && a.ks == 1 && a.stride == 1 && !a.up2 && !a.res && !a.src_mode && a.na >= 1 && a.no >= 1 &&
           a.na * a.no <= hd::NR && a.na * a.no == a.Cout && a.na <= 8 && a.Cin % 32 == 0 && a.Cin >= 32 && a.Cin <= 1024 && a.ldi % 8 == 0 &&
           a.in_bytes != 0 && a.Kpad >= a.Cin && a.det != nullptr && a.M > 0 && a.no < 65536 / 16 && !(a.opts & OPT_NO_HEAD_STREAM) &&
           head_lds_bytes(a.Cin) <= 160 * 1024;
}

hipError_t launch_head_stream(int dtype, const ConvArgs& a, hipStream_t s, int* variant)
{
    if (!head_stream_supported(dtype, a)) return hipErrorNotSupported;
    const size_t lds = head_lds_bytes(a.Cin);
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
    const int per_cu = lds <= 80 * 1024 ? 2 : 1;
    const int want = per_cu * n_cu * hd::NW;                       // wave steps that fill the device once
    const int mf = a.M / 64 >= want ? 4 : a.M / 32 >= want ? 2 : 1;
    // small levels (less than one 32-pixel step per wave slot of the device): the tile kernel's short-lived workgroups do better
    if (mf == 1 && !(a.opts & OPT_HEAD_STREAM_FORCE)) return hipErrorNotSupported;
    const int ngroups = (a.M + mf * 16 - 1) / (mf * 16);
    int gx = (ngroups + hd::NW - 1) / hd::NW;
    if (gx > per_cu * n_cu) gx = per_cu * n_cu;
    static size_t attr[3][16] = {{0}, {0}, {0}};
    const void* kern = mf == 4 ? reinterpret_cast<const void*>(head_stream_kernel<4>)
                     : mf == 2 ? reinterpret_cast<const void*>(head_stream_kernel<2>) : reinterpret_cast<const void*>(head_stream_kernel<1>);
    {
        const hipError_t e = ensure_lds_attr(kern, lds, a.device, attr[mf == 4 ? 0 : mf == 2 ? 1 : 2]);
        if (e != hipSuccess) return e;
    }
    if (mf == 4) hipLaunchKernelGGL(head_stream_kernel<4>, dim3(gx), dim3(hd::NT), lds, s, a);
    else if (mf == 2) hipLaunchKernelGGL(head_stream_kernel<2>, dim3(gx), dim3(hd::NT), lds, s, a);
    else hipLaunchKernelGGL(head_stream_kernel<1>, dim3(gx), dim3(hd::NT), lds, s, a);
    if (variant) *variant = 1500 + 48;
    return hipGetLastError();
}

}  // namespace sky
