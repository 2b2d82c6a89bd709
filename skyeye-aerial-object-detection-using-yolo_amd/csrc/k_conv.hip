// Implicit-GEMM convolution for gfx950 (MI355X): 1x1 / 3x3, stride 1 / 2, NHWC activations.
//
// Replaces ConvolutionBlock.forward = SiLU(BN(conv(x)))  (reference blocks.py:10-41), the residual add of
// BottleneckBlock (blocks.py:88-90), the nearest 2x upsample + concat of FeatureNeck (detector.py:214-219),
// and DetectionHead.forward + process_detections (detector.py:61-145) as epilogue variants of one kernel.
//
// GEMM view: D[cout][pixel] = sum_k W[cout][k] * P[pixel][k],  k = (ky, kx, cin), M = B*Ho*Wo pixels.
//   * The MFMA A operand is the WEIGHT tile and the B operand the PIXEL tile, so each lane ends up with 4
//     consecutive output channels of one pixel (C/D map: col = lane&15 -> pixel, row = 4*(lane>>4)+reg -> cout).
//   * One K-step is 128 bytes of K per row (64 bf16 / 32 fp32).  Both tiles sit in LDS as [row][128 B] with
//     the 16-byte chunk index XOR-swizzled by (row & 7): conflict-free for ds_read_b128 (checked against the
//     lane-group table of MI355X_MICROARCH.md, LDS section).
//   * Global -> register -> LDS staging, double-buffered: the loads of K-step t+1 are issued before the MFMAs
//     of step t and written to the other stage after them; one barrier per K-step.
//   * bf16: v_mfma_f32_16x16x32_bf16 (2 per row per K-step).  fp32 "exact" mode: v_mfma_f32_16x16x4_f32; each
//     lane's 16-byte chunk feeds 4 MFMAs (element j of the chunk in MFMA j), which permutes the summation
//     order inside a K-step but sums every k exactly once.
//   * Epilogue goes through LDS (fp32 [BM][BN+4]) so that global stores are whole 16-byte channel vectors.
//   * blockIdx is remapped so that consecutive logical tiles (same pixel rows, all N tiles) land on one XCD.
#include "sky_kernels.h"

#include "conv_frag.h"

#include <stdlib.h>

namespace sky {

// T = the element type of the weights: the bf16 engine's SiLU layers are packed in the exp2 domain (conv_frag.h: S1<__bf16>::silu)
template <typename T>
__device__ __forceinline__ float act_apply(float v, int act)
{
    if (act == ACT_SILU) return sizeof(T) == 2 ? S1<__bf16>::silu(v) : v / (1.0f + expf(-v));
    if (act == ACT_RELU) return v > 0.0f ? v : 0.0f;
    return v;
}

__device__ __forceinline__ float bf16_bits_to_f32(unsigned int b) { return __uint_as_float(b << 16); }

template <typename T>
struct TypeInfo;
template <>
struct TypeInfo<float> {
    static constexpr int EPC = 4;  // elements per 16-byte chunk
};
template <>
struct TypeInfo<__bf16> {
    static constexpr int EPC = 8;
};
template <>
struct TypeInfo<fp8_t> {
    static constexpr int EPC = 16;
};

template <typename T>
__device__ __forceinline__ void mma_chunk(const u32x4_t& wf, const u32x4_t& pf, f32x4_t& acc);

template <>
__device__ __forceinline__ void mma_chunk<__bf16>(const u32x4_t& wf, const u32x4_t& pf, f32x4_t& acc)
{
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf), __builtin_bit_cast(bf16x8_t, pf),
                                                  acc, 0, 0, 0);
}

template <>
__device__ __forceinline__ void mma_chunk<fp8_t>(const u32x4_t& wf, const u32x4_t& pf, f32x4_t& acc)
{
    S1<fp8_t>::mma(wf, pf, acc);
}

template <>
__device__ __forceinline__ void mma_chunk<float>(const u32x4_t& wf, const u32x4_t& pf, f32x4_t& acc)
{
#pragma unroll
    for (int j = 0; j < 4; ++j)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wf[j]), __uint_as_float(pf[j]), acc, 0, 0, 0);
}

template <typename T, int WM, int WN, int MF, int NF, typename TO = T>
__global__ void __launch_bounds__(WM* WN * 64) conv_igemm_kernel(const ConvArgs a)
{
    constexpr bool QS = sizeof(T) == 1;   // fp8 operands: acc * mult[cout] (input scale x weight scale) before the bias
    constexpr int NT = WM * WN * 64;
    constexpr int BM = WM * MF * 16;
    constexpr int BN = WN * NF * 16;
    constexpr int EPC = TypeInfo<T>::EPC;
    constexpr int RPP = NT / 8;                       // rows covered per pass of the loader
    constexpr int PCH = BM / RPP;                     // pixel-tile chunks per thread
    constexpr int WCH = (BN + RPP - 1) / RPP;         // weight-tile chunks per thread
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int OP = BN + 4;                        // epilogue tile pitch (floats)
    static_assert(BM % RPP == 0, "tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // ---- XCD-aware tile order (bijective, cdna_hip_programming.md 5 "XCD swizzle must be bijective") ----
    const int nblk = gridDim.x;
    const int q8 = nblk >> 3, r8 = nblk & 7;
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    const int ntile = lid % a.ntiles, mtile = lid / a.ntiles;
    const int m0 = mtile * BM, n0 = ntile * BN;

    const T* __restrict__ in = reinterpret_cast<const T*>(a.in);
    const T* __restrict__ wgt = reinterpret_cast<const T*>(a.w);
    const T* __restrict__ zero = reinterpret_cast<const T*>(a.zero);

    // ---- loader state: this thread always fetches 16-byte chunk `cc` of rows rbase + i*RPP ----
    const int cc = tid & 7;
    const int rbase = tid >> 3;
    long pbase[PCH];
    int piy[PCH], pix[PCH];
#pragma unroll
    for (int i = 0; i < PCH; ++i) {
        const int m = m0 + rbase + i * RPP;
        if (m < a.M) {
            const int ox = m % a.Wo;
            const int t = m / a.Wo;
            const int oy = t % a.Ho;
            const int b = t / a.Ho;
            piy[i] = oy * a.stride - a.pad;
            pix[i] = ox * a.stride - a.pad;
            pbase[i] = ((long)(b * a.H + piy[i]) * a.W + pix[i]) * a.ldi;
        } else {
            piy[i] = -(1 << 24);
            pix[i] = -(1 << 24);
            pbase[i] = 0;
        }
    }
    const int cpc = a.Cin / EPC;      // chunks per filter tap
    const int taps = a.ks * a.ks;
    int tap = 0, cidx = cc;
    while (cidx >= cpc) { cidx -= cpc; ++tap; }
    const int nk = a.Kpad / (8 * EPC);

    u32x4_t preg[PCH], wreg[WCH];

    auto load_tiles = [&](int kt) {
        const int ky = a.ks == 3 ? (tap * 11) >> 5 : 0;
        const int kx = tap - ky * a.ks;
        const long koff = ((long)ky * a.W + kx) * a.ldi + cidx * EPC;
        const bool tap_ok = tap < taps;
        // Every load is unconditional: out-of-image taps, rows past M and K padding read a 16-byte block of zeros
        // instead (pointer select, no branch).  A branch around each load makes hipcc wait vmcnt(0) per load, i.e.
        // 8 dependent memory round trips per K-step (cdna_hip_programming.md 5, ".s-level traps" (c)).
#pragma unroll
        for (int i = 0; i < PCH; ++i) {
            const bool ok = tap_ok && (unsigned)(piy[i] + ky) < (unsigned)a.H && (unsigned)(pix[i] + kx) < (unsigned)a.W;
            const T* src = ok ? in + pbase[i] + koff : zero;
            preg[i] = *reinterpret_cast<const u32x4_t*>(src);
        }
#pragma unroll
        for (int j = 0; j < WCH; ++j) {
            const int n = rbase + j * RPP;
            const T* src = (n < BN) ? wgt + (long)(n0 + n) * a.Kpad + (long)kt * (8 * EPC) + cc * EPC : zero;
            wreg[j] = *reinterpret_cast<const u32x4_t*>(src);
        }
        cidx += 8;
        while (cidx >= cpc) { cidx -= cpc; ++tap; }
    };
    auto store_tiles = [&](int stage) {
        char* pb = smem + stage * STAGE;
        char* wb = pb + BM * 128;
#pragma unroll
        for (int i = 0; i < PCH; ++i) {
            const int row = rbase + i * RPP;
            *reinterpret_cast<u32x4_t*>(pb + row * 128 + ((cc ^ (row & 7)) << 4)) = preg[i];
        }
#pragma unroll
        for (int j = 0; j < WCH; ++j) {
            const int n = rbase + j * RPP;
            if (n < BN) *reinterpret_cast<u32x4_t*>(wb + n * 128 + ((cc ^ (n & 7)) << 4)) = wreg[j];
        }
    };

    f32x4_t acc[NF][MF];
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int i = 0; i < MF; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    const int fr = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tiles(kt + 1);
        const char* pb = smem + cur * STAGE;
        const char* wb = pb + BM * 128;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int chunk = ((kk * 4 + fq) ^ (fr & 7)) << 4;
            u32x4_t wf[NF], pf[MF];
#pragma unroll
            for (int j = 0; j < NF; ++j)
                wf[j] = *reinterpret_cast<const u32x4_t*>(wb + (wn * NF * 16 + j * 16 + fr) * 128 + chunk);
#pragma unroll
            for (int i = 0; i < MF; ++i)
                pf[i] = *reinterpret_cast<const u32x4_t*>(pb + (wm * MF * 16 + i * 16 + fr) * 128 + chunk);
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int i = 0; i < MF; ++i) mma_chunk<T>(wf[j], pf[i], acc[j][i]);
        }
        if (kt + 1 < nk) store_tiles(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: accumulators -> LDS fp32 [BM][OP] -> bias / act / residual -> global ----
    float* ot = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const int pm = wm * MF * 16 + i * 16 + fr;
            const int cn = wn * NF * 16 + j * 16 + fq * 4;
            *reinterpret_cast<f32x4_t*>(ot + pm * OP + cn) = acc[j][i];
        }
    __syncthreads();

    if (a.head && a.ntiles == 1) {
        // DetectionHead.forward (detector.py:79-84): channel n = anchor*no + o -> raw[b, anchor, y, x, o]
        // process_detections (detector.py:131-141): sigmoid, xy = (2s - 0.5 + grid)*stride, wh = (2s)^2 * anchor_grid
        // All anchors sit in this N tile.  Written in OUTPUT order: for one anchor the tile's consecutive pixels are one
        // contiguous run of BM*no floats in raw and in det, so consecutive threads store consecutive addresses; the
        // pixel -> (x, y, image) divisions happen once per pixel, not once per element.
        float* gx = ot + BM * OP;                                   // [BM] grid x, [BM] grid y
        int* rowb = reinterpret_cast<int*>(gx + 2 * BM);           // [BM] raw cell base, [BM] det row base; -1 = pixel past M
        const int HoWo = a.Ho * a.Wo;
        if (tid < BM) {
            const int m = m0 + tid;
            if (m < a.M) {
                const int x = m % a.Wo;
                const int t = m / a.Wo;
                const int y = t % a.Ho;
                const int b = t / a.Ho;
                gx[tid] = (float)x;
                gx[BM + tid] = (float)y;
                rowb[tid] = m + b * (a.na - 1) * HoWo;                                   // + anchor*HoWo = raw cell
                rowb[BM + tid] = m + (int)(b * (a.det_rows - HoWo) + a.det_off);           // + anchor*HoWo = det row
            } else {
                rowb[tid] = -1;
            }
        }
        __syncthreads();
        const int per = BM * a.no;
        const unsigned magic = (unsigned)(0x100000000ull / (unsigned)a.no) + 1u;    // r / no == umulhi(r, magic) for r < 2^16
        // one element: raw logit + decoded value of (pixel ml, output o) of anchor `an`
        auto element = [&](int an, int ml, int o, float aw, float ah, float& rawv, float& detv) {
#pragma clang fp contract(off)
            const int n = an * a.no + o;
            const float v = (QS ? ot[ml * OP + n] * (a.mult ? a.mult[n] : 1.0f) : ot[ml * OP + n]) + a.bias[n];
            rawv = v;
            const float s = head_sigmoid<sizeof(T) <= 2>(v);
            float d;
            if (o == 0) d = (s * 2.0f - 0.5f + gx[ml]) * a.stride_px;
            else if (o == 1) d = (s * 2.0f - 0.5f + gx[BM + ml]) * a.stride_px;
            else if (o == 2 || o == 3) { const float t2 = s * 2.0f; d = (t2 * t2) * (o == 2 ? aw : ah); }
            else d = s;
            detv = d;
        };
        // 16-byte stores when the runs are 16-byte aligned: (tile start, image size, det offset) * no all multiples of 4 floats
        const bool vec = (per & 3) == 0 && (((long)m0 * a.no) & 3) == 0 && (((long)HoWo * a.no) & 3) == 0 &&
                         (((long)a.det_off * a.no) & 3) == 0 && (((long)a.det_rows * a.no) & 3) == 0 &&
                         ((reinterpret_cast<size_t>(a.raw) | reinterpret_cast<size_t>(a.det)) & 15) == 0;
        for (int an = 0; an < a.na; ++an) {
            const float aw = a.anchor_wh[an * 2], ah = a.anchor_wh[an * 2 + 1];
            if (vec) {
                for (int r4 = tid; r4 < per / 4; r4 += NT) {
                    const int r0 = r4 * 4;
                    const int ml0 = (int)__umulhi((unsigned)r0, magic), ml3 = (int)__umulhi((unsigned)(r0 + 3), magic);
                    const int c0 = rowb[ml0], c3 = rowb[ml3];
                    if (c0 >= 0 && c3 >= 0 && c3 - c0 == ml3 - ml0) {      // the 4 elements are contiguous in raw and in det
                        f32x4_t rv4, dv4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int r = r0 + e;
                            const int ml = (int)__umulhi((unsigned)r, magic);
                            float rv, dv;
                            element(an, ml, r - ml * a.no, aw, ah, rv, dv);
                            rv4[e] = rv;
                            dv4[e] = dv;
                        }
                        const int o0 = r0 - ml0 * a.no;
                        if (a.raw) *reinterpret_cast<f32x4_t*>(a.raw + (long)(c0 + an * HoWo) * a.no + o0) = rv4;
                        *reinterpret_cast<f32x4_t*>(a.det + (long)(rowb[BM + ml0] + an * HoWo) * a.no + o0) = dv4;
                    } else {
                        for (int e = 0; e < 4; ++e) {
                            const int r = r0 + e;
                            const int ml = (int)__umulhi((unsigned)r, magic);
                            const int o = r - ml * a.no;
                            if (rowb[ml] < 0) continue;
                            float rv, dv;
                            element(an, ml, o, aw, ah, rv, dv);
                            if (a.raw) a.raw[(long)(rowb[ml] + an * HoWo) * a.no + o] = rv;
                            a.det[(long)(rowb[BM + ml] + an * HoWo) * a.no + o] = dv;
                        }
                    }
                }
                continue;
            }
            for (int r = tid; r < per; r += NT) {
                const int ml = (int)__umulhi((unsigned)r, magic);
                const int o = r - ml * a.no;
                if (rowb[ml] < 0) continue;
                float rv, dv;
                element(an, ml, o, aw, ah, rv, dv);
                if (a.raw) a.raw[(long)(rowb[ml] + an * HoWo) * a.no + o] = rv;
                a.det[(long)(rowb[BM + ml] + an * HoWo) * a.no + o] = dv;
            }
        }
        return;
    }
    if (a.head) {
        // several N tiles (na * no > 128): element-wise form of the same epilogue
#pragma clang fp contract(off)
        for (int idx = tid; idx < BM * BN; idx += NT) {
            const int ml = idx / BN, nl = idx - ml * BN;
            const int m = m0 + ml, n = n0 + nl;
            if (m >= a.M || n >= a.Cout) continue;
            const float v = (QS ? ot[ml * OP + nl] * (a.mult ? a.mult[n] : 1.0f) : ot[ml * OP + nl]) + a.bias[n];
            const int an = n / a.no, o = n - an * a.no;
            const int x = m % a.Wo;
            const int t = m / a.Wo;
            const int y = t % a.Ho;
            const int b = t / a.Ho;
            const long cell = ((long)(b * a.na + an) * a.Ho + y) * a.Wo + x;
            if (a.raw) a.raw[cell * a.no + o] = v;
            const float s = head_sigmoid<sizeof(T) <= 2>(v);
            float d;
            if (o == 0) d = (s * 2.0f - 0.5f + (float)x) * a.stride_px;
            else if (o == 1) d = (s * 2.0f - 0.5f + (float)y) * a.stride_px;
            else if (o == 2 || o == 3) { const float t2 = s * 2.0f; d = (t2 * t2) * a.anchor_wh[an * 2 + (o - 2)]; }
            else d = s;
            const long row = (long)b * a.det_rows + a.det_off + ((long)an * a.Ho + y) * a.Wo + x;
            a.det[row * a.no + o] = d;
        }
        return;
    }

    if (a.out_f32 || sizeof(TO) == 4) {                 // fp32 output (exact engine; channel counts are multiples of 4): 4 channels per 16-byte store
        constexpr int V = 4;
        constexpr int groups = BN / V;
        for (int idx = tid; idx < BM * groups; idx += NT) {
            const int ml = idx / groups, g = idx - ml * groups;
            const int m = m0 + ml, n = n0 + g * V;
            if (m >= a.M || n >= a.Cout) continue;
            float v[V];
#pragma unroll
            for (int e = 0; e < V; ++e)
                v[e] = act_apply<T>((QS ? ot[ml * OP + g * V + e] * (a.mult ? a.mult[n + e] : 1.0f) : ot[ml * OP + g * V + e]) + a.bias[n + e], a.act);
            if (a.res) {
                const f32x4_t r = *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(a.res) + (long)m * a.ldr + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += r[e];
            }
            long p0 = m;
            int rep = 1;
            long dstep_y = 0;
            if (a.up2) {
                const int x = m % a.Wo;
                const int t = m / a.Wo;
                const int y = t % a.Ho;
                const int b = t / a.Ho;
                p0 = ((long)(b * 2 * a.Ho + 2 * y)) * (2 * a.Wo) + 2 * x;
                rep = 4;
                dstep_y = 2 * a.Wo;
            }
            for (int r = 0; r < rep; ++r) {
                const long p = p0 + (r & 1) + (r >> 1) * dstep_y;
                *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(a.out) + p * a.ldo + n) = f32x4_t{v[0], v[1], v[2], v[3]};
            }
        }
        return;
    }
    // output in TO: 8 channels per item (16 bytes of bf16, 2 x 16 of fp32, 8 of fp8) through the same Out8 pack the other kernels use
    constexpr int V = 8;
    constexpr int groups = BN / V;
    for (int idx = tid; idx < BM * groups; idx += NT) {
        const int ml = idx / groups, g = idx - ml * groups;
        const int m = m0 + ml, n = n0 + g * V;
        if (m >= a.M || n >= a.Cout) continue;
        float v[V];
        const float* src = ot + ml * OP + g * V;
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] = act_apply<T>((QS ? src[e] * (a.mult ? a.mult[n + e] : 1.0f) : src[e]) + a.bias[n + e], a.act);
        if (a.res) Out8<TO>::add(Out8<TO>::load(reinterpret_cast<const char*>(a.res) + ((long)m * a.ldr + n) * (long)sizeof(TO)), v, a.res_scale);
        // destination pixel(s)
        long p0 = m;
        int rep = 1;
        long dstep_y = 0;
        if (a.up2) {
            const int x = m % a.Wo;
            const int t = m / a.Wo;
            const int y = t % a.Ho;
            const int b = t / a.Ho;
            p0 = ((long)(b * 2 * a.Ho + 2 * y)) * (2 * a.Wo) + 2 * x;
            rep = 4;
            dstep_y = 2 * a.Wo;
        }
        const typename Out8<TO>::raw_t o = Out8<TO>::pack(v, a.out_inv_scale);
        for (int r = 0; r < rep; ++r) {
            const long p = p0 + (r & 1) + (r >> 1) * dstep_y;
            Out8<TO>::store(o, reinterpret_cast<char*>(a.out) + (p * a.ldo + n) * (long)sizeof(TO));
        }
    }
}

// ------------------------------------------------------------------------------------------------ host
int conv_pick_bn(int cout)
{
    if (cout % 128 == 0) return 128;
    if (cout <= 32) return 32;
    if (cout <= 48) return 48;
    if (cout <= 64) return 64;
    if (cout <= 96 || cout % 96 == 0) return 96;
    return 128;   // last N tile partially filled (stores are predicated on n < Cout)
}

size_t conv_weight_rows(int cout)
{
    const int bn = conv_pick_bn(cout);
    return (size_t)((cout + bn - 1) / bn) * bn;
}

int conv_k_step(int dtype) { return 256 / dtype_size(dtype); }   // packed K is padded to 256 bytes (one ring slab)

template <typename T, typename TO, int WM, int WN, int MF, int NF>
static hipError_t launch_one(const ConvArgs& a, hipStream_t s)
{
    constexpr int BM = WM * MF * 16, BN = WN * NF * 16;
    constexpr size_t epi = BM * (BN + 4) * 4 + 4 * BM * 4;   // epilogue tile + per-pixel head tables
    constexpr size_t lds = (2 * (BM + BN) * 128 > epi) ? 2 * (BM + BN) * 128 : epi;
    static size_t attr[16] = {0};
    auto kern = conv_igemm_kernel<T, WM, WN, MF, NF, TO>;
    {
        const hipError_t e = ensure_lds_attr(reinterpret_cast<const void*>(kern), lds, a.device, attr);
        if (e != hipSuccess) return e;
    }
    ConvArgs b = a;
    b.ntiles = (a.Cout + BN - 1) / BN;
    const int mtiles = (a.M + BM - 1) / BM;
    hipLaunchKernelGGL(kern, dim3(mtiles * b.ntiles), dim3(WM * WN * 64), lds, s, b);
    return hipGetLastError();
}

template <typename T, typename TO>
static hipError_t launch_t(const ConvArgs& a, hipStream_t s)
{
    switch (conv_pick_bn(a.Cout)) {
        case 128: return launch_one<T, TO, 2, 2, 4, 4>(a, s);   // 128 px x 128 cout
        case 96: return launch_one<T, TO, 4, 1, 2, 6>(a, s);    // 128 px x 96
        case 64: return launch_one<T, TO, 4, 1, 2, 4>(a, s);    // 128 px x 64
        case 48: return launch_one<T, TO, 4, 1, 2, 3>(a, s);    // 128 px x 48 (detection heads: 45)
        default: return launch_one<T, TO, 4, 1, 4, 2>(a, s);    // 256 px x 32
    }
}

hipError_t launch_conv(int dtype, const ConvArgs& a, hipStream_t s, int* variant, int* fused)
{
    if (fused) *fused = 0;
    if (a.ks == 1 && !a.head) {         // large-K 1x1 layers: the GEMM where it applies
        const hipError_t eg = launch_gemm1x1(dtype, a, s, variant);
        if (eg != hipErrorNotSupported) return eg;
    }
    if (a.in2) {                        // planned with a second input: only the streaming kernel reads one (engine checks conv_accepts_in2)
        const hipError_t e2 = launch_conv_stream(dtype, a, s, variant, fused);
        return e2 == hipErrorNotSupported ? hipErrorInvalidValue : e2;
    }
    if (a.head && !(a.opts & OPT_NO_STREAM)) {
        const hipError_t e = launch_head_stream(dtype, a, s, variant);
        if (e != hipErrorNotSupported) return e;
    }
    if (!(a.opts & OPT_NO_STREAM)) {   // A/B switch for profiling
        const hipError_t eh = launch_conv_halo(dtype, a, s, variant, fused);
        if (eh != hipErrorNotSupported) return eh;
        const hipError_t e = launch_conv_stream(dtype, a, s, variant, fused);
        if (e != hipErrorNotSupported) return e;
    }
    if (a.src_mode) return hipErrorNotSupported;      // only the narrow-input halo kernel reads raw frames (engine checks conv_accepts_raw)
    if (variant) *variant = 1000 + conv_pick_bn(a.Cout);
    const int odt = a.out_dt < 0 ? dtype : a.out_dt;
    if (dtype == 0 && odt == 0) return launch_t<float, float>(a, s);
    if (dtype == 1 && odt == 1) return launch_t<__bf16, __bf16>(a, s);
    if (dtype == 1 && odt == 2) return launch_t<__bf16, fp8_t>(a, s);      // the fp8 engine's stem on shapes the narrow halo kernel leaves
    if (dtype == 2 && odt == 2) return launch_t<fp8_t, fp8_t>(a, s);
    return hipErrorNotSupported;
}

}  // namespace sky
