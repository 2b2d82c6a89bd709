// The first two convolutions of the backbone as ONE kernel (bf16 engine, uint8 frames):
//     FocusBlock            blocks.py:152-182   space-to-depth + ConvolutionBlock(12, 32, 3, 1)     (backbone.py:48)
//     ConvolutionBlock      blocks.py:10-41     3x3 stride 2, 32 -> 64                              (backbone.py:50)
// Run as two launches the pair moves 157 MB of frames in, writes the 32-channel 640 x 640 stem map (839 MB per 32 frames of
// 1280 x 1280), reads it back and writes 419 MB: 2.3 GB for 0.58 GB of compulsory traffic, and both launches sit on the store /
// load path, not on the matrix pipe.  Here a workgroup owns a 16 x 16 tile of the STRIDE-2 OUTPUT and keeps everything between the
// frame and that tile in LDS:
//   A. the 70 x 72-byte raw blocks of the three colour planes (dword loads, prefetched into registers during the previous tile),
//      turned into the 35 x 35 space-to-depth tile (16 bf16 channels per pixel, two 16-byte planes) through the 256-entry
//      (bf16)(i / 255.0f) table -- the values the import kernel would have written (validate.py:236-238);
//   B. the stem convolution for the 33 x 33 stem pixels the tile's taps touch (69 MFMA pixel fragments over 8 waves, K = 9 taps x
//      32 bytes = 4.5 K-steps), bias + SiLU, bf16, written to LDS split by ROW / COLUMN PARITY so that the stride-2 taps of phase C
//      are unit-stride reads (cell (cy, cx) of parity (py, px) = stem pixel (2cy + py, 2cx + px)); zeros outside the image (they
//      are the stride-2 convolution's padding);
//   C. the stride-2 convolution from that tile (one 64-byte K-step per tap), bias + SiLU, 16-byte stores.
// One workgroup of 8 waves per CU with 155 KB of LDS; both weight sets stay resident ([K-step][row][64 B], rows in fragment order,
// 16-byte chunks swizzled so that every ds_read_b128 lane group hits 16 distinct bank quads).  The K order of both GEMMs and every
// rounding point equal the two-launch form (narrow halo kernel + streaming kernel): the result is bit-identical to it
// (tests/test_gpu_stem_down.py).  +6 % stem work (33 x 33 for 32 x 32 pixels).
#include "sky_kernels.h"

#include "conv_frag.h"

namespace sky {

namespace sd {
constexpr int NW = 8;                        // waves per workgroup
constexpr int NT = NW * 64;
constexpr int C1 = 32, C2 = 64;              // stem / stride-2 output channels
constexpr int TS = 16;                       // output tile side
constexpr int SS = 2 * TS + 1;               // stem pixels per side (33)
constexpr int HS = SS + 2;                   // space-to-depth pixels per side (35)
constexpr int RS = 2 * HS;                   // raw rows per colour (70)
constexpr int RDW = (2 * HS + 2) / 4;        // dwords per raw row: 72 bytes
constexpr int RAWP = RDW * 4;
constexpr int RAW_BYTES = 3 * RS * RAWP;     // 15 120
constexpr int NDW = (3 * RS * RDW + NT - 1) / NT;        // raw dwords per thread (8)
constexpr int S2D_SLOTS = (HS * HS + 15) / 16 * 16;       // 1232
constexpr int S2D_PLANE = S2D_SLOTS * 16;                 // 19 712 = 77 * 256
constexpr int S2D_BYTES = 2 * S2D_PLANE;
constexpr int NPIX = (S2D_SLOTS + NT - 1) / NT;           // space-to-depth pixels per thread (3)
// parity planes of the stem tile: cells per (parity, K-group) plane, padded to a multiple of 16 cells (256 B)
constexpr int PC00 = 304, PC01 = 272, PC10 = 272, PC11 = 256;      // 17x17 (pitch 17), 17x16 (16), 16x17 (17), 16x16 (16)
// The two column parities of a stem row are written by ALTERNATE lanes of one ds_write_b128 (phase B: lane fr -> stem pixel sx0 + fr, cell
// sx >> 1 of plane sx & 1): the store unit takes 8 consecutive lanes per LDS cycle = 4 cells of either plane, 64 bytes each.  With both
// planes at the same offset modulo 128 bytes those two runs fall on the SAME 16 banks (round 2 / 3: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
// = 0.205 for this kernel); the odd-column planes therefore start 64 bytes further, so that the eight lanes cover 32 distinct banks.
// (Every read of a plane is 16 consecutive cells of ONE plane: its banking does not depend on the plane's base.)
constexpr int PB00 = 0, PB01 = PB00 + 4 * PC00 * 16 + 64, PB10 = PB01 + 4 * PC01 * 16 + 64, PB11 = PB10 + 4 * PC10 * 16 + 64;
constexpr int STEM_BYTES = PB11 + 4 * PC11 * 16 + 64;     // 70 912
static_assert(PB00 % 128 == 0 && PB10 % 128 == 0 && PB01 % 128 == 64 && PB11 % 128 == 64, "odd-column planes 64 bytes off the even ones");
static_assert(STEM_BYTES % 256 == 0, "the space-to-depth tile behind the stem tile stays 256-byte aligned");
constexpr int KS1 = 5;                                    // 64-byte K-steps of the stem GEMM (288 bytes of K)
constexpr int W1_BYTES = KS1 * C1 * 64;                   // 10 240
constexpr int W2_BYTES = 9 * C2 * 64;                     // 36 864
constexpr int NPF = (SS * SS + 15) / 16;                  // stem pixel fragments (69)
constexpr int PFW = (NPF + NW - 1) / NW;                  // per wave (9)
constexpr int LDS_BYTES = STEM_BYTES + S2D_BYTES + W1_BYTES + W2_BYTES + (C1 + C2) * 4;
static_assert(RAW_BYTES <= STEM_BYTES, "the raw blocks are staged inside the (not yet written) stem tile");
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
static_assert(PFW % 3 == 0, "phase B walks the fragments of a wave in groups of three");

__device__ __forceinline__ int pcells(int par) { return par == 0 ? PC00 : par == 1 ? PC01 : par == 2 ? PC10 : PC11; }
__device__ __forceinline__ int pbase(int par) { return par == 0 ? PB00 : par == 1 ? PB01 : par == 2 ? PB10 : PB11; }
__device__ __forceinline__ int ppitch(int par) { return (par & 1) ? 16 : 17; }     // parity index = py * 2 + px
}  // namespace sd

// weights [rows][Kpad] bf16 (engine packing, K = (tap, cin)) -> LDS [K-step][row' = fragment * 16 + MFMA row][64 B], chunk c of a
// row stored at c ^ (((row' & 15) >> 3) << 1); the (fragment j, MFMA row r) -> channel permutation is the epilogues' usual one
template <int ROWS, int KSTEPS>
__device__ __forceinline__ void sd_stage_weights(const void* w, int kpad, int kbytes, char* lds, int tid)
{
    const char* src = reinterpret_cast<const char*>(w);
    for (int idx = tid; idx < KSTEPS * ROWS * 4; idx += sd::NT) {
        const int ks = idx / (ROWS * 4), rc = idx - ks * (ROWS * 4);
        const int rowp = rc >> 2, c = rc & 3;
        const int j = rowp >> 4, r = rowp & 15;
        const int ch = (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3);
        const int kb = ks * 64 + c * 16;
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (kb < kbytes) v = *reinterpret_cast<const u32x4_t*>(src + (long)ch * kpad * 2 + kb);
        *reinterpret_cast<u32x4_t*>(lds + ks * (ROWS * 64) + rowp * 64 + ((c ^ (((r >> 3) & 1) << 1)) << 4)) = v;
    }
}

// Stem pixel (sy, sx) of lane `fr` of pixel fragment f (round 4).  The 33 x 33 stem pixels were walked row-major, 16 per fragment: every second
// fragment then wraps from one tile row into the next and its 16 B operands are no longer one contiguous 256-byte run of the 35-pixel-wide
// space-to-depth tile -- 31 % of the LDS cycles of the stem GEMM's pixel reads were bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE =
// 0.205 for the kernel in rounds 2 and 3).  Now fragments 0 .. 65 are the two aligned halves of a row (columns 0 .. 15, 16 .. 31) and fragments
// 66 .. 68 walk column 32 downwards (pixels 560 bytes apart: 16 lanes on 16 distinct bank quads): the same 69 fragments, every read conflict-free.
// Returns false for a lane that holds no stem pixel (rows past 32 of the column fragments, fragments past 68); (sy, sx) is then a valid pixel.
__device__ __forceinline__ bool stem_pixel(int f, int fr, int& sy, int& sx)
{
    using namespace sd;
    if (f < 2 * SS) {
        sy = f >> 1;
        sx = (f & 1) * 16 + fr;
        return true;
    }
    sy = (f - 2 * SS) * 16 + fr;
    sx = SS - 1;
    const bool real = f < NPF && sy < SS;
    if (!real) sy = SS - 1;
    return real;
}

__global__ void __launch_bounds__(sd::NT) stem_down_kernel(const StemDownArgs a)
{
    using namespace sd;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const stem = smem;                                     // parity planes; the raw blocks are staged here first
    unsigned char* const rawl = reinterpret_cast<unsigned char*>(smem);
    char* const s2d = smem + STEM_BYTES;
    char* const w1 = s2d + S2D_BYTES;
    char* const w2 = w1 + W1_BYTES;
    float* const b1 = reinterpret_cast<float*>(w2 + W2_BYTES);
    float* const b2 = b1 + C1;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int Hs = a.Hr / 2, Ws = a.Wr / 2;                       // stem / space-to-depth map
    const int tiles_x = (a.Wo + TS - 1) / TS, tiles_y = (a.Ho + TS - 1) / TS;
    const int ntile = a.B * tiles_y * tiles_x;
    int tile, tstep, tend;                                   // XCD-aware tile order (conv_frag.h: tile_walk)
    tile_walk(ntile, tile, tstep, tend);
    if (tile >= tend) return;

    sd_stage_weights<C1, KS1>(a.w1, a.kpad1, 9 * 32, w1, tid);
    sd_stage_weights<C2, 9>(a.w2, a.kpad2, 9 * 64, w2, tid);
    for (int i = tid; i < C1; i += NT) b1[i] = a.bias1[i];
    for (int i = tid; i < C2; i += NT) b2[i] = a.bias2[i];

    auto decode_tile = [&](int t, int& bimg, int& y0, int& x0) {
        const int tx = t % tiles_x;
        const int q = t / tiles_x;
        bimg = q / tiles_y;
        y0 = (q - bimg * tiles_y) * TS;
        x0 = tx * TS;
    };
    // ---- phase A, part 1: the raw blocks of tile t into registers (zero outside the frame) ----
    unsigned int rawd[NDW];
    auto load_raw = [&](int t) {
        int bimg, y0, x0;
        decode_tile(t, bimg, y0, x0);
        const int ry0 = 4 * y0 - 4, rx0 = 4 * x0 - 4;             // multiples of 4: dword aligned
#pragma unroll
        for (int k = 0; k < NDW; ++k) {
            const int d = tid + k * NT;
            unsigned int v = 0u;
            if (d < 3 * RS * RDW) {
                const int r = d / RDW, cd = d - r * RDW;
                const int col = r / RS, ry = r - col * RS;
                const int y = ry0 + ry, x = rx0 + 4 * cd;
                if ((unsigned)y < (unsigned)a.Hr && x >= 0 && x + 3 < a.Wr)
                    v = *reinterpret_cast<const unsigned int*>(a.frames + (((long)bimg * 3 + col) * a.Hr + y) * a.Wr + x);
            }
            rawd[k] = v;
        }
    };
    // stem GEMM operand offsets of this lane: K-step ks covers 16-byte chunks ks*4 + fq of K = (tap, 2 chunks); chunk -> (tap, half)
    int toff[KS1];
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
        const int c = ks * 4 + fq;
        int tap = c >> 1;
        if (tap > 8) tap = 8;                                     // K padding: zero weights, any finite pixel
        const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
        toff[ks] = (c & 1) * S2D_PLANE + (ky * HS + kx) * 16;
    }
    const int aswz = ((fq ^ (((fr >> 3) & 1) << 1)) << 4);         // weight fragment: row fr of a fragment, chunk fq

    load_raw(tile);
    __syncthreads();                                              // weights, biases, table staged
#ifdef SKY_AB_SETPRIO
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) >= 4) __builtin_amdgcn_s_setprio(1);      // one-off A/B build (experiments/README.md)
#endif
    for (;;) {
        int bimg, y0, x0;
        decode_tile(tile, bimg, y0, x0);
        // ---- phase A, part 2: raw registers -> LDS block -> space-to-depth tile ----
#pragma unroll
        for (int k = 0; k < NDW; ++k) {
            const int d = tid + k * NT;
            if (d < 3 * RS * RDW) *reinterpret_cast<unsigned int*>(rawl + d * 4) = rawd[k];       // rows are RAWP = RDW * 4 bytes: d * 4 is (row, dword)
        }
        __syncthreads();
        // One item = one space-to-depth pixel: six 2-byte reads (colour x row of the 2x2 block), byte -> float by v_cvt_f32_ubyte,
        // times 1/255 (after the bf16 rounding equal to the reference's true division for every byte value: checked for all 256),
        // packed conversions, two 16-byte plane writes.  Channel = patch * 3 + colour, patches TL, BL, TR, BR (blocks.py:176-181).
#pragma unroll 1
        for (int k = 0; k < NPIX; ++k) {
            const int p = tid + k * NT;
            if (p < S2D_SLOTS) {
                const int hy = p / HS, hx = p - hy * HS;
                const int gy = 2 * y0 - 2 + hy, gx = 2 * x0 - 2 + hx;            // space-to-depth pixel in the frame's half-resolution map
                const bool inside = p < HS * HS && (unsigned)gy < (unsigned)Hs && (unsigned)gx < (unsigned)Ws;
                const int hyc = p < HS * HS ? hy : 0;
                float f[16];
#pragma unroll
                for (int col = 0; col < 3; ++col)
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy) {
                        const unsigned int two = *reinterpret_cast<const unsigned short*>(rawl + (col * RS + 2 * hyc + dy) * RAWP + 2 * hx);
                        f[dy * 3 + col] = (float)(two & 255u) * (1.0f / 255.0f);           // dx = 0: patches TL (dy 0), BL (dy 1)
                        f[6 + dy * 3 + col] = (float)(two >> 8) * (1.0f / 255.0f);         // dx = 1: patches TR, BR
                    }
#pragma unroll
                for (int e = 12; e < 16; ++e) f[e] = 0.0f;
                u32x4_t o0, o1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o0[e] = inside ? pack_bf16x2(f[2 * e], f[2 * e + 1]) : 0u;
                    o1[e] = inside ? pack_bf16x2(f[8 + 2 * e], f[8 + 2 * e + 1]) : 0u;
                }
                *reinterpret_cast<u32x4_t*>(s2d + p * 16) = o0;
                *reinterpret_cast<u32x4_t*>(s2d + S2D_PLANE + p * 16) = o1;
            }
        }
        __syncthreads();                                          // s2d tile complete; the raw block is dead, the stem tile may be written
        const int next = tile + tstep;
        if (next < tend) load_raw(next);                         // flies under phases B and C

        // ---- phase B: stem convolution on the 33 x 33 stem pixels: PFW fragments per wave in groups of three ----
        const f32x4_t bia0 = *reinterpret_cast<const f32x4_t*>(b1 + fq * 8);
        const f32x4_t bia1 = *reinterpret_cast<const f32x4_t*>(b1 + fq * 8 + 4);
        // Software pipeline: the MFMAs of group g + 1 are issued before the activation (VALU) of group g, so that the two pipes of
        // a SIMD overlap inside one wave (the eight waves of the workgroup move through the phases in step: no other wave does it).
        auto gemm3 = [&](int g3, f32x4_t (&acc)[2][3]) {
            int pb[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                int sy, sx;
                stem_pixel(wave * PFW + g3 * 3 + i, fr, sy, sx);  // (past the tile: a valid pixel, never stored)
                pb[i] = (sy * HS + sx) * 16;
                acc[0][i] = bia0;                                 // the accumulators start from the bias (k_conv_halo.hip: acc_start)
                acc[1][i] = bia1;
            }
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const u32x4_t wf0 = *reinterpret_cast<const u32x4_t*>(w1 + ks * (C1 * 64) + fr * 64 + aswz);
                const u32x4_t wf1 = *reinterpret_cast<const u32x4_t*>(w1 + ks * (C1 * 64) + (16 + fr) * 64 + aswz);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const u32x4_t pf = *reinterpret_cast<const u32x4_t*>(s2d + pb[i] + toff[ks]);
                    S1<__bf16>::mma(wf0, pf, acc[0][i]);
                    S1<__bf16>::mma(wf1, pf, acc[1][i]);
                }
            }
        };
        auto act3 = [&](int g3, const f32x4_t (&acc)[2][3]) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                int sy, sx;
                const bool real = stem_pixel(wave * PFW + g3 * 3 + i, fr, sy, sx);
                const int gy = 2 * y0 - 1 + sy, gx = 2 * x0 - 1 + sx;            // stem pixel in the 640 x 640 map
                const bool inside = (unsigned)gy < (unsigned)Hs && (unsigned)gx < (unsigned)Ws;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = S1<__bf16>::silu(acc[0][i][e]);
                    v[4 + e] = S1<__bf16>::silu(acc[1][i][e]);
                }
                Out8<__bf16>::raw_t o = Out8<__bf16>::pack(v, 1.0f);
                if (!inside) o.a = u32x4_t{0u, 0u, 0u, 0u};                       // the stride-2 convolution's zero padding
                const int par = (sy & 1) * 2 + (sx & 1);
                const int cell = (sy >> 1) * ppitch(par) + (sx >> 1);
                if (real) *reinterpret_cast<u32x4_t*>(stem + pbase(par) + fq * (pcells(par) * 16) + cell * 16) = o.a;
            }
        };
        {
            f32x4_t accA[2][3], accB[2][3];
            gemm3(0, accA);
            gemm3(1, accB);
            act3(0, accA);
            gemm3(2, accA);
            act3(1, accB);
            act3(2, accA);
        }
        __syncthreads();                                          // stem tile complete

        // ---- phase C: stride-2 convolution, tile rows 2 * wave and 2 * wave + 1 ----
        {
            f32x4_t acc[4][2];                                    // start from the bias: fragment j = channels (j >> 1) * 32 + fq * 8 + (j & 1) * 4 ..
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4_t c = *reinterpret_cast<const f32x4_t*>(b2 + (j >> 1) * 32 + fq * 8 + (j & 1) * 4);
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[j][i] = c;
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
                const int par = (ky & 1) * 2 + (kx & 1);
                const char* pl = stem + pbase(par) + fq * (pcells(par) * 16);
                u32x4_t pf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int y = 2 * wave + i;
                    pf[i] = *reinterpret_cast<const u32x4_t*>(pl + ((y + (ky >> 1)) * ppitch(par) + fr + (kx >> 1)) * 16);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32x4_t wf = *reinterpret_cast<const u32x4_t*>(w2 + tap * (C2 * 64) + (j * 16 + fr) * 64 + aswz);
#pragma unroll
                    for (int i = 0; i < 2; ++i) S1<__bf16>::mma(wf, pf[i], acc[j][i]);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int oy = y0 + 2 * wave + i, ox = x0 + fr;
                if (oy < a.Ho && ox < a.Wo) {
                    char* op = reinterpret_cast<char*>(a.out) + (((long)bimg * a.Ho + oy) * a.Wo + ox) * (long)a.ldo * 2;
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = S1<__bf16>::silu(acc[2 * s][i][e]);
                            v[4 + e] = S1<__bf16>::silu(acc[2 * s + 1][i][e]);
                        }
                        Out8<__bf16>::store(Out8<__bf16>::pack(v, 1.0f), op + (s * 32 + fq * 8) * 2);
                    }
                }
            }
        }
        if (next >= tend) break;
        tile = next;
        __syncthreads();                                          // everybody is done reading the stem tile (the next raw block goes there)
    }
}

bool stem_down_supported(const StemDownArgs& a)
{
    return a.c1 == sd::C1 && a.c2 == sd::C2 && a.Hr % 2 == 0 && a.Wr % 4 == 0 && a.kpad1 * 2 >= 9 * 32 && a.kpad2 * 2 >= 9 * 64 &&
           a.ldo % 8 == 0 && (double)a.B * 3.0 * a.Hr * a.Wr < 2147483000.0 && !(a.opts & OPT_NO_STEM_DOWN);
}

hipError_t launch_stem_down(const StemDownArgs& a, hipStream_t s)
{
    if (!stem_down_supported(a)) return hipErrorNotSupported;
    static size_t attr[16] = {0};
    {
        const hipError_t e = ensure_lds_attr(reinterpret_cast<const void*>(stem_down_kernel), sd::LDS_BYTES, a.device, attr);
        if (e != hipSuccess) return e;
    }
    const int ntile = a.B * ((a.Ho + sd::TS - 1) / sd::TS) * ((a.Wo + sd::TS - 1) / sd::TS);
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
    const int gx = ntile < n_cu ? ntile : n_cu;
    hipLaunchKernelGGL(stem_down_kernel, dim3(gx), dim3(sd::NT), sd::LDS_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace sky
