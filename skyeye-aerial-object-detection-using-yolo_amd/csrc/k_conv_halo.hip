// Halo-tile 3x3 convolution for gfx950 (MI355X): the stride-1, pad-1 3x3 ConvolutionBlocks with wide inputs
// (reference blocks.py:10-41; cv2 of BottleneckBlock with its residual add, blocks.py:88-90).
//
// The streaming kernel (k_conv_stream.hip) fetches every filter tap of every pixel from global memory: nine trips
// through the texture addresser per input byte, which is what bounds it (PMC: GRBM_TA_BUSY ~ 90 %, MFMA ~ 20 %).
// Here a workgroup owns a 16 x 16 tile of output pixels and brings the 18 x 18 input tile (tile + 1 pixel halo) into
// LDS ONCE per 128 bytes of input channels; all nine taps are then LDS reads.
//
//   * 4 waves per workgroup, 2 workgroups per CU (LDS 76.5 KB each): while one workgroup waits for its halo the other
//     owns the matrix pipes.  Wave w computes rows 4w..4w+3 of the tile (four 16-pixel MFMA fragments) against all
//     N_blk = 64 or 128 output channels: 16 or 32 MFMAs per pair of 4 + N_blk/16 fragment reads.
//   * HALO LAYOUT  [k-group plane f = 0..3][pixel p = hy*18 + hx, 352 slots][2 x 16 B]: the B operand of
//     v_mfma_*_16x16x32 wants lane (fr, fq) to hold 16 bytes of K-group fq of pixel fr; planes are 11264 B apart
//     (a multiple of 256) and the two 16-byte slots of a pixel are swapped on every other run of 8 pixels, so the 16
//     lanes of every ds_read_b128 lane group hit 16 distinct bank quads for ANY tap offset (tools/lds_bank_check.py).
//     The tile is written by LDS-DMA (buffer_load ... lds, 16 B per lane, no VGPR round trip): one instruction fills
//     1 KB = 32 consecutive pixels of one plane; the per-lane GLOBAL address does the de-interleave, and pixels outside
//     the image use offset 0xffffffff -> the range check writes zeros (= the conv's zero padding).
//   * WEIGHTS stream through a two-stage LDS ring, one slab = [N_blk rows][128 B] = one tap x 128 B of input channels,
//     also by LDS-DMA, 16-byte chunks XOR-swizzled by (row >> 1) & 7 (conflict-free fragment reads).  Rows are
//     permuted exactly like the streaming kernel's (fragment j, MFMA row r -> channel (j>>1)*32 + (r>>2)*8 + (j&1)*4 +
//     (r&3)) so a lane ends up with 8 consecutive output channels per fragment pair -> 16-byte stores from registers.
//   * one barrier per tap (64-byte K-steps x 2); the next slab's DMA is issued right after it.
#include "sky_kernels.h"

#include "conv_frag.h"

#include <stdlib.h>

namespace sky {

static constexpr int HWV = 4;                 // waves per workgroup
static constexpr int HPW = 18;                // halo tile edge (16 + 2)
static constexpr int HPIX = 352;              // pixel slots per plane (324 used), 11 DMA instructions of 32 pixels
static constexpr int HPL = HPIX * 32;         // bytes per plane: 11264 = 44 * 256
static constexpr int HALO_BYTES = 4 * HPL;    // 45056
static constexpr int HDMA = HPIX / 32;        // DMA instructions per plane

template <typename T, int NF>
__global__ void __launch_bounds__(HWV * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) conv_halo_kernel(const ConvArgs a)
{
    constexpr int NB = NF * 16;
    constexpr int WSLAB = NB * 128;               // bytes of one weight slab
    constexpr int WDMA = NB / 8 / HWV;            // weight DMA instructions per wave per slab (8 rows each)
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const halo = smem;
    char* const wring = smem + HALO_BYTES;
    float* const lbias = reinterpret_cast<float*>(smem + HALO_BYTES + 2 * WSLAB);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int Cb = a.Cin * (int)sizeof(T);
    const int nchunk = Cb >> 7;
    const int n0 = blockIdx.y * NB;
    const int tiles_x = (a.W + 15) >> 4, tiles_y = (a.H + 15) >> 4;
    const int ntile = a.B * tiles_y * tiles_x;
    const int pix_b = a.ldi * (int)sizeof(T);     // bytes between input pixels
    const int wpitch = a.Kpad * (int)sizeof(T);

    for (int i = tid; i < NB; i += HWV * 64) lbias[i] = a.bias[n0 + i];

    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, (int)a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, (int)((long)a.Cout * wpitch), 0x00020000);

    // ---- per-lane constants ----
    // halo DMA: this wave fills plane `wave`; lane -> pixel slot b*32 + (lane >> 1), 16-byte slot lane & 1
    int hrel[HDMA];                               // (hy * W + hx) * pix_b + channel byte of this lane's chunk, or -1
    int hyx[HDMA];                                // hy << 8 | hx
#pragma unroll
    for (int b = 0; b < HDMA; ++b) {
        const int p = b * 32 + (lane >> 1);
        const int hy = (p * 3641) >> 16, hx = p - hy * HPW;       // p / 18 for p < 352
        const int kk = (lane & 1) ^ ((p >> 3) & 1);
        hyx[b] = p < HPW * HPW ? (hy << 8 | hx) : -1;
        hrel[b] = (hy * a.W + hx) * pix_b + (kk * 4 + wave) * 16;
    }
    // weight DMA: instruction q of this wave fills LDS rows (wave*WDMA + q)*8 .. +7; lane -> row, swizzled chunk
    int wrel[WDMA];
#pragma unroll
    for (int q = 0; q < WDMA; ++q) {
        const int row = (wave * WDMA + q) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int j = row >> 4, r = row & 15;
        const int ch = (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3);
        wrel[q] = (n0 + ch) * wpitch + c * 16;
    }
    // fragment reads
    const int arow = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);               // weight fragment, K-step 0 (K-step 1: ^ 64)
    const int pb0 = fq * HPL + ((wave * 4) * HPW + fr) * 32;                 // pixel fragment 0, tap (0,0)

    auto issue_halo = [&](int bimg, int y0, int x0, int chunk) {
        const int base = ((bimg * a.H + (y0 - 1)) * a.W + (x0 - 1)) * pix_b + chunk * 128;
#pragma unroll
        for (int b = 0; b < HDMA; ++b) {
            const int hy = hyx[b] >> 8, hx = hyx[b] & 255;
            const bool ok = hyx[b] >= 0 && (unsigned)(y0 - 1 + hy) < (unsigned)a.H && (unsigned)(x0 - 1 + hx) < (unsigned)a.W;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(irsrc, (__attribute__((address_space(3))) void*)(halo + wave * HPL + b * 1024), 16,
                                                     ok ? base + hrel[b] : -1, 0, 0, 0);
        }
    };
    auto issue_w = [&](int tap, int chunk, int buf) {
        const int kb = tap * Cb + chunk * 128;
#pragma unroll
        for (int q = 0; q < WDMA; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(wring + buf * WSLAB + (wave * WDMA + q) * 1024),
                                                     16, wrel[q], kb, 0, 0);
    };

    f32x4_t acc[NF][4];
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    auto compute_tap = [&](int tap, int buf) {
        const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
        const int toff = (ky * HPW + kx) * 32;
        const char* wb = wring + buf * WSLAB;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            u32x4_t pf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int A = pb0 + toff + i * (HPW * 32);
                pf[i] = *reinterpret_cast<const u32x4_t*>(halo + A + ((((A >> 8) & 1) ^ kk) << 4));
            }
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const u32x4_t wf = *reinterpret_cast<const u32x4_t*>(wb + j * 2048 + (arow ^ (kk << 6)));
#pragma unroll
                for (int i = 0; i < 4; ++i) S1<T>::mma(wf, pf[i], acc[j][i]);
            }
        }
    };

    auto epilogue = [&](int bimg, int y0, int x0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int oy = y0 + wave * 4 + i, ox = x0 + fr;
            if (oy < a.H && ox < a.W) {
                const long m = ((long)bimg * a.H + oy) * a.W + ox;
#pragma unroll
                for (int s = 0; s < NF / 2; ++s) {
                    const int nl = s * 32 + fq * 8;
                    const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(lbias + nl);
                    const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(lbias + nl + 4);
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[2 * s][i][e] + b0[e];
                        v[4 + e] = acc[2 * s + 1][i][e] + b1[e];
                    }
                    if (a.act == ACT_SILU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = S1<T>::silu(v[e]);
                    } else if (a.act == ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
                    }
                    if (a.res) {
                        const char* rp = reinterpret_cast<const char*>(a.res) + (m * a.ldr + n0 + nl) * (long)sizeof(T);
                        const u32x4_t r0 = *reinterpret_cast<const u32x4_t*>(rp);
                        if (sizeof(T) == 2) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v[2 * e] += __uint_as_float(r0[e] << 16);
                                v[2 * e + 1] += __uint_as_float(r0[e] & 0xffff0000u);
                            }
                        } else {
                            const u32x4_t r1 = *reinterpret_cast<const u32x4_t*>(rp + 16);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v[e] += __uint_as_float(r0[e]);
                                v[4 + e] += __uint_as_float(r1[e]);
                            }
                        }
                    }
                    const int n = n0 + nl;
                    if (sizeof(T) == 2) {
                        u32x4_t o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const __bf16 lo = (__bf16)v[2 * e], hi = (__bf16)v[2 * e + 1];
                            o[e] = (unsigned int)__builtin_bit_cast(unsigned short, lo) | ((unsigned int)__builtin_bit_cast(unsigned short, hi) << 16);
                        }
                        *reinterpret_cast<u32x4_t*>(reinterpret_cast<unsigned short*>(a.out) + m * a.ldo + n) = o;
                    } else {
                        float* op = reinterpret_cast<float*>(a.out) + m * a.ldo + n;
                        *reinterpret_cast<f32x4_t*>(op) = f32x4_t{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4_t*>(op + 4) = f32x4_t{v[4], v[5], v[6], v[7]};
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < NF; ++j) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
    };

    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {    // uniform per workgroup
        const int tx = tile % tiles_x;
        const int q = tile / tiles_x;
        const int ty = q % tiles_y;
        const int bimg = q / tiles_y;
        const int y0 = ty * 16, x0 = tx * 16;
        for (int chunk = 0; chunk < nchunk; ++chunk) {
            __syncthreads();                       // every wave is done with the halo and with both weight stages
            issue_halo(bimg, y0, x0, chunk);
            issue_w(0, chunk, 0);
            for (int tap = 0; tap < 9; ++tap) {
                __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0): this wave's DMA (slab `tap`, the halo) has landed
                __syncthreads();                       // ... and everybody else's; compute(tap - 1) is over everywhere
                if (tap < 8) issue_w(tap + 1, chunk, (tap + 1) & 1);
                compute_tap(tap, tap & 1);
            }
        }
        epilogue(bimg, y0, x0);
    }
}

// ------------------------------------------------------------------------------------------------ host
template <typename T, int NF>
static hipError_t halo_launch(const ConvArgs& a, hipStream_t s, int n_cu)
{
    constexpr int NB = NF * 16;
    const size_t lds = HALO_BYTES + 2 * NB * 128 + NB * 4;
    auto kern = conv_halo_kernel<T, NF>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int ntile = a.B * ((a.H + 15) / 16) * ((a.W + 15) / 16);
    int gx = ntile < 2 * n_cu ? ntile : 2 * n_cu;
    hipLaunchKernelGGL(kern, dim3(gx, a.Cout / NB), dim3(HWV * 64), lds, s, a);
    return hipGetLastError();
}

// returns hipErrorNotSupported when the shape is not covered / not worth it (caller falls back to the streaming kernel)
hipError_t launch_conv_halo(int dtype, const ConvArgs& a, hipStream_t s, int* variant)
{
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int esz = dtype == 0 ? 4 : 2;
    if (a.ks != 3 || a.stride != 1 || a.pad != 1 || a.head || a.out_f32 || a.up2) return hipErrorNotSupported;
    if (((long)a.Cin * esz) % 128 != 0 || a.Cout % 64 != 0 || a.in_bytes == 0) return hipErrorNotSupported;
    if ((long)a.Kpad * esz < 9L * a.Cin * esz) return hipErrorNotSupported;
    if ((long)a.Cout * a.Kpad * esz >= (1L << 31)) return hipErrorNotSupported;
    const char* mode = getenv("SKY_CONV_HALO");   // "0": never, "force": whenever the shape is covered
    if (mode && mode[0] == '0') return hipErrorNotSupported;
    if (!(mode && mode[0] == 'f')) {
        // partially filled tiles waste matrix work: keep the streaming kernel when less than 3/4 of the tile grid is image
        const long covered = (long)((a.H + 15) / 16) * ((a.W + 15) / 16) * 256;
        if ((long)a.H * a.W * 4 < covered * 3) return hipErrorNotSupported;
    }
    const int nb = a.Cout % 128 == 0 ? 128 : 64;
    hipError_t e;
    if (dtype == 0) e = nb == 128 ? halo_launch<float, 8>(a, s, n_cu) : halo_launch<float, 4>(a, s, n_cu);
    else e = nb == 128 ? halo_launch<__bf16, 8>(a, s, n_cu) : halo_launch<__bf16, 4>(a, s, n_cu);
    if (e == hipSuccess && variant) *variant = 4000 + nb;
    return e;
}

}  // namespace sky
