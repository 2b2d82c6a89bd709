// Probe of the gfx950 fp8 (OCP e4m3fn) conversions and MFMA operand layouts the fp8 engine relies on (run on the GPU box):
//   1. v_cvt_f32_fp8 of all 256 byte values against the OCP e4m3fn table; v_cvt_pk_fp8_f32 round trip and overflow behaviour
//   2. D = A * B^T with 16-byte fragments per lane (lane = (row l & 15, K-group l >> 4)): two v_mfma_f32_16x16x32_fp8_fp8 on the
//      low / high 8 bytes, and with 32-byte fragments one v_mfma_scale_f32_16x16x128_f8f6f4 (scales 2^0), against a host sum
// build: hipcc -O3 --offload-arch=gfx950 tools/fp8_probe.hip -o tools/fp8_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

static float e4m3_to_f32(unsigned char b)
{
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v;
    if (e == 15 && m == 7) return NAN;
    if (e == 0) v = ldexpf((float)m, -9);
    else v = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -v : v;
}

__global__ void cvt_kernel(float* dec, const float* in, unsigned char* enc, int n)
{
    const int t = threadIdx.x;
    dec[t] = __builtin_amdgcn_cvt_f32_fp8(t, 0);
    for (int i = t; i < n; i += 256) {
        const int p = __builtin_amdgcn_cvt_pk_fp8_f32(in[i], 0.0f, 0, false);
        enc[i] = (unsigned char)(p & 255);
    }
}

// A [16][K] bytes, B [16][K] bytes (both row-major, K contiguous); D[i][j] = sum_k A[i][k] * B[j][k]
__global__ void mfma_kernel(const unsigned char* A, const unsigned char* B, int K, float* D32, float* D128)
{
    const int lane = threadIdx.x, fr = lane & 15, fq = lane >> 4;
    f32x4_t acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 64) {        // 64-byte K-step: lane reads 16 bytes of K-group fq
        const u32x4_t a = *reinterpret_cast<const u32x4_t*>(A + fr * K + k0 + fq * 16);
        const u32x4_t b = *reinterpret_cast<const u32x4_t*>(B + fr * K + k0 + fq * 16);
        const long alo = (long)a[0] | ((long)a[1] << 32), ahi = (long)a[2] | ((long)a[3] << 32);
        const long blo = (long)b[0] | ((long)b[1] << 32), bhi = (long)b[2] | ((long)b[3] << 32);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(alo, blo, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(ahi, bhi, acc, 0, 0, 0);
    }
    for (int e = 0; e < 4; ++e) D32[(fq * 4 + e) * 16 + fr] = acc[e];     // C/D: row = 4*(lane>>4)+e (A row), col = lane&15 (B row)
    f32x4_t acc2 = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 128) {       // 128-byte K-step: lane holds the two 16-byte pieces (K-step halves) of K-group fq
        i32x8_t a, b;
        for (int h = 0; h < 2; ++h) {
            const u32x4_t av = *reinterpret_cast<const u32x4_t*>(A + fr * K + k0 + h * 64 + fq * 16);
            const u32x4_t bv = *reinterpret_cast<const u32x4_t*>(B + fr * K + k0 + h * 64 + fq * 16);
            for (int e = 0; e < 4; ++e) { a[h * 4 + e] = (int)av[e]; b[h * 4 + e] = (int)bv[e]; }
        }
        acc2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc2, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    for (int e = 0; e < 4; ++e) D128[(fq * 4 + e) * 16 + fr] = acc2[e];
}

int main()
{
    float *dec, *in, *d32, *d128;
    unsigned char *enc, *A, *B;
    const int n = 4096, K = 256;
    hipMalloc(&dec, 256 * 4); hipMalloc(&in, n * 4); hipMalloc(&enc, n); hipMalloc(&A, 16 * K); hipMalloc(&B, 16 * K);
    hipMalloc(&d32, 1024); hipMalloc(&d128, 1024);
    float hin[n];
    srand(1);
    for (int i = 0; i < n; ++i) hin[i] = ldexpf((float)rand() / RAND_MAX * 2 - 1, rand() % 14 - 8);
    hin[0] = 448.0f; hin[1] = 449.0f; hin[2] = 464.0f; hin[3] = 480.0f; hin[4] = 1e6f; hin[5] = -1e6f; hin[6] = INFINITY; hin[7] = 465.0f;
    hin[8] = 0.0009765625f; hin[9] = 0.001953125f; hin[10] = 0.0029296875f; hin[11] = 1e-9f; hin[12] = NAN;
    hipMemcpy(in, hin, n * 4, hipMemcpyHostToDevice);
    unsigned char hA[16 * K], hB[16 * K];
    for (int i = 0; i < 16 * K; ++i) {
        do hA[i] = rand() & 255; while ((hA[i] & 0x7f) == 0x7f || ((hA[i] >> 3) & 15) > 9);
        do hB[i] = rand() & 255; while ((hB[i] & 0x7f) == 0x7f || ((hB[i] >> 3) & 15) > 9);
    }
    hipMemcpy(A, hA, 16 * K, hipMemcpyHostToDevice); hipMemcpy(B, hB, 16 * K, hipMemcpyHostToDevice);
    cvt_kernel<<<1, 256>>>(dec, in, enc, n);
    mfma_kernel<<<1, 64>>>(A, B, K, d32, d128);
    float hdec[256], h32[256], h128[256];
    unsigned char henc[n];
    hipMemcpy(hdec, dec, 1024, hipMemcpyDeviceToHost); hipMemcpy(henc, enc, n, hipMemcpyDeviceToHost);
    hipMemcpy(h32, d32, 1024, hipMemcpyDeviceToHost); hipMemcpy(h128, d128, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < 256; ++b) {
        const float r = e4m3_to_f32((unsigned char)b);
        if (!((isnan(r) && isnan(hdec[b])) || r == hdec[b])) { if (bad++ < 8) printf("decode 0x%02x: hw %g table %g\n", b, hdec[b], r); }
    }
    printf("decode: %d of 256 bytes differ from the OCP e4m3fn table\n", bad);
    for (int i = 0; i < 13; ++i) printf("encode %-14g -> 0x%02x (%g)\n", hin[i], henc[i], e4m3_to_f32(henc[i]));
    // round-to-nearest-even check against a host quantizer
    int bad_enc = 0;
    for (int i = 13; i < n; ++i) {
        float best = 1e30f; int bb = 0;
        for (int b = 0; b < 256; ++b) {
            const float r = e4m3_to_f32((unsigned char)b);
            if (isnan(r)) continue;
            const float d = fabsf(r - hin[i]);
            if (d < best || (d == best && !(b & 1) && (bb & 1))) { best = d; bb = b; }
        }
        if (e4m3_to_f32(henc[i]) != e4m3_to_f32((unsigned char)bb)) { if (bad_enc++ < 8) printf("encode %g: hw 0x%02x nearest 0x%02x\n", hin[i], henc[i], bb); }
    }
    printf("encode: %d of %d values differ from round-to-nearest-even\n", bad_enc, n - 13);
    double e32 = 0, e128 = 0, mx = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)e4m3_to_f32(hA[i * K + k]) * e4m3_to_f32(hB[j * K + k]);
            e32 = fmax(e32, fabs(s - h32[i * 16 + j])); e128 = fmax(e128, fabs(s - h128[i * 16 + j])); mx = fmax(mx, fabs(s));
        }
    printf("mfma 16x16x32 fp8 (2 per 16-byte fragment): max |err| %.3e of max |D| %.3e\n", e32, mx);
    printf("mfma_scale 16x16x128 f8f6f4 (32-byte fragment, scale 2^0): max |err| %.3e\n", e128);
    return 0;
}
