// Host-side launchers of the gfx950 kernels (one .hip file per family).  Internal to libskyeye_hip.so.
//
// Data layout in HBM (DESIGN.md "Layout"): activations are NHWC with an explicit pixel stride `ld` (elements)
// so that a tensor can be a channel slice of a wider concat buffer; element type T is float (exact mode) or
// bf16 (production).  Convolution weights are [Cout_pad][Kpad] with K = (ky, kx, cin) contiguous, BatchNorm
// folded, rows padded to the N tile and K padded to one 128-byte K-step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sky {

enum Act { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2 };

// Developer A/B switches.  They are read from the environment ONCE, by sky_plan (engine.cpp: read_plan_opts), stored with the
// plan, reported by sky_op_info and handed to the launchers as a bit set: no launcher calls getenv, so an environment variable
// that appears after planning cannot re-route a planned graph.  None is needed in production; the parity tests use them to
// force a kernel path (tests/test_gpu_conv_halo.py, test_gpu_determinism.py).
enum PlanOpt : unsigned {
    OPT_HALO_OFF = 1u << 0,          // SKY_CONV_HALO=0      halo-tile kernels never
    OPT_HALO_FORCE = 1u << 1,        // SKY_CONV_HALO=force  halo-tile kernels whenever the shape is covered
    OPT_NF8_OFF = 1u << 2,           // SKY_HALO_NF8=off     64-channel tiles everywhere
    OPT_NF8_SOLO = 1u << 3,          // SKY_HALO_NF8=solo    128-channel tiles alone on a CU
    OPT_S2_OFF = 1u << 4,            // SKY_HALO_S2=0        stride-2 layers on the streaming kernel
    OPT_NO_STREAM = 1u << 5,         // SKY_NO_STREAM        implicit-GEMM tile kernel for everything
    OPT_NO_RING = 1u << 6,           // SKY_NO_RING
    OPT_OLDGRID = 1u << 7,           // SKY_STREAM_OLDGRID
    OPT_NO_FUSED_IMPORT = 1u << 8,   // SKY_NO_FUSED_IMPORT  stem reads an imported tensor instead of the raw frames
    OPT_FUSE = 1u << 9,              // SKY_FUSE=1           1x1 convolutions in the producer's epilogue (measured neutral)
    OPT_NO_SPP_PYRAMID = 1u << 10,   // SKY_NO_SPP_PYRAMID
    OPT_ATTN_VALU = 1u << 11,        // SKY_ATTN_VALU        exact attention core for bf16 too
    OPT_NO_FUSE_CV1 = 1u << 12,      // SKY_NO_FUSE_CV1      bottleneck cv1 as its own launch (default: fused into the 3x3 where covered)
    OPT_NO_STEM_DOWN = 1u << 13,     // SKY_NO_STEM_DOWN     stem and first stride-2 convolution as two launches (default: one kernel where covered)
    OPT_NO_WINATTN = 1u << 14,       // SKY_NO_WINATTN       8 x 8 windows on the general flash kernel (default: the one-wave-per-(window, head) kernel)
    OPT_NO_CSP_STAGE = 1u << 15,     // SKY_NO_CSP_STAGE     the first CSP stage as four launches (default: one kernel where covered)
    OPT_NO_HEAD_STREAM = 1u << 21,   // SKY_NO_HEAD_STREAM   detection levels on the implicit-GEMM tile kernel (default: the streaming head kernel)
    OPT_HEAD_STREAM_FORCE = 1u << 22,// SKY_HEAD_STREAM=force the streaming head kernel on small levels too (default: only where 32-pixel steps fill the device)
    OPT_NO_BNECK128 = 1u << 23,      // SKY_NO_BNECK128      128-channel bottlenecks as two launches (default: one kernel, k_bneck.hip)
    OPT_NO_DEEP3X3 = 1u << 24,       // SKY_NO_DEEP3X3       wide 3x3 stride-1 layers on the halo-tile kernel (default: k_conv3x3_deep.hip where covered)
    OPT_NO_IN2 = 1u << 25,           // SKY_NO_IN2           neck concat buffers materialised (default: the upsampled half is read from the small map)
    OPT_NO_CV3_HEAD = 1u << 26,      // SKY_NO_CV3_HEAD      fpn_conv3.cv3 and detection level 0 as two launches (default: one kernel, k_head.hip)
    OPT_NO_GEMM1X1 = 1u << 27,       // SKY_NO_GEMM1X1       large-K 1x1 convolutions on the streaming kernel (default: the LDS-DMA GEMM, k_gemm1x1.hip)
    OPT_GEMM1X1_FORCE = 1u << 28,    // SKY_GEMM1X1=force    the GEMM wherever the shape is covered (default: K >= 256 and two tiles per CU or more)
    OPT_BNECK128_SOLO = 1u << 29,    // SKY_BNECK128=solo    128-channel bottlenecks on round 3's one-workgroup-per-CU kernel (default: two 4-wave workgroups per CU, k_bneck_w.hip)
    OPT_BNECK_PAIR = 1u << 30,       // SKY_BNECK128=pair    fused bottlenecks keep their plan (buffers, scale carriers) but run as their two launches (tests: the fp8 identity needs equal scales)
    OPT_NO_BNECK64W = 1u << 31,      // SKY_NO_BNECK64W      64-channel bottlenecks on the halo-tile kernel's fused form (default: three 4-wave workgroups per CU, k_bneck_w64.hip)
    OPT_SKIP_SHIFT = 16,             // SKY_HALO_SKIP=<bits> bisection: bit 0 stride-1, 1 stride-2, 2 narrow, 3 128-ch, 4 64-ch tiles
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: `state` is a function-local static of each launcher,
// one slot per device ordinal, holding the largest LDS size already granted there.
inline hipError_t ensure_lds_attr(const void* kern, size_t lds, int device, size_t (&state)[16])
{
    const int d = device >= 0 && device < 16 ? device : 0;
    if (lds <= state[d] && state[d] != 0) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) state[d] = lds;
    return e;
}

// One fused convolution launch: out = act(conv(in, w) + bias) (+ res), optionally written 2x-upsampled
// (nearest) or decoded as a detection level.
struct ConvArgs {
    const void* in;      // T, NHWC, channel offset already applied
    const void* w;       // T, [rows][Kpad]
    const float* bias;   // fp32 [>= ntiles*BN]
    void* out;           // T (or fp32 when out_f32), NHWC view
    const void* res;     // optional residual, same geometry as out
    const void* zero;    // >= 16 bytes of zeros in device memory (target of masked loads)
    int B, H, W, Cin, ldi;
    int Ho, Wo, Cout, ldo, ldr;
    int ks, stride, pad;
    int Kpad;            // elements
    int act;
    int up2;             // write each output pixel to the 2x2 block of a (2Ho, 2Wo) image
    int out_f32;
    int M;               // B*Ho*Wo
    int ntiles;          // N tiles
    int cpt_shift;       // log2(16-byte chunks per tap) or -1 (streaming kernel)
    int tile_w, tile_h;         // halo-tile kernels: output tile shape (tile_w * tile_h <= 256)
    unsigned magic_w, magic_h;  // halo-tile kernels: 2^16 / tile_w + 1, 2^16 / (tile_w + 2) + 1 (exact division of small indices)
    int dbg;             // kernel experiments (SKY_CONV_DBG): only read by builds with -DSKY_EXPERIMENTS, ignored otherwise
    // fp8 engine (SKY_FP8): real value = stored byte * per-tensor scale.  acc * mult[cout] (= input scale * weight scale of the
    // output channel) + bias, activation, + residual * res_scale, * out_inv_scale, saturate, e4m3.  mult == nullptr: 1.
    const float* mult;
    float out_inv_scale, res_scale;
    int out_dt;          // element type of out / res (sky_dtype); < 0: the compute type.  Lets a bf16 convolution (the stem) write fp8
    unsigned opts;       // PlanOpt bits of the plan
    int device, n_cu;    // device ordinal of the plan and its CU count (per-device launch geometry / LDS attributes)
    int src_mode;        // 0: `in` is an NHWC tensor of T.  1 / 2: `in` is the caller's raw [B, 3, 2H, 2W] uint8 / float32 NCHW
                         // frame batch and FocusBlock's space-to-depth + /255 + conversion are fused into the convolution's
                         // loader (narrow-input halo kernel only; conv_accepts_raw() tells whether a launch would take it)
    // optional second, fused 1x1 convolution (back-to-back GEMM in the epilogue): out2 = act2(W2 * out[:, koff:koff+cin2] + b2)
    // where `out` is this convolution's own (packed) output; honoured only by kernels that hold all Cout channels of a
    // pixel in one workgroup -- the launcher reports through *fused whether it was
    const void* f2_w;    // T [cout2 rows][f2_Kpad], null = nothing to fuse
    const float* f2_bias;
    void* f2_out;
    int f2_cin, f2_cout, f2_ldo, f2_act, f2_Kpad, f2_koff;
    unsigned f2_out_bytes;
    // fused cv1 of a BottleneckBlock (halo-tile kernel, CV1): `in` is the bottleneck's input x, u = SiLU(c1_w x + c1_bias) is computed
    // on the halo tile in LDS and the 3x3 runs on u; c1_res: add x (the shortcut, blocks.py:88-90), taken from the LDS tile
    const void* c1_w;    // T [Cin rows][c1_Kpad], BatchNorm folded, null = not fused
    const float* c1_bias;
    int c1_Kpad, c1_res;
    // fp8 engine (k_bneck_w8.hip): u = e4m3(SiLU(acc * c1_mult[ch] + c1_bias[ch]) * c1_out_inv_scale); `mult` then holds hidden scale x weight scale
    const float* c1_mult;
    float c1_out_inv_scale;
    unsigned out_bytes, res_bytes;   // extents of the output / residual views in bytes (0 = 2 GiB or more)
    unsigned in_bytes;   // extent of the input view in bytes (buffer descriptor range; 0 = 2 GiB or more: not addressable with int32 offsets)
    // optional second input of a 1x1 convolution (streaming kernel): the first in2_cin channels of K come from `in2`, read with the
    // nearest-2x upsampling of FeatureNeck (detector.py:214,218: output pixel (y, x) <- in2 pixel (y >> 1, x >> 1)) when in2_up2, the
    // remaining Cin - in2_cin channels from `in` -- cat([up(lateral(p)), q]) is then never materialised for its first half
    const void* in2;
    int in2_cin, ldi2, in2_up2;
    unsigned in2_bytes;
    // detection-level epilogue (DetectionHead.forward + process_detections, detector.py:61-145)
    int head;
    float* raw;          // [B, na, gh, gw, no] fp32
    float* det;          // [B, det_rows, no] fp32
    int na, no;
    long det_rows;       // rows per image over all levels
    long det_off;        // first row of this level
    float stride_px;     // max(H_in/gh, W_in/gw)
    float anchor_wh[16]; // anchors[level][a][w,h] * stride_px
};

// sigmoid of the detection decode (detector.py:131): IEEE division + expf in the exact (fp32) engine; v_exp_f32 / v_rcp_f32
// (1 ulp each, far inside the 1e-4 decode tolerance) in the bf16 engine, where the epilogue is VALU-bound.  The fused head
// epilogue and the standalone process_detections kernel use the same form per engine: they stay equal bit for bit.
#if defined(__HIPCC__)
template <bool FAST>
__device__ __forceinline__ float head_sigmoid(float v)
{
    if (FAST) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
    return 1.0f / (1.0f + expf(-v));
}
#endif

inline int dtype_size(int dtype) { return dtype == 0 ? 4 : dtype == 1 ? 2 : 1; }   // SKY_F32, SKY_BF16, SKY_FP8
int conv_pick_bn(int cout);                      // N tile chosen for a given Cout
size_t conv_weight_rows(int cout);               // rows the packed weight / bias must have
int conv_k_step(int dtype);                      // elements per 128-byte K-step
// *variant (optional) receives which kernel ran: 1000 + N tile = implicit-GEMM tile kernel, 2000 + N_blk = streaming kernel
// with resident weights, 3000 + N_blk = streaming kernel with the weight ring, 4000 + N_blk = halo-tile 3x3 kernel
hipError_t launch_conv(int dtype, const ConvArgs& a, hipStream_t s, int* variant = nullptr, int* fused = nullptr);
// streaming path (k_conv_stream.hip): weights resident in LDS; hipErrorNotSupported when the shape is not covered
hipError_t launch_conv_stream(int dtype, const ConvArgs& a, hipStream_t s, int* variant = nullptr, int* fused = nullptr);
// halo-tile path (k_conv_halo.hip): 3x3 stride 1 with >= 128 bytes of input channels, input tile staged once in LDS;
// variant 4000 + N_blk; hipErrorNotSupported when the shape is not covered.  SKY_CONV_HALO=0 disables, =force ignores
// the tile-fill heuristic.
hipError_t launch_conv_halo(int dtype, const ConvArgs& a, hipStream_t s, int* variant = nullptr, int* fused = nullptr);
// would launch_conv run this (3x3, stride 1, 16-channel) convolution on the kernel that reads raw frames (ConvArgs::src_mode)?
bool conv_accepts_raw(int dtype, const ConvArgs& a);
// would launch_conv run this 1x1 convolution with a second input (ConvArgs::in2*)?  (plan-time query)
bool conv_accepts_in2(int dtype, const ConvArgs& a);
// would launch_conv run this 3x3 convolution with the preceding 1x1 (ConvArgs::c1_*) fused on the halo tile?  (plan-time query:
// the engine then emits one launch for the cv1 + cv2 pair of a BottleneckBlock and gives it an output that does not alias x)
bool conv_accepts_cv1(int dtype, const ConvArgs& a);

// ---- stem + first stride-2 convolution in one kernel (k_stem_down.hip; bf16, uint8 NCHW frames, 3 -> 32 -> 64 channels) ----
struct StemDownArgs {
    const unsigned char* frames;   // [B, 3, Hr, Wr] uint8
    int B, Hr, Wr;                 // raw frame size; the stem map is Hr/2 x Wr/2
    const void* w1;                // stem weights, bf16 [32 rows][kpad1], K = (tap, 16 stored channels of which 12 are real), BN folded
    const float* bias1;
    int kpad1;
    const void* w2;                // stride-2 convolution, bf16 [64 rows][kpad2], K = (tap, 32)
    const float* bias2;
    int kpad2;
    void* out;                     // bf16 NHWC view [B, Ho, Wo, 64] with pixel stride ldo
    int Ho, Wo, ldo;
    int c1, c2;
    unsigned opts;
    int device, n_cu;
};
bool stem_down_supported(const StemDownArgs& a);
hipError_t launch_stem_down(const StemDownArgs& a, hipStream_t s);

// CSPBlock(c, c, n = 1) with hidden = c / 2 as one kernel (k_csp_stage.hip); weights in the engine's packing, BN folded
struct CspStageArgs {
    const void* in;                // bf16 NHWC view [B, H, W, c] with pixel stride ldi
    int ldi;
    unsigned in_bytes;             // extent of that view (32-bit buffer offsets); 0 = too large
    void* out;                     // bf16 NHWC view [B, H, W, c] with pixel stride ldo (another buffer than `in`)
    int ldo;
    int B, H, W, c, hidden;
    const void* w12;               // cv1 | cv2 stacked: [c rows][kpad12], K = c
    const void* wb1;               // bottleneck cv1: [hidden][kpadb1], K = hidden
    const void* wb2;               // bottleneck cv2 (3x3): [hidden][kpadb2], K = (tap, hidden)
    const void* w3;                // cv3: [c][kpad3], K = (bottleneck output | cv2 output)
    const float *b12, *bb1, *bb2, *b3;
    int kpad12, kpadb1, kpadb2, kpad3;
    int shortcut;                  // the bottleneck adds its input (blocks.py:88-90)
    unsigned opts;
    int device, n_cu;
};
// detection level (1x1 + bias + decode) as a byte streamer, bf16 (k_head.hip); hipErrorNotSupported: take the tile kernel
// CSP cv3 (1x1 128 -> 128) + the detection level that reads its output in one kernel (k_head.hip): `a` = the cv3 convolution, the level
// in f2_w / f2_bias / f2_Kpad + the head fields
bool cv3_head_supported(int dtype, const ConvArgs& a);
hipError_t launch_cv3_head(int dtype, const ConvArgs& a, hipStream_t s);
// 1x1 convolutions with K >= 256 and Cout % 256 == 0 as a 128 x 256-tile GEMM fed by LDS-DMA (k_gemm1x1.hip); reads ConvArgs::in2 too
bool gemm1x1_ok(int dtype, const ConvArgs& a);
hipError_t launch_gemm1x1(int dtype, const ConvArgs& a, hipStream_t s, int* variant);
bool head_stream_supported(int dtype, const ConvArgs& a);
hipError_t launch_head_stream(int dtype, const ConvArgs& a, hipStream_t s, int* variant);

// BottleneckBlock(128, 128) as one kernel (k_bneck.hip; bf16): ConvArgs of the 3x3 with c1_w / c1_bias / c1_Kpad / c1_res set
bool bneck128_shape_ok(const ConvArgs& a);
hipError_t launch_bneck128(const ConvArgs& a, hipStream_t s);
// the same block as two 4-wave workgroups per CU on 8 x 16 tiles (k_bneck_w.hip; round 4, the default)
// BottleneckBlock(64, 64), bf16, 8 x 16 tiles, three workgroups per CU (k_bneck_w64.hip)
bool bneck64w_shape_ok(const ConvArgs& a);
hipError_t launch_bneck64w(const ConvArgs& a, hipStream_t s);
// BottleneckBlock(64, 64) of the fp8 engine: 64-byte pixels, taps paired into 16x16x128 instructions (k_bneck_w64f8.hip)
bool bneck64w8_shape_ok(const ConvArgs& a);
hipError_t launch_bneck64w8(const ConvArgs& a, hipStream_t s);
// the same bottleneck in the fp8 engine (k_bneck_w8.hip): 128 channels = one 128-byte chunk, 16x16x128 block-scaled instructions
bool bneck128w8_shape_ok(const ConvArgs& a);
hipError_t launch_bneck128w8(const ConvArgs& a, hipStream_t s);
bool bneck128w_shape_ok(const ConvArgs& a);
hipError_t launch_bneck128w(const ConvArgs& a, hipStream_t s);

// deep-pipelined 3x3 stride 1 for Cin a multiple of 256 (k_conv3x3_deep.hip; bf16): variant 4600 + 128
bool conv3x3_deep_ok(int dtype, const ConvArgs& a);
hipError_t launch_conv3x3_deep(int dtype, const ConvArgs& a, hipStream_t s);

bool csp_stage_supported(const CspStageArgs& a);
hipError_t launch_csp_stage(const CspStageArgs& a, hipStream_t s);

// ---- layout / glue kernels (k_misc.hip) ----
// boundary conversion: caller tensor (NCHW/NHWC, fp32/u8) -> engine NHWC T with C padded to Cpad (zeros);
// s2d = FocusBlock's space-to-depth (blocks.py:176-181, order TL, BL, TR, BR); scale255 = x/255 (validate.py:238)
hipError_t launch_import(int dtype, const void* src, int src_u8, int src_nhwc, void* dst, int B, int C, int H, int W,
                         int Cpad, int ld, int s2d, int scale255, hipStream_t s, float out_inv_scale = 1.0f);
// engine NHWC T -> caller NCHW fp32
hipError_t launch_export(int dtype, const void* src, int ld, float* dst, int B, int C, int H, int W, hipStream_t s, float in_scale = 1.0f);
// max |x| over an NHWC view into *amax (float bits, atomicMax): activation-scale calibration of the fp8 engine
hipError_t launch_amax(int dtype, const void* x, int ld, long pixels, int C, unsigned int* amax, hipStream_t s);
// MaxPool2d(5, stride 1, pad 2), -inf padding (blocks.py:142-144; 9 = 5o5, 13 = 5o5o5 exactly)
hipError_t launch_spp_pyramid(int dtype, const void* src, int lds_, void* dst, int ldd, int B, int H, int W, int C, int level_stride,
                              hipStream_t s);
hipError_t launch_maxpool5(int dtype, const void* src, int lds_, void* dst, int ldd, int B, int H, int W, int C,
                           hipStream_t s);
// F.interpolate(mode='nearest') (detector.py:214,218), generic sizes
hipError_t launch_upsample(int dtype, const void* src, int lds_, void* dst, int ldd, int B, int H, int W, int C, int Ho,
                           int Wo, hipStream_t s);

// letterbox (augmentation.py:442-496): uint8 HWC frame -> resized (nh, nw, bilinear, OpenCV's 8-bit fixed point) + constant
// border, written HWC or CHW (optionally with reversed channel order)
hipError_t launch_letterbox(const unsigned char* src, int H0, int W0, unsigned char* dst, int H1, int W1, int nh, int nw, int top, int left,
                            int pad, int chw, int rev, hipStream_t s);

// DetectionHead.process_detections alone (detector.py:88-145): raw [B,na,gh,gw,no] -> det rows of one level
hipError_t launch_decode(int dtype, const float* raw, float* det, int B, int na, int gh, int gw, int no, long det_rows, long det_off,
                         float stride_px, const float* anchor_wh /*host, na*2, already * stride*/, hipStream_t s);

// ---- CBAM (attention.py:11-130) ----
// partial per-channel sum / max over pixel chunks: part[b][chunk][2][C]
hipError_t launch_ca_reduce(int dtype, const void* x, int ld, int B, int HW, int C, int nchunk, float* part,
                            hipStream_t s, float in_scale = 1.0f);
// finish reduce + shared MLP (no bias, ReLU) + sigmoid -> att[b][C]
hipError_t launch_ca_mlp(const float* part, int B, int HW, int C, int nchunk, int R, const float* w0, const float* w2,
                         float* att, hipStream_t s);
// per-pixel mean / max over channels of (x * att) -> stats[b][p][2]; att may be null (plain SpatialAttention)
hipError_t launch_sa_stats(int dtype, const void* x, int ld, const float* att, int B, int HW, int C, float* stats,
                           hipStream_t s, float in_scale = 1.0f);
// 7x7 conv (2->1, pad 3, no bias) + sigmoid -> gate[b][p]
hipError_t launch_sa_gate(const float* stats, const float* w, int B, int H, int W, float* gate, hipStream_t s);
// out = (x * att) * gate   (either factor may be null)
hipError_t launch_scale(int dtype, const void* x, int ldx, const float* att, const float* gate, void* out, int ldo,
                        int B, int HW, int C, hipStream_t s, float in_scale = 1.0f, float out_inv_scale = 1.0f);

// ---- attention side (k_attn.hip) ----
hipError_t launch_layernorm(int dtype, const void* x, int ldx, void* y, int ldy, const float* g, const float* b, long tokens, int C,
                            hipStream_t s);
// qkv [G, N, 3C] -> out [G, N, C]; bias [heads, N, N] and mask [nW, N, N] optional (fp32).  ws > 0: the G groups are
// the ws x ws windows of a [B, mh, mw] NHWC map addressed in place (N = ws * ws)
hipError_t launch_attention(int dtype, const void* qkv, int ldq, void* out, int ldo, int G, int N, int C, int heads, float scale,
                            const float* bias, const float* mask, int nW, int ws, int mh, int mw, hipStream_t s, unsigned opts = 0);
// CrossLayerAttention core: q [B,H,W,C], kv [B,h,w,2C] (v at channel offset v_off), scores scratch [B,H,W,heads] fp32
hipError_t launch_cla(int dtype, const void* q, int ldq, const void* kv, int ldkv, int v_off, float* scores, void* out, int ldo, int B, int H,
                      int W, int h, int w, int C, int heads, float scale, float r2, hipStream_t s);
hipError_t launch_export_tokens(int dtype, const void* src, int ld, float* dst, long tokens, int C, hipStream_t s);

// ---- NMS (k_nms.hip) ----
struct NmsArgs {
    const float* det;   // [B, N, no]
    int B, N, nc;
    float conf, iou, max_wh;
    int agnostic, multi_label, max_det, max_nms, mode;
    int n_classes;
    int classes[64];
    float* out;         // [B, max_det, 7], image blocks out_stride floats apart
    int* counts;        // [B], counts_stride ints apart
    long out_stride;    // floats between the row blocks of consecutive images (>= max_det * 7)
    long counts_stride; // ints between consecutive counts (>= 1)
    // workspace
    int* blk_counts;    // [B, nblk]
    int* totals;        // [B]
    unsigned long long* keys;  // [B, cap]
    unsigned long long* keys2; // [B, cap]: the other buffer of the merge levels
    float* cand;        // [B, cap, 4]  (score, conf, class, src row as int bits)
    long cap;           // power of two >= N * (multi_label ? nc : 1)
    int device;         // device ordinal of the handle (per-device LDS attribute)
};
size_t nms_workspace_bytes(int B, int N, int nc, int multi_label, long* cap_out);
hipError_t launch_nms(const NmsArgs& a, hipStream_t s);
// pairwise IoU [n, m] of the evaluation accounting (metrics.py:17-44); box1 [4, n] (box1_4xn) or [n, 4], box2 [m, 4]
hipError_t launch_box_iou(const float* box1, int n, int box1_4xn, const float* box2, int m, float* out, hipStream_t s);

// k_tta.hip: test-time augmentation / tiling front end
hipError_t launch_scale_img(const void* src, int src_u8, int planes, int H, int W, float* dst, int sh, int sw, int PH, int PW, int flip, float pad,
                            hipStream_t s);
hipError_t launch_map_detections(const float* src, int B, int N, int no, int row0, int rows, float scale, int flip, float img_h, float img_w,
                                 const int* origins, int tpi, float* dst, long dst_rows, long dst_row0, hipStream_t s);
hipError_t launch_offset_boxes(float* rows, const int* counts, int T, int R, int cols, const int* origins, hipStream_t s);
hipError_t launch_tile_gather(const unsigned char* src, int H0, int W0, int src_chw, const int* origins, int n, unsigned char* dst, int th, int tw,
                              int pad, int rev, hipStream_t s);

}  // namespace sky
