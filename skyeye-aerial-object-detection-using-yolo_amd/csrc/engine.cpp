// libskyeye_hip.so -- graph planner, weight preparation and the C ABI (include/skyeye_hip.h).
//
// The engine mirrors the reference's module tree (blocks.py / attention.py / backbone.py / detector.py) as a set
// of builder functions.  A builder run does two things at once: it records the state-dict entries the module
// expects (names as in the reference) and -- when a geometry is given -- emits a static list
// of fused kernel launches over one workspace arena:
//   * ConvolutionBlock  -> one implicit-GEMM launch, BatchNorm folded into the packed weights, SiLU in the epilogue
//   * CSPBlock          -> cv1 and cv2 share one GEMM (N = 2h) that writes the concat buffer; bottlenecks update
//                          the first half in place; torch.cat never materialises
//   * FeatureNeck       -> lateral convs write their output 2x-upsampled straight into the concat slice
//   * SPPBlock          -> cascade of 5x5 max-pools writing concat slices
//   * DetectionHead     -> 1x1 GEMM whose epilogue writes raw logits and decoded boxes
// Buffers are placed in the arena by lifetime (first/last use), so the working set stays small enough for the
// 256 MiB Infinity Cache to serve producer->consumer traffic at small batch.
#include "../../include/skyeye_hip.h"
#include "sky_kernels.h"
#include "build_hash.h"

#include <cstddef>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace sky {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define SKY_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            throw Error(SKY_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                \
    } while (0)

static std::string g_create_error;

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
    int64_t numel() const
    {
        int64_t n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};

struct ParamSpec {
    std::string name;
    std::vector<int64_t> shape;
};

// view of an activation tensor inside a workspace buffer (or a caller buffer when ext >= 0)
struct TV {
    int buf = -1;
    long off = 0;   // element offset (channel offset inside the pixel)
    int B = 0, H = 0, W = 0, C = 0, ld = 0;
    int ext = -1;   // 0..15 input slot, 16.. output slot
    int dt = 0;     // element type (sky_dtype) of a workspace tensor
    bool valid() const { return buf >= 0 || ext >= 0; }
};

enum OpKind { OP_IMPORT, OP_EXPORT, OP_CONV, OP_MAXPOOL5, OP_UPSAMPLE, OP_CA_REDUCE, OP_CA_MLP, OP_SA_STATS, OP_SA_GATE, OP_SCALE, OP_DECODE, OP_LAYERNORM, OP_ATTENTION, OP_CLA, OP_EXPORT_TOKENS };

struct Op {
    OpKind kind;
    TV in, out, res;
    // conv
    int wid = -1;          // index into Engine::convs
    int wid1 = -1;         // fused BottleneckBlock: packed weights of cv1 (the 1x1 computed on the halo tile of the 3x3, k_conv_halo.hip CV1)
    int c1_res = 0;        // ... with the shortcut x + cv2(cv1(x))
    TV hid;                // ... fp8 plans: the hidden tensor cv1(x) as a SCALE CARRIER (a workspace buffer nobody allocates; the calibration twin writes the real one)
    int stem_down = 0;     // this FocusBlock convolution and the stride-2 convolution behind it can run as ONE kernel (k_stem_down.hip)
    int csp_stage = 0;     // cv1|cv2 of a CSPBlock whose whole stage (this op and the next three) can run as ONE kernel (k_csp_stage.hip)
    int csp_member = 0;    // one of those next three ops: skipped at run time when the stage kernel ran
    int csp_shortcut = 0;
    int head_op = -1;      // a 1x1 128 -> 128 convolution whose output a detection level reads: index of that level's op (cv3 + level in one kernel, k_head.hip)
    int cin = 0, cout = 0, ks = 1, stride = 1, act = 0, up2 = 0;
    int head = 0, level = 0;
    int raw_ext = -1, det_ext = -1;
    long det_rows = 0, det_off = 0;
    float stride_px = 0;
    // import
    int s2d = 0, src_c = 0, src_h = 0, src_w = 0;
    // scratch buffers (fp32), buffer ids
    int s0 = -1, s1 = -1, s2 = -1;
    int nchunk = 0, R = 0;
    int f0 = -1, f1 = -1;  // fp32 device weights (index into Engine::fweights)
    int cdt = 0;           // compute type of a convolution (= element type of its input; the packed weights have it too)
    int Ho = 0, Wo = 0;
    double flops = 0;
    int variant = 0;       // which conv kernel ran last (launch_conv)
    int fuse_next = 0;     // the next op is a 1x1 convolution of (a channel slice of) this op's output: candidate for the
                           // producer's epilogue (back-to-back GEMM from the registers that hold the packed output)
    int fused_prev = 0;    // set on that next op; skipped at run time when the producer's kernel took it
    // attention
    int heads = 0, ntok = 0, nW = 0, mask_ext = -1, v_off = 0, force_nhwc = 0;
    int win = 0;           // attention over the win x win windows of the input map (0: over all H*W tokens of each image)
    float scale = 1.0f, r2 = 1.0f;
    TV in2;                // second activation input (CLA key/value tensor; 1x1 convolution: the first in2_cin channels of K, ConvArgs::in2)
    int in2_cin = 0, in2_up2 = 0;
};

struct DevConv {
    void* w = nullptr;
    float* bias = nullptr;
    float* mult = nullptr;           // fp8: per-output-channel input scale x weight scale (device, [rows]); set by apply_scales
    std::vector<float> w_scale;      // fp8: per-output-channel weight scale (host, [rows])
    int cdt = 0;                     // element type of the packed weights
    float pre = 1.0f;                // bf16 SiLU layers: log2(e), the factor weights and bias were packed with (exp2-domain activation)
    int Kpad = 0;
    size_t bytes = 0;
    // description for sky_packed_* (export of the engine's own weight file)
    int rows = 0, cout = 0, ks = 0, cin = 0;
    std::string name;      // state-dict name of the (first) source weight
};

struct Buffer {
    size_t bytes = 0;
    int first = 1 << 30, last = -1;
    size_t offset = 0;
    int dt = 0;          // element type (sky_dtype)
    float scale = 1.0f;  // fp8 buffers: real value = stored value * scale (calibrated, sky_calibrate); 1 otherwise
    int tie = -1;        // buffer whose scale this one must share (pooling / upsampling move bytes), -1 = itself
};

struct IoInfo {
    int ndim = 0;
    int64_t shape[5] = {0, 0, 0, 0, 0};
};

struct Engine {
    sky_config cfg;
    int dtype = 0;
    std::string err;
    std::map<std::string, HostTensor> weights;
    std::vector<ParamSpec> spec;
    // plan
    bool planned = false;
    bool weights_dirty = true;
    std::vector<Op> ops;
    std::vector<Buffer> bufs;
    std::vector<DevConv> convs;
    std::vector<float*> fweights;
    std::vector<void*> owned;   // device allocations to free on re-plan
    char* arena = nullptr;
    size_t arena_bytes = 0;
    void* zero_page = nullptr;   // 256 bytes of zeros: masked loads of the conv kernel point here
    std::vector<IoInfo> in_info, out_info;
    std::vector<sky_buffer> plan_inputs;
    double flops = 0, act_bytes = 0, weight_bytes = 0;
    // nms workspace
    void* nms_ws = nullptr;
    size_t nms_ws_bytes = 0;
    unsigned opts = 0;      // PlanOpt bits, read from the environment once per sky_plan
    unsigned extra_opts = 0;   // PlanOpt bits forced by the engine itself (the calibration twin plans without fused bottlenecks)
    // Sub-batch section: ops [0, sec_end) -- the high-resolution head of the detector graph, whose per-layer tensors of a whole
    // batch (0.2 - 0.8 GB at B = 32) are far beyond the 256 MiB Infinity Cache -- run slice by slice of `sec_sub` frames, all section
    // ops for one slice before the next slice, so that a layer's output is still cache-resident when the next layer reads it.
    int sec_end = 0, sec_sub = 0;
    int n_cu = 256;         // compute units of cfg.device

    bool calibrated = false;    // fp8: activation scales are set
    unsigned mask_opts = 0;     // PlanOpt bits this plan must not take from the environment (the calibration twin of an fp8 plan)
    // the calibration twin of an fp8 plan: element type and PlanOpt bits of the plan it measures for.  Decisions that change the BUFFER LIST (a small
    // lateral map read in place by its consumer, ConvArgs::in2) are taken as that plan takes them; the twin then materialises the tensor anyway
    int mirror_dtype = -1;
    unsigned mirror_opts = 0;
    int esize() const { return dtype_size(dtype); }
    int epc() const { return 16 / dtype_size(dtype); }
    void free_plan()
    {
        for (void* p : owned) (void)hipFree(p);
        owned.clear();
        if (arena) (void)hipFree(arena);
        if (zero_page) (void)hipFree(zero_page);
        zero_page = nullptr;
        arena = nullptr;
        arena_bytes = 0;
        ops.clear();
        bufs.clear();
        convs.clear();
        fweights.clear();
        planned = false;
    }
    ~Engine()
    {
        free_plan();
        if (nms_ws) (void)hipFree(nms_ws);
    }
};

static unsigned short f32_to_bf16(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// f32 -> e4m3fn, round to nearest even, saturating at +-448 (the device epilogues do the same: clamp, then v_cvt_pk_fp8_f32)
static unsigned char f32_to_e4m3(float f)
{
    if (std::isnan(f)) return 0x7f;
    const unsigned char sgn = std::signbit(f) ? 0x80 : 0;
    float a = std::fabs(f);
    if (a >= 448.0f) return sgn | 0x7e;
    if (a < std::ldexp(1.0f, -10)) return sgn;                       // below half of the smallest subnormal (2^-9): zero
    int ex;
    (void)std::frexp(a, &ex);                                        // a = m * 2^ex, m in [0.5, 1)
    int e = ex - 1;                                                  // a = (1.x) * 2^e
    if (e < -6) e = -6;                                              // subnormal range: fixed exponent
    const float q = std::ldexp(a, 3 - e);                            // in units of the mantissa step (8 steps per binade)
    float r = std::nearbyint(q);                                     // FE_TONEAREST: ties to even
    int mant = (int)r;
    if (e == -6 && mant < 8) return sgn | (unsigned char)mant;       // subnormal (exponent field 0)
    if (mant == 16) { mant = 8; ++e; }
    if (e > 8 || (e == 8 && mant - 8 > 6)) return sgn | 0x7e;
    return sgn | (unsigned char)(((e + 7) << 3) | (mant - 8));
}

// ------------------------------------------------------------------------------------------------ builder context
struct Ctx {
    Engine& e;
    bool emit;   // false: only collect the parameter spec
    std::map<std::string, bool> seen;

    explicit Ctx(Engine& eng, bool em) : e(eng), emit(em) {}

    void need(const std::string& name, std::vector<int64_t> shape)
    {
        if (!seen.count(name)) {
            seen[name] = true;
            e.spec.push_back({name, shape});
        }
        if (emit) {
            auto it = e.weights.find(name);
            if (it == e.weights.end()) throw Error(SKY_ERR_MISSING_WEIGHT, "missing weight '" + name + "'");
            if (it->second.shape != shape) {
                std::string s = "shape mismatch for '" + name + "': expected [";
                for (auto v : shape) s += std::to_string(v) + ",";
                s += "] got [";
                for (auto v : it->second.shape) s += std::to_string(v) + ",";
                throw Error(SKY_ERR_SHAPE, s + "]");
            }
        }
    }
    const HostTensor& W(const std::string& name) { return e.weights.at(name); }

    int new_buf(size_t bytes)
    {
        Buffer b;
        b.bytes = (bytes + 255) / 256 * 256;
        e.bufs.push_back(b);
        return (int)e.bufs.size() - 1;
    }
    TV new_tensor(int B, int H, int W_, int C, int dt = -1)
    {
        TV t;
        t.B = B; t.H = H; t.W = W_; t.C = C; t.ld = C;
        t.dt = dt < 0 ? e.dtype : dt;
        t.buf = emit ? new_buf((size_t)B * H * W_ * C * dtype_size(t.dt)) : 0;
        if (emit) e.bufs[t.buf].dt = t.dt;
        return t;
    }
    // the two buffers hold the same stored values (max-pooling, nearest upsampling move bytes): one scale for both
    void tie_scales(int a, int b)
    {
        if (!emit || a < 0 || b < 0 || a == b) return;
        auto root = [&](int i) { while (e.bufs[i].tie >= 0) i = e.bufs[i].tie; return i; };
        const int ra = root(a), rb = root(b);
        if (ra != rb) e.bufs[rb].tie = ra;
    }
    static TV slice(const TV& t, int c0, int C)
    {
        TV s = t;
        s.off = t.off + c0;
        s.C = C;
        return s;
    }
    void touch(int buf)
    {
        if (buf < 0) return;
        Buffer& b = e.bufs[buf];
        const int i = (int)e.ops.size();
        b.first = std::min(b.first, i);
        b.last = std::max(b.last, i);
    }
    void push(Op op)
    {
        if (!emit) return;
        // opt-in (SKY_FUSE=1): measured neutral on MI355X -- the producer's epilogue is VALU-bound, the second GEMM and
        // its epilogue cost what the separate 1x1 launch costs
        const bool no_fuse = !(e.opts & OPT_FUSE);
        if (!no_fuse && !e.ops.empty() && op.kind == OP_CONV) {
            Op& a = e.ops.back();
            const long koff = op.in.off - a.out.off;
            if (a.kind == OP_CONV && !a.head && !a.up2 && !op.head && !op.up2 && !op.res.valid() && op.ks == 1 && op.stride == 1 &&
                a.out.buf >= 0 && op.in.buf == a.out.buf && op.in.ld == a.out.ld && op.in.B == a.out.B && op.in.H == a.out.H &&
                op.in.W == a.out.W && koff >= 0 && koff + op.cin <= a.cout && koff % 32 == 0 && (op.cin == 32 || op.cin == 64) &&
                (op.cout == 32 || op.cout == 64) && (a.cout == 64 || a.cout == 128) && op.out.buf >= 0 && op.out.buf != a.in.buf &&
                op.out.buf != a.out.buf && op.out.buf != a.res.buf && !a.fused_prev) {
                a.fuse_next = 1;
                op.fused_prev = 1;
                const int i = (int)e.ops.size();
                Buffer& b = e.bufs[op.out.buf];          // written while the producer runs: alive one op earlier
                b.first = std::min(b.first, i - 1);
            }
        }
        touch(op.in.buf); touch(op.out.buf); touch(op.res.buf); touch(op.in2.buf);
        touch(op.s0); touch(op.s1); touch(op.s2);
        e.flops += op.flops;
        e.ops.push_back(op);
    }

    void check_channels(int c, const char* what, int dt = -1)
    {
        const int epc = 16 / dtype_size(dt < 0 ? e.dtype : dt);
        if (c <= 0 || c % epc != 0)
            throw Error(SKY_ERR_INVALID, std::string(what) + ": channel count " + std::to_string(c) +
                                             " is not a multiple of " + std::to_string(epc) + " (16-byte vectors)");
    }

    // upload helpers -------------------------------------------------------------------------------------
    int upload_f32(const std::vector<float>& v)
    {
        if (!emit) return -1;
        float* d = nullptr;
        SKY_HIP(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(float)));
        e.owned.push_back(d);
        SKY_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
        e.weight_bytes += v.size() * sizeof(float);
        e.fweights.push_back(d);
        return (int)e.fweights.size() - 1;
    }

    // Pack conv weights [Cout][taps*cin_store] (+BN fold) for one or two stacked convolutions.
    struct ConvSrc {
        std::string wname;   // conv weight
        std::string bn;      // bn prefix ("" = none)
        std::string bias;    // bias name ("" = none)
    };
    // act: the activation the kernels apply to this convolution.  bf16 + SiLU: the rows and the bias are packed in the EXP2 DOMAIN,
    // i.e. times log2(e) -- the accumulator then holds v' = v log2 e and SiLU is v' * rcp(fma(exp2(-v'), log2 e, log2 e)) = v / (1 + e^-v)
    // (conv_frag.h: S1<__bf16>::silu), one VALU multiplication per output value less.  The scaling happens in double before the
    // one bf16 rounding of a weight: same relative rounding error as the unscaled weight's.
    int pack_conv(const std::vector<ConvSrc>& srcs, int cin_real, int cin_store, int ks, int cdt = -1, int act = ACT_NONE)
    {
        if (!emit) return -1;
        if (cdt < 0) cdt = e.dtype;
#ifdef SKY_PRE_C
        const double pre = (cdt == SKY_BF16 && act == ACT_SILU) ? (double)(SKY_PRE_C) : 1.0;      // experiment, see conv_frag.h
#else
        const double pre = (cdt == SKY_BF16 && act == ACT_SILU) ? 1.4426950408889634 : 1.0;
#endif
        int cout = 0;
        for (auto& s : srcs) cout += (int)W(s.wname).shape[0];
        const int taps = ks * ks;
        const int kstep = conv_k_step(cdt);
        const int K = taps * cin_store;
        const int Kpad = (K + kstep - 1) / kstep * kstep;
        const size_t rows = std::max(conv_weight_rows(cout), (size_t)(cout + 63) / 64 * 64);   // zero rows up to either kernel's N tile
        std::vector<float> packed(rows * Kpad, 0.0f), bias(rows, 0.0f);
        int row0 = 0;
        for (auto& s : srcs) {
            const HostTensor& w = W(s.wname);
            const int co_n = (int)w.shape[0];
            for (int co = 0; co < co_n; ++co) {
                float scale = 1.0f, shift = 0.0f;
                if (!s.bn.empty()) {
                    // eval BatchNorm (eps 1e-5): y = (x - mean) / sqrt(var + eps) * gamma + beta
                    const float g = W(s.bn + "weight").data[co], b = W(s.bn + "bias").data[co];
                    const float mu = W(s.bn + "running_mean").data[co], var = W(s.bn + "running_var").data[co];
                    scale = g / std::sqrt(var + 1e-5f);
                    shift = b - mu * scale;
                }
                if (!s.bias.empty()) shift += W(s.bias).data[co] * scale;
                bias[row0 + co] = (float)((double)shift * pre);
                float* dst = packed.data() + (size_t)(row0 + co) * Kpad;
                for (int ci = 0; ci < cin_real; ++ci)
                    for (int t = 0; t < taps; ++t)
                        dst[t * cin_store + ci] = pre == 1.0 ? w.data[((size_t)co * cin_real + ci) * taps + t] * scale
                                                             : (float)((double)(w.data[((size_t)co * cin_real + ci) * taps + t] * scale) * pre);
            }
            row0 += co_n;
        }
        DevConv d;
        d.Kpad = Kpad;
        d.cdt = cdt;
        d.pre = (float)pre;
        d.bytes = packed.size() * dtype_size(cdt);
        d.rows = (int)rows; d.cout = cout; d.ks = ks; d.cin = cin_store;
        d.name = srcs.empty() ? std::string() : srcs[0].wname;
        SKY_HIP(hipMalloc(&d.w, d.bytes));
        e.owned.push_back(d.w);
        if (cdt == SKY_F32) {
            SKY_HIP(hipMemcpy(d.w, packed.data(), d.bytes, hipMemcpyHostToDevice));
        } else if (cdt == SKY_BF16) {
            std::vector<unsigned short> h(packed.size());
            for (size_t i = 0; i < packed.size(); ++i) h[i] = f32_to_bf16(packed[i]);
            SKY_HIP(hipMemcpy(d.w, h.data(), d.bytes, hipMemcpyHostToDevice));
        } else {
            // fp8 (OCP e4m3fn): one scale per output channel, max |w| of the BN-folded row -> 448; bias stays fp32
            std::vector<unsigned char> h(packed.size());
            d.w_scale.assign(rows, 1.0f);
            for (size_t r = 0; r < rows; ++r) {
                const float* src = packed.data() + r * Kpad;
                float amax = 0.0f;
                for (int k = 0; k < Kpad; ++k) amax = std::max(amax, std::fabs(src[k]));
                const float sc = amax > 0.0f ? amax / 448.0f : 1.0f;
                d.w_scale[r] = sc;
                for (int k = 0; k < Kpad; ++k) h[r * Kpad + k] = f32_to_e4m3(src[k] / sc);
            }
            SKY_HIP(hipMemcpy(d.w, h.data(), d.bytes, hipMemcpyHostToDevice));
            SKY_HIP(hipMalloc(&d.mult, rows * sizeof(float)));
            e.owned.push_back(d.mult);
            SKY_HIP(hipMemcpy(d.mult, d.w_scale.data(), rows * sizeof(float), hipMemcpyHostToDevice));
        }
        SKY_HIP(hipMalloc(&d.bias, rows * sizeof(float)));
        e.owned.push_back(d.bias);
        SKY_HIP(hipMemcpy(d.bias, bias.data(), rows * sizeof(float), hipMemcpyHostToDevice));
        e.weight_bytes += d.bytes + rows * sizeof(float);
        e.convs.push_back(d);
        return (int)e.convs.size() - 1;
    }
};

static int out_dim(int n, int k, int s) { return (n + 2 * (k / 2) - k) / s + 1; }

static void need_bn(Ctx& c, const std::string& p, int ch)
{
    c.need(p + "weight", {ch});
    c.need(p + "bias", {ch});
    c.need(p + "running_mean", {ch});
    c.need(p + "running_var", {ch});
}

struct ConvOpt {
    const TV* out_into = nullptr;
    const TV* res = nullptr;
    bool up2 = false;
    int cin_real = -1;   // when the stored tensor has zero-padded channels (stem)
    int out_dt = -1;     // element type of a freshly allocated output (default: the engine's activation type)
};

// ConvolutionBlock (blocks.py:10-41)
static TV conv_block(Ctx& c, const std::string& p, const TV& x, int cin, int cout, int k, int s, bool act, ConvOpt o = ConvOpt())
{
    const int cin_real = o.cin_real > 0 ? o.cin_real : cin;
    if (k != 1 && k != 3) throw Error(SKY_ERR_INVALID, p + ": kernel_size " + std::to_string(k) + " not supported (1 or 3)");
    c.need(p + "conv.weight", {cout, cin_real, k, k});
    need_bn(c, p + "bn.", cout);
    c.check_channels(cin, (p + "in_channels").c_str(), x.dt);
    c.check_channels(cout, (p + "out_channels").c_str());
    if (x.C != cin) throw Error(SKY_ERR_SHAPE, p + ": input has " + std::to_string(x.C) + " channels, expected " + std::to_string(cin));
    const int Ho = out_dim(x.H, k, s), Wo = out_dim(x.W, k, s);
    TV y;
    if (o.out_into) {
        y = *o.out_into;
        const int eh = o.up2 ? 2 * Ho : Ho, ew = o.up2 ? 2 * Wo : Wo;
        if (y.C != cout || y.H != eh || y.W != ew || y.B != x.B) throw Error(SKY_ERR_SHAPE, p + ": output slot geometry mismatch");
    } else {
        y = c.new_tensor(x.B, o.up2 ? 2 * Ho : Ho, o.up2 ? 2 * Wo : Wo, cout, o.out_dt);
    }
    Op op;
    op.kind = OP_CONV;
    op.in = x; op.out = y;
    if (o.res) op.res = *o.res;
    op.cin = cin; op.cout = cout; op.ks = k; op.stride = s; op.act = act ? ACT_SILU : ACT_NONE; op.up2 = o.up2 ? 1 : 0;
    op.Ho = Ho; op.Wo = Wo;
    op.cdt = x.dt;
    op.wid = c.pack_conv({{p + "conv.weight", p + "bn.", ""}}, cin_real, cin, k, op.cdt, op.act);
    op.flops = 2.0 * x.B * Ho * Wo * (double)cout * k * k * cin_real;
    c.push(op);
    return y;
}

// would the halo-tile kernel run cv2(cv1(x)) of this bottleneck as ONE launch (conv_accepts_cv1)?
static bool bottleneck_fusable(Ctx& c, const TV& x, int cin, int hidden, int cout)
{
    // the calibration twin of an fp8 plan answers for THAT plan (its tensors are all bf16; the plan's CSP tensors are fp8): the buffer lists must agree
    const bool mirror = c.e.mirror_dtype >= 0;
    const int dt = mirror ? c.e.mirror_dtype : (int)x.dt;
    const unsigned opts = mirror ? c.e.mirror_opts : c.e.opts;
    if (!c.emit || (dt != SKY_BF16 && dt != SKY_FP8) || cin != hidden || hidden != cout) return false;
    if (!mirror && c.e.dtype == SKY_FP8 && x.dt != SKY_FP8) return false;       // (an fp8 plan fuses its fp8 tensors only)
    const int esz = dtype_size(dt);
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.B = x.B; a.H = x.H; a.W = x.W; a.Cin = hidden; a.ldi = x.ld; a.Ho = x.H; a.Wo = x.W; a.Cout = cout; a.ldo = x.ld;
    a.ks = 3; a.stride = 1; a.pad = 1; a.M = x.B * x.H * x.W; a.act = ACT_SILU;
    a.Kpad = (9 * hidden + conv_k_step(dt) - 1) / conv_k_step(dt) * conv_k_step(dt);
    a.c1_Kpad = (cin + conv_k_step(dt) - 1) / conv_k_step(dt) * conv_k_step(dt);
    a.opts = opts; a.device = c.e.cfg.device; a.n_cu = c.e.n_cu; a.out_dt = dt;
    const double ext = ((double)a.M - 1.0) * x.ld * esz + hidden * (double)esz;
    a.in_bytes = a.out_bytes = ext < 2147483000.0 ? (unsigned)ext : 0u;
    return conv_accepts_cv1(dt, a);
}

// BottleneckBlock (blocks.py:69-90): x + cv2(cv1(x)) iff shortcut and cin == cout.  fused: one launch (cv1 on the halo tile of
// the 3x3); the output must then be another buffer than x (other tiles still read x's halo while this one writes).
static TV bottleneck(Ctx& c, const std::string& p, const TV& x, int cin, int cout, bool shortcut, float expansion,
                     const TV* out_into = nullptr, bool fused = false)
{
    const int hidden = (int)(cout * expansion);
    if (fused) {
        c.need(p + "cv1.conv.weight", {hidden, cin, 1, 1});
        need_bn(c, p + "cv1.bn.", hidden);
        c.need(p + "cv2.conv.weight", {cout, hidden, 3, 3});
        need_bn(c, p + "cv2.bn.", cout);
        TV y = out_into ? *out_into : c.new_tensor(x.B, x.H, x.W, cout);
        if (y.buf == x.buf) throw Error(SKY_ERR_INVALID, p + ": a fused bottleneck cannot run in place");
        if (c.e.mirror_dtype >= 0 || (c.e.opts & OPT_BNECK_PAIR)) {
            // calibration twin of an fp8 plan (or SKY_BNECK128=pair): the same buffers in the same order (y, then the hidden tensor), written by the two-launch form
            TV u = c.new_tensor(x.B, x.H, x.W, hidden);
            ConvOpt o1;
            o1.out_into = &u;
            conv_block(c, p + "cv1.", x, cin, hidden, 1, 1, true, o1);
            ConvOpt o2;
            o2.out_into = &y;
            if (shortcut && cin == cout) o2.res = &x;
            return conv_block(c, p + "cv2.", u, hidden, cout, 3, 1, true, o2);
        }
        Op op;
        if (x.dt == SKY_FP8) op.hid = c.new_tensor(1, 1, 1, hidden);       // its scale is the hidden tensor's (sky_calibrate); never allocated, never written
        op.kind = OP_CONV;
        op.in = x; op.out = y;
        op.cin = hidden; op.cout = cout; op.ks = 3; op.stride = 1; op.act = ACT_SILU;
        op.Ho = x.H; op.Wo = x.W;
        op.cdt = x.dt;
        op.wid1 = c.pack_conv({{p + "cv1.conv.weight", p + "cv1.bn.", ""}}, cin, cin, 1, op.cdt, ACT_SILU);
        op.wid = c.pack_conv({{p + "cv2.conv.weight", p + "cv2.bn.", ""}}, hidden, hidden, 3, op.cdt, op.act);
        op.c1_res = shortcut && cin == cout;
        op.flops = 2.0 * x.B * x.H * x.W * ((double)hidden * cin + (double)cout * 9 * hidden);
        c.push(op);
        return y;
    }
    TV u = conv_block(c, p + "cv1.", x, cin, hidden, 1, 1, true);
    ConvOpt o;
    o.out_into = out_into;
    if (shortcut && cin == cout) o.res = &x;
    return conv_block(c, p + "cv2.", u, hidden, cout, 3, 1, true, o);
}

// CSPBlock (blocks.py:93-123): cv3(cat(bottlenecks(cv1(x)), cv2(x)))
// up_src: the first up_src->C channels of x are cat's "2x-upsampled lateral map" half (detector.py:215,219): the fused cv1 | cv2 GEMM -- the
// only reader of x -- takes them from the small map itself (ConvArgs::in2), that half of x is never written
static TV csp(Ctx& c, const std::string& p, const TV& x, int cin, int cout, int n, bool shortcut, float expansion,
              const TV* out_into = nullptr, int out_dt = -1, const TV* up_src = nullptr)
{
    const int h = (int)(cout * expansion);
    c.need(p + "cv1.conv.weight", {h, cin, 1, 1});
    need_bn(c, p + "cv1.bn.", h);
    c.need(p + "cv2.conv.weight", {h, cin, 1, 1});
    need_bn(c, p + "cv2.bn.", h);
    c.check_channels(cin, (p + "in_channels").c_str());
    c.check_channels(h, (p + "hidden_channels").c_str());
    if (x.C != cin) throw Error(SKY_ERR_SHAPE, p + ": input channel mismatch");
    TV cat = c.new_tensor(x.B, x.H, x.W, 2 * h);
    {   // cv1 | cv2 as one GEMM with N = 2h, written straight into the concat buffer
        Op op;
        op.kind = OP_CONV;
        op.in = x; op.out = cat;
        if (up_src) {
            op.in = Ctx::slice(x, up_src->C, x.C - up_src->C);
            op.in2 = *up_src; op.in2_cin = up_src->C; op.in2_up2 = 1;
        }
        op.cin = cin; op.cout = 2 * h; op.ks = 1; op.stride = 1; op.act = ACT_SILU;
        op.Ho = x.H; op.Wo = x.W;
        op.cdt = x.dt;
        op.wid = c.pack_conv({{p + "cv1.conv.weight", p + "cv1.bn.", ""}, {p + "cv2.conv.weight", p + "cv2.bn.", ""}}, cin, cin, 1, op.cdt, op.act);
        op.flops = 2.0 * x.B * x.H * x.W * (double)(2 * h) * cin;
        c.push(op);
    }
    TV y1 = Ctx::slice(cat, 0, h);
    // BottleneckBlock(hidden, hidden, shortcut, 1.0), blocks.py:114-117: in place on y1, or -- where the fused cv1 + 3x3 kernel
    // covers the shape and there are at least two blocks -- y1 -> A -> B -> ... -> y1 through two scratch tensors
    const bool fuse = n >= 2 && bottleneck_fusable(c, y1, h, h, h);
    TV scratch[2];
    if (fuse)
        for (int k = 0; k < (n >= 3 ? 2 : 1); ++k) scratch[k] = c.new_tensor(x.B, x.H, x.W, h);
    TV cur = y1;
    for (int j = 0; j < n; ++j) {
        const std::string bp = p + "bottlenecks." + std::to_string(j) + ".";
        if (fuse) {
            const TV dst = j == n - 1 ? y1 : scratch[j & 1];
            cur = bottleneck(c, bp, cur, h, h, shortcut, 1.0f, &dst, true);
        } else {
            bottleneck(c, bp, y1, h, h, shortcut, 1.0f, &y1);
        }
    }
    ConvOpt o;
    o.out_into = out_into;
    o.out_dt = out_dt;
    TV out = conv_block(c, p + "cv3.", cat, 2 * h, cout, 1, 1, true, o);
    // One bottleneck, 64 -> 64 channels with 32 hidden (the first stage of skyeye_s): the four ops above stay in the plan (they are
    // what runs when the stage kernel does not cover the call) and are marked so that run() can replace them by ONE launch of
    // k_csp_stage.hip.  That kernel writes cv3's output while other tiles still read x: the output buffer is alive from the first op.
    if (c.emit && n == 1 && !fuse && cin == 64 && cout == 64 && h == 32 && x.dt == SKY_BF16 && out.dt == SKY_BF16 && out.buf >= 0 &&
        out.buf != x.buf && !(c.e.opts & OPT_NO_CSP_STAGE) && c.e.ops.size() >= 4) {
        const size_t i0 = c.e.ops.size() - 4;
        bool plain = c.e.ops[i0].kind == OP_CONV && c.e.ops[i0].out.buf == cat.buf;
        for (size_t k = i0; k < i0 + 4; ++k) plain = plain && c.e.ops[k].kind == OP_CONV && !c.e.ops[k].fuse_next && !c.e.ops[k].fused_prev;
        if (plain) {
            c.e.ops[i0].csp_stage = 1;
            c.e.ops[i0].csp_shortcut = shortcut ? 1 : 0;
            for (size_t k = i0 + 1; k < i0 + 4; ++k) c.e.ops[k].csp_member = 1;
            Buffer& b = c.e.bufs[out.buf];
            b.first = std::min(b.first, (int)i0);
        }
    }
    return out;
}

// SPPBlock (blocks.py:126-149), kernel sizes (5, 9, 13) = 5, 5o5, 5o5o5
static TV spp(Ctx& c, const std::string& p, const TV& x, int cin, int cout, const TV* out_into = nullptr)
{
    const int h = cin / 2;
    TV cat = c.new_tensor(x.B, x.H, x.W, 4 * h);
    TV s0 = Ctx::slice(cat, 0, h);
    ConvOpt o1;
    o1.out_into = &s0;
    conv_block(c, p + "cv1.", x, cin, h, 1, 1, true, o1);
    // small maps (the 40 x 40 of a 1280-pixel frame): the three cascaded pools in one launch that reads cv1's slice once
    const bool no_pyramid = (c.e.opts & OPT_NO_SPP_PYRAMID) != 0 || c.e.dtype == SKY_FP8;
    const int vec = c.e.epc();
    if (!no_pyramid && h % (2 * vec) == 0 && (long)x.H * x.W * 2 <= 4096) {
        Op op;
        op.kind = OP_MAXPOOL5;
        op.win = 3;
        op.in = Ctx::slice(cat, 0, h);
        op.out = Ctx::slice(cat, h, 3 * h);
        c.push(op);
    } else {
        for (int i = 0; i < 3; ++i) {       // input and output slices share the concat buffer, hence its fp8 scale
            Op op;
            op.kind = OP_MAXPOOL5;
            op.in = Ctx::slice(cat, i * h, h);
            op.out = Ctx::slice(cat, (i + 1) * h, h);
            c.push(op);
        }
    }
    ConvOpt o2;
    o2.out_into = out_into;
    return conv_block(c, p + "cv2.", cat, 4 * h, cout, 1, 1, true, o2);
}

// ChannelAttention / SpatialAttention / CombinedAttention (attention.py:11-130)
static TV cbam(Ctx& c, const std::string& p, const TV& x, int C, bool channel, bool spatial, int reduction, const TV* out_into = nullptr)
{
    const std::string pc = (channel && spatial) ? p + "channel_attention." : p;
    const std::string ps = (channel && spatial) ? p + "spatial_attention." : p;
    const int R = std::max(C / reduction, 1);
    c.check_channels(C, (p + "channels").c_str());
    if (x.C != C) throw Error(SKY_ERR_SHAPE, p + ": input channel mismatch");
    TV y = out_into ? *out_into : c.new_tensor(x.B, x.H, x.W, C);
    const int HW = x.H * x.W;
    int att_buf = -1, gate_buf = -1;
    if (channel) {
        c.need(pc + "shared_mlp.0.weight", {R, C});
        c.need(pc + "shared_mlp.2.weight", {C, R});
        if (C > 2048 || R > 128) throw Error(SKY_ERR_INVALID, p + ": channel attention supports C <= 2048, C/r <= 128");
        const int nchunk = std::max(1, std::min(64, HW / 256));
        const int part = c.emit ? c.new_buf((size_t)x.B * nchunk * 2 * C * 4) : -1;
        att_buf = c.emit ? c.new_buf((size_t)x.B * C * 4) : -1;
        Op r;
        r.kind = OP_CA_REDUCE; r.in = x; r.s0 = part; r.nchunk = nchunk;
        c.push(r);
        Op m;
        m.kind = OP_CA_MLP; m.in = x; m.s0 = part; m.s1 = att_buf; m.nchunk = nchunk; m.R = R;
        if (c.emit) { m.f0 = c.upload_f32(c.W(pc + "shared_mlp.0.weight").data); m.f1 = c.upload_f32(c.W(pc + "shared_mlp.2.weight").data); }
        m.flops = 2.0 * x.B * 2 * 2 * (double)C * R;
        c.push(m);
    }
    if (spatial) {
        c.need(ps + "conv.weight", {1, 2, 7, 7});
        const int stats = c.emit ? c.new_buf((size_t)x.B * HW * 2 * 4) : -1;
        gate_buf = c.emit ? c.new_buf((size_t)x.B * HW * 4) : -1;
        Op s;
        s.kind = OP_SA_STATS; s.in = x; s.s0 = att_buf; s.s1 = stats;
        c.push(s);
        Op g;
        g.kind = OP_SA_GATE; g.in = x; g.s0 = stats; g.s1 = gate_buf;
        if (c.emit) g.f0 = c.upload_f32(c.W(ps + "conv.weight").data);
        g.flops = 2.0 * x.B * HW * 98.0;
        c.push(g);
    }
    Op a;
    a.kind = OP_SCALE; a.in = x; a.out = y; a.s0 = att_buf; a.s1 = gate_buf;
    c.push(a);
    return y;
}

// nn.Linear / 1x1 nn.Conv2d with bias on NHWC tokens (a token is a pixel): same GEMM as a 1x1 convolution
static TV linear(Ctx& c, const std::string& wname, const std::string& bname, const TV& x, int cin, int cout, int act, bool conv4d,
                 const TV* out_into = nullptr, const TV* res = nullptr)
{
    if (conv4d) c.need(wname, {cout, cin, 1, 1}); else c.need(wname, {cout, cin});
    if (!bname.empty()) c.need(bname, {cout});
    c.check_channels(cin, (wname + " in_features").c_str());
    c.check_channels(cout, (wname + " out_features").c_str());
    if (x.C != cin) throw Error(SKY_ERR_SHAPE, wname + ": input has " + std::to_string(x.C) + " channels, expected " + std::to_string(cin));
    TV y = out_into ? *out_into : c.new_tensor(x.B, x.H, x.W, cout);
    Op op;
    op.kind = OP_CONV;
    op.in = x; op.out = y;
    if (res) op.res = *res;
    op.cin = cin; op.cout = cout; op.ks = 1; op.stride = 1; op.act = act;
    op.Ho = x.H; op.Wo = x.W;
    op.cdt = x.dt;
    op.wid = c.pack_conv({{wname, "", bname}}, cin, cin, 1, op.cdt, op.act);
    op.flops = 2.0 * x.B * x.H * x.W * (double)cout * cin;
    c.push(op);
    return y;
}

static void no_fp8(Ctx& c, const char* what)
{
    if (c.e.dtype == SKY_FP8)
        throw Error(SKY_ERR_INVALID, std::string(what) + " is not built for the fp8 engine (SKY_FP8 covers the convolutional detector; use SKY_BF16)");
}

static TV layernorm(Ctx& c, const std::string& p, const TV& x, int C)
{
    no_fp8(c, "LayerNorm / TransformerLayer");
    c.need(p + "weight", {C});
    c.need(p + "bias", {C});
    TV y = c.new_tensor(x.B, x.H, x.W, C);
    Op op;
    op.kind = OP_LAYERNORM;
    op.in = x; op.out = y;
    if (c.emit) { op.f0 = c.upload_f32(c.W(p + "weight").data); op.f1 = c.upload_f32(c.W(p + "bias").data); }
    c.push(op);
    return y;
}

static TV attention_core(Ctx& c, const TV& qkv, int C, int heads, float scale, int bias_f, int mask_ext, int nW, int win = 0)
{
    no_fp8(c, "attention");
    const int d = C / heads;
    if (C % heads || (d != 8 && d != 16 && d != 32 && d != 64 && d != 128))
        throw Error(SKY_ERR_INVALID, "attention head dimension " + std::to_string(d) + " not supported (8, 16, 32, 64, 128)");
    TV y = c.new_tensor(qkv.B, qkv.H, qkv.W, C);
    Op op;
    op.kind = OP_ATTENTION;
    op.in = qkv; op.out = y;
    op.heads = heads; op.ntok = win ? win * win : qkv.H * qkv.W; op.scale = scale; op.f0 = bias_f; op.mask_ext = mask_ext; op.nW = nW;
    op.win = win;
    op.flops = 4.0 * qkv.B * (double)qkv.H * qkv.W * op.ntok * C;
    c.push(op);
    return y;
}

// TransformerLayer.forward, eval mode (attention.py:282-309): x + MHA(LN1(x)); then + FFN(LN2(.)), FFN = Linear-ReLU-Linear
static TV transformer(Ctx& c, const std::string& p, const TV& x, int C, int heads, int ff)
{
    TV n1 = layernorm(c, p + "norm1.", x, C);
    TV qkv = linear(c, p + "self_attn.in_proj_weight", p + "self_attn.in_proj_bias", n1, C, 3 * C, ACT_NONE, false);
    TV att = attention_core(c, qkv, C, heads, 1.0f / std::sqrt((float)(C / heads)), -1, -1, 1);
    TV x1 = linear(c, p + "self_attn.out_proj.weight", p + "self_attn.out_proj.bias", att, C, C, ACT_NONE, false, nullptr, &x);
    TV n2 = layernorm(c, p + "norm2.", x1, C);
    TV f = linear(c, p + "feedforward.0.weight", p + "feedforward.0.bias", n2, C, ff, ACT_RELU, false);
    return linear(c, p + "feedforward.3.weight", p + "feedforward.3.bias", f, ff, C, ACT_NONE, false, nullptr, &x1);
}

// WindowedSelfAttention.forward (attention.py:358-399); x is [B_, N = ws*ws, C] (carried as B_ x 1 x N x C)
static TV windowed_attention(Ctx& c, const std::string& p, const TV& x, int C, int ws, int heads, int mask_ext, int nW)
{
    const int N = ws * ws, T = (2 * ws - 1) * (2 * ws - 1);
    c.need(p + "relative_position_bias_table", {T, heads});
    if (x.H * x.W != N) throw Error(SKY_ERR_SHAPE, p + ": tokens per window must equal window_size^2");
    TV qkv = linear(c, p + "qkv.weight", p + "qkv.bias", x, C, 3 * C, ACT_NONE, false);
    int bias_f = -1;
    if (c.emit) {   // bias[h][i][j] = table[index[i][j]][h], index as attention.py:342-353 (meshgrid 'ij')
        const std::vector<float>& tab = c.W(p + "relative_position_bias_table").data;
        std::vector<float> bias((size_t)heads * N * N);
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                const int dy = i / ws - j / ws + ws - 1, dx = i % ws - j % ws + ws - 1;
                const int idx = dy * (2 * ws - 1) + dx;
                for (int h = 0; h < heads; ++h) bias[((size_t)h * N + i) * N + j] = tab[(size_t)idx * heads + h];
            }
        bias_f = c.upload_f32(bias);
    }
    const int d = C / heads;
    TV att = attention_core(c, qkv, C, heads, 1.0f / std::sqrt((float)d), bias_f, mask_ext, nW);
    return linear(c, p + "proj.weight", p + "proj.bias", att, C, C, ACT_NONE, false);
}

// WindowedSelfAttention applied to a whole [B, H, W, C] map (the reference never calls the module; this
// is the build-defined wiring ahead of the detection convolutions): qkv and proj are per-pixel Linear layers, so only the
// attention core needs the windows -- it addresses them in place, no window_partition / window_reverse copies.
static TV windowed_attention_map(Ctx& c, const std::string& p, const TV& x, int C, int ws, int heads)
{
    const int N = ws * ws, T = (2 * ws - 1) * (2 * ws - 1);
    if (c.emit && (x.H % ws || x.W % ws))     // (the parameter-spec dry run uses a 64 x 64 frame: geometry is checked at plan time)
        throw Error(SKY_ERR_SHAPE, p + ": feature map " + std::to_string(x.H) + "x" + std::to_string(x.W) + " is not a multiple of the window size " + std::to_string(ws));
    c.need(p + "relative_position_bias_table", {T, heads});
    TV qkv = linear(c, p + "qkv.weight", p + "qkv.bias", x, C, 3 * C, ACT_NONE, false);
    int bias_f = -1;
    if (c.emit) {
        const std::vector<float>& tab = c.W(p + "relative_position_bias_table").data;
        std::vector<float> bias((size_t)heads * N * N);
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                const int dy = i / ws - j / ws + ws - 1, dx = i % ws - j % ws + ws - 1;
                const int idx = dy * (2 * ws - 1) + dx;
                for (int h = 0; h < heads; ++h) bias[((size_t)h * N + i) * N + j] = tab[(size_t)idx * heads + h];
            }
        bias_f = c.upload_f32(bias);
    }
    TV att = attention_core(c, qkv, C, heads, 1.0f / std::sqrt((float)(C / heads)), bias_f, -1, 1, ws);
    return linear(c, p + "proj.weight", p + "proj.bias", att, C, C, ACT_NONE, false);
}

// CrossLayerAttention.forward (attention.py:174-241), closed form; D4: key/value projections map Ck -> Cq
static TV cross_layer_attention(Ctx& c, const std::string& p, const TV& q, const TV& k, int Cq, int Ck, int Cout, int heads, int region,
                                const TV* res = nullptr, const TV* out_into = nullptr)
{
    no_fp8(c, "CrossLayerAttention");
    if (Cq % heads) throw Error(SKY_ERR_INVALID, p + ": query_channels must be divisible by heads");
    TV Q = linear(c, p + "query_projection.weight", p + "query_projection.bias", q, Cq, Cq, ACT_NONE, true);
    // key | value projections as one GEMM with N = 2*Cq
    c.need(p + "key_projection.weight", {Cq, Ck, 1, 1});
    c.need(p + "key_projection.bias", {Cq});
    c.need(p + "value_projection.weight", {Cq, Ck, 1, 1});
    c.need(p + "value_projection.bias", {Cq});
    c.check_channels(Ck, (p + "key_channels").c_str());
    if (k.C != Ck) throw Error(SKY_ERR_SHAPE, p + ": key tensor channel mismatch");
    TV KV = c.new_tensor(k.B, k.H, k.W, 2 * Cq);
    {
        Op op;
        op.kind = OP_CONV;
        op.in = k; op.out = KV;
        op.cin = Ck; op.cout = 2 * Cq; op.ks = 1; op.stride = 1; op.act = ACT_NONE;
        op.Ho = k.H; op.Wo = k.W;
        op.cdt = k.dt;
        op.wid = c.pack_conv({{p + "key_projection.weight", "", p + "key_projection.bias"}, {p + "value_projection.weight", "", p + "value_projection.bias"}}, Ck, Ck, 1, op.cdt);
        op.flops = 2.0 * k.B * k.H * k.W * (double)(2 * Cq) * Ck;
        c.push(op);
    }
    TV A = c.new_tensor(q.B, q.H, q.W, Cq);
    Op op;
    op.kind = OP_CLA;
    op.in = Q; op.in2 = KV; op.out = A;
    op.heads = heads; op.v_off = Cq;
    op.scale = 1.0f / std::sqrt((float)Cq);                       // attention.py:159: 1/sqrt(query_channels), not head dim
    op.r2 = (float)(region * region);
    op.s0 = c.emit ? c.new_buf((size_t)q.B * q.H * q.W * heads * 4) : -1;
    op.flops = 2.0 * q.B * q.H * q.W * (double)Cq * 9;
    c.push(op);
    return linear(c, p + "output_projection.weight", p + "output_projection.bias", A, Cq, Cout, ACT_NONE, true, out_into, res);
}

static int scaled_channels(int x, float wm) { return std::max((int)std::lround(std::nearbyint((double)x * wm)), 1); }
// Python's round() is banker's rounding; nearbyint under the default FE_TONEAREST mode matches it.
static int scaled_depth(int x, float dm) { return std::max((int)std::nearbyint((double)x * dm), 1); }

struct BackboneOut {
    TV p3, p4, p5;
    int c3, c4, c5;
};

// import the caller's image tensor with FocusBlock's space-to-depth folded in (blocks.py:170-182)
static TV import_focus(Ctx& c, int ext, int B, int C, int H, int W)
{
    if ((H & 1) || (W & 1)) throw Error(SKY_ERR_SHAPE, "FocusBlock needs even H and W");
    // the fp8 engine keeps the stem in bf16: pixel values i / 255 have 8 significant bits, e4m3 has 4
    const int dt = c.e.dtype == SKY_FP8 ? (int)SKY_BF16 : c.e.dtype;
    const int epc = std::max(16 / dtype_size(dt), c.e.dtype == SKY_FP8 ? 16 : 1);
    const int Cs = (4 * C + epc - 1) / epc * epc;
    TV t = c.new_tensor(B, H / 2, W / 2, Cs, dt);
    Op op;
    op.kind = OP_IMPORT;
    op.in.ext = ext; op.out = t; op.s2d = 1; op.src_c = C; op.src_h = H; op.src_w = W;
    c.push(op);
    return t;
}

static TV import_plain(Ctx& c, int ext, int B, int C, int H, int W, const TV* into = nullptr, bool tokens = false)
{
    TV t = into ? *into : c.new_tensor(B, H, W, C);
    Op op;
    op.kind = OP_IMPORT;
    op.force_nhwc = tokens ? 1 : 0;     // [B_, N, C] token tensors are channel-last at the boundary
    op.in.ext = ext; op.out = t; op.s2d = 0; op.src_c = C; op.src_h = H; op.src_w = W;
    c.push(op);
    return t;
}

static void export_nchw(Ctx& c, const TV& t, int ext_out)
{
    Op op;
    op.kind = OP_EXPORT;
    op.in = t; op.out.ext = 16 + ext_out;
    c.push(op);
    if (c.emit) {
        IoInfo io;
        io.ndim = 4;
        io.shape[0] = t.B; io.shape[1] = t.C; io.shape[2] = t.H; io.shape[3] = t.W;
        if ((int)c.e.out_info.size() <= ext_out) c.e.out_info.resize(ext_out + 1);
        c.e.out_info[ext_out] = io;
    }
}

static void export_tokens(Ctx& c, const TV& t, int ext_out)
{
    Op op;
    op.kind = OP_EXPORT_TOKENS;
    op.in = t; op.out.ext = 16 + ext_out;
    c.push(op);
    if (c.emit) {
        IoInfo io;
        io.ndim = 3;
        io.shape[0] = t.B; io.shape[1] = (int64_t)t.H * t.W; io.shape[2] = t.C;
        if ((int)c.e.out_info.size() <= ext_out) c.e.out_info.resize(ext_out + 1);
        c.e.out_info[ext_out] = io;
    }
}

// FocusBlock as a standalone module / backbone stem
static TV focus(Ctx& c, const std::string& p, int ext, int B, int C, int H, int W, int cout, int k)
{
    TV s = import_focus(c, ext, B, C, H, W);
    ConvOpt o;
    o.cin_real = 4 * C;
    return conv_block(c, p + "conv.", s, s.C, cout, k, 1, true, o);
}

// Backbone.forward (backbone.py:82-99).  p3/p4/p5 may be directed into concat slots of the neck.
static BackboneOut backbone(Ctx& c, const std::string& p, int ext, int B, int C, int H, int W, int base, float dm, float wm,
                            const TV* p3_into, const TV* p4_into, const TV* p5_into)
{
    const int c1 = scaled_channels(base, wm), c2 = scaled_channels(base * 2, wm), c3 = scaled_channels(base * 4, wm);
    const int c4 = scaled_channels(base * 8, wm), c5 = scaled_channels(base * 16, wm);
    TV s = focus(c, p + "stage1.0.", ext, B, C, H, W, c1, 3);                                   // backbone.py:48
    s = conv_block(c, p + "stage1.1.", s, c1, c2, 3, 2, true);                                  // :50
    if (c.emit && c.e.dtype == SKY_BF16 && C == 3 && c1 == 32 && c2 == 64 && !(c.e.opts & OPT_NO_STEM_DOWN) && c.e.ops.size() >= 3) {
        // import (space-to-depth) -> stem 3x3 -> 3x3 stride 2: with uint8 frames the three ops collapse into k_stem_down.hip at run time
        Op& down = c.e.ops[c.e.ops.size() - 1];
        Op& stem = c.e.ops[c.e.ops.size() - 2];
        const Op& imp = c.e.ops[c.e.ops.size() - 3];
        if (imp.kind == OP_IMPORT && imp.s2d && stem.kind == OP_CONV && down.kind == OP_CONV && stem.in.buf == imp.out.buf &&
            down.in.buf == stem.out.buf && down.in.off == stem.out.off && !stem.res.valid() && !down.res.valid() && !stem.fuse_next &&
            !down.fuse_next && !down.fused_prev) {
            stem.stem_down = 1;
            down.fused_prev = 1;
        }
    }
    s = csp(c, p + "stage1.2.", s, c2, c2, scaled_depth(3, dm), true, 0.5f);                    // :52
    s = conv_block(c, p + "stage2.0.", s, c2, c3, 3, 2, true);                                  // :58
    TV p3 = csp(c, p + "stage2.1.", s, c3, c3, scaled_depth(9, dm), true, 0.5f, p3_into);       // :60
    s = conv_block(c, p + "stage3.0.", p3, c3, c4, 3, 2, true);                                 // :66
    s = csp(c, p + "stage3.1.", s, c4, c4, scaled_depth(9, dm), true, 0.5f);                    // :68
    TV p4 = cbam(c, p + "stage3.2.", s, c4, true, true, 16, p4_into);                           // :70
    s = conv_block(c, p + "stage4.0.", p4, c4, c5, 3, 2, true);                                 // :76
    s = csp(c, p + "stage4.1.", s, c5, c5, scaled_depth(3, dm), true, 0.5f);                    // :78
    TV p5 = spp(c, p + "stage4.2.", s, c5, c5, p5_into);                                        // :79
    return {p3, p4, p5, c3, c4, c5};
}

struct NeckSlots {
    TV cat_p4m, cat_p3m, cat_p4c, cat_p5c;   // concat buffers
    TV p3, p4, p5;                           // where the backbone features must be written
};

static NeckSlots neck_alloc(Ctx& c, int B, int H3, int W3, int H4, int W4, int H5, int W5, int c3, int c4, int c5)
{
    NeckSlots n;
    n.cat_p4m = c.new_tensor(B, H4, W4, 2 * c4);     // cat([up(lateral5(p5)), p4])   detector.py:215
    n.cat_p3m = c.new_tensor(B, H3, W3, 2 * c3);     // cat([up(lateral4(p4)), p3])   detector.py:219
    n.cat_p4c = c.new_tensor(B, H4, W4, c3 + c4);    // cat([down3(p3'), p4'])        detector.py:224
    n.cat_p5c = c.new_tensor(B, H5, W5, c4 + c5);    // cat([down4(p4''), p5])        detector.py:228
    n.p3 = Ctx::slice(n.cat_p3m, c3, c3);
    n.p4 = Ctx::slice(n.cat_p4m, c4, c4);
    n.p5 = Ctx::slice(n.cat_p5c, c4, c5);
    return n;
}

// FeatureNeck.forward (detector.py:197-231); quirks kept: lateral_conv4 reads RAW p4, p5_cat takes RAW p5.
static void neck(Ctx& c, const std::string& p, const NeckSlots& n, int c3, int c4, int c5, TV out[3])
{
    const TV &p3 = n.p3, &p4 = n.p4, &p5 = n.p5;
    // would the CSP's fused cv1 | cv2 GEMM read the upsampled half of its input from the small lateral map itself (conv_accepts_in2)?
    auto reads_small = [&](const TV& cat, int c_lat, int hidden2) {
        const int dt = c.e.mirror_dtype >= 0 ? c.e.mirror_dtype : (int)c.e.dtype;          // (the twin of an fp8 plan asks for THAT plan)
        const unsigned opts = c.e.mirror_dtype >= 0 ? c.e.mirror_opts : c.e.opts;
        if (!c.emit || (opts & OPT_NO_IN2)) return false;
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        const int esz = dtype_size(dt);
        a.B = cat.B; a.H = cat.H; a.W = cat.W; a.Ho = cat.H; a.Wo = cat.W; a.Cin = cat.C; a.Cout = hidden2; a.ldi = cat.ld; a.ldo = hidden2;
        a.ks = 1; a.stride = 1; a.act = ACT_SILU; a.M = cat.B * cat.H * cat.W; a.out_dt = -1;
        a.Kpad = (cat.C + conv_k_step(dt) - 1) / conv_k_step(dt) * conv_k_step(dt);
        a.opts = opts; a.device = c.e.cfg.device; a.n_cu = c.e.n_cu;
        a.in2 = &a; a.in2_cin = c_lat; a.ldi2 = c_lat; a.in2_up2 = 1;
        const double ext = ((double)a.M - 1.0) * cat.ld * esz + (double)cat.C * esz;
        a.in_bytes = a.in2_bytes = a.out_bytes = ext < 2147483000.0 ? (unsigned)ext : 0u;
        return conv_accepts_in2(dt, a);
    };
    // returns the small lateral map when the consumer reads it directly (then the slot's half of the concat buffer stays unwritten)
    auto lateral = [&](const std::string& name, const TV& x, int cin, int cout, const TV& cat, int Ht, int Wt, int hidden2, TV& small) {
        const TV slot = Ctx::slice(cat, 0, cout);
        if (Ht == 2 * x.H && Wt == 2 * x.W && reads_small(cat, cout, hidden2)) {
            // fp8: the small map carries the concat buffer's scale (one multiplier per output channel covers both parts of K)
            small = conv_block(c, p + name, x, cin, cout, 1, 1, true);
            c.tie_scales(small.buf, cat.buf);
            if (c.e.mirror_dtype < 0) return true;
            Op u;                               // calibration twin: the same buffers as the plan it measures for, every tensor written
            u.kind = OP_UPSAMPLE; u.in = small; u.out = slot;
            c.push(u);
            return false;
        }
        if (Ht == 2 * x.H && Wt == 2 * x.W) {   // exact 2x: upsample in the epilogue
            ConvOpt o;
            o.out_into = &slot; o.up2 = true;
            conv_block(c, p + name, x, cin, cout, 1, 1, true, o);
        } else {                                // F.interpolate(size=...) for odd geometries
            TV t = conv_block(c, p + name, x, cin, cout, 1, 1, true);
            Op u;
            u.kind = OP_UPSAMPLE; u.in = t; u.out = slot;
            c.tie_scales(t.buf, slot.buf);
            c.push(u);
        }
        return false;
    };
    TV small5, small4;
    const bool s5 = lateral("lateral_conv5.", p5, c5, c4, n.cat_p4m, p4.H, p4.W, c4, small5);   // :210,214
    const bool s4 = lateral("lateral_conv4.", p4, c4, c3, n.cat_p3m, p3.H, p3.W, c3, small4);   // :211,218
    TV p4p_slot = Ctx::slice(n.cat_p4c, c3, c4);
    csp(c, p + "fpn_conv4.", n.cat_p4m, 2 * c4, c4, 3, true, 0.5f, &p4p_slot, -1, s5 ? &small5 : nullptr);          // :216
    // fp8 engine: the three maps the detection levels read stay bf16 (the last rounding before the box regression is bf16's, and
    // the detection convolutions then run in bf16 like the stem); the stride-2 convolutions that also read them write fp8 again
    const int odt = c.e.dtype == SKY_FP8 ? (int)SKY_BF16 : -1;
    out[0] = csp(c, p + "fpn_conv3.", n.cat_p3m, 2 * c3, c3, 3, true, 0.5f, nullptr, odt, s4 ? &small4 : nullptr);            // :220
    ConvOpt d3;
    TV d3_slot = Ctx::slice(n.cat_p4c, 0, c3);
    d3.out_into = &d3_slot;
    conv_block(c, p + "downsample3.", out[0], c3, c3, 3, 2, true, d3);                  // :223
    out[1] = csp(c, p + "pan_conv4.", n.cat_p4c, c3 + c4, c4, 3, true, 0.5f, nullptr, odt);           // :225
    ConvOpt d4;
    TV d4_slot = Ctx::slice(n.cat_p5c, 0, c4);
    d4.out_into = &d4_slot;
    conv_block(c, p + "downsample4.", out[1], c4, c4, 3, 2, true, d4);                  // :227
    out[2] = csp(c, p + "pan_conv5.", n.cat_p5c, c4 + c5, c5, 3, true, 0.5f, nullptr, odt);           // :229
}

// DetectionHead.forward + process_detections (detector.py:61-145) -> outputs [det, raw_0, ...]
static void head(Ctx& c, const std::string& p, const TV* feats, int nl, int nc, int na, const float* anchors, int in_h, int in_w,
                 int first_out)
{
    const int no = nc + 5;
    long rows = 0;
    for (int i = 0; i < nl; ++i) rows += (long)na * feats[i].H * feats[i].W;
    long off = 0;
    for (int i = 0; i < nl; ++i) {
        const TV& f = feats[i];
        const std::string l = p + "detection_layers." + std::to_string(i) + ".";
        c.need(l + "weight", {na * no, f.C, 1, 1});
        c.need(l + "bias", {na * no});
        c.check_channels(f.C, (l + "in_channels").c_str());
        if (na > 8) throw Error(SKY_ERR_INVALID, "at most 8 anchors per level");
        Op op;
        op.kind = OP_CONV;
        op.in = f;
        op.cin = f.C; op.cout = na * no; op.ks = 1; op.stride = 1; op.act = ACT_NONE;
        op.Ho = f.H; op.Wo = f.W;
        op.head = 1; op.level = i;
        op.raw_ext = 16 + first_out + 1 + i;
        op.det_ext = 16 + first_out;
        op.det_rows = rows; op.det_off = off;
        op.stride_px = (float)std::max((double)in_h / f.H, (double)in_w / f.W);          // detector.py:107-109
        op.cdt = f.dt;
        op.wid = c.pack_conv({{l + "weight", "", l + "bias"}}, f.C, f.C, 1, op.cdt);
        op.flops = 2.0 * f.B * f.H * f.W * (double)(na * no) * f.C;
        c.push(op);
        if (c.emit && !(c.e.opts & OPT_NO_CV3_HEAD) && f.C == 128 && f.dt == SKY_BF16 && f.buf >= 0) {
            // the convolution that writes this level's input (CSP cv3): both can run as one kernel (launch_cv3_head decides at run time)
            const int hi = (int)c.e.ops.size() - 1;
            for (int k = hi - 1; k >= 0; --k) {
                Op& pr = c.e.ops[k];
                if (pr.out.buf != f.buf) continue;
                if (pr.kind == OP_CONV && !pr.head && pr.out.off == f.off && pr.out.ld == f.ld && pr.ks == 1 && pr.stride == 1 && pr.cin == 128 && pr.cout == 128 &&
                    (pr.act == ACT_SILU || pr.act == ACT_NONE) && !pr.up2 && !pr.res.valid() && !pr.in2.valid() && pr.wid1 < 0 && !pr.fuse_next && !pr.fused_prev && !pr.csp_member &&
                    pr.cdt == SKY_BF16 && pr.out.dt == SKY_BF16)
                    pr.head_op = hi;
                break;                                         // (the last writer of the buffer, whatever it is)
            }
        }
        off += (long)na * f.H * f.W;
        if (c.emit) {
            IoInfo io;
            io.ndim = 5;
            io.shape[0] = f.B; io.shape[1] = na; io.shape[2] = f.H; io.shape[3] = f.W; io.shape[4] = no;
            if ((int)c.e.out_info.size() <= first_out + 1 + i) c.e.out_info.resize(first_out + 2 + i);
            c.e.out_info[first_out + 1 + i] = io;
        }
    }
    if (c.emit) {
        IoInfo io;
        io.ndim = 3;
        io.shape[0] = feats[0].B; io.shape[1] = rows; io.shape[2] = no;
        if ((int)c.e.out_info.size() <= first_out) c.e.out_info.resize(first_out + 1);
        c.e.out_info[first_out] = io;
    }
    (void)anchors;
}

static const float kDefaultAnchors[18] = {10, 13, 16, 30, 33, 23, 30, 61, 62, 45, 59, 119, 116, 90, 156, 198, 373, 326};   // detector.py:39-43

// ------------------------------------------------------------------------------------------------ module dispatch
struct Geometry {
    int n = 0;
    int64_t shape[SKY_MAX_IO][5];
    int ndim[SKY_MAX_IO];
};

static void expect_inputs(const Geometry& g, int n, const char* mod)
{
    if (g.n != n) throw Error(SKY_ERR_SHAPE, std::string(mod) + ": expected " + std::to_string(n) + " input(s), got " + std::to_string(g.n));
    for (int i = 0; i < n; ++i)
        if (g.ndim[i] != 4) throw Error(SKY_ERR_SHAPE, std::string(mod) + ": inputs must be 4-D [B,C,H,W]");
}

static void build(Ctx& c, const Geometry& g)
{
    Engine& e = c.e;
    const sky_config& cf = e.cfg;
    auto dim = [&](int i, int d) { return (int)g.shape[i][d]; };
    switch (cf.module) {
        case SKY_MOD_CONV_BLOCK: {
            expect_inputs(g, 1, "ConvolutionBlock");
            TV x = import_plain(c, 0, dim(0, 0), dim(0, 1), dim(0, 2), dim(0, 3));
            export_nchw(c, conv_block(c, "", x, cf.c_in, cf.c_out, cf.kernel_size, cf.stride, cf.activation != 0), 0);
            break;
        }
        case SKY_MOD_BOTTLENECK: {
            expect_inputs(g, 1, "BottleneckBlock");
            TV x = import_plain(c, 0, dim(0, 0), dim(0, 1), dim(0, 2), dim(0, 3));
            export_nchw(c, bottleneck(c, "", x, cf.c_in, cf.c_out, cf.shortcut != 0, cf.expansion), 0);
            break;
        }
        case SKY_MOD_CSP: {
            expect_inputs(g, 1, "CSPBlock");
            TV x = import_plain(c, 0, dim(0, 0), dim(0, 1), dim(0, 2), dim(0, 3));
            export_nchw(c, csp(c, "", x, cf.c_in, cf.c_out, cf.num_blocks, cf.shortcut != 0, cf.expansion), 0);
            break;
        }
        case SKY_MOD_SPP: {
            expect_inputs(g, 1, "SPPBlock");
            TV x = import_plain(c, 0, dim(0, 0), dim(0, 1), dim(0, 2), dim(0, 3));
            export_nchw(c, spp(c, "", x, cf.c_in, cf.c_out), 0);
            break;
        }
        case SKY_MOD_FOCUS: {
            expect_inputs(g, 1, "FocusBlock");
            if (dim(0, 1) != cf.c_in) throw Error(SKY_ERR_SHAPE, "FocusBlock: input channel mismatch");
            export_nchw(c, focus(c, "", 0, dim(0, 0), dim(0, 1), dim(0, 2), dim(0, 3), cf.c_out, cf.kernel_size), 0);
            break;
        }
        case SKY_MOD_CHANNEL_ATTENTION:
        case SKY_MOD_SPATIAL_ATTENTION:
        case SKY_MOD_COMBINED_ATTENTION: {
            expect_inputs(g, 1, "attention");
            TV x = import_plain(c, 0, dim(0, 0), dim(0, 1), dim(0, 2), dim(0, 3));
            const bool ch = cf.module != SKY_MOD_SPATIAL_ATTENTION, sp = cf.module != SKY_MOD_CHANNEL_ATTENTION;
            export_nchw(c, cbam(c, "", x, dim(0, 1), ch, sp, cf.reduction_ratio > 0 ? cf.reduction_ratio : 16), 0);
            break;
        }
        case SKY_MOD_BACKBONE: {
            expect_inputs(g, 1, "Backbone");
            BackboneOut b = backbone(c, "", 0, dim(0, 0), dim(0, 1), dim(0, 2), dim(0, 3), cf.base_channels, cf.depth_multiple,
                                     cf.width_multiple, nullptr, nullptr, nullptr);
            export_nchw(c, b.p3, 0);
            export_nchw(c, b.p4, 1);
            export_nchw(c, b.p5, 2);
            break;
        }
        case SKY_MOD_NECK: {
            expect_inputs(g, 3, "FeatureNeck");
            const int c3 = cf.level_channels[0], c4 = cf.level_channels[1], c5 = cf.level_channels[2];
            NeckSlots n = neck_alloc(c, dim(0, 0), dim(0, 2), dim(0, 3), dim(1, 2), dim(1, 3), dim(2, 2), dim(2, 3), c3, c4, c5);
            import_plain(c, 0, dim(0, 0), dim(0, 1), dim(0, 2), dim(0, 3), &n.p3);
            import_plain(c, 1, dim(1, 0), dim(1, 1), dim(1, 2), dim(1, 3), &n.p4);
            import_plain(c, 2, dim(2, 0), dim(2, 1), dim(2, 2), dim(2, 3), &n.p5);
            TV out[3];
            neck(c, "", n, c3, c4, c5, out);
            for (int i = 0; i < 3; ++i) export_nchw(c, out[i], i);
            break;
        }
        case SKY_MOD_HEAD: {
            const int nl = cf.num_levels;
            expect_inputs(g, nl, "DetectionHead");
            TV f[SKY_MAX_LEVELS];
            for (int i = 0; i < nl; ++i) f[i] = import_plain(c, i, dim(i, 0), dim(i, 1), dim(i, 2), dim(i, 3));
            head(c, "", f, nl, cf.nc, cf.num_anchors, cf.anchors, cf.input_h, cf.input_w, 0);
            break;
        }
        case SKY_MOD_DETECTOR:
        case SKY_MOD_ENHANCED_DETECTOR: {
            expect_inputs(g, 1, "SkyEyeDetector");
            const int B = dim(0, 0), C = dim(0, 1), H = dim(0, 2), W = dim(0, 3);
            if (H % 32 || W % 32) throw Error(SKY_ERR_SHAPE, "SkyEyeDetector: H and W must be multiples of the maximum stride 32 (check_img_size, general.py:248-268)");
            const float wm = cf.width_multiple;
            const int c3 = scaled_channels(cf.base_channels * 4, wm), c4 = scaled_channels(cf.base_channels * 8, wm),
                      c5 = scaled_channels(cf.base_channels * 16, wm);
            NeckSlots n = neck_alloc(c, B, H / 8, W / 8, H / 16, W / 16, H / 32, W / 32, c3, c4, c5);
            backbone(c, "backbone.backbone.", 0, B, C, H, W, cf.base_channels, cf.depth_multiple, wm, &n.p3, &n.p4, &n.p5);
            TV out[3];
            neck(c, "neck.", n, c3, c4, c5, out);                                        // D1: neck width scale = 1
            if (cf.module == SKY_MOD_ENHANCED_DETECTOR) {                                // detector.py:485-491 with D4
                TV p4e = cross_layer_attention(c, "cross_attention_p5_p4.", out[1], out[2], c4, c5, c4, 4, 2, &out[1]);
                TV p3e = cross_layer_attention(c, "cross_attention_p4_p3.", out[0], p4e, c3, c4, c3, 4, 2, &out[0]);
                out[0] = p3e;
                out[1] = p4e;
            }
            if (cf.reserved[0]) {   // D5 (build-defined "transformer prediction heads"): windows of 8 on P3 / P4, encoder layer on P5
                out[0] = windowed_attention_map(c, "head_attention.p3.", out[0], c3, 8, c3 / 32);
                out[1] = windowed_attention_map(c, "head_attention.p4.", out[1], c4, 8, c4 / 32);
                out[2] = transformer(c, "head_attention.p5.", out[2], c5, 8, 4 * c5);
            }
            head(c, "detection_head.", out, 3, cf.nc, cf.num_anchors, cf.anchors, H, W, 0);
            break;
        }
        case SKY_MOD_CROSS_LAYER_ATTENTION: {
            expect_inputs(g, 2, "CrossLayerAttention");
            TV q = import_plain(c, 0, dim(0, 0), dim(0, 1), dim(0, 2), dim(0, 3));
            TV k = import_plain(c, 1, dim(1, 0), dim(1, 1), dim(1, 2), dim(1, 3));
            export_nchw(c, cross_layer_attention(c, "", q, k, cf.c_in, cf.key_channels > 0 ? cf.key_channels : cf.c_in,
                                                 cf.c_out > 0 ? cf.c_out : cf.c_in, cf.heads > 0 ? cf.heads : 4,
                                                 cf.region_size > 0 ? cf.region_size : 2), 0);
            break;
        }
        case SKY_MOD_TRANSFORMER_LAYER: {
            expect_inputs(g, 1, "TransformerLayer");
            TV x = import_plain(c, 0, dim(0, 0), dim(0, 1), dim(0, 2), dim(0, 3));
            export_nchw(c, transformer(c, "", x, cf.c_in, cf.heads, cf.c_out > 0 ? cf.c_out : 4 * cf.c_in), 0);
            break;
        }
        case SKY_MOD_WINDOWED_ATTENTION: {
            if (g.n < 1 || g.n > 2 || g.ndim[0] != 3) throw Error(SKY_ERR_SHAPE, "WindowedSelfAttention: x must be [B*nW, N, C] (+ optional mask [nW, N, N])");
            const int Bw = dim(0, 0), N = dim(0, 1), C = dim(0, 2);
            if (C != cf.c_in) throw Error(SKY_ERR_SHAPE, "WindowedSelfAttention: channel mismatch");
            int nW = 1;
            if (g.n == 2) {
                if (g.ndim[1] != 3 || dim(1, 1) != N || dim(1, 2) != N || Bw % dim(1, 0)) throw Error(SKY_ERR_SHAPE, "WindowedSelfAttention: mask must be [nW, N, N] with B_ % nW == 0");
                nW = dim(1, 0);
            }
            TV x = import_plain(c, 0, Bw, C, 1, N, nullptr, true);
            export_tokens(c, windowed_attention(c, "", x, C, cf.window_size, cf.heads, g.n == 2 ? 1 : -1, nW), 0);
            break;
        }
        case SKY_MOD_DECODE: {
            // process_detections(outputs, input_shape): inputs are the raw levels [B, na, gh, gw, no]
            const int nl = cf.num_levels, no = cf.nc + 5;
            if (g.n != nl) throw Error(SKY_ERR_SHAPE, "process_detections: expected one raw tensor per level");
            long rows = 0;
            for (int i = 0; i < nl; ++i) {
                if (g.ndim[i] != 5 || dim(i, 1) != cf.num_anchors || dim(i, 4) != no) throw Error(SKY_ERR_SHAPE, "process_detections: raw level must be [B, na, gh, gw, nc+5]");
                rows += (long)cf.num_anchors * dim(i, 2) * dim(i, 3);
            }
            long off = 0;
            for (int i = 0; i < nl; ++i) {
                Op op;
                op.kind = OP_DECODE;
                op.in.ext = i; op.in.B = dim(i, 0);
                op.level = i; op.Ho = dim(i, 2); op.Wo = dim(i, 3);
                op.det_ext = 16; op.det_rows = rows; op.det_off = off;
                op.stride_px = (float)std::max((double)cf.input_h / dim(i, 2), (double)cf.input_w / dim(i, 3));
                c.push(op);
                off += (long)cf.num_anchors * dim(i, 2) * dim(i, 3);
            }
            if (c.emit) {
                IoInfo io;
                io.ndim = 3;
                io.shape[0] = dim(0, 0); io.shape[1] = rows; io.shape[2] = no;
                c.e.out_info.assign(1, io);
            }
            break;
        }
        case SKY_MOD_UTILITY:
            break;
        default:
            throw Error(SKY_ERR_INVALID, "module kind " + std::to_string(cf.module) + " is not implemented in this build");
    }
}

// ------------------------------------------------------------------------------------------------ arena
static void place_buffers(Engine& e)
{
    // greedy first-fit by first use; buffers whose lifetimes overlap never share bytes
    std::vector<int> order;
    for (int i = 0; i < (int)e.bufs.size(); ++i)
        if (e.bufs[i].last >= 0) order.push_back(i);
    std::sort(order.begin(), order.end(), [&](int a, int b) {
        return e.bufs[a].first != e.bufs[b].first ? e.bufs[a].first < e.bufs[b].first : e.bufs[a].bytes > e.bufs[b].bytes;
    });
    std::vector<int> placed;
    size_t top = 0;
    for (int i : order) {
        Buffer& b = e.bufs[i];
        std::vector<std::pair<size_t, size_t>> busy;
        for (int j : placed) {
            const Buffer& o = e.bufs[j];
            if (o.last < b.first || o.first > b.last) continue;
            busy.push_back({o.offset, o.offset + o.bytes});
        }
        std::sort(busy.begin(), busy.end());
        size_t pos = 0;
        for (auto& iv : busy) {
            if (pos + b.bytes <= iv.first) break;
            pos = std::max(pos, iv.second);
        }
        b.offset = pos;
        top = std::max(top, pos + b.bytes);
        placed.push_back(i);
    }
    e.arena_bytes = std::max<size_t>(top, 256);
}

static void* tv_ptr(const Engine& e, const TV& t, const sky_buffer* ins, int n_in, const sky_buffer* outs, int n_out)
{
    if (t.ext >= 16) {
        const int j = t.ext - 16;
        if (j >= n_out) throw Error(SKY_ERR_INVALID, "output buffer " + std::to_string(j) + " not provided");
        return outs[j].data;
    }
    if (t.ext >= 0) {
        if (t.ext >= n_in) throw Error(SKY_ERR_INVALID, "input buffer " + std::to_string(t.ext) + " not provided");
        return ins[t.ext].data;
    }
    if (t.buf < 0) return nullptr;
    return e.arena + e.bufs[t.buf].offset + (size_t)t.off * dtype_size(t.dt);
}

// scale of the (fp8) buffer a view lives in: real value = stored value * scale; 1 for bf16 / fp32 tensors and caller buffers
static float tv_scale(const Engine& e, const TV& t)
{
    if (t.buf < 0 || t.dt != SKY_FP8) return 1.0f;
    int i = t.buf;
    while (e.bufs[i].tie >= 0) i = e.bufs[i].tie;
    return e.bufs[i].scale;
}

static void* buf_ptr(const Engine& e, int buf) { return buf < 0 ? nullptr : (void*)(e.arena + e.bufs[buf].offset); }

// amax (calibration pass of the fp8 engine's bf16 twin): device array, one slot per workspace buffer, max |x| over every
// tensor an op writes there
static void run(Engine& e, const sky_buffer* ins, int n_in, const sky_buffer* outs, int n_out, hipStream_t s,
                hipEvent_t* marks = nullptr, unsigned int* amax = nullptr)
{
    const sky_config& cf = e.cfg;
    int op_index = 0;
    if (marks) SKY_HIP(hipEventRecord(marks[0], s));
    bool took_next = false;     // the previous op's kernel also computed this (fused_prev) op
    int skip_csp = 0;           // ops of a CSP stage still to skip (the stage kernel computed them)
    const void* raw_src = nullptr;   // set by a skipped FocusBlock import: the next convolution reads the caller's frames
    int raw_mode = 0;
    bool stem_down_now = false;      // set by a skipped import: the stem op launches the fused stem + stride-2 kernel
    StemDownArgs stem_down_args;
    memset(&stem_down_args, 0, sizeof(stem_down_args));
    int sl_b0 = 0, sl_nb = -1;       // batch slice the ops run on (sl_nb < 0: the whole batch)
    unsigned head_done = 0;          // detection levels already computed by the kernel of the convolution that feeds them
    auto exec_op = [&](size_t oi) {
        Op& op = e.ops[oi];
        if (op.kind == OP_CONV && op.head && (head_done >> op.level & 1u)) {
            head_done &= ~(1u << op.level);
            op.variant = 9000;
            ++op_index;
            if (marks) SKY_HIP(hipEventRecord(marks[op_index], s));
            return;
        }
        if (op.fused_prev && took_next) {
            took_next = false;
            op.variant = 9000;
            ++op_index;
            if (marks) SKY_HIP(hipEventRecord(marks[op_index], s));
            return;
        }
        took_next = false;
        if (op.csp_member && skip_csp > 0) {       // computed by the CSP stage kernel launched at the stage's first op
            --skip_csp;
            op.variant = 9000;
            ++op_index;
            if (marks) SKY_HIP(hipEventRecord(marks[op_index], s));
            return;
        }
        switch (op.kind) {
            case OP_IMPORT: {
                const sky_buffer& src = ins[op.in.ext];
                // FocusBlock: when the convolution that follows runs on the kernel that reads the raw frames itself
                // (space-to-depth, /255 and conversion in its halo loader) the import is skipped
                if (op.s2d && op.src_c == 3 && src.layout != SKY_NHWC && !op.force_nhwc && src.dtype == SKY_IO_U8 && oi + 2 < e.ops.size() &&
                    e.ops[oi + 1].stem_down) {
                    // uint8 frames: import + stem + stride-2 convolution in one kernel (the stem op below launches it)
                    const Op& st = e.ops[oi + 1];
                    const Op& dn = e.ops[oi + 2];
                    StemDownArgs sa;
                    memset(&sa, 0, sizeof(sa));
                    sa.frames = (const unsigned char*)src.data;
                    sa.B = st.in.B; sa.Hr = op.src_h; sa.Wr = op.src_w;
                    if (sl_nb >= 0) {
                        sa.frames += (size_t)sl_b0 * 3 * op.src_h * op.src_w;
                        sa.B = sl_nb;
                    }
                    sa.w1 = e.convs[st.wid].w; sa.bias1 = e.convs[st.wid].bias; sa.kpad1 = e.convs[st.wid].Kpad;
                    sa.w2 = e.convs[dn.wid].w; sa.bias2 = e.convs[dn.wid].bias; sa.kpad2 = e.convs[dn.wid].Kpad;
                    sa.out = (char*)tv_ptr(e, dn.out, ins, n_in, outs, n_out) + (sl_nb >= 0 ? (size_t)sl_b0 * dn.Ho * dn.Wo * dn.out.ld * 2 : 0);
                    sa.Ho = dn.Ho; sa.Wo = dn.Wo; sa.ldo = dn.out.ld; sa.c1 = st.cout; sa.c2 = dn.cout;
                    sa.opts = e.opts; sa.device = cf.device; sa.n_cu = e.n_cu;
                    if (stem_down_supported(sa)) {
                        stem_down_args = sa;
                        stem_down_now = true;
                        op.variant = 9100;
                        break;
                    }
                }
                if (op.s2d && op.src_c == 3 && src.layout != SKY_NHWC && !op.force_nhwc && oi + 1 < e.ops.size()) {
                    Op& nx = e.ops[oi + 1];
                    if (nx.kind == OP_CONV && nx.in.buf == op.out.buf && nx.in.off == op.out.off && nx.ks == 3 && nx.stride == 1 && !nx.res.valid()) {
                        ConvArgs t;
                        memset(&t, 0, sizeof(t));
                        t.B = nx.in.B; t.H = nx.in.H; t.W = nx.in.W; t.Cin = nx.cin; t.ldi = nx.in.ld; t.Ho = nx.Ho; t.Wo = nx.Wo; t.Cout = nx.cout;
                        t.ldo = nx.out.ld; t.ks = 3; t.stride = 1; t.pad = 1; t.Kpad = e.convs[nx.wid].Kpad; t.up2 = nx.up2; t.head = nx.head;
                        t.M = nx.in.B * nx.Ho * nx.Wo;
                        t.opts = e.opts; t.device = cf.device; t.n_cu = e.n_cu;
                        t.out_dt = nx.out.dt;
                        const double oext = (((double)t.M - 1.0) * nx.out.ld + nx.cout) * dtype_size(nx.out.dt);
                        // 2 GiB or more: the convolution is launched over batch slices (below), each within the 32-bit range
                        t.out_bytes = oext < 2147483000.0 ? (unsigned)oext : (unsigned)(((double)nx.Ho * nx.Wo - 1.0) * nx.out.ld * dtype_size(nx.out.dt) + 1);
                        if (conv_accepts_raw(nx.cdt, t)) {
                            raw_src = src.data;      // (the convolution's own slicing below offsets it)
                            raw_mode = src.dtype == SKY_IO_U8 ? 1 : 2;
                            op.variant = 9100;
                            break;
                        }
                    }
                }
                {
                    const size_t sbytes = (size_t)op.src_c * op.src_h * op.src_w * (src.dtype == SKY_IO_U8 ? 1 : 4);
                    const size_t dbytes = (size_t)op.out.H * op.out.W * op.out.ld * dtype_size(op.out.dt);
                    const int b0 = sl_nb >= 0 ? sl_b0 : 0, nb = sl_nb >= 0 ? sl_nb : op.out.B;
                    SKY_HIP(launch_import(op.out.dt, (const char*)src.data + b0 * sbytes, src.dtype == SKY_IO_U8, op.force_nhwc || src.layout == SKY_NHWC,
                                          (char*)tv_ptr(e, op.out, ins, n_in, outs, n_out) + b0 * dbytes, nb, op.src_c, op.src_h, op.src_w, op.out.C, op.out.ld,
                                          op.s2d, src.dtype == SKY_IO_U8, s, 1.0f / tv_scale(e, op.out)));
                }
                break;
            }
            case OP_EXPORT:
                SKY_HIP(launch_export(op.in.dt, tv_ptr(e, op.in, ins, n_in, outs, n_out), op.in.ld, (float*)tv_ptr(e, op.out, ins, n_in, outs, n_out),
                                      op.in.B, op.in.C, op.in.H, op.in.W, s, tv_scale(e, op.in)));
                break;
            case OP_CONV: {
                if (stem_down_now) {
                    stem_down_now = false;
                    SKY_HIP(launch_stem_down(stem_down_args, s));
                    op.variant = 8064;
                    took_next = true;                  // the stride-2 convolution behind this op is done too
                    break;
                }
                if (op.csp_stage && oi + 3 < e.ops.size() && !amax) {
                    // CSPBlock(64, 64, n = 1): cv1|cv2, bottleneck cv1, bottleneck cv2 (+ shortcut), cv3 in one launch
                    const Op &b1 = e.ops[oi + 1], &b2 = e.ops[oi + 2], &c3 = e.ops[oi + 3];
                    CspStageArgs ca;
                    memset(&ca, 0, sizeof(ca));
                    const int b0 = sl_nb >= 0 ? sl_b0 : 0, nb = sl_nb >= 0 ? sl_nb : op.in.B;
                    ca.in = (const char*)tv_ptr(e, op.in, ins, n_in, outs, n_out) + (size_t)b0 * op.in.H * op.in.W * op.in.ld * 2;
                    ca.ldi = op.in.ld;
                    ca.out = (char*)tv_ptr(e, c3.out, ins, n_in, outs, n_out) + (size_t)b0 * op.in.H * op.in.W * c3.out.ld * 2;
                    ca.ldo = c3.out.ld;
                    ca.B = nb; ca.H = op.in.H; ca.W = op.in.W; ca.c = op.cin; ca.hidden = b1.cout;
                    const double ext = ((double)nb * op.in.H * op.in.W - 1.0) * op.in.ld * 2 + (double)op.cin * 2;
                    ca.in_bytes = ext < 2147483000.0 ? (unsigned)ext : 0u;
                    const DevConv &d12 = e.convs[op.wid], &db1 = e.convs[b1.wid], &db2 = e.convs[b2.wid], &d3 = e.convs[c3.wid];
                    ca.w12 = d12.w; ca.b12 = d12.bias; ca.kpad12 = d12.Kpad;
                    ca.wb1 = db1.w; ca.bb1 = db1.bias; ca.kpadb1 = db1.Kpad;
                    ca.wb2 = db2.w; ca.bb2 = db2.bias; ca.kpadb2 = db2.Kpad;
                    ca.w3 = d3.w; ca.b3 = d3.bias; ca.kpad3 = d3.Kpad;
                    ca.shortcut = op.csp_shortcut;
                    ca.opts = e.opts; ca.device = cf.device; ca.n_cu = e.n_cu;
                    if (csp_stage_supported(ca)) {
                        SKY_HIP(launch_csp_stage(ca, s));
                        op.variant = 8564;
                        skip_csp = 3;
                        break;
                    }
                }
                ConvArgs a;
                memset(&a, 0, sizeof(a));
                const DevConv& d = e.convs[op.wid];
                a.in = tv_ptr(e, op.in, ins, n_in, outs, n_out);
                if (raw_src) { a.in = raw_src; a.src_mode = raw_mode; raw_src = nullptr; }
                a.w = d.w; a.bias = d.bias; a.Kpad = d.Kpad; a.zero = e.zero_page;
                a.opts = e.opts; a.device = cf.device; a.n_cu = e.n_cu;
                a.B = op.in.B; a.H = op.in.H; a.W = op.in.W; a.Cin = op.cin; a.ldi = op.in.ld;
                a.Ho = op.Ho; a.Wo = op.Wo; a.Cout = op.cout;
                a.ks = op.ks; a.stride = op.stride; a.pad = op.ks / 2;
                a.act = op.act; a.up2 = op.up2;
                a.M = op.in.B * op.Ho * op.Wo;
                {
                    const int ies = dtype_size(op.cdt);
                    const double ext = ((double)op.in.B * op.in.H * op.in.W - 1.0) * op.in.ld * ies + (double)op.cin * ies;
                    a.in_bytes = ext < 2147483000.0 ? (unsigned)ext : 0u;   // offsets are computed in int32
                }
                a.mult = d.mult;                       // fp8 operands only (nullptr otherwise)
                a.out_inv_scale = 1.0f; a.res_scale = 1.0f; a.out_dt = -1;
                if (op.head) {
                    a.head = 1;
                    TV r; r.ext = op.raw_ext;
                    TV dt; dt.ext = op.det_ext;
                    a.raw = (float*)tv_ptr(e, r, ins, n_in, outs, n_out);       // may be NULL: the caller does not want the raw level
                    a.det = (float*)tv_ptr(e, dt, ins, n_in, outs, n_out);
                    a.na = cf.num_anchors; a.no = cf.nc + 5;
                    a.det_rows = op.det_rows; a.det_off = op.det_off; a.stride_px = op.stride_px;
                    for (int k = 0; k < cf.num_anchors * 2; ++k)
                        a.anchor_wh[k] = cf.anchors[op.level * cf.num_anchors * 2 + k] * op.stride_px;   // detector.py:119-121 (D13)
                } else {
                    a.out = tv_ptr(e, op.out, ins, n_in, outs, n_out);
                    a.ldo = op.out.ld;
                    const double opix = (double)a.M * (op.up2 ? 4.0 : 1.0) - 1.0;
                    const int oes = dtype_size(op.out.dt);
                    a.out_dt = op.out.dt;
                    a.out_inv_scale = 1.0f / tv_scale(e, op.out);
                    const double oext = (opix * op.out.ld + op.cout) * oes;
                    a.out_bytes = oext < 2147483000.0 ? (unsigned)oext : 0u;
                    if (op.res.valid()) {
                        a.res = tv_ptr(e, op.res, ins, n_in, outs, n_out);
                        a.ldr = op.res.ld;
                        if (op.res.dt != op.out.dt) throw Error(SKY_ERR_INVALID, "residual and output element types differ");
                        a.res_scale = tv_scale(e, op.res);
                        const double rext = (((double)a.M - 1.0) * op.res.ld + op.cout) * oes;
                        a.res_bytes = rext < 2147483000.0 ? (unsigned)rext : 0u;
                    }
                }
                if (op.fuse_next && oi + 1 < e.ops.size()) {
                    const Op& f = e.ops[oi + 1];
                    const DevConv& d2 = e.convs[f.wid];
                    a.f2_w = d2.w; a.f2_bias = d2.bias; a.f2_Kpad = d2.Kpad;
                    a.f2_out = tv_ptr(e, f.out, ins, n_in, outs, n_out);
                    a.f2_cin = f.cin; a.f2_cout = f.cout; a.f2_ldo = f.out.ld; a.f2_act = f.act;
                    a.f2_koff = (int)(f.in.off - op.out.off);
                    const double fext = (((double)a.M - 1.0) * f.out.ld + f.cout) * dtype_size(f.out.dt);
                    a.f2_out_bytes = fext < 2147483000.0 ? (unsigned)fext : 0u;
                }
                if (op.in2.valid() && op.in2_cin > 0) {
                    a.in2 = tv_ptr(e, op.in2, ins, n_in, outs, n_out);
                    a.in2_cin = op.in2_cin; a.ldi2 = op.in2.ld; a.in2_up2 = op.in2_up2;
                    const double e2 = ((double)op.in2.B * op.in2.H * op.in2.W - 1.0) * op.in2.ld * dtype_size(op.cdt) + (double)op.in2_cin * dtype_size(op.cdt);
                    a.in2_bytes = e2 < 2147483000.0 ? (unsigned)e2 : 0u;
                }
                if (op.wid1 >= 0) {
                    const DevConv& d1 = e.convs[op.wid1];
                    a.c1_w = d1.w; a.c1_bias = d1.bias; a.c1_Kpad = d1.Kpad; a.c1_res = op.c1_res;
                    if (op.cdt == SKY_FP8) {
                        a.c1_mult = d1.mult;
                        a.c1_out_inv_scale = 1.0f / tv_scale(e, op.hid);
                        a.res_scale = tv_scale(e, op.in);            // the shortcut is x itself
                    }
                }
                if (op.head_op >= 0 && !amax && sl_nb < 0 && !a.f2_w && !a.src_mode) {
                    // CSP cv3 + the detection level that reads it: one kernel (k_head.hip), the level's op is skipped when it comes
                    const Op& h = e.ops[op.head_op];
                    const DevConv& dh = e.convs[h.wid];
                    ConvArgs f = a;
                    f.f2_w = dh.w; f.f2_bias = dh.bias; f.f2_Kpad = dh.Kpad;
                    TV r; r.ext = h.raw_ext;
                    TV dt; dt.ext = h.det_ext;
                    f.raw = (float*)tv_ptr(e, r, ins, n_in, outs, n_out);
                    f.det = (float*)tv_ptr(e, dt, ins, n_in, outs, n_out);
                    f.na = cf.num_anchors; f.no = cf.nc + 5;
                    f.det_rows = h.det_rows; f.det_off = h.det_off; f.stride_px = h.stride_px;
                    for (int k = 0; k < cf.num_anchors * 2; ++k) f.anchor_wh[k] = cf.anchors[h.level * cf.num_anchors * 2 + k] * h.stride_px;
                    const double fout = (double)a.M * a.ldo * 2;
                    if (h.level < 32 && a.in_bytes != 0 && fout < 2147483000.0 && tv_scale(e, op.out) == 1.0f && cv3_head_supported(op.cdt, f)) {
                        const hipError_t he = launch_cv3_head(op.cdt, f, s);
                        if (he == hipSuccess) {
                            op.variant = 1628;
                            head_done |= 1u << h.level;
                            break;
                        }
                        if (he != hipErrorNotSupported) SKY_HIP(he);
                    }
                }
                int fused = 0;
                // The kernels address a view with 32-bit byte offsets (buffer descriptors): a view of 2 GiB or more (skyeye_l's
                // 64-channel 768 x 768 maps at B = 32) is run as several launches over batch slices, each below the limit.
                const bool raw_in = a.src_mode != 0;
                const double in_img = raw_in ? 3.0 * (2.0 * a.H) * (2.0 * a.W) * (a.src_mode == 1 ? 1 : 4)
                                             : (double)a.H * a.W * a.ldi * dtype_size(op.cdt);
                const double out_img = op.head ? 0.0 : (double)a.Ho * a.Wo * (op.up2 ? 4.0 : 1.0) * a.ldo * dtype_size(op.out.dt);
                const double res_img = a.res ? (double)a.Ho * a.Wo * a.ldr * dtype_size(op.out.dt) : 0.0;
                const double worst = std::max(in_img, std::max(out_img, res_img));
                int bs = a.B;
                if (worst * a.B >= 2147483000.0 && !a.f2_w) bs = std::max(1, (int)(2147483000.0 / worst));
                const int first = sl_nb >= 0 ? sl_b0 : 0, last = sl_nb >= 0 ? sl_b0 + sl_nb : a.B;      // the batch range of this call
                if (sl_nb >= 0 && a.f2_w) throw Error(SKY_ERR_STATE, "sub-batch section with an epilogue-fused pair");
                if (bs >= a.B && sl_nb < 0) {
                    SKY_HIP(launch_conv(op.cdt, a, s, &op.variant, &fused));
                } else {
                    const ConvArgs a0 = a;
                    for (int b0 = first; b0 < last; b0 += bs) {
                        const int nb = std::min(bs, last - b0);
                        a = a0;
                        a.B = nb;
                        a.M = nb * a.Ho * a.Wo;
                        a.in = (const char*)a0.in + (size_t)(in_img * b0);
                        if (a0.in2) {       // the second input of a 1x1 convolution: its own image size
                            const double img2 = (double)op.in2.H * op.in2.W * a.ldi2 * dtype_size(op.cdt);
                            a.in2 = (const char*)a0.in2 + (size_t)(img2 * b0);
                            const double e2 = ((double)nb * op.in2.H * op.in2.W - 1.0) * a.ldi2 * dtype_size(op.cdt) + (double)op.in2_cin * dtype_size(op.cdt);
                            a.in2_bytes = e2 < 2147483000.0 ? (unsigned)e2 : 0u;
                        }
                        const double iext = raw_in ? in_img * nb : ((double)nb * a.H * a.W - 1.0) * a.ldi * dtype_size(op.cdt) + (double)op.cin * dtype_size(op.cdt);
                        a.in_bytes = iext < 2147483000.0 ? (unsigned)iext : 0u;
                        if (op.head) {
                            if (a0.raw) a.raw = a0.raw + (size_t)b0 * a.na * a.Ho * a.Wo * a.no;
                            a.det = a0.det + (size_t)b0 * a.det_rows * a.no;
                        } else {
                            const int oes = dtype_size(op.out.dt);
                            a.out = (char*)a0.out + (size_t)(out_img * b0);
                            const double oext = (((double)a.M * (op.up2 ? 4.0 : 1.0) - 1.0) * a.ldo + op.cout) * oes;
                            a.out_bytes = oext < 2147483000.0 ? (unsigned)oext : 0u;
                            if (a0.res) {
                                a.res = (const char*)a0.res + (size_t)(res_img * b0);
                                const double rext = (((double)a.M - 1.0) * a.ldr + op.cout) * oes;
                                a.res_bytes = rext < 2147483000.0 ? (unsigned)rext : 0u;
                            }
                        }
                        SKY_HIP(launch_conv(op.cdt, a, s, &op.variant, &fused));
                    }
                }
                took_next = fused != 0;
                break;
            }
            case OP_MAXPOOL5:
                if (op.win == 3) {
                    SKY_HIP(launch_spp_pyramid(op.in.dt, tv_ptr(e, op.in, ins, n_in, outs, n_out), op.in.ld, tv_ptr(e, op.out, ins, n_in, outs, n_out),
                                               op.out.ld, op.in.B, op.in.H, op.in.W, op.in.C, op.in.C, s));
                    break;
                }
                SKY_HIP(launch_maxpool5(op.in.dt, tv_ptr(e, op.in, ins, n_in, outs, n_out), op.in.ld, tv_ptr(e, op.out, ins, n_in, outs, n_out),
                                        op.out.ld, op.in.B, op.in.H, op.in.W, op.in.C, s));
                break;
            case OP_UPSAMPLE:
                SKY_HIP(launch_upsample(op.in.dt, tv_ptr(e, op.in, ins, n_in, outs, n_out), op.in.ld, tv_ptr(e, op.out, ins, n_in, outs, n_out),
                                        op.out.ld, op.in.B, op.in.H, op.in.W, op.in.C, op.out.H, op.out.W, s));
                break;
            case OP_CA_REDUCE:
                SKY_HIP(launch_ca_reduce(op.in.dt, tv_ptr(e, op.in, ins, n_in, outs, n_out), op.in.ld, op.in.B, op.in.H * op.in.W, op.in.C,
                                         op.nchunk, (float*)buf_ptr(e, op.s0), s, tv_scale(e, op.in)));
                break;
            case OP_CA_MLP:
                SKY_HIP(launch_ca_mlp((const float*)buf_ptr(e, op.s0), op.in.B, op.in.H * op.in.W, op.in.C, op.nchunk, op.R, e.fweights[op.f0],
                                      e.fweights[op.f1], (float*)buf_ptr(e, op.s1), s));
                break;
            case OP_SA_STATS:
                SKY_HIP(launch_sa_stats(op.in.dt, tv_ptr(e, op.in, ins, n_in, outs, n_out), op.in.ld, (const float*)buf_ptr(e, op.s0), op.in.B,
                                        op.in.H * op.in.W, op.in.C, (float*)buf_ptr(e, op.s1), s, tv_scale(e, op.in)));
                break;
            case OP_SA_GATE:
                SKY_HIP(launch_sa_gate((const float*)buf_ptr(e, op.s0), e.fweights[op.f0], op.in.B, op.in.H, op.in.W, (float*)buf_ptr(e, op.s1), s));
                break;
            case OP_LAYERNORM:
                SKY_HIP(launch_layernorm(e.dtype, tv_ptr(e, op.in, ins, n_in, outs, n_out), op.in.ld, tv_ptr(e, op.out, ins, n_in, outs, n_out),
                                         op.out.ld, e.fweights[op.f0], e.fweights[op.f1], (long)op.in.B * op.in.H * op.in.W, op.in.C, s));
                break;
            case OP_ATTENTION:
                SKY_HIP(launch_attention(e.dtype, tv_ptr(e, op.in, ins, n_in, outs, n_out), op.in.ld, tv_ptr(e, op.out, ins, n_in, outs, n_out),
                                         op.out.ld, op.win ? op.in.B * (op.in.H / op.win) * (op.in.W / op.win) : op.in.B, op.ntok, op.out.C, op.heads,
                                         op.scale, op.f0 >= 0 ? e.fweights[op.f0] : nullptr,
                                         op.mask_ext >= 0 ? (const float*)ins[op.mask_ext].data : nullptr, op.nW, op.win, op.in.H, op.in.W, s, e.opts));
                break;
            case OP_CLA:
                SKY_HIP(launch_cla(e.dtype, tv_ptr(e, op.in, ins, n_in, outs, n_out), op.in.ld, tv_ptr(e, op.in2, ins, n_in, outs, n_out),
                                   op.in2.ld, op.v_off, (float*)buf_ptr(e, op.s0), tv_ptr(e, op.out, ins, n_in, outs, n_out), op.out.ld, op.in.B,
                                   op.in.H, op.in.W, op.in2.H, op.in2.W, op.out.C, op.heads, op.scale, op.r2, s));
                break;
            case OP_EXPORT_TOKENS:
                SKY_HIP(launch_export_tokens(e.dtype, tv_ptr(e, op.in, ins, n_in, outs, n_out), op.in.ld, (float*)tv_ptr(e, op.out, ins, n_in, outs, n_out),
                                             (long)op.in.B * op.in.H * op.in.W, op.in.C, s));
                break;
            case OP_DECODE: {
                float awh[16];
                for (int k = 0; k < cf.num_anchors * 2; ++k) awh[k] = cf.anchors[op.level * cf.num_anchors * 2 + k] * op.stride_px;
                TV dt; dt.ext = op.det_ext;
                SKY_HIP(launch_decode(e.dtype, (const float*)ins[op.in.ext].data, (float*)tv_ptr(e, dt, ins, n_in, outs, n_out), op.in.B, cf.num_anchors,
                                      op.Ho, op.Wo, cf.nc + 5, op.det_rows, op.det_off, op.stride_px, awh, s));
                break;
            }
            case OP_SCALE:
                SKY_HIP(launch_scale(op.in.dt, tv_ptr(e, op.in, ins, n_in, outs, n_out), op.in.ld, (const float*)buf_ptr(e, op.s0),
                                     (const float*)buf_ptr(e, op.s1), tv_ptr(e, op.out, ins, n_in, outs, n_out), op.out.ld, op.in.B,
                                     op.in.H * op.in.W, op.in.C, s, tv_scale(e, op.in), 1.0f / tv_scale(e, op.out)));
                break;
        }
        if (amax && op.out.buf >= 0 && op.out.dt != SKY_FP8 && op.out.B > 0)
            SKY_HIP(launch_amax(op.out.dt, tv_ptr(e, op.out, ins, n_in, outs, n_out), op.out.ld, (long)op.out.B * op.out.H * op.out.W, op.out.C,
                                amax + op.out.buf, s));
        ++op_index;
        if (marks) SKY_HIP(hipEventRecord(marks[op_index], s));
    };
    size_t oi = 0;
    if (e.sec_end > 0 && !marks && !amax) {       // (per-op profiling and calibration passes run the plain order)
        const int B = e.ops[0].out.B;
        for (int b0 = 0; b0 < B; b0 += e.sec_sub) {
            sl_b0 = b0;
            sl_nb = std::min(e.sec_sub, B - b0);
            for (size_t k = 0; k < (size_t)e.sec_end; ++k) exec_op(k);
        }
        sl_b0 = 0;
        sl_nb = -1;
        oi = (size_t)e.sec_end;
    }
    for (; oi < e.ops.size(); ++oi) exec_op(oi);
}

static void collect_spec(Engine& e)
{
    e.spec.clear();
    Ctx c(e, false);
    Geometry g;
    const sky_config& cf = e.cfg;
    auto set = [&](int i, int b, int ch, int h, int w) {
        g.ndim[i] = 4;
        g.shape[i][0] = b; g.shape[i][1] = ch; g.shape[i][2] = h; g.shape[i][3] = w; g.shape[i][4] = 0;
    };
    switch (cf.module) {
        case SKY_MOD_DETECTOR: case SKY_MOD_ENHANCED_DETECTOR: case SKY_MOD_BACKBONE:
            g.n = 1; set(0, 1, cf.in_channels, 64, 64); break;
        case SKY_MOD_FOCUS:
            g.n = 1; set(0, 1, cf.c_in, 64, 64); break;
        case SKY_MOD_NECK:
            g.n = 3;
            for (int i = 0; i < 3; ++i) set(i, 1, cf.level_channels[i], 8 >> i, 8 >> i);
            break;
        case SKY_MOD_HEAD:
            g.n = cf.num_levels;
            for (int i = 0; i < cf.num_levels; ++i) set(i, 1, cf.level_channels[i], 8, 8);
            break;
        case SKY_MOD_CHANNEL_ATTENTION: case SKY_MOD_SPATIAL_ATTENTION: case SKY_MOD_COMBINED_ATTENTION:
            g.n = 1; set(0, 1, cf.c_in > 0 ? cf.c_in : 16, 8, 8); break;   // SpatialAttention() has no channel argument
        case SKY_MOD_DECODE: case SKY_MOD_UTILITY:
            return;   // no parameters
        case SKY_MOD_CROSS_LAYER_ATTENTION:
            g.n = 2; set(0, 1, cf.c_in, 8, 8); set(1, 1, cf.key_channels > 0 ? cf.key_channels : cf.c_in, 4, 4); break;
        case SKY_MOD_WINDOWED_ATTENTION:
            g.n = 1; g.ndim[0] = 3; g.shape[0][0] = 1; g.shape[0][1] = (int64_t)cf.window_size * cf.window_size; g.shape[0][2] = cf.c_in; break;
        default:
            g.n = 1; set(0, 1, cf.c_in, 8, 8); break;
    }
    build(c, g);
}

// The developer switches of sky_kernels.h (PlanOpt), read from the environment once per plan.
static unsigned read_plan_opts()
{
    unsigned o = 0;
    auto env = [](const char* k) { return getenv(k); };
    if (const char* v = env("SKY_CONV_HALO")) o |= v[0] == '0' ? OPT_HALO_OFF : v[0] == 'f' ? OPT_HALO_FORCE : 0u;
    if (const char* v = env("SKY_HALO_NF8")) o |= v[0] == 'o' ? OPT_NF8_OFF : v[0] == 's' ? OPT_NF8_SOLO : 0u;
    if (const char* v = env("SKY_HALO_S2")) o |= v[0] == '0' ? OPT_S2_OFF : 0u;
    if (env("SKY_NO_STREAM")) o |= OPT_NO_STREAM;
    if (env("SKY_NO_RING")) o |= OPT_NO_RING;
    if (env("SKY_STREAM_OLDGRID")) o |= OPT_OLDGRID;
    if (env("SKY_NO_FUSED_IMPORT")) o |= OPT_NO_FUSED_IMPORT;
    if (env("SKY_FUSE") && !env("SKY_NO_FUSE")) o |= OPT_FUSE | OPT_NO_FUSE_CV1;     // the older opt-in epilogue fusion: instead of the fused bottlenecks
    if (env("SKY_NO_SPP_PYRAMID")) o |= OPT_NO_SPP_PYRAMID;
    if (env("SKY_ATTN_VALU")) o |= OPT_ATTN_VALU;
    if (env("SKY_NO_FUSE_CV1")) o |= OPT_NO_FUSE_CV1;
    if (env("SKY_NO_STEM_DOWN")) o |= OPT_NO_STEM_DOWN;
    if (env("SKY_NO_WINATTN")) o |= OPT_NO_WINATTN;
    if (env("SKY_NO_CSP_STAGE")) o |= OPT_NO_CSP_STAGE;
    if (env("SKY_NO_HEAD_STREAM")) o |= OPT_NO_HEAD_STREAM;
    if (env("SKY_NO_BNECK128")) o |= OPT_NO_BNECK128;
    if (env("SKY_NO_BNECK64W")) o |= OPT_NO_BNECK64W;
    if (const char* v = env("SKY_BNECK128")) o |= v[0] == 's' ? OPT_BNECK128_SOLO : v[0] == 'p' ? OPT_BNECK_PAIR : 0u;
    if (env("SKY_NO_DEEP3X3")) o |= OPT_NO_DEEP3X3;
    if (env("SKY_NO_IN2")) o |= OPT_NO_IN2;
    if (env("SKY_NO_CV3_HEAD")) o |= OPT_NO_CV3_HEAD;
    if (env("SKY_NO_GEMM1X1")) o |= OPT_NO_GEMM1X1;
    if (const char* v = env("SKY_GEMM1X1")) o |= v[0] == 'f' ? OPT_GEMM1X1_FORCE : 0u;
    if (const char* v = env("SKY_HEAD_STREAM")) o |= v[0] == 'f' ? OPT_HEAD_STREAM_FORCE : 0u;
    if (const char* v = env("SKY_HALO_SKIP")) o |= ((unsigned)atoi(v) & 31u) << OPT_SKIP_SHIFT;
    return o;
}

// Every entry point that launches work runs on the handle's own device, whatever the caller's current device is.
struct DeviceGuard {
    int prev = -1, dev;
    explicit DeviceGuard(int d) : dev(d)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) SKY_HIP(hipSetDevice(dev));
    }
    ~DeviceGuard()
    {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
};

// The sub-batch section of a detector plan (Engine::sec_end): the leading run of convolutions on maps of at least 1/8 of the frame
// side.  Opt-in, SKY_SUBBATCH=<frames>: MEASURED SLOWER on MI355X (skyeye_s bf16 B = 32 @1280: 5 780 frames/s whole-batch order,
// 5 724 / 5 514 / 5 201 with slices of 16 / 8 / 4 frames) -- the producer's stores do not stay in the Infinity Cache for the consumer,
// and the slices quarter the parallelism of every launch.  Kept as a switch for the record (experiments/README.md).  Buffers the
// section touches are made to coexist: slice k + 1 of an early layer runs after slice k of a later one.
static void plan_section(Engine& e)
{
    e.sec_end = e.sec_sub = 0;
    const char* v = getenv("SKY_SUBBATCH");
    const int sub = v ? atoi(v) : 0;
    if (sub <= 0 || e.ops.size() < 4 || (e.cfg.module != SKY_MOD_DETECTOR && e.cfg.module != SKY_MOD_ENHANCED_DETECTOR)) return;
    if (e.ops[0].kind != OP_IMPORT || e.ops[0].out.B < 2 * sub) return;
    const int Hmin = e.ops[1].in.H / 4;
    size_t k = 1;
    for (; k < e.ops.size(); ++k) {
        const Op& op = e.ops[k];
        const bool folded = op.fused_prev && e.ops[k - 1].stem_down;          // the stride-2 convolution the stem kernel computes
        if (op.kind != OP_CONV || op.head || op.fuse_next || (op.fused_prev && !folded) || op.in.H < Hmin || op.in.B != e.ops[0].out.B) break;
    }
    if (k < 4) return;
    e.sec_end = (int)k;
    e.sec_sub = sub;
    for (size_t i = 0; i < k; ++i)
        for (int b : {e.ops[i].in.buf, e.ops[i].out.buf, e.ops[i].res.buf})
            if (b >= 0) {
                e.bufs[b].first = 0;
                e.bufs[b].last = std::max(e.bufs[b].last, (int)k - 1);
            }
}

static void plan(Engine& e, const Geometry& g)
{
    e.free_plan();
    e.opts = (read_plan_opts() | e.extra_opts) & ~e.mask_opts;
    if (e.dtype == SKY_FP8) e.opts &= ~(unsigned)OPT_FUSE;
    {
        hipDeviceProp_t prop;
        SKY_HIP(hipGetDeviceProperties(&prop, e.cfg.device));
        e.n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    e.flops = e.act_bytes = e.weight_bytes = 0;
    e.out_info.clear();
    e.spec.clear();
    Ctx c(e, true);
    build(c, g);
    plan_section(e);
    place_buffers(e);
    SKY_HIP(hipMalloc(&e.arena, e.arena_bytes));
    SKY_HIP(hipMalloc(&e.zero_page, 256));
    SKY_HIP(hipMemset(e.zero_page, 0, 256));
    for (const Op& op : e.ops) {
        auto bytes = [&](const TV& t) { return t.valid() ? (double)t.B * t.H * t.W * t.C * (t.buf >= 0 ? dtype_size(t.dt) : e.esize()) : 0.0; };
        e.act_bytes += bytes(op.in) + bytes(op.out) + bytes(op.res);
    }
    e.planned = true;
    e.weights_dirty = false;
    e.calibrated = e.dtype != SKY_FP8;
}

// fp8: per-output-channel multipliers of every fp8 convolution = scale of its input tensor x weight scale of the channel
static void apply_scales(Engine& e)
{
    for (const Op& op : e.ops) {
        if (op.kind != OP_CONV || op.cdt != SKY_FP8) continue;
        DevConv& d = e.convs[op.wid];
        const float si = op.wid1 >= 0 ? tv_scale(e, op.hid) : tv_scale(e, op.in);      // fused bottleneck: the 3x3 reads the hidden tensor
        std::vector<float> m(d.rows);
        for (int r = 0; r < d.rows; ++r) m[r] = si * d.w_scale[r];
        SKY_HIP(hipMemcpy(d.mult, m.data(), m.size() * sizeof(float), hipMemcpyHostToDevice));
        if (op.wid1 >= 0) {
            DevConv& d1 = e.convs[op.wid1];
            const float s1 = tv_scale(e, op.in);
            std::vector<float> m1(d1.rows);
            for (int r = 0; r < d1.rows; ++r) m1[r] = s1 * d1.w_scale[r];
            SKY_HIP(hipMemcpy(d1.mult, m1.data(), m1.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
}

static Geometry geometry_of(int n_inputs, const sky_buffer* in)
{
    Geometry g;
    g.n = n_inputs;
    for (int i = 0; i < n_inputs; ++i) {
        g.ndim[i] = in[i].ndim;
        for (int k = 0; k < 5; ++k) g.shape[i][k] = k < in[i].ndim ? in[i].shape[k] : 0;
        // boundary tensors are described logically as [B, C, H, W] whatever their memory layout
    }
    return g;
}

// Activation scales of the fp8 engine.  The same graph is planned once more as a bf16 "twin" on the calibration inputs (any batch
// size; the builder creates the workspace buffers in the same order for every dtype and geometry), run with an amax reduction
// behind every launch, and every fp8 buffer gets scale = max |x| / 448: the largest calibration value maps to the largest e4m3.
static void calibrate(Engine& e, int n_inputs, const sky_buffer* inputs, hipStream_t s)
{
    if (e.dtype != SKY_FP8) throw Error(SKY_ERR_STATE, "sky_calibrate: only the fp8 engine has activation scales");
    if (!e.planned) throw Error(SKY_ERR_STATE, "sky_calibrate before sky_plan");
    Engine tw;
    tw.cfg = e.cfg;
    tw.cfg.dtype = SKY_BF16;
    tw.dtype = SKY_BF16;
    tw.weights = e.weights;
    tw.extra_opts = OPT_NO_FUSE_CV1 | OPT_NO_STEM_DOWN | OPT_NO_CSP_STAGE | OPT_NO_IN2;      // same buffer list as the fp8 plan (fused bottlenecks add scratch tensors), every tensor materialised
    tw.mirror_dtype = SKY_FP8;          // ... including the small lateral maps that plan's CSP GEMMs read in place (neck(): lateral)
    tw.mirror_opts = e.opts;
    tw.mask_opts = OPT_FUSE;            // like the fp8 plan itself: an op computed in its producer's epilogue would never reach its amax reduction
    plan(tw, geometry_of(n_inputs, inputs));
    if (tw.bufs.size() != e.bufs.size()) throw Error(SKY_ERR_STATE, "sky_calibrate: the bf16 twin has another buffer list than the fp8 plan");
    std::vector<sky_buffer> outs(tw.out_info.size());
    std::vector<void*> tmp;
    std::vector<char> optional(outs.size(), 0);
    for (const Op& op : tw.ops)
        if (op.kind == OP_CONV && op.head && op.raw_ext >= 16 && op.raw_ext - 16 < (int)outs.size()) optional[op.raw_ext - 16] = 1;
    unsigned int* amax = nullptr;
    auto cleanup = [&] {
        for (void* p : tmp) (void)hipFree(p);
        if (amax) (void)hipFree(amax);
    };
    try {
        for (size_t i = 0; i < outs.size(); ++i) {
            memset(&outs[i], 0, sizeof(sky_buffer));
            if (optional[i]) continue;
            size_t n = 1;
            for (int k = 0; k < tw.out_info[i].ndim; ++k) n *= (size_t)tw.out_info[i].shape[k];
            void* p = nullptr;
            SKY_HIP(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(float)));
            tmp.push_back(p);
            outs[i].data = p;
        }
        SKY_HIP(hipMalloc(&amax, tw.bufs.size() * sizeof(unsigned int)));
        SKY_HIP(hipMemsetAsync(amax, 0, tw.bufs.size() * sizeof(unsigned int), s));
        run(tw, inputs, n_inputs, outs.data(), (int)outs.size(), s, nullptr, amax);
        std::vector<float> host(tw.bufs.size());
        SKY_HIP(hipMemcpyAsync(host.data(), amax, host.size() * sizeof(float), hipMemcpyDeviceToHost, s));
        SKY_HIP(hipStreamSynchronize(s));
        {   // an fp8 tensor some launch writes and whose range was never seen would silently get scale 1 (saturation at 448 or lost range)
            std::vector<char> written(tw.bufs.size(), 0);
            for (const Op& op : tw.ops)
                if (op.out.buf >= 0 && op.out.B > 0) written[op.out.buf] = 1;
            bool all_zero = true;
            for (size_t i = 0; i < host.size(); ++i) all_zero = all_zero && !(host[i] > 0.0f);
            for (size_t i = 0; i < e.bufs.size() && !all_zero; ++i)
                if (e.bufs[i].dt == SKY_FP8 && written[i] && !(host[i] > 0.0f))
                    throw Error(SKY_ERR_STATE, "sky_calibrate: workspace tensor " + std::to_string(i) + " is written by the graph but its range was not measured "
                                               "(all zeros on the calibration input, or a launch that skipped its amax reduction)");
        }
        for (size_t i = 0; i < e.bufs.size(); ++i) e.bufs[i].scale = (e.bufs[i].dt == SKY_FP8 && host[i] > 0.0f) ? host[i] / 448.0f : 1.0f;
        for (size_t i = 0; i < e.bufs.size(); ++i) {        // tied buffers: the root carries the larger range
            int r = (int)i;
            while (e.bufs[r].tie >= 0) r = e.bufs[r].tie;
            if (r != (int)i) e.bufs[r].scale = std::max(e.bufs[r].scale, e.bufs[i].scale);
        }
    } catch (...) {
        cleanup();
        throw;
    }
    cleanup();
    apply_scales(e);
    e.calibrated = true;
}

}  // namespace sky

using namespace sky;

struct sky_handle {
    Engine e;
    Geometry geom;
};

template <typename F>
static int guarded(sky_handle* h, F&& f)
{
    try {
        f();
        return SKY_OK;
    } catch (const Error& er) {
        if (h) h->e.err = er.what(); else g_create_error = er.what();
        return er.code;
    } catch (const std::exception& ex) {
        if (h) h->e.err = ex.what(); else g_create_error = ex.what();
        return SKY_ERR_INVALID;
    }
}

extern "C" {

int sky_abi_version(void) { return SKY_ABI_VERSION; }

const char* sky_build_info(void) { return SKY_SOURCE_HASH; }

int sky_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* sky_last_error(const sky_handle* h) { return h ? h->e.err.c_str() : g_create_error.c_str(); }

int sky_create(const sky_config* cfg, sky_handle** out)
{
    if (!cfg || !out) { g_create_error = "sky_create: null argument"; return SKY_ERR_INVALID; }
    if (cfg->struct_size != sizeof(sky_config)) { g_create_error = "sky_create: sky_config size mismatch (ABI)"; return SKY_ERR_INVALID; }
    sky_handle* h = nullptr;
    int rc = guarded(nullptr, [&] {
        h = new sky_handle();
        h->e.cfg = *cfg;
        h->e.dtype = cfg->dtype;
        if (cfg->dtype != SKY_F32 && cfg->dtype != SKY_BF16 && cfg->dtype != SKY_FP8) throw Error(SKY_ERR_INVALID, "unknown dtype");
        sky_config& c = h->e.cfg;
        if (c.base_channels <= 0) c.base_channels = 64;
        if (c.depth_multiple <= 0) c.depth_multiple = 1.0f;
        if (c.width_multiple <= 0) c.width_multiple = 1.0f;
        if (c.in_channels <= 0) c.in_channels = 3;
        if (c.num_levels <= 0) c.num_levels = 3;
        if (c.num_anchors <= 0) {
            c.num_anchors = 3;
            memcpy(c.anchors, kDefaultAnchors, sizeof(kDefaultAnchors));
        }
        if (c.num_levels > SKY_MAX_LEVELS || c.num_anchors > SKY_MAX_ANCHORS) throw Error(SKY_ERR_INVALID, "too many levels / anchors");
        if (c.expansion <= 0) c.expansion = 0.5f;
        collect_spec(h->e);
    });
    if (rc != SKY_OK) { delete h; return rc; }
    *out = h;
    return SKY_OK;
}

void sky_destroy(sky_handle* h) { delete h; }

int sky_num_params(const sky_handle* h) { return h ? (int)h->e.spec.size() : 0; }

int sky_param_info(const sky_handle* h, int index, const char** name, int32_t* ndim, int64_t shape[4])
{
    if (!h || index < 0 || index >= (int)h->e.spec.size()) return SKY_ERR_INVALID;
    const ParamSpec& p = h->e.spec[index];
    if (name) *name = p.name.c_str();
    if (ndim) *ndim = (int32_t)p.shape.size();
    if (shape) for (size_t i = 0; i < 4; ++i) shape[i] = i < p.shape.size() ? p.shape[i] : 0;
    return SKY_OK;
}

int sky_load_weights(sky_handle* h, const sky_tensor_desc* descs, int n)
{
    if (!h) return SKY_ERR_INVALID;
    return guarded(h, [&] {
        for (int i = 0; i < n; ++i) {
            const sky_tensor_desc& d = descs[i];
            if (!d.name || !d.data) throw Error(SKY_ERR_INVALID, "sky_load_weights: null name/data");
            HostTensor t;
            for (int k = 0; k < d.ndim; ++k) t.shape.push_back(d.shape[k]);
            t.data.assign((const float*)d.data, (const float*)d.data + t.numel());
            h->e.weights[d.name] = std::move(t);
        }
        h->e.weights_dirty = true;
    });
}

int sky_plan(sky_handle* h, int n_inputs, const sky_buffer* in)
{
    if (!h) return SKY_ERR_INVALID;
    return guarded(h, [&] {
        if (n_inputs < 1 || n_inputs > SKY_MAX_IO) throw Error(SKY_ERR_INVALID, "sky_plan: bad input count");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) throw Error(SKY_ERR_NO_DEVICE, "no HIP device visible: the HIP engine cannot run (there is no CPU fallback)");
        DeviceGuard dg(h->e.cfg.device);
        const Geometry g = geometry_of(n_inputs, in);
        plan(h->e, g);
        h->geom = g;
    });
}

int sky_num_outputs(const sky_handle* h) { return h ? (int)h->e.out_info.size() : 0; }

int sky_output_info(const sky_handle* h, int index, int32_t* ndim, int64_t shape[5])
{
    if (!h || index < 0 || index >= (int)h->e.out_info.size()) return SKY_ERR_INVALID;
    const IoInfo& io = h->e.out_info[index];
    if (ndim) *ndim = io.ndim;
    if (shape) for (int i = 0; i < 5; ++i) shape[i] = io.shape[i];
    return SKY_OK;
}

static void check_forward(sky_handle* h, int n_inputs, const sky_buffer* inputs, int n_outputs, const sky_buffer* outputs)
{
    if (!h->e.planned) throw Error(SKY_ERR_STATE, "sky_forward before sky_plan");
    if (h->e.weights_dirty) throw Error(SKY_ERR_STATE, "weights changed after sky_plan: call sky_plan again");
    if (!h->e.calibrated) throw Error(SKY_ERR_STATE, "fp8 engine without activation scales: call sky_calibrate (or sky_scales_write) after sky_plan");
    if (n_inputs != h->geom.n) throw Error(SKY_ERR_SHAPE, "sky_forward: input count differs from the plan");
    if (n_outputs != (int)h->e.out_info.size()) throw Error(SKY_ERR_SHAPE, "sky_forward: expected " + std::to_string(h->e.out_info.size()) + " outputs");
    for (int i = 0; i < n_inputs; ++i) {
        if (!inputs[i].data) throw Error(SKY_ERR_INVALID, "sky_forward: null input");
        for (int k = 0; k < h->geom.ndim[i]; ++k)
            if (inputs[i].shape[k] != h->geom.shape[i][k]) throw Error(SKY_ERR_SHAPE, "sky_forward: input shape differs from the plan");
    }
    // raw detection levels are optional (NULL = not wanted: the head epilogue then skips their stores); everything else is required
    std::vector<char> optional(n_outputs, 0);
    for (const Op& op : h->e.ops)
        if (op.kind == OP_CONV && op.head && op.raw_ext >= 16 && op.raw_ext - 16 < n_outputs) optional[op.raw_ext - 16] = 1;
    for (int i = 0; i < n_outputs; ++i) {
        if (!outputs[i].data && !optional[i]) throw Error(SKY_ERR_INVALID, "sky_forward: null output");
        if (!outputs[i].data) continue;
        // a caller-side cache that hands over the buffers of another geometry must fail here, not write past their end
        const IoInfo& oi = h->e.out_info[i];
        bool same = outputs[i].ndim == oi.ndim;
        for (int k = 0; same && k < oi.ndim; ++k) same = outputs[i].shape[k] == oi.shape[k];
        if (!same) throw Error(SKY_ERR_SHAPE, "sky_forward: output " + std::to_string(i) + " has another shape than the plan's (sky_output_info)");
    }
}

int sky_forward(sky_handle* h, int n_inputs, const sky_buffer* inputs, int n_outputs, const sky_buffer* outputs, void* stream)
{
    if (!h) return SKY_ERR_INVALID;
    return guarded(h, [&] {
        check_forward(h, n_inputs, inputs, n_outputs, outputs);
        DeviceGuard dg(h->e.cfg.device);
        run(h->e, inputs, n_inputs, outputs, n_outputs, (hipStream_t)stream);
    });
}

int sky_time_forward(sky_handle* h, int n_inputs, const sky_buffer* inputs, int n_outputs, const sky_buffer* outputs, void* stream,
                     int iters, float* ms_per_iter)
{
    if (!h) return SKY_ERR_INVALID;
    return guarded(h, [&] {
        check_forward(h, n_inputs, inputs, n_outputs, outputs);
        DeviceGuard dg(h->e.cfg.device);
        hipStream_t s = (hipStream_t)stream;
        hipEvent_t e0, e1;
        SKY_HIP(hipEventCreate(&e0));
        SKY_HIP(hipEventCreate(&e1));
        SKY_HIP(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) run(h->e, inputs, n_inputs, outputs, n_outputs, s);
        SKY_HIP(hipEventRecord(e1, s));
        SKY_HIP(hipEventSynchronize(e1));
        float ms = 0;
        SKY_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (ms_per_iter) *ms_per_iter = ms / (float)std::max(iters, 1);
    });
}

int sky_profile_forward(sky_handle* h, int n_inputs, const sky_buffer* inputs, int n_outputs, const sky_buffer* outputs, void* stream,
                        int iters, int max_ops, float* ms_per_op, double* flops_per_op, int32_t* tag_per_op, int32_t* n_ops)
{
    if (!h) return SKY_ERR_INVALID;
    return guarded(h, [&] {
        check_forward(h, n_inputs, inputs, n_outputs, outputs);
        DeviceGuard dg(h->e.cfg.device);
        const int n = (int)h->e.ops.size();
        if (n_ops) *n_ops = n;
        if (n > max_ops) throw Error(SKY_ERR_INVALID, "sky_profile_forward: max_ops too small");
        hipStream_t s = (hipStream_t)stream;
        std::vector<hipEvent_t> ev(n + 1);
        for (auto& e : ev) SKY_HIP(hipEventCreate(&e));
        std::vector<double> acc(n, 0.0);
        for (int it = 0; it < iters; ++it) {
            run(h->e, inputs, n_inputs, outputs, n_outputs, s, ev.data());
            SKY_HIP(hipEventSynchronize(ev[n]));
            for (int i = 0; i < n; ++i) {
                float ms = 0;
                SKY_HIP(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
                acc[i] += ms;
            }
        }
        for (auto& e : ev) (void)hipEventDestroy(e);
        for (int i = 0; i < n; ++i) {
            const Op& op = h->e.ops[i];
            if (ms_per_op) ms_per_op[i] = (float)(acc[i] / std::max(iters, 1));
            if (flops_per_op) flops_per_op[i] = op.flops;
            if (tag_per_op) tag_per_op[i] = (int)op.kind * 10000 + (op.kind == OP_CONV || op.kind == OP_IMPORT ? op.variant : 0);      // (import: 9100 = skipped, the stem reads the frames)
        }
    });
}

int sky_op_info(const sky_handle* h, int index, char* text, int text_len)
{
    if (!h || !text || index < 0 || index >= (int)h->e.ops.size()) return SKY_ERR_INVALID;
    static const char* names[] = {"import", "export", "conv", "maxpool5", "upsample", "ca_reduce", "ca_mlp", "sa_stats", "sa_gate", "scale", "decode", "layernorm", "attention", "cla", "export_tokens"};
    const Op& op = h->e.ops[index];
    if (op.kind == OP_CONV)
        snprintf(text, text_len, "conv %dx%d s%d %d->%d in %dx%dx%d out %dx%d ld %d/%d%s%s%s %s%d", op.ks, op.ks, op.stride, op.cin, op.cout,
                 op.in.B, op.in.H, op.in.W, op.Ho, op.Wo, op.in.ld, op.out.ld, op.res.valid() ? (op.in2_cin ? " +res in2" : " +res") : (op.in2_cin ? " in2" : ""), op.up2 ? " up2" : "",
                 op.head ? " head" : "", op.variant >= 9000 ? "fused-into-previous" : op.variant >= 8500 ? "csp-stage-fused-" : op.variant >= 8000 ? "stem+stride2-fused-" : op.variant == 1628 ? "cv3+head/" : op.variant == 3256 ? "gemm1x1/" : op.variant == 7128 ? "bneck" : op.variant == 7256 ? "bneck128x2/" : op.variant == 7257 ? "bneck128x2-fp8/" : op.variant == 7065 ? "bneck64x3/" : op.variant == 7066 ? "bneck64x4-fp8/" : op.variant >= 7000 ? "halo-cv1+3x3-" : op.variant >= 6000 ? "halo-s2-" : op.variant >= 5000 ? "halo-narrow" : op.variant >= 4600 ? "deep3x3/" : op.variant >= 4000 ? "halo" : op.variant >= 3000 ? "ring" : op.variant >= 2000 ? "stream" : op.variant >= 1500 ? "head-stream" : "tile", op.variant % (op.variant >= 1500 && op.variant < 2000 ? 500 : 1000));
    else
        snprintf(text, text_len, "%s in %dx%dx%dx%d out C%d", names[op.kind], op.in.B, op.in.H, op.in.W, op.in.C, op.out.C);
    if (h->e.opts) {   // developer switches this plan was made under (PlanOpt bits, sky_kernels.h)
        const size_t n = strlen(text);
        if ((int)n + 16 < text_len) snprintf(text + n, text_len - n, " opts=0x%x", h->e.opts);
    }
    return SKY_OK;
}

int sky_op_bytes(const sky_handle* h, int index, double* bytes)
{
    if (!h || !bytes || index < 0 || index >= (int)h->e.ops.size()) return SKY_ERR_INVALID;
    const Op& op = h->e.ops[index];
    const int es = h->e.esize();
    auto sz = [&](const TV& t, int esz) { return (t.buf >= 0 || t.ext >= 0) && t.B ? (double)t.B * t.H * t.W * t.C * (t.buf >= 0 ? dtype_size(t.dt) : esz) : 0.0; };
    double b = sz(op.in, es) + sz(op.out, es) + sz(op.res, es) + sz(op.in2, es);
    if (op.kind == OP_CONV && op.head) b += 2.0 * op.in.B * op.Ho * op.Wo * (double)op.cout * 4;   // raw + decoded, fp32
    if (op.kind == OP_IMPORT) b += (double)op.out.B * op.src_c * op.src_h * op.src_w;                // uint8 frames (4x for fp32 input)
    *bytes = b;
    return SKY_OK;
}

int sky_op_io_bytes(const sky_handle* h, int index, int with_raw, double* read_bytes, double* written_bytes)
{
    if (!h || !read_bytes || !written_bytes || index < 0 || index >= (int)h->e.ops.size()) return SKY_ERR_INVALID;
    const Op& op = h->e.ops[index];
    const int es = h->e.esize();
    auto sz = [&](const TV& t, int esz) { return (t.buf >= 0 || t.ext >= 0) && t.B ? (double)t.B * t.H * t.W * t.C * (t.buf >= 0 ? dtype_size(t.dt) : esz) : 0.0; };
    double r = sz(op.in, es) + sz(op.res, es) + sz(op.in2, es), w = sz(op.out, es);
    if (op.kind == OP_CONV && op.head) w += (with_raw ? 2.0 : 1.0) * op.in.B * op.Ho * op.Wo * (double)op.cout * 4;   // decoded rows (+ raw level), fp32
    if (op.kind == OP_IMPORT) r = (double)op.out.B * op.src_c * op.src_h * op.src_w;                 // the caller's frames as uint8 (4x for fp32 input)
    *read_bytes = r;
    *written_bytes = w;
    return SKY_OK;
}

int sky_plan_stats(const sky_handle* h, double* flops, double* activation_bytes, double* weight_bytes, int32_t* launches)
{
    if (!h || !h->e.planned) return SKY_ERR_STATE;
    if (flops) *flops = h->e.flops;
    if (activation_bytes) *activation_bytes = h->e.act_bytes;
    if (weight_bytes) *weight_bytes = h->e.weight_bytes;
    if (launches) *launches = (int32_t)h->e.ops.size();
    return SKY_OK;
}

int sky_num_packed(const sky_handle* h)
{
    if (!h || !h->e.planned) return SKY_ERR_STATE;
    return (int)h->e.convs.size();
}

int sky_packed_info(const sky_handle* h, int i, sky_packed_desc* out)
{
    if (!h || !out) return SKY_ERR_INVALID;
    if (!h->e.planned) return SKY_ERR_STATE;
    if (i < 0 || i >= (int)h->e.convs.size()) return SKY_ERR_INVALID;
    const DevConv& d = h->e.convs[i];
    memset(out, 0, sizeof(*out));
    snprintf(out->name, sizeof(out->name), "%s", d.name.c_str());
    out->rows = d.rows; out->cout = d.cout; out->kpad = d.Kpad; out->kernel_size = d.ks; out->cin = d.cin;
    out->dtype = d.cdt;
    return SKY_OK;
}

int sky_packed_read(sky_handle* h, int i, void* weights_host, size_t weight_bytes, float* bias_host, size_t bias_count)
{
    if (!h) return SKY_ERR_INVALID;
    return guarded(h, [&] {
        if (!h->e.planned) throw Error(SKY_ERR_STATE, "sky_packed_read: plan the graph first (weights are packed at plan time)");
        DeviceGuard dg(h->e.cfg.device);
        if (i < 0 || i >= (int)h->e.convs.size()) throw Error(SKY_ERR_INVALID, "sky_packed_read: index out of range");
        const DevConv& d = h->e.convs[i];
        if (weights_host) {
            if (weight_bytes < d.bytes) throw Error(SKY_ERR_INVALID, "sky_packed_read: weight buffer too small");
            SKY_HIP(hipMemcpy(weights_host, d.w, d.bytes, hipMemcpyDeviceToHost));
        }
        if (bias_host) {
            if (bias_count < (size_t)d.rows) throw Error(SKY_ERR_INVALID, "sky_packed_read: bias buffer too small");
            SKY_HIP(hipMemcpy(bias_host, d.bias, (size_t)d.rows * sizeof(float), hipMemcpyDeviceToHost));
        }
    });
}

int sky_packed_scales(sky_handle* h, int i, float* scales_host, size_t count)
{
    if (!h) return SKY_ERR_INVALID;
    return guarded(h, [&] {
        if (!h->e.planned) throw Error(SKY_ERR_STATE, "sky_packed_scales: plan the graph first");
        if (i < 0 || i >= (int)h->e.convs.size() || !scales_host) throw Error(SKY_ERR_INVALID, "sky_packed_scales: bad argument");
        const DevConv& d = h->e.convs[i];
        if (count < (size_t)d.rows) throw Error(SKY_ERR_INVALID, "sky_packed_scales: buffer too small");
        // real weight = stored * scale: fp8 rows carry their own scale, exp2-domain rows of the bf16 engine 1 / log2 e (pack_conv)
        for (int r = 0; r < d.rows; ++r) scales_host[r] = d.w_scale.empty() ? (float)(1.0 / (double)d.pre) : d.w_scale[r];
    });
}

int sky_calibrate(sky_handle* h, int n_inputs, const sky_buffer* inputs, void* stream)
{
    if (!h) return SKY_ERR_INVALID;
    return guarded(h, [&] {
        if (n_inputs != h->geom.n) throw Error(SKY_ERR_SHAPE, "sky_calibrate: input count differs from the plan");
        for (int i = 0; i < n_inputs; ++i)
            if (!inputs[i].data) throw Error(SKY_ERR_INVALID, "sky_calibrate: null input");
        DeviceGuard dg(h->e.cfg.device);
        calibrate(h->e, n_inputs, inputs, (hipStream_t)stream);
    });
}

int sky_num_scales(const sky_handle* h) { return h && h->e.planned ? (int)h->e.bufs.size() : 0; }

int sky_scales_read(sky_handle* h, float* scales_host, int n)
{
    if (!h) return SKY_ERR_INVALID;
    return guarded(h, [&] {
        if (!h->e.planned || !scales_host || n < (int)h->e.bufs.size()) throw Error(SKY_ERR_INVALID, "sky_scales_read: plan first / buffer too small");
        for (size_t i = 0; i < h->e.bufs.size(); ++i) {
            TV t; t.buf = (int)i; t.dt = h->e.bufs[i].dt;
            scales_host[i] = tv_scale(h->e, t);
        }
    });
}

int sky_scales_write(sky_handle* h, const float* scales_host, int n)
{
    if (!h) return SKY_ERR_INVALID;
    return guarded(h, [&] {
        if (!h->e.planned || !scales_host || n != (int)h->e.bufs.size()) throw Error(SKY_ERR_INVALID, "sky_scales_write: plan first; one scale per workspace buffer (sky_num_scales)");
        if (h->e.dtype != SKY_FP8) throw Error(SKY_ERR_STATE, "sky_scales_write: only the fp8 engine has activation scales");
        for (int i = 0; i < n; ++i)
            if (!(scales_host[i] > 0.0f)) throw Error(SKY_ERR_INVALID, "sky_scales_write: scales must be positive");
        DeviceGuard dg(h->e.cfg.device);
        for (int i = 0; i < n; ++i) {
            int r = i;
            while (h->e.bufs[r].tie >= 0) r = h->e.bufs[r].tie;
            if (h->e.bufs[i].dt == SKY_FP8) h->e.bufs[r].scale = scales_host[i];
        }
        apply_scales(h->e);
        h->e.calibrated = true;
    });
}

int sky_nms(sky_handle* h, const float* det, int B, int N, int nc, const sky_nms_params* p, float* out, int32_t* counts, void* stream)
{
    if (!h) return SKY_ERR_INVALID;
    return guarded(h, [&] {
        if (!det || !p || !out || !counts) throw Error(SKY_ERR_INVALID, "sky_nms: null argument");
        // the round-3 struct ends in front of out_image_stride: a caller built against it gets dense outputs
        const bool has_strides = p->struct_size == sizeof(sky_nms_params);
        if (!has_strides && p->struct_size != offsetof(sky_nms_params, out_image_stride))
            throw Error(SKY_ERR_INVALID, "sky_nms: sky_nms_params size mismatch (ABI)");
        if (B < 1 || N < 1 || nc < 1) throw Error(SKY_ERR_SHAPE, "sky_nms: bad geometry");
        if (p->max_detections < 1 || p->max_detections > 4096) throw Error(SKY_ERR_INVALID, "sky_nms: max_detections must be in [1, 4096]");
        if (p->n_classes > 64) throw Error(SKY_ERR_INVALID, "sky_nms: at most 64 class filters");
        DeviceGuard dg(h->e.cfg.device);
        NmsArgs a;
        memset(&a, 0, sizeof(a));
        a.device = h->e.cfg.device;
        a.det = det; a.B = B; a.N = N; a.nc = nc;
        a.conf = p->conf_threshold; a.iou = p->iou_threshold; a.max_wh = p->max_wh;
        a.agnostic = p->agnostic; a.max_det = p->max_detections; a.max_nms = p->max_nms; a.mode = p->mode;
        if (p->mode < 0 || p->mode > 2) throw Error(SKY_ERR_INVALID, "sky_nms: mode is 0 (literal), 1 (corrected) or 2 (box rows)");
        a.multi_label = p->multi_label && nc > 1 && p->mode != 2;       // metrics.py:396
        a.n_classes = p->n_classes;
        for (int i = 0; i < p->n_classes; ++i) a.classes[i] = p->classes[i];
        a.out = out; a.counts = counts;
        a.out_stride = has_strides && p->out_image_stride ? p->out_image_stride : (long)p->max_detections * 7;
        a.counts_stride = has_strides && p->counts_stride ? p->counts_stride : 1;
        if (a.out_stride < (long)p->max_detections * 7 || a.counts_stride < 1)
            throw Error(SKY_ERR_INVALID, "sky_nms: out_image_stride below max_detections * 7 or counts_stride below 1");
        long cap = 0;
        const size_t need = nms_workspace_bytes(B, N, nc, a.multi_label, &cap);
        if (need > h->e.nms_ws_bytes) {
            // A smaller workspace is retired, not freed: a captured hipGraph may still replay launches that point into it
            // (it goes with the handle, like every other device allocation of the plan).
            if (h->e.nms_ws) h->e.owned.push_back(h->e.nms_ws);
            h->e.nms_ws = nullptr;
            SKY_HIP(hipMalloc(&h->e.nms_ws, need));
            h->e.nms_ws_bytes = need;
        }
        a.cap = cap;
        char* w = (char*)h->e.nms_ws;
        const int nblk = (N + 255) / 256;
        a.blk_counts = (int*)w; w += ((size_t)B * nblk * sizeof(int) + 255) / 256 * 256;
        a.totals = (int*)w; w += ((size_t)B * sizeof(int) + 255) / 256 * 256;
        a.keys = (unsigned long long*)w; w += (size_t)B * cap * sizeof(unsigned long long);
        a.keys2 = (unsigned long long*)w; w += (size_t)B * cap * sizeof(unsigned long long);
        a.cand = (float*)w;
        SKY_HIP(launch_nms(a, (hipStream_t)stream));
    });
}

int sky_box_iou(sky_handle* h, const float* box1, int n, int box1_is_4xn, const float* box2, int m, float* out, void* stream)
{
    if (!h) return SKY_ERR_INVALID;
    return guarded(h, [&] {
        if (n < 0 || m < 0) throw Error(SKY_ERR_SHAPE, "sky_box_iou: negative box count");
        if ((n && !box1) || (m && !box2) || ((long)n * m && !out)) throw Error(SKY_ERR_INVALID, "sky_box_iou: null argument");
        DeviceGuard dg(h->e.cfg.device);
        SKY_HIP(launch_box_iou(box1, n, box1_is_4xn ? 1 : 0, box2, m, out, (hipStream_t)stream));
    });
}

int sky_letterbox(sky_handle* h, const uint8_t* src, int H0, int W0, uint8_t* dst, int H1, int W1, int new_h, int new_w, int top, int left,
                  int pad_value, int dst_chw, int reverse_channels, void* stream)
{
    if (!h) return SKY_ERR_INVALID;
    return guarded(h, [&] {
        if (!src || !dst) throw Error(SKY_ERR_INVALID, "sky_letterbox: null argument");
        if (H0 < 1 || W0 < 1 || new_h < 1 || new_w < 1 || top < 0 || left < 0 || top + new_h > H1 || left + new_w > W1)
            throw Error(SKY_ERR_SHAPE, "sky_letterbox: the resized frame plus its border must fit the destination");
        DeviceGuard dg(h->e.cfg.device);
        SKY_HIP(launch_letterbox(src, H0, W0, dst, H1, W1, new_h, new_w, top, left, pad_value & 255, dst_chw ? 1 : 0, reverse_channels ? 1 : 0,
                                 (hipStream_t)stream));
    });
}

int sky_scale_img(sky_handle* h, const void* src, int src_dtype, int B, int C, int H, int W, float* dst, int out_h, int out_w, int pad_h, int pad_w,
                  int flip, float pad_value, void* stream)
{
    if (!h) return SKY_ERR_INVALID;
    return guarded(h, [&] {
        if (!src || !dst) throw Error(SKY_ERR_INVALID, "sky_scale_img: null argument");
        if (src_dtype != SKY_IO_F32 && src_dtype != SKY_IO_U8) throw Error(SKY_ERR_INVALID, "sky_scale_img: source must be fp32 or uint8");
        if (flip != 0 && flip != 2 && flip != 3) throw Error(SKY_ERR_INVALID, "sky_scale_img: flip is 0, 2 (rows) or 3 (columns)");
        if (B < 1 || C < 1 || H < 1 || W < 1 || out_h < 1 || out_w < 1 || pad_h < out_h || pad_w < out_w)
            throw Error(SKY_ERR_SHAPE, "sky_scale_img: the resized image must be non-empty and fit the padded destination");
        DeviceGuard dg(h->e.cfg.device);
        SKY_HIP(launch_scale_img(src, src_dtype == SKY_IO_U8, B * C, H, W, dst, out_h, out_w, pad_h, pad_w, flip, pad_value, (hipStream_t)stream));
    });
}

int sky_map_detections(sky_handle* h, const float* src, int B, int N, int no, int row0, int rows, float scale, int flip, float img_h, float img_w,
                       const int32_t* origins, int tiles_per_image, float* dst, int64_t dst_rows, int64_t dst_row0, void* stream)
{
    if (!h) return SKY_ERR_INVALID;
    return guarded(h, [&] {
        if (!src || !dst) throw Error(SKY_ERR_INVALID, "sky_map_detections: null argument");
        if (flip != 0 && flip != 2 && flip != 3) throw Error(SKY_ERR_INVALID, "sky_map_detections: flip is 0, 2 (rows) or 3 (columns)");
        if (!(scale > 0.0f)) throw Error(SKY_ERR_INVALID, "sky_map_detections: scale must be positive");
        if (B < 1 || N < 1 || no < 5 || row0 < 0 || rows < 0 || row0 + rows > N || tiles_per_image < 1 || B % tiles_per_image != 0 || dst_row0 < 0 ||
            dst_row0 + (int64_t)tiles_per_image * rows > dst_rows)
            throw Error(SKY_ERR_SHAPE, "sky_map_detections: source rows [row0, row0 + rows) of every tile must fit the destination image");
        if (rows == 0) return;
        DeviceGuard dg(h->e.cfg.device);
        SKY_HIP(launch_map_detections(src, B, N, no, row0, rows, scale, flip, img_h, img_w, origins, tiles_per_image, dst, (long)dst_rows,
                                      (long)dst_row0, (hipStream_t)stream));
    });
}

int sky_offset_boxes(sky_handle* h, float* rows, const int32_t* counts, int T, int R, int cols, const int32_t* origins, void* stream)
{
    if (!h) return SKY_ERR_INVALID;
    return guarded(h, [&] {
        if (!rows || !counts || !origins) throw Error(SKY_ERR_INVALID, "sky_offset_boxes: null argument");
        if (T < 0 || R < 1 || cols < 4) throw Error(SKY_ERR_SHAPE, "sky_offset_boxes: rows must be [T, R, cols >= 4]");
        DeviceGuard dg(h->e.cfg.device);
        SKY_HIP(launch_offset_boxes(rows, counts, T, R, cols, origins, (hipStream_t)stream));
    });
}

int sky_tile_gather(sky_handle* h, const uint8_t* src, int H0, int W0, int src_chw, const int32_t* origins, int n, uint8_t* dst, int tile_h,
                    int tile_w, int pad_value, int reverse_channels, void* stream)
{
    if (!h) return SKY_ERR_INVALID;
    return guarded(h, [&] {
        if (!src || !dst || !origins) throw Error(SKY_ERR_INVALID, "sky_tile_gather: null argument");
        if (H0 < 1 || W0 < 1 || n < 1 || tile_h < 1 || tile_w < 1) throw Error(SKY_ERR_SHAPE, "sky_tile_gather: empty frame or tile");
        DeviceGuard dg(h->e.cfg.device);
        SKY_HIP(launch_tile_gather(src, H0, W0, src_chw ? 1 : 0, origins, n, dst, tile_h, tile_w, pad_value & 255, reverse_channels ? 1 : 0,
                                   (hipStream_t)stream));
    });
}

int sky_nms_fetch(sky_handle* h, const int32_t* counts_dev, int B, int32_t* counts_host, void* stream)
{
    if (!h) return SKY_ERR_INVALID;
    return guarded(h, [&] {
        DeviceGuard dg(h->e.cfg.device);
        SKY_HIP(hipMemcpyAsync(counts_host, counts_dev, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
        SKY_HIP(hipStreamSynchronize((hipStream_t)stream));
    });
}

}  // extern "C"
