/*
 * skyeye_hip.h -- C ABI of libskyeye_hip.so, the MI355X (gfx950) inference engine for the
 * SkyEye detection forward pass.
 *
 * The reference (UmaimaKhan01/SkyEye-Aerial-Object-Detection-using-Yolo) has no FFI: its boundary is the
 * Python class API of skyeye/core/models/detector.py.  Each entry point below therefore cites the Python
 * interface it stands in for; the ctypes shim in
 * skyeye-aerial-object-detection-using-yolo_amd/skyeye/_native.py binds exactly these symbols and
 * presents the reference's classes on top of them.
 *
 * Conventions
 *   - every function returns 0 on success or a negative sky_status; nothing throws across the ABI;
 *     sky_last_error(h) gives the text for the last failure on that handle (sky_last_error(NULL) for
 *     failures of sky_create itself)
 *   - the caller owns every input / output buffer; the engine owns weights and its workspace arena
 *   - device pointers are plain HIP device pointers (e.g. torch.Tensor.data_ptr()); `stream` is a
 *     hipStream_t passed as void* (NULL = the null stream)
 *   - one handle per (device, stream); a handle is not thread-safe, different handles are independent
 *   - all kernels are launched asynchronously on `stream`; host synchronisation happens only inside
 *     sky_nms_fetch (it returns counts to the host), sky_packed_read (weights to the host) and the timing hooks
 */
#ifndef SKYEYE_HIP_H
#define SKYEYE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKY_ABI_VERSION 1
#define SKY_MAX_LEVELS 4
#define SKY_MAX_ANCHORS 8
#define SKY_MAX_IO 4

typedef enum {
    SKY_OK = 0,
    SKY_ERR_INVALID = -1,       /* bad argument / unsupported configuration */
    SKY_ERR_MISSING_WEIGHT = -2,
    SKY_ERR_SHAPE = -3,
    SKY_ERR_HIP = -4,           /* a HIP runtime call failed */
    SKY_ERR_STATE = -5,         /* call order violated (forward before plan, ...) */
    SKY_ERR_NO_DEVICE = -6
} sky_status;

/* arithmetic type of the activations / MFMA operands (accumulation is always fp32) */
typedef enum {
    SKY_F32 = 0,   /* exact mode: v_mfma_f32_16x16x4_f32, used for the 1e-4 parity gate */
    SKY_BF16 = 1,  /* production mode: v_mfma_f32_16x16x32_bf16 */
    SKY_FP8 = 2    /* OCP e4m3fn weights (one scale per output channel) and activations (one calibrated scale per tensor, see
                    * sky_calibrate): v_mfma_f32_16x16x32_fp8_fp8 / v_mfma_scale_f32_16x16x128_f8f6f4; the stem (pixel values) stays
                    * bf16, accumulation, bias, SiLU, decode fp32.  Covers the convolutional graph (detector without attention heads).
                    * The reference's only reduced-precision hook is model.half() (validate.py:195-197, detect.py:107-108). */
} sky_dtype;

/* element type / layout of caller buffers at the boundary */
typedef enum { SKY_IO_F32 = 0, SKY_IO_U8 = 1 } sky_io_dtype;
typedef enum { SKY_NCHW = 0, SKY_NHWC = 1 } sky_layout;

/* Which reference module a handle implements.  Values name the reference class (file:line). */
typedef enum {
    SKY_MOD_DETECTOR = 0,          /* SkyEyeDetector           detector.py:234-371 (with SURVEY D1/D2/D3)      */
    SKY_MOD_ENHANCED_DETECTOR = 1, /* EnhancedSkyEyeDetector   detector.py:436-501 (with D4)                   */
    SKY_MOD_CONV_BLOCK = 2,        /* ConvolutionBlock         blocks.py:10-41                                */
    SKY_MOD_BOTTLENECK = 3,        /* BottleneckBlock          blocks.py:69-90                                */
    SKY_MOD_CSP = 4,               /* CSPBlock                 blocks.py:93-123                               */
    SKY_MOD_SPP = 5,               /* SPPBlock                 blocks.py:126-149                              */
    SKY_MOD_FOCUS = 6,             /* FocusBlock               blocks.py:152-182                              */
    SKY_MOD_CHANNEL_ATTENTION = 7, /* ChannelAttention         attention.py:11-60                             */
    SKY_MOD_SPATIAL_ATTENTION = 8, /* SpatialAttention         attention.py:63-98                             */
    SKY_MOD_COMBINED_ATTENTION = 9,/* CombinedAttention        attention.py:101-130                           */
    SKY_MOD_BACKBONE = 10,         /* Backbone / CSPDarknet    backbone.py:12-116                             */
    SKY_MOD_NECK = 11,             /* FeatureNeck              detector.py:148-231                            */
    SKY_MOD_HEAD = 12,             /* DetectionHead (+process_detections) detector.py:18-145                  */
    SKY_MOD_CROSS_LAYER_ATTENTION = 13, /* CrossLayerAttention attention.py:133-241                           */
    SKY_MOD_TRANSFORMER_LAYER = 14,     /* TransformerLayer    attention.py:244-309                           */
    SKY_MOD_WINDOWED_ATTENTION = 15,    /* WindowedSelfAttention attention.py:312-399                         */
    SKY_MOD_DECODE = 16,           /* DetectionHead.process_detections alone  detector.py:88-145 (inputs: raw levels) */
    SKY_MOD_UTILITY = 100          /* no graph: a handle that only carries the NMS workspace (metrics.py:361-457)  */
} sky_module;

/*
 * Constructor arguments.  Field names follow the reference's keyword arguments / YAML keys
 * (detector.py:255-285,393-405: nc, base_channels, depth_multiple, width_multiple, anchors).
 * Unused fields for a given module are ignored.
 */
typedef struct {
    uint32_t struct_size;       /* sizeof(sky_config), for ABI checking */
    int32_t module;             /* sky_module */
    int32_t dtype;              /* sky_dtype */
    int32_t device;             /* HIP device ordinal */
    /* detector / backbone */
    int32_t base_channels;      /* default 64 */
    float depth_multiple;       /* default 1.0 */
    float width_multiple;       /* default 1.0 */
    int32_t nc;                 /* number of classes */
    int32_t in_channels;        /* image channels (3) */
    int32_t num_levels;         /* detection levels (3) */
    int32_t num_anchors;        /* anchors per level (3) */
    float anchors[SKY_MAX_LEVELS * SKY_MAX_ANCHORS * 2]; /* [level][anchor][w,h], pixels */
    /* generic block arguments */
    int32_t c_in, c_out;        /* in_channels / out_channels (dim for attention / transformer) */
    int32_t kernel_size, stride;
    int32_t activation;         /* ConvolutionBlock(activation=) */
    int32_t num_blocks;         /* CSPBlock(num_blocks=) */
    int32_t shortcut;           /* BottleneckBlock(shortcut=) */
    float expansion;            /* BottleneckBlock / CSPBlock expansion */
    int32_t heads;              /* attention heads */
    int32_t window_size;        /* WindowedSelfAttention */
    int32_t region_size;        /* CrossLayerAttention */
    int32_t reduction_ratio;    /* ChannelAttention (16) */
    int32_t key_channels;       /* CrossLayerAttention key/value input channels (D4 when != c_in) */
    int32_t level_channels[SKY_MAX_LEVELS]; /* FeatureNeck in_channels / DetectionHead channels */
    int32_t input_h, input_w;   /* DetectionHead.process_detections(outputs, input_shape) for a standalone head */
    int32_t reserved[6];        /* [0] != 0: DETECTOR / ENHANCED_DETECTOR run the build-defined "transformer prediction
                                 * heads" (SURVEY App. A, D5): WindowedSelfAttention(window 8, C/32 heads) on P3 and P4 and
                                 * TransformerLayer(8 heads) on P5 ahead of the detection convolutions; parameters
                                 * head_attention.p3.* / .p4.* / .p5.*; P4 must be a multiple of 8, i.e. H, W multiples of 128.
                                 * [1..5] must be 0. */
} sky_config;

/* One named host fp32 tensor of a state dict (names as in SURVEY Appendix C, relative to the module). */
typedef struct {
    const char* name;
    const void* data;           /* host pointer, fp32 (int64 entries such as num_batches_tracked are skipped) */
    int32_t ndim;
    int64_t shape[4];
} sky_tensor_desc;

/* A caller buffer handed to sky_forward. */
typedef struct {
    void* data;                 /* device pointer */
    int32_t dtype;              /* sky_io_dtype */
    int32_t layout;             /* sky_layout (ignored for rank-3 token tensors) */
    int32_t ndim;
    int64_t shape[5];
} sky_buffer;

/* Arguments of the reference's non_max_suppression (metrics.py:361-369) + the build's semantics switch. */
typedef struct {
    uint32_t struct_size;
    float conf_threshold;       /* 0.25 */
    float iou_threshold;        /* 0.45 */
    int32_t agnostic;
    int32_t multi_label;
    int32_t max_detections;     /* 300 */
    int32_t max_nms;            /* 30000, metrics.py:393 */
    float max_wh;               /* 4096, metrics.py:392 */
    int32_t mode;               /* 0 = literal (file as written, D7-D9), 1 = corrected (YOLOv5 semantics), 2 = box rows: `det` rows
                                 * are already boxes (x1, y1, x2, y2, conf, cls, ...; row stride nc + 5 floats) -- the cross-tile
                                 * stage of tiled inference, whose input is the survivors of the per-tile NMS */
    int32_t n_classes;          /* length of `classes`, 0 = no filter */
    int32_t classes[64];
    /* round 4 (struct_size tells the library whether the caller knows them; a shorter struct means 0, 0): output strides, so that
     * the rows and the count of an image can land directly in a caller's exchange buffer (skyeye/distributed.py: BoxExchange --
     * one block of max_detections * 7 floats + the count per image, sent by ONE all-gather without repacking). */
    int32_t out_image_stride;   /* floats between the row blocks of consecutive images; 0 = max_detections * 7 (dense) */
    int32_t counts_stride;      /* int32 elements between consecutive counts; 0 = 1 (dense) */
} sky_nms_params;

typedef struct sky_handle sky_handle;

/* Library / device probes (no compute). */
int sky_abi_version(void);
/* "<16 hex digits>": sha-256 prefix of the sources (csrc/ *.hip, *.h, engine.cpp, Makefile, this header) the loaded library was built
 * from; bench.py / smoke() print it and __graft_entry__.build() compares it with the working tree (round 2 measured a stale object). */
const char* sky_build_info(void);
int sky_device_count(void);
const char* sky_last_error(const sky_handle* h);

/* SkyEyeDetector.__init__ / construct_model (detector.py:239-298, :409-433) and the block constructors. */
int sky_create(const sky_config* cfg, sky_handle** out);
void sky_destroy(sky_handle* h);

/* Number / names / shapes of the state-dict entries this module expects (nn.Module.state_dict()). */
int sky_num_params(const sky_handle* h);
int sky_param_info(const sky_handle* h, int index, const char** name, int32_t* ndim, int64_t shape[4]);

/* load_state_dict / load_from_pretrained (detector.py:343-371): copies, folds BN, converts dtype+layout. */
int sky_load_weights(sky_handle* h, const sky_tensor_desc* descs, int n);

/* Fix the input geometry (the reference is shape-polymorphic; the engine plans a static graph per shape). */
int sky_plan(sky_handle* h, int n_inputs, const sky_buffer* input_shapes);

/* Output shapes of the planned graph, in forward() order. */
int sky_num_outputs(const sky_handle* h);
int sky_output_info(const sky_handle* h, int index, int32_t* ndim, int64_t shape[5]);

/* nn.Module.forward (detector.py:300-324 for the detector: outputs = [detections, raw_P3, raw_P4, raw_P5]). */
int sky_forward(sky_handle* h, int n_inputs, const sky_buffer* inputs, int n_outputs, const sky_buffer* outputs,
                void* stream);

/* fp8 engine (SKY_FP8) only: activation scales.  After sky_plan, run the planned graph once in bf16 on representative inputs (any
 * batch size, same C, H, W) and give every fp8 workspace tensor the scale max|x| / 448.  sky_forward on an fp8 plan fails with
 * SKY_ERR_STATE until this (or sky_scales_write) has been called.  Synchronises `stream`.  The scales belong to the plan:
 * sky_num_scales / sky_scales_read / sky_scales_write save and restore them (one float per workspace buffer, 1 for non-fp8 ones),
 * sky_packed_scales returns the per-output-channel scales of packed convolution i: real weight = stored weight * scale.
 *   fp8:  the weight scale of the row (max |w| / 448); the bias rows are stored unscaled.
 *   bf16: ln 2 for the rows of a SiLU convolution -- the bf16 engine keeps those layers in the exp2 domain (weights AND bias are
 *         stored times log2 e, the activation is v' * rcp(fma(exp2(-v'), log2 e, log2 e)), csrc/conv_frag.h) so for them the real
 *         bias = stored bias * scale too; 1 for layers without SiLU.
 *   fp32: 1. */
int sky_calibrate(sky_handle* h, int n_inputs, const sky_buffer* inputs, void* stream);
int sky_num_scales(const sky_handle* h);
int sky_scales_read(sky_handle* h, float* scales_host, int n);
int sky_scales_write(sky_handle* h, const float* scales_host, int n);
int sky_packed_scales(sky_handle* h, int i, float* scales_host, size_t count);

/*
 * non_max_suppression (metrics.py:361-457) on device.
 *   det   [B, N, nc+5] fp32 device
 *   out   [B, max_detections, 7] fp32 device (rows packed at stride 7; literal mode with nc>1 fills 7
 *         columns, every other case 6)
 *   counts[B] int32 device
 * sky_nms is asynchronous; sky_nms_fetch copies counts to the host (synchronises the stream).
 */
int sky_nms(sky_handle* h, const float* det, int B, int N, int nc, const sky_nms_params* p, float* out,
            int32_t* counts, void* stream);
int sky_nms_fetch(sky_handle* h, const int32_t* counts_dev, int B, int32_t* counts_host, void* stream);

/* Export of the engine's own weight file (the reference's export.py is empty and fuse_conv_and_bn,
 * general imports at utils/__init__.py:22-25, is undefined): the convolution weights exactly as the kernels read them --
 * BatchNorm folded (w * gamma / sqrt(var + 1e-5), bias = beta - mean * scale, blocks.py:39-41 fused_forward), K ordered
 * (ky, kx, cin) and padded to `kpad`, rows padded to the N tile, element type of the engine -- plus the fp32 bias rows; times the
 * per-row factor sky_packed_scales reports they are the folded fp32 values (fp8 rows, exp2-domain rows of the bf16 engine).
 * Valid after sky_plan.  sky_packed_read copies to host buffers (either may be NULL). */
typedef struct sky_packed_desc {
    char name[128];        /* state-dict name of the source weight (the first one for fused GEMMs such as cv1|cv2) */
    int32_t rows, cout;    /* packed rows (>= cout, zero rows beyond) */
    int32_t kpad;          /* elements per row */
    int32_t kernel_size, cin;
    int32_t dtype;         /* sky_dtype of the elements */
} sky_packed_desc;
int sky_num_packed(const sky_handle* h);
int sky_packed_info(const sky_handle* h, int i, sky_packed_desc* out);
int sky_packed_read(sky_handle* h, int i, void* weights_host, size_t weight_bytes, float* bias_host, size_t bias_count);

/* letterbox (core/data/augmentation.py:442-496): src uint8 [H0, W0, 3] on the device -> resize to (new_h, new_w) with
 * OpenCV's 8-bit INTER_LINEAR arithmetic -> placed at (top, left) of a (H1, W1) frame filled with pad_value (114).  dst is
 * [H1, W1, 3] (dst_chw = 0, the reference's return layout) or [3, H1, W1] (dst_chw = 1; reverse_channels = 1 also flips
 * BGR <-> RGB, detect.py:133), i.e. directly the engine's uint8 input.  The caller computes the geometry
 * (skyeye.core.data.augmentation.letterbox does, with the reference's rounding).  Asynchronous on `stream`. */
int sky_letterbox(sky_handle* h, const uint8_t* src, int H0, int W0, uint8_t* dst, int H1, int W1, int new_h, int new_w,
                  int top, int left, int pad_value, int dst_chw, int reverse_channels, void* stream);

/* Test-time augmentation and tiling front end.  The reference plumbs `augment=` through both CLIs
 * (validate.py:245, detect.py:140) but has no body for it; the schedule is YOLOv5's _forward_augment (scales 1 / 0.83 / 0.67, the
 * middle pass flipped left-right), built from the reference's own scale_img.
 *
 * sky_scale_img: scale_img (utils/torch_utils.py:262-288) of `src.flip(flip)`: src [B, C, H, W] fp32, or uint8 (then /255 first,
 * validate.py:236-238) -> bilinear (align_corners = False, ATen's arithmetic) to (out_h, out_w) = (int(H*ratio), int(W*ratio)) at
 * the top-left of dst [B, C, pad_h, pad_w] fp32, the rest filled with pad_value (0.447).  flip: 0, 2 (rows) or 3 (columns).
 * out == in copies (ratio 1.0).  The caller computes the geometry (skyeye.utils.torch_utils.scale_img does). */
int sky_scale_img(sky_handle* h, const void* src, int src_dtype, int B, int C, int H, int W, float* dst, int out_h, int out_w, int pad_h, int pad_w,
                  int flip, float pad_value, void* stream);

/* Map decoded rows back to the frame of the original image and place them in the concatenated result: rows [row0, row0 + rows) of
 * src [B, N, no] go to dst[b / tiles_per_image][dst_row0 + (b % tiles_per_image) * rows + r]; dst is [B / tiles_per_image, dst_rows, no].
 * Columns 0..3 (cx, cy, w, h): `/= scale`; flip 3: cx = img_w - cx; flip 2: cy = img_h - cy (YOLOv5 _descale_pred); then, if
 * origins != NULL (device int32 [B, 2] = (y, x) of each tile), cx += x, cy += y.  Other columns are copied. */
int sky_map_detections(sky_handle* h, const float* src, int B, int N, int no, int row0, int rows, float scale, int flip, float img_h, float img_w,
                       const int32_t* origins, int tiles_per_image, float* dst, int64_t dst_rows, int64_t dst_row0, void* stream);

/* Tiled inference across ranks ("Tiled (C5)"; build-defined, the reference has no tiling): the survivors of the per-tile
 * NMS -- rows [T, R, cols] fp32 on the device, corner boxes (x1, y1, x2, y2, ...) in tile pixels, counts[T] valid rows per tile -- are
 * moved in place into the frame of the whole image: x += origins[t][1], y += origins[t][0] (device int32 [T, 2] = (y, x)); rows past
 * counts[t] are left untouched.  The collective that follows (all-gather of these fixed-capacity blocks over RCCL) belongs to the host
 * layer, skyeye/distributed.py: this library exports no collective of its own (one process per GPU owns its communicator); the
 * owner rank then calls sky_nms with mode = 2 on the gathered rows. */
int sky_offset_boxes(sky_handle* h, float* rows, const int32_t* counts, int T, int R, int cols, const int32_t* origins, void* stream);

/* Large-frame tiling: src uint8 [H0, W0, 3] (src_chw = 0) or [3, H0, W0] (src_chw = 1) on the device -> dst uint8 [n, 3, tile_h, tile_w],
 * tile t = the window at origins[t] = (y, x) (device int32 [n, 2]); pixels outside the frame = pad_value (114);
 * reverse_channels = 1 flips BGR <-> RGB (detect.py:133).  dst is directly the engine's uint8 input. */
int sky_tile_gather(sky_handle* h, const uint8_t* src, int H0, int W0, int src_chw, const int32_t* origins, int n, uint8_t* dst, int tile_h,
                    int tile_w, int pad_value, int reverse_channels, void* stream);

/* box_iou (metrics.py:17-44), the pairwise IoU of the evaluation accounting (validate.py:71-108 process_batch):
 * out[n, m] fp32 on the device.  box1_is_4xn = 1 reads box1 as [4, n] -- the indexing the file actually performs
 *  --, 0 as [n, 4]; box2 is [m, 4], corners (x1, y1, x2, y2).  Asynchronous on `stream`. */
int sky_box_iou(sky_handle* h, const float* box1, int n, int box1_is_4xn, const float* box2, int m, float* out,
                void* stream);

/* Engine statistics for the bench harness: algorithmic FLOPs (2*MAC over conv/linear) and activation bytes
 * of the planned graph, number of launches per forward. */
int sky_plan_stats(const sky_handle* h, double* flops, double* activation_bytes, double* weight_bytes,
                   int32_t* launches);

/* Timing hooks used by bench.py: time `iters` back-to-back forwards with hipEvents on `stream`
 * (events recorded on the stream the kernels are launched on). Returns mean milliseconds per forward. */
int sky_time_forward(sky_handle* h, int n_inputs, const sky_buffer* inputs, int n_outputs,
                     const sky_buffer* outputs, void* stream, int iters, float* ms_per_iter);

/* Per-launch timing of the planned graph (bench.py's roofline leg): runs `iters` forwards with a hipEvent recorded
 * on `stream` after every launch and returns, per launch, the mean milliseconds, the algorithmic FLOPs, and a tag
 * (op kind * 10000 + convolution kernel variant: 1000+N tile = implicit-GEMM tile kernel, 2000+N_blk = streaming kernel
 * with LDS-resident weights, 3000+N_blk = streaming kernel with the weight ring; 0 for non-GEMM launches). */
int sky_profile_forward(sky_handle* h, int n_inputs, const sky_buffer* inputs, int n_outputs, const sky_buffer* outputs,
                        void* stream, int iters, int max_ops, float* ms_per_op, double* flops_per_op, int32_t* tag_per_op,
                        int32_t* n_ops);

/* Human-readable description of launch `index` of the planned graph ("conv 3x3 s1 128->128 @80x80 ...") for
 * profiling reports.  Returns SKY_ERR_INVALID past the last launch. */
int sky_op_info(const sky_handle* h, int index, char* text, int text_len);

/* Algorithmic HBM bytes of launch `index` (activation input + output + residual views, in the engine's dtype). */
int sky_op_bytes(const sky_handle* h, int index, double* bytes);

/* The same, split into what the op reads (input, second input, residual) and what it writes (output map; a detection level: its decoded rows
 * and -- `with_raw` != 0 -- its raw level).  A launch that computes a chain of planned ops (sky_op_info: "fused-into-previous") reads the
 * first op's inputs and writes the last op's output (+ the outputs of the detection levels in the chain): profiling reports add those up
 * (bench.py: roofline.families.algorithmic_bytes_per_launch). */
int sky_op_io_bytes(const sky_handle* h, int index, int with_raw, double* read_bytes, double* written_bytes);

#ifdef __cplusplus
}
#endif
#endif /* SKYEYE_HIP_H */
