/*
 * yolo355 -- C ABI of the MI355X-native quantized slim-YOLOv2 inference engine.
 *
 * The reference (ZLkanyo009/Yolo-compression-and-deployment-in-FPGA) has no FFI boundary for
 * this path: the hot path is the Python surface of models/slim_yolo_v2.py and
 * utils/modules.py.  This header is the boundary the drop-in Python
 * classes (yolo355/models/slim_yolo_v2.py, yolo355/utils/modules.py) bind with ctypes; every
 * entry point names the reference code it replaces.  All functions return 0 on success or a
 * negative Y355_E* code; y355_last_error() gives the message of the last failure on the
 * calling thread.
 *
 * Ownership / threading: a y355_engine owns one GPU's device memory, its packed weights and
 * its workspaces.  One handle is single-threaded; different handles are independent (one per
 * GPU / process).  Host buffers passed in stay owned by the caller.  "dev" pointers are HIP
 * device pointers on the engine's GPU (e.g. torch.Tensor.data_ptr()).  Every launch goes to
 * the stream given at creation; calls are asynchronous unless stated otherwise.
 *
 * Data layout in HBM: activations are int8 NHWC with a one-pixel zero halo,
 * [B][H+2][W+2][C]; the fp32 NCHW network input is read directly by the first layer.
 */
#ifndef YOLO355_H
#define YOLO355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define Y355_MAX_ANCHORS 16
#define Y355_NUM_LAYERS 10       /* conv1..conv7 + pred, models/slim_yolo_v2.py:59-87 */
#define Y355_NUM_TRACKERS 11     /* a_tracker_in, a_tracker1..7, a_tracker_pred (:58-89) */

#define Y355_OK 0
#define Y355_EINVAL (-1)         /* bad argument */
#define Y355_EHIP (-2)           /* HIP runtime error */
#define Y355_ENOTREADY (-3)      /* weights / activation exponents missing */
#define Y355_ERANGE (-4)         /* exponent gap does not fit the int32 epilogue */
#define Y355_EGUARD (-5)         /* 2^15 head-room guard tripped (find=True semantics) */

typedef struct y355_engine y355_engine;

typedef struct y355_config {
    int32_t device_id;
    int32_t height, width;        /* network input size, multiples of 16 (input_size=[H,W]) */
    int32_t num_classes;          /* models/slim_yolo_v2.py:42 */
    int32_t num_anchors;
    float anchors[2 * Y355_MAX_ANCHORS]; /* (w,h) in grid units, data/config.py:10-14 */
    float conf_thresh;            /* :42, used at :189 */
    float nms_thresh;             /* :42, used at :171 */
    int32_t max_batch;
    int32_t max_det;              /* per-image cap of returned detections; 0 = all anchors */
    void *stream;                 /* hipStream_t to launch on; NULL = the default (null) stream */
    int32_t own_stream;           /* 1: ignore `stream`, create an engine-owned non-blocking stream */
} y355_config;

/* per-layer counters of the last run of that layer */
typedef struct y355_layer_stats {
    int64_t absmax_t;   /* max |t'| before requantisation (stats mode only) */
    int32_t frac_bits;  /* F': value = t' / 2^F'  (max|y| of slim_yolo_v2.py:22 = absmax_t/2^F') */
    int32_t reserved;
    int64_t saturated;  /* outputs with |q| > 127 that were clamped (reference has no clamp, :35) */
    int64_t guard;      /* outputs violating |y * 2^retune| < 2^15 (:222-227) */
} y355_layer_stats;

const char *y355_last_error(void);
int y355_version(void);

/* replaces SlimYOLOv2_quantize_bnfuse.__init__ / set_grid (models/slim_yolo_v2.py:42-109) */
int y355_create(const y355_config *cfg, y355_engine **out);
void y355_destroy(y355_engine *h);
int y355_set_thresholds(y355_engine *h, float conf_thresh, float nms_thresh);
/* engine options.  Y355_OPT_FUSE_FRONT (default 1): run input quantisation, conv1, pool1, conv2 and pool2
 * (models/slim_yolo_v2.py:218-244; first_conv + second_conv of c_embedding/yolo_forward.c:269-572) as one launch whose
 * 16-channel intermediate map stays on chip (y355_get_feature(h, 0, ...) returns Y355_ENOTREADY after such a forward);
 * 0 = one launch per layer (then y355_get_feature(0) is current after a forward).
 * Results are identical bit for bit either way. */
#define Y355_OPT_FUSE_FRONT 1
/* Y355_OPT_RING_WORKGROUPS (default 0 = one per CU): persistent workgroups per launch of the deep convolutions.  A handle that
 * has the GPU to itself wants them all; when several handles share it (bench.py: three), fewer workgroups that each walk more
 * tiles let launches of different handles run side by side and pay a workgroup's start-up once per several tiles
 * (192 of 256: +2.6 % images/s with three handles, -25 % for a handle running alone).  A non-zero value is the handle's
 * "throughput mode": the prediction layer then also runs on 13 x 26 tiles (128 work items per 64 images instead of 256: +0.9 %)
 * and the NMS pair walk on one workgroup per image instead of two (+1.3 %; a handle alone loses 3 % with it).  Results are
 * identical bit for bit in either mode. */
#define Y355_OPT_RING_WORKGROUPS 2
/* Y355_OPT_FUSE_PAIRS (default 1): run conv3_1, conv3_2 and pool3 (models/slim_yolo_v2.py:246-267; conv_normal calls 3 and 4 of
 * c_embedding/yolo_forward.c:1214-1225) as one launch whose 64-channel intermediate map stays in LDS
 * (y355_get_feature(h, 2, ...) returns Y355_ENOTREADY after such a forward); 0 = one launch per layer.  The fused launch is
 * used where both layers qualify for the fp32-exact epilogue without an accumulator shift and the map is at most 104 pixels
 * wide (network input up to 416 wide; wider maps fall back to the two launches); calibration, statistics runs and guarded
 * forwards (Y355_F_GUARD) always run layer by layer.  Schedule: conv3_1 on the first wave of every SIMD, conv3_2 + pool on the
 * second, one interval apart (+4.5 % images/s against two launches, three handles).  Round 5's two other schedules (every wave
 * alternating between the layers: +2.2 %; conv4_1 -> conv4_2 + pool4 fused the same way: bit-exact, no faster than its two
 * launches) left the library in round 6 -- scratch/pxpair_r5_all_three_kernels.hip, profiles/r05_notes.md sections 3-5.
 * Results are identical bit for bit in both settings.  Values other than 0 / 1: Y355_EINVAL. */
#define Y355_OPT_FUSE_PAIRS 3
int y355_set_option(y355_engine *h, int option, int value);

/* replaces load_state_dict of the quantized checkpoint: integer weights as produced by
 * quantize_layers (retune_bias_quantize.py:111-119): q_w[cout][cin][3][3] int8 with value
 * q_w / 2^e_w, q_b[cout] with value q_b / 2^e_b.  idx 0..9 = conv1..conv7, pred.  Host pointers. */
int y355_load_layer(y355_engine *h, int idx, const int8_t *q_w, const int32_t *q_b,
                    int cout, int cin, int e_w, int e_b);

/* activation exponents floor(log2(tracker.scale)) of the 11 AveragedRangeTrackers
 * (models/slim_yolo_v2.py:33): sa[0] input, sa[1..9] conv1..conv7, sa[10] pred. */
int y355_set_act_exponents(y355_engine *h, const int32_t *sa);
int y355_get_act_exponents(y355_engine *h, int32_t *sa);
int y355_set_act_exponent(y355_engine *h, int tracker, int32_t exponent);   /* one tracker */
/* per-layer scale_retune exponents used by the find=True guard
 * (retune_bias_quantize_findbest.py:122-141, c_embedding/yolo_forward.c:35) */
int y355_set_retune(y355_engine *h, const int32_t *retune);

/* --- calibration stepping (AveragedRangeTracker.quantize_activation, :16-38) -------------
 * max|x| of the fp32 network input (device pointer, NCHW [B,3,H,W]); synchronous. */
int y355_input_absmax(y355_engine *h, const float *x_dev, int batch, float *out_max);
/* run layer idx on the engine's current feature maps.  mode 0: normal fused
 * conv+bias+leaky+requant(+pool).  mode 1: statistics only (fills absmax_t, writes nothing).
 * Layer 0 reads x_dev (fp32 NCHW) and quantises it with sa[0]; other layers ignore x_dev. */
int y355_run_layer(y355_engine *h, int idx, int batch, int mode, const float *x_dev);
/* --- calibration in one call: AveragedRangeTracker.quantize_activation's state machine (models/slim_yolo_v2.py:16-38) for the 11
 * trackers of the path, kept in the handle in the reference's float32 arithmetic (scale = reciprocal(max) * 127 as Python's
 * int / tensor computes it; first call ever: scale += 127 / max even when frozen, :25-27; frozen: unchanged, :28-29; else
 * scale = scale * (1 - momentum) + 127 / max * momentum, :30-31; exponent = floor(log2(scale)), :33).
 *   y355_set_trackers / y355_get_trackers: the checkpoint buffers a_tracker*.scale / first_a (:13-14), 11 each, host arrays
 *   y355_calibrate: one step on a batch = what forward(x, quantization=True) does to the trackers -- layer by layer on the
 *     GPU, every tracker seeing max|activation| of its layer run with the exponents updated in front of it; leaves the 11
 *     exponents set in the handle (and in sa_out[11] / the maxima seen in max_out[11], either may be NULL).  freeze = the
 *     reference's eval mode (trackers only change on their first call), momentum = AveragedRangeTracker.momentum (0.1, :10).
 *     x_dev fp32 NCHW [B,3,H,W] device pointer; synchronous.
 *   y355_tracker_step: the update of ONE tracker as a host utility (no GPU): the arithmetic y355_calibrate applies. */
int y355_set_trackers(y355_engine *h, const float *scale, const int32_t *first_a);
int y355_get_trackers(y355_engine *h, float *scale, int32_t *first_a);
int y355_calibrate(y355_engine *h, const float *x_dev, int batch, int freeze, double momentum, int32_t *sa_out, float *max_out);
int y355_tracker_step(float *scale, int32_t *first_a, float max_abs, int freeze, double momentum, int32_t *exponent);
/* synchronous read-back of a layer's counters */
int y355_layer_stats_get(y355_engine *h, int idx, y355_layer_stats *out);
/* parity tap: copy layer idx's int8 output [B][C][Ho][Wo] (NCHW, halo stripped) to host. */
int y355_get_feature(y355_engine *h, int idx, int batch, int8_t *dst_host);

/* --- the hot path: replaces SlimYOLOv2_quantize_bnfuse.forward(x, quantization=True)
 * (models/slim_yolo_v2.py:212-358) for a whole batch.  x_dev fp32 NCHW [B,3,H,W].
 * Outputs (device pointers, caller-allocated, fixed-cap padded):
 *   boxes  f32 [B][max_det][4]  x1y1x2y2 normalised to [0,1], anchor-index order
 *   scores f32 [B][max_det]
 *   cls    i32 [B][max_det]
 *   count  i32 [B]
 * flags: Y355_F_GUARD also evaluates the find=True head-room guard. Asynchronous. */
#define Y355_F_GUARD 1
#define Y355_F_TAP 2     /* also keep the per-anchor decode (y355_get_candidates) */
int y355_forward(y355_engine *h, const float *x_dev, int batch, int flags,
                 float *boxes_dev, float *scores_dev, int32_t *cls_dev, int32_t *count_dev);
/* the step in front of the path: camera frames as cv2 delivers them, uint8 HWC BGR
 * [B][H][W][3] already at the network size (device pointer).  BaseTransform's (u/255 - mean)/std, the
 * BGR->RGB swap and the HWC->CHW permute (data/__init__.py:30-56, test.py:79) are fused into the first
 * layer's load with the reference's fp32 operations, so the outputs equal y355_forward on the
 * normalised tensor bit for bit while the layer reads a quarter of the bytes.
 * y355_set_normalization: mean / std in the reference's BGR order (defaults data/__init__.py:50). */
int y355_set_normalization(y355_engine *h, const float *mean_bgr, const float *std_bgr);
int y355_forward_u8(y355_engine *h, const uint8_t *frames_dev, int batch, int flags,
                    float *boxes_dev, float *scores_dev, int32_t *cls_dev, int32_t *count_dev);
/* frames of ANY size: the cv2.resize(image, (W, H)) of BaseTransform (data/__init__.py:36) runs on the GPU in front of
 * y355_forward_u8.  frames_dev uint8 HWC BGR [B][src_h][src_w][3]; INTER_LINEAR (cv2's default) as OpenCV computes it for
 * 8-bit images (fixed-point coefficients of 11 bits, imgproc/resize.cpp).  OpenCV is a third-party dependency the
 * reference does not vendor: parity of this stage is unpinned (oracle/resize_oracle.py restates the published algorithm).
 * resized_out_dev (or NULL): [B][H][W][3] copy of the resized frames.  All four output pointers NULL: resize only. */
int y355_forward_u8_resized(y355_engine *h, const uint8_t *frames_dev, int src_h, int src_w, int batch, int flags,
                            float *boxes_dev, float *scores_dev, int32_t *cls_dev, int32_t *count_dev,
                            uint8_t *resized_out_dev);
/* same, host pointers in and out (copies through engine-owned staging buffers); synchronous. */
int y355_forward_host(y355_engine *h, const float *x_host, int batch, int flags,
                      float *boxes, float *scores, int32_t *cls, int32_t *count);
/* sums over layers of the counters of the last forward; synchronous. */
int y355_forward_counters(y355_engine *h, int64_t *saturated, int64_t *guard);
/* parity tap of the head before thresholding (slim_yolo_v2.py:348-350 for every image):
 * boxes f32 [B][N][4], best-class scores f32 [B][N], classes i32 [B][N]; N = anchors per image.
 * Valid after y355_head_nms or a forward with Y355_F_TAP; host pointers; synchronous. */
int y355_get_candidates(y355_engine *h, int batch, float *boxes, float *scores, int32_t *cls);
/* evaluator-side rescale of a batch, in place, on the engine's stream: boxes[b][i] *= (w_b, h_b, w_b, h_b)
 * for i < count[b]; wh_dev = f32 [B][2] (width, height) of the original images.  Replaces `bboxes *= scale`
 * (test.py:88-90, utils/vocapi_evaluator_mask.py:71-72, utils/cocoapi_evaluator.py:77-85); same float32 results. */
int y355_scale_boxes(y355_engine *h, float *boxes_dev, const int32_t *count_dev, const float *wh_dev, int batch);
int y355_max_det(y355_engine *h);              /* effective per-image cap */
int y355_num_anchors_total(y355_engine *h);    /* N = Hs*Ws*A */

/* --- operator-level entry points (utils/modules.py Conv2d_fuse, unit tests) ----------------
 * One fused int8 layer on caller data, host pointers, synchronous:
 *   q_in  int8 [B][cin][H][W] (NCHW), q_w int8 [cout][cin][3][3], q_b int32 [cout]
 *   out   int8 [B][cout][Ho][Wo]; flags bit0 = LeakyReLU(0.125), bit1 = 2x2 max-pool, bit2 = ReLU
 * Requantisation: q_out = clamp(RNE(t' * 2^(sa_out - F'))), see DESIGN.md. */
#define Y355_OP_LEAKY 1
#define Y355_OP_POOL 2
#define Y355_OP_RELU 4   /* ReLU instead of LeakyReLU(0.125): Conv2d_fuse(..., leakyReLU=False), utils/modules.py:26 */
int y355_conv3x3_i8_fused(int device_id, const int8_t *q_in, const int8_t *q_w, const int32_t *q_b,
                          int batch, int cin, int cout, int height, int width,
                          int sa_in, int e_w, int e_b, int sa_out, int flags,
                          int8_t *out, y355_layer_stats *stats);
/* same operands, NO requantisation: out[b][cout][H][W] = t' (int64) and *frac_bits = F' with
 * Conv2d_fuse(x) == t' / 2^F' exactly (utils/modules.py:20-29 on fake-quantized operands). */
int y355_conv3x3_i8_raw(int device_id, const int8_t *q_in, const int8_t *q_w, const int32_t *q_b,
                        int batch, int cin, int cout, int height, int width,
                        int sa_in, int e_w, int e_b, int flags, int64_t *out, int32_t *frac_bits);
/* the two element-wise ops of the path in stand-alone form (the fused layers absorb them); host
 * pointers, synchronous, NCHW:
 *   y355_quantize_input_f32_i8: AveragedRangeTracker.quantize_activation on the network input
 *     (models/slim_yolo_v2.py:33-38): q = clamp(RNE(x * 2^sa), +-127); *clamped = values that hit the clamp
 *   y355_maxpool2x2_i8: nn.MaxPool2d(2, 2) (:61,65,71,77) on int8 [B][C][H][W] -> [B][C][H/2][W/2] */
int y355_quantize_input_f32_i8(int device_id, const float *x, size_t n, int sa, int8_t *q, int64_t *clamped);
int y355_maxpool2x2_i8(int device_id, const int8_t *in, int batch, int channels, int height, int width, int8_t *out);
/* operator API of the wider model families, stand-alone: host pointers, fp32 NCHW, synchronous.
 *   y355_reorg_f32: utils.modules.reorg_layer.forward (utils/modules.py:43-57), [B][C][H][W] -> [B][C*s*s][H/s][W/s]
 *     with out channel (sy*s+sx)*C + c; data movement, bit-exact
 *   y355_spp_f32: utils.modules.SPP.forward (:59-72), [B][C][H][W] -> [B][4C][H][W] =
 *     cat(x, max_pool2d(x,5,1,2), max_pool2d(x,9,1,4), max_pool2d(x,13,1,6)); bit-exact
 *   y355_conv2d_bf16: utils.modules.Conv2d.forward (:6-18) / backbone.darknet.Conv_BN_LeakyReLU (darknet.py:12-22) /
 *     one resblock branch (:24-38) with BatchNorm folded into (w, bias) by the caller: conv (ksize 1, or 3 with
 *     padding 1; stride 1, or 2 with ksize 3) + bias + LeakyReLU(neg_slope) [+ residual]; w [cout][cin][k][k];
 *     residual (or NULL) and out are [B][cout][Ho][Wo].  Operands are rounded to bf16 (RNE), accumulation is
 *     fp32 on the bf16 MFMA, the result is rounded to bf16 (out_fp32 != 0: kept in fp32, as the engines keep
 *     prediction maps; no residual then): parity with the fp32 reference is a tolerance.
 *   y355_maxpool2x2_f32: nn.MaxPool2d((2,2), 2) (backbone/darknet.py:49,55,63,74,83) on fp32; bit-exact */
int y355_reorg_f32(int device_id, const float *x, int batch, int channels, int height, int width, int stride, float *out);
int y355_spp_f32(int device_id, const float *x, int batch, int channels, int height, int width, float *out);
int y355_conv2d_bf16(int device_id, const float *x, const float *w, const float *bias, const float *residual,
                     int batch, int cin, int cout, int height, int width, int ksize, int stride, float neg_slope,
                     int out_fp32, float *out);
int y355_maxpool2x2_f32(int device_id, const float *in, int batch, int channels, int height, int width, float *out);
/* --- device-resident operator forms: the same operators for callers whose tensors already live on the GPU (the drop-in
 * modules called on CUDA tensors).  Device pointers (fp32 NCHW), launched on `stream` behind what it holds, asynchronous, no
 * host copy of any tensor.
 *   y355_reorg_f32_dev / y355_spp_f32_dev / y355_maxpool2x2_f32_dev / y355_upsample2x_f32_dev: the operators above, one launch
 *   y355_conv_op: a convolution whose weights are packed onto the device once.
 *     create_bf16 (w fp32 [cout][cin][k][k], bias or NULL, BN folded by the caller; host pointers) + y355_conv_op_forward =
 *       y355_conv2d_bf16's arithmetic (utils.modules.Conv2d, backbone.darknet.Conv_BN_LeakyReLU, resblock branch)
 *     create_i8 (q_w int8 [cout][cin][3][3], q_b int32 [cout], exponents; flags Y355_OP_LEAKY / Y355_OP_RELU) +
 *       y355_conv_op_forward_i8 = y355_conv3x3_i8_raw's: Conv2d_fuse on a DYADIC x (values q / 2^e, |q| <= 127; utils/modules.py:20-29
 *       on the fake-quantised operands of the quantized path), exact.  The input's exponent (-> *sa_in) and the verdict whether x
 *       is such a tensor (-> *exact; 0: out_dev untouched, take the bf16 route) are decided on the host: two 4-byte read-backs
 *       per call; no tensor leaves the device.
 *   One y355_conv_op is single-threaded; it owns its packed weights and growable workspaces. */
typedef struct y355_conv_op y355_conv_op;
int y355_reorg_f32_dev(const float *x_dev, int batch, int channels, int height, int width, int stride, float *out_dev, void *stream);
int y355_spp_f32_dev(const float *x_dev, int batch, int channels, int height, int width, float *out_dev, void *stream);
int y355_maxpool2x2_f32_dev(const float *in_dev, int batch, int channels, int height, int width, float *out_dev, void *stream);
int y355_upsample2x_f32_dev(const float *in_dev, int batch, int channels, int height, int width, float *out_dev, void *stream);
int y355_conv_op_create_bf16(int device_id, const float *w, const float *bias, int cin, int cout, int ksize, int stride, float neg_slope,
                             y355_conv_op **out);
int y355_conv_op_create_i8(int device_id, const int8_t *q_w, const int32_t *q_b, int cin, int cout, int e_w, int e_b, int flags,
                           y355_conv_op **out);
void y355_conv_op_destroy(y355_conv_op *op);
int y355_conv_op_forward(y355_conv_op *op, const float *x_dev, const float *residual_dev, int batch, int height, int width, int out_fp32,
                         float *out_dev, void *stream);
int y355_conv_op_forward_i8(y355_conv_op *op, const float *x_dev, int batch, int height, int width, float *out_dev, void *stream,
                            int32_t *sa_in, int32_t *exact);
/* F.interpolate(x, scale_factor=2.0, mode='bilinear', align_corners=True) (models/yolo_v3.py:211,215) on fp32:
 * [B][C][H][W] -> [B][C][2H][2W]; fp32 arithmetic (within 1e-6 of torch's) */
int y355_upsample2x_f32(int device_id, const float *in, int batch, int channels, int height, int width, float *out);
/* detection head on fp32 prediction maps, stand-alone (models/yolo_v2.py:183-210; models/tiny_yolo_v3.py:202-262):
 * pred[l] = NCHW [B][A*(5+C)][hs[l]][ws[l]] (host), 1 or 2 levels, channel layout [obj x A | cls x A*C | txtytwth x A*4];
 * anchors [nlev][A][2]; wh_mul = the stride for anchors in grid units (yolo_v2), 1 for anchors in pixels (v3 family).
 * Outputs as y355_forward: boxes f32 [B][max_det][4] normalised x1y1x2y2, scores, classes, counts; anchor-index order.
 * Images with more than 4096 anchors (yolo_v3 at 416 x 416: 10 647) are thresholded and compacted on the GPU before the
 * sort; at most 4096 anchors per image may pass conf_thresh (else Y355_EINVAL). */
int y355_head_f32(int device_id, int nlev, const float *const *pred, const int *hs, const int *ws, const float *strides,
                  const float *anchors, int num_anchors, int num_classes, int in_h, int in_w, float wh_mul,
                  float conf_thresh, float nms_thresh, int batch, int max_det, float *boxes, float *scores,
                  int32_t *cls, int32_t *count);
/* head only: pred int8 [B][A*(5+C)][Hs][Ws] NCHW host -> detections (host), synchronous.
 * Replaces slim_yolo_v2.py:330-358 (decode, score, threshold, per-class NMS). */
int y355_head_nms(y355_engine *h, const int8_t *pred_q, int batch, int sa_pred,
                  float *boxes, float *scores, int32_t *cls, int32_t *count);

/* --- multi-GPU exchange: the batch is sharded over the GPUs of one node (one process per GPU, rank r owns a
 * contiguous range of images, weights replicated, no collective on the data path).  The only exchange is ONE RCCL
 * all-gather per batch of the padded detections, packed into one buffer of fixed-size records:
 *   record = i32 count (-1: padding record of a ragged shard), i32 total (detections the image had; > count when
 *            y355_pack_dets_capped cut the record at max_det; -1 in a padding record), i32 pad[2], f32 boxes[max_det][4],
 *            f32 scores[max_det], i32 cls[max_det], rounded up to 16 bytes = y355_packed_det_bytes(max_det); entries past
 *            `count` are zero.
 * The reference has no multi-GPU code; these are the entry points a multi-process host binds.  RCCL is bound at run time
 * (the copy already loaded in the process, e.g. PyTorch's, is reused).
 *   y355_comm_unique_id   rank 0 makes the 128-byte id and ships it to the other ranks by any host channel
 *   y355_comm_init        every rank, after hipSetDevice-able device_id is known; collective
 *   y355_pack_dets        outputs of y355_forward (batch images) -> `records` >= batch records (the extra ones count -1)
 *   y355_allgather_dets   recv holds world * records records in rank order = global image order; asynchronous on `stream`
 *   y355_unpack_dets      records -> padded arrays: record r goes to row slot[r] (slot[r] < 0: dropped) */
typedef struct y355_comm y355_comm;
#define Y355_COMM_ID_BYTES 128
size_t y355_packed_det_bytes(int max_det);
int y355_comm_unique_id(void *id_out /*[Y355_COMM_ID_BYTES]*/);
int y355_comm_init(y355_comm **out, int world, int rank, const void *id, int device_id);
void y355_comm_destroy(y355_comm *c);
int y355_comm_world(y355_comm *c);
int y355_comm_rank(y355_comm *c);
int y355_pack_dets(const float *boxes_dev, const float *scores_dev, const int32_t *cls_dev, const int32_t *count_dev,
                   int batch, int records, int max_det, void *packed_dev, void *stream);
int y355_allgather_dets(y355_comm *c, const void *packed_send_dev, void *packed_recv_dev, int records, int max_det, void *stream);
/* the same with a record cap below the arrays' own: the arrays hold src_max_det entries per image, the records max_det
 * (the first max_det detections of an image in anchor order; count = min(count, max_det)) -- a gather that ships a fixed
 * 256 detections per image whatever the engine's own cap */
int y355_pack_dets_capped(const float *boxes_dev, const float *scores_dev, const int32_t *cls_dev, const int32_t *count_dev,
                          int batch, int records, int src_max_det, int max_det, void *packed_dev, void *stream);
int y355_unpack_dets(const void *packed_dev, const int32_t *slot_dev, int records, int max_det, float *boxes_dev,
                     float *scores_dev, int32_t *cls_dev, int32_t *count_dev, void *stream);

/* --- y355_pipeline: the throughput regime as a product entry point (csrc/pipeline.hip).  The reference's callers hand the
 * network one batch after another (test.py:84; the evaluator loops utils/vocapi_evaluator_mask.py:57-82,
 * utils/cocoapi_evaluator.py:70-98).  A pipeline owns `handles` engines, each on its own HIP stream, and deals the submitted
 * batches to them round-robin, so that the detection head / NMS of batch i runs beside the convolutions of batch i + 1
 * (+50 % images/s over one handle on one MI355X, profiles/).  Every ticket's result equals a stand-alone y355_forward's bit
 * for bit.  One pipeline is single-threaded like a handle.
 *   create      handles: 1..8, 0 = Y355_PIPE_DEFAULT_HANDLES, the measured optimum (4: 2 / 3 / 4 / 5 / 6 handles deliver 266 / 320 / 326 /
 *               256 / 300 k images/s on one MI355X, profiles/r06_notes.md); ring_workgroups: Y355_OPT_RING_WORKGROUPS of the handles while
 *               more than one shares the GPU, < 0 = the measured optimum (128).  cfg->stream / own_stream are ignored: every
 *               handle gets its own non-blocking stream (y355_pipeline_create) or the caller's (y355_pipeline_create_on).
 *   load_layer / set_act_exponents / set_retune / set_thresholds / set_normalization / set_option / set_trackers: the engine
 *               call of the same name on every handle
 *   calibrate   y355_calibrate on handle 0, then the same trackers and exponents on every handle
 *   submit      enqueue one forward (asynchronous), ticket numbers count from 0.  flags: Y355_F_* | Y355_PIPE_AFTER_STREAM (the
 *               input is produced on `caller_stream`: the forward is ordered behind what is queued there; without the flag the
 *               input must already be complete and caller_stream is ignored).  Outputs: the caller's four device buffers (shapes
 *               as y355_forward) or, all four NULL, pipeline-owned ones read with y355_pipeline_outputs.  A ticket and its
 *               pipeline-owned outputs live until y355_pipeline_depth() (= 2 x handles) further submits.
 *   wait        on_stream = 0: block the host until the ticket is done; 1: make `caller_stream` wait for it (no host block)
 *   release     optional: tell the pipeline that work queued on `caller_stream` so far is the last reader of the ticket's
 *               outputs -- the submit that reuses the ticket's slot is ordered behind it
 *   fetch       wait + copy the detections to host arrays [B][max_det][4] / [B][max_det] / [B][max_det] / [B]; synchronous
 *   scale_boxes y355_scale_boxes on the ticket's outputs (call right after its submit)
 *   counters    y355_forward_counters summed over the handles' last forwards
 *   engine(i)   handle i for the taps / statistics calls of the engine ABI (do not run forwards on it while tickets are in flight)
 *   stream      the HIP stream of the handle that runs `ticket` */
typedef struct y355_pipeline y355_pipeline;
#define Y355_PIPE_AFTER_STREAM 0x100
#define Y355_PIPE_DEFAULT_HANDLES 4
int y355_pipeline_create(const y355_config *cfg, int handles, int ring_workgroups, y355_pipeline **out);
/* the same on `handles` HIP streams of the caller (streams[i] for handle i; `handles` >= 1 here): for hosts whose allocator tracks
 * memory per stream (PyTorch) and must therefore own, and outlive, every stream its tensors are used on */
int y355_pipeline_create_on(const y355_config *cfg, int handles, int ring_workgroups, void *const *streams, y355_pipeline **out);
void y355_pipeline_destroy(y355_pipeline *p);
int y355_pipeline_handles(y355_pipeline *p);
int y355_pipeline_depth(y355_pipeline *p);
int y355_pipeline_max_det(y355_pipeline *p);
y355_engine *y355_pipeline_engine(y355_pipeline *p, int i);
void *y355_pipeline_stream(y355_pipeline *p, long long ticket);
int y355_pipeline_load_layer(y355_pipeline *p, int idx, const int8_t *q_w, const int32_t *q_b, int cout, int cin, int e_w, int e_b);
int y355_pipeline_set_act_exponents(y355_pipeline *p, const int32_t *sa);
int y355_pipeline_set_retune(y355_pipeline *p, const int32_t *retune);
int y355_pipeline_set_thresholds(y355_pipeline *p, float conf_thresh, float nms_thresh);
int y355_pipeline_set_normalization(y355_pipeline *p, const float *mean_bgr, const float *std_bgr);
int y355_pipeline_set_option(y355_pipeline *p, int option, int value);
int y355_pipeline_set_trackers(y355_pipeline *p, const float *scale, const int32_t *first_a);
int y355_pipeline_get_trackers(y355_pipeline *p, float *scale, int32_t *first_a);
int y355_pipeline_calibrate(y355_pipeline *p, const float *x_dev, int batch, int freeze, double momentum, int32_t *sa_out,
                            float *max_out);
int y355_pipeline_submit(y355_pipeline *p, const float *x_dev, int batch, int flags, void *caller_stream, float *boxes_dev,
                         float *scores_dev, int32_t *cls_dev, int32_t *count_dev, long long *ticket);
int y355_pipeline_submit_u8(y355_pipeline *p, const uint8_t *frames_dev, int batch, int flags, void *caller_stream, float *boxes_dev,
                            float *scores_dev, int32_t *cls_dev, int32_t *count_dev, long long *ticket);
int y355_pipeline_wait(y355_pipeline *p, long long ticket, int on_stream, void *caller_stream);
int y355_pipeline_outputs(y355_pipeline *p, long long ticket, float **boxes_dev, float **scores_dev, int32_t **cls_dev,
                          int32_t **count_dev, int *batch);
int y355_pipeline_release(y355_pipeline *p, long long ticket, void *caller_stream);
int y355_pipeline_fetch(y355_pipeline *p, long long ticket, float *boxes, float *scores, int32_t *cls, int32_t *count);
int y355_pipeline_scale_boxes(y355_pipeline *p, long long ticket, const float *wh_dev);
int y355_pipeline_counters(y355_pipeline *p, int64_t *saturated, int64_t *guard);
int y355_pipeline_sync(y355_pipeline *p);

/* --- measurement helpers: HIP events on the engine's stream ------------------------------ */
int y355_sync(y355_engine *h);
void *y355_stream(y355_engine *h);           /* the hipStream_t the handle launches on */
/* per-kernel device time of the last forward run with profiling enabled (ms);
 * slots 0..9 = layers, 10 = head decode, 11 = NMS.  y355_profile(h, 1) turns recording on. */
#define Y355_NUM_TIMERS 12
int y355_profile(y355_engine *h, int enable);
int y355_profile_get(y355_engine *h, float *ms /*[Y355_NUM_TIMERS]*/);
/* y355_profile(h, 2): additionally the layers' own kernel durations (ms, start / end timestamps of the launch itself, what
 * rocprofv3 reports), without the gaps between launches that the slots above include -- and which this mode widens, so take
 * the slots above from a run with y355_profile(h, 1).  0 for layers whose launch does not record them (today: recorded by
 * the six ring-kernel layers conv3_2 .. pred) */
int y355_profile_kernel_get(y355_engine *h, float *ms /*[10]*/);
/* the same for every launch of a forward: slots 0..9 = layers (the fused front end = slot 0, slot 1 then 0), 10 = head decode,
 * 11 = candidate sort, 12 = NMS pair walk, 13 = NMS rounds + output */
#define Y355_NUM_KERNEL_TIMERS 14
int y355_profile_kernels_get(y355_engine *h, float *ms /*[Y355_NUM_KERNEL_TIMERS]*/);
/* host utility: the int8 MFMA rate this GPU sustains (v_mfma_i32_16x16x64_i8 back to back on register operands, two waves per
 * SIMD on every CU, about `ms_target` milliseconds): the measured counterpart of the nominal 5.0 Pop/s bench.py divides by */
int y355_mfma_peak_i8(int device_id, float ms_target, float *tops_out, float *clock_ghz_out);
/* host-only: the 16 KiB of MFMA weight fragments the fused front end (csrc/front.hip: conv1 + pool1 + conv2 + pool2,
 * models/slim_yolo_v2.py:218-244) streams; q_w1 int8 [16][3][3][3], q_w2 int8 [32][16][3][3] (either may be null: its part
 * stays zero).  y355_load_layer does this itself; exported so that the layout can be checked without a GPU. */
int y355_pack_front_weights(const int8_t *q_w1, const int8_t *q_w2, int8_t *dst /*[16384]*/);
/* diagnostic builds (-DY355_DIAG=1): arm / read the s_memtime stamps of one conv layer */
int y355_debug_stamps(y355_engine *h, int layer, unsigned long long *out_host, int nwg);
/* diagnostics (env Y355_NMS_STAMPS=1): s_memtime stamps [4 kernels][256 workgroups][8 slots] of the head / NMS kernels */
int y355_debug_nms_stamps(unsigned long long *out_host);

/* ------------------------------------------------------------------------------------------
 * y355_net: the other model families of the path, table-driven (csrc/net.hip).
 *   Y355_ARCH_SLIM_V2  SlimYOLOv2 (models/slim_yolo_v2.py:386-422, forward :549-622): the fp32
 *                      conv+BN+LeakyReLU(0.125) model; 10 weight slots conv1..conv7, pred.
 *   Y355_ARCH_TINY_V3  YOLOv3tiny (models/tiny_yolo_v3.py:9-273, backbone/darknet.py:211-255);
 *                      13 weight slots in forward order: conv_1..conv_7, conv_set_2, conv_1x1_2,
 *                      conv_set_1, extra_conv_2, pred_2, pred_1.  anchors[] = level stride 16
 *                      first, then stride 32, in pixels (data/config.py:27-31).
 * dtype Y355_DT_BF16: weights are the BN-folded fp32 tensors (utils/bn_fuse.py:21-45, exact
 * fold), rounded to bf16 at load; activations bf16 NHWC with a zero halo; MFMA accumulates in
 * fp32; prediction maps stay fp32.  Same handle rules as y355_engine. */
#define Y355_ARCH_SLIM_V2 0
#define Y355_ARCH_TINY_V3 1
#define Y355_ARCH_YOLO_V2 2   /* myYOLOv2 (models/yolo_v2.py:9-232) on DarkNet-19 (backbone/darknet.py:40-110); bf16 only */
#define Y355_ARCH_YOLO_V3 3   /* myYOLOv3 (models/yolo_v3.py:9-304) on DarkNet-53 (backbone/darknet.py:112-161); bf16 only */
#define Y355_ARCH_YOLO_V3_SPP 4   /* myYOLOv3Spp (models/yolo_v3_spp.py): SPP in front of the stride-32 branch; bf16 only */
#define Y355_DT_INT8 0
#define Y355_DT_BF16 1
typedef struct y355_net y355_net;
typedef struct y355_net_config {
    int32_t device_id;
    int32_t arch, dtype;
    int32_t height, width;        /* multiples of 32 (Y355_ARCH_SLIM_V2: of 16) */
    int32_t num_classes;
    int32_t num_anchors;          /* per prediction level */
    float anchors[2 * Y355_MAX_ANCHORS];
    float conf_thresh, nms_thresh;
    int32_t max_batch, max_det;
    void *stream;
    int32_t own_stream;
} y355_net_config;

int y355_net_create(const y355_net_config *cfg, y355_net **out);
void y355_net_destroy(y355_net *h);
int y355_net_set_thresholds(y355_net *h, float conf_thresh, float nms_thresh);
/* Y355_NET_OPT_WORKGROUPS (default 0 = one per CU): the net's "throughput mode", as Y355_OPT_RING_WORKGROUPS of the q_bf engine:
 * persistent workgroups per launch of the 3x3 ring kernels (convr.hip) while several handles share the GPU, and the NMS pair walk on
 * one workgroup per image.  Results are identical bit for bit. */
#define Y355_NET_OPT_WORKGROUPS 1
/* Y355_NET_OPT_THIN_RESIDENT (default 1): the thin 3x3 layers of the bf16 graphs (SlimYOLOv2's conv3_1, conv3_2 + pool3, conv4_1:
 * K = 288 .. 576) keep their weights in registers and take the pixels as the MFMA's B operand (csrc/convpxb.hip, the bf16 form of the
 * q_bf engine's convpx.hip); 0 = the LDS-ring kernels of convr.hip as before.  Same arithmetic up to where the bias enters the fp32
 * accumulation: the two routes agree within 2 bf16 ulps, not bit for bit. */
#define Y355_NET_OPT_THIN_RESIDENT 2
int y355_net_set_option(y355_net *h, int option, int value);
int y355_net_num_layers(y355_net *h);
int y355_net_num_tensors(y355_net *h);
int y355_net_layer_shape(y355_net *h, int idx, int32_t *shape /*[4] cout,cin,kh,kw*/);
int y355_net_tensor_shape(y355_net *h, int idx, int32_t *shape /*[3] C,H,W*/);
/* replaces load_state_dict + fuse_conv_and_bn for one conv: w fp32 [cout][cin][k][k], b fp32 [cout]
 * (NULL = zero), host pointers */
int y355_net_load_layer_f32(y355_net *h, int idx, const float *w, const float *b, int cout, int cin, int ksize);
/* dtype Y355_DT_INT8: the power-of-two int8 recipe of the q_bf path (retune_bias_quantize.py:73-119,
 * models/slim_yolo_v2.py:16-38) applied to these graphs -- the reference itself has no int8 form of
 * them.  q_w int8 [cout][cin][k][k] (value q_w / 2^e_w), q_b int32 [cout] (value q_b / 2^e_b).
 * LeakyReLU(0.1) runs as the fixed-point slope 205/2048; DESIGN.md lists the integer semantics. */
int y355_net_load_layer_i8(y355_net *h, int idx, const int8_t *q_w, const int32_t *q_b, int cout, int cin,
                           int ksize, int e_w, int e_b);
/* activation exponents (value = q / 2^sa): sa_in for the network input, sa[t] per activation tensor in
 * graph order; a max-pool output takes its input's exponent (its entry is overridden), a concat
 * buffer has one exponent.  get returns the effective values. */
int y355_net_set_act_exponents(y355_net *h, int sa_in, const int32_t *sa, int n);
int y355_net_get_act_exponents(y355_net *h, int32_t *sa_in, int32_t *sa, int n);
/* outputs clamped to +-127 by the last forward of an int8 net; synchronous */
int y355_net_counters(y355_net *h, int64_t *saturated);
/* replaces SlimYOLOv2.forward (:549-601) / YOLOv3tiny.forward (tiny_yolo_v3.py:176-243) for a whole
 * batch; arguments and outputs as y355_forward. */
int y355_net_forward(y355_net *h, const float *x_dev, int batch, int flags,
                     float *boxes_dev, float *scores_dev, int32_t *cls_dev, int32_t *count_dev);
int y355_net_get_candidates(y355_net *h, int batch, float *boxes, float *scores, int32_t *cls);
/* parity tap: activation tensor idx (graph order, see csrc/net.hip) as fp32 NCHW on the host.
 * Tensor taps and calibration need a forward with Y355_F_TAP: on SlimYOLOv2 / YOLOv3tiny graphs a plain forward runs the first
 * two layers in one launch and never writes the first layer's map -- reading that tensor (here or through
 * y355_net_tensor_absmax) after such a forward returns Y355_ENOTREADY instead of stale data, as y355_get_feature(h, 0, ...)
 * does on the q_bf engine. */
int y355_net_get_tensor(y355_net *h, int idx, int batch, float *dst_host);
/* max |value| of an activation tensor of the last forward (calibration of the int8 recipe); see the note above */
int y355_net_tensor_absmax(y355_net *h, int idx, int batch, float *out_max);
int y355_net_max_det(y355_net *h);
int y355_net_num_anchors_total(y355_net *h);
int y355_net_sync(y355_net *h);
/* Diagnostics (no reference counterpart): the last forward's NMS work per image -- count[batch] candidates at or above
 * conf_thresh, nedges[2 * batch] = (suppressing pairs listed, 1 when the list was abandoned for the sorted walk).  Synchronous. */
int y355_net_debug_nms(y355_net *h, int batch, int32_t *count, int32_t *nedges);
/* heads with more than 4096 anchors per image (yolo_v3 at 416 x 416): *overflow = 1 if, in a forward since the last call, more
 * than 4096 anchors of an image passed conf_thresh (the excess was dropped: raise the threshold); synchronous; clears the flag */
int y355_net_overflow(y355_net *h, int *overflow);
int y355_net_profile(y355_net *h, int enable);
int y355_net_num_timers(y355_net *h);          /* ops + head decode + NMS */
int y355_net_profile_get(y355_net *h, float *ms);

#ifdef __cplusplus
}
#endif
#endif /* YOLO355_H */
