/*
 * playaid_hip.h -- C ABI of the MI355X (gfx950) action-recognition hot path.
 *
 * The reference (NathanBWaters/playaid_core) is pure Python and has no FFI; the
 * path sits behind three Python call boundaries. Each
 * entry point below names the reference interface it replaces. The Python host
 * in playaid_core_amd/ binds this library with ctypes and re-exposes the
 * reference's own classes (CNNActionDetector, AIRunner, YoloCrop); see
 * INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions
 *   - All data pointers are DEVICE pointers (HBM) unless the name ends in
 *     _host. The caller owns every buffer. `stream` is a hipStream_t passed as
 *     void* (NULL = default stream). Calls only enqueue work; nothing here
 *     synchronises except pa_create / pa_destroy / pa_profile_read and the
 *     *_sync helpers.
 *   - Return value: PA_OK (0) or a negative pa_status. pa_last_error() gives a
 *     human-readable message for the last failing call on that engine. The
 *     reference's asserts / exit() (ai_runner.py:240-242,447,458; fighter.py
 *     :356-362) become status codes; per-crop failures are reported in the
 *     per-crop `status` words, not as a failed call.
 *   - One engine per device; an engine is not thread-safe (the reference is
 *     single-threaded).
 *   - Frames are numbered from 1 like the reference's YOLO files
 *     (ai_runner.py:516); `frame0` arguments are 0-based indices into a clip.
 */
#ifndef PLAYAID_HIP_H
#define PLAYAID_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PA_ABI_VERSION 11
#define PA_WEIGHT_MAGIC 0x31574150 /* "PAW1" */
#define PA_LSTM_MAGIC 0x314c4150   /* "PAL1" */
#define PA_ENCODER_MAGIC 0x31454150 /* "PAE1" */
#define PA_FEATURE_STRIDE 1024     /* floats per cached feature row (1000 used) */
#define PA_CROP 128

typedef enum pa_status {
    PA_OK = 0,
    PA_ERR_INVALID_ARG = -1,
    PA_ERR_HIP = -2,          /* a HIP runtime call failed; see pa_last_error */
    PA_ERR_BAD_WEIGHTS = -3,  /* blob size / magic / shape mismatch */
    PA_ERR_CAPACITY = -4,     /* more frames/windows than the engine was created for */
    PA_ERR_NO_DEVICE = -5,
    PA_ERR_NOT_READY = -6     /* head asked for frames whose features are not cached */
} pa_status;

/* per-crop status words written by the preprocess stage */
#define PA_CROP_OK 0
#define PA_CROP_EMPTY 1        /* empty / off-screen slice: reference returns (False, None), fighter.py:356-362 */
#define PA_CROP_BAD_BOX 2      /* non-finite box or square side <= 0 (reference raises) */
#define PA_CROP_UPSCALE 3      /* reserved (was: square side < 128 px rejected; the enlarging branch is implemented now) */
#define PA_CROP_FILTER_TOO_WIDE 4 /* bicubic support beyond the kernel's table size (scale > 3.5) */
#define PA_CROP_BAD_FRAME 5    /* pa_backbone_frames_src: source index outside the frame buffer */

typedef struct pa_engine pa_engine;

/* Geometry of the path. Defaults mirror ai_runner.py:426-439 and
 * cnn_action_detector.py:14-27. */
typedef struct pa_config {
    int32_t abi_version;       /* PA_ABI_VERSION */
    int32_t device_id;         /* HIP device ordinal */
    int32_t sequence_length;   /* S, odd; 7 (ai_runner.py:432) */
    int32_t frame_delta;       /* 3 (ai_runner.py:433-437) */
    int32_t num_actions;       /* A; 63 (anim_ontology.py:592-600) */
    int32_t num_fighters;      /* crops per frame; 2 (ai_runner.py:240-242) */
    int32_t crop_padding;      /* 30 px (ai_runner.py:417-418) */
    int32_t max_batch_frames;  /* most frames handed to one pa_backbone_frames call */
    int32_t max_clip_frames;   /* feature-cache capacity in frames */
    int32_t max_frame_height;  /* scratch sizing for the resampler */
    int32_t max_frame_width;
    int32_t fighter_class_ids[4]; /* CHAR_LIST index of each fighter slot (constants.py:51) */
    int32_t compute_dtype;     /* PA_DTYPE_F32 (default, the reference's arithmetic), PA_DTYPE_BF16 or PA_DTYPE_EMULATED_F32 */
} pa_config;

/* compute_dtype. PA_DTYPE_BF16 (BASELINE.json configs[2]) stores the activations and folded
 * weights of the sixteen 3x3 convolutions in bf16 and multiplies them on the bf16 matrix cores
 * with fp32 accumulation; the stem, the fc, the temporal head and every interface stay fp32.
 * It is NOT within the 1e-4 parity bar of the fp32 path (tests state its own tolerance). */
#define PA_DTYPE_F32 0
#define PA_DTYPE_BF16 1
/* PA_DTYPE_EMULATED_F32 (ABI 11; never the default): fp32 inputs, fp32 outputs and fp32-accurate sums, but the products of the
 * convolutions run on the bf16 matrix cores -- every fp32 operand as three bf16 slices (an exact decomposition of its 24-bit
 * significand), the six leading cross products per fp32 product on v_mfma_f32_32x32x16_bf16, fp32 accumulation
 * (csrc/psgemm.hip). As close to a float64 run as the exact fp32 kernels are, and inside the same 1e-4 parity bars (the fp32
 * parity tests run under both values); summation order and rounding differ from the fp32 instruction's, so results are NOT
 * bit-identical to PA_DTYPE_F32. Layers without an emulated kernel run the exact fp32 one. bench.py's `value` / `dtype` stay
 * on PA_DTYPE_F32; this value is reported as the side block `emulated_fp32`. */
#define PA_DTYPE_EMULATED_F32 2

/* Result record per (frame, fighter): the fields AIRunner.action_recognition
 * derives at ai_runner.py:474-479. confidence% = prob * 100 is left to the host
 * (the reference multiplies in Python double). */
typedef struct pa_record {
    int32_t char_id;    /* CHAR_LIST.index(fighter) */
    int32_t action_id;  /* int(torch.argmax(predictions)) */
    float prob;         /* exp(logp[action_id]) */
    int32_t status;     /* PA_CROP_* of the window's middle crop */
} pa_record;

/* One row per kernel family of the last profiled call(s). */
typedef struct pa_kernel_stat {
    char name[48];
    int32_t launches;
    float total_ms;     /* sum of HIP-event durations on the engine's stream */
    double flops;       /* algorithmic FLOPs summed over those launches (a convolution: 2 x outputs x k x k x cin) */
    double bytes;       /* algorithmic (compulsory) HBM bytes summed over those launches */
    double flops_executed; /* multiply-adds the matrix cores actually ran, x 2: equal to `flops` except for launches of the
                              Winograd F(2x2, 3x3) kernel, which execute 4 / 9 of their layer's direct-form count (ABI 9) */
} pa_kernel_stat;

/* ---- lifetime ----------------------------------------------------------- */

/* Replaces CNNActionDetector.load_from_checkpoint(...).eval()
 * (ai_runner.py:164-168; cnn_action_detector.py:46-92).
 * weight_blob_host: int32 header {PA_WEIGHT_MAGIC, 1, S, A, 0,0,0,0} followed by
 * the fp32 tensors of the Lightning state_dict in this order, PyTorch layouts:
 *   model.cnn2d.conv1.weight[64,3,7,7], bn1.{weight,bias,running_mean,running_var}[64],
 *   for layer 1..4, block 0..1: conv1.weight, bn1.{w,b,mean,var}, conv2.weight,
 *     bn2.{w,b,mean,var}, then (block 0 of layers 2..4) downsample.0.weight,
 *     downsample.1.{w,b,mean,var};
 *   fc.weight[1000,512], fc.bias[1000];
 *   model.cnn1d.0.weight[512,1000,S], model.cnn1d.0.bias[512];
 *   model.classifier.0.weight[128,512], .bias[128];
 *   model.classifier.2.weight[A,128], .bias[A].
 * BatchNorm (eval, eps 1e-5) is folded into the preceding convolution in fp64
 * on the host and the tensors are re-laid-out for the kernels here. */
int pa_create(const pa_config* cfg, const void* weight_blob_host, size_t blob_bytes, pa_engine** out);
/* Weights for the other GPUs of a node (one broadcast of the weights). The
 * folded, re-laid-out tensors of an engine live in one device arena whose layout depends only on
 * (sequence_length, num_actions, compute_dtype): pa_weights_export copies it into a caller buffer of
 * pa_weights_arena_bytes() bytes (device to device, on `stream`); after that buffer has been broadcast
 * (RCCL), every other rank builds its engine from it with pa_create_from_arena -- no host copy of the
 * weights, no second BatchNorm fold. The reference has no counterpart (single process,
 * ai_runner.py:164-168). */
int pa_create_from_arena(const pa_config* cfg, const void* arena_dev, size_t arena_bytes, pa_engine** out);
size_t pa_weights_arena_bytes(const pa_engine* e);
int pa_weights_export(pa_engine* e, void* arena_dst_dev, size_t arena_bytes, void* stream);
void pa_destroy(pa_engine* e);
const char* pa_last_error(const pa_engine* e);
const char* pa_status_string(int status);
int pa_abi_version(void);
/* Expected blob size in bytes for (S, A). */
size_t pa_weight_blob_bytes(int sequence_length, int num_actions);

/* ---- b1: the operator --------------------------------------------------- */

/* Replaces `logp = model(x)` (cnn_action_detector.py:86-92; ai_runner.py:472).
 * x: float32[B,S,3,128,128] NCHW contiguous, values in [0,1].
 * logp: float32[B,A] log-probabilities. B*S <= max_batch_frames*num_fighters. */
int pa_infer_windows(pa_engine* e, const float* x, int32_t batch, float* logp, void* stream);

/* ---- a6: crop preprocessing --------------------------------------------- */

/* Replaces YoloCrop.square_crop(frame, 128, padding) (fighter.py:323-381) for
 * n frames x num_fighters boxes at once.
 * frames: uint8[n,H,W,3] HWC (BGR as cv2 delivers it, ai_runner.py:404-405).
 * boxes:  float64[n,num_fighters,4] normalised (cx,cy,w,h) (ai_runner.py:53-71).
 * crops:  uint8[n,num_fighters,128,128,3]; channel order kept when swap_rb=0
 *         (square_crop), reversed when swap_rb=1 (the BGR2RGB of ai_runner.py:448).
 * status: int32[n,num_fighters] PA_CROP_*; failed crops are all zero. */
int pa_square_crops(pa_engine* e, const uint8_t* frames, int32_t n, int32_t height, int32_t width,
                    const double* boxes, int32_t padding, int32_t swap_rb, uint8_t* crops,
                    int32_t* status, void* stream);

/* ---- a5: the runner's own input branch ----------------------------------- */

/* One crop image inside a byte buffer: what YOLOv5 --save-crop wrote and cv2.imread returns
 * (ai_runner.py:445-446), uint8 [height][width][3], BGR, rows packed. */
typedef struct pa_crop_image {
    int64_t offset;  /* byte offset of the image in `images` */
    int32_t height;
    int32_t width;
} pa_crop_image;

/* Replaces the per-frame body of AIRunner.get_action_recognition_input_for_frame (ai_runner.py:446-459)
 * for n_crops crop images of ANY size at once: BGR2RGB (swap_rb = 1) -> imutils.resize(width=128)
 * = cv2.resize(INTER_AREA) to (128, int(h * (128 / w))) -> ImageOps.pad((128, 128), black) when that is
 * not 128 rows (tall crops are BICUBIC-shrunk and letterboxed left/right, wide ones letterboxed top/bottom).
 * images: device bytes; desc: device pa_crop_image[n_crops]; inputs_u8: uint8[n_crops][128][128][3] (the
 * `frames` list of :459-464); status: int32[n_crops] PA_CROP_* (PA_CROP_BAD_BOX: descriptor outside the
 * buffer or larger than max_frame_height x max_frame_width; PA_CROP_EMPTY: int(h*128/w) == 0, where
 * cv2.resize raises; PA_CROP_FILTER_TOO_WIDE: taller than 3.5 x 128 rows after the resize). */
int pa_runner_inputs(pa_engine* e, const uint8_t* images, size_t images_bytes, const pa_crop_image* desc, int32_t n_crops,
                     int32_t swap_rb, uint8_t* inputs_u8, int32_t* status, void* stream);

/* pa_backbone_frames for a clip that arrives as crop IMAGES instead of frames + boxes (the reference's
 * actual on-disk hand-off, crops/<Fighter>/<video>_<n>.jpg after decoding): n frames x num_fighters images in
 * (frame, fighter) order -> runner inputs -> ResNet-18 features in the cache rows of frames frame0... */
int pa_backbone_crop_images(pa_engine* e, const uint8_t* images, size_t images_bytes, const pa_crop_image* desc, int32_t n,
                            int32_t frame0, uint8_t* crops_rgb, int32_t* status, void* stream);

/* ---- f1: detection post-processing --------------------------------------- */

/* Replaces the post-network half of the YOLOv5 subprocess AIRunner.run_yolo starts (ai_runner.py:191-224:
 * detect.py --max-det 2 --save-txt --save-conf --classes 2 3): from the detection head's decoded rows
 * pred[n_frames][rows][5 + num_classes] (cx cy w h in network-input pixels, objectness, class scores; device
 * float32) to the numbers of the label files the reference parses (ai_runner.py:53-94): objectness and
 * confidence gates (conf_thres, default 0.25), best class per row, class filter (class_mask bit c = class c
 * allowed; the reference passes classes 2 and 3), class-aware IoU NMS (iou_thres, default 0.45), at most
 * max_det (<= 8) boxes, mapped from the letterboxed net_height x net_width input back to the img_height x
 * img_width frame, rounded to pixels and normalised.
 * dets: float32[n_frames][max_det][6] = cls cx cy w h conf in LABEL-FILE order (detect.py writes
 * reversed(det): lowest confidence first); counts: int32[n_frames]. The host writes each row as six '%g'
 * fields (playaid_core_amd/detect.py). The arithmetic is the un-vendored ultralytics/yolov5 checkout's; the
 * contract is restated in oracle/detect.py ("parity unpinned"). */
int pa_detect_postprocess(pa_engine* e, const float* pred, int32_t n_frames, int32_t rows, int32_t num_classes,
                          float conf_thres, float iou_thres, uint32_t class_mask, int32_t max_det, int32_t net_height,
                          int32_t net_width, int32_t img_height, int32_t img_width, float* dets, int32_t* counts, void* stream);

/* ---- f1: the detection network ---------------------------------------------
 *
 * Replaces the network half of the YOLOv5 subprocess (ai_runner.py:191-224: `detect.py --weights <yolov5s checkpoint>
 * --source <video>`): frames -> letterbox (cv2.resize INTER_LINEAR to the un-padded size, 114-grey border, BGR -> RGB,
 * / 255) -> a fully convolutional network given as a LAYER TABLE -> Detect decode -> pred float32[n][rows][5 + nc] in
 * network-input pixels, i.e. the input of pa_detect_postprocess. The table (playaid_core_amd/yolov5.py builds YOLOv5s
 * v7.0 from an ultralytics state dict) arrives BatchNorm-folded: per convolution [cout][ky][kx][cin] at w_off and the bias
 * [cout] at b_off (float offsets into one blob); the 6x6 stem as [cout][6 ky][8 px][4 ch] (kx >= 6 and channel 3 zero, cout % 64 == 0). Activations are
 * zero-bordered NHWC device buffers of buf_floats_per_image[b] floats per image; a layer addresses a CHANNEL SLICE of a
 * buffer (coff = first channel, cstride = channels per pixel of the buffer), so concatenations are free. */
typedef struct pa_net_layer {
    int32_t kind;      /* 0 convolution, 3 stem 6x6/2 + SiLU on the letter-boxed image (cout == 32; weights float32[64][56]: lane l =
                          (kx / 3) * 32 + channel holds W[channel][c][ky][3 * (l / 32) + j] at ky * 9 + j * 3 + c), 4 max-pool 5x5/1,
                          5 nearest 2x up-sampling, 6 Detect decode of one scale (3 anchors) */
    int32_t cin, cout; /* kind 0: cin % 32 == 0, cout % 32 == 0 (pad with zero weights; cout % 64 != 0 needs in_pad == 1 for a 3x3 and no residual for a 1x1); kinds 4 / 5: cin = channels moved */
    int32_t ksize, stride;           /* kind 0: 1 | 3, 1 | 2 */
    int32_t in_h, in_w;              /* interior of the input */
    int32_t in_buf, in_coff, in_cstride, in_pad;
    int32_t out_buf, out_coff, out_cstride, out_pad;
    int32_t res_buf, res_coff;       /* residual slice (-1: none): the output's geometry and cstride */
    int32_t act;                     /* 0 none, 1 ReLU, 2 SiLU */
    int32_t res_after;               /* 1: residual added after the activation (YOLOv5 Bottleneck), 0: before (ResNet) */
    int32_t reserved;
    int64_t w_off, b_off;
    float aux[8];                    /* kind 6: stride of the scale, then 3 anchors (w, h) in network-input pixels */
} pa_net_layer;
typedef struct pa_detector pa_detector;
int pa_detector_create(int32_t device, const pa_net_layer* layers, int32_t n_layers, const int64_t* buf_floats_per_image, int32_t n_bufs,
                       const float* weights_host, size_t n_weights, int32_t max_images, int32_t net_h, int32_t net_w, int32_t num_classes,
                       pa_detector** out);
/* The same with the convolutions' arithmetic chosen (ABI 11): PA_DTYPE_F32 (= pa_detector_create) or PA_DTYPE_EMULATED_F32 -- the
 * 1x1 and stride-2 3x3 convolutions then run on the bf16 matrix cores with fp32-accurate sums (see PA_DTYPE_EMULATED_F32; the
 * stride-1 3x3 layers keep their exact Winograd kernel), and so does the 6x6 stem: the letter-boxed pixels are integers 0..255,
 * one exact bf16 value each, multiplied with the three bf16 slices of W / 255 (csrc/yolo.hip, stem6x6_bf16_kernel). Never the default. */
int pa_detector_create_dtype(int32_t device, const pa_net_layer* layers, int32_t n_layers, const int64_t* buf_floats_per_image, int32_t n_bufs,
                             const float* weights_host, size_t n_weights, int32_t max_images, int32_t net_h, int32_t net_w, int32_t num_classes,
                             int32_t compute_dtype, pa_detector** out);
void pa_detector_destroy(pa_detector* h);
const char* pa_detector_last_error(const pa_detector* h);
int pa_detector_rows(const pa_detector* h); /* rows of pred per image: 3 x sum over the decode layers of in_h x in_w */
/* frames uint8[n,H,W,3] BGR (device) -> pred float32[n][rows][5 + nc] (device). */
int pa_detector_forward(pa_detector* h, const uint8_t* frames, int32_t n, int32_t height, int32_t width, float* pred, void* stream);
/* Measurement aid: the same call with a HIP event between the table's layers; layer_us[i] (host, cap >= n_layers) = microseconds
 * layer i took on `stream`. Synchronises the stream. */
int pa_detector_forward_timed(pa_detector* h, const uint8_t* frames, int32_t n, int32_t height, int32_t width, float* pred, void* stream,
                              float* layer_us, int32_t cap);

/* Replaces AIRunner.clean_yolo_crops / clean_yolo_crops_for_fighter (ai_runner.py:226-289, 306-424) on the table
 * pa_detect_postprocess wrote (dets float32[n_labels][max_det][6], counts int32[n_labels]; label n = index n - 1), with no
 * label files in between: per fighter (cfg.fighter_class_ids) duplicate detections of its class are resolved (nearest
 * centre, L1, to the class's previous box), the frames it is missing in get boxes interpolated FROM THE END frame and
 * pixels from VideoCapture position j (one decoded frame late; a read past n_decoded_frames copies the previous crop),
 * and the fighter whose crops end first gets its last crop copied up to, not including, the other's last frame. The rows
 * are taken through the label file's '%g' formatting first, so the tables equal the host mirror's
 * (playaid_core_amd/label_cleaning.py, which works on the label text) bit for bit. All device pointers:
 * labels float64[n_labels][F][6] = every fighter's repaired label row (cls cx cy w h conf; cls < 0: none);
 * pixel_frame int32[n_labels][F] = decoded frame each crop is cut from (-1 none); pixel_box float64[n_labels][F][4];
 * crop_kind int32[n_labels][F]: 1 the detector's own crop (save_one_box of crop_row), 2 a square_crop repair, 0 none;
 * crop_row float32[n_labels][F][6]; info4 int32[4] = max_frames (number of the last non-empty label), error code
 * (0 ok; 1 duplicate detections of a class never seen before, ai_runner.py:343; 2 a gap before a fighter's first
 * detection, :375-378; 3 a fighter without any detection), the label number it happened at, duplicates resolved. Rows
 * behind max_frames of pixel_frame / crop_kind are written as "no crop" (-1 / 0) whatever the caller's memory held. Enqueue
 * only; the call uses ONE scratch per engine (sized at pa_create for max_clip_frames x 8 detections; a longer table grows it
 * behind a device synchronisation), so the calls of one engine belong on one stream, or on streams the caller orders. */
int pa_clean_detections(pa_engine* e, const float* dets, const int32_t* counts, int32_t n_labels, int32_t max_det, int32_t n_decoded_frames,
                        double* labels, int32_t* pixel_frame, double* pixel_box, int32_t* crop_kind, float* crop_row, int32_t* info4,
                        void* stream);

/* What the crop hand-off needs from pa_clean_detections' tables, worked out on the device (one launch instead of a dozen
 * generic element-wise / sort launches between the repair and the crops): det_index / src_own int32[n_labels][F] = the
 * det_index / src_frame arguments of pa_save_one_box_crops for the entries of kind 1 (-1 / 0 elsewhere; rows behind
 * max_frames count as "no crop"); the square_crop repairs (kind 2, rows < max_frames) compacted in entry order:
 * rep_entry int32[] = their (frame * F + fighter) entries, rep_boxes float64[][4], rep_src int32[] = source frames --
 * each with room for n_labels * F entries, padded with copies of the first repair to a whole number of F (the shape
 * pa_square_crops_src cuts); words5 int32[5] = info4 followed by the number of repairs (the five words a host waits
 * for). Enqueue only. */
int pa_detector_plan(pa_engine* e, const int32_t* pixel_frame, const double* pixel_box, const int32_t* crop_kind, const int32_t* info4,
                     int32_t n_labels, int32_t* det_index, int32_t* src_own, int32_t* rep_entry, double* rep_boxes, int32_t* rep_src,
                     int32_t* words5, void* stream);
/* The clip's crop-image descriptors once pa_save_one_box_crops has packed every chunk of `step_frames` frames into a region
 * of `region_bytes` of its own (descriptor offsets relative to the region): kind-1 entries move by their chunk's region,
 * the n_rep repairs' 128 x 128 x 3 images sit back to back from byte `rep_base`. desc[n_frames][F]. Enqueue only. */
int pa_detector_plan_desc(pa_engine* e, pa_crop_image* desc, const int32_t* crop_kind, int32_t n_frames, int32_t step_frames,
                          long long region_bytes, const int32_t* rep_entry, int32_t n_rep, long long rep_base, void* stream);
/* pa_square_crops with every crop cut from the frame src_frame names (int32[n][F], device; frames uint8[n_src,H,W,3]):
 * the square_crop repairs of clean_yolo_crops, whose pixels come from VideoCapture position j (ai_runner.py:404-418). */
int pa_square_crops_src(pa_engine* e, const uint8_t* frames, int32_t n_src, int32_t height, int32_t width, const double* boxes,
                        const int32_t* src_frame, int32_t n, int32_t padding, int32_t swap_rb, uint8_t* crops, int32_t* status, void* stream);

/* Replaces the crop half of the same subprocess (`--save-crop`, ai_runner.py:208) and the cv2.imread that reads each
 * crop back (:445-446): per (frame, fighter) YOLOv5 v7.0's utils/plots.py::save_one_box -- the label row's pixel box ->
 * xyxy2xywh -> wh * 1.02 + 10 -> xywh2xyxy -> .long() -> clip_boxes -> im[y1:y2, x1:x2] (float32 like torch) -- and, for
 * jpeg_quality 1..100, the pixel arithmetic of its Image.save(quality=95, subsampling=0) + a JPEG read (4:4:4 baseline
 * JPEG of an image of ANY size, edge blocks filled by edge replication; jpeg_quality 0 leaves the raw cut).
 * frames uint8[n_src,H,W,3] BGR; n clip frames; src_frame int32[n][num_fighters] (device; NULL = every crop from its own
 * frame, n_src == n) says which frame a crop's pixels come from (the reference's tail repair copies the last crop FILE of a
 * fighter to later frames, ai_runner.py:270-289); dets float32[n][max_det][6] / counts int32[n] as pa_detect_postprocess
 * wrote them (label rows, label-file order);
 * det_index int32[n][num_fighters] (device) = which detection of its frame each fighter's crop is cut from, -1 none, or
 * NULL = the first detection of the fighter's class (cfg.fighter_class_ids) in label order, i.e. the crop file whose name
 * carries no counter. The BGR images land back to back (16-byte aligned) in `images`, desc[n][num_fighters] describes them
 * (height = width = 0: no detection / empty rectangle / no room left in `images`, the latter also counted for
 * pa_device_errors): exactly what pa_runner_inputs / pa_backbone_crop_images take. Restated in oracle/detect.py
 * (rectangle: parity unpinned, YOLOv5 is not vendored) + oracle/jpeg.py::roundtrip_any (pinned to live libjpeg-turbo). */
int pa_save_one_box_crops(pa_engine* e, const uint8_t* frames, int32_t n_src, int32_t height, int32_t width, const float* dets,
                          const int32_t* counts, int32_t max_det, const int32_t* det_index, const int32_t* src_frame, int32_t n,
                          int32_t jpeg_quality, uint8_t* images, size_t images_capacity, pa_crop_image* desc, void* stream);

/* Boxes from the game log instead of a detector. Replaces the
 * projection half of Fighter.set_from_json (fighter.py:494-539: calculate_lookat_matrix,
 * calculate_intrinsic_matrix, project_point_to_pixel on four corners, all for the
 * reference's hard-coded 1280x720 image) + YoloCrop.from_pixel_coordinates (fighter.py:170-190).
 * log_rows: float64[n_rows,9] = pos_x, pos_y, camera_position xyz, camera_target_position
 * xyz, fov in degrees (STAGE_ENUM_TO_DATA[stage]["fov"], fighter.py:488); boxes:
 * float64[n_rows,4] normalised (cx,cy,w,h), directly usable as the `boxes` of the calls below. */
int pa_project_boxes(pa_engine* e, const double* log_rows, int32_t n_rows, double* boxes, void* stream);

/* ---- b2: the runner loop ------------------------------------------------ */

/* Start a clip of clip_frames frames (= the reference's max_frames,
 * ai_runner.py:244-245): sets the clamp range of the window sampler and marks
 * the feature cache empty. */
int pa_clip_begin(pa_engine* e, int32_t clip_frames);
/* A batch of n_clips INDEPENDENT clips of clip_frames frames each, processed as one clip of n_clips * clip_frames
 * frames (clip c = frames c * clip_frames ...): the backbone calls see one long clip -- more crops per launch --
 * while every window is clamped to its own clip's frame numbers, so each clip's records equal what it gets alone up
 * to fp32 rounding (launch sizes pick tiles / split-K factors) (the reference runs clips one after another; ai_runner.py:493-520 has no cross-clip state). Records of frame
 * numbers that are a multiple of clip_frames belong to no clip (a clip's frame numbers run 1 .. clip_frames - 1)
 * and are to be ignored. */
int pa_clip_begin_batch(pa_engine* e, int32_t n_clips, int32_t clip_frames);

/* Crop + backbone for frames frame0 .. frame0+n-1 (0-based) of the clip:
 * square_crop(pad) -> BGR2RGB -> /255 -> ResNet-18 -> 1000-d feature per
 * (frame, fighter), stored in the engine's feature cache. Replaces the crop
 * read + `self.model.cnn2d` part of ai_runner.py:443-472 with each crop run
 * through the backbone once instead of once per window.
 * crops_rgb (optional, may be NULL): uint8[n,num_fighters,128,128,3] copy of the
 * model inputs (the `frames` entry of action_recognition's dict, :489).
 * status (optional): int32[n,num_fighters]. */
int pa_backbone_frames(pa_engine* e, const uint8_t* frames, int32_t n, int32_t height, int32_t width,
                       const double* boxes, int32_t frame0, uint8_t* crops_rgb, int32_t* status,
                       void* stream);

/* The two halves of pa_backbone_frames as separate calls, so that a host can pipeline them
 * on two streams: crop preprocessing of chunk k+1 (VALU/LDS-bound) overlaps the backbone of
 * chunk k (MFMA-bound) on the same GPU. `slot` (0 or 1) selects one of two model-input
 * buffers inside the engine. Ordering is the caller's job:
 *   - all pa_preprocess_frames / pa_square_crops calls are stream-ordered with each other
 *     (they share scratch memory);
 *   - pa_backbone_slot(slot) runs after the pa_preprocess_frames that filled `slot`;
 *   - a slot is not refilled before the pa_backbone_slot that read it has passed its first
 *     kernel (in practice: record an event after pa_backbone_slot and make the preprocess
 *     stream wait on it; playaid_core_amd/parallel.py does exactly this). */
int pa_preprocess_frames(pa_engine* e, const uint8_t* frames, int32_t n, int32_t height, int32_t width,
                         const double* boxes, int32_t slot, uint8_t* crops_rgb, int32_t* status,
                         void* stream);
int pa_backbone_slot(pa_engine* e, int32_t slot, int32_t n, int32_t frame0, void* stream);

/* ---- f2 / a1: ingest without hardware decode ----------------------------- */

/* This image has no hardware decode path (no rocDecode / rocJPEG / VCN libraries, no cv2), so "decode" is the
 * hand-over of raw BGR frames (cv2.VideoCapture.read's output, ai_runner.py:404-405). When those
 * frames sit in HOST memory, uploading them whole costs 6.2 MB per 1080p frame of PCIe for the ~0.84 MB the two
 * crops read. pa_upload_crop_windows copies only each crop's slice -- the square_crop region fighter.py:335-343
 * cuts, computed on the host with the device plan's own arithmetic -- into a packed device buffer: one kernel
 * whose waves read the slice rows straight out of the host frames over PCIe (frames_host must be pinned,
 * device-visible memory: hipHostMalloc / torch pin_memory; desc_host too) and writes one descriptor per crop; pa_preprocess_windows is
 * pa_preprocess_frames reading those windows (bit-identical crops). height / width stay the FRAME's. */
typedef struct pa_crop_window {
    int64_t offset;      /* first byte of the slice in the window buffer (16-byte aligned) */
    int32_t pitch;       /* bytes per slice row in the window buffer (row_bytes rounded up to 16) */
    int32_t rows;        /* slice rows */
    int64_t src_offset;  /* first byte of the slice in the host frame buffer */
    int32_t src_pitch;   /* bytes per frame row */
    int32_t row_bytes;   /* slice width * 3 */
} pa_crop_window;
int pa_upload_crop_windows(pa_engine* e, const uint8_t* frames_host, int32_t n, int32_t height, int32_t width,
                           const double* boxes_host, int32_t padding, uint8_t* windows_dev, size_t windows_capacity,
                           pa_crop_window* desc_host, pa_crop_window* desc_dev, size_t* bytes_used, void* stream);
int pa_preprocess_windows(pa_engine* e, const uint8_t* windows_dev, const pa_crop_window* desc_dev, int32_t n, int32_t height,
                          int32_t width, const double* boxes, int32_t slot, uint8_t* crops_rgb, int32_t* status, void* stream);

/* ---- a1 / f2: Motion-JPEG decode on the device ------------------------------ */

/* Replaces the per-frame decode of cv2.VideoCapture.read / cv2.imread (ai_runner.py:153,404-405,446;
 * manuscript.py:154-155) for Motion-JPEG streams and JPEG image sequences: n baseline JPEG files (SOF0 / 8-bit SOF1,
 * Huffman, one interleaved scan; 4:2:0, 4:2:2, 4:4:4 or grey; with or without restart markers) -> uint8 [n][height][width][3]
 * frames in HBM, BGR like OpenCV (rgb = 0) or RGB (rgb = 1). The arithmetic is libjpeg(-turbo)'s defaults, which OpenCV's
 * JPEG reader runs: integer IDCT (JDCT_ISLOW), fancy chroma up-sampling, jdcolor.c; oracle/jpeg.py::decode restates it and is
 * pinned byte for byte against the live libjpeg-turbo. (OpenCV's FFmpeg backend for .avi containers runs FFmpeg's own
 * decoder, which differs from libjpeg in the last bit; that arithmetic is not restated.)
 *
 * A handle owns the scratch for up to max_frames frames of max_height x max_width and max_bytes compressed bytes per
 * call. pa_mjpeg_decode parses the marker segments on the host, then ENQUEUES: one host -> device copy of the compressed
 * bytes (the range of data_host that covers all n frames; pinned memory makes it asynchronous) on a copy stream of the
 * handle's own, which `stream` waits for -- so the upload of one call runs under the decode passes of the call before it;
 * data_host must stay untouched until `stream` has passed this call -- and, on `stream`, the restart-marker
 * scan + byte un-stuffing, Huffman decoding (one lane per subsequence of the stream, wherever it falls: the
 * decoder states at the subsequence borders are found by self-synchronisation -- a speculative pass, then verify passes
 * until nothing changes -- so a stream needs no restart markers to decode in parallel; restart markers, where present,
 * are exact entry points), IDCT, up-sampling + colour conversion. spans_host: int64[n][2], frame f = bytes [spans[f][0], spans[f][1]) of data_host, in
 * any order (container chunk headers between frames are never looked at); every frame of a call has the same size and
 * sampling (tables and restart interval may change per frame).
 * status_dev (optional): int32[n] device, 0 or a bit set: 1 invalid Huffman code, 2 restart markers do not match the
 * header's interval, 4 coefficient index overflow, 8 the decoder states had not settled after the enqueued verify
 * passes (decode again after pa_mjpeg_set_sync_rounds(h, 0)) -- such a frame's pixels are undefined, nothing is written
 * out of bounds. Malformed or unsupported HEADERS fail the call with PA_ERR_INVALID_ARG and name the frame in
 * pa_mjpeg_last_error; nothing is enqueued then. */
typedef struct pa_mjpeg pa_mjpeg;
int pa_mjpeg_create(int32_t device, int32_t max_frames, int32_t max_height, int32_t max_width, size_t max_bytes, pa_mjpeg** out);
void pa_mjpeg_destroy(pa_mjpeg* h);
const char* pa_mjpeg_last_error(const pa_mjpeg* h);
/* Host only, no device: the marker segments of one JPEG file as pa_mjpeg_decode reads them. info8 = height, width,
 * components, max horizontal / vertical sampling factor (2,2 = 4:2:0), restart interval in MCUs (0 = none), byte offset
 * of the entropy-coded data, 0. PA_ERR_INVALID_ARG with the reason in `why` for files the decoder does not take
 * (progressive, arithmetic, 12-bit, multi-scan, truncated headers). */
int pa_mjpeg_probe(const uint8_t* data_host, size_t nbytes, int32_t* info8, char* why, size_t why_bytes);
/* Verify passes per call: 1..16 are enqueued without looking at their outcome (default 8; a pass over a frame whose
 * previous pass changed nothing returns at once; 1080p frames at quality 95 settle in three, and a frame that has not
 * settled is flagged with status bit 8); 0 = exact mode: passes are repeated until one changes
 * nothing, which synchronises the stream once per pass (long runs of identical blocks -- black bars -- re-synchronise
 * slowly and can need many). pa_mjpeg_last_sync_rounds: how many the last call ran. */
int pa_mjpeg_set_sync_rounds(pa_mjpeg* h, int32_t rounds);
int pa_mjpeg_last_sync_rounds(const pa_mjpeg* h);
/* Frame groups per call, 1..4 (default 2): a call's frames are decoded in that many groups, each on a stream of the
 * handle's own (uploads on a further one), joined to `stream` at the end -- a group's upload and its latency-bound late
 * verify passes run under the other group's passes. 1 = everything on `stream`: the setting for a caller that keeps
 * several decoders busy on several streams itself. The decoded frames are the same for every value. */
int pa_mjpeg_set_groups(pa_mjpeg* h, int32_t groups);
/* Diagnostics of the entropy decoder (scripts/mjpeg_rate.py): for pass kind m = 0 speculative, 1 verify, 2 final of the most
 * recent call, out8_host[2m] = shader-clock cycles the first wave of the first frame spent in its symbol loop,
 * out8_host[2m + 1] = (100 MHz wall ticks << 32) | symbols it walked; out8_host[8 + 2m] = cycles of those spent outside
 * the straight-line symbol loop (ring top-ups, restart markers, general symbols), out8_host[9 + 2m] = how often it
 * left that loop. out8_host holds 16 values. Synchronises the device. */
int pa_mjpeg_debug_counters(unsigned long long* out8_host);
int pa_mjpeg_decode(pa_mjpeg* h, const uint8_t* data_host, const int64_t* spans_host, int32_t n, int32_t height, int32_t width,
                    int32_t rgb, uint8_t* frames_dev, int32_t* status_dev, void* stream);

/* pa_backbone_frames for frames that are NOT consecutive in the clip (a resolution bucket of a
 * mixed-resolution stream, BASELINE.json configs[4]): frame_ids[n] (device, int32, 0-based)
 * says where each frame's features go in the cache. The call does not touch host-side clip
 * state, so it can be captured into a hipGraph and replayed with new buffer contents; tell
 * the engine which frames are cached with pa_clip_mark_ready (host ids) before pa_head_frames. */
int pa_backbone_frames_indexed(pa_engine* e, const uint8_t* frames, int32_t n, int32_t height, int32_t width,
                               const double* boxes, const int32_t* frame_ids, uint8_t* crops_rgb,
                               int32_t* status, void* stream);
int pa_clip_mark_ready(pa_engine* e, const int32_t* frame_ids_host, int32_t n);
/* pa_backbone_frames_indexed cannot validate device-resident ids on the host: a frame id outside
 * [0, clip_frames) writes nothing and is counted on the device. This call synchronises `stream`, returns
 * the count since the last call in *bad_frame_ids_host (and clears it); PA_ERR_CAPACITY when non-zero. */
int pa_device_errors(pa_engine* e, int32_t* bad_frame_ids_host, void* stream);

/* pa_backbone_frames for a clip whose crops are NOT cut from their own frame: crop (i, p) of clip frame
 * frame0 + i comes from frames[src_frame[i*num_fighters + p]] (device int32, values in [0, n_src)).
 * This is what clean_yolo_crops_for_fighter does for repaired gaps -- it re-cuts a missing crop from
 * VideoCapture position j, one decoded frame late, per fighter (ai_runner.py:404-418) -- in one pass.
 * A source index outside the buffer gives that crop status PA_CROP_BAD_FRAME (all-zero pixels). */
int pa_backbone_frames_src(pa_engine* e, const uint8_t* frames, int32_t n_src, int32_t height, int32_t width,
                           const double* boxes, const int32_t* src_frame, int32_t n, int32_t frame0,
                           uint8_t* crops_rgb, int32_t* status, void* stream);

/* Window gather + Conv1d/MLP head + log_softmax + argmax for frame numbers
 * frame_num_lo .. frame_num_hi-1 (1-based, as run_action_recognition iterates
 * range(1, max_frames), ai_runner.py:508). Replaces
 * action_sample_from_frame_middle_out (dataset_utils.py:109-138) and the head
 * half of ai_runner.py:472-477.
 * records: pa_record[count,num_fighters]; logp (optional): float32[count,num_fighters,A]. */
int pa_head_frames(pa_engine* e, int32_t frame_num_lo, int32_t frame_num_hi, pa_record* records,
                   float* logp, void* stream);

/* pa_clip_begin + pa_backbone_frames(all n frames) + pa_head_frames(1..n-1):
 * AIRunner.run_action_recognition (ai_runner.py:493-520) for an n-frame clip.
 * records: pa_record[n-1,num_fighters]; logp optional float32[n-1,num_fighters,A]. */
int pa_infer_clip(pa_engine* e, const uint8_t* frames, int32_t n, int32_t height, int32_t width,
                  const double* boxes, pa_record* records, float* logp, uint8_t* crops_rgb,
                  int32_t* status, void* stream);

/* Feature-cache access for frame-parallel sharding: copy
 * the cached rows of frames frame0..frame0+n-1 out of / into the engine.
 * feats: float32[n,num_fighters,PA_FEATURE_STRIDE]. */
int pa_features_export(pa_engine* e, int32_t frame0, int32_t n, float* feats, void* stream);
int pa_features_import(pa_engine* e, int32_t frame0, int32_t n, const float* feats, void* stream);

/* Damage HUD crops (AIRunner.run_damage_detection, ai_runner.py:556-571 +
 * damage_crop_to_percent :114): YoloCrop.crop_img (fighter.py:316-321) = image[y1:y2, x1:x2] followed by
 * imutils.resize(width=out_w) = cv2.resize(INTER_AREA) to (out_w, int(h * (out_w / float(w)))), for up to four pixel
 * rectangles per frame (rects_host int32[n_rects][4] = x1, y1, x2, y2 from YoloCrop.xyxy_pixels). frames
 * uint8[n,H,W,3] (device) -> out uint8[n,n_rects,out_h_cap,out_w,3] (device, channel order kept; rows beyond a
 * rectangle's height untouched); the heights are returned in out_h_host[n_rects]. The recogniser the reference
 * then calls (PaddleOCR) is an external model and is not part of this library. */
int pa_crop_resize_width(pa_engine* e, const uint8_t* frames, int32_t n, int32_t height, int32_t width, const int32_t* rects_host,
                         int32_t n_rects, int32_t out_w, uint8_t* out, int32_t out_h_cap, int32_t* out_h_host, void* stream);

/* ---- the alternative temporal model --------
 *
 * RNNActionDetector (playaid/models/rnn_action_detector.py:55-95): the same torchvision resnet18 with
 * fc = Linear(512, 300), then nn.LSTM(300, 512, num_layers=3), Linear(512,128) + ReLU, Linear(128, A),
 * log_softmax -- one output row per (window, frame). The backbone half reuses an engine's kernels:
 *
 * pa_backbone_windows: x float32[n_crops,3,128,128] (NCHW, values k/255, device) -> feats
 * float32[n_crops,PA_FEATURE_STRIDE] (device), the engine's resnet18 output per crop (fc included; an
 * engine built from a state dict whose fc rows 300..999 are zero yields the 300 features and zeros). */
int pa_backbone_windows(pa_engine* e, const float* x, int32_t n_crops, float* feats, void* stream);

/* The recurrent head. blob: int32 header {PA_LSTM_MAGIC, 1, input_dim, hidden_dim, num_layers, num_actions, 0, 0},
 * then float32 per layer l: weight_ih_l [4H, in_l], weight_hh_l [4H, H], bias_ih_l [4H], bias_hh_l [4H] (torch
 * gate order i, f, g, o; in_0 = input_dim, in_l = H), then action_decoder.0.weight [128, H], .0.bias [128],
 * .2.weight [A, 128], .2.bias [A]. hidden_dim % 8 == 0, <= 512; max_rows bounds seq_len * batch. */
typedef struct pa_lstm pa_lstm;
size_t pa_lstm_blob_bytes(int32_t input_dim, int32_t hidden_dim, int32_t num_layers, int32_t num_actions);
int pa_lstm_create(int32_t device, int32_t input_dim, int32_t hidden_dim, int32_t num_layers, int32_t num_actions,
                   int32_t max_rows, const void* blob_host, size_t blob_bytes, pa_lstm** out);
void pa_lstm_destroy(pa_lstm* h);
const char* pa_lstm_last_error(const pa_lstm* h);
/* x float32[seq_len, batch, ld] (device; the first input_dim of every ld-float row are read), zero initial
 * state -> logp float32[seq_len * batch, num_actions] (device). As in the reference (:88-90, an nn.LSTM
 * without batch_first fed [B, S, 300]) seq_len is the number of WINDOWS and batch (<= 16) the frames of a
 * window: the state runs from one window to the next. */
int pa_lstm_forward(pa_lstm* h, const float* x, int32_t ld, int32_t seq_len, int32_t batch, float* logp, void* stream);
/* A layer's time steps run in ONE launch (H / 4 workgroups, an ordinary launch since ABI 10) whose workgroups hand h(t) to each
 * other as tagged 8-byte granules and therefore must all be resident at once. The library checks the grid against HALF of what
 * the EMPTY device holds (occupancy query x CUs) and otherwise launches one kernel per time step; the query cannot see kernels of
 * OTHER streams, so do not overlap pa_lstm_forward with launches that fill the chip for more than a few milliseconds (the
 * engine's lanes, the chain's decode / detector stages). If part of the grid is nevertheless kept off the device, a granule that
 * does not arrive within 20 ms ends the launch: that call's logp rows are then NaN (never garbage) and the status below says so.
 * Call this after synchronising the stream of a pa_lstm_forward: PA_ERR_HIP ONCE if that happened (the handle launches one
 * kernel per time step from then on and later calls are valid), PA_OK otherwise. */
int pa_lstm_last_status(pa_lstm* h);

/* A conv-net given as a table (the ResNet-50 backbone of ResnetTransformerDetector,
 * playaid/models/resnet_transformer_detector.py:37), run on the engine's fp32 convolution kernels. Weights arrive
 * BatchNorm-folded: per convolution [cout][ky][kx][cin] at w_off and the bias [cout] at b_off (float offsets into
 * one blob); the 7x7/2 stem as [64][7 ky][8 px][4 ch] (kx >= 7 and channel 3 zero). Activations: `n_bufs` device
 * buffers of buf_floats_per_crop[b] floats per crop, NHWC with a zero border of `pad` pixels; a bordered buffer
 * must keep one geometry for the whole table. Layers run in table order. */
typedef struct pa_conv_desc {
    int32_t kind;               /* 0 convolution, 1 stem 7x7/2 + ReLU + max-pool 3x3/2 of the 128x128x3 input, 2 global average pool */
    int32_t cin, cout;          /* kind 0: cin % 32 == 0, cout % 64 == 0; kind 2: cin = channels */
    int32_t ksize, stride;      /* 1 | 3, 1 | 2 */
    int32_t in_hw;              /* spatial size of the input interior (square) */
    int32_t in_buf, in_pad;     /* in_pad >= (ksize - 1) / 2 */
    int32_t out_buf, out_pad;
    int32_t res_buf;            /* residual added before the ReLU (geometry of the output), or -1 */
    int32_t relu;
    int64_t w_off, b_off;
} pa_conv_desc;
typedef struct pa_convnet pa_convnet;
int pa_convnet_create(int32_t device, const pa_conv_desc* descs, int32_t n_descs, const int64_t* buf_floats_per_crop,
                      int32_t n_bufs, const float* weights_host, size_t n_weights, int32_t max_crops, pa_convnet** out);
/* The same with the convolutions' arithmetic chosen (ABI 11): PA_DTYPE_F32 (= pa_convnet_create) or PA_DTYPE_EMULATED_F32 -- every
 * convolution that is not in Winograd form and has enough 128-pixel tiles at max_crops to fill half the chip runs on the emulated-fp32
 * persistent GEMM (csrc/psgemm.hip), the rest on the exact kernels. Never the default. */
int pa_convnet_create_dtype(int32_t device, const pa_conv_desc* descs, int32_t n_descs, const int64_t* buf_floats_per_crop,
                            int32_t n_bufs, const float* weights_host, size_t n_weights, int32_t max_crops, int32_t compute_dtype,
                            pa_convnet** out);
void pa_convnet_destroy(pa_convnet* h);
const char* pa_convnet_last_error(const pa_convnet* h);
/* x float32[n,3,128,128] (NCHW, device) -> out: the last layer's output buffer, float32[n, out_floats_per_crop]. */
int pa_convnet_forward(pa_convnet* h, const float* x, int32_t n, float* out, int32_t out_floats_per_crop, void* stream);

/* One stride-1 3x3 convolution (padding 1) on the Winograd F(2x2, 3x3) kernel of the fp32 convolution stack
 * (csrc/wino.hip) -- the operator the engine's ResNet-18 / the detector's Bottlenecks run their stride-1 3x3 layers on,
 * exposed for parity tests and measurements of single layers. x: float32[n][height + 2][width + 2][in_px_stride]
 * (device, zero border of one pixel, the first `cin` channels of every pixel are read); out:
 * float32[n][height + 2 out_pad][width + 2 out_pad][out_px_stride] (interior written, first `cout` channels);
 * residual: addressed like out, or NULL; act: 0 none, 1 ReLU, 2 SiLU; res_after: 1 = the residual is added after the
 * activation. height, width multiples of 4; cin % 8 == 0; cout % 32 == 0. Filters: pa_wino_transform_weights turns
 * BatchNorm-folded [cout][ky][kx][cin] (host) into the kernel's layout (host, pa_wino_weight_floats floats), which
 * the caller uploads. `bn` = output channels per workgroup (64 | 32), part of that layout and of the launch:
 * pa_wino_channels_per_workgroup gives the value the engine would pick for a layer of `cout` channels launched over
 * `sub_blocks` = n * height / 4 * width / 4 sub-blocks (32 when 64-channel workgroups would not fill the chip). Enqueue only. */
size_t pa_wino_weight_floats(int32_t cin, int32_t cout);
int pa_wino_channels_per_workgroup(int32_t cout, int64_t sub_blocks);
int pa_wino_transform_weights(const float* w_host, int32_t cin, int32_t cout, int32_t bn, float* ug_host);
int pa_wino_conv3x3(const float* x, const float* ug, const float* bias, const float* residual, float* out, int32_t n,
                    int32_t height, int32_t width, int32_t cin, int32_t cout, int32_t bn, int32_t in_px_stride,
                    int32_t out_px_stride, int32_t out_pad, int32_t act, int32_t res_after, void* stream);
/* The same launch with split-K scratch (ABI 10): where the layer's tiles alone would leave CUs without a workgroup (few pixels,
 * many channels: ResNet-18's 8 x 8 and 4 x 4 maps) the launcher lets 2 / 4 / 8 workgroups share a tile, each summing a run of
 * input-channel chunks; their partial output tiles meet in `slab` and the last workgroup to arrive adds them IN SPLIT ORDER
 * (results do not depend on arrival order) and runs the epilogue. slab: device scratch of slab_floats floats (16 MB covers
 * every launch: 256 x 512 x 32 floats); tickets: int32[n_tickets] on the device, ZERO before the first launch -- the kernel
 * leaves them zero; a launch with more tiles than tickets, or too small a slab, runs unsplit. Launches that share the scratch
 * belong on one stream. */
int pa_wino_conv3x3_splitk(const float* x, const float* ug, const float* bias, const float* residual, float* out, int32_t n,
                           int32_t height, int32_t width, int32_t cin, int32_t cout, int32_t bn, int32_t in_px_stride, int32_t out_px_stride,
                           int32_t out_pad, int32_t act, int32_t res_after, float* slab, size_t slab_floats, int32_t* tickets,
                           int32_t n_tickets, void* stream);

/* One convolution (1x1 or 3x3, stride 1 or 2, "same" padding) + bias + activation (+ residual) on the persistent implicit-GEMM
 * kernels -- the operator the detection network's layers (ai_runner.py:191-224's YOLOv5s) and, under PA_DTYPE_EMULATED_F32, the
 * ResNet-18's 3x3 convolutions (cnn_action_detector.py:16,32) run on -- exposed for parity tests and per-layer measurements.
 * x: float32[n][height + 2 in_pad][width + 2 in_pad][in_px_stride] (device, zero border, the first cin channels of a pixel are
 * read; in_pad >= (ksize - 1) / 2); out: float32[n][oh + 2 out_pad][ow + 2 out_pad][out_px_stride], oh = height / stride
 * (interior written, first cout channels); residual: addressed like out (it may BE out), or NULL; act: 0 none, 1 ReLU, 2 SiLU;
 * res_after: 1 = the residual is added after the activation. cin, cout multiples of 32; pixel strides multiples of 4 floats and
 * x, w, out, residual 16-byte aligned (PA_ERR_INVALID_ARG otherwise: the kernels move 16-byte units). Weights: pa_conv_pack_weights turns
 * BatchNorm-folded [cout][ky][kx][cin] fp32 (host) into what the kernel of `compute_dtype` reads (host, pa_conv_weight_bytes
 * bytes; the caller uploads it): PA_DTYPE_F32 = the same fp32 values (csrc/pigemm.hip; no residual), PA_DTYPE_EMULATED_F32 = three
 * bf16 slices per weight in the LDS stage-image order of csrc/psgemm.hip, whose tile width depends on has_residual. Enqueue only. */
size_t pa_conv_weight_bytes(int32_t cin, int32_t cout, int32_t ksize, int32_t compute_dtype, int32_t has_residual);
int pa_conv_pack_weights(const float* w_host, int32_t cin, int32_t cout, int32_t ksize, int32_t compute_dtype, int32_t has_residual,
                         void* out_host);
int pa_conv2d(const float* x, const void* w, const float* bias, const float* residual, float* out, int32_t n, int32_t height,
              int32_t width, int32_t cin, int32_t cout, int32_t ksize, int32_t stride, int32_t in_pad, int32_t in_px_stride,
              int32_t out_px_stride, int32_t out_pad, int32_t act, int32_t res_after, int32_t compute_dtype, void* stream);

/* Head of ResnetTransformerDetector (resnet_transformer_detector.py:41-93,141): Linear(in_dim, hidden_dim), the
 * enc_dim-value time encoding of the frame slot appended (d_model = hidden_dim + enc_dim, 32 per head),
 * num_layers post-norm nn.TransformerEncoderLayer (ReLU feed-forward of ff_dim), Linear(d_model, num_actions),
 * log_softmax. blob: int32[16] header {PA_ENCODER_MAGIC, 1, in_dim, hidden_dim, slots, enc_dim, num_heads,
 * num_layers, ff_dim, num_actions, 0...}, then float32: resnet_ffn.weight [hidden, in_dim], .bias, freq_encoding
 * [slots, enc_dim], per layer self_attn.in_proj_weight [3D, D], in_proj_bias, out_proj.weight [D, D], .bias,
 * linear1.weight [ff, D], .bias, linear2.weight [D, ff], .bias, norm1.weight, .bias, norm2.weight, .bias, then
 * classifier.weight [A, D], .bias. */
typedef struct pa_encoder pa_encoder;
size_t pa_encoder_blob_bytes(int32_t in_dim, int32_t hidden_dim, int32_t slots, int32_t enc_dim, int32_t num_layers,
                             int32_t ff_dim, int32_t num_actions);
int pa_encoder_create(int32_t device, int32_t in_dim, int32_t hidden_dim, int32_t slots, int32_t enc_dim, int32_t num_heads,
                      int32_t num_layers, int32_t ff_dim, int32_t num_actions, int32_t max_rows, const void* blob_host,
                      size_t blob_bytes, pa_encoder** out);
void pa_encoder_destroy(pa_encoder* h);
const char* pa_encoder_last_error(const pa_encoder* h);
/* feats float32[seq_len * batch, ld] (device, row r = l * batch + n; the first in_dim values of a row are read) ->
 * logp float32[seq_len * batch, num_actions]. As in the reference (an encoder without batch_first fed
 * [B, S, 256], :82-84) seq_len is the number of WINDOWS and batch = slots the frames of a window. */
int pa_encoder_forward(pa_encoder* h, const float* feats, int32_t ld, int32_t seq_len, int32_t batch, float* logp, void* stream);

/* ---- measurement -------------------------------------------------------- */

/* When enabled, every kernel launch is bracketed by HIP events recorded on the
 * stream it is launched on. pa_profile_read synchronises that stream, returns
 * one row per kernel family and clears the log. */
int pa_profile_enable(pa_engine* e, int32_t on);
int pa_profile_read(pa_engine* e, pa_kernel_stat* stats, int32_t max_stats, int32_t* n_stats);

/* quality 1..100: every 128 x 128 crop the engine cuts (pa_square_crops, pa_backbone_frames*, pa_infer_clip,
 * pa_preprocess_*) additionally goes through the pixel arithmetic of a baseline JPEG write + read at that quality
 * (4:2:0, integer DCT), as the reference's cv2.imwrite / cv2.imread of every crop does (ai_runner.py:420,446; OpenCV's
 * default quality is 95) -- the returned crops and the model input then carry the codec's loss. 0 (the default)
 * switches it off: crops are the exact resampler output. Crop IMAGES handed to pa_runner_inputs /
 * pa_backbone_crop_images are taken as already decoded. */
int pa_set_crop_jpeg_quality(pa_engine* e, int32_t quality);

/* Keeps `stream` busy for about `microseconds` (one spinning thread; 0..100000). A probe, not a workload: two
 * HIP streams that the runtime multiplexed onto one hardware queue run such kernels strictly in turn, two that sit
 * on different queues side by side (playaid_core_amd/parallel.py picks the streams of its lanes with it). */
int pa_stream_spin(pa_engine* e, int32_t microseconds, void* stream);

/* A gate the HOST opens: pa_stream_gate enqueues a one-thread kernel that keeps `stream` busy until pa_stream_gate_open is
 * called (a store to coherent pinned memory the kernel polls), or max_microseconds (1..100000) have passed -- the bound
 * makes a gate nobody opens cost that long instead of hanging the queue. One gate per engine at a time.
 * playaid_core_amd/parallel.py (ClipLanes) enqueues the first clip of every lane behind an event recorded after one gate
 * and opens it when the last of them is enqueued, so the lanes start together whatever the host's submit times were. */
int pa_stream_gate(pa_engine* e, int32_t max_microseconds, void* stream);
int pa_stream_gate_open(pa_engine* e);

/* Blocks until all work enqueued on `stream` is done (hipStreamSynchronize). */
int pa_stream_sync(pa_engine* e, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PLAYAID_HIP_H */
