// Internal declarations shared by the HIP translation units and the C-ABI host
// code. Not part of the public ABI (include/playaid_hip.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pa {

// SiLU x * sigmoid(x) of the conv epilogues (the detection network): v_exp_f32 + v_rcp_f32 (1 ulp each; the exponent's
// pre-multiply adds |x| * 6e-8 relative) instead of expf + an IEEE division -- ~6 instead of ~30 vector instructions per
// value, which for a 64-deep 1x1 convolution was as much issue time as the tile's matrix instructions.
__device__ __forceinline__ float silu_fast(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}

// ---------------------------------------------------------------------------
// implicit-GEMM engine (igemm.hip)
// ---------------------------------------------------------------------------
// out[m][n] = act( sum_{tap,kc} A(m,tap,kc) * W[n][tap*chunk + kc] + bias[n] (+ residual[m][n]) )
//
// A(m,tap,kc) is addressed in one of two ways:
//  * conv mode  (gather == nullptr): m -> (img, oy, ox); tap -> (ky, kx);
//      A = act + img*in_img_stride + (oy*stride + ky + off_y)*in_row_stride
//              + (ox*stride + kx + off_x)*in_px_stride + kc
//    Activations are NHWC with an explicit zero border, so no bounds checks.
//  * gather mode: row = gather[m*taps + tap]; A = act + row*in_px_stride + kc
//    (row < 0 -> zeros). Used for the temporal Conv1d over cached features.
struct GemmParams {
    const float* act;
    const float* wgt;       // [N][ktot], K contiguous
    const float* bias;      // [N] or nullptr
    const float* residual;  // same addressing as out, or nullptr
    float* out;
    float* slab;            // split-K partial sums [splitk][M][N]
    const int32_t* gather;
    int32_t M, N, ktot;
    int32_t taps, kw_taps, chunk;  // ktot == taps*chunk, chunk % 32 == 0
    int32_t howo, wo;              // output pixels per image, output width
    int32_t howo_shift, wo_shift;  // log2 of the above when both are powers of two, else -1 (set by the launcher)
    int32_t in_img_stride, in_row_stride, in_px_stride, stride, off_y, off_x;
    int32_t out_img_stride, out_row_stride, out_px_stride, out_pad;
    int32_t relu;           // activation of the epilogue: 0 none, 1 ReLU, 2 SiLU (x * sigmoid(x))
    int32_t res_after;      // 1: the residual is added AFTER the activation (YOLOv5's Bottleneck), 0: before (ResNet)
    // optional second im2col source appended to K (k2_steps * 32 values, 1x1 taps): act2 addressed as
    //   act2 + img*in2_img_stride + (oy*stride2 + off2)*in2_row_stride + (ox*stride2 + off2)*in2_px_stride + kc
    const float* act2;
    int32_t k2_steps, in2_img_stride, in2_row_stride, in2_px_stride, stride2, off2;
    int32_t splitk, ksteps_per_split;
    int32_t defer_reduce;   // 1: leave the split-K slabs to the consumer (the head MLP sums them itself); the
                            //    launcher writes the split count it actually used to *splitk_used
    int32_t* splitk_used;
    int32_t skip_w;         // 1: k % 4 == 3 always meets a zero weight (stem channel pad): those MFMAs are skipped
    int32_t tiles_m, tiles_n;
    // persistent GEMM (pigemm.hip), filled by its launcher: workgroups per XCD, output rows per image, ceil(2^32 / howo) and
    // ceil(2^32 / wo) for the scalar divisions, how often 31 consecutive pixels can wrap a row / that many rows an image
    // (beside tiles_m / tiles_n: the kernel's first scalar loads fetch them together)
    int32_t pg_per, pg_ho, pg_nwx, pg_nwy;
    uint32_t pg_magic_howo, pg_magic_wo;
    // persistent GEMM only: a second, nearest-neighbour x2 up-sampled copy of the output (first channel of the slice's padded pixel
    // (0, 0) of image 0; strides in floats of the UP-SAMPLED buffer; nullptr = none)
    float* up_out;
    int32_t up_img_stride, up_row_stride, up_px_stride, up_pad;
    // patch-resident 3x3 kernel (patchconv.hip), filled by its launcher: pixels per LDS patch buffer,
    // input row pitch and image size in pixels, tile -> first patch pixel, pixels in the whole buffer
    int32_t patch_slots, patch_pitch, img_px, tiles_per_img, p0_img, p0_row, total_px;
    int32_t swz_a, magic_pitch, magic_img, img_px_patch;  // chunk-swizzle key of a patch pixel (see patchconv.hip)
    int32_t xcd_m, xcd_n;  // the 8 XCDs as an xcd_m x xcd_n grid over (pixel tiles, channel tiles); 0 = contiguous runs of tiles
    // blocked form of the patch-resident kernel (maps whose width is not 32 / 16 / 8 / 4: the detection network's): output
    // pixel m = (block * blk_rows + r) * blk_cols + c, blocks row-major over each image; a tile's patch = the
    // (blk_rows + 2) x (blk_cols + 2) input rectangles of its blocks back to back. Set by launch_conv3x3_patch_blocked.
    int32_t blk_rows, blk_cols, blk_shift_c, blk_shift_px, blk_per_row, blk_per_img, n_blocks, src_pitch;
    // igemm_bf16.hip, stride-2 3x3 openers: a second 1x1 product off the SAME im2col rows -- the centre tap's rows are the
    // pixels the block's 1x1/2 downsample branch reads -- wgt2 = [N][chunk] bf16, out2 addressed like out (no bias, no
    // activation); nullptr = off
    const float* wgt2;
    float* out2;
    unsigned long long* clk;  // ablation builds only: in-kernel clock stamps
};

// BM x BN (x BK; 32 unless named)
enum GemmTile { TILE_128x128 = 0, TILE_128x64 = 1, TILE_64x64 = 2, TILE_128x64_K64 = 3, TILE_64x64_K64 = 4,
                TILE_256x128 = 5 /* igemm_bf16.hip only */ };

hipError_t launch_igemm(const GemmParams& p, GemmTile tile, hipStream_t s);
// persistent form of the same engine for short tiles (pigemm.hip): conv mode, no residual / second source / split-K;
// bm = 128 | 64, 64 output channels per tile; results equal to launch_igemm up to where the bias enters the sum
hipError_t launch_pgemm(const GemmParams& p, int bm, hipStream_t s);
// "emulated fp32" form of the persistent GEMM (psgemm.hip): fp32 activations, the weights as three bf16 slices in the kernel's
// stage-image layout (psgemm_pack_weights, psgemm_weight_elems of them), six bf16 matrix instructions per fp32 product, fp32
// accumulate. Conv mode; bias, ReLU / SiLU; out_floats = floats from p.out to the end of its buffer (the store descriptor's bounds).
int psgemm_pick_bn(int N, int residual);          // channels per workgroup (128 | 64 | 32; with a residual 64 at most): part of the weight layout
size_t psgemm_weight_elems(int N, int ktot, int residual);
void psgemm_pack_weights(const float* w, int N, int ktot, int residual, unsigned short* out);
hipError_t launch_psgemm(const GemmParams& p, const unsigned short* wsp, size_t out_floats, size_t up_floats, hipStream_t s);   // (up_floats: the same for p.up_out, 0 without one)
// bf16 activations / weights (uint16_t storage behind the float* fields, every count in elements),
// f32 accumulate on v_mfma_f32_32x32x16_bf16; conv mode only (igemm_bf16.hip)
hipError_t launch_igemm_bf16(const GemmParams& p, GemmTile tile, hipStream_t s);

// stride-1 3x3 convolution on bf16 with the activation patch resident in LDS (patchconv_bf16.hip); no second source
hipError_t launch_conv3x3_bf16_patch(const GemmParams& p, hipStream_t s);

// Persistent direct 7x7/2 stem (stem.hip): x = [crops][134][134][4] fp32 (3-pixel zero border,
// channel 3 = 0), wgt = [64][7 ky][8 px][4 ch] (the igemm stem layout), out = [crops][66][66][64].
struct StemParams {
    const float* x;
    const float* wgt;
    const float* bias;
    float* out;     // fp32, or bf16 (uint16_t storage) when out_bf16 != 0
    int32_t tiles;  // crops * 32 (one tile = two output rows of one crop)
    int32_t out_bf16;
    unsigned long long* clk;  // stamp builds only
};
hipError_t launch_stem7x7(const StemParams& p, hipStream_t s);
// Stem + BatchNorm + ReLU + 3x3/2 max-pool in one persistent kernel (stem_pool.hip): x as above (fp32) or bf16
// [crops][134][134][4]; wgt fp32 or bf16 [64][224]; out = pooled map [crops][34][34][64], fp32 or bf16.
struct StemPoolParams {
    const void* x;
    const void* wgt;
    const float* bias;
    void* out;
    int32_t crops;
    int32_t run;       // row pairs per run (set by the launcher)
    int32_t in_bf16;   // x and wgt are bf16: multiply on the bf16 matrix cores
    int32_t out_bf16;
};
hipError_t launch_stem_pool(const StemPoolParams& p, hipStream_t s);
hipError_t launch_splitk_reduce(const GemmParams& p, hipStream_t s);
// stride-1 3x3 convolution with the input patch resident in LDS across the nine taps (patchconv.hip); bm = 128 | 64
hipError_t launch_conv3x3_patch(const GemmParams& p, int bm, hipStream_t s);
// the same kernel over rectangular blocks of output pixels (8 x 16, 8 x 8 or 4 x 4, whichever divides the map): any map whose
// sides are multiples of 4; in_pad == 1; epilogue: bias, ReLU / SiLU, residual before or after the activation
hipError_t launch_conv3x3_patch_blocked(const GemmParams& p, hipStream_t s);

// stride-1 3x3 convolution as Winograd F(2x2, 3x3) on the fp32 matrix cores (wino.hip): zero-bordered NHWC input (border 1),
// any map whose sides are multiples of 4, cin % 8 == 0, cout % 32 == 0; epilogue: bias, ReLU / SiLU, residual before or
// after the activation. The input and output may be channel slices of wider buffers (px strides >= cin / cout).
struct WinoParams {
    const float* act;       // first channel of padded pixel (0, 0) of image 0
    const float* wgt;       // wino_transform_weights' image of the filters
    const float* bias;      // [cout] or nullptr
    const float* residual;  // addressed like out, or nullptr
    float* out;             // first channel of padded pixel (0, 0) of image 0
    int32_t n_img, height, width, cin, cout;
    int32_t bn;             // output channels per workgroup the filters were laid out for (wino_pick_bn: 64 | 32)
    int32_t in_px_stride, in_row_stride, in_img_stride;      // floats
    int32_t out_px_stride, out_row_stride, out_img_stride, out_pad;
    int32_t relu, res_after;  // as GemmParams
    int32_t n_sb, sb_per_row, sb_per_img, tiles_n;  // set by the launcher
    unsigned long long* clk;  // stamp launches only (PA_WINO_ABL=8)
    // split K (optional; the launcher turns it on for layers whose tiles do not fill the chip): scratch for the partial output
    // tiles -- slab_floats floats -- and one ticket per tile, tickets_cap of them, ZERO before the first launch (the kernel
    // leaves them zero); ksplit is set by the launcher
    float* slab;
    size_t slab_floats;
    int32_t* tickets;
    int32_t tickets_cap, ksplit;
    int32_t tiles_m, xcd_gm, xcd_gn;   // set by the launcher: pixel tiles; the 8 XCDs as a gm x gn grid over (pixel, channel) tiles, 0 = launch order
};
size_t wino_weight_floats(int cin, int cout);
// output channels per workgroup = per stage image of the filter layout (64 | 32); cin_split > 0: the caller launches with split-K
// scratch (WinoParams::slab, tickets) and the layer has that many input channels
int wino_pick_bn(int cout, long long n_sb, int cin_split = 0);
// w [cout][ky][kx][cin] (host) -> ug [wino_weight_floats] (host)
void wino_transform_weights(const float* w, int cin, int cout, int bn, float* ug);
hipError_t launch_wino3x3(const WinoParams& p, hipStream_t s);

// ---------------------------------------------------------------------------
// crop preprocessing (preprocess.hip)
// ---------------------------------------------------------------------------
#define PA_KSIZE_MAX 15
// Stage buffers of crop_fused_kernel; + 2176 B of INTER_AREA tables = 80000 B: two 512-thread
// workgroups per CU (16 waves). Measured ladder of this choice (64 x 1080p step, kernel alone):
// 256 threads / 50944 B (3 per CU) 0.189 ms -> 512 threads / 46592 B 0.169 -> 512 threads / 77824 B
// 0.144: eight waves per band halve a workgroup's lifetime, and the larger budget lets most crops use
// 8-row sub-bands (the 7-tap vertical support makes 4-row sub-bands recompute 1.7x of the horizontal
// pass, 8-row ones 1.35x).
#define PA_FUSED_LDS_BYTES 77824

struct CropPlan {
    int32_t status;
    int32_t frame;
    int32_t sx0, sy0, sw, sh;  // slice of the frame
    int32_t d;                 // square side
    int32_t rw, rh;            // size after ImageOps.contain
    int32_t px, py;            // paste offset inside the d x d canvas
    int32_t need_h, need_v;    // bicubic passes
    int32_t ksize_h, ksize_v;
    int32_t out_h;             // INTER_AREA destination height (width is 128)
    int32_t area_mode;         // 0 copy, 1 fast 2x2, 2 fast integer, 3 general
    int32_t iscale_x, iscale_y;
    int32_t fused_rb;          // output rows per LDS sub-band of the fused kernel (8/4/2/1), 0 = multi-kernel fallback
    int32_t pad_;
    double scale_x, scale_y;
    // Pillow coefficient tables of the two bicubic passes ([out][2 + PA_KSIZE_MAX]): the engine's cache of the
    // (2 * (d / 2) + 2 * padding -> d) pairs, or this crop's own rows of PreprocParams::coef for any other pair
    const int32_t* coef_h;
    const int32_t* coef_v;
};

// One entry of cv::computeResizeAreaTab per destination column / row of a crop (plan kernel -> fused kernel)
struct AreaTabPacked {
    uint32_t bits;  // s_first | n_mid << 16 | has_first << 24 | has_last << 25
    float a_first, a_mid, a_last;
};

struct CropWindow {  // == pa_crop_window of the public header
    int64_t offset;      // first byte of the crop's slice inside the window buffer
    int32_t pitch;       // bytes per slice row in the window buffer (row_bytes rounded up to 16)
    int32_t rows;        // slice rows
    int64_t src_offset;  // first byte of the slice inside the host frame buffer
    int32_t src_pitch;   // bytes per frame row
    int32_t row_bytes;   // slice width * 3
};
hipError_t launch_slice_upload(const uint8_t* frames_host, long long frames_bytes, const CropWindow* desc, uint8_t* windows, int ncrops,
                               hipStream_t s);

struct PreprocParams {
    const uint8_t* frames;  // [n_src][H][W][3]; with `windows`: the packed window buffer
    const CropWindow* windows;  // [ncrops] or nullptr: every crop's slice was uploaded on its own
    const double* boxes;    // [ncrops][4]
    const int32_t* src_frame;  // [ncrops] frame each crop is cut from, or nullptr = crop / fighters
    int32_t n_src;             // frames in the buffer (bound of src_frame)
    int32_t n_frames, height, width, fighters, padding, swap_rb;
    CropPlan* plans;        // [ncrops]
    int32_t* coef;          // [ncrops][2][maxdim][1 + 1 + PA_KSIZE_MAX]: xmin, count, kk[]
    int32_t coef_dim;       // maxdim (entries per axis per crop)
    const int32_t* coef_cache;  // tables of the pairs (2 * (d / 2) + 2 * coef_cache_pad -> d), d = 1 .. coef_cache_dmax, table d at row d * (d - 1) / 2
    int32_t coef_cache_pad, coef_cache_dmax;
    AreaTabPacked* area_tabs;   // [ncrops][2][128]: INTER_AREA tables of the 128 destination columns, then rows
    uint8_t* t1;            // [ncrops][t_stride] horizontally resampled slice
    uint8_t* t2;            // [ncrops][t_stride] fully resampled slice
    size_t t_stride;
    uint8_t* crops_u8;      // [ncrops][128][128][3] or nullptr
    float* crops_f32;       // [ncrops][134][134][4] zero-bordered, or nullptr
    int32_t crops_f32_is_bf16;  // the model input is stored as bf16 [ncrops][134][134][4] instead (bf16 conv path)
    int32_t* status;        // [ncrops] or nullptr
    int32_t* fallback_count;  // [1] number of crops routed to the multi-kernel fallback (zeroed per call)
    int32_t* fallback_list;   // [ncrops] their indices
    int32_t fused_lds;      // LDS budget of the fused kernel (set by the launcher; 0 forces the fallback)
    int32_t ablate;         // timing experiments: skip stages of the fused kernel (results wrong when != 0)
    uint8_t* dbg;           // debug builds only (PA_DEBUG_DUMP)
    int32_t dbg_crop, dbg_row;
};

hipError_t launch_preprocess(const PreprocParams& p, hipStream_t s);
// fills the engine's coefficient cache for one padding value (once per engine, or when the padding changes)
size_t coef_cache_ints(int dmax);
hipError_t launch_build_coef_cache(int32_t* cache, int padding, int dmax, hipStream_t s);

// crop images of any size -> runner inputs (ai_runner.py:446-459), see runner_input_kernel
struct CropImageDesc {  // == pa_crop_image of the public header
    int64_t offset;     // byte offset of the image inside `images`
    int32_t height, width;
};
struct RunnerInParams {
    const uint8_t* images;      // concatenated uint8 [h][w][3] images
    long long images_bytes;
    const CropImageDesc* desc;  // [n] (device)
    int32_t n, swap_rb, max_h, max_w;
    uint8_t* t1;                // scratch, t_stride bytes per image each
    uint8_t* t2;
    size_t t_stride;
    uint8_t* inputs_u8;         // [n][128][128][3] or nullptr
    float* inputs_f32;          // [n][134][134][4] zero-bordered model input (fp32, or bf16 storage) or nullptr
    int32_t inputs_f32_is_bf16;
    int32_t* status;            // [n] PA_CROP_* or nullptr
};
hipError_t launch_runner_inputs(const RunnerInParams& q, hipStream_t s);

// baseline-JPEG write + read of the 128 x 128 crops, in place (jpeg.hip)
struct JpegParams {
    uint8_t* crops_u8;   // [ncrops][128][128][3]
    const int32_t* qtab; // [2][64] quantisation tables (luminance, chrominance) in natural order, device
    void* x0;            // model input [ncrops][134][134][4] (fp32 or bf16) rewritten from the new pixels, or nullptr
    int32_t x0_bf16;
    int32_t bgr;         // memory order of the crops: 1 = B, G, R (cv2), 0 = R, G, B
};
hipError_t launch_jpeg_roundtrip(const JpegParams& p, int ncrops, hipStream_t s);

// label repair on the detection table (detect.hip::clean_labels_kernel)
struct CleanParams {
    const float* dets;      // [n_labels][max_det][6] label rows, label-file order
    const int32_t* counts;  // [n_labels]
    int32_t n_labels, max_det, n_decoded, fighters;
    int32_t class_ids[4];
    double* lab;            // [n_labels][F][6] repaired label row of each fighter (cls < 0: none)
    int32_t* pixel_frame;   // [n_labels][F]
    double* pixel_box;      // [n_labels][F][4]
    int32_t* crop_kind;     // [n_labels][F]
    float* crop_row;        // [n_labels][F][6]
    int32_t* info;          // [4]: max_frames, error code, error frame, duplicates resolved
    double* g6v;            // scratch [n_labels][max_det][6]: every row value through the label file's '%g' (filled by the launcher's first kernel)
};
hipError_t launch_clean_labels(const CleanParams& p, hipStream_t s);
struct CropImageDesc;
// what the crop hand-off needs from the repaired table, and the clip's crop-image descriptors (detect.hip; pa_detector_plan*)
hipError_t launch_detector_plan(const int32_t* pixel_frame, const double* pixel_box, const int32_t* crop_kind, const int32_t* info4, int n_labels, int F,
                                int32_t* det_index, int32_t* src_own, int32_t* rep_entry, double* rep_boxes, int32_t* rep_src, int32_t* words5,
                                hipStream_t s);
hipError_t launch_detector_desc(CropImageDesc* desc, const int32_t* crop_kind, int n_entries, int F, int step_frames, long long region,
                                const int32_t* rep_entry, int n_rep, long long rep_base, hipStream_t s);

// YOLOv5 save_one_box crops + their 4:4:4 JPEG write / read (savebox.hip)
struct SaveBoxRect { int32_t x1, y1, w, h; };
struct SaveBoxParams {
    const uint8_t* frames;      // [n][height][width][3] BGR
    int32_t height, width, fighters, n_entries;  // n_entries = frames * fighters
    const float* dets;          // [n][max_det][6] label rows (cls cx cy w h conf, normalised), label-file order
    const int32_t* counts;      // [n]
    int32_t max_det;
    const int32_t* det_index;   // [n_entries] detection of each (frame, fighter), -1 = none; nullptr = first of the class
    const int32_t* src_frame;   // [n_entries] frame (of n_src) the pixels are cut from; nullptr = the entry's own frame
    int32_t n_src;
    int32_t class_ids[4];
    float gain, pad;
    uint8_t* images;            // packed output
    unsigned long long capacity;
    CropImageDesc* desc;        // [n_entries]
    SaveBoxRect* rects;         // [n_entries] scratch
    int32_t* overflow;          // device counter: entries that did not fit `capacity`
    int32_t quality;            // 0 = no JPEG write / read
    int32_t qtab[128];          // quantisation tables (luminance, chrominance), natural order
};
hipError_t launch_save_one_box(const SaveBoxParams& p, hipStream_t s);

// crop_img + imutils.resize(width) of up to four pixel rectangles per frame (the damage HUD crops), see rect_resize_kernel
struct RectResizeParams {
    const uint8_t* frames;       // [n][height][width][3]
    int32_t height, width;
    int32_t n_rects;
    int32_t x1[4], y1[4], x2[4], y2[4];  // numpy slice image[y1:y2, x1:x2]
    int32_t out_w, out_h[4];     // destination width, int(h * (out_w / float(w))) per rectangle
    int32_t out_h_cap;           // rows per rectangle in `out`
    uint8_t* out;                // [n][n_rects][out_h_cap][out_w][3]
};
hipError_t launch_rect_resize(const RectResizeParams& q, int n_frames, hipStream_t s);
// per-device one-time setup of the crop stage (dynamic-LDS attribute); call with the device current
hipError_t preprocess_init_device();
// log rows [n][9] (pos_x,pos_y,cam xyz,target xyz,fov deg) -> normalised boxes [n][4]
hipError_t launch_project_boxes(const double* log, double* boxes, int32_t n, hipStream_t s);

// ---------------------------------------------------------------------------
// detection post-processing (detect.hip): head rows -> class filter, confidence, NMS, max_det, image-space boxes
// ---------------------------------------------------------------------------
struct DetectParams {
    const float* pred;   // [n_frames][rows][5 + nc]: cx cy w h (network-input pixels), objectness, class scores
    int32_t n_frames, rows, nc, max_det;
    float conf_thres, iou_thres;
    uint32_t class_mask;
    float pad_x, pad_y, gain;  // letterbox of the network input inside the image (scale_boxes)
    float img_w, img_h;
    float* dets;         // [n_frames][max_det][6]: cls cx cy w h conf, label-file order (lowest confidence first)
    int32_t* counts;     // [n_frames]
};
hipError_t launch_detect_nms(const DetectParams& q, hipStream_t s);

// ---------------------------------------------------------------------------
// small kernels (misc.hip)
// ---------------------------------------------------------------------------
// x[n][3][128][128] f32 -> zero-bordered [n][134][134][4]
// small dense layer on the vector units (side models: LSTM / transformer heads): C[m*ldc + n] = act(X[m*ld + :K] . W[n*K + :K] + bias[n])
hipError_t launch_linear_f32(const float* X, int ld, const float* W, const float* bias, float* C, int ldc, int M, int N, int K, int relu,
                             hipStream_t s);
hipError_t launch_spin(int32_t microseconds, hipStream_t s);
hipError_t launch_gate(const int* flag, int32_t max_microseconds, hipStream_t s);  // one thread busy for that long (stream-concurrency probe)
hipError_t launch_nchw_to_padded(const float* x, float* out, int32_t n, int32_t out_bf16, hipStream_t s);
// 3x3/2 max pool, padded [n][66][66][64] -> padded [n][34][34][64]
hipError_t launch_maxpool(const float* in, float* out, int32_t n, hipStream_t s);
// global average pool, padded [n][6][6][512] -> [n][512]
hipError_t launch_avgpool(const float* in, float* out, int32_t n, hipStream_t s);
hipError_t launch_maxpool_bf16(const void* in, void* out, int32_t n, hipStream_t s);  // bf16 storage
hipError_t launch_avgpool_bf16(const void* in, float* out, int32_t n, hipStream_t s);  // bf16 in, fp32 out
// window gather table: rows into the feature cache, see head_gather_kernel
hipError_t launch_window_gather(int32_t* gather, int32_t frame_num_lo, int32_t count, int32_t fighters,
                                int32_t seq, int32_t delta, int32_t max_frames, int32_t min_frame, int32_t sub_frames, hipStream_t s);
hipError_t launch_identity_gather(int32_t* gather, int32_t n, hipStream_t s);
hipError_t launch_scatter_rows(const float* feats, const int32_t* st, const int32_t* ids, float* cache, int32_t* cache_st,
                               int32_t n, int32_t fighters, int32_t clip_frames, int32_t* bad_ids, hipStream_t s);

struct HeadParams {
    const float* h1;      // [nwin][512] post-ReLU Conv1d output, or nullptr when the slabs below are given
    const float* slab;    // split-K partial sums of the Conv1d [splitk][nwin][512] (ordered sum + bias + ReLU done here)
    const float* b1;      // Conv1d bias [512]
    int32_t splitk;
    const float* w2;      // [128][512]
    const float* b2;      // [128]
    const float* w3;      // [A][128]
    const float* b3;      // [A]
    float* logp;          // [nwin][A] or nullptr
    void* records;        // pa_record[nwin] or nullptr
    const int32_t* crop_status;  // per feature row or nullptr
    const int32_t* gather;       // [nwin][seq] (middle slot gives the window's own crop)
    int32_t nwin, num_actions, fighters, seq;
    int32_t class_ids[4];
};
hipError_t launch_head_mlp(const HeadParams& p, hipStream_t s);

}  // namespace pa
