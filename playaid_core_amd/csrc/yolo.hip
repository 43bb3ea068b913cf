// A fully convolutional detection network given as a layer table (the YOLOv5 detector the
// reference shells out to, playaid/ai_runner.py:191-224), run on the engine's fp32 implicit-GEMM kernel: frames in, the
// decoded head rows out -- what pa_detect_postprocess takes.
//
// The host (playaid_core_amd/yolov5.py) folds BatchNorm, lays the weights out [cout][ky][kx][cin] and describes every
// layer with a pa_net_layer. Activations are zero-bordered NHWC buffers; a layer reads / writes a CHANNEL SLICE of a
// buffer (offset + pixel pitch), which is how the network's concatenations cost nothing: the producers write straight
// into their slice of the consumer's input. Kinds:
//   0  convolution 1x1 | 3x3, stride 1 | 2, + bias, activation (none / ReLU / SiLU), residual before or after it
//      (implicit GEMM on the matrix cores, igemm.hip; cin % 32 == 0, cout % 64 == 0: the table pads with zero weights)
//   3  the 6x6 / 2 stem on the letter-boxed RGB image + bias + SiLU: the same GEMM kernel, one tap per kernel row whose K
//      chunk is 8 consecutive NHWC4 pixels (weights [cout][6][8 px][4 ch], kx >= 6 and channel 3 zero)
//   4  max-pool 5x5 / 1 (SPPF), slice to slice
//   5  nearest-neighbour 2x up-sampling, slice to slice
//   6  Detect decode of one scale: sigmoid, grid / anchor arithmetic -> rows (cx, cy, w, h, obj, classes) in net pixels
// Before the table runs, letterbox_kernel does what detect.py's LoadImages does to a frame: cv2.resize(INTER_LINEAR) to
// the un-padded size, a 114-grey border up to the network input (a multiple of 32), BGR -> RGB, / 255.
#include "pa_kernels.h"
#include "../../include/playaid_hip.h"
#include <cstring>
#include <string>
#include <vector>

namespace pa {
namespace {

// cv2.resize(src, (new_w, new_h), INTER_LINEAR) on 8-bit pixels: OpenCV's fixed-point bilinear resizer (11-bit
// coefficients, HResize into int, VResizeLinear: ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2 >> 2), then
// copyMakeBorder(114), BGR -> RGB, / 255. out: [n][net_h + 4][net_w + 4][4] fp32 with a 2-pixel ZERO border (the stem's
// padding) and channel 3 = 0.
__global__ __launch_bounds__(256) void letterbox_kernel(const uint8_t* __restrict__ frames, int n, int H, int W, int new_h, int new_w,
                                                        int top, int left, int net_h, int net_w, float* __restrict__ out) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), img = blockIdx.z;
    if (x >= net_w || y >= net_h) return;
    int rgb[3] = {114, 114, 114};
    const int dy = y - top, dx = x - left;
    if (dy >= 0 && dy < new_h && dx >= 0 && dx < new_w) {
        const uint8_t* f = frames + (size_t)img * H * W * 3;
        if (new_h == H && new_w == W) {
            const uint8_t* s = f + ((size_t)dy * W + dx) * 3;
            rgb[0] = s[2]; rgb[1] = s[1]; rgb[2] = s[0];
        } else {
            // resize.cpp: fx = (dx + 0.5) * scale - 0.5 in float, sx = cvFloor(fx), clamped; cbuf = saturate_cast<short>(f * 2048)
            const double inv_x = (double)new_w / W, inv_y = (double)new_h / H;
            const double scale_x = 1.0 / inv_x, scale_y = 1.0 / inv_y;
            float fx = (float)((dx + 0.5) * scale_x - 0.5);
            int sx = (int)floorf(fx);
            fx -= sx;
            if (sx < 0) { fx = 0.f; sx = 0; }
            if (sx >= W - 1) { fx = 0.f; sx = W - 1; }
            float fy = (float)((dy + 0.5) * scale_y - 0.5);
            int sy = (int)floorf(fy);
            fy -= sy;
            if (sy < 0) { fy = 0.f; sy = 0; }
            if (sy >= H - 1) { fy = 0.f; sy = H - 1; }
            auto sat = [](float v) { const int r = (int)rintf(v); return r < -32768 ? -32768 : (r > 32767 ? 32767 : r); };
            const int a0 = sat((1.f - fx) * 2048.f), a1 = sat(fx * 2048.f);
            const int b0 = sat((1.f - fy) * 2048.f), b1 = sat(fy * 2048.f);
            const int sx1 = sx + 1 < W ? sx + 1 : sx, sy1 = sy + 1 < H ? sy + 1 : sy;
            const uint8_t* r0 = f + (size_t)sy * W * 3;
            const uint8_t* r1 = f + (size_t)sy1 * W * 3;
            for (int c = 0; c < 3; ++c) {
                const int S0 = r0[sx * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;
                const int S1 = r1[sx * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
                int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
                v = v < 0 ? 0 : (v > 255 ? 255 : v);
                rgb[2 - c] = v;
            }
        }
    }
    float4 o = make_float4((float)rgb[0] / 255.0f, (float)rgb[1] / 255.0f, (float)rgb[2] / 255.0f, 0.f);
    reinterpret_cast<float4*>(out)[((size_t)img * (net_h + 4) + y + 2) * (net_w + 4) + x + 2] = o;
}

struct SliceGeom {
    int h, w, pad, cstride, coff;
};
__device__ __forceinline__ size_t px_off(const SliceGeom& g, int img, int y, int x) {
    return (((size_t)img * (g.h + 2 * g.pad) + y + g.pad) * (g.w + 2 * g.pad) + x + g.pad) * g.cstride + g.coff;
}

// max-pool 5x5, stride 1, padding 2 (nn.MaxPool2d: the padding does not take part), 4 channels per thread
__global__ __launch_bounds__(256) void maxpool5_kernel(const float* __restrict__ in, SliceGeom gi, float* __restrict__ out, SliceGeom go, int n,
                                                       int c) {
    const int c4 = c / 4;
    const long long total = (long long)n * gi.h * gi.w * c4;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int q = (int)(t % c4);
        long long pix = t / c4;
        const int x = (int)(pix % gi.w);
        pix /= gi.w;
        const int y = (int)(pix % gi.h), img = (int)(pix / gi.h);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int dy = -2; dy <= 2; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= gi.h) continue;
            for (int dx = -2; dx <= 2; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= gi.w) continue;
                const float4 v = *reinterpret_cast<const float4*>(in + px_off(gi, img, yy, xx) + q * 4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        *reinterpret_cast<float4*>(out + px_off(go, img, y, x) + q * 4) = m;
    }
}

// nn.Upsample(scale_factor=2, mode="nearest"): out[y][x] = in[y / 2][x / 2]
__global__ __launch_bounds__(256) void upsample2_kernel(const float* __restrict__ in, SliceGeom gi, float* __restrict__ out, SliceGeom go, int n,
                                                        int c) {
    const int c4 = c / 4;
    const long long total = (long long)n * go.h * go.w * c4;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int q = (int)(t % c4);
        long long pix = t / c4;
        const int x = (int)(pix % go.w);
        pix /= go.w;
        const int y = (int)(pix % go.h), img = (int)(pix / go.h);
        *reinterpret_cast<float4*>(out + px_off(go, img, y, x) + q * 4) =
            *reinterpret_cast<const float4*>(in + px_off(gi, img, y >> 1, x >> 1) + q * 4);
    }
}

// models/yolo.py Detect.forward (inference): y = sigmoid(conv out); xy = (y * 2 + grid) * stride with grid = index - 0.5;
// wh = (y * 2)^2 * anchor; rows of one scale in (anchor, y, x) order at row0.
__global__ __launch_bounds__(256) void detect_decode_kernel(const float* __restrict__ in, SliceGeom gi, int n, int na, int no, float stride,
                                                            const float* __restrict__ anchors_px, float* __restrict__ pred, int rows_total,
                                                            int row0) {
    const long long total = (long long)n * na * gi.h * gi.w;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        long long r = t;
        const int x = (int)(r % gi.w);
        r /= gi.w;
        const int y = (int)(r % gi.h);
        r /= gi.h;
        const int a = (int)(r % na), img = (int)(r / na);
        const float* v = in + px_off(gi, img, y, x) + a * no;
        float* o = pred + ((size_t)img * rows_total + row0 + ((size_t)a * gi.h + y) * gi.w + x) * no;
        for (int k = 0; k < no; ++k) {
            const float s = 1.f / (1.f + expf(-v[k]));
            float val = s;
            if (k == 0) val = (s * 2.f + ((float)x - 0.5f)) * stride;
            else if (k == 1) val = (s * 2.f + ((float)y - 0.5f)) * stride;
            else if (k == 2) val = (s * 2.f) * (s * 2.f) * anchors_px[a * 2];
            else if (k == 3) val = (s * 2.f) * (s * 2.f) * anchors_px[a * 2 + 1];
            o[k] = val;
        }
    }
}

int grid_for(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace
}  // namespace pa

struct pa_detector {
    int device = 0, max_images = 0, net_h = 0, net_w = 0, nc = 0, rows = 0;
    std::vector<pa_net_layer> layers;
    std::vector<float*> bufs;
    std::vector<size_t> buf_floats;
    float* weights = nullptr;
    size_t n_weights = 0;
    float* x0 = nullptr;       // letter-boxed input [max_images][net_h + 4][net_w + 4][4]
    float* anchors = nullptr;  // device copy of the decode layers' anchors [n_decode][8]
    std::string last_error;
};

extern "C" {

const char* pa_detector_last_error(const pa_detector* h) { return h ? h->last_error.c_str() : "null handle"; }

void pa_detector_destroy(pa_detector* h) {
    if (!h) return;
    (void)hipFree(h->weights);
    (void)hipFree(h->x0);
    (void)hipFree(h->anchors);
    for (float* b : h->bufs) (void)hipFree(b);
    delete h;
}

int pa_detector_rows(const pa_detector* h) { return h ? h->rows : 0; }

int pa_detector_create(int32_t device, const pa_net_layer* layers, int32_t n_layers, const int64_t* buf_floats_per_image, int32_t n_bufs,
                       const float* weights_host, size_t n_weights, int32_t max_images, int32_t net_h, int32_t net_w, int32_t num_classes,
                       pa_detector** out) {
    if (!out) return PA_ERR_INVALID_ARG;
    *out = nullptr;
    if (!layers || n_layers < 1 || !buf_floats_per_image || n_bufs < 1 || !weights_host || n_weights < 1 || max_images < 1 || net_h < 32 ||
        net_w < 32 || net_h % 32 || net_w % 32 || num_classes < 1 || num_classes > 80)
        return PA_ERR_INVALID_ARG;
    pa_detector* h = new pa_detector();
    *out = h;
    h->device = device; h->max_images = max_images; h->net_h = net_h; h->net_w = net_w; h->nc = num_classes;
    h->layers.assign(layers, layers + n_layers);
    auto bad = [&](int i, const char* what) {
        h->last_error = "layer " + std::to_string(i) + ": " + what;
        return PA_ERR_INVALID_ARG;
    };
    auto slice_ok = [&](int buf, int hh, int ww, int pad, int cstride, int coff, int c) {
        if (buf < 0 || buf >= n_bufs || pad < 0 || cstride < c || coff < 0 || coff + c > cstride || (coff & 3) || (cstride & 3)) return false;
        return (long long)(hh + 2 * pad) * (ww + 2 * pad) * cstride <= buf_floats_per_image[buf];
    };
    int rows = 0, n_decode = 0;
    const int no = 5 + num_classes;
    for (int i = 0; i < n_layers; ++i) {
        const pa_net_layer& L = h->layers[i];
        if (L.in_h < 1 || L.in_w < 1) return bad(i, "bad size");
        const int oh = L.kind == 5 ? L.in_h * 2 : L.in_h / (L.stride > 0 ? L.stride : 1);
        const int ow = L.kind == 5 ? L.in_w * 2 : L.in_w / (L.stride > 0 ? L.stride : 1);
        if (L.kind == 0) {
            if ((L.ksize != 1 && L.ksize != 3) || (L.stride != 1 && L.stride != 2) || L.cin % 32 || L.cout % 64 || L.in_h % L.stride ||
                L.in_w % L.stride || L.in_pad < (L.ksize - 1) / 2 || L.act < 0 || L.act > 2)
                return bad(i, "unsupported convolution");
            if (!slice_ok(L.in_buf, L.in_h, L.in_w, L.in_pad, L.in_cstride, L.in_coff, L.cin) || (L.in_coff % 32) ||
                !slice_ok(L.out_buf, oh, ow, L.out_pad, L.out_cstride, L.out_coff, L.cout))
                return bad(i, "slice outside its buffer");
            if (L.res_buf >= 0 && (!slice_ok(L.res_buf, oh, ow, L.out_pad, L.out_cstride, L.res_coff, L.cout)))
                return bad(i, "residual must share the output's geometry and pixel pitch");
            if (L.w_off < 0 || L.b_off < 0 || (size_t)L.w_off + (size_t)L.cout * L.ksize * L.ksize * L.cin > n_weights ||
                (size_t)L.b_off + L.cout > n_weights)
                return bad(i, "weights outside the blob");
        } else if (L.kind == 3) {
            if (L.in_h != net_h || L.in_w != net_w || L.cout % 64 || L.w_off < 0 || (size_t)L.w_off + (size_t)L.cout * 192 > n_weights ||
                L.b_off < 0 || (size_t)L.b_off + L.cout > n_weights || !slice_ok(L.out_buf, net_h / 2, net_w / 2, L.out_pad, L.out_cstride, L.out_coff, L.cout))
                return bad(i, "bad stem");
        } else if (L.kind == 4 || L.kind == 5) {
            if (L.cin % 4 || !slice_ok(L.in_buf, L.in_h, L.in_w, L.in_pad, L.in_cstride, L.in_coff, L.cin) ||
                !slice_ok(L.out_buf, oh, ow, L.out_pad, L.out_cstride, L.out_coff, L.cin))
                return bad(i, "bad pool / up-sampling slice");
        } else if (L.kind == 6) {
            if (!slice_ok(L.in_buf, L.in_h, L.in_w, L.in_pad, L.in_cstride, L.in_coff, 3 * no)) return bad(i, "bad decode slice");
            rows += 3 * L.in_h * L.in_w;
            ++n_decode;
        } else {
            return bad(i, "unknown kind");
        }
    }
    if (rows < 1) return bad(n_layers - 1, "no decode layer");
    h->rows = rows;
    auto chk = [&](hipError_t e, const char* what) -> bool {
        if (e == hipSuccess) return true;
        h->last_error = std::string(what) + ": " + hipGetErrorString(e);
        return false;
    };
    if (!chk(hipSetDevice(device), "hipSetDevice")) return PA_ERR_NO_DEVICE;
    h->n_weights = n_weights;
    if (!chk(hipMalloc(&h->weights, n_weights * sizeof(float)), "hipMalloc weights")) return PA_ERR_HIP;
    if (!chk(hipMemcpy(h->weights, weights_host, n_weights * sizeof(float), hipMemcpyHostToDevice), "upload weights")) return PA_ERR_HIP;
    h->bufs.assign(n_bufs, nullptr);
    h->buf_floats.assign(buf_floats_per_image, buf_floats_per_image + n_bufs);
    for (int b = 0; b < n_bufs; ++b) {
        // (+ slack: a partial last tile of the GEMM reads rows past the last image)
        const size_t bytes = ((size_t)max_images * h->buf_floats[b] + 128 * 2048) * sizeof(float);
        if (!chk(hipMalloc(&h->bufs[b], bytes), "hipMalloc activations")) return PA_ERR_HIP;
        if (!chk(hipMemset(h->bufs[b], 0, bytes), "hipMemset activations")) return PA_ERR_HIP;  // the zero borders stay zero
    }
    // (+ slack: the stem's 8-pixel K chunks of the last row run two pixels past it, a partial last GEMM tile further)
    const size_t x0_bytes = ((size_t)max_images * (net_h + 4) * (net_w + 4) * 4 + 128 * 2048) * sizeof(float);
    if (!chk(hipMalloc(&h->x0, x0_bytes), "hipMalloc input")) return PA_ERR_HIP;
    if (!chk(hipMemset(h->x0, 0, x0_bytes), "hipMemset input")) return PA_ERR_HIP;
    std::vector<float> anc((size_t)n_decode * 8, 0.f);
    int di = 0;
    for (const pa_net_layer& L : h->layers)
        if (L.kind == 6) {
            for (int k = 0; k < 6; ++k) anc[(size_t)di * 8 + k] = L.aux[1 + k];
            ++di;
        }
    if (!chk(hipMalloc(&h->anchors, anc.size() * sizeof(float)), "hipMalloc anchors")) return PA_ERR_HIP;
    if (!chk(hipMemcpy(h->anchors, anc.data(), anc.size() * sizeof(float), hipMemcpyHostToDevice), "upload anchors")) return PA_ERR_HIP;
    return PA_OK;
}

static int detector_run(pa_detector* h, const uint8_t* frames, int32_t n, int32_t height, int32_t width, float* pred, void* stream,
                        std::vector<hipEvent_t>* ev) {
    if (!h) return PA_ERR_INVALID_ARG;
    auto fail = [&](int code, const std::string& msg) { h->last_error = msg; return code; };
    if (!frames || !pred || n < 1 || height < 1 || width < 1) return fail(PA_ERR_INVALID_ARG, "pa_detector_forward: bad argument");
    if (n > h->max_images) return fail(PA_ERR_CAPACITY, "pa_detector_forward: more images than max_images");
    hipStream_t s = (hipStream_t)stream;
#define DT_HIP(call)                                                                                         \
    do {                                                                                                     \
        hipError_t e__ = (call);                                                                             \
        if (e__ != hipSuccess) return fail(PA_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)
    // utils/augmentations.py letterbox (auto = False here: the table fixes the network input): r = min(net / frame), the
    // un-padded size rounded, the padding split in two with the -0.1 / +0.1 rounding
    const double r = std::min((double)h->net_h / height, (double)h->net_w / width);
    const int new_w = (int)lrint(width * r), new_h = (int)lrint(height * r);
    if (new_w > h->net_w || new_h > h->net_h || new_w < 1 || new_h < 1) return fail(PA_ERR_INVALID_ARG, "pa_detector_forward: frame does not fit the network input");
    const double dw = (h->net_w - new_w) / 2.0, dh = (h->net_h - new_h) / 2.0;
    const int top = (int)lrint(dh - 0.1), left = (int)lrint(dw - 0.1);
    hipLaunchKernelGGL(pa::letterbox_kernel, dim3((h->net_w + 63) / 64, (h->net_h + 3) / 4, n), dim3(256), 0, s, frames, n, height, width, new_h,
                       new_w, top, left, h->net_h, h->net_w, h->x0);
    DT_HIP(hipGetLastError());
    const int no = 5 + h->nc;
    int row0 = 0, di = 0;
    for (size_t li = 0; li < h->layers.size(); ++li) {
        const pa_net_layer& L = h->layers[li];
        if (ev) DT_HIP(hipEventRecord((*ev)[li], s));  // (profiling call only: layer li runs between events li and li + 1)
        if (L.kind == 3) {
            // the 6x6 / 2 stem as an implicit GEMM: one tap per kernel row, its K chunk = 8 consecutive NHWC4 pixels of
            // the letter-boxed image (kx 6, 7 and channel 3 meet zero weights), K = 6 x 32
            const int oh = h->net_h / 2, ow = h->net_w / 2;
            pa::GemmParams p;
            memset(&p, 0, sizeof(p));
            p.act = h->x0;
            p.wgt = h->weights + L.w_off;
            p.bias = h->weights + L.b_off;
            p.out = h->bufs[L.out_buf] + L.out_coff;
            p.M = n * oh * ow;
            p.N = L.cout;
            p.taps = 6; p.kw_taps = 1; p.chunk = 32; p.ktot = 192;
            p.howo = oh * ow; p.wo = ow;
            p.in_px_stride = 4;
            p.in_row_stride = (h->net_w + 4) * 4;
            p.in_img_stride = (h->net_h + 4) * (h->net_w + 4) * 4;
            p.stride = 2;
            p.out_px_stride = L.out_cstride;
            p.out_row_stride = (ow + 2 * L.out_pad) * L.out_cstride;
            p.out_img_stride = (oh + 2 * L.out_pad) * (ow + 2 * L.out_pad) * L.out_cstride;
            p.out_pad = L.out_pad;
            p.relu = 2;
            p.splitk = 1;
            const hipError_t pe = pa::launch_igemm(p, pa::TILE_128x64, s);
            if (pe != hipSuccess) return fail(PA_ERR_HIP, "stem: " + std::string(hipGetErrorString(pe)));
            continue;
        }
        if (L.kind == 4 || L.kind == 5) {
            pa::SliceGeom gi = {L.in_h, L.in_w, L.in_pad, L.in_cstride, L.in_coff};
            const int oh = L.kind == 5 ? L.in_h * 2 : L.in_h, ow = L.kind == 5 ? L.in_w * 2 : L.in_w;
            pa::SliceGeom go = {oh, ow, L.out_pad, L.out_cstride, L.out_coff};
            const long long total = (long long)n * oh * ow * (L.cin / 4);
            if (L.kind == 4)
                hipLaunchKernelGGL(pa::maxpool5_kernel, dim3(pa::grid_for(total)), dim3(256), 0, s, h->bufs[L.in_buf], gi, h->bufs[L.out_buf], go, n, L.cin);
            else
                hipLaunchKernelGGL(pa::upsample2_kernel, dim3(pa::grid_for(total)), dim3(256), 0, s, h->bufs[L.in_buf], gi, h->bufs[L.out_buf], go, n, L.cin);
            DT_HIP(hipGetLastError());
            continue;
        }
        if (L.kind == 6) {
            pa::SliceGeom gi = {L.in_h, L.in_w, L.in_pad, L.in_cstride, L.in_coff};
            const long long total = (long long)n * 3 * L.in_h * L.in_w;
            hipLaunchKernelGGL(pa::detect_decode_kernel, dim3(pa::grid_for(total)), dim3(256), 0, s, h->bufs[L.in_buf], gi, n, 3, no, L.aux[0],
                               h->anchors + (size_t)di * 8, pred, h->rows, row0);
            DT_HIP(hipGetLastError());
            row0 += 3 * L.in_h * L.in_w;
            ++di;
            continue;
        }
        const int oh = L.in_h / L.stride, ow = L.in_w / L.stride;
        const int in_wb = L.in_w + 2 * L.in_pad, in_hb = L.in_h + 2 * L.in_pad;
        const int out_wb = ow + 2 * L.out_pad, out_hb = oh + 2 * L.out_pad;
        pa::GemmParams p;
        memset(&p, 0, sizeof(p));
        p.act = h->bufs[L.in_buf] + L.in_coff;
        p.wgt = h->weights + L.w_off;
        p.bias = h->weights + L.b_off;
        p.residual = L.res_buf >= 0 ? h->bufs[L.res_buf] + L.res_coff : nullptr;
        p.out = h->bufs[L.out_buf] + L.out_coff;
        p.M = n * oh * ow;
        p.N = L.cout;
        p.taps = L.ksize * L.ksize;
        p.kw_taps = L.ksize;
        p.chunk = L.cin;
        p.ktot = p.taps * p.chunk;
        p.howo = oh * ow;
        p.wo = ow;
        p.in_px_stride = L.in_cstride;
        p.in_row_stride = in_wb * L.in_cstride;
        p.in_img_stride = in_hb * in_wb * L.in_cstride;
        p.stride = L.stride;
        p.off_y = p.off_x = L.in_pad - (L.ksize - 1) / 2;
        p.out_px_stride = L.out_cstride;
        p.out_row_stride = out_wb * L.out_cstride;
        p.out_img_stride = out_hb * out_wb * L.out_cstride;
        p.out_pad = L.out_pad;
        p.relu = L.act;
        p.res_after = L.res_after;
        p.splitk = 1;
        const long long t128 = (long long)((p.M + 127) / 128) * (p.N / 64);
        const pa::GemmTile tile = (p.N % 128 == 0 && t128 / 2 >= 512) ? pa::TILE_128x128 : (t128 >= 512 ? pa::TILE_128x64 : pa::TILE_64x64);
        const hipError_t pe = pa::launch_igemm(p, tile, s);
        if (pe != hipSuccess) return fail(PA_ERR_HIP, "layer " + std::to_string(li) + ": " + hipGetErrorString(pe));
    }
    if (ev) DT_HIP(hipEventRecord(ev->back(), s));
#undef DT_HIP
    return PA_OK;
}

int pa_detector_forward(pa_detector* h, const uint8_t* frames, int32_t n, int32_t height, int32_t width, float* pred, void* stream) {
    return detector_run(h, frames, n, height, width, pred, stream, nullptr);
}

int pa_detector_forward_timed(pa_detector* h, const uint8_t* frames, int32_t n, int32_t height, int32_t width, float* pred, void* stream,
                              float* layer_us, int32_t cap) {
    if (!h || !layer_us || cap < (int32_t)h->layers.size()) return PA_ERR_INVALID_ARG;
    std::vector<hipEvent_t> ev(h->layers.size() + 1);
    for (hipEvent_t& e : ev)
        if (hipEventCreate(&e) != hipSuccess) return PA_ERR_HIP;
    int rc = detector_run(h, frames, n, height, width, pred, stream, &ev);
    if (rc == PA_OK && hipEventSynchronize(ev.back()) != hipSuccess) rc = PA_ERR_HIP;
    for (size_t i = 0; rc == PA_OK && i < h->layers.size(); ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev[i], ev[i + 1]) != hipSuccess) rc = PA_ERR_HIP;
        layer_us[i] = ms * 1000.f;
    }
    for (hipEvent_t& e : ev) (void)hipEventDestroy(e);
    return rc;
}

}  // extern "C"
