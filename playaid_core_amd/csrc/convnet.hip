// A conv-net described by a table, run on the engine's fp32 convolution kernels (the
// ResNet-50 backbone of the reference's ResnetTransformerDetector, playaid/models/resnet_transformer_detector.py:37,
// "reusing the same backbone kernels").
//
// The host hands over BatchNorm-folded weights laid out [cout][ky][kx][cin] (stem: [64][7][8 px][4 ch]) and one
// pa_conv_desc per layer; activations live in a small set of device buffers, zero-bordered NHWC, each used with
// ONE geometry so that borders written as zero at creation stay zero. Layers run in table order on one stream:
//   kind 0  convolution k x k (k = 1 | 3), stride 1 | 2, + bias (+ residual) (+ ReLU): conv3x3_patch_kernel for the
//           stride-1 3x3 layers it covers, the im2col engine (igemm.hip) for everything else (1x1 = a GEMM);
//   kind 1  the 7x7/2 stem + BatchNorm + ReLU + 3x3/2 max-pool of a 128 x 128 x 3 input (stem_pool.hip);
//   kind 2  global average pool of the interior.
// Nothing here is specific to ResNet-50; the table in playaid_core_amd/resnet_transformer_detector.py is.
#include "pa_kernels.h"
#include "../../include/playaid_hip.h"
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace pa {
namespace {

// [n][(hw + 2 pad)^2][C] -> [n][C], mean over the interior
__global__ __launch_bounds__(256) void avgpool_any_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int hw, int pad, int C) {
    const int w = hw + 2 * pad;
    const size_t total = (size_t)n * C;
    const float inv = 1.f / (float)(hw * hw);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t img = i / C;
        const float* src = in + img * w * w * C + c;
        float sum = 0.f;
        for (int y = 0; y < hw; ++y)
            for (int x = 0; x < hw; ++x) sum += src[(size_t)((y + pad) * w + x + pad) * C];
        out[i] = sum * inv;
    }
}

}  // namespace
}  // namespace pa

struct pa_convnet {
    int device = 0, max_crops = 0;
    std::vector<pa_conv_desc> descs;
    std::vector<float*> bufs;
    std::vector<size_t> buf_floats;  // per crop
    float* weights = nullptr;
    size_t n_weights = 0;
    float* wino_weights = nullptr;      // the stride-1 3x3 layers' filters in the Winograd kernel's layout (wino.hip)
    std::vector<long long> wino_off;    // per layer: float offset into wino_weights, -1 = direct form
    std::vector<int> wino_bn;           // per layer: output channels per workgroup its filters were laid out for
    int compute_dtype = PA_DTYPE_F32;          // PA_DTYPE_EMULATED_F32: the layers listed in split_off run on psgemm.hip
    unsigned short* split_weights = nullptr;   // those layers' weights as three bf16 slices (psgemm_pack_weights)
    std::vector<long long> split_off;          // per layer: element offset into split_weights, -1 = the exact kernel
    float* x0 = nullptr;  // [max_crops][134][134][4] model input of the stem
    std::string last_error;
};

namespace {

int cn_fail(pa_convnet* h, int code, const std::string& msg) {
    if (h) h->last_error = msg;
    return code;
}

// interior size and channel count a layer leaves in its output buffer
void out_geom(const pa_conv_desc& d, int* hw, int* c) {
    if (d.kind == 1) { *hw = 32; *c = 64; }
    else if (d.kind == 2) { *hw = 1; *c = d.cin; }
    else { *hw = d.in_hw / d.stride; *c = d.cout; }
}

}  // namespace

extern "C" {

const char* pa_convnet_last_error(const pa_convnet* h) { return h ? h->last_error.c_str() : "null handle"; }

int pa_convnet_create(int32_t device, const pa_conv_desc* descs, int32_t n_descs, const int64_t* buf_floats_per_crop, int32_t n_bufs,
                      const float* weights_host, size_t n_weights, int32_t max_crops, pa_convnet** out) {
    return pa_convnet_create_dtype(device, descs, n_descs, buf_floats_per_crop, n_bufs, weights_host, n_weights, max_crops, PA_DTYPE_F32, out);
}

int pa_convnet_create_dtype(int32_t device, const pa_conv_desc* descs, int32_t n_descs, const int64_t* buf_floats_per_crop, int32_t n_bufs,
                            const float* weights_host, size_t n_weights, int32_t max_crops, int32_t compute_dtype, pa_convnet** out) {
    if (!out) return PA_ERR_INVALID_ARG;
    *out = nullptr;
    if (!descs || n_descs < 1 || !buf_floats_per_crop || n_bufs < 1 || !weights_host || n_weights < 1 || max_crops < 1 ||
        (compute_dtype != PA_DTYPE_F32 && compute_dtype != PA_DTYPE_EMULATED_F32))
        return PA_ERR_INVALID_ARG;
    pa_convnet* h = new pa_convnet();
    *out = h;
    h->device = device;
    h->compute_dtype = compute_dtype;
    h->max_crops = max_crops;
    h->descs.assign(descs, descs + n_descs);
    // validate the table: buffer indices, weight ranges, buffer sizes, one geometry per bordered buffer
    struct Geom { int hw = -1, pad = -1, c = -1; };
    std::vector<Geom> geom(n_bufs);
    auto use = [&](int b, int hw, int pad, int c, const char* what, int li) -> bool {
        if (b < 0 || b >= n_bufs) { h->last_error = "layer " + std::to_string(li) + ": bad " + what + " buffer"; return false; }
        const long long need = (long long)(hw + 2 * pad) * (hw + 2 * pad) * c;
        if (need > buf_floats_per_crop[b]) { h->last_error = "layer " + std::to_string(li) + ": " + what + " buffer too small"; return false; }
        Geom& g = geom[b];
        if (pad > 0 || g.pad > 0) {
            if (g.hw >= 0 && (g.hw != hw || g.pad != pad || g.c != c)) {
                h->last_error = "layer " + std::to_string(li) + ": zero-bordered buffer " + std::to_string(b) + " used with two geometries";
                return false;
            }
        }
        if (g.hw < 0 || pad > 0) { g.hw = hw; g.pad = pad; g.c = c; }
        return true;
    };
    for (int i = 0; i < n_descs; ++i) {
        const pa_conv_desc& d = h->descs[i];
        int ohw, oc;
        out_geom(d, &ohw, &oc);
        if (d.kind == 0) {
            if ((d.ksize != 1 && d.ksize != 3) || (d.stride != 1 && d.stride != 2) || d.cin % 32 != 0 || d.cout % 64 != 0 || d.in_hw < 1 ||
                d.in_hw % d.stride != 0 || d.in_pad < (d.ksize - 1) / 2 || d.out_pad < 0)
                return cn_fail(h, PA_ERR_INVALID_ARG, "layer " + std::to_string(i) + ": unsupported convolution");
            if (d.w_off < 0 || d.b_off < 0 || (size_t)d.w_off + (size_t)d.cout * d.ksize * d.ksize * d.cin > n_weights ||
                (size_t)d.b_off + d.cout > n_weights)
                return cn_fail(h, PA_ERR_BAD_WEIGHTS, "layer " + std::to_string(i) + ": weights outside the blob");
            if (!use(d.in_buf, d.in_hw, d.in_pad, d.cin, "input", i) || !use(d.out_buf, ohw, d.out_pad, oc, "output", i)) return PA_ERR_INVALID_ARG;
            if (d.res_buf >= 0 && !use(d.res_buf, ohw, d.out_pad, oc, "residual", i)) return PA_ERR_INVALID_ARG;
        } else if (d.kind == 1) {
            if (d.out_pad != 1 || d.w_off < 0 || (size_t)d.w_off + 64 * 224 > n_weights || d.b_off < 0 || (size_t)d.b_off + 64 > n_weights)
                return cn_fail(h, PA_ERR_INVALID_ARG, "layer " + std::to_string(i) + ": bad stem");
            if (!use(d.out_buf, 32, 1, 64, "output", i)) return PA_ERR_INVALID_ARG;
        } else if (d.kind == 2) {
            if (d.cin < 1 || d.in_hw < 1 || d.in_pad < 0) return cn_fail(h, PA_ERR_INVALID_ARG, "layer " + std::to_string(i) + ": bad pool");
            if (!use(d.in_buf, d.in_hw, d.in_pad, d.cin, "input", i) || !use(d.out_buf, 1, 0, d.cin, "output", i)) return PA_ERR_INVALID_ARG;
        } else {
            return cn_fail(h, PA_ERR_INVALID_ARG, "layer " + std::to_string(i) + ": unknown kind");
        }
    }
    auto chk = [&](hipError_t e, const char* what) -> bool {
        if (e == hipSuccess) return true;
        h->last_error = std::string(what) + ": " + hipGetErrorString(e);
        return false;
    };
    if (!chk(hipSetDevice(device), "hipSetDevice")) return PA_ERR_NO_DEVICE;
    h->n_weights = n_weights;
    if (!chk(hipMalloc(&h->weights, n_weights * sizeof(float)), "hipMalloc weights")) return PA_ERR_HIP;
    if (!chk(hipMemcpy(h->weights, weights_host, n_weights * sizeof(float), hipMemcpyHostToDevice), "upload weights")) return PA_ERR_HIP;
    {
        // stride-1 3x3 convolutions on maps of 8 x 8 and larger run as Winograd F(2x2, 3x3) (wino.hip, as in the engine's
        // ResNet-18: the 4 x 4 maps stay on the direct kernel); PA_CONVNET_WINO=0 keeps the direct form (A/B)
        static const int use_wino = getenv("PA_CONVNET_WINO") ? atoi(getenv("PA_CONVNET_WINO")) : 1;
        h->wino_off.assign(n_descs, -1);
        h->wino_bn.assign(n_descs, 0);
        size_t total = 0;
        for (int i = 0; i < n_descs; ++i) {
            const pa_conv_desc& d = h->descs[i];
            if (use_wino && d.kind == 0 && d.ksize == 3 && d.stride == 1 && d.in_pad == 1 && d.in_hw >= 8 && d.in_hw % 4 == 0 && d.cin % 8 == 0) {
                h->wino_off[i] = (long long)total;
                h->wino_bn[i] = pa::wino_pick_bn(d.cout, (long long)max_crops * (d.in_hw / 4) * (d.in_hw / 4));
                total += pa::wino_weight_floats(d.cin, d.cout);
            }
        }
        if (total) {
            std::vector<float> ug(total);
            for (int i = 0; i < n_descs; ++i)
                if (h->wino_off[i] >= 0) pa::wino_transform_weights(weights_host + h->descs[i].w_off, h->descs[i].cin, h->descs[i].cout, h->wino_bn[i], ug.data() + h->wino_off[i]);
            if (!chk(hipMalloc(&h->wino_weights, total * sizeof(float)), "hipMalloc Winograd filters")) return PA_ERR_HIP;
            if (!chk(hipMemcpy(h->wino_weights, ug.data(), total * sizeof(float), hipMemcpyHostToDevice), "upload Winograd filters")) return PA_ERR_HIP;
        }
    }
    h->split_off.assign(n_descs, -1);
    if (compute_dtype == PA_DTYPE_EMULATED_F32) {
        // every convolution that is not in Winograd form and whose 128-pixel tiles can fill at least half the chip at max_crops runs on
        // the emulated-fp32 persistent GEMM (psgemm.hip; below that its one-workgroup-per-CU grid is mostly empty and the exact
        // engine's 64 x 64 tiles are faster: profiles/r06_pgemm_split_layers.txt, ResNet-18's 8 x 8 and 4 x 4 maps)
        size_t total = 0;
        for (int i = 0; i < n_descs; ++i) {
            const pa_conv_desc& d = h->descs[i];
            if (d.kind != 0 || h->wino_off[i] >= 0 || d.cin % 32 || d.cout % 32) continue;
            const int ohw = d.in_hw / d.stride, bn = pa::psgemm_pick_bn(d.cout, d.res_buf >= 0);
            if (bn == 0 || (long long)(((long long)max_crops * ohw * ohw + 127) / 128) * (d.cout / bn) < 128) continue;
            h->split_off[i] = (long long)total;
            total += pa::psgemm_weight_elems(d.cout, d.ksize * d.ksize * d.cin, d.res_buf >= 0);
        }
        if (total) {
            std::vector<unsigned short> sw(total);
            for (int i = 0; i < n_descs; ++i)
                if (h->split_off[i] >= 0) {
                    const pa_conv_desc& d = h->descs[i];
                    pa::psgemm_pack_weights(weights_host + d.w_off, d.cout, d.ksize * d.ksize * d.cin, d.res_buf >= 0, sw.data() + h->split_off[i]);
                }
            if (!chk(hipMalloc(&h->split_weights, total * sizeof(unsigned short)), "hipMalloc split weights")) return PA_ERR_HIP;
            if (!chk(hipMemcpy(h->split_weights, sw.data(), total * sizeof(unsigned short), hipMemcpyHostToDevice), "upload split weights")) return PA_ERR_HIP;
        }
    }
    h->bufs.assign(n_bufs, nullptr);
    h->buf_floats.assign(buf_floats_per_crop, buf_floats_per_crop + n_bufs);
    for (int b = 0; b < n_bufs; ++b) {
        // (+ one 128-pixel tile of slack: a partial last tile of the patch kernel reads past the last crop)
        const size_t bytes = ((size_t)max_crops * h->buf_floats[b] + 128 * 2048) * sizeof(float);
        if (!chk(hipMalloc(&h->bufs[b], bytes), "hipMalloc activations")) return PA_ERR_HIP;
        if (!chk(hipMemset(h->bufs[b], 0, bytes), "hipMemset activations")) return PA_ERR_HIP;
    }
    const size_t x0_bytes = (size_t)max_crops * 134 * 134 * 4 * sizeof(float);
    if (!chk(hipMalloc(&h->x0, x0_bytes), "hipMalloc input")) return PA_ERR_HIP;
    if (!chk(hipMemset(h->x0, 0, x0_bytes), "hipMemset input")) return PA_ERR_HIP;
    return PA_OK;
}

void pa_convnet_destroy(pa_convnet* h) {
    if (!h) return;
    (void)hipFree(h->weights);
    (void)hipFree(h->wino_weights);
    (void)hipFree(h->split_weights);
    (void)hipFree(h->x0);
    for (float* b : h->bufs) (void)hipFree(b);
    delete h;
}

int pa_convnet_forward(pa_convnet* h, const float* x, int32_t n, float* out, int32_t out_floats_per_crop, void* stream) {
    if (!h) return PA_ERR_INVALID_ARG;
    if (!x || !out || n < 1) return cn_fail(h, PA_ERR_INVALID_ARG, "pa_convnet_forward: bad argument");
    if (n > h->max_crops) return cn_fail(h, PA_ERR_CAPACITY, "pa_convnet_forward: more crops than max_crops");
    hipStream_t s = (hipStream_t)stream;
#define CN_HIP(call)                                                                                  \
    do {                                                                                              \
        hipError_t e__ = (call);                                                                      \
        if (e__ != hipSuccess) return cn_fail(h, PA_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)
    CN_HIP(pa::launch_nchw_to_padded(x, h->x0, n, 0, s));
    for (size_t li = 0; li < h->descs.size(); ++li) {
        const pa_conv_desc& d = h->descs[li];
        if (d.kind == 1) {
            pa::StemPoolParams sp;
            memset(&sp, 0, sizeof(sp));
            sp.x = h->x0;
            sp.wgt = h->weights + d.w_off;
            sp.bias = h->weights + d.b_off;
            sp.out = h->bufs[d.out_buf];
            sp.crops = n;
            CN_HIP(pa::launch_stem_pool(sp, s));
            continue;
        }
        if (d.kind == 2) {
            const size_t total = (size_t)n * d.cin;
            int grid = (int)((total + 255) / 256);
            grid = grid > 2048 ? 2048 : grid;
            hipLaunchKernelGGL(pa::avgpool_any_kernel, dim3(grid), dim3(256), 0, s, h->bufs[d.in_buf], h->bufs[d.out_buf], n, d.in_hw, d.in_pad, d.cin);
            CN_HIP(hipGetLastError());
            continue;
        }
        const int out_hw = d.in_hw / d.stride;
        const int in_w = d.in_hw + 2 * d.in_pad, out_w = out_hw + 2 * d.out_pad;
        pa::GemmParams p;
        memset(&p, 0, sizeof(p));
        p.act = h->bufs[d.in_buf];
        p.wgt = h->weights + d.w_off;
        p.bias = h->weights + d.b_off;
        p.residual = d.res_buf >= 0 ? h->bufs[d.res_buf] : nullptr;
        p.out = h->bufs[d.out_buf];
        p.M = n * out_hw * out_hw;
        p.N = d.cout;
        p.taps = d.ksize * d.ksize;
        p.kw_taps = d.ksize;
        p.chunk = d.cin;
        p.ktot = p.taps * p.chunk;
        p.howo = out_hw * out_hw;
        p.wo = out_hw;
        p.in_px_stride = d.cin;
        p.in_row_stride = in_w * d.cin;
        p.in_img_stride = in_w * in_w * d.cin;
        p.stride = d.stride;
        p.off_y = p.off_x = d.in_pad - (d.ksize - 1) / 2;
        p.out_px_stride = d.cout;
        p.out_row_stride = out_w * d.cout;
        p.out_img_stride = out_w * out_w * d.cout;
        p.out_pad = d.out_pad;
        p.relu = d.relu;
        p.splitk = 1;
        // tile: the largest shape that still gives the chip ~two workgroups per CU
        const long long t128 = (long long)((p.M + 127) / 128) * (p.N / 64);
        const pa::GemmTile tile = (p.N % 128 == 0 && t128 / 2 >= 512) ? pa::TILE_128x128 : (t128 >= 512 ? pa::TILE_128x64 : pa::TILE_64x64);
        hipError_t pe = hipErrorInvalidValue;
        if (h->split_off[li] >= 0) pe = pa::launch_psgemm(p, h->split_weights + h->split_off[li], (size_t)n * p.out_img_stride, 0, s);
        if (pe == hipErrorInvalidValue && h->wino_off[li] >= 0) {
            pa::WinoParams q;
            memset(&q, 0, sizeof(q));
            q.act = p.act; q.wgt = h->wino_weights + h->wino_off[li]; q.bias = p.bias; q.residual = p.residual; q.out = p.out;
            q.n_img = n; q.height = d.in_hw; q.width = d.in_hw; q.cin = d.cin; q.cout = d.cout; q.bn = h->wino_bn[li];
            q.in_px_stride = p.in_px_stride; q.in_row_stride = p.in_row_stride; q.in_img_stride = p.in_img_stride;
            q.out_px_stride = p.out_px_stride; q.out_row_stride = p.out_row_stride; q.out_img_stride = p.out_img_stride; q.out_pad = p.out_pad;
            q.relu = p.relu;
            pe = pa::launch_wino3x3(q, s);
        }
        if (pe == hipErrorInvalidValue && d.ksize == 3 && d.stride == 1 && d.in_pad == 1) pe = pa::launch_conv3x3_patch(p, tile == pa::TILE_64x64 ? 64 : 128, s);
        if (pe == hipErrorInvalidValue) pe = pa::launch_igemm(p, tile, s);
        if (pe != hipSuccess) return cn_fail(h, PA_ERR_HIP, "layer " + std::to_string(li) + ": " + hipGetErrorString(pe));
    }
    // the last layer's output, interior only when it is bordered (a pooled vector has no border)
    const pa_conv_desc& last = h->descs.back();
    int ohw, oc;
    out_geom(last, &ohw, &oc);
    const int opad = last.kind == 0 ? last.out_pad : (last.kind == 1 ? 1 : 0);
    const int per_crop = (ohw + 2 * opad) * (ohw + 2 * opad) * oc;
    if (out_floats_per_crop != per_crop) return cn_fail(h, PA_ERR_INVALID_ARG, "pa_convnet_forward: out_floats_per_crop does not match the last layer");
    CN_HIP(hipMemcpyAsync(out, h->bufs[last.out_buf], (size_t)n * per_crop * sizeof(float), hipMemcpyDeviceToDevice, s));
#undef CN_HIP
    return PA_OK;
}

}  // extern "C"
