// Head of the reference's ResnetTransformerDetector (playaid/models/resnet_transformer_detector.py:41-93,141;
// SURVEY.md section 8f item 4): Linear(2048, 247) on the pooled ResNet-50 features, the 9-value time encoding of the
// frame's slot appended -> 256, three post-norm nn.TransformerEncoderLayer(d_model=256, nhead=8, dim_feedforward=2048,
// ReLU), Linear(256, A), log_softmax over the actions.
//
// The reference builds the encoder without batch_first and feeds it [B, S, 256] (:82-84), so torch takes dimension
// 0 -- the WINDOWS of the call -- as the sequence and the S frame slots as the batch: attention mixes the same slot
// of different windows. Reproduced as is: pa_encoder_forward takes (seq_len = windows, batch = slots) in torch's
// order, row r = l * batch + n.
//
// Side path (tens of rows): dense layers on the shared vector-unit linear kernel (lstm.hip), one wave per
// (slot, head) for the attention with an online softmax, one wave per row for LayerNorm. All fp32.
#include "pa_kernels.h"
#include "../../include/playaid_hip.h"
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

namespace pa {
namespace {

// X[r][hidden + j] = enc[(r % batch)][j]
__global__ void append_encoding_kernel(float* __restrict__ X, const float* __restrict__ enc, int rows, int batch, int D, int hidden, int enc_dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * enc_dim) return;
    const int r = i / enc_dim, j = i - r * enc_dim;
    X[(size_t)r * D + hidden + j] = enc[(r % batch) * enc_dim + j];
}

// Self-attention of one (batch element n, head h): queries / keys / values are rows l * batch + n of QKV [rows][3 D]
// at column offsets h * HD, D + h * HD, 2 D + h * HD. One wave; lane = query (strided over seq_len); online softmax.
template <int HD>
__global__ __launch_bounds__(64) void attention_kernel(const float* __restrict__ QKV, float* __restrict__ O, int seq_len, int batch, int D) {
    const int n = blockIdx.x, h = blockIdx.y;
    const float scale = 1.f / sqrtf((float)HD);
    for (int i = threadIdx.x; i < seq_len; i += 64) {
        const float* q = QKV + (size_t)(i * batch + n) * 3 * D + h * HD;
        float qr[HD], acc[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            qr[d] = q[d] * scale;  // torch scales the query before the product
            acc[d] = 0.f;
        }
        float m = -INFINITY, l = 0.f;
        for (int j = 0; j < seq_len; ++j) {
            const float* k = QKV + (size_t)(j * batch + n) * 3 * D + D + h * HD;
            const float* v = k + D;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) s = fmaf(qr[d], k[d], s);
            const float mn = fmaxf(m, s);
            const float corr = expf(m - mn), p = expf(s - mn);
            l = l * corr + p;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] = acc[d] * corr + p * v[d];
            m = mn;
        }
        float* o = O + (size_t)(i * batch + n) * D + h * HD;
        const float inv = 1.f / l;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] = acc[d] * inv;
    }
}

// X[r] = LayerNorm(X[r] + Y[r]) * g + b, eps 1e-5, one wave per row
__global__ __launch_bounds__(64) void add_layernorm_kernel(float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ g,
                                                           const float* __restrict__ b, int D) {
    const int r = blockIdx.x, lane = threadIdx.x;
    float* x = X + (size_t)r * D;
    const float* y = Y + (size_t)r * D;
    float sum = 0.f;
    for (int d = lane; d < D; d += 64) sum += x[d] + y[d];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum / (float)D;
    float var = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float t = x[d] + y[d] - mean;
        var += t * t;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) var += __shfl_xor(var, o, 64);
    const float rstd = 1.f / sqrtf(var / (float)D + 1e-5f);
    for (int d = lane; d < D; d += 64) x[d] = (x[d] + y[d] - mean) * rstd * g[d] + b[d];
}

// in-place log_softmax over the A values of a row
__global__ __launch_bounds__(64) void log_softmax_rows_kernel(float* __restrict__ X, int A) {
    float* x = X + (size_t)blockIdx.x * A;
    const int lane = threadIdx.x;
    float mx = -INFINITY;
    for (int a = lane; a < A; a += 64) mx = fmaxf(mx, x[a]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int a = lane; a < A; a += 64) sum += expf(x[a] - mx);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float lse = mx + logf(sum);
    for (int a = lane; a < A; a += 64) x[a] -= lse;
}

}  // namespace
}  // namespace pa

struct pa_encoder {
    int in_dim = 0, hidden = 0, slots = 0, enc_dim = 0, heads = 0, layers = 0, ff = 0, actions = 0, max_rows = 0, D = 0;
    float* weights = nullptr;
    float *ffn_w = nullptr, *ffn_b = nullptr, *enc = nullptr, *cls_w = nullptr, *cls_b = nullptr;
    float* ffn_pad = nullptr;  // [D][in_dim] + [D]: the front projection with zero rows for the encoding's columns, so that its
                               // N is whole matrix tiles (247 -> 256) and it runs on the MFMA kernel; append_encoding overwrites them
    struct Layer { float *in_w, *in_b, *out_w, *out_b, *l1_w, *l1_b, *l2_w, *l2_b, *n1_g, *n1_b, *n2_g, *n2_b; };
    std::vector<Layer> layer;
    float *x = nullptr, *qkv = nullptr, *att = nullptr, *y = nullptr, *f1 = nullptr;
    std::string last_error;
};

namespace {
size_t encoder_float_count(int in_dim, int hidden, int slots, int enc_dim, int layers, int ff, int actions) {
    const size_t D = (size_t)hidden + enc_dim;
    size_t n = (size_t)hidden * in_dim + hidden + (size_t)slots * enc_dim;
    n += (size_t)layers * (3 * D * D + 3 * D + D * D + D + (size_t)ff * D + ff + D * ff + D + 4 * D);
    n += (size_t)actions * D + actions;
    return n;
}
}  // namespace

extern "C" {

size_t pa_encoder_blob_bytes(int32_t in_dim, int32_t hidden_dim, int32_t slots, int32_t enc_dim, int32_t num_layers, int32_t ff_dim,
                             int32_t num_actions) {
    return 16 * sizeof(int32_t) + encoder_float_count(in_dim, hidden_dim, slots, enc_dim, num_layers, ff_dim, num_actions) * sizeof(float);
}

const char* pa_encoder_last_error(const pa_encoder* h) { return h ? h->last_error.c_str() : "null handle"; }

int pa_encoder_create(int32_t device, int32_t in_dim, int32_t hidden_dim, int32_t slots, int32_t enc_dim, int32_t num_heads,
                      int32_t num_layers, int32_t ff_dim, int32_t num_actions, int32_t max_rows, const void* blob_host, size_t blob_bytes,
                      pa_encoder** out) {
    if (!out) return PA_ERR_INVALID_ARG;
    *out = nullptr;
    const int D = hidden_dim + enc_dim;
    if (!blob_host || in_dim < 1 || hidden_dim < 1 || enc_dim < 0 || slots < 1 || num_heads < 1 || D % num_heads != 0 || D / num_heads != 32 ||
        num_layers < 1 || num_layers > 16 || ff_dim < 1 || num_actions < 1 || num_actions > 4096 || max_rows < 1)
        return PA_ERR_INVALID_ARG;  // (the attention kernel is instantiated for 32-wide heads: 256 / 8)
    const int32_t* hdr = reinterpret_cast<const int32_t*>(blob_host);
    if (blob_bytes != pa_encoder_blob_bytes(in_dim, hidden_dim, slots, enc_dim, num_layers, ff_dim, num_actions) || hdr[0] != PA_ENCODER_MAGIC ||
        hdr[1] != 1 || hdr[2] != in_dim || hdr[3] != hidden_dim || hdr[4] != slots || hdr[5] != enc_dim || hdr[6] != num_heads ||
        hdr[7] != num_layers || hdr[8] != ff_dim || hdr[9] != num_actions)
        return PA_ERR_BAD_WEIGHTS;
    pa_encoder* h = new pa_encoder();
    *out = h;
    h->in_dim = in_dim; h->hidden = hidden_dim; h->slots = slots; h->enc_dim = enc_dim; h->heads = num_heads; h->layers = num_layers;
    h->ff = ff_dim; h->actions = num_actions; h->max_rows = max_rows; h->D = D;
    auto chk = [&](hipError_t e, const char* what) -> bool {
        if (e == hipSuccess) return true;
        h->last_error = std::string(what) + ": " + hipGetErrorString(e);
        return false;
    };
    if (!chk(hipSetDevice(device), "hipSetDevice")) return PA_ERR_NO_DEVICE;
    const size_t nw = encoder_float_count(in_dim, hidden_dim, slots, enc_dim, num_layers, ff_dim, num_actions);
    if (!chk(hipMalloc(&h->weights, nw * sizeof(float)), "hipMalloc weights")) return PA_ERR_HIP;
    if (!chk(hipMemcpy(h->weights, hdr + 16, nw * sizeof(float), hipMemcpyHostToDevice), "upload weights")) return PA_ERR_HIP;
    float* p = h->weights;
    auto take = [&](size_t n) { float* r = p; p += n; return r; };
    const size_t Dz = D;
    h->ffn_w = take((size_t)hidden_dim * in_dim);
    h->ffn_b = take(hidden_dim);
    h->enc = take((size_t)slots * enc_dim);
    for (int l = 0; l < num_layers; ++l) {
        pa_encoder::Layer L;
        L.in_w = take(3 * Dz * Dz); L.in_b = take(3 * Dz);
        L.out_w = take(Dz * Dz); L.out_b = take(Dz);
        L.l1_w = take((size_t)ff_dim * Dz); L.l1_b = take(ff_dim);
        L.l2_w = take(Dz * ff_dim); L.l2_b = take(Dz);
        L.n1_g = take(Dz); L.n1_b = take(Dz); L.n2_g = take(Dz); L.n2_b = take(Dz);
        h->layer.push_back(L);
    }
    h->cls_w = take((size_t)num_actions * Dz);
    h->cls_b = take(num_actions);
    if (enc_dim > 0 && hidden_dim % 64 != 0 && D % 64 == 0) {
        const size_t nf = Dz * in_dim + Dz;
        if (!chk(hipMalloc(&h->ffn_pad, nf * sizeof(float)), "hipMalloc padded projection")) return PA_ERR_HIP;
        if (!chk(hipMemset(h->ffn_pad, 0, nf * sizeof(float)), "hipMemset")) return PA_ERR_HIP;
        if (!chk(hipMemcpy(h->ffn_pad, h->ffn_w, (size_t)hidden_dim * in_dim * sizeof(float), hipMemcpyDeviceToDevice), "copy projection")) return PA_ERR_HIP;
        if (!chk(hipMemcpy(h->ffn_pad + Dz * in_dim, h->ffn_b, (size_t)hidden_dim * sizeof(float), hipMemcpyDeviceToDevice), "copy projection bias")) return PA_ERR_HIP;
    }
    const size_t R = max_rows;
    if (!chk(hipMalloc(&h->x, R * Dz * sizeof(float)), "hipMalloc x")) return PA_ERR_HIP;
    if (!chk(hipMalloc(&h->qkv, R * 3 * Dz * sizeof(float)), "hipMalloc qkv")) return PA_ERR_HIP;
    if (!chk(hipMalloc(&h->att, R * Dz * sizeof(float)), "hipMalloc att")) return PA_ERR_HIP;
    if (!chk(hipMalloc(&h->y, R * Dz * sizeof(float)), "hipMalloc y")) return PA_ERR_HIP;
    if (!chk(hipMalloc(&h->f1, R * (size_t)ff_dim * sizeof(float)), "hipMalloc ff")) return PA_ERR_HIP;
    return PA_OK;
}

void pa_encoder_destroy(pa_encoder* h) {
    if (!h) return;
    (void)hipFree(h->weights);
    (void)hipFree(h->ffn_pad);
    (void)hipFree(h->x);
    (void)hipFree(h->qkv);
    (void)hipFree(h->att);
    (void)hipFree(h->y);
    (void)hipFree(h->f1);
    delete h;
}

int pa_encoder_forward(pa_encoder* h, const float* feats, int32_t ld, int32_t seq_len, int32_t batch, float* logp, void* stream) {
    if (!h) return PA_ERR_INVALID_ARG;
    auto bad = [&](int code, const char* msg) { h->last_error = msg; return code; };
    if (!feats || !logp || seq_len < 1 || batch < 1 || ld < h->in_dim) return bad(PA_ERR_INVALID_ARG, "pa_encoder_forward: bad argument");
    if (batch != h->slots) return bad(PA_ERR_INVALID_ARG, "pa_encoder_forward: batch must equal the number of encoded slots (sequence_length)");
    const long long rows_ll = (long long)seq_len * batch;
    if (rows_ll > h->max_rows) return bad(PA_ERR_CAPACITY, "pa_encoder_forward: seq_len * batch exceeds max_rows");
    hipStream_t s = (hipStream_t)stream;
    const int R = (int)rows_ll, D = h->D;
#define EN_HIP(call)                                                                                  \
    do {                                                                                              \
        hipError_t e__ = (call);                                                                      \
        if (e__ != hipSuccess) { h->last_error = std::string(#call) + ": " + hipGetErrorString(e__); return PA_ERR_HIP; } \
    } while (0)
    if (h->ffn_pad) EN_HIP(pa::launch_linear_f32(feats, ld, h->ffn_pad, h->ffn_pad + (size_t)D * h->in_dim, h->x, D, R, D, h->in_dim, 0, s));
    else EN_HIP(pa::launch_linear_f32(feats, ld, h->ffn_w, h->ffn_b, h->x, D, R, h->hidden, h->in_dim, 0, s));
    if (h->enc_dim > 0) {
        const int total = R * h->enc_dim;
        hipLaunchKernelGGL(pa::append_encoding_kernel, dim3((total + 255) / 256), dim3(256), 0, s, h->x, h->enc, R, batch, D, h->hidden, h->enc_dim);
        EN_HIP(hipGetLastError());
    }
    for (int l = 0; l < h->layers; ++l) {
        const pa_encoder::Layer& L = h->layer[l];
        EN_HIP(pa::launch_linear_f32(h->x, D, L.in_w, L.in_b, h->qkv, 3 * D, R, 3 * D, D, 0, s));
        hipLaunchKernelGGL(pa::attention_kernel<32>, dim3(batch, h->heads), dim3(64), 0, s, h->qkv, h->att, seq_len, batch, D);
        EN_HIP(hipGetLastError());
        EN_HIP(pa::launch_linear_f32(h->att, D, L.out_w, L.out_b, h->y, D, R, D, D, 0, s));
        hipLaunchKernelGGL(pa::add_layernorm_kernel, dim3(R), dim3(64), 0, s, h->x, h->y, L.n1_g, L.n1_b, D);
        EN_HIP(hipGetLastError());
        EN_HIP(pa::launch_linear_f32(h->x, D, L.l1_w, L.l1_b, h->f1, h->ff, R, h->ff, D, 1, s));
        EN_HIP(pa::launch_linear_f32(h->f1, h->ff, L.l2_w, L.l2_b, h->y, D, R, D, h->ff, 0, s));
        hipLaunchKernelGGL(pa::add_layernorm_kernel, dim3(R), dim3(64), 0, s, h->x, h->y, L.n2_g, L.n2_b, D);
        EN_HIP(hipGetLastError());
    }
    EN_HIP(pa::launch_linear_f32(h->x, D, h->cls_w, h->cls_b, logp, h->actions, R, h->actions, D, 0, s));
    hipLaunchKernelGGL(pa::log_softmax_rows_kernel, dim3(R), dim3(64), 0, s, logp, h->actions);
    EN_HIP(hipGetLastError());
#undef EN_HIP
    return PA_OK;
}

}  // extern "C"
