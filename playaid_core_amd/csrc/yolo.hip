// A fully convolutional detection network given as a layer table (the YOLOv5 detector the
// reference shells out to, playaid/ai_runner.py:191-224), run on the engine's fp32 implicit-GEMM kernel: frames in, the
// decoded head rows out -- what pa_detect_postprocess takes.
//
// The host (playaid_core_amd/yolov5.py) folds BatchNorm, lays the weights out [cout][ky][kx][cin] and describes every
// layer with a pa_net_layer. Activations are zero-bordered NHWC buffers; a layer reads / writes a CHANNEL SLICE of a
// buffer (offset + pixel pitch), which is how the network's concatenations cost nothing: the producers write straight
// into their slice of the consumer's input. Kinds:
//   0  convolution 1x1 | 3x3, stride 1 | 2, + bias, activation (none / ReLU / SiLU), residual before or after it
//      (implicit GEMM on the matrix cores; cin % 32 == 0, cout % 32 == 0). Layers without
//      a residual -- every 1x1 and stride-2 convolution -- run on the persistent engine (pigemm.hip), the others on igemm.hip
//   3  the 6x6 / 2 stem (3 -> 32 channels) on the letter-boxed RGB image + bias + SiLU: a direct convolution on the matrix
//      cores with K = 108 exactly (stem6x6_direct_kernel below; weights in its lane layout [64][56])
//   4  max-pool 5x5 / 1 (SPPF), slice to slice
//   5  nearest-neighbour 2x up-sampling, slice to slice
//   6  Detect decode of one scale: sigmoid, grid / anchor arithmetic -> rows (cx, cy, w, h, obj, classes) in net pixels
// Before the table runs, letterbox_kernel does what detect.py's LoadImages does to a frame: cv2.resize(INTER_LINEAR) to
// the un-padded size, a 114-grey border up to the network input (a multiple of 32), BGR -> RGB, / 255.
#include "pa_kernels.h"
#include "../../include/playaid_hip.h"
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace pa {
namespace {

// cv2.resize(src, (new_w, new_h), INTER_LINEAR) on 8-bit pixels: OpenCV's fixed-point bilinear resizer (11-bit
// coefficients, HResize into int, VResizeLinear: ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2 >> 2), then
// copyMakeBorder(114), BGR -> RGB, / 255. out: [n][net_h + 4][net_w + 4][4] fp32 with a 2-pixel ZERO border (the stem's
// padding) and channel 3 = 0.
// AS_INT_BF16 (the emulated-fp32 stem, stem6x6_bf16_kernel): the same pixels as INTEGERS 0..255 in bf16 -- [..][4] bf16, exact --
// the / 255 lives in that kernel's weights.
template <bool AS_INT_BF16>
__global__ __launch_bounds__(256) void letterbox_kernel(const uint8_t* __restrict__ frames, int n, int H, int W, int new_h, int new_w,
                                                        int top, int left, int net_h, int net_w, double scale_x, double scale_y,
                                                        float* __restrict__ out) {
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
            // (scale_x = 1.0 / ((double)new_w / W), scale_y likewise: computed once on the host, in double as OpenCV does)
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
            // A tap whose coefficient is zero is not fetched: at an exact 3:1 reduction -- 1080p to 640 x 360 -- the sample point
            // falls on a source pixel, a1 = b1 = 0 for every thread, and three byte loads remain of twelve.
            const uint8_t* r0 = f + (size_t)sy * W * 3;
            const uint8_t* r1 = f + (size_t)sy1 * W * 3;
            int S0[3], S1[3] = {0, 0, 0};
            for (int c = 0; c < 3; ++c) S0[c] = r0[sx * 3 + c] * a0;
            if (a1)
                for (int c = 0; c < 3; ++c) S0[c] += r0[sx1 * 3 + c] * a1;
            if (b1) {
                for (int c = 0; c < 3; ++c) S1[c] = r1[sx * 3 + c] * a0;
                if (a1)
                    for (int c = 0; c < 3; ++c) S1[c] += r1[sx1 * 3 + c] * a1;
            }
            for (int c = 0; c < 3; ++c) {
                int v = (((b0 * (S0[c] >> 4)) >> 16) + ((b1 * (S1[c] >> 4)) >> 16) + 2) >> 2;
                v = v < 0 ? 0 : (v > 255 ? 255 : v);
                rgb[2 - c] = v;
            }
        }
    }
    const size_t at = ((size_t)img * (net_h + 4) + y + 2) * (net_w + 4) + x + 2;
    if (AS_INT_BF16) {
        auto bf = [](int v) { return (unsigned)(__float_as_uint((float)v) >> 16); };   // (an integer below 256 has 8 significant bits: exact)
        reinterpret_cast<uint2*>(out)[at] = make_uint2(bf(rgb[0]) | (bf(rgb[1]) << 16), bf(rgb[2]));
    } else {
        reinterpret_cast<float4*>(out)[at] = make_float4((float)rgb[0] / 255.0f, (float)rgb[1] / 255.0f, (float)rgb[2] / 255.0f, 0.f);
    }
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

// SPPF's three cascaded 5x5 / 1 max-pools in one launch (models/common.py::SPPF: y1 = m(x), y2 = m(y1), y3 = m(y2)). With
// -inf padding a cascade of stride-1 max-pools is the max over the union of their windows: y1 = 5x5, y2 = 9x9, y3 = 13x13 of
// x, clipped at the border. One workgroup = one image x CG channels with the whole map in LDS: separable, the three
// horizontal maxima of every pixel, then the three vertical ones -> the three output slices. (Three launches of
// maxpool5_kernel: 28 us each on a 12 x 20 map -- launch latency, not work.)
template <int CG>
__global__ __launch_bounds__(256) void sppf_pools_kernel(const float* __restrict__ in, SliceGeom gi, float* __restrict__ out, SliceGeom g1,
                                                         SliceGeom g2, SliceGeom g3) {
    extern __shared__ float sm[];
    constexpr int Q = CG / 4;
    const int hw = gi.h * gi.w, img = blockIdx.y, c0 = blockIdx.x * CG;
    float4* xs = reinterpret_cast<float4*>(sm);   // [hw][Q]
    float4* h5 = xs + hw * Q;
    float4* h9 = h5 + hw * Q;
    float4* h13 = h9 + hw * Q;
    auto mx = [](float4 a, float4 b) { return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)); };
    for (int i = threadIdx.x; i < hw * Q; i += 256) {
        const int q = i % Q, pix = i / Q;
        xs[i] = *reinterpret_cast<const float4*>(in + px_off(gi, img, pix / gi.w, pix % gi.w) + c0 + q * 4);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < hw * Q; i += 256) {
        const int q = i % Q, pix = i / Q, y = pix / gi.w, x = pix % gi.w;
        float4 m = xs[i];
        float4 m5 = m, m9 = m, m13 = m;
#pragma unroll
        for (int d = 1; d <= 6; ++d) {
            if (x - d >= 0) m = mx(m, xs[i - d * Q]);
            if (x + d < gi.w) m = mx(m, xs[i + d * Q]);
            if (d == 2) m5 = m;
            if (d == 4) m9 = m;
        }
        m13 = m;
        h5[i] = m5; h9[i] = m9; h13[i] = m13;
        (void)y; (void)q;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < hw * Q; i += 256) {
        const int q = i % Q, pix = i / Q, y = pix / gi.w, x = pix % gi.w;
        const int row = gi.w * Q;
        float4 a = h5[i], b = h9[i], c = h13[i];
#pragma unroll
        for (int d = 1; d <= 6; ++d) {
            const bool up = y - d >= 0, dn = y + d < gi.h;
            if (d <= 2) { if (up) a = mx(a, h5[i - d * row]); if (dn) a = mx(a, h5[i + d * row]); }
            if (d <= 4) { if (up) b = mx(b, h9[i - d * row]); if (dn) b = mx(b, h9[i + d * row]); }
            if (up) c = mx(c, h13[i - d * row]);
            if (dn) c = mx(c, h13[i + d * row]);
        }
        *reinterpret_cast<float4*>(out + px_off(g1, img, y, x) + c0 + q * 4) = a;
        *reinterpret_cast<float4*>(out + px_off(g2, img, y, x) + c0 + q * 4) = b;
        *reinterpret_cast<float4*>(out + px_off(g3, img, y, x) + c0 + q * 4) = c;
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
// wh = (y * 2)^2 * anchor; rows of one scale in (anchor, y, x) order at row0. blockIdx.y = image x anchor, one thread per
// output VALUE of that slab: consecutive threads write consecutive floats of pred and read runs of `no` consecutive channels
// (32-bit index arithmetic: the slab of one image and anchor has h * w * no < 2^31 values).
__global__ __launch_bounds__(256) void detect_decode_kernel(const float* __restrict__ in, SliceGeom gi, int na, int no, float stride,
                                                            const float* __restrict__ anchors_px, float* __restrict__ pred, int rows_total,
                                                            int row0) {
    const unsigned t = blockIdx.x * 256u + threadIdx.x, hw = gi.h * gi.w;
    if (t >= hw * (unsigned)no) return;
    const int img = blockIdx.y / na, a = blockIdx.y - img * na;
    const unsigned pix = t / (unsigned)no, k = t - pix * no;
    const int y = pix / (unsigned)gi.w, x = pix - y * gi.w;
    const float v = in[px_off(gi, img, y, x) + a * no + k];
    const float s = 1.f / (1.f + expf(-v));
    float val = s;
    if (k == 0) val = (s * 2.f + ((float)x - 0.5f)) * stride;
    else if (k == 1) val = (s * 2.f + ((float)y - 0.5f)) * stride;
    else if (k == 2) val = (s * 2.f) * (s * 2.f) * anchors_px[a * 2];
    else if (k == 3) val = (s * 2.f) * (s * 2.f) * anchors_px[a * 2 + 1];
    pred[((size_t)img * rows_total + row0 + (size_t)a * hw) * no + t] = val;
}

// The 6x6 / 2 stem (models/yolov5s.yaml layer 0: Conv(3, 32, 6, 2, 2) + BatchNorm + SiLU) as a direct convolution on the
// matrix cores, operands straight from global memory -- no LDS, no barrier. One wave = a strip of 32 output columns x `rows`
// output rows x all 32 output channels:
//   * v_mfma_f32_32x32x2_f32 with the PIXELS as the row operand (a lane supplies output column lr) and the weights as the column
//     operand (a lane supplies channel lr); the k pair of a step is (kx = j, kx = 3 + j) of one (ky, channel): lanes 0-31 carry
//     the first, lanes 32-63 the second. K = 6 ky x 3 j x 3 channels x 2 = 108 exactly -- the im2col form of this layer (igemm.hip, one
//     8-pixel x 4-channel chunk per kernel row = K 192, 64 output channels) executed 3.6x the multiply-adds for the same result;
//   * a lane's 54 weights stay in registers for the whole kernel ([lane][ky][j][c], laid out by the host);
//   * per (input row, j) a lane reads ONE NHWC4 pixel (16 bytes: column 2 ox + 3 (lane >> 5) + j) and feeds its three
//     channels to three matrix instructions; the six input rows of an output row live in an eight-row register window that
//     slides by two rows per output row (six new 16-byte loads per 54 matrix instructions, issued one output row ahead);
//   * a lane ends up with ONE channel (lr) of 16 pixels (columns 8 g + 4 lh + e of the strip): bias + SiLU + 16 dword stores, each
//     writing that channel for 2 pixels x 32 lanes = two whole 128-byte lines. (Rounds 4-5 had the operands the other way round:
//     lane = pixel, four 16-byte stores each touching 32 lines -- 318 against 270 us per 64 frames, PA_STEM_LANE_IS_PIXEL below.)
struct StemDirectParams {
    const float* x;      // [n][net_h + 4][net_w + 4][4], 2-pixel zero border, channel 3 = 0
    const float* wlane;  // [64 lanes][56]: W[lane & 31][c][ky][3 * (lane >> 5) + j] at ky * 9 + j * 3 + c
    const float* bias;   // [32]
    float* out;          // slice of a zero-bordered NHWC buffer
    int32_t n, net_h, net_w, oh, ow;
    int32_t out_px_stride, out_row_stride, out_img_stride, out_pad;
    int32_t rows, row_blocks, col_blocks;  // rows % 4 == 0
};

// ROWS output rows per wave, fully unrolled and without a branch: hipcc's s_waitcnt placement is exact in straight-line code
// only -- with a row loop and a branch around the stores it waited for the row's own prefetch (vmcnt(2) behind six loads
// that nothing needed for another 54 matrix instructions), which cost the kernel half its matrix-pipe time. Rows and
// columns past the image are loaded from clamped addresses and their stores are dropped by the buffer's bounds check.
// A/B build -DPA_STEM_LANE_IS_PIXEL=1 (both stem kernels): the weights as the matrix instruction's ROW operand -- a lane ends up with
// one pixel and 16 bytes of it per store, every store instruction touching 32 cache lines. The product has the PIXELS as the row
// operand: a lane holds ONE channel of 16 pixels and a dword store writes that channel for 2 pixels x 32 lanes = two whole
// 128-byte lines (same products, same order along k: the same bits). Measured on the bf16 form: 218 -> 123 us
// (profiles/r06_detect_stem_bf16.txt).
#ifndef PA_STEM_LANE_IS_PIXEL
#define PA_STEM_LANE_IS_PIXEL 0
#endif
template <int ROWS>
__global__ __launch_bounds__(256, 2) void stem6x6_direct_kernel(const StemDirectParams p) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    static_assert(ROWS % 4 == 0, "the register window rotates with period four");
    const int lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
    const int strip = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (wave-uniform, and the compiler knows)
    const int strips = p.n * p.row_blocks * p.col_blocks;
    if (strip >= strips) return;  // (whole waves; the kernel has no barrier)
    const int cb = strip % p.col_blocks;
    const int rb = (strip / p.col_blocks) % p.row_blocks;
    const int img = strip / (p.col_blocks * p.row_blocks);
    const int ox = cb * 32 + lr, oy0 = rb * ROWS;
    const int oxc = ox < p.ow ? ox : p.ow - 1;
    const int in_w = p.net_w + 4, in_h = p.net_h + 4;
    const float* xin = p.x + ((size_t)img * in_h * in_w + 2 * oxc + 3 * lh) * 4;
    float w[56];
    {
        const f32x4* wl = reinterpret_cast<const f32x4*>(p.wlane + lane * 56);
#pragma unroll
        for (int i = 0; i < 14; ++i) {
            const f32x4 v = wl[i];
            w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
        }
    }
    f32x4 bias4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias4[g] = *reinterpret_cast<const f32x4*>(p.bias + 8 * g + 4 * lh);
    const float bias1 = p.bias[lr];
    // this image's slice of the output as a buffer: a store at offset >= num_records (0xFFFFFFFF here) goes nowhere
    const __amdgpu_buffer_rsrc_t out_rs =
        __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)img * p.out_img_stride, 0, p.out_img_stride * 4, 0x00020000);
    const unsigned col_off = ox < p.ow ? (unsigned)(((ox + p.out_pad) * p.out_px_stride + 4 * lh) * 4) : 0xFFFFFFFFu;

    f32x4 win[8][3];
    auto load_row = [&](int slot, int row) {  // input row `row` (clamped: rows past the image feed output rows nobody stores)
        const int rc = row < in_h ? row : in_h - 1;
        const float* r = xin + (size_t)rc * in_w * 4;
#pragma unroll
        for (int j = 0; j < 3; ++j) win[slot][j] = *reinterpret_cast<const f32x4*>(r + j * 4);
    };
#pragma unroll
    for (int ky = 0; ky < 6; ++ky) load_row(ky, 2 * oy0 + ky);
#pragma unroll
    for (int t = 0; t < ROWS; ++t) {
        const int u = t & 3, oy = oy0 + t;
        // the next output row's two new input rows, into the slots of the two rows this one no longer needs
        if (t + 1 < ROWS) {
            load_row((2 * u + 6) & 7, 2 * oy + 6);
            load_row((2 * u + 7) & 7, 2 * oy + 7);
        }
        __builtin_amdgcn_sched_barrier(0);  // (hipcc otherwise sinks the loads to their first use, one output row later)
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 6; ++ky)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const f32x4 px = win[(2 * u + ky) & 7][j];
                if (PA_STEM_LANE_IS_PIXEL) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ky * 9 + j * 3 + 0], px.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ky * 9 + j * 3 + 1], px.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ky * 9 + j * 3 + 2], px.z, acc, 0, 0, 0);
                } else {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(px.x, w[ky * 9 + j * 3 + 0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(px.y, w[ky * 9 + j * 3 + 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(px.z, w[ky * 9 + j * 3 + 2], acc, 0, 0, 0);
                }
            }
        __builtin_amdgcn_sched_barrier(0);
        if (!PA_STEM_LANE_IS_PIXEL) {
            // register 4 g + e: pixel column cb * 32 + 8 g + 4 lh + e, channel lr
            const int px0 = cb * 32 + 4 * lh;
            const unsigned o0 = (unsigned)(((px0 + p.out_pad) * p.out_px_stride + lr) * 4 + (oy + p.out_pad) * p.out_row_stride * 4);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = silu_fast(acc[4 * g + e] + bias1);
                    const bool live = oy < p.oh && px0 + 8 * g + e < p.ow;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), out_rs, live ? o0 + (unsigned)((8 * g + e) * p.out_px_stride * 4) : 0xFFFFFFFFu, 0, 0);
                }
            continue;
        }
        const unsigned row_off = oy < p.oh ? col_off + (unsigned)((oy + p.out_pad) * p.out_row_stride * 4) : 0xFFFFFFFFu;
        const unsigned off = col_off == 0xFFFFFFFFu ? 0xFFFFFFFFu : row_off;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]} + bias4[g];
            v.x = silu_fast(v.x); v.y = silu_fast(v.y);
            v.z = silu_fast(v.z); v.w = silu_fast(v.w);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), out_rs,
                                                   off == 0xFFFFFFFFu ? off : off + 32u * g, 0, 0);
        }
    }
}

// The same stem under PA_DTYPE_EMULATED_F32, on v_mfma_f32_32x32x16_bf16. The letter-boxed pixels are INTEGERS (cv2.resize's
// 8-bit output, detect.py divides them by 255 afterwards): an integer below 256 is ONE exact bf16 value, so the pixel operand
// needs no split at all and a product takes three matrix instructions -- the three bf16 slices of W / 255 -- instead of psgemm's
// six. (The generic emulated form of this layer, a six-tap implicit GEMM with the pixels split in registers, was 1.42x SLOWER
// than the exact kernel: 32 output channels give a split nothing to hide under. profiles/r06_detect_stem_emulated_ab.txt.)
//   * K = 6 ky x 6 kx x 4 (channel 3: zero weights) = 144 = nine k-steps of 16; a lane's eight consecutive k are TWO input pixels
//     = 16 bytes of the bf16 NHWC4 input, one load; lanes 0-31 take the even runs of eight, lanes 32-63 the odd ones;
//   * the 27 weight fragments (9 steps x 3 slices) stay in registers for the whole kernel, laid out by pa_detector_create;
//   * a lane's nine runs of an output row are a window that slides by three per output row (run r of row t + 1 = run r + 6 of
//     row t): three new 16-byte loads per 27 matrix instructions, issued three output rows ahead (18 register slots);
//   * accumulators start from the bias; SiLU and the four 16-byte stores as in the exact kernel.
// Against the reference's fl(k / 255) * w this computes k * (w / 255 to 24 bits): both one rounding away from the real product.
struct StemBf16Params {
    const unsigned short* x;      // [n][net_h + 4][net_w + 4][4] bf16 integers, 2-pixel zero border, channel 3 = 0
    const unsigned short* wfrag;  // [9 steps][3 slices][64 lanes][8]: slice of W[lane & 31][k] / 255, k = 16 s + 8 (lane >> 5) + i = 24 ky + 4 kx + c
    const float* bias;            // [32]
    float* out;
    int32_t n, net_h, net_w, oh, ow;
    int32_t out_px_stride, out_row_stride, out_img_stride, out_pad;
    int32_t row_blocks, col_blocks;
};

// timing experiments only (diagnostic build -DPA_STEM_ABL=n, results wrong): 1 = no matrix instructions, 2 = no SiLU, 4 = no stores
#ifndef PA_STEM_ABL
#define PA_STEM_ABL 0
#endif
template <int ROWS>
__global__ __launch_bounds__(256, 2) void stem6x6_bf16_kernel(const StemBf16Params p) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    // three new runs per output row, requested AHEAD output rows early (a row is only 27 matrix instructions, ~900 cycles, long)
    constexpr int AHEAD = 3, NSLOT = 9 + 3 * AHEAD;
    static_assert((3 * ROWS) % NSLOT == 0, "the run window rotates by three slots per output row");
    const int lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int strips = p.n * p.row_blocks * p.col_blocks;
    const int in_w = p.net_w + 4, in_h = p.net_h + 4;
    // the 27 weight fragments: loaded ONCE per wave, the wave then walks strips (the grid is what fits the chip, not one
    // workgroup per four strips: 10 240 strips would fetch the 27 KB 10 240 times and pay 10 240 window prologues cold)
    u32x4 wf[9][3];
#pragma unroll
    for (int st = 0; st < 9; ++st)
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) wf[st][sl] = *reinterpret_cast<const u32x4*>(p.wfrag + ((size_t)(st * 3 + sl) * 64 + lane) * 8);
    // The PIXELS are the matrix instruction's row operand here and the weights its column operand (the exact kernel has it the other
    // way round): a lane then ends up with ONE output channel (lr) of 16 pixels (8 g + 4 lh + e of the strip's 32 columns), and a
    // store instruction writes that channel for 2 pixels x 32 lanes = two whole 128-byte lines. With the roles as in the exact kernel
    // -- lane = pixel, 16 bytes of it per store -- every store touched 32 lines, 32 bytes of each, and the stores alone were 95 of
    // the kernel's 215 us (ablation builds, profiles/r06_detect_stem_bf16.txt).
    const float bias = p.bias[lr];
    // step st of this lane: run r = 2 st + lh = 3 ky + m -> input row 2 oy + ky, pixels 2 ox + 2 m and 2 ox + 2 m + 1
    int ky_of[9], m_of[9];
#pragma unroll
    for (int st = 0; st < 9; ++st) {
        const int r = 2 * st + lh;
        ky_of[st] = r / 3;
        m_of[st] = r - 3 * ky_of[st];
    }
    for (int strip = blockIdx.x * 4 + wave; strip < strips; strip += gridDim.x * 4) {   // (whole waves; the kernel has no barrier)
        const int cb = strip % p.col_blocks;
        const int rb = (strip / p.col_blocks) % p.row_blocks;
        const int img = strip / (p.col_blocks * p.row_blocks);
        const int ox = cb * 32 + lr, oy0 = rb * ROWS;
        const int oxc = ox < p.ow ? ox : p.ow - 1;
        const unsigned short* xin = p.x + ((size_t)img * in_h * in_w + 2 * oxc) * 4;
        // this image's slice of the output as a buffer: a store at offset >= num_records goes nowhere
        const __amdgpu_buffer_rsrc_t out_rs =
            __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)img * p.out_img_stride, 0, p.out_img_stride * 4, 0x00020000);
        // accumulator register 4 g + e of this lane: pixel column cb * 32 + 8 g + 4 lh + e, channel lr
        const int px0 = cb * 32 + 4 * lh;
        const unsigned col0 = (unsigned)(((px0 + p.out_pad) * p.out_px_stride + lr) * 4);
        u32x4 win[NSLOT];
        auto load_run = [&](int slot, int st, int oy) {  // (rows past the image: clamped, they feed output rows nobody stores)
            const int row = 2 * oy + ky_of[st];
            const int rc = row < in_h ? row : in_h - 1;
            win[slot] = *reinterpret_cast<const u32x4*>(xin + ((size_t)rc * in_w + 2 * m_of[st]) * 4);
        };
#pragma unroll
        for (int st = 0; st < 9; ++st) load_run(st, st, oy0);
#pragma unroll
        for (int a = 1; a < AHEAD; ++a)
#pragma unroll
            for (int st = 6; st < 9; ++st) load_run((3 * a + st) % NSLOT, st, oy0 + a);
#pragma unroll
        for (int t = 0; t < ROWS; ++t) {
            const int base = (3 * t) % NSLOT, oy = oy0 + t;
            // the three new runs (steps 6, 7, 8) of the output row AHEAD rows on, into slots no row before it still reads
            if (t + AHEAD < ROWS) {
#pragma unroll
                for (int st = 6; st < 9; ++st) load_run((base + 3 * AHEAD + st) % NSLOT, st, oy + AHEAD);
            }
            __builtin_amdgcn_sched_barrier(0);  // (as in the exact kernel: keep the loads ahead of their use)
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = bias;
#pragma unroll
            for (int sl = 2; sl >= 0; --sl)   // smallest slice first
#pragma unroll
                for (int st = 0; st < 9; ++st) {
                    if (PA_STEM_ABL & 1) { acc[0] += __builtin_bit_cast(f32x4, win[(base + st) % NSLOT]).x * __builtin_bit_cast(f32x4, wf[st][sl]).x; continue; }
                    if (PA_STEM_LANE_IS_PIXEL)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[st][sl]), __builtin_bit_cast(bf16x8, win[(base + st) % NSLOT]), acc, 0, 0, 0);
                    else
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, win[(base + st) % NSLOT]), __builtin_bit_cast(bf16x8, wf[st][sl]), acc, 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (PA_STEM_LANE_IS_PIXEL) {   // (bias added here: the accumulators started from one channel's value)
                const unsigned coff = ox < p.ow && oy < p.oh ? (unsigned)(((ox + p.out_pad) * p.out_px_stride + 4 * lh) * 4 + (oy + p.out_pad) * p.out_row_stride * 4) : 0xFFFFFFFFu;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + 8 * g + 4 * lh);
                    f32x4 v = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]} - f32x4{bias, bias, bias, bias} + b4;
                    v.x = silu_fast(v.x); v.y = silu_fast(v.y); v.z = silu_fast(v.z); v.w = silu_fast(v.w);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), out_rs,
                                                           coff == 0xFFFFFFFFu ? coff : coff + 32u * g, 0, 0);
                }
                continue;
            }
            const unsigned row_off = col0 + (unsigned)((oy + p.out_pad) * p.out_row_stride * 4);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[4 * g + e];
                    if (!(PA_STEM_ABL & 2)) v = silu_fast(v);
                    if ((PA_STEM_ABL & 4) && v != 12345.678f) continue;
                    const bool live = oy < p.oh && px0 + 8 * g + e < p.ow;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), out_rs,
                                                          live ? row_off + (unsigned)((8 * g + e) * p.out_px_stride * 4) : 0xFFFFFFFFu, 0, 0);
                }
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
    float* wino_weights = nullptr;      // the stride-1 3x3 layers' filters in the Winograd kernel's layout (wino.hip)
    std::vector<long long> wino_off;    // per layer: float offset into wino_weights, -1 = the layer runs in its direct form
    std::vector<int> wino_bn;           // per layer: output channels per workgroup its filters were laid out for
    int compute_dtype = PA_DTYPE_F32;   // PA_DTYPE_EMULATED_F32: the layers listed in split_off run on psgemm.hip
    unsigned short* stem_frag = nullptr;       // PA_DTYPE_EMULATED_F32: the stem's W / 255 as stem6x6_bf16_kernel's 27 register fragments (x0 then holds bf16 integers)
    unsigned short* split_weights = nullptr;   // those layers' weights as three bf16 slices in the kernel's stage-image order
    std::vector<long long> split_off;   // per layer: element offset into split_weights, -1 = the layer keeps its exact fp32 kernel
    hipStream_t side = nullptr;         // PA_DET_LANES=2: the second half batch's stream
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    float* x0 = nullptr;       // letter-boxed input [max_images][net_h + 4][net_w + 4][4]
    float* anchors = nullptr;  // device copy of the decode layers' anchors [n_decode][8]
    std::string last_error;
};

extern "C" {

const char* pa_detector_last_error(const pa_detector* h) { return h ? h->last_error.c_str() : "null handle"; }

void pa_detector_destroy(pa_detector* h) {
    if (!h) return;
    (void)hipFree(h->weights);
    (void)hipFree(h->wino_weights);
    (void)hipFree(h->split_weights);
    (void)hipFree(h->stem_frag);
    (void)hipFree(h->x0);
    (void)hipFree(h->anchors);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->side) (void)hipStreamDestroy(h->side);
    for (float* b : h->bufs) (void)hipFree(b);
    delete h;
}

int pa_detector_rows(const pa_detector* h) { return h ? h->rows : 0; }

int pa_detector_create(int32_t device, const pa_net_layer* layers, int32_t n_layers, const int64_t* buf_floats_per_image, int32_t n_bufs,
                       const float* weights_host, size_t n_weights, int32_t max_images, int32_t net_h, int32_t net_w, int32_t num_classes,
                       pa_detector** out) {
    return pa_detector_create_dtype(device, layers, n_layers, buf_floats_per_image, n_bufs, weights_host, n_weights, max_images, net_h, net_w, num_classes,
                                    PA_DTYPE_F32, out);
}

int pa_detector_create_dtype(int32_t device, const pa_net_layer* layers, int32_t n_layers, const int64_t* buf_floats_per_image, int32_t n_bufs,
                             const float* weights_host, size_t n_weights, int32_t max_images, int32_t net_h, int32_t net_w, int32_t num_classes,
                             int32_t compute_dtype, pa_detector** out) {
    if (!out) return PA_ERR_INVALID_ARG;
    *out = nullptr;
    if (!layers || n_layers < 1 || !buf_floats_per_image || n_bufs < 1 || !weights_host || n_weights < 1 || max_images < 1 || net_h < 32 ||
        net_w < 32 || net_h % 32 || net_w % 32 || num_classes < 1 || num_classes > 80 ||
        (compute_dtype != PA_DTYPE_F32 && compute_dtype != PA_DTYPE_EMULATED_F32))
        return PA_ERR_INVALID_ARG;
    pa_detector* h = new pa_detector();
    *out = h;
    h->device = device; h->max_images = max_images; h->net_h = net_h; h->net_w = net_w; h->nc = num_classes;
    h->compute_dtype = compute_dtype;
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
            if ((L.ksize != 1 && L.ksize != 3) || (L.stride != 1 && L.stride != 2) || L.cin % 32 || L.cout % 32 || L.in_h % L.stride ||
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
            if (L.in_h != net_h || L.in_w != net_w || L.cout != 32 || L.w_off < 0 || (size_t)L.w_off + (size_t)64 * 56 > n_weights || (L.w_off & 3) ||
                L.b_off < 0 || (size_t)L.b_off + L.cout > n_weights || (L.b_off & 3) ||
                !slice_ok(L.out_buf, net_h / 2, net_w / 2, L.out_pad, L.out_cstride, L.out_coff, L.cout))
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
    {
        // stride-1 3x3 convolutions on maps whose sides are multiples of four run as Winograd F(2x2, 3x3) (wino.hip: 4 / 9 of
        // the direct form's multiply-adds, the same fp32 matrix instructions); their filters are transformed here, once, in
        // fp64. PA_DET_WINO=0 keeps the direct patch-resident kernel (A/B)
        // (read per create, not once per process: scripts/yolov5_parity.py makes detectors with one layer at a time in this form
        // to see which of them moves the boxes; PA_DET_WINO_MASK = bit k set <=> the k-th eligible layer runs as Winograd)
        const int use_wino = getenv("PA_DET_WINO") ? atoi(getenv("PA_DET_WINO")) : 1;
        const unsigned long wino_mask = getenv("PA_DET_WINO_MASK") ? strtoul(getenv("PA_DET_WINO_MASK"), nullptr, 0) : ~0ul;
        h->wino_off.assign(n_layers, -1);
        h->wino_bn.assign(n_layers, 0);
        size_t total = 0;
        int eligible = 0;
        for (int i = 0; i < n_layers; ++i) {
            const pa_net_layer& L = h->layers[i];
            if (use_wino && L.kind == 0 && L.ksize == 3 && L.stride == 1 && L.in_pad == 1 && L.in_h % 4 == 0 && L.in_w % 4 == 0 && L.cin % 8 == 0 &&
                ((wino_mask >> (eligible++ & 63)) & 1ul)) {
                h->wino_off[i] = (long long)total;
                h->wino_bn[i] = pa::wino_pick_bn(L.cout, (long long)max_images * (L.in_h / 4) * (L.in_w / 4));
                total += pa::wino_weight_floats(L.cin, L.cout);
            }
        }
        if (total) {
            std::vector<float> ug(total);
            for (int i = 0; i < n_layers; ++i)
                if (h->wino_off[i] >= 0) pa::wino_transform_weights(weights_host + h->layers[i].w_off, h->layers[i].cin, h->layers[i].cout, h->wino_bn[i], ug.data() + h->wino_off[i]);
            if (!chk(hipMalloc(&h->wino_weights, total * sizeof(float)), "hipMalloc Winograd filters")) return PA_ERR_HIP;
            if (!chk(hipMemcpy(h->wino_weights, ug.data(), total * sizeof(float), hipMemcpyHostToDevice), "upload Winograd filters")) return PA_ERR_HIP;
        }
    }
    h->split_off.assign(n_layers, -1);
    if (compute_dtype == PA_DTYPE_EMULATED_F32) {
        // the 1x1 and the stride-2 3x3 convolutions (and, with PA_DET_EMU_S1=1, the stride-1 3x3 ones in place of their Winograd
        // form) on the emulated-fp32 kernel (psgemm.hip): weights split into three bf16 slices here, once
        const int emu_s1 = getenv("PA_DET_EMU_S1") ? atoi(getenv("PA_DET_EMU_S1")) : 0;
        // ... and the 6x6 / 2 stem on stem6x6_bf16_kernel: integer pixels as ONE exact bf16 value each, W / 255 as three bf16 slices
        // (PA_DET_EMU_STEM=0: the exact direct stem, A/B). The generic emulated form -- a six-tap implicit GEMM on psgemm.hip with the
        // pixels split in registers -- lived here until commit dd6847f and was 1.42x slower than the exact kernel
        // (profiles/r06_detect_stem_emulated_ab.txt).
        const int emu_stem = getenv("PA_DET_EMU_STEM") ? atoi(getenv("PA_DET_EMU_STEM")) : 1;
        size_t total = 0;
        for (int i = 0; i < n_layers; ++i) {
            const pa_net_layer& L = h->layers[i];
            if (L.kind == 3 && emu_stem && L.cout == 32 && L.ksize == 6 && L.stride == 2 && !h->stem_frag) {
                // lane layout of the direct kernel: W[ch][c][ky][3 half + j] sits at half * cout * 56 + ch * 56 + ky * 9 + j * 3 + c.
                // -> [step][slice][lane][8]: lane = 32 khalf + ch supplies k = 16 step + 8 khalf + i, k = 24 ky + 4 kx + c
                std::vector<unsigned short> frag((size_t)9 * 3 * 64 * 8, 0);
                auto rne = [](double v) {   // bf16 nearest-even of a double, and its value
                    float f = (float)v;
                    uint32_t u;
                    memcpy(&u, &f, 4);
                    u += 0x7fffu + ((u >> 16) & 1u);
                    return (unsigned short)(u >> 16);
                };
                auto val = [](unsigned short hq) { const uint32_t u = (uint32_t)hq << 16; float f; memcpy(&f, &u, 4); return (double)f; };
                for (int st = 0; st < 9; ++st)
                    for (int ln = 0; ln < 64; ++ln)
                        for (int e = 0; e < 8; ++e) {
                            const int ch = ln & 31, k = 16 * st + 8 * (ln >> 5) + e, ky = k / 24, kx = (k % 24) / 4, c = k % 4;
                            if (c == 3) continue;
                            double r = (double)weights_host[L.w_off + (size_t)(kx / 3) * L.cout * 56 + (size_t)ch * 56 + ky * 9 + (kx % 3) * 3 + c] / 255.0;
                            for (int sl = 0; sl < 3; ++sl) {
                                const unsigned short hq = rne(r);
                                frag[((size_t)(st * 3 + sl) * 64 + ln) * 8 + e] = hq;
                                r -= val(hq);
                            }
                        }
                if (!chk(hipMalloc(&h->stem_frag, frag.size() * sizeof(unsigned short)), "hipMalloc stem fragments")) return PA_ERR_HIP;
                if (!chk(hipMemcpy(h->stem_frag, frag.data(), frag.size() * sizeof(unsigned short), hipMemcpyHostToDevice), "upload stem fragments")) return PA_ERR_HIP;
                continue;
            }
            if (L.kind != 0 || L.cin % 32 || L.cout % 32) continue;
            if (L.ksize == 3 && L.stride == 1 && !emu_s1 && h->wino_off[i] >= 0) continue;
            const size_t n_el = pa::psgemm_weight_elems(L.cout, L.ksize * L.ksize * L.cin, L.res_buf >= 0);
            if (n_el == 0) continue;
            h->split_off[i] = (long long)total;
            total += n_el;
        }
        if (total) {
            std::vector<unsigned short> sw(total);
            for (int i = 0; i < n_layers; ++i)
                if (h->split_off[i] >= 0) {
                    const pa_net_layer& L = h->layers[i];
                    pa::psgemm_pack_weights(weights_host + L.w_off, L.cout, L.ksize * L.ksize * L.cin, L.res_buf >= 0, sw.data() + h->split_off[i]);
                }
            if (!chk(hipMalloc(&h->split_weights, total * sizeof(unsigned short)), "hipMalloc split weights")) return PA_ERR_HIP;
            if (!chk(hipMemcpy(h->split_weights, sw.data(), total * sizeof(unsigned short), hipMemcpyHostToDevice), "upload split weights")) return PA_ERR_HIP;
        }
    }
    h->bufs.assign(n_bufs, nullptr);
    h->buf_floats.assign(buf_floats_per_image, buf_floats_per_image + n_bufs);
    for (int b = 0; b < n_bufs; ++b) {
        // the convolution kernels address a buffer with 32-bit BYTE offsets (buffer_load ... lds): 2 GB per buffer
        if ((unsigned long long)max_images * (unsigned long long)h->buf_floats[b] + 128ull * 2048 >= (1ull << 29)) {
            h->last_error = "activation buffer " + std::to_string(b) + " would exceed 2 GB: lower max_images (frames are run in chunks of it)";
            return PA_ERR_CAPACITY;
        }
        // (+ slack: a partial last tile of the GEMM reads rows past the last image)
        const size_t bytes = ((size_t)max_images * h->buf_floats[b] + 128 * 2048) * sizeof(float);
        if (!chk(hipMalloc(&h->bufs[b], bytes), "hipMalloc activations")) return PA_ERR_HIP;
        if (!chk(hipMemset(h->bufs[b], 0, bytes), "hipMemset activations")) return PA_ERR_HIP;  // the zero borders stay zero
    }
    // (+ slack: the stem's 8-pixel K chunks of the last row run two pixels past it, a partial last GEMM tile further)
    const size_t x0_bytes = ((size_t)max_images * (net_h + 4) * (net_w + 4) * 4 + 128 * 2048) * sizeof(float);
    if (x0_bytes >= (1ull << 31)) {
        h->last_error = "the letter-boxed input would exceed 2 GB: lower max_images";
        return PA_ERR_CAPACITY;
    }
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

// i0: the first image's slot in the handle's buffers (frames / pred point at that image's data): two half batches can run on two
// streams side by side, each in its own image range of the same buffers (pa_detector_forward)
// [lbeg, lend): the layers to run (lend < 0: to the end; the letterbox belongs to layer 0) -- pa_detector_forward's blocked order.
static int detector_run(pa_detector* h, const uint8_t* frames, int32_t n, int32_t height, int32_t width, float* pred, void* stream,
                        std::vector<hipEvent_t>* ev, int i0 = 0, int lbeg = 0, int lend = -1) {
    if (!h) return PA_ERR_INVALID_ARG;
    auto fail = [&](int code, const std::string& msg) { h->last_error = msg; return code; };
    if (!frames || !pred || n < 1 || height < 1 || width < 1) return fail(PA_ERR_INVALID_ARG, "pa_detector_forward: bad argument");
    if (i0 < 0 || i0 + n > h->max_images) return fail(PA_ERR_CAPACITY, "pa_detector_forward: more images than max_images");
    hipStream_t s = (hipStream_t)stream;
    float* const X0 = h->x0 + (size_t)i0 * (h->net_h + 4) * (h->net_w + 4) * 4;
    unsigned short* const X0B = reinterpret_cast<unsigned short*>(h->x0) + (size_t)i0 * (h->net_h + 4) * (h->net_w + 4) * 4;   // the same buffer as bf16 (stem_frag)
    auto BUF = [&](int b) -> float* { return h->bufs[b] + (size_t)i0 * h->buf_floats[b]; };
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
    const double lb_scale_x = 1.0 / ((double)new_w / width), lb_scale_y = 1.0 / ((double)new_h / height);   // cv2.resize's inv_scale, inverted
    if (lbeg == 0) {
        if (h->stem_frag)   // (bf16 integers, four per pixel: image i0 starts half as many BYTES in)
            hipLaunchKernelGGL(pa::letterbox_kernel<true>, dim3((h->net_w + 63) / 64, (h->net_h + 3) / 4, n), dim3(256), 0, s, frames, n, height, width, new_h,
                               new_w, top, left, h->net_h, h->net_w, lb_scale_x, lb_scale_y, reinterpret_cast<float*>(X0B));
        else
            hipLaunchKernelGGL(pa::letterbox_kernel<false>, dim3((h->net_w + 63) / 64, (h->net_h + 3) / 4, n), dim3(256), 0, s, frames, n, height, width, new_h,
                               new_w, top, left, h->net_h, h->net_w, lb_scale_x, lb_scale_y, X0);
        DT_HIP(hipGetLastError());
    }
    const int no = 5 + h->nc;
    int row0 = 0, di = 0;
    for (int li = 0; li < lbeg; ++li)
        if (h->layers[li].kind == 6) {   // (the Detect rows in front of this range)
            row0 += 3 * h->layers[li].in_h * h->layers[li].in_w;
            ++di;
        }
    const size_t l_end = lend < 0 ? h->layers.size() : (size_t)lend;
    for (size_t li = (size_t)lbeg; li < l_end; ++li) {
        const pa_net_layer& L = h->layers[li];
        if (ev) DT_HIP(hipEventRecord((*ev)[li], s));  // (profiling call only: layer li runs between events li and li + 1)
        if (L.kind == 3 && h->stem_frag) {   // PA_DTYPE_EMULATED_F32: integer pixels, three bf16 slices of W / 255
            const int oh = h->net_h / 2, ow = h->net_w / 2;
            pa::StemBf16Params q;
            q.x = X0B;
            q.wfrag = h->stem_frag;
            q.bias = h->weights + L.b_off;
            q.out = BUF(L.out_buf) + L.out_coff;
            q.n = n; q.net_h = h->net_h; q.net_w = h->net_w; q.oh = oh; q.ow = ow;
            q.out_px_stride = L.out_cstride;
            q.out_row_stride = (ow + 2 * L.out_pad) * L.out_cstride;
            q.out_img_stride = (oh + 2 * L.out_pad) * (ow + 2 * L.out_pad) * L.out_cstride;
            q.out_pad = L.out_pad;
            q.row_blocks = (oh + 11) / 12;
            q.col_blocks = (ow + 31) / 32;
            if ((long long)q.out_img_stride * 4 >= (1ll << 32)) return fail(PA_ERR_INVALID_ARG, "stem: output image larger than a buffer descriptor spans");
            const long long strips = (long long)n * q.row_blocks * q.col_blocks;
            // two workgroups per CU (225 registers), each wave walking strips
            static const int cus = [] { hipDeviceProp_t pr; int d = 0; return hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess ? pr.multiProcessorCount : 256; }();
            const long long wgs = (strips + 3) / 4;
            hipLaunchKernelGGL(pa::stem6x6_bf16_kernel<12>, dim3((unsigned)(wgs < 2ll * cus ? wgs : 2ll * cus)), dim3(256), 0, s, q);
            DT_HIP(hipGetLastError());
            continue;
        }
        if (L.kind == 3) {
            const int oh = h->net_h / 2, ow = h->net_w / 2;
            pa::StemDirectParams q;
            q.x = X0;
            q.wlane = h->weights + L.w_off;
            q.bias = h->weights + L.b_off;
            q.out = BUF(L.out_buf) + L.out_coff;
            q.n = n; q.net_h = h->net_h; q.net_w = h->net_w; q.oh = oh; q.ow = ow;
            q.out_px_stride = L.out_cstride;
            q.out_row_stride = (ow + 2 * L.out_pad) * L.out_cstride;
            q.out_img_stride = (oh + 2 * L.out_pad) * (ow + 2 * L.out_pad) * L.out_cstride;
            q.out_pad = L.out_pad;
            q.rows = 12;  // 64 x 384 x 640: 10240 strips = five rounds of two waves per SIMD
            q.row_blocks = (oh + q.rows - 1) / q.rows;
            q.col_blocks = (ow + 31) / 32;
            if ((long long)q.out_img_stride * 4 >= (1ll << 32)) return fail(PA_ERR_INVALID_ARG, "stem: output image larger than a buffer descriptor spans");
            const long long strips = (long long)n * q.row_blocks * q.col_blocks;
            hipLaunchKernelGGL(pa::stem6x6_direct_kernel<12>, dim3((unsigned)((strips + 3) / 4)), dim3(256), 0, s, q);
            DT_HIP(hipGetLastError());
            continue;
        }
        if (L.kind == 4 && li + 2 < h->layers.size()) {
            // SPPF: three max-pools in a row, each reading the slice the one before wrote, all in one buffer -> one launch
            const pa_net_layer& L2 = h->layers[li + 1];
            const pa_net_layer& L3 = h->layers[li + 2];
            auto chained = [](const pa_net_layer& a, const pa_net_layer& b) {
                return b.kind == 4 && b.in_buf == a.out_buf && b.in_coff == a.out_coff && b.in_pad == a.out_pad && b.in_cstride == a.out_cstride &&
                       b.in_h == a.in_h && b.in_w == a.in_w && b.cin == a.cin && b.out_buf == a.out_buf;
            };
            static const int fuse = getenv("PA_DET_SPPF") ? atoi(getenv("PA_DET_SPPF")) : 1;  // 0: three launches (A/B)
            const int hw = L.in_h * L.in_w;
            if (fuse && chained(L, L2) && chained(L2, L3) && L.out_buf == L.in_buf && L.cin % 16 == 0 && hw <= 480) {
                static const int cg_force = getenv("PA_DET_SPPF_CG") ? atoi(getenv("PA_DET_SPPF_CG")) : 0;   // A/B
                const int cg = cg_force ? cg_force : (hw <= 240 ? 16 : 8);   // four float arrays of the map x cg channels in <= 60 KB of LDS
                pa::SliceGeom gi = {L.in_h, L.in_w, L.in_pad, L.in_cstride, L.in_coff};
                pa::SliceGeom g1 = {L.in_h, L.in_w, L.out_pad, L.out_cstride, L.out_coff};
                pa::SliceGeom g2 = {L.in_h, L.in_w, L2.out_pad, L2.out_cstride, L2.out_coff};
                pa::SliceGeom g3 = {L.in_h, L.in_w, L3.out_pad, L3.out_cstride, L3.out_coff};
                const size_t lds = (size_t)4 * hw * cg * sizeof(float);
                if (cg == 16)
                    hipLaunchKernelGGL(pa::sppf_pools_kernel<16>, dim3(L.cin / 16, n), dim3(256), lds, s, BUF(L.in_buf), gi, BUF(L.out_buf), g1, g2, g3);
                else
                    hipLaunchKernelGGL(pa::sppf_pools_kernel<8>, dim3(L.cin / 8, n), dim3(256), lds, s, BUF(L.in_buf), gi, BUF(L.out_buf), g1, g2, g3);
                DT_HIP(hipGetLastError());
                if (ev) {  // (profiling call: the two absorbed layers show as empty)
                    DT_HIP(hipEventRecord((*ev)[li + 1], s));
                    DT_HIP(hipEventRecord((*ev)[li + 2], s));
                }
                li += 2;
                continue;
            }
        }
        if (L.kind == 4 || L.kind == 5) {
            pa::SliceGeom gi = {L.in_h, L.in_w, L.in_pad, L.in_cstride, L.in_coff};
            const int oh = L.kind == 5 ? L.in_h * 2 : L.in_h, ow = L.kind == 5 ? L.in_w * 2 : L.in_w;
            pa::SliceGeom go = {oh, ow, L.out_pad, L.out_cstride, L.out_coff};
            const long long total = (long long)n * oh * ow * (L.cin / 4);
            if (L.kind == 4)
                hipLaunchKernelGGL(pa::maxpool5_kernel, dim3(pa::grid_for(total)), dim3(256), 0, s, BUF(L.in_buf), gi, BUF(L.out_buf), go, n, L.cin);
            else
                hipLaunchKernelGGL(pa::upsample2_kernel, dim3(pa::grid_for(total)), dim3(256), 0, s, BUF(L.in_buf), gi, BUF(L.out_buf), go, n, L.cin);
            DT_HIP(hipGetLastError());
            continue;
        }
        if (L.kind == 6) {
            pa::SliceGeom gi = {L.in_h, L.in_w, L.in_pad, L.in_cstride, L.in_coff};
            hipLaunchKernelGGL(pa::detect_decode_kernel, dim3((L.in_h * L.in_w * no + 255) / 256, n * 3), dim3(256), 0, s, BUF(L.in_buf), gi, 3, no,
                               L.aux[0], h->anchors + (size_t)di * 8, pred, h->rows, row0);
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
        p.act = BUF(L.in_buf) + L.in_coff;
        p.wgt = h->weights + L.w_off;
        p.bias = h->weights + L.b_off;
        p.residual = L.res_buf >= 0 ? BUF(L.res_buf) + L.res_coff : nullptr;
        p.out = BUF(L.out_buf) + L.out_coff;
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
        hipError_t pe;
        static const int use_pgemm = getenv("PA_DET_PGEMM") ? atoi(getenv("PA_DET_PGEMM")) : 1;  // 0: the one-tile-per-workgroup engine (A/B)
        static const int use_patch = getenv("PA_DET_PATCH") ? atoi(getenv("PA_DET_PATCH")) : 1;  // 0: im2col for the 3x3 convolutions (A/B)
        pe = hipErrorInvalidValue;
        // A convolution whose output slice the NEXT layer up-samples by two (model.10 -> model.11, model.14 -> model.15) writes the
        // up-sampled copy itself -- four more stores per output instead of a pass over HBM -- and the up-sampling layer is skipped
        // (PA_DET_UP_FUSE=0: A/B); the persistent GEMMs (exact and emulated) take it
        static const int up_fuse = getenv("PA_DET_UP_FUSE") ? atoi(getenv("PA_DET_UP_FUSE")) : 1;
        bool fused_up = false;
        size_t up_floats = 0;
        auto try_up = [&]() {
            if (!(up_fuse && li + 1 < h->layers.size() && L.act == 2 && !p.residual)) return;
            const pa_net_layer& U = h->layers[li + 1];
            if (U.kind == 5 && U.in_buf == L.out_buf && U.in_coff == L.out_coff && U.in_cstride == L.out_cstride && U.in_pad == L.out_pad &&
                U.cin == L.cout && U.in_h == oh && U.in_w == ow && U.out_buf != L.out_buf) {
                const int up_wb = 2 * ow + 2 * U.out_pad, up_hb = 2 * oh + 2 * U.out_pad;
                p.up_out = BUF(U.out_buf) + U.out_coff;
                p.up_px_stride = U.out_cstride;
                p.up_row_stride = up_wb * U.out_cstride;
                p.up_img_stride = up_hb * up_wb * U.out_cstride;
                p.up_pad = U.out_pad;
                up_floats = (size_t)n * p.up_img_stride - (size_t)U.out_coff;
                fused_up = true;
            }
        };
        auto up_done = [&]() {
            if (fused_up && pe == hipSuccess) {
                ++li;   // the up-sampling layer is done
                if (ev) (void)hipEventRecord((*ev)[li], s);   // (profiling call: the absorbed layer shows as empty)
            }
        };
        if (h->split_off[li] >= 0) {
            // emulated fp32 (psgemm.hip); out_floats: from the layer's first output channel to the end of the images in flight
            try_up();
            pe = pa::launch_psgemm(p, h->split_weights + h->split_off[li], (size_t)n * p.out_img_stride - (size_t)L.out_coff, up_floats, s);
            if (fused_up && pe == hipErrorInvalidValue) {
                p.up_out = nullptr;
                fused_up = false;
                pe = pa::launch_psgemm(p, h->split_weights + h->split_off[li], (size_t)n * p.out_img_stride - (size_t)L.out_coff, 0, s);
            }
            up_done();
        }
        if (pe == hipErrorInvalidValue && h->wino_off[li] >= 0) {
            pa::WinoParams q;
            memset(&q, 0, sizeof(q));
            q.act = p.act; q.wgt = h->wino_weights + h->wino_off[li]; q.bias = p.bias; q.residual = p.residual; q.out = p.out;
            q.n_img = n; q.height = L.in_h; q.width = L.in_w; q.cin = L.cin; q.cout = L.cout; q.bn = h->wino_bn[li];
            q.in_px_stride = p.in_px_stride; q.in_row_stride = p.in_row_stride; q.in_img_stride = p.in_img_stride;
            q.out_px_stride = p.out_px_stride; q.out_row_stride = p.out_row_stride; q.out_img_stride = p.out_img_stride; q.out_pad = p.out_pad;
            q.relu = p.relu; q.res_after = p.res_after;
            pe = pa::launch_wino3x3(q, s);
        }
        if (pe == hipErrorInvalidValue && use_patch && L.ksize == 3 && L.stride == 1 && L.in_pad == 1)
            pe = pa::launch_conv3x3_patch_blocked(p, s);  // input patch resident in LDS across the nine taps (patchconv.hip)
        if (pe != hipErrorInvalidValue) {
        } else if ((use_pgemm || p.N % 64) && !p.residual) {
            // 1x1 and stride-2 convolutions: persistent workgroups over runs of tiles (pigemm.hip)
            try_up();
            pe = pa::launch_pgemm(p, 0, s);
            if (fused_up && pe == hipErrorInvalidValue) {   // (a shape the fused form does not take: the layer alone, then the up-sampling as a layer)
                p.up_out = nullptr;
                fused_up = false;
                pe = pa::launch_pgemm(p, 0, s);
            }
            up_done();
        } else if (p.N % 64 == 0) {
            const pa::GemmTile tile = (p.N % 128 == 0 && t128 / 2 >= 512) ? pa::TILE_128x128 : (t128 >= 512 ? pa::TILE_128x64 : pa::TILE_64x64);
            pe = pa::launch_igemm(p, tile, s);
        }   // (32 output channels: the persistent and the patch kernel only -- the table builder keeps such layers on them)
        if (pe != hipSuccess) return fail(PA_ERR_HIP, "layer " + std::to_string(li) + ": " + hipGetErrorString(pe));
    }
    if (ev) DT_HIP(hipEventRecord(ev->back(), s));
#undef DT_HIP
    return PA_OK;
}

int pa_detector_forward(pa_detector* h, const uint8_t* frames, int32_t n, int32_t height, int32_t width, float* pred, void* stream) {
    // PA_DET_LANES=2 (A/B): the batch as two halves on two streams -- the caller's and one of the handle's own --, each in its own image
    // range of the activation buffers: one half's launch ramps, kernel tails and short layers run under the other half's steady state
    // (what parallel.ClipLanes does for the action CNN). Same kernels on the same images: results bit-identical to one batch
    // wherever a layer's tile shape does not depend on the batch size.
    static const int lanes = getenv("PA_DET_LANES") ? atoi(getenv("PA_DET_LANES")) : 1;
    // PA_DET_BLOCK=b (A/B): the layers on the large maps (down to the stride-2 convolution that produces the 1/16 map) run over b
    // images at a time, block after block on the caller's stream, then the rest of the network over the whole batch: a block's
    // layer outputs (63 MB at b = 16 where the batch's are 252 MB) are still in the 256 MB memory-side cache when the next layer
    // reads them. The short-K layers there run at the copy rate of their activations (profiles/r06_pgemm_split_defer.txt).
    static const int block = getenv("PA_DET_BLOCK") ? atoi(getenv("PA_DET_BLOCK")) : 0;
    if (h && block > 0 && n > block && frames && pred && lanes < 2) {
        int lb = 0;
        while (lb < (int)h->layers.size() && (h->layers[lb].kind == 3 || h->layers[lb].in_h > h->net_h / 16)) ++lb;
        // (a range must not end inside a group one launch absorbs: SPPF pools, a fused up-sampling -- none on the large maps' prefix)
        if (lb > 0 && lb < (int)h->layers.size() && h->layers[lb].kind == 0 && h->layers[lb - 1].kind == 0) {
            for (int b0 = 0; b0 < n; b0 += block) {
                const int nb = n - b0 < block ? n - b0 : block;
                const int rc = detector_run(h, frames + (size_t)b0 * height * width * 3, nb, height, width, pred + (size_t)b0 * h->rows * (5 + h->nc), stream,
                                            nullptr, b0, 0, lb);
                if (rc) return rc;
            }
            return detector_run(h, frames, n, height, width, pred, stream, nullptr, 0, lb, -1);
        }
    }
    if (!h || lanes < 2 || n < 16 || !frames || !pred) return detector_run(h, frames, n, height, width, pred, stream, nullptr);
    hipStream_t s = (hipStream_t)stream;
    if (!h->side) {
        if (hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
            h->last_error = "pa_detector_forward: cannot create the second lane's stream";
            return PA_ERR_HIP;
        }
    }
    const int half = n / 2;
    if (hipEventRecord(h->ev_fork, s) != hipSuccess || hipStreamWaitEvent(h->side, h->ev_fork, 0) != hipSuccess) return PA_ERR_HIP;
    int rc = detector_run(h, frames, half, height, width, pred, s, nullptr, 0);
    if (rc) return rc;
    rc = detector_run(h, frames + (size_t)half * height * width * 3, n - half, height, width, pred + (size_t)half * h->rows * (5 + h->nc), h->side, nullptr, half);
    if (rc) return rc;
    if (hipEventRecord(h->ev_join, h->side) != hipSuccess || hipStreamWaitEvent(s, h->ev_join, 0) != hipSuccess) return PA_ERR_HIP;
    return PA_OK;
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
