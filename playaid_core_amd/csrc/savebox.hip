// YOLOv5's crop hand-off on the device: what `detect.py --save-crop` leaves in crops/<Fighter>/<video>_<n>.jpg and what
// the reference's runner reads back with cv2.imread (playaid/ai_runner.py:208, 445-446). Per (frame, fighter):
//   save_one_box (utils/plots.py, v7.0; gain 1.02, pad 10): the label row's pixel box -> xyxy2xywh -> wh * 1.02 + 10 ->
//     xywh2xyxy -> .long() -> clip_boxes -> im[y1:y2, x1:x2] (BGR), all in float32 like torch
//   Image.fromarray(crop[..., ::-1]).save(f, quality=95, subsampling=0) ... cv2.imread(f): a 4:4:4 baseline JPEG write +
//     read of an image of ANY size = per 8x8 block (edge blocks filled by repeating the last row / column): RGB -> YCbCr,
//     forward DCT, quantise | de-quantise, inverse DCT, YCbCr -> RGB (libjpeg's integer arithmetic, jpeg_dct.h).
// The images land back to back in one device buffer with a pa_crop_image descriptor each, which is what
// pa_runner_inputs / pa_backbone_crop_images take. Bit-exact against oracle/detect.py::save_one_box (rectangle: parity
// unpinned, YOLOv5 is not vendored) + oracle/jpeg.py::roundtrip_any (pinned byte for byte to live libjpeg-turbo).
#include "pa_kernels.h"
#include "jpeg_dct.h"

namespace pa {
namespace {

// one workgroup: rectangles of all entries, exclusive scan of their byte sizes, descriptors
__global__ __launch_bounds__(256) void savebox_plan_kernel(const SaveBoxParams p) {
    __shared__ unsigned long long scan[256];
    __shared__ unsigned long long carry;
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < p.n_entries; base += 256) {
        const int e = base + tid;
        int x1 = 0, y1 = 0, x2 = 0, y2 = 0;
        if (e < p.n_entries) {
            const int frame = e / p.fighters, slot = e - frame * p.fighters;
            int k = -1;
            if (p.det_index) {
                k = p.det_index[e];
            } else {  // the first detection of the fighter's class in label order: the file name without a counter
                for (int i = 0; i < p.counts[frame] && i < p.max_det && k < 0; ++i)
                    if ((int)p.dets[((size_t)frame * p.max_det + i) * 6] == p.class_ids[slot]) k = i;
            }
            if (k >= 0 && k < p.max_det && k < p.counts[frame]) {
                const float* r = p.dets + ((size_t)frame * p.max_det + k) * 6;
                const float W = (float)p.width, H = (float)p.height;
                // back to the rounded pixel box the label row was written from (centres are multiples of 0.5)
                const float xc = rintf(2.0f * (r[1] * W)) / 2.0f, yc = rintf(2.0f * (r[2] * H)) / 2.0f;
                const float bw = rintf(r[3] * W) * p.gain + p.pad, bh = rintf(r[4] * H) * p.gain + p.pad;
                const float fx1 = xc - bw / 2.0f, fy1 = yc - bh / 2.0f, fx2 = xc + bw / 2.0f, fy2 = yc + bh / 2.0f;
                auto trunc_clip = [](float v, int hi) {
                    const int t = (v > -2.0e9f && v < 2.0e9f) ? (int)v : 0;  // .long(): toward zero
                    return t < 0 ? 0 : (t > hi ? hi : t);
                };
                x1 = trunc_clip(fx1, p.width); x2 = trunc_clip(fx2, p.width);
                y1 = trunc_clip(fy1, p.height); y2 = trunc_clip(fy2, p.height);
                if (x2 <= x1 || y2 <= y1) x1 = x2 = y1 = y2 = 0;
                if (p.src_frame && (unsigned)p.src_frame[e] >= (unsigned)p.n_src) x1 = x2 = y1 = y2 = 0;  // never dereferenced
            }
        }
        const unsigned long long bytes = ((unsigned long long)(y2 - y1) * (x2 - x1) * 3 + 15) & ~15ull;
        scan[tid] = bytes;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const unsigned long long v = tid >= o ? scan[tid - o] : 0;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        const unsigned long long off = carry + scan[tid] - bytes;
        if (e < p.n_entries) {
            const bool fits = off + bytes <= p.capacity;
            SaveBoxRect rc;
            rc.x1 = x1; rc.y1 = y1;
            rc.w = fits ? x2 - x1 : 0; rc.h = fits ? y2 - y1 : 0;
            p.rects[e] = rc;
            CropImageDesc d;
            d.offset = (long long)off;
            d.height = rc.h; d.width = rc.w;
            p.desc[e] = d;
            if (!fits && bytes) atomicAdd(p.overflow, 1);
        }
        __syncthreads();
        if (tid == 255) carry += scan[255];
        __syncthreads();
    }
}

// rows of the rectangle -> the packed image buffer (BGR kept); grid (row groups, entries)
__global__ __launch_bounds__(256) void savebox_copy_kernel(const SaveBoxParams p) {
    const int e = blockIdx.y;
    const SaveBoxRect rc = p.rects[e];
    if (rc.w <= 0 || rc.h <= 0) return;
    const int frame = p.src_frame ? p.src_frame[e] : e / p.fighters;
    const uint8_t* src = p.frames + ((size_t)frame * p.height + rc.y1) * p.width * 3 + (size_t)rc.x1 * 3;
    uint8_t* dst = p.images + p.desc[e].offset;
    const int row_bytes = rc.w * 3;
    for (int y = blockIdx.x; y < rc.h; y += gridDim.x) {
        const uint8_t* s = src + (size_t)y * p.width * 3;
        uint8_t* d = dst + (size_t)y * row_bytes;
        for (int i = threadIdx.x; i < row_bytes; i += 256) d[i] = s[i];
    }
}

// one 8x8 block of one component: FDCT -> quantise -> de-quantise -> IDCT, in place (values = sample - 128 on entry,
// sample 0..255 on exit)
__device__ __forceinline__ void block444(int* d, const int* __restrict__ q) {
    using namespace dct;
#pragma unroll
    for (int y = 0; y < 8; ++y) fdct8<true>(d + y * 8, 1);
#pragma unroll
    for (int x = 0; x < 8; ++x) fdct8<false>(d + x, 8);
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        // (|d| + dv / 2) / dv without an integer division (~25 instructions each, 192 per block): both operands are
        // below 2^24, so the float quotient is off by at most one and one correction step makes it exact
        const int qv = q[i], dv = qv << 3;
        const int x = abs(d[i]) + (dv >> 1);
        int a = (int)((float)x * __builtin_amdgcn_rcpf((float)dv));
        const int r = x - a * dv;
        a += r >= dv ? 1 : (r < 0 ? -1 : 0);
        d[i] = (d[i] < 0 ? -a : a) * qv;
    }
#pragma unroll
    for (int x = 0; x < 8; ++x) idct8<true>(d + x, 8);
#pragma unroll
    for (int y = 0; y < 8; ++y) idct8<false>(d + y * 8, 1);
#pragma unroll
    for (int i = 0; i < 64; ++i) d[i] = clamp255(d[i] + 128);
}

// one thread = one 8x8 block of one image, its three components one after the other; in place
__global__ __launch_bounds__(64) void jpeg444_kernel(const SaveBoxParams p) {
    using namespace dct;
    __shared__ int qt[2][64];
    const int e = blockIdx.y;
    if (threadIdx.x < 64) {
        qt[0][threadIdx.x] = p.qtab[threadIdx.x];
        qt[1][threadIdx.x] = p.qtab[64 + threadIdx.x];
    }
    __syncthreads();
    const CropImageDesc dsc = p.desc[e];
    const int h = dsc.height, w = dsc.width;
    if (h <= 0 || w <= 0) return;
    const int bw = (w + 7) >> 3, bh = (h + 7) >> 3;
    uint8_t* img = p.images + dsc.offset;
    for (int blk = blockIdx.x * 64 + threadIdx.x; blk < bw * bh; blk += gridDim.x * 64) {
        const int by = blk / bw, bx = blk - by * bw;
        // the block's pixels, the last row / column repeated past the edge (jcprepct.c); B, G, R in memory
        uint32_t px[64];
#pragma unroll
        for (int y = 0; y < 8; ++y) {
            const int yy = min(by * 8 + y, h - 1);
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const int xx = min(bx * 8 + x, w - 1);
                const uint8_t* s = img + ((size_t)yy * w + xx) * 3;
                px[y * 8 + x] = s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16);
            }
        }
        uint32_t out[64];
        int d[64];
        // Y
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const int b = px[i] & 0xff, g = (px[i] >> 8) & 0xff, r = (px[i] >> 16) & 0xff;
            d[i] = ((19595 * r + 38470 * g + 7471 * b + 32768) >> 16) - 128;
        }
        block444(d, qt[0]);
#pragma unroll
        for (int i = 0; i < 64; ++i) out[i] = (uint32_t)d[i];
        // Cb
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const int b = px[i] & 0xff, g = (px[i] >> 8) & 0xff, r = (px[i] >> 16) & 0xff;
            d[i] = ((-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16) - 128;
        }
        block444(d, qt[1]);
#pragma unroll
        for (int i = 0; i < 64; ++i) out[i] |= (uint32_t)d[i] << 8;
        // Cr
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const int b = px[i] & 0xff, g = (px[i] >> 8) & 0xff, r = (px[i] >> 16) & 0xff;
            d[i] = ((32768 * r + (128 << 16) + 32767 - 27439 * g - 5329 * b) >> 16) - 128;
        }
        block444(d, qt[1]);
        // jdcolor.c, stores inside the image only
#pragma unroll
        for (int y = 0; y < 8; ++y) {
            const int yy = by * 8 + y;
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const int xx = bx * 8 + x;
                if (yy < h && xx < w) {
                    const int i = y * 8 + x;
                    const int Y = out[i] & 0xff, xb = (int)((out[i] >> 8) & 0xff) - 128, xr = d[i] - 128;
                    uint8_t* o = img + ((size_t)yy * w + xx) * 3;
                    o[2] = (uint8_t)clamp255(Y + ((91881 * xr + 32768) >> 16));
                    o[1] = (uint8_t)clamp255(Y + ((-22554 * xb + 32768 - 46802 * xr) >> 16));
                    o[0] = (uint8_t)clamp255(Y + ((116130 * xb + 32768) >> 16));
                }
            }
        }
    }
}

}  // namespace

hipError_t launch_save_one_box(const SaveBoxParams& p, hipStream_t s) {
    if (p.n_entries <= 0) return hipSuccess;
    hipLaunchKernelGGL(savebox_plan_kernel, dim3(1), dim3(256), 0, s, p);
    hipLaunchKernelGGL(savebox_copy_kernel, dim3(32, p.n_entries), dim3(256), 0, s, p);
    if (p.quality > 0) {
        const int max_blocks = ((p.height + 7) / 8) * ((p.width + 7) / 8);
        const int gx = max_blocks / 64 < 64 ? (max_blocks + 63) / 64 : 64;
        hipLaunchKernelGGL(jpeg444_kernel, dim3(gx, p.n_entries), dim3(64), 0, s, p);
    }
    return hipGetLastError();
}

}  // namespace pa
