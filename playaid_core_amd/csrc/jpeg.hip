// The pixel arithmetic of a baseline JPEG write + read on the 128 x 128 crops: what the reference's
// cv2.imwrite(crop) ... cv2.imread(crop) does to every crop before the CNN sees it (playaid/ai_runner.py:420 writes the
// repaired crops, :446 reads all of them back; YOLOv5 --save-crop writes the others). The entropy coding in between is
// lossless, so the round trip is colour conversion -> 2x2 chroma down-sampling -> 8x8 forward DCT -> quantisation |
// de-quantisation -> inverse DCT -> "fancy" chroma up-sampling -> colour conversion, all in libjpeg's integer
// arithmetic (jccolor.c, jcsample.c h2v2_downsample, jfdctint.c, jcdctmgr.c, jidctint.c, jdsample.c
// h2v2_fancy_upsample, jdcolor.c) with OpenCV's defaults: quality 95 unless set otherwise, 4:2:0, JDCT_ISLOW.
// Bit-exact against oracle/jpeg.py, which is pinned byte for byte against the live libjpeg-turbo behind Pillow.
//
// One workgroup = one crop: the Y plane (16 KB) and the two down-sampled chroma planes (4 KB each) live in LDS as
// bytes; 512 threads; phase A one thread per 2x2 pixel quad (colour conversion + down-sampling), phase B one thread per
// 8x8 block with the block in registers (384 blocks per crop, one round), phase C one thread per 2x2 quad again (up-sampling + colour
// conversion + the crop / model-input stores). Off unless pa_set_crop_jpeg_quality was called.
#include "pa_kernels.h"
#include "jpeg_dct.h"

namespace pa {
namespace {

using namespace dct;

// One 8x8 block of plane `pl` (row pitch `pitch` bytes) through FDCT -> quantise -> de-quantise -> IDCT, in place.
__device__ void block_roundtrip(uint8_t* pl, int pitch, const int* __restrict__ q) {
    int d[64];
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        const uint2 v = *reinterpret_cast<const uint2*>(pl + y * pitch);
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            d[y * 8 + x] = (int)((v.x >> (8 * x)) & 0xff) - 128;
            d[y * 8 + 4 + x] = (int)((v.y >> (8 * x)) & 0xff) - 128;
        }
    }
#pragma unroll
    for (int y = 0; y < 8; ++y) fdct8<true>(d + y * 8, 1);
#pragma unroll
    for (int x = 0; x < 8; ++x) fdct8<false>(d + x, 8);
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        // jcdctmgr.c: divisor = quantval << 3, rounded half away from zero; jidctint.c multiplies by quantval again
        const int qv = q[i], dv = qv << 3;
        const int a = (abs(d[i]) + (dv >> 1)) / dv;
        d[i] = (d[i] < 0 ? -a : a) * qv;
    }
#pragma unroll
    for (int x = 0; x < 8; ++x) idct8<true>(d + x, 8);
#pragma unroll
    for (int y = 0; y < 8; ++y) idct8<false>(d + y * 8, 1);
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            int a = d[y * 8 + x] + 128, b = d[y * 8 + 4 + x] + 128;
            a = a < 0 ? 0 : (a > 255 ? 255 : a);
            b = b < 0 ? 0 : (b > 255 ? 255 : b);
            lo |= (uint32_t)a << (8 * x);
            hi |= (uint32_t)b << (8 * x);
        }
        *reinterpret_cast<uint2*>(pl + y * pitch) = make_uint2(lo, hi);
    }
}

// h2v2_fancy_upsample: chroma sample at full-resolution (y, x) from the 64 x 64 plane `c`
__device__ __forceinline__ int fancy(const uint8_t* c, int y, int x) {
    const int cy = y >> 1, cx = x >> 1;
    int ny = (y & 1) ? cy + 1 : cy - 1;   // the nearer neighbouring row: above for even rows, below for odd ones
    ny = ny < 0 ? 0 : (ny > 63 ? 63 : ny);
    const int col = 3 * c[cy * 64 + cx] + c[ny * 64 + cx];
    if (x & 1) {
        if (cx == 63) return (col * 4 + 7) >> 4;
        const int nxt = 3 * c[cy * 64 + cx + 1] + c[ny * 64 + cx + 1];
        return (col * 3 + nxt + 7) >> 4;
    }
    if (cx == 0) return (col * 4 + 8) >> 4;
    const int last = 3 * c[cy * 64 + cx - 1] + c[ny * 64 + cx - 1];
    return (col * 3 + last + 8) >> 4;
}

__global__ __launch_bounds__(512) void jpeg_roundtrip_kernel(const JpegParams p) {
    __shared__ __attribute__((aligned(16))) uint8_t yp[128 * 128];
    __shared__ __attribute__((aligned(16))) uint8_t cbp[64 * 64];
    __shared__ __attribute__((aligned(16))) uint8_t crp[64 * 64];
    __shared__ int qt[2][64];
    const int crop = blockIdx.x, tid = threadIdx.x;
    uint8_t* img = p.crops_u8 + (size_t)crop * 128 * 128 * 3;
    if (tid < 128) qt[tid >> 6][tid & 63] = p.qtab[tid];
    const int ri = p.bgr ? 2 : 0, bi = p.bgr ? 0 : 2;
    // ---- A: RGB -> YCbCr (jccolor.c), chroma 2x2 down-sampling with the bias 1, 2, 1, 2 along a row (jcsample.c)
    for (int qd = tid; qd < 64 * 64; qd += 512) {
        const int cy = qd >> 6, cx = qd & 63;
        int sb = 0, sr = 0;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int y = 2 * cy + dy, x = 2 * cx + dx;
                const uint8_t* s = img + (y * 128 + x) * 3;
                const int r = s[ri], g = s[1], b = s[bi];
                yp[y * 128 + x] = (uint8_t)((19595 * r + 38470 * g + 7471 * b + 32768) >> 16);
                sb += (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16;
                sr += (32768 * r + (128 << 16) + 32767 - 27439 * g - 5329 * b) >> 16;
            }
        const int bias = (cx & 1) ? 2 : 1;
        cbp[qd] = (uint8_t)((sb + bias) >> 2);
        crp[qd] = (uint8_t)((sr + bias) >> 2);
    }
    __syncthreads();
    // ---- B: every 8x8 block through the DCT pair: 256 luminance blocks, then 64 + 64 chrominance blocks
    if (tid < 256) {
        const int by = tid >> 4, bx = tid & 15;
        block_roundtrip(yp + (by * 8) * 128 + bx * 8, 128, qt[0]);
    } else if (tid < 384) {
        const int b = tid & 63, by = b >> 3, bx = b & 7;
        block_roundtrip((tid < 320 ? cbp : crp) + (by * 8) * 64 + bx * 8, 64, qt[1]);
    }
    __syncthreads();
    // ---- C: fancy up-sampling (jdsample.c), YCbCr -> RGB (jdcolor.c), stores
    for (int i = tid; i < 128 * 128; i += 512) {
        const int y = i >> 7, x = i & 127;
        const int yy = yp[i], xb = fancy(cbp, y, x) - 128, xr = fancy(crp, y, x) - 128;
        const int r = clamp255(yy + ((91881 * xr + 32768) >> 16));
        const int g = clamp255(yy + ((-22554 * xb + 32768 - 46802 * xr) >> 16));
        const int b = clamp255(yy + ((116130 * xb + 32768) >> 16));
        uint8_t* o = img + i * 3;
        o[ri] = (uint8_t)r; o[1] = (uint8_t)g; o[bi] = (uint8_t)b;
        if (p.x0) {
            // model input: the crop's channels in memory order / 255 in the zero-bordered NHWC4 buffer (fp32 or bf16)
            const float f0 = (float)(p.bgr ? b : r) / 255.0f, f1 = (float)g / 255.0f, f2 = (float)(p.bgr ? r : b) / 255.0f;
            const size_t oo = ((size_t)crop * 134 + (y + 3)) * 134 + (x + 3);
            if (p.x0_bf16) {
                uint32_t u[3] = {__float_as_uint(f0), __float_as_uint(f1), __float_as_uint(f2)};
#pragma unroll
                for (int k = 0; k < 3; ++k) u[k] = (u[k] + 0x7fffu + ((u[k] >> 16) & 1u)) >> 16;
                reinterpret_cast<uint2*>(p.x0)[oo] = make_uint2(u[0] | (u[1] << 16), u[2]);
            } else {
                reinterpret_cast<float4*>(p.x0)[oo] = make_float4(f0, f1, f2, 0.f);
            }
        }
    }
}

}  // namespace

hipError_t launch_jpeg_roundtrip(const JpegParams& p, int ncrops, hipStream_t s) {
    if (ncrops <= 0) return hipSuccess;
    hipLaunchKernelGGL(jpeg_roundtrip_kernel, dim3(ncrops), dim3(512), 0, s, p);
    return hipGetLastError();
}

}  // namespace pa
