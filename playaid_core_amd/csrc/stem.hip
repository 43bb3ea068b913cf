// 7x7/2 stem convolution of ResNet-18 (cnn_action_detector.py:18 in the reference ->
// torchvision resnet18.conv1 + bn1 + relu) as a persistent direct convolution for gfx950.
//
// The generic implicit-GEMM engine (igemm.hip) spends the stem on LDS fill: every k-step
// (one ky) re-stages a 128 x 32-float im2col tile whose rows are 8-pixel windows that
// overlap 4x, plus the weight tile, for only 12 MFMAs per wave. Here instead
//   * a workgroup is persistent and walks over (crop, output-row-pair) tiles;
//   * the folded weights of its 32 output channels live in 84 VGPRs per lane for the whole
//     kernel (k = ky*32 + px*4 + c, the c == 3 pad is never multiplied);
//   * the input patch of a tile (9 padded input rows x 134 px x 4 ch fp32 = 19296 B, ONE
//     contiguous block of the [134][134][4] crop) is copied global -> LDS once with
//     global_load_lds_dwordx4 into a double buffer while the previous tile computes, and the
//     overlapping 8-pixel windows are read straight out of it (5.8x less LDS fill than im2col);
//   * one barrier per tile (168 MFMAs per wave) instead of one per 12;
//   * lanes of a 16-lane ds_read_b128 phase are 2 chunks apart, which would be a 2-way bank
//     conflict; LDS chunk j holds patch chunk j ^ ((j >> 4) & 1) (applied on the DMA source
//     address, because the DMA destination is lane-linear), which makes the reads conflict-free.
// The summation order (ky, pixel pair, channel) is the one igemm.hip uses for the stem, so the
// two kernels produce bit-identical outputs.
#include "pa_kernels.h"
#ifdef PA_STAMP_BUILD
#include <cstdio>
#include <cstdlib>
#include <vector>
#endif

namespace pa {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int IN_W = 134;                  // padded crop width/height (128 + 2*3), 4 floats per pixel
constexpr int OUT_HW = 64;                 // output rows / columns
constexpr int OUT_W = 66;                  // padded output width (border 1 for the 3x3 max-pool)
constexpr int COUT = 64;
constexpr int PATCH_ROWS = 9;              // 2 output rows x stride 2 + 7 taps - 2
constexpr int PATCH_CH = PATCH_ROWS * IN_W;  // 16-byte chunks per patch (1206)
constexpr int STAGE_CH = 1280;             // chunks per LDS stage (5 passes of 256 lanes)
constexpr int KTOT = 224;                  // weight row stride of the igemm layout (7 ky x 32)

// 16-byte global -> LDS DMA, buffer form (see igemm.hip: behind the FLAT form hipcc turns every
// later wait into vmcnt(0) lgkmcnt(0)); source = descriptor base + `off` floats.
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, int off, float* lds_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_base, 16, off * 4, 0, 0, 0);
}

}  // namespace

__global__ __launch_bounds__(256, 2) void stem7x7_kernel(const StemParams p) {
    constexpr int TS = 64;  // row (floats) of the transposed output tile, as in igemm.hip's epilogue
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE_CH * 4 + 128 * TS];
    float* const tbuf = lds + 2 * STAGE_CH * 4;

    const int tid = threadIdx.x;
#ifdef PA_STAMP_BUILD
    const unsigned long long st0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st1 = 0, st_ep = 0;
    int ntile = 0;
#endif
    const int lane = tid & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave_id >> 1;  // output row of the pair
    const int wn = wave_id & 1;   // 32-channel half
    const int lr = lane & 31;
    const int lh = lane >> 5;

    // XCD-aware start: workgroups b and b+8 share an XCD; give each XCD a contiguous run of
    // tiles per sweep so neighbouring row pairs (which share 5 of 9 input rows) meet in one L2.
    const int nwg = gridDim.x;
    const int b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);

    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, -1, 0x00020000);
#define PA_STEM_ISSUE(TILE, BUF)                                                                   \
    {                                                                                              \
        const int img_ = (TILE) >> 5, oy0_ = ((TILE) & 31) * 2;                                    \
        const int src_ = (img_ * IN_W * IN_W + (2 * oy0_) * IN_W) * 4;                             \
        float* dst_ = lds + (BUF) * (STAGE_CH * 4) + wave_id * 256;                                \
        _Pragma("unroll") for (int i = 0; i < 5; ++i) {                                            \
            const int j = tid + 256 * i;                                                           \
            if (j < PATCH_CH) glds16(x_rs, src_ + (j ^ ((j >> 4) & 1)) * 4, dst_ + i * 1024);      \
        }                                                                                          \
    }

    int t = wg;
    if (t < p.tiles) PA_STEM_ISSUE(t, 0);

    // weights of this lane's output channel, resident for the whole kernel
    const int n = wn * 32 + lr;
    float bw[7][4][3];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p.wgt + (size_t)n * KTOT + ky * 32 + (2 * kk + lh) * 4);
            bw[ky][kk][0] = v.x;
            bw[ky][kk][1] = v.y;
            bw[ky][kk][2] = v.z;
        }
    const float bias = p.bias[n];
    const int c_lane = (2 * wm) * IN_W + 2 * lr + lh;  // chunk of (ky 0, mi 0, kk 0) for this lane

    __syncthreads();
#ifdef PA_STAMP_BUILD
    st1 = __builtin_amdgcn_s_memrealtime();
#endif
    int buf = 0;
    for (; t < p.tiles; t += nwg) {
        const int tn = t + nwg;
        if (tn < p.tiles) PA_STEM_ISSUE(tn, buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        const float* patch = lds + buf * (STAGE_CH * 4);

        f32x16 acc[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][e] = 0.f;

        f32x4 af[2][2];
#define PA_STEM_FRAGS(SET, G)                                                                      \
    {                                                                                              \
        _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) {                                         \
            const int c = c_lane + ((G) >> 2) * IN_W + 64 * mi + 2 * ((G)&3);                      \
            af[SET][mi] = *reinterpret_cast<const f32x4*>(patch + (c ^ ((c >> 4) & 1)) * 4);       \
        }                                                                                          \
    }
        PA_STEM_FRAGS(0, 0);
#pragma unroll
        for (int g = 0; g < 28; ++g) {  // g = ky*4 + kk
            const int ky = g >> 2, kk = g & 3;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const f32x4 a4 = af[g & 1][mi];
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bw[ky][kk][0], acc[mi], 0, 0, 0);
                if (mi == 0) {
                    // next group's operands: issued behind the first MFMA, >= 5 MFMAs (320 cycles) ahead of use
                    __builtin_amdgcn_sched_barrier(0);
                    if (g + 1 < 28) PA_STEM_FRAGS((g + 1) & 1, g + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bw[ky][kk][1], acc[mi], 0, 0, 0);
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bw[ky][kk][2], acc[mi], 0, 0, 0);
            }
        }
#undef PA_STEM_FRAGS

#ifdef PA_STAMP_BUILD
        const unsigned long long e0 = __builtin_amdgcn_s_memrealtime();
#endif
        // epilogue: + folded BN bias, ReLU, into the bordered NHWC map [crop][66][66][64]. The tile
        // (2 rows x 64 pixels x 64 channels) is transposed through LDS so that each thread stores 16
        // bytes (8 stores per thread instead of 32 four-byte ones per lane -- the stores of a whole
        // grid reaching its epilogue together otherwise queue up, see igemm.hip).
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ox = mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                float v = acc[mi][e] + bias;
                tbuf[(wm * 64 + ox) * TS + n] = v > 0.f ? v : 0.f;
            }
        __syncthreads();
        {
            const int img = t >> 5, oy0 = (t & 31) * 2;
            const int c4 = (tid & 15) * 4;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = (tid >> 4) + 16 * i;  // wm * 64 + ox
                const f32x4 v = *reinterpret_cast<const f32x4*>(tbuf + row * TS + c4);
                const size_t o = ((size_t)img * OUT_W * OUT_W + (size_t)(oy0 + (row >> 6) + 1) * OUT_W + (row & 63) + 1) * COUT + c4;
                if (p.out_bf16) {  // bf16 conv path: round to nearest even, 4 channels = 8 bytes
                    uint32_t u[4] = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
#pragma unroll
                    for (int k = 0; k < 4; ++k) u[k] = (u[k] + 0x7fffu + ((u[k] >> 16) & 1u)) >> 16;
                    uint2 pk;
                    pk.x = u[0] | (u[1] << 16);
                    pk.y = u[2] | (u[3] << 16);
                    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.out) + o) = pk;
                } else {
                    *reinterpret_cast<f32x4*>(p.out + o) = v;
                }
            }
        }
        __syncthreads();  // next patch landed (vmcnt drained) and every wave is done with this one
#ifdef PA_STAMP_BUILD
        st_ep += __builtin_amdgcn_s_memrealtime() - e0;
        ++ntile;
#endif
        buf ^= 1;
    }
#undef PA_STEM_ISSUE
#ifdef PA_STAMP_BUILD
    if (p.clk && tid == 0) {
        unsigned long long* o = p.clk + (size_t)blockIdx.x * 6;
        o[0] = st0; o[1] = st1; o[2] = __builtin_amdgcn_s_memrealtime(); o[3] = st_ep; o[4] = ntile;
        o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
#endif
}

hipError_t launch_stem7x7(const StemParams& p, hipStream_t s) {
    if (p.tiles <= 0) return hipErrorInvalidValue;
    int grid = p.tiles < 512 ? p.tiles : 512;  // 2 resident workgroups per CU
#ifdef PA_STAMP_BUILD
    static int calls = 0;
    static unsigned long long* sd = nullptr;
    const char* sf = getenv("PA_STEM_STAMP_FILE");
    const bool now = sf && calls++ == 8;
    StemParams q = p;
    if (now) {
        if (!sd) (void)hipMalloc(&sd, 512 * 6 * 8);
        q.clk = sd;
    }
    hipLaunchKernelGGL(stem7x7_kernel, dim3(grid), dim3(256), 0, s, q);
    if (now) {
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> h((size_t)grid * 6);
        (void)hipMemcpy(h.data(), sd, h.size() * 8, hipMemcpyDeviceToHost);
        FILE* f = fopen(sf, "wb");
        if (f) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
    }
    return hipGetLastError();
#else
    hipLaunchKernelGGL(stem7x7_kernel, dim3(grid), dim3(256), 0, s, p);
    return hipGetLastError();
#endif
}

}  // namespace pa
