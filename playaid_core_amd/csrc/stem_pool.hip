// 7x7/2 stem convolution + BatchNorm + ReLU + 3x3/2 max-pool of ResNet-18 in ONE persistent kernel
// (cnn_action_detector.py:16 in the reference -> torchvision resnet18.conv1/bn1/relu/maxpool).
//
// stem.hip writes the 64 x 64 x 64 stem map (134 MB per 128 crops) for a separate pooling kernel to read
// back. Here a workgroup walks a RUN of consecutive output row pairs of one crop; a pooled row needs stem
// rows 2r-1, 2r, 2r+1, i.e. this tile's two rows and the previous tile's second one, whose horizontal
// 3-max every thread simply keeps in registers from one tile to the next. A run that does not start at the
// top of a crop first computes the row pair above it without storing (one warm-up tile per run). The
// stem map never exists in memory: per tile 8 KB of pooled output are stored instead of 32 KB.
//
// The convolution itself is stem.hip's: weights of the lane's output channel resident in VGPRs, the
// tile's input patch (9 padded rows, one contiguous block of the crop) copied global -> LDS by LDS-DMA
// into a double buffer while the previous tile computes, overlapping 8-pixel windows read straight out of
// it, same (ky, pixel, channel) summation order -> the fp32 results are bit-identical to stem.hip +
// maxpool_kernel. BF16 = true (the bf16 conv path, BASELINE.json configs[2]) takes the crop as bf16
// NHWC4 pixels and multiplies on v_mfma_f32_32x32x16_bf16: 28 matrix instructions per tile and wave
// instead of 168 fp32 ones (consecutive lanes then read consecutive 16-byte chunks: no swizzle needed).
#include "pa_kernels.h"

namespace pa {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int IN_W = 134;   // padded crop width / height (128 + 2*3), 4 channels per pixel
constexpr int POOL_W = 34;  // padded pooled width (border 1 for the first 3x3 conv)
constexpr int COUT = 64;
constexpr int KTOT = 224;   // weight row stride (7 ky x 8 px x 4 ch)

template <bool BF16> struct StemGeom {
    static constexpr int ROW_CH = BF16 ? IN_W / 2 : IN_W;        // 16-byte chunks per padded input row
    static constexpr int PATCH_CH = 9 * ROW_CH;                  // 2 output rows x stride 2 + 7 taps - 2 = 9 input rows
    static constexpr int PASSES = (PATCH_CH + 255) / 256;
    static constexpr int STAGE_CH = PASSES * 256;
};

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, int off_bytes, float* lds_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_base, 16, off_bytes, 0, 0, 0);
}

__device__ __forceinline__ f32x4 max4(f32x4 a, f32x4 b) {
    return f32x4{fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)};
}

}  // namespace

template <bool BF16>
__global__ __launch_bounds__(256, 2) void stem_pool_kernel(const StemPoolParams p) {
    using G = StemGeom<BF16>;
    constexpr int TS = 64;  // row (floats) of the transposed stem tile
    __shared__ __attribute__((aligned(16))) float lds[2 * G::STAGE_CH * 4 + 128 * TS];
    float* const tbuf = lds + 2 * G::STAGE_CH * 4;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave_id >> 1;  // output row of the pair
    const int wn = wave_id & 1;   // 32-channel half
    const int lr = lane & 31;
    const int lh = lane >> 5;

    // XCD-aware order of the runs: workgroups b and b+8 share an XCD; neighbouring runs (same crop, shared
    // halo rows) meet in one L2
    const int nwg = gridDim.x;
    const int b = blockIdx.x;
    const int q = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
    const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
    const int runs_per_crop = 32 / p.run;
    const int runs_total = p.crops * runs_per_crop;

    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, -1, 0x00020000);
    // patch of row pair R of crop IMG -> stage BUF (chunk j of the patch lands in LDS chunk j; the fp32 image is
    // XOR-swizzled on the source side, see stem.hip)
#define SP_ISSUE(IMG, R, BUF)                                                                      \
    {                                                                                              \
        const int src_ = ((IMG) * IN_W + 4 * (R)) * G::ROW_CH;   /* first chunk of padded input row 4R */ \
        float* dst_ = lds + (BUF) * (G::STAGE_CH * 4) + wave_id * 256;                             \
        _Pragma("unroll") for (int i = 0; i < G::PASSES; ++i) {                                    \
            const int j = tid + 256 * i;                                                           \
            const int js = BF16 ? j : (j ^ ((j >> 4) & 1));                                        \
            if (j < G::PATCH_CH) dma16(x_rs, (src_ + js) * 16, dst_ + i * 1024);                   \
        }                                                                                          \
    }

    // walk: run -> (crop, first row pair r0); tiles r0-1 (warm-up, if r0 > 0), r0 .. r0+run-1
    int run_id = wg;
    if (run_id >= runs_total) return;
    int crop = run_id / runs_per_crop;
    int r0 = (run_id - crop * runs_per_crop) * p.run;
    int rr = r0 > 0 ? r0 - 1 : 0;
    SP_ISSUE(crop, rr, 0);

    // weights of this lane's output channel, resident for the whole kernel
    const int n = wn * 32 + lr;
    float bw[25][3];     // fp32: step s multiplies taps (2 s, 2 s + 1) of the 49 (ky, kx): this lane (half lh) holds tap 2 s + lh, channels 0-2
    u32x4 bwh[14];       // bf16: 16-wide k groups, this lane's 8 k = group*16 + 8*lh
    if constexpr (BF16) {
        const uint16_t* w = reinterpret_cast<const uint16_t*>(p.wgt) + (size_t)n * KTOT + 8 * lh;
#pragma unroll
        for (int g = 0; g < 14; ++g) bwh[g] = *reinterpret_cast<const u32x4*>(w + g * 16);
    } else {
        // K = 7 x 7 x 3 = 147 exactly (round 4): the k pair of a matrix instruction is two consecutive TAPS of the 49, one per
        // lane half, instead of two of the eight pixels of a kernel row (whose eighth met zero weights: 168 executed).
        // 25 steps x 3 channels = 75 instructions per 32 pixels instead of 84; tap 49 (step 24, upper half) is a zero weight.
        const float* w = reinterpret_cast<const float*>(p.wgt) + (size_t)n * KTOT;
#pragma unroll
        for (int st = 0; st < 25; ++st) {
            const int t0 = 2 * st, t1 = 2 * st + 1;
            const int o0 = (t0 / 7) * 32 + (t0 % 7) * 4, o1 = t1 < 49 ? (t1 / 7) * 32 + (t1 % 7) * 4 : -1;
            f32x4 v = *reinterpret_cast<const f32x4*>(w + (lh && o1 >= 0 ? o1 : o0));
            if (lh && o1 < 0) v = f32x4{0.f, 0.f, 0.f, 0.f};
            bw[st][0] = v.x;
            bw[st][1] = v.y;
            bw[st][2] = v.z;
        }
    }
    const float bias = p.bias[n];
    // chunk of (ky 0, pixel block 0, first k) for this lane
    const int c_lane = BF16 ? (2 * wm) * G::ROW_CH + lr + lh : (2 * wm) * IN_W + 2 * lr;

    // pooling: thread -> pooled pixels px = (tid >> 4) + 16 i (i = 0, 1), channels c4 .. c4+3
    const int c4 = (tid & 15) * 4;
    f32x4 hprev[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};  // horizontal 3-max of stem row 2r-1

    __syncthreads();
    int buf = 0;
    while (true) {
        // the tile after this one (for the prefetch)
        int n_run = run_id, n_crop = crop, n_r0 = r0, n_rr = rr + 1;
        bool more = true;
        if (n_rr == r0 + p.run) {
            n_run = run_id + nwg;
            more = n_run < runs_total;
            if (more) {
                n_crop = n_run / runs_per_crop;
                n_r0 = (n_run - n_crop * runs_per_crop) * p.run;
                n_rr = n_r0 > 0 ? n_r0 - 1 : 0;
            }
        }
        if (more) SP_ISSUE(n_crop, n_rr, buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        const float* patch = lds + buf * (G::STAGE_CH * 4);
        const bool warm = rr < r0;  // the row pair above the run: computed for the pooling's carry, not stored

        f32x16 acc[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][e] = 0.f;

        if constexpr (BF16) {
            u32x4 af[2][2];
#define SP_FRAGS_H(SET, GG)                                                                        \
    {                                                                                              \
        _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) {                                         \
            const int c = c_lane + ((GG) >> 1) * G::ROW_CH + 32 * mi + 2 * ((GG)&1);               \
            af[SET][mi] = *reinterpret_cast<const u32x4*>(patch + c * 4);                          \
        }                                                                                          \
    }
            SP_FRAGS_H(0, 0);
#pragma unroll
            for (int g = 0; g < 14; ++g) {  // g = ky*2 + half: pixels 4*half .. 4*half+3 of tap row ky
                if (g + 1 < 14) SP_FRAGS_H((g + 1) & 1, g + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[g & 1][mi]),
                                                                      __builtin_bit_cast(bf16x8, bwh[g]), acc[mi], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#undef SP_FRAGS_H
        } else {
            f32x4 af[2][2];
            // chunk of this lane's tap of step ST: (ky, kx) = divmod(2 ST + lh, 7); the upper half's tap 49 reads tap 48's pixel
#define SP_FRAGS(SET, ST)                                                                          \
    {                                                                                              \
        const int t0_ = 2 * (ST), t1_ = 2 * (ST) + 1 < 49 ? 2 * (ST) + 1 : 48;                      \
        const int off_ = lh ? (t1_ / 7) * IN_W + (t1_ % 7) : (t0_ / 7) * IN_W + (t0_ % 7);        \
        _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) {                                         \
            const int c = c_lane + off_ + 64 * mi;                                                 \
            af[SET][mi] = *reinterpret_cast<const f32x4*>(patch + (c ^ ((c >> 4) & 1)) * 4);       \
        }                                                                                          \
    }
            if (!(warm && wm == 0)) {  // (a warm-up tile is there for its SECOND stem row only: the first row's waves sit it out)
            SP_FRAGS(0, 0);
#pragma unroll
            for (int g = 0; g < 25; ++g) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const f32x4 a4 = af[g & 1][mi];
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bw[g][0], acc[mi], 0, 0, 0);
                    if (mi == 0) {
                        // next step's operands: issued behind the first MFMA, >= 5 MFMAs (320 cycles) ahead of use
                        __builtin_amdgcn_sched_barrier(0);
                        if (g + 1 < 25) SP_FRAGS((g + 1) & 1, g + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bw[g][1], acc[mi], 0, 0, 0);
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bw[g][2], acc[mi], 0, 0, 0);
                }
            }
            }
#undef SP_FRAGS
        }

        // + folded BN bias, ReLU; the tile (2 rows x 64 pixels x 64 channels) goes through LDS so that the
        // pooling threads see pixel-major rows of 64 channels
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ox = mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const float v = acc[mi][e] + bias;
                tbuf[(wm * 64 + ox) * TS + n] = v > 0.f ? v : 0.f;
            }
        __syncthreads();
        {
            const bool store = rr >= r0;  // not the warm-up tile
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int px = (tid >> 4) + 16 * i;
                f32x4 h[2];
#pragma unroll
                for (int row = 0; row < 2; ++row) {
                    const float* t = tbuf + (row * 64 + 2 * px) * TS + c4;
                    f32x4 m = max4(*reinterpret_cast<const f32x4*>(t), *reinterpret_cast<const f32x4*>(t + TS));
                    // column 2 px - 1: the zero border for px == 0 (every value is >= 0 after the ReLU, so a
                    // zero behaves like torch's -inf padding)
                    if (px > 0) m = max4(m, *reinterpret_cast<const f32x4*>(t - TS));
                    h[row] = m;
                }
                const f32x4 m = max4(max4(hprev[i], h[0]), h[1]);
                hprev[i] = h[1];
                if (store) {
                    const size_t o = (((size_t)crop * POOL_W + rr + 1) * POOL_W + px + 1) * COUT + c4;
                    if (p.out_bf16) {  // round to nearest even, 4 channels = 8 bytes
                        uint32_t u[4] = {__float_as_uint(m.x), __float_as_uint(m.y), __float_as_uint(m.z), __float_as_uint(m.w)};
#pragma unroll
                        for (int k = 0; k < 4; ++k) u[k] = (u[k] + 0x7fffu + ((u[k] >> 16) & 1u)) >> 16;
                        uint2 pk;
                        pk.x = u[0] | (u[1] << 16);
                        pk.y = u[2] | (u[3] << 16);
                        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.out) + o) = pk;
                    } else {
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + o) = m;
                    }
                }
            }
        }
        __syncthreads();  // next patch landed (vmcnt drained) and every wave is done with this one and with tbuf
        if (!more) break;
        if (n_run != run_id) {  // a new run starts (its first tile is its warm-up tile, or the top of a crop)
            hprev[0] = hprev[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        run_id = n_run; crop = n_crop; r0 = n_r0; rr = n_rr;
        buf ^= 1;
    }
#undef SP_ISSUE
}

hipError_t launch_stem_pool(const StemPoolParams& p_in, hipStream_t s) {
    StemPoolParams p = p_in;
    if (p.crops <= 0) return hipErrorInvalidValue;
    // one run per workgroup where possible: 512 workgroups (2 per CU) want crops * (32 / run) >= 512
    int per_crop = 1;
    while (per_crop < 8 && p.crops * per_crop < 512) per_crop *= 2;
    p.run = 32 / per_crop;
    const int runs = p.crops * per_crop;
    const int grid = runs < 512 ? runs : 512;
    if (p.in_bf16)
        hipLaunchKernelGGL(stem_pool_kernel<true>, dim3(grid), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL(stem_pool_kernel<false>, dim3(grid), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace pa
