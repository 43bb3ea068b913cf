// Implicit-GEMM convolution / linear engine for gfx950 (MI355X), exact fp32.
//
// One kernel family carries every FLOP-heavy layer of the path
// (cnn_action_detector.py:14-43 in the reference): the 7x7/2 stem, the sixteen
// 3x3 convolutions and three 1x1/2 downsamples of ResNet-18, the 512->1000 fc,
// the temporal Conv1d(1000->512, k=S) over cached features (gather mode) and --
// through the same code -- nothing else. BatchNorm is folded into the weights
// on the host, bias + residual + ReLU are fused into the epilogue.
//
// Structure (CDNA4):
//   * workgroup = 256 threads = 4 wave64 arranged 2x2 over a BM x BN tile;
//   * the im2col A-tile (BM output pixels x 32 k-values) and the weight B-tile
//     (BN channels x 32 k-values) are staged global -> registers -> LDS with
//     16-byte accesses; NHWC activations carry an explicit zero border so the
//     gather is pure address arithmetic (no predicates); next tile's global
//     loads are issued before the current tile's MFMAs (register prefetch);
//   * LDS rows are padded to 36 floats so that the ds_read_b128 operand reads
//     (16-lane groups, bank = dword % 64) are conflict-free;
//   * the inner product runs on v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate,
//     bit-identical to an fmaf chain, at the same peak as the f32 VALU but with
//     one operand VGPR per lane (guide: cdna_hip_programming.md section 3,
//     "FP32-input MFMA"). Each lane reads 4 consecutive k of its row with one
//     ds_read_b128 and feeds them to 4 consecutive MFMAs; lanes 0-31 take
//     k = 8j..8j+3, lanes 32-63 take k = 8j+4..8j+7 of each 8-wide k group.
//   * blockIdx is remapped so that the workgroups sharing an XCD (b % 8) own a
//     contiguous run of tiles (same A rows, neighbouring weight panels) in that
//     XCD's private L2;
//   * deterministic split-K (slabs + ordered reduce) for the small-M layers.
#include "pa_kernels.h"

namespace pa {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: stays in VGPRs (HIP's float4 class did not)

constexpr int BK = 32;
constexpr int LDS_STRIDE = 36;  // floats per LDS row (32 + 4 pad)

template <int BM, int BN, bool GATHER>
__global__ __launch_bounds__(256) void igemm_f32_kernel(const GemmParams p) {
    constexpr int MI = BM / 64;  // 32x32 MFMA tiles per wave along M
    constexpr int NI = BN / 64;
    constexpr int A_ROWS = BM / 32;  // staging rows per thread
    constexpr int B_ROWS = BN / 32;
    __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * LDS_STRIDE];
    float* As = lds;
    float* Bs = lds + BM * LDS_STRIDE;

    // XCD-aware (bijective) remap: blocks with equal b % 8 share an XCD.
    const int nwg = gridDim.x;
    const int b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    const int tiles_mn = p.tiles_m * p.tiles_n;
    const int z = wg / tiles_mn;
    const int t_id = wg - z * tiles_mn;
    const int tile_m = t_id / p.tiles_n;
    const int tile_n = t_id - tile_m * p.tiles_n;

    const int tid = threadIdx.x;
    const int colq = tid & 7;   // which float4 of the 32-wide k chunk
    const int row0 = tid >> 3;  // 0..31

    int a_off[A_ROWS];
    int b_off[B_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
        int m = tile_m * BM + row0 + 32 * i;
        m = m < p.M ? m : p.M - 1;
        if (GATHER) {
            a_off[i] = m * p.taps;
        } else {
            const int img = m / p.howo;
            const int rem = m - img * p.howo;
            const int oy = rem / p.wo;
            const int ox = rem - oy * p.wo;
            a_off[i] = img * p.in_img_stride + oy * p.stride * p.in_row_stride +
                       ox * p.stride * p.in_px_stride + colq * 4;
        }
    }
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) b_off[i] = (tile_n * BN + row0 + 32 * i) * p.ktot + colq * 4;

    const int nk = p.ktot / BK;
    const int cpt = p.chunk / BK;  // k-steps per tap
    const int ks_begin = z * p.ksteps_per_split;
    int ks_end = ks_begin + p.ksteps_per_split;
    ks_end = ks_end < nk ? ks_end : nk;

    f32x4 a_reg[A_ROWS];
    f32x4 b_reg[B_ROWS];

    // Global -> register stage of k-step `ks` (kept as a macro, not a lambda:
    // hipcc demotes arrays captured by a lambda to scratch memory, which puts a
    // vmcnt(0) right behind the loads and defeats the prefetch).
#define PA_PREFETCH(KS)                                                                                       \
    {                                                                                                         \
        const int ks__ = (KS);                                                                                \
        const int tap = ks__ / cpt;                                                                           \
        const int kc = (ks__ - tap * cpt) * BK;                                                               \
        if (GATHER) {                                                                                         \
            _Pragma("unroll") for (int i = 0; i < A_ROWS; ++i) {                                              \
                const int row = p.gather[a_off[i] + tap];                                                     \
                const int rowc = row >= 0 ? row : 0;                                                          \
                f32x4 v = *reinterpret_cast<const f32x4*>(p.act + (size_t)rowc * p.in_px_stride + kc + colq * 4); \
                if (row < 0) v = (f32x4)(0.f);                                                                \
                a_reg[i] = v;                                                                                 \
            }                                                                                                 \
        } else {                                                                                              \
            const int ky = tap / p.kw_taps;                                                                   \
            const int kx = tap - ky * p.kw_taps;                                                              \
            const int tapoff = (ky + p.off_y) * p.in_row_stride + (kx + p.off_x) * p.in_px_stride + kc;       \
            _Pragma("unroll") for (int i = 0; i < A_ROWS; ++i)                                                \
                a_reg[i] = *reinterpret_cast<const f32x4*>(p.act + a_off[i] + tapoff);                        \
        }                                                                                                     \
        const int koff = tap * p.chunk + kc;                                                                  \
        _Pragma("unroll") for (int i = 0; i < B_ROWS; ++i)                                                    \
            b_reg[i] = *reinterpret_cast<const f32x4*>(p.wgt + b_off[i] + koff);                              \
    }

    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31;  // MFMA row (A) / column (B) of this lane
    const int lh = lane >> 5;  // which k of the pair

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    const float* a_rd = As + (wm * (BM / 2) + lr) * LDS_STRIDE + 4 * lh;
    const float* b_rd = Bs + (wn * (BN / 2) + lr) * LDS_STRIDE + 4 * lh;

    if (ks_begin < ks_end) PA_PREFETCH(ks_begin);
    for (int ks = ks_begin; ks < ks_end; ++ks) {
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i)
            *reinterpret_cast<f32x4*>(As + (row0 + 32 * i) * LDS_STRIDE + colq * 4) = a_reg[i];
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i)
            *reinterpret_cast<f32x4*>(Bs + (row0 + 32 * i) * LDS_STRIDE + colq * 4) = b_reg[i];
        __syncthreads();
        // prefetch the next k-step (the last iteration re-loads its own tile: harmless, branch-free)
        PA_PREFETCH(ks + 1 < ks_end ? ks + 1 : ks);
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            f32x4 af[MI], bf[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                af[mi] = *reinterpret_cast<const f32x4*>(a_rd + mi * 32 * LDS_STRIDE + kk * 8);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                bf[ni] = *reinterpret_cast<const f32x4*>(b_rd + ni * 32 * LDS_STRIDE + kk * 8);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].x, bf[ni].x, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].y, bf[ni].y, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].z, bf[ni].z, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].w, bf[ni].w, acc[mi][ni], 0, 0, 0);
                }
        }
        __syncthreads();
    }

#undef PA_PREFETCH
    // Epilogue. 32x32 C/D map: col = lane & 31, row = (e & 3) + 8*(e >> 2) + 4*(lane >> 5).
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = wm * (BM / 2) + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            const int m = tile_m * BM + row;
            if (m >= p.M) continue;
            if (p.splitk > 1) {
                float* dst = p.slab + ((size_t)z * p.M + m) * p.N;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) dst[tile_n * BN + wn * (BN / 2) + ni * 32 + lr] = acc[mi][ni][e];
            } else {
                const int img = m / p.howo;
                const int rem = m - img * p.howo;
                const int oy = rem / p.wo;
                const int ox = rem - oy * p.wo;
                const size_t o = (size_t)img * p.out_img_stride + (size_t)(oy + p.out_pad) * p.out_row_stride +
                                 (size_t)(ox + p.out_pad) * p.out_px_stride;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int n = tile_n * BN + wn * (BN / 2) + ni * 32 + lr;
                    float v = acc[mi][ni][e];
                    if (p.bias) v += p.bias[n];
                    if (p.residual) v += p.residual[o + n];
                    if (p.relu) v = v > 0.f ? v : 0.f;
                    p.out[o + n] = v;
                }
            }
        }
    }
}

// Ordered (deterministic) split-K reduction with the fused epilogue.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p) {
    const int n4 = p.N >> 2;
    const size_t total = (size_t)p.M * n4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / n4);
        const int n = (int)(i - (size_t)m * n4) * 4;
        float4 s = *reinterpret_cast<const float4*>(p.slab + (size_t)m * p.N + n);
        for (int z = 1; z < p.splitk; ++z) {
            const float4 v = *reinterpret_cast<const float4*>(p.slab + ((size_t)z * p.M + m) * p.N + n);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const int img = m / p.howo;
        const int rem = m - img * p.howo;
        const int oy = rem / p.wo;
        const int ox = rem - oy * p.wo;
        const size_t o = (size_t)img * p.out_img_stride + (size_t)(oy + p.out_pad) * p.out_row_stride +
                         (size_t)(ox + p.out_pad) * p.out_px_stride + n;
        if (p.bias) {
            const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
            s.x += bv.x; s.y += bv.y; s.z += bv.z; s.w += bv.w;
        }
        if (p.residual) {
            const float4 rv = *reinterpret_cast<const float4*>(p.residual + o);
            s.x += rv.x; s.y += rv.y; s.z += rv.z; s.w += rv.w;
        }
        if (p.relu) {
            s.x = s.x > 0.f ? s.x : 0.f; s.y = s.y > 0.f ? s.y : 0.f;
            s.z = s.z > 0.f ? s.z : 0.f; s.w = s.w > 0.f ? s.w : 0.f;
        }
        *reinterpret_cast<float4*>(p.out + o) = s;
    }
}

template <int BM, int BN>
static hipError_t launch_tile(const GemmParams& p, hipStream_t s) {
    const int grid = p.tiles_m * p.tiles_n * p.splitk;
    if (p.gather)
        hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, true>), dim3(grid), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, false>), dim3(grid), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_igemm(const GemmParams& p_in, GemmTile tile, hipStream_t s) {
    GemmParams p = p_in;
    const int bm = tile == TILE_64x64 ? 64 : 128;
    const int bn = tile == TILE_128x128 ? 128 : 64;
    if (p.N % bn != 0 || p.chunk % BK != 0 || p.ktot != p.taps * p.chunk || p.M <= 0) return hipErrorInvalidValue;
    p.tiles_m = (p.M + bm - 1) / bm;
    p.tiles_n = p.N / bn;
    const int nk = p.ktot / BK;
    if (p.splitk < 1) p.splitk = 1;
    if (p.splitk > nk) p.splitk = nk;
    p.ksteps_per_split = (nk + p.splitk - 1) / p.splitk;
    p.splitk = (nk + p.ksteps_per_split - 1) / p.ksteps_per_split;  // no empty splits
    hipError_t err;
    switch (tile) {
        case TILE_128x128: err = launch_tile<128, 128>(p, s); break;
        case TILE_128x64: err = launch_tile<128, 64>(p, s); break;
        default: err = launch_tile<64, 64>(p, s); break;
    }
    if (err != hipSuccess) return err;
    if (p.splitk > 1) return launch_splitk_reduce(p, s);
    return hipSuccess;
}

hipError_t launch_splitk_reduce(const GemmParams& p, hipStream_t s) {
    const size_t total = (size_t)p.M * (p.N >> 2);
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace pa
