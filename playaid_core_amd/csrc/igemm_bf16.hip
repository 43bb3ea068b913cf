// bf16 implicit-GEMM convolution for gfx950 (BASELINE.json configs[2]: "bf16 conv path").
//
// Same structure as igemm.hip -- zero-bordered NHWC activations, im2col rows and weight rows
// copied global -> LDS with global_load_lds_dwordx4 into a three-stage ring, XOR swizzle on the
// DMA source address, 2x2 waves over a BM x BN tile, fused bias / residual / ReLU epilogue,
// deterministic split-K, optional second source (fused 1x1/2 downsample) -- but activations and
// BatchNorm-folded weights are stored as bf16 and the inner product runs on
// v_mfma_f32_32x32x16_bf16 (f32 accumulate): one ds_read_b128 per operand now feeds ONE matrix
// instruction that retires 16 k, where the fp32 kernel needs four instructions for 8 k.
// An LDS row is 128 B (BK = 64 bf16) or 256 B (BK = 128), byte-for-byte the geometry of the
// fp32 kernel's BK = 32 / 64, so the swizzle and the bank analysis carry over unchanged.
// The matrix pipe is no longer the limit here (MFMA time per k-step drops 8x): this kernel is
// bound by LDS fill and, across the layer, by HBM (bf16 ridge ~315 FLOP/B).
//
// Only conv mode (no gather): the temporal Conv1d head and the fc stay in fp32.
#include "pa_kernels.h"

namespace pa {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint16_t bf16_t;  // storage

namespace {

// 16-byte global -> LDS DMA, buffer form (see igemm.hip); source = descriptor base + `off` bf16 elements.
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, int off, float* lds_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_base, 16, off * 2, 0, 0, 0);
}

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((uint32_t)h << 16); }

__device__ __forceinline__ bf16_t f2bf(float f) {  // round to nearest even (inputs are finite)
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

__device__ __forceinline__ void split_m(const GemmParams& p, int m, int& img, int& oy, int& ox) {
    if (p.howo_shift >= 0) {
        img = m >> p.howo_shift;
        const int rem = m & (p.howo - 1);
        oy = rem >> p.wo_shift;
        ox = rem & (p.wo - 1);
    } else {
        img = m / p.howo;
        const int rem = m - img * p.howo;
        oy = rem / p.wo;
        ox = rem - oy * p.wo;
    }
}

}  // namespace

// BK in bf16 elements: 64 (128-byte LDS rows) or 128 (256-byte rows).
template <int BM, int BN, int BK>
__global__ __launch_bounds__(256) void igemm_bf16_kernel(const GemmParams p) {
    constexpr int MI = BM / 64;
    constexpr int NI = BN / 64;
    constexpr int ROW_F = BK / 2;          // floats (4-byte words) per LDS row: 32 or 64
    constexpr int CH = BK / 8;             // 16-byte chunks per row: 8 or 16
    constexpr int KG = CH / 2;             // 16-wide k groups per row (one MFMA each)
    constexpr int PASS_ROWS = 256 / CH;
    constexpr int A_ROWS = BM / PASS_ROWS;
    constexpr int B_ROWS = BN / PASS_ROWS;
    constexpr int STAGE = (BM + BN) * ROW_F;  // floats per LDS stage
    __shared__ __attribute__((aligned(16))) float lds[3 * STAGE];

    const bf16_t* act = reinterpret_cast<const bf16_t*>(p.act);
    const bf16_t* act2 = reinterpret_cast<const bf16_t*>(p.act2);
    const bf16_t* wgt = reinterpret_cast<const bf16_t*>(p.wgt);
    const bf16_t* residual = reinterpret_cast<const bf16_t*>(p.residual);
    bf16_t* out = reinterpret_cast<bf16_t*>(p.out);

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
    const int row0 = tid / CH;
    const int colq = (tid & (CH - 1)) ^ (CH == 8 ? ((row0 >> 1) & 7) : (row0 & 15));  // source-side swizzle

    int a_off[A_ROWS], a_off2[A_ROWS], b_off[B_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
        int m = tile_m * BM + row0 + PASS_ROWS * i;
        m = m < p.M ? m : p.M - 1;
        int img, oy, ox;
        split_m(p, m, img, oy, ox);
        a_off[i] = img * p.in_img_stride + oy * p.stride * p.in_row_stride + ox * p.stride * p.in_px_stride + colq * 8;
        a_off2[i] = 0;
        if (p.act2)
            a_off2[i] = img * p.in2_img_stride + (oy * p.stride2 + p.off2) * p.in2_row_stride +
                        (ox * p.stride2 + p.off2) * p.in2_px_stride + colq * 8;
    }
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) b_off[i] = (tile_n * BN + row0 + PASS_ROWS * i) * p.ktot + colq * 8;

    const int nk_main = (p.ktot - p.k2_steps * BK) / BK;
    const int nk = p.ktot / BK;
    const int ks_begin = z * p.ksteps_per_split;
    int ks_end = ks_begin + p.ksteps_per_split;
    ks_end = ks_end < nk ? ks_end : nk;

    int cur_kc, cur_kx, cur_ky;
    int issue_ks = ks_begin;
    {
        const int cpt = p.chunk / BK;
        const int ksm = ks_begin < nk_main ? ks_begin : nk_main;
        const int tap = ksm / cpt;
        cur_kc = (ksm - tap * cpt) * BK;
        cur_ky = tap / p.kw_taps;
        cur_kx = tap - cur_ky * p.kw_taps;
    }
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t act_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(act), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t wgt_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(wgt), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t act2_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(act2 ? act2 : act), 0, -1, 0x00020000);

#define PA_ISSUE_STAGE(BUF)                                                                                   \
    {                                                                                                         \
        float* As_w = lds + (BUF) * STAGE + wave_id * 256;                                                    \
        float* Bs_w = As_w + BM * ROW_F;                                                                      \
        if (issue_ks >= nk_main) {                                                                            \
            const int kc2 = (issue_ks - nk_main) * BK;                                                        \
            _Pragma("unroll") for (int i = 0; i < A_ROWS; ++i) glds16(act2_rs, a_off2[i] + kc2, As_w + i * 1024); \
            const int koff2 = nk_main * BK + kc2;                                                             \
            _Pragma("unroll") for (int i = 0; i < B_ROWS; ++i) glds16(wgt_rs, b_off[i] + koff2, Bs_w + i * 1024); \
        } else {                                                                                              \
            const int tap = cur_ky * p.kw_taps + cur_kx;                                                      \
            const int tapoff = (cur_ky + p.off_y) * p.in_row_stride + (cur_kx + p.off_x) * p.in_px_stride + cur_kc; \
            _Pragma("unroll") for (int i = 0; i < A_ROWS; ++i) glds16(act_rs, a_off[i] + tapoff, As_w + i * 1024); \
            const int koff = tap * p.chunk + cur_kc;                                                          \
            _Pragma("unroll") for (int i = 0; i < B_ROWS; ++i) glds16(wgt_rs, b_off[i] + koff, Bs_w + i * 1024); \
            cur_kc += BK;                                                                                     \
            if (cur_kc == p.chunk) {                                                                          \
                cur_kc = 0;                                                                                   \
                if (++cur_kx == p.kw_taps) {                                                                  \
                    cur_kx = 0;                                                                               \
                    ++cur_ky;                                                                                 \
                }                                                                                             \
            }                                                                                                 \
        }                                                                                                     \
        ++issue_ks;                                                                                           \
    }

    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31;
    const int lh = lane >> 5;  // which 8-wide half of a 16-wide k group

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    const int a_rd_off = (wm * (BM / 2) + lr) * ROW_F;
    const int b_rd_off = BM * ROW_F + (wn * (BN / 2) + lr) * ROW_F;
    const int swz = CH == 8 ? ((lr >> 1) & 7) : (lr & 15);

    // Epilogue operands (thread-mapped, see the epilogue) are fetched now so their latency hides
    // under the k loop: thread t owns channels [c8, c8+8) of rows r_t + ROWS_PP * i.
    constexpr int TS = BN;                 // fp32 row of the transposed tile (see igemm.hip)
    constexpr int CPR = BN / 8;            // 16-byte (8 x bf16) chunks per output row
    constexpr int ROWS_PP = 256 / CPR;
    constexpr int EP_IT = BM / ROWS_PP;
    static_assert(BM * TS <= 3 * STAGE, "transposed tile must fit the LDS ring");
    const bool direct_out = p.splitk <= 1;
    const int c8 = (tid & (CPR - 1)) * 8;
    const int r_t = tid / CPR;
    int o_t[EP_IT];
    uint4 res_t[EP_IT];
#pragma unroll
    for (int i = 0; i < EP_IT; ++i) {
        int m = tile_m * BM + r_t + ROWS_PP * i;
        m = m < p.M ? m : p.M - 1;
        int img, oy, ox;
        split_m(p, m, img, oy, ox);
        o_t[i] = img * p.out_img_stride + (oy + p.out_pad) * p.out_row_stride + (ox + p.out_pad) * p.out_px_stride +
                 tile_n * BN + c8;
        res_t[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (direct_out && residual) {
#pragma unroll
        for (int i = 0; i < EP_IT; ++i) res_t[i] = *reinterpret_cast<const uint4*>(residual + o_t[i]);
    }
    float bias8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bias8[k] = (direct_out && p.bias) ? p.bias[tile_n * BN + c8 + k] : 0.f;

    f32x4 af[2][MI], bf[2][NI];
#define PA_LOAD_FRAGS(SET, STAGE_PTR, G)                                                                      \
    {                                                                                                         \
        const int ch = (((G) * 2 + lh) ^ swz) * 4;                                                            \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                     \
            af[SET][mi] = *reinterpret_cast<const f32x4*>((STAGE_PTR) + a_rd_off + mi * 32 * ROW_F + ch);     \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                                     \
            bf[SET][ni] = *reinterpret_cast<const f32x4*>((STAGE_PTR) + b_rd_off + ni * 32 * ROW_F + ch);     \
    }

    // Every wave issues the same number of copies per stage, so the stage boundaries can use COUNTED waits
    // (vector-memory operations retire in order): stage ks+1 has landed once only the copies of stage ks+2 -- issued
    // at the top of this k-step -- are outstanding. A __syncthreads() here would drain vmcnt to 0 and with it the
    // whole ring (every copy would be waited for one k-step after its issue: the round-2 profile of this kernel).
    constexpr int NCOPY = A_ROWS + B_ROWS;
    static_assert(NCOPY == 12 || NCOPY == 8 || NCOPY == 6 || NCOPY == 4, "extend the counted wait below");
    if (ks_begin < ks_end) PA_ISSUE_STAGE(0);
    if (ks_begin + 1 < ks_end) PA_ISSUE_STAGE(1);
    if (ks_begin + 1 < ks_end) {
        if constexpr (NCOPY == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if constexpr (NCOPY == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr (NCOPY == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (ks_begin < ks_end) PA_LOAD_FRAGS(0, lds, 0);
    int buf = 0;
    for (int ks = ks_begin; ks < ks_end; ++ks) {
        const int buf1 = buf == 2 ? 0 : buf + 1;
        const int buf2 = buf1 == 2 ? 0 : buf1 + 1;
        // slot buf2 held stage ks-1: every wave passed the barrier of k-step ks-1 with those reads in registers
        const bool issued = ks + 2 < ks_end;
        if (issued) PA_ISSUE_STAGE(buf2);
        const bool has_next = ks + 1 < ks_end;
        __builtin_amdgcn_sched_barrier(0);
        const float* st_cur = lds + buf * STAGE;
        const float* st_next = lds + buf1 * STAGE;
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            // operands of the next group (or of the next stage's first group) are requested before
            // this group's matrix instructions so the LDS latency hides under them
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < KG) {
                PA_LOAD_FRAGS((g + 1) & 1, st_cur, g + 1);
            } else {
                // close the stage: this wave's reads of it are in registers; the next stage has landed for this
                // wave once only the newest copies are outstanding, and for everyone behind the barrier
                if (issued) {
                    if constexpr (NCOPY == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                    else if constexpr (NCOPY == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else if constexpr (NCOPY == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (has_next) PA_LOAD_FRAGS((g + 1) & 1, st_next, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, af[g & 1][mi]), __builtin_bit_cast(bf16x8, bf[g & 1][ni]), acc[mi][ni], 0, 0, 0);
        }
        buf = buf1;
    }
    __syncthreads();  // (the transposed tile below reuses the ring)
#undef PA_LOAD_FRAGS
#undef PA_ISSUE_STAGE

    // Epilogue: the fp32 accumulators are transposed through the idle LDS ring (rows of BN floats) so that each thread converts and stores 8 channels = 16 bytes per row, EP_IT
    // wide stores (and residual loads) per thread instead of 16*MI*NI two-byte ones per lane.
    float* const tbuf = lds;  // every wave left the k loop through its final barrier
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wm * (BM / 2) + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                tbuf[row * TS + wn * (BN / 2) + ni * 32 + lr] = acc[mi][ni][e];
            }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < EP_IT; ++i) {
        const int row = r_t + ROWS_PP * i;
        const int m = tile_m * BM + row;
        if (m >= p.M) continue;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(tbuf + row * TS + c8);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(tbuf + row * TS + c8 + 4);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        if (!direct_out) {
            float* dst = p.slab + ((size_t)z * p.M + m) * p.N + tile_n * BN + c8;
            *reinterpret_cast<f32x4*>(dst) = v0;
            *reinterpret_cast<f32x4*>(dst + 4) = v1;
        } else {
            const uint32_t rw[4] = {res_t[i].x, res_t[i].y, res_t[i].z, res_t[i].w};
            uint32_t pk[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a = v[2 * k] + bias8[2 * k] + __uint_as_float(rw[k] << 16);
                float b = v[2 * k + 1] + bias8[2 * k + 1] + __uint_as_float(rw[k] & 0xffff0000u);
                if (p.relu) {
                    a = a > 0.f ? a : 0.f;
                    b = b > 0.f ? b : 0.f;
                }
                pk[k] = (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16);
            }
            *reinterpret_cast<uint4*>(out + o_t[i]) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
    }
}

// Ordered split-K reduction (fp32 slabs) with the fused epilogue, bf16 out.
__global__ __launch_bounds__(256) void splitk_reduce_bf16_kernel(const GemmParams p) {
    const bf16_t* residual = reinterpret_cast<const bf16_t*>(p.residual);
    bf16_t* out = reinterpret_cast<bf16_t*>(p.out);
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
        int img, oy, ox;
        split_m(p, m, img, oy, ox);
        const size_t o = (size_t)img * p.out_img_stride + (size_t)(oy + p.out_pad) * p.out_row_stride +
                         (size_t)(ox + p.out_pad) * p.out_px_stride + n;
        if (p.bias) {
            const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
            s.x += bv.x; s.y += bv.y; s.z += bv.z; s.w += bv.w;
        }
        if (residual) {
            const ushort4 rv = *reinterpret_cast<const ushort4*>(residual + o);
            s.x += bf2f(rv.x); s.y += bf2f(rv.y); s.z += bf2f(rv.z); s.w += bf2f(rv.w);
        }
        if (p.relu) {
            s.x = s.x > 0.f ? s.x : 0.f; s.y = s.y > 0.f ? s.y : 0.f;
            s.z = s.z > 0.f ? s.z : 0.f; s.w = s.w > 0.f ? s.w : 0.f;
        }
        ushort4 ov;
        ov.x = f2bf(s.x); ov.y = f2bf(s.y); ov.z = f2bf(s.z); ov.w = f2bf(s.w);
        *reinterpret_cast<ushort4*>(out + o) = ov;
    }
}

template <int BM, int BN, int BK>
static hipError_t launch_tile_bf16(const GemmParams& p, hipStream_t s) {
    const int grid = p.tiles_m * p.tiles_n * p.splitk;
    hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, BK>), dim3(grid), dim3(256), 0, s, p);
    return hipGetLastError();
}

// Tiles as in launch_igemm; the *_K64 shapes mean 256-byte LDS rows (BK = 128 bf16), the others
// 128-byte rows (BK = 64 bf16). All counts in GemmParams (chunk, ktot, strides) are in elements.
hipError_t launch_igemm_bf16(const GemmParams& p_in, GemmTile tile, hipStream_t s) {
    GemmParams p = p_in;
    if (p.gather) return hipErrorInvalidValue;
    auto dims = [](GemmTile t, int* bm, int* bn, int* bk) {
        switch (t) {
            case TILE_256x128: *bm = 256; *bn = 128; *bk = 64; break;
            case TILE_128x128: *bm = 128; *bn = 128; *bk = 64; break;
            case TILE_128x64: *bm = 128; *bn = 64; *bk = 64; break;
            case TILE_64x64: *bm = 64; *bn = 64; *bk = 64; break;
            case TILE_128x64_K64: *bm = 128; *bn = 64; *bk = 128; break;
            default: *bm = 64; *bn = 64; *bk = 128; break;
        }
    };
    int bm, bn, bk;
    dims(tile, &bm, &bn, &bk);
    if (p.chunk % bk != 0 || (p.k2_steps && bk != 64)) {  // 64-channel layers / fused second source: 128-byte rows only
        tile = tile == TILE_128x64_K64 ? TILE_128x64 : (tile == TILE_64x64_K64 ? TILE_64x64 : tile);
        dims(tile, &bm, &bn, &bk);
    }
    if (p.N % bn != 0 || p.chunk % bk != 0 || p.ktot != p.taps * p.chunk + p.k2_steps * 64 || p.M <= 0) return hipErrorInvalidValue;
    if (p.k2_steps && (bk != 64 || !p.act2)) return hipErrorInvalidValue;
    auto ilog2 = [](int v) { int sh = 0; while ((1 << sh) < v) ++sh; return (1 << sh) == v ? sh : -1; };
    p.howo_shift = ilog2(p.howo);
    p.wo_shift = ilog2(p.wo);
    if (p.howo_shift < 0 || p.wo_shift < 0) p.howo_shift = p.wo_shift = -1;
    p.tiles_m = (p.M + bm - 1) / bm;
    p.tiles_n = p.N / bn;
    const int nk = p.ktot / bk;
    if (p.splitk < 1) p.splitk = 1;
    if (p.splitk > nk) p.splitk = nk;
    p.ksteps_per_split = (nk + p.splitk - 1) / p.splitk;
    p.splitk = (nk + p.ksteps_per_split - 1) / p.ksteps_per_split;
    hipError_t err;
    switch (tile) {
        case TILE_256x128: err = launch_tile_bf16<256, 128, 64>(p, s); break;
        case TILE_128x128: err = launch_tile_bf16<128, 128, 64>(p, s); break;
        case TILE_128x64: err = launch_tile_bf16<128, 64, 64>(p, s); break;
        case TILE_64x64: err = launch_tile_bf16<64, 64, 64>(p, s); break;
        case TILE_128x64_K64: err = launch_tile_bf16<128, 64, 128>(p, s); break;
        default: err = launch_tile_bf16<64, 64, 128>(p, s); break;
    }
    if (err != hipSuccess) return err;
    if (p.splitk > 1) {
        const size_t total = (size_t)p.M * (p.N >> 2);
        int grid = (int)((total + 255) / 256);
        if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL(splitk_reduce_bf16_kernel, dim3(grid), dim3(256), 0, s, p);
        return hipGetLastError();
    }
    return hipSuccess;
}

}  // namespace pa
