// bf16 implicit-GEMM convolution for gfx950 (BASELINE.json configs[2]: "bf16 conv path").
//
// Same structure as igemm.hip -- zero-bordered NHWC activations, im2col rows and weight rows
// copied global -> LDS with global_load_lds_dwordx4 into a three-stage ring, XOR swizzle on the
// DMA source address, 2x2 waves over a BM x BN tile, fused bias / residual / ReLU epilogue (here straight from the
// accumulators: the weights are the matrix instruction's row operand, a lane owns one pixel and stores 16 bytes),
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
//
// DS (the stride-2 3x3 convolution that opens layers 2-4): the block's 1x1/2 downsample branch reads exactly the pixels
// of this convolution's CENTRE tap, so its product rides on the k-steps of that tap -- the same im2col rows in LDS times
// a second weight tile (p.wgt2, copied only in those k-steps, which therefore count B_ROWS more copies in the waits)
// into a second accumulator set, stored to p.out2 by a second pass of the epilogue. The matrix pipe idles in this kernel
// (it is bound by its copies), so the branch costs its weight copies and its stores instead of a launch of its own.
template <int BM, int BN, int BK, bool DS = false>
__global__ __launch_bounds__(256) void igemm_bf16_kernel(const GemmParams p) {
    constexpr int MI = BM / 64;
    constexpr int NI = BN / 64;
    constexpr int ROW_F = BK / 2;          // floats (4-byte words) per LDS row: 32 or 64
    constexpr int CH = BK / 8;             // 16-byte chunks per row: 8 or 16
    constexpr int KG = CH / 2;             // 16-wide k groups per row (one MFMA each)
    constexpr int PASS_ROWS = 256 / CH;
    constexpr int A_ROWS = BM / PASS_ROWS;
    constexpr int B_ROWS = BN / PASS_ROWS;
    constexpr int STAGE = (BM + BN + (DS ? BN : 0)) * ROW_F;  // floats per LDS stage
    static_assert(!DS || BK == 64, "the centre tap is found by 64-deep k-steps");
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
    int b2_off[DS ? B_ROWS : 1];
    if constexpr (DS) {
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) b2_off[i] = (tile_n * BN + row0 + PASS_ROWS * i) * p.chunk + colq * 8;
    }

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
    const __amdgpu_buffer_rsrc_t wgt2_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DS ? p.wgt2 : p.wgt), 0, -1, 0x00020000);
    bool newest_ctr = false;  // DS: the stage issued last carried the second weight tile

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
            if constexpr (DS) {                                                                               \
                newest_ctr = cur_ky == 1 && cur_kx == 1;                                                      \
                if (newest_ctr) {                                                                             \
                    float* B2s_w = Bs_w + BN * ROW_F;                                                         \
                    _Pragma("unroll") for (int i = 0; i < B_ROWS; ++i) glds16(wgt2_rs, b2_off[i] + cur_kc, B2s_w + i * 1024); \
                }                                                                                             \
            }                                                                                                 \
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

    f32x16 acc2[DS ? MI : 1][DS ? NI : 1];
    if constexpr (DS) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc2[mi][ni][e] = 0.f;
    }

    const int a_rd_off = (wm * (BM / 2) + lr) * ROW_F;
    const int b_rd_off = BM * ROW_F + (wn * (BN / 2) + lr) * ROW_F;
    const int swz = CH == 8 ? ((lr >> 1) & 7) : (lr & 15);

    const bool direct_out = p.splitk <= 1;

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
    static_assert(!DS || NCOPY + B_ROWS == 12 || NCOPY + B_ROWS == 8, "extend the counted wait below");
#define PA_WAIT_COUNT(N_)                                                        \
    {                                                                            \
        if constexpr ((N_) == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); \
        else if constexpr ((N_) == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); \
        else if constexpr ((N_) == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); \
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                    \
    }
    // all but the newest stage's copies have landed
#define PA_WAIT_NEWEST()                                                         \
    {                                                                            \
        if constexpr (DS) {                                                      \
            if (newest_ctr) PA_WAIT_COUNT(NCOPY + B_ROWS) else PA_WAIT_COUNT(NCOPY) \
        } else PA_WAIT_COUNT(NCOPY)                                              \
    }
    if (ks_begin < ks_end) PA_ISSUE_STAGE(0);
    if (ks_begin + 1 < ks_end) PA_ISSUE_STAGE(1);
    if (ks_begin + 1 < ks_end) {
        PA_WAIT_NEWEST();
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
        if constexpr (DS) {
            // centre tap (k-steps 4 cpt .. 5 cpt - 1): the same rows times the second weight tile
            const int cpt = p.chunk / BK;
            if (ks >= 4 * cpt && ks < 5 * cpt) {
#pragma unroll
                for (int g = 0; g < KG; ++g) {
                    const int ch = ((g * 2 + lh) ^ swz) * 4;
                    f32x4 a2[MI], w2[NI];
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) a2[mi] = *reinterpret_cast<const f32x4*>(st_cur + a_rd_off + mi * 32 * ROW_F + ch);
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        w2[ni] = *reinterpret_cast<const f32x4*>(st_cur + b_rd_off + BN * ROW_F + ni * 32 * ROW_F + ch);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            acc2[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bf16x8, w2[ni]), __builtin_bit_cast(bf16x8, a2[mi]), acc2[mi][ni], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
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
                    PA_WAIT_NEWEST();
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
                        __builtin_bit_cast(bf16x8, bf[g & 1][ni]), __builtin_bit_cast(bf16x8, af[g & 1][mi]), acc[mi][ni], 0, 0, 0);
        }
        buf = buf1;
    }
#undef PA_LOAD_FRAGS
#undef PA_ISSUE_STAGE
#undef PA_WAIT_NEWEST
#undef PA_WAIT_COUNT

    // Epilogue, straight from the accumulators. The WEIGHTS are the matrix instruction's row operand, so a lane owns
    // ONE pixel (tile row wm BM/2 + 32 mi + lr) and, per 32-channel block, the runs 8 j + 4 lh + 0..3 (element e = 4 j + i).
    // Swapping halves between lanes l and l + 32 (v_permlane32_swap) turns two runs into eight consecutive channels
    // 16 j2 + 8 lh + 0..7 = one 16-byte store per lane: no transposition through LDS, no barrier.
    int obase[MI], mrow[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        mrow[mi] = tile_m * BM + wm * (BM / 2) + mi * 32 + lr;
        const int m = mrow[mi] < p.M ? mrow[mi] : p.M - 1;
        int img, oy, ox;
        split_m(p, m, img, oy, ox);
        obase[mi] = img * p.out_img_stride + (oy + p.out_pad) * p.out_row_stride + (ox + p.out_pad) * p.out_px_stride +
                    tile_n * BN + wn * (BN / 2) + 8 * lh;
    }
    if (!direct_out) {
        // split-K: fp32 partial sums, a lane's runs of four channels as 16-byte stores
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            if (mrow[mi] >= p.M) continue;
            float* dst = p.slab + ((size_t)z * p.M + mrow[mi]) * p.N + tile_n * BN + wn * (BN / 2) + 4 * lh;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<f32x4*>(dst + ni * 32 + 8 * j) =
                        f32x4{acc[mi][ni][4 * j], acc[mi][ni][4 * j + 1], acc[mi][ni][4 * j + 2], acc[mi][ni][4 * j + 3]};
        }
        return;
    }
    // FULL: bias + residual + activation (the convolution's own output); else plain rounding (the DS branch's)
#define PA_STORE_TILE(ACC, DST, FULL)                                                                          \
    {                                                                                                          \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                                      \
            _Pragma("unroll") for (int j2 = 0; j2 < 2; ++j2) {                                                 \
                float b8_[8];                                                                                  \
                _Pragma("unroll") for (int k = 0; k < 8; ++k)                                                  \
                    b8_[k] = ((FULL) && p.bias) ? p.bias[tile_n * BN + wn * (BN / 2) + ni * 32 + 16 * j2 + 8 * lh + k] : 0.f; \
                _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) {                                            \
                    float vv_[8];                                                                              \
                    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
                        const uint32_t x_ = __float_as_uint((ACC)[mi][ni][8 * j2 + i]);                        \
                        const uint32_t y_ = __float_as_uint((ACC)[mi][ni][8 * j2 + 4 + i]);                    \
                        const auto r_ = __builtin_amdgcn_permlane32_swap(x_, y_, false, false);                \
                        vv_[i] = __uint_as_float(r_[0]);                                                       \
                        vv_[4 + i] = __uint_as_float(r_[1]);                                                   \
                    }                                                                                          \
                    const int o_ = obase[mi] + ni * 32 + 16 * j2;                                              \
                    uint4 rv_ = make_uint4(0u, 0u, 0u, 0u);                                                    \
                    if ((FULL) && residual && mrow[mi] < p.M) rv_ = *reinterpret_cast<const uint4*>(residual + o_); \
                    const uint32_t rw_[4] = {rv_.x, rv_.y, rv_.z, rv_.w};                                      \
                    uint32_t pk_[4];                                                                           \
                    _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                            \
                        float a_ = vv_[2 * k] + b8_[2 * k] + __uint_as_float(rw_[k] << 16);                    \
                        float c_ = vv_[2 * k + 1] + b8_[2 * k + 1] + __uint_as_float(rw_[k] & 0xffff0000u);    \
                        if ((FULL) && p.relu) {                                                                \
                            a_ = a_ > 0.f ? a_ : 0.f;                                                          \
                            c_ = c_ > 0.f ? c_ : 0.f;                                                          \
                        }                                                                                      \
                        pk_[k] = (uint32_t)f2bf(a_) | ((uint32_t)f2bf(c_) << 16);                              \
                    }                                                                                          \
                    if (mrow[mi] < p.M) *reinterpret_cast<uint4*>((DST) + o_) = make_uint4(pk_[0], pk_[1], pk_[2], pk_[3]); \
                }                                                                                              \
            }                                                                                                  \
    }
    PA_STORE_TILE(acc, out, true);
    if constexpr (DS) {
        // the branch's tile: no bias (it sits in the bias of the convolution that adds this tensor as its residual)
        bf16_t* out2 = reinterpret_cast<bf16_t*>(p.out2);
        PA_STORE_TILE(acc2, out2, false);
    }
#undef PA_STORE_TILE
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

template <int BM, int BN, int BK, bool DS = false>
static hipError_t launch_tile_bf16(const GemmParams& p, hipStream_t s) {
    const int grid = p.tiles_m * p.tiles_n * p.splitk;
    hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, BK, DS>), dim3(grid), dim3(256), 0, s, p);
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
    if (p.out2) {  // second 1x1 product on the centre tap (see the kernel): 3x3 taps, 64-deep k-steps, one K split
        if (!p.wgt2 || p.taps != 9 || p.kw_taps != 3 || p.k2_steps || bk != 64 || p.splitk != 1 ||
            (tile != TILE_128x128 && tile != TILE_128x64))
            return hipErrorInvalidValue;
        p.ksteps_per_split = nk;
        return tile == TILE_128x128 ? launch_tile_bf16<128, 128, 64, true>(p, s) : launch_tile_bf16<128, 64, 64, true>(p, s);
    }
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
