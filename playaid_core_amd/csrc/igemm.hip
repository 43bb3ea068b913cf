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
//     (BN channels x 32 k-values) go global -> LDS directly with
//     global_load_lds_dwordx4 (no VGPR round trip, no ds_write: measured 15-18 %
//     of the loop in the register-staged version); NHWC activations carry an
//     explicit zero border so the gather is pure address arithmetic (no
//     predicates); two LDS stages, the loads of k-step k+1 are issued before
//     the MFMAs of step k, one barrier per step;
//   * an LDS-DMA wave instruction writes 1 KiB lane-linear (8 rows x 128 B), so
//     rows cannot be padded; bank conflicts are removed by an XOR swizzle on
//     the SOURCE address instead: LDS chunk c of row r holds logical 16-byte
//     chunk c ^ ((r >> 1) & 7), and the ds_read_b128 operand reads apply the
//     same XOR (16-lane groups, bank = dword % 64: conflict-free);
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
#include <cstdio>
#include <cstdlib>

namespace pa {

#ifndef PA_COUNTED_VMCNT
#define PA_COUNTED_VMCNT 0
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: stays in VGPRs (HIP's float4 class did not)


// 16-byte global -> LDS DMA, buffer form (buffer_load_dwordx4 ... offen lds). LDS destination =
// wave-uniform `lds_base` + lane*16, source = descriptor base + `off` floats. The FLAT form
// (global_load_lds) makes hipcc assume "a FLAT access may be pending" and turn every later wait
// into s_waitcnt vmcnt(0) lgkmcnt(0); behind the MUBUF form the waits stay counted.
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, int off, float* lds_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_base, 16, off * 4, 0, 0, 0);
}

// m -> (img, oy, ox) with shifts when the output plane is a power of two
// (every layer of this network), integer division otherwise.
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

// ABL: timing-only ablation bits for scripts/ablate_igemm.sh (results are wrong when != 0):
//   1 no global->LDS loads, 4 no barriers in the loop, 8 no MFMAs.
// SKIPW: every 4th k carries a zero weight (the stem's channel pad, k = ky*32 + kx*4 + c with c < 3),
// so the MFMA of each fragment's .w component is dropped -- 25 % fewer matrix instructions.
template <int BM, int BN, int BK, bool GATHER, int ABL = 0, bool SKIPW = false>
__global__ __launch_bounds__(256) void igemm_f32_kernel(const GemmParams p) {
    constexpr int MI = BM / 64;  // 32x32 MFMA tiles per wave along M
    constexpr int NI = BN / 64;
    constexpr int LDS_STRIDE = BK;       // floats per LDS row (unpadded: LDS-DMA writes are lane-linear)
    constexpr int CH = BK / 4;           // 16-byte chunks per row (8 or 16)
    constexpr int PASS_ROWS = 256 / CH;  // rows staged by one pass of the 256 threads (32 or 16)
    constexpr int WAVE_ROWS = 64 / CH;   // rows written by one LDS-DMA wave instruction (8 or 4)
    constexpr int A_ROWS = BM / PASS_ROWS;  // staging rows per thread
    constexpr int B_ROWS = BN / PASS_ROWS;
    constexpr int STAGE = (BM + BN) * LDS_STRIDE;  // floats per LDS stage
    __shared__ __attribute__((aligned(16))) float lds[3 * STAGE];

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
    const int row0 = tid / CH;  // staging row of this thread within a pass
    // LDS chunk c of staging row r receives logical chunk c ^ swz(r), swz(r) = (r>>1)&7 for
    // 128-byte rows (BK 32) and r&15 for 256-byte rows (BK 64); r = row0 + PASS_ROWS*i, so the
    // term only depends on row0.
    const int colq = (tid & (CH - 1)) ^ (BK == 32 ? ((row0 >> 1) & 7) : (row0 & 15));

    int a_off[A_ROWS];
    int b_off[B_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
        int m = tile_m * BM + row0 + PASS_ROWS * i;
        m = m < p.M ? m : p.M - 1;
        if (GATHER) {
            a_off[i] = m * p.taps;
        } else {
            int img, oy, ox;
            split_m(p, m, img, oy, ox);
            a_off[i] = img * p.in_img_stride + oy * p.stride * p.in_row_stride +
                       ox * p.stride * p.in_px_stride + colq * 4;
        }
    }
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) b_off[i] = (tile_n * BN + row0 + PASS_ROWS * i) * p.ktot + colq * 4;
    // optional second source for the last k2_steps k-steps (the 1x1/2 downsample branch of a
    // residual block, fused into conv2's accumulation as extra K)
    int a_off2[A_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
        a_off2[i] = 0;
        if (!GATHER && p.act2) {
            int m = tile_m * BM + row0 + PASS_ROWS * i;
            m = m < p.M ? m : p.M - 1;
            int img, oy, ox;
            split_m(p, m, img, oy, ox);
            a_off2[i] = img * p.in2_img_stride + (oy * p.stride2 + p.off2) * p.in2_row_stride +
                        (ox * p.stride2 + p.off2) * p.in2_px_stride + colq * 4;
        }
    }
    const int nk_main = (p.ktot - p.k2_steps * BK) / BK;

    const int nk = p.ktot / BK;
    const int ks_begin = z * p.ksteps_per_split;
    int ks_end = ks_begin + p.ksteps_per_split;
    ks_end = ks_end < nk ? ks_end : nk;

    // k-step cursor, advanced incrementally (one division at entry only)
    int cur_kc, cur_kx, cur_ky;
    int issue_ks = ks_begin;  // absolute index of the next k-step to issue
    {
        const int cpt = p.chunk / BK;
        const int ksm = ks_begin < nk_main ? ks_begin : nk_main;
        const int tap = ksm / cpt;
        cur_kc = (ksm - tap * cpt) * BK;
        cur_ky = tap / p.kw_taps;
        cur_kx = tap - cur_ky * p.kw_taps;
    }

    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t act_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.act), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t wgt_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wgt), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t act2_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.act2 ? p.act2 : p.act), 0, -1, 0x00020000);

    // Issue the LDS-DMA loads of the k-step under the cursor into stage BUF, then
    // advance the cursor. Each wave instruction fills WAVE_ROWS rows (1 KiB).
#define PA_ISSUE_STAGE(BUF)                                                                                   \
    {                                                                                                         \
        float* As_w = lds + (BUF) * STAGE + wave_id * 256;                                                    \
        float* Bs_w = As_w + BM * LDS_STRIDE;                                                                 \
        if (issue_ks >= nk_main) {                                                                            \
            const int kc2 = (issue_ks - nk_main) * BK;                                                        \
            _Pragma("unroll") for (int i = 0; i < A_ROWS; ++i)                                                \
                glds16(act2_rs, a_off2[i] + kc2, As_w + i * 1024);                                            \
            const int koff2 = nk_main * BK + kc2;                                                             \
            _Pragma("unroll") for (int i = 0; i < B_ROWS; ++i)                                                \
                glds16(wgt_rs, b_off[i] + koff2, Bs_w + i * 1024);                                            \
        } else {                                                                                              \
            const int tap = cur_ky * p.kw_taps + cur_kx;                                                      \
            if (GATHER) {                                                                                     \
                _Pragma("unroll") for (int i = 0; i < A_ROWS; ++i) {                                          \
                    const int row = p.gather[a_off[i] + tap];                                                 \
                    glds16(act_rs, row * p.in_px_stride + cur_kc + colq * 4, As_w + i * 1024);        \
                }                                                                                             \
            } else {                                                                                          \
                const int tapoff = (cur_ky + p.off_y) * p.in_row_stride + (cur_kx + p.off_x) * p.in_px_stride + cur_kc; \
                _Pragma("unroll") for (int i = 0; i < A_ROWS; ++i)                                            \
                    glds16(act_rs, a_off[i] + tapoff, As_w + i * 1024);                                       \
            }                                                                                                 \
            const int koff = tap * p.chunk + cur_kc;                                                          \
            _Pragma("unroll") for (int i = 0; i < B_ROWS; ++i)                                                \
                glds16(wgt_rs, b_off[i] + koff, Bs_w + i * 1024);                                             \
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
    const int lr = lane & 31;  // MFMA row (A) / column (B) of this lane
    const int lh = lane >> 5;  // which k of the pair

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    const int a_rd_off = (wm * (BM / 2) + lr) * LDS_STRIDE;
    const int b_rd_off = BM * LDS_STRIDE + (wn * (BN / 2) + lr) * LDS_STRIDE;
    const int swz = BK == 32 ? ((lr >> 1) & 7) : (lr & 15);  // row-dependent chunk XOR (same for every 32-row MFMA tile)

    // The epilogue moves the tile through LDS so that every thread stores 16 bytes (see below):
    // thread t owns channels [c4, c4+4) of rows r_t + ROWS_PP * i. Its bias / residual operands
    // are fetched now, so their latency hides under the whole k loop.
    constexpr int TS = BN;                 // row of the transposed tile (floats): unpadded is the conflict-free choice
                                           // for ds_read_b128's lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31}
    constexpr int CPR = BN / 4;            // 16-byte chunks per row
    constexpr int ROWS_PP = 256 / CPR;     // rows covered by one pass of the 256 threads
    constexpr int EP_IT = BM / ROWS_PP;
    static_assert(BM * TS <= 3 * STAGE, "transposed tile must fit the LDS ring");
    const bool direct_out = p.splitk <= 1;
    const int c4 = (tid & (CPR - 1)) * 4;
    const int r_t = tid / CPR;
    int o_t[EP_IT];
    f32x4 res_t[EP_IT];
#pragma unroll
    for (int i = 0; i < EP_IT; ++i) {
        int m = tile_m * BM + r_t + ROWS_PP * i;
        m = m < p.M ? m : p.M - 1;
        int img, oy, ox;
        split_m(p, m, img, oy, ox);
        o_t[i] = img * p.out_img_stride + (oy + p.out_pad) * p.out_row_stride + (ox + p.out_pad) * p.out_px_stride +
                 tile_n * BN + c4;
        res_t[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (direct_out && p.residual) {
#pragma unroll
        for (int i = 0; i < EP_IT; ++i) res_t[i] = *reinterpret_cast<const f32x4*>(p.residual + o_t[i]);
    }
    f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (direct_out && p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + tile_n * BN + c4);

    // Three-stage LDS ring, one barrier per k-step, no post-barrier bubble:
    //   step k: issue LDS-DMA of step k+2 into stage (k+2)%3
    //           MFMAs of step k from stage k%3; operand fragments are read one
    //           8-wide k group ahead, and the LAST group of the step already
    //           reads the first fragments of step k+1 from stage (k+1)%3
    //           __syncthreads()  (hipcc drains the DMA with vmcnt(0) first)
    // RAW: stage (k+1)%3 was drained + barriered at the end of step k-1, so it is
    //      visible to every wave during step k.
    // WAR: the DMA of step k+3 (issued in step k+1) overwrites stage k%3, whose
    //      last reads every wave completed before the barrier that ends step k.
    f32x4 af[2][MI], bf[2][NI];
#define PA_LOAD_FRAGS(SET, STAGE_PTR, KK)                                                                     \
    {                                                                                                         \
        const int ch = (((KK) * 2 + lh) ^ swz) * 4;                                                           \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                     \
            af[SET][mi] = *reinterpret_cast<const f32x4*>((STAGE_PTR) + a_rd_off + mi * 32 * LDS_STRIDE + ch); \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                                     \
            bf[SET][ni] = *reinterpret_cast<const f32x4*>((STAGE_PTR) + b_rd_off + ni * 32 * LDS_STRIDE + ch); \
    }
#ifdef PA_ABLATION_BUILD
    const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (ks_begin < ks_end) PA_ISSUE_STAGE(0);
    if (ks_begin + 1 < ks_end) PA_ISSUE_STAGE(1);
    __syncthreads();
    if (ks_begin < ks_end) PA_LOAD_FRAGS(0, lds, 0);
    int buf = 0;
    for (int ks = ks_begin; ks < ks_end; ++ks) {
        if (PA_COUNTED_VMCNT && ks > ks_begin) PA_LOAD_FRAGS(0, lds + buf * STAGE, 0);
        const int buf1 = buf == 2 ? 0 : buf + 1;
        const int buf2 = buf1 == 2 ? 0 : buf1 + 1;
        if (ks + 2 < ks_end && !(ABL & 1)) PA_ISSUE_STAGE(buf2);
        const bool has_next = ks + 1 < ks_end;
        // pin the issue order: hipcc otherwise sinks the loads below the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        const float* st_cur = lds + buf * STAGE;
        const float* st_next = lds + buf1 * STAGE;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            // The next group's reads are issued after the first two MFMAs of this
            // group (pinned with sched_barrier): >= 128 cycles of MFMA pipe time
            // remain to cover the LDS latency, and the lgkmcnt(0) hipcc puts in
            // front of the next group never waits on a read that was just issued.
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const f32x4 a4 = af[kk & 1][mi], b4 = bf[kk & 1][ni];
                    if (!(ABL & 8)) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[mi][ni], 0, 0, 0);
                    } else {
                        acc[mi][ni][0] += a4.x * b4.x + a4.y * b4.y + a4.z * b4.z + a4.w * b4.w;
                    }
                    if (mi == 0 && ni == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (kk + 1 < BK / 8) {
                            PA_LOAD_FRAGS((kk + 1) & 1, st_cur, kk + 1);
                        } else if (has_next && !PA_COUNTED_VMCNT) {
                            PA_LOAD_FRAGS(0, st_next, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (!(ABL & 8)) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[mi][ni], 0, 0, 0);
                        if (!SKIPW) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[mi][ni], 0, 0, 0);
                    }
                }
        }
        if (!(ABL & 4)) {
            if (PA_COUNTED_VMCNT) {
                // leave the stage issued in THIS step in flight across the barrier: only the older
                // stage (needed next step) must have landed. __syncthreads() would drain to vmcnt(0).
                if (ks + 2 < ks_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_ROWS + B_ROWS) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            } else {
                __syncthreads();
            }
        }
        buf = buf1;
    }
#undef PA_LOAD_FRAGS
#undef PA_ISSUE_STAGE
#ifdef PA_ABLATION_BUILD
    if (p.clk && tid == 0 && blockIdx.x == gridDim.x / 2) {
        p.clk[0] = __builtin_amdgcn_s_memtime() - clk_t0;
        p.clk[1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    }
#endif

    // Epilogue. The accumulators hold one output channel per lane (32x32 C/D map: col = lane & 31,
    // row = (e & 3) + 8*(e >> 2) + 4*(lane >> 5)): stored directly that is 16*MI*NI four-byte stores
    // per lane, and since every workgroup of a launch reaches its epilogue at about the same time
    // the stores queue up (timeline stamps: 8 us median, 16 us worst, of a 46 us workgroup). The
    // tile is therefore transposed through the (now idle) LDS ring -- rows of BN floats -- and each
    // thread moves 16 bytes per row: EP_IT dwordx4 stores (and residual loads) instead.
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
        f32x4 v = *reinterpret_cast<const f32x4*>(tbuf + row * TS + c4);
        if (!direct_out) {
            *reinterpret_cast<f32x4*>(p.slab + ((size_t)z * p.M + m) * p.N + tile_n * BN + c4) = v;
        } else {
            v += p.res_after ? bias4 : bias4 + res_t[i];
            if (p.relu == 1) {
                v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
                v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
            } else if (p.relu == 2) {
                v.x = silu_fast(v.x); v.y = silu_fast(v.y);
                v.z = silu_fast(v.z); v.w = silu_fast(v.w);
            }
            if (p.res_after) v += res_t[i];
            *reinterpret_cast<f32x4*>(p.out + o_t[i]) = v;
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
        int img, oy, ox;
        split_m(p, m, img, oy, ox);
        const size_t o = (size_t)img * p.out_img_stride + (size_t)(oy + p.out_pad) * p.out_row_stride +
                         (size_t)(ox + p.out_pad) * p.out_px_stride + n;
        if (p.bias) {
            const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
            s.x += bv.x; s.y += bv.y; s.z += bv.z; s.w += bv.w;
        }
        float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.residual) rv = *reinterpret_cast<const float4*>(p.residual + o);
        if (!p.res_after) { s.x += rv.x; s.y += rv.y; s.z += rv.z; s.w += rv.w; }
        if (p.relu == 1) {
            s.x = s.x > 0.f ? s.x : 0.f; s.y = s.y > 0.f ? s.y : 0.f;
            s.z = s.z > 0.f ? s.z : 0.f; s.w = s.w > 0.f ? s.w : 0.f;
        } else if (p.relu == 2) {
            s.x = silu_fast(s.x); s.y = silu_fast(s.y);
            s.z = silu_fast(s.z); s.w = silu_fast(s.w);
        }
        if (p.res_after) { s.x += rv.x; s.y += rv.y; s.z += rv.z; s.w += rv.w; }
        *reinterpret_cast<float4*>(p.out + o) = s;
    }
}

template <int BM, int BN, int BK>
static hipError_t launch_tile(const GemmParams& p, hipStream_t s) {
    const int grid = p.tiles_m * p.tiles_n * p.splitk;
#ifdef PA_ABLATION_BUILD
    static const int abl = getenv("PA_ABLATE") ? atoi(getenv("PA_ABLATE")) : 0;
    static unsigned long long* clk_dev = nullptr;
    static int clk_calls = 0;
    if (getenv("PA_CLK") && !p.gather) {
        if (!clk_dev) (void)hipMalloc(&clk_dev, 16);
        GemmParams q = p;
        q.clk = clk_dev;
        hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, BK, false, 0>), dim3(grid), dim3(256), 0, s, q);
        if (++clk_calls % 97 == 0) {
            unsigned long long h[2];
            (void)hipMemcpy(h, clk_dev, 16, hipMemcpyDeviceToHost);
            fprintf(stderr, "[igemm clk] tile %dx%dx%d M=%d N=%d K=%d: %llu cycles / %llu x10ns -> %.3f GHz\n", BM, BN, BK, p.M, p.N, p.ktot, h[0], h[1], (double)h[0] / (double)h[1] * 0.1);
        }
        return hipGetLastError();
    }
    if (!p.gather && abl) {
#define PA_ABL_CASE(V) case V: hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, BK, false, V>), dim3(grid), dim3(256), 0, s, p); return hipGetLastError();
        switch (abl) { PA_ABL_CASE(1) PA_ABL_CASE(4) PA_ABL_CASE(5) PA_ABL_CASE(8) PA_ABL_CASE(9) PA_ABL_CASE(12) default: break; }
    }
#endif
    if (p.gather)
        hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, BK, true>), dim3(grid), dim3(256), 0, s, p);
    else if (p.skip_w && BK == 32 && BN == 64)
        hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, (BN == 64 ? 32 : BK), false, 0, (BN == 64 && BK == 32)>), dim3(grid), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, BK, false>), dim3(grid), dim3(256), 0, s, p);
    return hipGetLastError();
}

static void tile_dims(GemmTile tile, int* bm, int* bn, int* bk) {
    switch (tile) {
        case TILE_128x128: *bm = 128; *bn = 128; *bk = 32; break;
        case TILE_128x64: *bm = 128; *bn = 64; *bk = 32; break;
        case TILE_64x64: *bm = 64; *bn = 64; *bk = 32; break;
        case TILE_128x64_K64: *bm = 128; *bn = 64; *bk = 64; break;
        default: *bm = 64; *bn = 64; *bk = 64; break;
    }
}

hipError_t launch_igemm(const GemmParams& p_in, GemmTile tile, hipStream_t s) {
    GemmParams p = p_in;
    int bm, bn, bk;
    tile_dims(tile, &bm, &bn, &bk);
    if (p.chunk % bk != 0 || p.k2_steps) {  // 32-wide taps (stem) or a fused second source: BK=32 shapes only
        tile = tile == TILE_128x64_K64 ? TILE_128x64 : TILE_64x64;
        tile_dims(tile, &bm, &bn, &bk);
    }
    if (p.N % bn != 0 || p.chunk % bk != 0 || p.ktot != p.taps * p.chunk + p.k2_steps * 32 || p.M <= 0) return hipErrorInvalidValue;
    if (p.k2_steps && (bk != 32 || !p.act2 || p.gather)) return hipErrorInvalidValue;
    auto ilog2 = [](int v) { int s = 0; while ((1 << s) < v) ++s; return (1 << s) == v ? s : -1; };
    p.howo_shift = ilog2(p.howo);
    p.wo_shift = ilog2(p.wo);
    if (p.howo_shift < 0 || p.wo_shift < 0) p.howo_shift = p.wo_shift = -1;
    p.tiles_m = (p.M + bm - 1) / bm;
    p.tiles_n = p.N / bn;
    const int nk = p.ktot / bk;
    if (p.splitk < 1) p.splitk = 1;
    if (p.splitk > nk) p.splitk = nk;
    p.ksteps_per_split = (nk + p.splitk - 1) / p.splitk;
    p.splitk = (nk + p.ksteps_per_split - 1) / p.ksteps_per_split;  // no empty splits
    hipError_t err;
    switch (tile) {
        case TILE_128x128: err = launch_tile<128, 128, 32>(p, s); break;
        case TILE_128x64: err = launch_tile<128, 64, 32>(p, s); break;
        case TILE_64x64: err = launch_tile<64, 64, 32>(p, s); break;
        case TILE_128x64_K64: err = launch_tile<128, 64, 64>(p, s); break;
        default: err = launch_tile<64, 64, 64>(p, s); break;
    }
    if (err != hipSuccess) return err;
    if (p.splitk_used) *p.splitk_used = p.splitk;
    if (p.splitk > 1 && !p.defer_reduce) return launch_splitk_reduce(p, s);
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
