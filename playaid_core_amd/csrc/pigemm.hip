// Persistent implicit-GEMM convolution for gfx950 (exact fp32): the engine of igemm.hip rebuilt for layers whose tiles are
// SHORT -- the 1x1 convolutions and the stride-2 3x3 convolutions of the detection network (yolo.hip; the YOLOv5 subprocess
// of playaid/ai_runner.py:191-224). A 1x1 convolution over 64 channels is two k-steps per 128 x 64 tile: with one tile per
// workgroup (igemm.hip) a workgroup's life is address arithmetic, one HBM round trip, 64 matrix instructions per wave, a
// transposition through LDS and the drain of its stores -- measured 27-55 TFLOP/s, 1.5-3 TB/s on layers that are bound by
// HBM. Here a workgroup is PERSISTENT over a run of pixel tiles of one channel column and the LDS ring never drains:
//   * the issue cursor (which k-step's operands are copied next) runs two steps ahead of the compute cursor ACROSS tile
//     boundaries, so a tile's first operands arrive under the previous tile's last matrix instructions;
//   * copies are LDS-DMA (buffer_load ... lds), waits are COUNTED (s_waitcnt vmcnt(N) with N = the wave's younger copies
//     and stores) and the barrier is a raw s_barrier: two stages stay in flight across every barrier, and a tile's output
//     stores are never waited for;
//   * the matrix instruction takes the WEIGHTS as its row operand and the pixels as its column operand (as
//     patchconv.hip): a lane owns one pixel and runs of four consecutive channels, so bias + activation + the 16-byte
//     stores run straight from the accumulators -- no transposition, no barrier, nothing between two tiles but the stores.
// Same k order as igemm.hip (tap, channel chunk, eight-wide group, lane half), so results are bit-identical to it.
#include "pa_kernels.h"

namespace pa {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ void pg_dma16(__amdgpu_buffer_rsrc_t rsrc, int off_floats, float* lds_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_base, 16, off_floats * 4, 0, 0, 0);
}

// m / d for 0 <= m < 2^24, 1 <= d < 2^16 (float reciprocal + one correction step either way)
__device__ __forceinline__ int pg_div(int m, int d, float rcp) {
    int q = (int)((float)m * rcp);
    int r = m - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) ++q;
    return q;
}

template <int N> __device__ __forceinline__ void pg_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

}  // namespace

// BM x BN tiles, 256 threads = 2 x 2 waves (BN = 64) or 4 x 1 (BN = 32: layers with 32 output channels run without zero
// padding), three LDS stages of (BM + BN) rows x 32 floats.
template <int BM, int BN = 64>
__global__ __launch_bounds__(256, 2) void pgemm_kernel(const GemmParams p) {
    constexpr int WM = BN == 64 ? 2 : 4;      // waves along the pixel rows
    constexpr int MI = BM / WM / 32;
    constexpr int A_ROWS = BM / 32, B_ROWS = BN / 32;
    constexpr int NLD = A_ROWS + B_ROWS;  // LDS-DMA wave instructions per k-step
    constexpr int NST = MI * 4;           // store wave instructions per tile
    constexpr int STAGE = (BM + BN) * 32;
    __shared__ __attribute__((aligned(16))) float lds[3 * STAGE];

    // --- this workgroup's tiles: one channel column, every lm-th pixel tile of its XCD's contiguous share -------------
    const int b = blockIdx.x, xcd = b & 7, local = b >> 3, per = gridDim.x >> 3;
    const int TN = p.tiles_n, TM = p.tiles_m;
    const int LM = per / TN;  // workgroups per XCD and channel column (the launcher makes per a multiple of TN)
    const int tile_n = local % TN, lm = local / TN;
    const int t_lo = (int)(((long long)xcd * TM) >> 3), t_hi = (int)(((long long)(xcd + 1) * TM) >> 3);
    const int nt = t_lo + lm < t_hi ? (t_hi - t_lo - lm + LM - 1) / LM : 0;
    if (nt == 0) return;
    const int nk = p.ktot >> 5;  // k-steps per tile
    const int total = nt * nk;

    const int tid = threadIdx.x;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = tid >> 3;
    const int colq = (tid & 7) ^ ((row0 >> 1) & 7);  // LDS chunk c of row r holds logical chunk c ^ ((r >> 1) & 7)
    const int lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int wm = BN == 64 ? wave_id >> 1 : wave_id, wn = BN == 64 ? wave_id & 1 : 0;
    const float rcp_howo = 1.0f / (float)p.howo, rcp_wo = 1.0f / (float)p.wo;

    const __amdgpu_buffer_rsrc_t act_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.act), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t wgt_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wgt), 0, -1, 0x00020000);

    // bias of this lane's channels ch0 + 8 g + 0..3: fetched before the first copy is in flight and pinned, so that the
    // compiler's wait for it (a full drain, as for any register load beside LDS-DMA) happens here and not in the loop
    const int ch0 = tile_n * BN + wn * 32 + 4 * lh;
    f32x4 bias4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        bias4[g] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + ch0 + 8 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("" : "+v"(bias4[g]));
    }

    int b_off[B_ROWS];
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) b_off[i] = (tile_n * BN + row0 + 32 * i) * p.ktot + colq * 4;

    // issue cursor: tile, its row offsets, (ky, kx, kc) of its next k-step
    int i_tile = t_lo + lm, i_ks = 0, i_ky = 0, i_kx = 0, i_kc = 0;
    int a_off[A_ROWS];
    auto rows_of = [&](int tile_m) {
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            int m = tile_m * BM + row0 + 32 * i;
            m = m < p.M ? m : p.M - 1;
            const int img = pg_div(m, p.howo, rcp_howo);
            const int rem = m - img * p.howo;
            const int oy = pg_div(rem, p.wo, rcp_wo);
            const int ox = rem - oy * p.wo;
            a_off[i] = img * p.in_img_stride + (oy * p.stride + p.off_y) * p.in_row_stride + (ox * p.stride + p.off_x) * p.in_px_stride + colq * 4;
        }
    };
    rows_of(i_tile);
    auto issue = [&](int slot) {
        float* As_w = lds + slot * STAGE + wave_id * 256;
        float* Bs_w = As_w + BM * 32;
        const int tapoff = i_ky * p.in_row_stride + i_kx * p.in_px_stride + i_kc;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) pg_dma16(act_rs, a_off[i] + tapoff, As_w + i * 1024);
        const int koff = (i_ky * p.kw_taps + i_kx) * p.chunk + i_kc;
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) pg_dma16(wgt_rs, b_off[i] + koff, Bs_w + i * 1024);
        i_kc += 32;
        if (i_kc == p.chunk) {
            i_kc = 0;
            if (++i_kx == p.kw_taps) { i_kx = 0; ++i_ky; }
        }
        if (++i_ks == nk) {  // on to the workgroup's next tile (past the last one: rows of a tile nobody computes; never issued)
            i_ks = 0; i_ky = 0; i_kx = 0; i_kc = 0;
            i_tile += LM;
            rows_of(i_tile < t_hi ? i_tile : t_hi - 1);
        }
    };

    const int a_rd = (wm * (BM / WM) + lr) * 32, b_rd = BM * 32 + (wn * 32 + lr) * 32;
    const int swz = (lr >> 1) & 7;

    issue(0);
    if (total > 1) issue(1);

    f32x16 acc[MI];
    int slot = 0, g = 0;
    for (int t = 0; t < nt; ++t) {
        const int tile_m = t_lo + lm + t * LM;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][e] = 0.f;
        for (int ks = 0; ks < nk; ++ks, ++g) {
            // Stage g must have landed. VMEM operations complete in order and this wave issued, after the copies of stage
            // g (during step g - 2): the stores of the tile step g - 2 closed, the copies of stage g + 1, the stores of the
            // tile step g - 1 closed -- whichever of those exist. Exactly that many may still be outstanding.
            const bool next = g + 1 < total;
            const int closed = (ks == 0 && g > 0 ? 1 : 0) + ((nk == 1 || ks == 1) && g > 1 ? 1 : 0);
            if (next) {
                if (closed == 2) pg_wait_vm<NLD + 2 * NST>();
                else if (closed == 1) pg_wait_vm<NLD + NST>();
                else pg_wait_vm<NLD>();
            } else {
                if (closed == 2) pg_wait_vm<2 * NST>();
                else if (closed == 1) pg_wait_vm<NST>();
                else pg_wait_vm<0>();
            }
            __builtin_amdgcn_s_barrier();   // every wave's share of stage g is in LDS; every wave is done with stage g - 1
            const int slot2 = slot == 0 ? 2 : slot - 1;   // (g + 2) % 3 == (g - 1) % 3
            if (g + 2 < total) issue(slot2);
            __builtin_amdgcn_sched_barrier(0);
            const float* st = lds + slot * STAGE;
            f32x4 af[2][MI], bf[2];
#define PG_FRAGS(SET, KK)                                                                                    \
    {                                                                                                        \
        const int ch_ = (((KK) * 2 + lh) ^ swz) * 4;                                                         \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                    \
            af[SET][mi] = *reinterpret_cast<const f32x4*>(st + a_rd + mi * 1024 + ch_);                      \
        bf[SET] = *reinterpret_cast<const f32x4*>(st + b_rd + ch_);                                          \
    }
            PG_FRAGS(0, 0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const f32x4 a4 = af[kk & 1][mi], b4 = bf[kk & 1];
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.x, a4.x, acc[mi], 0, 0, 0);
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.y, a4.y, acc[mi], 0, 0, 0);
                    if (mi == 0 && kk + 1 < 4) {
                        __builtin_amdgcn_sched_barrier(0);
                        PG_FRAGS((kk + 1) & 1, kk + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.z, a4.z, acc[mi], 0, 0, 0);
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.w, a4.w, acc[mi], 0, 0, 0);
                }
            }
#undef PG_FRAGS
            // this wave's LDS reads of the stage are complete (their results fed the matrix instructions above) before
            // it reaches the next barrier, behind which the stage is overwritten
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            slot = slot == 2 ? 0 : slot + 1;
        }
        // epilogue of the tile, straight from the accumulators: lane = pixel lr of its 32-pixel block, channels ch0 + 8 g + 0..3
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = tile_m * BM + wm * (BM / WM) + mi * 32 + lr;
            const int mc = m < p.M ? m : p.M - 1;
            const int img = pg_div(mc, p.howo, rcp_howo);
            const int rem = mc - img * p.howo;
            const int oy = pg_div(rem, p.wo, rcp_wo);
            const int ox = rem - oy * p.wo;
            const int o_px = img * p.out_img_stride + (oy + p.out_pad) * p.out_row_stride + (ox + p.out_pad) * p.out_px_stride + ch0;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                f32x4 v = f32x4{acc[mi][4 * gq], acc[mi][4 * gq + 1], acc[mi][4 * gq + 2], acc[mi][4 * gq + 3]};
                v += bias4[gq];
                if (p.relu == 1) {
                    v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
                    v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
                } else if (p.relu == 2) {
                    v.x = silu_fast(v.x); v.y = silu_fast(v.y);
                    v.z = silu_fast(v.z); v.w = silu_fast(v.w);
                }
                // (rows past M of a partial last tile were computed on clamped addresses and are dropped)
                // INVARIANT the counted waits rely on: a wave whose rows are all past M issues NO stores here (hipcc branches
                // around them), so its vmcnt(NLD + k * NST) would under-wait if another k-step of this workgroup followed. None
                // does: a workgroup walks its tiles in ASCENDING order (t_lo + lm, + LM, ...) and only the launch's last tile
                // (tiles_m - 1) can be partial, so a partial tile is always the last thing its workgroup computes. Any other
                // tile order must first make the store count independent of the predicate (launch_pgemm checks the premise).
                if (m < p.M) *reinterpret_cast<f32x4*>(p.out + o_px + 8 * gq) = v;
            }
        }
    }
}

// Conv mode of GemmParams (no gather, no second source, no residual, no split-K); bm = 128 | 64 | 0 (chosen here).
hipError_t launch_pgemm(const GemmParams& p_in, int bm, hipStream_t s) {
    GemmParams p = p_in;
    if (p.gather || p.k2_steps || p.residual || p.chunk % 32 != 0 || p.N % 32 != 0 || p.M <= 0 || p.M >= (1 << 24) || p.howo >= (1 << 16) ||
        p.ktot != p.taps * p.chunk || (bm != 128 && bm != 64 && bm != 0))
        return hipErrorInvalidValue;
    const int bn = p.N % 64 == 0 ? 64 : 32;   // 32: output channels that are not a multiple of 64 (128-row tiles only)
    if (bn == 32) bm = 128;
    p.tiles_n = p.N / bn;
    if (p.tiles_n > 64) return hipErrorInvalidValue;
    // Two workgroups per CU = 64 per XCD, a multiple of the channel columns, and no more per column than the XCD's share
    // of pixel tiles. bm = 0: the tile height whose slowest workgroup finishes first.
    auto plan = [&](int bm_, int* lm_out) {   // -> relative time of the slowest workgroup
        const int tm = (p.M + bm_ - 1) / bm_, share = (tm + 7) / 8;
        int lm = 64 / p.tiles_n;
        lm = lm < 1 ? 1 : (lm > share ? share : lm);
        *lm_out = lm;
        // tiles of its slowest workgroup x cost of a tile (a 64-row tile: half the matrix work + the same copies of the
        // weights), and a workgroup alone on its CU (<= 32 per XCD) runs ~1.6x as fast as one of a pair
        return (double)((share + lm - 1) / lm) * (bm_ == 128 ? 1.0 : 0.55) * (lm * p.tiles_n <= 32 ? 0.6 : 1.0);
    };
    int lm = 1;
    if (bm == 0) {
        int lm128, lm64;
        const double e128 = plan(128, &lm128), e64 = plan(64, &lm64);
        bm = e128 <= e64 ? 128 : 64;
        lm = bm == 128 ? lm128 : lm64;
    } else {
        (void)plan(bm, &lm);
    }
    p.tiles_m = (p.M + bm - 1) / bm;
    // the premise of the kernel's counted waits (see its store loop): every tile but the last is full
    if ((long long)(p.tiles_m - 1) * bm >= p.M || (long long)p.tiles_m * bm < p.M) return hipErrorInvalidValue;
    const int per = lm * p.tiles_n;
    const int grid = per * 8;
    if (bn == 32) hipLaunchKernelGGL((pgemm_kernel<128, 32>), dim3(grid), dim3(256), 0, s, p);
    else if (bm == 128) hipLaunchKernelGGL((pgemm_kernel<128>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((pgemm_kernel<64>), dim3(grid), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace pa
