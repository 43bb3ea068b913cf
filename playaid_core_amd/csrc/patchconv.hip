// Stride-1 3x3 convolution with the activation patch resident in LDS and the weights in a
// register ring (gfx950, exact fp32).
//
// The generic engine (igemm.hip) stages an im2col tile AND a weight tile per k-step, so every
// activation byte of a tile travels L2 -> LDS nine times (once per tap) and the four waves meet
// at a barrier every 32 k. Here the K loop runs channel chunk outermost: for each 32-channel
// chunk the tile's input patch -- (rows + 2) x (W + 2) pixels x 128 B, a CONTIGUOUS pixel range
// of the zero-bordered NHWC buffer (for the small maps: whole padded images) -- is copied to LDS
// once with global_load_lds_dwordx4 (double buffered, one 4 KB pass per tap), and the nine taps
// read their operand fragments from it at a per-tap pixel offset ky*(W+2)+kx. The weights skip
// LDS: every lane fetches the 4 x 16 B of its own output channel straight into a three-deep
// register ring two steps ahead. With no per-step stage there is no per-step barrier: ONE
// barrier per 9 k-steps (288 MFMAs per wave at 128x64) when the patch buffers swap.
//
// Everything else follows igemm.hip: 2x2 waves over a BM x 64 tile, v_mfma_f32_32x32x2_f32,
// 16-byte chunks XOR-swizzled by (pixel >> 1) & 7 on the DMA source so that ds_read_b128 of
// consecutive pixels is conflict-free, fused bias / residual / ReLU epilogue, ordered split-K
// over channel chunks. k is summed in (chunk, tap, channel) order -- a different rounding order
// than igemm.hip's (tap, chunk, channel), equally deterministic. Layers with the fused 1x1/2
// second source (block 0 conv2 of layers 2-4) and the stride-2 convs stay on igemm.hip.
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

// 16-byte global -> LDS DMA in its buffer form (buffer_load_dwordx4 ... offen lds): LDS
// destination = wave-uniform `lds_base` + lane*16, source = descriptor base + voff + soff bytes.
// The FLAT form (global_load_lds) makes hipcc treat every later wait as "a FLAT access may be
// pending" and emit s_waitcnt vmcnt(0) lgkmcnt(0); with the MUBUF form the waits for the weight
// ring stay counted (vmcnt(10) instead of a full drain twice per chunk).
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rsrc, int voff_bytes, int soff_bytes, float* lds_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_base, 16, voff_bytes,
                                             soff_bytes, 0, 0);
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

// m / d for 0 <= m < 2^24, 1 <= d < 2^16
__device__ __forceinline__ int pc_div(int m, int d) {
    int q = (int)((float)m * (1.0f / (float)d));
    int r = m - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) ++q;
    return q;
}

// blocked pixel order (GemmParams::blk_*): m -> (img, oy, ox) and the pixel's place (r, c) inside its block
__device__ __forceinline__ void split_m_blk(const GemmParams& p, int m, int& img, int& oy, int& ox) {
    const int blk = m >> p.blk_shift_px, rem = m & ((1 << p.blk_shift_px) - 1);
    const int r = rem >> p.blk_shift_c, c = rem & (p.blk_cols - 1);
    img = pc_div(blk, p.blk_per_img);
    const int sr = blk - img * p.blk_per_img;
    const int by = pc_div(sr, p.blk_per_row);
    oy = by * p.blk_rows + r;
    ox = (sr - by * p.blk_per_row) * p.blk_cols + c;
}

}  // namespace

// Patch buffer capacity in pixels: 128-row tiles of the 32- and 16-wide maps need 204 / 180,
// 64-row tiles of the 8- and 4-wide maps 100 / 144 (rounded up to whole 32-pixel DMA passes).
// A static array, not dynamic LDS: with `extern __shared__` hipcc cannot tell the DMA's target
// buffer from the one the ds_reads use and drains vmcnt to 0 in front of every operand read.
template <int BM> struct PatchCap { static constexpr int slots = BM == 128 ? 224 : 160; };

// KS2: split-K by two INSIDE the workgroup. 512 threads; waves 0-3 ("half" 0) take the first half
// of the channel chunks (and of the second-source steps), waves 4-7 the second half, each with its
// own pair of patch buffers; the two partial tiles meet in LDS in the epilogue. Same occupancy as
// two 256-thread split-K workgroups per CU, but no slab round trip through HBM and no reduce
// kernel behind the launch (the layer-4 convs: M = 2048 gives only 256 tiles of 64 x 64).
template <int BM, bool K2, bool KS2, bool BLK = false, int BN = 64>
__global__ __launch_bounds__(KS2 ? 512 : 256, 2) void conv3x3_patch_kernel(const GemmParams p) {
    static_assert(BN == 64 || (BN == 32 && BLK && !K2 && !KS2), "32 output channels per tile: blocked form only");
    constexpr int WM = BN == 64 ? 2 : 4;   // waves along the pixel rows (BN = 32: 4 x 1 waves, no zero-padded channels)
    constexpr int MI = BM / WM / 32;
    constexpr int PP = PatchCap<BM>::slots * 32;  // floats per patch buffer
    __shared__ __attribute__((aligned(16))) float pc_lds[(KS2 ? 2 : 1) * 2 * PP];
    const int half = KS2 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;
    float* const patch0 = pc_lds + half * 2 * PP;

    // XCD-aware (bijective) remap: blocks with equal b % 8 share an XCD.
    const int nwg = gridDim.x;
    const int b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    const int tiles_mn = p.tiles_m * p.tiles_n;
    const int z = KS2 ? half : wg / tiles_mn;   // which share of K this group of four waves sums
    const int t_id = KS2 ? wg : wg - (wg / tiles_mn) * tiles_mn;
    int tile_m = t_id / p.tiles_n;
    int tile_n = t_id - tile_m * p.tiles_n;
    if (p.xcd_m) {
        // Each XCD has its own L2, so every XCD fetches the activations AND the weights of its tiles from HBM once.
        // A contiguous run of tiles gives an XCD 1/8 of the pixels and ALL channel columns = the whole filter bank
        // (layer 4: 9.4 MB of weights x 8 XCDs = 75 of the 85 MB fetched per launch, rocprofv3 FETCH_SIZE). Arranging
        // the XCDs as an xcd_m x xcd_n grid over (pixel tiles, channel tiles) fetches acts / xcd_m + weights / xcd_n each.
        const int local = b >> 3;                      // this workgroup's rank inside its XCD
        const int tn_per = p.tiles_n / p.xcd_n, tm_per = p.tiles_m / p.xcd_m;
        const int lm = local / tn_per;
        tile_m = (xcd / p.xcd_n) * tm_per + lm;
        tile_n = (xcd % p.xcd_n) * tn_per + (local - lm * tn_per);
    }

    const int tid = threadIdx.x & 255;
#ifdef PA_STAMP_BUILD
    const unsigned long long st0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = tid >> 3;                              // pixel of this thread within a DMA pass
    const int colq = (tid & 7) ^ ((row0 >> 1) & 7);         // source chunk: slot c holds chunk c ^ ((pixel >> 1) & 7)

    // first pixel of this tile's patch inside the zero-bordered input buffer
    const int p0 = (tile_m / p.tiles_per_img) * p.p0_img + (tile_m % p.tiles_per_img) * p.p0_row;
    // Chunk swizzle. LDS slot c of patch pixel (row, col) [padded image coordinates] holds logical
    // chunk c ^ key, key = (R >> 1) & 7 with R = (A*row + col) & 15, A = W & 15: the 32 lanes of an
    // MFMA row group are 32 consecutive output pixels, i.e. R advances by one per lane whatever the
    // map width (8-wide maps wrap to the next row after 8 lanes, A = 8 keeps R counting), so every
    // 16-lane ds_read_b128 phase sees 16 distinct R = 16 distinct bank slots for all nine taps
    // (brute-forced for W = 32, 16, 8, 4; keying on the linear patch pixel instead cost 5-10 % of
    // the cycles in bank conflicts on the 16/8/4-wide maps: SQ_LDS_BANK_CONFLICT).
    const int band_row0 = p.p0_row ? (tile_m % p.tiles_per_img) * (p.p0_row / p.patch_pitch) : 0;
    // BLK: the patch is not one contiguous pixel range but the rectangles of the tile's blocks, so every DMA pass's source
    // offset (the same for every channel chunk) is worked out once: patch pixel -> (block of the tile, row, column) -> pixel of
    // the zero-bordered input
    int voffb[7];
    if (BLK) {
#pragma unroll
        for (int qq = 0; qq < 7; ++qq) {
            const int pp_ = row0 + 32 * qq;
            const int im_ = (pp_ * p.magic_img) >> 16;
            const int rem_ = pp_ - im_ * p.img_px_patch;
            const int rw_ = (rem_ * p.magic_pitch) >> 16;
            const int col_ = rem_ - rw_ * p.patch_pitch;
            const int key_ = ((p.swz_a * rw_ + col_) >> 1) & 7;
            int blk_ = tile_m * (BM >> p.blk_shift_px) + im_;
            blk_ = blk_ < p.n_blocks ? blk_ : p.n_blocks - 1;   // (slots past the tile's blocks, blocks past a partial last tile)
            const int img_ = pc_div(blk_, p.blk_per_img);
            const int sr_ = blk_ - img_ * p.blk_per_img;
            const int by_ = pc_div(sr_, p.blk_per_row);
            int px_ = img_ * p.img_px + (by_ * p.blk_rows + rw_) * p.src_pitch + (sr_ - by_ * p.blk_per_row) * p.blk_cols + col_;
            px_ = px_ < p.total_px ? px_ : p.total_px - 1;
            voffb[qq] = (px_ * p.in_px_stride + ((tid & 7) ^ key_) * 4) * 4;
        }
    }
    const int n_ch = p.chunk >> 5;                          // 32-channel chunks
    const int ch_begin = z * p.ksteps_per_split;            // (per split: whole chunks)
    int ch_end = ch_begin + p.ksteps_per_split;
    ch_end = ch_end < n_ch ? ch_end : n_ch;
    // fused 1x1/2 second source (block 0 of layers 2-4): k2_steps extra 32-channel steps after the
    // 3x3 chunks; their "patch" is the gathered BM x 32 tile
    // (shared evenly between the splits)
    const int nsplit = KS2 ? 2 : p.splitk;
    const int j0 = K2 ? (z * p.k2_steps) / nsplit : 0;
    const int k2n = K2 ? ((z + 1) * p.k2_steps) / nsplit - j0 : 0;
    constexpr int ROWS2 = BM / 32;
    int a_off2[ROWS2];
#pragma unroll
    for (int i = 0; i < ROWS2; ++i) {
        a_off2[i] = 0;
        if (K2 && k2n) {
            int m = tile_m * BM + row0 + 32 * i;
            m = m < p.M ? m : p.M - 1;
            int img, oy, ox;
            split_m(p, m, img, oy, ox);
            a_off2[i] = (img * p.in2_img_stride + (oy * p.stride2 + p.off2) * p.in2_row_stride +
                         (ox * p.stride2 + p.off2) * p.in2_px_stride + colq * 4) * 4;
        }
    }
    const __amdgpu_buffer_rsrc_t act2_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.act2 ? p.act2 : p.act), 0, -1, 0x00020000);
    // pass Q of the gathered tile of second-source step J into buffer PB
#define PC_TILE2_PASS(J, PB, Q) blds16(act2_rsrc, a_off2[Q], (j0 + (J)) * 128, patch0 + (PB) * PP + (Q) * 1024 + wave_id * 256)

    // One DMA pass (32 pixels x 128 B, one piece per wave) of chunk CH's patch into buffer PB.
    const __amdgpu_buffer_rsrc_t act_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.act), 0, -1, 0x00020000);
#define PC_PATCH_PASS(CH, PB, Q)                                                                   \
    {                                                                                              \
    if (BLK) {                                                                                     \
        blds16(act_rsrc, voffb[Q], (CH) * 128, patch0 + (PB) * PP + (Q) * 1024 + wave_id * 256);   \
    } else {                                                                                       \
        int px_ = p0 + row0 + 32 * (Q);                                                            \
        px_ = px_ < p.total_px ? px_ : p.total_px - 1;                                             \
        const int pp_ = row0 + 32 * (Q);                       /* pixel inside the patch */           \
        const int im_ = (pp_ * p.magic_img) >> 16;              /* padded image it belongs to */       \
        const int rem_ = pp_ - im_ * p.img_px_patch;                                                   \
        const int rw_ = (rem_ * p.magic_pitch) >> 16;           /* its padded row and column */        \
        const int key_ = ((p.swz_a * (rw_ + band_row0) + rem_ - rw_ * p.patch_pitch) >> 1) & 7;        \
        blds16(act_rsrc, (px_ * p.in_px_stride + ((tid & 7) ^ key_) * 4) * 4, (CH) * 128,              \
               patch0 + (PB) * PP + (Q) * 1024 + wave_id * 256);                                   \
    }                                                                                              \
    }

    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = BN == 64 ? wave >> 1 : wave, wn = BN == 64 ? wave & 1 : 0;
    const int lr = lane & 31;
    const int lh = lane >> 5;

    // Weights never touch LDS: lane (lr, lh) of a wave needs, for its output channel n and every
    // 8-wide k group kk of a step, the 4 floats W[n][k0 + 8 kk + 4 lh ..] -- four 16-byte global
    // loads per step into a three-deep register ring, issued two steps ahead. No weight stage
    // means no per-step barrier: the workgroup synchronises once per channel chunk (9 steps),
    // when the patch buffers swap.
    const int n_col = tile_n * BN + wn * 32 + lr;
    const float* wrow = p.wgt + (size_t)n_col * p.ktot + lh * 4;
    f32x4 breg[3][4];
#define PC_LOAD_B(STAGE, CH, TAP)                                                                  \
    {                                                                                              \
        const float* w_ = wrow + (TAP) * p.chunk + (CH) * 32;                                      \
        _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_)                                        \
            breg[STAGE][kk_] = *reinterpret_cast<const f32x4*>(w_ + kk_ * 8);                      \
    }

#define PC_LOAD_B2(STAGE, J)                                                                       \
    {                                                                                              \
        const int j_ = j0 + (J) < p.k2_steps ? j0 + (J) : p.k2_steps - 1;                          \
        const float* w_ = wrow + 9 * p.chunk + j_ * 32;                                            \
        _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_)                                        \
            breg[STAGE][kk_] = *reinterpret_cast<const f32x4*>(w_ + kk_ * 8);                      \
    }

    // patch pixel of this lane's MFMA row (tap 0,0) for each 32-row group
    int pbase[MI], rbase[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        int m = tile_m * BM + wm * (BM / WM) + mi * 32 + lr;
        m = m < p.M ? m : p.M - 1;
        if (BLK) {
            const int ml = wm * (BM / WM) + mi * 32 + lr;   // row of the tile
            const int rem = ml & ((1 << p.blk_shift_px) - 1);
            const int r = rem >> p.blk_shift_c, c = rem & (p.blk_cols - 1);
            pbase[mi] = (ml >> p.blk_shift_px) * p.img_px_patch + r * p.patch_pitch + c;
            rbase[mi] = p.swz_a * r + c;
        } else {
            int img, oy, ox;
            split_m(p, m, img, oy, ox);
            pbase[mi] = img * p.img_px + oy * p.patch_pitch + ox - p0;
            rbase[mi] = p.swz_a * oy + ox;
        }
    }

    f32x16 acc[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][e] = 0.f;

    const int npass = p.patch_slots >> 5;            // DMA passes per patch (<= 9); slots are a multiple of 32

    // ---- prologue: first patch, weights of the first two steps ------------------------------
    if (ch_begin < ch_end) {
        if (BLK) {
#pragma unroll
            for (int qq = 0; qq < 7; ++qq)
                if (qq < npass) PC_PATCH_PASS(ch_begin, 0, qq);
        } else {
            for (int qq = 0; qq < npass; ++qq) PC_PATCH_PASS(ch_begin, 0, qq);
        }
        PC_LOAD_B(0, ch_begin, 0);
        PC_LOAD_B(1, ch_begin, 1);
    }
    __syncthreads();
#ifdef PA_STAMP_BUILD
    const unsigned long long st1 = __builtin_amdgcn_s_memrealtime();
#endif

    int pb = 0;
    for (int ch = ch_begin; ch < ch_end; ++ch) {
        const float* patch = patch0 + pb * PP;
        const int chn = ch + 1 < ch_end ? ch + 1 : ch;
        const bool to_k2 = K2 && ch + 1 == ch_end && k2n > 0;  // last 3x3 chunk: prefetch the second source instead
        f32x4 af[2][MI];
#define PC_LOAD_A(SET, TAP, KK)                                                                    \
    {                                                                                              \
        const int toff_ = ((TAP) / 3) * p.patch_pitch + ((TAP) % 3);                               \
        const int roff_ = ((TAP) / 3) * p.swz_a + ((TAP) % 3);                                     \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) {                                        \
            const int px_ = pbase[mi] + toff_;                                                     \
            const int key_ = ((rbase[mi] + roff_) >> 1) & 7;                                       \
            af[SET][mi] = *reinterpret_cast<const f32x4*>(patch + px_ * 32 + ((((KK) * 2 + lh) ^ key_) << 2)); \
        }                                                                                          \
    }
        PC_LOAD_A(0, 0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // loads for later steps first: weights of step +2 (ring stage (tap + 2) % 3), and one
            // pass of the next chunk's patch per tap
            // (behind the last chunk both re-fetch that chunk, unused: the 9-tap body then has no
            // data-dependent branch, which keeps hipcc's vmcnt bookkeeping exact)
            if (tap + 2 < 9) {
                PC_LOAD_B((tap + 2) % 3, ch, tap + 2);
            } else if (to_k2) {
                PC_LOAD_B2((tap + 2) % 3, tap + 2 - 9);
            } else {
                PC_LOAD_B((tap + 2) % 3, chn, tap + 2 - 9);
            }
            if (to_k2) {
                if (tap < ROWS2) PC_TILE2_PASS(0, pb ^ 1, tap);
            } else if (tap < npass && (!BLK || tap < 7)) {
                PC_PATCH_PASS(chn, pb ^ 1, tap < 7 || !BLK ? tap : 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int g = tap * 4 + kk;  // fragment set alternates over the whole chunk
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const f32x4 a4 = af[g & 1][mi], b4 = breg[tap % 3][kk];
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.x, a4.x, acc[mi], 0, 0, 0);
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.y, a4.y, acc[mi], 0, 0, 0);
                    if (mi == 0) {
                        // next group's operands (next tap's first group at kk == 3): same patch
                        // buffer, no barrier in between
                        __builtin_amdgcn_sched_barrier(0);
                        if (kk + 1 < 4) {
                            PC_LOAD_A((g + 1) & 1, tap, kk + 1);
                        } else if (tap + 1 < 9) {
                            PC_LOAD_A((g + 1) & 1, tap + 1, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.z, a4.z, acc[mi], 0, 0, 0);
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.w, a4.w, acc[mi], 0, 0, 0);
                }
            }
        }
#undef PC_LOAD_A
        // the passes of the next patch (npass == 9: the last one was issued in tap 8) must have
        // landed, and every wave must be done with this patch before it is overwritten
        __syncthreads();
        pb ^= 1;
    }
    // ---- second-source steps: one 32-channel step each, tile j+1 and weights j+2 in flight ----
    if (K2 && k2n > 0) {
        if (ch_begin >= ch_end) {  // (split without 3x3 chunks: nothing was prefetched)
#pragma unroll
            for (int qq = 0; qq < ROWS2; ++qq) PC_TILE2_PASS(0, pb, qq);
            PC_LOAD_B2(0, 0);
            if (k2n > 1) PC_LOAD_B2(1, 1);
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < k2n) {
                const float* tile2 = patch0 + pb * PP;
                if (j + 2 < k2n) PC_LOAD_B2((j + 2) % 3, j + 2);
                if (j + 1 < k2n) {
#pragma unroll
                    for (int qq = 0; qq < ROWS2; ++qq) PC_TILE2_PASS(j + 1, pb ^ 1, qq);
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x4 a2[2][MI];
#define PC_LOAD_A2(SET, KK)                                                                        \
    {                                                                                              \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) {                                        \
            const int r_ = wm * (BM / 2) + mi * 32 + lr;                                           \
            a2[SET][mi] = *reinterpret_cast<const f32x4*>(tile2 + r_ * 32 + ((((KK) * 2 + lh) ^ ((r_ >> 1) & 7)) << 2)); \
        }                                                                                          \
    }
                PC_LOAD_A2(0, 0);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        const f32x4 a4 = a2[kk & 1][mi], b4 = breg[j % 3][kk];
                        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.x, a4.x, acc[mi], 0, 0, 0);
                        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.y, a4.y, acc[mi], 0, 0, 0);
                        if (mi == 0 && kk + 1 < 4) {
                            __builtin_amdgcn_sched_barrier(0);
                            PC_LOAD_A2((kk + 1) & 1, kk + 1);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.z, a4.z, acc[mi], 0, 0, 0);
                        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.w, a4.w, acc[mi], 0, 0, 0);
                    }
                }
#undef PC_LOAD_A2
                __syncthreads();
                pb ^= 1;
            }
        }
    }
#undef PC_PATCH_PASS
#undef PC_LOAD_B
#undef PC_LOAD_B2
#undef PC_TILE2_PASS
#ifdef PA_STAMP_BUILD
    const unsigned long long st2 = __builtin_amdgcn_s_memrealtime();
#endif

    // Epilogue. The matrix instructions take the WEIGHTS as their row operand and the pixels as their column
    // operand, so a lane owns ONE pixel (lane & 31 of its 32-pixel block) and, per accumulator group g = e >> 2,
    // FOUR CONSECUTIVE output channels 8 g + 4 (lane >> 5) + (e & 3) = 16 contiguous bytes of the NHWC output:
    // bias + residual + ReLU and the store happen straight from the accumulators with dwordx4 accesses (4 MI per
    // lane), no transposition through LDS, no barrier. (With the pixels as the row operand a lane held one channel
    // of 16 pixels = 32 four-byte stores, which queued up when a whole launch reached its epilogue together -- 8 us
    // median of a 46 us workgroup -- and the LDS transposition that replaced them still cost two barriers.)
    const bool direct_out = KS2 || p.splitk <= 1;
    if (KS2) {
        // the two halves of K meet in LDS: half 1 parks its accumulators lane-for-lane, half 0 adds them
        f32x4* xb = reinterpret_cast<f32x4*>(pc_lds);  // every wave left the k loop through the final barrier
        if (half == 1) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    xb[(mi * 4 + g) * 256 + tid] = f32x4{acc[mi][4 * g], acc[mi][4 * g + 1], acc[mi][4 * g + 2], acc[mi][4 * g + 3]};
        }
        __syncthreads();
        if (half == 1) return;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 o = xb[(mi * 4 + g) * 256 + tid];
                acc[mi][4 * g] += o.x; acc[mi][4 * g + 1] += o.y; acc[mi][4 * g + 2] += o.z; acc[mi][4 * g + 3] += o.w;
            }
    }
    const int ch0 = tile_n * BN + wn * 32 + 4 * lh;  // this lane's channels: ch0 + 8 g + 0..3
    f32x4 bias4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
        bias4[g] = (direct_out && p.bias) ? *reinterpret_cast<const f32x4*>(p.bias + ch0 + 8 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = tile_m * BM + wm * (BM / WM) + mi * 32 + lr;
        if (m >= p.M) continue;
        int img, oy, ox;
        if (BLK) split_m_blk(p, m, img, oy, ox); else split_m(p, m, img, oy, ox);
        const int o_px = img * p.out_img_stride + (oy + p.out_pad) * p.out_row_stride + (ox + p.out_pad) * p.out_px_stride + ch0;
        f32x4 res4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            res4[g] = (direct_out && p.residual) ? *reinterpret_cast<const f32x4*>(p.residual + o_px + 8 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v = f32x4{acc[mi][4 * g], acc[mi][4 * g + 1], acc[mi][4 * g + 2], acc[mi][4 * g + 3]};
            if (!direct_out) {
                *reinterpret_cast<f32x4*>(p.slab + ((size_t)z * p.M + m) * p.N + ch0 + 8 * g) = v;
            } else {
                if (BLK) {  // the detection network's epilogues: SiLU, residual after the activation
                    v += p.res_after ? bias4[g] : bias4[g] + res4[g];
                    if (p.relu == 1) {
                        v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
                        v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
                    } else if (p.relu == 2) {
                        v.x = silu_fast(v.x); v.y = silu_fast(v.y);
                        v.z = silu_fast(v.z); v.w = silu_fast(v.w);
                    }
                    if (p.res_after) v += res4[g];
                } else {
                    v += bias4[g] + res4[g];
                    if (p.relu) {
                        v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
                        v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
                    }
                }
                *reinterpret_cast<f32x4*>(p.out + o_px + 8 * g) = v;
            }
        }
    }
#ifdef PA_STAMP_BUILD
    if (p.clk && threadIdx.x == 0) {
        unsigned long long* o = p.clk + (size_t)blockIdx.x * 6;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = __builtin_amdgcn_s_memrealtime();
        o[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
        o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
    }
#endif
}

// p: GemmParams as for launch_igemm (conv mode, 3x3, stride 1, chunk = Cin, ktot = 9*Cin + 32*k2_steps),
// bm: 128 or 64. Fills the patch geometry, picks the split and launches; the ordered split-K
// reduce is igemm.hip's.
hipError_t launch_conv3x3_patch(const GemmParams& p_in, int bm, hipStream_t s) {
    GemmParams p = p_in;
    if (p.gather || p.taps != 9 || p.kw_taps != 3 || p.stride != 1 || p.chunk % 32 != 0 || p.N % 64 != 0 || p.M <= 0 ||
        p.k2_steps < 0 || p.k2_steps > 8 || p.k2_steps == 1 || p.ktot != 9 * p.chunk + 32 * p.k2_steps ||
        (p.k2_steps && !p.act2) || (bm != 128 && bm != 64))
        return hipErrorInvalidValue;
    auto ilog2 = [](int v) { int sh = 0; while ((1 << sh) < v) ++sh; return (1 << sh) == v ? sh : -1; };
    p.howo_shift = ilog2(p.howo);
    p.wo_shift = ilog2(p.wo);
    if (p.howo_shift < 0 || p.wo_shift < 0) p.howo_shift = p.wo_shift = -1;
    const int ho = p.howo / p.wo;
    p.patch_pitch = p.in_row_stride / p.in_px_stride;  // W + 2
    p.img_px = p.in_img_stride / p.in_px_stride;       // (H + 2) * (W + 2)
    int patch_px;
    if (p.howo >= bm) {  // a tile is a band of rows of one image
        if (p.howo % bm != 0 || bm % p.wo != 0) return hipErrorInvalidValue;
        const int rows = bm / p.wo;
        p.tiles_per_img = p.howo / bm;
        p.p0_img = p.img_px;
        p.p0_row = rows * p.patch_pitch;
        patch_px = (rows + 2) * p.patch_pitch;
    } else {  // a tile is a run of whole images
        if (bm % p.howo != 0) return hipErrorInvalidValue;
        const int imgs = bm / p.howo;
        p.tiles_per_img = 1;
        p.p0_img = imgs * p.img_px;
        p.p0_row = 0;
        patch_px = imgs * p.img_px;
    }
    (void)ho;
    p.patch_slots = (patch_px + 31) & ~31;  // whole 32-pixel DMA passes (no partially masked wave instruction)
    p.swz_a = p.wo & 15;
    p.magic_pitch = (65536 + p.patch_pitch - 1) / p.patch_pitch;  // x / pitch == (x * magic) >> 16 for x < 1024
    if (p.howo >= bm) {
        p.img_px_patch = 1 << 20;  // one band of one image: no image split inside the patch
        p.magic_img = 0;
    } else {
        p.img_px_patch = p.img_px;
        p.magic_img = (65536 + p.img_px - 1) / p.img_px;
    }
    p.total_px = (p.M / p.howo) * p.img_px;
    if (p.M % p.howo != 0) return hipErrorInvalidValue;
    p.tiles_m = (p.M + bm - 1) / bm;
    p.tiles_n = p.N / 64;
    const int n_ch = p.chunk / 32;
    if (p.splitk < 1) p.splitk = 1;
    if (p.splitk > n_ch) p.splitk = n_ch;
    p.ksteps_per_split = (n_ch + p.splitk - 1) / p.splitk;  // chunks per split
    p.splitk = (n_ch + p.ksteps_per_split - 1) / p.ksteps_per_split;
    if (p.patch_slots > (bm == 128 ? PatchCap<128>::slots : PatchCap<64>::slots)) return hipErrorInvalidValue;
    const size_t lds_bytes = 0;
    // a two-way split of 64-row tiles is done inside 512-thread workgroups (no slabs, no reduce
    // kernel) when the chunks and second-source steps divide evenly; PA_PATCH_KS2=0 disables (A/B)
    static const int want_ks2 = getenv("PA_PATCH_KS2") ? atoi(getenv("PA_PATCH_KS2")) : 1;
    const bool ks2 = want_ks2 && bm == 64 && p.splitk == 2 && n_ch % 2 == 0 && p.k2_steps % 2 == 0;
    if (ks2) {
        p.splitk = 1;
        p.ksteps_per_split = n_ch / 2;
    }
    const int grid = p.tiles_m * p.tiles_n * p.splitk;
    p.xcd_m = p.xcd_n = 0;
    if (p.splitk == 1 && grid % 8 == 0) {
        // bytes an XCD fetches for an (a x b) arrangement: input activations / a + weights / b; keep the default
        // (a = 8: contiguous runs) unless another divisor pair is at least 10 % cheaper
        const double act = (double)p.total_px * p.chunk * 4.0, wgt = (double)p.N * p.ktot * 4.0;
        double best = act / 8 + wgt;
        static const int use_grid = getenv("PA_XCD_GRID") ? atoi(getenv("PA_XCD_GRID")) : 1;
        for (int a = 4; a >= 1 && use_grid; a >>= 1) {
            const int bb = 8 / a;
            if (p.tiles_m % a || p.tiles_n % bb) continue;
            const double cost = act / a + wgt / bb;
            if (cost < 0.9 * best) {
                best = cost;
                p.xcd_m = a;
                p.xcd_n = bb;
            }
        }
    }
#ifdef PA_STAMP_BUILD
    // timeline stamps of every workgroup of launch number PA_STAMP_CALL, written to PA_STAMP_FILE
    static int stamp_calls = 0;
    static unsigned long long* stamp_dev = nullptr;
    const char* sf = getenv("PA_STAMP_FILE");
    const bool stamp_now = sf && getenv("PA_STAMP_CALL") && stamp_calls++ == atoi(getenv("PA_STAMP_CALL"));
    if (stamp_now) {
        if (!stamp_dev) (void)hipMalloc(&stamp_dev, (size_t)65536 * 6 * 8);
        p.clk = stamp_dev;
    }
#endif
#define PC_LAUNCH(BM_, K2_, KS2_, GRID_, THREADS_) \
    hipLaunchKernelGGL((conv3x3_patch_kernel<BM_, K2_, KS2_>), dim3(GRID_), dim3(THREADS_), lds_bytes, s, p)
    if (ks2) {
        if (p.k2_steps) PC_LAUNCH(64, true, true, grid, 512); else PC_LAUNCH(64, false, true, grid, 512);
    } else if (bm == 128) {
        if (p.k2_steps) PC_LAUNCH(128, true, false, grid, 256); else PC_LAUNCH(128, false, false, grid, 256);
    } else {
        if (p.k2_steps) PC_LAUNCH(64, true, false, grid, 256); else PC_LAUNCH(64, false, false, grid, 256);
    }
#undef PC_LAUNCH
#ifdef PA_STAMP_BUILD
    if (stamp_now) {
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> h((size_t)grid * 6);
        (void)hipMemcpy(h.data(), stamp_dev, h.size() * 8, hipMemcpyDeviceToHost);
        FILE* f = fopen(sf, "wb");
        if (f) {
            int hdr[4] = {grid, bm, p.M, p.chunk};
            fwrite(hdr, 4, 4, f);
            fwrite(h.data(), 8, h.size(), f);
            fclose(f);
        }
    }
#endif
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
    if (p.splitk > 1) return launch_splitk_reduce(p, s);
    return hipSuccess;
}

// Blocked form for maps of any width (see GemmParams::blk_*): 8 x 16 blocks in 128-row tiles (LDS image of a 16-wide map),
// 8 x 8 blocks (one or two per tile: an 8-wide map's) or 4 x 4 blocks (four per 64-row tile: a 4-wide map's) -- the three
// layouts whose chunk swizzles were searched, so the operand reads stay free of bank conflicts.
hipError_t launch_conv3x3_patch_blocked(const GemmParams& p_in, hipStream_t s) {
    GemmParams p = p_in;
    if (p.gather || p.taps != 9 || p.kw_taps != 3 || p.stride != 1 || p.chunk % 32 != 0 || p.N % 32 != 0 || p.M <= 0 || p.k2_steps ||
        p.ktot != 9 * p.chunk || p.off_y != 0 || p.off_x != 0 || p.M >= (1 << 24))
        return hipErrorInvalidValue;
    const int bn = p.N % 64 == 0 ? 64 : 32;
    const int ho = p.howo / p.wo;
    if (p.M % p.howo != 0) return hipErrorInvalidValue;
    int br, bc, bm;
    if (p.wo % 16 == 0 && ho % 8 == 0) { br = 8; bc = 16; bm = 128; }
    else if (p.wo % 8 == 0 && ho % 8 == 0) { br = 8; bc = 8; bm = ((p.M + 127) / 128) * (p.N / 64) >= 512 || bn == 32 ? 128 : 64; }
    else if (p.wo % 4 == 0 && ho % 4 == 0 && bn == 64) { br = 4; bc = 4; bm = 64; }
    else return hipErrorInvalidValue;
    auto ilog2 = [](int v) { int sh = 0; while ((1 << sh) < v) ++sh; return sh; };
    p.blk_rows = br; p.blk_cols = bc;
    p.blk_shift_c = ilog2(bc); p.blk_shift_px = ilog2(br * bc);
    p.blk_per_row = p.wo / bc;
    p.blk_per_img = (ho / br) * p.blk_per_row;
    p.n_blocks = p.M / (br * bc);
    p.src_pitch = p.in_row_stride / p.in_px_stride;
    p.img_px = p.in_img_stride / p.in_px_stride;
    p.patch_pitch = bc + 2;
    const int sub_px = (br + 2) * (bc + 2);
    p.img_px_patch = sub_px;
    p.magic_img = (65536 + sub_px - 1) / sub_px;
    p.magic_pitch = (65536 + p.patch_pitch - 1) / p.patch_pitch;
    p.swz_a = bc & 15;
    p.patch_slots = ((bm / (br * bc)) * sub_px + 31) & ~31;
    if (p.patch_slots > (bm == 128 ? PatchCap<128>::slots : PatchCap<64>::slots) || p.patch_slots > 7 * 32) return hipErrorInvalidValue;
    p.total_px = (p.M / p.howo) * p.img_px;
    p.howo_shift = p.wo_shift = -1;
    p.tiles_m = (p.M + bm - 1) / bm;
    p.tiles_n = p.N / bn;
    p.tiles_per_img = 1; p.p0_img = 0; p.p0_row = 0;
    p.splitk = 1;
    p.ksteps_per_split = p.chunk / 32;
    const int grid = p.tiles_m * p.tiles_n;
    p.xcd_m = p.xcd_n = 0;
    if (grid % 8 == 0) {  // (as launch_conv3x3_patch: the XCDs as a grid over pixel and channel tiles when that fetches less)
        const double act = (double)p.total_px * p.chunk * 4.0, wgt = (double)p.N * p.ktot * 4.0;
        double best = act / 8 + wgt;
        for (int a = 4; a >= 1; a >>= 1) {
            const int bb = 8 / a;
            if (p.tiles_m % a || p.tiles_n % bb) continue;
            const double cost = act / a + wgt / bb;
            if (cost < 0.9 * best) { best = cost; p.xcd_m = a; p.xcd_n = bb; }
        }
    }
    if (bn == 32) hipLaunchKernelGGL((conv3x3_patch_kernel<128, false, false, true, 32>), dim3(grid), dim3(256), 0, s, p);
    else if (bm == 128) hipLaunchKernelGGL((conv3x3_patch_kernel<128, false, false, true>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv3x3_patch_kernel<64, false, false, true>), dim3(grid), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace pa
