// Stride-1 3x3 convolution on bf16 activations / weights for gfx950 with the activation patch
// resident in LDS across the nine taps (BASELINE.json configs[2], "bf16 conv path").
//
// igemm_bf16.hip stages an im2col tile and a weight tile per 64-deep k-step, so every activation
// byte crosses L2 -> LDS nine times and a 128 x 64 tile pays 24 KB of LDS fill for 8 matrix
// instructions per wave: the matrix pipe idles at ~0.19 (rocprofv3, round 1). This kernel turns the
// K loop inside out like the fp32 patch kernel (patchconv.hip), with what bf16 rates demand on top:
//
//  * tile = 64*WAVES output pixels x 64 output channels; one wave = 64 pixels x 64 channels
//    (2 x 2 accumulators of v_mfma_f32_32x32x16_bf16): the weight stage of a k-step is shared by
//    every wave of the workgroup, so weight fill per matrix instruction falls with the pixel tile;
//  * K runs over 32-channel chunks; per chunk the tile's input patch -- (rows + 2) x (W + 2) padded
//    pixels x 64 B, a CONTIGUOUS pixel range of the zero-bordered NHWC buffer (whole padded images
//    for the 16-, 8- and 4-wide maps) -- is copied to LDS once (double buffered) and the nine taps
//    read their operands from it at pixel offset ky*(W+2)+kx: 1/9 of the activation fill;
//  * weights go through LDS in stages of THREE taps (one ky row, 12 KB) in a three-deep ring, one
//    workgroup barrier per stage = per 24 matrix instructions of a wave; every copy is LDS-DMA
//    (buffer_load ... lds) with counted vmcnt waits and a raw s_barrier, so the copies of stage
//    t+2 and of the next chunk's patch stay in flight across the barriers;
//  * the matrix instruction takes the WEIGHTS as its row operand and the pixels as its column
//    operand: a lane then owns one pixel and runs of four consecutive output channels, which two
//    v_permlane32_swap turn into eight consecutive channels = one 16-byte store per lane. No LDS
//    transposition, no barrier, no second pass in the epilogue;
//  * 16-byte slots of a pixel / weight row are XOR-swizzled on the DMA source so that every
//    ds_read_b128 lane group hits 16 distinct bank slots for all nine taps on every map width
//    (keys brute-forced by scripts/lds_swizzle_search.py).
//
// fp32 accumulation, bias + residual + ReLU in fp32, one rounding to bf16 at the store.
#include "pa_kernels.h"

namespace pa {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint16_t bf16_t;  // storage

namespace {

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, int voff_bytes, int soff_bytes, uint8_t* lds_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_base, 16, voff_bytes,
                                             soff_bytes, 0, 0);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {  // round to nearest even (finite inputs)
    uint32_t ua = __float_as_uint(a), ub = __float_as_uint(b);
    ua += 0x7fffu + ((ua >> 16) & 1u);
    ub += 0x7fffu + ((ub >> 16) & 1u);
    return (ua >> 16) | (ub & 0xffff0000u);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    // counted wait: all but the N youngest vector-memory operations of this wave are done
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}

// Geometry of one map width. A tile is 64*WAVES consecutive output pixels in (image, row, column)
// raster order: a band of rows of one image (W = 32), or whole images (W <= 16).
template <int W, int WAVES> struct PatchGeom {
    static constexpr int PXT = 64 * WAVES;
    static constexpr int PITCH = W + 2;
    static constexpr int IMG_PX = PITCH * PITCH;
    static constexpr int HW = W * W;
    static constexpr bool BAND = HW > PXT;                       // tile = rows of one image
    static constexpr int ROWS = BAND ? PXT / W : W;              // image rows per tile (band) / per image
    static constexpr int IMGS = BAND ? 1 : PXT / HW;             // images per tile
    static constexpr int TILES_PER_IMG = BAND ? HW / PXT : 1;
    static constexpr int PATCH_PX = BAND ? (ROWS + 2) * PITCH : IMGS * IMG_PX;
    static constexpr int NPIECE = (PATCH_PX + 15) / 16;          // 1 KB DMA pieces (16 pixels x 64 B)
    static constexpr int PATCH_BYTES = NPIECE * 1024;
    // swizzle key of padded pixel (row r, column c): ((c >> 2) * KA + r * KB) & 3 (lds_swizzle_search.py)
    static constexpr int KA = W == 32 ? 1 : (W == 16 ? 2 : 0);
    static constexpr int KB = W == 32 ? 0 : 1;
    static_assert(W == 32 || W == 16 || W == 8 || W == 4, "map widths of the ResNet-18 stages at 128 x 128 input");
    static_assert(BAND ? (PXT % W == 0 && HW % PXT == 0) : (PXT % HW == 0), "tile must be whole rows / whole images");
};

template <int CN> constexpr int WSTAGE_BYTES_OF = 3 * (32 * CN) * 64;  // three taps x 32 CN output channels x 32 bf16
constexpr int WSTAGE_BYTES = WSTAGE_BYTES_OF<2>;
constexpr int NSTG = 3;                    // weight ring depth

}  // namespace

// LDS: [patch 0][patch 1][weight ring]. One array (a second __shared__ object makes hipcc drain vmcnt
// in front of the operand reads).
//
// Workgroups are PERSISTENT over a run of consecutive pixel tiles of one 64-channel column: the stage
// sequence simply continues into the next tile (same weights, next tile's patch), so a tile's first
// patch and first weight stages arrive under the previous tile's last stages and the pipeline never
// drains; the epilogue stores straight from the accumulators in between. Everything lane-dependent
// (patch pixel of a lane, swizzle keys, DMA source offsets) is tile-independent; a tile contributes only
// scalar offsets.
//
// WRES (the 64 -> 64 channel convolutions of layer 1, whose whole filter bank is 73.7 KB): the weights are copied to
// LDS ONCE per workgroup and stay there, [chunk][tap][64 channels][64 B]; only activation patches stream, the
// stage ring, its copies, counted waits and per-stage barriers disappear (one barrier per chunk = 72 matrix
// instructions per wave), and a 512-pixel tile amortises the halo rows. These layers are HBM-bound.
//
// CN (round 5): 32-channel blocks per wave. CN = 2 is the kernel of rounds 2-4 (a wave = 64 pixels x 64 channels). CN = 4 makes a
// workgroup 64 WAVES pixels x 128 channels SHARING ONE PATCH -- the prototype round 4's verdict asked for: the patch crosses L2 ->
// LDS once per 128 channels instead of once per 64, the weight stage (24 KB) is shared by eight waves, 48 matrix instructions per
// wave and barrier. It holds 128 accumulator registers, so bias and residual are fetched in the epilogue, not ahead.
template <int W, int WAVES, bool WRES, int CN = 2>
__global__ __launch_bounds__(64 * WAVES) void conv3x3_bf16_patch_kernel(const GemmParams p) {
    using G = PatchGeom<W, WAVES>;
    static_assert(CN == 2 || (CN == 4 && !WRES), "");
    constexpr int BN = 32 * CN;                       // output channels per workgroup
    constexpr int TAPB = BN * 64;                     // bytes of one tap of a weight stage (BN rows x 64 B)
    constexpr int WSTAGE_BYTES = WSTAGE_BYTES_OF<CN>;
    constexpr int WRES_BYTES = 2 * 9 * 4096;  // two 32-channel chunks x nine taps x (64 rows x 64 B)
    constexpr int LDS_BYTES = 2 * G::PATCH_BYTES + (WRES ? WRES_BYTES : NSTG * WSTAGE_BYTES);
    __shared__ __attribute__((aligned(1024))) uint8_t lds[LDS_BYTES];
    uint8_t* const wring = lds + 2 * G::PATCH_BYTES;

    const bf16_t* residual = reinterpret_cast<const bf16_t*>(p.residual);
    bf16_t* out = reinterpret_cast<bf16_t*>(p.out);
    const int C = p.chunk;          // input channels
    const int n_ch = C >> 5;        // 32-channel chunks

    // XCD-aware (bijective) remap: blocks with equal b % 8 share an XCD and get a contiguous run of work
    // items; item v = (tile group, channel column): the columns of one tile group sit on one XCD (they
    // read the same patches), the weights of a column stay in that XCD's L2 for the whole group.
    const int nwg = gridDim.x;
    const int b = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
    const int v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
    const int grp = v / p.tiles_n;
    const int tile_n = v - grp * p.tiles_n;
    const int tile_first = grp * p.tiles_per_img;                      // (field reused: tiles per workgroup)
    int tile_cnt = p.tiles_m - tile_first;
    tile_cnt = tile_cnt < p.tiles_per_img ? tile_cnt : p.tiles_per_img;
    if (tile_cnt <= 0) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;

    // out-of-range bytes of the activation buffer (the last tile's patch pieces run past it) read as zero
    const __amdgpu_buffer_rsrc_t act_rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.act), 0, p.total_px * C * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t wgt_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wgt), 0, -1, 0x00020000);

    // byte offset of tile t's patch origin (its first padded pixel) in the activation buffer
    auto patch_origin = [&](int t) -> int {
        const int p0 = G::BAND ? (t / G::TILES_PER_IMG) * G::IMG_PX + (t % G::TILES_PER_IMG) * G::ROWS * G::PITCH
                               : t * G::IMGS * G::IMG_PX;
        return p0 * C * 2;
    };

    // ---- DMA jobs. A stage's copies are a flat list: 12 weight pieces (3 taps x 4 groups of 16 output
    // channels), then -- in the ky = 0 and ky = 1 stages -- half of the next patch's pieces. Wave w takes
    // jobs w, w + WAVES, ...; every wave issues the same NUMBER of copies per stage (the counted vmcnt
    // waits rely on it), surplus slots repeat the list's last piece.
    constexpr int HALF = WRES ? G::NPIECE : (G::NPIECE + 1) / 2;   // WRES: the whole next patch in the ky = 0 stage
    constexpr int NGRP = 2 * CN;                                    // groups of 16 output channels per tap
    constexpr int NWJ = WRES ? 0 : 3 * NGRP;                        // weight pieces per stage
    constexpr int CNT_P = (NWJ + HALF + WAVES - 1) / WAVES;  // copies per wave in a stage that also moves patch pieces
    constexpr int CNT_W = (NWJ + WAVES - 1) / WAVES;         // ... in the ky = 2 stage
    static_assert(CNT_P <= 12, "extend wait_vmcnt");
    // patch piece q: patch pixel 16 q + (lane >> 2), slot lane & 3. The swizzle key of a pixel depends on
    // its padded (row, column); with the tile origins this geometry allows it is the same for every tile.
    auto patch_voff = [&](int q) -> int {
        const int pp = 16 * q + (lane >> 2);
        const int rem = pp % G::IMG_PX;             // BAND: patch rows of one image, origin at column 0
        const int r = rem / G::PITCH, c = rem - r * G::PITCH;
        const int key = ((c >> 2) * G::KA + r * G::KB) & 3;
        return pp * C * 2 + (((lane & 3) ^ key) << 4);
    };
    static_assert(!G::BAND || G::KB == 0, "a row band's origin row must not enter the swizzle key");
    int pvoff[2][CNT_P];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < CNT_P; ++i) {
            int job = wave + WAVES * i;
            job = job < NWJ + HALF ? job : NWJ + HALF - 1;
            int q = job - NWJ + h * HALF;
            q = q < G::NPIECE ? q : G::NPIECE - 1;
            pvoff[h][i] = job >= NWJ ? patch_voff(q) : 0;
        }
    // weight piece job: tap kx = job >> 2, output channels 16 (job & 3) + (lane >> 2), slot (lane & 3) ^ ((n >> 2) & 3)
    const int wvoff_base = (tile_n * BN + (lane >> 2)) * p.ktot * 2;

    // copies of weight stage (chunk CH, tap row KY) into ring slot SLOT, plus -- WITH_PATCH -- half HALF_IDX of the
    // patch at byte offset PSOFF of the activation buffer into patch buffer PB
#define BP_ISSUE(CH, KY, SLOT, WITH_PATCH, HALF_IDX, PSOFF, PB)                                                      \
    {                                                                                                                \
        constexpr int CNT_ = (WITH_PATCH) ? CNT_P : CNT_W;                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < CNT_; ++i_) {                                                        \
            int job_ = wave + WAVES * i_;                                                                            \
            const int last_ = (WITH_PATCH) ? NWJ + HALF - 1 : NWJ - 1;                                               \
            job_ = job_ < last_ ? job_ : last_;                                                                      \
            if (job_ < NWJ) {                                                                                        \
                const int kx_ = job_ / NGRP, grp_ = job_ - kx_ * NGRP;                                               \
                const int n_ = 16 * grp_ + (lane >> 2);                                                              \
                const int voff_ = wvoff_base + 16 * grp_ * p.ktot * 2 + (((lane & 3) ^ ((n_ >> 2) & 3)) << 4);       \
                dma16(wgt_rs, voff_, (((KY) * 3 + kx_) * C + (CH) * 32) * 2, wring + (SLOT) * WSTAGE_BYTES + job_ * 1024); \
            } else {                                                                                                 \
                int q_ = job_ - NWJ + (HALF_IDX) * HALF;                                                             \
                q_ = q_ < G::NPIECE ? q_ : G::NPIECE - 1;                                                            \
                dma16(act_rs, pvoff[HALF_IDX][i_], (PSOFF), lds + (PB) * G::PATCH_BYTES + q_ * 1024);                \
            }                                                                                                        \
        }                                                                                                            \
    }

    // ---- operand addresses (tile-independent) ---------------------------------------------------
    // activation fragment of pixel block pi (32 pixels), tap (ky, kx), k group kg: lane reads slot
    // (2 kg + lh) ^ key of patch pixel pbase[pi] + ky*PITCH + kx
    int pbase[2], prow[2], pcol[2], obase[2];
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
        const int ml = wave * 64 + pi * 32 + lr;         // pixel inside the tile
        const int img = ml / G::HW, rem = ml - img * G::HW;   // BAND: img == 0, rem = row * W + column inside the band
        const int oy = rem / W, ox = rem - oy * W;
        pbase[pi] = (img * G::IMG_PX + oy * G::PITCH + ox) * 64;
        prow[pi] = oy;
        pcol[pi] = ox;
        obase[pi] = img * p.out_img_stride + (oy + p.out_pad) * p.out_row_stride + (ox + p.out_pad) * p.out_px_stride +
                    tile_n * BN + 8 * lh;
    }
    // output offset of tile t's first pixel
    auto out_origin = [&](int t) -> int {
        return G::BAND ? (t / G::TILES_PER_IMG) * p.out_img_stride + (t % G::TILES_PER_IMG) * G::ROWS * p.out_row_stride
                       : t * G::IMGS * p.out_img_stride;
    };
    // weight fragment of channel block ci: row n = 32 ci + lr of the stage, slot (2 kg + lh) ^ ((n >> 2) & 3)
    int wbase[CN][2];
#pragma unroll
    for (int ci = 0; ci < CN; ++ci)
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {
            const int n = 32 * ci + lr;
            wbase[ci][kg] = n * 64 + (((2 * kg + lh) ^ ((n >> 2) & 3)) << 4);
        }
    // bias of this lane's 4 x 8 output channels (channel column fixed for the workgroup)
    constexpr int CB = CN == 2 ? 2 : 1;   // (CN = 4: fetched in the epilogue)
    float bias8[CB][2][8];
#pragma unroll
    for (int ci = 0; ci < CB; ++ci)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (CN != 2) break;
            const float* bp = p.bias + tile_n * BN + 32 * ci + 16 * j + 8 * lh;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
            bias8[ci][j][0] = b0.x; bias8[ci][j][1] = b0.y; bias8[ci][j][2] = b0.z; bias8[ci][j][3] = b0.w;
            bias8[ci][j][4] = b1.x; bias8[ci][j][5] = b1.y; bias8[ci][j][6] = b1.z; bias8[ci][j][7] = b1.w;
        }

    f32x16 acc[2][CN];
#pragma unroll
    for (int pi = 0; pi < 2; ++pi)
#pragma unroll
        for (int ci = 0; ci < CN; ++ci)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[pi][ci][e] = 0.f;

    u32x4 fa[2][2], fw[2][CN];  // [set][pi | ci]
#define BP_LOAD(SET, PATCH, WST, KY, KX, KG)                                                                         \
    {                                                                                                                \
        _Pragma("unroll") for (int pi_ = 0; pi_ < 2; ++pi_) {                                                        \
            const int key_ = (((pcol[pi_] + (KX)) >> 2) * G::KA + (prow[pi_] + (KY)) * G::KB) & 3;                   \
            fa[SET][pi_] = *reinterpret_cast<const u32x4*>((PATCH) + pbase[pi_] + ((KY) * G::PITCH + (KX)) * 64 +    \
                                                           (((2 * (KG) + lh) ^ key_) << 4));                         \
        }                                                                                                            \
        _Pragma("unroll") for (int ci_ = 0; ci_ < CN; ++ci_)                                                         \
            fw[SET][ci_] = *reinterpret_cast<const u32x4*>((WST) + (KX) * (WRES ? 4096 : TAPB) + wbase[ci_][KG]);    \
    }
#define BP_MFMA(SET)                                                                                                 \
    {                                                                                                                \
        _Pragma("unroll") for (int pi_ = 0; pi_ < 2; ++pi_)                                                          \
            _Pragma("unroll") for (int ci_ = 0; ci_ < CN; ++ci_)                                                     \
                acc[pi_][ci_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[SET][ci_]),    \
                                                                        __builtin_bit_cast(bf16x8, fa[SET][pi_]),    \
                                                                        acc[pi_][ci_], 0, 0, 0);                     \
    }

    // ---- prologue: patch of the first tile's chunk 0, weight stages 0 and 1 ----------------------------
    {
        const int ps0 = patch_origin(tile_first);
        for (int q = wave; q < G::NPIECE; q += WAVES) dma16(act_rs, patch_voff(q), ps0, lds + q * 1024);
        if constexpr (WRES) {
            // the whole filter bank: piece j = (chunk * 9 + tap) * 4 + group of 16 output channels
            for (int j = wave; j < 2 * 9 * 4; j += WAVES) {
                const int tc = j >> 2, grp_ = j & 3, chn = tc / 9, tap = tc - chn * 9;
                const int n = 16 * grp_ + (lane >> 2);
                const int voff = wvoff_base + 16 * grp_ * p.ktot * 2 + (((lane & 3) ^ ((n >> 2) & 3)) << 4);
                dma16(wgt_rs, voff, (tap * C + chn * 32) * 2, wring + j * 1024);
            }
        } else {
            for (int job = wave; job < NWJ; job += WAVES) {
                const int kx = job / NGRP, grp_ = job - kx * NGRP;
                const int n = 16 * grp_ + (lane >> 2);
                const int voff = wvoff_base + 16 * grp_ * p.ktot * 2 + (((lane & 3) ^ ((n >> 2) & 3)) << 4);
                dma16(wgt_rs, voff, ((0 * 3 + kx) * C) * 2, wring + 0 * WSTAGE_BYTES + job * 1024);
                dma16(wgt_rs, voff, ((1 * 3 + kx) * C) * 2, wring + 1 * WSTAGE_BYTES + job * 1024);
            }
        }
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
    }
    BP_LOAD(0, lds, wring, 0, 0, 0);

    // ---- main loop over (tile, chunk); the three tap rows unrolled (ring slot of a stage is its ky) -----
    int pb = 0;  // patch buffer of the current chunk
    u32x4 res[2][CB][2];
    for (int tk = 0; tk < tile_cnt; ++tk) {
        const int tile = tile_first + tk;
        const int ps_tile = patch_origin(tile);
        const bool more_tiles = tk + 1 < tile_cnt;
        const int ps_next_tile = patch_origin(more_tiles ? tile + 1 : tile);
        for (int ch = 0; ch < n_ch; ++ch) {
            const uint8_t* patch = lds + pb * G::PATCH_BYTES;
            const int pb_next = pb ^ 1;
            const bool last_ch = ch + 1 == n_ch;
            // the patch that follows: next chunk of this tile, or chunk 0 of the next tile; behind the very
            // last chunk the current one is fetched again (unused) so that every stage issues the same copies
            const int ps_next = last_ch ? (more_tiles ? ps_next_tile : ps_tile + ch * 64) : ps_tile + (ch + 1) * 64;
            const bool has_next = !last_ch || more_tiles;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                // stage (ch, ky). Copies of the stage after next -> ring slot (ky + 2) % 3 (last read one stage
                // ago, whose closing barrier every wave has passed); in ky = 0, 1 also half of the next patch.
                // The weights wrap around at the end of a tile (same channel column).
#define BP_RESIDUAL()                                                                                                \
    {                                                                                                                \
        const int oo_ = out_origin(tile);                                                                            \
        _Pragma("unroll") for (int pi = 0; pi < 2; ++pi)                                                             \
            _Pragma("unroll") for (int ci = 0; ci < CB; ++ci)                                                        \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                        \
                    res[pi][ci][j] = *reinterpret_cast<const u32x4*>(residual + oo_ + obase[pi] + 32 * ci + 16 * j); \
    }
                // residual of this tile. Ring kernel: requested in the tile's last stage BEFORE that stage's copies, so
                // that the stage's counted wait also covers it (vmcnt is in order). WRES: one wait per chunk, so it is
                // requested at the start of the tile's last chunk, behind the patch copies, a whole chunk ahead.
                if (!WRES && CN == 2 && last_ch && ky == 2 && residual) BP_RESIDUAL();
                if constexpr (WRES) {
                    if (ky == 0) {
                        BP_ISSUE(0, 0, 0, true, 0, ps_next, pb_next);  // (only patch jobs exist)
                        if (last_ch && residual) BP_RESIDUAL();
                    }
                } else {
                    const int t2ky = (ky + 2) % 3;
                    int ch2 = ky == 0 ? ch : ch + 1;
                    ch2 = ch2 < n_ch ? ch2 : 0;
                    if (ky < 2) {
                        BP_ISSUE(ch2, t2ky, t2ky, true, ky, ps_next, pb_next);
                    } else {
                        BP_ISSUE(ch2, t2ky, t2ky, false, 0, ps_next, pb_next);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                // weights of this stage: ring slot ky, or the resident bank's (chunk, tap row)
                const uint8_t* wst = WRES ? wring + (ch * 9 + ky * 3) * 4096 : wring + ky * WSTAGE_BYTES;
                const uint8_t* wst_next = WRES ? wring + (ch * 9 + (ky + 1) * 3) * 4096 : wring + (ky + 1) * WSTAGE_BYTES;
                const uint8_t* wst_wrap = WRES ? wring + (last_ch ? 0 : (ch + 1) * 9) * 4096 : wring;  // first stage of the next chunk
#pragma unroll
                for (int g = 0; g < 6; ++g) {  // k group g = (kx, kg)
                    if (g < 5) {
                        // operands of the next group before this group's matrix instructions
                        BP_LOAD((g + 1) & 1, patch, wst, ky, (g + 1) >> 1, (g + 1) & 1);
                        __builtin_amdgcn_sched_barrier(0);
                        BP_MFMA(g & 1);
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                        // last group of the stage: its operands were requested one group ago. Close the stage --
                        // the copies of the next stage (issued a stage ago) have landed once only this stage's
                        // own copies are outstanding; every wave's reads of this stage are in registers -- then
                        // request the next stage's first operands and cover their latency with the last
                        // matrix instructions. (WRES: nothing to wait for inside a chunk; at its end the next
                        // patch -- and, behind a tile's last chunk, the residual -- must be there.)
                        if constexpr (WRES) {
                            if (ky == 2) {
                                wait_vmcnt<0>();
                                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                                __builtin_amdgcn_s_barrier();
                            }
                        } else {
                            if (ky < 2) wait_vmcnt<CNT_P>(); else wait_vmcnt<CNT_W>();
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            __builtin_amdgcn_s_barrier();
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (ky < 2) {
                            BP_LOAD(0, patch, wst_next, ky + 1, 0, 0);
                        } else if (has_next) {
                            BP_LOAD(0, lds + pb_next * G::PATCH_BYTES, wst_wrap, 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        BP_MFMA(1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            pb = pb_next;
        }
        // ---- epilogue of this tile: lane = one pixel; accumulator element e of block (pi, ci) is output
        // channel 32 ci + 8 (e >> 2) + 4 lh + (e & 3). Swapping halves between lanes l and l + 32 gives each
        // lane eight consecutive channels: 32 ci + 16 j + 8 lh + 0..7 (j = 0, 1) -> one 16-byte store. The next
        // tile's first copies are in flight meanwhile.
        const int oo = out_origin(tile);
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
            const bool valid = tile * G::PXT + wave * 64 + pi * 32 + lr < p.M;
#pragma unroll
            for (int ci = 0; ci < CN; ++ci)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float vv[8];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        // X = element group 2j (channels 16j + 4lh + i), Y = group 2j + 1 (channels 16j + 8 + 4lh + i)
                        const uint32_t x = __float_as_uint(acc[pi][ci][8 * j + i]);
                        const uint32_t y = __float_as_uint(acc[pi][ci][8 * j + 4 + i]);
                        const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
                        vv[i] = __uint_as_float(r[0]);      // lh = 0: own X (16j + i)       | lh = 1: partner's Y (16j + 8 + i)
                        vv[4 + i] = __uint_as_float(r[1]);  // lh = 0: partner's X (16j+4+i) | lh = 1: own Y (16j + 12 + i)
                    }
                    u32x4 rv = {0u, 0u, 0u, 0u};
                    float bb[8];
                    if constexpr (CN == 2) {
                        if (residual) rv = res[pi][ci][j];
#pragma unroll
                        for (int k = 0; k < 8; ++k) bb[k] = bias8[ci][j][k];
                    } else {
                        if (residual && valid) rv = *reinterpret_cast<const u32x4*>(residual + oo + obase[pi] + 32 * ci + 16 * j);
                        const float* bp = p.bias + tile_n * BN + 32 * ci + 16 * j + 8 * lh;
                        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
                        bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
                    }
                    uint32_t pk[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float a0 = vv[2 * k] + bb[2 * k] + __uint_as_float(rv[k] << 16);
                        float a1 = vv[2 * k + 1] + bb[2 * k + 1] + __uint_as_float(rv[k] & 0xffff0000u);
                        if (p.relu) {
                            a0 = a0 > 0.f ? a0 : 0.f;
                            a1 = a1 > 0.f ? a1 : 0.f;
                        }
                        pk[k] = pack_bf16x2(a0, a1);
                    }
                    if (valid) *reinterpret_cast<u32x4*>(out + oo + obase[pi] + 32 * ci + 16 * j) = u32x4{pk[0], pk[1], pk[2], pk[3]};
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[pi][ci][8 * j + e] = 0.f;
                }
        }
    }
#undef BP_LOAD
#undef BP_MFMA
#undef BP_ISSUE
#undef BP_RESIDUAL
    wait_vmcnt<0>();  // the tail's unused re-fetches must not outlive the workgroup's LDS
}

// p: GemmParams as for launch_igemm_bf16 (conv mode, 3x3, stride 1, no second source, chunk = Cin, ktot = 9 Cin,
// N % 64 == 0, bias != nullptr). Returns hipErrorInvalidValue for a geometry it does not cover (the caller
// then uses igemm_bf16.hip).
hipError_t launch_conv3x3_bf16_patch(const GemmParams& p_in, hipStream_t s) {
    GemmParams p = p_in;
    if (p.gather || p.taps != 9 || p.kw_taps != 3 || p.stride != 1 || p.chunk % 32 != 0 || p.N % 64 != 0 || p.M <= 0 ||
        p.k2_steps != 0 || p.ktot != 9 * p.chunk || !p.bias || p.wo * p.wo != p.howo || p.M % p.howo != 0)
        return hipErrorInvalidValue;
    const int W = p.wo;
    if (p.in_row_stride != (W + 2) * p.in_px_stride || p.in_img_stride != (W + 2) * (W + 2) * p.in_px_stride ||
        p.in_px_stride != p.chunk || p.off_y != 0 || p.off_x != 0)
        return hipErrorInvalidValue;
    p.total_px = (p.M / p.howo) * (W + 2) * (W + 2);
    if ((long long)p.total_px * p.chunk * 2 >= (1ll << 31)) return hipErrorInvalidValue;  // 32-bit buffer offsets
    static const int waves8 = getenv("PA_BF16_WAVES8") ? atoi(getenv("PA_BF16_WAVES8")) : 0;  // bit mask over {32,16,8}: flips the default
    static const int wres = getenv("PA_BF16_WRES") ? atoi(getenv("PA_BF16_WRES")) : 1;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidValue;
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    // PA_BF16_WIDE=1 (prototype, 16-wide maps with N % 128 == 0 only): eight waves over 512 pixels x 128 channels sharing one patch
    static const int wide = getenv("PA_BF16_WIDE") ? atoi(getenv("PA_BF16_WIDE")) : 0;
    if (wide && W == 16 && p.N % 128 == 0) {
        using G_ = PatchGeom<16, 8>;
        p.tiles_n = p.N / 128;
        p.tiles_m = (p.M + G_::PXT - 1) / G_::PXT;
        int slots_ = n_cu / p.tiles_n;
        slots_ = slots_ > 0 ? slots_ : 1;
        p.tiles_per_img = (p.tiles_m + slots_ - 1) / slots_;
        const int groups_ = (p.tiles_m + p.tiles_per_img - 1) / p.tiles_per_img;
        hipLaunchKernelGGL((conv3x3_bf16_patch_kernel<16, 8, false, 4>), dim3(groups_ * p.tiles_n), dim3(512), 0, s, p);
        return hipGetLastError();
    }
    p.tiles_n = p.N / 64;
    // persistent workgroups: WGPC per CU (what the LDS image allows), each a run of consecutive pixel tiles
    // of one channel column
#define BPL(W_, WV_) BPLX(W_, WV_, false)
#define BPLX(W_, WV_, WRES_)                                                                                         \
    {                                                                                                                \
        using G_ = PatchGeom<W_, WV_>;                                                                               \
        constexpr int lds_ = 2 * G_::PATCH_BYTES + ((WRES_) ? 2 * 9 * 4096 : NSTG * WSTAGE_BYTES);                   \
        const int wgpc_ = lds_ <= 81920 ? 2 : 1;                                                                     \
        p.tiles_m = (p.M + G_::PXT - 1) / G_::PXT;                                                                   \
        int slots_ = n_cu * wgpc_ / p.tiles_n;                                                                       \
        slots_ = slots_ > 0 ? slots_ : 1;                                                                            \
        p.tiles_per_img = (p.tiles_m + slots_ - 1) / slots_;          /* tiles per workgroup */                     \
        const int groups_ = (p.tiles_m + p.tiles_per_img - 1) / p.tiles_per_img;                                     \
        hipLaunchKernelGGL((conv3x3_bf16_patch_kernel<W_, WV_, WRES_>), dim3(groups_ * p.tiles_n), dim3(64 * WV_), 0, s, p); \
    }
    switch (W) {
        case 32:
            // layer 1 (64 -> 64 channels): the filter bank stays in LDS (PA_BF16_WRES=0: the ring kernel, for A/B runs)
            if (p.chunk == 64 && p.N == 64 && wres) BPLX(32, 8, true) else if (waves8 & 1) BPL(32, 8) else BPL(32, 4)
            break;
        case 16: if (waves8 & 2) BPL(16, 8) else BPL(16, 4) break;
        case 8: if (waves8 & 4) BPL(8, 4) else BPL(8, 8) break;
        case 4: BPL(4, 4) break;
        default: return hipErrorInvalidValue;
    }
#undef BPL
#undef BPLX
    return hipGetLastError();
}

}  // namespace pa
