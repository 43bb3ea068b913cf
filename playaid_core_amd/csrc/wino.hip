// Stride-1 3x3 convolution as Winograd F(2x2, 3x3) on the fp32 matrix cores (gfx950, fp32 throughout).
//
// A 3x3 convolution spends 9 multiply-adds per output value and input channel; Winograd's minimal filtering form spends
// 16 per 2x2 outputs = 4 per value: the input tile (4x4 pixels), the filter and the output tile are taken through
//     V = B^T d B          U = G g G^T (host, fp64, once)          Y = A^T (sum_cin U . V) A
// (Lavin & Gray's matrices; B and A hold only 0 / +-1, so both device-side transforms are additions). Between the two
// transforms stand 16 independent products [cout x cin] x [cin x tile], one per position of the 4x4 transformed tile --
// matrix-core work, 2.25x less of it than the direct form executes. Measured on the CPU restatement first (the whole
// ResNet-18 of the headline with every stride-1 3x3 in this form): max |dlogp| against a float64 run 3.7e-6, the direct
// fp32 form 3.3e-6 -- the rounding of the transforms does not show at the path's 1e-4 bar.
//
// Mapping. v_mfma_f32_16x16x4_f32 with the WEIGHTS as the row operand and the transformed tiles as the column operand:
// a lane then holds, for its tile (lane & 15) and four consecutive output channels 4 (lane >> 4) .. + 3, one accumulator
// quad per position. A wave owns 16 tiles x 32 channels x all 16 positions = 128 accumulator registers, so the OUTPUT
// transform, bias, residual, activation and the 16-byte stores run in the lane that accumulated -- no exchange, no LDS,
// no barrier behind the k loop. 16 tiles = four 4x4-pixel sub-blocks (2x2 tiles each, a 6x6 input patch each): any map
// whose sides are multiples of four.
//   workgroup = 8 waves: 4 tile groups (16 sub-blocks = 256 pixels) x 2 channel groups (64 channels), one to a CU
//            or 4 waves: 4 tile groups x 32 channels, two to a CU: 32-channel layers, and layers too small for the first form
//               (8 waves as 8 tile groups x 32 channels: the first shape of the 32-channel layers, kept behind PA_WINO_SMALL32=0)
//   Where even that leaves CUs without work the launcher splits K (ksplit workgroups per tile, see the epilogue).
// K loop over chunks of 8 input channels. A stage of the LDS ring holds, for one chunk,
//   U  [16 positions][BN / 16][16 channels][8 cin]   the transformed filters (BN = 64: 32 KB), the exact image the host laid
//                                                     out: the DMA is a straight copy (rows 8..15 of a 16-channel group
//                                                     are rotated by four dwords so that the 32 lanes of a ds_read_b64
//                                                     phase cover 64 distinct banks)
//   P  [sub-block][37 pixels][8 cin]                  the raw 6x6 patches (one pad pixel per sub-block spreads the tiles'
//                                                     pixels over the banks: two-way instead of four-way conflicts)
// both filled by buffer_load_dwordx4 ... lds. Per chunk a lane reads its tile's 16 raw pixels (ds_read_b64: two channels),
// transforms them (32 packed additions), and feeds 64 matrix instructions whose row operands come from U (32 ds_read_b64).
// The raw patch is the only activation traffic: 36 pixels per 16 outputs and chunk -- the patch-resident direct kernel
// (patchconv.hip) moves 1.4 pixels per output and chunk and nine operand reads per pixel; this one 2.25 and four.
//
// Built, measured and taken out again (bit-identical results, no gain; DESIGN.md section 5.1c, profiles/r05_wino_*): a
// PERSISTENT form -- one workgroup per CU over all its tiles, its two wave groups half a chunk apart (one transforms while the
// other multiplies), copies running on across tiles, a three-stage ring -- 56.9 against 57.1 us on layer 1, 51.8 against 48.8
// on layer 2 (the same with the older wave group multiplying first: 57.3 / 50.4); both row operands of a wave in one
// ds_read_b128; read-ahead depths 2 and 5. Per-wave stamps put an iteration of
// the chunk loop at 5190 cycles in EVERY form, against 4096 cycles of matrix-pipe time (v_mfma_f32_16x16x4_f32 issues every
// 32 cycles by itself: scripts/micro/mfma_f32_rate.hip); the operand reads, the copies and the transform each add their own
// time to a matrix-only skeleton instead of hiding under it (profiles/r05_wino_ablations_b2b.txt): SQ_VALU_MFMA_COEXEC_CYCLES is 0
// (profiles/r05_wino_pmc_coexec.txt) -- the fp32-input matrix instruction executes on the vector lanes, so every packed addition
// of the transform and every LDS return takes its cycles from the matrix pipe.
//
// Summation order: chunks ascending, two channels per instruction pair, fixed by the launch geometry alone (no atomics; where
// the launcher splits K the partial sums are added in split order by the last arriver): results are bitwise repeatable for a
// given launch geometry. The geometry -- tile shape and ksplit -- is chosen from the number of tiles, i.e. from the BATCH: an
// image's result is independent of the batch size only while no split engages (layers 1-2 of the ResNet); where it does
// (layers 3-4 at 64-128 crops) the input channels are summed in a different grouping at different batch sizes and an image's
// values move by fp32 rounding (bounded at 2e-5 by tests/test_wino.py::test_wino_split_k_results_move_by_rounding_only_across_batch_sizes).
#include "pa_kernels.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace pa {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef WN_SCALAR_T
#define WN_SCALAR_T 0   // 1: the input transform as scalar additions (A/B build; profiles/r06_wino_scalar_transform_ab.txt)
#endif
#ifndef WN_AHEAD
#define WN_AHEAD 3   // positions the row-operand reads run ahead of their matrix instructions (2 and 5 measure the same: profiles/r05_wino_operand_read_variants.txt)
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(3))) const volatile f32x2 lds_cv2;  // an LDS read hipcc may not merge with its neighbour

namespace {

// 16-byte global -> LDS DMA (buffer_load_dwordx4 ... offen lds): LDS destination = M0 + lane * 16, source = descriptor base +
// voff + soff bytes. As inline assembly, not the builtin: hipcc's wait insertion then knows nothing of these copies, and the
// kernel's own counted s_waitcnt vmcnt(N) in front of its barriers are the only waits for them. With the builtin it tracked
// them as LDS stores and, losing precision over the chunk loop's back edge, drained vmcnt to 0 in front of the first operand
// read of every third chunk -- behind copies issued a quarter of a chunk earlier.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 wn_rsrc(const void* base) {
    const unsigned long long a = (unsigned long long)base;
    return i32x4{__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff), -1, 0x00020000};
}
__device__ __forceinline__ void wn_blds16(i32x4 rsrc, int voff_bytes, int soff_bytes, float* lds_base) {
    const unsigned lds_addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)lds_base;
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :
                 : "v"(voff_bytes), "s"(rsrc), "s"(soff_bytes), "s"(lds_addr)
                 : "memory");
    // M0 is written here and cannot be declared: hipcc rejects "m0" on a clobber list ("reserved register ... undefined
    // behaviour"). The invariant instead: NOTHING else in this translation unit's kernels may live in M0 across the statement --
    // no s_movrel / v_movrel (indirect indexing of a register array: every acc / operand array here is fully unrolled), no
    // LDS-direct, no s_sendmsg with a payload, no builtin LDS-DMA (which sets M0 itself). tests/test_abi.py::
    // test_wino_object_uses_m0_only_for_its_lds_dma disassembles wino.o and fails the build check if any other use appears.
}

template <int TGN, int NCG> struct WinoGeo {
    static constexpr int TG = TGN;                          // tile groups = waves along the pixels
    static constexpr int NW = TGN * NCG;                    // waves per workgroup
    static constexpr int NSB = TG * 4;                      // 4x4-pixel sub-blocks per workgroup
    static constexpr int BN = 32 * NCG;                     // output channels per workgroup
    static constexpr int U_FLOATS = 16 * BN * 8;            // filters of one chunk
    static constexpr int U_INSTR = U_FLOATS / 256;          // wave-wide 1 KB DMA instructions per stage
    // patch stage: two planes (channels 0-3 / 4-7 of the chunk) of 16-byte pixel slots; a tile group's four 6x6 sub-blocks
    // start at slots 0 / 40 / 81 / 121 of its 160, which puts the 16 tiles of a wave on 16 distinct slots mod 16 for every
    // (dy, dx): the 32 lanes of a ds_read_b64 phase (two channel pairs of one plane) then cover all 64 banks
    static constexpr int TG_SLOTS = 160;
    static constexpr int PLANE_SLOTS = TG * TG_SLOTS;
    static constexpr int P_INSTR = 2 * PLANE_SLOTS / 64;    // 20 | 40
    static constexpr int P_FLOATS = 2 * PLANE_SLOTS * 4;
};

__device__ __forceinline__ int wn_sb_base(int s) { return s == 0 ? 0 : (s == 1 ? 40 : (s == 2 ? 81 : 121)); }

// m / d for 0 <= m < 2^24, 1 <= d < 2^16
__device__ __forceinline__ int wn_div(int m, int d) {
    int q = (int)((float)m * (1.0f / (float)d));
    int r = m - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) ++q;
    return q;
}

// one chunk: raw pixels -> V (registers), 16 positions x 2 channel quads-of-16 x 2 channels
// ABL (timing experiments, PA_WINO_ABL; results wrong when != 0): 1 no DMA behind the prologue's, 2 no matrix instructions,
// 4 no raw reads / input transform, 8 timeline stamps (results right), 16 no row-operand reads; -DPA_WINO_DIAG builds: 32 no final
// stores, 64 no residual loads
//
// One chunk: the lane's 16 raw pixels -> V (registers), then 16 positions x 2 channel groups x 2 channels of matrix
// instructions, the row operands read three positions ahead. ISSUE: the DMA instructions of a later chunk are handed out
// between the positions (issue(k), k = 0 .. KD - 1) instead of in one burst behind the barrier: eight waves issuing ~50 KB of
// LDS-DMA at once kept every wave in its address / texture queue for ~1400 cycles per chunk before its first LDS read
// (per-wave stamps, profiles/r05_wino_stamps_*.txt).
template <int GI, int ABL, int KD, bool ISSUE, typename IssueFn>
__device__ __forceinline__ void wino_chunk(const lds_f* ul, const lds_f* pl, f32x4 (&acc)[16][2], int a_off, int r_off, IssueFn issue,
                                           unsigned long long* stamp) {
    f32x2 d[4][4];
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int x = 0; x < 4; ++x)
            d[y][x] = (ABL & 4) ? f32x2{(float)r_off, 1.f} : *(lds_cv2*)(pl + r_off + (y * 6 + x) * 4);
    // row operands: three positions ahead of the matrix instructions that consume them
    // (volatile reads: hipcc would merge a pair into ds_read2st64_b64, which the LDS serves in 16-lane groups over 32 banks --
    // four-way conflicts on this image and half the rate of two ds_read_b64 even without)
    constexpr int AHEAD = WN_AHEAD;
    f32x2 a[16][2];
#define WN_LOAD_A(P)                                                       \
    {                                                                      \
        if (ABL & 16) { a[P][0] = f32x2{(float)a_off, 0.5f}; a[P][1] = f32x2{0.25f, (float)(P)}; } else {  \
        a[P][0] = *(lds_cv2*)(ul + a_off + ((P) * GI + 0) * 128);          \
        a[P][1] = *(lds_cv2*)(ul + a_off + ((P) * GI + 1) * 128);          \
        }                                                                  \
    }
#pragma unroll
    for (int p = 0; p < AHEAD; ++p) WN_LOAD_A(p);
    // the input transform: 32 additions on channel PAIRS (hipcc: v_pk_add_f32). WN_SCALAR_T=1 (A/B build)
    // issues them as 64 scalar v_add_f32 / v_sub_f32 instead -- as assembly, or hipcc pairs them up again.
#if WN_SCALAR_T
    auto add2 = [](f32x2 a_, f32x2 b_) {
        f32x2 r_;
        asm("v_add_f32 %0, %1, %2" : "=v"(r_.x) : "v"(a_.x), "v"(b_.x));
        asm("v_add_f32 %0, %1, %2" : "=v"(r_.y) : "v"(a_.y), "v"(b_.y));
        return r_;
    };
    auto sub2 = [](f32x2 a_, f32x2 b_) {
        f32x2 r_;
        asm("v_sub_f32 %0, %1, %2" : "=v"(r_.x) : "v"(a_.x), "v"(b_.x));
        asm("v_sub_f32 %0, %1, %2" : "=v"(r_.y) : "v"(a_.y), "v"(b_.y));
        return r_;
    };
#else
    auto add2 = [](f32x2 a_, f32x2 b_) { return a_ + b_; };
    auto sub2 = [](f32x2 a_, f32x2 b_) { return a_ - b_; };
#endif
    f32x2 tt[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        tt[0][x] = sub2(d[0][x], d[2][x]);
        tt[1][x] = add2(d[1][x], d[2][x]);
        tt[2][x] = sub2(d[2][x], d[1][x]);
        tt[3][x] = sub2(d[1][x], d[3][x]);
    }
    f32x2 v[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[4 * i + 0] = sub2(tt[i][0], tt[i][2]);
        v[4 * i + 1] = add2(tt[i][1], tt[i][2]);
        v[4 * i + 2] = sub2(tt[i][2], tt[i][1]);
        v[4 * i + 3] = sub2(tt[i][1], tt[i][3]);
    }
    if (ABL & 8) {  // (stamp build: when the transformed tile is complete)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 0" ::"v"(v[0].x), "v"(v[5].y), "v"(v[10].x), "v"(v[15].y));
        if (stamp) *stamp = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        __builtin_amdgcn_sched_barrier(0);
        if (p + AHEAD < 16) WN_LOAD_A(p + AHEAD);
        if (ISSUE) {
#pragma unroll
            for (int k = 0; k < KD; ++k)
                if ((k * 14) / KD == p) issue(k);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ABL & 2) {
            acc[p][0].x += a[p][0].x * v[p].x + a[p][0].y * v[p].y;
            acc[p][1].x += a[p][1].x * v[p].x + a[p][1].y * v[p].y;
            continue;
        }
        acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p][0].x, v[p].x, acc[p][0], 0, 0, 0);
        acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p][1].x, v[p].x, acc[p][1], 0, 0, 0);
        acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p][0].y, v[p].y, acc[p][0], 0, 0, 0);
        acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p][1].y, v[p].y, acc[p][1], 0, 0, 0);
    }
#undef WN_LOAD_A
}

template <int N> __device__ __forceinline__ void wn_wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is six bits");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// NST: stages of the LDS ring (3 where they fit: the DMA of chunk c + 2 is issued during chunk c, so the wait in front of a
// barrier is for copies issued a whole chunk earlier)
// (four-wave workgroups: two per CU -- without the bound hipcc spreads the kernel over 330 registers and one fits)
template <int TGN, int NCG, int NST, int ABL>
__global__ __launch_bounds__(64 * TGN * NCG, TGN * NCG == 4 ? 2 : 1) void wino3x3_kernel(const WinoParams p) {
    using G = WinoGeo<TGN, NCG>;
    constexpr int NW = G::NW;
    constexpr int KU = G::U_INSTR / NW;               // filter DMA instructions per wave and chunk
    constexpr int KPI = (G::P_INSTR + NW - 1) / NW;   // patch DMA instructions per wave and chunk
    constexpr int KD = KU + KPI;
    constexpr int D = NST - 1;                        // chunks a DMA is issued ahead of its use
    static_assert(G::U_INSTR % NW == 0 && NST >= 2 && NST <= 3, "");
    __shared__ __attribute__((aligned(16))) float u_lds0[G::U_FLOATS];
    __shared__ __attribute__((aligned(16))) float u_lds1[G::U_FLOATS];
    __shared__ __attribute__((aligned(16))) float u_lds2[NST == 3 ? G::U_FLOATS : 4];
    __shared__ __attribute__((aligned(16))) float p_lds0[G::P_FLOATS];
    __shared__ __attribute__((aligned(16))) float p_lds1[G::P_FLOATS];
    __shared__ __attribute__((aligned(16))) float p_lds2[NST == 3 ? G::P_FLOATS : 4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tg = wave % TGN;
    const int cg = wave / TGN;
    const int t = lane & 15, kq = lane >> 4;

    // XCD-aware (bijective) remap: the workgroups of an XCD are a contiguous run of (tile_m, tile_n), channel tiles fastest,
    // so the channel tiles that read one patch meet in one L2
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
    const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
    // split K (p.ksplit > 1; layers with too few tiles to fill the chip): ksplit consecutive workgroups share a tile, each sums
    // its run of n_chunks channel chunks; the partial OUTPUT tiles meet in the epilogue (the output transform is linear)
    const int ksn = p.ksplit > 1 ? p.ksplit : 1;
    int ks = wg % ksn;
    int wgt_ = wg / ksn;
    if (p.xcd_gn) {
        // The 8 XCDs as a grid of xcd_gm x xcd_gn over (pixel tiles, channel tiles): an XCD's L2 then fetches 1 / xcd_gn of the
        // filters and 1 / xcd_gm of the patches (in launch order it fetches ALL filters on a layer with few pixel tiles -- 134 MB
        // per layer-4 launch of ResNet-18 at 128 crops). The splits of a tile stay neighbours inside one XCD, channel tiles fastest.
        const int tm_per = p.tiles_m / p.xcd_gm, tn_per = p.tiles_n / p.xcd_gn, local = b >> 3;
        const int xm = xcd / p.xcd_gn, xn = xcd - xm * p.xcd_gn;
        ks = local % ksn;
        const int r = local / ksn;
        const int tm_l = r / tn_per;
        wgt_ = (xm * tm_per + tm_l) * p.tiles_n + xn * tn_per + (r - tm_l * tn_per);
    }
    const int tile_m = wgt_ / p.tiles_n, tile_n = wgt_ - tile_m * p.tiles_n;
    const int n_chunks = (p.cin >> 3) / (p.ksplit > 1 ? p.ksplit : 1);
    const int chunk0 = ks * n_chunks;

    unsigned long long* const clk = (ABL & 8) && p.clk && lane == 0 ? p.clk + ((size_t)blockIdx.x * NW + wave) * 64 : nullptr;
    if ((ABL & 8) && clk) clk[0] = __builtin_amdgcn_s_memtime();
#define WN_STAMP(K) if ((ABL & 8) && clk && (K) < 63) clk[K] = __builtin_amdgcn_s_memtime()

    const i32x4 act_rsrc = wn_rsrc(p.act), wgt_rsrc = wn_rsrc(p.wgt);
    const int u_soff0 = (tile_n * (p.cin >> 3) + chunk0) * G::U_FLOATS * 4;
    // DMA instruction K (0 .. KD - 1) of chunk C into stage (UL, PL): first the wave's share of the filter image (a straight
    // copy), then of the patch. Patch instructions past the stage's last (P_INSTR is not always a multiple of the wave count)
    // repeat the wave's previous one -- the same bytes to the same place -- so that every wave issues KD per chunk and one
    // counted wait serves all.
    int pvoff[KPI], pdst[KPI];
#define WN_ISSUE_U(UL, C, K) wn_blds16(wgt_rsrc, (((K) * NW + wave) * 64 + lane) * 16, u_soff0 + (C) * (G::U_FLOATS * 4), (UL) + ((K) * NW + wave) * 256)
#define WN_ISSUE_P(PL, C, J) wn_blds16(act_rsrc, pvoff[J], (C) * 32, (PL) + pdst[J])
#define WN_ISSUE_ALL(UL, PL, C)                                                                  \
    {                                                                                            \
        _Pragma("unroll") for (int k_ = 0; k_ < KU; ++k_) WN_ISSUE_U(UL, C, k_);                 \
        _Pragma("unroll") for (int k_ = 0; k_ < KPI; ++k_) WN_ISSUE_P(PL, C, k_);                \
    }
    // the first chunk's filters need no address work: on their way before anything else
#pragma unroll
    for (int k = 0; k < KU; ++k) WN_ISSUE_U(u_lds0, 0, k);

#pragma unroll
    for (int k = 0; k < KPI; ++k) {
        int idx = k * NW + wave;
        idx = idx < G::P_INSTR ? idx : idx - NW;
        pdst[k] = idx * 256;
        const int slot = idx * 64 + lane;
        const int h = slot >= G::PLANE_SLOTS ? 1 : 0;
        const int pi = slot - h * G::PLANE_SLOTS;
        int tgi = pi / G::TG_SLOTS;
        tgi = tgi < G::TG ? tgi : G::TG - 1;
        const int rem = pi - tgi * G::TG_SLOTS;
        const int s4 = rem >= 121 ? 3 : (rem >= 81 ? 2 : (rem >= 40 ? 1 : 0));
        int px = rem - wn_sb_base(s4);
        px = px < 36 ? px : 35;  // (pad slots: any valid pixel)
        int sb = tile_m * G::NSB + tgi * 4 + s4;
        sb = sb < p.n_sb ? sb : p.n_sb - 1;
        const int img = wn_div(sb, p.sb_per_img);
        const int rem2 = sb - img * p.sb_per_img;
        const int sby = wn_div(rem2, p.sb_per_row);
        const int sbx = rem2 - sby * p.sb_per_row;
        const int y = px / 6;
        const int x = px - y * 6;
        pvoff[k] = (img * p.in_img_stride + (4 * sby + y) * p.in_row_stride + (4 * sbx + x) * p.in_px_stride + h * 4 + chunk0 * 8) * 4;
    }
#pragma unroll
    for (int k = 0; k < KPI; ++k) WN_ISSUE_P(p_lds0, 0, k);
    if (D == 2 && n_chunks > 1) WN_ISSUE_ALL(u_lds1, p_lds1, 1);

    // operand addresses inside a stage
    const int a_off = cg * 2 * 128 + t * 8 + ((2 * kq + 4 * (t >> 3)) & 7);
    const int sbl_own = tg * 4 + (t >> 2);
    const int r_off = (kq >> 1) * (G::PLANE_SLOTS * 4) + (tg * G::TG_SLOTS + wn_sb_base(t >> 2) + ((t >> 1) & 1) * 12 + (t & 1) * 2) * 4 + (kq & 1) * 2;

    // where this lane's outputs go (its tile's 2 x 2 pixels, channels ch0 + 16 g + 0..3)
    const int sb_own = tile_m * G::NSB + sbl_own;
    const bool own = sb_own < p.n_sb;
    long o00;
    {
        const int sbc = own ? sb_own : p.n_sb - 1;
        const int img = wn_div(sbc, p.sb_per_img);
        const int rem = sbc - img * p.sb_per_img;
        const int sby = wn_div(rem, p.sb_per_row);
        const int sbx = rem - sby * p.sb_per_row;
        const int oy0 = 4 * sby + 2 * ((t >> 1) & 1), ox0 = 4 * sbx + 2 * (t & 1);
        o00 = (long)img * p.out_img_stride + (long)(oy0 + p.out_pad) * p.out_row_stride + (ox0 + p.out_pad) * p.out_px_stride +
              tile_n * G::BN + cg * 32 + 4 * kq;
    }
    const int ch0 = tile_n * G::BN + cg * 32 + 4 * kq;

    // The epilogue's bias is fetched HERE, behind the first chunk's copies (its round trip runs under theirs; the wait for the first
    // chunk, which follows, covers it) and pinned in registers for the life of the kernel: in front of the output transform it was
    // a round trip of its own per tile (every stride-1 3x3 of ResNet-18: -1.5 us). The residual is NOT fetched here: that made the
    // detector's in-place Bottlenecks 3-5 us slower (all workgroups of a round asking for it at once, in front of their first chunk);
    // it is requested in front of the LAST chunk's arithmetic (below).
    f32x4 bias4v[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        bias4v[g] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + ch0 + 16 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("" : "+v"(bias4v[g]));
    }

    f32x4 acc[16][2];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // chunk 0 landed (the copies of chunk 1, if any, stay in flight)
    if (D == 2 && n_chunks > 1) wn_wait_vmcnt<KD>(); else wn_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();

    float* const ust[3] = {u_lds0, u_lds1, u_lds2};
    float* const pst[3] = {p_lds0, p_lds1, p_lds2};
    // The residual's eight 16-byte values per lane are requested in front of the LAST chunk's arithmetic (no copy is in flight any
    // more then, so the counted waits above see nothing new) instead of behind it: a round trip less in front of the stores of a
    // kernel that has one workgroup per CU and nothing to hide it under. (Requested at the tile's start -- round 5 -- every
    // workgroup of a round asked at once, in front of its first chunk: slower.) A split tile's residual is the last arriver's alone.
    f32x4 res4[2][2][2];
    const bool res_early = p.residual && p.ksplit == 1 && !(ABL & 64);
    for (int c0 = 0; c0 < n_chunks; c0 += NST) {
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const int c = c0 + st;
            if (c >= n_chunks) break;
            WN_STAMP(1 + 3 * c);
            float* const ut = ust[(st + D) % NST];
            float* const pt = pst[(st + D) % NST];
            const bool more = c + D < n_chunks && !(ABL & 1);
            // (one copy of the chunk body with a wave-uniform branch around each DMA instruction: two copies -- with and without
            // the issue -- made hipcc spill ~400 registers at the join)
            auto issue = [&](int k) {
                if (!more) return;
                if (k < KU) { WN_ISSUE_U(ut, c + D, k); } else { WN_ISSUE_P(pt, c + D, k - KU); }
            };
            if (res_early && c == n_chunks - 1) {
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        res4[g][i >> 1][i & 1] = *reinterpret_cast<const f32x4*>(p.residual + o00 + (long)(i >> 1) * p.out_row_stride + (i & 1) * p.out_px_stride + 16 * g);
            }
            unsigned long long* const sp = (ABL & 8) && clk && 2 + 3 * c < 63 ? clk + 2 + 3 * c : nullptr;
            wino_chunk<G::BN / 16, ABL, KD, true>((const lds_f*)ust[st], (const lds_f*)pst[st], acc, a_off, r_off, issue, sp);
            WN_STAMP(3 + 3 * c);
            if (c + 1 < n_chunks) {
                // chunk c + 1 has landed for this thread (what was issued behind it stays in flight); behind the barrier it has
                // for every thread, and every wave is done with the stage the next chunk's issue overwrites
                if (D == 2 && more) wn_wait_vmcnt<KD>(); else wn_wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
            }
        }
    }
#undef WN_ISSUE_U
#undef WN_ISSUE_P
#undef WN_ISSUE_ALL

    // ---- output transform + epilogue, all in the accumulating lane ----------------------------------------------------
    // A^T m A over the 4x4 positions (p = 4 i + j): y[g][oy][ox] = the lane's 2 x 2 pixels x channels ch0 + 16 g + 0..3
    f32x4 y[2][2][2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        f32x4 s0[4], s1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s0[j] = acc[j][g] + acc[4 + j][g] + acc[8 + j][g];
            s1[j] = acc[4 + j][g] - acc[8 + j][g] - acc[12 + j][g];
        }
        y[g][0][0] = s0[0] + s0[1] + s0[2];
        y[g][0][1] = s0[1] - s0[2] - s0[3];
        y[g][1][0] = s1[0] + s1[1] + s1[2];
        y[g][1][1] = s1[1] - s1[2] - s1[3];
    }
    if (p.ksplit > 1) {
        // Every split writes its partial tile to slab[tile][split][value e][thread] (each store instruction 256 contiguous bytes)
        // with agent-scope stores (write-through: the splits of a tile may sit on different XCDs, whose L2s do not see each
        // other), waits for them, and draws a ticket; the LAST to arrive sums the ksplit partials IN SPLIT ORDER -- its own from
        // the slab as well, so the result does not depend on who was last -- and runs the epilogue.
        //
        // Why this is a sufficient publish on gfx950 although the source holds no release / acquire edge (the C++ memory model
        // alone would not promise it; this is the ISA-level hand-off MI355X_MICROARCH.md measures, "Valid forms" + the first row of
        // its table of sc1 hand-offs, and prices as the "splitk-seam" / "publish-large" rows: write-through slab stores):
        //   * EVERY store of the handed-off bytes is an agent-scope atomic store = global_store_dword ... sc1: written through the
        //     storing XCD's L2 to memory, nothing left dirty in a cache another XCD cannot see (condition 2);
        //   * every storing wave runs s_waitcnt vmcnt(0) after its stores (as inline assembly, which hipcc's wait pass cannot
        //     drop), so its bytes have LEFT the CU's memory pipe before it reaches the workgroup barrier; the barrier therefore
        //     orders the ticket add of lane 0 behind the completed stores of ALL waves of the workgroup (condition 3);
        //   * the ticket is ONE unsharded counter of agent-scope atomic adds (performed at L2 / memory, never cached): the
        //     workgroup whose add returns ksplit - 1 knows every other split's add -- and with it every other split's drained
        //     stores -- came earlier; the second barrier hands that knowledge to the other waves;
        //   * EVERY load of the slab by the last arriver is an agent-scope atomic load = global_load_dword ... sc1, which bypasses
        //     the (never refreshed) vector L1 and is served by L2 / memory; the lines were written sc1, so no XCD's L2 holds a
        //     stale copy of them (condition 1). A plain load here would be the "no acquire -> 24-50 % stale" row.
        // The relaxed order of the atomics only means the COMPILER may not be relied on to keep them in place: the two
        // __syncthreads and the asm volatile wait are its fences. tests/test_wino.py checks the split layers bitwise repeatable
        // over repeated launches under load (two lanes); a release on the add + an acquire fence in the last arriver would add a
        // buffer_wbl2 / buffer_inv pair (~3.5 us, the guide's fence rows) per workgroup for no further guarantee on this chip.
        constexpr int NT = 64 * TGN * NCG;
        float* const mine = p.slab + ((size_t)wgt_ * p.ksplit + ks) * (32 * NT) + tid;
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    __hip_atomic_store(mine + ((g * 4 + i) * 4 + e) * NT, y[g][i >> 1][i & 1][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __shared__ int last_s;
        __syncthreads();
        if (tid == 0) {
            const int ticket = __hip_atomic_fetch_add(p.tickets + wgt_, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_s = ticket == p.ksplit - 1;
            if (last_s) __hip_atomic_store(p.tickets + wgt_, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        }
        __syncthreads();
        if (!last_s) {
            if ((ABL & 8) && clk) clk[63] = __builtin_amdgcn_s_memtime();
            return;
        }
        const float* const all = p.slab + (size_t)wgt_ * p.ksplit * (32 * NT) + tid;
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) y[g][i >> 1][i & 1][e] = 0.f;
        for (int k = 0; k < p.ksplit; ++k) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        y[g][i >> 1][i & 1][e] += __hip_atomic_load(all + (size_t)k * (32 * NT) + ((g * 4 + i) * 4 + e) * NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!own) {
        if ((ABL & 8) && clk) clk[63] = __builtin_amdgcn_s_memtime();
        return;
    }
    // every residual value first, in one batch of loads: interleaved with the stores (which may alias them -- the detector's
    // Bottlenecks add in place) hipcc keeps load, wait, store, load, ... in program order, eight dependent round trips
    if (!res_early) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                res4[g][i >> 1][i & 1] = p.residual && !(ABL & 64) ? *reinterpret_cast<const f32x4*>(p.residual + o00 + (long)(i >> 1) * p.out_row_stride + (i & 1) * p.out_px_stride + 16 * g)
                                                   : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const f32x4 bias4 = bias4v[g];
#pragma unroll
        for (int oy = 0; oy < 2; ++oy)
#pragma unroll
            for (int ox = 0; ox < 2; ++ox) {
                const long o = o00 + (long)oy * p.out_row_stride + ox * p.out_px_stride + 16 * g;
                const f32x4 r4 = res4[g][oy][ox];
                f32x4 v = y[g][oy][ox] + (p.res_after ? bias4 : bias4 + r4);
                if (p.relu == 1) {
                    v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
                    v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
                } else if (p.relu == 2) {
                    v.x = silu_fast(v.x); v.y = silu_fast(v.y);
                    v.z = silu_fast(v.z); v.w = silu_fast(v.w);
                }
                if (p.res_after) v += r4;
                if ((ABL & 32) && v.x != 12345.678f) continue;
                *reinterpret_cast<f32x4*>(p.out + o) = v;
            }
    }
    if ((ABL & 8) && clk) clk[63] = __builtin_amdgcn_s_memtime();
#undef WN_STAMP
}


}  // namespace

size_t wino_weight_floats(int cin, int cout) { return (size_t)16 * cin * cout; }

// [cout][ky][kx][cin] (BatchNorm folded) -> the stage images the kernel copies: [cout / BN][cin / 8][16][BN / 16][16][8]
// with G g G^T evaluated in fp64 and rounded once
// Output channels per workgroup (= per stage image of the filter layout) for a layer of `cout` channels and `n_sb` 4x4-pixel
// sub-blocks per launch: 64 (eight waves: 4 tile groups x 2 channel groups) where that fills the chip; 32 -- four waves, two
// workgroups per CU -- where 64-channel workgroups would leave CUs idle (ResNet-18's layer 3 at 128 crops: 128 workgroups of 64
// channels take 81 us, 256 of 32 channels 65; the detector's 12 x 20 map at 64 frames has 240 and keeps 64: 83 against 121 us)
// and for layers of 32 channels. PA_WINO_BN=32|64 forces it (A/B).
int wino_pick_bn(int cout, long long n_sb, int cin_split) {
    static const int force = getenv("PA_WINO_BN") ? atoi(getenv("PA_WINO_BN")) : 0;
    if (cout % 64) return 32;
    if (force == 32 || force == 64) return force;
    long long wg64 = ((n_sb + 15) / 16) * (cout / 64);
    // a caller with split-K scratch (cin_split = the layer's input channels): the launcher multiplies the grid by up to the
    // largest power of two that leaves eight chunks per split -- 64-channel workgroups x 4 beat 32-channel ones x 4 on
    // ResNet-18's layers 3 / 4 (52-54 against 55 us, 56 against 61 us at 128 crops)
    if (cin_split > 0)
        for (int k2 = 2; k2 <= 8 && (cin_split / 8) % k2 == 0 && (cin_split / 8) / k2 >= 8 && wg64 * 2 <= 256; k2 *= 2) wg64 *= 2;
    return wg64 < 192 ? 32 : 64;
}

void wino_transform_weights(const float* w, int cin, int cout, int bn, float* ug) {
    const int gi_n = bn / 16, n_chunks = cin / 8;
    static const double Gm[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            double g[3][3], tmp[4][3], u[4][4];
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) g[ky][kx] = (double)w[((size_t)co * 9 + ky * 3 + kx) * cin + ci];
            for (int i = 0; i < 4; ++i)
                for (int kx = 0; kx < 3; ++kx) tmp[i][kx] = Gm[i][0] * g[0][kx] + Gm[i][1] * g[1][kx] + Gm[i][2] * g[2][kx];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) u[i][j] = tmp[i][0] * Gm[j][0] + tmp[i][1] * Gm[j][1] + tmp[i][2] * Gm[j][2];
            const int tile_n = co / bn, gi = (co % bn) / 16, r = co % 16, c = ci / 8, cc = ci % 8;
            for (int pp = 0; pp < 16; ++pp) {
                const size_t idx = ((((size_t)tile_n * n_chunks + c) * 16 + pp) * gi_n + gi) * 128 + r * 8 + ((cc + 4 * (r >> 3)) & 7);
                ug[idx] = (float)u[pp >> 2][pp & 3];
            }
        }
}

hipError_t launch_wino3x3(const WinoParams& p_in, hipStream_t s) {
    WinoParams p = p_in;
    if (!p.act || !p.wgt || !p.out || p.cin < 8 || p.cin % 8 || p.cout < 32 || p.cout % 32 || p.height < 4 || p.width < 4 || p.height % 4 ||
        p.width % 4 || p.n_img < 1 || p.in_px_stride < p.cin || p.out_px_stride < p.cout)
        return hipErrorInvalidValue;
    // every access is 16 bytes wide (LDS-DMA of four channels, dwordx4 stores / residual / bias loads): pixel pitches in whole
    // groups of four floats, every base 16-byte aligned
    auto mis = [](const void* q) { return (reinterpret_cast<unsigned long long>(q) & 15ull) != 0; };
    if (p.in_px_stride % 4 || p.out_px_stride % 4 || p.in_row_stride % 4 || p.in_img_stride % 4 || p.out_row_stride % 4 || p.out_img_stride % 4 ||
        mis(p.act) || mis(p.wgt) || mis(p.out) || (p.bias && mis(p.bias)) || (p.residual && mis(p.residual)))
        return hipErrorInvalidValue;
    p.sb_per_row = p.width / 4;
    p.sb_per_img = (p.height / 4) * p.sb_per_row;
    const long long n_sb = (long long)p.n_img * p.sb_per_img;
    if (n_sb >= (1 << 24) || p.sb_per_img >= (1 << 16)) return hipErrorInvalidValue;
    // byte offsets inside the buffer descriptors are 32-bit
    if ((long long)p.n_img * p.in_img_stride * 4 >= (1ll << 31) || (long long)wino_weight_floats(p.cin, p.cout) * 4 >= (1ll << 31)) return hipErrorInvalidValue;
    p.n_sb = (int)n_sb;
    const int bn = p.bn;
    if ((bn != 32 && bn != 64) || p.cout % bn) return hipErrorInvalidValue;
    // 32-channel workgroups are four waves (four tile groups), two to a CU: the detector's 32-channel Bottleneck (four chunks per
    // tile, its prologue and epilogue as long as its chunks) 154 -> 138 us against the eight-wave form, one to a CU
    // (PA_WINO_SMALL32=0: that form, A/B)
    static const int small32 = getenv("PA_WINO_SMALL32") ? atoi(getenv("PA_WINO_SMALL32")) : 1;
    const bool small_wg = bn == 32 && (p.cout % 64 == 0 || small32);
    const int nsb = bn == 64 ? 16 : (small_wg ? 16 : 32);
    p.tiles_n = p.cout / bn;
    const int tiles_m = (p.n_sb + nsb - 1) / nsb;
    // split K where the tiles alone leave CUs without a workgroup (ResNet-18's layers 3 and 4 at 128 crops: 256 four-wave / 64
    // eight-wave workgroups): the largest power of two that keeps >= 8 chunks per split and the grid within one workgroup of
    // eight waves (two of four) per CU; needs the caller's scratch (slab, tickets). PA_WINO_KS=1 turns it off, n forces n.
    static const int ks_force = getenv("PA_WINO_KS") ? atoi(getenv("PA_WINO_KS")) : 0;
    const int n_tiles = tiles_m * p.tiles_n, threads = (bn == 64 || !small_wg) ? 512 : 256;
    int ks = 1;
    if (p.slab && p.tickets && n_tiles <= p.tickets_cap) {
        const int target = threads == 512 ? 256 : 512;
        for (int k2 = 2; k2 <= 8; k2 *= 2)
            if ((p.cin / 8) % k2 == 0 && (p.cin / 8) / k2 >= 8 && n_tiles * k2 <= target && (size_t)n_tiles * k2 * 32 * threads <= p.slab_floats &&
                (ks_force == 0 || k2 <= ks_force))
                ks = k2;
        if (ks_force == 1) ks = 1;
    }
    p.ksplit = ks;
    const int grid = n_tiles * ks;
    // XCD grid (see the kernel): the (gm, gn) with gm x gn = 8 that divides the tile counts and fetches least -- filters x gm +
    // patches x gn -- if that is under 0.8 of what launch order fetches (filters x min(8, tiles_m ...) is its worst case: all
    // eight). PA_WINO_XCD_GRID=0: launch order always (A/B).
    static const int xcd_grid = getenv("PA_WINO_XCD_GRID") ? atoi(getenv("PA_WINO_XCD_GRID")) : 1;
    p.tiles_m = tiles_m;
    p.xcd_gm = p.xcd_gn = 0;
    if (xcd_grid && grid % 8 == 0) {
        const double wb = (double)wino_weight_floats(p.cin, p.cout) * 4.0, pb = (double)p.n_sb * 36.0 * p.cin * 4.0;
        // launch order: an XCD holds grid / 8 consecutive (tile, split) pairs = n_tiles / 8 consecutive tiles, channel tiles fastest
        const double per_xcd_tiles = n_tiles / 8.0;
        const double order_w = wb * 8.0 * std::min(1.0, per_xcd_tiles / p.tiles_n), order_p = pb * std::max(1.0, p.tiles_n / per_xcd_tiles);
        double best = 0.8 * (order_w + order_p);
        for (int gm = 1; gm <= 8; gm *= 2) {
            const int gn = 8 / gm;
            if (tiles_m % gm || p.tiles_n % gn) continue;
            const double t = wb * gm + pb * gn;
            if (t < best) { best = t; p.xcd_gm = gm; p.xcd_gn = gn; }
        }
    }
    static const int abl = getenv("PA_WINO_ABL") ? atoi(getenv("PA_WINO_ABL")) : 0;
    static const int nst = getenv("PA_WINO_STAGES") ? atoi(getenv("PA_WINO_STAGES")) : 2;  // 3: a three-stage ring (A/B: 62.3 against 61.5 us on layer 1, 95.2 against 92.1 on the 24 x 40 map -- the prologue then waits behind two chunks of copies)
#define WN_LAUNCH(ABL_)                                                                                        \
    if (bn == 64 && nst == 3) hipLaunchKernelGGL((wino3x3_kernel<4, 2, 3, ABL_>), dim3(grid), dim3(512), 0, s, p);    \
    else if (bn == 64) hipLaunchKernelGGL((wino3x3_kernel<4, 2, 2, ABL_>), dim3(grid), dim3(512), 0, s, p);           \
    else if (small_wg) hipLaunchKernelGGL((wino3x3_kernel<4, 1, 2, ABL_>), dim3(grid), dim3(256), 0, s, p);           \
    else hipLaunchKernelGGL((wino3x3_kernel<8, 1, 2, ABL_>), dim3(grid), dim3(512), 0, s, p)
    switch (abl) {
        case 1: WN_LAUNCH(1); break;
        case 2: WN_LAUNCH(2); break;
        case 3: WN_LAUNCH(3); break;
        case 7: WN_LAUNCH(7); break;
        case 16: WN_LAUNCH(16); break;
        case 17: WN_LAUNCH(17); break;
        case 20: WN_LAUNCH(20); break;
        case 21: WN_LAUNCH(21); break;
#ifdef PA_WINO_DIAG   // (diagnostic build only: 32 = no final stores, 64 = no residual loads)
        case 32: WN_LAUNCH(32); break;
        case 64: WN_LAUNCH(64); break;
        case 96: WN_LAUNCH(96); break;
#endif
        case 8: {
            // timeline stamps of one launch (the PA_WINO_STAMP_CALL-th) -> PA_WINO_STAMP_FILE: header {grid, waves, cin, n_sb}, then
            // [grid][waves][64] s_memtime values (scripts/wino_stamps.py)
            static int calls = 0;
            static unsigned long long* dev = nullptr;
            const char* sf = getenv("PA_WINO_STAMP_FILE");
            const int nw = bn == 64 ? 8 : (small_wg ? 4 : 8);
            const bool now = sf && calls++ == (getenv("PA_WINO_STAMP_CALL") ? atoi(getenv("PA_WINO_STAMP_CALL")) : 0);
            const size_t words = (size_t)grid * nw * 64;
            if (now) {
                if (dev) (void)hipFree(dev);
                if (hipMalloc(&dev, words * 8) != hipSuccess) return hipErrorOutOfMemory;
                (void)hipMemset(dev, 0, words * 8);
                p.clk = dev;
            }
            WN_LAUNCH(8);
            if (now) {
                (void)hipStreamSynchronize(s);
                std::vector<unsigned long long> hst(words);
                (void)hipMemcpy(hst.data(), dev, words * 8, hipMemcpyDeviceToHost);
                if (FILE* f = fopen(sf, "wb")) {
                    const int hdr[4] = {grid, nw, p.cin, p.n_sb};
                    fwrite(hdr, 4, 4, f);
                    fwrite(hst.data(), 8, words, f);
                    fclose(f);
                }
            }
            break;
        }
        default: WN_LAUNCH(0);
    }
#undef WN_LAUNCH
    return hipGetLastError();
}

}  // namespace pa

// ---- C ABI: the kernel as a single-layer operator (include/playaid_hip.h) --------------------------------------------------
#include "../../include/playaid_hip.h"

extern "C" {

size_t pa_wino_weight_floats(int32_t cin, int32_t cout) {
    if (cin < 8 || cin % 8 || cout < 32 || cout % 32) return 0;
    return pa::wino_weight_floats(cin, cout);
}

int pa_wino_channels_per_workgroup(int32_t cout, int64_t sub_blocks) {
    if (cout < 32 || cout % 32 || sub_blocks < 1) return 0;
    return pa::wino_pick_bn(cout, sub_blocks, 0);
}

int pa_wino_transform_weights(const float* w_host, int32_t cin, int32_t cout, int32_t bn, float* ug_host) {
    if (!w_host || !ug_host || cin < 8 || cin % 8 || cout < 32 || cout % 32 || (bn != 32 && bn != 64) || cout % bn) return PA_ERR_INVALID_ARG;
    pa::wino_transform_weights(w_host, cin, cout, bn, ug_host);
    return PA_OK;
}

static int wino_conv3x3_impl(const float* x, const float* ug, const float* bias, const float* residual, float* out, int32_t n, int32_t height,
                             int32_t width, int32_t cin, int32_t cout, int32_t bn, int32_t in_px_stride, int32_t out_px_stride, int32_t out_pad,
                             int32_t act, int32_t res_after, float* slab, size_t slab_floats, int32_t* tickets, int32_t n_tickets, void* stream) {
    if (!x || !ug || !out || n < 1 || out_pad < 0 || act < 0 || act > 2) return PA_ERR_INVALID_ARG;
    pa::WinoParams p{};
    p.slab = slab; p.slab_floats = slab_floats; p.tickets = tickets; p.tickets_cap = n_tickets;
    p.act = x; p.wgt = ug; p.bias = bias; p.residual = residual; p.out = out;
    p.n_img = n; p.height = height; p.width = width; p.cin = cin; p.cout = cout; p.bn = bn;
    p.in_px_stride = in_px_stride;
    p.in_row_stride = (width + 2) * in_px_stride;
    p.in_img_stride = (height + 2) * p.in_row_stride;
    p.out_px_stride = out_px_stride;
    p.out_row_stride = (width + 2 * out_pad) * out_px_stride;
    p.out_img_stride = (height + 2 * out_pad) * p.out_row_stride;
    p.out_pad = out_pad;
    p.relu = act;
    p.res_after = res_after;
    const hipError_t e = pa::launch_wino3x3(p, (hipStream_t)stream);
    return e == hipSuccess ? PA_OK : (e == hipErrorInvalidValue ? PA_ERR_INVALID_ARG : PA_ERR_HIP);
}

int pa_wino_conv3x3(const float* x, const float* ug, const float* bias, const float* residual, float* out, int32_t n, int32_t height,
                    int32_t width, int32_t cin, int32_t cout, int32_t bn, int32_t in_px_stride, int32_t out_px_stride, int32_t out_pad, int32_t act,
                    int32_t res_after, void* stream) {
    return wino_conv3x3_impl(x, ug, bias, residual, out, n, height, width, cin, cout, bn, in_px_stride, out_px_stride, out_pad, act, res_after, nullptr, 0,
                             nullptr, 0, stream);
}

int pa_wino_conv3x3_splitk(const float* x, const float* ug, const float* bias, const float* residual, float* out, int32_t n, int32_t height,
                           int32_t width, int32_t cin, int32_t cout, int32_t bn, int32_t in_px_stride, int32_t out_px_stride, int32_t out_pad,
                           int32_t act, int32_t res_after, float* slab, size_t slab_floats, int32_t* tickets, int32_t n_tickets, void* stream) {
    if (!slab || !tickets || n_tickets < 1 || (reinterpret_cast<unsigned long long>(slab) & 15ull)) return PA_ERR_INVALID_ARG;
    return wino_conv3x3_impl(x, ug, bias, residual, out, n, height, width, cin, cout, bn, in_px_stride, out_px_stride, out_pad, act, res_after, slab,
                             slab_floats, tickets, n_tickets, stream);
}

}  // extern "C"
