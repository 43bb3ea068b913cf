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
// Same k order as igemm.hip (tap, channel chunk, eight-wide group, lane half); the bias is the value the accumulators start
// from instead of a last addition, so results agree with igemm.hip to the rounding of that one reordering.
#include "pa_kernels.h"

#include <algorithm>
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace pa {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;

// 16 bytes per lane, HBM/L2 -> LDS: voff = the lane's byte offset (a register that lives as long as the tile), soff = the
// k-step's byte offset (scalar): no vector instruction between two copies
__device__ __forceinline__ void pg_dma16(__amdgpu_buffer_rsrc_t rsrc, int voff_bytes, int soff_bytes, float* lds_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_base, 16, voff_bytes, soff_bytes, 0, 0);
}

// n / d and the remainder for a WAVE-UNIFORM 0 <= n < 2^25 with magic = min(ceil(2^32 / d), 2^32 - 1), 1 <= d < 2^16: the
// estimate is off by at most one either way (n * (magic * d - 2^32) < 2^32 * d * 2^-7); everything on the scalar unit
__device__ __forceinline__ int pg_sdiv(int n, int d, unsigned magic, int& rem) {
    int q = (int)__umulhi((unsigned)n, magic);
    int r = n - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; r -= d; }
    rem = r;
    return q;
}

// (vmcnt is six bits: a count past 63 waits at 63 -- for more than it must, never for less)
template <int N> __device__ __forceinline__ void pg_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N > 63 ? 63 : N) : "memory"); }

}  // namespace

// BM x BN tiles, 256 threads = 2 x 2 waves (BN = 64) or 4 x 1 (BN = 32: layers with 32 output channels run without zero
// padding), three LDS stages of (BM + BN) rows x 32 floats.
//
// What a tile costs besides its matrix instructions is vector-ALU time -- scripts/micro/mfma_f32_mix.hip: next to a stream of
// v_mfma_f32_32x32x2_f32 an LDS read is free, a lone vector instruction costs ~12 cycles of the SIMD, a transcendental 16 (and
// v_mul_lo_u32, which the per-lane divisions were made of, is a quarter-rate instruction too) -- so the bookkeeping is kept off
// the vector unit:
//   * pixel index -> (image, row, column) of the tile's FIRST pixel on the scalar unit (pg_sdiv); a lane adds its pixel's
//     distance and folds the row / image wraps in with compares and selects (no multiply, no division per lane);
//   * a k-step's copies differ from the tile's first in a scalar byte offset only (buffer_load's soffset);
//   * the LDS read addresses of a k-step are eight adds in one burst (a burst costs about what a lone instruction does);
//   * SiLU on register pairs (v_pk_mul / v_pk_add around the two transcendentals).
// UP: the tile is ALSO written nearest-neighbour up-sampled by two into a second buffer (p.up_out: every output pixel to the 2 x 2
// pixels it becomes) -- YOLOv5's nn.Upsample behind model.10 / model.14 as four more stores of the producer instead of a pass
// of its own over HBM.
template <int BM, int BN = 64, bool SILU = true, bool STAMP = false, bool UP = false>
__global__ __launch_bounds__(256, 2) void pgemm_kernel(const GemmParams p) {
    // STAMP (scripts/pgemm_stamps.py): s_memtime per wave at entry [0], after the prologue [1], per tile t < 15 at 2 + 4 t:
    // tile start, first barrier passed, last matrix instruction issued, stores issued; exit [63]
    auto stamp = [&](int i) {
        if (STAMP) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if ((threadIdx.x & 63) == 0) p.clk[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + i] = t;
        }
    };
    stamp(0);
    constexpr int WM = BN == 64 ? 2 : 4;      // waves along the pixel rows
    constexpr int MI = BM / WM / 32;
    constexpr int A_ROWS = BM / 32, B_ROWS = BN / 32;
    constexpr int NLD = A_ROWS + B_ROWS;  // LDS-DMA wave instructions per k-step
    constexpr int NST = MI * 4 * (UP ? 5 : 1);   // store wave instructions per tile
    constexpr int STAGE = (BM + BN) * 32;
    __shared__ __attribute__((aligned(16))) float lds[3 * STAGE];

    // --- this workgroup's tiles: one channel column, every lm-th pixel tile of its XCD's contiguous share -------------
    const int b = blockIdx.x, xcd = b & 7, local = b >> 3, per = p.pg_per;
    const int TN = p.tiles_n, TM = p.tiles_m;
    const int LM = per / TN;  // workgroups per XCD and channel column (the launcher makes per a multiple of TN)
    const int tile_n = local % TN, lm = local / TN;
    const int t_lo = (int)(((long long)xcd * TM) >> 3), t_hi = (int)(((long long)(xcd + 1) * TM) >> 3);
    const int nt = t_lo + lm < t_hi ? (t_hi - t_lo - lm + LM - 1) / LM : 0;
    {   // the arguments the loop needs are asked for BEFORE the early exit below: one round trip of scalar loads instead of two
        const float *pa_ = p.act, *pw_ = p.wgt, *pb_ = p.bias;
        float* po_ = p.out;
        const int i0 = p.M, i1 = p.ktot, i2 = p.kw_taps, i3 = p.chunk, i4 = p.howo, i5 = p.wo, i6 = p.in_img_stride, i7 = p.in_row_stride,
                  i8 = p.in_px_stride, i9 = p.stride, i10 = p.off_y, i11 = p.off_x, i12 = p.out_img_stride, i13 = p.out_row_stride,
                  i14 = p.out_px_stride, i15 = p.out_pad, i16 = p.pg_ho, i17 = p.pg_nwx, i18 = p.pg_nwy;
        const unsigned u0 = p.pg_magic_howo, u1 = p.pg_magic_wo;
        asm volatile("" ::"s"(pa_), "s"(pw_), "s"(pb_), "s"(po_), "s"(i0), "s"(i1), "s"(i2), "s"(i3), "s"(i4), "s"(i5), "s"(i6), "s"(i7), "s"(i8), "s"(i9),
                     "s"(i10), "s"(i11), "s"(i12), "s"(i13), "s"(i14), "s"(i15), "s"(i16), "s"(i17), "s"(i18), "s"(u0), "s"(u1));
    }
    if (nt == 0) return;
    stamp(60);
    const int nk = p.ktot >> 5;  // k-steps per tile
    const int total = nt * nk;

    const int tid = threadIdx.x;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = tid >> 3;
    const int colq = (tid & 7) ^ ((row0 >> 1) & 7);  // LDS chunk c of row r holds logical chunk c ^ ((r >> 1) & 7)
    const int lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int wm = BN == 64 ? wave_id >> 1 : wave_id, wn = BN == 64 ? wave_id & 1 : 0;

    const __amdgpu_buffer_rsrc_t act_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.act), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t wgt_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wgt), 0, -1, 0x00020000);

    const int ch0 = tile_n * BN + wn * 32 + 4 * lh;
    f32x2 nl2e = f32x2{-1.44269504088896341f, -1.44269504088896341f}, one2 = f32x2{1.f, 1.f};
    asm volatile("" : "+v"(nl2e), "+v"(one2));   // (register pairs: v_pk_mul / v_pk_add take no literal)

    int b_off[B_ROWS];   // bytes
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) b_off[i] = ((tile_n * BN + row0 + 32 * i) * p.ktot + colq * 4) * 4;

    // --- pixel addressing: scalar base of a 32-pixel run + the lane's distance, wraps folded in -------------------------
    // input, in bytes: A(m) = img * IS + (oy * s + off_y) * RS + (ox * s + off_x) * PS
    const int in_ps = p.in_px_stride * p.stride * 4, in_rs = p.in_row_stride * p.stride * 4;
    const int in_wrap_x = in_rs - p.wo * in_ps;                          // column wo -> column 0 of the next row
    const int in_wrap_y = p.in_img_stride * 4 - p.pg_ho * in_rs;         // row ho -> row 0 of the next image
    const int in_org = (p.off_y * p.in_row_stride + p.off_x * p.in_px_stride) * 4;
    int in_lane = row0 * in_ps + colq * 16;
    asm volatile("" : "+v"(in_lane));   // (kept as a register: the compiler would fold it back into a per-tile multiply)
    int in_last;   // pixel M - 1: what the rows past M of a partial last tile read (computed, dropped)
    {
        int rem, ox;
        const int img = pg_sdiv(p.M - 1, p.howo, p.pg_magic_howo, rem);
        const int oy = pg_sdiv(rem, p.wo, p.pg_magic_wo, ox);
        in_last = img * (p.in_img_stride * 4) + oy * in_rs + ox * in_ps + in_org + colq * 16;
    }
    const int nwx = p.pg_nwx, nwy = p.pg_nwy;   // most row wraps over 31 pixels, most image wraps over that many rows
    auto in_offset = [&](int m_base) {   // byte offset of this lane's row of the 32-pixel run starting at the uniform m_base
        int rem, ox_b;
        const int img_b = pg_sdiv(m_base, p.howo, p.pg_magic_howo, rem);
        int oy = pg_sdiv(rem, p.wo, p.pg_magic_wo, ox_b);
        int off = img_b * (p.in_img_stride * 4) + oy * in_rs + ox_b * in_ps + in_org + in_lane;
        int ox = ox_b + row0;
        for (int w = 0; w < nwx; ++w) {
            const bool c = ox >= p.wo;
            ox -= c ? p.wo : 0;
            off += c ? in_wrap_x : 0;
            oy += c ? 1 : 0;
        }
        for (int w = 0; w < nwy; ++w) {
            const bool c = oy >= p.pg_ho;
            oy -= c ? p.pg_ho : 0;
            off += c ? in_wrap_y : 0;
        }
        return m_base + row0 < p.M ? off : in_last;
    };
    // output, in floats: O(m) = img * OIS + (oy + pad) * ORS + (ox + pad) * OPS + ch0
    const int out_wrap_x = p.out_row_stride - p.wo * p.out_px_stride;
    const int out_wrap_y = p.out_img_stride - p.pg_ho * p.out_row_stride;
    int out_lane = lr * p.out_px_stride + p.out_pad * (p.out_row_stride + p.out_px_stride) + ch0;
    asm volatile("" : "+v"(out_lane));
    // offset of this lane's pixel of the 32-pixel run at m_base in a buffer whose pixel (img, oy, ox) sits at img * is + oy * rs + ox * ps
    auto pix_offset = [&](int m_base, int is, int rs, int ps, int wrap_x, int wrap_y, int lane_const) {
        int rem, ox_b;
        const int img_b = pg_sdiv(m_base, p.howo, p.pg_magic_howo, rem);
        int oy = pg_sdiv(rem, p.wo, p.pg_magic_wo, ox_b);
        int off = img_b * is + oy * rs + ox_b * ps + lane_const;
        int ox = ox_b + lr;
        for (int w = 0; w < nwx; ++w) {
            const bool c = ox >= p.wo;
            ox -= c ? p.wo : 0;
            off += c ? wrap_x : 0;
            oy += c ? 1 : 0;
        }
        for (int w = 0; w < nwy; ++w) {
            const bool c = oy >= p.pg_ho;
            oy -= c ? p.pg_ho : 0;
            off += c ? wrap_y : 0;
        }
        return off;
    };
    auto out_offset = [&](int m_base) { return pix_offset(m_base, p.out_img_stride, p.out_row_stride, p.out_px_stride, out_wrap_x, out_wrap_y, out_lane); };
    // the up-sampled copy: pixel (oy, ox) -> (2 oy, 2 ox) .. (2 oy + 1, 2 ox + 1): the same walk with doubled row and pixel strides
    const int up_rs = 2 * p.up_row_stride, up_ps = 2 * p.up_px_stride;
    const int up_wrap_x = up_rs - p.wo * up_ps, up_wrap_y = p.up_img_stride - p.pg_ho * up_rs;
    int up_lane = lr * up_ps + p.up_pad * (p.up_row_stride + p.up_px_stride) + ch0;
    asm volatile("" : "+v"(up_lane));
    // issue cursor: tile, its row offsets, (ky, kx, kc) of its next k-step
    int i_tile = t_lo + lm, i_ks = 0, i_ky = 0, i_kx = 0, i_kc = 0;
    int a_off[A_ROWS];   // bytes
    auto rows_of = [&](int tile_m) {
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) a_off[i] = in_offset(tile_m * BM + 32 * i);
    };
    rows_of(i_tile);
    stamp(61);
    auto issue = [&](int slot) {
        float* As_w = lds + slot * STAGE + wave_id * 256;
        float* Bs_w = As_w + BM * 32;
        const int tapoff = (i_ky * p.in_row_stride + i_kx * p.in_px_stride + i_kc) * 4;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) pg_dma16(act_rs, a_off[i], tapoff, As_w + i * 1024);
        const int koff = ((i_ky * p.kw_taps + i_kx) * p.chunk + i_kc) * 4;
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) pg_dma16(wgt_rs, b_off[i], koff, Bs_w + i * 1024);
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

    // LDS read addresses of the four eight-wide k groups within a stage (bytes; the 32-row block is an immediate of the read)
    unsigned a_k[4], b_k[4];
    {
        const int a_rd = (wm * (BM / WM) + lr) * 32, b_rd = BM * 32 + (wn * 32 + lr) * 32;
        const int swz = (lr >> 1) & 7;
        const unsigned base = (unsigned)(size_t)(lds_f32*)lds;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ch_ = ((kk * 2 + lh) ^ swz) * 4;
            a_k[kk] = base + (a_rd + ch_) * 4;
            b_k[kk] = base + (b_rd + ch_) * 4;
            asm volatile("" : "+v"(a_k[kk]), "+v"(b_k[kk]));
        }
    }

    issue(0);
    stamp(62);
    if (total > 1) issue(1);
    // bias of this lane's channels ch0 + 8 g + 0..3, the value every accumulator of a tile starts from: fetched BEHIND the first
    // copies (its round trip runs under theirs; the kernel's first 4 us used to be this load alone) and pinned, so that the
    // compiler's wait for it -- a full drain, as for any register load beside LDS-DMA -- happens here, where the first stage
    // is awaited anyway, and not in the loop
    f32x16 biasv;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + ch0 + 8 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
        biasv[4 * g] = b4.x; biasv[4 * g + 1] = b4.y; biasv[4 * g + 2] = b4.z; biasv[4 * g + 3] = b4.w;
    }
    asm volatile("" : "+v"(biasv));
    stamp(1);

    f32x16 acc[MI];
    // one k-step on the stage at byte offset sb: 16 matrix instructions per 32-pixel block, operands read one k group
    // ahead; the eight read addresses in one burst of vector adds before the first of them
    auto kstep = [&](int sb, auto first_c) {
        constexpr bool FIRST = decltype(first_c)::value;   // a tile's first k-step: the accumulators start from the bias
        unsigned ra[4], rb[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            ra[kk] = a_k[kk] + sb;
            rb[kk] = b_k[kk] + sb;
            asm volatile("" : "+v"(ra[kk]), "+v"(rb[kk]));
        }
        f32x4 af[2][MI], bf[2];
#define PG_FRAGS(SET, KK)                                                                                    \
    {                                                                                                        \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                    \
            af[SET][mi] = *(const lds_f32x4*)(size_t)(ra[KK] + mi * 4096);                                   \
        bf[SET] = *(const lds_f32x4*)(size_t)(rb[KK]);                                                       \
    }
        PG_FRAGS(0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const f32x4 a4 = af[kk & 1][mi], b4 = bf[kk & 1];
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.x, a4.x, FIRST && kk == 0 ? biasv : acc[mi], 0, 0, 0);
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.y, a4.y, acc[mi], 0, 0, 0);
                if (mi == 0 && kk + 1 < 4) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (kk == 0) { PG_FRAGS(1, 1); } else if (kk == 1) { PG_FRAGS(0, 2); } else { PG_FRAGS(1, 3); }
                    __builtin_amdgcn_sched_barrier(0);
                }
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.z, a4.z, acc[mi], 0, 0, 0);
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.w, a4.w, acc[mi], 0, 0, 0);
            }
        }
#undef PG_FRAGS
    };

    int slot = 0, g = 0;
    // what precedes the matrix instructions of k-step g (the tile's ks-th): wait for its stage, barrier, request stage g + 2
    auto pre = [&](int ks, int t) {
        // Stage g must have landed. VMEM operations complete in order and this wave issued, after the copies of stage
        // g (during step g - 2): the stores of the tile step g - 2 closed, the copies of stage g + 1, the stores of the
        // tile step g - 1 closed -- whichever of those exist. Exactly that many may still be outstanding.
        const bool next = g + 1 < total;
        int closed = 0;
        if (ks == 0 && g > 0) closed = 1;
        if ((nk == 1 || ks == 1) && g > 1) ++closed;
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
        if (STAMP && ks == 0 && t < 15) stamp(3 + 4 * t);
        const int slot2 = slot == 0 ? 2 : slot - 1;   // (g + 2) % 3 == (g - 1) % 3
        if (g + 2 < total) issue(slot2);
        __builtin_amdgcn_sched_barrier(0);
    };
    // (a wave's LDS reads of a stage are complete before it reaches the next barrier, behind which the stage is overwritten:
    // every one of them fed a matrix instruction)
    for (int t = 0; t < nt; ++t) {
        const int tile_m = t_lo + lm + t * LM;
        if (t < 15) stamp(2 + 4 * t);
        pre(0, t);
        kstep(slot * (STAGE * 4), std::true_type{});
        slot = slot == 2 ? 0 : slot + 1;
        ++g;
        for (int ks = 1; ks < nk; ++ks, ++g) {
            pre(ks, t);
            kstep(slot * (STAGE * 4), std::false_type{});
            slot = slot == 2 ? 0 : slot + 1;
        }
        if (t < 15) stamp(4 + 4 * t);
        // epilogue of the tile, straight from the accumulators: lane = pixel lr of its 32-pixel block, channels ch0 + 8 g + 0..3
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m_base = tile_m * BM + wm * (BM / WM) + mi * 32;
            float* o_px = p.out + out_offset(m_base);
            const bool live = m_base + lr < p.M;
            f32x4 out4[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                f32x2 lo = f32x2{acc[mi][4 * gq], acc[mi][4 * gq + 1]}, hi = f32x2{acc[mi][4 * gq + 2], acc[mi][4 * gq + 3]};   // (bias inside)
                if (SILU) {   // x * 1 / (1 + 2^(-x log2 e)), as silu_fast
                    const f32x2 tl = lo * nl2e, th = hi * nl2e;
                    const f32x2 dl = f32x2{__builtin_amdgcn_exp2f(tl.x), __builtin_amdgcn_exp2f(tl.y)} + one2;
                    const f32x2 dh = f32x2{__builtin_amdgcn_exp2f(th.x), __builtin_amdgcn_exp2f(th.y)} + one2;
                    lo *= f32x2{__builtin_amdgcn_rcpf(dl.x), __builtin_amdgcn_rcpf(dl.y)};
                    hi *= f32x2{__builtin_amdgcn_rcpf(dh.x), __builtin_amdgcn_rcpf(dh.y)};
                } else if (p.relu == 1) {
                    lo.x = lo.x > 0.f ? lo.x : 0.f; lo.y = lo.y > 0.f ? lo.y : 0.f;
                    hi.x = hi.x > 0.f ? hi.x : 0.f; hi.y = hi.y > 0.f ? hi.y : 0.f;
                }
                // (rows past M of a partial last tile were computed on pixel M - 1 and are dropped)
                // INVARIANT the counted waits rely on: a wave whose rows are all past M issues NO stores here (hipcc branches
                // around them), so its vmcnt(NLD + k * NST) would under-wait if another k-step of this workgroup followed. None
                // does: a workgroup walks its tiles in ASCENDING order (t_lo + lm, + LM, ...) and only the launch's last tile
                // (tiles_m - 1) can be partial, so a partial tile is always the last thing its workgroup computes. Any other
                // tile order must first make the store count independent of the predicate (launch_pgemm checks the premise). That the
                // compiler emits exactly NST 16-byte stores per tile is checked on the built object
                // (tests/test_abi.py::test_pgemm_store_count_matches_the_counted_waits); psgemm.hip, the emulated-fp32 form of this
                // kernel, issues its stores unconditionally through a bounds-checked buffer descriptor instead.
                if (live) *reinterpret_cast<f32x4*>(o_px + 8 * gq) = f32x4{lo.x, lo.y, hi.x, hi.y};
                if (UP) { out4[gq] = f32x4{lo.x, lo.y, hi.x, hi.y}; }
            }
            if (UP && live) {   // (a launch with an up-sampled copy has no partial tile followed by a k-step either: same invariant)
                float* u_px = p.up_out + pix_offset(m_base, p.up_img_stride, up_rs, up_ps, up_wrap_x, up_wrap_y, up_lane);
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    float* u = u_px + (q4 >> 1) * p.up_row_stride + (q4 & 1) * p.up_px_stride;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) *reinterpret_cast<f32x4*>(u + 8 * gq) = out4[gq];
                }
            }
        }
        if (t < 15) stamp(5 + 4 * t);
    }
    stamp(63);
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
    // the kernel's pixel arithmetic (pg_sdiv and the wrap loops)
    auto magic = [](int d) { return (unsigned)std::min<unsigned long long>(((1ull << 32) + d - 1) / d, 0xffffffffull); };
    if (p.howo % p.wo != 0 || p.howo >= (1 << 16)) return hipErrorInvalidValue;
    p.pg_per = per;
    p.pg_ho = p.howo / p.wo;
    p.pg_magic_howo = magic(p.howo);
    p.pg_magic_wo = magic(p.wo);
    p.pg_nwx = 1 + 30 / p.wo;                         // column c + 31 <= wo - 1 + 31 wraps at most this often
    p.pg_nwy = (p.pg_ho - 1 + p.pg_nwx) / p.pg_ho;    // and row r + nwx that often
    const bool silu = p.relu == 2;
    // diagnostic: PA_PG_STAMP_FILE=<path> PA_PG_STAMP_SHAPE=M,K,N [PA_PG_STAMP_SKIP=n]: the (n + 1)-th launch of that shape runs
    // the stamped kernel and its per-wave clock stamps are written to the file (synchronises; scripts/pgemm_stamps.py)
    static const char* stamp_file = getenv("PA_PG_STAMP_FILE");
    if (stamp_file && bm == 128 && bn == 64 && silu) {
        static int sm = 0, sk = 0, sn = 0, skip = getenv("PA_PG_STAMP_SKIP") ? atoi(getenv("PA_PG_STAMP_SKIP")) : 3, seen = 0;
        if (!sm && getenv("PA_PG_STAMP_SHAPE")) sscanf(getenv("PA_PG_STAMP_SHAPE"), "%d,%d,%d", &sm, &sk, &sn);
        if (p.M == sm && p.ktot == sk && p.N == sn && seen++ == skip) {
            const size_t n = (size_t)grid * 4 * 64;
            unsigned long long* d = nullptr;
            if (hipMalloc(&d, n * 8) != hipSuccess) return hipErrorOutOfMemory;
            (void)hipMemsetAsync(d, 0, n * 8, s);
            p.clk = d;
            hipLaunchKernelGGL((pgemm_kernel<128, 64, true, true>), dim3(grid), dim3(256), 0, s, p);
            std::vector<unsigned long long> hst(n);
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(hst.data(), d, n * 8, hipMemcpyDeviceToHost);
            (void)hipFree(d);
            if (FILE* f = fopen(stamp_file, "wb")) {
                const int hdr[4] = {grid, p.ktot / 32, p.tiles_m, p.tiles_n};
                fwrite(hdr, 4, 4, f);
                fwrite(hst.data(), 8, n, f);
                fclose(f);
            }
            return hipGetLastError();
        }
    }
#define PA_PG_LAUNCH(BM_, BN_)                                                                               \
    do {                                                                                                     \
        if (p.up_out) hipLaunchKernelGGL((pgemm_kernel<BM_, BN_, true, false, true>), dim3(grid), dim3(256), 0, s, p); \
        else if (silu) hipLaunchKernelGGL((pgemm_kernel<BM_, BN_, true>), dim3(grid), dim3(256), 0, s, p);   \
        else hipLaunchKernelGGL((pgemm_kernel<BM_, BN_, false>), dim3(grid), dim3(256), 0, s, p);            \
    } while (0)
    if (p.up_out && (!silu || p.up_px_stride % 4 || p.up_row_stride % 4 || p.up_img_stride % 4 || (reinterpret_cast<unsigned long long>(p.up_out) & 15ull)))
        return hipErrorInvalidValue;   // (the up-sampled copy exists for SiLU layers only)
    if (bn == 32) PA_PG_LAUNCH(128, 32);
    else if (bm == 128) PA_PG_LAUNCH(128, 64);
    else PA_PG_LAUNCH(64, 64);
#undef PA_PG_LAUNCH
    return hipGetLastError();
}

}  // namespace pa
