// "Emulated fp32" persistent implicit-GEMM convolution for gfx950: fp32 results from the bf16 matrix cores.
//
// The exact fp32 matrix instruction (v_mfma_f32_32x32x2_f32, pigemm.hip) runs at the fp32 VECTOR rate, 1/16 of the bf16 rate,
// and shares its SIMD with the vector ALU (SQ_VALU_MFMA_COEXEC_CYCLES = 0: DESIGN.md 5.1c). Here every fp32 operand is split
// into three bf16 slices -- x = s0 + s1 + s2 with s0 = bf16(x), s1 = bf16(x - s0), s2 = bf16(x - s0 - s1), an EXACT
// decomposition of the 24-bit significand (each subtraction is exact in fp32) -- and a product a * b becomes the six leading
// cross products a_i * b_j, i + j <= 2, each one v_mfma_f32_32x32x16_bf16: a bf16 x bf16 product is exact in fp32 and the
// instruction accumulates in fp32. The dropped terms (i + j >= 3) are below 2^-24 |a b|: the sum is as close to the float64
// product as the exact fp32 instruction's (measured on the oracle network: 2.0e-6 against 3.5e-6 on log-probabilities,
// tests/emulated_fp32_study.py; per layer: tests/test_psgemm.py against float64). Six bf16 instructions cost 6/16 of the matrix
// time of the fp32 instructions they replace, and the bf16 pipe runs BESIDE the vector ALU (scripts/micro/mfma_f32_mix.hip).
// This is the reference's arithmetic (playaid/ai_runner.py:191-224's network, cnn_action_detector.py:29-43) to fp32 rounding,
// never the default: compute_dtype PA_DTYPE_EMULATED_F32 selects it, `value` / `dtype` of bench.py stay on the exact path.
//
// Layout of the work (pigemm.hip's persistent scheme, re-cut for a matrix pipe that is 2.7x faster):
//   * 512 threads, ONE workgroup per CU: waves 0-3 CONSUME (LDS reads, split, matrix instructions, epilogue), waves 4-7 LOAD (every
//     LDS-DMA copy, the pixel address arithmetic, the counted waits) -- one of each per SIMD. A tile is 128 pixels x BN channels
//     (BN = 128 | 64 | 32) and consumer wave w owns pixels 32 w .. 32 w + 31 x ALL BN channels: every activation value is split
//     exactly once per workgroup (a 2 x 2 wave grid would split it twice -- the split, 44 vector instructions per 8 values, is
//     the loop's second cost; with 64-channel tiles it is the first).
//   * activations stay fp32 in HBM and in LDS (LDS-DMA, pigemm's swizzled 128-byte rows) and are split in registers behind
//     their ds_read_b128; the weights were split at fold time (psgemm_pack_weights) into the exact LDS stage image --
//     [tile_n][k-step][3 planes][BN rows][32 k] bf16, 16-byte chunk c of row r at chunk c ^ ((r >> 2) & 3) -- so their DMA
//     is a straight 1 KiB-per-instruction copy and a ds_read_b128 of 16 rows x 4 chunks hits 64 distinct banks.
//   * the ring is NSTAGE k-steps (32 k each) deep and the pipeline is SKEWED by half a k-step: a k-step's second half
//     (k 16..31) is multiplied AFTER the barrier that opens the next k-step, from operands already in registers, while the
//     next k-step's first operands are read and split. A stage is therefore free as soon as its last operand read has
//     returned, one barrier earlier than its last matrix instruction: NSTAGE stages are in flight, not NSTAGE - 1.
//   * between two matrix instructions stand at most one LDS read and two or three instructions of the split, placed by hand
//     (half()): a lone wave per SIMD hides nothing, whatever takes longer than the 32 cycles a matrix instruction occupies the
//     pipe idles it. Measured (profiles/r06_pgemm_split_stamps.txt): 1592 cycles per k-step for 1536 of matrix time on the
//     big-K layers, at the 1.53-1.57 GHz the chip holds in this loop.
//   * the loaders' waits are counted: in front of barrier g + 1 exactly the copies of the (at most NSTAGE - 2) younger stages may
//     be outstanding, s_waitcnt vmcnt(k * NLD) (NLD checked on the built object: tests/test_abi.py). The consumers never wait on
//     vmcnt; their stores are buffer stores whose descriptor drops the rows past M (offset beyond num_records), so a wave issues
//     the SAME stores for every tile, live rows or not (ADVICE round 5 on pigemm.hip's predicate-dependent store count).
//   * the bias is the value the accumulators start from (read from LDS per tile); SiLU / ReLU and the 16-byte stores run
//     from the accumulators: lane = pixel, runs of four consecutive channels. (Measured and NOT kept, commit 32f1f5b: the epilogue
//     handed to the loader waves through 64 KB of LDS staging, and stores transposed to whole cache lines -- same speed and 5-15 %
//     slower: on the short-K layers the stores cost their bytes, 4.6 TB/s of mixed traffic, not issue slots.
//     profiles/r06_pgemm_split_defer.txt)
#include "pa_kernels.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

// timing experiments only (results wrong when != 0; a diagnostic build: hipcc -DPA_PS_ABL=n): 1 = no copies after the prologue,
// 2 = no matrix instructions
#ifndef PA_PS_ABL
#define PA_PS_ABL 0
#endif

namespace pa {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) const f32x4 lds_cf4;
typedef __attribute__((address_space(3))) const u32x4 lds_cu4;
typedef __attribute__((address_space(3))) float lds_f;

__device__ __forceinline__ i32x4 ps_rsrc(const void* base, unsigned num_bytes) {
    const unsigned long long a = (unsigned long long)base;
    return i32x4{__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff), (int)num_bytes, 0x00020000};
}

// 16 bytes per lane, L2 / HBM -> LDS at lds_addr + 16 * lane (buffer_load_dwordx4 ... lds; M0 = destination). As inline assembly
// (wino.hip's reasons): hipcc then knows nothing of these copies and the kernel's own counted waits are the only ones. M0 is written
// and read in the SAME statement (it is compiler-reserved and cannot be declared; nothing else here lives in it: tests/test_abi.py).
__device__ __forceinline__ void ps_dma16(i32x4 rsrc, int voff_bytes, int soff_bytes, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :
                 : "v"(voff_bytes), "s"(rsrc), "s"(soff_bytes), "s"(lds_addr)
                 : "memory");
}

// n / d and the remainder for a WAVE-UNIFORM 0 <= n < 2^25 (pigemm.hip's pg_sdiv)
__device__ __forceinline__ int ps_sdiv(int n, int d, unsigned magic, int& rem) {
    int q = (int)__umulhi((unsigned)n, magic);
    int r = n - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; r -= d; }
    rem = r;
    return q;
}

__device__ __forceinline__ unsigned ps_cvt_pk_bf16(float a, float b) {   // (lo = a, hi = b), round to nearest even
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// x - (the bf16 in the LOW / HIGH half of q, widened): exact in fp32. As assembly so that hipcc cannot pair two of them into a
// v_pk_add_f32, which costs ~13 cycles beside bf16 matrix instructions where two scalar ones cost 8 (MI355X_MICROARCH.md, price of a filler)
__device__ __forceinline__ float ps_sub_lo(float x, unsigned q) {
    float r;
    const unsigned w = q << 16;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(w));
    return r;
}
__device__ __forceinline__ float ps_sub_hi(float x, unsigned q) {
    float r;
    const unsigned w = q & 0xffff0000u;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(w));
    return r;
}

// eight fp32 values -> their three bf16 slices as matrix operands (element j of the fragment = value j)
__device__ __forceinline__ void ps_split8(const f32x4 lo, const f32x4 hi, u32x4& s0, u32x4& s1, u32x4& s2) {
    const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned q0 = ps_cvt_pk_bf16(x[2 * q], x[2 * q + 1]);
        const float r0 = ps_sub_lo(x[2 * q], q0), r1 = ps_sub_hi(x[2 * q + 1], q0);
        const unsigned q1 = ps_cvt_pk_bf16(r0, r1);
        const float t0 = ps_sub_lo(r0, q1), t1 = ps_sub_hi(r1, q1);
        s0[q] = q0;
        s1[q] = q1;
        s2[q] = ps_cvt_pk_bf16(t0, t1);
    }
}

__device__ __forceinline__ f32x16 ps_mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

constexpr int ps_b_pieces(int bn) { return bn == 32 ? 8 : bn * 3 / 16; }   // 1 KiB DMA pieces of a stage's weight image (32: 6 + 2 of padding)

}  // namespace

// ACT: 0 none, 1 ReLU, 2 SiLU. RES: a residual (addressed like the output) is added, before the activation (ResNet) or after it
// (YOLOv5's Bottleneck) by p.res_after; it may alias the output (in-place Bottlenecks: every value is read by the lane that writes it).
//
// 512 threads: waves 0-3 CONSUME (LDS reads, split, matrix instructions, epilogue), waves 4-7 LOAD (every LDS-DMA copy of the
// workgroup, the pixel address arithmetic, the counted waits); one of each per SIMD. An LDS-DMA instruction costs its wave 60-180
// cycles of issue (MI355X_MICROARCH.md, cycle constants) -- seven to ten of them per k-step in a wave whose matrix instructions
// want a slot every 32 cycles would idle the pipe a third of the time; in a partner wave they cost the consumer nothing but the
// shared barrier. The consumers never wait on vmcnt (their only vector-memory instructions are the tile's stores and residual loads).
template <int BN, int NSTAGE, int ACT, bool RES>
__global__ __launch_bounds__(512, 2) void psgemm_kernel(const GemmParams p, const unsigned short* __restrict__ wsp, unsigned out_bytes, unsigned up_bytes) {
    constexpr int BM = 128, CB = BN / 32;
    constexpr int A_BYTES = BM * 128;
    constexpr int PB = ps_b_pieces(BN) / 4;       // weight pieces per loader wave and k-step
    constexpr int B_BYTES = ps_b_pieces(BN) * 1024;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int NLD = 4 + PB;                   // LDS-DMA instructions per loader wave and k-step
    constexpr int NM = 6 * CB;                    // matrix instructions per half k-step
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NSTAGE * STAGE + BN * 4];

    // --- this workgroup's tiles: one channel column, every lm-th pixel tile of its XCD's contiguous share (as pigemm.hip) ----
    const int b = blockIdx.x, xcd = b & 7, local = b >> 3, per = p.pg_per;
    const int TN = p.tiles_n, TM = p.tiles_m;
    const int LM = per / TN;
    const int tile_n = local % TN, lm = local / TN;
    const int t_lo = (int)(((long long)xcd * TM) >> 3), t_hi = (int)(((long long)(xcd + 1) * TM) >> 3);
    const int nt = t_lo + lm < t_hi ? (t_hi - t_lo - lm + LM - 1) / LM : 0;
    if (nt == 0) return;
    const int nk = p.ktot >> 5;
    const int total = nt * nk;

    const int tid = threadIdx.x;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(lds_f*)(float*)lds;
    const int nwx = p.pg_nwx, nwy = p.pg_nwy;

    // bias of the column -> LDS (the accumulators of every tile start from it)
    float* const bias_s = (float*)(lds + NSTAGE * STAGE);
    if (tid < BN) bias_s[tid] = p.bias ? p.bias[tile_n * BN + tid] : 0.f;

    if (wave_id >= 4) {
        // =============================== loader waves ===============================
        const int lw = wave_id - 4, ltid = tid - 256;
        const int row0 = ltid >> 3;
        const int colq = (ltid & 7) ^ ((row0 >> 1) & 7);   // LDS chunk c of activation row r holds logical chunk c ^ ((r >> 1) & 7)
        const i32x4 act_rs = ps_rsrc(p.act, 0xffffffffu);
        const i32x4 wgt_rs = ps_rsrc(wsp + (size_t)tile_n * nk * (B_BYTES / 2), 0xffffffffu);
        // pixel addressing (pigemm.hip: scalar base of a 32-pixel run + the lane's distance, wraps folded in), in bytes
        const int in_ps = p.in_px_stride * p.stride * 4, in_rs = p.in_row_stride * p.stride * 4;
        const int in_wrap_x = in_rs - p.wo * in_ps;
        const int in_wrap_y = p.in_img_stride * 4 - p.pg_ho * in_rs;
        const int in_org = (p.off_y * p.in_row_stride + p.off_x * p.in_px_stride) * 4;
        int in_lane = row0 * in_ps + colq * 16;
        asm volatile("" : "+v"(in_lane));
        int in_last;   // pixel M - 1: what the rows past M of a partial last tile read (computed, dropped)
        {
            int rem, ox;
            const int img = ps_sdiv(p.M - 1, p.howo, p.pg_magic_howo, rem);
            const int oy = ps_sdiv(rem, p.wo, p.pg_magic_wo, ox);
            in_last = img * (p.in_img_stride * 4) + oy * in_rs + ox * in_ps + in_org + colq * 16;
        }
        auto in_offset = [&](int m_base) {
            int rem, ox_b;
            const int img_b = ps_sdiv(m_base, p.howo, p.pg_magic_howo, rem);
            int oy = ps_sdiv(rem, p.wo, p.pg_magic_wo, ox_b);
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
        int i_tile = t_lo + lm, i_ks = 0, i_ky = 0, i_kx = 0, i_kc = 0;
        int a_off[4];
        auto rows_of = [&](int tile_m) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a_off[i] = in_offset(tile_m * BM + 32 * i);
        };
        rows_of(i_tile);
        int b_lane = lw * PB * 1024 + lane * 16;
        asm volatile("" : "+v"(b_lane));
        auto issue = [&](int slot) {
            const unsigned sb = lds_base + slot * STAGE;
            const int tapoff = (i_ky * p.in_row_stride + i_kx * p.in_px_stride + i_kc) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) ps_dma16(act_rs, a_off[i], tapoff, sb + lw * 1024 + i * 4096);
            const int koff = i_ks * B_BYTES;
#pragma unroll
            for (int j = 0; j < PB; ++j) ps_dma16(wgt_rs, b_lane, koff + j * 1024, sb + A_BYTES + (lw * PB + j) * 1024);
            i_kc += 32;
            if (i_kc == p.chunk) {
                i_kc = 0;
                if (++i_kx == p.kw_taps) { i_kx = 0; ++i_ky; }
            }
            if (++i_ks == nk) {
                i_ks = 0; i_ky = 0; i_kx = 0; i_kc = 0;
                i_tile += LM;
                rows_of(i_tile < t_hi ? i_tile : t_hi - 1);
            }
        };
        // the stage in front of barrier g + 1 must have landed: exactly the copies of the (at most NSTAGE - 2) stages issued after it may be outstanding
        auto wait_stage = [&](int younger) {
            if (NSTAGE >= 4 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLD) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        static_assert(NSTAGE >= 2 && NSTAGE <= 4, "wait_stage covers rings of two to four stages");
        int slot = 0;
#pragma unroll
        for (int s = 0; s < NSTAGE; ++s)
            if (s < total) issue(s);
        {
            const int younger = (total < NSTAGE ? total : NSTAGE) - 1;   // stages 1 .. behind stage 0
            if (younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NLD) : "memory");
            else wait_stage(younger);
        }
        __builtin_amdgcn_s_barrier();   // stage 0 (and bias_s) in LDS
        for (int g = 0; g + 1 < total; ++g) {
            // stages issued so far: min(g + NSTAGE, total); behind stage g + 1: min(g + NSTAGE, total) - (g + 2)
            const int inflight = (g + NSTAGE < total ? g + NSTAGE : total) - (g + 2);
            wait_stage(inflight);
            __builtin_amdgcn_s_barrier();   // stage g + 1 landed; every consumer's reads of stage g have returned
            if (g + NSTAGE < total && !(PA_PS_ABL & 1)) issue(slot);
            slot = slot + 1 == NSTAGE ? 0 : slot + 1;
        }
        __builtin_amdgcn_s_barrier();   // (the consumers' barrier of the last k-step)
        return;
    }

    // =============================== consumer waves ===============================
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t res_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(RES ? p.residual : p.out), 0, (int)out_bytes, 0x00020000);
    const int ch0 = tile_n * BN + 4 * lh;
    // output, in BYTES: O(m) = (img * OIS + (oy + pad) * ORS + (ox + pad) * OPS + ch0) * 4
    const int out_wrap_x = (p.out_row_stride - p.wo * p.out_px_stride) * 4;
    const int out_wrap_y = (p.out_img_stride - p.pg_ho * p.out_row_stride) * 4;
    int out_lane = (lr * p.out_px_stride + p.out_pad * (p.out_row_stride + p.out_px_stride) + ch0) * 4;
    asm volatile("" : "+v"(out_lane));
    auto out_offset = [&](int m_base) -> unsigned {
        int rem, ox_b;
        const int img_b = ps_sdiv(m_base, p.howo, p.pg_magic_howo, rem);
        int oy = ps_sdiv(rem, p.wo, p.pg_magic_wo, ox_b);
        int off = (img_b * p.out_img_stride + oy * p.out_row_stride + ox_b * p.out_px_stride) * 4 + out_lane;
        int ox = ox_b + lr;
        for (int w = 0; w < nwx; ++w) {
            const bool c = ox >= p.wo;
            ox -= c ? p.wo : 0;
            off += c ? out_wrap_x : 0;
            oy += c ? 1 : 0;
        }
        for (int w = 0; w < nwy; ++w) {
            const bool c = oy >= p.pg_ho;
            oy -= c ? p.pg_ho : 0;
            off += c ? out_wrap_y : 0;
        }
        return m_base + lr < p.M ? (unsigned)off : 0x80000000u;   // past M: beyond num_records, dropped (the launcher keeps buffers under 2 GB)
    };

    // the nearest-neighbour x2 up-sampled copy (p.up_out, YOLOv5's nn.Upsample behind model.10 / model.14 as four more stores of the
    // producer): pixel (oy, ox) -> (2 oy, 2 ox) .. (2 oy + 1, 2 ox + 1): the same walk with doubled row and pixel strides
    const __amdgpu_buffer_rsrc_t up_rs = __builtin_amdgcn_make_buffer_rsrc(p.up_out ? p.up_out : p.out, 0, (int)(p.up_out ? up_bytes : out_bytes), 0x00020000);
    const int up_rs_b = 2 * p.up_row_stride * 4, up_ps_b = 2 * p.up_px_stride * 4;
    const int up_wrap_x = up_rs_b - p.wo * up_ps_b, up_wrap_y = p.up_img_stride * 4 - p.pg_ho * up_rs_b;
    int up_lane = lr * up_ps_b + (p.up_pad * (p.up_row_stride + p.up_px_stride) + ch0) * 4;
    asm volatile("" : "+v"(up_lane));
    auto up_offset = [&](int m_base) -> unsigned {
        int rem, ox_b;
        const int img_b = ps_sdiv(m_base, p.howo, p.pg_magic_howo, rem);
        int oy = ps_sdiv(rem, p.wo, p.pg_magic_wo, ox_b);
        int off = img_b * p.up_img_stride * 4 + oy * up_rs_b + ox_b * up_ps_b + up_lane;
        int ox = ox_b + lr;
        for (int w = 0; w < nwx; ++w) {
            const bool c = ox >= p.wo;
            ox -= c ? p.wo : 0;
            off += c ? up_wrap_x : 0;
            oy += c ? 1 : 0;
        }
        for (int w = 0; w < nwy; ++w) {
            const bool c = oy >= p.pg_ho;
            oy -= c ? p.pg_ho : 0;
            off += c ? up_wrap_y : 0;
        }
        return m_base + lr < p.M ? (unsigned)off : 0x80000000u;
    };

    // LDS read addresses (bytes) of the two k halves: this wave's pixel rows; the weight rows lr of each 32-channel block
    unsigned a_rd[2], a_rd2[2], b_rd[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        a_rd[h] = lds_base + (wave_id * 32 + lr) * 128 + (((4 * h + 2 * lh) ^ ((lr >> 1) & 7)) * 16);
        a_rd2[h] = a_rd[h] ^ 16u;
        b_rd[h] = lds_base + A_BYTES + (lr * 4 + ((2 * h + lh) ^ ((lr >> 2) & 3))) * 16;
        asm volatile("" : "+v"(a_rd[h]), "+v"(a_rd2[h]), "+v"(b_rd[h]));
    }

    u32x4 a_c[3], a_n[3], b_c[CB][3], b_n[CB][3];
    f32x16 acc[CB];
    auto read_b = [&](u32x4 (&bb)[CB][3], unsigned sb, int h) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int s = 0; s < 3; ++s) bb[cb][s] = *(lds_cu4*)(size_t)(b_rd[h] + sb + (s * BN + cb * 32) * 64);
    };
    // Half a k-step: NM matrix instructions (six cross products per 32-channel block, smallest terms first, the blocks taking
    // turns) from the operand registers (a, bb). Dealt out among them, ONE OR TWO PER GAP, go the next half's LDS reads (the raw
    // pixels first, then the weight fragments -> bn) and the 44 vector instructions that split the raw values (-> an). A lone
    // wave per SIMD hides nothing by itself: whatever stands between two matrix instructions and takes longer than the 32
    // cycles one of them occupies the pipe idles it (measured on the first form of this loop, reads in bursts of 14 and the
    // split in chunks of six: 2300 cycles per k-step for 1536 of matrix time; SQ_ACTIVE_INST / SQ_WAIT_INST in
    // profiles/r06_pgemm_split_pmc.txt). Per gap the issue budget is 32 - 8 (the matrix instruction's own): one ds_read_b128
    // (~13) and two or three vector instructions (4 each). The split's instruction order keeps every result two instructions away
    // from its first reader (two value pairs in lock step), so no dependent-issue stall and no hazard nop.
    // `pre`: matrix instructions issued before `open` runs (the barrier of the second half); reads start in gap `pre`.
    constexpr unsigned HI16 = 0xffff0000u;
    auto half = [&](const u32x4 (&a)[3], const u32x4 (&bb)[CB][3], u32x4 (&an)[3], u32x4 (&bn)[CB][3], unsigned sb, int h, auto pre_c, auto&& open) {
        constexpr int PRE = decltype(pre_c)::value;
        constexpr int TA[6] = {0, 1, 2, 0, 1, 0}, TB[6] = {2, 1, 0, 1, 0, 0};
        constexpr int NR = 2 + 3 * CB;                                     // LDS reads of the next half
        constexpr int OS = PRE + (NM >= 24 ? 6 : (NM >= 12 ? 3 : 1));      // first gap that takes split instructions
        f32x4 raw[2];
        float x[8], r[8];
        unsigned w[8], q0[4], q1[4];
        auto sop = [&](int k) {   // one instruction of the split; k = 22 G + j, G = which raw quad (pairs 2 G, 2 G + 1)
            const int G = k / 22, j = k % 22, pq = 2 * G + (j & 1), e = 2 * pq;
            if (j < 2) {
                if (j == 0) {
                    x[4 * G] = raw[G].x; x[4 * G + 1] = raw[G].y; x[4 * G + 2] = raw[G].z; x[4 * G + 3] = raw[G].w;
                }
                q0[pq] = ps_cvt_pk_bf16(x[e], x[e + 1]);
                an[0][pq] = q0[pq];
            } else if (j < 4) {
                asm("v_lshlrev_b32 %0, 16, %1" : "=v"(w[e]) : "v"(q0[pq]));
            } else if (j < 6) {
                asm("v_and_b32 %0, %1, %2" : "=v"(w[e + 1]) : "s"(HI16), "v"(q0[pq]));
            } else if (j < 8) {
                asm("v_sub_f32 %0, %1, %2" : "=v"(r[e]) : "v"(x[e]), "v"(w[e]));
            } else if (j < 10) {
                asm("v_sub_f32 %0, %1, %2" : "=v"(r[e + 1]) : "v"(x[e + 1]), "v"(w[e + 1]));
            } else if (j < 12) {
                q1[pq] = ps_cvt_pk_bf16(r[e], r[e + 1]);
                an[1][pq] = q1[pq];
            } else if (j < 14) {
                asm("v_lshlrev_b32 %0, 16, %1" : "=v"(w[e]) : "v"(q1[pq]));
            } else if (j < 16) {
                asm("v_and_b32 %0, %1, %2" : "=v"(w[e + 1]) : "s"(HI16), "v"(q1[pq]));
            } else if (j < 18) {
                asm("v_sub_f32 %0, %1, %2" : "=v"(r[e]) : "v"(r[e]), "v"(w[e]));
            } else if (j < 20) {
                asm("v_sub_f32 %0, %1, %2" : "=v"(r[e + 1]) : "v"(r[e + 1]), "v"(w[e + 1]));
            } else {
                an[2][pq] = ps_cvt_pk_bf16(r[e], r[e + 1]);
            }
        };
        auto rop = [&](int k) {   // one LDS read of the next half: the two raw pixel quads first (the split waits for them)
            if (k == 0) raw[0] = *(lds_cf4*)(size_t)(a_rd[h] + sb);
            else if (k == 1) raw[1] = *(lds_cf4*)(size_t)(a_rd2[h] + sb);
            else bn[(k - 2) % CB][(k - 2) / CB] = *(lds_cu4*)(size_t)(b_rd[h] + sb + (((k - 2) / CB) * BN + ((k - 2) % CB) * 32) * 64);
        };
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            if (i == PRE) open();
            const int term = i / CB, cb = i % CB;
            if (!(PA_PS_ABL & 2)) acc[cb] = ps_mfma(bb[cb][TB[term]], a[TA[term]], acc[cb]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                int at = PRE + k;
                at = at > NM - 1 ? NM - 1 : at;
                if (at == i) rop(k);
            }
#pragma unroll
            for (int k = 0; k < 44; ++k) {
                int at = OS + (k * (NM - OS)) / 44;
                if (at == i) sop(k);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto load_bias = [&]() {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b4 = *(const f32x4*)(bias_s + cb * 32 + 8 * g + 4 * lh);
                acc[cb][4 * g] = b4.x; acc[cb][4 * g + 1] = b4.y; acc[cb][4 * g + 2] = b4.z; acc[cb][4 * g + 3] = b4.w;
            }
    };

    __builtin_amdgcn_s_barrier();   // stage 0 (and bias_s) in LDS
    {
        const f32x4 lo = *(lds_cf4*)(size_t)a_rd[0], hi = *(lds_cf4*)(size_t)a_rd2[0];
        read_b(b_c, 0u, 0);
        ps_split8(lo, hi, a_c[0], a_c[1], a_c[2]);
    }

    // diagnostic build only (-DPA_PS_STAMP, p.clk = 8 words per consumer wave): shader clock (s_memtime) and the constant 100 MHz clock
    // (s_memrealtime) around the tile loop -> cycles per k-step and the clock the chip holds in this loop; and the same around the
    // k-steps alone (epilogues excluded). Values go to a buffer nothing else reads; in the product kernel no stamp executes.
#ifdef PA_PS_STAMP
    unsigned long long st_t0 = 0, st_r0 = 0, st_k = 0;
    if (p.clk) { st_t0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    int slot = 0, g = 0;
    for (int t = 0; t < nt; ++t) {
        const int tile_m = t_lo + lm + t * LM;
#ifdef PA_PS_STAMP
        const unsigned long long st_ks = p.clk ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
        load_bias();
        f32x4 res4[CB][4];
        const unsigned o_off = out_offset(tile_m * BM + wave_id * 32);
        const unsigned u_off = p.up_out ? up_offset(tile_m * BM + wave_id * 32) : 0u;
        for (int ks = 0; ks < nk; ++ks, ++g) {
            const unsigned sb = slot * STAGE;
            // ---- first half: multiply k 0..15 from registers; read and split k 16..31 of the same stage (the reads go out BEHIND the
            //      first matrix instruction: in front of it hipcc's wait for that instruction's operands, read in the previous
            //      iteration -- it loses the count over the loop's back edge -- would wait for brand-new reads as well) ----
            if (RES && BN < 128 && ks == nk - 1) {   // the tile's residual values: requested a k-step ahead of their use
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const unsigned off = o_off == 0x80000000u ? o_off : o_off + (unsigned)(cb * 32 + 8 * gq) * 4u;
                        res4[cb][gq] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res_rs, off, 0, 0));
                    }
            }
            half(a_c, b_c, a_n, b_n, sb, 1, std::integral_constant<int, 0>{}, [] {});
            // every LDS read of stage g has returned: behind the barrier below its slot is overwritten
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // ---- second half: multiply k 16..31 from registers; a few matrix instructions in, open k-step g + 1 (barrier) and read
            //      and split its first operands under the rest. (No branch around this for the last k-step: it then reads a slot
            //      nobody refills -- stale, unused -- and the loaders join one barrier more; straight-line code keeps hipcc's
            //      operand waits out of the matrix instructions' way.) ----
            const int nslot = slot + 1 == NSTAGE ? 0 : slot + 1;
            half(a_n, b_n, a_c, b_c, (unsigned)(nslot * STAGE), 0, std::integral_constant<int, (NM >= 24 ? 2 : 1)>{}, [] {
                __builtin_amdgcn_s_barrier();   // stage g + 1 landed (the loaders waited for it); stage g's slot is free
                __builtin_amdgcn_sched_barrier(0);
            });
            slot = nslot;
        }
#ifdef PA_PS_STAMP
        if (p.clk) st_k += __builtin_amdgcn_s_memtime() - st_ks;
#endif
        // ---- epilogue of the tile, straight from the accumulators: lane = pixel lr of the wave's 32, channels ch0 + 32 cb + 8 g + 0..3 ----
        if (RES && BN >= 128) {   // 128-channel tiles: the residual's 64 registers are free only now (the operand sets of the last half are dead)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const unsigned off = o_off == 0x80000000u ? o_off : o_off + (unsigned)(cb * 32 + 8 * gq) * 4u;
                    res4[cb][gq] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res_rs, off, 0, 0));
                }
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                f32x4 v = f32x4{acc[cb][4 * gq], acc[cb][4 * gq + 1], acc[cb][4 * gq + 2], acc[cb][4 * gq + 3]};   // (bias inside)
                if (RES && !p.res_after) v += res4[cb][gq];
                if (ACT == 2) {
                    v.x = silu_fast(v.x); v.y = silu_fast(v.y); v.z = silu_fast(v.z); v.w = silu_fast(v.w);
                } else if (ACT == 1) {
                    v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
                }
                if (RES && p.res_after) v += res4[cb][gq];
                const unsigned off = o_off == 0x80000000u ? o_off : o_off + (unsigned)(cb * 32 + 8 * gq) * 4u;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), out_rs, off, 0, 0);
                if (p.up_out) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const unsigned uo = u_off == 0x80000000u ? u_off : u_off + (unsigned)((q4 >> 1) * p.up_row_stride + (q4 & 1) * p.up_px_stride + cb * 32 + 8 * gq) * 4u;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), up_rs, uo, 0, 0);
                    }
                }
            }
    }
#ifdef PA_PS_STAMP
    if (p.clk && lane == 0) {
        unsigned long long* c = p.clk + ((size_t)blockIdx.x * 4 + wave_id) * 8;
        c[0] = __builtin_amdgcn_s_memtime() - st_t0;
        c[1] = __builtin_amdgcn_s_memrealtime() - st_r0;
        c[2] = st_k;
        c[3] = (unsigned long long)total;
        c[4] = (unsigned long long)nt;
    }
#endif
}

// channels per workgroup = per stage image of the weight layout (`residual` is part of the choice only for the A/B switch PA_PS_RES128:
// 64-channel tiles request the residual a k-step ahead of the epilogue, 128-channel tiles -- no registers to spare until the last
// operand sets are dead -- at its start)
int psgemm_pick_bn(int N, int residual) {
    static const int res128 = getenv("PA_PS_RES128") ? atoi(getenv("PA_PS_RES128")) : 1;   // 0: residual layers on 64-channel tiles (A/B)
    return N % 128 == 0 && (!residual || res128) ? 128 : (N % 64 == 0 ? 64 : (N % 32 == 0 ? 32 : 0));
}

size_t psgemm_weight_elems(int N, int ktot, int residual) {
    const int bn = psgemm_pick_bn(N, residual);
    if (bn == 0 || ktot % 32 != 0 || N <= 0 || ktot <= 0) return 0;
    return (size_t)(N / bn) * (ktot / 32) * (ps_b_pieces(bn) * 512);
}

static inline unsigned short ps_bf16_rne(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return (unsigned short)(u >> 16);   // inf / nan: truncate
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static inline float ps_bf16_f(unsigned short h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// w [N][ktot] fp32 (K contiguous: [cout][tap][cin], BatchNorm folded) -> the kernel's stage images (host):
// [tile_n][k-step][plane s][row r][chunk c'][8] bf16, plane s = the s-th bf16 slice, chunk c' of row r holding k 8 (c' ^ ((r >> 2) & 3)) ..
void psgemm_pack_weights(const float* w, int N, int ktot, int residual, unsigned short* out) {
    const int bn = psgemm_pick_bn(N, residual), nk = ktot / 32, tn_n = N / bn;
    const size_t stage = (size_t)ps_b_pieces(bn) * 512;   // elements
    memset(out, 0, psgemm_weight_elems(N, ktot, residual) * sizeof(unsigned short));
    for (int tn = 0; tn < tn_n; ++tn)
        for (int ks = 0; ks < nk; ++ks) {
            unsigned short* img = out + ((size_t)tn * nk + ks) * stage;
            for (int r = 0; r < bn; ++r)
                for (int c = 0; c < 4; ++c) {
                    const int cp = c ^ ((r >> 2) & 3);
                    for (int j = 0; j < 8; ++j) {
                        float x = w[(size_t)(tn * bn + r) * ktot + ks * 32 + c * 8 + j];
                        for (int s = 0; s < 3; ++s) {
                            const unsigned short hq = ps_bf16_rne(x);
                            img[((size_t)(s * bn + r) * 4 + cp) * 8 + j] = hq;
                            x -= ps_bf16_f(hq);
                        }
                    }
                }
        }
}

// Conv mode of GemmParams (no gather, no second source, no split-K); p.wgt is ignored, wsp = psgemm_pack_weights' image of it.
// out_floats: floats from p.out to the end of its buffer (bounds of the output descriptor, < 2^29).
hipError_t launch_psgemm(const GemmParams& p_in, const unsigned short* wsp, size_t out_floats, size_t up_floats, hipStream_t s) {
    GemmParams p = p_in;
    const int bn = psgemm_pick_bn(p.N, p.residual != nullptr);
    if (p.gather || p.k2_steps || bn == 0 || p.chunk % 32 != 0 || p.M <= 0 || p.M >= (1 << 24) || p.howo >= (1 << 16) ||
        p.ktot != p.taps * p.chunk || !wsp || out_floats == 0 || out_floats >= (1ull << 29))
        return hipErrorInvalidValue;
    p.tiles_n = p.N / bn;
    p.tiles_m = (p.M + 127) / 128;
    // one workgroup per CU = 32 per XCD, a multiple of the channel columns, no more per column than the XCD's share of pixel tiles
    const int share = (p.tiles_m + 7) / 8;
    int lm = 32 / p.tiles_n;
    lm = lm < 1 ? 1 : (lm > share ? share : lm);
    const int per = lm * p.tiles_n;
    const int grid = per * 8;
    auto magic = [](int d) { return (unsigned)std::min<unsigned long long>(((1ull << 32) + d - 1) / d, 0xffffffffull); };
    if (p.howo % p.wo != 0) return hipErrorInvalidValue;
    p.pg_per = per;
    p.pg_ho = p.howo / p.wo;
    p.pg_magic_howo = magic(p.howo);
    p.pg_magic_wo = magic(p.wo);
    p.pg_nwx = 1 + 30 / p.wo;
    p.pg_nwy = (p.pg_ho - 1 + p.pg_nwx) / p.pg_ho;
    const unsigned out_bytes = (unsigned)(out_floats * 4);
#ifdef PA_PS_STAMP
    // PA_PS_STAMP_FILE=<path>: every launch appends "M K N grid | median over consumer waves of: shader cycles, 100 MHz ticks, cycles in k-steps, k-steps, tiles" (synchronises)
    static const char* stamp_file = getenv("PA_PS_STAMP_FILE");
    unsigned long long* clk_dev = nullptr;
    if (stamp_file) {
        if (hipMalloc(&clk_dev, (size_t)grid * 4 * 8 * 8) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemsetAsync(clk_dev, 0, (size_t)grid * 4 * 8 * 8, s);
        p.clk = clk_dev;
    }
#endif
    if (p.up_out && (up_floats == 0 || up_floats >= (1ull << 29) || p.up_px_stride % 4 || p.up_row_stride % 4 || p.up_img_stride % 4 ||
                     (reinterpret_cast<unsigned long long>(p.up_out) & 15ull)))
        return hipErrorInvalidValue;
    const unsigned up_bytes = (unsigned)(up_floats * 4);
#define PA_PS_LAUNCH1(BN_, NS_, RES_)                                                                                                \
    do {                                                                                                                             \
        if (p.relu == 2) hipLaunchKernelGGL((psgemm_kernel<BN_, NS_, 2, RES_>), dim3(grid), dim3(512), 0, s, p, wsp, out_bytes, up_bytes);     \
        else if (p.relu == 1) hipLaunchKernelGGL((psgemm_kernel<BN_, NS_, 1, RES_>), dim3(grid), dim3(512), 0, s, p, wsp, out_bytes, up_bytes); \
        else hipLaunchKernelGGL((psgemm_kernel<BN_, NS_, 0, RES_>), dim3(grid), dim3(512), 0, s, p, wsp, out_bytes, up_bytes);                 \
    } while (0)
#define PA_PS_LAUNCH(BN_, NS_)                                                                                                       \
    do {                                                                                                                             \
        if (p.residual) PA_PS_LAUNCH1(BN_, NS_, true);                                                                               \
        else PA_PS_LAUNCH1(BN_, NS_, false);                                                                                         \
    } while (0)
    // ring depth: as deep as one workgroup per CU allows. PA_PS_STAGES=2 (A/B): a two-stage ring of 80 / 56 KB, which leaves a CU room for a
    // workgroup of another stream's kernel (the Motion-JPEG passes' 52 KB) beside this one -- the chain runs three streams
    static const int stages2 = getenv("PA_PS_STAGES") && atoi(getenv("PA_PS_STAGES")) == 2;
    if (stages2) {
        if (bn == 128) PA_PS_LAUNCH(128, 2);
        else if (bn == 64) PA_PS_LAUNCH(64, 2);
        else PA_PS_LAUNCH(32, 2);
    } else if (bn == 128) PA_PS_LAUNCH(128, 3);
    else if (bn == 64) PA_PS_LAUNCH(64, 4);
    else PA_PS_LAUNCH(32, 4);
#undef PA_PS_LAUNCH1
#ifdef PA_PS_STAMP
    if (clk_dev) {
        std::vector<unsigned long long> hst((size_t)grid * 4 * 8);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(hst.data(), clk_dev, hst.size() * 8, hipMemcpyDeviceToHost);
        (void)hipFree(clk_dev);
        std::vector<double> cyc, clkghz, perk, perk_in;
        for (int w = 0; w < grid * 4; ++w) {
            const unsigned long long* c = &hst[(size_t)w * 8];
            if (c[3] == 0 || c[1] == 0) continue;
            cyc.push_back((double)c[0]);
            clkghz.push_back((double)c[0] / (double)c[1] * 0.1);
            perk.push_back((double)c[0] / (double)c[3]);
            perk_in.push_back((double)c[2] / (double)c[3]);
        }
        auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        if (FILE* f = fopen(stamp_file, "a")) {
            fprintf(f, "M %d K %d N %d bn %d grid %d | wave lifetime %.0f cycles, in-kernel clock %.3f GHz, %.0f cycles per k-step (whole loop), %.0f inside the k-steps\n",
                    p.M, p.ktot, p.N, bn, grid, med(cyc), med(clkghz), med(perk), med(perk_in));
            fclose(f);
        }
    }
#endif
#undef PA_PS_LAUNCH
    return hipGetLastError();
}

}  // namespace pa

#include "../../include/playaid_hip.h"

extern "C" {

size_t pa_conv_weight_bytes(int32_t cin, int32_t cout, int32_t ksize, int32_t compute_dtype, int32_t has_residual) {
    if (cin <= 0 || cout <= 0 || (ksize != 1 && ksize != 3) || cin % 32 != 0 || cout % 32 != 0) return 0;
    if (compute_dtype == PA_DTYPE_F32) return (size_t)cout * ksize * ksize * cin * sizeof(float);
    if (compute_dtype == PA_DTYPE_EMULATED_F32) return pa::psgemm_weight_elems(cout, ksize * ksize * cin, has_residual) * sizeof(unsigned short);
    return 0;
}

int pa_conv_pack_weights(const float* w_host, int32_t cin, int32_t cout, int32_t ksize, int32_t compute_dtype, int32_t has_residual, void* out_host) {
    const size_t bytes = pa_conv_weight_bytes(cin, cout, ksize, compute_dtype, has_residual);
    if (!w_host || !out_host || bytes == 0) return PA_ERR_INVALID_ARG;
    if (compute_dtype == PA_DTYPE_F32) memcpy(out_host, w_host, bytes);
    else pa::psgemm_pack_weights(w_host, cout, ksize * ksize * cin, has_residual, static_cast<unsigned short*>(out_host));
    return PA_OK;
}

int pa_conv2d(const float* x, const void* w, const float* bias, const float* residual, float* out, int32_t n, int32_t height, int32_t width, int32_t cin,
              int32_t cout, int32_t ksize, int32_t stride, int32_t in_pad, int32_t in_px_stride, int32_t out_px_stride, int32_t out_pad, int32_t act,
              int32_t res_after, int32_t compute_dtype, void* stream) {
    if (!x || !w || !out || n <= 0 || height <= 0 || width <= 0 || (ksize != 1 && ksize != 3) || (stride != 1 && stride != 2) || height % stride || width % stride ||
        in_pad < (ksize - 1) / 2 || in_px_stride < cin || out_px_stride < cout || out_pad < 0 || act < 0 || act > 2 ||
        pa_conv_weight_bytes(cin, cout, ksize, compute_dtype, residual != nullptr) == 0)
        return PA_ERR_INVALID_ARG;
    // 16-byte units: the loaders' LDS-DMA reads and the epilogue's dwordx4 stores / residual loads move four floats at an address
    auto misaligned = [](const void* q) { return (reinterpret_cast<unsigned long long>(q) & 15ull) != 0; };
    if (in_px_stride % 4 || out_px_stride % 4 || misaligned(x) || misaligned(w) || misaligned(out) || misaligned(residual) ||
        (reinterpret_cast<unsigned long long>(bias) & 3ull))
        return PA_ERR_INVALID_ARG;
    const int oh = height / stride, ow = width / stride;
    const int in_wb = width + 2 * in_pad, in_hb = height + 2 * in_pad, out_wb = ow + 2 * out_pad, out_hb = oh + 2 * out_pad;
    if ((long long)n * in_hb * in_wb * in_px_stride >= (1ll << 29) || (long long)n * out_hb * out_wb * out_px_stride >= (1ll << 29)) return PA_ERR_CAPACITY;
    pa::GemmParams p;
    memset(&p, 0, sizeof(p));
    p.act = x;
    p.wgt = static_cast<const float*>(w);
    p.bias = bias;
    p.residual = residual;
    p.out = out;
    p.M = n * oh * ow;
    p.N = cout;
    p.taps = ksize * ksize;
    p.kw_taps = ksize;
    p.chunk = cin;
    p.ktot = p.taps * p.chunk;
    p.howo = oh * ow;
    p.wo = ow;
    p.in_px_stride = in_px_stride;
    p.in_row_stride = in_wb * in_px_stride;
    p.in_img_stride = in_hb * in_wb * in_px_stride;
    p.stride = stride;
    p.off_y = p.off_x = in_pad - (ksize - 1) / 2;
    p.out_px_stride = out_px_stride;
    p.out_row_stride = out_wb * out_px_stride;
    p.out_img_stride = out_hb * out_wb * out_px_stride;
    p.out_pad = out_pad;
    p.relu = act;
    p.res_after = res_after;
    p.splitk = 1;
    hipError_t e;
    if (compute_dtype == PA_DTYPE_EMULATED_F32) e = pa::launch_psgemm(p, static_cast<const unsigned short*>(w), (size_t)n * p.out_img_stride, 0, static_cast<hipStream_t>(stream));
    else if (residual) return PA_ERR_INVALID_ARG;   // (the exact persistent kernel has no residual epilogue: Winograd / the patch kernel take those layers)
    else e = pa::launch_pgemm(p, 0, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? PA_OK : (e == hipErrorInvalidValue ? PA_ERR_INVALID_ARG : PA_ERR_HIP);
}

}  // extern "C"
