// Motion-JPEG frame decode on the device: baseline JPEG files -> uint8 [n][H][W][3] BGR frames in HBM.
//
// Replaces the per-frame work of cv2.VideoCapture.read / cv2.imread in the reference (playaid/ai_runner.py:153,404-405,446;
// playaid/manuscript.py:154-155) for Motion-JPEG streams and JPEG image sequences. The arithmetic is libjpeg(-turbo)'s with
// its defaults, which is what OpenCV's JPEG reader runs: Huffman entropy decoding (T.81 F.2.2), de-quantisation + the
// slow-but-accurate integer IDCT (jidctint.c), "fancy" triangle-filter chroma up-sampling with libjpeg's edge
// replication (jdsample.c h2v2 / h2v1, jdmainct.c), YCbCr -> RGB (jdcolor.c). Bit-exact against oracle/jpeg.py::decode,
// which is pinned byte for byte against PIL.Image.open (live libjpeg-turbo).
//
// Pipeline per call (a call's frames go in GROUPS, each on a stream of the handle's own; see struct pa_mjpeg):
//   host : marker segments of every frame (SOF0/SOF1, DQT, DHT, DRI, SOS) -> frame descriptors + Huffman / quantisation
//          table sets (consecutive frames with identical tables share one set)
//   copy : the compressed bytes of each group (a copy stream that never waits); descriptors and table sets by
//          stage_copy_kernel out of the pinned staging buffers
//   unstuff_count_kernel / unstuff_write_kernel : the CLEAN stream of every frame (stuffed zeros and RSTm markers taken
//                  out, restart positions recorded)
//   sub_decode_kernel<0|1> + sub_verify_plan_kernel : one lane per subsequence of the clean stream finds the decoder
//                  state at its first bit (self-synchronisation, see below); nothing is stored but states and counts
//   sub_scan_kernel : block index at every subsequence's entry
//   sub_decode_kernel<2> : the final pass stores the non-zero quantised AC coefficients (int16) where the scan puts them
//                  -- block number in scan order, zig-zag index -- and every block's DC DIFFERENCE in an array apart
//   dc_scan_kernel : DC differences -> DC coefficients (running sums per component, from zero at every restart interval)
//   idct_kernel  : one thread per 8x8 block of the component rasters gathers its block from the scan-order buffer
//                  (de-zig-zag by constant indices), block in registers -> uint8 sample planes (padded to whole MCUs)
//   ycc420_kernel / ycc_kernel : up-sampling + colour conversion, 8 pixels x 4 (4:2:0) or FV rows per thread, 8-byte stores
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/playaid_hip.h"
#include "jpeg_dct.h"

namespace pa {
namespace mj {

constexpr int LB = 10;           // bits of the direct Huffman lookup
constexpr int CHUNK = 4096;      // bytes per workgroup of the marker scan (256 threads x 16 bytes)
constexpr int MAX_BLOCKS_MCU = 10;
constexpr int SUB_MIN = 128;     // bytes per subsequence (one lane) of the entropy decoder: a power of two >= this, per call
constexpr int RING_DW = 32;      // dwords of a lane's LDS ring (128 bytes of its stream)
constexpr int TOPUP = 16;        // symbols between two ring top-ups
constexpr int WG_SUBS = 256;     // subsequences per workgroup

enum { ERR_HUFF = 1, ERR_RST = 2, ERR_COEF = 4, ERR_SYNC = 8 };

// How far a symbol moves the zig-zag index (T.81 F.2.2.2): the DC symbol to 1; a coefficient RRRRSSSS past its run of
// zeros and itself; ZRL 16; EOB (libjpeg: any other symbol of size 0) beyond the end of the block from wherever it stands.
__host__ __device__ inline int symbol_advance(bool dc, int sym) {
    if (dc) return 1;
    const int s = sym & 15, r = sym >> 4;
    return s ? r + 1 : (r == 15 ? 16 : 64);
}

// Huffman tables as the decoder's lanes read them (LDS image = global layout). A 16-bit entry holds everything a symbol
// needs: code length (bits 0-4), number of extra bits s (5-8), advance of the zig-zag index (9-15); 0 = no entry.
// Codes of up to LB bits are found in lut1 under their first LB bits. Longer codes sit at the top of a canonical code
// space: whenever they all start with six 1-bits (every table whose long codes fill less than 1/64 of the code space --
// the standard tables, and what libjpeg's optimiser produces) they are found in lutB under bits 6..15, so both tables
// are read at once and neither look-up waits for the other. A code in neither table leaves both entries 0 and the lane
// walks the canonical MAXCODE list instead.
struct HuffTables {
    uint16_t lut1[4][1 << LB];  // DC0, DC1, AC0, AC1
    uint16_t lutB[4][1 << LB];
    int32_t maxcode[4][18];     // largest code of length l (-1: none); [17] = sentinel
    int32_t valoff[4][17];      // valptr[l] - mincode[l]
    uint8_t vals[4][256];
};
struct TableSet {
    HuffTables h;
    uint16_t q[4][64];  // quantisation tables, natural order
};
static_assert(sizeof(HuffTables) % 4 == 0, "copied to LDS as dwords");

struct FrameDesc {
    uint32_t scan_off, scan_len;  // entropy-coded segment inside the device byte buffer (EOI excluded)
    uint32_t clean_off;           // where the frame's un-stuffed stream starts in the clean buffer (16-byte aligned)
    int32_t ri, n_int, seg_base, tabset;
    int32_t sub_base, n_sub_cap;  // the frame's slice of the per-subsequence arrays
    uint8_t td[4], ta[4], tq[4];
};

struct Geom {
    int32_t ncomp, mcus_x, mcus_y, blocks_per_mcu;
    int32_t hs[3], vs[3];
    int32_t bx[3], by[3];   // padded block raster of each component
    int32_t blk_off[3];     // first block of the component inside a frame's coefficient / sample buffer
    int32_t blocks_per_frame;
    int32_t height, width;
    int32_t fh, fv;         // chroma up-sampling factors (1 | 2)
    int32_t sub_shift;      // log2 of the subsequence size in bytes
    int32_t f0;             // first frame of the launch: a call's frames are decoded in groups, each on a stream of its own
    uint8_t b_comp[MAX_BLOCKS_MCU], b_dy[MAX_BLOCKS_MCU], b_dx[MAX_BLOCKS_MCU];
};

// diagnostics (pa_mjpeg_debug_counters): shader-clock cycles, 100 MHz wall ticks and symbols of one wave's symbol loop
__device__ unsigned long long g_dbg[16];

// Frame descriptors and table sets, pinned host staging -> HBM, read over the link by the device itself. (hipMemcpyAsync
// of a few KB on a stream that has just been told to wait for another stream's event blocked the calling thread for the
// length of a whole decode, every third call, on ROCm 7.2: measured with PA_MJPEG_TRACE.)
__global__ __launch_bounds__(256) void stage_copy_kernel(const uint32_t* __restrict__ a, uint32_t* __restrict__ da, int na,
                                                         const uint32_t* __restrict__ b, uint32_t* __restrict__ db, int nb) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < na) da[i] = a[i];
    else if (i - na < nb) db[i - na] = b[i - na];
}

// ---- byte un-stuffing + restart markers --------------------------------------------------------------------------------------
//
// The entropy-coded segment of every frame is rewritten once into a CLEAN stream: stuffed zeros (FF 00 -> FF) and the
// RSTm markers are removed, and the clean offset at which every restart interval starts is recorded (seg_start). Bit
// positions in the clean stream are plain arithmetic, which is what lets the decoder below start anywhere.

// per 16 raw bytes [a, a + 16) of the scan [lo, hi): bit j of `keep` = byte a + j survives, of `mark` = byte a + j is the
// 0xFF of an RSTm marker
__device__ __forceinline__ void classify16(const uint8_t* bits, uint32_t a, uint32_t lo, uint32_t hi, uint32_t& keep, uint32_t& mark,
                                           uint32_t (&w)[4]) {
    const uint4 v = *reinterpret_cast<const uint4*>(bits + a);
    w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    const uint32_t prev = a > lo ? bits[a - 1] : 0, next = bits[a + 16];
    keep = 0;
    mark = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t b = (w[j >> 2] >> (8 * (j & 3))) & 0xff;
        const uint32_t bp = j ? (w[(j - 1) >> 2] >> (8 * ((j - 1) & 3))) & 0xff : prev;
        const uint32_t bn = j < 15 ? (w[(j + 1) >> 2] >> (8 * ((j + 1) & 3))) & 0xff : next;
        const uint32_t p = a + j;
        const bool in = p >= lo && p < hi;
        const bool is_mark = b == 0xff && (bn & 0xf8) == 0xd0 && p + 1 < hi;
        const bool drop = (bp == 0xff && p > lo && (b == 0 || (b & 0xf8) == 0xd0)) || is_mark;
        if (in && !drop) keep |= 1u << j;
        if (in && is_mark) mark |= 1u << j;
    }
}

// chunk_cnt[f][c] = (markers, kept bytes) of chunk c of frame f
__global__ __launch_bounds__(256) void unstuff_count_kernel(const uint8_t* __restrict__ bits, const FrameDesc* __restrict__ fd,
                                                            int2* __restrict__ chunk_cnt, int max_chunks, int f0) {
    __shared__ int red[8];
    const int f = blockIdx.y + f0, c = blockIdx.x, tid = threadIdx.x;
    const FrameDesc d = fd[f];
    const uint32_t lo = d.scan_off, hi = d.scan_off + d.scan_len;
    const uint32_t a = (lo & ~15u) + (uint32_t)c * CHUNK + tid * 16;
    int nm = 0, nk = 0;
    if (a < hi) {
        uint32_t keep, mark, w[4];
        classify16(bits, a, lo, hi, keep, mark, w);
        nm = __popc(mark);
        nk = __popc(keep);
    }
    for (int o = 32; o; o >>= 1) {
        nm += __shfl_down(nm, o, 64);
        nk += __shfl_down(nk, o, 64);
    }
    if ((tid & 63) == 0) {
        red[tid >> 6] = nm;
        red[4 + (tid >> 6)] = nk;
    }
    __syncthreads();
    if (tid == 0) chunk_cnt[f * max_chunks + c] = make_int2(red[0] + red[1] + red[2] + red[3], red[4] + red[5] + red[6] + red[7]);
}

__global__ __launch_bounds__(256) void unstuff_write_kernel(const uint8_t* __restrict__ bits, const FrameDesc* __restrict__ fd,
                                                            const int2* __restrict__ chunk_cnt, int max_chunks,
                                                            uint8_t* __restrict__ clean, uint32_t* __restrict__ seg_start,
                                                            uint32_t* __restrict__ clean_len, int32_t* __restrict__ status, int f0) {
    __shared__ int red[16];
    __shared__ int2 scan[256];
    const int f = blockIdx.y + f0, c = blockIdx.x, tid = threadIdx.x;
    const FrameDesc d = fd[f];
    const uint32_t lo = d.scan_off, hi = d.scan_off + d.scan_len;
    if ((lo & ~15u) + (uint32_t)c * CHUNK >= hi && c != 0) return;
    // markers / kept bytes in the chunks before this one, and in the whole frame
    int bm = 0, bk = 0, tm = 0, tk = 0;
    for (int i = tid; i < max_chunks; i += 256) {
        const int2 v = chunk_cnt[f * max_chunks + i];
        tm += v.x;
        tk += v.y;
        if (i < c) {
            bm += v.x;
            bk += v.y;
        }
    }
    for (int o = 32; o; o >>= 1) {
        bm += __shfl_down(bm, o, 64);
        bk += __shfl_down(bk, o, 64);
        tm += __shfl_down(tm, o, 64);
        tk += __shfl_down(tk, o, 64);
    }
    if ((tid & 63) == 0) {
        red[tid >> 6] = bm;
        red[4 + (tid >> 6)] = bk;
        red[8 + (tid >> 6)] = tm;
        red[12 + (tid >> 6)] = tk;
    }
    __syncthreads();
    bm = red[0] + red[1] + red[2] + red[3];
    bk = red[4] + red[5] + red[6] + red[7];
    tm = red[8] + red[9] + red[10] + red[11];
    tk = red[12] + red[13] + red[14] + red[15];
    if (c == 0 && tid == 0) {
        clean_len[f] = (uint32_t)tk;
        seg_start[d.seg_base] = 0;
        if (tm != d.n_int - 1) atomicOr(&status[f], ERR_RST);
    }
    const uint32_t a = (lo & ~15u) + (uint32_t)c * CHUNK + tid * 16;
    uint32_t keep = 0, mark = 0, w[4] = {0, 0, 0, 0};
    if (a < hi) classify16(bits, a, lo, hi, keep, mark, w);
    scan[tid] = make_int2(__popc(mark), __popc(keep));
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {  // inclusive scan of the per-thread counts
        int2 v = make_int2(0, 0);
        if (tid >= o) v = scan[tid - o];
        __syncthreads();
        scan[tid].x += v.x;
        scan[tid].y += v.y;
        __syncthreads();
    }
    int km = bm + scan[tid].x - __popc(mark);    // markers before this thread's bytes
    uint32_t kk = (uint32_t)(bk + scan[tid].y - __popc(keep));  // clean offset of this thread's first kept byte
    // The chunk's kept bytes are gathered in LDS, laid out like the 16-byte lines of the clean stream they go to, and
    // written with one 16-byte store per thread (byte stores only for the two lines shared with the neighbouring
    // chunks): 16 byte-wide stores per thread cost 113 us per 64 frames, this 4x less.
    __shared__ __attribute__((aligned(16))) uint8_t stage[CHUNK + 32];
    const uint32_t g0 = d.clean_off + (uint32_t)bk;  // clean-buffer offset of the chunk's first kept byte
    const uint32_t mis = g0 & 15u;
    uint32_t sl = mis + (kk - (uint32_t)bk);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (mark & (1u << j)) {
            // the interval after this marker starts at the clean offset reached so far
            ++km;
            if (km < d.n_int) seg_start[d.seg_base + km] = kk;
        }
        if (keep & (1u << j)) {
            stage[sl++] = (uint8_t)((w[j >> 2] >> (8 * (j & 3))) & 0xff);
            ++kk;
        }
    }
    __syncthreads();
    const uint32_t total = (uint32_t)scan[255].y;
    const uint32_t nlines = (mis + total + 15) >> 4;
    uint8_t* const line0 = clean + (g0 - mis);
    for (uint32_t q = tid; q < nlines; q += 256) {
        const uint32_t b0 = q == 0 ? mis : 0u;
        const uint32_t b1 = min(16u, mis + total - q * 16);
        if (b0 == 0 && b1 == 16) {
            *reinterpret_cast<uint4*>(line0 + q * 16) = *reinterpret_cast<const uint4*>(stage + q * 16);
        } else {
            for (uint32_t bb = b0; bb < b1; ++bb) line0[q * 16 + bb] = stage[q * 16 + bb];
        }
    }
}

// ---- entropy decoding --------------------------------------------------------------------------------------------------------
//
// One lane per subsequence (128 bytes or more, see below) of a frame's clean stream, wherever it falls (Weissenberger & Schmidt's
// self-synchronising scheme, restated for this layout). A lane's decoding state at a bit position is (block of the MCU,
// zig-zag index); given the right state at its entry a lane decodes exactly the codewords that START inside its
// subsequence and hands (overshoot bits, block, index) to the next lane.
//   pass A (MODE 0): every lane assumes the state of a block start at its first bit. Huffman streams re-synchronise by
//                    themselves, so most exit states are already right;
//   verify (MODE 1): every lane whose entry (= the predecessor's exit) differs from what it last decoded with decodes
//                    again; repeated until no exit state changes = the exact sequential states. Lanes that follow a
//                    restart marker are exact from the start: at a marker the state is known.
//   scan           : per frame, exclusive scan of the blocks completed, with resets at restart markers -> every lane's
//                    absolute block index at entry;
//   final (MODE 2) : decode once more and store the non-zero AC coefficients and the DC differences (int16) by block
//                    number in scan order (buffers cleared beforehand). Only this pass extracts values; the passes
//                    before it need code lengths, run lengths and sizes alone.
// The subsequence size is chosen per call (a power of two, about four MCUs of the stream: states settle within a
// couple of MCUs, so most lanes are right after pass A and one verify pass). Every lane keeps the next 128 bytes of its
// stream in an LDS ring (33-dword pitch: the 64 lanes' window reads fall on distinct banks) that all lanes top up
// together every 16 symbols with 16-byte loads; the 64-bit window is re-read from the ring at every symbol, so there is
// no refill branch inside the symbol loop.

struct SubCnt {     // what a subsequence contributes to the scan
    int32_t blk;    // blocks completed; bit 31: a restart marker lies inside, blk then counts from the frame start
};


template <int MODE, int LANES = WG_SUBS>
__global__ __launch_bounds__(LANES) void sub_decode_kernel(const uint8_t* __restrict__ clean, const FrameDesc* __restrict__ fd,
                                                         const TableSet* __restrict__ ts, const uint32_t* __restrict__ seg_start,
                                                         const uint32_t* __restrict__ clean_len, const Geom g,
                                                         const uint32_t* __restrict__ g_in, uint32_t* __restrict__ g_out,
                                                         uint32_t* __restrict__ used, SubCnt* __restrict__ cnt,
                                                         const SubCnt* __restrict__ entry, int16_t* __restrict__ coef,
                                                         int32_t* __restrict__ status, int32_t* __restrict__ changed,
                                                         const int32_t* __restrict__ changed_last, const int32_t* __restrict__ todo,
                                                         const int32_t* __restrict__ todo_cnt, int16_t* __restrict__ dcdiff) {
    // verify pass: a frame whose previous verify pass changed nothing has settled (changed_last = that pass's flags)
    if (MODE == 1 && changed_last && changed_last[blockIdx.y + g.f0] == 0) return;
    __shared__ HuffTables T;
    __shared__ uint32_t ring[LANES * (RING_DW + 1)];
    const int f = blockIdx.y + g.f0, tid = threadIdx.x;
    const FrameDesc d = fd[f];
    const uint32_t clen = clean_len[f];
    const int sh = g.sub_shift;
    const int nsub = (int)((clen + (1u << sh) - 1) >> sh);
    const int j0 = blockIdx.x * LANES;
    int j = j0 + tid;
    bool active;
    if (MODE == 1) {
        // a verify pass walks the COMPACT list of the frame's lanes whose entry changed (sub_verify_plan_kernel): the few
        // lanes that still move in a late pass fill whole waves instead of keeping one lane busy in every wave
        const int cnt = todo_cnt[f];
        if (j0 >= cnt) return;
        active = j < cnt;
        j = active ? todo[(size_t)d.sub_base + j] : 0;
    } else {
        if (j0 >= nsub) return;
        active = j < nsub && j < d.n_sub_cap;
    }
    if (MODE == 2 && blockIdx.x == 0 && tid == 0 && changed_last[f]) atomicOr(&status[f], ERR_SYNC);
    const size_t sj = (size_t)d.sub_base + j;
    const uint32_t entry_st = (MODE == 0 || j == 0 || !active) ? 0u : g_in[sj - 1];
    const bool mine = active;  // this lane decodes in this pass and records what it found
    {   // tables, block layout of an MCU
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&ts[d.tabset].h);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&T);
        for (int i = tid; i < (int)(sizeof(HuffTables) / 4); i += LANES) dst[i] = src[i];
    }
    // Huffman table of every block position of an MCU (baseline: two DC, two AC tables), one bit each, wave-uniform:
    // bit b = the DC table of block b, bit 16 + b = its AC table
    uint32_t tabsel = 0;
    for (int bb = 0; bb < g.blocks_per_mcu; ++bb) {
        const int c = g.b_comp[bb];
        tabsel |= (uint32_t)(d.td[c] & 1) << bb;
        tabsel |= (uint32_t)(d.ta[c] & 1) << (16 + bb);
    }
    __syncthreads();
    // positions are bits from the start of the frame's clean stream
    const uint32_t s_byte = (uint32_t)j << sh;
    const uint32_t end_bits = min((uint32_t)(j + 1) << sh, clen) * 8;
    uint32_t bitpos = s_byte * 8 + (entry_st & 31);
    int b = (entry_st >> 5) & 15, z = (entry_st >> 9) & 63;
    const int bpm = g.blocks_per_mcu;
    if (b >= bpm) b = 0;
    // this lane's ring: bytes [fill - 128, fill) of the stream, most significant bit first; dword X at ring[X & 31], and
    // ring[32] repeats ring[0] so that a window's two dwords are always neighbours
    uint32_t* const my_ring = ring + tid * (RING_DW + 1);
    const uint8_t* const stream = clean + d.clean_off;
    uint32_t fill = s_byte;  // multiple of 16
    auto top_up = [&]() {
        // keep the dwords (bitpos >> 5) and the one after it, fill the rest of the ring
        while (fill + 16 <= ((bitpos >> 5) << 2) + 4 * RING_DW) {
            const uint4 v = *reinterpret_cast<const uint4*>(stream + fill);
            uint32_t* q = my_ring + ((fill >> 2) & (RING_DW - 1));
            q[0] = __builtin_bswap32(v.x); q[1] = __builtin_bswap32(v.y);
            q[2] = __builtin_bswap32(v.z); q[3] = __builtin_bswap32(v.w);
            if (q == my_ring) my_ring[RING_DW] = q[0];
            fill += 16;
        }
    };
    // first restart boundary at or after this subsequence's first byte
    int nbk = d.n_int;  // index of the next boundary's interval; n_int = none
    uint32_t nb_bits = 0xffffffffu;
    if (active && d.n_int > 1) {
        int lo = 1, hi = d.n_int;  // lower_bound over seg_start[1 .. n_int)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (seg_start[d.seg_base + mid] < s_byte) lo = mid + 1; else hi = mid;
        }
        nbk = lo;
        if (nbk < d.n_int) nb_bits = seg_start[d.seg_base + nbk] * 8;
    }
    int nblk = 0, reset = 0, err = 0;
    int absblk = 0;  // MODE 2: index of the current block in scan order; else: first block of the last restart interval entered
    if (MODE == 2 && active) {
        const SubCnt e = entry[sj];
        absblk = e.blk & 0x7fffffff;
    }
    // final pass: coefficients in SCAN order (block absblk of the frame, zig-zag index inside the block), DC differences apart
    const int total_blocks = g.mcus_x * g.mcus_y * bpm;
    int16_t* const frame_coef = MODE == 2 ? coef + (size_t)f * total_blocks * 64 : nullptr;
    int16_t* const frame_dc = MODE == 2 ? dcdiff + (size_t)f * total_blocks : nullptr;
    // row of T.lut1 / T.lutB for the symbol in hand: DC tables 0-1, AC tables 2-3
    auto table_row = [&](bool dc, int blk) -> uint32_t { return ((tabsel >> (blk + (dc ? 0 : 16))) & 1u) | (dc ? 0u : 2u); };
    if (MODE == 2 && active && absblk % bpm != b) {  // the scan and the synchronised state disagree: corrupt stream
        err |= ERR_SYNC;
        active = false;
    }
    uint32_t lim = min(nb_bits, end_bits);
    // the next 32 bits of the stream, straight from the ring
    auto window = [&]() -> uint32_t {
        const uint32_t* q = my_ring + ((bitpos >> 5) & (RING_DW - 1));
        return (uint32_t)(((((uint64_t)q[0]) << 32) | q[1]) << (bitpos & 31) >> 32);
    };
    const uint16_t* const lut1 = &T.lut1[0][0];
    const uint16_t* const lutB = &T.lutB[0][0];
    // what the lane hands on, captured when it reaches the end of its subsequence (it free-runs after that)
    uint32_t x_state = 0;
    int x_blk = 0;
    bool gen = false;
    // diagnostics: only the first wave of the first frame reads the clocks (s_memtime + a wait in every wave's slow step
    // was 2 % of a pass)
    const bool stamp = __builtin_amdgcn_readfirstlane((int)(blockIdx.x == 0 && f == 0 && tid < 64)) != 0;
    unsigned long long c0 = 0, w0t = 0;
    if (stamp) {
        c0 = clock64();
        w0t = wall_clock64();
    }
    int it = 0;
    unsigned long long slow_cyc = 0;
    int outer = 0;
    while (__ballot(active)) {
        unsigned long long cs = 0;
        if (stamp) cs = clock64();
        ++outer;
        // ---- the slow step: ring top-up, restart markers, end of the subsequence, symbols the fast loop does not take
        if (MODE == 2 && absblk >= total_blocks) active = false;  // what follows the last block is padding
        if (active) {
            top_up();
            if (bitpos >= lim) {
                gen = false;
                if (bitpos >= nb_bits) {
                    // restart marker: byte aligned, block 0 of MCU nbk * ri, predictions zero (T.81 F.2.2.4 / E.2.4)
                    bitpos = nb_bits;
                    b = 0; z = 0;
                    reset = 1;
                    nblk = 0;
                    absblk = nbk * d.ri * bpm;
                    ++nbk;
                    nb_bits = nbk < d.n_int ? seg_start[d.seg_base + nbk] * 8 : 0xffffffffu;
                    lim = min(nb_bits, end_bits);
                }
                if (bitpos >= end_bits) {
                    active = false;
                    x_state = ((bitpos - end_bits) & 31) | ((uint32_t)b << 5) | ((uint32_t)z << 9);
                    // blocks since the last restart marker inside the subsequence (absblk = that marker's block), if any
                    x_blk = reset ? (int32_t)(((uint32_t)(absblk + nblk) & 0x7fffffffu) | 0x80000000u) : nblk;
                }
            } else if (gen) {
                // one symbol the general way: codes outside the look-up tables, padding in front of a marker, errors
                gen = false;
                const uint32_t win = window();
                const bool dc = z == 0;
                const uint32_t t = table_row(dc, b);
                uint32_t e = T.lut1[t][win >> (32 - LB)];
                if (e == 0 && !dc && (win >> 26) == 63) e = T.lutB[t][(win >> (26 - LB)) & ((1 << LB) - 1)];
                int len = e & 31, s = (e >> 5) & 15, adv = e >> 9;
                bool invalid = false;
                if (e == 0) {  // canonical search (T.81 F.2.2.3)
                    const uint32_t pk = win >> 16;
                    len = 17;
                    int sym = 0;
                    for (int l = LB + 1; l <= 16; ++l) {
                        const int code = (int)(pk >> (16 - l));
                        if (code <= T.maxcode[t][l]) {
                            sym = T.vals[t][(T.valoff[t][l] + code) & 255];
                            len = l;
                            break;
                        }
                    }
                    if (len == 17) {
                        // no such code: the 1-bits that pad the byte in front of a restart marker (caught below), a lane
                        // that is out of step (A / verify: move on by one bit), or a corrupt stream (final)
                        invalid = true;
                        len = 1;
                        sym = 0;
                    }
                    s = sym & 15;
                    adv = symbol_advance(dc, sym);
                }
                const int use = len + s;
                const uint32_t np = bitpos + use;
                if (np > nb_bits || (invalid && nb_bits - bitpos < 8)) {
                    bitpos = nb_bits;  // the padding bits in front of a restart marker, not a symbol
                } else if (invalid && MODE == 2) {
                    err |= ERR_HUFF;
                    active = false;
                } else {
                    bitpos = np;
                    int zn = z + adv;
                    const bool over = !dc && s && zn > 64;  // a coefficient beyond index 63
                    if (MODE == 2) {
                        const uint32_t raw = s ? (uint32_t)(win << len) >> (32 - s) : 0u;
                        const int v = (int)raw - ((int)raw < ((1 << s) >> 1) ? (1 << s) - 1 : 0);
                        if (over) {
                            err |= ERR_COEF;
                            active = false;
                        } else if (absblk < total_blocks) {
                            if (dc) {
                                frame_dc[absblk] = (int16_t)v;  // the difference; dc_scan_kernel adds the predictions up
                            } else if (s) {
                                frame_coef[(size_t)absblk * 64 + zn - 1] = (int16_t)v;
                            }
                        }
                    }
                    if (over) zn = 64;
                    if (zn >= 64) {
                        zn = 0;
                        ++nblk;
                        if (MODE == 2) ++absblk;
                        if (++b >= bpm) b = 0;
                    }
                    z = zn;
                }
            }
        }
        // ---- the fast loop: straight-line code, every lane; left as soon as one active lane meets anything else.
        // A step is two LDS round trips -- the window's two dwords from the ring, then the table look-ups side by side
        // -- and some seventy vector instructions; the passes are bound by instruction issue, not by those latencies.
        if (stamp) slow_cyc += clock64() - cs;
#pragma unroll
        for (int k = 0; k < TOPUP; ++k, ++it) {  // (`it` counts steps: a pair is one)
            const uint32_t win = window();
            const bool dc = z == 0;
            const uint32_t tb = table_row(dc, b) << LB;
            uint32_t eA = lut1[tb + (win >> (32 - LB))];
            uint32_t eB = lutB[tb + ((win >> (26 - LB)) & ((1 << LB) - 1))];
            // passes A / verify: the AC table's pair entry for these ten bits (lutB row = AC table number)
            uint32_t eP = MODE != 2 ? lutB[(tb & (1u << LB)) + (win >> (32 - LB))] : 0u;
            asm volatile("" : "+v"(eA), "+v"(eB), "+v"(eP));  // the look-ups in flight together, none behind a branch
            const uint32_t e = eA ? eA : (!dc && (win >> 26) == 63 ? eB : 0u);
            const int len = e & 31, s = (e >> 5) & 15;
            int adv = e >> 9, use = len + s;
            // two symbols in one step where the pair neither ends the block nor reaches the end of the subsequence / a
            // restart boundary (the exit state is taken at the FIRST symbol boundary behind the end)
            // (bitwise, not short-circuit: hipcc turns && chains over lane values into exec-mask branches)
            const int p_use = (int)(eP & 15), p_adv = (int)((eP >> 4) & 63), p_eob = (int)((eP >> 4) & 64);
            const bool pair = (MODE != 2) & !dc & ((eP >> 15) != 0) & (z + p_adv < 64) & (bitpos + (uint32_t)p_use < lim);
            if (MODE != 2) {
                use = pair ? p_use : use;
                adv = pair ? p_adv + p_eob : adv;  // (an EOB behind the coefficients ends the block: index beyond 64)
            }
            const uint32_t np = bitpos + use;
            const int zn = z + adv;  // DC: 1 | coefficient: past its zero run and itself | ZRL: + 16 | EOB: beyond 64
            const bool rare = !pair & ((bitpos >= lim) | (e == 0) | (np > nb_bits) | ((zn > 64) & (adv < 64)));
            if (__builtin_amdgcn_ballot_w64(active && rare) != 0) {
                gen = rare && bitpos < lim;
                break;
            }
            // commit
            bitpos = np;
            if (MODE == 2) {
                // the value (extra bits, T.81 F.2.2.1 EXTEND): nothing before the final pass needs it
                const uint32_t raw = s ? (uint32_t)(win << len) >> ((32 - s) & 31) : 0u;
                const int v = (int)raw - ((int)raw < ((1 << s) >> 1) ? (1 << s) - 1 : 0);
                // one store: the DC difference (dc_scan_kernel adds the predictions up) or a non-zero AC coefficient
                int16_t* const base = dc ? frame_dc : frame_coef;
                const uint32_t off = dc ? (uint32_t)absblk : (uint32_t)absblk * 64u + (uint32_t)(zn - 1);
                if (active && absblk < total_blocks && s) base[off] = (int16_t)v;
            }
            const bool bend = zn >= 64;
            z = bend ? 0 : zn;
            nblk += bend ? 1 : 0;
            if (MODE == 2) absblk += bend ? 1 : 0;
            const int bn = b + 1 == bpm ? 0 : b + 1;
            b = bend ? bn : b;
        }
    }
    if (stamp && tid == 0) {
        g_dbg[MODE * 2] = clock64() - c0;
        g_dbg[MODE * 2 + 1] = ((wall_clock64() - w0t) << 32) | (unsigned)it;
        g_dbg[8 + MODE * 2] = slow_cyc;
        g_dbg[8 + MODE * 2 + 1] = (unsigned)outer;
    }
    if (err) atomicOr(&status[f], err);
    if (MODE != 2 && mine) {
        if (MODE == 1 && x_state != g_in[sj]) atomicOr(&changed[f], 1);
        g_out[sj] = x_state;
        used[sj] = entry_st;
        SubCnt c;
        c.blk = x_blk;
        cnt[sj] = c;
    }
}

// Which lanes of a frame decode again in the next verify pass: those whose entry state (the predecessor's exit) is not the
// one they last decoded with. Their indices are compacted into todo[] (frame order), the others keep their exit state.
__global__ __launch_bounds__(1024) void sub_verify_plan_kernel(const FrameDesc* __restrict__ fd, const uint32_t* __restrict__ clean_len,
                                                               int sub_shift, const uint32_t* __restrict__ g_in, uint32_t* __restrict__ g_out,
                                                               const uint32_t* __restrict__ used, int32_t* __restrict__ todo,
                                                               int32_t* __restrict__ todo_cnt, const int32_t* __restrict__ changed_last, int f0) {
    __shared__ int sh[1024];
    const int f = blockIdx.x + f0, tid = threadIdx.x;
    if (changed_last && changed_last[f] == 0) {  // settled: nothing to do (both state buffers already agree)
        if (tid == 0) todo_cnt[f] = 0;
        return;
    }
    const FrameDesc d = fd[f];
    int nsub = (int)((clean_len[f] + (1u << sub_shift) - 1) >> sub_shift);
    nsub = nsub < d.n_sub_cap ? nsub : d.n_sub_cap;
    const int per = (nsub + 1023) / 1024;
    const int lo = tid * per, hi = min(lo + per, nsub);
    int cnt = 0;
    for (int j = lo; j < hi; ++j) {
        const size_t sj = (size_t)d.sub_base + j;
        const uint32_t entry_st = j == 0 ? 0u : g_in[sj - 1];
        cnt += entry_st != used[sj] ? 1 : 0;
    }
    sh[tid] = cnt;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = tid >= o ? sh[tid - o] : 0;
        __syncthreads();
        sh[tid] += v;
        __syncthreads();
    }
    int k = sh[tid] - cnt;
    for (int j = lo; j < hi; ++j) {
        const size_t sj = (size_t)d.sub_base + j;
        const uint32_t entry_st = j == 0 ? 0u : g_in[sj - 1];
        if (entry_st != used[sj]) todo[(size_t)d.sub_base + k++] = j;
        else g_out[sj] = g_in[sj];
    }
    if (tid == 1023) todo_cnt[f] = sh[1023];
}

// entry[j] = what lanes 0 .. j-1 of the frame accumulated: the absolute block index at lane j's entry
__global__ __launch_bounds__(1024) void sub_scan_kernel(const FrameDesc* __restrict__ fd, const uint32_t* __restrict__ clean_len,
                                                        const SubCnt* __restrict__ cnt, SubCnt* __restrict__ entry, int sub_shift, int f0) {
    __shared__ SubCnt sh[1024];
    const int f = blockIdx.x + f0, tid = threadIdx.x;
    const FrameDesc d = fd[f];
    int nsub = (int)((clean_len[f] + (1u << sub_shift) - 1) >> sub_shift);
    nsub = nsub < d.n_sub_cap ? nsub : d.n_sub_cap;
    const int per = (nsub + 1023) / 1024;
    const int lo = tid * per, hi = min(lo + per, nsub);
    auto combine = [](const SubCnt& a, const SubCnt& b) {  // a then b
        if (b.blk < 0) return b;
        SubCnt r;
        r.blk = a.blk + b.blk;  // keeps a's marker bit: b.blk < 2^30
        return r;
    };
    SubCnt acc = {0};
    for (int i = lo; i < hi; ++i) acc = combine(acc, cnt[(size_t)d.sub_base + i]);
    sh[tid] = acc;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        SubCnt v = {0};
        const bool has = tid >= o;
        if (has) v = sh[tid - o];
        __syncthreads();
        if (has) sh[tid] = combine(v, sh[tid]);
        __syncthreads();
    }
    SubCnt run = {0};
    if (tid > 0) run = sh[tid - 1];
    for (int i = lo; i < hi; ++i) {
        entry[(size_t)d.sub_base + i] = run;
        run = combine(run, cnt[(size_t)d.sub_base + i]);
    }
}

// DC differences (scan order, as the final pass left them) -> DC coefficients, in place: per component a running sum over
// the component's blocks in scan order, starting again from zero at every restart interval (T.81 F.2.1.3.1 / F.2.2.4).
// One workgroup per (frame, component); every thread owns a run of consecutive blocks of the component.
__global__ __launch_bounds__(1024) void dc_scan_kernel(int16_t* __restrict__ dc, const FrameDesc* __restrict__ fd, const Geom g) {
    __shared__ int w_sum[16];
    __shared__ int w_flag[16];
    const int f = blockIdx.x + g.f0, c = blockIdx.y, tid = threadIdx.x;
    if (c >= g.ncomp) return;
    const int bpm = g.blocks_per_mcu;
    int b0 = 0, nbc = 0;  // the component's blocks inside an MCU: b0 .. b0 + nbc
    for (int b = 0; b < bpm; ++b) {
        if (g.b_comp[b] == c) {
            if (nbc == 0) b0 = b;
            ++nbc;
        }
    }
    const int ri = fd[f].ri;
    const int mcus = g.mcus_x * g.mcus_y, n = mcus * nbc;
    int16_t* __restrict__ p = dc + (size_t)f * mcus * bpm + b0;
    const int per = (n + 1023) / 1024;
    const int lo = min(tid * per, n), hi = min(lo + per, n);
    // a thread's run, sixteen blocks at a time: the sixteen loads are in flight together (one after the other they cost a
    // memory latency each, 2 x 32 of them for the luma of a 1080p frame)
    constexpr int DCB = 16;  // loads in flight per thread
    auto walk = [&](int& run, int& flag, bool store) {
        int mcu = lo / nbc, t = lo - mcu * nbc;
        int in_ri = ri > 0 ? mcu % ri : 1;  // MCU's place in its restart interval, kept by counting (a modulo per block was
                                            // most of this kernel's instructions); without restart markers never 0
        for (int k0 = lo; k0 < hi; k0 += DCB) {
            int idx[DCB], val[DCB];
            bool rst[DCB];
#pragma unroll
            for (int i = 0; i < DCB; ++i) {
                idx[i] = mcu * bpm + t;
                rst[i] = t == 0 && in_ri == 0;
                if (++t == nbc) {
                    t = 0;
                    ++mcu;
                    if (ri > 0 && ++in_ri == ri) in_ri = 0;
                }
            }
#pragma unroll
            for (int i = 0; i < DCB; ++i) val[i] = k0 + i < hi ? (int)p[idx[i]] : 0;
#pragma unroll
            for (int i = 0; i < DCB; ++i) {
                if (k0 + i < hi) {
                    if (rst[i]) {
                        run = 0;
                        flag = 1;
                    }
                    run += val[i];
                    if (store) p[idx[i]] = (int16_t)run;
                }
            }
        }
    };
    int run = 0, flag = 0;
    walk(run, flag, false);
    // segmented inclusive scan of the threads' (sum, restart seen): a run with a restart inside forgets what precedes
    // it. Inside a wave by lane shuffles, across the sixteen waves through LDS: one barrier (twenty of them, ten scan steps
    // over 1024 threads, were most of this kernel's 54 us).
    const int lane = tid & 63, wave = tid >> 6;
    int ssum = run, sflag = flag;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int vs = __shfl_up(ssum, o), vf = __shfl_up(sflag, o);
        if (lane >= o && !sflag) {
            ssum += vs;
            sflag = vf;
        }
    }
    if (lane == 63) {
        w_sum[wave] = ssum;
        w_flag[wave] = sflag;
    }
    __syncthreads();
    int psum = 0;  // what the waves before this one leave
    for (int w = 0; w < wave; ++w) psum = w_flag[w] ? w_sum[w] : psum + w_sum[w];
    int esum = __shfl_up(ssum, 1), eflag = __shfl_up(sflag, 1);  // exclusive: the lanes before this one
    if (lane == 0) {
        esum = 0;
        eflag = 0;
    }
    run = eflag ? esum : psum + esum;
    walk(run, flag, true);
}

// ---- de-quantisation + inverse DCT -------------------------------------------------------------------------------------------

// One thread per block of the component rasters. Its coefficients lie where the scan put them: block
// mcu * blocks_per_mcu + (position inside the MCU) of the frame, zig-zag order, the DC coefficient in dc[] (dc_scan_kernel).
__global__ __launch_bounds__(256) void idct_kernel(const int16_t* __restrict__ coef, const int16_t* __restrict__ dc,
                                                   const FrameDesc* __restrict__ fd, const TableSet* __restrict__ ts, const Geom g,
                                                   uint8_t* __restrict__ planes, int n_frames) {
    using namespace dct;
    constexpr int ZZ[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                            41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                            30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)n_frames * g.blocks_per_frame) return;
    const int fl = (int)(t / g.blocks_per_frame), r = (int)(t - (long long)fl * g.blocks_per_frame);
    const int f = fl + g.f0;
    const int c = (g.ncomp > 2 && r >= g.blk_off[2]) ? 2 : ((g.ncomp > 1 && r >= g.blk_off[1]) ? 1 : 0);
    const int off_c = c == 0 ? g.blk_off[0] : (c == 1 ? g.blk_off[1] : g.blk_off[2]);
    const int bx_c = c == 0 ? g.bx[0] : (c == 1 ? g.bx[1] : g.bx[2]);
    const int hs_c = c == 0 ? g.hs[0] : (c == 1 ? g.hs[1] : g.hs[2]);
    const int vs_c = c == 0 ? g.vs[0] : (c == 1 ? g.vs[1] : g.vs[2]);
    const int b0_c = c == 0 ? 0 : (c == 1 ? g.hs[0] * g.vs[0] : g.hs[0] * g.vs[0] + g.hs[1] * g.vs[1]);
    const int rb = r - off_c;
    const int by = rb / bx_c, bx = rb - by * bx_c;
    const int my = by / vs_c, mx = bx / hs_c;
    const size_t sblk = (size_t)f * g.blocks_per_frame + (size_t)(my * g.mcus_x + mx) * g.blocks_per_mcu + b0_c + (by - my * vs_c) * hs_c +
                        (bx - mx * hs_c);  // blocks_per_frame = mcus * blocks_per_mcu: the rasters are padded to whole MCUs
    const uint16_t* __restrict__ q = ts[fd[f].tabset].q[fd[f].tq[c]];
    const uint4* src = reinterpret_cast<const uint4*>(coef + sblk * 64);
    int d[64];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint4 v = src[i];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            d[ZZ[i * 8 + 2 * j]] = (int)(int16_t)(w[j] & 0xffff) * (int)q[ZZ[i * 8 + 2 * j]];
            d[ZZ[i * 8 + 2 * j + 1]] = (int)(int16_t)(w[j] >> 16) * (int)q[ZZ[i * 8 + 2 * j + 1]];
        }
    }
    d[0] = (int)dc[sblk] * (int)q[0];
#pragma unroll
    for (int x = 0; x < 8; ++x) idct8<true>(d + x, 8);
#pragma unroll
    for (int y = 0; y < 8; ++y) idct8<false>(d + y * 8, 1);
    const int pitch = bx_c * 8;
    uint8_t* dst = planes + ((size_t)f * g.blocks_per_frame + off_c) * 64 + (size_t)(by * 8) * pitch + bx * 8;
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            lo |= (uint32_t)clamp255(d[y * 8 + x] + 128) << (8 * x);
            hi |= (uint32_t)clamp255(d[y * 8 + 4 + x] + 128) << (8 * x);
        }
        *reinterpret_cast<uint2*>(dst + (size_t)y * pitch) = make_uint2(lo, hi);
    }
}

// ---- up-sampling + colour conversion -----------------------------------------------------------------------------------------

// Chroma samples of 8 pixels x FV rows (pixel x0 .., rows y0 ..) by jdsample.c's fancy triangle filters. cw x ch = the
// component's real (down-sampled) size: libjpeg replicates ITS last row / column, not the padding of the block raster.
// With 2:1 horizontal sampling a thread needs six samples of a chroma row: four are one aligned dword, the outer two
// are the neighbouring lanes' (the lanes of a wave walk along the row), fetched from memory only at the ends of the
// wave and where the row's last column must be replicated.
__device__ __forceinline__ void chroma_row6(const uint8_t* __restrict__ row, int cx0, int cw, int pc, int lane, int (&s)[6]) {
    const uint32_t d = *reinterpret_cast<const uint32_t*>(row + min(cx0, pc - 4));
    const uint32_t l = (uint32_t)__shfl_up((int)d, 1), r = (uint32_t)__shfl_down((int)d, 1);
    if (cx0 + 5 > cw) {  // the row ends here: clamp every index
#pragma unroll
        for (int i = 0; i < 6; ++i) s[i] = row[min(max(cx0 - 1 + i, 0), cw - 1)];
        return;
    }
    s[1] = d & 0xff; s[2] = (d >> 8) & 0xff; s[3] = (d >> 16) & 0xff; s[4] = d >> 24;
    s[0] = lane == 0 ? (int)row[max(cx0 - 1, 0)] : (int)(l >> 24);
    s[5] = lane == 63 ? (int)row[cx0 + 4] : (int)(r & 0xff);
}

// jdsample.c h2v2_fancy_upsample for the two output rows of chroma row s0 (sa = the row above, sb = the row below, both
// already clamped to the component): 8 output samples each
__device__ __forceinline__ void h2v2_rows(const int (&s0)[6], const int (&sa)[6], const int (&sb)[6], int cx0, int cw, int (&o0)[8],
                                          int (&o1)[8]) {
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        int col[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) col[i] = 3 * s0[i] + (v ? sb[i] : sa[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cx = cx0 + i;
            const int e = cx == 0 ? (col[i + 1] * 4 + 8) >> 4 : (col[i + 1] * 3 + col[i] + 8) >> 4;
            const int od = cx >= cw - 1 ? (col[i + 1] * 4 + 7) >> 4 : (col[i + 1] * 3 + col[i + 2] + 7) >> 4;
            if (v) {
                o1[2 * i] = e;
                o1[2 * i + 1] = od;
            } else {
                o0[2 * i] = e;
                o0[2 * i + 1] = od;
            }
        }
    }
}

template <int FH, int FV>
__device__ __forceinline__ void chroma8(const uint8_t* __restrict__ C, int pc, int cw, int ch, int x0, int y0, int lane, int (&o)[FV][8]) {
    if (FH == 1 && FV == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) o[0][i] = C[(size_t)y0 * pc + min(x0 + i, cw - 1)];
    } else if (FH == 2 && FV == 1) {
        const int cx0 = x0 >> 1;
        int s[6];
        chroma_row6(C + (size_t)y0 * pc, cx0, cw, pc, lane, s);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cx = cx0 + i;
            o[0][2 * i] = cx == 0 ? s[i + 1] : (3 * s[i + 1] + s[i] + 1) >> 2;
            o[0][2 * i + 1] = cx >= cw - 1 ? s[i + 1] : (3 * s[i + 1] + s[i + 2] + 2) >> 2;
        }
    } else {
        const int cx0 = x0 >> 1, cy = y0 >> 1;
        const int ya = max(cy - 1, 0), yb = min(cy + 1, ch - 1);
        int s0[6], sa[6], sb[6];
        chroma_row6(C + (size_t)cy * pc, cx0, cw, pc, lane, s0);
        chroma_row6(C + (size_t)ya * pc, cx0, cw, pc, lane, sa);
        chroma_row6(C + (size_t)yb * pc, cx0, cw, pc, lane, sb);
        h2v2_rows(s0, sa, sb, cx0, cw, o[0], o[FV - 1]);
    }
}

// One output row of 8 pixels: jdcolor.c's YCbCr -> RGB on the up-sampled chroma, three 8-byte stores.
__device__ __forceinline__ void emit_row(const uint8_t* __restrict__ Y, int py, const Geom& g, int f, int y, int x0, bool colour,
                                         const int (&cb)[8], const int (&cr)[8], uint8_t* __restrict__ out, int rgb) {
    using namespace dct;
    if (y >= g.height) return;
    const uint2 yv = *reinterpret_cast<const uint2*>(Y + (size_t)y * py + x0);
    uint32_t px[24];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int yy = (int)(((i < 4 ? yv.x : yv.y) >> (8 * (i & 3))) & 0xff);
        int r = yy, gg = yy, bb = yy;
        if (colour) {
            const int xb = cb[i] - 128, xr = cr[i] - 128;
            // (24-bit multiplies: |x| <= 128 and the constants are below 2^17; the 32-bit multiply runs at a quarter of the rate)
            r = clamp255(yy + ((__mul24(91881, xr) + 32768) >> 16));
            gg = clamp255(yy + ((__mul24(-22554, xb) + 32768 + __mul24(-46802, xr)) >> 16));
            bb = clamp255(yy + ((__mul24(116130, xb) + 32768) >> 16));
        }
        px[3 * i] = (uint32_t)(rgb ? r : bb);
        px[3 * i + 1] = (uint32_t)gg;
        px[3 * i + 2] = (uint32_t)(rgb ? bb : r);
    }
    uint8_t* o = out + (((size_t)f * g.height + y) * g.width + x0) * 3;
    if ((g.width & 7) == 0) {  // whole groups of 8 pixels; row starts and x0 * 3 are multiples of 8 bytes
        uint32_t w[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) w[k] = px[4 * k] | (px[4 * k + 1] << 8) | (px[4 * k + 2] << 16) | (px[4 * k + 3] << 24);
        uint2* o2 = reinterpret_cast<uint2*>(o);
        o2[0] = make_uint2(w[0], w[1]);
        o2[1] = make_uint2(w[2], w[3]);
        o2[2] = make_uint2(w[4], w[5]);
    } else {
#pragma unroll
        for (int k = 0; k < 24; ++k)
            if (x0 + k / 3 < g.width) o[k] = (uint8_t)px[k];
    }
}

// 8 pixels x FV rows per thread: up-sampling, jdcolor.c's YCbCr -> RGB, three 8-byte stores per row. The 64 lanes of a
// wave are 64 neighbouring groups of one row pair; lanes past the right or bottom edge stay in step (clamped addresses,
// nothing stored): their neighbours take samples from them.
template <int FH, int FV>
__global__ __launch_bounds__(256) void ycc_kernel(const uint8_t* __restrict__ planes, const Geom g, uint8_t* __restrict__ out, int rgb) {
    using namespace dct;
    const int f = blockIdx.z + g.f0, lane = threadIdx.x & 63;
    const int x0r = (blockIdx.x * 64 + lane) * 8, y0r = (blockIdx.y * 4 + (threadIdx.x >> 6)) * FV;
    const bool valid = x0r < g.width && y0r < g.height;
    if (y0r >= g.height) return;  // whole waves
    const int x0 = min(x0r, g.bx[0] * 8 - 8), y0 = y0r;
    const uint8_t* fp = planes + (size_t)f * g.blocks_per_frame * 64;
    const int py = g.bx[0] * 8;
    const uint8_t* Y = fp + (size_t)g.blk_off[0] * 64;
    int cb[FV][8], cr[FV][8];
    const bool colour = g.ncomp == 3;
    if (colour) {
        const int pc = g.bx[1] * 8;
        const int cw = (g.width + FH - 1) / FH, ch = (g.height + FV - 1) / FV;  // jdmaster.c: downsampled_width / _height
        chroma8<FH, FV>(fp + (size_t)g.blk_off[1] * 64, pc, cw, ch, x0, y0, lane, cb);
        chroma8<FH, FV>(fp + (size_t)g.blk_off[2] * 64, pc, cw, ch, x0, y0, lane, cr);
    }
    if (!valid) return;
#pragma unroll
    for (int v = 0; v < FV; ++v) emit_row(Y, py, g, f, y0 + v, x0, colour, cb[v], cr[v], out, rgb);
}

// 4:2:0, the path's own sampling: FOUR output rows (two chroma rows) per thread. The second pair's chroma rows are the
// first pair's shifted by one, so a thread fetches 4 + 4 chroma rows for 4 output rows instead of 6 + 6, and a wave lives
// twice as long behind one round of memory latency.
__global__ __launch_bounds__(256) void ycc420_kernel(const uint8_t* __restrict__ planes, const Geom g, uint8_t* __restrict__ out, int rgb) {
    const int f = blockIdx.z + g.f0, lane = threadIdx.x & 63;
    const int x0r = (blockIdx.x * 64 + lane) * 8, y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 4;
    if (y0 >= g.height) return;  // whole waves
    const bool valid = x0r < g.width;
    const int x0 = min(x0r, g.bx[0] * 8 - 8);
    const uint8_t* fp = planes + (size_t)f * g.blocks_per_frame * 64;
    const int py = g.bx[0] * 8, pc = g.bx[1] * 8;
    const uint8_t* Y = fp + (size_t)g.blk_off[0] * 64;
    const int cw = (g.width + 1) / 2, ch = (g.height + 1) / 2;
    const int cx0 = x0 >> 1, cy = y0 >> 1;
    const int r_m1 = max(cy - 1, 0), r_1 = min(cy + 1, ch - 1), r_2 = min(cy + 2, ch - 1);
    const bool second = y0 + 2 < g.height;  // (then cy + 1 <= ch - 1)
    int cb[4][8], cr[4][8];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const uint8_t* C = fp + (size_t)(c ? g.blk_off[2] : g.blk_off[1]) * 64;
        int a[6], b[6], d[6], e[6];
        chroma_row6(C + (size_t)r_m1 * pc, cx0, cw, pc, lane, a);
        chroma_row6(C + (size_t)cy * pc, cx0, cw, pc, lane, b);
        chroma_row6(C + (size_t)r_1 * pc, cx0, cw, pc, lane, d);
        chroma_row6(C + (size_t)r_2 * pc, cx0, cw, pc, lane, e);
        if (c) {
            h2v2_rows(b, a, d, cx0, cw, cr[0], cr[1]);
            h2v2_rows(d, b, e, cx0, cw, cr[2], cr[3]);
        } else {
            h2v2_rows(b, a, d, cx0, cw, cb[0], cb[1]);
            h2v2_rows(d, b, e, cx0, cw, cb[2], cb[3]);
        }
    }
    if (!valid) return;
#pragma unroll
    for (int v = 0; v < 4; ++v)
        if (v < 2 || second) emit_row(Y, py, g, f, y0 + v, x0, true, cb[v], cr[v], out, rgb);
}

// ---- host: marker segments ---------------------------------------------------------------------------------------------------

struct Parsed {
    int height = 0, width = 0, ncomp = 0;
    int cid[3] = {0, 0, 0}, h[3] = {1, 1, 1}, v[3] = {1, 1, 1}, tq[3] = {0, 0, 0}, td[3] = {0, 0, 0}, ta[3] = {0, 0, 0};
    uint16_t q[4][64];
    uint8_t counts[4][16];  // DC0 DC1 AC0 AC1
    uint8_t syms[4][256];
    bool qdef[4] = {false, false, false, false}, hdef[4] = {false, false, false, false};
    int ri = 0;
    size_t scan_off = 0;
};

const uint8_t h_zigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                              41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                              30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// T.81 Annex B. Returns nullptr or what is wrong with the file.
const char* parse_header(const uint8_t* d, size_t n, Parsed& P) {
    if (n < 4 || d[0] != 0xff || d[1] != 0xd8) return "no SOI marker";
    size_t pos = 2;
    bool sof = false;
    P = Parsed();  // nothing carries over from the previous file (tables, restart interval)
    memset(P.q, 0, sizeof P.q);
    memset(P.counts, 0, sizeof P.counts);
    memset(P.syms, 0, sizeof P.syms);
    for (;;) {
        if (pos + 4 > n) return "truncated before SOS";
        if (d[pos] != 0xff) return "marker expected";
        const int m = d[pos + 1];
        if (m == 0xff) {
            ++pos;
            continue;
        }
        const size_t seg = ((size_t)d[pos + 2] << 8) | d[pos + 3];
        if (seg < 2 || pos + 2 + seg > n) return "truncated segment";
        const uint8_t* b = d + pos + 4;
        const size_t len = seg - 2;
        if (m == 0xdb) {
            size_t i = 0;
            while (i < len) {
                const int pq = b[i] >> 4, tq = b[i] & 15;
                if (pq != 0) return "16-bit quantisation table (not baseline)";
                if (tq > 3 || i + 65 > len) return "bad DQT";
                for (int k = 0; k < 64; ++k) P.q[tq][h_zigzag[k]] = b[i + 1 + k];
                P.qdef[tq] = true;
                i += 65;
            }
        } else if (m == 0xc0 || m == 0xc1) {
            if (len < 6 || b[0] != 8) return "only 8-bit samples";
            P.height = (b[1] << 8) | b[2];
            P.width = (b[3] << 8) | b[4];
            P.ncomp = b[5];
            if (P.ncomp != 1 && P.ncomp != 3) return "1 or 3 components expected";
            if (len < (size_t)(6 + 3 * P.ncomp)) return "bad SOF";
            for (int c = 0; c < P.ncomp; ++c) {
                P.cid[c] = b[6 + 3 * c];
                P.h[c] = b[7 + 3 * c] >> 4;
                P.v[c] = b[7 + 3 * c] & 15;
                P.tq[c] = b[8 + 3 * c];
                if (P.tq[c] > 3) return "bad quantisation table selector";
            }
            sof = true;
        } else if (m >= 0xc2 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc) {
            return "progressive / lossless / arithmetic-coded JPEG (only baseline Huffman is decoded)";
        } else if (m == 0xc4) {
            size_t i = 0;
            while (i < len) {
                if (i + 17 > len) return "bad DHT";
                const int tc = b[i] >> 4, th = b[i] & 15;
                if (tc > 1 || th > 1) return "Huffman table id beyond the baseline's 0 / 1";
                int ns = 0;
                for (int k = 0; k < 16; ++k) ns += b[i + 1 + k];
                if (ns > 256 || i + 17 + ns > len) return "bad DHT";
                const int t = tc * 2 + th;
                memcpy(P.counts[t], b + i + 1, 16);
                memset(P.syms[t], 0, 256);
                memcpy(P.syms[t], b + i + 17, ns);
                P.hdef[t] = true;
                i += 17 + ns;
            }
        } else if (m == 0xdd) {
            if (len < 2) return "bad DRI";
            P.ri = (b[0] << 8) | b[1];
        } else if (m == 0xda) {
            if (!sof) return "SOS before SOF";
            if (len < 1 || b[0] != P.ncomp || len < (size_t)(4 + 2 * P.ncomp)) return "multi-scan file (one interleaved scan expected)";
            for (int c = 0; c < P.ncomp; ++c) {
                if (b[1 + 2 * c] != P.cid[c]) return "scan component order differs from the frame header";
                P.td[c] = b[2 + 2 * c] >> 4;
                P.ta[c] = b[2 + 2 * c] & 15;
                if (P.td[c] > 1 || P.ta[c] > 1) return "Huffman table selector beyond the baseline's 0 / 1";
                if (!P.hdef[P.td[c]] || !P.hdef[2 + P.ta[c]]) return "scan uses a Huffman table the file does not define";
                if (!P.qdef[P.tq[c]]) return "frame uses a quantisation table the file does not define";
            }
            if (b[1 + 2 * P.ncomp] != 0 || b[2 + 2 * P.ncomp] != 63) return "spectral selection in a baseline scan";
            P.scan_off = pos + 2 + seg;
            return nullptr;
        } else if (m == 0xd9) {
            return "EOI before SOS";
        }
        pos += 2 + seg;
    }
}

// Table t (0 DC0, 1 DC1, 2 AC0, 3 AC1) of a set from a DHT segment's counts / symbols.
void build_hufftab(HuffTables& T, int t, const uint8_t* counts, const uint8_t* syms) {
    const bool dc = t < 2;
    auto entry = [&](int len, int sym) -> uint16_t {
        return (uint16_t)(len | ((sym & 15) << 5) | (symbol_advance(dc, sym) << 9));
    };
    memset(T.lut1[t], 0, sizeof T.lut1[t]);
    memset(T.lutB[t], 0, sizeof T.lutB[t]);
    memcpy(T.vals[t], syms, 256);
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        T.valoff[t][l] = k - code;
        for (int i = 0; i < counts[l - 1]; ++i, ++k, ++code) {
            if (code >= (1 << l)) continue;  // over-subscribed table: left to the canonical search (and its error)
            if (l <= LB) {
                const int lo = code << (LB - l), hi = (code + 1) << (LB - l);
                for (int e = lo; e < hi; ++e) T.lut1[t][e] = entry(l, syms[k]);
            } else if (!dc && (code >> (l - 6)) == 63) {
                // six leading 1-bits: found under bits 6 .. 6 + LB - 1 of the window (AC tables only: rows 0 / 1 of
                // lutB hold the AC tables' PAIR entries, see below; a DC code of more than LB bits -- a difference
                // of 1024 or more -- takes the canonical search)
                const int rest = l - 6;  // <= 10 bits after them
                const int lo = (code & ((1 << rest) - 1)) << (LB - rest), hi = lo + (1 << (LB - rest));
                for (int e = lo; e < hi; ++e) T.lutB[t][e] = entry(l, syms[k]);
            }
        }
        T.maxcode[t][l] = counts[l - 1] ? code - 1 : -1;
        code <<= 1;
    }
    T.maxcode[t][17] = 0x7fffffff;
    T.maxcode[t][0] = -1;
    T.valoff[t][0] = 0;
    if (!dc) {
        // PAIR entries of AC table t - 2 (lutB row t - 2): where the first LB bits of the window hold TWO whole symbols
        // (code + extra bits each) -- a coefficient or ZRL, then a coefficient, ZRL or EOB -- the entry gives what they
        // consume and how far they move the zig-zag index together: bit 15 set, bits 0-3 total bits (2 .. 10), bits 4-9
        // the advance of the coefficients / ZRLs (1 .. 32), bit 10: the second symbol is EOB. The passes that only look
        // for the decoder state (A, verify) take such a pair in one step.
        uint16_t* pair = T.lutB[t - 2];
        for (int w = 0; w < (1 << LB); ++w) {
            pair[w] = 0;
            const uint16_t e1 = T.lut1[t][w];
            if (!e1) continue;
            const int use1 = (e1 & 31) + ((e1 >> 5) & 15), adv1 = e1 >> 9;
            if (use1 >= LB || adv1 >= 64) continue;
            const uint16_t e2 = T.lut1[t][(w << use1) & ((1 << LB) - 1)];
            if (!e2) continue;
            const int use2 = (e2 & 31) + ((e2 >> 5) & 15), adv2 = e2 >> 9;
            if (use1 + use2 > LB) continue;
            pair[w] = adv2 >= 64 ? (uint16_t)(0x8000 | 0x400 | (use1 + use2) | (adv1 << 4))
                                 : (uint16_t)(0x8000 | (use1 + use2) | ((adv1 + adv2) << 4));
        }
    }
}

}  // namespace mj
}  // namespace pa

using namespace pa::mj;

namespace {
constexpr int MAX_ROUNDS = 16;  // verify passes enqueued without looking at the result (pa_mjpeg_set_sync_rounds)
constexpr int MAX_GROUPS = 4;   // frame groups of a call, each on its own stream (pa_mjpeg_set_groups)
}

struct pa_mjpeg {
    int device = 0, max_frames = 0, max_h = 0, max_w = 0;
    size_t max_bytes = 0;
    size_t max_blocks = 0;      // per frame
    size_t max_subs = 0, max_segs = 0;
    size_t clean_bytes = 0;     // size of a set's clean stream
    int max_chunks_cap = 0;
    int sync_rounds = 8;
    int sub_shift_override = 0;  // tuning: log2 of the subsequence size, 0 = chosen from the stream
    int wg_lanes = WG_SUBS;      // lanes per workgroup of the entropy passes: 256, or 128 (34.8 KB of LDS instead of 51.7: fits beside two
                                 // workgroups of the fp32 convolution kernel on a CU; PA_MJPEG_WG_LANES)
    // Device scratch, TWO sets used in turn (like the pinned staging below): a call never touches the memory of the call
    // before it, so its first groups start while that call's last groups are still decoding.
    struct Set {
        uint8_t* d_bits = nullptr;
        uint8_t* d_clean = nullptr;
        FrameDesc* d_fd = nullptr;
        TableSet* d_ts = nullptr;
        int2* d_chunk = nullptr;
        uint32_t* d_seg = nullptr;
        uint32_t* d_clean_len = nullptr;
        uint32_t* d_g[2] = {nullptr, nullptr};
        uint32_t* d_used = nullptr;
        SubCnt* d_cnt = nullptr;
        SubCnt* d_entry = nullptr;
        int32_t* d_changed = nullptr;  // [MAX_ROUNDS + 1][max_frames]
        int32_t* d_todo = nullptr;     // [max_subs] compact lane lists of a verify pass
        int32_t* d_todo_cnt = nullptr; // [max_frames]
        int16_t* d_coef = nullptr;
        int16_t* d_dc = nullptr;       // [max_frames][blocks of a frame in scan order] DC differences
        uint8_t* d_planes = nullptr;
        int32_t* d_status = nullptr;
        hipEvent_t done[MAX_GROUPS] = {};  // group g of the set's last call has finished
        bool used = false;
    } set[2];
    // pinned host staging, two sets used in turn
    FrameDesc* h_fd[2] = {nullptr, nullptr};
    TableSet* h_ts[2] = {nullptr, nullptr};
    int32_t* h_flag = nullptr;
    hipEvent_t staged[2] = {nullptr, nullptr};
    // A call's frames are decoded in GROUPS, each on a stream of the handle's own: the entropy passes of a group are bound
    // by instruction issue and by the latency of single waves (late verify passes keep a handful of waves busy), so groups
    // whose phases are out of step fill each other's idle time, and the upload of one group runs under the passes of the
    // group before it. The groups' uploads go one after the other (they share the link anyway), which is what puts their
    // phases out of step. Measured on 64 x 1080p quality-95 frames per call (scripts/mjpeg_rate.py): 16.4 k frames/s with
    // one group (everything on the caller's stream), 22.5 k with two, 20.9 k with three. A caller that keeps several
    // DECODERS busy on streams of its own (bench.py's decode_inclusive: three) gets the same effect at the size of whole
    // calls and should leave each at one group: 28.1 k with three decoders x one group, 22.3 k with three x two.
    hipStream_t gstream[MAX_GROUPS] = {};
    // The uploads have a stream to themselves that never waits for another stream: hipMemcpyAsync on a stream with a
    // pending cross-stream wait blocks the CALLING THREAD until the wait is over (ROCm 7.2, seen with PA_MJPEG_TRACE).
    // What they overwrite -- the set's bytes of two calls ago -- is known to be dead on the host: see `done`.
    hipStream_t copy_stream = nullptr;
    hipEvent_t fork = nullptr, prologue = nullptr, up[MAX_GROUPS] = {};
    int groups = 2;
    bool staged_used[2] = {false, false};
    int turn = 0;
    int last_rounds = 0;
    std::string last_error;
};


extern "C" {

const char* pa_mjpeg_last_error(const pa_mjpeg* h) { return h ? h->last_error.c_str() : "null handle"; }

void pa_mjpeg_destroy(pa_mjpeg* h) {
    if (!h) return;
    for (auto& S : h->set) {
        void* dev[] = {S.d_bits, S.d_clean, S.d_fd, S.d_ts, S.d_chunk, S.d_seg, S.d_clean_len, S.d_g[0], S.d_g[1], S.d_used,
                       S.d_cnt, S.d_entry, S.d_changed, S.d_todo, S.d_todo_cnt, S.d_coef, S.d_dc, S.d_planes, S.d_status};
        for (void* p : dev) (void)hipFree(p);
        for (hipEvent_t e : S.done)
            if (e) (void)hipEventDestroy(e);
    }
    for (int i = 0; i < 2; ++i) {
        if (h->h_fd[i]) (void)hipHostFree(h->h_fd[i]);
        if (h->h_ts[i]) (void)hipHostFree(h->h_ts[i]);
        if (h->staged[i]) (void)hipEventDestroy(h->staged[i]);
    }
    if (h->h_flag) (void)hipHostFree(h->h_flag);
    for (int g = 0; g < MAX_GROUPS; ++g) {
        if (h->gstream[g]) {
            (void)hipStreamSynchronize(h->gstream[g]);
            (void)hipStreamDestroy(h->gstream[g]);
        }
        if (h->up[g]) (void)hipEventDestroy(h->up[g]);
    }
    if (h->copy_stream) {
        (void)hipStreamSynchronize(h->copy_stream);
        (void)hipStreamDestroy(h->copy_stream);
    }
    if (h->fork) (void)hipEventDestroy(h->fork);
    if (h->prologue) (void)hipEventDestroy(h->prologue);
    delete h;
}

int pa_mjpeg_create(int32_t device, int32_t max_frames, int32_t max_height, int32_t max_width, size_t max_bytes, pa_mjpeg** out) {
    if (!out) return PA_ERR_INVALID_ARG;
    *out = nullptr;
    if (max_frames < 1 || max_height < 1 || max_width < 1 || max_height > 65535 || max_width > 65535 || max_bytes < 1024 ||
        max_bytes > 0xe0000000ull)
        return PA_ERR_INVALID_ARG;
    pa_mjpeg* h = new pa_mjpeg();
    *out = h;  // handed back on failure too (pa_mjpeg_last_error, then pa_mjpeg_destroy)
    h->device = device; h->max_frames = max_frames; h->max_h = max_height; h->max_w = max_width; h->max_bytes = max_bytes;
    if (const char* e = getenv("PA_MJPEG_SUB_SHIFT")) h->sub_shift_override = atoi(e);  // tuning knob (scripts/mjpeg_rate.py)
    auto chk = [&](hipError_t e, const char* what) -> bool {
        if (e == hipSuccess) return true;
        h->last_error = std::string(what) + ": " + hipGetErrorString(e);
        return false;
    };
    if (!chk(hipSetDevice(device), "hipSetDevice")) return PA_ERR_NO_DEVICE;
    // worst case: three full-resolution components, padded to 16-pixel MCUs
    const size_t bw = ((size_t)max_width + 15) / 16 * 2, bh = ((size_t)max_height + 15) / 16 * 2;
    const size_t n = (size_t)max_frames;
    h->max_blocks = 3 * bw * bh;
    h->max_chunks_cap = (int)(max_bytes / CHUNK) + 3;
    h->max_subs = max_bytes / SUB_MIN + 2 * n + 2;
    h->max_segs = n * bw * bh + n;
    const size_t clean_bytes = max_bytes + 32 * n + 4096;
    h->clean_bytes = clean_bytes;
    if (const char* e = getenv("PA_MJPEG_GROUPS")) h->groups = atoi(e);  // tuning knob (scripts/mjpeg_rate.py)
    if (const char* e = getenv("PA_MJPEG_WG_LANES")) h->wg_lanes = atoi(e) == 128 ? 128 : WG_SUBS;
    h->groups = h->groups < 1 ? 1 : (h->groups > MAX_GROUPS ? MAX_GROUPS : h->groups);
    for (auto& S : h->set) {
        if (!chk(hipMalloc(&S.d_bits, max_bytes + 64), "hipMalloc bitstream")) return PA_ERR_HIP;
        if (!chk(hipMalloc(&S.d_clean, clean_bytes), "hipMalloc clean stream")) return PA_ERR_HIP;
        if (!chk(hipMalloc(&S.d_fd, n * sizeof(FrameDesc)), "hipMalloc descriptors")) return PA_ERR_HIP;
        if (!chk(hipMalloc(&S.d_ts, n * sizeof(TableSet)), "hipMalloc tables")) return PA_ERR_HIP;
        if (!chk(hipMalloc(&S.d_chunk, (n * h->max_chunks_cap) * sizeof(int2)), "hipMalloc chunk counts")) return PA_ERR_HIP;
        if (!chk(hipMalloc(&S.d_seg, h->max_segs * sizeof(uint32_t)), "hipMalloc restart positions")) return PA_ERR_HIP;
        if (!chk(hipMalloc(&S.d_clean_len, n * sizeof(uint32_t)), "hipMalloc clean lengths")) return PA_ERR_HIP;
        for (int i = 0; i < 2; ++i)
            if (!chk(hipMalloc(&S.d_g[i], h->max_subs * sizeof(uint32_t)), "hipMalloc states")) return PA_ERR_HIP;
        if (!chk(hipMalloc(&S.d_used, h->max_subs * sizeof(uint32_t)), "hipMalloc states")) return PA_ERR_HIP;
        if (!chk(hipMalloc(&S.d_cnt, h->max_subs * sizeof(SubCnt)), "hipMalloc counts")) return PA_ERR_HIP;
        if (!chk(hipMalloc(&S.d_entry, h->max_subs * sizeof(SubCnt)), "hipMalloc entries")) return PA_ERR_HIP;
        if (!chk(hipMalloc(&S.d_changed, (MAX_ROUNDS + 1) * n * sizeof(int32_t)), "hipMalloc flags")) return PA_ERR_HIP;
        if (!chk(hipMalloc(&S.d_todo, h->max_subs * sizeof(int32_t)), "hipMalloc lane lists")) return PA_ERR_HIP;
        if (!chk(hipMalloc(&S.d_todo_cnt, n * sizeof(int32_t)), "hipMalloc lane counts")) return PA_ERR_HIP;
        if (!chk(hipMalloc(&S.d_coef, n * h->max_blocks * 64 * sizeof(int16_t)), "hipMalloc coefficients")) return PA_ERR_HIP;
        if (!chk(hipMalloc(&S.d_dc, n * h->max_blocks * sizeof(int16_t)), "hipMalloc DC differences")) return PA_ERR_HIP;
        if (!chk(hipMalloc(&S.d_planes, n * h->max_blocks * 64), "hipMalloc sample planes")) return PA_ERR_HIP;
        if (!chk(hipMalloc(&S.d_status, n * sizeof(int32_t)), "hipMalloc status")) return PA_ERR_HIP;
        if (!chk(hipMemset(S.d_bits, 0, max_bytes + 64), "hipMemset")) return PA_ERR_HIP;
        if (!chk(hipMemset(S.d_clean, 0, clean_bytes), "hipMemset")) return PA_ERR_HIP;
        for (int g = 0; g < MAX_GROUPS; ++g)
            if (!chk(hipEventCreateWithFlags(&S.done[g], hipEventDisableTiming), "hipEventCreate")) return PA_ERR_HIP;
    }
    for (int i = 0; i < 2; ++i) {
        if (!chk(hipHostMalloc(&h->h_fd[i], n * sizeof(FrameDesc)), "hipHostMalloc")) return PA_ERR_HIP;
        if (!chk(hipHostMalloc(&h->h_ts[i], n * sizeof(TableSet)), "hipHostMalloc")) return PA_ERR_HIP;
        if (!chk(hipEventCreateWithFlags(&h->staged[i], hipEventDisableTiming), "hipEventCreate")) return PA_ERR_HIP;
    }
    // A/B knob PA_MJPEG_PRIO=1: the groups' streams at the device's greatest priority (the entropy passes are chains of
    // short, thinly occupied kernels; beside a detector that fills every CU they wait for a slot at each link of the chain)
    int prio = 0;
    if (const char* e = std::getenv("PA_MJPEG_PRIO")) {
        int least = 0, greatest = 0;
        if (std::atoi(e) > 0 && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess) prio = greatest;
    }
    for (int g = 0; g < MAX_GROUPS; ++g) {
        if (!chk(hipStreamCreateWithPriority(&h->gstream[g], hipStreamNonBlocking, prio), "hipStreamCreate")) return PA_ERR_HIP;
        if (!chk(hipEventCreateWithFlags(&h->up[g], hipEventDisableTiming), "hipEventCreate")) return PA_ERR_HIP;
    }
    if (!chk(hipStreamCreateWithPriority(&h->copy_stream, hipStreamNonBlocking, prio), "hipStreamCreate")) return PA_ERR_HIP;
    if (!chk(hipEventCreateWithFlags(&h->fork, hipEventDisableTiming), "hipEventCreate")) return PA_ERR_HIP;
    if (!chk(hipEventCreateWithFlags(&h->prologue, hipEventDisableTiming), "hipEventCreate")) return PA_ERR_HIP;
    if (!chk(hipHostMalloc(&h->h_flag, n * sizeof(int32_t)), "hipHostMalloc")) return PA_ERR_HIP;
    return PA_OK;
}

int pa_mjpeg_probe(const uint8_t* data_host, size_t nbytes, int32_t* info8, char* why, size_t why_bytes) {
    if (!data_host || !info8) return PA_ERR_INVALID_ARG;
    Parsed P;
    const char* msg = parse_header(data_host, nbytes, P);
    if (why && why_bytes) snprintf(why, why_bytes, "%s", msg ? msg : "");
    if (msg) return PA_ERR_INVALID_ARG;
    int hmax = 1, vmax = 1;
    for (int c = 0; c < P.ncomp; ++c) {
        hmax = P.h[c] > hmax ? P.h[c] : hmax;
        vmax = P.v[c] > vmax ? P.v[c] : vmax;
    }
    info8[0] = P.height; info8[1] = P.width; info8[2] = P.ncomp; info8[3] = hmax; info8[4] = vmax; info8[5] = P.ri;
    info8[6] = (int32_t)P.scan_off; info8[7] = 0;
    return PA_OK;
}

int pa_mjpeg_set_sync_rounds(pa_mjpeg* h, int32_t rounds) {
    if (!h || rounds < 0 || rounds > MAX_ROUNDS) return PA_ERR_INVALID_ARG;
    h->sync_rounds = rounds;
    return PA_OK;
}

int pa_mjpeg_last_sync_rounds(const pa_mjpeg* h) { return h ? h->last_rounds : 0; }

int pa_mjpeg_set_groups(pa_mjpeg* h, int32_t groups) {
    if (!h || groups < 1 || groups > MAX_GROUPS) return PA_ERR_INVALID_ARG;
    h->groups = groups;
    return PA_OK;
}

int pa_mjpeg_debug_counters(unsigned long long* out8_host) {
    if (!out8_host) return PA_ERR_INVALID_ARG;
    return hipMemcpyFromSymbol(out8_host, HIP_SYMBOL(g_dbg), 16 * sizeof(unsigned long long)) == hipSuccess ? PA_OK : PA_ERR_HIP;
}

int pa_mjpeg_decode(pa_mjpeg* h, const uint8_t* data_host, const int64_t* spans_host, int32_t n, int32_t height, int32_t width,
                    int32_t rgb, uint8_t* frames_dev, int32_t* status_dev, void* stream) {
    if (!h) return PA_ERR_INVALID_ARG;
    auto bad = [&](int code, const std::string& msg) { h->last_error = msg; return code; };
    if (!data_host || !spans_host || !frames_dev || n < 1 || height < 1 || width < 1)
        return bad(PA_ERR_INVALID_ARG, "pa_mjpeg_decode: bad argument");
    if (n > h->max_frames) return bad(PA_ERR_CAPACITY, "pa_mjpeg_decode: more frames than max_frames");
    if (height > h->max_h || width > h->max_w) return bad(PA_ERR_CAPACITY, "pa_mjpeg_decode: frame larger than max_height x max_width");
    // one copy moves the byte range that covers every frame of the call
    int64_t base = spans_host[0], top = spans_host[1];
    for (int f = 0; f < n; ++f) {
        const int64_t o = spans_host[2 * f], e = spans_host[2 * f + 1];
        if (o < 0 || e <= o) return bad(PA_ERR_INVALID_ARG, "pa_mjpeg_decode: frame " + std::to_string(f) + " has an empty or negative byte span");
        base = o < base ? o : base;
        top = e > top ? e : top;
    }
    const int64_t total = top - base;
    if ((size_t)total > h->max_bytes) return bad(PA_ERR_CAPACITY, "pa_mjpeg_decode: compressed bytes exceed max_bytes");
    hipStream_t s = (hipStream_t)stream;
    auto chk = [&](hipError_t e, const char* what) -> bool {
        if (e == hipSuccess) return true;
        h->last_error = std::string(what) + ": " + hipGetErrorString(e);
        return false;
    };
    static const bool trace = getenv("PA_MJPEG_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_in = now();
    double t_mark[8] = {0};
    if (!chk(hipSetDevice(h->device), "hipSetDevice")) return PA_ERR_HIP;
    const int k = h->turn;
    h->turn ^= 1;
    if (h->staged_used[k] && !chk(hipEventSynchronize(h->staged[k]), "hipEventSynchronize")) return PA_ERR_HIP;
    // the set's call before last has finished (at most two calls are in flight): nothing enqueued below has to wait for it
    if (h->set[k].used)
        for (int o = 0; o < MAX_GROUPS; ++o)
            if (!chk(hipEventSynchronize(h->set[k].done[o]), "hipEventSynchronize")) return PA_ERR_HIP;
    t_mark[0] = now();
    FrameDesc* fd = h->h_fd[k];
    TableSet* ts = h->h_ts[k];
    Geom g;
    memset(&g, 0, sizeof g);
    Parsed first, prev, cur;
    int n_sets = 0;
    size_t seg_total = 0, sub_total = 0, clean_total = 0;
    uint32_t max_scan = 0;
    int max_sub = 1;
    for (int f = 0; f < n; ++f) {
        const int64_t o = spans_host[2 * f], e = spans_host[2 * f + 1];
        const char* msg = parse_header(data_host + o, (size_t)(e - o), cur);
        if (msg) return bad(PA_ERR_INVALID_ARG, "pa_mjpeg_decode: frame " + std::to_string(f) + ": " + msg);
        if (cur.height != height || cur.width != width)
            return bad(PA_ERR_INVALID_ARG, "pa_mjpeg_decode: frame " + std::to_string(f) + " is " + std::to_string(cur.width) + "x" +
                                               std::to_string(cur.height) + ", the call said " + std::to_string(width) + "x" +
                                               std::to_string(height));
        if (f == 0) {
            first = cur;
            g.ncomp = cur.ncomp;
            int hmax = 1, vmax = 1;
            for (int c = 0; c < cur.ncomp; ++c) {
                hmax = cur.h[c] > hmax ? cur.h[c] : hmax;
                vmax = cur.v[c] > vmax ? cur.v[c] : vmax;
            }
            if (cur.ncomp == 3) {
                const bool ok = cur.h[0] == hmax && cur.v[0] == vmax && cur.h[1] == cur.h[2] && cur.v[1] == cur.v[2] && cur.h[1] == 1 &&
                                cur.v[1] == 1 && ((hmax == 1 && vmax == 1) || (hmax == 2 && vmax == 1) || (hmax == 2 && vmax == 2));
                if (!ok) return bad(PA_ERR_INVALID_ARG, "pa_mjpeg_decode: chroma sampling other than 4:4:4 / 4:2:2 / 4:2:0");
            }
            const bool single = cur.ncomp == 1;  // T.81 A.2.2: a one-component scan is not interleaved
            g.mcus_x = (width + 8 * (single ? 1 : hmax) - 1) / (8 * (single ? 1 : hmax));
            g.mcus_y = (height + 8 * (single ? 1 : vmax) - 1) / (8 * (single ? 1 : vmax));
            g.fh = hmax; g.fv = vmax;
            int nb = 0, off = 0;
            for (int c = 0; c < cur.ncomp; ++c) {
                g.hs[c] = single ? 1 : cur.h[c];
                g.vs[c] = single ? 1 : cur.v[c];
                g.bx[c] = g.mcus_x * g.hs[c];
                g.by[c] = g.mcus_y * g.vs[c];
                g.blk_off[c] = off;
                off += g.bx[c] * g.by[c];
                for (int v = 0; v < g.vs[c]; ++v)
                    for (int x = 0; x < g.hs[c]; ++x, ++nb) {
                        g.b_comp[nb] = (uint8_t)c; g.b_dy[nb] = (uint8_t)v; g.b_dx[nb] = (uint8_t)x;
                    }
            }
            g.blocks_per_mcu = nb;
            g.blocks_per_frame = off;
            g.height = height; g.width = width;
            if ((size_t)off > h->max_blocks) return bad(PA_ERR_CAPACITY, "pa_mjpeg_decode: block raster exceeds the handle's capacity");
        } else {
            bool same = cur.ncomp == first.ncomp;
            for (int c = 0; same && c < cur.ncomp; ++c) same = cur.h[c] == first.h[c] && cur.v[c] == first.v[c];
            if (!same) return bad(PA_ERR_INVALID_ARG, "pa_mjpeg_decode: frame " + std::to_string(f) + " changes the sampling factors");
        }
        // tables: consecutive frames with identical DQT / DHT content share one device table set
        bool share = f > 0;
        if (share) share = memcmp(cur.q, prev.q, sizeof cur.q) == 0 && memcmp(cur.counts, prev.counts, sizeof cur.counts) == 0 &&
                           memcmp(cur.syms, prev.syms, sizeof cur.syms) == 0 && memcmp(cur.hdef, prev.hdef, sizeof cur.hdef) == 0;
        if (!share) {
            TableSet& T = ts[n_sets++];
            memset(&T.h, 0, sizeof T.h);
            for (int t = 0; t < 4; ++t)
                if (cur.hdef[t]) build_hufftab(T.h, t, cur.counts[t], cur.syms[t]);
            memcpy(T.q, cur.q, sizeof T.q);
        }
        prev = cur;
        FrameDesc& d = fd[f];
        memset(&d, 0, sizeof d);
        // the entropy-coded segment ends in front of the EOI marker (fill bytes / container padding may follow it)
        int64_t end = e;
        while (end - o > (int64_t)cur.scan_off + 2 && !(data_host[end - 2] == 0xff && data_host[end - 1] == 0xd9) && e - end < 16) --end;
        if (data_host[end - 2] == 0xff && data_host[end - 1] == 0xd9) end -= 2; else end = e;
        d.scan_off = (uint32_t)(o - base + (int64_t)cur.scan_off);
        d.scan_len = (uint32_t)((end - o) - (int64_t)cur.scan_off);
        d.ri = cur.ri;
        const int mcus = g.mcus_x * g.mcus_y;
        d.n_int = cur.ri ? (mcus + cur.ri - 1) / cur.ri : 1;
        d.seg_base = (int32_t)seg_total;
        seg_total += (size_t)d.n_int;
        d.sub_base = (int32_t)sub_total;

        d.clean_off = (uint32_t)clean_total;
        clean_total += ((size_t)d.scan_len + 31) & ~(size_t)15;
        d.tabset = n_sets - 1;
        for (int c = 0; c < cur.ncomp; ++c) {
            d.td[c] = (uint8_t)cur.td[c]; d.ta[c] = (uint8_t)cur.ta[c]; d.tq[c] = (uint8_t)cur.tq[c];
        }
        max_scan = d.scan_len > max_scan ? d.scan_len : max_scan;
    }
    // subsequence size: about four MCUs of the stream, a power of two. Measured at 1080p / quality 95 (1 MB per frame,
    // scripts/mjpeg_sync_probe.py, scripts/mjpeg_rate.py): the slowest lanes need about ten MCUs to fall into step, i.e.
    // two verify passes at 1024 bytes, three at 512, eleven at 128; a lane's symbol takes ~1000 cycles whether one or two
    // waves share its SIMD (~115 instructions, one LDS round trip), so 512-byte lanes (two waves per SIMD) finish a
    // pass in half the time of 1024-byte lanes and win although they need one more pass: 5.0 vs 5.8 ms per 64 frames.
    {
        size_t bytes = 0;
        for (int f = 0; f < n; ++f) bytes += fd[f].scan_len;
        const size_t per_mcu = bytes / ((size_t)n * g.mcus_x * g.mcus_y) + 1;
        int sh = 8;  // the power of two nearest to four MCUs, 256 bytes .. 8 KB
        while ((3u << sh) / 2 < 4 * per_mcu && sh < 13) ++sh;
        if (h->sub_shift_override >= 7 && h->sub_shift_override <= 13) sh = h->sub_shift_override;
        g.sub_shift = sh;
        for (int f = 0; f < n; ++f) {
            FrameDesc& d = fd[f];
            d.sub_base = (int32_t)sub_total;
            d.n_sub_cap = (int32_t)(((d.scan_len + (1u << sh) - 1) >> sh) + 1);
            sub_total += (size_t)d.n_sub_cap;
            max_sub = d.n_sub_cap > max_sub ? d.n_sub_cap : max_sub;
        }
    }
    t_mark[1] = now();
    const int max_chunks = (int)(max_scan / CHUNK) + 2;
    if (max_chunks > h->max_chunks_cap) return bad(PA_ERR_CAPACITY, "pa_mjpeg_decode: scan longer than the handle's chunk table");
    if (seg_total > h->max_segs || sub_total > h->max_subs) return bad(PA_ERR_CAPACITY, "pa_mjpeg_decode: more restart intervals / subsequences than the handle holds");
    // spans may overlap or repeat (the same frame several times): what bounds the clean stream is the SUM of the scans, not
    // the byte range the spans cover
    if (clean_total + 4096 > h->clean_bytes) return bad(PA_ERR_CAPACITY, "pa_mjpeg_decode: the frames' entropy-coded segments add up to more than max_bytes");
    pa_mjpeg::Set& S = h->set[k];
    // exact mode looks at flags on the host between passes: one group, on the caller's stream
    const int G = h->sync_rounds > 0 ? (h->groups < n ? h->groups : n) : 1;
    const bool forked = G > 1;
    // Nothing below depends on what the caller's stream holds except the WRITES of the decoded frames (the caller may
    // still be reading the output buffer): the groups wait for `fork` only in front of their last kernel, so a call's
    // entropy passes start while the call before it -- which the caller's stream has to wait for -- is still decoding.
    if (forked && !chk(hipEventRecord(h->fork, s), "hipEventRecord")) return PA_ERR_HIP;
    hipStream_t q0 = forked ? h->gstream[0] : s;
    // descriptors and tables -> HBM, flags cleared: on the first group's stream, which the other groups wait for
    {
        static_assert(sizeof(FrameDesc) % 4 == 0 && sizeof(TableSet) % 4 == 0, "copied as dwords");
        const int na = (int)((size_t)n * sizeof(FrameDesc) / 4), nb = (int)((size_t)n_sets * sizeof(TableSet) / 4);
        hipLaunchKernelGGL(stage_copy_kernel, dim3((na + nb + 255) / 256), dim3(256), 0, q0, reinterpret_cast<const uint32_t*>(fd),
                           reinterpret_cast<uint32_t*>(S.d_fd), na, reinterpret_cast<const uint32_t*>(ts),
                           reinterpret_cast<uint32_t*>(S.d_ts), nb);
    }
    if (!chk(hipEventRecord(h->staged[k], q0), "hipEventRecord")) return PA_ERR_HIP;
    h->staged_used[k] = true;
    t_mark[2] = now();
    if (!chk(hipMemsetAsync(S.d_status, 0, (size_t)n * sizeof(int32_t), q0), "clear status")) return PA_ERR_HIP;
    if (!chk(hipMemsetAsync(S.d_changed, 0, (size_t)(MAX_ROUNDS + 1) * h->max_frames * sizeof(int32_t), q0), "clear flags")) return PA_ERR_HIP;
    if (forked && !chk(hipEventRecord(h->prologue, q0), "hipEventRecord")) return PA_ERR_HIP;
    t_mark[3] = now();
    const int total_blocks = g.mcus_x * g.mcus_y * g.blocks_per_mcu;
    int rounds_run = 0;
    // the groups' compressed bytes (the byte range that covers a group's frames), one group after the other
    for (int gi = 0; gi < G; ++gi) {
        const int f0 = (int)((long long)n * gi / G), f1 = (int)((long long)n * (gi + 1) / G);
        hipStream_t cq = forked ? h->copy_stream : s;
        int64_t gb = spans_host[2 * f0], gt = spans_host[2 * f0 + 1];
        for (int f = f0; f < f1; ++f) {
            gb = spans_host[2 * f] < gb ? spans_host[2 * f] : gb;
            gt = spans_host[2 * f + 1] > gt ? spans_host[2 * f + 1] : gt;
        }
        if (!chk(hipMemcpyAsync(S.d_bits + (gb - base), data_host + gb, (size_t)(gt - gb), hipMemcpyHostToDevice, cq), "upload bitstream")) return PA_ERR_HIP;
        // readers run a few bytes past the end of a scan: zeros behind the last byte of the call
        if (gt == top && !chk(hipMemsetAsync(S.d_bits + total, 0, 64, cq), "pad bitstream")) return PA_ERR_HIP;
        if (forked && !chk(hipEventRecord(h->up[gi], cq), "hipEventRecord")) return PA_ERR_HIP;
    }
    for (int gi = 0; gi < G; ++gi) {
        const int f0 = (int)((long long)n * gi / G), f1 = (int)((long long)n * (gi + 1) / G), ng = f1 - f0;
        hipStream_t q = forked ? h->gstream[gi] : s;
        Geom gg = g;
        gg.f0 = f0;
        if (gi > 0 && !chk(hipStreamWaitEvent(q, h->prologue, 0), "hipStreamWaitEvent")) return PA_ERR_HIP;
        // the group's coefficient buffers, cleared while its bytes are still on their way
        if (!chk(hipMemsetAsync(S.d_coef + (size_t)f0 * total_blocks * 64, 0, (size_t)ng * total_blocks * 64 * sizeof(int16_t), q), "clear coefficients")) return PA_ERR_HIP;
        if (!chk(hipMemsetAsync(S.d_dc + (size_t)f0 * total_blocks, 0, (size_t)ng * total_blocks * sizeof(int16_t), q), "clear DC differences")) return PA_ERR_HIP;
        if (forked && !chk(hipStreamWaitEvent(q, h->up[gi], 0), "hipStreamWaitEvent")) return PA_ERR_HIP;
        hipLaunchKernelGGL(unstuff_count_kernel, dim3(max_chunks, ng), dim3(256), 0, q, S.d_bits, S.d_fd, S.d_chunk, max_chunks, f0);
        hipLaunchKernelGGL(unstuff_write_kernel, dim3(max_chunks, ng), dim3(256), 0, q, S.d_bits, S.d_fd, S.d_chunk, max_chunks, S.d_clean,
                           S.d_seg, S.d_clean_len, S.d_status, f0);
        const int wl = h->wg_lanes;
        const dim3 sgrid((max_sub + wl - 1) / wl, ng);
        int cur_g = 0;
#define MJ_LAUNCH(MODE_, ...)                                                                                         \
    do {                                                                                                              \
        if (wl == 128) hipLaunchKernelGGL((sub_decode_kernel<MODE_, 128>), sgrid, dim3(128), 0, q, __VA_ARGS__);      \
        else hipLaunchKernelGGL((sub_decode_kernel<MODE_, WG_SUBS>), sgrid, dim3(WG_SUBS), 0, q, __VA_ARGS__);        \
    } while (0)
        MJ_LAUNCH(0, S.d_clean, S.d_fd, S.d_ts, S.d_seg, S.d_clean_len, gg,
                  (const uint32_t*)nullptr, S.d_g[0], S.d_used, S.d_cnt, (const SubCnt*)nullptr, (int16_t*)nullptr, S.d_status,
                  (int32_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)nullptr, (int16_t*)nullptr);
        auto verify = [&](int slot, int prev_slot) {
            int32_t* flag = S.d_changed + (size_t)slot * h->max_frames;
            const int32_t* prev = prev_slot >= 0 ? S.d_changed + (size_t)prev_slot * h->max_frames : nullptr;
            hipLaunchKernelGGL(sub_verify_plan_kernel, dim3(ng), dim3(1024), 0, q, S.d_fd, S.d_clean_len, g.sub_shift, S.d_g[cur_g],
                               S.d_g[cur_g ^ 1], S.d_used, S.d_todo, S.d_todo_cnt, prev, f0);
            MJ_LAUNCH(1, S.d_clean, S.d_fd, S.d_ts, S.d_seg, S.d_clean_len, gg,
                      S.d_g[cur_g], S.d_g[cur_g ^ 1], S.d_used, S.d_cnt, (const SubCnt*)nullptr, (int16_t*)nullptr, S.d_status, flag,
                      prev, S.d_todo, S.d_todo_cnt, (int16_t*)nullptr);
            cur_g ^= 1;
        };
        int last_slot = MAX_ROUNDS;  // an all-zero row unless a verify pass wrote it
        if (h->sync_rounds > 0) {
            for (int r = 0; r < h->sync_rounds; ++r) verify(r, r - 1);
            last_slot = h->sync_rounds - 1;
            rounds_run = h->sync_rounds;
        } else {
            // exact mode: verify until a pass changes nothing, looking at the flags on the host (synchronises the stream)
            int rounds = 0;
            for (;;) {
                if (!chk(hipMemsetAsync(S.d_changed, 0, (size_t)h->max_frames * sizeof(int32_t), q), "clear flags")) return PA_ERR_HIP;
                verify(0, -1);
                ++rounds;
                if (!chk(hipMemcpyAsync(h->h_flag, S.d_changed, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, q), "read flags")) return PA_ERR_HIP;
                if (!chk(hipStreamSynchronize(q), "hipStreamSynchronize")) return PA_ERR_HIP;
                bool any = false;
                for (int f = 0; f < n; ++f) any = any || h->h_flag[f] != 0;
                if (!any) break;
                if (rounds > max_sub + 2) return bad(PA_ERR_HIP, "pa_mjpeg_decode: synchronisation did not settle");
            }
            last_slot = 0;
            rounds_run = rounds;
        }
        hipLaunchKernelGGL(sub_scan_kernel, dim3(ng), dim3(1024), 0, q, S.d_fd, S.d_clean_len, S.d_cnt, S.d_entry, g.sub_shift, f0);
        MJ_LAUNCH(2, S.d_clean, S.d_fd, S.d_ts, S.d_seg, S.d_clean_len, gg,
                  S.d_g[cur_g], (uint32_t*)nullptr, S.d_used, S.d_cnt, S.d_entry, S.d_coef, S.d_status, (int32_t*)nullptr,
                  S.d_changed + (size_t)last_slot * h->max_frames, (const int32_t*)nullptr, (const int32_t*)nullptr, S.d_dc);
#undef MJ_LAUNCH
        hipLaunchKernelGGL(dc_scan_kernel, dim3(ng, g.ncomp), dim3(1024), 0, q, S.d_dc, S.d_fd, gg);
        const long long nblk = (long long)ng * g.blocks_per_frame;
        hipLaunchKernelGGL(idct_kernel, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, q, S.d_coef, S.d_dc, S.d_fd, S.d_ts, gg, S.d_planes, ng);
        const int fv = g.ncomp == 3 ? g.fv : 1, fhh = g.ncomp == 3 ? g.fh : 1;
        const dim3 grid((width + 511) / 512, (height + 4 * fv - 1) / (4 * fv), ng);
        if (forked && !chk(hipStreamWaitEvent(q, h->fork, 0), "hipStreamWaitEvent")) return PA_ERR_HIP;  // the output buffer is the caller's
        if (fhh == 2 && fv == 2)
            hipLaunchKernelGGL(ycc420_kernel, dim3((width + 511) / 512, (height + 15) / 16, ng), dim3(256), 0, q, S.d_planes, gg, frames_dev, rgb);
        else if (fhh == 2) hipLaunchKernelGGL((ycc_kernel<2, 1>), grid, dim3(256), 0, q, S.d_planes, gg, frames_dev, rgb);
        else hipLaunchKernelGGL((ycc_kernel<1, 1>), grid, dim3(256), 0, q, S.d_planes, gg, frames_dev, rgb);
        if (!chk(hipEventRecord(S.done[gi], q), "hipEventRecord")) return PA_ERR_HIP;
    }
    // groups the call did not use: their events must not hold a later call back with a stale record
    for (int gi = G; gi < MAX_GROUPS; ++gi)
        if (!chk(hipEventRecord(S.done[gi], s), "hipEventRecord")) return PA_ERR_HIP;
    S.used = true;
    h->last_rounds = rounds_run;
    if (forked)
        for (int gi = 0; gi < G; ++gi)
            if (!chk(hipStreamWaitEvent(s, S.done[gi], 0), "hipStreamWaitEvent")) return PA_ERR_HIP;
    if (status_dev && !chk(hipMemcpyAsync(status_dev, S.d_status, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, s), "copy status"))
        return PA_ERR_HIP;
    if (!chk(hipGetLastError(), "kernel launch")) return PA_ERR_HIP;
    if (trace)
        fprintf(stderr, "pa_mjpeg_decode host us: staging wait %.0f, headers %.0f, descriptors %.0f, clears %.0f, groups (copies + launches) %.0f\n",
                t_mark[0] - t_in, t_mark[1] - t_mark[0], t_mark[2] - t_mark[1], t_mark[3] - t_mark[2], now() - t_mark[3]);
    return PA_OK;
}

}  // extern "C"
