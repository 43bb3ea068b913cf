// Motion-JPEG frame decode on the device: baseline JPEG files -> uint8 [n][H][W][3] BGR frames in HBM.
//
// Replaces the per-frame work of cv2.VideoCapture.read / cv2.imread in the reference (playaid/ai_runner.py:153,404-405,446;
// playaid/manuscript.py:154-155) for Motion-JPEG streams and JPEG image sequences. The arithmetic is libjpeg(-turbo)'s with
// its defaults, which is what OpenCV's JPEG reader runs: Huffman entropy decoding (T.81 F.2.2), de-quantisation + the
// slow-but-accurate integer IDCT (jidctint.c), "fancy" triangle-filter chroma up-sampling with libjpeg's edge
// replication (jdsample.c h2v2 / h2v1, jdmainct.c), YCbCr -> RGB (jdcolor.c). Bit-exact against oracle/jpeg.py::decode,
// which is pinned byte for byte against PIL.Image.open (live libjpeg-turbo).
//
// Pipeline per call (all on the caller's stream):
//   host : marker segments of every frame (SOF0/SOF1, DQT, DHT, DRI, SOS) -> frame descriptors + Huffman / quantisation
//          table sets (consecutive frames with identical tables share one set)
//   copy : the compressed bytes, the descriptors and the table sets, host -> HBM
//   rst_count_kernel / rst_write_kernel : positions of the RSTm markers of every frame, in stream order
//   huff_kernel  : ONE LANE PER RESTART INTERVAL walks its interval's bit stream (byte-stuffing handled in the reader,
//                  10-bit direct lookup + canonical fallback for longer codes, tables in LDS) and scatters the non-zero
//                  quantised coefficients (int16, natural order) into per-component block rasters; DC prediction is
//                  local to an interval by definition of the restart marker. A stream without DRI is one interval per
//                  frame -- correct, but one lane per frame.
//   idct_kernel  : one thread per 8x8 block, block in registers -> uint8 sample planes (padded to whole MCUs)
//   ycc_kernel   : up-sampling + colour conversion, 8 pixels x FV rows per thread, 8-byte stores
#include <hip/hip_runtime.h>
#include <stdint.h>

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

enum { ERR_HUFF = 1, ERR_RST = 2, ERR_COEF = 4 };

struct HuffTab {
    uint16_t lut[1 << LB];  // len << 8 | symbol for codes of <= LB bits, 0 = longer
    int32_t maxcode[18];    // largest code of length l (-1: none); [17] = sentinel
    int32_t valoff[17];     // valptr[l] - mincode[l]
    uint8_t vals[256];
};
struct TableSet {
    HuffTab h[4];       // DC0, DC1, AC0, AC1
    uint16_t q[4][64];  // quantisation tables, natural order
};
static_assert(sizeof(HuffTab) % 4 == 0, "copied to LDS as dwords");

struct FrameDesc {
    uint32_t scan_off, scan_len;  // entropy-coded segment inside the device byte buffer
    int32_t ri, n_int, int_base, tabset;
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
    uint8_t b_comp[MAX_BLOCKS_MCU], b_dy[MAX_BLOCKS_MCU], b_dx[MAX_BLOCKS_MCU];
};

struct BlkInfo {  // huff_kernel: one block position of an MCU
    int32_t base, bx;
    uint8_t vs, hs, dy, dx, tdc, tac, comp, pad;
};
static_assert(sizeof(BlkInfo) == 16, "one ds_read_b128");

__constant__ uint8_t k_zigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                     41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                     30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ---- restart markers ---------------------------------------------------------------------------------------------------------

// markers whose 0xFF byte lies in this thread's 16 bytes [a, a + 16) and inside the scan [lo, hi)
__device__ __forceinline__ uint32_t rst_mask(const uint8_t* bits, uint32_t a, uint32_t lo, uint32_t hi) {
    const uint4 v = *reinterpret_cast<const uint4*>(bits + a);
    const uint32_t w[5] = {v.x, v.y, v.z, v.w, bits[a + 16]};
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t b0 = (w[j >> 2] >> (8 * (j & 3))) & 0xff, b1 = (w[(j + 1) >> 2] >> (8 * ((j + 1) & 3))) & 0xff;
        const uint32_t p = a + j;
        if (b0 == 0xff && (b1 & 0xf8) == 0xd0 && p >= lo && p + 1 < hi) m |= 1u << j;
    }
    return m;
}

__global__ __launch_bounds__(256) void rst_count_kernel(const uint8_t* __restrict__ bits, const FrameDesc* __restrict__ fd,
                                                        int32_t* __restrict__ chunk_cnt, int max_chunks) {
    __shared__ int red[4];
    const int f = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
    const FrameDesc d = fd[f];
    const uint32_t lo = d.scan_off, hi = d.scan_off + d.scan_len;
    const uint32_t a = (lo & ~15u) + (uint32_t)c * CHUNK + tid * 16;
    int cnt = 0;
    if (d.ri && a < hi) cnt = __popc(rst_mask(bits, a, lo, hi));
    for (int o = 32; o; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) chunk_cnt[f * max_chunks + c] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void rst_write_kernel(const uint8_t* __restrict__ bits, const FrameDesc* __restrict__ fd,
                                                        const int32_t* __restrict__ chunk_cnt, int max_chunks,
                                                        uint32_t* __restrict__ rst_pos, int32_t* __restrict__ status) {
    __shared__ int red[8];
    __shared__ int scan[256];
    const int f = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
    const FrameDesc d = fd[f];
    if (!d.ri) return;
    // markers in the chunks before this one, and in the whole frame
    int before = 0, total = 0;
    for (int i = tid; i < max_chunks; i += 256) {
        const int v = chunk_cnt[f * max_chunks + i];
        total += v;
        if (i < c) before += v;
    }
    for (int o = 32; o; o >>= 1) {
        before += __shfl_down(before, o, 64);
        total += __shfl_down(total, o, 64);
    }
    if ((tid & 63) == 0) {
        red[tid >> 6] = before;
        red[4 + (tid >> 6)] = total;
    }
    __syncthreads();
    before = red[0] + red[1] + red[2] + red[3];
    total = red[4] + red[5] + red[6] + red[7];
    if (c == 0 && tid == 0 && total != d.n_int - 1) atomicOr(&status[f], ERR_RST);
    const uint32_t lo = d.scan_off, hi = d.scan_off + d.scan_len;
    const uint32_t a = (lo & ~15u) + (uint32_t)c * CHUNK + tid * 16;
    const uint32_t m = a < hi ? rst_mask(bits, a, lo, hi) : 0;
    scan[tid] = __popc(m);
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {  // inclusive scan of the per-thread counts
        const int v = tid >= o ? scan[tid - o] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    int k = before + scan[tid] - __popc(m);
    for (uint32_t mm = m; mm; mm &= mm - 1, ++k)
        if (k < d.n_int - 1) rst_pos[d.int_base + k] = a + (uint32_t)__ffs(mm) - 1 + 2;  // first byte after the marker
}

// ---- entropy decoding --------------------------------------------------------------------------------------------------------

struct BitReader {
    const uint8_t* bits;
    uint32_t pos, end;
    uint64_t buf;   // valid bits at the top
    int nbits;
    bool marker;    // a marker (or the end of the data) was reached: zeros from here on (T.81 F.2.2.5)

    __device__ __forceinline__ void refill() {
        if (nbits > 32) return;
        // bytes pos .. pos + 3 from two aligned dwords
        const uint32_t* w = reinterpret_cast<const uint32_t*>(bits) + (pos >> 2);
        const uint32_t x = __builtin_amdgcn_alignbyte(w[1], w[0], pos & 3);
        uint32_t v;
        if (!marker && pos + 4 <= end && (((~x) - 0x01010101u) & x & 0x80808080u) == 0) {
            v = __builtin_bswap32(x);
            pos += 4;
        } else {
            v = 0;
            for (int k = 0; k < 4; ++k) {
                uint32_t b = 0;
                if (!marker) {
                    if (pos >= end) {
                        marker = true;
                    } else {
                        b = bits[pos];
                        if (b == 0xff) {
                            if (bits[pos + 1] == 0) {
                                pos += 2;  // stuffed zero
                            } else {
                                marker = true;
                                b = 0;
                            }
                        } else {
                            ++pos;
                        }
                    }
                }
                v = (v << 8) | b;
            }
        }
        buf |= (uint64_t)v << (32 - nbits);
        nbits += 32;
    }
    __device__ __forceinline__ uint32_t peek16() const { return (uint32_t)(buf >> 48); }
    __device__ __forceinline__ void skip(int n) {
        buf <<= n;
        nbits -= n;
    }
    // the next s bits (0 <= s <= 16)
    __device__ __forceinline__ uint32_t take(int s) {
        const uint32_t v = (uint32_t)((buf >> 1) >> (63 - s));
        skip(s);
        return v;
    }
};

__global__ __launch_bounds__(64) void huff_kernel(const uint8_t* __restrict__ bits, const FrameDesc* __restrict__ fd,
                                                  const TableSet* __restrict__ ts, const uint32_t* __restrict__ rst_pos, const Geom g,
                                                  int16_t* __restrict__ coef, int32_t* __restrict__ status) {
    __shared__ HuffTab tab[4];
    __shared__ uint8_t zz[64];
    __shared__ BlkInfo binfo[MAX_BLOCKS_MCU];
    const int f = blockIdx.y, lane = threadIdx.x;
    const FrameDesc d = fd[f];
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(ts[d.tabset].h);
        uint32_t* dst = reinterpret_cast<uint32_t*>(tab);
        for (int i = lane; i < (int)(sizeof(HuffTab) * 4 / 4); i += 64) dst[i] = src[i];
        zz[lane] = k_zigzag[lane];
        if (lane < g.blocks_per_mcu) {
            // everything a lane needs when it moves on to block `lane` of an MCU, as one 16-byte LDS read
            const int c = g.b_comp[lane];
            BlkInfo bi;
            bi.base = c == 0 ? g.blk_off[0] : (c == 1 ? g.blk_off[1] : g.blk_off[2]);
            bi.bx = c == 0 ? g.bx[0] : (c == 1 ? g.bx[1] : g.bx[2]);
            bi.vs = (uint8_t)(c == 0 ? g.vs[0] : (c == 1 ? g.vs[1] : g.vs[2]));
            bi.hs = (uint8_t)(c == 0 ? g.hs[0] : (c == 1 ? g.hs[1] : g.hs[2]));
            bi.dy = g.b_dy[lane];
            bi.dx = g.b_dx[lane];
            bi.tdc = fd[f].td[c];
            bi.tac = (uint8_t)(2 + fd[f].ta[c]);
            bi.comp = (uint8_t)c;
            bi.pad = 0;
            binfo[lane] = bi;
        }
    }
    __syncthreads();
    const int iv = blockIdx.x * 64 + lane;
    bool active = iv < d.n_int;
    const int mcus = g.mcus_x * g.mcus_y;
    const int mcu0 = active ? iv * d.ri : 0;
    int mleft = d.ri ? min(d.ri, mcus - mcu0) : mcus;
    int my = mcu0 / g.mcus_x, mx = mcu0 - my * g.mcus_x;
    BitReader br;
    br.bits = bits;
    br.end = d.scan_off + d.scan_len;
    br.pos = d.scan_off;
    if (active && iv > 0) br.pos = rst_pos[d.int_base + iv - 1];
    if (br.pos < d.scan_off || br.pos > br.end) active = false;  // marker list shorter than the header promised
    br.buf = 0;
    br.nbits = 0;
    br.marker = false;
    int16_t* const frame_coef = coef + (size_t)f * g.blocks_per_frame * 64;
    int pred0 = 0, pred1 = 0, pred2 = 0;
    int b = 0, z = 0, err = 0;
    BlkInfo bi = binfo[0];
    int16_t* blk = frame_coef + ((size_t)bi.base + (size_t)(my * bi.vs + bi.dy) * bi.bx + mx * bi.hs + bi.dx) * 64;
    if (mleft <= 0) active = false;
    while (__ballot(active)) {
        if (active) {
            br.refill();
            const bool dc = z == 0;
            const HuffTab& T = tab[dc ? bi.tdc : bi.tac];
            const uint32_t pk = br.peek16();
            const uint32_t e = T.lut[pk >> (16 - LB)];
            int len = e >> 8, sym = e & 0xff;
            if (e == 0) {
                len = 17;
                for (int l = LB + 1; l <= 16; ++l) {
                    const int code = (int)(pk >> (16 - l));
                    if (code <= T.maxcode[l]) {
                        sym = T.vals[(T.valoff[l] + code) & 255];
                        len = l;
                        break;
                    }
                }
                if (len == 17) {
                    err |= ERR_HUFF;
                    active = false;
                    len = 0;
                }
            }
            br.skip(len);
            const int r = dc ? 0 : sym >> 4, s = dc ? sym : sym & 15;
            int v = 0;
            if (s) {
                const int raw = (int)br.take(s);
                v = raw < (1 << (s - 1)) ? raw - (1 << s) + 1 : raw;
            }
            if (dc) {
                if (s > 11) {
                    err |= ERR_HUFF;
                    active = false;
                }
                int p = bi.comp == 0 ? pred0 : (bi.comp == 1 ? pred1 : pred2);
                p += v;
                if (bi.comp == 0) pred0 = p; else if (bi.comp == 1) pred1 = p; else pred2 = p;
                if (p) blk[0] = (int16_t)p;
                z = 1;
            } else if (s) {
                z += r;
                if (z > 63) {
                    err |= ERR_COEF;
                    active = false;
                    z = 63;
                }
                blk[zz[z]] = (int16_t)v;
                ++z;
            } else {
                z = r == 15 ? z + 16 : 64;  // ZRL | EOB
            }
            if (z >= 64) {
                z = 0;
                if (++b == g.blocks_per_mcu) {
                    b = 0;
                    if (++mx == g.mcus_x) {
                        mx = 0;
                        ++my;
                    }
                    if (--mleft == 0) active = false;
                }
                bi = binfo[b];
                blk = frame_coef + ((size_t)bi.base + (size_t)(my * bi.vs + bi.dy) * bi.bx + mx * bi.hs + bi.dx) * 64;
            }
        }
    }
    if (err) atomicOr(&status[f], err);
}

// ---- de-quantisation + inverse DCT -------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void idct_kernel(const int16_t* __restrict__ coef, const FrameDesc* __restrict__ fd,
                                                   const TableSet* __restrict__ ts, const Geom g, uint8_t* __restrict__ planes,
                                                   int n_frames) {
    using namespace dct;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)n_frames * g.blocks_per_frame) return;
    const int f = (int)(t / g.blocks_per_frame), r = (int)(t - (long long)f * g.blocks_per_frame);
    const int c = (g.ncomp > 2 && r >= g.blk_off[2]) ? 2 : ((g.ncomp > 1 && r >= g.blk_off[1]) ? 1 : 0);
    const int off_c = c == 0 ? g.blk_off[0] : (c == 1 ? g.blk_off[1] : g.blk_off[2]);
    const int bx_c = c == 0 ? g.bx[0] : (c == 1 ? g.bx[1] : g.bx[2]);
    const int rb = r - off_c;
    const int by = rb / bx_c, bx = rb - by * bx_c;
    const uint16_t* __restrict__ q = ts[fd[f].tabset].q[fd[f].tq[c]];
    const uint4* src = reinterpret_cast<const uint4*>(coef + (size_t)t * 64);
    int d[64];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint4 v = src[i];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            d[i * 8 + 2 * j] = (int)(int16_t)(w[j] & 0xffff) * (int)q[i * 8 + 2 * j];
            d[i * 8 + 2 * j + 1] = (int)(int16_t)(w[j] >> 16) * (int)q[i * 8 + 2 * j + 1];
        }
    }
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
template <int FH, int FV>
__device__ __forceinline__ void chroma8(const uint8_t* __restrict__ C, int pc, int cw, int ch, int x0, int y0, int (&o)[FV][8]) {
    if (FH == 1 && FV == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) o[0][i] = C[(size_t)y0 * pc + min(x0 + i, cw - 1)];
    } else if (FH == 2 && FV == 1) {
        const int cx0 = x0 >> 1;
        int s[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) s[i] = C[(size_t)y0 * pc + min(max(cx0 - 1 + i, 0), cw - 1)];
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
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int cx = min(max(cx0 - 1 + i, 0), cw - 1);
            s0[i] = C[(size_t)cy * pc + cx];
            sa[i] = C[(size_t)ya * pc + cx];
            sb[i] = C[(size_t)yb * pc + cx];
        }
#pragma unroll
        for (int v = 0; v < FV; ++v) {
            int col[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) col[i] = 3 * s0[i] + (v ? sb[i] : sa[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cx = cx0 + i;
                o[v][2 * i] = cx == 0 ? (col[i + 1] * 4 + 8) >> 4 : (col[i + 1] * 3 + col[i] + 8) >> 4;
                o[v][2 * i + 1] = cx >= cw - 1 ? (col[i + 1] * 4 + 7) >> 4 : (col[i + 1] * 3 + col[i + 2] + 7) >> 4;
            }
        }
    }
}

// 8 pixels x FV rows per thread: up-sampling, jdcolor.c's YCbCr -> RGB, three 8-byte stores per row.
template <int FH, int FV>
__global__ __launch_bounds__(256) void ycc_kernel(const uint8_t* __restrict__ planes, const Geom g, uint8_t* __restrict__ out, int rgb) {
    using namespace dct;
    const int f = blockIdx.z;
    const int x0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 8, y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * FV;
    if (x0 >= g.width || y0 >= g.height) return;
    const uint8_t* fp = planes + (size_t)f * g.blocks_per_frame * 64;
    const int py = g.bx[0] * 8;
    const uint8_t* Y = fp + (size_t)g.blk_off[0] * 64;
    int cb[FV][8], cr[FV][8];
    const bool colour = g.ncomp == 3;
    if (colour) {
        const int pc = g.bx[1] * 8;
        const int cw = (g.width + FH - 1) / FH, ch = (g.height + FV - 1) / FV;  // jdmaster.c: downsampled_width / _height
        chroma8<FH, FV>(fp + (size_t)g.blk_off[1] * 64, pc, cw, ch, x0, y0, cb);
        chroma8<FH, FV>(fp + (size_t)g.blk_off[2] * 64, pc, cw, ch, x0, y0, cr);
    }
#pragma unroll
    for (int v = 0; v < FV; ++v) {
        const int y = y0 + v;
        if (y < g.height) {
            const uint2 yv = *reinterpret_cast<const uint2*>(Y + (size_t)y * py + x0);
            uint32_t px[24];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int yy = (int)(((i < 4 ? yv.x : yv.y) >> (8 * (i & 3))) & 0xff);
                int r = yy, gg = yy, bb = yy;
                if (colour) {
                    const int xb = cb[v][i] - 128, xr = cr[v][i] - 128;
                    r = clamp255(yy + ((91881 * xr + 32768) >> 16));
                    gg = clamp255(yy + ((-22554 * xb + 32768 - 46802 * xr) >> 16));
                    bb = clamp255(yy + ((116130 * xb + 32768) >> 16));
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
    }
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

void build_hufftab(HuffTab& T, const uint8_t* counts, const uint8_t* syms) {
    memset(&T, 0, sizeof T);
    memcpy(T.vals, syms, 256);
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        T.valoff[l] = k - code;
        for (int i = 0; i < counts[l - 1]; ++i, ++k, ++code) {
            if (l <= LB) {
                const int lo = code << (LB - l), hi = (code + 1) << (LB - l);
                for (int e = lo; e < hi && e < (1 << LB); ++e) T.lut[e] = (uint16_t)((l << 8) | syms[k]);
            }
        }
        T.maxcode[l] = counts[l - 1] ? code - 1 : -1;
        code <<= 1;
    }
    T.maxcode[17] = 0x7fffffff;
}

}  // namespace mj
}  // namespace pa

using namespace pa::mj;

struct pa_mjpeg {
    int device = 0, max_frames = 0, max_h = 0, max_w = 0;
    size_t max_bytes = 0;
    size_t max_blocks = 0;      // per frame
    int max_chunks_cap = 0;
    uint8_t* d_bits = nullptr;
    FrameDesc* d_fd = nullptr;
    TableSet* d_ts = nullptr;
    int32_t* d_chunk = nullptr;
    uint32_t* d_rst = nullptr;
    int16_t* d_coef = nullptr;
    uint8_t* d_planes = nullptr;
    int32_t* d_status = nullptr;
    // pinned host staging, two sets used in turn
    FrameDesc* h_fd[2] = {nullptr, nullptr};
    TableSet* h_ts[2] = {nullptr, nullptr};
    hipEvent_t staged[2] = {nullptr, nullptr};
    bool staged_used[2] = {false, false};
    int turn = 0;
    int last_height = 0, last_width = 0;
    std::string last_error;
};

extern "C" {

const char* pa_mjpeg_last_error(const pa_mjpeg* h) { return h ? h->last_error.c_str() : "null handle"; }

void pa_mjpeg_destroy(pa_mjpeg* h) {
    if (!h) return;
    (void)hipFree(h->d_bits);
    (void)hipFree(h->d_fd);
    (void)hipFree(h->d_ts);
    (void)hipFree(h->d_chunk);
    (void)hipFree(h->d_rst);
    (void)hipFree(h->d_coef);
    (void)hipFree(h->d_planes);
    (void)hipFree(h->d_status);
    for (int i = 0; i < 2; ++i) {
        if (h->h_fd[i]) (void)hipHostFree(h->h_fd[i]);
        if (h->h_ts[i]) (void)hipHostFree(h->h_ts[i]);
        if (h->staged[i]) (void)hipEventDestroy(h->staged[i]);
    }
    delete h;
}

int pa_mjpeg_create(int32_t device, int32_t max_frames, int32_t max_height, int32_t max_width, size_t max_bytes, pa_mjpeg** out) {
    if (!out) return PA_ERR_INVALID_ARG;
    *out = nullptr;
    if (max_frames < 1 || max_height < 1 || max_width < 1 || max_height > 65535 || max_width > 65535 || max_bytes < 1024 ||
        max_bytes > 0xf0000000ull)
        return PA_ERR_INVALID_ARG;
    pa_mjpeg* h = new pa_mjpeg();
    *out = h;  // handed back on failure too (pa_mjpeg_last_error, then pa_mjpeg_destroy)
    h->device = device; h->max_frames = max_frames; h->max_h = max_height; h->max_w = max_width; h->max_bytes = max_bytes;
    auto chk = [&](hipError_t e, const char* what) -> bool {
        if (e == hipSuccess) return true;
        h->last_error = std::string(what) + ": " + hipGetErrorString(e);
        return false;
    };
    if (!chk(hipSetDevice(device), "hipSetDevice")) return PA_ERR_NO_DEVICE;
    // worst case: three full-resolution components, padded to 16-pixel MCUs
    const size_t bw = ((size_t)max_width + 15) / 16 * 2, bh = ((size_t)max_height + 15) / 16 * 2;
    h->max_blocks = 3 * bw * bh;
    h->max_chunks_cap = (int)(max_bytes / CHUNK) + 3;
    const size_t n = (size_t)max_frames;
    if (!chk(hipMalloc(&h->d_bits, max_bytes + 64), "hipMalloc bitstream")) return PA_ERR_HIP;
    if (!chk(hipMalloc(&h->d_fd, n * sizeof(FrameDesc)), "hipMalloc descriptors")) return PA_ERR_HIP;
    if (!chk(hipMalloc(&h->d_ts, n * sizeof(TableSet)), "hipMalloc tables")) return PA_ERR_HIP;
    if (!chk(hipMalloc(&h->d_chunk, (n * h->max_chunks_cap) * sizeof(int32_t)), "hipMalloc chunk counts")) return PA_ERR_HIP;
    if (!chk(hipMalloc(&h->d_rst, n * bw * bh * sizeof(uint32_t)), "hipMalloc restart positions")) return PA_ERR_HIP;
    if (!chk(hipMalloc(&h->d_coef, n * h->max_blocks * 64 * sizeof(int16_t)), "hipMalloc coefficients")) return PA_ERR_HIP;
    if (!chk(hipMalloc(&h->d_planes, n * h->max_blocks * 64), "hipMalloc sample planes")) return PA_ERR_HIP;
    if (!chk(hipMalloc(&h->d_status, n * sizeof(int32_t)), "hipMalloc status")) return PA_ERR_HIP;
    if (!chk(hipMemset(h->d_bits, 0, max_bytes + 64), "hipMemset")) return PA_ERR_HIP;
    for (int i = 0; i < 2; ++i) {
        if (!chk(hipHostMalloc(&h->h_fd[i], n * sizeof(FrameDesc)), "hipHostMalloc")) return PA_ERR_HIP;
        if (!chk(hipHostMalloc(&h->h_ts[i], n * sizeof(TableSet)), "hipHostMalloc")) return PA_ERR_HIP;
        if (!chk(hipEventCreateWithFlags(&h->staged[i], hipEventDisableTiming), "hipEventCreate")) return PA_ERR_HIP;
    }
    return PA_OK;
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
    if (!chk(hipSetDevice(h->device), "hipSetDevice")) return PA_ERR_HIP;
    const int k = h->turn;
    h->turn ^= 1;
    if (h->staged_used[k] && !chk(hipEventSynchronize(h->staged[k]), "hipEventSynchronize")) return PA_ERR_HIP;
    FrameDesc* fd = h->h_fd[k];
    TableSet* ts = h->h_ts[k];
    Geom g;
    memset(&g, 0, sizeof g);
    Parsed first, prev, cur;
    int n_sets = 0, int_total = 0;
    uint32_t max_scan = 0;
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
            for (int t = 0; t < 4; ++t) {
                if (cur.hdef[t]) build_hufftab(T.h[t], cur.counts[t], cur.syms[t]);
                else memset(&T.h[t], 0, sizeof(HuffTab));
            }
            memcpy(T.q, cur.q, sizeof T.q);
        }
        prev = cur;
        FrameDesc& d = fd[f];
        memset(&d, 0, sizeof d);
        d.scan_off = (uint32_t)(o - base + (int64_t)cur.scan_off);
        d.scan_len = (uint32_t)((e - o) - (int64_t)cur.scan_off);
        d.ri = cur.ri;
        const int mcus = g.mcus_x * g.mcus_y;
        d.n_int = cur.ri ? (mcus + cur.ri - 1) / cur.ri : 1;
        d.int_base = int_total;
        int_total += d.n_int;
        d.tabset = n_sets - 1;
        for (int c = 0; c < cur.ncomp; ++c) {
            d.td[c] = (uint8_t)cur.td[c]; d.ta[c] = (uint8_t)cur.ta[c]; d.tq[c] = (uint8_t)cur.tq[c];
        }
        max_scan = d.scan_len > max_scan ? d.scan_len : max_scan;
    }
    const int max_chunks = (int)(max_scan / CHUNK) + 2;
    if (max_chunks > h->max_chunks_cap) return bad(PA_ERR_CAPACITY, "pa_mjpeg_decode: scan longer than the handle's chunk table");
    int max_int = 1;
    for (int f = 0; f < n; ++f) max_int = fd[f].n_int > max_int ? fd[f].n_int : max_int;
    if (!chk(hipMemcpyAsync(h->d_bits, data_host + base, (size_t)total, hipMemcpyHostToDevice, s), "upload bitstream")) return PA_ERR_HIP;
    if (!chk(hipMemsetAsync(h->d_bits + total, 0, 64, s), "pad bitstream")) return PA_ERR_HIP;
    if (!chk(hipMemcpyAsync(h->d_fd, fd, (size_t)n * sizeof(FrameDesc), hipMemcpyHostToDevice, s), "upload descriptors")) return PA_ERR_HIP;
    if (!chk(hipMemcpyAsync(h->d_ts, ts, (size_t)n_sets * sizeof(TableSet), hipMemcpyHostToDevice, s), "upload tables")) return PA_ERR_HIP;
    if (!chk(hipEventRecord(h->staged[k], s), "hipEventRecord")) return PA_ERR_HIP;
    h->staged_used[k] = true;
    if (!chk(hipMemsetAsync(h->d_status, 0, (size_t)n * sizeof(int32_t), s), "clear status")) return PA_ERR_HIP;
    if (!chk(hipMemsetAsync(h->d_coef, 0, (size_t)n * g.blocks_per_frame * 64 * sizeof(int16_t), s), "clear coefficients")) return PA_ERR_HIP;
    bool any_ri = false;
    for (int f = 0; f < n; ++f) any_ri = any_ri || fd[f].ri != 0;
    if (any_ri) {
        hipLaunchKernelGGL(rst_count_kernel, dim3(max_chunks, n), dim3(256), 0, s, h->d_bits, h->d_fd, h->d_chunk, max_chunks);
        hipLaunchKernelGGL(rst_write_kernel, dim3(max_chunks, n), dim3(256), 0, s, h->d_bits, h->d_fd, h->d_chunk, max_chunks, h->d_rst,
                           h->d_status);
    }
    hipLaunchKernelGGL(huff_kernel, dim3((max_int + 63) / 64, n), dim3(64), 0, s, h->d_bits, h->d_fd, h->d_ts, h->d_rst, g, h->d_coef,
                       h->d_status);
    const long long nblk = (long long)n * g.blocks_per_frame;
    hipLaunchKernelGGL(idct_kernel, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, s, h->d_coef, h->d_fd, h->d_ts, g, h->d_planes, n);
    const int fv = g.ncomp == 3 ? g.fv : 1, fhh = g.ncomp == 3 ? g.fh : 1;
    const dim3 grid((width + 511) / 512, (height + 4 * fv - 1) / (4 * fv), n);
    if (fhh == 2 && fv == 2) hipLaunchKernelGGL((ycc_kernel<2, 2>), grid, dim3(256), 0, s, h->d_planes, g, frames_dev, rgb);
    else if (fhh == 2) hipLaunchKernelGGL((ycc_kernel<2, 1>), grid, dim3(256), 0, s, h->d_planes, g, frames_dev, rgb);
    else hipLaunchKernelGGL((ycc_kernel<1, 1>), grid, dim3(256), 0, s, h->d_planes, g, frames_dev, rgb);
    if (status_dev && !chk(hipMemcpyAsync(status_dev, h->d_status, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, s), "copy status"))
        return PA_ERR_HIP;
    if (!chk(hipGetLastError(), "kernel launch")) return PA_ERR_HIP;
    h->last_height = height; h->last_width = width;
    return PA_OK;
}

}  // extern "C"
