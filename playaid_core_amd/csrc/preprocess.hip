// Crop preprocessing on gfx950: frame + normalised box -> 128x128 model input.
//
// Device-side equivalent of YoloCrop.square_crop(frame, 128, padding=30)
// (playaid/fighter.py:323-381) followed by the BGR2RGB + /255 of
// playaid/ai_runner.py:448,463. The reference composes three third-party
// resamplers; each is reproduced bit-for-bit in integer / IEEE arithmetic:
//
//   1. numpy slice of the padded square      (fighter.py:335-343)
//   2. Pillow ImageOps.pad(slice, (d,d))     (fighter.py:346-357)
//        = ImageOps.contain (aspect-preserving BICUBIC resize, 22-bit fixed
//          point, horizontal pass then vertical pass, u8 between the passes)
//          pasted centred on a black d x d canvas
//   3. imutils.resize(width=128) = cv2.resize(INTER_AREA) to (128, int(d*(128/d)))
//        (fighter.py:364) -- copy / 2x2 / integer-scale / fractional paths
//   4. ImageOps.pad to 128x128 when step 3 produced 127 rows (fighter.py:369-373)
//
// This file is compiled with -ffp-contract=off: the coefficient tables are
// computed in double precision exactly as Pillow's precompute_coeffs does, and
// the INTER_AREA fractional path accumulates in fp32 with separate multiply
// and add like OpenCV's generic C++ -- a fused multiply-add would change bits.
//
// The stage is HBM/L2 streaming work (no matrix shape): one thread per output
// element, consecutive lanes on consecutive bytes. Five small kernels keep every
// size general (any box, any frame): a fused LDS kernel for crops whose bands fit 48 KB,
// a multi-pass fallback over global scratch for the rest.
#include "pa_kernels.h"
#ifdef PA_STAMP_BUILD
#include <cstdio>
#include <vector>
#endif
#include "../../include/playaid_hip.h"
#include <cstdlib>

namespace pa {

#define PRECISION_BITS 22
#define COEF_ROW (2 + PA_KSIZE_MAX)

__device__ __forceinline__ bool to_int_checked(double v, int* out) {
    if (!(v > -2.0e9 && v < 2.0e9)) return false;  // also rejects NaN
    *out = (int)v;                                 // C cast == Python int(): truncation
    return true;
}

// numpy basic-slice length for image[start:stop] with start >= 0.
__device__ __forceinline__ void np_slice(int start, int stop, int size, int* s0, int* len) {
    if (start > size) start = size;
    if (stop < 0) {
        stop += size;
        if (stop < 0) stop = 0;
    }
    if (stop > size) stop = size;
    *s0 = start;
    *len = stop > start ? stop - start : 0;
}

__device__ __forceinline__ int bicubic_ksize(int in_size, int out_size) {
    double scale = (double)(float)in_size / out_size;
    double filterscale = scale < 1.0 ? 1.0 : scale;
    return (int)ceil(2.0 * filterscale) * 2 + 1;
}


// Pillow precompute_coeffs bounds (first tap, tap count) of output coordinate
// xx for a pass in_size -> out_size: the same arithmetic as bicubic_coef_row.
__device__ __forceinline__ void bicubic_bounds(int in_size, int out_size, int xx, int* xmin, int* cnt) {
    const double scale = (double)(float)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const double center = 0.0 + (xx + 0.5) * scale;
    int lo = (int)(center - support + 0.5);
    if (lo < 0) lo = 0;
    int hi = (int)(center + support + 0.5);
    if (hi > in_size) hi = in_size;
    *xmin = lo;
    *cnt = hi - lo;
}

// computeResizeAreaTab for one destination coordinate.
struct AreaTab {
    int s_first;     // source index of entry 0
    int n;           // number of entries
    float a_first;   // alpha of a leading partial cell (if has_first)
    float a_mid;     // alpha of the full cells
    float a_last;    // alpha of a trailing partial cell (if has_last)
    int has_first, n_mid, has_last;
};

__device__ __forceinline__ AreaTab area_tab(int dx, double scale, int ssize) {
    AreaTab t;
    const double fsx1 = dx * scale;
    const double fsx2 = fsx1 + scale;
    const double cell = fmin(scale, ssize - fsx1);
    int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
    sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
    sx1 = sx1 < sx2 ? sx1 : sx2;
    t.has_first = (sx1 - fsx1 > 1e-3) ? 1 : 0;
    t.a_first = (float)((sx1 - fsx1) / cell);
    t.n_mid = sx2 - sx1;
    t.a_mid = (float)(1.0 / cell);
    t.has_last = (fsx2 - sx2 > 1e-3) ? 1 : 0;
    t.a_last = (float)(fmin(fmin(fsx2 - sx2, 1.0), cell) / cell);
    t.s_first = sx1 - t.has_first;
    t.n = t.has_first + t.n_mid + t.has_last;
    return t;
}

// 16-byte LDS form of an AreaTab (s_first < 2^16, n_mid < 2^8 for any square side the plan accepts)
__device__ __forceinline__ AreaTabPacked area_pack(const AreaTab& t) {
    AreaTabPacked q;
    q.bits = (uint32_t)t.s_first | ((uint32_t)t.n_mid << 16) | ((uint32_t)t.has_first << 24) | ((uint32_t)t.has_last << 25);
    q.a_first = t.a_first;
    q.a_mid = t.a_mid;
    q.a_last = t.a_last;
    return q;
}

__device__ __forceinline__ AreaTab area_unpack(const AreaTabPacked& q) {
    AreaTab t;
    t.s_first = (int)(q.bits & 0xffff);
    t.n_mid = (int)((q.bits >> 16) & 0xff);
    t.has_first = (int)((q.bits >> 24) & 1);
    t.has_last = (int)((q.bits >> 25) & 1);
    t.n = t.has_first + t.n_mid + t.has_last;
    t.a_first = q.a_first;
    t.a_mid = q.a_mid;
    t.a_last = q.a_last;
    return t;
}

// (by value, as selects of values: taking the table by reference made hipcc keep it in scratch memory and turn the
// choice into an indexed scratch load -- which gave the whole fused kernel a private segment)
__device__ __forceinline__ float area_alpha(const AreaTab t, int k) {
    const float a_first = t.a_first, a_mid = t.a_mid, a_last = t.a_last;
    const int hf = t.has_first, hm = t.has_first + t.n_mid;
    float r = a_last;
    r = k < hm ? a_mid : r;
    r = k < hf ? a_first : r;
    return r;
}


// Rows each stage must hold in LDS to produce output rows [r0, r1) of the
// 128-row crop: canvas rows [dy0,dy1) -> resized rows [ry0,ry1) -> slice rows
// [ty0,ty1) (the vertical pass's taps).
struct BandRows {
    int dy0, dy1, ry0, ry1, ty0, ty1;
};

__device__ __forceinline__ BandRows band_rows(const CropPlan& pl, int r0, int r1) {
    BandRows b;
    b.dy0 = b.dy1 = b.ry0 = b.ry1 = b.ty0 = b.ty1 = 0;
    if (r1 > pl.out_h) r1 = pl.out_h;
    if (r0 >= r1) return b;
    if (pl.area_mode == 0) {
        b.dy0 = r0;
        b.dy1 = r1;
    } else if (pl.area_mode == 1) {
        b.dy0 = 2 * r0;
        b.dy1 = 2 * r1;
    } else if (pl.area_mode == 2) {
        b.dy0 = r0 * pl.iscale_y;
        b.dy1 = r1 * pl.iscale_y;
    } else {
        const AreaTab t0 = area_tab(r0, pl.scale_y, pl.d);
        const AreaTab t1 = area_tab(r1 - 1, pl.scale_y, pl.d);
        b.dy0 = t0.s_first;
        b.dy1 = t1.s_first + t1.n;
    }
    int ry0 = b.dy0 - pl.py, ry1 = b.dy1 - pl.py;
    ry0 = ry0 < 0 ? 0 : (ry0 > pl.rh ? pl.rh : ry0);
    ry1 = ry1 < 0 ? 0 : (ry1 > pl.rh ? pl.rh : ry1);
    b.ry0 = ry0;
    b.ry1 = ry1;
    if (ry1 <= ry0) return b;
    if (pl.need_v) {
        int ymin, cnt;
        bicubic_bounds(pl.sh, pl.rh, ry0, &ymin, &cnt);
        b.ty0 = ymin;
        bicubic_bounds(pl.sh, pl.rh, ry1 - 1, &ymin, &cnt);
        b.ty1 = ymin + cnt;
    } else {
        b.ty0 = ry0;
        b.ty1 = ry1;
    }
    return b;
}

__device__ __forceinline__ int kk_dbg(const int32_t* row, int t) { return row[2 + t]; }

// First byte and row pitch of a crop's slice: inside the whole frame, or -- window ingest (pa_preprocess_windows) --
// inside the packed copy of just that slice that the host uploaded.
__device__ __forceinline__ const uint8_t* slice_ptr(const PreprocParams& p, int crop, const CropPlan& pl, size_t* pitch) {
    if (p.windows) {
        *pitch = (size_t)p.windows[crop].pitch;
        return p.frames + p.windows[crop].offset;
    }
    *pitch = (size_t)p.width * 3;
    return p.frames + ((size_t)pl.frame * p.height + pl.sy0) * p.width * 3 + (size_t)pl.sx0 * 3;
}
__device__ __forceinline__ int align_up(int v, int a) { return (v + a - 1) / a * a; }

// LDS layout of one sub-band: B0 source rows | B1 after the horizontal pass; B2 (after the
// vertical pass) reuses B0's space when both passes run (B0 is dead once H is done).
struct BandLds {
    int p0, p1;          // row pitches in bytes (multiples of 4)
    int off1, off2, total;
};

__device__ __forceinline__ BandLds band_lds(const CropPlan& pl, const BandRows& b) {
    BandLds l;
    l.p0 = align_up(pl.sw * 3, 4) + 4;
    l.p1 = align_up(pl.rw * 3, 4) + 4;
    const int n0 = b.ty1 - b.ty0, n2 = b.ry1 - b.ry0;
    const int s0 = align_up(n0 * l.p0, 16), s1 = align_up(n0 * l.p1, 16), s2 = align_up(n2 * l.p1, 16);
    l.off1 = s0;
    if (pl.need_h && pl.need_v) {
        l.off2 = 0;
        l.total = s0 + s1 > s2 ? s0 + s1 : s2;
    } else if (pl.need_h) {
        l.off2 = 0;  // unused
        l.total = s0 + s1;
    } else if (pl.need_v) {
        l.off2 = s0;
        l.total = s0 + s2;
    } else {
        l.off2 = 0;
        l.total = s0;
    }
    return l;
}

__device__ __forceinline__ double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// Pillow precompute_coeffs + normalize_coeffs_8bpc for output coordinate xx of a pass in_size -> out_size
// into row[0] = first tap, row[1] = tap count, row[2..] = coefficients.
// (out of line: the plan kernel runs once per clip with its code cold, every line of it fetched from memory, and the
// passes of almost every crop are in the engine's cache)
__device__ __noinline__ void bicubic_coef_row(int in_size, int out_size, int xx, int32_t* row) {
    const double scale = (double)(float)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const double ss = 1.0 / filterscale;
    const double center = 0.0 + (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double k[PA_KSIZE_MAX];
    double ww = 0.0;
    for (int x = 0; x < PA_KSIZE_MAX; ++x) {
        double w = 0.0;
        if (x < xmax) {
            w = bicubic_filter((x + xmin - center + 0.5) * ss);
            ww += w;
        }
        k[x] = w;
    }
    row[0] = xmin;
    row[1] = xmax;
    for (int x = 0; x < PA_KSIZE_MAX; ++x) {
        double v = k[x];
        if (x < xmax && ww != 0.0) v = v / ww;
        row[2 + x] = v < 0 ? (int)(-0.5 + v * (double)(1 << PRECISION_BITS)) : (int)(0.5 + v * (double)(1 << PRECISION_BITS));
    }
}

// One wave per crop: every lane derives the geometry (cheap, identical), then the lanes
// share the search for the largest LDS sub-band (128 candidate sub-bands in parallel).
// Workgroup = one crop: wave 0 plans, then all four waves fill the crop's two Pillow coefficient tables.
__global__ __launch_bounds__(256) void crop_plan_kernel(const PreprocParams p) {
    __shared__ CropPlan plan_sh;
    const int crop = blockIdx.x;
    if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    CropPlan pl;
    pl.status = PA_CROP_OK;
    pl.frame = p.src_frame ? p.src_frame[crop] : crop / p.fighters;
    pl.sx0 = pl.sy0 = pl.sw = pl.sh = 0;
    pl.d = pl.rw = pl.rh = pl.px = pl.py = 0;
    pl.need_h = pl.need_v = pl.ksize_h = pl.ksize_v = 0;
    pl.out_h = 0;
    pl.area_mode = 0;
    pl.iscale_x = pl.iscale_y = 1;
    pl.fused_rb = 0;
    pl.pad_ = 0;
    pl.coef_h = pl.coef_v = nullptr;
    pl.scale_x = pl.scale_y = 1.0;
    const double* b = p.boxes + (size_t)crop * 4;
    const int W = p.width, H = p.height, pad = p.padding;
    if (!p.windows && (unsigned)pl.frame >= (unsigned)p.n_src) {  // a source index outside the frame buffer: never dereferenced
        pl.frame = 0;
        pl.status = PA_CROP_BAD_FRAME;
    }
    int cx, cy, cw, ch;
    // YoloCrop.yolo_pixels (fighter.py:305-314)
    if (pl.status == PA_CROP_OK && (!to_int_checked(b[0] * W, &cx) || !to_int_checked(b[1] * H, &cy) ||
                                    !to_int_checked(b[2] * W, &cw) || !to_int_checked(b[3] * H, &ch))) {
        pl.status = PA_CROP_BAD_BOX;
    }
    if (pl.status == PA_CROP_OK) {
        const int d = cw > ch ? cw : ch;
        pl.d = d;
        if (d <= 0 || d > 16384) pl.status = PA_CROP_BAD_BOX;
    }
    if (pl.status == PA_CROP_OK) {
        const int d = pl.d;
        const int half = d / 2;  // int(square_dim / 2)
        int y0 = cy - half - pad, y1 = cy + half + pad, x0 = cx - half - pad, x1 = cx + half + pad;
        y0 = y0 > 0 ? y0 : 0;
        x0 = x0 > 0 ? x0 : 0;
        y1 = y1 < H ? y1 : H;
        x1 = x1 < W ? x1 : W;
        np_slice(y0, y1, H, &pl.sy0, &pl.sh);
        np_slice(x0, x1, W, &pl.sx0, &pl.sw);
        if (pl.sh != d || pl.sw != d) {
            // ImageOps.pad(raw_crop, (d, d), color="black")
            if (pl.sh == 0 || pl.sw == 0) {
                pl.status = PA_CROP_EMPTY;
            } else {
                int rw = d, rh = d;
                const double im_ratio = (double)pl.sw / (double)pl.sh;
                if (im_ratio != 1.0) {
                    if (im_ratio > 1.0) {
                        const int nh = (int)rint((double)pl.sh / (double)pl.sw * (double)d);
                        if (nh != d) rh = nh;
                    } else {
                        const int nw = (int)rint((double)pl.sw / (double)pl.sh * (double)d);
                        if (nw != d) rw = nw;
                    }
                }
                if (rw <= 0 || rh <= 0) {
                    pl.status = PA_CROP_EMPTY;
                } else {
                    pl.rw = rw;
                    pl.rh = rh;
                    pl.need_h = rw != pl.sw;
                    pl.need_v = rh != pl.sh;
                    if (rw != d)
                        pl.px = (int)rint((double)(d - rw) * 0.5);
                    else if (rh != d)
                        pl.py = (int)rint((double)(d - rh) * 0.5);
                    if (pl.need_h) pl.ksize_h = bicubic_ksize(pl.sw, rw);
                    if (pl.need_v) pl.ksize_v = bicubic_ksize(pl.sh, rh);
                    if (pl.ksize_h > PA_KSIZE_MAX || pl.ksize_v > PA_KSIZE_MAX) pl.status = PA_CROP_FILTER_TOO_WIDE;
                    if ((size_t)pl.sh * rw * 3 > p.t_stride || (size_t)rh * rw * 3 > p.t_stride || rw > p.coef_dim ||
                        rh > p.coef_dim)
                        pl.status = PA_CROP_FILTER_TOO_WIDE;
                }
            }
        } else {
            pl.rw = pl.rh = d;
        }
    }
    if (pl.status == PA_CROP_OK) {
        const int d = pl.d;
        {
            // imutils.resize(width=128): dim = (128, int(h * (128 / float(w))))
            const double r = 128.0 / (double)d;
            pl.out_h = (int)((double)d * r);
            const double inv_sx = 128.0 / (double)d;
            const double inv_sy = (double)pl.out_h / (double)d;
            pl.scale_x = 1.0 / inv_sx;
            pl.scale_y = 1.0 / inv_sy;
            if (pl.scale_x < 1.0 || pl.scale_y < 1.0) {
                // source smaller than 128 px: cv::resize emulates INTER_AREA with its fixed-point
                // bilinear resizer (area_pixel mode 4); rare, served by the fallback kernel only
                pl.area_mode = 4;
            } else if (d == PA_CROP && pl.out_h == PA_CROP) {
                pl.area_mode = 0;
            } else {
                pl.iscale_x = (int)rint(pl.scale_x);  // saturate_cast<int>(double) == cvRound
                pl.iscale_y = (int)rint(pl.scale_y);
                const bool fast = fabs(pl.scale_x - pl.iscale_x) < 2.220446049250313e-16 &&
                                  fabs(pl.scale_y - pl.iscale_y) < 2.220446049250313e-16;
                pl.area_mode = fast ? ((pl.iscale_x == 2 && pl.iscale_y == 2) ? 1 : 2) : 3;
            }
        }
    }
    if (p.ablate & 16) pl.fused_rb = 2;
    if (pl.status == PA_CROP_OK && pl.area_mode != 4 && !(p.ablate & 16)) {
        // largest sub-band height whose LDS stages fit the fused kernel's budget
#pragma unroll 1
        for (int rb = 8; rb >= 1 && pl.fused_rb == 0; rb >>= 1) {
            int worst = 0;
#pragma unroll 1
            for (int r0 = lane * rb; r0 < PA_CROP; r0 += 64 * rb) {
                const BandRows b = band_rows(pl, r0, r0 + rb);
                const int need = band_lds(pl, b).total;
                worst = need > worst ? need : worst;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const int o = __shfl_xor(worst, d, 64);
                worst = o > worst ? o : worst;
            }
            if (worst <= p.fused_lds) pl.fused_rb = rb;
        }
    }
    if (lane == 0) {
        if (pl.status == PA_CROP_OK && pl.fused_rb == 0) {
            const int slot = atomicAdd(p.fallback_count, 1);
            p.fallback_list[slot] = crop;
        }
        // coefficient tables: a pass (2 * (d / 2) + 2 * padding -> d) -- the slice of every crop the frame edge does not
        // clip (fighter.py:334-343: centre -/+ int(d / 2) -/+ padding) -- reads the engine's cache; any other pair is
        // computed below into this crop's own rows
        for (int axis = 0; axis < 2; ++axis) {
            const int in_size = axis ? pl.sh : pl.sw, out_size = axis ? pl.rh : pl.rw;
            const bool cached = p.coef_cache && in_size == 2 * (out_size / 2) + 2 * p.coef_cache_pad && out_size >= 1 &&
                                out_size <= p.coef_cache_dmax;
            const int32_t* tab = cached ? p.coef_cache + (size_t)out_size * (out_size - 1) / 2 * COEF_ROW
                                        : p.coef + (size_t)(crop * 2 + axis) * p.coef_dim * COEF_ROW;
            if (axis) pl.coef_v = tab; else pl.coef_h = tab;
        }
        p.plans[crop] = pl;
        plan_sh = pl;
        if (p.status) p.status[crop] = pl.status;
    }
    }
    __syncthreads();
    const CropPlan pl = plan_sh;
    if (pl.status != PA_CROP_OK) return;
    // cv::computeResizeAreaTab for the 128 destination columns and rows (fp64 with divisions): once per crop, not once
    // per workgroup of the fused kernel
    if (pl.area_mode == 3 && p.area_tabs && !(p.ablate & 32)) {
        const int t = threadIdx.x;
        AreaTabPacked q = {0u, 0.f, 0.f, 0.f};
        if (t < PA_CROP) q = area_pack(area_tab(t, pl.scale_x, pl.d));
        else if (t - PA_CROP < pl.out_h) q = area_pack(area_tab(t - PA_CROP, pl.scale_y, pl.d));
        p.area_tabs[(size_t)crop * 2 * PA_CROP + t] = q;
    }
    // Pillow precompute_coeffs + normalize_coeffs_8bpc for the passes the cache does not hold (one thread per output coordinate)
    for (int axis = 0; axis < 2; ++axis) {
        if (!(axis ? pl.need_v : pl.need_h)) continue;
        const int in_size = axis ? pl.sh : pl.sw, out_size = axis ? pl.rh : pl.rw;
        int32_t* own = p.coef + (size_t)(crop * 2 + axis) * p.coef_dim * COEF_ROW;
        if ((axis ? pl.coef_v : pl.coef_h) != own) continue;
        for (int xx = threadIdx.x; xx < out_size; xx += 256) bicubic_coef_row(in_size, out_size, xx, own + (size_t)xx * COEF_ROW);
    }
}

// The engine's coefficient cache: table d = the pass (2 * (d / 2) + 2 * padding -> d), at row d * (d - 1) / 2.
__global__ __launch_bounds__(256) void coef_cache_kernel(int32_t* __restrict__ cache, int padding, int dmax) {
    const int d = blockIdx.y + 1, xx = blockIdx.x * 256 + threadIdx.x;
    if (d > dmax || xx >= d) return;
    bicubic_coef_row(2 * (d / 2) + 2 * padding, d, xx, cache + ((size_t)d * (d - 1) / 2 + xx) * COEF_ROW);
}

// Pillow clip8: (acc >> 22) clamped to 0..255. Returns a 32-bit value on purpose: with a
// uint8_t return type hipcc 7.2 packed the four bytes of the vertical pass through 16-bit
// v_bitop3_b16 operations and bytes 2/3 of each dword came out wrong on gfx950.
__device__ __forceinline__ uint32_t clip8(int v) {
    v >>= PRECISION_BITS;
    return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// ---------------------------------------------------------------------------
// Multi-pass fallback for crops whose bands exceed the fused kernel's LDS budget (very
// large boxes): same arithmetic, intermediates in global scratch (t1, t2). One workgroup
// walks the compact list the plan kernel built and runs the three passes of a crop back to
// back (workgroup barrier + fence between them), so the common case -- empty list -- costs
// one tiny launch.
// ImagingResampleHorizontal_8bpc over the slice rows: t1[y][xx][c].
__device__ void fallback_h(const PreprocParams& p, int crop, const CropPlan& pl) {
    if (!pl.need_h) return;
    const int total = pl.sh * pl.rw;
    size_t src_pitch;
    const uint8_t* src = slice_ptr(p, crop, pl, &src_pitch);
    uint8_t* dst = p.t1 + (size_t)crop * p.t_stride;
    const int32_t* coef = pl.coef_h;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int y = i / pl.rw;
        const int xx = i - y * pl.rw;
        const int32_t* row = coef + (size_t)xx * COEF_ROW;
        const int xmin = row[0], cnt = row[1];
        const uint8_t* s = src + (size_t)y * src_pitch + (size_t)xmin * 3;
        int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
        for (int x = 0; x < cnt; ++x) {
            const int k = row[2 + x];
            a0 += __mul24((int)s[3 * x + 0], k);  // u8 x 22-bit fixed point: fits v_mad_i32_i24
            a1 += __mul24((int)s[3 * x + 1], k);
            a2 += __mul24((int)s[3 * x + 2], k);
        }
        uint8_t* o = dst + (size_t)i * 3;
        o[0] = (uint8_t)clip8(a0);
        o[1] = (uint8_t)clip8(a1);
        o[2] = (uint8_t)clip8(a2);
    }
}

// ImagingResampleVertical_8bpc: t2[yy][x][c] from t1 (or the slice when no horizontal pass ran).
__device__ void fallback_v(const PreprocParams& p, int crop, const CropPlan& pl) {
    if (!pl.need_v) return;
    const int total = pl.rh * pl.rw;
    const uint8_t* src;
    size_t src_pitch;
    if (pl.need_h) {
        src = p.t1 + (size_t)crop * p.t_stride;
        src_pitch = (size_t)pl.rw * 3;
    } else {
        src = slice_ptr(p, crop, pl, &src_pitch);
    }
    uint8_t* dst = p.t2 + (size_t)crop * p.t_stride;
    const int32_t* coef = pl.coef_v;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int yy = i / pl.rw;
        const int x = i - yy * pl.rw;
        const int32_t* row = coef + (size_t)yy * COEF_ROW;
        const int ymin = row[0], cnt = row[1];
        const uint8_t* s = src + (size_t)ymin * src_pitch + (size_t)x * 3;
        int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
        for (int y = 0; y < cnt; ++y) {
            const int k = row[2 + y];
            a0 += __mul24((int)s[0], k);
            a1 += __mul24((int)s[1], k);
            a2 += __mul24((int)s[2], k);
            s += src_pitch;
        }
        uint8_t* o = dst + (size_t)i * 3;
        o[0] = (uint8_t)clip8(a0);
        o[1] = (uint8_t)clip8(a1);
        o[2] = (uint8_t)clip8(a2);
    }
}

struct Canvas {
    const uint8_t* src;
    size_t pitch;
    int px, py, rw, rh;
    // d x d black canvas with the resized slice pasted at (px, py)
    __device__ __forceinline__ void load(int y, int x, int& c0, int& c1, int& c2) const {
        y -= py;
        x -= px;
        if ((unsigned)y < (unsigned)rh && (unsigned)x < (unsigned)rw) {
            const uint8_t* s = src + (size_t)y * pitch + (size_t)x * 3;
            c0 = s[0];
            c1 = s[1];
            c2 = s[2];
        } else {
            c0 = c1 = c2 = 0;
        }
    }
};

__device__ __forceinline__ int cv_saturate_u8(float v) {
    const int r = (int)rintf(v);  // cvRound: round half to even
    return r < 0 ? 0 : (r > 255 ? 255 : r);
}

// One destination pixel of cv::resize INTER_AREA (d x d canvas -> out_h x 128).
// CV::load(y, x, c0, c1, c2) returns the canvas pixel (black outside the paste).
template <class CV>
__device__ __forceinline__ void area_pixel(const CropPlan& pl, const CV& cv, int dy, int dx, int& o0, int& o1, int& o2) {
    if (pl.area_mode == 0) {
        cv.load(dy, dx, o0, o1, o2);
    } else if (pl.area_mode == 1) {
        int s0 = 2, s1 = 2, s2 = 2;
        for (int yy = 0; yy < 2; ++yy)
            for (int xx = 0; xx < 2; ++xx) {
                int c0, c1, c2;
                cv.load(dy * 2 + yy, dx * 2 + xx, c0, c1, c2);
                s0 += c0; s1 += c1; s2 += c2;
            }
        o0 = s0 >> 2; o1 = s1 >> 2; o2 = s2 >> 2;
    } else if (pl.area_mode == 2) {
        int s0 = 0, s1 = 0, s2 = 0;
        for (int yy = 0; yy < pl.iscale_y; ++yy)
            for (int xx = 0; xx < pl.iscale_x; ++xx) {
                int c0, c1, c2;
                cv.load(dy * pl.iscale_y + yy, dx * pl.iscale_x + xx, c0, c1, c2);
                s0 += c0; s1 += c1; s2 += c2;
            }
        const float scale = 1.f / (float)(pl.iscale_x * pl.iscale_y);
        o0 = cv_saturate_u8((float)s0 * scale);
        o1 = cv_saturate_u8((float)s1 * scale);
        o2 = cv_saturate_u8((float)s2 * scale);
    } else if (pl.area_mode == 4) {
        // enlarging: cv::hal::resize runs the 8-bit bilinear resizer with area-mode coefficients
        // (s = floor(d*scale), f = (d+1) - (s+1)*inv_scale, f <= 0 ? 0 : f - floor(f); weights
        // cvRound(w * 2048)); HResizeLinear keeps 11 fraction bits, VResizeLinear computes
        // (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2. Columns whose right neighbour
        // would leave the source use S[last]*2048; rows only clip the index.
        const int ss = pl.d;
        const double inv_x = 128.0 / (double)ss, inv_y = (double)pl.out_h / (double)ss;  // dsize / ssize, as cv::resize forms them
        int sx = (int)floor((double)dx * pl.scale_x);
        float fx = (float)((double)(dx + 1) - (double)(sx + 1) * inv_x);
        fx = fx <= 0.f ? 0.f : fx - floorf(fx);
        const bool plain = sx + 1 >= ss;
        if (sx >= ss - 1) {
            fx = 0.f;
            sx = ss - 1;
        }
        const int a0 = (int)rintf((1.f - fx) * 2048.f), a1 = (int)rintf(fx * 2048.f);
        const int sy = (int)floor((double)dy * pl.scale_y);
        float fy = (float)((double)(dy + 1) - (double)(sy + 1) * inv_y);
        fy = fy <= 0.f ? 0.f : fy - floorf(fy);
        const int b0 = (int)rintf((1.f - fy) * 2048.f), b1 = (int)rintf(fy * 2048.f);
        const int r0 = sy < ss - 1 ? sy : ss - 1;
        const int r1 = sy + 1 < ss - 1 ? sy + 1 : ss - 1;
        int h0[3], h1[3];
        {
            int c0, c1, c2, e0 = 0, e1 = 0, e2 = 0;
            cv.load(r0, sx, c0, c1, c2);
            if (!plain) cv.load(r0, sx + 1, e0, e1, e2);
            h0[0] = plain ? c0 * 2048 : c0 * a0 + e0 * a1;
            h0[1] = plain ? c1 * 2048 : c1 * a0 + e1 * a1;
            h0[2] = plain ? c2 * 2048 : c2 * a0 + e2 * a1;
            cv.load(r1, sx, c0, c1, c2);
            if (!plain) cv.load(r1, sx + 1, e0, e1, e2);
            h1[0] = plain ? c0 * 2048 : c0 * a0 + e0 * a1;
            h1[1] = plain ? c1 * 2048 : c1 * a0 + e1 * a1;
            h1[2] = plain ? c2 * 2048 : c2 * a0 + e2 * a1;
        }
        o0 = ((((b0 * (h0[0] >> 4)) >> 16) + ((b1 * (h1[0] >> 4)) >> 16) + 2) >> 2) & 0xff;
        o1 = ((((b0 * (h0[1] >> 4)) >> 16) + ((b1 * (h1[1] >> 4)) >> 16) + 2) >> 2) & 0xff;
        o2 = ((((b0 * (h0[2] >> 4)) >> 16) + ((b1 * (h1[2] >> 4)) >> 16) + 2) >> 2) & 0xff;
    } else {
        const AreaTab tx = area_tab(dx, pl.scale_x, pl.d);
        const AreaTab ty = area_tab(dy, pl.scale_y, pl.d);
        float sum0 = 0.f, sum1 = 0.f, sum2 = 0.f;
        for (int j = 0; j < ty.n; ++j) {
            const float beta = area_alpha(ty, j);
            float b0 = 0.f, b1 = 0.f, b2 = 0.f;
            for (int k = 0; k < tx.n; ++k) {
                const float alpha = area_alpha(tx, k);
                int c0, c1, c2;
                cv.load(ty.s_first + j, tx.s_first + k, c0, c1, c2);
                b0 = b0 + (float)c0 * alpha;
                b1 = b1 + (float)c1 * alpha;
                b2 = b2 + (float)c2 * alpha;
            }
            sum0 = sum0 + beta * b0;
            sum1 = sum1 + beta * b1;
            sum2 = sum2 + beta * b2;
        }
        o0 = cv_saturate_u8(sum0);
        o1 = cv_saturate_u8(sum1);
        o2 = cv_saturate_u8(sum2);
    }
}

// channel swap + u8 crop + /255 fp32 zero-bordered NHWC4 model input for pixel i of `crop`
__device__ __forceinline__ void write_crop_pixel(const PreprocParams& p, int crop, int i, int o0, int o1, int o2) {
    if (p.swap_rb) {
        const int t = o0;
        o0 = o2;
        o2 = t;
    }
    if (p.crops_u8) {
        uint8_t* o = p.crops_u8 + ((size_t)crop * PA_CROP * PA_CROP + i) * 3;
        o[0] = (uint8_t)o0;
        o[1] = (uint8_t)o1;
        o[2] = (uint8_t)o2;
    }
    if (p.crops_f32) {
        float4 v;
        v.x = (float)o0 / 255.0f;
        v.y = (float)o1 / 255.0f;
        v.z = (float)o2 / 255.0f;
        v.w = 0.f;
        const int dy = i >> 7, dx = i & 127;
        const size_t o = ((size_t)crop * 134 + (dy + 3)) * 134 + (dx + 3);
        if (p.crops_f32_is_bf16) {  // bf16 conv path: the /255 quotient rounded to bf16 (nearest even)
            uint32_t u[3] = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z)};
#pragma unroll
            for (int k = 0; k < 3; ++k) u[k] = (u[k] + 0x7fffu + ((u[k] >> 16) & 1u)) >> 16;
            reinterpret_cast<uint2*>(p.crops_f32)[o] = make_uint2(u[0] | (u[1] << 16), u[2]);
        } else {
            reinterpret_cast<float4*>(p.crops_f32)[o] = v;
        }
    }
}

// INTER_AREA from the global-memory intermediates, final black pad to 128 rows.
__device__ void fallback_area(const PreprocParams& p, int crop, const CropPlan& pl) {
    Canvas cv;
    cv.px = pl.px; cv.py = pl.py; cv.rw = pl.rw; cv.rh = pl.rh;
    if (pl.need_v) {
        cv.src = p.t2 + (size_t)crop * p.t_stride;
        cv.pitch = (size_t)pl.rw * 3;
    } else if (pl.need_h) {
        cv.src = p.t1 + (size_t)crop * p.t_stride;
        cv.pitch = (size_t)pl.rw * 3;
    } else {
        cv.src = slice_ptr(p, crop, pl, &cv.pitch);
    }
    for (int i = threadIdx.x; i < PA_CROP * PA_CROP; i += blockDim.x) {
        const int dy = i >> 7, dx = i & 127;
        int o0 = 0, o1 = 0, o2 = 0;
        if (dy < pl.out_h) area_pixel(pl, cv, dy, dx, o0, o1, o2);
        write_crop_pixel(p, crop, i, o0, o1, o2);
    }
}

// Runs as the LAST row of crop_fused_kernel's grid (blockIdx.y == number of crops): no launch of its own.
__device__ void crop_fallback_body(const PreprocParams& p) {
    const int count = *p.fallback_count;
    for (int fb = blockIdx.x; fb < count; fb += gridDim.x) {
        const int crop = p.fallback_list[fb];
        const CropPlan pl = p.plans[crop];
        fallback_h(p, crop, pl);
        __threadfence();
        __syncthreads();
        fallback_v(p, crop, pl);
        __threadfence();
        __syncthreads();
        fallback_area(p, crop, pl);
    }
    // re-arm the list for the next call (every block has read `count`; the last one to leave resets it)
    __syncthreads();
    if (threadIdx.x == 0) {
        const int done = atomicAdd(p.fallback_count + 1, 1) + 1;
        if (done == (int)gridDim.x) {
            p.fallback_count[1] = 0;
            __threadfence();
            p.fallback_count[0] = 0;
        }
    }
}

// ---------------------------------------------------------------------------
// Fused path: one workgroup = 8 output rows of one crop, every intermediate in
// LDS. Per sub-band of fused_rb rows:
//   stage 0  slice rows [ty0,ty1) -> B0, dword loads, re-aligned with
//            v_alignbyte so every LDS row starts on a 4-byte boundary
//   stage H  Pillow horizontal pass B0 -> B1 (thread = output column, its
//            <= 15 fixed-point coefficients live in registers across the rows)
//   stage V  Pillow vertical pass B1 -> B2 (thread = 4 output bytes: one
//            ds_read_b32 per tap row feeds four accumulators)
//   stage A  INTER_AREA from B2 (black outside the paste) -> u8 crop + fp32 input
// The frame is read from HBM exactly once (about 0.42 MB per crop at 1080p).
extern __shared__ __attribute__((aligned(16))) uint8_t pa_smem[];

struct LdsCanvas {
    int base, pitch;  // byte offset of resized row ry0 in pa_smem, row pitch
    int px, py, rw, ry0, ry1;
    __device__ __forceinline__ void load(int y, int x, int& c0, int& c1, int& c2) const {
        y -= py;
        x -= px;
        if (y >= ry0 && y < ry1 && (unsigned)x < (unsigned)rw) {
            const int o = base + (y - ry0) * pitch + x * 3;
            c0 = pa_smem[o];
            c1 = pa_smem[o + 1];
            c2 = pa_smem[o + 2];
        } else {
            c0 = c1 = c2 = 0;
        }
    }
};

constexpr int CF_NT = 512;         // threads of a crop_fused_kernel workgroup
constexpr int CF_NW = CF_NT / 64;   // its waves
__global__ __launch_bounds__(CF_NT) void crop_fused_kernel(const PreprocParams p) {
#ifdef PA_STAMP_BUILD
    const unsigned long long st0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int crop = blockIdx.y;
    if (crop == p.n_frames * p.fighters) {  // the grid's extra row: crops whose bands do not fit the LDS budget
        crop_fallback_body(p);
        return;
    }
    const int band0 = blockIdx.x * 8;  // first of this workgroup's 8 output rows
    const int tid = threadIdx.x;
    const CropPlan pl = p.plans[crop];
    if (pl.status != PA_CROP_OK) {
        // failed crop: all-zero rows (the fallback kernels skip it as well)
        for (int i = tid; i < 8 * PA_CROP; i += CF_NT) write_crop_pixel(p, crop, band0 * PA_CROP + i, 0, 0, 0);
        return;
    }
    if (!pl.fused_rb) return;
    size_t frame_pitch;
    const uint8_t* slice = slice_ptr(p, crop, pl, &frame_pitch);
    const int32_t* coef_h = pl.coef_h;
    const int32_t* coef_v = pl.coef_v;
    const int rb = pl.fused_rb;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // computeResizeAreaTab entries for the 128 destination columns and this workgroup's 8
    // destination rows, computed once (fp64 with divisions) and kept behind the stage buffers
    AreaTabPacked* tabs = reinterpret_cast<AreaTabPacked*>(pa_smem + p.fused_lds);
    if (pl.area_mode == 3) {
        const AreaTabPacked* at = p.area_tabs + (size_t)crop * 2 * PA_CROP;  // the plan kernel's
        if (tid < PA_CROP) tabs[tid] = at[tid];
        else if (tid < PA_CROP + 8 && band0 + (tid - PA_CROP) < pl.out_h) tabs[tid] = at[PA_CROP + band0 + (tid - PA_CROP)];
    }
    __syncthreads();
    for (int r0 = band0; r0 < band0 + 8; r0 += rb) {
        const BandRows b = band_rows(pl, r0, r0 + rb);
        const BandLds L = band_lds(pl, b);
        const int n0 = b.ty1 - b.ty0, n2 = b.ry1 - b.ry0;
        // ---- stage 0: slice rows -> B0 (aligned dwords) ----------------------
        // wave w takes rows w, w+4, ...; its lanes take consecutive dwords of the row, four
        // independent 256-byte segments in flight per iteration (the loop is latency-bound).
        if (!(p.ablate & 1)) {
            const int row_dwords = (pl.sw * 3 + 3) >> 2;
            const int row_bytes = pl.sw * 3;
            for (int y = wave; y < n0; y += CF_NW) {
                const uintptr_t ga = (uintptr_t)(slice + (size_t)(b.ty0 + y) * frame_pitch);
                const uint32_t* g4 = reinterpret_cast<const uint32_t*>(ga & ~(uintptr_t)3);
                const uint32_t sh = (uint32_t)(ga & 3);  // wave-uniform misalignment of this row
                uint32_t* dst = reinterpret_cast<uint32_t*>(pa_smem + y * L.p0);
                for (int j0 = 0; j0 < row_dwords; j0 += 256) {
                    uint32_t lo[4], hi[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int j = j0 + lane + 64 * u;
                        lo[u] = j < row_dwords ? g4[j] : 0u;
                        // the next dword is only dereferenced when the row really straddles it
                        hi[u] = (sh && j < row_dwords && 4 * (j + 1) < (int)sh + row_bytes) ? g4[j + 1] : 0u;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int j = j0 + lane + 64 * u;
                        if (j < row_dwords) dst[j] = __builtin_amdgcn_alignbyte(hi[u], lo[u], sh);
                    }
                }
            }
        }
        __syncthreads();
        // ---- stage H: B0 -> B1 ----------------------------------------------
        int in_base = 0, in_pitch = L.p0;
        if (pl.need_h && !(p.ablate & 2)) {
            // work item = (output column, chunk of 4 rows): the column's <= 15 coefficients are
            // fetched once per item and reused for its rows; ~6 items per thread keeps all 256
            // threads busy (one item per column would leave the second pass 3/4 empty).
            const int chunks = (n0 + 3) >> 2;
            const int items = chunks * pl.rw;
            for (int it = tid; it < items; it += CF_NT) {
                const int ck = it / pl.rw, xx = it - ck * pl.rw;
                const int32_t* row = coef_h + (size_t)xx * COEF_ROW;
                const int xmin = row[0], cnt = row[1];
                const int y_end = (ck * 4 + 4) < n0 ? (ck * 4 + 4) : n0;
                if (pl.ksize_h <= 7) {
                    // common case (scale <= 1.5): 7 taps fully unrolled; coefficients beyond cnt are
                    // zero, so the extra byte reads (still inside the LDS allocation) add nothing
                    int k[7];
#pragma unroll
                    for (int t = 0; t < 7; ++t) k[t] = t < cnt ? row[2 + t] : 0;
                    // the 7 taps x 3 channels are 21 consecutive bytes starting at byte 3*xmin of the
                    // (4-byte aligned) LDS row: fetch the 6 aligned dwords that cover them (3x
                    // ds_read2_b32 instead of 21 ds_read_u8 -- this stage is LDS-instruction bound),
                    // funnel-shift them into place, then every byte sits at a compile-time position.
                    const int sb = xmin * 3;
                    const uint32_t shb = (uint32_t)(sb & 3);
                    const int a_dw = sb >> 2;
                    // the chunk's 4 rows are independent: all 24 LDS reads are issued before the first
                    // use (the stage is latency-bound at 3 waves per SIMD otherwise). Rows past the end
                    // of the band are clamped for the reads and skipped for the store.
                    uint32_t w[4][6];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int y = ck * 4 + u < n0 ? ck * 4 + u : n0 - 1;
                        const uint32_t* src = reinterpret_cast<const uint32_t*>(pa_smem + y * L.p0) + a_dw;
#pragma unroll
                        for (int q = 0; q < 6; ++q) w[u][q] = src[q];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        uint32_t r[6];
#pragma unroll
                        for (int q = 0; q < 5; ++q) r[q] = __builtin_amdgcn_alignbyte(w[u][q + 1], w[u][q], shb);
                        r[5] = __builtin_amdgcn_alignbyte(0u, w[u][5], shb);
                        int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
#pragma unroll
                        for (int t = 0; t < 7; ++t) {
                            const int j0 = 3 * t, j1 = 3 * t + 1, j2 = 3 * t + 2;
                            a0 += __mul24((int)((r[j0 >> 2] >> (8 * (j0 & 3))) & 0xff), k[t]);
                            a1 += __mul24((int)((r[j1 >> 2] >> (8 * (j1 & 3))) & 0xff), k[t]);
                            a2 += __mul24((int)((r[j2 >> 2] >> (8 * (j2 & 3))) & 0xff), k[t]);
                        }
                        const int y = ck * 4 + u;
                        if (y < n0) {
                            const int d = L.off1 + y * L.p1 + xx * 3;
                            pa_smem[d + 0] = (uint8_t)clip8(a0);
                            pa_smem[d + 1] = (uint8_t)clip8(a1);
                            pa_smem[d + 2] = (uint8_t)clip8(a2);
                        }
                    }
                } else {
                    for (int y = ck * 4; y < y_end; ++y) {
                        int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
                        const int s = y * L.p0 + xmin * 3;
                        for (int t = 0; t < cnt; ++t) {
                            const int kt = row[2 + t];
                            a0 += __mul24((int)pa_smem[s + 3 * t + 0], kt);
                            a1 += __mul24((int)pa_smem[s + 3 * t + 1], kt);
                            a2 += __mul24((int)pa_smem[s + 3 * t + 2], kt);
                        }
                        const int d = L.off1 + y * L.p1 + xx * 3;
                        pa_smem[d + 0] = (uint8_t)clip8(a0);
                        pa_smem[d + 1] = (uint8_t)clip8(a1);
                        pa_smem[d + 2] = (uint8_t)clip8(a2);
                    }
                }
            }
            in_base = L.off1;
            in_pitch = L.p1;
            __syncthreads();
#ifdef PA_EXTRA_SYNC
            __threadfence_block();
            __syncthreads();
            __builtin_amdgcn_s_sleep(100);
            __syncthreads();
#endif
        }
        // ---- stage V: (B1 | B0) -> B2 ------------------------------------------
        // wave w takes output rows w, w+4, ...: the row's coefficients are wave-uniform (loaded
        // once), each lane produces 4 bytes from one ds_read_b32 per tap row.
        if (pl.need_v && !(p.ablate & 4)) {
            const int row_dwords = (pl.rw * 3 + 3) >> 2;
            const uint32_t* lds32 = reinterpret_cast<const uint32_t*>(pa_smem);
            const int pitch_dw = in_pitch >> 2;
            for (int yy = wave; yy < n2; yy += CF_NW) {
                const int32_t* row = coef_v + (size_t)(b.ry0 + yy) * COEF_ROW;
                const int ymin = row[0], cnt = row[1];
                const int s0 = (in_base >> 2) + (ymin - b.ty0) * pitch_dw;
                uint32_t* dst = reinterpret_cast<uint32_t*>(pa_smem + L.off2 + yy * L.p1);
                if (pl.ksize_v <= 7) {
                    // common case: 7 taps fully unrolled (a 15-way unroll spilled the coefficient
                    // array to scratch); taps beyond cnt have zero coefficients and re-read the
                    // last valid row
                    int k[7];
#pragma unroll
                    for (int t = 0; t < 7; ++t) k[t] = t < cnt ? row[2 + t] : 0;
                    for (int j0 = 0; j0 < row_dwords; j0 += 256) {
                        // four independent dwords per lane: 28 LDS reads in flight
                        int acc[4][4];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc[u][c] = 1 << (PRECISION_BITS - 1);
#pragma unroll
                        for (int t = 0; t < 7; ++t) {
                            const int tr = t < cnt ? t : cnt - 1;
                            uint32_t v[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int j = j0 + lane + 64 * u;
                                v[u] = lds32[s0 + tr * pitch_dw + (j < row_dwords ? j : row_dwords - 1)];
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                acc[u][0] += __mul24((int)(v[u] & 0xff), k[t]);
                                acc[u][1] += __mul24((int)((v[u] >> 8) & 0xff), k[t]);
                                acc[u][2] += __mul24((int)((v[u] >> 16) & 0xff), k[t]);
                                acc[u][3] += __mul24((int)(v[u] >> 24), k[t]);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int j = j0 + lane + 64 * u;
                            // hipcc 7.2 fuses "two (x >> 22) clamped to u8, packed" into v_ashr_pk_u8_i32
                            // and then ORs its result as if bits 16..31 were zero; on gfx950 they are not,
                            // which corrupted bytes 2 and 3 of every dword. The empty asm keeps the four
                            // clamped values opaque so the pack is plain shifts and ORs.
                            uint32_t c0 = clip8(acc[u][0]), c1 = clip8(acc[u][1]), c2 = clip8(acc[u][2]), c3 = clip8(acc[u][3]);
                            asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
                            if (j < row_dwords) dst[j] = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
                        }
                    }
                } else {
                    for (int j = lane; j < row_dwords; j += 64) {
                        int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0, a3 = a0;
                        for (int t = 0; t < cnt; ++t) {
                            const uint32_t v = lds32[s0 + t * pitch_dw + j];
                            const int kt = row[2 + t];
                            a0 += __mul24((int)(v & 0xff), kt);
                            a1 += __mul24((int)((v >> 8) & 0xff), kt);
                            a2 += __mul24((int)((v >> 16) & 0xff), kt);
                            a3 += __mul24((int)(v >> 24), kt);
                        }
                        uint32_t c0 = clip8(a0), c1 = clip8(a1), c2 = clip8(a2), c3 = clip8(a3);
                        asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
                        dst[j] = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
                    }
                }
            }
            in_base = L.off2;
            in_pitch = L.p1;
            __syncthreads();
        }
#ifdef PA_DEBUG_DUMP
        if (p.dbg && crop == p.dbg_crop && r0 == p.dbg_row) {
            int* meta = reinterpret_cast<int*>(p.dbg);
            if (tid == 0) {
                meta[0] = b.dy0; meta[1] = b.dy1; meta[2] = b.ry0; meta[3] = b.ry1; meta[4] = b.ty0; meta[5] = b.ty1;
                meta[6] = L.p0; meta[7] = L.p1; meta[8] = L.off1; meta[9] = L.off2; meta[10] = L.total; meta[11] = rb;
                meta[12] = pl.sw; meta[13] = pl.sh; meta[14] = pl.rw; meta[15] = pl.rh; meta[16] = pl.px; meta[17] = pl.py;
                meta[18] = pl.need_h; meta[19] = pl.need_v; meta[20] = pl.d; meta[21] = pl.area_mode; meta[22] = pl.sx0; meta[23] = pl.sy0;
            }
            for (int i = tid; i < L.total; i += CF_NT) p.dbg[256 + i] = pa_smem[i];
        }
#endif
        // ---- stage A: INTER_AREA + pad + outputs --------------------------------
        if (!(p.ablate & 8)) {
            LdsCanvas cv;
            cv.base = in_base;
            cv.pitch = in_pitch;
            cv.px = pl.px; cv.py = pl.py; cv.rw = pl.rw;
            cv.ry0 = b.ry0; cv.ry1 = b.ry1;
            for (int i = tid; i < rb * PA_CROP; i += CF_NT) {
                const int dy = r0 + (i >> 7), dx = i & 127;
                int o0 = 0, o1 = 0, o2 = 0;
                if (dy < pl.out_h) {
                    if (pl.area_mode == 3) {
                        // fractional INTER_AREA, the common case. Same fp32 operation order as
                        // area_pixel(); taps that fall on the black canvas add +0.0f in the
                        // reference order, so clipping the tap ranges to the pasted region is
                        // exact. One (unaligned) 32-bit LDS read fetches the three channels.
                        // table entries as plain scalars (passing the structs by reference made
                        // hipcc keep them in scratch memory)
                        const AreaTabPacked qx = tabs[dx], qy = tabs[PA_CROP + dy - band0];
                        const int x_first = (int)(qx.bits & 0xffff), x_mid = (int)((qx.bits >> 16) & 0xff);
                        const int x_hf = (int)((qx.bits >> 24) & 1), x_n = x_hf + x_mid + (int)((qx.bits >> 25) & 1);
                        const int y_first = (int)(qy.bits & 0xffff), y_mid = (int)((qy.bits >> 16) & 0xff);
                        const int y_hf = (int)((qy.bits >> 24) & 1), y_n = y_hf + y_mid + (int)((qy.bits >> 25) & 1);
#define PA_ALPHA_X(K) ((K) < x_hf ? qx.a_first : ((K) < x_hf + x_mid ? qx.a_mid : qx.a_last))
#define PA_ALPHA_Y(J) ((J) < y_hf ? qy.a_first : ((J) < y_hf + y_mid ? qy.a_mid : qy.a_last))
                        int k_lo = pl.px - x_first, k_hi = pl.px + pl.rw - x_first;
                        k_lo = k_lo > 0 ? k_lo : 0;
                        k_hi = k_hi < x_n ? k_hi : x_n;
                        int j_lo = pl.py - y_first, j_hi = pl.py + pl.rh - y_first;
                        j_lo = j_lo > 0 ? j_lo : 0;
                        j_hi = j_hi < y_n ? j_hi : y_n;
                        float sum0 = 0.f, sum1 = 0.f, sum2 = 0.f;
                        const int col0 = (x_first + k_lo - pl.px) * 3;
                        const int nk = k_hi - k_lo;
                        if (nk <= 6) {
                            // <= 6 taps x 3 channels = <= 18 bytes per row: six aligned dwords cover them
                            // at any alignment; funnel-shifted, every byte sits at a static position
                            float al[6];
#pragma unroll
                            for (int k = 0; k < 6; ++k) al[k] = k < nk ? PA_ALPHA_X(k_lo + k) : 0.f;
                            for (int j = j_lo; j < j_hi; ++j) {
                                const float beta = PA_ALPHA_Y(j);
                                const int o = in_base + (y_first + j - pl.py - b.ry0) * in_pitch + col0;
                                const uint32_t* src = reinterpret_cast<const uint32_t*>(pa_smem + (o & ~3));
                                const uint32_t shb = (uint32_t)(o & 3);
                                uint32_t w[6], r[5];
#pragma unroll
                                for (int q = 0; q < 6; ++q) w[q] = src[q];
#pragma unroll
                                for (int q = 0; q < 5; ++q) r[q] = __builtin_amdgcn_alignbyte(w[q + 1], w[q], shb);
                                float b0 = 0.f, b1 = 0.f, b2 = 0.f;
#pragma unroll
                                for (int k = 0; k < 6; ++k) {
                                    if (k < nk) {
                                        const int q0 = 3 * k, q1 = 3 * k + 1, q2 = 3 * k + 2;
                                        b0 = b0 + (float)((r[q0 >> 2] >> (8 * (q0 & 3))) & 0xff) * al[k];
                                        b1 = b1 + (float)((r[q1 >> 2] >> (8 * (q1 & 3))) & 0xff) * al[k];
                                        b2 = b2 + (float)((r[q2 >> 2] >> (8 * (q2 & 3))) & 0xff) * al[k];
                                    }
                                }
                                sum0 = sum0 + beta * b0;
                                sum1 = sum1 + beta * b1;
                                sum2 = sum2 + beta * b2;
                            }
                        } else {
                            for (int j = j_lo; j < j_hi; ++j) {
                                const float beta = PA_ALPHA_Y(j);
                                int o = in_base + (y_first + j - pl.py - b.ry0) * in_pitch + col0;
                                float b0 = 0.f, b1 = 0.f, b2 = 0.f;
                                for (int k = k_lo; k < k_hi; ++k) {
                                    const float alpha = PA_ALPHA_X(k);
                                    b0 = b0 + (float)pa_smem[o] * alpha;
                                    b1 = b1 + (float)pa_smem[o + 1] * alpha;
                                    b2 = b2 + (float)pa_smem[o + 2] * alpha;
                                    o += 3;
                                }
                                sum0 = sum0 + beta * b0;
                                sum1 = sum1 + beta * b1;
                                sum2 = sum2 + beta * b2;
                            }
                        }
#undef PA_ALPHA_X
#undef PA_ALPHA_Y
                        o0 = cv_saturate_u8(sum0);
                        o1 = cv_saturate_u8(sum1);
                        o2 = cv_saturate_u8(sum2);
                    } else {
                        area_pixel(pl, cv, dy, dx, o0, o1, o2);
                    }
                }
                write_crop_pixel(p, crop, dy * PA_CROP + dx, o0, o1, o2);
            }
        }
        __syncthreads();  // the next sub-band reuses the LDS stages
#ifdef PA_STAMP_BUILD
        if (p.dbg && threadIdx.x == 0 && r0 + rb >= band0 + 8) {
            unsigned long long* o = reinterpret_cast<unsigned long long*>(p.dbg) + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4;
            o[0] = st0; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = (unsigned long long)pl.d; o[3] = rb;
        }
#endif
    }
}

// ---------------------------------------------------------------------------
// Log-projection boxes: the reference's current source of
// fighter boxes. Fighter.set_from_json (playaid/fighter.py:494-539) projects four corners
// around the logged world position through a look-at camera (calculate_lookat_matrix :87-121,
// calculate_intrinsic_matrix :66-84, project_point_to_pixel :124-155, all for a hard-coded
// 1280x720 image) and YoloCrop.from_pixel_coordinates (:170-190) turns the rounded pixels
// into a normalised box. One thread per (frame, fighter), fp64 like numpy.
// log row: pos_x, pos_y, cam_x, cam_y, cam_z, tgt_x, tgt_y, tgt_z, fov_degrees.
__global__ void project_boxes_kernel(const double* __restrict__ log, double* __restrict__ boxes, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* r = log + (size_t)i * 9;
    const double W = 1280.0, H = 720.0;
    // look-at basis
    double fx = r[2] - r[5], fy = r[3] - r[6], fz = r[4] - r[7];
    const double fn = sqrt(fx * fx + fy * fy + fz * fz);
    fx /= fn; fy /= fn; fz /= fn;
    double rx = 1.0 * fz - 0.0 * fy, ry = 0.0 * fx - 0.0 * fz, rz = 0.0 * fy - 1.0 * fx;  // cross(up, forward)
    const double rn = sqrt(rx * rx + ry * ry + rz * rz);
    rx /= rn; ry /= rn; rz /= rn;
    const double ux = fy * rz - fz * ry, uy = fz * rx - fx * rz, uz = fx * ry - fy * rx;  // cross(forward, right)
    const double fov_rad = r[8] * (3.141592653589793 / 180.0);
    const double focal = W / (2.0 * tan(fov_rad / 2.0));
    const double dxs[4] = {-10.0, 10.0, -10.0, 10.0};
    const double dys[4] = {20.0, 20.0, -3.0, -3.0};
    double px[4], py[4];
    for (int c = 0; c < 4; ++c) {
        // inverse of [R | t] with orthonormal rows R = (right, up, -forward): R^T (p - t)
        const double dx = r[0] + dxs[c] - r[2], dy = r[1] + dys[c] - r[3], dz = 0.0 - r[4];
        const double cxm = rx * dx + ux * dy - fx * dz;
        const double cym = ry * dx + uy * dy - fy * dz;
        const double czm = rz * dx + uz * dy - fz * dz;
        const double nx = cxm / czm, ny = cym / czm;
        const double ix = focal * nx + W / 2.0;
        const double iy = H - (focal * ny + H / 2.0);
        px[c] = rint(ix);  // np.round: half to even
        py[c] = rint(iy);
    }
    const double cx = (px[0] + px[1] + px[2] + px[3]) / 4.0, cy = (py[0] + py[1] + py[2] + py[3]) / 4.0;
    const double bw = fmax(fmax(px[0], px[1]), fmax(px[2], px[3])) - fmin(fmin(px[0], px[1]), fmin(px[2], px[3]));
    const double bh = fmax(fmax(py[0], py[1]), fmax(py[2], py[3])) - fmin(fmin(py[0], py[1]), fmin(py[2], py[3]));
    double* o = boxes + (size_t)i * 4;
    o[0] = cx / W;
    o[1] = cy / H;
    o[2] = bw / W;
    o[3] = bh / H;
}

hipError_t launch_project_boxes(const double* log, double* boxes, int32_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(project_boxes_kernel, dim3((n + 127) / 128), dim3(128), 0, s, log, boxes, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// The runner's own input branch (playaid/ai_runner.py:446-459): a crop IMAGE of any size -- what
// YOLOv5 --save-crop writes and cv2.imread returns, uint8 [h][w][3] BGR -- becomes the 128 x 128 model
// input by  BGR2RGB -> imutils.resize(width=128) = cv2.resize(INTER_AREA) to (128, int(h * (128 / w)))
// -> ImageOps.pad((128, 128), black) when that is not 128 rows (Pillow: aspect-preserving BICUBIC
// "contain", pasted centred). The order of the two resamplers is the reverse of square_crop's, and the
// INTER_AREA source is not square, so this is its own kernel: one workgroup per image, the three
// passes over L2-resident scratch (t1: INTER_AREA result, t2: after the horizontal bicubic pass, then
// t1's second half: after the vertical pass), coefficient tables in LDS. Same integer / IEEE arithmetic
// as the fused path (shared helpers above), bit-exact against oracle/yolo_crop.runner_input_from_crop.
struct RunnerInPlan {
    int status;
    int sw, sh;          // source image
    int ow, oh;          // INTER_AREA destination width (128 for runner inputs) and height
    int mode;            // 0 copy, 1 2x2, 2 integer, 3 general, 4 enlarging (bilinear emulation)
    int isx, isy;
    double scale_x, scale_y;
    int rw, rh, px, py;  // ImageOps.pad: size after contain, paste offset (rw == 128 && rh == 128: no pad step)
    int need_h, need_v;
};

struct WhCanvas {  // plain [rows][cols][3] bytes
    const uint8_t* src;
    int pitch;
    __device__ __forceinline__ void load(int y, int x, int& c0, int& c1, int& c2) const {
        const uint8_t* s = src + (size_t)y * pitch + x * 3;
        c0 = s[0]; c1 = s[1]; c2 = s[2];
    }
};

// One destination pixel of cv::resize(INTER_AREA) from an sw x sh image to ow x oh (area_pixel() with
// separate axes; the source is a plain image, no paste window).
__device__ __forceinline__ void area_pixel_wh(const RunnerInPlan& pl, const WhCanvas& cv, int dy, int dx, int& o0, int& o1, int& o2) {
    if (pl.mode == 0) {
        cv.load(dy, dx, o0, o1, o2);
    } else if (pl.mode == 1) {
        int s0 = 2, s1 = 2, s2 = 2;
        for (int yy = 0; yy < 2; ++yy)
            for (int xx = 0; xx < 2; ++xx) {
                int c0, c1, c2;
                cv.load(dy * 2 + yy, dx * 2 + xx, c0, c1, c2);
                s0 += c0; s1 += c1; s2 += c2;
            }
        o0 = s0 >> 2; o1 = s1 >> 2; o2 = s2 >> 2;
    } else if (pl.mode == 2) {
        int s0 = 0, s1 = 0, s2 = 0;
        for (int yy = 0; yy < pl.isy; ++yy)
            for (int xx = 0; xx < pl.isx; ++xx) {
                int c0, c1, c2;
                cv.load(dy * pl.isy + yy, dx * pl.isx + xx, c0, c1, c2);
                s0 += c0; s1 += c1; s2 += c2;
            }
        const float scale = 1.f / (float)(pl.isx * pl.isy);
        o0 = cv_saturate_u8((float)s0 * scale);
        o1 = cv_saturate_u8((float)s1 * scale);
        o2 = cv_saturate_u8((float)s2 * scale);
    } else if (pl.mode == 4) {
        // cv::resize's 8-bit bilinear resizer with area-mode coefficients (see area_pixel())
        const double inv_x = (double)pl.ow / (double)pl.sw, inv_y = (double)pl.oh / (double)pl.sh;
        int sx = (int)floor((double)dx * pl.scale_x);
        float fx = (float)((double)(dx + 1) - (double)(sx + 1) * inv_x);
        fx = fx <= 0.f ? 0.f : fx - floorf(fx);
        const bool plain = sx + 1 >= pl.sw;
        if (sx >= pl.sw - 1) {
            fx = 0.f;
            sx = pl.sw - 1;
        }
        const int a0 = (int)rintf((1.f - fx) * 2048.f), a1 = (int)rintf(fx * 2048.f);
        const int sy = (int)floor((double)dy * pl.scale_y);
        float fy = (float)((double)(dy + 1) - (double)(sy + 1) * inv_y);
        fy = fy <= 0.f ? 0.f : fy - floorf(fy);
        const int b0 = (int)rintf((1.f - fy) * 2048.f), b1 = (int)rintf(fy * 2048.f);
        const int r0 = sy < pl.sh - 1 ? sy : pl.sh - 1;
        const int r1 = sy + 1 < pl.sh - 1 ? sy + 1 : pl.sh - 1;
        int h0[3], h1[3];
        int c0, c1, c2, e0 = 0, e1 = 0, e2 = 0;
        cv.load(r0, sx, c0, c1, c2);
        if (!plain) cv.load(r0, sx + 1, e0, e1, e2);
        h0[0] = plain ? c0 * 2048 : c0 * a0 + e0 * a1;
        h0[1] = plain ? c1 * 2048 : c1 * a0 + e1 * a1;
        h0[2] = plain ? c2 * 2048 : c2 * a0 + e2 * a1;
        cv.load(r1, sx, c0, c1, c2);
        if (!plain) cv.load(r1, sx + 1, e0, e1, e2);
        h1[0] = plain ? c0 * 2048 : c0 * a0 + e0 * a1;
        h1[1] = plain ? c1 * 2048 : c1 * a0 + e1 * a1;
        h1[2] = plain ? c2 * 2048 : c2 * a0 + e2 * a1;
        o0 = ((((b0 * (h0[0] >> 4)) >> 16) + ((b1 * (h1[0] >> 4)) >> 16) + 2) >> 2) & 0xff;
        o1 = ((((b0 * (h0[1] >> 4)) >> 16) + ((b1 * (h1[1] >> 4)) >> 16) + 2) >> 2) & 0xff;
        o2 = ((((b0 * (h0[2] >> 4)) >> 16) + ((b1 * (h1[2] >> 4)) >> 16) + 2) >> 2) & 0xff;
    } else {
        const AreaTab tx = area_tab(dx, pl.scale_x, pl.sw);
        const AreaTab ty = area_tab(dy, pl.scale_y, pl.sh);
        float sum0 = 0.f, sum1 = 0.f, sum2 = 0.f;
        for (int j = 0; j < ty.n; ++j) {
            const float beta = area_alpha(ty, j);
            float b0 = 0.f, b1 = 0.f, b2 = 0.f;
            for (int k = 0; k < tx.n; ++k) {
                const float alpha = area_alpha(tx, k);
                int c0, c1, c2;
                cv.load(ty.s_first + j, tx.s_first + k, c0, c1, c2);
                b0 = b0 + (float)c0 * alpha;
                b1 = b1 + (float)c1 * alpha;
                b2 = b2 + (float)c2 * alpha;
            }
            sum0 = sum0 + beta * b0;
            sum1 = sum1 + beta * b1;
            sum2 = sum2 + beta * b2;
        }
        o0 = cv_saturate_u8(sum0);
        o1 = cv_saturate_u8(sum1);
        o2 = cv_saturate_u8(sum2);
    }
}

// What get_action_recognition_input_for_frame does to one crop image (ai_runner.py:446-459), decided once per image: the
// INTER_AREA destination, the branch of cv::resize it takes, ImageOps.pad's contain size / paste offset / bicubic passes.
__device__ void runner_plan(const RunnerInParams& q, int crop, RunnerInPlan& pl, long long& off) {
    pl.status = PA_CROP_OK;
    off = q.desc[crop].offset;
    pl.sh = q.desc[crop].height;
    pl.sw = q.desc[crop].width;
    pl.ow = PA_CROP; pl.oh = 0; pl.mode = 0; pl.isx = pl.isy = 1; pl.scale_x = pl.scale_y = 1.0;
    pl.rw = pl.rh = PA_CROP; pl.px = pl.py = 0; pl.need_h = pl.need_v = 0;
    if (pl.sh < 1 || pl.sw < 1 || pl.sh > q.max_h || pl.sw > q.max_w || off < 0 ||
        off + (long long)pl.sh * pl.sw * 3 > q.images_bytes)
        pl.status = PA_CROP_BAD_BOX;
    if (pl.status == PA_CROP_OK) {
        // imutils.resize(width=128): dim = (128, int(h * (128 / float(w))))
        const double r = 128.0 / (double)pl.sw;
        const double ohd = (double)pl.sh * r;
        pl.oh = ohd < 2.0e9 ? (int)ohd : 0;
        if (pl.oh < 1) pl.status = PA_CROP_EMPTY;  // cv2.resize raises for an empty destination
    }
    if (pl.status == PA_CROP_OK) {
        const double inv_sx = 128.0 / (double)pl.sw, inv_sy = (double)pl.oh / (double)pl.sh;
        pl.scale_x = 1.0 / inv_sx;
        pl.scale_y = 1.0 / inv_sy;
        if (pl.sw == PA_CROP && pl.sh == pl.oh) {
            pl.mode = 0;
        } else if (pl.scale_x < 1.0 || pl.scale_y < 1.0) {
            pl.mode = 4;
        } else {
            pl.isx = (int)rint(pl.scale_x);
            pl.isy = (int)rint(pl.scale_y);
            const bool fast = fabs(pl.scale_x - pl.isx) < 2.220446049250313e-16 && fabs(pl.scale_y - pl.isy) < 2.220446049250313e-16;
            pl.mode = fast ? ((pl.isx == 2 && pl.isy == 2) ? 1 : 2) : 3;
        }
        if (pl.oh != PA_CROP) {
            // ImageOps.pad(img 128 x oh, (128, 128)): contain keeps the aspect ratio (ImageOps.py)
            const double im_ratio = 128.0 / (double)pl.oh;
            if (im_ratio > 1.0) {
                const int nh = (int)rint((double)pl.oh / 128.0 * 128.0);
                if (nh != PA_CROP) pl.rh = nh;
            } else {
                const int nw = (int)rint(128.0 / (double)pl.oh * 128.0);
                if (nw != PA_CROP) pl.rw = nw;
            }
            if (pl.rw < 1 || pl.rh < 1) pl.status = PA_CROP_EMPTY;
            pl.need_h = pl.rw != PA_CROP;
            pl.need_v = pl.rh != pl.oh;
            if (pl.rw != PA_CROP)
                pl.px = (int)rint((double)(PA_CROP - pl.rw) * 0.5);
            else if (pl.rh != PA_CROP)
                pl.py = (int)rint((double)(PA_CROP - pl.rh) * 0.5);
            if ((pl.need_h && bicubic_ksize(PA_CROP, pl.rw) > PA_KSIZE_MAX) || (pl.need_v && bicubic_ksize(pl.oh, pl.rh) > PA_KSIZE_MAX))
                pl.status = PA_CROP_FILTER_TOO_WIDE;
        }
        if ((size_t)(pl.oh + (pl.need_v ? pl.rh : 0)) * PA_CROP * 3 > q.t_stride) pl.status = PA_CROP_FILTER_TOO_WIDE;  // scratch
    }
}

// The three steps of a runner input as three launches over (image, part) grids -- INTER_AREA | Pillow's horizontal pass | its
// vertical pass fused with the paste -- instead of one workgroup per image walking all of them between barriers (round 3:
// 128 workgroups on a 256-CU part, 0.39 ms per 128 crop images; the kernel boundaries are the barriers now, and every
// launch has 4 x as many workgroups). Every workgroup works the image's plan out again (a few dozen scalar operations).
// Same arithmetic, same order per output value: bit-identical results.
constexpr int RI_PARTS = 4;
template <int STAGE>
__global__ __launch_bounds__(256) void runner_input_stage_kernel(const RunnerInParams q) {
    __shared__ int32_t coef[PA_CROP][COEF_ROW];  // the pad step's table of this stage's axis: <= 128 output coordinates
    const int crop = blockIdx.x, part = blockIdx.y;
    const int tid = threadIdx.x;
    const int i0 = part * 256 + tid, istep = RI_PARTS * 256;
    PreprocParams out;  // write_crop_pixel() only reads these fields
    out.swap_rb = q.swap_rb;
    out.crops_u8 = q.inputs_u8;
    out.crops_f32 = q.inputs_f32;
    out.crops_f32_is_bf16 = q.inputs_f32_is_bf16;
    RunnerInPlan pl;
    long long off;
    runner_plan(q, crop, pl, off);
    if (STAGE == 0 && part == 0 && tid == 0 && q.status) q.status[crop] = pl.status;
    if (pl.status != PA_CROP_OK) {
        if (STAGE == 2)
            for (int i = i0; i < PA_CROP * PA_CROP; i += istep) write_crop_pixel(out, crop, i, 0, 0, 0);
        return;
    }
    uint8_t* A = q.t1 + (size_t)crop * q.t_stride;   // [oh][128][3] behind INTER_AREA
    uint8_t* B = q.t2 + (size_t)crop * q.t_stride;   // [oh][rw][3] behind the horizontal pass
    if (STAGE == 0) {
        WhCanvas cv;
        cv.src = q.images + off;
        cv.pitch = pl.sw * 3;
        for (int i = i0; i < pl.oh * PA_CROP; i += istep) {
            const int dy = i >> 7, dx = i & 127;
            int o0, o1, o2;
            area_pixel_wh(pl, cv, dy, dx, o0, o1, o2);
            uint8_t* o = A + (size_t)i * 3;
            o[0] = (uint8_t)o0; o[1] = (uint8_t)o1; o[2] = (uint8_t)o2;
        }
        return;
    }
    if (STAGE == 1) {  // ImagingResampleHorizontal_8bpc: [oh][128] -> B as [oh][rw]
        if (!pl.need_h) return;
        if (tid < pl.rw) bicubic_coef_row(PA_CROP, pl.rw, tid, coef[tid]);  // (fp64, one output coordinate per thread)
        __syncthreads();
        for (int i = i0; i < pl.oh * pl.rw; i += istep) {
            const int y = i / pl.rw, xx = i - y * pl.rw;
            const int32_t* row = coef[xx];
            const uint8_t* sp = A + ((size_t)y * PA_CROP + row[0]) * 3;
            int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
            for (int t = 0; t < row[1]; ++t) {
                const int k = row[2 + t];
                a0 += __mul24((int)sp[3 * t + 0], k);
                a1 += __mul24((int)sp[3 * t + 1], k);
                a2 += __mul24((int)sp[3 * t + 2], k);
            }
            uint8_t* o = B + (size_t)i * 3;
            o[0] = (uint8_t)clip8(a0); o[1] = (uint8_t)clip8(a1); o[2] = (uint8_t)clip8(a2);
        }
        return;
    }
    // STAGE 2: ImagingResampleVertical_8bpc for the canvas pixels that need it, paste on the black 128 x 128 canvas, channel
    // swap, u8 + model input
    const uint8_t* cur = pl.need_h ? B : A;
    const int cur_w = pl.need_h ? pl.rw : PA_CROP;
    const int cur_h = pl.need_v ? pl.rh : pl.oh;
    if (pl.need_v) {
        if (tid < pl.rh) bicubic_coef_row(pl.oh, pl.rh, tid, coef[tid]);
        __syncthreads();
    }
    for (int i = i0; i < PA_CROP * PA_CROP; i += istep) {
        const int y = (i >> 7) - pl.py, x = (i & 127) - pl.px;
        int o0 = 0, o1 = 0, o2 = 0;
        if ((unsigned)y < (unsigned)cur_h && (unsigned)x < (unsigned)cur_w) {
            if (pl.need_v) {
                const int32_t* row = coef[y];
                const uint8_t* sp = cur + ((size_t)row[0] * cur_w + x) * 3;
                int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
                for (int t = 0; t < row[1]; ++t) {
                    const int k = row[2 + t];
                    a0 += __mul24((int)sp[0], k);
                    a1 += __mul24((int)sp[1], k);
                    a2 += __mul24((int)sp[2], k);
                    sp += (size_t)cur_w * 3;
                }
                o0 = clip8(a0); o1 = clip8(a1); o2 = clip8(a2);
            } else {
                const uint8_t* sp = cur + ((size_t)y * cur_w + x) * 3;
                o0 = sp[0]; o1 = sp[1]; o2 = sp[2];
            }
        }
        write_crop_pixel(out, crop, i, o0, o1, o2);
    }
}

hipError_t launch_runner_inputs(const RunnerInParams& q, hipStream_t s) {
    if (q.n <= 0) return hipSuccess;
    const dim3 grid(q.n, RI_PARTS);
    hipLaunchKernelGGL(runner_input_stage_kernel<0>, grid, dim3(256), 0, s, q);
    hipLaunchKernelGGL(runner_input_stage_kernel<1>, grid, dim3(256), 0, s, q);
    hipLaunchKernelGGL(runner_input_stage_kernel<2>, grid, dim3(256), 0, s, q);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Window ingest: the frames stay in (pinned, device-visible) HOST memory and only every crop's source slice
// crosses PCIe. One wave copies one slice row at a time straight out of host memory -- coalesced dword reads
// from the 4-byte-aligned address below the row start, re-aligned with v_alignbyte like stage 0 of the fused
// kernel -- into the packed window buffer (row pitch padded to 16 bytes). A 2-D hipMemcpy per crop was measured
// at ~2 ms each on this stack; this kernel is one launch for all crops and is bound by the link.
__global__ __launch_bounds__(256) void slice_upload_kernel(const uint8_t* __restrict__ frames_host, const CropWindow* __restrict__ desc,
                                                           uint8_t* __restrict__ windows, long long frames_bytes) {
    const int crop = blockIdx.y;
    const CropWindow w = desc[crop];
    const int rows = w.rows;
    const int row_bytes = w.row_bytes;
    if (rows <= 0 || row_bytes <= 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int row_dwords = (row_bytes + 3) >> 2;
    for (int y = wave; y < rows; y += nwaves) {
        const long long so = w.src_offset + (long long)y * w.src_pitch;
        const uintptr_t ga = (uintptr_t)(frames_host + so);
        const uint32_t* g4 = reinterpret_cast<const uint32_t*>(ga & ~(uintptr_t)3);
        const uint32_t sh = (uint32_t)(ga & 3);
        // bytes of the frame buffer left from the aligned address on: never read past its end
        const long long left = frames_bytes - (so - (long long)sh);
        uint32_t* dst = reinterpret_cast<uint32_t*>(windows + w.offset + (size_t)y * w.pitch);
        // the link's latency is microseconds: every lane requests up to six dwords (a 1.5 KB row per pass) before the
        // first one is used
        for (int j0 = 0; j0 < row_dwords; j0 += 384) {
            uint32_t lo[6], hi[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int j = j0 + lane + 64 * u;
                lo[u] = j < row_dwords ? g4[j] : 0u;
                hi[u] = (sh && j < row_dwords && 4ll * (j + 1) < (long long)sh + row_bytes && 4ll * (j + 2) <= left) ? g4[j + 1] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int j = j0 + lane + 64 * u;
                if (j < row_dwords) dst[j] = __builtin_amdgcn_alignbyte(hi[u], lo[u], sh);
            }
        }
    }
}

hipError_t launch_slice_upload(const uint8_t* frames_host, long long frames_bytes, const CropWindow* desc, uint8_t* windows, int ncrops,
                               hipStream_t s) {
    if (ncrops <= 0) return hipSuccess;
    hipLaunchKernelGGL(slice_upload_kernel, dim3(32, ncrops), dim3(256), 0, s, frames_host, desc, windows, frames_bytes);
    return hipGetLastError();
}

// YoloCrop.crop_img (fighter.py:316-321) + imutils.resize(width = out_w) (= cv2.resize INTER_AREA to
// (out_w, int(h * (out_w / float(w)))), imutils 0.5.4) for up to four fixed pixel rectangles per frame: the damage
// HUD crops of AIRunner.run_damage_detection (ai_runner.py:556-571, damage_crop_to_percent :114). One workgroup per
// (rectangle, frame), one thread per destination pixel, every INTER_AREA branch of area_pixel_wh (the HUD boxes are
// ENLARGED at 720p / 1080p: OpenCV's bilinear emulation). Channel order is kept (BGR in, BGR out).
__global__ __launch_bounds__(256) void rect_resize_kernel(const RectResizeParams q) {
    const int r = blockIdx.x, f = blockIdx.y;
    RunnerInPlan pl;
    pl.status = PA_CROP_OK;
    pl.sw = q.x2[r] - q.x1[r];
    pl.sh = q.y2[r] - q.y1[r];
    pl.ow = q.out_w;
    pl.oh = q.out_h[r];
    pl.rw = pl.rh = pl.px = pl.py = pl.need_h = pl.need_v = 0;
    pl.isx = pl.isy = 1;
    const double inv_sx = (double)pl.ow / (double)pl.sw, inv_sy = (double)pl.oh / (double)pl.sh;
    pl.scale_x = 1.0 / inv_sx;
    pl.scale_y = 1.0 / inv_sy;
    if (pl.sw == pl.ow && pl.sh == pl.oh) {
        pl.mode = 0;
    } else if (pl.scale_x < 1.0 || pl.scale_y < 1.0) {
        pl.mode = 4;
    } else {
        pl.isx = (int)rint(pl.scale_x);
        pl.isy = (int)rint(pl.scale_y);
        const bool fast = fabs(pl.scale_x - pl.isx) < 2.220446049250313e-16 && fabs(pl.scale_y - pl.isy) < 2.220446049250313e-16;
        pl.mode = fast ? ((pl.isx == 2 && pl.isy == 2) ? 1 : 2) : 3;
    }
    WhCanvas cv;
    cv.pitch = q.width * 3;
    cv.src = q.frames + ((size_t)f * q.height + q.y1[r]) * q.width * 3 + (size_t)q.x1[r] * 3;
    uint8_t* dst = q.out + ((size_t)f * q.n_rects + r) * q.out_h_cap * q.out_w * 3;
    for (int i = threadIdx.x; i < pl.oh * pl.ow; i += 256) {
        const int dy = i / pl.ow, dx = i - dy * pl.ow;
        int o0, o1, o2;
        area_pixel_wh(pl, cv, dy, dx, o0, o1, o2);
        uint8_t* o = dst + (size_t)i * 3;
        o[0] = (uint8_t)o0; o[1] = (uint8_t)o1; o[2] = (uint8_t)o2;
    }
}

hipError_t launch_rect_resize(const RectResizeParams& q, int n_frames, hipStream_t s) {
    if (n_frames <= 0 || q.n_rects <= 0) return hipSuccess;
    hipLaunchKernelGGL(rect_resize_kernel, dim3(q.n_rects, n_frames), dim3(256), 0, s, q);
    return hipGetLastError();
}

hipError_t preprocess_init_device() {
    // dynamic LDS beyond the 64 KiB default; the attribute is per device, so pa_create sets it for its own
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&crop_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t launch_preprocess(const PreprocParams& p_in, hipStream_t s) {
    PreprocParams p = p_in;
    static const int budget = getenv("PA_FUSED_LDS") ? atoi(getenv("PA_FUSED_LDS")) : PA_FUSED_LDS_BYTES;
    p.fused_lds = budget & ~15;
    static const int ablate = getenv("PA_PRE_ABLATE") ? atoi(getenv("PA_PRE_ABLATE")) : 0;  // timing experiments only
    p.ablate = ablate;
    const int ncrops = p.n_frames * p.fighters;
    if (ncrops <= 0) return hipSuccess;
    hipLaunchKernelGGL(crop_plan_kernel, dim3(ncrops), dim3(256), 0, s, p);
#ifdef PA_STAMP_BUILD
    static int calls = 0;
    static unsigned long long* sd = nullptr;
    const char* sf = getenv("PA_CROP_STAMP_FILE");
    const bool now = sf && calls++ == 5;
    if (now) {
        if (!sd) (void)hipMalloc(&sd, (size_t)16 * ncrops * 32);
        (void)hipMemsetAsync(sd, 0, (size_t)16 * ncrops * 32, s);
        p.dbg = reinterpret_cast<uint8_t*>(sd);
    }
#endif
    hipLaunchKernelGGL(crop_fused_kernel, dim3(PA_CROP / 8, ncrops + 1), dim3(CF_NT), p.fused_lds + (PA_CROP + 8) * sizeof(AreaTabPacked), s, p);
    { hipError_t e1 = hipGetLastError(); if (e1 != hipSuccess) return e1; }
#ifdef PA_STAMP_BUILD
    if (now) {
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> h((size_t)16 * ncrops * 4);
        (void)hipMemcpy(h.data(), sd, h.size() * 8, hipMemcpyDeviceToHost);
        FILE* f = fopen(sf, "wb");
        if (f) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
        p.dbg = nullptr;
    }
#endif
    return hipGetLastError();
}

size_t coef_cache_ints(int dmax) { return (size_t)dmax * (dmax + 1) / 2 * COEF_ROW; }

hipError_t launch_build_coef_cache(int32_t* cache, int padding, int dmax, hipStream_t s) {
    if (dmax <= 0) return hipSuccess;
    hipLaunchKernelGGL(coef_cache_kernel, dim3((dmax + 255) / 256, dmax), dim3(256), 0, s, cache, padding, dmax);
    return hipGetLastError();
}

}  // namespace pa
