// Detection post-processing on gfx950: what the reference gets back from its YOLOv5 subprocess
// (playaid/ai_runner.py:191-224: detect.py --max-det 2 --save-txt --save-conf --classes 2 3) computed from the
// detection head's decoded output rows. The arithmetic is ultralytics/yolov5's (un-vendored, unpinned checkout):
// utils/general.py::non_max_suppression -- objectness gate, conf = obj * cls, best class per row, class filter,
// class-offset torchvision.ops.nms, max_det -- then scale_boxes + clip + round and xyxy2xywh / gn; the contract
// is written down in oracle/detect.py and this kernel matches it bit for bit (file built with
// -ffp-contract=off: every multiply and add rounds separately, as the fp32 tensor ops do).
//
// One workgroup per frame. With max_det in the single digits the greedy NMS is max_det arg-max passes over the
// rows (a row is out when its IoU with an already kept box of its class exceeds the threshold), each a strided
// scan + wave64 shuffle reduction of a 64-bit (score, -row) key: HBM/L2 streaming, no sort.
#include "pa_kernels.h"

namespace pa {

namespace {

constexpr int DET_MAX = 8;
constexpr float MAX_WH = 7680.f;  // class offset of the batched NMS

struct Cand {
    float score;
    int cls;
    float b0, b1, b2, b3;  // class-offset box
    float x1, y1, x2, y2;  // plain box
    float area;
};

// candidate of one head row, or score <= 0 when the row is filtered out
__device__ __forceinline__ bool make_cand(const float* __restrict__ p, int nc, float conf_thres, uint32_t mask, Cand& c) {
    const float obj = p[4];
    if (!(obj > conf_thres)) return false;
    float best = p[5] * obj;
    int j = 0;
    for (int k = 1; k < nc; ++k) {
        const float v = p[5 + k] * obj;
        if (v > best) {  // first maximum
            best = v;
            j = k;
        }
    }
    if (!(best > conf_thres) || !((mask >> j) & 1u)) return false;
    const float hw = p[2] / 2.f, hh = p[3] / 2.f;
    c.score = best;
    c.cls = j;
    c.x1 = p[0] - hw; c.y1 = p[1] - hh; c.x2 = p[0] + hw; c.y2 = p[1] + hh;
    const float off = (float)j * MAX_WH;
    c.b0 = c.x1 + off; c.b1 = c.y1 + off; c.b2 = c.x2 + off; c.b3 = c.y2 + off;
    c.area = (c.b2 - c.b0) * (c.b3 - c.b1);
    return true;
}

}  // namespace

// One workgroup of NMS_THREADS per frame: the scan over a frame's rows (15120 at 384 x 640, 44 bytes each) is latency-bound, so the
// workgroup is as wide as one can be -- 1024 threads keep four times the loads of 256 in flight (49 -> ~25 us per 64 frames).
constexpr int NMS_THREADS = 1024;
__global__ __launch_bounds__(NMS_THREADS) void detect_nms_kernel(const DetectParams q) {
    __shared__ unsigned long long wave_best[NMS_THREADS / 64];
    __shared__ Cand kept[DET_MAX];
    __shared__ int kept_row[DET_MAX];
    __shared__ int n_kept;
    const int frame = blockIdx.x;
    const int tid = threadIdx.x;
    const int stride = 5 + q.nc;
    const float* pred = q.pred + (size_t)frame * q.rows * stride;
    if (tid == 0) n_kept = 0;
    __syncthreads();
    for (int k = 0; k < q.max_det; ++k) {
        const int nk = n_kept;
        unsigned long long best = 0ull;  // (score bits << 32) | ~row: larger score first, then the lower row
        for (int r = tid; r < q.rows; r += NMS_THREADS) {
            Cand c;
            if (!make_cand(pred + (size_t)r * stride, q.nc, q.conf_thres, q.class_mask, c)) continue;
            bool out = false;
            for (int i = 0; i < nk; ++i) {
                if (kept_row[i] == r) {
                    out = true;
                    break;
                }
                const Cand& a = kept[i];
                const float xx1 = fmaxf(a.b0, c.b0), yy1 = fmaxf(a.b1, c.b1);
                const float xx2 = fminf(a.b2, c.b2), yy2 = fminf(a.b3, c.b3);
                const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
                const float inter = w * h;
                const float ovr = inter / ((a.area + c.area) - inter);
                if (ovr > q.iou_thres) {
                    out = true;
                    break;
                }
            }
            if (out) continue;
            const unsigned long long key = ((unsigned long long)__float_as_uint(c.score) << 32) | (unsigned long long)(0xffffffffu - (unsigned)r);
            best = key > best ? key : best;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned lo = __shfl_xor((unsigned)(best & 0xffffffffu), d, 64), hi = __shfl_xor((unsigned)(best >> 32), d, 64);
            const unsigned long long o = ((unsigned long long)hi << 32) | lo;
            best = o > best ? o : best;
        }
        if ((tid & 63) == 0) wave_best[tid >> 6] = best;
        __syncthreads();
        if (tid == 0) {
            unsigned long long b = wave_best[0];
            for (int w = 1; w < NMS_THREADS / 64; ++w) b = wave_best[w] > b ? wave_best[w] : b;
            if (b != 0ull) {
                const int r = (int)(0xffffffffu - (unsigned)(b & 0xffffffffu));
                Cand c;
                (void)make_cand(pred + (size_t)r * stride, q.nc, q.conf_thres, q.class_mask, c);
                kept[nk] = c;
                kept_row[nk] = r;
                n_kept = nk + 1;
            }
        }
        __syncthreads();
        if (n_kept == nk) break;  // nothing left
    }
    // scale_boxes + clip + round, xyxy2xywh / gn; label-file order = reversed(det): lowest confidence first
    const int n = n_kept;
    if (tid < n) {
        const Cand& c = kept[tid];
        float x1 = (c.x1 - q.pad_x) / q.gain, x2 = (c.x2 - q.pad_x) / q.gain;
        float y1 = (c.y1 - q.pad_y) / q.gain, y2 = (c.y2 - q.pad_y) / q.gain;
        x1 = rintf(fminf(fmaxf(x1, 0.f), q.img_w)); x2 = rintf(fminf(fmaxf(x2, 0.f), q.img_w));
        y1 = rintf(fminf(fmaxf(y1, 0.f), q.img_h)); y2 = rintf(fminf(fmaxf(y2, 0.f), q.img_h));
        float* o = q.dets + ((size_t)frame * q.max_det + (n - 1 - tid)) * 6;
        o[0] = (float)c.cls;
        o[1] = ((x1 + x2) / 2.f) / q.img_w;
        o[2] = ((y1 + y2) / 2.f) / q.img_h;
        o[3] = (x2 - x1) / q.img_w;
        o[4] = (y2 - y1) / q.img_h;
        o[5] = c.score;
    }
    if (tid >= n && tid < q.max_det) {
        float* o = q.dets + ((size_t)frame * q.max_det + tid) * 6;
        for (int i = 0; i < 6; ++i) o[i] = 0.f;
    }
    if (tid == 0) q.counts[frame] = n;
}

hipError_t launch_detect_nms(const DetectParams& q, hipStream_t s) {
    if (q.n_frames <= 0) return hipSuccess;
    if (q.max_det < 1 || q.max_det > DET_MAX || q.nc < 1 || q.nc > 32 || q.rows < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(detect_nms_kernel, dim3(q.n_frames), dim3(NMS_THREADS), 0, s, q);
    return hipGetLastError();
}

}  // namespace pa

// ---- label repair on the device (rows a3 / f1) ---------------------------------------------------------------------------
//
// AIRunner.clean_yolo_crops / clean_yolo_crops_for_fighter (playaid/ai_runner.py:226-289, 306-424) on the detection table
// pa_detect_postprocess wrote, without the label files in between: per fighter the reference resolves duplicate
// detections of its class (nearest centre, L1, to the class's previous box), interpolates the boxes of frames it is
// missing in (measured from the END frame, pixels from VideoCapture position j = one decoded frame late), and copies the
// last crop file of the fighter whose detections end first. The host mirror (playaid_core_amd/label_cleaning.py) takes
// the same decisions on the label TEXT; to agree with it bit for bit the rows are first taken through the label file's
// '%g' formatting (six significant digits) as detect.py writes and float() reads them.
namespace pa {
namespace {

__device__ double pow10i(int k) {  // 10^k, k in -22 .. 22 (exact in double from 0 up)
    const double t[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                          1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    k = k < -22 ? -22 : (k > 22 ? 22 : k);
    return k >= 0 ? t[k] : 1.0 / t[-k];
}

// float(('%g' % v)): v rounded to six significant decimal digits, as a double
__device__ double g6(float vf) {
    const double v = (double)vf;
    if (v == 0.0 || !(fabs(v) < 1e30)) return v;
    const double a = fabs(v);
    int e = (int)floor(log10(a));
    double scaled = e <= 5 ? a * pow10i(5 - e) : a / pow10i(e - 5);
    if (scaled < 99999.5) { --e; scaled = e <= 5 ? a * pow10i(5 - e) : a / pow10i(e - 5); }
    if (scaled >= 999999.5) { ++e; scaled = e <= 5 ? a * pow10i(5 - e) : a / pow10i(e - 5); }
    double r = rint(scaled);
    if (r >= 1e6) { r /= 10.0; ++e; }
    const double out = e <= 5 ? r / pow10i(5 - e) : r * pow10i(e - 5);
    return v < 0 ? -out : out;
}

struct Lab { double v[6]; };  // cls cx cy w h conf

// every value of the detection table through '%g' once, in parallel: the walk below is ONE thread per fighter, and ~20 of
// these roundings (log10, powers of ten, fp64) per frame on its dependent chain made it 0.5-0.8 ms per 64-frame clip
__global__ __launch_bounds__(256) void g6_rows_kernel(const float* __restrict__ dets, double* __restrict__ out, int total) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < total) out[i] = g6(dets[i]);
}

__global__ void clean_labels_kernel(const CleanParams p) {
    // one thread per fighter; the frames are walked in order (every step depends on the previous one)
    __shared__ int last_sh[4];
    __shared__ int maxf_sh;
    const int f = threadIdx.x;
    const int F = p.fighters, n = p.n_labels, md = p.max_det;
    if (f < 4) last_sh[f] = 0;
    if (f == 0) {
        int mf = 0;
        for (int i = 0; i < n; ++i)
            if (p.counts[i] > 0) mf = i + 1;  // the last label file that is not empty (:244-245)
        maxf_sh = mf;
        p.info[0] = mf;
        p.info[1] = 0; p.info[2] = 0; p.info[3] = 0;
    }
    __syncthreads();
    const int maxf = maxf_sh;
    // rows behind the last labelled frame hold "no crop": callers count crop kinds over the whole table, and the table may be
    // recycled memory (an earlier clip's kinds)
    for (int i = maxf * F + f; i < n * F; i += blockDim.x) {
        p.pixel_frame[i] = -1;
        p.crop_kind[i] = 0;
    }
    if (f >= F) return;
    const int cid = p.class_ids[f];
    auto fail = [&](int code, int frame) {
        if (atomicCAS(&p.info[1], 0, code) == 0) p.info[2] = frame;
    };
    auto LAB = [&](int i) -> double* { return p.lab + ((size_t)i * F + f) * 6; };
    // -- duplicates (:314-359) and the detector's own crops
    bool have_prev = false;
    double pcx = 0, pcy = 0;
    for (int i = 0; i < maxf; ++i) {
        double* L = LAB(i);
        p.pixel_frame[i * F + f] = -1;
        p.crop_kind[i * F + f] = 0;
        for (int k = 0; k < 4; ++k) p.pixel_box[((size_t)i * F + f) * 4 + k] = 0.0;
        for (int k = 0; k < 6; ++k) p.crop_row[((size_t)i * F + f) * 6 + k] = 0.f;
        L[0] = -1.0;
        int cnt = 0, first = -1, best = -1;
        double bestd = 10000.0;
        const int c = p.counts[i] < md ? p.counts[i] : md;
        for (int k = 0; k < c; ++k) {
            const double* r = p.g6v + ((size_t)i * md + k) * 6;
            if ((int)r[0] != cid) continue;
            ++cnt;
            if (first < 0) first = k;
            if (have_prev) {
                const double d = fabs(r[1] - pcx) + fabs(r[2] - pcy);
                if (d < bestd) { bestd = d; best = k; }
            }
        }
        if (cnt == 0) continue;
        int pick = first;
        if (cnt > 1) {
            if (!have_prev) { fail(1, i + 1); return; }  // "We should have cleaned out the duplicates at this point" (:343)
            pick = best;
            atomicAdd(&p.info[3], 1);
        }
        const double* r = p.g6v + ((size_t)i * md + pick) * 6;
        for (int k = 0; k < 6; ++k) L[k] = r[k];
        L[0] = (double)(int)L[0];
        have_prev = true;
        pcx = L[1]; pcy = L[2];
        p.pixel_frame[i * F + f] = i;
        for (int k = 0; k < 4; ++k) p.pixel_box[((size_t)i * F + f) * 4 + k] = L[1 + k];
        p.crop_kind[i * F + f] = 1;
        // the crop file without a counter in its name is the class's first detection in label order (:247-258)
        const double* r0 = p.g6v + ((size_t)i * md + first) * 6;
        for (int k = 0; k < 6; ++k) p.crop_row[((size_t)i * F + f) * 6 + k] = (float)r0[k];
    }
    // -- gaps (:361-424)
    int latest = 1, last = 0;
    for (int cur = 1; cur <= maxf; ++cur) {
        if (p.crop_kind[(cur - 1) * F + f] != 1) continue;
        last = cur;
        if (cur - latest > 1) {
            const double* S = LAB(latest - 1);
            if (S[0] < 0) { fail(2, latest); return; }  // "missing start_yolo_crop" (:375-378)
            const double* E = LAB(cur - 1);
            Lab s, e;
            for (int k = 0; k < 6; ++k) { s.v[k] = S[k]; e.v[k] = E[k]; }
            for (int j = latest + 1; j < cur; ++j) {
                const double pct = (double)(cur - j) / (double)(cur - latest);  // measured from the END frame (:389-390)
                double* L = LAB(j - 1);
                L[0] = s.v[0];
                for (int k = 1; k < 6; ++k) L[k] = s.v[k] + (pct * (e.v[k] - s.v[k]));
                const int o = (j - 1) * F + f;
                if (j < p.n_decoded) {  // VideoCapture position j = decoded frame index j (:405-406)
                    p.pixel_frame[o] = j;
                    for (int k = 0; k < 4; ++k) p.pixel_box[(size_t)o * 4 + k] = L[1 + k];
                    p.crop_kind[o] = 2;
                } else {  // the read failed: the previous frame's crop image is copied (:407-416)
                    const int q = (j - 2) * F + f;
                    p.pixel_frame[o] = p.pixel_frame[q];
                    for (int k = 0; k < 4; ++k) p.pixel_box[(size_t)o * 4 + k] = p.pixel_box[(size_t)q * 4 + k];
                    p.crop_kind[o] = p.crop_kind[q];
                    for (int k = 0; k < 6; ++k) p.crop_row[(size_t)o * 6 + k] = p.crop_row[(size_t)q * 6 + k];
                }
            }
        }
        latest = cur;
    }
    // -- tail (:270-289): the fighter whose crops end first gets its last crop file copied up to, not including, the
    // other's last frame. "Last" = the last frame that holds a crop of any kind.
    int lf = 0;
    for (int i = 0; i < maxf; ++i)
        if (p.pixel_frame[i * F + f] >= 0) lf = i + 1;
    last_sh[f] = lf;
    __syncthreads();
    if (lf == 0) { fail(3, 0); return; }  // a fighter without any detection
    int mx = 0;
    for (int q = 0; q < F; ++q) mx = last_sh[q] > mx ? last_sh[q] : mx;
    const int src = (lf - 1) * F + f;
    for (int i = lf; i < mx; ++i) {
        const int o = (i - 1) * F + f;
        p.pixel_frame[o] = p.pixel_frame[src];
        for (int k = 0; k < 4; ++k) p.pixel_box[(size_t)o * 4 + k] = p.pixel_box[(size_t)src * 4 + k];
        p.crop_kind[o] = p.crop_kind[src];
        for (int k = 0; k < 6; ++k) p.crop_row[(size_t)o * 6 + k] = p.crop_row[(size_t)src * 6 + k];
    }
    (void)last;
}

}  // namespace

hipError_t launch_clean_labels(const CleanParams& p, hipStream_t s) {
    const int total = p.n_labels * p.max_det * 6;
    hipLaunchKernelGGL(g6_rows_kernel, dim3((total + 255) / 256), dim3(256), 0, s, p.dets, p.g6v, total);
    hipLaunchKernelGGL(clean_labels_kernel, dim3(1), dim3(64), 0, s, p);
    return hipGetLastError();
}


// ---- what the crop hand-off needs from the repaired table (detector_path.py), on the device ---------------------------------
//
// The tables clean_labels_kernel wrote decide three launches: which detection each (frame, fighter) entry's save_one_box crop
// is cut from, which entries are square_crop repairs (their boxes and source frames, in entry order) and how many there are.
// PyTorch's generic kernels did this with ~12 launches per clip (where / full_like / argsort / gathers / arange); it is one
// workgroup's worth of work.
namespace {

// det_index / src_own: the save_one_box inputs (entries of kind 1). rep_*: the entries of kind 2, compacted in ascending entry
// order and padded with copies of the first to a whole number of frames (pa_square_crops_src cuts F crops per "frame").
// words5 = info4 + the number of repairs.
__global__ __launch_bounds__(256) void detector_plan_kernel(const int32_t* __restrict__ pixel_frame, const double* __restrict__ pixel_box,
                                                            const int32_t* __restrict__ crop_kind, const int32_t* __restrict__ info4, int n_labels, int F,
                                                            int32_t* __restrict__ det_index, int32_t* __restrict__ src_own, int32_t* __restrict__ rep_entry,
                                                            double* __restrict__ rep_boxes, int32_t* __restrict__ rep_src, int32_t* __restrict__ words5) {
    __shared__ int wave_cnt[4];
    __shared__ int base_sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int maxf = info4[0];
    const int total = n_labels * F;
    if (tid == 0) base_sh = 0;
    __syncthreads();
    for (int e0 = 0; e0 < total; e0 += 256) {
        const int e = e0 + tid;
        int kind = 0;
        if (e < total) {
            const int row = e / F;
            kind = row < maxf ? crop_kind[e] : 0;
            det_index[e] = kind == 1 ? e - row * F : -1;
            src_own[e] = kind == 1 ? pixel_frame[e] : 0;
        }
        const unsigned long long m = __ballot(kind == 2);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = base_sh;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (kind == 2) {
            const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
            rep_entry[pos] = e;
            rep_src[pos] = pixel_frame[e];
            for (int k = 0; k < 4; ++k) rep_boxes[(size_t)pos * 4 + k] = pixel_box[(size_t)e * 4 + k];
        }
        __syncthreads();
        if (tid == 0) base_sh += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    const int n_rep = base_sh;
    const int padded = ((n_rep + F - 1) / F) * F;
    for (int pos = n_rep + tid; pos < padded; pos += 256) {  // (n_rep > 0 here)
        rep_entry[pos] = -1;
        rep_src[pos] = rep_src[0];
        for (int k = 0; k < 4; ++k) rep_boxes[(size_t)pos * 4 + k] = rep_boxes[k];
    }
    if (tid < 4) words5[tid] = info4[tid];
    if (tid == 4) words5[4] = n_rep;
}

// the descriptors of a clip's crop images once every chunk of frames has packed its own region: the detector's crops move by
// their chunk's region, the repairs' 128 x 128 images sit back to back from `rep_base`
__global__ __launch_bounds__(256) void detector_desc_kernel(CropImageDesc* __restrict__ desc, const int32_t* __restrict__ crop_kind, int n_entries, int F,
                                                            int step_frames, long long region, const int32_t* __restrict__ rep_entry, int n_rep,
                                                            long long rep_base) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_entries && crop_kind[i] == 1) desc[i].offset += (long long)((i / F) / step_frames) * region;
    if (i < n_rep) {
        CropImageDesc d;
        d.offset = rep_base + (long long)i * (128 * 128 * 3);
        d.height = 128;
        d.width = 128;
        desc[rep_entry[i]] = d;
    }
}

}  // namespace

hipError_t launch_detector_plan(const int32_t* pixel_frame, const double* pixel_box, const int32_t* crop_kind, const int32_t* info4, int n_labels, int F,
                                int32_t* det_index, int32_t* src_own, int32_t* rep_entry, double* rep_boxes, int32_t* rep_src, int32_t* words5,
                                hipStream_t s) {
    hipLaunchKernelGGL(detector_plan_kernel, dim3(1), dim3(256), 0, s, pixel_frame, pixel_box, crop_kind, info4, n_labels, F, det_index, src_own,
                       rep_entry, rep_boxes, rep_src, words5);
    return hipGetLastError();
}

hipError_t launch_detector_desc(CropImageDesc* desc, const int32_t* crop_kind, int n_entries, int F, int step_frames, long long region,
                                const int32_t* rep_entry, int n_rep, long long rep_base, hipStream_t s) {
    const int n = n_entries > n_rep ? n_entries : n_rep;
    hipLaunchKernelGGL(detector_desc_kernel, dim3((n + 255) / 256), dim3(256), 0, s, desc, crop_kind, n_entries, F, step_frames, region, rep_entry,
                       n_rep, rep_base);
    return hipGetLastError();
}

}  // namespace pa
