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

__global__ __launch_bounds__(256) void detect_nms_kernel(const DetectParams q) {
    __shared__ unsigned long long wave_best[4];
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
        for (int r = tid; r < q.rows; r += 256) {
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
            for (int w = 1; w < 4; ++w) b = wave_best[w] > b ? wave_best[w] : b;
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
    hipLaunchKernelGGL(detect_nms_kernel, dim3(q.n_frames), dim3(256), 0, s, q);
    return hipGetLastError();
}

}  // namespace pa
