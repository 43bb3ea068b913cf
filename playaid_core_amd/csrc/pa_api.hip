// C-ABI host side of the MI355X action-recognition path (include/playaid_hip.h).
// Owns device buffers, folds/re-lays-out the weights, sequences the kernels of
// one forward on the caller's stream. No torch types, no hidden synchronisation
// in the enqueue calls.
#include "../../include/playaid_hip.h"
#include "pa_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace pa;

namespace {

constexpr int WINO_TICKETS = 4096;   // tiles of one Winograd launch that may use split K (more: no split)

struct ConvLayer {
    std::string name;
    int cin, cout, kh, kw, stride;  // logical conv
    int in_hw, out_hw;              // spatial size (square) of input / output interior
    int in_pad, out_pad;            // zero-border widths of the buffers
    int off;                        // tap origin inside the padded input
    int taps, kw_taps, chunk;       // igemm K geometry
    int in_px_stride;               // floats per input pixel
    float* in = nullptr;
    float* out = nullptr;
    float* residual = nullptr;
    float* wgt = nullptr;   // device [cout][taps*chunk]
    float* wino_wgt = nullptr;  // fp32 stride-1 3x3 layers run as Winograd F(2x2, 3x3): the filters in wino.hip's layout, else nullptr
    int wino_bn = 0;            // ... laid out for workgroups of this many output channels (wino_pick_bn at the full batch)
    float* bias = nullptr;  // device [cout]
    int relu = 1;
    GemmTile tile = TILE_128x64;
    int splitk = 1;
    bool forced = false;  // tile/split-K pinned by PA_FORCE_* (tuning), else chosen per launch
    // fused 1x1 stride-2 downsample branch (block 0 of layers 2-4): extra K read from `in2`
    float* in2 = nullptr;
    int in2_c = 0, in2_hw = 0, in2_stride = 1;
    // bf16 path: the downsample branch is its own small GEMM into `ds_out` (bf16, same padded layout as `out`),
    // which the 3x3 conv then adds as its residual (patchconv_bf16.hip has no second source)
    float* ds_wgt = nullptr;  // [cout][in2_c] bf16
    float* ds_out = nullptr;
    // ... unless the block's stride-2 opener computes it on its centre tap (igemm_bf16.hip, DS): set on the OPENER
    // (`ds_next_*` = the next layer's ds_wgt / ds_out) and on the conv that adds the result (`ds_fused`)
    float* ds_next_wgt = nullptr;
    float* ds_next_out = nullptr;
    bool ds_fused = false;
    bool ds_probe = false;  // PA_BF16_DS_FUSE=2: the opener takes the fused launch's tile and K split, the branch stays separate
    double k_alg = 0;  // algorithmic K (unpadded) for FLOP accounting
    // compute_dtype PA_DTYPE_EMULATED_F32: the layer's weights as three bf16 slices in psgemm.hip's stage-image order (the layer then
    // runs on that kernel), and the same for the 1x1/2 branch GEMM of a Winograd-form block opener's conv2
    unsigned short* split_wgt = nullptr;
    unsigned short* ds_split_wgt = nullptr;
};

struct ProfEntry {
    int name_id;
    hipEvent_t start, stop;
    double flops, bytes;
    double flops_executed;  // = flops unless the launch runs an algorithm that executes fewer (Winograd: 4 / 9)
};

}  // namespace

struct pa_engine {
    pa_config cfg;
    std::string last_error;
    int max_crops = 0;   // crops per backbone call
    bool emu = false;    // cfg.compute_dtype == PA_DTYPE_EMULATED_F32: fp32 buffers and interfaces, the layers that have ConvLayer::split_wgt on psgemm.hip
    bool bf16 = false;   // cfg.compute_dtype == PA_DTYPE_BF16: the 3x3 conv stack stores bf16 (buffers keep their fp32 size)
    int cache_rows = 0;  // feature-cache rows
    // clip state
    int clip_frames = 0;
    int sub_frames = 0;  // > 0: the clip is a batch of independent clips of this many frames (pa_clip_begin_batch)
    int* gate_flag = nullptr;    // coherent pinned word the host opens pa_stream_gate's kernel with
    int jpeg_quality = 0;        // > 0: every crop goes through a baseline-JPEG write + read (pa_set_crop_jpeg_quality)
    int32_t* jpeg_qtab = nullptr; // [2][64] device
    uint8_t* crops_tmp = nullptr; // [max_crops][128][128][3]: the crops when the caller did not ask for them
    std::vector<char> ready;
    // device memory (all freed in pa_destroy)
    std::vector<void*> allocs;
    // every folded / re-laid-out weight tensor lives in ONE device arena at offsets that depend only on
    // (S, A, compute_dtype): a rank that received the arena over RCCL adopts it with one device copy
    // (pa_create_from_arena) instead of folding the blob again
    uint8_t* arena = nullptr;
    size_t arena_cap = 0, arena_used = 0;
    std::vector<uint8_t> arena_stage;  // host mirror while pa_create folds (empty in adopt mode and afterwards)
    bool adopt = false;
    int32_t* dev_errors = nullptr;  // [4] device-side error counters ([0] = scatter ids outside the clip, [1] = crop images that did not fit)
    int32_t* savebox_rects = nullptr;  // [max_crops][4] scratch of pa_save_one_box_crops
    double* clean_g6 = nullptr;        // scratch of pa_clean_detections: the detection table through '%g' (grown on demand)
    size_t clean_g6_cap = 0;
    float* x0 = nullptr;      // slot 0 of the model-input double buffer [max_crops][134][134][4]
    float* x0_slot[2] = {nullptr, nullptr};
    int32_t* pre_status[2] = {nullptr, nullptr};  // per-slot crop status written by the preprocess stage
    float* pooled = nullptr;  // [max_crops][512]
    float* feats_tmp = nullptr;  // [max_crops][1024] (b1 path)
    float* cache = nullptr;      // [cache_rows][1024]
    int32_t* cache_status = nullptr;  // [cache_rows]
    float* h1 = nullptr;              // [max_crops][512]
    int32_t* gather = nullptr;        // [max_crops][S]
    int gather_key[4] = {-1, -1, -1, -1}; // (f0, cnt, clip_frames, sub_frames) the table in `gather` was built for; -1 = none (other users of the buffer reset it)
    float* slab = nullptr;
    size_t slab_floats = 0;
    float* wino_tickets = nullptr;  // int32[2][WINO_TICKETS], zero between launches: split-K tickets of the Winograd kernel, one set per half batch
    std::vector<ConvLayer> convs;  // stem + 19 convs
    ConvLayer fc;
    float* stem_wgt_bf16 = nullptr;  // bf16 path: the folded stem weights [64][224] as bf16
    float* c1 = nullptr;  // stem output [max_crops][66][66][64] (only the unfused stem + pool fallback writes it)
    float* p1 = nullptr;  // maxpool output [max_crops][34][34][64]
    float* layer4_out = nullptr;
    // head weights
    float* w1d = nullptr;  // [512][S*1024]
    float* b1d = nullptr;
    float *w2 = nullptr, *b2 = nullptr, *w3 = nullptr, *b3 = nullptr;
    GemmTile head_tile = TILE_64x64;
    int head_splitk = 16;
    // preprocess scratch
    CropPlan* plans = nullptr;
    int32_t* coef = nullptr;
    int coef_dim = 0;
    int32_t* coef_cache = nullptr;  // Pillow tables of the passes (2 * (d / 2) + 2 * padding -> d), built for one padding at a time
    int coef_cache_pad = -1, coef_cache_dmax = 0;
    AreaTabPacked* area_tabs = nullptr;
    uint8_t *t1 = nullptr, *t2 = nullptr;
    size_t t_stride = 0;
    int32_t* status_tmp = nullptr;
    int32_t* fallback = nullptr;  // [1 + max_crops]: count, then crop indices
    // two-way interleave of the backbone (two half batches on two streams)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_ds_fork = nullptr, ev_ds_join = nullptr;   // PA_DS_SIDE: a block's 1x1/2 branch GEMM on the side stream, under its opener
    int interleave = 0;  // measured +0.5 % only (kernels of two streams do not overlap usefully); kept as a knob
    // profiling
    bool profiling = false;
    bool profile_layers = false;  // PA_PROFILE_LAYERS=1: one row per conv layer instead of per kernel family
    std::vector<std::string> prof_names;
    std::vector<ProfEntry> prof_log;
    std::vector<hipEvent_t> event_pool;
};

namespace {

int fail(pa_engine* e, int code, const std::string& msg) {
    if (e) e->last_error = msg;
    return code;
}

#define HIPCHK(e, call)                                                                              \
    do {                                                                                             \
        hipError_t err__ = (call);                                                                   \
        if (err__ != hipSuccess)                                                                     \
            return fail((e), PA_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(err__));      \
    } while (0)

template <typename T>
int dev_alloc(pa_engine* e, T** ptr, size_t count, bool zero) {
    void* p = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    HIPCHK(e, hipMalloc(&p, bytes));
    e->allocs.push_back(p);
    if (zero) HIPCHK(e, hipMemset(p, 0, bytes));
    *ptr = reinterpret_cast<T*>(p);
    return PA_OK;
}

// Bump-allocate `bytes` in the weight arena (256-byte aligned) and, unless the arena is being adopted,
// stage the host data for the single upload at the end of pa_create.
int arena_put(pa_engine* e, void** dst, const void* host, size_t bytes) {
    const size_t off = (e->arena_used + 255) & ~(size_t)255;
    if (off + bytes > e->arena_cap) return fail(e, PA_ERR_BAD_WEIGHTS, "weight arena overflow");
    e->arena_used = off + bytes;
    *dst = e->arena + off;
    if (!e->adopt) memcpy(e->arena_stage.data() + off, host, bytes);
    return PA_OK;
}

int upload(pa_engine* e, float** dst, const std::vector<float>& host) {
    return arena_put(e, reinterpret_cast<void**>(dst), host.data(), host.size() * sizeof(float));
}

// fp32 -> bf16 (round to nearest even) weights of the bf16 conv path; the device pointer keeps
// the float* type of ConvLayer::wgt, the kernels reinterpret it.
int upload_bf16(pa_engine* e, float** dst, const std::vector<float>& host) {
    std::vector<uint16_t> h(host.size());
    if (!e->adopt)
        for (size_t i = 0; i < host.size(); ++i) {
            uint32_t u;
            memcpy(&u, &host[i], 4);
            u += 0x7fffu + ((u >> 16) & 1u);
            h[i] = (uint16_t)(u >> 16);
        }
    return arena_put(e, reinterpret_cast<void**>(dst), h.data(), h.size() * sizeof(uint16_t));
}

// fp32 [cout][ktot] -> three bf16 slices per weight in psgemm.hip's stage-image order (PA_DTYPE_EMULATED_F32)
int upload_split(pa_engine* e, unsigned short** dst, const std::vector<float>& host, int cout, int ktot, bool residual) {
    std::vector<unsigned short> h(psgemm_weight_elems(cout, ktot, residual), 0);
    if (h.empty()) return PA_ERR_INVALID_ARG;
    if (!e->adopt) psgemm_pack_weights(host.data(), cout, ktot, residual, h.data());
    return arena_put(e, reinterpret_cast<void**>(dst), h.data(), h.size() * sizeof(unsigned short));
}

// --- weight blob walking -----------------------------------------------------
struct BlobReader {
    const float* base;
    size_t pos, limit;  // in floats
    bool dry;  // adopting a prepared arena: only the sizes matter, nothing returned by take() is read
    const float* take(size_t n) {
        if (pos + n > limit) return nullptr;
        const float* r = dry ? base : base + pos;
        pos += n;
        return r;
    }
};

size_t blob_float_count(int S, int A) {
    size_t n = 0;
    auto conv = [&](int co, int ci, int k) { n += (size_t)co * ci * k * k; };
    auto bn = [&](int c) { n += 4 * (size_t)c; };
    conv(64, 3, 7);
    bn(64);
    int cin = 64;
    const int widths[4] = {64, 128, 256, 512};
    for (int li = 0; li < 4; ++li) {
        const int co = widths[li];
        for (int b = 0; b < 2; ++b) {
            const int ci = b == 0 ? cin : co;
            conv(co, ci, 3);
            bn(co);
            conv(co, co, 3);
            bn(co);
            if (b == 0 && li > 0) {
                conv(co, ci, 1);
                bn(co);
            }
        }
        cin = co;
    }
    n += 1000 * 512 + 1000;
    n += (size_t)512 * 1000 * S + 512;
    n += 128 * 512 + 128;
    n += (size_t)A * 128 + A;
    return n;
}

// Fold eval-mode BatchNorm (eps 1e-5) into conv weights in fp64 and re-lay-out
// OIHW -> [cout][tap][chunk] (tap = ky*kw_taps + kx, channels innermost, zero
// padded to `chunk`). The 7x7 stem uses tap = ky and chunk = 8 px * 4 ch.
bool fold_conv(BlobReader& br, const ConvLayer& L, std::vector<float>& w_out, std::vector<float>& b_out) {
    const size_t wn = (size_t)L.cout * L.cin * L.kh * L.kw;
    const float* w = br.take(wn);
    const float* gamma = br.take(L.cout);
    const float* beta = br.take(L.cout);
    const float* mean = br.take(L.cout);
    const float* var = br.take(L.cout);
    if (!w || !gamma || !beta || !mean || !var) return false;
    const int ktot = L.taps * L.chunk;
    w_out.assign((size_t)L.cout * ktot, 0.f);
    b_out.assign(L.cout, 0.f);
    if (br.dry) return true;
    const bool stem = (L.kh == 7);
    for (int co = 0; co < L.cout; ++co) {
        const double scale = (double)gamma[co] / std::sqrt((double)var[co] + 1e-5);
        b_out[co] = (float)((double)beta[co] - (double)mean[co] * scale);
        for (int ci = 0; ci < L.cin; ++ci)
            for (int ky = 0; ky < L.kh; ++ky)
                for (int kx = 0; kx < L.kw; ++kx) {
                    const double v = (double)w[(((size_t)co * L.cin + ci) * L.kh + ky) * L.kw + kx] * scale;
                    size_t k;
                    if (stem)
                        k = (size_t)ky * L.chunk + kx * 4 + ci;
                    else
                        k = (size_t)(ky * L.kw + kx) * L.chunk + ci;
                    w_out[(size_t)co * ktot + k] = (float)v;
                }
    }
    return true;
}

void choose_tile(int M, int N, int nk, GemmTile* tile, int* splitk) {
    // Aim for >= 2 workgroups per CU (512) so that the un-pipelined K loop of
    // one workgroup hides behind another's; fall back to split-K when even the
    // smallest tile cannot fill the chip.
    const int t128x128 = (N % 128 == 0) ? ((M + 127) / 128) * (N / 128) : 0;
    const int t128x64 = ((M + 127) / 128) * (N / 64);
    const int t64x64 = ((M + 63) / 64) * (N / 64);
    int tiles;
    if (t128x128 >= 512) {
        *tile = TILE_128x128;
        tiles = t128x128;
    } else if (t128x64 >= 512) {
        *tile = TILE_128x64;
        tiles = t128x64;
    } else {
        *tile = TILE_64x64;
        tiles = t64x64;
    }
    int sk = 1;
    while (tiles * sk < 512 && nk / (sk * 2) >= 8) sk *= 2;
    *splitk = sk;
}

// --- profiling ------------------------------------------------------------------
int prof_name_id(pa_engine* e, const char* name) {
    for (size_t i = 0; i < e->prof_names.size(); ++i)
        if (e->prof_names[i] == name) return (int)i;
    e->prof_names.push_back(name);
    return (int)e->prof_names.size() - 1;
}

hipEvent_t get_event(pa_engine* e) {
    if (!e->event_pool.empty()) {
        hipEvent_t ev = e->event_pool.back();
        e->event_pool.pop_back();
        return ev;
    }
    hipEvent_t ev;
    if (hipEventCreate(&ev) != hipSuccess) return nullptr;
    return ev;
}

struct ProfScope {
    pa_engine* e;
    hipStream_t s;
    ProfEntry ent;
    bool on;
    ProfScope(pa_engine* e_, hipStream_t s_, const char* name, double flops, double bytes, double executed = -1.0)
        : e(e_), s(s_), on(e_->profiling) {
        if (!on) return;
        ent.name_id = prof_name_id(e, name);
        ent.flops = flops;
        ent.flops_executed = executed >= 0.0 ? executed : flops;
        ent.bytes = bytes;
        ent.start = get_event(e);
        ent.stop = get_event(e);
        if (!ent.start || !ent.stop) {
            on = false;
            return;
        }
        (void)hipEventRecord(ent.start, s);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(ent.stop, s);
        e->prof_log.push_back(ent);
    }
};

// crop0: first crop of this launch inside the layer's buffers (the two interleaved half
// batches address disjoint crop ranges of the same buffers); slab_off: its split-K slab region.
// ds_stream: where the 1x1/2 branch GEMM of a Winograd-form block opener's conv2 goes (nullptr: `s`, right in front of the convolution);
// ds_only: launch just that GEMM and return; ds_done: it has been launched already (by a ds_only call), only its result is added
int run_conv(pa_engine* e, const ConvLayer& L, int crop0, int ncrops, size_t slab_off, hipStream_t s, const char* prof_name,
             bool ds_only = false, bool ds_done = false) {
    GemmParams p;
    memset(&p, 0, sizeof(p));
    const int in_w = L.in_hw + 2 * L.in_pad;
    const int out_w = L.out_hw + 2 * L.out_pad;
    const size_t in_crop = (size_t)in_w * in_w * L.in_px_stride;
    const size_t out_crop = (size_t)out_w * out_w * L.cout;
    const bool bf = e->bf16 && L.kh == 3;  // bf16 conv path: the 3x3 stack (buffers addressed in 2-byte elements)
    const bool bf_ds = bf && L.in2 && L.ds_wgt;  // downsample branch as its own GEMM, added as the residual
    const bool f32_ds = !bf && L.in2 && L.ds_wgt && L.wino_wgt;  // the same for the fp32 Winograd form of the layer
    const size_t es = bf ? 2 : 4;
    auto at = [es](float* base, size_t elems) { return reinterpret_cast<float*>(reinterpret_cast<char*>(base) + elems * es); };
    p.act = at(L.in, crop0 * in_crop);
    p.wgt = L.wgt;
    p.bias = L.bias;
    p.residual = L.residual ? at(L.residual, crop0 * out_crop) : nullptr;
    p.out = at(L.out, crop0 * out_crop);
    p.slab = e->slab + slab_off;
    p.gather = nullptr;
    p.M = ncrops * L.out_hw * L.out_hw;
    p.N = L.cout;
    p.taps = L.taps;
    p.kw_taps = L.kw_taps;
    p.chunk = L.chunk;
    p.ktot = L.taps * L.chunk;
    p.skip_w = L.kh == 7;  // stem: 3 channels in 4-float pixels
    p.howo = L.out_hw * L.out_hw;
    p.wo = L.out_hw;
    p.in_px_stride = L.in_px_stride;
    p.in_row_stride = in_w * L.in_px_stride;
    p.in_img_stride = (int)in_crop;
    p.stride = L.stride;
    p.off_y = L.off;
    p.off_x = L.off;
    p.out_px_stride = L.cout;
    p.out_row_stride = out_w * L.cout;
    p.out_img_stride = (int)out_crop;
    p.out_pad = L.out_pad;
    p.relu = L.relu;
    if (bf_ds && L.ds_fused) {
        p.residual = at(L.ds_out, crop0 * out_crop);  // written by the block's opener
    } else if (bf_ds) {
        GemmParams d;
        memset(&d, 0, sizeof(d));
        const int w2 = L.in2_hw + 2;  // zero-bordered block input
        d.act = at(L.in2, (size_t)crop0 * w2 * w2 * L.in2_c);
        d.wgt = L.ds_wgt;
        d.out = at(L.ds_out, crop0 * out_crop);
        d.slab = p.slab;
        d.M = p.M; d.N = p.N;
        d.taps = 1; d.kw_taps = 1; d.chunk = L.in2_c; d.ktot = L.in2_c;
        d.howo = p.howo; d.wo = p.wo;
        d.in_px_stride = L.in2_c; d.in_row_stride = w2 * L.in2_c; d.in_img_stride = w2 * w2 * L.in2_c;
        d.stride = L.in2_stride; d.off_y = 1; d.off_x = 1;
        d.out_px_stride = p.out_px_stride; d.out_row_stride = p.out_row_stride; d.out_img_stride = p.out_img_stride;
        d.out_pad = p.out_pad;
        d.relu = 0; d.splitk = 1;
        ProfScope ps(e, s, prof_name, 2.0 * d.M * d.N * L.in2_c,
                     2.0 * ((double)ncrops * L.in2_hw * L.in2_hw * L.in2_c / (L.in2_stride * L.in2_stride) + (double)d.M * d.N + (double)d.N * L.in2_c));
        HIPCHK(e, launch_igemm_bf16(d, TILE_128x64, s));  // (128x128 / 256x128 tiles measured 2-7 us slower here)
        p.residual = d.out;
    } else if (f32_ds) {
        GemmParams d;
        memset(&d, 0, sizeof(d));
        const int w2 = L.in2_hw + 2;  // zero-bordered block input
        d.act = L.in2 + (size_t)crop0 * w2 * w2 * L.in2_c;
        d.wgt = L.ds_wgt;
        d.out = L.ds_out + crop0 * out_crop;
        d.slab = p.slab;
        d.M = p.M; d.N = p.N;
        d.taps = 1; d.kw_taps = 1; d.chunk = L.in2_c; d.ktot = L.in2_c;
        d.howo = p.howo; d.wo = p.wo;
        d.in_px_stride = L.in2_c; d.in_row_stride = w2 * L.in2_c; d.in_img_stride = w2 * w2 * L.in2_c;
        d.stride = L.in2_stride; d.off_y = 1; d.off_x = 1;
        d.out_px_stride = p.out_px_stride; d.out_row_stride = p.out_row_stride; d.out_img_stride = p.out_img_stride;
        d.out_pad = p.out_pad;
        d.relu = 0; d.splitk = 1;
        ProfScope ps(e, s, prof_name, ds_done ? 0.0 : 2.0 * d.M * d.N * L.in2_c,
                     ds_done ? 0.0 : 4.0 * ((double)ncrops * L.in2_hw * L.in2_hw * L.in2_c / (L.in2_stride * L.in2_stride) + (double)d.M * d.N + (double)d.N * L.in2_c));
        if (ds_done) {
        } else if (e->emu && L.ds_split_wgt) {
            HIPCHK(e, launch_psgemm(d, L.ds_split_wgt, (size_t)ncrops * out_crop, 0, s));
        } else {
            GemmTile dt;
            int dsk;
            choose_tile(d.M, d.N, L.in2_c / 32, &dt, &dsk);
            HIPCHK(e, launch_igemm(d, dt, s));
        }
        if (ds_only) return PA_OK;
        p.residual = d.out;
    } else if (L.in2) {
        const int w2 = L.in2_hw + 2;  // zero-bordered block input
        p.act2 = at(L.in2, (size_t)crop0 * w2 * w2 * L.in2_c);
        p.k2_steps = L.in2_c / (bf ? 64 : 32);
        p.in2_px_stride = L.in2_c;
        p.in2_row_stride = w2 * L.in2_c;
        p.in2_img_stride = w2 * w2 * L.in2_c;
        p.stride2 = L.in2_stride;
        p.off2 = 1;
        p.ktot += L.in2_c;
    }
    GemmTile tile = L.tile;
    int splitk = L.splitk;
    if (!L.forced) choose_tile(p.M, p.N, p.ktot / (bf ? 64 : 32), &tile, &splitk);  // per launch: M depends on the batch
    p.splitk = splitk;
    const size_t slab_avail = e->slab_floats > slab_off ? e->slab_floats - slab_off : 0;
    if ((size_t)p.splitk * p.M * p.N > slab_avail) p.splitk = 1;
    const bool ds_here = bf && L.ds_next_wgt;  // this launch also computes the next conv's 1x1/2 branch
    if (ds_here) {
        p.wgt2 = L.ds_next_wgt;
        p.out2 = at(L.ds_next_out, crop0 * out_crop);
    }
    if (ds_here || (bf && L.ds_probe)) {
        tile = (p.N % 128 == 0 && ((p.M + 127) / 128) * (p.N / 128) >= 256) ? TILE_128x128 : TILE_128x64;
        p.splitk = 1;
    }
    // (the branch's FLOPs and bytes are booked where it runs: its own launch, or the opener's)
    const double k_main = ((bf_ds || f32_ds) ? L.k_alg - L.in2_c : L.k_alg) + (ds_here ? L.cin : 0);
    const double flops = 2.0 * p.M * p.N * k_main;
    const double bytes = (double)es * ((double)ncrops * L.in_hw * L.in_hw * L.cin + (double)p.M * p.N * ((L.residual || bf_ds || f32_ds || ds_here) ? 2 : 1) +
                                       (double)p.N * k_main + ((L.in2 && !bf_ds && !f32_ds) ? (double)ncrops * L.in2_hw * L.in2_hw * L.in2_c : 0.0));
    const bool as_wino = !bf && L.wino_wgt && !p.act2;
    ProfScope ps(e, s, prof_name, flops, bytes, as_wino ? flops * 4.0 / 9.0 : flops);
    // stride-1 3x3 layers: input patch resident in LDS across the nine taps (patchconv.hip);
    // PA_PATCH=0 keeps the generic im2col engine for A/B runs
    static const int use_patch = getenv("PA_PATCH") ? atoi(getenv("PA_PATCH")) : 1;
    static const int use_bf16_patch = getenv("PA_BF16_PATCH") ? atoi(getenv("PA_BF16_PATCH")) : 1;
    if (e->emu && L.split_wgt && !p.act2 && !p.residual) {
        HIPCHK(e, launch_psgemm(p, L.split_wgt, (size_t)ncrops * out_crop, 0, s));
    } else if (bf) {
        hipError_t pe = hipErrorInvalidValue;
        if (use_bf16_patch && L.stride == 1 && !p.act2) pe = launch_conv3x3_bf16_patch(p, s);
        if (pe == hipErrorInvalidValue) {  // stride-2 convs, fused 1x1/2 second source
            // a 256-pixel tile halves the weight fill per pixel (this kernel is bound by its L2 -> LDS copies) where it
            // still leaves every CU a workgroup; PA_BF16_T256=0 for A/B runs
            static const int t256 = getenv("PA_BF16_T256") ? atoi(getenv("PA_BF16_T256")) : 1;
            if (t256 && !p.out2 && tile == TILE_128x128 && p.splitk == 1 && ((p.M + 255) / 256) * (p.N / 128) >= 256) tile = TILE_256x128;
            pe = launch_igemm_bf16(p, tile, s);
        }
        HIPCHK(e, pe);
    } else if (L.wino_wgt && !p.act2) {
        WinoParams q;
        memset(&q, 0, sizeof(q));
        q.act = p.act; q.wgt = L.wino_wgt; q.bias = p.bias; q.residual = p.residual; q.out = p.out;
        q.n_img = ncrops; q.height = L.out_hw; q.width = L.out_hw; q.cin = L.cin; q.cout = L.cout; q.bn = L.wino_bn;
        q.in_px_stride = p.in_px_stride; q.in_row_stride = p.in_row_stride; q.in_img_stride = p.in_img_stride;
        q.out_px_stride = p.out_px_stride; q.out_row_stride = p.out_row_stride; q.out_img_stride = p.out_img_stride; q.out_pad = p.out_pad;
        q.relu = p.relu;
        if (e->wino_tickets && e->slab) {   // split-K scratch: this half batch's region
            q.slab = e->slab + slab_off;
            q.slab_floats = e->slab_floats > slab_off ? std::min(e->slab_floats - slab_off, e->slab_floats / 2) : 0;
            q.tickets = reinterpret_cast<int32_t*>(e->wino_tickets) + (slab_off ? WINO_TICKETS : 0);
            q.tickets_cap = WINO_TICKETS;
        }
        HIPCHK(e, launch_wino3x3(q, s));
    } else if (use_patch && L.kh == 3 && L.stride == 1) {
        int bm = tile == TILE_64x64 || tile == TILE_64x64_K64 ? 64 : 128;
        const int howo = L.out_hw * L.out_hw, in_w2 = L.out_hw + 2;
        const int px128 = howo >= 128 ? (128 / L.out_hw + 2) * in_w2 : (128 / howo) * in_w2 * in_w2;
        if (bm == 128 && px128 > 224) bm = 64;  // keep two workgroups per CU (2 patch buffers + weight ring <= 80 KB)
        hipError_t pe = launch_conv3x3_patch(p, bm, s);
        if (pe == hipErrorInvalidValue) pe = launch_igemm(p, tile, s);  // geometry the patch kernel does not cover
        HIPCHK(e, pe);
    } else {
        // Stride-2 3x3 convolutions that need no split-K run on the persistent form of the engine (pigemm.hip) with 64-row tiles:
        // two tiles per workgroup slot, so the second tile's first operands arrive under the first one's matrix instructions
        // (one 128-row tile per slot, round 4's A/B, gained nothing: 48.39 k against 48.53 k frames/s). At 128 crops: 50.3 / 50.2 /
        // 54.0 -> 48.8 / 46.6 / 52.7 us for the three openers. Same k order as igemm.hip; the bias is the value pgemm's accumulators start
        // from and igemm's last addition: results agree to that rounding. PA_S2_PGEMM=0: igemm.hip (A/B), 1: pgemm's own choice of tile height
        static const int use_pgemm = getenv("PA_S2_PGEMM") ? atoi(getenv("PA_S2_PGEMM")) : 2;
        hipError_t pe = hipErrorInvalidValue;
        if (use_pgemm && L.kh == 3 && L.stride == 2 && !p.act2 && !p.residual && p.splitk <= 1 && !p.gather) pe = launch_pgemm(p, use_pgemm == 2 ? 64 : 0, s);
        if (pe == hipErrorInvalidValue) pe = launch_igemm(p, tile, s);
        HIPCHK(e, pe);
    }
    return PA_OK;
}

int run_backbone_part(pa_engine* e, int crop0, int ncrops, const float* x_in, float* feats_out, size_t slab_off, hipStream_t s) {
    int rc;
    ConvLayer stem = e->convs[0];
    stem.in = const_cast<float*>(x_in);  // x_in / feats_out are the caller's bases: crop0 is applied by run_conv
    static const bool stem_igemm = getenv("PA_STEM_IGEMM") && atoi(getenv("PA_STEM_IGEMM"));  // A/B knob: generic engine
    auto stem_part = [&](int c0, int n, hipStream_t st) -> int {
        if (stem_igemm && !e->bf16) return run_conv(e, stem, c0, n, slab_off, st, "igemm_conv7x7_stem");
        StemParams sp;
        memset(&sp, 0, sizeof(sp));
        sp.x = x_in + (size_t)c0 * 134 * 134 * 4;
        sp.wgt = stem.wgt;
        sp.bias = stem.bias;
        sp.out = reinterpret_cast<float*>(reinterpret_cast<char*>(e->c1) + (size_t)c0 * 66 * 66 * 64 * (e->bf16 ? 2 : 4));
        sp.tiles = n * 32;
        sp.out_bf16 = e->bf16 ? 1 : 0;
        const double px = (double)n * 64 * 64;
        ProfScope ps(e, st, "stem_conv7x7", 2.0 * px * 64 * 147, 4.0 * ((double)n * 128 * 128 * 3 + px * 64 + 64.0 * 147));
        HIPCHK(e, launch_stem7x7(sp, st));
        return PA_OK;
    };
    auto pool_part = [&](int c0, int n, hipStream_t st) -> int {
        ProfScope ps(e, st, "maxpool3x3", 0.0, (e->bf16 ? 2.0 : 4.0) * n * (64.0 * 64 * 64 + 32.0 * 32 * 64));
        if (e->bf16)
            HIPCHK(e, launch_maxpool_bf16(reinterpret_cast<uint16_t*>(e->c1) + (size_t)c0 * 66 * 66 * 64,
                                          reinterpret_cast<uint16_t*>(e->p1) + (size_t)c0 * 34 * 34 * 64, n, st));
        else
            HIPCHK(e, launch_maxpool(e->c1 + (size_t)c0 * 66 * 66 * 64, e->p1 + (size_t)c0 * 34 * 34 * 64, n, st));
        return PA_OK;
    };
    // Default: stem + BatchNorm + ReLU + max-pool in one persistent kernel (stem_pool.hip), the 64 x 64 stem map never
    // reaches memory. PA_STEM_POOL=0 keeps the two-kernel form for A/B runs: there the max-pool (pure HBM/L2
    // streaming) of one half batch runs on the side stream underneath the stem (pure matrix work) of the other.
    static const int fused_pool = getenv("PA_STEM_POOL") ? atoi(getenv("PA_STEM_POOL")) : 1;
    static const int overlap_pool = getenv("PA_POOL_OVERLAP") ? atoi(getenv("PA_POOL_OVERLAP")) : 1;
    if (fused_pool && !stem_igemm) {
        StemPoolParams sp;
        memset(&sp, 0, sizeof(sp));
        const size_t es_in = e->bf16 ? 2 : 4;
        sp.x = reinterpret_cast<const char*>(x_in) + (size_t)crop0 * 134 * 134 * 4 * es_in;
        sp.wgt = e->bf16 ? e->stem_wgt_bf16 : stem.wgt;
        sp.bias = stem.bias;
        sp.out = reinterpret_cast<char*>(e->p1) + (size_t)crop0 * 34 * 34 * 64 * es_in;
        sp.crops = ncrops;
        sp.in_bf16 = sp.out_bf16 = e->bf16 ? 1 : 0;
        const double px = (double)ncrops * 64 * 64;
        ProfScope ps(e, s, "stem_conv7x7_pool", 2.0 * px * 64 * 147,
                     (double)es_in * ((double)ncrops * 128 * 128 * 4 + (double)ncrops * 32 * 32 * 64 + 64.0 * 147));
        HIPCHK(e, launch_stem_pool(sp, s));
    } else if (e->bf16) {
        return fail(e, PA_ERR_INVALID_ARG, "the bf16 path has no unfused stem (PA_STEM_POOL=0 / PA_STEM_IGEMM=1 are fp32 A/B knobs)");
    } else if (overlap_pool && !e->profiling && !e->interleave && e->side && s != e->side && ncrops >= 64) {
        const int half = (ncrops / 2 + 7) & ~7;
        if ((rc = stem_part(crop0, half, s))) return rc;
        HIPCHK(e, hipEventRecord(e->ev_fork, s));
        HIPCHK(e, hipStreamWaitEvent(e->side, e->ev_fork, 0));
        if ((rc = pool_part(crop0, half, e->side))) return rc;
        HIPCHK(e, hipEventRecord(e->ev_join, e->side));
        if ((rc = stem_part(crop0 + half, ncrops - half, s))) return rc;
        if ((rc = pool_part(crop0 + half, ncrops - half, s))) return rc;
        HIPCHK(e, hipStreamWaitEvent(s, e->ev_join, 0));
    } else {
        if ((rc = stem_part(crop0, ncrops, s))) return rc;
        if ((rc = pool_part(crop0, ncrops, s))) return rc;
    }
    // PA_DS_SIDE=1 (A/B): the 1x1/2 branch GEMM of a block (layers 2-4: ~10 us each, a few dozen tiles) is
    // launched on the side stream BEFORE the block's stride-2 opener instead of between the opener and conv2 -- both read the block
    // input, so the small GEMM runs under the opener and conv2 only waits for an event
    static const int ds_side = getenv("PA_DS_SIDE") ? atoi(getenv("PA_DS_SIDE")) : 0;
    bool ds_pending = false;
    for (size_t i = 1; i < e->convs.size(); ++i) {
        const ConvLayer& L = e->convs[i];
        const char* pn = e->profile_layers ? L.name.c_str() : (L.kh == 3 ? "igemm_conv3x3" : "igemm_conv1x1_ds");
        if (ds_side && !e->profiling && !e->bf16 && e->side && s != e->side && i + 1 < e->convs.size()) {
            const ConvLayer& Nx = e->convs[i + 1];
            if (Nx.in2 && Nx.ds_wgt && Nx.wino_wgt && Nx.in2 == L.in && L.stride == 2) {
                HIPCHK(e, hipEventRecord(e->ev_ds_fork, s));
                HIPCHK(e, hipStreamWaitEvent(e->side, e->ev_ds_fork, 0));
                if ((rc = run_conv(e, Nx, crop0, ncrops, slab_off, e->side, "igemm_conv1x1_ds", /*ds_only=*/true))) return rc;
                HIPCHK(e, hipEventRecord(e->ev_ds_join, e->side));
                ds_pending = true;
            }
        }
        const bool use_pending = ds_pending && L.in2 && L.ds_wgt && L.wino_wgt;
        if (use_pending) HIPCHK(e, hipStreamWaitEvent(s, e->ev_ds_join, 0));
        rc = run_conv(e, L, crop0, ncrops, slab_off, s, pn, false, use_pending);
        if (use_pending) ds_pending = false;
        if (rc) return rc;
    }
    {
        ProfScope ps(e, s, "avgpool", 0.0, ncrops * ((e->bf16 ? 2.0 : 4.0) * 16.0 * 512 + 4.0 * 512));
        if (e->bf16)
            HIPCHK(e, launch_avgpool_bf16(reinterpret_cast<uint16_t*>(e->layer4_out) + (size_t)crop0 * 36 * 512,
                                          e->pooled + (size_t)crop0 * 512, ncrops, s));
        else
            HIPCHK(e, launch_avgpool(e->layer4_out + (size_t)crop0 * 36 * 512, e->pooled + (size_t)crop0 * 512, ncrops, s));
    }
    ConvLayer fc = e->fc;
    fc.out = feats_out;
    rc = run_conv(e, fc, crop0, ncrops, slab_off, s, "igemm_fc");
    return rc;
}

// The backbone of one batch. Large batches are cut into two halves that run on two streams
// (fork/join with events): every layer is then two kernels of half the grid that drift out of
// phase, so the launch gaps, prologues, epilogues and split-K reduces of one half are covered
// by the steady-state MFMA loop of the other (the layers' outside-the-loop time was ~14 %).
int run_backbone(pa_engine* e, int ncrops, const float* x_in, float* feats_out, hipStream_t s) {
    if (!e->interleave || ncrops < 64 || !e->side) return run_backbone_part(e, 0, ncrops, x_in, feats_out, 0, s);
    const int half = (ncrops / 2 + 7) & ~7;
    HIPCHK(e, hipEventRecord(e->ev_fork, s));
    HIPCHK(e, hipStreamWaitEvent(e->side, e->ev_fork, 0));
    int rc = run_backbone_part(e, 0, half, x_in, feats_out, 0, s);
    if (rc) return rc;
    rc = run_backbone_part(e, half, ncrops - half, x_in, feats_out, e->slab_floats / 2, e->side);
    if (rc) return rc;
    HIPCHK(e, hipEventRecord(e->ev_join, e->side));
    HIPCHK(e, hipStreamWaitEvent(s, e->ev_join, 0));
    return PA_OK;
}

int run_head(pa_engine* e, int nwin, const float* feats, const int32_t* gather, const int32_t* crop_status,
             pa_record* records, float* logp, hipStream_t s) {
    const int S = e->cfg.sequence_length;
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.act = feats;
    p.wgt = e->w1d;
    p.bias = e->b1d;
    p.out = e->h1;
    p.slab = e->slab;
    p.gather = gather;
    p.M = nwin;
    p.N = 512;
    p.taps = S;
    p.kw_taps = 1;
    p.chunk = PA_FEATURE_STRIDE;
    p.ktot = S * PA_FEATURE_STRIDE;
    p.howo = 1;
    p.wo = 1;
    p.in_px_stride = PA_FEATURE_STRIDE;
    p.out_px_stride = 512;
    p.out_row_stride = 512;
    p.out_img_stride = 512;
    p.relu = 1;
    p.splitk = e->head_splitk;
    if ((size_t)p.splitk * p.M * p.N > e->slab_floats) p.splitk = 1;
    int32_t splitk_used = 1;
    p.defer_reduce = 1;  // the MLP kernel sums the slabs itself
    p.splitk_used = &splitk_used;
    {
        ProfScope ps(e, s, "igemm_conv1d_head", 2.0 * nwin * 512.0 * 1000.0 * S,
                     4.0 * (512.0 * 1000 * S + (double)nwin * S * 1000 + nwin * 512.0));
        HIPCHK(e, launch_igemm(p, e->head_tile, s));
    }
    HeadParams h;
    memset(&h, 0, sizeof(h));
    h.h1 = splitk_used > 1 ? nullptr : e->h1;
    h.slab = e->slab;
    h.b1 = e->b1d;
    h.splitk = splitk_used;
    h.w2 = e->w2;
    h.b2 = e->b2;
    h.w3 = e->w3;
    h.b3 = e->b3;
    h.logp = logp;
    h.records = records;
    h.crop_status = crop_status;
    h.gather = gather;
    h.nwin = nwin;
    h.num_actions = e->cfg.num_actions;
    h.fighters = e->cfg.num_fighters;
    h.seq = S;
    for (int i = 0; i < 4; ++i) h.class_ids[i] = e->cfg.fighter_class_ids[i];
    {
        ProfScope ps(e, s, "head_mlp_logsoftmax", 2.0 * nwin * (512.0 * 128 + 128.0 * e->cfg.num_actions),
                     4.0 * (nwin * 512.0 + 512.0 * 128 + 128.0 * e->cfg.num_actions));
        HIPCHK(e, launch_head_mlp(h, s));
    }
    return PA_OK;
}

int run_preprocess(pa_engine* e, const uint8_t* frames, int n, int height, int width, const double* boxes, int padding,
                   int swap_rb, uint8_t* crops_u8, float* crops_f32, int32_t* status, hipStream_t s,
                   const int32_t* src_frame = nullptr, int n_src = 0, const pa_crop_window* windows = nullptr) {
    PreprocParams p;
    memset(&p, 0, sizeof(p));
    p.frames = frames;
    p.windows = reinterpret_cast<const CropWindow*>(windows);
    p.boxes = boxes;
    p.src_frame = src_frame;
    p.n_src = src_frame ? n_src : n;
    p.n_frames = n;
    p.height = height;
    p.width = width;
    p.fighters = e->cfg.num_fighters;
    p.padding = padding;
    p.swap_rb = swap_rb;
    p.plans = e->plans;
    p.coef = e->coef;
    p.coef_dim = e->coef_dim;
    // the cache holds the reference's padding (30); any other value computes its tables per crop, as every clipped crop does
    p.coef_cache = e->coef_cache_pad == padding ? e->coef_cache : nullptr;
    p.coef_cache_pad = padding;
    p.coef_cache_dmax = e->coef_cache_dmax;
    p.area_tabs = e->area_tabs;
    p.t1 = e->t1;
    p.t2 = e->t2;
    p.t_stride = e->t_stride;
    p.crops_u8 = crops_u8;
    p.crops_f32 = crops_f32;
    p.crops_f32_is_bf16 = e->bf16 ? 1 : 0;
    p.status = status;
    p.fallback_count = e->fallback;
    p.fallback_list = e->fallback + 4;
    const double ncrops = (double)n * e->cfg.num_fighters;
    ProfScope ps(e, s, "preprocess_crops", 0.0, ncrops * (375.0 * 375 * 3 + 49152.0 * 5));
    if (e->jpeg_quality > 0 && !p.crops_u8) p.crops_u8 = e->crops_tmp;   // the round trip works on the u8 crops
    HIPCHK(e, launch_preprocess(p, s));
    if (e->jpeg_quality > 0) {
        JpegParams j;
        memset(&j, 0, sizeof(j));
        j.crops_u8 = p.crops_u8;
        j.qtab = e->jpeg_qtab;
        j.x0 = crops_f32;
        j.x0_bf16 = e->bf16 ? 1 : 0;
        j.bgr = swap_rb ? 0 : 1;   // frames are B, G, R; swap_rb turns the crops into R, G, B
        ProfScope pj(e, s, "jpeg_roundtrip", 0.0, ncrops * 49152.0 * 2);
        HIPCHK(e, launch_jpeg_roundtrip(j, (int)ncrops, s));
    }
    return PA_OK;
}

}  // namespace

// ===============================================================================
// C ABI
// ===============================================================================

extern "C" {

int pa_abi_version(void) { return PA_ABI_VERSION; }

const char* pa_status_string(int status) {
    switch (status) {
        case PA_OK: return "ok";
        case PA_ERR_INVALID_ARG: return "invalid argument";
        case PA_ERR_HIP: return "HIP runtime error";
        case PA_ERR_BAD_WEIGHTS: return "weight blob does not match the expected layout";
        case PA_ERR_CAPACITY: return "request exceeds the capacity the engine was created with";
        case PA_ERR_NO_DEVICE: return "no usable HIP device";
        case PA_ERR_NOT_READY: return "features of a required frame are not in the cache";
        default: return "unknown status";
    }
}

const char* pa_last_error(const pa_engine* e) { return e ? e->last_error.c_str() : "null engine"; }

size_t pa_weight_blob_bytes(int S, int A) { return 8 * sizeof(int32_t) + blob_float_count(S, A) * sizeof(float); }

}  // extern "C"

namespace {

// pa_create (blob != nullptr: fold the Lightning tensors) and pa_create_from_arena (arena_dev != nullptr:
// adopt a prepared weight arena of `src_bytes` bytes) share everything but the source of the weights.
int create_impl(const pa_config* cfg, const void* blob, size_t src_bytes, const void* arena_dev, pa_engine** out) {
    if (!cfg || (!blob && !arena_dev) || !out) return PA_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->abi_version != PA_ABI_VERSION) return PA_ERR_INVALID_ARG;
    const int S = cfg->sequence_length, A = cfg->num_actions, F = cfg->num_fighters;
    if (S < 1 || S % 2 == 0 || S > 15 || A < 1 || A > 64 || F < 1 || F > 4 || cfg->max_batch_frames < 1 ||
        cfg->max_clip_frames < 1 || cfg->max_frame_height < 1 || cfg->max_frame_width < 1 || cfg->crop_padding < 0 ||
        (cfg->compute_dtype != PA_DTYPE_F32 && cfg->compute_dtype != PA_DTYPE_BF16 && cfg->compute_dtype != PA_DTYPE_EMULATED_F32))
        return PA_ERR_INVALID_ARG;
    const int32_t* hdr = reinterpret_cast<const int32_t*>(blob);
    if (blob && (src_bytes != pa_weight_blob_bytes(S, A) || hdr[0] != PA_WEIGHT_MAGIC || hdr[1] != 1 || hdr[2] != S || hdr[3] != A))
        return PA_ERR_BAD_WEIGHTS;
    pa_engine* e = new pa_engine();
    e->cfg = *cfg;
    e->bf16 = cfg->compute_dtype == PA_DTYPE_BF16;
    e->emu = cfg->compute_dtype == PA_DTYPE_EMULATED_F32;
    e->adopt = blob == nullptr;
    {
        const char* pl = getenv("PA_PROFILE_LAYERS");
        e->profile_layers = pl && pl[0] == '1';
    }
    *out = e;  // handed back even on failure so the caller can read pa_last_error, then pa_destroy
    {
        int ndev = 0;
        const hipError_t derr = hipGetDeviceCount(&ndev);
        if (derr != hipSuccess || ndev <= 0 || cfg->device_id < 0 || cfg->device_id >= ndev)
            return fail(e, PA_ERR_NO_DEVICE,
                        std::string("hipGetDeviceCount: ") + hipGetErrorString(derr) + ", devices=" + std::to_string(ndev) +
                            ", requested device " + std::to_string(cfg->device_id));
    }
    HIPCHK(e, hipSetDevice(cfg->device_id));
    HIPCHK(e, preprocess_init_device());
    const int NC = cfg->max_batch_frames * F;
    e->max_crops = NC;
    e->cache_rows = cfg->max_clip_frames * F;
    int rc;
#define ALLOC(ptr, count, zero)                         \
    do {                                                \
        rc = dev_alloc(e, &(ptr), (size_t)(count), zero); \
        if (rc != PA_OK) return rc;                     \
    } while (0)

    ALLOC(e->x0_slot[0], (size_t)NC * 134 * 134 * 4, true);
    ALLOC(e->x0_slot[1], (size_t)NC * 134 * 134 * 4, true);
    e->x0 = e->x0_slot[0];
    ALLOC(e->savebox_rects, (size_t)NC * 4, true);
    ALLOC(e->pre_status[0], (size_t)NC, true);
    ALLOC(e->pre_status[1], (size_t)NC, true);
    ALLOC(e->c1, (size_t)NC * 66 * 66 * 64, true);
    ALLOC(e->p1, (size_t)NC * 34 * 34 * 64, true);
    ALLOC(e->pooled, (size_t)NC * 512, true);
    ALLOC(e->feats_tmp, (size_t)NC * PA_FEATURE_STRIDE, true);
    ALLOC(e->jpeg_qtab, 128, true);
    ALLOC(e->crops_tmp, (size_t)NC * PA_CROP * PA_CROP * 3, true);
    ALLOC(e->cache, (size_t)e->cache_rows * PA_FEATURE_STRIDE, true);
    ALLOC(e->cache_status, (size_t)e->cache_rows, true);
    ALLOC(e->h1, (size_t)NC * 512, true);
    ALLOC(e->gather, (size_t)NC * S, true);
    ALLOC(e->status_tmp, (size_t)NC, true);
    ALLOC(e->dev_errors, 4, true);
    // scratch of pa_clean_detections, sized for the longest clip and the widest detection table (max_det <= 8): the enqueue-only
    // call never allocates for a table the engine was made for (a longer one still grows it, behind a device synchronisation)
    e->clean_g6_cap = (size_t)cfg->max_clip_frames * 8 * 6;
    ALLOC(e->clean_g6, e->clean_g6_cap, false);

    // ---- layer table + weights --------------------------------------------
    // arena capacity: the folded tensors are the blob's plus zero padding (stem K 147 -> 224, fc and
    // Conv1d 1000 -> 1024) and 256-byte alignment per tensor; 1.25x + 1 MiB covers every (S, A)
    // (+ 16 / 9 of the 3x3 filters once more: the Winograd layout of the fp32 stride-1 3x3 layers, kept beside the direct one)
    e->arena_cap = pa_weight_blob_bytes(S, A) / 4 * 13 + (1u << 20);
    {
        void* ar = nullptr;
        HIPCHK(e, hipMalloc(&ar, e->arena_cap));
        e->allocs.push_back(ar);
        e->arena = reinterpret_cast<uint8_t*>(ar);
    }
    if (!e->adopt) e->arena_stage.assign(e->arena_cap, 0);
    static const float dry_source[1] = {0.f};  // adopt mode: take() only has to return non-null
    const float* blob_f = e->adopt ? dry_source : reinterpret_cast<const float*>(hdr + 8);
    BlobReader br{blob_f, 0, blob_float_count(S, A), e->adopt};
    std::vector<float> keep_w, keep_b;  // host copy of the last conv added with keep=true (not uploaded yet)
    auto add_conv = [&](const std::string& name, int cin, int cout, int k, int stride, int in_hw, float* in, float* outb,
                        float* residual, int relu, bool keep = false) -> int {
        ConvLayer L;
        L.name = name;
        L.cin = cin;
        L.cout = cout;
        L.kh = L.kw = k;
        L.stride = stride;
        L.in_hw = in_hw;
        L.out_hw = in_hw / stride;
        L.in = in;
        L.out = outb;
        L.residual = residual;
        L.relu = relu;
        if (k == 7) {
            L.in_pad = 3; L.out_pad = 1; L.off = 0;
            L.taps = 7; L.kw_taps = 1; L.chunk = 32; L.in_px_stride = 4;
        } else if (k == 3) {
            L.in_pad = 1; L.out_pad = 1; L.off = 0;
            L.taps = 9; L.kw_taps = 3; L.chunk = cin; L.in_px_stride = cin;
        } else {
            L.in_pad = 1; L.out_pad = 1; L.off = 1;
            L.taps = 1; L.kw_taps = 1; L.chunk = cin; L.in_px_stride = cin;
        }
        L.k_alg = (double)cin * k * k;
        std::vector<float> w, b;
        if (!fold_conv(br, L, w, b)) return fail(e, PA_ERR_BAD_WEIGHTS, "weight blob too short at " + name);
        int r2 = PA_OK;
        if (keep) {
            keep_w.swap(w);
            keep_b.swap(b);
        } else {
            r2 = (e->bf16 && k == 3) ? upload_bf16(e, &L.wgt, w) : upload(e, &L.wgt, w);
            if (r2) return r2;
            // Winograd F(2x2, 3x3) for the fp32 stride-1 3x3 layers (wino.hip; measured per layer at 128 crops,
            // profiles/r05_resnet_layer_times_wino_options.txt). Layer 4's 4 x 4 maps -- 64 tiles at 128 crops -- were slower there
            // than on the direct kernel (117-160 against 84 us) until the kernel learnt to split K. PA_WINO=0: none (A/B);
            // PA_WINO_MIN_HW=8: layers 1-3 only (A/B)
            static const int wino_mode = getenv("PA_WINO") ? atoi(getenv("PA_WINO")) : 1;
            static const int wino_min_hw = getenv("PA_WINO_MIN_HW") ? atoi(getenv("PA_WINO_MIN_HW")) : 4;
            if (!e->bf16 && k == 3 && stride == 1 && wino_mode && L.out_hw >= wino_min_hw && L.out_hw % 4 == 0) {
                L.wino_bn = wino_pick_bn(cout, (long long)NC * (L.out_hw / 4) * (L.out_hw / 4), cin);
                std::vector<float> ug(wino_weight_floats(cin, cout), 0.f);
                if (!br.dry) wino_transform_weights(w.data(), cin, cout, L.wino_bn, ug.data());
                r2 = upload(e, &L.wino_wgt, ug);
                if (r2) return r2;
            }
            // PA_DTYPE_EMULATED_F32: the stride-2 3x3 openers of layers 2-4 on psgemm.hip (the stride-1 3x3 layers keep their exact
            // Winograd kernel: per layer the two measure the same, profiles/r06_pgemm_split_layers.txt)
            if (e->emu && k == 3 && stride == 2 && cin % 32 == 0 && (r2 = upload_split(e, &L.split_wgt, w, cout, 9 * cin, false))) return r2;
            if (e->bf16 && k == 7 && (r2 = upload_bf16(e, &e->stem_wgt_bf16, w))) return r2;
            r2 = upload(e, &L.bias, b);
            if (r2) return r2;
        }
        choose_tile(NC * L.out_hw * L.out_hw, cout, L.taps * L.chunk / ((e->bf16 && k == 3) ? 64 : 32), &L.tile, &L.splitk);
        // tuning knobs (scripts/tune_tiles.py): PA_FORCE_TILE=0|1|2, PA_FORCE_SPLITK=n
        if (const char* ft = getenv("PA_FORCE_TILE")) {
            const int t = atoi(ft);
            if (t >= 0 && t <= 4 && !(t == 0 && cout % 128 != 0)) { L.tile = (GemmTile)t; L.forced = true; }
        }
        if (const char* fs = getenv("PA_FORCE_SPLITK")) { L.splitk = std::max(1, atoi(fs)); L.forced = true; }
        e->convs.push_back(L);
        return PA_OK;
    };

    rc = add_conv("conv1", 3, 64, 7, 2, 128, e->x0, e->c1, nullptr, 1);
    if (rc) return rc;
    {
        const int widths[4] = {64, 128, 256, 512};
        const int hw_in[4] = {32, 32, 16, 8};
        float* cur = e->p1;
        int cin = 64;
        for (int li = 0; li < 4; ++li) {
            const int co = widths[li];
            const int stride = li == 0 ? 1 : 2;
            const int hw_out = hw_in[li] / stride;
            const size_t buf = (size_t)NC * (hw_out + 2) * (hw_out + 2) * co;
            float *mid, *outA, *outB;
            ALLOC(mid, buf, true);
            ALLOC(outA, buf, true);
            ALLOC(outB, buf, true);
            const std::string pre = "layer" + std::to_string(li + 1);
            // block 0 (blob order: conv1, bn1, conv2, bn2, downsample)
            rc = add_conv(pre + ".0.conv1", cin, co, 3, stride, hw_in[li], cur, mid, nullptr, 1);
            if (rc) return rc;
            if (li == 0) {
                rc = add_conv(pre + ".0.conv2", co, co, 3, 1, hw_out, mid, outA, cur, 1);
                if (rc) return rc;
            } else {
                // conv2 and the 1x1/2 downsample of the block input are ONE implicit GEMM:
                //   out = relu( [W2' | Wd'] . [im2col3x3(mid) ; x(2y,2x)] + (b2' + bd') )
                // (K = 9*co + cin): no separate downsample launch, no residual tensor.
                rc = add_conv(pre + ".0.conv2", co, co, 3, 1, hw_out, mid, outA, nullptr, 1, /*keep=*/true);
                if (rc) return rc;
                ConvLayer D;  // only used to fold the downsample weights
                D.name = pre + ".0.downsample";
                D.cin = cin; D.cout = co; D.kh = D.kw = 1; D.taps = 1; D.kw_taps = 1; D.chunk = cin;
                std::vector<float> wd, bd;
                if (!fold_conv(br, D, wd, bd)) return fail(e, PA_ERR_BAD_WEIGHTS, "weight blob too short at " + D.name);
                ConvLayer& L = e->convs.back();
                const int k_main = 9 * co, k_all = k_main + cin;
                std::vector<float> wf((size_t)co * k_all), bf(co);
                for (int o = 0; o < co; ++o) {
                    memcpy(&wf[(size_t)o * k_all], &keep_w[(size_t)o * k_main], sizeof(float) * k_main);
                    memcpy(&wf[(size_t)o * k_all + k_main], &wd[(size_t)o * cin], sizeof(float) * cin);
                    bf[o] = (float)((double)keep_b[o] + (double)bd[o]);
                }
                if (e->bf16) {
                    rc = upload_bf16(e, &L.wgt, keep_w);  // [co][9 co]: the 3x3 alone
                    if (rc) return rc;
                    rc = upload_bf16(e, &L.ds_wgt, wd);   // [co][cin]
                    if (rc) return rc;
                    ALLOC(L.ds_out, buf, true);
                    // PA_BF16_DS_FUSE=0: the branch as its own GEMM (round 2-3), for A/B runs
                    static const int ds_fuse = getenv("PA_BF16_DS_FUSE") ? atoi(getenv("PA_BF16_DS_FUSE")) : 1;
                    ConvLayer& C1 = e->convs[e->convs.size() - 2];  // the block's stride-2 opener: same input, same output shape
                    if (ds_fuse && C1.in == cur && C1.stride == 2 && C1.cout == co && C1.out_hw == hw_out && C1.out_pad == L.out_pad &&
                        C1.chunk == cin && C1.chunk % 64 == 0 && !C1.forced) {
                        if (ds_fuse == 2) {
                            C1.ds_probe = true;  // (A/B: same opener launch, branch as its own GEMM -> bit-identical results)
                        } else {
                            C1.ds_next_wgt = L.ds_wgt;
                            C1.ds_next_out = L.ds_out;
                            L.ds_fused = true;
                        }
                    }
                } else {
                    rc = upload(e, &L.wgt, wf);
                    if (rc) return rc;
                    // the same layer in Winograd form: the 3x3 alone on wino.hip, the 1x1/2 branch as a small GEMM of its own whose
                    // result the 3x3 adds as its residual (PA_WINO >= 1 and the map sizes the kernel is faster on; else the fused
                    // implicit GEMM above)
                    static const int wino_mode = getenv("PA_WINO") ? atoi(getenv("PA_WINO")) : 1;
                    static const int wino_ds = getenv("PA_WINO_DS") ? atoi(getenv("PA_WINO_DS")) : 1;
                    static const int wino_min_hw = getenv("PA_WINO_MIN_HW") ? atoi(getenv("PA_WINO_MIN_HW")) : 4;
                    if (wino_mode && wino_ds && hw_out >= wino_min_hw && hw_out % 4 == 0) {
                        L.wino_bn = wino_pick_bn(co, (long long)NC * (hw_out / 4) * (hw_out / 4), co);
                        std::vector<float> ug(wino_weight_floats(co, co), 0.f);
                        if (!br.dry) wino_transform_weights(keep_w.data(), co, co, L.wino_bn, ug.data());
                        rc = upload(e, &L.wino_wgt, ug);
                        if (rc) return rc;
                        rc = upload(e, &L.ds_wgt, wd);   // [co][cin] fp32
                        if (rc) return rc;
                        if (e->emu && (rc = upload_split(e, &L.ds_split_wgt, wd, co, cin, false))) return rc;
                        ALLOC(L.ds_out, buf, true);
                    }
                }
                rc = upload(e, &L.bias, bf);
                if (rc) return rc;
                L.in2 = cur;
                L.in2_c = cin;
                L.in2_hw = hw_in[li];
                L.in2_stride = stride;
                L.k_alg += cin;
            }
            rc = add_conv(pre + ".1.conv1", co, co, 3, 1, hw_out, outA, mid, nullptr, 1);
            if (rc) return rc;
            rc = add_conv(pre + ".1.conv2", co, co, 3, 1, hw_out, mid, outB, outA, 1);
            if (rc) return rc;
            cur = outB;
            cin = co;
        }
        e->layer4_out = cur;
    }
    {  // fc 512 -> 1000 (N padded to 1024 with zero rows so cached rows are 1024 wide)
        const float* w = br.take((size_t)1000 * 512);
        const float* b = br.take(1000);
        if (!w || !b) return fail(e, PA_ERR_BAD_WEIGHTS, "weight blob too short at fc");
        std::vector<float> wp((size_t)PA_FEATURE_STRIDE * 512, 0.f), bp(PA_FEATURE_STRIDE, 0.f);
        if (!br.dry) {
            memcpy(wp.data(), w, sizeof(float) * 1000 * 512);
            memcpy(bp.data(), b, sizeof(float) * 1000);
        }
        ConvLayer& L = e->fc;
        L.name = "fc";
        L.cin = 512; L.cout = PA_FEATURE_STRIDE; L.kh = L.kw = 1; L.stride = 1;
        L.in_hw = 1; L.out_hw = 1; L.in_pad = 0; L.out_pad = 0; L.off = 0;
        L.taps = 1; L.kw_taps = 1; L.chunk = 512; L.in_px_stride = 512;
        L.in = e->pooled; L.out = nullptr; L.residual = nullptr; L.relu = 0;
        L.k_alg = 512.0 * 1000.0 / 1024.0;
        rc = upload(e, &L.wgt, wp);
        if (rc) return rc;
        rc = upload(e, &L.bias, bp);
        if (rc) return rc;
        // 32 tiles of 64x64 only: split the 16 k-steps four ways (measured 16.9 / 14.5 / 12.3 us for 1 / 2 / 4)
        L.tile = TILE_64x64;
        L.splitk = 4;
        L.forced = true;
        if (const char* fs = getenv("PA_FC_SPLITK")) L.splitk = std::max(1, atoi(fs));
    }
    {  // Conv1d(1000 -> 512, k=S): [512][1000][S] -> [512][S][1024]
        const float* w = br.take((size_t)512 * 1000 * S);
        const float* b = br.take(512);
        if (!w || !b) return fail(e, PA_ERR_BAD_WEIGHTS, "weight blob too short at cnn1d");
        std::vector<float> wp((size_t)512 * S * PA_FEATURE_STRIDE, 0.f);
        for (int o = 0; o < 512 && !br.dry; ++o)
            for (int c = 0; c < 1000; ++c)
                for (int t = 0; t < S; ++t)
                    wp[((size_t)o * S + t) * PA_FEATURE_STRIDE + c] = w[((size_t)o * 1000 + c) * S + t];
        rc = upload(e, &e->w1d, wp);
        if (rc) return rc;
        rc = upload(e, &e->b1d, br.dry ? std::vector<float>(512) : std::vector<float>(b, b + 512));
        if (rc) return rc;
        choose_tile(NC, 512, S * 32, &e->head_tile, &e->head_splitk);
        // 16 tiles x 224 k-steps: 16 splits of 14 steps beat 32 of 7 (26.2 vs 28.6 us incl. the reduce)
        if (e->head_splitk > 16) e->head_splitk = 16;
        if (const char* hs = getenv("PA_HEAD_SPLITK")) e->head_splitk = std::max(1, atoi(hs));
    }
    {
        const float* w2 = br.take(128 * 512);
        const float* b2 = br.take(128);
        const float* w3 = br.take((size_t)A * 128);
        const float* b3 = br.take(A);
        if (!w2 || !b2 || !w3 || !b3 || br.pos != br.limit) return fail(e, PA_ERR_BAD_WEIGHTS, "weight blob size mismatch at classifier");
        // head_mlp_kernel reads both matrices k-major (lanes = outputs): w2 -> [512][128], w3 -> [128][64]
        std::vector<float> w2t((size_t)512 * 128), w3t((size_t)128 * 64, 0.f);
        for (int o = 0; o < 128 && !br.dry; ++o)
            for (int k = 0; k < 512; ++k) w2t[(size_t)k * 128 + o] = w2[(size_t)o * 512 + k];
        for (int a = 0; a < A && !br.dry; ++a)
            for (int k = 0; k < 128; ++k) w3t[(size_t)k * 64 + a] = w3[(size_t)a * 128 + k];
        if ((rc = upload(e, &e->w2, w2t))) return rc;
        if ((rc = upload(e, &e->b2, br.dry ? std::vector<float>(128) : std::vector<float>(b2, b2 + 128)))) return rc;
        if ((rc = upload(e, &e->w3, w3t))) return rc;
        if ((rc = upload(e, &e->b3, br.dry ? std::vector<float>(A) : std::vector<float>(b3, b3 + A)))) return rc;
    }
    // the arena is complete: one host-to-device copy (fold mode) or one device copy (adopt mode)
    if (e->adopt) {
        if (src_bytes != e->arena_used) return fail(e, PA_ERR_BAD_WEIGHTS, "weight arena size does not match this configuration");
        HIPCHK(e, hipMemcpy(e->arena, arena_dev, e->arena_used, hipMemcpyDeviceToDevice));
    } else {
        HIPCHK(e, hipMemcpy(e->arena, e->arena_stage.data(), e->arena_used, hipMemcpyHostToDevice));
        std::vector<uint8_t>().swap(e->arena_stage);
    }
    // split-K slabs: the largest splitk*M*N over all layers
    {
        size_t need = 0;
        for (const ConvLayer& L : e->convs)
            if (L.splitk > 1) need = std::max(need, (size_t)L.splitk * NC * L.out_hw * L.out_hw * L.cout);
        if (e->fc.splitk > 1) need = std::max(need, (size_t)e->fc.splitk * NC * PA_FEATURE_STRIDE);
        if (e->head_splitk > 1) need = std::max(need, (size_t)e->head_splitk * NC * 512);
        // the Winograd kernel's split-K scratch (wino.hip): at most one launch's worth of partial output tiles -- a grid of 256
        // eight-wave (512 four-wave) workgroups x 32 floats per thread -- and a ticket per tile
        bool any_wino = false;
        for (const ConvLayer& L : e->convs) any_wino = any_wino || L.wino_wgt;
        if (any_wino) {
            need = std::max(need, (size_t)256 * 512 * 32);
            ALLOC(e->wino_tickets, (size_t)2 * WINO_TICKETS, true);
        }
        need *= 2;  // one region per interleaved half batch
        e->slab_floats = need;
        ALLOC(e->slab, need, false);
    }
    // preprocess scratch
    e->coef_dim = std::max(cfg->max_frame_height, cfg->max_frame_width);
    // per-crop scratch of the resamplers: a whole frame slice, and at least the runner-input branch's worst
    // case (INTER_AREA result of 448 rows + the 128 rows behind the vertical pass, 128 px x 3 B each)
    e->t_stride = std::max((size_t)cfg->max_frame_height * cfg->max_frame_width * 3, (size_t)(448 + 128) * PA_CROP * 3);
    ALLOC(e->plans, (size_t)NC, true);
    ALLOC(e->fallback, (size_t)NC + 4, true);
    ALLOC(e->coef, (size_t)NC * 2 * e->coef_dim * (2 + PA_KSIZE_MAX), false);
    // a crop side cannot exceed the shorter frame side without being clipped (and then it is not a cached pair); 1024 caps the
    // cache at 36 MB
    e->coef_cache_dmax = std::min(std::min(cfg->max_frame_height, cfg->max_frame_width), 1024);
    ALLOC(e->coef_cache, coef_cache_ints(e->coef_cache_dmax), false);
    e->coef_cache_pad = 30;  // the one padding the reference's runner passes (ai_runner.py:418)
    HIPCHK(e, launch_build_coef_cache(e->coef_cache, e->coef_cache_pad, e->coef_cache_dmax, nullptr));
    ALLOC(e->area_tabs, (size_t)NC * 2 * PA_CROP, true);
    ALLOC(e->t1, (size_t)NC * e->t_stride, false);
    ALLOC(e->t2, (size_t)NC * e->t_stride, false);
#undef ALLOC
    if (const char* il = getenv("PA_INTERLEAVE")) e->interleave = atoi(il);
    HIPCHK(e, hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
    HIPCHK(e, hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    HIPCHK(e, hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    HIPCHK(e, hipEventCreateWithFlags(&e->ev_ds_fork, hipEventDisableTiming));
    HIPCHK(e, hipEventCreateWithFlags(&e->ev_ds_join, hipEventDisableTiming));
    HIPCHK(e, hipDeviceSynchronize());
    return PA_OK;
}

}  // namespace

extern "C" {

int pa_create(const pa_config* cfg, const void* blob, size_t blob_bytes, pa_engine** out) {
    if (!blob) return PA_ERR_INVALID_ARG;
    return create_impl(cfg, blob, blob_bytes, nullptr, out);
}

int pa_create_from_arena(const pa_config* cfg, const void* arena_dev, size_t arena_bytes, pa_engine** out) {
    if (!arena_dev) return PA_ERR_INVALID_ARG;
    return create_impl(cfg, nullptr, arena_bytes, arena_dev, out);
}

size_t pa_weights_arena_bytes(const pa_engine* e) { return e ? e->arena_used : 0; }

int pa_weights_export(pa_engine* e, void* dst_dev, size_t bytes, void* stream) {
    if (!e || !dst_dev || bytes != e->arena_used) return fail(e, PA_ERR_INVALID_ARG, "pa_weights_export: size must be pa_weights_arena_bytes()");
    HIPCHK(e, hipMemcpyAsync(dst_dev, e->arena, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PA_OK;
}

void pa_destroy(pa_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->cfg.device_id);
    (void)hipDeviceSynchronize();
    for (ProfEntry& p : e->prof_log) {
        (void)hipEventDestroy(p.start);
        (void)hipEventDestroy(p.stop);
    }
    for (hipEvent_t ev : e->event_pool) (void)hipEventDestroy(ev);
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    if (e->ev_ds_fork) (void)hipEventDestroy(e->ev_ds_fork);
    if (e->ev_ds_join) (void)hipEventDestroy(e->ev_ds_join);
    if (e->side) (void)hipStreamDestroy(e->side);
    if (e->gate_flag) (void)hipHostFree(e->gate_flag);
    for (void* p : e->allocs) (void)hipFree(p);
    delete e;
}

int pa_infer_windows(pa_engine* e, const float* x, int32_t batch, float* logp, void* stream) {
    if (!e || !x || !logp || batch < 1) return fail(e, PA_ERR_INVALID_ARG, "pa_infer_windows: bad argument");
    const int S = e->cfg.sequence_length;
    hipStream_t s = (hipStream_t)stream;
    // the backbone scratch holds max_crops crops: run the windows in groups
    const int win_per_pass = e->max_crops / S;
    if (win_per_pass < 1) return fail(e, PA_ERR_CAPACITY, "pa_infer_windows: engine smaller than one window");
    for (int w0 = 0; w0 < batch; w0 += win_per_pass) {
        const int nw = std::min(win_per_pass, batch - w0);
        const int ncrops = nw * S;
        {
            ProfScope ps(e, s, "nchw_to_nhwc4", 0.0, ncrops * (49152.0 * 4 + 134.0 * 134 * 16));
            HIPCHK(e, launch_nchw_to_padded(x + (size_t)w0 * S * 3 * 128 * 128, e->x0, ncrops, e->bf16 ? 1 : 0, s));
        }
        int rc = run_backbone(e, ncrops, e->x0, e->feats_tmp, s);
        if (rc) return rc;
        HIPCHK(e, launch_identity_gather(e->gather, ncrops, s));
        e->gather_key[0] = -1;  // the clip head's cached window table is gone
        rc = run_head(e, nw, e->feats_tmp, e->gather, nullptr, nullptr, logp + (size_t)w0 * e->cfg.num_actions, s);
        if (rc) return rc;
    }
    return PA_OK;
}

int pa_backbone_windows(pa_engine* e, const float* x, int32_t n_crops, float* feats, void* stream) {
    if (!e || !x || !feats || n_crops < 1) return fail(e, PA_ERR_INVALID_ARG, "pa_backbone_windows: bad argument");
    hipStream_t s = (hipStream_t)stream;
    for (int c0 = 0; c0 < n_crops; c0 += e->max_crops) {  // the backbone scratch holds max_crops crops
        const int n = std::min(e->max_crops, n_crops - c0);
        {
            ProfScope ps(e, s, "nchw_to_nhwc4", 0.0, n * (49152.0 * 4 + 134.0 * 134 * 16));
            HIPCHK(e, launch_nchw_to_padded(x + (size_t)c0 * 3 * 128 * 128, e->x0, n, e->bf16 ? 1 : 0, s));
        }
        const int rc = run_backbone(e, n, e->x0, feats + (size_t)c0 * PA_FEATURE_STRIDE, s);
        if (rc) return rc;
    }
    return PA_OK;
}

int pa_crop_resize_width(pa_engine* e, const uint8_t* frames, int32_t n, int32_t height, int32_t width, const int32_t* rects_host,
                         int32_t n_rects, int32_t out_w, uint8_t* out, int32_t out_h_cap, int32_t* out_h_host, void* stream) {
    if (!e || !frames || !rects_host || !out || !out_h_host || n < 1 || height < 1 || width < 1 || n_rects < 1 || n_rects > 4 || out_w < 1 ||
        out_h_cap < 1)
        return fail(e, PA_ERR_INVALID_ARG, "pa_crop_resize_width: bad argument");
    RectResizeParams q;
    memset(&q, 0, sizeof(q));
    q.frames = frames;
    q.height = height;
    q.width = width;
    q.n_rects = n_rects;
    q.out_w = out_w;
    q.out_h_cap = out_h_cap;
    q.out = out;
    for (int r = 0; r < n_rects; ++r) {
        const int x1 = rects_host[4 * r], y1 = rects_host[4 * r + 1], x2 = rects_host[4 * r + 2], y2 = rects_host[4 * r + 3];
        if (x1 < 0 || y1 < 0 || x2 > width || y2 > height || x2 <= x1 || y2 <= y1)
            return fail(e, PA_ERR_INVALID_ARG, "pa_crop_resize_width: empty rectangle or outside the frame (cv2.resize raises on an empty image)");
        // imutils.resize: r = width / float(w); dim = (width, int(h * r))
        const double ratio = (double)out_w / (double)(x2 - x1);
        const int oh = (int)((double)(y2 - y1) * ratio);
        if (oh < 1) return fail(e, PA_ERR_INVALID_ARG, "pa_crop_resize_width: destination height 0");
        if (oh > out_h_cap) return fail(e, PA_ERR_CAPACITY, "pa_crop_resize_width: out_h_cap too small");
        q.x1[r] = x1; q.y1[r] = y1; q.x2[r] = x2; q.y2[r] = y2;
        q.out_h[r] = oh;
        out_h_host[r] = oh;
    }
    HIPCHK(e, launch_rect_resize(q, n, (hipStream_t)stream));
    return PA_OK;
}

int pa_square_crops(pa_engine* e, const uint8_t* frames, int32_t n, int32_t height, int32_t width, const double* boxes,
                    int32_t padding, int32_t swap_rb, uint8_t* crops, int32_t* status, void* stream) {
    if (!e || !frames || !boxes || !crops || n < 1 || height < 1 || width < 1 || padding < 0)
        return fail(e, PA_ERR_INVALID_ARG, "pa_square_crops: bad argument");
    if (n > e->cfg.max_batch_frames || height > e->cfg.max_frame_height || width > e->cfg.max_frame_width)
        return fail(e, PA_ERR_CAPACITY, "pa_square_crops: frames exceed engine capacity");
    return run_preprocess(e, frames, n, height, width, boxes, padding, swap_rb, crops, nullptr, status, (hipStream_t)stream);
}

namespace {
int run_runner_inputs(pa_engine* e, const uint8_t* images, size_t images_bytes, const pa_crop_image* desc, int n, int swap_rb,
                      uint8_t* inputs_u8, float* inputs_f32, int32_t* status, hipStream_t s) {
    RunnerInParams q;
    memset(&q, 0, sizeof(q));
    q.images = images;
    q.images_bytes = (long long)images_bytes;
    q.desc = reinterpret_cast<const CropImageDesc*>(desc);
    q.n = n;
    q.swap_rb = swap_rb;
    q.max_h = e->cfg.max_frame_height;
    q.max_w = e->cfg.max_frame_width;
    q.t1 = e->t1;
    q.t2 = e->t2;
    q.t_stride = e->t_stride;
    q.inputs_u8 = inputs_u8;
    q.inputs_f32 = inputs_f32;
    q.inputs_f32_is_bf16 = e->bf16 ? 1 : 0;
    q.status = status;
    ProfScope ps(e, s, "runner_inputs", 0.0, (double)images_bytes + (double)n * 49152.0 * 5);
    HIPCHK(e, launch_runner_inputs(q, s));
    return PA_OK;
}
}  // namespace

int pa_runner_inputs(pa_engine* e, const uint8_t* images, size_t images_bytes, const pa_crop_image* desc, int32_t n_crops,
                     int32_t swap_rb, uint8_t* inputs_u8, int32_t* status, void* stream) {
    static_assert(sizeof(pa_crop_image) == sizeof(CropImageDesc), "descriptor layouts must agree");
    if (!e || !images || !desc || !inputs_u8 || n_crops < 1) return fail(e, PA_ERR_INVALID_ARG, "pa_runner_inputs: bad argument");
    if (n_crops > e->max_crops) return fail(e, PA_ERR_CAPACITY, "pa_runner_inputs: more crops than max_batch_frames * num_fighters");
    return run_runner_inputs(e, images, images_bytes, desc, n_crops, swap_rb, inputs_u8, nullptr, status, (hipStream_t)stream);
}

int pa_backbone_crop_images(pa_engine* e, const uint8_t* images, size_t images_bytes, const pa_crop_image* desc, int32_t n,
                            int32_t frame0, uint8_t* crops_rgb, int32_t* status, void* stream) {
    if (!e || !images || !desc || n < 1 || frame0 < 0) return fail(e, PA_ERR_INVALID_ARG, "pa_backbone_crop_images: bad argument");
    if (e->clip_frames < 1) return fail(e, PA_ERR_INVALID_ARG, "pa_backbone_crop_images: call pa_clip_begin first");
    if (n > e->cfg.max_batch_frames || frame0 + n > e->clip_frames)
        return fail(e, PA_ERR_CAPACITY, "pa_backbone_crop_images: frames exceed engine / clip capacity");
    hipStream_t s = (hipStream_t)stream;
    const int F = e->cfg.num_fighters;
    int rc = run_runner_inputs(e, images, images_bytes, desc, n * F, 1, crops_rgb, e->x0_slot[0], e->pre_status[0], s);
    if (rc) return rc;
    if (status) HIPCHK(e, hipMemcpyAsync(status, e->pre_status[0], sizeof(int32_t) * n * F, hipMemcpyDeviceToDevice, s));
    return pa_backbone_slot(e, 0, n, frame0, stream);
}

namespace {
// The slice of YoloCrop.square_crop (fighter.py:305-343) on the host: the arithmetic of crop_plan_kernel up to the
// numpy slice, so that the window the host uploads is exactly the region the device plan will read.
void host_slice(const double* b, int W, int H, int pad, int* sy0, int* sx0, int* sh, int* sw) {
    *sy0 = *sx0 = *sh = *sw = 0;
    const double v[4] = {b[0] * W, b[1] * H, b[2] * W, b[3] * H};
    int iv[4];
    for (int k = 0; k < 4; ++k) {
        if (!(v[k] > -2.0e9 && v[k] < 2.0e9)) return;
        iv[k] = (int)v[k];
    }
    const int cx = iv[0], cy = iv[1], d = iv[2] > iv[3] ? iv[2] : iv[3];
    if (d <= 0 || d > 16384) return;
    const int half = d / 2;
    auto np_slice = [](int start, int stop, int size, int* s0, int* len) {
        if (start > size) start = size;
        if (stop < 0) {
            stop += size;
            if (stop < 0) stop = 0;
        }
        if (stop > size) stop = size;
        *s0 = start;
        *len = stop > start ? stop - start : 0;
    };
    np_slice(std::max(cy - half - pad, 0), std::min(cy + half + pad, H), H, sy0, sh);
    np_slice(std::max(cx - half - pad, 0), std::min(cx + half + pad, W), W, sx0, sw);
}
}  // namespace

int pa_upload_crop_windows(pa_engine* e, const uint8_t* frames_host, int32_t n, int32_t height, int32_t width,
                           const double* boxes_host, int32_t padding, uint8_t* windows_dev, size_t windows_capacity,
                           pa_crop_window* desc_host, pa_crop_window* desc_dev, size_t* bytes_used, void* stream) {
    static_assert(sizeof(pa_crop_window) == sizeof(CropWindow), "descriptor layouts must agree");
    if (!e || !frames_host || !boxes_host || !windows_dev || !desc_host || !desc_dev || n < 1 || height < 1 || width < 1 || padding < 0)
        return fail(e, PA_ERR_INVALID_ARG, "pa_upload_crop_windows: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int F = e->cfg.num_fighters;
    // the upload kernel dereferences frames_host on the device: it must be pinned (device-visible) host memory,
    // a pageable pointer would be a GPU page fault
    for (const void* hp : {(const void*)frames_host, (const void*)(frames_host + (size_t)n * height * width * 3 - 1), (const void*)desc_host}) {
        hipPointerAttribute_t attr;
        const hipError_t pe = hipPointerGetAttributes(&attr, hp);
        if (pe != hipSuccess || (attr.type != hipMemoryTypeHost && attr.type != hipMemoryTypeManaged && attr.type != hipMemoryTypeDevice)) {
            (void)hipGetLastError();
            return fail(e, PA_ERR_INVALID_ARG, "pa_upload_crop_windows: frames_host / desc_host must be pinned host memory (hipHostMalloc, torch pin_memory)");
        }
    }
    size_t off = 0;
    for (int i = 0; i < n * F; ++i) {
        int sy0, sx0, sh, sw;
        host_slice(boxes_host + (size_t)i * 4, width, height, padding, &sy0, &sx0, &sh, &sw);
        pa_crop_window& d = desc_host[i];
        d.offset = (int64_t)off;
        d.row_bytes = sw * 3;
        d.pitch = (sw * 3 + 15) & ~15;
        d.rows = sh;
        d.src_offset = (int64_t)((((size_t)(i / F) * height + sy0) * width + sx0) * 3);
        d.src_pitch = width * 3;
        const size_t bytes = (size_t)sh * d.pitch;
        if (sh == 0 || sw == 0) {  // empty / bad box: the device plan reports it and reads nothing
            d.rows = d.row_bytes = d.pitch = 0;
            continue;
        }
        if (off + bytes + 16 > windows_capacity) return fail(e, PA_ERR_CAPACITY, "pa_upload_crop_windows: window buffer too small");
        off += bytes;
    }
    // descriptors first (tiny, from the caller's pinned array), then ONE kernel that pulls every slice row out of
    // the pinned host frames over PCIe
    HIPCHK(e, hipMemcpyAsync(desc_dev, desc_host, sizeof(pa_crop_window) * n * F, hipMemcpyHostToDevice, s));
    HIPCHK(e, launch_slice_upload(frames_host, (long long)n * height * width * 3, reinterpret_cast<const CropWindow*>(desc_dev), windows_dev,
                                  n * F, s));
    if (bytes_used) *bytes_used = off;
    return PA_OK;
}

int pa_preprocess_windows(pa_engine* e, const uint8_t* windows_dev, const pa_crop_window* desc_dev, int32_t n, int32_t height,
                          int32_t width, const double* boxes, int32_t slot, uint8_t* crops_rgb, int32_t* status, void* stream) {
    if (!e || !windows_dev || !desc_dev || !boxes || n < 1 || height < 1 || width < 1 || slot < 0 || slot > 1)
        return fail(e, PA_ERR_INVALID_ARG, "pa_preprocess_windows: bad argument");
    if (n > e->cfg.max_batch_frames || height > e->cfg.max_frame_height || width > e->cfg.max_frame_width)
        return fail(e, PA_ERR_CAPACITY, "pa_preprocess_windows: frames exceed engine capacity");
    hipStream_t s = (hipStream_t)stream;
    const int F = e->cfg.num_fighters;
    int rc = run_preprocess(e, windows_dev, n, height, width, boxes, e->cfg.crop_padding, 1, crops_rgb, e->x0_slot[slot],
                            e->pre_status[slot], s, nullptr, 0, desc_dev);
    if (rc) return rc;
    if (status) HIPCHK(e, hipMemcpyAsync(status, e->pre_status[slot], sizeof(int32_t) * n * F, hipMemcpyDeviceToDevice, s));
    return PA_OK;
}

int pa_detect_postprocess(pa_engine* e, const float* pred, int32_t n_frames, int32_t rows, int32_t num_classes, float conf_thres,
                          float iou_thres, uint32_t class_mask, int32_t max_det, int32_t net_height, int32_t net_width,
                          int32_t img_height, int32_t img_width, float* dets, int32_t* counts, void* stream) {
    if (!e || !pred || !dets || !counts || n_frames < 1 || rows < 1 || num_classes < 1 || num_classes > 32 || max_det < 1 ||
        max_det > 8 || net_height < 1 || net_width < 1 || img_height < 1 || img_width < 1)
        return fail(e, PA_ERR_INVALID_ARG, "pa_detect_postprocess: bad argument");
    DetectParams q;
    memset(&q, 0, sizeof(q));
    q.pred = pred;
    q.n_frames = n_frames; q.rows = rows; q.nc = num_classes; q.max_det = max_det;
    q.conf_thres = conf_thres; q.iou_thres = iou_thres; q.class_mask = class_mask;
    // scale_boxes: gain and pad in double (Python floats), applied as float32 scalars to the float32 boxes
    const double gain = std::min((double)net_height / img_height, (double)net_width / img_width);
    q.gain = (float)gain;
    q.pad_x = (float)(((double)net_width - img_width * gain) / 2.0);
    q.pad_y = (float)(((double)net_height - img_height * gain) / 2.0);
    q.img_w = (float)img_width; q.img_h = (float)img_height;
    q.dets = dets; q.counts = counts;
    ProfScope ps(e, (hipStream_t)stream, "detect_nms", 0.0, (double)n_frames * rows * (5 + num_classes) * 4.0 * max_det);
    HIPCHK(e, launch_detect_nms(q, (hipStream_t)stream));
    return PA_OK;
}

int pa_project_boxes(pa_engine* e, const double* log_rows, int32_t n_rows, double* boxes, void* stream) {
    if (!e || !log_rows || !boxes || n_rows < 1) return fail(e, PA_ERR_INVALID_ARG, "pa_project_boxes: bad argument");
    HIPCHK(e, launch_project_boxes(log_rows, boxes, n_rows, (hipStream_t)stream));
    return PA_OK;
}

int pa_clip_begin(pa_engine* e, int32_t clip_frames) {
    if (!e || clip_frames < 1) return fail(e, PA_ERR_INVALID_ARG, "pa_clip_begin: bad argument");
    if (clip_frames > e->cfg.max_clip_frames) return fail(e, PA_ERR_CAPACITY, "pa_clip_begin: clip longer than max_clip_frames");
    e->clip_frames = clip_frames;
    e->sub_frames = 0;
    e->ready.assign(clip_frames, 0);
    return PA_OK;
}

int pa_clip_begin_batch(pa_engine* e, int32_t n_clips, int32_t clip_frames) {
    if (!e || n_clips < 1 || clip_frames < 2) return fail(e, PA_ERR_INVALID_ARG, "pa_clip_begin_batch: bad argument");
    if ((long long)n_clips * clip_frames > e->cfg.max_clip_frames)
        return fail(e, PA_ERR_CAPACITY, "pa_clip_begin_batch: n_clips * clip_frames exceeds max_clip_frames");
    const int rc = pa_clip_begin(e, n_clips * clip_frames);
    if (rc) return rc;
    e->sub_frames = clip_frames;
    return PA_OK;
}

int pa_preprocess_frames(pa_engine* e, const uint8_t* frames, int32_t n, int32_t height, int32_t width, const double* boxes,
                         int32_t slot, uint8_t* crops_rgb, int32_t* status, void* stream) {
    if (!e || !frames || !boxes || n < 1 || height < 1 || width < 1 || slot < 0 || slot > 1)
        return fail(e, PA_ERR_INVALID_ARG, "pa_preprocess_frames: bad argument");
    if (n > e->cfg.max_batch_frames || height > e->cfg.max_frame_height || width > e->cfg.max_frame_width)
        return fail(e, PA_ERR_CAPACITY, "pa_preprocess_frames: frames exceed engine capacity");
    hipStream_t s = (hipStream_t)stream;
    const int F = e->cfg.num_fighters;
    int rc = run_preprocess(e, frames, n, height, width, boxes, e->cfg.crop_padding, 1, crops_rgb, e->x0_slot[slot],
                            e->pre_status[slot], s);
    if (rc) return rc;
    if (status) HIPCHK(e, hipMemcpyAsync(status, e->pre_status[slot], sizeof(int32_t) * n * F, hipMemcpyDeviceToDevice, s));
    return PA_OK;
}

int pa_backbone_slot(pa_engine* e, int32_t slot, int32_t n, int32_t frame0, void* stream) {
    if (!e || n < 1 || frame0 < 0 || slot < 0 || slot > 1) return fail(e, PA_ERR_INVALID_ARG, "pa_backbone_slot: bad argument");
    if (e->clip_frames < 1) return fail(e, PA_ERR_INVALID_ARG, "pa_backbone_slot: call pa_clip_begin first");
    if (n > e->cfg.max_batch_frames || frame0 + n > e->clip_frames)
        return fail(e, PA_ERR_CAPACITY, "pa_backbone_slot: frames exceed engine / clip capacity");
    hipStream_t s = (hipStream_t)stream;
    const int F = e->cfg.num_fighters;
    HIPCHK(e, hipMemcpyAsync(e->cache_status + (size_t)frame0 * F, e->pre_status[slot], sizeof(int32_t) * n * F,
                             hipMemcpyDeviceToDevice, s));
    int rc = run_backbone(e, n * F, e->x0_slot[slot], e->cache + (size_t)frame0 * F * PA_FEATURE_STRIDE, s);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) e->ready[frame0 + i] = 1;
    return PA_OK;
}

int pa_backbone_frames(pa_engine* e, const uint8_t* frames, int32_t n, int32_t height, int32_t width, const double* boxes,
                       int32_t frame0, uint8_t* crops_rgb, int32_t* status, void* stream) {
    if (!e || frame0 < 0) return fail(e, PA_ERR_INVALID_ARG, "pa_backbone_frames: bad argument");
    if (e->clip_frames < 1) return fail(e, PA_ERR_INVALID_ARG, "pa_backbone_frames: call pa_clip_begin first");
    if (frame0 + n > e->clip_frames) return fail(e, PA_ERR_CAPACITY, "pa_backbone_frames: frames exceed clip capacity");
    int rc = pa_preprocess_frames(e, frames, n, height, width, boxes, 0, crops_rgb, status, stream);
    if (rc) return rc;
    return pa_backbone_slot(e, 0, n, frame0, stream);
}

int pa_backbone_frames_indexed(pa_engine* e, const uint8_t* frames, int32_t n, int32_t height, int32_t width,
                               const double* boxes, const int32_t* frame_ids, uint8_t* crops_rgb, int32_t* status, void* stream) {
    if (!e || !frame_ids) return fail(e, PA_ERR_INVALID_ARG, "pa_backbone_frames_indexed: bad argument");
    if (e->clip_frames < 1) return fail(e, PA_ERR_INVALID_ARG, "pa_backbone_frames_indexed: call pa_clip_begin first");
    int rc = pa_preprocess_frames(e, frames, n, height, width, boxes, 0, crops_rgb, status, stream);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int F = e->cfg.num_fighters;
    rc = run_backbone(e, n * F, e->x0_slot[0], e->feats_tmp, s);
    if (rc) return rc;
    ProfScope ps(e, s, "scatter_features", 0.0, 2.0 * n * F * PA_FEATURE_STRIDE * 4.0);
    HIPCHK(e, launch_scatter_rows(e->feats_tmp, e->pre_status[0], frame_ids, e->cache, e->cache_status, n, F, e->clip_frames,
                                  e->dev_errors, s));
    return PA_OK;
}

int pa_clean_detections(pa_engine* e, const float* dets, const int32_t* counts, int32_t n_labels, int32_t max_det, int32_t n_decoded_frames,
                        double* labels, int32_t* pixel_frame, double* pixel_box, int32_t* crop_kind, float* crop_row, int32_t* info4,
                        void* stream) {
    if (!e || !dets || !counts || !labels || !pixel_frame || !pixel_box || !crop_kind || !crop_row || !info4 || n_labels < 1 ||
        max_det < 1 || n_decoded_frames < 1)
        return fail(e, PA_ERR_INVALID_ARG, "pa_clean_detections: bad argument");
    pa::CleanParams p;
    p.dets = dets; p.counts = counts; p.n_labels = n_labels; p.max_det = max_det; p.n_decoded = n_decoded_frames;
    p.fighters = e->cfg.num_fighters;
    for (int i = 0; i < 4; ++i) p.class_ids[i] = e->cfg.fighter_class_ids[i];
    p.lab = labels; p.pixel_frame = pixel_frame; p.pixel_box = pixel_box; p.crop_kind = crop_kind; p.crop_row = crop_row; p.info = info4;
    const size_t need = (size_t)n_labels * max_det * 6;
    if (need > e->clean_g6_cap) {  // (a table longer than max_clip_frames x 8 detections: the old scratch may still be read by a call in flight)
        HIPCHK(e, hipDeviceSynchronize());
        double* grown = nullptr;
        HIPCHK(e, hipMalloc(&grown, need * sizeof(double)));
        e->allocs.push_back(grown);  // (the scratch it replaces stays in the engine's list and is freed with it)
        e->clean_g6 = grown;
        e->clean_g6_cap = need;
    }
    p.g6v = e->clean_g6;
    HIPCHK(e, pa::launch_clean_labels(p, (hipStream_t)stream));
    return PA_OK;
}

int pa_detector_plan(pa_engine* e, const int32_t* pixel_frame, const double* pixel_box, const int32_t* crop_kind, const int32_t* info4,
                     int32_t n_labels, int32_t* det_index, int32_t* src_own, int32_t* rep_entry, double* rep_boxes, int32_t* rep_src,
                     int32_t* words5, void* stream) {
    if (!e || !pixel_frame || !pixel_box || !crop_kind || !info4 || !det_index || !src_own || !rep_entry || !rep_boxes || !rep_src || !words5 ||
        n_labels < 1)
        return fail(e, PA_ERR_INVALID_ARG, "pa_detector_plan: bad argument");
    HIPCHK(e, pa::launch_detector_plan(pixel_frame, pixel_box, crop_kind, info4, n_labels, e->cfg.num_fighters, det_index, src_own, rep_entry,
                                       rep_boxes, rep_src, words5, (hipStream_t)stream));
    return PA_OK;
}

int pa_detector_plan_desc(pa_engine* e, pa_crop_image* desc, const int32_t* crop_kind, int32_t n_frames, int32_t step_frames,
                          long long region_bytes, const int32_t* rep_entry, int32_t n_rep, long long rep_base, void* stream) {
    if (!e || !desc || !crop_kind || n_frames < 1 || step_frames < 1 || region_bytes < 0 || n_rep < 0 || (n_rep && !rep_entry) || rep_base < 0)
        return fail(e, PA_ERR_INVALID_ARG, "pa_detector_plan_desc: bad argument");
    HIPCHK(e, pa::launch_detector_desc(reinterpret_cast<pa::CropImageDesc*>(desc), crop_kind, n_frames * e->cfg.num_fighters, e->cfg.num_fighters,
                                       step_frames, region_bytes, rep_entry, n_rep, rep_base, (hipStream_t)stream));
    return PA_OK;
}

int pa_square_crops_src(pa_engine* e, const uint8_t* frames, int32_t n_src, int32_t height, int32_t width, const double* boxes,
                        const int32_t* src_frame, int32_t n, int32_t padding, int32_t swap_rb, uint8_t* crops, int32_t* status, void* stream) {
    if (!e || !frames || !boxes || !src_frame || !crops || n < 1 || n_src < 1 || height < 1 || width < 1 || padding < 0)
        return fail(e, PA_ERR_INVALID_ARG, "pa_square_crops_src: bad argument");
    if (n > e->cfg.max_batch_frames || height > e->cfg.max_frame_height || width > e->cfg.max_frame_width)
        return fail(e, PA_ERR_CAPACITY, "pa_square_crops_src: crops exceed engine capacity");
    return run_preprocess(e, frames, n, height, width, boxes, padding, swap_rb, crops, nullptr, status, (hipStream_t)stream, src_frame, n_src);
}

int pa_save_one_box_crops(pa_engine* e, const uint8_t* frames, int32_t n_src, int32_t height, int32_t width, const float* dets,
                          const int32_t* counts, int32_t max_det, const int32_t* det_index, const int32_t* src_frame, int32_t n,
                          int32_t jpeg_quality, uint8_t* images, size_t images_capacity, pa_crop_image* desc, void* stream) {
    if (!e || !frames || !dets || !counts || !images || !desc || n < 1 || n_src < 1 || height < 1 || width < 1 || max_det < 1 ||
        jpeg_quality < 0 || jpeg_quality > 100 || (!src_frame && n_src != n))
        return fail(e, PA_ERR_INVALID_ARG, "pa_save_one_box_crops: bad argument");
    const int F = e->cfg.num_fighters;
    if (n * F > e->max_crops) return fail(e, PA_ERR_CAPACITY, "pa_save_one_box_crops: more frames than max_batch_frames");
    pa::SaveBoxParams p;
    memset(&p, 0, sizeof p);
    p.frames = frames; p.height = height; p.width = width; p.fighters = F; p.n_entries = n * F;
    p.dets = dets; p.counts = counts; p.max_det = max_det; p.det_index = det_index;
    p.src_frame = src_frame; p.n_src = n_src;
    for (int i = 0; i < 4; ++i) p.class_ids[i] = e->cfg.fighter_class_ids[i];
    p.gain = 1.02f; p.pad = 10.0f;  // save_one_box's defaults, which detect.py does not override
    p.images = images; p.capacity = images_capacity;
    p.desc = reinterpret_cast<pa::CropImageDesc*>(desc);
    p.rects = reinterpret_cast<pa::SaveBoxRect*>(e->savebox_rects);
    p.overflow = e->dev_errors + 1;
    p.quality = jpeg_quality;
    if (jpeg_quality > 0) {
        // jpeg_set_quality(quality, force_baseline = TRUE) on the standard tables (jcparam.c), natural order
        static const int lum[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
                                    14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
                                    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
        static const int chr[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
                                    47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
        const int scale = jpeg_quality < 50 ? 5000 / jpeg_quality : 200 - jpeg_quality * 2;
        for (int i = 0; i < 64; ++i) {
            const int a = (lum[i] * scale + 50) / 100, b = (chr[i] * scale + 50) / 100;
            p.qtab[i] = a < 1 ? 1 : (a > 255 ? 255 : a);
            p.qtab[64 + i] = b < 1 ? 1 : (b > 255 ? 255 : b);
        }
    }
    ProfScope ps(e, (hipStream_t)stream, "save_one_box_crops", 0.0, 0.0);
    HIPCHK(e, pa::launch_save_one_box(p, (hipStream_t)stream));
    return PA_OK;
}

int pa_device_errors(pa_engine* e, int32_t* bad_frame_ids_host, void* stream) {
    if (!e || !bad_frame_ids_host) return fail(e, PA_ERR_INVALID_ARG, "pa_device_errors: bad argument");
    hipStream_t s = (hipStream_t)stream;
    int32_t h[4] = {0, 0, 0, 0};
    HIPCHK(e, hipMemcpyAsync(h, e->dev_errors, sizeof(h), hipMemcpyDeviceToHost, s));
    HIPCHK(e, hipMemsetAsync(e->dev_errors, 0, sizeof(h), s));
    HIPCHK(e, hipStreamSynchronize(s));
    *bad_frame_ids_host = h[0];
    if (h[1] != 0)
        return fail(e, PA_ERR_CAPACITY, "pa_save_one_box_crops: " + std::to_string(h[1]) + " crop image(s) did not fit the image buffer");
    if (h[0] != 0)
        return fail(e, PA_ERR_CAPACITY, "pa_backbone_frames_indexed: " + std::to_string(h[0]) +
                                            " frame id(s) outside the clip were skipped on the device");
    return PA_OK;
}

int pa_backbone_frames_src(pa_engine* e, const uint8_t* frames, int32_t n_src, int32_t height, int32_t width,
                           const double* boxes, const int32_t* src_frame, int32_t n, int32_t frame0, uint8_t* crops_rgb,
                           int32_t* status, void* stream) {
    if (!e || !frames || !boxes || !src_frame || n < 1 || n_src < 1 || frame0 < 0 || height < 1 || width < 1)
        return fail(e, PA_ERR_INVALID_ARG, "pa_backbone_frames_src: bad argument");
    if (e->clip_frames < 1) return fail(e, PA_ERR_INVALID_ARG, "pa_backbone_frames_src: call pa_clip_begin first");
    if (n > e->cfg.max_batch_frames || frame0 + n > e->clip_frames || height > e->cfg.max_frame_height ||
        width > e->cfg.max_frame_width)
        return fail(e, PA_ERR_CAPACITY, "pa_backbone_frames_src: frames exceed engine / clip capacity");
    hipStream_t s = (hipStream_t)stream;
    const int F = e->cfg.num_fighters;
    int rc = run_preprocess(e, frames, n, height, width, boxes, e->cfg.crop_padding, 1, crops_rgb, e->x0_slot[0],
                            e->pre_status[0], s, src_frame, n_src);
    if (rc) return rc;
    if (status) HIPCHK(e, hipMemcpyAsync(status, e->pre_status[0], sizeof(int32_t) * n * F, hipMemcpyDeviceToDevice, s));
    return pa_backbone_slot(e, 0, n, frame0, stream);
}

int pa_clip_mark_ready(pa_engine* e, const int32_t* frame_ids_host, int32_t n) {
    if (!e || !frame_ids_host || n < 0) return fail(e, PA_ERR_INVALID_ARG, "pa_clip_mark_ready: bad argument");
    for (int i = 0; i < n; ++i) {
        if (frame_ids_host[i] < 0 || frame_ids_host[i] >= e->clip_frames)
            return fail(e, PA_ERR_CAPACITY, "pa_clip_mark_ready: frame id outside the clip");
        e->ready[frame_ids_host[i]] = 1;
    }
    return PA_OK;
}

int pa_head_frames(pa_engine* e, int32_t lo, int32_t hi, pa_record* records, float* logp, void* stream) {
    if (!e || lo < 1 || hi <= lo) return fail(e, PA_ERR_INVALID_ARG, "pa_head_frames: bad frame range");
    if (e->clip_frames < 1) return fail(e, PA_ERR_INVALID_ARG, "pa_head_frames: call pa_clip_begin first");
    if (hi > e->clip_frames) return fail(e, PA_ERR_CAPACITY, "pa_head_frames: frame numbers run to max_frames - 1");
    const int S = e->cfg.sequence_length, F = e->cfg.num_fighters, D = e->cfg.frame_delta;
    const int mid = S / 2, reach = std::abs(D) * mid * mid;
    // every frame a window of [lo, hi) can touch must be cached
    {
        const int first = std::max(1, lo - reach), last = std::min(e->clip_frames - 1, hi - 1 + reach);
        for (int f = first; f <= last; ++f)
            if (!e->ready[f - 1]) return fail(e, PA_ERR_NOT_READY, "pa_head_frames: features of frame " + std::to_string(f) + " missing");
    }
    hipStream_t s = (hipStream_t)stream;
    const int frames_per_pass = e->max_crops / F;
    for (int f0 = lo; f0 < hi; f0 += frames_per_pass) {
        const int cnt = std::min(frames_per_pass, hi - f0);
        // the index table only depends on (first frame, count, clip length): a steady stream of
        // equal clips (bench.py, the frame-parallel runner) re-uses it instead of re-launching
        if (e->gather_key[0] != f0 || e->gather_key[1] != cnt || e->gather_key[2] != e->clip_frames || e->gather_key[3] != e->sub_frames) {
            HIPCHK(e, launch_window_gather(e->gather, f0, cnt, F, S, D, e->clip_frames, 1, e->sub_frames, s));
            e->gather_key[0] = f0; e->gather_key[1] = cnt; e->gather_key[2] = e->clip_frames; e->gather_key[3] = e->sub_frames;
        }
        const size_t o = (size_t)(f0 - lo) * F;
        int rc = run_head(e, cnt * F, e->cache, e->gather, e->cache_status, records ? records + o : nullptr,
                          logp ? logp + o * e->cfg.num_actions : nullptr, s);
        if (rc) return rc;
    }
    return PA_OK;
}

int pa_infer_clip(pa_engine* e, const uint8_t* frames, int32_t n, int32_t height, int32_t width, const double* boxes,
                  pa_record* records, float* logp, uint8_t* crops_rgb, int32_t* status, void* stream) {
    if (!e || n < 2) return fail(e, PA_ERR_INVALID_ARG, "pa_infer_clip: need at least 2 frames");
    int rc = pa_clip_begin(e, n);
    if (rc) return rc;
    const int F = e->cfg.num_fighters;
    const size_t frame_bytes = (size_t)height * width * 3;
    for (int f0 = 0; f0 < n; f0 += e->cfg.max_batch_frames) {
        const int cnt = std::min(e->cfg.max_batch_frames, n - f0);
        rc = pa_backbone_frames(e, frames + (size_t)f0 * frame_bytes, cnt, height, width, boxes + (size_t)f0 * F * 4, f0,
                                crops_rgb ? crops_rgb + (size_t)f0 * F * PA_CROP * PA_CROP * 3 : nullptr,
                                status ? status + (size_t)f0 * F : nullptr, stream);
        if (rc) return rc;
    }
    return pa_head_frames(e, 1, n, records, logp, stream);
}

int pa_features_export(pa_engine* e, int32_t frame0, int32_t n, float* feats, void* stream) {
    if (!e || !feats || frame0 < 0 || n < 1 || frame0 + n > e->clip_frames)
        return fail(e, PA_ERR_INVALID_ARG, "pa_features_export: bad range (call pa_clip_begin first)");
    for (int i = 0; i < n; ++i)
        if (!e->ready[frame0 + i])
            return fail(e, PA_ERR_NOT_READY, "pa_features_export: features of frame " + std::to_string(frame0 + i + 1) +
                                                 " are not cached in this clip");
    const size_t row = (size_t)e->cfg.num_fighters * PA_FEATURE_STRIDE;
    HIPCHK(e, hipMemcpyAsync(feats, e->cache + (size_t)frame0 * row, sizeof(float) * n * row, hipMemcpyDeviceToDevice,
                             (hipStream_t)stream));
    return PA_OK;
}

int pa_features_import(pa_engine* e, int32_t frame0, int32_t n, const float* feats, void* stream) {
    if (!e || !feats || frame0 < 0 || n < 1 || frame0 + n > e->clip_frames)
        return fail(e, PA_ERR_INVALID_ARG, "pa_features_import: bad range (call pa_clip_begin first)");
    const size_t row = (size_t)e->cfg.num_fighters * PA_FEATURE_STRIDE;
    HIPCHK(e, hipMemcpyAsync(e->cache + (size_t)frame0 * row, feats, sizeof(float) * n * row, hipMemcpyDeviceToDevice,
                             (hipStream_t)stream));
    for (int i = 0; i < n; ++i) e->ready[frame0 + i] = 1;
    return PA_OK;
}

int pa_profile_enable(pa_engine* e, int32_t on) {
    if (!e) return PA_ERR_INVALID_ARG;
    e->profiling = on != 0;
    return PA_OK;
}

int pa_profile_read(pa_engine* e, pa_kernel_stat* stats, int32_t max_stats, int32_t* n_stats) {
    if (!e || !stats || !n_stats || max_stats < 1) return fail(e, PA_ERR_INVALID_ARG, "pa_profile_read: bad argument");
    HIPCHK(e, hipDeviceSynchronize());
    std::vector<pa_kernel_stat> acc(e->prof_names.size());
    for (size_t i = 0; i < acc.size(); ++i) {
        memset(&acc[i], 0, sizeof(pa_kernel_stat));
        snprintf(acc[i].name, sizeof(acc[i].name), "%s", e->prof_names[i].c_str());
    }
    for (ProfEntry& p : e->prof_log) {
        float ms = 0.f;
        HIPCHK(e, hipEventElapsedTime(&ms, p.start, p.stop));
        acc[p.name_id].launches += 1;
        acc[p.name_id].total_ms += ms;
        acc[p.name_id].flops += p.flops;
        acc[p.name_id].bytes += p.bytes;
        acc[p.name_id].flops_executed += p.flops_executed;
        e->event_pool.push_back(p.start);
        e->event_pool.push_back(p.stop);
    }
    e->prof_log.clear();
    int n = 0;
    for (size_t i = 0; i < acc.size() && n < max_stats; ++i)
        if (acc[i].launches > 0) stats[n++] = acc[i];
    *n_stats = n;
    return PA_OK;
}

int pa_set_crop_jpeg_quality(pa_engine* e, int32_t quality) {
    if (!e || quality < 0 || quality > 100) return fail(e, PA_ERR_INVALID_ARG, "pa_set_crop_jpeg_quality: 0 (off) .. 100");
    if (quality > 0) {
        // jpeg_set_quality(quality, force_baseline = TRUE) on the standard tables (jcparam.c), natural order
        static const int lum[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
                                    14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
                                    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
        static const int chr[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
                                    47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
        const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
        int32_t tab[128];
        for (int i = 0; i < 64; ++i) {
            const int a = (lum[i] * scale + 50) / 100, b = (chr[i] * scale + 50) / 100;
            tab[i] = a < 1 ? 1 : (a > 255 ? 255 : a);
            tab[64 + i] = b < 1 ? 1 : (b > 255 ? 255 : b);
        }
        HIPCHK(e, hipMemcpy(e->jpeg_qtab, tab, sizeof(tab), hipMemcpyHostToDevice));
    }
    e->jpeg_quality = quality;
    return PA_OK;
}

int pa_stream_spin(pa_engine* e, int32_t microseconds, void* stream) {
    if (!e || microseconds < 0 || microseconds > 100000) return fail(e, PA_ERR_INVALID_ARG, "pa_stream_spin: 0..100000 us");
    HIPCHK(e, launch_spin(microseconds, (hipStream_t)stream));
    return PA_OK;
}

int pa_stream_gate(pa_engine* e, int32_t max_microseconds, void* stream) {
    if (!e || max_microseconds < 1 || max_microseconds > 100000) return fail(e, PA_ERR_INVALID_ARG, "pa_stream_gate: 1..100000 us");
    if (!e->gate_flag) HIPCHK(e, hipHostMalloc((void**)&e->gate_flag, 64, hipHostMallocCoherent | hipHostMallocMapped));
    __atomic_store_n(e->gate_flag, 0, __ATOMIC_SEQ_CST);
    HIPCHK(e, launch_gate(e->gate_flag, max_microseconds, (hipStream_t)stream));
    return PA_OK;
}

int pa_stream_gate_open(pa_engine* e) {
    if (!e) return PA_ERR_INVALID_ARG;
    if (e->gate_flag) __atomic_store_n(e->gate_flag, 1, __ATOMIC_SEQ_CST);
    return PA_OK;
}

int pa_stream_sync(pa_engine* e, void* stream) {
    HIPCHK(e, hipStreamSynchronize((hipStream_t)stream));
    return PA_OK;
}

}  // extern "C"
