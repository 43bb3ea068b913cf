// Small memory-bound kernels around the implicit-GEMM engine: layout change
// for the b1 entry point, the two pooling layers, the window index table and
// the MLP + log-softmax + argmax tail. All are streaming / reduction work:
// 16-byte accesses where the layout allows, wave64 shuffle reductions, no LDS
// beyond one staged row.
#include "pa_kernels.h"
#include "../../include/playaid_hip.h"

namespace pa {

// x[n][3][128][128] (NCHW fp32, the tensor `model(x)` receives,
// cnn_action_detector.py:29-31) -> zero-bordered NHWC4 [n][134][134][4].
__global__ __launch_bounds__(256) void nchw_to_padded_kernel(const float* __restrict__ x, float* __restrict__ out, int n, int out_bf16) {
    const size_t total = (size_t)n * 128 * 128;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int img = (int)(i >> 14);
        const int pix = (int)(i & 16383);
        const int y = pix >> 7, xx = pix & 127;
        const float* s = x + (size_t)img * 3 * 16384 + pix;
        float4 v;
        v.x = s[0];
        v.y = s[16384];
        v.z = s[32768];
        v.w = 0.f;
        const size_t o = ((size_t)img * 134 + (y + 3)) * 134 + (xx + 3);
        if (out_bf16) {  // bf16 conv path: the stem multiplies bf16 pixels (round to nearest even)
            uint32_t u[3] = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z)};
#pragma unroll
            for (int k = 0; k < 3; ++k) u[k] = (u[k] + 0x7fffu + ((u[k] >> 16) & 1u)) >> 16;
            reinterpret_cast<uint2*>(out)[o] = make_uint2(u[0] | (u[1] << 16), u[2]);
        } else {
            reinterpret_cast<float4*>(out)[o] = v;
        }
    }
}

hipError_t launch_nchw_to_padded(const float* x, float* out, int32_t n, int32_t out_bf16, hipStream_t s) {
    const size_t total = (size_t)n * 16384;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(nchw_to_padded_kernel, dim3(grid), dim3(256), 0, s, x, out, n, out_bf16);
    return hipGetLastError();
}

// max_pool2d(3, stride 2, padding 1) on the post-ReLU stem output. Input is the
// zero-bordered [n][66][66][64]; since every value is >= 0 the zero border
// gives the same maximum as torch's -inf padding. Output zero-bordered
// [n][34][34][64].
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ in, float* __restrict__ out, int n) {
    const size_t total = (size_t)n * 32 * 32 * 16;  // float4 units
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i & 15);
        const int ox = (int)((i >> 4) & 31);
        const int oy = (int)((i >> 9) & 31);
        const int img = (int)(i >> 14);
        const float4* src = reinterpret_cast<const float4*>(in) + (((size_t)img * 66 + oy * 2) * 66 + ox * 2) * 16 + c4;
        float4 m = src[0];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 v = src[((size_t)ky * 66 + kx) * 16];
                m.x = fmaxf(m.x, v.x);
                m.y = fmaxf(m.y, v.y);
                m.z = fmaxf(m.z, v.z);
                m.w = fmaxf(m.w, v.w);
            }
        reinterpret_cast<float4*>(out)[(((size_t)img * 34 + oy + 1) * 34 + ox + 1) * 16 + c4] = m;
    }
}

hipError_t launch_maxpool(const float* in, float* out, int32_t n, hipStream_t s) {
    const size_t total = (size_t)n * 32 * 32 * 16;
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(maxpool_kernel, dim3(grid), dim3(256), 0, s, in, out, n);
    return hipGetLastError();
}

// bf16 storage variant of the max-pool (bf16 conv path): 8 channels per thread (one 16-byte load).
// Non-negative bf16 values order like their bit patterns, so the maximum is an integer max.
__global__ __launch_bounds__(256) void maxpool_bf16_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int n) {
    const size_t total = (size_t)n * 32 * 32 * 8;  // 8-channel units
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i & 7);
        const int ox = (int)((i >> 3) & 31);
        const int oy = (int)((i >> 8) & 31);
        const int img = (int)(i >> 13);
        const uint4* src = reinterpret_cast<const uint4*>(in) + (((size_t)img * 66 + oy * 2) * 66 + ox * 2) * 8 + c8;
        uint4 m = src[0];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const uint4 v = src[((size_t)ky * 66 + kx) * 8];
                const uint32_t mv[4] = {m.x, m.y, m.z, m.w}, vv[4] = {v.x, v.y, v.z, v.w};
                uint32_t r[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t lo = (mv[k] & 0xffffu) > (vv[k] & 0xffffu) ? (mv[k] & 0xffffu) : (vv[k] & 0xffffu);
                    const uint32_t hi = (mv[k] >> 16) > (vv[k] >> 16) ? (mv[k] >> 16) : (vv[k] >> 16);
                    r[k] = lo | (hi << 16);
                }
                m = make_uint4(r[0], r[1], r[2], r[3]);
            }
        reinterpret_cast<uint4*>(out)[(((size_t)img * 34 + oy + 1) * 34 + ox + 1) * 8 + c8] = m;
    }
}

hipError_t launch_maxpool_bf16(const void* in, void* out, int32_t n, hipStream_t s) {
    const size_t total = (size_t)n * 32 * 32 * 8;
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(maxpool_bf16_kernel, dim3(grid), dim3(256), 0, s, (const uint16_t*)in, (uint16_t*)out, n);
    return hipGetLastError();
}

// bf16 layer4 output -> fp32 pooled features (the fc and the head stay in fp32).
__global__ __launch_bounds__(256) void avgpool_bf16_kernel(const uint16_t* __restrict__ in, float* __restrict__ out, int n) {
    const size_t total = (size_t)n * 512;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i & 511);
        const int img = (int)(i >> 9);
        const uint16_t* src = in + (size_t)img * 36 * 512 + c;
        float sum = 0.f;
#pragma unroll
        for (int y = 1; y <= 4; ++y)
#pragma unroll
            for (int x = 1; x <= 4; ++x) sum += __uint_as_float((uint32_t)src[(size_t)(y * 6 + x) * 512] << 16);
        out[i] = sum * 0.0625f;
    }
}

hipError_t launch_avgpool_bf16(const void* in, float* out, int32_t n, hipStream_t s) {
    const size_t total = (size_t)n * 512;
    const int grid = (int)((total + 255) / 256);
    hipLaunchKernelGGL(avgpool_bf16_kernel, dim3(grid), dim3(256), 0, s, (const uint16_t*)in, out, n);
    return hipGetLastError();
}

// adaptive_avg_pool2d((1,1)) over the 4x4 interior of the zero-bordered
// [n][6][6][512] layer4 output -> [n][512].
__global__ __launch_bounds__(256) void avgpool_kernel(const float* __restrict__ in, float* __restrict__ out, int n) {
    const size_t total = (size_t)n * 128;  // float4 units
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i & 127);
        const int img = (int)(i >> 7);
        const float4* src = reinterpret_cast<const float4*>(in) + (size_t)img * 36 * 128 + c4;
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int y = 1; y <= 4; ++y)
#pragma unroll
            for (int x = 1; x <= 4; ++x) {
                const float4 v = src[(size_t)(y * 6 + x) * 128];
                sum.x += v.x;
                sum.y += v.y;
                sum.z += v.z;
                sum.w += v.w;
            }
        sum.x *= 0.0625f;
        sum.y *= 0.0625f;
        sum.z *= 0.0625f;
        sum.w *= 0.0625f;
        reinterpret_cast<float4*>(out)[i] = sum;
    }
}

hipError_t launch_avgpool(const float* in, float* out, int32_t n, hipStream_t s) {
    const size_t total = (size_t)n * 128;
    const int grid = (int)((total + 255) / 256);
    hipLaunchKernelGGL(avgpool_kernel, dim3(grid), dim3(256), 0, s, in, out, n);
    return hipGetLastError();
}

// action_sample_from_frame_middle_out (dataset_utils.py:109-138) on the device:
// window w = (frame_num_lo + w / fighters, fighter w % fighters); slot t reads
// frame number clamp(f -/+ delta*(mid-t)^2) and hence feature-cache row
// (frame_num - 1) * fighters + fighter.
// sub_frames > 0: the clip is a batch of independent clips of sub_frames frames each (pa_clip_begin_batch); frame
// number f belongs to clip c = (f - 1) / sub_frames and its window is clamped to that clip's own frame numbers
// c * sub_frames + 1 .. c * sub_frames + sub_frames - 1.
__global__ void window_gather_kernel(int32_t* gather, int frame_num_lo, int count, int fighters, int seq, int delta,
                                     int max_frames, int min_frame, int sub_frames) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = count * fighters * seq;
    if (i >= total) return;
    const int t = i % seq;
    const int w = i / seq;
    const int fighter = w % fighters;
    const int f = frame_num_lo + w / fighters;
    if (sub_frames > 0) {
        const int c = (f - 1) / sub_frames;
        min_frame = c * sub_frames + 1;
        max_frames = (c + 1) * sub_frames;
    }
    const int mid = seq / 2;
    int off = delta * (mid - t) * (mid - t);
    if (off < 0) off = -off;
    int fn;
    if (t <= mid) {
        fn = f - off;
        fn = fn > min_frame ? fn : min_frame;
    } else {
        fn = f + off;
        fn = fn < max_frames - 1 ? fn : max_frames - 1;
    }
    gather[i] = (fn - 1) * fighters + fighter;
}

hipError_t launch_window_gather(int32_t* gather, int32_t frame_num_lo, int32_t count, int32_t fighters, int32_t seq,
                                int32_t delta, int32_t max_frames, int32_t min_frame, int32_t sub_frames, hipStream_t s) {
    const int total = count * fighters * seq;
    hipLaunchKernelGGL(window_gather_kernel, dim3((total + 255) / 256), dim3(256), 0, s, gather, frame_num_lo, count,
                       fighters, seq, delta, max_frames, min_frame, sub_frames);
    return hipGetLastError();
}

// feats[i][fighter][:] -> cache[(ids[i]*fighters + fighter)][:], status likewise: places the
// features of a resolution bucket (non-contiguous frame numbers) into the clip's cache. An id
// outside [0, clip_frames) writes nothing and bumps the engine's device error counter.
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ feats, const int32_t* __restrict__ st,
                                                           const int32_t* __restrict__ ids, float* __restrict__ cache,
                                                           int32_t* __restrict__ cache_st, int n, int fighters,
                                                           int clip_frames, int32_t* __restrict__ bad_ids) {
    const int row = blockIdx.x;  // 0 .. n*fighters-1
    const int id = ids[row / fighters];
    if ((unsigned)id >= (unsigned)clip_frames) {
        if (threadIdx.x == 0 && row % fighters == 0) atomicAdd(bad_ids, 1);
        return;
    }
    const int dst = id * fighters + row % fighters;
    const float4* s4 = reinterpret_cast<const float4*>(feats + (size_t)row * PA_FEATURE_STRIDE);
    float4* d4 = reinterpret_cast<float4*>(cache + (size_t)dst * PA_FEATURE_STRIDE);
    d4[threadIdx.x] = s4[threadIdx.x];
    if (threadIdx.x == 0) cache_st[dst] = st[row];
}

hipError_t launch_scatter_rows(const float* feats, const int32_t* st, const int32_t* ids, float* cache, int32_t* cache_st,
                               int32_t n, int32_t fighters, int32_t clip_frames, int32_t* bad_ids, hipStream_t s) {
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(n * fighters), dim3(PA_FEATURE_STRIDE / 4), 0, s, feats, st, ids, cache,
                       cache_st, n, fighters, clip_frames, bad_ids);
    return hipGetLastError();
}

__global__ void identity_gather_kernel(int32_t* gather, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) gather[i] = i;
}

hipError_t launch_identity_gather(int32_t* gather, int32_t n, hipStream_t s) {
    hipLaunchKernelGGL(identity_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, s, gather, n);
    return hipGetLastError();
}

// classifier of SpatialStreamCNN (cnn_action_detector.py:27,41): Linear(512,128)
// + ReLU + Linear(128,A), then F.log_softmax (:92), argmax and exp (ai_runner.py
// :474-477). One workgroup (1024 threads) per window; the 512-vector sits in LDS,
// the hidden layer is 128 outputs x 8 k-slices with coalesced k-major weight reads,
// the A <= 64 logits live one per lane of wave 0 for the softmax reductions.
__global__ __launch_bounds__(1024) void head_mlp_kernel(const HeadParams p) {
    // w2 is stored transposed [512 k][128 o] and w3 as [128 k][64 (A padded)] so that the lanes
    // of a wave (consecutive outputs) read consecutive floats; h1/h2 come from LDS broadcasts.
    __shared__ float h1[512];
    __shared__ float part[8][128];
    __shared__ float h2[128];
    const int w = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    if (tid < 512) {
        if (p.h1) {
            h1[tid] = p.h1[(size_t)w * 512 + tid];
        } else {
            // the Conv1d's split-K slabs, summed in slab order, + bias, ReLU: exactly what
            // splitk_reduce_kernel would have written (that launch is saved)
            float s = p.slab[(size_t)w * 512 + tid];
            for (int z = 1; z < p.splitk; ++z) s += p.slab[((size_t)z * p.nwin + w) * 512 + tid];
            s += p.b1[tid];
            h1[tid] = s > 0.f ? s : 0.f;
        }
    }
    __syncthreads();
    {
        // 1024 threads = 128 outputs x 8 k-slices of 64: every load is independent and coalesced
        const int o = tid & 127, slice = tid >> 7;
        const float* wc = p.w2 + (size_t)slice * 64 * 128 + o;
        const float* hh = h1 + slice * 64;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int k = 0; k < 64; k += 4) {
            s0 += wc[(size_t)(k + 0) * 128] * hh[k + 0];
            s1 += wc[(size_t)(k + 1) * 128] * hh[k + 1];
            s2 += wc[(size_t)(k + 2) * 128] * hh[k + 2];
            s3 += wc[(size_t)(k + 3) * 128] * hh[k + 3];
        }
        part[slice][o] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (tid < 128) {
        float s = p.b2[tid];
#pragma unroll
        for (int i = 0; i < 8; ++i) s += part[i][tid];
        h2[tid] = s > 0.f ? s : 0.f;
    }
    __syncthreads();
    if (wave != 0) return;
    const int A = p.num_actions;
    float logit = -INFINITY;
    {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 8
        for (int k = 0; k < 128; k += 4) {
            s0 += p.w3[(k + 0) * 64 + lane] * h2[k + 0];
            s1 += p.w3[(k + 1) * 64 + lane] * h2[k + 1];
            s2 += p.w3[(k + 2) * 64 + lane] * h2[k + 2];
            s3 += p.w3[(k + 3) * 64 + lane] * h2[k + 3];
        }
        if (lane < A) logit = (s0 + s1) + (s2 + s3) + p.b3[lane];
    }
    float mx = logit;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    float ex = lane < A ? expf(logit - mx) : 0.f;
    float se = ex;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) se += __shfl_xor(se, d, 64);
    const float lp = logit - mx - logf(se);
    if (p.logp && lane < A) p.logp[(size_t)w * A + lane] = lp;
    // argmax with first-index tie break (torch.argmax)
    float bv = lane < A ? lp : -INFINITY;
    int bi = lane;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float ov = __shfl_xor(bv, d, 64);
        const int oi = __shfl_xor(bi, d, 64);
        if (ov > bv || (ov == bv && oi < bi)) {
            bv = ov;
            bi = oi;
        }
    }
    if (p.records && lane == 0) {
        pa_record r;
        const int fighter = w % p.fighters;
        r.char_id = p.class_ids[fighter];
        r.action_id = bi;
        r.prob = expf(bv);
        int st = 0;
        if (p.crop_status && p.gather) {
            const int row = p.gather[(size_t)w * p.seq + p.seq / 2];
            if (row >= 0) st = p.crop_status[row];
        }
        r.status = st;
        reinterpret_cast<pa_record*>(p.records)[w] = r;
    }
}

hipError_t launch_head_mlp(const HeadParams& p, hipStream_t s) {
    if (p.nwin <= 0) return hipSuccess;
    if (p.num_actions > 64 || p.num_actions < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(head_mlp_kernel, dim3(p.nwin), dim3(1024), 0, s, p);
    return hipGetLastError();
}

// A single-thread kernel that keeps its stream busy for `ticks` of the 100 MHz real-time clock: the probe
// parallel._concurrent_streams uses to find out whether two HIP streams execute side by side (streams multiplexed
// onto one hardware queue run strictly in turn).
__global__ void spin_kernel(long long ticks, int* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int n = 0;
    // (the iteration cap bounds the kernel to about a second whatever the clock does)
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks && n < (1 << 24)) ++n;
    if (ticks < 0) *sink = n;  // (never true: keeps the loop observable)
}

// A gate: keeps its stream busy until the HOST stores a non-zero value to *flag (coherent pinned memory), or max_ticks
// (100 MHz) have passed -- the bound is what makes it safe: a host that never opens the gate costs max_ticks, not a hang.
// parallel.ClipLanes holds the first clip of every lane behind one such gate and opens it when all of them are enqueued.
__global__ void gate_kernel(const int* flag, long long max_ticks, int* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int n = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0 &&
           (long long)(__builtin_amdgcn_s_memrealtime() - t0) < max_ticks && n < (1 << 24)) {
        __builtin_amdgcn_s_sleep(8);
        ++n;
    }
    if (max_ticks < 0) *sink = n;
}

hipError_t launch_gate(const int* flag, int32_t max_microseconds, hipStream_t s) {
    hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(1), 0, s, flag, (long long)max_microseconds * 100, (int*)nullptr);
    return hipGetLastError();
}

hipError_t launch_spin(int32_t microseconds, hipStream_t s) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, s, (long long)microseconds * 100, (int*)nullptr);
    return hipGetLastError();
}

}  // namespace pa
