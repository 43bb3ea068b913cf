// Temporal head of the reference's alternative model, RNNActionDetector
// (playaid/models/rnn_action_detector.py:55-95): ResNet-18 features (fc -> 300) ->
// nn.LSTM(300, 512, num_layers=3) -> Linear(512,128) + ReLU -> Linear(128, A) -> log_softmax, one output row per
// (window, frame). The backbone runs on the engine's convolution kernels (pa_backbone_windows); this file is the
// recurrent part and its decoder.
//
// The reference feeds the LSTM a [B, S, 300] tensor WITHOUT batch_first, so torch treats dimension 0 -- the
// windows -- as time and dimension 1 -- the S frames of a window -- as the batch (:88-90): the state runs from
// one window to the next. That is reproduced as is: pa_lstm_forward takes (seq_len, batch) in torch's order.
//
// Per layer: the input projection of every time step is one GEMM ([L*N, in] x [in, 4H], bias b_ih), then one
// small launch per time step adds h(t-1) W_hh^T + b_hh, applies the gates (torch order i, f, g, o) and writes
// h(t), c(t). All fp32; gates with expf / tanhf. This is a side path (7-row batches, launch-latency bound), not
// the hot loop: no MFMA tiling here on purpose.
#include "pa_kernels.h"
#include "../../include/playaid_hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace pa {
namespace {

constexpr int LSTM_NMAX = 16;  // batch rows of a time step (the reference's S <= 15)
constexpr int LSTM_UNITS_DEFAULT = 4;  // hidden units per workgroup of lstm_layer_kernel: 128 workgroups at H = 512. Measured per (layer, step), us:
                                       // 8 units 11.0, 4 units 8.8 (sweep 3.3 + product 4.6 + gates 0.7), 2 units 9.4 (5.2 + 3.4 + 0.8: 256 CUs
                                       // sweeping the same 28 KB), 1 unit 13.5 (profiles/r04_lstm_units.txt)

// C[m*ldc + n] = act(sum_k X[m*ld + k] * W[n*K + k] + bias[n])   (M x K) x (N x K)^T, 64 x 64 tiles, 4 x 4 per thread
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ X, int ld, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ Cm, int ldc, int M, int N, int K,
                                                         int relu) {
    __shared__ float xs[16][65], ws[16][65];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int i = threadIdx.x; i < 64 * 16; i += 256) {
            const int r = i >> 4, k = i & 15;
            xs[k][r] = (m0 + r < M && k0 + k < K) ? X[(size_t)(m0 + r) * ld + k0 + k] : 0.f;
            ws[k][r] = (n0 + r < N && k0 + k < K) ? W[(size_t)(n0 + r) * K + k0 + k] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = xs[k][ty * 4 + i];
                b[i] = ws[k][tx * 4 + i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
            if (m < M && n < N) {
                float v = acc[i][j] + (bias ? bias[n] : 0.f);
                if (relu) v = v > 0.f ? v : 0.f;
                Cm[(size_t)m * ldc + n] = v;
            }
        }
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// One time step of one layer. Workgroup = 8 hidden units (32 gate rows), wave g = gate g (i, f, g, o).
// pre: [N][4H] input projection (+ b_ih) of this step; h_prev: [N][H] or nullptr at t = 0 (zero state).
__global__ __launch_bounds__(256) void lstm_step_kernel(const float* __restrict__ pre, const float* __restrict__ w_hh,
                                                        const float* __restrict__ b_hh, const float* __restrict__ h_prev,
                                                        float* __restrict__ c, float* __restrict__ h_out, int N, int H) {
    extern __shared__ float sm[];
    float* hs = sm;                         // [N][H]
    float* gs = sm + (size_t)N * H;         // [4][8][LSTM_NMAX]
    const int j0 = blockIdx.x * 8;
    const int lane = threadIdx.x & 63, gate = threadIdx.x >> 6;
    if (h_prev) {
        for (int i = threadIdx.x; i < N * H; i += 256) hs[i] = h_prev[i];
        __syncthreads();
        for (int u = 0; u < 8; ++u) {
            const float* wr = w_hh + (size_t)(gate * H + j0 + u) * H;
            float acc[LSTM_NMAX];
#pragma unroll
            for (int n = 0; n < LSTM_NMAX; ++n) acc[n] = 0.f;
            for (int k = lane; k < H; k += 64) {
                const float w = wr[k];
#pragma unroll
                for (int n = 0; n < LSTM_NMAX; ++n)
                    if (n < N) acc[n] = fmaf(w, hs[n * H + k], acc[n]);
            }
#pragma unroll
            for (int n = 0; n < LSTM_NMAX; ++n) {
                float v = acc[n];
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
                if (lane == 0 && n < N) gs[(gate * 8 + u) * LSTM_NMAX + n] = v;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < 8 * LSTM_NMAX) {
        const int u = threadIdx.x / LSTM_NMAX, n = threadIdx.x % LSTM_NMAX;
        if (n < N) {
            const int j = j0 + u;
            float g4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // torch: (x W_ih^T + b_ih) + (h W_hh^T + b_hh)
                const float rec = (h_prev ? gs[(g * 8 + u) * LSTM_NMAX + n] : 0.f) + b_hh[g * H + j];
                g4[g] = pre[(size_t)n * 4 * H + g * H + j] + rec;
            }
            const float ig = sigmoidf(g4[0]), fg = sigmoidf(g4[1]), gg = tanhf(g4[2]), og = sigmoidf(g4[3]);
            const float c_prev = h_prev ? c[n * H + j] : 0.f;
            const float cn = fg * c_prev + ig * gg;
            c[n * H + j] = cn;
            h_out[n * H + j] = og * tanhf(cn);
        }
    }
}


// One LAYER in one launch (round 4): H / U workgroups (U = 4 hidden units each at H = 512: 128 workgroups; round 3 ran 64; 256
// were measured slower, see LSTM_UNITS_DEFAULT) stay resident for all time steps. A workgroup's 4 U rows of W_hh live in REGISTERS -- thread =
// (row, part), part strides the row in 16-byte pieces, 16 floats per thread at H = 512 -- and its c state too, so a step moves
// only h(t-1) in and U x N values out.
// Between the steps there is no barrier and no fence. Round 3's step (one atomic counter + agent-scope release / acquire
// fences around plain stores and loads) cost 20-27 us whatever the arithmetic: five dependent trips to memory, two of them
// whole-L2 maintenance. Here every h value travels as ONE 8-byte GRANULE {value, tag = t + 1} written by one write-through
// (sc1) store into a double-buffered [2][N][H] array; a consumer sweeps the N x H granules of step t - 1 with sc1 loads (up to
// sixteen in flight per thread) and simply re-reads a granule whose tag is not t yet: arrival and payload are the same 8 bytes, so
// nothing has to be ordered against anything (MI355X_MICROARCH.md, "allgather": ~3 us for this size). Two buffers suffice: who
// writes h(t + 1) has read all of h(t), which nobody could write before everybody had read h(t - 1). hseq gets the same
// values by plain stores for the next layer / the decoder (visible at the end of the launch). The grid must be CO-RESIDENT
// (every workgroup waits for every other's granules): the host launches it only when the device holds H / U such workgroups at
// once (hipOccupancyMaxActiveBlocksPerMultiprocessor x compute units) -- an ordinary launch; round 4 used
// hipLaunchCooperativeKernel for that check alone (no grid-wide sync is used), and the runtime's cooperative queue turned out
// to be what aborted at process exit under rocprofv3 (profiles/README.md, round 5). A granule that does not arrive within
// 20 ms sets *err: the decoder then writes NaN rows, pa_lstm_last_status reports it once and the handle falls back to one
// launch per step. force_timeout (PA_LSTM_FORCE_TIMEOUT=1, tests): workgroup 0 withholds its granules of step 0.
// MF (U == 4 only, H % 64 == 0): the recurrent product W_hh[16 rows of this workgroup][H] x h(t-1)[H][N <= 16] on the fp32 matrix cores --
// ONE v_mfma_f32_16x16x4_f32 tile (rows = the workgroup's 4 gates x 4 units, columns = the batch), K = H split over the four waves
// (H / 16 instructions each: 32 at H = 512), the four partial tiles summed by the gate threads. The row operand (a lane's 32
// weights) stays in registers for the whole launch as before; the column operand comes from the h tile in LDS with eight
// ds_read_b128 per step (row pitch H + 8 floats: the 16 lanes of a read phase cover 64 distinct banks). The vector form did these
// 16 x 16 x 512 multiply-adds as 512 dependent fmaf per thread plus four shuffle rounds: 4.6 us of a step's 8.6; this one ~0.9.
// fp32 throughout (the instruction is an fmaf chain); the summation order differs from the vector form's.
template <int U, bool MF = false>
__global__ __launch_bounds__(256) void lstm_layer_kernel(const float* __restrict__ pre_all, const float* __restrict__ w_hh,
                                                         const float* __restrict__ b_hh, float* __restrict__ hseq, int L, int N, int H,
                                                         unsigned long long* __restrict__ gran, int* __restrict__ err,
                                                         unsigned long long* __restrict__ dbg, int force_timeout) {
    constexpr int R = 4 * U, P = 256 / R, KQ = 128 / P;
    unsigned long long d_sweep = 0, d_prod = 0, d_gate = 0, d_t = 0;  // PA_LSTM_STAMP=1: where a step's time goes (100 MHz ticks)  // rows, lanes per row, float4 pieces per lane (H <= 512)
    static_assert(!MF || U == 4, "the matrix-core product is one 16-row tile");
    extern __shared__ float sm[];
    float* hs = sm;                              // [N][HP], HP = H (+ 8 with MF)
    const int HP = MF ? H + 8 : H;
    const int hshift = (H & (H - 1)) == 0 ? __ffs(H) - 1 : -1;
    float* gs = hs + (size_t)N * HP;             // [R][LSTM_NMAX]; MF: [4 waves][R][LSTM_NMAX] partial tiles
    const int j0 = blockIdx.x * U;
    const int row = threadIdx.x / P, part = threadIdx.x % P;   // row = gate * U + unit
    const int hq = H >> 2, nh = N * H;
    float4 wreg[KQ];
    // MF: lane (r = lane & 15, kq = lane >> 4) of wave w holds W[row r][k], k = w * (H / 4) + 16 j + 4 kq + i for j < H / 64, i < 4
    const int mf_lane = threadIdx.x & 63, mf_wave = threadIdx.x >> 6, mf_r = mf_lane & 15, mf_kq = mf_lane >> 4;
    if (MF) {
        const float* wr = w_hh + (size_t)((mf_r / U) * H + j0 + (mf_r % U)) * H + mf_wave * (H >> 2) + 4 * mf_kq;
#pragma unroll
        for (int q = 0; q < KQ; ++q) wreg[q] = q < (H >> 6) ? *reinterpret_cast<const float4*>(wr + 16 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        const float4* wr = reinterpret_cast<const float4*>(w_hh + (size_t)((row / U) * H + j0 + (row % U)) * H);
#pragma unroll
        for (int q = 0; q < KQ; ++q) wreg[q] = q * P + part < hq ? wr[q * P + part] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float c_reg = 0.f;  // thread (u, n) = threadIdx.x < U * LSTM_NMAX owns c[n][j0 + u]
    for (int t = 0; t < L; ++t) {
        const float* pre = pre_all + (size_t)t * N * 4 * H;
        if (t > 0) {
            // h(t - 1): sweep its granules; a tag that is not t yet means the producer has not stored it -- read it again
            const unsigned long long* g = gran + (size_t)((t - 1) & 1) * nh;
            int failed = 0;
            if (dbg) d_t = wall_clock64();
            for (int base = threadIdx.x; base < nh; base += 256 * 16) {
                // every pass re-reads ALL granules still missing at once (one trip to memory per pass, not one per granule)
                unsigned pending = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (base + k * 256 < nh) pending |= 1u << k;
                const unsigned long long t0 = wall_clock64();
                while (pending) {
                    unsigned long long v[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        if (pending & (1u << k)) v[k] = __hip_atomic_load(g + base + k * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        if ((pending & (1u << k)) && (unsigned)(v[k] >> 32) == (unsigned)t) {
                            const int gi = base + k * 256;   // granule n * H + j -> the h tile's row n (pitch HP)
                            hs[MF ? gi + (hshift >= 0 ? gi >> hshift : gi / H) * 8 : gi] = __uint_as_float((unsigned)v[k]);
                            pending &= ~(1u << k);
                        }
                    if (pending) {
                        if (wall_clock64() - t0 > 2000000ull || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                            __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            failed = 1;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
            }
            if (__syncthreads_or(failed)) return;
            if (dbg) { const unsigned long long n_ = wall_clock64(); d_sweep += n_ - d_t; d_t = n_; }
            if (MF) {
                typedef float mf4 __attribute__((ext_vector_type(4)));
                mf4 d4 = mf4{0.f, 0.f, 0.f, 0.f};
                const int nn = mf_r < N ? mf_r : N - 1;   // (columns past the batch read a valid row; their results are never used)
                const float* hb = hs + (size_t)nn * HP + mf_wave * (H >> 2) + 4 * mf_kq;
#pragma unroll
                for (int q = 0; q < KQ; ++q)
                    if (q < (H >> 6)) {
                        const float4 hv = *reinterpret_cast<const float4*>(hb + 16 * q);
                        const float4 w = wreg[q];
                        d4 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, hv.x, d4, 0, 0, 0);
                        d4 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, hv.y, d4, 0, 0, 0);
                        d4 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, hv.z, d4, 0, 0, 0);
                        d4 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, hv.w, d4, 0, 0, 0);
                    }
                // this wave's partial tile: rows 4 kq + i, column lane & 15
                float* gp = gs + mf_wave * (R * LSTM_NMAX);
                gp[(4 * mf_kq + 0) * LSTM_NMAX + mf_r] = d4.x;
                gp[(4 * mf_kq + 1) * LSTM_NMAX + mf_r] = d4.y;
                gp[(4 * mf_kq + 2) * LSTM_NMAX + mf_r] = d4.z;
                gp[(4 * mf_kq + 3) * LSTM_NMAX + mf_r] = d4.w;
            } else {
                float acc[LSTM_NMAX];
    #pragma unroll
                for (int n = 0; n < LSTM_NMAX; ++n) acc[n] = 0.f;
    #pragma unroll
                for (int q = 0; q < KQ; ++q) {
                    const int k4 = q * P + part;  // consecutive lanes read consecutive 16-byte pieces of h: no bank conflicts
                    if (k4 < hq) {
                        const float4 w = wreg[q];
    #pragma unroll
                        for (int n = 0; n < LSTM_NMAX; ++n)
                            if (n < N) {
                                const float4 hv = reinterpret_cast<const float4*>(hs + (size_t)n * H)[k4];
                                acc[n] = fmaf(w.x, hv.x, acc[n]);
                                acc[n] = fmaf(w.y, hv.y, acc[n]);
                                acc[n] = fmaf(w.z, hv.z, acc[n]);
                                acc[n] = fmaf(w.w, hv.w, acc[n]);
                            }
                    }
                }
    #pragma unroll
                for (int n = 0; n < LSTM_NMAX; ++n)
                    if (n < N) {
                        float v = acc[n];
    #pragma unroll
                        for (int d = 1; d < P; d <<= 1) v += __shfl_xor(v, d, 64);
                        if (part == 0) gs[row * LSTM_NMAX + n] = v;
                    }
            }
            __syncthreads();
            if (dbg) { const unsigned long long n_ = wall_clock64(); d_prod += n_ - d_t; d_t = n_; }
        }
        if (threadIdx.x < U * LSTM_NMAX) {
            const int u = threadIdx.x / LSTM_NMAX, n = threadIdx.x % LSTM_NMAX;
            if (n < N) {
                const int j = j0 + u;
                float g4[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float rsum = 0.f;
                    if (t > 0) {
                        rsum = gs[(g * U + u) * LSTM_NMAX + n];
                        if (MF) rsum = ((rsum + gs[R * LSTM_NMAX + (g * U + u) * LSTM_NMAX + n]) + gs[2 * R * LSTM_NMAX + (g * U + u) * LSTM_NMAX + n]) +
                                       gs[3 * R * LSTM_NMAX + (g * U + u) * LSTM_NMAX + n];
                    }
                    const float rec = rsum + b_hh[g * H + j];
                    g4[g] = pre[(size_t)n * 4 * H + g * H + j] + rec;
                }
                const float ig = sigmoidf(g4[0]), fg = sigmoidf(g4[1]), gg = tanhf(g4[2]), og = sigmoidf(g4[3]);
                const float cn = fg * (t > 0 ? c_reg : 0.f) + ig * gg;
                c_reg = cn;
                const float hv = og * tanhf(cn);
                hseq[(size_t)t * nh + n * H + j] = hv;
                if (!(force_timeout && blockIdx.x == 0 && t == 0))
                __hip_atomic_store(gran + (size_t)(t & 1) * nh + n * H + j, ((unsigned long long)(unsigned)(t + 1) << 32) | __float_as_uint(hv),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // (gs is read above and rewritten only behind the next step's first barrier; hs likewise)
        if (dbg && t > 0) d_gate += wall_clock64() - d_t;
    }
    if (dbg && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) { dbg[0] = d_sweep; dbg[1] = d_prod; dbg[2] = d_gate; dbg[3] = (unsigned long long)L; }
}

// action_decoder + log_softmax for one row: Linear(H,128) + ReLU -> Linear(128,A) -> log_softmax
__global__ __launch_bounds__(128) void lstm_decode_kernel(const float* __restrict__ h, const float* __restrict__ w1,
                                                          const float* __restrict__ b1, const float* __restrict__ w2,
                                                          const float* __restrict__ b2, float* __restrict__ logp, int H, int A,
                                                          const int* __restrict__ sync_words, int n_layers) {
    __shared__ float hid[128];
    __shared__ float lg[64];
    const int row = blockIdx.x, t = threadIdx.x;
    // a layer kernel whose grid barrier gave up left garbage behind: say so in the result instead of passing it on
    bool failed = false;
    for (int l = 0; l < n_layers; ++l) failed |= sync_words[2 * l + 1] != 0;
    if (failed) {
        if (t < A) logp[(size_t)row * A + t] = __builtin_nanf("");
        return;
    }
    const float* hr = h + (size_t)row * H;
    {
        const float* w = w1 + (size_t)t * H;
        float a = 0.f;
        for (int k = 0; k < H; ++k) a = fmaf(w[k], hr[k], a);
        a += b1[t];
        hid[t] = a > 0.f ? a : 0.f;
    }
    __syncthreads();
    if (t < A) {
        const float* w = w2 + (size_t)t * 128;
        float a = 0.f;
        for (int k = 0; k < 128; ++k) a = fmaf(w[k], hid[k], a);
        lg[t] = a + b2[t];
    }
    __syncthreads();
    if (t < A) {
        float mx = -INFINITY;
        for (int k = 0; k < A; ++k) mx = fmaxf(mx, lg[k]);
        float sum = 0.f;
        for (int k = 0; k < A; ++k) sum += expf(lg[k] - mx);
        logp[(size_t)row * A + t] = lg[t] - mx - logf(sum);
    }
}

}  // namespace

hipError_t launch_linear_f32(const float* X, int ld, const float* W, const float* bias, float* Cm, int ldc, int M, int N, int K, int relu,
                             hipStream_t s) {
    // Enough rows and whole tiles: the implicit-GEMM engine's fp32 MFMA kernel, the rows addressed as the pixels of one
    // 1 x M image under a 1x1 filter (the transformer's projections and 256 <-> 2048 feed-forward layers, the LSTM's
    // input projections). Anything else -- the class heads' 63 columns, a 300-wide K -- stays on the VALU tiles below.
    if (M >= 64 && N % 64 == 0 && K % 32 == 0 && (relu == 0 || relu == 1) && ld % 4 == 0 && ldc % 4 == 0) {
        GemmParams p;
        memset(&p, 0, sizeof(p));
        p.act = X;
        p.wgt = W;
        p.bias = bias;
        p.out = Cm;
        p.M = M;
        p.N = N;
        p.taps = 1; p.kw_taps = 1; p.chunk = K; p.ktot = K;
        p.howo = M; p.wo = M;
        p.in_px_stride = ld; p.in_row_stride = 0; p.in_img_stride = 0;
        p.stride = 1;
        p.out_px_stride = ldc; p.out_row_stride = 0; p.out_img_stride = 0;
        p.relu = relu;
        p.splitk = 1;
        const long long t128 = (long long)((M + 127) / 128) * (N / 64);
        return launch_igemm(p, t128 >= 512 ? TILE_128x64 : TILE_64x64, s);
    }
    hipLaunchKernelGGL(linear_f32_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, s, X, ld, W, bias, Cm, ldc, M, N, K, relu);
    return hipGetLastError();
}

}  // namespace pa

struct pa_lstm {
    int in_dim = 0, hid = 0, layers = 0, actions = 0, max_rows = 0, device = 0;
    std::vector<float*> w_ih, w_hh, b_ih, b_hh;
    float *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr;
    float *weights = nullptr;            // one allocation behind all of the above
    float *pre = nullptr, *hseq[2] = {nullptr, nullptr}, *c = nullptr;
    int* sync_words = nullptr;            // [2 * layers]: per layer an unused word and the error word of its launch
    unsigned long long* gran = nullptr;   // [2][LSTM_NMAX][512] h granules {value, tag} of the layer in flight
    unsigned long long* dbg = nullptr;    // PA_LSTM_STAMP=1: in-kernel clock sums of the last layer launch
    int* sync_host = nullptr;             // pinned copy of the error words
    hipEvent_t sync_ev = nullptr;         // behind the last forward's copies into sync_host: the host reads the words only once it is done
    bool sync_pending = false;            // a forward has been enqueued since the words were last read
    int occ_blocks = -1, occ_key = -1;    // co-resident lstm_layer_kernel workgroups the device holds for batch * 16 + U == occ_key
    bool persistent = true;               // one launch per layer (lstm_layer_kernel); false after a barrier timeout
    std::string last_error;
};

namespace {
size_t lstm_float_count(int in_dim, int hid, int layers, int actions) {
    size_t n = 0;
    for (int l = 0; l < layers; ++l) n += (size_t)4 * hid * (l == 0 ? in_dim : hid) + (size_t)4 * hid * hid + 8 * (size_t)hid;
    n += (size_t)128 * hid + 128 + (size_t)actions * 128 + actions;
    return n;
}
}  // namespace

extern "C" {

size_t pa_lstm_blob_bytes(int32_t input_dim, int32_t hidden_dim, int32_t num_layers, int32_t num_actions) {
    return 8 * sizeof(int32_t) + lstm_float_count(input_dim, hidden_dim, num_layers, num_actions) * sizeof(float);
}

const char* pa_lstm_last_error(const pa_lstm* h) { return h ? h->last_error.c_str() : "null handle"; }

int pa_lstm_create(int32_t device, int32_t input_dim, int32_t hidden_dim, int32_t num_layers, int32_t num_actions, int32_t max_rows,
                   const void* blob_host, size_t blob_bytes, pa_lstm** out) {
    if (!out) return PA_ERR_INVALID_ARG;
    *out = nullptr;
    if (!blob_host || input_dim < 1 || hidden_dim < 8 || hidden_dim % 8 != 0 || hidden_dim > 512 || num_layers < 1 || num_layers > 8 ||
        num_actions < 1 || num_actions > 64 || max_rows < 1)
        return PA_ERR_INVALID_ARG;
    const int32_t* hdr = reinterpret_cast<const int32_t*>(blob_host);
    if (blob_bytes != pa_lstm_blob_bytes(input_dim, hidden_dim, num_layers, num_actions) || hdr[0] != PA_LSTM_MAGIC || hdr[1] != 1 ||
        hdr[2] != input_dim || hdr[3] != hidden_dim || hdr[4] != num_layers || hdr[5] != num_actions)
        return PA_ERR_BAD_WEIGHTS;
    pa_lstm* h = new pa_lstm();
    *out = h;  // handed back on failure too (pa_lstm_last_error, then pa_lstm_destroy)
    h->in_dim = input_dim; h->hid = hidden_dim; h->layers = num_layers; h->actions = num_actions; h->max_rows = max_rows;
    h->device = device;
    auto chk = [&](hipError_t e, const char* what) -> bool {
        if (e == hipSuccess) return true;
        h->last_error = std::string(what) + ": " + hipGetErrorString(e);
        return false;
    };
    if (!chk(hipSetDevice(device), "hipSetDevice")) return PA_ERR_NO_DEVICE;
    const size_t nw = lstm_float_count(input_dim, hidden_dim, num_layers, num_actions);
    const size_t H = hidden_dim;
    if (!chk(hipMalloc(&h->weights, nw * sizeof(float)), "hipMalloc weights")) return PA_ERR_HIP;
    if (!chk(hipMemcpy(h->weights, hdr + 8, nw * sizeof(float), hipMemcpyHostToDevice), "upload weights")) return PA_ERR_HIP;
    float* p = h->weights;
    for (int l = 0; l < num_layers; ++l) {
        const size_t in_l = l == 0 ? input_dim : hidden_dim;
        h->w_ih.push_back(p); p += 4 * H * in_l;
        h->w_hh.push_back(p); p += 4 * H * H;
        h->b_ih.push_back(p); p += 4 * H;
        h->b_hh.push_back(p); p += 4 * H;
    }
    h->w1 = p; p += 128 * H;
    h->b1 = p; p += 128;
    h->w2 = p; p += (size_t)num_actions * 128;
    h->b2 = p;
    if (!chk(hipMalloc(&h->pre, (size_t)max_rows * 4 * H * sizeof(float)), "hipMalloc pre")) return PA_ERR_HIP;
    for (int i = 0; i < 2; ++i)
        if (!chk(hipMalloc(&h->hseq[i], (size_t)max_rows * H * sizeof(float)), "hipMalloc h")) return PA_ERR_HIP;
    if (!chk(hipMalloc(&h->c, (size_t)pa::LSTM_NMAX * H * sizeof(float)), "hipMalloc c")) return PA_ERR_HIP;
    if (!chk(hipMalloc(&h->sync_words, 2 * 8 * sizeof(int)), "hipMalloc barrier words")) return PA_ERR_HIP;
    if (!chk(hipMalloc(&h->gran, (size_t)2 * pa::LSTM_NMAX * 512 * sizeof(unsigned long long)), "hipMalloc granules")) return PA_ERR_HIP;
    if (getenv("PA_LSTM_STAMP") && !chk(hipMalloc(&h->dbg, 4 * sizeof(unsigned long long)), "hipMalloc stamps")) return PA_ERR_HIP;
    if (!chk(hipHostMalloc(&h->sync_host, 2 * 8 * sizeof(int)), "hipHostMalloc")) return PA_ERR_HIP;
    memset(h->sync_host, 0, 2 * 8 * sizeof(int));
    if (!chk(hipEventCreateWithFlags(&h->sync_ev, hipEventDisableTiming), "hipEventCreate")) return PA_ERR_HIP;
    // one launch per layer needs H / U workgroups co-resident (U = 2 above 256 hidden units): PA_LSTM_STEPS=1 forces the
    // per-step kernels (A/B); a grid the device cannot hold at once switches to them by itself (pa_lstm_forward)
    if (getenv("PA_LSTM_STEPS")) h->persistent = false;
    return PA_OK;
}

// The error words of the forwards enqueued so far: waits for the copies behind the last forward (so: for that forward) before
// reading the pinned words -- a status query is a synchronisation point of the handle's last stream.
int pa_lstm_last_status(pa_lstm* h) {
    if (!h) return PA_ERR_INVALID_ARG;
    if (h->sync_pending && h->sync_ev) {
        (void)hipEventSynchronize(h->sync_ev);
        h->sync_pending = false;
    }
    bool failed = false;
    for (int l = 0; l < h->layers; ++l)
        if (h->sync_host && h->sync_host[2 * l + 1]) failed = true;
    if (!failed) return PA_OK;
    // reported ONCE: the handle leaves the per-layer path for good, so the words are cleared and later calls are valid
    h->persistent = false;
    memset(h->sync_host, 0, 2 * 8 * sizeof(int));
    h->last_error = "pa_lstm_forward: the grid barrier of the per-layer kernel timed out (its workgroups were not all resident); "
                    "that call's log-probabilities are NaN, later calls launch one kernel per time step";
    return PA_ERR_HIP;
}

void pa_lstm_destroy(pa_lstm* h) {
    if (!h) return;
    (void)hipFree(h->weights);
    (void)hipFree(h->pre);
    (void)hipFree(h->hseq[0]);
    (void)hipFree(h->hseq[1]);
    (void)hipFree(h->c);
    (void)hipFree(h->sync_words);
    (void)hipFree(h->gran);
    (void)hipFree(h->dbg);
    // (the pinned words may still be the target of a copy in flight: wait for it before they go)
    if (h->sync_ev) {
        if (h->sync_pending) (void)hipEventSynchronize(h->sync_ev);
        (void)hipEventDestroy(h->sync_ev);
    }
    if (h->sync_host) (void)hipHostFree(h->sync_host);
    delete h;
}

int pa_lstm_forward(pa_lstm* h, const float* x, int32_t ld, int32_t seq_len, int32_t batch, float* logp, void* stream) {
    if (!h) return PA_ERR_INVALID_ARG;
    auto bad = [&](int code, const char* msg) { h->last_error = msg; return code; };
    if (!x || !logp || seq_len < 1 || batch < 1 || ld < h->in_dim) return bad(PA_ERR_INVALID_ARG, "pa_lstm_forward: bad argument");
    if (batch > pa::LSTM_NMAX) return bad(PA_ERR_CAPACITY, "pa_lstm_forward: more than 16 rows per time step");
    const long long rows = (long long)seq_len * batch;
    if (rows > h->max_rows) return bad(PA_ERR_CAPACITY, "pa_lstm_forward: seq_len * batch exceeds max_rows");
    hipStream_t s = (hipStream_t)stream;
    const int H = h->hid, M = (int)rows;
    const size_t step_lds = ((size_t)batch * H + 4 * 8 * pa::LSTM_NMAX) * sizeof(float);
    (void)hipMemsetAsync(h->sync_words, 0, 2 * 8 * sizeof(int), s);  // barrier counters and error words of this call (the decoder reads the latter)
    for (int l = 0; l < h->layers; ++l) {
        const float* in = l == 0 ? x : h->hseq[(l - 1) & 1];
        const int in_ld = l == 0 ? ld : H, K = l == 0 ? h->in_dim : H;
        (void)pa::launch_linear_f32(in, in_ld, h->w_ih[l], h->b_ih[l], h->pre, 4 * H, M, 4 * H, K, 0, s);
        float* hs = h->hseq[l & 1];
        // (a timeout of an EARLIER forward the caller has not asked about yet -- read only once that forward's copies are done)
        if (h->persistent && l == 0 && h->sync_pending && hipEventQuery(h->sync_ev) == hipSuccess) {
            for (int q = 0; q < h->layers; ++q)
                if (h->sync_host[2 * q + 1]) h->persistent = false;
        }
        if (h->persistent) {
            // all time steps of the layer in one launch whose workgroups must be co-resident
            static const int u_env = getenv("PA_LSTM_UNITS") ? atoi(getenv("PA_LSTM_UNITS")) : 0;   // tuning: hidden units per workgroup
            int U = u_env == 1 || u_env == 2 || u_env == 4 || u_env == 8 ? u_env : pa::LSTM_UNITS_DEFAULT;
            while (U > 1 && H % U) U >>= 1;
            // the recurrent product on the matrix cores where its tile shape fits (PA_LSTM_MFMA=0: the vector form, A/B)
            static const int mfma_env = getenv("PA_LSTM_MFMA") ? atoi(getenv("PA_LSTM_MFMA")) : 1;
            const bool mf = mfma_env && U == 4 && H % 64 == 0;
            const size_t lds = mf ? ((size_t)batch * (H + 8) + (size_t)4 * 4 * U * pa::LSTM_NMAX) * sizeof(float)
                                  : ((size_t)batch * H + (size_t)4 * U * pa::LSTM_NMAX) * sizeof(float);
            const float* a_pre = h->pre; const float* a_w = h->w_hh[l]; const float* a_b = h->b_hh[l];
            int a_L = seq_len, a_N = batch, a_H = H;
            unsigned long long* a_gran = h->gran; int* a_err = h->sync_words + 2 * l + 1;
            unsigned long long* a_dbg = h->dbg;
            static const int force_to = getenv("PA_LSTM_FORCE_TIMEOUT") ? atoi(getenv("PA_LSTM_FORCE_TIMEOUT")) : 0;
            int a_force = force_to;
            void* args[] = {&a_pre, &a_w, &a_b, &hs, &a_L, &a_N, &a_H, &a_gran, &a_err, &a_dbg, &a_force};
            (void)hipMemsetAsync(h->gran, 0, (size_t)2 * batch * H * sizeof(unsigned long long), s);   // tag 0 = not written
            const void* fn = U == 8   ? reinterpret_cast<const void*>(&pa::lstm_layer_kernel<8>)
                             : mf     ? reinterpret_cast<const void*>(&pa::lstm_layer_kernel<4, true>)
                             : U == 4 ? reinterpret_cast<const void*>(&pa::lstm_layer_kernel<4>)
                             : U == 2 ? reinterpret_cast<const void*>(&pa::lstm_layer_kernel<2>)
                                      : reinterpret_cast<const void*>(&pa::lstm_layer_kernel<1>);
            // an ordinary launch, behind the check a cooperative launch would make: the device must hold the grid at once (the
            // kernel only needs co-residency, not the runtime's grid sync; the cooperative QUEUE the runtime creates for
            // hipLaunchCooperativeKernel was the one thing the rnn workload had that the others did not when it aborted
            // inside exit() under rocprofv3)
            if (h->occ_key != batch * 32 + U * 2 + (mf ? 1 : 0)) {  // (the kernel's LDS grows with the batch)
                h->occ_key = batch * 32 + U * 2 + (mf ? 1 : 0);
                int per_cu = 0, cus = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds) != hipSuccess) per_cu = 0;
                if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess) cus = 0;
                (void)hipGetLastError();
                h->occ_blocks = per_cu * cus;
            }
            // HEADROOM: the occupancy query describes an EMPTY device; kernels of other streams (lanes, the chain's
            // stages) hold slots of their own when this grid is dispatched. The grid may therefore use at most HALF of what the
            // empty device holds (H = 512: 128 workgroups against 256 CUs x >= 2), so that it fits beside a neighbour that leaves
            // every CU half free; a neighbour that fills the chip for longer than the kernel's 20 ms patience still ends in the
            // timeout path below -- NaN rows and PA_ERR_HIP from pa_lstm_last_status, never silent garbage
            // (tests/test_rnn_detector.py::test_lstm_forward_beside_a_busy_stream_is_correct_or_reported).
            const hipError_t ce = 2 * (H / U) <= h->occ_blocks ? hipLaunchKernel(fn, dim3(H / U), dim3(256), args, lds, s) : hipErrorCooperativeLaunchTooLarge;
            if (ce == hipSuccess) {
                (void)hipMemcpyAsync(h->sync_host + 2 * l, h->sync_words + 2 * l, 2 * sizeof(int), hipMemcpyDeviceToHost, s);
                continue;
            }
            (void)hipGetLastError();   // the grid cannot be co-resident on this device (or no cooperative launches): per-step kernels
            h->persistent = false;
        }
        for (int t = 0; t < seq_len; ++t)
            hipLaunchKernelGGL(pa::lstm_step_kernel, dim3(H / 8), dim3(256), step_lds, s, h->pre + (size_t)t * batch * 4 * H, h->w_hh[l],
                               h->b_hh[l], t ? hs + (size_t)(t - 1) * batch * H : nullptr, h->c, hs + (size_t)t * batch * H, batch, H);
    }
    hipLaunchKernelGGL(pa::lstm_decode_kernel, dim3(M), dim3(128), 0, s, h->hseq[(h->layers - 1) & 1], h->w1, h->b1, h->w2, h->b2, logp, H,
                       h->actions, h->sync_words, h->layers);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return bad(PA_ERR_HIP, hipGetErrorString(e));
    (void)hipEventRecord(h->sync_ev, s);
    h->sync_pending = true;
    if (h->dbg) {   // (measurement aid: synchronises)
        unsigned long long d[4];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(d, h->dbg, sizeof d, hipMemcpyDeviceToHost);
        static int printed = 0;
        if (printed++ < 3 && d[3] > 1)
            fprintf(stderr, "[lstm stamps] per step of the last layer: sweep %.2f us, product %.2f us, gates + stores %.2f us\n",
                    d[0] * 0.01 / (d[3] - 1), d[1] * 0.01 / (d[3] - 1), d[2] * 0.01 / (d[3] - 1));
    }
    return PA_OK;
}

}  // extern "C"
