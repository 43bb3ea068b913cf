// Pure v_mfma_f32_32x32x2_f32 issue-rate probe: what the chip sustains with no memory traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, bool RANDOM>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, unsigned long long* clk) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    // pseudo-random operands (full-range mantissas) so the datapath toggles like real data
    unsigned h = (threadIdx.x + blockIdx.x * 977u) * 2654435761u;
    float xs[8], ys[8];
    for (int j = 0; j < 8; ++j) { h = h * 1664525u + 1013904223u; xs[j] = (int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f; h = h * 1664525u + 1013904223u; ys[j] = (int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f; }
    if (!RANDOM) for (int j = 0; j < 8; ++j) { xs[j] = 0.5f; ys[j] = 0.25f; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[(2 * a) & 7], ys[(2 * a) & 7], acc[a], 0, 0, 0);
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[(2 * a + 1) & 7], ys[(2 * a + 1) & 7], acc[a], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}
int main() {
    float* out; unsigned long long* clk;
    const int grid_max = 256 * 8;
    hipMalloc(&out, grid_max * 256 * 4); hipMalloc(&clk, grid_max * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rnd = 0; rnd < 2; ++rnd)
    for (int wpc : {1, 2}) {           // workgroups per CU
        const int grid = 256 * wpc, iters = 10000;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (rnd) hipLaunchKernelGGL((mfma_loop<1, true>), dim3(grid), dim3(256), 0, 0, out, iters * 4, clk);
            else hipLaunchKernelGGL((mfma_loop<2, true>), dim3(grid), dim3(256), 0, 0, out, iters * 2, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(grid * 2); hipMemcpy(h.data(), clk, grid * 16, hipMemcpyDeviceToHost);
            double flops = (double)grid * 4 * iters * 4 * 2 * 32 * 32 * 2 * 2;
            double ghz = (double)h[0] / (double)h[1] * 0.1;
            printf("chains=%d wg/CU %d: %.3f ms  %.1f TF  in-kernel clock %.3f GHz (cycles %llu)\n", rnd ? 1 : 2, wpc, ms, flops / ms / 1e9, ghz, h[0]);
        }
    }
    return 0;
}
