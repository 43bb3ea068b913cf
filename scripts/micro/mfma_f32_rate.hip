// Issue rate of the fp32 matrix instructions as the Winograd kernel uses them: two waves per SIMD, each cycling through 32
// accumulator quads (v_mfma_f32_16x16x4_f32) or 8 accumulator blocks (v_mfma_f32_32x32x2_f32), operands in registers, nothing
// else in the loop. Prints s_memtime ticks per instruction and SIMD, and the wall-clock rate.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_f32_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k16(float* out, unsigned long long* clk, int iters, float a0, float b0) {
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            acc[2 * p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[2 * p], 0, 0, 0);
            acc[2 * p + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[2 * p + 1], 0, 0, 0);
            acc[2 * p] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, acc[2 * p], 0, 0, 0);
            acc[2 * p + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, acc[2 * p + 1], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * WAVES + threadIdx.x / 64] = t1 - t0;
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k32(float* out, unsigned long long* clk, int iters, float a0, float b0) {
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * WAVES + threadIdx.x / 64] = t1 - t0;
}

template <typename K>
void run(const char* name, K kern, int waves, int per_iter, double flop_per_inst, int grid) {
    float* out; unsigned long long* clk;
    hipMalloc(&out, (size_t)grid * waves * 64 * 4);
    hipMalloc(&clk, (size_t)grid * waves * 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * waves), 0, 0, out, clk, iters, 1.0f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[64];
    hipMemcpy(h, clk, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost);
    const double inst_per_wave = (double)iters * per_iter;
    const double ticks_per_inst_simd = (double)h[0] / inst_per_wave / (waves / 4.0);   // waves / 4 waves share a SIMD
    const double tflops = (double)grid * waves * inst_per_wave * flop_per_inst / (ms * 1e-3) / 1e12;
    printf("%-34s grid %4d x %d waves: %.2f s_memtime ticks per instruction and SIMD, %.3f ms, %.1f TFLOP/s, ticks per us %.0f\n", name, grid, waves,
           ticks_per_inst_simd, ms, tflops, (double)h[0] / (ms * 1e3));
    hipFree(out); hipFree(clk);
}

int main() {
    for (int grid : {256, 512}) {
        run("v_mfma_f32_16x16x4_f32, 8 waves", k16<8>, 8, 64, 2048.0, grid);
        run("v_mfma_f32_16x16x4_f32, 4 waves", k16<4>, 4, 64, 2048.0, grid);
        run("v_mfma_f32_32x32x2_f32, 8 waves", k32<8>, 8, 32, 4096.0, grid);
        run("v_mfma_f32_32x32x2_f32, 4 waves", k32<4>, 4, 32, 4096.0, grid);
    }
    return 0;
}
