// What an fp32 matrix instruction shares its SIMD with: 32 x v_mfma_f32_32x32x2_f32 per loop iteration (2048 issue cycles) with,
// spread evenly between them, R LDS reads (ds_read_b128 or ds_read_b64, results feed the next iteration's operands), V full-rate
// vector instructions (v_fma_f32) and T transcendental ones (v_exp_f32). Two waves per SIMD (workgroups of four waves, two per
// CU), like the GEMM kernels. Prints SIMD cycles per iteration and wave -- 2048 when the extra work is free. A second section does
// the same around v_mfma_f32_32x32x16_bf16 (1024 cycles per iteration).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_f32_mix.hip -o /tmp/mfma_mix && /tmp/mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) volatile f32x4 lds_v4;
typedef __attribute__((address_space(3))) volatile f32x2 lds_v2;

// R reads of WIDTH floats (4: b128, 2: b64), V v_fma_f32, T v_exp_f32 per 32 matrix instructions. Every instruction of the loop is
// a volatile asm statement, so the order written here is the order issued; reads of one half iteration feed the operands of
// the next, behind one s_waitcnt.
template <int R, int WIDTH, int V, int T>
__global__ __launch_bounds__(256, 2) void mix(float* out, unsigned long long* clk, int iters, float a0, float b0) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 64 * 32];   // a 128-byte row per lane and wave, chunks swizzled
    for (int i = threadIdx.x; i < 4 * 64 * 32; i += 256) lds[i] = a0 + i * 1e-6f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned row = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(lds + (wave * 64 + lane) * 32);
    const int swz = (lane >> 1) & 7;
    unsigned addr[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) addr[r] = row + ((r ^ swz) * 16);
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    f32x4 opnd[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) opnd[s][i] = f32x4{a0, b0, a0, b0};
    float va[4] = {a0, b0, a0 + 1.f, b0 + 1.f};
    float ta[4] = {a0, b0, a0 * 0.5f, b0 * 0.5f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int cur = half, nxt = half ^ 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 32; ++j) {
#pragma unroll
                for (int r = (j * R) / 32; r < ((j + 1) * R) / 32; ++r) {
                    if (WIDTH == 4) {
                        asm volatile("ds_read_b128 %0, %1" : "=v"(opnd[nxt][r & 7]) : "v"(addr[r & 7]) : "memory");
                    } else {
                        f32x2 v;
                        asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr[r & 7]) : "memory");
                        opnd[nxt][r & 7].x = v.x;
                        opnd[nxt][r & 7].y = v.y;
                    }
                }
#pragma unroll
                for (int q = (j * V) / 32; q < ((j + 1) * V) / 32; ++q) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(va[q & 3]) : "v"(a0), "v"(b0));
#pragma unroll
                for (int q = (j * T) / 32; q < ((j + 1) * T) / 32; ++q) asm volatile("v_exp_f32 %0, %0" : "+v"(ta[q & 3]));
                const f32x4 o = opnd[cur][j & 7];
                const float a = (j & 8) ? o.z : o.x, b = (j & 8) ? o.w : o.y;
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[j & 3]) : "v"(a), "v"(b));
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = va[0] + va[1] + va[2] + va[3] + ta[0] + ta[1] + ta[2] + ta[3];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int i = 0; i < 8; ++i) s += opnd[s2][i].x + opnd[s2][i].z;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) clk[blockIdx.x * 4 + wave] = t1 - t0;
}

// The same loop around 32 x v_mfma_f32_32x32x16_bf16 (8 passes: 1024 issue cycles per iteration at full rate): does vector work run
// BESIDE the bf16 matrix instruction, which has a pipe of its own?
template <int R, int V, int T>
__global__ __launch_bounds__(256, 2) void mixbf(float* out, unsigned long long* clk, int iters, float a0, float b0) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 64 * 32];
    for (int i = threadIdx.x; i < 4 * 64 * 32; i += 256) lds[i] = a0 + i * 1e-6f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned row = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(lds + (wave * 64 + lane) * 32);
    const int swz = (lane >> 1) & 7;
    unsigned addr[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) addr[r] = row + ((r ^ swz) * 16);
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    f32x4 opnd[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) opnd[s][i] = f32x4{a0, b0, a0, b0};
    float va[4] = {a0, b0, a0 + 1.f, b0 + 1.f};
    float ta[4] = {a0, b0, a0 * 0.5f, b0 * 0.5f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int cur = half, nxt = half ^ 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 32; ++j) {
#pragma unroll
                for (int r = (j * R) / 32; r < ((j + 1) * R) / 32; ++r)
                    asm volatile("ds_read_b128 %0, %1" : "=v"(opnd[nxt][r & 7]) : "v"(addr[r & 7]) : "memory");
#pragma unroll
                for (int q = (j * V) / 32; q < ((j + 1) * V) / 32; ++q) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(va[q & 3]) : "v"(a0), "v"(b0));
#pragma unroll
                for (int q = (j * T) / 32; q < ((j + 1) * T) / 32; ++q) asm volatile("v_exp_f32 %0, %0" : "+v"(ta[q & 3]));
                // (eight bf16 values per operand = one 128-bit register quad; the bit patterns do not matter for the timing)
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j & 3]) : "v"(opnd[cur][j & 7]), "v"(opnd[cur][(j + 1) & 7]));
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = va[0] + va[1] + va[2] + va[3] + ta[0] + ta[1] + ta[2] + ta[3];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int i = 0; i < 8; ++i) s += opnd[s2][i].x + opnd[s2][i].z;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) clk[blockIdx.x * 4 + wave] = t1 - t0;
}

// Round 6: the Winograd kernel's own instruction, v_mfma_f32_16x16x4_f32 -- 64 per loop iteration (2048
// issue cycles, as one chunk of wino.hip) -- with the chunk's input transform beside it either as PK packed additions
// (v_pk_add_f32: two floats per instruction, what hipcc emits for the kernel's f32x2 arithmetic) or as V scalar ones (v_add_f32).
// 32 packed = 64 scalar additions = one chunk's transform.
template <int V, int PK>
__global__ __launch_bounds__(256, 2) void mix16(float* out, unsigned long long* clk, int iters, float a0, float b0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float va[8];
    f32x2 pa[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { va[i] = a0 + i; pa[i] = f32x2{a0 + i, b0 - i}; }
    f32x2 pb = f32x2{b0, a0};
    asm volatile("" : "+v"(pb));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
#pragma unroll
            for (int q = (j * V) / 64; q < ((j + 1) * V) / 64; ++q) asm volatile("v_add_f32 %0, %0, %1" : "+v"(va[q & 7]) : "v"(b0));
#pragma unroll
            for (int q = (j * PK) / 64; q < ((j + 1) * PK) / 64; ++q) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pa[q & 7]) : "v"(pb));
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[j & 7]) : "v"(a0), "v"(b0));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += va[i] + pa[i].x + pa[i].y + acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) clk[blockIdx.x * 4 + wave] = t1 - t0;
}

static double ticks_per_cycle = 0.0;   // s_memtime ticks per SIMD cycle, from the bare loop (2048 cycles per iteration and wave, two waves)

static double base_cycles = 2048.0;     // matrix issue cycles of one iteration and wave (bf16 section: 1024)
static double flop_per_mfma = 4096.0;   // (bf16 32x32x16: 32768)

template <typename K>
void run(const char* name, K kern) {
    const int grid = 512, iters = 1000;
    float* out; unsigned long long* clk;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    hipMalloc(&clk, (size_t)grid * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, clk, iters, 1.0f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[4];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const double ticks_iter = (double)h[0] / iters;                 // of one wave; its SIMD ran two waves' iterations in that time
    if (ticks_per_cycle == 0.0) ticks_per_cycle = ticks_iter / 4096.0;
    const double cyc = ticks_iter / ticks_per_cycle / 2.0;
    const double tflops = (double)grid * 4 * iters * 32 * flop_per_mfma / (ms * 1e-3) / 1e12;
    printf("%-46s %7.0f SIMD cycles per iteration and wave (matrix: %.0f), +%5.0f, %.3f ms, %.1f TFLOP/s\n", name, cyc, base_cycles, cyc - base_cycles, ms, tflops);
    hipFree(out); hipFree(clk);
}

int main() {
    run("bare", mix<0, 4, 0, 0>);
    run("4 ds_read_b128", mix<4, 4, 0, 0>);
    run("8 ds_read_b128", mix<8, 4, 0, 0>);
    run("12 ds_read_b128 (128 x 64 tile, 2 x 2 waves)", mix<12, 4, 0, 0>);
    run("16 ds_read_b128", mix<16, 4, 0, 0>);
    run("32 ds_read_b128", mix<32, 4, 0, 0>);
    run("8 ds_read_b64", mix<8, 2, 0, 0>);
    run("16 ds_read_b64", mix<16, 2, 0, 0>);
    run("32 ds_read_b64", mix<32, 2, 0, 0>);
    run("16 v_fma_f32", mix<0, 4, 16, 0>);
    run("32 v_fma_f32", mix<0, 4, 32, 0>);
    run("64 v_fma_f32", mix<0, 4, 64, 0>);
    run("128 v_fma_f32", mix<0, 4, 128, 0>);
    run("8 v_exp_f32", mix<0, 4, 0, 8>);
    run("16 v_exp_f32", mix<0, 4, 0, 16>);
    run("32 v_exp_f32", mix<0, 4, 0, 32>);
    run("64 v_exp_f32", mix<0, 4, 0, 64>);
    run("12 b128 + 32 fma", mix<12, 4, 32, 0>);
    run("12 b128 + 32 fma + 16 exp", mix<12, 4, 32, 16>);
    printf("--- v_mfma_f32_32x32x16_bf16 (the cycle figure keeps the fp32 section's clock calibration)\n");
    base_cycles = 1024.0;
    flop_per_mfma = 32768.0;
    run("bf16 bare", mixbf<0, 0, 0>);
    run("bf16 + 16 ds_read_b128", mixbf<16, 0, 0>);
    run("bf16 + 32 ds_read_b128", mixbf<32, 0, 0>);
    run("bf16 + 32 v_fma_f32", mixbf<0, 32, 0>);
    run("bf16 + 64 v_fma_f32", mixbf<0, 64, 0>);
    run("bf16 + 128 v_fma_f32", mixbf<0, 128, 0>);
    run("bf16 + 256 v_fma_f32", mixbf<0, 256, 0>);
    run("bf16 + 32 v_exp_f32", mixbf<0, 0, 32>);
    run("bf16 + 64 v_exp_f32", mixbf<0, 0, 64>);
    run("bf16 + 32 b128 + 128 fma + 32 exp", mixbf<32, 128, 32>);
    printf("--- v_mfma_f32_16x16x4_f32 x 64 per iteration (wino.hip's chunk): packed against scalar additions beside it\n");
    base_cycles = 2048.0;
    flop_per_mfma = 2048.0;
    run("16x16x4 bare", mix16<0, 0>);
    run("16x16x4 + 32 v_pk_add_f32 (a chunk's transform)", mix16<0, 32>);
    run("16x16x4 + 64 v_add_f32 (the same, scalar)", mix16<64, 0>);
    run("16x16x4 + 64 v_pk_add_f32", mix16<0, 64>);
    run("16x16x4 + 128 v_add_f32", mix16<128, 0>);
    return 0;
}
