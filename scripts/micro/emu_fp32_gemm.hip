// PROTOTYPE, measurement only: an fp32 GEMM C[M][N] = A[M][K] x B[N][K]^T whose products run on
// the bf16 matrix cores -- "emulated fp32". Every fp32 operand is split into three bf16 slices (each the bf16 rounding of what the slices
// before it left); the six leading cross products a_i b_j, i + j <= 2, are six v_mfma_f32_32x32x16_bf16 accumulating in fp32 (a bf16 x bf16
// product is exact in fp32). The weights B are split once on the host (three bf16 planes); the activations A stay fp32 in memory and in LDS and
// are split IN REGISTERS behind their LDS read: v_cvt_pk_bf16_f32 + shift / mask / v_pk_add_f32 per pair of values -- vector work that runs
// beside the bf16 matrix instructions for free (scripts/micro/mfma_f32_mix.hip). Beside it the same tile loop on the exact fp32 matrix
// instruction (v_mfma_f32_32x32x2_f32) for reference. Prints time, fp32-equivalent TFLOP/s and the error of both against a float64 product.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/emu_fp32_gemm.hip -o /tmp/emu && /tmp/emu
// Plain global -> register -> LDS double buffering (no LDS-DMA, no persistence): a prototype of the inner loop, not a product kernel.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 64, BK = 32;
constexpr int AP = BK + 4;        // fp32 row pitch of the A image in LDS (floats): 144 bytes -> the 32 rows of a read land on distinct banks
constexpr int BP = BK + 8;        // bf16 row pitch of a B slice in LDS (elements): 80 bytes

// bf16 pair (lo = first value) of two floats, round to nearest even
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// eight fp32 values -> three bf16x8 operands (slices 0, 1, 2)
__device__ __forceinline__ void split8(const f32x4 lo, const f32x4 hi, u32x4& s0, u32x4& s1, u32x4& s2) {
    float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned q0 = cvt_pk_bf16(x[2 * p], x[2 * p + 1]);
        f32x2 r = f32x2{x[2 * p], x[2 * p + 1]} - f32x2{__uint_as_float(q0 << 16), __uint_as_float(q0 & 0xffff0000u)};
        const unsigned q1 = cvt_pk_bf16(r.x, r.y);
        r -= f32x2{__uint_as_float(q1 << 16), __uint_as_float(q1 & 0xffff0000u)};
        const unsigned q2 = cvt_pk_bf16(r.x, r.y);
        s0[p] = q0; s1[p] = q1; s2[p] = q2;
    }
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// EMU: products on the bf16 pipe (B = three pre-split bf16 planes [3][N][K]); else the exact fp32 instruction (B = fp32 [N][K])
template <bool EMU>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const float* __restrict__ A, const void* __restrict__ Bv, float* __restrict__ C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) float As[2][BM * AP];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[2][EMU ? 3 * BN * BP : 2 * BN * AP];   // (fp32 form: BN x AP floats)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;                    // 2 x 2 waves: 64 x 32 per wave
    const int tile_n = blockIdx.x % (N / BN), tile_m = blockIdx.x / (N / BN);
    const float* Ag = A + (size_t)tile_m * BM * K;
    const uint16_t* Bh = reinterpret_cast<const uint16_t*>(Bv);
    const float* Bf = reinterpret_cast<const float*>(Bv);
    // global -> registers: A 128 x 32 floats = 1024 float4 (4 per thread); B emu: 3 x 64 x 32 bf16 = 768 x 16 B (3 per thread); B fp32: 512 float4 (2)
    f32x4 ra[4];
    u32x4 rb[3];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, r = e >> 3, c = e & 7;
            ra[i] = *reinterpret_cast<const f32x4*>(Ag + (size_t)r * K + k0 + c * 4);
        }
        if (EMU) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int e = tid + 256 * i, s = e >> 8, r = (e >> 2) & 63, c = e & 3;   // slice s, row r, 8-element chunk c
                rb[i] = *reinterpret_cast<const u32x4*>(Bh + ((size_t)s * N + tile_n * BN + r) * K + k0 + c * 8);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = tid + 256 * i, r = e >> 3, c = e & 7;
                rb[i] = *reinterpret_cast<const u32x4*>(Bf + (size_t)(tile_n * BN + r) * K + k0 + c * 4);
            }
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, r = e >> 3, c = e & 7;
            *reinterpret_cast<f32x4*>(&As[buf][r * AP + c * 4]) = ra[i];
        }
        if (EMU) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int e = tid + 256 * i, s = e >> 8, r = (e >> 2) & 63, c = e & 3;
                *reinterpret_cast<u32x4*>(&Bs[buf][(s * BN + r) * BP + c * 8]) = rb[i];
            }
        } else {
            float* Bsf = reinterpret_cast<float*>(Bs[buf]);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = tid + 256 * i, r = e >> 3, c = e & 7;
                *reinterpret_cast<u32x4*>(&Bsf[r * AP + c * 4]) = rb[i];
            }
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][e] = 0.f;
    gload(0);
    lstore(0);
    __syncthreads();
    const int nk = K / BK;
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nk) gload((ks + 1) * BK);
        if (EMU) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {   // two k = 16 slices of the 32-deep stage; this lane's eight k values: kk * 16 + lh * 8 ..
                u32x4 b0 = *reinterpret_cast<const u32x4*>(&Bs[buf][(0 * BN + wn * 32 + lr) * BP + kk * 16 + lh * 8]);
                u32x4 b1 = *reinterpret_cast<const u32x4*>(&Bs[buf][(1 * BN + wn * 32 + lr) * BP + kk * 16 + lh * 8]);
                u32x4 b2 = *reinterpret_cast<const u32x4*>(&Bs[buf][(2 * BN + wn * 32 + lr) * BP + kk * 16 + lh * 8]);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const float* ar = &As[buf][(wm * 64 + mi * 32 + lr) * AP + kk * 16 + lh * 8];
                    u32x4 a0, a1, a2;
                    split8(*reinterpret_cast<const f32x4*>(ar), *reinterpret_cast<const f32x4*>(ar + 4), a0, a1, a2);
                    // weights as the row operand, pixels as the column operand (a lane ends up with one pixel and runs of four channels),
                    // smallest terms first
                    acc[mi] = mfma_bf16(b2, a0, acc[mi]);
                    acc[mi] = mfma_bf16(b1, a1, acc[mi]);
                    acc[mi] = mfma_bf16(b0, a2, acc[mi]);
                    acc[mi] = mfma_bf16(b1, a0, acc[mi]);
                    acc[mi] = mfma_bf16(b0, a1, acc[mi]);
                    acc[mi] = mfma_bf16(b0, a0, acc[mi]);
                }
            }
        } else {
            const float* Bsf = reinterpret_cast<const float*>(Bs[buf]);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {   // eight k values per lane half and step, as pigemm.hip
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(&Bsf[(wn * 32 + lr) * AP + (kk * 2 + lh) * 4]);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(&As[buf][(wm * 64 + mi * 32 + lr) * AP + (kk * 2 + lh) * 4]);
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.x, a4.x, acc[mi], 0, 0, 0);
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.y, a4.y, acc[mi], 0, 0, 0);
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.z, a4.z, acc[mi], 0, 0, 0);
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4.w, a4.w, acc[mi], 0, 0, 0);
                }
            }
        }
        if (ks + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
    // lane = pixel lr of its 32-row block, channels wn * 32 + 8 g + 4 lh + 0..3
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        float* o = C + (size_t)(tile_m * BM + wm * 64 + mi * 32 + lr) * N + tile_n * BN + wn * 32 + 4 * lh;
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(o + 8 * g) = f32x4{acc[mi][4 * g], acc[mi][4 * g + 1], acc[mi][4 * g + 2], acc[mi][4 * g + 3]};
    }
}

static uint16_t bf16_rne(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int main() {
    static const int shapes[3][3] = {{65536, 256, 1024}, {245760, 128, 128}, {61440, 256, 512}};
    for (const auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        std::vector<float> A((size_t)M * K), B((size_t)N * K);
        srand(1);
        for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
        for (auto& v : B) v = ((float)rand() / RAND_MAX * 2.f - 1.f) / sqrtf((float)K);
        std::vector<uint16_t> B3((size_t)3 * N * K);
        for (size_t i = 0; i < (size_t)N * K; ++i) {
            float r = B[i];
            for (int s = 0; s < 3; ++s) {
                const uint16_t h = bf16_rne(r);
                B3[(size_t)s * N * K + i] = h;
                r -= bf16_f(h);
            }
        }
        float *dA, *dB, *dC;
        uint16_t* dB3;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dB3, B3.size() * 2); hipMalloc(&dC, (size_t)M * N * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB3, B3.data(), B3.size() * 2, hipMemcpyHostToDevice);
        const int grid = (M / BM) * (N / BN);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        printf("M %d N %d K %d (%.2f GFLOP)\n", M, N, K, 2.0 * M * N * K / 1e9);
        for (int emu = 0; emu < 2; ++emu) {
            auto launch = [&]() {
                if (emu) hipLaunchKernelGGL(gemm_kernel<true>, dim3(grid), dim3(256), 0, 0, dA, (const void*)dB3, dC, M, N, K);
                else hipLaunchKernelGGL(gemm_kernel<false>, dim3(grid), dim3(256), 0, 0, dA, (const void*)dB, dC, M, N, K);
            };
            for (int i = 0; i < 3; ++i) launch();
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 10;
            std::vector<float> C((size_t)1024 * N);
            hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);   // first 1024 rows against float64
            double emax = 0, cmax = 0;
            for (int m = 0; m < 1024; m += 7)
                for (int n = 0; n < N; n += 3) {
                    double s = 0;
                    for (int k = 0; k < K; ++k) s += (double)A[(size_t)m * K + k] * (double)B[(size_t)n * K + k];
                    emax = fmax(emax, fabs(s - (double)C[(size_t)m * N + n]));
                    cmax = fmax(cmax, fabs(s));
                }
            printf("  %-46s %8.1f us  %7.1f TFLOP/s (fp32-equivalent)   max |err| against float64 %.3e (largest |c| %.2f)\n",
                   emu ? "six bf16 cross products per fp32 product" : "exact fp32 matrix instruction", ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12, emax, cmax);
        }
        hipFree(dA); hipFree(dB); hipFree(dB3); hipFree(dC);
    }
    return 0;
}
