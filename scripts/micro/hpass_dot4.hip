// Micro-experiment: Pillow's horizontal 8bpc pass (7 taps of 22-bit fixed-point coefficients on
// interleaved BGR bytes in LDS) in three exact forms, timed in isolation on the shape the crop stage runs it on:
//   A  crop_fused_kernel's: six aligned dwords per window, v_alignbyte, then per tap and channel one byte extract (v_bfe_u32)
//      + one 24-bit multiply-add                                                                      (21 x 2 per output pixel)
//   B  the same interleaved window, tap bytes of a channel gathered with v_perm_b32, coefficients as base-256 digits of
//      (k + 2^22): three v_dot4_u32_u8 + one sum-of-pixels dot per four taps; D0 + (D1 << 8) + (D2 << 16) - (S << 22) wraps
//      around in 32 bits to the exact sum
//   C  B on CHANNEL-PLANAR rows (what a re-laid-out stage 0 would give): no gathering, two aligned dwords per channel
// Every form must produce the same bytes. Build: hipcc --offload-arch=gfx950 -O3 -o build/hpass_dot4 scripts/micro/hpass_dot4.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int PREC = 22;
constexpr int ROWS = 64, IN_W = 384, OUT_W = 320;   // one sub-band: 64 rows of a (d + 60) -> d pass
constexpr int PITCH = IN_W * 3 + 16;                // interleaved row pitch (multiple of 4)
constexpr int PPITCH = IN_W + 4;                    // planar row pitch per channel

__device__ __forceinline__ int clip8(int v) { v >>= PREC; return v < 0 ? 0 : (v > 255 ? 255 : v); }

template <int FORM>
__global__ __launch_bounds__(512) void hpass_kernel(const uint8_t* __restrict__ in, const uint8_t* __restrict__ planar, const int* __restrict__ coef,
                                                    uint8_t* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t sm[];
    uint8_t* src = sm;                       // FORM C: three planes of ROWS x PPITCH; else ROWS x PITCH
    uint8_t* dst = sm + ROWS * PITCH;
    const int tid = threadIdx.x;
    const uint8_t* g = (FORM == 2 ? planar : in) + (size_t)blockIdx.x * ROWS * PITCH;
    for (int i = tid; i < ROWS * PITCH / 16; i += 512) reinterpret_cast<uint4*>(src)[i] = reinterpret_cast<const uint4*>(g)[i];
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        for (int item = tid; item < OUT_W * (ROWS / 4); item += 512) {
            const int xx = item % OUT_W, ck = item / OUT_W;
            const int* row = coef + xx * 9;
            const int xmin = row[0];
            int k[7];
#pragma unroll
            for (int t = 0; t < 7; ++t) k[t] = row[2 + t];
            if (FORM == 0) {
                const int sb = xmin * 3;
                const uint32_t shb = (uint32_t)(sb & 3);
                const int a_dw = sb >> 2;
                uint32_t w[4][6];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t* s = reinterpret_cast<const uint32_t*>(src + (ck * 4 + u) * PITCH) + a_dw;
#pragma unroll
                    for (int q = 0; q < 6; ++q) w[u][q] = s[q];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    uint32_t r[6];
#pragma unroll
                    for (int q = 0; q < 5; ++q) r[q] = __builtin_amdgcn_alignbyte(w[u][q + 1], w[u][q], shb);
                    r[5] = __builtin_amdgcn_alignbyte(0u, w[u][5], shb);
                    int a0 = 1 << (PREC - 1), a1 = a0, a2 = a0;
#pragma unroll
                    for (int t = 0; t < 7; ++t) {
                        const int j0 = 3 * t, j1 = 3 * t + 1, j2 = 3 * t + 2;
                        a0 += __mul24((int)((r[j0 >> 2] >> (8 * (j0 & 3))) & 0xff), k[t]);
                        a1 += __mul24((int)((r[j1 >> 2] >> (8 * (j1 & 3))) & 0xff), k[t]);
                        a2 += __mul24((int)((r[j2 >> 2] >> (8 * (j2 & 3))) & 0xff), k[t]);
                    }
                    uint8_t* d = dst + (ck * 4 + u) * (OUT_W * 3) + xx * 3;
                    d[0] = (uint8_t)clip8(a0); d[1] = (uint8_t)clip8(a1); d[2] = (uint8_t)clip8(a2);
                }
            } else {
                // digits of k + 2^22, four taps per dword (taps 0-3, taps 4-6 + a zero)
                uint32_t dg[2][3];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int dd = 0; dd < 3; ++dd) {
                        uint32_t v = 0;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int tt = 4 * h + t;
                            const uint32_t kb = tt < 7 ? (uint32_t)(k[tt] + (1 << PREC)) : 0u;
                            v |= ((kb >> (8 * dd)) & 0xffu) << (8 * t);
                        }
                        dg[h][dd] = v;
                    }
                const uint32_t ones0 = 0x01010101u, ones1 = 0x00010101u;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    uint32_t px[3][2];   // [channel][tap group]: four tap bytes of one channel
                    if (FORM == 1) {
                        const int sb = xmin * 3;
                        const uint32_t shb = (uint32_t)(sb & 3);
                        const uint32_t* s = reinterpret_cast<const uint32_t*>(src + (ck * 4 + u) * PITCH) + (sb >> 2);
                        uint32_t w[6], r[6];
#pragma unroll
                        for (int q = 0; q < 6; ++q) w[q] = s[q];
#pragma unroll
                        for (int q = 0; q < 5; ++q) r[q] = __builtin_amdgcn_alignbyte(w[q + 1], w[q], shb);
                        r[5] = __builtin_amdgcn_alignbyte(0u, w[5], shb);
                        // byte j of the 24-byte window = byte (j & 3) of r[j >> 2]; v_perm_b32(hi, lo, sel): sel 0-3 -> lo, 4-7 -> hi
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            // taps 0-3: bytes c, 3 + c, 6 + c, 9 + c (dwords 0, 0 | 1, 1 | 2, 2): two perms
                            const int j0 = c, j1 = 3 + c, j2 = 6 + c, j3 = 9 + c;
                            const uint32_t lo = __builtin_amdgcn_perm(r[j1 >> 2], r[j0 >> 2], (uint32_t)((j0 & 3) | ((((j1 >> 2) != (j0 >> 2) ? 4 : 0) + (j1 & 3)) << 8)));
                            const uint32_t hi = __builtin_amdgcn_perm(r[j3 >> 2], r[j2 >> 2], (uint32_t)((j2 & 3) | ((((j3 >> 2) != (j2 >> 2) ? 4 : 0) + (j3 & 3)) << 8)));
                            px[c][0] = (lo & 0xffffu) | (hi << 16);
                            const int j4 = 12 + c, j5 = 15 + c, j6 = 18 + c;
                            const uint32_t lo2 = __builtin_amdgcn_perm(r[j5 >> 2], r[j4 >> 2], (uint32_t)((j4 & 3) | ((((j5 >> 2) != (j4 >> 2) ? 4 : 0) + (j5 & 3)) << 8)));
                            px[c][1] = (lo2 & 0xffffu) | (((r[j6 >> 2] >> (8 * (j6 & 3))) & 0xffu) << 16);
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const uint8_t* pr = src + c * ROWS * PPITCH + (ck * 4 + u) * PPITCH;
                            const uint32_t shb = (uint32_t)(xmin & 3);
                            const uint32_t* s = reinterpret_cast<const uint32_t*>(pr) + (xmin >> 2);
                            const uint32_t w0 = s[0], w1 = s[1], w2 = s[2];
                            px[c][0] = __builtin_amdgcn_alignbyte(w1, w0, shb);
                            px[c][1] = __builtin_amdgcn_alignbyte(w2, w1, shb) & 0x00ffffffu;
                        }
                    }
                    int a[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        uint32_t d0 = __builtin_amdgcn_udot4(px[c][0], dg[0][0], 0u, false);
                        d0 = __builtin_amdgcn_udot4(px[c][1], dg[1][0], d0, false);
                        uint32_t d1 = __builtin_amdgcn_udot4(px[c][0], dg[0][1], 0u, false);
                        d1 = __builtin_amdgcn_udot4(px[c][1], dg[1][1], d1, false);
                        uint32_t d2 = __builtin_amdgcn_udot4(px[c][0], dg[0][2], 0u, false);
                        d2 = __builtin_amdgcn_udot4(px[c][1], dg[1][2], d2, false);
                        uint32_t sp = __builtin_amdgcn_udot4(px[c][0], ones0, 0u, false);
                        sp = __builtin_amdgcn_udot4(px[c][1], ones1, sp, false);
                        a[c] = (int)(d0 + (d1 << 8) + (d2 << 16) - (sp << PREC) + (1u << (PREC - 1)));
                    }
                    uint8_t* d = dst + (ck * 4 + u) * (OUT_W * 3) + xx * 3;
                    d[0] = (uint8_t)clip8(a[0]); d[1] = (uint8_t)clip8(a[1]); d[2] = (uint8_t)clip8(a[2]);
                }
            }
        }
        __syncthreads();
    }
    uint8_t* o = out + (size_t)blockIdx.x * ROWS * OUT_W * 3;
    for (int i = tid; i < ROWS * OUT_W * 3 / 4; i += 512) reinterpret_cast<uint32_t*>(o)[i] = reinterpret_cast<const uint32_t*>(dst)[i];
}

int main() {
    const int blocks = 1024, iters = 20;
    std::vector<uint8_t> in((size_t)blocks * ROWS * PITCH), planar(in.size(), 0);
    uint32_t s = 12345;
    for (auto& b : in) { s = s * 1664525u + 1013904223u; b = (uint8_t)(s >> 24); }
    for (int b = 0; b < blocks; ++b)
        for (int y = 0; y < ROWS; ++y)
            for (int x = 0; x < IN_W; ++x)
                for (int c = 0; c < 3; ++c)
                    planar[(size_t)b * ROWS * PITCH + c * ROWS * PPITCH + y * PPITCH + x] = in[(size_t)b * ROWS * PITCH + y * PITCH + x * 3 + c];
    static_assert(3 * ROWS * PPITCH <= ROWS * PITCH, "planes fit the interleaved band");
    // Pillow's precompute_coeffs for bicubic (a = -0.5), IN_W -> OUT_W, 22-bit rounding
    std::vector<int> coef(OUT_W * 9, 0);
    const double scale = (double)IN_W / OUT_W, support = 2.0 * scale;
    auto bic = [](double x) { const double a = -0.5; x = x < 0 ? -x : x; return x < 1 ? ((a + 2) * x - (a + 3)) * x * x + 1 : (x < 2 ? (((x - 5) * x + 8) * x - 4) * a : 0.0); };
    for (int xx = 0; xx < OUT_W; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5), xmax = (int)(center + support + 0.5);
        xmin = xmin < 0 ? 0 : xmin; xmax = xmax > IN_W ? IN_W : xmax;
        int n = xmax - xmin; if (n > 7) n = 7;
        double w[7], ww = 0;
        for (int t = 0; t < n; ++t) { w[t] = bic((t + xmin - center + 0.5) / scale); ww += w[t]; }
        coef[xx * 9] = xmin; coef[xx * 9 + 1] = n;
        for (int t = 0; t < n; ++t) { const double v = w[t] / ww * (1 << PREC); coef[xx * 9 + 2 + t] = (int)(v < 0 ? v - 0.5 : v + 0.5); }
    }
    uint8_t *d_in, *d_pl, *d_out[3]; int* d_coef;
    hipMalloc(&d_in, in.size()); hipMalloc(&d_pl, in.size()); hipMalloc(&d_coef, coef.size() * 4);
    hipMemcpy(d_in, in.data(), in.size(), hipMemcpyHostToDevice); hipMemcpy(d_pl, planar.data(), in.size(), hipMemcpyHostToDevice);
    hipMemcpy(d_coef, coef.data(), coef.size() * 4, hipMemcpyHostToDevice);
    const size_t out_bytes = (size_t)blocks * ROWS * OUT_W * 3, lds = ROWS * PITCH + ROWS * OUT_W * 3;
    for (auto& o : d_out) hipMalloc(&o, out_bytes);
    hipFuncSetAttribute((const void*)hpass_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)hpass_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)hpass_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const char* names[3] = {"A  bfe + mad24 (crop_fused_kernel's)", "B  v_perm gather + dot4 digits, interleaved rows", "C  dot4 digits, channel-planar rows"};
    std::vector<std::vector<uint8_t>> host(3, std::vector<uint8_t>(out_bytes));
    for (int f = 0; f < 3; ++f) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (f == 0) hipLaunchKernelGGL(hpass_kernel<0>, dim3(blocks), dim3(512), lds, 0, d_in, d_pl, d_coef, d_out[0], iters);
            if (f == 1) hipLaunchKernelGGL(hpass_kernel<1>, dim3(blocks), dim3(512), lds, 0, d_in, d_pl, d_coef, d_out[1], iters);
            if (f == 2) hipLaunchKernelGGL(hpass_kernel<2>, dim3(blocks), dim3(512), lds, 0, d_in, d_pl, d_coef, d_out[2], iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(host[f].data(), d_out[f], out_bytes, hipMemcpyDeviceToHost);
        const double px = (double)blocks * ROWS * OUT_W * iters;
        printf("%-52s %8.3f ms  %6.1f G output pixels/s  %s\n", names[f], ms, px / ms * 1e-6, f == 0 ? "" : (host[f] == host[0] ? "== A (bit-exact)" : "DIFFERS from A"));
    }
    return 0;
}
