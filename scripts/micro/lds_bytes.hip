// Does a ds_read_b32 see bytes written by other lanes' ds_write_b8 (after a barrier)?
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
__global__ void k(unsigned* out, int base, int pitch, int n) {
    const int tid = threadIdx.x;
    // phase 1: thread t writes 3 bytes of pixel t into row r (like the H pass)
    for (int r = 0; r < 4; ++r)
        for (int xx = tid; xx < n; xx += 256) {
            const int d = base + r * pitch + xx * 3;
            sm[d + 0] = (unsigned char)(xx * 7 + r);
            sm[d + 1] = (unsigned char)(xx * 13 + r * 3);
            sm[d + 2] = (unsigned char)(xx * 29 + r * 5);
        }
    __syncthreads();
    const unsigned* s32 = reinterpret_cast<const unsigned*>(sm);
    int bad = 0;
    const int row_dw = (n * 3) >> 2;
    for (int i = tid; i < 4 * row_dw; i += 256) {
        const int r = i / row_dw, j = i - r * row_dw;
        const int bo = base + r * pitch + j * 4;
        const unsigned v = s32[bo >> 2];
        const unsigned w = sm[bo] | (sm[bo + 1] << 8) | (sm[bo + 2] << 16) | (sm[bo + 3] << 24);
        if (v != w) { ++bad; if (bad == 1) { out[2 + tid * 2] = v; out[3 + tid * 2] = w; } }
    }
    atomicAdd(&out[0], bad);
}
int main() {
    unsigned* out; hipMalloc(&out, 4096); 
    for (int base : {0, 17216}) for (int pitch : {1732, 1736}) {
        hipMemset(out, 0, 4096);
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 65536, 0, out, base, pitch, 576);
        unsigned h[1024]; hipMemcpy(h, out, 4096, hipMemcpyDeviceToHost);
        printf("base %d pitch %d: mismatching dwords %u  (first: dword %08x bytes %08x)\n", base, pitch, h[0], h[2], h[3]);
    }
    return 0;
}
