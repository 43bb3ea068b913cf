// libjpeg's slow-but-accurate integer DCT pair (jfdctint.c / jidctint.c, JDCT_ISLOW), one dimension of an 8x8 block held in
// registers. Shared by the crops' JPEG round trip (jpeg.hip) and the Motion-JPEG frame decoder (mjpeg.hip); bit-exact
// against oracle/jpeg.py, which is pinned byte for byte against the live libjpeg-turbo behind Pillow.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pa {
namespace dct {

constexpr int CB = 13, P1 = 2;  // CONST_BITS, PASS1_BITS
constexpr int F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270, F_0_899976223 = 7373,
              F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069, F_2_053119869 = 16819,
              F_2_562915447 = 20995, F_3_072711026 = 25172;

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// jfdctint.c, one dimension. FIRST: row pass (outputs scaled up by 2^PASS1_BITS), else column pass.
template <bool FIRST> __device__ __forceinline__ void fdct8(int* d, int stride) {
    const int d0 = d[0], d1 = d[stride], d2 = d[2 * stride], d3 = d[3 * stride], d4 = d[4 * stride], d5 = d[5 * stride],
              d6 = d[6 * stride], d7 = d[7 * stride];
    int tmp0 = d0 + d7, tmp7 = d0 - d7, tmp1 = d1 + d6, tmp6 = d1 - d6, tmp2 = d2 + d5, tmp5 = d2 - d5, tmp3 = d3 + d4, tmp4 = d3 - d4;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    constexpr int n = FIRST ? CB - P1 : CB + P1;
    d[0] = FIRST ? (tmp10 + tmp11) << P1 : descale(tmp10 + tmp11, P1);
    d[4 * stride] = FIRST ? (tmp10 - tmp11) << P1 : descale(tmp10 - tmp11, P1);
    int z1 = (tmp12 + tmp13) * F_0_541196100;
    d[2 * stride] = descale(z1 + tmp13 * F_0_765366865, n);
    d[6 * stride] = descale(z1 + tmp12 * (-F_1_847759065), n);
    z1 = tmp4 + tmp7;
    int z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;
    const int z5 = (z3 + z4) * F_1_175875602;
    tmp4 *= F_0_298631336; tmp5 *= F_2_053119869; tmp6 *= F_3_072711026; tmp7 *= F_1_501321110;
    z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
    z3 += z5; z4 += z5;
    d[7 * stride] = descale(tmp4 + z1 + z3, n);
    d[5 * stride] = descale(tmp5 + z2 + z4, n);
    d[3 * stride] = descale(tmp6 + z2 + z3, n);
    d[stride] = descale(tmp7 + z1 + z4, n);
}

// jidctint.c, one dimension on de-quantised values. FIRST: column pass, else row pass (down to sample scale).
template <bool FIRST> __device__ __forceinline__ void idct8(int* d, int stride) {
    const int i0 = d[0], i1 = d[stride], i2 = d[2 * stride], i3 = d[3 * stride], i4 = d[4 * stride], i5 = d[5 * stride],
              i6 = d[6 * stride], i7 = d[7 * stride];
    int z1 = (i2 + i6) * F_0_541196100;
    int tmp2 = z1 + i6 * (-F_1_847759065), tmp3 = z1 + i2 * F_0_765366865;
    int tmp0 = (i0 + i4) << CB, tmp1 = (i0 - i4) << CB;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = i7; tmp1 = i5; tmp2 = i3; tmp3 = i1;
    z1 = tmp0 + tmp3;
    int z2 = tmp1 + tmp2, z3 = tmp0 + tmp2, z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * F_1_175875602;
    tmp0 *= F_0_298631336; tmp1 *= F_2_053119869; tmp2 *= F_3_072711026; tmp3 *= F_1_501321110;
    z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    constexpr int n = FIRST ? CB - P1 : CB + P1 + 3;
    d[0] = descale(tmp10 + tmp3, n);
    d[7 * stride] = descale(tmp10 - tmp3, n);
    d[stride] = descale(tmp11 + tmp2, n);
    d[6 * stride] = descale(tmp11 - tmp2, n);
    d[2 * stride] = descale(tmp12 + tmp1, n);
    d[5 * stride] = descale(tmp12 - tmp1, n);
    d[3 * stride] = descale(tmp13 + tmp0, n);
    d[4 * stride] = descale(tmp13 - tmp0, n);
}

__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

}  // namespace dct
}  // namespace pa
