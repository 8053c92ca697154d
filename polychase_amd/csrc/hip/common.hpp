// common.hpp -- shared device helpers and plane geometry for the gfx950 kernels.
//
// All float arithmetic that has to match the CPU oracle bit-for-bit is written as separate
// operations and the library is compiled with -ffp-contract=off; hipcc's default correctly rounded
// fp32 divide/sqrt (-fhip-fp32-correctly-rounded-divide-sqrt) is relied upon.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pc {

// Left padding (bytes) of every pyramid plane: >= window, and 16 so interior rows are 16-B aligned.
constexpr int kPadX = 16;

// One pyramid level resident in HBM.  `img` / `der` point at the INTERIOR origin (x=0, y=0); the
// padding (win rows above/below, kPadX bytes left, >= win bytes right) is addressable with negative
// offsets.  `der` is the Scharr plane: one int32 per pixel = (int16 dx) | (int16 dy << 16), zero in
// the padding (OpenCV derivBorder = BORDER_CONSTANT); its pitch in pixels equals `pitch`.
// `img16` is the image once more as uint16 = pixel << 7 (same pitch in pixels, same REFLECT_101 padding, plus
// kImg16SlackRows addressable zero rows above and below): the operand format of the LK inner loop
// (kernels_lk3.hip: a v_dot2_i32_i16 of two such pixels with the 14-bit bilinear weights is 128 x the
// interpolated value, so the rounding shift of CV_DESCALE comes for free as "the high half of the register").
struct Level {
    uint8_t* img;
    int32_t* der;
    uint16_t* img16;
    int w, h;
    int pitch;  // bytes per image row == int32 per derivative row == uint16 per img16 row
};
constexpr int kImg16SlackRows = 2;

// cv::borderInterpolate(p, len, BORDER_REFLECT_101)
__host__ __device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = (p < 0) ? -p : 2 * (len - 1) - p;
    return p;
}

// Order-preserving map float -> uint32 (a < b  <=>  key(a) < key(b)), used for atomicMax and for the
// candidate sort keys (value desc, then linear index desc == gftt.cc:7-12).
__host__ __device__ __forceinline__ uint32_t float_to_ordered(float f) {
    union { float f; uint32_t u; } c;
    c.f = f;
    return (c.u & 0x80000000u) ? ~c.u : (c.u | 0x80000000u);
}
__host__ __device__ __forceinline__ float ordered_to_float(uint32_t k) {
    union { float f; uint32_t u; } c;
    c.u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return c.f;
}

#define PC_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

}  // namespace pc
