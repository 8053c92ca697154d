// kernels_image.hip -- gray conversion and LK pyramid construction on gfx950.
//
// Replaces cv::cvtColor(COLOR_RGB2GRAY) (reference cpp/opticalflow.cc:259,:298) and
// cv::buildOpticalFlowPyramid (cpp/opticalflow.cc:180-187): pyrDown 5x5, REFLECT_101 padding,
// Scharr derivative planes.  Pure integer arithmetic => bit-exact against the oracle.
// All kernels are HBM-streaming stencils: 4 pixels per lane, dword loads/stores on 16-B aligned
// interior rows.
#include "kernels.hpp"

namespace pc {

// ------------------------------------------------------------------------------------------------
// K1  rgb2gray: Y = (9798 R + 19235 G + 3735 B + 2^14) >> 15   (OpenCV 4.x RGB2Gray<uchar>)
// One lane = 4 pixels = 3 dwords in, 1 dword out.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t gray_of(uint32_t r, uint32_t g, uint32_t b) {
    return (r * 9798u + g * 19235u + b * 3735u + (1u << 14)) >> 15;
}

__global__ __launch_bounds__(256) void rgb2gray_x4_kernel(const uint8_t* __restrict__ rgb,
                                                          size_t rgb_pitch, uint8_t* __restrict__ dst,
                                                          int dst_pitch, int w4, int h) {
    const int gx = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (gx >= w4) return;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(rgb + (size_t)y * rgb_pitch) + 3 * gx;
    const uint32_t d0 = src[0], d1 = src[1], d2 = src[2];
    const uint32_t g0 = gray_of(d0 & 0xff, (d0 >> 8) & 0xff, (d0 >> 16) & 0xff);
    const uint32_t g1 = gray_of(d0 >> 24, d1 & 0xff, (d1 >> 8) & 0xff);
    const uint32_t g2 = gray_of((d1 >> 16) & 0xff, d1 >> 24, d2 & 0xff);
    const uint32_t g3 = gray_of((d2 >> 8) & 0xff, (d2 >> 16) & 0xff, d2 >> 24);
    reinterpret_cast<uint32_t*>(dst + (size_t)y * dst_pitch)[gx] = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
}

// generic (unaligned / width not a multiple of 4) fallback: one pixel per lane
__global__ __launch_bounds__(256) void rgb2gray_x1_kernel(const uint8_t* __restrict__ rgb,
                                                          size_t rgb_pitch, uint8_t* __restrict__ dst,
                                                          int dst_pitch, int w, int h) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w) return;
    const uint8_t* s = rgb + (size_t)y * rgb_pitch + 3 * (size_t)x;
    dst[(size_t)y * dst_pitch + x] = (uint8_t)gray_of(s[0], s[1], s[2]);
}

void launch_rgb2gray(const uint8_t* rgb, size_t rgb_pitch, const Level& l0, hipStream_t s) {
    const bool aligned = ((reinterpret_cast<uintptr_t>(rgb) & 3) == 0) && ((rgb_pitch & 3) == 0) &&
                         ((l0.w & 3) == 0);
    if (aligned) {
        const int w4 = l0.w / 4;
        dim3 grid((w4 + 255) / 256, l0.h);
        hipLaunchKernelGGL(rgb2gray_x4_kernel, grid, dim3(256), 0, s, rgb, rgb_pitch, l0.img, l0.pitch, w4, l0.h);
    } else {
        dim3 grid((l0.w + 255) / 256, l0.h);
        hipLaunchKernelGGL(rgb2gray_x1_kernel, grid, dim3(256), 0, s, rgb, rgb_pitch, l0.img, l0.pitch, l0.w, l0.h);
    }
}

// Float frames as Blender hands them out (H x W x C float32, C = 3 or 4, values nominally in [0,1]):
// the addon converts them with numpy `(image * 255).astype(np.uint8)` before provide_frame
// (blender_addon/operators/analysis.py:221-233), i.e. an fp32 multiply and a truncating cast whose
// out-of-range behaviour is "low 8 bits of the int32".  Same arithmetic here, fused with RGB2GRAY.
__device__ __forceinline__ uint32_t float_channel_to_u8(float v) {
    const float s = v * 255.0f;
    // C-style float -> int32 (truncate; NaN / out of int32 range -> INT_MIN like cvttss2si), then the low byte
    const int i = (s >= -2147483648.0f && s < 2147483648.0f) ? (int)s : (int)0x80000000;
    return (uint32_t)i & 0xffu;
}

__global__ __launch_bounds__(256) void rgbf32_to_gray_kernel(const float* __restrict__ src, size_t src_pitch, int channels,
                                                             uint8_t* __restrict__ dst, int dst_pitch, int w, int h) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w) return;
    const float* p = reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(src) + (size_t)y * src_pitch) +
                     (size_t)x * channels;
    dst[(size_t)y * dst_pitch + x] =
        (uint8_t)gray_of(float_channel_to_u8(p[0]), float_channel_to_u8(p[1]), float_channel_to_u8(p[2]));
}

void launch_rgbf32_to_gray(const float* rgb, size_t rgb_pitch, int channels, const Level& l0, hipStream_t s) {
    dim3 grid((l0.w + 255) / 256, l0.h);
    hipLaunchKernelGGL(rgbf32_to_gray_kernel, grid, dim3(256), 0, s, rgb, rgb_pitch, channels, l0.img, l0.pitch, l0.w, l0.h);
}

__global__ __launch_bounds__(256) void copy_gray_kernel(const uint8_t* __restrict__ src, size_t src_pitch,
                                                        uint8_t* __restrict__ dst, int dst_pitch, int w, int h) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w) return;
    dst[(size_t)y * dst_pitch + x] = src[(size_t)y * src_pitch + x];
}

void launch_copy_gray(const uint8_t* gray, size_t gray_pitch, const Level& l0, hipStream_t s) {
    dim3 grid((l0.w + 255) / 256, l0.h);
    hipLaunchKernelGGL(copy_gray_kernel, grid, dim3(256), 0, s, gray, gray_pitch, l0.img, l0.pitch, l0.w, l0.h);
}

// ------------------------------------------------------------------------------------------------
// K7  pyrDown: separable [1 4 6 4 1], integer, (sum + 128) >> 8, REFLECT_101 on the un-padded
// source, dst = ((sw+1)/2, (sh+1)/2)   (OpenCV PyrDownInvoker<FixPtCast<uchar,8>>).
// One lane = 4 horizontally adjacent outputs: needs source columns 2x-2 .. 2x+8 (11 bytes) of 5 rows.
// Interior lanes read them as 4 aligned dwords per row (x is a multiple of 4 => 2x-2 = 8k-2);
// lanes touching the image border take the per-byte reflect path.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pyr_row_fast(const uint8_t* __restrict__ row, int sx0, int out[4]) {
    // bytes sx0-2 .. sx0+8 ; sx0 % 8 == 0 and the row base is 16-B aligned => (sx0-4) is 4-aligned
    const uint32_t* p = reinterpret_cast<const uint32_t*>(row + sx0 - 4);
    const uint32_t a = p[0], b = p[1], c = p[2], d = p[3];
    int v[11];
    v[0] = (a >> 16) & 0xff; v[1] = a >> 24;
    v[2] = b & 0xff; v[3] = (b >> 8) & 0xff; v[4] = (b >> 16) & 0xff; v[5] = b >> 24;
    v[6] = c & 0xff; v[7] = (c >> 8) & 0xff; v[8] = (c >> 16) & 0xff; v[9] = c >> 24;
    v[10] = d & 0xff;
#pragma unroll
    for (int i = 0; i < 4; i++)
        out[i] = v[2 * i + 2] * 6 + (v[2 * i + 1] + v[2 * i + 3]) * 4 + v[2 * i] + v[2 * i + 4];
}

__device__ __forceinline__ void pyr_row_slow(const uint8_t* __restrict__ row, int sx0, int sw, int out[4]) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int c = sx0 + 2 * i;
        out[i] = row[reflect101(c, sw)] * 6 + (row[reflect101(c - 1, sw)] + row[reflect101(c + 1, sw)]) * 4 +
                 row[reflect101(c - 2, sw)] + row[reflect101(c + 2, sw)];
    }
}

__global__ __launch_bounds__(256) void pyrdown_kernel(const uint8_t* __restrict__ src, int spitch, int sw,
                                                      int sh, uint8_t* __restrict__ dst, int dpitch, int dw,
                                                      int dh) {
    const int gx = blockIdx.x * blockDim.x + threadIdx.x;  // group of 4 outputs
    const int y = blockIdx.y;
    const int x0 = gx * 4;
    if (x0 >= dw) return;
    const int sx0 = 2 * x0;
    const bool fast = (sx0 >= 4) && (sx0 + 8 < sw);
    int acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int sy = reflect101(2 * y - 2 + k, sh);
        const uint8_t* row = src + (size_t)sy * spitch;
        int r[4];
        if (fast) pyr_row_fast(row, sx0, r);
        else pyr_row_slow(row, sx0, sw, r);
        const int wgt = (k == 2) ? 6 : ((k == 1 || k == 3) ? 4 : 1);
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] += r[i] * wgt;
    }
    uint8_t* d = dst + (size_t)y * dpitch + x0;
    if (x0 + 3 < dw) {
        const uint32_t o = (uint32_t)((acc[0] + 128) >> 8) | ((uint32_t)((acc[1] + 128) >> 8) << 8) |
                           ((uint32_t)((acc[2] + 128) >> 8) << 16) | ((uint32_t)((acc[3] + 128) >> 8) << 24);
        *reinterpret_cast<uint32_t*>(d) = o;
    } else {
        for (int i = 0; i < 4 && x0 + i < dw; i++) d[i] = (uint8_t)((acc[i] + 128) >> 8);
    }
}

void launch_pyrdown(const Level& src, const Level& dst, hipStream_t s) {
    const int groups = (dst.w + 3) / 4;
    dim3 grid((groups + 255) / 256, dst.h);
    hipLaunchKernelGGL(pyrdown_kernel, grid, dim3(256), 0, s, src.img, src.pitch, src.w, src.h, dst.img,
                       dst.pitch, dst.w, dst.h);
}

// ------------------------------------------------------------------------------------------------
// REFLECT_101 padding of `win` pixels on every side (copyMakeBorder in buildOpticalFlowPyramid).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void border_kernel(uint8_t* __restrict__ img, int pitch, int w, int h, int win) {
    const int pw = w + 2 * win;
    const int n_tb = pw * 2 * win;       // top + bottom bands, full padded width
    const int n_lr = h * 2 * win;        // left + right bands of the interior rows
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tb + n_lr) return;
    int x, y;
    if (i < n_tb) {
        const int r = i / pw;
        x = i - r * pw - win;
        y = (r < win) ? r - win : h + (r - win);
    } else {
        const int j = i - n_tb;
        y = j / (2 * win);
        const int c = j - y * 2 * win;
        x = (c < win) ? c - win : w + (c - win);
    }
    img[(ptrdiff_t)y * pitch + x] = img[(ptrdiff_t)reflect101(y, h) * pitch + reflect101(x, w)];
}

void launch_border(const Level& l, int win, hipStream_t s) {
    const int total = (l.w + 2 * win) * 2 * win + l.h * 2 * win;
    hipLaunchKernelGGL(border_kernel, dim3((total + 255) / 256), dim3(256), 0, s, l.img, l.pitch, l.w, l.h, win);
}

// ------------------------------------------------------------------------------------------------
// uint16 copy of a padded plane, value = pixel << 7 (Level::img16): every row of the padded plane, the whole
// pitch (4 pixels per lane; the bytes outside [-win, w + win) are never consumed, they only have to be addressable).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void widen_kernel(const uint8_t* __restrict__ img, uint16_t* __restrict__ img16, int pitch,
                                                    int y0) {
    const int gx = blockIdx.x * blockDim.x + threadIdx.x;   // dword of the row, counted from the plane's left edge
    if (gx * 4 >= pitch) return;
    const ptrdiff_t o = (ptrdiff_t)(y0 + (int)blockIdx.y) * pitch - kPadX + 4 * gx;
    const uint32_t d = *reinterpret_cast<const uint32_t*>(img + o);
    const uint32_t lo = __builtin_amdgcn_perm(0u, d, 0x0c010c00u) << 7;   // (p0, p1) as 16-bit lanes
    const uint32_t hi = __builtin_amdgcn_perm(0u, d, 0x0c030c02u) << 7;   // (p2, p3)
    *reinterpret_cast<uint2*>(img16 + o) = make_uint2(lo, hi);
}

void launch_widen(const Level& l, int win, hipStream_t s) {
    dim3 grid((l.pitch / 4 + 255) / 256, l.h + 2 * win);
    hipLaunchKernelGGL(widen_kernel, grid, dim3(256), 0, s, l.img, l.img16, l.pitch, -win);
}

// ------------------------------------------------------------------------------------------------
// K6  Scharr derivative plane (OpenCV ScharrDerivInvoker): t0 = 3*(above+below) + 10*cur,
// t1 = below - above; dx = t0[x+1] - t0[x-1]; dy = 3*(t1[x+1] + t1[x-1]) + 10*t1[x].
// REFLECT_101 inside the level == the already filled 1-px image border.
// One lane = 4 pixels: 3 rows x 6 bytes in (two aligned dwords + neighbours), 4 dwords (dx|dy) out.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scharr_kernel(const uint8_t* __restrict__ img, int pitch, int w, int h,
                                                     int32_t* __restrict__ der) {
    const int gx = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int x0 = gx * 4;
    if (x0 >= w) return;
    int t0[6], t1[6];
    const uint8_t* r0 = img + (ptrdiff_t)(y - 1) * pitch + x0 - 1;
    const uint8_t* r1 = r0 + pitch;
    const uint8_t* r2 = r1 + pitch;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const int a = r0[i], c = r1[i], b = r2[i];
        t0[i] = (a + b) * 3 + c * 10;
        t1[i] = b - a;
    }
    int32_t out[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int dx = t0[i + 2] - t0[i];
        const int dy = (t1[i + 2] + t1[i]) * 3 + t1[i + 1] * 10;
        out[i] = (int32_t)((uint32_t)(dx & 0xffff) | ((uint32_t)(dy & 0xffff) << 16));
    }
    int32_t* d = der + (size_t)y * pitch + x0;
    if (x0 + 3 < w) {
        *reinterpret_cast<int4*>(d) = make_int4(out[0], out[1], out[2], out[3]);
    } else {
        for (int i = 0; i < 4 && x0 + i < w; i++) d[i] = out[i];
    }
}

void launch_scharr(const Level& l, hipStream_t s) {
    const int groups = (l.w + 3) / 4;
    dim3 grid((groups + 255) / 256, l.h);
    hipLaunchKernelGGL(scharr_kernel, grid, dim3(256), 0, s, l.img, l.pitch, l.w, l.h, l.der);
}

}  // namespace pc
