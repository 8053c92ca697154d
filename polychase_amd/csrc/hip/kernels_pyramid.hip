// kernels_pyramid.hip -- one fused, LDS-tiled kernel per pyramid level (K1 + K6 + K7 of DESIGN.md section 4).
//
// Replaces, per level l of cv::buildOpticalFlowPyramid (reference cpp/opticalflow.cc:180-187) and the
// cv::cvtColor(COLOR_RGB2GRAY) in front of it (cpp/opticalflow.cc:259,:298):
//     level 0:  RGB / gray / float RGB(A) frame  -> gray tile                      (RGB2Gray, 15-bit coefficients)
//     level l:  parent level's padded plane      -> pyrDown tile (5x5, (sum + 128) >> 8, REFLECT_101)
//   and then, from that tile in LDS (64 x 16 pixels + a 1-px halo), in the same kernel:
//     the u8 image plane, its REFLECT_101 padding of `win` pixels (copyMakeBorder), the uint16 (pixel << 7) plane
//     the LK kernel reads (with the same padding) and the Scharr derivative plane (ScharrDerivInvoker).
// Round 1 ran three kernels per level over global memory (pyrDown, border, Scharr; 18 single-byte loads per lane in
// the last one) plus a gray kernel; here every plane is written once with 4/8/16-byte stores and the level's pixels are
// read from HBM once.  Pure integer arithmetic: bit-exact against oracle/pc_oracle.c and against the unfused
// kernels of kernels_image.hip, which stay as the cross-check (POLYCHASE_PYRAMID_VARIANT=1).
#include "kernels.hpp"

namespace pc {

static thread_local int tl_helper_prio = 0;
int helper_prio_arg() { return tl_helper_prio; }
void set_helper_prio(int hi) { tl_helper_prio = hi ? 1 : 0; }

namespace {

constexpr int TW = 64, TH = 16;                   // output tile
constexpr int CW = TW + 2, CH = TH + 2;           // + 1-px halo for the Scharr stencil
constexpr int C_X0 = 3;                           // LDS column of the halo column x0 - 1 (interior starts 4-byte aligned)
constexpr int C_PITCH = 72;
constexpr int P_H = 2 * TH + 7;                   // parent rows a tile + halo needs (5x5 taps, stride 2)
constexpr int P_PITCH = 144;                      // parent bytes per row: 2 * (TW + 4) + 8, loaded as 36 dwords
constexpr int H_Q = (CW + 3) / 4;                 // horizontal pass: quads of outputs per row (17)
constexpr int H_PITCH = 4 * H_Q;                  // uint16 per row (68)

__device__ __forceinline__ uint32_t gray_of(uint32_t r, uint32_t g, uint32_t b) {
    return (r * 9798u + g * 19235u + b * 3735u + (1u << 14)) >> 15;
}
// numpy `(x * 255).astype(uint8)` of the addon (blender_addon/operators/analysis.py:221-233); see kernels_image.hip
__device__ __forceinline__ uint32_t float_channel_to_u8(float v) {
    const float s = v * 255.0f;
    const int i = (s >= -2147483648.0f && s < 2147483648.0f) ? (int)s : (int)0x80000000;
    return (uint32_t)i & 0xffu;
}

template <int SRC>
__device__ __forceinline__ uint32_t source_gray(const LevelSource& in, int x, int y) {
    const uint8_t* row = in.src + (size_t)y * in.src_pitch;
    if (SRC == SRC_RGB8) {
        const uint8_t* s = row + 3 * (size_t)x;
        return gray_of(s[0], s[1], s[2]);
    } else if (SRC == SRC_GRAY8) {
        return row[x];
    } else {
        const float* s = reinterpret_cast<const float*>(row) + (size_t)x * in.channels;
        return gray_of(float_channel_to_u8(s[0]), float_channel_to_u8(s[1]), float_channel_to_u8(s[2]));
    }
}

// where pixel coordinate v of a `len`-long axis is mirrored to by a REFLECT_101 border of `win` pixels:
// m[0] = itself, m[1] = the pad position below 0 (or INT_MIN), m[2] = the pad position above len - 1 (or INT_MIN)
constexpr int kNone = -(1 << 30);
__device__ __forceinline__ void mirrors(int v, int len, int win, int m[3]) {
    m[0] = v;
    m[1] = (v >= 1 && v <= win) ? -v : kNone;
    m[2] = (v <= len - 2 && v >= len - 1 - win) ? 2 * (len - 1) - v : kNone;
}

}  // namespace

template <int SRC>
__global__ __launch_bounds__(256) void level_kernel(const LevelSource in, const Level out, const int win, int hi_prio) {
    helper_priority(hi_prio);
    __shared__ __attribute__((aligned(16))) uint8_t s_c[CH][C_PITCH];           // the level's tile + halo
    __shared__ __attribute__((aligned(16))) uint8_t s_p[SRC == SRC_PYR ? P_H : 1][P_PITCH];
    __shared__ __attribute__((aligned(16))) uint16_t s_h[SRC == SRC_PYR ? P_H : 1][H_PITCH];

    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int w = out.w, h = out.h;
    if (in.clear) {
        const int stride = (int)(gridDim.x * gridDim.y) * 256;
        for (int i = (int)(blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid; i < in.clear_words; i += stride) in.clear[i] = 0u;
    }

    // ---- phase A: every in-image pixel of [x0 - 1, x0 + TW] x [y0 - 1, y0 + TH] into s_c ----
    if constexpr (SRC == SRC_PYR) {
        const Level& P = in.parent;
        const int px0 = 2 * x0 - 4, py0 = 2 * y0 - 4;     // parent position of s_p[0][0]; px0 is 4-byte aligned
        const int ymin = -win, ymax = P.h + win - 1, xmax = P.pitch - kPadX - 4;
        for (int i = tid; i < P_H * (P_PITCH / 4); i += 256) {
            const int r = i / (P_PITCH / 4), d = i - r * (P_PITCH / 4);
            // clamped positions are never consumed by an in-image output (they lie beyond the parent's 2-px reach)
            const int yy = min(max(py0 + r, ymin), ymax), xx = min(px0 + 4 * d, xmax);
            *reinterpret_cast<uint32_t*>(&s_p[r][4 * d]) = *reinterpret_cast<const uint32_t*>(P.img + (ptrdiff_t)yy * P.pitch + xx);
        }
        __syncthreads();
        // horizontal [1 4 6 4 1] at stride 2: output c (level column x0 - 1 + c) reads parent bytes 2c .. 2c + 4 of the row
        for (int i = tid; i < P_H * H_Q; i += 256) {
            const int r = i / H_Q, q = i - r * H_Q;
            const uint32_t* p = reinterpret_cast<const uint32_t*>(&s_p[r][8 * q]);
            const uint32_t a = p[0], b = p[1], c = p[2];
            int v[11];
            v[0] = a & 0xff; v[1] = (a >> 8) & 0xff; v[2] = (a >> 16) & 0xff; v[3] = a >> 24;
            v[4] = b & 0xff; v[5] = (b >> 8) & 0xff; v[6] = (b >> 16) & 0xff; v[7] = b >> 24;
            v[8] = c & 0xff; v[9] = (c >> 8) & 0xff; v[10] = (c >> 16) & 0xff;
            uint32_t o[4];
#pragma unroll
            for (int k = 0; k < 4; k++) o[k] = (uint32_t)(v[2 * k + 2] * 6 + (v[2 * k + 1] + v[2 * k + 3]) * 4 + v[2 * k] + v[2 * k + 4]);
            *reinterpret_cast<uint2*>(&s_h[r][4 * q]) = make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
        }
        __syncthreads();
        // vertical [1 4 6 4 1] at stride 2, two columns per lane
        for (int i = tid; i < CH * (H_PITCH / 2); i += 256) {
            const int cy = i / (H_PITCH / 2), cp = i - cy * (H_PITCH / 2);
            const int c = 2 * cp;
            if (c >= CW) continue;
            uint32_t t[5];
#pragma unroll
            for (int k = 0; k < 5; k++) t[k] = *reinterpret_cast<const uint32_t*>(&s_h[2 * cy + k][c]);
            const uint32_t lo = ((t[0] & 0xffff) + (t[4] & 0xffff) + ((t[1] & 0xffff) + (t[3] & 0xffff)) * 4 + (t[2] & 0xffff) * 6 + 128) >> 8;
            const uint32_t hi = ((t[0] >> 16) + (t[4] >> 16) + ((t[1] >> 16) + (t[3] >> 16)) * 4 + (t[2] >> 16) * 6 + 128) >> 8;
            s_c[cy][C_X0 + c] = (uint8_t)lo;
            if (c + 1 < CW) s_c[cy][C_X0 + c + 1] = (uint8_t)hi;
        }
    } else {
        // interior of the tile: 4 pixels per lane
        const int r = tid >> 4, g = tid & 15;
        const int x = x0 + 4 * g, y = y0 + r;
        if (y < h && x < w) {
            uint32_t px[4] = {0, 0, 0, 0};
            if (in.aligned && x + 3 < w) {
                const uint8_t* row = in.src + (size_t)y * in.src_pitch;
                if (SRC == SRC_RGB8) {
                    const uint32_t* s = reinterpret_cast<const uint32_t*>(row) + 3 * (x >> 2);
                    const uint32_t d0 = s[0], d1 = s[1], d2 = s[2];
                    px[0] = gray_of(d0 & 0xff, (d0 >> 8) & 0xff, (d0 >> 16) & 0xff);
                    px[1] = gray_of(d0 >> 24, d1 & 0xff, (d1 >> 8) & 0xff);
                    px[2] = gray_of((d1 >> 16) & 0xff, d1 >> 24, d2 & 0xff);
                    px[3] = gray_of((d2 >> 8) & 0xff, (d2 >> 16) & 0xff, d2 >> 24);
                } else if (SRC == SRC_GRAY8) {
                    const uint32_t d = *reinterpret_cast<const uint32_t*>(row + x);
                    px[0] = d & 0xff; px[1] = (d >> 8) & 0xff; px[2] = (d >> 16) & 0xff; px[3] = d >> 24;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) px[k] = source_gray<SRC>(in, x + k, y);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (x + k < w) px[k] = source_gray<SRC>(in, x + k, y);
            }
            *reinterpret_cast<uint32_t*>(&s_c[r + 1][C_X0 + 1 + 4 * g]) = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
        }
        // halo ring: neighbours that belong to the adjacent tiles (in-image positions only)
        if (tid < 2 * CW + 2 * TH) {
            int cx, cy;
            if (tid < 2 * CW) {
                cy = tid < CW ? 0 : CH - 1;
                cx = tid < CW ? tid : tid - CW;
            } else {
                const int k = tid - 2 * CW;
                cy = 1 + (k >> 1);
                cx = (k & 1) ? CW - 1 : 0;
            }
            const int ax = x0 - 1 + cx, ay = y0 - 1 + cy;
            if (ax >= 0 && ax < w && ay >= 0 && ay < h) s_c[cy][C_X0 + cx] = (uint8_t)source_gray<SRC>(in, ax, ay);
        }
    }
    __syncthreads();
    // positions one pixel outside the image take the REFLECT_101 value (the Scharr stencil's border, and the first
    // ring of the padding); only tiles that touch the image edge have any
    const bool edge_tile = x0 == 0 || y0 == 0 || x0 + TW >= w || y0 + TH >= h;
    if (edge_tile) {
        for (int i = tid; i < CW * CH; i += 256) {
            const int cy = i / CW, cx = i - cy * CW;
            const int ax = x0 - 1 + cx, ay = y0 - 1 + cy;
            if ((ax < 0 || ax >= w || ay < 0 || ay >= h) && ax >= -1 && ax <= w && ay >= -1 && ay <= h) {
                // the source is an in-image position of this tile's block: written in phase A, not by this pass
                const int sx = reflect101(ax, w) - (x0 - 1), sy = reflect101(ay, h) - (y0 - 1);
                if (sx >= 0 && sx < CW && sy >= 0 && sy < CH) s_c[cy][C_X0 + cx] = s_c[sy][C_X0 + sx];
            }
        }
        __syncthreads();
    }

    // ---- phase B: the level's planes, 4 pixels per lane ----
    const int r = tid >> 4, g = tid & 15;
    const int x = x0 + 4 * g, y = y0 + r;
    if (y >= h || x >= w) return;
    // bytes x - 1 .. x + 4 of the three rows: the dwords at columns 4g, 4g + 4, 4g + 8 (aligned)
    uint32_t rows[3][3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(&s_c[r + k][4 * g]);
        rows[k][0] = p[0];
        rows[k][1] = p[1];
        rows[k][2] = p[2];
    }
    const uint32_t centre = rows[1][1];                  // pixels x .. x + 3
    int t0[6], t1[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        // byte (3 + i) of the 12-byte row
        const int sh = 8 * ((3 + i) & 3), wd = (3 + i) >> 2;
        const int a = (rows[0][wd] >> sh) & 0xff, c = (rows[1][wd] >> sh) & 0xff, b = (rows[2][wd] >> sh) & 0xff;
        t0[i] = (a + b) * 3 + c * 10;
        t1[i] = b - a;
    }
    int32_t d[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int dx = t0[i + 2] - t0[i];
        const int dy = (t1[i + 2] + t1[i]) * 3 + t1[i + 1] * 10;
        d[i] = (int32_t)((uint32_t)(dx & 0xffff) | ((uint32_t)(dy & 0xffff) << 16));
    }
    const uint32_t w16lo = __builtin_amdgcn_perm(0u, centre, 0x0c010c00u) << 7;   // (p0, p1) << 7 as 16-bit lanes
    const uint32_t w16hi = __builtin_amdgcn_perm(0u, centre, 0x0c030c02u) << 7;
    uint8_t* img = out.img + (ptrdiff_t)y * out.pitch + x;
    uint16_t* img16 = out.img16 + (ptrdiff_t)y * out.pitch + x;
    int32_t* der = out.der + (ptrdiff_t)y * out.pitch + x;
    if (x + 3 < w) {
        *reinterpret_cast<uint32_t*>(img) = centre;
        *reinterpret_cast<uint2*>(img16) = make_uint2(w16lo, w16hi);
        *reinterpret_cast<int4*>(der) = make_int4(d[0], d[1], d[2], d[3]);
    } else {
        for (int i = 0; i < 4 && x + i < w; i++) {
            const uint32_t v = (centre >> (8 * i)) & 0xff;
            img[i] = (uint8_t)v;
            img16[i] = (uint16_t)(v << 7);
            der[i] = d[i];
        }
    }
    // REFLECT_101 padding (copyMakeBorder in buildOpticalFlowPyramid): every pixel writes its own mirror images, so
    // no tile depends on a neighbour's data however the tiles fall on the image edge
    if (x <= win || x + 3 >= w - 1 - win || y <= win || y >= h - 1 - win) {
        int my[3];
        mirrors(y, h, win, my);
        for (int i = 0; i < 4 && x + i < w; i++) {
            const uint32_t v = (centre >> (8 * i)) & 0xff;
            int mx[3];
            mirrors(x + i, w, win, mx);
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b < 3; b++) {
                    if ((a | b) == 0 || my[a] == kNone || mx[b] == kNone) continue;
                    const ptrdiff_t o = (ptrdiff_t)my[a] * out.pitch + mx[b];
                    out.img[o] = (uint8_t)v;
                    out.img16[o] = (uint16_t)(v << 7);
                }
        }
    }
}

void launch_level(const LevelSource& in, const Level& out, int win, hipStream_t s) {
    dim3 grid((out.w + TW - 1) / TW, (out.h + TH - 1) / TH);
    switch (in.kind) {
        case SRC_PYR: hipLaunchKernelGGL(level_kernel<SRC_PYR>, grid, dim3(256), 0, s, in, out, win, helper_prio_arg()); break;
        case SRC_RGB8: hipLaunchKernelGGL(level_kernel<SRC_RGB8>, grid, dim3(256), 0, s, in, out, win, helper_prio_arg()); break;
        case SRC_GRAY8: hipLaunchKernelGGL(level_kernel<SRC_GRAY8>, grid, dim3(256), 0, s, in, out, win, helper_prio_arg()); break;
        default: hipLaunchKernelGGL(level_kernel<SRC_RGBF32>, grid, dim3(256), 0, s, in, out, win, helper_prio_arg()); break;
    }
}

}  // namespace pc
