// kernels_gftt.hip -- Shi-Tomasi corner detection (GoodFeaturesToTrack, reference cpp/feature_detection/gftt.cc:14-192)
// entirely on gfx950:
//   K2  cv::cornerMinEigenVal(block 3, Sobel 3) + per-grid-cell cv::minMaxLoc             gftt.cc:35, :61-63
//   K3  per-cell cv::threshold(THRESH_TOZERO) + cv::dilate 3x3 + strict-interior maxima   gftt.cc:64-86
//   K4  std::sort(greaterThanPtr) of the candidates                                       gftt.cc:7-12, :98
//   K5  greedy min-distance suppression in that order                                     gftt.cc:100-164
//   ordered compaction of the accepted corners -> keypoints in acceptance order            gftt.cc:157-162
// Float order follows oracle/pc_oracle.c exactly (no FMA contraction; the fp64 box sums are exact, so their
// summation order is free).
//
// Dense maps kept per frame: the min-eig map (float, written by K2, read by K3 and K5) and ONE byte per pixel of
// candidate state (0 no candidate, 1 candidate, 2 accepted, 3 rejected; written by K3, updated by K5).  Round 1 had
// two more dword maps (priority + decision, 8 bytes per pixel written per frame); a neighbour's priority is now
// read from the min-eig map itself.
#include <algorithm>
#include <cstring>
#include <type_traits>
#include <rocprim/device/device_radix_sort.hpp>

#include <cmath>

#include "kernels.hpp"

namespace pc {

// ------------------------------------------------------------------------------------------------
// K2  min-eigenvalue map.  Tile 64 x 16 outputs per 256-lane workgroup.
//   LDS stage 1: gray tile with a 2-px ring, read straight from the level-0 plane (its REFLECT_101 padding is the
//                reflection cornerMinEigenVal's Sobel wants), as aligned dwords.
//   LDS stage 2: covariance products (Dx^2, DxDy, Dy^2) on the tile + 1-px ring; a lane walks 6 rows of one column
//                and keeps the row filters of the 3 x 3 Sobel in registers.  Positions outside the image take the
//                value AT THE REFLECTED POSITION (boxFilter's BORDER_REFLECT_101 applies to the covariance image,
//                not to the gray image: DxDy changes sign when evaluated on mirrored pixels).
//   Stage 3: 3 x 3 box sums in fp64 (exact, so separable: three-term row sums first), min eigenvalue,
//                per-cell max via LDS, then one global atomicMax per cell and workgroup.
// ------------------------------------------------------------------------------------------------
constexpr int TW = 64, TH = 16;
constexpr int GH = TH + 4, G_PITCH = 72;  // gray tile: columns x0 - 4 .. x0 + 67 (18 aligned dwords), rows y0 - 2 .. y0 + 17
constexpr int CW = TW + 2, CH = TH + 2;   // covariance tile: x0 - 1 .. x0 + 64, y0 - 1 .. y0 + 16

// Dy's row pass: the smoothing taps [1, 2, 1] * scale over (a, b, c).  Canonical: the generic row filter's t = f1 a; t += f0 b;
// t += f1 c.  ROW_FMA (PC_ARITH_SOBEL_ROW_FMA): the same chain fused -- what the 8u -> 32f vector row filter of an
// AVX2-dispatched build computes (v_muladd from a zero accumulator: the first term is a plain product)
template <bool ROW_FMA>
__device__ __forceinline__ float smooth_row(float a, float b, float c, float f1, float f0) {
    float t = f1 * a;
    if (ROW_FMA) {
        t = __fmaf_rn(f0, b, t);
        t = __fmaf_rn(f1, c, t);
    } else {
        t += f0 * b;
        t += f1 * c;
    }
    return t;
}

// SOBEL_FMA (PC_ARITH_SOBEL_FMA) is a template parameter, not a launch argument: the extra code of that mode took the
// kernel from 64 to 106 VGPRs, and a helper wavefront with more than 104 does not fit beside three LK wavefronts per SIMD
// (512 - 3 x 136): the detection then only ran in the gaps of the LK launches -- 4K pipeline 710 -> 537 frames/s, caught by
// the round's last profile run.  tests/test_kernel_resources_cpu.py now holds every helper kernel to its budget.
// SOBEL: bit 0 = the fused column pass of Dx (PC_ARITH_SOBEL_FMA), bit 1 = the fused row pass of Dy (PC_ARITH_SOBEL_ROW_FMA)
template <int SOBEL>
__global__ __launch_bounds__(256) void min_eig_kernel(const uint8_t* __restrict__ img, int pitch, int w, int h,
                                                      float* __restrict__ eig, GfttGrid g,
                                                      uint32_t* __restrict__ cell_max, float f1, float f0, int hi_prio) {
    constexpr bool sobel_fma = (SOBEL & 1) != 0, row_fma = (SOBEL & 2) != 0;
    helper_priority(hi_prio);
    __shared__ __attribute__((aligned(16))) uint8_t s_gray[GH][G_PITCH];
    __shared__ float s_cxx[CH][CW + 1];
    __shared__ float s_cxy[CH][CW + 1];
    __shared__ float s_cyy[CH][CW + 1];
    __shared__ uint32_t s_max[4];

    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    if (tid < 4) s_max[tid] = 0u;

    // stage 1: gray tile.  Rows / columns more than 2 px outside the image are clamped (never consumed).
    {
        const int xmax = pitch - kPadX - 4;
        for (int i = tid; i < GH * (G_PITCH / 4); i += 256) {
            const int r = i / (G_PITCH / 4), d = i - r * (G_PITCH / 4);
            const int yy = min(max(y0 - 2 + r, -2), h + 1), xx = min(x0 - 4 + 4 * d, xmax);
            *reinterpret_cast<uint32_t*>(&s_gray[r][4 * d]) = *reinterpret_cast<const uint32_t*>(img + (ptrdiff_t)yy * pitch + xx);
        }
    }
    __syncthreads();

    // stage 2: lane = (column cx, band of 6 covariance rows); gray rows of covariance row cy are cy .. cy + 2,
    // gray columns of covariance column cx are cx + 2 .. cx + 4 (tile column 0 is x0 - 4)
    if (tid < 3 * CW) {
        const int cx = tid % CW, band = tid / CW;
        const int ax = x0 - 1 + cx;
        float rx[3], ry[3];   // row filters of gray rows cy, cy + 1 (the two carried over), cy + 2
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint8_t* gp = &s_gray[6 * band + k][cx + 2];
            const float a = gp[0], b = gp[1], c = gp[2];
            rx[k] = c - a;
            ry[k] = smooth_row<row_fma>(a, b, c, f1, f0);
        }
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int cy = 6 * band + k;
            const uint8_t* gp = &s_gray[cy + 2][cx + 2];
            const float a = gp[0], b = gp[1], c = gp[2];
            rx[2] = c - a;
            ry[2] = smooth_row<row_fma>(a, b, c, f1, f0);
            // Dx: row [-1,0,1] then column (S0 + S2)*f1 + S1*f0;  Dy: row [1,2,1]*scale then column S2 - S0
            // PC_ARITH_SOBEL_FMA: v_muladd(S0 + S2, k1, S1 * k0) fused, as the AVX2 build of the filter executes it
            const float dx = sobel_fma ? __fmaf_rn(rx[0] + rx[2], f1, rx[1] * f0) : (rx[0] + rx[2]) * f1 + rx[1] * f0;
            const float dy = ry[2] - ry[0];
            const int ay = y0 - 1 + cy;
            const bool in = ax >= 0 && ax < w && ay >= 0 && ay < h;
            s_cxx[cy][cx] = in ? dx * dx : 0.f;
            s_cxy[cy][cx] = in ? dx * dy : 0.f;
            s_cyy[cy][cx] = in ? dy * dy : 0.f;
            rx[0] = rx[1]; rx[1] = rx[2];
            ry[0] = ry[1]; ry[1] = ry[2];
        }
    }
    __syncthreads();
    // covariance ring outside the image <- the reflected (in-image) position; only tiles on the image edge have one
    if (x0 == 0 || y0 == 0 || x0 + TW >= w || y0 + TH >= h) {
        for (int i = tid; i < CW * CH; i += 256) {
            const int cy = i / CW, cx = i - cy * CW;
            const int ax = x0 - 1 + cx, ay = y0 - 1 + cy;
            if ((ax < 0 || ax >= w || ay < 0 || ay >= h) && ax >= -1 && ax <= w && ay >= -1 && ay <= h) {
                const int sx = reflect101(ax, w) - (x0 - 1), sy = reflect101(ay, h) - (y0 - 1);
                if (sx >= 0 && sx < CW && sy >= 0 && sy < CH) {
                    s_cxx[cy][cx] = s_cxx[sy][sx];
                    s_cxy[cy][cx] = s_cxy[sy][sx];
                    s_cyy[cy][cx] = s_cyy[sy][sx];
                }
            }
        }
        __syncthreads();
    }

    // stage 3: lane = (column tx, 4 consecutive rows): six three-term row sums per channel, then the column sums
    const int tx = tid & 63, ty0 = (tid >> 6) * 4;
    const int x = x0 + tx;
    const int cell_x0 = x0 / g.cell_w, cell_y0 = y0 / g.cell_h;
    const bool small_cells = (g.cell_w < TW) || (g.cell_h < TH);
    const int cell_bx = (cell_x0 + 1) * g.cell_w, cell_by = (cell_y0 + 1) * g.cell_h;   // block-uniform
    double hxx[6], hxy[6], hyy[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const int cy = ty0 + k;
        hxx[k] = ((double)s_cxx[cy][tx] + (double)s_cxx[cy][tx + 1]) + (double)s_cxx[cy][tx + 2];
        hxy[k] = ((double)s_cxy[cy][tx] + (double)s_cxy[cy][tx + 1]) + (double)s_cxy[cy][tx + 2];
        hyy[k] = ((double)s_cyy[cy][tx] + (double)s_cyy[cy][tx + 1]) + (double)s_cyy[cy][tx + 2];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int y = y0 + ty0 + k;
        if (x < w && y < h) {
            double sxx = (hxx[k] + hxx[k + 1]) + hxx[k + 2];
            double sxy = (hxy[k] + hxy[k + 1]) + hxy[k + 2];
            double syy = (hyy[k] + hyy[k + 1]) + hyy[k + 2];
            // (With the fused column filter a cancelling Dx leaves a residual of ~1e-10 instead of 0, its square is ~1e-19
            // beside sums of ~1e-3: the fp64 box sums are no longer exact and their ORDER shows in the last bit of one pixel
            // in ~10^5.  The oracle's order is this one: rows left to right, then the row sums top to bottom -- 0.0 + h[k] is
            // exact, so (h[k] + h[k + 1]) + h[k + 2] is that order.)
            const float a = (float)sxx * 0.5f;
            const float b = (float)sxy;
            const float c = (float)syy * 0.5f;
            const float t = a - c;
            float e = (a + c) - sqrtf(t * t + b * b);
#ifdef PC_MINEIG_DUMP   // debugging aid: the covariance tile instead of the response (column x - 1 + PC_MINEIG_DUMP, channel xy)
            e = s_cxy[ty0 + k + 1][tx + PC_MINEIG_DUMP];
#endif
            eig[(size_t)y * w + x] = e;
            const uint32_t key = float_to_ordered(e);
            if (small_cells) {
                const int cx = x / g.cell_w, cy = y / g.cell_h;
                atomicMax(&cell_max[cy * g.cols + cx], key);
            } else {
                // a tile spans at most two cells per axis: the cell of a pixel is a comparison with the next cell
                // boundary, not a division (5 per lane before)
                atomicMax(&s_max[(y >= cell_by ? 2 : 0) + (x >= cell_bx ? 1 : 0)], key);
            }
        }
    }
    __syncthreads();
    if (!small_cells && tid < 4) {
        const int cx = cell_x0 + (tid & 1), cy = cell_y0 + (tid >> 1);
        if (cx < g.cols && cy < g.rows && s_max[tid] != 0u) atomicMax(&cell_max[cy * g.cols + cx], s_max[tid]);
    }
}

// ------------------------------------------------------------------------------------------------
// K2, round 4: the same map without LDS.  A lane owns 2 output columns x ME_R output rows and walks down its band once,
// everything in registers:
//   per gray row (2 aligned dword loads = columns x - 2 .. x + 3): the Sobel row filters of the 4 covariance columns
//       x - 1 .. x + 2 -- the products f1 * g and f0 * g of a pixel are formed once and shared by the three columns that use it;
//   per covariance row (the gray rows above / at / below it are the last three walked): Dx, Dy, their three products for the
//       4 columns, the window rows' sums ((c[j-1] + c[j]) + c[j+1] in fp64) of the 2 output columns;
//   per output row: the three row sums added top to bottom, the eigenvalue, the running maximum of the lane.
// Covariance positions outside the image take the value at the REFLECTED position (see the tiled kernel): a column by a
// register copy inside the lane (-1 <- 1; w <- w - 2 lies at most two columns left of it), a row by substituting the
// other neighbour's row sums in the vertical sum (-1 <- 1, h <- h - 2).  A wavefront covers 64 x (2 * ME_R) pixels; the
// per-cell maximum is ONE atomic per wavefront when its tile lies in one grid cell (wave-uniform test; else four
// registers for the <= 2 x 2 cells a tile can touch, or per-pixel atomics for cells smaller than a tile).
// Measured against the tiled kernel (tools/prep_bench.py, profiles/r04_*_prep.json; SQ_INSTS_VALU per pixel: r04_*_pipeline_by_kernel_pmc.json).
// ------------------------------------------------------------------------------------------------
constexpr int ME_R = 8;                       // output rows per lane
constexpr int ME_TW = 64, ME_TH = 2 * ME_R;   // pixels per wavefront: 32 column pairs x 2 bands

template <int SOBEL>
__global__ __launch_bounds__(256) void min_eig_fused_kernel(const uint8_t* __restrict__ img, int pitch, int w, int h,
                                                            float* __restrict__ eig, GfttGrid g,
                                                            uint32_t* __restrict__ cell_max, float f1, float f0, int hi_prio) {
    helper_priority(hi_prio);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the four wavefronts of a workgroup: 2 x 2 tiles
    const int tx0 = (blockIdx.x * 2 + (wave & 1)) * ME_TW, ty0 = (blockIdx.y * 2 + (wave >> 1)) * ME_TH;
    if (tx0 >= w || ty0 >= h) return;          // wave-uniform
    const int x = tx0 + 2 * (lane & 31), y0 = ty0 + ME_R * (lane >> 5);
    const bool lane_on = x < w && y0 < h;
    // grid cells of the tile's corners (wave-uniform)
    const int cx_lo = tx0 / g.cell_w, cx_hi = min(tx0 + ME_TW - 1, w - 1) / g.cell_w;
    const int cy_lo = ty0 / g.cell_h, cy_hi = min(ty0 + ME_TH - 1, h - 1) / g.cell_h;
    const bool one_cell = cx_lo == cx_hi && cy_lo == cy_hi;
    const bool four_regs = !one_cell && g.cell_w >= ME_TW && g.cell_h >= ME_TH;   // <= 2 cells per axis
    const int bx = (cx_lo + 1) * g.cell_w, by = (cy_lo + 1) * g.cell_h;          // the next cell boundaries
    uint32_t kmax[4] = {0u, 0u, 0u, 0u};
    // does the tile touch the image border?  (wave-uniform: the interior tiles skip every fix-up)
    const bool edge_tile = tx0 == 0 || ty0 == 0 || tx0 + ME_TW + 1 > w || ty0 + ME_TH + 1 > h;

    // the walk, compiled twice: EDGE = false has no fix-up at all (as selects the fix-ups were a third of the kernel's instructions)
    auto walk = [&](auto edge_tag) {
        constexpr bool edge = decltype(edge_tag)::value;
        // columns x - 2 .. x + 3 of a gray row as bytes o .. o + 5 of two aligned dwords
        const int a = (x - 2) & ~3;
        const uint32_t o8 = (uint32_t)(x - 2 - a) * 8u;     // 0 or 16
        const uint8_t* col = img + a;
        // right image edge inside this lane's covariance columns (index 0..3 = columns x - 1 .. x + 2): column w takes w - 2
        const int cw = w - (x - 1);                          // index of column w; 2 or 3 when it matters (x <= w - 1)
        float rx[3][4], ry[3][4];
        // Vertical sums without keeping three rows of row sums: when the row sums h of covariance row ay arrive they
        // complete output row ay - 1 (acc + h, acc = h[ay - 2] + h[ay - 1]) and form the next acc = hprev + h.  Rows outside
        // the image: output 0 is (h[1] + h[0]) + h[1] = (acc + h) + h with acc = h[0] alone; output h - 1 is
        // (h[h - 2] + h[h - 1]) + h[h - 2] = acc + hprev, emitted one step early (at ay = h - 1, where h[h - 2] is still here).
        double hprev[3][2], acc[3][2];
        auto emit = [&](int y, const double (&sv)[3][2]) {
            float e[2];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const float a2 = (float)sv[0][j] * 0.5f;
                const float b2 = (float)sv[1][j];
                const float c2 = (float)sv[2][j] * 0.5f;
                const float t = a2 - c2;
                e[j] = (a2 + c2) - sqrtf(t * t + b2 * b2);
            }
            float* out = eig + (size_t)y * w + x;
            const bool two = x + 1 < w;
            if (two && ((w & 1) == 0)) {
                *reinterpret_cast<float2*>(out) = make_float2(e[0], e[1]);   // y * w + x is even
            } else {
                out[0] = e[0];
                if (two) out[1] = e[1];
            }
            const uint32_t k0 = float_to_ordered(e[0]), k1 = two ? float_to_ordered(e[1]) : 0u;
            if (one_cell) {
                kmax[0] = max(kmax[0], max(k0, k1));
            } else if (four_regs) {
                const int qy = y >= by ? 2 : 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t v0 = (qy + (x >= bx ? 1 : 0)) == q ? k0 : 0u;
                    const uint32_t v1 = (qy + (x + 1 >= bx ? 1 : 0)) == q ? k1 : 0u;
                    kmax[q] = max(kmax[q], max(v0, v1));
                }
            } else {
                atomicMax(&cell_max[(y / g.cell_h) * g.cols + x / g.cell_w], k0);
                if (two) atomicMax(&cell_max[(y / g.cell_h) * g.cols + (x + 1) / g.cell_w], k1);
            }
        };
#pragma unroll
        for (int k = 0; k < ME_R + 4; k++) {
            const int gy = min(y0 - 2 + k, h);               // rows past h feed nothing that is stored
            const uint32_t d0 = *reinterpret_cast<const uint32_t*>(col + (ptrdiff_t)gy * pitch);
            const uint32_t d1 = *reinterpret_cast<const uint32_t*>(col + (ptrdiff_t)gy * pitch + 4);
            const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, o8), hi = d1 >> o8;
            float gv[6];
            gv[0] = (float)(lo & 0xffu);
            gv[1] = (float)((lo >> 8) & 0xffu);
            gv[2] = (float)((lo >> 16) & 0xffu);
            gv[3] = (float)(lo >> 24);
            gv[4] = (float)(hi & 0xffu);
            gv[5] = (float)((hi >> 8) & 0xffu);
            constexpr bool SOBEL_FMA = (SOBEL & 1) != 0, ROW_FMA = (SOBEL & 2) != 0;
            float p1[6], p0[6];
#pragma unroll
            for (int i = 0; i < 6; i++) p1[i] = f1 * gv[i];
            if (!ROW_FMA) {
#pragma unroll
                for (int i = 1; i < 5; i++) p0[i] = f0 * gv[i];
            }
            const int cur = k % 3, prev = (k + 2) % 3, pp = (k + 1) % 3;   // gray rows gy, gy - 1, gy - 2
#pragma unroll
            for (int c = 0; c < 4; c++) {
                rx[cur][c] = gv[c + 2] - gv[c];              // == (0 - g[-1]) + g[+1]: exact either way
                float t = p1[c];                             // t = f1 * s[-1]; t += f0 * s[0]; t += f1 * s[+1]
                if (ROW_FMA) {                               // ... the same chain fused (smooth_row)
                    t = __fmaf_rn(f0, gv[c + 1], t);
                    t = __fmaf_rn(f1, gv[c + 2], t);
                } else {
                    t += p0[c + 1];
                    t += p1[c + 2];
                }
                ry[cur][c] = t;
            }
            if (k < 2) continue;
            // covariance row ay (the middle one of the three gray rows)
            const int ay = y0 + k - 3;
            float cxx[4], cxy[4], cyy[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float dx = SOBEL_FMA ? __fmaf_rn(rx[pp][c] + rx[cur][c], f1, rx[prev][c] * f0)
                                           : (rx[pp][c] + rx[cur][c]) * f1 + rx[prev][c] * f0;
                const float dy = ry[cur][c] - ry[pp][c];
                cxx[c] = dx * dx;
                cxy[c] = dx * dy;
                cyy[c] = dy * dy;
            }
            if (edge) {
                if (x == 0) {                                // column -1 <- column 1
                    cxx[0] = cxx[2];
                    cxy[0] = cxy[2];
                    cyy[0] = cyy[2];
                }
#pragma unroll
                for (int c = 2; c < 4; c++) {                // column w <- column w - 2
                    if (cw == c) {
                        cxx[c] = cxx[c - 2];
                        cxy[c] = cxy[c - 2];
                        cyy[c] = cyy[c - 2];
                    }
                }
            }
            double hc[3][2];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                hc[0][j] = ((double)cxx[j] + (double)cxx[j + 1]) + (double)cxx[j + 2];
                hc[1][j] = ((double)cxy[j] + (double)cxy[j + 1]) + (double)cxy[j + 2];
                hc[2][j] = ((double)cyy[j] + (double)cyy[j + 1]) + (double)cyy[j + 2];
            }
            if (k >= 4) {
                // output row y = ay - 1: its rows y - 1, y are in acc, row y + 1 is hc
                const int y = ay - 1;
                if (y < h && !(edge && y == h - 1 && y > 0)) {
                    double sv[3][2];
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
#pragma unroll
                        for (int j = 0; j < 2; j++) {
                            double v = acc[ch][j] + hc[ch][j];
                            if (edge && y == 0) v = v + hc[ch][j];      // acc holds h[0] alone: (h[0] + h[1]) + h[1]
                            sv[ch][j] = v;
                        }
                    emit(y, sv);
                }
            }
            if (k >= 3) {
                // acc for output row ay = rows ay - 1 (hprev) + ay (hc); a bottom row h - 1 is complete with h[h - 2] = hprev
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
#pragma unroll
                    for (int j = 0; j < 2; j++) acc[ch][j] = (edge && ay == 0) ? hc[ch][j] : hprev[ch][j] + hc[ch][j];
                if (edge && ay == h - 1 && ay > 0) {
                    double sv[3][2];
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
#pragma unroll
                        for (int j = 0; j < 2; j++) sv[ch][j] = acc[ch][j] + hprev[ch][j];
                    emit(ay, sv);
                }
            }
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
#pragma unroll
                for (int j = 0; j < 2; j++) hprev[ch][j] = hc[ch][j];
        }
    };
    if (lane_on) {
        if (edge_tile) walk(std::true_type{});
        else walk(std::false_type{});
    }
    // per-cell maxima of the wavefront
    if (one_cell || four_regs) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (q > 0 && one_cell) break;
            uint32_t v = kmax[q];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
            const int cx = cx_lo + (q & 1), cy = cy_lo + (q >> 1);
            if (lane == 0 && v != 0u && cx < g.cols && cy < g.rows) atomicMax(&cell_max[cy * g.cols + cx], v);
        }
    }
}

// POLYCHASE_MINEIG_VARIANT=1: the LDS-tiled kernel of rounds 1-3 (the cross-check of the fused one; tests/test_env_variants_gpu.py)
static int min_eig_variant() {
    static const int v = [] {
        const char* e = getenv("POLYCHASE_MINEIG_VARIANT");
        return e ? atoi(e) : 0;
    }();
    return v;
}

template <int SOBEL>
static void launch_min_eig_mode(const Level& l0, float* eig, const GfttGrid& g, uint32_t* cell_max, float f1, float f0, hipStream_t s) {
    if (min_eig_variant() == 0) {
        dim3 grid2((l0.w + 2 * ME_TW - 1) / (2 * ME_TW), (l0.h + 2 * ME_TH - 1) / (2 * ME_TH));
        hipLaunchKernelGGL(min_eig_fused_kernel<SOBEL>, grid2, dim3(256), 0, s, l0.img, l0.pitch, l0.w, l0.h, eig, g, cell_max, f1, f0, helper_prio_arg());
        return;
    }
    dim3 grid((l0.w + TW - 1) / TW, (l0.h + TH - 1) / TH);
    hipLaunchKernelGGL(min_eig_kernel<SOBEL>, grid, dim3(256), 0, s, l0.img, l0.pitch, l0.w, l0.h, eig, g, cell_max, f1, f0, helper_prio_arg());
}
void launch_min_eig(const Level& l0, float* eig, const GfttGrid& g, uint32_t* cell_max, int sobel_fma, hipStream_t s) {
    // scale = 1 / (2^(ksize-1) * block_size * 255), folded into the smoothing taps (see oracle)
    const double scale_d = 1.0 / (4.0 * 3.0 * 255.0);
    const float f1 = (float)(1.0 * scale_d), f0 = (float)(2.0 * scale_d);
    switch (sobel_fma & 3) {
        case 0: launch_min_eig_mode<0>(l0, eig, g, cell_max, f1, f0, s); break;
        case 1: launch_min_eig_mode<1>(l0, eig, g, cell_max, f1, f0, s); break;
        case 2: launch_min_eig_mode<2>(l0, eig, g, cell_max, f1, f0, s); break;
        default: launch_min_eig_mode<3>(l0, eig, g, cell_max, f1, f0, s); break;
    }
}

// ------------------------------------------------------------------------------------------------
// K2, general form: any block_size, any gradient_size (Sobel 3 / 5 / 7, Scharr = -1), cornerMinEigenVal or cornerHarris
// (gftt.cc:31-36).  The detector's default (block 3, Sobel 3, min-eig) has the fused kernel above; everything else -- never used
// by the addon -- takes two plain
// kernels with the same arithmetic as oracle/pc_oracle.c: the covariance products of every pixel into three float planes,
// then block x block box sums of those planes in fp64 (BORDER_REFLECT_101 applies to the covariance image, anchor
// block / 2) and the response.  Scale 1 / (2^(ksize-1) * block * 255) folded into the smoothing taps.
// ------------------------------------------------------------------------------------------------
// Taps of the derivative filters (cornerEigenValsVecs, imgproc/corner.cpp; oracle/pc_oracle.c: corner_response): the
// binomial Sobel kernels of getSobelKernels for apertures 3 / 5 / 7 and Scharr's 3 / 10 / 3 for aperture -1, the scale
// 1 / (2^(taps-1) * block * 255) (x 1/2 for Scharr) folded into the SMOOTHING taps as float x float like `kx *= scale`.
struct SobelTaps {
    float sm[7];   // smoothing taps x scale
    float dv[7];   // derivative taps (small integers)
};
static bool make_sobel_taps(int gradient_size, int block_size, SobelTaps* t, int* taps_out) {
    static const int sm3[3] = {1, 2, 1}, dv3[3] = {-1, 0, 1};
    static const int sm5[5] = {1, 4, 6, 4, 1}, dv5[5] = {-1, -2, 0, 2, 1};
    static const int sm7[7] = {1, 6, 15, 20, 15, 6, 1}, dv7[7] = {-1, -4, -5, 0, 5, 4, 1};
    static const int smS[3] = {3, 10, 3};
    if (gradient_size != 3 && gradient_size != 5 && gradient_size != 7 && gradient_size != -1) return false;
    const int taps = gradient_size > 0 ? gradient_size : 3;
    const int* sm = gradient_size == 3 ? sm3 : gradient_size == 5 ? sm5 : gradient_size == 7 ? sm7 : smS;
    const int* dv = gradient_size == 5 ? dv5 : gradient_size == 7 ? dv7 : dv3;
    double scale_d = (double)(1 << (taps - 1)) * block_size;
    if (gradient_size < 0) scale_d *= 2.0;
    scale_d = 1.0 / (scale_d * 255.0);
    const float scale_f = (float)scale_d;
    for (int k = 0; k < 7; k++) {
        t->sm[k] = k < taps ? (float)sm[k] * scale_f : 0.f;
        t->dv[k] = k < taps ? (float)dv[k] : 0.f;
    }
    *taps_out = taps;
    return true;
}

// Covariance products of one pixel for a TAPS x TAPS aperture.  The row pass is sepFilter2D's generic row filter (taps in
// order, one rounding per smoothing tap; bit 1 of sobel_fma: the fused chain of the vector row filter), the column pass the
// symmetric / anti-symmetric column filter's order: centre tap first, then the pairs outwards (bit 0 of sobel_fma: each
// v_muladd fused, the AVX2 dispatch).  Borders by index reflection (BORDER_REFLECT_101): no reliance on the plane's padding,
// whose width follows the LK window.
template <int TAPS>
__global__ __launch_bounds__(256) void cov_kernel(const uint8_t* __restrict__ img, int pitch, int w, int h, float* __restrict__ cov,
                                                  SobelTaps T, int sobel_fma, int hi_prio) {
    helper_priority(hi_prio);
    constexpr int R = TAPS / 2;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    int xs[TAPS];
#pragma unroll
    for (int i = 0; i < TAPS; i++) xs[i] = reflect101(x + i - R, w);
    float rdx[TAPS], rdy[TAPS];
#pragma unroll
    for (int k = 0; k < TAPS; k++) {
        const uint8_t* row = img + (ptrdiff_t)reflect101(y + k - R, h) * pitch;
        float d = 0.f, t = 0.f;
#pragma unroll
        for (int i = 0; i < TAPS; i++) {
            const float v = row[xs[i]];
            if (i == 0) {
                d = T.dv[0] * v;
                t = T.sm[0] * v;
            } else {
                d += T.dv[i] * v;
                t = (sobel_fma & 2) ? __fmaf_rn(T.sm[i], v, t) : t + T.sm[i] * v;
            }
        }
        rdx[k] = d;
        rdy[k] = t;
    }
    float dx = rdx[R] * T.sm[R];
#pragma unroll
    for (int j = 1; j <= R; j++) {
        const float pair = rdx[R + j] + rdx[R - j];
        dx = (sobel_fma & 1) ? __fmaf_rn(pair, T.sm[R + j], dx) : pair * T.sm[R + j] + dx;
    }
    float dy;
    if (TAPS == 3) {
        dy = rdy[2] - rdy[0];
    } else {
        dy = T.dv[R + 1] * (rdy[R + 1] - rdy[R - 1]);
#pragma unroll
        for (int j = 2; j <= R; j++) dy = (rdy[R + j] - rdy[R - j]) * T.dv[R + j] + dy;
    }
    const size_t n = (size_t)w * h, i = (size_t)y * w + x;
    cov[i] = dx * dx;
    cov[n + i] = dx * dy;
    cov[2 * n + i] = dy * dy;
}

// Large blocks: the row sums of the box filter once per pixel (RowSum), three fp64 planes; box_response_kernel then adds
// `block` of them per pixel (ColumnSum) instead of block^2 products -- the same additions in the same order, the same bits.
__global__ __launch_bounds__(256) void box_rows_kernel(const float* __restrict__ cov, int w, int h, int block, double* __restrict__ rows, int hi_prio) {
    helper_priority(hi_prio);
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const size_t n = (size_t)w * h, row = (size_t)y * w;
    const int a0 = block / 2;
    double rxx = 0, rxy = 0, ryy = 0;
    for (int i = 0; i < block; i++) {
        const size_t at = row + reflect101(x + i - a0, w);
        rxx += (double)cov[at];
        rxy += (double)cov[n + at];
        ryy += (double)cov[2 * n + at];
    }
    rows[row + x] = rxx;
    rows[n + row + x] = rxy;
    rows[2 * n + row + x] = ryy;
}

__global__ __launch_bounds__(256) void box_response_kernel(const float* __restrict__ cov, const double* __restrict__ rows, int w, int h, int block, int harris, double harris_k,
                                                           float* __restrict__ eig, GfttGrid g, uint32_t* __restrict__ cell_max, int hi_prio) {
    helper_priority(hi_prio);
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const size_t n = (size_t)w * h;
    const int a0 = block / 2;
    // the oracle's order (RowSum then ColumnSum): each window row left to right, the row sums top to bottom.  Exact for
    // 8-bit inputs in the canonical arithmetic; under PC_ARITH_SOBEL_FMA the order defines the last bit of ~1 pixel in 10^5.
    double sxx = 0, sxy = 0, syy = 0;
    if (rows) {
        for (int j = 0; j < block; j++) {
            const size_t at = (size_t)reflect101(y + j - a0, h) * w + x;
            sxx += rows[at];
            sxy += rows[n + at];
            syy += rows[2 * n + at];
        }
    } else {
        for (int j = 0; j < block; j++) {
            const size_t row = (size_t)reflect101(y + j - a0, h) * w;
            double rxx = 0, rxy = 0, ryy = 0;
            for (int i = 0; i < block; i++) {
                const size_t at = row + reflect101(x + i - a0, w);
                rxx += (double)cov[at];
                rxy += (double)cov[n + at];
                ryy += (double)cov[2 * n + at];
            }
            sxx += rxx;
            sxy += rxy;
            syy += ryy;
        }
    }
    float e;
    if (harris) {
        // calcHarris' scalar expression: (float)(a * c - b * b - k * (a + c) * (a + c)), the k term in double; harris == 2
        // (PC_ARITH_SOBEL_FMA, the detector as an x86 build executes it): the first w / 4 * 4 columns of a row go through
        // calcHarris' vector loop -- all in float with (float)k, no fused operations (the build has -ffp-contract=off)
        const float a = (float)sxx, b = (float)sxy, c = (float)syy;
        const float ac = a * c, bb = b * b;
        const float det = ac - bb, tr = a + c;
        if (harris == 2 && x < (w / 4) * 4) {
            const float kt = (float)harris_k * tr;
            e = det - kt * tr;
        } else {
            e = (float)((double)det - harris_k * (double)tr * (double)tr);
        }
    } else {
        const float a = (float)sxx * 0.5f, b = (float)sxy, c = (float)syy * 0.5f;
        const float t = a - c;
        e = (a + c) - sqrtf(t * t + b * b);
    }
    eig[(size_t)y * w + x] = e;
    // per-cell maximum: one atomic per wavefront when its 64 pixels of a row lie in one cell
    uint32_t key = float_to_ordered(e);
    const int cell = (y / g.cell_h) * g.cols + x / g.cell_w;
    const int cell0 = __builtin_amdgcn_readfirstlane(cell);
    if (__all(cell == cell0)) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) key = max(key, (uint32_t)__shfl_xor((int)key, d));
        if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) atomicMax(&cell_max[cell0], key);
    } else {
        atomicMax(&cell_max[cell], key);
    }
}

bool launch_corner_response(const Level& l0, float* eig, float* cov, double* box_rows, const GfttGrid& g, uint32_t* cell_max, int block_size, int gradient_size,
                            bool harris, double harris_k, int sobel_fma, hipStream_t s) {
    SobelTaps T;
    int taps = 0;
    if (!make_sobel_taps(gradient_size, block_size, &T, &taps)) return false;
    dim3 grid((l0.w + 63) / 64, (l0.h + 3) / 4);
    if (taps == 3)
        hipLaunchKernelGGL(cov_kernel<3>, grid, dim3(256), 0, s, l0.img, l0.pitch, l0.w, l0.h, cov, T, sobel_fma, helper_prio_arg());
    else if (taps == 5)
        hipLaunchKernelGGL(cov_kernel<5>, grid, dim3(256), 0, s, l0.img, l0.pitch, l0.w, l0.h, cov, T, sobel_fma, helper_prio_arg());
    else
        hipLaunchKernelGGL(cov_kernel<7>, grid, dim3(256), 0, s, l0.img, l0.pitch, l0.w, l0.h, cov, T, sobel_fma, helper_prio_arg());
    if (box_rows) hipLaunchKernelGGL(box_rows_kernel, grid, dim3(256), 0, s, cov, l0.w, l0.h, block_size, box_rows, helper_prio_arg());
    hipLaunchKernelGGL(box_response_kernel, grid, dim3(256), 0, s, cov, (const double*)box_rows, l0.w, l0.h, block_size, harris ? ((sobel_fma & 1) ? 2 : 1) : 0, harris_k, eig, g, cell_max,
                       helper_prio_arg());
    return true;
}

// ------------------------------------------------------------------------------------------------
// K3  threshold (per cell of the NEIGHBOUR) + 3x3 dilate + local-max test.  A workgroup walks a 64 x 64 block as four
// 64 x 16 tiles of the min-eig map, each staged (already thresholded) in LDS with a 1-px ring; 4 pixels per lane.
// Writes the candidate byte of EVERY pixel (so the map needs no clearing between frames) and appends keys =
// ordered(value) << 32 | (y*w + x) with ONE global atomic per workgroup: appends to one counter from different CUs
// cost ~23 ns each and serialise (a 64 x 16 workgroup per atomic took 0.19 ms at 4K for that reason alone).
// ------------------------------------------------------------------------------------------------
constexpr int NMS_SUB = 4;   // 64 x 16 tiles per workgroup

// Bucket of a candidate for the sort (K4): the ordered-uint image of a float is monotone and piecewise linear in
// log(value), so equal slices of it hold similar numbers of corner responses (measured on the benchmark clips with
// 8192 buckets: at most 62 candidates per bucket at 1080p, 222 at 4K).  Bucket 0 holds the LARGEST values.
struct SortRange {
    uint32_t hi, shift;
};
__host__ __device__ __forceinline__ SortRange make_sort_range(uint32_t lo, uint32_t hi) {
    SortRange r;
    r.hi = hi;
    r.shift = 0;
    const uint32_t span = hi > lo ? hi - lo : 0u;
    while ((span >> r.shift) >= (uint32_t)kSortBuckets) r.shift++;
    return r;
}
__device__ __forceinline__ uint32_t bucket_of(uint32_t ord, const SortRange& r) {
    const uint32_t d = r.hi > ord ? r.hi - ord : 0u;
    const uint32_t b = d >> r.shift;
    return b < (uint32_t)kSortBuckets ? b : (uint32_t)kSortBuckets - 1u;
}

__global__ __launch_bounds__(256) void nms_kernel(const float* __restrict__ eig, int w, int h, GfttGrid g,
                                                  const uint32_t* __restrict__ cell_max, double quality_level,
                                                  unsigned long long* __restrict__ keys, uint32_t cap,
                                                  uint32_t* __restrict__ counter, uint8_t* __restrict__ cstate,
                                                  uint32_t* __restrict__ sort_params, uint32_t* __restrict__ hist,
                                                  uint32_t* __restrict__ ticket, uint32_t* __restrict__ bucket_offsets,
                                                  uint32_t* __restrict__ bin_hist, int n_tiles, int hi_prio) {
    helper_priority(hi_prio);
    __shared__ float s_thr[kMaxGridCells];
    __shared__ uint32_t s_hi, s_lo;
    __shared__ float s_v[CH][CW + 2];
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_base;
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TW;
    const int ncells = g.rows * g.cols;
    if (tid == 0) {
        s_hi = 0u;
        s_lo = 0xffffffffu;
    }
    __syncthreads();
    for (int i = tid; i < ncells; i += 256) {
        // cv::threshold on CV_32F compares with (float)(maxVal * quality_level), maxVal a double
        const float mx = ordered_to_float(cell_max[i]);
        s_thr[i] = (float)((double)mx * quality_level);
        // value range of the candidates (bucket sort, see bucket_of): above the smallest threshold, up to the largest maximum
        atomicMax(&s_hi, cell_max[i]);
        atomicMin(&s_lo, float_to_ordered(s_thr[i] > 0.f ? s_thr[i] : 0.f));
    }
    __syncthreads();
    const SortRange range = make_sort_range(s_lo, s_hi);
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        if (tid == 0) {
            sort_params[0] = range.hi;
            sort_params[1] = range.shift;
        }
        // the tile histogram the compaction counts the keypoints into (LK visiting order)
        if (bin_hist)
            for (int i = tid; i < n_tiles; i += 256) bin_hist[i] = 0u;
    }
    const int r = tid >> 4, q = tid & 15;
    const int x = x0 + 4 * q;
    uint32_t flags = 0;                 // bit 4 * sub + i: pixel i of this lane's quad in tile `sub` is a candidate
    float vals[NMS_SUB][4];
#pragma unroll
    for (int sub = 0; sub < NMS_SUB; sub++) {
        const int y0 = (blockIdx.y * NMS_SUB + sub) * TH;
        const int y = y0 + r;
        __syncthreads();   // s_thr is complete / the previous tile's readers are done
        // thresholded values; outside the image: 0 (cv::dilate ignores those positions, and a candidate is > 0).
        // A tile + ring spans at most two grid cells per axis when the cells are at least that large: the cell of a
        // position is then a comparison with the next cell boundary, not a division.
        const bool big_cells = g.cell_w >= CW && g.cell_h >= CH;
        const int ccx0 = max(x0 - 1, 0) / g.cell_w, ccy0 = max(y0 - 1, 0) / g.cell_h;     // block-uniform
        const int bx = (ccx0 + 1) * g.cell_w, by = (ccy0 + 1) * g.cell_h;
        auto thr_at = [&](int ax, int ay) -> float {
            const int cx = big_cells ? ccx0 + (ax >= bx) : ax / g.cell_w;
            const int cy = big_cells ? ccy0 + (ay >= by) : ay / g.cell_h;
            return s_thr[cy * g.cols + cx];
        };
        if (y0 < h) {
            // interior of the tile: one aligned float4 per lane where the row length allows it
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (y < h && x < w) {
                const float* row = eig + (size_t)y * w;
                if (x + 3 < w && (w & 3) == 0) {
                    const float4 e = *reinterpret_cast<const float4*>(row + x);
                    v[0] = e.x; v[1] = e.y; v[2] = e.z; v[3] = e.w;
                } else {
                    for (int i = 0; i < 4 && x + i < w; i++) v[i] = row[x + i];
                }
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = (x + i < w && v[i] > thr_at(x + i, y)) ? v[i] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; i++) s_v[r + 1][4 * q + 1 + i] = v[i];
            // ring: top and bottom rows (CW each), left and right columns (TH each)
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
                float e = 0.f;
                if (ax >= 0 && ax < w && ay >= 0 && ay < h) {
                    e = eig[(size_t)ay * w + ax];
                    e = (e > thr_at(ax, ay)) ? e : 0.f;
                }
                s_v[cy][cx] = e;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; i++) vals[sub][i] = 0.f;
        if (y < h && x < w) {
            // column maxima of the three rows for columns x - 1 .. x + 4
            float cm[6];
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const float a = s_v[r][4 * q + i], b = s_v[r + 1][4 * q + i], c = s_v[r + 2][4 * q + i];
                const float m = a > b ? a : b;
                cm[i] = m > c ? m : c;
            }
            uint32_t f4 = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float val = s_v[r + 1][4 * q + 1 + i];
                float m = cm[i] > cm[i + 1] ? cm[i] : cm[i + 1];
                m = m > cm[i + 2] ? m : cm[i + 2];
                const int xi = x + i;
                if (val != 0.f && val == m && xi >= 1 && xi < w - 1 && y >= 1 && y < h - 1) {
                    f4 |= 1u << i;
                    vals[sub][i] = val;
                }
            }
            flags |= f4 << (4 * sub);
            uint8_t* cs = cstate + (size_t)y * w + x;
            if (x + 3 < w && ((w & 3) == 0)) {
                *reinterpret_cast<uint32_t*>(cs) = (f4 & 1u) | ((f4 & 2u) << 7) | ((f4 & 4u) << 14) | ((f4 & 8u) << 21);
            } else {
                for (int i = 0; i < 4 && x + i < w; i++) cs[i] = (uint8_t)((f4 >> i) & 1u);
            }
        }
    }
    // workgroup-aggregated append
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t cnt = (uint32_t)__popc(flags);
    uint32_t incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    if (tid == 0) {
        const uint32_t total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        s_base = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    if (cnt != 0) {
        uint32_t pos = s_base + incl - cnt;
        for (int wv = 0; wv < wave; wv++) pos += s_wave[wv];
#pragma unroll
        for (int sub = 0; sub < NMS_SUB; sub++) {
            const int y = (blockIdx.y * NMS_SUB + sub) * TH + r;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (flags & (1u << (4 * sub + i))) {
                    const uint32_t ord = float_to_ordered(vals[sub][i]);
                    if (pos < cap) {
                        keys[pos] = ((unsigned long long)ord << 32) | (unsigned long long)(uint32_t)(y * w + x + i);
                        atomicAdd(&hist[bucket_of(ord, range)], 1u);
                    }
                    pos++;
                }
            }
        }
    }
    // K4, first step: the workgroup that finishes last turns the bucket counts into bucket offsets (descending value)
    if (ticket && last_workgroup(ticket, gridDim.x * gridDim.y)) {
        __shared__ uint32_t s_scan[256];
        const uint32_t total = scan_exclusive_256_fixed<kSortBuckets / 256>(hist, bucket_offsets, s_scan);
        if (tid == 0) bucket_offsets[kSortBuckets] = total;
    }
}

void launch_nms(const float* eig, int w, int h, const GfttGrid& g, const uint32_t* cell_max, double quality_level,
                unsigned long long* keys, uint32_t cap, uint32_t* counter, uint8_t* cstate, uint32_t* sort_params, uint32_t* hist,
                uint32_t* ticket, uint32_t* bucket_offsets, uint32_t* bin_hist, hipStream_t s) {
    dim3 grid((w + TW - 1) / TW, (h + NMS_SUB * TH - 1) / (NMS_SUB * TH));
    hipLaunchKernelGGL(nms_kernel, grid, dim3(256), 0, s, eig, w, h, g, cell_max, quality_level, keys, cap, counter, cstate,
                       sort_params, hist, ticket, bucket_offsets, bin_hist, bin_hist ? bin_num_tiles(w, h) : 0, helper_prio_arg());
}

// ------------------------------------------------------------------------------------------------
// K4  sort of the candidates, descending (value, address) = the processing order of the greedy loop (gftt.cc:7-12, :98),
// without the candidate count on the host: K3 has counted the candidates per value bucket; here an exclusive scan of
// the 8192 counts, a scatter into bucket order and a rank sort inside every bucket (one wavefront per bucket, keys in
// LDS).  Three launches; rocPRIM's sort of the same keys is 9 (1080p) to 18 (4K) launches and needs the count on the
// host, which tied the ordering phase of detection to a host round trip.  A bucket with more keys than fit its LDS
// buffer raises the overflow flag: the caller then sorts with rocPRIM (sort_keys_desc).
// ------------------------------------------------------------------------------------------------
constexpr int kBucketLds = 512;    // keys a bucket may hold on the fast path (4 KB of LDS; measured maximum 222 at 4K)

__global__ __launch_bounds__(256) void bucket_scatter_kernel(const unsigned long long* __restrict__ keys, uint32_t cap,
                                                             const uint32_t* __restrict__ counter, const uint32_t* __restrict__ sort_params,
                                                             const uint32_t* __restrict__ offsets, uint32_t* __restrict__ cursor,
                                                             unsigned long long* __restrict__ out, int hi_prio) {
    helper_priority(hi_prio);
    const uint32_t n = min(*counter, cap);
    SortRange range;
    range.hi = sort_params[0];
    range.shift = sort_params[1];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned long long key = keys[i];
        const uint32_t b = bucket_of((uint32_t)(key >> 32), range);
        out[offsets[b] + atomicAdd(&cursor[b], 1u)] = key;
    }
}

constexpr int kBucketsPerWave = 4;

// Four wavefronts per workgroup, each on its own four consecutive buckets in its own 4 KB of LDS (no barrier between them; 16 KB
// per workgroup stays inside the helpers' LDS budget beside the LK wavefronts, which hold 120 of a CU's 160 KB).  (Rounds 2-4:
// one wavefront per workgroup = 2048 workgroups per launch.)
constexpr int kSortWaves = 4;
__global__ __launch_bounds__(64 * kSortWaves) void bucket_sort_kernel(const unsigned long long* __restrict__ in, const uint32_t* __restrict__ offsets,
                                                         unsigned long long* __restrict__ out, uint32_t* __restrict__ overflow, int hi_prio) {
    helper_priority(hi_prio);
    __shared__ unsigned long long s_all[kSortWaves][kBucketLds];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long* const s_keys = s_all[wave];
    const int b0 = (blockIdx.x * kSortWaves + wave) * kBucketsPerWave;
    if (b0 >= kSortBuckets) return;
    uint32_t o[kBucketsPerWave + 1];
#pragma unroll
    for (int k = 0; k <= kBucketsPerWave; k++) o[k] = offsets[min(b0 + k, kSortBuckets)];
    const uint32_t lo = o[0], n4 = o[kBucketsPerWave] - lo;
    if (n4 == 0) return;   // nothing in these buckets
    if (n4 <= (uint32_t)kBucketLds) {
        // The four buckets TOGETHER (round 5): one trip to memory for all their keys, and every lane ranks a key inside ITS
        // bucket -- at 4K a bucket holds 43 keys on average, so a bucket at a time left a third of the lanes idle and paid
        // four dependent memory latencies per wavefront (158 us per frame, 476 at worst: profiles/r04_c3_rocprofv3_kernel_stats.csv).
        // (all loads of a lane in flight together: kBucketLds / 64 of them at most)
        unsigned long long ld[kBucketLds / 64];
#pragma unroll
        for (int q = 0; q < kBucketLds / 64; q++) {
            const uint32_t i = lane + 64 * q;
            ld[q] = i < n4 ? in[lo + i] : 0ull;
        }
#pragma unroll
        for (int q = 0; q < kBucketLds / 64; q++) {
            const uint32_t i = lane + 64 * q;
            if (i < n4) s_keys[i] = ld[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        uint32_t longest = 0;
#pragma unroll
        for (int k = 0; k < kBucketsPerWave; k++) longest = max(longest, o[k + 1] - o[k]);
        for (uint32_t i = lane; i < n4; i += 64) {
            const unsigned long long mine = s_keys[i];
            uint32_t a = lo, b = o[1];
#pragma unroll
            for (int k = 1; k < kBucketsPerWave; k++)
                if (lo + i >= o[k]) {
                    a = o[k];
                    b = o[k + 1];
                }
            // The rank of the key inside its bucket.  A UNIFORM trip count (the longest of the four buckets; a lane past the end
            // of its own re-reads its last key and adds nothing) so that the loop unrolls and eight LDS reads are in flight at a
            // time: with per-lane bounds every iteration sat out one LDS latency -- the 222-key buckets of a 4K frame took 100 us
            // that way, in a kernel whose comparisons are worth 5.
            const uint32_t first = a - lo, len = b - a;
            uint32_t rank = 0;
#pragma unroll 8
            for (uint32_t j = 0; j < longest; j++) {
                const unsigned long long other = s_keys[first + min(j, len - 1u)];
                rank += (j < len && other > mine) ? 1u : 0u;   // keys are distinct (the address part)
            }
            out[a + rank] = mine;
        }
        return;
    }
    // more keys than the buffer holds at once: bucket by bucket
    for (int b = b0; b < b0 + kBucketsPerWave && b < kSortBuckets; b++) {
        const uint32_t base = offsets[b], n = offsets[b + 1] - base;
        if (n == 0) continue;
        if (n > (uint32_t)kBucketLds) {
            // beyond the fast path: the caller redoes the frame; the kernels queued behind this one still run, so they
            // must find VALID keys (copied unsorted), not whatever the buffer held before
            if (lane == 0) atomicOr(overflow, 1u);
            for (uint32_t i = lane; i < n; i += 64) out[base + i] = in[base + i];
            continue;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the previous bucket's readers are done (one wavefront: program order)
        for (uint32_t i = lane; i < n; i += 64) s_keys[i] = in[base + i];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (uint32_t i = lane; i < n; i += 64) {
            const unsigned long long mine = s_keys[i];
            uint32_t rank = 0;
#pragma unroll 8
            for (uint32_t j = 0; j < n; j++) rank += s_keys[j] > mine ? 1u : 0u;   // keys are distinct (the address part)
            out[base + rank] = mine;
        }
    }
}

void launch_bucket_sort(const unsigned long long* keys, uint32_t cap, uint32_t n_launch, const uint32_t* counter, const uint32_t* sort_params,
                        const uint32_t* offsets, uint32_t* cursor, unsigned long long* scratch, unsigned long long* out,
                        uint32_t* overflow, hipStream_t s) {
    const unsigned blocks = std::max(1u, std::min<unsigned>(1024u, (n_launch + 255u) / 256u));
    hipLaunchKernelGGL(bucket_scatter_kernel, dim3(blocks), dim3(256), 0, s, keys, cap, counter, sort_params, offsets, cursor, scratch, helper_prio_arg());
    hipLaunchKernelGGL(bucket_sort_kernel, dim3((kSortBuckets / kBucketsPerWave + kSortWaves - 1) / kSortWaves), dim3(64 * kSortWaves), 0, s, scratch, offsets,
                       out, overflow, helper_prio_arg());
}

// ------------------------------------------------------------------------------------------------
// K5  exact min-distance suppression on the GPU, candidates in PRIORITY ORDER.
// The reference's greedy loop (gftt.cc:100-164) accepts a candidate iff no ALREADY ACCEPTED candidate lies within
// min_distance; processing order = (value desc, address desc) = the order of `keys` here.  Equivalently: a candidate
// is accepted iff every higher-priority candidate within the radius is rejected -- the lexicographically-first
// maximal independent set of the conflict graph.  Lane i owns candidate i and re-evaluates until all its
// higher-priority neighbours are decided; decisions are final and monotone, so asynchronous evaluation reaches
// exactly the sequential result.
// Progress under ANY occupancy: lane i only ever waits for candidates j < i, i.e. for its own or LOWER-numbered
// workgroups.  Workgroups are handed out in index order (per XCD), so the lowest-numbered unfinished workgroup is
// always running and never waits on an unfinished one: no assumption that the grid is resident (round 1 sized the
// grid to "what fits" and relied on it).  The spin bound stays as a tripwire.
// States (one byte per pixel, agent-scope relaxed atomics: per-XCD L2s are not coherent): see the header.
// Afterwards accepted_per_block[b] = accepted candidates of workgroup b (input of the ordered compaction).
// ------------------------------------------------------------------------------------------------
// The tail of the suppression kernel: its last workgroup turns the per-workgroup counts of accepted candidates into
// their exclusive scan (input of the ordered compaction) and leaves the keypoint count (truncated to max_corners,
// gftt.cc:160-162) on the device.
struct AcceptedScan {
    uint32_t* ticket;       // zero before the launch
    uint32_t* n_out;        // keypoint count
    uint32_t* overflow;     // fast path: bit 4 = more candidates than the launches of this detection cover
    uint32_t max_corners;
};
__device__ __forceinline__ void finish_accepted_scan(uint32_t* per_block, const AcceptedScan& fin, uint32_t n_max, const uint32_t* n_dev) {
    if (!last_workgroup(fin.ticket, gridDim.x)) return;
    __shared__ uint32_t s_scan[256];
    // the launches of this detection cover n_max candidates (sized from the previous frames' counts): more than that
    // were not all processed -- flag it, the caller redoes the frame
    if (threadIdx.x == 0 && n_dev && fin.overflow && *n_dev > n_max) atomicOr(fin.overflow, 4u);
    const uint32_t total = scan_exclusive_256(per_block, per_block, (int)gridDim.x, s_scan);
    if (threadIdx.x == 0) *fin.n_out = (fin.max_corners > 0 && total > fin.max_corners) ? fin.max_corners : total;
}

constexpr uint8_t CS_CAND = 1, CS_ACCEPTED = 2, CS_REJECTED = 3;
constexpr int SUP_NB = 12;  // higher-priority neighbours cached in registers
constexpr int SUP_BLOCK = 256;

__device__ __forceinline__ uint8_t cs_load(const uint8_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cs_store(uint8_t* p, uint8_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(SUP_BLOCK) void suppress_sorted_kernel(const unsigned long long* __restrict__ keys, uint32_t n_max,
                                                                    const uint32_t* __restrict__ n_dev, int w, int h,
                                                                    const float* __restrict__ eig, uint8_t* cstate,
                                                                    const int2* __restrict__ offsets, int n_offsets,
                                                                    const int* __restrict__ row_hw, int R,
                                                                    uint32_t* __restrict__ accepted_per_block,
                                                                    uint32_t* __restrict__ stuck, AcceptedScan fin, int hi_prio) {
    helper_priority(hi_prio);
    const uint32_t n = n_dev ? min(*n_dev, n_max) : n_max;   // the launch covers n_max; workgroups past the count leave at once
    const uint32_t i = blockIdx.x * SUP_BLOCK + threadIdx.x;
    const bool live = i < n;
    uint32_t my_val = 0, my_idx = 0;
    uint32_t nb[SUP_NB];
    int n_nb = 0;
    bool overflow = false;
    int x = 0, y = 0;
    if (live) {
        const unsigned long long key = keys[i];
        my_val = (uint32_t)(key >> 32);
        my_idx = (uint32_t)key;
        y = (int)(my_idx / (uint32_t)w);
        x = (int)(my_idx - (uint32_t)y * (uint32_t)w);
        // The neighbourhood (the offsets table: dx^2 + dy^2 < min_distance^2) row by row: row dy spans |dx| <= row_hw[dy + R].
        // Its state bytes are read as aligned dwords -- ~40 independent loads per lane for min_distance 5 where the
        // offset-by-offset walk made 80 byte loads, each behind a branch.  Plain loads: "is a candidate" (non-zero) never
        // changes during this kernel, and a byte that changes 1 -> 2 / 3 under the load stays non-zero.
        for (int dy = -R; dy <= R; dy++) {
            const int ny = y + dy, hw = row_hw[dy + R];
            if (ny < 0 || ny >= h || hw < 0) continue;
            const uint32_t lin0 = (uint32_t)(ny * w + max(x - hw, 0)), lin1 = (uint32_t)(ny * w + min(x + hw, w - 1));
            for (uint32_t a = lin0 & ~3u; a <= lin1; a += 4u) {
                uint32_t v = *reinterpret_cast<const uint32_t*>(cstate + a);
                // bytes of this dword outside [lin0, lin1]
                if (a < lin0) v &= 0xffffffffu << (8u * (lin0 - a));
                if (a + 3u > lin1) v &= 0xffffffffu >> (8u * (a + 3u - lin1));
                while (v) {
                    const uint32_t b = (uint32_t)(__ffs((int)v) - 1) >> 3;
                    v &= ~(0xffu << (8u * b));
                    const uint32_t nidx = a + b;
                    if (nidx == my_idx) continue;
                    const uint32_t nval = float_to_ordered(eig[nidx]);
                    if (!(nval > my_val || (nval == my_val && nidx > my_idx))) continue;  // lower priority
                    if (n_nb < SUP_NB) {
#pragma unroll
                        for (int k = 0; k < SUP_NB; k++)
                            if (k == n_nb) nb[k] = nidx;
                        n_nb++;
                    } else {
                        overflow = true;
                    }
                }
            }
        }
    }
    // NOTE on control flow: a lane must publish its decision INSIDE the loop and keep iterating (idle) until the
    // whole wave is done.  With a per-lane exit the store would sit in the loop's exit block, which a wave only
    // executes after ALL its lanes left the loop -- a lane waiting for a neighbour owned by the same wave would then
    // never see it (SIMT deadlock).
    bool done = !live;
    bool accepted = false;
    for (uint32_t spin = 0;; spin++) {
        if (!done) {
            bool blocked = false, rejected = false;
            if (overflow) {
                for (int o = 0; o < n_offsets; o++) {
                    const int nx = x + offsets[o].x, ny = y + offsets[o].y;
                    if (nx < 0 || nx >= w || ny < 0 || ny >= h) continue;
                    const uint32_t nidx = (uint32_t)(ny * w + nx);
                    const uint8_t st = cs_load(&cstate[nidx]);
                    if (st == 0) continue;
                    const uint32_t nval = float_to_ordered(eig[nidx]);
                    if (!(nval > my_val || (nval == my_val && nidx > my_idx))) continue;
                    rejected |= (st == CS_ACCEPTED);
                    blocked |= (st == CS_CAND);
                }
            } else {
#pragma unroll
                for (int k = 0; k < SUP_NB; k++) {
                    if (k < n_nb) {
                        const uint8_t st = cs_load(&cstate[nb[k]]);
                        rejected |= (st == CS_ACCEPTED);
                        blocked |= (st == CS_CAND);
                    }
                }
            }
            if (rejected || !blocked) {
                cs_store(&cstate[my_idx], rejected ? CS_REJECTED : CS_ACCEPTED);
                accepted = !rejected;
                done = true;
            }
        }
        if (__all(done)) break;           // wave-uniform exit
        if (spin > (1u << 22)) {          // tripwire: report instead of hanging the GPU
            if (!done) atomicAdd(stuck, 1u);
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    const int c = __syncthreads_count(accepted);
    if (threadIdx.x == 0) publish(&accepted_per_block[blockIdx.x], (uint32_t)c);
    finish_accepted_scan(accepted_per_block, fin, n_max, n_dev);
}

// no suppression (min_distance < 1, gftt.cc:165-181): every candidate is accepted
__global__ __launch_bounds__(SUP_BLOCK) void accept_all_kernel(const unsigned long long* __restrict__ keys, uint32_t n_max,
                                                               const uint32_t* __restrict__ n_dev, uint8_t* cstate,
                                                               uint32_t* __restrict__ accepted_per_block, AcceptedScan fin, int hi_prio) {
    helper_priority(hi_prio);
    const uint32_t n = n_dev ? min(*n_dev, n_max) : n_max;
    const uint32_t i = blockIdx.x * SUP_BLOCK + threadIdx.x;
    if (i < n) cstate[(uint32_t)keys[i]] = CS_ACCEPTED;
    const uint32_t first = blockIdx.x * SUP_BLOCK;
    if (threadIdx.x == 0) publish(&accepted_per_block[blockIdx.x], first < n ? min((uint32_t)SUP_BLOCK, n - first) : 0u);
    finish_accepted_scan(accepted_per_block, fin, n_max, n_dev);
}

// accepted candidates, in priority order = acceptance order of the greedy loop -> Point2f((float)x, (float)y) (gftt.cc:157)
__global__ __launch_bounds__(SUP_BLOCK) void accepted_scatter_kernel(const unsigned long long* __restrict__ keys, uint32_t n_max,
                                                                     const uint32_t* __restrict__ n_dev, int w,
                                                                     const uint8_t* __restrict__ cstate,
                                                                     const uint32_t* __restrict__ block_offset, uint32_t max_corners,
                                                                     float2* __restrict__ xy, uint32_t* __restrict__ bin_hist, int tiles_x,
                                                                     int n_tiles, uint32_t* __restrict__ ticket, int hi_prio) {
    helper_priority(hi_prio);
    __shared__ uint32_t s_wave[SUP_BLOCK / 64];
    const uint32_t n = n_dev ? min(*n_dev, n_max) : n_max;
    const uint32_t i = blockIdx.x * SUP_BLOCK + threadIdx.x;
    uint32_t idx = 0;
    bool keep = false;
    if (i < n) {
        idx = (uint32_t)keys[i];
        keep = cstate[idx] == CS_ACCEPTED;
    }
    const unsigned long long ballot = __ballot(keep);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_wave[wave] = (uint32_t)__popcll(ballot);
    __syncthreads();
    if (keep) {
        uint32_t pos = block_offset[blockIdx.x] + (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull));
        for (int wv = 0; wv < wave; wv++) pos += s_wave[wv];
        if (!(max_corners > 0 && pos >= max_corners)) {
            const uint32_t y = idx / (uint32_t)w, x = idx - y * (uint32_t)w;
            xy[pos] = make_float2((float)x, (float)y);
            if (bin_hist) atomicAdd(&bin_hist[min(n_tiles - 1, (int)((y >> 6) * tiles_x + (x >> 6)))], 1u);   // = bin_count_kernel
        }
    }
    // the last workgroup turns the tile counts into the tiles' first positions in the LK visiting order (= bin_scan_kernel)
    if (bin_hist && ticket && last_workgroup(ticket, gridDim.x)) {
        __shared__ uint32_t s_scan[256];
        (void)scan_exclusive_256(bin_hist, bin_hist, n_tiles, s_scan);
    }
}

// ------------------------------------------------------------------------------------------------
// K5 for LARGE radii (min_distance > kSuppressMaxTableRadius): the reference's own algorithm (gftt.cc:100-164) -- candidates in
// priority order against a grid of the corners accepted so far -- by ONE wavefront.  The table-driven kernel above reads the
// pi R^2 state bytes around every candidate (and walks them again per spin when more than 12 higher-priority neighbours exist):
// fine at the detector's 5 px, hopeless at 200.  With a large radius almost every candidate is rejected by a corner accepted long
// before it, and a frame has few corners (w h / (pi R^2 / 4) at most): 64 candidates per step, lane l = candidate base + l:
//   * every lane tests its candidate against the accepted corners of the 3 x 3 grid cells around it (cell size = ceil(R): a cell
//     holds at most four corners that keep the distance, kLargeCellCap slots);
//   * the candidates that survive are settled among themselves in lane (= priority) order: the lowest surviving lane is accepted,
//     enters the grid, and knocks out the surviving lanes within the radius; repeat.
// The same predicate as everywhere ((float)(dx^2 + dy^2) < min_distance^2 as a double), so the same corners as the reference's
// loop.  Then the workgroup's 256 lanes scan the per-block counts for the ordered compaction.  ~2 us per 64 candidates.
// `grid`: gw * gh * (1 + 2 * kLargeCellCap) zeroed words.
// ------------------------------------------------------------------------------------------------
constexpr int kLargeCellCap = 8;
constexpr int kLargeCellWords = 1 + 2 * kLargeCellCap;
__global__ __launch_bounds__(256) void suppress_large_radius_kernel(const unsigned long long* __restrict__ keys, uint32_t n_max,
                                                                    const uint32_t* __restrict__ n_dev, int w, uint8_t* cstate, double r2,
                                                                    int cell, int gw, int gh, uint32_t* grid,
                                                                    uint32_t* __restrict__ accepted_per_block, int nblocks,
                                                                    uint32_t* __restrict__ stuck, uint32_t* __restrict__ n_out,
                                                                    uint32_t* __restrict__ overflow, uint32_t max_corners, int hi_prio) {
    helper_priority(hi_prio);
    __shared__ uint32_t s_scan[256];
    const uint32_t n = n_dev ? min(*n_dev, n_max) : n_max;
    for (int b = threadIdx.x; b < nblocks + 1; b += 256) publish(&accepted_per_block[b], 0u);
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        auto gload = [&](size_t at) { return __hip_atomic_load(grid + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        auto gstore = [&](size_t at, uint32_t v) { __hip_atomic_store(grid + at, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        uint32_t block_accepted = 0;
        for (uint32_t base = 0; base < n; base += 64) {
            const uint32_t i = base + lane;
            const bool live = i < n;
            const uint32_t idx = live ? (uint32_t)keys[i] : 0u;
            const int y = (int)(idx / (uint32_t)w), x = (int)(idx - (uint32_t)y * (uint32_t)w);
            const int cx = x / cell, cy = y / cell;
            bool pending = live;
            if (live) {
                for (int gy = max(cy - 1, 0); gy <= min(cy + 1, gh - 1) && pending; gy++)
                    for (int gx = max(cx - 1, 0); gx <= min(cx + 1, gw - 1) && pending; gx++) {
                        const size_t at = (size_t)(gy * gw + gx) * kLargeCellWords;
                        const int cnt = min((int)gload(at), kLargeCellCap);
                        for (int k = 0; k < cnt; k++) {
                            const float dx = (float)(x - (int)gload(at + 1 + 2 * k)), dy = (float)(y - (int)gload(at + 2 + 2 * k));
                            if ((double)(dx * dx + dy * dy) < r2) pending = false;   // gftt.cc:134-141
                        }
                    }
            }
            bool accepted = false;
            unsigned long long pm = __ballot(pending);
            while (pm) {
                const int winner = __ffsll((long long)pm) - 1;   // the highest priority among the survivors
                const int ax = __shfl(x, winner), ay = __shfl(y, winner);
                if (lane == winner) {
                    accepted = true;
                    pending = false;
                    const size_t at = (size_t)(cy * gw + cx) * kLargeCellWords;
                    const uint32_t cnt = gload(at);
                    if (cnt < (uint32_t)kLargeCellCap) {
                        gstore(at + 1 + 2 * cnt, (uint32_t)x);
                        gstore(at + 2 + 2 * cnt, (uint32_t)y);
                    } else {
                        atomicAdd(stuck, 1u);   // cannot happen (four corners fit a cell): reported like the other kernel's tripwire
                    }
                    gstore(at, cnt + 1u);
                } else if (pending) {
                    const float dx = (float)(x - ax), dy = (float)(y - ay);
                    if ((double)(dx * dx + dy * dy) < r2) pending = false;
                }
                pm = __ballot(pending);
            }
            if (live) cs_store(&cstate[idx], accepted ? CS_ACCEPTED : CS_REJECTED);
            block_accepted += (uint32_t)__popcll(__ballot(accepted));
            if (((base >> 6) & (SUP_BLOCK / 64 - 1)) == (SUP_BLOCK / 64 - 1) || base + 64 >= n) {
                if (lane == 0) publish(&accepted_per_block[base / SUP_BLOCK], block_accepted);
                block_accepted = 0;
            }
            __builtin_amdgcn_s_waitcnt(0);   // this step's corners are in the grid before the next step reads it
        }
    }
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const uint32_t total = scan_exclusive_256(accepted_per_block, accepted_per_block, nblocks, s_scan);
    if (threadIdx.x == 0) {
        if (n_dev && overflow && *n_dev > n_max) atomicOr(overflow, 4u);
        *n_out = (max_corners > 0 && total > max_corners) ? max_corners : total;
    }
}
// cell size of the grid: ceil(min_distance), at most the frame's larger side (a radius beyond that: one cell -- and no overflow of
// the conversion for absurd values)
static int suppress_large_cell(int w, int h, double min_distance) {
    const double side = (double)std::max(w, h);
    return std::max(1, (int)std::ceil(std::min(min_distance, side)));
}
int suppress_large_grid_words(int w, int h, double min_distance) {
    const int cell = suppress_large_cell(w, h, min_distance);
    return ((w + cell - 1) / cell) * ((h + cell - 1) / cell) * kLargeCellWords;
}

int suppress_num_blocks(uint32_t n) { return (int)((n + SUP_BLOCK - 1) / SUP_BLOCK); }

// Suppression + ordered compaction: TWO launches.  `tickets` = two zeroed arrays of last_workgroup_words(blocks) words,
// ticket_stride words apart (one per launch, see last_workgroup);
// `n_out` receives the keypoint count, per_block is scratch [suppress_num_blocks(n_max) + 1].  With `bin_hist` (zeroed,
// bin_num_tiles words) the compaction also leaves the tiles' first positions of the LK visiting order there.
void launch_suppress_and_compact(const unsigned long long* keys, uint32_t n_max, const uint32_t* n_dev, int w, int h, const float* eig,
                                 uint8_t* cstate, const int2* offsets, int n_offsets, const int* row_hw, int R, bool suppress,
                                 uint32_t* per_block,
                                 uint32_t* stuck, uint32_t max_corners, float2* xy, uint32_t* n_out, uint32_t* bin_hist,
                                 uint32_t* overflow, uint32_t* tickets, uint32_t ticket_stride, double large_min_distance, uint32_t* large_grid,
                                 hipStream_t s) {
    uint32_t* const accepted_per_block = per_block;
    const int nb = suppress_num_blocks(n_max);
    if (nb == 0) return;
    const AcceptedScan fin{tickets, n_out, overflow, max_corners};
    if (suppress && large_grid) {
        const int cell = suppress_large_cell(w, h, large_min_distance), gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
        (void)hipMemsetAsync(large_grid, 0, (size_t)gw * gh * kLargeCellWords * sizeof(uint32_t), s);
        hipLaunchKernelGGL(suppress_large_radius_kernel, dim3(1), dim3(256), 0, s, keys, n_max, n_dev, w, cstate, large_min_distance * large_min_distance,
                           cell, gw, gh, large_grid, accepted_per_block, nb, stuck, n_out, overflow, max_corners, helper_prio_arg());
    } else if (suppress)
        hipLaunchKernelGGL(suppress_sorted_kernel, dim3(nb), dim3(SUP_BLOCK), 0, s, keys, n_max, n_dev, w, h, eig, cstate, offsets,
                           n_offsets, row_hw, R, accepted_per_block, stuck, fin, helper_prio_arg());
    else
        hipLaunchKernelGGL(accept_all_kernel, dim3(nb), dim3(SUP_BLOCK), 0, s, keys, n_max, n_dev, cstate, accepted_per_block, fin, helper_prio_arg());
    const int tiles_x = (w + 63) >> 6, n_tiles = bin_hist ? bin_num_tiles(w, h) : 0;
    hipLaunchKernelGGL(accepted_scatter_kernel, dim3(nb), dim3(SUP_BLOCK), 0, s, keys, n_max, n_dev, w, cstate, per_block, max_corners, xy,
                       bin_hist, tiles_x, n_tiles, tickets + ticket_stride, helper_prio_arg());
}

// ------------------------------------------------------------------------------------------------
// K4 (fallback)  descending 64-bit radix sort (value desc, linear index desc) with the count on the host.
// ------------------------------------------------------------------------------------------------
hipError_t sort_keys_desc(void* temp, size_t& temp_bytes, unsigned long long* keys_in,
                          unsigned long long* keys_out, uint32_t n, hipStream_t s) {
    return rocprim::radix_sort_keys_desc(temp, temp_bytes, keys_in, keys_out, (size_t)n, 0u, 64u, s);
}

}  // namespace pc
