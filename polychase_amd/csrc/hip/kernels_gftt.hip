// kernels_gftt.hip -- Shi-Tomasi corner response and candidate extraction on gfx950.
//
// Replaces, on the path GoodFeaturesToTrack (reference cpp/feature_detection/gftt.cc:14-192):
//   K2  cv::cornerMinEigenVal(block 3, Sobel 3)              gftt.cc:35
//       + per-grid-cell cv::minMaxLoc                         gftt.cc:61-63
//   K3  per-cell cv::threshold(THRESH_TOZERO)                 gftt.cc:64-65
//       + cv::dilate 3x3 + strict-interior local maxima       gftt.cc:70-86
//   K4  std::sort(greaterThanPtr)                             gftt.cc:7-12, :98   (64-bit radix sort)
// Float order follows oracle/pc_oracle.c exactly (no FMA contraction; fp64 box sums are exact).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "kernels.hpp"

namespace pc {

// ------------------------------------------------------------------------------------------------
// K2  min-eigenvalue map.  Tile 64x16 outputs per 256-lane workgroup.
//   LDS stage 1: gray tile with a 2-px REFLECT_101 halo (68 x 20 bytes, stored as u8).
//   LDS stage 2: covariance products (Dx^2, DxDy, Dy^2) on the tile + 1-px halo (66 x 18 x 3 floats),
//                evaluated at the REFLECTED coordinate for positions outside the image (boxFilter's
//                BORDER_REFLECT_101 applies to the covariance image, not to the gray image).
//   Stage 3: 3x3 box sums in fp64 (exact), min eigenvalue, per-cell max via LDS then global atomicMax.
// ------------------------------------------------------------------------------------------------
constexpr int TW = 64, TH = 16;
constexpr int GW = TW + 4, GH = TH + 4;   // gray tile
constexpr int CW = TW + 2, CH = TH + 2;   // covariance tile

__global__ __launch_bounds__(256) void min_eig_kernel(const uint8_t* __restrict__ img, int pitch, int w, int h,
                                                      float* __restrict__ eig, GfttGrid g,
                                                      uint32_t* __restrict__ cell_max, float f1, float f0) {
    __shared__ uint8_t s_gray[GH][GW];
    __shared__ float s_cxx[CH][CW + 1];
    __shared__ float s_cxy[CH][CW + 1];
    __shared__ float s_cyy[CH][CW + 1];
    __shared__ uint32_t s_max[4];

    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    if (tid < 4) s_max[tid] = 0u;

    // stage 1: gray tile, absolute coords [x0-2, x0+TW+2) x [y0-2, y0+TH+2), reflected into the image
    for (int i = tid; i < GW * GH; i += 256) {
        const int ty = i / GW, tx = i - ty * GW;
        const int ax = reflect101(x0 - 2 + tx, w), ay = reflect101(y0 - 2 + ty, h);
        s_gray[ty][tx] = img[(size_t)ay * pitch + ax];
    }
    __syncthreads();

    // stage 2: Sobel + products at covariance positions [x0-1, x0+TW+1) x [y0-1, y0+TH+1)
    for (int i = tid; i < CW * CH; i += 256) {
        const int cy = i / CW, cx = i - cy * CW;
        // absolute position, reflected into the image, then back to gray-tile coordinates
        int ax = x0 - 1 + cx, ay = y0 - 1 + cy;
        float vxx = 0.f, vxy = 0.f, vyy = 0.f;
        // positions more than one pixel outside the image are never consumed
        if (ax >= -1 && ax <= w && ay >= -1 && ay <= h) {
            ax = reflect101(ax, w);
            ay = reflect101(ay, h);
            const int tx = ax - (x0 - 2), ty = ay - (y0 - 2);
            // neighbours of an in-image position: reflect at the image border
            const int txm = reflect101(ax - 1, w) - (x0 - 2), txp = reflect101(ax + 1, w) - (x0 - 2);
            const int tym = reflect101(ay - 1, h) - (y0 - 2), typ = reflect101(ay + 1, h) - (y0 - 2);
            const float g00 = s_gray[tym][txm], g01 = s_gray[tym][tx], g02 = s_gray[tym][txp];
            const float g10 = s_gray[ty][txm], g11 = s_gray[ty][tx], g12 = s_gray[ty][txp];
            const float g20 = s_gray[typ][txm], g21 = s_gray[typ][tx], g22 = s_gray[typ][txp];
            // Dx: row [-1,0,1] then column (S0 + S2)*f1 + S1*f0
            const float rx0 = g02 - g00, rx1 = g12 - g10, rx2 = g22 - g20;
            const float dx = (rx0 + rx2) * f1 + rx1 * f0;
            // Dy: row ((f1*a + f0*b) + f1*c) then column S2 - S0
            float ry0 = f1 * g00; ry0 += f0 * g01; ry0 += f1 * g02;
            float ry2 = f1 * g20; ry2 += f0 * g21; ry2 += f1 * g22;
            const float dy = ry2 - ry0;
            (void)g11;
            vxx = dx * dx;
            vxy = dx * dy;
            vyy = dy * dy;
        }
        s_cxx[cy][cx] = vxx;
        s_cxy[cy][cx] = vxy;
        s_cyy[cy][cx] = vyy;
    }
    __syncthreads();

    // stage 3: 4 outputs per lane (rows ty, ty+4, ty+8, ty+12 of column tx)
    const int tx = tid & 63, tyb = tid >> 6;
    const int x = x0 + tx;
    const int cell_x0 = x0 / g.cell_w, cell_y0 = y0 / g.cell_h;
    const bool small_cells = (g.cell_w < TW) || (g.cell_h < TH);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int ty = tyb + 4 * k;
        const int y = y0 + ty;
        if (x < w && y < h) {
            double sxx = 0.0, sxy = 0.0, syy = 0.0;
#pragma unroll
            for (int j = 0; j < 3; j++)
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    sxx += (double)s_cxx[ty + j][tx + i];
                    sxy += (double)s_cxy[ty + j][tx + i];
                    syy += (double)s_cyy[ty + j][tx + i];
                }
            const float a = (float)sxx * 0.5f;
            const float b = (float)sxy;
            const float c = (float)syy * 0.5f;
            const float t = a - c;
            const float e = (a + c) - sqrtf(t * t + b * b);
            eig[(size_t)y * w + x] = e;
            const uint32_t key = float_to_ordered(e);
            const int cx = x / g.cell_w, cy = y / g.cell_h;
            if (small_cells) {
                atomicMax(&cell_max[cy * g.cols + cx], key);
            } else {
                atomicMax(&s_max[(cy - cell_y0) * 2 + (cx - cell_x0)], key);
            }
        }
    }
    __syncthreads();
    if (!small_cells && tid < 4) {
        const int cx = cell_x0 + (tid & 1), cy = cell_y0 + (tid >> 1);
        if (cx < g.cols && cy < g.rows && s_max[tid] != 0u) atomicMax(&cell_max[cy * g.cols + cx], s_max[tid]);
    }
}

void launch_min_eig(const Level& l0, float* eig, const GfttGrid& g, uint32_t* cell_max, hipStream_t s) {
    // scale = 1 / (2^(ksize-1) * block_size * 255), folded into the smoothing taps (see oracle)
    const double scale_d = 1.0 / (4.0 * 3.0 * 255.0);
    const float f1 = (float)(1.0 * scale_d), f0 = (float)(2.0 * scale_d);
    dim3 grid((l0.w + TW - 1) / TW, (l0.h + TH - 1) / TH);
    hipLaunchKernelGGL(min_eig_kernel, grid, dim3(256), 0, s, l0.img, l0.pitch, l0.w, l0.h, eig, g, cell_max, f1, f0);
}

// ------------------------------------------------------------------------------------------------
// K3  threshold (per cell of the NEIGHBOUR) + 3x3 dilate + local-max test + wave-ballot compaction.
// One lane per pixel of the strict interior; keys = ordered(value) << 32 | (y*w + x).
// ------------------------------------------------------------------------------------------------
constexpr int NMS_R = 16, NMS_TH = 4 * NMS_R;  // rows per lane, tile height

__global__ __launch_bounds__(256) void nms_compact_kernel(const float* __restrict__ eig, int w, int h, GfttGrid g,
                                                          const uint32_t* __restrict__ cell_max,
                                                          double quality_level,
                                                          unsigned long long* __restrict__ keys, uint32_t cap,
                                                          uint32_t* __restrict__ counter,
                                                          uint32_t* __restrict__ cmap, uint32_t* __restrict__ state) {
    __shared__ float s_thr[kMaxGridCells];
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_base;
    const int ncells = g.rows * g.cols;
    for (int i = threadIdx.x; i < ncells; i += blockDim.x) {
        // cv::threshold on CV_32F compares with (float)(maxVal * quality_level), maxVal a double
        const float mx = ordered_to_float(cell_max[i]);
        s_thr[i] = (float)((double)mx * quality_level);
    }
    __syncthreads();

    // tile 64 x NMS_TH: lane = column, NMS_R rows per lane (wave, wave+4, ...)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane;
    float vals[NMS_R];
    uint32_t flags = 0;
    const bool x_in = (x >= 1 && x < w - 1);
    int cxs[3] = {0, 0, 0};
    if (x_in) {
        cxs[0] = (x - 1) / g.cell_w;
        cxs[1] = x / g.cell_w;
        cxs[2] = (x + 1) / g.cell_w;
    }
#pragma unroll
    for (int k = 0; k < NMS_R; k++) {
        const int y = blockIdx.y * NMS_TH + wave + 4 * k;
        vals[k] = 0.f;
        if (x_in && y >= 1 && y < h - 1) {
            const int cys[3] = {(y - 1) / g.cell_h, y / g.cell_h, (y + 1) / g.cell_h};
            const float c = eig[(size_t)y * w + x];
            const float val = (c > s_thr[cys[1] * g.cols + cxs[1]]) ? c : 0.f;
            if (val != 0.f) {
                float m = val;
#pragma unroll
                for (int j = 0; j < 3; j++)
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        const float e = eig[(size_t)(y + j - 1) * w + (x + i - 1)];
                        const float v = (e > s_thr[cys[j] * g.cols + cxs[i]]) ? e : 0.f;
                        m = (v > m) ? v : m;
                    }
                if (val == m) {
                    flags |= 1u << k;
                    vals[k] = val;
                }
            }
        }
    }
    // dense priority map for the suppression kernel: ordered(value) at candidates, 0 elsewhere;
    // every pixel is written, so the maps need no clearing between frames
    if (cmap && x < w) {
#pragma unroll
        for (int k = 0; k < NMS_R; k++) {
            const int y = blockIdx.y * NMS_TH + wave + 4 * k;
            if (y < h) {
                cmap[(size_t)y * w + x] = (flags & (1u << k)) ? float_to_ordered(vals[k]) : 0u;
                state[(size_t)y * w + x] = 0u;
            }
        }
    }
    // workgroup-aggregated append: one global atomic per 64 x 64 tile
    const uint32_t cnt = (uint32_t)__popc(flags);
    uint32_t incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        s_base = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    if (cnt == 0) return;
    uint32_t pos = s_base + incl - cnt;
    for (int wv = 0; wv < wave; wv++) pos += s_wave[wv];
#pragma unroll
    for (int k = 0; k < NMS_R; k++) {
        if (flags & (1u << k)) {
            const int y = blockIdx.y * NMS_TH + wave + 4 * k;
            if (pos < cap)
                keys[pos] = ((unsigned long long)float_to_ordered(vals[k]) << 32) | (unsigned long long)(uint32_t)(y * w + x);
            pos++;
        }
    }
}

void launch_nms_compact(const float* eig, int w, int h, const GfttGrid& g, const uint32_t* cell_max,
                        double quality_level, unsigned long long* keys, uint32_t cap, uint32_t* counter,
                        uint32_t* cmap, uint32_t* state, hipStream_t s) {
    dim3 grid((w + 63) / 64, (h + NMS_TH - 1) / NMS_TH);
    hipLaunchKernelGGL(nms_compact_kernel, grid, dim3(256), 0, s, eig, w, h, g, cell_max, quality_level, keys, cap,
                       counter, cmap, state);
}

// ------------------------------------------------------------------------------------------------
// K5  exact min-distance suppression on the GPU.
// The reference's greedy loop (gftt.cc:100-164) accepts a candidate iff no ALREADY ACCEPTED candidate
// lies within min_distance; processing order = (value desc, address desc).  Equivalently: a
// candidate is accepted iff every higher-priority candidate within the radius is rejected -- the
// lexicographically-first maximal independent set of the conflict graph.  Each candidate is owned by
// one lane of a fully resident grid and re-evaluates until all its higher-priority neighbours are
// decided; decisions are final and monotone, so asynchronous evaluation reaches exactly the
// sequential result.  Progress: the highest-priority undecided candidate is always decidable, and a
// lane round-robins over its candidates (never spins on one), so no cycle of waits can form.
// States (dense u32 map, agent-scope relaxed atomics: per-XCD L2s are not coherent): 0 undecided,
// 1 accepted, 2 rejected.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t ST_ACCEPTED = 1u, ST_REJECTED = 2u;

// Evaluate one candidate against the current states of its higher-priority neighbours.
// Returns 0 = still blocked, ST_ACCEPTED or ST_REJECTED.
__device__ __forceinline__ uint32_t suppress_eval_scan(uint32_t my_val, uint32_t my_idx, int x, int y, int w, int h,
                                                       const uint32_t* __restrict__ cmap, uint32_t* state,
                                                       const int2* __restrict__ offsets, int n_offsets) {
    bool blocked = false;
    for (int o = 0; o < n_offsets; o++) {
        const int nx = x + offsets[o].x, ny = y + offsets[o].y;
        if (nx < 0 || nx >= w || ny < 0 || ny >= h) continue;
        const uint32_t nidx = (uint32_t)(ny * w + nx);
        const uint32_t nval = cmap[nidx];
        if (nval == 0u) continue;
        if (!(nval > my_val || (nval == my_val && nidx > my_idx))) continue;  // lower priority
        const uint32_t st = __hip_atomic_load(&state[nidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (st == ST_ACCEPTED) return ST_REJECTED;
        if (st == 0u) blocked = true;
    }
    return blocked ? 0u : ST_ACCEPTED;
}

constexpr int SUP_NB = 12;  // higher-priority neighbours cached in registers by the fast path

__global__ __launch_bounds__(256) void suppress_kernel(const unsigned long long* __restrict__ keys,
                                                       const uint32_t* __restrict__ counter, uint32_t cap, int w, int h,
                                                       const uint32_t* __restrict__ cmap, uint32_t* state,
                                                       const int2* __restrict__ offsets, int n_offsets,
                                                       uint32_t* __restrict__ stuck) {
    const uint32_t n = min(*counter, cap);
    const uint32_t T = gridDim.x * blockDim.x;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n) return;
    if (n <= T) {
        // fast path (the usual case): one candidate per lane.  Scan the neighbourhood once, remember
        // the higher-priority candidates, then only poll their states.
        const unsigned long long key = keys[tid];
        const uint32_t my_val = (uint32_t)(key >> 32), my_idx = (uint32_t)key;
        const int y = (int)(my_idx / (uint32_t)w), x = (int)(my_idx - (uint32_t)y * (uint32_t)w);
        uint32_t nb[SUP_NB];
        int n_nb = 0;
        bool overflow = false;
        for (int o = 0; o < n_offsets; o++) {
            const int nx = x + offsets[o].x, ny = y + offsets[o].y;
            if (nx < 0 || nx >= w || ny < 0 || ny >= h) continue;
            const uint32_t nidx = (uint32_t)(ny * w + nx);
            const uint32_t nval = cmap[nidx];
            if (nval == 0u) continue;
            if (!(nval > my_val || (nval == my_val && nidx > my_idx))) continue;
            if (n_nb < SUP_NB) {
#pragma unroll
                for (int k = 0; k < SUP_NB; k++)
                    if (k == n_nb) nb[k] = nidx;
                n_nb++;
            } else {
                overflow = true;
            }
        }
        // NOTE on control flow: a lane must publish its decision INSIDE the loop and keep iterating
        // (idle) until the whole wave is done.  With a per-lane `return` the store would sit in the
        // loop's exit block, which a wave only executes after ALL its lanes left the loop -- a lane
        // waiting for a neighbour owned by the same wave would then never see it (SIMT deadlock).
        bool done = false;
        for (uint32_t spin = 0;; spin++) {
            if (!done) {
                uint32_t decision;
                if (overflow) {
                    decision = suppress_eval_scan(my_val, my_idx, x, y, w, h, cmap, state, offsets, n_offsets);
                } else {
                    bool blocked = false, rejected = false;
#pragma unroll
                    for (int k = 0; k < SUP_NB; k++) {
                        if (k < n_nb) {
                            const uint32_t st = __hip_atomic_load(&state[nb[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            rejected |= (st == ST_ACCEPTED);
                            blocked |= (st == 0u);
                        }
                    }
                    decision = rejected ? ST_REJECTED : (blocked ? 0u : ST_ACCEPTED);
                }
                if (decision != 0u) {
                    __hip_atomic_store(&state[my_idx], decision, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    done = true;
                }
            }
            if (__all(done)) break;           // wave-uniform exit
            if (spin > (1u << 20)) {          // bounded spin: report instead of hanging the GPU
                if (!done) atomicAdd(stuck, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        return;
    }
    // general path: several candidates per lane, visited round-robin (never spin on one candidate)
    const uint32_t mine = (n - tid + T - 1) / T;   // candidates tid, tid+T, ...
    uint32_t remaining = mine;
    for (uint32_t spin = 0; remaining > 0; spin++) {
        for (uint32_t j = 0; j < mine; j++) {
            const unsigned long long key = keys[tid + (size_t)j * T];
            const uint32_t my_val = (uint32_t)(key >> 32), my_idx = (uint32_t)key;
            if (__hip_atomic_load(&state[my_idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) continue;
            const int y = (int)(my_idx / (uint32_t)w), x = (int)(my_idx - (uint32_t)y * (uint32_t)w);
            const uint32_t decision = suppress_eval_scan(my_val, my_idx, x, y, w, h, cmap, state, offsets, n_offsets);
            if (decision != 0u) {
                __hip_atomic_store(&state[my_idx], decision, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                remaining--;
            }
        }
        if (remaining > 0) {
            if (spin > (1u << 20)) {
                atomicAdd(stuck, 1u);
                return;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
}

// accepted candidates -> dense list (order irrelevant: sorted afterwards); grid-stride so that the
// launch does not need the candidate count on the host
__global__ __launch_bounds__(256) void collect_accepted_kernel(const unsigned long long* __restrict__ keys,
                                                               const uint32_t* __restrict__ counter, uint32_t cap,
                                                               const uint32_t* __restrict__ state, int take_all,
                                                               unsigned long long* __restrict__ out,
                                                               uint32_t* __restrict__ out_counter) {
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_base;
    const uint32_t n = min(*counter, cap);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        unsigned long long key = 0ull;
        bool keep = false;
        if (i < n) {
            key = keys[i];
            keep = take_all || state[(uint32_t)key] == ST_ACCEPTED;
        }
        const unsigned long long ballot = __ballot(keep);
        __syncthreads();  // previous iteration's readers of s_wave / s_base are done
        if (lane == 0) s_wave[wave] = (uint32_t)__popcll(ballot);
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
            s_base = total ? atomicAdd(out_counter, total) : 0u;
        }
        __syncthreads();
        if (keep) {
            uint32_t pos = s_base + (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull));
            for (int wv = 0; wv < wave; wv++) pos += s_wave[wv];
            out[pos] = key;
        }
    }
}

__global__ __launch_bounds__(256) void keys_to_xy_kernel(const unsigned long long* __restrict__ keys, int n, int w,
                                                         float2* __restrict__ xy) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t idx = (uint32_t)keys[i];
    const uint32_t y = idx / (uint32_t)w;
    xy[i] = make_float2((float)(idx - y * (uint32_t)w), (float)y);  // Point2f((float)x, (float)y), gftt.cc:157
}

void launch_suppress(const unsigned long long* keys, const uint32_t* counter, uint32_t cap, int w, int h,
                     const uint32_t* cmap, uint32_t* state, const int2* offsets, int n_offsets, uint32_t* stuck,
                     int resident_blocks, hipStream_t s) {
    hipLaunchKernelGGL(suppress_kernel, dim3(resident_blocks), dim3(256), 0, s, keys, counter, cap, w, h, cmap, state,
                       offsets, n_offsets, stuck);
}

void launch_collect_accepted(const unsigned long long* keys, const uint32_t* counter, uint32_t cap,
                             const uint32_t* state, int take_all, unsigned long long* out, uint32_t* out_counter,
                             hipStream_t s) {
    hipLaunchKernelGGL(collect_accepted_kernel, dim3(1024), dim3(256), 0, s, keys, counter, cap, state, take_all, out,
                       out_counter);
}

void launch_keys_to_xy(const unsigned long long* keys, int n, int w, float2* xy, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(keys_to_xy_kernel, dim3((n + 255) / 256), dim3(256), 0, s, keys, n, w, xy);
}

// ------------------------------------------------------------------------------------------------
// K4  descending 64-bit radix sort (value desc, linear index desc).
// ------------------------------------------------------------------------------------------------
hipError_t sort_keys_desc(void* temp, size_t& temp_bytes, unsigned long long* keys_in,
                          unsigned long long* keys_out, uint32_t n, hipStream_t s) {
    return rocprim::radix_sort_keys_desc(temp, temp_bytes, keys_in, keys_out, (size_t)n, 0u, 64u, s);
}

}  // namespace pc
