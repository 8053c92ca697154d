// kernels_lk.hip -- pyramidal Lucas-Kanade on gfx950 (K8-K10) + status compaction.
//
// Replaces cv::calcOpticalFlowPyrLK as called at reference cpp/opticalflow.cc:119-125 and the
// status==1 filter of cpp/opticalflow.cc:130-147.  Arithmetic follows oracle/pc_oracle.c (which
// restates OpenCV's LKTrackerInvoker): 14-bit fixed-point bilinear weights, int16 patches,
// structure tensor / mismatch vector accumulated EXACTLY in integers, one rounding to fp32, 2x2
// solve in fp32 without FMA contraction.
//
// Mapping: one 16-lane DPP row per (keypoint, target) pair -> 4 pairs per wavefront, 16 per
// 256-lane workgroup.  A row owns the WIN x WIN window (pixel p = lane + 16k); window sums are
// all-reduced inside the row with row_ror DPP adds (no LDS, no cross-row traffic), so every lane of
// the row holds the same A, b, delta and the convergence branches are row-uniform.  Consecutive rows
// are the targets of one keypoint, so the I-side gathers of a wave hit the same cache lines.
#include "kernels.hpp"

namespace pc {

constexpr int W_BITS = 14;

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
// all-reduce (sum) across the 16 lanes of a DPP row: row_ror 8, 4, 2, 1
__device__ __forceinline__ int row_allreduce_add(int v) {
    v += dpp_i32<0x128>(v);
    v += dpp_i32<0x124>(v);
    v += dpp_i32<0x122>(v);
    v += dpp_i32<0x121>(v);
    return v;
}

// exact float of (hi * 2^16 + lo): both parts fit an int32; fp64 holds the integer exactly, the
// fp64 -> fp32 conversion rounds once (== (float)(int64) of the oracle).
__device__ __forceinline__ float exact_sum_to_float(int hi, int lo) {
    const double d = (double)hi * 65536.0 + (double)lo;
    return (float)d;
}

struct Weights {
    int w00, w01, w10, w11;
};
__device__ __forceinline__ Weights bilinear_weights(float a, float b) {
    Weights w;
    w.w00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
    w.w01 = __float2int_rn(a * (1.f - b) * (float)(1 << W_BITS));
    w.w10 = __float2int_rn((1.f - a) * b * (float)(1 << W_BITS));
    w.w11 = (1 << W_BITS) - w.w00 - w.w01 - w.w10;
    return w;
}

__device__ __forceinline__ int interp_u8(const uint8_t* __restrict__ p, int pitch, const Weights& w) {
    return PC_DESCALE((int)p[0] * w.w00 + (int)p[1] * w.w01 + (int)p[pitch] * w.w10 + (int)p[pitch + 1] * w.w11,
                      W_BITS - 5);
}

template <int WIN>
__global__ __launch_bounds__(256) void lk_kernel(const LKParams p) {
    constexpr int NPX = WIN * WIN;
    constexpr int K = (NPX + 15) / 16;
    const int gid = (int)((blockIdx.x * 256u + threadIdx.x) >> 4);
    const int l16 = threadIdx.x & 15;
    if (gid >= p.n * p.n_targets) return;  // whole rows exit together
    const int feat = gid / p.n_targets;
    const int tgt = gid - feat * p.n_targets;

    // window offsets owned by this lane
    int off_x[K], off_y[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int q = l16 + 16 * k;
        off_y[k] = q / WIN;
        off_x[k] = q - off_y[k] * WIN;
    }

    const float2 pt = p.pts[feat];
    const float half_win = (float)(WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    float nx = 0.f, ny = 0.f;
    bool status = true;
    float err = 0.f;

    for (int level = p.max_level; level >= 0; --level) {
        const Level L = p.src[level];
        const uint8_t* __restrict__ J = p.tgt[tgt][level];
        const int pitch = L.pitch;
        const float lscale = 1.f / (float)(1 << level);
        float px = pt.x * lscale, py = pt.y * lscale;
        float qx, qy;
        if (level == p.max_level) {
            qx = px;
            qy = py;
        } else {
            qx = nx * 2.f;
            qy = ny * 2.f;
        }
        nx = qx;
        ny = qy;

        px -= half_win;
        py -= half_win;
        const int ipx = (int)floorf(px), ipy = (int)floorf(py);
        if (ipx < -WIN || ipx >= L.w || ipy < -WIN || ipy >= L.h) {
            if (level == 0) {
                status = false;
                err = 0.f;
            }
            continue;
        }
        Weights wI = bilinear_weights(px - (float)ipx, py - (float)ipy);

        // ---- I side: patch, derivative patch, structure tensor ----
        int Ival[K];
        int Dxy[K];  // (int16 ix) | (int16 iy << 16)
        int sA11 = 0, sA12 = 0, sA22 = 0;
        {
            const uint8_t* __restrict__ Ibase = L.img + (ptrdiff_t)ipy * pitch + ipx;
            const int32_t* __restrict__ Dbase = L.der + (ptrdiff_t)ipy * pitch + ipx;
#pragma unroll
            for (int k = 0; k < K; k++) {
                Ival[k] = 0;
                Dxy[k] = 0;
                if (l16 + 16 * k < NPX) {
                    const int o = off_y[k] * pitch + off_x[k];
                    Ival[k] = interp_u8(Ibase + o, pitch, wI);
                    const int32_t d00 = Dbase[o], d01 = Dbase[o + 1], d10 = Dbase[o + pitch], d11 = Dbase[o + pitch + 1];
                    const int ix = PC_DESCALE((int)(int16_t)(d00 & 0xffff) * wI.w00 + (int)(int16_t)(d01 & 0xffff) * wI.w01 +
                                                  (int)(int16_t)(d10 & 0xffff) * wI.w10 + (int)(int16_t)(d11 & 0xffff) * wI.w11,
                                              W_BITS);
                    const int iy = PC_DESCALE((d00 >> 16) * wI.w00 + (d01 >> 16) * wI.w01 + (d10 >> 16) * wI.w10 +
                                                  (d11 >> 16) * wI.w11,
                                              W_BITS);
                    Dxy[k] = (int)((uint32_t)(ix & 0xffff) | ((uint32_t)iy << 16));
                    sA11 += ix * ix;
                    sA12 += ix * iy;
                    sA22 += iy * iy;
                }
            }
        }
        // |ix|,|iy| <= 4080: per-lane partials fit int32 (K * 2^24), the row totals may not for
        // WIN > 11, so they are reduced as exact (hi, lo) 16-bit halves like the b sums below
        const float A11 = exact_sum_to_float(row_allreduce_add(sA11 >> 16), row_allreduce_add(sA11 & 0xffff)) * FLT_SCALE;
        const float A12 = exact_sum_to_float(row_allreduce_add(sA12 >> 16), row_allreduce_add(sA12 & 0xffff)) * FLT_SCALE;
        const float A22 = exact_sum_to_float(row_allreduce_add(sA22 >> 16), row_allreduce_add(sA22 & 0xffff)) * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float tdiff = A11 - A22;
        const float min_eig = (A22 + A11 - sqrtf(tdiff * tdiff + 4.f * A12 * A12)) / (float)(2 * WIN * WIN);
        if (min_eig < p.min_eig_thr || D < 1.1920928955078125e-07f /* FLT_EPSILON */) {
            if (level == 0) status = false;
            continue;
        }
        D = 1.f / D;

        // ---- iterations ----
        qx -= half_win;
        qy -= half_win;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < p.max_iters; j++) {
            const int iqx = (int)floorf(qx), iqy = (int)floorf(qy);
            if (iqx < -WIN || iqx >= L.w || iqy < -WIN || iqy >= L.h) {
                if (level == 0) status = false;
                break;
            }
            const Weights wJ = bilinear_weights(qx - (float)iqx, qy - (float)iqy);
            const uint8_t* __restrict__ Jbase = J + (ptrdiff_t)iqy * pitch + iqx;
            int sb1 = 0, sb2 = 0;  // per-lane partials: <= 7 * 8160 * 4080 < 2^31
#pragma unroll
            for (int k = 0; k < K; k++) {
                if (l16 + 16 * k < NPX) {
                    const int diff = interp_u8(Jbase + off_y[k] * pitch + off_x[k], pitch, wJ) - Ival[k];
                    sb1 += diff * (int)(int16_t)(Dxy[k] & 0xffff);
                    sb2 += diff * (Dxy[k] >> 16);
                }
            }
            // exact 64-bit row sums via (hi, lo) 16-bit split
            const int b1lo = row_allreduce_add(sb1 & 0xffff), b1hi = row_allreduce_add(sb1 >> 16);
            const int b2lo = row_allreduce_add(sb2 & 0xffff), b2hi = row_allreduce_add(sb2 >> 16);
            const float b1 = exact_sum_to_float(b1hi, b1lo) * FLT_SCALE;
            const float b2 = exact_sum_to_float(b2hi, b2lo) * FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * D;
            const float dy = (A12 * b1 - A11 * b2) * D;
            qx += dx;
            qy += dy;
            nx = qx + half_win;
            ny = qy + half_win;
            if ((double)dx * (double)dx + (double)dy * (double)dy <= p.eps_sq) break;
            if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                nx -= dx * 0.5f;
                ny -= dy * 0.5f;
                break;
            }
            pdx = dx;
            pdy = dy;
        }

        // ---- L1 patch error at level 0 ----
        if (status && level == 0) {
            const float ex = nx - half_win, ey = ny - half_win;
            const int iex = (int)floorf(ex), iey = (int)floorf(ey);
            if (iex < -WIN || iex >= L.w || iey < -WIN || iey >= L.h) {
                status = false;
                continue;
            }
            const Weights wE = bilinear_weights(ex - (float)iex, ey - (float)iey);
            const uint8_t* __restrict__ Jbase = J + (ptrdiff_t)iey * pitch + iex;
            int se = 0;
#pragma unroll
            for (int k = 0; k < K; k++) {
                if (l16 + 16 * k < NPX) {
                    const int diff = interp_u8(Jbase + off_y[k] * pitch + off_x[k], pitch, wE) - Ival[k];
                    se += diff < 0 ? -diff : diff;
                }
            }
            se = row_allreduce_add(se);  // <= 256 * 8160 < 2^24: exact in fp32 too
            err = ((float)se * 1.f) / (float)(32 * WIN * WIN);
        }
    }

    if (l16 == 0) {
        const size_t o = (size_t)tgt * p.n + feat;
        p.out_xy[o] = make_float2(nx, ny);
        p.out_status[o] = status ? 1 : 0;
        p.out_err[o] = status ? err : 0.f;
    }
}

template <int WIN>
static void launch_lk_t(const LKParams& p, hipStream_t s) {
    const long long rows = (long long)p.n * p.n_targets;
    const unsigned blocks = (unsigned)((rows + 15) / 16);
    if (blocks == 0) return;
    hipLaunchKernelGGL(lk_kernel<WIN>, dim3(blocks), dim3(256), 0, s, p);
}

bool launch_lk(const LKParams& p, int win, hipStream_t s) {
    switch (win) {
#define PC_LK_CASE(W) case W: launch_lk_t<W>(p, s); return true;
        PC_LK_CASE(3) PC_LK_CASE(4) PC_LK_CASE(5) PC_LK_CASE(6) PC_LK_CASE(7) PC_LK_CASE(8) PC_LK_CASE(9)
        PC_LK_CASE(10) PC_LK_CASE(11) PC_LK_CASE(12) PC_LK_CASE(13) PC_LK_CASE(14) PC_LK_CASE(15) PC_LK_CASE(16)
#undef PC_LK_CASE
        default: return false;
    }
}

// ------------------------------------------------------------------------------------------------
// Ordered compaction of status == 1 rows (opticalflow.cc:130-147): count per 1024-keypoint block,
// exclusive scan of the block counts (one small workgroup), scatter.
// ------------------------------------------------------------------------------------------------
constexpr int CB = 1024;
int compact_num_blocks(int n) { return (n + CB - 1) / CB; }

__global__ __launch_bounds__(1024) void compact_count_kernel(const uint8_t* __restrict__ status, int n, int nblocks,
                                                             uint32_t* __restrict__ block_counts) {
    const int t = blockIdx.y;
    const int i = blockIdx.x * CB + threadIdx.x;
    const bool keep = (i < n) && (status[(size_t)t * n + i] == 1);
    const int c = __syncthreads_count(keep);
    if (threadIdx.x == 0) block_counts[(size_t)t * nblocks + blockIdx.x] = (uint32_t)c;
}

// one workgroup: turns block_counts into exclusive offsets (global, target-major) + row_offset[]
__global__ __launch_bounds__(256) void compact_scan_kernel(uint32_t* __restrict__ block_counts, int nblocks,
                                                           int n_targets, long long* __restrict__ row_offset) {
    __shared__ long long s_part[256];
    const int total = nblocks * n_targets;
    const int per = (total + 255) / 256;
    const int b = threadIdx.x * per, e = min(b + per, total);
    long long sum = 0;
    for (int i = b; i < e; i++) sum += block_counts[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long run = 0;
        for (int i = 0; i < 256; i++) {
            const long long v = s_part[i];
            s_part[i] = run;
            run += v;
        }
    }
    __syncthreads();
    long long run = s_part[threadIdx.x];
    for (int i = b; i < e; i++) {
        const uint32_t c = block_counts[i];
        if (i % nblocks == 0) row_offset[i / nblocks] = run;
        block_counts[i] = (uint32_t)run;  // total rows < 2^32 (n_targets * n)
        run += c;
    }
    if (e == total && b < e) row_offset[n_targets] = run;
    if (total == 0 && threadIdx.x == 0)
        for (int t = 0; t <= n_targets; t++) row_offset[t] = 0;
}

__global__ __launch_bounds__(1024) void compact_scatter_kernel(const float2* __restrict__ xy,
                                                               const uint8_t* __restrict__ status,
                                                               const float* __restrict__ err, int n, int nblocks,
                                                               const uint32_t* __restrict__ block_offsets,
                                                               uint32_t* __restrict__ out_idx,
                                                               float2* __restrict__ out_xy, float* __restrict__ out_err) {
    __shared__ uint32_t s_wave[16];
    const int t = blockIdx.y;
    const int i = blockIdx.x * CB + threadIdx.x;
    const size_t src = (size_t)t * n + i;
    const bool keep = (i < n) && (status[src] == 1);
    const unsigned long long ballot = __ballot(keep);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_wave[wave] = (uint32_t)__popcll(ballot);
    __syncthreads();
    uint32_t base = block_offsets[(size_t)t * nblocks + blockIdx.x];
    for (int wv = 0; wv < wave; wv++) base += s_wave[wv];
    if (keep) {
        const uint32_t pos = base + (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull));
        out_idx[pos] = (uint32_t)i;
        out_xy[pos] = xy[src];
        out_err[pos] = err[src];
    }
}

void launch_compact(const float2* xy, const uint8_t* status, const float* err, int n, int n_targets,
                    uint32_t* block_counts, long long* row_offset, uint32_t* out_idx, float2* out_xy,
                    float* out_err, hipStream_t s) {
    const int nblocks = compact_num_blocks(n);
    if (nblocks > 0)
        hipLaunchKernelGGL(compact_count_kernel, dim3(nblocks, n_targets), dim3(CB), 0, s, status, n, nblocks, block_counts);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(256), 0, s, block_counts, nblocks, n_targets, row_offset);
    if (nblocks > 0)
        hipLaunchKernelGGL(compact_scatter_kernel, dim3(nblocks, n_targets), dim3(CB), 0, s, xy, status, err, n, nblocks,
                           block_counts, out_idx, out_xy, out_err);
}

}  // namespace pc
