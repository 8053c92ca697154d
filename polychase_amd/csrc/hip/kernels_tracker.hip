// kernels_tracker.hip -- GPU kernels of the "Track Sequence" path (reference cpp/tracker.cc):
//   K12  closest-hit ray casting of source keypoints onto the mesh   (tracker.cc:64-78; Embree
//        rtcIntersect1 in the reference, cpp/ray_casting.cc:65-121).  One lane per ray walking the
//        LBVH of bvh.hpp with a per-lane stack in LDS; the per-triangle test is the reference's own
//        Moeller-Trumbore (cpp/ray_casting.h:125-179).  The exhaustive sweep over all triangles
//        (staged through LDS) is kept as the validation path of the hierarchy.
//   K11  PnP residual / Jacobian / normal equations / cost          (cpp/pnp/pnp_problem.h:52-99,
//        cpp/pnp/lev_marq.h:231-356), deterministic two-stage reduction.
// fp32 throughout like the reference (Float = float, cpp/eigen_typedefs.h).
#include "bvh.hpp"
#include <algorithm>

#include "pnp_lm.hpp"
#include "kernels.hpp"

namespace pc {

// ------------------------------------------------------------------------------------------------
// K12 ray casting
// ------------------------------------------------------------------------------------------------
constexpr int RC_TILE = 256;

__global__ __launch_bounds__(256) void raycast_sweep_kernel(const float* __restrict__ verts, const uint32_t* __restrict__ tris,
                                                      int n_tris, const uint32_t* __restrict__ mask, int check_mask,
                                                      RayCamera cam, const float2* __restrict__ xy, int n,
                                                      uint8_t* __restrict__ hit, float* __restrict__ pos,
                                                      uint32_t* __restrict__ prim, float* __restrict__ uvt) {
    __shared__ float s_p1[RC_TILE][3], s_e1[RC_TILE][3], s_e2[RC_TILE][3];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    float ox = cam.origin[0], oy = cam.origin[1], oz = cam.origin[2];
    float dx = 0.f, dy = 0.f, dz = 1.f;
    if (live) {
        // CameraIntrinsics::Unproject (cpp/pnp/types.h:95-98) then rotate into object space
        const float2 p = xy[i];
        const float ux = cam.sign * ((p.x - cam.cx) / cam.fx), uy = cam.sign * ((p.y - cam.cy) / cam.fy), uz = cam.sign;
        dx = cam.m[0] * ux + cam.m[1] * uy + cam.m[2] * uz;
        dy = cam.m[3] * ux + cam.m[4] * uy + cam.m[5] * uz;
        dz = cam.m[6] * ux + cam.m[7] * uy + cam.m[8] * uz;
    }
    float best_t = __builtin_inff(), best_u = 0.f, best_v = 0.f;
    int best = -1;
    for (int base = 0; base < n_tris; base += RC_TILE) {
        __syncthreads();
        const int t = base + threadIdx.x;
        if (t < n_tris) {
            const uint32_t a = tris[3 * t], b = tris[3 * t + 1], c = tris[3 * t + 2];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float p1 = verts[3 * a + k];
                s_p1[threadIdx.x][k] = p1;
                s_e1[threadIdx.x][k] = verts[3 * b + k] - p1;
                s_e2[threadIdx.x][k] = verts[3 * c + k] - p1;
            }
        }
        __syncthreads();
        const int cnt = min(RC_TILE, n_tris - base);
        if (live) {
            for (int k = 0; k < cnt; k++) {
                const float e1x = s_e1[k][0], e1y = s_e1[k][1], e1z = s_e1[k][2];
                const float e2x = s_e2[k][0], e2y = s_e2[k][1], e2z = s_e2[k][2];
                // ray_cross_e2 = dir x edge2
                const float cx = dy * e2z - dz * e2y, cy = dz * e2x - dx * e2z, cz = dx * e2y - dy * e2x;
                const float det = e1x * cx + e1y * cy + e1z * cz;
                if (det > -1e-10f && det < 1e-10f) continue;
                const float inv_det = 1.0f / det;
                const float sx = ox - s_p1[k][0], sy = oy - s_p1[k][1], sz = oz - s_p1[k][2];
                const float u = inv_det * (sx * cx + sy * cy + sz * cz);
                if (u < 0.0f || u > 1.0f) continue;
                // s_cross_e1 = s x edge1
                const float qx = sy * e1z - sz * e1y, qy = sz * e1x - sx * e1z, qz = sx * e1y - sy * e1x;
                const float v = inv_det * (dx * qx + dy * qy + dz * qz);
                if (v < 0.0f || u + v > 1.0f) continue;
                const float tt = inv_det * (e2x * qx + e2y * qy + e2z * qz);
                if (tt < 0.0f) continue;
                if (tt < best_t) {  // closest hit; ties keep the lower triangle index
                    best_t = tt;
                    best_u = u;
                    best_v = v;
                    best = base + k;
                }
            }
        }
    }
    if (!live) return;
    bool ok = best >= 0;
    // a masked closest triangle is a miss, not a pass-through (ray_casting.cc:104-106)
    if (ok && check_mask && ((mask[best >> 5] >> (best & 31)) & 1u)) ok = false;
    hit[i] = ok ? 1 : 0;
    if (ok) {
        const uint32_t a = tris[3 * best], b = tris[3 * best + 1], c = tris[3 * best + 2];
        const float w0 = 1.0f - best_u - best_v;
#pragma unroll
        for (int k = 0; k < 3; k++)  // Triangle::Barycentric (geometry.h:17-19)
            pos[3 * i + k] = w0 * verts[3 * a + k] + best_u * verts[3 * b + k] + best_v * verts[3 * c + k];
        prim[i] = (uint32_t)best;
        uvt[3 * i] = best_u;
        uvt[3 * i + 1] = best_v;
        uvt[3 * i + 2] = best_t;
    } else {
        pos[3 * i] = pos[3 * i + 1] = pos[3 * i + 2] = 0.f;
        prim[i] = 0xffffffffu;
        uvt[3 * i] = uvt[3 * i + 1] = uvt[3 * i + 2] = 0.f;
    }
}

void launch_raycast_sweep(const float* verts, const uint32_t* tris, int n_tris, const uint32_t* mask, int check_mask,
                          const RayCamera& cam, const float2* xy, int n, uint8_t* hit, float* pos, uint32_t* prim,
                          float* uvt, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(raycast_sweep_kernel, dim3((n + 255) / 256), dim3(256), 0, s, verts, tris, n_tris, mask, check_mask,
                       cam, xy, n, hit, pos, prim, uvt);
}

constexpr int RC_BLOCK = 128;

__global__ __launch_bounds__(RC_BLOCK) void raycast_bvh_kernel(BvhView B, const uint32_t* __restrict__ mask, int check_mask,
                                                               RayCamera cam, const float2* __restrict__ xy, int n,
                                                               uint8_t* __restrict__ hit, float* __restrict__ pos,
                                                               uint32_t* __restrict__ prim, float* __restrict__ uvt) {
    __shared__ int s_stack[kBvhStack][RC_BLOCK];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // CameraIntrinsics::Unproject (cpp/pnp/types.h:95-98) then rotate into object space
    const float2 p = xy[i];
    const float ux = cam.sign * ((p.x - cam.cx) / cam.fx), uy = cam.sign * ((p.y - cam.cy) / cam.fy), uz = cam.sign;
    const float dx = cam.m[0] * ux + cam.m[1] * uy + cam.m[2] * uz;
    const float dy = cam.m[3] * ux + cam.m[4] * uy + cam.m[5] * uz;
    const float dz = cam.m[6] * ux + cam.m[7] * uy + cam.m[8] * uz;
    float best_t, best_u, best_v;
    const int best = bvh_closest_hit(B, cam.origin[0], cam.origin[1], cam.origin[2], dx, dy, dz, &s_stack[0][threadIdx.x], RC_BLOCK,
                                     &best_t, &best_u, &best_v);
    bool ok = best >= 0;
    // a masked closest triangle is a miss, not a pass-through (ray_casting.cc:104-106)
    if (ok && check_mask && ((mask[best >> 5] >> (best & 31)) & 1u)) ok = false;
    hit[i] = ok ? 1 : 0;
    if (ok) {
        const uint32_t a = B.tris[3 * best], b = B.tris[3 * best + 1], c = B.tris[3 * best + 2];
        const float w0 = 1.0f - best_u - best_v;
#pragma unroll
        for (int k = 0; k < 3; k++)  // Triangle::Barycentric (geometry.h:17-19)
            pos[3 * i + k] = w0 * B.verts[3 * a + k] + best_u * B.verts[3 * b + k] + best_v * B.verts[3 * c + k];
        prim[i] = (uint32_t)best;
        uvt[3 * i] = best_u;
        uvt[3 * i + 1] = best_v;
        uvt[3 * i + 2] = best_t;
    } else {
        pos[3 * i] = pos[3 * i + 1] = pos[3 * i + 2] = 0.f;
        prim[i] = 0xffffffffu;
        uvt[3 * i] = uvt[3 * i + 1] = uvt[3 * i + 2] = 0.f;
    }
}

void launch_raycast(const BvhView& bvh, const uint32_t* mask, int check_mask, const RayCamera& cam, const float2* xy, int n,
                    uint8_t* hit, float* pos, uint32_t* prim, float* uvt, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(raycast_bvh_kernel, dim3((n + RC_BLOCK - 1) / RC_BLOCK), dim3(RC_BLOCK), 0, s, bvh, mask, check_mask, cam, xy,
                       n, hit, pos, prim, uvt);
}

// ------------------------------------------------------------------------------------------------
// K12b correspondences of one source frame (cpp/tracker.cc:52-92), built on the device:
//   pass 1  per match: pixel = keypoints[src_idx], closest hit under the source camera (the code of
//           raycast_bvh_kernel), world = model * hit; flag + world point stay in scratch, the block's hit
//           count goes to block_counts
//   pass 2  one block: exclusive scan of the block counts on top of the set's running size
//   pass 3  per block: order-preserving scatter of the hits behind the block's offset
// Match order is kept, so the arrays equal what the host loop of the reference would have appended.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RC_BLOCK) void corr_cast_kernel(BvhView B, const uint32_t* __restrict__ mask, int check_mask,
                                                             RayCamera cam, CorrModel model, const float2* __restrict__ kps,
                                                             int n_kps, const uint32_t* __restrict__ src_idx, int n,
                                                             uint8_t* __restrict__ flag, float* __restrict__ world,
                                                             int* __restrict__ block_counts, int* __restrict__ bad_index) {
    __shared__ int s_stack[kBvhStack][RC_BLOCK];
    __shared__ int s_count[RC_BLOCK / 64];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool ok = false;
    if (i < n) {
        const uint32_t k = src_idx[i];
        if (k >= (uint32_t)n_kps) {
            atomicExch(bad_index, 1);   // CHECK_LT(idx, keypoints.size()), tracker.cc:61
        } else {
            const float2 p = kps[k];
            const float ux = cam.sign * ((p.x - cam.cx) / cam.fx), uy = cam.sign * ((p.y - cam.cy) / cam.fy), uz = cam.sign;
            const float dx = cam.m[0] * ux + cam.m[1] * uy + cam.m[2] * uz;
            const float dy = cam.m[3] * ux + cam.m[4] * uy + cam.m[5] * uz;
            const float dz = cam.m[6] * ux + cam.m[7] * uy + cam.m[8] * uz;
            float best_t, best_u, best_v;
            const int best = bvh_closest_hit(B, cam.origin[0], cam.origin[1], cam.origin[2], dx, dy, dz,
                                             &s_stack[0][threadIdx.x], RC_BLOCK, &best_t, &best_u, &best_v);
            ok = best >= 0;
            if (ok && check_mask && ((mask[best >> 5] >> (best & 31)) & 1u)) ok = false;
            if (ok) {
                const uint32_t a = B.tris[3 * best], b = B.tris[3 * best + 1], c = B.tris[3 * best + 2];
                const float w0 = 1.0f - best_u - best_v;
                float pos[3];
#pragma unroll
                for (int q = 0; q < 3; q++)  // Triangle::Barycentric (geometry.h:17-19)
                    pos[q] = w0 * B.verts[3 * a + q] + best_u * B.verts[3 * b + q] + best_v * B.verts[3 * c + q];
#pragma unroll
                for (int r = 0; r < 3; r++)   // model_matrix * hit (tracker.cc:80-82), left to right like the host code
                    world[3 * (size_t)i + r] = model.m[4 * r] * pos[0] + model.m[4 * r + 1] * pos[1] + model.m[4 * r + 2] * pos[2] +
                                               model.m[4 * r + 3];
            }
        }
        flag[i] = ok ? 1 : 0;
    }
    const unsigned long long b = __ballot(ok);
    if ((threadIdx.x & 63) == 0) s_count[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        int c = 0;
        for (int w = 0; w < RC_BLOCK / 64; w++) c += s_count[w];
        block_counts[blockIdx.x] = c;
    }
}

// counter[0] = correspondences in the set; block_offsets[b] = where block b's hits go
__global__ __launch_bounds__(256) void corr_scan_kernel(const int* __restrict__ block_counts, int nblocks, int* __restrict__ counter,
                                                        int* __restrict__ block_offsets) {
    __shared__ int s_sum[256];
    const int per = (nblocks + 255) / 256;
    const int lo = threadIdx.x * per, hi = min(lo + per, nblocks);
    int local = 0;
    for (int b = lo; b < hi; b++) local += block_counts[b];
    s_sum[threadIdx.x] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = counter[0];
        for (int t = 0; t < 256; t++) {
            const int v = s_sum[t];
            s_sum[t] = run;
            run += v;
        }
        counter[0] = run;
    }
    __syncthreads();
    int run = s_sum[threadIdx.x];
    for (int b = lo; b < hi; b++) {
        block_offsets[b] = run;
        run += block_counts[b];
    }
}

__global__ __launch_bounds__(RC_BLOCK) void corr_scatter_kernel(const uint8_t* __restrict__ flag, const float* __restrict__ world,
                                                                const float2* __restrict__ tgt, int n,
                                                                const int* __restrict__ block_offsets, float* __restrict__ X,
                                                                float2* __restrict__ x) {
    __shared__ int s_count[RC_BLOCK / 64];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = i < n && flag[i];
    const unsigned long long b = __ballot(ok);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_count[wave] = __popcll(b);
    __syncthreads();
    if (!ok) return;
    int off = block_offsets[blockIdx.x] + __popcll(b & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; w++) off += s_count[w];
    X[3 * (size_t)off] = world[3 * (size_t)i];
    X[3 * (size_t)off + 1] = world[3 * (size_t)i + 1];
    X[3 * (size_t)off + 2] = world[3 * (size_t)i + 2];
    x[off] = tgt[i];
}

int corr_num_blocks(int n) { return (n + RC_BLOCK - 1) / RC_BLOCK; }

void launch_corr_append(const BvhView& bvh, const uint32_t* mask, int check_mask, const RayCamera& cam, const CorrModel& model,
                        const float2* kps, int n_kps, const uint32_t* src_idx, const float2* tgt, int n, uint8_t* flag,
                        float* world, int* block_counts, int* block_offsets, int* counter, int* bad_index, float* X, float2* x,
                        hipStream_t s) {
    if (n <= 0) return;
    const int nb = corr_num_blocks(n);
    hipLaunchKernelGGL(corr_cast_kernel, dim3(nb), dim3(RC_BLOCK), 0, s, bvh, mask, check_mask, cam, model, kps, n_kps, src_idx, n,
                       flag, world, block_counts, bad_index);
    hipLaunchKernelGGL(corr_scan_kernel, dim3(1), dim3(256), 0, s, block_counts, nb, counter, block_offsets);
    hipLaunchKernelGGL(corr_scatter_kernel, dim3(nb), dim3(RC_BLOCK), 0, s, flag, world, tgt, n, block_offsets, X, x);
}

// ------------------------------------------------------------------------------------------------
// K11 PnP
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float loss_weight(int type, float scale, float r2) {
    // cpp/pnp/robust_loss.h:47-104
    if (type == 0) return 1.0f;
    if (type == 1) {  // Huber
        if (r2 <= scale * scale) return 1.0f;
        return scale / sqrtf(r2);
    }
    const float inv_sq = 1.0f / (scale * scale);  // Cauchy
    return fmaxf(1.17549435e-38f, 1.0f / (1.0f + r2 * inv_sq));
}
__device__ __forceinline__ float loss_value(int type, float scale, float r2) {
    if (type == 0) return r2;
    if (type == 1) {
        if (r2 <= scale * scale) return r2;
        const float r = sqrtf(r2);
        return scale * (2.0f * r - scale);
    }
    const float sq = scale * scale;
    return sq * log1pf(r2 * (1.0f / sq));
}

constexpr int PNP_ACC = 56;  // 45 (JtJ lower) + 9 (Jtr) + 1 (valid count) + 1 (cost, summed exactly like pnp_cost_kernel)

// block-wide sum of NV values per thread (256 threads); result valid in thread 0
template <int NV>
__device__ __forceinline__ void block_reduce(float (&v)[NV], float (*s_part)[NV]) {
#pragma unroll
    for (int k = 0; k < NV; k++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v[k] += __shfl_xor(v[k], d);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < NV; k++) s_part[wave][k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < NV; k++) v[k] = (s_part[0][k] + s_part[1][k]) + (s_part[2][k] + s_part[3][k]);
}

// one correspondence's terms of the 56 sums (cpp/pnp/pnp_problem.h:63-99, lev_marq.h:231-297): Z world point, (ox, oy) its
// observation, `weight` != 0
__device__ __forceinline__ void pnp_accumulate(float Zx, float Zy, float Zz, float ox, float oy, float weight, const PnPParams& p,
                                               float (&acc)[PNP_ACC]) {
    // RtZ = R Z + t  (Pose::ApplyWithJac, cpp/pose.h:60-78)
    const float ax = p.R[0] * Zx + p.R[1] * Zy + p.R[2] * Zz + p.t[0];
    const float ay = p.R[3] * Zx + p.R[4] * Zy + p.R[5] * Zz + p.t[1];
    const float az = p.R[6] * Zx + p.R[7] * Zy + p.R[8] * Zz + p.t[2];
    // ProjectWithJac (cpp/pnp/types.h:69-93)
    const float zx = p.fx * ax / az + p.cx, zy = p.fy * ay / az + p.cy;
    const float rx = zx - ox, ry = zy - oy;
    const float d00 = p.fx / az, d02 = -p.fx * ax / (az * az);
    const float d11 = p.fy / az, d12 = -p.fy * ay / (az * az);
    // dRtZ_dR = R * Skew(-Z):  Skew(-Z) = [[0, Zz, -Zy], [-Zz, 0, Zx], [Zy, -Zx, 0]]
    float M[9];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const float r0 = p.R[3 * r], r1 = p.R[3 * r + 1], r2 = p.R[3 * r + 2];
        M[3 * r] = -r1 * Zz + r2 * Zy;
        M[3 * r + 1] = r0 * Zz - r2 * Zx;
        M[3 * r + 2] = -r0 * Zy + r1 * Zx;
    }
    float J0[9], J1[9];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        J0[c] = d00 * M[c] + d02 * M[6 + c];
        J1[c] = d11 * M[3 + c] + d12 * M[6 + c];
    }
    J0[3] = d00; J0[4] = 0.f; J0[5] = d02;
    J1[3] = 0.f; J1[4] = d11; J1[5] = d12;
    J0[6] = p.optimize_focal ? p.aspect_ratio * ax / az : 0.f;
    J1[6] = p.optimize_focal ? ay / az : 0.f;
    J0[7] = p.optimize_pp ? 1.f : 0.f; J0[8] = 0.f;
    J1[7] = 0.f; J1[8] = p.optimize_pp ? 1.f : 0.f;
    const float r2n = rx * rx + ry * ry;
    const float tw = weight * loss_weight(p.loss_type, p.loss_scale, r2n);
    int o = 0;
#pragma unroll
    for (int a = 0; a < 9; a++)
#pragma unroll
        for (int b = 0; b <= a; b++) acc[o++] += tw * (J0[a] * J0[b] + J1[a] * J1[b]);
#pragma unroll
    for (int a = 0; a < 9; a++) acc[45 + a] += J0[a] * (tw * rx) + J1[a] * (tw * ry);
    acc[54] += 1.0f;
    // the cost of these parameters, term for term what pnp_cost_kernel adds (same per-thread order, same
    // reduction tree): the LM loop gets the candidate's cost and its normal equations from ONE sweep
    const bool behind = p.convention_opencv ? (az < 0.0f) : (az > 0.0f);
    acc[55] += weight * loss_value(p.loss_type, p.loss_scale, behind ? __builtin_inff() : r2n);
}

__device__ __forceinline__ void pnp_normal_eq_body(const float* __restrict__ X, const float* __restrict__ x,
                                                   const float* __restrict__ w, int n, const PnPParams& p,
                                                   float* __restrict__ partials) {
    __shared__ float s_part[4][PNP_ACC];
    float acc[PNP_ACC];
#pragma unroll
    for (int k = 0; k < PNP_ACC; k++) acc[k] = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float weight = w ? w[i] : 1.0f;
        if (weight == 0.0f) continue;
        pnp_accumulate(X[3 * i], X[3 * i + 1], X[3 * i + 2], x[2 * i], x[2 * i + 1], weight, p, acc);
    }
    block_reduce<PNP_ACC>(acc, s_part);
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < PNP_ACC; k++) partials[(size_t)k * gridDim.x + blockIdx.x] = acc[k];   // value-major: the second stage reads rows
}

__global__ __launch_bounds__(256) void pnp_normal_eq_kernel(const float* __restrict__ X, const float* __restrict__ x,
                                                            const float* __restrict__ w, int n, PnPParams p,
                                                            float* __restrict__ partials) {
    pnp_normal_eq_body(X, x, w, n, p, partials);
}
// the same sweep for the device-resident solver: parameters from its state, nothing to do once it has finished
__global__ __launch_bounds__(256) void pnp_normal_eq_lm_kernel(const float* __restrict__ X, const float* __restrict__ x,
                                                               const float* __restrict__ w, int n,
                                                               const LmState* __restrict__ st, float* __restrict__ partials) {
    if (st->done) return;
    const PnPParams p = st->sweep;
    pnp_normal_eq_body(X, x, w, n, p, partials);
}

// cost (lev_marq.h:316-356) and inlier count (solvers.cc:31-47) in one pass: acc = {cost, valid, inliers, pad}
__device__ __forceinline__ void pnp_cost_body(const float* __restrict__ X, const float* __restrict__ x,
                                              const float* __restrict__ w, int n, const PnPParams& p,
                                              float max_inlier_err_sq, float* __restrict__ partials) {
    __shared__ float s_part[4][4];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float weight = w ? w[i] : 1.0f;
        const float Zx = X[3 * i], Zy = X[3 * i + 1], Zz = X[3 * i + 2];
        const float ax = p.R[0] * Zx + p.R[1] * Zy + p.R[2] * Zz + p.t[0];
        const float ay = p.R[3] * Zx + p.R[4] * Zy + p.R[5] * Zz + p.t[1];
        const float az = p.R[6] * Zx + p.R[7] * Zy + p.R[8] * Zz + p.t[2];
        // PnPProblem::Evaluate (pnp_problem.h:52-61): behind the camera -> FLT_MAX residual
        const bool behind = p.convention_opencv ? (az < 0.0f) : (az > 0.0f);
        float r2n;
        if (behind) {
            r2n = __builtin_inff();  // FLT_MAX^2 + FLT_MAX^2 overflows to +inf in fp32
        } else {
            const float rx = p.fx * ax / az + p.cx - x[2 * i], ry = p.fy * ay / az + p.cy - x[2 * i + 1];
            r2n = rx * rx + ry * ry;
        }
        if (r2n < max_inlier_err_sq) acc[2] += 1.0f;
        if (weight == 0.0f) continue;
        acc[0] += weight * loss_value(p.loss_type, p.loss_scale, r2n);
        acc[1] += 1.0f;
    }
    block_reduce<4>(acc, s_part);
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < 4; k++) partials[(size_t)k * gridDim.x + blockIdx.x] = acc[k];
}

__global__ __launch_bounds__(256) void pnp_cost_kernel(const float* __restrict__ X, const float* __restrict__ x,
                                                       const float* __restrict__ w, int n, PnPParams p,
                                                       float max_inlier_err_sq, float* __restrict__ partials) {
    pnp_cost_body(X, x, w, n, p, max_inlier_err_sq, partials);
}
// inlier pass of the device-resident solver (on the accepted parameters; meaningful once the solver has finished)
__global__ __launch_bounds__(256) void pnp_cost_lm_kernel(const float* __restrict__ X, const float* __restrict__ x,
                                                          const float* __restrict__ w, int n, const LmState* __restrict__ st,
                                                          float* __restrict__ partials) {
    if (!st->done) return;
    const PnPParams p = st->sweep;
    pnp_cost_body(X, x, w, n, p, st->cfg.max_inlier_err_sq, partials);
}
// second reduction stage of a sweep + the decision step between two sweeps (pnp_lm.hpp): the 56 sums are formed like
// in pnp_finalize_kernel (one wavefront per value, fixed order), then one lane runs the solver's 9x9 algebra on a copy
// of the state in LDS
__global__ __launch_bounds__(1024) void pnp_lm_reduce_consume_kernel(const float* __restrict__ partials, int nblocks,
                                                                      LmState* __restrict__ st) {
    __shared__ float s_out[PNP_ACC];
    __shared__ LmState s_state;
    if (st->done) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = wave; k < PNP_ACC; k += 16) {
        float s = 0.f;
        for (int b = lane; b < nblocks; b += 64) s += partials[(size_t)k * nblocks + b];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
        if (lane == 0) s_out[k] = s;
    }
    constexpr int kWords = (int)(sizeof(LmState) / sizeof(uint32_t));
    static_assert(sizeof(LmState) % sizeof(uint32_t) == 0, "LmState is copied word by word");
    const uint32_t* src = reinterpret_cast<const uint32_t*>(st);
    uint32_t* cpy = reinterpret_cast<uint32_t*>(&s_state);
    for (int i = threadIdx.x; i < kWords; i += blockDim.x) cpy[i] = src[i];
    __syncthreads();
    if (threadIdx.x == 0) lm_consume(s_state, s_out);
    __syncthreads();
    uint32_t* dst = reinterpret_cast<uint32_t*>(st);
    for (int i = threadIdx.x; i < kWords; i += blockDim.x) dst[i] = cpy[i];
}

// second stage: fixed-order sum of the per-block partials (deterministic run to run).  One wavefront per value:
// lane l adds partials l, l+64, ... in order, then the 64 lane sums go through a fixed shuffle tree.  (A single
// lane walking all <= 512 partials of a value, one dependent load after the other, took longer than the sweep itself.)
__global__ __launch_bounds__(1024) void pnp_finalize_kernel(const float* __restrict__ partials, int nblocks, int nv,
                                                            float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = wave; k < nv; k += 16) {
        float s = 0.f;
        for (int b = lane; b < nblocks; b += 64) s += partials[(size_t)k * nblocks + b];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
        if (lane == 0) out[k] = s;
    }
}

// `iterations` rounds of [sweep of the state's parameters, reduce, decide]; then the inlier pass (a no-op unless
// the solver has finished) -- all enqueued without waiting
void launch_pnp_lm_rounds(const float* X, const float* x, const float* w, int n, LmState* st, int iterations, float* partials,
                          float* partials4, float* out4, hipStream_t s) {
    const int nb = pnp_num_blocks(n);
    for (int k = 0; k < iterations; k++) {
        hipLaunchKernelGGL(pnp_normal_eq_lm_kernel, dim3(nb), dim3(256), 0, s, X, x, w, n, st, partials);
        hipLaunchKernelGGL(pnp_lm_reduce_consume_kernel, dim3(1), dim3(1024), 0, s, partials, nb, st);
    }
    hipLaunchKernelGGL(pnp_cost_lm_kernel, dim3(nb), dim3(256), 0, s, X, x, w, n, st, partials4);
    hipLaunchKernelGGL(pnp_finalize_kernel, dim3(1), dim3(1024), 0, s, partials4, nb, 4, out4);
}

int pnp_num_blocks(int n) {
    const int b = (n + 255) / 256;
    return b < 1 ? 1 : (b > 512 ? 512 : b);
}

void launch_pnp_normal_eq(const float* X, const float* x, const float* w, int n, const PnPParams& p, float* partials,
                          float* out56, hipStream_t s) {
    const int nb = pnp_num_blocks(n);
    hipLaunchKernelGGL(pnp_normal_eq_kernel, dim3(nb), dim3(256), 0, s, X, x, w, n, p, partials);
    hipLaunchKernelGGL(pnp_finalize_kernel, dim3(1), dim3(1024), 0, s, partials, nb, PNP_ACC, out56);
}

void launch_pnp_cost(const float* X, const float* x, const float* w, int n, const PnPParams& p, float max_err_sq,
                     float* partials, float* out4, hipStream_t s) {
    const int nb = pnp_num_blocks(n);
    hipLaunchKernelGGL(pnp_cost_kernel, dim3(nb), dim3(256), 0, s, X, x, w, n, p, max_err_sq, partials);
    hipLaunchKernelGGL(pnp_finalize_kernel, dim3(1), dim3(1024), 0, s, partials, nb, 4, out4);
}

// ------------------------------------------------------------------------------------------------
// SolveFrame in two launches (reference cpp/tracker.cc:36-131 = correspondences of every source + SolvePnPIterative)
//
//   track_cast_kernel   ONE launch for all source frames of the frame being solved: lane = one match of one source;
//                       gather the source keypoint, closest hit under THAT source's camera, model transform -- the code of
//                       corr_cast_kernel -- and pts[i] = (world point, 1) for a hit, (0, 0, 0, 0) for a miss.  No
//                       compaction: the solver skips invalid entries, their count is one of its sums.
//   track_lm_kernel     the WHOLE Levenberg-Marquardt loop (lev_marq.h:132-228) + the inlier pass (solvers.cc:31-47) as one
//                       persistent launch: every workgroup keeps its share of the correspondences, sweeps them with the
//                       parameters of the round, publishes its 56 partial sums and arrives at a grid barrier; workgroup 0
//                       -- which holds the solver's state in LDS for the whole launch -- adds the partials in a fixed
//                       order, takes the decision (lm_consume: one lane, 9x9 algebra in registers), publishes the next
//                       parameters and releases the others.  Rounds stop at `done`; the result goes straight to pinned
//                       host memory.  Round 4 enqueued 13 x (sweep, reduce + decide) + 2 launches per frame, each round
//                       15 + 15 us of a mostly idle GPU (profiles/r05_head_c5_timeline.json).
//
// Cross-workgroup data follows common.hpp's rules for this chip (the XCDs' L2s are not coherent with one another): what
// another workgroup will read is written with agent-scope atomic stores and read either with agent-scope atomic loads (the
// round word, the parameters) or with plain loads behind ONE acquire fence in the reading workgroup (the partials).
// Co-residency: at most 256 workgroups of 256 lanes (one per CU of this chip), 12 KB of LDS each: every workgroup of the launch
// is resident as long as 288 of every SIMD's 512 registers are free (the kernel holds 281 per lane: a lane's correspondences
// live in registers), and nothing else on the GPU waits for this kernel.  The spin
// loops sleep between polls and give up after kTrackSpinLimit ticks (status 2) instead of hanging: the host then solves the
// frame -- and the rest of the run -- with the per-source building blocks (csrc/host/track_sequence.cc).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RC_BLOCK) void track_cast_kernel(TrackCastArgs a) {
    __shared__ int s_stack[kBvhStack][RC_BLOCK];
    // the block's source: the last one whose first block is <= blockIdx (blocks of a source are consecutive)
    int sel = 0;
    for (int k = 1; k < a.n_sources; k++)
        if ((int)blockIdx.x >= a.src[k].block_begin) sel = k;
    sel = __builtin_amdgcn_readfirstlane(sel);
    const TrackSource& S = a.src[sel];
    const int local = ((int)blockIdx.x - S.block_begin) * RC_BLOCK + (int)threadIdx.x;
    if (local >= S.n_matches) return;
    const size_t i = (size_t)S.begin + (size_t)local;
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t k = S.idx[local];
    a.obs[i] = S.tgt[local];
    if (k >= (uint32_t)S.n_kps) {
        atomicExch(a.bad_index, 1);   // CHECK_LT(idx, keypoints.size()), tracker.cc:61
    } else {
        RayCamera cam = S.cam;
        if (S.cam_dev) cam = *S.cam_dev;   // uniform: the camera the LM launch in front of this one left on the device
        const float2 p = S.kps[k];
        const float ux = cam.sign * ((p.x - cam.cx) / cam.fx), uy = cam.sign * ((p.y - cam.cy) / cam.fy), uz = cam.sign;
        const float dx = cam.m[0] * ux + cam.m[1] * uy + cam.m[2] * uz;
        const float dy = cam.m[3] * ux + cam.m[4] * uy + cam.m[5] * uz;
        const float dz = cam.m[6] * ux + cam.m[7] * uy + cam.m[8] * uz;
        float best_t, best_u, best_v;
        const int best = bvh_closest_hit(a.bvh, cam.origin[0], cam.origin[1], cam.origin[2], dx, dy, dz, &s_stack[0][threadIdx.x],
                                         RC_BLOCK, &best_t, &best_u, &best_v);
        bool ok = best >= 0;
        if (ok && a.check_mask && ((a.mask[best >> 5] >> (best & 31)) & 1u)) ok = false;   // ray_casting.cc:104-106
        if (ok) {
            const BvhView& B = a.bvh;
            const uint32_t va = B.tris[3 * best], vb = B.tris[3 * best + 1], vc = B.tris[3 * best + 2];
            const float w0 = 1.0f - best_u - best_v;
            float pos[3];
#pragma unroll
            for (int q = 0; q < 3; q++)  // Triangle::Barycentric (geometry.h:17-19)
                pos[q] = w0 * B.verts[3 * va + q] + best_u * B.verts[3 * vb + q] + best_v * B.verts[3 * vc + q];
            const float* m = a.model.m;   // model_matrix * hit (tracker.cc:80-82), left to right like the host code
            out.x = m[0] * pos[0] + m[1] * pos[1] + m[2] * pos[2] + m[3];
            out.y = m[4] * pos[0] + m[5] * pos[1] + m[6] * pos[2] + m[7];
            out.z = m[8] * pos[0] + m[9] * pos[1] + m[10] * pos[2] + m[11];
            out.w = 1.0f;
        }
    }
    a.pts[i] = out;
}

int track_cast_blocks(int n_matches) { return (n_matches + RC_BLOCK - 1) / RC_BLOCK; }
void launch_track_cast(const TrackCastArgs& a, int total_blocks, hipStream_t s) {
    if (total_blocks <= 0) return;
    hipLaunchKernelGGL(track_cast_kernel, dim3(total_blocks), dim3(RC_BLOCK), 0, s, a);
}

namespace {

constexpr int kParamWords = (int)(sizeof(PnPParams) / sizeof(uint32_t));
static_assert(sizeof(PnPParams) % sizeof(uint32_t) == 0, "PnPParams travels word by word");
// Words of TrackLmArgs::sync (all zero before a launch, left zero by it):
//   [kSyncAbort]                a lane's wait ran out: everybody leaves
//   [kSyncParams .. + 2 * (kParamWords + 1))   the decision of a round as 64-bit words (tag << 32 | value), tag = the round it
//                               is FOR: the parameters to evaluate and `done`.  A tagged word is its own flag: a reader needs
//                               ONE round trip to learn that the round has begun and what to evaluate (a round word polled
//                               first and the parameters fetched after cost two).
//   [kSyncFlags + b]            workgroup b has published its partial sums of round r: r + 1
constexpr int kSyncAbort = 0, kSyncParams = 2, kSyncFlags = 64;
static_assert(kSyncParams + 2 * (kParamWords + 1) <= kSyncFlags && kSyncFlags + 256 <= kTrackSyncWords, "sync layout");
// wall_clock64 ticks (100 MHz) a wait may last: 100 ms.  A launch whose workgroups do not all become resident (a renderer on the
// same GPU -- the addon's normal life -- holds the registers) gives the CUs back after that time; the host retries once and then
// solves with the per-source building blocks (track_sequence.cc).  Round 5 spun for 5 s.
constexpr long long kTrackSpinLimit = 10000000ll;

__device__ __forceinline__ uint32_t peek(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long peek64(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void publish64(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Polls until `ready()` holds in every lane that called; false when the launch was aborted or the limit ran out.  Called by
// whole wavefronts (all lanes take part in the loop; lanes with nothing to wait for pass ready() = true).
template <typename Ready>
__device__ __forceinline__ bool spin_until(Ready ready, uint32_t* sync) {
    const long long t0 = wall_clock64();
    for (uint32_t it = 0;; it++) {
        if (__all(ready())) return true;
        __builtin_amdgcn_s_sleep(2);
        if ((it & 63u) == 63u) {
            if (peek(sync + kSyncAbort)) return false;
            if (wall_clock64() - t0 > kTrackSpinLimit) {
                publish(sync + kSyncAbort, 1u);
                return false;
            }
        }
    }
}

// The wavefront's sum of v, in lane 63: six additions whose second operand comes over DPP (neighbour, other pair, other quad, other
// half of the row of 16, then the rows' totals handed on by row_bcast 15 / 31) -- a v_mov_b32_dpp + v_add_f32 each; the
// ds_bpermute + wait + add of a shuffle butterfly cost the sweep of 56 values a microsecond per round.  Lanes a pattern leaves
// out receive -0.0f, the identity of the addition.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_term(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp((int)0x80000000u, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_in_lane63(float v) {
    v += dpp_term<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v += dpp_term<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v += dpp_term<0x141, 0xf>(v);   // row_half_mirror
    v += dpp_term<0x140, 0xf>(v);   // row_mirror: every lane of a row holds the row's sum
    v += dpp_term<0x142, 0xa>(v);   // row_bcast 15 into rows 1 and 3
    v += dpp_term<0x143, 0xc>(v);   // row_bcast 31 into rows 2 and 3
    return v;
}
// wave sums -> LDS -> lanes k < NV publish the workgroup's sum of value k (value-major like pnp_normal_eq_body)
template <int NV>
__device__ __forceinline__ void block_reduce_publish(float (&v)[NV], float (*s_part)[NV], float* __restrict__ partials) {
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = wave_sum_in_lane63(v[k]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();   // the previous round's readers of s_part are done
    if (lane == 63)
#pragma unroll
        for (int k = 0; k < NV; k++) s_part[wave][k] = v[k];
    __syncthreads();
    if ((int)threadIdx.x < NV) {
        const int k = threadIdx.x;
        const float sum = (s_part[0][k] + s_part[1][k]) + (s_part[2][k] + s_part[3][k]);
        __hip_atomic_store(partials + (size_t)k * gridDim.x + blockIdx.x, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// this workgroup's partial sums of round `arrival - 1` are published: each lane waits until the memory system has
// acknowledged its own stores, the barrier collects the workgroup, one lane raises the workgroup's flag
__device__ __forceinline__ void grid_arrive(uint32_t* sync, uint32_t arrival) {
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) publish(sync + kSyncFlags + blockIdx.x, arrival);
}
// workgroup 0: until every workgroup's flag says `arrival` (lane t watches the flags t, t + 256); the result in s_ok
__device__ __forceinline__ void wait_for_all(uint32_t* sync, uint32_t arrival, int* s_ok) {
    const int G = gridDim.x, t = threadIdx.x;   // G <= 256 = the lanes of this workgroup: lane t watches the flag of workgroup t
    bool a = t >= G;
    const bool ok = spin_until(
        [&]() {
            if (!a) a = peek(sync + kSyncFlags + t) == arrival;
            return a;
        },
        sync);
    if (t == 0) *s_ok = 1;
    __syncthreads();
    if (!ok) *s_ok = 0;   // any wavefront that gave up
    __syncthreads();
}

// Sum of `nblocks` (<= 256) partials of each of NV values by one 256-lane workgroup: wave w takes the values w, w + 4, ...;
// lane l adds the partials l, l + 64, ... (4 of them), then a fixed shuffle tree.  ALL loads of a lane are issued before the
// first is consumed: behind the acquire fence every one of them misses the L2, and a loop that waits for each load in turn
// (round 4's second stage, and the first version of this kernel: 14 values x 8 dependent misses per lane) took longer than
// the sweep it follows.
template <int NV>
__device__ __forceinline__ void reduce_partials(const float* __restrict__ partials, int nblocks, float* s_out) {
    constexpr int PER_WAVE = (NV + 3) / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float v[PER_WAVE][4];
#pragma unroll
    for (int q = 0; q < PER_WAVE; q++) {
        const int k = wave + 4 * q;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int b = lane + 64 * j;
            v[q][j] = (k < NV && b < nblocks) ? partials[(size_t)k * nblocks + b] : 0.f;
        }
    }
#pragma unroll
    for (int q = 0; q < PER_WAVE; q++) {
        const int k = wave + 4 * q;
        const float s = wave_sum_in_lane63((v[q][0] + v[q][1]) + (v[q][2] + v[q][3]));
        if (lane == 63 && k < NV) s_out[k] = s;
    }
}

}  // namespace

// SourceCamera (csrc/host/track_sequence.cc) on the device, for TrackChainSlot::ray: Pose::Rt4x4 (types.h; the rotation by
// Quatf::ToRotationMatrix = lm_make_params' R), MatMul4(view, model) in float, Inverse4 in double with partial pivoting
// (linalg.h) -- the host's operations in the host's order, so the bits are the host's.  Static indices only (the pivot row is
// picked with selects): nothing goes to scratch memory.  One lane, once per launch.
__device__ __forceinline__ void track_source_camera(const LmCamera& c, const float* model, RayCamera* out) {
    const float tx = 2 * c.qx, ty = 2 * c.qy, tz = 2 * c.qz;
    const float twx = tx * c.qw, twy = ty * c.qw, twz = tz * c.qw;
    const float txx = tx * c.qx, txy = ty * c.qx, txz = tz * c.qx;
    const float tyy = ty * c.qy, tyz = tz * c.qy, tzz = tz * c.qz;
    const float view[16] = {1 - (tyy + tzz), txy - twz, txz + twy, c.t[0], txy + twz, 1 - (txx + tzz), tyz - twx, c.t[1],
                            txz - twy,       tyz + twx, 1 - (txx + tyy), c.t[2], 0.f, 0.f, 0.f, 1.f};
    double a[4][8];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) s += view[4 * i + k] * model[4 * k + j];
            a[i][j] = (double)s;
            a[i][4 + j] = (i == j) ? 1.0 : 0.0;
        }
#pragma unroll
    for (int col = 0; col < 4; col++) {
        int piv = col;
        double best = fabs(a[col][col]);
#pragma unroll
        for (int r = col + 1; r < 4; r++) {
            const double v = fabs(a[r][col]);
            if (v > best) {
                best = v;
                piv = r;
            }
        }
        // (a singular view * model: the host throws when it makes this camera; here the launch that reads the slot runs on
        // garbage for at most max_rounds rounds and the host reports the error when it gets to the frame)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            double pr = a[col][j];
#pragma unroll
            for (int r = col + 1; r < 4; r++) pr = (piv == r) ? a[r][j] : pr;
#pragma unroll
            for (int r = col + 1; r < 4; r++) a[r][j] = (piv == r) ? a[col][j] : a[r][j];
            a[col][j] = pr;
        }
        const double inv = 1.0 / a[col][col];
#pragma unroll
        for (int j = 0; j < 8; j++) a[col][j] *= inv;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (r == col) continue;
            const double f = a[r][col];
            if (f != 0.0) {
#pragma unroll
                for (int j = 0; j < 8; j++) a[r][j] -= f * a[col][j];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int cc = 0; cc < 3; cc++) out->m[3 * r + cc] = (float)a[r][4 + cc];
        out->origin[r] = (float)a[r][7];
    }
    out->fx = c.fx;
    out->fy = c.fy;
    out->cx = c.cx;
    out->cy = c.cy;
    out->sign = c.convention_opencv ? 1.0f : -1.0f;
}

// the decision of one round on top of lm_consume: the correspondence count is only known once the first sweep has
// counted the valid entries (PnPProblem's n: pnp_problem.h:34-35, solvers.cc:54-55, tracker.cc:95-97)
__device__ __forceinline__ void track_consume(LmState& s, const float* out56, int round, int max_rounds) {
    if (s.phase == 0) {
        s.n_valid = (int)out56[54];
        if (s.n_valid < 3) {   // "Not enough features" (tracker.cc:95-97): nothing is solved
            s.status = 1;
            lm_finish(s);
            return;
        }
        if (s.n_valid == 3 && (s.cfg.optimize_focal || s.cfg.optimize_pp)) {
            // intrinsics are only optimised with more than 3 points: evaluate the initial parameters again without
            // their columns (the round stays phase 0)
            s.cfg.optimize_focal = 0;
            s.cfg.optimize_pp = 0;
            lm_make_params(s.cam, s.cfg, &s.sweep);
            return;
        }
    }
    lm_consume(s, out56);
    if (!s.done && round + 1 >= max_rounds) {
        s.status = 3;
        lm_finish(s);
    }
}

// track_consume by the workgroup's first wavefront (lm_consume_wave, pnp_lm.hpp): the same decision, the same bits
__device__ __forceinline__ void track_consume_wave(LmState& s, const float* out56, int round, int max_rounds, int lane) {
    lm_wave_sync();
    if (s.phase == 0) {
        const int n_valid = (int)out56[54];
        const bool drop_intrinsics = n_valid == 3 && (s.cfg.optimize_focal || s.cfg.optimize_pp);
        lm_wave_sync();
        if (lane == 0) {
            s.n_valid = n_valid;
            if (n_valid < 3) {   // "Not enough features" (tracker.cc:95-97): nothing is solved
                s.status = 1;
                lm_finish(s);
            } else if (drop_intrinsics) {
                s.cfg.optimize_focal = 0;
                s.cfg.optimize_pp = 0;
                lm_make_params(s.cam, s.cfg, &s.sweep);
            }
        }
        lm_wave_sync();
        if (n_valid < 3 || drop_intrinsics) return;
    }
    lm_consume_wave(s, out56, lane);
    lm_wave_sync();
    if (!s.done && round + 1 >= max_rounds) {
        lm_wave_sync();
        if (lane == 0) {
            s.status = 3;
            lm_finish(s);
        }
        lm_wave_sync();
    }
}

__global__ __launch_bounds__(256) void track_lm_kernel(TrackLmArgs a) {
    __shared__ float s_part[4][PNP_ACC];
    __shared__ float s_out[PNP_ACC];
    __shared__ LmState s_state;                  // workgroup 0: the solver's state for the whole launch
    __shared__ uint32_t s_pw[kParamWords + 1];
    __shared__ int s_ok;
    const int tid = threadIdx.x, G = gridDim.x;
    const size_t T = (size_t)G * 256, gtid = (size_t)blockIdx.x * 256 + tid;
    PnPParams p;
    // round 0: the initial parameters, from the launch arguments -- or from what the launch in front of this one left on the device
    // (uniform loads; that kernel has ended: its stores are visible)
    LmCamera cam0 = a.cam;
    if (a.chain_in) cam0 = a.chain_in->cam;
    lm_make_params(cam0, a.cfg, &p);
    {
        uint32_t* pw = reinterpret_cast<uint32_t*>(&p);
#pragma unroll
        for (int k = 0; k < kParamWords; k++) pw[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)pw[k]);   // uniform: scalar registers
    }
    // This lane's first kTrackKeep correspondences stay in registers for the whole launch (150 k matches over 65 536 lanes: all
    // of them): the ~12 sweeps of a frame then start with arithmetic instead of a trip to the L2.  The order in which a lane
    // adds its terms is unchanged.
    constexpr int kTrackKeep = 3;
    float4 keepP[kTrackKeep];
    float2 keepO[kTrackKeep];
#pragma unroll
    for (int k = 0; k < kTrackKeep; k++) {
        const size_t i = gtid + (size_t)k * T;
        keepP[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        keepO[k] = make_float2(0.f, 0.f);
        if (i < (size_t)a.n) {
            keepP[k] = a.pts[i];
            keepO[k] = a.obs[i];
        }
    }
    uint32_t round = 0;
    bool aborted = false;
    // where the launch's time goes, measured by workgroup 0 (100 MHz ticks, TrackLmOut::ticks): its own sweep + publish,
    // waiting for the other workgroups, adding the partials, the decision, publishing it, fetching the parameters
    long long tk[kTrackTickPhases] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long t_begin = wall_clock64();
    long long t_mark = t_begin;
    auto tick = [&](int phase) {
        if (blockIdx.x == 0 && tid == 0) {
            const long long now = wall_clock64();
            tk[phase] += now - t_mark;
            t_mark = now;
        }
    };
    for (;;) {
        // ---- the sweep of this workgroup's correspondences (pnp_normal_eq_body's per-lane order) ----
        float acc[PNP_ACC];
#pragma unroll
        for (int k = 0; k < PNP_ACC; k++) acc[k] = 0.f;
#pragma unroll
        for (int k = 0; k < kTrackKeep; k++)
            if (keepP[k].w != 0.0f) pnp_accumulate(keepP[k].x, keepP[k].y, keepP[k].z, keepO[k].x, keepO[k].y, 1.0f, p, acc);
        for (size_t i = gtid + (size_t)kTrackKeep * T; i < (size_t)a.n; i += T) {
            const float4 P = a.pts[i];
            if (P.w == 0.0f) continue;
            const float2 o = a.obs[i];
            pnp_accumulate(P.x, P.y, P.z, o.x, o.y, 1.0f, p, acc);
        }
        block_reduce_publish<PNP_ACC>(acc, s_part, a.partials);
        grid_arrive(a.sync, round + 1u);
        tick(0);
        unsigned long long* pub = reinterpret_cast<unsigned long long*>(a.sync + kSyncParams);
        if (blockIdx.x == 0) {
            wait_for_all(a.sync, round + 1u, &s_ok);
            if (!s_ok) { aborted = true; break; }
            tick(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            reduce_partials<PNP_ACC>(a.partials, G, s_out);
            __syncthreads();
            tick(2);
            if (tid < 64) {   // the workgroup's first wavefront
                if (round == 0 && tid == 0) {
                    LmState& h = s_state;
                    uint32_t* z = reinterpret_cast<uint32_t*>(&h);
                    for (int i = 0; i < (int)(sizeof(LmState) / sizeof(uint32_t)); i++) z[i] = 0u;
                    h.cfg = a.cfg;
                    h.cam = cam0;
                    h.cam_new = cam0;
                    h.lambda = a.cfg.initial_lambda;
                    h.v = 2.0f;
                    h.grad_norm = -1.0f;
                    h.step_norm = -1.0f;
                    h.rebuild = 1;
                    lm_make_params(h.cam, h.cfg, &h.sweep);
                }
                // the decision: the 9x9 algebra dealt out over the wavefront's lanes (same operations, same order per result:
                // the serial code's bits), or -- serial_decision, the cross-check -- on one lane
                if (!a.serial_decision) track_consume_wave(s_state, s_out, (int)round, a.max_rounds, tid);
                else if (tid == 0) track_consume(s_state, s_out, (int)round, a.max_rounds);
            }
            __syncthreads();
            tick(3);
            // the decision, tagged with the round it is for
            const uint32_t* sw = reinterpret_cast<const uint32_t*>(&s_state.sweep);
            const unsigned long long tag = (unsigned long long)(round + 1u) << 32;
            if (tid < kParamWords) publish64(pub + tid, tag | sw[tid]);
            if (tid == kParamWords) publish64(pub + kParamWords, tag | (uint32_t)s_state.done);
            tick(4);
        }
        // ---- everyone: the parameters of the next round (or the accepted ones once done), as soon as they are there ----
        if (tid < 64) {
            unsigned long long v = 0;
            const bool mine = tid <= kParamWords;
            const bool ok = spin_until(
                [&]() {
                    if (mine) v = peek64(pub + tid);
                    return !mine || (uint32_t)(v >> 32) == round + 1u;
                },
                a.sync);
            if (mine) s_pw[tid] = (uint32_t)v;
            if (tid == 0) s_ok = ok ? 1 : 0;
        }
        __syncthreads();
        if (!s_ok) { aborted = true; break; }
        uint32_t* pw = reinterpret_cast<uint32_t*>(&p);
#pragma unroll
        for (int k = 0; k < kParamWords; k++) pw[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_pw[k]);
        const int done = __builtin_amdgcn_readfirstlane((int)s_pw[kParamWords]);
        round++;
        tick(5);
        if (done) break;
    }
    auto hand_over = [&]() { __hip_atomic_store(&a.out->seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); };
    if (aborted) {
        if (blockIdx.x == 0 && tid == 0) {
            a.out->status = 2;
            hand_over();
        }
        return;
    }
    // ---- inlier pass on the accepted parameters (solvers.cc:31-47; pnp_cost_body's terms) ----
    {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        auto inlier_term = [&](const float4& P, const float2& o) {
            const float ax = p.R[0] * P.x + p.R[1] * P.y + p.R[2] * P.z + p.t[0];
            const float ay = p.R[3] * P.x + p.R[4] * P.y + p.R[5] * P.z + p.t[1];
            const float az = p.R[6] * P.x + p.R[7] * P.y + p.R[8] * P.z + p.t[2];
            const bool behind = p.convention_opencv ? (az < 0.0f) : (az > 0.0f);
            float r2n;
            if (behind) {
                r2n = __builtin_inff();
            } else {
                const float rx = p.fx * ax / az + p.cx - o.x, ry = p.fy * ay / az + p.cy - o.y;
                r2n = rx * rx + ry * ry;
            }
            if (r2n < a.cfg.max_inlier_err_sq) acc[2] += 1.0f;
            acc[0] += loss_value(p.loss_type, p.loss_scale, r2n);
            acc[1] += 1.0f;
        };
#pragma unroll
        for (int k = 0; k < kTrackKeep; k++)
            if (keepP[k].w != 0.0f) inlier_term(keepP[k], keepO[k]);
        for (size_t i = gtid + (size_t)kTrackKeep * T; i < (size_t)a.n; i += T) {
            const float4 P = a.pts[i];
            if (P.w == 0.0f) continue;
            inlier_term(P, a.obs[i]);
        }
        block_reduce_publish<4>(acc, reinterpret_cast<float(*)[4]>(&s_part[0][0]), a.partials);
        grid_arrive(a.sync, round + 1u);
    }
    if (blockIdx.x != 0) return;
    wait_for_all(a.sync, round + 1u, &s_ok);
    if (!s_ok) {
        if (tid == 0) {
            a.out->status = 2;
            hand_over();
        }
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    reduce_partials<4>(a.partials, G, s_out);
    __syncthreads();
    // every workgroup has arrived for the last time and reads nothing more: the words go back to zero for the next launch
    for (int i = tid; i < kTrackSyncWords; i += 256) publish(a.sync + i, 0u);
    if (tid == 0) {
        const LmState& h = s_state;
        TrackLmOut* o = a.out;
        o->cam = h.cam;
        o->iterations = h.iterations;
        o->invalid_steps = h.invalid_steps;
        o->initial_cost = h.initial_cost;
        o->cost = h.cost;
        o->lambda = h.lambda;
        o->step_norm = h.step_norm;
        o->grad_norm = h.grad_norm;
        o->n_valid = h.n_valid;
        o->inliers = a.cfg.max_inlier_err_sq > 0.0f ? (int)s_out[2] : 0;
        o->rounds = (int)round;
        tick(6);
        tk[7] = wall_clock64() - t_begin;
        for (int k = 0; k < kTrackTickPhases; k++) o->ticks[k] = (uint32_t)tk[k];
        o->begin_tick = (unsigned long long)t_begin;
        o->end_tick = (unsigned long long)wall_clock64();
        o->status = h.status;
        o->bad_index = *a.bad_index;
        hand_over();
        if (a.chain_out) {   // after the hand-over: the host does not wait for this, the next launch on the stream does
            a.chain_out->cam = h.cam;
            track_source_camera(h.cam, a.model, &a.chain_out->ray);
        }
    }
}

// At most ONE workgroup per CU: the launch needs every workgroup resident at once, and with one per CU that holds as long as 272
// of every SIMD's 512 registers are free -- two per CU (512 workgroups, 239 VGPRs each) left no room for anything else on the GPU.
// Measured on C5 (150 k matches per frame): the sweep gets a third loop trip, the partial sums and flags halve.  The cap is the
// device's own CU count (a partitioned GPU -- CPX / NPS modes -- or a smaller part has fewer than this chip's 256, and a
// grid that cannot be resident spins until the time limit).
static int track_lm_max_blocks() {
    static int cap[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (cap[dev] == 0) {
        int cus = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, track_lm_kernel, 256, 0) != hipSuccess) per_cu = 1;
        cap[dev] = per_cu < 1 ? 1 : std::min(256, cus);   // (kSyncFlags holds 256 flags)
    }
    return cap[dev];
}
int track_lm_blocks(int n) {
    const int b = (n + 255) / 256, cap = track_lm_max_blocks();
    return b < 1 ? 1 : (b > cap ? cap : b);
}
void launch_track_lm(const TrackLmArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(track_lm_kernel, dim3(track_lm_blocks(a.n)), dim3(256), 0, s, a);
}

// The damped 9x9 solve of the device-resident LM on its own (pc_debug_llt9: the reference's float32 known-answer test,
// cpp/examples/levmarq_ill_conditioned_float32_issue.cpp, is run on THIS copy of the factorisation too)
__global__ void llt9_debug_kernel(const float* __restrict__ a81, const float* __restrict__ b9, float* __restrict__ l81,
                                  float* __restrict__ x9, int* __restrict__ ok) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float a[81], x[9];
    for (int i = 0; i < 81; i++) a[i] = a81[i];
    const bool good = lm_cholesky9(a);
    for (int i = 0; i < 9; i++) x[i] = 0.f;
    if (good) lm_cholesky9_solve(a, b9, x);
    for (int i = 0; i < 81; i++) l81[i] = a[i];
    for (int i = 0; i < 9; i++) x9[i] = x[i];
    *ok = good ? 1 : 0;
}
void launch_llt9_debug(const float* a81, const float* b9, float* l81, float* x9, int* ok, hipStream_t s) {
    hipLaunchKernelGGL(llt9_debug_kernel, dim3(1), dim3(64), 0, s, a81, b9, l81, x9, ok);
}

}  // namespace pc
