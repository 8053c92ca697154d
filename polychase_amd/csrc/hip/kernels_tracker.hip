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
        const float Zx = X[3 * i], Zy = X[3 * i + 1], Zz = X[3 * i + 2];
        // RtZ = R Z + t  (Pose::ApplyWithJac, cpp/pose.h:60-78)
        const float ax = p.R[0] * Zx + p.R[1] * Zy + p.R[2] * Zz + p.t[0];
        const float ay = p.R[3] * Zx + p.R[4] * Zy + p.R[5] * Zz + p.t[1];
        const float az = p.R[6] * Zx + p.R[7] * Zy + p.R[8] * Zz + p.t[2];
        // ProjectWithJac (cpp/pnp/types.h:69-93)
        const float zx = p.fx * ax / az + p.cx, zy = p.fy * ay / az + p.cy;
        const float rx = zx - x[2 * i], ry = zy - x[2 * i + 1];
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
