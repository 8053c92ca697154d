// kernels_refiner.hip -- GPU kernels of "Refine Sequence" (reference cpp/refiner.cc:199-647 with
// cpp/pnp/lev_marq.h:653-824): two-camera reprojection residuals through the mesh.
//   cost:      RefinementProblemBase::Evaluate (refiner.cc:274-361) + LevMarqSparseSolver::TotalCost
//              (lev_marq.h:773-824): per residual, ray of the source keypoint -> cached triangle or
//              closest-hit ray cast -> target camera; per-edge sums, normalised by the edge's valid
//              residual count, weighted by the edge weight.
//   normal eq: EvaluateWithJacobian (refiner.cc:363-506) + BuildNormalEquations (lev_marq.h:653-771):
//              per edge a (2B x 2B) J^T W J block and a 2B vector, B = 6 or 9 parameters per camera.
// One workgroup per edge; the reference scatters edges with relaxed float atomics (not reproducible
// run to run), here every edge block is reduced in a fixed order and assembled on the host.
#include "kernels.hpp"

namespace pc {

__device__ __forceinline__ float3 mul3(const float* m, float3 v) {  // row-major 3x3
    return make_float3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z,
                       m[6] * v.x + m[7] * v.y + m[8] * v.z);
}
__device__ __forceinline__ float3 mul3t(const float* m, float3 v) {  // transpose
    return make_float3(m[0] * v.x + m[3] * v.y + m[6] * v.z, m[1] * v.x + m[4] * v.y + m[7] * v.z,
                       m[2] * v.x + m[5] * v.y + m[8] * v.z);
}
__device__ __forceinline__ float3 cross3(float3 a, float3 b) {
    return make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float dot3(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float3 sub3(float3 a, float3 b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float3 add3(float3 a, float3 b) { return make_float3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ float3 scale3(float3 a, float s) { return make_float3(a.x * s, a.y * s, a.z * s); }
// affine 4x4 (row-major) applied to a point, with the homogeneous divide of hnormalized()
__device__ __forceinline__ float3 xform_point(const float* m, float3 p) {
    const float w = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    return make_float3((m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3]) / w, (m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7]) / w,
                       (m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]) / w);
}
__device__ __forceinline__ float3 xform_dir(const float* m, float3 d) {
    return make_float3(m[0] * d.x + m[1] * d.y + m[2] * d.z, m[4] * d.x + m[5] * d.y + m[6] * d.z,
                       m[8] * d.x + m[9] * d.y + m[10] * d.z);
}

__device__ __forceinline__ float3 load_vertex(const float* __restrict__ verts, uint32_t i) {
    return make_float3(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
}

// Moeller-Trumbore of the reference (cpp/ray_casting.h:125-179); returns t or -1
__device__ __forceinline__ float ray_triangle(float3 o, float3 d, float3 p1, float3 p2, float3 p3, float3* hit) {
    const float3 e1 = sub3(p2, p1), e2 = sub3(p3, p1);
    const float3 c = cross3(d, e2);
    const float det = dot3(e1, c);
    if (det > -1e-10f && det < 1e-10f) return -1.f;
    const float inv = 1.0f / det;
    const float3 s = sub3(o, p1);
    const float u = inv * dot3(s, c);
    if (u < 0.0f || u > 1.0f) return -1.f;
    const float3 q = cross3(s, e1);
    const float v = inv * dot3(d, q);
    if (v < 0.0f || u + v > 1.0f) return -1.f;
    const float t = inv * dot3(e2, q);
    if (t < 0.0f) return -1.f;
    *hit = add3(o, scale3(d, t));
    return t;
}

__device__ __forceinline__ float refine_loss_value(int type, float scale, float r2) {
    if (type == 0) return r2;
    if (type == 1) {
        if (r2 <= scale * scale) return r2;
        return scale * (2.0f * sqrtf(r2) - scale);
    }
    const float sq = scale * scale;
    return sq * log1pf(r2 * (1.0f / sq));
}
__device__ __forceinline__ float refine_loss_weight(int type, float scale, float r2) {
    if (type == 0) return 1.0f;
    if (type == 1) return (r2 <= scale * scale) ? 1.0f : scale / sqrtf(r2);
    return fmaxf(1.17549435e-38f, 1.0f / (1.0f + r2 * (1.0f / (scale * scale))));
}

// ------------------------------------------------------------------------------------------------
// cost: one workgroup per edge -> edge_out[e] = {sum of losses, valid count}
//   memory: per residual 4 + 8 B streamed, 8 + 4 B gathered through the keypoint index, 48 B of the cached triangle's vertices
//   (tri_verts: the three vertices side by side, one dependent load instead of index -> vertex).  As in the normal-equation
//   kernel the loads of residual i + 2 (indices) and i + 1 (keypoint, cached triangle) are issued before residual i is worked on.
//   closest-hit fallback (no cached triangle, or the ray misses it): the traversal stack of a lane is a column of LDS, and a
//   stack for every lane of the workgroup (64 KiB) held the kernel at two workgroups per CU although, after the first sweep,
//   hardly a ray needs it.  There is ONE wave's worth of stacks now; a wave with lanes that need the fallback takes the lock on
//   it for the duration of their traversals.  When every ray needs it (first sweep) the four waves of a workgroup take turns,
//   and four times as many workgroups are resident: as many traversing lanes per CU as before.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void refine_tri_verts_kernel(RefineProblemView P, float4* __restrict__ tv) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= P.n_tris) return;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float3 v = load_vertex(P.verts, P.tris[3 * t + k]);
        tv[3 * t + k] = make_float4(v.x, v.y, v.z, 0.f);
    }
}

void launch_refine_tri_verts(const RefineProblemView& P, float4* tri_verts, hipStream_t s) {
    if (P.n_tris <= 0) return;
    hipLaunchKernelGGL(refine_tri_verts_kernel, dim3((P.n_tris + 255) / 256), dim3(256), 0, s, P, tri_verts);
}

__global__ __launch_bounds__(256) void refine_cost_kernel(RefineProblemView P, const RefineCamera* __restrict__ cams,
                                                          int loss_type, float loss_scale,
                                                          double2* __restrict__ edge_out) {
    __shared__ double s_part[4][2];
    __shared__ int s_stack[kBvhStack][64];   // traversal stacks of ONE wave
    __shared__ int s_stack_lock;
    const int e = blockIdx.x;
    const int lane = (int)threadIdx.x & 63;
    if (threadIdx.x == 0) s_stack_lock = 0;
    __syncthreads();
    const int src = P.edge_src[e], tgt = P.edge_tgt[e];
    const RefineCamera cs = cams[src], ct = cams[tgt];
    const int r0 = P.edge_offset[e], r1 = P.edge_offset[e + 1];
    const uint32_t kp_base = (uint32_t)P.kp_offset[src];
    // ray origin: Pose::Center() = -R^T t, then into object space (refiner.cc:310-321)
    const float3 center = scale3(mul3t(cs.R, make_float3(cs.t[0], cs.t[1], cs.t[2])), -1.0f);
    const float3 o_obj = xform_point(P.model_inv, center);
    const float4* __restrict__ tri_verts = P.tri_verts;
    double cost = 0.0, valid = 0.0;  // fp64 sums: the LM accept test compares costs that differ in the 6th digit
    if (r1 > r0) {
        const int last = r1 - 1;
        int r = r0 + (int)threadIdx.x;
        uint32_t kp_a = kp_base + P.res_src_kp[min(r, last)];
        float2 tp_a = P.res_tgt_xy[min(r, last)];
        uint32_t kp_c = kp_base + P.res_src_kp[min(r + 256, last)];
        float2 tp_c = P.res_tgt_xy[min(r + 256, last)];
        uint32_t prim_a = P.prim_cache[kp_a];
        float2 sp_a = P.kp_xy[kp_a];
        // the trip count is uniform over the wave (the lock below is taken by whole waves): lanes past the end idle
        const int trips = (r1 - (r0 + ((int)threadIdx.x & ~63)) + 255) / 256;
        for (int it = 0; it < trips; it++, r += 256) {
            const bool in = r < r1;
            // the cached triangle's vertices first: the load that is waited for
            const bool cached = prim_a != 0xffffffffu;
            const uint32_t prim_safe = cached ? prim_a : 0u;
            const float4 va = tri_verts[3 * prim_safe], vb = tri_verts[3 * prim_safe + 1], vc = tri_verts[3 * prim_safe + 2];
            // the next residual's keypoint and cached triangle, the one after's indices
            const uint32_t kp_n = kp_c;
            const uint32_t prim_n = P.prim_cache[kp_n];
            const float2 sp_n = P.kp_xy[kp_n];
            const float2 tp_n = tp_c;
            kp_c = kp_base + P.res_src_kp[min(r + 512, last)];
            tp_c = P.res_tgt_xy[min(r + 512, last)];

            const float3 dir_cam = make_float3(cs.sign * ((sp_a.x - cs.cx) / cs.fx), cs.sign * ((sp_a.y - cs.cy) / cs.fy), cs.sign);
            const float3 d_obj = xform_dir(P.model_inv, mul3t(cs.R, dir_cam));
            bool found = false;
            float3 p_obj = make_float3(0.f, 0.f, 0.f);
            if (cached)  // cached triangle first (refiner.cc:323-331)
                found = ray_triangle(o_obj, d_obj, make_float3(va.x, va.y, va.z), make_float3(vb.x, vb.y, vb.z), make_float3(vc.x, vc.y, vc.z),
                                     &p_obj) >= 0.f;
            const bool fallback = in && !found;
            if (__any(fallback)) {   // closest hit over the whole mesh, masked closest triangle = miss (:333-345)
                if (lane == 0)
                    while (atomicCAS(&s_stack_lock, 0, 1) != 0) __builtin_amdgcn_s_sleep(4);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (fallback) {
                    float bt, bu, bv;
                    const int best = bvh_closest_hit(P.bvh, o_obj.x, o_obj.y, o_obj.z, d_obj.x, d_obj.y, d_obj.z, &s_stack[0][lane], 64, &bt,
                                                     &bu, &bv);
                    if (best >= 0 && !((P.mask[best >> 5] >> (best & 31)) & 1u)) {
                        // Embree reports the barycentric point; identical to o + t d up to rounding
                        found = true;
                        p_obj = add3(o_obj, scale3(d_obj, bt));
                        P.prim_cache[kp_a] = (uint32_t)best;
                    } else {
                        P.prim_cache[kp_a] = 0xffffffffu;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) atomicExch(&s_stack_lock, 0);
            }
            if (in && found) {
                const float3 pw = xform_point(P.model, p_obj);
                const float3 pc = add3(mul3(ct.R, pw), make_float3(ct.t[0], ct.t[1], ct.t[2]));
                const bool behind = ct.sign > 0.f ? (pc.z < 0.0f) : (pc.z > 0.0f);
                if (!behind) {
                    const float rx = ct.fx * pc.x / pc.z + ct.cx - tp_a.x, ry = ct.fy * pc.y / pc.z + ct.cy - tp_a.y;
                    cost += (double)refine_loss_value(loss_type, loss_scale, rx * rx + ry * ry);
                    valid += 1.0;
                }
            }
            kp_a = kp_n;
            prim_a = prim_n;
            sp_a = sp_n;
            tp_a = tp_n;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        cost += __shfl_xor(cost, d);
        valid += __shfl_xor(valid, d);
    }
    if ((threadIdx.x & 63) == 0) {
        s_part[threadIdx.x >> 6][0] = cost;
        s_part[threadIdx.x >> 6][1] = valid;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        edge_out[e] = make_double2((s_part[0][0] + s_part[1][0]) + (s_part[2][0] + s_part[3][0]),
                                   (s_part[0][1] + s_part[1][1]) + (s_part[2][1] + s_part[3][1]));
}

// ------------------------------------------------------------------------------------------------
// normal equations, register-resident: every lane walks residuals r, r + stride, ... of its edge, evaluates the 2 x 2B
// Jacobian in fp32 registers and adds w J^T J and J^T (w r) into fp64 accumulators of its own; lanes are summed in a fixed
// tree when the edge is done.  No LDS and no barrier in the loop, all lanes busy in both halves of the work (the tiled kernel
// of rounds 2-4 had 128 of 256 lanes evaluating and 90 accumulating out of LDS: 9.65 ms per sweep at 44.7 M residuals).
//   registers: the lower triangle of a 2B x 2B block is 78 (B = 6) or 171 (B = 9) doubles -- with 9 parameters per camera that
//   is more than a lane holds, so the ROWS of the triangle are dealt out to ROLES groups of waves; every group walks all
//   residuals of the edge and evaluates the Jacobian itself (fp32, cheap next to the fp64 sums).
//   memory: per residual 4 + 8 B streamed, 8 + 4 B gathered through the keypoint index, 32 B of the triangle's plane.  The
//   loads of residual i + 2 (indices) and i + 1 (keypoint, cached triangle) are issued before residual i is evaluated: the
//   only load a lane waits for is the plane of its own triangle, issued first (vmcnt counts in order).
// ------------------------------------------------------------------------------------------------
// world-space plane of every triangle: {n.x, n.y, n.z, 0}, {p0.x, p0.y, p0.z, 0} with the expressions the per-residual
// evaluation used (refiner.cc:419-428) -- the model matrix is fixed for the life of a problem
__global__ __launch_bounds__(256) void refine_tri_plane_kernel(RefineProblemView P, float4* __restrict__ plane) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= P.n_tris) return;
    const float3 p1 = load_vertex(P.verts, P.tris[3 * t]), p2 = load_vertex(P.verts, P.tris[3 * t + 1]), p3 = load_vertex(P.verts, P.tris[3 * t + 2]);
    const float3 n_obj = cross3(sub3(p2, p1), sub3(p3, p1));
    // normal = (model_inv^T)_{3x3} * n_obj
    const float3 n = make_float3(P.model_inv[0] * n_obj.x + P.model_inv[4] * n_obj.y + P.model_inv[8] * n_obj.z,
                                 P.model_inv[1] * n_obj.x + P.model_inv[5] * n_obj.y + P.model_inv[9] * n_obj.z,
                                 P.model_inv[2] * n_obj.x + P.model_inv[6] * n_obj.y + P.model_inv[10] * n_obj.z);
    const float3 p0 = make_float3(P.model[0] * p1.x + P.model[1] * p1.y + P.model[2] * p1.z + P.model[3],
                                  P.model[4] * p1.x + P.model[5] * p1.y + P.model[6] * p1.z + P.model[7],
                                  P.model[8] * p1.x + P.model[9] * p1.y + P.model[10] * p1.z + P.model[11]);
    plane[2 * t] = make_float4(n.x, n.y, n.z, 0.f);
    plane[2 * t + 1] = make_float4(p0.x, p0.y, p0.z, 0.f);
}

void launch_refine_tri_planes(const RefineProblemView& P, float4* plane, hipStream_t s) {
    if (P.n_tris <= 0) return;
    hipLaunchKernelGGL(refine_tri_plane_kernel, dim3((P.n_tris + 255) / 256), dim3(256), 0, s, P, plane);
}

// what does not change along an edge
struct NeqEdge {
    RefineCamera cs, ct;
    float3 origin;       // CenterWithJac: dO/dR = Skew(center), dO/dt = -R^T
    float weight;
    uint32_t kp_base;    // first keypoint of the source frame
    bool src_fixed, tgt_fixed;
};

// EvaluateWithJacobian (refiner.cc:363-506) of one residual: J0 / J1 = the two rows of [J_src | J_tgt], wr = w * residual.
// false: the residual does not take part (`usable` false: no cached triangle, refiner.cc:386-390; or a check on the way).
template <int B>
__device__ __forceinline__ bool neq_residual(const NeqEdge& E, bool usable, float2 sp, float2 tp, float4 plane_n, float4 plane_p, int loss_type,
                                             float loss_scale, int opt_f, int opt_pp, float* __restrict__ J0, float* __restrict__ J1,
                                             float* wrx, float* wry, float* w) {
    constexpr int N = 2 * B;
    const RefineCamera& cs = E.cs;
    const RefineCamera& ct = E.ct;
#pragma unroll
    for (int k = 0; k < N; k++) J0[k] = J1[k] = 0.f;
    const float s = cs.sign;
    const float3 dir_cam = make_float3(s * (sp.x - cs.cx) / cs.fx, s * (sp.y - cs.cy) / cs.fy, s);
    const float3 dir_w = mul3t(cs.R, dir_cam);  // DerotateWithJac: d/dDirCam = R^T, d/dR = Skew(dirWorld)
    const float3 n = make_float3(plane_n.x, plane_n.y, plane_n.z), p0 = make_float3(plane_p.x, plane_p.y, plane_p.z);
    // IntersectWithJac(ray, plane) (cpp/ray_casting.h:76-112): fp64 for d.n and t
    const double ddn = (double)dot3(dir_w, n);
    bool ok = !(ddn > -1e-10 && ddn < 1e-10);  // the reference CHECKs; treat as invalid
    const double tpar = (double)dot3(sub3(p0, E.origin), n) / ddn;
    const float tf = (float)tpar, inv_ddn = (float)(1.0 / ddn);
    const float3 X = add3(E.origin, scale3(dir_w, tf));
    // A = I - dir n^T / (d.n);  dX/dOrigin = A, dX/dDir = A * t
    float A[9];
    const float dv[3] = {dir_w.x, dir_w.y, dir_w.z}, nv[3] = {n.x, n.y, n.z};
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) A[3 * i + j] = (i == j ? 1.0f : 0.0f) - dv[i] * nv[j] * inv_ddn;
    const float3 tt = make_float3(ct.t[0], ct.t[1], ct.t[2]);
    const float3 Xc = add3(mul3(ct.R, X), tt);
    const bool behind = ct.sign > 0.f ? (Xc.z < 0.0f) : (Xc.z > 0.0f);
    ok = ok && !behind;
    const float rx = ct.fx * Xc.x / Xc.z + ct.cx - tp.x, ry = ct.fy * Xc.y / Xc.z + ct.cy - tp.y;
    // dp/dXCam (2x3)
    const float d00 = ct.fx / Xc.z, d02 = -ct.fx * Xc.x / (Xc.z * Xc.z);
    const float d11 = ct.fy / Xc.z, d12 = -ct.fy * Xc.y / (Xc.z * Xc.z);
    // dp/dX = dp/dXCam * R_t
    float G[6];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        G[j] = d00 * ct.R[j] + d02 * ct.R[6 + j];
        G[3 + j] = d11 * ct.R[3 + j] + d12 * ct.R[6 + j];
    }
    // H = dp/dX * A  (2x3)
    float H[6];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        H[j] = G[0] * A[j] + G[1] * A[3 + j] + G[2] * A[6 + j];
        H[3 + j] = G[3] * A[j] + G[4] * A[3 + j] + G[5] * A[6 + j];
    }
    if (!E.src_fixed) {
        // J_src[:,0:3] = H * (Skew(origin) + t * Skew(dirWorld)); Skew is linear: = H * Skew(origin + t dir),
        // Skew(v) = [[0,-vz,vy],[vz,0,-vx],[-vy,vx,0]]
        const float3 v = add3(E.origin, scale3(dir_w, tf));
        const float S[9] = {0.f, -v.z, v.y, v.z, 0.f, -v.x, -v.y, v.x, 0.f};
#pragma unroll
        for (int j = 0; j < 3; j++) {
            J0[j] = H[0] * S[j] + H[1] * S[3 + j] + H[2] * S[6 + j];
            J1[j] = H[3] * S[j] + H[4] * S[3 + j] + H[5] * S[6 + j];
        }
        // J_src[:,3:6] = H * (-R_s^T)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            J0[3 + j] = -(H[0] * cs.R[3 * j] + H[1] * cs.R[3 * j + 1] + H[2] * cs.R[3 * j + 2]);
            J1[3 + j] = -(H[3] * cs.R[3 * j] + H[4] * cs.R[3 * j + 1] + H[5] * cs.R[3 * j + 2]);
        }
        if (B == 9) {
            // J_src[:,6:9] = (H t) * R_s^T * dDirCam/dIntrin   (UnprojectWithJac, types.h:100-125)
            float Q[6];  // (H * t) * R_s^T : 2x3
#pragma unroll
            for (int j = 0; j < 3; j++) {
                Q[j] = tf * (H[0] * cs.R[3 * j] + H[1] * cs.R[3 * j + 1] + H[2] * cs.R[3 * j + 2]);
                Q[3 + j] = tf * (H[3] * cs.R[3 * j] + H[4] * cs.R[3 * j + 1] + H[5] * cs.R[3 * j + 2]);
            }
            const float u00 = s * (cs.cx - sp.x) / (cs.fy * cs.fy * cs.aspect), u01 = -s / cs.fx;
            const float u10 = s * (cs.cy - sp.y) / (cs.fy * cs.fy), u12 = -s / cs.fy;
            if (opt_f) {
                J0[6] = Q[0] * u00 + Q[1] * u10;
                J1[6] = Q[3] * u00 + Q[4] * u10;
            }
            if (opt_pp) {
                J0[7] = Q[0] * u01;
                J0[8] = Q[1] * u12;
                J1[7] = Q[3] * u01;
                J1[8] = Q[4] * u12;
            }
        }
    }
    if (!E.tgt_fixed) {
        // J_tgt[:,0:3] = dp/dXCam * R_t * Skew(-X); J_tgt[:,3:6] = dp/dXCam
        const float S[9] = {0.f, X.z, -X.y, -X.z, 0.f, X.x, X.y, -X.x, 0.f};
#pragma unroll
        for (int j = 0; j < 3; j++) {
            J0[B + j] = G[0] * S[j] + G[1] * S[3 + j] + G[2] * S[6 + j];
            J1[B + j] = G[3] * S[j] + G[4] * S[3 + j] + G[5] * S[6 + j];
        }
        J0[B + 3] = d00;
        J0[B + 5] = d02;
        J1[B + 4] = d11;
        J1[B + 5] = d12;
        if (B == 9) {
            if (opt_f) {
                J0[B + 6] = ct.aspect * Xc.x / Xc.z;
                J1[B + 6] = Xc.y / Xc.z;
            }
            if (opt_pp) {
                J0[B + 7] = 1.0f;
                J1[B + 8] = 1.0f;
            }
        }
    }
    const float wgt = E.weight * refine_loss_weight(loss_type, loss_scale, rx * rx + ry * ry);
    // no branch above: a residual that does not take part leaves with zeros (whatever the arithmetic made of it), so that the
    // caller's sums run unconditionally -- a conditional update of ~90 fp64 accumulators costs a register copy of each
    ok = ok && usable;
#pragma unroll
    for (int k = 0; k < N; k++) {
        J0[k] = ok ? J0[k] : 0.f;
        J1[k] = ok ? J1[k] : 0.f;
    }
    *wrx = ok ? wgt * rx : 0.f;
    *wry = ok ? wgt * ry : 0.f;
    *w = ok ? wgt : 0.f;
    return ok;
}

// sum over the wave in a fixed order, by data-parallel primitives (no LDS round trips: 90 of these end an edge): the 16 lanes
// of a row by butterflies, then the four rows in order
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]  (lane ^ 1)
    v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]  (lane ^ 2)
    v += dpp_f64<0x141>(v);   // row_half_mirror      (other quad of the 8-lane half)
    v += dpp_f64<0x140>(v);   // row_mirror           (other half of the row)
    return ((readlane_f64(v, 0) + readlane_f64(v, 16)) + readlane_f64(v, 32)) + readlane_f64(v, 48);
}

// rows [LO, HI) of the lower triangle (+ the same rows of J^T r) over the residuals first, first + stride, ... of edge e, then
// summed over the wave: lane 0 leaves NT + NG + 1 doubles (triangle rows, gradient rows, valid count) in `red`
template <int B, int LO, int HI>
__device__ __forceinline__ void neq_accumulate(const RefineProblemView& P, const NeqEdge& E, int e, int first, int stride, int loss_type,
                                               float loss_scale, int opt_f, int opt_pp, double* __restrict__ red) {
    constexpr int N = 2 * B;
    constexpr int T0 = LO * (LO + 1) / 2, NT = HI * (HI + 1) / 2 - T0, NG = HI - LO;
    double tri[NT], g[NG];
#pragma unroll
    for (int k = 0; k < NT; k++) tri[k] = 0.0;
#pragma unroll
    for (int k = 0; k < NG; k++) g[k] = 0.0;
    int n_valid = 0;
    const int r0 = P.edge_offset[e], r1 = P.edge_offset[e + 1];
    const uint32_t kp_base = E.kp_base;
    const float4* __restrict__ planes = P.tri_plane;
    if (r1 > r0) {
        const int last = r1 - 1;
        // software pipeline: indices two residuals ahead, keypoint + cached triangle one ahead
        int r = r0 + first;
        uint32_t kp_b = P.res_src_kp[min(r, last)];
        float2 tp_a = P.res_tgt_xy[min(r, last)];
        uint32_t kp_c = P.res_src_kp[min(r + stride, last)];
        float2 tp_c = P.res_tgt_xy[min(r + stride, last)];
        uint32_t prim_a = P.prim_cache[kp_base + kp_b];
        float2 sp_a = P.kp_xy[kp_base + kp_b];
        for (; r < r1; r += stride) {
            // this residual's plane first: the one load that is waited for
            const bool cached = prim_a != 0xffffffffu;   // refiner.cc:386-390
            const uint32_t prim_safe = cached ? prim_a : 0u;
            const float4 pn = planes[2 * prim_safe], pp = planes[2 * prim_safe + 1];
            // the next residual's keypoint and cached triangle, the one after's indices
            const uint32_t prim_n = P.prim_cache[kp_base + kp_c];
            const float2 sp_n = P.kp_xy[kp_base + kp_c];
            const float2 tp_n = tp_c;
            kp_c = P.res_src_kp[min(r + 2 * stride, last)];
            tp_c = P.res_tgt_xy[min(r + 2 * stride, last)];
            float J0[N], J1[N], wrx, wry, w;
            n_valid += neq_residual<B>(E, cached, sp_a, tp_a, pn, pp, loss_type, loss_scale, opt_f, opt_pp, J0, J1, &wrx, &wry, &w) ? 1 : 0;
            {
                const double wd = (double)w, rxd = (double)wrx, ryd = (double)wry;
                double d0[HI], d1[HI];
#pragma unroll
                for (int b = 0; b < HI; b++) {
                    d0[b] = (double)J0[b];
                    d1[b] = (double)J1[b];
                }
#pragma unroll
                for (int a = LO; a < HI; a++) {
                    const double a0 = wd * d0[a], a1 = wd * d1[a];   // exact: products of two floats
#pragma unroll
                    for (int b = 0; b <= a; b++) {
                        const int k = a * (a + 1) / 2 + b - T0;
                        tri[k] = fma(a0, d0[b], fma(a1, d1[b], tri[k]));
                    }
                    g[a - LO] = fma(d0[a], rxd, fma(d1[a], ryd, g[a - LO]));
                }
            }
            prim_a = prim_n;
            sp_a = sp_n;
            tp_a = tp_n;
        }
    }
    const int lane = (int)threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < NT; k++) {
        const double v = wave_sum_f64(tri[k]);
        if (lane == 0) red[k] = v;
    }
#pragma unroll
    for (int k = 0; k < NG; k++) {
        const double v = wave_sum_f64(g[k]);
        if (lane == 0) red[NT + k] = v;
    }
    const double v = wave_sum_f64((double)n_valid);
    if (lane == 0) red[NT + NG] = v;
}

// the role's waves summed in wave order, normalised by the edge's valid count (kShouldNormalize, lev_marq.h:705-710)
template <int B, int LO, int HI>
__device__ __forceinline__ void neq_finalize(int e, int first, int stride, int waves, int red_stride, const double* __restrict__ red,
                                             double* __restrict__ edge_blocks, int* __restrict__ edge_valid) {
    constexpr int N = 2 * B;
    constexpr int T0 = LO * (LO + 1) / 2, NT = HI * (HI + 1) / 2 - T0, NG = HI - LO, NACC = N * (N + 1) / 2 + N;
    double nv = 0.0;
    for (int wv = 0; wv < waves; wv++) nv += red[wv * red_stride + NT + NG];
    double* out = edge_blocks + (size_t)e * NACC;
    for (int k = first; k < NT + NG; k += stride) {
        double v = 0.0;
        for (int wv = 0; wv < waves; wv++) v += red[wv * red_stride + k];
        if (nv > 0.0) v /= nv;
        out[k < NT ? T0 + k : N * (N + 1) / 2 + LO + (k - NT)] = v;
    }
    if (LO == 0 && first == 0) edge_valid[e] = (int)nv;
}

// rows of the triangle per role: balanced by the number of entries
template <int B, int ROLES>
struct NeqRows;
template <>
struct NeqRows<6, 1> {
    static constexpr int at[2] = {0, 12};
};
template <>
struct NeqRows<9, 2> {
    static constexpr int at[3] = {0, 13, 18};
};

template <int B, int ROLES, int MIN_WAVES>
__global__ __launch_bounds__(256, MIN_WAVES) void refine_normal_eq_kernel(RefineProblemView P, const RefineCamera* __restrict__ cams,
                                                               int loss_type, float loss_scale, int opt_f, int opt_pp,
                                                               double* __restrict__ edge_blocks, int* __restrict__ edge_valid) {
    constexpr int N = 2 * B, NRMAX = N * (N + 1) / 2 + N + 1;
    constexpr int WPR = 4 / ROLES;   // waves per role
    using Rows = NeqRows<B, ROLES>;
    __shared__ double s_red[4 * NRMAX];
    const int e = blockIdx.x;
    const int src = P.edge_src[e], tgt = P.edge_tgt[e];
    NeqEdge E;
    E.cs = cams[src];
    E.ct = cams[tgt];
    E.origin = scale3(mul3t(E.cs.R, make_float3(E.cs.t[0], E.cs.t[1], E.cs.t[2])), -1.0f);
    E.weight = P.edge_weight[e];
    E.kp_base = (uint32_t)P.kp_offset[src];
    E.src_fixed = P.frame_fixed[src] != 0;
    E.tgt_fixed = P.frame_fixed[tgt] != 0;
    const int wave = (int)threadIdx.x >> 6;
    const int role = wave / WPR, first = (int)threadIdx.x - role * WPR * 64, stride = WPR * 64;
    double* red = s_red + (size_t)wave * NRMAX;
    double* role_red = s_red + (size_t)role * WPR * NRMAX;
    // one branch per role: wave-uniform, and no barrier inside
#define PC_NEQ_ROLE(K)                                                                                                          \
    if (ROLES > K && role == K)                                                                                                 \
        neq_accumulate<B, Rows::at[K < ROLES ? K : 0], Rows::at[K < ROLES ? K + 1 : 1]>(P, E, e, first, stride, loss_type, loss_scale, opt_f, \
                                                                                       opt_pp, red);
    PC_NEQ_ROLE(0)
    PC_NEQ_ROLE(1)
    PC_NEQ_ROLE(2)
    PC_NEQ_ROLE(3)
#undef PC_NEQ_ROLE
    __syncthreads();
#define PC_NEQ_ROLE(K)                                                                                                          \
    if (ROLES > K && role == K)                                                                                                 \
        neq_finalize<B, Rows::at[K < ROLES ? K : 0], Rows::at[K < ROLES ? K + 1 : 1]>(e, first, stride, WPR, NRMAX, role_red, edge_blocks, \
                                                                                     edge_valid);
    PC_NEQ_ROLE(0)
    PC_NEQ_ROLE(1)
    PC_NEQ_ROLE(2)
    PC_NEQ_ROLE(3)
#undef PC_NEQ_ROLE
}

// one workgroup per edge: every residual's source keypoint exists (the check of pc_refine_problem_create, where the arrays are)
__global__ __launch_bounds__(256) void refine_validate_kernel(RefineProblemView P, int* __restrict__ bad) {
    const int e = blockIdx.x;
    const int src = P.edge_src[e];
    const uint32_t src_kps = (uint32_t)(P.kp_offset[src + 1] - P.kp_offset[src]);
    const int r1 = P.edge_offset[e + 1];
    uint32_t worst = 0;
    bool any = false;
    for (int r = P.edge_offset[e] + (int)threadIdx.x; r < r1; r += 256) {
        const uint32_t kp = P.res_src_kp[r];
        if (kp >= src_kps) {
            any = true;
            worst = kp > worst ? kp : worst;
        }
    }
    if (any) {
        atomicMin(&bad[0], e);
        atomicMax(&bad[1], (int)(worst & 0x7fffffffu));
    }
}

void launch_refine_validate(const RefineProblemView& P, int* bad, hipStream_t s) {
    if (P.n_edges <= 0) return;
    hipLaunchKernelGGL(refine_validate_kernel, dim3(P.n_edges), dim3(256), 0, s, P, bad);
}

void launch_refine_cost(const RefineProblemView& P, const RefineCamera* cams, int loss_type, float loss_scale,
                        double2* edge_out, hipStream_t s) {
    if (P.n_edges <= 0) return;
    hipLaunchKernelGGL(refine_cost_kernel, dim3(P.n_edges), dim3(256), 0, s, P, cams, loss_type, loss_scale, edge_out);
}

void launch_refine_normal_eq(const RefineProblemView& P, const RefineCamera* cams, int loss_type, float loss_scale,
                             int block_len, int opt_f, int opt_pp, double* edge_blocks, int* edge_valid, hipStream_t s) {
    if (P.n_edges <= 0) return;
    // measured at 44.7 M residuals (C5): B = 6 with all rows on every lane and two waves per SIMD (256 VGPRs, 3 dwords of scratch)
    // 1.8 ms, rows over two roles 3.0 ms; B = 9 with two roles at one wave per SIMD (340 VGPRs) 5.2 ms, four roles at two waves
    // 6.8 ms, two roles squeezed into 256 VGPRs (41 dwords of scratch) 7.3 ms.  The tiled kernel of rounds 2-4: 9.7 / 11.5 ms.
    if (block_len == 9)
        hipLaunchKernelGGL((refine_normal_eq_kernel<9, 2, 1>), dim3(P.n_edges), dim3(256), 0, s, P, cams, loss_type, loss_scale, opt_f,
                           opt_pp, edge_blocks, edge_valid);
    else
        hipLaunchKernelGGL((refine_normal_eq_kernel<6, 1, 2>), dim3(P.n_edges), dim3(256), 0, s, P, cams, loss_type, loss_scale, opt_f,
                           opt_pp, edge_blocks, edge_valid);
}

}  // namespace pc
