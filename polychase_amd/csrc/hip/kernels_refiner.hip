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
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void refine_cost_kernel(RefineProblemView P, const RefineCamera* __restrict__ cams,
                                                          int loss_type, float loss_scale,
                                                          double2* __restrict__ edge_out) {
    __shared__ double s_part[4][2];
    __shared__ int s_stack[kBvhStack][256];   // per-lane traversal stacks
    const int e = blockIdx.x;
    const int src = P.edge_src[e], tgt = P.edge_tgt[e];
    const RefineCamera cs = cams[src], ct = cams[tgt];
    const int r0 = P.edge_offset[e], r1 = P.edge_offset[e + 1];
    const int kp_base = P.kp_offset[src];
    // ray origin: Pose::Center() = -R^T t, then into object space (refiner.cc:310-321)
    const float3 center = scale3(mul3t(cs.R, make_float3(cs.t[0], cs.t[1], cs.t[2])), -1.0f);
    const float3 o_obj = xform_point(P.model_inv, center);
    double cost = 0.0, valid = 0.0;  // fp64 sums: the LM accept test compares costs that differ in the 6th digit
    for (int r = r0 + (int)threadIdx.x; r < r1; r += 256) {
        const uint32_t kp = (uint32_t)kp_base + P.res_src_kp[r];
        const float2 sp = P.kp_xy[kp];
        const float3 dir_cam = make_float3(cs.sign * ((sp.x - cs.cx) / cs.fx), cs.sign * ((sp.y - cs.cy) / cs.fy), cs.sign);
        const float3 d_obj = xform_dir(P.model_inv, mul3t(cs.R, dir_cam));
        bool found = false;
        float3 p_obj = make_float3(0.f, 0.f, 0.f);
        uint32_t prim = P.prim_cache[kp];
        if (prim != 0xffffffffu) {  // cached triangle first (refiner.cc:323-331)
            const uint32_t a = P.tris[3 * prim], b = P.tris[3 * prim + 1], c = P.tris[3 * prim + 2];
            found = ray_triangle(o_obj, d_obj, load_vertex(P.verts, a), load_vertex(P.verts, b), load_vertex(P.verts, c), &p_obj) >= 0.f;
        }
        if (!found) {  // closest hit over the whole mesh, masked closest triangle = miss (:333-345)
            float bt, bu, bv;
            const int best = bvh_closest_hit(P.bvh, o_obj.x, o_obj.y, o_obj.z, d_obj.x, d_obj.y, d_obj.z, &s_stack[0][threadIdx.x], 256,
                                             &bt, &bu, &bv);
            if (best >= 0 && !((P.mask[best >> 5] >> (best & 31)) & 1u)) {
                // Embree reports the barycentric point; identical to o + t d up to rounding
                found = true;
                p_obj = add3(o_obj, scale3(d_obj, bt));
                P.prim_cache[kp] = (uint32_t)best;
            } else {
                P.prim_cache[kp] = 0xffffffffu;
            }
        }
        if (!found) continue;
        const float3 pw = xform_point(P.model, p_obj);
        const float3 pc = add3(mul3(ct.R, pw), make_float3(ct.t[0], ct.t[1], ct.t[2]));
        const bool behind = ct.sign > 0.f ? (pc.z < 0.0f) : (pc.z > 0.0f);
        if (behind) continue;
        const float2 tp = P.res_tgt_xy[r];
        const float rx = ct.fx * pc.x / pc.z + ct.cx - tp.x, ry = ct.fy * pc.y / pc.z + ct.cy - tp.y;
        cost += (double)refine_loss_value(loss_type, loss_scale, rx * rx + ry * ry);
        valid += 1.0;
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
// normal equations: one workgroup per edge.  Residuals are processed in tiles of RT; each lane
// evaluates one residual's 2 x 2B Jacobian into LDS, then the (2B)(2B+1)/2 + 2B accumulators are
// spread over the lanes and summed over the tile.
// ------------------------------------------------------------------------------------------------
constexpr int RT = 128;

template <int B>
__global__ __launch_bounds__(256) void refine_normal_eq_kernel(RefineProblemView P, const RefineCamera* __restrict__ cams,
                                                               int loss_type, float loss_scale, int opt_f, int opt_pp,
                                                               double* __restrict__ edge_blocks, int* __restrict__ edge_valid) {
    constexpr int N = 2 * B;                   // columns of J_pair
    constexpr int NACC = N * (N + 1) / 2 + N;  // lower triangle of JtJ_pair + Jtr_pair
    __shared__ float s_J[RT][2 * N + 3];       // per residual: J row 0, J row 1, w*rx, w*ry, w
    __shared__ int s_valid;
    const int e = blockIdx.x;
    const int src = P.edge_src[e], tgt = P.edge_tgt[e];
    const RefineCamera cs = cams[src], ct = cams[tgt];
    const int r0 = P.edge_offset[e], r1 = P.edge_offset[e + 1];
    const int kp_base = P.kp_offset[src];
    if (threadIdx.x == 0) s_valid = 0;

    // accumulators owned by this lane: entry ids a = tid, tid + 256, ...
    constexpr int PER = (NACC + 255) / 256;
    double acc[PER];  // fp64: J^T J stays positive semi-definite to 1e-16, which the weakly damped chain needs
    int ia[PER], ib[PER];  // (row, col) of the entry, or (i, -1) for Jtr
#pragma unroll
    for (int k = 0; k < PER; k++) {
        acc[k] = 0.0;
        const int id = (int)threadIdx.x + 256 * k;
        ia[k] = -1;
        ib[k] = -1;
        if (id < N * (N + 1) / 2) {
            int row = 0;
            while ((row + 1) * (row + 2) / 2 <= id) row++;
            ia[k] = row;
            ib[k] = id - row * (row + 1) / 2;
        } else if (id < NACC) {
            ia[k] = id - N * (N + 1) / 2;
        }
    }

    // source-camera quantities shared by all residuals of the edge
    const float3 ts = make_float3(cs.t[0], cs.t[1], cs.t[2]);
    const float3 origin = scale3(mul3t(cs.R, ts), -1.0f);  // CenterWithJac: dO/dR = Skew(center), dO/dt = -R^T
    for (int base = r0; base < r1; base += RT) {
        __syncthreads();
        const int r = base + (int)threadIdx.x;
        if (threadIdx.x < RT) {
            float* row = s_J[threadIdx.x];
#pragma unroll
            for (int k = 0; k < 2 * N + 3; k++) row[k] = 0.f;
            bool ok = r < r1;
            uint32_t prim = 0xffffffffu;
            uint32_t kp = 0;
            if (ok) {
                kp = (uint32_t)kp_base + P.res_src_kp[r];
                prim = P.prim_cache[kp];
                ok = prim != 0xffffffffu;  // refiner.cc:386-390
            }
            if (ok) {
                const float2 sp = P.kp_xy[kp];
                const float s = cs.sign;
                const float3 dir_cam = make_float3(s * (sp.x - cs.cx) / cs.fx, s * (sp.y - cs.cy) / cs.fy, s);
                const float3 dir_w = mul3t(cs.R, dir_cam);  // DerotateWithJac: d/dDirCam = R^T, d/dR = Skew(dirWorld)
                // plane of the cached triangle in world space (refiner.cc:419-428)
                const float3 p1 = load_vertex(P.verts, P.tris[3 * prim]), p2 = load_vertex(P.verts, P.tris[3 * prim + 1]),
                             p3 = load_vertex(P.verts, P.tris[3 * prim + 2]);
                const float3 n_obj = cross3(sub3(p2, p1), sub3(p3, p1));
                // normal = (model_inv^T)_{3x3} * n_obj
                const float3 n = make_float3(P.model_inv[0] * n_obj.x + P.model_inv[4] * n_obj.y + P.model_inv[8] * n_obj.z,
                                             P.model_inv[1] * n_obj.x + P.model_inv[5] * n_obj.y + P.model_inv[9] * n_obj.z,
                                             P.model_inv[2] * n_obj.x + P.model_inv[6] * n_obj.y + P.model_inv[10] * n_obj.z);
                const float3 p0 = make_float3(P.model[0] * p1.x + P.model[1] * p1.y + P.model[2] * p1.z + P.model[3],
                                              P.model[4] * p1.x + P.model[5] * p1.y + P.model[6] * p1.z + P.model[7],
                                              P.model[8] * p1.x + P.model[9] * p1.y + P.model[10] * p1.z + P.model[11]);
                // IntersectWithJac(ray, plane) (cpp/ray_casting.h:76-112): fp64 for d.n and t
                const double ddn = (double)dot3(dir_w, n);
                if (ddn > -1e-10 && ddn < 1e-10) ok = false;  // the reference CHECKs; treat as invalid
                if (ok) {
                    const double tpar = (double)dot3(sub3(p0, origin), n) / ddn;
                    const float tf = (float)tpar, inv_ddn = (float)(1.0 / ddn);
                    const float3 X = add3(origin, scale3(dir_w, tf));
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
                    if (behind) ok = false;
                    if (ok) {
                        const float2 tp = P.res_tgt_xy[r];
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
                        if (!P.frame_fixed[src]) {
                            // J_src[:,0:3] = H * (Skew(origin) + t * Skew(dirWorld)) ; Skew(v) = [[0,-vz,vy],[vz,0,-vx],[-vy,vx,0]]
                            const float3 m = add3(origin, scale3(dir_w, tf));  // Skew is linear: Skew(origin) + t Skew(dir) = Skew(origin + t dir)
                            (void)m;
                            const float3 so = origin, sd = scale3(dir_w, tf);
                            const float3 v = add3(so, sd);
                            const float S[9] = {0.f, -v.z, v.y, v.z, 0.f, -v.x, -v.y, v.x, 0.f};
#pragma unroll
                            for (int j = 0; j < 3; j++) {
                                row[j] = H[0] * S[j] + H[1] * S[3 + j] + H[2] * S[6 + j];
                                row[N + j] = H[3] * S[j] + H[4] * S[3 + j] + H[5] * S[6 + j];
                            }
                            // J_src[:,3:6] = H * (-R_s^T)
#pragma unroll
                            for (int j = 0; j < 3; j++) {
                                row[3 + j] = -(H[0] * cs.R[3 * j] + H[1] * cs.R[3 * j + 1] + H[2] * cs.R[3 * j + 2]);
                                row[N + 3 + j] = -(H[3] * cs.R[3 * j] + H[4] * cs.R[3 * j + 1] + H[5] * cs.R[3 * j + 2]);
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
                                    row[6] = Q[0] * u00 + Q[1] * u10;
                                    row[N + 6] = Q[3] * u00 + Q[4] * u10;
                                }
                                if (opt_pp) {
                                    row[7] = Q[0] * u01;
                                    row[8] = Q[1] * u12;
                                    row[N + 7] = Q[3] * u01;
                                    row[N + 8] = Q[4] * u12;
                                }
                            }
                        }
                        if (!P.frame_fixed[tgt]) {
                            // J_tgt[:,0:3] = dp/dXCam * R_t * Skew(-X); J_tgt[:,3:6] = dp/dXCam
                            const float S[9] = {0.f, X.z, -X.y, -X.z, 0.f, X.x, X.y, -X.x, 0.f};
#pragma unroll
                            for (int j = 0; j < 3; j++) {
                                row[B + j] = G[0] * S[j] + G[1] * S[3 + j] + G[2] * S[6 + j];
                                row[N + B + j] = G[3] * S[j] + G[4] * S[3 + j] + G[5] * S[6 + j];
                            }
                            row[B + 3] = d00; row[B + 5] = d02;
                            row[N + B + 4] = d11; row[N + B + 5] = d12;
                            if (B == 9) {
                                if (opt_f) {
                                    row[B + 6] = ct.aspect * Xc.x / Xc.z;
                                    row[N + B + 6] = Xc.y / Xc.z;
                                }
                                if (opt_pp) {
                                    row[B + 7] = 1.0f;
                                    row[N + B + 8] = 1.0f;
                                }
                            }
                        }
                        const float wgt = P.edge_weight[e] * refine_loss_weight(loss_type, loss_scale, rx * rx + ry * ry);
                        row[2 * N] = wgt * rx;
                        row[2 * N + 1] = wgt * ry;
                        row[2 * N + 2] = wgt;
                        atomicAdd(&s_valid, 1);
                    }
                }
            }
        }
        __syncthreads();
        const int cnt = min(RT, r1 - base);
#pragma unroll
        for (int k = 0; k < PER; k++) {
            if (ia[k] < 0) continue;
            double sum = 0.0;
            if (ib[k] >= 0) {
                for (int i = 0; i < cnt; i++) {
                    const float* row = s_J[i];
                    sum += (double)row[2 * N + 2] *
                           ((double)row[ia[k]] * (double)row[ib[k]] + (double)row[N + ia[k]] * (double)row[N + ib[k]]);
                }
            } else {
                for (int i = 0; i < cnt; i++) {
                    const float* row = s_J[i];
                    sum += (double)row[ia[k]] * (double)row[2 * N] + (double)row[N + ia[k]] * (double)row[2 * N + 1];
                }
            }
            acc[k] += sum;
        }
    }
    __syncthreads();
    // per-edge normalisation by the number of valid residuals (kShouldNormalize, lev_marq.h:705-710)
    const int nv = s_valid;
    double* out = edge_blocks + (size_t)e * NACC;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int id = (int)threadIdx.x + 256 * k;
        if (id < NACC) out[id] = (nv > 0) ? acc[k] / (double)nv : acc[k];
    }
    if (threadIdx.x == 0) edge_valid[e] = nv;
}

void launch_refine_cost(const RefineProblemView& P, const RefineCamera* cams, int loss_type, float loss_scale,
                        double2* edge_out, hipStream_t s) {
    if (P.n_edges <= 0) return;
    hipLaunchKernelGGL(refine_cost_kernel, dim3(P.n_edges), dim3(256), 0, s, P, cams, loss_type, loss_scale, edge_out);
}

void launch_refine_normal_eq(const RefineProblemView& P, const RefineCamera* cams, int loss_type, float loss_scale,
                             int block_len, int opt_f, int opt_pp, double* edge_blocks, int* edge_valid, hipStream_t s) {
    if (P.n_edges <= 0) return;
    if (block_len == 9)
        hipLaunchKernelGGL(refine_normal_eq_kernel<9>, dim3(P.n_edges), dim3(256), 0, s, P, cams, loss_type, loss_scale, opt_f,
                           opt_pp, edge_blocks, edge_valid);
    else
        hipLaunchKernelGGL(refine_normal_eq_kernel<6>, dim3(P.n_edges), dim3(256), 0, s, P, cams, loss_type, loss_scale, opt_f,
                           opt_pp, edge_blocks, edge_valid);
}

}  // namespace pc
