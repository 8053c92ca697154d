// bvh.hpp -- linear BVH over the mesh triangles (Karras 2012 construction on Morton-sorted
// centroids) and the per-lane closest-hit traversal shared by the tracker's ray-cast kernel (K12)
// and the refiner's cost kernel (K13).  Replaces the Embree scene of the reference
// (cpp/ray_casting.cc:23-63 build, :65-121 rtcIntersect1).
//
// The per-triangle test is the reference's own Moeller-Trumbore (cpp/ray_casting.h:125-179), so a
// traversal returns exactly what a brute-force sweep over all triangles returns: the smallest t, and
// among equal t the lowest triangle index.  Boxes are padded and the slab test is conservative, so
// the hierarchy only ever prunes triangles the ray cannot reach.
#pragma once

#include "common.hpp"

namespace pc {

// One internal node: the boxes of its two children and their links.  A link >= 0 is an internal node,
// a link < 0 is the leaf (sorted position) ~link.
struct __attribute__((aligned(16))) BvhNode {
    float lo0[3], hi0[3];
    float lo1[3], hi1[3];
    int left, right;
    int pad[2];
};
static_assert(sizeof(BvhNode) == 64, "BvhNode is one 64-byte line");

struct BvhView {
    const BvhNode* nodes;          // n_tris - 1 internal nodes, root = 0 (none when n_tris == 1)
    const uint32_t* leaf_tri;      // sorted position -> triangle index
    const float* verts;            // N x 3
    const uint32_t* tris;          // M x 3
    int n_tris;
};

constexpr int kBvhStack = 64;      // Morton keys are 64-bit and unique: depth <= 64

// Moeller-Trumbore of the reference (cpp/ray_casting.h:125-179); false on a miss
__device__ __forceinline__ bool bvh_ray_triangle(float ox, float oy, float oz, float dx, float dy, float dz, const float* p1,
                                                 const float* p2, const float* p3, float* t_out, float* u_out, float* v_out) {
    const float e1x = p2[0] - p1[0], e1y = p2[1] - p1[1], e1z = p2[2] - p1[2];
    const float e2x = p3[0] - p1[0], e2y = p3[1] - p1[1], e2z = p3[2] - p1[2];
    const float cx = dy * e2z - dz * e2y, cy = dz * e2x - dx * e2z, cz = dx * e2y - dy * e2x;  // dir x edge2
    const float det = e1x * cx + e1y * cy + e1z * cz;
    if (det > -1e-10f && det < 1e-10f) return false;
    const float inv_det = 1.0f / det;
    const float sx = ox - p1[0], sy = oy - p1[1], sz = oz - p1[2];
    const float u = inv_det * (sx * cx + sy * cy + sz * cz);
    if (u < 0.0f || u > 1.0f) return false;
    const float qx = sy * e1z - sz * e1y, qy = sz * e1x - sx * e1z, qz = sx * e1y - sy * e1x;  // s x edge1
    const float v = inv_det * (dx * qx + dy * qy + dz * qz);
    if (v < 0.0f || u + v > 1.0f) return false;
    const float t = inv_det * (e2x * qx + e2y * qy + e2z * qz);
    if (t < 0.0f) return false;
    *t_out = t;
    *u_out = u;
    *v_out = v;
    return true;
}

// entry distance of the ray into a (padded) box, or +inf when it misses or starts beyond `t_max`
__device__ __forceinline__ float bvh_box_entry(const float* lo, const float* hi, float ox, float oy, float oz, float ix, float iy,
                                               float iz, float t_max) {
    // fminf / fmaxf drop a NaN operand (0 * inf when the origin lies on a slab plane of an axis-parallel ray)
    const float ax = (lo[0] - ox) * ix, bx = (hi[0] - ox) * ix;
    const float ay = (lo[1] - oy) * iy, by = (hi[1] - oy) * iy;
    const float az = (lo[2] - oz) * iz, bz = (hi[2] - oz) * iz;
    const float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    const float t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), t_max));
    return (t0 <= t1 * 1.0000004f) ? t0 : __builtin_inff();
}

// Closest hit of one ray (direction not normalised, like the reference).  `stack` holds kBvhStack
// entries `stack_stride` ints apart (a per-lane column of an LDS array).  Returns the triangle index or -1.
__device__ __forceinline__ int bvh_closest_hit(const BvhView& B, float ox, float oy, float oz, float dx, float dy, float dz,
                                               int* stack, int stack_stride, float* t_out, float* u_out, float* v_out) {
    float best_t = __builtin_inff(), best_u = 0.f, best_v = 0.f;
    int best = -1;
    auto test_leaf = [&](int leaf) {
        const int tri = (int)B.leaf_tri[leaf];
        const uint32_t a = B.tris[3 * tri], b = B.tris[3 * tri + 1], c = B.tris[3 * tri + 2];
        float t, u, v;
        if (bvh_ray_triangle(ox, oy, oz, dx, dy, dz, B.verts + 3 * a, B.verts + 3 * b, B.verts + 3 * c, &t, &u, &v) &&
            (t < best_t || (t == best_t && tri < best))) {
            best_t = t;
            best_u = u;
            best_v = v;
            best = tri;
        }
    };
    if (B.n_tris == 1) {
        test_leaf(0);
    } else if (B.n_tris > 1) {
        const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;   // +-inf for axis-parallel rays
        int sp = 0, node = 0;
        for (;;) {
            const BvhNode nd = B.nodes[node];
            // children beyond the current best are skipped; equal distance is kept (tie -> lowest index)
            float e0 = bvh_box_entry(nd.lo0, nd.hi0, ox, oy, oz, ix, iy, iz, best_t);
            float e1 = bvh_box_entry(nd.lo1, nd.hi1, ox, oy, oz, ix, iy, iz, best_t);
            int c0 = nd.left, c1 = nd.right;
            if (e1 < e0) {   // nearer child first
                const float te = e0; e0 = e1; e1 = te;
                const int tc = c0; c0 = c1; c1 = tc;
            }
            int next = -1;
            const bool h0 = e0 != __builtin_inff(), h1 = e1 != __builtin_inff();
            if (h0 && c0 < 0) test_leaf(~c0);
            if (h1 && c1 < 0) test_leaf(~c1);
            if (h0 && c0 >= 0) next = c0;
            if (h1 && c1 >= 0) {
                if (next < 0) next = c1;
                else if (sp < kBvhStack) stack[(sp++) * stack_stride] = c1;
            }
            if (next < 0) {
                if (sp == 0) break;
                next = stack[(--sp) * stack_stride];
            }
            node = next;
        }
    }
    *t_out = best_t;
    *u_out = best_u;
    *v_out = best_v;
    return best;
}

// ---- construction (kernels_bvh.hip) ----
struct BvhBuildScratch {
    unsigned long long* keys_in;   // n
    unsigned long long* keys_out;  // n
    float* box_lo;                 // (2n - 1) x 3: internal nodes [0, n-1), leaves [n-1, 2n-1)
    float* box_hi;
    int* parent;                   // 2n - 1
    int* visits;                   // n - 1
    int* left;                     // n - 1 child links (>= 0 internal node, < 0 leaf ~position)
    int* right;                    // n - 1
    uint32_t* bounds;              // 6 words: centroid bounds as order-preserving integer keys
};
size_t bvh_sort_temp_bytes(int n_tris);
// Builds nodes / leaf_tri for the mesh; everything is stream-ordered on `s`.
// `pad` grows every box (absolute, in mesh units).
hipError_t bvh_build(const float* verts, const uint32_t* tris, int n_tris, float pad, const BvhBuildScratch& sc, void* sort_temp,
                     size_t sort_temp_bytes, BvhNode* nodes, uint32_t* leaf_tri, hipStream_t s);

}  // namespace pc
