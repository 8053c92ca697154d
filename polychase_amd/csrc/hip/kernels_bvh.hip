// kernels_bvh.hip -- LBVH construction on the GPU (Karras, "Maximizing Parallelism in the
// Construction of BVHs, Octrees, and k-d Trees", HPG 2012): Morton codes of the triangle centroids,
// radix sort, one lane per internal node for the topology, bottom-up box fitting with one atomic
// counter per node.  Replaces rtcCommitScene of the reference (cpp/ray_casting.cc:23-63).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "bvh.hpp"

namespace pc {

namespace {

__device__ __forceinline__ uint32_t expand_bits10(uint32_t v) {  // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__device__ __forceinline__ void tri_centroid(const float* __restrict__ verts, const uint32_t* __restrict__ tris, int t, float c[3]) {
    const uint32_t a = tris[3 * t], b = tris[3 * t + 1], d = tris[3 * t + 2];
#pragma unroll
    for (int k = 0; k < 3; k++) c[k] = (verts[3 * a + k] + verts[3 * b + k] + verts[3 * d + k]) * (1.0f / 3.0f);
}

__global__ __launch_bounds__(256) void bvh_bounds_init_kernel(uint32_t* bounds) {
    if (threadIdx.x < 3) bounds[threadIdx.x] = 0xffffffffu;      // min of ordered keys
    else if (threadIdx.x < 6) bounds[threadIdx.x] = 0u;          // max
}

// centroid bounds: block reduction, then atomics on order-preserving integer keys
__global__ __launch_bounds__(256) void bvh_bounds_kernel(const float* __restrict__ verts, const uint32_t* __restrict__ tris, int n,
                                                         uint32_t* __restrict__ bounds) {
    __shared__ float s_lo[4][3], s_hi[4][3];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    if (i < n) {
        float c[3];
        tri_centroid(verts, tris, i, c);
#pragma unroll
        for (int k = 0; k < 3; k++) lo[k] = hi[k] = c[k];
    }
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], d));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], d));
        }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            s_lo[threadIdx.x >> 6][k] = lo[k];
            s_hi[threadIdx.x >> 6][k] = hi[k];
        }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        const float l = fminf(fminf(s_lo[0][k], s_lo[1][k]), fminf(s_lo[2][k], s_lo[3][k]));
        const float h = fmaxf(fmaxf(s_hi[0][k], s_hi[1][k]), fmaxf(s_hi[2][k], s_hi[3][k]));
        if (l <= h) {
            atomicMin(&bounds[k], float_to_ordered(l));
            atomicMax(&bounds[3 + k], float_to_ordered(h));
        }
    }
}

// key = 30-bit Morton code of the centroid << 32 | triangle index (unique, so every split is defined)
__global__ __launch_bounds__(256) void bvh_morton_kernel(const float* __restrict__ verts, const uint32_t* __restrict__ tris, int n,
                                                         const uint32_t* __restrict__ bounds, unsigned long long* __restrict__ keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float c[3];
    tri_centroid(verts, tris, i, c);
    uint32_t q[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float lo = ordered_to_float(bounds[k]), hi = ordered_to_float(bounds[3 + k]);
        const float ext = hi - lo;
        float f = ext > 0.f ? (c[k] - lo) / ext : 0.f;
        f = fminf(fmaxf(f * 1024.0f, 0.0f), 1023.0f);
        q[k] = (uint32_t)f;
    }
    const uint32_t code = (expand_bits10(q[0]) << 2) | (expand_bits10(q[1]) << 1) | expand_bits10(q[2]);
    keys[i] = ((unsigned long long)code << 32) | (unsigned long long)(uint32_t)i;
}

// length of the common prefix of keys i and j, -1 outside [0, n)
__device__ __forceinline__ int bvh_delta(const unsigned long long* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));
}

// topology: one lane per internal node (Karras 2012, algorithm of section 4)
__global__ __launch_bounds__(256) void bvh_topology_kernel(const unsigned long long* __restrict__ keys, int n, int* __restrict__ left,
                                                           int* __restrict__ right, int* __restrict__ parent,
                                                           uint32_t* __restrict__ leaf_tri, int* __restrict__ visits) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) leaf_tri[i] = (uint32_t)(keys[i] & 0xffffffffull);
    if (i >= n - 1) return;
    visits[i] = 0;
    const int d = (bvh_delta(keys, n, i, i + 1) - bvh_delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = bvh_delta(keys, n, i, i - d);
    int lmax = 2;
    while (bvh_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (bvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = bvh_delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (bvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const int lc = (lo == gamma) ? ~gamma : gamma;            // leaf links are ~position
    const int rc = (hi == gamma + 1) ? ~(gamma + 1) : (gamma + 1);
    left[i] = lc;
    right[i] = rc;
    parent[lc < 0 ? (n - 1) + ~lc : lc] = i;
    parent[rc < 0 ? (n - 1) + ~rc : rc] = i;
    if (i == 0) parent[0] = -1;
}

// boxes: every leaf walks up; the second visitor of a node owns both children's boxes
__global__ __launch_bounds__(256) void bvh_fit_kernel(const float* __restrict__ verts, const uint32_t* __restrict__ tris,
                                                      const uint32_t* __restrict__ leaf_tri, int n, const int* __restrict__ left,
                                                      const int* __restrict__ right, const int* __restrict__ parent,
                                                      int* __restrict__ visits, float* __restrict__ box_lo, float* __restrict__ box_hi,
                                                      float pad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t = (int)leaf_tri[i];
    const uint32_t a = tris[3 * t], b = tris[3 * t + 1], c = tris[3 * t + 2];
    int self = (n - 1) + i;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float x = verts[3 * a + k], y = verts[3 * b + k], z = verts[3 * c + k];
        box_lo[3 * self + k] = fminf(fminf(x, y), z) - pad;
        box_hi[3 * self + k] = fmaxf(fmaxf(x, y), z) + pad;
    }
    if (n == 1) return;
    int node = parent[self];
    while (node >= 0) {
        __threadfence();                                   // publish this subtree's boxes
        if (atomicAdd(&visits[node], 1) == 0) return;      // first visitor: the sibling subtree is not finished
        __threadfence();
        const int lc = left[node], rc = right[node];
        const int li = lc < 0 ? (n - 1) + ~lc : lc, ri = rc < 0 ? (n - 1) + ~rc : rc;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            // agent-scope loads: the sibling's boxes were written by a wave that may sit on another XCD
            const float l0 = __hip_atomic_load(&box_lo[3 * li + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float l1 = __hip_atomic_load(&box_lo[3 * ri + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float h0 = __hip_atomic_load(&box_hi[3 * li + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float h1 = __hip_atomic_load(&box_hi[3 * ri + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&box_lo[3 * node + k], fminf(l0, l1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&box_hi[3 * node + k], fmaxf(h0, h1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        node = parent[node];
    }
}

__global__ __launch_bounds__(256) void bvh_pack_kernel(int n, const int* __restrict__ left, const int* __restrict__ right,
                                                       const float* __restrict__ box_lo, const float* __restrict__ box_hi,
                                                       BvhNode* __restrict__ nodes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int lc = left[i], rc = right[i];
    const int li = lc < 0 ? (n - 1) + ~lc : lc, ri = rc < 0 ? (n - 1) + ~rc : rc;
    BvhNode nd;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        nd.lo0[k] = box_lo[3 * li + k];
        nd.hi0[k] = box_hi[3 * li + k];
        nd.lo1[k] = box_lo[3 * ri + k];
        nd.hi1[k] = box_hi[3 * ri + k];
    }
    nd.left = lc;
    nd.right = rc;
    nd.pad[0] = nd.pad[1] = 0;
    nodes[i] = nd;
}

}  // namespace

size_t bvh_sort_temp_bytes(int n_tris) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_keys(nullptr, bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (size_t)std::max(n_tris, 1),
                                   0, 64, (hipStream_t) nullptr);
    return bytes;
}

hipError_t bvh_build(const float* verts, const uint32_t* tris, int n, float pad, const BvhBuildScratch& sc, void* sort_temp,
                     size_t sort_temp_bytes, BvhNode* nodes, uint32_t* leaf_tri, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const dim3 blk(256), grid((unsigned)((n + 255) / 256));
    uint32_t* bounds = sc.bounds;
    hipLaunchKernelGGL(bvh_bounds_init_kernel, dim3(1), dim3(64), 0, s, bounds);
    hipLaunchKernelGGL(bvh_bounds_kernel, grid, blk, 0, s, verts, tris, n, bounds);
    hipLaunchKernelGGL(bvh_morton_kernel, grid, blk, 0, s, verts, tris, n, bounds, sc.keys_in);
    hipError_t e = rocprim::radix_sort_keys(sort_temp, sort_temp_bytes, sc.keys_in, sc.keys_out, (size_t)n, 0, 64, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bvh_topology_kernel, grid, blk, 0, s, sc.keys_out, n, sc.left, sc.right, sc.parent, leaf_tri, sc.visits);
    // `pad`: every box is grown by a small fraction of the scene size, so that a ray grazing a box face is not pruned
    hipLaunchKernelGGL(bvh_fit_kernel, grid, blk, 0, s, verts, tris, leaf_tri, n, sc.left, sc.right, sc.parent, sc.visits, sc.box_lo,
                       sc.box_hi, pad);
    if (n > 1) hipLaunchKernelGGL(bvh_pack_kernel, grid, blk, 0, s, n, sc.left, sc.right, sc.box_lo, sc.box_hi, nodes);
    return hipGetLastError();
}

}  // namespace pc
