// common.hpp -- shared device helpers and plane geometry for the gfx950 kernels.
//
// All float arithmetic that has to match the CPU oracle bit-for-bit is written as separate
// operations and the library is compiled with -ffp-contract=off; hipcc's default correctly rounded
// fp32 divide/sqrt (-fhip-fp32-correctly-rounded-divide-sqrt) is relied upon.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pc {

// Left padding (bytes) of every pyramid plane: >= PC_MAX_WINDOW + 1 (an LK region starts one position left of its window,
// which may start a window's width left of the image), and a multiple of 16 so that interior rows are 16-B aligned.
constexpr int kPadX = 32;

// One pyramid level resident in HBM.  `img` / `der` point at the INTERIOR origin (x=0, y=0); the
// padding (win rows above/below, kPadX bytes left, >= win bytes right) is addressable with negative
// offsets.  `der` is the Scharr plane: one int32 per pixel = (int16 dx) | (int16 dy << 16), zero in
// the padding (OpenCV derivBorder = BORDER_CONSTANT); its pitch in pixels equals `pitch`.
// `img16` is the image once more as uint16 = pixel << 7 (same pitch in pixels, same REFLECT_101 padding, plus
// kImg16SlackRows addressable zero rows above and below): the operand format of the LK inner loop
// (kernels_lk3.hip: a v_dot2_i32_i16 of two such pixels with the 14-bit bilinear weights is 128 x the
// interpolated value, so the rounding shift of CV_DESCALE comes for free as "the high half of the register").
struct Level {
    uint8_t* img;
    int32_t* der;
    uint16_t* img16;
    int w, h;
    int pitch;  // bytes per image row == int32 per derivative row == uint16 per img16 row
};
constexpr int kImg16SlackRows = 2;

// cv::borderInterpolate(p, len, BORDER_REFLECT_101)
__host__ __device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = (p < 0) ? -p : 2 * (len - 1) - p;
    return p;
}

// Order-preserving map float -> uint32 (a < b  <=>  key(a) < key(b)), used for atomicMax and for the
// candidate sort keys (value desc, then linear index desc == gftt.cc:7-12).
__host__ __device__ __forceinline__ uint32_t float_to_ordered(float f) {
    union { float f; uint32_t u; } c;
    c.f = f;
    return (c.u & 0x80000000u) ? ~c.u : (c.u | 0x80000000u);
}
__host__ __device__ __forceinline__ float ordered_to_float(uint32_t k) {
    union { float f; uint32_t u; } c;
    c.u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return c.f;
}

#define PC_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

#ifdef __HIPCC__
// Issue priority of the helper kernels (frame preparation, compaction).  Beside a running LK launch a helper wavefront
// shares its SIMD with three LK wavefronts; at equal priority it gets a quarter of the issue slots.  When the chain of
// helper kernels of a frame -- a dozen dependent launches on one stream -- then takes longer than an LK launch, it and not
// LK sets the step (tools/lane_probe.py at 1080p: 0.25 ms per step with the job lanes alone, 0.35 with the detection
// beside them although it adds only 11 % to the instructions).  s_setprio puts a wavefront in front of the SIMD's
// arbiter: the helpers then run at the speed of a lone wavefront and LK takes every other slot (1080p: 0.31 ms).  When
// the chain keeps up anyway (4K: LK 1.26 ms, the chain 0.6) the low priority is the better one -- the helpers fill the
// slots LK leaves, 1.41 ms per step against 1.47 -- so the priority is a launch argument that the analyzer sets from
// what it observes (api_analyzer.hip: HelperPriorityControl).
__device__ __forceinline__ void helper_priority(int hi) {
    if (hi) __builtin_amdgcn_s_setprio(3);
}

// "The last workgroup finishes the job": true in the workgroup that arrives last (of `total`) at `*ticket` (zero before
// the launch).  Every lane of every workgroup must call it.  What it buys: a dependent single-workgroup step (a scan of
// per-workgroup counts) runs in the tail of the kernel that produced its input instead of as a launch of its own --
// beside a running LK launch every dependent launch of the frame-preparation chain costs ~12 us whatever it computes
// (DESIGN.md section 3).
// NO RELEASE FENCE: the XCDs' L2s are not coherent with one another, so an agent-scope release is an L2 write-back --
// in every workgroup of the launch (measured with a fence pair: the NMS kernel 3x slower, and the LK launch that shares
// those L2s 17 % slower).  Instead the data that crosses workgroups here must be written with agent-scope ATOMICS
// (atomicAdd, pc::publish): those go past the L2 to the coherent level by themselves.  Each lane waits until the memory
// system has acknowledged its own operations (s_waitcnt), the barrier collects the workgroup, then one lane takes the
// ticket.  The last workgroup alone pays one acquire (an L2 invalidate of its XCD) and may then read with plain loads.
__device__ __forceinline__ void publish(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Tickets in two levels: atomics on ONE word from all over the chip serialise at ~14 ns each (8500 workgroups: 120 us,
// measured), so a workgroup counts into the word of its group of 32 and only the last of a group counts the groups.
// `tickets`: last_workgroup_words(total) zeroed words.
constexpr uint32_t kTicketGroup = 32;
__host__ __device__ __forceinline__ uint32_t last_workgroup_words(uint32_t total) { return 1u + (total + kTicketGroup - 1u) / kTicketGroup; }
__device__ __forceinline__ bool last_workgroup(uint32_t* tickets, uint32_t total) {
    __shared__ uint32_t s_is_last;
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    __builtin_amdgcn_s_waitcnt(0);     // vmcnt = lgkmcnt = 0: this lane's stores and atomics are acknowledged
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        const uint32_t id = blockIdx.y * gridDim.x + blockIdx.x;
        const uint32_t group = id / kTicketGroup, n_groups = (total + kTicketGroup - 1u) / kTicketGroup;
        const uint32_t in_group = min(kTicketGroup, total - group * kTicketGroup);
        uint32_t last = 0u;
        if (atomicAdd(&tickets[1u + group], 1u) == in_group - 1u) last = (atomicAdd(&tickets[0], 1u) == n_groups - 1u) ? 1u : 0u;
        s_is_last = last;
    }
    __syncthreads();
    const bool last = s_is_last != 0u;
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    return last;
}

// Exclusive scan of in[0..n) into out[0..n) (may alias) by one 256-lane workgroup; returns the total (in every lane).
__device__ __forceinline__ uint32_t scan_exclusive_256(const uint32_t* in, uint32_t* out, int n, uint32_t* s_sum /* [256] */) {
    const int tid = (int)threadIdx.x;
    const int per = (n + 255) / 256;
    const int b = tid * per, e = min(b + per, n);
    uint32_t s = 0;
    for (int i = b; i < e; i++) s += in[i];
    __syncthreads();
    s_sum[tid] = s;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t v = (tid >= d) ? s_sum[tid - d] : 0u;
        __syncthreads();
        s_sum[tid] += v;
        __syncthreads();
    }
    uint32_t run = s_sum[tid] - s;
    for (int i = b; i < e; i++) {
        const uint32_t c = in[i];
        out[i] = run;
        run += c;
    }
    return s_sum[255];
}
// the same for n = 256 * PER words, PER a multiple of 4: every lane's loads are in flight together
template <int PER>
__device__ __forceinline__ uint32_t scan_exclusive_256_fixed(const uint32_t* in, uint32_t* out, uint32_t* s_sum /* [256] */) {
    static_assert(PER % 4 == 0, "whole uint4s");
    const int tid = (int)threadIdx.x;
    uint4 v[PER / 4];
#pragma unroll
    for (int k = 0; k < PER / 4; k++) v[k] = reinterpret_cast<const uint4*>(in + tid * PER)[k];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < PER / 4; k++) s += v[k].x + v[k].y + v[k].z + v[k].w;
    __syncthreads();
    s_sum[tid] = s;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t t = (tid >= d) ? s_sum[tid - d] : 0u;
        __syncthreads();
        s_sum[tid] += t;
        __syncthreads();
    }
    uint32_t run = s_sum[tid] - s;
#pragma unroll
    for (int k = 0; k < PER / 4; k++) {
        uint4 o;
        o.x = run; run += v[k].x;
        o.y = run; run += v[k].y;
        o.z = run; run += v[k].z;
        o.w = run; run += v[k].w;
        reinterpret_cast<uint4*>(out + tid * PER)[k] = o;
    }
    return s_sum[255];
}
#endif

}  // namespace pc
