// coresidency.hip -- what can run BESIDE the LK launches?  (measurement aid, not part of the library)
//
// The LK kernel keeps three wavefronts of 136 VGPRs on every SIMD: 104 registers per lane and ~37 KB of LDS per CU are
// left.  This file launches, on the null stream of the calling process, copies of a fixed number of bytes done three ways:
//   kind 0  "fat"   256-lane workgroups that ALLOCATE 280 registers (256 VGPRs + 24 AGPRs) and 19.7 KB of LDS -- the
//                   resource shape of rcclGenericKernel in this image's librccl.so (gfx950 code object)
//   kind 1  "slim"  the same copy loop in 64 registers, no LDS
//   kind 2  hipMemcpyAsync device-to-device (what a peer push over xGMI is issued as; on one GPU the runtime does it with a
//           small copy kernel)
//   kind 3  hipMemcpyAsync device-to-PINNED-HOST on the null stream, in `workgroups` equal parts one after the other
//   kind 4  the same parts on `workgroups` streams of their own (copy-only streams: the SDMA engines work side by side, as
//           the pushes to seven peers must -- do such streams disturb the job lanes, whose hardware queues they share?)
// tools/coresidency_probe.py runs them while pc_analyzer's job lanes are busy and reports how far each falls behind.
//
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/coresidency.hip -o tools/bin/libcoresidency.so
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

namespace {
uint4 *g_src = nullptr, *g_dst = nullptr;
void* g_host = nullptr;
std::vector<hipStream_t> g_streams;
std::vector<hipEvent_t> g_join;
size_t g_bytes = 0;
std::vector<hipEvent_t> g_ev;
size_t g_used = 0;

__device__ __forceinline__ void copy_loop(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void fat_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    __shared__ uint32_t pad[19744 / 4];
    if (n16 == ~size_t(0)) pad[threadIdx.x] = 1;          // keeps the allocation
    asm volatile("" ::: "v255", "a23");                    // 256 VGPRs + 24 AGPRs = 280 registers per lane
    copy_loop(src, dst, n16);
    if (n16 == ~size_t(0)) dst[0].x = pad[0];
}

__global__ __launch_bounds__(256) void slim_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    copy_loop(src, dst, n16);
}
}  // namespace

extern "C" {

int cr_init(size_t bytes, int max_launches) {
    g_bytes = bytes & ~size_t(15);
    if (hipMalloc(&g_src, g_bytes) != hipSuccess || hipMalloc(&g_dst, g_bytes) != hipSuccess) return 1;
    (void)hipMemset(g_src, 1, g_bytes);
    if (hipHostMalloc(&g_host, g_bytes, hipHostMallocDefault) != hipSuccess) return 4;
    g_streams.resize(8);
    g_join.resize(8);
    for (size_t i = 0; i < g_streams.size(); i++) {
        if (hipStreamCreateWithFlags(&g_streams[i], hipStreamNonBlocking) != hipSuccess) return 5;
        if (hipEventCreateWithFlags(&g_join[i], hipEventDisableTiming) != hipSuccess) return 6;
    }
    g_ev.resize((size_t)max_launches * 2);
    for (hipEvent_t& e : g_ev)
        if (hipEventCreate(&e) != hipSuccess) return 2;
    g_used = 0;
    return hipDeviceSynchronize() == hipSuccess ? 0 : 3;
}

// one copy of the buffer on the null stream, bracketed by events
int cr_launch(int kind, int workgroups) {
    if (g_used + 2 > g_ev.size()) return 1;
    (void)hipEventRecord(g_ev[g_used], nullptr);
    if (kind == 0)
        hipLaunchKernelGGL(fat_copy_kernel, dim3(workgroups), dim3(256), 0, nullptr, g_src, g_dst, g_bytes / 16);
    else if (kind == 1)
        hipLaunchKernelGGL(slim_copy_kernel, dim3(workgroups), dim3(256), 0, nullptr, g_src, g_dst, g_bytes / 16);
    else if (kind == 2)
        (void)hipMemcpyAsync(g_dst, g_src, g_bytes, hipMemcpyDeviceToDevice, nullptr);
    else {
        const int parts = workgroups < 1 ? 1 : (workgroups > 8 ? 8 : workgroups);
        const size_t per = (g_bytes / parts) & ~size_t(15);
        for (int p = 0; p < parts; p++) {
            char* const d = static_cast<char*>(g_host) + (size_t)p * per;
            const char* const s = reinterpret_cast<const char*>(g_src) + (size_t)p * per;
            if (kind == 3) {
                (void)hipMemcpyAsync(d, s, per, hipMemcpyDeviceToHost, nullptr);
            } else {   // independent streams, no dependency on anything: cr_collect waits for them (the events on the null
                       // stream say nothing about these copies; the probe reports the backlog and the step time)
                (void)hipMemcpyAsync(d, s, per, hipMemcpyDeviceToHost, g_streams[p]);
            }
        }
    }
    (void)hipEventRecord(g_ev[g_used + 1], nullptr);
    g_used += 2;
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

// 1 if everything launched so far has finished
int cr_idle() {
    for (hipStream_t st : g_streams)
        if (hipStreamQuery(st) != hipSuccess) return 0;
    return hipStreamQuery(nullptr) == hipSuccess ? 1 : 0;
}

// waits for the null stream; mean / max duration of the launches since the last call (ms, event to event)
int cr_collect(double* mean_ms, double* max_ms, int* n) {
    if (hipStreamSynchronize(nullptr) != hipSuccess) return 1;
    for (hipStream_t st : g_streams)
        if (hipStreamSynchronize(st) != hipSuccess) return 1;
    double sum = 0, mx = 0;
    for (size_t i = 0; i + 1 < g_used; i += 2) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, g_ev[i], g_ev[i + 1]);
        sum += ms;
        if (ms > mx) mx = ms;
    }
    *n = (int)(g_used / 2);
    *mean_ms = *n ? sum / *n : 0;
    *max_ms = mx;
    g_used = 0;
    return 0;
}
}
