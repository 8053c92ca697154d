// ta_probe.hip -- what a gather of image-region rows costs in the CU's texture addresser (TA) / L1 (TCP).
//
// The LK staging reads, per wavefront, 16 regions (4-lane groups, each in another image) of 13 rows x 26 bytes at 2-byte-aligned
// origins.  This probe issues that access pattern in isolation -- VEC dwords per lane, LPR lanes per region row, origin alignment
// ALIGN -- on every CU at the LK kernel's occupancy (12 wavefronts per CU), from L2-resident images, and reports the time per
// load instruction; run under `rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD TA_BUSY_avr` it gives the L1 accesses
// per instruction (tools/ta_probe.py).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int IMG_W = 512, IMG_H = 512, N_IMG = 8;   // 8 x 512 KB of uint16: L2-resident (4 MB per XCD)
constexpr int LOADS = 8;                              // load instructions per region "staging"

template <int VEC> struct __attribute__((packed, aligned(2))) Raw { uint32_t d[VEC]; };

// ALIGN: 0 = origin column a multiple of 8 pixels (16-byte aligned), 1 = even (4-byte aligned), 2 = any (2-byte aligned), 3 = odd
template <int VEC, int LPR, int ALIGN>
__global__ __launch_bounds__(64) void probe(const uint16_t* __restrict__ imgs, int iters, uint32_t* __restrict__ sink) {
    const int lane = threadIdx.x, grp = lane >> 2, lg = lane & 3;
    const uint16_t* img = imgs + (size_t)(grp & 7) * IMG_W * IMG_H;
    constexpr int ROWS_PER_LOAD = 4 / LPR;
    const int sub = lg / LPR, part = lg % LPR;     // the lane's row within a load, its piece of the row
    uint32_t acc = 0;
    uint32_t h = blockIdx.x * 2654435761u + grp * 40503u;
    for (int it = 0; it < iters; it++) {
        h = h * 1664525u + 1013904223u;
        int ox = (h >> 8) % (IMG_W - 64), oy = (h >> 20) % (IMG_H - 64);
        if (ALIGN == 0) ox &= ~7;
        if (ALIGN == 1) ox &= ~1;
        if (ALIGN == 3) ox |= 1;
        const uint16_t* src = img + (oy + sub) * IMG_W + ox + part * VEC * 2;
        Raw<VEC> v[LOADS];
#pragma unroll
        for (int k = 0; k < LOADS; k++) {
            v[k] = *reinterpret_cast<const Raw<VEC>*>(src);
            src += ROWS_PER_LOAD * IMG_W;
        }
#pragma unroll
        for (int k = 0; k < LOADS; k++)
#pragma unroll
            for (int j = 0; j < VEC; j++) acc ^= v[k].d[j];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int VEC, int LPR, int ALIGN>
static void run(const uint16_t* imgs, uint32_t* sink, const char* filter) {
    char name[64];
    std::snprintf(name, sizeof name, "vec%d_lpr%d_align%d", VEC, LPR, ALIGN);
    if (filter && !std::strstr(name, filter)) return;
    const int iters = 400, blocks = 256 * 12;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((probe<VEC, LPR, ALIGN>), dim3(blocks), dim3(64), 0, 0, imgs, 20, sink);   // warm the caches
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((probe<VEC, LPR, ALIGN>), dim3(blocks), dim3(64), 0, 0, imgs, iters, sink);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    // per CU: 12 wavefronts x iters x LOADS load instructions
    const double instr_per_cu = 12.0 * iters * LOADS, cycles = ms * 1e-3 * 2.4e9;
    std::printf("{\"pattern\": \"%s\", \"dwords_per_lane\": %d, \"lanes_per_row\": %d, \"align\": %d, \"ms\": %.4f, \"cycles_per_load_instruction_and_cu\": %.1f, "
                "\"cycles_per_region_row_and_cu\": %.2f}\n",
                name, VEC, LPR, ALIGN, ms, cycles / instr_per_cu, cycles / (instr_per_cu * 16 * (4 / LPR)));
}

int main(int argc, char** argv) {
    const char* filter = argc > 1 ? argv[1] : nullptr;
    const size_t n = (size_t)N_IMG * IMG_W * IMG_H + 4096;
    std::vector<uint16_t> host(n);
    for (size_t i = 0; i < n; i++) host[i] = (uint16_t)(i * 2654435761u >> 13);
    uint16_t* imgs;
    uint32_t* sink;
    CHECK(hipMalloc(&imgs, n * 2));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemcpy(imgs, host.data(), n * 2, hipMemcpyHostToDevice));
#define ALIGNS(V, L) run<V, L, 0>(imgs, sink, filter); run<V, L, 1>(imgs, sink, filter); run<V, L, 2>(imgs, sink, filter); run<V, L, 3>(imgs, sink, filter);
    ALIGNS(4, 1)   // stage_region: a lane = a whole row (dwordx4 of the dwordx4 + dwordx3 pair)
    ALIGNS(3, 1)
    ALIGNS(4, 2)   // stage_region_paired
    ALIGNS(2, 4)   // four lanes per row, dwordx2
    ALIGNS(2, 2)
    ALIGNS(1, 4)
    ALIGNS(2, 1)
    ALIGNS(1, 1)
    return 0;
}
