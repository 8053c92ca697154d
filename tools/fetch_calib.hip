// fetch_calib.hip -- known-byte kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (run ON the GPU box
// under tools/fetch_calib.py).  MI355X_MICROARCH.md (HBM section): FETCH_SIZE reports exactly half of a wide coalesced
// read; "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern".
//   stream_read16   every lane reads 16 B, consecutive lanes consecutive addresses            (the guide's case)
//   rows28          the LK kernel's staging access: a lane reads ONE region row = 28 B (7 dwords) at a 2-byte aligned
//                   address; the 4 lanes of a group read 4 consecutive rows of a 13-row region at a random place of a
//                   plane with the pitch of a 1080p uint16 plane (kernels_lk3.hip: stage_region, RowRegs<3>)
//   stream_write16  every lane writes 16 B, consecutive lanes consecutive addresses            (LK's record stores)
// The buffers are 2 GiB (8x the Infinity Cache) and every byte / line is touched once.  The program prints, per kernel,
// the bytes the lanes asked for and the distinct 64-B and 128-B lines those bytes lie in.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void stream_read16(const uint4* __restrict__ src, size_t n, uint32_t* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

struct __attribute__((packed, aligned(2))) Row28 { uint32_t d[7]; };
// one wavefront = 16 groups of 4 lanes; group g of wavefront w stages region (w * 16 + g): rows lg, lg + 4, lg + 8, lg + 12
__global__ void rows28(const uint16_t* __restrict__ plane, const uint32_t* __restrict__ origins, int pitch, int n_regions, uint32_t* sink) {
    const int lane = threadIdx.x & 63, lg = lane & 3;
    const int region = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 16 + (lane >> 2);
    if (region >= n_regions) return;
    const uint16_t* base = plane + origins[region];
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int r = min(lg + 4 * k, 12);
        const Row28 v = *reinterpret_cast<const Row28*>(base + (size_t)r * pitch);
#pragma unroll
        for (int j = 0; j < 7; j++) acc ^= v.d[j];
    }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ void stream_write16(uint4* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    uint8_t* buf;
    uint32_t* sink;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 1, bytes));
    // rows28: regions of 13 rows x 14 pixels (28 B) in a plane of pitch 1984 uint16 (align64(16 + 1920 + 10)), one region per
    // 16 rows x 64 columns cell so that no two regions share a line
    const int pitch = 1984;
    const size_t plane_px = bytes / 2;
    const int rows_total = (int)(plane_px / pitch);
    std::vector<uint32_t> origins;
    uint32_t rng = 12345u;
    for (int cy = 0; cy + 16 <= rows_total; cy += 16)
        for (int cx = 0; cx + 128 <= pitch; cx += 128) {   // 128 px = 256 B cells
            rng = rng * 1664525u + 1013904223u;
            const int ox = (int)((rng >> 8) % 100u), oy = (int)((rng >> 20) % 3u);   // pixel offsets: 2-byte aligned addresses
            origins.push_back((uint32_t)((size_t)(cy + oy) * pitch + cx + ox));
        }
    const int n_regions = (int)origins.size();
    std::set<uint64_t> l64, l128;
    size_t asked = 0;
    for (uint32_t o : origins)
        for (int r = 0; r < 13; r++) {
            const uint64_t a = ((uint64_t)o + (uint64_t)r * pitch) * 2, b = a + 27;
            asked += 28;
            for (uint64_t x = a / 64; x <= b / 64; x++) l64.insert(x);
            for (uint64_t x = a / 128; x <= b / 128; x++) l128.insert(x);
        }
    uint32_t* d_orig;
    CHECK(hipMalloc(&d_orig, origins.size() * 4));
    CHECK(hipMemcpy(d_orig, origins.data(), origins.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(stream_read16, dim3(256 * 16), dim3(256), 0, 0, reinterpret_cast<const uint4*>(buf), bytes / 16, sink);
        hipLaunchKernelGGL(rows28, dim3((n_regions / 16 + 3) / 4 + 1), dim3(256), 0, 0, reinterpret_cast<const uint16_t*>(buf), d_orig, pitch, n_regions, sink);
        hipLaunchKernelGGL(stream_write16, dim3(256 * 16), dim3(256), 0, 0, reinterpret_cast<uint4*>(buf), bytes / 16);
        CHECK(hipDeviceSynchronize());
    }
    printf("{\"stream_read16\": {\"bytes\": %zu}, \"rows28\": {\"bytes_asked\": %zu, \"regions\": %d, \"lines64_bytes\": %zu, \"lines128_bytes\": %zu}, "
           "\"stream_write16\": {\"bytes\": %zu}, \"launches_each\": 3}\n",
           bytes, asked, n_regions, l64.size() * 64, l128.size() * 128, bytes);
    return 0;
}
