// pageable_upload_probe.hip -- how to get 1.2 GB that reader threads have just written into anonymous memory onto the GPU and
// the memory back to the kernel (run ON the GPU box):
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/pageable_upload_probe tools/probes/pageable_upload_probe.hip && /tmp/pageable_upload_probe
// Variants: 4-KiB pages or MADV_HUGEPAGE; plain hipMemcpyAsync of the pageable pieces, hipHostRegister around them, or a
// hand-made staging pipeline (threads copy into page-locked blocks, the copy engine takes those).  Per variant: fill (first
// touch by 8 threads), AnonHugePages of the process, copy, release (munmap).  DESIGN.md section 5 quotes the outcome.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static long anon_huge_kb() {
    FILE* f = fopen("/proc/self/smaps_rollup", "r");
    if (!f) return -1;
    char line[256];
    long kb = -1;
    while (fgets(line, sizeof line, f))
        if (sscanf(line, "AnonHugePages: %ld kB", &kb) == 1) break;
    fclose(f);
    return kb;
}

int main(int argc, char** argv) {
    (void)hipSetDevice(0);
    const int kParts = 8;
    const size_t part_bytes = size_t{150} << 20;   // 8 x 150 MB
    char* dev = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&dev), kParts * part_bytes) != hipSuccess) return 1;
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    {   // the first copy of a process pays one-time set-up: keep it out of the numbers
        std::vector<char> warm(64 << 20, 1);
        (void)hipMemcpyAsync(dev, warm.data(), warm.size(), hipMemcpyHostToDevice, s);
        (void)hipStreamSynchronize(s);
    }
    const char* names[] = {"4 KiB pages, plain copy", "huge pages, plain copy", "huge pages, register around the copy",
                           "huge pages, own staging x4 threads", "4 KiB pages, own staging x4 threads", "huge pages, plain copy (again)"};
    const int huge[] = {0, 1, 1, 1, 0, 1}, how[] = {0, 0, 1, 2, 2, 0};
    for (int v = 0; v < 6; v++) {
        char* part[kParts];
        double t0 = now_ms();
        for (int k = 0; k < kParts; k++) {
            part[k] = static_cast<char*>(mmap(nullptr, part_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
            if (huge[v]) (void)madvise(part[k], part_bytes, MADV_HUGEPAGE);
        }
        {
            std::vector<std::thread> th;
            for (int k = 0; k < kParts; k++) th.emplace_back([&, k] { memset(part[k], k + 1, part_bytes); });
            for (auto& t : th) t.join();
        }
        double t1 = now_ms();
        const long hp = anon_huge_kb();
        if (how[v] == 0 || how[v] == 1) {
            if (how[v] == 1)
                for (int k = 0; k < kParts; k++) (void)hipHostRegister(part[k], part_bytes, hipHostRegisterDefault);
            for (int k = 0; k < kParts; k++) (void)hipMemcpyAsync(dev + k * part_bytes, part[k], part_bytes, hipMemcpyHostToDevice, s);
            (void)hipStreamSynchronize(s);
            if (how[v] == 1)
                for (int k = 0; k < kParts; k++) (void)hipHostUnregister(part[k]);
        } else {
            // 4 threads, each with two page-locked 8-MiB blocks and a stream: memcpy into a block, async copy of the block
            const int kThreads = 4;
            const size_t block = size_t{8} << 20;
            char* pinned = nullptr;
            (void)hipHostMalloc(reinterpret_cast<void**>(&pinned), kThreads * 2 * block, hipHostMallocDefault);
            std::vector<std::thread> th;
            for (int t = 0; t < kThreads; t++)
                th.emplace_back([&, t] {
                    (void)hipSetDevice(0);
                    hipStream_t ts;
                    (void)hipStreamCreateWithFlags(&ts, hipStreamNonBlocking);
                    hipEvent_t done[2];
                    for (auto& e : done) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
                    int cur = 0, used[2] = {0, 0};
                    for (int k = t * kParts / kThreads; k < (t + 1) * kParts / kThreads; k++)
                        for (size_t at = 0; at < part_bytes; at += block) {
                            const size_t n = std::min(block, part_bytes - at);
                            char* stage = pinned + (t * 2 + cur) * block;
                            if (used[cur]) (void)hipEventSynchronize(done[cur]);
                            memcpy(stage, part[k] + at, n);
                            (void)hipMemcpyAsync(dev + k * part_bytes + at, stage, n, hipMemcpyHostToDevice, ts);
                            (void)hipEventRecord(done[cur], ts);
                            used[cur] = 1;
                            cur ^= 1;
                        }
                    (void)hipStreamSynchronize(ts);
                    for (auto& e : done) (void)hipEventDestroy(e);
                    (void)hipStreamDestroy(ts);
                });
            for (auto& t : th) t.join();
            (void)hipHostFree(pinned);
        }
        double t2 = now_ms();
        for (int k = 0; k < kParts; k++) munmap(part[k], part_bytes);
        double t3 = now_ms();
        unsigned char probe = 0;
        (void)hipMemcpy(&probe, dev + 3 * part_bytes + 12345, 1, hipMemcpyDeviceToHost);
        printf("%-40s fill %6.1f ms  AnonHugePages %7ld kB  copy %6.1f ms (%5.1f GB/s)  release %6.1f ms  [byte %d]\n", names[v], t1 - t0, hp,
               t2 - t1, kParts * part_bytes / (t2 - t1) / 1e6, t3 - t2, (int)probe);
    }
    return 0;
}
