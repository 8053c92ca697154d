// pinned_alloc_probe.hip -- what page-locked staging memory costs on the GPU box (run there):
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/pinned_alloc_probe tools/probes/pinned_alloc_probe.hip && /tmp/pinned_alloc_probe
// hipHostMalloc / hipHostFree by size, alone and from 8 threads at once; hipMalloc of 1.2 GB; a pageable and a pinned
// host-to-device copy of 256 MB.  DESIGN.md section 5 ("Refine Sequence": the segment loader) quotes the numbers.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    (void)hipSetDevice(0);
    void* warm = nullptr;
    (void)hipMalloc(&warm, 1 << 20);
    for (size_t mb : {1, 2, 4, 16, 64, 256}) {
        double a = 0, f = 0;
        const int reps = mb >= 64 ? 2 : 8;
        for (int r = 0; r < reps; r++) {
            void* p = nullptr;
            double t0 = now_ms();
            if (hipHostMalloc(&p, mb << 20, hipHostMallocDefault) != hipSuccess) return 1;
            double t1 = now_ms();
            (void)hipHostFree(p);
            double t2 = now_ms();
            a += t1 - t0, f += t2 - t1;
        }
        printf("hipHostMalloc %4zu MB: %.2f ms, hipHostFree %.2f ms\n", mb, a / reps, f / reps);
    }
    {
        double t0 = now_ms();
        std::vector<std::thread> th;
        for (int k = 0; k < 8; k++)
            th.emplace_back([] {
                (void)hipSetDevice(0);
                void* p[6];
                for (auto& q : p) (void)hipHostMalloc(&q, 2 << 20, hipHostMallocDefault);
                for (auto& q : p) (void)hipHostFree(q);
            });
        for (auto& t : th) t.join();
        printf("8 threads x 6 x (hipHostMalloc 2 MB + free): %.2f ms wall\n", now_ms() - t0);
    }
    {
        void* d = nullptr;
        double t0 = now_ms();
        (void)hipMalloc(&d, size_t{1200} << 20);
        double t1 = now_ms();
        (void)hipFree(d);
        printf("hipMalloc 1.2 GB: %.2f ms, hipFree %.2f ms\n", t1 - t0, now_ms() - t1);
    }
    {
        const size_t n = size_t{256} << 20;
        void *d = nullptr, *pin = nullptr;
        (void)hipMalloc(&d, n);
        char* page = static_cast<char*>(malloc(n));
        memset(page, 1, n);
        (void)hipHostMalloc(&pin, n, hipHostMallocDefault);
        memset(pin, 1, n);
        for (int r = 0; r < 2; r++) {
            double t0 = now_ms();
            (void)hipMemcpy(d, page, n, hipMemcpyHostToDevice);
            double t1 = now_ms();
            (void)hipMemcpy(d, pin, n, hipMemcpyHostToDevice);
            double t2 = now_ms();
            printf("256 MB host to device: pageable %.2f ms (%.1f GB/s), pinned %.2f ms (%.1f GB/s)\n", t1 - t0, n / (t1 - t0) / 1e6, t2 - t1,
                   n / (t2 - t1) / 1e6);
        }
        double t0 = now_ms();
        free(page);
        printf("free(256 MB touched): %.2f ms\n", now_ms() - t0);
    }
    FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
    if (f) {
        char line[128] = {0};
        if (fgets(line, sizeof line, f)) printf("transparent_hugepage: %s", line);
        fclose(f);
    }
    return 0;
}
