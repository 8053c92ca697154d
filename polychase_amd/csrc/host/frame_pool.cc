#include "frame_pool.h"

#include <deque>
#include <mutex>
#include <stdexcept>
#include <string>

#include "../../../include/polychase_hip.h"

namespace {

// A buffer is released by its owner only after the GPU has consumed it (analysis_driver.cc keeps the owner until
// pc_analyzer_frame_ingested), so reuse is safe at once; a few idle buffers are kept so that acquire rarely allocates
constexpr size_t kMinIdle = 2;   // released buffers that queue up before the oldest one is reused
constexpr size_t kMaxIdle = 24;  // beyond this the oldest idle buffers are freed

struct Pool {
    std::mutex mtx;
    size_t bytes = 0;                 // all pooled buffers have this size (frames of one clip)
    std::deque<void*> idle;           // oldest first
};

Pool& ThePool() {
    static Pool* pool = new Pool();   // never destroyed: at process exit the HIP runtime may already be gone
    return *pool;
}

void Release(void* buffer, size_t bytes) {
    Pool& p = ThePool();
    std::lock_guard<std::mutex> lk(p.mtx);
    if (bytes != p.bytes) {           // the clip size changed meanwhile
        pc_host_buffer_free(buffer);
        return;
    }
    p.idle.push_back(buffer);
    while (p.idle.size() > kMaxIdle) {
        pc_host_buffer_free(p.idle.front());
        p.idle.pop_front();
    }
}

}  // namespace

std::shared_ptr<void> AcquirePinnedFrameBuffer(size_t bytes) {
    Pool& p = ThePool();
    void* buffer = nullptr;
    {
        std::lock_guard<std::mutex> lk(p.mtx);
        if (bytes != p.bytes) {       // new frame size: start over
            for (void* b : p.idle) pc_host_buffer_free(b);
            p.idle.clear();
            p.bytes = bytes;
        }
        if (p.idle.size() > kMinIdle) {
            buffer = p.idle.front();
            p.idle.pop_front();
        }
    }
    if (!buffer && pc_host_buffer_alloc(bytes, &buffer) != PC_OK)
        throw std::runtime_error(std::string("pc_host_buffer_alloc: ") + pc_last_error());
    return std::shared_ptr<void>(buffer, [bytes](void* b) { Release(b, bytes); });
}
