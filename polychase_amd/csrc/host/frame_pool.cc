#include "frame_pool.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

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

namespace {

// A handful of copy workers, started on first use and never joined (like the pool: process lifetime).  Every copy is
// a Job object of its own; a worker that is still looking for a piece of the previous job keeps looking THERE.
class CopyCrew {
public:
    static CopyCrew& Get() {
        static CopyCrew* crew = new CopyCrew();
        return *crew;
    }
    void Copy(uint8_t* dst, const uint8_t* src, size_t bytes) {
        constexpr size_t kPiece = size_t(1) << 20;
        if (n_workers_ == 0 || bytes < 4 * kPiece) {
            std::memcpy(dst, src, bytes);
            return;
        }
        auto job = std::make_shared<Job>();
        job->dst = dst;
        job->src = src;
        job->bytes = bytes;
        job->piece = std::max(kPiece, (bytes / (size_t)(n_workers_ + 1) + 4095) & ~size_t(4095));
        job->pending = (bytes + job->piece - 1) / job->piece;
        {
            std::lock_guard<std::mutex> lk(mtx_);
            current_ = job;
            generation_++;
        }
        cv_.notify_all();
        Work(*job);   // the caller copies pieces too
        std::unique_lock<std::mutex> lk(job->mtx);
        job->done.wait(lk, [&] { return job->pending == 0; });
    }

private:
    struct Job {
        uint8_t* dst = nullptr;
        const uint8_t* src = nullptr;
        size_t bytes = 0, piece = 1;
        std::atomic<size_t> next{0};
        std::mutex mtx;
        std::condition_variable done;
        size_t pending = 0;
    };
    CopyCrew() {
        unsigned hw = std::thread::hardware_concurrency();
        int want = hw > 2 ? (int)std::min(6u, hw - 1) : 0;
        if (const char* e = std::getenv("POLYCHASE_COPY_THREADS")) want = std::max(0, std::min(32, std::atoi(e) - 1));
        n_workers_ = want;
        for (int i = 0; i < want; i++) std::thread([this] { Loop(); }).detach();
    }
    static void Work(Job& j) {
        for (;;) {
            const size_t off = j.next.fetch_add(j.piece, std::memory_order_relaxed);
            if (off >= j.bytes) return;
            std::memcpy(j.dst + off, j.src + off, std::min(j.piece, j.bytes - off));
            std::lock_guard<std::mutex> lk(j.mtx);
            if (--j.pending == 0) j.done.notify_all();
        }
    }
    void Loop() {
        unsigned long long seen = 0;
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> lk(mtx_);
                cv_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
                job = current_;
            }
            if (job) Work(*job);
        }
    }
    int n_workers_ = 0;
    std::mutex mtx_;
    std::condition_variable cv_;
    std::shared_ptr<Job> current_;
    unsigned long long generation_ = 0;
};

}  // namespace

void CopyFrameBytes(void* dst, const void* src, size_t bytes) {
    CopyCrew::Get().Copy(static_cast<uint8_t*>(dst), static_cast<const uint8_t*>(src), bytes);
}
