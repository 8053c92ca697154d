// multi_gpu.cc -- see multi_gpu.h.  The single-process loop this shards is the reference's cpp/opticalflow.cc:209-321; the
// store it feeds is cpp/opticalflow.cc:149-151 / cpp/database.cc:183-214.
#include "multi_gpu.h"

#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <exception>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../../include/polychase_hip.h"
#include "gpu_context.h"
#include "utils.h"

namespace {

using Clock = std::chrono::steady_clock;
double Since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

[[noreturn]] void Fail(const std::string& what) { throw std::runtime_error("multi-GPU analysis: " + what); }
void CheckAbi(int rc, const char* what) {
    if (rc != PC_OK) Fail(std::string(what) + ": " + pc_last_error());
}

// ---- blocking TCP with whole-message semantics -------------------------------------------------------------------
class Socket {
   public:
    Socket() = default;
    explicit Socket(int fd) : fd_(fd) {
        const int one = 1;
        setsockopt(fd_, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
    }
    Socket(Socket&& o) noexcept : fd_(o.fd_) { o.fd_ = -1; }
    Socket& operator=(Socket&& o) noexcept {
        Close();
        fd_ = o.fd_;
        o.fd_ = -1;
        return *this;
    }
    ~Socket() { Close(); }
    void Close() {
        if (fd_ >= 0) ::close(fd_);
        fd_ = -1;
    }
    bool Ok() const { return fd_ >= 0; }
    void Send(const void* p, size_t n) const {
        const char* c = static_cast<const char*>(p);
        while (n > 0) {
            const ssize_t k = ::send(fd_, c, n, MSG_NOSIGNAL);
            if (k < 0 && errno == EINTR) continue;
            if (k <= 0) Fail("control connection lost while sending");
            c += k;
            n -= static_cast<size_t>(k);
        }
    }
    void Recv(void* p, size_t n) const {
        char* c = static_cast<char*>(p);
        while (n > 0) {
            const ssize_t k = ::recv(fd_, c, n, 0);
            if (k < 0 && errno == EINTR) continue;
            if (k < 0 && (errno == EAGAIN || errno == EWOULDBLOCK)) Fail("control connection: a receive timed out");   // (no socket of this file has a timeout)
            if (k <= 0) Fail("control connection lost while receiving (the peer ended)");
            c += k;
            n -= static_cast<size_t>(k);
        }
    }
    template <typename T>
    void SendValue(const T& v) const { Send(&v, sizeof(T)); }
    template <typename T>
    T RecvValue() const {
        T v;
        Recv(&v, sizeof(T));
        return v;
    }

   private:
    int fd_ = -1;
};

sockaddr_in Address(const std::string& host, int port) {
    sockaddr_in a{};
    a.sin_family = AF_INET;
    a.sin_port = htons(static_cast<uint16_t>(port));
    if (inet_pton(AF_INET, host.c_str(), &a.sin_addr) != 1) Fail("master_addr must be a dotted IPv4 address, got " + host);
    return a;
}

Socket ConnectTo(const std::string& host, int port, double timeout_s) {
    const auto t0 = Clock::now();
    const sockaddr_in a = Address(host, port);
    for (;;) {
        const int fd = ::socket(AF_INET, SOCK_STREAM, 0);
        if (fd < 0) Fail("socket() failed");
        if (::connect(fd, reinterpret_cast<const sockaddr*>(&a), sizeof(a)) == 0) return Socket(fd);
        ::close(fd);
        if (Since(t0) > timeout_s) Fail("could not reach rank 0 at " + host + ":" + std::to_string(port));
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
}

// rank 0: connections of ranks 1 .. world - 1, indexed by rank
std::vector<Socket> AcceptRanks(const std::string& host, int port, int world, double timeout_s) {
    const int lfd = ::socket(AF_INET, SOCK_STREAM, 0);
    if (lfd < 0) Fail("socket() failed");
    const int one = 1;
    setsockopt(lfd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    const sockaddr_in a = Address(host, port);
    if (::bind(lfd, reinterpret_cast<const sockaddr*>(&a), sizeof(a)) != 0 || ::listen(lfd, world) != 0) {
        ::close(lfd);
        Fail("rank 0 cannot listen on " + host + ":" + std::to_string(port));
    }
    // The deadline applies to the RENDEZVOUS only: poll() on the listening socket and on a fresh connection's rank
    // announcement.  (Round 4 set SO_RCVTIMEO on the listening socket, which Linux hands down to every accepted socket: all
    // of rank 0's later receives -- headers, payloads, the final flag -- then failed after connect_timeout_s of silence, i.e.
    // whenever a rank needed longer than that between two pieces.)  The accepted sockets block without a limit: a peer that
    // dies closes its end and the receive returns 0.
    const auto t0 = Clock::now();
    auto wait_readable = [&](int fd) {
        for (;;) {
            const double left = timeout_s - Since(t0);
            if (left <= 0) return false;
            pollfd p{};
            p.fd = fd;
            p.events = POLLIN;
            const int rc = ::poll(&p, 1, static_cast<int>(std::min(left * 1000.0 + 1.0, 1000.0)));
            if (rc > 0) return true;
            if (rc < 0 && errno != EINTR) return false;
        }
    };
    std::vector<Socket> out(static_cast<size_t>(world));
    for (int k = 1; k < world; k++) {
        const int fd = wait_readable(lfd) ? ::accept(lfd, nullptr, nullptr) : -1;
        if (fd < 0) {
            ::close(lfd);
            Fail("rank 0 waited for " + std::to_string(world - 1) + " ranks, " + std::to_string(k - 1) + " connected");
        }
        Socket s(fd);
        if (!wait_readable(fd)) {
            ::close(lfd);
            Fail("a connection did not announce its rank");
        }
        const int32_t r = s.RecvValue<int32_t>();
        if (r < 1 || r >= world || out[static_cast<size_t>(r)].Ok()) {
            ::close(lfd);
            Fail("a connection announced rank " + std::to_string(r));
        }
        out[static_cast<size_t>(r)] = std::move(s);
    }
    ::close(lfd);
    return out;
}

// ---- a bounded queue between the driver's thread and the transfer thread ---------------------------------------------
template <typename T>
class BoundedQueue {
   public:
    explicit BoundedQueue(size_t depth) : depth_(depth) {}
    // false: the queue was closed by the other side (it failed)
    bool Push(T v, double* seconds_blocked = nullptr) {
        std::unique_lock<std::mutex> lk(m_);
        const auto t0 = Clock::now();
        cv_.wait(lk, [&] { return q_.size() < depth_ || closed_; });
        if (seconds_blocked) *seconds_blocked += Since(t0);
        if (closed_) return false;
        q_.push_back(std::move(v));
        cv_.notify_all();
        return true;
    }
    bool Pop(T* v, double* seconds_blocked = nullptr) {
        std::unique_lock<std::mutex> lk(m_);
        const auto t0 = Clock::now();
        cv_.wait(lk, [&] { return !q_.empty() || closed_; });
        if (seconds_blocked) *seconds_blocked += Since(t0);
        if (q_.empty()) return false;
        *v = std::move(q_.front());
        q_.pop_front();
        cv_.notify_all();
        return true;
    }
    void Close() {
        std::lock_guard<std::mutex> lk(m_);
        closed_ = true;
        cv_.notify_all();
    }

   private:
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<T> q_;
    size_t depth_;
    bool closed_ = false;
};

struct DeviceBuffer {
    int device = 0;
    void* p = nullptr;
    size_t bytes = 0;
    DeviceBuffer() = default;
    DeviceBuffer(int dev, size_t n) : device(dev), bytes(n) { CheckAbi(pc_peer_buffer_alloc(dev, n, &p), "device buffer"); }
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    ~DeviceBuffer() {
        if (p) pc_peer_buffer_free(device, p);
    }
};

constexpr int64_t kEnd = 0, kFailed = -1;
struct Header {
    int64_t bytes, frames, first;
};

// a piece on its way to rank 0: staged on the device (rccl) or on the host (tcp)
struct Outgoing {
    int slot = -1;      // staging slot (rccl)
    std::vector<uint8_t> host;
    Header h{kEnd, 0, 0};
};
struct Incoming {
    int rank = 0;
    Header h{};
    std::vector<uint8_t> host;
    std::exception_ptr error;
    bool end = false;
};

size_t LogPartBytes(const VideoInfo& vi, const MultiGpuConfig& cfg) {
    // the analyzer's record: header + keypoints (8 B) + 8 flows x 16 B per keypoint (polychase_amd/distributed.py: log_capacity_bytes)
    const size_t kp = cfg.keypoints_per_frame ? cfg.keypoints_per_frame : static_cast<size_t>(vi.width) * vi.height / 40 + 4096;
    const size_t per = 256 + kp * 8 + kp * 8 * 16 + 64;
    return ((static_cast<size_t>(cfg.piece_frames) + 1) * per + 15) / 16 * 16;
}

}  // namespace

MultiGpuResult GenerateOpticalFlowDatabaseMultiGpu(const VideoInfo& video_info, FrameAccessorFunction frame_accessor,
                                                   OpticalFlowProgressCallback callback, const std::string& database_path,
                                                   const MultiGpuConfig& cfg, const GFTTOptions& detector_options,
                                                   const OpticalFlowOptions& flow_options) {
    const int world = cfg.world_size, rank = cfg.rank;
    if (world < 1 || rank < 0 || rank >= world) Fail("rank " + std::to_string(rank) + " of " + std::to_string(world));
    if (cfg.transport != "rccl" && cfg.transport != "tcp") Fail("transport must be rccl or tcp");
    if (cfg.piece_frames < 1) Fail("piece_frames must be >= 1");
    const bool rccl = cfg.transport == "rccl";
    const int device = cfg.device >= 0 ? cfg.device : rank;   // handed to the driver and to the communicator explicitly: no
                                                              // process-global state is touched (round 4 set POLYCHASE_DEVICE)
    const auto t0 = Clock::now();
    MultiGpuResult res;
    res.shard_begin = video_info.first_frame + static_cast<int32_t>(static_cast<int64_t>(video_info.num_frames) * rank / world);
    res.shard_end = video_info.first_frame + static_cast<int32_t>(static_cast<int64_t>(video_info.num_frames) * (rank + 1) / world);

    OpticalFlowShard shard;
    shard.begin = res.shard_begin;
    shard.end = res.shard_end;
    shard.device = device;
    if (world == 1) {
        GenerateOpticalFlowShard(video_info, frame_accessor, callback, database_path, shard, detector_options, flow_options, &res.stats);
        res.cancelled = shard.cancelled;
        res.seconds_analysis = res.seconds_total = Since(t0);
        return res;
    }

    // the communicator lives on a context of its own ON THIS RANK'S DEVICE: the staging / log buffers are allocated there, and
    // the process-wide shared context (csrc/host/gpu_context.h) may already exist on another device in a host that has used
    // AcceleratedMesh or track_sequence before
    pc_context* comm_ctx = nullptr;
    pc_comm* comm = nullptr;
    struct CommGuard {
        pc_comm*& c;
        pc_context*& ctx;
        ~CommGuard() {
            if (c) pc_comm_destroy(c);
            if (ctx) pc_context_destroy(ctx);
        }
    } comm_guard{comm, comm_ctx};
    if (rccl) CheckAbi(pc_context_create(device, &comm_ctx), "pc_context_create");

    if (rank == 0) {
        // ---- the owner of the database ----
        std::vector<Socket> peers = AcceptRanks(cfg.master_addr, cfg.master_port, world, cfg.connect_timeout_s);
        if (rccl) {
            unsigned char id[PC_COMM_ID_BYTES];
            CheckAbi(pc_comm_unique_id(id), "pc_comm_unique_id");
            for (int r = 1; r < world; r++) peers[static_cast<size_t>(r)].Send(id, sizeof(id));
            CheckAbi(pc_comm_create(comm_ctx, id, world, 0, &comm), "pc_comm_create");
        }
        BoundedQueue<Incoming> arrived(2);
        std::atomic<bool> abort{false};
        std::thread receiver([&] {
            int r = 1;
            try {
                std::unique_ptr<DeviceBuffer> dev;
                for (; r < world; r++) {
                    const Socket& s = peers[static_cast<size_t>(r)];
                    for (;;) {
                        if (abort.load()) Fail("rank 0's own shard failed");
                        s.SendValue<int64_t>(1);                         // credit: send your next piece
                        const Header h = s.RecvValue<Header>();
                        if (h.bytes == kFailed) Fail("rank " + std::to_string(r) + " failed: its records are incomplete");
                        if (h.bytes == kEnd) break;
                        Incoming in;
                        in.rank = r;
                        in.h = h;
                        in.host.resize(static_cast<size_t>(h.bytes));
                        if (rccl) {
                            if (!dev || dev->bytes < static_cast<size_t>(h.bytes))
                                dev.reset(new DeviceBuffer(device, static_cast<size_t>(h.bytes) + static_cast<size_t>(h.bytes) / 4 + 4096));
                            CheckAbi(pc_comm_recv(comm, dev->p, static_cast<uint64_t>(h.bytes), r), "pc_comm_recv");
                            CheckAbi(pc_peer_buffer_download(device, in.host.data(), dev->p, in.host.size()), "download of a received piece");
                        } else {
                            s.Recv(in.host.data(), in.host.size());
                        }
                        if (!arrived.Push(std::move(in))) Fail("rank 0 stopped storing");
                    }
                }
                Incoming done;
                done.end = true;
                arrived.Push(std::move(done));
            } catch (...) {
                // whoever has not been asked yet must not wait for a credit for ever
                for (int rr = r; rr < world; rr++) {
                    try {
                        peers[static_cast<size_t>(rr)].SendValue<int64_t>(-1);
                    } catch (...) {
                    }
                }
                Incoming bad;
                bad.error = std::current_exception();
                arrived.Push(std::move(bad));
            }
        });
        std::exception_ptr failure;
        try {
            if (cfg.synthetic_shard) {
                OpticalFlowRecordWriter own(database_path);
                cfg.synthetic_shard(0, [&](const void* bytes, size_t n, int32_t, int) {
                    OpticalFlowRunStats ws;
                    own.Write(static_cast<const uint8_t*>(bytes), n, &ws);
                    res.stats.keypoint_rows_written += ws.keypoint_rows_written;
                    res.stats.flow_rows_written += ws.flow_rows_written;
                });
                own.Close();
            } else {
                GenerateOpticalFlowShard(video_info, frame_accessor, callback, database_path, shard, detector_options, flow_options, &res.stats);
            }
            res.seconds_analysis = Since(t0);
            OpticalFlowRecordWriter writer(database_path);
            for (;;) {
                Incoming in;
                if (!arrived.Pop(&in)) Fail("the receiver ended without a word");
                if (in.error) std::rethrow_exception(in.error);
                if (in.end) break;
                OpticalFlowRunStats ws;
                writer.Write(in.host.data(), in.host.size(), &ws);
                res.stats.keypoint_rows_written += ws.keypoint_rows_written;
                res.stats.flow_rows_written += ws.flow_rows_written;
                res.stats.seconds_db += ws.seconds_db;
                res.pieces++;
                res.bytes_moved += in.host.size();
            }
            writer.Close();
        } catch (...) {
            failure = std::current_exception();
            abort.store(true);
            arrived.Close();
        }
        receiver.join();
        if (failure) std::rethrow_exception(failure);
        // a rank whose progress callback cancelled leaves a hole in the clip: every rank reports it
        int64_t cancelled = shard.cancelled ? 1 : 0;
        for (int r = 1; r < world; r++) cancelled |= peers[static_cast<size_t>(r)].RecvValue<int64_t>();
        for (int r = 1; r < world; r++) peers[static_cast<size_t>(r)].SendValue<int64_t>(cancelled);
        res.cancelled = cancelled != 0;
        res.seconds_total = Since(t0);
        return res;
    }

    // ---- a rank that analyses into a device log and hands the pieces to rank 0 ----
    Socket master = ConnectTo(cfg.master_addr, cfg.master_port, cfg.connect_timeout_s);
    master.SendValue<int32_t>(rank);
    if (rccl) {
        unsigned char id[PC_COMM_ID_BYTES];
        master.Recv(id, sizeof(id));
        CheckAbi(pc_comm_create(comm_ctx, id, world, rank, &comm), "pc_comm_create");
    }
    if (cfg.synthetic_shard && rccl) Fail("synthetic_shard needs transport tcp");
    const size_t part = LogPartBytes(video_info, cfg);
    std::unique_ptr<DeviceBuffer> log_owner;
    if (!cfg.synthetic_shard) log_owner.reset(new DeviceBuffer(device, 2 * part));
    DeviceBuffer no_log;
    DeviceBuffer& log = log_owner ? *log_owner : no_log;
    constexpr int kDepth = 2;
    std::unique_ptr<DeviceBuffer> staging[kDepth];
    if (rccl)
        for (auto& s : staging) s.reset(new DeviceBuffer(device, part));
    BoundedQueue<Outgoing> outgoing(kDepth);
    BoundedQueue<int> free_slots(kDepth);
    for (int k = 0; k < kDepth; k++) free_slots.Push(k);
    std::exception_ptr sender_error;
    std::thread sender([&] {
        try {
            for (;;) {
                Outgoing o;
                if (!outgoing.Pop(&o)) return;
                const int64_t credit = master.RecvValue<int64_t>();     // rank 0 wants the next piece
                if (credit < 0) Fail("rank 0 aborted the run");
                master.SendValue<Header>(o.h);
                if (o.h.bytes <= 0) return;
                if (rccl) {
                    CheckAbi(pc_comm_send(comm, staging[o.slot]->p, static_cast<uint64_t>(o.h.bytes), 0), "pc_comm_send");
                    free_slots.Push(o.slot);
                } else {
                    master.Send(o.host.data(), o.host.size());
                }
                res.bytes_moved += static_cast<size_t>(o.h.bytes);
                res.pieces++;
            }
        } catch (...) {
            sender_error = std::current_exception();
            outgoing.Close();
            free_slots.Close();
        }
    });
    shard.device_log = log.p;
    shard.capacity_bytes = 2 * part;
    shard.log_buffers = 2;
    shard.piece_frames = cfg.piece_frames;
    shard.host_records = false;
    shard.on_piece = [&](int, size_t offset, size_t bytes, int32_t first_frame1, int n_frames) {
        Outgoing o;
        o.h = Header{static_cast<int64_t>(bytes), n_frames, first_frame1};
        const uint8_t* src = static_cast<const uint8_t*>(log.p) + offset;
        if (rccl) {
            if (!free_slots.Pop(&o.slot, &res.seconds_blocked)) Fail("the sender ended (see its error)");    // waits while two pieces are staged
            CheckAbi(pc_peer_copy_async(device, staging[o.slot]->p, src, bytes, nullptr), "staging copy");
            // the copy runs on the null stream: a blocking one-byte download behind it on the same stream waits for it
            unsigned char probe;
            CheckAbi(pc_peer_buffer_download(device, &probe, staging[o.slot]->p, 1), "staging copy");
        } else {
            o.host.resize(bytes);
            CheckAbi(pc_peer_buffer_download(device, o.host.data(), src, bytes), "download of a piece");
        }
        if (!outgoing.Push(std::move(o), &res.seconds_blocked)) Fail("the sender ended (see its error)");
    };
    std::exception_ptr failure;
    try {
        if (cfg.synthetic_shard) {
            cfg.synthetic_shard(rank, [&](const void* bytes, size_t n, int32_t first_frame1, int n_frames) {
                Outgoing o;
                o.h = Header{static_cast<int64_t>(n), n_frames, first_frame1};
                o.host.assign(static_cast<const uint8_t*>(bytes), static_cast<const uint8_t*>(bytes) + n);
                if (!outgoing.Push(std::move(o), &res.seconds_blocked)) Fail("the sender ended (see its error)");
            });
        } else {
            GenerateOpticalFlowShard(video_info, frame_accessor, callback, "", shard, detector_options, flow_options, &res.stats);
        }
    } catch (...) {
        failure = std::current_exception();
    }
    res.seconds_analysis = Since(t0) - res.seconds_blocked;
    {
        Outgoing last;
        last.h = Header{failure ? kFailed : kEnd, 0, 0};
        outgoing.Push(std::move(last));
    }
    sender.join();
    if (sender_error) std::rethrow_exception(sender_error);    // the first cause (rank 0 gone) rather than its echo
    if (failure) std::rethrow_exception(failure);
    master.SendValue<int64_t>(shard.cancelled ? 1 : 0);
    res.cancelled = master.RecvValue<int64_t>() != 0;
    res.seconds_total = Since(t0);
    return res;
}
