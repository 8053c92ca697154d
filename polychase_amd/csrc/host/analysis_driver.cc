// analysis_driver.cc -- driver of the video-analysis path on top of the C ABI (include/polychase_hip.h).
//
// Control flow mirrors the reference (cpp/opticalflow.cc:209-321): sequential frame1 loop, progress
// callback + cancellation (:238-247), missing-frame errors (:251-254, :311-315), keypoints read from
// the database or detected (:168-178), skips {-8,-4,-2,-1,1,2,4,8} (:76-77), existing pairs skipped
// (:286), status==1 rows written (:130-151).  What differs is WHERE the work happens: every frame
// is uploaded and turned into gray + pyramid exactly once (the reference does it per pair,
// :298-302) and kept in a 17-slot ring on the GPU; frame1 jobs are pipelined through pc_analyzer,
// and the 8 pairs of a frame run as one LK launch instead of 8 TBB tasks.
#include "analysis.h"
#include "numa_pin.h"

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <filesystem>
#include <map>
#include <mutex>
#include <stdexcept>
#include <thread>

#include "../../../include/polychase_hip.h"
#include "debug_images.h"
#include "flow_database.h"
#include "utils.h"

namespace {

constexpr int32_t kImageSkips[8] = {-8, -4, -2, -1, 1, 2, 4, 8};  // cpp/opticalflow.cc:76-77
constexpr int kRing = 17;      // cpp/opticalflow_thread.h:34-79 (SequentialWrapper<17>)
constexpr int kGpuDepth = 3;                        // frame1 jobs kept in flight on the GPU
constexpr int kWriteBacklog = 4;                    // collected records waiting for SQLite
constexpr int kMaxJobs = kGpuDepth + kWriteBacklog + 1;  // pinned result slots: a slot is reused kMaxJobs submits later

[[noreturn]] void ThrowHip(const char* what) {
    throw std::runtime_error(std::string(what) + ": " + pc_last_error());
}

struct Engine {
    pc_context* ctx = nullptr;
    pc_analyzer* an = nullptr;
    // what the engine was created for: an idle engine is reused by a run with the same key
    int device = -1;
    uint32_t width = 0, height = 0;
    pc_gftt_options gopt{};
    pc_flow_options fopt{};
    Engine() = default;
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;
    ~Engine() {
        if (an) pc_analyzer_destroy(an);
        if (ctx) pc_context_destroy(ctx);
    }
    // settings a context / analyzer reads from the environment when it is CREATED (arithmetic mode, stream layout, gate,
    // helper priority, kernel variants): an engine parked under other values must not serve this run
    std::string env_key;
    static std::string EnvKey() {
        std::string k;
        for (const char* name : {"POLYCHASE_ARITH", "POLYCHASE_COPY_STREAM", "POLYCHASE_HELPER_PRIO", "POLYCHASE_LK_GATE", "POLYCHASE_LK_LANES",
                                 "POLYCHASE_DETECT_STREAMS", "POLYCHASE_GFTT_SLOW_PATH", "POLYCHASE_INGEST_DMA"}) {
            const char* v = std::getenv(name);
            k += v ? v : "";
            k += '\x1f';
        }
        return k;
    }
    // field by field: the structs have padding, memcmp would compare it
    bool Matches(int dev, uint32_t w, uint32_t h, const pc_gftt_options& g, const pc_flow_options& f) const {
        return dev == device && w == width && h == height && g.quality_level == gopt.quality_level && g.min_distance == gopt.min_distance &&
               g.block_size == gopt.block_size && g.gradient_size == gopt.gradient_size && g.max_corners == gopt.max_corners &&
               g.use_harris == gopt.use_harris && g.harris_k == gopt.harris_k && g.grid_rows == gopt.grid_rows && g.grid_cols == gopt.grid_cols &&
               f.window_size == fopt.window_size && f.max_level == fopt.max_level && f.term_max_iters == fopt.term_max_iters &&
               f.term_epsilon == fopt.term_epsilon && f.min_eigen_threshold == fopt.min_eigen_threshold && env_key == EnvKey();
    }
};

// ONE idle engine per process.  Creating a context + analyzer (streams, 20 frame slabs, detection scratch, pinned
// result buffers, the copy engines' warm-up) takes 10-20 ms and destroying them 50-80 ms (every hipFree synchronises):
// 0.25 ms per frame of a 300-frame clip at 1080p, more than the analysis of a frame takes.  A run that ends normally
// parks its engine here (pc_analyzer_reset: allocations kept, no state); the next run with the same device, geometry and
// options takes it, any other run replaces it.  release_cached_engine() / POLYCHASE_ENGINE_CACHE=0 give the memory back.
//
// A parked engine holds ~20 frame slabs of GPU memory (several GB at 4K) inside the HOST's process -- Blender -- for a call
// that may never come: it is given back after POLYCHASE_ENGINE_CACHE_IDLE_S seconds without a taker (default 120; 0: keep it
// until release_cached_engine()).  The timer is a thread that EXISTS ONLY WHILE AN ENGINE IS PARKED: it sleeps in one timed
// condition-variable wait until the deadline, and whoever takes, replaces or releases the engine wakes and JOINS it -- no
// thread of this library is left running in a host that holds no parked engine, and none polls.  At process exit (an atexit
// handler registered with the first parked engine, i.e. after the HIP runtime's own and therefore run before them) the timer
// is told to end without touching the GPU and is joined; an engine still parked then is left to the operating system -- no
// GPU call is made while the process is going down.
class EngineCache {
   public:
    static std::unique_ptr<Engine> Take(int dev, uint32_t w, uint32_t h, const pc_gftt_options& g, const pc_flow_options& f) {
        std::unique_ptr<Engine> e = Exchange(nullptr);
        if (e && !e->Matches(dev, w, h, g, f)) e.reset();   // destroyed here, outside every lock
        return e;
    }
    static void Park(std::unique_ptr<Engine> e) {
        static const bool enabled = !(std::getenv("POLYCHASE_ENGINE_CACHE") && std::atoi(std::getenv("POLYCHASE_ENGINE_CACHE")) == 0);
        if (!enabled || !e || pc_analyzer_reset(e->an) != PC_OK) return;   // e is destroyed
        Exchange(std::move(e));   // the engine it replaces, if any, is destroyed here
    }
    static void Clear() { Exchange(nullptr); }
    // testing aid: is a timer thread alive right now?
    static bool TimerRunning() {
        State& st = S();
        std::lock_guard<std::mutex> lk(st.m);
        return st.timer_alive;
    }

   private:
    struct State {
        std::mutex lifecycle;            // serialises Take / Park / Clear / exit (rare calls); never held by the timer
        std::mutex m;                    // slot, deadline, flags
        std::condition_variable cv;
        std::unique_ptr<Engine> slot;
        std::chrono::steady_clock::time_point deadline;
        std::thread timer;
        bool stop = false, exiting = false, exit_hook = false, timer_alive = false;
    };
    static State& S() {
        // leaked on purpose: at process exit the HIP runtime may be gone before a static destructor would run
        static State* s = new State();
        return *s;
    }
    static double IdleSeconds() {
        static const double idle_s = [] {
            const char* v = std::getenv("POLYCHASE_ENGINE_CACHE_IDLE_S");
            return v ? std::atof(v) : 120.0;
        }();
        return idle_s;
    }
    // puts `e` (or nothing) into the slot and returns what was there; the timer of the old engine is ended and joined,
    // a new one is started for the new engine
    static std::unique_ptr<Engine> Exchange(std::unique_ptr<Engine> e) {
        State& st = S();
        std::lock_guard<std::mutex> life(st.lifecycle);
        std::unique_ptr<Engine> old;
        std::thread old_timer;
        {
            std::lock_guard<std::mutex> lk(st.m);
            old = std::move(st.slot);
            old_timer = std::move(st.timer);
            st.stop = true;
        }
        st.cv.notify_all();
        if (old_timer.joinable()) old_timer.join();   // at most the 50-80 ms of an engine it is just destroying
        if (e) {
            std::lock_guard<std::mutex> lk(st.m);
            st.stop = false;
            if (st.exiting) return old;   // too late to park anything: `e` is destroyed by the caller's unique_ptr
            st.slot = std::move(e);
            const double idle_s = IdleSeconds();
            if (idle_s > 0) {
                st.deadline = std::chrono::steady_clock::now() + std::chrono::duration_cast<std::chrono::steady_clock::duration>(
                                                                     std::chrono::duration<double>(idle_s));
                st.timer_alive = true;
                st.timer = std::thread(TimerMain);
                if (!st.exit_hook) {
                    st.exit_hook = true;
                    std::atexit(AtExit);
                }
            }
        }
        return old;
    }
    static void TimerMain() {
        State& st = S();
        std::unique_ptr<Engine> idle;
        {
            std::unique_lock<std::mutex> lk(st.m);
            st.cv.wait_until(lk, st.deadline, [&] { return st.stop || st.exiting; });
            st.timer_alive = false;
            if (st.stop || st.exiting) return;
            idle = std::move(st.slot);
        }
        // destroyed here, outside the lock; Exchange / AtExit join this thread, so the destruction is complete before a
        // new engine is parked or the process goes on exiting
    }
    static void AtExit() {
        State& st = S();
        std::thread t;
        {
            std::lock_guard<std::mutex> lk(st.m);
            st.exiting = true;
            t = std::move(st.timer);
        }
        st.cv.notify_all();
        if (t.joinable()) t.join();
    }
};

double Now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// SQLite inserts on their own thread, overlapping the GPU pipeline.  Records are consumed in
// submission order straight from the analyzer's pinned buffers (no copy); the backlog bound keeps a
// buffer from being reused before it is written.  `db_mtx` also serialises the driver's reads
// (the connection is opened NOMUTEX like the reference's, cpp/database.cc:71-74).
class RecordWriter {
   public:
    RecordWriter(Database* db, std::mutex* db_mtx, OpticalFlowRunStats* stats)
        : db_(db), db_mtx_(db_mtx), stats_(stats), batch_(BatchFrames()), thread_([this] { Run(); }) {}
    static int BatchFrames() {
        const char* env = std::getenv("POLYCHASE_DB_BATCH_FRAMES");   // read per run
        const int n = env ? std::atoi(env) : 8;
        return n < 1 ? 1 : n;
    }
    ~RecordWriter() {
        {
            std::lock_guard<std::mutex> lk(mtx_);
            quit_ = true;
        }
        cv_.notify_all();
        if (thread_.joinable()) thread_.join();
    }
    // blocks while kWriteBacklog records are already waiting
    void Enqueue(const pc_frame_result& r) {
        std::unique_lock<std::mutex> lk(mtx_);
        cv_.wait(lk, [&] { return static_cast<int>(queue_.size()) < kWriteBacklog || error_; });
        Rethrow();
        queue_.push_back(r);
        cv_.notify_all();
    }
    void Flush() {
        std::unique_lock<std::mutex> lk(mtx_);
        cv_.wait(lk, [&] { return (queue_.empty() && !busy_) || error_; });
        Rethrow();
    }
    double seconds() const { return seconds_; }

   private:
    struct WriterStopped {};
    void Rethrow() {
        if (error_) std::rethrow_exception(error_);
    }
    // Records go in as transactions of up to batch_ frames (POLYCHASE_DB_BATCH_FRAMES, default 8; 1 = one transaction per
    // frame, round 4's behaviour): a commit costs the journal's fsync-free bookkeeping plus the page-cache flush of what the
    // transaction touched, ~8 % of the insert at one frame per commit (profiles/r03_dbfloor.json: one transaction per clip
    // +8 %).  A transaction also ends whenever the queue runs empty -- a slow frame source (a renderer) has every frame
    // committed as it arrives -- and at Flush() / destruction.  What a crash or an exception loses is at most the batch's
    // finished frames: the file is rolled back to the last commit and a resumed run recomputes exactly those rows
    // (cpp/opticalflow.cc:168-178, :286: rows that exist are skipped), bit for bit.
    void Run() {
        numa::PinThisThreadNearGpu(nullptr, "analysis: database writer");
        bool in_transaction = false;
        int in_batch = 0;
        auto fail = [&](std::exception_ptr e) {
            if (in_transaction) {
                try {
                    std::lock_guard<std::mutex> dblk(*db_mtx_);
                    db_->Rollback();   // or every later Begin fails with "cannot start a transaction within a transaction"
                } catch (...) {
                }
                in_transaction = false;
                in_batch = 0;
            }
            std::lock_guard<std::mutex> lk(mtx_);
            if (!error_) error_ = e;   // the first error is the cause; later ones are its echo
        };
        for (;;) {
            pc_frame_result r;
            bool last_of_burst = false;
            {
                std::unique_lock<std::mutex> lk(mtx_);
                cv_.wait(lk, [&] { return !queue_.empty() || quit_; });
                if (queue_.empty()) break;
                r = queue_.front();
                last_of_burst = queue_.size() == 1;
                busy_ = true;
            }
            try {
                const double t0 = Now();
                std::lock_guard<std::mutex> dblk(*db_mtx_);
                {
                    std::lock_guard<std::mutex> lk(mtx_);
                    if (error_) throw WriterStopped{};   // an earlier record failed: nothing more is written
                }
                if (!in_transaction) {
                    db_->Begin();
                    in_transaction = true;
                }
                if (r.keypoints_detected && !db_->KeypointsExist(r.frame1)) {
                    db_->WriteKeypoints(r.frame1, r.keypoints_xy, static_cast<size_t>(r.n_keypoints));
                    stats_->keypoint_rows_written++;
                }
                for (int t = 0; t < r.n_targets; t++) {
                    const int64_t a = r.row_offset[t], b = r.row_offset[t + 1];
                    db_->WriteImagePairFlow(r.frame1, r.targets[t], r.src_indices + a, r.tgt_xy + 2 * a, r.flow_err + a,
                                            static_cast<size_t>(b - a));
                    stats_->flow_rows_written++;
                }
                if (++in_batch >= batch_ || last_of_burst) {
                    db_->Commit();
                    in_transaction = false;
                    in_batch = 0;
                }
                seconds_ += Now() - t0;
            } catch (const WriterStopped&) {
            } catch (...) {
                fail(std::current_exception());
            }
            {
                std::lock_guard<std::mutex> lk(mtx_);
                queue_.pop_front();  // only now may the pinned buffers of this record be reused (SQLite has copied the blobs)
                busy_ = false;
            }
            cv_.notify_all();
        }
        if (in_transaction) {   // (only when the last record's write threw before its commit and the rollback failed too)
            try {
                std::lock_guard<std::mutex> dblk(*db_mtx_);
                db_->Commit();
            } catch (...) {
                fail(std::current_exception());
            }
        }
    }

    Database* db_;
    std::mutex* db_mtx_;
    OpticalFlowRunStats* stats_;
    std::mutex mtx_;
    std::condition_variable cv_;
    std::deque<pc_frame_result> queue_;
    bool quit_ = false, busy_ = false;
    std::exception_ptr error_;
    double seconds_ = 0;
    int batch_ = 8;
    std::thread thread_;   // last: it starts in the constructor's initialiser list
};

}  // namespace

// One piece of a shard's device log: the frame1 jobs submitted into one part of the log buffer
struct LogPiece {
    int index = 0;
    size_t offset = 0, bytes = 0;
    int32_t first_frame1 = 0;
    int submitted = 0, collected = 0;
    bool closed = false;
};

static void RunAnalysis(const VideoInfo& video_info, FrameAccessorFunction frame_accessor,
                        OpticalFlowProgressCallback callback, const std::string& database_path,
                        const GFTTOptions& detector_options, const OpticalFlowOptions& flow_options,
                        OpticalFlowRunStats* stats, OpticalFlowShard* shard, bool write_images) {
    CHECK(frame_accessor);
    const double t_begin = Now();
    std::unique_ptr<Database> db;
    if (!database_path.empty()) db = std::make_unique<Database>(database_path, /*bulk_writer=*/true);
    // Bulk load: rows go in under a rollback journal and the file is put back into WAL mode on every way out (done,
    // cancelled, exception) -- 2.3x the insert rate of WAL mode on this workload (tools/dbprobe/dbbench.py), same
    // file format in the end.  A process that dies in between leaves a rollback-journal database with a hot
    // journal, which the next Open() -- ours or the reference's -- recovers and switches to WAL.
    // POLYCHASE_DB_BULK_LOAD=0 keeps WAL mode throughout.
    struct BulkLoad {
        Database* db = nullptr;
        ~BulkLoad() {
            if (!db) return;
            try {
                db->SetJournalMode("WAL");
            } catch (...) {
            }
        }
    } bulk_load;   // declared before the writer: destroyed after it has drained
    if (db) {
        const char* env = std::getenv("POLYCHASE_DB_BULK_LOAD");
        if (!(env && env[0] == '0') && db->SetJournalMode("TRUNCATE") == "truncate") bulk_load.db = db.get();
    }

    const int32_t from = video_info.first_frame;
    const int32_t to = video_info.first_frame + static_cast<int32_t>(video_info.num_frames);
    // frame1 ids of this run, and the frames it has to see (a shard: 8 more on both sides, as targets only)
    const int32_t f1_begin = shard ? std::max(shard->begin, from) : from;
    const int32_t f1_end = shard ? std::min(shard->end, to) : to;
    const int32_t res_begin = shard ? std::max(from, f1_begin - 8) : from;
    const int32_t res_end = shard ? std::min(to, f1_end + 8) : to;

    pc_gftt_options gopt{};
    gopt.quality_level = detector_options.quality_level;
    gopt.min_distance = detector_options.min_distance;
    gopt.block_size = detector_options.block_size;
    gopt.gradient_size = detector_options.gradient_size;
    gopt.max_corners = detector_options.max_corners;
    gopt.use_harris = detector_options.use_harris ? 1 : 0;
    gopt.harris_k = detector_options.harris_k;
    gopt.grid_rows = detector_options.grid_rows;
    gopt.grid_cols = detector_options.grid_cols;
    pc_flow_options fopt{};
    fopt.window_size = flow_options.window_size;
    fopt.max_level = flow_options.max_level;
    fopt.term_max_iters = flow_options.term_max_iters;
    fopt.term_epsilon = flow_options.term_epsilon;
    fopt.min_eigen_threshold = flow_options.min_eigen_threshold;

    // owners of frames the GPU may still be reading; declared BEFORE the engine, i.e. destroyed AFTER it: on every way
    // out (exceptions included) the analyzer has synchronised its streams before a buffer goes back to the pool
    std::deque<std::pair<int32_t, std::shared_ptr<void>>> frames_in_flight;
    int device = 0;
    if (shard && shard->device >= 0) device = shard->device;   // a rank of the multi-GPU analysis names its GPU itself
    else if (const char* env = std::getenv("POLYCHASE_DEVICE")) device = std::atoi(env);
    std::unique_ptr<Engine> engine = EngineCache::Take(device, video_info.width, video_info.height, gopt, fopt);
    const bool engine_reused = engine != nullptr;
    if (!engine) {
        engine = std::make_unique<Engine>();
        engine->device = device;
        engine->width = video_info.width;
        engine->height = video_info.height;
        engine->gopt = gopt;
        engine->fopt = fopt;
        engine->env_key = Engine::EnvKey();
        if (pc_context_create(device, &engine->ctx) != PC_OK) ThrowHip("pc_context_create");
        if (pc_analyzer_create(engine->ctx, static_cast<int>(video_info.width), static_cast<int>(video_info.height), &gopt,
                               &fopt, kRing, kMaxJobs, &engine->an) != PC_OK)
            ThrowHip("pc_analyzer_create");
    }
    Engine& eng = *engine;
    // the threads that feed this GPU stay on its NUMA node (numa_pin.h): this one for the duration of the call, the writer and its
    // page-write worker (started below / when the database opens its file) for their lives
    numa::ScopedPin near_gpu(eng.ctx, "analysis: calling thread");
    // The shard's device log: `log_buffers` equal parts of the caller's buffer, filled in turn; a part is handed to
    // on_piece when every job submitted into it has been collected, and reused `log_buffers` pieces later.
    const bool with_log = shard && shard->device_log != nullptr;
    const int n_parts = with_log ? std::max(1, shard->log_buffers) : 1;
    const size_t part_bytes = with_log ? (shard->capacity_bytes / static_cast<size_t>(n_parts)) & ~static_cast<size_t>(15) : 0;
    std::deque<LogPiece> pieces;   // oldest first; the back one is open (jobs are still being submitted into it)
    int next_piece = 0;
    auto log_part = [&](int piece) { return static_cast<uint8_t*>(shard->device_log) + static_cast<size_t>(piece % n_parts) * part_bytes; };
    if (with_log) {
        CHECK(part_bytes >= 512);
        if (pc_analyzer_set_device_log(eng.an, log_part(0), part_bytes) != PC_OK) ThrowHip("pc_analyzer_set_device_log");
        if (!db && !shard->host_records && pc_analyzer_set_host_records(eng.an, 0) != PC_OK) ThrowHip("pc_analyzer_set_host_records");
    }

    // write_images (opticalflow.cc:228-232, :265-267): <database dir>/frames/%06d.png and keypoints_%06d.png of every
    // frame1.  The keypoints are known when the job is collected: host copies of the frames wait for that here.
    std::string frames_dir;
    std::map<int32_t, std::vector<uint8_t>> debug_frames;
    if (write_images) {
        const std::filesystem::path dir = std::filesystem::path(database_path).parent_path() / "frames";
        std::filesystem::create_directory(dir);
        frames_dir = dir.string();
    }
    OpticalFlowRunStats local_stats;
    local_stats.seconds_setup = Now() - t_begin;
    local_stats.engine_reused = engine_reused;
    struct StageClock {   // adds the time of a scope to one of the stage counters
        double* acc;
        double t0;
        explicit StageClock(double* a) : acc(a), t0(Now()) {}
        ~StageClock() { *acc += Now() - t0; }
    };
    std::mutex db_mtx;
    // The reference asks the database before every piece of work whether it is there already (keypoints :168-178, pairs :286):
    // that is how a cancelled analysis resumes.  A database that holds NO keypoints row of this clip when the run starts cannot
    // answer "yes" to any of those questions later either -- the rows this run writes are of frames whose questions have
    // been asked (a flow row needs its frame's keypoints row: the schema's foreign key) -- so the questions are not asked: ten
    // lock acquisitions per frame that each waited for the writer thread's current record (0.8 ms at 1080p) and kept its queue
    // empty, i.e. every record in a transaction of its own whatever POLYCHASE_DB_BATCH_FRAMES said.
    bool resume = false;
    if (db) {
        const int32_t lo = db->GetMinImageIdWithKeypoints(), hi = db->GetMaxImageIdWithKeypoints();
        resume = lo != kInvalidId && hi != kInvalidId && hi >= video_info.first_frame &&
                 lo < video_info.first_frame + static_cast<int32_t>(video_info.num_frames);
    }
    std::unique_ptr<RecordWriter> writer;
    if (db) writer = std::make_unique<RecordWriter>(db.get(), &db_mtx, &local_stats);

    // RequestFrame + shape checks (cpp/opticalflow.cc:189-202)
    auto fetch = [&](int32_t frame_id) -> std::optional<FrameView> {
        std::optional<FrameView> f;
        {
            StageClock clk(&local_stats.seconds_accessor);
            f = frame_accessor(frame_id);
        }
        if (f) {
            CHECK_EQ(static_cast<uint32_t>(f->rows), video_info.height);
            CHECK_EQ(static_cast<uint32_t>(f->cols), video_info.width);
            if (f->elem_size == 4) CHECK(f->channels == 3 || f->channels == 4);
            else CHECK_EQ(static_cast<uint32_t>(f->channels), 3u);
        }
        return f;
    };

    auto hand_off_complete_pieces = [&]() {
        while (!pieces.empty() && pieces.front().closed && pieces.front().collected == pieces.front().submitted) {
            const LogPiece pc = pieces.front();
            pieces.pop_front();
            shard->used_bytes = pc.bytes;
            shard->pieces++;
            if (shard->on_piece) shard->on_piece(pc.index, pc.offset, pc.bytes, pc.first_frame1, pc.submitted);
        }
    };
    auto collect_one = [&]() {
        pc_frame_result r;
        {
            StageClock clk(&local_stats.seconds_collect);
            if (pc_analyzer_collect(eng.an, &r) != PC_OK) ThrowHip("pc_analyzer_collect");
        }
        local_stats.frames_processed++;
        if (write_images) {
            auto it = debug_frames.find(r.frame1);
            if (it != debug_frames.end()) {
                SaveImageForDebugging(it->second.data(), static_cast<int>(video_info.width), static_cast<int>(video_info.height), r.frame1,
                                      frames_dir, r.keypoints_xy, r.n_keypoints);
                debug_frames.erase(debug_frames.begin(), std::next(it));   // frame1 ids only grow
            }
        }
        if (writer) {
            StageClock clk(&local_stats.seconds_writer_wait);
            writer->Enqueue(r);
        }
        if (with_log) {
            // jobs come back in submission order: this one belongs to the oldest piece that still misses one
            for (LogPiece& pc : pieces)
                if (pc.collected < pc.submitted) {
                    pc.collected++;
                    break;
                }
            hand_off_complete_pieces();
        }
    };
    // the open piece takes no more jobs: its bytes are final; the next job goes to the next part of the buffer -- once
    // the piece that used that part has been handed off
    auto close_piece = [&]() {
        if (pieces.empty() || pieces.back().closed) return;
        LogPiece& pc = pieces.back();
        if (pc_analyzer_device_log_used(eng.an, &pc.bytes) != PC_OK) ThrowHip("pc_analyzer_device_log_used");
        pc.closed = true;
        hand_off_complete_pieces();
    };
    auto open_piece = [&](int32_t first_frame1) {
        const int index = next_piece++;
        while (!pieces.empty() && pieces.front().index <= index - n_parts) {   // that part of the buffer is still in use
            if (pc_analyzer_pending(eng.an) == 0) throw std::logic_error("device log piece neither collected nor handed over");
            collect_one();
        }
        LogPiece pc;
        pc.index = index;
        pc.offset = static_cast<size_t>(index % n_parts) * part_bytes;
        pc.first_frame1 = first_frame1;
        pieces.push_back(pc);
        if (index > 0 && pc_analyzer_redirect_device_log(eng.an, log_part(index), part_bytes) != PC_OK)
            ThrowHip("pc_analyzer_redirect_device_log");
    };
    auto drain = [&]() {
        while (pc_analyzer_pending(eng.an) > 0) collect_one();
        if (with_log) close_piece();
        if (writer) {
            StageClock clk(&local_stats.seconds_writer_wait);
            writer->Flush();
        }
    };
    auto finish_stats = [&]() {
        local_stats.seconds_total = Now() - t_begin;
        local_stats.seconds_db = writer ? writer->seconds() : 0.0;
        if (stats) *stats = local_stats;
    };

    // A frame handed over as device / pinned memory is read by the GPU after put_frame has returned: its owner is kept
    // until the analyzer reports the pixels consumed (not "eight newer frames later", which a GPU busy with other work
    // -- Blender rendering on the same device -- could outlast)
    auto release_ingested = [&](bool wait) {
        while (!frames_in_flight.empty()) {
            if (!pc_analyzer_frame_ingested(eng.an, frames_in_flight.front().first)) {
                if (!wait) break;
                std::this_thread::sleep_for(std::chrono::microseconds(50));
                continue;
            }
            frames_in_flight.pop_front();
        }
    };
    int32_t highest_put = res_begin - 1;
    Keypoints known;
    for (int32_t frame_id1 = f1_begin; frame_id1 < f1_end; frame_id1++) {
        if (callback) {
            const float progress = shard ? static_cast<float>(frame_id1 - f1_begin) / static_cast<float>(std::max(1, f1_end - f1_begin))
                                         : static_cast<float>(frame_id1 - from) / static_cast<float>(video_info.num_frames);
            const bool ok = callback(progress, "Processing frame " + std::to_string(frame_id1));
            if (!ok) {
                drain();  // jobs already on the GPU are complete work: keep them (a shard: their pieces have been handed off)
                if (shard) shard->cancelled = true;
                finish_stats();
                release_ingested(true);
                EngineCache::Park(std::move(engine));
                callback(1.0f, "Cancelled");
                return;
            }
        }
        // make frame1 .. frame1+8 resident, and one frame more (its pyramid is then ready a whole step before the first
        // launch that reads it); every frame is requested exactly once, in increasing order
        const int32_t upto = std::min(frame_id1 + 8 + PC_ANALYZER_LOOKAHEAD, res_end - 1);
        for (int32_t fid = std::max(highest_put + 1, std::max(res_begin, frame_id1 - 8)); fid <= upto; fid++) {
            std::optional<FrameView> f = fetch(fid);
            if (!f) {
                drain();
                if (fid == frame_id1)
                    throw std::runtime_error("Rquested frame #" + std::to_string(fid) + " was not provided");
                throw std::runtime_error(
                    "Exiting optical flow generation prematurely because some frames were not provided");
            }
            bool will_detect = fid >= f1_begin && fid < f1_end;   // the halo of a shard is tracked into, never from
            if (db && resume && will_detect) {
                std::lock_guard<std::mutex> lk(db_mtx);
                will_detect = !db->KeypointsExist(fid);
            }
            static const bool ingest_dma = !(std::getenv("POLYCHASE_INGEST_DMA") && std::atoi(std::getenv("POLYCHASE_INGEST_DMA")) == 0);
            const int where = !f->on_device ? 0 : (f->pinned_host && f->owner && ingest_dma ? PC_FRAME_PINNED_HOST : 1);
            StageClock put_clk(&local_stats.seconds_put);
            const int put_rc =
                f->elem_size == 4
                    ? pc_analyzer_put_frame_f32(eng.an, fid, reinterpret_cast<const float*>(f->data), f->row_pitch, f->channels,
                                                where, will_detect ? 1 : 0)
                    : pc_analyzer_put_frame(eng.an, fid, f->data, f->row_pitch, where, will_detect ? 1 : 0);
            if (write_images && fid >= f1_begin && fid < f1_end) {
                // tightly packed 8-bit RGB on the host; float frames like the addon's `(image * 255).astype(np.uint8)`
                const size_t w = video_info.width, h = video_info.height, row = static_cast<size_t>(f->row_pitch);
                std::vector<uint8_t> raw(row * h);
                if (f->on_device && !f->pinned_host) {
                    if (pc_context_download(eng.ctx, raw.data(), f->data, row * (h - 1) + w * f->channels * f->elem_size) != PC_OK)
                        ThrowHip("pc_context_download");
                } else {
                    std::memcpy(raw.data(), f->data, row * (h - 1) + w * f->channels * f->elem_size);
                }
                std::vector<uint8_t>& img = debug_frames[fid];
                img.resize(w * h * 3);
                for (size_t y = 0; y < h; y++)
                    for (size_t x = 0; x < w; x++)
                        for (int c = 0; c < 3; c++) {
                            if (f->elem_size == 4) {
                                const float val = reinterpret_cast<const float*>(raw.data() + y * row)[x * f->channels + c] * 255.f;
                                img[(y * w + x) * 3 + c] = static_cast<uint8_t>(val < 0.f ? 0.f : (val > 255.f ? 255.f : val));
                            } else {
                                img[(y * w + x) * 3 + c] = raw[y * row + x * 3 + c];
                            }
                        }
            }
            // the owner first: put_frame may have enqueued a copy out of the buffer before it failed, and
            // frames_in_flight outlives the engine (whose destructor synchronises) while `f` does not
            if (f->on_device && f->owner) frames_in_flight.emplace_back(fid, std::move(f->owner));
            if (put_rc != PC_OK) ThrowHip("pc_analyzer_put_frame");
            release_ingested(frames_in_flight.size() > 24);   // bounded: the pool behind the owners is finite
            highest_put = fid;
        }
        // ReadOrGenerateKeypoints (:168-178)
        if (db && resume) {
            known.clear();
            {
                std::lock_guard<std::mutex> lk(db_mtx);
                db->ReadKeypoints(frame_id1, known);
            }
            if (!known.empty() &&
                pc_analyzer_set_keypoints(eng.an, frame_id1, known[0].data(), static_cast<int>(known.size())) != PC_OK)
                ThrowHip("pc_analyzer_set_keypoints");
        }
        int32_t targets[PC_MAX_TARGETS];
        int n_targets = 0;
        for (int32_t skip : kImageSkips) {
            const int32_t frame_id2 = frame_id1 + skip;
            if (frame_id2 < from || frame_id2 >= to) continue;               // :282
            if (db && resume) {
                std::lock_guard<std::mutex> lk(db_mtx);
                if (db->ImagePairFlowExists(frame_id1, frame_id2)) continue;  // :286
            }
            targets[n_targets++] = frame_id2;
        }
        while (pc_analyzer_pending(eng.an) >= kGpuDepth) collect_one();
        if (with_log) {
            if (!pieces.empty() && !pieces.back().closed && shard->piece_frames > 0 && pieces.back().submitted >= shard->piece_frames)
                close_piece();
            if (pieces.empty() || pieces.back().closed) open_piece(frame_id1);
        }
        StageClock submit_clk(&local_stats.seconds_submit);
        int rc = pc_analyzer_submit(eng.an, frame_id1, targets, n_targets);
        if (rc == PC_E_CAPACITY && with_log && pieces.back().submitted > 0 && (shard->piece_frames > 0 || n_parts > 1)) {
            // the record does not fit what is left of this part of the log: the piece ends here
            close_piece();
            open_piece(frame_id1);
            rc = pc_analyzer_submit(eng.an, frame_id1, targets, n_targets);
        }
        if (rc != PC_OK) ThrowHip("pc_analyzer_submit");
        if (with_log) pieces.back().submitted++;
    }
    drain();
    finish_stats();
    release_ingested(true);               // no pinned frame buffer is still being read
    EngineCache::Park(std::move(engine));   // (an exception on the way leaves `engine` to its destructor instead)
    if (callback) callback(1.0f, "Done");
}

void ReleaseCachedEngine() { EngineCache::Clear(); }
bool EngineCacheTimerRunning() { return EngineCache::TimerRunning(); }

void GenerateOpticalFlowDatabase(const VideoInfo& video_info, FrameAccessorFunction frame_accessor,
                                 OpticalFlowProgressCallback callback, const std::string& database_path,
                                 const GFTTOptions& detector_options, const OpticalFlowOptions& flow_options,
                                 bool write_images, OpticalFlowRunStats* stats) {
    RunAnalysis(video_info, std::move(frame_accessor), std::move(callback), database_path, detector_options, flow_options, stats,
                nullptr, write_images);
}

void GenerateOpticalFlowShard(const VideoInfo& video_info, FrameAccessorFunction frame_accessor, OpticalFlowProgressCallback callback,
                              const std::string& database_path, OpticalFlowShard& shard, const GFTTOptions& detector_options,
                              const OpticalFlowOptions& flow_options, OpticalFlowRunStats* stats) {
    CHECK(shard.begin <= shard.end);
    CHECK(shard.device_log != nullptr || !database_path.empty());
    shard.used_bytes = 0;
    shard.pieces = 0;
    shard.cancelled = false;
    RunAnalysis(video_info, std::move(frame_accessor), std::move(callback), database_path, detector_options, flow_options, stats, &shard, false);
}

size_t GenerateOpticalFlowRecords(const VideoInfo& video_info, FrameAccessorFunction frame_accessor,
                                  OpticalFlowProgressCallback callback, int32_t shard_begin, int32_t shard_end,
                                  void* device_log, size_t capacity_bytes, const GFTTOptions& detector_options,
                                  const OpticalFlowOptions& flow_options, OpticalFlowRunStats* stats) {
    CHECK(device_log != nullptr);
    OpticalFlowShard shard;
    shard.begin = shard_begin;
    shard.end = shard_end;
    shard.device_log = device_log;
    shard.capacity_bytes = capacity_bytes;
    GenerateOpticalFlowShard(video_info, std::move(frame_accessor), std::move(callback), "", shard, detector_options, flow_options, stats);
    return shard.used_bytes;
}

// ---- OpticalFlowRecordWriter: record logs -> database, one connection (bulk-load journal) for many logs --------------
struct OpticalFlowRecordWriter::Impl {
    Database db;
    bool bulk = false;
    explicit Impl(const std::string& path) : db(path, /*bulk_writer=*/true) {
        const char* env = std::getenv("POLYCHASE_DB_BULK_LOAD");
        bulk = !(env && env[0] == '0') && db.SetJournalMode("TRUNCATE") == "truncate";   // see RunAnalysis
    }
    ~Impl() {
        if (!bulk) return;
        try {
            db.SetJournalMode("WAL");
        } catch (...) {
        }
    }
};
OpticalFlowRecordWriter::OpticalFlowRecordWriter(const std::string& database_path) {
    CHECK(!database_path.empty());
    impl_ = std::make_unique<Impl>(database_path);
}
OpticalFlowRecordWriter::~OpticalFlowRecordWriter() = default;
void OpticalFlowRecordWriter::Close() { impl_.reset(); }

void OpticalFlowRecordWriter::Write(const uint8_t* log, size_t bytes, OpticalFlowRunStats* stats) {
    if (!impl_) throw std::runtime_error("OpticalFlowRecordWriter is closed");
    Database& db = impl_->db;
    OpticalFlowRunStats local;
    Keypoints known;
    const double t0 = Now();
    auto up16 = [](size_t v) { return (v + 15) & ~static_cast<size_t>(15); };
    const char* batch_env = std::getenv("POLYCHASE_DB_BATCH_FRAMES");
    const int batch = std::max(1, batch_env ? std::atoi(batch_env) : 8);
    int in_batch = 0;
    size_t o = 0;
    struct RollbackOnThrow {   // an exception between Begin and Commit must not leave the connection inside a transaction
        Database& db;
        int& open;
        ~RollbackOnThrow() {
            if (open > 0 && std::uncaught_exceptions() > 0) {
                try {
                    db.Rollback();
                } catch (...) {
                }
            }
        }
    } rollback_guard{db, in_batch};
    while (o < bytes) {
        // header (128 B): magic, frame1, keypoints, targets, target ids [8], rows; then the packed record:
        // row offsets (128 B) | keypoints | src indices | tgt xy | errors, every part 16-byte aligned
        if (o + 256 > bytes) throw std::runtime_error("truncated optical-flow record log");
        const long long* hdr = reinterpret_cast<const long long*>(log + o);
        if (hdr[0] != PC_LOG_MAGIC) throw std::runtime_error("corrupt optical-flow record log");
        const int32_t frame1 = static_cast<int32_t>(hdr[1]);
        const size_t n = static_cast<size_t>(hdr[2]), rows = static_cast<size_t>(hdr[12]);
        const int nt = static_cast<int>(hdr[3]);
        if (nt < 0 || nt > PC_MAX_TARGETS) throw std::runtime_error("corrupt optical-flow record log");
        const long long* off = reinterpret_cast<const long long*>(log + o + 128);
        const size_t o_kps = o + 256, o_idx = up16(o_kps + n * 8), o_xy = up16(o_idx + rows * 4), o_err = up16(o_xy + rows * 8);
        const size_t end = up16(o_err + rows * 4);
        if (end > bytes) throw std::runtime_error("truncated optical-flow record log");
        if (in_batch == 0) db.Begin();
        if (!db.KeypointsExist(frame1)) {
            db.WriteKeypoints(frame1, reinterpret_cast<const float*>(log + o_kps), n);
            local.keypoint_rows_written++;
        } else {
            // the flows of this record index the record's keypoints: a stored row must be the same list (a database of
            // another run -- other detector options, the reference's detector -- would end up with flows that point
            // at the wrong corners; the single-process path tracks FROM the stored keypoints instead, :168-178)
            known.clear();
            db.ReadKeypoints(frame1, known);
            if (known.size() != n || (n > 0 && std::memcmp(known[0].data(), log + o_kps, n * 8) != 0)) {
                if (in_batch > 0) db.Commit();   // the records before this one are complete: they stay
                else db.Rollback();              // this record opened the transaction: nothing in it
                in_batch = 0;
                throw std::runtime_error("keypoints of frame " + std::to_string(frame1) +
                                         " stored in the database differ from the record's: refusing to mix two analyses");
            }
        }
        for (int t = 0; t < nt; t++) {
            const int32_t frame2 = static_cast<int32_t>(hdr[4 + t]);
            if (db.ImagePairFlowExists(frame1, frame2)) continue;
            const size_t a = static_cast<size_t>(off[t]), b = static_cast<size_t>(off[t + 1]);
            if (b < a || b > rows) throw std::runtime_error("corrupt optical-flow record log");
            db.WriteImagePairFlow(frame1, frame2, reinterpret_cast<const uint32_t*>(log + o_idx) + a,
                                  reinterpret_cast<const float*>(log + o_xy) + 2 * a, reinterpret_cast<const float*>(log + o_err) + a, b - a);
            local.flow_rows_written++;
        }
        // transactions of up to POLYCHASE_DB_BATCH_FRAMES records (default 8, like the single-process writer thread); every
        // Write() ends with a commit, so a piece that has been stored is on disk when the call returns
        if (++in_batch >= batch) {
            db.Commit();
            in_batch = 0;
        }
        local.frames_processed++;
        o = end;
    }
    if (in_batch > 0) db.Commit();
    local.seconds_db = local.seconds_total = Now() - t0;
    if (stats) *stats = local;
}

void WriteOpticalFlowRecords(const std::string& database_path, const uint8_t* log, size_t bytes, OpticalFlowRunStats* stats) {
    OpticalFlowRecordWriter w(database_path);
    w.Write(log, bytes, stats);
}
