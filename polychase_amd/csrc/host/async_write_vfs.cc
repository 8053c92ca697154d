// async_write_vfs.cc -- see async_write_vfs.h.
#include "async_write_vfs.h"

#include "numa_pin.h"

#include <sqlite3.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <set>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int kSlabBytes = 65536;   // SQLite's largest page

std::atomic<unsigned long long> g_deferred_writes{0}, g_deferred_bytes{0}, g_direct_writes{0}, g_drains{0}, g_drains_waited{0}, g_slab_waits{0};

// ---- the deferred writes of one database file -------------------------------------------------------------------------------
class Writer {
   public:
    Writer(sqlite3_file* real, int n_lanes, int n_slabs) : real_(real), lanes_(static_cast<size_t>(n_lanes)) {
        const size_t bytes = static_cast<size_t>(n_slabs) * kSlabBytes;
        void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) throw std::bad_alloc();
        (void)madvise(p, bytes, MADV_HUGEPAGE);
        slabs_ = static_cast<char*>(p);
        slab_bytes_ = bytes;
        for (int i = n_slabs - 1; i >= 0; i--) free_.push_back(i);
        try {
            for (Lane& lane : lanes_) lane.thread = std::thread([this, &lane] { Run(lane); });
        } catch (...) {
            // a thread could not be created: the lanes that did start are stopped and joined -- destroying a joinable
            // std::thread terminates the process
            for (Lane& lane : lanes_) {
                if (!lane.thread.joinable()) continue;
                {
                    std::lock_guard<std::mutex> lk(lane.m);
                    lane.stop = true;
                }
                lane.cv.notify_one();
                lane.thread.join();
            }
            munmap(slabs_, slab_bytes_);
            throw;
        }
    }
    ~Writer() {
        (void)Drain();
        for (Lane& lane : lanes_) {
            {
                std::lock_guard<std::mutex> lk(lane.m);
                lane.stop = true;
            }
            lane.cv.notify_one();
        }
        for (Lane& lane : lanes_) lane.thread.join();
        munmap(slabs_, slab_bytes_);
    }
    Writer(const Writer&) = delete;
    Writer& operator=(const Writer&) = delete;

    // copies `amt` <= kSlabBytes bytes and returns; the write itself happens on the lane of the page
    int Write(const void* buf, int amt, sqlite3_int64 off) {
        if (error_.load(std::memory_order_acquire)) return SQLITE_IOERR_WRITE;
        int slab;
        {
            std::unique_lock<std::mutex> lk(free_m_);
            if (free_.empty()) {
                g_slab_waits++;
                free_cv_.wait(lk, [&] { return !free_.empty(); });
            }
            slab = free_.back();
            free_.pop_back();
        }
        std::memcpy(slabs_ + static_cast<size_t>(slab) * kSlabBytes, buf, static_cast<size_t>(amt));
        pending_.fetch_add(1, std::memory_order_acq_rel);
        {
            std::lock_guard<std::mutex> lk(ranges_m_);
            in_flight_.insert({off, amt});
            end_ = std::max(end_, off + amt);
        }
        Lane& lane = lanes_[static_cast<size_t>((off / kSlabBytes) % static_cast<sqlite3_int64>(lanes_.size()))];
        {
            std::lock_guard<std::mutex> lk(lane.m);
            lane.q.push_back({slab, amt, off});
        }
        lane.cv.notify_one();
        g_deferred_writes++;
        g_deferred_bytes += static_cast<unsigned long long>(amt);
        return SQLITE_OK;
    }
    // every write handed over so far is in the file (or has failed) when this returns
    int Drain() {
        g_drains++;
        if (pending_.load(std::memory_order_acquire) != 0) {
            g_drains_waited++;
            std::unique_lock<std::mutex> lk(idle_m_);
            idle_cv_.wait(lk, [&] { return pending_.load(std::memory_order_acquire) == 0; });
        }
        return error_.load(std::memory_order_acquire) ? SQLITE_IOERR_WRITE : SQLITE_OK;
    }
    // the writes that overlap [off, off + amt) are in the file (or have failed) when this returns; the others stay in flight --
    // SQLite re-reads pages its small cache has dropped (b-tree and pointer-map pages written long ago) all the time
    int WaitFor(sqlite3_int64 off, int amt) {
        if (pending_.load(std::memory_order_acquire) != 0) {
            std::unique_lock<std::mutex> lk(ranges_m_);
            auto overlaps = [&] {
                // entries are keyed by offset and at most kSlabBytes long
                for (auto it = in_flight_.lower_bound({off - kSlabBytes, 0}); it != in_flight_.end() && it->first < off + amt; ++it)
                    if (it->first + it->second > off) return true;
                return false;
            };
            if (overlaps()) {
                g_drains_waited++;
                ranges_cv_.wait(lk, [&] { return !overlaps(); });
            }
        }
        return error_.load(std::memory_order_acquire) ? SQLITE_IOERR_WRITE : SQLITE_OK;
    }
    // one past the last byte ever handed over (the size of the file once everything pending is in it, unless it was longer)
    sqlite3_int64 End() {
        std::lock_guard<std::mutex> lk(ranges_m_);
        return end_;
    }
    void ForgetEnd() {
        std::lock_guard<std::mutex> lk(ranges_m_);
        end_ = 0;
    }

   private:
    struct Job {
        int slab, amt;
        sqlite3_int64 off;
    };
    struct Lane {
        std::mutex m;
        std::condition_variable cv;
        std::deque<Job> q;
        bool stop = false;
        std::thread thread;
    };
    void Run(Lane& lane) {
        numa::PinThisThreadNearGpu(nullptr, "analysis: database page writer");
        for (;;) {
            Job job;
            {
                std::unique_lock<std::mutex> lk(lane.m);
                lane.cv.wait(lk, [&] { return lane.stop || !lane.q.empty(); });
                if (lane.q.empty()) return;
                job = lane.q.front();
                lane.q.pop_front();
            }
            // the default VFS writes with pwrite(2) on the descriptor of `real_`: safe from several threads for disjoint
            // ranges; one page never has two writes in flight (same lane, in order)
            const int rc = real_->pMethods->xWrite(real_, slabs_ + static_cast<size_t>(job.slab) * kSlabBytes, job.amt, job.off);
            if (rc != SQLITE_OK) error_.store(rc, std::memory_order_release);
            {
                std::lock_guard<std::mutex> lk(ranges_m_);
                in_flight_.erase(in_flight_.find({job.off, job.amt}));
            }
            ranges_cv_.notify_all();
            {
                std::lock_guard<std::mutex> lk(free_m_);
                free_.push_back(job.slab);
            }
            free_cv_.notify_one();
            if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lk(idle_m_);
                idle_cv_.notify_all();
            }
        }
    }

    sqlite3_file* real_;
    std::vector<Lane> lanes_;
    char* slabs_ = nullptr;
    size_t slab_bytes_ = 0;
    std::mutex free_m_;
    std::condition_variable free_cv_;
    std::vector<int> free_;
    std::atomic<long> pending_{0};
    std::atomic<int> error_{0};
    std::mutex idle_m_;
    std::condition_variable idle_cv_;
    std::mutex ranges_m_;
    std::condition_variable ranges_cv_;
    std::multiset<std::pair<sqlite3_int64, int>> in_flight_;   // (offset, bytes) of the writes handed over and not yet carried out
    sqlite3_int64 end_ = 0;
};

// What the files of one database share: the main file's writer (while that file is open) and whether the database is in
// WAL mode right now (a -wal file is open: checkpoints then write the main file without a commit point this VFS could
// hold back, so nothing is deferred).
struct DbState {
    std::mutex m;
    Writer* writer = nullptr;
    std::atomic<int> wal_open{0};
    int Drain() {
        std::lock_guard<std::mutex> lk(m);
        return writer ? writer->Drain() : SQLITE_OK;
    }
};

std::mutex g_registry_m;
std::map<std::string, std::shared_ptr<DbState>> g_registry;   // by the main file's full path

enum class Kind { kMain, kJournal, kWal, kOther };

struct File {
    sqlite3_file base;
    sqlite3_file* real;
    Kind kind;
    std::shared_ptr<DbState>* state;   // heap-allocated: SQLite owns this struct's memory as plain bytes
    Writer* writer;                     // kMain with deferral: owned here
    std::string* path;                  // kMain: the registry key
};

File* Self(sqlite3_file* f) { return reinterpret_cast<File*>(f); }
int DrainDb(File* f) { return (f->state && *f->state) ? (*f->state)->Drain() : SQLITE_OK; }

int XClose(sqlite3_file* file) {
    File* f = Self(file);
    int rc = SQLITE_OK;
    if (f->kind == Kind::kMain && f->writer) {
        rc = f->writer->Drain();
        {
            std::lock_guard<std::mutex> lk((*f->state)->m);
            (*f->state)->writer = nullptr;
        }
        delete f->writer;
        f->writer = nullptr;
        std::lock_guard<std::mutex> lk(g_registry_m);
        g_registry.erase(*f->path);
    } else if (f->kind == Kind::kJournal) {
        rc = DrainDb(f);   // a journal that goes away is a commit point
    } else if (f->kind == Kind::kWal && f->state && *f->state) {
        (*f->state)->wal_open--;
    }
    const int rc2 = f->real->pMethods ? f->real->pMethods->xClose(f->real) : SQLITE_OK;
    delete f->state;
    delete f->path;
    f->state = nullptr;
    f->path = nullptr;
    return rc != SQLITE_OK ? rc : rc2;
}

int XRead(sqlite3_file* file, void* buf, int amt, sqlite3_int64 off) {
    File* f = Self(file);
    if (f->writer) {
        const int rc = f->writer->WaitFor(off, amt);
        if (rc != SQLITE_OK) return rc;
    }
    return f->real->pMethods->xRead(f->real, buf, amt, off);
}

int XWrite(sqlite3_file* file, const void* buf, int amt, sqlite3_int64 off) {
    File* f = Self(file);
    if (f->kind == Kind::kMain && f->writer) {
        if (amt <= kSlabBytes && (*f->state)->wal_open.load() == 0) return f->writer->Write(buf, amt, off);
        const int rc = f->writer->Drain();
        if (rc != SQLITE_OK) return rc;
    } else if (f->kind == Kind::kJournal && off == 0) {
        const int rc = DrainDb(f);   // the journal header is (re)written at the points that begin or end a transaction
        if (rc != SQLITE_OK) return rc;
    }
    g_direct_writes++;
    return f->real->pMethods->xWrite(f->real, buf, amt, off);
}

int XTruncate(sqlite3_file* file, sqlite3_int64 size) {
    File* f = Self(file);
    const int rc = (f->kind == Kind::kMain && f->writer) ? f->writer->Drain() : (f->kind == Kind::kJournal ? DrainDb(f) : SQLITE_OK);
    if (rc != SQLITE_OK) return rc;   // journal_mode=TRUNCATE commits by truncating the journal: not before the pages are in
    if (f->kind == Kind::kMain && f->writer) f->writer->ForgetEnd();   // drained: the file's own size is the truth again
    return f->real->pMethods->xTruncate(f->real, size);
}

int XSync(sqlite3_file* file, int flags) {
    File* f = Self(file);
    const int rc = (f->kind == Kind::kMain && f->writer) ? f->writer->Drain() : (f->kind == Kind::kJournal ? DrainDb(f) : SQLITE_OK);
    if (rc != SQLITE_OK) return rc;
    return f->real->pMethods->xSync(f->real, flags);
}

int XFileSize(sqlite3_file* file, sqlite3_int64* size) {
    File* f = Self(file);
    const int rc = f->real->pMethods->xFileSize(f->real, size);
    if (rc == SQLITE_OK && f->writer) *size = std::max(*size, f->writer->End());   // what it will be once the pending pages are in
    return rc;
}

int XLock(sqlite3_file* file, int level) { return Self(file)->real->pMethods->xLock(Self(file)->real, level); }

int XUnlock(sqlite3_file* file, int level) {
    File* f = Self(file);
    int rc = SQLITE_OK;
    if (f->writer && level < SQLITE_LOCK_RESERVED) rc = f->writer->Drain();   // others may read the file from here on
    const int rc2 = f->real->pMethods->xUnlock(f->real, level);
    return rc != SQLITE_OK ? rc : rc2;
}

int XCheckReservedLock(sqlite3_file* file, int* out) { return Self(file)->real->pMethods->xCheckReservedLock(Self(file)->real, out); }
int XFileControl(sqlite3_file* file, int op, void* arg) { return Self(file)->real->pMethods->xFileControl(Self(file)->real, op, arg); }
int XSectorSize(sqlite3_file* file) { return Self(file)->real->pMethods->xSectorSize(Self(file)->real); }
// Without SQLITE_IOCAP_BATCH_ATOMIC: on a file system that offers it (F2FS), SQLite commits with the
// BEGIN / COMMIT_ATOMIC_WRITE file controls and no journal -- controls that would pass straight to the real file while the page
// writes they cover are still queued here.  Masked out, SQLite takes the rollback-journal path this VFS is written for.
int XDeviceCharacteristics(sqlite3_file* file) {
    return Self(file)->real->pMethods->xDeviceCharacteristics(Self(file)->real) & ~SQLITE_IOCAP_BATCH_ATOMIC;
}
int XShmMap(sqlite3_file* file, int page, int page_size, int extend, void volatile** out) {
    return Self(file)->real->pMethods->xShmMap(Self(file)->real, page, page_size, extend, out);
}
int XShmLock(sqlite3_file* file, int offset, int n, int flags) { return Self(file)->real->pMethods->xShmLock(Self(file)->real, offset, n, flags); }
void XShmBarrier(sqlite3_file* file) { Self(file)->real->pMethods->xShmBarrier(Self(file)->real); }
int XShmUnmap(sqlite3_file* file, int delete_flag) { return Self(file)->real->pMethods->xShmUnmap(Self(file)->real, delete_flag); }

const sqlite3_io_methods kMethodsV1 = {1,     XClose,  XRead,   XWrite,  XTruncate,          XSync,        XFileSize,
                                       XLock, XUnlock, XCheckReservedLock, XFileControl, XSectorSize, XDeviceCharacteristics,
                                       nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
const sqlite3_io_methods kMethodsV2 = {2,     XClose,  XRead,   XWrite,  XTruncate,          XSync,        XFileSize,
                                       XLock, XUnlock, XCheckReservedLock, XFileControl, XSectorSize, XDeviceCharacteristics,
                                       XShmMap, XShmLock, XShmBarrier, XShmUnmap, nullptr, nullptr};

// ---- the VFS ----------------------------------------------------------------------------------------------------------------
sqlite3_vfs* Parent(sqlite3_vfs* v) { return static_cast<sqlite3_vfs*>(v->pAppData); }

bool EndsWith(const std::string& s, const char* suffix, std::string* stem) {
    const size_t n = std::strlen(suffix);
    if (s.size() < n || s.compare(s.size() - n, n, suffix) != 0) return false;
    *stem = s.substr(0, s.size() - n);
    return true;
}

std::shared_ptr<DbState> Lookup(const std::string& main_path) {
    std::lock_guard<std::mutex> lk(g_registry_m);
    auto it = g_registry.find(main_path);
    return it == g_registry.end() ? nullptr : it->second;
}

int VOpen(sqlite3_vfs* v, const char* name, sqlite3_file* file, int flags, int* out_flags) {
    File* f = Self(file);
    f->base.pMethods = nullptr;
    f->real = reinterpret_cast<sqlite3_file*>(f + 1);
    f->kind = Kind::kOther;
    f->state = nullptr;
    f->writer = nullptr;
    f->path = nullptr;
    const int rc = Parent(v)->xOpen(Parent(v), name, f->real, flags, out_flags);
    if (rc != SQLITE_OK || !f->real->pMethods) return rc;
    try {
        std::string stem;
        if ((flags & SQLITE_OPEN_MAIN_DB) && name) {
            f->kind = Kind::kMain;
            // one worker: buffered writes to one file serialise on the inode's lock, so more threads only queue up behind each
            // other (measured: 1 thread +24 % on C2 and +47 % on C3 end to end, 2-8 threads no better than none) -- what is won
            // is the overlap of SQLite's own copying with the kernel's
            int lanes = 1;
            if (const char* env = std::getenv("POLYCHASE_DB_WRITE_THREADS")) lanes = std::max(0, std::min(16, std::atoi(env)));
            std::lock_guard<std::mutex> lk(g_registry_m);
            if (lanes > 0 && g_registry.find(name) == g_registry.end()) {   // a second connection on the same file: straight through
                f->writer = new Writer(f->real, lanes, 1024);   // 64 MiB of pages in flight at most
                f->path = new std::string(name);
                f->state = new std::shared_ptr<DbState>(std::make_shared<DbState>());
                (*f->state)->writer = f->writer;
                g_registry[*f->path] = *f->state;
            }
        } else if ((flags & SQLITE_OPEN_MAIN_JOURNAL) && name && EndsWith(name, "-journal", &stem)) {
            f->kind = Kind::kJournal;
            if (auto st = Lookup(stem)) f->state = new std::shared_ptr<DbState>(st);
        } else if ((flags & SQLITE_OPEN_WAL) && name && EndsWith(name, "-wal", &stem)) {
            f->kind = Kind::kWal;
            if (auto st = Lookup(stem)) {
                st->wal_open++;
                const int drained = st->Drain();   // from here on the main file is written directly
                f->state = new std::shared_ptr<DbState>(st);
                if (drained != SQLITE_OK) {
                    XClose(file);
                    return drained;
                }
            }
        }
    } catch (...) {
        delete f->writer;
        delete f->path;
        delete f->state;
        f->real->pMethods->xClose(f->real);
        return SQLITE_NOMEM;
    }
    f->base.pMethods = f->real->pMethods->iVersion >= 2 ? &kMethodsV2 : &kMethodsV1;
    return SQLITE_OK;
}

int VDelete(sqlite3_vfs* v, const char* name, int sync_dir) {
    std::string stem;
    if (name && EndsWith(name, "-journal", &stem))   // journal_mode=DELETE commits by deleting the journal
        if (auto st = Lookup(stem)) {
            const int rc = st->Drain();
            if (rc != SQLITE_OK) return rc;
        }
    return Parent(v)->xDelete(Parent(v), name, sync_dir);
}
int VAccess(sqlite3_vfs* v, const char* name, int flags, int* out) { return Parent(v)->xAccess(Parent(v), name, flags, out); }
int VFullPathname(sqlite3_vfs* v, const char* name, int n, char* out) { return Parent(v)->xFullPathname(Parent(v), name, n, out); }
void* VDlOpen(sqlite3_vfs* v, const char* name) { return Parent(v)->xDlOpen(Parent(v), name); }
void VDlError(sqlite3_vfs* v, int n, char* msg) { Parent(v)->xDlError(Parent(v), n, msg); }
void (*VDlSym(sqlite3_vfs* v, void* h, const char* sym))(void) { return Parent(v)->xDlSym(Parent(v), h, sym); }
void VDlClose(sqlite3_vfs* v, void* h) { Parent(v)->xDlClose(Parent(v), h); }
int VRandomness(sqlite3_vfs* v, int n, char* out) { return Parent(v)->xRandomness(Parent(v), n, out); }
int VSleep(sqlite3_vfs* v, int us) { return Parent(v)->xSleep(Parent(v), us); }
int VCurrentTime(sqlite3_vfs* v, double* t) { return Parent(v)->xCurrentTime(Parent(v), t); }
int VGetLastError(sqlite3_vfs* v, int n, char* msg) { return Parent(v)->xGetLastError(Parent(v), n, msg); }
int VCurrentTimeInt64(sqlite3_vfs* v, sqlite3_int64* t) { return Parent(v)->xCurrentTimeInt64(Parent(v), t); }

}  // namespace

const char* AsyncWriteVfsName() {
    static const char* name = [] () -> const char* {
        sqlite3_vfs* parent = sqlite3_vfs_find(nullptr);
        if (!parent || parent->iVersion < 2) return nullptr;
        static sqlite3_vfs vfs;
        std::memset(&vfs, 0, sizeof vfs);
        vfs.iVersion = 2;
        vfs.szOsFile = static_cast<int>(sizeof(File)) + parent->szOsFile;
        vfs.mxPathname = parent->mxPathname;
        vfs.zName = "polychase_async_write";
        vfs.pAppData = parent;
        vfs.xOpen = VOpen;
        vfs.xDelete = VDelete;
        vfs.xAccess = VAccess;
        vfs.xFullPathname = VFullPathname;
        vfs.xDlOpen = VDlOpen;
        vfs.xDlError = VDlError;
        vfs.xDlSym = VDlSym;
        vfs.xDlClose = VDlClose;
        vfs.xRandomness = VRandomness;
        vfs.xSleep = VSleep;
        vfs.xCurrentTime = VCurrentTime;
        vfs.xGetLastError = VGetLastError;
        vfs.xCurrentTimeInt64 = VCurrentTimeInt64;
        return sqlite3_vfs_register(&vfs, 0) == SQLITE_OK ? vfs.zName : nullptr;
    }();
    return name;
}

AsyncWriteVfsCounters AsyncWriteVfsTotals() {
    AsyncWriteVfsCounters c;
    c.deferred_writes = g_deferred_writes.load();
    c.deferred_bytes = g_deferred_bytes.load();
    c.direct_writes = g_direct_writes.load();
    c.drains = g_drains.load();
    c.drains_that_waited = g_drains_waited.load();
    c.waits_for_a_slab = g_slab_waits.load();
    return c;
}
