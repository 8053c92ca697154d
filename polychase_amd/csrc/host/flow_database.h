// flow_database.h -- the on-disk optical-flow database (SQLite).  File format identical to the
// reference's (cpp/database.h:36-100, cpp/database.cc:64-135, :350-400): same two tables, same
// pragmas in the same order, same statements, blobs = raw little-endian arrays -- databases written
// by either implementation are interchangeable and a cancelled analysis can be resumed by the other.
// The class keeps the reference's public method names because the Python surface
// (`polychase_core.Database`, cpp/polychase_pybind.cc:71-109) is built from them.
#pragma once

#include <sqlite3.h>

#include <array>
#include <cstdint>
#include <functional>
#include <limits>
#include <string>
#include <vector>

static constexpr int32_t kInvalidId = std::numeric_limits<int32_t>::max();

using Keypoint = std::array<float, 2>;  // one (x, y) pair, 8 bytes in the blob
using Keypoints = std::vector<Keypoint>;
using KeypointsIndices = std::vector<uint32_t>;
using FlowErrors = std::vector<float>;

// One row of `optical_flow`: matches of frame image_id_from tracked into image_id_to.
struct ImagePairFlow {
    int32_t image_id_from = 0;
    int32_t image_id_to = 0;
    KeypointsIndices src_kps_indices;  // ascending indices into the keypoints of image_id_from
    Keypoints tgt_kps;
    FlowErrors flow_errors;
    void Clear() {
        src_kps_indices.clear();
        tgt_kps.clear();
        flow_errors.clear();
    }
};

class Database {
   public:
    // bulk_writer (not in the reference): the connection of the analysis that inserts the rows -- its page writes to the
    // database file are carried out by worker threads (async_write_vfs.h; POLYCHASE_DB_WRITE_THREADS=0: like any other)
    explicit Database(const std::string& path, bool bulk_writer = false) { Open(path, bulk_writer); }
    Database(Database&& other) noexcept;
    Database(const Database&) = delete;
    Database& operator=(const Database&) = delete;
    ~Database();

    void Open(const std::string& path, bool bulk_writer = false);
    void Close();
    const std::string& Path() const { return path_; }   // what Open() was given (a second read connection can be opened on it)

    // ---- keypoints(image_id PK, rows, keypoints BLOB) ----
    bool KeypointsExist(int32_t image_id) const;
    Keypoints ReadKeypoints(int32_t image_id) const;
    void ReadKeypoints(int32_t image_id, Keypoints& out) const;  // leaves `out` untouched if absent
    // Not in the reference: the blob straight into memory the caller chooses once the row count is known (the tracker's
    // page-locked staging buffers: one copy from SQLite's page cache instead of two).  false if the row is absent.
    bool ReadKeypointsInto(int32_t image_id, const std::function<float*(size_t rows)>& place, size_t* rows_out) const;
    // ... or not copied at all: `visit` sees the blob where SQLite holds it (valid during the call only; no alignment promised:
    // read it with memcpy).  The refiner filters a frame's keypoints straight out of it.
    bool VisitKeypoints(int32_t image_id, const std::function<void(size_t rows, const void* xy)>& visit) const;
    void WriteKeypoints(int32_t image_id, const Keypoints& keypoints);
    void WriteKeypoints(int32_t image_id, const float* xy, size_t rows);
    int32_t GetMinImageIdWithKeypoints() const;
    int32_t GetMaxImageIdWithKeypoints() const;

    // ---- optical_flow(image_id_from, image_id_to, rows, 3 BLOBs) ----
    bool ImagePairFlowExists(int32_t image_id_from, int32_t image_id_to) const;
    ImagePairFlow ReadImagePairFlow(int32_t image_id_from, int32_t image_id_to) const;
    void ReadImagePairFlow(int32_t image_id_from, int32_t image_id_to, ImagePairFlow& out) const;
    // Not in the reference: the two columns the tracker consumes (tracker.cc:56-86), without flow_errors
    void ReadImagePairMatches(int32_t image_id_from, int32_t image_id_to, KeypointsIndices& src_kps_indices,
                              Keypoints& tgt_kps) const;
    // ... and straight into caller-chosen memory: place(rows, &idx, &tgt) names where the two columns go
    bool ReadImagePairMatchesInto(int32_t image_id_from, int32_t image_id_to,
                                  const std::function<void(size_t rows, uint32_t** idx, float** tgt_xy)>& place, size_t* rows_out) const;
    bool VisitImagePairMatches(int32_t image_id_from, int32_t image_id_to,
                               const std::function<void(size_t rows, const void* idx, const void* tgt_xy)>& visit) const;
    void WriteImagePairFlow(const ImagePairFlow& flow);
    void WriteImagePairFlow(int32_t image_id_from, int32_t image_id_to, const KeypointsIndices& src_kps_indices,
                            const Keypoints& tgt_kps, const FlowErrors& flow_errors);
    void WriteImagePairFlow(int32_t image_id_from, int32_t image_id_to, const uint32_t* idx, const float* tgt_xy,
                            const float* err, size_t rows);
    std::vector<int32_t> FindOpticalFlowsFromImage(int32_t image_id_from) const;
    void FindOpticalFlowsFromImage(int32_t image_id_from, std::vector<int32_t>& append_to) const;
    std::vector<int32_t> FindOpticalFlowsToImage(int32_t image_id_to) const;
    void FindOpticalFlowsToImage(int32_t image_id_to, std::vector<int32_t>& append_to) const;

    // Not in the reference: explicit transactions, so one frame's rows go in as one batch.
    void Begin();
    void Commit();
    void Rollback();
    // Not in the reference: `PRAGMA journal_mode=<mode>`; returns the mode now in effect (lower case).  The analysis
    // driver loads its rows under a rollback journal ("truncate") and puts the file back into WAL mode -- what the
    // reference's Open() sets (cpp/database.cc:88-92) -- when it is done: with 5 MB of blobs per frame, WAL mode
    // writes every page twice (log, then checkpoint), a rollback journal writes new pages once.
    std::string SetJournalMode(const char* mode);

   private:
    enum Statement {
        kReadKeypoints,
        kWriteKeypoints,
        kReadFlow,
        kReadMatches,
        kWriteFlow,
        kFlowsFrom,
        kFlowsTo,
        kKeypointsExist,
        kFlowExists,
        kMinImageId,
        kMaxImageId,
        kNumStatements
    };
    sqlite3_stmt* Stmt(Statement s) const { return statements_[s]; }
    void Exec(const char* sql, int line) const;
    void FinalizeAll();
    void CollectIds(Statement s, int32_t key, std::vector<int32_t>& append_to) const;
    bool HasRow(Statement s, int32_t key_a, const int32_t* key_b) const;
    int32_t ScalarOrInvalid(Statement s) const;

    sqlite3* db_ = nullptr;
    std::string path_;
    std::array<sqlite3_stmt*, kNumStatements> statements_{};
};
