// flow_database.cc -- see flow_database.h.  The SQL text (schema, pragmas, the ten statements) is the
// file format and therefore identical to the reference (cpp/database.cc:76-135, :353-399).
#include "flow_database.h"

#include "async_write_vfs.h"

#include <cctype>

#include <cstdlib>

#include <cstring>
#include <stdexcept>
#include <utility>

#include "utils.h"

namespace {

// error text of the reference: "SQLite error [file:line]: message" (cpp/database.cc:12-36)
[[noreturn]] void Fail(int line, const char* msg) {
    throw std::runtime_error(StrFormat("SQLite error [%s:%d]: %s", __FILE__, line, msg ? msg : "Unknown error"));
}

int Ok(int rc, int line) {
    if (rc != SQLITE_OK && rc != SQLITE_ROW && rc != SQLITE_DONE) Fail(line, sqlite3_errstr(rc));
    return rc;
}
#define SQL_OK(expr) Ok((expr), __LINE__)

constexpr const char* kSchema[] = {
    R"(
        CREATE TABLE IF NOT EXISTS keypoints(
            image_id   INTEGER  PRIMARY KEY  NOT NULL,
            rows       INTEGER               NOT NULL,
            keypoints  BLOB                  NOT NULL
        );
    )",
    R"(
        CREATE TABLE IF NOT EXISTS optical_flow(
            image_id_from           INTEGER  NOT NULL,
            image_id_to             INTEGER  NOT NULL,
            rows                    INTEGER  NOT NULL,
            src_keypoints_indices   BLOB     NOT NULL,
            tgt_keypoints           BLOB     NOT NULL,
            flow_errors             BLOB     NOT NULL,
            PRIMARY KEY(image_id_from, image_id_to),
            FOREIGN KEY(image_id_from) REFERENCES keypoints(image_id) ON DELETE CASCADE
        );
    )",
};

// same order as the reference's Open(): auto_vacuum comes after journal_mode and is therefore inert
constexpr const char* kPragmas[] = {
    "PRAGMA synchronous=OFF", "PRAGMA journal_mode=WAL", "PRAGMA temp_store=MEMORY",
    "PRAGMA foreign_keys=ON", "PRAGMA auto_vacuum=1",
};

// indexed by Database::Statement
constexpr const char* kStatementSql[] = {
    "SELECT rows, keypoints FROM keypoints WHERE image_id = ?;",
    "INSERT INTO keypoints(image_id, rows, keypoints) VALUES(?, ?, ?);",
    "SELECT rows, src_keypoints_indices, tgt_keypoints, flow_errors FROM optical_flow WHERE image_id_from = ? AND "
    "image_id_to = ?;",
    "SELECT rows, src_keypoints_indices, tgt_keypoints FROM optical_flow WHERE image_id_from = ? AND image_id_to = ?;",
    "INSERT INTO optical_flow(image_id_from, image_id_to, rows, src_keypoints_indices, tgt_keypoints, "
    "flow_errors) VALUES(?, ?, ?, ?, ?, ?);",
    "SELECT image_id_to FROM optical_flow WHERE image_id_from = ?",
    "SELECT image_id_from FROM optical_flow WHERE image_id_to = ?",
    "SELECT 1 FROM keypoints WHERE image_id = ?;",
    "SELECT 1 FROM optical_flow WHERE image_id_from = ? AND image_id_to = ?;",
    "SELECT MIN(image_id) FROM keypoints;",
    "SELECT MAX(image_id) FROM keypoints;",
};

// RAII reset: statements are reused, every exit path (including exceptions) must reset them
struct Resetter {
    sqlite3_stmt* s;
    ~Resetter() { sqlite3_reset(s); }
};

template <typename T>
void BlobToVector(sqlite3_stmt* stmt, int col, size_t rows, std::vector<T>& out) {
    out.assign(rows, T{});
    const size_t bytes = static_cast<size_t>(sqlite3_column_bytes(stmt, col));
    CHECK_EQ(out.size() * sizeof(T), bytes);
    if (bytes) std::memcpy(out.data(), sqlite3_column_blob(stmt, col), bytes);
}

// a NULL pointer would bind SQL NULL and trip NOT NULL: empty blobs bind a valid dummy address
void BindBlob(sqlite3_stmt* stmt, int col, const void* data, size_t bytes, int line) {
    static const char kNothing = 0;
    Ok(sqlite3_bind_blob(stmt, col, bytes ? data : &kNothing, static_cast<int>(bytes), SQLITE_STATIC), line);
}

}  // namespace

Database::Database(Database&& other) noexcept : db_(std::exchange(other.db_, nullptr)), path_(std::move(other.path_)) {
    statements_ = other.statements_;
    other.statements_.fill(nullptr);
}

Database::~Database() {
    try {
        Close();
    } catch (...) {
    }
}

void Database::Exec(const char* sql, int line) const {
    char* err = nullptr;
    if (sqlite3_exec(db_, sql, nullptr, nullptr, &err) != SQLITE_OK) {
        const std::string msg = err ? err : "Unknown error";
        sqlite3_free(err);
        Fail(line, msg.c_str());
    }
}

void Database::Open(const std::string& path, bool bulk_writer) {
    Close();
    path_ = path;
    // NOMUTEX like the reference: callers serialise access (one writer thread here)
    SQL_OK(sqlite3_open_v2(path.c_str(), &db_, SQLITE_OPEN_READWRITE | SQLITE_OPEN_CREATE | SQLITE_OPEN_NOMUTEX,
                           bulk_writer ? AsyncWriteVfsName() : nullptr));
    // another connection may be in the middle of a commit (the analysis bulk-loads under a rollback journal, so a reader
    // that opens the file meanwhile meets a locked database): wait instead of failing with SQLITE_BUSY
    sqlite3_busy_timeout(db_, 10000);
    // Not in the reference: NEW databases are created with 64 KiB pages (the blobs of one frame are ~5 MB: with SQLite's
    // default of 4 KiB most of the insert time goes into overflow-page chains; C2 with the insert in the loop: 425 -> 890
    // frames/s).  The page size is a property of the file -- the reference reads and appends to such a database
    // unchanged, and an existing database keeps the size it was created with.  POLYCHASE_DB_PAGE_SIZE=4096 gives the
    // reference's (SQLite's) default.
    {
        int v = 65536;
        if (const char* ps = std::getenv("POLYCHASE_DB_PAGE_SIZE")) v = std::atoi(ps);
        if (v >= 512 && v <= 65536 && (v & (v - 1)) == 0) Exec(("PRAGMA page_size=" + std::to_string(v)).c_str(), __LINE__);
    }
    for (const char* pragma : kPragmas) Exec(pragma, __LINE__);
    for (const char* table : kSchema) Exec(table, __LINE__);
    for (int i = 0; i < kNumStatements; i++) SQL_OK(sqlite3_prepare_v2(db_, kStatementSql[i], -1, &statements_[i], nullptr));
}

void Database::FinalizeAll() {
    for (sqlite3_stmt*& s : statements_) {
        if (s) sqlite3_finalize(s);
        s = nullptr;
    }
}

void Database::Close() {
    if (!db_) return;
    FinalizeAll();
    sqlite3_close_v2(db_);
    db_ = nullptr;
}

std::string Database::SetJournalMode(const char* mode) {
    sqlite3_stmt* stmt = nullptr;
    const std::string sql = mode[0] ? std::string("PRAGMA journal_mode=") + mode : std::string("PRAGMA journal_mode");
    SQL_OK(sqlite3_prepare_v2(db_, sql.c_str(), -1, &stmt, nullptr));
    std::string now;
    const int rc = sqlite3_step(stmt);
    if (rc == SQLITE_ROW) {
        if (const unsigned char* t = sqlite3_column_text(stmt, 0)) now = reinterpret_cast<const char*>(t);
    }
    sqlite3_finalize(stmt);
    if (rc == SQLITE_BUSY || rc == SQLITE_LOCKED) {
        // another connection holds the file: the mode stays what it is -- report that instead of failing
        if (!mode[0]) SQL_OK(rc);
        return SetJournalMode("");   // empty mode: query only
    }
    if (rc != SQLITE_ROW && rc != SQLITE_DONE) SQL_OK(rc);
    for (char& c : now) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
    return now;
}

void Database::Begin() { Exec("BEGIN", __LINE__); }
void Database::Commit() { Exec("COMMIT", __LINE__); }
void Database::Rollback() { Exec("ROLLBACK", __LINE__); }

// ---- generic helpers -------------------------------------------------------------------------
bool Database::HasRow(Statement s, int32_t key_a, const int32_t* key_b) const {
    sqlite3_stmt* stmt = Stmt(s);
    Resetter reset{stmt};
    SQL_OK(sqlite3_bind_int(stmt, 1, key_a));
    if (key_b) SQL_OK(sqlite3_bind_int(stmt, 2, *key_b));
    return SQL_OK(sqlite3_step(stmt)) == SQLITE_ROW;
}

void Database::CollectIds(Statement s, int32_t key, std::vector<int32_t>& append_to) const {
    sqlite3_stmt* stmt = Stmt(s);
    Resetter reset{stmt};
    SQL_OK(sqlite3_bind_int(stmt, 1, key));
    while (SQL_OK(sqlite3_step(stmt)) == SQLITE_ROW) append_to.push_back(sqlite3_column_int(stmt, 0));
}

int32_t Database::ScalarOrInvalid(Statement s) const {
    sqlite3_stmt* stmt = Stmt(s);
    Resetter reset{stmt};
    // MIN()/MAX() over an empty table yield one NULL row, read as 0 -- same as the reference
    return SQL_OK(sqlite3_step(stmt)) == SQLITE_ROW ? sqlite3_column_int(stmt, 0) : kInvalidId;
}

// ---- keypoints -------------------------------------------------------------------------------
bool Database::KeypointsExist(int32_t image_id) const { return HasRow(kKeypointsExist, image_id, nullptr); }
int32_t Database::GetMinImageIdWithKeypoints() const { return ScalarOrInvalid(kMinImageId); }
int32_t Database::GetMaxImageIdWithKeypoints() const { return ScalarOrInvalid(kMaxImageId); }

void Database::ReadKeypoints(int32_t image_id, Keypoints& out) const {
    sqlite3_stmt* stmt = Stmt(kReadKeypoints);
    Resetter reset{stmt};
    SQL_OK(sqlite3_bind_int(stmt, 1, image_id));
    if (SQL_OK(sqlite3_step(stmt)) != SQLITE_ROW) return;
    const int rows = sqlite3_column_int(stmt, 0);
    CHECK(rows >= 0);
    BlobToVector(stmt, 1, static_cast<size_t>(rows), out);
}

bool Database::ReadKeypointsInto(int32_t image_id, const std::function<float*(size_t rows)>& place, size_t* rows_out) const {
    sqlite3_stmt* stmt = Stmt(kReadKeypoints);
    Resetter reset{stmt};
    SQL_OK(sqlite3_bind_int(stmt, 1, image_id));
    if (SQL_OK(sqlite3_step(stmt)) != SQLITE_ROW) return false;
    const int rows = sqlite3_column_int(stmt, 0);
    CHECK(rows >= 0);
    const size_t bytes = static_cast<size_t>(sqlite3_column_bytes(stmt, 1));
    CHECK_EQ(static_cast<size_t>(rows) * sizeof(Keypoint), bytes);
    float* dst = place(static_cast<size_t>(rows));
    if (bytes) std::memcpy(dst, sqlite3_column_blob(stmt, 1), bytes);
    if (rows_out) *rows_out = static_cast<size_t>(rows);
    return true;
}

bool Database::VisitKeypoints(int32_t image_id, const std::function<void(size_t rows, const void* xy)>& visit) const {
    sqlite3_stmt* stmt = Stmt(kReadKeypoints);
    Resetter reset{stmt};
    SQL_OK(sqlite3_bind_int(stmt, 1, image_id));
    if (SQL_OK(sqlite3_step(stmt)) != SQLITE_ROW) return false;
    const int rows = sqlite3_column_int(stmt, 0);
    CHECK(rows >= 0);
    const void* blob = sqlite3_column_blob(stmt, 1);
    CHECK_EQ(static_cast<size_t>(rows) * sizeof(Keypoint), static_cast<size_t>(sqlite3_column_bytes(stmt, 1)));
    visit(static_cast<size_t>(rows), blob);
    return true;
}

Keypoints Database::ReadKeypoints(int32_t image_id) const {
    Keypoints k;
    ReadKeypoints(image_id, k);
    return k;
}

void Database::WriteKeypoints(int32_t image_id, const float* xy, size_t rows) {
    sqlite3_stmt* stmt = Stmt(kWriteKeypoints);
    Resetter reset{stmt};
    SQL_OK(sqlite3_bind_int(stmt, 1, image_id));
    SQL_OK(sqlite3_bind_int(stmt, 2, static_cast<int>(rows)));
    BindBlob(stmt, 3, xy, rows * sizeof(Keypoint), __LINE__);
    SQL_OK(sqlite3_step(stmt));  // plain INSERT: a second write of the same image_id throws
}

void Database::WriteKeypoints(int32_t image_id, const Keypoints& keypoints) {
    WriteKeypoints(image_id, keypoints.empty() ? nullptr : keypoints.front().data(), keypoints.size());
}

// ---- optical_flow ----------------------------------------------------------------------------
bool Database::ImagePairFlowExists(int32_t from, int32_t to) const { return HasRow(kFlowExists, from, &to); }

void Database::ReadImagePairFlow(int32_t from, int32_t to, ImagePairFlow& out) const {
    sqlite3_stmt* stmt = Stmt(kReadFlow);
    Resetter reset{stmt};
    SQL_OK(sqlite3_bind_int(stmt, 1, from));
    SQL_OK(sqlite3_bind_int(stmt, 2, to));
    if (SQL_OK(sqlite3_step(stmt)) != SQLITE_ROW) return;
    const size_t rows = static_cast<size_t>(sqlite3_column_int(stmt, 0));
    BlobToVector(stmt, 1, rows, out.src_kps_indices);
    BlobToVector(stmt, 2, rows, out.tgt_kps);
    BlobToVector(stmt, 3, rows, out.flow_errors);
    out.image_id_from = from;
    out.image_id_to = to;
}

void Database::ReadImagePairMatches(int32_t from, int32_t to, KeypointsIndices& idx, Keypoints& tgt) const {
    idx.clear();
    tgt.clear();
    sqlite3_stmt* stmt = Stmt(kReadMatches);
    Resetter reset{stmt};
    SQL_OK(sqlite3_bind_int(stmt, 1, from));
    SQL_OK(sqlite3_bind_int(stmt, 2, to));
    if (SQL_OK(sqlite3_step(stmt)) != SQLITE_ROW) return;
    const size_t rows = static_cast<size_t>(sqlite3_column_int(stmt, 0));
    BlobToVector(stmt, 1, rows, idx);
    BlobToVector(stmt, 2, rows, tgt);
}

bool Database::ReadImagePairMatchesInto(int32_t from, int32_t to,
                                        const std::function<void(size_t rows, uint32_t** idx, float** tgt_xy)>& place,
                                        size_t* rows_out) const {
    sqlite3_stmt* stmt = Stmt(kReadMatches);
    Resetter reset{stmt};
    SQL_OK(sqlite3_bind_int(stmt, 1, from));
    SQL_OK(sqlite3_bind_int(stmt, 2, to));
    if (SQL_OK(sqlite3_step(stmt)) != SQLITE_ROW) return false;
    const size_t rows = static_cast<size_t>(sqlite3_column_int(stmt, 0));
    const size_t idx_bytes = static_cast<size_t>(sqlite3_column_bytes(stmt, 1)), tgt_bytes = static_cast<size_t>(sqlite3_column_bytes(stmt, 2));
    CHECK_EQ(rows * sizeof(uint32_t), idx_bytes);
    CHECK_EQ(rows * sizeof(Keypoint), tgt_bytes);
    uint32_t* idx = nullptr;
    float* tgt = nullptr;
    place(rows, &idx, &tgt);
    if (idx_bytes) std::memcpy(idx, sqlite3_column_blob(stmt, 1), idx_bytes);
    if (tgt_bytes) std::memcpy(tgt, sqlite3_column_blob(stmt, 2), tgt_bytes);
    if (rows_out) *rows_out = rows;
    return true;
}

bool Database::VisitImagePairMatches(int32_t from, int32_t to,
                                     const std::function<void(size_t rows, const void* idx, const void* tgt_xy)>& visit) const {
    sqlite3_stmt* stmt = Stmt(kReadMatches);
    Resetter reset{stmt};
    SQL_OK(sqlite3_bind_int(stmt, 1, from));
    SQL_OK(sqlite3_bind_int(stmt, 2, to));
    if (SQL_OK(sqlite3_step(stmt)) != SQLITE_ROW) return false;
    const size_t rows = static_cast<size_t>(sqlite3_column_int(stmt, 0));
    // both pointers first, then the sizes: a later sqlite3_column_blob() may not move an earlier one's memory, but a type
    // conversion could -- there is none here (blob columns read as blobs)
    const void* idx = sqlite3_column_blob(stmt, 1);
    const void* tgt = sqlite3_column_blob(stmt, 2);
    CHECK_EQ(rows * sizeof(uint32_t), static_cast<size_t>(sqlite3_column_bytes(stmt, 1)));
    CHECK_EQ(rows * sizeof(Keypoint), static_cast<size_t>(sqlite3_column_bytes(stmt, 2)));
    visit(rows, idx, tgt);
    return true;
}

ImagePairFlow Database::ReadImagePairFlow(int32_t from, int32_t to) const {
    ImagePairFlow f;
    ReadImagePairFlow(from, to, f);
    return f;
}

void Database::WriteImagePairFlow(int32_t from, int32_t to, const uint32_t* idx, const float* tgt_xy, const float* err,
                                  size_t rows) {
    sqlite3_stmt* stmt = Stmt(kWriteFlow);
    Resetter reset{stmt};
    SQL_OK(sqlite3_bind_int(stmt, 1, from));
    SQL_OK(sqlite3_bind_int(stmt, 2, to));
    SQL_OK(sqlite3_bind_int(stmt, 3, static_cast<int>(rows)));
    BindBlob(stmt, 4, idx, rows * sizeof(uint32_t), __LINE__);
    BindBlob(stmt, 5, tgt_xy, rows * sizeof(Keypoint), __LINE__);
    BindBlob(stmt, 6, err, rows * sizeof(float), __LINE__);
    SQL_OK(sqlite3_step(stmt));
}

void Database::WriteImagePairFlow(int32_t from, int32_t to, const KeypointsIndices& idx, const Keypoints& tgt,
                                  const FlowErrors& err) {
    CHECK_EQ(tgt.size(), idx.size());
    CHECK_EQ(err.size(), idx.size());
    WriteImagePairFlow(from, to, idx.data(), tgt.empty() ? nullptr : tgt.front().data(), err.data(), idx.size());
}

void Database::WriteImagePairFlow(const ImagePairFlow& f) {
    WriteImagePairFlow(f.image_id_from, f.image_id_to, f.src_kps_indices, f.tgt_kps, f.flow_errors);
}

void Database::FindOpticalFlowsFromImage(int32_t from, std::vector<int32_t>& append_to) const {
    CollectIds(kFlowsFrom, from, append_to);
}
void Database::FindOpticalFlowsToImage(int32_t to, std::vector<int32_t>& append_to) const {
    CollectIds(kFlowsTo, to, append_to);
}
std::vector<int32_t> Database::FindOpticalFlowsFromImage(int32_t from) const {
    std::vector<int32_t> ids;
    CollectIds(kFlowsFrom, from, ids);
    return ids;
}
std::vector<int32_t> Database::FindOpticalFlowsToImage(int32_t to) const {
    std::vector<int32_t> ids;
    CollectIds(kFlowsTo, to, ids);
    return ids;
}
