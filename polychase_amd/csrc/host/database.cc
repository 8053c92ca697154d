// database.cc -- see database.h.  Schema / pragmas / statements: reference cpp/database.cc:64-135,
// :350-400; blobs are raw memcpy of the vectors (:137-158).
#include "database.h"

#include <cstring>
#include <stdexcept>
#include <utility>

#include "utils.h"

namespace {

[[noreturn]] void ThrowSqlite(const char* file, int line, const char* msg) {
    throw std::runtime_error(StrFormat("SQLite error [%s:%d]: %s", file, line, msg ? msg : "Unknown error"));
}

int SqliteCall(int rc, const char* file, int line) {
    switch (rc) {
        case SQLITE_OK:
        case SQLITE_ROW:
        case SQLITE_DONE:
            return rc;
        default:
            ThrowSqlite(file, line, sqlite3_errstr(rc));
    }
}

void SqliteExec(sqlite3* db, const char* sql, const char* file, int line) {
    char* err = nullptr;
    const int rc = sqlite3_exec(db, sql, nullptr, nullptr, &err);
    if (rc != SQLITE_OK) {
        const std::string msg = err ? err : "Unknown error";
        sqlite3_free(err);
        ThrowSqlite(file, line, msg.c_str());
    }
}

#define SQLITE3_CALL(expr) SqliteCall((expr), __FILE__, __LINE__)
#define SQLITE3_EXEC(db, sql) SqliteExec((db), (sql), __FILE__, __LINE__)

template <typename T>
void ReadBlob(sqlite3_stmt* stmt, size_t rows, int col, std::vector<T>& vec) {
    vec.clear();
    vec.resize(rows);
    const size_t num_bytes = static_cast<size_t>(sqlite3_column_bytes(stmt, col));
    CHECK_EQ(vec.size() * sizeof(T), num_bytes);
    if (num_bytes) std::memcpy(reinterpret_cast<char*>(vec.data()), sqlite3_column_blob(stmt, col), num_bytes);
}

// sqlite3_bind_blob(NULL, 0) would bind SQL NULL and violate NOT NULL: bind a zero-length blob
void BindBlob(sqlite3_stmt* stmt, int col, const void* data, size_t bytes) {
    static const char kEmpty = 0;
    SQLITE3_CALL(sqlite3_bind_blob(stmt, col, bytes ? data : &kEmpty, static_cast<int>(bytes), SQLITE_STATIC));
}

}  // namespace

Database::Database(const std::string& path) { Open(path); }

Database::Database(Database&& o) noexcept {
    database_ = std::exchange(o.database_, nullptr);
    sql_stmt_read_keypoints_ = std::exchange(o.sql_stmt_read_keypoints_, nullptr);
    sql_stmt_write_keypoints_ = std::exchange(o.sql_stmt_write_keypoints_, nullptr);
    sql_stmt_read_image_pair_flows_ = std::exchange(o.sql_stmt_read_image_pair_flows_, nullptr);
    sql_stmt_write_image_pair_flows_ = std::exchange(o.sql_stmt_write_image_pair_flows_, nullptr);
    sql_stmt_find_flows_from_image_ = std::exchange(o.sql_stmt_find_flows_from_image_, nullptr);
    sql_stmt_find_flows_to_image_ = std::exchange(o.sql_stmt_find_flows_to_image_, nullptr);
    sql_stmt_keypoints_exist_ = std::exchange(o.sql_stmt_keypoints_exist_, nullptr);
    sql_stmt_pair_flow_exist_ = std::exchange(o.sql_stmt_pair_flow_exist_, nullptr);
    sql_stmt_min_image_id_ = std::exchange(o.sql_stmt_min_image_id_, nullptr);
    sql_stmt_max_image_id_ = std::exchange(o.sql_stmt_max_image_id_, nullptr);
}

Database::~Database() {
    try {
        Close();
    } catch (...) {
    }
}

void Database::Open(const std::string& path) {
    Close();
    SQLITE3_CALL(sqlite3_open_v2(path.c_str(), &database_,
                                 SQLITE_OPEN_READWRITE | SQLITE_OPEN_CREATE | SQLITE_OPEN_NOMUTEX, nullptr));
    SQLITE3_EXEC(database_, "PRAGMA synchronous=OFF");
    SQLITE3_EXEC(database_, "PRAGMA journal_mode=WAL");
    SQLITE3_EXEC(database_, "PRAGMA temp_store=MEMORY");
    SQLITE3_EXEC(database_, "PRAGMA foreign_keys=ON");
    SQLITE3_EXEC(database_, "PRAGMA auto_vacuum=1");
    CreateTables();
    PrepareSQLStatements();
}

void Database::Close() {
    if (database_ != nullptr) {
        FinalizeSQLStatements();
        sqlite3_close_v2(database_);
        database_ = nullptr;
    }
}

void Database::CreateTables() const {
    // identical text to the reference so that sqlite_master matches byte for byte
    const char* keypoints_sql = R"(
        CREATE TABLE IF NOT EXISTS keypoints(
            image_id   INTEGER  PRIMARY KEY  NOT NULL,
            rows       INTEGER               NOT NULL,
            keypoints  BLOB                  NOT NULL
        );
    )";
    SQLITE3_EXEC(database_, keypoints_sql);
    const char* flow_sql = R"(
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
    )";
    SQLITE3_EXEC(database_, flow_sql);
}

Keypoints Database::ReadKeypoints(int32_t image_id) const {
    Keypoints k;
    ReadKeypoints(image_id, k);
    return k;
}

void Database::ReadKeypoints(int32_t image_id, Keypoints& keypoints) const {
    sqlite3_stmt* stmt = sql_stmt_read_keypoints_;
    SQLITE3_CALL(sqlite3_bind_int(stmt, 1, image_id));
    const int rc = SQLITE3_CALL(sqlite3_step(stmt));
    if (rc != SQLITE_ROW) {
        SQLITE3_CALL(sqlite3_reset(stmt));
        return;
    }
    const int rows = sqlite3_column_int(stmt, 0);
    CHECK(rows >= 0);
    ReadBlob(stmt, static_cast<size_t>(rows), 1, keypoints);
    SQLITE3_CALL(sqlite3_reset(stmt));
}

void Database::WriteKeypoints(int32_t image_id, const Keypoints& keypoints) {
    WriteKeypoints(image_id, keypoints.empty() ? nullptr : keypoints[0].data(), keypoints.size());
}

void Database::WriteKeypoints(int32_t image_id, const float* xy, size_t rows) {
    sqlite3_stmt* stmt = sql_stmt_write_keypoints_;
    SQLITE3_CALL(sqlite3_bind_int(stmt, 1, image_id));
    SQLITE3_CALL(sqlite3_bind_int(stmt, 2, static_cast<int>(rows)));
    BindBlob(stmt, 3, xy, rows * 2 * sizeof(float));
    const int rc = sqlite3_step(stmt);
    sqlite3_reset(stmt);
    SQLITE3_CALL(rc);
}

void Database::WriteImagePairFlow(int32_t from, int32_t to, const KeypointsIndices& idx, const Keypoints& tgt,
                                  const FlowErrors& err) {
    const size_t rows = idx.size();
    CHECK_EQ(tgt.size(), rows);
    CHECK_EQ(err.size(), rows);
    WriteImagePairFlow(from, to, idx.data(), rows ? tgt[0].data() : nullptr, err.data(), rows);
}

void Database::WriteImagePairFlow(const ImagePairFlow& f) {
    WriteImagePairFlow(f.image_id_from, f.image_id_to, f.src_kps_indices, f.tgt_kps, f.flow_errors);
}

void Database::WriteImagePairFlow(int32_t from, int32_t to, const uint32_t* idx, const float* tgt_xy,
                                  const float* err, size_t rows) {
    sqlite3_stmt* stmt = sql_stmt_write_image_pair_flows_;
    SQLITE3_CALL(sqlite3_bind_int(stmt, 1, from));
    SQLITE3_CALL(sqlite3_bind_int(stmt, 2, to));
    SQLITE3_CALL(sqlite3_bind_int(stmt, 3, static_cast<int>(rows)));
    BindBlob(stmt, 4, idx, rows * sizeof(uint32_t));
    BindBlob(stmt, 5, tgt_xy, rows * 2 * sizeof(float));
    BindBlob(stmt, 6, err, rows * sizeof(float));
    const int rc = sqlite3_step(stmt);
    sqlite3_reset(stmt);
    SQLITE3_CALL(rc);
}

void Database::ReadImagePairFlow(int32_t from, int32_t to, ImagePairFlow& flow) const {
    sqlite3_stmt* stmt = sql_stmt_read_image_pair_flows_;
    SQLITE3_CALL(sqlite3_bind_int(stmt, 1, from));
    SQLITE3_CALL(sqlite3_bind_int(stmt, 2, to));
    const int rc = SQLITE3_CALL(sqlite3_step(stmt));
    if (rc != SQLITE_ROW) {
        SQLITE3_CALL(sqlite3_reset(stmt));
        return;
    }
    const size_t rows = static_cast<size_t>(sqlite3_column_int(stmt, 0));
    ReadBlob(stmt, rows, 1, flow.src_kps_indices);
    ReadBlob(stmt, rows, 2, flow.tgt_kps);
    ReadBlob(stmt, rows, 3, flow.flow_errors);
    flow.image_id_from = from;
    flow.image_id_to = to;
    SQLITE3_CALL(sqlite3_reset(stmt));
}

ImagePairFlow Database::ReadImagePairFlow(int32_t from, int32_t to) const {
    ImagePairFlow f;
    ReadImagePairFlow(from, to, f);
    return f;
}

std::vector<int32_t> Database::FindOpticalFlowsFromImage(int32_t image_id_from) const {
    std::vector<int32_t> r;
    FindOpticalFlowsFromImage(image_id_from, r);
    return r;
}

void Database::FindOpticalFlowsFromImage(int32_t image_id_from, std::vector<int32_t>& result) const {
    sqlite3_stmt* stmt = sql_stmt_find_flows_from_image_;
    SQLITE3_CALL(sqlite3_bind_int(stmt, 1, image_id_from));
    while (SQLITE3_CALL(sqlite3_step(stmt)) == SQLITE_ROW) result.push_back(sqlite3_column_int(stmt, 0));
    SQLITE3_CALL(sqlite3_reset(stmt));
}

std::vector<int32_t> Database::FindOpticalFlowsToImage(int32_t image_id_to) const {
    std::vector<int32_t> r;
    FindOpticalFlowsToImage(image_id_to, r);
    return r;
}

void Database::FindOpticalFlowsToImage(int32_t image_id_to, std::vector<int32_t>& result) const {
    sqlite3_stmt* stmt = sql_stmt_find_flows_to_image_;
    SQLITE3_CALL(sqlite3_bind_int(stmt, 1, image_id_to));
    while (SQLITE3_CALL(sqlite3_step(stmt)) == SQLITE_ROW) result.push_back(sqlite3_column_int(stmt, 0));
    SQLITE3_CALL(sqlite3_reset(stmt));
}

bool Database::KeypointsExist(int32_t image_id) const {
    sqlite3_stmt* stmt = sql_stmt_keypoints_exist_;
    SQLITE3_CALL(sqlite3_bind_int(stmt, 1, image_id));
    const bool exists = SQLITE3_CALL(sqlite3_step(stmt)) == SQLITE_ROW;
    SQLITE3_CALL(sqlite3_reset(stmt));
    return exists;
}

bool Database::ImagePairFlowExists(int32_t from, int32_t to) const {
    sqlite3_stmt* stmt = sql_stmt_pair_flow_exist_;
    SQLITE3_CALL(sqlite3_bind_int(stmt, 1, from));
    SQLITE3_CALL(sqlite3_bind_int(stmt, 2, to));
    const bool exists = SQLITE3_CALL(sqlite3_step(stmt)) == SQLITE_ROW;
    SQLITE3_CALL(sqlite3_reset(stmt));
    return exists;
}

static int32_t SingleIntOrInvalid(sqlite3_stmt* stmt) {
    const int rc = SQLITE3_CALL(sqlite3_step(stmt));
    if (rc != SQLITE_ROW) {
        SQLITE3_CALL(sqlite3_reset(stmt));
        return kInvalidId;
    }
    const int32_t id = sqlite3_column_int(stmt, 0);
    SQLITE3_CALL(sqlite3_reset(stmt));
    return id;
}

int32_t Database::GetMinImageIdWithKeypoints() const { return SingleIntOrInvalid(sql_stmt_min_image_id_); }
int32_t Database::GetMaxImageIdWithKeypoints() const { return SingleIntOrInvalid(sql_stmt_max_image_id_); }

void Database::Begin() { SQLITE3_EXEC(database_, "BEGIN"); }
void Database::Commit() { SQLITE3_EXEC(database_, "COMMIT"); }

void Database::PrepareSQLStatements() {
    auto prep = [&](const char* sql, sqlite3_stmt** stmt) {
        SQLITE3_CALL(sqlite3_prepare_v2(database_, sql, -1, stmt, nullptr));
    };
    prep("SELECT rows, keypoints FROM keypoints WHERE image_id = ?;", &sql_stmt_read_keypoints_);
    prep("INSERT INTO keypoints(image_id, rows, keypoints) VALUES(?, ?, ?);", &sql_stmt_write_keypoints_);
    prep("SELECT rows, src_keypoints_indices, tgt_keypoints, flow_errors FROM optical_flow WHERE image_id_from = ? AND "
         "image_id_to = ?;",
         &sql_stmt_read_image_pair_flows_);
    prep("INSERT INTO optical_flow(image_id_from, image_id_to, rows, src_keypoints_indices, tgt_keypoints, "
         "flow_errors) VALUES(?, ?, ?, ?, ?, ?);",
         &sql_stmt_write_image_pair_flows_);
    prep("SELECT image_id_to FROM optical_flow WHERE image_id_from = ?", &sql_stmt_find_flows_from_image_);
    prep("SELECT image_id_from FROM optical_flow WHERE image_id_to = ?", &sql_stmt_find_flows_to_image_);
    prep("SELECT 1 FROM keypoints WHERE image_id = ?;", &sql_stmt_keypoints_exist_);
    prep("SELECT 1 FROM optical_flow WHERE image_id_from = ? AND image_id_to = ?;", &sql_stmt_pair_flow_exist_);
    prep("SELECT MIN(image_id) FROM keypoints;", &sql_stmt_min_image_id_);
    prep("SELECT MAX(image_id) FROM keypoints;", &sql_stmt_max_image_id_);
}

void Database::FinalizeSQLStatements() {
    sqlite3_stmt** all[] = {&sql_stmt_read_keypoints_,        &sql_stmt_write_keypoints_,
                            &sql_stmt_read_image_pair_flows_, &sql_stmt_write_image_pair_flows_,
                            &sql_stmt_find_flows_from_image_, &sql_stmt_find_flows_to_image_,
                            &sql_stmt_keypoints_exist_,       &sql_stmt_pair_flow_exist_,
                            &sql_stmt_min_image_id_,          &sql_stmt_max_image_id_};
    for (auto s : all) {
        if (*s) sqlite3_finalize(*s);
        *s = nullptr;
    }
}
