// database.h -- the on-disk optical-flow database (SQLite), byte-compatible with the reference
// (cpp/database.h:36-100, cpp/database.cc): same schema, pragmas, statements and raw
// little-endian blob layout, so databases written by either implementation are interchangeable
// and a cancelled analysis can be resumed by the other.
#pragma once

#include <sqlite3.h>

#include <array>
#include <cstdint>
#include <limits>
#include <string>
#include <vector>

static constexpr int32_t kInvalidId = std::numeric_limits<int32_t>::max();

using Keypoint = std::array<float, 2>;            // Eigen::Vector2f in the reference: 2 x f32
using Keypoints = std::vector<Keypoint>;
using KeypointsIndices = std::vector<uint32_t>;
using FlowErrors = std::vector<float>;

struct ImagePairFlow {
    int32_t image_id_from = 0;
    int32_t image_id_to = 0;
    KeypointsIndices src_kps_indices;
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
    explicit Database(const std::string& path);
    Database(Database&& other) noexcept;
    Database(const Database&) = delete;
    ~Database();

    void Open(const std::string& path);
    void Close();

    Keypoints ReadKeypoints(int32_t image_id) const;
    void ReadKeypoints(int32_t image_id, Keypoints& keypoints) const;
    void WriteKeypoints(int32_t image_id, const Keypoints& keypoints);
    void WriteKeypoints(int32_t image_id, const float* xy, size_t rows);

    ImagePairFlow ReadImagePairFlow(int32_t image_id_from, int32_t image_id_to) const;
    void ReadImagePairFlow(int32_t image_id_from, int32_t image_id_to, ImagePairFlow& flow) const;
    void WriteImagePairFlow(int32_t image_id_from, int32_t image_id_to, const KeypointsIndices& src_kps_indices,
                            const Keypoints& tgt_kps, const FlowErrors& flow_errors);
    void WriteImagePairFlow(const ImagePairFlow& flow);
    void WriteImagePairFlow(int32_t image_id_from, int32_t image_id_to, const uint32_t* idx, const float* tgt_xy,
                            const float* err, size_t rows);

    std::vector<int32_t> FindOpticalFlowsFromImage(int32_t image_id_from) const;
    void FindOpticalFlowsFromImage(int32_t image_id_from, std::vector<int32_t>& result) const;
    std::vector<int32_t> FindOpticalFlowsToImage(int32_t image_id_to) const;
    void FindOpticalFlowsToImage(int32_t image_id_to, std::vector<int32_t>& result) const;

    bool KeypointsExist(int32_t image_id) const;
    bool ImagePairFlowExists(int32_t image_id_from, int32_t image_id_to) const;
    int32_t GetMinImageIdWithKeypoints() const;
    int32_t GetMaxImageIdWithKeypoints() const;

    // Not in the reference: explicit transactions so that the writer can batch one frame's rows.
    void Begin();
    void Commit();

   private:
    void CreateTables() const;
    void PrepareSQLStatements();
    void FinalizeSQLStatements();

    sqlite3* database_ = nullptr;
    sqlite3_stmt* sql_stmt_read_keypoints_ = nullptr;
    sqlite3_stmt* sql_stmt_write_keypoints_ = nullptr;
    sqlite3_stmt* sql_stmt_read_image_pair_flows_ = nullptr;
    sqlite3_stmt* sql_stmt_write_image_pair_flows_ = nullptr;
    sqlite3_stmt* sql_stmt_find_flows_from_image_ = nullptr;
    sqlite3_stmt* sql_stmt_find_flows_to_image_ = nullptr;
    sqlite3_stmt* sql_stmt_keypoints_exist_ = nullptr;
    sqlite3_stmt* sql_stmt_pair_flow_exist_ = nullptr;
    sqlite3_stmt* sql_stmt_min_image_id_ = nullptr;
    sqlite3_stmt* sql_stmt_max_image_id_ = nullptr;
};
