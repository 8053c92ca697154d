// analysis.h -- "Analyze Video" on the MI355X path: option structs, frame hand-off type and the
// entry point GenerateOpticalFlowDatabase.
//
// The structs carry the reference's field names and defaults because polychase_core exposes them
// one to one (cpp/polychase_pybind.cc:119-145): GFTTOptions <- cpp/feature_detection/gftt.h:5-21,
// OpticalFlowOptions and VideoInfo <- cpp/opticalflow.h:20-33.  cv::Mat is replaced by FrameView.
#pragma once

#include <cstdint>
#include <functional>
#include <memory>
#include <optional>
#include <string>

// ---- options -----------------------------------------------------------------------------------

// Shi-Tomasi detector with per-grid-cell quality threshold.  On the HIP path block_size and
// gradient_size must be 3 and use_harris false (what the addon always uses).
struct GFTTOptions {
    double quality_level = 0.01;  // keep responses above quality_level * (max response of the grid cell)
    double min_distance = 5.0;    // greedy suppression radius, pixels
    int block_size = 3;
    int gradient_size = 3;
    int max_corners = 0;          // 0: unlimited
    bool use_harris = false;
    double harris_k = 0.04;
    int grid_rows = 4;            // not exposed to Python by the reference either
    int grid_cols = 4;
};

// Pyramidal Lucas-Kanade.  max_level counts pyramid levels ABOVE level 0 (OpenCV semantics).
struct OpticalFlowOptions {
    int window_size = 10;
    int max_level = 3;
    int term_max_iters = 30;
    double term_epsilon = 0.01;
    double min_eigen_threshold = 1e-4;
};

// The clip: frames first_frame .. first_frame + num_frames - 1, all width x height.
struct VideoInfo {
    uint32_t width;
    uint32_t height;
    int32_t first_frame;
    uint32_t num_frames;
};

// ---- frame hand-off ----------------------------------------------------------------------------

// One RGB frame, rows x cols x 3 uint8.  `data` is host memory, or HIP device memory of the GPU in
// use when on_device is set (zero-copy ingestion); `owner` keeps it alive while the frame is read.
struct FrameView {
    const uint8_t* data = nullptr;
    int rows = 0;
    int cols = 0;
    int channels = 0;
    int elem_size = 1;     // 1: uint8 (channels == 3); 4: float32 as Blender hands frames out (channels 3 or 4),
                           // converted on the GPU like the addon's `(image * 255).astype(np.uint8)`
    size_t row_pitch = 0;  // bytes between rows
    bool on_device = false;
    bool pinned_host = false;   // with on_device: `data` is page-locked host memory (frame_pool.h) -> PC_FRAME_PINNED_HOST
    std::shared_ptr<void> owner;
};

// Returns the frame with the given id, or nothing if it cannot be supplied.
using FrameAccessorFunction = std::function<std::optional<FrameView>(int32_t frame_id)>;
// (progress in [0,1], message); returning false cancels the run.
using OpticalFlowProgressCallback = std::function<bool(float progress, const std::string& progress_message)>;

// What one run did (not in the reference; used by bench / tests).
struct OpticalFlowRunStats {
    int frames_processed = 0;
    int keypoint_rows_written = 0;
    int flow_rows_written = 0;
    double seconds_total = 0;
    double seconds_db = 0;
    // where the driver's thread spent the call (the stage clock of tools/e2e_bench.py)
    double seconds_setup = 0;        // database open, context + analyzer creation (before the first frame is asked for)
    double seconds_accessor = 0;     // inside frame_accessor (Python: the GIL, the copy into a pinned buffer)
    double seconds_put = 0;          // pc_analyzer_put_frame* (enqueue of the frame's preparation)
    double seconds_submit = 0;       // pc_analyzer_submit (keypoint count of frame1 + enqueue of its job)
    double seconds_collect = 0;      // pc_analyzer_collect (waiting for the GPU)
    double seconds_writer_wait = 0;  // RecordWriter::Enqueue / Flush (waiting for SQLite)
    double seconds_callback = 0;     // progress callback + database look-ups of the loop
    bool engine_reused = false;      // the call took the process's parked engine (context + analyzer) instead of creating one
};

// ---- entry point -------------------------------------------------------------------------------

// Detects keypoints in every frame and tracks them into the frames at distance 1, 2, 4 and 8 on
// both sides, storing `keypoints` and `optical_flow` rows in the SQLite file at database_path
// (created if missing; existing rows are kept and not recomputed, so an interrupted run resumes).
// An empty database_path produces the records without storing them.  write_images: the reference's debug dump
// (opticalflow.cc:80-96): <directory of the database>/frames/%06d.png and keypoints_%06d.png of every frame1.
void GenerateOpticalFlowDatabase(const VideoInfo& video_info, FrameAccessorFunction frame_accessor,
                                 OpticalFlowProgressCallback callback, const std::string& database_path,
                                 const GFTTOptions& detector_options = {}, const OpticalFlowOptions& flow_options = {},
                                 bool write_images = false, OpticalFlowRunStats* stats = nullptr);

// Frees the idle engine a finished run has parked for the next one (GPU memory of ~20 resident frames; analysis_driver.cc:
// EngineCache).  POLYCHASE_ENGINE_CACHE=0 disables the parking altogether.
void ReleaseCachedEngine();
// testing aid: true while the idle timer of a parked engine exists (a thread that lives only as long as an engine is parked)
bool EngineCacheTimerRunning();

// ---- multi-GPU analysis (one process per GPU, frame1 ranges sharded, RCCL only for the stitch) ----------

// One shard of the clip: the frame1 loop of GenerateOpticalFlowDatabase (cpp/opticalflow.cc:237-316) for
// frame1 in [begin, end) only.  Frames up to 8 outside the shard are requested from the accessor as tracking targets
// (gray + pyramid, no detection); pairs are clipped to the clip of `video_info`, so the records do not depend on how
// the clip is cut.  Where the records go:
//   * database_path non-empty: into the SQLite file, exactly like GenerateOpticalFlowDatabase (the rank that owns the
//     file runs its own shard this way: the inserts overlap the analysis);
//   * device_log non-null: every frame1's record (keypoints + the status == 1 rows of its pairs) is appended on the
//     GPU to `device_log` (HIP device memory of the GPU in use, 16-byte aligned) in the analyzer's log format
//     (include/polychase_hip.h, pc_analyzer_set_device_log) -- the bytes the ranks exchange over RCCL.
// The log buffer is used as `log_buffers` equal parts that are filled in turn, one PIECE each: a piece ends after
// `piece_frames` frame1s (0: never) or when the next record does not fit the part.  on_piece is called on the
// driver's thread when all records of a piece are complete in device memory; the part may be overwritten as soon as
// the call returns (with two parts the analysis of the next piece runs meanwhile).  Memory is therefore bounded by the
// buffer, whatever the length of the clip.
struct OpticalFlowShard {
    int32_t begin = 0, end = 0;
    void* device_log = nullptr;
    size_t capacity_bytes = 0;
    int log_buffers = 1;
    int piece_frames = 0;
    std::function<void(int piece, size_t offset_bytes, size_t bytes, int32_t first_frame1, int n_frames)> on_piece;
    bool host_records = true;   // false (and no database): the records are not downloaded to host memory as well
    int device = -1;            // HIP device of the run; -1: $POLYCHASE_DEVICE, else 0
    // results
    size_t used_bytes = 0;      // bytes of the last piece (a run with one piece: of the run)
    int pieces = 0;
    bool cancelled = false;     // the progress callback returned false: the records end early
};
void GenerateOpticalFlowShard(const VideoInfo& video_info, FrameAccessorFunction frame_accessor, OpticalFlowProgressCallback callback,
                              const std::string& database_path, OpticalFlowShard& shard, const GFTTOptions& detector_options = {},
                              const OpticalFlowOptions& flow_options = {}, OpticalFlowRunStats* stats = nullptr);
// The same with ONE piece in a buffer of capacity_bytes and nothing stored.  Returns the bytes used; throws
// ("device log full") when the log is too small.
size_t GenerateOpticalFlowRecords(const VideoInfo& video_info, FrameAccessorFunction frame_accessor,
                                  OpticalFlowProgressCallback callback, int32_t shard_begin, int32_t shard_end,
                                  void* device_log, size_t capacity_bytes, const GFTTOptions& detector_options = {},
                                  const OpticalFlowOptions& flow_options = {}, OpticalFlowRunStats* stats = nullptr);

// Stores a record log (host copy, e.g. one rank's part of the all-gather) in the database: per record one transaction
// with the `keypoints` row and its `optical_flow` rows, rows that exist are kept (opticalflow.cc:168-178, :286) --
// the statements of the single-process run in the same order, so logs written in frame order give the same file.
// A `keypoints` row that exists must hold the record's keypoints (else: std::runtime_error, nothing of the record stored).
void WriteOpticalFlowRecords(const std::string& database_path, const uint8_t* log, size_t bytes,
                             OpticalFlowRunStats* stats = nullptr);
// The same for a sequence of logs (the pieces rank 0 receives): one connection, rows loaded under a rollback journal
// like GenerateOpticalFlowDatabase's own inserts, the file back in WAL mode when the writer is closed or destroyed.
class OpticalFlowRecordWriter {
   public:
    explicit OpticalFlowRecordWriter(const std::string& database_path);
    ~OpticalFlowRecordWriter();
    void Write(const uint8_t* log, size_t bytes, OpticalFlowRunStats* stats = nullptr);
    void Close();

   private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};
