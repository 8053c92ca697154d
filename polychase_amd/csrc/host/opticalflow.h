// opticalflow.h -- "Analyze Video": GenerateOpticalFlowDatabase on the MI355X path.
// Same entry point, option structs, callbacks, progress messages and database output as the
// reference (cpp/opticalflow.h:14-41, cpp/opticalflow.cc:209-321); cv::Mat is replaced by FrameView.
#pragma once

#include <cstdint>
#include <functional>
#include <memory>
#include <optional>
#include <string>

// H x W x 3 RGB u8 image (the reference's accessor returns cv::Mat of that shape).
struct FrameView {
    const uint8_t* data = nullptr;
    int rows = 0, cols = 0, channels = 0;
    size_t row_pitch = 0;          // bytes
    bool on_device = false;        // data is HIP device memory of the GPU in use
    std::shared_ptr<void> owner;   // keeps `data` alive
};

using FrameAccessorFunction = std::function<std::optional<FrameView>(int32_t frame_id)>;
using OpticalFlowProgressCallback = std::function<bool(float progress, const std::string& progress_message)>;

struct VideoInfo {  // cpp/opticalflow.h:20-25
    uint32_t width;
    uint32_t height;
    int32_t first_frame;
    uint32_t num_frames;
};

struct GFTTOptions {  // cpp/feature_detection/gftt.h:5-21
    double quality_level = 0.01;
    double min_distance = 5.0;
    int block_size = 3;
    int gradient_size = 3;
    int max_corners = 0;
    bool use_harris = false;
    double harris_k = 0.04;
    int grid_rows = 4;
    int grid_cols = 4;
};

struct OpticalFlowOptions {  // cpp/opticalflow.h:27-33
    int window_size = 10;
    int max_level = 3;
    int term_max_iters = 30;
    double term_epsilon = 0.01;
    double min_eigen_threshold = 1e-4;
};

// Statistics of one run (not in the reference; used by bench/tests).
struct OpticalFlowRunStats {
    int frames_processed = 0;
    int keypoint_rows_written = 0;
    int flow_rows_written = 0;
    double seconds_total = 0, seconds_db = 0;
};

// database_path empty => records are produced but not stored (bench mode).
void GenerateOpticalFlowDatabase(const VideoInfo& video_info, FrameAccessorFunction frame_accessor,
                                 OpticalFlowProgressCallback callback, const std::string& database_path,
                                 const GFTTOptions& detector_options = {}, const OpticalFlowOptions& flow_options = {},
                                 bool write_images = false, OpticalFlowRunStats* stats = nullptr);
