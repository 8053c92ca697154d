// analysis_thread.h -- OpticalFlowThread: GenerateOpticalFlowDatabase on a worker thread that asks
// Python for frames (reference cpp/opticalflow_thread.h:21-32, :81-205; driven by
// blender_addon/operators/analysis.py:242-285).  Messages: OpticalFlowRequest{frame_id} (answer
// with ProvideFrame), OpticalFlowProgress, CppException, and a final `true`.
#pragma once

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <variant>
#include <vector>

#include "analysis.h"
#include "frame_pool.h"
#include "worker.h"

struct OpticalFlowProgress {
    float progress;
    std::string progress_message;
};

struct OpticalFlowRequest {
    int32_t frame_id;
};

using OpticalFlowThreadMessage = std::variant<OpticalFlowProgress, OpticalFlowRequest, bool, CppException>;

class OpticalFlowThread : public Worker<OpticalFlowThreadMessage> {
   public:
    OpticalFlowThread(VideoInfo video_info, std::string database_path, GFTTOptions detector_options = {},
                      OpticalFlowOptions flow_options = {}, bool write_images = false) {
        Start(
            [=] {
                GenerateOpticalFlowDatabase(
                    video_info, [this](int32_t id) { return WaitForFrame(id); },
                    [this](float p, const std::string& msg) { return ReportProgress(p, msg); }, database_path,
                    detector_options, flow_options, write_images);
            },
            [this](const std::string& what) {
                // request_stop() while the worker waits for a frame makes the accessor come back empty, which
                // the driver reports as a missing frame (cpp/opticalflow.cc:251-254): here a clean cancellation
                if (StopRequested()) Push(OpticalFlowProgress{1.0f, "Cancelled"});
                else Push(CppException{what});
            });
    }
    ~OpticalFlowThread() override {
        RequestStop();
        Join();
    }

    void RequestStop() {
        {
            std::lock_guard<std::mutex> lk(frame_mtx_);
            stop_ = true;
        }
        frame_cv_.notify_all();
    }

    // Deep copy on entry (cpp/opticalflow_thread.h:120-132): the caller's buffer is free on return.  The copy
    // lands in pinned memory, which the GPU reads directly (frame_pool.h).
    void ProvideFrame(int32_t frame_id, const uint8_t* data, int rows, int cols, int channels, size_t row_pitch,
                      int elem_size = 1) {
        const size_t row_bytes = static_cast<size_t>(cols) * channels * elem_size;
        std::shared_ptr<void> pixels = AcquirePinnedFrameBuffer(static_cast<size_t>(rows) * row_bytes);
        uint8_t* dst = static_cast<uint8_t*>(pixels.get());
        for (int y = 0; y < rows; y++) std::copy_n(data + y * row_pitch, row_bytes, dst + y * row_bytes);
        FrameView view;
        view.data = dst;
        view.rows = rows;
        view.cols = cols;
        view.channels = channels;
        view.elem_size = elem_size;
        view.row_pitch = row_bytes;
        view.on_device = true;   // device-accessible (pinned) host memory
        view.pinned_host = true;
        view.owner = std::move(pixels);
        {
            std::lock_guard<std::mutex> lk(frame_mtx_);
            provided_id_ = frame_id;
            provided_ = std::move(view);
        }
        frame_cv_.notify_all();
    }

   private:
    bool StopRequested() {
        std::lock_guard<std::mutex> lk(frame_mtx_);
        return stop_;
    }
    bool ReportProgress(float progress, const std::string& msg) {
        Push(OpticalFlowProgress{progress, msg});
        return !StopRequested();
    }
    std::optional<FrameView> WaitForFrame(int32_t frame_id) {
        Push(OpticalFlowRequest{frame_id});
        std::unique_lock<std::mutex> lk(frame_mtx_);
        const bool signalled =
            frame_cv_.wait_for(lk, std::chrono::seconds(10), [&] { return provided_.has_value() || stop_; });
        if (stop_) return std::nullopt;
        // on a timeout the reference dereferences an empty optional (cpp/opticalflow_thread.h:145-158)
        if (!signalled) throw std::runtime_error("Timed out waiting for frame " + std::to_string(frame_id));
        if (provided_id_ != frame_id)
            throw std::runtime_error("Requested frame " + std::to_string(frame_id) + " but got " +
                                     std::to_string(provided_id_));
        std::optional<FrameView> frame = std::move(provided_);
        provided_.reset();
        return frame;
    }

    std::mutex frame_mtx_;
    std::condition_variable frame_cv_;
    std::optional<FrameView> provided_;
    int32_t provided_id_ = 0;
    bool stop_ = false;
};
