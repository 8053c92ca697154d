// opticalflow_thread.h -- asynchronous wrapper around GenerateOpticalFlowDatabase with the message
// protocol of the reference (cpp/opticalflow_thread.h:21-32, :81-205): the worker pushes
// OpticalFlowRequest{frame_id} and blocks until ProvideFrame(frame_id, image) is called, pushes
// OpticalFlowProgress, and finishes with an optional error followed by `true`.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <optional>
#include <thread>
#include <variant>
#include <vector>

#include "opticalflow.h"

struct OpticalFlowProgress {
    float progress;
    std::string progress_message;
};

struct OpticalFlowRequest {
    int32_t frame_id;
};

// The reference stores std::unique_ptr<std::exception> (sliced: what() == "std::exception"); the
// message is kept here.
struct CppException {
    std::string message;
    const char* what() const { return message.c_str(); }
};

using OpticalFlowThreadMessage = std::variant<OpticalFlowProgress, OpticalFlowRequest, bool, CppException>;

template <typename Message>
class MessageQueue {
   public:
    void push(Message m) {
        std::lock_guard<std::mutex> lk(mtx_);
        q_.push_back(std::move(m));
    }
    std::optional<Message> try_pop() {
        std::lock_guard<std::mutex> lk(mtx_);
        if (q_.empty()) return std::nullopt;
        Message m = std::move(q_.front());
        q_.pop_front();
        return m;
    }
    bool empty() const {
        std::lock_guard<std::mutex> lk(mtx_);
        return q_.empty();
    }

   private:
    mutable std::mutex mtx_;
    std::deque<Message> q_;
};

class OpticalFlowThread {
   public:
    OpticalFlowThread(VideoInfo video_info, std::string database_path, GFTTOptions detector_options = {},
                      OpticalFlowOptions flow_options = {}, bool write_images = false)
        : video_info_(video_info),
          database_path_(std::move(database_path)),
          detector_options_(detector_options),
          flow_options_(flow_options),
          write_images_(write_images) {
        worker_ = std::thread([this] { Work(); });
    }

    ~OpticalFlowThread() { Join(); }

    void RequestStop() {
        {
            std::lock_guard<std::mutex> lk(mtx_);
            stop_ = true;
        }
        cv_.notify_all();
    }

    void Join() {
        if (worker_.joinable()) worker_.join();
    }

    std::optional<OpticalFlowThreadMessage> TryPop() { return queue_.try_pop(); }
    bool Empty() const { return queue_.empty(); }

    // Deep copy on entry (cpp/opticalflow_thread.h:120-132): the caller's buffer is free on return.
    void ProvideFrame(int32_t frame_id, const uint8_t* data, int rows, int cols, int channels, size_t row_pitch) {
        auto buf = std::make_shared<std::vector<uint8_t>>(static_cast<size_t>(rows) * cols * channels);
        const size_t row_bytes = static_cast<size_t>(cols) * channels;
        for (int y = 0; y < rows; y++) std::copy_n(data + y * row_pitch, row_bytes, buf->data() + y * row_bytes);
        FrameView v;
        v.data = buf->data();
        v.rows = rows;
        v.cols = cols;
        v.channels = channels;
        v.row_pitch = row_bytes;
        v.owner = buf;
        {
            std::lock_guard<std::mutex> lk(mtx_);
            provided_ = std::make_pair(frame_id, std::move(v));
        }
        cv_.notify_all();
    }

   private:
    void Work() {
        auto accessor = [this](int32_t frame_id) -> std::optional<FrameView> {
            queue_.push(OpticalFlowRequest{frame_id});
            std::unique_lock<std::mutex> lk(mtx_);
            const bool got = cv_.wait_for(lk, std::chrono::seconds(10), [&] { return provided_.has_value() || stop_; });
            if (stop_) return std::nullopt;
            // the reference dereferences an empty optional here on timeout (:145-158)
            if (!got) throw std::runtime_error("Timed out waiting for frame " + std::to_string(frame_id));
            if (provided_->first != frame_id)
                throw std::runtime_error("Requested frame " + std::to_string(frame_id) + " but got " +
                                         std::to_string(provided_->first));
            FrameView frame = std::move(provided_->second);
            provided_.reset();
            return frame;
        };
        auto progress = [this](float p, const std::string& msg) {
            queue_.push(OpticalFlowProgress{p, msg});
            std::lock_guard<std::mutex> lk(mtx_);
            return !stop_;
        };
        try {
            GenerateOpticalFlowDatabase(video_info_, accessor, progress, database_path_, detector_options_,
                                        flow_options_, write_images_);
        } catch (const std::exception& e) {
            bool stopped;
            {
                std::lock_guard<std::mutex> lk(mtx_);
                stopped = stop_;
            }
            // request_stop() while the worker waits for a frame makes the accessor return nothing;
            // the reference reports that as an error (":251-254"), here it is a clean cancellation
            if (stopped) queue_.push(OpticalFlowProgress{1.0f, "Cancelled"});
            else queue_.push(CppException{e.what()});
        } catch (...) {
            queue_.push(CppException{"Unknown exception type. This should never happen!"});
        }
        queue_.push(true);
    }

    const VideoInfo video_info_;
    const std::string database_path_;
    const GFTTOptions detector_options_;
    const OpticalFlowOptions flow_options_;
    const bool write_images_;

    MessageQueue<OpticalFlowThreadMessage> queue_;
    std::optional<std::pair<int32_t, FrameView>> provided_;
    std::mutex mtx_;
    std::condition_variable cv_;
    bool stop_ = false;
    std::thread worker_;
};
