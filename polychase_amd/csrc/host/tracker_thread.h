// tracker_thread.h -- asynchronous TrackSequence with the reference's message protocol
// (cpp/tracker_thread.h:16-17, :62-84): FrameTrackingResult per frame, optional error, final `true`.
#pragma once

#include <atomic>
#include <memory>
#include <thread>
#include <variant>

#include "opticalflow_thread.h"  // MessageQueue, CppException
#include "tracker.h"

using TrackerThreadMessage = std::variant<FrameTrackingResult, bool, CppException>;

class TrackerThread {
   public:
    TrackerThread(std::string database_path, int32_t frame_from, int32_t frame_to_inclusive,
                  SceneTransformations scene_transform, std::shared_ptr<const AcceleratedMesh> accel_mesh,
                  bool optimize_focal_length, bool optimize_principal_point, BundleOptions bundle_opts)
        : database_path_(std::move(database_path)),
          frame_from_(frame_from),
          frame_to_inclusive_(frame_to_inclusive),
          scene_transform_(scene_transform),
          accel_mesh_(std::move(accel_mesh)),
          optimize_focal_length_(optimize_focal_length),
          optimize_principal_point_(optimize_principal_point),
          bundle_opts_(bundle_opts) {
        worker_ = std::thread([this] { Work(); });
    }
    ~TrackerThread() { Join(); }
    void RequestStop() { stop_.store(true); }
    void Join() {
        if (worker_.joinable()) worker_.join();
    }
    std::optional<TrackerThreadMessage> TryPop() { return queue_.try_pop(); }
    bool Empty() const { return queue_.empty(); }

   private:
    void Work() {
        auto callback = [this](const FrameTrackingResult& r) {
            queue_.push(r);
            return !stop_.load();
        };
        try {
            TrackSequence(database_path_, frame_from_, frame_to_inclusive_, scene_transform_, *accel_mesh_, callback,
                          optimize_focal_length_, optimize_principal_point_, bundle_opts_);
        } catch (const std::exception& e) {
            queue_.push(CppException{e.what()});
        } catch (...) {
            queue_.push(CppException{"Unknown exception type. This should never happen!"});
        }
        queue_.push(true);
    }

    const std::string database_path_;
    const int32_t frame_from_, frame_to_inclusive_;
    const SceneTransformations scene_transform_;
    const std::shared_ptr<const AcceleratedMesh> accel_mesh_;
    const bool optimize_focal_length_, optimize_principal_point_;
    const BundleOptions bundle_opts_;
    MessageQueue<TrackerThreadMessage> queue_;
    std::atomic<bool> stop_{false};
    std::thread worker_;
};
