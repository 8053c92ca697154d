// tracking_thread.h -- TrackerThread: TrackSequence on a worker thread (reference
// cpp/tracker_thread.h:16-17, :62-84).  Messages: one FrameTrackingResult per solved frame,
// CppException on failure, and a final `true`.
#pragma once

#include <atomic>
#include <memory>
#include <variant>

#include "track_sequence.h"
#include "worker.h"

using TrackerThreadMessage = std::variant<FrameTrackingResult, bool, CppException>;

class TrackerThread : public Worker<TrackerThreadMessage> {
   public:
    TrackerThread(std::string database_path, int32_t frame_from, int32_t frame_to_inclusive,
                  SceneTransformations scene_transform, std::shared_ptr<const AcceleratedMesh> accel_mesh,
                  bool optimize_focal_length, bool optimize_principal_point, BundleOptions bundle_opts) {
        Start([=] {
            TrackSequence(
                database_path, frame_from, frame_to_inclusive, scene_transform, *accel_mesh,
                [this](const FrameTrackingResult& r) {
                    Push(r);
                    return !stop_.load();
                },
                optimize_focal_length, optimize_principal_point, bundle_opts);
        });
    }
    ~TrackerThread() override {
        RequestStop();
        Join();
    }
    void RequestStop() { stop_.store(true); }

   private:
    std::atomic<bool> stop_{false};
};
