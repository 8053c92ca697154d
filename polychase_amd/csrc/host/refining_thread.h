// refining_thread.h -- RefinerThread: RefineTrajectory on a worker thread (reference
// cpp/refiner_thread.h:16-96).  Messages: RefineTrajectoryUpdate per LM iteration, CppException on
// failure, and a final `true`.  The trajectory is shared with Python and updated in place.
#pragma once

#include <atomic>
#include <memory>
#include <variant>

#include "trajectory_refiner.h"
#include "worker.h"

using RefinerThreadMessage = std::variant<RefineTrajectoryUpdate, bool, CppException>;

class RefinerThread : public Worker<RefinerThreadMessage> {
   public:
    RefinerThread(std::string database_path, std::shared_ptr<CameraTrajectory> traj, Mat4f model_matrix,
                  std::shared_ptr<const AcceleratedMesh> mesh, bool optimize_focal_length, bool optimize_principal_point,
                  BundleOptions bundle_opts) {
        Start([=, this] {
            RefineTrajectory(
                database_path, *traj, model_matrix, *mesh, optimize_focal_length, optimize_principal_point,
                [this](RefineTrajectoryUpdate update) {
                    Push(std::move(update));
                    return !stop_.load();
                },
                bundle_opts);
        });
    }
    ~RefinerThread() override { Join(); }  // like the reference: waits, does not cancel

    void RequestStop() { stop_.store(true); }

   private:
    std::atomic<bool> stop_{false};
};
