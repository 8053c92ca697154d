// gpu_context.h -- process-wide pc_context for objects that Python creates without a context
// (AcceleratedMesh, track_sequence).  Device = $POLYCHASE_DEVICE (default 0).
#pragma once

#include "../../../include/polychase_hip.h"

#include <mutex>

pc_context* SharedGpuContext();  // throws std::runtime_error when no HIP device is usable

// The shared context is ONE HIP stream plus scratch buffers (its own, and the per-mesh ones of pc_mesh), and the C ABI
// is not thread safe per context.  Python may call ray_cast / AcceleratedMesh / SolvePnP from its thread while a
// TrackerThread or RefinerThread runs (the reference's Embree scene allows that), so every section that enqueues on
// the shared context and reads the results back holds this lock from the first enqueue to the synchronisation that
// ends it: one ray-cast call, one mask upload, the GPU part of one tracked frame, one refinement evaluation.
// Recursive: SolveFrame holds it around calls that take it themselves (SyncMask, SolvePnPIterativeOnGpu).
std::recursive_mutex& SharedGpuMutex();
struct GpuSection {
    std::lock_guard<std::recursive_mutex> lock{SharedGpuMutex()};
};
