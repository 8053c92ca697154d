// gpu_context.h -- process-wide pc_context for objects that Python creates without a context
// (AcceleratedMesh, track_sequence).  Device = $POLYCHASE_DEVICE (default 0).
#pragma once

#include "../../../include/polychase_hip.h"

pc_context* SharedGpuContext();  // throws std::runtime_error when no HIP device is usable
