// multi_gpu.h -- the multi-GPU analysis as ONE C++ call per rank, no Python and no torch underneath.
//
// One process per GPU (any launcher: a shell loop, mpirun, tools/multi_gpu_analyze.cc, torchrun).  Every rank calls
// GenerateOpticalFlowDatabaseMultiGpu with the same clip and its own rank; frame1 ranges are sharded contiguously
// (shard r = [first + r n / R, first + (r + 1) n / R)), each rank analyses its shard with the single-GPU
// driver (GenerateOpticalFlowShard = the loop of cpp/opticalflow.cc:237-316 on a sub-range), and the records travel to rank 0
// -- the owner of the SQLite file -- IN FRAME ORDER, piece by piece, while the analysis runs:
//   * rank 0 stores its own shard directly (the single-GPU path) and meanwhile receives the first pieces of rank 1;
//   * every other rank appends its records to a two-part device log; a finished part is staged and sent when rank 0 asks
//     for it (credit flow control: memory is bounded by two parts + two staged pieces per rank, whatever the clip length);
//   * the payload moves with RCCL (ncclSend / ncclRecv over xGMI; include/polychase_hip.h: pc_comm_send / pc_comm_recv);
//     credits, piece headers and RCCL's unique id travel over a TCP control connection to rank 0 (blocking host
//     sockets: no collective kernel sits on a GPU spinning for a peer that is seconds away).
// The database does not depend on the number of ranks (tests/test_multi_gpu_cpp_gpu.py).  THIS is the one implementation of the
// protocol (round 6; rounds 3-5 carried a Python twin, distributed.OrderedPieceGather): polychase_amd/analyze.py calls it through
// polychase_core.generate_optical_flow_database_multi_gpu, a C++ host calls it directly (tools/multi_gpu/multi_gpu_analyze.cc).
#pragma once

#include <cstdint>
#include <functional>
#include <string>

#include "analysis.h"

struct MultiGpuConfig {
    int world_size = 1;
    int rank = 0;
    std::string master_addr = "127.0.0.1";   // rank 0 listens here
    int master_port = 29611;
    int device = -1;                         // HIP device of this rank; -1: the rank (one process per GPU of one node)
    int piece_frames = 16;                   // frame1s per piece of the record log
    size_t keypoints_per_frame = 0;          // sizing of a log part; 0: one keypoint per 40 pixels + slack
    // "rccl": ncclSend / ncclRecv, device to device (the product).  "tcp": the payload over the control connection through
    // host memory -- a testing aid for boxes where the ranks share ONE GPU (RCCL refuses two ranks on one device).
    std::string transport = "rccl";
    double connect_timeout_s = 120.0;
    // Testing aid (tests/test_distributed_cpu.py: the protocol on a box without a GPU, transport "tcp"): when set, NO analysis
    // runs -- the rank's shard is whatever this function hands to `emit`, piece by piece, in frame order: a piece = a packed record
    // log as the analyzer writes it (bytes, first frame1, number of frame1s).  Rank 0 stores its pieces directly, the other ranks
    // send theirs through the same queue, credits and headers as the analyzer's pieces.  May throw (a failing rank).
    std::function<void(int rank, const std::function<void(const void* bytes, size_t n, int32_t first_frame1, int n_frames)>& emit)> synthetic_shard;
};

struct MultiGpuResult {
    int32_t shard_begin = 0, shard_end = 0;
    int pieces = 0;                 // ranks > 0: pieces sent; rank 0: pieces received
    size_t bytes_moved = 0;
    bool cancelled = false;         // some rank's progress callback returned false: the clip has a hole
    double seconds_analysis = 0, seconds_total = 0, seconds_blocked = 0;
    OpticalFlowRunStats stats;      // of this rank's shard (rank 0: plus the rows of the received pieces)
};

// Collective over the ranks.  frame_accessor / callback as in GenerateOpticalFlowDatabase; database_path is used by rank 0.
// Throws std::runtime_error on every rank when any rank fails (a failing rank tells rank 0, rank 0 tells the rest).
MultiGpuResult GenerateOpticalFlowDatabaseMultiGpu(const VideoInfo& video_info, FrameAccessorFunction frame_accessor,
                                                   OpticalFlowProgressCallback callback, const std::string& database_path,
                                                   const MultiGpuConfig& config, const GFTTOptions& detector_options = {},
                                                   const OpticalFlowOptions& flow_options = {});
