// numa_pin.h -- keep the host threads that feed the GPU on the GPU's own NUMA node.
//
// the same commit on two boxes gave 461 vs 366 frames/s for 4K analysis with the SQLite insert and 3480 vs 2858
// for the first tracking call; the host's share of a tracked frame was 56 vs 116 us.  On a two-socket host a thread that lands
// on the socket the GPU does NOT hang off reaches page-locked memory, the doorbells and the GPU's BAR across the socket link,
// and the page cache it writes into is remote for the one that reads it.  The threads this library owns -- the database
// writer and its page-write worker, the tracker's read-ahead thread, the worker threads of the *Thread classes -- ask for the
// CPUs of the GPU's node; the calling thread of a synchronous call (track_sequence, refine_trajectory,
// generate_optical_flow_database) does so for the duration of the call and gets its old mask back.  POLYCHASE_NUMA_PIN=0: no
// thread is touched.  A box with one node, a GPU without a node (numa_node = -1), a container whose cpuset excludes the node:
// nothing happens.
#pragma once

#include <string>

struct pc_context;

namespace numa {

// The NUMA node of the context's GPU (sysfs numa_node of its PCI function), or -1.
int GpuNode(pc_context* ctx);
// Restricts the CALLING thread to the CPUs of the GPU's node that its current mask allows.  `role` names the thread in
// Placement().  True if the mask was changed.
bool PinThisThreadNearGpu(pc_context* ctx, const char* role);
// The same for the duration of a scope.
class ScopedPin {
   public:
    ScopedPin(pc_context* ctx, const char* role);
    ~ScopedPin();
    ScopedPin(const ScopedPin&) = delete;
    ScopedPin& operator=(const ScopedPin&) = delete;

   private:
    bool changed_ = false;
    unsigned long saved_[16] = {0};   // cpu_set_t of up to 1024 CPUs
};
// What happened, as a JSON object (bench.py's `host` block): nodes, the GPU's node and its CPU list, and per role the CPUs
// the thread was allowed before / after.
std::string Placement();

}  // namespace numa
