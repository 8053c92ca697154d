// frame_pool.h -- page-locked host buffers for frames on their way to the GPU.
// The reference clones every provided frame into a cv::Mat (cpp/opticalflow_thread.h:120-132); here the
// one copy a frame needs lands in pinned memory that the RGB->gray kernel reads over PCIe by itself, so
// the driver thread never blocks in a pageable host-to-device copy.  Buffers are recycled: Acquire()
// hands out the least recently released buffer of the right size, and the analysis driver never has
// more than 4 frames between a put and the kernel that consumed it, so a pool that keeps kMinIdle
// released buffers before reusing one cannot hand out memory the GPU still reads.
#pragma once

#include <cstddef>
#include <memory>

// Pinned buffer of at least `bytes`; returned to the pool when the last reference goes away.
std::shared_ptr<void> AcquirePinnedFrameBuffer(size_t bytes);

// memcpy for frames: large copies are cut into pieces for a few persistent worker threads.  One core copies ~8 GB/s into
// pinned memory -- 0.8 ms for a 1080p uint8 frame, 4 ms for Blender's float32 RGBA pixels (33 MB), i.e. 250 frames/s
// before the GPU has seen anything; the PCIe link behind it takes 50 GB/s.
void CopyFrameBytes(void* dst, const void* src, size_t bytes);
