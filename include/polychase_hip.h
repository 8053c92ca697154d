/*
 * polychase_hip.h -- C ABI of the MI355X (gfx950) video-analysis hot path.
 *
 * This is the drop-in boundary BELOW the reference's pybind11 module `polychase_core`
 * (/root/reference/cpp/polychase_pybind.cc:29): the reference has no FFI of its own, its host C++
 * calls OpenCV directly.  Each entry point below replaces one OpenCV / in-repo call on the path
 * GenerateOpticalFlowDatabase (cpp/opticalflow.cc:209-321) and cites it.  Plain pointers and sizes
 * only; no torch / Eigen / OpenCV types.  All functions return 0 on success, a negative PC_E_* code
 * on failure; pc_last_error() returns the message of the calling thread's last failure.
 *
 * Memory: "host" pointers are ordinary (or pinned) CPU memory; "device" pointers are HIP device
 * memory of the context's GPU (e.g. torch.Tensor.data_ptr()).  Work is enqueued on the context's HIP
 * stream; functions that return data to the host synchronise that stream themselves.
 *
 * There is NO CPU fallback: every function fails with PC_E_NO_DEVICE when no gfx950 device is
 * usable.
 */
#ifndef POLYCHASE_HIP_H_
#define POLYCHASE_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PC_OK 0
#define PC_E_INVALID (-1)     /* bad argument / unsupported option */
#define PC_E_NO_DEVICE (-2)   /* no usable HIP device */
#define PC_E_HIP (-3)         /* a HIP runtime call failed */
#define PC_E_CAPACITY (-4)    /* caller buffer too small */
#define PC_E_STATE (-5)       /* call sequence error (e.g. frame has no keypoints) */

#define PC_MAX_TARGETS 8      /* the reference tracks each frame into <= 8 neighbours (opticalflow.cc:76-77) */
#define PC_MAX_LEVELS 16      /* a pyramid ends where the next level would be <= the window (>= 3 px): 14 levels at most for the
                                 2^30 pixels a frame may have; any max_level >= 0 is accepted (OpenCV: maxLevel is free) */
#define PC_MAX_WINDOW 31      /* OpticalFlowOptions.window_size is free in the reference (opticalflow.h:27-33); OpenCV's own default is 21.
                                 3: lk4, 4..10: the two-keypoint kernel (lk3), 11..31: lk4 (one keypoint per wavefront, 8 lanes per target) */

typedef struct pc_context pc_context;
typedef struct pc_frame pc_frame;

/* GFTTOptions, cpp/feature_detection/gftt.h:5-21 (same fields, same defaults) */
typedef struct pc_gftt_options {
    double quality_level; /* 0.01 */
    double min_distance;  /* 5.0   (any value >= 0: up to 64 the table-driven parallel suppression, above it the reference's loop against a grid of accepted corners on one wavefront) */
    int block_size;       /* 3   (any size >= 1: 3 runs the tiled kernel, others the general pair of kernels) */
    int gradient_size;    /* 3   (aperture of cornerEigenValsVecs' derivative: Sobel 3 / 5 / 7, or -1 = Scharr; anything else: PC_E_INVALID) */
    int max_corners;      /* 0 = unlimited */
    int use_harris;       /* 0   (1: cornerHarris, gftt.cc:31-33 -- calcHarris' scalar expression (k in double) in the canonical
                                 mode; under PC_ARITH_SOBEL_FMA the first width / 4 * 4 columns of a row as the vector loop of an
                                 x86 build computes them, in float with (float)k, and the scalar expression for the rest;
                                 bit-exact against the oracle's restatement of either -- the reference's addon never takes
                                 this branch) */
    double harris_k;      /* 0.04 */
    int grid_rows;        /* 4 */
    int grid_cols;        /* 4 */
} pc_gftt_options;

/* OpticalFlowOptions, cpp/opticalflow.h:27-33 */
typedef struct pc_flow_options {
    int window_size;            /* 10  (3..PC_MAX_WINDOW) */
    int max_level;              /* 3   (any value >= 0: the pyramid ends where the next level would be <= the window, as in OpenCV) */
    int term_max_iters;         /* 30 */
    double term_epsilon;        /* 0.01 */
    double min_eigen_threshold; /* 1e-4 */
} pc_flow_options;

void pc_gftt_default_options(pc_gftt_options* o);
void pc_flow_default_options(pc_flow_options* o);

const char* pc_last_error(void);
/* Library / build identification ("polychase_hip gfx950 ..."). */
const char* pc_version(void);

/* Process-wide preparation of the HIP runtime for the engine's streams: raises GPU_MAX_HW_QUEUES to 16 unless the
 * environment already holds a value (the engine keeps five streams busy; the runtime maps all streams of a process onto
 * that many hardware queues, 4 by default, and two streams on one queue wait for each other's commands: DESIGN.md
 * section 3).  The runtime reads the variable when the process FIRST touches HIP, so this must run before that;
 * pc_context_create calls it, hosts that initialise HIP themselves (torch, Cycles) call it earlier -- the polychase_core
 * module does when it is imported.  Idempotent.  *runtime_was_up (may be null) = 1 when the ROCm runtime of this process
 * was already initialised at the first call (the setting then has no effect on this process); *hw_queues (may be null) =
 * the value of the variable after the call.  Nothing happens at dlopen any more (round 3 used a library constructor). */
int pc_runtime_init(int* runtime_was_up, int* hw_queues);

/* ---- context: one per GPU (one process per GPU in multi-GPU runs) ---- */
int pc_context_create(int device_index, pc_context** out);
void pc_context_destroy(pc_context* ctx);
int pc_context_synchronize(pc_context* ctx);
/* "0000:c1:00.0" of the context's GPU (hipDeviceGetPCIBusId): the host side looks up the GPU's NUMA node with it (csrc/host/numa_pin.h) */
int pc_context_pci_bus_id(pc_context* ctx, char* buf, int len);
/* Arithmetic mode: where OpenCV's result depends on how the host executes it, which execution the GPU reproduces.
 *   PC_ARITH_CANONICAL      no FMA anywhere; the LK sums (structure tensor, mismatch vector) exact in integers, rounded once
 *   PC_ARITH_LK_X86_ORDER   the LK sums in fp32 in the order of LKTrackerInvoker's CV_SIMD128 path on x86 (four vector lanes
 *                           over the first (win / 8) * 8 columns, a scalar accumulator over the rest; calcOpticalFlowPyrLK,
 *                           cpp/opticalflow.cc:119-125): differs from the canonical order only where a window's partial
 *                           sums exceed 2^24 (step edges), by <= ~2e-3 px.  Windows 4-11 run the canonical integer data
 *                           path plus a proof that the fp32 sums are exact, and the x86 order itself where the proof fails
 *                           (+10-18 % on the LK launch); other windows run the generic kernel
 *   PC_ARITH_SOBEL_FMA      the fused multiply-add of the AVX2-dispatched symmetric column filter of Sobel inside
 *                           cornerMinEigenVal (cpp/feature_detection/gftt.cc:35): same corners, the (value, address)
 *                           order of near-ties -- i.e. keypoint indices -- as a stock x86 build produces them; with
 *                           use_harris also calcHarris' vector loop (float, (float)k) over the first width / 4 * 4 columns
 *   PC_ARITH_OPENCV_X86     both
 *   PC_ARITH_SOBEL_ROW_FMA  (not part of the default) additionally the ROW pass of Dy -- the smoothing taps [1, 2, 1] * scale
 *                           over the 8-bit pixels -- as a fused chain t = k0 a; t = fma(k1, b, t); t = fma(k0, c, t): what the
 *                           8u -> 32f vector row filter computes IF the linked OpenCV has one and dispatches it to AVX2 (a
 *                           second hypothesis about the real build; `tests/opencv_crosscheck.py` on a machine with cv2 tells
 *                           which one holds and names the POLYCHASE_ARITH value to use)
 * Bit for bit what oracle/pc_oracle.c computes under pco_set_opencv_emulation(flags).  Default: PC_ARITH_OPENCV_X86 -- the
 * execution of the OpenCV build the reference links (vcpkg, x86-64: SSE baseline, AVX2 / FMA3 dispatched)
 * -- or the environment variable POLYCHASE_ARITH = canonical | opencv_x86 | lk_x86 | sobel_fma | sobel_fma_rows | opencv_x86_rows at
 * context creation. */
#define PC_ARITH_CANONICAL 0
#define PC_ARITH_LK_X86_ORDER 1
#define PC_ARITH_SOBEL_FMA 2
#define PC_ARITH_SOBEL_ROW_FMA 4
#define PC_ARITH_OPENCV_X86 3
/* Synchronous copy of `bytes` bytes of this context's device memory to host memory (debug dump of frames that were handed
 * over as device memory, cpp/opticalflow.cc:80-96). */
int pc_context_download(pc_context* ctx, void* dst_host, const void* src_device, size_t bytes);
int pc_context_set_arithmetic(pc_context* ctx, int flags);
int pc_context_get_arithmetic(const pc_context* ctx);
/* hipStream_t the context enqueues on (for callers that time with HIP events / torch streams). */
void* pc_context_stream(pc_context* ctx);
/* Timing of the context's kernels with HIP events on the stream each launch is enqueued on.  `class_mask` bit k
 * enables kernel class PC_K_k (0 = off, 0xff = all); enabled classes accumulate (launches, total ms).
 * Two event records per timed launch: keep the mask to the class of interest inside timed regions.
 * pc_analyzer overlaps the LK launches of consecutive frames (two job lanes), so `total_ms` -- the sum of the
 * launches' own start-to-end durations -- exceeds the time the class kept the GPU busy; pc_context_get_busy_time
 * returns that: the length of the union of the launches' intervals. */
int pc_context_enable_timing(pc_context* ctx, int class_mask);
int pc_context_get_timing(pc_context* ctx, int kernel_class, int* launches, double* total_ms);
int pc_context_get_busy_time(pc_context* ctx, int kernel_class, double* busy_ms);
int pc_context_reset_timing(pc_context* ctx);
/* Diagnostics of the LK kernel (only a library compiled with -DPC_LK_PROFILE counts anything; otherwise all zeros):
 * shader-clock cycles summed over the wavefronts of the latest launch, per phase --
 * [0] I-side staging, [1] I-side evaluation + 2x2 system, [2] patch pick-up, [3] J-region staging, [4] iterations,
 * [5] error pass, [6] wavefront life time, [7] wavefronts, [8] wavefront-iterations, [9] region stagings. */
#define PC_LK_PROFILE_SLOTS 16
int pc_debug_lk_profile(pc_context* ctx, unsigned long long* out /* [PC_LK_PROFILE_SLOTS] */);
/* Diagnostics of PC_ARITH_LK_X86_ORDER on the two-keypoint LK kernel (windows 4-11).  That kernel runs the canonical
 * integer data path and proves, per level and per iteration, that the fp32 sums of the x86 order would be exact (then
 * they equal the canonical values); only where the proof fails it evaluates the sums in the x86 order.  out (may be
 * null) receives the counters accumulated since counting was enabled: [0] (pair, iteration)s decided by the proof,
 * [1] evaluated in the x86 order, [2] (keypoint, level)s, [3] of those with the structure tensor in the x86 order.
 * enable != 0 (re)starts counting from zero, 0 stops it.  Counting costs a few atomics per wavefront. */
int pc_debug_lk_x86_stats(pc_context* ctx, int enable, unsigned long long* out /* [4] */);
/* Diagnostics of the device-resident PnP solver: ONE damped 9x9 system through the in-kernel float32 Cholesky
 * factorisation and solve that pc_pnp_solve uses (row-major a81, only the lower triangle is read; l81 receives the
 * factor, x9 the solution of L L^T x = b; *positive_definite = 0 and x9 = 0 when the factorisation fails).  Exists so
 * that the reference's known-answer test (cpp/examples/levmarq_ill_conditioned_float32_issue.cpp:16-63) runs on the
 * device copy of the solver, not only on the host one. */
int pc_debug_llt9(pc_context* ctx, const float* a81, const float* b9, float* l81, float* x9, int* positive_definite);
#define PC_K_GRAY 0
#define PC_K_PYRAMID 1
#define PC_K_MINEIG 2
#define PC_K_NMS 3
#define PC_K_SORT 4
#define PC_K_SUPPRESS 5
#define PC_K_LK 6
#define PC_K_COMPACT 7
#define PC_K_COUNT 8

/* ---- frame: gray image + LK pyramid (+ Scharr derivative planes) + keypoints, resident in HBM ----
 * Replaces the per-frame state of cpp/opticalflow.cc:223-226 (frame1_gray, features, frame1_pyramid)
 * and OpticalFlowCache (:18-37). */
int pc_frame_create(pc_context* ctx, int width, int height, int window_size, int max_level,
                    pc_frame** out);
/* Values of the `on_device` argument of the pc_frame_set_* / pc_analyzer_put_frame* calls:
 * 0 = pageable host memory (copied synchronously: the caller may reuse it when the call returns), 1 = memory the GPU can
 * read (device memory, or pc_host_buffer_alloc memory read over PCIe by the kernels), PC_FRAME_PINNED_HOST = page-locked
 * host memory (pc_host_buffer_alloc) that the caller leaves untouched until the frame has been consumed
 * (pc_analyzer_frame_ingested; for pc_frame_set_*: the next synchronising call): fetched by the copy engine, no wait. */
#define PC_FRAME_PINNED_HOST 2
void pc_frame_destroy(pc_frame* f);

/* cv::cvtColor(COLOR_RGB2GRAY) (opticalflow.cc:259,:298) + cv::buildOpticalFlowPyramid
 * (opticalflow.cc:180-187) in one call.  rgb: H rows of W*3 bytes, row_pitch bytes apart.
 * on_device != 0: rgb is a device pointer; otherwise host memory (copied to the GPU first).
 * Clears the frame's keypoints. */
int pc_frame_set_rgb(pc_context* ctx, pc_frame* f, const uint8_t* rgb, size_t row_pitch, int on_device);
/* Same, from the float32 image Blender hands out: H rows of W*channels floats (channels = 3 or 4,
 * alpha ignored), row_pitch BYTES apart.  Replaces the addon's numpy conversion
 * `(image_data * 255).astype(np.uint8)` (blender_addon/operators/analysis.py:221-233) followed by
 * cvtColor: the same fp32 multiply and truncating cast, on the GPU. */
int pc_frame_set_rgb_f32(pc_context* ctx, pc_frame* f, const float* rgb, size_t row_pitch, int channels, int on_device);
/* Page-locked host memory that the GPU reads directly (hipHostMalloc): a frame copied into such a buffer
 * can be handed to pc_frame_set_rgb* / pc_analyzer_put_frame* with on_device = 1 -- the gray kernel then
 * pulls the pixels over PCIe itself, asynchronously, instead of a blocking pageable copy into a staging
 * buffer.  The buffer must stay untouched until the frame has been converted (for the analyzer: until 4
 * more frames have been put).  Replaces the cv::Mat clone of opticalflow_thread.h:120-132. */
int pc_host_buffer_alloc(size_t bytes, void** out);
void pc_host_buffer_free(void* buffer);
/* Same, from an 8-bit gray image (tests / callers that already hold gray). */
int pc_frame_set_gray(pc_context* ctx, pc_frame* f, const uint8_t* gray, size_t row_pitch, int on_device);

int pc_frame_num_levels(const pc_frame* f); /* = maxLevel returned by buildOpticalFlowPyramid + 1 */
int pc_frame_level_size(const pc_frame* f, int level, int* width, int* height);
/* Read-back (tests, debugging).  Layouts are tightly packed, matching OpenCV's padded pyramid:
 * image  (h + 2*win) x (w + 2*win) u8, REFLECT_101 border;
 * deriv  (h + 2*win) x (w + 2*win) x 2 int16 (dx,dy interleaved), zero border. */
int pc_frame_download_gray(pc_context* ctx, const pc_frame* f, uint8_t* out_gray);
int pc_frame_download_level(pc_context* ctx, const pc_frame* f, int level, uint8_t* out_padded);
int pc_frame_download_deriv(pc_context* ctx, const pc_frame* f, int level, int16_t* out_padded);

/* GoodFeaturesToTrack (cpp/feature_detection/gftt.cc:14-192, called at opticalflow.cc:160) on the
 * frame's gray image; keypoints stay on the device, in acceptance order. */
int pc_frame_detect(pc_context* ctx, pc_frame* f, const pc_gftt_options* opt);
/* cv::cornerMinEigenVal map of the last pc_frame_detect (gftt.cc:35), w*h floats (tests). */
int pc_frame_download_min_eig(pc_context* ctx, const pc_frame* f, float* out_eig);
/* Number of local-maximum candidates of the last pc_frame_detect (gftt.cc:76-86) (tests). */
int pc_frame_num_candidates(pc_context* ctx, const pc_frame* f, int* out_n);
int pc_frame_num_keypoints(pc_context* ctx, const pc_frame* f, int* out_n);
int pc_frame_download_keypoints(pc_context* ctx, const pc_frame* f, float* out_xy, int capacity);
/* Keypoints read back from the database on resume (opticalflow.cc:168-178). host pointer, n x 2. */
int pc_frame_set_keypoints(pc_context* ctx, pc_frame* f, const float* xy, int n);

/* cv::calcOpticalFlowPyrLK(frame1 pyramid, target pyramid, frame1 keypoints, ...) (opticalflow.cc:119-125)
 * for n_targets targets in one launch.  Raw outputs (host pointers), target-major:
 *   next_xy [n_targets][N][2], status [n_targets][N], err [n_targets][N],  N = keypoints of frame1. */
int pc_lk_track(pc_context* ctx, const pc_frame* frame1, const pc_frame* const* targets, int n_targets,
                const pc_flow_options* opt, float* next_xy, uint8_t* status, float* err);

/* Same + the status==1 filter of opticalflow.cc:130-147, compacted on the device, ascending
 * keypoint index.  Host outputs, each with room for n_targets*N rows; rows of target t start at
 * row_offset[t] (row_offset has n_targets+1 entries; row_offset[n_targets] = total rows). */
int pc_lk_track_filtered(pc_context* ctx, const pc_frame* frame1, const pc_frame* const* targets,
                         int n_targets, const pc_flow_options* opt, uint32_t* src_indices,
                         float* tgt_xy, float* flow_err, int64_t* row_offset);

/* ---- analyzer: the pipelined per-clip engine behind GenerateOpticalFlowDatabase ----
 * (cpp/opticalflow.cc:209-321).  Holds a ring of resident frames (the reference's 17-frame
 * SequentialWrapper cache, cpp/opticalflow_thread.h:34-79, generalised), builds every frame's gray
 * image and pyramid ONCE (the reference rebuilds them per pair, opticalflow.cc:298-302), and runs
 * frame1 jobs asynchronously: submit() enqueues, collect() hands back the records of the oldest
 * job in pinned host memory.  Internally three HIP streams: frame preparation (gray, pyramid,
 * detection, keypoint ordering) and two job lanes that take the frame1 jobs alternately -- a job's LK
 * launch, its compaction, device-log append and record download stay on one lane, so the launches of
 * consecutive frames overlap; per-slot events carry the dependencies.
 * One analyzer per context; calls are not thread-safe. */
typedef struct pc_analyzer pc_analyzer;

typedef struct pc_frame_result {
    int32_t frame1;
    int32_t n_keypoints;
    int32_t keypoints_detected;            /* 1: detected by this run (a new `keypoints` row), 0: supplied */
    const float* keypoints_xy;             /* n_keypoints x 2 (gftt.cc acceptance order) */
    int32_t n_targets;
    int32_t targets[PC_MAX_TARGETS];       /* frame ids, in submit order */
    int64_t row_offset[PC_MAX_TARGETS + 1];/* rows of target t: [row_offset[t], row_offset[t+1]) */
    const uint32_t* src_indices;           /* src_keypoints_indices, ascending per target */
    const float* tgt_xy;                   /* tgt_keypoints, rows x 2 */
    const float* flow_err;                 /* flow_errors */
} pc_frame_result;

#define PC_ANALYZER_SPARE_SLOTS 3
#define PC_ANALYZER_LOOKAHEAD 1
int pc_analyzer_create(pc_context* ctx, int width, int height, const pc_gftt_options* gftt,
                       const pc_flow_options* flow, int ring_frames, int max_jobs, pc_analyzer** out);
void pc_analyzer_destroy(pc_analyzer* a);
/* Back to the state after pc_analyzer_create -- no resident frame, no job, no device log -- keeping every allocation
 * (frame slabs, detection scratch, pinned result buffers, streams and events): the next clip of the same geometry and
 * options starts without the tens of milliseconds of creation and destruction (analysis_driver.cc keeps one idle engine
 * per process).  Fails with PC_E_STATE while jobs are in flight.  Synchronises the analyzer's streams. */
int pc_analyzer_reset(pc_analyzer* a);
/* Make `frame_id` resident (evicting the frame ring_frames + PC_ANALYZER_SPARE_SLOTS ids back: the ring holds three
 * slots more than asked for, so that a put never waits for LK launches that still read older frames and the caller
 * can run PC_ANALYZER_LOOKAHEAD = 1 frame ahead of the window frame1 - 8 .. frame1 + 8 of the current submit):
 * RGB->gray + pyramid, and, when will_detect != 0, the dense part of GoodFeaturesToTrack.
 * Replaces RequestFrame + cvtColor + GeneratePyramid (opticalflow.cc:249-263, :287-302). */
int pc_analyzer_put_frame(pc_analyzer* a, int32_t frame_id, const uint8_t* rgb, size_t row_pitch,
                          int on_device, int will_detect);
/* pc_analyzer_put_frame for a float32 frame (see pc_frame_set_rgb_f32). */
int pc_analyzer_put_frame_f32(pc_analyzer* a, int32_t frame_id, const float* rgb, size_t row_pitch, int channels,
                              int on_device, int will_detect);
int pc_analyzer_has_frame(const pc_analyzer* a, int32_t frame_id);
/* 1 when the pixels handed to pc_analyzer_put_frame for `frame_id` are no longer read (its gray conversion has
 * finished, or the frame has left the ring), 0 while the GPU may still read them.  Device / pinned sources are read
 * asynchronously: the caller keeps such a buffer alive, and unmodified, until this returns 1. */
int pc_analyzer_frame_ingested(const pc_analyzer* a, int32_t frame_id);
/* Keypoints already stored in the database for this frame (resume, opticalflow.cc:168-178). */
int pc_analyzer_set_keypoints(pc_analyzer* a, int32_t frame_id, const float* xy, int n);
/* Enqueue one iteration of the outer loop (opticalflow.cc:259-309) for frame1 and the given
 * resident target frames (0..PC_MAX_TARGETS of them; the caller has already dropped pairs whose
 * flow exists, opticalflow.cc:286).  Fails with PC_E_STATE when max_jobs jobs are in flight. */
int pc_analyzer_submit(pc_analyzer* a, int32_t frame1, const int32_t* targets, int n_targets);
int pc_analyzer_pending(const pc_analyzer* a);
/* Device-resident record log (multi-GPU stitch): when set, every submitted job also appends its
 * records to `d_log` (device memory owned by the caller, e.g. a torch tensor that is later handed
 * to an RCCL all-gather) without a host round trip.  Layout per job, all little-endian, 16-B aligned:
 *   int64 hdr[16] = {PC_LOG_MAGIC, frame1, n_keypoints, n_targets, targets[8], rows_capacity, 0, 0, 0}
 *   int64 row_offset[16]      (first n_targets+1 valid; written by the GPU)
 *   float  keypoints[n_keypoints][2]
 *   uint32 src_indices[rows_capacity]; float tgt_xy[rows_capacity][2]; float flow_err[rows_capacity]
 * with rows_capacity = n_keypoints * n_targets (rows beyond row_offset[n_targets] are unspecified).
 * Passing NULL detaches the log.  Appending fails with PC_E_CAPACITY when the buffer is full. */
#define PC_LOG_MAGIC 0x50434c4f47303031ll /* "PCLOG001" */
int pc_analyzer_set_device_log(pc_analyzer* a, void* d_log, size_t capacity_bytes);
int pc_analyzer_device_log_used(const pc_analyzer* a, size_t* bytes);
/* The jobs submitted from now on append to another buffer, from its offset 0, without waiting for anything (set_device_log
 * synchronises the streams): the multi-GPU driver hands the log over piece by piece while the analysis keeps running
 * (analysis_driver.cc, polychase_amd/analyze.py -- the store of cpp/opticalflow.cc:149-151 overlapped with the analysis).
 * The bytes of the previous buffer are complete once the last job submitted into it has been collected.
 * A job whose record does not fit is refused by pc_analyzer_submit with PC_E_CAPACITY BEFORE anything of it is enqueued:
 * the caller can redirect the log and submit the same frame1 again. */
int pc_analyzer_redirect_device_log(pc_analyzer* a, void* d_log, size_t capacity_bytes);
/* enabled = 0: the records of the following jobs are not downloaded to pinned host memory (a rank whose records leave
 * through the device log only); pc_analyzer_collect then returns counts and NULL array pointers.  Default 1. */
int pc_analyzer_set_host_records(pc_analyzer* a, int enabled);
/* Wait for the oldest submitted job.  Pointers stay valid until the job slot is reused, i.e. for
 * the next max_jobs-1 submits. */
int pc_analyzer_collect(pc_analyzer* a, pc_frame_result* out);

/* ---- "Track Sequence" path (cpp/tracker.cc:36-131): batched ray casting + PnP accumulation ---- */
typedef struct pc_mesh pc_mesh;
typedef struct pc_pnp_problem pc_pnp_problem;

/* Mesh of AcceleratedMesh (cpp/ray_casting.cc:21-63): vertices n_vertices x 3 f32 row-major,
 * triangles n_triangles x 3 u32 (host pointers; copied to the GPU). */
int pc_mesh_create(pc_context* ctx, const float* vertices, int n_vertices, const uint32_t* triangles,
                   int n_triangles, pc_mesh** out);
/* masked_triangles bitset (cpp/geometry.h:58-66), n_words >= ceil(n_triangles / 32). */
int pc_mesh_set_mask(pc_context* ctx, pc_mesh* mesh, const uint32_t* mask_words, int n_words);
void pc_mesh_destroy(pc_mesh* mesh);

/* Camera of GetRayObjectSpace (cpp/ray_casting.h:53-63): inv = (view * model)^-1. */
typedef struct pc_ray_camera {
    float dir_matrix[9]; /* inv.block<3,3>(0,0), row-major */
    float origin[3];     /* inv.col(3).head<3>() */
    float fx, fy, cx, cy;
    float unproject_sign; /* CameraIntrinsics::Unproject: +1 OpenCV convention, -1 OpenGL (types.h:95-98) */
} pc_ray_camera;

/* RayCast(accel_mesh, scene_transform, pos, check_mask) (cpp/ray_casting.cc:123-133) for n pixel
 * positions in one launch.  Closest hit, no back-face culling; a masked closest triangle is a miss.
 * Host in/out: xy n x 2; hit n bytes; pos n x 3 (barycentric point, object space); prim n;
 * uvt n x 3 = (u, v, t). */
int pc_raycast_pixels(pc_context* ctx, const pc_mesh* mesh, const pc_ray_camera* cam, const float* xy, int n,
                      int check_mask, uint8_t* hit, float* pos, uint32_t* prim, float* uvt);
/* The same result by an exhaustive sweep over every triangle instead of the hierarchy pc_mesh_create
 * builds (Embree's BVH in the reference, ray_casting.cc:23-63): the validation path of that hierarchy. */
int pc_raycast_pixels_sweep(pc_context* ctx, const pc_mesh* mesh, const pc_ray_camera* cam, const float* xy, int n,
                            int check_mask, uint8_t* hit, float* pos, uint32_t* prim, float* uvt);

/* ---- 3D-2D correspondences of the frame being solved, built and kept on the GPU (cpp/tracker.cc:52-97) ----
 * SolveFrame gathers, per source frame that already has a pose, the matches of the flow source -> frame: the
 * source keypoint is cast onto the mesh under the source camera (tracker.cc:64-78), a hit is moved to world space
 * with the model matrix (:80-82) and paired with the tracked position (:86).  pc_corr_set_append does that for one
 * source in one pass on the device -- gather, ray cast, transform, order-preserving compaction -- and appends to
 * the set; nothing but a counter comes back to the host.  The PnP problem then reads the set in place. */
typedef struct pc_corr_set pc_corr_set;
int pc_corr_set_create(pc_context* ctx, pc_corr_set** out);
void pc_corr_set_destroy(pc_corr_set* set);
int pc_corr_set_clear(pc_context* ctx, pc_corr_set* set);
/* Makes a set fit for an unrelated run without giving its device memory back: waits for the set's own work (its copy stream
 * and the context's stream), forgets the cached keypoint arrays (they are keyed by frame id: another database may use the same
 * ids) and empties the set.  The host side parks ONE set per process between TrackSequence calls (csrc/host/track_sequence.cc):
 * creating the set's streams, page-locked words and device arrays is 6-8 ms of a call (cpp/tracker.cc:133-192 has no such
 * state: the reference allocates per frame on the host). */
int pc_corr_set_recycle(pc_context* ctx, pc_corr_set* set);
/* keypoints_xy: n_keypoints x 2 of the source frame; src_idx / tgt_xy: the n_matches rows of the flow
 * (src_keypoints_indices, tgt_keypoints).  Host pointers (pinned memory from pc_host_buffer_alloc is copied
 * asynchronously: it must stay untouched until pc_corr_set_size or a PnP call has returned).  keypoints_key >= 0
 * names the keypoint array (e.g. the source frame id): the set keeps the device copies of the last 16 keys and
 * skips the upload when it sees one again (a frame is a source for up to 8 targets); -1 = always upload.
 * model_matrix: row-major 4x4, rows 0-2 used.  An index >= n_keypoints is an error (tracker.cc:61). */
int pc_corr_set_append(pc_context* ctx, pc_corr_set* set, const pc_mesh* mesh, const pc_ray_camera* cam,
                       const float* model_matrix, long long keypoints_key, const float* keypoints_xy, int n_keypoints,
                       const uint32_t* src_idx, const float* tgt_xy, int n_matches, int check_mask);
/* number of correspondences so far (waits for the appends). */
int pc_corr_set_size(pc_context* ctx, pc_corr_set* set, int* n);
/* debugging / tests: world points n x 3 and image points n x 2, in append order. */
int pc_corr_set_download(pc_context* ctx, pc_corr_set* set, float* world_xyz, float* image_xy);
/* A PnP problem over the set's arrays, no copy: the set must stay unchanged while the problem is alive. */
int pc_pnp_problem_from_set(pc_context* ctx, pc_corr_set* set, pc_pnp_problem** out);

/* PnPProblem (cpp/pnp/pnp_problem.h:11-142): object points X n x 3, image points x n x 2,
 * optional per-residual weights (NULL = 1). Host pointers; copied to the GPU once per frame. */
int pc_pnp_problem_create(pc_context* ctx, const float* X, const float* x, const float* weights, int n,
                          pc_pnp_problem** out);
void pc_pnp_problem_destroy(pc_pnp_problem* prob);

typedef struct pc_pnp_params {
    float R[9];               /* rotation matrix of the pose (row-major) */
    float t[3];
    float fx, fy, cx, cy, aspect_ratio;
    int convention_opencv;    /* CameraConvention::OpenCV ? 1 : 0 */
    int optimize_focal_length, optimize_principal_point;
    int loss_type;            /* BundleOptions::LossType: 0 TRIVIAL, 1 HUBER, 2 CAUCHY */
    float loss_scale;
} pc_pnp_params;

/* LevMarqDenseSolver::BuildNormalEquations (cpp/pnp/lev_marq.h:231-297) without the diagonal
 * clamp: JtJ lower triangle packed row-major (45), Jtr (9), number of valid residuals. */
int pc_pnp_normal_equations(pc_context* ctx, const pc_pnp_problem* prob, const pc_pnp_params* params,
                            float* jtj_lower45, float* jtr9, int* valid);
/* The same sweep also returns TotalCost of `params` (bit for bit what pc_pnp_total_cost returns): the LM loop
 * (lev_marq.h:132-228) gets a candidate's cost and -- should the step be accepted -- the next iteration's normal
 * equations from one launch and one read-back instead of two. */
int pc_pnp_normal_equations_cost(pc_context* ctx, const pc_pnp_problem* prob, const pc_pnp_params* params,
                                 float* jtj_lower45, float* jtr9, int* valid, float* cost);
/* ---- the whole of SolvePnPIterative on the device (cpp/pnp/solvers.cc:11-71, lev_marq.h:132-228) ----
 * The LM loop alternates residual sweeps with 9x9 algebra that decides what to evaluate next; driven from the
 * host every decision costs a launch + read-back + synchronisation.  pc_pnp_solve keeps the solver's state in
 * device memory: a one-lane kernel takes the decision (same fp32 arithmetic, same order of operations), the host
 * enqueues [sweep, reduce, decide] rounds without waiting and reads the state back once per batch of rounds. */
typedef struct pc_pnp_camera {
    float q_xyzw[4];          /* pose rotation (Eigen order) */
    float t[3];
    float fx, fy, cx, cy, aspect_ratio;
    int convention_opencv;
} pc_pnp_camera;
typedef struct pc_pnp_solve_options {
    int max_iterations;                                   /* BundleOptions (pnp/types.h:200-215) */
    float initial_lambda, min_lambda, max_lambda, gradient_tol, step_tol;
    int loss_type;
    float loss_scale;
    int optimize_focal_length, optimize_principal_point;
    float f_low, f_high, cx_low, cx_high, cy_low, cy_high; /* CameraIntrinsics::GetBounds (types.h:156-192) */
    float max_inlier_error;                               /* <= 0: no inlier pass */
    int rounds_hint;                                      /* rounds enqueued before the first read-back (0 = default) */
} pc_pnp_solve_options;
typedef struct pc_pnp_solve_result {
    pc_pnp_camera camera;
    int iterations, invalid_steps;                        /* BundleStats (pnp/types.h:217-225) */
    float initial_cost, cost, lambda, step_norm, grad_norm;
    int inliers;                                          /* residuals below max_inlier_error (solvers.cc:31-47) */
} pc_pnp_solve_result;
int pc_pnp_solve(pc_context* ctx, pc_pnp_problem* prob, const pc_pnp_camera* initial, const pc_pnp_solve_options* options,
                 pc_pnp_solve_result* result);
/* ---- SolveFrame in one call (cpp/tracker.cc:36-131): the correspondences of EVERY source frame + SolvePnPIterative ----
 * Two launches and one wait per frame: one ray-cast launch over the matches of all sources (each under its own
 * camera; world points stay on the GPU, no compaction), then the whole Levenberg-Marquardt loop and the inlier pass as
 * ONE persistent launch -- the workgroups keep their correspondences, meet at a grid barrier once per evaluated
 * parameter set, one workgroup takes the solver's decision (lev_marq.h:146-221) and the rounds stop when it is done; the
 * result is written to pinned host memory by the kernel.  (pc_corr_set_append + pc_pnp_solve -- 12 + 28 launches per
 * frame -- remain as the building blocks and as the cross-check: same world points bit for bit, poses equal to fp32
 * summation order.)
 * sources[k]: camera of source frame k (already solved), its keypoints (cached by keypoints_key like
 * pc_corr_set_append) and where its matches -- src_keypoints_indices (n_matches x uint32 at idx_offset) and
 * tgt_keypoints (n_matches x 2 floats at tgt_offset; byte offsets, 4- / 8-aligned) -- lie inside the block `matches`
 * of matches_bytes bytes: the block travels to the GPU in ONE transfer (host memory; pc_host_buffer_alloc memory is
 * fetched by the copy engine without a staging copy).
 * options->optimize_*: what the caller ASKS for; intrinsics are only optimised with more than 3 correspondences
 * (pnp_problem.h:34-35), decided on the device where the count is known.
 * result->n_correspondences < 3: "not enough features" (tracker.cc:95-97), nothing was solved. */
typedef struct pc_track_source {
    pc_ray_camera cam;
    long long keypoints_key;
    const float* keypoints_xy;    /* host, n_keypoints x 2 */
    int n_keypoints;
    int n_matches;
    size_t idx_offset, tgt_offset;
} pc_track_source;
typedef struct pc_track_solve_result {
    pc_pnp_solve_result pnp;
    int n_matches;                /* rows of all sources */
    int n_correspondences;        /* ... whose ray hit the (unmasked) mesh */
    int rounds;                   /* parameter sets evaluated by the persistent launch */
    /* where the persistent launch spent its time, measured by its deciding workgroup in 100 MHz ticks and summed over the
     * rounds: [0] residual sweep + publishing the partial sums, [1] waiting for the other workgroups, [2] adding the
     * partials, [3] the decision (9x9 algebra; by a wavefront since round 6), [4] publishing it, [5] fetching the next parameters,
     * [6] inlier pass, [7] the whole launch */
    unsigned lm_ticks[8];
    unsigned long long lm_begin_tick, lm_end_tick;   /* the GPU's 100 MHz clock when the LM kernel began / handed over its result */
} pc_track_solve_result;
int pc_track_solve_frame(pc_context* ctx, pc_corr_set* set, const pc_mesh* mesh, const float* model_matrix, int check_mask,
                         const pc_track_source* sources, int n_sources, const void* matches, size_t matches_bytes,
                         const pc_pnp_camera* initial, const pc_pnp_solve_options* options, pc_track_solve_result* result);
/* The same in three steps, so that a caller overlaps the host's part and the transfer of frame f + 1 with the launches of
 * frame f (csrc/host/track_sequence.cc): upload = the matches block (and the keypoints of sources the set has not cached yet)
 * on the set's own copy stream, into the next of three device blocks -- callable while frames are in flight, before the
 * cameras of its sources are known; launch = the two launches behind that upload, for the block uploaded last; finish =
 * wait + result.  `matches` and the keypoint arrays must stay untouched until the frame's finish. */
int pc_track_frame_upload(pc_context* ctx, pc_corr_set* set, const void* matches, size_t matches_bytes, const pc_track_source* sources,
                          int n_sources);
int pc_track_frame_launch(pc_context* ctx, pc_corr_set* set, const pc_mesh* mesh, const float* model_matrix, int check_mask,
                          const pc_track_source* sources, int n_sources, const pc_pnp_camera* initial, const pc_pnp_solve_options* options);
/* Allocates what the first tracked frame of a clip with up to n_matches matches per frame and n_keypoints keypoints per source
 * frame would otherwise allocate inside pc_track_frame_upload / _launch (the set's copy stream and event, the three match blocks,
 * the per-match arrays, the barrier words, the page-locked result words, `n_cached` keypoint arrays): 6-8 ms of a process's
 * first TrackSequence call.  Everything still grows on demand. */
int pc_corr_set_reserve(pc_context* ctx, pc_corr_set* set, int n_matches, int n_keypoints, int n_cached);
/* The launch with its inputs taken from the launch enqueued JUST BEFORE it on this set, on the device: up to TWO frames may be in
 * flight (finished in launch order), so that the GPU goes from frame f to frame f + 1 without the host in between -- SolveFrame of
 * f + 1 needs the pose of f twice (tracker.cc:43-50: f is one of its source frames; :111-119: f's pose is its initial guess), and
 * both are left on the device by f's launch.
 *   chained_source  index into `sources` of the source frame whose camera (pc_track_source::cam is ignored for it) is the result
 *                   of the previous launch, or -1;
 *   chain_initial   != 0: the initial camera is the previous launch's result (`initial` is ignored);
 * both require that a launch of this set has run since it was created / recycled.  Sources of a second frame in flight must
 * have keypoints_key >= 0.  If the previous launch fails, this one runs on whatever it left (bounded by max_iterations) and the
 * caller, who learns of the failure first, drops its result. */
int pc_track_frame_launch_chained(pc_context* ctx, pc_corr_set* set, const pc_mesh* mesh, const float* model_matrix, int check_mask,
                                  const pc_track_source* sources, int n_sources, int chained_source, const pc_pnp_camera* initial,
                                  int chain_initial, const pc_pnp_solve_options* options);
/* waits for the OLDEST frame in flight */
int pc_track_frame_finish(pc_context* ctx, pc_corr_set* set, pc_track_solve_result* result);
/* debugging / tests: (world x, y, z, hit ? 1 : 0) of the first n matches of the last pc_track_solve_frame, in match order
 * (source after source). */
int pc_track_download_points(pc_context* ctx, pc_corr_set* set, int n, float* world_xyzw);

/* LevMarqDenseSolver::TotalCost (lev_marq.h:316-356) and the inlier count of SolvePnPIterative
 * (cpp/pnp/solvers.cc:31-47) in one pass. */
int pc_pnp_total_cost(pc_context* ctx, const pc_pnp_problem* prob, const pc_pnp_params* params,
                      float max_inlier_error_sq, float* cost, int* valid, int* inliers);

/* ---- "Refine Sequence" path (cpp/refiner.cc:199-725, cpp/pnp/lev_marq.h:391-871) ----
 * The segment's keypoints and flows live on the GPU; per LM iteration the host sends the camera
 * trajectory and receives the cost or the per-edge normal-equation blocks. */
typedef struct pc_refine_problem pc_refine_problem;

typedef struct pc_refine_desc {
    int n_frames;                 /* frames of the segment, index 0 = first frame */
    int n_edges;                  /* flows (image_id_from -> image_id_to) inside the segment */
    const int32_t* kp_offset;     /* [n_frames + 1]: keypoints of frame f = kp_xy[kp_offset[f] .. kp_offset[f+1]) */
    const float* kp_xy;           /* all (bbox-filtered) keypoints, x y */
    const int32_t* edge_src;      /* [n_edges] frame index of image_id_from */
    const int32_t* edge_tgt;      /* [n_edges] frame index of image_id_to */
    const int32_t* edge_offset;   /* [n_edges + 1] into the residual arrays */
    const uint32_t* res_src_kp;   /* per residual: keypoint index within the source frame */
    const float* res_tgt_xy;      /* per residual: tracked position in the target frame */
    const float* edge_weight;     /* [n_edges] GlobalRefinementProblem::EdgeWeight (refiner.cc:596-599) */
    float model_matrix[16];       /* object -> world, row-major */
    float model_matrix_inv[16];
    int block_len;                /* parameters per camera: 6, or 9 when intrinsics are optimised */
    int optimize_focal_length, optimize_principal_point;
} pc_refine_desc;

/* one camera of the trajectory, 20 floats */
typedef struct pc_refine_camera {
    float R[9];
    float t[3];
    float fx, fy, cx, cy, aspect_ratio;
    float unproject_sign;         /* +1 OpenCV, -1 OpenGL */
    float reserved[2];
} pc_refine_camera;

int pc_refine_problem_create(pc_context* ctx, const pc_mesh* mesh, const pc_refine_desc* desc,
                             pc_refine_problem** out);
/* The same problem with its three large arrays handed over in pieces: the keypoints and the residuals of consecutive runs of
 * frames (what the reader threads of CachedDatabase's replacement each hold -- a 300-frame 1080p segment is 1.2 GB, and joining
 * the pieces on the host first costs as much as reading them).  desc->kp_xy, res_src_kp and res_tgt_xy must be NULL; the parts,
 * in order, make up kp_offset[n_frames] keypoints and edge_offset[n_edges] residuals.  A residual that names a keypoint its
 * source frame does not have is found on the device and refused like pc_refine_problem_create refuses it. */
typedef struct pc_refine_part {
    const float* kp_xy;           /* n_keypoints (x, y) */
    int64_t n_keypoints;
    const uint32_t* res_src_kp;   /* n_residuals */
    const float* res_tgt_xy;      /* n_residuals (x, y) */
    int64_t n_residuals;
} pc_refine_part;
int pc_refine_problem_create_parts(pc_context* ctx, const pc_mesh* mesh, const pc_refine_desc* desc, const pc_refine_part* parts,
                                   int n_parts, pc_refine_problem** out);
void pc_refine_problem_destroy(pc_refine_problem* prob);
/* LevMarqSparseSolver::TotalCost (lev_marq.h:773-824) over RefinementProblemBase::Evaluate
 * (refiner.cc:274-361); updates the per-keypoint triangle cache like the reference does. */
int pc_refine_total_cost(pc_context* ctx, pc_refine_problem* prob, const pc_refine_camera* cameras,
                         int loss_type, float loss_scale, double* cost);
/* LevMarqSparseSolver::BuildNormalEquations (lev_marq.h:653-771) without the scatter: per edge the
 * lower triangle of the (2B x 2B) block (packed row-major, (2B)(2B+1)/2 doubles) followed by the 2B
 * gradient entries, normalised by the edge's valid-residual count (returned in edge_valid).
 * Jacobians are evaluated in fp32 like the reference; the sums are fp64 so that J^T J stays positive
 * semi-definite to rounding (long weakly-damped chains need it). */
int pc_refine_normal_equations(pc_context* ctx, pc_refine_problem* prob, const pc_refine_camera* cameras,
                               int loss_type, float loss_scale, double* edge_blocks, int* edge_valid);

/* GPU time of the sweeps of this problem so far: HIP events on the context's stream around the kernel of every
 * pc_refine_total_cost / pc_refine_normal_equations call (launches, summed milliseconds).  Any pointer may be NULL. */
int pc_refine_problem_timing(const pc_refine_problem* prob, int* cost_launches, double* cost_ms, int* normal_eq_launches,
                             double* normal_eq_ms);

/* ---- record exchange between the ranks of a multi-GPU analysis (one process per GPU) --------------------------------
 * The stitch of the flow database (the reference's store, cpp/opticalflow.cc:149-151, sharded): a rank exports a
 * device buffer, its peers map it INTO THEIR OWN device's address space (HIP IPC) and write their record pieces into it with
 * device-to-device copies on a stream of their own device -- xGMI is point to point, the copy engines move the bytes, no
 * kernel and no queue on the peer's GPU is involved.  polychase_amd/distributed.py: PeerLogStitch. */
#define PC_PEER_HANDLE_BYTES 64
int pc_peer_buffer_alloc(int device_index, size_t bytes, void** device_ptr);           /* hipMalloc on that device */
int pc_peer_buffer_free(int device_index, void* device_ptr);
int pc_peer_buffer_export(int device_index, void* device_ptr, unsigned char handle[PC_PEER_HANDLE_BYTES]);
/* maps a peer's exported buffer for `device_index` (this process's GPU); *device_ptr is valid for copies issued here */
int pc_peer_buffer_open(int device_index, const unsigned char handle[PC_PEER_HANDLE_BYTES], void** device_ptr);
int pc_peer_buffer_close(int device_index, void* device_ptr);
/* hipMemcpyAsync(dst, src, bytes, device-to-device) on `stream` (a hipStream_t of device_index; NULL = the null stream) */
int pc_peer_copy_async(int device_index, void* dst, const void* src, size_t bytes, void* stream);
/* blocking device-to-host copy (verification of what the peers wrote) */
int pc_peer_buffer_download(int device_index, void* dst_host, const void* src_device, size_t bytes);

/* ---- the same exchange over RCCL, for C / C++ hosts (no torch): api_comm.hip ---------------------------------------------
 * One communicator per rank (= process = GPU).  Rank 0 makes the id (ncclGetUniqueId) and hands its PC_COMM_ID_BYTES bytes
 * to the other ranks by any means (csrc/host/multi_gpu.cc: a TCP socket); every rank then calls pc_comm_create, which is
 * collective (ncclCommInitRank).  librccl.so.1 is loaded at the first of these calls: PC_E_NO_DEVICE when it is missing.
 *   pc_comm_all_gather_log   BASELINE.json's "RCCL all-gather of the flow DB": every rank contributes `bytes` bytes at
 *                            `piece` (device memory; sizes may differ) and receives all ranks' pieces in `recv` (device
 *                            memory, world * slot_bytes; rank r's piece at r * slot_bytes, sizes_host[r] bytes of it valid).
 *                            Two ncclAllGather calls (the sizes as uint64, then the payload padded to slot_bytes) on the
 *                            communicator's own stream; returns when both are complete.
 *   pc_comm_send / _recv     the ordered gather of the product: ncclSend / ncclRecv of exactly `bytes` bytes of device
 *                            memory, blocking until the transfer is complete.
 * What they replace in the reference: nothing it has (cpp/opticalflow.cc:209-321 is one process) -- the hand-over of
 * opticalflow.cc:149-151 (flows -> database) when the frames were analysed on another GPU. */
typedef struct pc_comm pc_comm;
#define PC_COMM_ID_BYTES 128
int pc_comm_unique_id(void* id /* [PC_COMM_ID_BYTES] */);
int pc_comm_create(pc_context* ctx, const void* id /* [PC_COMM_ID_BYTES] */, int world_size, int rank, pc_comm** out);
void pc_comm_destroy(pc_comm* comm);
int pc_comm_world_size(const pc_comm* comm);
int pc_comm_rank(const pc_comm* comm);
int pc_comm_all_gather_log(pc_comm* comm, const void* piece_device, uint64_t bytes, void* recv_device, uint64_t slot_bytes,
                           uint64_t* sizes_host /* [world_size] */);
int pc_comm_send(pc_comm* comm, const void* src_device, uint64_t bytes, int dst_rank);
int pc_comm_recv(pc_comm* comm, void* dst_device, uint64_t bytes, int src_rank);

#ifdef __cplusplus
}
#endif
#endif /* POLYCHASE_HIP_H_ */
