// kernels.hpp -- host-side launchers of the gfx950 kernels (implemented in kernels_*.hip).
#pragma once

#include "bvh.hpp"
#include "common.hpp"

namespace pc {

// Issue priority the helper kernels (frame preparation, compaction) are launched with from this thread: 0 or 1
// (common.hpp: helper_priority).  Set by the analyzer around its enqueues; stage-level calls leave it at 0.
int helper_prio_arg();
void set_helper_prio(int hi);

// ---- kernels_image.hip ----
// K1: RGB u8 -> gray u8 written straight into the level-0 plane interior (cvtColor RGB2GRAY).
void launch_rgb2gray(const uint8_t* rgb, size_t rgb_pitch, const Level& l0, hipStream_t s);
// float32 RGB / RGBA (numpy `(x * 255).astype(uint8)` semantics) -> gray u8 in the level-0 interior
void launch_rgbf32_to_gray(const float* rgb, size_t rgb_pitch, int channels, const Level& l0, hipStream_t s);
// gray u8 (arbitrary pitch) -> level-0 interior
void launch_copy_gray(const uint8_t* gray, size_t gray_pitch, const Level& l0, hipStream_t s);
// K7: pyrDown 5x5 (src interior -> dst interior)
void launch_pyrdown(const Level& src, const Level& dst, hipStream_t s);
// REFLECT_101 border of width `win` around the interior
void launch_border(const Level& l, int win, hipStream_t s);
// padded u8 plane -> the uint16 (pixel << 7) plane the LK kernel reads (needs the border to be filled)
void launch_widen(const Level& l, int win, hipStream_t s);
// K6: Scharr derivative plane of the interior (needs the 1-px border to be filled)
void launch_scharr(const Level& l, hipStream_t s);

// ---- kernels_pyramid.hip ----
// One fused kernel per pyramid level: the level's pixels (gray conversion of the frame for level 0, pyrDown of the
// parent's padded plane otherwise) -> u8 plane + REFLECT_101 padding + uint16 plane + Scharr plane.
enum LevelSourceKind { SRC_PYR = 0, SRC_RGB8 = 1, SRC_GRAY8 = 2, SRC_RGBF32 = 3 };
struct LevelSource {
    int kind;
    const uint8_t* src;   // level 0: the frame on the device (u8 RGB, u8 gray or float32 RGB / RGBA)
    size_t src_pitch;     // bytes per frame row
    int channels;         // floats per pixel (SRC_RGBF32)
    int aligned;          // src and src_pitch are multiples of 4: the u8 sources are read as dwords
    Level parent;         // SRC_PYR
    uint32_t* clear;      // clear_words words zeroed by the launch (the detection's counters: saves the fill command
    int clear_words;      // in front of the detection chain), or null
};
// requires out.w > win + 1 and out.h > win + 1 (single reflection in the padding); smaller levels take the unfused kernels
void launch_level(const LevelSource& in, const Level& out, int win, hipStream_t s);

// ---- kernels_gftt.hip ----
struct GfttGrid {
    int rows, cols;      // grid_rows, grid_cols (>= 1)
    int cell_w, cell_h;  // ceil(W/cols), ceil(H/rows)
};
constexpr int kMaxGridCells = 1024;   // grid_rows x grid_cols (not bound by the reference's module -- polychase_pybind.cc:128-136 leaves the 4 x 4 default -- but free in the C ABI)
// K2: cornerMinEigenVal (block 3, Sobel 3) of the level-0 interior + per-cell max (ordered keys,
// cell_max must be zeroed first).
// sobel_fma, bit 0: the column pass of Dx as ONE fused multiply-add (PC_ARITH_SOBEL_FMA: the AVX2 dispatch of OpenCV's filter);
// bit 1: the row pass of Dy as a fused chain (PC_ARITH_SOBEL_ROW_FMA)
void launch_min_eig(const Level& l0, float* eig, const GfttGrid& g, uint32_t* cell_max, int sobel_fma, hipStream_t s);
// K2 for any block_size, any gradient_size (3, 5, 7: Sobel; -1: Scharr) and for cornerHarris (gftt.cc:31-36): two plain kernels;
// cov = 3 * w * h floats of scratch; box_rows = 3 * w * h doubles (the box filter's row sums: block^2 -> 2 * block reads per pixel,
// the same additions in the same order) or null.  false: not an aperture OpenCV has
constexpr int kBoxRowsFromBlock = 12;   // callers pass box_rows from this block size on
bool launch_corner_response(const Level& l0, float* eig, float* cov, double* box_rows, const GfttGrid& g, uint32_t* cell_max, int block_size, int gradient_size,
                            bool harris, double harris_k, int sobel_fma, hipStream_t s);
// K3: per-cell THRESH_TOZERO + 3x3 dilate + strict-interior local maxima -> 64-bit keys
// (ordered(value) << 32 | y*w+x) appended to `keys` (capacity `cap`), count in *counter; cstate (w*h bytes): 1 at
// candidates, 0 elsewhere, every pixel written; sort_params[2] / hist[kSortBuckets]: value range and per-bucket counts
// of the candidates for launch_bucket_sort (hist zeroed by the caller).  The workgroup that finishes last scans the
// bucket counts into bucket_offsets[kSortBuckets + 1] (`ticket`: pc::last_workgroup_words(workgroups) zeroed words); bin_hist (may be null):
// bin_num_tiles(w, h) words zeroed here for launch_suppress_and_compact.
constexpr int kSortBuckets = 8192;
void launch_nms(const float* eig, int w, int h, const GfttGrid& g, const uint32_t* cell_max, double quality_level,
                unsigned long long* keys, uint32_t cap, uint32_t* counter, uint8_t* cstate, uint32_t* sort_params, uint32_t* hist,
                uint32_t* ticket, uint32_t* bucket_offsets, uint32_t* bin_hist, hipStream_t s);
// K4: the candidates in descending (value, address) order -> out; no count on the host (scan of the bucket counts,
// scatter into bucket order via `scratch`, rank sort per bucket).  offsets[kSortBuckets + 1], cursor[kSortBuckets]
// (zeroed by the caller); n_launch sizes the scatter's grid (it walks all candidates whatever the grid);
// *overflow |= 1 when a bucket exceeds the fast path (then: sort_keys_desc).
void launch_bucket_sort(const unsigned long long* keys, uint32_t cap, uint32_t n_launch, const uint32_t* counter, const uint32_t* sort_params,
                        const uint32_t* offsets, uint32_t* cursor, unsigned long long* scratch, unsigned long long* out,
                        uint32_t* overflow, hipStream_t s);
// offsets[n_offsets]: the (dx, dy) of the suppression neighbourhood; row_hw[2 R + 1]: the same set as half-widths per row
// (row dy holds |dx| <= row_hw[dy + R], -1: none), the form the kernel walks.
// K5: exact greedy min-distance suppression (gftt.cc:100-164) over the candidates SORTED by priority (keys descending),
// then the accepted ones in priority order -> float2 keypoints (truncated to max_corners if > 0).  TWO launches:
//   suppression: cstate becomes 2 (accepted) / 3 (rejected) at every candidate; its last workgroup scans the accepted
//     counts of the workgroups (per_block: suppress_num_blocks(n_max) + 1 words of scratch) and writes *n_out;
//   compaction: keypoints written; with bin_hist (bin_num_tiles(w, h) zeroed words, see launch_nms) it counts the
//     keypoints per 64x64 tile and its last workgroup scans the counts (= the input of launch_spatial_bins_counted).
// The number of candidates is min(*n_dev, n_max) (n_dev may be null); the launches cover n_max; *overflow |= 4 when
// *n_dev exceeds n_max.  suppress == false: min_distance < 1, everything is accepted (gftt.cc:165-181).  *stuck != 0
// afterwards means the spin bound hit.  tickets: two zeroed arrays (pc::last_workgroup_words of the workgroup count), ticket_stride apart.
int suppress_num_blocks(uint32_t n);
void launch_suppress_and_compact(const unsigned long long* keys, uint32_t n_max, const uint32_t* n_dev, int w, int h, const float* eig,
                                 uint8_t* cstate, const int2* offsets, int n_offsets, const int* row_hw, int R, bool suppress,
                                 uint32_t* per_block,
                                 uint32_t* stuck, uint32_t max_corners, float2* xy, uint32_t* n_out, uint32_t* bin_hist,
                                 uint32_t* overflow, uint32_t* tickets, uint32_t ticket_stride, double large_min_distance, uint32_t* large_grid,
                                 hipStream_t s);
// min_distance above this: the neighbourhood table is not built; large_grid (suppress_large_grid_words words of scratch) and
// large_min_distance select the one-wavefront greedy kernel (the reference's loop against a grid of accepted corners)
constexpr double kSuppressMaxTableRadius = 64.0;
int suppress_large_grid_words(int w, int h, double min_distance);
// K4 fallback: descending radix sort of the candidate keys (rocPRIM), count on the host.  temp may be null to query bytes.
hipError_t sort_keys_desc(void* temp, size_t& temp_bytes, unsigned long long* keys_in,
                          unsigned long long* keys_out, uint32_t n, hipStream_t s);

// ---- kernels_lk.hip ----
constexpr int kMaxLevels = 16;   // == PC_MAX_LEVELS (internal.hpp asserts it)
struct LKParams {
    Level src[kMaxLevels];             // frame1 levels
    const uint8_t* tgt[8][kMaxLevels]; // [target][level] interior origins (same geometry as src)
    const uint16_t* tgt16[8][kMaxLevels];  // the same levels as uint16 (pixel << 7) planes (Level::img16)
    int n_targets;
    int max_level;            // effective (min over pyramids)
    int n;                    // number of keypoints
    const float2* pts;
    const uint32_t* perm;     // visiting order (spatially binned keypoint indices) or null
    int blocks_per_xcd;       // filled in by launch_lk
    int max_iters;
    double eps_sq;
    float min_eig_thr;
    // lk3: the smallest float x with fl(x / (2 win^2)) >= min_eig_thr (filled in by launch_lk3; NaN: not available).
    // fl(x / c) is monotone in x, so "fl(x / c) < thr" is "x < this" -- the per-level IEEE division becomes a compare.
    float min_eig_num_thr;
    // Raw result records in VISITING order: out_rec[slot * 8 + target] = (next.x, next.y, err, bits(status)); slot s
    // tracks keypoint perm[s].  A wavefront's 16 results are 256 contiguous bytes (they used to be 48 scattered
    // 8/4/1-byte stores through the permutation: 8x write amplification, profiles/lk_hbm_traffic.json of round 1).
    float4* out_rec;
    unsigned long long* prof; // per-phase cycle sums (PC_LK_PROFILE builds), or null
    // "every workgroup of this launch has been handed out": the workgroup with the highest index stores gate_value here
    // when it starts (workgroups are dispatched in index order).  launch_lk_gate makes a stream wait for it, so that
    // the next launch fills the tail of this one -- and no more than the tail.  May be null.
    uint32_t* gate;
    uint32_t gate_value;
    int x86_order;            // PC_ARITH_LK_X86_ORDER: fp32 lane sums in the order of OpenCV's SSE path (X86 = true in both kernels)
    // lk3, x86_order, diagnostics (or null): [0] iterations decided by the exactness proof, [1] iterations evaluated in the
    // x86 order, [2] (keypoint, level) pairs, [3] of those with the structure tensor evaluated in the x86 order
    unsigned long long* x86_stats;
};
// the stream waits (one idle wavefront) until *gate has reached `value` (wrap-around compare); gives up after ~50 ms
// and stores 1 to *timed_out (device-visible host memory, may be null)
void launch_lk_gate(const uint32_t* gate, uint32_t value, uint32_t* timed_out, hipStream_t s);
__device__ __forceinline__ void lk_signal_dispatched(const LKParams& p) {
    if (p.gate && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
        __hip_atomic_store(p.gate, p.gate_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
constexpr int kRecStride = 8;   // records per slot (= PC_MAX_TARGETS)
// K8-K10: pyramidal LK, one 16-lane DPP row per (keypoint, target).  Returns false if the window
// size is unsupported.
bool launch_lk(const LKParams& p, int win, hipStream_t s);
// two keypoints per wavefront on the uint16 planes, dword-per-position LDS regions (kernels_lk3.hip); windows 4..11 (launch_lk sends 11 to lk4)
bool launch_lk3(const LKParams& p, int win, hipStream_t s);
// one keypoint per wavefront, 8 lanes per target, on the uint16 planes (lk4_kernel.hpp); windows 3 and 11..31, spread over
// three translation units (kernels_lk4{a,b,c}.hip)
bool launch_lk4a(const LKParams& p, int win, hipStream_t s);
bool launch_lk4b(const LKParams& p, int win, hipStream_t s);
bool launch_lk4c(const LKParams& p, int win, hipStream_t s);
bool lk_profile_enabled();   // library compiled with -DPC_LK_PROFILE: LKParams::prof takes 16 words per wavefront
// counting sort of keypoint indices by 64x64 tile -> perm[n]; hist: bin_num_tiles(w, h) words of scratch
int bin_num_tiles(int w, int h);
// slot_of[i] = position of keypoint i in perm (the inverse permutation).  n_dev (may be null): the number of keypoints
// is read from device memory (<= n, which then only sizes the launch).
void launch_spatial_bins(const float2* pts, int n, const uint32_t* n_dev, int w, int h, uint32_t* hist, uint32_t* perm,
                         uint32_t* slot_of, hipStream_t s);
// the same when `hist` already holds the tiles' first positions (launch_suppress_and_compact): the scatter only.
// copy_words > 0: the launch also copies copy_src[0 .. copy_words) to copy_dst (device-visible host memory: the
// detection's counters reach the host without a copy command behind the kernel).
void launch_spatial_bins_counted(const float2* pts, int n, const uint32_t* n_dev, int w, int h, uint32_t* hist, uint32_t* perm,
                                 uint32_t* slot_of, const uint32_t* copy_src, uint32_t* copy_dst, int copy_words, hipStream_t s);

// Ordered compaction of status==1 rows per target (opticalflow.cc:130-147).
// rec / slot_of: the LK kernel's raw records (visiting order) and the inverse visiting order.
// scratch: compact_scratch_words(n, n_targets) words; scratch_fresh: the buffer has not been through a launch_compact
// since it was allocated (its ticket words are then zeroed first; afterwards the launches keep them zero).
// row_offset: [n_targets+1] int64 (device).
constexpr int kCompactTicketWords = 1024;   // pc::last_workgroup_words(blocks) for up to 8.3 M keypoints
constexpr int kCompactMaxKeypoints = 8000000;   // callers refuse more
size_t compact_scratch_words(int n, int n_targets);
void launch_compact(const float4* rec, const uint32_t* slot_of, int n, int n_targets, uint32_t* scratch, bool scratch_fresh,
                    long long* row_offset, uint32_t* out_idx, float2* out_xy, float* out_err, hipStream_t s);
// raw records -> [target][n] arrays in keypoint order (pc_lk_track)
void launch_unpack_records(const float4* rec, const uint32_t* slot_of, int n, int n_targets, float2* xy, uint8_t* status,
                           float* err, hipStream_t s);
int compact_num_blocks(int n);
// keypoints -> packed record buffer (both 16-byte aligned)
void launch_copy_keypoints(const float2* src, float2* dst, int n, hipStream_t s);


// ---- kernels_tracker.hip ----
struct RayCamera {
    float m[9];        // 3x3 block of (view * model)^-1, row-major: camera direction -> object space
    float origin[3];   // translation column of (view * model)^-1
    float fx, fy, cx, cy;
    float sign;        // CameraIntrinsics::Unproject: +1 OpenCV, -1 OpenGL (types.h:95-98)
};
struct PnPParams {
    float R[9];
    float t[3];
    float fx, fy, cx, cy, aspect_ratio;
    int convention_opencv;
    int optimize_focal, optimize_pp;
    int loss_type;     // 0 trivial, 1 Huber, 2 Cauchy (BundleOptions::LossType)
    float loss_scale;
};
// closest hit per pixel through the mesh's LBVH
void launch_raycast(const BvhView& bvh, const uint32_t* mask, int check_mask, const RayCamera& cam, const float2* xy, int n,
                    uint8_t* hit, float* pos, uint32_t* prim, float* uvt, hipStream_t s);
// the same by an exhaustive sweep over all triangles (validation of the hierarchy)
void launch_raycast_sweep(const float* verts, const uint32_t* tris, int n_tris, const uint32_t* mask, int check_mask,
                          const RayCamera& cam, const float2* xy, int n, uint8_t* hit, float* pos, uint32_t* prim,
                          float* uvt, hipStream_t s);
// correspondences of one source frame (tracker.cc:52-92): gather + cast + model transform + order-preserving append
struct CorrModel {
    float m[12];       // rows 0-2 of the model matrix, row-major
};
int corr_num_blocks(int n);
void launch_corr_append(const BvhView& bvh, const uint32_t* mask, int check_mask, const RayCamera& cam, const CorrModel& model,
                        const float2* kps, int n_kps, const uint32_t* src_idx, const float2* tgt, int n, uint8_t* flag,
                        float* world, int* block_counts, int* block_offsets, int* counter, int* bad_index, float* X, float2* x,
                        hipStream_t s);
int pnp_num_blocks(int n);
// out56: [0..44] JtJ lower triangle (row-major packed), [45..53] Jtr, [54] valid residual count, [55] cost
void launch_pnp_normal_eq(const float* X, const float* x, const float* w, int n, const PnPParams& p, float* partials,
                          float* out56, hipStream_t s);
// device-resident LM (pnp_lm.hpp)
struct LmState;
void launch_pnp_lm_rounds(const float* X, const float* x, const float* w, int n, LmState* st, int iterations, float* partials,
                          float* partials4, float* out4, hipStream_t s);
// out4: [0] cost, [1] valid residuals, [2] inliers (r^2 < max_err_sq)
void launch_pnp_cost(const float* X, const float* x, const float* w, int n, const PnPParams& p, float max_err_sq,
                     float* partials, float* out4, hipStream_t s);


// ---- SolveFrame in two launches (kernels_tracker.hip: track_cast_kernel, track_lm_kernel) ----
constexpr int kTrackMaxSources = 8;     // flows into a frame: the skips -8 .. +8 (cpp/opticalflow.cc:76-77)
constexpr int kTrackSyncWords = 64 + 256; // barrier words of track_lm_kernel: zero before the launch, left zero by it
struct TrackSource {                    // one source frame of the frame being solved
    RayCamera cam;                      // the source's camera, object space (GetRayObjectSpace, ray_casting.h:53-63)
    const RayCamera* cam_dev;           // or null.  Not null: the camera is read from device memory -- the source is the frame whose
                                        // LM launch sits in FRONT of this one on the stream and leaves its camera there
                                        // (TrackChainSlot::ray): the host has not seen that pose yet
    const float2* kps;                  // its keypoints (device)
    const uint32_t* idx;                // its matches (device): src_keypoints_indices ...
    const float2* tgt;                  // ... and tgt_keypoints of the flow source -> frame
    int n_kps;
    int begin, n_matches;               // its rows of the frame's arrays pts / obs
    int block_begin;                    // first workgroup of track_cast_kernel that works on them
};
struct TrackCastArgs {
    BvhView bvh;
    const uint32_t* mask;
    int check_mask;
    CorrModel model;
    int n_sources;
    TrackSource src[kTrackMaxSources];
    float4* pts;                        // out, per match: world point + 1, or zeros for a miss
    float2* obs;                        // out, per match: its tracked position (the sources' tgt arrays, one after the other)
    int* bad_index;                     // set when an index lies past its source's keypoints (tracker.cc:61)
};
int track_cast_blocks(int n_matches);
void launch_track_cast(const TrackCastArgs& a, int total_blocks, hipStream_t s);

struct LmConfig;
struct LmCamera;
struct TrackLmOut;
struct TrackLmArgs;
int track_lm_blocks(int n);
void launch_track_lm(const TrackLmArgs& a, hipStream_t s);

// lm_cholesky9 + lm_cholesky9_solve (pnp_lm.hpp) of one system on the device; all pointers device memory
void launch_llt9_debug(const float* a81, const float* b9, float* l81, float* x9, int* ok, hipStream_t s);


// ---- kernels_refiner.hip ----
struct RefineCamera {   // one frame of the trajectory
    float R[9];
    float t[3];
    float fx, fy, cx, cy, aspect;
    float sign;         // +1 OpenCV convention, -1 OpenGL (Unproject / IsBehind)
};
struct RefineProblemView {
    int n_frames, n_edges, n_tris;
    const int* kp_offset;          // [n_frames + 1]
    const float2* kp_xy;           // keypoints of all frames
    const int* edge_src;           // frame index of image_id_from
    const int* edge_tgt;
    const int* edge_offset;        // [n_edges + 1] into the residual arrays
    const uint32_t* res_src_kp;    // keypoint index within the source frame
    const float2* res_tgt_xy;
    const float* edge_weight;
    const uint8_t* frame_fixed;    // first / last frame of the segment: no Jacobian (refiner.cc:611-612)
    uint32_t* prim_cache;          // per keypoint: cached triangle or 0xffffffff (refiner.cc:547-559)
    const float4* tri_plane;       // per triangle: world-space normal, world-space first vertex (launch_refine_tri_planes)
    const float4* tri_verts;       // per triangle: its three vertices, object space (launch_refine_tri_verts)
    const float* verts;
    const uint32_t* tris;
    const uint32_t* mask;
    BvhView bvh;                   // closest-hit ray casts when the cached triangle is missed
    float model[16], model_inv[16];
};
// bad[0] = lowest edge index (or INT_MAX) with a residual that names a keypoint its source frame does not have, bad[1] = one such
// keypoint index of that launch (bad must hold {INT_MAX, 0} before)
void launch_refine_validate(const RefineProblemView& P, int* bad, hipStream_t s);
// plane[2 t], plane[2 t + 1] = world-space normal and first vertex of triangle t (what EvaluateWithJacobian intersects with)
void launch_refine_tri_planes(const RefineProblemView& P, float4* plane, hipStream_t s);
// tri_verts[3 t + k] = vertex k of triangle t
void launch_refine_tri_verts(const RefineProblemView& P, float4* tri_verts, hipStream_t s);
// edge_out[e] = {sum of losses over valid residuals, valid count}, accumulated in fp64
void launch_refine_cost(const RefineProblemView& P, const RefineCamera* cams, int loss_type, float loss_scale,
                        double2* edge_out, hipStream_t s);
// edge_blocks[e]: lower triangle of the (2B x 2B) JtJ pair block (row-major packed) followed by the 2B
// Jtr pair vector (fp32 Jacobians, fp64 sums), normalised by the edge's valid count; edge_valid[e] = that count
void launch_refine_normal_eq(const RefineProblemView& P, const RefineCamera* cams, int loss_type, float loss_scale,
                             int block_len, int opt_f, int opt_pp, double* edge_blocks, int* edge_valid, hipStream_t s);

}  // namespace pc
