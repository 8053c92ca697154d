/*
 * pc_oracle.h -- CPU restatement ("oracle") of the Polychase video-analysis hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (polychase_amd/, include/) may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * PARITY STATUS: **parity unpinned at the OpenCV boundary.**  The reference
 * (/root/reference/cpp/opticalflow.cc, cpp/feature_detection/gftt.cc) delegates all pixel
 * arithmetic to OpenCV 4 (vcpkg tag 2025.06.13), which is neither vendored in the reference nor
 * installed in the build image, and the reference has no tests / golden vectors for this path.
 * This file restates the published OpenCV 4.x algorithms (imgproc: cvtColor, Sobel/sepFilter2D,
 * boxFilter, cornerMinEigenVal, threshold, dilate, pyrDown; video: buildOpticalFlowPyramid,
 * calcOpticalFlowPyrLK) in plain C and is cross-checked in tests/ against (a) independent
 * scipy.ndimage restatements of the integer stages, (b) an independent float64 numpy Lucas-Kanade
 * and (c) analytic ground truth on synthetic motion.  Where OpenCV's own result depends on the
 * SIMD dispatch of the machine it runs on (FMA contraction in v_muladd, 4-lane float partial sums
 * in the LK accumulators) this file fixes ONE canonical order, documented at each site:
 *   - no FMA contraction anywhere (compile with -ffp-contract=off),
 *   - LK structure-tensor / mismatch sums are accumulated exactly in integers and rounded to
 *     float once.
 * Everything the reference does in its own source (grid thresholding, candidate ordering, greedy
 * min-distance suppression, status filtering, pair enumeration) is followed exactly and cited.
 */
#ifndef PC_ORACLE_H_
#define PC_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- cv::cvtColor(COLOR_RGB2GRAY), 8U  (reference call: cpp/opticalflow.cc:259, :298) ---- */
void pco_rgb2gray(const uint8_t* rgb, int w, int h, uint8_t* gray);

/* ---- cv::cornerMinEigenVal(gray, eig, block_size=3, ksize=3)  (cpp/feature_detection/gftt.cc:35) ----
 * Only block_size >= 1 (anchor at centre) and ksize == 3 are restated. Returns 0 on success. */
int pco_min_eigen_val(const uint8_t* gray, int w, int h, int block_size, int ksize, float* eig);
/* cv::cornerHarris(gray, dst, block_size, ksize, k)  (gftt.cc:31-33), scalar expression of calcHarris. */
int pco_corner_harris(const uint8_t* gray, int w, int h, int block_size, int ksize, double k, float* dst);

/* GFTTOptions (cpp/feature_detection/gftt.h:5-21) */
typedef struct {
    double quality_level; /* 0.01 */
    double min_distance;  /* 5.0 */
    int block_size;       /* 3 */
    int gradient_size;    /* 3 */
    int max_corners;      /* 0 = unlimited */
    int use_harris;       /* 0 */
    double harris_k;      /* 0.04 */
    int grid_rows;        /* 4 */
    int grid_cols;        /* 4 */
} pco_gftt_options;

void pco_gftt_default_options(pco_gftt_options* o);

/* GoodFeaturesToTrack (cpp/feature_detection/gftt.cc:14-192), mask empty.
 * xy_out: capacity*2 floats, corners in acceptance order.  Returns the number of corners, or -1 on
 * unsupported options, or -(needed) - 1 if capacity is too small.
 * Optional outputs (may be NULL): eig_thresholded (w*h floats: the min-eig map after per-cell
 * THRESH_TOZERO), n_candidates (local maxima before suppression). */
int pco_gftt(const uint8_t* gray, int w, int h, const pco_gftt_options* opt, float* xy_out,
             int capacity, float* eig_thresholded, int* n_candidates);

/* ---- cv::buildOpticalFlowPyramid(img, pyr, Size(win,win), max_level, withDerivatives=true,
 *      pyrBorder=REFLECT_101, derivBorder=CONSTANT)  (cpp/opticalflow.cc:180-187) ---- */
typedef struct pco_pyramid pco_pyramid;
pco_pyramid* pco_pyramid_build(const uint8_t* gray, int w, int h, int win, int max_level);
void pco_pyramid_free(pco_pyramid* p);
int pco_pyramid_num_levels(const pco_pyramid* p); /* = returned maxLevel + 1 */
int pco_pyramid_win(const pco_pyramid* p);
void pco_pyramid_level_size(const pco_pyramid* p, int level, int* w, int* h);
/* padded planes: (w + 2*win) x (h + 2*win), row-major, tightly packed */
const uint8_t* pco_pyramid_image(const pco_pyramid* p, int level);
const int16_t* pco_pyramid_deriv(const pco_pyramid* p, int level); /* interleaved (dx,dy) */

/* ---- cv::calcOpticalFlowPyrLK(prevPyr, nextPyr, pts, ..., Size(win,win), max_level,
 *      TermCriteria(COUNT+EPS, max_iters, eps), flags=0, min_eig_threshold)
 *      (cpp/opticalflow.cc:119-125) ----
 * pts: n*2 floats.  next_pts: n*2, status: n bytes, err: n floats (0 where status==0). */
void pco_lk(const pco_pyramid* prev, const pco_pyramid* next, const float* pts, int n,
            int max_level, int max_iters, double eps, double min_eig_threshold, float* next_pts,
            uint8_t* status, float* err);

/* OpticalFlowOptions (cpp/opticalflow.h:27-33) */
typedef struct {
    int window_size;            /* 10 */
    int max_level;              /* 3 */
    int term_max_iters;         /* 30 */
    double term_epsilon;        /* 0.01 */
    double min_eigen_threshold; /* 1e-4 */
} pco_flow_options;
void pco_flow_default_options(pco_flow_options* o);

/* ---- Whole-clip CPU path, structured like GenerateOpticalFlowDatabase (cpp/opticalflow.cc:209-321):
 * for every frame1: gray, detect, pyramid; for each skip in {-8,-4,-2,-1,1,2,4,8}: gray+pyramid of
 * frame2 REBUILT per pair (:298-302), LK, status filter.  Used as the cpu_baseline "port" and as
 * the end-to-end oracle.  `threads` pair-threads (the reference caps TBB at 4, :270-271);
 * feature_threads > 1 additionally splits the features of each LK call (OpenCV parallel_for_).
 *
 * Output through a callback per record so that callers can hash / store without a DB:
 *   kind 0: keypoints of frame_from (n rows; xy = n*2 floats)
 *   kind 1: flow frame_from -> frame_to (n rows; idx, xy, err)
 * The callback is invoked under a mutex, in nondeterministic pair order within a frame. */
typedef void (*pco_record_cb)(void* user, int kind, int32_t frame_from, int32_t frame_to, int n,
                              const uint32_t* idx, const float* xy, const float* err);

/* frames: array of n_frames pointers to H*W*3 RGB u8.  first_frame: id of frames[0].
 * frame1 range processed: [f1_begin, f1_end) (ids); targets are clipped to the clip range.
 * Returns 0 on success. */
int pco_analyze_clip(const uint8_t* const* frames, int n_frames, int w, int h, int32_t first_frame,
                     int32_t f1_begin, int32_t f1_end, const pco_gftt_options* gopt,
                     const pco_flow_options* fopt, int threads, int feature_threads,
                     pco_record_cb cb, void* user);

/* Emulation of the x86 SIMD execution order of OpenCV (see pc_oracle.c); 0 = the canonical order; the default is
 * PCO_EMU_OPENCV_X86, like the GPU library's PC_ARITH_OPENCV_X86.  Process-wide, set before calling; not thread safe
 * against running calls. */
#define PCO_EMU_LK_SIMD 1
#define PCO_EMU_SOBEL_FMA 2   /* the detector as an x86 build executes it: FMA in the Sobel column filter, calcHarris' float vector loop */
#define PCO_EMU_OPENCV_X86 3
#define PCO_EMU_SOBEL_ROW_FMA 4   /* a second hypothesis: also the 8u -> 32f row smoothing of Dy as a fused chain (not in the default) */
void pco_set_opencv_emulation(int flags);
int pco_get_opencv_emulation(void);

/* Diagnostic only: when buf != NULL, pco_lk records the iterations it ran for (point, level) at
 * buf[point * levels + level] (0 = level skipped).  Single-threaded callers only. */
void pco_set_lk_iter_trace(uint8_t* buf, int levels);

#ifdef __cplusplus
}
#endif
#endif /* PC_ORACLE_H_ */
