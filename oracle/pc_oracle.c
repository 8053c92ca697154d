/*
 * pc_oracle.c -- CPU restatement of the Polychase analysis hot path.  See pc_oracle.h for the
 * scope/parity statement ("parity unpinned" at the OpenCV boundary).  TEST INFRASTRUCTURE ONLY.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fPIC -shared pc_oracle.c -lm -lpthread
 * (-ffp-contract=off is REQUIRED: the canonical float order below assumes no FMA fusion.)
 */
#define _GNU_SOURCE
#include "pc_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

/* cv::borderInterpolate(p, len, BORDER_REFLECT_101) */
static inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

/* ------------------------------------------------------------------------------------------- */
/* cvtColor RGB2GRAY 8U: OpenCV 4.x RGB2Gray<uchar>, 15-bit coefficients RY15=9798, GY15=19235,
 * BY15=3735, CV_DESCALE(x, 15).  Reference call site: cpp/opticalflow.cc:259. */
void pco_rgb2gray(const uint8_t* rgb, int w, int h, uint8_t* gray) {
    const size_t n = (size_t)w * (size_t)h;
    for (size_t i = 0; i < n; i++) {
        const int r = rgb[3 * i], g = rgb[3 * i + 1], b = rgb[3 * i + 2];
        gray[i] = (uint8_t)((r * 9798 + g * 19235 + b * 3735 + (1 << 14)) >> 15);
    }
}

/* ------------------------------------------------------------------------------------------- */
/* ------------------------------------------------------------------------------------------- */
/* Where OpenCV's result depends on how the host executes it, the oracle has a canonical order (no FMA, LK sums exact in
 * integers: flags = 0) and an emulation of the x86 SIMD execution (pco_set_opencv_emulation).  THE DEFAULT IS THE X86
 * EXECUTION (both flags): what the OpenCV build the reference links most probably runs (vcpkg, x86-64, SSE baseline with
 * AVX2 / FMA3 dispatch) -- the GPU library has the same default (PC_ARITH_OPENCV_X86) and matches
 * either order bit for bit.  tests/test_oracle_emulation_cpu.py and DESIGN.md section 2 report the differences between the
 * two; tests/opencv_crosscheck.py checks both against a real cv2 where one exists:
 *   PCO_EMU_LK_SIMD    LKTrackerInvoker, CV_SIMD128 path (video/lkpyramid.cpp, baseline SSE2/SSE3 build): the
 *                      structure tensor and the mismatch vector are summed in fp32 -- four vector lanes over the
 *                      first (win/8)*8 columns plus a scalar accumulator over the rest, combined at the end --
 *                      instead of exactly in integers.  v_muladd is mul + add there (no FMA in the baseline).
 *   PCO_EMU_SOBEL_FMA  the symmetric 3-tap column filter of Sobel (imgproc/filter.simd.hpp, dispatched to AVX2
 *                      on any recent x86): v_muladd(S0 + S2, k1, v_muladd(S1, k0, 0)) with a fused multiply-add. */
static int g_emulation = PCO_EMU_OPENCV_X86;
void pco_set_opencv_emulation(int flags) { g_emulation = flags; }
int pco_get_opencv_emulation(void) { return g_emulation; }

/* cornerMinEigenVal (OpenCV imgproc/corner.cpp cornerEigenValsVecs + calcMinEigenVal).
 *   scale = 1 / (2^(ksize-1) * block_size * 255)            (8U input)
 *   Dx = Sobel(src, 32F, 1, 0, 3, scale): row kernel [-1,0,1] (exact), column kernel
 *        [1,2,1]*scale applied by the symmetric 3-tap column filter as (S0 + S2)*f1 + S1*f0.
 *   Dy = Sobel(src, 32F, 0, 1, 3, scale): row kernel [1,2,1]*scale applied by the generic row
 *        filter as ((f1*s[-1] + f0*s[0]) + f1*s[+1]); column kernel [-1,0,1] as S2 - S0.
 *   (Sobel() folds the scale into the *smoothing* kernel.)  All BORDER_REFLECT_101.
 *   cov = (Dx*Dx, Dx*Dy, Dy*Dy) in fp32; boxFilter(block x block, normalize=false) with fp64 sums
 *   (exact for 8-bit inputs in the canonical order: every product is a multiple of 2^-47 and |sum| < 2^6 -- NOT under
 *   PCO_EMU_SOBEL_FMA, where a cancelling Dx leaves a ~1e-10 residual: there the order below -- the window's rows summed
 *   left to right, the row sums added top to bottom, the structure of OpenCV's RowSum + ColumnSum -- defines the result),
 *   rounded to fp32; eig = (a + c) - sqrtf((a - c)*(a - c) + b*b), a = cxx*0.5f, b = cxy, c = cyy*0.5f.
 * Canonical choice: no FMA contraction (OpenCV's AVX2 dispatch may fuse v_muladd). */
static int corner_response(const uint8_t* gray, int w, int h, int block_size, int ksize, int harris, double harris_k, float* eig);
int pco_min_eigen_val(const uint8_t* gray, int w, int h, int block_size, int ksize, float* eig) {
    return corner_response(gray, w, h, block_size, ksize, 0, 0.0, eig);
}
/* cv::cornerHarris(gray, dst, block_size, ksize, k)  (cpp/feature_detection/gftt.cc:31-33): the same covariance sums
 * (cornerEigenValsVecs, same scale), then calcHarris' scalar expression
 *     dst = (float)(a * c - b * b - k * (a + c) * (a + c)),   a = cxx, b = cxy, c = cyy (float), k double
 * evaluated as C does: a*c, b*b and their difference in float; k * (a + c) * (a + c) in double.  An x86 build computes
 * only the last width % 4 columns of a row that way: the vector loop of calcHarris takes the first width / 4 * 4 columns in
 * float with (float)k, (a*c - b*b) - ((k*(a + c))*(a + c)), no fused operations -- emulated under PCO_EMU_SOBEL_FMA (the
 * "detector as an x86 build executes it" bit).  [recalled] */
int pco_corner_harris(const uint8_t* gray, int w, int h, int block_size, int ksize, double k, float* dst) {
    return corner_response(gray, w, h, block_size, ksize, 1, k, dst);
}
/* The derivative filters of cornerEigenValsVecs (imgproc/corner.cpp) for every aperture the reference's GFTTOptions can
 * carry (cpp/feature_detection/gftt.cc:31-36, bound at cpp/polychase_pybind.cc:128-136):  [recalled]
 *   ksize 3, 5, 7  Sobel: getSobelKernels' binomial taps -- smoothing [1,2,1] / [1,4,6,4,1] / [1,6,15,20,15,6,1], derivative
 *                  [-1,0,1] / [-1,-2,0,2,1] / [-1,-4,-5,0,5,4,1]; scale = 1 / (2^(ksize-1) * block_size * 255)
 *   ksize -1       Scharr (cornerEigenValsVecs calls Scharr() for aperture_size < 0): smoothing [3,10,3], derivative [-1,0,1];
 *                  scale = 1 / (2^2 * block_size * 255) / 2
 * Sobel() / Scharr() fold the scale into the SMOOTHING kernel of each image: `kx *= scale` is Mat::convertTo on a CV_32F
 * kernel, i.e. tap_f = (float)tap * (float)scale (cvtScale32f works in float).  sepFilter2D, 8U -> 32F through a 32F buffer:
 *   row pass     the generic RowFilter<uchar, float>: s = k[0]*S[0]; s += k[i]*S[i] in tap order (exact for the integer
 *                derivative taps; one rounding per tap for the scaled smoothing taps -- PCO_EMU_SOBEL_ROW_FMA: the vector row
 *                filter's fused chain from a zero accumulator)
 *   column pass  symmetric (smoothing) kernel: s = k[c]*S[c]; s = muladd(S[c+j] + S[c-j], k[c+j], s), j = 1 .. r -- the
 *                order of SymmColumnSmallVec_32f (ksize 3: (S0 + S2)*k1 + S1*k0, the same sum) and SymmColumnVec_32f /
 *                SymmColumnFilter's scalar tail (5, 7); PCO_EMU_SOBEL_FMA fuses each muladd (AVX2 dispatch).
 *                anti-symmetric (derivative) kernel: ksize 3 / Scharr S[c+1] - S[c-1] (the |k1| == 1 special case); 5 and 7:
 *                s = k[c+1]*(S[c+1] - S[c-1]); s = muladd(S[c+j] - S[c-j], k[c+j], s) -- the taps 2, 1 / 5, 4, 1 make every
 *                product but the first exact and the first is rounded either way: fused and unfused give the same bits.
 * All BORDER_REFLECT_101. */
static int corner_response(const uint8_t* gray, int w, int h, int block_size, int ksize, int harris, double harris_k, float* eig) {
    if ((ksize != 3 && ksize != 5 && ksize != 7 && ksize != -1) || block_size < 1 || w < 1 || h < 1) return -1;
    static const int sm3[3] = {1, 2, 1}, dv3[3] = {-1, 0, 1};
    static const int sm5[5] = {1, 4, 6, 4, 1}, dv5[5] = {-1, -2, 0, 2, 1};
    static const int sm7[7] = {1, 6, 15, 20, 15, 6, 1}, dv7[7] = {-1, -4, -5, 0, 5, 4, 1};
    static const int smS[3] = {3, 10, 3};
    const int taps = ksize > 0 ? ksize : 3, r = taps / 2;
    const int* sm = ksize == 3 ? sm3 : ksize == 5 ? sm5 : ksize == 7 ? sm7 : smS;
    const int* dv = ksize == 5 ? dv5 : ksize == 7 ? dv7 : dv3;
    double scale_d = (double)(1 << (taps - 1)) * block_size;
    if (ksize < 0) scale_d *= 2.0;
    scale_d = 1.0 / (scale_d * 255.0);
    const float scale_f = (float)scale_d;
    float smf[7];
    for (int k = 0; k < taps; k++) smf[k] = (float)sm[k] * scale_f;
    const size_t n = (size_t)w * (size_t)h;

    /* row pass */
    float* rdx = (float*)malloc(n * sizeof(float)); /* derivative taps along x: exact integers */
    float* rdy = (float*)malloc(n * sizeof(float)); /* smoothed along x, scaled */
    float* cov = (float*)malloc(n * 3 * sizeof(float));
    if (!rdx || !rdy || !cov) {
        free(rdx); free(rdy); free(cov);
        return -2;
    }
    for (int y = 0; y < h; y++) {
        const uint8_t* s = gray + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            float d = 0.0f, t = 0.0f;
            for (int k = 0; k < taps; k++) {
                const float v = (float)s[reflect101(x + k - r, w)];
                if (k == 0) {
                    d = (float)dv[0] * v;
                    t = smf[0] * v;
                } else {
                    d += (float)dv[k] * v;
                    if (g_emulation & PCO_EMU_SOBEL_ROW_FMA) t = fmaf(smf[k], v, t); /* the vector row filter's v_muladd chain, fused */
                    else t += smf[k] * v;
                }
            }
            rdx[(size_t)y * w + x] = d;
            rdy[(size_t)y * w + x] = t;
        }
    }
    /* column pass + covariance products */
    for (int y = 0; y < h; y++) {
        int yy[7];
        for (int k = 0; k < taps; k++) yy[k] = reflect101(y + k - r, h);
        for (int x = 0; x < w; x++) {
            float dx = rdx[(size_t)yy[r] * w + x] * smf[r];
            for (int j = 1; j <= r; j++) {
                const float pair = rdx[(size_t)yy[r + j] * w + x] + rdx[(size_t)yy[r - j] * w + x];
                if (g_emulation & PCO_EMU_SOBEL_FMA) dx = fmaf(pair, smf[r + j], dx);
                else dx = pair * smf[r + j] + dx;
            }
            float dy;
            if (taps == 3) {
                dy = rdy[(size_t)yy[2] * w + x] - rdy[(size_t)yy[0] * w + x];
            } else {
                dy = (float)dv[r + 1] * (rdy[(size_t)yy[r + 1] * w + x] - rdy[(size_t)yy[r - 1] * w + x]);
                for (int j = 2; j <= r; j++)
                    dy = (rdy[(size_t)yy[r + j] * w + x] - rdy[(size_t)yy[r - j] * w + x]) * (float)dv[r + j] + dy;
            }
            float* c = cov + ((size_t)y * w + x) * 3;
            c[0] = dx * dx;
            c[1] = dx * dy;
            c[2] = dy * dy;
        }
    }
    /* box filter (anchor = centre: block/2), fp64 sums, and min eigenvalue */
    const int a0 = block_size / 2;
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            /* boxFilter = RowSum (each window row summed left to right) followed by ColumnSum (the row sums added top to
             * bottom), both in fp64 (imgproc/box_filter.simd.hpp: RowSum<float, double>, ColumnSum<double, float>): that
             * order.  OpenCV's RUNNING updates of both sums (s += new - old along a row / down a column) are not restated:
             * they give the same bits wherever the sums are exact -- everywhere in the canonical execution, and all but
             * about one pixel in 10^5 under PCO_EMU_SOBEL_FMA (see the header comment). */
            double sxx = 0, sxy = 0, syy = 0;
            for (int j = 0; j < block_size; j++) {
                const int yy = reflect101(y + j - a0, h);
                double rxx = 0, rxy = 0, ryy = 0;
                for (int i = 0; i < block_size; i++) {
                    const int xx = reflect101(x + i - a0, w);
                    const float* c = cov + ((size_t)yy * w + xx) * 3;
                    rxx += (double)c[0];
                    rxy += (double)c[1];
                    ryy += (double)c[2];
                }
                sxx += rxx;
                sxy += rxy;
                syy += ryy;
            }
            if (harris) {
                const float a = (float)sxx, b = (float)sxy, c = (float)syy;
                const float ac = a * c, bb = b * b;
                const float det = ac - bb;
                const float tr = a + c;
                if ((g_emulation & PCO_EMU_SOBEL_FMA) && x < (w / 4) * 4) {
                    /* calcHarris' vector loop (128-bit universal intrinsics, and 8 at a time in corner.avx.cpp, compiled
                     * without FMA): all in float with (float)k: (a*c - b*b) - ((k*(a + c))*(a + c)) */
                    const float kf = (float)harris_k;
                    const float kt = kf * tr;
                    eig[(size_t)y * w + x] = det - kt * tr;
                } else {
                    eig[(size_t)y * w + x] = (float)((double)det - harris_k * (double)tr * (double)tr);
                }
                continue;
            }
            const float a = (float)sxx * 0.5f;
            const float b = (float)sxy;
            const float c = (float)syy * 0.5f;
            const float t = a - c;
            eig[(size_t)y * w + x] = (a + c) - sqrtf(t * t + b * b);
        }
    }
    free(rdx); free(rdy); free(cov);
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
void pco_gftt_default_options(pco_gftt_options* o) {
    o->quality_level = 0.01;
    o->min_distance = 5.0;
    o->block_size = 3;
    o->gradient_size = 3;
    o->max_corners = 0;
    o->use_harris = 0;
    o->harris_k = 0.04;
    o->grid_rows = 4;
    o->grid_cols = 4;
}

typedef struct {
    float val;
    int32_t idx; /* y*w + x == the "address" tie-break of gftt.cc:7-12 */
} cand_t;

/* gftt.cc:7-12  greaterThanPtr: value desc, then address desc */
static int cand_cmp(const void* pa, const void* pb) {
    const cand_t* a = (const cand_t*)pa;
    const cand_t* b = (const cand_t*)pb;
    if (a->val > b->val) return -1;
    if (a->val < b->val) return 1;
    if (a->idx > b->idx) return -1;
    if (a->idx < b->idx) return 1;
    return 0;
}

int pco_gftt(const uint8_t* gray, int w, int h, const pco_gftt_options* opt, float* xy_out,
             int capacity, float* eig_thresholded, int* n_candidates) {
    /* gftt.cc:18-19 */
    if (!(opt->quality_level > 0 && opt->min_distance >= 0 && opt->max_corners >= 0)) return -1;
    if (w <= 0 || h <= 0) return 0; /* gftt.cc:23-27 */
    const size_t n = (size_t)w * (size_t)h;
    float* eig = (float*)malloc(n * sizeof(float));
    if (!eig) return -1;
    if (corner_response(gray, w, h, opt->block_size, opt->gradient_size, opt->use_harris, opt->harris_k, eig) != 0) {   /* gftt.cc:31-36 */
        free(eig);
        return -1;
    }

    /* gftt.cc:38-67: per-cell max and THRESH_TOZERO, in place.  cv::threshold on CV_32F compares
     * against (float)thresh. */
    const int grid_rows = opt->grid_rows > 1 ? opt->grid_rows : 1;
    const int grid_cols = opt->grid_cols > 1 ? opt->grid_cols : 1;
    const int block_h = (h + grid_rows - 1) / grid_rows;
    const int block_w = (w + grid_cols - 1) / grid_cols;
    for (int gy = 0; gy < grid_rows; gy++) {
        for (int gx = 0; gx < grid_cols; gx++) {
            const int y0 = gy * block_h, x0 = gx * block_w;
            const int y1 = (y0 + block_h < h) ? y0 + block_h : h;
            const int x1 = (x0 + block_w < w) ? x0 + block_w : w;
            if (y0 >= y1 || x0 >= x1) continue; /* empty cv::Rect: nothing to do */
            float mx = eig[(size_t)y0 * w + x0];
            for (int y = y0; y < y1; y++)
                for (int x = x0; x < x1; x++)
                    if (eig[(size_t)y * w + x] > mx) mx = eig[(size_t)y * w + x];
            const double max_val = (double)mx;
            const float thr = (float)(max_val * opt->quality_level);
            for (int y = y0; y < y1; y++)
                for (int x = x0; x < x1; x++) {
                    float* p = &eig[(size_t)y * w + x];
                    *p = (*p > thr) ? *p : 0.0f;
                }
        }
    }
    if (eig_thresholded) memcpy(eig_thresholded, eig, n * sizeof(float));

    /* gftt.cc:69-86: 3x3 dilate (outside = -inf) and strict-interior local maxima */
    size_t cap = 1024, total = 0;
    cand_t* cands = (cand_t*)malloc(cap * sizeof(cand_t));
    for (int y = 1; y < h - 1; y++) {
        for (int x = 1; x < w - 1; x++) {
            const float val = eig[(size_t)y * w + x];
            if (val == 0) continue;
            float m = val;
            for (int j = -1; j <= 1; j++)
                for (int i = -1; i <= 1; i++) {
                    const float v = eig[(size_t)(y + j) * w + (x + i)];
                    if (v > m) m = v;
                }
            if (val == m) {
                if (total == cap) {
                    cap *= 2;
                    cands = (cand_t*)realloc(cands, cap * sizeof(cand_t));
                }
                cands[total].val = val;
                cands[total].idx = (int32_t)(y * w + x);
                total++;
            }
        }
    }
    if (n_candidates) *n_candidates = (int)total;
    if (total == 0) {
        free(cands); free(eig);
        return 0;
    }
    qsort(cands, total, sizeof(cand_t), cand_cmp); /* gftt.cc:98 (comparator is a total order) */

    int ncorners = 0;
    int overflow = 0;
    if (opt->min_distance >= 1) {
        /* gftt.cc:100-164 */
        const int cell_size = (int)lrint(opt->min_distance); /* cvRound */
        const int grid_w = (w + cell_size - 1) / cell_size;
        const int grid_h = (h + cell_size - 1) / cell_size;
        /* per-cell vectors as linked lists */
        int32_t* head = (int32_t*)malloc((size_t)grid_w * grid_h * sizeof(int32_t));
        int32_t* next = (int32_t*)malloc(total * sizeof(int32_t));
        float* acc_xy = (float*)malloc(total * 2 * sizeof(float));
        for (size_t i = 0; i < (size_t)grid_w * grid_h; i++) head[i] = -1;
        const double min_dist_sq = opt->min_distance * opt->min_distance;
        for (size_t i = 0; i < total; i++) {
            const int y = cands[i].idx / w;
            const int x = cands[i].idx - y * w;
            const int xc = x / cell_size, yc = y / cell_size;
            const int x1 = xc - 1 > 0 ? xc - 1 : 0, y1 = yc - 1 > 0 ? yc - 1 : 0;
            const int x2 = xc + 1 < grid_w - 1 ? xc + 1 : grid_w - 1;
            const int y2 = yc + 1 < grid_h - 1 ? yc + 1 : grid_h - 1;
            int good = 1;
            for (int yy = y1; yy <= y2 && good; yy++)
                for (int xx = x1; xx <= x2 && good; xx++)
                    for (int32_t j = head[yy * grid_w + xx]; j >= 0; j = next[j]) {
                        const float dx = (float)x - acc_xy[2 * j];
                        const float dy = (float)y - acc_xy[2 * j + 1];
                        if ((double)(dx * dx + dy * dy) < min_dist_sq) {
                            good = 0;
                            break;
                        }
                    }
            if (good) {
                acc_xy[2 * ncorners] = (float)x;
                acc_xy[2 * ncorners + 1] = (float)y;
                next[ncorners] = head[yc * grid_w + xc];
                head[yc * grid_w + xc] = ncorners;
                if (ncorners < capacity) {
                    xy_out[2 * ncorners] = (float)x;
                    xy_out[2 * ncorners + 1] = (float)y;
                } else {
                    overflow = 1;
                }
                ncorners++;
                if (opt->max_corners > 0 && ncorners == opt->max_corners) break;
            }
        }
        free(head); free(next); free(acc_xy);
    } else {
        /* gftt.cc:165-181 */
        for (size_t i = 0; i < total; i++) {
            const int y = cands[i].idx / w;
            const int x = cands[i].idx - y * w;
            if (ncorners < capacity) {
                xy_out[2 * ncorners] = (float)x;
                xy_out[2 * ncorners + 1] = (float)y;
            } else {
                overflow = 1;
            }
            ncorners++;
            if (opt->max_corners > 0 && ncorners == opt->max_corners) break;
        }
    }
    free(cands); free(eig);
    if (overflow) return -ncorners - 1;
    return ncorners;
}

/* ------------------------------------------------------------------------------------------- */
/* buildOpticalFlowPyramid (OpenCV video/lkpyramid.cpp).  */
#define PCO_MAX_LEVELS 16
struct pco_pyramid {
    int win, nlevels;
    int w[PCO_MAX_LEVELS], h[PCO_MAX_LEVELS];
    uint8_t* img[PCO_MAX_LEVELS];  /* padded (w+2win)x(h+2win), REFLECT_101 */
    int16_t* der[PCO_MAX_LEVELS];  /* padded, zeros outside, interleaved dx,dy */
};

/* copyMakeBorder(REFLECT_101 | ISOLATED) of the w x h interior that already sits inside `pad` */
static void pad_reflect101(uint8_t* pad, int w, int h, int win) {
    const int pw = w + 2 * win;
    for (int y = -win; y < h + win; y++) {
        const int sy = reflect101(y, h);
        uint8_t* drow = pad + (size_t)(y + win) * pw + win;
        const uint8_t* srow = pad + (size_t)(sy + win) * pw + win;
        for (int x = -win; x < w + win; x++) {
            if (y >= 0 && y < h && x >= 0 && x < w) continue;
            drow[x] = srow[reflect101(x, w)];
        }
    }
}

/* cv::pyrDown 8U (PyrDownInvoker<FixPtCast<uchar,8>>): separable [1 4 6 4 1], integer,
 * (sum + 128) >> 8, BORDER_REFLECT_101 on the un-padded source, dst = ((sw+1)/2, (sh+1)/2). */
static void pyr_down(const uint8_t* src, int sstride, int sw, int sh, uint8_t* dst, int dstride,
                     int dw, int dh) {
    int* rows = (int*)malloc((size_t)5 * dw * sizeof(int));
    for (int y = 0; y < dh; y++) {
        for (int k = 0; k < 5; k++) {
            const int sy = reflect101(2 * y - 2 + k, sh);
            const uint8_t* s = src + (size_t)sy * sstride;
            int* r = rows + (size_t)k * dw;
            for (int x = 0; x < dw; x++) {
                const int xm2 = reflect101(2 * x - 2, sw), xm1 = reflect101(2 * x - 1, sw);
                const int xc = reflect101(2 * x, sw);
                const int xp1 = reflect101(2 * x + 1, sw), xp2 = reflect101(2 * x + 2, sw);
                r[x] = s[xc] * 6 + (s[xm1] + s[xp1]) * 4 + s[xm2] + s[xp2];
            }
        }
        for (int x = 0; x < dw; x++) {
            const int v = rows[2 * dw + x] * 6 + (rows[dw + x] + rows[3 * dw + x]) * 4 + rows[x] +
                          rows[4 * dw + x];
            dst[(size_t)y * dstride + x] = (uint8_t)((v + 128) >> 8);
        }
    }
    free(rows);
}

/* calcScharrDeriv (lkpyramid.cpp ScharrDerivInvoker): rows/cols REFLECT_101 inside the level,
 * t0 = 3*(above+below) + 10*cur, t1 = below - above; dx = t0[x+1]-t0[x-1],
 * dy = 3*(t1[x+1]+t1[x-1]) + 10*t1[x]. */
static void scharr_deriv(const uint8_t* src, int sstride, int w, int h, int16_t* dst, int dstride) {
    int* t0 = (int*)malloc((size_t)(w + 2) * sizeof(int));
    int* t1 = (int*)malloc((size_t)(w + 2) * sizeof(int));
    for (int y = 0; y < h; y++) {
        const uint8_t* r0 = src + (size_t)(y > 0 ? y - 1 : (h > 1 ? 1 : 0)) * sstride;
        const uint8_t* r1 = src + (size_t)y * sstride;
        const uint8_t* r2 = src + (size_t)(y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0)) * sstride;
        for (int x = 0; x < w; x++) {
            t0[x + 1] = (r0[x] + r2[x]) * 3 + r1[x] * 10;
            t1[x + 1] = r2[x] - r0[x];
        }
        const int x0 = (w > 1 ? 1 : 0), x1 = (w > 1 ? w - 2 : 0);
        t0[0] = t0[x0 + 1]; t0[w + 1] = t0[x1 + 1];
        t1[0] = t1[x0 + 1]; t1[w + 1] = t1[x1 + 1];
        int16_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) {
            d[2 * x] = (int16_t)(t0[x + 2] - t0[x]);
            d[2 * x + 1] = (int16_t)((t1[x + 2] + t1[x]) * 3 + t1[x + 1] * 10);
        }
    }
    free(t0); free(t1);
}

pco_pyramid* pco_pyramid_build(const uint8_t* gray, int w, int h, int win, int max_level) {
    if (win <= 2 || max_level < 0 || w < 1 || h < 1) return NULL;
    if (max_level >= PCO_MAX_LEVELS) max_level = PCO_MAX_LEVELS - 1; /* no frame of <= 2^30 pixels has that many levels (loop below) */
    pco_pyramid* p = (pco_pyramid*)calloc(1, sizeof(pco_pyramid));
    p->win = win;
    int lw = w, lh = h;
    for (int level = 0; level <= max_level; level++) {
        const int pw = lw + 2 * win, ph = lh + 2 * win;
        p->w[level] = lw;
        p->h[level] = lh;
        p->img[level] = (uint8_t*)calloc((size_t)pw * ph, 1);
        p->der[level] = (int16_t*)calloc((size_t)pw * ph * 2, sizeof(int16_t));
        uint8_t* interior = p->img[level] + (size_t)win * pw + win;
        if (level == 0) {
            for (int y = 0; y < lh; y++) memcpy(interior + (size_t)y * pw, gray + (size_t)y * w, (size_t)lw);
        } else {
            const int spw = p->w[level - 1] + 2 * win;
            const uint8_t* sint = p->img[level - 1] + (size_t)win * spw + win;
            pyr_down(sint, spw, p->w[level - 1], p->h[level - 1], interior, pw, lw, lh);
        }
        pad_reflect101(p->img[level], lw, lh, win);
        scharr_deriv(interior, pw, lw, lh, p->der[level] + ((size_t)win * pw + win) * 2, pw * 2);
        p->nlevels = level + 1;
        /* lkpyramid.cpp: stop when the next level would be <= winSize */
        lw = (lw + 1) / 2;
        lh = (lh + 1) / 2;
        if (lw <= win || lh <= win) break;
    }
    return p;
}

void pco_pyramid_free(pco_pyramid* p) {
    if (!p) return;
    for (int i = 0; i < p->nlevels; i++) {
        free(p->img[i]);
        free(p->der[i]);
    }
    free(p);
}
int pco_pyramid_num_levels(const pco_pyramid* p) { return p->nlevels; }
int pco_pyramid_win(const pco_pyramid* p) { return p->win; }
void pco_pyramid_level_size(const pco_pyramid* p, int level, int* w, int* h) {
    *w = p->w[level];
    *h = p->h[level];
}
const uint8_t* pco_pyramid_image(const pco_pyramid* p, int level) { return p->img[level]; }
const int16_t* pco_pyramid_deriv(const pco_pyramid* p, int level) { return p->der[level]; }

/* ------------------------------------------------------------------------------------------- */
/* calcOpticalFlowPyrLK / LKTrackerInvoker (OpenCV video/lkpyramid.cpp), flags = 0.
 * Canonical accumulation: iA11/iA12/iA22/ib1/ib2 are exact integer sums (OpenCV x86 uses 4-lane
 * float partial sums whose rounding depends on the SIMD width); they are converted to float once
 * and scaled by FLT_SCALE = 2^-20. */
#define W_BITS 14
#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_floor_f(float v) { return (int)floorf(v); }

/* Optional diagnostic: iterations executed per (point, level), for scheduling studies of the GPU
 * kernel (tools/lk_divergence.py).  Not thread safe; NULL (default) disables it. */
static uint8_t* g_iter_trace = NULL;
static int g_iter_trace_levels = 0;
void pco_set_lk_iter_trace(uint8_t* buf, int levels) {
    g_iter_trace = buf;
    g_iter_trace_levels = levels;
}

static void lk_range(const pco_pyramid* P, const pco_pyramid* N, const float* pts, int i0, int i1,
                     int max_level, int max_iters, double eps_sq, float min_eig_thr,
                     float* next_pts, uint8_t* status, float* err) {
    const int win = P->win;
    const float half_win = (float)(win - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    int16_t* Iwin = (int16_t*)malloc((size_t)win * win * 3 * sizeof(int16_t));
    int16_t* dIwin = Iwin + (size_t)win * win;

    for (int pt = i0; pt < i1; pt++) {
        status[pt] = 1;
        err[pt] = 0.f;
        float nx = 0.f, ny = 0.f; /* nextPts[pt] as stored between levels */
        for (int level = max_level; level >= 0; level--) {
            const int lw = P->w[level], lh = P->h[level];
            const int pw = lw + 2 * win;
            const uint8_t* I = P->img[level] + (size_t)win * pw + win;
            const int16_t* dI = P->der[level] + ((size_t)win * pw + win) * 2;
            const uint8_t* J = N->img[level] + (size_t)win * pw + win;
            const int stepI = pw, dstep = pw * 2, stepJ = pw;

            const float lscale = (float)(1. / (1 << level));
            float px = pts[2 * pt] * lscale, py = pts[2 * pt + 1] * lscale;
            float qx, qy;
            if (level == max_level) {
                qx = px;
                qy = py;
            } else {
                qx = nx * 2.f;
                qy = ny * 2.f;
            }
            nx = qx;
            ny = qy;

            px -= half_win;
            py -= half_win;
            const int ipx = cv_floor_f(px), ipy = cv_floor_f(py);
            if (ipx < -win || ipx >= lw || ipy < -win || ipy >= lh) {
                if (level == 0) {
                    status[pt] = 0;
                    err[pt] = 0.f;
                }
                continue;
            }
            float a = px - (float)ipx, b = py - (float)ipy;
            int iw00 = cv_round_f((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
            int iw01 = cv_round_f(a * (1.f - b) * (float)(1 << W_BITS));
            int iw10 = cv_round_f((1.f - a) * b * (float)(1 << W_BITS));
            int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

            int64_t iA11 = 0, iA12 = 0, iA22 = 0;
            /* x86 SIMD order (PCO_EMU_LK_SIMD): lane j of the 4-lane fp32 accumulators takes columns x = j mod 4 of
             * the first simd_w columns (every product is exact in fp32: |ix|, |iy| <= 4080), the scalar fp32
             * accumulator the rest, row by row; "iA11 += buf[0] + buf[1] + buf[2] + buf[3]" at the end */
            const int emu = g_emulation & PCO_EMU_LK_SIMD;
            const int simd_w = (win / 8) * 8;
            float qA11[4] = {0, 0, 0, 0}, qA12[4] = {0, 0, 0, 0}, qA22[4] = {0, 0, 0, 0};
            float fA11 = 0.f, fA12 = 0.f, fA22 = 0.f;
            for (int y = 0; y < win; y++) {
                const uint8_t* src = I + (ptrdiff_t)(y + ipy) * stepI + ipx;
                const int16_t* dsrc = dI + (ptrdiff_t)(y + ipy) * dstep + ipx * 2;
                for (int x = 0; x < win; x++, dsrc += 2) {
                    const int ival = DESCALE(src[x] * iw00 + src[x + 1] * iw01 +
                                                 src[x + stepI] * iw10 + src[x + stepI + 1] * iw11,
                                             W_BITS - 5);
                    const int ixval = DESCALE(dsrc[0] * iw00 + dsrc[2] * iw01 + dsrc[dstep] * iw10 +
                                                  dsrc[dstep + 2] * iw11,
                                              W_BITS);
                    const int iyval = DESCALE(dsrc[1] * iw00 + dsrc[3] * iw01 +
                                                  dsrc[dstep + 1] * iw10 + dsrc[dstep + 3] * iw11,
                                              W_BITS);
                    Iwin[y * win + x] = (int16_t)ival;
                    dIwin[(y * win + x) * 2] = (int16_t)ixval;
                    dIwin[(y * win + x) * 2 + 1] = (int16_t)iyval;
                    iA11 += (int64_t)ixval * ixval;
                    iA12 += (int64_t)ixval * iyval;
                    iA22 += (int64_t)iyval * iyval;
                    if (emu) {
                        const float fx = (float)ixval, fy = (float)iyval;
                        if (x < simd_w) {
                            qA22[x & 3] = fy * fy + qA22[x & 3];
                            qA12[x & 3] = fx * fy + qA12[x & 3];
                            qA11[x & 3] = fx * fx + qA11[x & 3];
                        } else {
                            fA11 += (float)(ixval * ixval);
                            fA12 += (float)(ixval * iyval);
                            fA22 += (float)(iyval * iyval);
                        }
                    }
                }
            }
            float A11, A12, A22;
            if (emu) {
                fA11 += qA11[0] + qA11[1] + qA11[2] + qA11[3];
                fA12 += qA12[0] + qA12[1] + qA12[2] + qA12[3];
                fA22 += qA22[0] + qA22[1] + qA22[2] + qA22[3];
                A11 = fA11 * FLT_SCALE;
                A12 = fA12 * FLT_SCALE;
                A22 = fA22 * FLT_SCALE;
            } else {
                A11 = (float)iA11 * FLT_SCALE;
                A12 = (float)iA12 * FLT_SCALE;
                A22 = (float)iA22 * FLT_SCALE;
            }
            float D = A11 * A22 - A12 * A12;
            const float tdiff = A11 - A22;
            const float min_eig = (A22 + A11 - sqrtf(tdiff * tdiff + 4.f * A12 * A12)) /
                                  (float)(2 * win * win);
            if (min_eig < min_eig_thr || D < FLT_EPSILON) {
                if (level == 0) status[pt] = 0;
                continue;
            }
            D = 1.f / D;

            qx -= half_win;
            qy -= half_win;
            float pdx = 0.f, pdy = 0.f;
            for (int j = 0; j < max_iters; j++) {
                const int iqx = cv_floor_f(qx), iqy = cv_floor_f(qy);
                if (iqx < -win || iqx >= lw || iqy < -win || iqy >= lh) {
                    if (level == 0) status[pt] = 0;
                    break;
                }
                if (g_iter_trace && level < g_iter_trace_levels)
                    g_iter_trace[(size_t)pt * g_iter_trace_levels + level] = (uint8_t)(j + 1);
                a = qx - (float)iqx;
                b = qy - (float)iqy;
                iw00 = cv_round_f((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
                iw01 = cv_round_f(a * (1.f - b) * (float)(1 << W_BITS));
                iw10 = cv_round_f((1.f - a) * b * (float)(1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                int64_t ib1 = 0, ib2 = 0;
                /* x86 SIMD order: per group of 8 columns the products of columns (c, c + 4) are added as int32 pairs
                 * (v_dotprod), converted to fp32 and accumulated in lane (c & 1) * 2 of qb0 (c = 0, 1) or qb1 (c = 2, 3)
                 * for b1, the lane after it for b2; the scalar fp32 accumulator takes the remaining columns;
                 * "ib1 += bbuf[0] + bbuf[2]" with bbuf = qb0 + qb1 at the end */
                float qb0[4] = {0, 0, 0, 0}, qb1[4] = {0, 0, 0, 0}, fb1 = 0.f, fb2 = 0.f;
                for (int y = 0; y < win; y++) {
                    const uint8_t* Jp = J + (ptrdiff_t)(y + iqy) * stepJ + iqx;
                    int dcol[16];
                    for (int x = 0; x < win; x++) {
                        const int diff = DESCALE(Jp[x] * iw00 + Jp[x + 1] * iw01 +
                                                     Jp[x + stepJ] * iw10 + Jp[x + stepJ + 1] * iw11,
                                                 W_BITS - 5) -
                                         Iwin[y * win + x];
                        ib1 += (int64_t)diff * dIwin[(y * win + x) * 2];
                        ib2 += (int64_t)diff * dIwin[(y * win + x) * 2 + 1];
                        if (emu) {
                            if (x < simd_w) {
                                dcol[x & 7] = diff;
                                if ((x & 7) == 7) {
                                    const int16_t* d8 = dIwin + (y * win + (x - 7)) * 2;
                                    for (int c = 0; c < 4; c++) {
                                        const int p1 = dcol[c] * d8[2 * c] + dcol[c + 4] * d8[2 * (c + 4)];
                                        const int p2 = dcol[c] * d8[2 * c + 1] + dcol[c + 4] * d8[2 * (c + 4) + 1];
                                        float* q = (c < 2) ? qb0 : qb1;
                                        q[(c & 1) * 2] += (float)p1;
                                        q[(c & 1) * 2 + 1] += (float)p2;
                                    }
                                }
                            } else {
                                fb1 += (float)(diff * dIwin[(y * win + x) * 2]);
                                fb2 += (float)(diff * dIwin[(y * win + x) * 2 + 1]);
                            }
                        }
                    }
                }
                float b1, b2;
                if (emu) {
                    float bbuf[4];
                    for (int k = 0; k < 4; k++) bbuf[k] = qb0[k] + qb1[k];
                    fb1 += bbuf[0] + bbuf[2];
                    fb2 += bbuf[1] + bbuf[3];
                    b1 = fb1 * FLT_SCALE;
                    b2 = fb2 * FLT_SCALE;
                } else {
                    b1 = (float)ib1 * FLT_SCALE;
                    b2 = (float)ib2 * FLT_SCALE;
                }
                const float dx = (A12 * b2 - A22 * b1) * D;
                const float dy = (A12 * b1 - A11 * b2) * D;
                qx += dx;
                qy += dy;
                nx = qx + half_win;
                ny = qy + half_win;
                if ((double)dx * (double)dx + (double)dy * (double)dy <= eps_sq) break;
                if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                    nx -= dx * 0.5f;
                    ny -= dy * 0.5f;
                    break;
                }
                pdx = dx;
                pdy = dy;
            }

            if (status[pt] && level == 0) {
                const float ex = nx - half_win, ey = ny - half_win;
                const int iex = cv_floor_f(ex), iey = cv_floor_f(ey);
                if (iex < -win || iex >= lw || iey < -win || iey >= lh) {
                    status[pt] = 0;
                    continue;
                }
                const float aa = ex - (float)iex, bb = ey - (float)iey;
                iw00 = cv_round_f((1.f - aa) * (1.f - bb) * (float)(1 << W_BITS));
                iw01 = cv_round_f(aa * (1.f - bb) * (float)(1 << W_BITS));
                iw10 = cv_round_f((1.f - aa) * bb * (float)(1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                /* errval += |(float)diff|: every |diff| <= 8160 and the sum < 2^24, so the float
                 * accumulation is exact; an integer sum is identical. */
                int64_t esum = 0;
                for (int y = 0; y < win; y++) {
                    const uint8_t* Jp = J + (ptrdiff_t)(y + iey) * stepJ + iex;
                    for (int x = 0; x < win; x++) {
                        const int diff = DESCALE(Jp[x] * iw00 + Jp[x + 1] * iw01 +
                                                     Jp[x + stepJ] * iw10 + Jp[x + stepJ + 1] * iw11,
                                                 W_BITS - 5) -
                                         Iwin[y * win + x];
                        esum += diff < 0 ? -diff : diff;
                    }
                }
                /* errval * 1.f / (32*w*cn*h): multiplication by 1.f then float division */
                err[pt] = ((float)esum * 1.f) / (float)(32 * win * win);
            }
        }
        next_pts[2 * pt] = nx;
        next_pts[2 * pt + 1] = ny;
        if (!status[pt]) err[pt] = 0.f;
    }
    free(Iwin);
}

static int effective_max_level(const pco_pyramid* a, const pco_pyramid* b, int max_level) {
    int m = max_level;
    if (a->nlevels - 1 < m) m = a->nlevels - 1;
    if (b->nlevels - 1 < m) m = b->nlevels - 1;
    return m;
}

static double clamp_eps_sq(double eps) {
    /* criteria.epsilon = min(max(eps, 0.), 10.); epsilon *= epsilon */
    double e = eps < 0. ? 0. : (eps > 10. ? 10. : eps);
    return e * e;
}
static int clamp_iters(int it) { return it < 0 ? 0 : (it > 100 ? 100 : it); }

void pco_lk(const pco_pyramid* prev, const pco_pyramid* next, const float* pts, int n, int max_level,
            int max_iters, double eps, double min_eig_threshold, float* next_pts, uint8_t* status,
            float* err) {
    lk_range(prev, next, pts, 0, n, effective_max_level(prev, next, max_level),
             clamp_iters(max_iters), clamp_eps_sq(eps), (float)min_eig_threshold, next_pts, status,
             err);
}

void pco_flow_default_options(pco_flow_options* o) {
    o->window_size = 10;
    o->max_level = 3;
    o->term_max_iters = 30;
    o->term_epsilon = 0.01;
    o->min_eigen_threshold = 1e-4;
}

/* ------------------------------------------------------------------------------------------- */
/* Whole-clip driver shaped like GenerateOpticalFlowDatabase (cpp/opticalflow.cc:209-321). */
static const int32_t kSkips[8] = {-8, -4, -2, -1, 1, 2, 4, 8}; /* opticalflow.cc:76-77 */

typedef struct {
    const pco_pyramid *P, *N;
    const float* pts;
    int i0, i1, max_level, max_iters;
    double eps_sq;
    float thr;
    float* next_pts;
    uint8_t* status;
    float* err;
} lk_job;

static void* lk_job_main(void* arg) {
    lk_job* j = (lk_job*)arg;
    lk_range(j->P, j->N, j->pts, j->i0, j->i1, j->max_level, j->max_iters, j->eps_sq, j->thr,
             j->next_pts, j->status, j->err);
    return NULL;
}

typedef struct {
    const uint8_t* const* frames;
    int n_frames, w, h;
    int32_t first_frame, frame1;
    const pco_flow_options* fopt;
    const pco_pyramid* pyr1;
    const float* feats;
    int n_feats;
    int feature_threads;
    pco_record_cb cb;
    void* user;
    pthread_mutex_t* mtx;
    int* next_skip;
} pair_ctx;

static void run_pair(pair_ctx* c, int32_t frame2) {
    const size_t npx = (size_t)c->w * c->h;
    uint8_t* gray2 = (uint8_t*)malloc(npx);
    pco_rgb2gray(c->frames[frame2 - c->first_frame], c->w, c->h, gray2); /* :298 */
    pco_pyramid* pyr2 =
        pco_pyramid_build(gray2, c->w, c->h, c->fopt->window_size, c->fopt->max_level); /* :301 */
    const int n = c->n_feats;
    float* nxt = (float*)malloc((size_t)(n > 0 ? n : 1) * 2 * sizeof(float));
    uint8_t* st = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));
    float* er = (float*)malloc((size_t)(n > 0 ? n : 1) * sizeof(float));
    const int ml = effective_max_level(c->pyr1, pyr2, c->fopt->max_level);
    const int ft = c->feature_threads > 1 ? c->feature_threads : 1;
    if (ft == 1 || n < ft * 64) {
        lk_range(c->pyr1, pyr2, c->feats, 0, n, ml, clamp_iters(c->fopt->term_max_iters),
                 clamp_eps_sq(c->fopt->term_epsilon), (float)c->fopt->min_eigen_threshold, nxt, st, er);
    } else {
        pthread_t* th = (pthread_t*)malloc((size_t)ft * sizeof(pthread_t));
        lk_job* jobs = (lk_job*)malloc((size_t)ft * sizeof(lk_job));
        for (int t = 0; t < ft; t++) {
            jobs[t] = (lk_job){c->pyr1, pyr2, c->feats, (int)((int64_t)n * t / ft),
                               (int)((int64_t)n * (t + 1) / ft), ml,
                               clamp_iters(c->fopt->term_max_iters),
                               clamp_eps_sq(c->fopt->term_epsilon),
                               (float)c->fopt->min_eigen_threshold, nxt, st, er};
            pthread_create(&th[t], NULL, lk_job_main, &jobs[t]);
        }
        for (int t = 0; t < ft; t++) pthread_join(th[t], NULL);
        free(th); free(jobs);
    }
    /* opticalflow.cc:130-147: keep status == 1 */
    uint32_t* idx = (uint32_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(uint32_t));
    int m = 0;
    for (int i = 0; i < n; i++) {
        if (st[i] == 1) {
            idx[m] = (uint32_t)i;
            nxt[2 * m] = nxt[2 * i];
            nxt[2 * m + 1] = nxt[2 * i + 1];
            er[m] = er[i];
            m++;
        }
    }
    pthread_mutex_lock(c->mtx);
    c->cb(c->user, 1, c->frame1, frame2, m, idx, nxt, er); /* :149 */
    pthread_mutex_unlock(c->mtx);
    free(idx); free(nxt); free(st); free(er); free(gray2);
    pco_pyramid_free(pyr2);
}

static void* pair_worker(void* arg) {
    pair_ctx* c = (pair_ctx*)arg;
    for (;;) {
        pthread_mutex_lock(c->mtx);
        const int k = (*c->next_skip)++;
        pthread_mutex_unlock(c->mtx);
        if (k >= 8) break;
        const int32_t frame2 = c->frame1 + kSkips[k];
        if (frame2 < c->first_frame || frame2 >= c->first_frame + c->n_frames) continue; /* :282 */
        run_pair(c, frame2);
    }
    return NULL;
}

int pco_analyze_clip(const uint8_t* const* frames, int n_frames, int w, int h, int32_t first_frame,
                     int32_t f1_begin, int32_t f1_end, const pco_gftt_options* gopt,
                     const pco_flow_options* fopt, int threads, int feature_threads,
                     pco_record_cb cb, void* user) {
    if (threads < 1) threads = 1;
    if (threads > 8) threads = 8;
    const size_t npx = (size_t)w * h;
    uint8_t* gray1 = (uint8_t*)malloc(npx);
    int cap = (int)(npx / 4 + 16);
    float* feats = (float*)malloc((size_t)cap * 2 * sizeof(float));
    pthread_mutex_t mtx;
    pthread_mutex_init(&mtx, NULL);
    int rc = 0;
    for (int32_t f1 = f1_begin; f1 < f1_end; f1++) {
        if (f1 < first_frame || f1 >= first_frame + n_frames) continue;
        pco_rgb2gray(frames[f1 - first_frame], w, h, gray1); /* :259 */
        const int n = pco_gftt(gray1, w, h, gopt, feats, cap, NULL, NULL); /* :261 */
        if (n < 0) {
            rc = -1;
            break;
        }
        pthread_mutex_lock(&mtx);
        cb(user, 0, f1, f1, n, NULL, feats, NULL);
        pthread_mutex_unlock(&mtx);
        pco_pyramid* pyr1 = pco_pyramid_build(gray1, w, h, fopt->window_size, fopt->max_level); /* :263 */
        int next_skip = 0;
        pair_ctx c = {frames, n_frames, w, h, first_frame, f1, fopt, pyr1, feats, n,
                      feature_threads, cb, user, &mtx, &next_skip};
        if (threads == 1) {
            pair_worker(&c);
        } else {
            pthread_t th[8];
            for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, pair_worker, &c);
            for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
        }
        pco_pyramid_free(pyr1);
    }
    pthread_mutex_destroy(&mtx);
    free(gray1); free(feats);
    return rc;
}
