// kernels_lk.hip -- pyramidal Lucas-Kanade on gfx950 (K8-K10) + status compaction.
//
// Replaces cv::calcOpticalFlowPyrLK as called at reference cpp/opticalflow.cc:119-125 and the
// status==1 filter of cpp/opticalflow.cc:130-147.  Arithmetic follows oracle/pc_oracle.c (which
// restates OpenCV's LKTrackerInvoker): 14-bit fixed-point bilinear weights, int16 patches,
// structure tensor / mismatch vector accumulated EXACTLY in integers, one rounding to fp32, 2x2
// solve in fp32 without FMA contraction.
//
// Mapping (v3): one wavefront per keypoint, one group of 8 lanes per target (<= 8 targets).  The
// I side (window patch, Scharr patch, structure tensor) does not depend on the target: the whole
// wave stages and evaluates it ONCE per level and hands every group its pixels through LDS.  A
// group owns the WIN x WIN window of its target (pixel p = lane + 8k).
//   * Every gather goes through LDS: per pyramid level the group stages (a) the I window as
//     "byte pairs" P[c] = (I[c], I[c+1]) and the raw Scharr window, (b) a (WIN+7) x (WIN+7..) search
//     region of the target image J in the same pair format.  A lane owns a window COLUMN, so one
//     pixel of one LK iteration is ONE aligned ds_read_u16 (the row below; the row above is the
//     previous pixel's) -> the 4 bilinear taps in one dword, and two
//     v_dot4_u32_u8 (the 14-bit weights are split w = 128*wh + wl so they fit u8 lanes).  The region
//     is re-staged only when the window leaves it.
//   * Window sums are all-reduced inside the group with DPP adds (quad_perm xor1/xor2 +
//     row_half_mirror): no LDS traffic, no cross-group traffic, so every lane holds the same A, b,
//     delta and the convergence branches are group-uniform.
#include <algorithm>
#include <cstdlib>

#include "lk_common.hpp"

namespace pc {

// One wavefront per keypoint; group g = lane / 8 tracks it into target g (v3).
//
// X86 = true: the sums of the structure tensor and of the mismatch vector in the ORDER an x86 OpenCV build executes them
// (LKTrackerInvoker's CV_SIMD128 path: four fp32 lane accumulators over the first (WIN / 8) * 8 columns, a scalar fp32
// accumulator over the rest, combined at the end) instead of exactly in integers -- PC_ARITH_LK_X86_ORDER, bit for bit
// oracle/pc_oracle.c under PCO_EMU_LK_SIMD.  Where every partial sum stays below 2^24 the two orders agree; on step
// edges they differ by up to ~2e-3 px.  The fp32 accumulations are sequential by definition,
// so this mode runs on the generic kernel only and costs about 3x its iteration.
template <int WIN, bool X86>
__global__ __launch_bounds__(256) void lk_kernel(const LKParams p) {
    using G = LKGeo<WIN>;
    constexpr int GL = 8;
    constexpr int NPX = WIN * WIN;
    constexpr int K = (NPX + GL - 1) / GL;
    constexpr int KW = (NPX + 63) / 64;  // pixels per lane in the cooperative (wave-wide) I-side pass
    constexpr int SIMD_W = (WIN / 8) * 8;                      // columns the x86 path handles with vector lanes
    constexpr int DIF_DW = X86 ? 8 * NPX : 0;                  // X86: every group's per-pixel differences of one iteration
    __shared__ __attribute__((aligned(16))) uint32_t s_buf[4][G::WAVE_DW + DIF_DW];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = lane >> 3, lg = lane & 7;
    // Workgroup b runs on XCD b % 8 (observed dispatch order; speed only).  Keypoints are visited in
    // spatially binned order (p.perm) and each XCD gets one contiguous eighth of that order, so the
    // windows a private L2 sees belong to one image region.
    lk_signal_dispatched(p);
    const int lb = (int)(blockIdx.x & 7u) * p.blocks_per_xcd + (int)(blockIdx.x >> 3);
    const int slot = lb * 4 + wave;
    if ((int)(blockIdx.x >> 3) >= p.blocks_per_xcd || slot >= p.n) return;  // whole waves exit together
    const int feat = p.perm ? (int)p.perm[slot] : slot;
    const bool tgt_active = grp < p.n_targets;
    const int tgt = tgt_active ? grp : 0;

    uint32_t* const wbase = &s_buf[wave][0];
    uint8_t* const ibuf = reinterpret_cast<uint8_t*>(wbase);                       // I window, pair format
    uint8_t* const dbuf = reinterpret_cast<uint8_t*>(wbase + G::I_DW);             // raw Scharr window
    uint32_t* const xbuf = wbase + G::I_DW + G::D_DW;                              // (Ival, Dxy) exchange
    uint8_t* const jbuf = reinterpret_cast<uint8_t*>(wbase + G::I_DW + G::D_DW + G::X_DW + grp * G::J_DW);
    int32_t* const dif = reinterpret_cast<int32_t*>(wbase + G::WAVE_DW) + grp * NPX;   // X86 only

    // Window pixels owned by this lane.  Main part: lane lg < WIN owns COLUMN lg (rows 0..WIN-1), so
    // the bottom taps of row y are the top taps of row y+1 and one LDS read per pixel suffices.
    // Extra part (WIN > 8): the remaining (WIN-8) columns are dealt out pixel by pixel.
    constexpr int KM = WIN;                                            // main slots
    constexpr int NEXTRA = (WIN > GL) ? (WIN - GL) * WIN : 0;          // pixels outside the first 8 columns
    constexpr int KE = (NEXTRA + GL - 1) / GL;                         // extra slots per lane
    static_assert(KM + KE == K || WIN < GL, "slot count");
    const bool main_valid = lg < WIN;
    int offE[KE > 0 ? KE : 1], qE[KE > 0 ? KE : 1];
#pragma unroll
    for (int e = 0; e < KE; e++) {
        const int r = lg + GL * e;
        const int col = GL + r / WIN, row = r - (r / WIN) * WIN;
        const bool ok = r < NEXTRA;
        offE[e] = ok ? row * G::PAIR_PITCH + 2 * col : 0;   // slots past the window read pixel 0, contribute 0
        qE[e] = ok ? row * WIN + col : -1;
    }

    const float2 pt = p.pts[feat];
    const float half_win = (float)(WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    float nx = 0.f, ny = 0.f;
    bool status = true;
    float err = 0.f;

    for (int level = p.max_level; level >= 0; --level) {
        const Level L = p.src[level];
        const uint8_t* __restrict__ J = p.tgt[tgt][level];
        const int pitch = L.pitch;
        const float lscale = 1.f / (float)(1 << level);
        float px = pt.x * lscale, py = pt.y * lscale;
        float qx, qy;
        if (level == p.max_level) {
            qx = px;
            qy = py;
        } else {
            qx = nx * 2.f;
            qy = ny * 2.f;
        }
        nx = qx;
        ny = qy;

        // ---- I side: identical for all targets -> computed once by the whole wave ----
        px -= half_win;
        py -= half_win;
        const int ipx = (int)floorf(px), ipy = (int)floorf(py);
        if (ipx < -WIN || ipx >= L.w || ipy < -WIN || ipy >= L.h) {   // wave-uniform
            if (level == 0) {
                status = false;
                err = 0.f;
            }
            continue;
        }
        const Weights wI = bilinear_weights(px - (float)ipx, py - (float)ipy);
        // signed 16-bit weight pairs for the derivative taps: (w00, w01) and (w10, w11)
        const uint32_t wrow0 = wI.r0, wrow1 = wI.r1;

        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        {
            DerivWindow<WIN, 64> dw;
            dw.load(L.der + (ptrdiff_t)(ipy * pitch + ipx), pitch, lane);
            stage_pairs_auto<WIN, 64, WIN + 1>(L.img, pitch, L.h, ipx & ~3, ipy, ibuf, lane);
            dw.store(reinterpret_cast<int32_t*>(dbuf), lane);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        int sA11 = 0, sA12 = 0, sA22 = 0;
        {
            const uint8_t* ib = ibuf + 2 * (ipx & 3);
#pragma unroll
            for (int m = 0; m < KW; m++) {
                const int q = lane + 64 * m;
                if (q < NPX) {
                    const int y = q / WIN, x = q - y * WIN;
                    const uint16_t* qp = reinterpret_cast<const uint16_t*>(ib + y * G::PAIR_PITCH + 2 * x);
                    const int ival = interp_pairs(widen_pair(qp[0]), widen_pair(qp[G::RWB]), wI);
                    const uint32_t* d = reinterpret_cast<const uint32_t*>(dbuf) + y * G::D_PITCH + x;
                    const uint32_t d00 = d[0], d01 = d[1], d10 = d[G::D_PITCH], d11 = d[G::D_PITCH + 1];
                    // (dx00, dx01), (dx10, dx11), (dy00, dy01), (dy10, dy11)
                    const uint32_t dx0 = __builtin_amdgcn_perm(d01, d00, 0x05040100u);
                    const uint32_t dx1 = __builtin_amdgcn_perm(d11, d10, 0x05040100u);
                    const uint32_t dy0 = __builtin_amdgcn_perm(d01, d00, 0x07060302u);
                    const uint32_t dy1 = __builtin_amdgcn_perm(d11, d10, 0x07060302u);
                    const int ix = sdot2(dx1, wrow1, sdot2(dx0, wrow0, 1 << (W_BITS - 1))) >> W_BITS;
                    const int iy = sdot2(dy1, wrow1, sdot2(dy0, wrow0, 1 << (W_BITS - 1))) >> W_BITS;
                    xbuf[2 * q] = (uint32_t)ival;
                    xbuf[2 * q + 1] = (uint32_t)(ix & 0xffff) | ((uint32_t)iy << 16);
                    sA11 += __mul24(ix, ix);   // |ix|, |iy| <= 4080
                    sA12 += __mul24(ix, iy);
                    sA22 += __mul24(iy, iy);
                }
            }
        }
        float A11, A12, A22;
        if constexpr (X86) {
            // lane j < 4: the vector lane that takes columns j, j + 4, ... < SIMD_W of every row, in row order;
            // lane 4: the scalar accumulator over the remaining columns; then fA += q0 + q1 + q2 + q3
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            float q11 = 0.f, q12 = 0.f, q22 = 0.f;
            for (int y = 0; y < WIN; y++) {
                if (lane < 4) {
                    for (int x = lane; x < SIMD_W; x += 4) {
                        const uint32_t d = xbuf[2 * (y * WIN + x) + 1];
                        const float fx = (float)(int)(int16_t)(d & 0xffffu), fy = (float)((int)d >> 16);
                        q22 = fy * fy + q22;
                        q12 = fx * fy + q12;
                        q11 = fx * fx + q11;
                    }
                } else if (lane == 4) {
                    for (int x = SIMD_W; x < WIN; x++) {
                        const uint32_t d = xbuf[2 * (y * WIN + x) + 1];
                        const int ix = (int)(int16_t)(d & 0xffffu), iy = (int)d >> 16;
                        q11 += (float)(ix * ix);
                        q12 += (float)(ix * iy);
                        q22 += (float)(iy * iy);
                    }
                }
            }
            auto combine = [](float q) {
                const float s = ((__shfl(q, 0) + __shfl(q, 1)) + __shfl(q, 2)) + __shfl(q, 3);
                return __shfl(q, 4) + s;
            };
            A11 = combine(q11) * FLT_SCALE;
            A12 = combine(q12) * FLT_SCALE;
            A22 = combine(q22) * FLT_SCALE;
        } else {
            // |ix|,|iy| <= 4080: per-lane partials fit int32; totals reduced as exact (hi, lo) halves
            A11 = wave_exact_sum(sA11) * FLT_SCALE;
            A12 = wave_exact_sum(sA12) * FLT_SCALE;
            A22 = wave_exact_sum(sA22) * FLT_SCALE;
        }
        float D = A11 * A22 - A12 * A12;
        const float tdiff = A11 - A22;
        const float min_eig = (A22 + A11 - sqrtf(tdiff * tdiff + 4.f * A12 * A12)) / (float)(2 * WIN * WIN);
        if (min_eig < p.min_eig_thr || D < 1.1920928955078125e-07f /* FLT_EPSILON */) {   // wave-uniform
            if (level == 0) status = false;
            continue;
        }
        D = 1.f / D;
        if (!tgt_active) continue;  // idle groups only help with the I side

        // every group picks up the pixels it owns
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        int Ival[KM + KE];  // ival_bias(I patch value): the accumulator init of interp_diff
        int Dxy[KM + KE];  // (int16 ix) | (int16 iy << 16); 0 for slots without a pixel
#pragma unroll
        for (int k = 0; k < KM; k++) {
            const uint2 v = main_valid ? *reinterpret_cast<const uint2*>(xbuf + 2 * (k * WIN + lg)) : make_uint2(0u, 0u);
            Ival[k] = ival_bias((int)v.x);
            Dxy[k] = (int)v.y;
        }
#pragma unroll
        for (int e = 0; e < KE; e++) {
            const uint2 v = (qE[e] >= 0) ? *reinterpret_cast<const uint2*>(xbuf + 2 * qE[e]) : make_uint2(0u, 0u);
            Ival[KM + e] = ival_bias((int)v.x);
            Dxy[KM + e] = (int)v.y;
        }

        // ---- iterations on the staged J region ----
        qx -= half_win;
        qy -= half_win;
        float pdx = 0.f, pdy = 0.f;
        int rx0 = 0, ry0 = 0;
        bool staged = false;
        for (int j = 0; j < p.max_iters; j++) {
            const int iqx = (int)floorf(qx), iqy = (int)floorf(qy);
            if (iqx < -WIN || iqx >= L.w || iqy < -WIN || iqy >= L.h) {
                if (level == 0) status = false;
                break;
            }
            if (!staged || iqx < rx0 || iqx + WIN > rx0 + G::RWB || iqy < ry0 || iqy + WIN + 1 > ry0 + G::RH) {
                rx0 = (iqx - G::MX) & ~3;
                ry0 = iqy - G::MY;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                stage_pairs_auto<WIN, GL, G::RH>(J, pitch, L.h, rx0, ry0, jbuf, lg);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                staged = true;
            }
            const Weights wJ = bilinear_weights(qx - (float)iqx, qy - (float)iqy);
            const uint8_t* jb = jbuf + (iqy - ry0) * G::PAIR_PITCH + 2 * (iqx - rx0);
            int sb1 = 0, sb2 = 0;  // per-lane partials: <= K * 8160 * 4080
            // b1 += diff * ix, b2 += diff * iy: |diff| <= 8160 fits int16, (ix, iy) are the halves of Dxy
            // (slots without a pixel have Dxy == 0)
            {
                const uint8_t* cb = jb + 2 * lg;
                uint32_t top = widen_pair(*reinterpret_cast<const uint16_t*>(cb));
#pragma unroll
                for (int k = 0; k < KM; k++) {
                    const uint32_t bot = widen_pair(*reinterpret_cast<const uint16_t*>(cb + (k + 1) * G::PAIR_PITCH));
                    const int diff = interp_diff(top, bot, wJ, Ival[k]);
                    top = bot;
                    if constexpr (X86) {
                        if (main_valid) dif[k * WIN + lg] = diff;
                    } else {
                        sb1 = mad16_lo(diff, (uint32_t)Dxy[k], sb1);
                        sb2 = mad16_hi(diff, (uint32_t)Dxy[k], sb2);
                    }
                }
#pragma unroll
                for (int e = 0; e < KE; e++) {
                    const uint16_t* q = reinterpret_cast<const uint16_t*>(jb + offE[e]);
                    const int diff = interp_diff(widen_pair(q[0]), widen_pair(q[G::RWB]), wJ, Ival[KM + e]);
                    if constexpr (X86) {
                        if (qE[e] >= 0) dif[qE[e]] = diff;
                    } else {
                        sb1 = mad16_lo(diff, (uint32_t)Dxy[KM + e], sb1);
                        sb2 = mad16_hi(diff, (uint32_t)Dxy[KM + e], sb2);
                    }
                }
            }
            float b1, b2;
            if constexpr (X86) {
                // per 8 columns the products of columns (c, c + 4) are added as int32 pairs (v_dotprod), converted to
                // fp32 and accumulated row by row in vector lane c (lanes 0-3 of the group); the scalar accumulator
                // (lane 4) takes the remaining columns; b = scalar + ((q[c=0] + q[c=2]) + (q[c=1] + q[c=3]))
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                float q1 = 0.f, q2 = 0.f;
                for (int y = 0; y < WIN; y++) {
                    if (lg < 4) {
                        for (int x0 = 0; x0 < SIMD_W; x0 += 8) {
                            const int qa = y * WIN + x0 + lg, qb = qa + 4;
                            const int da = dif[qa], db = dif[qb];
                            const uint32_t ea = xbuf[2 * qa + 1], eb = xbuf[2 * qb + 1];
                            const int p1 = da * (int)(int16_t)(ea & 0xffffu) + db * (int)(int16_t)(eb & 0xffffu);
                            const int p2 = da * ((int)ea >> 16) + db * ((int)eb >> 16);
                            q1 += (float)p1;
                            q2 += (float)p2;
                        }
                    } else if (lg == 4) {
                        for (int x = SIMD_W; x < WIN; x++) {
                            const int qq = y * WIN + x;
                            const int d = dif[qq];
                            const uint32_t e2 = xbuf[2 * qq + 1];
                            q1 += (float)(d * (int)(int16_t)(e2 & 0xffffu));
                            q2 += (float)(d * ((int)e2 >> 16));
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the differences are consumed: the next iteration may overwrite them
                auto combine = [](float q) {
                    const float v = (__shfl(q, 0, 8) + __shfl(q, 2, 8)) + (__shfl(q, 1, 8) + __shfl(q, 3, 8));
                    return __shfl(q, 4, 8) + v;
                };
                b1 = combine(q1) * FLT_SCALE;
                b2 = combine(q2) * FLT_SCALE;
            } else if constexpr ((long long)K * 8160 * 4080 < (1ll << 29)) {
                b1 = group8_exact_sum_small(sb1) * FLT_SCALE;
                b2 = group8_exact_sum_small(sb2) * FLT_SCALE;
            } else {
                b1 = group_exact_sum<GL>(sb1) * FLT_SCALE;
                b2 = group_exact_sum<GL>(sb2) * FLT_SCALE;
            }
            const float dx = (A12 * b2 - A22 * b1) * D;
            const float dy = (A12 * b1 - A11 * b2) * D;
            qx += dx;
            qy += dy;
            nx = qx + half_win;
            ny = qy + half_win;
            if ((double)dx * (double)dx + (double)dy * (double)dy <= p.eps_sq) break;
            if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                nx -= dx * 0.5f;
                ny -= dy * 0.5f;
                break;
            }
            pdx = dx;
            pdy = dy;
        }

        // ---- L1 patch error at level 0 ----
        if (status && level == 0) {
            const float ex = nx - half_win, ey = ny - half_win;
            const int iex = (int)floorf(ex), iey = (int)floorf(ey);
            if (iex < -WIN || iex >= L.w || iey < -WIN || iey >= L.h) {
                status = false;
                continue;
            }
            if (!staged || iex < rx0 || iex + WIN > rx0 + G::RWB || iey < ry0 || iey + WIN + 1 > ry0 + G::RH) {
                rx0 = (iex - G::MX) & ~3;
                ry0 = iey - G::MY;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                stage_pairs_auto<WIN, GL, G::RH>(J, pitch, L.h, rx0, ry0, jbuf, lg);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                staged = true;
            }
            const Weights wE = bilinear_weights(ex - (float)iex, ey - (float)iey);
            const uint8_t* jb = jbuf + (iey - ry0) * G::PAIR_PITCH + 2 * (iex - rx0);
            int se = 0;
            {
                const uint8_t* cb = jb + 2 * lg;
                uint32_t top = widen_pair(*reinterpret_cast<const uint16_t*>(cb));
#pragma unroll
                for (int k = 0; k < KM; k++) {
                    const uint32_t bot = widen_pair(*reinterpret_cast<const uint16_t*>(cb + (k + 1) * G::PAIR_PITCH));
                    const int diff = interp_diff(top, bot, wE, Ival[k]);
                    top = bot;
                    se += main_valid ? (diff < 0 ? -diff : diff) : 0;
                }
#pragma unroll
                for (int e = 0; e < KE; e++) {
                    const uint16_t* q = reinterpret_cast<const uint16_t*>(jb + offE[e]);
                    const int diff = interp_diff(widen_pair(q[0]), widen_pair(q[G::RWB]), wE, Ival[KM + e]);
                    se += (qE[e] >= 0) ? (diff < 0 ? -diff : diff) : 0;
                }
            }
            se = group_allreduce_add<GL>(se);  // <= 256 * 8160 < 2^24: exact in fp32 too
            err = ((float)se * 1.f) / (float)(32 * WIN * WIN);
        }
    }

    // one 16-byte record per (slot, target): the wavefront's results are contiguous
    if (lg == 0 && tgt_active)
        p.out_rec[(size_t)slot * kRecStride + tgt] = make_float4(nx, ny, status ? err : 0.f, __uint_as_float(status ? 1u : 0u));
}

template <int WIN>
static void launch_lk_t(const LKParams& p0, hipStream_t s) {
    LKParams p = p0;
    const int blocks = (p.n + 3) / 4;   // one wavefront per keypoint, 4 per workgroup
    if (blocks == 0) return;
    p.blocks_per_xcd = (blocks + 7) / 8;
    if (p.x86_order) hipLaunchKernelGGL((lk_kernel<WIN, true>), dim3((unsigned)p.blocks_per_xcd * 8u), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((lk_kernel<WIN, false>), dim3((unsigned)p.blocks_per_xcd * 8u), dim3(256), 0, s, p);
}

// POLYCHASE_LK_VARIANT=1 forces the generic one-keypoint-per-wavefront kernel of this file (windows up to 16: the cross-check of
// the two product kernels; round 1's two-keypoint kernel on the u8 planes, kernels_lk2.hip, was removed in round 3 after its last
// measurement: profiles/r03_c2_lk_variants.jsonl);
// default: the two-keypoint kernel on the uint16 planes (kernels_lk3.hip) for windows 4..10, the eight-lanes-per-target kernel
// (lk4_kernel.hpp) for window 3 and windows 11..31
static int lk_variant() {
    static const int v = [] {
        const char* e = getenv("POLYCHASE_LK_VARIANT");
        return e ? atoi(e) : 0;
    }();
    return v;
}

bool launch_lk(const LKParams& p, int win, hipStream_t s) {
    const int v = lk_variant();
    // window 11: the eight-lanes-per-target kernel (0.43 ms per C2 launch against 0.62 on the two-keypoint kernel, whose 11-px
    // instance has no registers left for its unrolled ordered tensor and read-ahead; POLYCHASE_LK_VARIANT=3 keeps it reachable)
    if (v == 0 && win == 11 && launch_lk4a(p, win, s)) return true;
    if ((v == 0 || v == 3) && launch_lk3(p, win, s)) return true;
    if ((v != 1 || win > 16) && (launch_lk4a(p, win, s) || launch_lk4b(p, win, s) || launch_lk4c(p, win, s))) return true;
    switch (win) {
#define PC_LK_CASE(W) case W: launch_lk_t<W>(p, s); return true;
        PC_LK_CASE(3) PC_LK_CASE(4) PC_LK_CASE(5) PC_LK_CASE(6) PC_LK_CASE(7) PC_LK_CASE(8) PC_LK_CASE(9)
        PC_LK_CASE(10) PC_LK_CASE(11) PC_LK_CASE(12) PC_LK_CASE(13) PC_LK_CASE(14) PC_LK_CASE(15) PC_LK_CASE(16)
#undef PC_LK_CASE
        default: return false;
    }
}

// ------------------------------------------------------------------------------------------------
// Spatial binning of the keypoints (counting sort by 64x64 tile, raster order of tiles): the order
// in which LK visits keypoints.  Keypoints are stored by corner response, i.e. randomly in space;
// visiting them tile by tile keeps the gathers of concurrently running waves inside one image
// region (L2 hits instead of fabric requests).  Results are written by keypoint index, so the order
// inside a tile (atomics) does not affect the output.
// ------------------------------------------------------------------------------------------------
constexpr int BIN_SHIFT = 6;

__global__ __launch_bounds__(256) void bin_count_kernel(const float2* __restrict__ pts, int n_max, const uint32_t* __restrict__ n_dev, int tiles_x, int n_tiles,
                                                        uint32_t* __restrict__ hist, int hi_prio) {
    helper_priority(hi_prio);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = n_dev ? min((int)*n_dev, n_max) : n_max;
    if (i >= n) return;
    const float2 p = pts[i];
    const int t = min(n_tiles - 1, max(0, ((int)p.y >> BIN_SHIFT) * tiles_x + ((int)p.x >> BIN_SHIFT)));
    atomicAdd(&hist[t], 1u);
}

// (256 lanes: beside a running LK launch a workgroup needs its wavefronts resident together, and 1024-lane
// workgroups -- 4 wavefronts per SIMD -- do not fit the registers three LK wavefronts per SIMD leave over; they
// waited for the whole LK launch to drain, measured 2 ms for a 50-us kernel)
__global__ __launch_bounds__(256) void bin_scan_kernel(uint32_t* __restrict__ hist, int n_tiles, int hi_prio) {
    helper_priority(hi_prio);
    __shared__ uint32_t s_sum[256];
    const int per = (n_tiles + 255) / 256;
    const int b = threadIdx.x * per, e = min(b + per, n_tiles);
    uint32_t s = 0;
    for (int i = b; i < e; i++) s += hist[i];
    s_sum[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t v = (threadIdx.x >= (unsigned)d) ? s_sum[threadIdx.x - d] : 0u;
        __syncthreads();
        s_sum[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = s_sum[threadIdx.x] - s;  // exclusive prefix of this lane's range
    for (int i = b; i < e; i++) {
        const uint32_t c = hist[i];
        hist[i] = run;
        run += c;
    }
}

__global__ __launch_bounds__(256) void bin_scatter_kernel(const float2* __restrict__ pts, int n_max, const uint32_t* __restrict__ n_dev, int tiles_x, int n_tiles,
                                                          uint32_t* __restrict__ cursor, uint32_t* __restrict__ perm,
                                                          uint32_t* __restrict__ slot_of, const uint32_t* __restrict__ copy_src,
                                                          uint32_t* __restrict__ copy_dst, int copy_words, int hi_prio) {
    helper_priority(hi_prio);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && (int)threadIdx.x < copy_words) copy_dst[threadIdx.x] = copy_src[threadIdx.x];
    const int n = n_dev ? min((int)*n_dev, n_max) : n_max;
    if (i >= n) return;
    const float2 p = pts[i];
    const int t = min(n_tiles - 1, max(0, ((int)p.y >> BIN_SHIFT) * tiles_x + ((int)p.x >> BIN_SHIFT)));
    const uint32_t pos = atomicAdd(&cursor[t], 1u);
    perm[pos] = (uint32_t)i;
    slot_of[i] = pos;
}

// One wavefront that idles until the launch ahead (on another stream) has handed out all its workgroups.
// The launch ahead was enqueued first and normally runs already.  If the two streams share a hardware queue, though (the
// process holds more HIP streams than the runtime has queues), this kernel sits in FRONT of the launch it waits for: after
// ~50 ms it gives up and raises *timed_out (pinned host memory) -- the analyzer then stops using the gate.
__global__ __launch_bounds__(64) void lk_gate_kernel(const uint32_t* gate, uint32_t value, uint32_t* timed_out) {
    if (threadIdx.x != 0) return;
    for (uint32_t spin = 0; spin < 25000u; spin++) {
        const uint32_t v = __hip_atomic_load(const_cast<uint32_t*>(gate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int32_t)(v - value) >= 0) return;
        __builtin_amdgcn_s_sleep(64);
    }
    if (timed_out) __hip_atomic_store(timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_lk_gate(const uint32_t* gate, uint32_t value, uint32_t* timed_out, hipStream_t s) {
    hipLaunchKernelGGL(lk_gate_kernel, dim3(1), dim3(64), 0, s, gate, value, timed_out);
}

int bin_num_tiles(int w, int h) { return ((w + 63) >> BIN_SHIFT) * ((h + 63) >> BIN_SHIFT); }

void launch_spatial_bins(const float2* pts, int n, const uint32_t* n_dev, int w, int h, uint32_t* hist, uint32_t* perm,
                         uint32_t* slot_of, hipStream_t s) {
    if (n <= 0) return;
    const int tiles_x = (w + 63) >> BIN_SHIFT, n_tiles = bin_num_tiles(w, h);
    (void)hipMemsetAsync(hist, 0, (size_t)n_tiles * sizeof(uint32_t), s);
    hipLaunchKernelGGL(bin_count_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pts, n, n_dev, tiles_x, n_tiles, hist, helper_prio_arg());
    hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(256), 0, s, hist, n_tiles, helper_prio_arg());
    hipLaunchKernelGGL(bin_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pts, n, n_dev, tiles_x, n_tiles, hist, perm, slot_of,
                       (const uint32_t*)nullptr, (uint32_t*)nullptr, 0, helper_prio_arg());
}

void launch_spatial_bins_counted(const float2* pts, int n, const uint32_t* n_dev, int w, int h, uint32_t* hist, uint32_t* perm,
                                 uint32_t* slot_of, const uint32_t* copy_src, uint32_t* copy_dst, int copy_words, hipStream_t s) {
    const int tiles_x = (w + 63) >> BIN_SHIFT, n_tiles = bin_num_tiles(w, h);
    hipLaunchKernelGGL(bin_scatter_kernel, dim3((std::max(n, 1) + 255) / 256), dim3(256), 0, s, pts, n, n_dev, tiles_x, n_tiles, hist, perm,
                       slot_of, copy_src, copy_dst, copy_words, helper_prio_arg());
}

// ------------------------------------------------------------------------------------------------
// Ordered compaction of status == 1 rows (opticalflow.cc:130-147) in TWO launches: count per block of 256 keypoints --
// whose last workgroup turns the counts into offsets (pc::last_workgroup: the scan is the tail of the kernel that
// produces its input, not a launch of its own) -- and scatter.  The LK kernel leaves its records in visiting order
// (8 per slot, 128 contiguous bytes); here lane = (keypoint i, target t) with t the fast index, so the 8 lanes of a
// keypoint read exactly that line through the inverse permutation and write one run per target in ascending keypoint
// order.  A 256-lane workgroup walks its 256 keypoints in 8 passes of 32.
// (Round 2: 32 keypoints per workgroup and a scan kernel of ONE workgroup that walked nblocks x 8 counters in two serial
// passes -- 10 k counters at 1080p, 40 k at 4K: 58 / 294 us on the job lane between the LK launch and the download.)
// scratch: [kCompactTicketWords tickets, zero before the launch and zeroed again by the scatter][n_targets x nblocks counts]
// ------------------------------------------------------------------------------------------------
constexpr int CF = 256;                  // keypoints per workgroup
constexpr int CT = 256;                  // lanes (small workgroups: see bin_scan_kernel)
constexpr int CPASS = CF * kRecStride / CT;   // passes of CT / kRecStride = 32 keypoints
int compact_num_blocks(int n) { return (n + CF - 1) / CF; }
size_t compact_scratch_words(int n, int n_targets) {
    return (size_t)kCompactTicketWords + (size_t)compact_num_blocks(n) * (size_t)std::max(n_targets, 1) + 1;
}

__device__ __forceinline__ unsigned long long target_lanes(int t) { return 0x0101010101010101ull << t; }

// block_counts[t][b] -> exclusive offsets (global, target-major) + row_offset[]; all 256 lanes of one workgroup
__device__ __forceinline__ void compact_scan_body(uint32_t* __restrict__ block_counts, int nblocks, int n_targets,
                                                  long long* __restrict__ row_offset) {
    __shared__ long long s_part[256];
    const int total = nblocks * n_targets;
    const int per = (total + 255) / 256;
    const int b = threadIdx.x * per, e = min(b + per, total);
    long long sum = 0;
    for (int i = b; i < e; i++) sum += block_counts[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {   // inclusive scan of the 256 partial sums
        const long long v = (threadIdx.x >= (unsigned)d) ? s_part[threadIdx.x - d] : 0ll;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    long long run = s_part[threadIdx.x] - sum;   // exclusive
    for (int i = b; i < e; i++) {
        const uint32_t c = block_counts[i];
        if (i % nblocks == 0) row_offset[i / nblocks] = run;
        block_counts[i] = (uint32_t)run;  // total rows < 2^32 (n_targets * n)
        run += c;
    }
    if (e == total && b < e) row_offset[n_targets] = run;
}

__global__ __launch_bounds__(CT) void compact_count_kernel(const float4* __restrict__ rec, const uint32_t* __restrict__ slot_of,
                                                             int n, int n_targets, int nblocks,
                                                             uint32_t* __restrict__ tickets, uint32_t* __restrict__ block_counts,
                                                             long long* __restrict__ row_offset, int hi_prio) {
    helper_priority(hi_prio);
    __shared__ uint32_t s_cnt[kRecStride];
    const int t = threadIdx.x & 7, lane = threadIdx.x & 63;
    if (threadIdx.x < kRecStride) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t mine = 0;   // lanes 0-7 of a wavefront: rows of target `lane` among the wavefront's keypoints
#pragma unroll
    for (int k = 0; k < CPASS; k++) {
        const int i = blockIdx.x * CF + k * (CT / kRecStride) + (int)(threadIdx.x >> 3);
        bool keep = false;
        if (i < n && t < n_targets) keep = __float_as_uint(rec[(size_t)slot_of[i] * kRecStride + t].w) == 1u;
        const unsigned long long b = __ballot(keep);
        if (lane < kRecStride) mine += (uint32_t)__popcll(b & target_lanes(lane));
    }
    if (lane < kRecStride && mine) atomicAdd(&s_cnt[lane], mine);
    __syncthreads();
    // agent-scope store: the workgroup that finishes last reads the counts of all the others (last_workgroup's contract)
    if ((int)threadIdx.x < n_targets) publish(&block_counts[(size_t)threadIdx.x * nblocks + blockIdx.x], s_cnt[threadIdx.x]);
    if (last_workgroup(tickets, (uint32_t)nblocks)) compact_scan_body(block_counts, nblocks, n_targets, row_offset);
}

__global__ __launch_bounds__(CT) void compact_scatter_kernel(const float4* __restrict__ rec, const uint32_t* __restrict__ slot_of,
                                                               int n, int n_targets, int nblocks, uint32_t* __restrict__ tickets,
                                                               const uint32_t* __restrict__ block_offsets,
                                                               uint32_t* __restrict__ out_idx,
                                                               float2* __restrict__ out_xy, float* __restrict__ out_err, int hi_prio) {
    helper_priority(hi_prio);
    // rows of target t in (pass k, wavefront w), then their exclusive prefix in keypoint order = (k, w) order
    __shared__ uint32_t s_cnt[CPASS * (CT / 64)][kRecStride];
    const int t = threadIdx.x & 7, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the tickets of the count kernel are zero again for the next launch on this lane
    if (blockIdx.x == 0)
        for (uint32_t k = threadIdx.x; k < last_workgroup_words((uint32_t)nblocks); k += CT) tickets[k] = 0u;
    float4 r[CPASS];
    unsigned long long bal[CPASS];
#pragma unroll
    for (int k = 0; k < CPASS; k++) {
        const int i = blockIdx.x * CF + k * (CT / kRecStride) + (int)(threadIdx.x >> 3);
        r[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool live = i < n && t < n_targets;
        if (live) r[k] = rec[(size_t)slot_of[i] * kRecStride + t];
        bal[k] = __ballot(live && __float_as_uint(r[k].w) == 1u);
        if (lane < kRecStride) s_cnt[k * (CT / 64) + wave][lane] = (uint32_t)__popcll(bal[k] & target_lanes(lane));
    }
    __syncthreads();
    {
        // lane (q, t): exclusive prefix over q' < q of target t (32 x 8 entries, one per lane)
        const int q = threadIdx.x >> 3;
        uint32_t run = 0;
        for (int qq = 0; qq < q; qq++) run += s_cnt[qq][t];
        __syncthreads();
        s_cnt[q][t] = run;
        __syncthreads();
    }
    const uint32_t base = (t < n_targets) ? block_offsets[(size_t)t * nblocks + blockIdx.x] : 0u;
#pragma unroll
    for (int k = 0; k < CPASS; k++) {
        if ((bal[k] >> lane) & 1ull) {
            const int i = blockIdx.x * CF + k * (CT / kRecStride) + (int)(threadIdx.x >> 3);
            const uint32_t pos = base + s_cnt[k * (CT / 64) + wave][t] +
                                 (uint32_t)__popcll(bal[k] & target_lanes(t) & ((1ull << lane) - 1ull));
            out_idx[pos] = (uint32_t)i;
            out_xy[pos] = make_float2(r[k].x, r[k].y);
            out_err[pos] = r[k].z;
        }
    }
}

// raw records -> the [target][n] arrays of pc_lk_track, keypoint order
__global__ __launch_bounds__(256) void unpack_records_kernel(const float4* __restrict__ rec, const uint32_t* __restrict__ slot_of,
                                                             int n, float2* __restrict__ xy, uint8_t* __restrict__ status,
                                                             float* __restrict__ err, int hi_prio) {
    helper_priority(hi_prio);
    const int i = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (i >= n) return;
    const float4 r = rec[(size_t)slot_of[i] * kRecStride + t];
    const size_t o = (size_t)t * n + i;
    xy[o] = make_float2(r.x, r.y);
    status[o] = (uint8_t)__float_as_uint(r.w);
    err[o] = r.z;
}

void launch_unpack_records(const float4* rec, const uint32_t* slot_of, int n, int n_targets, float2* xy, uint8_t* status,
                           float* err, hipStream_t s) {
    if (n <= 0 || n_targets <= 0) return;
    hipLaunchKernelGGL(unpack_records_kernel, dim3((n + 255) / 256, n_targets), dim3(256), 0, s, rec, slot_of, n, xy, status, err, helper_prio_arg());
}

// keypoints of frame1 -> the job's packed record buffer (device to device, 16 bytes per lane)
__global__ __launch_bounds__(256) void copy_keypoints_kernel(const float2* __restrict__ src, float2* __restrict__ dst, int n, int hi_prio) {
    helper_priority(hi_prio);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // one pair of keypoints
    if (2 * i + 1 < n) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
    else if (2 * i < n) dst[2 * i] = src[2 * i];
}
void launch_copy_keypoints(const float2* src, float2* dst, int n, hipStream_t s) {
    if (n <= 0) return;
    const int pairs = (n + 1) / 2;
    hipLaunchKernelGGL(copy_keypoints_kernel, dim3((pairs + 255) / 256), dim3(256), 0, s, src, dst, n, helper_prio_arg());
}

void launch_compact(const float4* rec, const uint32_t* slot_of, int n, int n_targets, uint32_t* scratch, bool scratch_fresh,
                    long long* row_offset, uint32_t* out_idx, float2* out_xy, float* out_err, hipStream_t s) {
    const int nblocks = compact_num_blocks(n);
    uint32_t* const tickets = scratch;
    uint32_t* const block_counts = scratch + kCompactTicketWords;
    // tickets: zero at the first use of a (re)allocated scratch buffer -- also when this call has nothing to compact:
    // the next one no longer knows that the buffer is new --, afterwards the scatter kernel leaves them zero
    if (scratch_fresh) (void)hipMemsetAsync(tickets, 0, (size_t)kCompactTicketWords * sizeof(uint32_t), s);
    if (nblocks == 0 || n_targets <= 0) {
        (void)hipMemsetAsync(row_offset, 0, (size_t)(std::max(n_targets, 0) + 1) * sizeof(long long), s);
        return;
    }
    static_assert((long long)(kCompactTicketWords - 1) * kTicketGroup * CF >= kCompactMaxKeypoints, "tickets for kCompactMaxKeypoints");
    hipLaunchKernelGGL(compact_count_kernel, dim3(nblocks), dim3(CT), 0, s, rec, slot_of, n, n_targets, nblocks, tickets, block_counts,
                       row_offset, helper_prio_arg());
    hipLaunchKernelGGL(compact_scatter_kernel, dim3(nblocks), dim3(CT), 0, s, rec, slot_of, n, n_targets, nblocks, tickets,
                       block_counts, out_idx, out_xy, out_err, helper_prio_arg());
}

}  // namespace pc
