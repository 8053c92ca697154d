// lk4_kernel.hpp -- pyramidal Lucas-Kanade (K8-K10) for the windows the two-keypoint kernel does not take: one keypoint per
// wavefront, EIGHT lanes per target, on the uint16 planes.  Windows 3 and 11 .. PC_MAX_WINDOW (31).
//
// Same arithmetic and results as kernels_lk.hip / kernels_lk3.hip (bit for bit; all follow oracle/pc_oracle.c, which restates
// cv::calcOpticalFlowPyrLK as called at reference cpp/opticalflow.cc:119-125 -- OpticalFlowOptions.window_size is a free
// read-write attribute there, cpp/opticalflow.h:27-33, and OpenCV's own default is 21).
//
// Mapping.  Group g = lanes 8g .. 8g + 7 tracks the wavefront's keypoint into target g.  Lane lg of a group owns the window
// COLUMNS lg, lg + 8, lg + 16, lg + 24 ("chains": the bottom taps of a row are the top taps of the next one, one aligned
// ds_read_b32 per pixel).  With 8 lanes per pair a 15-px window is 28-30 pixels per lane -- what the two-keypoint kernel's 4
// lanes carry at its 10-px window: the same share of per-iteration set-up, half the pairs per wavefront.
//   * J regions: (WIN + 3) rows x (WIN + 2) positions per group in LDS, one DWORD per position = (p[c] << 7) | (p[c+1] << 7) << 16,
//     the operand format of v_dot2_i32_i16 (kernels_lk3.hip); staged in 8-pixel chunks, consecutive lanes take consecutive
//     chunks (the lanes of a group then touch two or three plane rows per load instruction, not eight).
//   * The I side (window patch, Scharr patch, structure tensor) is evaluated once per level by the whole wavefront and handed to
//     the groups through LDS in the layout the iteration consumes: one 16-byte ENTRY per (chain, step, lane) =
//     {bias of the upper-run pixel, bias of the lower-run pixel, (ix_upper, ix_lower), (iy_upper, iy_lower)} -- a chain is walked as
//     two runs (rows 0 .. H1-1 and H1 .. WIN-1) whose step-s pixels share the v_perm + 2 v_dot2 that accumulate the mismatch
//     vector (3 instructions per two pixels).  Windows up to 21 px keep their entries in REGISTERS for the level (64 VGPRs at
//     16 px, 132 at 21), larger ones read them with one ds_read_b128 per two pixels (a 31-px window would need 248 registers;
//     24 px in registers: one wavefront per SIMD, measured 4-12 % slower).
//   * Sums are exact integers, reduced over the group with DPP adds, ONE rounding (== the oracle's (float)(int64)).
//   * X86 (PC_ARITH_LK_X86_ORDER): the canonical data path plus the PROOF that OpenCV's fp32 lane sums would be exact
//     (kernels_lk3.hip: S11, S22 < 2^24 per level, sum d^2 * max(S11, S22) <= 2^48 per iteration); where it fails, the sums
//     in the x86 order: vector lane j = the pair sums of columns (j, j + 4) of every 8-column block, row by row -- lanes j and
//     j + 4 of the group own those columns, one DPP row shift brings the partner's product over --; the scalar accumulator
//     over the columns past the last full block travels from lane to lane with v_add_f32 + DPP row shifts, in row-major order.
#pragma once

#include <cmath>
#include <limits>

#include "lk_common.hpp"

namespace pc {

template <int WIN>
struct LK4Geo {
    static constexpr int GL = 8, NPX = WIN * WIN;
    static constexpr int NCH = (WIN + GL - 1) / GL;            // column chains of a lane: columns lg + 8 m
    static constexpr int NLAST = WIN - GL * (NCH - 1);         // lanes that own a column of the last chain
    static constexpr int H1 = (WIN + 1) / 2, H2 = WIN - H1;    // rows of a chain's upper / lower run
    static constexpr int MX = 1, MY = 1;                       // search margin of a staged region
    static constexpr int RWP = WIN + 2 * MX;                   // positions per region row
    static constexpr int RH = WIN + 1 + 2 * MY;                // region rows
    static constexpr int CHK = (RWP + 7) / 8;                  // 8-position chunks per region row (a chunk loads 10 pixels)
    static constexpr int PITCH = 4 * ((RWP + 3) / 4);          // dwords per region row
    static constexpr int J_DW = RH * PITCH;                    // per-group J region (16-byte multiple)
    static constexpr int I_ROWS = WIN + 1, I_CHK = (WIN + 1 + 7) / 8, I_PITCH = 4 * ((WIN + 1 + 3) / 4);
    static constexpr int I_DW = I_ROWS * I_PITCH;              // I window, same format
    static constexpr int D_PITCH = WIN + 1, D_DW = (((WIN + 1) * (WIN + 1)) + 3) & ~3;   // raw Scharr window
    static constexpr int X_DW = NCH * H1 * GL * 4;             // the entries (see the header)
    static constexpr bool REG = WIN <= 21;                     // entries in registers for the level
    // x86 order
    static constexpr int SIMD_W = (WIN / 8) * 8, NB = SIMD_W / 8, NXS = WIN - SIMD_W, NS = NXS * WIN;
    static constexpr int CL = WIN * (SIMD_W / 4);              // terms of a vector lane's chain (structure tensor)
    static constexpr int PA_DW = 3 * NPX + 16;                 // LDS of the ordered structure tensor
    static constexpr int max3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
    static constexpr int AREA_DW = (max3(8 * J_DW, I_DW + D_DW, PA_DW) + 3) & ~3;   // regions / I-side windows / ordered tensor
    static constexpr int WAVE_DW = X_DW + AREA_DW;
};

// ROWS x CHK chunks of 8 positions from a uint16 plane into position dwords (row pitch PITCH dwords), NL lanes; the
// positions of a row's last chunk past PITCH are not written.  Each chunk reads 10 pixels (20 bytes, 2-byte aligned).
template <int NL, int ROWS, int CHK, int PITCH>
__device__ __forceinline__ void lk4_stage(const uint16_t* __restrict__ src, int pitch, uint32_t* dst, int l) {
    constexpr int TOTAL = ROWS * CHK, TRIPS = (TOTAL + NL - 1) / NL;
    constexpr bool HALF_LAST = PITCH - 8 * (CHK - 1) <= 4;
    constexpr int B = TRIPS < 6 ? TRIPS : 6;   // trips in flight (5 VGPRs each)
    struct __attribute__((packed, aligned(2))) Raw { uint32_t d[5]; };
    asm volatile("" : "+v"(l));   // the (row, chunk) of a trip are per-lane values: keep them out of the callers' loops
#pragma unroll
    for (int k0 = 0; k0 < TRIPS; k0 += B) {
        Raw v[B];
        int r[B], c[B];
#pragma unroll
        for (int b = 0; b < B; b++) {
            if (k0 + b < TRIPS) {
                // lanes past the end repeat the last chunk (same address, same values) instead of branching
                const int t = min(l + NL * (k0 + b), TOTAL - 1);
                r[b] = t / CHK;
                c[b] = t - r[b] * CHK;
                v[b] = *reinterpret_cast<const Raw*>(src + (ptrdiff_t)__mul24(r[b], pitch) + 8 * c[b]);
            }
        }
#pragma unroll
        for (int b = 0; b < B; b++) {
            if (k0 + b < TRIPS) {
                const uint32_t d0 = v[b].d[0], d1 = v[b].d[1], d2 = v[b].d[2], d3 = v[b].d[3], d4 = v[b].d[4];
                uint32_t* o = dst + r[b] * PITCH + 8 * c[b];
                *reinterpret_cast<uint4*>(o) = make_uint4(d0, __builtin_amdgcn_alignbit(d1, d0, 16), d1, __builtin_amdgcn_alignbit(d2, d1, 16));
                if (!HALF_LAST || c[b] != CHK - 1)
                    *reinterpret_cast<uint4*>(o + 4) = make_uint4(d2, __builtin_amdgcn_alignbit(d3, d2, 16), d3, __builtin_amdgcn_alignbit(d4, d3, 16));
            }
        }
    }
}

// R = 128 * (sum of the 4 weighted taps) + bias; hi16(R) is the CV_DESCALEd sample minus the I value (kernels_lk3.hip)
__device__ __forceinline__ int lk4_interp_r(uint32_t top, uint32_t bot, uint32_t r0, uint32_t r1, int bias) {
    const int t = __builtin_amdgcn_sdot2(__builtin_bit_cast(pc_short2, top), __builtin_bit_cast(pc_short2, r0), bias, true);
    return sdot2(bot, r1, t);
}
// the bilinear weights as signed 16-bit pairs (w00, w01), (w10, w11): bilinear_weights(a, b).r0 / .r1 (kernels_lk3.hip: packed_weights)
__device__ __forceinline__ void lk4_weights(float a, float b, uint32_t& r0, uint32_t& r1) {
    constexpr float S = (float)(1 << W_BITS), M = 12582912.f;
    constexpr uint32_t MB = 0x4B400000u;   // bits of M
    const float na = 1.f - a, nbs = (1.f - b) * S, bs = b * S;
    const uint32_t t00 = __float_as_uint(na * nbs + M), t01 = __float_as_uint(a * nbs + M), t10 = __float_as_uint(na * bs + M);
    const uint32_t w11 = ((1u << W_BITS) + 3u * MB) - (t00 + t01 + t10);
    r0 = __builtin_amdgcn_perm(t01, t00, 0x05040100u);
    r1 = __builtin_amdgcn_perm(w11, t10, 0x05040100u);
}
__device__ __forceinline__ int lk4_bias_of(int ival) { return (1 << 15) - (ival << 16); }
// acc + hi16(a) * lo16(b) / hi16(a) * hi16(b)
__device__ __forceinline__ int lk4_mad_hl(int a, uint32_t b, int acc) {
    int d;
    asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(acc));
    return d;
}
__device__ __forceinline__ int lk4_mad_hh(int a, uint32_t b, int acc) {
    int d;
    asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(acc));
    return d;
}
template <int CTRL>
__device__ __forceinline__ float lk4_dpp_f32(float v) { return __int_as_float(dpp_i32<CTRL>(__float_as_int(v))); }

// the x86-ordered structure tensor (PC_ARITH_LK_X86_ORDER where S11 / S22 >= 2^24): every lane writes the three products of the
// pixels it evaluated to their place in their chain (vector lane x & 3 over the first SIMD_W columns row by row, the scalar
// accumulator over the rest in row-major order); lane 5 k + c adds chain c of quantity k sequentially;
// A = scalar + (((q0 + q1) + q2) + q3).  Called by the whole wavefront.
template <int WIN, int KW>
__device__ __forceinline__ void lk4_x86_structure_tensor(float* pa, const uint32_t (&dxy)[KW], int lane, float& S11, float& S12, float& S22) {
    using G = LK4Geo<WIN>;
    constexpr int NPX = G::NPX;
    constexpr int MAXL = G::CL > G::NS ? G::CL : G::NS;
#pragma unroll
    for (int m = 0; m < KW; m++) {
        const int q = lane + 64 * m;
        if (q < NPX) {
            const int y = q / WIN, x = q - y * WIN;
            const int ix = (int)(int16_t)(dxy[m] & 0xffffu), iy = (int)dxy[m] >> 16;
            const int off = x < G::SIMD_W ? (x & 3) * G::CL + y * (G::SIMD_W / 4) + (x >> 2) : 4 * G::CL + y * G::NXS + (x - G::SIMD_W);
            pa[off] = (float)__mul24(ix, ix);              // |products| <= 4080^2 < 2^24: exact
            pa[NPX + off] = (float)__mul24(ix, iy);
            pa[2 * NPX + off] = (float)__mul24(iy, iy);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane < 15) {
        const int k = lane / 5, c = lane - 5 * k;
        const int len = c < 4 ? G::CL : G::NS;
        const float* src = pa + k * NPX + c * G::CL;
        float acc = 0.f;
#pragma unroll 8
        for (int i = 0; i < MAXL; i++) {
            const float v = src[i < len ? i : 0];
            acc = i < len ? v + acc : acc;
        }
        pa[3 * NPX + lane] = acc;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const float* r = pa + 3 * NPX;
    S11 = r[4] + (((r[0] + r[1]) + r[2]) + r[3]);
    S12 = r[9] + (((r[5] + r[6]) + r[7]) + r[8]);
    S22 = r[14] + (((r[10] + r[11]) + r[12]) + r[13]);
}

// smallest float x with fl(x / c) >= thr (c > 0, thr finite and positive); NaN if the search does not settle
static inline float lk4_division_threshold(float thr, float c) {
    if (!(thr > 0.f) || !std::isfinite(thr)) return std::numeric_limits<float>::quiet_NaN();
    float x = thr * c;
    if (!std::isfinite(x)) return std::numeric_limits<float>::quiet_NaN();
    for (int i = 0; i < 64 && !(x / c >= thr); i++) x = std::nextafterf(x, std::numeric_limits<float>::infinity());
    for (int i = 0; i < 64; i++) {
        const float y = std::nextafterf(x, -std::numeric_limits<float>::infinity());
        if (!(y / c >= thr)) break;
        x = y;
    }
    const float below = std::nextafterf(x, -std::numeric_limits<float>::infinity());
    if (!(x / c >= thr) || (below / c >= thr)) return std::numeric_limits<float>::quiet_NaN();
    return x;
}

template <int WIN, bool X86>
__global__ __launch_bounds__(64, (WIN <= 16 ? 3 : (WIN <= 21 ? 2 : 1))) void lk4_kernel(const LKParams p) {
    using G = LK4Geo<WIN>;
    constexpr int GL = G::GL, NPX = G::NPX, NCH = G::NCH, H1 = G::H1, H2 = G::H2;
    constexpr int KW = (NPX + 63) / 64;   // pixels per lane in the wave-wide I-side pass
    constexpr bool PARTIAL = G::NLAST < GL;   // the last chain has lanes without a column
    constexpr int NE = G::REG ? NCH * H1 : 1;
    __shared__ __attribute__((aligned(16))) uint32_t s_buf[G::WAVE_DW];   // one wavefront per workgroup

    const int lane = threadIdx.x & 63;
    const int grp = lane >> 3, lg = lane & 7;
    // Workgroup b runs on XCD b % 8; each XCD takes one contiguous eighth of the (spatially binned) keypoint order
    lk_signal_dispatched(p);
    const int slot = (int)(blockIdx.x & 7u) * p.blocks_per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= p.blocks_per_xcd || slot >= p.n) return;
    const int feat = p.perm ? (int)p.perm[slot] : slot;
    const bool tgt_active = grp < p.n_targets;
    const int tgt = tgt_active ? grp : 0;

    uint32_t* const xbuf = s_buf;                                   // the entries: live for the whole level
    uint32_t* const area = s_buf + G::X_DW;
    uint32_t* const ibuf = area;                                    // I window, position dwords
    int32_t* const dbuf = reinterpret_cast<int32_t*>(area + G::I_DW);   // raw Scharr window
    uint32_t* const jbuf = area + grp * G::J_DW;                    // this group's J region: aliases the two above

    // a lane without a column in the last chain reads another lane's (its pixels are masked out of every sum)
    const bool last_valid = !PARTIAL || lg < G::NLAST;
    const int col_last = GL * (NCH - 1) + (last_valid ? lg : 0);

    const float2 pt = p.pts[feat];
    const float half_win = (float)(WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    float nx = 0.f, ny = 0.f;
    bool status = true;
    float err = 0.f;
    int x86_levels = 0, x86_iters = 0;   // diagnostics (LKParams::x86_stats), see kernels_lk3.hip
    (void)x86_levels;
    (void)x86_iters;

    for (int level = p.max_level; level >= 0; --level) {
        const Level L = p.src[level];
        const uint16_t* __restrict__ J16 = p.tgt16[tgt][level];
        const int pitch = L.pitch;
        const float lscale = __uint_as_float((uint32_t)(127 - level) << 23);   // 2^-level
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

        // ---- I side: identical for all targets -> computed once by the whole wavefront ----
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
        uint32_t wI0, wI1;
        lk4_weights(px - (float)ipx, py - (float)ipy, wI0, wI1);

        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the previous level's J regions are dead
        {
            DerivWindow<WIN, 64> dw;
            dw.load(L.der + (ptrdiff_t)(__mul24(ipy, pitch) + ipx), pitch, lane);
            lk4_stage<64, G::I_ROWS, G::I_CHK, G::I_PITCH>(L.img16 + (ptrdiff_t)__mul24(ipy, pitch) + ipx, pitch, ibuf, lane);
            dw.store(dbuf, lane);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        int sA11 = 0, sA12 = 0, sA22 = 0;
        uint32_t dxy_mine[KW];   // X86: the derivatives of the pixels this lane evaluated (for the ordered structure tensor)
#pragma unroll
        for (int m = 0; m < KW; m++) {
            const int q = lane + 64 * m;
            dxy_mine[m] = 0u;
            if (q < NPX) {
                const int y = q / WIN, x = q - y * WIN;
                const uint32_t* qp = ibuf + y * G::I_PITCH + x;
                const int ival = lk4_interp_r(qp[0], qp[G::I_PITCH], wI0, wI1, 1 << 15) >> 16;
                const uint32_t* d = reinterpret_cast<const uint32_t*>(dbuf) + y * G::D_PITCH + x;
                const uint32_t d00 = d[0], d01 = d[1], d10 = d[G::D_PITCH], d11 = d[G::D_PITCH + 1];
                // (dx00, dx01), (dx10, dx11), (dy00, dy01), (dy10, dy11)
                const uint32_t dx0 = __builtin_amdgcn_perm(d01, d00, 0x05040100u);
                const uint32_t dx1 = __builtin_amdgcn_perm(d11, d10, 0x05040100u);
                const uint32_t dy0 = __builtin_amdgcn_perm(d01, d00, 0x07060302u);
                const uint32_t dy1 = __builtin_amdgcn_perm(d11, d10, 0x07060302u);
                const int ix = sdot2(dx1, wI1, sdot2(dx0, wI0, 1 << (W_BITS - 1))) >> W_BITS;
                const int iy = sdot2(dy1, wI1, sdot2(dy0, wI0, 1 << (W_BITS - 1))) >> W_BITS;
                // the pixel's place in its owner's entries: chain x / 8, lane x % 8, run y / H1, step y % H1
                const bool up = y < H1;
                uint32_t* e = xbuf + (((x >> 3) * H1 + (up ? y : y - H1)) * GL + (x & 7)) * 4;
                e[up ? 0 : 1] = (uint32_t)lk4_bias_of(ival);
                uint16_t* e16 = reinterpret_cast<uint16_t*>(e);
                e16[4 + (up ? 0 : 1)] = (uint16_t)ix;
                e16[6 + (up ? 0 : 1)] = (uint16_t)iy;
                dxy_mine[m] = (uint32_t)(ix & 0xffff) | ((uint32_t)iy << 16);
                sA11 += __mul24(ix, ix);   // |ix|, |iy| <= 4080: a lane's KW <= 16 pixels stay below 2^31
                sA12 += __mul24(ix, iy);
                sA22 += __mul24(iy, iy);
            }
        }
        // the wavefront's totals as exact (hi, lo) halves: NPX * 4080^2 does not fit int32 past an 11-px window
        const int h11 = wave_sum_i32(sA11 >> 16), l11 = wave_sum_i32(sA11 & 0xffff);
        const int h22 = wave_sum_i32(sA22 >> 16), l22 = wave_sum_i32(sA22 & 0xffff);
        const long long S11 = ((long long)h11 << 16) + l11, S22 = ((long long)h22 << 16) + l22;
        float A11 = exact_sum_to_float(h11, l11) * FLT_SCALE;
        float A12 = wave_exact_sum(sA12) * FLT_SCALE;
        float A22 = exact_sum_to_float(h22, l22) * FLT_SCALE;
        float cert_s = 0.f;   // X86: max(S11, S22), the structure-tensor factor of the mismatch vector's exactness bound
        if constexpr (X86) {
            // S11, S22 < 2^24: every partial sum of ix^2, iy^2 and ix iy in any order is exact -- the x86 order gives the
            // canonical values above.  Otherwise the ordered evaluation (wave-uniform: one keypoint per wavefront).
            const long long smax = S11 > S22 ? S11 : S22;
            cert_s = (float)smax;
            if (smax >= (1ll << 24)) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the I / Scharr windows are consumed
                float f11, f12, f22;
                lk4_x86_structure_tensor<WIN, KW>(reinterpret_cast<float*>(area), dxy_mine, lane, f11, f12, f22);
                A11 = f11 * FLT_SCALE;
                A12 = f12 * FLT_SCALE;
                A22 = f22 * FLT_SCALE;
                x86_levels += 1 << 16;
            }
            x86_levels += 1;
        }
        // |sum over any subset of the window of diff * ix| <= sqrt(NPX) * 8160 * sqrt(S11) (Cauchy-Schwarz, |diff| <= 8160):
        // below 2^31 the mismatch vector's sums can be formed in int32 (then ONE conversion, the same single rounding)
        constexpr long long kSmallS = (1ll << 62) / ((long long)NPX * 8160 * 8160);
        const bool int_sums = S11 < kSmallS && S22 < kSmallS;   // wave-uniform
        float D = A11 * A22 - A12 * A12;
        const float tdiff = A11 - A22;
        const float min_eig_num = A22 + A11 - sqrtf(tdiff * tdiff + 4.f * A12 * A12);
        // min_eig = min_eig_num / (2 WIN^2) < thr, without the division (LKParams::min_eig_num_thr)
        const bool weak = (p.min_eig_num_thr == p.min_eig_num_thr) ? (min_eig_num < p.min_eig_num_thr)
                                                                    : (min_eig_num / (float)(2 * WIN * WIN) < p.min_eig_thr);
        if (weak || D < 1.1920928955078125e-07f /* FLT_EPSILON */) {   // wave-uniform
            if (level == 0) status = false;
            continue;
        }
        D = 1.f / D;
        D *= FLT_SCALE;   // b = sum * 2^-20 enters the solve only through products that are multiplied by D (kernels_lk3.hip)
        // the J regions alias the I-side windows (and the ordered tensor's products): all lanes are done with them
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (!tgt_active) continue;  // idle groups only help with the I side

        // entry (chain m, step st) of this lane: {bias upper, bias lower, (ix upper, ix lower), (iy upper, iy lower)}
        const uint32_t* const ent = xbuf + lg * 4;
        uint4 ereg[NE];
        if constexpr (G::REG) {
#pragma unroll
            for (int k = 0; k < NE; k++) ereg[k] = *reinterpret_cast<const uint4*>(ent + k * (GL * 4));
        }
        auto entry = [&](int m, int st) -> uint4 {
            if constexpr (G::REG) return ereg[m * H1 + st];
            else return *reinterpret_cast<const uint4*>(ent + (m * H1 + st) * (GL * 4));
        };
        auto bias_at = [&](int m, int r) -> int {   // the bias of row r of chain m
            if constexpr (G::REG) return (int)(r < H1 ? ereg[m * H1 + r].x : ereg[m * H1 + r - H1].y);
            else return (int)ent[(m * H1 + (r < H1 ? r : r - H1)) * (GL * 4) + (r < H1 ? 0 : 1)];
        };

        // ---- iterations on the staged J region ----
        qx -= half_win;
        qy -= half_win;
        float pdx = 0.f, pdy = 0.f;
        int rx0 = 0, ry0 = 0;
        bool staged = false;
        bool x86_ordered = false;   // X86: after the first failed proof the rest of the level runs in the x86 order (wave-uniform)
        (void)x86_ordered;
        for (int j = 0; j < p.max_iters; j++) {
            const float fqx = floorf(qx), fqy = floorf(qy);
            const int iqx = (int)fqx, iqy = (int)fqy;
            if ((unsigned)(iqx + WIN) >= (unsigned)(L.w + WIN) || (unsigned)(iqy + WIN) >= (unsigned)(L.h + WIN)) {
                if (level == 0) status = false;
                break;
            }
            int ox = iqx - rx0, oy = iqy - ry0;
            if (!staged || (unsigned)ox > (unsigned)(2 * G::MX) || (unsigned)oy > (unsigned)(2 * G::MY)) {
                ox = G::MX;
                oy = G::MY;
                rx0 = iqx - ox;
                ry0 = iqy - oy;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                lk4_stage<GL, G::RH, G::CHK, G::PITCH>(J16 + (ptrdiff_t)__mul24(ry0, pitch) + rx0, pitch, jbuf, lg);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                staged = true;
            }
            uint32_t wJ0, wJ1;
            lk4_weights(qx - fqx, qy - fqy, wJ0, wJ1);
            const uint32_t* jq = jbuf + __mul24(oy, G::PITCH) + ox;
            float b1 = 0.f, b2 = 0.f;
            bool run_ordered = X86 && x86_ordered;
            if (!run_ordered) {
                // Canonical: exact integer sums.  Every chain is walked as two runs (upper / lower rows); step st handles
                // pixel st of all runs: first dot products, second dot products, then the accumulations in pairs.
                int sb1[NCH], sb2[NCH], dd[NCH];
                uint32_t top[2 * NCH];
                const uint32_t* rb[2 * NCH];
#pragma unroll
                for (int u = 0; u < 2 * NCH; u++) {
                    const int m = u >> 1;
                    rb[u] = jq + (m == NCH - 1 ? col_last : lg + GL * m) + ((u & 1) ? H1 * G::PITCH : 0);
                    top[u] = rb[u][0];
                }
#pragma unroll
                for (int m = 0; m < NCH; m++) sb1[m] = sb2[m] = dd[m] = 0;
#pragma unroll
                for (int st = 0; st < H1; st++) {
                    int R[2 * NCH];
                    uint4 e[NCH];
#pragma unroll
                    for (int m = 0; m < NCH; m++) e[m] = entry(m, st);
#pragma unroll
                    for (int u = 0; u < 2 * NCH; u++) {
                        const int len = (u & 1) ? H2 : H1;
                        if (st < len) {
                            const uint32_t bot = rb[u][(st + 1) * G::PITCH];
                            R[u] = lk4_interp_r(top[u], bot, wJ0, wJ1, (int)((u & 1) ? e[u >> 1].y : e[u >> 1].x));
                            top[u] = bot;
                        }
                    }
#pragma unroll
                    for (int m = 0; m < NCH; m++) {
                        if (st < H2) {
                            uint32_t Rp = __builtin_amdgcn_perm((uint32_t)R[2 * m + 1], (uint32_t)R[2 * m], 0x07060302u);
                            if (PARTIAL && m == NCH - 1) Rp = last_valid ? Rp : 0u;
                            sb1[m] = sdot2(Rp, e[m].z, sb1[m]);
                            sb2[m] = sdot2(Rp, e[m].w, sb2[m]);
                            if constexpr (X86) dd[m] = sdot2(Rp, Rp, dd[m]);
                        } else {   // an odd window's middle row: the upper run only
                            int Ru = R[2 * m];
                            if (PARTIAL && m == NCH - 1) Ru = last_valid ? Ru : 0;
                            sb1[m] = lk4_mad_hl(Ru, e[m].z, sb1[m]);
                            sb2[m] = lk4_mad_hl(Ru, e[m].w, sb2[m]);
                            if constexpr (X86) dd[m] = lk4_mad_hh(Ru, (uint32_t)Ru, dd[m]);
                        }
                    }
                }
                // a chain's sums fit int32 (31 * 8160 * 4080 < 2^31); the lane's and the group's only when int_sums
                if (int_sums) {
                    int t1 = sb1[0], t2 = sb2[0];
#pragma unroll
                    for (int m = 1; m < NCH; m++) {
                        t1 += sb1[m];
                        t2 += sb2[m];
                    }
                    b1 = (float)group_allreduce_add<GL>(t1);
                    b2 = (float)group_allreduce_add<GL>(t2);
                } else {
                    int h1 = 0, l1 = 0, h2 = 0, l2 = 0;
#pragma unroll
                    for (int m = 0; m < NCH; m++) {
                        h1 += sb1[m] >> 16;
                        l1 += sb1[m] & 0xffff;
                        h2 += sb2[m] >> 16;
                        l2 += sb2[m] & 0xffff;
                    }
                    b1 = exact_sum_to_float(group_allreduce_add<GL>(h1), group_allreduce_add<GL>(l1));
                    b2 = exact_sum_to_float(group_allreduce_add<GL>(h2), group_allreduce_add<GL>(l2));
                }
                if constexpr (X86) {
                    // sum over the window of |d ix| <= sqrt(sum d^2 * S11) <= 2^24 (and the same for iy): every partial sum of the
                    // x86 order is exact and its result is the float of the integer total, b1 / b2 above.  The margin (2^-8)
                    // covers the roundings of this test itself (a dozen fp32 operations, each < 2^-23 relative).
                    float ddf = (float)dd[0];   // a chain's sum of squares fits int32: 31 * 8160^2 < 2^31
#pragma unroll
                    for (int m = 1; m < NCH; m++) ddf += (float)dd[m];
                    ddf += lk4_dpp_f32<0xB1>(ddf);
                    ddf += lk4_dpp_f32<0x4E>(ddf);
                    ddf += lk4_dpp_f32<0x141>(ddf);
                    const bool proven = ddf * cert_s <= 281474976710656.f * (1.f - 1.f / 256.f);
                    if (__any(!proven)) {
                        x86_ordered = true;
                        run_ordered = true;
                    }
                }
            }
            if constexpr (X86) {
                if (run_ordered) {
                    // The x86 order.  Rows in order; per row the lane's pixel of every chain: P = d * (ix, iy).  Chains of full
                    // 8-column blocks: lanes 0-3 add the product of lane + 4 (the int32 pair sum of columns c and c + 4),
                    // convert, accumulate -- vector lane c.  The chain of the remaining columns: the scalar accumulator,
                    // handed from lane to lane in column order (row shift by one; from the last column back to lane 0).
                    float q1 = 0.f, q2 = 0.f, s1 = 0.f, s2 = 0.f;
                    uint32_t top[NCH];
                    const uint32_t* cb[NCH];
#pragma unroll
                    for (int m = 0; m < NCH; m++) {
                        cb[m] = jq + (m == NCH - 1 ? col_last : lg + GL * m);
                        top[m] = cb[m][0];
                    }
#pragma unroll
                    for (int r = 0; r < WIN; r++) {
                        const int st = r < H1 ? r : r - H1;
#pragma unroll
                        for (int m = 0; m < NCH; m++) {
                            const uint4 e = entry(m, st);
                            const uint32_t bot = cb[m][(r + 1) * G::PITCH];
                            int R = lk4_interp_r(top[m], bot, wJ0, wJ1, (int)(r < H1 ? e.x : e.y));
                            top[m] = bot;
                            if (PARTIAL && m == NCH - 1) R = last_valid ? R : 0;
                            const int P1 = r < H1 ? lk4_mad_hl(R, e.z, 0) : lk4_mad_hh(R, e.z, 0);
                            const int P2 = r < H1 ? lk4_mad_hl(R, e.w, 0) : lk4_mad_hh(R, e.w, 0);
                            if (m < G::NB) {
                                const int T1 = P1 + dpp_i32<0x104>(P1), T2 = P2 + dpp_i32<0x104>(P2);   // row_shl:4: lane l takes lane l + 4
                                q1 += (float)T1;
                                q2 += (float)T2;
                            } else {
                                const float f1 = (float)P1, f2 = (float)P2;
#pragma unroll
                                for (int i = 0; i < G::NXS; i++) {
                                    if (G::NXS == 1) {
                                        s1 = s1 + f1;
                                        s2 = s2 + f2;
                                    } else if (i == 0) {
                                        s1 = lk4_dpp_f32<0x100 + (G::NXS > 1 ? G::NXS - 1 : 1)>(s1) + f1;   // row_shl:(NXS - 1): lane 0 takes the last column's lane
                                        s2 = lk4_dpp_f32<0x100 + (G::NXS > 1 ? G::NXS - 1 : 1)>(s2) + f2;
                                    } else {
                                        s1 = lk4_dpp_f32<0x111>(s1) + f1;   // row_shr:1: lane l takes lane l - 1
                                        s2 = lk4_dpp_f32<0x111>(s2) + f2;
                                    }
                                }
                            }
                        }
                    }
                    if constexpr (G::NB > 0) {
                        // bbuf[k] = qb0[k] + qb1[k]: lanes (0, 2) and (1, 3); fb += bbuf[0] + bbuf[2]
                        const float t1 = q1 + lk4_dpp_f32<0x4E>(q1), t2 = q2 + lk4_dpp_f32<0x4E>(q2);
                        const float u1 = t1 + lk4_dpp_f32<0xB1>(t1), u2 = t2 + lk4_dpp_f32<0xB1>(t2);
                        b1 = __shfl(u1, 0, GL);
                        b2 = __shfl(u2, 0, GL);
                    }
                    if constexpr (G::NXS > 0) {
                        const float v1 = __shfl(s1, G::NXS - 1, GL), v2 = __shfl(s2, G::NXS - 1, GL);
                        b1 = G::NB > 0 ? v1 + b1 : v1;
                        b2 = G::NB > 0 ? v2 + b2 : v2;
                    }
                    x86_iters += 1 << 16;
                } else {
                    x86_iters += 1;
                }
            }
            const float dx = (A12 * b2 - A22 * b1) * D;
            const float dy = (A12 * b1 - A11 * b2) * D;
            qx += dx;
            qy += dy;
            nx = qx + half_win;
            ny = qy + half_win;
            if ((double)dx * (double)dx + (double)dy * (double)dy <= p.eps_sq) break;
            // |float| < 0.01 (a double) <=> |float| <= 0.01f: 0.01f = 0x1.47ae14p-7 is the largest float below 0.01
            if (j > 0 && fabsf(dx + pdx) <= 0x1.47ae14p-7f && fabsf(dy + pdy) <= 0x1.47ae14p-7f) {
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
            if (!staged || iex < rx0 || iex > rx0 + 2 * G::MX || iey < ry0 || iey > ry0 + 2 * G::MY) {
                rx0 = iex - G::MX;
                ry0 = iey - G::MY;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                lk4_stage<GL, G::RH, G::CHK, G::PITCH>(J16 + (ptrdiff_t)__mul24(ry0, pitch) + rx0, pitch, jbuf, lg);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                staged = true;
            }
            uint32_t wE0, wE1;
            lk4_weights(ex - (float)iex, ey - (float)iey, wE0, wE1);
            const uint32_t* jq = jbuf + (iey - ry0) * G::PITCH + (iex - rx0);
            int se = 0;
#pragma unroll
            for (int m = 0; m < NCH; m++) {
                const uint32_t* cb = jq + (m == NCH - 1 ? col_last : lg + GL * m);
                uint32_t top = cb[0];
                int sm = 0;
#pragma unroll
                for (int r = 0; r < WIN; r++) {
                    const uint32_t bot = cb[(r + 1) * G::PITCH];
                    const int diff = lk4_interp_r(top, bot, wE0, wE1, bias_at(m, r)) >> 16;
                    top = bot;
                    sm += diff < 0 ? -diff : diff;
                }
                se += (PARTIAL && m == NCH - 1 && !last_valid) ? 0 : sm;
            }
            se = group_allreduce_add<GL>(se);   // <= 961 * 8160 < 2^24: exact in fp32 too
            err = ((float)se * 1.f) / (float)(32 * WIN * WIN);
        }
    }

    // one 16-byte record per (slot, target): the wavefront's results are contiguous
    if (lg == 0 && tgt_active)
        p.out_rec[(size_t)slot * kRecStride + tgt] = make_float4(nx, ny, status ? err : 0.f, __uint_as_float(status ? 1u : 0u));
    if constexpr (X86) {
        if (p.x86_stats) {   // diagnostics only (pc_debug_lk_x86_stats)
            if (lg == 0 && tgt_active) {
                atomicAdd(&p.x86_stats[0], (unsigned long long)(x86_iters & 0xffff));
                atomicAdd(&p.x86_stats[1], (unsigned long long)(x86_iters >> 16));
            }
            if (lane == 0) {
                atomicAdd(&p.x86_stats[2], (unsigned long long)(x86_levels & 0xffff));
                atomicAdd(&p.x86_stats[3], (unsigned long long)(x86_levels >> 16));
            }
        }
    }
}

template <int WIN>
static void launch_lk4_t(const LKParams& p0, hipStream_t s) {
    LKParams p = p0;
    p.min_eig_num_thr = lk4_division_threshold(p.min_eig_thr, (float)(2 * WIN * WIN));
    const int blocks = p.n;   // one keypoint per wavefront, one wavefront per workgroup
    if (blocks == 0) return;
    p.blocks_per_xcd = (blocks + 7) / 8;
    if (p.x86_order) hipLaunchKernelGGL((lk4_kernel<WIN, true>), dim3((unsigned)p.blocks_per_xcd * 8u), dim3(64), 0, s, p);
    else hipLaunchKernelGGL((lk4_kernel<WIN, false>), dim3((unsigned)p.blocks_per_xcd * 8u), dim3(64), 0, s, p);
}

}  // namespace pc
