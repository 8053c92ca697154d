// lk_common.hpp -- device helpers shared by the pyramidal LK kernels (kernels_lk.hip: one keypoint per
// wavefront, 8 lanes per target; kernels_lk3.hip: two keypoints per wavefront, 4 lanes per target):
// fixed-point bilinear weights, interpolation on "byte pair" LDS rows, exact integer sums, staging of
// image regions into LDS.  Arithmetic follows oracle/pc_oracle.c (OpenCV's LKTrackerInvoker).
#pragma once

#include "kernels.hpp"

namespace pc {

constexpr int W_BITS = 14;
#ifndef PC_LK_STAGE_BATCH
#define PC_LK_STAGE_BATCH 13   // region rows a lane keeps in flight while staging (4 VGPRs each)
#endif
#ifndef PC_LK_STAGE_BATCH_IRR
#define PC_LK_STAGE_BATCH_IRR 3   // the same for windows whose regions do not map onto whole rows per group
#endif
#ifndef PC_LK_MARGIN
#define PC_LK_MARGIN 1
#endif

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
// all-reduce (sum) across a group of GL lanes (GL = 8 or 16, groups are aligned)
template <int GL>
__device__ __forceinline__ int group_allreduce_add(int v) {
    v += dpp_i32<0xB1>(v);    // quad_perm [1,0,3,2]  (lane ^ 1)
    v += dpp_i32<0x4E>(v);    // quad_perm [2,3,0,1]  (lane ^ 2)
    v += dpp_i32<0x141>(v);   // row_half_mirror      (other quad of the 8-lane half)
    if (GL == 16) v += dpp_i32<0x140>(v);  // row_mirror (other half of the row)
    return v;
}

// exact float of (hi * 2^16 + lo): each term is exactly representable (|hi| < 2^24, |lo| < 2^24),
// so the single fp32 add rounds the exact integer once (== (float)(int64) of the oracle).
__device__ __forceinline__ float exact_sum_to_float(int hi, int lo) {
    return (float)hi * 65536.f + (float)lo;
}
template <int GL>
__device__ __forceinline__ float group_exact_sum(int partial) {
    return exact_sum_to_float(group_allreduce_add<GL>(partial >> 16), group_allreduce_add<GL>(partial & 0xffff));
}

struct Weights {
    int w00, w01, w10, w11;   // w11 may be -1 (rounding of the other three), never smaller
    uint32_t r0, r1;          // signed 16-bit pairs (w00, w01) and (w10, w11) for v_dot2_i32_i16
    bool neg11;
};
__device__ __forceinline__ Weights bilinear_weights(float a, float b) {
    Weights w;
    w.w00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
    w.w01 = __float2int_rn(a * (1.f - b) * (float)(1 << W_BITS));
    w.w10 = __float2int_rn((1.f - a) * b * (float)(1 << W_BITS));
    w.w11 = (1 << W_BITS) - w.w00 - w.w01 - w.w10;
    w.neg11 = w.w11 < 0;
    w.r0 = (uint32_t)w.w00 | ((uint32_t)w.w01 << 16);             // 0 <= w00, w01, w10 <= 2^14
    w.r1 = (uint32_t)w.w10 | ((uint32_t)w.w11 << 16);             // w11 == -1 -> 0xffff in the high half
    return w;
}

typedef short pc_short2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int sdot2(uint32_t a, uint32_t b, int c) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(pc_short2, a), __builtin_bit_cast(pc_short2, b), c, false);
}

// (A[c], A[c+1]) read as one u16 -> the same two values as 16-bit lanes of a dword
__device__ __forceinline__ uint32_t widen_pair(uint32_t pair) { return __builtin_amdgcn_perm(0u, pair, 0x0c010c00u); }
// CV_DESCALE(sum_t tap_t * w_t, W_BITS - 5) of the taps (p00, p01) = top, (p10, p11) = bot: two
// v_dot2_i32_i16 with the signed weights (so w11 == -1 needs no special case); the sum is >= 1.
__device__ __forceinline__ int interp_pairs(uint32_t top, uint32_t bot, const Weights& w) {
    return sdot2(bot, w.r1, sdot2(top, w.r0, 1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
}
// interp_pairs(top, bot, w) - ival with the subtraction folded into the accumulator:
// bias = 2^(W_BITS-6) - ival * 2^(W_BITS-5), and floor((S - 512 i) / 512) == floor(S / 512) - i.
// The first dot product asks for the clamping form: nothing here can saturate (|S| < 2^23), but that
// form is the three-address VOP3P encoding, which leaves `bias` intact without a copy (the
// two-address v_dot2c the compiler otherwise picks needs a v_mov per pixel).
__device__ __forceinline__ int interp_diff(uint32_t top, uint32_t bot, const Weights& w, int bias) {
    const int t = __builtin_amdgcn_sdot2(__builtin_bit_cast(pc_short2, top), __builtin_bit_cast(pc_short2, w.r0), bias, true);
    return sdot2(bot, w.r1, t) >> (W_BITS - 5);
}
__device__ __forceinline__ int ival_bias(int ival) { return (1 << (W_BITS - 5 - 1)) - (ival << (W_BITS - 5)); }
// acc + (int16)a * (int16)b.lo / b.hi in one instruction (v_mad_i32_i16, op_sel picks the half)
__device__ __forceinline__ int mad16_lo(int a, uint32_t b, int acc) {
    int d;
    asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(acc));
    return d;
}
__device__ __forceinline__ int mad16_hi(int a, uint32_t b, int acc) {
    int d;
    asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[0,1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(acc));
    return d;
}
// Sum over an 8-lane group of per-lane partials with |partial| < 2^29, as ONE rounding of the exact
// integer: quad sums stay below 2^31, the two quads are added exactly in fp64.
__device__ __forceinline__ float group8_exact_sum_small(int partial) {
    int v = partial + dpp_i32<0xB1>(partial);
    v += dpp_i32<0x4E>(v);
    const int other = dpp_i32<0x141>(v);
    return (float)((double)v + (double)other);
}

template <int WIN>
struct LKGeo {
    static constexpr int NPX = WIN * WIN;
    // search margin of the staged J region: 1 px measured fastest (C2 LK launch 0.575 ms; 0 px 0.638, 2 px 0.730,
    // 3 px 0.655): a small region is cheap to stage (13 x 4 dwords) and at most levels the window does not leave it
    static constexpr int MX = PC_LK_MARGIN, MY = PC_LK_MARGIN;
    static constexpr int RW_DW = (WIN + 1 + 2 * MX + 3 + 3) / 4;      // raw dwords per region row
    static constexpr int RWB = RW_DW * 4;                             // positions (bytes) per region row
    static constexpr int RH = WIN + 1 + 2 * MY;                       // region rows
    static constexpr int PAIR_PITCH = RWB * 2;                        // bytes per row in pair format
    static constexpr int J_DW = RH * PAIR_PITCH / 4 + 1;              // per-group J region (odd dword stride)
    static constexpr int I_DW = (WIN + 1) * PAIR_PITCH / 4;           // per-wave I window in pair format
    static constexpr int D_PITCH = WIN + 1;                           // dwords per Scharr window row
    static constexpr int D_DW = (((WIN + 1) * D_PITCH) + 1) & ~1;     // per-wave raw Scharr window (even: the exchange buffer behind it must be 8-byte aligned, see kernels_lk3.hip)
    static constexpr int X_DW = NPX * 2;                              // per-wave exchange: (Ival, Dxy) per pixel
    static constexpr int WAVE_DW = ((I_DW + D_DW + X_DW + 8 * J_DW + 1) / 2) * 2;  // 8-B aligned
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Stage `nrows` rows of an u8 plane, starting at (rx0, ry0) (rx0 4-aligned relative to the interior
// origin), as byte pairs: P[r][c] = (A[r][c], A[r][c+1]) for c in [0, RWB), using NL lanes.
// CLAMP: addresses are clamped to the padded plane; clamped positions are never consumed by a
// window that passed the bounds check.
template <int WIN, int NL, bool CLAMP>
__device__ __forceinline__ void stage_pairs(const uint8_t* __restrict__ img, int pitch, int lh, int rx0, int ry0,
                                            int nrows, uint8_t* buf, int l) {
    using G = LKGeo<WIN>;
    const int total = nrows * G::RW_DW;
    const int xmax = pitch - kPadX - 4;
    int r = l / G::RW_DW, m = l - r * G::RW_DW;
    for (int i = l; i < total; i += NL) {
        int yy = ry0 + r, xb = rx0 + 4 * m, xb1 = xb + 4;
        if (CLAMP) {
            yy = clampi(yy, -WIN, lh + WIN - 1);
            xb = clampi(xb, -kPadX, xmax);
            xb1 = clampi(xb1, -kPadX, xmax);
        }
        const uint8_t* rowp = img + (ptrdiff_t)(yy * pitch);
        const uint32_t d0 = *reinterpret_cast<const uint32_t*>(rowp + xb);
        const uint32_t d1 = *reinterpret_cast<const uint32_t*>(rowp + xb1);
        // v_perm_b32: byte pool = {d0: indices 0-3, d1: indices 4-7}
        const uint32_t p0 = __builtin_amdgcn_perm(d1, d0, 0x02010100u);  // (A0,A1),(A1,A2)
        const uint32_t p1 = __builtin_amdgcn_perm(d1, d0, 0x04030302u);  // (A2,A3),(A3,A4)
        *reinterpret_cast<uint2*>(buf + r * G::PAIR_PITCH + 8 * m) = make_uint2(p0, p1);
        // advance (r, m) by NL elements without a division
        m += NL % G::RW_DW;
        r += NL / G::RW_DW;
        if (m >= G::RW_DW) {
            m -= G::RW_DW;
            r++;
        }
    }
}

// The same staging for a region that lies inside the padded plane.  ALL of a lane's global loads are issued
// before the first one is consumed: the loop used to be "load, wait, permute, store" per trip, i.e. one exposed
// memory latency per region row (13 per J region, 19 per pyramid level and wavefront), and with three
// wavefronts per SIMD nothing covers them -- that serialisation, not instruction issue, was most of a
// wavefront's life time.  Costs 2 * TRIPS VGPRs while the loads are in flight.
// one batch of trips [k0, k0 + B): loads first, then permute + store.  const_k: k0 is a compile-time constant at
// the (inlined) call site, so only a trip that can run past the end clamps its index (a clamped index is a per-lane
// VGPR value; the other trips keep constant row numbers).
template <int WIN, int NL, int TOTAL, int B>
__device__ __forceinline__ void stage_pairs_batch(const uint8_t* base, int pitch, uint8_t* buf, int l, int k0, bool const_k) {
    using G = LKGeo<WIN>;
    uint32_t d0[B], d1[B];
    int r[B], m[B];
#pragma unroll
    for (int b = 0; b < B; b++) {
        // lanes past the end handle the last item once more (same address, same value) instead of branching:
        // a predicated tail makes the compiler sink its load behind the first wait
        int i = l + NL * (k0 + b);
        if (!const_k || NL * (k0 + b + 1) > TOTAL) i = min(i, TOTAL - 1);
        r[b] = i / G::RW_DW;
        m[b] = i - r[b] * G::RW_DW;
        const uint8_t* src = base + (ptrdiff_t)(r[b] * pitch) + 4 * m[b];
        d0[b] = *reinterpret_cast<const uint32_t*>(src);
        d1[b] = *reinterpret_cast<const uint32_t*>(src + 4);
    }
#pragma unroll
    for (int b = 0; b < B; b++) {
        const uint32_t p0 = __builtin_amdgcn_perm(d1[b], d0[b], 0x02010100u);  // (A0,A1),(A1,A2)
        const uint32_t p1 = __builtin_amdgcn_perm(d1[b], d0[b], 0x04030302u);  // (A2,A3),(A3,A4)
        *reinterpret_cast<uint2*>(buf + r[b] * G::PAIR_PITCH + 8 * m[b]) = make_uint2(p0, p1);
    }
}

template <int WIN, int NL, int NROWS>
__device__ __forceinline__ void stage_pairs_inside(const uint8_t* __restrict__ img, int pitch, int rx0, int ry0,
                                                   uint8_t* buf, int l) {
    using G = LKGeo<WIN>;
    constexpr int TOTAL = NROWS * G::RW_DW;
    constexpr int TRIPS = (TOTAL + NL - 1) / NL;
    // When the lanes of a group map onto whole region rows (NL % RW_DW == 0: windows 7..10 with 4-lane groups) the
    // (row, dword) of every trip is a constant plus the lane's, and the trips are unrolled in batches of
    // PC_LK_STAGE_BATCH.  Otherwise the indices are per-lane values: fully unrolled, the compiler hoists one set per
    // trip out of the iteration loops and keeps them in registers for the whole kernel (250 VGPRs at WIN = 11) --
    // those windows walk a real loop of batches.
    constexpr bool REGULAR = (NL % G::RW_DW == 0) || (G::RW_DW % NL == 0);
    const uint8_t* const base = img + (ptrdiff_t)(ry0 * pitch) + rx0;
    if constexpr (REGULAR) {
        constexpr int B = TRIPS <= PC_LK_STAGE_BATCH ? TRIPS : PC_LK_STAGE_BATCH;
        constexpr int FULL = TRIPS / B, REST = TRIPS - FULL * B;
#pragma unroll
        for (int q = 0; q < FULL; q++) {
            stage_pairs_batch<WIN, NL, TOTAL, B>(base, pitch, buf, l, q * B, true);
            if (q + 1 < FULL || REST > 0) __builtin_amdgcn_sched_barrier(0);   // keep the next batch's loads below
        }
        if constexpr (REST > 0) stage_pairs_batch<WIN, NL, TOTAL, REST>(base, pitch, buf, l, FULL * B, true);
    } else {
        constexpr int B = TRIPS < PC_LK_STAGE_BATCH_IRR ? TRIPS : PC_LK_STAGE_BATCH_IRR;
#pragma unroll 1
        for (int k0 = 0; k0 < TRIPS; k0 += B) stage_pairs_batch<WIN, NL, TOTAL, B>(base, pitch, buf, l, k0, false);
    }
}

template <int WIN, int NL, int NROWS>
__device__ __forceinline__ void stage_pairs_auto(const uint8_t* __restrict__ img, int pitch, int lh, int rx0, int ry0,
                                                 uint8_t* buf, int l) {
    using G = LKGeo<WIN>;
    const bool inside = (ry0 >= -WIN) && (ry0 + NROWS <= lh + WIN) && (rx0 >= -kPadX) &&
                        (rx0 + G::RWB + 4 <= pitch - kPadX);
    if (inside) stage_pairs_inside<WIN, NL, NROWS>(img, pitch, rx0, ry0, buf, l);
    else stage_pairs<WIN, NL, true>(img, pitch, lh, rx0, ry0, NROWS, buf, l);
}

// Raw Scharr window ((WIN+1)^2 dwords at `Dbase`, row pitch `pitch` dwords) -> LDS with NL lanes.  load() only
// issues the loads, so the caller can put the I-window staging between load() and store() and pay one memory
// latency for both.
template <int WIN, int NL>
struct DerivWindow {
    static constexpr int TOTAL = (WIN + 1) * (WIN + 1);
    static constexpr int TRIPS = (TOTAL + NL - 1) / NL;
    int32_t v[TRIPS];
    __device__ __forceinline__ void load(const int32_t* __restrict__ Dbase, int pitch, int l) {
#pragma unroll
        for (int k = 0; k < TRIPS; k++) {
            const int i = min(l + NL * k, TOTAL - 1);   // lanes past the end repeat the last item (see stage_pairs_inside)
            const int r = i / (WIN + 1), c = i - r * (WIN + 1);
            v[k] = Dbase[r * pitch + c];
        }
    }
    __device__ __forceinline__ void store(int32_t* dbuf, int l) const {
#pragma unroll
        for (int k = 0; k < TRIPS; k++) {
            dbuf[min(l + NL * k, TOTAL - 1)] = v[k];
        }
    }
};

// wave-wide exact integer sum -> fp32: DPP inside each 16-lane row, then the 4 row results
// (lanes 0,16,32,48) through readlane
__device__ __forceinline__ int wave_sum_i32(int v) {
    v = group_allreduce_add<16>(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ float wave_exact_sum(int partial) {
    return exact_sum_to_float(wave_sum_i32(partial >> 16), wave_sum_i32(partial & 0xffff));
}

}  // namespace pc
