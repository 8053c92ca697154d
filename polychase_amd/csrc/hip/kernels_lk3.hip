// kernels_lk3.hip -- pyramidal Lucas-Kanade (K8-K10), two keypoints per wavefront, on the uint16 planes.
//
// Same arithmetic and results as kernels_lk.hip (bit for bit; both follow oracle/pc_oracle.c,
// which restates cv::calcOpticalFlowPyrLK as called at reference cpp/opticalflow.cc:119-125), same mapping as
// round 1's lk2 kernel -- lanes 0-31 track keypoint A, lanes 32-63 keypoint B, group g = 4 lanes tracks the keypoint into target
// g, the I side is evaluated once per keypoint by its half-wave -- but another data path for the inner loop:
//
//   * The images are read from the uint16 planes (Level::img16, value = pixel << 7).  In LDS a region keeps one
//     DWORD per position: Q[r][c] = (p[r][c] << 7) | (p[r][c+1] << 7) << 16, i.e. the two horizontal taps of a
//     bilinear sample as the 16-bit lanes v_dot2_i32_i16 wants.  One pixel of one iteration is then
//         ds_read_b32 (aligned)                       the row below; the row above is the previous pixel's
//         v_dot2_i32_i16  x 2                         R = 128 * (sum of the 4 weighted taps) + bias
//         v_mad_i32_i16 op_sel:[1,...] x 2            b += hi16(R) * (ix, iy)
//     because hi16(R) = (S + 2^8 - 2^9 * I) >> 9 = CV_DESCALE(S, W_BITS - 5) - I when bias = 2^15 - I * 2^16:
//     the rounding shift is "the high half of the register" and costs nothing.  lk2 needed, per pixel, an
//     unaligned ds_read_u16 (5.5 LDS stall cycles each: SQ_LDS_UNALIGNED_STALL was 62 % of the LDS pipe's busy
//     time, which itself was 65 % of the launch), a v_perm to widen the byte pair and a v_ashr: 6 VALU + a slow
//     LDS read per pixel then, 4 VALU + a 2-cycle LDS read now.
//   * A region is (WIN + 1 + 2) rows x (WIN + 2) positions at its exact origin (no 4-byte alignment slack: the
//     uint16 plane is read with 2-byte aligned dwordx4 loads); 16 regions of a 10 x 10 window are 9.75 KB per
//     wavefront, so the 12 wavefronts of a CU (3 per SIMD at this register count) fit the 160 KB of LDS.
//   * The uint16 planes carry two addressable slack rows above and below the padded image: a region never needs
//     address clamping, there is ONE staging path.
#include <cmath>
#include <limits>

#include "lk_common.hpp"

namespace pc {

// What the inner loop is made of (all of it measured, DESIGN_HISTORY.md "The LK kernel"): the b-vector accumulation takes the
// pixels of two runs at a time -- (hi16 R_a, hi16 R_b) packed by one v_perm_b32, then b1 += R_a * ix_a + R_b * ix_b as ONE
// v_dot2_i32_i16 (and one for b2): 3 instructions per two pixels where the v_mad_i32_i16 form takes 4; the weights are rounded by
// the 1.5 * 2^23 addition and packed with v_perm_b32 (packed_weights), the fractional parts come from the floor values, image /
// region tests are unsigned compares, the oscillation test runs in fp32; LDS reads run one step ahead of the arithmetic.
// The experiments that were measured and NOT taken -- staging twice (the LDS-DMA ceiling), early loads of a level's first region,
// restaging all groups together, staging ahead of the motion, two lanes per row for the I windows, issue priorities, 2 / 4
// wavefronts per SIMD, two wavefronts per workgroup, the unpaired / untrimmed data paths -- are no longer in this file: the
// patch that holds them is tools/lk_variants/r04_experiments.patch (applies to the round-4 kernel, git tag-less: commit 8b4ea23),
// their numbers are in profiles/r03_*_lk_variants.jsonl, r04_*_lk_staging_cost.jsonl, r04_lk_early_staging.jsonl,
// r04_*_lk_restage.jsonl, r04_lk_paired_*.{jsonl,txt}.

// -DPC_LK_PROFILE: per-phase shader-clock sums (pc_debug_lk_profile); costs ~10 % of the launch
#ifdef PC_LK_PROFILE
#define PC_PROF_DECL unsigned long long prof_t = __builtin_readcyclecounter(), prof_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long prof_t0 = prof_t;
#define PC_PROF(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); prof_acc[k] += t_ - prof_t; prof_t = t_; } while (0)
#define PC_PROF_COUNT(k) do { prof_acc[k] += 1; } while (0)
#else
#define PC_PROF_DECL
#define PC_PROF(k) do { } while (0)
#define PC_PROF_COUNT(k) do { } while (0)
#endif

template <int WIN>
struct LK3Geo {
    static constexpr int GL = 4, NPX = WIN * WIN;
    static constexpr int MX = 1, MY = 1;                      // search margin of a staged region
    static constexpr int RWP = WIN + 2 * MX;                  // positions per region row
    static constexpr int CH = (RWP + 3) / 4;                  // 4-position chunks per row
    static constexpr int PITCH = 4 * CH;                      // dwords per region row
    static constexpr int RH = WIN + 1 + 2 * MY;               // region rows
    static constexpr int J_DW = RH * PITCH;                   // per-group J region (16-byte multiple)
    static constexpr int I_CH = (WIN + 3) / 4, I_PITCH = 4 * I_CH, I_ROWS = WIN + 1;
    static constexpr int I_DW = I_ROWS * I_PITCH;             // I window of one keypoint, same format
    // raw Scharr window.  D_DW is kept EVEN: the exchange buffer behind it is read with ds_read_b64, and at an odd dword
    // offset every one of those reads is 8-byte misaligned -- 62 LDS stall cycles each, SQ_LDS_UNALIGNED_STALL = 60 % of
    // the LDS pipe's busy time, which itself was 82 % of the launch (rounds 1 and 2 until this was found)
    static constexpr int D_PITCH = WIN + 1, D_DW = (((WIN + 1) * (WIN + 1)) + 1) & ~1;
    static constexpr int X_DW = 2 * NPX;                      // (bias, Dxy) exchange of one keypoint
    static constexpr int HALF_I_DW = ((I_DW + D_DW + X_DW + 3) / 4) * 4;
    static constexpr int AREA_DW = ((16 * J_DW > 2 * HALF_I_DW ? 16 * J_DW : 2 * HALF_I_DW) + 1) & ~1;   // regions / I-side windows
    static constexpr int WAVE_DW = AREA_DW;
    // window pixels of a lane: NCH column chains (columns lg + 4c) + a run of the remaining columns
    static constexpr int WM = (WIN / GL) * GL, NCH = WM / GL, KM = NCH * WIN;
    static constexpr int NEXTRA = (WIN - WM) * WIN, KE = (NEXTRA + GL - 1) / GL, K = KM + KE;
    // the runs e in [lg * KE, (lg + 1) * KE) of the column-major extra pixels never continue into the next column
    static constexpr bool RUNS = KE > 0 && ((WIN - WM) == 1 || WIN % (KE > 0 ? KE : 1) == 0);
};

// one region row in flight: 2 * CHN + 1 dwords of uint16 pixels -> 4 * CHN position dwords
template <int CHN>
struct RowRegs {
    struct __attribute__((packed, aligned(2))) Raw { uint32_t d[2 * CHN + 1]; };
    Raw v;
    __device__ __forceinline__ void load(const uint16_t* __restrict__ src) { v = *reinterpret_cast<const Raw*>(src); }
    __device__ __forceinline__ void store(uint32_t* dst) const {
#pragma unroll
        for (int c = 0; c < CHN; c++) {
            const uint32_t a = v.d[2 * c], b = v.d[2 * c + 1], e = v.d[2 * c + 2];
            // ds_write2_b32 stores two ARBITRARY registers to two dwords; the 16-byte store the compiler forms wants
            // four consecutive registers and pays a v_mov for each loaded dword (LDS operations of a wavefront
            // complete in order, so the compiler's lgkmcnt waits stay sufficient with these in the queue)
            const uint32_t p1 = __builtin_amdgcn_alignbit(b, a, 16), p3 = __builtin_amdgcn_alignbit(e, b, 16);
            const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)(dst);
            asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" : : "v"(addr), "v"(a), "v"(p1), "n"(4 * c), "n"(4 * c + 1) : "memory");
            asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" : : "v"(addr), "v"(b), "v"(p3), "n"(4 * c + 2), "n"(4 * c + 3) : "memory");
        }
    }
};

// R = 128 * (sum of the 4 weighted taps) + bias; hi16(R) is the CV_DESCALEd sample minus the I value (see the header)
__device__ __forceinline__ int interp_r(uint32_t top, uint32_t bot, const Weights& w, int bias) {
    const int t = __builtin_amdgcn_sdot2(__builtin_bit_cast(pc_short2, top), __builtin_bit_cast(pc_short2, w.r0), bias, true);
    return sdot2(bot, w.r1, t);
}
// bilinear_weights(a, b).r0 / .r1 in 14 instructions instead of 25.  cvRound(x) for 0 <= x < 2^22 is the low mantissa
// bits of fl(x + 1.5 * 2^23) (the addition rounds to nearest even at unit granularity); scaling one factor by 2^14 before
// the product instead of the product after it is exact, so the products round identically; the 16-bit halves are picked
// out of the sums' bit patterns with v_perm_b32 and w11 = 2^14 - w00 - w01 - w10 is formed on those patterns.
__device__ __forceinline__ Weights packed_weights(float a, float b) {
    constexpr float S = (float)(1 << W_BITS), M = 12582912.f;
    constexpr uint32_t MB = 0x4B400000u;   // bits of M
    const float na = 1.f - a, nbs = (1.f - b) * S, bs = b * S;
    const uint32_t t00 = __float_as_uint(na * nbs + M), t01 = __float_as_uint(a * nbs + M), t10 = __float_as_uint(na * bs + M);
    const uint32_t w11 = ((1u << W_BITS) + 3u * MB) - (t00 + t01 + t10);
    Weights w;
    w.w00 = w.w01 = w.w10 = w.w11 = 0;
    w.neg11 = false;
    w.r0 = __builtin_amdgcn_perm(t01, t00, 0x05040100u);
    w.r1 = __builtin_amdgcn_perm(w11, t10, 0x05040100u);
    return w;
}
__device__ __forceinline__ int bias_of(int ival) { return (1 << 15) - (ival << 16); }
// acc + hi16(a) * (int16)b.lo / b.hi
__device__ __forceinline__ int mad16_hl(int a, uint32_t b, int acc) {
    int d;
    asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(acc));
    return d;
}
__device__ __forceinline__ int mad16_hh(int a, uint32_t b, int acc) {
    int d;
    asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(acc));
    return d;
}
// hi16(a) * (int16)b.lo / b.hi: the same with a literal zero accumulator (no register for it)
__device__ __forceinline__ int mul16_hl(int a, uint32_t b) {
    int d;
    asm("v_mad_i32_i16 %0, %1, %2, 0 op_sel:[1,0,0,0]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ int mul16_hh(int a, uint32_t b) {
    int d;
    asm("v_mad_i32_i16 %0, %1, %2, 0 op_sel:[1,1,0,0]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

__device__ __forceinline__ int half_sum3_i32(int v) {
    v = group_allreduce_add<16>(v);
    return v + __shfl_xor(v, 16);
}
__device__ __forceinline__ float half_exact_sum3(int partial) {
    return exact_sum_to_float(half_sum3_i32(partial >> 16), half_sum3_i32(partial & 0xffff));
}
// the same when the total is known to fit int32 (NPX * 4080^2 < 2^31): one reduction, one conversion -- one rounding
__device__ __forceinline__ float half_exact_sum3_small(int partial) { return (float)half_sum3_i32(partial); }
// v_dot2_i32_i16 with a zero accumulator (the compiler's form is a v_mov 0 and the accumulating VOP2 encoding)
__device__ __forceinline__ int sdot2_zero(uint32_t a, uint32_t b) {
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// sum over a 4-lane group as ONE rounding of the exact integer
template <int K>
__device__ __forceinline__ float group4_exact_sum3(int partial) {
    if constexpr ((long long)K * 8160 * 4080 < (1ll << 30)) {   // pair sums stay below 2^31: add the pairs in fp64
        const int v = partial + dpp_i32<0xB1>(partial);
        const int other = dpp_i32<0x4E>(v);
        return (float)((double)v + (double)other);
    } else {
        int hi = partial >> 16, lo = partial & 0xffff;
        hi += dpp_i32<0xB1>(hi);
        hi += dpp_i32<0x4E>(hi);
        lo += dpp_i32<0xB1>(lo);
        lo += dpp_i32<0x4E>(lo);
        return exact_sum_to_float(hi, lo);
    }
}

// The J region of one group, staged with TWO LANES PER ROW.  Giving every lane whole rows (rounds 2-4) means each of a lane's
// 8 loads (10-px window) sends 64 lanes to 64 different rows of 16 different images -- 76 L1 accesses per instruction
// (TCP_TOTAL_CACHE_ACCESSES / SQ_INSTS_VMEM_RD, profiles/r04_lk_tcp_counters.txt), 608 per staging, and the texture
// addresser of the CU, which takes one access per cycle, is busy or stalled 68 % of the launch: the twelve wavefronts of a
// CU queue up behind each other's stagings.  Here lanes (0, 1) of a group read the two 16-byte halves of row 2k, lanes
// (2, 3) those of row 2k + 1: one dwordx4 per lane and row pair -- 7 instead of 8 loads, and the two lanes of a row share
// their cache line(s).  Position 8h + 7 needs the first pixel of the other half: one DPP move.  Positions past the region's
// pitch are not written (lane h = 1: the upper four for a 12-position pitch, all eight for an 8-position one).
// The loads read uint16 columns rx0 .. rx0 + 15, at most 16 - (WIN + 3) past the row's last needed one: the next row of the
// plane, or the tail slack behind the frame's last plane (pc_frame_create).
template <int WIN>
__device__ __forceinline__ void stage_region_paired(const uint16_t* __restrict__ J16, int pitch, int rx0, int ry0, uint32_t* jbuf, int lg) {
    using G = LK3Geo<WIN>;
    static_assert(G::RWP + 1 <= 16 && G::PITCH >= 8 && G::PITCH <= 16, "a region row is at most 16 pixels: two dwordx4");
    // The row offsets derived from lg are invariants of the iteration loop this is called from; hoisted out of it they sit in
    // registers across the kernel's most register-hungry stretch.  Staging is rare: recompute them here (the empty asm hides
    // the invariance from the optimiser).
    asm volatile("" : "+v"(lg));
    constexpr int NI = (G::RH + 1) / 2;
    constexpr bool ODD = (G::RH & 1) != 0;   // the last load of lanes (2, 3) repeats row RH - 1
    static_assert((2 * (NI - 1) + 1) * G::PITCH + 7 < 256, "ds_write2_b32 offsets are 8 bits of dwords");
    struct __attribute__((packed, aligned(2))) Raw { uint32_t d[4]; };
    const int s = lg >> 1, h = lg & 1;
    const uint16_t* src = J16 + (ptrdiff_t)(__mul24(ry0 + s, pitch) + rx0 + 8 * h);
    const int last_step = ODD ? (s ? pitch : 2 * pitch) : 2 * pitch;
    Raw v[NI];
#pragma unroll
    for (int k = 0; k < NI; k++) {
        v[k] = *reinterpret_cast<const Raw*>(src);
        if (k + 1 < NI) src += (k + 2 == NI) ? last_step : 2 * pitch;
    }
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)(jbuf + s * G::PITCH + 8 * h);
    // the repeated row of an odd region: lanes (2, 3) write row RH - 1 again (the same values as lanes (0, 1))
    const uint32_t addr_last = ODD ? (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)(jbuf + (G::RH - 1) * G::PITCH + 8 * h) : 0u;
    const bool lo_on = G::PITCH >= 12 || h == 0, hi_on = G::PITCH >= 16 || h == 0;
    if (lo_on) {
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const uint32_t d0 = v[k].d[0], d1 = v[k].d[1], d2 = v[k].d[2];
            const uint32_t p1 = __builtin_amdgcn_alignbit(d1, d0, 16), p3 = __builtin_amdgcn_alignbit(d2, d1, 16);
            if (ODD && k == NI - 1) {
                asm volatile("ds_write2_b32 %0, %1, %2 offset0:0 offset1:1" : : "v"(addr_last), "v"(d0), "v"(p1) : "memory");
                asm volatile("ds_write2_b32 %0, %1, %2 offset0:2 offset1:3" : : "v"(addr_last), "v"(d1), "v"(p3) : "memory");
            } else {
                asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" : : "v"(addr), "v"(d0), "v"(p1), "n"(2 * k * G::PITCH), "n"(2 * k * G::PITCH + 1) : "memory");
                asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" : : "v"(addr), "v"(d1), "v"(p3), "n"(2 * k * G::PITCH + 2), "n"(2 * k * G::PITCH + 3) : "memory");
            }
        }
    }
    // (the DPP moves read the other lane's register whether or not that lane writes: they stay outside the masked block)
    uint32_t nx[NI];
#pragma unroll
    for (int k = 0; k < NI; k++) nx[k] = (uint32_t)dpp_i32<0xF5>((int)v[k].d[0]);   // quad_perm [1, 1, 3, 3]
    if (hi_on) {
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const uint32_t d2 = v[k].d[2], d3 = v[k].d[3];
            const uint32_t p5 = __builtin_amdgcn_alignbit(d3, d2, 16), p7 = __builtin_amdgcn_alignbit(nx[k], d3, 16);
            if (ODD && k == NI - 1) {
                asm volatile("ds_write2_b32 %0, %1, %2 offset0:4 offset1:5" : : "v"(addr_last), "v"(d2), "v"(p5) : "memory");
                asm volatile("ds_write2_b32 %0, %1, %2 offset0:6 offset1:7" : : "v"(addr_last), "v"(d3), "v"(p7) : "memory");
            } else {
                asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" : : "v"(addr), "v"(d2), "v"(p5), "n"(2 * k * G::PITCH + 4), "n"(2 * k * G::PITCH + 5) : "memory");
                asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" : : "v"(addr), "v"(d3), "v"(p7), "n"(2 * k * G::PITCH + 6), "n"(2 * k * G::PITCH + 7) : "memory");
            }
        }
    }
}
// ------------------------------------------------------------------------------------------------------------------
// X86 = true (PC_ARITH_LK_X86_ORDER): the sums of the structure tensor and of the mismatch vector as an x86 OpenCV build
// forms them -- LKTrackerInvoker's CV_SIMD128 path: over the first SIMD_W = (WIN / 8) * 8 columns vector lane j = x & 3
// accumulates in fp32, row by row (the mismatch vector: the int32 pair sum of columns (c, c + 4), converted, one term per
// row), a scalar fp32 accumulator takes the remaining columns in row-major order, combined at the end -- bit for bit
// oracle/pc_oracle.c under PCO_EMU_LK_SIMD.
//
// Every term of those sums is an integer.  While the sum of the MAGNITUDES of a sum's terms stays <= 2^24, every partial
// sum in ANY order is an integer <= 2^24 in magnitude, i.e. exact in fp32: the x86 order then gives the float of the
// exact integer total -- the canonical value.  So the kernel runs the canonical integer data path and PROVES, per level
// for the structure tensor (S11, S22 < 2^24; |ix iy| <= (ix^2 + iy^2) / 2) and per iteration for the mismatch vector
// (Cauchy-Schwarz: sum |d ix| <= sqrt(sum d^2 * S11); sum d^2 costs one v_dot2 per pixel pair), that the bound holds; only
// where the proof fails does it evaluate the sums in the x86 order (x86_structure_tensor / x86_mismatch_ordered below).
// The lane mapping is already the right one for that: lane lg of a group owns columns lg and lg + 4 = vector lane x & 3,
// and the int32 pair sum of columns (c, c + 4) is one v_dot2_i32_i16.
template <int WIN>
struct X86Geo {
    using G = LK3Geo<WIN>;
    static constexpr int SIMD_W = (WIN / 8) * 8, NXS = WIN - SIMD_W, NS = NXS * WIN;   // vector columns, scalar columns, scalar steps
    static constexpr int CL = WIN * (SIMD_W / 4);                                       // terms of a vector lane's chain (structure tensor)
    static constexpr int KSC0 = SIMD_W == 0 ? G::KM : 0;   // WIN < 8: the canonical column chain belongs to the scalar chain
    static constexpr int NSL = KSC0 + G::KE;               // scalar-chain slots of a lane
    static constexpr int PA_DW = 3 * G::NPX + 16;          // LDS of the ordered structure tensor, per half
    static_assert(2 * G::HALF_I_DW + 2 * PA_DW <= G::AREA_DW, "the ordered structure tensor's products fit behind the I-side buffers");
    static_assert(SIMD_W == 0 || (SIMD_W == 8 && G::WM == 8 && G::NCH == 2), "vector lanes = the two column chains");
    // scalar step s (row-major over the scalar columns) -> the lane that owns the pixel and its index among that lane's slots
    static constexpr int sx(int s) { return SIMD_W + s % (NXS > 0 ? NXS : 1); }
    static constexpr int sy(int s) { return s / (NXS > 0 ? NXS : 1); }
    static constexpr bool in_chain0(int s) { return SIMD_W == 0 && sx(s) < G::WM; }
    static constexpr int se(int s) { return (sx(s) - G::WM) * WIN + sy(s); }            // column-major index among the extra pixels
    static constexpr int lane_of(int s) { return in_chain0(s) ? sx(s) : se(s) / (G::KE > 0 ? G::KE : 1); }
    static constexpr int slot_of(int s) { return in_chain0(s) ? sy(s) : KSC0 + se(s) % (G::KE > 0 ? G::KE : 1); }
};

__device__ __forceinline__ float dpp_bcast4(float v, int src_lane) {   // the value of lane `src_lane` of the 4-lane group
    const int i = __float_as_int(v);
    switch (src_lane) {
        case 0: return __int_as_float(dpp_i32<0x00>(i));
        case 1: return __int_as_float(dpp_i32<0x55>(i));
        case 2: return __int_as_float(dpp_i32<0xAA>(i));
        default: return __int_as_float(dpp_i32<0xFF>(i));
    }
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) { return __int_as_float(dpp_i32<CTRL>(__float_as_int(v))); }
// acc + (the value of lane `src_lane` of the 4-lane group) as ONE v_add_f32 with a DPP operand, volatile: the steps of a
// sequential chain stay in place (left to the scheduler, the 2 x NS broadcasts are all formed first -- 40 registers)
__device__ __forceinline__ float dpp_add4(float acc, float v, int src_lane) {
    switch (src_lane) {
        case 0: asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v)); break;
        case 1: asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v)); break;
        case 2: asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v)); break;
        default: asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v)); break;
    }
    return acc;
}

// acc1 + d * ix, acc2 + d * iy for the pixel in slot k (d = the high half of R), the derivatives picked out of the paired
// register layout of the iteration loop by the multiply-add's op_sel: no v_perm
template <int WIN, int K, bool FIRST>
__device__ __forceinline__ void x86_mad_dxy(int R, const int (&Dxy)[K], int k, int& acc1, int& acc2) {
    using G = LK3Geo<WIN>;
    constexpr int H1 = (WIN + 1) / 2, H2 = WIN - H1;
    // FIRST: the accumulators are not read (a literal zero in the instruction)
    auto hl = [](int a, uint32_t b, int acc) { return FIRST ? mul16_hl(a, b) : mad16_hl(a, b, acc); };
    auto hh = [](int a, uint32_t b, int acc) { return FIRST ? mul16_hh(a, b) : mad16_hh(a, b, acc); };
    if (k < G::KM) {
        const int c = k / WIN, r = k - c * WIN;
        if (r < H2) {
            acc1 = hl(R, (uint32_t)Dxy[c * WIN + r], acc1);
            acc2 = hl(R, (uint32_t)Dxy[c * WIN + H1 + r], acc2);
        } else if (r >= H1) {
            acc1 = hh(R, (uint32_t)Dxy[c * WIN + r - H1], acc1);
            acc2 = hh(R, (uint32_t)Dxy[c * WIN + r], acc2);
        } else {   // an odd window's middle row: (ix, iy) in one register
            acc1 = hl(R, (uint32_t)Dxy[k], acc1);
            acc2 = hh(R, (uint32_t)Dxy[k], acc2);
        }
        return;
    }
    const int e = k - G::KM;
    if (G::KE > 1 && G::RUNS) {
        if ((e & 1) == 0 && e + 1 < G::KE) {
            acc1 = hl(R, (uint32_t)Dxy[k], acc1);
            acc2 = hl(R, (uint32_t)Dxy[k + 1], acc2);
            return;
        }
        if (e & 1) {
            acc1 = hh(R, (uint32_t)Dxy[k - 1], acc1);
            acc2 = hh(R, (uint32_t)Dxy[k], acc2);
            return;
        }
    }
    acc1 = hl(R, (uint32_t)Dxy[k], acc1);
    acc2 = hh(R, (uint32_t)Dxy[k], acc2);
}

// The mismatch vector of one iteration in the x86 order (see above).  Returns the raw sums (b * 2^20) in all lanes of the group.
template <int WIN, int K, int KEA>
__device__ __forceinline__ void x86_mismatch_ordered(const uint32_t* jq, int lg, const Weights& wJ, const int (&Bias)[K], const int (&Dxy)[K],
                                                     int e_off, const int (&offE)[KEA], float& b1, float& b2) {
    using G = LK3Geo<WIN>;
    using X = X86Geo<WIN>;
    float q1 = 0.f, q2 = 0.f;   // this vector lane's accumulators
    if constexpr (X::SIMD_W == 8) {
        const uint32_t* cb0 = jq + lg;
        const uint32_t* cb1 = jq + lg + G::GL;
        // q += (float)(int32 pair sum of columns c and c + 4), one term per row: two multiply-adds per quantity (op_sel picks
        // the derivative's half), one conversion, one addition.  Rows go in pairs between scheduling barriers (the second
        // row's multiply-adds fill the wait states of the first's) with the LDS reads of the next pair issued ahead of the
        // arithmetic; the barriers keep the pairs in place: left to itself the scheduler forms all pair sums first and
        // holds them in registers the kernel does not have.
        uint32_t rowA[2] = {cb0[0], cb1[0]}, rowB[2] = {cb0[G::PITCH], cb1[G::PITCH]}, rowC[2] = {0u, 0u};
        if (WIN > 1) {
            rowC[0] = cb0[2 * G::PITCH];
            rowC[1] = cb1[2 * G::PITCH];
        }
#pragma unroll
        for (int r = 0; r < WIN; r += 2) {
            const bool two = r + 1 < WIN;
            uint32_t nxtB[2] = {0u, 0u}, nxtC[2] = {0u, 0u};   // rows r + 3 and r + 4 of the region: the bottoms of the next pair
            if (r + 2 < WIN) {
                nxtB[0] = cb0[(r + 3) * G::PITCH];
                nxtB[1] = cb1[(r + 3) * G::PITCH];
                if (r + 3 < WIN) {
                    nxtC[0] = cb0[(r + 4) * G::PITCH];
                    nxtC[1] = cb1[(r + 4) * G::PITCH];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const int R0 = interp_r(rowA[0], rowB[0], wJ, Bias[r]), R1 = interp_r(rowA[1], rowB[1], wJ, Bias[WIN + r]);
            int S0 = 0, S1 = 0;
            if (two) {
                S0 = interp_r(rowB[0], rowC[0], wJ, Bias[r + 1]);
                S1 = interp_r(rowB[1], rowC[1], wJ, Bias[WIN + r + 1]);
            }
            int t1, t2, u1 = 0, u2 = 0;
            x86_mad_dxy<WIN, K, true>(R0, Dxy, r, t1, t2);
            if (two) x86_mad_dxy<WIN, K, true>(S0, Dxy, r + 1, u1, u2);
            x86_mad_dxy<WIN, K, false>(R1, Dxy, WIN + r, t1, t2);
            if (two) x86_mad_dxy<WIN, K, false>(S1, Dxy, WIN + r + 1, u1, u2);
            q1 += (float)t1;
            q2 += (float)t2;
            if (two) {
                q1 += (float)u1;
                q2 += (float)u2;
            }
            rowA[0] = rowC[0];
            rowA[1] = rowC[1];
            rowB[0] = nxtB[0];
            rowB[1] = nxtB[1];
            rowC[0] = nxtC[0];
            rowC[1] = nxtC[1];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the scalar chain: every lane converts the products of the pixels it owns, then the terms are added in row-major
    // order -- in all four lanes alike, each term broadcast from its owner (v_add_f32 with a DPP operand)
    float f1[X::NSL > 0 ? X::NSL : 1], f2[X::NSL > 0 ? X::NSL : 1];
    if constexpr (X::KSC0 > 0) {
        const uint32_t* cb = jq + lg;
        uint32_t top = cb[0];
#pragma unroll
        for (int r = 0; r < WIN; r++) {
            const uint32_t bot = cb[(r + 1) * G::PITCH];
            const int R = interp_r(top, bot, wJ, Bias[r]);
            top = bot;
            int p1, p2;
            x86_mad_dxy<WIN, K, true>(R, Dxy, r, p1, p2);
            f1[r] = (float)p1;
            f2[r] = (float)p2;
        }
    }
    if constexpr (G::KE > 0) {
        if constexpr (G::RUNS) {
            const uint32_t* cb = jq + e_off;
            uint32_t top = cb[0];
#pragma unroll
            for (int e = 0; e < G::KE; e++) {
                const uint32_t bot = cb[(e + 1) * G::PITCH];
                const int R = interp_r(top, bot, wJ, Bias[G::KM + e]);
                top = bot;
                int p1, p2;
                x86_mad_dxy<WIN, K, true>(R, Dxy, G::KM + e, p1, p2);
                f1[X::KSC0 + e] = (float)p1;
                f2[X::KSC0 + e] = (float)p2;
            }
        } else {
#pragma unroll
            for (int e = 0; e < G::KE; e++) {
                const uint32_t* q = jq + offE[e];
                const int R = interp_r(q[0], q[G::PITCH], wJ, Bias[G::KM + e]);
                const uint32_t u = (uint32_t)Dxy[G::KM + e];
                f1[X::KSC0 + e] = (float)mad16_hl(R, u, 0);
                f2[X::KSC0 + e] = (float)mad16_hh(R, u, 0);
            }
        }
    }
    float s1 = 0.f, s2 = 0.f;
    asm volatile("s_nop 1" ::: "memory");   // a DPP read of a register a VALU instruction has just written needs two wait states
#pragma unroll
    for (int s = 0; s < X::NS; s++) {
        s1 = dpp_add4(s1, f1[X::slot_of(s)], X::lane_of(s));
        s2 = dpp_add4(s2, f2[X::slot_of(s)], X::lane_of(s));
    }
    if constexpr (X::SIMD_W == 8) {
        // bbuf[k] = qb0[k] + qb1[k]: lanes (0, 2) and (1, 3); fb += bbuf[0] + bbuf[2]
        const float t1 = q1 + dpp_f32<0x4E>(q1), t2 = q2 + dpp_f32<0x4E>(q2);
        const float u1 = t1 + dpp_f32<0xB1>(t1), u2 = t2 + dpp_f32<0xB1>(t2);
        // lanes 1 and 3 hold (q1 + q3) + (q0 + q2): the same bits (one commutative addition of the same two values)
        b1 = s1 + u1;
        b2 = s2 + u2;
    } else {
        b1 = s1;
        b2 = s2;
    }
}

// The structure tensor of both keypoints of the wavefront in the x86 order: each lane writes the three products of the
// pixels it evaluated into LDS at their position in their chain; lane 5 k + c of a half adds chain c (0-3: vector lanes, 4: the
// scalar accumulator) of quantity k sequentially; A = scalar + (((q0 + q1) + q2) + q3).  Called by the whole wavefront.
template <int WIN>
__device__ __forceinline__ void x86_structure_tensor(uint32_t* wbase, const uint32_t* xbuf_o, int lane_o, bool i_in, float& S11, float& S12, float& S22) {
    using G = LK3Geo<WIN>;
    using X = X86Geo<WIN>;
    constexpr int NPX = G::NPX, KW = (NPX + 31) / 32;
    constexpr int MAXL = X::CL > X::NS ? X::CL : X::NS;
    const int l32 = lane_o & 31;
    float* const pa = reinterpret_cast<float*>(wbase + 2 * G::HALF_I_DW) + (lane_o >> 5) * X::PA_DW;
    if (i_in) {
#pragma unroll
        for (int m = 0; m < KW; m++) {
            const int q = l32 + 32 * m;
            if (q < NPX) {
                const int y = q / WIN, x = q - y * WIN;
                const uint32_t d = xbuf_o[2 * q + 1];   // written by this lane
                const int ix = (int)(int16_t)(d & 0xffffu), iy = (int)d >> 16;
                const int off = x < X::SIMD_W ? (x & 3) * X::CL + y * (X::SIMD_W / 4) + (x >> 2) : 4 * X::CL + y * X::NXS + (x - X::SIMD_W);
                pa[off] = (float)__mul24(ix, ix);              // |products| <= 4080^2 < 2^24: exact
                pa[NPX + off] = (float)__mul24(ix, iy);
                pa[2 * NPX + off] = (float)__mul24(iy, iy);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (i_in && l32 < 15) {
        const int k = l32 / 5, c = l32 - 5 * k;
        const int len = c < 4 ? X::CL : X::NS;
        const float* src = pa + k * NPX + c * X::CL;
        float acc = 0.f;
        // all LDS reads of the chain in flight, then the additions (unrolled by four the loop sat out an LDS latency per batch:
        // ~0.7 iteration-equivalents per ordered tensor); the 11-px window has no registers for that (it would drop to two
        // wavefronts per SIMD)
        constexpr int kUnroll = WIN <= 10 ? MAXL : 4;
#pragma unroll kUnroll
        for (int i = 0; i < MAXL; i++) {
            const float v = src[i < len ? i : 0];
            acc = i < len ? v + acc : acc;
        }
        pa[3 * NPX + l32] = acc;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (i_in) {
        const float* r = pa + 3 * NPX;
        S11 = r[4] + (((r[0] + r[1]) + r[2]) + r[3]);
        S12 = r[9] + (((r[5] + r[6]) + r[7]) + r[8]);
        S22 = r[14] + (((r[10] + r[11]) + r[12]) + r[13]);
    }
}

template <int WIN, bool X86>
__global__ __launch_bounds__(64) void lk3_kernel(const LKParams p) {
    using G = LK3Geo<WIN>;
    constexpr int GL = G::GL, NPX = G::NPX, NCH = G::NCH, KM = G::KM, KE = G::KE, K = G::K;
    constexpr int KW = (NPX + 31) / 32;   // pixels per lane in the half-wave I-side pass
    // + slack: slots past a lane's run of extra pixels read up to KE rows below the last region (and contribute 0)
    constexpr int WAVE_DW = G::WAVE_DW + (KE > 0 ? (KE + 1) * G::PITCH : 0);
    __shared__ __attribute__((aligned(16))) uint32_t s_buf[1][WAVE_DW];   // one wavefront per workgroup

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, l32 = lane & 31, grp = (lane >> 2) & 7, lg = lane & 3;
    // Workgroup b runs on XCD b % 8; each XCD takes one contiguous eighth of the (spatially binned) keypoint order
    lk_signal_dispatched(p);
    // Occupancy cap.  121 registers would let four wavefronts share a SIMD (15 per CU, then LDS-bound): 3 % faster alone,
    // but those 15 hold 153 of the CU's 160 KB of LDS and the frame-preparation kernels (16 KB tiles) starve beside them --
    // at 4K the min-eig kernel took 0.83 ms instead of 0.33 and the pipeline dropped from 640 to 557 fps.  A wavefront that
    // ALLOCATES 136 registers leaves room for three per SIMD, and with them 104 VGPRs per lane and 37 KB of LDS per CU to
    // whatever runs beside LK.  The clobber below is the whole mechanism.
    asm volatile("; occupancy cap" ::: "v135");
    // (dealing the order to the XCDs in tile-sized chunks instead of contiguous eighths was tried: 2-6 % slower, the
    // launch's tail is not an imbalance between XCDs)
    const int lb = (int)(blockIdx.x & 7u) * p.blocks_per_xcd + (int)(blockIdx.x >> 3);
    const int first = (lb + wave) * 2;  // first of this wave's two keypoint slots
    if ((int)(blockIdx.x >> 3) >= p.blocks_per_xcd || first >= p.n) return;   // whole waves exit together
    const int slot = first + half;
    const bool kp_valid = slot < p.n;                 // n odd: the last wave's second half idles
    const int slot_c = kp_valid ? slot : first;
    const int feat = p.perm ? (int)p.perm[slot_c] : slot_c;
    const bool tgt_active = kp_valid && grp < p.n_targets;
    const int tgt = grp < p.n_targets ? grp : 0;

    uint32_t* const wbase = &s_buf[wave][0];
    uint32_t* const ibuf = wbase + half * G::HALF_I_DW;                          // I window, position dwords
    int32_t* const dbuf = reinterpret_cast<int32_t*>(ibuf + G::I_DW);            // raw Scharr window
    uint32_t* const xbuf = ibuf + G::I_DW + G::D_DW;                             // (bias, Dxy) exchange
    uint32_t* const jbuf = wbase + (half * 8 + grp) * G::J_DW;                   // aliases the above

    // the lane's run of the columns that do not fill a chain (column-major order of those pixels)
    int e_off = 0, e_len = 0, e_q0 = 0;          // RUNS: first top position (dwords from the window origin), length, pixel index
    int offE[KE > 0 ? KE : 1], qE[KE > 0 ? KE : 1];
    if constexpr (KE > 0) {
        if constexpr (G::RUNS) {
            const int e0 = lg * KE;
            e_len = max(0, min(KE, G::NEXTRA - e0));
            const int col = G::WM + (e_len > 0 ? e0 / WIN : 0), row0 = e_len > 0 ? e0 % WIN : 0;
            e_off = row0 * G::PITCH + col;
            e_q0 = row0 * WIN + col;
        } else {
#pragma unroll
            for (int e = 0; e < KE; e++) {
                const int r = lg * KE + e;
                const int col = G::WM + r / WIN, row = r - (r / WIN) * WIN;
                const bool ok = r < G::NEXTRA;
                offE[e] = ok ? row * G::PITCH + col : 0;   // slots past the window read pixel 0, contribute 0
                qE[e] = ok ? row * WIN + col : -1;
            }
        }
    }

    const float2 pt = p.pts[feat];
    const float half_win = (float)(WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    float nx = 0.f, ny = 0.f;
    bool status = true;
    float err = 0.f;
    PC_PROF_DECL
    // X86 diagnostics (LKParams::x86_stats): levels (low half) / levels with the ordered structure tensor (high half);
    // iterations decided by the exactness proof / evaluated in the x86 order
    int x86_levels = 0, x86_iters = 0;
    (void)x86_levels;
    (void)x86_iters;

    for (int level = p.max_level; level >= 0; --level) {
        const Level L = p.src[level];
        const uint16_t* __restrict__ J16 = p.tgt16[tgt][level];
        const int pitch = L.pitch;
        const float lscale = __uint_as_float((uint32_t)(127 - level) << 23);   // 2^-level, the value of 1.f / (1 << level)
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

        // ---- I side: identical for all targets of a keypoint -> computed once by its half-wave ----
        px -= half_win;
        py -= half_win;
        const int ipx = (int)floorf(px), ipy = (int)floorf(py);
        const bool i_in = !(ipx < -WIN || ipx >= L.w || ipy < -WIN || ipy >= L.h);   // uniform per half
        if (!i_in && level == 0) {
            status = false;
            err = 0.f;
        }
        const Weights wI = packed_weights(px - (float)ipx, py - (float)ipy);
        const uint32_t wrow0 = wI.r0, wrow1 = wI.r1;

        // Lane-derived LDS addresses of the I side and the pick-up are invariants of this level loop; hoisted out of it
        // they would occupy ~20 VGPRs across the iteration loop below, the kernel's most register-hungry stretch.  An
        // opaque copy of the lane index makes them per-level values that die before that loop (a handful of extra
        // integer instructions per level).
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int l32_o = lane_o & 31, lg_o = lane_o & 3;
        uint32_t* const ibuf_o = wbase + (lane_o >> 5) * G::HALF_I_DW;
        int32_t* const dbuf_o = reinterpret_cast<int32_t*>(ibuf_o + G::I_DW);
        uint32_t* const xbuf_o = ibuf_o + G::I_DW + G::D_DW;
        int e_q0_o = 0;
        if constexpr (KE > 0 && G::RUNS) {
            const int e0 = lg_o * KE;
            const int len = max(0, min(KE, G::NEXTRA - e0));
            e_q0_o = (len > 0 ? e0 % WIN : 0) * WIN + G::WM + (len > 0 ? e0 / WIN : 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the previous level's J regions are dead
        if (i_in) {
            DerivWindow<WIN, 32> dw;
            dw.load(L.der + (ptrdiff_t)(__mul24(ipy, pitch) + ipx), pitch, l32_o);
            // I window: lane r < WIN + 1 stages row r (the other lanes repeat the last row)
            RowRegs<G::I_CH> row;
            const int r = min(l32_o, G::I_ROWS - 1);
            row.load(L.img16 + (ptrdiff_t)__mul24(ipy + r, pitch) + ipx);
            row.store(ibuf_o + r * G::I_PITCH);
            dw.store(dbuf_o, l32_o);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        PC_PROF(0);
        int sA11 = 0, sA12 = 0, sA22 = 0;
        if (i_in) {
#pragma unroll
            for (int m = 0; m < KW; m++) {
                const int q = l32_o + 32 * m;
                if (q < NPX) {
                    const int y = q / WIN, x = q - y * WIN;
                    const uint32_t* qp = ibuf_o + y * G::I_PITCH + x;
                    const int ival = interp_r(qp[0], qp[G::I_PITCH], wI, 1 << 15) >> 16;
                    const uint32_t* d = reinterpret_cast<const uint32_t*>(dbuf_o) + y * G::D_PITCH + x;
                    const uint32_t d00 = d[0], d01 = d[1], d10 = d[G::D_PITCH], d11 = d[G::D_PITCH + 1];
                    // (dx00, dx01), (dx10, dx11), (dy00, dy01), (dy10, dy11)
                    const uint32_t dx0 = __builtin_amdgcn_perm(d01, d00, 0x05040100u);
                    const uint32_t dx1 = __builtin_amdgcn_perm(d11, d10, 0x05040100u);
                    const uint32_t dy0 = __builtin_amdgcn_perm(d01, d00, 0x07060302u);
                    const uint32_t dy1 = __builtin_amdgcn_perm(d11, d10, 0x07060302u);
                    const int ix = sdot2(dx1, wrow1, sdot2(dx0, wrow0, 1 << (W_BITS - 1))) >> W_BITS;
                    const int iy = sdot2(dy1, wrow1, sdot2(dy0, wrow0, 1 << (W_BITS - 1))) >> W_BITS;
                    xbuf_o[2 * q] = (uint32_t)bias_of(ival);
                    xbuf_o[2 * q + 1] = (uint32_t)(ix & 0xffff) | ((uint32_t)iy << 16);
                    sA11 += __mul24(ix, ix);   // |ix|, |iy| <= 4080
                    sA12 += __mul24(ix, iy);
                    sA22 += __mul24(iy, iy);
                }
            }
        }
        // per-lane partials fit int32; the half's totals are reduced as exact (hi, lo) halves
        static_assert((long long)NPX * 4080 * 4080 < (1ll << 31), "structure tensor sums fit int32");
        const int S11 = half_sum3_i32(sA11), S22 = half_sum3_i32(sA22);
        float A11 = (float)S11 * FLT_SCALE;
        float A12 = half_exact_sum3_small(sA12) * FLT_SCALE;
        float A22 = (float)S22 * FLT_SCALE;
        float cert_s = 0.f;   // X86: max(S11, S22), the structure-tensor factor of the mismatch vector's exactness bound
        if constexpr (X86) {
            // S11, S22 < 2^24: every partial sum of ix^2, iy^2 and ix iy in any order is exact -- the x86 order gives the
            // canonical values above.  Otherwise (wavefront-uniform branch) both keypoints take the ordered evaluation,
            // which gives the same bits for a keypoint that did not need it.
            cert_s = (float)max(S11, S22);
            if (__any(i_in && (S11 >= (1 << 24) || S22 >= (1 << 24)))) {
                float f11 = 0.f, f12 = 0.f, f22 = 0.f;
                x86_structure_tensor<WIN>(wbase, xbuf_o, lane_o, i_in, f11, f12, f22);
                if (i_in) {
                    A11 = f11 * FLT_SCALE;
                    A12 = f12 * FLT_SCALE;
                    A22 = f22 * FLT_SCALE;
                }
                x86_levels += 1 << 16;
            }
            x86_levels += 1;
        }
        // |sum over any subset of the window of diff * ix| <= sqrt(NPX) * 8160 * sqrt(S11) (Cauchy-Schwarz, |diff| <= 8160):
        // below 2^31 the b-vector's sums over the four lanes of a group can be formed in int32 (then ONE conversion, the
        // same single rounding as the fp64 route).  True for every keypoint of the benchmark clips; a window full of
        // 0 / 255 edges (S = 100 * 4080^2) takes the fp64 route.  Wavefront-uniform: both keypoints must qualify.
        constexpr long long kSmallS = (1ll << 62) / ((long long)NPX * 8160 * 8160);
        const bool int_sums = __all((!i_in) || ((long long)S11 < kSmallS && (long long)S22 < kSmallS));
        float D = A11 * A22 - A12 * A12;
        const float tdiff = A11 - A22;
        const float min_eig_num = A22 + A11 - sqrtf(tdiff * tdiff + 4.f * A12 * A12);
        // min_eig = min_eig_num / (2 WIN^2) < thr, without the division (LKParams::min_eig_num_thr)
        const bool weak = (p.min_eig_num_thr == p.min_eig_num_thr) ? (min_eig_num < p.min_eig_num_thr)
                                                                    : (min_eig_num / (float)(2 * WIN * WIN) < p.min_eig_thr);
        bool lvl_ok = i_in;
        if (i_in && (weak || D < 1.1920928955078125e-07f /* FLT_EPSILON */)) {
            if (level == 0) status = false;
            lvl_ok = false;
        }
        D = 1.f / D;
        // b = sum * 2^-20 enters the solve only through products that are then multiplied by D: the power of two commutes
        // with every rounding on the way (nothing gets near the denormals), so it is applied to D once per level instead
        // of to both sums in every iteration
        D *= FLT_SCALE;
        lvl_ok = lvl_ok && tgt_active;   // idle groups only help with the I side

        // every group picks up the pixels it owns
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        PC_PROF(1);
        int Bias[K];  // 2^15 - ival * 2^16: the accumulator init of interp_r
        int Dxy[K];   // (int16 ix) | (int16 iy << 16); 0 for slots without a pixel
#pragma unroll
        for (int k = 0; k < K; k++) {
            Bias[k] = 0;
            Dxy[k] = 0;
        }
        if (lvl_ok) {
#pragma unroll
            for (int c = 0; c < NCH; c++)
#pragma unroll
                for (int r = 0; r < WIN; r++) {
                    const uint2 v = *reinterpret_cast<const uint2*>(xbuf_o + 2 * (r * WIN + lg_o + GL * c));
                    Bias[c * WIN + r] = (int)v.x;
                    Dxy[c * WIN + r] = (int)v.y;
                }
            // rows r and H1 + r of a chain (the same step of its two runs) share their registers: slot r holds
            // (ix_a, ix_b), slot H1 + r holds (iy_a, iy_b); an odd window's middle row keeps the (ix, iy) form
#pragma unroll
            for (int c = 0; c < NCH; c++)
#pragma unroll
                for (int r = 0; r < WIN - (WIN + 1) / 2; r++) {
                    const uint32_t a = (uint32_t)Dxy[c * WIN + r], b = (uint32_t)Dxy[c * WIN + (WIN + 1) / 2 + r];
                    Dxy[c * WIN + r] = (int)__builtin_amdgcn_perm(b, a, 0x05040100u);
                    Dxy[c * WIN + (WIN + 1) / 2 + r] = (int)__builtin_amdgcn_perm(b, a, 0x07060302u);
                }
#pragma unroll
            for (int e = 0; e < KE; e++) {
                uint2 v = make_uint2(0u, 0u);
                if constexpr (G::RUNS) {
                    if (e < e_len) v = *reinterpret_cast<const uint2*>(xbuf_o + 2 * (e_q0_o + e * WIN));
                } else {
                    if (qE[e] >= 0) v = *reinterpret_cast<const uint2*>(xbuf_o + 2 * qE[e]);
                }
                Bias[KM + e] = (int)v.x;
                Dxy[KM + e] = (int)v.y;
            }
            if constexpr (KE > 1 && G::RUNS) {
                // the run of the remaining columns: pixels 2m and 2m + 1 share slots 2m (ix pair) and 2m + 1 (iy pair)
#pragma unroll
                for (int e = 0; e + 1 < KE; e += 2) {
                    const uint32_t a = (uint32_t)Dxy[KM + e], b = (uint32_t)Dxy[KM + e + 1];
                    Dxy[KM + e] = (int)__builtin_amdgcn_perm(b, a, 0x05040100u);
                    Dxy[KM + e + 1] = (int)__builtin_amdgcn_perm(b, a, 0x07060302u);
                }
            }
        }
        // the J regions alias the I-side buffers of BOTH halves: no group may stage before every group has read
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        PC_PROF(2);
        if (!lvl_ok) continue;

        // ---- iterations on the staged J region ----
        qx -= half_win;
        qy -= half_win;
        float pdx = 0.f, pdy = 0.f;
        int rx0 = 0, ry0 = 0;
        bool staged = false;
        // X86: once the exactness proof of an iteration fails for any pair of the wavefront, the rest of the level runs in
        // the x86 order for all of them (always correct; the proof only saves work) -- wavefront-uniform, so that a
        // wavefront never executes both paths iteration after iteration
        bool x86_ordered = false;
        (void)x86_ordered;
        for (int j = 0; j < p.max_iters; j++) {
            PC_PROF_COUNT(8);
            const float fqx = floorf(qx), fqy = floorf(qy);
            const int iqx = (int)fqx, iqy = (int)fqy;
            if ((unsigned)(iqx + WIN) >= (unsigned)(L.w + WIN) || (unsigned)(iqy + WIN) >= (unsigned)(L.h + WIN)) {
                if (level == 0) status = false;
                break;
            }
            int ox = iqx - rx0, oy = iqy - ry0;
            bool restage = !staged || (unsigned)ox > (unsigned)(2 * G::MX) || (unsigned)oy > (unsigned)(2 * G::MY);
            if (restage) {
                ox = G::MX;
                oy = G::MY;
                rx0 = iqx - ox;
                ry0 = iqy - oy;
                PC_PROF(4);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                stage_region_paired<WIN>(J16, pitch, rx0, ry0, jbuf, lg);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                staged = true;
                PC_PROF_COUNT(9);
                PC_PROF(3);
            }
            const Weights wJ = packed_weights(qx - fqx, qy - fqy);   // (float)iqx == fqx
            const uint32_t* jq = jbuf + __mul24(oy, G::PITCH) + ox;
            int sb1 = 0, sb2 = 0;  // per-lane partials: <= K * 8160 * 4080
            int dd0 = 0, dd1 = 0;  // X86: the lane's sum of squared differences (<= K * 8160^2 < 2^31), for the exactness proof
            (void)dd0;
            (void)dd1;
            constexpr bool kExtrasFull = G::NEXTRA == GL * KE;   // every lane's run of extra pixels is complete
            (void)kExtrasFull;
            if (!X86 || !x86_ordered) {
            // A pixel is a dependent chain dot2 -> dot2 -> mad with wait states after each dot product; walked one
            // pixel after the other that chain, not instruction issue, sets the pace (lk2 and a first version of this
            // loop: 2 x s_nop 2 per pixel).  So the lane's pixels are cut into NR independent vertical runs -- every
            // column chain in two halves, plus the run of the remaining columns -- and step s handles pixel s of all
            // runs: all first dot products, then all second ones, then the accumulations.
            {
                constexpr int H1 = (WIN + 1) / 2, H2 = WIN - H1;          // rows of a chain's upper / lower half
                constexpr int NRC = 2 * NCH;                              // chain halves
                constexpr int NR = NRC + ((KE > 0 && G::RUNS) ? 1 : 0);
                constexpr int STEPS = (KE > 0 && G::RUNS && KE > H1) ? KE : H1;
                uint32_t top[NR > 0 ? NR : 1];
                const uint32_t* rb[NR > 0 ? NR : 1];
                int tb1 = 0, tb2 = 0;
                int r_even = 0;   // the extra run's result of the even step, waiting for its partner
                (void)r_even;
#pragma unroll
                for (int u = 0; u < NRC; u++) {
                    rb[u] = jq + lg + GL * (u >> 1) + ((u & 1) ? H1 * G::PITCH : 0);
                    top[u] = rb[u][0];
                }
                if constexpr (NR > NRC) {
                    rb[NRC] = jq + e_off;
                    top[NRC] = rb[NRC][0];
                }
                // the LDS reads run one step ahead of the arithmetic (a ds_read takes 60+ cycles to return)
                uint32_t bot[NR > 0 ? NR : 1], nxt[NR > 0 ? NR : 1];
#pragma unroll
                for (int u = 0; u < NR; u++) bot[u] = rb[u][G::PITCH];
#pragma unroll
                for (int st = 0; st < STEPS; st++) {
                    int R[NR > 0 ? NR : 1];
#pragma unroll
                    for (int u = 0; u < NR; u++) {
                        const int len = u < NRC ? ((u & 1) ? H2 : H1) : KE;
                        if (st + 1 < len) nxt[u] = rb[u][(st + 2) * G::PITCH];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < NR; u++) {
                        const int len = u < NRC ? ((u & 1) ? H2 : H1) : KE;
                        const int k = u < NRC ? (u >> 1) * WIN + ((u & 1) ? H1 : 0) + st : KM + st;
                        if (st < len)
                            R[u] = __builtin_amdgcn_sdot2(__builtin_bit_cast(pc_short2, top[u]), __builtin_bit_cast(pc_short2, wJ.r0), Bias[k], true);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < NR; u++) {
                        const int len = u < NRC ? ((u & 1) ? H2 : H1) : KE;
                        if (st < len) {
                            R[u] = sdot2(bot[u], wJ.r1, R[u]);
                            top[u] = bot[u];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (st < H2) {
                        uint32_t Rp[NCH > 0 ? NCH : 1];
#pragma unroll
                        for (int c = 0; c < NCH; c++) Rp[c] = __builtin_amdgcn_perm((uint32_t)R[2 * c + 1], (uint32_t)R[2 * c], 0x07060302u);
#pragma unroll
                        for (int c = 0; c < NCH; c++) {
                            if (st == 0 && c < 2) {   // the first term of an accumulator
                                (c ? tb1 : sb1) = sdot2_zero(Rp[c], (uint32_t)Dxy[c * WIN + st]);
                                (c ? tb2 : sb2) = sdot2_zero(Rp[c], (uint32_t)Dxy[c * WIN + H1 + st]);
                            } else if (c & 1) {
                                tb1 = sdot2(Rp[c], (uint32_t)Dxy[c * WIN + st], tb1);
                                tb2 = sdot2(Rp[c], (uint32_t)Dxy[c * WIN + H1 + st], tb2);
                            } else {
                                sb1 = sdot2(Rp[c], (uint32_t)Dxy[c * WIN + st], sb1);
                                sb2 = sdot2(Rp[c], (uint32_t)Dxy[c * WIN + H1 + st], sb2);
                            }
                        }
                        if constexpr (X86) {
#pragma unroll
                            for (int c = 0; c < NCH; c++) {
                                if (c & 1) dd1 = sdot2(Rp[c], Rp[c], dd1);
                                else dd0 = sdot2(Rp[c], Rp[c], dd0);
                            }
                        }
                    } else {   // an odd window's middle row: the upper runs only
#pragma unroll
                        for (int c = 0; c < NCH; c++) {
                            if (st < H1) {
                                sb1 = mad16_hl(R[2 * c], (uint32_t)Dxy[c * WIN + st], sb1);
                                sb2 = mad16_hh(R[2 * c], (uint32_t)Dxy[c * WIN + st], sb2);
                                if constexpr (X86) dd0 = mad16_hh(R[2 * c], (uint32_t)R[2 * c], dd0);
                            }
                        }
                    }
                    if constexpr (NR > NRC) {
                        // the run of the remaining columns: its pixels of steps 2m and 2m + 1 as a pair as well
                        if (st < KE) {
                            if ((st & 1) == 0 && st + 1 < KE) {
                                r_even = R[NRC];
                            } else if (st & 1) {
                                const uint32_t rp = __builtin_amdgcn_perm((uint32_t)R[NRC], (uint32_t)r_even, 0x07060302u);
                                tb1 = sdot2(rp, (uint32_t)Dxy[KM + st - 1], tb1);
                                tb2 = sdot2(rp, (uint32_t)Dxy[KM + st], tb2);
                                if constexpr (X86 && kExtrasFull) dd1 = sdot2(rp, rp, dd1);
                            } else {
                                tb1 = mad16_hl(R[NRC], (uint32_t)Dxy[KM + st], tb1);
                                tb2 = mad16_hh(R[NRC], (uint32_t)Dxy[KM + st], tb2);
                                if constexpr (X86 && kExtrasFull) dd1 = mad16_hh(R[NRC], (uint32_t)R[NRC], dd1);
                            }
                            if constexpr (X86 && !kExtrasFull) {   // slots past a short run hold arbitrary differences: keep them out
                                const int d2 = mad16_hh(R[NRC], (uint32_t)R[NRC], 0);
                                dd1 += st < e_len ? d2 : 0;
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < NR; u++) bot[u] = nxt[u];
                    __builtin_amdgcn_sched_barrier(0);
                }
                sb1 += tb1;   // integer sums: any order gives the same bits
                sb2 += tb2;
            }
            if constexpr (KE > 0 && !G::RUNS) {
                int R[KE];
#pragma unroll
                for (int e = 0; e < KE; e++)
                    R[e] = __builtin_amdgcn_sdot2(__builtin_bit_cast(pc_short2, jq[offE[e]]), __builtin_bit_cast(pc_short2, wJ.r0), Bias[KM + e], true);
#pragma unroll
                for (int e = 0; e < KE; e++) R[e] = sdot2(jq[offE[e] + G::PITCH], wJ.r1, R[e]);
#pragma unroll
                for (int e = 0; e < KE; e++) {
                    sb1 = mad16_hl(R[e], (uint32_t)Dxy[KM + e], sb1);
                    sb2 = mad16_hh(R[e], (uint32_t)Dxy[KM + e], sb2);
                    if constexpr (X86) {
                        const int d2 = mad16_hh(R[e], (uint32_t)R[e], 0);
                        dd0 += (kExtrasFull || qE[e] >= 0) ? d2 : 0;
                    }
                }
            }
            }   // !x86_ordered
            float b1 = 0.f, b2 = 0.f;
            bool run_ordered = X86 && x86_ordered;
            if (!run_ordered) {
                if (int_sums) {
                    int t1 = sb1 + dpp_i32<0xB1>(sb1), t2 = sb2 + dpp_i32<0xB1>(sb2);
                    t1 += dpp_i32<0x4E>(t1);
                    t2 += dpp_i32<0x4E>(t2);
                    b1 = (float)t1;
                    b2 = (float)t2;
                } else {
                    b1 = group4_exact_sum3<K>(sb1);
                    b2 = group4_exact_sum3<K>(sb2);
                }
                if constexpr (X86) {
                    // sum over the window of |d ix| <= sqrt(sum d^2 * S11) <= 2^24 (and the same for iy): every partial sum of
                    // the x86 order is exact and its result is the float of the integer total, b1 / b2 above.  The margin
                    // (2^-10) covers the roundings of this test itself (five fp32 operations, < 2^-21 relative).
                    float ddf = (float)(dd0 + dd1);
                    ddf += dpp_f32<0xB1>(ddf);
                    ddf += dpp_f32<0x4E>(ddf);
                    const bool proven = ddf * cert_s <= 281474976710656.f * (1.f - 1.f / 1024.f);
                    if (__any(!proven)) {
                        x86_ordered = true;
                        run_ordered = true;
                    }
                }
            }
            if constexpr (X86) {
                if (run_ordered) {
                    x86_mismatch_ordered<WIN, K, (KE > 0 ? KE : 1)>(jq, lg, wJ, Bias, Dxy, e_off, offE, b1, b2);
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
        PC_PROF(4);
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
                stage_region_paired<WIN>(J16, pitch, rx0, ry0, jbuf, lg);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                staged = true;
            }
            const Weights wE = packed_weights(ex - (float)iex, ey - (float)iey);
            const uint32_t* jq = jbuf + (iey - ry0) * G::PITCH + (iex - rx0);
            int se = 0;
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const uint32_t* cb = jq + lg + GL * c;
                uint32_t top = cb[0];
#pragma unroll
                for (int r = 0; r < WIN; r++) {
                    const uint32_t bot = cb[(r + 1) * G::PITCH];
                    const int diff = interp_r(top, bot, wE, Bias[c * WIN + r]) >> 16;
                    top = bot;
                    se += diff < 0 ? -diff : diff;
                }
            }
            if constexpr (KE > 0) {
                if constexpr (G::RUNS) {
                    const uint32_t* cb = jq + e_off;
                    uint32_t top = cb[0];
#pragma unroll
                    for (int e = 0; e < KE; e++) {
                        const uint32_t bot = cb[(e + 1) * G::PITCH];
                        const int diff = interp_r(top, bot, wE, Bias[KM + e]) >> 16;
                        top = bot;
                        se += (e < e_len) ? (diff < 0 ? -diff : diff) : 0;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < KE; e++) {
                        const uint32_t* q = jq + offE[e];
                        const int diff = interp_r(q[0], q[G::PITCH], wE, Bias[KM + e]) >> 16;
                        se += (qE[e] >= 0) ? (diff < 0 ? -diff : diff) : 0;
                    }
                }
            }
            se += dpp_i32<0xB1>(se);   // <= 256 * 8160 < 2^24: exact in fp32 too
            se += dpp_i32<0x4E>(se);
            err = ((float)se * 1.f) / (float)(32 * WIN * WIN);
        }
        PC_PROF(5);
    }
#ifdef PC_LK_PROFILE
    if (lane == 0 && p.prof) {   // one row per wavefront, summed on the host (atomics on ten words would clog the L2)
        prof_acc[6] = __builtin_readcyclecounter() - prof_t0;
        prof_acc[7] = 1;
        for (int k = 0; k < 10; k++) p.prof[(size_t)(first >> 1) * 16 + k] = prof_acc[k];
    }
#endif

    // one 16-byte record per (slot, target): the wavefront's results are contiguous
    if (lg == 0 && tgt_active)
        p.out_rec[(size_t)slot * kRecStride + tgt] = make_float4(nx, ny, status ? err : 0.f, __uint_as_float(status ? 1u : 0u));
    if constexpr (X86) {
        if (p.x86_stats) {   // diagnostics only (pc_debug_lk_x86_stats)
            if (lg == 0 && tgt_active) {
                atomicAdd(&p.x86_stats[0], (unsigned long long)(x86_iters & 0xffff));
                atomicAdd(&p.x86_stats[1], (unsigned long long)(x86_iters >> 16));
            }
            if (l32 == 0 && kp_valid) {
                atomicAdd(&p.x86_stats[2], (unsigned long long)(x86_levels & 0xffff));
                atomicAdd(&p.x86_stats[3], (unsigned long long)(x86_levels >> 16));
            }
        }
    }
}

// smallest float x with fl(x / c) >= thr (c > 0, thr finite and positive); NaN if the search does not settle
static float division_threshold(float thr, float c) {
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

template <int WIN>
static void launch_lk3_t(const LKParams& p0, hipStream_t s) {
    LKParams p = p0;
    p.min_eig_num_thr = division_threshold(p.min_eig_thr, (float)(2 * WIN * WIN));
    const int per_block = 2;   // two keypoints per wavefront, one wavefront per workgroup
    const int blocks = (p.n + per_block - 1) / per_block;
    if (blocks == 0) return;
    p.blocks_per_xcd = (blocks + 7) / 8;
    if (p.x86_order) hipLaunchKernelGGL((lk3_kernel<WIN, true>), dim3((unsigned)p.blocks_per_xcd * 8u), dim3(64), 0, s, p);
    else hipLaunchKernelGGL((lk3_kernel<WIN, false>), dim3((unsigned)p.blocks_per_xcd * 8u), dim3(64), 0, s, p);
}

bool lk_profile_enabled() {
#ifdef PC_LK_PROFILE
    return true;
#else
    return false;
#endif
}

bool launch_lk3(const LKParams& p, int win, hipStream_t s) {
    if (!p.src[0].img16) return false;
    switch (win) {
#define PC_LK_CASE(W) case W: launch_lk3_t<W>(p, s); return true;
        PC_LK_CASE(4) PC_LK_CASE(5) PC_LK_CASE(6) PC_LK_CASE(7) PC_LK_CASE(8) PC_LK_CASE(9) PC_LK_CASE(10) PC_LK_CASE(11)
#undef PC_LK_CASE
        default: return false;
    }
}

}  // namespace pc
