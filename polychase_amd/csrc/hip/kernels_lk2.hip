// kernels_lk2.hip -- pyramidal Lucas-Kanade, variant with TWO keypoints per wavefront (K8-K10).
//
// Same arithmetic and results as kernels_lk.hip (bit for bit; both follow oracle/pc_oracle.c), other
// mapping: lanes 0-31 track keypoint A, lanes 32-63 keypoint B; inside a half, group g = 4 lanes
// tracks the keypoint into target g (<= 8 targets).  Everything that does not depend on the window
// pixels -- level set-up, bilinear weights, the 2x2 solve, the convergence tests, loop control --
// is issued once per wavefront and now serves 16 (keypoint, target) pairs instead of 8; that
// uniform work is about half of the one-keypoint kernel's instruction stream.
//   * I side (patch, Scharr patch, structure tensor): one half-wave per keypoint, through LDS.
//   * J side: a group stages its (WIN+7)-row search region as byte pairs and owns the window's
//     columns lg, lg+4, ... as column chains (one ds_read_u16 per pixel) plus a share of the
//     columns that do not fill a chain.
//   * The I-side buffers alias the J regions (they are dead once every group has picked up its
//     pixels), which keeps the LDS footprint at 16 regions per wavefront.
#include "lk_common.hpp"

namespace pc {

#ifndef PC_LK2_WAVES
#define PC_LK2_WAVES 1   // wavefronts per workgroup
#endif
#ifndef PC_LK2_DXY_LDS
// 1: the (ix, iy) gradients stay in LDS and are re-read every iteration: 128 instead of 137 VGPRs, 4 waves/SIMD, and
// the launch alone is 2 % faster -- but four such waves fill a SIMD's register file, the preparation kernels of the
// other stream no longer fit beside them, and the frame rate DROPS 15 % (1413 vs 1665 fps at C2).  Kept at 0.
#define PC_LK2_DXY_LDS 0
#endif

template <int WIN>
struct LK2Geo {
    using G = LKGeo<WIN>;
    static constexpr int GL = 4;
    static constexpr int WM = (WIN / GL) * GL;            // columns handled as per-lane column chains
    static constexpr int NCH = WM / GL;                   // chains per lane
    static constexpr int KM = NCH * WIN;                  // chain slots per lane
    static constexpr int NEXTRA = (WIN - WM) * WIN;       // pixels of the remaining columns
    static constexpr int KE = (NEXTRA + GL - 1) / GL;     // their slots per lane
    static constexpr int K = KM + KE;
    static constexpr int HALF_I_DW = G::I_DW + G::D_DW + G::X_DW;   // I-side buffers of one keypoint
    static constexpr int J_ALL_DW = 16 * G::J_DW;
    static constexpr int RAW_DW = J_ALL_DW > 2 * HALF_I_DW ? J_ALL_DW : 2 * HALF_I_DW;
    static constexpr int DXY_DW = PC_LK2_DXY_LDS ? 2 * G::NPX : 0;   // gradients of both keypoints, kept for the whole level
    static constexpr int WAVE_DW = ((RAW_DW + DXY_DW + 1) / 2) * 2;
};

// every lane gets the sum over its 32-lane half: DPP inside the 16-lane rows, then the other row
__device__ __forceinline__ int half_sum_i32(int v) {
    v = group_allreduce_add<16>(v);
    return v + __shfl_xor(v, 16);
}
__device__ __forceinline__ float half_exact_sum(int partial) {
    return exact_sum_to_float(half_sum_i32(partial >> 16), half_sum_i32(partial & 0xffff));
}
// sum over a 4-lane group as ONE rounding of the exact integer
template <int K>
__device__ __forceinline__ float group4_exact_sum(int partial) {
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

template <int WIN>
__global__ __launch_bounds__(64 * PC_LK2_WAVES) void lk2_kernel(const LKParams p) {
    using G = LKGeo<WIN>;
    using G2 = LK2Geo<WIN>;
    constexpr int GL = G2::GL, NPX = WIN * WIN, NCH = G2::NCH, KM = G2::KM, KE = G2::KE, K = G2::K;
    constexpr int KW = (NPX + 31) / 32;   // pixels per lane in the half-wave I-side pass
    __shared__ __attribute__((aligned(16))) uint32_t s_buf[PC_LK2_WAVES][G2::WAVE_DW];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, l32 = lane & 31, grp = (lane >> 2) & 7, lg = lane & 3;
    // Workgroup b runs on XCD b % 8; each XCD takes one contiguous eighth of the (spatially binned) keypoint order
    lk_signal_dispatched(p);
    const int lb = (int)(blockIdx.x & 7u) * p.blocks_per_xcd + (int)(blockIdx.x >> 3);
    const int first = (lb * PC_LK2_WAVES + wave) * 2;  // first of this wave's two keypoint slots
    if ((int)(blockIdx.x >> 3) >= p.blocks_per_xcd || first >= p.n) return;   // whole waves exit together
    const int slot = first + half;
    const bool kp_valid = slot < p.n;                 // n odd: the last wave's second half idles
    const int slot_c = kp_valid ? slot : first;
    const int feat = p.perm ? (int)p.perm[slot_c] : slot_c;
    const bool tgt_active = kp_valid && grp < p.n_targets;
    const int tgt = grp < p.n_targets ? grp : 0;

    uint32_t* const wbase = &s_buf[wave][0];
    uint8_t* const ibuf = reinterpret_cast<uint8_t*>(wbase + half * G2::HALF_I_DW);                 // I window, pair format
    uint8_t* const dbuf = reinterpret_cast<uint8_t*>(wbase + half * G2::HALF_I_DW + G::I_DW);       // raw Scharr window
    uint32_t* const xbuf = wbase + half * G2::HALF_I_DW + G::I_DW + G::D_DW;                        // (Ival, Dxy) exchange
    uint8_t* const jbuf = reinterpret_cast<uint8_t*>(wbase + (half * 8 + grp) * G::J_DW);           // aliases the above
    uint32_t* const gbuf = wbase + G2::RAW_DW + half * NPX;                                         // (ix, iy) per pixel, not aliased

    // window pixels of the remaining columns, dealt out pixel by pixel
    int offE[KE > 0 ? KE : 1], qE[KE > 0 ? KE : 1];
#pragma unroll
    for (int e = 0; e < KE; e++) {
        const int r = lg + GL * e;
        const int col = G2::WM + r / WIN, row = r - (r / WIN) * WIN;
        const bool ok = r < G2::NEXTRA;
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

        // ---- I side: identical for all targets of a keypoint -> computed once by its half-wave ----
        px -= half_win;
        py -= half_win;
        const int ipx = (int)floorf(px), ipy = (int)floorf(py);
        const bool i_in = !(ipx < -WIN || ipx >= L.w || ipy < -WIN || ipy >= L.h);   // uniform per half
        if (!i_in && level == 0) {
            status = false;
            err = 0.f;
        }
        const Weights wI = bilinear_weights(px - (float)ipx, py - (float)ipy);
        const uint32_t wrow0 = wI.r0, wrow1 = wI.r1;

        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the previous level's J regions are dead
        if (i_in) {
            DerivWindow<WIN, 32> dw;
            dw.load(L.der + (ptrdiff_t)(ipy * pitch + ipx), pitch, l32);
            stage_pairs_auto<WIN, 32, WIN + 1>(L.img, pitch, L.h, ipx & ~3, ipy, ibuf, l32);
            dw.store(reinterpret_cast<int32_t*>(dbuf), l32);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        int sA11 = 0, sA12 = 0, sA22 = 0;
        if (i_in) {
            const uint8_t* ib = ibuf + 2 * (ipx & 3);
#pragma unroll
            for (int m = 0; m < KW; m++) {
                const int q = l32 + 32 * m;
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
                    if (PC_LK2_DXY_LDS) gbuf[q] = (uint32_t)(ix & 0xffff) | ((uint32_t)iy << 16);
                    sA11 += __mul24(ix, ix);   // |ix|, |iy| <= 4080
                    sA12 += __mul24(ix, iy);
                    sA22 += __mul24(iy, iy);
                }
            }
        }
        // per-lane partials fit int32; the half's totals are reduced as exact (hi, lo) halves
        const float A11 = half_exact_sum(sA11) * FLT_SCALE;
        const float A12 = half_exact_sum(sA12) * FLT_SCALE;
        const float A22 = half_exact_sum(sA22) * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float tdiff = A11 - A22;
        const float min_eig = (A22 + A11 - sqrtf(tdiff * tdiff + 4.f * A12 * A12)) / (float)(2 * WIN * WIN);
        bool lvl_ok = i_in;
        if (i_in && (min_eig < p.min_eig_thr || D < 1.1920928955078125e-07f /* FLT_EPSILON */)) {
            if (level == 0) status = false;
            lvl_ok = false;
        }
        D = 1.f / D;
        lvl_ok = lvl_ok && tgt_active;   // idle groups only help with the I side

        // every group picks up the pixels it owns
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        int Ival[K];  // ival_bias(I patch value): the accumulator init of interp_diff
        int Dxy[PC_LK2_DXY_LDS ? 1 : K];   // (int16 ix) | (int16 iy << 16); 0 for slots without a pixel
#pragma unroll
        for (int k = 0; k < K; k++) {
            Ival[k] = 0;
            if (!PC_LK2_DXY_LDS) Dxy[k] = 0;
        }
        if (lvl_ok) {
#pragma unroll
            for (int c = 0; c < NCH; c++)
#pragma unroll
                for (int r = 0; r < WIN; r++) {
                    const uint2 v = *reinterpret_cast<const uint2*>(xbuf + 2 * (r * WIN + lg + GL * c));
                    Ival[c * WIN + r] = ival_bias((int)v.x);
                    if (!PC_LK2_DXY_LDS) Dxy[c * WIN + r] = (int)v.y;
                }
#pragma unroll
            for (int e = 0; e < KE; e++) {
                const uint2 v = (qE[e] >= 0) ? *reinterpret_cast<const uint2*>(xbuf + 2 * qE[e]) : make_uint2(0u, 0u);
                Ival[KM + e] = ival_bias((int)v.x);
                if (!PC_LK2_DXY_LDS) Dxy[KM + e] = (int)v.y;
            }
        }
        // the J regions alias the I-side buffers of BOTH halves: no group may stage before every group has read
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (!lvl_ok) continue;

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
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const uint8_t* cb = jb + 2 * (lg + GL * c);
                uint32_t top = widen_pair(*reinterpret_cast<const uint16_t*>(cb));
#pragma unroll
                for (int r = 0; r < WIN; r++) {
                    const uint32_t bot = widen_pair(*reinterpret_cast<const uint16_t*>(cb + (r + 1) * G::PAIR_PITCH));
                    const int diff = interp_diff(top, bot, wJ, Ival[c * WIN + r]);
                    top = bot;
                    const uint32_t g = PC_LK2_DXY_LDS ? gbuf[r * WIN + lg + GL * c] : (uint32_t)Dxy[PC_LK2_DXY_LDS ? 0 : c * WIN + r];
                    sb1 = mad16_lo(diff, g, sb1);
                    sb2 = mad16_hi(diff, g, sb2);
                }
            }
#pragma unroll
            for (int e = 0; e < KE; e++) {
                const uint16_t* q = reinterpret_cast<const uint16_t*>(jb + offE[e]);
                const int diff = interp_diff(widen_pair(q[0]), widen_pair(q[G::RWB]), wJ, Ival[KM + e]);
                // slots past the window (qE < 0) read pixel 0's gradient: their diff must not count
                const uint32_t g = PC_LK2_DXY_LDS ? (qE[e] >= 0 ? gbuf[qE[e]] : 0u) : (uint32_t)Dxy[PC_LK2_DXY_LDS ? 0 : KM + e];
                sb1 = mad16_lo(diff, g, sb1);
                sb2 = mad16_hi(diff, g, sb2);
            }
            const float b1 = group4_exact_sum<K>(sb1) * FLT_SCALE;
            const float b2 = group4_exact_sum<K>(sb2) * FLT_SCALE;
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
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const uint8_t* cb = jb + 2 * (lg + GL * c);
                uint32_t top = widen_pair(*reinterpret_cast<const uint16_t*>(cb));
#pragma unroll
                for (int r = 0; r < WIN; r++) {
                    const uint32_t bot = widen_pair(*reinterpret_cast<const uint16_t*>(cb + (r + 1) * G::PAIR_PITCH));
                    const int diff = interp_diff(top, bot, wE, Ival[c * WIN + r]);
                    top = bot;
                    se += diff < 0 ? -diff : diff;
                }
            }
#pragma unroll
            for (int e = 0; e < KE; e++) {
                const uint16_t* q = reinterpret_cast<const uint16_t*>(jb + offE[e]);
                const int diff = interp_diff(widen_pair(q[0]), widen_pair(q[G::RWB]), wE, Ival[KM + e]);
                se += (qE[e] >= 0) ? (diff < 0 ? -diff : diff) : 0;
            }
            se += dpp_i32<0xB1>(se);   // <= 256 * 8160 < 2^24: exact in fp32 too
            se += dpp_i32<0x4E>(se);
            err = ((float)se * 1.f) / (float)(32 * WIN * WIN);
        }
    }

    // one 16-byte record per (slot, target): the wavefront's results are contiguous
    if (lg == 0 && tgt_active)
        p.out_rec[(size_t)slot * kRecStride + tgt] = make_float4(nx, ny, status ? err : 0.f, __uint_as_float(status ? 1u : 0u));
}

template <int WIN>
static void launch_lk2_t(const LKParams& p0, hipStream_t s) {
    LKParams p = p0;
    const int per_block = 2 * PC_LK2_WAVES;   // two keypoints per wavefront
    const int blocks = (p.n + per_block - 1) / per_block;
    if (blocks == 0) return;
    p.blocks_per_xcd = (blocks + 7) / 8;
    hipLaunchKernelGGL((lk2_kernel<WIN>), dim3((unsigned)p.blocks_per_xcd * 8u), dim3(64 * PC_LK2_WAVES), 0, s, p);
}

bool launch_lk2(const LKParams& p, int win, hipStream_t s) {
    switch (win) {
#define PC_LK_CASE(W) case W: launch_lk2_t<W>(p, s); return true;
        PC_LK_CASE(4) PC_LK_CASE(5) PC_LK_CASE(6) PC_LK_CASE(7) PC_LK_CASE(8) PC_LK_CASE(9) PC_LK_CASE(10) PC_LK_CASE(11)
#undef PC_LK_CASE
        default: return false;
    }
}

}  // namespace pc
