// valu_issue.hip -- issue rate of single VALU instructions on gfx950, per SIMD, with 1-3 resident wavefronts per SIMD
// (run ON the GPU box; built by tools/valu_issue.py).
//
// One workgroup of 256 * W lanes per CU (a 96-KB LDS allocation keeps a second workgroup off the CU), i.e. exactly W
// wavefronts on every SIMD.  Each wavefront issues REPS x 32 copies of ONE instruction on 8 independent accumulators
// (no instruction reads the result of the 7 before it) between two s_memtime reads; reported: shader cycles per
// wave-instruction and per SIMD (= elapsed / (instructions x W)), and the same from the host's event time at the
// nominal 2.4 GHz.  The LK kernel's issue ceiling (bench.py valu_roofline.peak) is set from these numbers.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int REPS = 8192;

// 8 accumulators a0..a7, operands x, y (VGPRs).  OP(acc) expands to one asm statement.
#define BODY(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#define BODY4(OP) BODY(OP) BODY(OP) BODY(OP) BODY(OP)

#define KERNEL(NAME, OP)                                                                                  \
    __global__ __launch_bounds__(1024) void NAME(unsigned long long* out, unsigned x0, unsigned y0) {        \
        extern __shared__ unsigned s_pad[];                                                                   \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        unsigned x = x0 + threadIdx.x, y = y0 ^ threadIdx.x;                                                 \
        asm volatile("" : "+v"(x), "+v"(y));                                                                  \
        __syncthreads();                                                                                      \
        const unsigned long long w0 = wall_clock64();                                                         \
        const unsigned long long t0 = __builtin_readcyclecounter();                                           \
        for (int r = 0; r < REPS; r++) { BODY4(OP) }                                                          \
        const unsigned long long t1 = __builtin_readcyclecounter();                                           \
        const unsigned long long w1 = wall_clock64();                                                         \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345u) s_pad[threadIdx.x] = 1;                       \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                     \
        if (threadIdx.x == 0) out[(size_t)gridDim.x * 16 + blockIdx.x] = w1 - w0;                             \
    }

// ---- single instructions ----
#define OP_FMA_F32(a) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
#define OP_MUL_F32(a) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a) : "v"(x));
#define OP_ADD_F32(a) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a) : "v"(x));
#define OP_ADD_U32(a) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a) : "v"(x));
#define OP_AND_B32(a) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a) : "v"(x));
#define OP_LSHL(a) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a));
#define OP_ASHR(a) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(a));
#define OP_MOV(a) asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(x));
#define OP_MAD_U24(a) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
#define OP_MAD_I24(a) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
#define OP_MUL_LO(a) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a) : "v"(x));
#define OP_DOT2_I16(a) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
#define OP_DOT2_I16_Z(a) asm volatile("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(a) : "v"(x), "v"(y));
#define OP_DOT2C(a) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
#define OP_DOT4_I8(a) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
#define OP_DOT2_F16(a) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
#define OP_MAD_I32_I16(a) asm volatile("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"(a) : "v"(x), "v"(y));
#define OP_MAD_I32_I16_P(a) asm volatile("v_mad_i32_i16 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
#define OP_PERM(a) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(a) : "v"(x), "v"(y));
#define OP_ALIGNBIT(a) asm volatile("v_alignbit_b32 %0, %1, %0, 16" : "+v"(a) : "v"(x));
#define OP_BFE(a) asm volatile("v_bfe_i32 %0, %0, 3, 9" : "+v"(a));
#define OP_PK_FMA_F32(a) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d##a) : "v"(dx), "v"(dy));
#define OP_PK_MAD_I16(a) asm volatile("v_pk_mad_i16 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
#define OP_PK_MUL_LO_U16(a) asm volatile("v_pk_mul_lo_u16 %0, %1, %0" : "+v"(a) : "v"(x));
#define OP_PK_ADD_U16(a) asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(a) : "v"(x));
#define OP_PK_FMA_F16(a) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
#define OP_CVT_F32_I32(a) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a));
#define OP_CVT_I32_F32(a) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a));
#define OP_FLOOR_F32(a) asm volatile("v_floor_f32 %0, %0" : "+v"(a));
#define OP_RCP_F32(a) asm volatile("v_rcp_f32 %0, %0" : "+v"(a));
#define OP_SQRT_F32(a) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a));
#define OP_DPP_ADD(a) asm volatile("v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a));
#define OP_SAD_U8(a) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
#define OP_MAX_F32(a) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a) : "v"(x));
#define OP_CNDMASK(a) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(x));
#define OP_LSHL_ADD(a) asm volatile("v_lshl_add_u32 %0, %1, 2, %0" : "+v"(a) : "v"(x));
#define OP_ADD3(a) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
#define OP_MAD_U16(a) asm volatile("v_mad_u16 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
#define OP_MAD_I16(a) asm volatile("v_mad_i16 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
#define OP_MAD_U32_U16(a) asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
#define OP_FMA_F64(a) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d##a) : "v"(dx), "v"(dy));
#define OP_ADD_F64(a) asm volatile("v_add_f64 %0, %1, %0" : "+v"(d##a) : "v"(dx));
#define OP_CVT_F64_F32(a) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d##a) : "v"(x));
#define OP_CVT_F32_F64(a) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a) : "v"(dx));
// the LK pixel: dot2 -> dot2 -> (perm + 2 dot2 per two pixels); here per accumulator: dot2, dot2, perm, dot2 (3.5 of 4)
#define OP_LKMIX(a) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0\n v_perm_b32 %0, %1, %0, %2" : "+v"(a) : "v"(x), "v"(y));
#define OP_FMA_DOT2(a) asm volatile("v_fma_f32 %0, %1, %2, %0\n v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));

KERNEL(k_fma_f32, OP_FMA_F32)
KERNEL(k_mul_f32, OP_MUL_F32)
KERNEL(k_add_f32, OP_ADD_F32)
KERNEL(k_add_u32, OP_ADD_U32)
KERNEL(k_and_b32, OP_AND_B32)
KERNEL(k_lshl, OP_LSHL)
KERNEL(k_ashr, OP_ASHR)
KERNEL(k_mov, OP_MOV)
KERNEL(k_mad_u24, OP_MAD_U24)
KERNEL(k_mad_i24, OP_MAD_I24)
KERNEL(k_mul_lo, OP_MUL_LO)
KERNEL(k_dot2_i16, OP_DOT2_I16)
KERNEL(k_dot2_i16_zero, OP_DOT2_I16_Z)
KERNEL(k_dot2c_i16, OP_DOT2C)
KERNEL(k_dot4_i8, OP_DOT4_I8)
KERNEL(k_dot2_f16, OP_DOT2_F16)
KERNEL(k_mad_i32_i16_opsel, OP_MAD_I32_I16)
KERNEL(k_mad_i32_i16, OP_MAD_I32_I16_P)
KERNEL(k_perm, OP_PERM)
KERNEL(k_alignbit, OP_ALIGNBIT)
KERNEL(k_bfe, OP_BFE)
KERNEL(k_pk_mad_i16, OP_PK_MAD_I16)
KERNEL(k_pk_mul_lo_u16, OP_PK_MUL_LO_U16)
KERNEL(k_pk_add_u16, OP_PK_ADD_U16)
KERNEL(k_pk_fma_f16, OP_PK_FMA_F16)
KERNEL(k_cvt_f32_i32, OP_CVT_F32_I32)
KERNEL(k_cvt_i32_f32, OP_CVT_I32_F32)
KERNEL(k_floor_f32, OP_FLOOR_F32)
KERNEL(k_rcp_f32, OP_RCP_F32)
KERNEL(k_sqrt_f32, OP_SQRT_F32)
KERNEL(k_dpp_add, OP_DPP_ADD)
KERNEL(k_sad_u8, OP_SAD_U8)
KERNEL(k_max_f32, OP_MAX_F32)
KERNEL(k_cndmask, OP_CNDMASK)
KERNEL(k_lshl_add, OP_LSHL_ADD)
KERNEL(k_add3, OP_ADD3)
KERNEL(k_mad_u16, OP_MAD_U16)
KERNEL(k_mad_i16, OP_MAD_I16)
KERNEL(k_mad_u32_u16, OP_MAD_U32_U16)
KERNEL(k_lkmix_dot2_perm, OP_LKMIX)
KERNEL(k_fma_then_dot2, OP_FMA_DOT2)

// dependent chains: every instruction reads the result of the one before it (issue-to-issue latency of a chain)
#undef BODY
#define BODY(OP) OP(a0) OP(a0) OP(a0) OP(a0) OP(a0) OP(a0) OP(a0) OP(a0)
KERNEL(k_dep_fma_f32, OP_FMA_F32)
KERNEL(k_dep_add_u32, OP_ADD_U32)
KERNEL(k_dep_mad_u24, OP_MAD_U24)
KERNEL(k_dep_dot2_i16, OP_DOT2_I16)
KERNEL(k_dep_mad_i32_i16, OP_MAD_I32_I16)
KERNEL(k_dep_perm, OP_PERM)
KERNEL(k_dep_dot4_i8, OP_DOT4_I8)
KERNEL(k_dep_pk_mad_i16, OP_PK_MAD_I16)
#undef BODY
#define BODY(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)

// 64-bit operands
#define KERNEL64(NAME, OP)                                                                                \
    __global__ __launch_bounds__(1024) void NAME(unsigned long long* out, unsigned x0, unsigned y0) {        \
        extern __shared__ unsigned s_pad[];                                                                   \
        double da0 = threadIdx.x, da1 = da0 + 1, da2 = da0 + 2, da3 = da0 + 3, da4 = da0 + 4, da5 = da0 + 5, da6 = da0 + 6, da7 = da0 + 7; \
        unsigned a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;                              \
        double dx = (double)x0 * 1e-9, dy = (double)y0 * 1e-9;                                                \
        unsigned x = x0 + threadIdx.x;                                                                        \
        asm volatile("" : "+v"(dx), "+v"(dy), "+v"(x));                                                       \
        __syncthreads();                                                                                      \
        const unsigned long long w0 = wall_clock64();                                                         \
        const unsigned long long t0 = __builtin_readcyclecounter();                                           \
        for (int r = 0; r < REPS; r++) { BODY4(OP) }                                                          \
        const unsigned long long t1 = __builtin_readcyclecounter();                                           \
        const unsigned long long w1 = wall_clock64();                                                         \
        if (da0 + da1 + da2 + da3 + da4 + da5 + da6 + da7 + (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7) == 0.12345) s_pad[threadIdx.x] = 1; \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                     \
        if (threadIdx.x == 0) out[(size_t)gridDim.x * 16 + blockIdx.x] = w1 - w0;                             \
    }
KERNEL64(k_pk_fma_f32, OP_PK_FMA_F32)
KERNEL64(k_fma_f64, OP_FMA_F64)
KERNEL64(k_add_f64, OP_ADD_F64)
KERNEL64(k_cvt_f64_f32, OP_CVT_F64_F32)
KERNEL64(k_cvt_f32_f64, OP_CVT_F32_F64)

struct Entry {
    const char* name;
    void (*fn)(unsigned long long*, unsigned, unsigned);
    int per_op;   // instructions per OP expansion
};
#define E(n) {#n, n, 1}
static const Entry kEntries[] = {
    E(k_fma_f32), E(k_mul_f32), E(k_add_f32), E(k_add_u32), E(k_and_b32), E(k_lshl), E(k_ashr), E(k_mov), E(k_mad_u24), E(k_mad_i24),
    E(k_mul_lo), E(k_dot2_i16), E(k_dot2_i16_zero), E(k_dot2c_i16), E(k_dot4_i8), E(k_dot2_f16), E(k_mad_i32_i16_opsel), E(k_mad_i32_i16),
    E(k_perm), E(k_alignbit), E(k_bfe), E(k_pk_mad_i16), E(k_pk_mul_lo_u16), E(k_pk_add_u16), E(k_pk_fma_f16), E(k_cvt_f32_i32),
    E(k_cvt_i32_f32), E(k_floor_f32), E(k_rcp_f32), E(k_sqrt_f32), E(k_dpp_add), E(k_sad_u8), E(k_max_f32), E(k_cndmask), E(k_lshl_add),
    E(k_add3), E(k_mad_u16), E(k_mad_i16), E(k_mad_u32_u16), {"k_lkmix_dot2_perm", k_lkmix_dot2_perm, 2}, {"k_fma_then_dot2", k_fma_then_dot2, 2},
    E(k_dep_fma_f32), E(k_dep_add_u32), E(k_dep_mad_u24), E(k_dep_dot2_i16), E(k_dep_mad_i32_i16), E(k_dep_perm), E(k_dep_dot4_i8), E(k_dep_pk_mad_i16),
    E(k_pk_fma_f32), E(k_fma_f64), E(k_add_f64), E(k_cvt_f64_f32), E(k_cvt_f32_f64),
};

int main(int argc, char** argv) {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned long long* d_out;
    CHECK(hipMalloc(&d_out, (size_t)cus * 2 * 17 * sizeof(unsigned long long)));
    std::vector<unsigned long long> h((size_t)cus * 2 * 17);
    int wall_khz = 0;
    CHECK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"reps_x32\": %d, \"results\": [\n", prop.gcnArchName, cus, prop.clockRate / 1000, REPS);
    bool first = true;
    for (const Entry& e : kEntries) {
        if (argc > 1 && !strstr(e.name, argv[1])) continue;
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(e.fn), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        for (int cfg = 0; cfg < 6; cfg++) {
            const int W = cfg < 4 ? cfg + 1 : 4, B = cfg < 4 ? 1 : (cfg == 4 ? 2 : 2);
            if (cfg == 4) continue;   // (W, B): (1,1) (2,1) (3,1) (4,1) (4,2): 1-4 and 8 wavefronts per SIMD
            const size_t lds = B == 1 ? 96 * 1024 : 64 * 1024;
            const int WS = W * B;
            const double n_inst = (double)REPS * 32.0 * e.per_op;
            for (int rep = 0; rep < 3; rep++) {   // the last repetition counts (clocks up, code cached)
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(e.fn, dim3(cus * B), dim3(256 * W), lds, 0, d_out, 12345u + rep, 777u);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
            }
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(h.data(), d_out, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            double sum = 0;
            unsigned long long mx = 0;
            double wall = 0;
            for (int b = 0; b < cus * B; b++) {
                for (int w = 0; w < 4 * W; w++) {
                    sum += (double)h[(size_t)b * 16 + w];
                    mx = std::max(mx, h[(size_t)b * 16 + w]);
                }
                wall += (double)h[(size_t)cus * B * 16 + b];
            }
            const double avg = sum / (cus * B * 4.0 * W);
            const double wall_ns = wall / (cus * B) * 1e6 / wall_khz;   // one wavefront's loop, nanoseconds
            // s_memtime counts at a fixed 100 MHz on gfx9 parts: convert with the event time as well
            printf("%s  {\"op\": \"%s\", \"waves_per_simd\": %d, \"memtime_ticks_per_inst_per_simd\": %.4f, \"ns_per_inst_per_simd\": %.4f, "
                   "\"memtime_ghz\": %.4f, \"event_ms\": %.4f, \"event_ns_per_inst_per_simd\": %.4f}",
                   first ? "" : ",\n", e.name + 2, WS, avg / (n_inst * WS), wall_ns / (n_inst * WS), avg / wall_ns, ms,
                   (double)ms * 1e6 / (n_inst * WS));
            first = false;
        }
    }
    printf("\n]}\n");
    return 0;
}
