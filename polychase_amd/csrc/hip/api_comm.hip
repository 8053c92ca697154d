// api_comm.hip -- the C ABI of the multi-GPU stitch over RCCL (include/polychase_hip.h: pc_comm_*).
//
// One process per GPU.  The video-analysis path shards over frame1 ranges with no data-path collective;
// the ONE exchange is the stitch of the flow records: an all-gather of the ranks' device-log pieces (every rank ends up
// with every rank's records -- the collective BASELINE.json's north_star names) or an ordered gather to the rank that
// owns the SQLite file (send / recv).  These entry points give a C or C++ host both without torch: RCCL is called
// directly (ncclAllGather, ncclSend, ncclRecv).  librccl is resolved at the first pc_comm_* call (dlopen), so a
// single-GPU host -- Blender with the addon -- does not need it to be installed.  The reference has no counterpart: its
// frame loop, cpp/opticalflow.cc:209-321, is one process; what these calls replace is the in-process hand-over of
// opticalflow.cc:149-151 (flows -> database) when the frames were analysed on another GPU.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "api_internal.hpp"

using namespace pc_api;

namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // the copy a host has mapped already (torch bundles one under the same SONAME) is reused by the loader
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            r.error = std::string("librccl.so.1 could not be loaded: ") + (dlerror() ? dlerror() : "?");
            return;
        }
#define PC_RCCL_SYM(field, name)                                                     \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name));               \
    if (!r.field && r.error.empty()) r.error = std::string("librccl lacks ") + name;
        PC_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
        PC_RCCL_SYM(CommInitRank, "ncclCommInitRank")
        PC_RCCL_SYM(CommDestroy, "ncclCommDestroy")
        PC_RCCL_SYM(AllGather, "ncclAllGather")
        PC_RCCL_SYM(Send, "ncclSend")
        PC_RCCL_SYM(Recv, "ncclRecv")
        PC_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef PC_RCCL_SYM
    });
    return r;
}

}  // namespace

struct pc_comm {
    int device = 0, world = 1, rank = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;            // the collectives' own stream: never behind the analysis
    pc::DevBuf<unsigned long long> sizes;    // [world + 1]: this rank's size, then everybody's
    pc::DevBuf<uint8_t> staging;             // the padded copy of a piece (an all-gather wants equal counts)
};

#define PC_NCCL(expr)                                                                                          \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess) return fail(PC_E_HIP, "%s failed: %s", #expr, rccl().GetErrorString(r_));       \
    } while (0)

extern "C" {

int pc_comm_unique_id(void* id) {
    if (!id) return fail(PC_E_INVALID, "null id");
    Rccl& r = rccl();
    if (!r.error.empty()) return fail(PC_E_NO_DEVICE, "%s", r.error.c_str());
    static_assert(sizeof(ncclUniqueId) == PC_COMM_ID_BYTES, "PC_COMM_ID_BYTES is RCCL's ncclUniqueId");
    ncclUniqueId u;
    PC_NCCL(r.GetUniqueId(&u));
    std::memcpy(id, &u, sizeof(u));
    return PC_OK;
}

int pc_comm_create(pc_context* ctx, const void* id, int world_size, int rank, pc_comm** out) {
    if (!ctx || !id || !out) return fail(PC_E_INVALID, "null argument");
    *out = nullptr;
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail(PC_E_INVALID, "rank %d of %d", rank, world_size);
    Rccl& r = rccl();
    if (!r.error.empty()) return fail(PC_E_NO_DEVICE, "%s", r.error.c_str());
    PC_HIP(hipSetDevice(ctx->device));
    pc_comm* c = new (std::nothrow) pc_comm();
    if (!c) return fail(PC_E_INVALID, "out of host memory");
    c->device = ctx->device;
    c->world = world_size;
    c->rank = rank;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclResult_t nr = r.CommInitRank(&c->comm, world_size, u, rank);
    if (nr != ncclSuccess) {
        delete c;
        return fail(PC_E_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world_size, r.GetErrorString(nr));
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = c->sizes.ensure((size_t)world_size + 1);
    if (e != hipSuccess) {
        r.CommDestroy(c->comm);
        delete c;
        return fail(PC_E_HIP, "pc_comm_create: %s", hipGetErrorString(e));
    }
    *out = c;
    return PC_OK;
}

void pc_comm_destroy(pc_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) rccl().CommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    c->sizes.release();
    c->staging.release();
    delete c;
}

int pc_comm_world_size(const pc_comm* c) { return c ? c->world : -1; }
int pc_comm_rank(const pc_comm* c) { return c ? c->rank : -1; }

int pc_comm_all_gather_log(pc_comm* c, const void* piece, uint64_t bytes, void* recv, uint64_t slot_bytes, uint64_t* sizes_host) {
    if (!c || !recv || !sizes_host || (!piece && bytes)) return fail(PC_E_INVALID, "null argument");
    if (bytes > slot_bytes) return fail(PC_E_CAPACITY, "piece of %llu bytes does not fit the slot of %llu", (unsigned long long)bytes,
                                        (unsigned long long)slot_bytes);
    Rccl& r = rccl();
    PC_HIP(hipSetDevice(c->device));
    // 1. the sizes (one uint64 per rank), 2. the payload padded to the slot: two all-gathers on the comm's stream
    const unsigned long long mine = bytes;
    PC_HIP(hipMemcpyAsync(c->sizes.p, &mine, sizeof(mine), hipMemcpyHostToDevice, c->stream));
    PC_NCCL(r.AllGather(c->sizes.p, c->sizes.p + 1, 1, ncclUint64, c->comm, c->stream));
    PC_HIP(c->staging.ensure(slot_bytes));
    if (bytes) PC_HIP(hipMemcpyAsync(c->staging.p, piece, bytes, hipMemcpyDeviceToDevice, c->stream));
    PC_NCCL(r.AllGather(c->staging.p, recv, slot_bytes, ncclUint8, c->comm, c->stream));
    std::vector<unsigned long long> got((size_t)c->world);
    PC_HIP(hipMemcpyAsync(got.data(), c->sizes.p + 1, sizeof(unsigned long long) * c->world, hipMemcpyDeviceToHost, c->stream));
    PC_HIP(hipStreamSynchronize(c->stream));
    for (int k = 0; k < c->world; k++) sizes_host[k] = got[(size_t)k];
    return PC_OK;
}

int pc_comm_send(pc_comm* c, const void* src, uint64_t bytes, int dst) {
    if (!c || !src) return fail(PC_E_INVALID, "null argument");
    if (dst < 0 || dst >= c->world || dst == c->rank) return fail(PC_E_INVALID, "bad destination rank %d", dst);
    PC_HIP(hipSetDevice(c->device));
    PC_NCCL(rccl().Send(src, bytes, ncclUint8, dst, c->comm, c->stream));
    PC_HIP(hipStreamSynchronize(c->stream));
    return PC_OK;
}

int pc_comm_recv(pc_comm* c, void* dst_buf, uint64_t bytes, int src) {
    if (!c || !dst_buf) return fail(PC_E_INVALID, "null argument");
    if (src < 0 || src >= c->world || src == c->rank) return fail(PC_E_INVALID, "bad source rank %d", src);
    PC_HIP(hipSetDevice(c->device));
    PC_NCCL(rccl().Recv(dst_buf, bytes, ncclUint8, src, c->comm, c->stream));
    PC_HIP(hipStreamSynchronize(c->stream));
    return PC_OK;
}

}  // extern "C"
