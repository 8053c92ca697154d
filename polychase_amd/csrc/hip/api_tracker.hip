// api_tracker.hip -- C ABI of the tracking and refinement paths: meshes + LBVH ray casting, PnP accumulation
// (reference cpp/tracker.cc, cpp/pnp/*), and the refiner's per-edge sweeps (cpp/refiner.cc, cpp/pnp/lev_marq.h).
#include <chrono>
#include <mutex>
#include <limits>

#include "api_internal.hpp"
#include "pnp_lm.hpp"

using namespace pc_api;

// =============================================================================================
// tracker path: meshes, batched ray casting, PnP accumulation
// =============================================================================================

struct pc_mesh {
    pc_context* ctx = nullptr;
    int n_vertices = 0, n_triangles = 0;
    DevBuf<float> verts;
    DevBuf<uint32_t> tris, mask;
    std::vector<uint32_t> mask_sent;   // what `mask` holds (all zero after creation): an unchanged mask is not sent again
    // LBVH (bvh.hpp): n_triangles - 1 internal nodes + the sorted leaf order
    DevBuf<pc::BvhNode> bvh_nodes;
    DevBuf<uint32_t> bvh_leaf_tri;
    pc::BvhView bvh() const {
        pc::BvhView v;
        v.nodes = bvh_nodes.p;
        v.leaf_tri = bvh_leaf_tri.p;
        v.verts = verts.p;
        v.tris = tris.p;
        v.n_tris = n_triangles;
        return v;
    }
    // per-call scratch
    DevBuf<float2> d_xy;
    DevBuf<uint8_t> d_hit;
    DevBuf<float> d_pos, d_uvt;
    DevBuf<uint32_t> d_prim;
};

struct pc_pnp_problem {
    pc_context* ctx = nullptr;
    int n = 0;
    bool has_weights = false;
    // what the sweeps read and write: the problem's own buffers, or those of the pc_corr_set it was made from
    const float *X = nullptr, *x = nullptr, *w = nullptr;
    float *partials = nullptr, *out = nullptr, *h_out = nullptr;
    DevBuf<float> own_X, own_x, own_w, own_partials, own_out;
    PinBuf<float> own_h_out;
    // pc_pnp_solve: the solver's state on the device, its pinned mirror, partials of the inlier pass (own or the set's)
    pc::LmState *lm_state = nullptr, *lm_host = nullptr;
    float* lm_partials4 = nullptr;
    DevBuf<pc::LmState> own_lm_state;
    PinBuf<pc::LmState> own_lm_host;
    DevBuf<float> own_lm_partials4;
};

// 3D-2D correspondences of the frame being solved (include/polychase_hip.h)
struct pc_corr_set {
    pc_context* ctx = nullptr;
    DevBuf<float> X;             // world points, capacity x 3
    DevBuf<float2> x;            // image points
    size_t capacity = 0;
    int upper = 0;               // matches appended so far (>= the number of hits): the capacity the arrays need
    DevBuf<int> counter;         // [0] correspondences, [1] bad-index flag
    PinBuf<int> h_counter;
    // per-append scratch
    DevBuf<uint8_t> flag;
    DevBuf<float> world;
    DevBuf<int> block_counts, block_offsets;
    DevBuf<uint32_t> d_idx;
    DevBuf<float2> d_tgt;
    // device copies of recently used keypoint arrays
    struct Cached {
        long long key = -1;
        int n = 0;
        uint64_t stamp = 0;
        DevBuf<float2> xy;
    };
    Cached cache[32];   // two frames in flight + the one being uploaded, up to 8 sources each
    DevBuf<float2> uncached_kps[pc::kTrackMaxSources];
    uint64_t clock = 0;
    // PnP scratch shared by the problems made from this set
    DevBuf<float> partials, out, lm_partials4;
    PinBuf<float> h_out;
    DevBuf<pc::LmState> lm_state;
    PinBuf<pc::LmState> lm_host;
    // pc_track_solve_frame: all matches of the frame (device copies), their world points, the solver's barrier words
    DevBuf<uint8_t> t_block[3];   // the matches of the (up to two) frames in flight and of the one uploaded behind them
    int t_cur = 0;                // the block of the last pc_track_frame_upload
    size_t t_block_bytes = 0;
    hipStream_t copy_stream = nullptr;
    hipEvent_t upload_done = nullptr;
    // Up to TWO frames in flight, finished in launch order: the second one's launches sit behind the first one's on the stream and
    // may take the first one's camera from the device (pc_track_frame_launch_chained), so the GPU goes from one frame to the next
    // without waiting for the host to see a pose.
    struct Flight {
        int n = 0;                // matches (0: nothing was launched, finish reports "no correspondences")
        uint32_t seq = 0;         // TrackLmOut::seq of its LM launch; the result is in t_out[seq & 1]
    };
    Flight t_flight[2];
    int t_inflight = 0;           // entries of t_flight in use, oldest first
    bool t_chain_valid = false;   // the launch enqueued last wrote t_chain
    DevBuf<pc::TrackChainSlot> t_chain;
    DevBuf<uint32_t> t_sync;
    DevBuf<float2> t_obs;
    DevBuf<float4> t_pts;
    DevBuf<float> t_partials;
    PinBuf<pc::TrackLmOut> t_out;   // two: launch `seq` reports into t_out[seq & 1]
    uint32_t t_seq = 0;   // number of the LM launch enqueued last (TrackLmOut::seq)
    bool t_sync_zero = false;
};

extern "C" {

int pc_mesh_create(pc_context* ctx, const float* vertices, int n_vertices, const uint32_t* triangles,
                   int n_triangles, pc_mesh** out) {
    if (!ctx || !out || n_vertices < 0 || n_triangles < 0 || (n_vertices > 0 && !vertices) ||
        (n_triangles > 0 && !triangles))
        return fail(PC_E_INVALID, "bad argument");
    *out = nullptr;
    for (int i = 0; i < 3 * n_triangles; i++)
        if (triangles[i] >= (uint32_t)n_vertices) return fail(PC_E_INVALID, "triangle index %u out of range", triangles[i]);
    PC_HIP(hipSetDevice(ctx->device));
    pc_mesh* m = new (std::nothrow) pc_mesh();
    if (!m) return fail(PC_E_INVALID, "out of host memory");
    m->ctx = ctx;
    m->n_vertices = n_vertices;
    m->n_triangles = n_triangles;
    m->mask_sent.assign((size_t)((n_triangles + 31) / 32), 0u);
    const int words = (n_triangles + 31) / 32 + 4;
    hipError_t e = m->verts.ensure((size_t)std::max(1, n_vertices) * 3);
    if (e == hipSuccess) e = m->tris.ensure((size_t)std::max(1, n_triangles) * 3);
    if (e == hipSuccess) e = m->mask.ensure((size_t)words);
    if (e == hipSuccess && n_vertices) e = hipMemcpyAsync(m->verts.p, vertices, (size_t)n_vertices * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && n_triangles) e = hipMemcpyAsync(m->tris.p, triangles, (size_t)n_triangles * 3 * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(m->mask.p, 0, (size_t)words * sizeof(uint32_t), ctx->stream);
    // acceleration structure (rtcCommitScene in the reference, ray_casting.cc:23-63): LBVH built on the GPU
    if (e == hipSuccess && n_triangles > 0) {
        const size_t n = (size_t)n_triangles;
        float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        for (int v = 0; v < n_vertices; v++)
            for (int k = 0; k < 3; k++) {
                const float x = vertices[3 * (size_t)v + k];
                if (v == 0 || x < lo[k]) lo[k] = x;
                if (v == 0 || x > hi[k]) hi[k] = x;
            }
        const float extent = std::max(std::max(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
        const float pad = 1e-5f * extent + 1e-30f;
        DevBuf<unsigned long long> keys_in, keys_out;
        DevBuf<float> box_lo, box_hi;
        DevBuf<int> links;      // parent (2n-1) | visits (n) | left (n) | right (n)
        DevBuf<uint32_t> bounds;
        DevBuf<uint8_t> sort_temp;
        const size_t temp_bytes = pc::bvh_sort_temp_bytes(n_triangles);
        e = m->bvh_nodes.ensure(n);
        if (e == hipSuccess) e = m->bvh_leaf_tri.ensure(n);
        if (e == hipSuccess) e = keys_in.ensure(n);
        if (e == hipSuccess) e = keys_out.ensure(n);
        if (e == hipSuccess) e = box_lo.ensure(3 * (2 * n));
        if (e == hipSuccess) e = box_hi.ensure(3 * (2 * n));
        if (e == hipSuccess) e = links.ensure(5 * n + 8);
        if (e == hipSuccess) e = bounds.ensure(8);
        if (e == hipSuccess) e = sort_temp.ensure(temp_bytes + 16);
        if (e == hipSuccess) {
            pc::BvhBuildScratch sc;
            sc.keys_in = keys_in.p;
            sc.keys_out = keys_out.p;
            sc.box_lo = box_lo.p;
            sc.box_hi = box_hi.p;
            sc.parent = links.p;
            sc.visits = links.p + 2 * n;
            sc.left = links.p + 3 * n;
            sc.right = links.p + 4 * n;
            sc.bounds = bounds.p;
            e = pc::bvh_build(m->verts.p, m->tris.p, n_triangles, pad, sc, sort_temp.p, temp_bytes, m->bvh_nodes.p,
                              m->bvh_leaf_tri.p, ctx->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        keys_in.release();
        keys_out.release();
        box_lo.release();
        box_hi.release();
        links.release();
        bounds.release();
        sort_temp.release();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        pc_mesh_destroy(m);
        return fail(PC_E_HIP, "mesh upload failed: %s", hipGetErrorString(e));
    }
    *out = m;
    return PC_OK;
}

int pc_mesh_set_mask(pc_context* ctx, pc_mesh* mesh, const uint32_t* mask_words, int n_words) {
    if (!ctx || !mesh || !mask_words) return fail(PC_E_INVALID, "null argument");
    const int need = (mesh->n_triangles + 31) / 32;
    if (n_words < need) return fail(PC_E_INVALID, "mask has %d words, %d needed", n_words, need);
    if (need > 0) {
        // the tracker sends the mask before every frame (it can be edited through inner_mut() at any time): a transfer and
        // a wait only when a bit has changed
        if (mesh->mask_sent.size() == (size_t)need && std::memcmp(mesh->mask_sent.data(), mask_words, (size_t)need * sizeof(uint32_t)) == 0)
            return PC_OK;
        PC_HIP(hipSetDevice(ctx->device));
        PC_HIP(hipMemcpyAsync(mesh->mask.p, mask_words, (size_t)need * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        PC_HIP(hipStreamSynchronize(ctx->stream));
        mesh->mask_sent.assign(mask_words, mask_words + need);
    }
    return PC_OK;
}

void pc_mesh_destroy(pc_mesh* m) {
    if (!m) return;
    if (m->ctx) {
        (void)hipSetDevice(m->ctx->device);
        (void)hipStreamSynchronize(m->ctx->stream);
    }
    m->verts.release();
    m->tris.release();
    m->mask.release();
    m->bvh_nodes.release();
    m->bvh_leaf_tri.release();
    m->d_xy.release();
    m->d_hit.release();
    m->d_pos.release();
    m->d_uvt.release();
    m->d_prim.release();
    delete m;
}

static int raycast_pixels(pc_context* ctx, const pc_mesh* mesh_c, const pc_ray_camera* cam, const float* xy, int n,
                          int check_mask, uint8_t* hit, float* pos, uint32_t* prim, float* uvt, bool sweep) {
    if (!ctx || !mesh_c || !cam || n < 0) return fail(PC_E_INVALID, "bad argument");
    if (n == 0) return PC_OK;
    if (!xy || !hit || !pos || !prim || !uvt) return fail(PC_E_INVALID, "null buffer");
    pc_mesh* mesh = const_cast<pc_mesh*>(mesh_c);
    PC_HIP(hipSetDevice(ctx->device));
    PC_HIP(mesh->d_xy.ensure((size_t)n));
    PC_HIP(mesh->d_hit.ensure((size_t)n));
    PC_HIP(mesh->d_pos.ensure((size_t)n * 3));
    PC_HIP(mesh->d_uvt.ensure((size_t)n * 3));
    PC_HIP(mesh->d_prim.ensure((size_t)n));
    pc::RayCamera rc;
    std::memcpy(rc.m, cam->dir_matrix, sizeof(rc.m));
    std::memcpy(rc.origin, cam->origin, sizeof(rc.origin));
    rc.fx = cam->fx;
    rc.fy = cam->fy;
    rc.cx = cam->cx;
    rc.cy = cam->cy;
    rc.sign = cam->unproject_sign;
    PC_HIP(hipMemcpyAsync(mesh->d_xy.p, xy, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));
    if (sweep)
        pc::launch_raycast_sweep(mesh->verts.p, mesh->tris.p, mesh->n_triangles, mesh->mask.p, check_mask, rc, mesh->d_xy.p, n,
                                 mesh->d_hit.p, mesh->d_pos.p, mesh->d_prim.p, mesh->d_uvt.p, ctx->stream);
    else
        pc::launch_raycast(mesh->bvh(), mesh->mask.p, check_mask, rc, mesh->d_xy.p, n, mesh->d_hit.p, mesh->d_pos.p,
                           mesh->d_prim.p, mesh->d_uvt.p, ctx->stream);
    PC_HIP(hipMemcpyAsync(hit, mesh->d_hit.p, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    PC_HIP(hipMemcpyAsync(pos, mesh->d_pos.p, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    PC_HIP(hipMemcpyAsync(prim, mesh->d_prim.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    PC_HIP(hipMemcpyAsync(uvt, mesh->d_uvt.p, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    PC_HIP(hipStreamSynchronize(ctx->stream));
    return PC_OK;
}

int pc_raycast_pixels(pc_context* ctx, const pc_mesh* mesh, const pc_ray_camera* cam, const float* xy, int n, int check_mask,
                      uint8_t* hit, float* pos, uint32_t* prim, float* uvt) {
    return raycast_pixels(ctx, mesh, cam, xy, n, check_mask, hit, pos, prim, uvt, false);
}

int pc_raycast_pixels_sweep(pc_context* ctx, const pc_mesh* mesh, const pc_ray_camera* cam, const float* xy, int n,
                            int check_mask, uint8_t* hit, float* pos, uint32_t* prim, float* uvt) {
    return raycast_pixels(ctx, mesh, cam, xy, n, check_mask, hit, pos, prim, uvt, true);
}

// ---- correspondence set ----------------------------------------------------------------------
int pc_corr_set_create(pc_context* ctx, pc_corr_set** out) {
    if (!ctx || !out) return fail(PC_E_INVALID, "null argument");
    *out = nullptr;
    PC_HIP(hipSetDevice(ctx->device));
    pc_corr_set* s = new (std::nothrow) pc_corr_set();
    if (!s) return fail(PC_E_INVALID, "out of host memory");
    s->ctx = ctx;
    hipError_t e = s->counter.ensure(2);
    if (e == hipSuccess) e = s->h_counter.ensure(2);
    if (e == hipSuccess) e = s->out.ensure(64);
    if (e == hipSuccess) e = s->h_out.ensure(64);
    if (e == hipSuccess) e = s->lm_state.ensure(1);
    if (e == hipSuccess) e = s->lm_host.ensure(1);
    if (e == hipSuccess) e = hipMemsetAsync(s->counter.p, 0, 2 * sizeof(int), ctx->stream);
    if (e != hipSuccess) {
        pc_corr_set_destroy(s);
        return fail(PC_E_HIP, "correspondence set: %s", hipGetErrorString(e));
    }
    *out = s;
    return PC_OK;
}

void pc_corr_set_destroy(pc_corr_set* s) {
    if (!s) return;
    if (s->ctx) {
        (void)hipSetDevice(s->ctx->device);
        (void)hipStreamSynchronize(s->ctx->stream);
        if (s->copy_stream) (void)hipStreamSynchronize(s->copy_stream);
    }
    s->X.release();
    s->x.release();
    s->counter.release();
    s->h_counter.release();
    s->flag.release();
    s->world.release();
    s->block_counts.release();
    s->block_offsets.release();
    s->d_idx.release();
    s->d_tgt.release();
    for (auto& c : s->cache) c.xy.release();
    for (auto& b : s->uncached_kps) b.release();
    s->partials.release();
    s->out.release();
    s->h_out.release();
    s->lm_partials4.release();
    s->lm_state.release();
    s->lm_host.release();
    for (auto& b : s->t_block) b.release();
    if (s->copy_stream) (void)hipStreamDestroy(s->copy_stream);
    if (s->upload_done) (void)hipEventDestroy(s->upload_done);
    s->t_sync.release();
    s->t_obs.release();
    s->t_pts.release();
    s->t_partials.release();
    s->t_out.release();
    s->t_chain.release();
    delete s;
}

int pc_corr_set_clear(pc_context* ctx, pc_corr_set* s) {
    if (!ctx || !s) return fail(PC_E_INVALID, "null argument");
    PC_HIP(hipSetDevice(ctx->device));
    PC_HIP(hipMemsetAsync(s->counter.p, 0, 2 * sizeof(int), ctx->stream));
    s->upper = 0;
    return PC_OK;
}

int pc_corr_set_recycle(pc_context* ctx, pc_corr_set* s) {
    if (!ctx || !s) return fail(PC_E_INVALID, "null argument");
    if (s->t_inflight > 0) return fail(PC_E_STATE, "a frame is in flight: pc_track_frame_finish first");
    PC_HIP(hipSetDevice(ctx->device));
    PC_HIP(hipStreamSynchronize(ctx->stream));
    if (s->copy_stream) PC_HIP(hipStreamSynchronize(s->copy_stream));
    for (auto& c : s->cache) {
        c.key = -1;
        c.n = 0;
        c.stamp = 0;
    }
    s->clock = 0;
    s->t_chain_valid = false;
    s->t_block_bytes = 0;
    return pc_corr_set_clear(ctx, s);
}

// grow X / x to `need` correspondences, keeping what they hold
static int corr_reserve(pc_context* ctx, pc_corr_set* s, size_t need) {
    if (need <= s->capacity) return PC_OK;
    const size_t want = need + need / 2 + 4096;
    DevBuf<float> nX;
    DevBuf<float2> nx;
    PC_HIP(nX.ensure(want * 3));
    hipError_t e = nx.ensure(want);
    if (e != hipSuccess) {
        nX.release();
        return fail(PC_E_HIP, "correspondence set: %s", hipGetErrorString(e));
    }
    if (s->capacity) {   // at most `capacity` entries are live; stream order puts the copies behind the appends
        (void)hipMemcpyAsync(nX.p, s->X.p, s->capacity * 3 * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream);
        (void)hipMemcpyAsync(nx.p, s->x.p, s->capacity * sizeof(float2), hipMemcpyDeviceToDevice, ctx->stream);
        PC_HIP(hipStreamSynchronize(ctx->stream));
    }
    s->X.release();
    s->x.release();
    s->X = nX;
    s->x = nx;
    s->capacity = want;
    return PC_OK;
}

// device copy of a source frame's keypoints: cached by key (a frame is a source for up to 8 targets), else uploaded
static int corr_keypoints(pc_context* ctx, pc_corr_set* s, long long keypoints_key, const float* keypoints_xy, int n_keypoints,
                          const float2** out, int uncached_slot = 0, hipStream_t stream = nullptr) {
    if (!stream) stream = ctx->stream;
    const float2* d_kps = nullptr;
    if (keypoints_key >= 0) {
        pc_corr_set::Cached* slot = nullptr;
        for (auto& c : s->cache)
            if (c.key == keypoints_key && c.n == n_keypoints) slot = &c;
        if (!slot) {
            slot = &s->cache[0];
            for (auto& c : s->cache)
                if (c.stamp < slot->stamp) slot = &c;
            PC_HIP(slot->xy.ensure((size_t)std::max(n_keypoints, 1)));
            PC_HIP(hipMemcpyAsync(slot->xy.p, keypoints_xy, (size_t)n_keypoints * sizeof(float2), hipMemcpyHostToDevice, stream));
            slot->key = keypoints_key;
            slot->n = n_keypoints;
        }
        slot->stamp = ++s->clock;
        d_kps = slot->xy.p;
    } else {
        DevBuf<float2>& buf = s->uncached_kps[uncached_slot];
        PC_HIP(buf.ensure((size_t)std::max(n_keypoints, 1)));
        PC_HIP(hipMemcpyAsync(buf.p, keypoints_xy, (size_t)n_keypoints * sizeof(float2), hipMemcpyHostToDevice, stream));
        d_kps = buf.p;
    }
    *out = d_kps;
    return PC_OK;
}

int pc_corr_set_append(pc_context* ctx, pc_corr_set* s, const pc_mesh* mesh, const pc_ray_camera* cam, const float* model_matrix,
                       long long keypoints_key, const float* keypoints_xy, int n_keypoints, const uint32_t* src_idx,
                       const float* tgt_xy, int n_matches, int check_mask) {
    if (!ctx || !s || !mesh || !cam || !model_matrix || n_matches < 0 || n_keypoints < 0) return fail(PC_E_INVALID, "bad argument");
    if (n_matches == 0) return PC_OK;
    if (!keypoints_xy || !src_idx || !tgt_xy) return fail(PC_E_INVALID, "null buffer");
    PC_HIP(hipSetDevice(ctx->device));
    int rc = corr_reserve(ctx, s, (size_t)s->upper + (size_t)n_matches);
    if (rc != PC_OK) return rc;
    const int nb = pc::corr_num_blocks(n_matches);
    PC_HIP(s->flag.ensure((size_t)n_matches));
    PC_HIP(s->world.ensure((size_t)n_matches * 3));
    PC_HIP(s->block_counts.ensure((size_t)nb));
    PC_HIP(s->block_offsets.ensure((size_t)nb));
    PC_HIP(s->d_idx.ensure((size_t)n_matches));
    PC_HIP(s->d_tgt.ensure((size_t)n_matches));
    const float2* d_kps = nullptr;
    rc = corr_keypoints(ctx, s, keypoints_key, keypoints_xy, n_keypoints, &d_kps);
    if (rc != PC_OK) return rc;
    PC_HIP(hipMemcpyAsync(s->d_idx.p, src_idx, (size_t)n_matches * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    PC_HIP(hipMemcpyAsync(s->d_tgt.p, tgt_xy, (size_t)n_matches * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));
    pc::RayCamera rcam;
    std::memcpy(rcam.m, cam->dir_matrix, sizeof(rcam.m));
    std::memcpy(rcam.origin, cam->origin, sizeof(rcam.origin));
    rcam.fx = cam->fx;
    rcam.fy = cam->fy;
    rcam.cx = cam->cx;
    rcam.cy = cam->cy;
    rcam.sign = cam->unproject_sign;
    pc::CorrModel mm;
    std::memcpy(mm.m, model_matrix, sizeof(mm.m));
    pc::launch_corr_append(mesh->bvh(), mesh->mask.p, check_mask, rcam, mm, d_kps, n_keypoints, s->d_idx.p, s->d_tgt.p, n_matches,
                           s->flag.p, s->world.p, s->block_counts.p, s->block_offsets.p, s->counter.p, s->counter.p + 1, s->X.p,
                           s->x.p, ctx->stream);
    s->upper += n_matches;
    return PC_OK;
}

int pc_corr_set_size(pc_context* ctx, pc_corr_set* s, int* n) {
    if (!ctx || !s || !n) return fail(PC_E_INVALID, "null argument");
    PC_HIP(hipSetDevice(ctx->device));
    PC_HIP(hipMemcpyAsync(s->h_counter.p, s->counter.p, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PC_HIP(hipStreamSynchronize(ctx->stream));
    if (s->h_counter.p[1]) return fail(PC_E_INVALID, "a source keypoint index is out of range");
    *n = s->h_counter.p[0];
    return PC_OK;
}

int pc_corr_set_download(pc_context* ctx, pc_corr_set* s, float* world_xyz, float* image_xy) {
    int n = 0;
    int rc = pc_corr_set_size(ctx, s, &n);
    if (rc != PC_OK) return rc;
    if (n == 0) return PC_OK;
    if (world_xyz) PC_HIP(hipMemcpyAsync(world_xyz, s->X.p, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (image_xy) PC_HIP(hipMemcpyAsync(image_xy, s->x.p, (size_t)n * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream));
    PC_HIP(hipStreamSynchronize(ctx->stream));
    return PC_OK;
}

int pc_pnp_problem_from_set(pc_context* ctx, pc_corr_set* s, pc_pnp_problem** out) {
    if (!ctx || !s || !out) return fail(PC_E_INVALID, "null argument");
    *out = nullptr;
    int n = 0;
    int rc = pc_corr_set_size(ctx, s, &n);
    if (rc != PC_OK) return rc;
    if (n < 1) return fail(PC_E_INVALID, "the correspondence set is empty");
    PC_HIP(s->partials.ensure((size_t)pc::pnp_num_blocks(n) * 56));
    PC_HIP(s->lm_partials4.ensure((size_t)pc::pnp_num_blocks(n) * 4));
    pc_pnp_problem* p = new (std::nothrow) pc_pnp_problem();
    if (!p) return fail(PC_E_INVALID, "out of host memory");
    p->ctx = ctx;
    p->n = n;
    p->X = s->X.p;
    p->x = reinterpret_cast<const float*>(s->x.p);
    p->partials = s->partials.p;
    p->out = s->out.p;
    p->h_out = s->h_out.p;
    p->lm_state = s->lm_state.p;
    p->lm_host = s->lm_host.p;
    p->lm_partials4 = s->lm_partials4.p;
    *out = p;
    return PC_OK;
}

int pc_pnp_problem_create(pc_context* ctx, const float* X, const float* x, const float* weights, int n,
                          pc_pnp_problem** out) {
    if (!ctx || !out || n < 1 || !X || !x) return fail(PC_E_INVALID, "bad argument");
    *out = nullptr;
    PC_HIP(hipSetDevice(ctx->device));
    pc_pnp_problem* p = new (std::nothrow) pc_pnp_problem();
    if (!p) return fail(PC_E_INVALID, "out of host memory");
    p->ctx = ctx;
    p->n = n;
    p->has_weights = weights != nullptr;
    const int nb = pc::pnp_num_blocks(n);
    hipError_t e = p->own_X.ensure((size_t)n * 3);
    if (e == hipSuccess) e = p->own_x.ensure((size_t)n * 2);
    if (e == hipSuccess && weights) e = p->own_w.ensure((size_t)n);
    if (e == hipSuccess) e = p->own_partials.ensure((size_t)nb * 56);
    if (e == hipSuccess) e = p->own_out.ensure(64);
    if (e == hipSuccess) e = p->own_h_out.ensure(64);
    if (e == hipSuccess) e = hipMemcpyAsync(p->own_X.p, X, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(p->own_x.p, x, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && weights)
        e = hipMemcpyAsync(p->own_w.p, weights, (size_t)n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    p->X = p->own_X.p;
    p->x = p->own_x.p;
    p->w = p->own_w.p;
    p->partials = p->own_partials.p;
    p->out = p->own_out.p;
    p->h_out = p->own_h_out.p;
    if (e != hipSuccess) {
        pc_pnp_problem_destroy(p);
        return fail(PC_E_HIP, "PnP upload failed: %s", hipGetErrorString(e));
    }
    *out = p;
    return PC_OK;
}

void pc_pnp_problem_destroy(pc_pnp_problem* p) {
    if (!p) return;
    if (p->ctx) {
        (void)hipSetDevice(p->ctx->device);
        (void)hipStreamSynchronize(p->ctx->stream);
    }
    p->own_X.release();
    p->own_x.release();
    p->own_w.release();
    p->own_partials.release();
    p->own_out.release();
    p->own_h_out.release();
    p->own_lm_state.release();
    p->own_lm_host.release();
    p->own_lm_partials4.release();
    delete p;
}

static pc::PnPParams to_kernel_params(const pc_pnp_params* q) {
    pc::PnPParams p;
    std::memcpy(p.R, q->R, sizeof(p.R));
    std::memcpy(p.t, q->t, sizeof(p.t));
    p.fx = q->fx;
    p.fy = q->fy;
    p.cx = q->cx;
    p.cy = q->cy;
    p.aspect_ratio = q->aspect_ratio;
    p.convention_opencv = q->convention_opencv;
    p.optimize_focal = q->optimize_focal_length;
    p.optimize_pp = q->optimize_principal_point;
    p.loss_type = q->loss_type;
    p.loss_scale = q->loss_scale;
    return p;
}

int pc_pnp_normal_equations(pc_context* ctx, const pc_pnp_problem* prob, const pc_pnp_params* params,
                            float* jtj_lower45, float* jtr9, int* valid) {
    return pc_pnp_normal_equations_cost(ctx, prob, params, jtj_lower45, jtr9, valid, nullptr);
}

int pc_pnp_normal_equations_cost(pc_context* ctx, const pc_pnp_problem* prob, const pc_pnp_params* params,
                                 float* jtj_lower45, float* jtr9, int* valid, float* cost) {
    if (!ctx || !prob || !params || !jtj_lower45 || !jtr9) return fail(PC_E_INVALID, "null argument");
    if (params->loss_type < 0 || params->loss_type > 2) return fail(PC_E_INVALID, "Unknown loss type: %d", params->loss_type);
    PC_HIP(hipSetDevice(ctx->device));
    pc::launch_pnp_normal_eq(prob->X, prob->x, prob->has_weights ? prob->w : nullptr, prob->n,
                             to_kernel_params(params), prob->partials, prob->out, ctx->stream);
    PC_HIP(hipMemcpyAsync(prob->h_out, prob->out, 56 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    PC_HIP(hipStreamSynchronize(ctx->stream));
    std::memcpy(jtj_lower45, prob->h_out, 45 * sizeof(float));
    std::memcpy(jtr9, prob->h_out + 45, 9 * sizeof(float));
    if (valid) *valid = (int)prob->h_out[54];
    if (cost) *cost = prob->h_out[55];
    return PC_OK;
}

static void fill_lm_config(const pc_pnp_solve_options* o, pc::LmConfig* c) {
    c->max_iterations = o->max_iterations;
    c->initial_lambda = o->initial_lambda;
    c->min_lambda = o->min_lambda;
    c->max_lambda = o->max_lambda;
    c->gradient_tol = o->gradient_tol;
    c->step_tol = o->step_tol;
    c->f_low = o->f_low;
    c->f_high = o->f_high;
    c->cx_low = o->cx_low;
    c->cx_high = o->cx_high;
    c->cy_low = o->cy_low;
    c->cy_high = o->cy_high;
    c->optimize_focal = o->optimize_focal_length ? 1 : 0;
    c->optimize_pp = o->optimize_principal_point ? 1 : 0;
    c->loss_type = o->loss_type;
    c->loss_scale = o->loss_scale;
    c->max_inlier_err_sq = o->max_inlier_error > 0.0f ? o->max_inlier_error * o->max_inlier_error : 0.0f;
}
static void fill_lm_camera(const pc_pnp_camera* in, pc::LmCamera* c) {
    c->qx = in->q_xyzw[0];
    c->qy = in->q_xyzw[1];
    c->qz = in->q_xyzw[2];
    c->qw = in->q_xyzw[3];
    for (int i = 0; i < 3; i++) c->t[i] = in->t[i];
    c->fx = in->fx;
    c->fy = in->fy;
    c->cx = in->cx;
    c->cy = in->cy;
    c->aspect_ratio = in->aspect_ratio;
    c->convention_opencv = in->convention_opencv;
}
static void read_lm_camera(const pc::LmCamera& c, pc_pnp_camera* out) {
    out->q_xyzw[0] = c.qx;
    out->q_xyzw[1] = c.qy;
    out->q_xyzw[2] = c.qz;
    out->q_xyzw[3] = c.qw;
    for (int i = 0; i < 3; i++) out->t[i] = c.t[i];
    out->fx = c.fx;
    out->fy = c.fy;
    out->cx = c.cx;
    out->cy = c.cy;
    out->aspect_ratio = c.aspect_ratio;
    out->convention_opencv = c.convention_opencv;
}

int pc_pnp_solve(pc_context* ctx, pc_pnp_problem* prob, const pc_pnp_camera* initial, const pc_pnp_solve_options* o,
                 pc_pnp_solve_result* result) {
    if (!ctx || !prob || !initial || !o || !result) return fail(PC_E_INVALID, "null argument");
    if (o->loss_type < 0 || o->loss_type > 2) return fail(PC_E_INVALID, "Unknown loss type: %d", o->loss_type);
    PC_HIP(hipSetDevice(ctx->device));
    const int nb = pc::pnp_num_blocks(prob->n);
    if (!prob->lm_state) {   // a stand-alone problem: its own solver scratch, on first use
        PC_HIP(prob->own_lm_state.ensure(1));
        PC_HIP(prob->own_lm_host.ensure(1));
        PC_HIP(prob->own_lm_partials4.ensure((size_t)nb * 4));
        prob->lm_state = prob->own_lm_state.p;
        prob->lm_host = prob->own_lm_host.p;
        prob->lm_partials4 = prob->own_lm_partials4.p;
    }
    pc::LmState& h = *prob->lm_host;
    std::memset(&h, 0, sizeof(h));
    fill_lm_config(o, &h.cfg);
    fill_lm_camera(initial, &h.cam);
    h.cam_new = h.cam;
    h.lambda = o->initial_lambda;
    h.v = 2.0f;
    h.grad_norm = -1.0f;
    h.step_norm = -1.0f;
    h.rebuild = 1;
    pc::lm_make_params(h.cam, h.cfg, &h.sweep);   // phase 0: evaluate the initial parameters
    PC_HIP(hipMemcpyAsync(prob->lm_state, &h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
    // every round evaluates one parameter set; the first one is the initial set, Cholesky failures take no round
    int rounds = o->rounds_hint > 0 ? o->rounds_hint : 13;
    int enqueued = 0;
    const int limit = o->max_iterations + 2;
    const float* w = prob->has_weights ? prob->w : nullptr;
    for (;;) {
        rounds = std::max(1, std::min(rounds, limit - enqueued));
        pc::launch_pnp_lm_rounds(prob->X, prob->x, w, prob->n, prob->lm_state, rounds, prob->partials, prob->lm_partials4,
                                 prob->out + 56, ctx->stream);
        enqueued += rounds;
        PC_HIP(hipMemcpyAsync(&h, prob->lm_state, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        PC_HIP(hipMemcpyAsync(prob->h_out, prob->out + 56, 4 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        PC_HIP(hipStreamSynchronize(ctx->stream));
        if (h.done) break;
        if (enqueued >= limit) return fail(PC_E_STATE, "the PnP solver did not finish within %d rounds", enqueued);
        rounds = 6;
    }
    read_lm_camera(h.cam, &result->camera);
    result->iterations = h.iterations;
    result->invalid_steps = h.invalid_steps;
    result->initial_cost = h.initial_cost;
    result->cost = h.cost;
    result->lambda = h.lambda;
    result->step_norm = h.step_norm;
    result->grad_norm = h.grad_norm;
    result->inliers = o->max_inlier_error > 0.0f ? (int)prob->h_out[2] : 0;
    return PC_OK;
}

static void to_ray_camera(const pc_ray_camera* cam, pc::RayCamera* rc) {
    std::memcpy(rc->m, cam->dir_matrix, sizeof(rc->m));
    std::memcpy(rc->origin, cam->origin, sizeof(rc->origin));
    rc->fx = cam->fx;
    rc->fy = cam->fy;
    rc->cx = cam->cx;
    rc->cy = cam->cy;
    rc->sign = cam->unproject_sign;
}

static int check_track_sources(const pc_track_source* sources, int n_sources, size_t matches_bytes, size_t* total_out) {
    if (n_sources < 0 || (n_sources > 0 && !sources)) return fail(PC_E_INVALID, "bad argument");
    if (n_sources > pc::kTrackMaxSources) return fail(PC_E_INVALID, "%d source frames, at most %d", n_sources, pc::kTrackMaxSources);
    size_t total = 0;
    for (int k = 0; k < n_sources; k++) {
        const pc_track_source& q = sources[k];
        if (q.n_matches < 0 || q.n_keypoints < 0 || (q.n_matches > 0 && !q.keypoints_xy)) return fail(PC_E_INVALID, "bad source %d", k);
        const size_t rows = (size_t)q.n_matches;
        if (rows && ((q.idx_offset & 3u) || (q.tgt_offset & 7u) || q.idx_offset > matches_bytes || rows * 4 > matches_bytes - q.idx_offset ||
                     q.tgt_offset > matches_bytes || rows * 8 > matches_bytes - q.tgt_offset))
            return fail(PC_E_INVALID, "the matches of source %d do not lie inside the block", k);
        total += rows;
    }
    if (total > (size_t)1 << 30) return fail(PC_E_INVALID, "too many matches");
    *total_out = total;
    return PC_OK;
}

int pc_corr_set_reserve(pc_context* ctx, pc_corr_set* s, int n_matches, int n_keypoints, int n_cached) {
    if (!ctx || !s || n_matches < 0 || n_keypoints < 0 || n_cached < 0) return fail(PC_E_INVALID, "bad argument");
    if (s->t_inflight > 0) return fail(PC_E_STATE, "a frame is in flight: pc_track_frame_finish first");
    PC_HIP(hipSetDevice(ctx->device));
    if (!s->copy_stream) {
        PC_HIP(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
        PC_HIP(hipEventCreateWithFlags(&s->upload_done, hipEventDisableTiming));
    }
    const size_t total = (size_t)n_matches;
    for (auto& b : s->t_block) PC_HIP(b.ensure(total * 12 + 16 * 2 * pc::kTrackMaxSources + 16));   // per source: uint32 indices + float2 targets, 16-byte aligned
    PC_HIP(s->t_obs.ensure(total));
    PC_HIP(s->t_pts.ensure(total));
    PC_HIP(s->t_partials.ensure((size_t)pc::track_lm_blocks(n_matches) * 56));
    PC_HIP(s->t_out.ensure(2));
    PC_HIP(s->t_chain.ensure(1));
    PC_HIP(s->t_sync.ensure((size_t)pc::kTrackSyncWords));
    if (!s->t_sync_zero) {
        PC_HIP(hipMemsetAsync(s->t_sync.p, 0, pc::kTrackSyncWords * sizeof(uint32_t), ctx->stream));
        s->t_sync_zero = true;
    }
    int k = 0;
    for (auto& c : s->cache)
        if (k++ < n_cached) PC_HIP(c.xy.ensure((size_t)std::max(n_keypoints, 1)));
    PC_HIP(hipStreamSynchronize(ctx->stream));
    return PC_OK;
}

int pc_track_frame_upload(pc_context* ctx, pc_corr_set* s, const void* matches, size_t matches_bytes, const pc_track_source* sources,
                          int n_sources) {
    if (!ctx || !s || (matches_bytes > 0 && !matches)) return fail(PC_E_INVALID, "bad argument");
    size_t total = 0;
    int rc = check_track_sources(sources, n_sources, matches_bytes, &total);
    if (rc != PC_OK) return rc;
    PC_HIP(hipSetDevice(ctx->device));
    if (!s->copy_stream) {
        PC_HIP(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
        PC_HIP(hipEventCreateWithFlags(&s->upload_done, hipEventDisableTiming));
    }
    // the other block: the launch that reads the current one may still be running
    s->t_cur = (s->t_cur + 1) % 3;
    s->t_block_bytes = matches_bytes;
    PC_HIP(s->t_block[s->t_cur].ensure(std::max<size_t>(matches_bytes, 16)));
    if (matches_bytes) PC_HIP(hipMemcpyAsync(s->t_block[s->t_cur].p, matches, matches_bytes, hipMemcpyHostToDevice, s->copy_stream));
    // the keypoints of sources the set has not seen yet (normally ONE: the frame solved last), into the set's cache
    for (int k = 0; k < n_sources; k++) {
        if (sources[k].keypoints_key < 0 || sources[k].n_matches == 0) continue;
        const float2* unused = nullptr;
        rc = corr_keypoints(ctx, s, sources[k].keypoints_key, sources[k].keypoints_xy, sources[k].n_keypoints, &unused, 0, s->copy_stream);
        if (rc != PC_OK) return rc;
    }
    PC_HIP(hipEventRecord(s->upload_done, s->copy_stream));
    return PC_OK;
}

int pc_track_frame_launch(pc_context* ctx, pc_corr_set* s, const pc_mesh* mesh, const float* model_matrix, int check_mask,
                          const pc_track_source* sources, int n_sources, const pc_pnp_camera* initial, const pc_pnp_solve_options* o) {
    return pc_track_frame_launch_chained(ctx, s, mesh, model_matrix, check_mask, sources, n_sources, -1, initial, 0, o);
}

int pc_track_frame_launch_chained(pc_context* ctx, pc_corr_set* s, const pc_mesh* mesh, const float* model_matrix, int check_mask,
                                  const pc_track_source* sources, int n_sources, int chained_source, const pc_pnp_camera* initial,
                                  int chain_initial, const pc_pnp_solve_options* o) {
    if (!ctx || !s || !mesh || !model_matrix || !initial || !o) return fail(PC_E_INVALID, "bad argument");
    if (o->loss_type < 0 || o->loss_type > 2) return fail(PC_E_INVALID, "Unknown loss type: %d", o->loss_type);
    if (s->t_inflight >= 2) return fail(PC_E_STATE, "two frames are in flight: pc_track_frame_finish first");
    if (!s->copy_stream) return fail(PC_E_STATE, "pc_track_frame_upload first");
    if (chained_source >= n_sources) return fail(PC_E_INVALID, "chained_source %d of %d sources", chained_source, n_sources);
    if ((chained_source >= 0 || chain_initial) && !s->t_chain_valid)
        return fail(PC_E_STATE, "nothing to chain to: no launch of this set has left its camera on the device");
    size_t total = 0;
    int rc = check_track_sources(sources, n_sources, s->t_block_bytes, &total);
    if (rc != PC_OK) return rc;
    pc_corr_set::Flight& fl = s->t_flight[s->t_inflight];
    if (total == 0) {   // no correspondences: finish reports "not enough features"
        fl.n = 0;
        fl.seq = 0;
        s->t_inflight++;
        return PC_OK;
    }
    for (int k = 0; k < n_sources; k++)
        if (s->t_inflight > 0 && sources[k].n_matches > 0 && sources[k].keypoints_key < 0)
            return fail(PC_E_STATE, "a second frame in flight needs keyed keypoint arrays (keypoints_key >= 0)");
    PC_HIP(hipSetDevice(ctx->device));
    const int n = (int)total;
    const int nb = pc::track_lm_blocks(n);
    PC_HIP(s->t_obs.ensure(total));
    PC_HIP(s->t_pts.ensure(total));
    PC_HIP(s->t_partials.ensure((size_t)nb * 56));
    PC_HIP(s->t_out.ensure(2));
    PC_HIP(s->t_chain.ensure(1));
    PC_HIP(s->t_sync.ensure((size_t)pc::kTrackSyncWords));
    if (!s->t_sync_zero) {
        PC_HIP(hipMemsetAsync(s->t_sync.p, 0, pc::kTrackSyncWords * sizeof(uint32_t), ctx->stream));
        s->t_sync_zero = true;   // from here on every launch leaves the words zero
    }
    PC_HIP(hipStreamWaitEvent(ctx->stream, s->upload_done, 0));
    const uint8_t* block = s->t_block[s->t_cur].p;
    pc::TrackCastArgs ca;
    std::memset(&ca, 0, sizeof(ca));
    ca.bvh = mesh->bvh();
    ca.mask = mesh->mask.p;
    ca.check_mask = check_mask;
    std::memcpy(ca.model.m, model_matrix, sizeof(ca.model.m));
    int blocks = 0, begin = 0, used = 0;
    for (int k = 0; k < n_sources; k++) {
        const pc_track_source& q = sources[k];
        if (q.n_matches == 0) continue;
        pc::TrackSource& t = ca.src[used];
        to_ray_camera(&q.cam, &t.cam);
        t.cam_dev = (k == chained_source) ? &s->t_chain.p->ray : nullptr;
        rc = corr_keypoints(ctx, s, q.keypoints_key, q.keypoints_xy, q.n_keypoints, &t.kps, used);   // cached by the upload, else sent now
        if (rc != PC_OK) return rc;
        t.idx = reinterpret_cast<const uint32_t*>(block + q.idx_offset);
        t.tgt = reinterpret_cast<const float2*>(block + q.tgt_offset);
        t.n_kps = q.n_keypoints;
        t.begin = begin;
        t.n_matches = q.n_matches;
        t.block_begin = blocks;
        begin += q.n_matches;
        blocks += pc::track_cast_blocks(q.n_matches);
        used++;
    }
    ca.n_sources = used;
    ca.pts = s->t_pts.p;
    ca.obs = s->t_obs.p;
    ca.bad_index = s->counter.p + 1;
    pc::launch_track_cast(ca, blocks, ctx->stream);
    pc::TrackLmArgs la;
    std::memset(&la, 0, sizeof(la));
    la.pts = s->t_pts.p;
    la.obs = s->t_obs.p;
    la.n = n;
    fill_lm_config(o, &la.cfg);
    fill_lm_camera(initial, &la.cam);
    la.chain_in = chain_initial ? s->t_chain.p : nullptr;
    la.chain_out = s->t_chain.p;
    {   // the model matrix as 4x4 (the caller passes its rows 0-2 as the first 12 floats of a row-major 4x4)
        std::memcpy(la.model, model_matrix, 16 * sizeof(float));
    }
    la.partials = s->t_partials.p;
    la.sync = s->t_sync.p;
    la.seq = s->t_seq + 1u;
    pc::TrackLmOut* const slot = s->t_out.p + (la.seq & 1u);
    la.out = slot;
    la.max_rounds = o->max_iterations + 3;   // the initial sweep, one per iteration, one more for the 3-point case
    {
        const char* sd = getenv("POLYCHASE_TRACK_SERIAL_DECISION");   // read per launch: the tests flip it
        la.serial_decision = (sd && sd[0] == '1') ? 1 : 0;
    }
    la.bad_index = s->counter.p + 1;         // zero unless an earlier call found a bad index and has not been cleared
    slot->status = -1;
    slot->bad_index = 0;
    slot->seq = la.seq ^ 0x80000000u;        // anything but this launch's number (the word was never initialised)
    pc::launch_track_lm(la, ctx->stream);
    // both launches are enqueued: only now is the frame in flight (an error return above used to leave it so)
    s->t_seq = la.seq;
    s->t_chain_valid = true;
    fl.n = n;
    fl.seq = la.seq;
    s->t_inflight++;
    return PC_OK;
}

int pc_track_frame_finish(pc_context* ctx, pc_corr_set* s, pc_track_solve_result* result) {
    if (!ctx || !s || !result) return fail(PC_E_INVALID, "null argument");
    if (s->t_inflight < 1) return fail(PC_E_STATE, "no frame is in flight");
    const pc_corr_set::Flight fl = s->t_flight[0];   // the oldest
    s->t_flight[0] = s->t_flight[1];
    s->t_inflight--;
    std::memset(result, 0, sizeof(*result));
    result->n_matches = fl.n;
    if (fl.n == 0) return PC_OK;
    PC_HIP(hipSetDevice(ctx->device));
    const pc::TrackLmOut* const slot = s->t_out.p + (fl.seq & 1u);
    {
        // The kernel writes its result into page-locked host memory and its launch number last (system-scope release): poll
        // that word instead of waiting for the stream -- the pose is here before the other workgroups have left the GPU and the
        // queue has signalled, and the next frame's launches are already on their way by then.  The stream wait stays as the
        // fall-back (a kernel that died never writes the word).
        const uint32_t* word = &slot->seq;
        bool seen = false;
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spins = 0; !seen; spins++) {
            seen = __atomic_load_n(word, __ATOMIC_ACQUIRE) == fl.seq;
            if (seen) break;
            __builtin_ia32_pause();
            if ((spins & 0x3ffu) == 0x3ffu && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(500)) break;
        }
        if (!seen) {
            PC_HIP(hipStreamSynchronize(ctx->stream));
            if (__atomic_load_n(word, __ATOMIC_ACQUIRE) != fl.seq) {
                s->t_sync_zero = false;
                s->t_chain_valid = false;
                return fail(PC_E_STATE, "the PnP solver's launch ended without a result");
            }
        }
    }
    const pc::TrackLmOut& out = *slot;
    if (out.status == 2 || out.status < 0) {
        s->t_sync_zero = false;   // the barrier words are in an unknown state
        s->t_chain_valid = false;
        return fail(PC_E_STATE, "the PnP solver's workgroups did not all become resident (status %d)", out.status);
    }
    if (out.bad_index) {
        (void)hipMemsetAsync(s->counter.p, 0, 2 * sizeof(int), ctx->stream);
        return fail(PC_E_INVALID, "a source keypoint index is out of range");
    }
    if (out.status == 3) return fail(PC_E_STATE, "the PnP solver did not finish within %d rounds", out.rounds);
    result->n_correspondences = out.n_valid;
    result->rounds = out.rounds;
    for (int k = 0; k < 8; k++) result->lm_ticks[k] = out.ticks[k];
    result->lm_begin_tick = out.begin_tick;
    result->lm_end_tick = out.end_tick;
    if (out.status == 1) return PC_OK;   // fewer than 3 correspondences: nothing solved
    read_lm_camera(out.cam, &result->pnp.camera);
    result->pnp.iterations = out.iterations;
    result->pnp.invalid_steps = out.invalid_steps;
    result->pnp.initial_cost = out.initial_cost;
    result->pnp.cost = out.cost;
    result->pnp.lambda = out.lambda;
    result->pnp.step_norm = out.step_norm;
    result->pnp.grad_norm = out.grad_norm;
    result->pnp.inliers = out.inliers;
    return PC_OK;
}

int pc_track_solve_frame(pc_context* ctx, pc_corr_set* s, const pc_mesh* mesh, const float* model_matrix, int check_mask,
                         const pc_track_source* sources, int n_sources, const void* matches, size_t matches_bytes,
                         const pc_pnp_camera* initial, const pc_pnp_solve_options* o, pc_track_solve_result* result) {
    if (!result) return fail(PC_E_INVALID, "null argument");
    std::memset(result, 0, sizeof(*result));
    if (s && s->t_inflight > 0) return fail(PC_E_STATE, "a frame is in flight: pc_track_frame_finish first");
    int rc = pc_track_frame_upload(ctx, s, matches, matches_bytes, sources, n_sources);
    if (rc == PC_OK) rc = pc_track_frame_launch(ctx, s, mesh, model_matrix, check_mask, sources, n_sources, initial, o);
    if (rc != PC_OK) return rc;
    return pc_track_frame_finish(ctx, s, result);
}

int pc_track_download_points(pc_context* ctx, pc_corr_set* s, int n, float* world_xyzw) {
    if (!ctx || !s || n < 0 || (n > 0 && !world_xyzw)) return fail(PC_E_INVALID, "bad argument");
    if ((size_t)n > s->t_pts.cap) return fail(PC_E_INVALID, "the last pc_track_solve_frame had fewer matches");
    PC_HIP(hipSetDevice(ctx->device));
    if (n) PC_HIP(hipMemcpyAsync(world_xyzw, s->t_pts.p, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    PC_HIP(hipStreamSynchronize(ctx->stream));
    return PC_OK;
}

int pc_pnp_total_cost(pc_context* ctx, const pc_pnp_problem* prob, const pc_pnp_params* params,
                      float max_inlier_error_sq, float* cost, int* valid, int* inliers) {
    if (!ctx || !prob || !params || !cost) return fail(PC_E_INVALID, "null argument");
    if (params->loss_type < 0 || params->loss_type > 2) return fail(PC_E_INVALID, "Unknown loss type: %d", params->loss_type);
    PC_HIP(hipSetDevice(ctx->device));
    pc::launch_pnp_cost(prob->X, prob->x, prob->has_weights ? prob->w : nullptr, prob->n, to_kernel_params(params),
                        max_inlier_error_sq, prob->partials, prob->out, ctx->stream);
    PC_HIP(hipMemcpyAsync(prob->h_out, prob->out, 4 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    PC_HIP(hipStreamSynchronize(ctx->stream));
    *cost = prob->h_out[0];
    if (valid) *valid = (int)prob->h_out[1];
    if (inliers) *inliers = (int)prob->h_out[2];
    return PC_OK;
}


int pc_debug_llt9(pc_context* ctx, const float* a81, const float* b9, float* l81, float* x9, int* positive_definite) {
    if (!ctx || !a81 || !b9 || !l81 || !x9 || !positive_definite) return fail(PC_E_INVALID, "null argument");
    PC_HIP(hipSetDevice(ctx->device));
    float* d = nullptr;
    PC_HIP(hipMalloc(reinterpret_cast<void**>(&d), (81 + 9 + 81 + 9 + 1) * sizeof(float)));
    hipError_t e = hipMemcpyAsync(d, a81, 81 * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + 81, b9, 9 * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        pc::launch_llt9_debug(d, d + 81, d + 90, d + 171, reinterpret_cast<int*>(d + 180), ctx->stream);
        e = hipMemcpyAsync(l81, d + 90, 81 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(x9, d + 171, 9 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(positive_definite, d + 180, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(PC_E_HIP, "pc_debug_llt9: %s", hipGetErrorString(e));
    return PC_OK;
}

// =============================================================================================
// refiner path
// =============================================================================================
}  // extern "C"

struct pc_refine_problem {
    pc_context* ctx = nullptr;
    const pc_mesh* mesh = nullptr;
    int n_frames = 0, n_edges = 0, block_len = 6, opt_f = 0, opt_pp = 0;
    size_t n_kp = 0, n_res = 0;
    DevBuf<int> kp_offset, edge_src, edge_tgt, edge_offset, edge_valid;
    DevBuf<float2> kp_xy, res_tgt_xy;
    DevBuf<double2> edge_cost;
    DevBuf<uint32_t> res_src_kp, prim_cache;
    DevBuf<float4> tri_plane, tri_verts;
    hipEvent_t sweep_begin = nullptr, sweep_end = nullptr;   // around the kernel of every sweep
    int cost_launches = 0, neq_launches = 0;
    double cost_ms = 0.0, neq_ms = 0.0;
    DevBuf<float> edge_weight;
    DevBuf<double> edge_blocks;
    DevBuf<uint8_t> frame_fixed;
    DevBuf<pc::RefineCamera> cams;
    PinBuf<double2> h_edge_cost;
    std::vector<float> h_edge_weight;
    float model[16], model_inv[16];
};

namespace {

pc::RefineProblemView refine_view(const pc_refine_problem* p) {
    pc::RefineProblemView v;
    v.n_frames = p->n_frames;
    v.n_edges = p->n_edges;
    v.n_tris = p->mesh->n_triangles;
    v.kp_offset = p->kp_offset.p;
    v.kp_xy = p->kp_xy.p;
    v.edge_src = p->edge_src.p;
    v.edge_tgt = p->edge_tgt.p;
    v.edge_offset = p->edge_offset.p;
    v.res_src_kp = p->res_src_kp.p;
    v.res_tgt_xy = p->res_tgt_xy.p;
    v.edge_weight = p->edge_weight.p;
    v.frame_fixed = p->frame_fixed.p;
    v.prim_cache = p->prim_cache.p;
    v.tri_plane = p->tri_plane.p;
    v.tri_verts = p->tri_verts.p;
    v.verts = p->mesh->verts.p;
    v.tris = p->mesh->tris.p;
    v.mask = p->mesh->mask.p;
    v.bvh = p->mesh->bvh();
    std::memcpy(v.model, p->model, sizeof(v.model));
    std::memcpy(v.model_inv, p->model_inv, sizeof(v.model_inv));
    return v;
}

int upload_cameras(pc_context* ctx, pc_refine_problem* p, const pc_refine_camera* cameras) {
    std::vector<pc::RefineCamera> h((size_t)p->n_frames);
    for (int f = 0; f < p->n_frames; f++) {
        std::memcpy(h[f].R, cameras[f].R, sizeof(h[f].R));
        std::memcpy(h[f].t, cameras[f].t, sizeof(h[f].t));
        h[f].fx = cameras[f].fx;
        h[f].fy = cameras[f].fy;
        h[f].cx = cameras[f].cx;
        h[f].cy = cameras[f].cy;
        h[f].aspect = cameras[f].aspect_ratio;
        h[f].sign = cameras[f].unproject_sign;
    }
    PC_HIP(hipMemcpyAsync(p->cams.p, h.data(), h.size() * sizeof(pc::RefineCamera), hipMemcpyHostToDevice, ctx->stream));
    PC_HIP(hipStreamSynchronize(ctx->stream));  // `h` is pageable and local
    return PC_OK;
}

template <typename T, typename U>
hipError_t upload(DevBuf<T>& dst, const U* src, size_t n, hipStream_t s) {
    static_assert(sizeof(T) == sizeof(U) || sizeof(T) == 2 * sizeof(U), "layout");
    hipError_t e = dst.ensure(n ? n : 1);
    if (e == hipSuccess && n) e = hipMemcpyAsync(dst.p, src, n * sizeof(T), hipMemcpyHostToDevice, s);
    return e;
}

}  // namespace

extern "C" {

int pc_refine_problem_create_parts(pc_context* ctx, const pc_mesh* mesh, const pc_refine_desc* d, const pc_refine_part* parts,
                                   int n_parts, pc_refine_problem** out) {
    if (!ctx || !mesh || !d || !out || (n_parts > 0 && !parts) || n_parts < 0) return fail(PC_E_INVALID, "null argument");
    *out = nullptr;
    if (d->n_frames < 3) return fail(PC_E_INVALID, "a segment needs more than 2 frames");  // CHECK(traj.Count() > 2)
    if (d->n_edges < 0 || (d->block_len != 6 && d->block_len != 9)) return fail(PC_E_INVALID, "bad problem description");
    if (!d->kp_offset || !d->edge_offset || (d->n_edges > 0 && (!d->edge_src || !d->edge_tgt || !d->edge_weight)))
        return fail(PC_E_INVALID, "null array");
    if (d->kp_xy || d->res_src_kp || d->res_tgt_xy) return fail(PC_E_INVALID, "the large arrays come in the parts: leave them NULL in the description");
    if (d->kp_offset[0] != 0 || d->edge_offset[0] != 0) return fail(PC_E_INVALID, "offsets start at 0");
    for (int f = 0; f < d->n_frames; f++)
        if (d->kp_offset[f + 1] < d->kp_offset[f]) return fail(PC_E_INVALID, "kp_offset decreases at frame %d", f);
    const size_t n_kp = (size_t)d->kp_offset[d->n_frames], n_res = (size_t)d->edge_offset[d->n_edges];
    for (int e = 0; e < d->n_edges; e++) {
        if (d->edge_src[e] < 0 || d->edge_src[e] >= d->n_frames || d->edge_tgt[e] < 0 || d->edge_tgt[e] >= d->n_frames ||
            d->edge_src[e] == d->edge_tgt[e])
            return fail(PC_E_INVALID, "edge %d connects invalid frames", e);
        if (d->edge_offset[e + 1] < d->edge_offset[e]) return fail(PC_E_INVALID, "edge_offset decreases at edge %d", e);
    }
    {
        size_t kp_sum = 0, res_sum = 0;
        for (int t = 0; t < n_parts; t++) {
            const pc_refine_part& part = parts[t];
            if (part.n_keypoints < 0 || part.n_residuals < 0 || (part.n_keypoints > 0 && !part.kp_xy) ||
                (part.n_residuals > 0 && (!part.res_src_kp || !part.res_tgt_xy)))
                return fail(PC_E_INVALID, "part %d: null array or negative count", t);
            kp_sum += (size_t)part.n_keypoints;
            res_sum += (size_t)part.n_residuals;
        }
        if (kp_sum != n_kp || res_sum != n_res)
            return fail(PC_E_INVALID, "the parts hold %zu keypoints and %zu residuals, the offsets say %zu and %zu", kp_sum, res_sum, n_kp, n_res);
    }
    PC_HIP(hipSetDevice(ctx->device));
    pc_refine_problem* p = new (std::nothrow) pc_refine_problem();
    if (!p) return fail(PC_E_INVALID, "out of host memory");
    p->ctx = ctx;
    p->mesh = mesh;
    p->n_frames = d->n_frames;
    p->n_edges = d->n_edges;
    p->block_len = d->block_len;
    p->opt_f = d->optimize_focal_length ? 1 : 0;
    p->opt_pp = d->optimize_principal_point ? 1 : 0;
    p->n_kp = n_kp;
    p->n_res = n_res;
    std::memcpy(p->model, d->model_matrix, sizeof(p->model));
    std::memcpy(p->model_inv, d->model_matrix_inv, sizeof(p->model_inv));
    std::vector<uint8_t> fixed((size_t)d->n_frames, 0);
    fixed.front() = fixed.back() = 1;  // IsGroundTruth (refiner.cc:268-271)
    const int B2 = 2 * d->block_len, nacc = B2 * (B2 + 1) / 2 + B2;
    hipStream_t s = ctx->stream;
    hipError_t e = upload(p->kp_offset, d->kp_offset, (size_t)d->n_frames + 1, s);
    if (e == hipSuccess) e = upload(p->edge_src, d->edge_src, (size_t)d->n_edges, s);
    if (e == hipSuccess) e = upload(p->edge_tgt, d->edge_tgt, (size_t)d->n_edges, s);
    if (e == hipSuccess) e = upload(p->edge_offset, d->edge_offset, (size_t)d->n_edges + 1, s);
    if (e == hipSuccess) e = upload(p->edge_weight, d->edge_weight, (size_t)d->n_edges, s);
    if (e == hipSuccess) e = upload(p->frame_fixed, fixed.data(), fixed.size(), s);
    if (e == hipSuccess) e = p->kp_xy.ensure(n_kp ? n_kp : 1);
    if (e == hipSuccess) e = p->res_src_kp.ensure(n_res ? n_res : 1);
    if (e == hipSuccess) e = p->res_tgt_xy.ensure(n_res ? n_res : 1);
    {   // Every part to its place.  The pieces are pageable host memory: the runtime page-locks the source of a large copy on
        // the fly, which is what bounds this (17 GB/s from 4-KiB pages, 38-44 GB/s from 2-MiB pages: the host side allocates
        // its parts with MADV_HUGEPAGE).  Measured and not better: hipHostRegister around the copies, several threads with a
        // stream each, a hand-made staging pipeline through page-locked blocks (tools/probes/pageable_upload_probe.hip).
        size_t kp_at = 0, res_at = 0;
        for (int t = 0; t < n_parts && e == hipSuccess; t++) {
            const pc_refine_part& part = parts[t];
            const size_t nk = (size_t)part.n_keypoints, nr = (size_t)part.n_residuals;
            if (nk) e = hipMemcpyAsync(p->kp_xy.p + kp_at, part.kp_xy, nk * sizeof(float2), hipMemcpyHostToDevice, s);
            if (e == hipSuccess && nr) e = hipMemcpyAsync(p->res_src_kp.p + res_at, part.res_src_kp, nr * sizeof(uint32_t), hipMemcpyHostToDevice, s);
            if (e == hipSuccess && nr) e = hipMemcpyAsync(p->res_tgt_xy.p + res_at, part.res_tgt_xy, nr * sizeof(float2), hipMemcpyHostToDevice, s);
            kp_at += nk;
            res_at += nr;
        }
    }
    if (e == hipSuccess) e = p->prim_cache.ensure(n_kp ? n_kp : 1);
    if (e == hipSuccess) e = hipMemsetAsync(p->prim_cache.p, 0xff, (n_kp ? n_kp : 1) * sizeof(uint32_t), s);
    if (e == hipSuccess) e = p->edge_cost.ensure((size_t)std::max(1, d->n_edges));
    if (e == hipSuccess) e = p->h_edge_cost.ensure((size_t)std::max(1, d->n_edges));
    if (e == hipSuccess) e = p->edge_valid.ensure((size_t)std::max(1, d->n_edges));
    if (e == hipSuccess) e = p->edge_blocks.ensure((size_t)std::max(1, d->n_edges) * nacc);
    if (e == hipSuccess) e = p->cams.ensure((size_t)d->n_frames);
    if (e == hipSuccess) e = hipEventCreate(&p->sweep_begin);
    if (e == hipSuccess) e = hipEventCreate(&p->sweep_end);
    if (e == hipSuccess) e = p->tri_plane.ensure(2 * (size_t)std::max(1, mesh->n_triangles));
    if (e == hipSuccess) e = p->tri_verts.ensure(3 * (size_t)std::max(1, mesh->n_triangles));
    if (e == hipSuccess) {
        pc::launch_refine_tri_planes(refine_view(p), p->tri_plane.p, s);
        pc::launch_refine_tri_verts(refine_view(p), p->tri_verts.p, s);
        e = hipGetLastError();
    }
    // the residuals' keypoint indices are checked where they now are (edge_valid is free until the first sweep)
    int bad[2] = {std::numeric_limits<int>::max(), 0};
    if (e == hipSuccess && d->n_edges > 0) {
        e = p->edge_valid.ensure((size_t)std::max(2, d->n_edges));
        if (e == hipSuccess) e = hipMemcpyAsync(p->edge_valid.p, bad, sizeof(bad), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) {
            pc::launch_refine_validate(refine_view(p), p->edge_valid.p, s);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(bad, p->edge_valid.p, sizeof(bad), hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        pc_refine_problem_destroy(p);
        return fail(PC_E_HIP, "refine problem upload failed: %s", hipGetErrorString(e));
    }
    if (bad[0] != std::numeric_limits<int>::max()) {
        const int edge = bad[0];
        const int src_kps = d->kp_offset[d->edge_src[edge] + 1] - d->kp_offset[d->edge_src[edge]];
        pc_refine_problem_destroy(p);
        return fail(PC_E_INVALID, "edge %d references a keypoint (up to %d) its source frame does not have (%d keypoints)", edge, bad[1], src_kps);
    }
    p->h_edge_weight.assign(d->edge_weight, d->edge_weight + d->n_edges);
    *out = p;
    return PC_OK;
}

int pc_refine_problem_create(pc_context* ctx, const pc_mesh* mesh, const pc_refine_desc* d, pc_refine_problem** out) {
    if (!ctx || !mesh || !d || !out) return fail(PC_E_INVALID, "null argument");
    *out = nullptr;
    if (d->n_frames < 3) return fail(PC_E_INVALID, "a segment needs more than 2 frames");
    if (d->n_edges < 0 || !d->kp_offset || !d->edge_offset) return fail(PC_E_INVALID, "bad problem description");
    pc_refine_part whole{};
    whole.kp_xy = d->kp_xy;
    whole.n_keypoints = d->kp_offset[d->n_frames];
    whole.res_src_kp = d->res_src_kp;
    whole.res_tgt_xy = d->res_tgt_xy;
    whole.n_residuals = d->edge_offset[d->n_edges];
    pc_refine_desc small = *d;
    small.kp_xy = nullptr;
    small.res_src_kp = nullptr;
    small.res_tgt_xy = nullptr;
    return pc_refine_problem_create_parts(ctx, mesh, &small, &whole, 1, out);
}

void pc_refine_problem_destroy(pc_refine_problem* p) {
    if (!p) return;
    if (p->ctx) {
        (void)hipSetDevice(p->ctx->device);
        (void)hipStreamSynchronize(p->ctx->stream);
    }
    p->kp_offset.release();
    p->edge_src.release();
    p->edge_tgt.release();
    p->edge_offset.release();
    p->edge_valid.release();
    p->kp_xy.release();
    p->res_tgt_xy.release();
    p->edge_cost.release();
    p->res_src_kp.release();
    p->prim_cache.release();
    p->tri_plane.release();
    if (p->sweep_begin) (void)hipEventDestroy(p->sweep_begin);
    if (p->sweep_end) (void)hipEventDestroy(p->sweep_end);
    p->tri_verts.release();
    p->edge_weight.release();
    p->edge_blocks.release();
    p->frame_fixed.release();
    p->cams.release();
    p->h_edge_cost.release();
    delete p;
}

int pc_refine_total_cost(pc_context* ctx, pc_refine_problem* p, const pc_refine_camera* cameras, int loss_type,
                         float loss_scale, double* cost) {
    if (!ctx || !p || !cameras || !cost) return fail(PC_E_INVALID, "null argument");
    if (loss_type < 0 || loss_type > 2) return fail(PC_E_INVALID, "Unknown loss type: %d", loss_type);
    PC_HIP(hipSetDevice(ctx->device));
    int rc = upload_cameras(ctx, p, cameras);
    if (rc != PC_OK) return rc;
    *cost = 0.0;
    if (p->n_edges == 0) return PC_OK;
    PC_HIP(hipEventRecord(p->sweep_begin, ctx->stream));
    pc::launch_refine_cost(refine_view(p), p->cams.p, loss_type, loss_scale, p->edge_cost.p, ctx->stream);
    PC_HIP(hipEventRecord(p->sweep_end, ctx->stream));
    PC_HIP(hipMemcpyAsync(p->h_edge_cost.p, p->edge_cost.p, (size_t)p->n_edges * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
    PC_HIP(hipStreamSynchronize(ctx->stream));
    {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p->sweep_begin, p->sweep_end) == hipSuccess) p->cost_ms += ms, p->cost_launches++;
    }
    // cost = sum_e edge_weight * (edge loss sum / valid)   (lev_marq.h:812-820), fixed edge order
    double total = 0.0;
    for (int e = 0; e < p->n_edges; e++) {
        const float w = p->h_edge_weight[(size_t)e];
        if (w == 0.0f) continue;
        double edge_cost = p->h_edge_cost.p[e].x;
        if (p->h_edge_cost.p[e].y > 0.0) edge_cost /= p->h_edge_cost.p[e].y;
        total += (double)w * edge_cost;
    }
    *cost = total;
    return PC_OK;
}

int pc_refine_normal_equations(pc_context* ctx, pc_refine_problem* p, const pc_refine_camera* cameras, int loss_type,
                               float loss_scale, double* edge_blocks, int* edge_valid) {
    if (!ctx || !p || !cameras || !edge_blocks) return fail(PC_E_INVALID, "null argument");
    if (loss_type < 0 || loss_type > 2) return fail(PC_E_INVALID, "Unknown loss type: %d", loss_type);
    PC_HIP(hipSetDevice(ctx->device));
    int rc = upload_cameras(ctx, p, cameras);
    if (rc != PC_OK) return rc;
    if (p->n_edges == 0) return PC_OK;
    const int B2 = 2 * p->block_len, nacc = B2 * (B2 + 1) / 2 + B2;
    PC_HIP(hipEventRecord(p->sweep_begin, ctx->stream));
    pc::launch_refine_normal_eq(refine_view(p), p->cams.p, loss_type, loss_scale, p->block_len, p->opt_f, p->opt_pp,
                                p->edge_blocks.p, p->edge_valid.p, ctx->stream);
    PC_HIP(hipEventRecord(p->sweep_end, ctx->stream));
    PC_HIP(hipMemcpyAsync(edge_blocks, p->edge_blocks.p, (size_t)p->n_edges * nacc * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (edge_valid)
        PC_HIP(hipMemcpyAsync(edge_valid, p->edge_valid.p, (size_t)p->n_edges * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PC_HIP(hipStreamSynchronize(ctx->stream));
    {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p->sweep_begin, p->sweep_end) == hipSuccess) p->neq_ms += ms, p->neq_launches++;
    }
    return PC_OK;
}

int pc_refine_problem_timing(const pc_refine_problem* p, int* cost_launches, double* cost_ms, int* normal_eq_launches, double* normal_eq_ms) {
    if (!p) return fail(PC_E_INVALID, "null argument");
    if (cost_launches) *cost_launches = p->cost_launches;
    if (cost_ms) *cost_ms = p->cost_ms;
    if (normal_eq_launches) *normal_eq_launches = p->neq_launches;
    if (normal_eq_ms) *normal_eq_ms = p->neq_ms;
    return PC_OK;
}

}  // extern "C"
