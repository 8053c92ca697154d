// trajectory_refiner.cc -- host side of "Refine Sequence".
//   segment loading   : CachedDatabase (cpp/refiner.cc:18-197) flattened into the CSR arrays the GPU
//                       problem wants (keypoints per frame, residuals per edge)
//   LM driver         : LevMarqSparseSolver::Solve (cpp/pnp/lev_marq.h:503-601), Step
//                       (cpp/refiner.cc:508-540, :659-690)
//   linear algebra    : the frames of a segment only connect to frames at most `max |i-j|` apart, so
//                       J^T J is block-banded; a banded Cholesky (fp64) replaces Eigen::SimplicialLLT
//   residual sweeps   : on the GPU through pc_refine_* (kernels_refiner.hip), one workgroup per edge
#include "trajectory_refiner.h"

#include "stage_clock.h"

#include <sys/mman.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <new>
#include <stdexcept>
#include <thread>
#include <utility>
#include <vector>

#include "band_matrix.h"
#include "flow_database.h"
#include "gpu_context.h"
#include "numa_pin.h"

namespace {

// ---------------------------------------------------------------------------------------------
// segment data
// ---------------------------------------------------------------------------------------------
// A growing array of trivially copyable elements in anonymous memory that asks for transparent huge pages.  The
// gigabyte-sized arrays of a segment are appended to by their reader thread, handed to the GPU once and given back: with 4-KiB
// pages the kernel faults in, page-locks (the runtime pins the source of a large pageable copy) and unmaps 300 000 pages --
// measured at 300 frames of 1080p: 78 ms for the copy and 163 ms for the release; 2-MiB pages divide the page count by 512
// (tools/probes/pinned_alloc_probe.hip; the box runs transparent_hugepage=madvise).  Growth remaps the pages instead of
// copying them.  New elements are not initialised: the thread that appends is the first to touch the pages.
template <class T>
class GrowArray {
   public:
    GrowArray() = default;
    GrowArray(GrowArray&& o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_) { o.p_ = nullptr, o.n_ = o.cap_ = 0; }
    GrowArray& operator=(GrowArray&& o) noexcept {
        if (this != &o) {
            Release();
            p_ = o.p_, n_ = o.n_, cap_ = o.cap_;
            o.p_ = nullptr, o.n_ = o.cap_ = 0;
        }
        return *this;
    }
    GrowArray(const GrowArray&) = delete;
    GrowArray& operator=(const GrowArray&) = delete;
    ~GrowArray() { Release(); }
    // room for `more` further elements; returns where the next one goes
    T* Room(size_t more) {
        if (n_ + more > cap_) {
            const size_t want_bytes = RoundUp(std::max((n_ + more) * sizeof(T), cap_ * sizeof(T) * 3 / 2));
            void* q = p_ ? mremap(p_, cap_ * sizeof(T), want_bytes, MREMAP_MAYMOVE)
                         : mmap(nullptr, want_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (q == MAP_FAILED) throw std::bad_alloc();
            (void)madvise(q, want_bytes, MADV_HUGEPAGE);   // advice: refused where the kernel has no THP, harmless then
            p_ = static_cast<T*>(q);
            cap_ = want_bytes / sizeof(T);
        }
        return p_ + n_;
    }
    void Grew(size_t by) { n_ += by; }
    const T* data() const { return p_; }
    size_t size() const { return n_; }
    // The pages are dropped with MADV_DONTNEED first: that runs under the shared address-space lock, so several threads give
    // their arrays back side by side (munmap alone holds the lock exclusively while the kernel clears the pages).
    void Release() {
        if (p_) {
            (void)madvise(p_, cap_ * sizeof(T), MADV_DONTNEED);
            munmap(p_, cap_ * sizeof(T));
        }
        p_ = nullptr, n_ = cap_ = 0;
    }

   private:
    static constexpr size_t kChunk = size_t{32} << 20;   // multiples of 32 MiB (virtual: untouched pages cost nothing)
    static size_t RoundUp(size_t bytes) { return (bytes + kChunk - 1) / kChunk * kChunk; }
    T* p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
};

// what one reader thread brings back: a run of consecutive frames, offsets relative to the part
struct SegmentPart {
    std::vector<int32_t> kp_count;                 // per frame of the part
    std::vector<int32_t> edge_src, edge_tgt, edge_count;
    std::vector<float> edge_weight;
    GrowArray<float> kp_xy;                        // bbox-filtered keypoints, x y
    GrowArray<uint32_t> res_src_kp;                // per residual: keypoint index within its (filtered) source frame
    GrowArray<float> res_tgt_xy;
};

// The segment as the GPU problem wants it (CSR over frames and over edges).  The three large arrays stay in the parts they
// were read into: pc_refine_problem_create_parts puts every part at its place on the device, the host never joins them.
struct Segment {
    int32_t first_frame = 0;
    int32_t n_frames = 0;
    std::vector<int32_t> kp_offset;   // n_frames + 1
    std::vector<int32_t> edge_src, edge_tgt, edge_offset;
    std::vector<float> edge_weight;
    std::vector<SegmentPart> parts;
    int32_t NumEdges() const { return static_cast<int32_t>(edge_src.size()); }
    int32_t NumResiduals() const { return edge_offset.back(); }
};

struct Box2 {
    float lo[2], hi[2];
    bool Contains(float x, float y) const { return x > lo[0] && y > lo[1] && x < hi[0] && y < hi[1]; }
};

// image-space bounding box of the mesh's bounding box seen from `state`, padded by 20 px
// (TransformBbox + ComputeBbox, refiner.cc:18-72)
Box2 ProjectedMeshBox(const float pmin[3], const float pmax[3], const CameraState& state, const Mat4f& model_matrix) {
    const CameraIntrinsics& in = state.intrinsics;
    const Mat4f K = {in.fx, 0, in.cx, 0, 0, in.fy, in.cy, 0, 0, 0, -(110.0f / 90.0f), -2.0f * 100.0f * 10.0f / 90.0f, 0, 0, 1, 0};
    const Mat4f mvp = MatMul4(MatMul4(K, state.pose.Rt4x4()), model_matrix);
    Box2 box{{std::numeric_limits<float>::max(), std::numeric_limits<float>::max()},
             {std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest()}};
    for (int corner = 0; corner < 8; corner++) {
        const float p[3] = {(corner & 4) ? pmax[0] : pmin[0], (corner & 2) ? pmax[1] : pmin[1], (corner & 1) ? pmax[2] : pmin[2]};
        float h[4];
        for (int r = 0; r < 4; r++) h[r] = mvp[4 * r] * p[0] + mvp[4 * r + 1] * p[1] + mvp[4 * r + 2] * p[2] + mvp[4 * r + 3];
        for (int a = 0; a < 2; a++) {
            const float c = h[a] / h[3];
            box.lo[a] = std::min(box.lo[a], c);
            box.hi[a] = std::max(box.hi[a], c);
        }
    }
    constexpr float kPadding = 20.0f;
    for (int a = 0; a < 2; a++) {
        box.lo[a] -= kPadding;
        box.hi[a] += kPadding;
    }
    return box;
}

void MeshBounds(const Mesh& mesh, float pmin[3], float pmax[3]) {
    for (int a = 0; a < 3; a++) {
        pmin[a] = std::numeric_limits<float>::max();
        pmax[a] = std::numeric_limits<float>::lowest();
    }
    for (size_t v = 0; v < mesh.NumVertices(); v++)
        for (int a = 0; a < 3; a++) {
            pmin[a] = std::min(pmin[a], mesh.vertices[3 * v + a]);
            pmax[a] = std::max(pmax[a], mesh.vertices[3 * v + a]);
        }
}

// frames [frame_lo, frame_hi] of the segment.  Keypoints and matches are filtered straight out of the blobs SQLite holds.
SegmentPart LoadSegmentPart(const Database& db, const CameraTrajectory& traj, const float pmin[3], const float pmax[3],
                            const Mat4f& model_matrix, int32_t frame_lo, int32_t frame_hi) {
    constexpr uint32_t kDropped = std::numeric_limits<uint32_t>::max();
    SegmentPart part;
    std::vector<uint32_t> remap;
    std::vector<int32_t> targets;
    for (int32_t frame = frame_lo; frame <= frame_hi; frame++) {
        // keypoints inside the projected mesh box, order preserved (FilterKeypoints, refiner.cc:162-186)
        const Box2 box = ProjectedMeshBox(pmin, pmax, *traj.Get(frame), model_matrix);
        uint32_t kept = 0;
        remap.clear();
        db.VisitKeypoints(frame, [&](size_t rows, const void* blob) {
            remap.resize(rows);
            float* out = part.kp_xy.Room(2 * rows);
            const char* in = static_cast<const char*>(blob);
            for (size_t j = 0; j < rows; j++) {
                float xy[2];
                std::memcpy(xy, in + j * sizeof(Keypoint), sizeof xy);
                if (!box.Contains(xy[0], xy[1])) {
                    remap[j] = kDropped;
                    continue;
                }
                remap[j] = kept;
                out[2 * kept] = xy[0];
                out[2 * kept + 1] = xy[1];
                kept++;
            }
            part.kp_xy.Grew(2 * static_cast<size_t>(kept));
        });
        part.kp_count.push_back(static_cast<int32_t>(kept));

        // flows to frames inside the segment, restricted to the kept keypoints (LoadFrameFlows, :117-160)
        targets.clear();
        db.FindOpticalFlowsFromImage(frame, targets);
        const float dist = static_cast<float>(std::min(frame - traj.FirstFrame(), traj.LastFrame() - frame));
        for (int32_t to : targets) {
            if (!traj.IsValidFrame(to)) continue;
            size_t added = 0;
            // flow_errors are not used (refiner.cc:117-160)
            db.VisitImagePairMatches(frame, to, [&](size_t rows, const void* idx_blob, const void* tgt_blob) {
                uint32_t* out_kp = part.res_src_kp.Room(rows);
                float* out_xy = part.res_tgt_xy.Room(2 * rows);
                const char* idx_in = static_cast<const char*>(idx_blob);
                const char* tgt_in = static_cast<const char*>(tgt_blob);
                for (size_t j = 0; j < rows; j++) {
                    uint32_t src;
                    std::memcpy(&src, idx_in + j * sizeof(uint32_t), sizeof src);
                    CHECK_LT(static_cast<size_t>(src), remap.size());
                    if (remap[src] == kDropped) continue;
                    out_kp[added] = remap[src];
                    std::memcpy(out_xy + 2 * added, tgt_in + j * sizeof(Keypoint), sizeof(Keypoint));
                    added++;
                }
                part.res_src_kp.Grew(added);
                part.res_tgt_xy.Grew(2 * added);
            });
            if (added == 0) continue;
            CHECK_LT(added, static_cast<size_t>(std::numeric_limits<int32_t>::max()));
            part.edge_src.push_back(frame - traj.FirstFrame());
            part.edge_tgt.push_back(to - traj.FirstFrame());
            part.edge_count.push_back(static_cast<int32_t>(added));
            part.edge_weight.push_back(1.0f / (dist + 1.0f));  // FrameWeight(image_id_from), refiner.cc:249-256
        }
    }
    return part;
}

// The whole segment (CachedDatabase, refiner.cc:71-197).  A 300-frame 1080p clip holds 1.5 GB of blobs: the frames
// are dealt out to reader threads, each with its own read connection (the file is in WAL mode: readers do not
// block each other); the small per-frame and per-edge arrays are joined in frame order, the large ones stay where they were
// read -- the result does not depend on the thread count.
Segment LoadSegment(const std::string& database_path, const CameraTrajectory& traj, const Mesh& mesh, const Mat4f& model_matrix) {
    const int32_t n = static_cast<int32_t>(traj.Count());
    // 16 connections read 1.2 GB in 24 ms on the GPU box (8: 34 ms; 32 and more queue up behind each other in Open())
    int n_threads = static_cast<int>(std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())));
    if (const char* env = std::getenv("POLYCHASE_DB_READERS")) n_threads = std::max(1, std::atoi(env));
    n_threads = std::max(1, std::min(n_threads, n / 8));   // short segments: not worth a second connection
    float pmin[3], pmax[3];
    MeshBounds(mesh, pmin, pmax);
    Segment seg;
    seg.first_frame = traj.FirstFrame();
    seg.n_frames = n;
    seg.parts.resize(static_cast<size_t>(n_threads));
    std::vector<std::exception_ptr> errors(static_cast<size_t>(n_threads));
    // opened one after the other: Open() issues pragmas and CREATE TABLE IF NOT EXISTS, which take the write lock
    std::vector<std::unique_ptr<Database>> connections;
    {
        StageClock::Scope sc("refine/load: open connections");
        for (int t = 0; t < n_threads; t++) connections.push_back(std::make_unique<Database>(database_path));
    }
    auto work = [&](int t) {
        try {
            const Database& db = *connections[static_cast<size_t>(t)];
            const int32_t lo = traj.FirstFrame() + static_cast<int32_t>(static_cast<int64_t>(n) * t / n_threads);
            const int32_t hi = traj.FirstFrame() + static_cast<int32_t>(static_cast<int64_t>(n) * (t + 1) / n_threads) - 1;
            seg.parts[static_cast<size_t>(t)] = LoadSegmentPart(db, traj, pmin, pmax, model_matrix, lo, hi);
        } catch (...) {
            errors[static_cast<size_t>(t)] = std::current_exception();
        }
    };
    {
        StageClock::Scope sc("refine/load: parallel read");
        std::vector<std::thread> threads;
        for (int t = 1; t < n_threads; t++) threads.emplace_back(work, t);
        work(0);
        for (auto& th : threads) th.join();
    }
    {
        StageClock::Scope sc("refine/load: close connections");
        connections.clear();
    }
    for (const auto& e : errors)
        if (e) std::rethrow_exception(e);
    seg.kp_offset.push_back(0);
    seg.edge_offset.push_back(0);
    int64_t n_res = 0;
    for (const SegmentPart& p : seg.parts) {
        for (int32_t c : p.kp_count) seg.kp_offset.push_back(seg.kp_offset.back() + c);
        for (int32_t c : p.edge_count) {
            n_res += c;
            CHECK_LT(n_res, static_cast<int64_t>(std::numeric_limits<int32_t>::max()));
            seg.edge_offset.push_back(static_cast<int32_t>(n_res));
        }
        seg.edge_src.insert(seg.edge_src.end(), p.edge_src.begin(), p.edge_src.end());
        seg.edge_tgt.insert(seg.edge_tgt.end(), p.edge_tgt.begin(), p.edge_tgt.end());
        seg.edge_weight.insert(seg.edge_weight.end(), p.edge_weight.begin(), p.edge_weight.end());
    }
    CHECK_EQ(seg.kp_offset.size(), static_cast<size_t>(n) + 1);
    return seg;
}

double Norm(const std::vector<double>& v) {
    double s = 0.0;
    for (double x : v) s += x * x;
    return std::sqrt(s);
}

[[noreturn]] void ThrowHip(const char* what) { throw std::runtime_error(std::string(what) + ": " + pc_last_error()); }

struct GpuRefineProblem {
    pc_refine_problem* p = nullptr;
    ~GpuRefineProblem() {
        GpuSection section;
        pc_refine_problem_destroy(p);
    }
};

void PackCameras(const CameraTrajectory& traj, std::vector<pc_refine_camera>& out) {
    out.resize(traj.Count());
    for (size_t i = 0; i < out.size(); i++) {
        const CameraState& s = *traj.Get(traj.FirstFrame() + static_cast<int32_t>(i));
        const Mat3f R = s.pose.R();
        pc_refine_camera& c = out[i];
        for (int k = 0; k < 9; k++) c.R[k] = R[k];
        for (int k = 0; k < 3; k++) c.t[k] = s.pose.t[k];
        c.fx = s.intrinsics.fx;
        c.fy = s.intrinsics.fy;
        c.cx = s.intrinsics.cx;
        c.cy = s.intrinsics.cy;
        c.aspect_ratio = s.intrinsics.aspect_ratio;
        c.unproject_sign = s.intrinsics.convention == CameraConvention::OpenCV ? 1.0f : -1.0f;
        c.reserved[0] = c.reserved[1] = 0.f;
    }
}

// RefinementProblemBase::Step (refiner.cc:508-540)
void StepCamera(CameraState& state, const float* dp, bool opt_f, bool opt_pp, const CameraIntrinsics::Bounds& b) {
    state.pose.q = QuatStepPost(state.pose.q, Vec3f{dp[0], dp[1], dp[2]});
    state.pose.t = state.pose.t + Vec3f{dp[3], dp[4], dp[5]};
    if (opt_f) {
        state.intrinsics.fy = state.intrinsics.fy + dp[6];
        state.intrinsics.fx = state.intrinsics.fy * state.intrinsics.aspect_ratio;
        state.intrinsics.fy = std::clamp(state.intrinsics.fy, b.f_low, b.f_high);
        state.intrinsics.fx = std::clamp(state.intrinsics.fx, b.f_low, b.f_high);
    }
    if (opt_pp) {
        state.intrinsics.cx = std::clamp(state.intrinsics.cx + dp[7], b.cx_low, b.cx_high);
        state.intrinsics.cy = std::clamp(state.intrinsics.cy + dp[8], b.cy_low, b.cy_high);
    }
}

// The GPU-resident problem of one segment plus the host-side normal equations.
class RefineSession {
   public:
    RefineSession(const std::string& database_path, const CameraTrajectory& traj, const Mat4f& model_matrix,
                  const AcceleratedMesh& mesh, bool opt_f, bool opt_pp, const BundleOptions& opts)
        : opts_(opts),
          loss_type_(static_cast<int>(opts.loss_type)),
          block_((opt_f || opt_pp) ? 9 : 6),  // refiner.cc:227-232
          ctx_(SharedGpuContext()) {
        if (loss_type_ < 0 || loss_type_ > 2) throw std::runtime_error("Unknown loss type: " + std::to_string(loss_type_));
        CHECK(traj.Count() > 2);  // refiner.cc:660
        for (int32_t frame = traj.FirstFrame(); frame <= traj.LastFrame(); frame++) CHECK(traj.IsFrameFilled(frame));
        {
            StageClock::Scope sc("refine/load segment (SQLite)");
            seg_ = LoadSegment(database_path, traj, mesh.Inner(), model_matrix);
        }
        Mat4f model_inv;
        if (!Inverse4(model_matrix, &model_inv)) throw std::runtime_error("model_matrix is singular");
        pc_refine_desc desc{};
        desc.n_frames = seg_.n_frames;
        desc.n_edges = seg_.NumEdges();
        desc.kp_offset = seg_.kp_offset.data();
        desc.edge_src = seg_.edge_src.data();
        desc.edge_tgt = seg_.edge_tgt.data();
        desc.edge_offset = seg_.edge_offset.data();
        desc.edge_weight = seg_.edge_weight.data();
        std::copy(model_matrix.begin(), model_matrix.end(), desc.model_matrix);
        std::copy(model_inv.begin(), model_inv.end(), desc.model_matrix_inv);
        desc.block_len = block_;
        desc.optimize_focal_length = opt_f ? 1 : 0;
        desc.optimize_principal_point = opt_pp ? 1 : 0;
        std::vector<pc_refine_part> parts;
        for (const SegmentPart& p : seg_.parts) {
            pc_refine_part part{};
            part.kp_xy = p.kp_xy.data();
            part.n_keypoints = static_cast<int64_t>(p.kp_xy.size() / 2);
            part.res_src_kp = p.res_src_kp.data();
            part.res_tgt_xy = p.res_tgt_xy.data();
            part.n_residuals = static_cast<int64_t>(p.res_src_kp.size());
            parts.push_back(part);
        }
        {
            // Evaluate ray casts with check_mask = true (refiner.cc:335): send the current mask bits
            GpuSection section;   // mask upload + problem upload: one section on the shared context (gpu_context.h)
            if (pc_mesh_set_mask(ctx_, mesh.Gpu(), mesh.Inner().masked_triangles.data(),
                                 static_cast<int>(mesh.Inner().masked_triangles.size())) != PC_OK)
                ThrowHip("pc_mesh_set_mask");
            StageClock::Scope sc("refine/upload");
            if (pc_refine_problem_create_parts(ctx_, mesh.Gpu(), &desc, parts.data(), static_cast<int>(parts.size()), &gpu_.p) != PC_OK)
                ThrowHip("pc_refine_problem_create_parts");
        }
        {
            // The device holds the large arrays now.  This kernel clears pages when they are freed (1.2 GB: 70 ms on one
            // core, tools/probes/pageable_upload_probe.hip), so every part is given back by a thread of its own.  (Doing it
            // behind the solver's back was measured too: the first sweeps slow down by as much as this takes.)
            StageClock::Scope sc("refine/release host copy");
            std::vector<std::thread> threads;
            for (size_t t = 1; t < seg_.parts.size(); t++) threads.emplace_back([part = &seg_.parts[t]] { *part = SegmentPart(); });
            if (!seg_.parts.empty()) seg_.parts[0] = SegmentPart();
            for (auto& th : threads) th.join();
            seg_.parts.clear();
        }

        // J^T J pattern (lev_marq.h:421-487): diagonal blocks + one off-diagonal block per connected pair
        int reach = 0;
        for (int e = 0; e < seg_.NumEdges(); e++) reach = std::max(reach, std::abs(seg_.edge_src[e] - seg_.edge_tgt[e]));
        half_bandwidth_ = reach * block_ + block_ - 1;
        JtJ = BandMatrix(NumParams(), half_bandwidth_);
        Jtr.assign(NumParams(), 0.0);
        diag.assign(NumParams(), 0.0);
        const int pair = 2 * block_;
        edge_blocks_.resize(static_cast<size_t>(std::max(1, seg_.NumEdges())) * (pair * (pair + 1) / 2 + pair));
    }

    // the kernels' own durations (HIP events) into the stage report, beside the host's view of the same calls
    void ReportGpuTimes() const {
        int cost_n = 0, neq_n = 0;
        double cost_ms = 0.0, neq_ms = 0.0;
        if (pc_refine_problem_timing(gpu_.p, &cost_n, &cost_ms, &neq_n, &neq_ms) != PC_OK) return;
        StageClock::Add("refine/cost sweep: kernel (GPU clock)", cost_ms, cost_n);
        StageClock::Add("refine/normal equations: kernel (GPU clock)", neq_ms, neq_n);
        StageClock::Add("refine/residuals (count, not ms)", static_cast<double>(seg_.NumResiduals()), 1);
    }

    int BlockLength() const { return block_; }
    int NumParams() const { return block_ * seg_.n_frames; }
    int HalfBandwidth() const { return half_bandwidth_; }
    const Segment& Data() const { return seg_; }

    // LevMarqSparseSolver::TotalCost (lev_marq.h:773-824)
    double TotalCost(const CameraTrajectory& traj) {
        PackCameras(traj, cams_);
        double cost = 0.0;
        StageClock::Scope sc("refine/cost sweep");
        GpuSection section;
        if (pc_refine_total_cost(ctx_, gpu_.p, cams_.data(), loss_type_, opts_.loss_scale, &cost) != PC_OK)
            ThrowHip("pc_refine_total_cost");
        return cost;
    }

    // LevMarqSparseSolver::BuildNormalEquations (lev_marq.h:653-771) -> JtJ, Jtr, diag
    void BuildNormalEquations(const CameraTrajectory& traj) {
        PackCameras(traj, cams_);
        StageClock::Scope sc("refine/normal equations (sweep + host assembly)");
        {
            GpuSection section;
            if (pc_refine_normal_equations(ctx_, gpu_.p, cams_.data(), loss_type_, opts_.loss_scale, edge_blocks_.data(), nullptr) !=
                PC_OK)
                ThrowHip("pc_refine_normal_equations");
        }
        JtJ.SetZero();
        std::fill(Jtr.begin(), Jtr.end(), 0.0);
        const int B = block_, pair = 2 * B, tri = pair * (pair + 1) / 2;
        // fixed edge order: the same input gives the same system bit for bit (the reference scatters with
        // relaxed float atomics, lev_marq.h:718-767)
        for (int e = 0; e < seg_.NumEdges(); e++) {
            if (seg_.edge_weight[e] == 0.0f) continue;
            const double* blk = edge_blocks_.data() + static_cast<size_t>(e) * (tri + pair);
            const int base[2] = {seg_.edge_src[e] * B, seg_.edge_tgt[e] * B};
            int o = 0;
            for (int r = 0; r < pair; r++)
                for (int c = 0; c <= r; c++, o++) {
                    int gr = base[r / B] + r % B, gc = base[c / B] + c % B;
                    if (gc > gr) std::swap(gr, gc);  // cross block of an edge whose source is the later frame
                    JtJ.At(gr, gc) += blk[o];
                }
            for (int r = 0; r < pair; r++) Jtr[base[r / B] + r % B] += blk[tri + r];
        }
        for (int i = 0; i < NumParams(); i++) diag[i] = std::min(std::max(JtJ.At(i, i), 1e-6), 1e32);  // :770
    }

    BandMatrix JtJ{0, 0};
    std::vector<double> Jtr, diag;

   private:
    const BundleOptions opts_;
    const int loss_type_, block_;
    pc_context* ctx_;
    Segment seg_;
    GpuRefineProblem gpu_;
    int half_bandwidth_ = 0;
    std::vector<pc_refine_camera> cams_;
    std::vector<double> edge_blocks_;
};

}  // namespace

void RefineTrajectory(const std::string& database_path, CameraTrajectory& traj, const Mat4f& model_matrix,
                      const AcceleratedMesh& mesh, bool optimize_focal_length, bool optimize_principal_point,
                      RefineTrajectoryCallback callback, BundleOptions opts) {
    StageClock::Begin();
    numa::ScopedPin near_gpu(SharedGpuContext(), "refinement: calling thread");   // numa_pin.h
    RefineSession session(database_path, traj, model_matrix, mesh, optimize_focal_length, optimize_principal_point, opts);
    const int B = session.BlockLength(), n_params = session.NumParams();
    const CameraIntrinsics::Bounds bounds = traj.Get(traj.FirstFrame())->intrinsics.GetBounds();  // refiner.cc:690
    BandMatrix& JtJ = session.JtJ;
    BandMatrix damped = JtJ;
    std::vector<double>&Jtr = session.Jtr, &diag = session.diag;
    std::vector<double> step(n_params), JtJ_step(n_params);
    std::vector<float> step_f(n_params);

    RefineTrajectoryUpdate update;
    auto report = [&](const BundleStats& stats) {  // callback_wrapper, refiner.cc:673-681
        update.progress = static_cast<float>(stats.iterations) / opts.max_iterations;
        char msg[96];
        std::snprintf(msg, sizeof msg, "Cost: %.02f (Initial: %.02f)", static_cast<double>(stats.cost),
                      static_cast<double>(stats.initial_cost));
        update.message = msg;
        update.stats = stats;
        return callback ? callback(update) : true;
    };

    // ---- LevMarqSparseSolver::Solve (lev_marq.h:503-601) ----
    // The reference runs this loop in fp32 (Float); costs, the system and the step are fp64 here and rounded into
    // BundleStats, the camera update itself stays fp32.
    BundleStats stats;
    double cost = session.TotalCost(traj);
    stats.cost = static_cast<Float>(cost);
    stats.initial_cost = stats.cost;
    stats.grad_norm = -1;
    stats.step_norm = -1;
    stats.invalid_steps = 0;
    stats.lambda = opts.initial_lambda;

    CameraTrajectory traj_new = traj;
    Float v = 2.0f;
    bool rebuild = true;
    for (stats.iterations = 0; stats.iterations < opts.max_iterations; ++stats.iterations) {
        if (rebuild) {
            session.BuildNormalEquations(traj);
            stats.grad_norm = static_cast<Float>(Norm(Jtr));
            if (stats.grad_norm < opts.gradient_tol) break;
        }
        // ComputeStep (:826-842): damp the clamped diagonal, factorise, solve
        damped = JtJ;
        for (int i = 0; i < n_params; i++) {
            damped.At(i, i) = diag[i] * (1.0 + stats.lambda);
            JtJ.At(i, i) = diag[i];
        }
        bool factorized;
        {
            StageClock::Scope sc("refine/banded Cholesky");
            factorized = damped.Factorize();
        }
        if (!factorized) {
            stats.invalid_steps++;
            if (stats.lambda == opts.max_lambda) break;
            stats.lambda = std::min(opts.max_lambda, stats.lambda * v);
            v = 2 * v;
            rebuild = false;
            continue;
        }
        damped.Solve(Jtr, step);
        for (int i = 0; i < n_params; i++) {
            step[i] = -step[i];
            step_f[i] = static_cast<float>(step[i]);
        }
        stats.step_norm = static_cast<Float>(Norm(step));
        if (stats.step_norm < opts.step_tol) break;

        // GlobalRefinementProblem::Step (refiner.cc:618-646): the first and the last camera are constant
        for (int32_t frame = traj.FirstFrame() + 1; frame <= traj.LastFrame() - 1; frame++) {
            CameraState camera = *traj.Get(frame);
            StepCamera(camera, &step_f[static_cast<size_t>(B) * (frame - traj.FirstFrame())], optimize_focal_length,
                       optimize_principal_point, bounds);
            traj_new.Set(frame, camera);
        }
        const double cost_new = session.TotalCost(traj_new);

        if (cost_new < cost) {
            const double actual = cost_new - cost;
            JtJ.Multiply(step, JtJ_step);
            double expected = 0;
            for (int i = 0; i < n_params; i++) expected += step[i] * (2.0 * Jtr[i] + JtJ_step[i]);
            const double rho = actual / expected;
            if (rho > 0) {
                const double factor = std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rho - 1.0, 3));
                stats.lambda = std::clamp(static_cast<Float>(stats.lambda * factor), opts.min_lambda, opts.max_lambda);
            }
            for (int32_t frame = traj.FirstFrame() + 1; frame <= traj.LastFrame() - 1; frame++)
                traj.Set(frame, *traj_new.Get(frame));
            cost = cost_new;
            stats.cost = static_cast<Float>(cost_new);
            v = 2;
            rebuild = true;
        } else {
            stats.invalid_steps++;
            if (stats.lambda == opts.max_lambda) break;
            stats.lambda = std::min(opts.max_lambda, stats.lambda * v);
            v = 2 * v;
            rebuild = false;
        }
        if (!report(stats)) break;
    }
    report(stats);
    session.ReportGpuTimes();
    StageClock::Report("RefineTrajectory");
}

RefinementSystem EvaluateRefinementSystem(const std::string& database_path, const CameraTrajectory& traj,
                                          const Mat4f& model_matrix, const AcceleratedMesh& mesh, bool optimize_focal_length,
                                          bool optimize_principal_point, const BundleOptions& opts) {
    RefineSession session(database_path, traj, model_matrix, mesh, optimize_focal_length, optimize_principal_point, opts);
    RefinementSystem out;
    out.cost = static_cast<float>(session.TotalCost(traj));  // fills the triangle cache the Jacobians use
    session.BuildNormalEquations(traj);
    const int n = session.NumParams(), bw = session.HalfBandwidth();
    out.num_params = n;
    out.block_length = session.BlockLength();
    out.num_edges = session.Data().NumEdges();
    out.num_residuals = session.Data().NumResiduals();
    out.num_keypoints = session.Data().kp_offset.back();
    out.JtJ.assign(static_cast<size_t>(n) * n, 0.f);
    for (int r = 0; r < n; r++)
        for (int c = std::max(0, r - bw); c <= r; c++) out.JtJ[static_cast<size_t>(r) * n + c] = out.JtJ[static_cast<size_t>(c) * n + r] = static_cast<float>(session.JtJ.At(r, c));
    out.Jtr.assign(session.Jtr.begin(), session.Jtr.end());
    return out;
}
