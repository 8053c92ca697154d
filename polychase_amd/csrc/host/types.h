// types.h -- camera / pose / bundle types of the tracking path, same names, fields, defaults and
// formulas as the reference (cpp/pose.h, cpp/pnp/types.h, cpp/geometry.h, cpp/camera_trajectory.h,
// cpp/pnp/solvers.h, cpp/tracker.h).
#pragma once

#include <algorithm>
#include <cstdint>
#include <limits>
#include <optional>
#include <vector>

#include "linalg.h"
#include "utils.h"

enum class CameraConvention { OpenGL, OpenCV };  // pnp/types.h:13-16

struct Pose {  // cpp/pose.h:9-160
    Quatf q;
    Vec3f t{0, 0, 0};
    Pose() = default;
    Pose(const Quatf& qq, const Vec3f& tt) : q(qq), t(tt) {}
    Pose(const Mat3f& R, const Vec3f& tt) : q(Quatf::FromRotationMatrix(R)), t(tt) {}
    Mat3f R() const { return q.ToRotationMatrix(); }
    Mat4f Rt4x4() const {
        const Mat3f r = R();
        return {r[0], r[1], r[2], t[0], r[3], r[4], r[5], t[1], r[6], r[7], r[8], t[2], 0, 0, 0, 1};
    }
    Vec3f Apply(const Vec3f& p) const { return Rotate(q, p) + t; }
    Vec3f Derotate(const Vec3f& p) const { return Rotate(q.Conjugate(), p); }
    Vec3f Center() const { return -Derotate(t); }
    Pose Inverse() const { return Pose(q.Conjugate(), -Derotate(t)); }
    static Pose FromRt(const Mat4f& m) {
        return Pose(Mat3f{m[0], m[1], m[2], m[4], m[5], m[6], m[8], m[9], m[10]}, Vec3f{m[3], m[7], m[11]});
    }
};

struct CameraIntrinsics {  // pnp/types.h:18-193
    float fx = 1, fy = 1, cx = 0, cy = 0, aspect_ratio = 1, width = 0, height = 0;
    CameraConvention convention = CameraConvention::OpenGL;

    Vec2f Project(const Vec3f& x) const { return {fx * x[0] / x[2] + cx, fy * x[1] / x[2] + cy}; }
    Vec3f Unproject(const Vec2f& x) const {
        const float s = convention == CameraConvention::OpenCV ? 1.0f : -1.0f;
        return {s * ((x[0] - cx) / fx), s * ((x[1] - cy) / fy), s};
    }
    bool IsBehind(const Vec3f& p) const { return convention == CameraConvention::OpenCV ? p[2] < 0.0f : p[2] > 0.0f; }

    struct Bounds {
        Float f_low = 0, f_high = 0, cx_low = 0, cx_high = 0, cy_low = 0, cy_high = 0;
    };
    Bounds GetBounds(Float min_fov_deg = 15, Float max_fov_deg = 160) const {  // :156-192
        const Float min_fov = min_fov_deg * static_cast<Float>(M_PI) / 180;
        const Float max_fov = max_fov_deg * static_cast<Float>(M_PI) / 180;
        const Float min_tan = std::tan(min_fov / 2), max_tan = std::tan(max_fov / 2);
        Bounds b;
        if (convention == CameraConvention::OpenGL) {
            b.f_low = -(width / 2.0f) / min_tan;
            b.f_high = -(width / 2.0f) / max_tan;
        } else {
            b.f_high = (width / 2.0f) / min_tan;
            b.f_low = (width / 2.0f) / max_tan;
        }
        b.cx_low = 0.0f;
        b.cx_high = width;
        b.cy_low = 0.0f;
        b.cy_high = height;
        CHECK(b.f_low < b.f_high);
        CHECK(b.cx_low < b.cx_high);
        CHECK(b.cy_low < b.cy_high);
        return b;
    }
};

struct CameraState {
    CameraIntrinsics intrinsics;
    Pose pose;
};

struct BundleOptions {  // pnp/types.h:200-215
    size_t max_iterations = 100;
    size_t max_allowed_parallelism = 8;  // unused: the accumulation runs on the GPU
    enum class LossType { TRIVIAL, HUBER, CAUCHY } loss_type = LossType::HUBER;
    Float loss_scale = 1.0;
    Float gradient_tol = 1e-10;
    Float step_tol = 1e-8;
    Float initial_lambda = 1e-5;
    Float min_lambda = 1e-10;
    Float max_lambda = 1e10;
    bool verbose = false;
};

struct BundleStats {  // pnp/types.h:217-225
    size_t iterations = 0;
    Float initial_cost = 0;
    Float cost = 0;
    Float lambda = 0;
    size_t invalid_steps = 0;
    Float step_norm = 0;
    Float grad_norm = 0;
};

struct PnPResult {  // pnp/solvers.h:9-13
    CameraState camera;
    BundleStats bundle_stats;
    Float inlier_ratio = 0.0f;
};

struct PnPOptions {  // pnp/solvers.h:15-20
    BundleOptions bundle_opts;
    Float max_inlier_error = 0;
    bool optimize_focal_length = false;
    bool optimize_principal_point = false;
};

// cpp/geometry.h:52-152
struct Mesh {
    std::vector<float> vertices;            // N x 3
    std::vector<uint32_t> triangles;        // M x 3
    std::vector<uint32_t> masked_triangles; // bitset, padded to a multiple of 4 words

    Mesh() = default;
    Mesh(std::vector<float> v, std::vector<uint32_t> t, std::vector<uint32_t> m)
        : vertices(std::move(v)), triangles(std::move(t)), masked_triangles(std::move(m)) {
        CHECK_EQ(vertices.size() % 3, 0u);
        CHECK_EQ(triangles.size() % 3, 0u);
        const size_t n_ints = (NumTriangles() + 31) / 32;
        const size_t padded = n_ints + (4 - n_ints % 4) % 4;
        if (masked_triangles.empty()) masked_triangles.assign(padded, 0u);
        CHECK_GE(masked_triangles.size(), padded);
    }
    size_t NumVertices() const { return vertices.size() / 3; }
    size_t NumTriangles() const { return triangles.size() / 3; }
    bool IsTriangleMasked(uint32_t tri) const {
        CHECK_LT(tri / 32, masked_triangles.size());
        return (masked_triangles[tri / 32] & (1u << (tri % 32))) != 0;
    }
    void MaskTriangle(uint32_t tri) {
        CHECK_LT(tri / 32, masked_triangles.size());
        masked_triangles[tri / 32] |= (1u << (tri % 32));
    }
    void UnmaskTriangle(uint32_t tri) {
        CHECK_LT(tri / 32, masked_triangles.size());
        masked_triangles[tri / 32] &= ~(1u << (tri % 32));
    }
    void ToggleMaskTriangle(uint32_t tri) {
        CHECK_LT(tri / 32, masked_triangles.size());
        masked_triangles[tri / 32] ^= (1u << (tri % 32));
    }
};

struct SceneTransformations {  // geometry.h:156-163
    Mat4f model_matrix = Identity4();
    Mat4f view_matrix = Identity4();
    CameraIntrinsics intrinsics;
};

struct RayHit {  // ray_casting.h:15-21
    Vec3f pos{0, 0, 0};
    Vec3f normal{0, 0, 0};
    Vec2f barycentric_coordinate{0, 0};
    float t = 0;
    uint32_t primitive_id = 0;
};

// cpp/camera_trajectory.h:14-90
class CameraTrajectory {
   public:
    CameraTrajectory() = default;
    CameraTrajectory(int32_t first_frame_id, size_t count) : states_(count), first_frame_id_(first_frame_id) {}
    bool IsValidFrame(int32_t frame_id) const { return Index(frame_id) < Count(); }
    bool IsFrameFilled(int32_t frame_id) const { return IsValidFrame(frame_id) && Get(frame_id).has_value(); }
    const std::optional<CameraState>& Get(int32_t frame_id) const {
        const size_t i = Index(frame_id);
        CHECK(i < Count());
        return states_[i];
    }
    void Set(int32_t frame_id, const CameraState& s) {
        const size_t i = Index(frame_id);
        CHECK(i < Count());
        states_[i] = s;
    }
    void Clear(int32_t frame_id) {
        const size_t i = Index(frame_id);
        CHECK(i < Count());
        states_[i] = std::nullopt;
    }
    size_t Count() const { return states_.size(); }
    int32_t FirstFrame() const { return first_frame_id_; }
    int32_t LastFrame() const { return first_frame_id_ + static_cast<int32_t>(states_.size()) - 1; }
    size_t Index(int32_t frame_id) const { return static_cast<size_t>(frame_id - first_frame_id_); }

   private:
    std::vector<std::optional<CameraState>> states_;
    int32_t first_frame_id_ = 0;
};

struct FrameTrackingResult {  // cpp/tracker.h:15-21
    int32_t frame = 0;
    Pose pose;
    CameraIntrinsics intrinsics;
    BundleStats bundle_stats;
    Float inlier_ratio = 0;
};
