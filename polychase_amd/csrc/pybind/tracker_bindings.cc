// tracker_bindings.cc -- tracking half of the reference's Python surface
// (cpp/polychase_pybind.cc:30-69, :147-171, :194-347).
#include <pybind11/functional.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "../host/band_matrix.h"
#include "../host/pnp.h"
#include "../host/ray_casting.h"
#include "../host/refining_thread.h"
#include "../host/track_sequence.h"
#include "../host/tracking_thread.h"
#include "np_helpers.h"

namespace {

py::array_t<float> Mat4ToNumpy(const Mat4f& m) {
    py::array_t<float> a({4, 4});
    std::memcpy(a.mutable_data(), m.data(), sizeof(float) * 16);
    return a;
}
Mat4f NumpyToMat4(const F32Array& a) {
    if (a.ndim() != 2 || a.shape(0) != 4 || a.shape(1) != 4) throw py::value_error("expected a 4x4 matrix");
    Mat4f m;
    std::memcpy(m.data(), a.data(), sizeof(float) * 16);
    return m;
}

struct PinUpdate {  // cpp/pin_mode.h (out of scope: only the type is kept importable)
    uint32_t pin_idx;
    Vec2f pos;
};
enum class TransformationType { Camera, Model };

[[noreturn]] void NotInThisBuild(const char* what) {
    throw std::runtime_error(std::string(what) +
                             " is not part of the MI355X hot-path build (see DESIGN.md, out of scope / next)");
}

}  // namespace

void BindTracker(py::module_& m) {
    py::class_<Mesh>(m, "Mesh")
        .def_property(
            "vertices",
            [](const Mesh& s) {
                py::array_t<float> a({static_cast<py::ssize_t>(s.NumVertices()), static_cast<py::ssize_t>(3)});
                if (!s.vertices.empty()) std::memcpy(a.mutable_data(), s.vertices.data(), s.vertices.size() * 4);
                return a;
            },
            [](Mesh& s, const F32Array& a) { s.vertices = NumpyToVec1<float>(a); })
        .def_property(
            "triangles",
            [](const Mesh& s) {
                py::array_t<uint32_t> a({static_cast<py::ssize_t>(s.NumTriangles()), static_cast<py::ssize_t>(3)});
                if (!s.triangles.empty()) std::memcpy(a.mutable_data(), s.triangles.data(), s.triangles.size() * 4);
                return a;
            },
            [](Mesh& s, const U32Array& a) { s.triangles = NumpyToVec1<uint32_t>(a); })
        .def_property(
            "masked_triangles", [](const Mesh& s) { return Vec1ToNumpy(s.masked_triangles); },
            [](Mesh& s, const U32Array& a) { s.masked_triangles = NumpyToVec1<uint32_t>(a); })
        .def("is_triangle_masked", &Mesh::IsTriangleMasked)
        .def("mask_triangle", &Mesh::MaskTriangle)
        .def("unmask_triangle", &Mesh::UnmaskTriangle)
        .def("toggle_mask_triangle", &Mesh::ToggleMaskTriangle);

    py::class_<AcceleratedMesh, std::shared_ptr<AcceleratedMesh>>(m, "AcceleratedMesh")
        .def(py::init([](const F32Array& vertices, const U32Array& triangles, const U32Array& masked) {
                 if (vertices.ndim() != 2 || vertices.shape(1) != 3) throw py::value_error("vertices must be (N, 3)");
                 if (triangles.ndim() != 2 || triangles.shape(1) != 3) throw py::value_error("triangles must be (M, 3)");
                 return std::make_shared<AcceleratedMesh>(NumpyToVec1<float>(vertices), NumpyToVec1<uint32_t>(triangles),
                                                          NumpyToVec1<uint32_t>(masked));
             }),
             py::arg("vertices"), py::arg("triangles"), py::arg("masked_triangles") = U32Array(0))
        .def("inner", &AcceleratedMesh::Inner, py::return_value_policy::reference_internal)
        .def("inner_mut", &AcceleratedMesh::InnerMut, py::return_value_policy::reference_internal);

    py::enum_<TransformationType>(m, "TransformationType")
        .value("Camera", TransformationType::Camera)
        .value("Model", TransformationType::Model);

    py::enum_<CameraConvention>(m, "CameraConvention")
        .value("OpenGL", CameraConvention::OpenGL)
        .value("OpenCV", CameraConvention::OpenCV);

    py::class_<CameraIntrinsics>(m, "CameraIntrinsics")
        .def(py::init([](float fx, float fy, float cx, float cy, float aspect_ratio, float width, float height,
                         CameraConvention convention) {
                 CameraIntrinsics k;
                 k.fx = fx; k.fy = fy; k.cx = cx; k.cy = cy;
                 k.aspect_ratio = aspect_ratio; k.width = width; k.height = height; k.convention = convention;
                 return k;
             }),
             py::arg("fx"), py::arg("fy"), py::arg("cx"), py::arg("cy"), py::arg("aspect_ratio"), py::arg("width"),
             py::arg("height"), py::arg("convention") = CameraConvention::OpenGL)
        .def_readwrite("fx", &CameraIntrinsics::fx)
        .def_readwrite("fy", &CameraIntrinsics::fy)
        .def_readwrite("cx", &CameraIntrinsics::cx)
        .def_readwrite("cy", &CameraIntrinsics::cy)
        .def_readwrite("aspect_ratio", &CameraIntrinsics::aspect_ratio)
        .def_readwrite("width", &CameraIntrinsics::width)
        .def_readwrite("height", &CameraIntrinsics::height)
        .def_readwrite("convention", &CameraIntrinsics::convention);

    py::class_<SceneTransformations>(m, "SceneTransformations")
        .def(py::init([](const F32Array& model, const F32Array& view, const CameraIntrinsics& k) {
                 SceneTransformations s;
                 s.model_matrix = NumpyToMat4(model);
                 s.view_matrix = NumpyToMat4(view);
                 s.intrinsics = k;
                 return s;
             }),
             py::arg("model_matrix"), py::arg("view_matrix"), py::arg("intrinsics"))
        .def_property(
            "model_matrix", [](const SceneTransformations& s) { return Mat4ToNumpy(s.model_matrix); },
            [](SceneTransformations& s, const F32Array& a) { s.model_matrix = NumpyToMat4(a); })
        .def_property(
            "view_matrix", [](const SceneTransformations& s) { return Mat4ToNumpy(s.view_matrix); },
            [](SceneTransformations& s, const F32Array& a) { s.view_matrix = NumpyToMat4(a); })
        .def_readwrite("intrinsics", &SceneTransformations::intrinsics);

    py::class_<RayHit>(m, "RayHit")
        .def_property(
            "pos", [](const RayHit& h) { return ArrToNumpy<3>(h.pos); },
            [](RayHit& h, const F32Array& a) { h.pos = NumpyToArr<3>(a); })
        .def_property(
            "normal", [](const RayHit& h) { return ArrToNumpy<3>(h.normal); },
            [](RayHit& h, const F32Array& a) { h.normal = NumpyToArr<3>(a); })
        .def_property(
            "barycentric_coordinate", [](const RayHit& h) { return ArrToNumpy<2>(h.barycentric_coordinate); },
            [](RayHit& h, const F32Array& a) { h.barycentric_coordinate = NumpyToArr<2>(a); })
        .def_readwrite("t", &RayHit::t)
        .def_readwrite("primitive_id", &RayHit::primitive_id);

    py::class_<PinUpdate>(m, "PinUpdate")
        .def(py::init([](uint32_t idx, const F32Array& pos) { return PinUpdate{idx, NumpyToArr<2>(pos)}; }),
             py::arg("pin_idx"), py::arg("pin_pos"))
        .def_readwrite("pin_idx", &PinUpdate::pin_idx)
        .def_property(
            "pos", [](const PinUpdate& p) { return ArrToNumpy<2>(p.pos); },
            [](PinUpdate& p, const F32Array& a) { p.pos = NumpyToArr<2>(a); });

    py::class_<Pose>(m, "Pose")
        .def(py::init<>())
        // exposed WXYZ (Blender order), stored XYZW like Eigen (polychase_pybind.cc:219-232)
        .def_property(
            "q", [](const Pose& p) { return ArrToNumpy<4>(Vec4f{p.q.w, p.q.x, p.q.y, p.q.z}); },
            [](Pose& p, const F32Array& a) {
                const Vec4f q = NumpyToArr<4>(a);
                p.q = Quatf::FromWXYZ(q[0], q[1], q[2], q[3]);
            })
        .def_property(
            "t", [](const Pose& p) { return ArrToNumpy<3>(p.t); },
            [](Pose& p, const F32Array& a) { p.t = NumpyToArr<3>(a); })
        .def("_Rt4x4", [](const Pose& p) {   // not in the reference's binding: the view matrix the tracker derives (tests)
            const Mat4f m = p.Rt4x4();
            py::array_t<float> out({py::ssize_t{4}, py::ssize_t{4}});
            std::memcpy(out.mutable_data(), m.data(), sizeof(float) * 16);
            return out;
        });

    py::class_<CameraState>(m, "CameraState")
        .def(py::init<>())
        .def(py::init([](const CameraIntrinsics& k, const Pose& p) { return CameraState{k, p}; }), py::arg("intrinsics"),
             py::arg("pose"))
        .def_readwrite("intrinsics", &CameraState::intrinsics)
        .def_readwrite("pose", &CameraState::pose);

    py::enum_<BundleOptions::LossType>(m, "LossType")
        .value("Trivial", BundleOptions::LossType::TRIVIAL)
        .value("Huber", BundleOptions::LossType::HUBER)
        .value("Cauchy", BundleOptions::LossType::CAUCHY);

    py::class_<BundleOptions>(m, "BundleOptions")
        .def(py::init<>())
        .def_readwrite("max_iterations", &BundleOptions::max_iterations)
        .def_readwrite("max_allowed_parallelism", &BundleOptions::max_allowed_parallelism)
        .def_readwrite("loss_type", &BundleOptions::loss_type)
        .def_readwrite("loss_scale", &BundleOptions::loss_scale)
        .def_readwrite("gradient_tol", &BundleOptions::gradient_tol)
        .def_readwrite("step_tol", &BundleOptions::step_tol)
        .def_readwrite("initial_lambda", &BundleOptions::initial_lambda)
        .def_readwrite("min_lambda", &BundleOptions::min_lambda)
        .def_readwrite("max_lambda", &BundleOptions::max_lambda)
        .def_readwrite("verbose", &BundleOptions::verbose);

    py::class_<BundleStats>(m, "BundleStats")
        .def(py::init<>())
        .def_readwrite("iterations", &BundleStats::iterations)
        .def_readwrite("initial_cost", &BundleStats::initial_cost)
        .def_readwrite("cost", &BundleStats::cost)
        .def_readwrite("lambda", &BundleStats::lambda)
        .def_readwrite("invalid_steps", &BundleStats::invalid_steps)
        .def_readwrite("step_norm", &BundleStats::step_norm)
        .def_readwrite("grad_norm", &BundleStats::grad_norm)
        .def("__repr__", [](const BundleStats& s) {
            return StrFormat("BundleStats(iterations=%zu, initial_cost=%g, cost=%g, lambda=%g, invalid_steps=%zu, "
                             "step_norm=%g, grad_norm=%g)",
                             s.iterations, s.initial_cost, s.cost, s.lambda, s.invalid_steps, s.step_norm, s.grad_norm);
        });

    py::class_<PnPResult>(m, "PnPResult")
        .def(py::init<>())
        .def_readwrite("camera", &PnPResult::camera)
        .def_readwrite("bundle_stats", &PnPResult::bundle_stats)
        .def_readwrite("inlier_ratio", &PnPResult::inlier_ratio);

    py::class_<FrameTrackingResult>(m, "FrameTrackingResult")
        .def_readwrite("frame", &FrameTrackingResult::frame)
        .def_readwrite("pose", &FrameTrackingResult::pose)
        .def_readwrite("intrinsics", &FrameTrackingResult::intrinsics)
        .def_readwrite("bundle_stats", &FrameTrackingResult::bundle_stats)
        .def_readwrite("inlier_ratio", &FrameTrackingResult::inlier_ratio);

    py::class_<CameraTrajectory, std::shared_ptr<CameraTrajectory>>(m, "CameraTrajectory")
        .def(py::init<int32_t, size_t>(), py::arg("first_frame_id"), py::arg("count"))
        .def("is_valid_frame", &CameraTrajectory::IsValidFrame, py::arg("frame_id"))
        .def("is_frame_filled", &CameraTrajectory::IsFrameFilled, py::arg("frame_id"))
        .def("get", &CameraTrajectory::Get, py::arg("frame_id"))
        .def("set", &CameraTrajectory::Set, py::arg("frame_id"), py::arg("state"))
        .def("count", &CameraTrajectory::Count)
        .def("first_frame", &CameraTrajectory::FirstFrame)
        .def("last_frame", &CameraTrajectory::LastFrame);

    py::class_<TrackerThread>(m, "TrackerThread")
        .def(py::init<std::string, int32_t, int32_t, SceneTransformations, std::shared_ptr<const AcceleratedMesh>, bool,
                      bool, BundleOptions>(),
             py::arg("database_path"), py::arg("frame_from"), py::arg("frame_to_inclusive"), py::arg("scene_transform"),
             py::arg("accel_mesh"), py::arg("optimize_focal_length"), py::arg("optimize_principal_point"),
             py::arg("bundle_opts"))
        .def("request_stop", &TrackerThread::RequestStop)
        .def("join", &TrackerThread::Join, py::call_guard<py::gil_scoped_release>())
        .def("try_pop", &TrackerThread::TryPop)
        .def("empty", &TrackerThread::Empty);

    py::class_<RefineTrajectoryUpdate>(m, "RefineTrajectoryUpdate")
        .def_readwrite("progress", &RefineTrajectoryUpdate::progress)
        .def_readwrite("message", &RefineTrajectoryUpdate::message)
        .def_readwrite("stats", &RefineTrajectoryUpdate::stats);

    py::class_<RefinerThread>(m, "RefinerThread")
        .def(py::init([](std::string database_path, std::shared_ptr<CameraTrajectory> traj, const F32Array& model_matrix,
                         std::shared_ptr<const AcceleratedMesh> mesh, bool opt_f, bool opt_pp, BundleOptions bundle_opts) {
                 return std::make_unique<RefinerThread>(std::move(database_path), std::move(traj), NumpyToMat4(model_matrix),
                                                        std::move(mesh), opt_f, opt_pp, bundle_opts);
             }),
             py::arg("database_path"), py::arg("camera_trajectory"), py::arg("model_matrix"), py::arg("mesh"),
             py::arg("optimize_focal_length"), py::arg("optimize_principal_point"), py::arg("bundle_opts"))
        .def("request_stop", &RefinerThread::RequestStop)
        .def("join", &RefinerThread::Join, py::call_guard<py::gil_scoped_release>())
        .def("try_pop", &RefinerThread::TryPop)
        .def("empty", &RefinerThread::Empty);

    m.def(
        "ray_cast",
        [](const AcceleratedMesh& mesh, const SceneTransformations& st, const F32Array& pos, bool check_mask) {
            return RayCast(mesh, st, NumpyToArr<2>(pos), check_mask);
        },
        py::arg("accel_mesh"), py::arg("scene_transform"), py::arg("pos"), py::arg("check_mask"));

    m.def("find_transformation", [](py::args, py::kwargs) { NotInThisBuild("find_transformation (pin mode)"); });
    m.def(
        "refine_trajectory",
        [](const std::string& database_path, CameraTrajectory& traj, const F32Array& model_matrix, const AcceleratedMesh& mesh,
           bool opt_f, bool opt_pp, py::object callback, BundleOptions bundle_opts) {
            RefineTrajectoryCallback cb;
            if (!callback.is_none())
                cb = [&](RefineTrajectoryUpdate u) {
                    py::gil_scoped_acquire gil;
                    return callback(std::move(u)).cast<bool>();
                };
            const Mat4f model = NumpyToMat4(model_matrix);
            py::gil_scoped_release release;  // polychase_pybind.cc:347
            RefineTrajectory(database_path, traj, model, mesh, opt_f, opt_pp, cb, bundle_opts);
        },
        py::arg("database_path"), py::arg("camera_trajectory"), py::arg("model_matrix"), py::arg("mesh"),
        py::arg("optimize_focal_length"), py::arg("optimize_principal_point"), py::arg("callback"),
        py::arg("bundle_opts") = BundleOptions());

    m.def(
        "track_sequence",
        [](const std::string& database_path, int32_t frame_from, int32_t frame_to_inclusive,
           const SceneTransformations& st, const AcceleratedMesh& mesh, py::object callback, bool opt_f, bool opt_pp,
           BundleOptions bundle_opts) {
            TrackingCallback cb;
            if (!callback.is_none())
                cb = [&](const FrameTrackingResult& r) {
                    py::gil_scoped_acquire gil;
                    return callback(r).cast<bool>();
                };
            py::gil_scoped_release release;  // polychase_pybind.cc:340
            TrackSequence(database_path, frame_from, frame_to_inclusive, st, mesh, cb, opt_f, opt_pp, bundle_opts);
        },
        py::arg("database_path"), py::arg("frame_from"), py::arg("frame_to_inclusive"), py::arg("scene_transform"),
        py::arg("accel_mesh"), py::arg("callback"), py::arg("optimize_focal_length") = false,
        py::arg("optimize_principal_point") = false, py::arg("bundle_opts") = BundleOptions());

    // Not in the reference's module: the refiner's cost and normal equations at a trajectory, and the banded
    // Cholesky solve it uses (tests).
    m.def(
        "_refinement_system",
        [](const std::string& database_path, const CameraTrajectory& traj, const F32Array& model_matrix,
           const AcceleratedMesh& mesh, bool opt_f, bool opt_pp, BundleOptions bundle_opts) {
            const RefinementSystem sys = EvaluateRefinementSystem(database_path, traj, NumpyToMat4(model_matrix), mesh, opt_f,
                                                                  opt_pp, bundle_opts);
            py::array_t<float> JtJ({sys.num_params, sys.num_params});
            std::memcpy(JtJ.mutable_data(), sys.JtJ.data(), sys.JtJ.size() * sizeof(float));
            py::array_t<float> Jtr(sys.num_params);
            std::memcpy(Jtr.mutable_data(), sys.Jtr.data(), sys.Jtr.size() * sizeof(float));
            py::dict d;
            d["cost"] = sys.cost;
            d["JtJ"] = JtJ;
            d["Jtr"] = Jtr;
            d["block_length"] = sys.block_length;
            d["num_edges"] = sys.num_edges;
            d["num_residuals"] = sys.num_residuals;
            d["num_keypoints"] = sys.num_keypoints;
            return d;
        },
        py::arg("database_path"), py::arg("camera_trajectory"), py::arg("model_matrix"), py::arg("mesh"),
        py::arg("optimize_focal_length"), py::arg("optimize_principal_point"), py::arg("bundle_opts") = BundleOptions());
    m.def(
        "_banded_llt_solve",
        [](const F32Array& A, int half_bandwidth, const F32Array& b) -> py::object {
            const int n = static_cast<int>(A.shape(0));
            BandMatrix M(n, half_bandwidth);
            for (int r = 0; r < n; r++)
                for (int c = std::max(0, r - half_bandwidth); c <= r; c++) M.At(r, c) = A.at(r, c);
            if (!M.Factorize()) return py::none();
            std::vector<double> rhs(b.data(), b.data() + n), x;
            M.Solve(rhs, x);
            py::array_t<double> out(n);
            std::memcpy(out.mutable_data(), x.data(), n * sizeof(double));
            return out;
        },
        py::arg("A"), py::arg("half_bandwidth"), py::arg("b"));

    // Not in the reference's module: direct access to SolvePnPIterative (cpp/pnp/solvers.h:22-29) and to the
    // batched ray cast, for tests and benchmarks.
    m.def(
        "_solve_pnp_iterative",
        [](const F32Array& object_points, const F32Array& image_points, const CameraState& initial, BundleOptions bo,
           float max_inlier_error, bool opt_f, bool opt_pp) {
            if (object_points.ndim() != 2 || object_points.shape(1) != 3 || image_points.ndim() != 2 ||
                image_points.shape(1) != 2 || object_points.shape(0) != image_points.shape(0))
                throw py::value_error("expected (N,3) object points and (N,2) image points");
            PnPResult r;
            r.camera = initial;
            PnPOptions o;
            o.bundle_opts = bo;
            o.max_inlier_error = max_inlier_error;
            o.optimize_focal_length = opt_f;
            o.optimize_principal_point = opt_pp;
            SolvePnPIterative(object_points.data(), image_points.data(), nullptr, static_cast<size_t>(object_points.shape(0)),
                              o, r);
            return r;
        },
        py::arg("object_points"), py::arg("image_points"), py::arg("initial"), py::arg("bundle_opts") = BundleOptions(),
        py::arg("max_inlier_error") = 12.0f, py::arg("optimize_focal_length") = false,
        py::arg("optimize_principal_point") = false);

    m.def(
        "_ray_cast_pixels",
        [](const AcceleratedMesh& mesh, const SceneTransformations& st, const F32Array& xy, bool check_mask, bool exhaustive) {
            if (xy.ndim() != 2 || xy.shape(1) != 2) throw py::value_error("expected (N,2) pixel positions");
            std::vector<std::optional<RayHit>> hits;
            mesh.RayCastPixels(st, xy.data(), static_cast<size_t>(xy.shape(0)), check_mask, hits, exhaustive);
            return hits;
        },
        py::arg("accel_mesh"), py::arg("scene_transform"), py::arg("xy"), py::arg("check_mask"),
        py::arg("exhaustive") = false);   // exhaustive: sweep over all triangles instead of the BVH (validation)

    // the correspondences SolveFrame builds on the GPU for one frame (tests)
    m.def(
        "_frame_correspondences",
        [](const std::string& database_path, const CameraTrajectory& traj, const F32Array& model_matrix, int32_t frame,
           const AcceleratedMesh& mesh) {
            if (model_matrix.size() != 16) throw py::value_error("expected a 4x4 model matrix");
            Mat4f model;
            std::memcpy(model.data(), model_matrix.data(), sizeof(float) * 16);
            std::vector<float> world, image;
            {
                const Database db{database_path};
                FrameCorrespondences(db, traj, model, frame, mesh, world, image);
            }
            const py::ssize_t n = static_cast<py::ssize_t>(image.size() / 2);
            py::array_t<float> w({n, py::ssize_t{3}}), x({n, py::ssize_t{2}});
            if (n) {
                std::memcpy(w.mutable_data(), world.data(), world.size() * sizeof(float));
                std::memcpy(x.mutable_data(), image.data(), image.size() * sizeof(float));
            }
            return py::make_tuple(w, x);
        },
        py::arg("database_path"), py::arg("trajectory"), py::arg("model_matrix"), py::arg("frame"), py::arg("accel_mesh"));

    // 9x9 lower Cholesky solve used by the LM step (known-answer test of
    // cpp/examples/levmarq_ill_conditioned_float32_issue.cpp)
    m.def("_llt9_solve", [](const F32Array& lower_9x9, const F32Array& rhs9) {
        if (lower_9x9.size() != 81 || rhs9.size() != 9) throw py::value_error("expected 9x9 and 9");
        float L[81], x[9];
        std::memcpy(L, lower_9x9.data(), sizeof(L));
        const bool ok = CholeskyLower<9>(L);
        if (!ok) throw std::runtime_error("matrix is not positive definite");
        CholeskySolve<9>(L, rhs9.data(), x);
        py::array_t<float> out(9);
        std::memcpy(out.mutable_data(), x, sizeof(x));
        return out;
    });
}
