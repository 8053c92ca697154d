// module.cc -- the `polychase_core` extension module with the reference's Python surface
// (cpp/polychase_pybind.cc:29-348), backed by the MI355X path.  Blender's addon imports it with
// `from polychase_core import *` (blender_addon/core.py:16-22) and is meant to run unchanged.
#include <pybind11/functional.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <deque>

#include "../../../include/polychase_hip.h"
#include "../host/debug_images.h"
#include "../host/async_write_vfs.h"
#include "../host/flow_database.h"
#include "../host/analysis.h"
#include "../host/frame_pool.h"
#include "../host/analysis_thread.h"
#include "../host/multi_gpu.h"
#include "../host/stage_clock.h"
#include <chrono>
#include <thread>

#include "../host/numa_pin.h"
#include "../host/track_sequence.h"
#include "np_helpers.h"

#ifdef PC_WITH_TRACKER
void BindTracker(py::module_& m);  // tracker_bindings.cc
#endif

namespace {

using U8Array = py::array_t<uint8_t, py::array::c_style>;

// Python frame (numpy H x W x 3 uint8, or a torch CUDA tensor of that shape) -> FrameView.
// Host arrays are deep-copied under the GIL; device tensors are referenced (the caller keeps the
// py::object alive, see GenerateOpticalFlowDatabasePy).
std::optional<FrameView> FrameFromPython(const py::object& obj, std::deque<py::object>& keep_alive) {
    if (obj.is_none()) return std::nullopt;
    if (py::hasattr(obj, "is_cuda") && obj.attr("is_cuda").cast<bool>()) {
        py::object t = obj;
        if (!t.attr("is_contiguous")().cast<bool>()) t = t.attr("contiguous")();
        // the analysis runs on its own HIP stream: wait for whatever produced the tensor
        py::module_::import("torch").attr("cuda").attr("current_stream")(t.attr("device")).attr("synchronize")();
        const auto shape = t.attr("shape").cast<std::vector<py::ssize_t>>();
        if (shape.size() != 3) throw py::value_error("frame must have shape (H, W, 3)");
        const std::string dtype = py::str(t.attr("dtype")).cast<std::string>();
        if (dtype != "torch.uint8" && dtype != "torch.float32") throw py::value_error("frame must be uint8 or float32");
        FrameView v;
        v.elem_size = dtype == "torch.float32" ? 4 : 1;
        v.data = reinterpret_cast<const uint8_t*>(t.attr("data_ptr")().cast<uintptr_t>());
        v.rows = static_cast<int>(shape[0]);
        v.cols = static_cast<int>(shape[1]);
        v.channels = static_cast<int>(shape[2]);
        v.row_pitch = static_cast<size_t>(shape[1] * shape[2]) * v.elem_size;
        v.on_device = true;
        keep_alive.push_back(t);
        while (keep_alive.size() > 40) keep_alive.pop_front();
        return v;
    }
    if (py::isinstance<py::array>(obj) && py::array(obj).dtype().is(py::dtype::of<float>())) {
        // Blender's float pixels: converted on the GPU (no numpy `(x * 255).astype(uint8)` pass on the host)
        F32Array fa = F32Array::ensure(obj);
        if (!fa || fa.ndim() != 3) throw py::value_error("float frame must have shape (H, W, 3|4)");
        std::shared_ptr<void> buf = AcquirePinnedFrameBuffer(static_cast<size_t>(fa.size()) * sizeof(float));
        {
            py::gil_scoped_release nogil;   // `fa` keeps the array alive
            CopyFrameBytes(buf.get(), fa.data(), static_cast<size_t>(fa.size()) * sizeof(float));
        }
        FrameView v;
        v.on_device = true;   // pinned host memory (frame_pool.h)
        v.pinned_host = true;
        v.data = static_cast<const uint8_t*>(buf.get());
        v.rows = static_cast<int>(fa.shape(0));
        v.cols = static_cast<int>(fa.shape(1));
        v.channels = static_cast<int>(fa.shape(2));
        v.elem_size = 4;
        v.row_pitch = static_cast<size_t>(fa.shape(1) * fa.shape(2)) * sizeof(float);
        v.owner = buf;
        return v;
    }
    U8Array a = U8Array::ensure(obj);
    if (!a || a.ndim() != 3) throw py::value_error("frame must be a uint8 array of shape (H, W, 3)");
    std::shared_ptr<void> buf = AcquirePinnedFrameBuffer(static_cast<size_t>(a.size()));
    {
        py::gil_scoped_release nogil;   // `a` keeps the array alive
        CopyFrameBytes(buf.get(), a.data(), static_cast<size_t>(a.size()));
    }
    FrameView v;
    v.on_device = true;   // pinned host memory (frame_pool.h)
    v.pinned_host = true;
    v.data = static_cast<const uint8_t*>(buf.get());
    v.rows = static_cast<int>(a.shape(0));
    v.cols = static_cast<int>(a.shape(1));
    v.channels = static_cast<int>(a.shape(2));
    v.row_pitch = static_cast<size_t>(a.shape(1) * a.shape(2));
    v.owner = buf;
    return v;
}

OpticalFlowRunStats GenerateOpticalFlowDatabasePy(const VideoInfo& video_info, py::object frame_accessor,
                                                  py::object callback, const std::string& database_path,
                                                  const GFTTOptions& detector_options,
                                                  const OpticalFlowOptions& flow_options, bool write_images) {
    std::deque<py::object> keep_alive;
    FrameAccessorFunction accessor;
    if (!frame_accessor.is_none())
        accessor = [&](int32_t frame_id) -> std::optional<FrameView> {
            py::gil_scoped_acquire gil;
            return FrameFromPython(frame_accessor(frame_id), keep_alive);
        };
    OpticalFlowProgressCallback cb;
    if (!callback.is_none())
        cb = [&](float progress, const std::string& msg) -> bool {
            py::gil_scoped_acquire gil;
            return callback(progress, msg).cast<bool>();
        };
    OpticalFlowRunStats stats;
    {
        py::gil_scoped_release release;  // polychase_pybind.cc:332
        GenerateOpticalFlowDatabase(video_info, accessor, cb, database_path, detector_options, flow_options,
                                    write_images, &stats);
    }
    keep_alive.clear();
    return stats;
}

// one shard of a multi-process analysis: records into a device log (a torch uint8 CUDA tensor's memory)
py::tuple GenerateOpticalFlowRecordsPy(const VideoInfo& video_info, py::object frame_accessor, py::object callback,
                                       int32_t shard_begin, int32_t shard_end, uintptr_t device_log, size_t capacity_bytes,
                                       const GFTTOptions& detector_options, const OpticalFlowOptions& flow_options) {
    std::deque<py::object> keep_alive;
    FrameAccessorFunction accessor;
    if (!frame_accessor.is_none())
        accessor = [&](int32_t frame_id) -> std::optional<FrameView> {
            py::gil_scoped_acquire gil;
            return FrameFromPython(frame_accessor(frame_id), keep_alive);
        };
    OpticalFlowProgressCallback cb;
    if (!callback.is_none())
        cb = [&](float progress, const std::string& msg) -> bool {
            py::gil_scoped_acquire gil;
            return callback(progress, msg).cast<bool>();
        };
    OpticalFlowRunStats stats;
    size_t used = 0;
    {
        py::gil_scoped_release release;
        used = GenerateOpticalFlowRecords(video_info, accessor, cb, shard_begin, shard_end, reinterpret_cast<void*>(device_log),
                                          capacity_bytes, detector_options, flow_options, &stats);
    }
    keep_alive.clear();
    return py::make_tuple(used, stats);
}

// one shard with everything the multi-rank driver needs: records into the database (the rank that owns the file) and / or
// into a device log that is handed over piece by piece (on_piece(piece, offset_bytes, bytes, first_frame1, n_frames),
// called with the GIL on the driver's thread; the part of the log may be reused when it returns)
py::dict GenerateOpticalFlowShardPy(const VideoInfo& video_info, py::object frame_accessor, py::object callback,
                                    const std::string& database_path, int32_t shard_begin, int32_t shard_end, uintptr_t device_log,
                                    size_t capacity_bytes, int log_buffers, int piece_frames, py::object on_piece, bool host_records,
                                    const GFTTOptions& detector_options, const OpticalFlowOptions& flow_options) {
    std::deque<py::object> keep_alive;
    FrameAccessorFunction accessor;
    if (!frame_accessor.is_none())
        accessor = [&](int32_t frame_id) -> std::optional<FrameView> {
            py::gil_scoped_acquire gil;
            return FrameFromPython(frame_accessor(frame_id), keep_alive);
        };
    OpticalFlowProgressCallback cb;
    if (!callback.is_none())
        cb = [&](float progress, const std::string& msg) -> bool {
            py::gil_scoped_acquire gil;
            return callback(progress, msg).cast<bool>();
        };
    OpticalFlowShard shard;
    shard.begin = shard_begin;
    shard.end = shard_end;
    shard.device_log = reinterpret_cast<void*>(device_log);
    shard.capacity_bytes = capacity_bytes;
    shard.log_buffers = log_buffers;
    shard.piece_frames = piece_frames;
    shard.host_records = host_records;
    if (!on_piece.is_none())
        shard.on_piece = [&](int piece, size_t offset, size_t bytes, int32_t first_frame1, int n_frames) {
            py::gil_scoped_acquire gil;
            on_piece(piece, offset, bytes, first_frame1, n_frames);
        };
    OpticalFlowRunStats stats;
    {
        py::gil_scoped_release release;
        GenerateOpticalFlowShard(video_info, accessor, cb, database_path, shard, detector_options, flow_options, &stats);
    }
    keep_alive.clear();
    py::dict out;
    out["used_bytes"] = shard.used_bytes;
    out["pieces"] = shard.pieces;
    out["cancelled"] = shard.cancelled;
    out["stats"] = stats;
    return out;
}

// GenerateOpticalFlowDatabaseMultiGpu (csrc/host/multi_gpu.h): this rank's part of the multi-GPU analysis, RCCL + TCP control
// channel underneath, no torch.  Not in the reference (its analysis is one process).
py::dict GenerateOpticalFlowDatabaseMultiGpuPy(const VideoInfo& video_info, py::object frame_accessor, py::object callback,
                                               const std::string& database_path, int world_size, int rank, const std::string& master_addr,
                                               int master_port, int device, int piece_frames, const std::string& transport,
                                               size_t keypoints_per_frame, const GFTTOptions& detector_options,
                                               const OpticalFlowOptions& flow_options) {
    std::deque<py::object> keep_alive;
    FrameAccessorFunction accessor;
    if (!frame_accessor.is_none())
        accessor = [&](int32_t frame_id) -> std::optional<FrameView> {
            py::gil_scoped_acquire gil;
            return FrameFromPython(frame_accessor(frame_id), keep_alive);
        };
    OpticalFlowProgressCallback cb;
    if (!callback.is_none())
        cb = [&](float progress, const std::string& msg) -> bool {
            py::gil_scoped_acquire gil;
            return callback(progress, msg).cast<bool>();
        };
    MultiGpuConfig cfg;
    cfg.world_size = world_size;
    cfg.rank = rank;
    cfg.master_addr = master_addr;
    cfg.master_port = master_port;
    cfg.device = device;
    cfg.piece_frames = piece_frames;
    cfg.transport = transport;
    cfg.keypoints_per_frame = keypoints_per_frame;
    MultiGpuResult r;
    {
        py::gil_scoped_release release;
        r = GenerateOpticalFlowDatabaseMultiGpu(video_info, accessor, cb, database_path, cfg, detector_options, flow_options);
    }
    keep_alive.clear();
    py::dict out;
    out["shard"] = py::make_tuple(r.shard_begin, r.shard_end);
    out["pieces"] = r.pieces;
    out["bytes_moved"] = r.bytes_moved;
    out["cancelled"] = r.cancelled;
    out["seconds_analysis"] = r.seconds_analysis;
    out["seconds_total"] = r.seconds_total;
    out["seconds_blocked"] = r.seconds_blocked;
    out["stats"] = r.stats;
    return out;
}

OpticalFlowRunStats WriteOpticalFlowRecordsPy(const std::string& database_path, const U8Array& log, size_t bytes) {
    if (bytes > static_cast<size_t>(log.size())) throw py::value_error("bytes exceeds the buffer");
    OpticalFlowRunStats stats;
    {
        py::gil_scoped_release release;
        WriteOpticalFlowRecords(database_path, log.data(), bytes, &stats);
    }
    return stats;
}

}  // namespace

PYBIND11_MODULE(polychase_core, m) {
    m.doc() = "polychase_core on MI355X: drop-in for the reference's pybind11 module (video-analysis path)";
    // a hardware queue per stream of the engine: must be in the environment before the process first touches HIP -- importing
    // the module is the earliest moment the add-on gives us (include/polychase_hip.h: pc_runtime_init)
    {
        int was_up = 0, queues = 0;
        pc_runtime_init(&was_up, &queues);
        m.attr("_runtime_was_up_at_import") = was_up != 0;   // not in the reference (diagnostics)
        m.attr("_gpu_max_hw_queues") = queues;
    }

    m.def("_async_write_counters", [] {   // not in the reference (tests, diagnostics): totals of csrc/host/async_write_vfs.h
        const AsyncWriteVfsCounters c = AsyncWriteVfsTotals();
        py::dict d;
        d["deferred_writes"] = c.deferred_writes;
        d["deferred_bytes"] = c.deferred_bytes;
        d["direct_writes"] = c.direct_writes;
        d["drains"] = c.drains;
        d["drains_that_waited"] = c.drains_that_waited;
        d["waits_for_a_slab"] = c.waits_for_a_slab;
        return d;
    });
    py::class_<Database>(m, "Database")
        .def(py::init<const std::string&>(), py::arg("path"))
        // not in the reference (tests): the connection mode of the analysis' writer, csrc/host/async_write_vfs.h
        .def_static("_open_bulk_writer", [](const std::string& path) { return std::make_unique<Database>(path, true); }, py::arg("path"))
        .def("open", [](Database& db, const std::string& path) { db.Open(path); }, py::arg("path"))
        .def("close", &Database::Close)
        .def("_set_journal_mode", &Database::SetJournalMode, py::arg("mode"))   // not in the reference (tests)
        .def("_begin", &Database::Begin)                                          // ... explicit transactions (tests)
        .def("_commit", &Database::Commit)
        .def("_rollback", &Database::Rollback)
        .def("read_keypoints", [](const Database& db, int32_t id) { return VecToNumpy<2>(db.ReadKeypoints(id)); },
             py::arg("image_id"))
        .def("write_keypoints",
             [](Database& db, int32_t id, const F32Array& kps) { db.WriteKeypoints(id, NumpyToVec<2>(kps)); },
             py::arg("image_id"), py::arg("keypoints"))
        .def("read_image_pair_flow",
             py::overload_cast<int32_t, int32_t>(&Database::ReadImagePairFlow, py::const_), py::arg("image_id_from"),
             py::arg("image_id_to"))
        .def("write_image_pair_flow",
             [](Database& db, int32_t from, int32_t to, const U32Array& idx, const F32Array& tgt, const F32Array& err) {
                 db.WriteImagePairFlow(from, to, NumpyToVec1<uint32_t>(idx), NumpyToVec<2>(tgt), NumpyToVec1<float>(err));
             },
             py::arg("image_id_from"), py::arg("image_id_to"), py::arg("src_kps_indices"), py::arg("tgt_kps"),
             py::arg("flow_errors"))
        .def("write_image_pair_flow", py::overload_cast<const ImagePairFlow&>(&Database::WriteImagePairFlow),
             py::arg("image_pair_flow"))
        .def("find_optical_flows_from_image",
             py::overload_cast<int32_t>(&Database::FindOpticalFlowsFromImage, py::const_), py::arg("image_id_from"))
        .def("find_optical_flows_to_image", py::overload_cast<int32_t>(&Database::FindOpticalFlowsToImage, py::const_),
             py::arg("image_id_to"))
        .def("keypoints_exist", &Database::KeypointsExist, py::arg("image_id"))
        .def("image_pair_flow_exists", &Database::ImagePairFlowExists, py::arg("image_id_from"),
             py::arg("image_id_to"))
        .def("get_min_image_id_with_keypoints", &Database::GetMinImageIdWithKeypoints)
        .def("get_max_image_id_with_keypoints", &Database::GetMaxImageIdWithKeypoints);

    py::class_<ImagePairFlow>(m, "ImagePairFlow")
        .def(py::init<>())
        .def_readwrite("image_id_from", &ImagePairFlow::image_id_from)
        .def_readwrite("image_id_to", &ImagePairFlow::image_id_to)
        .def_property(
            "src_kps_indices", [](const ImagePairFlow& f) { return Vec1ToNumpy(f.src_kps_indices); },
            [](ImagePairFlow& f, const U32Array& a) { f.src_kps_indices = NumpyToVec1<uint32_t>(a); })
        .def_property(
            "tgt_kps", [](const ImagePairFlow& f) { return VecToNumpy<2>(f.tgt_kps); },
            [](ImagePairFlow& f, const F32Array& a) { f.tgt_kps = NumpyToVec<2>(a); })
        .def_property(
            "flow_errors", [](const ImagePairFlow& f) { return Vec1ToNumpy(f.flow_errors); },
            [](ImagePairFlow& f, const F32Array& a) { f.flow_errors = NumpyToVec1<float>(a); });

    py::class_<VideoInfo>(m, "VideoInfo")
        .def(py::init([](uint32_t width, uint32_t height, uint32_t first_frame, uint32_t num_frames) {
                 return VideoInfo{width, height, static_cast<int32_t>(first_frame), num_frames};
             }),
             py::arg("width"), py::arg("height"), py::arg("first_frame"), py::arg("num_frames"))
        .def_readwrite("width", &VideoInfo::width)
        .def_readwrite("height", &VideoInfo::height)
        .def_readwrite("first_frame", &VideoInfo::first_frame)
        .def_readwrite("num_frames", &VideoInfo::num_frames);

    // grid_rows / grid_cols are not exposed by the reference either (polychase_pybind.cc:128-136)
    py::class_<GFTTOptions>(m, "GFTTOptions")
        .def(py::init<>())
        .def_readwrite("quality_level", &GFTTOptions::quality_level)
        .def_readwrite("min_distance", &GFTTOptions::min_distance)
        .def_readwrite("block_size", &GFTTOptions::block_size)
        .def_readwrite("gradient_size", &GFTTOptions::gradient_size)
        .def_readwrite("max_corners", &GFTTOptions::max_corners)
        .def_readwrite("use_harris", &GFTTOptions::use_harris)
        .def_readwrite("harris_k", &GFTTOptions::harris_k);

    py::class_<OpticalFlowOptions>(m, "OpticalFlowOptions")
        .def(py::init<>())
        .def_readwrite("window_size", &OpticalFlowOptions::window_size)
        .def_readwrite("max_level", &OpticalFlowOptions::max_level)
        .def_readwrite("term_max_iters", &OpticalFlowOptions::term_max_iters)
        .def_readwrite("term_epsilon", &OpticalFlowOptions::term_epsilon)
        .def_readwrite("min_eigen_threshold", &OpticalFlowOptions::min_eigen_threshold);

    py::class_<OpticalFlowProgress>(m, "OpticalFlowProgress")
        .def_readonly("progress", &OpticalFlowProgress::progress)
        .def_readonly("progress_message", &OpticalFlowProgress::progress_message);

    py::class_<OpticalFlowRequest>(m, "OpticalFlowRequest").def_readonly("frame_id", &OpticalFlowRequest::frame_id);

    py::class_<CppException>(m, "CppException").def("what", [](const CppException& e) { return e.message; });

    py::class_<OpticalFlowRunStats>(m, "OpticalFlowRunStats")
        .def_readonly("frames_processed", &OpticalFlowRunStats::frames_processed)
        .def_readonly("keypoint_rows_written", &OpticalFlowRunStats::keypoint_rows_written)
        .def_readonly("flow_rows_written", &OpticalFlowRunStats::flow_rows_written)
        .def_readonly("seconds_total", &OpticalFlowRunStats::seconds_total)
        .def_readonly("seconds_db", &OpticalFlowRunStats::seconds_db)
        .def_readonly("seconds_setup", &OpticalFlowRunStats::seconds_setup)
        .def_readonly("engine_reused", &OpticalFlowRunStats::engine_reused)
        .def_readonly("seconds_accessor", &OpticalFlowRunStats::seconds_accessor)
        .def_readonly("seconds_put", &OpticalFlowRunStats::seconds_put)
        .def_readonly("seconds_submit", &OpticalFlowRunStats::seconds_submit)
        .def_readonly("seconds_collect", &OpticalFlowRunStats::seconds_collect)
        .def_readonly("seconds_writer_wait", &OpticalFlowRunStats::seconds_writer_wait);

    py::class_<OpticalFlowThread>(m, "OpticalFlowThread")
        .def(py::init<VideoInfo, std::string, GFTTOptions, OpticalFlowOptions, bool>(), py::arg("video_info"),
             py::arg("database_path"), py::arg("detector_options") = GFTTOptions{},
             py::arg("OpticalFlowOptions") = OpticalFlowOptions{},  // sic: polychase_pybind.cc:186
             py::arg("write_images") = false)
        .def("request_stop", &OpticalFlowThread::RequestStop)
        .def("join", &OpticalFlowThread::Join, py::call_guard<py::gil_scoped_release>())
        .def("try_pop", &OpticalFlowThread::TryPop)
        .def("empty", &OpticalFlowThread::Empty)
        .def("provide_frame", [](OpticalFlowThread& t, int32_t frame_id, const py::array& frame) {
            if (frame.ndim() != 3) throw py::value_error("frame must be an array of shape (H, W, 3)");
            if (frame.dtype().is(py::dtype::of<float>())) {   // Blender's float pixels, (H, W, 3|4)
                const F32Array f = F32Array::ensure(frame);
                t.ProvideFrame(frame_id, reinterpret_cast<const uint8_t*>(f.data()), static_cast<int>(f.shape(0)),
                               static_cast<int>(f.shape(1)), static_cast<int>(f.shape(2)),
                               static_cast<size_t>(f.shape(1) * f.shape(2)) * sizeof(float), 4);
                return;
            }
            const U8Array u = U8Array::ensure(frame);
            if (!u) throw py::value_error("frame must be a uint8 (H, W, 3) or float32 (H, W, 3|4) array");
            t.ProvideFrame(frame_id, u.data(), static_cast<int>(u.shape(0)), static_cast<int>(u.shape(1)),
                           static_cast<int>(u.shape(2)), static_cast<size_t>(u.shape(1) * u.shape(2)));
        });

    // not in the reference: the two halves of a multi-GPU analysis (polychase_amd/analyze.py launches them, one
    // process per GPU, and all-gathers the record logs over RCCL in between)
    m.def("generate_optical_flow_records", &GenerateOpticalFlowRecordsPy, py::arg("video_info"), py::arg("frame_accessor_function"),
          py::arg("callback"), py::arg("shard_begin"), py::arg("shard_end"), py::arg("device_log"), py::arg("capacity_bytes"),
          py::arg("detector_options") = GFTTOptions{}, py::arg("flow_options") = OpticalFlowOptions{});
    // not in the reference's module: its SaveImageForDebugging (cpp/opticalflow.cc:80-96) by itself, for the tests
    m.def("_save_image_for_debugging", [](const U8Array& rgb, int32_t frame_id, const std::string& dir, const F32Array& kps) {
        if (rgb.ndim() != 3 || rgb.shape(2) != 3) throw py::value_error("rgb must be (H, W, 3) uint8");
        std::vector<uint8_t> img(rgb.data(), rgb.data() + rgb.size());
        SaveImageForDebugging(img.data(), static_cast<int>(rgb.shape(1)), static_cast<int>(rgb.shape(0)), frame_id, dir, kps.data(),
                              static_cast<int>(kps.size() / 2));
    });
    m.def("_engine_cache_timer_running", &EngineCacheTimerRunning);   // testing aid
    // measurement aid (not in the reference): the stage totals of the last finished call of that kind -- "TrackCameraTrajectory",
    // "RefineTrajectory" -- as {label: (milliseconds or count, calls)} (csrc/host/stage_clock.h)
    m.def("_stage_report", [](const std::string& title) {
        py::dict d;
        for (const auto& kv : StageClock::Last(title)) d[py::str(kv.first)] = py::make_tuple(kv.second.first, kv.second.second);
        return d;
    });
    // measurement aid (not in the reference): where the library's host threads were put (csrc/host/numa_pin.h), as a JSON string
    m.def("_thread_placement", [] { return numa::Placement(); });
    // not in the reference: gives back what the calls keep for the next one -- the parked analysis engine, the tracker's parked
    // correspondence set and its pool of page-locked blocks
    m.def("release_cached_engine", [] {
        ReleaseCachedEngine();
        ReleaseTrackerCaches();
    });
    m.def("generate_optical_flow_shard", &GenerateOpticalFlowShardPy, py::arg("video_info"), py::arg("frame_accessor_function"),
          py::arg("callback"), py::arg("database_path"), py::arg("shard_begin"), py::arg("shard_end"), py::arg("device_log") = 0,
          py::arg("capacity_bytes") = 0, py::arg("log_buffers") = 1, py::arg("piece_frames") = 0, py::arg("on_piece") = py::none(),
          py::arg("host_records") = true, py::arg("detector_options") = GFTTOptions{}, py::arg("flow_options") = OpticalFlowOptions{});
    m.def("generate_optical_flow_database_multi_gpu", &GenerateOpticalFlowDatabaseMultiGpuPy, py::arg("video_info"),
          py::arg("frame_accessor_function"), py::arg("callback"), py::arg("database_path"), py::arg("world_size"), py::arg("rank"),
          py::arg("master_addr") = "127.0.0.1", py::arg("master_port") = 29611, py::arg("device") = -1, py::arg("piece_frames") = 16,
          py::arg("transport") = "rccl", py::arg("keypoints_per_frame") = 0, py::arg("detector_options") = GFTTOptions{},
          py::arg("flow_options") = OpticalFlowOptions{});
    // Testing aid (not in the reference): GenerateOpticalFlowDatabaseMultiGpu's protocol -- credits, headers, ordered pieces, failure
    // propagation, the cancelled flag -- with the rank's shard given as ready-made record logs instead of an analysis
    // (MultiGpuConfig::synthetic_shard, transport "tcp"): runs on a box without a GPU (tests/test_distributed_cpu.py).
    // pieces: [(bytes-like log, first_frame1, n_frames)]; delay_ms: sleep before every piece; fail_after: raise after that many.
    m.def("_multi_gpu_protocol_selftest",
          [](int world_size, int rank, int master_port, const std::string& database_path, py::list pieces, int delay_ms, int fail_after) {
              struct Piece {
                  std::string bytes;
                  int32_t first;
                  int frames;
              };
              std::vector<Piece> ps;
              for (const py::handle& h : pieces) {
                  const py::tuple t = h.cast<py::tuple>();
                  ps.push_back(Piece{t[0].cast<py::bytes>(), t[1].cast<int32_t>(), t[2].cast<int>()});
              }
              MultiGpuConfig cfg;
              cfg.world_size = world_size;
              cfg.rank = rank;
              cfg.master_port = master_port;
              cfg.transport = "tcp";
              cfg.connect_timeout_s = 60.0;
              cfg.synthetic_shard = [&](int, const std::function<void(const void*, size_t, int32_t, int)>& emit) {
                  int sent = 0;
                  for (const Piece& p : ps) {
                      if (fail_after >= 0 && sent == fail_after) throw std::runtime_error("synthetic shard failed on purpose");
                      if (delay_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(delay_ms));
                      emit(p.bytes.data(), p.bytes.size(), p.first, p.frames);
                      sent++;
                  }
                  if (fail_after >= 0 && sent == fail_after) throw std::runtime_error("synthetic shard failed on purpose");
              };
              MultiGpuResult r;
              {
                  py::gil_scoped_release release;
                  r = GenerateOpticalFlowDatabaseMultiGpu(VideoInfo{}, nullptr, nullptr, database_path, cfg);
              }
              py::dict out;
              out["pieces"] = r.pieces;
              out["bytes_moved"] = r.bytes_moved;
              out["cancelled"] = r.cancelled;
              out["keypoint_rows_written"] = r.stats.keypoint_rows_written;
              out["flow_rows_written"] = r.stats.flow_rows_written;
              return out;
          },
          py::arg("world_size"), py::arg("rank"), py::arg("master_port"), py::arg("database_path"), py::arg("pieces"),
          py::arg("delay_ms") = 0, py::arg("fail_after") = -1);
    py::class_<OpticalFlowRecordWriter>(m, "OpticalFlowRecordWriter")
        .def(py::init<const std::string&>(), py::arg("database_path"))
        .def("write",
             [](OpticalFlowRecordWriter& w, const U8Array& log, size_t bytes) {
                 if (bytes > static_cast<size_t>(log.size())) throw py::value_error("bytes exceeds the buffer");
                 OpticalFlowRunStats stats;
                 {
                     py::gil_scoped_release release;
                     w.Write(log.data(), bytes, &stats);
                 }
                 return stats;
             },
             py::arg("log"), py::arg("bytes"))
        .def("close", &OpticalFlowRecordWriter::Close);
    m.def("write_optical_flow_records", &WriteOpticalFlowRecordsPy, py::arg("database_path"), py::arg("log"), py::arg("bytes"));
    m.def("generate_optical_flow_database", &GenerateOpticalFlowDatabasePy, py::arg("video_info"),
          py::arg("frame_accessor_function"), py::arg("callback"), py::arg("database_path"),
          py::arg("detector_options") = GFTTOptions{}, py::arg("flow_options") = OpticalFlowOptions{},
          py::arg("write_images") = false);

#ifdef PC_WITH_TRACKER
    BindTracker(m);
#endif
}
