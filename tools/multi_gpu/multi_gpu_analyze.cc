// multi_gpu_analyze.cc -- the multi-GPU analysis launched and run by C++ alone (no Python, no torch):
//
//     polychase_multi_gpu --gpus N --database clip.db [--width 1920 --height 1080 --frames 64 --max-level 3]
//                         [--transport rccl|tcp] [--share-gpu] [--port 29611] [--piece-frames 16] [--frames-file clip.rgb]
//
// forks N ranks BEFORE the process touches HIP (one process per GPU; --share-gpu puts every rank on GPU 0, which RCCL
// refuses: use it with --transport tcp, the testing aid of csrc/host/multi_gpu.h), each rank calls
// GenerateOpticalFlowDatabaseMultiGpu over a clip every rank can produce: a deterministic procedural texture under a slow
// drift (integer arithmetic only: the same bytes on every rank and in tests/test_multi_gpu_cpp_gpu.py), or --frames-file, a
// raw file of frames x height x width x 3 bytes.  Rank 0 prints one JSON line.  The loop being sharded is the reference's
// cpp/opticalflow.cc:209-321.
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "../../polychase_amd/csrc/host/multi_gpu.h"

namespace {

// value noise on a 16-px lattice, bilinear, two octaves; integer arithmetic only
inline uint32_t Hash(uint32_t x, uint32_t y) {
    uint32_t h = x * 0x9E3779B1u ^ (y + 0x7F4A7C15u) * 0x85EBCA77u;
    h ^= h >> 15;
    h *= 0xC2B2AE3Du;
    h ^= h >> 13;
    return h;
}
inline int Lattice(int x, int y, int cell) {   // x, y >= 0, result 0..255 (fixed point bilinear of the lattice values)
    const int cx = x / cell, cy = y / cell, fx = x % cell, fy = y % cell;
    const int v00 = Hash(cx, cy) & 255, v10 = Hash(cx + 1, cy) & 255, v01 = Hash(cx, cy + 1) & 255, v11 = Hash(cx + 1, cy + 1) & 255;
    const int top = v00 * (cell - fx) + v10 * fx, bot = v01 * (cell - fx) + v11 * fx;
    return (top * (cell - fy) + bot * fy) / (cell * cell);
}
void RenderFrame(int w, int h, int t, uint8_t* rgb) {
    const int ox = 64 + t, oy = 64 + t / 2;   // the drift: one pixel per frame in x, half in y
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int v = (2 * Lattice(x + ox, y + oy, 6) + Lattice(x + ox + 1000, y + oy + 500, 17) + 1) / 3;
            uint8_t* p = rgb + (static_cast<size_t>(y) * w + x) * 3;
            p[0] = p[1] = p[2] = static_cast<uint8_t>(v);
        }
}

const char* Arg(int argc, char** argv, const char* name, const char* def) {
    for (int i = 1; i + 1 < argc; i++)
        if (std::strcmp(argv[i], name) == 0) return argv[i + 1];
    return def;
}
bool Flag(int argc, char** argv, const char* name) {
    for (int i = 1; i < argc; i++)
        if (std::strcmp(argv[i], name) == 0) return true;
    return false;
}

int RunRank(int rank, int world, int argc, char** argv) {
    const int w = std::atoi(Arg(argc, argv, "--width", "640")), h = std::atoi(Arg(argc, argv, "--height", "480"));
    const int n = std::atoi(Arg(argc, argv, "--frames", "32")), first = std::atoi(Arg(argc, argv, "--first-frame", "1"));
    const std::string db = Arg(argc, argv, "--database", "");
    const std::string file = Arg(argc, argv, "--frames-file", "");
    MultiGpuConfig cfg;
    cfg.world_size = world;
    cfg.rank = rank;
    cfg.master_port = std::atoi(Arg(argc, argv, "--port", "29611"));
    cfg.transport = Arg(argc, argv, "--transport", "rccl");
    cfg.piece_frames = std::atoi(Arg(argc, argv, "--piece-frames", "16"));
    cfg.device = Flag(argc, argv, "--share-gpu") ? 0 : rank;
    OpticalFlowOptions fo;
    fo.max_level = std::atoi(Arg(argc, argv, "--max-level", "3"));
    const size_t frame_bytes = static_cast<size_t>(w) * h * 3;
    std::ifstream in;
    if (!file.empty()) {
        in.open(file, std::ios::binary);
        if (!in) {
            std::fprintf(stderr, "cannot open %s\n", file.c_str());
            return 2;
        }
    }
    FrameAccessorFunction accessor = [&](int32_t id) -> std::optional<FrameView> {
        auto buf = std::make_shared<std::vector<uint8_t>>(frame_bytes);
        if (in.is_open()) {
            in.seekg(static_cast<std::streamoff>(frame_bytes) * (id - first));
            in.read(reinterpret_cast<char*>(buf->data()), static_cast<std::streamsize>(frame_bytes));
            if (!in) return std::nullopt;
        } else {
            RenderFrame(w, h, id - first, buf->data());
        }
        FrameView v;
        v.data = buf->data();
        v.rows = h;
        v.cols = w;
        v.channels = 3;
        v.elem_size = 1;
        v.row_pitch = static_cast<size_t>(w) * 3;
        v.owner = buf;
        return v;
    };
    try {
        const VideoInfo vi{static_cast<uint32_t>(w), static_cast<uint32_t>(h), first, static_cast<uint32_t>(n)};
        const MultiGpuResult r = GenerateOpticalFlowDatabaseMultiGpu(vi, accessor, nullptr, db, cfg, GFTTOptions{}, fo);
        if (rank == 0)
            std::printf("{\"world_size\": %d, \"transport\": \"%s\", \"frames\": %d, \"width\": %d, \"height\": %d, \"pieces_received\": %d, "
                        "\"bytes_received\": %zu, \"keypoint_rows\": %d, \"flow_rows\": %d, \"seconds_analysis_rank0\": %.4f, "
                        "\"seconds_database\": %.4f, \"seconds_total\": %.4f, \"frames_per_s\": %.2f, \"cancelled\": %s}\n",
                        world, cfg.transport.c_str(), n, w, h, r.pieces, r.bytes_moved, r.stats.keypoint_rows_written,
                        r.stats.flow_rows_written, r.seconds_analysis, r.stats.seconds_db, r.seconds_total, n / r.seconds_total,
                        r.cancelled ? "true" : "false");
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "[rank %d] %s\n", rank, e.what());
        return 1;
    }
}

}  // namespace

int main(int argc, char** argv) {
    const int world = std::atoi(Arg(argc, argv, "--gpus", "1"));
    if (world < 1 || std::string(Arg(argc, argv, "--database", "")).empty()) {
        std::fprintf(stderr, "usage: %s --gpus N --database clip.db [--width W --height H --frames n --max-level L --transport rccl|tcp "
                             "--share-gpu --port P --piece-frames k --frames-file raw.rgb]\n", argv[0]);
        return 2;
    }
    if (const char* r = std::getenv("POLYCHASE_RANK")) return RunRank(std::atoi(r), world, argc, argv);   // launched by something else
    std::remove(Arg(argc, argv, "--database", ""));
    std::vector<pid_t> kids;
    for (int r = 1; r < world; r++) {
        const pid_t pid = fork();            // before any HIP call: every rank initialises its own runtime
        if (pid == 0) return RunRank(r, world, argc, argv);
        if (pid < 0) {
            std::perror("fork");
            return 2;
        }
        kids.push_back(pid);
    }
    int rc = RunRank(0, world, argc, argv);
    for (pid_t k : kids) {
        int st = 0;
        waitpid(k, &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = rc ? rc : 1;
    }
    return rc;
}
