// debug_images.cc -- see debug_images.h.  OpenCV pieces restated from memory of OpenCV 4.x (no OpenCV in this image):
// cv::RNG (core/operations.hpp: multiply-with-carry, coefficient 4164903690, default state 0xffffffff),
// cv::drawMarker MARKER_CROSS (imgproc/drawing.cpp: two 1-px lines of markerSize / 2 on both sides of the position).
#include "debug_images.h"

#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>

namespace {

void PutU32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back(static_cast<uint8_t>(x >> 24));
    v.push_back(static_cast<uint8_t>(x >> 16));
    v.push_back(static_cast<uint8_t>(x >> 8));
    v.push_back(static_cast<uint8_t>(x));
}

void PutChunk(std::vector<uint8_t>& out, const char type[4], const uint8_t* data, size_t n) {
    PutU32(out, static_cast<uint32_t>(n));
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    if (n) out.insert(out.end(), data, data + n);
    PutU32(out, static_cast<uint32_t>(crc32(0L, out.data() + at, static_cast<uInt>(n + 4))));
}

struct CvRng {   // cv::RNG
    uint64_t state = 0xffffffffull;
    unsigned Next() {
        state = static_cast<uint64_t>(static_cast<unsigned>(state)) * 4164903690u + static_cast<unsigned>(state >> 32);
        return static_cast<unsigned>(state);
    }
    unsigned operator()(unsigned n) { return Next() % n; }
};

}  // namespace

void WritePngRgb(const std::string& path, const uint8_t* rgb, int width, int height, size_t pitch) {
    if (width <= 0 || height <= 0) throw std::runtime_error("WritePngRgb: empty image");
    // scanlines with filter type 0 in front
    const size_t row = static_cast<size_t>(width) * 3;
    std::vector<uint8_t> raw((row + 1) * static_cast<size_t>(height));
    for (int y = 0; y < height; y++) {
        raw[(row + 1) * y] = 0;
        std::memcpy(&raw[(row + 1) * y + 1], rgb + pitch * static_cast<size_t>(y), row);
    }
    uLongf zn = compressBound(static_cast<uLong>(raw.size()));
    std::vector<uint8_t> z(zn);
    if (compress2(z.data(), &zn, raw.data(), static_cast<uLong>(raw.size()), 1) != Z_OK) throw std::runtime_error("WritePngRgb: deflate failed");
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<uint8_t> ihdr;
    PutU32(ihdr, static_cast<uint32_t>(width));
    PutU32(ihdr, static_cast<uint32_t>(height));
    const uint8_t tail[5] = {8, 2, 0, 0, 0};   // bit depth 8, colour type 2 (RGB), deflate, adaptive filtering, no interlace
    ihdr.insert(ihdr.end(), tail, tail + 5);
    PutChunk(out, "IHDR", ihdr.data(), ihdr.size());
    PutChunk(out, "IDAT", z.data(), zn);
    PutChunk(out, "IEND", nullptr, 0);
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    if (std::fclose(f) != 0 || !ok) throw std::runtime_error("cannot write " + path);
}

void DrawKeypointMarkers(uint8_t* rgb, int width, int height, const float* kps, int n) {
    CvRng rng;
    auto put = [&](int x, int y, const uint8_t c[3]) {
        if (x < 0 || x >= width || y < 0 || y >= height) return;
        uint8_t* p = rgb + (static_cast<size_t>(y) * width + x) * 3;
        p[0] = c[0];
        p[1] = c[1];
        p[2] = c[2];
    };
    for (int i = 0; i < n; i++) {
        // cv::Scalar(rng(256), rng(256), rng(256)) on the BGR image: B, G, R in call order
        const uint8_t b = static_cast<uint8_t>(rng(256)), g = static_cast<uint8_t>(rng(256)), r = static_cast<uint8_t>(rng(256));
        const uint8_t c[3] = {r, g, b};
        const int x = static_cast<int>(kps[2 * i] + (kps[2 * i] >= 0 ? 0.5f : -0.5f)), y = static_cast<int>(kps[2 * i + 1] + (kps[2 * i + 1] >= 0 ? 0.5f : -0.5f));
        for (int d = -5; d <= 5; d++) put(x + d, y, c);   // markerSize 10: five pixels to both sides
        for (int d = -5; d <= 5; d++) put(x, y + d, c);
    }
}

void SaveImageForDebugging(uint8_t* rgb, int width, int height, int32_t frame_id, const std::string& dir,
                           const float* kps, int n) {
    char name[64];
    std::snprintf(name, sizeof name, "%06d.png", frame_id);
    WritePngRgb(dir + "/" + name, rgb, width, height, static_cast<size_t>(width) * 3);
    DrawKeypointMarkers(rgb, width, height, kps, n);
    std::snprintf(name, sizeof name, "keypoints_%06d.png", frame_id);
    WritePngRgb(dir + "/" + name, rgb, width, height, static_cast<size_t>(width) * 3);
}
