// debug_images.h -- the reference's debug dump of analysed frames (cpp/opticalflow.cc:80-96 SaveImageForDebugging):
// <dir>/%06d.png = the RGB frame, <dir>/keypoints_%06d.png = the frame with a cross marker on every keypoint.
#pragma once

#include <cstdint>
#include <string>

// 8-bit RGB, `pitch` bytes per row -> PNG (colour type 2, no interlace; zlib level 1).  Throws std::runtime_error.
void WritePngRgb(const std::string& path, const uint8_t* rgb, int width, int height, size_t pitch);

// cv::drawMarker(img, keypoint, colour, cv::MARKER_CROSS, 10) for every keypoint, on a tightly packed RGB image, with the
// colours of the reference: a copy of cv::theRNG() (state 0xffffffff) draws (B, G, R) = rng(256) x 3 per keypoint, i.e.
// the same colour sequence in every frame.  Keypoints are (x, y) floats with integer values (gftt.cc:157).
void DrawKeypointMarkers(uint8_t* rgb, int width, int height, const float* keypoints_xy, int n_keypoints);

// both files of one frame (opticalflow.cc:83-95); `rgb` is modified (markers)
void SaveImageForDebugging(uint8_t* rgb, int width, int height, int32_t frame_id, const std::string& dir,
                           const float* keypoints_xy, int n_keypoints);
