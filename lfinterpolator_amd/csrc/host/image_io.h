// image_io.h — RGBA8 image decode/encode used on the two steps either side of the hot path (see image_io.cpp).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace lfi {

struct Image
{
    int width{0}, height{0};
    std::vector<uint8_t> pixels; // RGBA8, tightly packed
};

// throws std::runtime_error("Cannot load image <path> …") like the reference (src/lfLoader.cpp:37-38)
Image loadImage(const std::string &path);
void writePng(const std::string &path, int width, int height, int channels, const uint8_t *data, size_t strideBytes);
void writePpm(const std::string &path, int width, int height, const uint8_t *rgba, size_t strideBytes);

} // namespace lfi
