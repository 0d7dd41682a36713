// host_capi.cpp — C entry points over the host parameterisation, so that the Python plumbing (bench.py, tests) computes
// the kernel parameters with the very code the C++ Interpolator uses (params.cpp).
#include <cstring>
#include <exception>

#include "params.h"

extern "C" {

// Fills the caller's arrays: focused[N], offsets[N], weights[views*N], ids[32]; returns 0 or -1 (message in err).
int lfi_host_build_params(int cols, int rows, int width, int height, const char *trajectory, float focus, float range,
                          float effect, float aspect, int views, lfi_int2 *focused, lfi_float2 *offsets, uint16_t *weights,
                          int32_t *ids, int32_t *n_ids, int32_t block_radius[2], char *err, size_t err_len)
{
    try
    {
        lfi::Parameterizer p({cols, rows}, {width, height, 4});
        lfi::HostParams hp = p.build(trajectory, focus, range, effect, aspect, views);
        std::memcpy(focused, hp.focusedOffsets.data(), sizeof(lfi_int2) * hp.focusedOffsets.size());
        std::memcpy(offsets, hp.offsets.data(), sizeof(lfi_float2) * hp.offsets.size());
        std::memcpy(weights, hp.weights.data(), sizeof(uint16_t) * hp.weights.size());
        std::memcpy(ids, hp.focusMapIDs.data(), sizeof(int32_t) * hp.focusMapIDs.size());
        *n_ids = static_cast<int32_t>(hp.focusMapIDs.size());
        block_radius[0] = hp.blockRadius[0];
        block_radius[1] = hp.blockRadius[1];
        return 0;
    }
    catch(const std::exception &e)
    {
        if(err && err_len)
        {
            std::strncpy(err, e.what(), err_len - 1);
            err[err_len - 1] = 0;
        }
        return -1;
    }
}

uint16_t lfi_host_float_to_half(float v)
{
    return lfi::floatToHalfBits(v);
}

float lfi_host_half_to_float(uint16_t h)
{
    return lfi::halfBitsToFloat(h);
}

} // extern "C"

// ---- image I/O and the loader, for the tests of the steps either side of the hot path ----------------------------------
#include "image_io.h"
#include "lfLoader.h"

extern "C" {

// decode a PNG / PPM into caller memory (RGBA8); call with rgba == NULL to query the size
int lfi_host_load_image(const char *path, int *width, int *height, uint8_t *rgba, char *err, size_t err_len)
{
    try
    {
        lfi::Image img = lfi::loadImage(path);
        *width = img.width;
        *height = img.height;
        if(rgba)
            std::memcpy(rgba, img.pixels.data(), img.pixels.size());
        return 0;
    }
    catch(const std::exception &e)
    {
        if(err && err_len)
        {
            std::strncpy(err, e.what(), err_len - 1);
            err[err_len - 1] = 0;
        }
        return -1;
    }
}

int lfi_host_write_png(const char *path, int width, int height, int channels, const uint8_t *data, char *err, size_t err_len)
{
    try
    {
        lfi::writePng(path, width, height, channels, data, static_cast<size_t>(width) * channels);
        return 0;
    }
    catch(const std::exception &e)
    {
        if(err && err_len)
        {
            std::strncpy(err, e.what(), err_len - 1);
            err[err_len - 1] = 0;
        }
        return -1;
    }
}

// LfLoader::loadData on a directory: returns cols/rows/resolution and (if planes != NULL) the images in g = col*rows+row order
int lfi_host_load_grid(const char *path, int *cols, int *rows, int *width, int *height, uint8_t *planes, char *err, size_t err_len)
{
    try
    {
        LfLoader loader;
        loader.loadData(path);
        const lfi::IVec2 cr = loader.getColsRows();
        const lfi::IVec3 res = loader.imageResolution();
        *cols = cr.x;
        *rows = cr.y;
        *width = res.x;
        *height = res.y;
        if(planes)
            for(int col = 0; col < cr.x; col++)
                for(int row = 0; row < cr.y; row++)
                    std::memcpy(planes + loader.imageSize() * (static_cast<size_t>(col) * cr.y + row), loader.image({col, row}).data(), loader.imageSize());
        return 0;
    }
    catch(const std::exception &e)
    {
        if(err && err_len)
        {
            std::strncpy(err, e.what(), err_len - 1);
            err[err_len - 1] = 0;
        }
        return -1;
    }
}

} // extern "C"
