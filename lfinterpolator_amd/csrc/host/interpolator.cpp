// interpolator.cpp — see interpolator.h.  Follows the control flow of reference src/interpolator.cu:36-50, 95-137,
// 248-316: load the grid, upload it, compute offsets / weights / focus-map ids / constants on the host, optionally
// estimate the focus map, run the benchmark loop, store NN.png (+ mapK.png).
#include "interpolator.h"

#include <filesystem>
#include <iostream>
#include <stdexcept>

#include "image_io.h"
#include "lfLoader.h"
#include "loadingbar.hpp"

int Interpolator::defaultDevice = 0;

Interpolator::Interpolator(std::string inputPath) : input{inputPath}
{
    init();
}

Interpolator::Interpolator(lfi::IVec2 inColsRows, lfi::IVec2 inResolution, uint32_t seed, int inDevice)
    : device{inDevice}, colsRows{inColsRows}, resolution{inResolution.x, inResolution.y, 4}
{
    check(lfi_create(device, &context));
    check(lfi_set_grid(context, colsRows.x, colsRows.y, resolution.x, resolution.y));
    check(lfi_fill_synthetic(context, seed));
    check(lfi_sync(context));
}

Interpolator::~Interpolator()
{
    if(context)
        lfi_destroy(context);
}

void Interpolator::check(int status) const
{
    if(status != LFI_OK)
        throw std::runtime_error(std::string("GPU error: ") + lfi_last_error(context));
}

void Interpolator::init()
{
    check(lfi_create(device, &context));
    loadGPUData();
}

void Interpolator::loadGPUData()
{
    LfLoader lfLoader;
    lfLoader.loadData(input);
    colsRows = lfLoader.getColsRows();
    resolution = lfLoader.imageResolution();

    std::cout << "Uploading data to GPU..." << std::endl;
    LoadingBar bar(lfLoader.imageCount());
    check(lfi_set_grid(context, colsRows.x, colsRows.y, resolution.x, resolution.y));
    for(int col = 0; col < colsRows.x; col++)
        for(int row = 0; row < colsRows.y; row++)
        {
            // image id = col*rows + row, the order the reference creates its surfaces in (src/interpolator.cu:106-113)
            check(lfi_upload_image(context, col * colsRows.y + row, lfLoader.image({col, row}).data(), static_cast<size_t>(resolution.x) * channels));
            bar.add();
        }
}

void Interpolator::interpolate(std::string outputPath, std::string trajectory, float inFocus, float inRange, std::string method, float effect, float aspect)
{
    focus = inFocus;
    range = inRange;
    int methodID;
    if(method == "TEN_WM")
        methodID = LFI_METHOD_TEN_WM;
    else if(method == "STD")
        methodID = LFI_METHOD_STD;
    else
        throw std::runtime_error("The specified interpolation method does not exist!");

    lfi::Parameterizer parameterizer(colsRows, resolution);
    lfi::HostParams params = parameterizer.build(trajectory, focus, range, effect, aspect, viewCount);
    if(referenceMapQuirk)
        params.flags |= LFI_FLAG_REFERENCE_MAP_QUIRK;
    const lfi_params abi = params.abi();
    check(lfi_set_params(context, &abi));

    const int allFocus = inRange > 0;
    if(allFocus)
    {
        std::cout << "Estimating focus map..." << std::endl;
        check(lfi_focus_map(context));
    }

    std::cout << "Rendering views..." << std::endl;
    std::cout << "Elapsed time: " << std::endl;
    lfi_bench_stats stats{};
    // the reference's mean includes its cold first launch; one warm-up launch is excluded here
    check(lfi_benchmark(context, methodID, allFocus, 0, viewCount, 1, static_cast<int>(kernelBenchmarkRuns), &stats));
    averageTime = stats.mean_ms;
    std::cout << "Average time of " << std::to_string(kernelBenchmarkRuns) << " runs: " << stats.mean_ms << " ms" << std::endl;
    const double seconds = stats.median_ms / 1000.0;
    std::cout << "Median " << stats.median_ms << " ms, min " << stats.min_ms << " ms: " << viewCount / seconds << " views/s, "
              << static_cast<double>(viewCount) * resolution.x * resolution.y / seconds / 1e9 << " Gpix/s" << std::endl;
    storeResults(outputPath);
}

void Interpolator::storeResults(std::string path)
{
    std::cout << "Storing results..." << std::endl;
    constexpr int MAP_COUNT{2};
    int count = viewCount;
    if(range > 0)
        count += MAP_COUNT;
    std::filesystem::create_directories(path);
    LoadingBar bar(count);
    const size_t pitch = static_cast<size_t>(resolution.x) * channels;
    std::vector<uint8_t> data(pitch * resolution.y, 255);
    for(int i = 0; i < count; i++)
    {
        auto fileName = std::filesystem::path(path) / (std::string(((i < 10) ? "0" : "")) + std::to_string(i) + ".png");
        if(i >= viewCount)
        {
            fileName = std::filesystem::path(path) / ("map" + std::to_string(i - viewCount) + ".png");
            check(lfi_download_map(context, i - viewCount, data.data(), pitch));
        }
        else
            check(lfi_download_view(context, i, data.data(), pitch));
        lfi::writePng(fileName.string(), resolution.x, resolution.y, static_cast<int>(channels), data.data(), pitch);
        bar.add();
    }
    if(quiltTiles.x > 0 && quiltTiles.y > 0)
    {
        if(quiltTiles.x * quiltTiles.y > viewCount)
            throw std::runtime_error("The quilt has more tiles than rendered views!");
        std::cout << "Storing quilt..." << std::endl;
        const size_t quiltPitch = pitch * quiltTiles.x;
        std::vector<uint8_t> quilt(quiltPitch * resolution.y * quiltTiles.y);
        check(lfi_download_quilt(context, quiltTiles.x, quiltTiles.y, 0, quilt.data(), quiltPitch));
        lfi::writePng((std::filesystem::path(path) / "quilt.png").string(), resolution.x * quiltTiles.x, resolution.y * quiltTiles.y,
                      static_cast<int>(channels), quilt.data(), quiltPitch);
    }
}
