// interpolator.cpp — see interpolator.h.  Follows the control flow of reference src/interpolator.cu:36-50, 95-137,
// 248-316: load the grid, upload it, compute offsets / weights / focus-map ids / constants on the host, optionally
// estimate the focus map, run the benchmark loop, store NN.png (+ mapK.png).
#include "interpolator.h"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <filesystem>
#include <future>
#include <iostream>
#include <mutex>
#include <stdexcept>
#include <thread>

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
    for(size_t i = 1; i < contexts.size(); i++)
        lfi_destroy(contexts[i]);
    if(context)
        lfi_destroy(context);
}

void Interpolator::check(int status) const
{
    check(status, context);
}

void Interpolator::check(int status, lfi_ctx *where) const
{
    if(status != LFI_OK)
        throw std::runtime_error(std::string("GPU error: ") + lfi_last_error(where));
}

// One context per GPU: the grid is broadcast once from the GPU it was loaded on, then every GPU gets its contiguous range of
// views (its own rows of the weight matrix; offsets are the same everywhere) — no further communication.
void Interpolator::shardOverGpus(const lfi::HostParams &params)
{
    if(contexts.empty())
    {
        contexts.push_back(context);
        for(int g = 1; g < gpuCount; g++)
        {
            lfi_ctx *extra = nullptr;
            check(lfi_create(device + g, &extra), nullptr);
            contexts.push_back(extra);
            check(lfi_set_grid(extra, colsRows.x, colsRows.y, resolution.x, resolution.y), extra);
        }
        if(gpuCount > 1)
        {
            std::cout << "Broadcasting the light field to " << gpuCount << " GPUs..." << std::endl;
            check(lfi_broadcast_grid(contexts.data(), gpuCount, 0));
        }
    }
    viewStart.assign(gpuCount + 1, 0);
    for(int g = 0; g < gpuCount; g++)
        viewStart[g + 1] = viewStart[g] + viewCount / gpuCount + (g < viewCount % gpuCount ? 1 : 0);
    const size_t n = params.offsets.size();
    for(int g = 0; g < gpuCount; g++)
    {
        lfi::HostParams part = params;
        part.views = viewStart[g + 1] - viewStart[g];
        part.weights.assign(params.weights.begin() + viewStart[g] * n, params.weights.begin() + viewStart[g + 1] * n);
        const lfi_params abi = part.abi();
        check(lfi_set_params(contexts[g], &abi), contexts[g]);
    }
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
            // asynchronous: staged through page-locked slots on the library's copy stream, so the host walks on to the next image
            // while this one crosses PCIe (the reference copies synchronously: src/interpolator.cu:85-93, 106-113)
            check(lfi_upload_image_async(context, col * colsRows.y + row, lfLoader.image({col, row}).data(), static_cast<size_t>(resolution.x) * channels));
            bar.add();
        }
    check(lfi_upload_wait(context));
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
    if(unifiedFocusMap)
        params.flags |= LFI_FLAG_UNIFIED_FOCUS_MAP;
    if(gpuCount < 1 || gpuCount > viewCount)
        throw std::runtime_error("The number of GPUs has to be between 1 and the number of views!");
    shardOverGpus(params);
    // The views never leave the library except through lfi_download_view / _quilt, which re-create the constant alpha: fixed-focus
    // TEN_WM renders therefore use the alpha-free byte-plane layout (a quarter fewer bytes written per launch, csrc/hip/blend_p3.hpp);
    // the other renders keep the reference's RGBA planes (they would pay a conversion pass).  Output files are identical.
    for(lfi_ctx *c : contexts)
        check(lfi_set_output_layout(c, (methodID == LFI_METHOD_TEN_WM && !(inRange > 0)) ? LFI_LAYOUT_PLANAR_RGB : LFI_LAYOUT_RGBA), c);

    const int allFocus = inRange > 0;
    if(allFocus)
    {
        std::cout << "Estimating focus map..." << std::endl;
        for(lfi_ctx *c : contexts)
            check(lfi_focus_map(c), c);
    }

    std::cout << "Rendering views..." << std::endl;
    std::cout << "Elapsed time: " << std::endl;
    double medianMs;
    if(gpuCount == 1)
    {
        lfi_bench_stats stats{};
        // the reference's mean includes its cold first launch; one warm-up launch is excluded here
        check(lfi_benchmark(context, methodID, allFocus, 0, viewCount, 1, static_cast<int>(kernelBenchmarkRuns), &stats));
        averageTime = stats.mean_ms;
        medianMs = stats.median_ms;
        std::cout << "Average time of " << std::to_string(kernelBenchmarkRuns) << " runs: " << stats.mean_ms << " ms" << std::endl;
        std::cout << "Median " << stats.median_ms << " ms, min " << stats.min_ms << " ms";
    }
    else
    {
        // all GPUs launch their view ranges concurrently; a run ends when the slowest GPU has finished
        const auto launchAll = [&] {
            for(int g = 0; g < gpuCount; g++)
                check(lfi_render(contexts[g], methodID, allFocus, 0, viewStart[g + 1] - viewStart[g]), contexts[g]);
            for(lfi_ctx *c : contexts)
                check(lfi_sync(c), c);
        };
        launchAll();
        std::vector<double> times;
        for(size_t i = 0; i < kernelBenchmarkRuns; i++)
        {
            const auto t0 = std::chrono::steady_clock::now();
            launchAll();
            times.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
        double sum = 0;
        for(double t : times)
            sum += t;
        averageTime = static_cast<float>(sum / times.size());
        std::sort(times.begin(), times.end());
        medianMs = times[times.size() / 2];
        std::cout << "Average time of " << std::to_string(kernelBenchmarkRuns) << " runs on " << gpuCount << " GPUs: " << averageTime << " ms" << std::endl;
        std::cout << "Median " << medianMs << " ms (host clock around all GPUs)";
    }
    const double seconds = medianMs / 1000.0;
    std::cout << ": " << viewCount / seconds << " views/s, "
              << static_cast<double>(viewCount) * resolution.x * resolution.y / seconds / 1e9 << " Gpix/s" << std::endl;
    storeResults(outputPath);
}

void Interpolator::storeResults(std::string path)
{
    // Same files as the reference (NN.png, mapK.png: src/interpolator.cu:299-316).  The device→host copies land in a ring of
    // page-locked buffers and the PNG encoding runs on worker threads, so the GPU copy of view i+1 overlaps the compression
    // of view i (the reference downloads and encodes one view at a time on one thread).
    std::cout << "Storing results..." << std::endl;
    constexpr int MAP_COUNT{2};
    int count = viewCount;
    if(range > 0)
        count += MAP_COUNT;
    std::filesystem::create_directories(path);
    LoadingBar bar(count);
    const size_t pitch = static_cast<size_t>(resolution.x) * channels;
    const size_t imageBytes = pitch * resolution.y;
    const int workers = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const int slots = workers + 1;
    uint8_t *ring = nullptr;
    const bool pinned = lfi_alloc_pinned(imageBytes * slots, reinterpret_cast<void **>(&ring)) == LFI_OK;
    std::vector<uint8_t> pageable;
    if(!pinned)
    {
        pageable.resize(imageBytes * slots);
        ring = pageable.data();
    }
    std::mutex mutex;
    std::condition_variable slotFree;
    std::vector<bool> busy(slots, false);
    std::vector<std::future<void>> jobs;
    std::string firstError;
    for(int i = 0; i < count; i++)
    {
        int slot;
        {
            std::unique_lock<std::mutex> lock(mutex);
            slotFree.wait(lock, [&] { return std::find(busy.begin(), busy.end(), false) != busy.end(); });
            slot = static_cast<int>(std::find(busy.begin(), busy.end(), false) - busy.begin());
            busy[slot] = true;
        }
        uint8_t *data = ring + imageBytes * slot;
        auto fileName = std::filesystem::path(path) / (std::string(((i < 10) ? "0" : "")) + std::to_string(i) + ".png");
        if(i >= viewCount)
        {
            fileName = std::filesystem::path(path) / ("map" + std::to_string(i - viewCount) + ".png");
            check(lfi_download_map(context, i - viewCount, data, pitch));
        }
        else
        {
            int g = 0;
            while(g + 1 < gpuCount && i >= viewStart[g + 1])
                g++;
            check(lfi_download_view(contexts[g], i - viewStart[g], data, pitch), contexts[g]);
        }
        jobs.push_back(std::async(std::launch::async, [&, slot, data, fileName] {
            try
            {
                lfi::writePng(fileName.string(), resolution.x, resolution.y, static_cast<int>(channels), data, pitch);
            }
            catch(const std::exception &e)
            {
                std::lock_guard<std::mutex> lock(mutex);
                if(firstError.empty())
                    firstError = e.what();
            }
            {
                std::lock_guard<std::mutex> lock(mutex);
                busy[slot] = false;
            }
            slotFree.notify_one();
        }));
        bar.add();
    }
    for(auto &job : jobs)
        job.get();
    if(pinned)
        lfi_free_pinned(ring);
    if(!firstError.empty())
        throw std::runtime_error(firstError);
    if(quiltTiles.x > 0 && quiltTiles.y > 0)
    {
        if(quiltTiles.x * quiltTiles.y > viewCount)
            throw std::runtime_error("The quilt has more tiles than rendered views!");
        std::cout << "Storing quilt..." << std::endl;
        const size_t quiltPitch = pitch * quiltTiles.x;
        std::vector<uint8_t> quilt(quiltPitch * resolution.y * quiltTiles.y);
        // every GPU assembles the tiles of ITS views on the device and copies them into their place in the one host image
        const int tiles = quiltTiles.x * quiltTiles.y;
        for(int g = 0; g < gpuCount; g++)
        {
            const int first = viewStart[g], last = std::min(g + 1 < gpuCount ? viewStart[g + 1] : viewCount, tiles);
            if(first < last)
                check(lfi_download_quilt_tiles(contexts[g], quiltTiles.x, quiltTiles.y, first, last - first, 0, quilt.data(), quiltPitch), contexts[g]);
        }
        lfi::writePng((std::filesystem::path(path) / "quilt.png").string(), resolution.x * quiltTiles.x, resolution.y * quiltTiles.y,
                      static_cast<int>(channels), quilt.data(), quiltPitch);
    }
}
