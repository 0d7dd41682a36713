// interpolator.h — host orchestration of the light-field interpolation.  The public surface is source-compatible with
// the reference's Interpolator (reference src/interpolator.h:5-37): same constructor, destructor and interpolate()
// signature, same method strings ("STD", "TEN_WM"), same exceptions.  All device work goes through the C-ABI of
// include/lfi.h; this file includes no HIP header.
#pragma once

#include <string>
#include <vector>

#include "../../../include/lfi.h"
#include "params.h"
#include "vec.h"

class Interpolator
{
    public:
        Interpolator(std::string inputPath);
        ~Interpolator();
        void interpolate(std::string outputPath, std::string trajectory, float focus, float range, std::string method, float effect, float aspect);

        // additions (defaults keep the reference's behaviour)
        void setViewCount(int count) { viewCount = count; }                // reference: always 64 (src/kernels.cu:11-13)
        void setBenchmarkRuns(size_t runs) { kernelBenchmarkRuns = runs; } // reference: 100 (src/interpolator.h:13)
        static void setDefaultDevice(int index) { defaultDevice = index; } // GPU used by Interpolator(path)
        // all-focus TEN_WM reads the filtered map 1 like STD instead of the reference's map 0 (src/kernels.cu:430 vs :326)
        void setUnifiedFocusMap(bool on) { unifiedFocusMap = on; }
        // also write quilt.png: the views as one image of cols × rows tiles (what scripts/viewsToQuilt.sh montages, 5×9 there)
        void setQuilt(lfi::IVec2 tiles) { quiltTiles = tiles; }
        float lastAverageTime() const { return averageTime; }
        // render on GPUs 0 … count-1 of this node: views are split into contiguous ranges, the grid is broadcast once (RCCL)
        void setGpuCount(int count) { gpuCount = count; }

        // synthetic cols×rows grid of width×height images instead of a directory
        Interpolator(lfi::IVec2 colsRows, lfi::IVec2 resolution, uint32_t seed, int device = 0);

    private:
        size_t kernelBenchmarkRuns{100};
        int viewCount{LFI_REFERENCE_VIEWS};
        static int defaultDevice;
        int device{defaultDevice};
        bool unifiedFocusMap{false};
        lfi::IVec2 quiltTiles{0, 0};
        lfi_ctx *context{nullptr};
        int gpuCount{1};
        std::vector<lfi_ctx *> contexts; // one per GPU; contexts[0] == context
        std::vector<int> viewStart;      // first view of each GPU's range (size gpuCount + 1)
        float focus{0};
        float range{0};
        float averageTime{0};
        size_t channels{4};
        lfi::IVec2 colsRows;
        lfi::IVec3 resolution;
        std::string input;
        void init();
        void loadGPUData();
        void storeResults(std::string path);
        void check(int status) const;
        void check(int status, lfi_ctx *where) const;
        void shardOverGpus(const lfi::HostParams &params);
};
