// params.h — host parameterisation of the shift-and-sum kernels: everything the reference's Interpolator computes on
// the CPU before a launch (reference src/interpolator.cu:139-246, 318-337), kept above the C-ABI so that every consumer
// of include/lfi.h is handed identical bytes.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/lfi.h"
#include "vec.h"

namespace lfi {

// IEEE binary16 bit pattern of a float, round-to-nearest-even: static_cast<half>(float) on the host
// (reference src/interpolator.cu:219)
uint16_t floatToHalfBits(float value);
float halfBitsToFloat(uint16_t bits);

// The reference's per-launch parameter block (its __constant__ symbols + the weight matrix), owned by value.
struct HostParams
{
    int views{LFI_REFERENCE_VIEWS};
    std::vector<lfi_int2> focusedOffsets;   // [N]
    std::vector<lfi_float2> offsets;        // [N]
    std::vector<uint16_t> weights;          // [views][N] fp16 bits
    std::vector<int32_t> focusMapIDs;       // ≤ 32
    float focus{0}, range{0};
    int blockRadius[2]{1, 1};
    uint32_t flags{0};

    // view of this block for lfi_set_params; valid while *this is alive and unmodified
    lfi_params abi() const;
};

class Parameterizer
{
    public:
        Parameterizer(IVec2 colsRows, IVec3 resolution) : colsRows{colsRows}, resolution{resolution} {}

        // "startCol,startRow,endCol,endRow" in normalised grid coordinates → absolute grid coordinates
        Vec4 interpretTrajectory(const std::string &trajectory) const;
        std::vector<Vec2> generateTrajectory(Vec4 startEndPoints, int views) const;
        std::vector<float> generateWeights(Vec2 coords, float effect) const;
        std::vector<uint16_t> weightMatrix(Vec4 startEndPoints, float effect, int views) const;
        void offsets(float aspect, float focus, Vec4 startEndPoints, std::vector<lfi_float2> &offsets,
                     std::vector<lfi_int2> &focusedOffsets) const;
        std::vector<int32_t> selectFocusMapViews(Vec4 startEndPoints) const;
        IVec2 blockRadius() const;

        // everything at once: what Interpolator::interpolate prepares before its launches
        HostParams build(const std::string &trajectory, float focus, float range, float effect, float aspect,
                         int views = LFI_REFERENCE_VIEWS) const;

    private:
        IVec2 colsRows;
        IVec3 resolution;
};

Vec2 trajectoryCenter(Vec4 startEndPoints);

} // namespace lfi
