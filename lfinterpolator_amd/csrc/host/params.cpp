// params.cpp — see params.h.  Float arithmetic follows the operation order of the reference's host code
// (reference src/interpolator.cu:139-246, 318-337); compile with -ffp-contract=off so nothing is fused.
#include "params.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <sstream>
#include <stdexcept>

namespace lfi {

uint16_t floatToHalfBits(float value)
{
    uint32_t u;
    std::memcpy(&u, &value, sizeof(u));
    const uint16_t sign = static_cast<uint16_t>((u >> 16) & 0x8000u);
    const uint32_t mag = u & 0x7fffffffu;
    if(mag > 0x7f800000u) // NaN
        return static_cast<uint16_t>(sign | 0x7e00u);
    if(mag >= 0x477ff000u) // ≥ 65520 rounds to infinity
        return static_cast<uint16_t>(sign | 0x7c00u);
    if(mag < 0x38800000u) // below the smallest normal half (2^-14): subnormal result, unit 2^-24
    {
        if(mag <= 0x33000000u) // ≤ 2^-25: rounds to zero (the tie at exactly 2^-25 goes to the even 0)
            return sign;
        const uint32_t exponent = mag >> 23;
        const uint32_t mantissa = (mag & 0x7fffffu) | 0x800000u;
        const uint32_t shift = 126u - exponent; // value = mantissa · 2^(exponent-150) = (mantissa >> shift) · 2^-24
        uint32_t q = mantissa >> shift;
        const uint32_t rem = mantissa & ((1u << shift) - 1u);
        const uint32_t half = 1u << (shift - 1u);
        if(rem > half || (rem == half && (q & 1u)))
            q++;
        return static_cast<uint16_t>(sign | q);
    }
    uint32_t q = ((mag >> 23) - 112u) << 10 | ((mag & 0x7fffffu) >> 13);
    const uint32_t rem = mag & 0x1fffu;
    if(rem > 0x1000u || (rem == 0x1000u && (q & 1u)))
        q++; // a carry out of the mantissa correctly bumps the exponent
    return static_cast<uint16_t>(sign | q);
}

float halfBitsToFloat(uint16_t bits)
{
    const uint32_t sign = static_cast<uint32_t>(bits & 0x8000u) << 16;
    const uint32_t exponent = (bits >> 10) & 0x1fu;
    uint32_t mantissa = bits & 0x3ffu;
    uint32_t u;
    if(exponent == 0)
    {
        if(mantissa == 0)
            u = sign;
        else
        {
            int e = -1;
            do
            {
                mantissa <<= 1;
                e++;
            } while(!(mantissa & 0x400u));
            u = sign | static_cast<uint32_t>(112 - e) << 23 | (mantissa & 0x3ffu) << 13;
        }
    }
    else if(exponent == 31)
        u = sign | 0x7f800000u | mantissa << 13;
    else
        u = sign | (exponent + 112u) << 23 | mantissa << 13;
    float f;
    std::memcpy(&f, &u, sizeof(f));
    return f;
}

lfi_params HostParams::abi() const
{
    lfi_params p{};
    p.views = views;
    p.focused_offsets = focusedOffsets.data();
    p.offsets = offsets.data();
    p.weights_fp16 = weights.data();
    p.focus_map_ids = focusMapIDs.empty() ? nullptr : focusMapIDs.data();
    p.n_focus_ids = static_cast<int32_t>(focusMapIDs.size());
    p.focus = focus;
    p.range = range;
    p.block_radius[0] = blockRadius[0];
    p.block_radius[1] = blockRadius[1];
    p.flags = flags;
    return p;
}

Vec2 trajectoryCenter(Vec4 startEndPoints)
{
    return startEndPoints.xy() + (startEndPoints.zw() - startEndPoints.xy()) * 0.5f;
}

// reference src/interpolator.cu:318-337; unlike the reference, a malformed string is an error instead of
// uninitialised coordinates
Vec4 Parameterizer::interpretTrajectory(const std::string &trajectory) const
{
    std::stringstream stream(trajectory);
    std::string token;
    Vec4 absolute;
    int i = 0;
    const int dims[2] = {colsRows.x, colsRows.y};
    while(std::getline(stream, token, ','))
    {
        if(i >= 4)
            throw std::runtime_error("The trajectory " + trajectory + " has more than four coordinates!");
        float value;
        try
        {
            value = std::stof(token);
        }
        catch(const std::exception &)
        {
            throw std::runtime_error("The trajectory " + trajectory + " is not in the format startCol,startRow,endCol,endRow!");
        }
        absolute[i] = value * static_cast<float>(dims[i % 2] - 1);
        i++;
    }
    if(i != 4)
        throw std::runtime_error("The trajectory " + trajectory + " is not in the format startCol,startRow,endCol,endRow!");
    return absolute;
}

// reference src/interpolator.cu:174-182 with the view count as a parameter (64 there)
std::vector<Vec2> Parameterizer::generateTrajectory(Vec4 startEndPoints, int views) const
{
    std::vector<Vec2> trajectory;
    trajectory.reserve(views);
    if(views <= 1)
    {
        trajectory.push_back(startEndPoints.xy());
        return trajectory;
    }
    const Vec2 step = (startEndPoints.zw() - startEndPoints.xy()) / static_cast<float>(views - 1);
    for(int i = 0; i < views; i++)
        trajectory.push_back(startEndPoints.xy() + step * static_cast<float>(i));
    return trajectory;
}

// reference src/interpolator.cu:156-172
std::vector<float> Parameterizer::generateWeights(Vec2 coords, float effect) const
{
    const float maxDistance = distance(Vec2{0, 0}, toVec2(colsRows));
    float weightSum = 0;
    std::vector<float> weightVals;
    weightVals.reserve(static_cast<size_t>(colsRows.x) * colsRows.y);
    for(int col = 0; col < colsRows.x; col++)
        for(int row = 0; row < colsRows.y; row++)
        {
            float weight = maxDistance - distance(coords, Vec2{static_cast<float>(col), static_cast<float>(row)});
            weight = powf(weight, effect);
            weightSum += weight;
            weightVals.push_back(weight);
        }
    for(auto &weight : weightVals)
        weight /= weightSum;
    return weightVals;
}

// reference src/interpolator.cu:209-224 (the host half: the upload is lfi_set_params)
std::vector<uint16_t> Parameterizer::weightMatrix(Vec4 startEndPoints, float effect, int views) const
{
    std::vector<uint16_t> matrix;
    matrix.reserve(static_cast<size_t>(views) * colsRows.x * colsRows.y);
    for(const auto &view : generateTrajectory(startEndPoints, views))
        for(float w : generateWeights(view, effect))
            matrix.push_back(floatToHalfBits(w));
    return matrix;
}

// reference src/interpolator.cu:226-246
void Parameterizer::offsets(float aspect, float focus, Vec4 startEndPoints, std::vector<lfi_float2> &outOffsets,
                            std::vector<lfi_int2> &outFocused) const
{
    outOffsets.clear();
    outFocused.clear();
    const Vec2 center = trajectoryCenter(startEndPoints);
    const float offsetAspect = (static_cast<float>(resolution.x) / resolution.y) / aspect;
    const Vec2 res{static_cast<float>(resolution.x), static_cast<float>(resolution.y)};
    for(int col = 0; col < colsRows.x; col++)
        for(int row = 0; row < colsRows.y; row++)
        {
            const Vec2 position{static_cast<float>(col), static_cast<float>(row)};
            Vec2 offset = (center - position) / toVec2(colsRows);
            offset = offset * res;
            offset.y *= offsetAspect;
            outOffsets.push_back({offset.x, offset.y});
            const IVec2 rounded = roundToInt(offset * Vec2{focus, focus});
            outFocused.push_back({rounded.x, rounded.y});
        }
}

// reference src/interpolator.cu:194-207; at most 32 ids (the reference indexes 32 unconditionally: SURVEY.md D4) and ties
// ordered by id (std::sort leaves them unspecified there)
std::vector<int32_t> Parameterizer::selectFocusMapViews(Vec4 startEndPoints) const
{
    std::vector<std::pair<float, int32_t>> distances;
    const Vec2 center = trajectoryCenter(startEndPoints);
    for(int col = 0; col < colsRows.x; col++)
        for(int row = 0; row < colsRows.y; row++)
            distances.push_back({distance(Vec2{static_cast<float>(col), static_cast<float>(row)}, center),
                                 static_cast<int32_t>(distances.size())});
    std::stable_sort(distances.begin(), distances.end(),
                     [](const auto &a, const auto &b) { return a.first < b.first; });
    std::vector<int32_t> ids;
    const size_t count = std::min<size_t>(LFI_MAX_FOCUS_IDS, distances.size());
    for(size_t i = 0; i < count; i++)
        ids.push_back(distances[i].second);
    return ids;
}

// reference src/interpolator.cu:139-146; a zero radius (image narrower than 100 px) never advances the tap loops of the
// focus-map kernel, so it is raised to 1
IVec2 Parameterizer::blockRadius() const
{
    constexpr int PIXEL_SIZE_FACTOR{100};
    IVec2 radius{resolution.x / PIXEL_SIZE_FACTOR, resolution.y / PIXEL_SIZE_FACTOR};
    if((radius.x % 2) != 0)
        radius.x++;
    if((radius.y % 2) != 0)
        radius.y++;
    radius.x = std::max(radius.x, 1);
    radius.y = std::max(radius.y, 1);
    return radius;
}

HostParams Parameterizer::build(const std::string &trajectory, float focus, float range, float effect, float aspect,
                                int views) const
{
    if(views < 1)
        throw std::runtime_error("The number of views has to be positive!");
    HostParams p;
    p.views = views;
    p.focus = focus;
    p.range = range;
    const Vec4 points = interpretTrajectory(trajectory);
    offsets(aspect, focus, points, p.offsets, p.focusedOffsets);
    p.weights = weightMatrix(points, effect, views);
    p.focusMapIDs = selectFocusMapViews(points);
    const IVec2 radius = blockRadius();
    p.blockRadius[0] = radius.x;
    p.blockRadius[1] = radius.y;
    return p;
}

} // namespace lfi
