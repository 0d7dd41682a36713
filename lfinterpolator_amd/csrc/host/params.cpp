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

// The point half way along the trajectory, start + (end − start)·0.5 per coordinate (what reference src/interpolator.cu:189-192 computes)
Vec2 trajectoryCenter(Vec4 startEndPoints)
{
    const float dCol = startEndPoints.z - startEndPoints.x, dRow = startEndPoints.w - startEndPoints.y;
    return {startEndPoints.x + dCol * 0.5f, startEndPoints.y + dRow * 0.5f};
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

namespace {

// Euclidean distance as glm::distance rounds it: the squares and their sum in float, then sqrtf
inline float gridDistance(float colA, float rowA, float colB, float rowB)
{
    const float dc = colB - colA, dr = rowB - rowA;
    return std::sqrt(dc * dc + dr * dr);
}

} // namespace

// Camera i of `views` equally spaced cameras from the trajectory's start to its end (a single view sits on the start point).  The step is
// (end − start) / (views − 1), the camera start + step·i — the reference's rounding (src/interpolator.cu:174-182; 64 views there).
std::vector<Vec2> Parameterizer::generateTrajectory(Vec4 startEndPoints, int views) const
{
    std::vector<Vec2> cameras(static_cast<size_t>(std::max(views, 1)), Vec2{startEndPoints.x, startEndPoints.y});
    if(views <= 1)
        return cameras;
    const float last = static_cast<float>(views - 1);
    const float stepCol = (startEndPoints.z - startEndPoints.x) / last, stepRow = (startEndPoints.w - startEndPoints.y) / last;
    for(int i = 0; i < views; i++)
    {
        const float fi = static_cast<float>(i);
        cameras[i] = {startEndPoints.x + stepCol * fi, startEndPoints.y + stepRow * fi};
    }
    return cameras;
}

// One view's blending weights, image g = col·rows + row: (dmax − distance(camera, image))^effect, normalised by their sum accumulated in
// image order — float powf, float running sum, one division per weight (src/interpolator.cu:156-172)
std::vector<float> Parameterizer::generateWeights(Vec2 coords, float effect) const
{
    const int cols = colsRows.x, rows = colsRows.y;
    const float dmax = gridDistance(0.0f, 0.0f, static_cast<float>(cols), static_cast<float>(rows));
    std::vector<float> w(static_cast<size_t>(cols) * rows);
    float total = 0.0f;
    for(size_t g = 0; g < w.size(); g++)
    {
        const float col = static_cast<float>(g / rows), row = static_cast<float>(g % rows);
        const float closeness = dmax - gridDistance(coords.x, coords.y, col, row);
        w[g] = powf(closeness, effect);
        total += w[g];
    }
    for(float &value : w)
        value /= total;
    return w;
}

// [views][N] fp16 bit patterns, rounded to nearest even like static_cast<half>(float) on the host (src/interpolator.cu:209-224; the upload
// is lfi_set_params)
std::vector<uint16_t> Parameterizer::weightMatrix(Vec4 startEndPoints, float effect, int views) const
{
    const size_t n = static_cast<size_t>(colsRows.x) * colsRows.y;
    const std::vector<Vec2> cameras = generateTrajectory(startEndPoints, views);
    std::vector<uint16_t> matrix(cameras.size() * n);
    for(size_t v = 0; v < cameras.size(); v++)
    {
        const std::vector<float> row = generateWeights(cameras[v], effect);
        std::transform(row.begin(), row.end(), matrix.begin() + v * n, floatToHalfBits);
    }
    return matrix;
}

// Per image: the shift of the image against the trajectory's centre at focus 1, in pixels — ((centre − position) / grid) · resolution,
// the row component also times (width / height) / aspect — and the same times `focus`, rounded half away from zero, for the fixed-focus
// kernels (src/interpolator.cu:226-246)
void Parameterizer::offsets(float aspect, float focus, Vec4 startEndPoints, std::vector<lfi_float2> &outOffsets,
                            std::vector<lfi_int2> &outFocused) const
{
    const int cols = colsRows.x, rows = colsRows.y;
    const size_t n = static_cast<size_t>(cols) * rows;
    outOffsets.assign(n, lfi_float2{0.0f, 0.0f});
    outFocused.assign(n, lfi_int2{0, 0});
    const Vec2 centre = trajectoryCenter(startEndPoints);
    const float width = static_cast<float>(resolution.x), height = static_cast<float>(resolution.y);
    const float rowScale = (static_cast<float>(resolution.x) / resolution.y) / aspect;
    for(size_t g = 0; g < n; g++)
    {
        const float col = static_cast<float>(g / rows), row = static_cast<float>(g % rows);
        const float shiftX = ((centre.x - col) / static_cast<float>(cols)) * width;
        const float shiftY = (((centre.y - row) / static_cast<float>(rows)) * height) * rowScale;
        outOffsets[g] = {shiftX, shiftY};
        outFocused[g] = {static_cast<int>(std::round(shiftX * focus)), static_cast<int>(std::round(shiftY * focus))};
    }
}

// reference src/interpolator.cu:194-207; at most 32 ids (the reference indexes 32 unconditionally: SURVEY.md D4) and ties
// ordered by id (std::sort leaves them unspecified there)
std::vector<int32_t> Parameterizer::selectFocusMapViews(Vec4 startEndPoints) const
{
    const int rows = colsRows.y;
    const size_t n = static_cast<size_t>(colsRows.x) * rows;
    const Vec2 centre = trajectoryCenter(startEndPoints);
    std::vector<float> dist(n);
    std::vector<int32_t> order(n);
    for(size_t g = 0; g < n; g++)
    {
        dist[g] = gridDistance(static_cast<float>(g / rows), static_cast<float>(g % rows), centre.x, centre.y);
        order[g] = static_cast<int32_t>(g);
    }
    // nearest first; images at the same distance in id order
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return dist[a] < dist[b]; });
    order.resize(std::min<size_t>(LFI_MAX_FOCUS_IDS, n));
    return order;
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
