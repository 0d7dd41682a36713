// vec.h — the few vector types the host code needs (the reference uses glm, which is a system package there:
// reference CMakeLists.txt:22).  Component-wise float arithmetic in the same operation order as glm's templates, so
// the parameter bytes handed to the device are the ones the reference computes.
#pragma once

#include <cmath>

namespace lfi {

struct Vec2
{
    float x{0}, y{0};
};
struct IVec2
{
    int x{0}, y{0};
};
struct IVec3
{
    int x{0}, y{0}, z{0};
};
struct Vec4
{
    float x{0}, y{0}, z{0}, w{0};
    Vec2 xy() const { return {x, y}; }
    Vec2 zw() const { return {z, w}; }
    float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : (i == 2 ? z : w)); }
};

inline Vec2 operator+(Vec2 a, Vec2 b) { return {a.x + b.x, a.y + b.y}; }
inline Vec2 operator-(Vec2 a, Vec2 b) { return {a.x - b.x, a.y - b.y}; }
inline Vec2 operator*(Vec2 a, Vec2 b) { return {a.x * b.x, a.y * b.y}; }
inline Vec2 operator/(Vec2 a, Vec2 b) { return {a.x / b.x, a.y / b.y}; }
inline Vec2 operator*(Vec2 a, float s) { return {a.x * s, a.y * s}; }
inline Vec2 operator/(Vec2 a, float s) { return {a.x / s, a.y / s}; }
inline Vec2 toVec2(IVec2 v) { return {static_cast<float>(v.x), static_cast<float>(v.y)}; }

// glm::length / glm::distance: sqrt(dot(d, d)) with dot = d.x*d.x + d.y*d.y
inline float distance(Vec2 a, Vec2 b)
{
    Vec2 d = b - a;
    Vec2 sq = d * d;
    return std::sqrt(sq.x + sq.y);
}

// glm::round → std::round (half away from zero), then the int conversion of glm::ivec2(vec2)
inline IVec2 roundToInt(Vec2 v) { return {static_cast<int>(std::round(v.x)), static_cast<int>(std::round(v.y))}; }

} // namespace lfi
