// focus_map.hpp — focus-map estimation and smoothing for the all-focus render.
//
// Replaces Kernels::FocusMap::estimate / filter (reference src/kernels.cu:239-280) and their helpers
// focusDispersion (:196-217), ElementRange (:173-194), MinDispersion (:219-237), distance (:167-170).
// Per pixel: 32 focus candidates f_i = fma(step, i, focus); for each, the colour range (max−min per channel, then the
// largest channel) over the n_focus_ids nearest grid images at 3×3 taps around the warped position, summed over the
// taps; the first strict minimum wins and is stored as round((f*−focus)/range·255) in map 0.  filter() box-averages
// map 0 over [x−rx, x+rx) × [y−ry, y+ry), r = blockRadius/10, into map 1.
#pragma once

#include <float.h>

#include "lfi_device.hpp"

namespace lfi {

__global__ void __launch_bounds__(256) focus_estimate(const KernelArgs a)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int W = a.width, H = a.height;
    if(x >= W || y >= H)
        return;
    constexpr int STEPS = 32; // src/kernels.cu:245
    const float step = __fdiv_rn(a.range, static_cast<float>(STEPS - 1));
    const int rx = a.radius_x, ry = a.radius_y;
    float best_d = FLT_MAX, best_f = 0.0f;
    for(int i = 0; i < STEPS; i++)
    {
        const float f = __builtin_fmaf(step, static_cast<float>(i), a.focus);
        float lo[9][3], hi[9][3];
#pragma unroll
        for(int t = 0; t < 9; t++)
#pragma unroll
            for(int c = 0; c < 3; c++)
            {
                lo[t][c] = FLT_MAX;
                hi[t][c] = FLT_MIN; // sic (src/kernels.cu:178): the smallest positive normal, not -FLT_MAX
            }
        for(int k = 0; k < a.n_focus_ids; k++)
        {
            const int g = a.focus_ids[k];
            const lfi_float2 off = a.offsets[g];
            const int cx = warp_float(x, f, off.x);
            const int cy = warp_float(y, f, off.y);
#pragma unroll
            for(int tx = 0; tx < 3; tx++) // x outer, y inner: src/kernels.cu:208-210
#pragma unroll
                for(int ty = 0; ty < 3; ty++)
                {
                    const uint32_t px = fetch_px(a.grid, W, H, g, cx + (tx - 1) * rx, cy + (ty - 1) * ry);
                    const int t = tx * 3 + ty;
                    const float p0 = static_cast<float>(px & 0xffu), p1 = static_cast<float>((px >> 8) & 0xffu),
                                p2 = static_cast<float>((px >> 16) & 0xffu);
                    lo[t][0] = fminf(lo[t][0], p0);
                    hi[t][0] = fmaxf(hi[t][0], p0);
                    lo[t][1] = fminf(lo[t][1], p1);
                    hi[t][1] = fmaxf(hi[t][1], p1);
                    lo[t][2] = fminf(lo[t][2], p2);
                    hi[t][2] = fmaxf(hi[t][2], p2);
                }
        }
        float total = 0.0f;
#pragma unroll
        for(int t = 0; t < 9; t++)
            total += fmaxf(fmaxf(fabsf(lo[t][0] - hi[t][0]), fabsf(lo[t][1] - hi[t][1])), fabsf(lo[t][2] - hi[t][2]));
        if(total < best_d) // first strict minimum: src/kernels.cu:227
        {
            best_d = total;
            best_f = f;
        }
    }
    const float normalized = __fdiv_rn(best_f - a.focus, a.range);
    const uint32_t m = static_cast<uint32_t>(roundf(normalized * 255.0f)) & 0xffu;
    reinterpret_cast<uint32_t *>(a.maps)[(size_t)y * W + x] = m | (m << 8) | (m << 16) | 0xff000000u;
}

__global__ void __launch_bounds__(256) focus_filter(const KernelArgs a)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int W = a.width, H = a.height;
    if(x >= W || y >= H)
        return;
    const int rx = max(a.radius_x / 10, 1), ry = max(a.radius_y / 10, 1); // ≥1: SURVEY.md defect D6
    const uint32_t *map0 = reinterpret_cast<const uint32_t *>(a.maps);
    float avg = 0.0f;
    int count = 0;
    for(int tx = x - rx; tx < x + rx; tx++)
        for(int ty = y - ry; ty < y + ry; ty++)
        {
            avg += static_cast<float>(map0[clampi(ty, 0, H - 1) * W + clampi(tx, 0, W - 1)] & 0xffu);
            count++;
        }
    avg = __fdiv_rn(avg, static_cast<float>(count));
    const uint32_t m = static_cast<uint32_t>(roundf(avg)) & 0xffu;
    reinterpret_cast<uint32_t *>(a.maps)[(size_t)W * H + (size_t)y * W + x] = m | (m << 8) | (m << 16) | 0xff000000u;
}

// focusCoords dump for the integer-warp parity test (src/kernels.cu:72-82): unclamped coordinates of image g
__global__ void __launch_bounds__(256) dump_coords(const KernelArgs a, const int g, const int all_focus, lfi_int2 *__restrict__ out)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int W = a.width, H = a.height;
    if(x >= W || y >= H)
        return;
    lfi_int2 c;
    if(all_focus)
    {
        const float f = decode_focus(a.maps + (size_t)a.map_index * W * H * 4, W, H, x, y, a.focus, a.range);
        const lfi_float2 off = a.offsets[g];
        c.x = warp_float(x, f, off.x);
        c.y = warp_float(y, f, off.y);
    }
    else
    {
        const lfi_int2 off = a.focused[g];
        c.x = x + off.x;
        c.y = y + off.y;
    }
    out[(size_t)y * W + x] = c;
}

// synthetic light field of SURVEY.md §8(d), same hash as oracle lfo_hash32
__global__ void __launch_bounds__(256) fill_synthetic(uint8_t *__restrict__ grid, const int n_images, const int W, const int H,
                                                     const uint32_t seed)
{
    const size_t total = (size_t)n_images * W * H;
    for(size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x)
    {
        const uint32_t x = uint32_t(idx % W);
        const uint32_t y = uint32_t((idx / W) % H);
        const uint32_t g = uint32_t(idx / ((size_t)W * H));
        uint32_t hy = mix32(mix32(seed + g * 0x9e3779b9u) + y * 0x85ebca6bu);
        uint32_t px = 0xff000000u;
#pragma unroll
        for(uint32_t c = 0; c < 3; c++)
            px |= (mix32(hy + x * 0xc2b2ae35u + c) >> 24) << (8 * c);
        reinterpret_cast<uint32_t *>(grid)[idx] = px;
    }
}

} // namespace lfi
