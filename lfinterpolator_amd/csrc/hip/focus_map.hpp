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


// ------------------------------------------------------------------------------------------------------------------------
// focus_estimate_packed — the same function as focus_estimate, restructured for the L1/VALU limits that bound it
// (9216 taps per pixel): each lane owns FOUR consecutive pixels of a row, so a tap is one 16-byte load whenever the four
// warped coordinates are consecutive and unclamped in x (checked per lane with the exact per-pixel arithmetic; otherwise the
// whole wave takes per-pixel clamped fetches — identical results); the per-channel min/max over the views run on
// v_pk_min_u16 / v_pk_max_u16 with two pixels per instruction (bytes widened to u16 pairs by v_perm_b32).
//
// Exactness of the integer formulation against the reference's float one (src/kernels.cu:173-217): pixel values are integers
// 0..255, so every min/max/range is an exact integer — except that the reference starts its running maximum at FLT_MIN
// (:178), so a channel whose samples are all zero yields |0 - FLT_MIN| = FLT_MIN instead of 0.  A tap's dispersion is
// therefore its integer range when that is ≥ 1, else FLT_MIN if any channel's maximum is zero, else 0; the float sum over the
// nine taps equals the integer sum S when S ≥ 1 (the tiny terms are absorbed) and k·FLT_MIN otherwise, k = number of FLT_MIN
// taps.  Comparing the keys (S > 0 ? 16·S : k), k ≤ 9, reproduces "total < best" exactly, first strict minimum included.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u16x2 as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ uint32_t as_u32(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }

template <int C>
__device__ __forceinline__ u16x2 channel_pair(uint32_t px_a, uint32_t px_b)
{
    // [15:0] = byte C of px_a, [31:16] = byte C of px_b
    constexpr uint32_t sel = 0x0c000c00u | uint32_t(C) | (uint32_t(4 + C) << 16);
    return as_u16x2(__builtin_amdgcn_perm(px_b, px_a, sel));
}

// PPL pixels per lane (2 or 4); one wave per workgroup (a wave lives ~1 ms: fine-grained dispatch fills the tail)
template <int PPL, int WPE>
__global__ void __launch_bounds__(64, WPE) focus_estimate_packed(const KernelArgs a)
{
    constexpr int NP = PPL / 2; // pixel pairs per lane
    const int lane = threadIdx.x & 63;
    const int W = a.width, H = a.height;
    const int x0 = (blockIdx.x * 64 + lane) * PPL;     // first of this lane's pixels
    const int y = a.map_y0 + blockIdx.y;               // one row per wave, of the rows asked for (a row window computes a band)
    if(y >= min(H, a.map_y0 + a.map_rows))
        return; // wave-uniform
    const bool lane_active = x0 < W;
    constexpr int STEPS = 32; // src/kernels.cu:245
    const float step = __fdiv_rn(a.range, static_cast<float>(STEPS - 1));
    const int rx = a.radius_x, ry = a.radius_y;
    const uint32_t *grid32 = reinterpret_cast<const uint32_t *>(a.grid);
    const size_t plane_px = (size_t)W * (size_t)a.in_rows; // the rows this context holds (the host checked that they cover the samples)
    typedef const __attribute__((address_space(4))) float *const_float_ptr;
    typedef const __attribute__((address_space(4))) int32_t *const_int_ptr;
    const const_float_ptr c_offsets = (const_float_ptr)(uintptr_t)a.offsets;
    const const_int_ptr c_ids = (const_int_ptr)(uintptr_t)a.focus_ids;

    uint32_t best_key[PPL];
    int best_i[PPL];
#pragma unroll
    for(int j = 0; j < PPL; j++)
    {
        best_key[j] = 0xffffffffu;
        best_i[j] = 0;
    }

    for(int i = 0; i < STEPS; i++)
    {
        const float f = __builtin_fmaf(step, static_cast<float>(i), a.focus);
        // running min / max per tap (9), pixel pair (2: pixels {0,1} and {2,3}) and channel (3), as u16 pairs
        u16x2 lo[9][NP][3], hi[9][NP][3];
#pragma unroll
        for(int t = 0; t < 9; t++)
#pragma unroll
            for(int p = 0; p < NP; p++)
#pragma unroll
                for(int c = 0; c < 3; c++)
                {
                    lo[t][p][c] = as_u16x2(0x00ff00ffu);
                    hi[t][p][c] = as_u16x2(0u);
                }
        for(int k = 0; k < a.n_focus_ids; k++)
        {
            const int g = c_ids[k];
            const float offx = c_offsets[2 * g], offy = c_offsets[2 * g + 1];
            const uint32_t *plane = grid32 + (size_t)g * plane_px;
            int cx[PPL];
#pragma unroll
            for(int j = 0; j < PPL; j++)
                cx[j] = warp_float(x0 + j, f, offx);
            const int cy = warp_float(y, f, offy);
            // 16-byte taps are valid when the four sample columns are consecutive and no x-clamp can touch them
            bool consecutive = true;
#pragma unroll
            for(int j = 1; j < PPL; j++)
                consecutive = consecutive && (cx[j] == cx[0] + j);
            // per lane: one wide load per tap is valid when the sample columns are consecutive and no x-clamp can touch them
            const bool vec_ok = consecutive && (cx[0] - rx >= 0) && (cx[PPL - 1] + rx <= W - 1) && (x0 + PPL - 1 < W);
            if(lane_active)
            {
#pragma unroll
                for(int ty = 0; ty < 3; ty++)
                {
                    const uint32_t *row = plane + (size_t)(clampi(cy + (ty - 1) * ry, 0, H - 1) - a.in_y0) * W; // clamp in the image, index the held rows
#pragma unroll
                    for(int tx = 0; tx < 3; tx++)
                    {
                        uint32_t px[PPL];
                        if(vec_ok)
                        {
                            if constexpr(PPL == 4)
                            {
                                const u32x4_a4 v = *reinterpret_cast<const u32x4_a4 *>(row + cx[0] + (tx - 1) * rx);
                                px[0] = v.x;
                                px[1] = v.y;
                                px[2] = v.z;
                                px[3] = v.w;
                            }
                            else
                            {
                                const u32x2_a4 v = *reinterpret_cast<const u32x2_a4 *>(row + cx[0] + (tx - 1) * rx);
                                px[0] = v.x;
                                px[1] = v.y;
                            }
                        }
                        else
                        {
#pragma unroll
                            for(int j = 0; j < PPL; j++)
                                px[j] = row[clampi(cx[j] + (tx - 1) * rx, 0, W - 1)];
                        }
                        const int t = tx * 3 + ty;
#pragma unroll
                        for(int p = 0; p < NP; p++)
                        {
                            const u16x2 cr = channel_pair<0>(px[2 * p], px[2 * p + 1]);
                            const u16x2 cg = channel_pair<1>(px[2 * p], px[2 * p + 1]);
                            const u16x2 cb = channel_pair<2>(px[2 * p], px[2 * p + 1]);
                            lo[t][p][0] = __builtin_elementwise_min(lo[t][p][0], cr);
                            hi[t][p][0] = __builtin_elementwise_max(hi[t][p][0], cr);
                            lo[t][p][1] = __builtin_elementwise_min(lo[t][p][1], cg);
                            hi[t][p][1] = __builtin_elementwise_max(hi[t][p][1], cg);
                            lo[t][p][2] = __builtin_elementwise_min(lo[t][p][2], cb);
                            hi[t][p][2] = __builtin_elementwise_max(hi[t][p][2], cb);
                        }
                    }
                }
            }
        }
        // dispersion of this focus candidate: integer sum S and FLT_MIN-tap count k per pixel
#pragma unroll
        for(int p = 0; p < NP; p++)
        {
            u16x2 sum = as_u16x2(0u), nonzero = as_u16x2(0u);
#pragma unroll
            for(int t = 0; t < 9; t++)
            {
                const u16x2 d0 = hi[t][p][0] - lo[t][p][0], d1 = hi[t][p][1] - lo[t][p][1], d2 = hi[t][p][2] - lo[t][p][2];
                const u16x2 dmax = __builtin_elementwise_max(__builtin_elementwise_max(d0, d1), d2);
                const u16x2 hmin = __builtin_elementwise_min(__builtin_elementwise_min(hi[t][p][0], hi[t][p][1]), hi[t][p][2]);
                sum += dmax;
                // a tap is a "FLT_MIN tap" iff its range is 0 and some channel's maximum is 0; flag = 0 exactly then
                const u16x2 flag = __builtin_elementwise_min(as_u16x2(as_u32(dmax) | as_u32(hmin)), as_u16x2(0x00010001u));
                nonzero += flag;
            }
#pragma unroll
            for(int q = 0; q < 2; q++)
            {
                const uint32_t S = q ? (as_u32(sum) >> 16) : (as_u32(sum) & 0xffffu);
                const uint32_t kq = 9u - (q ? (as_u32(nonzero) >> 16) : (as_u32(nonzero) & 0xffffu));
                const uint32_t key = S > 0 ? (S << 4) : kq;
                const int j = 2 * p + q;
                // MinDispersion::add (src/kernels.cu:225-231): strict <, starting from FLT_MAX (every key is below 0xffffffff)
                if(key < best_key[j])
                {
                    best_key[j] = key;
                    best_i[j] = i;
                }
            }
        }
    }
    if(!lane_active)
        return;
    uint32_t out[PPL];
#pragma unroll
    for(int j = 0; j < PPL; j++)
    {
        const float best_f = __builtin_fmaf(step, static_cast<float>(best_i[j]), a.focus);
        const float normalized = __fdiv_rn(best_f - a.focus, a.range);
        const uint32_t m = static_cast<uint32_t>(roundf(normalized * 255.0f)) & 0xffu;
        out[j] = m | (m << 8) | (m << 16) | 0xff000000u;
    }
    uint32_t *dst = reinterpret_cast<uint32_t *>(a.maps) + (size_t)y * W + x0;
#pragma unroll
    for(int j = 0; j < PPL; j++)
        if(x0 + j < W)
            dst[j] = out[j];
}

// ------------------------------------------------------------------------------------------------------------------------
// focus_estimate_lds — packed formulation with the source window staged in LDS.
// The packed kernel above is limited by L1 tag traffic: nine taps per (view, candidate) per pixel, each wave-load touching
// two cache lines per lane-quad, although the three x-taps overlap almost completely.  Here one wave = 128 pixels of a row
// (lane l owns pixels l and l+64).  For each (candidate, view) the wave copies the three source rows it needs — a window of
// 128 + 2·rx + 2·SLACK pixels starting just left of lane 0's sample — with ONE 16 B/lane load per row into LDS, and all
// 18 taps of a lane are conflict-free ds_read_b32 at its exact per-pixel coordinates.  The window of the NEXT (candidate, view)
// is fetched into registers while the current one is processed.  If any lane's taps would leave the staged window or touch an
// x-clamp (image borders; the odd float-rounding outlier is covered by SLACK), the wave falls back to per-pixel clamped
// fetches for that iteration.  Same integer (sum, FLT_MIN-count) key as focus_estimate_packed: bit-exact.
constexpr int FOCUS_LDS_ROW = 256; // window pixels per row (max 128 + 2*rx + 2*SLACK)
constexpr int FOCUS_LDS_SLACK = 4;

__global__ void __launch_bounds__(64, 4) focus_estimate_lds(const KernelArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t win[3 * FOCUS_LDS_ROW];
    const int lane = threadIdx.x & 63;
    const int W = a.width, H = a.height;
    const int X0 = blockIdx.x * 128;
    const int y = blockIdx.y;
    const int xa = X0 + lane, xb = X0 + 64 + lane; // this lane's two pixels
    constexpr int STEPS = 32;                      // src/kernels.cu:245
    const float step = __fdiv_rn(a.range, static_cast<float>(STEPS - 1));
    const int rx = a.radius_x, ry = a.radius_y;
    const int need = 128 + 2 * rx + 2 * FOCUS_LDS_SLACK; // ≤ FOCUS_LDS_ROW, checked by the host
    const uint32_t *grid32 = reinterpret_cast<const uint32_t *>(a.grid);
    const size_t plane_px = (size_t)W * (size_t)H;
    typedef const __attribute__((address_space(4))) float *const_float_ptr;
    typedef const __attribute__((address_space(4))) int32_t *const_int_ptr;
    const const_float_ptr c_offsets = (const_float_ptr)(uintptr_t)a.offsets;
    const const_int_ptr c_ids = (const_int_ptr)(uintptr_t)a.focus_ids;
    const int n_ids = a.n_focus_ids;
    const int total = STEPS * n_ids;

    // per-iteration geometry: everything a (candidate i, view k) pair needs to stage and to tap
    struct Geo
    {
        const uint32_t *plane;
        int cxa, cxb, cy, wx0;
        bool fast;
    };
    auto geometry = [&](int i, int k) -> Geo {
        const float f = __builtin_fmaf(step, static_cast<float>(i), a.focus);
        const int g = c_ids[k];
        const float offx = c_offsets[2 * g], offy = c_offsets[2 * g + 1];
        Geo q;
        q.plane = grid32 + (size_t)g * plane_px;
        q.cxa = warp_float(xa, f, offx);
        q.cxb = warp_float(xb, f, offx);
        q.cy = warp_float(y, f, offy);
        q.wx0 = __builtin_amdgcn_readfirstlane(q.cxa) - rx - FOCUS_LDS_SLACK;
        const bool in_a = xa >= W || (q.cxa - rx >= q.wx0 && q.cxa + rx < q.wx0 + need);
        const bool in_b = xb >= W || (q.cxb - rx >= q.wx0 && q.cxb + rx < q.wx0 + need);
        const bool ok = in_a && in_b && q.wx0 >= 0 && q.wx0 + need <= W;
        q.fast = __builtin_amdgcn_ballot_w64(ok) == ~0ull;
        return q;
    };
    auto fetch_rows = [&](const Geo &q, u32x4 (&rows)[3]) {
        if(q.fast && 4 * lane < need)
        {
#pragma unroll
            for(int ty = 0; ty < 3; ty++)
                rows[ty] = *reinterpret_cast<const u32x4_a4 *>(q.plane + (size_t)clampi(q.cy + (ty - 1) * ry, 0, H - 1) * W + q.wx0 + 4 * lane);
        }
    };

    uint32_t best_key[2] = {0xffffffffu, 0xffffffffu};
    int best_i[2] = {0, 0};
    u16x2 lo[9][3], hi[9][3]; // [tap][channel], u16 pair = (pixel a, pixel b)

    Geo cur = geometry(0, 0);
    u32x4 rows[3];
    fetch_rows(cur, rows);
    int i = 0, k = 0;
    for(int it = 0; it < total; it++)
    {
        if(k == 0)
        {
#pragma unroll
            for(int t = 0; t < 9; t++)
#pragma unroll
                for(int c = 0; c < 3; c++)
                {
                    lo[t][c] = as_u16x2(0x00ff00ffu);
                    hi[t][c] = as_u16x2(0u);
                }
        }
        // stage the current window (its loads were issued one iteration ago) …
        if(cur.fast && 4 * lane < need)
        {
#pragma unroll
            for(int ty = 0; ty < 3; ty++)
                *reinterpret_cast<u32x4 *>(win + ty * FOCUS_LDS_ROW + 4 * lane) = rows[ty];
        }
        // … and start fetching the next one
        Geo nxt = cur;
        const int nk = k + 1 == n_ids ? 0 : k + 1, ni = k + 1 == n_ids ? i + 1 : i;
        if(it + 1 < total)
        {
            nxt = geometry(ni, nk);
            fetch_rows(nxt, rows);
        }
        // taps (single-wave workgroup: the LDS queue is in order, so the reads below see the writes above)
        uint32_t pa[9], pb[9];
        if(cur.fast)
        {
            const uint32_t *wa = win + (cur.cxa - cur.wx0), *wb = win + (cur.cxb - cur.wx0);
#pragma unroll
            for(int ty = 0; ty < 3; ty++)
#pragma unroll
                for(int tx = 0; tx < 3; tx++)
                {
                    pa[tx * 3 + ty] = wa[ty * FOCUS_LDS_ROW + (tx - 1) * rx];
                    pb[tx * 3 + ty] = wb[ty * FOCUS_LDS_ROW + (tx - 1) * rx];
                }
        }
        else
        {
#pragma unroll
            for(int ty = 0; ty < 3; ty++)
            {
                const uint32_t *row = cur.plane + (size_t)clampi(cur.cy + (ty - 1) * ry, 0, H - 1) * W;
#pragma unroll
                for(int tx = 0; tx < 3; tx++)
                {
                    pa[tx * 3 + ty] = row[clampi(cur.cxa + (tx - 1) * rx, 0, W - 1)];
                    pb[tx * 3 + ty] = row[clampi(cur.cxb + (tx - 1) * rx, 0, W - 1)];
                }
            }
        }
#pragma unroll
        for(int t = 0; t < 9; t++)
        {
            const u16x2 cr = channel_pair<0>(pa[t], pb[t]), cg = channel_pair<1>(pa[t], pb[t]), cb = channel_pair<2>(pa[t], pb[t]);
            lo[t][0] = __builtin_elementwise_min(lo[t][0], cr);
            hi[t][0] = __builtin_elementwise_max(hi[t][0], cr);
            lo[t][1] = __builtin_elementwise_min(lo[t][1], cg);
            hi[t][1] = __builtin_elementwise_max(hi[t][1], cg);
            lo[t][2] = __builtin_elementwise_min(lo[t][2], cb);
            hi[t][2] = __builtin_elementwise_max(hi[t][2], cb);
        }
        if(k == n_ids - 1)
        {
            u16x2 sum = as_u16x2(0u), nonzero = as_u16x2(0u);
#pragma unroll
            for(int t = 0; t < 9; t++)
            {
                const u16x2 d0 = hi[t][0] - lo[t][0], d1 = hi[t][1] - lo[t][1], d2 = hi[t][2] - lo[t][2];
                const u16x2 dmax = __builtin_elementwise_max(__builtin_elementwise_max(d0, d1), d2);
                const u16x2 hmin = __builtin_elementwise_min(__builtin_elementwise_min(hi[t][0], hi[t][1]), hi[t][2]);
                sum += dmax;
                nonzero += __builtin_elementwise_min(as_u16x2(as_u32(dmax) | as_u32(hmin)), as_u16x2(0x00010001u));
            }
#pragma unroll
            for(int q = 0; q < 2; q++)
            {
                const uint32_t S = q ? (as_u32(sum) >> 16) : (as_u32(sum) & 0xffffu);
                const uint32_t kq = 9u - (q ? (as_u32(nonzero) >> 16) : (as_u32(nonzero) & 0xffffu));
                const uint32_t key = S > 0 ? (S << 4) : kq;
                if(key < best_key[q])
                {
                    best_key[q] = key;
                    best_i[q] = i;
                }
            }
        }
        cur = nxt;
        i = ni;
        k = nk;
    }
#pragma unroll
    for(int q = 0; q < 2; q++)
    {
        const int x = q ? xb : xa;
        if(x < W)
        {
            const float best_f = __builtin_fmaf(step, static_cast<float>(best_i[q]), a.focus);
            const float normalized = __fdiv_rn(best_f - a.focus, a.range);
            const uint32_t m = static_cast<uint32_t>(roundf(normalized * 255.0f)) & 0xffu;
            reinterpret_cast<uint32_t *>(a.maps)[(size_t)y * W + x] = m | (m << 8) | (m << 16) | 0xff000000u;
        }
    }
}

__global__ void __launch_bounds__(256) focus_filter(const KernelArgs a)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = a.map_y0 + blockIdx.y * 4 + (threadIdx.x >> 6); // the rows asked for (a row window filters its band)
    const int W = a.width, H = a.height;
    if(x >= W || y >= min(H, a.map_y0 + a.map_rows))
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

// focus_filter from LDS (round 5).  The box mean sums integers 0…255 — exact in fp32 in any order while the window holds fewer than 65,793 taps —
// so the sum is taken separably: a workgroup stages the low bytes of map 0 under its 64 × 32 tile plus the window's reach ((64 + 2·rx) ×
// (32 + 2·ry) values, clamped per tap like the reference's loops), sums 2·rx of them per element along x, then walks down its columns with a
// running sum of 2·ry of those.  One dword load per staged pixel instead of 4·rx·ry per output pixel (16 at the 4K configuration: the plain
// kernel runs at the L1's rate, 0.13 ms for an 8 MB map).  Both LDS arrays hold DWORDS: with bytes and u16 the LDS was busy 72 % of the run
// time at 28 cycles per wave-access (sub-dword accesses of neighbouring lanes serialise) and the kernel took 60 µs.  The host checks
// rx, ry ≤ 128 and the LDS size; larger windows take focus_filter.
constexpr int FF_TW = 64, FF_TH = 32;
inline size_t focus_filter_tiled_lds(const int rx, const int ry) { return 4u * (size_t(FF_TW + 2 * rx) + FF_TW) * size_t(FF_TH + 2 * ry); }

__global__ void __launch_bounds__(256) focus_filter_tiled(const KernelArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t ff_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int W = a.width, H = a.height;
    const int rx = max(a.radius_x / 10, 1), ry = max(a.radius_y / 10, 1); // ≥1: SURVEY.md defect D6
    const int PW = FF_TW + 2 * rx, PH = FF_TH + 2 * ry;
    const int x0 = blockIdx.x * FF_TW, y0 = a.map_y0 + blockIdx.y * FF_TH;
    const int y_end = min(H, a.map_y0 + a.map_rows);
    uint32_t *const vals = ff_lds, *const hsum = ff_lds + PW * PH;
    const uint32_t *map0 = reinterpret_cast<const uint32_t *>(a.maps);
    // (the staged region is walked linearly by the whole workgroup, twelve elements per thread whose loads are all issued before the first value
    // is stored: one memory latency per tile at the usual radii, not one per row)
    constexpr int STAGE_N = 12;
    const int n_stage = PW * PH;
    const float inv_pw = 1.0f / static_cast<float>(PW);
    for(int base = threadIdx.x; base < n_stage; base += 256 * STAGE_N)
    {
        uint32_t px[STAGE_N];
#pragma unroll
        for(int k = 0; k < STAGE_N; k++)
        {
            const int idx = min(base + 256 * k, n_stage - 1); // (elements past the region: a harmless repeat of its last one, not stored)
            int r = static_cast<int>(static_cast<float>(idx) * inv_pw); // idx / PW, within one
            r += (r + 1) * PW <= idx ? 1 : (r * PW > idx ? -1 : 0);
            px[k] = map0[(size_t)clampi(y0 - ry + r, 0, H - 1) * W + clampi(x0 - rx + idx - r * PW, 0, W - 1)];
        }
#pragma unroll
        for(int k = 0; k < STAGE_N; k++)
            if(base + 256 * k < n_stage)
                vals[base + 256 * k] = px[k] & 0xffu;
    }
    __syncthreads();
    for(int r = wave; r < PH; r += 4)
    {
        uint32_t s = 0;
#pragma unroll 4
        for(int j = 0; j < 2 * rx; j++)
            s += vals[r * PW + lane + j];
        hsum[r * FF_TW + lane] = s;
    }
    __syncthreads();
    const int x = x0 + lane, r0 = wave * (FF_TH / 4);
    if(x >= W || y0 + r0 >= y_end)
        return;
    uint32_t s = 0;
#pragma unroll 4
    for(int j = 0; j < 2 * ry; j++)
        s += hsum[(r0 + j) * FF_TW + lane];
    const float count = static_cast<float>(4 * rx * ry);
    for(int i = 0; i < FF_TH / 4 && y0 + r0 + i < y_end; i++)
    {
        const uint32_t m = static_cast<uint32_t>(roundf(__fdiv_rn(static_cast<float>(s), count))) & 0xffu;
        reinterpret_cast<uint32_t *>(a.maps)[(size_t)W * H + (size_t)(y0 + r0 + i) * W + x] = m | (m << 8) | (m << 16) | 0xff000000u;
        s += hsum[(r0 + i + 2 * ry) * FF_TW + lane] - hsum[(r0 + i) * FF_TW + lane];
    }
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
// rows_held rows per plane starting at global row y0 (the whole image: rows_held = H, y0 = 0)
__global__ void __launch_bounds__(256) fill_synthetic(uint8_t *__restrict__ grid, const int g_first, const int n_images, const int W, const int H,
                                                     const int y0, const uint32_t seed)
{
    // images g_first … g_first + n_images − 1 (grid points at plane 0)
    grid += (size_t)g_first * W * H * 4;
    const size_t total = (size_t)n_images * W * H;
    for(size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x)
    {
        const uint32_t x = uint32_t(idx % W);
        const uint32_t y = uint32_t((idx / W) % H) + uint32_t(y0);
        const uint32_t g = uint32_t(idx / ((size_t)W * H)) + uint32_t(g_first);
        uint32_t hy = mix32(mix32(seed + g * 0x9e3779b9u) + y * 0x85ebca6bu);
        uint32_t px = 0xff000000u;
#pragma unroll
        for(uint32_t c = 0; c < 3; c++)
            px |= (mix32(hy + x * 0xc2b2ae35u + c) >> 24) << (8 * c);
        reinterpret_cast<uint32_t *>(grid)[idx] = px;
    }
}

// A structured synthetic light field for focus-map measurements: a texture T of 8×8-pixel cells of random
// colours seen at a piecewise-constant focus f*(block) — blocks of 1024×1024 pixels, each at one of four of the estimate's own
// candidates f_i = fma(range/31, i, focus), i ∈ {3, 11, 20, 28} (src/kernels.cu:245-249): image g shows
// T(x − f*·offset_g.x, y − f*·offset_g.y), so sampling every image at p + f·offset_g (focusCoords, reference src/kernels.cu:78-82)
// shows the same texel in all of them exactly when f = f* — the estimate then yields a piecewise-constant map, as real scenes do
// and hash noise does not.  Same grid-stride shape as fill_synthetic.
__global__ void __launch_bounds__(256) fill_scene(uint8_t *__restrict__ grid, const lfi_float2 *__restrict__ offsets, const int n_images, const int W,
                                                 const int H, const int y0, const uint32_t seed, const float focus, const float range)
{
    const size_t total = (size_t)n_images * W * H;
    for(size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x)
    {
        const int x = int(idx % W);
        const int y = int((idx / W) % H) + y0;
        const int g = int(idx / ((size_t)W * H));
        const uint32_t level = mix32(seed ^ (uint32_t(x >> 10) * 73856093u) ^ (uint32_t(y >> 10) * 19349663u)) & 3u;
        const float fs = __builtin_fmaf(__fdiv_rn(range, 31.0f), float(3u + 8u * level + (level >> 1)), focus); // candidates 3, 11, 20, 28
        const lfi_float2 off = offsets[g];
        // the integer shift the warp applies at f*: (int)fma(f, offset, coord) = coord + floor(f·offset) wherever the sum is ≥ 0
        const int u = x - int(floorf(fs * off.x)), v = y - int(floorf(fs * off.y));
        const uint32_t cell = mix32(mix32(seed + uint32_t(u >> 3) * 0x9e3779b9u) + uint32_t(v >> 3) * 0x85ebca6bu);
        reinterpret_cast<uint32_t *>(grid)[idx] = 0xff000000u | (cell & 0x00ffffffu);
    }
}

} // namespace lfi
