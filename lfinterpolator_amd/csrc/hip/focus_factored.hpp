// focus_factored.hpp — the focus-map estimate with the 3×3 tap block factored out (default "factored" variant).
//
// Same function as focus_estimate (focus_map.hpp; reference src/kernels.cu:196-258), 9× less sampling work.
//
// Observation: the reference samples view g for candidate focus f_i at  T_g,i(p) + t·r,  t ∈ {−1,0,1}²,  where
// T_g,i(p) = (int)fma(f_i, offset_g, p) (src/kernels.cu:78-82).  For almost every pixel T_g,i(p) = p + s_g,i with one
// integer shift s_g,i = floor(f_i·offset_g) per (view, candidate), so the tap (p, t) reads exactly what the CENTRE tap of
// pixel q = p + t·r reads.  The per-tap colour range therefore is a per-candidate image
//     E_i(q) = range over the views of I_g[clamp(q + s_g,i)]          (q over the image extended by r on every side)
// and the dispersion of a pixel is the sum of nine samples of E_i — 1024 instead of 9216 image samples per pixel.
//
// Exactness: the shortcut is used for (pixel, candidate) only where it is PROVEN identical.  focus_plan_flags evaluates,
// for every column x (row y), candidate and view, the reference's own coordinate arithmetic and compares the three CLAMPED
// tap coordinates with the ones the uniform shift gives; any difference (truncation toward zero left of / above the image,
// or a float rounding of fma(f, offset, x) across an integer) flags (x, i) / (y, i).  Flagged pairs — bands of ≈r columns /
// rows per view with a negative shift, a few % of all pairs — are evaluated tap by tap by focus_exact, a persistent kernel over
// compact lists of flagged columns and rows, into K; focus_pick takes K instead of the factored sum wherever a flag is set.
// The integer key (16·S + k, see focus_map.hpp) is the same as in the other variants, so results are bit-identical.
//
// Clamp-to-edge is taken out of the hot loops: focus_pad copies the ≤32 sampled images into planes padded by the largest
// shift + r on every side (edge pixels replicated), so every sample of focus_range and focus_exact is an unclamped load at
// scalar base + lane offset.
//
// Passes (all on the context's stream):  plan_shifts → plan_flags → plan_lists → plan_prefix, focus_pad → focus_range (E) →
// focus_exact (K) → focus_pick (map 0) → focus_filter (map 1).
#pragma once

#include "focus_map.hpp"

namespace lfi {

constexpr int FOCUS_STEPS = 32; // src/kernels.cu:245
constexpr int FOCUS_MAX_IDS = 32;

struct FocusWork
{
    int32_t *shifts;    // [32][32][4]  sx, sy, image id (−1: unused slot), 0
    uint32_t *badx;     // [W]   bit i: column x needs the exact path for candidate i
    uint32_t *bady;     // [H]
    uint16_t *cols;     // [32][W]  flagged columns of candidate i, ascending
    uint16_t *rows;     // [32][H]
    int32_t *ncols;     // [32]
    int32_t *nrows;     // [32]
    uint32_t *prefix;   // [33] prefix sums of nrows
    uint16_t *E;        // [32][He_p][We_p]  16·range + (FLT_MIN tap ? 1 : 0) over the extended image
    uint16_t *K;        // [32][H][W]        exact keys of flagged (pixel, candidate) pairs
    int32_t We_p, He_p; // pitches of E: W + 2rx rounded up to 256, H + 2ry rounded up to 4
    int64_t *deltas;    // [32][32]  byte offset of (slot k's padded plane, row sy, column sx) from the padded base
    uint32_t *pad;      // [n_ids][Hp][Wp]  padded copies: pad[k][yy][xx] = I_ids[k][clamp(yy − Py)][clamp(xx − Px)]
    int32_t Wp, Hp, Px, Py;
};

typedef const __attribute__((address_space(4))) int64_t *focus_const_i64_ptr;

typedef const __attribute__((address_space(4))) int32_t *focus_const_int_ptr;
typedef const __attribute__((address_space(4))) float *focus_const_float_ptr;
typedef const __attribute__((address_space(4))) uint32_t *focus_const_u32_ptr;

__device__ __forceinline__ float focus_candidate(const KernelArgs &a, int i)
{
    const float step = __fdiv_rn(a.range, static_cast<float>(FOCUS_STEPS - 1));
    return __builtin_fmaf(step, static_cast<float>(i), a.focus);
}

// one thread per (candidate, view slot)
__global__ void __launch_bounds__(1024) focus_plan_shifts(const KernelArgs a, const FocusWork w)
{
    const int i = threadIdx.x >> 5, k = threadIdx.x & 31;
    int32_t *dst = w.shifts + 4 * (i * FOCUS_MAX_IDS + k);
    if(k >= a.n_focus_ids)
    {
        dst[0] = dst[1] = dst[3] = 0;
        dst[2] = -1;
        return;
    }
    const float f = focus_candidate(a, i);
    const int g = a.focus_ids[k];
    const lfi_float2 off = a.offsets[g];
    // the product of two floats is exact in double: the shift every pixel far from a rounding boundary gets
    dst[0] = static_cast<int>(floor(static_cast<double>(f) * static_cast<double>(off.x)));
    dst[1] = static_cast<int>(floor(static_cast<double>(f) * static_cast<double>(off.y)));
    dst[2] = g;
    dst[3] = 0;
    w.deltas[i * FOCUS_MAX_IDS + k] = (((int64_t)k * w.Hp + dst[1]) * w.Wp + dst[0]) * 4;
}

// pad[k][yy][xx] = I_ids[k][clamp(yy − Py)][clamp(xx − Px)]; grid (ceil(Wp/256), Hp, n_ids), a lane writes 4 pixels
__global__ void __launch_bounds__(64) focus_pad(const KernelArgs a, const FocusWork w)
{
    const int xx = (blockIdx.x * 64 + threadIdx.x) * 4;
    if(xx >= w.Wp)
        return;
    const int yy = blockIdx.y, k = blockIdx.z;
    const int W = a.width, H = a.height;
    const uint32_t *row = reinterpret_cast<const uint32_t *>(a.grid) + (size_t)a.focus_ids[k] * ((size_t)W * H) +
                          (size_t)clampi(yy - w.Py, 0, H - 1) * W;
    const int x = xx - w.Px;
    u32x4 v;
    if(x >= 0 && x + 3 <= W - 1)
        v = *reinterpret_cast<const u32x4_a4 *>(row + x);
    else
    {
        v.x = row[clampi(x, 0, W - 1)];
        v.y = row[clampi(x + 1, 0, W - 1)];
        v.z = row[clampi(x + 2, 0, W - 1)];
        v.w = row[clampi(x + 3, 0, W - 1)];
    }
    *reinterpret_cast<u32x4 *>(w.pad + ((size_t)k * w.Hp + yy) * w.Wp + xx) = v; // Wp is a multiple of 4
}

// grid (ceil(max(W,H)/256), 32 candidates, 2 axes): is the uniform shift exact for this column / row?
__global__ void __launch_bounds__(256) focus_plan_flags(const KernelArgs a, const FocusWork w)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    const int axis = blockIdx.z;
    const int L = axis ? a.height : a.width;
    if(c >= L)
        return;
    const int r = axis ? a.radius_y : a.radius_x;
    const float f = focus_candidate(a, i);
    bool bad = false;
    for(int k = 0; k < a.n_focus_ids; k++)
    {
        const int32_t *s = w.shifts + 4 * (i * FOCUS_MAX_IDS + k);
        const lfi_float2 off = a.offsets[s[2]];
        const int T = warp_float(c, f, axis ? off.y : off.x);
        const int U = c + s[axis];
#pragma unroll
        for(int t = -1; t <= 1; t++)
            bad = bad || (clampi(T + t * r, 0, L - 1) != clampi(U + t * r, 0, L - 1));
    }
    if(bad)
        atomicOr((axis ? w.bady : w.badx) + c, 1u << i);
}

// grid (32 candidates, 2 axes), one wave each: ordered compaction of the flagged columns / rows
__global__ void __launch_bounds__(64) focus_plan_lists(const KernelArgs a, const FocusWork w)
{
    const int i = blockIdx.x, axis = blockIdx.y, lane = threadIdx.x;
    const int L = axis ? a.height : a.width;
    const uint32_t *bad = axis ? w.bady : w.badx;
    uint16_t *list = (axis ? w.rows : w.cols) + (size_t)i * L;
    int count = 0;
    for(int c0 = 0; c0 < L; c0 += 64)
    {
        const int c = c0 + lane;
        const bool flagged = c < L && ((bad[c] >> i) & 1u);
        const uint64_t m = __builtin_amdgcn_ballot_w64(flagged);
        if(flagged)
            list[count + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = static_cast<uint16_t>(c);
        count += __builtin_popcountll(m);
    }
    if(lane == 0)
        (axis ? w.nrows : w.ncols)[i] = count;
}

// one thread: prefix sums of the flagged-row counts (focus_exact walks the flagged rows of all candidates as one sequence)
__global__ void focus_plan_prefix(const KernelArgs a, const FocusWork w)
{
    uint32_t sum = 0;
    for(int i = 0; i < FOCUS_STEPS; i++)
    {
        w.prefix[i] = sum;
        sum += uint32_t(w.nrows[i]);
    }
    w.prefix[FOCUS_STEPS] = sum;
}

// E_i(q): one workgroup = 256 extended columns × 4 extended rows (one row per wave) of CPW consecutive candidates; a lane
// owns 4 consecutive pixels.  The CPW candidates of a wave read nearly the same source lines (their shifts differ by a few
// pixels), back to back, so all but the first are L1 hits.  Work order: candidate group fastest, tiles row-major, one
// contiguous run of the sequence per XCD (all 32 candidates of a tile meet in one L2).
template <int CPW>
__global__ void __launch_bounds__(256) focus_range(const KernelArgs a, const FocusWork w, const uint32_t nblocks, const int striped)
{
    constexpr int GROUPS = FOCUS_STEPS / CPW;
    const uint32_t tiles_x = uint32_t(w.We_p) >> 8, tiles_y = uint32_t(w.He_p) >> 2;
    uint32_t tx, ty, group;
    if(striped)
    {
        if(!stripe_map(blockIdx.x, tiles_x, tiles_y, GROUPS, tx, ty, group))
            return;
    }
    else
    {
        const uint32_t work = xcd_contiguous(blockIdx.x, nblocks);
        group = work % GROUPS;
        tx = (work / GROUPS) % tiles_x;
        ty = (work / GROUPS) / tiles_x;
    }
    const int i0 = int(group) * CPW;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ey = ty * 4 + wave;              // row in E
    const int qy = ey - a.radius_y;            // extended image row
    const int qx_wave = tx * 256 - a.radius_x; // extended column of lane 0's first pixel
    const focus_const_i64_ptr deltas = (focus_const_i64_ptr)(uintptr_t)(w.deltas + i0 * FOCUS_MAX_IDS);
    // wave-uniform base: the padded position of (qx_wave, qy) in slot 0; always ≥ one row / column inside the padding
    const uint8_t *wave_base = reinterpret_cast<const uint8_t *>(w.pad) + ((size_t)(qy + w.Py) * w.Wp + (size_t)(qx_wave + w.Px)) * 4;
    const uint32_t lane_off = 16u * lane;

    u16x2 lo[CPW][2][3], hi[CPW][2][3]; // [candidate][pixel pair][channel]
#pragma unroll
    for(int c = 0; c < CPW; c++)
#pragma unroll
        for(int p = 0; p < 2; p++)
#pragma unroll
            for(int ch = 0; ch < 3; ch++)
            {
                lo[c][p][ch] = as_u16x2(0x00ff00ffu);
                hi[c][p][ch] = as_u16x2(0u);
            }
    const int n_ids = a.n_focus_ids;
    for(int k = 0; k < n_ids; k++)
    {
        uint32_t px[CPW][4];
#pragma unroll
        for(int c = 0; c < CPW; c++)
        {
            const u32x4_a4 v = *reinterpret_cast<const u32x4_a4 *>(wave_base + deltas[c * FOCUS_MAX_IDS + k] + lane_off);
            px[c][0] = v.x;
            px[c][1] = v.y;
            px[c][2] = v.z;
            px[c][3] = v.w;
        }
#pragma unroll
        for(int c = 0; c < CPW; c++)
#pragma unroll
            for(int p = 0; p < 2; p++)
            {
                const u16x2 cr = channel_pair<0>(px[c][2 * p], px[c][2 * p + 1]);
                const u16x2 cg = channel_pair<1>(px[c][2 * p], px[c][2 * p + 1]);
                const u16x2 cb = channel_pair<2>(px[c][2 * p], px[c][2 * p + 1]);
                lo[c][p][0] = __builtin_elementwise_min(lo[c][p][0], cr);
                hi[c][p][0] = __builtin_elementwise_max(hi[c][p][0], cr);
                lo[c][p][1] = __builtin_elementwise_min(lo[c][p][1], cg);
                hi[c][p][1] = __builtin_elementwise_max(hi[c][p][1], cg);
                lo[c][p][2] = __builtin_elementwise_min(lo[c][p][2], cb);
                hi[c][p][2] = __builtin_elementwise_max(hi[c][p][2], cb);
            }
    }
#pragma unroll
    for(int c = 0; c < CPW; c++)
    {
        u32x2 out;
#pragma unroll
        for(int p = 0; p < 2; p++)
        {
            const u16x2 d0 = hi[c][p][0] - lo[c][p][0], d1 = hi[c][p][1] - lo[c][p][1], d2 = hi[c][p][2] - lo[c][p][2];
            const u16x2 dmax = __builtin_elementwise_max(__builtin_elementwise_max(d0, d1), d2);
            const u16x2 hmin = __builtin_elementwise_min(__builtin_elementwise_min(hi[c][p][0], hi[c][p][1]), hi[c][p][2]);
            // FLT_MIN tap (focus_map.hpp): range 0 and an all-zero channel
            const u16x2 nz = __builtin_elementwise_min(as_u16x2(as_u32(dmax) | as_u32(hmin)), as_u16x2(0x00010001u));
            const uint32_t e = (as_u32(dmax) << 4) + (0x00010001u - as_u32(nz)); // per half: 16·range ≤ 4080, no carry
            if(p == 0)
                out.x = e;
            else
                out.y = e;
        }
        uint16_t *dst = w.E + ((size_t)(i0 + c) * w.He_p + ey) * w.We_p + tx * 256 + 4 * lane;
        *reinterpret_cast<u32x2 *>(dst) = out;
    }
}

// exact key of one (pixel, candidate): the reference's arithmetic tap by tap, in the integer formulation of focus_map.hpp.
// Samples come from the padded planes: the unclamped coordinate, offset by (Px, Py), holds the clamp-to-edge value.
__device__ __forceinline__ uint32_t focus_exact_key(const KernelArgs &a, const FocusWork &w, const int x, const int y, const int i)
{
    const int rx = a.radius_x, ry = a.radius_y;
    const float f = focus_candidate(a, i);
    const focus_const_float_ptr c_offsets = (focus_const_float_ptr)(uintptr_t)a.offsets;
    const focus_const_int_ptr c_ids = (focus_const_int_ptr)(uintptr_t)a.focus_ids;
    const size_t plane_bytes = (size_t)w.Wp * w.Hp * 4;
    // per tap: (R, G) as a u16 pair and B
    u16x2 lo_rg[9], hi_rg[9], lo_b[9], hi_b[9];
#pragma unroll
    for(int t = 0; t < 9; t++)
    {
        lo_rg[t] = lo_b[t] = as_u16x2(0x00ff00ffu);
        hi_rg[t] = hi_b[t] = as_u16x2(0u);
    }
    const uint8_t *plane = reinterpret_cast<const uint8_t *>(w.pad);
    for(int k = 0; k < a.n_focus_ids; k++, plane += plane_bytes)
    {
        const int g = c_ids[k];
        const float offx = c_offsets[2 * g], offy = c_offsets[2 * g + 1];
        const int cx = warp_float(x, f, offx), cy = warp_float(y, f, offy);
        // top-left tap; all nine taps are at non-negative, wave-uniform byte offsets from it
        const uint32_t corner = uint32_t((cy - ry + w.Py) * w.Wp + (cx - rx + w.Px)) * 4u;
#pragma unroll
        for(int ty = 0; ty < 3; ty++)
#pragma unroll
            for(int tx = 0; tx < 3; tx++)
            {
                const uint32_t tap = uint32_t(ty * ry * w.Wp + tx * rx) * 4u;
                const uint32_t px = *reinterpret_cast<const uint32_t *>(plane + tap + corner);
                const int t = tx * 3 + ty;
                const u16x2 rg = as_u16x2(__builtin_amdgcn_perm(0u, px, 0x0c010c00u));
                const u16x2 b = as_u16x2(__builtin_amdgcn_perm(0u, px, 0x0c020c02u));
                lo_rg[t] = __builtin_elementwise_min(lo_rg[t], rg);
                hi_rg[t] = __builtin_elementwise_max(hi_rg[t], rg);
                lo_b[t] = __builtin_elementwise_min(lo_b[t], b);
                hi_b[t] = __builtin_elementwise_max(hi_b[t], b);
            }
    }
    uint32_t S = 0, kmin = 0;
#pragma unroll
    for(int t = 0; t < 9; t++)
    {
        const uint32_t d_rg = as_u32(hi_rg[t] - lo_rg[t]), d_b = as_u32(hi_b[t] - lo_b[t]) & 0xffffu;
        const uint32_t dmax = max(max(d_rg & 0xffffu, d_rg >> 16), d_b);
        const uint32_t h_rg = as_u32(hi_rg[t]);
        const uint32_t hmin = min(min(h_rg & 0xffffu, h_rg >> 16), as_u32(hi_b[t]) & 0xffffu);
        S += dmax;
        kmin += (dmax | hmin) == 0u ? 1u : 0u;
    }
    return S > 0 ? (S << 4) : kmin;
}

// Persistent kernel over the flagged (pixel, candidate) pairs.  The work is split by XCD (blocks b, b+8, … share one): XCD x
// takes the flagged columns in its eighth of the rows — unit (row, candidate), candidates of a row back to back — and then
// the flagged rows in its eighth of the 64-column chunks — unit (candidate, flagged row, chunk) — so the taps that
// neighbouring rows, bands and candidates share meet in one L2.  gridDim.x must be a multiple of 8.
__global__ void __launch_bounds__(256) focus_exact(const KernelArgs a, const FocusWork w)
{
    const int lane = threadIdx.x & 63;
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t wave_id = __builtin_amdgcn_readfirstlane((blockIdx.x >> 3) * 4 + (threadIdx.x >> 6));
    const uint32_t n_waves = (gridDim.x >> 3) * 4;
    const int W = a.width, H = a.height;
    const focus_const_u32_ptr prefix = (focus_const_u32_ptr)(uintptr_t)w.prefix;
    const focus_const_int_ptr ncols = (focus_const_int_ptr)(uintptr_t)w.ncols;

    // flagged columns
    const uint32_t y0 = uint32_t(H) * xcd / 8u, y1 = uint32_t(H) * (xcd + 1u) / 8u;
    for(uint32_t u = wave_id; u < (y1 - y0) * FOCUS_STEPS; u += n_waves)
    {
        const int y = int(y0 + u / FOCUS_STEPS), i = int(u % FOCUS_STEPS);
        const int n = ncols[i];
        for(int idx = lane; idx - lane < n; idx += 64)
        {
            const bool active = idx < n;
            const int x = active ? w.cols[(size_t)i * W + idx] : 0;
            const uint32_t key = focus_exact_key(a, w, x, y, i);
            if(active)
                w.K[((size_t)i * H + y) * W + x] = static_cast<uint16_t>(key);
        }
    }
    // flagged rows
    const uint32_t chunks_w = uint32_t((W + 63) / 64);
    const uint32_t c0 = chunks_w * xcd / 8u, c1 = chunks_w * (xcd + 1u) / 8u, nc = c1 - c0;
    const uint32_t total_rows = prefix[FOCUS_STEPS];
    int i = 0;
    for(uint32_t u = wave_id; u < total_rows * nc; u += n_waves)
    {
        const uint32_t r = u / nc;
        while(r >= prefix[i + 1])
            i++;
        const int y = w.rows[(size_t)i * H + (r - prefix[i])];
        const int xx = int(c0 + u % nc) * 64 + lane;
        const bool active = xx < W;
        const int x = active ? xx : 0;
        const uint32_t key = focus_exact_key(a, w, x, y, i);
        if(active)
            w.K[((size_t)i * H + y) * W + x] = static_cast<uint16_t>(key);
    }
}

// dispersion per candidate = nine samples of E (or the exact key where flagged); first strict minimum → map 0.
// Blocks of 4 rows × 64·PPL pixels, a lane owns PPL ∈ {1, 2} adjacent pixels; PPL = 2 reads both pixels' samples with one
// dword load and needs an even radius_x (the E columns x + rx ± rx of an even x are then dword aligned) — the reference always
// produces one (src/interpolator.cu:143-146).  Vertical stripes per XCD (stripe_map) keep the ±r rows of all 32 candidates in
// one L2 (row bands per XCD for images narrower than 8 blocks).  Every sample is scalar plane base + per-lane tap offset.
template <int PPL>
__global__ void __launch_bounds__(256) focus_pick(const KernelArgs a, const FocusWork w, const uint32_t nblocks, const int striped)
{
    const int W = a.width, H = a.height;
    const uint32_t blocks_x = uint32_t(W + 64 * PPL - 1) / uint32_t(64 * PPL), blocks_y = uint32_t(H + 3) / 4u;
    uint32_t bx, by, unused;
    if(striped)
    {
        if(!stripe_map(blockIdx.x, blocks_x, blocks_y, 1u, bx, by, unused))
            return;
    }
    else
    {
        const uint32_t block = xcd_contiguous(blockIdx.x, nblocks);
        bx = block % blocks_x;
        by = block / blocks_x;
    }
    const int x = (int(bx) * 64 + int(threadIdx.x & 63)) * PPL;
    const int y = int(by) * 4 + int(threadIdx.x >> 6);
    if(y >= H) // wave-uniform
        return;
    const int rx = a.radius_x, ry = a.radius_y;
    // lanes past the right edge compute pixel 0 and store nothing; the second pixel of a lane at x = W − 1 (odd W) reads
    // one element past a row of badx / E / K, inside the workspace, and is not stored either
    const int xs = x < W ? x : 0;
    uint32_t flagged[PPL];
#pragma unroll
    for(int j = 0; j < PPL; j++)
        flagged[j] = w.badx[xs + j] | w.bady[y];
    bool any = false;
#pragma unroll
    for(int j = 0; j < PPL; j++)
        any = any || flagged[j] != 0u;
    const bool wave_flagged = __builtin_amdgcn_ballot_w64(any) != 0ull;

    uint32_t tap[9]; // byte offsets of the nine samples inside a candidate's plane of E
#pragma unroll
    for(int ty = 0; ty < 3; ty++)
#pragma unroll
        for(int tx = 0; tx < 3; tx++)
            tap[ty * 3 + tx] = uint32_t((y + ty * ry) * w.We_p + (xs + tx * rx)) * 2u;
    const size_t plane_bytes = (size_t)w.He_p * w.We_p * 2;
    const uint8_t *plane = reinterpret_cast<const uint8_t *>(w.E);
    const uint16_t *exact = w.K + (size_t)y * W + xs;

    uint32_t best_key[PPL];
    int best_i[PPL];
#pragma unroll
    for(int j = 0; j < PPL; j++)
    {
        best_key[j] = 0xffffffffu;
        best_i[j] = 0;
    }
    auto candidate = [&](const int i, const bool with_exact) {
        uint32_t sum[PPL];
        if constexpr(PPL == 2)
        {
            u16x2 acc = as_u16x2(0u); // 9 · 4081 < 65536 per half
#pragma unroll
            for(int t = 0; t < 9; t++)
                acc += as_u16x2(*reinterpret_cast<const uint32_t *>(plane + tap[t]));
            sum[0] = as_u32(acc) & 0xffffu;
            sum[1] = as_u32(acc) >> 16;
        }
        else
        {
            sum[0] = 0;
#pragma unroll
            for(int t = 0; t < 9; t++)
                sum[0] += *reinterpret_cast<const uint16_t *>(plane + tap[t]);
        }
#pragma unroll
        for(int j = 0; j < PPL; j++)
        {
            uint32_t key = sum[j] >= 16u ? (sum[j] & ~15u) : sum[j];
            if(with_exact)
            {
                const uint32_t k_exact = exact[j]; // unconditional: flagged lanes are rare, a divergent load costs more
                key = ((flagged[j] >> i) & 1u) ? k_exact : key;
            }
            if(key < best_key[j]) // MinDispersion::add (src/kernels.cu:225-231): strict <
            {
                best_key[j] = key;
                best_i[j] = i;
            }
        }
        plane += plane_bytes;
        exact += (size_t)H * W;
    };
    if(wave_flagged)
    {
#pragma unroll 2
        for(int i = 0; i < FOCUS_STEPS; i++)
            candidate(i, true);
    }
    else
    {
#pragma unroll 4
        for(int i = 0; i < FOCUS_STEPS; i++)
            candidate(i, false);
    }
#pragma unroll
    for(int j = 0; j < PPL; j++)
        if(x + j < W)
        {
            const float best_f = focus_candidate(a, best_i[j]);
            const float normalized = __fdiv_rn(best_f - a.focus, a.range);
            const uint32_t m = static_cast<uint32_t>(roundf(normalized * 255.0f)) & 0xffu;
            reinterpret_cast<uint32_t *>(a.maps)[(size_t)y * W + x + j] = m | (m << 8) | (m << 16) | 0xff000000u;
        }
}

} // namespace lfi
