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
// or a float rounding of fma(f, offset, x) across an integer) flags (x, i) / (y, i) — bands of ≈r columns / rows per view
// with a negative shift, ≈6 % of all pairs.  The coordinates are separable, so a pair flagged on ONE axis still factors
// along the other:
//   row y flagged, column not:  the tap rows are whatever the reference computes for (y, view), the tap columns are
//       uniform shifts → three lines E'(i, y, ty)(qx) (focus_lines_rows), nine samples of those (focus_line_keys → K);
//   column x flagged, row not:  three lines E''(i, x, tx)(qy) (focus_lines_cols), likewise;
//   both flagged: the taps where BOTH axes fail (usually one) tap by tap (focus_exact), the others from the lines / E as above;
//   more flagged rows / columns than the line buffers hold: all nine taps tap by tap, focus_exact → K.
// focus_pick takes K instead of the factored sum wherever a flag is set.
// The integer key (16·S + k, see focus_map.hpp) is the same as in the other variants, so results are bit-identical.
//
// Clamp-to-edge is taken out of the hot loops: focus_pad copies the ≤32 sampled images into planes padded by the largest
// shift + r on every side (edge pixels replicated), so every sample is an unclamped load at scalar base + lane offset.
//
// Flags are kept per tap (round 3): a line is computed only for the taps where the uniform shift fails — almost always the one tap
// at +r — and the other taps of a flagged row / column are read from E like everybody else's.
//
// Passes:  plan_shifts → plan_flags → plan_lists → plan_prefix, focus_pad → focus_range (E),
// focus_flagged = {focus_lines_rows (Er), focus_lines_cols (Ec), focus_exact (K)} → focus_line_keys (K; reads Er, Ec AND E) →
// focus_pick (map 0) → focus_filter (map 1).
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
    uint32_t *tapx;     // [3][W] bit i: tap tx = t − 1 of column x is where the uniform shift fails for candidate i (badx = the OR of the three)
    uint32_t *tapy;     // [3][H]
    uint16_t *cols;     // [32][W]  flagged columns of candidate i, ascending
    uint16_t *rows;     // [32][H]
    int32_t *ncols;     // [32]
    int32_t *nrows;     // [32]
    uint32_t *prefix;   // [3][33] prefix sums over the candidates of nrows, of ncols and of ceil(ncols/64) (column chunks)
    uint32_t *rowbase;  // [H]  line slot of (row y, candidate i) = rowbase[y] + popcount(bady[y] & ((1 << i) − 1)):
    uint32_t *colbase;  // [W]  exclusive prefix sums of popcount(bady / badx)
    uint16_t *Er;       // [R_cap][3][We_p]      E'(slot, ty)(qx): row lines of the flagged-row slots below R_cap
    uint16_t *Ec;       // [3][He_p][C_cap]      E''(tx)(qy)(slot): column lines of the flagged-column slots below C_cap
    int32_t R_cap, C_cap;
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

struct FocusPatch;
__device__ __forceinline__ void focus_plan_shift(const KernelArgs &a, const FocusWork &w);
template <int CPW>
__device__ __forceinline__ void focus_plan_patch(const KernelArgs &a, const FocusWork &w, FocusPatch *plans, int group, int k);

// one thread per (candidate, view slot); then (cpw = 8 or 4) the first 32 / cpw · 32 threads plan focus_range_t's patches per (candidate group, view slot)
__global__ void __launch_bounds__(1024) focus_plan_shifts(const KernelArgs a, const FocusWork w, FocusPatch *plans, const int cpw)
{
    focus_plan_shift(a, w);
    __syncthreads(); // (one workgroup: its global stores are visible to its own threads behind the barrier)
    const int i = threadIdx.x >> 5, k = threadIdx.x & 31;
    if(cpw == 8 && i < FOCUS_STEPS / 8)
        focus_plan_patch<8>(a, w, plans, i, k);
    else if(cpw == 4 && i < FOCUS_STEPS / 4)
        focus_plan_patch<4>(a, w, plans, i, k);
}

__device__ __forceinline__ void focus_plan_shift(const KernelArgs &a, const FocusWork &w)
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

// pad[k][yy][xx] = I_ids[k][clamp(yy − Py)][clamp(xx − Px)]; grid (ceil(Wp/256), ceil(Hp/FOCUS_PAD_ROWS), n_ids), a lane writes 4 pixels of
// FOCUS_PAD_ROWS consecutive padded rows (it needs every row's successor for the alpha byte: walking down, each source row is read once)
// k0: the first sampled image of the range this launch pads (gridDim.z of them: all, or the ones that changed)
constexpr int FOCUS_PAD_ROWS = 8;
__global__ void __launch_bounds__(64) focus_pad(const KernelArgs a, const FocusWork w, const int k0)
{
    const int xx = (blockIdx.x * 64 + threadIdx.x) * 4;
    if(xx >= w.Wp)
        return;
    const int yy0 = blockIdx.y * FOCUS_PAD_ROWS, k = k0 + blockIdx.z;
    const int W = a.width, H = a.height;
    const uint32_t *plane = reinterpret_cast<const uint32_t *>(a.grid) + (size_t)a.focus_ids[k] * ((size_t)W * H);
    const int x = xx - w.Px;
    const bool inside = x >= 0 && x + 3 <= W - 1;
    auto load_row = [&](const int yy) { // the four source pixels of padded row yy (clamp-to-edge both ways)
        const uint32_t *row = plane + (size_t)clampi(yy - w.Py, 0, H - 1) * W;
        u32x4 v;
        if(inside)
            v = *reinterpret_cast<const u32x4_a4 *>(row + x);
        else
        {
            v.x = row[clampi(x, 0, W - 1)];
            v.y = row[clampi(x + 1, 0, W - 1)];
            v.z = row[clampi(x + 2, 0, W - 1)];
            v.w = row[clampi(x + 3, 0, W - 1)];
        }
        return v;
    };
    u32x4 v = load_row(yy0);
#pragma unroll
    for(int r = 0; r < FOCUS_PAD_ROWS; r++)
    {
        const int yy = yy0 + r;
        if(yy >= w.Hp) // block-uniform
            break;
        const u32x4 b = load_row(yy + 1); // the padded row below
        // The alpha byte of a padded pixel carries the BLUE of the pixel one padded row below (no pass reads alpha: the reference's
        // ElementRange looks at R, G, B only, src/kernels.cu:173-194): a padded pixel then widens to focus_range_t's LDS slot
        // [R, G | B, B'] with two single-source v_perm_b32.
        u32x4 o;
        o.x = __builtin_amdgcn_perm(b.x, v.x, 0x06020100u);
        o.y = __builtin_amdgcn_perm(b.y, v.y, 0x06020100u);
        o.z = __builtin_amdgcn_perm(b.z, v.z, 0x06020100u);
        o.w = __builtin_amdgcn_perm(b.w, v.w, 0x06020100u);
        *reinterpret_cast<u32x4 *>(w.pad + ((size_t)k * w.Hp + yy) * w.Wp + xx) = o; // Wp is a multiple of 4
        v = b;
    }
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
    // per tap: almost every flag comes from truncation toward zero left of / above the image, which moves only the tap at +r (the
    // taps at −r and 0 clamp to the edge either way) — the line passes then compute one line per flagged row / column, not three
    bool bad[3] = {false, false, false};
    for(int k = 0; k < a.n_focus_ids; k++)
    {
        const int32_t *s = w.shifts + 4 * (i * FOCUS_MAX_IDS + k);
        const lfi_float2 off = a.offsets[s[2]];
        const int T = warp_float(c, f, axis ? off.y : off.x);
        const int U = c + s[axis];
#pragma unroll
        for(int t = -1; t <= 1; t++)
            bad[t + 1] = bad[t + 1] || (clampi(T + t * r, 0, L - 1) != clampi(U + t * r, 0, L - 1));
    }
    if(bad[0] || bad[1] || bad[2])
        atomicOr((axis ? w.bady : w.badx) + c, 1u << i);
#pragma unroll
    for(int t = 0; t < 3; t++)
        if(bad[t])
            atomicOr((axis ? w.tapy : w.tapx) + (size_t)t * L + c, 1u << i);
}

// line slot of a flagged (row or column c, candidate i) pair
__device__ __forceinline__ uint32_t line_slot(const uint32_t base_c, const uint32_t bad_c, const int i)
{
    return base_c + uint32_t(__builtin_popcount(bad_c & ((1u << i) - 1u)));
}

// grid (32 candidates, 2 axes), one wave each: ordered compaction of the flagged columns / rows; the wave of candidate 0
// also scans the per-coordinate flag counts into rowbase / colbase
__global__ void __launch_bounds__(64) focus_plan_lists(const KernelArgs a, const FocusWork w)
{
    const int i = blockIdx.x, axis = blockIdx.y, lane = threadIdx.x;
    const int L = axis ? a.height : a.width;
    const uint32_t *bad = axis ? w.bady : w.badx;
    uint16_t *list = (axis ? w.rows : w.cols) + (size_t)i * L;
    uint32_t *base = axis ? w.rowbase : w.colbase;
    int count = 0;
    uint32_t running = 0;
    for(int c0 = 0; c0 < L; c0 += 64)
    {
        const int c = c0 + lane;
        const uint32_t bits = c < L ? bad[c] : 0u;
        const bool flagged = (bits >> i) & 1u;
        const uint64_t m = __builtin_amdgcn_ballot_w64(flagged);
        if(flagged)
            list[count + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = static_cast<uint16_t>(c);
        count += __builtin_popcountll(m);
        if(i == 0)
        {
            uint32_t incl = uint32_t(__builtin_popcount(bits)); // inclusive scan over the wave
#pragma unroll
            for(int d = 1; d < 64; d <<= 1)
            {
                const uint32_t up = __shfl_up(incl, d, 64);
                incl += lane >= d ? up : 0u;
            }
            if(c < L)
                base[c] = running + incl - uint32_t(__builtin_popcount(bits));
            running += __shfl(incl, 63, 64);
        }
    }
    if(lane == 0)
        (axis ? w.nrows : w.ncols)[i] = count;
}

// one thread: the three prefix tables (the line and exact kernels walk all candidates' lists as one sequence)
__global__ void focus_plan_prefix(const KernelArgs a, const FocusWork w)
{
    uint32_t rows = 0, cols = 0, chunks = 0;
    for(int i = 0; i < FOCUS_STEPS; i++)
    {
        w.prefix[i] = rows;
        w.prefix[33 + i] = cols;
        w.prefix[66 + i] = chunks;
        rows += uint32_t(w.nrows[i]);
        cols += uint32_t(w.ncols[i]);
        chunks += uint32_t(w.ncols[i] + 63) / 64u;
    }
    w.prefix[32] = rows;
    w.prefix[65] = cols;
    w.prefix[98] = chunks;
}

// Two samples per instruction.  A u16 lane that holds a byte (0 … 255) is the bit pattern of a non-negative fp16 SUBNORMAL, and the
// float order of non-negative fp16 values is the integer order of their bit patterns, so gfx950's three-operand packed fp16
// minimum / maximum (v_pk_minimum3_f16 / v_pk_maximum3_f16: IEEE-754-2019 minimum / maximum, no NaN can occur) reduce (accumulator,
// sample of view k, sample of view k + 1) to the same bytes as two v_pk_min_u16 / v_pk_max_u16 would — IF the instructions leave
// subnormals alone: kernels are compiled with fp16 denormals enabled (.amdhsa_float_denorm_mode_16_64 3) and lfi_debug_pk_minmax3_f16
// checks all 256³ byte triples on the device (tests/test_gpu_parity.py::test_pk_minmax3_f16_on_bytes).  Round 2's range pass issued
// 12 packed min / max per four pixels and view (VALU-issue-bound, profiles/r02_pmc_focus_range_summary.txt); this issues 6.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 min3_bytes(const u16x2 acc, const u16x2 a, const u16x2 b)
{
    // written as a CHAIN ((acc ∧ a) ∧ b): the instruction selector folds a chain link by link into minimum3; written as acc ∧ (a ∧ b)
    // two consecutive reductions became minimum3(acc, a ∧ b, c ∧ d) plus two two-operand minima — three instructions per four samples
    const f16x2 m = __builtin_elementwise_minimum(__builtin_elementwise_minimum(__builtin_bit_cast(f16x2, acc), __builtin_bit_cast(f16x2, a)),
                                                  __builtin_bit_cast(f16x2, b));
    return __builtin_bit_cast(u16x2, m);
}
__device__ __forceinline__ u16x2 max3_bytes(const u16x2 acc, const u16x2 a, const u16x2 b)
{
    const f16x2 m = __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(f16x2, acc), __builtin_bit_cast(f16x2, a)),
                                                  __builtin_bit_cast(f16x2, b));
    return __builtin_bit_cast(u16x2, m);
}

// hardware probe behind lfi_debug_pk_minmax3_f16: every byte triple (a, b, c) — 2^24 threads, two triples each (one per u16 half) —
// through min3_bytes / max3_bytes against integer min / max; counts the halves that differ
__global__ void __launch_bounds__(256) probe_pk_minmax3(uint32_t *mismatches)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const uint32_t a0 = t & 255u, b0 = (t >> 8) & 255u, c0 = (t >> 16) & 255u;
    const uint32_t a1 = 255u - a0, b1 = b0 ^ 0x5au, c1 = (c0 * 7u + 3u) & 255u;
    const u16x2 A = as_u16x2(a0 | (a1 << 16)), B = as_u16x2(b0 | (b1 << 16)), C = as_u16x2(c0 | (c1 << 16));
    const uint32_t mn = as_u32(min3_bytes(C, A, B)), mx = as_u32(max3_bytes(C, A, B));
    const uint32_t want_mn = min(min(a0, b0), c0) | (min(min(a1, b1), c1) << 16), want_mx = max(max(a0, b0), c0) | (max(max(a1, b1), c1) << 16);
    uint32_t bad = 0;
    bad += (mn & 0xffffu) != (want_mn & 0xffffu);
    bad += (mn >> 16) != (want_mn >> 16);
    bad += (mx & 0xffffu) != (want_mx & 0xffffu);
    bad += (mx >> 16) != (want_mx >> 16);
    if(bad)
        atomicAdd(mismatches, bad);
}

// running per-channel min / max of FOUR consecutive pixels of a lane (two u16 pairs per channel) and the E encoding
struct RangeAcc4
{
    u16x2 lo[2][3], hi[2][3];
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for(int p = 0; p < 2; p++)
#pragma unroll
            for(int ch = 0; ch < 3; ch++)
            {
                lo[p][ch] = as_u16x2(0x00ff00ffu);
                hi[p][ch] = as_u16x2(0u);
            }
    }
    __device__ __forceinline__ void add(const u32x4 v)
    {
        const uint32_t px[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for(int p = 0; p < 2; p++)
        {
            const u16x2 cr = channel_pair<0>(px[2 * p], px[2 * p + 1]);
            const u16x2 cg = channel_pair<1>(px[2 * p], px[2 * p + 1]);
            const u16x2 cb = channel_pair<2>(px[2 * p], px[2 * p + 1]);
            lo[p][0] = __builtin_elementwise_min(lo[p][0], cr);
            hi[p][0] = __builtin_elementwise_max(hi[p][0], cr);
            lo[p][1] = __builtin_elementwise_min(lo[p][1], cg);
            hi[p][1] = __builtin_elementwise_max(hi[p][1], cg);
            lo[p][2] = __builtin_elementwise_min(lo[p][2], cb);
            hi[p][2] = __builtin_elementwise_max(hi[p][2], cb);
        }
    }
    // the samples of TWO views at once (min3 / max3)
    __device__ __forceinline__ void add2(const u32x4 va, const u32x4 vb)
    {
#ifdef LFI_AB_NO_MINMAX3 // measurement builds: the same loops with round 2's two-operand u16 reductions (A/B on one box)
        add(va);
        add(vb);
        return;
#endif
        const uint32_t pa[4] = {va.x, va.y, va.z, va.w}, pb[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
        for(int p = 0; p < 2; p++)
        {
            const u16x2 ar = channel_pair<0>(pa[2 * p], pa[2 * p + 1]), br = channel_pair<0>(pb[2 * p], pb[2 * p + 1]);
            const u16x2 ag = channel_pair<1>(pa[2 * p], pa[2 * p + 1]), bg = channel_pair<1>(pb[2 * p], pb[2 * p + 1]);
            const u16x2 ab = channel_pair<2>(pa[2 * p], pa[2 * p + 1]), bb = channel_pair<2>(pb[2 * p], pb[2 * p + 1]);
            lo[p][0] = min3_bytes(lo[p][0], ar, br);
            hi[p][0] = max3_bytes(hi[p][0], ar, br);
            lo[p][1] = min3_bytes(lo[p][1], ag, bg);
            hi[p][1] = max3_bytes(hi[p][1], ag, bg);
            lo[p][2] = min3_bytes(lo[p][2], ab, bb);
            hi[p][2] = max3_bytes(hi[p][2], ab, bb);
        }
    }
    // four u16: 16·range + (FLT_MIN tap ? 1 : 0), see focus_map.hpp
    __device__ __forceinline__ u32x2 encode() const
    {
        u32x2 out;
#pragma unroll
        for(int p = 0; p < 2; p++)
        {
            const u16x2 d0 = hi[p][0] - lo[p][0], d1 = hi[p][1] - lo[p][1], d2 = hi[p][2] - lo[p][2];
            const u16x2 dmax = __builtin_elementwise_max(__builtin_elementwise_max(d0, d1), d2);
            const u16x2 hmin = __builtin_elementwise_min(__builtin_elementwise_min(hi[p][0], hi[p][1]), hi[p][2]);
            // FLT_MIN tap: range 0 and an all-zero channel
            const u16x2 nz = __builtin_elementwise_min(as_u16x2(as_u32(dmax) | as_u32(hmin)), as_u16x2(0x00010001u));
            const uint32_t e = (as_u32(dmax) << 4) + (0x00010001u - as_u32(nz)); // per half: 16·range ≤ 4080, no carry
            if(p == 0)
                out.x = e;
            else
                out.y = e;
        }
        return out;
    }
};

// the same for ONE pixel per lane: (R, G) as a u16 pair, and B
struct RangeAcc1
{
    u16x2 lo_rg, hi_rg, lo_b, hi_b;
    __device__ __forceinline__ void init()
    {
        lo_rg = lo_b = as_u16x2(0x00ff00ffu);
        hi_rg = hi_b = as_u16x2(0u);
    }
    __device__ __forceinline__ void add(const uint32_t px)
    {
        const u16x2 rg = as_u16x2(__builtin_amdgcn_perm(0u, px, 0x0c010c00u));
        const u16x2 b = as_u16x2(__builtin_amdgcn_perm(0u, px, 0x0c020c02u));
        lo_rg = __builtin_elementwise_min(lo_rg, rg);
        hi_rg = __builtin_elementwise_max(hi_rg, rg);
        lo_b = __builtin_elementwise_min(lo_b, b);
        hi_b = __builtin_elementwise_max(hi_b, b);
    }
    __device__ __forceinline__ void add2(const uint32_t pa, const uint32_t pb) // two views at once (min3 / max3)
    {
#ifdef LFI_AB_NO_MINMAX3
        add(pa);
        add(pb);
        return;
#endif
        const u16x2 rga = as_u16x2(__builtin_amdgcn_perm(0u, pa, 0x0c010c00u)), rgb = as_u16x2(__builtin_amdgcn_perm(0u, pb, 0x0c010c00u));
        // B: the two halves of lo_b / hi_b are independent accumulators (`add` feeds both the same value, this feeds view a's B to the
        // low half and view b's to the high half); result() folds them
        const u16x2 bb = as_u16x2(__builtin_amdgcn_perm(pb, pa, 0x0c060c02u));
        lo_rg = min3_bytes(lo_rg, rga, rgb);
        hi_rg = max3_bytes(hi_rg, rga, rgb);
        lo_b = __builtin_elementwise_min(lo_b, bb);
        hi_b = __builtin_elementwise_max(hi_b, bb);
    }
    __device__ __forceinline__ void result(uint32_t &range, uint32_t &flt_min_tap) const
    {
        const uint32_t hb = as_u32(hi_b), lb = as_u32(lo_b);
        const uint32_t hi_bv = max(hb & 0xffffu, hb >> 16), lo_bv = min(lb & 0xffffu, lb >> 16);
        const uint32_t d_rg = as_u32(hi_rg - lo_rg), d_b = hi_bv - lo_bv;
        range = max(max(d_rg & 0xffffu, d_rg >> 16), d_b);
        const uint32_t h_rg = as_u32(hi_rg);
        const uint32_t hmin = min(min(h_rg & 0xffffu, h_rg >> 16), hi_bv);
        flt_min_tap = (range | hmin) == 0u ? 1u : 0u;
    }
};

// walk a 33-entry prefix table forward: segment of a non-decreasing sequence of element indices (uniform)
__device__ __forceinline__ void prefix_walk(const focus_const_u32_ptr prefix, const uint32_t e, int &seg)
{
    while(e >= prefix[seg + 1])
        seg++;
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

    RangeAcc4 acc[CPW];
#pragma unroll
    for(int c = 0; c < CPW; c++)
        acc[c].init();
    const int n_ids = a.n_focus_ids;
    // Software pipeline over the views (round 2: the loop used to load a view's four deltas through the scalar cache, wait, load
    // its four sample vectors, wait, and only then reduce — both latencies exposed per view): the samples of view k + 1 are in
    // flight while view k is reduced, and the deltas of view k + 2 are already on their way through the scalar cache.
    auto load_deltas = [&](const int k, int64_t (&d)[CPW]) {
#pragma unroll
        for(int c = 0; c < CPW; c++)
            d[c] = deltas[c * FOCUS_MAX_IDS + k];
    };
    auto load_samples = [&](const int64_t (&d)[CPW], u32x4 (&px)[CPW]) {
#pragma unroll
        for(int c = 0; c < CPW; c++)
            px[c] = *reinterpret_cast<const u32x4_a4 *>(wave_base + d[c] + lane_off);
    };
    // No branches in the loop: past the last view the indices clamp to it — reducing a view twice changes no minimum or maximum —
    // so the compiler's vmcnt / lgkmcnt counts are exact (with conditional loads it has to assume they were not issued and waits
    // for everything).
    // Round 3: two views per reduction (RangeAcc4::add2: min3 / max3), so the pipeline moves in PAIRS of views: the samples of pair
    // j + 1 are in flight while pair j is reduced, the deltas of pair j + 2 are on their way.
    int64_t dA[CPW], dB[CPW], dC[CPW], dD[CPW];
    u32x4 pA[CPW], pB[CPW], pC[CPW], pD[CPW];
    const int last = n_ids - 1;
    load_deltas(0, dA);
    load_deltas(min(1, last), dB);
    load_samples(dA, pA);
    load_samples(dB, pB);
    load_deltas(min(2, last), dC);
    load_deltas(min(3, last), dD);
    int k = 0;
    for(; k + 4 < n_ids; k += 4)
    {
        load_samples(dC, pC);                 // views k + 2, k + 3
        load_samples(dD, pD);
        load_deltas(min(k + 4, last), dA);
        load_deltas(min(k + 5, last), dB);
#pragma unroll
        for(int c = 0; c < CPW; c++)
            acc[c].add2(pA[c], pB[c]);        // views k, k + 1
        load_samples(dA, pA);                 // views k + 4, k + 5
        load_samples(dB, pB);
        load_deltas(min(k + 6, last), dC);
        load_deltas(min(k + 7, last), dD);
#pragma unroll
        for(int c = 0; c < CPW; c++)
            acc[c].add2(pC[c], pD[c]);        // views k + 2, k + 3
    }
    // tail: views k, k + 1 are loaded; k + 2, k + 3 (clamped to the last view) are not
    load_samples(dC, pC);
    load_samples(dD, pD);
#pragma unroll
    for(int c = 0; c < CPW; c++)
        acc[c].add2(pA[c], pB[c]);
#pragma unroll
    for(int c = 0; c < CPW; c++)
        acc[c].add2(pC[c], pD[c]);
#pragma unroll
    for(int c = 0; c < CPW; c++)
    {
        uint16_t *dst = w.E + ((size_t)(i0 + c) * w.He_p + ey) * w.We_p + tx * 256 + 4 * lane;
        *reinterpret_cast<u32x2 *>(dst) = acc[c].encode();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// focus_range_t (round 5): E_i(q) with the views' samples UNPACKED ONCE into LDS and reduced from there.
//
// focus_range spends half of its vector instructions widening bytes to u16 lanes (a v_perm_b32 per channel pair, per use) and runs at the
// L1's tag rate; round 4's LDS-staged variant removed the L1 and kept the v_perm's (VALU-paced, the same time).  What the unit probes of
// round 5 say (tools/probe_range_units.hip, profiles/r05_range_unit_probes.txt): v_pk_minimum3 / maximum3_f16 and v_perm_b32 issue every 5
// cycles; LDS reads need NATURAL alignment (a ds_read_b64 at a 4-byte boundary runs 20 times slower), so samples addressed per pixel want
// one 8-byte slot per pixel.
//
// So: focus_pad stores the BLUE of the pixel one row below in the (unused) alpha byte, and a padded pixel widens to the slot
// [R, G | B, B'] of four u16 with two v_perm_b32 — ONCE per staged pixel, by waves that do nothing else.  A reducing lane owns one column
// and TWO row pairs of the tile; for the pair (y, y + 1) it reads slot(y) and slot(y + 1) and uses the three registers (R0,G0), (R1,G1),
// (B0,B1) — six bytes in u16 lanes, no instruction spent on widening — and reduces two views per v_pk_minimum3 / maximum3_f16: 1.5 vector
// instructions per (pixel, candidate, view) where focus_range issues 3.  A workgroup = 64 columns × 32 rows of CPW consecutive candidates;
// per step it holds the patches of TWO views (each the tile plus the span of the group's shifts: at most 96 × 50 pixels) and every staged
// pixel serves CPW candidates.  Host-checked preconditions (launch_focus_factored; else focus_range): within a group a view's shifts span
// at most 32 pixels and 18 rows.  What was tried on the way (profiles/r05_notes.md): typed loads that unpack for free
// (buffer_load_format_d16_xyzw, format 8_8_8_8 UINT: exact from any byte address, but 16 cycles per wave-load of ONE pixel per lane), every
// wave loading and reducing in turn (the phases added up), one workgroup per tile × group (a workgroup that owns the LDS cannot overlap
// its successor's start-up: 2.5 → 1.8 ms when made persistent).
#ifndef FRT_RPW
#define FRT_RPW 4 // rows of the tile per reducing wave: 4 (eight reducing waves, tile 64 × 32) or 2 (twelve, tile 64 × 24)
#endif
#ifndef FRT_DEPTH
#define FRT_DEPTH 2 // units (four LDS reads each) a reducing wave keeps in flight
#endif
constexpr int FRT_TW = 64;                          // tile: extended columns (one per lane); rows: FRT_RPW per reducing wave
#ifndef FRT_LWAVES
#define FRT_LWAVES 4 // loading waves: 4 (beside eight reducing waves, tile 64 × 32) or 6 (beside six, tile 64 × 24)
#endif
constexpr int FRT_NW = FRT_RPW == 4 ? 12 - FRT_LWAVES : 12, FRT_TH = FRT_RPW * FRT_NW; // reducing waves per workgroup, tile rows
constexpr int FRT_MAX_DX = 32, FRT_MAX_DY = 18;     // the largest span of a view's shifts within a candidate group
constexpr int FRT_PW = FRT_TW + FRT_MAX_DX;         // a view's patch in LDS: pixels per row (the row pitch) …
constexpr int FRT_PR = FRT_TH + FRT_MAX_DY;         // … × rows, at most
constexpr int FRT_PR_LDS = FRT_PR + 2;              // rows of slots per view in LDS: the patches are fetched in blocks of four rows
constexpr int FRT_VIEW_B = FRT_PW * FRT_PR_LDS * 8; // 39,936 bytes of 8-byte slots; two views per step, two steps in LDS: 159,744 bytes

// per (candidate group, view): where the view's patch starts in the padded planes, its size, and where each candidate of the group reads
struct FocusPatch
{
    uint32_t src;      // byte offset in the padded planes of the patch's first pixel for the tile at extended (0, 0)
    uint32_t pw_pr;    // patch width (pixels) | rows << 16
    uint32_t d[4];     // per candidate c: byte offset of its samples inside the patch, (dy·FRT_PW + dx)·8, u16 pairs (c even: low half)
    uint32_t unused[2];
};

// one thread per (group, view slot); behind focus_plan_shifts' barrier
template <int CPW>
__device__ __forceinline__ void focus_plan_patch(const KernelArgs &a, const FocusWork &w, FocusPatch *plans, const int group, const int k)
{
    if(k >= a.n_focus_ids)
        return;
    int sx[CPW], sy[CPW], ox = INT32_MAX, oy = INT32_MAX, mx = INT32_MIN, my = INT32_MIN;
#pragma unroll
    for(int c = 0; c < CPW; c++)
    {
        sx[c] = w.shifts[4 * ((group * CPW + c) * FOCUS_MAX_IDS + k)];
        sy[c] = w.shifts[4 * ((group * CPW + c) * FOCUS_MAX_IDS + k) + 1];
        ox = min(ox, sx[c]), oy = min(oy, sy[c]), mx = max(mx, sx[c]), my = max(my, sy[c]);
    }
    FocusPatch p{};
    p.src = uint32_t((((int64_t)k * w.Hp + oy + w.Py) * w.Wp + ox + w.Px) * 4);
    p.pw_pr = uint32_t(FRT_TW + mx - ox) | (uint32_t(FRT_TH + my - oy) << 16);
#pragma unroll
    for(int c = 0; c < CPW; c++)
        p.d[c >> 1] |= uint32_t(((sy[c] - oy) * FRT_PW + (sx[c] - ox)) * 8) << (16 * (c & 1));
    plans[group * FOCUS_MAX_IDS + k] = p;
}

// One workgroup per CU (the patches of two steps fill the LDS): EIGHT REDUCING waves and FOUR LOADING waves.  Issued by the reducing waves
// themselves — in a burst in front of the barrier, or between the candidates — the loads held every wave while nothing was reduced (the
// three phases added up: profiles/r05_notes.md).  So the roles are split: the loading waves fetch step s + 1's patches, widen them and store
// its slots into the other half of the LDS while the reducing waves work on step s; ONE barrier per step.  Loading wave L: view L & 1 of the
// step's pair, the row blocks ≡ L >> 1 (mod 2) of its patch (details at the loading branch below).
constexpr int FRT_LW = FRT_LWAVES; // loading waves: view L & 1 of a step's pair, the row blocks ≡ L >> 1 (mod FRT_LW / 2) of its patch
// measurement builds (-DFRT_TRACE=1, tools/range_trace.sh): clocks per wave and category, read back by lfi_debug_frt_trace
#ifndef FRT_TRACE
#define FRT_TRACE 0
#endif
#if FRT_TRACE
__device__ unsigned long long lfi_frt_trace_buf[256 * 16 * 8];
#define FRT_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define FRT_T(var)
#endif
// At most 136 registers per lane (tests/test_abi_library.py checks the code object): three waves per SIMD then leave room for ONE wave of
// focus_flagged (100 registers, no LDS) on every SIMD — the flagged passes run BESIDE this kernel on the side stream, as they ran beside focus_range; with
// 3 × 168 registers taken they queued behind it and the focus map took as long as before (profiles/r05_notes.md).
template <int CPW>
__global__ void __launch_bounds__(64 * (FRT_NW + FRT_LW), 1) focus_range_t(const KernelArgs a, const FocusWork w, const FocusPatch *plans, const uint32_t pad_bytes,
                                                                           const uint32_t nblocks, const int striped)
{
    constexpr int GROUPS = FOCUS_STEPS / CPW;
    __shared__ __attribute__((aligned(16))) uint8_t lds[4 * FRT_VIEW_B];
    const uint32_t tiles_x = uint32_t(w.We_p) / FRT_TW, tiles_y = (uint32_t(w.He_p) + FRT_TH - 1) / FRT_TH;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_ids = a.n_focus_ids;
    // PERSISTENT: one workgroup per CU walks the work items (tile × candidate group) b, b + gridDim.x, … (gridDim.x a multiple of 8: a
    // workgroup stays on its XCD's items) — a workgroup that owns the whole LDS cannot overlap its successor's start-up with its own tail.
    // The LDS halves alternate per STEP across items; the barrier protocol needs nothing at an item's end (see below).
    auto work_item = [&](const uint32_t b, uint32_t &tx, uint32_t &ty, uint32_t &group) {
        if(striped)
            return stripe_map(b, tiles_x, tiles_y, GROUPS, tx, ty, group);
        const uint32_t work = xcd_contiguous(b, nblocks);
        group = work % GROUPS;
        tx = (work / GROUPS) % tiles_x;
        ty = (work / GROUPS) / tiles_x;
        return true;
    };
    uint32_t half = 0; // which half of the LDS the current step's slots are in
    if(wave >= FRT_NW)
    {
        // ---- a loading wave: view lw & 1 of every step's pair; the row blocks ≡ lw >> 1 (mod 2) of its patch.  Loads of FOUR pixels per
        // lane (buffer_load_dwordx4: the texture path takes 16 cycles per wave-load whatever its width — typed one-pixel loads, which
        // unpack for free, made it the bottleneck: profiles/r05_notes.md), widened here with two v_perm_b32 per pixel — these waves have
        // nothing else to compute — and stored as two ds_write_b128 per lane (8-byte stores from one wave per SIMD run at a third of
        // their rate).  "main": 64 columns × blocks of 4 rows (lane & 15 = quad of pixels, lane >> 4 = row); "extra": the 32 columns behind
        // them × blocks of 8 rows (lane & 7 = quad, lane >> 3 = row; all 32 columns whatever the patch's width — columns beyond it land
        // in slots nobody reads, as do rows beyond its height: FRT_PR_LDS rows of slots exist).
        // (tools/range_trace.py, clocks per step at 4K: a loading wave spends 500-570 issuing its fetch, 220-240 waiting for the loads, 1,450-1,520
        // widening and storing and 570 at the barrier; a reducing wave 1,380 reducing, 240 on its share of the epilogue and 1,320 at the barrier:
        // the loading side is the step's critical path, and its stores queue behind the reducing waves' reads in the LDS.  Raising the loading
        // waves' priority (s_setprio 3) changes nothing: 1,630 against 1,610 µs.)
        constexpr int NP = FRT_LW / 2; // loading waves per view
        const int lw = wave - FRT_NW, v = lw & 1, par = lw >> 1;
        const uint32_t row_b = uint32_t(w.Wp) * 4u;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(w.pad, 0, int(pad_bytes), 0x00020000);
        const int v_main = (lane >> 4) * int(row_b) + (lane & 15) * 16;
        constexpr int MAIN_N = ((FRT_PR + 3) / 4 + NP - 1) / NP, EXTRA_N = ((FRT_PR + 7) / 8 + NP - 1) / NP; // row blocks of one wave
        constexpr int MAIN_SURE = FRT_TH / (4 * NP);                                                        // … that every patch has
        uint8_t *const st_main = lds + v * FRT_VIEW_B + ((4 * par + (lane >> 4)) * FRT_PW + 4 * (lane & 15)) * 8;
        // The "extra" part is as wide as the patch needs it — 2, 4 or 8 quads of pixels, i.e. blocks of 32, 16 or 8 rows per wave-load (most
        // views' shifts span few columns: a third of all stored slots lay beyond their patches with a fixed width of 8 quads)
        struct Step // one step of one work item, as the loading wave sees it
        {
            uint32_t b;        // work item
            int k;             // first view of the step's pair
            uint32_t tile_off; // byte offset of the tile's first extended pixel relative to extended (0, 0) (modular)
            focus_const_u32_ptr plan_words;
            bool valid;
        };
        auto first_step_of = [&](uint32_t b) {
            Step st{b, 0, 0u, nullptr, false};
            for(; st.b < nblocks; st.b += gridDim.x)
            {
                uint32_t tx, ty, group;
                if(!work_item(st.b, tx, ty, group))
                    continue;
                st.tile_off = uint32_t((int(ty) * FRT_TH - a.radius_y) * w.Wp + (int(tx) * FRT_TW - a.radius_x)) * 4u;
                st.plan_words = (focus_const_u32_ptr)(uintptr_t)(plans + group * FOCUS_MAX_IDS);
                st.valid = true;
                break;
            }
            return st;
        };
        auto next_step = [&](const Step &st) {
            if(st.k + 2 < n_ids)
            {
                Step nx = st;
                nx.k += 2;
                return nx;
            }
            return first_step_of(st.b + gridDim.x);
        };
        struct Geo // what a fetched step's stores need to know
        {
            uint32_t pr; // rows of the patch
            int sh;      // extra part: log2 of its quads per row (1, 2, 3), 0: no extra columns
        };
        // Two register sets: the loads of step s + 1 — of the NEXT work item behind an item's last step — are issued BEFORE the slots of step
        // s are stored, so the texture path never idles while a wave waits for its last load and stores.
        u32x4 mA[MAIN_N], eA[EXTRA_N], mB[MAIN_N], eB[EXTRA_N];
        Geo gA{0, 0}, gB{0, 0};
#if FRT_TRACE
        unsigned long long tr_fetch = 0, tr_wait = 0, tr_store = 0, tr_bar = 0, tr_steps = 0;
        int n_fetched = 0;
#endif
        auto fetch = [&](const Step &st, u32x4 (&m)[MAIN_N], u32x4 (&e)[EXTRA_N], Geo &g) {
            FRT_T(t0);
            const int kv = min(st.k + v, n_ids - 1); // an odd tail reduces the last view twice: no minimum or maximum changes
            const focus_const_u32_ptr pl = st.plan_words + kv * 8;
            const uint32_t src = pl[0] + st.tile_off;
            g.pr = pl[1] >> 16;
            const uint32_t ew = (pl[1] & 0xffffu) - FRT_TW;
            g.sh = ew == 0 ? 0 : ew <= 8 ? 1 : ew <= 16 ? 2 : 3;
#if defined(FRT_ABL) && (FRT_ABL & 1) // measurement builds: no loads (the slots get whatever the registers hold)
#pragma unroll
            for(int i = 0; i < MAIN_N; i++)
                m[i] = u32x4{src + i, g.pr, src, g.pr};
#pragma unroll
            for(int i = 0; i < EXTRA_N; i++)
                e[i] = u32x4{src, g.pr + i, g.pr, src};
            return;
#endif
#if FRT_TRACE
            n_fetched = 0;
#endif
#pragma unroll
            for(int i = 0; i < MAIN_N; i++)
                if(i < MAIN_SURE || uint32_t(4 * (NP * i + par)) < g.pr) // wave-uniform: the block's first row
                {
                    m[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, v_main, int(src + uint32_t(4 * (NP * i + par)) * row_b), 0);
#if FRT_TRACE
                    n_fetched++;
#endif
                }
            if(g.sh)
            {
                const int rows = 64 >> g.sh; // rows per block
                const int v_extra = (lane >> g.sh) * int(row_b) + 4 * FRT_TW + (lane & ((1 << g.sh) - 1)) * 16;
#pragma unroll
                for(int i = 0; i < EXTRA_N; i++)
                    if(uint32_t(rows * (NP * i + par)) < g.pr)
                    {
                        e[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, v_extra, int(src + uint32_t(rows * (NP * i + par)) * row_b), 0);
#if FRT_TRACE
                        n_fetched++;
#endif
                    }
            }
#if FRT_TRACE
            asm volatile("" ::: "memory");
            FRT_T(t1);
            tr_fetch += t1 - t0;
#endif
        };
        // A lane's four slots are 32 contiguous bytes, stored by two ds_write_b128.  The 8 lanes a 16-byte store serves together would hit every
        // bank twice if all of them stored their lower half first (23 % of all LDS cycles were conflicts), so the lanes with bit 2 set store their
        // UPPER half first.  Which pixels a lane's first store holds is decided inside the v_perm_b32 that widens them anyway (a per-lane selector
        // over the pair (pixel 0 | pixel 2), (pixel 1 | pixel 3)): no extra instruction.  (Done with selects it cost four v_cndmask per lane-load
        // and was slower than the conflicts, 1.54 → 1.60 ms.  Without them the conflicts are gone — LDS busy 58 % → 45 % of the run time — and
        // the kernel is 1 % faster: it is bound by neither unit, profiles/r05_notes.md.)
        const bool upper_first = (lane & 4) != 0;
        const uint32_t sel_rg = upper_first ? 0x0c050c04u : 0x0c010c00u, sel_bb = upper_first ? 0x0c070c06u : 0x0c030c02u;
        const uint32_t first_off = upper_first ? 16u : 0u;
        auto put = [&](uint8_t *const dst, const u32x4 px) { // four padded pixels [R,G,B,B'] → four slots [R,G | B,B'] of u16
            u32x4 s0, s1;
            s0.x = __builtin_amdgcn_perm(px.z, px.x, sel_rg), s0.y = __builtin_amdgcn_perm(px.z, px.x, sel_bb);
            s0.z = __builtin_amdgcn_perm(px.w, px.y, sel_rg), s0.w = __builtin_amdgcn_perm(px.w, px.y, sel_bb);
            s1.x = __builtin_amdgcn_perm(px.x, px.z, sel_rg), s1.y = __builtin_amdgcn_perm(px.x, px.z, sel_bb);
            s1.z = __builtin_amdgcn_perm(px.y, px.w, sel_rg), s1.w = __builtin_amdgcn_perm(px.y, px.w, sel_bb);
            *reinterpret_cast<u32x4 *>(dst + first_off) = s0;
            *reinterpret_cast<u32x4 *>(dst + (first_off ^ 16u)) = s1;
        };
        auto store = [&](const u32x4 (&m)[MAIN_N], const u32x4 (&e)[EXTRA_N], const Geo &g, const int younger) {
            // (every wave has passed the previous barrier: the reducing waves are done with this half, which held the step before the one they reduce now)
#if FRT_TRACE
            {
                FRT_T(t0);
                // wait for THIS set's loads (the other set's `younger` loads were issued after them)
                switch(younger)
                {
                    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
                    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
                    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
                    default: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
                }
                FRT_T(t1);
                tr_wait += t1 - t0;
            }
            FRT_T(ts0);
#else
            (void)younger;
#endif
#pragma unroll
            for(int i = 0; i < MAIN_N; i++)
                if(i < MAIN_SURE || uint32_t(4 * (NP * i + par)) < g.pr)
                    put(st_main + half + 4 * NP * i * FRT_PW * 8, m[i]);
            if(g.sh)
            {
                const int rows = 64 >> g.sh;
                const int row0 = rows * par + (lane >> g.sh); // this lane's row in the wave's first block
                uint8_t *const st_extra = lds + v * FRT_VIEW_B + half + (row0 * FRT_PW + FRT_TW + 4 * (lane & ((1 << g.sh) - 1))) * 8;
#pragma unroll
                for(int i = 0; i < EXTRA_N; i++)
                    if(uint32_t(rows * (NP * i + par)) < g.pr && row0 + NP * rows * i < FRT_PR_LDS) // (a block may reach below the rows of slots)
                        put(st_extra + NP * rows * i * FRT_PW * 8, e[i]);
            }
#if FRT_TRACE
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            FRT_T(ts1);
            tr_store += ts1 - ts0;
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this wave's slots are in LDS
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#if FRT_TRACE
            FRT_T(ts2);
            tr_bar += ts2 - ts1;
            tr_steps++;
#endif
            half ^= 2u * FRT_VIEW_B;
        };
        Step sA = first_step_of(blockIdx.x);
        if(!sA.valid)
            return;
        fetch(sA, mA, eA, gA);
        int young = 0; // (measurement builds) loads issued behind the set about to be stored
        for(;;)
        {
            const Step sB = next_step(sA);
            young = 0;
            if(sB.valid)
            {
                fetch(sB, mB, eB, gB);
#if FRT_TRACE
                young = n_fetched;
#endif
            }
            store(mA, eA, gA, young);
            if(!sB.valid)
                break;
            sA = next_step(sB);
            young = 0;
            if(sA.valid)
            {
                fetch(sA, mA, eA, gA);
#if FRT_TRACE
                young = n_fetched;
#endif
            }
            store(mB, eB, gB, young);
            if(!sA.valid)
                break;
        }
#if FRT_TRACE
        if(lane == 0 && blockIdx.x < 256)
        {
            unsigned long long *o = lfi_frt_trace_buf + (blockIdx.x * 16 + wave) * 8;
            o[0] = tr_fetch, o[1] = tr_wait, o[2] = tr_store, o[3] = tr_bar, o[4] = tr_steps;
        }
#endif
        return;
    }
    // ---- a reducing wave: rows 4·wave … 4·wave + 3 of the tile, column lane
    // LDS address of this lane's column in the first of its four rows
    constexpr int PAIRS = FRT_RPW / 2; // row pairs per reducing lane
    const uint32_t rd_addr = uint32_t(uintptr_t((__attribute__((address_space(3))) void *)lds)) + uint32_t(FRT_RPW * wave * FRT_PW + lane) * 8u;
#if FRT_TRACE
    unsigned long long rr_bar = 0, rr_red = 0, rr_epi = 0, rr_steps = 0;
    const unsigned long long rr_begin = __builtin_amdgcn_s_memtime();
#endif
    for(uint32_t b = blockIdx.x; b < nblocks; b += gridDim.x)
    {
    uint32_t tx, ty, group;
    if(!work_item(b, tx, ty, group))
        continue;
    const focus_const_u32_ptr plan_words = (focus_const_u32_ptr)(uintptr_t)(plans + group * FOCUS_MAX_IDS);
    // accumulators per candidate: [R,G] of the four rows, [B,B'] of the two row pairs — running minima and maxima
    u16x2 lo[CPW][3 * PAIRS], hi[CPW][3 * PAIRS];
#pragma unroll
    for(int c = 0; c < CPW; c++)
#pragma unroll
        for(int r = 0; r < 3 * PAIRS; r++)
        {
            lo[c][r] = as_u16x2(0x00ff00ffu);
            hi[c][r] = as_u16x2(0u);
        }
    for(int k = 0; k < n_ids; k += 2)
    {
        uint32_t dc[2][4];
#pragma unroll
        for(int v = 0; v < 2; v++)
        {
            const focus_const_u32_ptr pl = plan_words + min(k + v, n_ids - 1) * 8;
#pragma unroll
            for(int q = 0; q < 4; q++)
                dc[v][q] = pl[2 + q];
        }
        FRT_T(tb0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (the plan words; the previous step's reads were consumed)
        __builtin_amdgcn_s_barrier();                       // step k's slots are in LDS
        asm volatile("" ::: "memory");
        FRT_T(tb1);
        // The reduction.  LDS reads by inline assembly: every slot is read whole (ds_read_b64) although of the second row of a pair only
        // [R,G] is used — the compiler narrows such a read to ds_read_b32, and 4-byte reads at an 8-byte lane stride run into two-way bank
        // conflicts (25 % of all LDS cycles, profiles/r05_pmc_range_t_first.txt).  Unit of the pipeline = one ROW PAIR of one candidate: four
        // reads (two slots × two views), six reductions; the reads of unit u + 1 are issued before unit u is reduced — two sets of 8 registers,
        // where whole candidates in flight would take two sets of 16 and push the kernel over its register budget.  LDS returns in order,
        // so lgkmcnt(4) = "the older four reads have landed" (a scalar load still in flight — the next step's plan words — only makes the
        // wait longer: it shares the counter).
        auto slot_addr = [&](const int c, const int v) {
            return rd_addr + (((dc[v][c >> 1] >> (16 * (c & 1))) & 0xffffu) + half + uint32_t(v) * FRT_VIEW_B);
        };
        auto read4 = [&](const uint32_t pa, const uint32_t pb, const int pair, u32x2 (&q)[4]) {
            if(pair == 0)
            {
                asm volatile("ds_read_b64 %0, %1" : "=v"(q[0]) : "v"(pa));
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(q[1]) : "v"(pa), "n"(FRT_PW * 8));
                asm volatile("ds_read_b64 %0, %1" : "=v"(q[2]) : "v"(pb));
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(q[3]) : "v"(pb), "n"(FRT_PW * 8));
            }
            else
            {
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(q[0]) : "v"(pa), "n"(2 * FRT_PW * 8));
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(q[1]) : "v"(pa), "n"(3 * FRT_PW * 8));
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(q[2]) : "v"(pb), "n"(2 * FRT_PW * 8));
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(q[3]) : "v"(pb), "n"(3 * FRT_PW * 8));
            }
        };
        auto reduce = [&](const int c, const int pair, u32x2 (&q)[4], const int units_in_flight) {
            if(units_in_flight >= 3)
                asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
            else if(units_in_flight == 2)
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
            else if(units_in_flight == 1)
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
            // accumulators of the pair: [R,G] of its two rows, [B,B']
            const int r0 = 2 * pair, r1 = 2 * pair + 1, rb = 2 * PAIRS + pair;
            lo[c][r0] = min3_bytes(lo[c][r0], as_u16x2(q[0].x), as_u16x2(q[2].x));
            hi[c][r0] = max3_bytes(hi[c][r0], as_u16x2(q[0].x), as_u16x2(q[2].x));
            lo[c][r1] = min3_bytes(lo[c][r1], as_u16x2(q[1].x), as_u16x2(q[3].x));
            hi[c][r1] = max3_bytes(hi[c][r1], as_u16x2(q[1].x), as_u16x2(q[3].x));
            lo[c][rb] = min3_bytes(lo[c][rb], as_u16x2(q[0].y), as_u16x2(q[2].y));
            hi[c][rb] = max3_bytes(hi[c][rb], as_u16x2(q[0].y), as_u16x2(q[2].y));
            // (pins this unit's reduction in front of the reads after the next: left to the scheduler it sinks below them and the reads
            // take more registers)
            asm volatile("" : "+v"(lo[c][r0]), "+v"(lo[c][r1]), "+v"(lo[c][rb]), "+v"(hi[c][r0]), "+v"(hi[c][r1]), "+v"(hi[c][rb]));
        };
#if defined(FRT_ABL) && (FRT_ABL & 2) // measurement builds: no reduction (barriers only)
        half ^= 2u * FRT_VIEW_B;
        continue;
#endif
        constexpr int UNITS = PAIRS * CPW, D = FRT_DEPTH;
        static_assert(D >= 2 && D <= 4 && D <= UNITS, "two to four units in flight (lgkmcnt counts to 15)");
        u32x2 q[D][4];
        uint32_t pa = 0u, pb = 0u;
        auto issue_unit = [&](const int u) { // the reads of unit u = (candidate u / PAIRS, row pair u % PAIRS)
            if(u % PAIRS == 0) // the unit starts a candidate
                pa = slot_addr(u / PAIRS, 0), pb = slot_addr(u / PAIRS, 1);
            read4(pa, pb, u % PAIRS, q[u % D]);
        };
#pragma unroll
        for(int u = 0; u < D - 1; u++)
            issue_unit(u);
#pragma unroll
        for(int u = 0; u < UNITS; u++)
        {
            if(u + D - 1 < UNITS)
                issue_unit(u + D - 1);
            reduce(u / PAIRS, u % PAIRS, q[u % D], (UNITS - 1 - u) < (D - 1) ? (UNITS - 1 - u) : (D - 1));
        }
#if FRT_TRACE
        asm volatile("" ::: "memory");
        FRT_T(tb2);
        rr_bar += tb1 - tb0, rr_red += tb2 - tb1, rr_steps++;
#endif
        half ^= 2u * FRT_VIEW_B;
    }
    // E: 16·range + (FLT_MIN tap ? 1 : 0) per pixel (focus_map.hpp); a lane holds rows 4·wave … 4·wave + 3 of its column
    FRT_T(te0);
    const int ey0 = int(ty) * FRT_TH + FRT_RPW * wave;
#pragma unroll
    for(int c = 0; c < CPW; c++)
#pragma unroll
        for(int u = 0; u < PAIRS; u++)
        {
            const u16x2 d0 = hi[c][2 * u] - lo[c][2 * u], d1 = hi[c][2 * u + 1] - lo[c][2 * u + 1], db = hi[c][2 * PAIRS + u] - lo[c][2 * PAIRS + u];
            // (row 0, row 1) pairs per channel: low halves = R, high halves = G of the two rows
            const u16x2 dr = as_u16x2(__builtin_amdgcn_perm(as_u32(d1), as_u32(d0), 0x05040100u)), dg = as_u16x2(__builtin_amdgcn_perm(as_u32(d1), as_u32(d0), 0x07060302u));
            const u16x2 dmax = __builtin_elementwise_max(__builtin_elementwise_max(dr, dg), db);
            const u16x2 hr = as_u16x2(__builtin_amdgcn_perm(as_u32(hi[c][2 * u + 1]), as_u32(hi[c][2 * u]), 0x05040100u));
            const u16x2 hg = as_u16x2(__builtin_amdgcn_perm(as_u32(hi[c][2 * u + 1]), as_u32(hi[c][2 * u]), 0x07060302u));
            const u16x2 hmin = __builtin_elementwise_min(__builtin_elementwise_min(hr, hg), hi[c][2 * PAIRS + u]);
            const u16x2 nz = __builtin_elementwise_min(as_u16x2(as_u32(dmax) | as_u32(hmin)), as_u16x2(0x00010001u));
            const uint32_t enc = (as_u32(dmax) << 4) + (0x00010001u - as_u32(nz)); // per half: 16·range ≤ 4080, no carry
            const int ey = ey0 + 2 * u;
            // One dword store per lane instead of two 16-bit ones: neighbouring lanes swap — the even lane stores both columns' values of
            // row ey, the odd lane both of row ey + 1 (ey is even and He_p a multiple of 4: both rows exist or neither).
            const uint32_t other = uint32_t(__builtin_amdgcn_mov_dpp(int(enc), 0xB1, 0xf, 0xf, false)); // quad_perm [1, 0, 3, 2]
            const uint32_t both = __builtin_amdgcn_perm(other, enc, (lane & 1) ? 0x03020706u : 0x05040100u);
            uint32_t *dst = reinterpret_cast<uint32_t *>(w.E + ((size_t)(group * CPW + c) * w.He_p + ey) * w.We_p + (tx * FRT_TW + (lane & ~1)) + ((lane & 1) ? w.We_p : 0));
            if(ey < w.He_p) // wave-uniform: the last tile row of an image whose extended height is not a multiple of the tile's
                *dst = both;
        }
#if FRT_TRACE
    asm volatile("" ::: "memory");
    FRT_T(te1);
    rr_epi += te1 - te0;
#endif
    } // work items
#if FRT_TRACE
    if(lane == 0 && blockIdx.x < 256)
    {
        unsigned long long *o = lfi_frt_trace_buf + (blockIdx.x * 16 + wave) * 8;
        o[0] = rr_bar, o[1] = rr_red, o[2] = rr_epi, o[3] = __builtin_amdgcn_s_memtime() - rr_begin, o[4] = rr_steps;
    }
#endif
}

// 64-bit readlane (lane index uniform)
__device__ __forceinline__ uint64_t readlane64(const uint64_t v, const int lane)
{
    const uint32_t lo = __builtin_amdgcn_readlane(uint32_t(v), lane), hi = __builtin_amdgcn_readlane(uint32_t(v >> 32), lane);
    return (uint64_t(hi) << 32) | lo;
}

// E'(slot, ty)(qx) for the flagged rows: like focus_range, but the source row of view k is the one the reference's
// arithmetic gives for (row y, view k) plus (ty − 1)·ry.  Persistent.  XCD x (blocks b ≡ x mod 8) takes the 256-column tiles
// ≡ x mod 8; its waves stride over the flagged (row, candidate) entries × those tiles — neighbouring rows read almost the same
// source lines, which therefore meet in one L2.  Per-view parameters live in lane k's registers (no scalar loads in the view loop); the samples of view k + 1
// are in flight while view k is reduced.
__device__ __forceinline__ void focus_lines_rows(const KernelArgs &a, const FocusWork &w, const uint32_t block, const uint32_t blocks)
{
    const int lane = threadIdx.x & 63;
    const uint32_t xcd = block & 7u;
    const uint32_t wave_id = __builtin_amdgcn_readfirstlane((block >> 3) * 4 + (threadIdx.x >> 6));
    const uint32_t n_waves = (blocks >> 3) * 4; // blocks is a multiple of 8
    const uint32_t tiles_x = uint32_t(w.We_p) >> 8;
    const uint32_t my_tiles = tiles_x > xcd ? (tiles_x - xcd + 7u) / 8u : 0u;
    const int n_ids = a.n_focus_ids;
    const int kk = lane < n_ids ? lane : 0;
    const float offy_l = a.offsets[a.focus_ids[kk]].y;
    const uint8_t *pad = reinterpret_cast<const uint8_t *>(w.pad) + 16u * lane;
    const size_t tap_stride = (size_t)a.radius_y * w.Wp * 4;
    // units = the flagged (row, candidate) entries of the row lists (candidate-major, rows ascending: focus_plan_lists / focus_plan_prefix) × this
    // XCD's tiles.  (Rounds 3–4 walked ALL (row, tile, candidate) triples and skipped the unflagged ones after one load each: 270 dependent
    // L2 round trips per wave — the pass's whole time once it ran at one wave per SIMD beside the persistent range kernel.)
    const focus_const_u32_ptr row_prefix = (focus_const_u32_ptr)(uintptr_t)w.prefix;
    const uint32_t entries = row_prefix[32];
    int i = 0;
    for(uint32_t u = wave_id; u < entries * my_tiles; u += n_waves)
    {
        const uint32_t e = u / my_tiles, tile = xcd + 8u * (u % my_tiles);
        prefix_walk(row_prefix, e, i); // (u only grows: the walk continues where the last unit's ended)
        const int y = __builtin_amdgcn_readfirstlane(int(w.rows[(size_t)i * a.height + (e - row_prefix[i])]));
        const uint32_t flags = __builtin_amdgcn_readfirstlane(w.bady[y]);
        {
            const uint32_t slot = line_slot(__builtin_amdgcn_readfirstlane(w.rowbase[y]), flags, i);
            if(slot >= uint32_t(w.R_cap))
                continue; // focus_exact (B)
            const float f = focus_candidate(a, i);
            // lane k: byte offset of view k's ty = 0 sample of this tile's first column
            const int sx_l = w.shifts[4 * (i * FOCUS_MAX_IDS + kk)];
            const int row_l = warp_float(y, f, offy_l) - a.radius_y + w.Py;
            const uint64_t off_l = (((uint64_t)kk * w.Hp + row_l) * w.Wp + uint64_t(int(tile) * 256 - a.radius_x + w.Px + sx_l)) * 4u;
            // the taps (ty = t − 1) of this row that need their own line for this candidate: wave-uniform, usually just t = 2
            uint32_t tm = 0u;
#pragma unroll
            for(int t = 0; t < 3; t++)
                tm |= ((__builtin_amdgcn_readfirstlane(w.tapy[(size_t)t * a.height + y]) >> i) & 1u) << t;
            // One tap at a time (almost always there is just one: t = 2), EIGHT views in flight: these passes are latency-bound, and since
            // round 5 they run at one wave per SIMD beside the persistent range kernel (focus_range_t) — their memory-level parallelism per
            // wave is what sets their time (three taps' registers for a two-pair prefetch took 0.73 ms beside it).
            const int last = n_ids - 1;
#pragma unroll 1
            for(int t = 0; t < 3; t++)
            {
                if(!(tm & (1u << t))) // wave-uniform
                    continue;
                RangeAcc4 acc;
                acc.init();
                u32x4 va[4], vb[4];
                auto fetch4 = [&](const int k0, u32x4 (&v)[4]) { // views k0 … k0 + 3, clamped to the last (reducing a view twice changes nothing)
#pragma unroll
                    for(int q = 0; q < 4; q++)
                        v[q] = *reinterpret_cast<const u32x4_a4 *>(pad + readlane64(off_l, min(k0 + q, last)) + t * tap_stride);
                };
                fetch4(0, va);
                fetch4(4, vb);
                for(int k = 0; k < n_ids; k += 8)
                {
                    acc.add2(va[0], va[1]);
                    acc.add2(va[2], va[3]);
                    if(k + 8 < n_ids)
                        fetch4(k + 8, va);
                    if(k + 4 < n_ids)
                    {
                        acc.add2(vb[0], vb[1]);
                        acc.add2(vb[2], vb[3]);
                    }
                    if(k + 12 < n_ids)
                        fetch4(k + 12, vb);
                }
                // (the other lines of the slot stay unwritten: focus_line_keys takes those taps from E)
                uint16_t *dst = w.Er + ((size_t)slot * 3 + t) * w.We_p + tile * 256 + 4 * lane;
                *reinterpret_cast<u32x2 *>(dst) = acc.encode();
            }
        }
    }
}

// E''(tx)(qy)(slot) for the flagged columns: a lane owns one flagged column (its sample columns are the reference's
// arithmetic for (column x, view k) plus (tx − 1)·rx), a wave takes (64 flagged columns of a candidate, 2 extended rows) units;
// the source rows are uniform shifts.  XCD x takes the x-th eighth of the row blocks and walks it with all candidates'
// chunks of a row block back to back (their bands overlap: one L2 serves them).  Same register-resident parameters and
// one-view-ahead fetch as focus_lines_rows.  Round 5: EIGHT rows per unit and one tap at a time (usually there is one): 32 loads in flight per
// lane instead of 12, a quarter of the units (each starts with a chain of dependent list look-ups) — the pass is latency-bound and runs at one
// wave per SIMD beside the persistent range kernel.
constexpr int FOCUS_COL_ROWS = 4;
__device__ __forceinline__ void focus_lines_cols(const KernelArgs &a, const FocusWork &w, const uint32_t block, const uint32_t blocks)
{
    constexpr int R = FOCUS_COL_ROWS;
    const int lane = threadIdx.x & 63;
    const uint32_t xcd = block & 7u;
    const uint32_t wave_id = __builtin_amdgcn_readfirstlane((block >> 3) * 4 + (threadIdx.x >> 6));
    const uint32_t n_waves = (blocks >> 3) * 4; // blocks is a multiple of 8
    const focus_const_u32_ptr chunk_prefix = (focus_const_u32_ptr)(uintptr_t)(w.prefix + 66);
    const focus_const_int_ptr ncols = (focus_const_int_ptr)(uintptr_t)w.ncols;
    const uint32_t row_blocks = (uint32_t(w.He_p) + uint32_t(R) - 1u) / uint32_t(R); // (the last block may reach below the extended image: its rows are not stored)
    const uint32_t rb0 = row_blocks * xcd / 8u, rb1 = row_blocks * (xcd + 1u) / 8u;
    const uint32_t chunks = chunk_prefix[32];
    const int n_ids = a.n_focus_ids;
    const int kk = lane < n_ids ? lane : 0;
    const float offx_l = a.offsets[a.focus_ids[kk]].x;
    const uint8_t *pad = reinterpret_cast<const uint8_t *>(w.pad);
    const int rx = a.radius_x;
    const size_t row_bytes = (size_t)w.Wp * 4;
    for(uint32_t u = wave_id; u < (rb1 - rb0) * chunks; u += n_waves)
    {
        const uint32_t rb = rb0 + u / chunks, chunk = u % chunks;
        int i = 0;
        prefix_walk(chunk_prefix, chunk, i);
        const int j = int(chunk - chunk_prefix[i]) * 64 + lane;
        const bool listed = j < ncols[i];
        const int x = listed ? w.cols[(size_t)i * a.width + j] : 0;
        const uint32_t cs = line_slot(w.colbase[x], w.badx[x], i);
        const bool active = listed && cs < uint32_t(w.C_cap);
        if(__builtin_amdgcn_ballot_w64(active) == 0ull)
            continue; // focus_exact (C)
        const float f = focus_candidate(a, i);
        // lane k: byte offset of view k's source row for the block's first extended row
        const int sy_l = w.shifts[4 * (i * FOCUS_MAX_IDS + kk) + 1];
        const uint64_t off_l = ((uint64_t)kk * w.Hp + uint64_t(int(rb) * R - a.radius_y + sy_l + w.Py)) * row_bytes;
        // the taps (tx = t − 1) of this lane's column that need their own line for this candidate (per lane; usually just t = 2): the
        // other loads are masked off — an inactive lane makes no request — and their lines stay unwritten (focus_line_keys reads E there)
        uint32_t tm = 0u;
#pragma unroll
        for(int t = 0; t < 3; t++)
            tm |= ((w.tapx[(size_t)t * a.width + x] >> i) & 1u) << t;
        tm = active ? tm : 0u;
        const int last = n_ids - 1;
#pragma unroll 1
        for(int t = 0; t < 3; t++)
        {
            const bool want = (tm >> t) & 1u;
            if(__builtin_amdgcn_ballot_w64(want) == 0ull)
                continue;
            RangeAcc1 acc[R]; // [row of the block]
#pragma unroll
            for(int r = 0; r < R; r++)
                acc[r].init();
            uint32_t cur[R], nxt[R], cur2[R], nxt2[R];
            auto fetch = [&](const int k, uint32_t (&v)[R]) {
                const float offx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(uint32_t, offx_l), k));
                const uint32_t left = uint32_t(warp_float(x, f, offx) - rx + w.Px + t * rx) * 4u; // this tap's sample column, bytes
                const uint8_t *row = pad + readlane64(off_l, k);
#pragma unroll
                for(int r = 0; r < R; r++)
                    v[r] = want ? *reinterpret_cast<const uint32_t *>(row + r * row_bytes + left) : 0u;
            };
            auto reduce2 = [&](const uint32_t (&va)[R], const uint32_t (&vb)[R]) { // two views per reduction (min3 / max3)
#pragma unroll
                for(int r = 0; r < R; r++)
                    acc[r].add2(va[r], vb[r]);
            };
            fetch(0, cur);
            fetch(min(1, last), nxt);
            int k = 0;
            for(; k + 4 < n_ids; k += 4)
            {
                fetch(k + 2, cur2);
                fetch(k + 3, nxt2);
                reduce2(cur, nxt);
                fetch(k + 4, cur);
                fetch(min(k + 5, last), nxt);
                reduce2(cur2, nxt2);
            }
            fetch(min(k + 2, last), cur2);
            fetch(min(k + 3, last), nxt2);
            reduce2(cur, nxt);
            reduce2(cur2, nxt2);
            if(want)
            {
#pragma unroll
                for(int r = 0; r < R; r++)
                    if(int(rb) * R + r < w.He_p)
                    {
                        uint32_t range, tiny;
                        acc[r].result(range, tiny);
                        w.Ec[((size_t)t * w.He_p + rb * R + r) * w.C_cap + cs] = static_cast<uint16_t>((range << 4) + tiny);
                    }
            }
        }
    }
}

// exact key of one (pixel, candidate): the reference's arithmetic tap by tap, in the integer formulation of focus_map.hpp.
// Samples come from the padded planes: the unclamped coordinate, offset by (Px, Py), holds the clamp-to-edge value.
// m9: the taps to compute (bit 3·tx + ty, per lane; the other taps' loads are masked off).  Returns Σ over those taps of
// 16·range + (FLT_MIN tap ? 1 : 0) — with all nine taps, focus_key_encode of it is the pixel's key.
__device__ __forceinline__ uint32_t focus_exact_taps(const KernelArgs &a, const FocusWork &w, const int x, const int y, const int i, const uint32_t m9)
{
    const int rx = a.radius_x, ry = a.radius_y;
    const float f = focus_candidate(a, i);
    const focus_const_float_ptr c_offsets = (focus_const_float_ptr)(uintptr_t)a.offsets;
    const focus_const_int_ptr c_ids = (focus_const_int_ptr)(uintptr_t)a.focus_ids;
    const size_t plane_bytes = (size_t)w.Wp * w.Hp * 4;
    RangeAcc1 acc[9];
#pragma unroll
    for(int t = 0; t < 9; t++)
        acc[t].init();
    const uint8_t *pad = reinterpret_cast<const uint8_t *>(w.pad);
    // top-left tap of view k; all nine taps are at non-negative, wave-uniform byte offsets from it
    auto corner_of = [&](const int k) {
        const int g = c_ids[k];
        const float offx = c_offsets[2 * g], offy = c_offsets[2 * g + 1];
        const int cx = warp_float(x, f, offx), cy = warp_float(y, f, offy);
        return pad + (size_t)k * plane_bytes + uint32_t((cy - ry + w.Py) * w.Wp + (cx - rx + w.Px)) * 4u;
    };
    const int last = a.n_focus_ids - 1;
    for(int k = 0; k <= last; k += 2) // two views per reduction (min3 / max3); an odd tail repeats the last view
    {
        const uint8_t *ca = corner_of(k), *cb = corner_of(min(k + 1, last));
#pragma unroll
        for(int ty = 0; ty < 3; ty++)
#pragma unroll
            for(int tx = 0; tx < 3; tx++)
            {
                const uint32_t tap = uint32_t(ty * ry * w.Wp + tx * rx) * 4u;
                const bool want = (m9 >> (tx * 3 + ty)) & 1u;
                const uint32_t pa = want ? *reinterpret_cast<const uint32_t *>(ca + tap) : 0u;
                const uint32_t pb = want ? *reinterpret_cast<const uint32_t *>(cb + tap) : 0u;
                acc[tx * 3 + ty].add2(pa, pb);
            }
    }
    uint32_t sum = 0;
#pragma unroll
    for(int t = 0; t < 9; t++)
    {
        uint32_t range, tiny;
        acc[t].result(range, tiny);
        sum += ((m9 >> t) & 1u) ? (range << 4) + tiny : 0u;
    }
    return sum;
}

// the key of a pixel from the sum of its nine taps' 16·range + tiny: 16·S if any range is non-zero, else the FLT_MIN count (focus_map.hpp)
__device__ __forceinline__ uint32_t focus_key_encode(const uint32_t sum)
{
    return sum >= 16u ? (sum & ~15u) : sum;
}

// Tap-by-tap keys for what the line buffers do not cover.  Persistent, three unit sequences:
//   (A) flagged row of a candidate × that candidate's flagged columns (pairs flagged on both axes);
//   (B) flagged rows whose line slot is ≥ R_cap × every column;   (C) flagged columns whose line slot is ≥ C_cap × every row.
// (B) and (C) run only when there are more flagged pairs than line slots.
__device__ __forceinline__ void focus_exact(const KernelArgs &a, const FocusWork &w, const uint32_t block, const uint32_t blocks)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave_id = __builtin_amdgcn_readfirstlane(block * 4 + (threadIdx.x >> 6));
    const uint32_t n_waves = blocks * 4;
    const int W = a.width, H = a.height;
    const focus_const_u32_ptr prefix = (focus_const_u32_ptr)(uintptr_t)w.prefix;
    const focus_const_int_ptr ncols = (focus_const_int_ptr)(uintptr_t)w.ncols;
    const uint32_t row_entries = prefix[32], col_entries = prefix[65];

    // flagged columns of candidate i in row y.  A pair whose row AND column both have a line slot gets only its CORNER taps here — the
    // taps where the uniform shift fails on both axes, usually the one at (+r, +r) — as a raw partial sum; focus_line_keys adds the
    // other taps from the row's lines, the column's lines and E.  Pairs without a slot on either axis get all nine taps (the whole key).
    auto columns = [&](const int i, const int y, const bool overflow_only) {
        const int n = ncols[i];
        const bool row_has_slot = __builtin_amdgcn_readfirstlane(line_slot(w.rowbase[y], w.bady[y], i)) < uint32_t(w.R_cap);
        uint32_t tmy = 0u; // wave-uniform
#pragma unroll
        for(int t = 0; t < 3; t++)
            tmy |= ((__builtin_amdgcn_readfirstlane(w.tapy[(size_t)t * H + y]) >> i) & 1u) << t;
        for(int idx = lane; idx - lane < n; idx += 64)
        {
            bool active = idx < n;
            const int x = active ? w.cols[(size_t)i * W + idx] : 0;
            const bool col_has_slot = line_slot(w.colbase[x], w.badx[x], i) < uint32_t(w.C_cap);
            if(overflow_only)
                active = active && !col_has_slot;
            if(__builtin_amdgcn_ballot_w64(active) == 0ull)
                continue;
            const bool whole = overflow_only || !row_has_slot || !col_has_slot;
            uint32_t m9 = 0x1ffu;
            if(!whole)
            {
                m9 = 0u;
#pragma unroll
                for(int tx = 0; tx < 3; tx++)
                    m9 |= ((w.tapx[(size_t)tx * W + x] >> i) & 1u) ? tmy << (3 * tx) : 0u;
            }
            const uint32_t sum = focus_exact_taps(a, w, x, y, i, active ? m9 : 0u);
            if(active)
                w.K[((size_t)i * H + y) * W + x] = static_cast<uint16_t>(whole ? focus_key_encode(sum) : sum);
        }
    };
    // (A)
    int i = 0;
    for(uint32_t entry = wave_id; entry < row_entries; entry += n_waves)
    {
        prefix_walk(prefix, entry, i);
        columns(i, w.rows[(size_t)i * H + (entry - prefix[i])], false);
    }
    // (B)
    if(row_entries > uint32_t(w.R_cap))
    {
        const uint32_t chunks_w = uint32_t((W + 63) / 64);
        i = 0;
        for(uint32_t u = wave_id; u < row_entries * chunks_w; u += n_waves)
        {
            const uint32_t entry = u / chunks_w;
            prefix_walk(prefix, entry, i);
            const int y = __builtin_amdgcn_readfirstlane(int(w.rows[(size_t)i * H + (entry - prefix[i])]));
            if(__builtin_amdgcn_readfirstlane(line_slot(w.rowbase[y], w.bady[y], i)) < uint32_t(w.R_cap))
                continue;
            const int xx = int(u % chunks_w) * 64 + lane;
            const bool active = xx < W;
            const uint32_t key = focus_key_encode(focus_exact_taps(a, w, active ? xx : 0, y, i, active ? 0x1ffu : 0u));
            if(active)
                w.K[((size_t)i * H + y) * W + xx] = static_cast<uint16_t>(key);
        }
    }
    // (C)
    if(col_entries > uint32_t(w.C_cap))
        for(uint32_t u = wave_id; u < uint32_t(FOCUS_STEPS) * uint32_t(H); u += n_waves)
            columns(int(u / uint32_t(H)), int(u % uint32_t(H)), true);
}

// The three passes over flagged pairs are independent, small and latency-bound: one launch, the workgroups split between
// them (gridDim.x = 3·per_pass, per_pass a multiple of 8), so that they run side by side — next to focus_range on the
// main stream — instead of one after the other.
__global__ void __launch_bounds__(256) focus_flagged(const KernelArgs a, const FocusWork w, const uint32_t per_pass, const uint32_t first_pass)
{
    const uint32_t pass = first_pass + blockIdx.x / per_pass, block = blockIdx.x % per_pass; // wave-uniform
    if(pass == 0)
        focus_lines_rows(a, w, block, per_pass);
    else if(pass == 1)
        focus_lines_cols(a, w, block, per_pass);
    else
        focus_exact(a, w, block, per_pass);
}

// K for the pairs flagged on one axis: the nine samples of the row lines Er / column lines Ec, summed and encoded, so that
// focus_pick has a single override source.  Persistent, two unit sequences:
//   (R) flagged row of a candidate (line slot < R_cap) × 64-column chunk, skipping the columns flagged for that candidate;
//   (C) 64 flagged columns of a candidate (line slot < C_cap) × block of 8 rows, skipping the rows flagged for it.
__global__ void __launch_bounds__(256) focus_line_keys(const KernelArgs a, const FocusWork w)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave_id = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const uint32_t n_waves = gridDim.x * 4;
    const int W = a.width, H = a.height, rx = a.radius_x, ry = a.radius_y;
    const focus_const_u32_ptr prefix = (focus_const_u32_ptr)(uintptr_t)w.prefix;
    const focus_const_u32_ptr chunk_prefix = prefix + 66;
    const focus_const_int_ptr ncols = (focus_const_int_ptr)(uintptr_t)w.ncols;
    auto encode = [](const uint32_t sum) { return focus_key_encode(sum); };
    // (R)  a wave's work item = four 64-pixel chunks of one flagged (row, candidate) entry: the entry's row, slot and tap flags are a chain
    // of dependent loads, paid once per item
    constexpr uint32_t CG = 4;
    const uint32_t chunks_w = (uint32_t((W + 63) / 64) + CG - 1) / CG;
    int i = 0;
#if defined(LK_SKIP) && (LK_SKIP & 1) // measurement builds (tools/line_keys_parts.sh): the kernel without one of its three parts
    if(false)
#endif
    for(uint32_t u = wave_id; u < prefix[32] * chunks_w; u += n_waves)
    {
        const uint32_t entry = u / chunks_w;
        prefix_walk(prefix, entry, i);
        const int y = __builtin_amdgcn_readfirstlane(int(w.rows[(size_t)i * H + (entry - prefix[i])]));
        const uint32_t slot = __builtin_amdgcn_readfirstlane(line_slot(w.rowbase[y], w.bady[y], i));
        if(slot >= uint32_t(w.R_cap))
            continue; // focus_exact (B)
        // per tap row: the row's own line where the uniform shift fails for it (wave-uniform), the candidate's plane of E otherwise
        const uint16_t *line[3];
#pragma unroll
        for(int ty = 0; ty < 3; ty++)
        {
            const bool own = (__builtin_amdgcn_readfirstlane(w.tapy[(size_t)ty * H + y]) >> i) & 1u;
            line[ty] = own ? w.Er + ((size_t)slot * 3 + ty) * w.We_p : w.E + ((size_t)i * w.He_p + y + ty * ry) * w.We_p;
        }
#pragma unroll
        for(uint32_t c = 0; c < CG; c++)
        {
            const int x = int((u % chunks_w) * CG + c) * 64 + lane;
            if(x >= W || ((w.badx[x] >> i) & 1u))
                continue; // flagged on both axes: focus_exact (A)
            uint32_t sum = 0;
#pragma unroll
            for(int ty = 0; ty < 3; ty++)
#pragma unroll
                for(int tx = 0; tx < 3; tx++)
                    sum += line[ty][x + tx * rx];
            w.K[((size_t)i * H + y) * W + x] = static_cast<uint16_t>(encode(sum));
        }
    }
    // (C)  a wave's unit = 64 flagged columns of a candidate (a lane each) × a COMB of CR rows ry apart: the tap rows of comb row k are comb rows
    // k, k + 1, k + 2, so the three-tap sums H(q) = Σ_tx tap(x + tx·rx, q) of CR + 2 rows of E (or of the column's own lines) serve CR rows of keys
    // — 3·(CR + 2) samples per lane instead of 9·CR, every row of E under a chunk read once (blocks of 8 adjacent rows left the re-use of a row,
    // ry and 2·ry rows further down, to the L2: FETCH_SIZE 604 MB for ≈ 100 MB of lines, profiles/r05_notes.md §5).
    constexpr int CR = 6; // (4 / 6 / 8 / 12 rows per comb: 107 / 102 / 99 / 96 µs with the chunks' XCD affinity, 93 / 94 µs for 6 / 12 without)
    const int rd = ry > 0 ? ry : 1;
    const uint32_t combs = uint32_t((H + CR * rd - 1) / (CR * rd)) * uint32_t(rd); // bands of CR·ry rows, ry combs each
    // Chunks in plain order over all waves.  (Blocks of adjacent rows wanted a chunk of columns on ONE XCD — 150 → 130 µs then; with the combs every
    // row of E under a chunk is read once and the affinity costs: 102 against 93 µs.)
    int ci = 0;
#if defined(LK_SKIP) && (LK_SKIP & 2)
    if(false)
#endif
    for(uint32_t u = wave_id; u < chunk_prefix[32] * combs; u += n_waves)
    {
        const uint32_t chunk = u / combs, comb = u % combs;
        prefix_walk(chunk_prefix, chunk, ci);
        const int j = int(chunk - chunk_prefix[ci]) * 64 + lane;
        if(j >= ncols[ci])
            continue;
        const int x = w.cols[(size_t)ci * W + j];
        const uint32_t cs = line_slot(w.colbase[x], w.badx[x], ci);
        if(cs >= uint32_t(w.C_cap))
            continue; // focus_exact (C)
        const uint16_t *tap_base[3]; // per tap column (per lane): the column's own line, or the candidate's plane of E
        uint32_t tap_pitch[3];
#pragma unroll
        for(int tx = 0; tx < 3; tx++)
        {
            const bool own = (w.tapx[(size_t)tx * W + x] >> ci) & 1u;
            tap_base[tx] = own ? w.Ec + (size_t)tx * w.He_p * w.C_cap + cs : w.E + (size_t)ci * w.He_p * w.We_p + x + tx * rx;
            tap_pitch[tx] = own ? uint32_t(w.C_cap) : uint32_t(w.We_p);
        }
        const int band = int(comb) / rd, y0 = band * CR * rd + (int(comb) - band * rd); // wave-uniform
        if(y0 >= H)
            continue;
        // all 3·(CR + 2) samples in flight together (rows below the extended image: the last one again, never used)
        uint32_t hsum[CR + 2];
#pragma unroll
        for(int m = 0; m < CR + 2; m++)
        {
            const uint32_t q = uint32_t(min(y0 + m * ry, w.He_p - 1));
            hsum[m] = uint32_t(tap_base[0][q * tap_pitch[0]]) + uint32_t(tap_base[1][q * tap_pitch[1]]) + uint32_t(tap_base[2][q * tap_pitch[2]]);
        }
#pragma unroll
        for(int k = 0; k < CR; k++)
        {
            const int y = y0 + k * rd;
            if(y < H && !((w.bady[y] >> ci) & 1u)) // (a row flagged for the candidate: focus_exact (A))
                w.K[((size_t)ci * H + y) * W + x] = static_cast<uint16_t>(encode(hsum[k] + hsum[k + 1] + hsum[k + 2]));
        }
    }
    // (A') pairs flagged on both axes whose row and column both have a line slot: focus_exact left the corner taps' partial sum in K;
    // the other taps come from the row's own lines (ty flagged), the column's own lines (tx flagged) or E (neither)
    int ai = 0;
#if defined(LK_SKIP) && (LK_SKIP & 4)
    if(false)
#endif
    for(uint32_t entry = wave_id; entry < prefix[32]; entry += n_waves)
    {
        prefix_walk(prefix, entry, ai);
        const int y = __builtin_amdgcn_readfirstlane(int(w.rows[(size_t)ai * H + (entry - prefix[ai])]));
        const uint32_t rs = __builtin_amdgcn_readfirstlane(line_slot(w.rowbase[y], w.bady[y], ai));
        if(rs >= uint32_t(w.R_cap))
            continue; // focus_exact wrote whole keys for this row
        bool owny[3]; // wave-uniform
#pragma unroll
        for(int t = 0; t < 3; t++)
            owny[t] = (__builtin_amdgcn_readfirstlane(w.tapy[(size_t)t * H + y]) >> ai) & 1u;
        const int nc = ncols[ai];
        for(int idx = lane; idx < nc; idx += 64)
        {
            const int x = w.cols[(size_t)ai * W + idx];
            const uint32_t cs = line_slot(w.colbase[x], w.badx[x], ai);
            if(cs >= uint32_t(w.C_cap))
                continue; // a whole key
            uint16_t *kp = w.K + ((size_t)ai * H + y) * W + x;
            uint32_t sum = *kp;
#pragma unroll
            for(int tx = 0; tx < 3; tx++)
            {
                const bool ownx = (w.tapx[(size_t)tx * W + x] >> ai) & 1u;
#pragma unroll
                for(int ty = 0; ty < 3; ty++)
                {
                    if(ownx && owny[ty])
                        continue; // a corner tap: in the partial sum
                    uint32_t v;
                    if(owny[ty]) // the row's own line (uniform shift in x)
                        v = w.Er[((size_t)rs * 3 + ty) * w.We_p + x + tx * rx];
                    else if(ownx) // the column's own line (uniform shift in y)
                        v = w.Ec[((size_t)tx * w.He_p + y + ty * ry) * w.C_cap + cs];
                    else
                        v = w.E[((size_t)ai * w.He_p + y + ty * ry) * w.We_p + x + tx * rx];
                    sum += v;
                }
            }
            *kp = static_cast<uint16_t>(encode(sum));
        }
    }
}

// dispersion per candidate = nine samples of E, or K where the pair is flagged on either axis; first strict minimum → map 0.
// Blocks of 4 rows × 64·PPL pixels, a lane owns PPL ∈ {1, 2} adjacent pixels; PPL = 2 reads both pixels' samples with one
// dword load and needs an even radius_x (the E columns x + rx ± rx of an even x are then dword aligned) — the reference always
// produces one (src/interpolator.cu:143-146).  Every sample is scalar plane base + per-lane tap offset.
template <int PPL>
__global__ void __launch_bounds__(256) focus_pick(const KernelArgs a, const FocusWork w, const int striped)
{
    const int W = a.width, H = a.height;
    const uint32_t blocks_x = uint32_t(W + 64 * PPL - 1) / uint32_t(64 * PPL);
    uint32_t bx, by;
    if(striped)
    {
        // round 3: one vertical stripe of blocks per XCD, walked top to bottom.  A row of E is sampled by three pixel rows 2·ry apart; in
        // row-major order over the whole width those meet in no L2 (2·ry rows × 32 candidates × the image width: 11 MB at 4K) and E came
        // from the fabric three times (TCC hit rate 4 %, profiles/r02_pmc_focus_range_summary.txt); an eighth of the width fits.
        uint32_t in;
        if(!stripe_map(blockIdx.x, blocks_x, uint32_t(H + 3) / 4u, 1u, bx, by, in))
            return;
    }
    else
    {
        bx = blockIdx.x % blocks_x;
        by = blockIdx.x / blocks_x;
    }
    const int x = (int(bx) * 64 + int(threadIdx.x & 63)) * PPL;
    const int y = __builtin_amdgcn_readfirstlane(int(by) * 4 + int(threadIdx.x >> 6));
    if(y >= H) // wave-uniform
        return;
    const int rx = a.radius_x, ry = a.radius_y;
    // lanes past the right edge compute pixel 0 and store nothing; the second pixel of a lane at x = W − 1 (odd W) reads
    // one element past a row of badx / E / K, inside the workspace, and is not stored either
    // (round 5, both measured slower than these 4-byte loads' 272 µs at 4K and removed — profiles/r05_notes.md §2: FOUR pixels per lane, two
    // dwords per sample load: 436 µs (8-byte loads at 4-byte alignment, half the waves); E staged through LDS by aligned 16-byte LDS-DMA
    // pieces, five wave-loads per batch of four candidates instead of 36: 469 µs (a barrier and an exposed fetch latency per batch))
    const int xs = x < W ? x : 0;
    const uint32_t flagged_y = __builtin_amdgcn_readfirstlane(w.bady[y]);
    uint32_t flagged[PPL];
    bool any = false;
#pragma unroll
    for(int j = 0; j < PPL; j++)
    {
        flagged[j] = w.badx[xs + j] | flagged_y;
        any = any || flagged[j] != 0u;
    }
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
                const uint32_t k_exact = exact[j]; // unconditional: a divergent load costs more than the unused values
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
#pragma unroll 4
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

// focus_pick with the tap block taken apart (round 5): the dispersion of a pixel is Σ_ty Σ_tx E(x + tx·rx, y + ty·ry) — a vertical three-sum
// V(x) = Σ_ty E(x, y + ty·ry) followed by a horizontal one over V(x), V(x + rx), V(x + 2rx).  A wave owns 128 adjacent pixels of one row (two
// per lane, as focus_pick<2>); per candidate it loads the three tap rows ONCE over the columns its pixels' taps span — the 128 columns of its
// own pixels by all lanes, the 2·rx columns behind them by the first rx lanes — sums them vertically (packed u16) and fetches V(x + rx),
// V(x + 2rx) from the lanes rx/2 and rx further on (ds_bpermute_b32: the LDS crossbar, no memory): 3 full + 3 partial wave-loads per candidate
// instead of 9 full ones.  focus_pick<2> runs at the L1's tag rate (4-byte loads at 4-byte alignment: 5.6 tag accesses per load, the L1 busy
// 64 % of the kernel); the sums are the same integers in any order, so the map is bit-identical.  Needs an even radius_x ≤ 64 (the reference's
// is even, src/interpolator.cu:143-146; 64 ↔ images up to 6,400 pixels wide) — else focus_pick<PPL>.  Same block → pixel mapping (stripes per XCD).
#ifndef FPS_NW
#define FPS_NW 4 // (measurement builds: 2 … 16)
#endif
constexpr int FPS_WAVES = FPS_NW; // waves per workgroup of focus_pick_sep
#ifndef FPS_R
#define FPS_R 4 // (measurement builds: 1 … 8)
#endif
constexpr int FPS_ROWS = FPS_R; // rows of the map per wave (radius_y apart).  Waves × rows at 4K (profiles/r05_focus_pick_experiments.txt): 16 × 1 / 2 / 3 / 4 →
                                // 239 / 218 (208 with candidates in groups of two) / 237 / 233 µs; 8 × 4 → 199; 4 × 4 → 199 (the product); 4 × 6 → 202; 4 × 8 → 213; 2 × 8 → 207
// rows of workgroups of focus_pick_sep: bands of FPS_WAVES·FPS_ROWS·ry rows of the map, ry workgroups each
__host__ __device__ __forceinline__ uint32_t focus_pick_sep_block_rows(const int H, const int ry)
{
    const int d = ry > 0 ? ry : 1; // (a zero radius: the taps coincide; any row distance serves)
    return uint32_t((H + FPS_WAVES * FPS_ROWS * d - 1) / (FPS_WAVES * FPS_ROWS * d)) * uint32_t(d);
}

__global__ void __launch_bounds__(64 * FPS_WAVES) focus_pick_sep(const KernelArgs a, const FocusWork w)
{
    constexpr int R = FPS_ROWS, NR = R + 2; // rows of the map per wave, rows of E it loads for them
    const int W = a.width, H = a.height;
    const uint32_t blocks_x = uint32_t(W + 127) / 128u;
    const uint32_t bx = blockIdx.x % blocks_x, by = blockIdx.x / blocks_x;
    const int lane = int(threadIdx.x & 63);
    const int x0 = int(bx) * 128, x = x0 + 2 * lane;
    const int rx = a.radius_x, ry = a.radius_y;
    // A workgroup's waves and a wave's R rows are ry APART (row j of a band of NW·R·ry rows, then j + ry, j + 2·ry, …): the tap rows of a row are
    // its own and the next two of that sequence, so a wave loads R + 2 rows of E for R rows of the map (in registers) and the uses of a row of E
    // by the neighbouring waves fall on one CU at about the same time.  Workgroups in row-major order.  MEASURED, not derived (profiles/
    // r05_focus_pick_experiments.txt): four ADJACENT rows per workgroup in stripes per XCD (focus_pick<2>'s mapping) took 267 µs at 4K whatever the
    // kernel did per wave — loads per candidate, registers, occupancy, E's layout, HALF the bytes per sample — and longer the more rows a stripe had
    // in flight; 16 / 8 / 4 waves of one row each, ry apart, row-major: 238 / 253 / 272 µs; 16 waves of two rows: 208-218; four waves of four: 199.
    // (Not by fewer bytes from the fabric: FETCH_SIZE × 2 was 1.3 GB for 16 × 2 rows against 0.88 GB for the stripes.)
    const int rd = ry > 0 ? ry : 1;
    const int band = int(by) / rd, j0 = int(by) - band * rd;
    const int y0 = __builtin_amdgcn_readfirstlane(((band * FPS_WAVES + int(threadIdx.x >> 6)) * R) * rd + j0);
    if(y0 >= H) // wave-uniform
        return;
    // flags and K are indexed by image columns: lanes past the right edge take column 0 and store nothing.  E is indexed by EXTENDED columns
    // and every lane loads its own (x + 1 < We_p: We_p ≥ W + 2·rx rounded up to 256): the lanes past the edge hold taps of the lanes before it.
    // Rows of the wave below the image are computed like the others (their loads past a plane's end return 0) and not stored.
    const int xs = x < W ? x : 0;
    uint32_t flagged[R][2], any_flag = 0u;
#pragma unroll
    for(int r = 0; r < R; r++)
    {
        const uint32_t flagged_y = __builtin_amdgcn_readfirstlane(w.bady[min(y0 + r * rd, H - 1)]);
        flagged[r][0] = w.badx[xs] | flagged_y;
        flagged[r][1] = w.badx[xs + 1] | flagged_y;
        any_flag |= flagged[r][0] | flagged[r][1];
    }
    const bool wave_flagged = __builtin_amdgcn_ballot_w64(any_flag != 0u) != 0ull;
    // the candidates with a flagged pair anywhere in this wave: K is read for those only (the others read candidate 0's resident lines)
    uint32_t wave_mask = 0u;
    if(wave_flagged)
    {
        uint32_t m = any_flag;
#pragma unroll
        for(int off = 32; off >= 1; off >>= 1)
            m |= uint32_t(__shfl_xor(int(m), off));
        wave_mask = __builtin_amdgcn_readfirstlane(m);
    }
    // byte offsets inside a candidate's plane of E: this lane's two pixels in the wave's NR rows of E, and (lanes below rx) two of the 2·rx
    // columns behind the wave's 128; the other lanes' offset is out of the descriptor's range — a buffer load returns 0 for it without a memory access
    uint32_t main_off[NR], extra_off[NR];
#pragma unroll
    for(int q = 0; q < NR; q++)
    {
        const uint32_t row = uint32_t(y0 + q * ry) * uint32_t(w.We_p);
        main_off[q] = (row + uint32_t(x)) * 2u;
        extra_off[q] = lane < rx ? (row + uint32_t(x0 + 128 + 2 * lane)) * 2u : 0xfffffff0u;
    }
    // lane l's taps tx = 1, 2 are lanes l + rx/2 and l + rx of the wave's 64 + rx lanes-worth of V: beyond lane 63 they are in the extra part
    const int src1 = lane + (rx >> 1), src2 = lane + rx;
    const int idx1 = 4 * (src1 & 63), idx2 = 4 * (src2 & 63);
    const bool own1 = src1 < 64, own2 = src2 < 64;
    const uint32_t e_plane_bytes = uint32_t(w.He_p) * uint32_t(w.We_p) * 2u, k_plane_bytes = uint32_t(H) * uint32_t(W) * 2u; // < 2^28
    uint32_t k_off[R];
#pragma unroll
    for(int r = 0; r < R; r++)
        k_off[r] = (uint32_t(min(y0 + r * rd, H - 1)) * uint32_t(W) + uint32_t(xs)) * 2u;
    uint32_t best_key[R][2];
    int best_i[R][2];
#pragma unroll
    for(int r = 0; r < R; r++)
        best_key[r][0] = best_key[r][1] = 0xffffffffu, best_i[r][0] = best_i[r][1] = 0;
    // Candidates in groups of two: a group's loads of E (and its keys of K) are issued together, then summed.
#ifndef FPS_G
#define FPS_G 2 // (measurement builds: 1 / 2 / 4 / 8 candidates per group → 212 / 208 / 217 / 241 µs at 4K; two groups in flight, -DFPS_DB=1: 212 (G = 1), 237 (G = 2))
#endif
#ifndef FPS_DB
#define FPS_DB 0
#endif
    constexpr int G = FPS_G;
    static_assert(FOCUS_STEPS % (G * (1 + FPS_DB)) == 0, "whole groups");
    auto load_group = [&](const int i0, uint32_t (&em)[G][NR], uint32_t (&ee)[G][NR], uint32_t (&kx)[G][R][2], auto exact_tag) {
#pragma unroll
        for(int g = 0; g < G; g++)
        {
            const __amdgpu_buffer_rsrc_t re =
                __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint8_t *>(w.E) + (size_t)(i0 + g) * e_plane_bytes, 0, int(e_plane_bytes), 0x00020000);
#pragma unroll
            for(int q = 0; q < NR; q++)
            {
#ifdef FPS_HALF // measurement builds: half the bytes per sample (what an 8-bit E would move; the maps are wrong)
                em[g][q] = uint32_t(__builtin_amdgcn_raw_buffer_load_b16(re, int(main_off[q] >> 1), 0, 0));
                ee[g][q] = uint32_t(__builtin_amdgcn_raw_buffer_load_b16(re, int(extra_off[q] == 0xfffffff0u ? extra_off[q] : extra_off[q] >> 1), 0, 0));
#else
                em[g][q] = uint32_t(__builtin_amdgcn_raw_buffer_load_b32(re, int(main_off[q]), 0, 0));
                ee[g][q] = uint32_t(__builtin_amdgcn_raw_buffer_load_b32(re, int(extra_off[q]), 0, 0));
#endif
            }
            if constexpr(decltype(exact_tag)::value)
            {
                const size_t kplane = ((wave_mask >> (i0 + g)) & 1u) ? size_t(i0 + g) : size_t(0); // wave-uniform
                const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint8_t *>(w.K) + kplane * k_plane_bytes, 0, int(k_plane_bytes), 0x00020000);
#pragma unroll
                for(int r = 0; r < R; r++)
                {
                    kx[g][r][0] = uint32_t(__builtin_amdgcn_raw_buffer_load_b16(rk, int(k_off[r]), 0, 0));
                    kx[g][r][1] = uint32_t(__builtin_amdgcn_raw_buffer_load_b16(rk, int(k_off[r] + 2u), 0, 0));
                }
            }
        }
    };
    auto reduce_group = [&](const int i0, const uint32_t (&em)[G][NR], const uint32_t (&ee)[G][NR], const uint32_t (&kx)[G][R][2], auto exact_tag) {
        u16x2 vm[G][R], ve[G][R]; // V of this lane's pixel pair, of its extra pair, per row of the map: 3 · 4081 per half
#pragma unroll
        for(int g = 0; g < G; g++)
#pragma unroll
            for(int r = 0; r < R; r++)
            {
                vm[g][r] = as_u16x2(em[g][r]) + as_u16x2(em[g][r + 1]) + as_u16x2(em[g][r + 2]);
                ve[g][r] = as_u16x2(ee[g][r]) + as_u16x2(ee[g][r + 1]) + as_u16x2(ee[g][r + 2]);
            }
        uint32_t t1m[G][R], t1e[G][R], t2m[G][R], t2e[G][R];
#pragma unroll
        for(int g = 0; g < G; g++)
#pragma unroll
            for(int r = 0; r < R; r++)
            {
                t1m[g][r] = uint32_t(__builtin_amdgcn_ds_bpermute(idx1, int(as_u32(vm[g][r]))));
                t1e[g][r] = uint32_t(__builtin_amdgcn_ds_bpermute(idx1, int(as_u32(ve[g][r]))));
                t2m[g][r] = uint32_t(__builtin_amdgcn_ds_bpermute(idx2, int(as_u32(vm[g][r]))));
                t2e[g][r] = uint32_t(__builtin_amdgcn_ds_bpermute(idx2, int(as_u32(ve[g][r]))));
            }
#pragma unroll
        for(int g = 0; g < G; g++)
#pragma unroll
            for(int r = 0; r < R; r++)
            {
                const int i = i0 + g;
                const u16x2 acc = vm[g][r] + as_u16x2(own1 ? t1m[g][r] : t1e[g][r]) + as_u16x2(own2 ? t2m[g][r] : t2e[g][r]); // 9 · 4081 < 65536 per half
                const uint32_t sum[2] = {as_u32(acc) & 0xffffu, as_u32(acc) >> 16};
#pragma unroll
                for(int j = 0; j < 2; j++)
                {
                    uint32_t key = sum[j] >= 16u ? (sum[j] & ~15u) : sum[j];
                    if constexpr(decltype(exact_tag)::value)
                        key = ((flagged[r][j] >> i) & 1u) ? (kx[g][r][j] & 0xffffu) : key;
                    if(key < best_key[r][j]) // MinDispersion::add (src/kernels.cu:225-231): strict <
                    {
                        best_key[r][j] = key;
                        best_i[r][j] = i;
                    }
                }
            }
    };
    auto all_candidates = [&](auto exact_tag) {
        uint32_t em[G][NR], ee[G][NR], kx[G][R][2];
#if FPS_DB // measurement builds: two groups in flight
        uint32_t emB[G][NR], eeB[G][NR], kxB[G][R][2];
        load_group(0, em, ee, kx, exact_tag);
#pragma unroll 1
        for(int i0 = 0; i0 < FOCUS_STEPS; i0 += 2 * G)
        {
            load_group(i0 + G, emB, eeB, kxB, exact_tag);
            __builtin_amdgcn_sched_barrier(0);
            reduce_group(i0, em, ee, kx, exact_tag);
            __builtin_amdgcn_sched_barrier(0);
            if(i0 + 2 * G < FOCUS_STEPS)
                load_group(i0 + 2 * G, em, ee, kx, exact_tag);
            __builtin_amdgcn_sched_barrier(0);
            reduce_group(i0 + G, emB, eeB, kxB, exact_tag);
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
#endif
#pragma unroll 1
        for(int i0 = 0; i0 < FOCUS_STEPS; i0 += G)
        {
            load_group(i0, em, ee, kx, exact_tag);
            __builtin_amdgcn_sched_barrier(0);
            reduce_group(i0, em, ee, kx, exact_tag);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if(wave_flagged)
        all_candidates(std::true_type{});
    else
        all_candidates(std::false_type{});
#pragma unroll
    for(int r = 0; r < R; r++)
    {
        const int y = y0 + r * rd;
        if(y >= H) // wave-uniform
            break;
#pragma unroll
        for(int j = 0; j < 2; j++)
            if(x + j < W)
            {
                const float best_f = focus_candidate(a, best_i[r][j]);
                const float normalized = __fdiv_rn(best_f - a.focus, a.range);
                const uint32_t m = static_cast<uint32_t>(roundf(normalized * 255.0f)) & 0xffu;
                reinterpret_cast<uint32_t *>(a.maps)[(size_t)y * W + x + j] = m | (m << 8) | (m << 16) | 0xff000000u;
            }
    }
}

} // namespace lfi
