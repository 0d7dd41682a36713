// lfi_device.hpp — device-side helpers shared by the gfx950 kernels: the kernel argument block that replaces the
// reference's __constant__ symbols (reference src/kernels.cu:7-70), the warp / clamp-to-edge fetch
// (src/kernels.cu:72-82, 119-126), the focus-map decode (src/kernels.cu:134-137) and the synthetic-LF hash.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../../include/lfi.h"

namespace lfi {

constexpr int WAVE = 64;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// dword-aligned wide accesses: shifted image rows start at arbitrary pixel (4-byte) alignment
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

// Everything a blend / focus kernel reads besides pixels.  Passed by value as the kernel argument
// (kernarg segment → scalar loads), where the reference uses cudaMemcpyToSymbol (src/interpolator.cu:134-136,151-153).
struct KernelArgs
{
    const uint8_t *__restrict__ grid;       // [N][H][W][4]   input planes          (inputSurfaces)
    uint8_t *__restrict__ views;            // [V][H][W][4]   output planes         (outputSurfaces)
    uint8_t *__restrict__ maps;             // [2][H][W][4]   focus maps            (mapSurfaces)
    const lfi_int2 *__restrict__ focused;   // [Kpad]         integer offsets       (focusedOffsets), zero padded
    const lfi_float2 *__restrict__ offsets; // [Kpad]         float offsets         (offsets), zero padded
    const uint16_t *__restrict__ w16;       // [Vpad][Kpad]   fp16 weights, zero padded (weights, src/interpolator.cu:211)
    const uint16_t *__restrict__ w16s;      // [Vpad][Kpad]   the same weights × 2^15 (exact) for the packed epilogue; valid iff all in [0,2)
    const float *__restrict__ w32;          // [Vpad][Kpad]   the same weights widened to f32 (exact)
    const float *__restrict__ w32t;         // [Kpad][Vpad]   transposed copy for view-contiguous scalar loads
    const int32_t *__restrict__ focus_ids;  // [n_focus_ids]                        (focusMapIDs)
    float *__restrict__ prequant;           // optional [H][W][3] accumulators of view `prequant_view`
    int32_t prequant_view;
    int32_t width, height;                  // constants[0..1]: the FULL image size (clamping and parameters refer to it)
    // Row window (spatial multi-GPU sharding): this context holds rows [in_y0, in_y0+in_rows) of every input plane and
    // renders output rows [out_y0, out_y0+out_rows) into planes of out_rows rows.  Whole image: 0, height, 0, height.
    int32_t in_y0, in_rows, out_y0, out_rows;
    int32_t map_y0, map_rows;               // focus-map kernels: the rows of the maps to compute (whole image: 0, height)
    int32_t n_images;                       // constants[5]
    int32_t k_pad;                          // n_images rounded up to 16
    int32_t v_pad;                          // views rounded up to 64
    int32_t v0, v1;                         // view range of this launch
    int32_t n_focus_ids;
    int32_t radius_x, radius_y;             // constants[9..10]
    int32_t fo_min_x, fo_max_x, fo_min_y, fo_max_y; // bounds of focusedOffsets over the n_images images (interior-tile tests)
    // planar copy of the inputs (blend_planar.hpp), or nullptr: [image][channel R,G,B][rows held][planar_pitch] bytes,
    // pixel x of a row at byte x + planar_padx, edges replicated into the padding
    const uint8_t *__restrict__ planar;
    // … plus a per-image phase 0…127 (planar_phase[g]): pixel x of image g sits at byte x + planar_padx + planar_phase[g], chosen when the
    // copy is built so that the 128-byte runs of the offsets in use then ARE cache lines (the pitch is a multiple of 128; round 3 aligned to
    // dwords only: the boundary sectors a run shares with its neighbours were fetched twice — 10–19 % of the reads on 15×15 grids); any
    // other offsets are served correctly from the same copy, only their runs straddle lines
    const int32_t *__restrict__ planar_phase;
    int32_t planar_pitch, planar_padx;
    int32_t views_pitch;                    // planar view layout (blend_p3.hpp): bytes per row of a byte plane [view][R,G,B][out_rows][views_pitch]
    float std_band;                         // blend_planar<STDF>: half-width of the band around x.5 inside which a sum is recomputed exactly
    int32_t map_index;                      // which focus map an all-focus render reads
    float focus, range;                     // inFocus, inRange
    uint32_t flags;
};

// The band method's bound on the matrix core's accumulation error per addend (in units of S, sums below 512) for MORE than 64 images
// (blend_stdx, blend_stdxa, blend_afs): 2^-17, a quarter ulp(512), MEASURED on gfx950 — at most 0.21 ulp(512) = 0.83·2^-17 per addend over
// 800 adversarial operand sets (tests/test_gpu_parity.py::test_mfma_f16_accumulation_error_bound asserts 2^-17; rounds 2-3 had probed 2
// sets and seen 0.09 ulp) — plus the band's 2^-12 of margin.  LFI_FLAG_STD_ANALYTIC_BAND: 2^-15, true of any accumulator that keeps 24
// bits — twice the launch time at 15x15; a budget of 2^-16 would cost 8-12 % (profiles/r04_std_band_cost.txt), so the asserted bound stays
// the default here, while up to 64 images the analytic band is the default (lfi_context.hpp make_args: it is free there).
__host__ __device__ __forceinline__ float std_accumulation_bound(const uint32_t flags)
{
    return (flags & LFI_FLAG_STD_ANALYTIC_BAND) ? 0x1p-15f : 0x1p-17f;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi)
{
    return min(max(v, lo), hi);
}

// surf2Dread<uchar4>(…, cudaBoundaryModeClamp): src/kernels.cu:119-126
__device__ __forceinline__ uint32_t fetch_px(const uint8_t *__restrict__ grid, int width, int height, int g, int x, int y)
{
    x = clampi(x, 0, width - 1);
    y = clampi(y, 0, height - 1);
    const uint32_t *plane = reinterpret_cast<const uint32_t *>(grid) + (size_t)g * (size_t)(width * height);
    return plane[y * width + x];
}

// loadFocusFromMap: src/kernels.cu:134-137 (explicit fma: the contraction nvcc applies)
__device__ __forceinline__ float decode_focus(const uint8_t *__restrict__ map_plane, int width, int height, int x, int y,
                                              float focus, float range)
{
    x = clampi(x, 0, width - 1);
    y = clampi(y, 0, height - 1);
    uint32_t m = reinterpret_cast<const uint32_t *>(map_plane)[y * width + x] & 0xffu;
    float t = __fdiv_rn(static_cast<float>(m), 255.0f);
    return __builtin_fmaf(t, range, focus);
}

// focusCoords(int2, int, float): src/kernels.cu:78-82 — C truncation of fma(focus, offset, coord)
__device__ __forceinline__ int warp_float(int coord, float focus, float offset)
{
    return static_cast<int>(__builtin_fmaf(focus, offset, static_cast<float>(coord)));
}

// blocks b, b+8, b+16… share an XCD (and its L2); give each XCD one contiguous run of the tile sequence so that
// tiles that share cache lines at their edges meet in one L2.  Bijective for any grid size.
__device__ __forceinline__ uint32_t xcd_contiguous(uint32_t b, uint32_t nblocks)
{
    const uint32_t xcd = b & 7u, idx = b >> 3;
    const uint32_t q = nblocks >> 3, rem = nblocks & 7u;
    return xcd * q + min(xcd, rem) + idx;
}

// Internal bit of KernelArgs::flags (not an LFI_FLAG_*: set by the dispatcher per launch): the launch's tiles share NO cache lines with their
// neighbours — fixed-focus renders from the planar copy while its per-image phases are tuned for the offsets in use (every 128-byte run IS a
// line), all-focus renders (per-pixel gathers) — so block b takes tile b: neighbouring tiles run at the same time on DIFFERENT XCDs and their
// rows' neighbouring lines stream from memory together.  Round 5, second session (profiles/r05_notes.md §8): against the contiguous runs per XCD
// config 4's ranks −8 %, config 4 whole −3 %, config 5 with RGBA views −8 %, all-focus STD −3 %; with STALE phases (a fixed-focus sweep: runs
// straddle lines, neighbours share them) the contiguous runs win by 9–14 % — hence a flag and not a rule.
constexpr uint32_t LFI_KFLAG_PLAIN_TILE_ORDER = 1u << 31;
__device__ __forceinline__ uint32_t tile_of_block(uint32_t b, uint32_t nblocks, uint32_t flags)
{
    return (flags & LFI_KFLAG_PLAIN_TILE_ORDER) ? b : xcd_contiguous(b, nblocks);
}

// Vertical stripes: XCD x owns tile columns [x·tiles_x/8, (x+1)·tiles_x/8) and walks them row by row, `inner` work items per
// tile back to back.  For kernels whose tiles re-read a band of rows above and below: the band × stripe width × planes is
// what has to stay in one L2, instead of band × image width.  Launch 8·stripe_blocks_per_xcd(...) blocks; returns false
// for the surplus blocks of the narrower stripes.  Needs tiles_x ≥ 8 (use xcd_contiguous otherwise).
__host__ __device__ __forceinline__ uint32_t stripe_blocks_per_xcd(uint32_t tiles_x, uint32_t tiles_y, uint32_t inner)
{
    return ((tiles_x + 7u) / 8u) * tiles_y * inner;
}

__device__ __forceinline__ bool stripe_map(uint32_t b, uint32_t tiles_x, uint32_t tiles_y, uint32_t inner, uint32_t &tx,
                                           uint32_t &ty, uint32_t &in)
{
    const uint32_t xcd = b & 7u, idx = b >> 3;
    const uint32_t x0 = tiles_x * xcd / 8u, x1 = tiles_x * (xcd + 1u) / 8u, nx = x1 - x0;
    if(idx >= nx * tiles_y * inner)
        return false;
    in = idx % inner;
    const uint32_t t = idx / inner;
    tx = x0 + t % nx;
    ty = t / nx;
    return true;
}

__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

} // namespace lfi
