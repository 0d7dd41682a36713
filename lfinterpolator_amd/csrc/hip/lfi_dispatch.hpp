// lfi_dispatch.hpp — which kernel serves a render: the variant tables (lfi_set_variant), the launchers of every blend kernel, the
// derived planar input copy (ensure_planar) and the rules that pick between them (wants_planar, wants_p3, launch_blend).
// Replaces the method / allFocus dispatch of Interpolator::interpolate (reference src/interpolator.cu:270-290).
// Included by lfi_hip.hip only (one translation unit), after lfi_context.hpp.
#pragma once

#include "lfi_context.hpp"
#include "blend_std.hpp"
#include "blend_ten.hpp"
#include "blend_ten_persist.hpp"
#include "blend_planar.hpp"
#include "blend_p3.hpp"
#include "blend_stdx.hpp"
#include "blend_stdxa.hpp"
#include "blend_af.hpp"
#include "blend_wave.hpp"
#include "lfi_band_probe.hpp"

namespace {

struct Variant
{
    const char *name;
    void (*launch)(const lfi_ctx *, const KernelArgs &, bool all_focus);
    bool packed_epilogue; // TEN_WM: needs weights in [0,2) (×2^15 copy)
    bool prequant = false; // can dump pre-quantisation accumulators (the generic kernels only)
    bool row_window = false; // honours a row window (the persistent kernels)
    bool planar = false;     // reads the planar copy of the inputs when the launch qualifies (else its launcher falls back)
};

int next_sweep_direction(const lfi_ctx *c);

template <int PXL, int MT>
void launch_ten_direct(const lfi_ctx *c, const KernelArgs &a, bool all_focus)
{
    const int tiles_x = (a.width + 32 * PXL - 1) / (32 * PXL);
    const int n_tiles = tiles_x * a.height;
    const int passes = (a.v1 - a.v0 + 32 * MT - 1) / (32 * MT);
    const int vpw = passes >= 4 ? 4 : (passes >= 2 ? 2 : 1);
    const int tiles_per_wg = 4 / vpw;
    const dim3 grid((n_tiles + tiles_per_wg - 1) / tiles_per_wg), block(256);
    hipStream_t st = stream_of(c);
    if(flags_of(c) & LFI_FLAG_TEN_ROUND_PER_BATCH)
    {
        // the reference's half-accumulator model (oracle M16), exact: fp64 on the vector pipe, one pixel per lane (blend_ten.hpp)
        note_kernel(c, "blend_ten_m16");
        if(all_focus)
            hipLaunchKernelGGL(lfi::blend_ten_m16<true>, pixel_grid_of(c), dim3(256), 0, st, a);
        else
            hipLaunchKernelGGL(lfi::blend_ten_m16<false>, pixel_grid_of(c), dim3(256), 0, st, a);
        return;
    }
    note_kernel(c, "blend_ten_direct");
    if(all_focus)
        hipLaunchKernelGGL((lfi::blend_ten_direct<PXL, MT, true>), grid, block, 0, st, a, tiles_x, n_tiles, passes, vpw);
    else
        hipLaunchKernelGGL((lfi::blend_ten_direct<PXL, MT, false>), grid, block, 0, st, a, tiles_x, n_tiles, passes, vpw);
}

template <bool STD, int MT, bool NT_STORE, int KC = 64, int WGS = 2>
void launch_persist(const lfi_ctx *c, const KernelArgs &a, bool all_focus)
{
    constexpr int TPX = 128, VPP = MT * 32;
    const int tiles_x = (a.width + TPX - 1) / TPX;
    const int n_tiles = tiles_x * a.out_rows;
    const int passes = (a.v1 - a.v0 + VPP - 1) / VPP;
    // persistent: WGS workgroups per CU (2 x 80 KB of LDS at KC = 64), each walks tiles j, j+G, j+2G ...
    const dim3 grid(std::min(n_tiles, WGS * cu_count_of(c))), block(256);
    note_kernel(c, STD ? (all_focus ? "blend_persist<STD,allfocus>" : "blend_persist<STD>") : (all_focus ? "blend_persist<TEN_WM,allfocus>" : "blend_persist<TEN_WM>"));
    if constexpr(!STD && MT == 2 && KC == 64 && WGS == 2)
        if(all_focus && a.views == c->views && c->out_layout == LFI_LAYOUT_PLANAR_RGB)
        {
            // the planar layout's byte planes written by the kernel itself (launch_blend: persist_writes_planar_views)
            hipLaunchKernelGGL((lfi::blend_persist<false, 2, true, NT_STORE, 64, 2, true>), grid, block, 0, stream_of(c), a, tiles_x, n_tiles, passes);
            return;
        }
    if(all_focus)
        hipLaunchKernelGGL((lfi::blend_persist<STD, MT, true, NT_STORE, KC, WGS>), grid, block, 0, stream_of(c), a, tiles_x, n_tiles, passes);
    else
        hipLaunchKernelGGL((lfi::blend_persist<STD, MT, false, NT_STORE, KC, WGS>), grid, block, 0, stream_of(c), a, tiles_x, n_tiles, passes);
}

// TEN_WM from the planar copy of the inputs (blend_planar.hpp) when launch_blend has validated it for this launch
// (a.planar != nullptr: fixed focus), else blend_persist
template <bool NT_STORE, int RING3 = 1>
void launch_planar(const lfi_ctx *c, const KernelArgs &a, bool all_focus)
{
    if(!a.planar || all_focus)
    {
        launch_persist<false, 2, NT_STORE>(c, a, all_focus);
        return;
    }
    const int tiles_x = (a.width + 127) / 128;
    const int n_tiles = tiles_x * a.out_rows;
    const int passes = (a.v1 - a.v0 + 63) / 64;
    const dim3 grid(std::min(n_tiles, 2 * cu_count_of(c))), block(256);
    note_kernel(c, "blend_planar<TEN_WM>");
    hipLaunchKernelGGL((lfi::blend_planar<2, NT_STORE>), grid, block, 0, stream_of(c), a, tiles_x, n_tiles, passes, RING3, next_sweep_direction(c));
}

// RGBA views (the reference's layout, the default) by blend_p3 with its RGBA epilogue (round 4) where it pays — fixed focus, the planar
// copy of the inputs validated for this launch, TWO TO FOUR chunks of images (15×15 grids: −8 % at configs 3 and 5; with one chunk
// blend_planar is as fast, and blend_persist faster for several view passes: profiles/r04_rgba_p3_ab.txt), a wave's 16 RGBA planes
// addressable with 32 bits — else blend_planar / blend_persist.
void launch_p3(const lfi_ctx *c, const KernelArgs &a_in, bool rgba_out = false);
bool p3_rgba_planes_fit(const lfi_ctx *c)
{
    return (uint64_t)16 * (uint64_t)c->out_rows * (uint64_t)c->width * 4u < (1ull << 32);
}
void launch_p3_rgba(const lfi_ctx *c, const KernelArgs &a, bool all_focus)
{
    if(!a.planar || all_focus || a.k_pad <= lfi::P3_KC || a.k_pad > 4 * lfi::P3_KC || !p3_rgba_planes_fit(c))
    {
        launch_planar<true>(c, a, all_focus);
        return;
    }
    launch_p3(c, a, true);
}

// wave-private pipelines (blend_wave.hpp) where they apply — fixed focus, one K-chunk, one view pass — else blend_persist
template <bool STD, int MT, bool NT_STORE>
void launch_wave(const lfi_ctx *c, const KernelArgs &a, bool all_focus)
{
    if(all_focus || a.k_pad > 64 || a.v1 - a.v0 > 32 * MT)
    {
        launch_persist<STD, MT, NT_STORE>(c, a, all_focus);
        return;
    }
    const int tiles_x = (a.width + 127) / 128;
    const int n_tiles = tiles_x * a.out_rows;
    const dim3 grid(std::min(n_tiles, 2 * cu_count_of(c))), block(256);
    note_kernel(c, STD ? "blend_wave<STD>" : "blend_wave<TEN_WM>");
    hipLaunchKernelGGL((lfi::blend_wave<STD, MT, NT_STORE>), grid, block, 0, stream_of(c), a, tiles_x, n_tiles);
}

// STD through blend_planar<STDF> (MFMA sum + exact recomputation inside the rounding band) when launch_blend has validated the
// planar copy and the weights for it, else the exact-fp32 MFMA kernels
// GATHER_ONCE = true ("filtered_gather_once"): all-focus renders of three or four chunks of images by blend_afs (round 4: 64-pixel tiles whose
// whole stack of samples stays in LDS, every sample gathered once) instead of blend_stdxa, which gathers chunks 2 and 3 a second time for
// the chain.  Half the fabric traffic, the same bytes — and not faster (4.8 against 4.7 ms at config 5: bound by instruction issue, see
// profiles/r04_notes.md), so it is a selectable variant and the second implementation in the parity tests, not the default.
template <bool GATHER_ONCE>
void launch_std_filtered_t(const lfi_ctx *c, const KernelArgs &a_in, bool all_focus)
{
    if(all_focus && c->weights_scalable && c->weights_sum_ok && !a_in.prequant && a_in.k_pad <= 4 * 64)
    {
        // all-focus: blend_stdxa — the band method on the per-pixel gather pipeline (RGBA planes; no derived copy); one launch per 64 views
        const int tiles_x = (a_in.width + 127) / 128;
        const int n_tiles = tiles_x * a_in.out_rows;
        const dim3 grid(std::min(n_tiles, 2 * cu_count_of(c))), block(256);
        const int nch = (a_in.k_pad + 63) / 64;
        if(nch >= 3 && GATHER_ONCE)
        {
            // three or four chunks of images: blend_afs — 64-pixel tiles whose whole stack of samples stays in LDS, every sample gathered
            // once (blend_stdxa gathers chunks 2 and 3 a second time for the chain)
            const int tiles_x64 = (a_in.width + lfi::AF_TPX - 1) / lfi::AF_TPX;
            const int n_tiles64 = tiles_x64 * a_in.out_rows;
            const dim3 grid64(std::min(n_tiles64, 2 * cu_count_of(c)));
            note_kernel(c, "blend_afs<STD,allfocus>");
            for(int v0 = a_in.v0; v0 < a_in.v1; v0 += 64)
            {
                KernelArgs a = a_in;
                a.v0 = v0;
                a.v1 = std::min(v0 + 64, a_in.v1);
                if(nch == 3)
                    hipLaunchKernelGGL((lfi::blend_afs<true, 3>), grid64, block, 0, stream_of(c), a, tiles_x64, n_tiles64);
                else
                    hipLaunchKernelGGL((lfi::blend_afs<true, 4>), grid64, block, 0, stream_of(c), a, tiles_x64, n_tiles64);
            }
            return;
        }
        note_kernel(c, "blend_stdxa<STD,allfocus>");
        // the planar layout's byte planes written by the kernel itself (launch_blend: stdxa_writes_planar_views)
        const bool planar_views_af = a_in.views == c->views && c->out_layout == LFI_LAYOUT_PLANAR_RGB;
        for(int v0 = a_in.v0; v0 < a_in.v1; v0 += 64)
        {
            KernelArgs a = a_in;
            a.v0 = v0;
            a.v1 = std::min(v0 + 64, a_in.v1);
#define LFI_SXA_LAUNCH(N)                                                                                                                       \
    do                                                                                                                                          \
    {                                                                                                                                           \
        if(planar_views_af)                                                                                                                     \
            hipLaunchKernelGGL((lfi::blend_stdxa<true, N, true>), grid, block, 0, stream_of(c), a, tiles_x, n_tiles);                           \
        else                                                                                                                                    \
            hipLaunchKernelGGL((lfi::blend_stdxa<true, N>), grid, block, 0, stream_of(c), a, tiles_x, n_tiles);                                 \
    } while(0)
            switch(nch)
            {
                case 1: LFI_SXA_LAUNCH(1); break;
                case 2: LFI_SXA_LAUNCH(2); break;
                case 3: LFI_SXA_LAUNCH(3); break;
                default: LFI_SXA_LAUNCH(4); break;
            }
#undef LFI_SXA_LAUNCH
        }
        return;
    }
    if(!a_in.planar || all_focus || a_in.k_pad > 4 * lfi::P3_KC)
    {
        // (launch_blend hands this launcher the planar byte views only together with a valid planar copy or for the all-focus branch above:
        // the RGBA-store kernels below never see them)
        launch_wave<true, 2, true>(c, a_in, all_focus);
        return;
    }
    // planar views written directly (launch_blend has pointed a.views at the byte planes: stdx_writes_planar_views): blend_stdx also for ONE
    // chunk of images — blend_planar<STDF> would go through an RGBA scratch copy of all views and a conversion pass (config 2: 0.45 ms
    // against 0.23).  With RGBA views blend_planar<STDF> stays: as fast at config 2, 15 % faster for one rank of config 4
    // (profiles/r04_rgba_p3_ab.txt).
    const bool planar_views = a_in.views == c->views && c->out_layout == LFI_LAYOUT_PLANAR_RGB;
    if(a_in.k_pad > 64 || planar_views)
    {
        // more than one chunk of images (15×15 grids): blend_stdx — the band method with the chain's bytes fetched a second time;
        // one launch per 64 views (the accumulators of a wave hold 16 views)
        const int tiles_x = (a_in.width + lfi::P3_TPX - 1) / lfi::P3_TPX;
        const int n_tiles = tiles_x * a_in.out_rows;
        const dim3 grid(std::min(n_tiles, 2 * cu_count_of(c))), block(256);
        const int nch = (a_in.k_pad + lfi::P3_KC - 1) / lfi::P3_KC;
        const int reverse = next_sweep_direction(c);
        // planar views written directly: launch_blend has pointed a.views at the byte planes (stdx_writes_planar_views)
        const bool planar_out = planar_views;
        note_kernel(c, "blend_stdx<STD>");
        for(int v0 = a_in.v0; v0 < a_in.v1; v0 += 64)
        {
            KernelArgs a = a_in;
            a.v0 = v0;
            a.v1 = std::min(v0 + 64, a_in.v1);
#define LFI_SX_LAUNCH(N)                                                                                                                        \
    do                                                                                                                                          \
    {                                                                                                                                           \
        if(planar_out)                                                                                                                          \
            hipLaunchKernelGGL((lfi::blend_stdx<true, N, true>), grid, block, 0, stream_of(c), a, tiles_x, n_tiles, reverse);                   \
        else                                                                                                                                    \
            hipLaunchKernelGGL((lfi::blend_stdx<true, N>), grid, block, 0, stream_of(c), a, tiles_x, n_tiles, reverse);                         \
    } while(0)
            switch(nch)
            {
                case 1: LFI_SX_LAUNCH(1); break; // (planar views only)
                case 2: LFI_SX_LAUNCH(2); break;
                case 3: LFI_SX_LAUNCH(3); break;
                default: LFI_SX_LAUNCH(4); break;
            }
#undef LFI_SX_LAUNCH
        }
        return;
    }
    const KernelArgs &a = a_in;
    const int tiles_x = (a.width + 127) / 128;
    const int n_tiles = tiles_x * a.out_rows;
    const int passes = (a.v1 - a.v0 + 63) / 64;
    const dim3 grid(std::min(n_tiles, 2 * cu_count_of(c))), block(256);
    note_kernel(c, "blend_planar<STDF>");
    hipLaunchKernelGGL((lfi::blend_planar<2, true, true>), grid, block, 0, stream_of(c), a, tiles_x, n_tiles, passes, 0, next_sweep_direction(c));
}

template <int PXL, int MT>
void launch_std_mfma(const lfi_ctx *c, const KernelArgs &a, bool all_focus)
{
    const int tiles_x = (a.width + 32 * PXL - 1) / (32 * PXL);
    const int n_tiles = tiles_x * a.height;
    const int passes = (a.v1 - a.v0 + 32 * MT - 1) / (32 * MT);
    const int vpw = passes >= 4 ? 4 : (passes >= 2 ? 2 : 1);
    const int tiles_per_wg = 4 / vpw;
    const dim3 grid((n_tiles + tiles_per_wg - 1) / tiles_per_wg), block(256);
    note_kernel(c, "blend_std_mfma");
    if(all_focus)
        hipLaunchKernelGGL((lfi::blend_std_mfma<PXL, MT, true>), grid, block, 0, stream_of(c), a, tiles_x, n_tiles, passes, vpw);
    else
        hipLaunchKernelGGL((lfi::blend_std_mfma<PXL, MT, false>), grid, block, 0, stream_of(c), a, tiles_x, n_tiles, passes, vpw);
}

void launch_std_valu(const lfi_ctx *c, const KernelArgs &a, bool all_focus)
{
    note_kernel(c, "blend_std_valu");
    if(all_focus)
        hipLaunchKernelGGL((lfi::blend_std_valu<true, 16>), pixel_grid_of(c), dim3(256), 0, stream_of(c), a);
    else
        hipLaunchKernelGGL((lfi::blend_std_valu<false, 16>), pixel_grid_of(c), dim3(256), 0, stream_of(c), a);
}

void launch_std_vfma(const lfi_ctx *c, const KernelArgs &a, bool all_focus)
{
    note_kernel(c, "blend_std_vfma");
    if(all_focus)
        hipLaunchKernelGGL((lfi::blend_std_vfma<true>), pixel_grid_of(c), dim3(256), 0, stream_of(c), a);
    else
        hipLaunchKernelGGL((lfi::blend_std_vfma<false>), pixel_grid_of(c), dim3(256), 0, stream_of(c), a);
}

// first entry = default ("auto")
const Variant kTenVariants[] = {
    {"p3_rgba_nt", launch_p3_rgba, true, false, true, true},        // blend_p3 with the RGBA epilogue; blend_planar / blend_persist where it does not apply
    {"planar_m2_nt", launch_planar<true>, true, false, true, true}, // blend_persist where blend_planar does not apply
    {"persist_m2_nt", launch_persist<false, 2, true>, true, false, true},
    {"wave_m2_nt", launch_wave<false, 2, true>, true, false, true},
    {"direct_p1m2", launch_ten_direct<1, 2>, false, true}, // generic: any weights, pre-quantisation dump, per-batch rounding
};
const Variant kStdVariants[] = {
    {"filtered_m2_nt", launch_std_filtered_t<false>, false, false, true, true}, // blend_wave / blend_persist where it does not apply
    {"filtered_gather_once", launch_std_filtered_t<true>, false, false, true, true}, // the same with blend_afs for all-focus renders of 3–4 chunks
    {"wave_m2_nt", launch_wave<true, 2, true>, false, false, true},    // blend_persist where blend_wave does not apply
    {"persist_m2_nt", launch_persist<true, 2, true>, false, false, true},
    {"mfma_p1m2", launch_std_mfma<1, 2>, false, true}, // generic: pre-quantisation dump
    {"valu", launch_std_valu, false, true},             // the reference-shaped one-pixel-per-thread kernel: exactness anchor
    {"vfma", launch_std_vfma, false, true},             // the non-tensor wavefront kernel
};
const int kNumTenVariants = sizeof(kTenVariants) / sizeof(kTenVariants[0]);
const int kNumStdVariants = sizeof(kStdVariants) / sizeof(kStdVariants[0]);
int find_variant(const Variant *table, int n, const char *name)
{
    for(int i = 0; i < n; i++)
        if(std::strcmp(table[i].name, name) == 0)
            return i;
    return 0;
}
// the generic kernels: plain fp32 epilogue, any weights, pre-quantisation dump, per-batch rounding (TEN_WM)
const int kGenericTenVariant = find_variant(kTenVariants, kNumTenVariants, "direct_p1m2");
const int kGenericStdVariant = find_variant(kStdVariants, kNumStdVariants, "mfma_p1m2");

int launch_blend(lfi_ctx *c, int method, int all_focus, const KernelArgs &a);

int check_render_args(lfi_ctx *c, int method, int v0, int v1)
{
    if(!c)
        return LFI_EINVAL;
    if(!c->grid && !c->inputs_released)
        return fail(c, LFI_EINVAL, "lfi_set_grid has not been called");
    if(!c->have_params)
        return fail(c, LFI_EINVAL, "lfi_set_params has not been called");
    if(method != LFI_METHOD_STD && method != LFI_METHOD_TEN_WM)
        return fail(c, LFI_EINVAL, "The specified interpolation method does not exist!");
    if(v0 < 0 || v1 > c->views_n || v0 >= v1)
        return fail(c, LFI_EINVAL, "view range [v0, v1) outside [0, views)");
    if(!c->views)
        return fail(c, LFI_EINVAL, "the context has no view buffer (a failed allocation: lfi_set_params again)");
    return LFI_OK;
}

// Consecutive launches over the same inputs (the reference's 100-launch loop, a trajectory streamed in blocks, a focus sweep) walk
// the tiles in opposite directions: the input rows a launch read last are the ones the next launch reads first, so part of them
// is still in the 256 MB Infinity Cache (config 2: −7 %, profiles/r02_p3_alternate.txt).  Same work, same bytes requested; fewer
// of them come from HBM.  LFI_FLAG_SINGLE_SWEEP_DIRECTION turns it off (every launch ascending, as a cold launch behaves).
int next_sweep_direction(const lfi_ctx *c)
{
    if(c->flags & LFI_FLAG_SINGLE_SWEEP_DIRECTION)
        return 0;
    return int(c->sweep_launches++ & 1u);
}


// Make the planar copy of the inputs valid for a fixed-focus launch with the current parameters; returns false (and leaves the
// launch on the RGBA planes) when the copy may not be used: inputs the library cannot track, absurd offsets.
// tune: also make the copy's per-image phases fit the CURRENT integer offsets (every 128-byte run of a tile then IS one cache line) — a
// rebuild.  lfi_prepare / lfi_benchmark ask for it; launch_blend asks once the same offsets have been rendered LFI_RETUNE_AFTER times (the
// reference's 100-launch loop, a trajectory streamed at one focus), so a fixed-focus sweep — new offsets every render — never pays a
// rebuild per render.  What stale phases cost such a sweep (round 5, profiles/r05_pmc_sweep_summary.txt, fixed_focus_sweep_*.txt): the
// launch takes 6–10 % longer (config 2: 137.6 → 151.2 µs, config 5: 1.44 → 1.53 ms) — twice the L1 tag accesses (a lane's 16-byte piece
// straddles sectors, a run two lines) and 9 % more bytes from HBM (boundary lines shared with the neighbouring tiles) — against a rebuild
// of 1.8 launches' time (0.25 ms at config 2): per-image phases cannot survive a change of -f (every image's offset moves by its own
// amount), so for one render per parameter set the stale copy is the cheaper choice; `also.config*_fixed_focus_sweep_step` times it.
// LFI_PLANAR_ALIGN: what the runs of the offsets in use are aligned to by the per-image phases — 128 (round 4): a tile's 128-byte run of
// an image row is then exactly ONE cache line (no boundary sectors fetched with the neighbouring tiles: the launches of several chunks
// over-fetched 10–19 %, profiles/r04_pmc_traffic_summary.txt); 4 = round 3's dword alignment.  Power of two, ≤ 128.
#ifndef LFI_PLANAR_ALIGN
#define LFI_PLANAR_ALIGN 128
#endif
// are the planar copy's per-image phases the ones that align the runs of the offsets in use?  (then no tile shares a cache line with its
// neighbours: LFI_KFLAG_PLAIN_TILE_ORDER, lfi_device.hpp)
bool planar_phases_tuned(const lfi_ctx *c)
{
    if(!c->planar || (int)c->planar_phase.size() != c->n)
        return false;
    for(int g = 0; g < c->n; g++)
        if((c->h_focused[g].x + c->planar_padx + c->planar_phase[g]) & (LFI_PLANAR_ALIGN - 1))
            return false;
    return true;
}

bool ensure_planar(lfi_ctx *c, bool tune = false)
{
    if(!c->grid_tracked)
        return false;
    const int reach = std::max(std::max(std::abs(c->fo_min[0]), std::abs(c->fo_max[0])), 0);
    if(reach > 4 * c->width + 4096)
        return false;
    // a tile's 128-byte run starts up to `reach` pixels left of column 0 (left padding: reach, rounded up to whole dwords so that the
    // build's dword stores stay aligned) and, in the last tile of a row, ends up to `reach` pixels past the last tile's 128th pixel
    const bool valid = c->planar && c->planar_version == c->grid_version && c->planar_reach >= reach && (int)c->planar_phase.size() == c->n;
    auto tuned = [&] {
        for(int g = 0; g < c->n; g++)
            if((c->h_focused[g].x + c->planar_padx + c->planar_phase[g]) & (LFI_PLANAR_ALIGN - 1))
                return false;
        return true;
    };
    if(valid && (!tune || tuned()))
        return true;
    if(c->inputs_released)
        return valid; // nothing to rebuild from: the copy serves the offsets it was built for (stale phases cost a launch 6–10 %), or the render is refused
    // the copy in place fits and only SOME images were replaced since it was brought up to date (lfi_upload_image, a partial fill): their
    // planes only — 1/N of a rebuild per image
    if(c->planar && c->planar_version != 0 && c->planar_reach >= reach && (int)c->planar_phase.size() == c->n &&
       c->grid_full_version <= c->planar_version && (!tune || tuned()))
    {
        for(int g = 0; g < c->n;)
        {
            if(!image_changed_since(c, g, c->planar_version))
            {
                g++;
                continue;
            }
            int g1 = g + 1;
            while(g1 < c->n && image_changed_since(c, g1, c->planar_version))
                g1++;
            hipLaunchKernelGGL(lfi::planar_build, dim3((c->planar_pitch / 4 + 255) / 256, c->in_rows, g1 - g), dim3(256), 0, c->stream, c->grid,
                               c->planar, c->width, c->in_rows, c->planar_pitch, c->planar_padx, c->d_planar_phase, g);
            g = g1;
        }
        if(hipGetLastError() != hipSuccess)
            return false;
        c->planar_version = c->grid_version;
        return true;
    }
    // a copy that has to GROW (a sweep towards larger offsets) is padded for a quarter more than asked for: every growth is a rebuild, and
    // a reallocation of up to gigabytes if the planes no longer fit the allocation
    const int built_for = c->planar && reach > c->planar_reach ? reach + reach / 4 + 8 : std::max(reach, c->planar_reach);
    const int padx = (built_for + 3) / 4 * 4;
    const int tiles_w = (c->width + 127) / 128 * 128;
    constexpr int pitch_unit = LFI_PLANAR_ALIGN > 16 ? LFI_PLANAR_ALIGN : 16; // rows start on the alignment unit
    int pitch = (padx + (LFI_PLANAR_ALIGN - 1) + tiles_w + built_for + pitch_unit - 1) / pitch_unit * pitch_unit; // + the largest phase
    // (one more line per row where the pitch comes to a multiple of 2048 bytes — 8×8 @4K: exactly 4096 —, or an odd number of lines per row for
    // every shape: measured, ± 2 % either way by box: profiles/r04_planar_align_ab.txt)
#ifdef LFI_MEASUREMENT_BUILD // tools/plane_skew.py: do the 192 plane streams collide on HBM channels?  Extra bytes per plane row / per plane.
    static const int extra_pitch = [] { const char *e = std::getenv("LFI_PLANAR_EXTRA_PITCH"); return e ? std::atoi(e) : 0; }();
    pitch += extra_pitch / 16 * 16;
#endif
    // blend_p3 / blend_stdx address a row as row·pitch with a 24-bit multiply, and a lane's byte inside its octet of images (8 images
    // × 3 planes, plus the row and the run) with 32 bits
    if(c->in_rows >= (1 << 24) || pitch >= (1 << 24) || (uint64_t)26 * c->in_rows * pitch >= (1ull << 32))
        return false;
    const size_t bytes = (size_t)c->n * 3 * c->in_rows * pitch; // the rows this context holds (a row window: band + halo)
    if(bytes > c->planar_bytes) // (a larger allocation serves smaller planes too)
    {
        if(c->planar)
            (void)hipFree(c->planar);
        c->planar = nullptr;
        c->planar_bytes = 0;
        c->planar_version = 0;
        if(hipMalloc(reinterpret_cast<void **>(&c->planar), bytes) != hipSuccess)
        {
            (void)hipGetLastError(); // not enough memory for the copy: render from the RGBA planes
            c->planar = nullptr;
            return false;
        }
        c->planar_bytes = bytes;
    }
    if(!c->d_planar_phase && hipMalloc(reinterpret_cast<void **>(&c->d_planar_phase), sizeof(int32_t) * LFI_MAX_IMAGES) != hipSuccess)
    {
        (void)hipGetLastError();
        c->d_planar_phase = nullptr;
        return false;
    }
    // the phases: (offset + padx + phase) ≡ 0 mod LFI_PLANAR_ALIGN for the offsets in use now.  They travel through one of two page-locked buffers and a
    // stream-ordered copy (as lfi_set_params' blob does): kernels of earlier launches that read the old phases are ordered before the
    // copy, the build and every later launch after it, and the host never waits for the stream (round 3 copied from a pageable vector
    // and synchronised the stream inside lfi_render).
    if(!c->phase_staging)
    {
        if(hipHostMalloc(reinterpret_cast<void **>(&c->phase_staging), sizeof(int32_t) * 2 * LFI_MAX_IMAGES, hipHostMallocDefault) != hipSuccess ||
           hipEventCreateWithFlags(&c->ev_phase[0], hipEventDisableTiming) != hipSuccess ||
           hipEventCreateWithFlags(&c->ev_phase[1], hipEventDisableTiming) != hipSuccess)
        {
            (void)hipGetLastError();
            return false;
        }
    }
    else if(hipEventSynchronize(c->ev_phase[c->phase_slot]) != hipSuccess) // the copy out of this buffer, two rebuilds ago, has run
        return false;
    c->planar_phase.assign(c->n, 0);
    int32_t *staged = c->phase_staging + (size_t)c->phase_slot * LFI_MAX_IMAGES;
    for(int g = 0; g < c->n; g++)
        staged[g] = c->planar_phase[g] = (LFI_PLANAR_ALIGN - ((c->h_focused[g].x + padx) & (LFI_PLANAR_ALIGN - 1))) & (LFI_PLANAR_ALIGN - 1);
    c->planar_version = 0;
    if(hipMemcpyAsync(c->d_planar_phase, staged, sizeof(int32_t) * c->n, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
       hipEventRecord(c->ev_phase[c->phase_slot], c->stream) != hipSuccess)
        return false;
    c->phase_slot ^= 1;
    c->planar_padx = padx;
    c->planar_reach = built_for;
    c->planar_pitch = pitch;
    hipLaunchKernelGGL(lfi::planar_build, dim3((pitch / 4 + 255) / 256, c->in_rows, c->n), dim3(256), 0, c->stream, c->grid, c->planar,
                       c->width, c->in_rows, pitch, padx, c->d_planar_phase, 0);
    if(hipGetLastError() != hipSuccess)
        return false;
    c->planar_version = c->grid_version;
    return true;
}

// launch_blend's policy for the rebuild above.  A rebuild re-converts the whole copy (0.25 ms at config 2, ≈ 3 ms at config 5) for a 6–10 %
// gain per launch (14 µs at config 2, 0.09 ms at config 5): it pays for itself after ≈ 20–30 launches.  So a render only retunes once the same integer offsets have been rendered
// LFI_RETUNE_AFTER times (the reference's own loop is 100 launches over one parameter set, src/interpolator.cu:270-295; round 3 retuned
// at the third launch, a net loss for short runs); lfi_prepare and lfi_benchmark retune at once, outside any render.
constexpr unsigned LFI_RETUNE_AFTER = 32;
bool tune_planar_now(lfi_ctx *c)
{
    return c->launches_with_offsets++ >= LFI_RETUNE_AFTER;
}

// Would this launch read the planar copy of the inputs?
bool wants_planar(const lfi_ctx *c, int method, int all_focus, const KernelArgs &a)
{
    if(all_focus || a.prequant || !c->weights_scalable)
        return false;
    // (Rounds 2–3 kept launches that write many more views than they read images — 256 views from 64 images — off the copy: +6 % then.  With the
    // line-aligned copy it is the faster source there too: TEN_WM with RGBA views 2.04 against 2.17–2.28 ms, STD 3.6 against 7.2 ms for the
    // exact-fp32 kernel, profiles/r04_rgba_p3_ab.txt.)
    if(method == LFI_METHOD_TEN_WM)
        return kTenVariants[c->ten_variant].planar && !(c->flags & LFI_FLAG_TEN_ROUND_PER_BATCH);
    // STD: blend_planar<STDF> (one chunk of images) / blend_stdx (up to four) — weights for which their error bounds hold
    return method == LFI_METHOD_STD && kStdVariants[c->std_variant].planar && c->weights_sum_ok && a.k_pad <= 4 * lfi::P3_KC;
}

// planar view layout: does blend_p3 serve this launch?  (TEN_WM, fixed focus, weights in [0, 2) for the packed epilogue, no
// debug modes; the planar input copy must be usable)
bool wants_p3(const lfi_ctx *c, int method, int all_focus, const KernelArgs &a)
{
    // blend_p3's epilogue addresses a wave's 48 byte planes (16 views × 3 channels) with one 32-bit per-lane offset
    const bool planes_fit = (uint64_t)48 * (uint64_t)c->out_rows * (uint64_t)view_pitch(c) < (1ull << 32);
    return c->out_layout == LFI_LAYOUT_PLANAR_RGB && method == LFI_METHOD_TEN_WM && !all_focus && !a.prequant && c->weights_scalable &&
           !(c->flags & LFI_FLAG_TEN_ROUND_PER_BATCH) && kTenVariants[c->ten_variant].planar && a.k_pad <= 4 * lfi::P3_KC && planes_fit;
}

// planar view layout: does blend_stdx write the byte planes of this STD launch directly?  (fixed focus, up to 256 images, the default STD
// variants, weights for which the band method's bounds hold, the planar input copy usable; the addressing of a plane row: 32-bit)
bool stdx_writes_planar_views(const lfi_ctx *c, int method, int all_focus, const KernelArgs &a)
{
    const bool planes_fit = (uint64_t)48 * (uint64_t)c->out_rows * (uint64_t)view_pitch(c) < (1ull << 32);
    return c->out_layout == LFI_LAYOUT_PLANAR_RGB && method == LFI_METHOD_STD && c->std_variant <= 1 && planes_fit &&
           wants_planar(c, method, all_focus, a);
}

// planar view layout: does blend_persist write the byte planes of this all-focus TEN_WM render directly?  (the default variants — both end in
// blend_persist for all-focus renders —, weights in [0, 2) for the packed epilogue, no debug modes)
bool persist_writes_planar_views(const lfi_ctx *c, int method, int all_focus, const KernelArgs &a)
{
    // store_tile_planar (blend_core.hpp) addresses (3·view + channel)·plane bytes for up to 32 views of a pass with one 32-bit offset
    const bool planes_fit = (uint64_t)96 * (uint64_t)c->out_rows * (uint64_t)view_pitch(c) < (1ull << 32);
    return c->out_layout == LFI_LAYOUT_PLANAR_RGB && method == LFI_METHOD_TEN_WM && all_focus && c->ten_variant <= 1 && c->weights_scalable && !a.prequant &&
           !(c->flags & LFI_FLAG_TEN_ROUND_PER_BATCH) && planes_fit;
}

// … and blend_stdxa those of this all-focus STD render?  (the default STD variant, weights for which the band method's bounds hold)
bool stdxa_writes_planar_views(const lfi_ctx *c, int method, int all_focus, const KernelArgs &a)
{
    return c->out_layout == LFI_LAYOUT_PLANAR_RGB && method == LFI_METHOD_STD && all_focus && c->std_variant == 0 && c->weights_scalable && c->weights_sum_ok &&
           !a.prequant && a.k_pad <= 4 * 64 && (uint64_t)192 * (uint64_t)c->out_rows * (uint64_t)view_pitch(c) < (1ull << 32);
}

// Does a render with these arguments read the derived planar copy of the inputs?  ONE predicate for launch_blend's two branches,
// lfi_prepare and lfi_benchmark (round 2: lfi_prepare tested wants_planar only and built nothing for launches that blend_p3 serves
// beyond wants_planar's view limit — 256 views from 64 images — so the first render carried the build).
bool wants_derived_copy(const lfi_ctx *c, int method, int all_focus, const KernelArgs &a)
{
    // planar views: blend_p3 where it serves the launch; everything else — and every launch of the RGBA layout — goes through
    // launch_blend_rgba, whose kernels read the copy under wants_planar's conditions
    return (c->out_layout == LFI_LAYOUT_PLANAR_RGB && wants_p3(c, method, all_focus, a)) || wants_planar(c, method, all_focus, a);
}

void launch_p3(const lfi_ctx *c, const KernelArgs &a_in, bool rgba_out)
{
    const int tiles_x = (a_in.width + lfi::P3_TPX - 1) / lfi::P3_TPX;
    const int n_tiles = tiles_x * a_in.out_rows;
#ifdef LFI_MEASUREMENT_BUILD // LFI_P3_WGS = workgroups per CU in the grid (1 or 2): does a launch scale with the waves per CU?
    static const int wgs_env = [] { const char *e = std::getenv("LFI_P3_WGS"); return e ? std::atoi(e) : 2; }();
    const dim3 grid(std::min(n_tiles, wgs_env * cu_count_of(c))), block(256);
#else
    const dim3 grid(std::min(n_tiles, 2 * cu_count_of(c))), block(256);
#endif
    const int nch = (a_in.k_pad + lfi::P3_KC - 1) / lfi::P3_KC;
    note_kernel(c, rgba_out ? "blend_p3<TEN_WM,rgba>" : "blend_p3<TEN_WM>");
#ifdef LFI_MEASUREMENT_BUILD
    // measurement builds only (make HIPFLAGS+=-DLFI_MEASUREMENT_BUILD, tools/p3_ablate.py): where does a unit's time go?  The ablated
    // kernels write garbage by construction, so the production library does not contain them and reads no such environment variable.
    static const int ablate = [] {
        const char *e = std::getenv("LFI_P3_ABLATE");
        return e ? std::atoi(e) : 0;
    }();
    if(ablate >= 1 && ablate <= 3 && ((nch == 1 && a_in.v1 - a_in.v0 <= 256) || (nch == 4 && a_in.v1 - a_in.v0 <= 64)))
    {
        note_kernel(c, "blend_p3<ABLATION>");
        const int abl_passes = nch == 1 ? (a_in.v1 - a_in.v0 + 63) / 64 : 1;
#define LFI_P3_ABL(N, A) hipLaunchKernelGGL((lfi::blend_p3<true, N, A, (N == 1 ? 1 : 2), (N == 1 ? 4 : 1)>), grid, dim3(N == 1 ? 256 : 128), 0, stream_of(c), a_in, tiles_x, n_tiles, abl_passes, 0)
        if(nch == 1)
        {
            if(ablate == 1) LFI_P3_ABL(1, 1); else if(ablate == 2) LFI_P3_ABL(1, 2); else LFI_P3_ABL(1, 3);
        }
        else
        {
            if(ablate == 1) LFI_P3_ABL(4, 1); else if(ablate == 2) LFI_P3_ABL(4, 2); else LFI_P3_ABL(4, 3);
        }
#undef LFI_P3_ABL
        return;
    }
#endif
    const int reverse = next_sweep_direction(c);
    // Views per wave: 16 (four waves per workgroup, two per SIMD) when the launch is paced by its memory pipeline — one chunk of
    // images — and 32 (two waves per workgroup, one per SIMD, the pixel operand built once for two MFMAs) when several chunks make the
    // k-loop the pacer (15×15 grids: −13 % at 4K, profiles/r02_p3_vg.txt).  Measurement builds: LFI_P3_VG = 1 / 2 forces either (tools/p3_vg.py).
#ifdef LFI_MEASUREMENT_BUILD
    static const int vg_env = [] {
        const char *e = std::getenv("LFI_P3_VG");
        return e ? std::atoi(e) : 0;
    }();
#endif
    const dim3 block2(128);
    if(nch == 1)
    {
        // one chunk of images: every 64-view pass of a tile reads the same LDS-resident pixels — the inputs are fetched once per
        // 256 views (four passes: the weight fragments a wave keeps in registers)
        int launch_no = 0;
        for(int v0 = a_in.v0; v0 < a_in.v1; v0 += 256, launch_no++)
        {
            KernelArgs a = a_in;
            a.v0 = v0;
            a.v1 = std::min(v0 + 256, a_in.v1);
            const int passes = (a.v1 - a.v0 + 63) / 64;
            const int dir = reverse ^ (launch_no & 1);
#ifdef LFI_MEASUREMENT_BUILD
            if(vg_env == 2 && passes == 1)
            {
                hipLaunchKernelGGL((lfi::blend_p3<true, 1, 0, 2>), grid, block2, 0, stream_of(c), a, tiles_x, n_tiles, 1, dir);
                continue;
            }
#endif
            // (one chunk of images: the planar views' epilogue only — launch_p3_rgba keeps blend_planar / blend_persist there)
            if(passes == 1)
                hipLaunchKernelGGL((lfi::blend_p3<true, 1>), grid, block, 0, stream_of(c), a, tiles_x, n_tiles, 1, dir);
            else
                hipLaunchKernelGGL((lfi::blend_p3<true, 1, 0, 1, 4>), grid, block, 0, stream_of(c), a, tiles_x, n_tiles, passes, dir);
        }
        return;
    }
    // several chunks: one launch per 64 views
    for(int v0 = a_in.v0; v0 < a_in.v1; v0 += 64)
    {
        KernelArgs a = a_in;
        a.v0 = v0;
        a.v1 = std::min(v0 + 64, a_in.v1);
#define LFI_P3_LAUNCH(N)                                                                                                                        \
    do                                                                                                                                          \
    {                                                                                                                                           \
        if(rgba_out)                                                                                                                            \
            hipLaunchKernelGGL((lfi::blend_p3<true, N, 0, 2, 1, true>), grid, block2, 0, stream_of(c), a, tiles_x, n_tiles, 1, reverse);        \
        else                                                                                                                                    \
            hipLaunchKernelGGL((lfi::blend_p3<true, N, 0, 2>), grid, block2, 0, stream_of(c), a, tiles_x, n_tiles, 1, reverse);                 \
    } while(0)
#ifdef LFI_MEASUREMENT_BUILD
        if(vg_env == 1)
        {
            if(nch == 2) hipLaunchKernelGGL((lfi::blend_p3<true, 2, 0, 1>), grid, block, 0, stream_of(c), a, tiles_x, n_tiles, 1, reverse);
            else if(nch == 3) hipLaunchKernelGGL((lfi::blend_p3<true, 3, 0, 1>), grid, block, 0, stream_of(c), a, tiles_x, n_tiles, 1, reverse);
            else hipLaunchKernelGGL((lfi::blend_p3<true, 4, 0, 1>), grid, block, 0, stream_of(c), a, tiles_x, n_tiles, 1, reverse);
            continue;
        }
#endif
        switch(nch)
        {
            case 2: LFI_P3_LAUNCH(2); break;
            case 3: LFI_P3_LAUNCH(3); break;
            default: LFI_P3_LAUNCH(4); break;
        }
#undef LFI_P3_LAUNCH
    }
}

// planar_decided: launch_blend has already made the derived copy valid for this launch (ensure_planar succeeded there) — the copy is
// taken as it is, no second ensure_planar (the second call counted the launch twice and could be the one that rebuilds and fails,
// after launch_blend had committed to kernels that write byte planes)
int launch_blend_rgba(lfi_ctx *c, int method, int all_focus, const KernelArgs &a_in, bool planar_decided = false);

int launch_blend(lfi_ctx *c, int method, int all_focus, const KernelArgs &a_in)
{
    if(int rc = join_uploads(c))
        return rc;
    if(all_focus && a_in.map_index == 1)
        if(int rc = join_filter(c)) // the filtered map may still be in the making on the side stream
            return rc;
    if(method == LFI_METHOD_STD && a_in.k_pad > 64 && c->weights_scalable && c->weights_sum_ok && !a_in.prequant && !(a_in.flags & LFI_FLAG_STD_ANALYTIC_BAND))
    {
        // more than 64 images: the band's measured bound must hold on THIS device (lfi_band_probe.hpp: measured once per device)
        bool forced = false;
        if(int rc = std_band_forced_analytic(c, &forced))
            return rc;
        if(forced)
        {
            KernelArgs a = a_in;
            a.flags |= LFI_FLAG_STD_ANALYTIC_BAND;
            return launch_blend(c, method, all_focus, a);
        }
    }
    if(c->inputs_released && !(wants_derived_copy(c, method, all_focus, a_in) && ensure_planar(c)))
        return fail(c, LFI_EINVAL, "the RGBA inputs were released (lfi_release_inputs): only fixed-focus renders whose offsets the planar copy was built for "
                                   "are served (no all-focus render, debug mode, weights outside [0, 2) or larger offsets) - upload the images again");
    if(c->out_layout != LFI_LAYOUT_PLANAR_RGB)
        return launch_blend_rgba(c, method, all_focus, a_in);
    if(wants_p3(c, method, all_focus, a_in) && ensure_planar(c, tune_planar_now(c)))
    {
        KernelArgs a = a_in;
        a.planar = c->planar;
        a.planar_pitch = c->planar_pitch;
        a.planar_padx = c->planar_padx;
        a.planar_phase = c->d_planar_phase; // allocated by ensure_planar, possibly just now
        if(planar_phases_tuned(c))
            a.flags |= lfi::LFI_KFLAG_PLAIN_TILE_ORDER;
        launch_p3(c, a);
        LFI_HIP(c, hipGetLastError());
        return LFI_OK;
    }
    if(stdx_writes_planar_views(c, method, all_focus, a_in) && ensure_planar(c, tune_planar_now(c)))
        // fixed-focus STD: blend_stdx writes the byte planes itself (a.views are the context's planar views)
        return launch_blend_rgba(c, method, all_focus, a_in, true);
    if(persist_writes_planar_views(c, method, all_focus, a_in) || stdxa_writes_planar_views(c, method, all_focus, a_in))
        // all-focus renders: blend_persist / blend_stdxa write the byte planes themselves (quad transposes in their epilogues: store_tile_planar)
        return launch_blend_rgba(c, method, all_focus, a_in);
    // every other render (debug modes, non-default variants, weights outside [0, 2) or summing above 2) goes through the RGBA kernels into a
    // scratch copy of the views and is converted to byte planes afterwards
    const size_t need = rgba_out_plane_bytes(c) * c->views_n;
    if(c->rgba_scratch_bytes != need)
    {
        if(c->rgba_scratch)
            (void)hipFree(c->rgba_scratch);
        c->rgba_scratch = nullptr;
        c->rgba_scratch_bytes = 0;
        LFI_HIP(c, hipMalloc(reinterpret_cast<void **>(&c->rgba_scratch), need));
        c->rgba_scratch_bytes = need;
    }
    KernelArgs a = a_in;
    a.views = c->rgba_scratch;
    if(int rc = launch_blend_rgba(c, method, all_focus, a))
        return rc;
    const int pitch = view_pitch(c);
    hipLaunchKernelGGL(lfi::views_rgba_to_planar, dim3((pitch / 4 + 255) / 256, c->out_rows, a.v1 - a.v0), dim3(256), 0, c->stream,
                       reinterpret_cast<const uint32_t *>(c->rgba_scratch + rgba_out_plane_bytes(c) * a.v0), c->views + out_plane_bytes(c) * a.v0,
                       c->width, c->out_rows, pitch);
    LFI_HIP(c, hipGetLastError());
    return LFI_OK;
}

int launch_blend_rgba(lfi_ctx *c, int method, int all_focus, const KernelArgs &a_in, bool planar_decided)
{
    KernelArgs a = a_in;
    if(planar_decided || (wants_planar(c, method, all_focus, a) && ensure_planar(c, tune_planar_now(c))))
    {
        a.planar = c->planar;
        a.planar_pitch = c->planar_pitch;
        a.planar_padx = c->planar_padx;
        a.planar_phase = c->d_planar_phase; // allocated by ensure_planar, possibly just now
        if(planar_phases_tuned(c))
            a.flags |= lfi::LFI_KFLAG_PLAIN_TILE_ORDER;
    }
    if(all_focus)
        a.flags |= lfi::LFI_KFLAG_PLAIN_TILE_ORDER; // per-pixel gathers: no lines shared between neighbouring tiles by construction
    if(c->windowed)
    {
        // a row window is honoured by the persistent kernels only
        const bool ten = method == LFI_METHOD_TEN_WM;
        const Variant &v = ten ? kTenVariants[c->ten_variant] : kStdVariants[c->std_variant];
        if(a.prequant || !v.row_window || (c->flags & LFI_FLAG_TEN_ROUND_PER_BATCH) || (ten && v.packed_epilogue && !c->weights_scalable))
            return fail(c, LFI_EINVAL, "with a row window only renders with the default (persistent) kernels and weights in [0,2) are supported");
        if(all_focus)
        {
            // every image row an all-focus render of the band can sample must be held: (int)fma(f, offset.y, y) for f between the
            // ends of the focus range (the map decodes to focus + m/255·range), y in the band; ±1 for float rounding
            const float f_lo = std::min(c->focus, c->focus + c->range), f_hi = std::max(c->focus, c->focus + c->range);
            for(const lfi_float2 &o : c->h_offsets)
            {
                const double d_lo = std::min((double)f_lo * o.y, (double)f_hi * o.y), d_hi = std::max((double)f_lo * o.y, (double)f_hi * o.y);
                const int H = c->height;
                const int lo = std::min(std::max((int)std::floor(c->out_y0 + d_lo) - 1, 0), H - 1);
                const int hi = std::min(std::max((int)std::ceil(c->out_y0 + c->out_rows - 1 + d_hi) + 1, 0), H - 1);
                if(lo < c->in_y0 || hi >= c->in_y0 + c->in_rows)
                    return fail(c, LFI_EINVAL, "the input row window does not cover the rows an all-focus render of this band samples");
            }
        }
    }
    if(method == LFI_METHOD_TEN_WM)
    {
        // the generic kernel (direct_p1m2) serves what the packed-epilogue kernels cannot: the per-batch rounding debug
        // mode, pre-quantisation dumps, and weights outside [0, 2)
        int variant = c->ten_variant;
        if((c->flags & LFI_FLAG_TEN_ROUND_PER_BATCH) || (kTenVariants[variant].packed_epilogue && !c->weights_scalable) ||
           (a.prequant && !kTenVariants[variant].prequant))
            variant = kGenericTenVariant;
        kTenVariants[variant].launch(c, a, all_focus != 0);
    }
    else if(method == LFI_METHOD_STD)
    {
        int variant = c->std_variant;
        if(a.prequant && !kStdVariants[variant].prequant)
            variant = kGenericStdVariant;
        kStdVariants[variant].launch(c, a, all_focus != 0);
    }
    else
        // the reference throws here (src/interpolator.cu:289-290)
        return fail(c, LFI_EINVAL, "The specified interpolation method does not exist!");
    LFI_HIP(c, hipGetLastError());
    return LFI_OK;
}

// device pointer and pitch of view v as an RGBA plane of out_rows rows: the view itself, or (planar layout) its expansion into the
// context's one-plane staging buffer — valid until the next call, ordered on the context's stream
int rgba_plane_of_view(lfi_ctx *c, int v, const uint8_t **out)
{
    if(c->out_layout != LFI_LAYOUT_PLANAR_RGB)
    {
        *out = c->views + out_plane_bytes(c) * v;
        return LFI_OK;
    }
    const size_t need = rgba_out_plane_bytes(c);
    if(c->dl_plane_bytes != need)
    {
        if(c->dl_plane)
            (void)hipFree(c->dl_plane);
        c->dl_plane = nullptr;
        c->dl_plane_bytes = 0;
        LFI_HIP(c, hipMalloc(reinterpret_cast<void **>(&c->dl_plane), need));
        c->dl_plane_bytes = need;
    }
    hipLaunchKernelGGL(lfi::view_planar_to_rgba, dim3(((c->width + 3) / 4 + 255) / 256, c->out_rows), dim3(256), 0, c->stream,
                       c->views + out_plane_bytes(c) * v, reinterpret_cast<uint32_t *>(c->dl_plane), c->width, c->out_rows, view_pitch(c));
    LFI_HIP(c, hipGetLastError());
    *out = c->dl_plane;
    return LFI_OK;
}

} // namespace
