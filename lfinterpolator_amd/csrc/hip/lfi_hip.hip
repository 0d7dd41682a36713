// lfi_hip.hip — the C-ABI of include/lfi.h on HIP for gfx950: context, device memory, parameter upload and kernel
// launches.  This translation unit is the whole device-facing half of the reference's Interpolator
// (reference src/interpolator.cu:13-154, 194-316); the arithmetic that produces the parameters stays in the host code
// above the ABI (lfinterpolator_amd/csrc/host).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/lfi.h"
#include "blend_std.hpp"
#include "blend_ten.hpp"
#include "blend_ten_persist.hpp"
#include "blend_planar.hpp"
#include "blend_p3.hpp"
#include "blend_wave.hpp"
#include "focus_factored.hpp"
#include "lfi_device.hpp"
#include "quality.hpp"

using lfi::KernelArgs;

namespace {

thread_local std::string g_create_error;

struct Variant
{
    const char *name;
    void (*launch)(const lfi_ctx *, const KernelArgs &, bool all_focus);
    bool packed_epilogue; // TEN_WM: needs weights in [0,2) (×2^15 copy)
    bool prequant = false; // can dump pre-quantisation accumulators (the generic kernels only)
    bool row_window = false; // honours a row window (the persistent kernels)
    bool planar = false;     // reads the planar copy of the inputs when the launch qualifies (else its launcher falls back)
};
extern const Variant kTenVariants[];
extern const Variant kStdVariants[];
extern const int kNumTenVariants, kNumStdVariants;

} // namespace

struct lfi_ctx
{
    int device = 0;
    int cu_count = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_order = nullptr; // orders the work of the stream a caller switches away from before the stream it switches to
    // asynchronous uploads (lfi_upload_image_async) and downloads of lfi_render_stream: a copy stream
    hipStream_t copy_stream = nullptr;
    bool uploads_pending = false; // copies enqueued on copy_stream that the compute stream has not been ordered after yet
    hipEvent_t ev_uploads = nullptr;
    // side stream of the factored focus-map estimate (its small passes overlap the large ones), created on first use
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_pad = nullptr, ev_join = nullptr;
    int cols = 0, rows = 0, n = 0, width = 0, height = 0;
    // row window (lfi_set_row_window): input rows held / output rows rendered; the whole image by default
    int in_y0 = 0, in_rows = 0, out_y0 = 0, out_rows = 0;
    bool windowed = false;
    uint8_t *grid = nullptr;
    bool own_grid = false;
    size_t grid_bytes = 0;
    uint8_t *maps = nullptr;
    uint8_t *views = nullptr;
    bool own_views = false;
    size_t views_bytes = 0;
    int out_layout = LFI_LAYOUT_RGBA;  // device layout of the views (lfi_set_output_layout)
    uint8_t *rgba_scratch = nullptr;   // planar layout: RGBA planes of all views for the kernels that only write RGBA (converted after the launch)
    size_t rgba_scratch_bytes = 0;
    uint8_t *dl_plane = nullptr;       // planar layout: one RGBA plane that downloads expand a view into
    size_t dl_plane_bytes = 0;
    // parameter block
    bool have_params = false;
    int views_n = 0, k_pad = 0, v_pad = 0, n_focus_ids = 0;
    void *param_blob = nullptr; // one allocation holding all parameter arrays
    size_t blob_off_w16 = 0, blob_weights_bytes = 0; // the four weight arrays inside the blob (what lfi_render_stream replaces per block)
    // lfi_render_stream: page-locked staging for two blocks' weight arrays, a second set of views, events
    uint8_t *stream_staging[2] = {nullptr, nullptr};
    size_t stream_staging_bytes = 0;
    uint8_t *views2 = nullptr;
    size_t views2_bytes = 0;
    hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_rendered[2] = {nullptr, nullptr}, ev_d2h[2] = {nullptr, nullptr};
    lfi::QualitySums *quality_sums = nullptr; // lfi_compare_view
    uint8_t *quality_ref = nullptr;
    size_t quality_ref_bytes = 0;
    lfi_int2 *d_focused = nullptr;
    lfi_float2 *d_offsets = nullptr;
    uint16_t *d_w16 = nullptr, *d_w16s = nullptr;
    bool weights_scalable = false; // every weight finite and in [0, 2): the ×2^15 copy is exact and the packed epilogue valid
    bool weights_sum_ok = false;   // … and every view's weights sum to at most 2: blend_planar<STDF>'s error bounds hold (sums < 512)
    float *d_w32 = nullptr, *d_w32t = nullptr;
    int32_t *d_ids = nullptr;
    float focus = 0, range = 0;
    int radius[2] = {1, 1};
    int fo_min[2] = {0, 0}, fo_max[2] = {0, 0}; // bounds of the integer offsets
    uint32_t flags = 0;
    float *prequant = nullptr;
    std::vector<lfi_float2> h_focus_offsets; // offsets of the focus_map_ids images (host copy: sizes the padded planes)
    std::vector<lfi_float2> h_offsets;       // offsets of all images (host copy: row-window coverage checks of all-focus renders)
    // planar copy of the inputs for blend_planar (built on demand; valid while planar_version == grid_version)
    uint8_t *planar = nullptr;
    size_t planar_bytes = 0;
    int planar_pitch = 0, planar_padx = 0;
    uint64_t grid_version = 1, planar_version = 0;
    bool grid_tracked = true; // every write to the planes goes through this library (or is announced by lfi_grid_modified)
    void *focus_ws = nullptr; // workspace of the factored focus-map estimate (plan, E, K), allocated on first use
    size_t focus_ws_bytes = 0;
    int ten_variant = 0, std_variant = 0, focus_variant = 0;
    mutable const char *last_kernel = ""; // the blend kernel the last render launched (lfi_last_kernel_name)
    mutable unsigned sweep_launches = 0;  // blend_p3 / blend_planar alternate their sweep direction from launch to launch
    float derived_build_ms = 0.0f;        // duration of the last planar_build (measured by lfi_prepare only)
    std::string err;
};

namespace {

// The context's own views in the planar layout live in UNCACHED device memory: they are write-only for the renders (full 128-byte
// lines, non-temporal stores), and planes that bypass the caches leave more of the Infinity Cache to the inputs the next launch
// re-reads (tools/views_mtype.py, config 2: 150 µs against 156 µs per launch).  RGBA views stay in ordinary memory (the STD band
// epilogue patches single bytes behind its dword stores).  LFI_VIEWS_MEMORY=default|uncached|finegrained overrides (experiments).
hipError_t alloc_views(uint8_t **out, size_t bytes, bool planar_layout)
{
    static const int forced = [] {
        const char *e = std::getenv("LFI_VIEWS_MEMORY");
        return !e ? -1 : (std::strcmp(e, "uncached") == 0 ? 1 : (std::strcmp(e, "finegrained") == 0 ? 2 : 0));
    }();
    const int kind = forced >= 0 ? forced : (planar_layout ? 1 : 0);
    if(kind == 1)
        return hipExtMallocWithFlags(reinterpret_cast<void **>(out), bytes, hipDeviceMallocUncached);
    if(kind == 2)
        return hipExtMallocWithFlags(reinterpret_cast<void **>(out), bytes, hipDeviceMallocFinegrained);
    return hipMalloc(reinterpret_cast<void **>(out), bytes);
}

int fail(lfi_ctx *ctx, int code, const std::string &msg)
{
    if(ctx)
        ctx->err = msg;
    else
        g_create_error = msg;
    return code;
}

#define LFI_HIP(ctx, call)                                                                                            \
    do                                                                                                                \
    {                                                                                                                 \
        hipError_t e_ = (call);                                                                                       \
        if(e_ != hipSuccess)                                                                                          \
            return fail(ctx, e_ == hipErrorOutOfMemory ? LFI_ENOMEM : LFI_EHIP,                                       \
                        std::string(#call) + ": " + hipGetErrorString(e_));                                           \
    } while(0)

int bind(lfi_ctx *ctx)
{
    LFI_HIP(ctx, hipSetDevice(ctx->device));
    return LFI_OK;
}

// Order everything enqueued on the compute stream from now on after the asynchronous uploads issued so far (no host wait).
int join_uploads(lfi_ctx *c)
{
    if(!c->uploads_pending)
        return LFI_OK;
    LFI_HIP(c, hipEventRecord(c->ev_uploads, c->copy_stream));
    LFI_HIP(c, hipStreamWaitEvent(c->stream, c->ev_uploads, 0));
    c->uploads_pending = false;
    return LFI_OK;
}

int ensure_copy_stream(lfi_ctx *c)
{
    if(c->copy_stream)
        return LFI_OK;
    LFI_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    LFI_HIP(c, hipEventCreateWithFlags(&c->ev_uploads, hipEventDisableTiming));
    return LFI_OK;
}

size_t plane_bytes(const lfi_ctx *c) // a whole-image plane (focus maps; inputs and outputs without a row window)
{
    return (size_t)c->width * c->height * 4;
}

size_t in_plane_bytes(const lfi_ctx *c)
{
    return (size_t)c->width * c->in_rows * 4;
}

// planar view layout: bytes per row of a byte plane — a multiple of 16 so that every 8-byte store of blend_p3 is aligned and stays
// inside its row whatever the width
int view_pitch(const lfi_ctx *c)
{
    return (c->width + 15) / 16 * 16;
}

size_t rgba_out_plane_bytes(const lfi_ctx *c)
{
    return (size_t)c->width * c->out_rows * 4;
}

size_t out_plane_bytes(const lfi_ctx *c) // one view as stored on the device
{
    if(c->out_layout == LFI_LAYOUT_PLANAR_RGB)
        return (size_t)3 * c->out_rows * view_pitch(c);
    return rgba_out_plane_bytes(c);
}

KernelArgs make_args(const lfi_ctx *c, int v0, int v1, int all_focus_method)
{
    KernelArgs a{};
    a.grid = c->grid;
    a.views = c->views;
    a.maps = c->maps;
    a.focused = c->d_focused;
    a.offsets = c->d_offsets;
    a.w16 = c->d_w16;
    a.w16s = c->d_w16s;
    a.w32 = c->d_w32;
    a.w32t = c->d_w32t;
    a.focus_ids = c->d_ids;
    a.prequant = nullptr;
    a.prequant_view = -1;
    a.width = c->width;
    a.height = c->height;
    a.in_y0 = c->in_y0;
    a.in_rows = c->in_rows;
    a.out_y0 = c->out_y0;
    a.out_rows = c->out_rows;
    a.map_y0 = 0;
    a.map_rows = c->height;
    a.n_images = c->n;
    a.k_pad = c->k_pad;
    a.v_pad = c->v_pad;
    a.v0 = v0;
    a.v1 = v1;
    a.n_focus_ids = c->n_focus_ids;
    a.planar = nullptr; // set by launch_blend when the copy is valid for this launch
    // blend_planar<STDF>: chain bound N·2^-16 (half an ulp below 512 per fmaf: arithmetic) + MFMA accumulation bound N·2^-17 (a
    // quarter ulp per addend: MEASURED on gfx950 — chains of v_mfma_f32_32x32x16_f16 on operands built to expose alignment
    // truncation stay within 0.086 ulp per addend, tests/test_gpu_parity.py::test_mfma_f16_accumulation_error_bound asserts the
    // quarter ulp used here) + 2^-11 of margin
    a.std_band = float(c->n) * (0x1p-16f + 0x1p-17f) + 0x1p-11f;
    a.planar_pitch = c->planar_pitch;
    a.planar_padx = c->planar_padx;
    a.views_pitch = view_pitch(c);
    a.fo_min_x = c->fo_min[0];
    a.fo_max_x = c->fo_max[0];
    a.fo_min_y = c->fo_min[1];
    a.fo_max_y = c->fo_max[1];
    a.radius_x = c->radius[0];
    a.radius_y = c->radius[1];
    // the reference reads map 1 in Standard::process and map 0 in Tensors::process (src/kernels.cu:326 vs :430): reproduced by
    // default; LFI_FLAG_UNIFIED_FOCUS_MAP makes both read the filtered map
    a.map_index = 1;
    if(all_focus_method == LFI_METHOD_TEN_WM && !(c->flags & LFI_FLAG_UNIFIED_FOCUS_MAP))
        a.map_index = 0;
    a.focus = c->focus;
    a.range = c->range;
    a.flags = c->flags;
    return a;
}

dim3 pixel_grid(const lfi_ctx *c)
{
    return dim3((c->width + 63) / 64, (c->height + 3) / 4, 1);
}

hipStream_t stream_of(const lfi_ctx *c);
void note_kernel(const lfi_ctx *c, const char *name);
int next_sweep_direction(const lfi_ctx *c);
uint32_t flags_of(const lfi_ctx *c);
dim3 pixel_grid_of(const lfi_ctx *c);
int cu_count_of(const lfi_ctx *c);

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
    note_kernel(c, "blend_ten_direct");
    if constexpr(PXL == 1 && MT == 2)
    {
        if(flags_of(c) & LFI_FLAG_TEN_ROUND_PER_BATCH)
        {
            if(all_focus)
                hipLaunchKernelGGL((lfi::blend_ten_direct<PXL, MT, true, true>), grid, block, 0, st, a, tiles_x, n_tiles, passes, vpw);
            else
                hipLaunchKernelGGL((lfi::blend_ten_direct<PXL, MT, false, true>), grid, block, 0, st, a, tiles_x, n_tiles, passes, vpw);
            return;
        }
    }
    if(all_focus)
        hipLaunchKernelGGL((lfi::blend_ten_direct<PXL, MT, true, false>), grid, block, 0, st, a, tiles_x, n_tiles, passes, vpw);
    else
        hipLaunchKernelGGL((lfi::blend_ten_direct<PXL, MT, false, false>), grid, block, 0, st, a, tiles_x, n_tiles, passes, vpw);
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
void launch_std_filtered(const lfi_ctx *c, const KernelArgs &a, bool all_focus)
{
    if(!a.planar || all_focus || a.k_pad > 64)
    {
        launch_wave<true, 2, true>(c, a, all_focus);
        return;
    }
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
    {"planar_m2_nt", launch_planar<true>, true, false, true, true}, // blend_persist where blend_planar does not apply
    {"persist_m2_nt", launch_persist<false, 2, true>, true, false, true},
    {"wave_m2_nt", launch_wave<false, 2, true>, true, false, true},
    {"direct_p1m2", launch_ten_direct<1, 2>, false, true}, // generic: any weights, pre-quantisation dump, per-batch rounding
};
const Variant kStdVariants[] = {
    {"filtered_m2_nt", launch_std_filtered, false, false, true, true}, // blend_wave / blend_persist where it does not apply
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
    if(!c->grid)
        return fail(c, LFI_EINVAL, "lfi_set_grid has not been called");
    if(!c->have_params)
        return fail(c, LFI_EINVAL, "lfi_set_params has not been called");
    if(method != LFI_METHOD_STD && method != LFI_METHOD_TEN_WM)
        return fail(c, LFI_EINVAL, "The specified interpolation method does not exist!");
    if(v0 < 0 || v1 > c->views_n || v0 >= v1)
        return fail(c, LFI_EINVAL, "view range [v0, v1) outside [0, views)");
    return LFI_OK;
}

hipStream_t stream_of(const lfi_ctx *c) { return c->stream; }
void note_kernel(const lfi_ctx *c, const char *name) { c->last_kernel = name; }
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
uint32_t flags_of(const lfi_ctx *c) { return c->flags; }
dim3 pixel_grid_of(const lfi_ctx *c) { return pixel_grid(c); }
int cu_count_of(const lfi_ctx *c) { return c->cu_count; }

// Make the planar copy of the inputs valid for a fixed-focus launch with the current parameters; returns false (and leaves the
// launch on the RGBA planes) when the copy may not be used: inputs the library cannot track, absurd offsets.
bool ensure_planar(lfi_ctx *c)
{
    if(!c->grid_tracked)
        return false;
    const int reach = std::max(std::max(std::abs(c->fo_min[0]), std::abs(c->fo_max[0])), 0);
    if(reach > 4 * c->width + 4096)
        return false;
    // a tile's 128-byte run may start `reach` pixels left of column 0 and, for the ragged last tile of a row, end 127 pixels past
    // the row plus `reach`: pad by reach + 128 on both sides
    const int need = (reach + 128 + 3) / 4 * 4;
    if(c->planar && c->planar_version == c->grid_version && c->planar_padx >= need)
        return true;
    const int padx = std::max(need, c->planar_padx);
    const int pitch = (c->width + 2 * padx + 15) / 16 * 16;
    // blend_p3 addresses a row as (shift·rows + row)·pitch with 24-bit multiplies, and a lane's byte inside its octet of images (8
    // images × 12 planes) with 32 bits
    if(c->in_rows >= (1 << 22) || pitch >= (1 << 24) || (uint64_t)100 * c->in_rows * pitch >= (1ull << 32))
        return false;
    const size_t bytes = (size_t)c->n * 12 * c->in_rows * pitch; // the rows this context holds (a row window: band + halo)
    if(bytes != c->planar_bytes)
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
    c->planar_padx = padx;
    c->planar_pitch = pitch;
    hipLaunchKernelGGL(lfi::planar_build, dim3((pitch / 4 + 255) / 256, c->in_rows, c->n), dim3(256), 0, c->stream, c->grid, c->planar,
                       c->width, c->in_rows, pitch, padx);
    if(hipGetLastError() != hipSuccess)
        return false;
    c->planar_version = c->grid_version;
    return true;
}

// Would this launch read the planar copy of the inputs?  It pays where reads are a large share of the traffic: not for launches
// that write many more views than they read images (config 4 on one GPU, 256 views from 64 images: +6 % — the byte-wise operand
// assembly repeats per view pass).
bool wants_planar(const lfi_ctx *c, int method, int all_focus, const KernelArgs &a)
{
    if(all_focus || a.prequant || !c->weights_scalable || a.v1 - a.v0 > std::max(c->n, 64))
        return false;
    if(method == LFI_METHOD_TEN_WM)
        return kTenVariants[c->ten_variant].planar && !(c->flags & LFI_FLAG_TEN_ROUND_PER_BATCH);
    // STD: blend_planar<STDF> — one chunk, and weights for which its error bounds hold
    return method == LFI_METHOD_STD && kStdVariants[c->std_variant].planar && c->weights_sum_ok && a.k_pad <= 64;
}

// planar view layout: does blend_p3 serve this launch?  (TEN_WM, fixed focus, weights in [0, 2) for the packed epilogue, no
// debug modes; the planar input copy must be usable)
bool wants_p3(const lfi_ctx *c, int method, int all_focus, const KernelArgs &a)
{
    return c->out_layout == LFI_LAYOUT_PLANAR_RGB && method == LFI_METHOD_TEN_WM && !all_focus && !a.prequant && c->weights_scalable &&
           !(c->flags & LFI_FLAG_TEN_ROUND_PER_BATCH) && kTenVariants[c->ten_variant].planar && a.k_pad <= 4 * lfi::P3_KC;
}

void launch_p3(const lfi_ctx *c, const KernelArgs &a_in)
{
    const int tiles_x = (a_in.width + lfi::P3_TPX - 1) / lfi::P3_TPX;
    const int n_tiles = tiles_x * a_in.out_rows;
    const dim3 grid(std::min(n_tiles, 2 * cu_count_of(c))), block(256);
    const int nch = (a_in.k_pad + lfi::P3_KC - 1) / lfi::P3_KC;
    note_kernel(c, "blend_p3<TEN_WM>");
    // measurement builds (tools/p3_ablate.py): where does a unit's time go?  Never set in production; outputs are garbage.
    static const int ablate = [] {
        const char *e = std::getenv("LFI_P3_ABLATE");
        return e ? std::atoi(e) : 0;
    }();
    if(ablate >= 1 && ablate <= 3 && (nch == 1 || (nch == 4 && a_in.v1 - a_in.v0 <= 64)))
    {
        note_kernel(c, "blend_p3<ABLATION>");
        const int abl_passes = nch == 1 ? (a_in.v1 - a_in.v0 + 63) / 64 : 1;
#define LFI_P3_ABL(N, A) hipLaunchKernelGGL((lfi::blend_p3<true, N, A, (N == 1 ? 1 : 2)>), grid, dim3(N == 1 ? 256 : 128), 0, stream_of(c), a_in, tiles_x, n_tiles, abl_passes, 0)
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
    const int reverse = next_sweep_direction(c);
    // Views per wave: 16 (four waves per workgroup, two per SIMD) when the launch is paced by its memory pipeline — one chunk of
    // images — and 32 (two waves per workgroup, one per SIMD, the pixel operand built once for two MFMAs) when several chunks make the
    // k-loop the pacer (15×15 grids: −13 % at 4K, profiles/r02_p3_vg.txt).  LFI_P3_VG = 1 / 2 forces either (measurements only).
    static const int vg_env = [] {
        const char *e = std::getenv("LFI_P3_VG");
        return e ? std::atoi(e) : 0;
    }();
    const dim3 block2(128);
    if(nch == 1)
    {
        // one chunk of images: every 64-view pass of a tile reads the same LDS-resident pixels (inputs fetched once per launch)
        const int passes = (a_in.v1 - a_in.v0 + 63) / 64;
        if(vg_env == 2)
            hipLaunchKernelGGL((lfi::blend_p3<true, 1, 0, 2>), grid, block2, 0, stream_of(c), a_in, tiles_x, n_tiles, passes, reverse);
        else
            hipLaunchKernelGGL((lfi::blend_p3<true, 1>), grid, block, 0, stream_of(c), a_in, tiles_x, n_tiles, passes, reverse);
        return;
    }
    // several chunks: one launch per 64 views
    for(int v0 = a_in.v0; v0 < a_in.v1; v0 += 64)
    {
        KernelArgs a = a_in;
        a.v0 = v0;
        a.v1 = std::min(v0 + 64, a_in.v1);
#define LFI_P3_LAUNCH(N)                                                                                                                        \
    if(vg_env == 1)                                                                                                                             \
        hipLaunchKernelGGL((lfi::blend_p3<true, N, 0, 1>), grid, block, 0, stream_of(c), a, tiles_x, n_tiles, 1, reverse);                        \
    else                                                                                                                                        \
        hipLaunchKernelGGL((lfi::blend_p3<true, N, 0, 2>), grid, block2, 0, stream_of(c), a, tiles_x, n_tiles, 1, reverse)
        switch(nch)
        {
            case 2: LFI_P3_LAUNCH(2); break;
            case 3: LFI_P3_LAUNCH(3); break;
            default: LFI_P3_LAUNCH(4); break;
        }
#undef LFI_P3_LAUNCH
    }
}

int launch_blend_rgba(lfi_ctx *c, int method, int all_focus, const KernelArgs &a_in);

int launch_blend(lfi_ctx *c, int method, int all_focus, const KernelArgs &a_in)
{
    if(int rc = join_uploads(c))
        return rc;
    if(c->out_layout != LFI_LAYOUT_PLANAR_RGB)
        return launch_blend_rgba(c, method, all_focus, a_in);
    if(wants_p3(c, method, all_focus, a_in) && ensure_planar(c))
    {
        KernelArgs a = a_in;
        a.planar = c->planar;
        a.planar_pitch = c->planar_pitch;
        a.planar_padx = c->planar_padx;
        launch_p3(c, a);
        LFI_HIP(c, hipGetLastError());
        return LFI_OK;
    }
    // every other render (STD, all-focus, debug modes, weights outside [0, 2)) goes through the RGBA kernels into a scratch copy of
    // the views and is converted to byte planes afterwards
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

int launch_blend_rgba(lfi_ctx *c, int method, int all_focus, const KernelArgs &a_in)
{
    KernelArgs a = a_in;
    if(wants_planar(c, method, all_focus, a) && ensure_planar(c))
    {
        a.planar = c->planar;
        a.planar_pitch = c->planar_pitch;
        a.planar_padx = c->planar_padx;
    }
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

// The four device forms of a block of `rows` weight rows: fp16 as given, ×2^15 (exact; valid iff every weight is finite and in
// [0, 2)), f32, and f32 transposed — written at base + 0 / off_w16s / off_w32 / off_w32t (the region must be zero-initialised:
// padding rows and images stay zero).  *scalable / *sums_ok: the dispatch conditions lfi_set_params records.
void fill_weight_arrays(const uint16_t *weights_fp16, int rows, int n, int k_pad, int v_pad, uint8_t *base, size_t off_w16s, size_t off_w32, size_t off_w32t,
                        bool *scalable_out, bool *sums_ok_out)
{
    uint16_t *w16 = reinterpret_cast<uint16_t *>(base);
    uint16_t *w16s = reinterpret_cast<uint16_t *>(base + off_w16s);
    float *w32 = reinterpret_cast<float *>(base + off_w32);
    float *w32t = reinterpret_cast<float *>(base + off_w32t);
    bool scalable = true;
    for(int v = 0; v < rows; v++)
        for(int g = 0; g < n; g++)
        {
            const uint16_t h = weights_fp16[(size_t)v * n + g];
            const float f = static_cast<float>(__builtin_bit_cast(_Float16, h)); // half → float is exact
            w16[(size_t)v * k_pad + g] = h;
            // × 2^15 is exact in fp16 for every finite weight in [0, 2) (subnormals become normal, 1.999 → 65472)
            if(!(f >= 0.0f && f < 2.0f))
                scalable = false;
            else
                w16s[(size_t)v * k_pad + g] = __builtin_bit_cast(uint16_t, static_cast<_Float16>(f * 32768.0f));
            w32[(size_t)v * k_pad + g] = f;
            w32t[(size_t)g * v_pad + v] = f;
        }
    bool sums_ok = scalable;
    for(int v = 0; v < rows && sums_ok; v++)
    {
        double sum = 0;
        for(int g = 0; g < n; g++)
            sum += w32[(size_t)v * k_pad + g];
        sums_ok = sum <= 2.0;
    }
    *scalable_out = scalable;
    *sums_ok_out = sums_ok;
}

void free_params(lfi_ctx *c)
{
    if(c->param_blob)
        (void)hipFree(c->param_blob);
    c->param_blob = nullptr;
    c->have_params = false;
}

void free_views(lfi_ctx *c)
{
    if(c->own_views && c->views)
        (void)hipFree(c->views);
    c->views = nullptr;
    c->own_views = false;
    c->views_bytes = 0;
    if(c->rgba_scratch)
        (void)hipFree(c->rgba_scratch);
    c->rgba_scratch = nullptr;
    c->rgba_scratch_bytes = 0;
    if(c->dl_plane)
        (void)hipFree(c->dl_plane);
    c->dl_plane = nullptr;
    c->dl_plane_bytes = 0;
    if(c->views2)
        (void)hipFree(c->views2);
    c->views2 = nullptr;
    c->views2_bytes = 0;
    if(c->quality_ref)
        (void)hipFree(c->quality_ref);
    c->quality_ref = nullptr;
    c->quality_ref_bytes = 0;
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

void free_grid(lfi_ctx *c)
{
    if(c->own_grid && c->grid)
        (void)hipFree(c->grid);
    c->grid = nullptr;
    c->own_grid = false;
    c->grid_bytes = 0;
    if(c->maps)
        (void)hipFree(c->maps);
    c->maps = nullptr;
    if(c->prequant)
        (void)hipFree(c->prequant);
    c->prequant = nullptr;
    if(c->focus_ws)
        (void)hipFree(c->focus_ws);
    c->focus_ws = nullptr;
    c->focus_ws_bytes = 0;
    if(c->planar)
        (void)hipFree(c->planar);
    c->planar = nullptr;
    c->planar_bytes = 0;
    c->planar_version = 0;
}

// the factored estimate (focus_factored.hpp): carve the workspace, then plan → pad → E → exact keys → pick.
// Returns LFI_OK with *done = false when the padded planes would be unreasonably large (the caller takes another variant).
int launch_focus_factored(lfi_ctx *ctx, const KernelArgs &a, bool *done)
{
    *done = false;
    const int W = ctx->width, H = ctx->height, rx = ctx->radius[0], ry = ctx->radius[1];
    lfi::FocusWork w{};
    w.We_p = (W + 2 * rx + 255) / 256 * 256;
    w.He_p = (H + 2 * ry + 3) / 4 * 4;
    // largest |shift| any candidate gives any sampled image: candidates are monotone in i, so the ends bound them
    const float step = ctx->range / 31.0f;
    const double fmax = std::max(std::fabs((double)ctx->focus), std::fabs((double)std::fmaf(step, 31.0f, ctx->focus)));
    double ox = 0, oy = 0;
    for(const lfi_float2 &o : ctx->h_focus_offsets)
    {
        ox = std::max(ox, std::fabs((double)o.x));
        oy = std::max(oy, std::fabs((double)o.y));
    }
    if(!(fmax * ox < 1e6 && fmax * oy < 1e6))
        return LFI_OK;
    const int Sx = (int)std::ceil(fmax * ox) + 1, Sy = (int)std::ceil(fmax * oy) + 1; // ≥ |floor(δ)| and ≥ |floor(δ)+1|
    w.Px = Sx + rx;
    w.Py = Sy + ry;
    w.Wp = (w.Px + std::max(W + Sx + rx, w.We_p - rx + Sx) + 3) / 4 * 4;
    w.Hp = w.Py + std::max(H + Sy + ry, w.He_p - ry + Sy);
    const size_t pad_bytes = sizeof(uint32_t) * (size_t)ctx->n_focus_ids * w.Hp * w.Wp;
    if(pad_bytes > ((size_t)16 << 30))
        return LFI_OK;
    size_t at = 0;
    auto carve = [&](size_t bytes) {
        const size_t here = at;
        at += (bytes + 255) / 256 * 256;
        return here;
    };
    const size_t o_shifts = carve(sizeof(int32_t) * 4 * lfi::FOCUS_STEPS * lfi::FOCUS_MAX_IDS);
    const size_t o_badx = carve(sizeof(uint32_t) * W), o_bady = carve(sizeof(uint32_t) * H);
    const size_t o_cols = carve(sizeof(uint16_t) * lfi::FOCUS_STEPS * W), o_rows = carve(sizeof(uint16_t) * lfi::FOCUS_STEPS * H);
    const size_t o_ncols = carve(sizeof(int32_t) * lfi::FOCUS_STEPS), o_nrows = carve(sizeof(int32_t) * lfi::FOCUS_STEPS);
    const size_t o_prefix = carve(sizeof(uint32_t) * 3 * 33);
    const size_t o_rowbase = carve(sizeof(uint32_t) * H), o_colbase = carve(sizeof(uint32_t) * (W + 1));
    // line buffers for 4× the typical number of flagged rows / columns (three bands of r per candidate ≈ 0.03·H each);
    // anything beyond takes the tap-by-tap path
    w.R_cap = 4 * H;
    w.C_cap = 4 * W;
    const size_t o_Er = carve(sizeof(uint16_t) * (size_t)w.R_cap * 3 * w.We_p);
    const size_t o_Ec = carve(sizeof(uint16_t) * (size_t)w.C_cap * 3 * w.He_p);
    const size_t o_E = carve(sizeof(uint16_t) * lfi::FOCUS_STEPS * (size_t)w.He_p * w.We_p);
    const size_t o_K = carve(sizeof(uint16_t) * lfi::FOCUS_STEPS * (size_t)H * W);
    const size_t o_deltas = carve(sizeof(int64_t) * lfi::FOCUS_STEPS * lfi::FOCUS_MAX_IDS);
    const size_t o_pad = carve(pad_bytes);
    if(ctx->focus_ws_bytes != at)
    {
        if(ctx->focus_ws)
            (void)hipFree(ctx->focus_ws);
        ctx->focus_ws = nullptr;
        ctx->focus_ws_bytes = 0;
        LFI_HIP(ctx, hipMalloc(&ctx->focus_ws, at));
        ctx->focus_ws_bytes = at;
    }
    uint8_t *base = static_cast<uint8_t *>(ctx->focus_ws);
    w.shifts = reinterpret_cast<int32_t *>(base + o_shifts);
    w.badx = reinterpret_cast<uint32_t *>(base + o_badx);
    w.bady = reinterpret_cast<uint32_t *>(base + o_bady);
    w.cols = reinterpret_cast<uint16_t *>(base + o_cols);
    w.rows = reinterpret_cast<uint16_t *>(base + o_rows);
    w.ncols = reinterpret_cast<int32_t *>(base + o_ncols);
    w.nrows = reinterpret_cast<int32_t *>(base + o_nrows);
    w.prefix = reinterpret_cast<uint32_t *>(base + o_prefix);
    w.rowbase = reinterpret_cast<uint32_t *>(base + o_rowbase);
    w.colbase = reinterpret_cast<uint32_t *>(base + o_colbase);
    w.Er = reinterpret_cast<uint16_t *>(base + o_Er);
    w.Ec = reinterpret_cast<uint16_t *>(base + o_Ec);
    w.E = reinterpret_cast<uint16_t *>(base + o_E);
    w.K = reinterpret_cast<uint16_t *>(base + o_K);
    w.deltas = reinterpret_cast<int64_t *>(base + o_deltas);
    w.pad = reinterpret_cast<uint32_t *>(base + o_pad);
    // Two streams: the plan and the flagged-pair passes are small, latency-bound kernels; they run beside the padded copy
    // and the range pass (bandwidth / VALU bound) instead of in front of them.
    //   main:  plan_shifts ─┬─ pad ─┬─ range ───────────────────────────────┬─ pick (→ filter, by the caller)
    //   aux:                └─ flags → lists → prefix ─┴─ {lines_rows, lines_cols, exact} → line_keys ─────┘
    if(!ctx->aux_stream)
    {
        int prio_low = 0, prio_high = 0; // numerically lower = higher priority: the small passes should not queue behind the big ones
        LFI_HIP(ctx, hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
        LFI_HIP(ctx, hipStreamCreateWithPriority(&ctx->aux_stream, hipStreamNonBlocking, prio_high));
        LFI_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        LFI_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_pad, hipEventDisableTiming));
        LFI_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    }
    hipStream_t st = ctx->stream;
    hipStream_t aux = ctx->aux_stream;
    // host launch order = the critical path first: the main stream's kernels are enqueued before the side stream's
    hipLaunchKernelGGL(lfi::focus_plan_shifts, dim3(1), dim3(1024), 0, st, a, w);
    LFI_HIP(ctx, hipEventRecord(ctx->ev_fork, st));
    hipLaunchKernelGGL(lfi::focus_pad, dim3((w.Wp + 255) / 256, w.Hp, ctx->n_focus_ids), dim3(64), 0, st, a, w);
    LFI_HIP(ctx, hipEventRecord(ctx->ev_pad, st));
    const uint32_t tiles_x = uint32_t(w.We_p / 256), tiles_y = uint32_t(w.He_p / 4);
    {
        constexpr int CPW = 4, GROUPS = lfi::FOCUS_STEPS / CPW;
        const int striped = tiles_x >= 8;
        const uint32_t nblocks = striped ? 8u * lfi::stripe_blocks_per_xcd(tiles_x, tiles_y, GROUPS) : tiles_x * tiles_y * GROUPS;
        hipLaunchKernelGGL(lfi::focus_range<CPW>, dim3(nblocks), dim3(256), 0, st, a, w, nblocks, striped);
    }
    LFI_HIP(ctx, hipStreamWaitEvent(aux, ctx->ev_fork, 0));
    LFI_HIP(ctx, hipMemsetAsync(w.badx, 0, o_cols - o_badx, aux)); // badx and bady are adjacent
    hipLaunchKernelGGL(lfi::focus_plan_flags, dim3((std::max(W, H) + 255) / 256, lfi::FOCUS_STEPS, 2), dim3(256), 0, aux, a, w);
    hipLaunchKernelGGL(lfi::focus_plan_lists, dim3(lfi::FOCUS_STEPS, 2), dim3(64), 0, aux, a, w);
    hipLaunchKernelGGL(lfi::focus_plan_prefix, dim3(1), dim3(1), 0, aux, a, w);
    LFI_HIP(ctx, hipStreamWaitEvent(aux, ctx->ev_pad, 0));
    {
        const uint32_t per_pass = uint32_t(ctx->cu_count) * 4u / 8u * 8u;
        hipLaunchKernelGGL(lfi::focus_flagged, dim3(3 * per_pass), dim3(256), 0, aux, a, w, per_pass);
    }
    hipLaunchKernelGGL(lfi::focus_line_keys, dim3(ctx->cu_count * 8), dim3(256), 0, aux, a, w);
    LFI_HIP(ctx, hipEventRecord(ctx->ev_join, aux));
    LFI_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_join, 0));
    {
        // two pixels per lane need dword-aligned sample pairs: even radius_x (the reference's is)
        const int ppl = (rx % 2 == 0 && W >= 2) ? 2 : 1;
        const uint32_t nblocks = uint32_t((W + 64 * ppl - 1) / (64 * ppl)) * uint32_t((H + 3) / 4);
        if(ppl == 2)
            hipLaunchKernelGGL(lfi::focus_pick<2>, dim3(nblocks), dim3(256), 0, st, a, w);
        else
            hipLaunchKernelGGL(lfi::focus_pick<1>, dim3(nblocks), dim3(256), 0, st, a, w);
    }
    LFI_HIP(ctx, hipGetLastError());
    *done = true;
    return LFI_OK;
}

} // namespace

// RCCL through dlopen: the library has no link-time dependency on it and single-GPU users never load it
namespace {
struct Rccl
{
    typedef void *comm_t;
    int (*CommInitAll)(comm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};

const Rccl &rccl()
{
    static Rccl r = [] {
        Rccl x;
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if(!h)
            h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if(!h)
            return x;
        x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(dlsym(h, "ncclGroupStart"));
        x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
        x.Broadcast = reinterpret_cast<decltype(x.Broadcast)>(dlsym(h, "ncclBroadcast"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        x.ok = x.CommInitAll && x.CommDestroy && x.GroupStart && x.GroupEnd && x.Broadcast && x.GetErrorString;
        return x;
    }();
    return r;
}
} // namespace

extern "C" {

int lfi_abi_version(void)
{
    return LFI_ABI_VERSION;
}

int lfi_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if(e != hipSuccess)
    {
        g_create_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
        return LFI_ENODEVICE;
    }
    return n;
}

const char *lfi_last_error(const lfi_ctx *ctx)
{
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int lfi_create(int device, lfi_ctx **out_ctx)
{
    if(!out_ctx)
        return fail(nullptr, LFI_EINVAL, "out_ctx is NULL");
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if(e != hipSuccess || n <= 0)
        return fail(nullptr, LFI_ENODEVICE,
                    std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                        "); this library has no CPU fallback");
    if(device < 0 || device >= n)
        return fail(nullptr, LFI_EINVAL, "device index out of range");
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if(e != hipSuccess)
        return fail(nullptr, LFI_EHIP, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
    if(std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, LFI_ENODEVICE,
                    std::string("device is ") + prop.gcnArchName + "; the kernels in this library are built for gfx950 only");
    lfi_ctx *c = new lfi_ctx();
    c->device = device;
    c->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if(hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
       hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
       hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming) != hipSuccess)
    {
        delete c;
        return fail(nullptr, LFI_EHIP, "could not create stream/events on the device");
    }
    c->stream = c->own_stream;
    *out_ctx = c;
    return LFI_OK;
}

int lfi_destroy(lfi_ctx *ctx)
{
    if(!ctx)
        return LFI_EINVAL;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    free_params(ctx);
    free_views(ctx);
    free_grid(ctx);
    if(ctx->ev0)
        (void)hipEventDestroy(ctx->ev0);
    if(ctx->ev1)
        (void)hipEventDestroy(ctx->ev1);
    if(ctx->ev_order)
        (void)hipEventDestroy(ctx->ev_order);
    if(ctx->copy_stream)
    {
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamDestroy(ctx->copy_stream);
    }
    for(int i = 0; i < 2; i++)
    {
        if(ctx->stream_staging[i])
            (void)hipHostFree(ctx->stream_staging[i]);
        if(ctx->ev_h2d[i])
            (void)hipEventDestroy(ctx->ev_h2d[i]);
        if(ctx->ev_rendered[i])
            (void)hipEventDestroy(ctx->ev_rendered[i]);
        if(ctx->ev_d2h[i])
            (void)hipEventDestroy(ctx->ev_d2h[i]);
    }
    if(ctx->quality_sums)
        (void)hipFree(ctx->quality_sums);
    if(ctx->ev_uploads)
        (void)hipEventDestroy(ctx->ev_uploads);
    if(ctx->ev_fork)
        (void)hipEventDestroy(ctx->ev_fork);
    if(ctx->ev_pad)
        (void)hipEventDestroy(ctx->ev_pad);
    if(ctx->ev_join)
        (void)hipEventDestroy(ctx->ev_join);
    if(ctx->aux_stream)
        (void)hipStreamDestroy(ctx->aux_stream);
    if(ctx->own_stream)
        (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return LFI_OK;
}

int lfi_set_stream(lfi_ctx *ctx, void *hip_stream)
{
    if(!ctx)
        return LFI_EINVAL;
    hipStream_t next = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    if(next == ctx->stream)
        return LFI_OK;
    // The context keeps device state that launches on the new stream depend on and that may still be in flight on the old one: the
    // input planes (lfi_fill_synthetic, uploads), the derived planar copy (planar_build), the focus maps and workspace, the views.
    // Everything enqueued so far is ordered before everything enqueued from now on, without blocking the host.
    if(int rc = bind(ctx))
        return rc;
    LFI_HIP(ctx, hipEventRecord(ctx->ev_order, ctx->stream));
    LFI_HIP(ctx, hipStreamWaitEvent(next, ctx->ev_order, 0));
    ctx->stream = next;
    return LFI_OK;
}

int lfi_set_grid(lfi_ctx *ctx, int cols, int rows, int width, int height)
{
    if(!ctx)
        return LFI_EINVAL;
    if(cols < 1 || rows < 1 || width < 1 || height < 1)
        return fail(ctx, LFI_EINVAL, "grid dimensions must be positive");
    if((long)cols * rows > LFI_MAX_IMAGES)
        return fail(ctx, LFI_EINVAL, "more than LFI_MAX_IMAGES (256) grid images");
    if((size_t)width * height > (size_t)1 << 26)
        return fail(ctx, LFI_EINVAL, "image too large");
    if(int rc = bind(ctx))
        return rc;
    if(int rc = lfi_upload_wait(ctx))
        return rc;
    LFI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    free_params(ctx);
    free_views(ctx);
    free_grid(ctx);
    ctx->cols = cols;
    ctx->rows = rows;
    ctx->n = cols * rows;
    ctx->width = width;
    ctx->height = height;
    ctx->in_y0 = ctx->out_y0 = 0;
    ctx->in_rows = ctx->out_rows = height;
    ctx->windowed = false;
    ctx->grid_bytes = plane_bytes(ctx) * ctx->n;
    LFI_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->grid), ctx->grid_bytes));
    ctx->own_grid = true;
    ctx->grid_version++;
    ctx->grid_tracked = true;
    LFI_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->maps), plane_bytes(ctx) * 2));
    LFI_HIP(ctx, hipMemsetAsync(ctx->maps, 0, plane_bytes(ctx) * 2, ctx->stream));
    LFI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LFI_OK;
}

int lfi_set_row_window(lfi_ctx *ctx, int out_y0, int out_y1, int in_y0, int in_y1)
{
    if(!ctx)
        return LFI_EINVAL;
    if(!ctx->n)
        return fail(ctx, LFI_EINVAL, "lfi_set_grid has not been called");
    if(!(0 <= out_y0 && out_y0 < out_y1 && out_y1 <= ctx->height && 0 <= in_y0 && in_y0 < in_y1 && in_y1 <= ctx->height))
        return fail(ctx, LFI_EINVAL, "row window outside the image");
    if(int rc = bind(ctx))
        return rc;
    if(int rc = lfi_upload_wait(ctx))
        return rc;
    LFI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    free_params(ctx);
    free_views(ctx);
    if(ctx->own_grid && ctx->grid)
        (void)hipFree(ctx->grid);
    ctx->grid = nullptr;
    ctx->own_grid = false;
    ctx->in_y0 = in_y0;
    ctx->in_rows = in_y1 - in_y0;
    ctx->out_y0 = out_y0;
    ctx->out_rows = out_y1 - out_y0;
    ctx->windowed = !(in_y0 == 0 && in_y1 == ctx->height && out_y0 == 0 && out_y1 == ctx->height);
    ctx->grid_bytes = in_plane_bytes(ctx) * ctx->n;
    LFI_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->grid), ctx->grid_bytes));
    ctx->own_grid = true;
    ctx->grid_version++;
    ctx->grid_tracked = true;
    return LFI_OK;
}

int lfi_upload_image(lfi_ctx *ctx, int g, const uint8_t *rgba, size_t pitch_bytes)
{
    if(!ctx)
        return LFI_EINVAL;
    if(!ctx->grid)
        return fail(ctx, LFI_EINVAL, "lfi_set_grid has not been called");
    if(g < 0 || g >= ctx->n || !rgba || pitch_bytes < (size_t)ctx->width * 4)
        return fail(ctx, LFI_EINVAL, "bad image index, pointer or pitch");
    if(int rc = bind(ctx))
        return rc;
    if(int rc = join_uploads(ctx))
        return rc;
    // rgba addresses row 0 of the whole image; only the rows this context holds are copied
    LFI_HIP(ctx, hipMemcpy2DAsync(ctx->grid + in_plane_bytes(ctx) * g, (size_t)ctx->width * 4, rgba + (size_t)ctx->in_y0 * pitch_bytes,
                                  pitch_bytes, (size_t)ctx->width * 4, ctx->in_rows, hipMemcpyHostToDevice, ctx->stream));
    LFI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->grid_version++;
    return LFI_OK;
}

int lfi_upload_image_async(lfi_ctx *ctx, int g, const uint8_t *rgba, size_t pitch_bytes)
{
    if(!ctx)
        return LFI_EINVAL;
    if(!ctx->grid)
        return fail(ctx, LFI_EINVAL, "lfi_set_grid has not been called");
    if(g < 0 || g >= ctx->n || !rgba || pitch_bytes < (size_t)ctx->width * 4)
        return fail(ctx, LFI_EINVAL, "bad image index, pointer or pitch");
    if(int rc = bind(ctx))
        return rc;
    if(int rc = ensure_copy_stream(ctx))
        return rc;
    if(!ctx->uploads_pending)
    {
        // first copy of a batch: renders already enqueued on the compute stream may still read the planes
        LFI_HIP(ctx, hipEventRecord(ctx->ev_order, ctx->stream));
        LFI_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_order, 0));
    }
    // rgba addresses row 0 of the whole image; the rows this context holds are copied.  A page-locked source is DMA'd in place
    // (the call returns at once); a pageable one is staged by the HIP runtime through its own page-locked buffers before the call
    // returns (≈45–50 GB/s of host time on this platform — a staging ring of our own with a host memcpy in front measured
    // 20 GB/s, profiles/r02_upload_time.txt), so the caller's buffer is free on return either way.
    const size_t row_bytes = (size_t)ctx->width * 4;
    LFI_HIP(ctx, hipMemcpy2DAsync(ctx->grid + in_plane_bytes(ctx) * g, row_bytes, rgba + (size_t)ctx->in_y0 * pitch_bytes, pitch_bytes, row_bytes,
                                  ctx->in_rows, hipMemcpyHostToDevice, ctx->copy_stream));
    ctx->uploads_pending = true;
    ctx->grid_version++;
    return LFI_OK;
}

int lfi_upload_wait(lfi_ctx *ctx)
{
    if(!ctx)
        return LFI_EINVAL;
    if(int rc = bind(ctx))
        return rc;
    if(int rc = join_uploads(ctx))
        return rc;
    if(ctx->copy_stream)
        LFI_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    return LFI_OK;
}

int lfi_attach_grid(lfi_ctx *ctx, void *device_ptr, size_t bytes)
{
    if(!ctx)
        return LFI_EINVAL;
    if(!ctx->n)
        return fail(ctx, LFI_EINVAL, "lfi_set_grid has not been called");
    if(!device_ptr || bytes < in_plane_bytes(ctx) * ctx->n)
        return fail(ctx, LFI_EINVAL, "attached grid buffer is NULL or smaller than N*rows*W*4 bytes");
    if(reinterpret_cast<uintptr_t>(device_ptr) % 16)
        return fail(ctx, LFI_EINVAL, "attached grid buffer must be 16-byte aligned");
    if(int rc = bind(ctx))
        return rc;
    if(int rc = lfi_upload_wait(ctx))
        return rc;
    LFI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if(ctx->own_grid && ctx->grid)
        (void)hipFree(ctx->grid);
    ctx->grid = static_cast<uint8_t *>(device_ptr);
    ctx->own_grid = false;
    ctx->grid_bytes = bytes;
    ctx->grid_version++;
    ctx->grid_tracked = false; // the caller writes this buffer itself: see lfi_grid_modified
    return LFI_OK;
}

int lfi_broadcast_grid(lfi_ctx *const *ctxs, int n, int root)
{
    if(!ctxs || n < 1 || root < 0 || root >= n)
        return LFI_EINVAL;
    for(int i = 0; i < n; i++)
        if(!ctxs[i])
            return LFI_EINVAL;
    lfi_ctx *r0 = ctxs[root];
    if(!r0->grid)
        return fail(r0, LFI_EINVAL, "lfi_set_grid has not been called on the root context");
    if(n == 1)
        return LFI_OK;
    std::vector<int> devs(n);
    for(int i = 0; i < n; i++)
    {
        lfi_ctx *c = ctxs[i];
        if(!c->grid || c->n != r0->n || c->width != r0->width || c->height != r0->height || c->in_y0 != r0->in_y0 || c->in_rows != r0->in_rows)
            return fail(r0, LFI_EINVAL, "all contexts of a broadcast must describe the same grid and row window");
        devs[i] = c->device;
        for(int j = 0; j < i; j++)
            if(devs[j] == devs[i])
                return fail(r0, LFI_EINVAL, "the contexts of a broadcast must sit on distinct devices");
    }
    const Rccl &nc = rccl();
    if(!nc.ok)
        return fail(r0, LFI_EHIP, "librccl.so could not be loaded");
    for(int i = 0; i < n; i++)
        if(ctxs[i]->uploads_pending)
            if(int rc = lfi_upload_wait(ctxs[i]))
                return rc;
    int caller_device = -1; // the loop below walks the contexts' devices; the caller's current device is restored afterwards
    (void)hipGetDevice(&caller_device);
    std::vector<Rccl::comm_t> comms(n, nullptr);
    int rc = nc.CommInitAll(comms.data(), n, devs.data());
    if(rc != 0)
    {
        if(caller_device >= 0)
            (void)hipSetDevice(caller_device);
        return fail(r0, LFI_EHIP, std::string("ncclCommInitAll: ") + nc.GetErrorString(rc));
    }
    const size_t bytes = in_plane_bytes(r0) * r0->n;
    constexpr int NCCL_UINT8 = 1;
    int status = LFI_OK;
    rc = nc.GroupStart();
    for(int i = 0; i < n && rc == 0; i++)
    {
        if(hipSetDevice(ctxs[i]->device) != hipSuccess)
        {
            rc = -1;
            break;
        }
        rc = nc.Broadcast(r0->grid, ctxs[i]->grid, bytes, NCCL_UINT8, root, comms[i], ctxs[i]->stream);
    }
    const int rc_end = nc.GroupEnd();
    if(rc != 0 || rc_end != 0)
        status = fail(r0, LFI_EHIP, std::string("ncclBroadcast: ") + (rc > 0 ? nc.GetErrorString(rc) : rc_end ? nc.GetErrorString(rc_end) : "hipSetDevice failed"));
    for(int i = 0; i < n; i++)
    {
        (void)hipSetDevice(ctxs[i]->device);
        if(hipStreamSynchronize(ctxs[i]->stream) != hipSuccess && status == LFI_OK)
            status = fail(r0, LFI_EHIP, "hipStreamSynchronize after the broadcast failed");
    }
    for(int i = 0; i < n; i++)
    {
        (void)nc.CommDestroy(comms[i]);
        ctxs[i]->grid_version++;
    }
    if(caller_device >= 0)
        (void)hipSetDevice(caller_device);
    return status;
}

int lfi_grid_device_ptr(lfi_ctx *ctx, void **out_ptr, size_t *out_bytes)
{
    if(!ctx || !out_ptr)
        return LFI_EINVAL;
    if(ctx->uploads_pending && bind(ctx) == LFI_OK)
        (void)lfi_upload_wait(ctx); // the caller is about to read or write the planes itself
    *out_ptr = ctx->grid;
    if(out_bytes)
        *out_bytes = ctx->grid ? in_plane_bytes(ctx) * ctx->n : 0;
    ctx->grid_tracked = false; // the caller may write through the pointer: see lfi_grid_modified
    ctx->grid_version++;
    return LFI_OK;
}

int lfi_grid_modified(lfi_ctx *ctx)
{
    if(!ctx)
        return LFI_EINVAL;
    if(!ctx->grid)
        return fail(ctx, LFI_EINVAL, "lfi_set_grid has not been called");
    ctx->grid_version++;
    ctx->grid_tracked = true; // the caller announces its writes from now on
    return LFI_OK;
}

int lfi_fill_synthetic_images(lfi_ctx *ctx, uint32_t seed, int g0, int g1)
{
    if(!ctx)
        return LFI_EINVAL;
    if(!ctx->grid)
        return fail(ctx, LFI_EINVAL, "lfi_set_grid has not been called");
    if(g0 < 0 || g1 > ctx->n || g0 > g1)
        return fail(ctx, LFI_EINVAL, "image range [g0, g1) outside the grid");
    if(g0 == g1)
        return LFI_OK;
    if(int rc = bind(ctx))
        return rc;
    if(int rc = join_uploads(ctx))
        return rc;
    ctx->grid_version++;
    hipLaunchKernelGGL(lfi::fill_synthetic, dim3(256 * 16), dim3(256), 0, ctx->stream, ctx->grid, g0, g1 - g0, ctx->width,
                       ctx->in_rows, ctx->in_y0, seed);
    LFI_HIP(ctx, hipGetLastError());
    return LFI_OK;
}

int lfi_fill_synthetic_scene(lfi_ctx *ctx, uint32_t seed)
{
    if(!ctx)
        return LFI_EINVAL;
    if(!ctx->grid || !ctx->have_params)
        return fail(ctx, LFI_EINVAL, "lfi_set_grid / lfi_set_params have not been called (the scene is built from the images' offsets)");
    if(int rc = bind(ctx))
        return rc;
    ctx->grid_version++;
    hipLaunchKernelGGL(lfi::fill_scene, dim3(256 * 16), dim3(256), 0, ctx->stream, ctx->grid, ctx->d_offsets, ctx->n, ctx->width, ctx->in_rows,
                       ctx->in_y0, seed, ctx->focus, ctx->range);
    LFI_HIP(ctx, hipGetLastError());
    return LFI_OK;
}

int lfi_fill_synthetic(lfi_ctx *ctx, uint32_t seed)
{
    return ctx ? lfi_fill_synthetic_images(ctx, seed, 0, ctx->n) : LFI_EINVAL;
}

int lfi_set_params(lfi_ctx *ctx, const lfi_params *p)
{
    if(!ctx)
        return LFI_EINVAL;
    if(!ctx->n)
        return fail(ctx, LFI_EINVAL, "lfi_set_grid has not been called");
    if(!p || p->views < 1 || !p->focused_offsets || !p->offsets || !p->weights_fp16)
        return fail(ctx, LFI_EINVAL, "lfi_params: views < 1 or a required array is NULL");
    if(p->n_focus_ids < 0 || p->n_focus_ids > LFI_MAX_FOCUS_IDS || (p->n_focus_ids > 0 && !p->focus_map_ids))
        return fail(ctx, LFI_EINVAL, "lfi_params: bad focus_map_ids");
    for(int i = 0; i < p->n_focus_ids; i++)
        if(p->focus_map_ids[i] < 0 || p->focus_map_ids[i] >= ctx->n)
            return fail(ctx, LFI_EINVAL, "lfi_params: focus_map_ids entry outside the grid");
    if(p->views > 4096)
        return fail(ctx, LFI_EINVAL, "lfi_params: more than 4096 views");
    if(int rc = bind(ctx))
        return rc;
    LFI_HIP(ctx, hipStreamSynchronize(ctx->stream));

    if(ctx->windowed)
        for(int g = 0; g < ctx->n; g++)
        {
            const int oy = p->focused_offsets[g].y, H = ctx->height;
            const int lo = std::min(std::max(ctx->out_y0 + oy, 0), H - 1), hi = std::min(std::max(ctx->out_y0 + ctx->out_rows - 1 + oy, 0), H - 1);
            if(lo < ctx->in_y0 || hi >= ctx->in_y0 + ctx->in_rows)
                return fail(ctx, LFI_EINVAL, "the input row window does not cover the rows image " + std::to_string(g) + " is sampled at");
        }
    const int n = ctx->n, V = p->views;
    const int k_pad = (n + 15) / 16 * 16;
    // 64 spare rows: a view range may start anywhere, and a wave always reads whole 32-row tiles
    const int v_pad = (V + 63) / 64 * 64 + 64;
    const bool views_changed = V != ctx->views_n;

    // host staging of the padded arrays, one blob
    const size_t off_focused = 0;
    const size_t off_offsets = off_focused + sizeof(lfi_int2) * k_pad;
    const size_t off_w16 = off_offsets + sizeof(lfi_float2) * k_pad;
    const size_t off_w16s = (off_w16 + sizeof(uint16_t) * (size_t)v_pad * k_pad + 15) / 16 * 16;
    const size_t off_w32 = (off_w16s + sizeof(uint16_t) * (size_t)v_pad * k_pad + 15) / 16 * 16;
    const size_t off_w32t = off_w32 + sizeof(float) * (size_t)v_pad * k_pad;
    const size_t off_ids = off_w32t + sizeof(float) * (size_t)v_pad * k_pad;
    const size_t total = off_ids + sizeof(int32_t) * LFI_MAX_FOCUS_IDS;
    std::vector<uint8_t> blob(total, 0);
    std::memcpy(blob.data() + off_focused, p->focused_offsets, sizeof(lfi_int2) * n);
    int fo_min[2] = {p->focused_offsets[0].x, p->focused_offsets[0].y}, fo_max[2] = {fo_min[0], fo_min[1]};
    for(int g = 1; g < n; g++)
    {
        fo_min[0] = std::min(fo_min[0], p->focused_offsets[g].x);
        fo_max[0] = std::max(fo_max[0], p->focused_offsets[g].x);
        fo_min[1] = std::min(fo_min[1], p->focused_offsets[g].y);
        fo_max[1] = std::max(fo_max[1], p->focused_offsets[g].y);
    }
    std::memcpy(blob.data() + off_offsets, p->offsets, sizeof(lfi_float2) * n);
    bool scalable = true, sums_ok = true;
    fill_weight_arrays(p->weights_fp16, V, n, k_pad, v_pad, blob.data() + off_w16, off_w16s - off_w16, off_w32 - off_w16, off_w32t - off_w16, &scalable,
                       &sums_ok);
    if(p->n_focus_ids)
        std::memcpy(blob.data() + off_ids, p->focus_map_ids, sizeof(int32_t) * p->n_focus_ids);

    free_params(ctx);
    LFI_HIP(ctx, hipMalloc(&ctx->param_blob, total));
    LFI_HIP(ctx, hipMemcpy(ctx->param_blob, blob.data(), total, hipMemcpyHostToDevice));
    uint8_t *base = static_cast<uint8_t *>(ctx->param_blob);
    ctx->d_focused = reinterpret_cast<lfi_int2 *>(base + off_focused);
    ctx->d_offsets = reinterpret_cast<lfi_float2 *>(base + off_offsets);
    ctx->d_w16 = reinterpret_cast<uint16_t *>(base + off_w16);
    ctx->d_w16s = reinterpret_cast<uint16_t *>(base + off_w16s);
    ctx->weights_scalable = scalable;
    ctx->weights_sum_ok = sums_ok;
    ctx->d_w32 = reinterpret_cast<float *>(base + off_w32);
    ctx->d_w32t = reinterpret_cast<float *>(base + off_w32t);
    ctx->d_ids = reinterpret_cast<int32_t *>(base + off_ids);
    ctx->blob_off_w16 = off_w16;
    ctx->blob_weights_bytes = off_ids - off_w16;
    ctx->k_pad = k_pad;
    ctx->v_pad = v_pad;
    ctx->views_n = V;
    ctx->n_focus_ids = p->n_focus_ids;
    ctx->h_offsets.assign(p->offsets, p->offsets + n);
    ctx->h_focus_offsets.clear();
    for(int k = 0; k < p->n_focus_ids; k++)
        ctx->h_focus_offsets.push_back(p->offsets[p->focus_map_ids[k]]);
    ctx->focus = p->focus;
    ctx->range = p->range;
    for(int d = 0; d < 2; d++)
    {
        ctx->fo_min[d] = fo_min[d];
        ctx->fo_max[d] = fo_max[d];
    }
    ctx->radius[0] = std::max(p->block_radius[0], 1);
    ctx->radius[1] = std::max(p->block_radius[1], 1);
    ctx->flags = p->flags;

    if(views_changed || !ctx->views)
    {
        const bool was_attached = ctx->views && !ctx->own_views;
        if(!(was_attached && ctx->views_bytes >= out_plane_bytes(ctx) * V))
        {
            free_views(ctx);
            ctx->views_bytes = out_plane_bytes(ctx) * V;
            LFI_HIP(ctx, alloc_views(&ctx->views, ctx->views_bytes, ctx->out_layout == LFI_LAYOUT_PLANAR_RGB));
            ctx->own_views = true;
        }
    }
    ctx->have_params = true;
    return LFI_OK;
}

int lfi_attach_views(lfi_ctx *ctx, void *device_ptr, size_t bytes)
{
    if(!ctx)
        return LFI_EINVAL;
    if(!ctx->have_params)
        return fail(ctx, LFI_EINVAL, "lfi_set_params has not been called");
    if(!device_ptr || bytes < out_plane_bytes(ctx) * ctx->views_n)
        return fail(ctx, LFI_EINVAL, "attached view buffer is NULL or smaller than the views in the current layout (lfi_view_layout)");
    if(reinterpret_cast<uintptr_t>(device_ptr) % 16)
        return fail(ctx, LFI_EINVAL, "attached view buffer must be 16-byte aligned");
    if(int rc = bind(ctx))
        return rc;
    LFI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    free_views(ctx);
    ctx->views = static_cast<uint8_t *>(device_ptr);
    ctx->views_bytes = bytes;
    return LFI_OK;
}

int lfi_set_output_layout(lfi_ctx *ctx, int layout)
{
    if(!ctx)
        return LFI_EINVAL;
    if(layout != LFI_LAYOUT_RGBA && layout != LFI_LAYOUT_PLANAR_RGB)
        return fail(ctx, LFI_EINVAL, "unknown view layout");
    if(!ctx->n)
        return fail(ctx, LFI_EINVAL, "lfi_set_grid has not been called");
    if(layout == ctx->out_layout)
        return LFI_OK;
    if(int rc = bind(ctx))
        return rc;
    LFI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    free_views(ctx); // an attached buffer is dropped too: its size belongs to the old layout
    ctx->out_layout = layout;
    if(ctx->have_params)
    {
        ctx->views_bytes = out_plane_bytes(ctx) * ctx->views_n;
        LFI_HIP(ctx, alloc_views(&ctx->views, ctx->views_bytes, ctx->out_layout == LFI_LAYOUT_PLANAR_RGB));
        ctx->own_views = true;
    }
    return LFI_OK;
}

int lfi_view_layout(lfi_ctx *ctx, lfi_view_layout_info *out)
{
    if(!ctx || !out)
        return LFI_EINVAL;
    if(!ctx->n)
        return fail(ctx, LFI_EINVAL, "lfi_set_grid has not been called");
    out->layout = ctx->out_layout;
    out->rows = ctx->out_rows;
    if(ctx->out_layout == LFI_LAYOUT_PLANAR_RGB)
    {
        out->row_pitch_bytes = (size_t)view_pitch(ctx);
        out->plane_stride_bytes = (size_t)ctx->out_rows * view_pitch(ctx);
    }
    else
    {
        out->row_pitch_bytes = (size_t)ctx->width * 4;
        out->plane_stride_bytes = 0;
    }
    out->view_stride_bytes = out_plane_bytes(ctx);
    return LFI_OK;
}

int lfi_views_device_ptr(lfi_ctx *ctx, void **out_ptr, size_t *out_bytes)
{
    if(!ctx || !out_ptr)
        return LFI_EINVAL;
    *out_ptr = ctx->views;
    if(out_bytes)
        *out_bytes = ctx->views ? out_plane_bytes(ctx) * ctx->views_n : 0;
    return LFI_OK;
}

int lfi_focus_map(lfi_ctx *ctx)
{
    if(!ctx)
        return LFI_EINVAL;
    if(!ctx->grid || !ctx->have_params)
        return fail(ctx, LFI_EINVAL, "lfi_set_grid / lfi_set_params have not been called");
    if(ctx->n_focus_ids < 1)
        return fail(ctx, LFI_EINVAL, "no focus_map_ids in the parameters");
    if(!(ctx->range > 0.0f))
        return fail(ctx, LFI_EINVAL, "focus range must be > 0 for the focus map");
    if(int rc = bind(ctx))
        return rc;
    if(int rc = join_uploads(ctx))
        return rc;
    KernelArgs a = make_args(ctx, 0, ctx->views_n, LFI_METHOD_STD);
    if(ctx->windowed)
    {
        // Row window (spatial sharding): the maps are whole-image planes, but only the band's rows are computed — map 0 for the band
        // plus the filter's reach above and below, map 1 for the band — by the one-wave-per-row kernel from the input rows held.
        const int H = ctx->height, fry = std::max(ctx->radius[1] / 10, 1), ry = ctx->radius[1];
        const int e0 = std::max(ctx->out_y0 - fry, 0), e1 = std::min(ctx->out_y0 + ctx->out_rows + fry, H);
        const float f_lo = std::min(ctx->focus, ctx->focus + ctx->range), f_hi = std::max(ctx->focus, ctx->focus + ctx->range);
        for(const lfi_float2 &o : ctx->h_focus_offsets)
        {
            const double d_lo = std::min((double)f_lo * o.y, (double)f_hi * o.y), d_hi = std::max((double)f_lo * o.y, (double)f_hi * o.y);
            const int lo = std::min(std::max((int)std::floor(e0 + d_lo) - 1 - ry, 0), H - 1);
            const int hi = std::min(std::max((int)std::ceil(e1 - 1 + d_hi) + 1 + ry, 0), H - 1);
            if(lo < ctx->in_y0 || hi >= ctx->in_y0 + ctx->in_rows)
                return fail(ctx, LFI_EINVAL, "the input row window does not cover the rows the focus map of this band samples");
        }
        a.map_y0 = e0;
        a.map_rows = e1 - e0;
        hipLaunchKernelGGL((lfi::focus_estimate_packed<2, 4>), dim3((ctx->width + 127) / 128, a.map_rows), dim3(64), 0, ctx->stream, a);
        LFI_HIP(ctx, hipGetLastError());
        a.map_y0 = ctx->out_y0;
        a.map_rows = ctx->out_rows;
        hipLaunchKernelGGL(lfi::focus_filter, dim3((ctx->width + 63) / 64, (ctx->out_rows + 3) / 4), dim3(256), 0, ctx->stream, a);
        LFI_HIP(ctx, hipGetLastError());
        return LFI_OK;
    }
    // the LDS-staged kernel needs its window (128 + 2·radius_x + slack pixels) to fit a 256-pixel LDS row
    const bool lds_fits = 128 + 2 * ctx->radius[0] + 2 * lfi::FOCUS_LDS_SLACK <= lfi::FOCUS_LDS_ROW;
    // "factored" (default): W and H must fit the 16-bit column / row lists
    bool done = false;
    if(ctx->focus_variant == 0 && ctx->width <= 65535 && ctx->height <= 65535)
        if(int rc = launch_focus_factored(ctx, a, &done))
            return rc;
    if(done)
        ;
    else if(ctx->focus_variant <= 1 && lds_fits) // "lds"
        hipLaunchKernelGGL(lfi::focus_estimate_lds, dim3((ctx->width + 127) / 128, ctx->height), dim3(64), 0, ctx->stream, a);
    else if(ctx->focus_variant == 3) // "plain": one pixel per lane, float min/max exactly as the reference writes it
        hipLaunchKernelGGL(lfi::focus_estimate, pixel_grid(ctx), dim3(256), 0, ctx->stream, a);
    else // "packed_p2" (also the fallback of "lds" for very large radii)
        hipLaunchKernelGGL((lfi::focus_estimate_packed<2, 4>), dim3((ctx->width + 127) / 128, ctx->height), dim3(64), 0, ctx->stream, a);
    LFI_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(lfi::focus_filter, pixel_grid(ctx), dim3(256), 0, ctx->stream, a);
    LFI_HIP(ctx, hipGetLastError());
    return LFI_OK;
}

int lfi_render(lfi_ctx *ctx, int method, int all_focus, int v0, int v1)
{
    if(int rc = check_render_args(ctx, method, v0, v1))
        return rc;
    if(int rc = bind(ctx))
        return rc;
    const KernelArgs a = make_args(ctx, v0, v1, method);
    return launch_blend(ctx, method, all_focus, a);
}

int lfi_prepare(lfi_ctx *ctx, int method, int all_focus, int v0, int v1)
{
    if(int rc = check_render_args(ctx, method, v0, v1))
        return rc;
    if(int rc = bind(ctx))
        return rc;
    if(int rc = join_uploads(ctx))
        return rc;
    const KernelArgs a = make_args(ctx, v0, v1, method);
    ctx->derived_build_ms = 0.0f;
    if(wants_planar(ctx, method, all_focus, a))
    {
        const uint64_t before = ctx->planar_version;
        LFI_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        const bool ok = ensure_planar(ctx);
        LFI_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        LFI_HIP(ctx, hipEventSynchronize(ctx->ev1));
        if(ok && ctx->planar_version != before)
            LFI_HIP(ctx, hipEventElapsedTime(&ctx->derived_build_ms, ctx->ev0, ctx->ev1));
    }
    return LFI_OK;
}

int lfi_memory_info(lfi_ctx *ctx, lfi_memory *out)
{
    if(!ctx || !out)
        return LFI_EINVAL;
    out->grid_bytes = ctx->grid ? in_plane_bytes(ctx) * ctx->n : 0;
    out->derived_bytes = ctx->planar ? ctx->planar_bytes : 0;
    out->views_bytes = ctx->views ? ctx->views_bytes : 0;
    out->maps_bytes = ctx->maps ? plane_bytes(ctx) * 2 : 0;
    out->workspace_bytes = ctx->focus_ws_bytes;
    out->derived_build_ms = ctx->derived_build_ms;
    return LFI_OK;
}

const char *lfi_last_kernel_name(const lfi_ctx *ctx)
{
    return ctx ? ctx->last_kernel : "";
}

int lfi_render_stream(lfi_ctx *ctx, int method, int all_focus, const uint16_t *weights_fp16, int total_views, uint8_t *host_out, size_t pitch_bytes)
{
    if(int rc = check_render_args(ctx, method, 0, 1))
        return rc;
    if(!weights_fp16 || total_views < 1)
        return fail(ctx, LFI_EINVAL, "lfi_render_stream: weights are NULL or total_views < 1");
    if(host_out && (ctx->out_layout != LFI_LAYOUT_RGBA || pitch_bytes < (size_t)ctx->width * 4))
        return fail(ctx, LFI_EINVAL, "lfi_render_stream: downloads need the RGBA view layout and a pitch of at least width*4 bytes");
    if(int rc = bind(ctx))
        return rc;
    const int V = ctx->views_n, n = ctx->n, k_pad = ctx->k_pad, v_pad = ctx->v_pad;
    const size_t wbytes = ctx->blob_weights_bytes;
    const size_t off_w16s = (sizeof(uint16_t) * (size_t)v_pad * k_pad + 15) / 16 * 16; // the layout lfi_set_params laid out
    const size_t off_w32 = (off_w16s + sizeof(uint16_t) * (size_t)v_pad * k_pad + 15) / 16 * 16;
    const size_t off_w32t = off_w32 + sizeof(float) * (size_t)v_pad * k_pad;
    if(ctx->stream_staging_bytes != wbytes)
    {
        for(int i = 0; i < 2; i++)
        {
            if(ctx->stream_staging[i])
                (void)hipHostFree(ctx->stream_staging[i]);
            ctx->stream_staging[i] = nullptr;
        }
        ctx->stream_staging_bytes = 0;
        for(int i = 0; i < 2; i++)
            LFI_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->stream_staging[i]), wbytes, hipHostMallocDefault));
        ctx->stream_staging_bytes = wbytes;
    }
    for(int i = 0; i < 2; i++)
        if(!ctx->ev_h2d[i])
        {
            LFI_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_h2d[i], hipEventDisableTiming));
            LFI_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_rendered[i], hipEventDisableTiming));
            LFI_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_d2h[i], hipEventDisableTiming));
        }
    if(host_out)
    {
        if(int rc = ensure_copy_stream(ctx))
            return rc;
        const size_t need = out_plane_bytes(ctx) * V;
        if(ctx->views2_bytes != need)
        {
            if(ctx->views2)
                (void)hipFree(ctx->views2);
            ctx->views2 = nullptr;
            ctx->views2_bytes = 0;
            LFI_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->views2), need));
            ctx->views2_bytes = need;
        }
    }
    if(int rc = join_uploads(ctx))
        return rc;
    const int n_blocks = (total_views + V - 1) / V;
    uint8_t *dev_weights = static_cast<uint8_t *>(ctx->param_blob) + ctx->blob_off_w16;
    uint8_t *const vbuf[2] = {ctx->views, host_out ? ctx->views2 : ctx->views};
    uint8_t *const views_saved = ctx->views;
    int status = LFI_OK;
    for(int b = 0; b < n_blocks && status == LFI_OK; b++)
    {
        const int slot = b & 1;
        const int nv = std::min(V, total_views - b * V);
        // the staging slot is free once the copy of block b − 2 out of it has been executed
        if(b >= 2)
            LFI_HIP(ctx, hipEventSynchronize(ctx->ev_h2d[slot]));
        std::memset(ctx->stream_staging[slot], 0, wbytes);
        bool scalable = true, sums_ok = true;
        fill_weight_arrays(weights_fp16 + (size_t)b * V * n, nv, n, k_pad, v_pad, ctx->stream_staging[slot], off_w16s, off_w32, off_w32t, &scalable, &sums_ok);
        ctx->weights_scalable = scalable; // the dispatch of THIS block's launch (read at enqueue time)
        ctx->weights_sum_ok = sums_ok;
        // stream order protects the device arrays: the previous block's kernel is ahead of this copy on the same stream
        LFI_HIP(ctx, hipMemcpyAsync(dev_weights, ctx->stream_staging[slot], wbytes, hipMemcpyHostToDevice, ctx->stream));
        LFI_HIP(ctx, hipEventRecord(ctx->ev_h2d[slot], ctx->stream));
        if(host_out && b >= 2)
            LFI_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_d2h[slot], 0)); // block b − 2 has left this set of views
        ctx->views = vbuf[slot];
        const KernelArgs a = make_args(ctx, 0, nv, method);
        status = launch_blend(ctx, method, all_focus, a);
        ctx->views = views_saved;
        if(status != LFI_OK || !host_out)
            continue;
        LFI_HIP(ctx, hipEventRecord(ctx->ev_rendered[slot], ctx->stream));
        LFI_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_rendered[slot], 0));
        const size_t view_bytes = pitch_bytes * ctx->height;
        for(int v = 0; v < nv; v++)
            LFI_HIP(ctx, hipMemcpy2DAsync(host_out + ((size_t)b * V + v) * view_bytes + (size_t)ctx->out_y0 * pitch_bytes, pitch_bytes,
                                          vbuf[slot] + out_plane_bytes(ctx) * v, (size_t)ctx->width * 4, (size_t)ctx->width * 4, ctx->out_rows,
                                          hipMemcpyDeviceToHost, ctx->copy_stream));
        LFI_HIP(ctx, hipEventRecord(ctx->ev_d2h[slot], ctx->copy_stream));
    }
    LFI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if(host_out)
        LFI_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    return status;
}

int lfi_compare_view(lfi_ctx *ctx, int v, const uint8_t *reference_rgba, size_t pitch_bytes, lfi_quality *out)
{
    if(!ctx || !out)
        return LFI_EINVAL;
    if(!ctx->views || !ctx->have_params)
        return fail(ctx, LFI_EINVAL, "nothing rendered yet");
    if(v < 0 || v >= ctx->views_n || !reference_rgba || pitch_bytes < (size_t)ctx->width * 4)
        return fail(ctx, LFI_EINVAL, "bad view index, pointer or pitch");
    if(ctx->windowed)
        return fail(ctx, LFI_EINVAL, "lfi_compare_view needs the whole view (no row window)");
    if(int rc = bind(ctx))
        return rc;
    const size_t need = plane_bytes(ctx);
    if(ctx->quality_ref_bytes != need)
    {
        if(ctx->quality_ref)
            (void)hipFree(ctx->quality_ref);
        ctx->quality_ref = nullptr;
        ctx->quality_ref_bytes = 0;
        LFI_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->quality_ref), need));
        ctx->quality_ref_bytes = need;
    }
    if(!ctx->quality_sums)
        LFI_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->quality_sums), sizeof(lfi::QualitySums)));
    LFI_HIP(ctx, hipMemcpy2DAsync(ctx->quality_ref, (size_t)ctx->width * 4, reference_rgba, pitch_bytes, (size_t)ctx->width * 4, ctx->height,
                                  hipMemcpyHostToDevice, ctx->stream));
    LFI_HIP(ctx, hipMemsetAsync(ctx->quality_sums, 0, sizeof(lfi::QualitySums), ctx->stream));
    const uint8_t *view = nullptr;
    if(int rc = rgba_plane_of_view(ctx, v, &view))
        return rc;
    const int bw = (ctx->width + 3) / 4, bh = (ctx->height + 3) / 4; // 4×4 blocks
    hipLaunchKernelGGL(lfi::quality_reduce, dim3((bw + 15) / 16, (bh + 15) / 16), dim3(256), 0, ctx->stream, reinterpret_cast<const uint32_t *>(view),
                       reinterpret_cast<const uint32_t *>(ctx->quality_ref), ctx->width, ctx->height, ctx->quality_sums);
    LFI_HIP(ctx, hipGetLastError());
    lfi::QualitySums sums{};
    LFI_HIP(ctx, hipMemcpyAsync(&sums, ctx->quality_sums, sizeof(sums), hipMemcpyDeviceToHost, ctx->stream));
    LFI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const double px = (double)ctx->width * ctx->height;
    double mse_all = 0, ssim_all = 0;
    for(int c = 0; c < 3; c++)
    {
        out->mse[c] = (double)sums.sq_err[c] / px;
        out->psnr[c] = out->mse[c] > 0 ? 10.0 * std::log10(255.0 * 255.0 / out->mse[c]) : INFINITY;
        out->ssim[c] = sums.windows ? sums.ssim[c] / (double)sums.windows : 1.0;
        mse_all += out->mse[c] / 3.0;
        ssim_all += out->ssim[c] / 3.0;
    }
    out->psnr_all = mse_all > 0 ? 10.0 * std::log10(255.0 * 255.0 / mse_all) : INFINITY;
    out->ssim_all = ssim_all;
    return LFI_OK;
}

int lfi_sync(lfi_ctx *ctx)
{
    if(!ctx)
        return LFI_EINVAL;
    if(int rc = bind(ctx))
        return rc;
    if(int rc = join_uploads(ctx))
        return rc;
    LFI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LFI_OK;
}

int lfi_timer_start(lfi_ctx *ctx)
{
    if(!ctx)
        return LFI_EINVAL;
    if(int rc = bind(ctx))
        return rc;
    LFI_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return LFI_OK;
}

int lfi_timer_stop(lfi_ctx *ctx, float *out_ms)
{
    if(!ctx || !out_ms)
        return LFI_EINVAL;
    if(int rc = bind(ctx))
        return rc;
    LFI_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    LFI_HIP(ctx, hipEventSynchronize(ctx->ev1));
    LFI_HIP(ctx, hipEventElapsedTime(out_ms, ctx->ev0, ctx->ev1));
    return LFI_OK;
}

int lfi_benchmark(lfi_ctx *ctx, int method, int all_focus, int v0, int v1, int warmup, int runs, lfi_bench_stats *out)
{
    if(int rc = check_render_args(ctx, method, v0, v1))
        return rc;
    if(!out || runs < 1 || warmup < 0)
        return fail(ctx, LFI_EINVAL, "lfi_benchmark: runs must be ≥ 1 and out_stats non-NULL");
    if(int rc = bind(ctx))
        return rc;
    const KernelArgs a = make_args(ctx, v0, v1, method);
    // the derived input copy is (re)built here, not inside the first timed launch
    if(wants_planar(ctx, method, all_focus, a))
        (void)ensure_planar(ctx);
    for(int i = 0; i < warmup; i++)
        if(int rc = launch_blend(ctx, method, all_focus, a))
            return rc;
    LFI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<float> t(runs);
    for(int i = 0; i < runs; i++)
    {
        LFI_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        if(int rc = launch_blend(ctx, method, all_focus, a))
            return rc;
        LFI_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        LFI_HIP(ctx, hipEventSynchronize(ctx->ev1));
        LFI_HIP(ctx, hipEventElapsedTime(&t[i], ctx->ev0, ctx->ev1));
    }
    float b2b = 0;
    LFI_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for(int i = 0; i < runs; i++)
        if(int rc = launch_blend(ctx, method, all_focus, a))
            return rc;
    LFI_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    LFI_HIP(ctx, hipEventSynchronize(ctx->ev1));
    LFI_HIP(ctx, hipEventElapsedTime(&b2b, ctx->ev0, ctx->ev1));
    std::vector<float> s = t;
    std::sort(s.begin(), s.end());
    double sum = 0;
    for(float x : t)
        sum += x;
    out->runs = runs;
    out->mean_ms = float(sum / runs);
    out->median_ms = (runs & 1) ? s[runs / 2] : 0.5f * (s[runs / 2 - 1] + s[runs / 2]);
    out->min_ms = s.front();
    out->max_ms = s.back();
    out->back_to_back_ms = b2b / runs;
    return LFI_OK;
}

int lfi_download_view(lfi_ctx *ctx, int v, uint8_t *rgba, size_t pitch_bytes)
{
    if(!ctx)
        return LFI_EINVAL;
    if(!ctx->views || !ctx->have_params)
        return fail(ctx, LFI_EINVAL, "nothing rendered yet");
    if(v < 0 || v >= ctx->views_n || !rgba || pitch_bytes < (size_t)ctx->width * 4)
        return fail(ctx, LFI_EINVAL, "bad view index, pointer or pitch");
    if(int rc = bind(ctx))
        return rc;
    // rgba addresses row 0 of the whole view; the rows this context rendered are written at their place
    const uint8_t *src = nullptr;
    if(int rc = rgba_plane_of_view(ctx, v, &src))
        return rc;
    LFI_HIP(ctx, hipMemcpy2DAsync(rgba + (size_t)ctx->out_y0 * pitch_bytes, pitch_bytes, src,
                                  (size_t)ctx->width * 4, (size_t)ctx->width * 4, ctx->out_rows, hipMemcpyDeviceToHost, ctx->stream));
    LFI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LFI_OK;
}

int lfi_download_map(lfi_ctx *ctx, int k, uint8_t *rgba, size_t pitch_bytes)
{
    if(!ctx)
        return LFI_EINVAL;
    if(!ctx->maps)
        return fail(ctx, LFI_EINVAL, "lfi_set_grid has not been called");
    if(k < 0 || k > 1 || !rgba || pitch_bytes < (size_t)ctx->width * 4)
        return fail(ctx, LFI_EINVAL, "bad map index, pointer or pitch");
    if(int rc = bind(ctx))
        return rc;
    LFI_HIP(ctx, hipMemcpy2DAsync(rgba, pitch_bytes, ctx->maps + plane_bytes(ctx) * k, (size_t)ctx->width * 4,
                                  (size_t)ctx->width * 4, ctx->height, hipMemcpyDeviceToHost, ctx->stream));
    LFI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LFI_OK;
}

int lfi_download_quilt(lfi_ctx *ctx, int tiles_x, int tiles_y, int v0, uint8_t *rgba, size_t pitch_bytes)
{
    if(!ctx)
        return LFI_EINVAL;
    if(!ctx->views || !ctx->have_params)
        return fail(ctx, LFI_EINVAL, "nothing rendered yet");
    if(tiles_x < 1 || tiles_y < 1 || v0 < 0 || (long)v0 + (long)tiles_x * tiles_y > ctx->views_n)
        return fail(ctx, LFI_EINVAL, "quilt needs tiles_x*tiles_y views starting at v0 inside [0, views)");
    if(!rgba || pitch_bytes < (size_t)tiles_x * ctx->width * 4)
        return fail(ctx, LFI_EINVAL, "bad quilt pointer or pitch");
    if(int rc = bind(ctx))
        return rc;
    // one strided device→host copy per tile straight into its place in the quilt: no staging buffer, no extra kernel
    for(int ty = 0; ty < tiles_y; ty++)
        for(int tx = 0; tx < tiles_x; tx++)
        {
            const int v = v0 + ty * tiles_x + tx;
            uint8_t *dst = rgba + ((size_t)ty * ctx->height + ctx->out_y0) * pitch_bytes + (size_t)tx * ctx->width * 4;
            const uint8_t *src = nullptr; // planar layout: expanded into the staging plane, which the copy below reads in stream order
            if(int rc = rgba_plane_of_view(ctx, v, &src))
                return rc;
            LFI_HIP(ctx, hipMemcpy2DAsync(dst, pitch_bytes, src, (size_t)ctx->width * 4,
                                          (size_t)ctx->width * 4, ctx->out_rows, hipMemcpyDeviceToHost, ctx->stream));
        }
    LFI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LFI_OK;
}

int lfi_alloc_pinned(size_t bytes, void **out_ptr)
{
    if(!out_ptr || bytes == 0)
        return LFI_EINVAL;
    *out_ptr = nullptr;
    const hipError_t e = hipHostMalloc(out_ptr, bytes, hipHostMallocDefault);
    if(e != hipSuccess)
    {
        g_create_error = std::string("hipHostMalloc: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? LFI_ENOMEM : LFI_EHIP;
    }
    return LFI_OK;
}

int lfi_free_pinned(void *ptr)
{
    if(!ptr)
        return LFI_OK;
    return hipHostFree(ptr) == hipSuccess ? LFI_OK : LFI_EHIP;
}

int lfi_upload_map(lfi_ctx *ctx, int k, const uint8_t *rgba, size_t pitch_bytes)
{
    if(!ctx)
        return LFI_EINVAL;
    if(!ctx->maps)
        return fail(ctx, LFI_EINVAL, "lfi_set_grid has not been called");
    if(k < 0 || k > 1 || !rgba || pitch_bytes < (size_t)ctx->width * 4)
        return fail(ctx, LFI_EINVAL, "bad map index, pointer or pitch");
    if(int rc = bind(ctx))
        return rc;
    LFI_HIP(ctx, hipMemcpy2DAsync(ctx->maps + plane_bytes(ctx) * k, (size_t)ctx->width * 4, rgba, pitch_bytes,
                                  (size_t)ctx->width * 4, ctx->height, hipMemcpyHostToDevice, ctx->stream));
    LFI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LFI_OK;
}

const char *lfi_list_variants(int method)
{
    static std::string ten, std_;
    if(ten.empty())
    {
        for(int i = 0; i < kNumTenVariants; i++)
            ten += std::string(ten.empty() ? "" : ",") + kTenVariants[i].name;
        for(int i = 0; i < kNumStdVariants; i++)
            std_ += std::string(std_.empty() ? "" : ",") + kStdVariants[i].name;
    }
    if(method == LFI_METHOD_TEN_WM)
        return ten.c_str();
    if(method == LFI_METHOD_STD)
        return std_.c_str();
    if(method == LFI_KERNEL_FOCUS_ESTIMATE)
        return "factored,lds,packed_p2,plain";
    return "";
}

int lfi_set_variant(lfi_ctx *ctx, int method, const char *name)
{
    if(!ctx || !name)
        return LFI_EINVAL;
    const bool is_auto = std::strcmp(name, "auto") == 0;
    if(method == LFI_METHOD_TEN_WM)
    {
        if(is_auto)
        {
            ctx->ten_variant = 0;
            return LFI_OK;
        }
        for(int i = 0; i < kNumTenVariants; i++)
            if(std::strcmp(name, kTenVariants[i].name) == 0)
            {
                ctx->ten_variant = int(i);
                return LFI_OK;
            }
    }
    else if(method == LFI_METHOD_STD)
    {
        if(is_auto)
        {
            ctx->std_variant = 0;
            return LFI_OK;
        }
        for(int i = 0; i < kNumStdVariants; i++)
            if(std::strcmp(name, kStdVariants[i].name) == 0)
            {
                ctx->std_variant = int(i);
                return LFI_OK;
            }
    }
    else if(method == LFI_KERNEL_FOCUS_ESTIMATE)
    {
        static const char *const names[] = {"factored", "lds", "packed_p2", "plain"};
        if(is_auto)
        {
            ctx->focus_variant = 0;
            return LFI_OK;
        }
        for(int i = 0; i < 4; i++)
            if(std::strcmp(name, names[i]) == 0)
            {
                ctx->focus_variant = i;
                return LFI_OK;
            }
    }
    return fail(ctx, LFI_EINVAL, std::string("unknown kernel variant ") + name);
}

int lfi_download_coords(lfi_ctx *ctx, int g, int all_focus, int map_index, lfi_int2 *out_hw)
{
    if(!ctx)
        return LFI_EINVAL;
    if(!ctx->grid || !ctx->have_params)
        return fail(ctx, LFI_EINVAL, "lfi_set_grid / lfi_set_params have not been called");
    if(g < 0 || g >= ctx->n || !out_hw || map_index < 0 || map_index > 1)
        return fail(ctx, LFI_EINVAL, "bad image index, map index or pointer");
    if(ctx->windowed)
        return fail(ctx, LFI_EINVAL, "coordinate dumps are not supported with a row window");
    if(int rc = bind(ctx))
        return rc;
    const size_t bytes = sizeof(lfi_int2) * (size_t)ctx->width * ctx->height;
    lfi_int2 *d = nullptr;
    LFI_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&d), bytes));
    KernelArgs a = make_args(ctx, 0, ctx->views_n, LFI_METHOD_STD);
    a.map_index = map_index;
    hipLaunchKernelGGL(lfi::dump_coords, pixel_grid(ctx), dim3(256), 0, ctx->stream, a, g, all_focus, d);
    hipError_t e = hipGetLastError();
    if(e == hipSuccess)
        e = hipMemcpyAsync(out_hw, d, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if(e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    LFI_HIP(ctx, e);
    return LFI_OK;
}

int lfi_download_prequant(lfi_ctx *ctx, int method, int all_focus, int v, float *out_hw3)
{
    if(int rc = check_render_args(ctx, method, v, v + 1))
        return rc;
    if(!out_hw3)
        return fail(ctx, LFI_EINVAL, "out_hw3 is NULL");
    if(int rc = bind(ctx))
        return rc;
    const size_t bytes = sizeof(float) * 3 * (size_t)ctx->width * ctx->height;
    if(!ctx->prequant)
        LFI_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->prequant), bytes));
    KernelArgs a = make_args(ctx, v, v + 1, method);
    a.prequant = ctx->prequant;
    a.prequant_view = v;
    if(int rc = launch_blend(ctx, method, all_focus, a))
        return rc;
    LFI_HIP(ctx, hipMemcpyAsync(out_hw3, ctx->prequant, bytes, hipMemcpyDeviceToHost, ctx->stream));
    LFI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LFI_OK;
}

int lfi_debug_mfma_f16(lfi_ctx *ctx, const uint16_t *a_32x16, const uint16_t *b_16x32, float *c_32x32)
{
    if(!ctx || !a_32x16 || !b_16x32 || !c_32x32)
        return LFI_EINVAL;
    if(int rc = bind(ctx))
        return rc;
    uint8_t *d = nullptr;
    LFI_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&d), 1024 + 1024 + 4096));
    hipError_t e = hipMemcpyAsync(d, a_32x16, 1024, hipMemcpyHostToDevice, ctx->stream);
    if(e == hipSuccess)
        e = hipMemcpyAsync(d + 1024, b_16x32, 1024, hipMemcpyHostToDevice, ctx->stream);
    if(e == hipSuccess)
    {
        hipLaunchKernelGGL(lfi::probe_mfma_f16, dim3(1), dim3(64), 0, ctx->stream, reinterpret_cast<const uint16_t *>(d),
                           reinterpret_cast<const uint16_t *>(d + 1024), reinterpret_cast<float *>(d + 2048));
        e = hipGetLastError();
    }
    if(e == hipSuccess)
        e = hipMemcpyAsync(c_32x32, d + 2048, 4096, hipMemcpyDeviceToHost, ctx->stream);
    if(e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    LFI_HIP(ctx, e);
    return LFI_OK;
}

int lfi_debug_mfma_f16_chain(lfi_ctx *ctx, int shape, int k, const uint16_t *a_32xk, const uint16_t *b_kx32, float *c_32x32)
{
    if(!ctx || !a_32xk || !b_kx32 || !c_32x32 || (shape != 0 && shape != 1) || k < 32 || k > 256 || k % 32)
        return LFI_EINVAL;
    if(int rc = bind(ctx))
        return rc;
    const size_t ab = (size_t)32 * k * sizeof(uint16_t);
    uint8_t *d = nullptr;
    LFI_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&d), 2 * ab + 4096));
    hipError_t e = hipMemcpyAsync(d, a_32xk, ab, hipMemcpyHostToDevice, ctx->stream);
    if(e == hipSuccess)
        e = hipMemcpyAsync(d + ab, b_kx32, ab, hipMemcpyHostToDevice, ctx->stream);
    if(e == hipSuccess)
    {
        const uint16_t *da = reinterpret_cast<const uint16_t *>(d), *db = reinterpret_cast<const uint16_t *>(d + ab);
        float *dc = reinterpret_cast<float *>(d + 2 * ab);
        if(shape == 0)
            hipLaunchKernelGGL(lfi::probe_mfma_f16_chain<0>, dim3(1), dim3(64), 0, ctx->stream, da, db, k, dc);
        else
            hipLaunchKernelGGL(lfi::probe_mfma_f16_chain<1>, dim3(1), dim3(64), 0, ctx->stream, da, db, k, dc);
        e = hipGetLastError();
    }
    if(e == hipSuccess)
        e = hipMemcpyAsync(c_32x32, d + 2 * ab, 4096, hipMemcpyDeviceToHost, ctx->stream);
    if(e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    LFI_HIP(ctx, e);
    return LFI_OK;
}

} // extern "C"
