// lfi_context.hpp — the context behind include/lfi.h's opaque lfi_ctx: device memory, streams and events, the parameter block, and
// the small helpers every entry point uses (error convention, stream joins, plane sizes, the kernel-argument block).
// This is the device-facing state of the reference's Interpolator (reference src/interpolator.cu:36-154: surfaces, __constant__ symbols,
// the weights allocation) as one object per GPU.  Included by lfi_hip.hip only (one translation unit).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/lfi.h"
#include "lfi_device.hpp"
#include "quality.hpp"

using lfi::KernelArgs;

namespace {

thread_local std::string g_create_error;

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
    // the focus map's filter (map 0 → map 1) runs on the side stream behind the pick: an all-focus TEN_WM render, which reads map 0
    // (src/kernels.cu:430), does not wait for it; whatever reads map 1 or writes either map joins it first (join_filter)
    hipEvent_t ev_pick = nullptr, ev_filter = nullptr;
    bool filter_pending = false;
    int cols = 0, rows = 0, n = 0, width = 0, height = 0;
    // row window (lfi_set_row_window): input rows held / output rows rendered; the whole image by default
    int in_y0 = 0, in_rows = 0, out_y0 = 0, out_rows = 0;
    bool windowed = false;
    uint8_t *grid = nullptr;
    bool own_grid = false;
    size_t grid_bytes = 0;
    // lfi_release_inputs: the RGBA planes are gone, the derived planar copy is the only copy of the inputs (fixed-focus renders through
    // the planar kernels only); a later lfi_upload_image goes through a one-image staging plane straight into the copy
    bool inputs_released = false;
    uint8_t *stage_plane = nullptr;
    size_t stage_plane_bytes = 0;
    // set by lfi_prepare for a render that reads the planar copy: from then on images that arrive (lfi_upload_image, lfi_fill_synthetic_images)
    // refresh their planes of the copy AT ONCE instead of at the next render — the first render after a load is then only a launch
    bool eager_planar = false;
    uint8_t *maps = nullptr;
    uint8_t *views = nullptr;
    bool own_views = false;
    size_t views_bytes = 0;
    int out_layout = LFI_LAYOUT_RGBA;  // device layout of the views (lfi_set_output_layout)
    uint8_t *rgba_scratch = nullptr;   // planar layout: RGBA planes of all views for the kernels that only write RGBA (converted after the launch)
    size_t rgba_scratch_bytes = 0;
    uint8_t *dl_plane = nullptr;       // planar layout: one RGBA plane that downloads expand a view into
    size_t dl_plane_bytes = 0;
    uint8_t *quilt = nullptr;          // lfi_download_quilt[_tiles]: the quilt's rows of tiles as one RGBA image (grows, kept)
    size_t quilt_bytes = 0;
    // parameter block
    bool have_params = false;
    int views_n = 0, k_pad = 0, v_pad = 0, n_focus_ids = 0;
    void *param_blob = nullptr; // one allocation holding all parameter arrays
    size_t param_blob_bytes = 0;
    // lfi_set_params with an unchanged blob size (a focus sweep, a new trajectory with as many views): the new arrays go through one of two
    // page-locked staging buffers and a stream-ordered copy — no synchronisation, no allocation
    uint8_t *param_staging[2] = {nullptr, nullptr};
    size_t param_staging_bytes = 0;
    hipEvent_t ev_param[2] = {nullptr, nullptr};
    int param_slot = 0;
    // … and there are TWO copies of the arrays on the device: a replacement is copied into the idle one on the copy stream, beside the renders
    // still running from the other (round 5: in stream order behind them it cost a fixed-focus sweep 22 µs per step, profiles/r05_notes.md)
    size_t param_half_stride = 0;
    int param_half = 0;
    hipEvent_t ev_half_done[2] = {nullptr, nullptr}; // recorded on the compute stream when the context switches away from a half
    bool half_done_recorded[2] = {false, false};
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
    int planar_pitch = 0, planar_padx = 0, planar_reach = 0; // bytes per plane row; left padding; the largest |x offset| it was built for
    int32_t *d_planar_phase = nullptr;      // [LFI_MAX_IMAGES] per-image phase of the planar copy (device)
    std::vector<int32_t> planar_phase;      // the same on the host
    int32_t *phase_staging = nullptr;       // page-locked, 2 × LFI_MAX_IMAGES: a rebuild's phases go to the device in stream order, no host wait
    hipEvent_t ev_phase[2] = {nullptr, nullptr};
    int phase_slot = 0;
    std::vector<lfi_int2> h_focused;        // the integer offsets of the current parameters (host copy)
    unsigned launches_with_offsets = 0;     // fixed-focus launches since the integer offsets last changed
    uint64_t grid_version = 1, planar_version = 0;
    // which images changed: grid_full_version = grid_version at the last change that may have touched every image; img_version[g] = at the
    // last change of image g alone (lfi_upload_image[_async], lfi_fill_synthetic_images).  A derived copy brought up to date at version v
    // holds image g's current pixels iff max(img_version[g], grid_full_version) ≤ v: one replaced image costs one image's planes.
    uint64_t grid_full_version = 1;
    std::vector<uint64_t> img_version;
    bool grid_tracked = true; // every write to the planes goes through this library (or is announced by lfi_grid_modified)
    void *focus_ws = nullptr; // workspace of the factored focus-map estimate (plan, E, K), allocated on first use
    size_t focus_ws_bytes = 0;
    // the estimate's padded copies of the sampled images (focus_pad, the tail of focus_ws) depend on the inputs and on a BOUND of the
    // candidates' shifts only: kept between lfi_focus_map calls (a focus sweep over one light field — BASELINE config 5 — pads once)
    // while pad_version == grid_version, the same images are sampled with the same block radius, and pad_shift still covers the request
    uint64_t pad_version = 0;
    int pad_shift[2] = {0, 0}, pad_radius[2] = {0, 0};
    std::vector<int32_t> pad_ids, h_focus_ids;
    int ten_variant = 0, std_variant = 0, focus_variant = 0;
    mutable const char *last_kernel = ""; // the blend kernel the last render launched (lfi_last_kernel_name)
    mutable unsigned sweep_launches = 0;  // blend_p3 / blend_planar alternate their sweep direction from launch to launch
    float derived_build_ms = 0.0f;        // duration of the last planar_build (measured by lfi_prepare only)
    std::string err;
};

namespace {

// Every allocation of the views is ordinary device memory (hipMalloc) and goes back to the runtime with hipFree.  Rounds 2-3 placed the
// library's own planar views in hipDeviceMallocUncached memory (write-only planes that bypass the caches: -3.5 % at config 2 in round 2) and
// then had to keep every such block alive for the whole process, because contexts created after freed uncached ranges had been recycled as
// ordinary memory rendered garbage (profiles/r03_notes.md section 6).  Round 3's own A/B (profiles/r03_views_memory_ab.txt) shows no gain
// left from the placement outside box noise, so round 4 removed it together with the immortal pool: no allocation outlives its context.
// Measurement builds: LFI_VIEWS_MEMORY=uncached|finegrained still selects the other kinds for A/B runs.
hipError_t alloc_views(uint8_t **out, size_t bytes)
{
#ifdef LFI_MEASUREMENT_BUILD
    const char *e = std::getenv("LFI_VIEWS_MEMORY");
    if(e && std::strcmp(e, "uncached") == 0)
        return hipExtMallocWithFlags(reinterpret_cast<void **>(out), bytes, hipDeviceMallocUncached);
    if(e && std::strcmp(e, "finegrained") == 0)
        return hipExtMallocWithFlags(reinterpret_cast<void **>(out), bytes, hipDeviceMallocFinegrained);
#endif
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

// Order everything enqueued on the compute stream from now on after the focus-map filter that may still run on the side stream.
int join_filter(lfi_ctx *c)
{
    if(!c->filter_pending)
        return LFI_OK;
    LFI_HIP(c, hipStreamWaitEvent(c->stream, c->ev_filter, 0));
    c->filter_pending = false;
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

// planar view layout: bytes per row of a byte plane — a multiple of 128, so that every row starts on a cache line and every store
// instruction of blend_p3 (16 lanes × 8 bytes) writes ONE whole line.  With W rounded up to 16 only, a width that is not a multiple of 128
// put the rows at odd line phases and every store wrote two partial lines — into uncached memory: 1904 pixels 385 µs, 1936 pixels
// 366 µs per config-2-like launch against 153 µs at 1920 (tools/plane_skew.py, profiles/r03_plane_skew.txt).
int view_pitch(const lfi_ctx *c)
{
    return (c->width + 127) / 128 * 128;
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
    a.planar_phase = c->d_planar_phase;
    // blend_planar<STDF> (up to 64 images): chain bound N·2^-16 (half an ulp below 512 per fmaf: arithmetic) + the matrix core's accumulation
    // bound + 2^-11 of margin.  Accumulation bound: ANALYTIC by default since round 4, N·2^-15 (one whole fp16-product ulp per addend: true of
    // any accumulator that keeps ≥ 24 bits, nothing measured) — at these sizes the wider band costs nothing (tools/std_band_cost.py: config 2
    // 0.262 against 0.270 ms).  LFI_FLAG_STD_MEASURED_BAND: N·2^-17, a quarter ulp per addend, MEASURED on gfx950 (chains of
    // v_mfma_f32_32x32x16_f16 on operands built to expose alignment truncation stay within 0.086 ulp per addend;
    // tests/test_gpu_parity.py::test_mfma_f16_accumulation_error_bound asserts the quarter ulp) — rounds 2-3's default.
    a.std_band = float(c->n) * (0x1p-16f + 0x1p-15f) + 0x1p-11f;
    if((c->flags & LFI_FLAG_STD_MEASURED_BAND) && !(c->flags & LFI_FLAG_STD_ANALYTIC_BAND))
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


inline hipStream_t stream_of(const lfi_ctx *c) { return c->stream; }
inline void note_kernel(const lfi_ctx *c, const char *name) { c->last_kernel = name; }
inline uint32_t flags_of(const lfi_ctx *c) { return c->flags; }
inline dim3 pixel_grid_of(const lfi_ctx *c) { return pixel_grid(c); }
inline int cu_count_of(const lfi_ctx *c) { return c->cu_count; }

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

// every image may have changed / images [g0, g1) changed
void touch_all(lfi_ctx *c)
{
    c->grid_full_version = ++c->grid_version;
}

void touch_images(lfi_ctx *c, int g0, int g1)
{
    ++c->grid_version;
    if((int)c->img_version.size() != c->n)
    {
        c->img_version.assign(c->n, 0);
        c->grid_full_version = c->grid_version; // no per-image record yet
        return;
    }
    for(int g = g0; g < g1; g++)
        c->img_version[g] = c->grid_version;
}

bool image_changed_since(const lfi_ctx *c, int g, uint64_t version)
{
    // no per-image record (nothing but whole-grid changes so far: touch_images starts the record and bumps grid_full_version when it does):
    // the whole-grid version decides.  (Round 3 answered "changed" here, and contexts filled only by whole-grid calls — lfi_fill_synthetic_scene,
    // an attached grid + lfi_grid_modified — re-padded all sampled images on EVERY lfi_focus_map, one launch per image: 0.5 ms at 4K.)
    if((int)c->img_version.size() != c->n)
        return c->grid_full_version > version;
    return c->grid_full_version > version || c->img_version[g] > version;
}

void free_params(lfi_ctx *c)
{
    if(c->param_blob)
        (void)hipFree(c->param_blob);
    c->param_blob = nullptr;
    c->param_blob_bytes = 0;
    c->param_half = 0;
    c->half_done_recorded[0] = c->half_done_recorded[1] = false;
    c->have_params = false;
}

// the copy of the parameter arrays that launches enqueued from now on read
uint8_t *param_base(const lfi_ctx *c)
{
    return static_cast<uint8_t *>(c->param_blob) + (size_t)c->param_half * c->param_half_stride;
}

void free_param_staging(lfi_ctx *c)
{
    for(int i = 0; i < 2; i++)
    {
        if(c->param_staging[i])
            (void)hipHostFree(c->param_staging[i]);
        c->param_staging[i] = nullptr;
        if(c->ev_param[i])
            (void)hipEventDestroy(c->ev_param[i]);
        c->ev_param[i] = nullptr;
        if(c->ev_half_done[i])
            (void)hipEventDestroy(c->ev_half_done[i]);
        c->ev_half_done[i] = nullptr;
    }
    c->param_staging_bytes = 0;
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
    if(c->quilt)
        (void)hipFree(c->quilt);
    c->quilt = nullptr;
    c->quilt_bytes = 0;
    if(c->views2)
        (void)hipFree(c->views2);
    c->views2 = nullptr;
    c->views2_bytes = 0;
    if(c->quality_ref)
        (void)hipFree(c->quality_ref);
    c->quality_ref = nullptr;
    c->quality_ref_bytes = 0;
}

// the one-image staging plane of uploads after lfi_release_inputs is sized by the row window in force when it was allocated, and the eager
// refresh lfi_prepare switched on belongs to the planes it was switched on for: both go whenever the input planes are replaced
void drop_stage_plane(lfi_ctx *c)
{
    if(c->stage_plane)
        (void)hipFree(c->stage_plane);
    c->stage_plane = nullptr;
    c->stage_plane_bytes = 0;
    c->eager_planar = false;
}

void free_grid(lfi_ctx *c)
{
    if(c->own_grid && c->grid)
        (void)hipFree(c->grid);
    c->grid = nullptr;
    c->own_grid = false;
    c->grid_bytes = 0;
    c->inputs_released = false;
    drop_stage_plane(c);
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
    c->pad_version = 0;
    if(c->planar)
        (void)hipFree(c->planar);
    c->planar = nullptr;
    c->planar_bytes = 0;
    c->planar_version = 0;
    if(c->d_planar_phase)
        (void)hipFree(c->d_planar_phase);
    c->d_planar_phase = nullptr;
    if(c->phase_staging)
        (void)hipHostFree(c->phase_staging);
    c->phase_staging = nullptr;
    for(int i = 0; i < 2; i++)
    {
        if(c->ev_phase[i])
            (void)hipEventDestroy(c->ev_phase[i]);
        c->ev_phase[i] = nullptr;
    }
}

} // namespace
