/*
 * lfi.h — C-ABI of the MI355X light-field interpolation hot path (liblfi_hip.so).
 *
 * This is the drop-in boundary for the reference's device-facing call sites: everything
 * `Interpolator` (reference src/interpolator.cu) does through the CUDA runtime — surface allocation and upload,
 * cudaMemcpyToSymbol of the parameter block, the four kernel launches, event timing, download — is replaced by
 * the calls below.  Plain pointers and sizes only; no C++/torch types.  Each entry point cites the reference lines
 * it replaces (paths relative to the reference repository root).
 *
 * Conventions
 *  - every function returns 0 on success or a negative LFI_E* code; lfi_last_error() returns the message
 *    (the reference checks no CUDA return code at all: src/interpolator.cu:291-292);
 *  - one context per GPU, used from one thread at a time; work is enqueued on the context's HIP stream and is
 *    asynchronous unless stated otherwise;
 *  - images, views and maps are tightly packed RGBA8 planes (row pitch = width*4 bytes) — the linear-HBM
 *    replacement of the reference's cudaArray surfaces; image id g = col*rows + row (src/interpolator.cu:106-113);
 *  - there is no CPU fallback: a missing GPU or code object is an error.
 */
#ifndef LFI_H
#define LFI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LFI_ABI_VERSION 1

typedef struct lfi_ctx lfi_ctx;

typedef struct lfi_int2 { int32_t x, y; } lfi_int2;
typedef struct lfi_float2 { float x, y; } lfi_float2;

/* error codes */
enum {
    LFI_OK = 0,
    LFI_EINVAL = -1,    /* bad argument / call order */
    LFI_EHIP = -2,      /* a HIP runtime call failed */
    LFI_ENODEVICE = -3, /* no usable gfx950 device */
    LFI_ENOMEM = -4
};

/* interpolation methods: the reference's -m strings (src/interpolator.cu:274,282; src/main.cpp:20-22) */
enum {
    LFI_METHOD_STD = 0,    /* "STD":    exact-fp32 ordered FMA chain, RN-even quantisation (src/kernels.cu:289-343) */
    LFI_METHOD_TEN_WM = 1  /* "TEN_WM": fp16 matrix-core contraction, truncating quantisation (src/kernels.cu:345-462) */
};
#define LFI_KERNEL_FOCUS_ESTIMATE 2 /* not a render method: selects variants of FocusMap::estimate in lfi_set_variant */

/* lfi_params.flags */
enum {
    /* By default all-focus renders read the focus maps the reference's kernels read: Standard::process the filtered map 1
     * (src/kernels.cu:326), Tensors::process the unfiltered map 0 (src/kernels.cu:430) — an inconsistency of the reference
     * (SURVEY.md defect D7) that is reproduced so that outputs match it.  With this flag both methods read map 1. */
    LFI_FLAG_UNIFIED_FOCUS_MAP = 1u,
    /* TEN_WM debug numerics: re-round the accumulator to fp16 after every 16-image batch — the reference's half-accumulator WMMA model
     * (wmma::mma_sync with half fragments, src/kernels.cu:418-447; oracle model M16) instead of one final rounding.  Byte for byte the
     * oracle's M16 since round 5: the sums of a batch are formed exactly (fp64 on the vector pipe, one pixel per lane) and rounded once.
     * A debug mode: two orders of magnitude slower than the matrix-core kernels. */
    LFI_FLAG_TEN_ROUND_PER_BATCH = 2u,
    /* Fixed-focus launches over the planar input copy alternate their sweep direction from launch to launch, so that the input rows
     * one launch read last — still in the 256 MB Infinity Cache — are the first the next launch reads (repeated renders of one light
     * field: the reference's 100-launch loop, trajectory blocks, focus sweeps).  With this flag every launch walks the image in
     * ascending order, as a single cold launch does.  Results are identical either way. */
    LFI_FLAG_SINGLE_SWEEP_DIRECTION = 4u,
    /* STD through the band method (fp16 matrix-core sum, exact fmaf chain only for sums near x.5): size the band with the ANALYTIC
     * bound on the matrix core's accumulation error (one whole fp16-product ulp per addend, true of any accumulator that keeps ≥ 24
     * bits even if every addition truncated) instead of the bound measured on gfx950 (a quarter ulp per addend; asserted by
     * tests/test_gpu_parity.py::test_mfma_f16_accumulation_error_bound over 800 adversarial operand sets).  Same bytes, more sums
     * recomputed.  Light fields of up to 64 images use the analytic band BY DEFAULT since round 4 (it costs nothing there); with more
     * images (15x15 grids) it doubles the launch time (tools/std_band_cost.py), so there it stays opt-in, for callers who do not want
     * bit-exactness to rest on a measured property of the hardware. */
    LFI_FLAG_STD_ANALYTIC_BAND = 8u,
    /* Up to 64 images: the band of the measured bound (rounds 2-3's default) instead of the analytic one.  Same bytes. */
    LFI_FLAG_STD_MEASURED_BAND = 16u,
    /* Test hook of the band's self-check (lfi_std_band_info): behave as if the device had FAILED the measurement the measured bound
     * rests on — STD launches over more than 64 images then take the analytic band.  Same bytes. */
    LFI_FLAG_STD_BAND_PROBE_FAIL = 32u
};

#define LFI_MAX_IMAGES 256     /* MAX_IMAGES, src/kernels.cu:60 */
#define LFI_MAX_FOCUS_IDS 32   /* FOCUS_MAP_IDS_COUNT, src/kernels.cu:68 */
#define LFI_REFERENCE_VIEWS 64 /* VIEW_TOTAL_COUNT, src/kernels.cu:11-13 (a runtime parameter here) */

/*
 * The reference's __constant__ parameter block (src/kernels.cu:15-17, 63-69) as one struct.  All pointers are HOST
 * pointers; lfi_set_params copies what they point to.  The host code above this ABI computes these values
 * (lfinterpolator_amd/csrc/host: same arithmetic as src/interpolator.cu:139-246) so every backend sees identical bytes.
 */
typedef struct lfi_params {
    int32_t views;                    /* V: number of output views / rows of the weight matrix */
    const lfi_int2 *focused_offsets;  /* [N] round(offset*focus): focusedOffsets, src/interpolator.cu:241-244 */
    const lfi_float2 *offsets;        /* [N] offsets, src/interpolator.cu:240,245 */
    const uint16_t *weights_fp16;     /* [V][N] IEEE binary16 bit patterns, row-major: src/interpolator.cu:211-223 */
    const int32_t *focus_map_ids;     /* [n_focus_ids] focusMapIDs, src/interpolator.cu:203-206 (may be NULL if 0) */
    int32_t n_focus_ids;              /* ≤ LFI_MAX_FOCUS_IDS */
    float focus;                      /* inFocus, src/interpolator.cu:152 */
    float range;                      /* inRange, src/interpolator.cu:153 */
    int32_t block_radius[2];          /* constants[9..10], src/interpolator.cu:142-150 */
    uint32_t flags;                   /* LFI_FLAG_* (bit 31 is reserved for the library and ignored) */
} lfi_params;

typedef struct lfi_bench_stats {
    int32_t runs;
    float mean_ms;   /* mean of per-launch event times — what the reference prints (src/interpolator.cu:270-295) */
    float median_ms;
    float min_ms;
    float max_ms;
    float back_to_back_ms; /* (one event pair around `runs` consecutive launches) / runs */
} lfi_bench_stats;

/* ---- lifetime --------------------------------------------------------------------------------------------- */

/* Interpolator::Interpolator / init (src/interpolator.cu:36-50): bind to HIP device `device`, create the stream. */
int lfi_create(int device, lfi_ctx **out_ctx);
/* Interpolator::~Interpolator (src/interpolator.cu:41-44) — frees what the context owns; no device reset. */
int lfi_destroy(lfi_ctx *ctx);
/* message of the last failure on this context (ctx == NULL: last failure of lfi_create on this thread) */
const char *lfi_last_error(const lfi_ctx *ctx);
int lfi_abi_version(void);
/* number of visible HIP devices (≥0) or a negative error */
int lfi_device_count(void);

/* ---- light-field grid: replaces loadGPUData / createSurfaceObject / loadImageToArray (src/interpolator.cu:73-137) --- */

/* Declare a cols×rows grid of width×height RGBA8 images and allocate N = cols*rows input planes + 2 focus maps. */
int lfi_set_grid(lfi_ctx *ctx, int cols, int rows, int width, int height);
/* Row window for spatial (row-band) multi-GPU sharding — SURVEY.md §8(f).2; no counterpart in the reference.  After
 * lfi_set_grid: this context renders output rows [out_y0, out_y1) only and holds input rows [in_y0, in_y1) of every image only
 * (band + the halo the warp reaches into); planes shrink accordingly, so G GPUs each read and write ≈1/G of the bytes.  Host
 * pointers passed to upload / download / quilt calls keep addressing row 0 of the WHOLE image; attached device buffers hold
 * the window's rows only.  lfi_set_params verifies that the input rows cover every row a fixed-focus render samples; lfi_focus_map
 * (which then computes the band's rows of the maps: map 0 for the band plus the filter's reach, map 1 for the band) and all-focus
 * renders verify their own, larger reach — the warp at both ends of [focus, focus + range], plus the block radius for the map —
 * and fail with LFI_EINVAL if the held rows fall short (lfinterpolator_amd/sharding.py input_rows_all_focus computes them). */
int lfi_set_row_window(lfi_ctx *ctx, int out_y0, int out_y1, int in_y0, int in_y1);
/* cudaMemcpy2DToArray of one image (src/interpolator.cu:91): copies; the caller keeps ownership.  Synchronous. */
int lfi_upload_image(lfi_ctx *ctx, int g, const uint8_t *rgba, size_t pitch_bytes);
/* The same copy, asynchronous: enqueued on the context's copy stream — no synchronisation with the compute
 * stream per image, the GPU keeps rendering while images cross PCIe.  A page-locked source (lfi_alloc_pinned) is DMA'd in place:
 * the call returns at once and the buffer must stay valid until lfi_upload_wait / lfi_sync.  A pageable source is staged by the
 * HIP runtime before the call returns (the buffer is free on return; the host thread is busy for the copy's duration).
 * Everything that uses the planes afterwards (renders, focus map, fills) is ordered after the pending copies by an event,
 * without a host wait. */
int lfi_upload_image_async(lfi_ctx *ctx, int g, const uint8_t *rgba, size_t pitch_bytes);
/* host wait for the asynchronous uploads issued so far */
int lfi_upload_wait(lfi_ctx *ctx);
/* Use caller-owned device memory ([N][H][W][4] u8, ≥ N*H*W*4 bytes) for the input planes instead of the context's
 * own allocation — lets the caller fill it (e.g. an RCCL broadcast into a tensor it owns).  Call after lfi_set_grid. */
int lfi_attach_grid(lfi_ctx *ctx, void *device_ptr, size_t bytes);
/* Single-process multi-GPU: copy the input planes of ctxs[root] into every other context's planes with ONE RCCL broadcast over
 * xGMI (ncclCommInitAll + ncclBroadcast; RCCL is loaded on first use) — the light field then lives on every GPU and rendering
 * needs no further collective.  All contexts must sit on distinct devices and describe the same grid and row
 * window.  n == 1 is a no-op.  Synchronous.  (One process per GPU instead: broadcast the attached buffers with your own
 * communicator, as bench.py does through torch.distributed.) */
int lfi_broadcast_grid(lfi_ctx *const *ctxs, int n, int root);
/* Fixed-focus use only: make the derived planar copy of the inputs (3 bytes per pixel and image, built and tuned for the CURRENT
 * parameters' offsets now if it is not yet) the ONLY copy and free the RGBA planes — the inputs' footprint drops from 1.9x to 0.9x of
 * the RGBA bytes (BASELINE config 5: 14.2 -> 6.8 GB).  Afterwards: fixed-focus TEN_WM / STD renders through the default kernels are
 * served as before, for any parameters whose offsets the copy's padding covers (it is padded a quarter beyond the current ones);
 * lfi_upload_image[_async] replaces an image through a one-image staging plane (synchronously); everything that needs the RGBA planes
 * — lfi_focus_map, all-focus renders, debug modes, weights outside [0, 2), larger offsets, lfi_fill_synthetic*, lfi_grid_device_ptr,
 * lfi_broadcast_grid — is refused with LFI_EINVAL until lfi_set_grid / lfi_attach_grid start over.  An attached grid is only forgotten.
 * The reference keeps its inputs as cudaArrays for the lifetime of the object (src/interpolator.cu:73-93, loadGPUData :95-137). */
int lfi_release_inputs(lfi_ctx *ctx);

/* device pointer / size of the input planes currently in use */
int lfi_grid_device_ptr(lfi_ctx *ctx, void **out_ptr, size_t *out_bytes);
/* Tell the library that the contents of the input planes changed behind its back (a write through lfi_grid_device_ptr, or into
 * a buffer given to lfi_attach_grid — e.g. the RCCL broadcast bench.py does after attaching).  The renders keep a derived copy
 * of the inputs (planar, alpha dropped: 25 % fewer bytes to read per launch — DESIGN.md §4.1) which uploads through this API
 * invalidate by themselves.  A context whose planes are attached or whose pointer has been handed out reads the RGBA planes
 * directly on every launch, as the reference reads its surfaces, until this call has been made once: from then on the caller is
 * trusted to repeat it after every such write.  No counterpart in the reference (its inputs are immutable after loadGPUData). */
int lfi_grid_modified(lfi_ctx *ctx);
/* fill the input planes on the device with the synthetic light field of SURVEY.md §8(d):
 * byte = hash32(seed, g, y, x, c) >> 24, alpha 255 (identical to oracle lfo_fill_synthetic) */
int lfi_fill_synthetic(lfi_ctx *ctx, uint32_t seed);
/* the same for images [g0, g1) only — a rank of an all-gather distribution generates (or uploads) just its own slice */
int lfi_fill_synthetic_images(lfi_ctx *ctx, uint32_t seed, int g0, int g1);
/* a STRUCTURED synthetic light field for focus-map measurements: a texture of 8×8-pixel cells seen at a
 * piecewise-constant focus (1024×1024-pixel blocks, each at one of four of the estimate's candidates inside [focus, focus + range]
 * of the current parameters) — image g shows T(p − f*·offsets[g]), so the estimate finds a piecewise-constant map as in real
 * scenes (hash noise gives a noise map, and all-focus renders from a noise map gather one cache line per pixel).  Needs
 * lfi_set_params (offsets, focus, range).  Measurement only. */
int lfi_fill_synthetic_scene(lfi_ctx *ctx, uint32_t seed);

/* ---- parameters: replaces loadGPUOffsets / loadGPUWeights / selectFocusMapViews / loadGPUConstants
 *      (src/interpolator.cu:139-154, 194-246) ------------------------------------------------------------------
 * The arrays are copied before the call returns (the caller's memory is free again).  A call that keeps the number of views (a focus
 * sweep, another trajectory of the same length) replaces the device arrays IN STREAM ORDER through page-locked staging: renders already
 * enqueued keep the parameters they were enqueued with, later ones see the new ones, and the context's stream is not drained.  A call
 * that changes the number of views synchronises and reallocates. */
int lfi_set_params(lfi_ctx *ctx, const lfi_params *params);
/* Device layout of the rendered views.  LFI_LAYOUT_RGBA (default): [V][rows][W] RGBA8 dwords — the linear image of the
 * reference's 64 output surfaces.  LFI_LAYOUT_PLANAR_RGB (opt-in): alpha-free byte planes [V][3: R,G,B][rows][pitch] — the alpha
 * the reference's kernels write is the constant 255 (uchar4{…, 255}, src/kernels.cu:393, :309), a quarter of the bytes a render
 * writes; in this layout it is not stored and lfi_download_view / _quilt re-create it, so host-side results are byte-identical.
 * TEN_WM fixed-focus renders write the planes directly (csrc/hip/blend_p3.hpp); every other render goes through the RGBA kernels
 * and is converted.  Call after lfi_set_grid; frees the context's views (an attached buffer is dropped: attach again with the new
 * size); rows = the rows this context renders (all, or its row window). */
enum { LFI_LAYOUT_RGBA = 0, LFI_LAYOUT_PLANAR_RGB = 1 };
int lfi_set_output_layout(lfi_ctx *ctx, int layout);
typedef struct lfi_view_layout_info {
    int32_t layout;            /* LFI_LAYOUT_* */
    int32_t rows;              /* rows per plane */
    size_t row_pitch_bytes;    /* RGBA: W*4; planar: W rounded up to 128 (every plane row starts on a cache line) */
    size_t plane_stride_bytes; /* planar: bytes from a view's R plane to its G plane; RGBA: 0 */
    size_t view_stride_bytes;  /* bytes from view v to view v+1 */
} lfi_view_layout_info;
int lfi_view_layout(lfi_ctx *ctx, lfi_view_layout_info *out);
/* caller-owned device memory for the V views in the current layout (V * view_stride_bytes); call after lfi_set_params */
int lfi_attach_views(lfi_ctx *ctx, void *device_ptr, size_t bytes);
int lfi_views_device_ptr(lfi_ctx *ctx, void **out_ptr, size_t *out_bytes);

/* ---- kernels ------------------------------------------------------------------------------------------------ */

/* FocusMap::estimate + FocusMap::filter launches (src/interpolator.cu:261-266): fills maps 0 and 1.
 * The estimate reads edge-padded copies of the <= 32 sampled images; they depend on the inputs only (and on a bound of the shifts),
 * so they are kept between calls and rebuilt when the images change (any upload / fill through this library, lfi_grid_modified for
 * writes through the raw pointer), when other images are sampled, or when the shifts outgrow the padding: a focus sweep over one
 * light field (the reference's focusMapCompare.sh loop) pads once. */
int lfi_focus_map(lfi_ctx *ctx);
/* One launch of Tensors::process / Standard::process (src/interpolator.cu:274-288) for views [v0, v1).
 * all_focus != 0 selects the <true> instantiations (per-pixel focus from the focus map). */
int lfi_render(lfi_ctx *ctx, int method, int all_focus, int v0, int v1);
/* Trajectory streaming: renders a camera path of total_views views — weights_fp16 is [total_views][N], built
 * for the WHOLE path by the host code (one trajectory centre, so the offsets set by lfi_set_params stay valid) — in blocks of V
 * views (V = lfi_params.views) against the resident inputs, with no host synchronisation between blocks: block b+1's weight
 * arrays are prepared on the host and copied while block b renders (page-locked double-buffered staging, stream-ordered device
 * copy), and, when host_out != NULL, block b's views are copied to host_out + (b·V + v)·pitch_bytes·H on a second stream from a
 * second set of views while block b+1 renders.  host_out should be page-locked (lfi_alloc_pinned) and needs the RGBA view
 * layout; NULL renders only (the views of the last block remain on the device).  Returns after everything has completed.
 * Afterwards the context's weights are those of the last block.  The reference renders exactly 64 views per run
 * (src/kernels.cu:11-14); this is its loop over successive 64-view segments of a longer path. */
int lfi_render_stream(lfi_ctx *ctx, int method, int all_focus, const uint16_t *weights_fp16, int total_views, uint8_t *host_out, size_t pitch_bytes);
/* Do now what the first such lfi_render would otherwise do before its launch: (re)build the derived, alpha-free planar copy of
 * the inputs if that launch would read it, and time it.  Optional; synchronous.  No counterpart in the
 * reference (its surfaces are read as uploaded). */
int lfi_prepare(lfi_ctx *ctx, int method, int all_focus, int v0, int v1);
/* device memory this context holds, and what the derived copy cost to build (ms, as measured by the last lfi_prepare that
 * built it; 0 otherwise) — the capacity side of the planar-copy trade, reported by bench.py */
typedef struct lfi_memory {
    size_t grid_bytes;      /* input planes (RGBA) */
    size_t derived_bytes;   /* planar copy of the inputs: 3 bytes per pixel and image (+ padding) */
    size_t views_bytes;     /* output planes */
    size_t maps_bytes;      /* focus maps */
    size_t workspace_bytes; /* focus-map workspace + (planar view layout) the RGBA scratch copy of the views that renders other than TEN_WM and
                             * STD on more than 64 images go through, and the one-plane staging buffer of downloads */
    float derived_build_ms;
} lfi_memory;
int lfi_memory_info(lfi_ctx *ctx, lfi_memory *out);
/* The band method's self-check (round 5).  STD over more than 64 images (Standard::process, src/kernels.cu:289-343, computed as fp16
 * matrix-core sums + the exact fmaf chain inside a band around x.5) sizes that band with a bound on the matrix core's accumulation error
 * that was MEASURED on gfx950 (a quarter ulp per addend).  The first such launch on a device therefore repeats the measurement there —
 * chains of the MFMA instructions the kernels use over adversarial operand sets, exact sums on the host, once per device and process —
 * and if any sum exceeds the budget every such launch on that device takes the analytic band instead (LFI_FLAG_STD_ANALYTIC_BAND's:
 * same bytes, more sums recomputed).  This call runs the check if it has not run yet and reports it. */
typedef struct lfi_std_band {
    int32_t probed;           /* 1 once the device has been measured */
    int32_t within_budget;    /* 1: every sum within N*2^-17 of the exact one */
    int32_t analytic_forced;  /* 1: this context's STD launches over more than 64 images take the analytic band although the caller
                                 did not ask for it (the device failed the check, or LFI_FLAG_STD_BAND_PROBE_FAIL) */
    int32_t sums;             /* sums checked */
    float worst_fraction;     /* the largest error seen, as a fraction of the budget */
    float probe_ms;           /* what the check cost, once per device (host wall clock) */
    char message[160];
} lfi_std_band;
int lfi_std_band_info(lfi_ctx *ctx, lfi_std_band *out);
/* name of the blend kernel the last lfi_render / lfi_benchmark of this context launched ("" before the first) — lets the
 * measurement harness label its numbers with what actually ran (dispatch depends on shape, weights and mode) */
const char *lfi_last_kernel_name(const lfi_ctx *ctx);
/* The reference's benchmark loop (src/interpolator.cu:270-295) with warm-up launches excluded. Synchronous. */
int lfi_benchmark(lfi_ctx *ctx, int method, int all_focus, int v0, int v1, int warmup, int runs,
                  lfi_bench_stats *out_stats);
/* hipEvent pair on the context's stream (the reference's Timer, src/interpolator.cu:13-34) */
int lfi_timer_start(lfi_ctx *ctx);
int lfi_timer_stop(lfi_ctx *ctx, float *out_ms); /* records, synchronises, returns elapsed ms */
int lfi_sync(lfi_ctx *ctx);

/* ---- results: replaces storeResults' cudaMemcpy2DFromArray (src/interpolator.cu:309).  Synchronous. --------- */
int lfi_download_view(lfi_ctx *ctx, int v, uint8_t *rgba, size_t pitch_bytes);
int lfi_download_map(lfi_ctx *ctx, int k, uint8_t *rgba, size_t pitch_bytes);
/* Views v0 … v0+tiles_x*tiles_y-1 as ONE image of tiles_x × tiles_y tiles, filled left to right, top to bottom — what
 * scripts/viewsToQuilt.sh builds with ImageMagick `montage -tile 5x9` from the NN.png files (Looking-Glass quilt).
 * rgba: (tiles_y*H) rows of pitch_bytes ≥ tiles_x*W*4.  Synchronous. */
int lfi_download_quilt(lfi_ctx *ctx, int tiles_x, int tiles_y, int v0, uint8_t *rgba, size_t pitch_bytes);
/* The same for PART of a quilt: views v0 … v0+n-1 of this context become tiles first_tile … first_tile+n-1 (row-major) of the
 * tiles_x × tiles_y quilt whose top-left pixel is at rgba — a trajectory sharded over several GPUs (src/interpolator.cu has one GPU;
 * the CLI's -g): every context fills its own tiles of one host image.  The tiles are assembled on the device by one kernel (the
 * planar view layout is expanded on the fly) and copied in at most three rectangles; lfi_download_quilt is this with every tile. */
int lfi_download_quilt_tiles(lfi_ctx *ctx, int tiles_x, int tiles_y, int first_tile, int n, int v0, uint8_t *rgba, size_t pitch_bytes);
int lfi_upload_map(lfi_ctx *ctx, int k, const uint8_t *rgba, size_t pitch_bytes); /* tests: inject a focus map */

/* PSNR / SSIM of rendered view v against a reference image on the host, reduced on the device — replaces
 * scripts/imageQualityMetrics.sh:1-12 (ffmpeg psnr / ssim on two PNGs).  Definitions (csrc/hip/quality.hpp): PSNR per colour
 * channel from the mean squared error over all pixels, "all" from the mean of the three MSEs; SSIM per channel = mean over all 8×8
 * windows at stride 4 of the standard index with C1 = (0.01·255)², C2 = (0.03·255)², "all" = mean of the channels.  Synchronous. */
typedef struct lfi_quality {
    double mse[3], psnr[3], psnr_all; /* identical images: mse 0, psnr +inf */
    double ssim[3], ssim_all;
} lfi_quality;
int lfi_compare_view(lfi_ctx *ctx, int v, const uint8_t *reference_rgba, size_t pitch_bytes, lfi_quality *out);

/* Page-locked host memory for uploads / downloads at full PCIe rate (hipHostMalloc); optional — any host pointer works. */
int lfi_alloc_pinned(size_t bytes, void **out_ptr);
int lfi_free_pinned(void *ptr);

/* ---- plumbing ------------------------------------------------------------------------------------------------- */

/* enqueue on a caller-owned hipStream_t (NULL restores the context's own stream).  Work already enqueued on the stream in use so
 * far (fills, uploads, the derived input copy, focus maps, renders) is ordered before everything enqueued on the new one by an
 * event — no host synchronisation; the stream switched away from must still exist at the time of the call. */
int lfi_set_stream(lfi_ctx *ctx, void *hip_stream);
/* choose a kernel variant by name for a method — or for the focus-map estimate with LFI_KERNEL_FOCUS_ESTIMATE —
 * ("auto" = default); used by the benchmark harness and the parity tests */
int lfi_set_variant(lfi_ctx *ctx, int method, const char *name);
/* comma separated variant names available for a method */
const char *lfi_list_variants(int method);

/* ---- debug / parity hooks --------------------------------------------------------------------------------------- */

/* unclamped warped sample coordinates of image g for every pixel ([H][W] int2), computed on the device:
 * focusCoords (src/kernels.cu:72-82).  Synchronous. */
int lfi_download_coords(lfi_ctx *ctx, int g, int all_focus, int map_index, lfi_int2 *out_hw);
/* accumulator values before quantisation ([H][W][3] float) of view v: fp32 sums for STD, the fp16-rounded value for
 * TEN_WM.  Renders view v again into a scratch buffer.  Synchronous. */
int lfi_download_prequant(lfi_ctx *ctx, int method, int all_focus, int v, float *out_hw3);
/* hardware probe: C[32x32] = A[32x16] (fp16 bits) · B[16x32] (fp16 bits) with one v_mfma_f32_32x32x16_f16,
 * row-major in/out — checks the fragment lane maps and fp16-subnormal handling with exact data */
int lfi_debug_mfma_f16(lfi_ctx *ctx, const uint16_t *a_32x16, const uint16_t *b_16x32, float *c_32x32);
/* hardware probe: gfx950's three-operand packed fp16 minimum / maximum (v_pk_minimum3_f16 / v_pk_maximum3_f16) on u16 lanes that
 * hold bytes, i.e. fp16 subnormal bit patterns, over all 256³ byte triples (twice: once per half) against integer min / max;
 * *out_mismatches = the number of halves that differ (0 on hardware that leaves subnormals alone).  The focus-map range passes
 * reduce two views per instruction with them (csrc/hip/focus_factored.hpp; FocusMap::ElementRange, src/kernels.cu:173-194). */
int lfi_debug_pk_minmax3_f16(lfi_ctx *ctx, uint32_t *out_mismatches);
/* hardware probe: C[32x32] = A[32xk] · B[kx32] accumulated as the kernels accumulate — shape 0: k/16 chained
 * v_mfma_f32_32x32x16_f16, shape 1: k/32 chained v_mfma_f32_16x16x32_f16 per quadrant; k a multiple of 32, ≤ 256.  Measures the
 * matrix pipe's fp32 accumulation error, which the default STD kernel's rounding band assumes a bound for. */
int lfi_debug_mfma_f16_chain(lfi_ctx *ctx, int shape, int k, const uint16_t *a_32xk, const uint16_t *b_kx32, float *c_32x32);

#ifdef __cplusplus
}
#endif
#endif /* LFI_H */
