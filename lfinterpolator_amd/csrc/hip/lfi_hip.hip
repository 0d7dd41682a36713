// lfi_hip.hip — the C-ABI of include/lfi.h on HIP for gfx950: every extern "C" entry point.  This translation unit is the whole
// device-facing half of the reference's Interpolator (reference src/interpolator.cu:13-154, 194-316); the arithmetic that produces
// the parameters stays in the host code above the ABI (lfinterpolator_amd/csrc/host).  Its parts:
//   lfi_context.hpp      the context (device memory, streams, parameter block) and its helpers
//   lfi_dispatch.hpp     variant tables, kernel launchers, the derived planar copy, launch_blend
//   lfi_focus_sched.hpp  the focus-map estimate's workspace and pass graph
//   lfi_rccl.hpp         RCCL loader for lfi_broadcast_grid
#include "lfi_context.hpp"
#include "lfi_dispatch.hpp"
#include "lfi_focus_sched.hpp"
#include "lfi_rccl.hpp"

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
    if(ctx->aux_stream)
        (void)hipStreamSynchronize(ctx->aux_stream); // a focus-map filter may still run there
    ctx->filter_pending = false;
    free_params(ctx);
    free_param_staging(ctx);
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
    if(ctx->ev_pick)
        (void)hipEventDestroy(ctx->ev_pick);
    if(ctx->ev_filter)
        (void)hipEventDestroy(ctx->ev_filter);
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
    if(int rc = join_filter(ctx))
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
    // device addressing: pixel indices are 32-bit (W·H ≤ 2^26 leaves room for the ×4 bytes and the per-image strides the kernels fold
    // into scalar bases), and rows / columns are multiplied with 24-bit multiplies (__umul24: blend_persist all-focus, blend_p3)
    if((size_t)width * height > (size_t)1 << 26 || width >= (1 << 24) || height >= (1 << 24))
        return fail(ctx, LFI_EINVAL, "image too large (width·height ≤ 2^26 and width, height < 2^24)");
    if(int rc = bind(ctx))
        return rc;
    if(int rc = lfi_upload_wait(ctx))
        return rc;
    if(int rc = join_filter(ctx))
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
    touch_all(ctx);
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
    ctx->inputs_released = false;
    drop_stage_plane(ctx); // sized for the window in force when it was allocated (a larger window overflowed it)
    ctx->in_y0 = in_y0;
    ctx->in_rows = in_y1 - in_y0;
    ctx->out_y0 = out_y0;
    ctx->out_rows = out_y1 - out_y0;
    ctx->windowed = !(in_y0 == 0 && in_y1 == ctx->height && out_y0 == 0 && out_y1 == ctx->height);
    ctx->grid_bytes = in_plane_bytes(ctx) * ctx->n;
    LFI_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->grid), ctx->grid_bytes));
    ctx->own_grid = true;
    touch_all(ctx);
    ctx->grid_tracked = true;
    return LFI_OK;
}

int lfi_upload_image(lfi_ctx *ctx, int g, const uint8_t *rgba, size_t pitch_bytes)
{
    if(!ctx)
        return LFI_EINVAL;
    if(!ctx->grid && !ctx->inputs_released)
        return fail(ctx, LFI_EINVAL, "lfi_set_grid has not been called");
    if(g < 0 || g >= ctx->n || !rgba || pitch_bytes < (size_t)ctx->width * 4)
        return fail(ctx, LFI_EINVAL, "bad image index, pointer or pitch");
    if(int rc = bind(ctx))
        return rc;
    if(int rc = join_uploads(ctx))
        return rc;
    if(ctx->inputs_released)
    {
        // the RGBA planes are gone: the image goes through a one-image staging plane straight into its planes of the planar copy
        if(ctx->stage_plane && ctx->stage_plane_bytes < in_plane_bytes(ctx))
            drop_stage_plane(ctx);
        if(!ctx->stage_plane)
        {
            LFI_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->stage_plane), in_plane_bytes(ctx)));
            ctx->stage_plane_bytes = in_plane_bytes(ctx);
        }
        LFI_HIP(ctx, hipMemcpy2DAsync(ctx->stage_plane, (size_t)ctx->width * 4, rgba + (size_t)ctx->in_y0 * pitch_bytes, pitch_bytes, (size_t)ctx->width * 4,
                                      ctx->in_rows, hipMemcpyHostToDevice, ctx->stream));
        // planar_build reads image g at grid + g·plane: hand it a base that puts the staging plane there
        hipLaunchKernelGGL(lfi::planar_build, dim3((ctx->planar_pitch / 4 + 255) / 256, ctx->in_rows, 1), dim3(256), 0, ctx->stream,
                           ctx->stage_plane - in_plane_bytes(ctx) * (size_t)g, ctx->planar, ctx->width, ctx->in_rows, ctx->planar_pitch, ctx->planar_padx,
                           ctx->d_planar_phase, g);
        LFI_HIP(ctx, hipGetLastError());
        LFI_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return LFI_OK;
    }
    // rgba addresses row 0 of the whole image; only the rows this context holds are copied
    LFI_HIP(ctx, hipMemcpy2DAsync(ctx->grid + in_plane_bytes(ctx) * g, (size_t)ctx->width * 4, rgba + (size_t)ctx->in_y0 * pitch_bytes,
                                  pitch_bytes, (size_t)ctx->width * 4, ctx->in_rows, hipMemcpyHostToDevice, ctx->stream));
    LFI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    touch_images(ctx, g, g + 1);
    if(ctx->eager_planar && ctx->have_params)
        (void)ensure_planar(ctx); // this image's planes of the derived copy, now (stream-ordered; a failure leaves it to the next render)
    return LFI_OK;
}

int lfi_upload_image_async(lfi_ctx *ctx, int g, const uint8_t *rgba, size_t pitch_bytes)
{
    if(!ctx)
        return LFI_EINVAL;
    if(ctx->inputs_released)
        return lfi_upload_image(ctx, g, rgba, pitch_bytes); // released inputs: through the staging plane, synchronously
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
    touch_images(ctx, g, g + 1);
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
    ctx->inputs_released = false;
    drop_stage_plane(ctx);
    ctx->grid_bytes = bytes;
    touch_all(ctx);
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
        touch_all(ctxs[i]);
    }
    if(caller_device >= 0)
        (void)hipSetDevice(caller_device);
    return status;
}

int lfi_release_inputs(lfi_ctx *ctx)
{
    if(!ctx)
        return LFI_EINVAL;
    if(ctx->inputs_released)
        return LFI_OK;
    if(!ctx->grid || !ctx->have_params)
        return fail(ctx, LFI_EINVAL, "lfi_set_grid / lfi_set_params have not been called");
    if(int rc = bind(ctx))
        return rc;
    if(int rc = lfi_upload_wait(ctx))
        return rc;
    // the planar copy, complete and tuned for the current offsets, becomes the only copy of the inputs
    if(!ensure_planar(ctx, true))
        return fail(ctx, LFI_EINVAL, "the planar copy of the inputs cannot be built (inputs the library does not track, absurd offsets, out of memory)");
    LFI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if(ctx->own_grid && ctx->grid)
        LFI_HIP(ctx, hipFree(ctx->grid));
    ctx->grid = nullptr;
    ctx->own_grid = false;
    ctx->grid_bytes = 0;
    ctx->inputs_released = true;
    return LFI_OK;
}

int lfi_grid_device_ptr(lfi_ctx *ctx, void **out_ptr, size_t *out_bytes)
{
    if(!ctx || !out_ptr)
        return LFI_EINVAL;
    if(ctx->inputs_released) // before anything is touched: the query must not cost the context its only copy of the inputs
        return fail(ctx, LFI_EINVAL, "the RGBA inputs were released (lfi_release_inputs): lfi_set_grid and upload the images again");
    if(ctx->uploads_pending && bind(ctx) == LFI_OK)
        (void)lfi_upload_wait(ctx); // the caller is about to read or write the planes itself
    *out_ptr = ctx->grid;
    if(out_bytes)
        *out_bytes = ctx->grid ? in_plane_bytes(ctx) * ctx->n : 0;
    ctx->grid_tracked = false; // the caller may write through the pointer: see lfi_grid_modified
    touch_all(ctx);
    return LFI_OK;
}

int lfi_grid_modified(lfi_ctx *ctx)
{
    if(!ctx)
        return LFI_EINVAL;
    if(ctx->inputs_released)
        return fail(ctx, LFI_EINVAL, "the RGBA inputs were released (lfi_release_inputs): lfi_set_grid and upload the images again");
    if(!ctx->grid)
        return fail(ctx, LFI_EINVAL, "lfi_set_grid has not been called");
    touch_all(ctx);
    ctx->grid_tracked = true; // the caller announces its writes from now on
    return LFI_OK;
}

int lfi_fill_synthetic_images(lfi_ctx *ctx, uint32_t seed, int g0, int g1)
{
    if(!ctx)
        return LFI_EINVAL;
    if(ctx->inputs_released)
        return fail(ctx, LFI_EINVAL, "the RGBA inputs were released (lfi_release_inputs): lfi_set_grid and upload the images again");
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
    touch_images(ctx, g0, g1);
    hipLaunchKernelGGL(lfi::fill_synthetic, dim3(256 * 16), dim3(256), 0, ctx->stream, ctx->grid, g0, g1 - g0, ctx->width,
                       ctx->in_rows, ctx->in_y0, seed);
    LFI_HIP(ctx, hipGetLastError());
    if(ctx->eager_planar && ctx->have_params)
        (void)ensure_planar(ctx); // as lfi_upload_image
    return LFI_OK;
}

int lfi_fill_synthetic_scene(lfi_ctx *ctx, uint32_t seed)
{
    if(!ctx)
        return LFI_EINVAL;
    if(ctx->inputs_released)
        return fail(ctx, LFI_EINVAL, "the RGBA inputs were released (lfi_release_inputs): lfi_set_grid and upload the images again");
    if(!ctx->grid || !ctx->have_params)
        return fail(ctx, LFI_EINVAL, "lfi_set_grid / lfi_set_params have not been called (the scene is built from the images' offsets)");
    if(int rc = bind(ctx))
        return rc;
    touch_all(ctx);
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
    // Same blob size as the arrays in place (and as many views: nothing else to reallocate): stage in page-locked memory and copy in
    // stream order — launches already enqueued keep reading the old arrays, later ones see the new; the context's stream is not
    // drained.  Otherwise (first call, another view count or grid): synchronise, reallocate, copy synchronously.
    const bool in_place = ctx->param_blob && ctx->param_blob_bytes == total && !views_changed && ctx->views;
    std::vector<uint8_t> blob_pageable;
    uint8_t *blob_ptr = nullptr;
    if(in_place)
    {
        if(ctx->param_staging_bytes != total)
        {
            free_param_staging(ctx);
            for(int i = 0; i < 2; i++)
            {
                LFI_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->param_staging[i]), total, hipHostMallocDefault));
                LFI_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_param[i], hipEventDisableTiming));
                LFI_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_half_done[i], hipEventDisableTiming));
            }
            ctx->param_staging_bytes = total;
        }
        else
            LFI_HIP(ctx, hipEventSynchronize(ctx->ev_param[ctx->param_slot])); // the copy out of this buffer, two calls ago, has run
        blob_ptr = ctx->param_staging[ctx->param_slot];
        std::memset(blob_ptr, 0, total);
    }
    else
    {
        LFI_HIP(ctx, hipStreamSynchronize(ctx->stream));
        blob_pageable.assign(total, 0);
        blob_ptr = blob_pageable.data();
    }
    struct
    {
        uint8_t *p;
        uint8_t *data() const { return p; }
    } blob{blob_ptr};
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

    if(in_place)
    {
        // into the device copy no launch reads any more, on the copy stream: beside the renders still running from the other copy.  Order:
        // the launches that read the idle copy were all enqueued before the context switched away from it (ev_half_done, recorded then);
        // launches enqueued from now on wait for this copy.
        if(int rc = ensure_copy_stream(ctx))
            return rc;
        const int cur = ctx->param_half, next = cur ^ 1;
        LFI_HIP(ctx, hipEventRecord(ctx->ev_half_done[cur], ctx->stream));
        ctx->half_done_recorded[cur] = true;
        if(ctx->half_done_recorded[next])
            LFI_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_half_done[next], 0));
        LFI_HIP(ctx, hipMemcpyAsync(static_cast<uint8_t *>(ctx->param_blob) + (size_t)next * ctx->param_half_stride, blob.data(), total, hipMemcpyHostToDevice,
                                    ctx->copy_stream));
        LFI_HIP(ctx, hipEventRecord(ctx->ev_param[ctx->param_slot], ctx->copy_stream));
        LFI_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_param[ctx->param_slot], 0));
        ctx->param_slot ^= 1;
        ctx->param_half = next;
    }
    else
    {
        free_params(ctx);
        ctx->param_half_stride = (total + 255) / 256 * 256;
        LFI_HIP(ctx, hipMalloc(&ctx->param_blob, 2 * ctx->param_half_stride));
        ctx->param_blob_bytes = total;
        LFI_HIP(ctx, hipMemcpy(ctx->param_blob, blob.data(), total, hipMemcpyHostToDevice));
    }
    uint8_t *base = param_base(ctx);
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
    if((int)ctx->h_focused.size() != n || std::memcmp(ctx->h_focused.data(), p->focused_offsets, sizeof(lfi_int2) * n) != 0)
        ctx->launches_with_offsets = 0; // new integer offsets: the planar copy's phases may no longer fit them (ensure_planar)
    ctx->h_focused.assign(p->focused_offsets, p->focused_offsets + n);
    ctx->h_focus_offsets.clear();
    for(int k = 0; k < p->n_focus_ids; k++)
        ctx->h_focus_offsets.push_back(p->offsets[p->focus_map_ids[k]]);
    ctx->h_focus_ids.assign(p->focus_map_ids, p->focus_map_ids + p->n_focus_ids);
    ctx->focus = p->focus;
    ctx->range = p->range;
    for(int d = 0; d < 2; d++)
    {
        ctx->fo_min[d] = fo_min[d];
        ctx->fo_max[d] = fo_max[d];
    }
    ctx->radius[0] = std::max(p->block_radius[0], 1);
    ctx->radius[1] = std::max(p->block_radius[1], 1);
    ctx->flags = p->flags & 0x7fffffffu; // (bit 31 is the dispatcher's own: LFI_KFLAG_PLAIN_TILE_ORDER, lfi_device.hpp)

    if(views_changed || !ctx->views)
    {
        const bool was_attached = ctx->views && !ctx->own_views;
        if(!(was_attached && ctx->views_bytes >= out_plane_bytes(ctx) * V))
        {
            free_views(ctx);
            ctx->views_bytes = out_plane_bytes(ctx) * V;
            ctx->have_params = false; // until the views exist: a failed allocation must not leave renders a null pointer to store through
            LFI_HIP(ctx, alloc_views(&ctx->views, ctx->views_bytes));
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
        ctx->have_params = false; // a failed allocation leaves the context without parameters (lfi_set_params again), not with null views
        LFI_HIP(ctx, alloc_views(&ctx->views, ctx->views_bytes));
        ctx->own_views = true;
        ctx->have_params = true;
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

// map 1 = the box mean of map 0 over the rows a.map_y0 … + a.map_rows: from LDS (focus_filter_tiled) when the window allows it
static void launch_focus_filter(lfi_ctx *ctx, const lfi::KernelArgs &a, hipStream_t st)
{
    const int rx = std::max(ctx->radius[0] / 10, 1), ry = std::max(ctx->radius[1] / 10, 1);
    const size_t lds = lfi::focus_filter_tiled_lds(rx, ry);
    if(rx <= 128 && ry <= 128 && lds <= 64u * 1024u)
        hipLaunchKernelGGL(lfi::focus_filter_tiled, dim3((ctx->width + lfi::FF_TW - 1) / lfi::FF_TW, (a.map_rows + lfi::FF_TH - 1) / lfi::FF_TH), dim3(256), lds, st, a);
    else
        hipLaunchKernelGGL(lfi::focus_filter, dim3((ctx->width + 63) / 64, (a.map_rows + 3) / 4), dim3(256), 0, st, a);
}

int lfi_focus_map(lfi_ctx *ctx)
{
    if(!ctx)
        return LFI_EINVAL;
    if(ctx->inputs_released)
        return fail(ctx, LFI_EINVAL, "the RGBA inputs were released (lfi_release_inputs): the focus map needs them - upload the images again (lfi_set_grid)");
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
    if(int rc = join_filter(ctx)) // the previous map's filter still reads map 0, which this call rewrites
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
        launch_focus_filter(ctx, a, ctx->stream);
        LFI_HIP(ctx, hipGetLastError());
        return LFI_OK;
    }
    // the LDS-staged kernel needs its window (128 + 2·radius_x + slack pixels) to fit a 256-pixel LDS row
    const bool lds_fits = 128 + 2 * ctx->radius[0] + 2 * lfi::FOCUS_LDS_SLACK <= lfi::FOCUS_LDS_ROW;
    // "factored" (default): W and H must fit the 16-bit column / row lists
    bool done = false;
    if((ctx->focus_variant == 0 || ctx->focus_variant == 4) && ctx->width <= 65535 && ctx->height <= 65535)
        if(int rc = launch_focus_factored(ctx, a, &done, ctx->focus_variant == 4))
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
    if(done && ctx->aux_stream)
    {
        // map 1 = the box mean of map 0 (FocusMap::filter, src/kernels.cu:260-280), on the side stream: the reference's Tensors::process<true>
        // reads map 0 (src/kernels.cu:430), so a TEN_WM all-focus render enqueued next runs BESIDE the filter (0.13 ms at 4K) instead of
        // behind it; everything that reads map 1 or writes a map joins the side stream first (join_filter)
        if(!ctx->ev_pick)
        {
            LFI_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_pick, hipEventDisableTiming));
            LFI_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_filter, hipEventDisableTiming));
        }
        LFI_HIP(ctx, hipEventRecord(ctx->ev_pick, ctx->stream));
        LFI_HIP(ctx, hipStreamWaitEvent(ctx->aux_stream, ctx->ev_pick, 0));
        launch_focus_filter(ctx, a, ctx->aux_stream);
        LFI_HIP(ctx, hipGetLastError());
        LFI_HIP(ctx, hipEventRecord(ctx->ev_filter, ctx->aux_stream));
        ctx->filter_pending = true;
        return LFI_OK;
    }
    launch_focus_filter(ctx, a, ctx->stream);
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
    if(wants_derived_copy(ctx, method, all_focus, a)) // the same predicate chain as launch_blend
    {
        ctx->eager_planar = true; // images that arrive from now on refresh their planes of the copy at once
        const uint64_t before = ctx->planar_version;
        LFI_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        const bool ok = ensure_planar(ctx, true);
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
    out->workspace_bytes = ctx->focus_ws_bytes + ctx->rgba_scratch_bytes + ctx->dl_plane_bytes;
    out->derived_build_ms = ctx->derived_build_ms;
    return LFI_OK;
}

int lfi_std_band_info(lfi_ctx *ctx, lfi_std_band *out)
{
    if(!ctx || !out)
        return LFI_EINVAL;
    if(int rc = bind(ctx))
        return rc;
    const BandProbe *p = nullptr;
    if(int rc = run_band_probe(ctx, &p))
        return rc;
    std::memset(out, 0, sizeof(*out));
    out->probed = p->done;
    out->within_budget = p->ok;
    out->analytic_forced = (!p->ok || (ctx->flags & LFI_FLAG_STD_BAND_PROBE_FAIL)) && !(ctx->flags & LFI_FLAG_STD_ANALYTIC_BAND);
    out->sums = p->sums;
    out->worst_fraction = p->worst;
    out->probe_ms = p->ms;
    std::snprintf(out->message, sizeof(out->message),
                  p->ok ? (ctx->flags & LFI_FLAG_STD_BAND_PROBE_FAIL ? "measured bound holds (worst %.3f of the budget over %d sums); analytic band forced by LFI_FLAG_STD_BAND_PROBE_FAIL"
                                                                      : "measured bound holds on this device: worst %.3f of the budget over %d sums")
                        : "measured bound VIOLATED on this device (worst %.3f of the budget over %d sums): STD over more than 64 images takes the analytic band",
                  p->worst, p->sums);
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
    uint8_t *dev_weights = param_base(ctx) + ctx->blob_off_w16;
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
    if(int rc = join_filter(ctx))
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
    if(int rc = join_filter(ctx)) // a timed region ends when ALL its work has: the side stream's filter included
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
    if(wants_derived_copy(ctx, method, all_focus, a))
        (void)ensure_planar(ctx, true);
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
    if(int rc = join_filter(ctx))
        return rc;
    LFI_HIP(ctx, hipMemcpy2DAsync(rgba, pitch_bytes, ctx->maps + plane_bytes(ctx) * k, (size_t)ctx->width * 4,
                                  (size_t)ctx->width * 4, ctx->height, hipMemcpyDeviceToHost, ctx->stream));
    LFI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LFI_OK;
}

int lfi_download_quilt_tiles(lfi_ctx *ctx, int tiles_x, int tiles_y, int first_tile, int n, int v0, uint8_t *rgba, size_t pitch_bytes)
{
    if(!ctx)
        return LFI_EINVAL;
    if(!ctx->views || !ctx->have_params)
        return fail(ctx, LFI_EINVAL, "nothing rendered yet");
    if(tiles_x < 1 || tiles_y < 1 || first_tile < 0 || n < 1 || (long)first_tile + n > (long)tiles_x * tiles_y || v0 < 0 || (long)v0 + n > ctx->views_n)
        return fail(ctx, LFI_EINVAL, "quilt needs the tiles inside the quilt and as many views starting at v0 inside [0, views)");
    if(!rgba || pitch_bytes < (size_t)tiles_x * ctx->width * 4)
        return fail(ctx, LFI_EINVAL, "bad quilt pointer or pitch");
    if(int rc = bind(ctx))
        return rc;
    // The rows of tiles these tiles touch, assembled on the device by ONE kernel (planar views are expanded on the fly: no per-view
    // conversion pass), then copied to the host in one rectangle — or three when the first / last row of tiles is only partly this
    // context's (a trajectory sharded over several GPUs: every context fills its own tiles of the same host image).
    const int tr0 = first_tile / tiles_x, tr1 = (first_tile + n - 1) / tiles_x;
    const int W = ctx->width, rows = ctx->out_rows;
    const size_t qrow = (size_t)tiles_x * W * 4;
    const size_t need = qrow * rows * (size_t)(tr1 - tr0 + 1);
    if(ctx->quilt_bytes < need)
    {
        if(ctx->quilt)
            (void)hipFree(ctx->quilt);
        ctx->quilt = nullptr;
        ctx->quilt_bytes = 0;
        LFI_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->quilt), need));
        ctx->quilt_bytes = need;
    }
    const dim3 grid(((W + 3) / 4 + 255) / 256, rows, n), block(256);
    if(ctx->out_layout == LFI_LAYOUT_PLANAR_RGB)
        hipLaunchKernelGGL(lfi::quilt_assemble<true>, grid, block, 0, ctx->stream, ctx->views, reinterpret_cast<uint32_t *>(ctx->quilt), W, rows, view_pitch(ctx),
                           out_plane_bytes(ctx), v0, first_tile, tiles_x);
    else
        hipLaunchKernelGGL(lfi::quilt_assemble<false>, grid, block, 0, ctx->stream, ctx->views, reinterpret_cast<uint32_t *>(ctx->quilt), W, rows, 0,
                           out_plane_bytes(ctx), v0, first_tile, tiles_x);
    LFI_HIP(ctx, hipGetLastError());
    const int c_first = first_tile % tiles_x, c_last = (first_tile + n - 1) % tiles_x;
    // rows of tiles [ra, rb] × tile columns [ca, cb] → the host image (a row window: only the band's rows of every tile exist)
    auto copy_rect = [&](int ra, int rb, int ca, int cb) -> hipError_t {
        const size_t w_bytes = (size_t)(cb - ca + 1) * W * 4;
        if(rows == ctx->height) // the tiles' rows are contiguous in the host image too: one copy for all rows of tiles
            return hipMemcpy2DAsync(rgba + (size_t)ra * ctx->height * pitch_bytes + (size_t)ca * W * 4, pitch_bytes,
                                    ctx->quilt + (size_t)(ra - tr0) * rows * qrow + (size_t)ca * W * 4, qrow, w_bytes, (size_t)(rb - ra + 1) * rows,
                                    hipMemcpyDeviceToHost, ctx->stream);
        for(int r = ra; r <= rb; r++)
        {
            const hipError_t e = hipMemcpy2DAsync(rgba + ((size_t)r * ctx->height + ctx->out_y0) * pitch_bytes + (size_t)ca * W * 4, pitch_bytes,
                                                  ctx->quilt + (size_t)(r - tr0) * rows * qrow + (size_t)ca * W * 4, qrow, w_bytes, rows, hipMemcpyDeviceToHost, ctx->stream);
            if(e != hipSuccess)
                return e;
        }
        return hipSuccess;
    };
    if(tr0 == tr1)
        LFI_HIP(ctx, copy_rect(tr0, tr0, c_first, c_last));
    else
    {
        int full0 = tr0, full1 = tr1;
        if(c_first != 0)
        {
            LFI_HIP(ctx, copy_rect(tr0, tr0, c_first, tiles_x - 1));
            full0++;
        }
        if(c_last != tiles_x - 1)
        {
            LFI_HIP(ctx, copy_rect(tr1, tr1, 0, c_last));
            full1--;
        }
        if(full0 <= full1)
            LFI_HIP(ctx, copy_rect(full0, full1, 0, tiles_x - 1));
    }
    LFI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LFI_OK;
}

int lfi_download_quilt(lfi_ctx *ctx, int tiles_x, int tiles_y, int v0, uint8_t *rgba, size_t pitch_bytes)
{
    if(ctx && (tiles_x < 1 || tiles_y < 1))
        return fail(ctx, LFI_EINVAL, "quilt needs tiles_x*tiles_y views starting at v0 inside [0, views)");
    return lfi_download_quilt_tiles(ctx, tiles_x, tiles_y, 0, tiles_x * tiles_y, v0, rgba, pitch_bytes);
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
    if(int rc = join_filter(ctx))
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
        return "factored,lds,packed_p2,plain,factored_direct";
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
        static const char *const names[] = {"factored", "lds", "packed_p2", "plain", "factored_direct"};
        if(is_auto)
        {
            ctx->focus_variant = 0;
            return LFI_OK;
        }
        for(int i = 0; i < 5; i++)
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
    if(ctx->inputs_released)
        return fail(ctx, LFI_EINVAL, "the RGBA inputs were released (lfi_release_inputs): lfi_set_grid and upload the images again");
    if(!ctx->grid || !ctx->have_params)
        return fail(ctx, LFI_EINVAL, "lfi_set_grid / lfi_set_params have not been called");
    if(g < 0 || g >= ctx->n || !out_hw || map_index < 0 || map_index > 1)
        return fail(ctx, LFI_EINVAL, "bad image index, map index or pointer");
    if(ctx->windowed)
        return fail(ctx, LFI_EINVAL, "coordinate dumps are not supported with a row window");
    if(int rc = bind(ctx))
        return rc;
    if(int rc = join_filter(ctx))
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

int lfi_debug_pk_minmax3_f16(lfi_ctx *ctx, uint32_t *out_mismatches)
{
    if(!ctx || !out_mismatches)
        return LFI_EINVAL;
    if(int rc = bind(ctx))
        return rc;
    uint32_t *d = nullptr;
    LFI_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&d), sizeof(uint32_t)));
    hipError_t e = hipMemsetAsync(d, 0, sizeof(uint32_t), ctx->stream);
    if(e == hipSuccess)
    {
        hipLaunchKernelGGL(lfi::probe_pk_minmax3, dim3(1u << 16), dim3(256), 0, ctx->stream, d);
        e = hipGetLastError();
    }
    if(e == hipSuccess)
        e = hipMemcpyAsync(out_mismatches, d, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
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

#if LFI_SX_TRACE
// measurement builds only (blend_stdx.hpp): the per-workgroup unit clocks of the last blend_stdx launch; not part of include/lfi.h
int lfi_debug_sx_trace(unsigned long long *out, int n_words)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(lfi::lfi_sx_trace_buf), sizeof(unsigned long long) * std::min(n_words, 1024 * 32)) == hipSuccess ? 0 : 1;
}
#endif
#if FRT_TRACE
// measurement builds only (focus_factored.hpp): clocks per wave and category of the last focus_range_t launch; not part of include/lfi.h
int lfi_debug_frt_trace(unsigned long long *out, int n_words)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(lfi::lfi_frt_trace_buf), sizeof(unsigned long long) * std::min(n_words, 256 * 16 * 8)) == hipSuccess ? 0 : 1;
}
#endif
#if LFI_P3_TRACE
// measurement builds only (blend_p3.hpp): the per-workgroup unit clocks of the last blend_p3 launch; not part of include/lfi.h
int lfi_debug_p3_trace(unsigned long long *out, int n_words)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(lfi::lfi_p3_trace_buf), sizeof(unsigned long long) * std::min(n_words, 1024 * 8)) == hipSuccess ? 0 : 1;
}
#endif
} // extern "C"

