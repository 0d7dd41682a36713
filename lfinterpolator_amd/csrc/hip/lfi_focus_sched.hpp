// lfi_focus_sched.hpp — scheduling of the focus-map estimate: the factored pipeline's workspace and its two-stream pass graph
// (focus_factored.hpp), and the choice between it, the row-window path and the other estimate variants.
// Replaces the FocusMap::estimate / FocusMap::filter launches (reference src/interpolator.cu:261-266).
// Included by lfi_hip.hip only (one translation unit), after lfi_context.hpp.
#pragma once

#include "lfi_context.hpp"
#include "focus_factored.hpp"

namespace {

// the factored estimate (focus_factored.hpp): carve the workspace, then plan → pad → E → exact keys → pick.
// Returns LFI_OK with *done = false when the padded planes would be unreasonably large (the caller takes another variant).
// direct_range: the range pass by focus_range (rounds 1-4's kernel: every use loads and widens its own samples) even where focus_range_t
// applies (variant "factored_direct": the second implementation in the parity tests, and the A/B partner)
int launch_focus_factored(lfi_ctx *ctx, const KernelArgs &a, bool *done, bool direct_range)
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
    int Sx = (int)std::ceil(fmax * ox) + 1, Sy = (int)std::ceil(fmax * oy) + 1; // ≥ |floor(δ)| and ≥ |floor(δ)+1|
    // The padded planes of an earlier call serve this one if the inputs have not changed since and their padding covers these shifts
    // (any larger padding gives the same samples): then the geometry is theirs.  New planes are padded to the next multiple of 8, so
    // that the neighbouring steps of a focus sweep find them large enough.
    // (also when only SOME images were replaced since — lfi_upload_image —: then only the planes of the sampled images among them are redone)
    const bool pad_kept = ctx->grid_tracked && ctx->focus_ws && ctx->pad_version != 0 && ctx->grid_full_version <= ctx->pad_version &&
                          ctx->pad_ids == ctx->h_focus_ids && ctx->pad_radius[0] == rx && ctx->pad_radius[1] == ry && ctx->pad_shift[0] >= Sx &&
                          ctx->pad_shift[1] >= Sy;
    // Planes that have to GROW (an ascending sweep) grow by a quarter more than asked for: every change of the geometry rebuilds the planes
    // and may reallocate a workspace of gigabytes (≈ 80 ms per step when it happened on every step of a sweep).
    auto padded = [](const int need, const int had) { return ((had > 0 && need > had ? need + need / 4 : need) + 7) / 8 * 8; };
    Sx = pad_kept ? ctx->pad_shift[0] : padded(Sx, ctx->pad_shift[0]);
    Sy = pad_kept ? ctx->pad_shift[1] : padded(Sy, ctx->pad_shift[1]);
    w.Px = Sx + rx;
    w.Py = Sy + ry;
    w.Wp = (w.Px + std::max(W + Sx + rx, w.We_p - rx + Sx) + 3) / 4 * 4;
    w.Hp = w.Py + std::max(H + Sy + ry, w.He_p - ry + Sy);
    // (+ FRT_PR + 16 rows of slack behind the last plane: focus_range_t fetches whole patches, also for the rows of its last tiles that
    // lie below the extended image)
    const size_t pad_bytes = sizeof(uint32_t) * ((size_t)ctx->n_focus_ids * w.Hp + lfi::FRT_PR + 16) * w.Wp;
    if(pad_bytes > ((size_t)16 << 30))
        return LFI_OK;
    // Can the range pass unpack its samples once into LDS (focus_range_t)?  Within every group of CPW consecutive candidates a view's integer
    // shifts — floor(f_i · offset), the device's own arithmetic (focus_plan_shifts) — must span at most FRT_MAX_DX pixels and FRT_MAX_DY rows:
    // groups of 8 candidates if that holds, else groups of 4, else focus_range.  Typed loads address the planes with 32 bits.
    int range_cpw = 0;
    if(!direct_range && pad_bytes < ((size_t)1 << 32))
        for(int cpw : {8, 4})
        {
            bool fits = true;
            for(size_t k = 0; k < ctx->h_focus_offsets.size() && fits; k++)
                for(int i0 = 0; i0 < lfi::FOCUS_STEPS && fits; i0 += cpw)
                {
                    int lo[2] = {INT32_MAX, INT32_MAX}, hi[2] = {INT32_MIN, INT32_MIN};
                    for(int i = i0; i < i0 + cpw; i++)
                    {
                        const float f = std::fmaf(step, static_cast<float>(i), ctx->focus);
                        const int sx = static_cast<int>(std::floor(static_cast<double>(f) * static_cast<double>(ctx->h_focus_offsets[k].x)));
                        const int sy = static_cast<int>(std::floor(static_cast<double>(f) * static_cast<double>(ctx->h_focus_offsets[k].y)));
                        lo[0] = std::min(lo[0], sx), hi[0] = std::max(hi[0], sx), lo[1] = std::min(lo[1], sy), hi[1] = std::max(hi[1], sy);
                    }
                    fits = hi[0] - lo[0] <= lfi::FRT_MAX_DX && hi[1] - lo[1] <= lfi::FRT_MAX_DY;
                }
            if(fits)
            {
                range_cpw = cpw;
                break;
            }
        }
    size_t at = 0;
    auto carve = [&](size_t bytes) {
        const size_t here = at;
        at += (bytes + 255) / 256 * 256;
        return here;
    };
    const size_t o_shifts = carve(sizeof(int32_t) * 4 * lfi::FOCUS_STEPS * lfi::FOCUS_MAX_IDS);
    const size_t o_badx = carve(sizeof(uint32_t) * W), o_bady = carve(sizeof(uint32_t) * H);
    const size_t o_tapx = carve(sizeof(uint32_t) * 3 * W), o_tapy = carve(sizeof(uint32_t) * 3 * H); // cleared with badx / bady: adjacent
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
    const size_t o_patches = carve(sizeof(lfi::FocusPatch) * (lfi::FOCUS_STEPS / 4) * lfi::FOCUS_MAX_IDS);
    const size_t o_pad = carve(pad_bytes);
    if(ctx->focus_ws_bytes < at) // a larger workspace serves smaller geometries too
    {
        if(ctx->focus_ws)
            (void)hipFree(ctx->focus_ws);
        ctx->focus_ws = nullptr;
        ctx->focus_ws_bytes = 0;
        ctx->pad_version = 0;
        LFI_HIP(ctx, hipMalloc(&ctx->focus_ws, at));
        ctx->focus_ws_bytes = at;
    }
    uint8_t *base = static_cast<uint8_t *>(ctx->focus_ws);
    w.shifts = reinterpret_cast<int32_t *>(base + o_shifts);
    w.badx = reinterpret_cast<uint32_t *>(base + o_badx);
    w.bady = reinterpret_cast<uint32_t *>(base + o_bady);
    w.tapx = reinterpret_cast<uint32_t *>(base + o_tapx);
    w.tapy = reinterpret_cast<uint32_t *>(base + o_tapy);
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
    const bool e_32bit = sizeof(uint16_t) * (size_t)w.He_p * w.We_p < ((size_t)1 << 31); // focus_pick_sep: a candidate's plane of E behind one buffer descriptor
    w.K = reinterpret_cast<uint16_t *>(base + o_K);
    w.deltas = reinterpret_cast<int64_t *>(base + o_deltas);
    w.pad = reinterpret_cast<uint32_t *>(base + o_pad);
    lfi::FocusPatch *patches = reinterpret_cast<lfi::FocusPatch *>(base + o_patches);
    // Two streams: the plan and the flagged-pair passes are small, latency-bound kernels; they run beside the padded copy
    // and the range pass (bandwidth / VALU bound) instead of in front of them.
    //   main:  plan_shifts ─┬─ pad ─┬─ range ───────────────────────────────────────────┬─ line_keys → pick (→ filter, by the caller)
    //   aux:                └─ flags → lists → prefix ─┴─ {lines_rows, lines_cols, exact} ─┘
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
    // (the patch plans of focus_range_t by the same launch: one kernel less in front of the range pass)
    hipLaunchKernelGGL(lfi::focus_plan_shifts, dim3(1), dim3(1024), 0, st, a, w, patches, range_cpw);
    LFI_HIP(ctx, hipEventRecord(ctx->ev_fork, st));
    bool padded_any = true; // something was enqueued between ev_fork and the range pass
    if(pad_kept && ctx->pad_version != 0) // (a reallocated workspace cleared pad_version)
    {
        padded_any = false;
        for(int k = 0; k < ctx->n_focus_ids; k++)
            if(image_changed_since(ctx, ctx->h_focus_ids[k], ctx->pad_version))
            {
                hipLaunchKernelGGL(lfi::focus_pad, dim3((w.Wp + 255) / 256, (w.Hp + lfi::FOCUS_PAD_ROWS - 1) / lfi::FOCUS_PAD_ROWS, 1), dim3(64), 0, st, a, w, k);
                padded_any = true;
            }
        ctx->pad_version = ctx->grid_version;
    }
    else
    {
        hipLaunchKernelGGL(lfi::focus_pad, dim3((w.Wp + 255) / 256, (w.Hp + lfi::FOCUS_PAD_ROWS - 1) / lfi::FOCUS_PAD_ROWS, ctx->n_focus_ids), dim3(64), 0, st, a, w, 0);
        ctx->pad_version = ctx->grid_tracked ? ctx->grid_version : 0;
        ctx->pad_shift[0] = Sx;
        ctx->pad_shift[1] = Sy;
        ctx->pad_radius[0] = rx;
        ctx->pad_radius[1] = ry;
        ctx->pad_ids = ctx->h_focus_ids;
    }
    // (planes kept and nothing re-padded: ev_fork says all the side stream needs to know — one event packet less in front of the range pass)
    if(padded_any)
        LFI_HIP(ctx, hipEventRecord(ctx->ev_pad, st));
    const uint32_t tiles_x = uint32_t(w.We_p / 256), tiles_y = uint32_t(w.He_p / 4);
    if(range_cpw)
    {
        const uint32_t groups = uint32_t(lfi::FOCUS_STEPS / range_cpw);
        const uint32_t ttx = uint32_t(w.We_p / lfi::FRT_TW), tty = uint32_t((w.He_p + lfi::FRT_TH - 1) / lfi::FRT_TH);
        const int striped = ttx >= 8; // (row-major order of the items instead: range pass 1.64 → 1.71 ms at 4K)
        const uint32_t nblocks = striped ? 8u * lfi::stripe_blocks_per_xcd(ttx, tty, groups) : ttx * tty * groups;
        // persistent: one workgroup per CU (it owns the whole LDS), a multiple of 8 so that a workgroup's work items stay on its XCD
        const uint32_t grid = std::min(nblocks, uint32_t(std::max(ctx->cu_count / 8 * 8, 8)));
        if(range_cpw == 8)
            hipLaunchKernelGGL(lfi::focus_range_t<8>, dim3(grid), dim3(64 * (lfi::FRT_NW + lfi::FRT_LW)), 0, st, a, w, patches, uint32_t(pad_bytes), nblocks, striped);
        else
            hipLaunchKernelGGL(lfi::focus_range_t<4>, dim3(grid), dim3(64 * (lfi::FRT_NW + lfi::FRT_LW)), 0, st, a, w, patches, uint32_t(pad_bytes), nblocks, striped);
    }
    else
    {
        constexpr int CPW = 4, GROUPS = lfi::FOCUS_STEPS / CPW;
        const int striped = tiles_x >= 8;
        const uint32_t nblocks = striped ? 8u * lfi::stripe_blocks_per_xcd(tiles_x, tiles_y, GROUPS) : tiles_x * tiles_y * GROUPS;
        hipLaunchKernelGGL(lfi::focus_range<CPW>, dim3(nblocks), dim3(256), 0, st, a, w, nblocks, striped);
    }
    LFI_HIP(ctx, hipStreamWaitEvent(aux, ctx->ev_fork, 0));
    LFI_HIP(ctx, hipMemsetAsync(w.badx, 0, o_cols - o_badx, aux)); // badx, bady, tapx, tapy are adjacent
    hipLaunchKernelGGL(lfi::focus_plan_flags, dim3((std::max(W, H) + 255) / 256, lfi::FOCUS_STEPS, 2), dim3(256), 0, aux, a, w);
    hipLaunchKernelGGL(lfi::focus_plan_lists, dim3(lfi::FOCUS_STEPS, 2), dim3(64), 0, aux, a, w);
    hipLaunchKernelGGL(lfi::focus_plan_prefix, dim3(1), dim3(1), 0, aux, a, w);
    if(padded_any)
        LFI_HIP(ctx, hipStreamWaitEvent(aux, ctx->ev_pad, 0));
    {
        const uint32_t per_pass = uint32_t(ctx->cu_count) * 4u / 8u * 8u;
#ifdef LFI_MEASUREMENT_BUILD // one launch per pass, so that a kernel trace shows what each of the three costs
        for(uint32_t pass = 0; pass < 3; pass++)
            hipLaunchKernelGGL(lfi::focus_flagged, dim3(per_pass), dim3(256), 0, aux, a, w, per_pass, pass);
#else
        hipLaunchKernelGGL(lfi::focus_flagged, dim3(3 * per_pass), dim3(256), 0, aux, a, w, per_pass, 0u);
#endif
    }
    // The keys of single-axis pairs take their unflagged taps from E: focus_line_keys runs behind BOTH streams' work.  On the main stream —
    // the flagged passes end before the range pass does, so the join is an event already signalled, and the keys, the pick and whatever the
    // caller enqueues next follow the range pass in one queue (on the side stream they cost two more hops between queues, ≈ 10 µs each).
    LFI_HIP(ctx, hipEventRecord(ctx->ev_join, aux));
    LFI_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_join, 0));
    hipLaunchKernelGGL(lfi::focus_line_keys, dim3(ctx->cu_count * 8), dim3(256), 0, st, a, w);
    {
        // two pixels per lane need dword-aligned sample pairs: even radius_x (the reference's is)
        const int ppl = (rx % 2 == 0 && W >= 2) ? 2 : 1;
        const uint32_t blocks_x = uint32_t((W + 64 * ppl - 1) / (64 * ppl)), blocks_y = uint32_t((H + 3) / 4);
        const int striped = blocks_x >= 8;
        const uint32_t nblocks = striped ? 8u * lfi::stripe_blocks_per_xcd(blocks_x, blocks_y, 1u) : blocks_x * blocks_y;
        // the tap block taken apart (focus_pick_sep): even radius_x of at most 64; variant "factored_direct" keeps focus_pick<2>, the other implementation
        if(ppl == 2 && rx <= 64 && e_32bit && !direct_range)
        {
            // (row-major order of the workgroups, not stripes per XCD: this kernel's re-use of E's rows happens inside a workgroup)
            const uint32_t nb = blocks_x * lfi::focus_pick_sep_block_rows(H, ry);
            hipLaunchKernelGGL(lfi::focus_pick_sep, dim3(nb), dim3(64 * lfi::FPS_WAVES), 0, st, a, w);
        }
        else if(ppl == 2)
            hipLaunchKernelGGL(lfi::focus_pick<2>, dim3(nblocks), dim3(256), 0, st, a, w, striped);
        else
            hipLaunchKernelGGL(lfi::focus_pick<1>, dim3(nblocks), dim3(256), 0, st, a, w, striped);
    }
    LFI_HIP(ctx, hipGetLastError());
    *done = true;
    return LFI_OK;
}

} // namespace
