// blend_wave.hpp — TEN_WM / STD with a private LDS-DMA pipeline per WAVE (no workgroup barrier in the tile loop).
//
// blend_persist (blend_ten_persist.hpp) shares one 128-pixel tile between the four waves of a workgroup: every wave fetches a
// quarter of the images for all 128 pixels, so a barrier per unit couples the waves — and, through the matrix pipe that each
// SIMD's two resident waves share, the two workgroups of a CU: they drift into the same phase and their epilogues and DMA
// issue stop overlapping the other's MFMAs (profiles/r01_notes.md, STD ablations).  Here each wave owns 32 pixels of the tile
// and fetches ALL images for them into its own double buffer (64 images × 128 B = 8 KB per buffer), so nothing in the loop
// waits for another wave.  The price: the weight fragments must not change from unit to unit — they are loaded once per
// workgroup — so this kernel serves launches with one K-chunk (≤ 64 images) and one view pass (≤ 32·MT views): BASELINE
// configs 1, 2 and 4, each bench step.  Everything else takes blend_persist.
//
//      per wave, per tile:   s_waitcnt vmcnt(#stores of the previous epilogue)   → the tile's 8 DMA pieces have landed
//                            issue the 8 LDS-DMA pieces of the next tile into the other buffer
//                            compute the tile from LDS (same operand maps as blend_persist)
//                            epilogue + 32 stores
//
// LDS: 8 KB (TEN_WM, fp16 fragments) or 16 KB (STD, f32) of weights + 4 waves × 2 × 8 KB = 72 / 80 KB per workgroup, two
// workgroups per CU.
// k-loops and epilogue: blend_core.hpp (shared with blend_persist); LDS-DMA helpers: blend_ten_persist.hpp.
// Replaces Kernels::Tensors::process<false> / Kernels::Standard::process<false> (reference src/kernels.cu:289-343, 398-461).
#pragma once

#include "blend_ten_persist.hpp"

namespace lfi {

template <bool STD, int MT, bool NT_STORE>
__global__ void __launch_bounds__(256, 2) blend_wave(const KernelArgs a, const int tiles_x, const int n_tiles)
{
    constexpr int KC = 64, KS = KC / 16, VPP = MT * 32, TPX = 32;
    // weights: TEN_WM the fp16 fragments [k-octet][view] × 16 B (8 KB); STD f32 [image][view] (16 KB, no conversions in the k-loop)
    constexpr int W_DW = STD ? KC * VPP : (KC / 8) * VPP * 4;
    constexpr int PX_DW = KC * TPX;          // one pixel buffer of one wave
    __shared__ __attribute__((aligned(16))) uint32_t lds[W_DW + 4 * 2 * PX_DW];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int W = a.width, H = a.height;
    const uint32_t *grid32 = reinterpret_cast<const uint32_t *>(a.grid);
    const size_t plane_px = (size_t)W * (size_t)a.in_rows;   // one input plane as held by this context
    const size_t oplane_px = (size_t)W * (size_t)a.out_rows; // one output plane
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane(uint32_t(uintptr_t((lds_ptr_t)lds)));
    const uint32_t px_base = lds_base + W_DW * 4 + uint32_t(wave) * (2 * PX_DW * 4);
    const int kc = a.k_pad; // ≤ KC (host)

    // ---- once per workgroup: the weight fragments (as blend_persist's issue(), all four waves) ---------------------------
    if constexpr(STD)
    {
        // rows of the transposed f32 copy (w32t[image][v_pad]): 16 lanes × 16 B = one image's VPP = 64 views, 4 images per piece
        static_assert(!STD || MT == 2, "the f32 weight copy below assumes 64 views per pass");
        for(int k4 = wave; 4 * k4 < kc; k4 += 4)
            dma16(a.w32t + (size_t)(4 * k4 + (lane >> 4)) * a.v_pad + a.v0 + 4 * (lane & 15), lds_base + uint32_t(k4) * 1024u);
    }
    else
    {
        for(int o = wave; 8 * o < kc; o += 4)
            if(lane < VPP)
                dma16(a.w16s + (size_t)(a.v0 + lane) * a.k_pad + 8 * o, lds_base + uint32_t(o) * (VPP * 16));
    }

    // ---- per lane, once: the image this lane fetches in each of the 8 DMA pieces, its integer offsets, and the element offset
    // of its 16 bytes for the tile at (0, 0) — for tiles whose every sample is inside the image (wave-uniform test against
    // the bounds of the offsets) a piece's address is that constant plus one wave-uniform tile term.
    // piece i moves images 8i … 8i+7, eight lanes (16 B = 4 pixels each) per image: lane l ↔ image 8i + (l >> 3), pixels 4(l & 7)…
    int ox[8], oy[8], gi[8];
    int64_t lane_elem[8];
#pragma unroll
    for(int i = 0; i < 8; i++)
    {
        gi[i] = min(8 * i + (lane >> 3), a.n_images - 1); // padded images (zero weights) re-read the last one
        const lfi_int2 o = a.focused[gi[i]];
        ox[i] = o.x;
        oy[i] = o.y;
        lane_elem[i] = (int64_t)gi[i] * (int64_t)plane_px + (int64_t)(o.y - a.in_y0) * W + o.x + 4 * (lane & 7);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier(); // the only one: weights in LDS
    asm volatile("" ::: "memory");

    // tile (tx, ty) of the 128-pixel tile grid; this wave's 32 pixels start at column 128 tx + 32 wave of output row ty
    auto issue = [&](const int tx, const int ty, const int buf) {
        const int y = a.out_y0 + ty; // global row
        const int xw = tx * 128 + wave * TPX;
        const uint32_t dst = px_base + uint32_t(buf) * (PX_DW * 4);
        if(xw + a.fo_min_x >= 0 && xw + TPX + a.fo_max_x <= W && y + a.fo_min_y >= 0 && y + a.fo_max_y <= H - 1)
        {
            const uint32_t *tile = grid32 + ((size_t)y * W + xw); // wave-uniform
#pragma unroll
            for(int i = 0; i < 8; i++)
                if(8 * i < kc)
                    dma16(tile + lane_elem[i], dst + uint32_t(i) * 1024u);
            return;
        }
#pragma unroll
        for(int i = 0; i < 8; i++)
        {
            if(8 * i >= kc)
                break;
            const int sy = clampi(y + oy[i], 0, H - 1) - a.in_y0; // clamp in the full image, then index the held rows
            const int sx = xw + ox[i] + 4 * (lane & 7);
            const bool inside = sx >= 0 && sx + 4 <= W;
            if(__builtin_amdgcn_ballot_w64(inside) == ~0ull)
                dma16(grid32 + (size_t)gi[i] * plane_px + (size_t)sy * W + sx, dst + uint32_t(i) * 1024u);
            else
            {
                // a run crosses the left / right border: per-pixel clamp-to-edge addresses (reference src/kernels.cu:125),
                // two images × 32 pixels per instruction; the image's offsets come from the lane that holds them
#pragma unroll
                for(int s = 0; s < 4; s++)
                {
                    const int src_lane = 8 * (2 * s + h);
                    const int g = __builtin_amdgcn_ds_bpermute(4 * src_lane, gi[i]);
                    const int oxi = __builtin_amdgcn_ds_bpermute(4 * src_lane, ox[i]);
                    const int oyi = __builtin_amdgcn_ds_bpermute(4 * src_lane, oy[i]);
                    const int syy = clampi(y + oyi, 0, H - 1) - a.in_y0;
                    const int sxx = clampi(xw + oxi + r, 0, W - 1);
                    dma4(grid32 + (size_t)g * plane_px + (size_t)syy * W + sxx, dst + uint32_t(i) * 1024u + uint32_t(s) * 256u);
                }
            }
        }
    };

    // ---- this workgroup's tiles j, j+G, j+2G … as (tx, ty), advanced without divisions; a wave whose 32 pixels start beyond
    // the row's end skips the tile ------------------------------------------------------------------------------------------
    const int G = gridDim.x;
    const int step_y = G / tiles_x, step_x = G - step_y * tiles_x;
    const int rows_out = n_tiles / tiles_x;
    auto advance = [&](int &tx, int &ty) {
        do
        {
            tx += step_x;
            ty += step_y;
            if(tx >= tiles_x)
            {
                tx -= tiles_x;
                ty++;
            }
        } while(ty < rows_out && tx * 128 + wave * TPX >= W);
    };
    const int t0 = int(xcd_contiguous(blockIdx.x, gridDim.x));
    int ty = t0 / tiles_x, tx = t0 - ty * tiles_x;
    if(ty < rows_out && tx * 128 + wave * TPX >= W)
        advance(tx, ty);
    if(ty >= rows_out)
        return;
    int buf = 0;
    int prev_stores = 0; // store instructions of the previous epilogue (the youngest VMEM operations of this wave)
    issue(tx, ty, 0);

    f32x16 acc[MT][3]; // never cleared: the first MFMA of a tile takes a zero C operand (unit_ten / unit_std, ZERO_FIRST)

    const u32x4 *w_buf = reinterpret_cast<const u32x4 *>(lds);
    while(true)
    {
        int ntx = tx, nty = ty;
        advance(ntx, nty);
        // this wave's pieces of the current tile have landed; the previous epilogue's stores may still be in flight (vmcnt
        // retires in order and the stores are the youngest operations)
        if(prev_stores >= 32)
            asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else if(prev_stores >= 24)
            asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if(prev_stores >= 16)
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if(prev_stores >= 8)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if(nty < rows_out)
            issue(ntx, nty, buf ^ 1); // this wave finished reading that buffer one tile ago (program order)

        // ---- compute the tile ------------------------------------------------------------------------------------------------
        const uint32_t *px_buf = lds + W_DW + wave * (2 * PX_DW) + buf * PX_DW;
        if constexpr(!STD)
            unit_ten<MT, TPX, KS, true>(px_buf + r + 8 * h * TPX, w_buf, r, h, kc, acc);
        else
            unit_std<MT, TPX, KS, true, true>(px_buf + r + h * TPX, lds, r, h, kc, acc);

        // ---- epilogue ------------------------------------------------------------------------------------------------------------
        {
            // STD is bound by the matrix pipe the SIMD's two waves share: this wave's epilogue (VALU + stores) should run
            // under the other wave's MFMAs and be over quickly, not take the issue slots that wave leaves (measured −3 %)
            if constexpr(STD)
                __builtin_amdgcn_s_setprio(1);
            prev_stores = store_tile<STD, MT, NT_STORE, false>(a, acc, a.v0, ty, tx * 128 + wave * TPX, r, h, oplane_px);
            if constexpr(STD)
                __builtin_amdgcn_s_setprio(0);
        }
        if(nty >= rows_out)
            break;
        tx = ntx;
        ty = nty;
        buf ^= 1;
    }
}

} // namespace lfi
