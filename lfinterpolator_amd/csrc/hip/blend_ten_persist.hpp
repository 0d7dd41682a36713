// blend_ten_persist.hpp — TEN_WM (and STD) as a persistent, double-buffered LDS-DMA pipeline (the production kernels).
//
// Measurements that shaped it (profiles/r01_notes.md): the memory system sustains this access pattern (64 shifted input
// planes gathered, 64 output planes scattered) at ≈5.4–5.8 TB/s with trivial compute; a kernel that does
// load-tile → barrier → compute → store once per workgroup exposes the whole HBM latency per tile (waves parked 84 % of
// their life).  So each workgroup here is persistent and keeps a pipeline of "units" (one unit = one K-chunk of one pixel
// tile for one view pass):
//
//      top of unit u:   s_waitcnt vmcnt(#stores of the previous epilogue)   → this wave's DMA pieces of unit u have landed,
//                                                                             the previous tile's stores stay in flight
//                       s_barrier                                           → everybody's pieces have landed AND everybody has
//                                                                             finished reading the other buffer
//                       issue LDS-DMA of unit u+1 into the other buffer     (pixels 32 KB + weight fragments 8 KB)
//                       compute unit u from LDS (ds_read → v_perm → MFMA)
//                       last chunk of the tile: packed-fp16 epilogue + 32 stores per wave
//
// One barrier per unit, reads of unit u+1 and writes of tile u-1 overlap the MFMA/VALU work of unit u inside the same
// workgroup; two workgroups per CU (2 × 80 KB LDS) fill each other's gaps.
//
// The DMA instructions are issued from inline asm on purpose: hipcc would otherwise (correctly, but fatally for the
// pipeline) wait for vmcnt(0) before the first ds_read after a global_load_lds it knows about.  Because they are invisible
// to the compiler, the loop contains NO compiler-tracked global loads (offsets come through scalar loads = lgkmcnt, weights
// through LDS), so the only s_waitcnt vmcnt in the loop is the hand-counted one above.  vmcnt retires in order and counts
// stores, which is why the count of epilogue stores (a compile-time constant for full tiles, else we wait for everything)
// appears in it.
//
// Contraction, operand maps, subnormal-pixel trick and packed epilogue: blend_ten.hpp / epilogue_packed.hpp.
// Replaces Kernels::Tensors::process<allFocus> (reference src/kernels.cu:398-461).
#pragma once

#include "blend_core.hpp"

namespace lfi {

// LDS-DMA from inline asm: lane l's 16 (4) bytes at gptr land at LDS byte address lds_addr + 16 l (4 l).
// M0 carries the LDS base; it is written in the same statement that uses it (hipcc does not preserve M0 across statements).
__device__ __forceinline__ void dma16(const void *gptr, uint32_t lds_addr)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
}
__device__ __forceinline__ void dma4(const void *gptr, uint32_t lds_addr)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gptr), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
}

// the same with a wave-uniform 64-bit base (SGPR pair) and a 32-bit per-lane byte offset: no 64-bit vector address arithmetic
__device__ __forceinline__ void dma4_s(const void *sbase, uint32_t voff, uint32_t lds_addr)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
}

__device__ __forceinline__ void dma16_s(const void *sbase, uint32_t voff, uint32_t lds_addr)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
}

template <int MT, int KC_>
struct PersistCfg
{
    static constexpr int NW = 4;                 // waves per workgroup, side by side along the row
    static constexpr int TPX = 128;              // pixels per tile: 32 per wave, one per lane column
    static constexpr int KC = KC_;               // images per chunk (64, or 32 to fit more workgroups per CU)
    static constexpr int VPP = MT * 32;          // views per pass
    static constexpr int PX_DW = KC * TPX;       // dwords of pixels per pixel buffer (32 KB)
    static constexpr int W_DW = (KC / 8) * VPP * 4; // dwords of weight fragments per weight buffer: [k-octet][view] × 16 B
    static constexpr int LDS_DW = 2 * PX_DW + 2 * W_DW; // two pixel buffers, then two weight buffers
};

// STD = false: TEN_WM (fp16 MFMA, packed truncating epilogue, weights ×2^15);  STD = true: the exact-fp32 path of
// Kernels::Standard::process (reference src/kernels.cu:289-343) on v_mfma_f32_32x32x2_f32 — see blend_std.hpp — in the same pipeline.
// PLANAR_VIEWS: the views are the planar layout's byte planes (store_tile_planar, blend_core.hpp) instead of RGBA planes
template <bool STD, int MT, bool ALLFOCUS, bool NT_STORE, int KC_ = 64, int WGS = 2, bool PLANAR_VIEWS = false>
__global__ void __launch_bounds__(256, WGS)
    blend_persist(const KernelArgs a, const int tiles_x, const int n_tiles, const int view_passes)
{
    using C = PersistCfg<MT, KC_>;
    constexpr int TPX = C::TPX, KC = C::KC, VPP = C::VPP, KS = KC / 16;
    __shared__ __attribute__((aligned(16))) uint32_t lds[C::LDS_DW];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int W = a.width, H = a.height;
    const uint32_t *grid32 = reinterpret_cast<const uint32_t *>(a.grid);
    const size_t plane_px = (size_t)W * (size_t)a.in_rows;      // one input plane as held by this context
    const size_t oplane_px = (size_t)W * (size_t)a.out_rows;    // one output plane
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane(uint32_t(uintptr_t((lds_ptr_t)lds)));
    // the offset tables are read through the scalar cache (constant address space): wave-uniform, and counted by lgkmcnt, so
    // they never touch the vmcnt bookkeeping of the pipeline
    typedef const __attribute__((address_space(4))) int32_t *const_int_ptr;
    typedef const __attribute__((address_space(4))) float *const_float_ptr;
    const const_int_ptr c_focused = (const_int_ptr)(uintptr_t)a.focused;   // [g] = {x, y}
    const const_float_ptr c_offsets = (const_float_ptr)(uintptr_t)a.offsets;

    // ---- issue the LDS-DMA of one unit (tile t, view pass `pass`, chunk k0) into buffer b ------------------------------------
    // integer offsets of this lane's image in each of the wave's DMA slots (slot j ↔ image pair wave + NW*j of a chunk):
    // all scalar loads of a chunk are issued back to back (one wait), and when the whole stack fits one chunk they are
    // loaded once for the kernel — the offsets do not depend on the tile
    constexpr int SLOTS = KC / 2 / C::NW;
    struct SlotOffsets
    {
        int ox[SLOTS], oy[SLOTS];
    };
    auto load_offsets = [&](int k0) {
        SlotOffsets so;
        const int kn = min(min(KC, a.k_pad - k0), a.n_images - k0);
        int o0x[SLOTS], o0y[SLOTS], o1x[SLOTS], o1y[SLOTS];
#pragma unroll
        for(int j = 0; j < SLOTS; j++)
        {
            const int q = wave + C::NW * j;
            const int g0 = min(k0 + 2 * q, k0 + max(kn, 1) - 1), g1 = min(g0 + 1, k0 + max(kn, 1) - 1);
            o0x[j] = c_focused[2 * g0];
            o0y[j] = c_focused[2 * g0 + 1];
            o1x[j] = c_focused[2 * g1];
            o1y[j] = c_focused[2 * g1 + 1];
        }
#pragma unroll
        for(int j = 0; j < SLOTS; j++)
        {
            so.ox[j] = h ? o1x[j] : o0x[j];
            so.oy[j] = h ? o1y[j] : o0y[j];
        }
        return so;
    };

    // all-focus: the focus values of this lane's two pixels of a tile (pixels x0 + lane and x0 + 64 + lane), raw map dwords first.  They
    // are fetched ONE UNIT EARLIER than the gathers that need them, and only when the tile changes (the chunks of a tile share them):
    // round 2 loaded them inside `issue`, right after the barrier — two dependent L2 latencies (the compiler waits with vmcnt(0) after
    // each) in front of every unit's gathers.
    const uint8_t *map_plane = ALLFOCUS ? a.maps + (size_t)a.map_index * (size_t)W * H * 4 : nullptr; // maps are whole-image planes
    auto map_raw = [&](const int t, uint32_t &m0, uint32_t &m1) {
        const int ty = t / tiles_x;
        const int y = clampi(a.out_y0 + ty, 0, H - 1);
        const int x0 = (t - ty * tiles_x) * TPX;
        const uint32_t *row = reinterpret_cast<const uint32_t *>(map_plane) + (size_t)y * W;
        m0 = row[clampi(x0 + lane, 0, W - 1)];
        m1 = row[clampi(x0 + 64 + lane, 0, W - 1)];
    };
    // loadFocusFromMap (src/kernels.cu:134-137): decode_focus's arithmetic on a map dword that is already here
    auto map_focus = [&](const uint32_t m) { return __builtin_fmaf(__fdiv_rn(static_cast<float>(m & 0xffu), 255.0f), a.range, a.focus); };
    auto issue = [&](int t, int pass, int k0, int pb, int wb, bool with_pixels, const SlotOffsets &so, const float f0, const float f1) {
        const int ty = t / tiles_x;
        const int y = a.out_y0 + ty; // global row
        const int x0 = (t - ty * tiles_x) * TPX;
        const int kc = min(KC, a.k_pad - k0);
        const int kn = min(kc, a.n_images - k0);
        const uint32_t px_addr = lds_base + uint32_t(pb) * (C::PX_DW * 4);
        const uint32_t w_addr = lds_base + 2 * C::PX_DW * 4 + uint32_t(wb) * (C::W_DW * 4);
        // weight fragments: one instruction per k-octet o: lane l ← W[v0 + pass*VPP + l][k0 + 8o … +7] (×2^15), 16 bytes,
        // landing at slot o*VPP + l: the A-fragment reads of a k-step are then two contiguous 512-B runs (conflict-free)
        for(int o = wave; 8 * o < kc; o += C::NW)
        {
            if(lane < VPP)
            {
                const uint16_t *src = (STD ? a.w16 : a.w16s) + (size_t)(a.v0 + pass * VPP + lane) * a.k_pad + k0 + 8 * o;
                dma16(src, w_addr + uint32_t(o) * (VPP * 16));
            }
        }
        if(!with_pixels)
            return; // a later view pass of a single-chunk tile: the pixels are already in LDS
        if constexpr(!ALLFOCUS)
        {
            // pixels: 16 B/lane pieces, two images per instruction (half-wave hh ↔ image 2q+hh, lane column c ↔ pixels 4c…4c+3);
            // slot j of this wave is image pair q = wave + NW*j
#pragma unroll
            for(int j = 0; j < SLOTS; j++)
            {
                const int q = wave + C::NW * j;
                if(2 * q < kn)
                {
                    const int g0 = k0 + 2 * q;
                    const int g1 = min(g0 + 1, k0 + kn - 1);       // odd tail: the second half reloads the last image
                    const int ox = so.ox[j], oy = so.oy[j];         // this half-wave's image: g0 (h = 0) or g1 (h = 1)
                    const int g = h ? g1 : g0;
                    const int sy = clampi(y + oy, 0, H - 1) - a.in_y0; // clamp in the full image, then index the held rows
                    const int sx = x0 + ox + 4 * r;
                    const bool inside = (sx >= 0) && (sx + 4 <= W) && (2 * q + 1 < kn || h == 0);
                    const bool all_inside = __builtin_amdgcn_ballot_w64(inside) == ~0ull;
                    if(all_inside)
                        dma16(grid32 + (size_t)g * plane_px + (size_t)sy * W + sx, px_addr + uint32_t(q) * 1024u);
                    else
                    {
                        // a run crosses the left/right border (or the tile is ragged): per-pixel clamp-to-edge addresses
                        // (reference src/kernels.cu:125), 64 pixels per instruction; the other half-wave's offsets come by DPP-free
                        // broadcast through readlane (wave-uniform per image)
#pragma unroll
                        for(int img = 0; img < 2; img++)
                        {
                            if(2 * q + img >= kn)
                                break;
                            const int oxi = __builtin_amdgcn_readlane(ox, img * 32), oyi = __builtin_amdgcn_readlane(oy, img * 32);
                            const int syy = clampi(y + oyi, 0, H - 1) - a.in_y0;
                            const uint32_t *row = grid32 + (size_t)(g0 + img) * plane_px + (size_t)syy * W;
#pragma unroll
                            for(int s = 0; s < 2; s++)
                            {
                                const int sxx = clampi(x0 + oxi + 64 * s + lane, 0, W - 1);
                                dma4(row + sxx, px_addr + uint32_t(2 * q + img) * 512u + uint32_t(s) * 256u);
                            }
                        }
                    }
                }
            }
        }
        else
        {
            // all-focus: the warp depends on the pixel's own focus value (src/kernels.cu:78-82): per-pixel gather, 64 pixels per DMA
            // instruction.  Per pixel and image: fma → truncate → clamp (v_med3) for x and y, one 24-bit multiply-add for the pixel
            // index; the plane's base (minus the rows above the held window) stays in an SGPR pair, the images' float offsets are
            // read one iteration ahead through the scalar cache.  (The first version spent ≈400 issue cycles per image on 64-bit
            // vector multiplies and an exposed scalar load: all-focus renders ran 3× slower than fixed-focus ones.)
            const float xf0 = static_cast<float>(x0 + lane), xf1 = static_cast<float>(x0 + 64 + lane), yf = static_cast<float>(y);
            int gi = wave;
            float ox = 0.0f, oy = 0.0f;
            if(gi < kn)
            {
                ox = c_offsets[2 * (k0 + gi)];
                oy = c_offsets[2 * (k0 + gi) + 1];
            }
            for(; gi < kn; gi += C::NW)
            {
                const int gn = min(gi + C::NW, kn - 1);
                const float ox_n = c_offsets[2 * (k0 + gn)], oy_n = c_offsets[2 * (k0 + gn) + 1]; // for the next iteration
                // warp_float(coord, focus, offset) = (int)fmaf(focus, offset, float(coord)), then clamp-to-edge (src/kernels.cu:125)
                const int sx0 = clampi(static_cast<int>(__builtin_fmaf(f0, ox, xf0)), 0, W - 1); // min(max()) → v_med3_i32
                const int sy0 = clampi(static_cast<int>(__builtin_fmaf(f0, oy, yf)), 0, H - 1);
                const int sx1 = clampi(static_cast<int>(__builtin_fmaf(f1, ox, xf1)), 0, W - 1);
                const int sy1 = clampi(static_cast<int>(__builtin_fmaf(f1, oy, yf)), 0, H - 1);
                // rows are indexed inside the held window: fold −in_y0·W into the scalar base (W, H < 2^24: lfi_set_grid)
                const uint32_t *base = grid32 + (size_t)(k0 + gi) * plane_px - (size_t)a.in_y0 * W;
                dma4_s(base, (__umul24(uint32_t(sy0), uint32_t(W)) + uint32_t(sx0)) << 2, px_addr + uint32_t(gi) * 512u);
                dma4_s(base, (__umul24(uint32_t(sy1), uint32_t(W)) + uint32_t(sx1)) << 2, px_addr + uint32_t(gi) * 512u + 256u);
                ox = ox_n;
                oy = oy_n;
            }
        }
    };

    // ---- unit sequence of this workgroup: tiles j, j+G, j+2G … ; for each: passes × chunks --------------------------------------
    // Pixel buffers alternate per tile when the whole image stack fits one chunk (then every view pass of the tile reuses
    // them) and per unit otherwise; weight buffers alternate per unit.
    const int G = gridDim.x;
    const bool single_chunk = a.k_pad <= KC;
    int t = int(tile_of_block(blockIdx.x, gridDim.x, a.flags));
    if(t >= n_tiles)
        return;
    int pass = 0, k0 = 0, pbuf = 0, wbuf = 0;
    int prev_stores = 0; // store instructions this wave issued in the previous epilogue (they are the youngest VMEM ops)
    SlotOffsets slot_offsets = load_offsets(0);
    float f0 = 0.0f, f1 = 0.0f; // the focus values of the tile whose units are being issued
    if constexpr(ALLFOCUS)
    {
        uint32_t m0, m1;
        map_raw(t, m0, m1);
        f0 = map_focus(m0);
        f1 = map_focus(m1);
    }
    issue(t, 0, 0, 0, 0, true, slot_offsets, f0, f1);

    f32x16 acc[MT][3];
#pragma unroll
    for(int m = 0; m < MT; m++)
#pragma unroll
        for(int c = 0; c < 3; c++)
#pragma unroll
            for(int e = 0; e < 16; e++)
                acc[m][c][e] = 0.0f;

    while(true)
    {
        // next unit
        int nt = t, npass = pass, nk0 = k0 + KC;
        if(nk0 >= a.k_pad)
        {
            nk0 = 0;
            npass = pass + 1;
            if(npass >= view_passes)
            {
                npass = 0;
                nt = t + G;
            }
        }
        const bool have_next = nt < n_tiles;
        const bool next_needs_pixels = !single_chunk || nt != t;
        const int npbuf = next_needs_pixels ? (pbuf ^ 1) : pbuf;

        // the next unit starts a new tile: its map values now, so that they travel beside the pieces in flight.  (Younger than the stores
        // counted below: the wait only becomes stricter.)
        uint32_t m0n = 0u, m1n = 0u;
        const bool new_tile = ALLFOCUS && have_next && nt != t;
        if(new_tile)
            map_raw(nt, m0n, m1n);

        // (A) this wave's pieces of the current unit have landed; the previous epilogue's stores may still be in flight.
        // vmcnt retires in order and the stores are the youngest operations, so waiting for "at most prev_stores outstanding"
        // covers every DMA piece; the immediate is the largest threshold not above the exact count.
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
        __builtin_amdgcn_s_barrier(); // everybody's pieces have landed; everybody is done with the buffers about to be refilled
        asm volatile("" ::: "memory");
        if(have_next)
        {
            if(!single_chunk && !ALLFOCUS)
                slot_offsets = load_offsets(nk0);
            if(new_tile)
            {
                f0 = map_focus(m0n);
                f1 = map_focus(m1n);
            }
            issue(nt, npass, nk0, npbuf, wbuf ^ 1, next_needs_pixels, slot_offsets, f0, f1);
        }

        // ---- compute the current unit ------------------------------------------------------------------------------------------------
        const int kc = min(KC, a.k_pad - k0);
        const uint32_t *px_buf = lds + pbuf * C::PX_DW;
        const u32x4 *w_buf = reinterpret_cast<const u32x4 *>(lds + 2 * C::PX_DW + wbuf * C::W_DW);
        if constexpr(!STD)
            unit_ten<MT, TPX, KS, false>(px_buf + wave * 32 + r + 8 * h * TPX, w_buf, r, h, kc, acc);
        else
            unit_std<MT, TPX, KS, false>(px_buf + wave * 32 + r + h * TPX, w_buf, r, h, kc, acc);

        // ---- last chunk of the tile: epilogue ----------------------------------------------------------------------------------------
        prev_stores = 0;
        if(k0 + KC >= a.k_pad)
        {
            const int y = t / tiles_x; // row inside the output window
            const int xw = (t - y * tiles_x) * TPX + wave * 32;
            if constexpr(PLANAR_VIEWS)
                prev_stores = store_tile_planar<STD, MT, NT_STORE, true>(a, acc, a.v0 + pass * VPP, y, xw, r, h);
            else
                prev_stores = store_tile<STD, MT, NT_STORE, true>(a, acc, a.v0 + pass * VPP, y, xw, r, h, oplane_px);
        }

        if(!have_next)
            break;
        t = nt;
        pass = npass;
        k0 = nk0;
        pbuf = npbuf;
        wbuf ^= 1;
    }
}

} // namespace lfi
