// blend_stdxa.hpp — ALL-FOCUS STD (Kernels::Standard::process<true>, reference src/kernels.cu:312-342: the exact fp32 fmaf chain over
// per-pixel warped samples) at matrix-core speed: the band method of blend_stdx.hpp on blend_persist's all-focus pipeline.
//
// Rounds 1–2 ran every all-focus STD render on the exact-fp32 matrix instruction (7.5 ms at config 5, 0.61 of the fp32 peak).  Here, as in
// blend_stdx: fp16 MFMA sums over all chunks of images (descending), the band test and the RGBA stores of the rounded bytes after the last
// one (chunk 0), the sums inside the band queued per wave (two per lane) and recomputed with the chain itself — chunk 0 from the buffer
// still resident, chunk 1 from the buffer the previous unit left, chunks 2 … from a SECOND gather of the same tile (cache traffic) — then
// byte patches.  What differs from blend_stdx:
//   * the pipeline is blend_persist<…, ALLFOCUS>'s: per pixel and image one 4-byte LDS-DMA gather at (int)fma(f, offset, coord) clamped
//     (src/kernels.cu:78-82, :125), f decoded from the focus map; two 32 KB pixel buffers [image][128 pixels] of RGBA dwords and two 8 KB
//     weight buffers per workgroup, one unit ahead; wave = 32 pixels × 64 views on v_mfma_f32_32x32x16_f16;
//   * the chain reads its weights from the weight buffer in LDS (fp16 ×2^15, [k-octet][view] × 16 B), its pixel bytes from the pixel buffer;
//   * the two workgroups of a CU use all 160 KB of LDS, so the queue is compacted across lanes with ds_permute (no LDS allocation) and
//     lives in registers;
//   * one chunk (≤ 64 images) is served too: then there are no C units at all.
// Band, error bounds, preconditions and exactness argument: blend_stdx.hpp.  Bit-exact against the oracle (same tests as every STD kernel).
#pragma once

#include <type_traits>

#include "blend_ten_persist.hpp"

namespace lfi {

template <int N, int I = 0, typename F>
__device__ __forceinline__ void sxa_for_each_unit(F &&f)
{
    if constexpr(I < N)
    {
        f(std::integral_constant<int, I>{});
        sxa_for_each_unit<N, I + 1>(f);
    }
}

// NCH: chunks of 64 images (1 … 4).  One launch renders views [a.v0, min(a.v1, a.v0 + 64)).
// PLANAR_VIEWS (round 4): the views are the planar layout's byte planes [view][R,G,B][out_rows][views_pitch]; the four lanes of a quad transpose
// their RGBA dwords (store_tile_planar, blend_core.hpp) and the chain's patches address a byte of a plane.
template <bool NT_STORE, int NCH, bool PLANAR_VIEWS = false>
__global__ void __launch_bounds__(256, 2) blend_stdxa(const KernelArgs a, const int tiles_x, const int n_tiles)
{
    static_assert(NCH >= 1 && NCH <= 4, "one to four chunks of 64 images");
    constexpr int MT = 2;
    using C = PersistCfg<MT, 64>;
    constexpr int TPX = C::TPX, KC = C::KC, VPP = C::VPP, KS = KC / 16;
    constexpr int NU = 2 * NCH - 1; // units per tile: M(NCH−1) … M(1), MC(0), C(1) … C(NCH−1)
    constexpr int QCAP = 128;       // queued sums per wave and tile: two per lane
    __shared__ __attribute__((aligned(16))) uint32_t lds[C::LDS_DW];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int W = a.width, H = a.height;
    const uint32_t *grid32 = reinterpret_cast<const uint32_t *>(a.grid);
    const size_t plane_px = (size_t)W * (size_t)a.in_rows;
    const size_t oplane_px = (size_t)W * (size_t)a.out_rows;
    // the byte of (view a.v0 + view, pixel px of the wave's 32 at row ty / column xw, channel c) in the views
    auto out_byte = [&](const int ty, const int xw, const uint32_t view, const uint32_t px, const uint32_t c) -> uint8_t * {
        if constexpr(PLANAR_VIEWS)
            return a.views + (((size_t)(a.v0 + int(view)) * 3 + c) * a.out_rows + ty) * a.views_pitch + (xw + int(px));
        else
            return a.views + (((size_t)(a.v0 + int(view)) * oplane_px + (size_t)ty * W + xw + px) * 4 + c);
    };
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane(uint32_t(uintptr_t((lds_ptr_t)lds)));
    typedef const __attribute__((address_space(4))) float *const_float_ptr;
    const const_float_ptr c_offsets = (const_float_ptr)(uintptr_t)a.offsets;
    const uint8_t *map_plane = a.maps + (size_t)a.map_index * (size_t)W * H * 4; // maps are whole-image planes

    auto unit_chunk = [](const int u) { return u < NCH ? NCH - 1 - u : u - NCH + 1; };

    // ---- LDS-DMA of one unit (tile t, chunk) into buffers b: weight fragments + the per-pixel gather (blend_persist's) --------------------
    // the focus values of this lane's two pixels of a tile: fetched one unit before the tile's first gathers, kept for all its chunks
    // (blend_ten_persist.hpp: map_raw / map_focus)
    auto map_raw = [&](const int t, uint32_t &m0, uint32_t &m1) {
        const int ty = t / tiles_x;
        const int y = clampi(a.out_y0 + ty, 0, H - 1);
        const int x0 = (t - ty * tiles_x) * TPX;
        const uint32_t *row = reinterpret_cast<const uint32_t *>(map_plane) + (size_t)y * W;
        m0 = row[clampi(x0 + lane, 0, W - 1)];
        m1 = row[clampi(x0 + 64 + lane, 0, W - 1)];
    };
    auto map_focus = [&](const uint32_t m) { return __builtin_fmaf(__fdiv_rn(static_cast<float>(m & 0xffu), 255.0f), a.range, a.focus); };
    auto issue = [&](const int t, const int chunk, const int b, const float f0, const float f1) {
        const int ty = t / tiles_x;
        const int y = a.out_y0 + ty;
        const int x0 = (t - ty * tiles_x) * TPX;
        const int k0 = KC * chunk;
        const int kc = min(KC, a.k_pad - k0);
        const int kn = min(kc, a.n_images - k0);
        const uint32_t px_addr = lds_base + uint32_t(b) * (C::PX_DW * 4);
        const uint32_t w_addr = lds_base + 2 * C::PX_DW * 4 + uint32_t(b) * (C::W_DW * 4);
        for(int o = wave; 8 * o < kc; o += C::NW)
            if(lane < VPP)
                dma16(a.w16s + (size_t)(a.v0 + lane) * a.k_pad + k0 + 8 * o, w_addr + uint32_t(o) * (VPP * 16));
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
            const int sx0 = clampi(static_cast<int>(__builtin_fmaf(f0, ox, xf0)), 0, W - 1);
            const int sy0 = clampi(static_cast<int>(__builtin_fmaf(f0, oy, yf)), 0, H - 1);
            const int sx1 = clampi(static_cast<int>(__builtin_fmaf(f1, ox, xf1)), 0, W - 1);
            const int sy1 = clampi(static_cast<int>(__builtin_fmaf(f1, oy, yf)), 0, H - 1);
            const uint32_t *base = grid32 + (size_t)(k0 + gi) * plane_px - (size_t)a.in_y0 * W;
            dma4_s(base, (__umul24(uint32_t(sy0), uint32_t(W)) + uint32_t(sx0)) << 2, px_addr + uint32_t(gi) * 512u);
            dma4_s(base, (__umul24(uint32_t(sy1), uint32_t(W)) + uint32_t(sx1)) << 2, px_addr + uint32_t(gi) * 512u + 256u);
            ox = ox_n;
            oy = oy_n;
        }
    };

    // ---- band constants (blend_stdx.hpp) ------------------------------------------------------------------------------------------------
    const float nf = float(a.n_images);
    const float c0 = nf * std_accumulation_bound(a.flags) + 0x1p-12f;
    const float bmax_acc = (c0 + nf * 0x1p-16f) * 0x1p-9f;
    const float base_acc = (0.5f - c0) * 0x1p-9f;
    const float chain_acc = nf * 0x1p-24f;
    auto byte_of15 = [](const float s15) { return __builtin_bit_cast(uint32_t, s15 * 0x1p-15f + 8388608.0f) & 0xffu; }; // the overflow path's scale (bytes as integers)
    auto byte_of = [](const float s9) { return __builtin_bit_cast(uint32_t, s9 * 0x1p9f + 8388608.0f) & 0xffu; };     // the chain's: 2^-9 · sum

    // ---- the chain over one chunk for a queued sum = pixel of the wave (5 bits) | view << 5 (6 bits) | channel << 11, from buffers b -------
    auto chain = [&](const int b, const int kc, const uint32_t entry, float &s15) {
        const uint32_t px = entry & 31u, view = (entry >> 5) & 63u, ch = (entry >> 11) & 3u;
        const uint8_t *pb = reinterpret_cast<const uint8_t *>(lds + b * C::PX_DW) + (wave * 32 + px) * 4 + ch;
        const u32x4 *wb = reinterpret_cast<const u32x4 *>(lds + 2 * C::PX_DW + b * C::W_DW) + view; // [k-octet][view]: eight ×2^15 halves
        float s = s15;
#pragma unroll
        for(int ks = 0; ks < KS; ks++)
        {
            if(16 * ks >= kc) // wave-uniform
                break;
            // all LDS reads of sixteen images before their first fma (blend_stdx.hpp: one LDS latency per image otherwise)
            uint32_t pbyte[16];
            const u32x4 wlo = wb[(2 * ks) * VPP], whi = wb[(2 * ks + 1) * VPP];
#pragma unroll
            for(int j = 0; j < 16; j++)
                pbyte[j] = pb[(16 * ks + j) * TPX * 4];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for(int j = 0; j < 16; j++)
            {
                const uint32_t pair = j < 8 ? wlo[j >> 1] : whi[(j - 8) >> 1];
                const float w = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(j & 1 ? pair >> 16 : pair)));
                // the byte AS an fp16 subnormal (byte·2^-24): v_fma_mix_f32 widens both halves itself, no v_cvt_f32_ubyte per image (blend_stdx.hpp)
                s = __builtin_fmaf(static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(pbyte[j]))), w, s); // addWeighted, src/kernels.cu:292-299
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        s15 = s;
    };

    // ---- unit sequence of this workgroup: tiles j, j + G, …; NU units each; two buffers, one unit ahead --------------------------------------
    const int G = gridDim.x;
    int t = int(tile_of_block(blockIdx.x, gridDim.x, a.flags));
    if(t >= n_tiles)
        return;
    int buf = 0, prev_stores = 0;
    float f0, f1; // the focus values of the tile whose units are being issued
    {
        uint32_t m0, m1;
        map_raw(t, m0, m1);
        f0 = map_focus(m0);
        f1 = map_focus(m1);
    }
    issue(t, unit_chunk(0), 0, f0, f1);
    const int nviews = min(a.v1 - a.v0, VPP);
    uint32_t entry0 = 0u, entry1 = 0u; // bit 15: valid
    float s0 = 0.0f, s1 = 0.0f;
    int queued = 0; // wave-uniform
    f32x16 acc[MT][3];

    auto unit = [&](auto u_tag) -> bool {
        constexpr int u = decltype(u_tag)::value;
        constexpr int cc = u < NCH ? NCH - 1 - u : u - NCH + 1;
        constexpr bool is_m = u < NCH, is_mc = u == NCH - 1, is_last = u == NU - 1;
        // the last unit of a tile issues the next tile's first gathers: that tile's map values now, beside the pieces in flight (younger
        // than the stores counted below: the wait only becomes stricter)
        uint32_t m0n = 0u, m1n = 0u;
        if(u == NU - 1 && t + G < n_tiles)
            map_raw(t + G, m0n, m1n);
        // this wave's pieces of the current unit have landed; the previous epilogue's stores (the youngest operations) may be in flight
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
        const bool last_of_tile = u == NU - 1;
        const int nt = last_of_tile ? t + G : t;
        const bool have_next = nt < n_tiles;
        // The unit after MC(0) is C(1), and chunk 1 of this very tile is what the unit before MC(0), M(1), left in the other pair of buffers
        // (pixels and weights): nothing is fetched for it.  (blend_stdx's ring of three has overwritten that buffer by then.)
        constexpr bool next_is_resident = is_mc && NCH >= 2;
        if(last_of_tile && have_next)
        {
            f0 = map_focus(m0n);
            f1 = map_focus(m1n);
        }
        if(have_next && !next_is_resident)
            issue(nt, unit_chunk(last_of_tile ? 0 : u + 1), buf ^ 1, f0, f1);
        prev_stores = 0;

        const int kc = min(KC, a.k_pad - KC * cc);
        const int ty = t / tiles_x; // row inside the output window
        const int xw = (t - ty * tiles_x) * TPX + wave * 32;
        const uint32_t *px_buf = lds + buf * C::PX_DW;
        const u32x4 *w_buf = reinterpret_cast<const u32x4 *>(lds + 2 * C::PX_DW + buf * C::W_DW);
        if constexpr(is_m)
        {
            if constexpr(u == 0)
                unit_ten<MT, TPX, KS, true>(px_buf + wave * 32 + r + 8 * h * TPX, w_buf, r, h, kc, acc); // the first chunk of a tile: zero C operand
            else
                unit_ten<MT, TPX, KS, false>(px_buf + wave * 32 + r + 8 * h * TPX, w_buf, r, h, kc, acc);
        }
        if constexpr(is_mc)
        {
            // ---- epilogue: round every sum, store RGBA, find the sums inside the band (accumulator e of M-tile m ↔ view 32m + (e&3) + 8(e>>2) + 4h)
            uint32_t flagged[3] = {0u, 0u, 0u}; // per channel: bit 16m + e
            uint32_t valid_mask = 0u;
            const bool lane_x_ok = xw + r < W;
            uint8_t *plane0 = a.views + ((size_t)a.v0 * oplane_px + (size_t)ty * W + xw) * 4;
            const uint32_t lane_off = (uint32_t(r) + uint32_t(4 * h) * uint32_t(oplane_px)) * 4u;
            uint8_t *pbase = PLANAR_VIEWS ? a.views + ((size_t)a.v0 * 3 * a.out_rows + ty) * a.views_pitch + xw : nullptr;
            uint32_t plane_e = uint32_t(a.out_rows) * uint32_t(a.views_pitch);
            asm volatile("" : "+s"(plane_e));
            if(xw < W)
            {
#pragma unroll
                for(int m = 0; m < MT; m++)
                {
                    const int nvalid = min(nviews - m * 32, 32);
                    if(nvalid <= 0)
                        continue;
                    const int lim = max(nvalid - 4 * h, 0);
                    const int n_e = 4 * (lim >> 3) + min(lim & 7, 4); // the lane's valid sums are a prefix of the e sequence
                    valid_mask |= lane_x_ok ? ((n_e >= 16 ? 0xffffu : (1u << n_e) - 1u)) << (16 * m) : 0u;
#pragma unroll
                    for(int e = 0; e < 16; e++)
                    {
                        const int vrow = (e & 3) + 8 * (e >> 2); // + 4h per half-wave
                        if(vrow >= nvalid) // wave-uniform
                            continue;
                        uint32_t bits[3];
#pragma unroll
                        for(int c = 0; c < 3; c++)
                        {
                            const float v = acc[m][c][e];
                            const float tt = v + 16384.0f; // rounds S̃ to an integer (RN-even), left in the low mantissa bits
                            const float dist = v - (tt - 16384.0f);
                            const float pow2 = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v + bmax_acc) & 0x7f800000u);
                            const float inside = __builtin_fmaf(-chain_acc, pow2, base_acc);
                            flagged[c] |= __builtin_fabsf(dist) > inside ? 1u << (16 * m + e) : 0u;
                            bits[c] = __builtin_bit_cast(uint32_t, tt);
                        }
                        const uint32_t rg = __builtin_amdgcn_perm(bits[1], bits[0], 0x0c0c0400u); // [R, G, 0, 0]
                        const uint32_t rgba = __builtin_amdgcn_perm(bits[2], rg, 0x0d040100u);  // [R, G, B, 0xff]
                        prev_stores++; // lane (r = 0, h = 0) is active whenever vrow < nvalid and xw < W: the store is issued
                        if constexpr(PLANAR_VIEWS)
                        {
                            // the quad's 4 × 4 bytes transposed: lanes 0 / 2 / 1 of a quad hold four R / G / B bytes (store_tile_planar)
                            const uint32_t swapped = uint32_t(__builtin_amdgcn_mov_dpp(int(rgba), 0xB1, 0xf, 0xf, false));
                            const uint32_t pair = __builtin_amdgcn_perm(swapped, rgba, (r & 1) ? 0x03070206u : 0x05010400u);
                            const uint32_t other = uint32_t(__builtin_amdgcn_mov_dpp(int(pair), 0x4E, 0xf, 0xf, false));
                            const uint32_t px4 = __builtin_amdgcn_perm(other, pair, (r & 2) ? 0x03020706u : 0x05040100u);
                            const int role = (r & 3) == 0 ? 0 : ((r & 3) == 2 ? 1 : ((r & 3) == 1 ? 2 : 3));
                            if(role < 3 && vrow + 4 * h < nvalid) // (past the right edge: the row's padding — the pitch is a multiple of 128)
                            {
                                // wave-uniform base + a 32-bit offset (64 views × 3 planes < 4 GB: checked on the host), the plane size opaque per tile
                                uint32_t *o4 = reinterpret_cast<uint32_t *>(pbase + (3u * uint32_t(m * 32 + vrow + 4 * h) + uint32_t(role)) * plane_e + uint32_t(r & ~3));
                                if constexpr(NT_STORE)
                                    __builtin_nontemporal_store(px4, o4);
                                else
                                    *o4 = px4;
                            }
                            continue;
                        }
                        uint32_t *out = reinterpret_cast<uint32_t *>(plane0 + (size_t)(m * 32 + vrow) * oplane_px * 4 + lane_off);
                        if(lane_x_ok && vrow + 4 * h < nvalid)
                        {
                            if constexpr(NT_STORE)
                                __builtin_nontemporal_store(rgba, out);
                            else
                                *out = rgba;
                        }
                    }
                }
            }
#pragma unroll
            for(int c = 0; c < 3; c++)
                flagged[c] &= valid_mask;
            // ---- compact the flagged sums across the wave into entry0 (slots 0 … 63) and entry1 (64 … 127) with ds_permute; what does not
            // fit is recomputed here and now from global memory
            entry0 = entry1 = 0u;
            int count = 0;
            while(true)
            {
                const uint32_t any_bits = flagged[0] | flagged[1] | flagged[2];
                const bool mine = any_bits != 0u;
                const uint64_t mk = __builtin_amdgcn_ballot_w64(mine);
                if(mk == 0ull)
                    break;
                const int c = flagged[0] ? 0 : (flagged[1] ? 1 : 2);
                const uint32_t word = c == 0 ? flagged[0] : (c == 1 ? flagged[1] : flagged[2]);
                const int bit = mine ? __builtin_ctz(word) : 0;
                const uint32_t rest = word & (word - 1u);
                flagged[0] = c == 0 ? rest : flagged[0];
                flagged[1] = c == 1 ? rest : flagged[1];
                flagged[2] = c == 2 ? rest : flagged[2];
                const int e = bit & 15, m = bit >> 4;
                const uint32_t view = uint32_t(32 * m + (e & 3) + 8 * (e >> 2) + 4 * h);
                const uint32_t code = 0x8000u | uint32_t(r) | (view << 5) | (uint32_t(c) << 11);
                const int n = __builtin_popcountll(mk);
                const int dest = count + __builtin_amdgcn_mbcnt_hi(uint32_t(mk >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mk), 0u));
                if(count < 64) // wave-uniform: some of this round's sums land in slots below 64
                {
                    // senders write their code to lane `dest`; everybody else writes 0 to a lane outside [count, count + n): 0 or 63
                    const int lo_end = min(count + n, 64);
                    const bool send = mine && dest < 64;
                    const int tgt = send ? dest : (count > 0 ? 0 : 63);
                    const uint32_t got = uint32_t(__builtin_amdgcn_ds_permute(4 * tgt, int(send ? code : 0u)));
                    entry0 = (lane >= count && lane < lo_end) ? got : entry0;
                }
                if(count + n > 64 && count < QCAP) // … and some in slots 64 … 127
                {
                    const int hi_lo = max(count, 64) - 64, hi_hi = min(count + n, QCAP) - 64;
                    const bool send = mine && dest >= 64 && dest < QCAP;
                    const int tgt = send ? dest - 64 : (hi_lo > 0 ? 0 : 63);
                    const uint32_t got = uint32_t(__builtin_amdgcn_ds_permute(4 * tgt, int(send ? code : 0u)));
                    entry1 = (lane >= hi_lo && lane < hi_hi) ? got : entry1;
                }
                const bool spill = mine && dest >= QCAP;
                if(__builtin_amdgcn_ballot_w64(spill) != 0ull) // wave-uniform; rare (adversarial inputs: every sum a tie)
                {
                    if(spill)
                    {
                        const int x = xw + r, y = a.out_y0 + ty;
                        const float f = decode_focus(map_plane, W, H, x, y, a.focus, a.range);
                        const uint16_t *wrow = a.w16s + (size_t)(a.v0 + view) * a.k_pad;
                        float s = 0.0f;
                        for(int g = 0; g < a.n_images; g++)
                        {
                            const lfi_float2 o = a.offsets[g];
                            const int sx = clampi(warp_float(x, f, o.x), 0, W - 1), sy = clampi(warp_float(y, f, o.y), 0, H - 1) - a.in_y0;
                            const uint32_t p = (grid32[(size_t)g * plane_px + (size_t)sy * W + sx] >> (8 * c)) & 0xffu;
                            const float w = static_cast<float>(__builtin_bit_cast(_Float16, wrow[g]));
                            s = __builtin_fmaf(static_cast<float>(p), w, s);
                        }
                        *out_byte(ty, xw, view, uint32_t(r), uint32_t(c)) = static_cast<uint8_t>(byte_of15(s));
                    }
                }
                count += n;
            }
            queued = min(count, QCAP);
            s0 = s1 = 0.0f;
        }
        if constexpr(!is_m || is_mc)
        {
            // chain over chunk cc for the queued sums: MC — chunk 0, still in its buffers; C — a chunk gathered again
            if(queued > 0)
                chain(buf, kc, entry0, s0);
            if(queued > 64)
                chain(buf, kc, entry1, s1);
        }
        if constexpr(is_last)
        {
            // the chain's bytes over the rounded ones (same wave as the dword stores: in order).  Not counted in prev_stores (an undercount
            // only makes the next wait stricter).
            if(entry0 & 0x8000u)
                *out_byte(ty, xw, (entry0 >> 5) & 63u, entry0 & 31u, (entry0 >> 11) & 3u) = static_cast<uint8_t>(byte_of(s0));
            if(queued > 64 && (entry1 & 0x8000u))
                *out_byte(ty, xw, (entry1 >> 5) & 63u, entry1 & 31u, (entry1 >> 11) & 3u) = static_cast<uint8_t>(byte_of(s1));
        }
        if(!have_next)
            return false;
        t = nt;
        buf ^= 1;
        return true;
    };
    bool more = true;
    while(more)
        sxa_for_each_unit<NU>([&](auto u_tag) {
            if(more)
                more = unit(u_tag);
        });
}

} // namespace lfi
