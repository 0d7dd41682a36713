// blend_core.hpp — the per-tile arithmetic shared by the LDS-DMA pipelines (blend_persist, blend_wave): the k-loop of one
// unit from LDS (TEN_WM: fp16 MFMA on subnormal pixel bytes; STD: exact fp32 MFMA) and the epilogue (quantise + stores).
//
// Layout the k-loops read: pixels [image][TPX] dwords (a lane's column pointer `col` is passed in), weight fragments
// [k-octet][view] × 16 B (eight fp16, images 8o … 8o+7).  Operand maps: blend_ten.hpp; packed epilogue: epilogue_packed.hpp.
// Replaces the arithmetic of Kernels::Tensors::process / Kernels::Standard::process (reference src/kernels.cu:289-343, 398-461).
#pragma once

#include "epilogue_packed.hpp"

namespace lfi {

// STD quantisation (uch4, reference src/kernels.cu:301-310): (unsigned char)__float2int_rn(sum), alpha 255.
// sum + 2^23 rounds to an integer with the default round-to-nearest-even and leaves it in the low mantissa bits, so the
// low byte IS the reference's result — including its two's-complement wrap for sums outside 0..255 — for |sum| < 2^22.
__device__ __forceinline__ void quantize_tile_rn(const f32x16 &cr, const f32x16 &cg, const f32x16 &cb, uint32_t (&rgba)[16])
{
#pragma unroll
    for(int e = 0; e < 16; e++)
    {
        const uint32_t tr = __builtin_bit_cast(uint32_t, cr[e] + 8388608.0f);
        const uint32_t tg = __builtin_bit_cast(uint32_t, cg[e] + 8388608.0f);
        const uint32_t tb = __builtin_bit_cast(uint32_t, cb[e] + 8388608.0f);
        const uint32_t rg = __builtin_amdgcn_perm(tg, tr, 0x0c0c0400u); // [R, G, 0, 0]
        rgba[e] = __builtin_amdgcn_perm(tb, rg, 0x0d040100u);           // [R, G, B, 0xff]
    }
}

// TEN_WM k-loop of one unit: kc images (multiple of 16, ≤ 16·KS).  col = this lane's pixel column at the first image of its
// k-half (pixel r, image 8h); ZERO_FIRST: the accumulators are not read by the first k-step (its C operand is the inline
// constant 0), so the caller never has to clear them.
template <int MT, int TPX, int KS, bool ZERO_FIRST>
__device__ __forceinline__ void unit_ten(const uint32_t *col, const u32x4 *w_buf, const int r, const int h, const int kc,
                                         f32x16 (&acc)[MT][3])
{
    constexpr int VPP = MT * 32;
    f32x16 zero16;
#pragma unroll
    for(int e = 0; e < 16; e++)
        zero16[e] = 0.0f;
#pragma unroll
    for(int ks = 0; ks < KS; ks++)
    {
        if(16 * ks < kc)
        {
            half8 wfrag[MT];
#pragma unroll
            for(int m = 0; m < MT; m++)
                wfrag[m] = __builtin_bit_cast(half8, w_buf[(2 * ks + h) * VPP + m * 32 + r]);
            uint32_t px[8];
#pragma unroll
            for(int j = 0; j < 8; j++)
                px[j] = col[(16 * ks + j) * TPX];
            u32x4 bc[3];
#pragma unroll
            for(int q = 0; q < 4; q++)
            {
                bc[0][q] = pack_subnormal_pair<0>(px[2 * q], px[2 * q + 1]);
                bc[1][q] = pack_subnormal_pair<1>(px[2 * q], px[2 * q + 1]);
                bc[2][q] = pack_subnormal_pair<2>(px[2 * q], px[2 * q + 1]);
            }
#pragma unroll
            for(int c = 0; c < 3; c++)
            {
                const half8 bfrag = __builtin_bit_cast(half8, bc[c]);
#pragma unroll
                for(int m = 0; m < MT; m++)
                    acc[m][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfrag[m], bfrag, (ZERO_FIRST && ks == 0) ? zero16 : acc[m][c], 0, 0, 0);
            }
        }
    }
}

// TEN_WM k-loop of one unit from PLANAR pixel bytes (blend_planar.hpp): LDS bytes [channel][image of the chunk][TPX pixels], col =
// this lane's pixel at image 8h of channel 0.  A pixel byte is the mantissa of the fp16 subnormal b·2^-24, so the B operand of
// an image pair is lo | hi << 16 of two byte reads.  Otherwise as unit_ten.
template <int MT, int TPX, int KC, bool ZERO_FIRST>
__device__ __forceinline__ void unit_ten_bytes(const uint8_t *col, const u32x4 *w_buf, const int r, const int h, const int kc,
                                               f32x16 (&acc)[MT][3])
{
    constexpr int VPP = MT * 32, KS = KC / 16;
    f32x16 zero16;
#pragma unroll
    for(int e = 0; e < 16; e++)
        zero16[e] = 0.0f;
#pragma unroll
    for(int ks = 0; ks < KS; ks++)
    {
        if(16 * ks < kc)
        {
            half8 wfrag[MT];
#pragma unroll
            for(int m = 0; m < MT; m++)
                wfrag[m] = __builtin_bit_cast(half8, w_buf[(2 * ks + h) * VPP + m * 32 + r]);
            u32x4 bc[3];
#pragma unroll
            for(int c = 0; c < 3; c++)
#pragma unroll
                for(int q = 0; q < 4; q++)
                {
                    const uint32_t lo = col[(c * KC + 16 * ks + 2 * q) * TPX], hi = col[(c * KC + 16 * ks + 2 * q + 1) * TPX];
                    bc[c][q] = lo | (hi << 16);
                }
#pragma unroll
            for(int c = 0; c < 3; c++)
            {
                const half8 bfrag = __builtin_bit_cast(half8, bc[c]);
#pragma unroll
                for(int m = 0; m < MT; m++)
                    acc[m][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfrag[m], bfrag, (ZERO_FIRST && ks == 0) ? zero16 : acc[m][c], 0, 0, 0);
            }
        }
    }
}

// STD k-loop of one unit, exact fp32: MFMA q of a k-step multiplies image pair (16ks+2q, 16ks+2q+1): k = 0 ↔ lower half-wave,
// k = 1 ↔ upper, accumulated in that order = the reference's ascending-g fmaf chain (src/kernels.cu:328-338).
// col = this lane's pixel column at image h.  Weights, W32 = false: the fp16 fragments of the TEN_WM layout, this lane's view
// row of a k-step's 16 images in two octets (image 2q+h sits in bits [16h, 16h+16) of dword q), widened on the fly (exact);
// W32 = true: f32 weights [image][VPP views] (widened once per workgroup), one ds_read_b32 per operand and no conversion —
// on this path every VALU instruction is a cycle group the fp32 matrix instruction does not get (same FMA lanes).
// The LDS reads of k-step ks+1 are issued BEFORE the 48 MFMAs of k-step ks (register double buffer, scheduling barrier), so
// the matrix pipe never waits for LDS latency inside the loop.
template <int MT, int TPX, int KS, bool ZERO_FIRST, bool W32 = false>
__device__ __forceinline__ void unit_std(const uint32_t *col, const void *weights, const int r, const int h, const int kc,
                                         f32x16 (&acc)[MT][3])
{
    constexpr int VPP = MT * 32;
    f32x16 zero16;
#pragma unroll
    for(int e = 0; e < 16; e++)
        zero16[e] = 0.0f;
    const u32x4 *w_buf = static_cast<const u32x4 *>(weights);
    const float *wf = static_cast<const float *>(weights) + h * VPP + r; // W32: this lane's view column at image h
    uint32_t px[2][8];
    u32x4 wlo[2][MT], whi[2][MT];
    float w32[2][8][MT];
    auto load_step = [&](int ks, int slot) {
#pragma unroll
        for(int q = 0; q < 8; q++)
            px[slot][q] = col[(16 * ks + 2 * q) * TPX];
        if constexpr(W32)
        {
#pragma unroll
            for(int q = 0; q < 8; q++)
#pragma unroll
                for(int m = 0; m < MT; m++)
                    w32[slot][q][m] = wf[(16 * ks + 2 * q) * VPP + m * 32];
        }
        else
        {
#pragma unroll
            for(int m = 0; m < MT; m++)
            {
                wlo[slot][m] = w_buf[(2 * ks) * VPP + m * 32 + r];
                whi[slot][m] = w_buf[(2 * ks + 1) * VPP + m * 32 + r];
            }
        }
    };
    const uint32_t sh = 16u * uint32_t(h);
    load_step(0, 0);
#pragma unroll
    for(int ks = 0; ks < KS; ks++)
    {
        if(16 * ks < kc)
        {
            const int cur = ks & 1;
            if(ks + 1 < KS && 16 * (ks + 1) < kc)
                load_step(ks + 1, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for(int q = 0; q < 8; q++)
            {
                float wq[MT];
#pragma unroll
                for(int m = 0; m < MT; m++)
                {
                    if constexpr(W32)
                        wq[m] = w32[cur][q][m];
                    else
                    {
                        const uint32_t d = q < 4 ? wlo[cur][m][q] : whi[cur][m][q - 4];
                        wq[m] = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(d >> sh))); // exact
                    }
                }
                const uint32_t p = px[cur][q];
                const float pc[3] = {static_cast<float>(p & 0xffu), static_cast<float>((p >> 8) & 0xffu),
                                     static_cast<float>((p >> 16) & 0xffu)};
#pragma unroll
                for(int c = 0; c < 3; c++)
#pragma unroll
                    for(int m = 0; m < MT; m++)
                        acc[m][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[m], pc[c], (ZERO_FIRST && ks == 0 && q == 0) ? zero16 : acc[m][c], 0, 0, 0);
            }
        }
    }
}

// Epilogue of one wave's 32 pixels × (MT·32 views from vbase): quantise and store the views inside [.., a.v1); returns the
// exact number of store instructions issued (every branch around a store is wave-uniform) for the caller's vmcnt
// bookkeeping.  y = row inside the output window, xw = the wave's first column.  CLEAR: zero the accumulators afterwards.
template <bool STD, int MT, bool NT_STORE, bool CLEAR>
__device__ __forceinline__ int store_tile(const KernelArgs &a, f32x16 (&acc)[MT][3], const int vbase, const int y, const int xw,
                                          const int r, const int h, const size_t oplane_px)
{
    const int W = a.width;
    int n_st = 0;
#pragma unroll
    for(int m = 0; m < MT; m++)
    {
        const int view_m = vbase + m * 32;
        const int nvalid = min(a.v1 - view_m, 32); // views of this M-tile inside the launch's range
        if(nvalid > 0 && xw < W)
        {
            uint32_t rgba[16];
            if constexpr(STD)
                quantize_tile_rn(acc[m][0], acc[m][1], acc[m][2], rgba);
            else
                quantize_tile_packed(acc[m][0], acc[m][1], acc[m][2], rgba);
            // wave-uniform 64-bit base per store + one 32-bit per-lane byte offset (4 views × a plane < 4 GB): no per-store
            // vector address arithmetic
            uint8_t *ubase = a.views + ((size_t)view_m * oplane_px + (size_t)y * W + xw) * 4;
            const uint32_t lane_off = (uint32_t(r) + uint32_t(4 * h) * uint32_t(oplane_px)) * 4u;
            if(nvalid == 32 && xw + 32 <= W)
            {
                n_st += 16; // full M-tile, full run: 16 unpredicated stores
#pragma unroll
                for(int e = 0; e < 16; e++)
                {
                    uint32_t *out = reinterpret_cast<uint32_t *>(ubase + (size_t)((e & 3) + 8 * (e >> 2)) * oplane_px * 4 + lane_off);
                    if constexpr(NT_STORE)
                        __builtin_nontemporal_store(rgba[e], out);
                    else
                        *out = rgba[e];
                }
            }
            else
            {
                const bool lane_x_ok = xw + r < W;
#pragma unroll
                for(int e = 0; e < 16; e++)
                {
                    const int vrow = (e & 3) + 8 * (e >> 2); // + 4h in lane_off
                    if(vrow < nvalid) // wave-uniform; lane (r = 0, h = 0) is then always active, so the store is issued
                    {
                        n_st++;
                        uint32_t *out = reinterpret_cast<uint32_t *>(ubase + (size_t)vrow * oplane_px * 4 + lane_off);
                        if(lane_x_ok && vrow + 4 * h < nvalid)
                        {
                            if constexpr(NT_STORE)
                                __builtin_nontemporal_store(rgba[e], out);
                            else
                                *out = rgba[e];
                        }
                    }
                }
            }
        }
        if constexpr(CLEAR)
        {
#pragma unroll
            for(int c = 0; c < 3; c++)
#pragma unroll
                for(int e = 0; e < 16; e++)
                    acc[m][c][e] = 0.0f;
        }
    }
    return n_st;
}

// The same tile into the PLANAR view layout (alpha-free byte planes [view][R,G,B][out_rows][views_pitch], blend_p3.hpp) — round 4, for the
// all-focus TEN_WM render: a lane holds the RGBA dword of ONE pixel (r) per view; the four lanes of a quad (four neighbouring pixels) transpose
// their 4 × 4 bytes with two DPP moves and two v_perm per view — lane 0 of the quad ends up with the four R bytes, lane 2 with G, lane 1 with
// B — and store one dword each: a wave writes a whole 32-byte sector per view and channel.  The pitch is a multiple of 128 ≥ W, so pixels
// past the right edge land in the row's padding.  Returns the number of store instructions issued (wave-uniform).
template <bool STD, int MT, bool NT_STORE, bool CLEAR>
__device__ __forceinline__ int store_tile_planar(const KernelArgs &a, f32x16 (&acc)[MT][3], const int vbase, const int y, const int xw, const int r,
                                                 const int h)
{
    int n_st = 0;
    uint32_t plane_b = uint32_t(a.out_rows) * uint32_t(a.views_pitch); // bytes of a byte plane; opaque per tile (no offsets kept across tiles)
    asm volatile("" : "+s"(plane_b));
    const uint32_t sel1 = (r & 1) ? 0x03070206u : 0x05010400u; // odd lane: [Y.b2, X.b2, Y.b3, X.b3], even: [X.b0, Y.b0, X.b1, Y.b1]
    const uint32_t sel2 = (r & 2) ? 0x03020706u : 0x05040100u; // upper pair: [Z.b2, Z.b3, P.b2, P.b3], lower: [P.b0, P.b1, Z.b0, Z.b1]
    const int role = (r & 3) == 0 ? 0 : ((r & 3) == 2 ? 1 : ((r & 3) == 1 ? 2 : 3)); // the plane this lane stores: R, G, B, none
#pragma unroll
    for(int m = 0; m < MT; m++)
    {
        const int view_m = vbase + m * 32;
        const int nvalid = min(a.v1 - view_m, 32); // views of this M-tile inside the launch's range
        if(nvalid > 0 && xw < a.width)
        {
            uint32_t rgba[16];
            if constexpr(STD)
                quantize_tile_rn(acc[m][0], acc[m][1], acc[m][2], rgba);
            else
                quantize_tile_packed(acc[m][0], acc[m][1], acc[m][2], rgba);
            uint8_t *ubase = a.views + ((size_t)view_m * 3 * a.out_rows + y) * a.views_pitch + xw;
            const uint32_t lane_off = (uint32_t(12 * h) + uint32_t(role)) * plane_b + uint32_t(r & ~3);
#pragma unroll
            for(int e = 0; e < 16; e++)
            {
                const int vrow = (e & 3) + 8 * (e >> 2); // + 4h in lane_off
                const uint32_t swapped = uint32_t(__builtin_amdgcn_mov_dpp(int(rgba[e]), 0xB1, 0xf, 0xf, false)); // quad_perm [1, 0, 3, 2]
                const uint32_t pair = __builtin_amdgcn_perm(swapped, rgba[e], sel1);
                const uint32_t other = uint32_t(__builtin_amdgcn_mov_dpp(int(pair), 0x4E, 0xf, 0xf, false)); // quad_perm [2, 3, 0, 1]
                const uint32_t px4 = __builtin_amdgcn_perm(other, pair, sel2);
                if(vrow < nvalid) // wave-uniform; lane (r = 0, h = 0) is then always active, so the store is issued
                {
                    n_st++;
                    uint32_t *out = reinterpret_cast<uint32_t *>(ubase + uint32_t(3 * vrow) * plane_b + lane_off);
                    if(role < 3 && vrow + 4 * h < nvalid)
                    {
                        if constexpr(NT_STORE)
                            __builtin_nontemporal_store(px4, out);
                        else
                            *out = px4;
                    }
                }
            }
        }
        if constexpr(CLEAR)
        {
#pragma unroll
            for(int c = 0; c < 3; c++)
#pragma unroll
                for(int e = 0; e < 16; e++)
                    acc[m][c][e] = 0.0f;
        }
    }
    return n_st;
}

} // namespace lfi
