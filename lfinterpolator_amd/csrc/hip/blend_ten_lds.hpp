// blend_ten_lds.hpp — TEN_WM with the view-stack tile staged in LDS by LDS-DMA (the production kernel).
//
// Same contraction as blend_ten.hpp (out[view][pixel] = Σ_image W[view][image]·I_image[pixel+offset_image] on
// v_mfma_f32_32x32x16_f16, pixel bytes entering as fp16 subnormals b·2^-24, fp32 accumulation), restructured around what
// the counters of the direct kernel showed — it was VALU-bound (≈2000 VALU instructions per 32-pixel tile, most of them the
// float→fp16→u8 epilogue and per-load address arithmetic), not HBM-bound:
//
//   load phase     the workgroup copies, for every image of the K-chunk, the (shifted, clamped) run of TPX pixels of its
//                  output row straight from HBM into LDS with global_load_lds: no VGPR, ~15 VALU instructions per 1 KiB,
//                  the whole chunk (64 images × 128 pixels = 32 KB) in flight at once.  Interior runs are 16 B/lane pieces,
//                  border runs and all-focus gathers 4 B/lane pieces with per-pixel clamp-to-edge addresses.
//                  LDS image: tile[image][pixel] dwords = the "view-stack tile".  The weight fragments of the chunk are
//                  fetched into registers while the DMA is in flight.
//   compute phase  per 16-image k-step a lane reads its pixel column from 8 images (ds_read, immediate offsets), zero-extends
//                  bytes to fp16 subnormals (one v_perm_b32 per two values) and issues the MFMAs.  With one chunk (N ≤ KC)
//                  the tile stays in LDS for every view pass of the workgroup.
//   epilogue       packed fp16: the weights are uploaded pre-scaled by 2^15, so acc = true·2^-9; v_cvt_pk_f16_f32 rounds
//                  two accumulators to fp16 (RN-even — this IS the reference's fp16 accumulator rounding, exact because
//                  scaling by 2^-9 commutes with it wherever the result can be non-zero), v_pk_min_f16 saturates at 255,
//                  and v_pk_add_f16 with 2.0 under a round-toward-zero window leaves floor(true) in the low mantissa bits
//                  (2 + n·2^-9 has bit pattern 0x4000|n): truncation like __half2uchar_rz (reference src/kernels.cu:393)
//                  for 1 VALU per two values.  Three v_perm_b32 assemble two RGBA pixels.  ≈6 VALU per pixel·view.
//
// Preconditions of the packed epilogue: every weight finite and in [0, 2) (the reference's weights are a convex
// combination: src/interpolator.cu:156-172).  lfi_set_params checks this; other weights use the generic kernel.
//
// Replaces Kernels::Tensors::process<allFocus> (reference src/kernels.cu:398-461): its shared-memory pixel staging
// (pixelsToSharedMemory :372-385) exists to transpose the tile for WMMA; here LDS is the landing zone of an asynchronous
// gather and no transpose is needed.
#pragma once

#include "blend_ten.hpp"
#include "lfi_device.hpp"

namespace lfi {

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

#define LFI_QA(i) "v_pk_min_f16 %" #i ", %" #i ", %24\n\tv_pk_add_f16 %" #i ", %" #i ", %25\n\t"

// 16 accumulators (one 32-view M-tile of one pixel column) × 3 channels, acc = true·2^-9  →  16 RGBA8 dwords
__device__ __forceinline__ void quantize_tile_packed(const f32x16 &cr, const f32x16 &cg, const f32x16 &cb, uint32_t (&rgba)[16])
{
    uint32_t h[24]; // [channel][pair of consecutive accumulator registers = two views]
#pragma unroll
    for(int p = 0; p < 8; p++)
    {
        const float2_t fr = {cr[2 * p], cr[2 * p + 1]}, fg = {cg[2 * p], cg[2 * p + 1]}, fb = {cb[2 * p], cb[2 * p + 1]};
        h[p] = __builtin_bit_cast(uint32_t, __builtin_convertvector(fr, half2_t));      // v_cvt_pk_f16_f32: RN-even
        h[8 + p] = __builtin_bit_cast(uint32_t, __builtin_convertvector(fg, half2_t));
        h[16 + p] = __builtin_bit_cast(uint32_t, __builtin_convertvector(fb, half2_t));
    }
    const uint32_t k255 = 0x37f837f8u; // 255·2^-9 twice
    const uint32_t two = 0x40004000u;  // 2.0 twice
    // Everything that depends on the fp16 rounding mode sits in ONE asm statement together with the two mode writes
    // (MODE[3:2] = fp16/fp64 rounding: 3 = toward zero, 0 = nearest even), so nothing else can be scheduled inside the window.
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3\n\ts_nop 1\n\t" LFI_QA(0) LFI_QA(1) LFI_QA(2) LFI_QA(3) LFI_QA(4)
                     LFI_QA(5) LFI_QA(6) LFI_QA(7) LFI_QA(8) LFI_QA(9) LFI_QA(10) LFI_QA(11) LFI_QA(12) LFI_QA(13) LFI_QA(14)
                         LFI_QA(15) LFI_QA(16) LFI_QA(17) LFI_QA(18) LFI_QA(19) LFI_QA(20) LFI_QA(21) LFI_QA(22) LFI_QA(23)
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0\n\ts_nop 1"
                 : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(h[4]), "+v"(h[5]), "+v"(h[6]), "+v"(h[7]), "+v"(h[8]),
                   "+v"(h[9]), "+v"(h[10]), "+v"(h[11]), "+v"(h[12]), "+v"(h[13]), "+v"(h[14]), "+v"(h[15]), "+v"(h[16]),
                   "+v"(h[17]), "+v"(h[18]), "+v"(h[19]), "+v"(h[20]), "+v"(h[21]), "+v"(h[22]), "+v"(h[23])
                 : "s"(k255), "s"(two));
    // each half now holds 0x4000 | n with n = floor(true) ≤ 255 in its low byte
#pragma unroll
    for(int p = 0; p < 8; p++)
    {
        const uint32_t rg = __builtin_amdgcn_perm(h[8 + p], h[p], 0x06020400u); // [R0, G0, R1, G1]
        rgba[2 * p] = __builtin_amdgcn_perm(h[16 + p], rg, 0x0d040100u);         // [R0, G0, B0, 0xff]
        rgba[2 * p + 1] = __builtin_amdgcn_perm(h[16 + p], rg, 0x0d060302u);     // [R1, G1, B1, 0xff]
    }
}

// NT   pixels per lane (N-tiles per wave): lane r of a wave owns pixels NT*r … NT*r+NT-1 of the wave's run
// MT   32-view M-tiles per wave
// WPX  waves of the workgroup side by side along the row; WV waves stacked along views
// KC   images per LDS chunk (multiple of 16)
// WPE  workgroups per CU the register allocation must allow
template <int NT, int MT, int WPX, int WV, int KC, int WPE, bool ALLFOCUS>
__global__ void __launch_bounds__(WPX *WV * 64, (WPE * WPX * WV + 3) / 4)
    blend_ten_lds(const KernelArgs a, const int tiles_x, const int n_tiles, const int view_passes)
{
    constexpr int NW = WPX * WV;          // waves per workgroup
    constexpr int WAVE_PX = 32 * NT;      // pixels per wave
    constexpr int TPX = WPX * WAVE_PX;    // pixels per workgroup tile (one row)
    constexpr int VPP = WV * MT * 32;     // views per pass of the workgroup
    constexpr int KS = KC / 16;           // k-steps per chunk
    static_assert(TPX % 64 == 0 && KC % 16 == 0, "tile geometry");
    __shared__ __attribute__((aligned(16))) uint32_t tile[KC * TPX];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wpx = wave % WPX, wv = wave / WPX;

    const int t = int(xcd_contiguous(blockIdx.x, gridDim.x));
    if(t >= n_tiles)
        return;
    const int y = t / tiles_x;
    const int x0 = (t - y * tiles_x) * TPX;
    const int W = a.width, H = a.height;
    const uint32_t *grid32 = reinterpret_cast<const uint32_t *>(a.grid);
    const size_t plane_px = (size_t)W * (size_t)H;
    const int xw = x0 + wpx * WAVE_PX;  // first pixel of this wave in the compute phase (wave-uniform)
    const int xl = xw + r * NT;         // first pixel of this lane
    const bool single_chunk = a.k_pad <= KC;

    for(int pass = 0; pass < view_passes; pass++)
    {
        const int vbase = pass * VPP + wv * (MT * 32);
        f32x16 acc[MT][NT][3];
#pragma unroll
        for(int m = 0; m < MT; m++)
#pragma unroll
            for(int i = 0; i < NT; i++)
#pragma unroll
                for(int c = 0; c < 3; c++)
#pragma unroll
                    for(int e = 0; e < 16; e++)
                        acc[m][i][c][e] = 0.0f;

        for(int k0 = 0; k0 < a.k_pad; k0 += KC)
        {
            const int kc = min(KC, a.k_pad - k0);    // images in this chunk (multiple of 16)
            const int kn = min(kc, a.n_images - k0); // of which real (the rest carries zero weights: any bytes will do)
            const bool load = pass == 0 || !single_chunk;
            if(load)
            {
                if(k0 > 0 || pass > 0)
                    __syncthreads(); // every wave has finished reading the previous chunk

                // ------------- load phase: LDS-DMA of the chunk -------------------------------------------------------------
                // The LDS image is cut into 64-dword units (64 pixels of one image); one 16 B/lane instruction fills four
                // consecutive units: lane l lands on dword 256q + 4l, i.e. image (256q+4l)/TPX, pixel (256q+4l)%TPX.
                const int n_units = kn * (TPX / 64);
                if constexpr(!ALLFOCUS)
                {
                    for(int q = wave; 4 * q < n_units; q += NW)
                    {
                        const int d = 256 * q + 4 * lane;
                        const int gi = min(d / TPX, kn - 1);
                        const int pxo = d % TPX;
                        const lfi_int2 off = a.focused[k0 + gi];
                        const int sy = clampi(y + off.y, 0, H - 1);
                        const int sx = x0 + off.x + pxo;
                        const bool inside = (sx >= 0) && (sx + 4 <= W) && (d / TPX < kn);
                        if(__builtin_amdgcn_ballot_w64(inside) == ~0ull)
                        {
                            const uint32_t *src = grid32 + (size_t)(k0 + gi) * plane_px + (size_t)sy * W + sx;
                            __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(tile + 256 * q), 16, 0, 0);
                        }
                        else
                        {
                            // a run crosses the left/right image border (or the tile is ragged): per-pixel clamp-to-edge
                            // addresses (reference src/kernels.cu:125), one 64-pixel unit per 4 B/lane instruction
#pragma unroll
                            for(int s = 0; s < 4; s++)
                            {
                                const int u = 4 * q + s; // wave-uniform
                                if(u >= n_units)
                                    break;
                                const int g = k0 + u / (TPX / 64);
                                const int px0 = (u % (TPX / 64)) * 64;
                                const lfi_int2 o2 = a.focused[g];
                                const int sy2 = clampi(y + o2.y, 0, H - 1);
                                const int sx2 = clampi(x0 + o2.x + px0 + lane, 0, W - 1);
                                const uint32_t *src = grid32 + (size_t)g * plane_px + (size_t)sy2 * W + sx2;
                                __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(tile + 64 * u), 4, 0, 0);
                            }
                        }
                    }
                }
                else
                {
                    // all-focus: the warp depends on the pixel's own focus value (src/kernels.cu:78-82): per-pixel gather
                    const uint8_t *map_plane = a.maps + (size_t)a.map_index * plane_px * 4;
                    float fpx[TPX / 64];
#pragma unroll
                    for(int s = 0; s < TPX / 64; s++)
                        fpx[s] = decode_focus(map_plane, W, H, x0 + s * 64 + lane, y, a.focus, a.range);
                    for(int gi = wave; gi < kn; gi += NW)
                    {
                        const int g = k0 + gi;
                        const lfi_float2 off = a.offsets[g];
                        const uint32_t *plane = grid32 + (size_t)g * plane_px;
#pragma unroll
                        for(int s = 0; s < TPX / 64; s++)
                        {
                            const int sx = clampi(warp_float(x0 + s * 64 + lane, fpx[s], off.x), 0, W - 1);
                            const int sy = clampi(warp_float(y, fpx[s], off.y), 0, H - 1);
                            __builtin_amdgcn_global_load_lds(plane + (size_t)sy * W + sx, (lds_ptr_t)(tile + gi * TPX + s * 64), 4, 0, 0);
                        }
                    }
                }
            }

            // weight fragments of the whole chunk, fetched while the DMA is in flight: lane (r,h) of M-tile m holds
            // W[view m*32+r][k0 + 16 ks + 8h … +7] (scaled by 2^15) = 16 contiguous bytes of the padded matrix
            half8 wfrag[KS][MT];
#pragma unroll
            for(int ks = 0; ks < KS; ks++)
#pragma unroll
                for(int m = 0; m < MT; m++)
                {
                    const int kk = min(16 * ks, kc - 16); // k-steps past the chunk's end are never used
                    const uint16_t *wrow = a.w16s + (size_t)(a.v0 + vbase + m * 32 + r) * a.k_pad + k0 + kk + 8 * h;
                    wfrag[ks][m] = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4 *>(wrow));
                }

            if(load)
            {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's pieces have landed
                __syncthreads();                                  // … and everybody else's
            }

            // ------------- compute phase ------------------------------------------------------------------------------------------
            const uint32_t *col = tile + xw - x0 + r * NT + 8 * h * TPX; // this lane's pixel column, first image of its k-half
#pragma unroll
            for(int ks = 0; ks < KS; ks++)
            {
                if(16 * ks < kc)
                {
                    uint32_t px[8][NT];
#pragma unroll
                    for(int j = 0; j < 8; j++)
                    {
                        const uint32_t *src = col + (16 * ks + j) * TPX;
                        if constexpr(NT == 1)
                            px[j][0] = src[0];
                        else if constexpr(NT == 2)
                        {
                            const u32x2 v = *reinterpret_cast<const u32x2 *>(src);
                            px[j][0] = v.x;
                            px[j][1] = v.y;
                        }
                        else
                        {
                            const u32x4 v = *reinterpret_cast<const u32x4 *>(src);
                            px[j][0] = v.x;
                            px[j][1] = v.y;
                            px[j][2] = v.z;
                            px[j][3] = v.w;
                        }
                    }
#pragma unroll
                    for(int i = 0; i < NT; i++)
                    {
                        u32x4 bc[3];
#pragma unroll
                        for(int q = 0; q < 4; q++)
                        {
                            bc[0][q] = pack_subnormal_pair<0>(px[2 * q][i], px[2 * q + 1][i]);
                            bc[1][q] = pack_subnormal_pair<1>(px[2 * q][i], px[2 * q + 1][i]);
                            bc[2][q] = pack_subnormal_pair<2>(px[2 * q][i], px[2 * q + 1][i]);
                        }
#pragma unroll
                        for(int c = 0; c < 3; c++)
                        {
                            const half8 b = __builtin_bit_cast(half8, bc[c]);
#pragma unroll
                            for(int m = 0; m < MT; m++)
                                acc[m][i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfrag[ks][m], b, acc[m][i][c], 0, 0, 0);
                        }
                    }
                }
            }
        }

        // ------------- epilogue: packed fp16 rounding + truncation, RGBA assembly, stores ------------------------------------------
        const bool run_inside = (xl + NT <= W) && (NT == 1 || (W % NT) == 0);
#pragma unroll
        for(int m = 0; m < MT; m++)
        {
            uint32_t rgba[NT][16];
#pragma unroll
            for(int i = 0; i < NT; i++)
                quantize_tile_packed(acc[m][i][0], acc[m][i][1], acc[m][i][2], rgba[i]);
            // wave-uniform base of M-tile m at this wave's first pixel; the lane adds its pixel and its half-wave's 4 views
            const int view_m = a.v0 + vbase + m * 32;
            uint32_t *ubase = reinterpret_cast<uint32_t *>(a.views) + (size_t)view_m * plane_px + (size_t)y * W + xw;
            const uint32_t lane_off = uint32_t(r * NT) + uint32_t(4 * h) * uint32_t(plane_px);
#pragma unroll
            for(int e = 0; e < 16; e++)
            {
                const int vrow = (e & 3) + 8 * (e >> 2); // + 4h in lane_off
                if(view_m + vrow + 4 * h >= a.v1)
                    continue;
                uint32_t *out = ubase + (size_t)vrow * plane_px + lane_off;
                uint32_t px[NT];
#pragma unroll
                for(int i = 0; i < NT; i++)
                    px[i] = rgba[i][e];
                if(run_inside)
                    store_run<NT>(out, px);
                else
                {
#pragma unroll
                    for(int i = 0; i < NT; i++)
                        if(xl + i < W)
                            out[i] = px[i];
                }
            }
        }
    }
}

} // namespace lfi
