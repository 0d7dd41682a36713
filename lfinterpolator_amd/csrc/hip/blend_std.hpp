// blend_std.hpp — the STD path: exact-fp32 ordered FMA chain, round-to-nearest-even quantisation.
//
// Replaces Kernels::Standard::process<allFocus> (reference src/kernels.cu:312-342; addWeighted :292-299, uch4 :301-310):
//      sum[v][c] = fmaf(float(px_g[c]), float(w_half[v][g]), sum[v][c])   for g = 0 … N-1 in order
//      out[v][c] = (unsigned char)__float2int_rn(sum[v][c]),  alpha = 255
//
// Two kernels compute exactly that chain:
//  * blend_std_valu  — one pixel per lane, v_fma_f32, weights through scalar loads: the plain wavefront kernel and the
//                      bit-exactness anchor (every operation is an IEEE fma in the reference's order by construction);
//  * blend_std_mfma  — v_mfma_f32_32x32x2_f32, whose result is bit-for-bit the k-ordered fmaf chain
//                      D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)) (cdna_hip_programming.md §3 "FP32-input MFMA"); it runs at the
//                      fp32 vector peak while leaving the VALU free for address arithmetic and the epilogue.
//                      Same orientation as the TEN_WM kernel: A = weights (lane l: view l&31, image l>>5),
//                      B = pixels (lane l: pixel l&31, image l>>5), so the image pair of one MFMA is (2q, 2q+1) in order.
#pragma once

#include <type_traits>

#include "blend_ten.hpp"
#include "lfi_device.hpp"

namespace lfi {

__device__ __forceinline__ uint32_t quant_rn(float s)
{
    return static_cast<uint32_t>(__float2int_rn(s)) & 0xffu; // uch4: src/kernels.cu:301-310
}

template <bool ALLFOCUS, int VCH>
__global__ void __launch_bounds__(256) blend_std_valu(const KernelArgs a)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int W = a.width, H = a.height;
    if(x >= W || y >= H)
        return;
    const size_t plane_px = (size_t)W * (size_t)H;
    float focus_px = 0.0f;
    if constexpr(ALLFOCUS)
        focus_px = decode_focus(a.maps + (size_t)a.map_index * plane_px * 4, W, H, x, y, a.focus, a.range);

    for(int vb = a.v0; vb < a.v1; vb += VCH)
    {
        float acc[VCH][3];
#pragma unroll
        for(int v = 0; v < VCH; v++)
            acc[v][0] = acc[v][1] = acc[v][2] = 0.0f;
        for(int g = 0; g < a.n_images; g++) // ascending g: src/kernels.cu:328
        {
            int sx, sy;
            if constexpr(ALLFOCUS)
            {
                const lfi_float2 off = a.offsets[g];
                sx = warp_float(x, focus_px, off.x);
                sy = warp_float(y, focus_px, off.y);
            }
            else
            {
                const lfi_int2 off = a.focused[g];
                sx = x + off.x;
                sy = y + off.y;
            }
            const uint32_t px = fetch_px(a.grid, W, H, g, sx, sy);
            const float pr = static_cast<float>(px & 0xffu);
            const float pg = static_cast<float>((px >> 8) & 0xffu);
            const float pb = static_cast<float>((px >> 16) & 0xffu);
            const float *__restrict__ wrow = a.w32t + (size_t)g * a.v_pad + vb; // wave-uniform → scalar loads
#pragma unroll
            for(int v = 0; v < VCH; v++)
            {
                const float w = wrow[v];
                acc[v][0] = __builtin_fmaf(pr, w, acc[v][0]);
                acc[v][1] = __builtin_fmaf(pg, w, acc[v][1]);
                acc[v][2] = __builtin_fmaf(pb, w, acc[v][2]);
            }
        }
#pragma unroll
        for(int v = 0; v < VCH; v++)
        {
            const int view = vb + v;
            if(view >= a.v1)
                break;
            const uint32_t rgba = quant_rn(acc[v][0]) | (quant_rn(acc[v][1]) << 8) | (quant_rn(acc[v][2]) << 16) | 0xff000000u;
            reinterpret_cast<uint32_t *>(a.views)[(size_t)view * plane_px + (size_t)y * W + x] = rgba;
            if(a.prequant != nullptr && view == a.prequant_view)
            {
                float *pq = a.prequant + ((size_t)y * W + x) * 3;
                pq[0] = acc[v][0];
                pq[1] = acc[v][1];
                pq[2] = acc[v][2];
            }
        }
    }
}

// The non-tensor wavefront kernel (north_star: "the non-tensor path kept as a coalesced-HBM wavefront kernel"): one pixel per lane,
// ONE pass over the inputs for 64 views — 192 fp32 accumulators per lane as 96 register pairs, v_pk_fma_f32 on (view 2j, view 2j+1)
// pairs with the pair's weights in an SGPR pair (scalar loads of w32t rows: wave-uniform) and the pixel value broadcast to both
// halves; each accumulator still sees its images in ascending order, one IEEE fma per image: the reference's chain, bit for bit
// (src/kernels.cu:292-299, 328-338).  The pixel of image g + 2 is fetched while image g is accumulated (two waves per SIMD cover
// the rest of the latency); a half-wave reads 128 contiguous bytes per image row, stores are 256 contiguous bytes per view.
// Bound: the fp32 vector pipe — 6·N·V flops per pixel at 157.3 TFLOP/s.
typedef float float2v __attribute__((ext_vector_type(2)));

template <bool ALLFOCUS>
__global__ void __launch_bounds__(256, 2) blend_std_vfma(const KernelArgs a)
{
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * 64 + lane;
    const int y = blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = a.width, H = a.height;
    if(y >= H)
        return;
    const bool x_ok = x < W;
    const int xc = min(x, W - 1);
    const size_t plane_px = (size_t)W * (size_t)H;
    const uint32_t *grid32 = reinterpret_cast<const uint32_t *>(a.grid);
    // offsets and weights are read through the constant address space: wave-uniform scalar loads (lgkmcnt), which also keeps the
    // vector-memory queue to the pixel fetches (the compiler cannot prove by itself that the views do not alias them)
    typedef const __attribute__((address_space(4))) int32_t *const_int_ptr;
    typedef const __attribute__((address_space(4))) float *const_float_ptr;
    const const_int_ptr c_focused = (const_int_ptr)(uintptr_t)a.focused;
    const const_float_ptr c_offsets = (const_float_ptr)(uintptr_t)a.offsets;
    const const_float_ptr c_w32t = (const_float_ptr)(uintptr_t)a.w32t;
    float focus_px = 0.0f;
    if constexpr(ALLFOCUS)
        focus_px = decode_focus(a.maps + (size_t)a.map_index * plane_px * 4, W, H, xc, y, a.focus, a.range);
    // per-image offsets (integer, or float for the all-focus warp) as two dwords, read one stage ahead of the fetch that uses them
    struct Off2
    {
        uint32_t x, y;
    };
    auto load_off = [&](const int g) {
        Off2 o;
        if constexpr(ALLFOCUS)
        {
            o.x = __builtin_bit_cast(uint32_t, c_offsets[2 * g]);
            o.y = __builtin_bit_cast(uint32_t, c_offsets[2 * g + 1]);
        }
        else
        {
            o.x = uint32_t(c_focused[2 * g]);
            o.y = uint32_t(c_focused[2 * g + 1]);
        }
        return o;
    };
    auto fetch = [&](const int g, const Off2 o) {
        int sx, sy;
        if constexpr(ALLFOCUS)
        {
            sx = warp_float(xc, focus_px, __builtin_bit_cast(float, o.x));
            sy = warp_float(y, focus_px, __builtin_bit_cast(float, o.y));
        }
        else
        {
            sx = xc + int(o.x);
            sy = y + int(o.y);
        }
        return grid32[(size_t)g * plane_px + (size_t)clampi(sy, 0, H - 1) * W + clampi(sx, 0, W - 1)];
    };
    const int n = a.n_images;
    for(int vb = a.v0; vb < a.v1; vb += 64)
    {
        float2v acc[32][3];
#pragma unroll
        for(int j = 0; j < 32; j++)
#pragma unroll
            for(int c = 0; c < 3; c++)
                acc[j][c] = float2v{0.0f, 0.0f};
        // Weights: 32 per stage (two s_load_dwordx16) into one of two SGPR sets; a stage first makes sure ITS set has arrived, then
        // issues the loads of the next stage's set, then runs its 48 v_pk_fma_f32 — scalar loads return out of order, so the only
        // wait there is is lgkmcnt(0), and it has to sit BEFORE the next loads are issued for their latency to hide under the FMAs.
        auto load32 = [&](const int g, const int half, float (&w)[32]) {
            const const_float_ptr wrow = c_w32t + (size_t)g * a.v_pad + vb + 32 * half; // rows are padded to 64 + 64 views
#pragma unroll
            for(int i = 0; i < 32; i++)
                w[i] = wrow[i];
        };
        auto arrived = [&](const float (&w)[32]) { asm volatile("" ::"s"(w[0]), "s"(w[16])); };
        auto fma32 = [&](auto half_tag, const float (&w)[32], const float2v r2, const float2v g2, const float2v b2) {
            constexpr int HALF = decltype(half_tag)::value;
#pragma unroll
            for(int j = 0; j < 16; j++)
            {
                const float2v w2 = {w[2 * j], w[2 * j + 1]};
                acc[16 * HALF + j][0] = __builtin_elementwise_fma(r2, w2, acc[16 * HALF + j][0]);
                acc[16 * HALF + j][1] = __builtin_elementwise_fma(g2, w2, acc[16 * HALF + j][1]);
                acc[16 * HALF + j][2] = __builtin_elementwise_fma(b2, w2, acc[16 * HALF + j][2]);
            }
        };
        // Pixels: a ring of four registers, image g + 3 is fetched while image g is accumulated (no register moves: the loop is
        // unrolled by four).  Scalar loads (weights, offsets) are issued only right after a stage's lgkmcnt(0), so each has a whole
        // stage of FMAs to arrive.
        float wA[32], wB[32];
        load32(0, 0, wA);
        uint32_t px[4];
#pragma unroll
        for(int u = 0; u < 3; u++)
            px[u] = fetch(min(u, n - 1), load_off(min(u, n - 1)));
        Off2 off_next = load_off(min(3, n - 1));
        for(int g0 = 0; g0 < n; g0 += 4) // ascending g: src/kernels.cu:328
        {
#pragma unroll
            for(int u = 0; u < 4; u++)
            {
                const int g = g0 + u;
                if(g >= n) // wave-uniform
                    break;
                arrived(wA); // … and off_next
                px[(u + 3) & 3] = fetch(min(g + 3, n - 1), off_next);
                load32(g, 1, wB);
                off_next = load_off(min(g + 4, n - 1));
                const uint32_t p0 = px[u];
                const float pr = static_cast<float>(p0 & 0xffu), pg = static_cast<float>((p0 >> 8) & 0xffu), pb = static_cast<float>((p0 >> 16) & 0xffu);
                const float2v r2 = {pr, pr}, g2 = {pg, pg}, b2 = {pb, pb};
                __builtin_amdgcn_sched_barrier(0);
                fma32(std::integral_constant<int, 0>{}, wA, r2, g2, b2);
                __builtin_amdgcn_sched_barrier(0);
                arrived(wB);
                load32(min(g + 1, n - 1), 0, wA);
                __builtin_amdgcn_sched_barrier(0);
                fma32(std::integral_constant<int, 1>{}, wB, r2, g2, b2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        uint32_t *out = reinterpret_cast<uint32_t *>(a.views) + (size_t)vb * plane_px + (size_t)y * W + x;
#pragma unroll
        for(int j = 0; j < 32; j++)
#pragma unroll
            for(int e = 0; e < 2; e++)
            {
                const int view = vb + 2 * j + e;
                if(view < a.v1 && x_ok)
                {
                    // uch4 (src/kernels.cu:301-310): sum + 2^23 leaves (unsigned char)__float2int_rn(sum) in the low mantissa byte
                    const uint32_t tr = __builtin_bit_cast(uint32_t, acc[j][0][e] + 8388608.0f), tg = __builtin_bit_cast(uint32_t, acc[j][1][e] + 8388608.0f),
                                   tb = __builtin_bit_cast(uint32_t, acc[j][2][e] + 8388608.0f);
                    const uint32_t rg = __builtin_amdgcn_perm(tg, tr, 0x0c0c0400u);
                    __builtin_nontemporal_store(__builtin_amdgcn_perm(tb, rg, 0x0d040100u), out + (size_t)(2 * j + e) * plane_px);
                    if(a.prequant != nullptr && view == a.prequant_view)
                    {
                        float *pq = a.prequant + ((size_t)y * W + x) * 3;
                        pq[0] = acc[j][0][e];
                        pq[1] = acc[j][1][e];
                        pq[2] = acc[j][2][e];
                    }
                }
            }
    }
}

// Same tiling as blend_ten_direct (one wave = 32*PXL pixels of a row × 32*MT views per pass) on the exact-f32 MFMA.
template <int PXL, int MT, bool ALLFOCUS>
__global__ void __launch_bounds__(256) blend_std_mfma(const KernelArgs a, const int tiles_x, const int n_tiles,
                                                      const int view_passes, const int vpw)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;

    const int tiles_per_wg = 4 / vpw;
    const int tile = int(xcd_contiguous(blockIdx.x, gridDim.x)) * tiles_per_wg + wave / vpw;
    if(tile >= n_tiles)
        return;
    const int y = tile / tiles_x;
    const int x0 = (tile - y * tiles_x) * (32 * PXL);
    const int xl = x0 + r * PXL;
    const int W = a.width, H = a.height;
    const uint32_t *grid32 = reinterpret_cast<const uint32_t *>(a.grid);
    const size_t plane_px = (size_t)W * (size_t)H;

    float focus_px[PXL];
    if constexpr(ALLFOCUS)
    {
        const uint8_t *map_plane = a.maps + (size_t)a.map_index * plane_px * 4;
#pragma unroll
        for(int i = 0; i < PXL; i++)
            focus_px[i] = decode_focus(map_plane, W, H, xl + i, y, a.focus, a.range);
    }

    for(int pass = wave % vpw; pass < view_passes; pass += vpw)
    {
        const int vbase = pass * (32 * MT);
        f32x16 acc[MT][PXL][3];
#pragma unroll
        for(int m = 0; m < MT; m++)
#pragma unroll
            for(int i = 0; i < PXL; i++)
#pragma unroll
                for(int c = 0; c < 3; c++)
#pragma unroll
                    for(int e = 0; e < 16; e++)
                        acc[m][i][c][e] = 0.0f;

        for(int kb = 0; kb < a.k_pad; kb += 16)
        {
            // image of MFMA q for this half-wave: kb + 2q + h  (k = 0 → lower half-wave, k = 1 → upper: ascending g)
            PixelRun<PXL> px[8];
#pragma unroll
            for(int q = 0; q < 8; q++)
            {
                const int g = kb + 2 * q + h;
                const int gi = min(g, a.n_images - 1);
                const uint32_t *plane = grid32 + (size_t)gi * plane_px;
                if constexpr(ALLFOCUS)
                {
                    const lfi_float2 off = a.offsets[g];
#pragma unroll
                    for(int i = 0; i < PXL; i++)
                    {
                        int sx = clampi(warp_float(xl + i, focus_px[i], off.x), 0, W - 1);
                        int sy = clampi(warp_float(y, focus_px[i], off.y), 0, H - 1);
                        px[q].v[i] = plane[sy * W + sx];
                    }
                }
                else
                {
                    const lfi_int2 off = a.focused[g];
                    const int sy = clampi(y + off.y, 0, H - 1);
                    const int sx = xl + off.x;
                    const uint32_t *row = plane + sy * W;
                    if(sx >= 0 && sx + PXL <= W)
                        px[q].load(row + sx);
                    else
                    {
#pragma unroll
                        for(int i = 0; i < PXL; i++)
                            px[q].v[i] = row[clampi(sx + i, 0, W - 1)];
                    }
                }
            }
            // weights: 16 consecutive floats of this lane's view row; MFMA q takes element 2q + h
            float wq[MT][8];
#pragma unroll
            for(int m = 0; m < MT; m++)
            {
                const float *wrow = a.w32 + (size_t)(a.v0 + vbase + m * 32 + r) * a.k_pad + kb;
#pragma unroll
                for(int q4 = 0; q4 < 4; q4++)
                {
                    const float4 t = *reinterpret_cast<const float4 *>(wrow + 4 * q4);
                    wq[m][2 * q4 + 0] = h ? t.y : t.x;
                    wq[m][2 * q4 + 1] = h ? t.w : t.z;
                }
            }
#pragma unroll
            for(int q = 0; q < 8; q++)
#pragma unroll
                for(int i = 0; i < PXL; i++)
                {
                    const uint32_t p = px[q].v[i];
                    const float pc[3] = {static_cast<float>(p & 0xffu), static_cast<float>((p >> 8) & 0xffu),
                                         static_cast<float>((p >> 16) & 0xffu)};
#pragma unroll
                    for(int c = 0; c < 3; c++)
#pragma unroll
                        for(int m = 0; m < MT; m++)
                            acc[m][i][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[m][q], pc[c], acc[m][i][c], 0, 0, 0);
                }
        }

        const bool run_inside = (xl + PXL <= W) && (PXL == 1 || (W % PXL) == 0);
#pragma unroll
        for(int m = 0; m < MT; m++)
#pragma unroll
            for(int e = 0; e < 16; e++)
            {
                const int view = a.v0 + vbase + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if(view >= a.v1)
                    continue;
                uint32_t rgba[PXL];
#pragma unroll
                for(int i = 0; i < PXL; i++)
                    rgba[i] = quant_rn(acc[m][i][0][e]) | (quant_rn(acc[m][i][1][e]) << 8) |
                              (quant_rn(acc[m][i][2][e]) << 16) | 0xff000000u;
                uint32_t *out = reinterpret_cast<uint32_t *>(a.views) + (size_t)view * plane_px + (size_t)y * W + xl;
                if(run_inside)
                    store_run<PXL>(out, rgba);
                else
                {
#pragma unroll
                    for(int i = 0; i < PXL; i++)
                        if(xl + i < W)
                            out[i] = rgba[i];
                }
                if(a.prequant != nullptr && view == a.prequant_view)
                {
#pragma unroll
                    for(int i = 0; i < PXL; i++)
                        if(xl + i < W)
                        {
                            float *pq = a.prequant + ((size_t)y * W + xl + i) * 3;
                            pq[0] = acc[m][i][0][e];
                            pq[1] = acc[m][i][1][e];
                            pq[2] = acc[m][i][2][e];
                        }
                }
            }
    }
}

} // namespace lfi
