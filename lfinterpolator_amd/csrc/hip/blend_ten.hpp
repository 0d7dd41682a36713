// blend_ten.hpp — the TEN_WM path on CDNA4 matrix cores.
//
// Replaces Kernels::Tensors::process<allFocus> (reference src/kernels.cu:398-461, helpers :353-396).
// The reference contracts  pixels[32×16] · weights[16×8]  per warp with WMMA m32n8k16 and fp16 accumulators, staging the
// pixel tile through shared memory to transpose it.  Here the contraction is turned round so that no transpose exists:
//
//      out[view][pixel] (per colour channel)  =  Σ_image  W[view][image] · I_image[pixel + offset_image]
//
//   v_mfma_f32_32x32x16_f16:  A = weights  (lane l: view  l&31, images 8(l>>5)+j, j<8)
//                             B = pixels   (lane l: pixel l&31, images 8(l>>5)+j)
//                             D            (lane l: pixel l&31, views (r&3)+8(r>>2)+4(l>>5), r<16)
//
// A lane gathers its own pixel(s) from 8 images straight from HBM — consecutive lanes read consecutive pixels, so a
// half-wave reads one contiguous run per image — and for a fixed view a half-wave holds consecutive output pixels, so
// stores are contiguous runs too.  With PXL pixels per lane the lane's dwordx{PXL} load feeds PXL independent N-tiles
// (pixel = PXL*lane + i belongs to N-tile i), which keeps loads AND stores 4*PXL bytes wide without any shuffle.
//
// Numerics: a pixel byte b enters the MFMA as the fp16 SUBNORMAL with mantissa b (= b·2^-24): the
// byte→half conversion is a zero-extension (one v_perm_b32 per two values) and is exact; products with fp16 weights are
// exact in fp32; the MFMA accumulates in fp32; the epilogue scales by 2^24 (exact), rounds ONCE to fp16 (RN-even) and
// truncates to u8 with saturation like __half2uchar_rz (src/kernels.cu:393).  The reference's accumulator is fp16 and is
// re-rounded per 16-image batch; LFI_FLAG_TEN_ROUND_PER_BATCH reproduces that model EXACTLY, in double precision on the vector pipe
// (blend_ten_m16 below), for parity debugging.
#pragma once

#include "lfi_device.hpp"

namespace lfi {

// u8 (bytes of two RGBA dwords) → two fp16 subnormals packed in a dword: [15:0] = byte c of lo, [31:16] = byte c of hi
template <int C>
__device__ __forceinline__ uint32_t pack_subnormal_pair(uint32_t lo, uint32_t hi)
{
    // v_perm_b32 D = {S0,S1}: selector 0-3 → bytes of S1 (lo), 4-7 → bytes of S0 (hi), 0x0c → 0x00
    constexpr uint32_t sel = 0x0c000c00u | uint32_t(C) | (uint32_t(4 + C) << 16);
    return __builtin_amdgcn_perm(hi, lo, sel);
}

// truncating fp16 quantisation of one accumulator (already scaled back to 0..255 units)
__device__ __forceinline__ uint32_t quant_trunc_f16(float x, float &rounded)
{
    _Float16 h = static_cast<_Float16>(x); // v_cvt_f16_f32, RN-even
    float y = static_cast<float>(h);
    rounded = y;
    y = fminf(fmaxf(y, 0.0f), 255.0f); // saturate, NaN → 0
    return static_cast<uint32_t>(y);   // v_cvt_u32_f32 truncates
}

template <int PXL>
struct PixelRun;
template <>
struct PixelRun<1>
{
    uint32_t v[1];
    __device__ __forceinline__ void load(const uint32_t *p) { v[0] = *p; }
};
template <>
struct PixelRun<2>
{
    uint32_t v[2];
    __device__ __forceinline__ void load(const uint32_t *p)
    {
        u32x2_a4 t = *reinterpret_cast<const u32x2_a4 *>(p);
        v[0] = t.x;
        v[1] = t.y;
    }
};
template <>
struct PixelRun<4>
{
    uint32_t v[4];
    __device__ __forceinline__ void load(const uint32_t *p)
    {
        u32x4_a4 t = *reinterpret_cast<const u32x4_a4 *>(p);
        v[0] = t.x;
        v[1] = t.y;
        v[2] = t.z;
        v[3] = t.w;
    }
};

template <int PXL>
__device__ __forceinline__ void store_run(uint32_t *p, const uint32_t (&v)[PXL])
{
    if constexpr(PXL == 1)
        *p = v[0];
    else if constexpr(PXL == 2)
    {
        u32x2_a4 t = {v[0], v[1]};
        *reinterpret_cast<u32x2_a4 *>(p) = t;
    }
    else
    {
        u32x4_a4 t = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<u32x4_a4 *>(p) = t;
    }
}

// One wave = one run of 32*PXL pixels of one image row × 32*MT views per pass.
// A 256-thread workgroup holds 4 waves: VPW of them share a pixel run and take different view passes.
template <int PXL, int MT, bool ALLFOCUS>
__global__ void __launch_bounds__(256) blend_ten_direct(const KernelArgs a, const int tiles_x, const int n_tiles,
                                                        const int view_passes, const int vpw)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;

    const int tiles_per_wg = 4 / vpw;
    const int tile = int(xcd_contiguous(blockIdx.x, gridDim.x)) * tiles_per_wg + wave / vpw;
    if(tile >= n_tiles)
        return; // whole wave leaves together: no lane reaches an MFMA without its partners (SURVEY.md defect D8)
    const int y = tile / tiles_x;
    const int x0 = (tile - y * tiles_x) * (32 * PXL);
    const int xl = x0 + r * PXL; // this lane's first pixel
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
        const int vbase = pass * (32 * MT); // relative to v0; rows of the padded weight matrix start at v0 + vbase
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
            const int gk = kb + 8 * h; // this half-wave's first image of the k-step
            // ---- gather: 8 images × PXL pixels per lane -------------------------------------------------------------
            PixelRun<PXL> px[8];
#pragma unroll
            for(int j = 0; j < 8; j++)
            {
                const int g = gk + j;
                const int gi = min(g, a.n_images - 1); // padded images carry zero weights; any valid pixel will do
                const uint32_t *plane = grid32 + (size_t)gi * plane_px;
                if constexpr(ALLFOCUS)
                {
                    const lfi_float2 off = a.offsets[g];
#pragma unroll
                    for(int i = 0; i < PXL; i++)
                    {
                        int sx = clampi(warp_float(xl + i, focus_px[i], off.x), 0, W - 1);
                        int sy = clampi(warp_float(y, focus_px[i], off.y), 0, H - 1);
                        px[j].v[i] = plane[sy * W + sx];
                    }
                }
                else
                {
                    const lfi_int2 off = a.focused[g];
                    const int sy = clampi(y + off.y, 0, H - 1);
                    const int sx = xl + off.x;
                    const uint32_t *row = plane + sy * W;
                    if(sx >= 0 && sx + PXL <= W)
                        px[j].load(row + sx); // interior: one 4*PXL-byte load
                    else
                    {
#pragma unroll
                        for(int i = 0; i < PXL; i++)
                            px[j].v[i] = row[clampi(sx + i, 0, W - 1)]; // clamp-to-edge (src/kernels.cu:125)
                    }
                }
            }
            // ---- weights: A fragment = 8 consecutive halves of one row of the padded matrix ---------------------------
            half8 wfrag[MT];
#pragma unroll
            for(int m = 0; m < MT; m++)
            {
                const uint16_t *wrow = a.w16 + (size_t)(a.v0 + vbase + m * 32 + r) * a.k_pad + gk;
                u32x4 t = *reinterpret_cast<const u32x4 *>(wrow);
                wfrag[m] = __builtin_bit_cast(half8, t);
            }
            // ---- contraction ----------------------------------------------------------------------------------------------
#pragma unroll
            for(int i = 0; i < PXL; i++)
            {
                u32x4 bc[3];
#pragma unroll
                for(int q = 0; q < 4; q++)
                {
                    bc[0][q] = pack_subnormal_pair<0>(px[2 * q].v[i], px[2 * q + 1].v[i]);
                    bc[1][q] = pack_subnormal_pair<1>(px[2 * q].v[i], px[2 * q + 1].v[i]);
                    bc[2][q] = pack_subnormal_pair<2>(px[2 * q].v[i], px[2 * q + 1].v[i]);
                }
#pragma unroll
                for(int c = 0; c < 3; c++)
                {
                    const half8 b = __builtin_bit_cast(half8, bc[c]);
#pragma unroll
                    for(int m = 0; m < MT; m++)
                        acc[m][i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfrag[m], b, acc[m][i][c], 0, 0, 0);
                }
            }
        }

        // ---- epilogue: scale, round to fp16 once, truncate, pack RGBA, store PXL pixels per (lane, view) -----------------------
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
                float pre[PXL][3];
#pragma unroll
                for(int i = 0; i < PXL; i++)
                {
                    uint32_t q0 = quant_trunc_f16(acc[m][i][0][e] * 16777216.0f, pre[i][0]);
                    uint32_t q1 = quant_trunc_f16(acc[m][i][1][e] * 16777216.0f, pre[i][1]);
                    uint32_t q2 = quant_trunc_f16(acc[m][i][2][e] * 16777216.0f, pre[i][2]);
                    rgba[i] = q0 | (q1 << 8) | (q2 << 16) | 0xff000000u;
                }
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
                            pq[0] = pre[i][0];
                            pq[1] = pre[i][1];
                            pq[2] = pre[i][2];
                        }
                }
            }
    }
}

// ---- LFI_FLAG_TEN_ROUND_PER_BATCH: the reference's half-accumulator model, EXACTLY (round 5) ------------------------------------------
// wmma::mma_sync with a half accumulator fragment (src/kernels.cu:418-447) re-rounds the running sum to fp16 after every batch of 16
// images; the oracle's model M16 is acc ← RN_fp16(acc + Σ_16 w·p) with the inner sum EXACT.  Rounds 1–4 reproduced it on the matrix pipe
// (an fp32 MFMA accumulation of the 16 products, then a rounding to fp16): the pipe's own fp32 roundings and the second rounding made
// ≈ 1e-4 of the bytes differ.  A matrix core cannot give the exact sum (its accumulator keeps 24 bits, the products of a batch span up to 43),
// so the debug mode computes it where it CAN be exact — in double precision on the vector pipe (products of an fp16 weight and a byte and
// sums of 16 of them, 48 bits at most, are exact in fp64) — and rounds once, to nearest even, with the oracle's own algorithm.  One pixel
// per lane, 8 views per pass; a debug mode, not a fast one.
__device__ __forceinline__ uint32_t f64_to_f16_rne(const double x) // finite x (the oracle's lfo_f64_to_f16, oracle/lfi_oracle.c)
{
    const uint32_t sign = x < 0.0 ? 0x8000u : 0u;
    const double a = __builtin_fabs(x);
    if(a >= 65520.0)
        return sign | 0x7c00u;
    if(a < 0x1p-14)                                           // subnormal result (or the smallest normal, by a carry): multiples of 2^-24
        return sign | uint32_t(__builtin_rint(a * 0x1p24));
    int e = int((__builtin_bit_cast(uint64_t, a) >> 52) & 0x7ffu) - 1023;
    double q = __builtin_rint(__builtin_ldexp(a, 10 - e));   // in [1024, 2048], exact scaling, ONE rounding (v_rndne_f64: to even)
    if(q >= 2048.0)
    {
        q = 1024.0;
        e++;
    }
    return sign | (uint32_t(e + 15) << 10) | (uint32_t(q) - 1024u);
}

template <bool ALLFOCUS>
__global__ void __launch_bounds__(256) blend_ten_m16(const KernelArgs a)
{
    constexpr int VP = 8;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int W = a.width, H = a.height;
    if(x >= W || y >= H)
        return;
    const uint32_t *grid32 = reinterpret_cast<const uint32_t *>(a.grid);
    const size_t plane_px = (size_t)W * (size_t)H;
    float focus_px = 0.0f;
    if constexpr(ALLFOCUS)
        focus_px = decode_focus(a.maps + (size_t)a.map_index * plane_px * 4, W, H, x, y, a.focus, a.range);
    for(int vb = a.v0; vb < a.v1; vb += VP)
    {
        uint32_t acc[VP][3]; // fp16 bit patterns
#pragma unroll
        for(int v = 0; v < VP; v++)
            acc[v][0] = acc[v][1] = acc[v][2] = 0u;
        for(int kb = 0; kb < a.k_pad; kb += 16)
        {
            double s[VP][3];
#pragma unroll
            for(int v = 0; v < VP; v++)
                s[v][0] = s[v][1] = s[v][2] = 0.0;
            for(int k = 0; k < 16; k++)
            {
                const int g = kb + k, gi = min(g, a.n_images - 1); // padded images carry zero weights; any valid pixel will do
                int sx, sy;
                if constexpr(ALLFOCUS)
                {
                    const lfi_float2 off = a.offsets[gi];
                    sx = clampi(warp_float(x, focus_px, off.x), 0, W - 1);
                    sy = clampi(warp_float(y, focus_px, off.y), 0, H - 1);
                }
                else
                {
                    const lfi_int2 off = a.focused[gi];
                    sx = clampi(x + off.x, 0, W - 1);
                    sy = clampi(y + off.y, 0, H - 1);
                }
                const uint32_t px = grid32[(size_t)gi * plane_px + (size_t)sy * W + sx];
                const double p0 = double(px & 255u), p1 = double((px >> 8) & 255u), p2 = double((px >> 16) & 255u);
#pragma unroll
                for(int v = 0; v < VP; v++)
                {
                    const double wv = double(__builtin_bit_cast(_Float16, a.w16[(size_t)(vb + v) * a.k_pad + g])); // rows up to v_pad exist, zero-filled
                    s[v][0] = __builtin_fma(wv, p0, s[v][0]); // exact: the products and their partial sums fit 53 bits
                    s[v][1] = __builtin_fma(wv, p1, s[v][1]);
                    s[v][2] = __builtin_fma(wv, p2, s[v][2]);
                }
            }
#pragma unroll
            for(int v = 0; v < VP; v++)
#pragma unroll
                for(int c = 0; c < 3; c++)
                    acc[v][c] = f64_to_f16_rne(double(__builtin_bit_cast(_Float16, uint16_t(acc[v][c]))) + s[v][c]);
        }
#pragma unroll
        for(int v = 0; v < VP; v++)
        {
            const int view = vb + v;
            if(view >= a.v1)
                break;
            uint32_t rgba = 0xff000000u;
            float pre[3];
#pragma unroll
            for(int c = 0; c < 3; c++)
            {
                pre[c] = float(__builtin_bit_cast(_Float16, uint16_t(acc[v][c])));
                rgba |= uint32_t(fminf(fmaxf(pre[c], 0.0f), 255.0f)) << (8 * c); // __half2uchar_rz: truncate, saturate
            }
            reinterpret_cast<uint32_t *>(a.views)[(size_t)view * plane_px + (size_t)y * W + x] = rgba;
            if(a.prequant != nullptr && view == a.prequant_view)
            {
                float *pq = a.prequant + ((size_t)y * W + x) * 3;
                pq[0] = pre[0], pq[1] = pre[1], pq[2] = pre[2];
            }
        }
    }
}

// hardware probe used by lfi_debug_mfma_f16: one v_mfma_f32_32x32x16_f16 on row-major A[32][16], B[16][32]
__global__ void __launch_bounds__(64) probe_mfma_f16(const uint16_t *__restrict__ a, const uint16_t *__restrict__ b,
                                                    float *__restrict__ c)
{
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    half8 fa, fb;
#pragma unroll
    for(int j = 0; j < 8; j++)
    {
        fa[j] = __builtin_bit_cast(_Float16, a[r * 16 + 8 * h + j]);
        fb[j] = __builtin_bit_cast(_Float16, b[(8 * h + j) * 32 + r]);
    }
    f32x16 d;
#pragma unroll
    for(int e = 0; e < 16; e++)
        d[e] = 0.0f;
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, d, 0, 0, 0);
#pragma unroll
    for(int e = 0; e < 16; e++)
        c[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = d[e];
}

// hardware probe used by lfi_debug_mfma_f16_chain: C[32][32] = A[32][K] · B[K][32] (row-major fp16 bit patterns, K a multiple of
// 32) accumulated exactly as the kernels accumulate — SHAPE 0: K/16 chained v_mfma_f32_32x32x16_f16 (blend_planar, blend_persist),
// SHAPE 1: K/32 chained v_mfma_f32_16x16x32_f16 per 16×16 quadrant (blend_p3); the accumulator of one instruction is the C operand
// of the next.  Measures what the matrix pipe's fp32 accumulation does to sums whose addends differ by many binades.
template <int SHAPE>
__global__ void __launch_bounds__(64) probe_mfma_f16_chain(const uint16_t *__restrict__ a, const uint16_t *__restrict__ b, const int K,
                                                          float *__restrict__ c)
{
    const int lane = threadIdx.x & 63;
    if constexpr(SHAPE == 0)
    {
        const int r = lane & 31, h = lane >> 5;
        f32x16 d;
#pragma unroll
        for(int e = 0; e < 16; e++)
            d[e] = 0.0f;
        for(int k0 = 0; k0 < K; k0 += 16)
        {
            half8 fa, fb;
#pragma unroll
            for(int j = 0; j < 8; j++)
            {
                fa[j] = __builtin_bit_cast(_Float16, a[r * K + k0 + 8 * h + j]);
                fb[j] = __builtin_bit_cast(_Float16, b[(k0 + 8 * h + j) * 32 + r]);
            }
            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, d, 0, 0, 0);
        }
#pragma unroll
        for(int e = 0; e < 16; e++)
            c[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = d[e];
    }
    else
    {
        typedef float f32x4_t __attribute__((ext_vector_type(4)));
        const int n = lane & 15, kg = lane >> 4;
        for(int qi = 0; qi < 2; qi++)
            for(int qj = 0; qj < 2; qj++)
            {
                f32x4_t d = {0.0f, 0.0f, 0.0f, 0.0f};
                for(int k0 = 0; k0 < K; k0 += 32)
                {
                    half8 fa, fb;
#pragma unroll
                    for(int j = 0; j < 8; j++)
                    {
                        fa[j] = __builtin_bit_cast(_Float16, a[(16 * qi + n) * K + k0 + 8 * kg + j]);
                        fb[j] = __builtin_bit_cast(_Float16, b[(k0 + 8 * kg + j) * 32 + 16 * qj + n]);
                    }
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, d, 0, 0, 0);
                }
#pragma unroll
                for(int i = 0; i < 4; i++)
                    c[(16 * qi + 4 * kg + i) * 32 + 16 * qj + n] = d[i];
            }
    }
}

} // namespace lfi
