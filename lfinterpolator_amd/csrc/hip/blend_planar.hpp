// blend_planar.hpp — TEN_WM from a planar copy of the inputs: 3 bytes per pixel read instead of 4.
//
// blend_persist / blend_wave move 4·W·H·(N + V) bytes per launch and sit within ≈10 % of what the memory system gives that
// pattern (profiles/r01_notes.md §3).  A quarter of the bytes READ is the alpha channel, which neither method uses
// (reference src/kernels.cu:292-299, 353-370 read .x .y .z only).  An RGB-interleaved layout cannot be fetched by LDS-DMA at
// arbitrary pixel shifts (sources must be dword aligned), a planar one can if every (image, channel) plane is kept in four
// copies shifted by 0…3 bytes: a run that starts at pixel x0 is then a dword-aligned run of copy (x0 + pad) mod 4.  The copies
// are padded left and right by more than the largest offset with replicated edge pixels, so clamp-to-edge in x
// (cudaBoundaryModeClamp, src/kernels.cu:125) needs no per-pixel path at all; in y the row index is clamped per image.
// The memory system gives this pattern (192 streams of 32-byte runs per wave instead of 64 of 128 bytes) the full saving:
// gather + scatter 164–167 µs instead of 183–185 µs at config 2 (tools/ablate.hip, "planar").
//
// planar_build — once per change of the inputs (the context tracks them, include/lfi.h lfi_grid_modified): 12 byte planes per
//                image from its RGBA plane.
// blend_planar — the workgroup pipeline of blend_persist (one K-chunk ≤ 64 images, one view pass ≤ 64 views, fixed focus, whole
//                image; weights loaded once) with 24 LDS-DMA pieces of 8 runs × 128 bytes per 128-pixel tile, 6 per wave (a run
//                per wave-sized tile would be 32 bytes: four times the cache-line requests per byte — measured slower than
//                the RGBA kernel); LDS bytes [channel][image][128 pixels]; the MFMA B operand is assembled from byte reads
//                (a pixel byte IS the fp16 subnormal's mantissa).
// Arithmetic, weights, operand maps and epilogue are those of blend_persist / blend_wave (blend_core.hpp): identical output bytes.
// Replaces Kernels::Tensors::process<false> (reference src/kernels.cu:398-461).
#pragma once

#include "blend_wave.hpp"

namespace lfi {

// grid (ceil(pitch/1024), H, N): a thread writes one dword (4 consecutive bytes) of all 12 planes of its image row
__global__ void __launch_bounds__(256) planar_build(const uint8_t *__restrict__ grid, uint8_t *__restrict__ planar, const int W, const int H,
                                                    const int pitch, const int padx)
{
    const int j4 = (blockIdx.x * 256 + threadIdx.x) * 4; // first byte column of this thread's dword
    if(j4 >= pitch)
        return;
    const int y = blockIdx.y, g = blockIdx.z;
    const uint32_t *row = reinterpret_cast<const uint32_t *>(grid) + ((size_t)g * H + y) * W;
    // bytes j4 … j4+3 of shift copy k hold pixels j4 + k − padx … : seven consecutive pixels cover the four copies
    uint32_t px[7];
#pragma unroll
    for(int i = 0; i < 7; i++)
        px[i] = row[clampi(j4 - padx + i, 0, W - 1)];
#pragma unroll
    for(int c = 0; c < 3; c++)
#pragma unroll
        for(int k = 0; k < 4; k++)
        {
            uint32_t v = 0;
#pragma unroll
            for(int b = 0; b < 4; b++)
                v |= ((px[k + b] >> (8 * c)) & 0xffu) << (8 * b);
            uint8_t *plane = planar + ((((size_t)g * 3 + c) * 4 + k) * H + y) * pitch;
            *reinterpret_cast<uint32_t *>(plane + j4) = v;
        }
}

template <int MT, bool NT_STORE>
__global__ void __launch_bounds__(256, 2) blend_planar(const KernelArgs a, const int tiles_x, const int n_tiles)
{
    constexpr int KC = 64, KS = KC / 16, VPP = MT * 32, TPX = 128;
    constexpr int W_DW = (KC / 8) * VPP * 4; // fp16 weight fragments [k-octet][view] × 16 B
    constexpr int PX_B = 3 * KC * TPX;       // bytes of one pixel buffer: [channel][image][128 pixels] = 24 KB
    constexpr int PIECES = PX_B / 1024;      // 24 LDS-DMA pieces per tile, 6 per wave
    __shared__ __attribute__((aligned(16))) uint32_t lds[W_DW + 2 * (PX_B / 4)];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int W = a.width, H = a.height;
    const size_t oplane_px = (size_t)W * (size_t)a.out_rows;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane(uint32_t(uintptr_t((lds_ptr_t)lds)));
    const uint32_t px_base = lds_base + W_DW * 4;
    const int kc = a.k_pad; // ≤ KC (host)

    // ---- once per workgroup: the weight fragments ------------------------------------------------------------------------------
    for(int o = wave; 8 * o < kc; o += 4)
        if(lane < VPP)
            dma16(a.w16s + (size_t)(a.v0 + lane) * a.k_pad + 8 * o, lds_base + uint32_t(o) * (VPP * 16));

    // ---- per lane, once: piece p = wave + 4 j (j = 0…5) moves the runs 8p … 8p+7 of the tile — run ρ = channel·64 + image, 128
    // bytes = 128 pixels, eight lanes × 16 bytes — so this lane serves run 8p + (lane >> 3): its image's offsets and the byte
    // offset of its (image, channel) plane group (shift 0)
    constexpr int PPW = PIECES / 4;
    int ox[PPW], oy[PPW];
    size_t plane0[PPW];
    bool live[PPW];
    const size_t shift_stride = (size_t)H * a.planar_pitch; // one byte plane
#pragma unroll
    for(int j = 0; j < PPW; j++)
    {
        const int run = 8 * (wave + 4 * j) + (lane >> 3), c = run / KC, gq = run % KC;
        live[j] = gq < kc; // chunks shorter than 64 images: those runs are never read
        const int g = min(gq, a.n_images - 1); // padded images (zero weights) re-read the last one
        const lfi_int2 o = a.focused[g];
        ox[j] = o.x;
        oy[j] = o.y;
        plane0[j] = ((size_t)g * 3 + c) * 4 * shift_stride;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier(); // weights in LDS
    asm volatile("" ::: "memory");

    auto issue = [&](const int t, const int buf) {
        const int ty = t / tiles_x;
        const int y = a.out_y0 + ty;
        const int x0 = (t - ty * tiles_x) * TPX;
        const uint32_t dst = px_base + uint32_t(buf) * PX_B;
#pragma unroll
        for(int j = 0; j < PPW; j++)
        {
            // the run of this lane's image starts at pixel x0 + ox: dword aligned in shift copy (x0 + ox + padx) & 3
            const int sy = clampi(y + oy[j], 0, H - 1);
            const int start = x0 + ox[j] + a.planar_padx; // ≥ 0: the padding exceeds every offset
            const int k = start & 3;
            const uint8_t *src = a.planar + plane0[j] + ((size_t)k * H + sy) * a.planar_pitch + (start - k) + 16 * (lane & 7);
            // whole pieces are skipped only (kc is a multiple of 16, a piece is 8 images of one channel)
            if(__builtin_amdgcn_ballot_w64(live[j]) != 0ull)
                dma16(src, dst + uint32_t(wave + 4 * j) * 1024u);
        }
    };

    // ---- tiles j, j+G, … (as blend_persist): wait own pieces → barrier → issue next → compute → epilogue --------------------------
    const int G = gridDim.x;
    int t = int(xcd_contiguous(blockIdx.x, gridDim.x));
    if(t >= n_tiles)
        return;
    int buf = 0;
    int prev_stores = 0;
    issue(t, 0);

    f32x16 acc[MT][3];
    f32x16 zero16;
#pragma unroll
    for(int e = 0; e < 16; e++)
        zero16[e] = 0.0f;
    const u32x4 *w_buf = reinterpret_cast<const u32x4 *>(lds);
    const uint8_t *px_bytes = reinterpret_cast<const uint8_t *>(lds + W_DW) + wave * 32 + r + 8 * h * TPX;
    while(true)
    {
        const int nt = t + G;
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
        __builtin_amdgcn_s_barrier(); // everybody's pieces have landed; everybody is done with the buffer about to be refilled
        asm volatile("" ::: "memory");
        if(nt < n_tiles)
            issue(nt, buf ^ 1);

        // ---- k-loop: as unit_ten (blend_core.hpp), the B operand assembled from bytes: image g of channel c at byte
        // (c·64 + g)·128 + pixel of the buffer; this lane's pixel 32·wave + r, images 16 ks + 8 h + j
        const uint8_t *col = px_bytes + buf * PX_B;
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
                        bc[c][q] = lo | (hi << 16); // two fp16 subnormals b·2^-24
                    }
#pragma unroll
                for(int c = 0; c < 3; c++)
                {
                    const half8 bfrag = __builtin_bit_cast(half8, bc[c]);
#pragma unroll
                    for(int m = 0; m < MT; m++)
                        acc[m][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfrag[m], bfrag, ks == 0 ? zero16 : acc[m][c], 0, 0, 0);
                }
            }
        }

        {
            const int ty = t / tiles_x;
            prev_stores = store_tile<false, MT, NT_STORE, false>(a, acc, a.v0, ty, (t - ty * tiles_x) * TPX + wave * 32, r, h, oplane_px);
        }
        if(nt >= n_tiles)
            break;
        t = nt;
        buf ^= 1;
    }
}

} // namespace lfi
