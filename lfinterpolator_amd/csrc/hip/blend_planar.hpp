// blend_planar.hpp — TEN_WM from a planar copy of the inputs: 3 bytes per pixel read instead of 4.
//
// blend_persist / blend_wave move 4·W·H·(N + V) bytes per launch and sit within ≈10 % of what the memory system gives that
// pattern (profiles/r01_notes.md §3).  A quarter of the bytes READ is the alpha channel, which neither method uses
// (reference src/kernels.cu:292-299, 353-370 read .x .y .z only).  The derived copy keeps every (image, channel) as ONE byte plane,
// padded left and right by more than the largest offset with replicated edge pixels, so clamp-to-edge in x
// (cudaBoundaryModeClamp, src/kernels.cu:125) needs no per-pixel path at all; in y the row index is clamped per image.  A tile's
// 128-pixel run of an image starts at an arbitrary BYTE of its plane row; LDS-DMA fetches it from there (round 3,
// tools/probe_ldsdma_bytes.hip: byte-aligned sources are legal, land lane-linear and cost what a dword-aligned run that straddles
// two cache lines costs — which every shifted run does; rounds 1–2 assumed dword alignment and kept FOUR byte-shifted copies of
// every plane, 12 B per pixel·image instead of 3).
// The memory system gives this pattern (192 byte-plane streams instead of 64 RGBA-plane streams) the full saving: gather +
// scatter with no arithmetic 164–167 µs instead of 183–185 µs at config 2 (tools/ablate.hip, "planar").
//
// planar_build — once per change of the inputs (the context tracks them, include/lfi.h lfi_grid_modified): 3 byte planes per
//                image from its RGBA plane.
// blend_planar — the workgroup pipeline of blend_persist (fixed focus; any number of images and views; row windows): 24 LDS-DMA
//                pieces of 8 runs × 128 bytes per unit, 6 per wave (a run per wave-sized tile would be 32 bytes: four times the
//                cache-line requests per byte — measured slower than the RGBA kernel); LDS bytes [channel][image][128 pixels];
//                the MFMA B operand is assembled from byte reads (a pixel byte IS the fp16 subnormal's mantissa).
// Arithmetic, weights, operand maps and epilogue are those of blend_persist / blend_wave (blend_core.hpp): identical output bytes.
//
// STDF = true — the STD method (exact fp32 fmaf chain, src/kernels.cu:292-310) at TEN_WM speed for launches of one chunk:
// the chain's result differs from the exact sum S by at most N·2^-16 (N ≤ 64 roundings of half an ulp of a value below 512 —
// the host checks that every view's weights are non-negative and sum to at most 2), and so does the fp16-MFMA accumulation of
// the same exactly-representable products by at most N·2^-15 even if every one of its additions truncated.  Wherever the MFMA
// sum is farther than std_band (≥ the two bounds together, plus a margin) from every half-integer, rounding it gives the
// byte the chain gives; the few per cent of (pixel, view, channel) sums inside the band are queued in LDS and recomputed with
// the chain itself (v_fma_f32, images ascending) from the pixel bytes and weights still resident in LDS, 64 at a time, and
// patched with byte stores behind the tile's dword stores.  Bit-exact against the oracle like blend_wave<STD> (same tests).
// Replaces Kernels::Tensors::process<false> (reference src/kernels.cu:398-461).
#pragma once

#include "blend_wave.hpp"

namespace lfi {

// grid (ceil(pitch/1024), H, N): a thread writes one dword (4 consecutive bytes) of the 3 planes of its image row
// H = rows HELD per plane (the whole image, or the input rows of a row window)
// g0: the first image of the range this launch converts (gridDim.z images: all of them, or the ones that changed)
__global__ void __launch_bounds__(256) planar_build(const uint8_t *__restrict__ grid, uint8_t *__restrict__ planar, const int W, const int H,
                                                    const int pitch, const int padx, const int32_t *__restrict__ phase, const int g0)
{
    const int j4 = (blockIdx.x * 256 + threadIdx.x) * 4; // first byte column of this thread's dword
    if(j4 >= pitch)
        return;
    const int y = blockIdx.y, g = g0 + blockIdx.z;
    const uint32_t *row = reinterpret_cast<const uint32_t *>(grid) + ((size_t)g * H + y) * W;
    const int first = padx + phase[g]; // byte of pixel 0
    uint32_t px[4]; // byte j of a plane row holds pixel j − first, edges replicated
#pragma unroll
    for(int i = 0; i < 4; i++)
        px[i] = row[clampi(j4 - first + i, 0, W - 1)];
#pragma unroll
    for(int c = 0; c < 3; c++)
    {
        const uint32_t lo = __builtin_amdgcn_perm(px[1], px[0], 0x0c0c0400u + 0x0101u * uint32_t(c)); // [p0.c, p1.c, 0, 0]
        const uint32_t hi = __builtin_amdgcn_perm(px[3], px[2], 0x0c0c0400u + 0x0101u * uint32_t(c));
        uint8_t *plane = planar + (((size_t)g * 3 + c) * H + y) * pitch;
        *reinterpret_cast<uint32_t *>(plane + j4) = lo | (hi << 16);
    }
}

// STDF epilogue of one wave's 32 pixels × (MT·32 views from vbase): acc holds S·2^-9 (fp16 MFMA of pixel subnormals and ×2^15
// weights).  Stores round(S) for every valid output like store_tile<STD>, queues the (lane, m, channel, e) whose S lies within
// a.std_band of a half-integer, and recomputes those with the reference's fmaf chain from LDS (px = this wave's pixel bytes
// [channel][image][TPX], wh = the weight fragments as halves [k-octet][view][8], ×2^15), patching single bytes.  Returns the
// number of store instructions issued; clears acc.
template <int MT, bool NT_STORE, int KC, int TPX>
__device__ __forceinline__ int store_tile_filtered(const KernelArgs &a, f32x16 (&acc)[MT][3], const int vbase, const int y, const int xw,
                                                   const int lane, const size_t oplane_px, const uint8_t *px, const uint16_t *wh,
                                                   const int kc, uint16_t *queue)
{
    constexpr int VPP = MT * 32;
    const int W = a.width, r = lane & 31, h = lane >> 5;
    int n_st = 0, count = 0; // count: queued entries (wave-uniform)
    uint8_t *plane0 = a.views + ((size_t)vbase * oplane_px + (size_t)y * W + xw) * 4;
    const uint32_t lane_off = (uint32_t(r) + uint32_t(4 * h) * uint32_t(oplane_px)) * 4u; // < 2^32: four planes of ≤ 2^26 pixels
    uint64_t stride = (uint64_t)oplane_px * 4;                                            // bytes between the planes of consecutive views
    asm volatile("" : "+s"(stride));
    auto patch = [&](const int n) { // recompute the first n ≤ 64 queued sums exactly, one per lane
        const uint32_t code = queue[lane < n ? lane : 0];
        const int src = code & 63, e = (code >> 6) & 15, c = (code >> 10) & 3, m = (code >> 12) & 1;
        const int vrel = m * 32 + (e & 3) + 8 * (e >> 2) + 4 * (src >> 5), pxl = src & 31;
        const uint8_t *pb = px + (size_t)(c * KC) * TPX + pxl;
        const u32x4 *wb = reinterpret_cast<const u32x4 *>(wh) + vrel; // this view's eight ×2^15 halves of a k-octet
        // the chain on the ×2^15 weights and the bytes AS fp16 subnormals (byte·2^-24: v_fma_mix_f32 widens both halves itself, no
        // v_cvt_f32_ubyte per image): a power-of-two scaling commutes with every rounding (nothing leaves the normal range: the
        // smallest non-zero product is 2^-48, sums stay below 2^0), so s15 = 2^-9 · (the reference's running sum) exactly
        float s15 = 0.0f;
        for(int o = 0; 8 * o < kc; o++)
        {
            const u32x4 w8 = wb[o * VPP];
#pragma unroll
            for(int j = 0; j < 8; j++)
            {
                const uint32_t pair = w8[j >> 1];
                const float w = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(j & 1 ? pair >> 16 : pair)));
                const float pf = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(pb[(8 * o + j) * TPX])));
                s15 = __builtin_fmaf(pf, w, s15); // addWeighted, src/kernels.cu:292-299
            }
        }
        const uint32_t byte = __builtin_bit_cast(uint32_t, s15 * 0x1p9f + 8388608.0f) & 0xffu; // (unsigned char)__float2int_rn(sum)
        if(lane < n)
            plane0[((size_t)vrel * oplane_px + pxl) * 4 + c] = static_cast<uint8_t>(byte);
        n_st++;
    };
    // acc = S·2^-9: adding 2^14 rounds it (RN-even) to a multiple of 2^-9, i.e. S to an integer, left in the low mantissa bits
    const float inside = (0.5f - a.std_band) * 0x1p-9f;
    static_assert(MT <= 2, "the per-lane flag words hold 2 × 16 accumulator registers per channel");
    uint32_t flagged[3] = {0u, 0u, 0u}; // per lane and channel: bit m·16 + e ↔ the sum in acc[m][c][e] needs the chain
    uint32_t valid_mask = 0u;           // bit m·16 + e ↔ this lane stores that sum
#pragma unroll
    for(int m = 0; m < MT; m++)
    {
        const int view_m = vbase + m * 32;
        const int nvalid = min(a.v1 - view_m, 32);
        if(nvalid > 0 && xw < W)
        {
            const bool lane_x_ok = xw + r < W;
            // which of the 16 sums of this m the lane stores: rows (e & 3) + 8·(e >> 2) + 4h below nvalid — a prefix of the e sequence
            const int lim = max(nvalid - 4 * h, 0);
            const int n_e = 4 * (lim >> 3) + min(lim & 7, 4);
            valid_mask |= lane_x_ok ? ((1u << n_e) - 1u) << (16 * m) : 0u;
#pragma unroll
            for(int e0 = 0; e0 < 16; e0 += 2)
            {
                // two sums per channel at a time: accumulator registers e0, e0 + 1 are an aligned pair, so t = acc + 2^14 and the
                // distance d = acc − (t − 2^14) are three v_pk_add_f32 per pair (paired across channels by the compiler they needed moves)
                float2_t d[3];
                u32x2 tb[3]; // the bits of t.  NOT `bit_cast<uint32_t>(t[j])`: where only some bits of a float element's bit pattern are used
                             // (here the low byte) hipcc 7.2 reads element 0 for every element — reproduced in ten lines (`a.y & 0xff` of a
                             // loaded float2 loads one dword), caught by the one-hot STD test.  Casting the whole vector is compiled correctly.
#pragma unroll
                for(int c = 0; c < 3; c++)
                {
                    const float2_t a2 = {acc[m][c][e0], acc[m][c][e0 + 1]};
                    const float2_t t = a2 + 16384.0f;
                    d[c] = a2 - (t - 16384.0f);
                    tb[c] = __builtin_bit_cast(u32x2, t);
                }
#pragma unroll
                for(int j = 0; j < 2; j++)
                {
                    const int e = e0 + j;
                    const int vrow = (e & 3) + 8 * (e >> 2); // + 4h per half-wave
                    if(vrow >= nvalid) // wave-uniform
                        continue;
                    const bool valid = lane_x_ok && vrow + 4 * h < nvalid;
                    uint32_t bits[3];
#pragma unroll
                    for(int c = 0; c < 3; c++)
                    {
                        bits[c] = tb[c][j];
                        // branch-free: a select between two constants and an OR per sum, the lane's validity applied once at the end
                        // (written as `if(valid && …) flagged |= …` the compiler emitted an exec-mask branch per sum)
                        flagged[c] |= __builtin_fabsf(d[c][j]) > inside ? 1u << (m * 16 + e) : 0u;
                    }
                    const uint32_t rg = __builtin_amdgcn_perm(bits[1], bits[0], 0x0c0c0400u); // [R, G, 0, 0]
                    const uint32_t rgba = __builtin_amdgcn_perm(bits[2], rg, 0x0d040100u);  // [R, G, B, 0xff]
                    // Address = a wave-uniform 64-bit row base in SGPRs (scalar arithmetic) + ONE 32-bit per-lane offset, the store written
                    // with its scalar-base form.  The epilogue is VALU-bound; left to the compiler every store cost a 64-bit vector add, and
                    // the 32 row offsets — invariant over the kernel — were hoisted: as per-lane values 64 VGPRs, as scalars spilled to VGPR
                    // lanes (two v_readlane per store).  `stride` is opaque per tile so that they are recomputed by the scalar unit instead.
                    const uint8_t *row = plane0 + (uint64_t)(m * 32 + vrow) * stride;
                    n_st++;
                    if(valid)
                    {
                        if constexpr(NT_STORE)
                            asm volatile("global_store_dword %0, %1, %2 nt" ::"v"(lane_off), "v"(rgba), "s"(row) : "memory");
                        else
                            asm volatile("global_store_dword %0, %1, %2" ::"v"(lane_off), "v"(rgba), "s"(row) : "memory");
                    }
                }
            }
        }
#pragma unroll
        for(int c = 0; c < 3; c++)
#pragma unroll
            for(int e = 0; e < 16; e++)
                acc[m][c][e] = 0.0f;
    }
#pragma unroll
    for(int c = 0; c < 3; c++)
        flagged[c] &= valid_mask;
    // compact the flagged sums of the wave into the queue, one per lane and round, and recompute them 64 at a time
    while(true)
    {
        const bool any = (flagged[0] | flagged[1] | flagged[2]) != 0u;
        const uint64_t mask = __builtin_amdgcn_ballot_w64(any);
        if(mask == 0ull)
            break;
        if(any)
        {
            const int c = flagged[0] ? 0 : (flagged[1] ? 1 : 2);
            const uint32_t word = c == 0 ? flagged[0] : (c == 1 ? flagged[1] : flagged[2]);
            const int b = __builtin_ctz(word);
            const uint32_t rest = word & (word - 1u);
            flagged[0] = c == 0 ? rest : flagged[0];
            flagged[1] = c == 1 ? rest : flagged[1];
            flagged[2] = c == 2 ? rest : flagged[2];
            const int ahead = __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
            queue[count + ahead] = static_cast<uint16_t>(lane | ((b & 15) << 6) | (c << 10) | ((b >> 4) << 12));
        }
        count += __builtin_popcountll(mask);
        if(count >= 64)
        {
            patch(64);
            const uint16_t tail = queue[64 + (lane < count - 64 ? lane : 0)];
            if(lane < count - 64)
                queue[lane] = tail;
            count -= 64;
        }
    }
    if(count > 0)
        patch(count);
    return n_st;
}

// Units as in blend_persist: (tile, view pass, chunk of ≤ 64 images); the pixel bytes of a tile stay in LDS for every view
// pass when the image stack fits one chunk; the weight fragments are fetched per unit — or once per workgroup when there is a
// single chunk and a single pass (config 2, every bench step).
template <int MT, bool NT_STORE, bool STDF = false>
__global__ void __launch_bounds__(256, 2) blend_planar(const KernelArgs a, const int tiles_x, const int n_tiles, const int view_passes, const int ring3,
                                                       const int reverse)
{
    constexpr int KC = 64, VPP = MT * 32, TPX = 128;
    constexpr int W_DW = (KC / 8) * VPP * 4; // fp16 weight fragments [k-octet][view] × 16 B
    constexpr int PX_B = 3 * KC * TPX;       // bytes of one pixel buffer: [channel][image of the chunk][128 pixels] = 24 KB
    constexpr int OFF_DW = 2 * LFI_MAX_IMAGES; // the integer offsets of every image (per-lane lookups by ds_read, not vector loads)
    // LDS (80 KB, two workgroups per CU): pixel buffers 0 and 1, weight buffer 0, and a third 24 KB region that is pixel buffer 2
    // when the launch is one chunk and one view pass (three-deep ring, weights loaded once) and holds weight buffer 1 and the
    // offset table otherwise (the one-chunk case reads the table before the first DMA into that region)
    constexpr int THIRD_DW = 2 * (PX_B / 4) + W_DW;
    static_assert(W_DW + OFF_DW + 256 <= PX_B / 4, "weight buffer 1, the offset table and the STDF queues must fit the third pixel buffer");
    __shared__ __attribute__((aligned(16))) uint32_t lds[3 * (PX_B / 4) + W_DW];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int W = a.width, H = a.height;
    const size_t oplane_px = (size_t)W * (size_t)a.out_rows;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane(uint32_t(uintptr_t((lds_ptr_t)lds)));
    const size_t shift_stride = (size_t)a.in_rows * a.planar_pitch; // one byte plane: the rows this context holds
    int2 *off_table = reinterpret_cast<int2 *>(lds + THIRD_DW + W_DW);
    for(int g = threadIdx.x; g < a.n_images; g += 256)
    {
        const lfi_int2 o = a.focused[g];
        off_table[g] = make_int2(o.x + a.planar_phase[g], o.y); // the image's phase inside the planar copy folded into its x offset
    }
    const bool single_chunk = a.k_pad <= KC;
    const bool static_weights = single_chunk && view_passes == 1;

    auto issue_weights = [&](const int pass, const int k0, const int wb) {
        const int kc = min(KC, a.k_pad - k0);
        const uint32_t w_addr = lds_base + (wb ? THIRD_DW : 2 * (PX_B / 4)) * 4;
        for(int o = wave; 8 * o < kc; o += 4)
            if(lane < VPP)
                dma16(a.w16s + (size_t)(a.v0 + pass * VPP + lane) * a.k_pad + k0 + 8 * o, w_addr + uint32_t(o) * (VPP * 16));
    };
    // pixel bytes of one unit: piece p (1 KB) = channel p / 8, images 8 (p % 8) … +7 of the chunk, 128 bytes = 128 pixels each (eight
    // lanes × 16 bytes per image); wave w moves pieces w, w + 4, …  What a lane needs per piece — its image's integer offsets and
    // the byte offset of that image's (channel, shift 0) plane — depends on the chunk only: looked up once per kernel when the
    // stack is one chunk, once per unit otherwise.
    struct Pieces
    {
        int ox[6], oy[6];
        size_t plane0[6];
    };
    auto lookup = [&](const int k0) {
        Pieces pc;
#pragma unroll
        for(int j = 0; j < 6; j++)
        {
            const int p = wave + 4 * j, c = p >> 3, octet = p & 7;
            const int g = min(k0 + 8 * octet + (lane >> 3), a.n_images - 1); // padded images (zero weights) re-read the last one
            const int2 o = off_table[g];
            pc.ox[j] = o.x;
            pc.oy[j] = o.y;
            pc.plane0[j] = ((size_t)g * 3 + c) * shift_stride;
        }
        return pc;
    };
    // reverse: this launch walks the tile sequence backwards (consecutive launches alternate: Infinity Cache reuse, see launch_p3)
    auto tile_of = [&](const int t_seq) { return reverse ? n_tiles - 1 - t_seq : t_seq; };
    auto issue_pixels = [&](const int t_seq, const int k0, const int pb, const Pieces &pc) {
        const int t = tile_of(t_seq);
        const int ty = t / tiles_x;
        const int y = a.out_y0 + ty;
        const int x0 = (t - ty * tiles_x) * TPX;
        const int kc = min(KC, a.k_pad - k0);
        const uint32_t dst = lds_base + (pb == 2 ? THIRD_DW * 4 : uint32_t(pb) * PX_B);
#pragma unroll
        for(int j = 0; j < 6; j++)
        {
            const int p = wave + 4 * j;
            if(8 * (p & 7) >= kc)
                continue; // wave-uniform: the chunk is shorter (its length is a multiple of 16)
            // the run starts at pixel x0 + ox = byte x0 + ox + padx of the plane row, ANY byte (LDS-DMA sources need no alignment:
            // tools/probe_ldsdma_bytes.hip); the padding exceeds every offset
            const int sy = clampi(y + pc.oy[j], 0, H - 1) - a.in_y0; // clamp in the full image, then index the held rows
            const int start = x0 + pc.ox[j] + a.planar_padx;
            dma16(a.planar + pc.plane0[j] + (size_t)sy * a.planar_pitch + start + 16 * (lane & 7), dst + uint32_t(p) * 1024u);
        }
    };

    const int G = gridDim.x;
    int t = int(tile_of_block(blockIdx.x, gridDim.x, a.flags));
    if(t >= n_tiles)
        return;
    __syncthreads(); // the offset table is complete
    int pass = 0, k0 = 0, pbuf = 0, wbuf = 0;
    int prev_stores = 0;
    Pieces pieces = lookup(0);
    issue_weights(0, 0, 0);
    issue_pixels(t, 0, 0, pieces);

    if(static_weights && ring3 && !STDF)
    {
        // ---- one chunk, one view pass: a ring of three pixel buffers, tiles fetched TWO ahead (24 KB per tile: with two
        // buffers only 48 KB per CU would be in flight).  Stream of this wave's VMEM operations: … DMA(u+2) stores(u) DMA(u+3)
        // stores(u+1) …, so tile u has landed when at most stores(u−2) + DMA(u+1) + stores(u−1) operations are outstanding.
        const int kc = a.k_pad;
        int n_dma = 0; // DMA instructions of this wave per tile
#pragma unroll
        for(int j = 0; j < 6; j++)
            n_dma += 8 * ((wave + 4 * j) & 7) < kc ? 1 : 0;
        if(t + G < n_tiles)
            issue_pixels(t + G, 0, 1, pieces);
        f32x16 acc3[MT][3]; // never cleared: the first MFMA of a tile takes a zero C operand
        const u32x4 *w_buf = reinterpret_cast<const u32x4 *>(lds + 2 * (PX_B / 4));
        int buf3 = 0, st1 = 0, st2 = 0; // stores of the previous and of the one-before-previous epilogue
        while(true)
        {
            const bool next_in_flight = t + G < n_tiles;
            const int allowed = st2 + (next_in_flight ? n_dma : 0) + st1;
            switch(min(allowed, 63) >> 3)
            {
                case 7: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
                case 6: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
            __builtin_amdgcn_s_barrier(); // everybody's pieces of tile t have landed; everybody is done with tile t − 1's buffer
            asm volatile("" ::: "memory");
            const int buf_next2 = buf3 == 0 ? 2 : buf3 - 1; // (buf3 + 2) % 3 = the buffer tile t − 1 used
            if(t + 2 * G < n_tiles)
                issue_pixels(t + 2 * G, 0, buf_next2, pieces);

            const uint8_t *col = reinterpret_cast<const uint8_t *>(lds + (buf3 == 2 ? THIRD_DW : buf3 * (PX_B / 4))) + wave * 32 + r + 8 * h * TPX;
            unit_ten_bytes<MT, TPX, KC, true>(col, w_buf, r, h, kc, acc3);
            const int tt = tile_of(t), ty = tt / tiles_x;
            st2 = st1;
            st1 = store_tile<false, MT, NT_STORE, false>(a, acc3, a.v0, ty, (tt - ty * tiles_x) * TPX + wave * 32, r, h, oplane_px);
            if(!next_in_flight)
                break;
            t += G;
            buf3 = buf3 == 2 ? 0 : buf3 + 1;
        }
        return;
    }

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
        // next unit (chunks innermost, then view passes, then tiles)
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
        const int nwbuf = static_weights ? wbuf : (wbuf ^ 1);

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
            if(!static_weights)
                issue_weights(npass, nk0, nwbuf);
            if(next_needs_pixels)
            {
                if(!single_chunk)
                    pieces = lookup(nk0);
                issue_pixels(nt, nk0, npbuf, pieces);
            }
        }

        const int kc = min(KC, a.k_pad - k0);
        const u32x4 *w_buf = reinterpret_cast<const u32x4 *>(lds + (wbuf ? THIRD_DW : 2 * (PX_B / 4)));
        unit_ten_bytes<MT, TPX, KC, false>(reinterpret_cast<const uint8_t *>(lds) + pbuf * PX_B + wave * 32 + r + 8 * h * TPX, w_buf, r, h, kc, acc);

        prev_stores = 0;
        if(k0 + KC >= a.k_pad) // last chunk of the tile's pass: epilogue
        {
            const int tt = tile_of(t), ty = tt / tiles_x;
            if constexpr(STDF)
                prev_stores = store_tile_filtered<MT, NT_STORE, KC, TPX>(
                    a, acc, a.v0 + pass * VPP, ty, (tt - ty * tiles_x) * TPX + wave * 32, lane, oplane_px,
                    reinterpret_cast<const uint8_t *>(lds) + pbuf * PX_B + wave * 32, reinterpret_cast<const uint16_t *>(w_buf), kc,
                    reinterpret_cast<uint16_t *>(lds + THIRD_DW + W_DW + OFF_DW) + wave * 128);
            else
                prev_stores = store_tile<false, MT, NT_STORE, true>(a, acc, a.v0 + pass * VPP, ty, (tt - ty * tiles_x) * TPX + wave * 32, r, h, oplane_px);
        }
        if(!have_next)
            break;
        t = nt;
        pass = npass;
        k0 = nk0;
        pbuf = npbuf;
        wbuf = nwbuf;
    }
}

} // namespace lfi
