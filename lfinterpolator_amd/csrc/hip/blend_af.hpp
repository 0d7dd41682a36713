// blend_af.hpp — ALL-FOCUS renders of light fields with MORE than 128 images (15×15 grids: three or four chunks of 64 images), round 4.
//
//   blend_afs  STD     Kernels::Standard::process<true> (reference src/kernels.cu:312-342): the exact fp32 fmaf chain over per-pixel
//                      warped samples, by the band method of blend_stdx.hpp / blend_stdxa.hpp — with every sample gathered ONCE.
//   (blend_aft, a TEN_WM kernel on the same pipeline, was built, measured slower than blend_persist — 3.02 against 2.65 ms at config 5 — and
//    removed in round 4: profiles/r04_notes.md §9.)
//
// Why a second all-focus STD kernel.  blend_stdxa (round 3) keeps blend_persist's geometry — tiles of 128 pixels, two 32 KB pixel buffers per
// workgroup — so a tile's stack of 225 images (115 KB) does not stay in LDS, and the chain's bytes of chunks 2 and 3 are gathered a SECOND
// time: 31 GB of fabric traffic for 7.5 GB of samples, 4.7 ms at BASELINE config 5 (profiles/r03_pmc_allfocus_summary.txt).  The per-pixel
// gather is what an all-focus render costs (tools/probe_gather.hip: reading the samples of the structured scene's estimated map ONCE, with
// no arithmetic and no stores, takes 2.1 ms; a constant map 1.1 ms), so it must not be paid twice.  Here a tile is 64 pixels: the whole
// stack is 4 × 16 KB, it STAYS in LDS from the first MFMA to the last link of the chain, and a fifth slot lets the next tile's first chunk
// travel early.  What made room: no weight buffers at all — a wave's MFMA A fragments (its 32 views × all images, ×2^15) live in
// registers for the whole launch (blend_p3's arrangement), and the chain reads its weights from them across lanes (ds_bpermute).
//
//   workgroup = 4 waves = 64 pixels × 64 views: wave w ↔ pixels 32(w&1) + r, views 32(w>>1) + …  on v_mfma_f32_32x32x16_f16
//   LDS       = (NCH + 1) slots × [64 images][64 pixels] RGBA dwords (80 KB for four chunks: two workgroups per CU)
//   per tile  M(1) … M(NCH−1), MC(0), C(1) … C(NCH−1): fp16 MFMA k-loops in fetch order; after the last one the band test, the RGBA
//             stores of the rounded bytes and the queue of sums inside the band (blend_stdxa's, two per lane); then the chain over the
//             chunks in ascending image order, each from the slot its chunk landed in; byte patches after the last link
//   fetches   in the order c1, c2, …, c(NCH−1), c0 into a ring of NCH + 1 slots (a tile's slots shift by one from tile to tile):
//             the next tile's c1 goes to the spare slot at the top of unit 1, its c(k) to the slot of this tile's c(k−1) once C(k−1)
//             is done (top of C(k)), c0 to the slot of c(NCH−1) at the top of the next tile's first unit — so every fetch is issued at
//             least one long (MFMA) unit before its data is needed, usually two or three.
//   waits     one hand-counted s_waitcnt vmcnt per M unit (pieces issued after the needed fetch stay in flight: the wave keeps a running
//             count of its LDS-DMA instructions and the count at the end of each fetch), one barrier per unit.  The loop contains no
//             compiler-tracked vector load: the focus-map values of a tile come through the SCALAR cache (64 dwords of one map row,
//             s_load + v_writelane), offsets through scalar loads, weights from registers.
//
// Band, error bounds, preconditions and the exactness argument: blend_stdx.hpp.  Bit-exact against the oracle (same tests as every STD kernel).
#pragma once

#include <type_traits>

#include "blend_p3.hpp"
#include "blend_ten_persist.hpp"

namespace lfi {

// measurement builds only (hipcc -DLFI_AF_ABL=mask, tools/af_ablate.sh): 1 = no chain arithmetic, 2 = no MFMA k-loops, 4 = no gathers,
// 8 = no band test / stores / queue.  Outputs are wrong by construction.
#ifndef LFI_AF_ABL
#define LFI_AF_ABL 0
#endif

constexpr int AF_TPX = 64;                   // pixels per tile (one row)
constexpr int AF_KC = 64;                    // images per chunk
constexpr int AF_SLOT_DW = AF_KC * AF_TPX;   // dwords per slot (16 KB)

// all but the wave's `allowed` youngest vector-memory operations are done (rounded down to a multiple of four: stricter, never wrong)
__device__ __forceinline__ void af_wait(const int allowed)
{
    switch(min(allowed, 63) >> 2)
    {
        case 15: LFI_P3_WAIT(60); break;
        case 14: LFI_P3_WAIT(56); break;
        case 13: LFI_P3_WAIT(52); break;
        case 12: LFI_P3_WAIT(48); break;
        case 11: LFI_P3_WAIT(44); break;
        case 10: LFI_P3_WAIT(40); break;
        case 9: LFI_P3_WAIT(36); break;
        case 8: LFI_P3_WAIT(32); break;
        case 7: LFI_P3_WAIT(28); break;
        case 6: LFI_P3_WAIT(24); break;
        case 5: LFI_P3_WAIT(20); break;
        case 4: LFI_P3_WAIT(16); break;
        case 3: LFI_P3_WAIT(12); break;
        case 2: LFI_P3_WAIT(8); break;
        case 1: LFI_P3_WAIT(4); break;
        default: LFI_P3_WAIT(0); break;
    }
}

// The raw map dwords of a tile's 64 pixels, one per lane.  Whole tiles: four s_load_dwordx16 through the scalar cache and 64
// v_writelane — counted by lgkmcnt, invisible to the pipeline's vmcnt bookkeeping.  The ragged last tile of a row (and only it) uses
// per-lane clamped loads, which the compiler waits for with vmcnt(0): a drain of the prefetch once per image row at most.
__device__ __forceinline__ uint32_t af_map_raw(const uint8_t *map_plane, const int W, const int H, const int x0, const int y_img, const int lane)
{
    const int y = clampi(y_img, 0, H - 1);
    if(x0 + AF_TPX <= W) // wave-uniform
    {
        typedef const __attribute__((address_space(4))) uint32_t *const_u32_ptr;
        const const_u32_ptr row = (const_u32_ptr)(uintptr_t)(reinterpret_cast<const uint32_t *>(map_plane) + (size_t)y * W + x0);
        uint32_t m = 0u;
#pragma unroll
        for(int i = 0; i < AF_TPX; i++)
            asm("v_writelane_b32 %0, %1, %2" : "+v"(m) : "s"(row[i]), "n"(i)); // (this clang has no writelane builtin)
        return m;
    }
    return reinterpret_cast<const uint32_t *>(map_plane)[(size_t)y * W + clampi(x0 + lane, 0, W - 1)];
}

// loadFocusFromMap (src/kernels.cu:134-137): decode_focus's arithmetic on a map dword that is already here
__device__ __forceinline__ float af_map_focus(const uint32_t m, const float focus, const float range)
{
    return __builtin_fmaf(__fdiv_rn(static_cast<float>(m & 0xffu), 255.0f), range, focus);
}

// The per-pixel gather of one chunk of a tile into a slot: per image one 4-byte LDS-DMA per lane at (int)fma(f, offset, coord), clamped
// (src/kernels.cu:78-82, :125); wave w moves images w, w + 4, ….  ox_tab / oy_tab: the chunk's float offsets, image 64·chunk + l in lane l
// (registers filled once per launch: a v_readlane per image instead of a scalar load and its wait).  Returns the wave's DMA instructions
// (wave-uniform).  (Measured and removed, profiles/r04_notes.md: one 16-byte-per-lane DMA for four images whose shifts are uniform over
// the tile — the test costs more vector instructions than the three DMA instructions it saves; the kernel is bound by instruction issue.)
__device__ __forceinline__ int af_gather(const KernelArgs &a, const uint32_t *grid32, const size_t plane_px, const uint32_t slot_addr, const int chunk, const int x0,
                                         const int y_img, const float f, const int wave, const int lane, const float ox_tab, const float oy_tab)
{
    const int W = a.width, H = a.height;
    const int k0 = AF_KC * chunk;
    const int kn = min(AF_KC, a.n_images - k0);
    const float xf = static_cast<float>(x0 + lane), yf = static_cast<float>(y_img);
    int count = 0;
    // rows are indexed inside the held window: −in_y0·W folded into the scalar base (W, H < 2^24: lfi_set_grid)
    const uint32_t *base = grid32 + (size_t)(k0 + wave) * plane_px - (size_t)a.in_y0 * W;
    for(int gi = wave; gi < kn; gi += 4)
    {
        const float ox = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ox_tab), gi));
        const float oy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, oy_tab), gi));
        const int sx = clampi(static_cast<int>(__builtin_fmaf(f, ox, xf)), 0, W - 1); // min(max()) → v_med3_i32
        const int sy = clampi(static_cast<int>(__builtin_fmaf(f, oy, yf)), 0, H - 1);
        if constexpr(!(LFI_AF_ABL & 4))
        {
            dma4_s(base, (__umul24(uint32_t(sy), uint32_t(W)) + uint32_t(sx)) << 2, slot_addr + uint32_t(gi) * (AF_TPX * 4u));
            count++;
        }
        base += 4 * plane_px;
    }
    return count;
}

// fp16 MFMA k-loop of one chunk from a slot: this wave's 32 pixels × 32 views, weights from the A fragments in registers.  The LDS reads of
// k-step ks + 1 are issued before the MFMAs of k-step ks (register double buffer + a scheduling barrier): written as read → permute → MFMA
// per k-step, every k-step began with an exposed LDS latency (0.84 ms of k-loops at config 5, tools/af_ablate.sh).
template <bool ZERO_FIRST>
__device__ __forceinline__ void af_mfma_unit(const uint32_t *col, const half8 *wk, const int kc, f32x16 (&acc)[3])
{
    constexpr int KS = AF_KC / 16;
    f32x16 zero16;
#pragma unroll
    for(int e = 0; e < 16; e++)
        zero16[e] = 0.0f;
    uint32_t px[2][8];
#pragma unroll
    for(int j = 0; j < 8; j++)
        px[0][j] = col[j * AF_TPX];
#pragma unroll
    for(int ks = 0; ks < KS; ks++)
    {
        if(16 * ks < kc) // wave-uniform: a chunk's length is a multiple of 16
        {
            if(ks + 1 < KS && 16 * (ks + 1) < kc)
            {
#pragma unroll
                for(int j = 0; j < 8; j++)
                    px[(ks + 1) & 1][j] = col[(16 * (ks + 1) + j) * AF_TPX];
            }
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t(&p)[8] = px[ks & 1];
            u32x4 bc[3];
#pragma unroll
            for(int q = 0; q < 4; q++)
            {
                bc[0][q] = pack_subnormal_pair<0>(p[2 * q], p[2 * q + 1]);
                bc[1][q] = pack_subnormal_pair<1>(p[2 * q], p[2 * q + 1]);
                bc[2][q] = pack_subnormal_pair<2>(p[2 * q], p[2 * q + 1]);
            }
#pragma unroll
            for(int c = 0; c < 3; c++)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wk[ks], __builtin_bit_cast(half8, bc[c]), (ZERO_FIRST && ks == 0) ? zero16 : acc[c], 0, 0, 0);
        }
    }
}

// ---- STD ---------------------------------------------------------------------------------------------------------------------------------
// NCH: chunks of 64 images (3 or 4; fewer chunks: blend_stdxa, which re-gathers nothing there).  One launch renders views [a.v0, min(a.v1, a.v0 + 64)).
template <bool NT_STORE, int NCH>
__global__ void __launch_bounds__(256, 2) blend_afs(const KernelArgs a, const int tiles_x, const int n_tiles)
{
    static_assert(NCH >= 3 && NCH <= 4, "three or four chunks of 64 images");
    constexpr int NS = NCH + 1;     // slots
    constexpr int NU = 2 * NCH - 1; // units per tile: M(1) … M(NCH−1), MC(0), C(1) … C(NCH−1)
    constexpr int QCAP = 128;       // queued sums per wave and tile: two per lane
    constexpr int KS = AF_KC / 16;
    __shared__ __attribute__((aligned(16))) uint32_t lds[NS * AF_SLOT_DW];
    static_assert(sizeof(lds) <= 81920, "two workgroups per CU");

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int ph = wave & 1, vh = wave >> 1; // this wave's pixel half and view half
    const int W = a.width, H = a.height;
    const uint32_t *grid32 = reinterpret_cast<const uint32_t *>(a.grid);
    const size_t plane_px = (size_t)W * (size_t)a.in_rows;
    const size_t oplane_px = (size_t)W * (size_t)a.out_rows;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane(uint32_t(uintptr_t((lds_ptr_t)lds)));
    const uint8_t *map_plane = a.maps + (size_t)a.map_index * (size_t)W * H * 4; // maps are whole-image planes

    // this wave's 32 views (a.v0 + 32·vh …): all their weights ×2^15 as MFMA A fragments (k-step s = images 16s … 16s+15); lane (r, h)
    // holds view r, images 16s + 8h + j — the chain reads them back through ds_bpermute
    const int vw0 = a.v0 + 32 * vh;
    half8 wreg[KS * NCH];
#pragma unroll
    for(int s = 0; s < KS * NCH; s++)
    {
        const int k = 16 * s + 8 * h;
        u32x4 w = {0u, 0u, 0u, 0u};
        if(k < a.k_pad) // rows are k_pad halves long (a multiple of 16): nothing is read across a row's end
            w = *reinterpret_cast<const u32x4 *>(a.w16s + (size_t)(vw0 + r) * a.k_pad + k);
        wreg[s] = __builtin_bit_cast(half8, w);
    }
    // the compiler's wait for these loads belongs HERE, before any LDS-DMA is in flight (blend_p3.hpp)
#pragma unroll
    for(int s = 0; s < KS * NCH; s++)
        asm volatile("" : "+v"(wreg[s]));

    // the float offsets of all images, chunk c's image 64c + l in lane l of ox_tab[c] / oy_tab[c]: the gathers read them with v_readlane
    float ox_tab[NCH], oy_tab[NCH];
#pragma unroll
    for(int c = 0; c < NCH; c++)
    {
        const lfi_float2 o = a.offsets[min(AF_KC * c + lane, a.n_images - 1)];
        ox_tab[c] = o.x;
        oy_tab[c] = o.y;
    }
#pragma unroll
    for(int c = 0; c < NCH; c++)
        asm volatile("" : "+v"(ox_tab[c]), "+v"(oy_tab[c]));

    const int G = gridDim.x;
    const int t0 = int(tile_of_block(blockIdx.x, gridDim.x, a.flags));
    if(t0 >= n_tiles)
        return;
    const int T = (n_tiles - 1 - t0) / G + 1; // this workgroup's tiles: t0, t0 + G, …

    auto tile_xy = [&](const int j, int &ty, int &x0) {
        const int t = t0 + j * G;
        ty = t / tiles_x; // row inside the output window
        x0 = (t - ty * tiles_x) * AF_TPX;
    };
    auto fetch_order = [](const int c) { return c == 0 ? NCH - 1 : c - 1; }; // position of chunk c among a tile's fetches (and M units)
    auto wrap = [](const int s) { return s >= NS ? s - NS : s; };

    // ---- band constants (blend_stdx.hpp) ------------------------------------------------------------------------------------------------
    const float nf = float(a.n_images);
    const float c0 = nf * std_accumulation_bound(a.flags) + 0x1p-12f;
    const float bmax_acc = (c0 + nf * 0x1p-16f) * 0x1p-9f;
    const float base_acc = (0.5f - c0) * 0x1p-9f;
    const float chain_acc = nf * 0x1p-24f;
    auto byte_of15 = [](const float s15) { return __builtin_bit_cast(uint32_t, s15 * 0x1p-15f + 8388608.0f) & 0xffu; }; // the overflow path's scale (bytes as integers)
    auto byte_of = [](const float s9) { return __builtin_bit_cast(uint32_t, s9 * 0x1p9f + 8388608.0f) & 0xffu; };     // the chain's: 2^-9 · sum

    // ---- the chain over chunk CC for a queued sum = pixel of the wave (5 bits) | view of the wave << 5 (5 bits) | channel << 11 ------------
    // pixel bytes from the chunk's slot; weights from the wave's A fragments across lanes (view v's images 16ks + 8hh + j: lane v + 32hh,
    // dword j/2).  All LDS traffic of a k-step is issued before its first fma (blend_stdx.hpp).  Every lane runs it (ds_bpermute reads
    // nothing from inactive lanes); lanes without a sum run it on entry 0 and store nothing.
    auto chain = [&](auto cc_tag, const int slot, const int kc, const uint32_t entry, float &s15) {
        constexpr int cc = decltype(cc_tag)::value;
        const uint32_t px = entry & 31u, view = (entry >> 5) & 31u, ch = (entry >> 11) & 3u;
        const uint8_t *pb = reinterpret_cast<const uint8_t *>(lds + slot * AF_SLOT_DW) + (32 * ph + px) * 4 + ch;
        float s = s15;
        uint32_t w2[2][8], pbyte[2][16];
        // the LDS traffic of k-step ks + 1 — 8 weight pairs across lanes, 16 pixel bytes — is issued before the 16 dependent fmas of k-step ks
        auto stage = [&](const int ks, uint32_t (&w)[8], uint32_t (&p)[16]) {
            const u32x4 wv = __builtin_bit_cast(u32x4, wreg[KS * cc + ks]);
#pragma unroll
            for(int hh = 0; hh < 2; hh++)
#pragma unroll
                for(int d = 0; d < 4; d++)
                    w[4 * hh + d] = uint32_t(__builtin_amdgcn_ds_bpermute(4 * (int(view) + 32 * hh), int(wv[d])));
#pragma unroll
            for(int j = 0; j < 16; j++)
                p[j] = pb[(16 * ks + j) * AF_TPX * 4];
        };
        stage(0, w2[0], pbyte[0]);
#pragma unroll
        for(int ks = 0; ks < KS; ks++)
        {
            if(16 * ks >= kc) // wave-uniform
                break;
            if(ks + 1 < KS && 16 * (ks + 1) < kc)
                stage(ks + 1, w2[(ks + 1) & 1], pbyte[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for(int j = 0; j < 16; j++) // image 64·cc + 16·ks + j, ascending
            {
                const uint32_t pair = w2[ks & 1][j >> 1];
                const float w = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(j & 1 ? pair >> 16 : pair)));
                // the byte AS an fp16 subnormal (byte·2^-24): v_fma_mix_f32 widens both halves itself, no v_cvt_f32_ubyte per image (blend_stdx.hpp)
                s = __builtin_fmaf(static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(pbyte[ks & 1][j]))), w, s); // addWeighted, src/kernels.cu:292-299
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        s15 = s;
    };

    // ---- prologue: the first tile's focus values, its chunks c1 … c(NCH−1) (c0 follows at the top of its first unit) -----------------------
    int issued = 0;            // this wave's LDS-DMA instructions so far
    int mark_cur[NCH] = {};    // `issued` at the end of the fetch of chunk c of the tile being computed …
    int mark_next[NCH] = {};   // … and of the next tile
    int slot_base = 0;         // slot of the current tile's first fetch; the next tile's is slot_base − 1 (mod NS)
    float f_issue;             // the focus value of this lane's pixel of the tile whose chunks are being issued
    {
        int ty, x0;
        tile_xy(0, ty, x0);
        f_issue = af_map_focus(af_map_raw(map_plane, W, H, x0, a.out_y0 + ty, lane), a.focus, a.range);
#pragma unroll
        for(int c = 1; c < NCH; c++)
        {
            issued += af_gather(a, grid32, plane_px, lds_base + uint32_t(wrap(slot_base + fetch_order(c))) * (AF_SLOT_DW * 4u), c, x0, a.out_y0 + ty, f_issue, wave, lane, ox_tab[c], oy_tab[c]);
            mark_cur[c] = issued;
        }
    }
    uint32_t m_next = 0u; // raw map dword of this lane's pixel of the NEXT tile (loaded at the top of a tile's first unit, decoded at its second)
    int jt = 0;           // the tile being computed (ordinal)
    int ty, x0, tyn = 0, x0n = 0; // row (inside the output window) and first column of the tile being computed / of the next one
    tile_xy(0, ty, x0);
    if(T > 1)
        tile_xy(1, tyn, x0n);
    const int nviews = __builtin_amdgcn_readfirstlane(min(a.v1 - vw0, 32)); // ≤ 0: this wave only helps with the DMA and the MFMAs
    uint32_t entry0 = 0u, entry1 = 0u; // bit 15: valid
    float s0 = 0.0f, s1 = 0.0f;
    int queued = 0; // wave-uniform
    f32x16 acc[3];

    auto unit = [&](auto u_tag) {
        constexpr int u = decltype(u_tag)::value;
        constexpr int cc = u < NCH - 1 ? u + 1 : (u == NCH - 1 ? 0 : u - NCH + 1);
        constexpr bool is_m = u <= NCH - 1, is_mc = u == NCH - 1, is_last = u == NU - 1;
        const bool has_next = jt + 1 < T;
        if constexpr(u == 0)
        {
            // the next tile's map values: through the scalar cache, before the barrier (its wait covers their latency)
            if(has_next)
                m_next = af_map_raw(map_plane, W, H, x0n, a.out_y0 + tyn, lane);
        }
        if constexpr(is_m)
        {
            // this wave's pieces of chunk cc have landed; what it issued after them stays in flight.  (Unit 0 of a tile: c0 is not yet
            // issued — the wait comes first.)  Stores and byte patches are not counted: an undercount only makes the wait stricter.
            af_wait(issued - mark_cur[cc]);
        }
        __builtin_amdgcn_s_barrier(); // everybody's pieces have landed; everybody is done with the slot about to be refilled
        asm volatile("" ::: "memory");
        // ---- fetches of this unit (file header) --------------------------------------------------------------------------------------------
        if constexpr(u == 0)
        {
            // c0 of THIS tile into the slot of the previous tile's c(NCH−1), whose last link of the chain has run
            issued += af_gather(a, grid32, plane_px, lds_base + uint32_t(wrap(slot_base + fetch_order(0))) * (AF_SLOT_DW * 4u), 0, x0, a.out_y0 + ty, f_issue, wave, lane, ox_tab[0], oy_tab[0]);
            mark_cur[0] = issued;
        }
        const int slot_base_next = slot_base == 0 ? NS - 1 : slot_base - 1;
        if constexpr(u == 1)
        {
            if(has_next)
            {
                f_issue = af_map_focus(m_next, a.focus, a.range);
                // c1 of the NEXT tile into the spare slot
                issued += af_gather(a, grid32, plane_px, lds_base + uint32_t(wrap(slot_base_next + fetch_order(1))) * (AF_SLOT_DW * 4u), 1, x0n, a.out_y0 + tyn, f_issue, wave, lane, ox_tab[1], oy_tab[1]);
                mark_next[1] = issued;
            }
        }
        if constexpr(u >= NCH + 1)
        {
            // C(k), k ≥ 2: the next tile's c(k) into the slot of this tile's c(k−1), done with C(k−1)
            constexpr int k = u - NCH + 1;
            if(has_next)
            {
                issued += af_gather(a, grid32, plane_px, lds_base + uint32_t(wrap(slot_base_next + fetch_order(k))) * (AF_SLOT_DW * 4u), k, x0n, a.out_y0 + tyn, f_issue, wave, lane, ox_tab[k], oy_tab[k]);
                mark_next[k] = issued;
            }
        }

        const int kc = min(AF_KC, a.k_pad - AF_KC * cc);
        const int slot = wrap(slot_base + fetch_order(cc));
        const int xw = x0 + 32 * ph;
        if constexpr(is_m && !(LFI_AF_ABL & 2))
        {
            const uint32_t *col = lds + slot * AF_SLOT_DW + 32 * ph + r + 8 * h * AF_TPX;
            if constexpr(u == 0)
                af_mfma_unit<true>(col, &wreg[KS * cc], kc, acc); // the first chunk of a tile: zero C operand
            else
                af_mfma_unit<false>(col, &wreg[KS * cc], kc, acc);
        }
        if constexpr(is_mc && !(LFI_AF_ABL & 8))
        {
            // ---- epilogue: round every sum, store RGBA, find the sums inside the band (accumulator e ↔ view (e&3) + 8(e>>2) + 4h of the wave)
            uint32_t flagged[3] = {0u, 0u, 0u}; // per channel: bit e
            uint32_t valid_mask = 0u;
            const bool lane_x_ok = xw + r < W;
            uint8_t *plane0 = a.views + ((size_t)vw0 * oplane_px + (size_t)ty * W + xw) * 4;
            const uint32_t lane_off = (uint32_t(r) + uint32_t(4 * h) * uint32_t(oplane_px)) * 4u;
            if(xw < W && nviews > 0)
            {
                const int lim = max(nviews - 4 * h, 0);
                const int n_e = 4 * (lim >> 3) + min(lim & 7, 4); // the lane's valid sums are a prefix of the e sequence
                valid_mask = lane_x_ok ? (n_e >= 16 ? 0xffffu : (1u << n_e) - 1u) : 0u;
#pragma unroll
                for(int e = 0; e < 16; e++)
                {
                    const int vrow = (e & 3) + 8 * (e >> 2); // + 4h per half-wave
                    if(vrow >= nviews) // wave-uniform
                        continue;
                    uint32_t bits[3];
#pragma unroll
                    for(int c = 0; c < 3; c++)
                    {
                        const float v = acc[c][e];
                        const float tt = v + 16384.0f; // rounds S̃ to an integer (RN-even), left in the low mantissa bits
                        const float dist = v - (tt - 16384.0f);
                        const float pow2 = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v + bmax_acc) & 0x7f800000u);
                        const float inside = __builtin_fmaf(-chain_acc, pow2, base_acc);
                        flagged[c] |= __builtin_fabsf(dist) > inside ? 1u << e : 0u;
                        bits[c] = __builtin_bit_cast(uint32_t, tt);
                    }
                    const uint32_t rg = __builtin_amdgcn_perm(bits[1], bits[0], 0x0c0c0400u); // [R, G, 0, 0]
                    const uint32_t rgba = __builtin_amdgcn_perm(bits[2], rg, 0x0d040100u);  // [R, G, B, 0xff]
                    uint32_t *out = reinterpret_cast<uint32_t *>(plane0 + (size_t)vrow * oplane_px * 4 + lane_off);
                    if(lane_x_ok && vrow + 4 * h < nviews)
                    {
                        if constexpr(NT_STORE)
                            __builtin_nontemporal_store(rgba, out);
                        else
                            *out = rgba;
                    }
                }
            }
#pragma unroll
            for(int c = 0; c < 3; c++)
                flagged[c] &= valid_mask;
            // ---- compact the flagged sums across the wave into entry0 (slots 0 … 63) and entry1 (64 … 127) with ds_permute; what does not
            // fit is recomputed here and now from global memory (blend_stdxa.hpp)
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
                const int e = mine ? __builtin_ctz(word) : 0;
                const uint32_t rest = word & (word - 1u);
                flagged[0] = c == 0 ? rest : flagged[0];
                flagged[1] = c == 1 ? rest : flagged[1];
                flagged[2] = c == 2 ? rest : flagged[2];
                const uint32_t view = uint32_t((e & 3) + 8 * (e >> 2) + 4 * h);
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
                        const uint16_t *wrow = a.w16s + (size_t)(vw0 + view) * a.k_pad;
                        float s = 0.0f;
                        for(int g = 0; g < a.n_images; g++)
                        {
                            const lfi_float2 o = a.offsets[g];
                            const int sx = clampi(warp_float(x, f, o.x), 0, W - 1), sy = clampi(warp_float(y, f, o.y), 0, H - 1) - a.in_y0;
                            const uint32_t p = (grid32[(size_t)g * plane_px + (size_t)sy * W + sx] >> (8 * c)) & 0xffu;
                            const float w = static_cast<float>(__builtin_bit_cast(_Float16, wrow[g]));
                            s = __builtin_fmaf(static_cast<float>(p), w, s);
                        }
                        (plane0 + ((size_t)view * oplane_px + r) * 4)[c] = static_cast<uint8_t>(byte_of15(s));
                    }
                }
                count += n;
            }
            queued = min(count, QCAP);
            s0 = s1 = 0.0f;
        }
        if constexpr(u >= NCH - 1 && !(LFI_AF_ABL & 1))
        {
            // the chain over chunk cc for the queued sums, from the slot the chunk landed in
            if(queued > 0)
                chain(std::integral_constant<int, cc>{}, slot, kc, entry0, s0);
            if(queued > 64)
                chain(std::integral_constant<int, cc>{}, slot, kc, entry1, s1);
        }
        if constexpr(is_last)
        {
            // the chain's bytes over the rounded ones (same wave as the dword stores: in order)
            uint8_t *plane0 = a.views + ((size_t)vw0 * oplane_px + (size_t)ty * W + xw) * 4;
            if(entry0 & 0x8000u)
                (plane0 + ((size_t)((entry0 >> 5) & 31u) * oplane_px + (entry0 & 31u)) * 4)[(entry0 >> 11) & 3u] = static_cast<uint8_t>(byte_of(s0));
            if(queued > 64 && (entry1 & 0x8000u))
                (plane0 + ((size_t)((entry1 >> 5) & 31u) * oplane_px + (entry1 & 31u)) * 4)[(entry1 >> 11) & 3u] = static_cast<uint8_t>(byte_of(s1));
            // next tile
            jt++;
            slot_base = slot_base_next;
            ty = tyn;
            x0 = x0n;
            if(jt + 1 < T)
                tile_xy(jt + 1, tyn, x0n);
#pragma unroll
            for(int c = 1; c < NCH; c++)
                mark_cur[c] = mark_next[c];
        }
    };
    while(jt < T)
        p3_for_each_chunk<NU>([&](auto u_tag) { unit(u_tag); });
}

} // namespace lfi
