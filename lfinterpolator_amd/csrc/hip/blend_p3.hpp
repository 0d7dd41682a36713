// blend_p3.hpp — TEN_WM with three bytes per pixel on BOTH sides of the kernel: the planar, alpha-free copy of the inputs
// (blend_planar.hpp: one byte plane per image and channel) is read, and the views are written as alpha-free byte planes [view][R,G,B][rows][pitch].
//
// Why: the blend is HBM-bound and the memory system gives this access pattern ≈4.8–5 TB/s whatever the kernel does
// (tools/ablate_out.hip: the same gather with no arithmetic, 929 MB moved: 188–197 µs with RGBA views, 796 MB: 160–165 µs with
// planar views, on one box) — bytes are the lever left.  A quarter of the bytes blend_planar WRITES is the constant alpha = 255
// (reference src/kernels.cu:393 `uchar4{…, 255}`).  This kernel does not write it: views are stored as byte planes when the
// caller opts in (lfi_set_output_layout, include/lfi.h) and the alpha byte is re-created when a view is downloaded.
//
// Byte planes want a lane to own CONSECUTIVE pixels of one view and channel (a dword or more per store, 128 B per plane row and
// store instruction).  blend_planar's wave = 32 pixels × 64 views on v_mfma_f32_32x32x16_f16 gives a lane one pixel; here a
// wave = 128 pixels × 16 views on v_mfma_f32_16x16x32_f16:
//      A = weights  (lane l: view l&15,  images 8(l>>4)+j, j < 8)        — 16 views of this wave, held in registers for the launch
//      B = pixels   (lane l: column l&15, images 8(l>>4)+j)             — eight MFMAs "blk" = 0..7 per k-step and channel;
//                                                                          column n of block blk IS pixel 8n + blk of the tile
//      D            (lane l: column l&15, views 4(l>>4)+i, i < 4)
// so lane (n, kg) ends up with pixels 8n … 8n+7 of views 4kg+i: one 8-byte store per (i, channel), 16 lanes = 128 B of one plane
// row, four views per store instruction.  The column ↔ pixel map costs nothing: one ds_read_b64 at [channel][image][8n] returns
// the lane's eight pixels of an image — one byte per block — and a v_perm_b32 per image pair and block builds the fp16-subnormal
// operand (a pixel byte IS the subnormal's mantissa, blend_ten.hpp).  The four waves of a workgroup share the 128-pixel tile and
// split the 64 views of a pass.
//
// Pipeline: persistent workgroups, a ring of three pixel buffers in LDS filled by LDS-DMA two units ahead (unit = tile × chunk of
// 64 images), one barrier and one hand-counted s_waitcnt vmcnt per unit, as in blend_planar's one-chunk path — generalised to
// any number of chunks (NCH ≤ 4, compile time: the weight fragments of all chunks stay in registers, and the unit loop is unrolled
// over the chunks of a tile so that they are registers named at compile time).  Launches of several chunks run TWO waves of 32 views
// per workgroup instead of four of 16 (template parameter VG below): the pixel operand is then built once for two MFMAs.
// LDS image of a buffer: [channel][octet of images][8 images × 128 B], octets padded so that the four octets a ds_read_b64
// instruction touches (lanes kg = 0..3) fall into different banks.
// Arithmetic and quantisation are blend_planar's (weights ×2^15, RN-even to fp16, saturate, truncate — src/kernels.cu:387-396):
// identical bytes, tested against the oracle and against blend_planar.
// Replaces Kernels::Tensors::process<false> (reference src/kernels.cu:398-461).
#pragma once

#include <type_traits>

#include "blend_planar.hpp"

namespace lfi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int P3_TPX = 128;               // pixels per tile
constexpr int P3_KC = 64;                 // images per chunk
constexpr int P3_CH_B = 8 * 1024 + 4 * 128; // one channel of a buffer: 8 octets of 1 KB, +128 B in front of every odd octet
constexpr int P3_BUF_B = 3 * P3_CH_B;     // 26,112 B
// byte offset of octet o (images 8o … 8o+7, 128 B each) inside a channel: a ds_read_b64 instruction is served in two groups of
// 32 lanes over 64 banks of 4 B (MI355X_MICROARCH.md §LDS); lanes of kg = l>>4 read octet 4ks + kg, 128 contiguous bytes per
// 16 lanes, so octets 4ks and 4ks+1 (and 4ks+2, 4ks+3) must start 128 B apart modulo 256
__host__ __device__ constexpr int p3_octet_off(int o)
{
    return o * 1024 + 128 * ((o + 1) >> 1);
}

#define LFI_P3_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// measurement builds (-DLFI_P3_TRACE=1, tools/p3_trace.sh): per workgroup, the clocks wave 0 spends in the parts of a unit, summed over its
// units (lfi_debug_p3_trace reads them back).  A stamp drains the wave's LDS operations first: it perturbs what it measures a little.
#ifndef LFI_P3_TRACE
#define LFI_P3_TRACE 0
#endif
#ifndef LFI_P3_INTERLEAVE
#define LFI_P3_INTERLEAVE 1
#endif
#if LFI_P3_TRACE
__device__ unsigned long long lfi_p3_trace_buf[1024 * 8];
#define LFI_P3_STAMP(var) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#endif

template <int N, int I = 0, typename F>
__device__ __forceinline__ void p3_for_each_chunk(F &&f)
{
    if constexpr(I < N)
    {
        f(std::integral_constant<int, I>{});
        p3_for_each_chunk<N, I + 1>(f);
    }
}

// planar views: plane (view, channel) at ((view·3 + channel)·rows)·pitch, pixel x of row y at y·pitch + x
// view_passes > 1 (more than 64 views; NCH == 1 only — the host splits other launches): the tile stays in LDS for every pass.  MP =
// the most passes a launch may have (1 or 4): the weight fragments of ALL passes stay in registers (8 per pass).  Round 2 fetched the
// next pass's fragments while the current one computed; the compiler, which cannot count the predicated stores issued after those
// loads, waited for them with s_waitcnt vmcnt(0) — a drain of the wave's stores AND its LDS-DMA prefetch once per pass
// (profiles/r03_notes.md §11: config 4 whole 1.90 → 1.69 ms).
// ABL (measurement builds only, LFI_P3_ABLATE): 0 = the kernel; 1 = no k-loop (DMA + barriers + stores of zeros); 2 = no DMA (the
// k-loop runs on whatever LDS holds); 3 = no stores.  Outputs of ABL != 0 are garbage by construction.
// VG: groups of 16 views per wave.  1: four waves per workgroup, two waves per SIMD (memory-bound launches: one chunk of images).
// 2: TWO waves per workgroup, 32 views each — the pixel operand of a block (LDS read + v_perm) is built once for two MFMAs, and the
// kernel needs more than 256 registers (192 accumulators), which pins one wave to each SIMD: two workgroups per CU as before, each
// wave alone on its SIMD.  For launches of several chunks, where the k-loop (not the DMA) sets the pace with four waves.
// RGBA_OUT (round 4): the views are RGBA planes [view][out_rows][W] — the reference's layout, the library's default — written from the same
// pipeline: a lane packs its eight pixels of a view into eight dwords (two v_perm per pixel, alpha = 255: uchar4{…, 255}, src/kernels.cu:393) and
// stores them with two adjacent 16-byte stores (the halves of a 32-byte sector in neighbouring instructions: profiles/r04_notes.md §13).
template <bool NT_STORE, int NCH, int ABL = 0, int VG = 1, int MP = 1, bool RGBA_OUT = false>
__global__ void __launch_bounds__(256 / VG, 2) blend_p3(const KernelArgs a, const int tiles_x, const int n_tiles, const int view_passes, const int reverse)
{
    static_assert(VG == 1 || VG == 2, "16 or 32 views per wave");
    static_assert(MP == 1 || NCH == 1, "several view passes: one chunk of images only");
    constexpr int NW = 4 / VG;  // waves per workgroup
    constexpr int OPW = 8 / NW; // octets of a chunk (and channel) each wave fetches
    __shared__ __attribute__((aligned(16))) uint8_t lds[3 * P3_BUF_B + LFI_MAX_IMAGES * 8];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, kg = lane >> 4;
    const int H = a.height;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane(uint32_t(uintptr_t((lds_ptr_t)lds)));
    const size_t shift_stride = (size_t)a.in_rows * a.planar_pitch; // one byte plane of the planar inputs
    int2 *off_table = reinterpret_cast<int2 *>(lds + 3 * P3_BUF_B);
    for(int g = threadIdx.x; g < a.n_images; g += 64 * NW)
    {
        const lfi_int2 o = a.focused[g];
        off_table[g] = make_int2(o.x + a.planar_phase[g], o.y); // the image's phase inside the planar copy folded into its x offset
    }

    // this wave's 16 views of a pass (views v0 + 64·pass + 16·wave …): all their weights, as MFMA A fragments (k-step s = images
    // 32s … 32s+31)
    const int vw0 = a.v0 + 16 * VG * wave;
    auto load_weights = [&](const int pass, half8 (&w_out)[VG][2 * NCH]) {
#pragma unroll
        for(int vg = 0; vg < VG; vg++)
#pragma unroll
            for(int s = 0; s < 2 * NCH; s++)
            {
                const int k = 32 * s + 8 * kg;
                u32x4 w = {0u, 0u, 0u, 0u};
                if(k < a.k_pad) // rows are k_pad halves long (a multiple of 16): nothing is read across a row's end
                    w = *reinterpret_cast<const u32x4 *>(a.w16s + (size_t)(vw0 + 16 * vg + 64 * pass + n) * a.k_pad + k);
                w_out[vg][s] = __builtin_bit_cast(half8, w);
            }
    };
    half8 wreg[VG][2 * NCH];
    load_weights(0, wreg);
    half8 wmore[MP > 1 ? MP - 1 : 1][VG][2 * NCH]; // passes 1 … MP − 1
    if constexpr(MP > 1)
    {
#pragma unroll
        for(int p = 1; p < MP; p++)
            load_weights(p < view_passes ? p : 0, wmore[p - 1]); // (a pass the launch does not have: any valid rows, never used)
    }
    // the loads above are the only vector loads the compiler knows about: make it wait for them HERE, before any LDS-DMA is in
    // flight, instead of with a vmcnt(0) in front of the first MFMA (which would also drain the pipeline's first three tiles)
#pragma unroll
    for(int vg = 0; vg < VG; vg++)
#pragma unroll
        for(int s = 0; s < 2 * NCH; s++)
        {
            asm volatile("" : "+v"(wreg[vg][s]));
            if constexpr(MP > 1)
            {
#pragma unroll
                for(int p = 1; p < MP; p++)
                    asm volatile("" : "+v"(wmore[p - 1][vg][s]));
            }
        }

    const int G = gridDim.x;
    const int t0 = int(tile_of_block(blockIdx.x, gridDim.x, a.flags));
    if(t0 >= n_tiles)
        return;
    __syncthreads(); // the offset table is complete

    // pixel bytes of one unit: piece p (1 KB) = channel p / 8, octet p % 8 of the chunk; wave w moves pieces w, w + 4, …, i.e.
    // octets w and w + 4 of every channel.  Addressing: a wave-uniform 64-bit base per octet and channel (SGPR pair: the plane of
    // the octet's first image) + a 32-bit per-lane byte offset (the lane's image inside the octet, its row and column) — no 64-bit
    // vector arithmetic in the loop.  Per octet a lane keeps its image's integer offsets and its image's distance from the base.
    struct Pieces
    {
        int ox[OPW], oy[OPW];
        uint32_t img_off[OPW]; // (this lane's image − the octet's first image) · 3 planes, in bytes (< 2^32: checked on the host)
        const uint8_t *sbase[OPW]; // the R plane of the octet's first image (wave-uniform; that image clamped to the last one)
    };
    auto lookup = [&](const int chunk) {
        Pieces pc;
#pragma unroll
        for(int o2 = 0; o2 < OPW; o2++)
        {
            const int octet = wave + NW * o2;
            const int g_base = min(P3_KC * chunk + 8 * octet, a.n_images - 1);
            const int dg = min(lane >> 3, a.n_images - 1 - g_base); // padded images (zero weights) re-read the last one
            const int2 o = off_table[g_base + dg];
            pc.ox[o2] = o.x;
            pc.oy[o2] = o.y;
            pc.img_off[o2] = uint32_t(dg) * 3u * uint32_t(shift_stride);
            pc.sbase[o2] = a.planar + (size_t)__builtin_amdgcn_readfirstlane(g_base) * 3 * shift_stride;
        }
        return pc;
    };
    // A cursor over this workgroup's tiles t0, t0 + G, …: the tile's row and column are stepped, not divided out per unit (odd launches walk
    // the image backwards: see launch_p3).  One integer division per cursor and kernel instead of one per unit and use.
    struct TileCursor
    {
        int ty, tx;
    };
    const int step_y = G / tiles_x, step_x = G - step_y * tiles_x;
    auto cursor_at = [&](const int t_seq) {
        const int t = reverse ? n_tiles - 1 - t_seq : t_seq;
        TileCursor c;
        c.ty = t / tiles_x;
        c.tx = t - c.ty * tiles_x;
        return c;
    };
    auto cursor_step = [&](TileCursor &c) { // t_seq += G
        if(reverse)
        {
            c.tx -= step_x;
            c.ty -= step_y;
            if(c.tx < 0)
            {
                c.tx += tiles_x;
                c.ty--;
            }
        }
        else
        {
            c.tx += step_x;
            c.ty += step_y;
            if(c.tx >= tiles_x)
            {
                c.tx -= tiles_x;
                c.ty++;
            }
        }
    };
    // one octet of images (three LDS-DMA instructions) of a unit's fetch; returns the number of DMA instructions issued by this wave (wave-uniform)
    auto issue_piece = [&](const TileCursor &tc, const int chunk, const int buf, const Pieces &pc, const int o2) {
        const int y = a.out_y0 + tc.ty;
        const int x0 = tc.tx * P3_TPX;
        const int kc = min(P3_KC, a.k_pad - P3_KC * chunk);
        const uint32_t dst = lds_base + uint32_t(buf) * P3_BUF_B;
        int count = 0;
        {
            const int octet = wave + NW * o2;
            if(8 * octet >= kc)
                return 0; // wave-uniform: the chunk is shorter (its length is a multiple of 16)
            // the run starts at pixel x0 + ox; the padding exceeds every offset
            const int sy = clampi(y + pc.oy[o2], 0, H - 1) - a.in_y0; // clamp in the full image, then index the held rows
            const int start = x0 + pc.ox[o2] + a.planar_padx; // any byte of the plane row (byte-aligned LDS-DMA: blend_planar.hpp)
            // sy·pitch with a full-rate 24-bit multiply (rows, pitch < 2^24; an octet's 24 planes < 2^32 bytes: checked on the host)
            const uint32_t voff = pc.img_off[o2] + __umul24(uint32_t(sy), uint32_t(a.planar_pitch)) + uint32_t(start) + 16u * uint32_t(lane & 7);
            const uint8_t *sbase = pc.sbase[o2];
            if constexpr(ABL != 2)
            {
#pragma unroll
                for(int ch = 0; ch < 3; ch++)
                    dma16_s(sbase + (size_t)ch * shift_stride, voff, dst + uint32_t(ch * P3_CH_B + p3_octet_off(octet)));
                count += 3;
            }
            else
                asm volatile("" ::"v"(voff), "s"(sbase), "s"(dst));
        }
        return count;
    };
    auto issue = [&](const TileCursor &tc, const int chunk, const int buf, const Pieces &pc) {
        int count = 0;
#pragma unroll
        for(int o2 = 0; o2 < OPW; o2++)
            count += issue_piece(tc, chunk, buf, pc, o2);
        return count;
    };

    f32x4 acc[VG][8][3]; // [view group][block = pixel 8n + blk][channel]: views 16·group + 4kg + i
    auto clear_acc = [&] {
#pragma unroll
        for(int vg = 0; vg < VG; vg++)
#pragma unroll
            for(int b = 0; b < 8; b++)
#pragma unroll
                for(int ch = 0; ch < 3; ch++)
                    acc[vg][b][ch] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    };
    clear_acc();

    // k-loop of one unit from buffer `buf`: wk = the chunk's two weight fragments
    const uint32_t lane_px = uint32_t(1024 * kg + 128 * ((kg + 1) >> 1) + 8 * n); // p3_octet_off(kg) + this lane's 8 pixels
    // `fresh` (one-chunk launches): the first k-step takes a zero C operand, so the accumulators are never cleared
    auto compute = [&](const half8 (&wk)[VG][2], const int buf, const int kc, auto fresh_tag, auto &&mid) {
        constexpr bool fresh = decltype(fresh_tag)::value;
        if constexpr(ABL == 1)
            return;
        const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
        const uint8_t *pb = lds + buf * P3_BUF_B + lane_px;
        // groups = (k-step, channel): eight ds_read_b64 (the lane's eight pixels of images 8kg … 8kg+7), then eight MFMAs.  The reads
        // of group i + 1 are issued BEFORE the MFMAs of group i (register double buffer, pinned by a scheduling barrier): left to
        // itself the compiler issues them one MFMA ahead of their use and every group start waits for LDS (tools/p3_ablate.py: the
        // k-loop alone was 0.7 of the kernel's time at 15×15 grids).
        const int n_groups = kc > 32 ? 6 : 3; // wave-uniform: a chunk of ≤ 32 images has one k-step
        u32x2 d[2][8];
        auto load_group = [&](const int grp, u32x2 (&dst)[8]) {
            const int ks = grp / 3, ch = grp - 3 * ks;
#pragma unroll
            for(int j = 0; j < 8; j++)
                dst[j] = *reinterpret_cast<const u32x2 *>(pb + ch * P3_CH_B + p3_octet_off(4 * ks) + 128 * j);
        };
        load_group(0, d[0]);
#pragma unroll
        for(int grp = 0; grp < 6; grp++)
        {
            if(grp >= n_groups)
                break;
            if(grp + 1 < 6 && grp + 1 < n_groups)
                load_group(grp + 1, d[(grp + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const int ks = grp / 3, ch = grp - 3 * ks;
            const u32x2(&dc)[8] = d[grp & 1];
#pragma unroll
            for(int b = 0; b < 8; b++)
            {
                u32x4 bf;
#pragma unroll
                for(int q = 0; q < 4; q++)
                {
                    const uint32_t lo = b < 4 ? dc[2 * q].x : dc[2 * q].y, hi = b < 4 ? dc[2 * q + 1].x : dc[2 * q + 1].y;
                    // [15:0] = byte (b & 3) of image 2q, [31:16] = the same byte of image 2q + 1: two fp16 subnormals
                    bf[q] = __builtin_amdgcn_perm(hi, lo, 0x0c000c00u | uint32_t(b & 3) | (uint32_t(4 + (b & 3)) << 16));
                }
#pragma unroll
                for(int vg = 0; vg < VG; vg++)
                    acc[vg][b][ch] =
                        __builtin_amdgcn_mfma_f32_16x16x32_f16(wk[vg][ks], __builtin_bit_cast(half8, bf), (ks == 0 && fresh) ? zero4 : acc[vg][b][ch], 0, 0, 0);
            }
            // behind the group's MFMAs (they run in the matrix pipe while the wave issues this): a slice of the next fetch's issue work
            __builtin_amdgcn_sched_barrier(0);
            mid(grp);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto no_mid = [](const int) {};

    // epilogue of tile t: quantise (acc = S·2^-9), pack eight pixels per (view, channel), store; returns the number of store
    // instructions issued (wave-uniform)
    const uint32_t plane_b = uint32_t(a.out_rows) * uint32_t(a.views_pitch); // bytes of one byte plane (< 2^26·… checked on the host)
    auto epilogue = [&](const f32x4 (&acc)[8][3], const TileCursor &tc, const int vw, const int nvalid) {
        const int ty = tc.ty; // row inside the output window
        const int x0 = tc.tx * P3_TPX;
        uint32_t hq[48]; // [(i·3 + channel)·4 + block pair]: two halves, 0x4000 | byte after the rounding-mode window
#pragma unroll
        for(int i = 0; i < 4; i++)
#pragma unroll
            for(int ch = 0; ch < 3; ch++)
#pragma unroll
                for(int p = 0; p < 4; p++)
                {
                    const float2_t f = {acc[2 * p][ch][i], acc[2 * p + 1][ch][i]};
                    hq[(i * 3 + ch) * 4 + p] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, half2_t)); // v_cvt_pk_f16_f32: RN-even
                }
        const uint32_t k255 = 0x37f837f8u; // 255·2^-9 twice
        const uint32_t two = 0x40004000u;  // 2.0 twice
#pragma unroll
        for(int half = 0; half < 2; half++)
        {
#define h(k) hq[24 * half + (k)]
            // saturate at 255, then + 2.0 under round-toward-zero leaves floor(S) in the low mantissa byte (epilogue_packed.hpp);
            // everything that depends on the fp16 rounding mode sits in ONE asm statement together with the two mode writes
            asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3\n\ts_nop 1\n\t" LFI_QA(0) LFI_QA(1) LFI_QA(2) LFI_QA(3) LFI_QA(4)
                             LFI_QA(5) LFI_QA(6) LFI_QA(7) LFI_QA(8) LFI_QA(9) LFI_QA(10) LFI_QA(11) LFI_QA(12) LFI_QA(13) LFI_QA(14)
                                 LFI_QA(15) LFI_QA(16) LFI_QA(17) LFI_QA(18) LFI_QA(19) LFI_QA(20) LFI_QA(21) LFI_QA(22) LFI_QA(23)
                         "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0\n\ts_nop 1"
                         : "+v"(h(0)), "+v"(h(1)), "+v"(h(2)), "+v"(h(3)), "+v"(h(4)), "+v"(h(5)), "+v"(h(6)), "+v"(h(7)), "+v"(h(8)),
                           "+v"(h(9)), "+v"(h(10)), "+v"(h(11)), "+v"(h(12)), "+v"(h(13)), "+v"(h(14)), "+v"(h(15)), "+v"(h(16)),
                           "+v"(h(17)), "+v"(h(18)), "+v"(h(19)), "+v"(h(20)), "+v"(h(21)), "+v"(h(22)), "+v"(h(23))
                         : "s"(k255), "s"(two));
#undef h
        }
        if constexpr(RGBA_OUT)
        {
            // wave-uniform 64-bit base (the tile's first pixel in the wave's first view) + a 32-bit per-lane offset (16 RGBA planes < 4 GB:
            // checked on the host); the plane size opaque per tile, so that the offsets are not computed once per kernel and kept in registers
            uint8_t *ubase = a.views + (((size_t)vw * a.out_rows + ty) * a.width + x0) * 4;
            uint32_t vplane = uint32_t(a.out_rows) * uint32_t(a.width) * 4u;
            asm volatile("" : "+s"(vplane));
            const bool full_x = x0 + 8 * n + 8 <= a.width;
            int n_st = 0;
#pragma unroll
            for(int i = 0; i < 4; i++)
            {
                if(i >= nvalid) // wave-uniform; otherwise lane (n = 0, kg = 0) is active below
                    continue;
                uint32_t rgba[8];
#pragma unroll
                for(int b = 0; b < 8; b++)
                {
                    const int pr = b >> 1; // the pixel's half of its pair: the byte sits in the low byte of that half
                    const uint32_t rg = __builtin_amdgcn_perm(hq[(i * 3 + 1) * 4 + pr], hq[(i * 3 + 0) * 4 + pr], (b & 1) ? 0x0c0c0602u : 0x0c0c0400u); // [R, G, 0, 0]
                    rgba[b] = __builtin_amdgcn_perm(hq[(i * 3 + 2) * 4 + pr], rg, (b & 1) ? 0x0d060100u : 0x0d040100u);                            // [R, G, B, 0xff]
                }
                const bool view_ok = 4 * kg + i < nvalid;
                uint32_t *out = reinterpret_cast<uint32_t *>(ubase + uint32_t(4 * kg + i) * vplane) + 8 * n;
                if constexpr(ABL == 3)
                {
                    asm volatile("" ::"v"(rgba[0]), "v"(rgba[7]), "v"(out));
                    continue;
                }
                if(full_x)
                {
                    if(view_ok)
                    {
                        const u32x4 lo4 = {rgba[0], rgba[1], rgba[2], rgba[3]}, hi4 = {rgba[4], rgba[5], rgba[6], rgba[7]};
                        if constexpr(NT_STORE)
                        {
                            __builtin_nontemporal_store(lo4, reinterpret_cast<u32x4_a4 *>(out));
                            __builtin_nontemporal_store(hi4, reinterpret_cast<u32x4_a4 *>(out + 4));
                        }
                        else
                        {
                            *reinterpret_cast<u32x4_a4 *>(out) = lo4;
                            *reinterpret_cast<u32x4_a4 *>(out + 4) = hi4;
                        }
                    }
                }
                else if(view_ok) // the ragged right edge of the image: pixel by pixel
                {
#pragma unroll
                    for(int b = 0; b < 8; b++)
                        if(x0 + 8 * n + b < a.width)
                            out[b] = rgba[b];
                }
                // counted for the vmcnt bookkeeping only where certainly issued (lane n = 0, kg = 0 takes the full-width branch); an
                // undercount only makes the next waits stricter
                if(x0 + 8 <= a.width)
                    n_st += 2;
            }
            return n_st;
        }
        // 16 bytes per lane and store (round 3): the MFMA leaves a lane 8 pixels of four views; neighbouring lanes (n, n ^ 1) swap — the even
        // lane takes both lanes' pixels of view i0, the odd lane both lanes' pixels of view i0 + 1 (two DPP moves per channel) — so that one
        // dwordx4 store per lane replaces two dwordx2 stores: half the store instructions, still whole 128-byte lines (eight lanes × 16
        // bytes per view and plane row).  The 8-byte stores were issue-bound where a launch writes more than it reads: 256 views from 64 images,
        // BASELINE config 4 on one GPU (profiles/r03_notes.md §10).
        // wave-uniform 64-bit base per store + one 32-bit per-lane byte offset (12 planes < 4 GB)
        uint8_t *ubase = a.views + ((size_t)vw * 3 * a.out_rows + ty) * a.views_pitch + x0;
        const bool odd = n & 1;
        const uint32_t lane_off = uint32_t(12 * kg + (odd ? 3 : 0)) * plane_b + uint32_t(8 * (n & ~1));
        const bool x_ok = x0 + 8 * (n & ~1) < a.views_pitch; // the pitch is a multiple of 128 ≥ W: a whole 16-byte store stays inside the row
        int n_st = 0;
#pragma unroll
        for(int ip = 0; ip < 2; ip++)
        {
            const int i0 = 2 * ip;
            if(i0 >= nvalid) // wave-uniform; otherwise lane (n = 0, kg = 0) is active and every store below is issued
                continue;
#pragma unroll
            for(int ch = 0; ch < 3; ch++)
            {
                const uint32_t *q0 = &hq[(i0 * 3 + ch) * 4], *q1 = &hq[((i0 + 1) * 3 + ch) * 4];
                const uint32_t a0 = __builtin_amdgcn_perm(q0[1], q0[0], 0x06040200u), a1 = __builtin_amdgcn_perm(q0[3], q0[2], 0x06040200u); // view i0
                const uint32_t b0 = __builtin_amdgcn_perm(q1[1], q1[0], 0x06040200u), b1 = __builtin_amdgcn_perm(q1[3], q1[2], 0x06040200u); // view i0 + 1
                // what the partner needs: the even lane gives away its pixels of view i0 + 1, the odd lane its pixels of view i0
                const uint32_t r0 = uint32_t(__builtin_amdgcn_mov_dpp(int(odd ? a0 : b0), 0xB1, 0xf, 0xf, false)); // quad_perm [1, 0, 3, 2]
                const uint32_t r1 = uint32_t(__builtin_amdgcn_mov_dpp(int(odd ? a1 : b1), 0xB1, 0xf, 0xf, false));
                const u32x4 px16 = {odd ? r0 : a0, odd ? r1 : a1, odd ? b0 : r0, odd ? b1 : r1}; // 16 pixels from 8(n & ~1) of this lane's view
                u32x4 *out = reinterpret_cast<u32x4 *>(ubase + (size_t)(3 * i0 + ch) * plane_b + lane_off);
                if constexpr(ABL == 3)
                {
                    asm volatile("" ::"v"(px16), "v"(out));
                    continue;
                }
                n_st++;
                if(x_ok && 4 * kg + i0 + (odd ? 1 : 0) < nvalid)
                {
                    if constexpr(NT_STORE)
                        __builtin_nontemporal_store(px16, out);
                    else
                        *out = px16;
                }
            }
        }
        return n_st;
    };

    // ---- the unit sequence of this workgroup: tiles t0, t0 + G, …; chunks 0 … NCH−1 of each -------------------------------------
    int it = t0, ic = 0; // issue cursor: the next unit to fetch
    TileCursor itc = cursor_at(t0);
    auto advance_issue = [&] {
        if(++ic == NCH)
        {
            ic = 0;
            it += G;
            cursor_step(itc);
        }
    };
    Pieces pc = lookup(0);
    issue(itc, ic, 0, pc);
    advance_issue();
    if constexpr(NCH > 1)
        pc = lookup(ic);
    bool have1 = it < n_tiles; // a unit after the current one exists (and is in flight)
    int nd1 = 0;               // its DMA instructions
    if(have1)
    {
        nd1 = issue(itc, ic, 1, pc);
        advance_issue();
        if constexpr(NCH > 1)
            pc = lookup(ic);
    }
    int buf = 0;
    TileCursor ctc = cursor_at(t0); // compute cursor (the chunk is the compile-time argument of `unit`)
#if LFI_P3_TRACE
    unsigned long long tr_wait = 0ull, tr_issue = 0ull, tr_kloop = 0ull, tr_drain = 0ull, tr_epi = 0ull, tr_units = 0ull, tr_begin;
    LFI_P3_STAMP(tr_begin);
#endif
    int st1 = 0, st2 = 0; // store instructions of the previous unit's epilogue and of the one before
    // One unit; the chunk index is a compile-time constant (the loop below is unrolled over the chunks of a tile), so the chunk's
    // weight fragments are registers named at compile time — a runtime index would put them into scratch, and wave-uniform selects
    // over all chunks cost a copy of the fragments.  Returns false after the last unit of this workgroup.
    auto unit = [&](auto cc_tag) -> bool {
        constexpr int cc = decltype(cc_tag)::value;
        // VMEM operations of this wave younger than the current unit's DMA: stores(u−2), DMA(u+1), stores(u−1) — they may stay in
        // flight; vmcnt retires in order, so "at most that many outstanding" means the current unit's pieces have landed
#if LFI_P3_TRACE
        unsigned long long tA, tB, tC, tD;
        LFI_P3_STAMP(tA);
#endif
        const int allowed = st2 + (have1 ? nd1 : 0) + st1;
        switch(min(allowed, 63) >> 2) // (VG 2: up to 24 + 12 + 24)
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
        __builtin_amdgcn_s_barrier(); // everybody's pieces of this unit have landed; everybody is done with the previous unit's buffer
        asm volatile("" ::: "memory");
#if LFI_P3_TRACE
        LFI_P3_STAMP(tB);
#endif
        const bool have2 = have1 && it < n_tiles;
        int nd2 = 0;
        // LFI_P3_INTERLEAVE (several chunks: one wave per SIMD, nothing else to fill its issue slots): the fetch is issued in slices BEHIND the
        // groups of the k-loop, whose MFMAs keep the matrix pipe busy meanwhile, instead of in front of it
        constexpr bool interleave = LFI_P3_INTERLEAVE && NCH > 1;
        const int buf_next = buf == 0 ? 2 : buf - 1; // (buf + 2) % 3: the buffer the previous unit used
        if(have2 && !interleave)
        {
            nd2 = issue(itc, ic, buf_next, pc);
            advance_issue();
            if constexpr(NCH > 1)
                pc = lookup(ic); // for the unit after that: off the critical path of the next barrier
        }
        auto fetch_slice = [&](const int grp) { // slices 0 … OPW − 1: an octet each; slice OPW: the cursor and the next lookup
            if(!have2)
                return;
            if(grp < OPW)
                nd2 += issue_piece(itc, ic, buf_next, pc, grp);
            else if(grp == OPW)
            {
                advance_issue();
                pc = lookup(ic);
            }
        };
        const int kc = min(P3_KC, a.k_pad - P3_KC * cc);
#if LFI_P3_TRACE
        LFI_P3_STAMP(tC);
        tr_wait += tB - tA;
        tr_issue += tC - tB;
        tr_units++;
#endif
        st2 = st1;
        st1 = 0;
        if constexpr(NCH == 1)
        {
            // every view pass of the tile from the same LDS buffer, its weight fragments from registers (no loads in this loop: see
            // the kernel's header)
            p3_for_each_chunk<MP>([&](auto pass_tag) {
                constexpr int pass = decltype(pass_tag)::value;
                if(pass >= view_passes) // uniform
                    return;
                const int vw = vw0 + 64 * pass;
                const int nvalid = __builtin_amdgcn_readfirstlane(min(a.v1 - vw, 16 * VG)); // ≤ 0: nothing to do for this wave
                if(nvalid > 0)
                {
                    half8 wk[VG][2];
#pragma unroll
                    for(int vg = 0; vg < VG; vg++)
                    {
                        wk[vg][0] = pass == 0 ? wreg[vg][0] : wmore[pass > 0 ? pass - 1 : 0][vg][0];
                        wk[vg][1] = pass == 0 ? wreg[vg][1] : wmore[pass > 0 ? pass - 1 : 0][vg][1];
                    }
                    compute(wk, buf, kc, std::true_type{}, no_mid);
#if LFI_P3_TRACE
                    LFI_P3_STAMP(tD);
                    tr_kloop += tD - tC;
#endif
                    // Single-pass launches: all of this wave's fetches (the two units in flight) land before its stores go out.
                    // Measured, not designed — round 2's kernel held such a wait by accident (the compiler's, for pass-weight loads
                    // that a one-pass launch never issues); without it config 2 runs 2–3 % and one rank of config 4 15 % slower, placed
                    // in front of the k-loop 1 % and 4 % slower (profiles/r03_p3_drain_ab*.txt).  The two workgroups of a CU fall into
                    // step: one streams its tiles in while the other computes and writes.  Launches of several passes or several
                    // chunks are slower with it.
                    if constexpr(MP == 1)
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if LFI_P3_TRACE
                    {
                        unsigned long long tE;
                        LFI_P3_STAMP(tE);
                        tr_drain += tE - tD;
                        tD = tE;
                    }
#endif
#pragma unroll
                    for(int vg = 0; vg < VG; vg++)
                        if(nvalid > 16 * vg) // wave-uniform
                            st1 += epilogue(acc[vg], ctc, vw + 16 * vg, min(nvalid - 16 * vg, 16));
#if LFI_P3_TRACE
                    LFI_P3_STAMP(tC);
                    tr_epi += tC - tD;
#endif
                }
            });
        }
        else
        {
            const int nvalid = __builtin_amdgcn_readfirstlane(min(a.v1 - vw0, 16 * VG)); // ≤ 0: this wave only helps with the DMA
            if constexpr(interleave)
                if(nvalid <= 0)
                {
#pragma unroll
                    for(int g = 0; g <= OPW; g++)
                        fetch_slice(g);
                }
            if(nvalid > 0)
            {
                half8 wk[VG][2];
#pragma unroll
                for(int vg = 0; vg < VG; vg++)
                {
                    wk[vg][0] = wreg[vg][2 * cc];
                    wk[vg][1] = wreg[vg][2 * cc + 1];
                }
                if constexpr(interleave)
                {
                    compute(wk, buf, kc, std::integral_constant<bool, cc == 0>{}, fetch_slice);
                    const int done = kc > 32 ? 6 : 3; // the groups the k-loop had: the slices it did not reach
#pragma unroll
                    for(int g = 3; g <= OPW; g++)
                        if(g >= done)
                            fetch_slice(g);
                }
                else
                    compute(wk, buf, kc, std::integral_constant<bool, cc == 0>{}, no_mid); // chunk 0 starts from a zero C operand: no clears
#if LFI_P3_TRACE
                LFI_P3_STAMP(tD);
                tr_kloop += tD - tC;
#endif
                if constexpr(cc == NCH - 1)
                {
#pragma unroll
                    for(int vg = 0; vg < VG; vg++)
                        if(nvalid > 16 * vg) // wave-uniform
                            st1 += epilogue(acc[vg], ctc, vw0 + 16 * vg, min(nvalid - 16 * vg, 16));
#if LFI_P3_TRACE
                    LFI_P3_STAMP(tC);
                    tr_epi += tC - tD;
#endif
                }
            }
        }
        if(!have1)
            return false;
        if constexpr(cc == NCH - 1)
            cursor_step(ctc);
        buf = buf == 2 ? 0 : buf + 1;
        have1 = have2;
        nd1 = nd2;
        return true;
    };
    bool more = true;
    while(more)
        p3_for_each_chunk<NCH>([&](auto cc_tag) {
            if(more)
                more = unit(cc_tag);
        });
#if LFI_P3_TRACE
    if(threadIdx.x == 0 && blockIdx.x < 1024)
    {
        unsigned long long t_end;
        LFI_P3_STAMP(t_end);
        unsigned long long *o = lfi_p3_trace_buf + blockIdx.x * 8;
        o[0] = tr_wait, o[1] = tr_issue, o[2] = tr_kloop, o[3] = tr_drain, o[4] = tr_epi, o[5] = tr_units, o[6] = t_end - tr_begin, o[7] = gridDim.x;
    }
#endif
}

// ---- layout conversions for the planar view layout ------------------------------------------------------------------------------

// RGBA planes [views][rows][W] dwords → byte planes [views][3][rows][pitch]: for renders that only the RGBA kernels serve
// (STD, all-focus, debug modes) when the context's views are planar.  grid (ceil(pitch/4 / 256), rows, views)
__global__ void __launch_bounds__(256) views_rgba_to_planar(const uint32_t *__restrict__ rgba, uint8_t *__restrict__ planar, const int W, const int rows,
                                                            const int pitch)
{
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if(x4 >= pitch)
        return;
    const int y = blockIdx.y, v = blockIdx.z;
    const uint32_t *row = rgba + ((size_t)v * rows + y) * W;
    uint32_t px[4];
#pragma unroll
    for(int i = 0; i < 4; i++)
        px[i] = x4 + i < W ? row[x4 + i] : 0u;
#pragma unroll
    for(int ch = 0; ch < 3; ch++)
    {
        const uint32_t lo = __builtin_amdgcn_perm(px[1], px[0], 0x0c0c0400u + 0x0101u * uint32_t(ch)); // [p0.ch, p1.ch, 0, 0]
        const uint32_t hi = __builtin_amdgcn_perm(px[3], px[2], 0x0c0c0400u + 0x0101u * uint32_t(ch));
        *reinterpret_cast<uint32_t *>(planar + (((size_t)v * 3 + ch) * rows + y) * pitch + x4) = lo | (hi << 16);
    }
}

// one view: byte planes → an RGBA plane with alpha 255 (uchar4{…, 255}, reference src/kernels.cu:393); grid (ceil(W/4 / 256), rows)
__global__ void __launch_bounds__(256) view_planar_to_rgba(const uint8_t *__restrict__ planar_view, uint32_t *__restrict__ rgba, const int W, const int rows,
                                                           const int pitch)
{
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if(x4 >= W)
        return;
    const int y = blockIdx.y;
    uint32_t c[3];
#pragma unroll
    for(int ch = 0; ch < 3; ch++)
        c[ch] = *reinterpret_cast<const uint32_t *>(planar_view + ((size_t)ch * rows + y) * pitch + x4); // pitch is a multiple of 16
#pragma unroll
    for(int i = 0; i < 4; i++)
        if(x4 + i < W)
            rgba[(size_t)y * W + x4 + i] = ((c[0] >> (8 * i)) & 0xffu) | (((c[1] >> (8 * i)) & 0xffu) << 8) | (((c[2] >> (8 * i)) & 0xffu) << 16) | 0xff000000u;
}

// The quilt (scripts/viewsToQuilt.sh: `montage -tile 5x9` of the NN.png files) assembled on the device: views v0 … v0 + n − 1 of the context
// become tiles first … first + n − 1 (row-major) of a quilt of tiles_x columns; `quilt` is the RGBA image of the quilt ROWS OF TILES that
// these tiles touch (tile row first / tiles_x is its row 0), qw = tiles_x·W pixels wide.  PLANAR: the views are byte planes
// [view][R,G,B][rows][pitch] and are expanded on the fly (alpha 255, uchar4{…, 255}: src/kernels.cu:393), else RGBA planes.
// grid (ceil(W/4 / 256), rows, n); a lane moves four pixels.
template <bool PLANAR>
__global__ void __launch_bounds__(256) quilt_assemble(const uint8_t *__restrict__ views, uint32_t *__restrict__ quilt, const int W, const int rows, const int pitch,
                                                      const size_t view_stride, const int v0, const int first, const int tiles_x)
{
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if(x4 >= W)
        return;
    const int y = blockIdx.y, i = blockIdx.z;
    const int tile = first + i, trow = tile / tiles_x - first / tiles_x, tcol = tile % tiles_x;
    const uint8_t *view = views + (size_t)(v0 + i) * view_stride;
    uint32_t px[4];
    if constexpr(PLANAR)
    {
        uint32_t c[3];
#pragma unroll
        for(int ch = 0; ch < 3; ch++)
            c[ch] = *reinterpret_cast<const uint32_t *>(view + ((size_t)ch * rows + y) * pitch + x4); // pitch is a multiple of 128
#pragma unroll
        for(int k = 0; k < 4; k++)
            px[k] = ((c[0] >> (8 * k)) & 0xffu) | (((c[1] >> (8 * k)) & 0xffu) << 8) | (((c[2] >> (8 * k)) & 0xffu) << 16) | 0xff000000u;
    }
    else
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(view) + (size_t)y * W + x4;
#pragma unroll
        for(int k = 0; k < 4; k++)
            px[k] = x4 + k < W ? src[k] : 0u;
    }
    uint32_t *dst = quilt + ((size_t)trow * rows + y) * ((size_t)tiles_x * W) + (size_t)tcol * W + x4;
#pragma unroll
    for(int k = 0; k < 4; k++)
        if(x4 + k < W)
            dst[k] = px[k];
}

} // namespace lfi
