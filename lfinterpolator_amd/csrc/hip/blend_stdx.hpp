// blend_stdx.hpp — STD (the exact fp32 fmaf chain of Kernels::Standard::process, reference src/kernels.cu:292-342) for light fields of
// MORE than 64 images (15×15 grids: four chunks of images) at matrix-core speed: the band method of blend_planar<STDF>, extended over
// chunks.
//
// blend_planar<STDF> decides most bytes from the fp16-MFMA sum and recomputes the sums that land within a band around x.5 with the
// chain itself — from the pixel bytes of ALL images of the tile, which must still be in LDS.  With 225 images a tile's stack is 86 KB:
// it does not stay (round 2 therefore ran the exact-fp32 MFMA kernel on 15×15 grids: 6.9 ms at config 5, fp32-pipe-bound).
// Here the chain's bytes come back a second time instead:
//
//   per tile, units  M(NCH−1) … M(1)   fp16 MFMA k-loops of blend_p3 over the chunks in DESCENDING order (the sum's order is free)
//                    MC(0)             the last k-loop, on chunk 0; epilogue: band test on all sums, RGBA stores of the rounded bytes,
//                                      the sums inside the band queued per wave (≤ 128; more take a slow path); the chain's first 64
//                                      images for the queued sums — chunk 0 is the one buffer still resident
//                    C(1) … C(NCH−1)   the chain continued in ascending image order over the same chunks: chunks 1 and 2 are still in the
//                                      buffers M(1) and M(2) left (sx_buffer below); chunk 3 is fetched AGAIN through the same LDS-DMA
//                                      ring (it left this workgroup's L2 / the Infinity Cache microseconds ago).  After the last unit
//                                      the chain's bytes are patched over the rounded ones (byte stores by the wave that wrote the
//                                      dwords: same-wave stores to one address retire in order).
//
// so HBM sees the inputs once; the second fetch is cache traffic.  Pipeline (persistent workgroups, ring of three buffers two units
// ahead, one barrier and one hand-counted vmcnt wait per unit), operand maps, DMA addressing: blend_p3.hpp, four waves of 16 views.
//
// The band.  acc = S̃·2^-9 is the MFMA estimate of the exact sum S (weights ×2^15, pixel bytes as fp16 subnormals).
//   |S̃ − S| ≤ N·2^-17   accumulation error of the matrix pipe on sums below 512: MEASURED on gfx950 and asserted by
//                        tests/test_gpu_parity.py::test_mfma_f16_accumulation_error_bound (LFI_FLAG_STD_ANALYTIC_BAND: N·2^-15, analytic);
//   |s_N − S| ≤ N·½ulp(s_N)   the chain: N roundings to nearest of non-negative partial sums that only grow (weights and pixels are
//                        non-negative, rounding is monotone), so every partial sum is ≤ s_N and every rounding error ≤ half an ulp of s_N.
//                        The binade is taken from S̃ + (the widest band), which is ≥ s_N.  (blend_planar<STDF> budgets N·2^-16, half an
//                        ulp below 512, for every sum; at S ≈ 127 this is four times narrower — the queue is what costs time here.)
// A sum farther than the two together (+ 2^-12) from every half-integer rounds to the chain's byte; the others are recomputed.
// Bit-exact against the oracle like every STD kernel (same tests).  Preconditions (host): weights finite, in [0, 2), every view's
// weights sum to at most 2; the planar input copy.
#pragma once

#include "blend_p3.hpp"

namespace lfi {

constexpr int SX_QCAP = 128; // queued (pixel, view, channel) sums per wave and tile: two per lane
// measurement builds only (hipcc -DLFI_SX_ABL=n, tools/stdx_ablate.sh): 1 no chain arithmetic (the C units still fetch), 2 no byte
// patches, 3 no second fetch of the last chunk, 4 no band test (nothing queued), 5 no RGBA stores, 6 every tile's RGBA stores into the first
// tile's bytes (stores that stay in the L2); the output is wrong by construction.  (Rounds of ablations without the C units or the band test: profiles/r03_stdx_ablation_*.txt, an earlier form of the kernel.)
#ifndef LFI_SX_ABL
#define LFI_SX_ABL 0
#endif
#ifndef LFI_SX_PHASE
#define LFI_SX_PHASE 0
#endif
#ifndef LFI_SX_NT
#define LFI_SX_NT 1 // 0: ordinary stores instead of nontemporal ones (measurement)
#endif
// measurement builds (-DLFI_SX_TRACE=1, tools/stdx_trace.sh): per workgroup and unit slot, the clocks wave 0 spent in the wait + barrier and in
// the unit's work, summed over the tiles (lfi_debug_sx_trace reads them back)
#ifndef LFI_SX_TRACE
#define LFI_SX_TRACE 0
#endif
#if LFI_SX_TRACE
__device__ unsigned long long lfi_sx_trace_buf[1024 * 32];
#endif

// The ring buffer of unit `sl` (0 … 2·NCH − 2: M(NCH−1) … M(1), MC(0), C(1) … C(NCH−1)) of a workgroup's j-th tile.  Fetches run two units
// ahead, each into the buffer of the unit that finished last — except that C(1) and C(2) fetch NOTHING: chunks 1 and 2 of the tile are what
// M(1) and M(2) left in their buffers, which the assignment below leaves untouched until C(1) and C(2) are done (the all-focus kernel's
// observation, blend_stdxa.hpp, carried over to a ring of three).  With four chunks only chunk 3 — overwritten by MC(0)'s chunk 0 — comes
// back a second time; with two or three chunks nothing does.  Written out, the assignment repeats with period two tiles, the second tile
// mirrored (2 − buffer):
template <int NCH>
__host__ __device__ constexpr int sx_buffer(const int j, const int sl)
{
    constexpr int t2[3] = {0, 1, 0}, t3[5] = {0, 1, 2, 1, 0}, t4[7] = {0, 1, 2, 0, 2, 1, 0};
    if(NCH == 1)
        return j % 3; // one unit per tile (MC(0): k-loop, band test, the whole chain from the buffer it just used): a plain ring of three
    const int v = NCH == 2 ? t2[sl] : (NCH == 3 ? t3[sl] : t4[sl]);
    return (j & 1) ? 2 - v : v;
}

// PLANAR_OUT: the views are alpha-free byte planes [view][R,G,B][out_rows][views_pitch] (the library's planar layout, blend_p3.hpp) and are
// written directly — eight bytes per lane, view and channel; a tile row of a plane is one 128-byte line — instead of RGBA planes (round 4: the
// planar layout used to send STD through an RGBA scratch copy of all views and a conversion pass).
template <bool NT_STORE, int NCH, bool PLANAR_OUT = false>
__global__ void __launch_bounds__(256, 2) blend_stdx(const KernelArgs a, const int tiles_x, const int n_tiles, const int reverse)
{
    static_assert(NCH >= 1 && NCH <= 4, "one to four chunks of 64 images");
    constexpr int NW = 4, OPW = 2;
    constexpr int NU = 2 * NCH - 1; // unit slots per iteration
    constexpr int QUEUE_OFF = 3 * P3_BUF_B + LFI_MAX_IMAGES * 8;
    __shared__ __attribute__((aligned(16))) uint8_t lds[QUEUE_OFF + NW * SX_QCAP * 2];
    static_assert(sizeof(lds) <= 81920, "two workgroups per CU");

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, kg = lane >> 4;
    const int W = a.width, H = a.height;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane(uint32_t(uintptr_t((lds_ptr_t)lds)));
    const size_t shift_stride = (size_t)a.in_rows * a.planar_pitch; // one byte plane of the planar inputs
    int2 *off_table = reinterpret_cast<int2 *>(lds + 3 * P3_BUF_B);
    for(int g = threadIdx.x; g < a.n_images; g += 64 * NW)
    {
        const lfi_int2 o = a.focused[g];
        off_table[g] = make_int2(o.x + a.planar_phase[g], o.y); // the image's phase inside the planar copy folded into its x offset
    }
    uint16_t *queue = reinterpret_cast<uint16_t *>(lds + QUEUE_OFF) + wave * SX_QCAP;

    // this wave's 16 views (v0 + 16·wave …): all their weights ×2^15 as MFMA A fragments (k-step s = images 32s … 32s+31); lane l holds
    // view l&15, images 32s + 8(l>>4) + j — the chain reads them back through ds_bpermute
    const int vw0 = a.v0 + 16 * wave;
    half8 wreg[2 * NCH];
#pragma unroll
    for(int s = 0; s < 2 * NCH; s++)
    {
        const int k = 32 * s + 8 * kg;
        u32x4 w = {0u, 0u, 0u, 0u};
        if(k < a.k_pad) // rows are k_pad halves long (a multiple of 16): nothing is read across a row's end
            w = *reinterpret_cast<const u32x4 *>(a.w16s + (size_t)(vw0 + n) * a.k_pad + k);
        wreg[s] = __builtin_bit_cast(half8, w);
    }
    // the compiler's wait for these loads belongs HERE, before any LDS-DMA is in flight (blend_p3.hpp)
#pragma unroll
    for(int s = 0; s < 2 * NCH; s++)
        asm volatile("" : "+v"(wreg[s]));

#if LFI_SX_TRACE
    __shared__ unsigned long long tr[32];
    if(threadIdx.x < 32)
        tr[threadIdx.x] = 0ull;
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    const int G = gridDim.x;
    const int t0 = int(tile_of_block(blockIdx.x, gridDim.x, a.flags));
    if(t0 >= n_tiles)
        return;
    __syncthreads(); // the offset table is complete
#if LFI_SX_PHASE
    // measurement builds: the second workgroup of every CU (dispatch order: XCD = id % 8, CU = (id / 8) % 32) starts late, so that the
    // two workgroups of a CU are not in the same unit of their tiles (≥ 16: every other workgroup of an XCD instead)
    if((blockIdx.x >> 3) & (LFI_SX_PHASE >= 16 ? 1 : 32))
        for(int i = 0; i < (LFI_SX_PHASE & 15); i++)
            __builtin_amdgcn_s_sleep(127);
#endif

    // ---- LDS-DMA of one unit: blend_p3's pieces (wave w moves octets w and w + 4 of every channel) ----------------------------------
    struct Pieces
    {
        int ox[OPW], oy[OPW];
        uint32_t img_off[OPW];
        int g_base[OPW];
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
            pc.g_base[o2] = g_base;
        }
        return pc;
    };
    auto tile_xy = [&](const int t_seq, int &ty, int &x0) {
        const int t = reverse ? n_tiles - 1 - t_seq : t_seq;
        ty = t / tiles_x; // row inside the output window
        x0 = (t - ty * tiles_x) * P3_TPX;
    };
    auto issue = [&](const int t_seq, const int chunk, const int buf, const Pieces &pc) {
        int ty, x0;
        tile_xy(t_seq, ty, x0);
        const int y = a.out_y0 + ty;
        const int kc = min(P3_KC, a.k_pad - P3_KC * chunk);
        const uint32_t dst = lds_base + uint32_t(buf) * P3_BUF_B;
        int count = 0;
#pragma unroll
        for(int o2 = 0; o2 < OPW; o2++)
        {
            const int octet = wave + NW * o2;
            if(8 * octet >= kc)
                continue; // wave-uniform: the chunk is shorter (its length is a multiple of 16)
            const int sy = clampi(y + pc.oy[o2], 0, H - 1) - a.in_y0;
            const int start = x0 + pc.ox[o2] + a.planar_padx; // any byte of the plane row (byte-aligned LDS-DMA: blend_planar.hpp)
            // sy·pitch with a full-rate 24-bit multiply (rows, pitch < 2^24; an octet's 24 planes < 2^32 bytes: checked on the host)
            const uint32_t voff = pc.img_off[o2] + __umul24(uint32_t(sy), uint32_t(a.planar_pitch)) + uint32_t(start) + 16u * uint32_t(lane & 7);
            const uint8_t *sbase = a.planar + (size_t)__builtin_amdgcn_readfirstlane(pc.g_base[o2]) * 3 * shift_stride;
#pragma unroll
            for(int ch = 0; ch < 3; ch++)
                dma16_s(sbase + (size_t)ch * shift_stride, voff, dst + uint32_t(ch * P3_CH_B + p3_octet_off(octet)));
            count += 3;
        }
        return count;
    };

    // ---- the MFMA k-loop of one unit (blend_p3's, 16 views per wave) ----------------------------------------------------------------
    f32x4 acc[8][3]; // [block = pixel 8n + blk][channel]: views 4kg + i, acc = S̃·2^-9
    const uint32_t lane_px = uint32_t(1024 * kg + 128 * ((kg + 1) >> 1) + 8 * n); // p3_octet_off(kg) + this lane's 8 pixels
    auto compute = [&](const half8 (&wk)[2], const int buf, const int kc, auto fresh_tag) {
        constexpr bool fresh = decltype(fresh_tag)::value;
        const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
        const uint8_t *pb = lds + buf * P3_BUF_B + lane_px;
        const int n_groups = kc > 32 ? 6 : 3;
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
                    bf[q] = __builtin_amdgcn_perm(hi, lo, 0x0c000c00u | uint32_t(b & 3) | (uint32_t(4 + (b & 3)) << 16));
                }
                acc[b][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wk[ks], __builtin_bit_cast(half8, bf), (ks == 0 && fresh) ? zero4 : acc[b][ch], 0, 0, 0);
            }
        }
    };

    // ---- band constants (file header) ------------------------------------------------------------------------------------------------
    const float nf = float(a.n_images);
    const float c0 = nf * std_accumulation_bound(a.flags) + 0x1p-12f; // accumulation error + margin, in units of S
    const float bmax_acc = (c0 + nf * 0x1p-16f) * 0x1p-9f;  // the widest band (sums below 512), in units of acc
    const float base_acc = (0.5f - c0) * 0x1p-9f;           // inside(acc) = base − N·2^-24·pow2(acc + bmax): |d| above it ⇒ recompute
    const float chain_acc = nf * 0x1p-24f;

    // ---- the chain over one chunk for this lane's queued sums, from buffer `buf` (images ascending: src/kernels.cu:328-338) ----------
    // A queued sum = (pixel of the tile, view of the wave, channel) packed as px | v16 << 7 | ch << 11, and its running value s15 =
    // 2^-9 · (the reference's running sum; weights ×2^15, bytes ×2^-24): a power-of-two scaling commutes with every rounding (no partial
    // sum leaves the normal range: the smallest non-zero product is 2^-48).
    // weights: the wave's A fragments, fetched across lanes (view v16's images 32ks + 8kq + j sit in lane v16 + 16kq, dword j/2).
    // All LDS traffic of a k-step — 32 pixel bytes, 16 weight pairs — is issued BEFORE its first fma (staging registers + a scheduling
    // barrier): written as read → convert → fma per image the compiler emitted exactly that, one LDS latency per image (0.6 ms of
    // a 2.9 ms config-5 launch, tools/stdx_ablate.sh).
    auto pack_entry = [](const uint32_t code) { // queue code (lane | bit << 6 | i << 11) → px | v16 << 7 | ch << 11
        const uint32_t src = code & 63u, bit = (code >> 6) & 31u, i = code >> 11;
        const uint32_t b = (bit * 11u) >> 5; // bit / 3 for bit < 24
        return (8u * (src & 15u) + b) | ((4u * (src >> 4) + i) << 7) | ((bit - 3u * b) << 11);
    };
    auto chain = [&](auto cc_tag, const int buf, const int kc, const uint32_t entry, float &s15) {
        constexpr int cc = decltype(cc_tag)::value;
        const uint32_t px = entry & 127u, v16 = (entry >> 7) & 15u, ch = entry >> 11;
        const uint8_t *pb = lds + buf * P3_BUF_B + ch * P3_CH_B + px;
        float s = s15;
#pragma unroll
        for(int ks = 0; ks < 2; ks++)
        {
            if(32 * ks >= kc) // wave-uniform: a chunk of ≤ 32 images has one k-step
                break;
            uint32_t w2[16], pbyte[32]; // staged per k-step (48 registers: a whole chunk's 96 spilled in the MC unit)
            const u32x4 wv = __builtin_bit_cast(u32x4, wreg[2 * cc + ks]);
#pragma unroll
            for(int kq = 0; kq < 4; kq++)
            {
                const int src_lane = int(v16) + 16 * kq;
#pragma unroll
                for(int jp = 0; jp < 4; jp++)
                    w2[4 * kq + jp] = uint32_t(__builtin_amdgcn_ds_bpermute(4 * src_lane, int(wv[jp])));
#pragma unroll
                for(int j = 0; j < 8; j++)
                    pbyte[8 * kq + j] = pb[p3_octet_off(4 * ks + kq) + 128 * j];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for(int kk = 0; kk < 32; kk++) // image 64·cc + 32·ks + kk, ascending
            {
                const uint32_t pair = w2[kk >> 1];
                const float w = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(kk & 1 ? pair >> 16 : pair)));
                // the byte AS an fp16 subnormal (= byte·2^-24, what the MFMA's B operand holds too): v_fma_mix_f32 widens both halves
                // itself, no v_cvt_f32_ubyte
                const float pf = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(pbyte[kk])));
                s = __builtin_fmaf(pf, w, s); // addWeighted, src/kernels.cu:292-299
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        s15 = s;
    };
    // (unsigned char)__float2int_rn(sum) (uch4, src/kernels.cu:301-310): + 2^23 rounds to nearest-even and leaves the integer in the low bits
    auto byte_of15 = [](const float s15) { return __builtin_bit_cast(uint32_t, s15 * 0x1p-15f + 8388608.0f) & 0xffu; }; // the spill path's scale
    auto byte_of = [](const float s9) { return __builtin_bit_cast(uint32_t, s9 * 0x1p9f + 8388608.0f) & 0xffu; };

    // the byte of (view vw0 + v16, pixel px of the tile at row ty / column x0, channel ch) in the views
    // = a wave-uniform 64-bit base (the tile's first byte in the wave's first plane) + a 32-bit offset (a wave's 16 views are at most 48 byte
    // planes / 16 RGBA planes: below 4 GB, checked on the host)
    const uint32_t plane_b = uint32_t(a.out_rows) * uint32_t(PLANAR_OUT ? a.views_pitch : 4 * a.width); // bytes of a byte plane / of an RGBA plane
    auto out_base = [&](const int ty, const int x0) -> uint8_t * {
        if constexpr(PLANAR_OUT)
            return a.views + ((size_t)vw0 * 3 * a.out_rows + ty) * a.views_pitch + x0;
        else
            return a.views + (((size_t)vw0 * a.out_rows + ty) * a.width + x0) * 4;
    };
    auto out_off = [&](const uint32_t plane_bytes, const uint32_t v16, const uint32_t px, const uint32_t ch) -> uint32_t {
        if constexpr(PLANAR_OUT)
            return (3u * v16 + ch) * plane_bytes + px;
        else
            return v16 * plane_bytes + 4u * px + ch;
    };

    // ---- the unit sequence of this workgroup: tiles t0, t0 + G, … (T of them), 2·NCH − 1 units each ------------------------------------
    // (The C units of a tile interleaved with the next tile's M units, so that no fetch is issued with two short units of lead, measured
    // SLOWER — 3.05 against 2.93 ms at config 5, profiles/r03_stdx_interleave_ab.txt: the second fetch is traffic, not latency.)
    const int T = (n_tiles - 1 - t0) / G + 1;
    auto slot_valid = [&](const int j, const int sl) { return j < T && sl < NU; };
    auto slot_chunk = [](const int sl) { return sl < NCH ? NCH - 1 - sl : sl - NCH + 1; };
    // C(1) and C(2) read the buffers M(1) and M(2) left: nothing was fetched into them in between (sx_buffer: only C(3)'s chunk was
    // overwritten, by MC(0)'s)
    auto slot_fetches = [](const int sl) { return sl < NCH || (sl > NCH + 1 && LFI_SX_ABL != 3); };
    int ij = 0, isl = 0; // issue cursor: the next unit to fetch; past the end when ij ≥ T
    auto advance_issue = [&] {
        do
        {
            if(++isl == NU)
            {
                isl = 0;
                ij++;
            }
        } while(ij < T && !slot_valid(ij, isl));
    };
    // fetch the unit under the cursor (if it fetches at all); returns this wave's DMA instructions
    auto issue_cursor = [&](const Pieces &pc) { return slot_fetches(isl) ? issue(t0 + ij * G, slot_chunk(isl), sx_buffer<NCH>(ij, isl), pc) : 0; };
    Pieces pc = lookup(slot_chunk(0));
    issue_cursor(pc);
    advance_issue();
    bool have1 = ij < T; // a unit after the current one exists (and is in flight, if it fetches)
    int nd1 = 0;
    if(have1)
    {
        pc = lookup(slot_chunk(isl));
        nd1 = issue_cursor(pc);
        advance_issue();
    }
    if(ij < T)
        pc = lookup(slot_chunk(isl));
    int cj = 0; // compute cursor: the tile (the unit is the compile-time argument of `unit`)
    int st1 = 0, st2 = 0;
    const int nvalid = __builtin_amdgcn_readfirstlane(min(a.v1 - vw0, 16)); // ≤ 0: this wave only helps with the DMA
    // the queued sums of the tile whose MC unit ran last: two per lane
    uint32_t entry0 = 0u, entry1 = 0u;
    float s0 = 0.0f, s1 = 0.0f;
    int queued = 0; // wave-uniform

    auto unit = [&](auto sl_tag) -> bool {
        constexpr int sl = decltype(sl_tag)::value;
        constexpr bool is_c = sl >= NCH;
        constexpr bool is_mc = sl == NCH - 1;
        constexpr int cc = sl < NCH ? NCH - 1 - sl : sl - NCH + 1;
        constexpr bool is_last_c = sl == NU - 1;
        const int buf = sx_buffer<NCH>(cj, sl);
#if LFI_SX_TRACE
        const unsigned long long tA = __builtin_amdgcn_s_memtime();
#endif
        const int allowed = st2 + (have1 ? nd1 : 0) + st1;
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
        __builtin_amdgcn_s_barrier(); // everybody's pieces of this unit have landed; everybody is done with the previous unit's buffer
        asm volatile("" ::: "memory");
#if LFI_SX_TRACE
        const unsigned long long tB = __builtin_amdgcn_s_memtime();
#endif
        const bool have2 = have1 && ij < T;
        int nd2 = 0;
        if(have2)
        {
            nd2 = issue_cursor(pc); // into the buffer of a unit that has finished (sx_buffer)
            advance_issue();
            if(ij < T)
                pc = lookup(slot_chunk(isl)); // for the unit after that: off the critical path of the next barrier
        }
        const int kc = min(P3_KC, a.k_pad - P3_KC * cc);
        st2 = st1;
        st1 = 0;
        if(nvalid > 0)
        {
            int ty, x0;
            tile_xy(t0 + cj * G, ty, x0);
            if constexpr(!is_c)
            {
                half8 wk[2] = {wreg[2 * cc], wreg[2 * cc + 1]};
                compute(wk, buf, kc, std::integral_constant<bool, sl == 0>{}); // the first chunk of a tile starts from a zero C operand
            }
            if constexpr(is_mc)
            {
                // ---- epilogue: round every sum, store RGBA, find the sums inside the band -----------------------------------------------
                uint32_t mask[4] = {0u, 0u, 0u, 0u}; // [i]: bit 3b + channel ↔ acc[b][channel][i] needs the chain
                uint32_t px_ok = 0u;                 // the same 24 bits: pixel 8n + b inside the image
#pragma unroll
                for(int b = 0; b < 8; b++)
                    px_ok |= (x0 + 8 * n + b < W) ? 7u << (3 * b) : 0u;
                uint8_t *obase = out_base(LFI_SX_ABL == 6 ? 0 : ty, LFI_SX_ABL == 6 ? 0 : x0);
                // opaque per tile: the compiler otherwise computes every store's per-lane offset once per kernel and keeps them all (24
                // registers for the planar stores: spills)
                uint32_t plane_e = plane_b;
                asm volatile("" : "+s"(plane_e));
                const bool full_x = x0 + 8 * n + 8 <= W;
#pragma unroll
                for(int i = 0; i < 4; i++)
                {
                    if(i >= nvalid) // wave-uniform; otherwise lane (n = 0, kg = 0) is active below
                        continue;
                    uint32_t rgba[8];
                    uint32_t pl[3][2]; // PLANAR_OUT: [channel][pixels 0–3 | 4–7] of this view, a byte per pixel
#pragma unroll
                    for(int b = 0; b < 8; b++)
                    {
                        uint32_t bits[3];
#pragma unroll
                        for(int ch = 0; ch < 3; ch++)
                        {
                            const float v = acc[b][ch][i];
                            const float t = v + 16384.0f; // rounds S̃ to an integer (RN-even), left in the low mantissa bits
                            const float dist = v - (t - 16384.0f);
                            const float pow2 = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v + bmax_acc) & 0x7f800000u);
                            const float inside = __builtin_fmaf(-chain_acc, pow2, base_acc);
                            if constexpr(LFI_SX_ABL != 4)
                                mask[i] |= __builtin_fabsf(dist) > inside ? 1u << (3 * b + ch) : 0u;
                            bits[ch] = __builtin_bit_cast(uint32_t, t);
                            if constexpr(PLANAR_OUT)
                            {
                                // the rounded byte into position b & 3 of the channel's dword (one v_perm per sum)
                                const uint32_t sel = (b & 3) == 0 ? 0x0c0c0c04u : ((b & 3) == 1 ? 0x0c0c0400u : ((b & 3) == 2 ? 0x0c040100u : 0x04020100u));
                                pl[ch][b >> 2] = __builtin_amdgcn_perm(bits[ch], (b & 3) == 0 ? 0u : pl[ch][b >> 2], sel);
                            }
                        }
                        if constexpr(!PLANAR_OUT)
                        {
                            const uint32_t rg = __builtin_amdgcn_perm(bits[1], bits[0], 0x0c0c0400u); // [R, G, 0, 0]
                            rgba[b] = __builtin_amdgcn_perm(bits[2], rg, 0x0d040100u);               // [R, G, B, 0xff]
                        }
                    }
                    const bool view_ok = 4 * kg + i < nvalid;
                    mask[i] &= view_ok ? px_ok : 0u;
                    if constexpr(PLANAR_OUT)
                    {
                        // the pitch is a multiple of 128 ≥ W: a tile's row of a plane is a whole 128-byte line, also past the image's right edge
                        if(view_ok && LFI_SX_ABL != 5)
                        {
#pragma unroll
                            for(int ch = 0; ch < 3; ch++)
                            {
                                const u32x2 px8 = {pl[ch][0], pl[ch][1]};
                                u32x2 *o8 = reinterpret_cast<u32x2 *>(obase + out_off(plane_e, uint32_t(4 * kg + i), uint32_t(8 * n), uint32_t(ch)));
                                if constexpr(NT_STORE && LFI_SX_NT)
                                    __builtin_nontemporal_store(px8, o8);
                                else
                                    *o8 = px8;
                            }
                        }
                        st1 += 3;
                        continue;
                    }
                    uint32_t *out = reinterpret_cast<uint32_t *>(obase + out_off(plane_e, uint32_t(4 * kg + i), uint32_t(8 * n), 0u));
                    if(full_x)
                    {
                        if(view_ok && LFI_SX_ABL != 5)
                        {
                            const u32x4 lo4 = {rgba[0], rgba[1], rgba[2], rgba[3]}, hi4 = {rgba[4], rgba[5], rgba[6], rgba[7]};
                            if constexpr(NT_STORE && LFI_SX_NT)
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
                            if(x0 + 8 * n + b < W)
                                out[b] = rgba[b];
                    }
                    // counted for the vmcnt bookkeeping only where certainly issued (lane n = 0, kg = 0 takes the full-width branch);
                    // an undercount only makes the next waits stricter
                    if(x0 + 8 <= W)
                        st1 += 2;
                }
                // ---- compact the flagged sums into the wave's queue; what does not fit is recomputed here and now, from global memory ----
                int count = 0;
                while(true)
                {
                    const uint32_t any_bits = mask[0] | mask[1] | mask[2] | mask[3];
                    const uint64_t m = __builtin_amdgcn_ballot_w64(any_bits != 0u);
                    if(m == 0ull)
                        break;
                    const int i_sel = mask[0] ? 0 : (mask[1] ? 1 : (mask[2] ? 2 : 3));
                    const uint32_t word = i_sel == 0 ? mask[0] : (i_sel == 1 ? mask[1] : (i_sel == 2 ? mask[2] : mask[3]));
                    const int bit = any_bits ? __builtin_ctz(word) : 0;
                    const uint32_t rest = word & (word - 1u);
                    mask[0] = i_sel == 0 ? rest : mask[0];
                    mask[1] = i_sel == 1 ? rest : mask[1];
                    mask[2] = i_sel == 2 ? rest : mask[2];
                    mask[3] = i_sel == 3 ? rest : mask[3];
                    const int ahead = __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
                    const int slot = count + ahead;
                    const bool mine = any_bits != 0u;
                    if(mine && slot < SX_QCAP)
                        queue[slot] = static_cast<uint16_t>(uint32_t(lane) | (uint32_t(bit) << 6) | (uint32_t(i_sel) << 11));
                    const bool spill = mine && slot >= SX_QCAP;
                    if(__builtin_amdgcn_ballot_w64(spill) != 0ull) // wave-uniform; rare (adversarial inputs: every sum a tie)
                    {
                        // this lane's own sum, tap by tap from the planar copy (pixel x at byte x + padx) and the weight table
                        const uint32_t b = (uint32_t(bit) * 11u) >> 5, ch = uint32_t(bit) - 3u * b;
                        const int x = x0 + 8 * n + int(b), y = a.out_y0 + ty;
                        const int view = vw0 + 4 * kg + i_sel;
                        float s = 0.0f;
                        if(spill)
                        {
                            const uint16_t *wrow = a.w16s + (size_t)view * a.k_pad;
                            for(int g = 0; g < a.n_images; g++)
                            {
                                const int2 o = off_table[g];
                                const int sy = clampi(y + o.y, 0, H - 1) - a.in_y0;
                                const uint8_t p = a.planar[((size_t)g * 3 + ch) * shift_stride + (size_t)sy * a.planar_pitch + (x + o.x + a.planar_padx)];
                                const float w = static_cast<float>(__builtin_bit_cast(_Float16, wrow[g]));
                                s = __builtin_fmaf(static_cast<float>(p), w, s);
                            }
                            obase[out_off(plane_e, uint32_t(4 * kg + i_sel), uint32_t(8 * n) + b, ch)] = static_cast<uint8_t>(byte_of15(s));
                        }
                    }
                    count += __builtin_popcountll(m);
                }
                queued = min(count, SX_QCAP);
                const uint32_t code0 = queue[lane < queued ? lane : 0], code1 = queue[64 + lane < queued ? 64 + lane : 0];
                entry0 = lane < queued ? pack_entry(code0) : 0u; // lanes without a sum run the chain on entry 0 and store nothing
                entry1 = 64 + lane < queued ? pack_entry(code1) : 0u;
                s0 = s1 = 0.0f;
            }
            if constexpr((is_c || is_mc) && LFI_SX_ABL == 0)
            {
                // chain over chunk cc for the queued sums: MC — chunk 0 of this tile, still in its buffer; C' — the previous tile's chunk
                chain(std::integral_constant<int, cc>{}, buf, kc, entry0, s0);
                if(queued > 64)
                    chain(std::integral_constant<int, cc>{}, buf, kc, entry1, s1);
            }
            if constexpr(is_last_c && LFI_SX_ABL != 2)
            {
                // the chain's bytes over the rounded ones (not counted in st1: an undercount is safe)
                uint8_t *obase = out_base(ty, x0);
                if(lane < queued)
                    obase[out_off(plane_b, (entry0 >> 7) & 15u, entry0 & 127u, entry0 >> 11)] = static_cast<uint8_t>(byte_of(s0));
                if(64 + lane < queued)
                    obase[out_off(plane_b, (entry1 >> 7) & 15u, entry1 & 127u, entry1 >> 11)] = static_cast<uint8_t>(byte_of(s1));
            }
        }
#if LFI_SX_TRACE
        {
            const unsigned long long tC = __builtin_amdgcn_s_memtime();
            if(threadIdx.x == 0)
            {
                tr[2 * sl] += tB - tA;
                tr[2 * sl + 1] += tC - tB;
                if constexpr(is_mc)
                {
                    tr[15] += (unsigned long long)queued;
                    tr[16] += queued > 64;
                    tr[17] += queued == 0;
                    tr[18] += queued >= SX_QCAP;
                }
            }
        }
#endif
        if(!have1)
            return false;
        have1 = have2;
        nd1 = nd2;
        return true;
    };
    bool more = true;
    while(more)
    {
        p3_for_each_chunk<NU>([&](auto sl_tag) {
            if(more && slot_valid(cj, decltype(sl_tag)::value))
                more = unit(sl_tag);
        });
        cj++;
    }
#if LFI_SX_TRACE
    if(threadIdx.x == 0)
        tr[14] = __builtin_amdgcn_s_memtime() - t_begin;
    __syncthreads();
    if(threadIdx.x < 32 && blockIdx.x < 1024)
        lfi_sx_trace_buf[blockIdx.x * 32 + threadIdx.x] = tr[threadIdx.x];
#endif
}

} // namespace lfi
