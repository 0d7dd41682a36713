// ablate.hip — what does the memory system give this access pattern?  (config 2: 64 input planes read at shifted positions,
// 64 output planes written; 1920x1080 RGBA8.)  Kernels: plain copy, gather-only, scatter-only, gather+scatter, with
// different run lengths per wave, bytes per lane, XCD mapping and store policy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <string>

#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while(0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

constexpr int W = 1920, H = 1080, NIMG = 64, NV = 64;
constexpr size_t PLANE = (size_t)W * H; // pixels

__device__ __forceinline__ uint32_t xcd_contig(uint32_t b, uint32_t nb)
{
    const uint32_t xcd = b & 7u, idx = b >> 3, q = nb >> 3, rem = nb & 7u;
    return xcd * q + min(xcd, rem) + idx;
}

__global__ void __launch_bounds__(256) k_copy(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n)
{
    for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        dst[i] = src[i];
}

struct Offs { int ox[64], oy[64]; };

// one wave = a run of 64*PXL pixels of one row; 4 waves of a block = 4 consecutive runs.  READ: 64 shifted planes;
// WRITE: 64 planes.  PXL = dwords per lane (1 or 4).
template <int PXL, bool READ, bool WRITE, bool XCD, bool NT, int UNROLL>
__global__ void __launch_bounds__(256) k_pattern(const uint32_t *__restrict__ grid, uint32_t *__restrict__ views, const Offs offs, int tiles_x, int n_tiles)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t b = XCD ? xcd_contig(blockIdx.x, gridDim.x) : blockIdx.x;
    const int tile = b * 4 + wave;
    if(tile >= n_tiles) return;
    const int y = tile / tiles_x, x0 = (tile - y * tiles_x) * (64 * PXL);
    const int xl = x0 + lane * PXL;
    uint32_t acc[PXL];
#pragma unroll
    for(int i = 0; i < PXL; i++) acc[i] = lane * 0x01010101u;
    if(READ)
    {
#pragma unroll UNROLL
        for(int g = 0; g < NIMG; g++)
        {
            const int sy = min(max(y + offs.oy[g], 0), H - 1);
            int sx = xl + offs.ox[g];
            sx = min(max(sx, 0), W - PXL); // (clamped as a run: enough for a bandwidth probe)
            const uint32_t *p = grid + (size_t)g * PLANE + (size_t)sy * W + sx;
            if(PXL == 1) acc[0] ^= *p;
            else { u32x4_a4 v = *reinterpret_cast<const u32x4_a4 *>(p); acc[0] ^= v.x; acc[1 % PXL] ^= v.y; acc[2 % PXL] ^= v.z; acc[3 % PXL] ^= v.w; }
        }
    }
    if(WRITE)
    {
#pragma unroll UNROLL
        for(int v = 0; v < NV; v++)
        {
            uint32_t *p = views + (size_t)v * PLANE + (size_t)y * W + xl;
            if(PXL == 1) { if(NT) __builtin_nontemporal_store(acc[0] + v, p); else *p = acc[0] + v; }
            else
            {
                u32x4 t = {acc[0] + v, acc[1 % PXL], acc[2 % PXL], acc[3 % PXL]};
                if(NT) __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(p)); else *reinterpret_cast<u32x4 *>(p) = t;
            }
        }
    }
    else
    {
        uint32_t s = 0;
#pragma unroll
        for(int i = 0; i < PXL; i++) s ^= acc[i];
        if(s == 0x12345678u) views[tile] = s; // keeps the loads alive
    }
}

// PLANAR read probe: the inputs as 3 x 64 byte planes (one per image and colour channel, no alpha): a wave = a run of 256 pixels
// of one row, a lane reads one dword (4 pixels of one channel) per (image, channel) = 192 loads of 256 B per wave instead of
// 64 loads of 1 KB; writes as k_pattern<4> (64 planes, 16 B per lane).  Would dropping the alpha bytes from the read side pay?
template <bool WRITE, bool NT, int UNROLL>
__global__ void __launch_bounds__(256) k_planar(const uint32_t *__restrict__ planes, uint32_t *__restrict__ views, const Offs offs, int tiles_x, int n_tiles)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t b = xcd_contig(blockIdx.x, gridDim.x);
    const int tile = b * 4 + wave;
    if(tile >= n_tiles) return;
    const int y = tile / tiles_x, x0 = (tile - y * tiles_x) * 256;
    uint32_t acc[4] = {lane * 0x01010101u, 1u, 2u, 3u};
    const size_t plane_dw = PLANE / 4; // dwords per byte plane
#pragma unroll UNROLL
    for(int g = 0; g < NIMG; g++)
    {
        const int sy = min(max(y + offs.oy[g], 0), H - 1);
        int sx = x0 + 4 * lane + (offs.ox[g] & ~3); // dword-aligned (the real thing would pick one of four byte-shifted copies)
        sx = min(max(sx, 0), W - 4);
#pragma unroll
        for(int c = 0; c < 3; c++)
            acc[c] ^= planes[(size_t)(3 * g + c) * plane_dw + ((size_t)sy * W + sx) / 4];
    }
    if(WRITE)
    {
#pragma unroll UNROLL
        for(int v = 0; v < NV; v++)
        {
            uint32_t *p = views + (size_t)v * PLANE + (size_t)y * W + x0 + 4 * lane;
            u32x4 t = {acc[0] + v, acc[1], acc[2], acc[3]};
            if(NT) __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(p)); else *reinterpret_cast<u32x4 *>(p) = t;
        }
    }
    else if((acc[0] ^ acc[1] ^ acc[2]) == 0x12345678u) views[tile] = acc[0];
}

template <typename F> float time_it(F f, int runs = 10);

typedef __attribute__((address_space(3))) void *lds_ptr_t;
// LDS-DMA gather: a 256-thread workgroup stages 64 images x 128 pixels (32 KB) like blend_ten_lds' load phase, waits, and
// (optionally) writes 64 output runs of 128 pixels.  X4: 16 B/lane pieces; ALIGN: round offsets to 4 pixels (16 B aligned sources).
template <bool X4, bool ALIGN, bool WRITE, int LDSKB>
__global__ void __launch_bounds__(256) k_dma(const uint32_t *__restrict__ grid, uint32_t *__restrict__ views, const Offs offs, int tiles_x, int n_tiles)
{
    __shared__ __attribute__((aligned(16))) uint32_t tile[LDSKB * 256];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t t = xcd_contig(blockIdx.x, gridDim.x);
    const int y = t / tiles_x, x0 = (t - y * tiles_x) * 128;
    if(X4)
    {
        for(int q = wave; q < 32; q += 4)
        {
            const int g = 2 * q + (lane >> 5);
            int ox = offs.ox[g]; if(ALIGN) ox &= ~3;
            const int sy = min(max(y + offs.oy[g], 0), H - 1);
            int sx = x0 + ox + 4 * (lane & 31);
            sx = min(max(sx, 0), W - 4);
            __builtin_amdgcn_global_load_lds(grid + (size_t)g * PLANE + (size_t)sy * W + sx, (lds_ptr_t)(tile + 256 * q), 16, 0, 0);
        }
    }
    else
    {
        for(int u = wave; u < 128; u += 4)
        {
            const int g = u >> 1;
            const int sy = min(max(y + offs.oy[g], 0), H - 1);
            int sx = x0 + offs.ox[g] + (u & 1) * 64 + lane;
            sx = min(max(sx, 0), W - 1);
            __builtin_amdgcn_global_load_lds(grid + (size_t)g * PLANE + (size_t)sy * W + sx, (lds_ptr_t)(tile + 64 * u), 4, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t acc = 0;
#pragma unroll
    for(int j = 0; j < 8; j++) acc ^= tile[(wave * 16 + j) * 128 + lane];
    if(WRITE)
    {
        // wave w writes views 16w..16w+15, 128 px each as 2 x 64-lane dword stores
#pragma unroll 8
        for(int v = 0; v < 16; v++)
            for(int s = 0; s < 2; s++)
                views[(size_t)(wave * 16 + v) * PLANE + (size_t)y * W + x0 + s * 64 + lane] = acc + v;
    }
    else if(acc == 0x12345678u) views[t] = acc;
}

template <bool X4, bool ALIGN, bool WRITE, int LDSKB>
void run_dma(const char *name, const uint32_t *grid, uint32_t *views, const Offs &o)
{
    const int tiles_x = W / 128, n_tiles = tiles_x * H;
    float ms = time_it([&] { hipLaunchKernelGGL((k_dma<X4, ALIGN, WRITE, LDSKB>), dim3(n_tiles), dim3(256), 0, 0, grid, views, o, tiles_x, n_tiles); });
    const double bytes = 4.0 * W * H * (NIMG + (WRITE ? NV : 0));
    printf("%-44s %8.1f us  %7.0f GB/s\n", name, ms * 1e3, bytes / ms / 1e6);
}

template <typename F>
float time_it(F f, int runs)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f(); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for(int i = 0; i < runs; i++) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

template <int PXL, bool READ, bool WRITE, bool XCD, bool NT, int UNROLL>
void run(const char *name, const uint32_t *grid, uint32_t *views, const Offs &o)
{
    const int tiles_x = W / (64 * PXL), n_tiles = tiles_x * H;
    const int blocks = (n_tiles + 3) / 4;
    float ms = time_it([&] { hipLaunchKernelGGL((k_pattern<PXL, READ, WRITE, XCD, NT, UNROLL>), dim3(blocks), dim3(256), 0, 0, grid, views, o, tiles_x, n_tiles); });
    const double bytes = 4.0 * W * H * ((READ ? NIMG : 0) + (WRITE ? NV : 0));
    printf("%-44s %8.1f us  %7.0f GB/s\n", name, ms * 1e3, bytes / ms / 1e6);
}

template <bool WRITE, bool NT, int UNROLL>
void run_planar(const char *name, const uint32_t *planes, uint32_t *views, const Offs &o)
{
    const int tiles_x = W / 256, n_tiles = tiles_x * H;
    const int blocks = (n_tiles + 3) / 4;
    float ms = time_it([&] { hipLaunchKernelGGL((k_planar<WRITE, NT, UNROLL>), dim3(blocks), dim3(256), 0, 0, planes, views, o, tiles_x, n_tiles); });
    const double moved = 1.0 * W * H * (3 * NIMG + (WRITE ? 4 * NV : 0));
    printf("%-44s %8.1f us  %7.0f GB/s moved (%.0f MB)\n", name, ms * 1e3, moved / ms / 1e6, moved / 1e6);
}

int main()
{
    uint32_t *grid, *views;
    CK(hipMalloc(&grid, PLANE * 4 * NIMG)); CK(hipMalloc(&views, PLANE * 4 * NV));
    CK(hipMemset(grid, 1, PLANE * 4 * NIMG)); CK(hipMemset(views, 2, PLANE * 4 * NV));
    Offs o;
    for(int col = 0; col < 8; col++) for(int row = 0; row < 8; row++) { int g = col * 8 + row; o.ox[g] = (int)lroundf((3.5f - col) / 8 * 1920 * 0.23f); o.oy[g] = (int)lroundf((3.5f - row) / 8 * 1080 * 0.997f * 0.23f); }
    {
        size_t n = PLANE * NIMG / 4;
        float ms = time_it([&] { hipLaunchKernelGGL(k_copy, dim3(256 * 8), dim3(256), 0, 0, (const u32x4 *)grid, (u32x4 *)views, n); });
        printf("%-44s %8.1f us  %7.0f GB/s (read+write)\n", "plain copy 531MB->531MB grid-stride x4", ms * 1e3, 2.0 * PLANE * 4 * NIMG / ms / 1e6);
    }
    run<1, true, false, true, false, 8>("gather only  dword xcd u8", grid, views, o);
    run<1, true, false, true, false, 16>("gather only  dword xcd u16", grid, views, o);
    run<1, true, false, true, false, 64>("gather only  dword xcd u64", grid, views, o);
    run<1, true, false, false, false, 16>("gather only  dword noxcd u16", grid, views, o);
    run<4, true, false, true, false, 8>("gather only  x4 xcd u8", grid, views, o);
    run<4, true, false, true, false, 16>("gather only  x4 xcd u16", grid, views, o);
    run<4, true, false, false, false, 16>("gather only  x4 noxcd u16", grid, views, o);
    run<1, false, true, true, false, 16>("scatter only dword xcd u16", grid, views, o);
    run<1, false, true, true, true, 16>("scatter only dword xcd nt u16", grid, views, o);
    run<1, false, true, false, false, 16>("scatter only dword noxcd u16", grid, views, o);
    run<4, false, true, true, false, 16>("scatter only x4 xcd u16", grid, views, o);
    run<4, false, true, true, true, 16>("scatter only x4 xcd nt u16", grid, views, o);
    run<4, false, true, false, false, 16>("scatter only x4 noxcd u16", grid, views, o);
    run<1, true, true, true, false, 16>("gather+scatter dword xcd u16", grid, views, o);
    run<1, true, true, true, true, 16>("gather+scatter dword xcd nt u16", grid, views, o);
    run<4, true, true, true, false, 16>("gather+scatter x4 xcd u16", grid, views, o);
    run<4, true, true, true, true, 16>("gather+scatter x4 xcd nt u16", grid, views, o);
    run<4, true, true, false, false, 16>("gather+scatter x4 noxcd u16", grid, views, o);
    run<4, true, true, false, true, 16>("gather+scatter x4 noxcd nt u16", grid, views, o);
    run_planar<false, false, 16>("planar gather only (3 B/px) u16", grid, views, o);
    run_planar<true, true, 16>("planar gather (3 B/px) + scatter x4 nt u16", grid, views, o);
    run_planar<true, true, 8>("planar gather (3 B/px) + scatter x4 nt u8", grid, views, o);
    run<4, true, true, true, true, 16>("gather+scatter x4 xcd nt u16 (again)", grid, views, o);
    run_dma<true, false, false, 32>("dma gather x4 unaligned 32KB/WG", grid, views, o);
    run_dma<true, true, false, 32>("dma gather x4 aligned16 32KB/WG", grid, views, o);
    run_dma<false, false, false, 32>("dma gather dword 32KB/WG", grid, views, o);
    run_dma<true, false, false, 64>("dma gather x4 unaligned (64KB LDS: 2 WG/CU)", grid, views, o);
    run_dma<true, false, true, 32>("dma gather x4 unaligned + scatter", grid, views, o);
    run_dma<true, true, true, 32>("dma gather x4 aligned + scatter", grid, views, o);
    run_dma<false, false, true, 32>("dma gather dword + scatter", grid, views, o);
    return 0;
}
