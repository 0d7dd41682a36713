// probe (round 3): what does HBM take when a kernel ONLY writes?  Config 4 whole on one GPU (8×8 @4K, 256 views) is 80 % stores
// (6.4 GB of views against 1.6 GB of inputs) and runs at 2.0 ms = 4.0 TB/s: is that the write rate of the part, or of the pattern
// (768 view planes open at once, 128 bytes per plane and store)?
//   linear    every workgroup streams one contiguous range (16 B per lane, consecutive lanes consecutive addresses)
//   planes    blend_p3's pattern: persistent workgroups over 128-pixel tiles, per tile 128 bytes into each of P plane rows
// each on ordinary device memory and on uncached memory (hipDeviceMallocUncached: where the planar views live), non-temporal stores.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe_store_rate tools/probe_store_rate.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while(0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t xcd_contig(uint32_t b, uint32_t nb)
{
    const uint32_t xcd = b & 7u, idx = b >> 3, q = nb >> 3, rem = nb & 7u;
    return xcd * q + (xcd < rem ? xcd : rem) + idx;
}

__global__ void __launch_bounds__(256, 2) linear(uint8_t *dst, size_t bytes)
{
    const size_t per_wg = (bytes / gridDim.x) & ~size_t(4095);
    uint8_t *p = dst + per_wg * xcd_contig(blockIdx.x, gridDim.x);
    const u32x4 v = {threadIdx.x, 1u, 2u, 3u};
    for(size_t o = 16 * threadIdx.x; o < per_wg; o += 4096)
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p + o));
}

// P planes of W×H bytes; tile = 128 pixels of one row; a wave writes 16 planes per store instruction group the way blend_p3 does:
// lane pair (n, n^1) → 16 bytes, eight pairs = 128 bytes of one plane row, the wave's 8 (kg, view) combinations = 8 planes per instruction
__global__ void __launch_bounds__(256, 2) planes(uint8_t *dst, int W, int H, int P, int run)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles_x = W / run, n_tiles = tiles_x * H;
    const size_t plane_b = (size_t)W * H;
    const u32x4 v = {threadIdx.x, 1u, 2u, 3u};
    const int per_row = run / 16; // lanes per plane row and instruction
    const int rows_per_inst = 64 / per_row;
    for(int t = int(xcd_contig(blockIdx.x, gridDim.x)); t < n_tiles; t += gridDim.x)
    {
        const int y = t / tiles_x, x0 = (t - y * tiles_x) * run;
        uint8_t *row = dst + (size_t)y * W + x0 + 16 * (lane % per_row);
        for(int p = wave * rows_per_inst + lane / per_row; p < P; p += 4 * rows_per_inst)
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(row + (size_t)p * plane_b));
    }
}

// two x-adjacent 128-byte tiles per workgroup and iteration.  MODE 0: per plane row, tile A's 128 bytes and tile B's back to back;
// MODE 1: all of tile A's stores, then all of tile B's; MODE 2: the same with ≈1 µs (s_sleep) between the tiles — two units of blend_p3
template <int MODE>
__global__ void __launch_bounds__(256, 2) pairs(uint8_t *dst, int W, int H, int P)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pairs_x = W / 256, n_pairs = pairs_x * H;
    const size_t plane_b = (size_t)W * H;
    const u32x4 v = {threadIdx.x, 1u, 2u, 3u};
    for(int t = int(xcd_contig(blockIdx.x, gridDim.x)); t < n_pairs; t += gridDim.x)
    {
        const int y = t / pairs_x, x0 = (t - y * pairs_x) * 256;
        uint8_t *row = dst + (size_t)y * W + x0 + 16 * (lane & 7);
        if(MODE == 0)
        {
            for(int p = wave * 8 + (lane >> 3); p < P; p += 32)
            {
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(row + (size_t)p * plane_b));
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(row + 128 + (size_t)p * plane_b));
            }
        }
        else
        {
            for(int half = 0; half < 2; half++)
            {
                for(int p = wave * 8 + (lane >> 3); p < P; p += 32)
                    __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(row + 128 * half + (size_t)p * plane_b));
                if(MODE == 2)
                    __builtin_amdgcn_s_sleep(40);
            }
        }
    }
}

int main(int argc, char **argv)
{
    const int W = 3840, H = 2160;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for(int uncached = 0; uncached < 2; uncached++)
        for(int P : {192, 768})
        {
            const size_t bytes = (size_t)W * H * P;
            uint8_t *d = nullptr;
            if(uncached)
                CK(hipExtMallocWithFlags((void **)&d, bytes, hipDeviceMallocUncached));
            else
                CK(hipMalloc((void **)&d, bytes));
            auto time = [&](const char *what, auto launch) {
                for(int i = 0; i < 2; i++)
                    launch();
                CK(hipDeviceSynchronize());
                float best = 1e9f;
                for(int rep = 0; rep < 3; rep++)
                {
                    CK(hipEventRecord(e0));
                    for(int i = 0; i < 4; i++)
                        launch();
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    best = std::min(best, ms / 4);
                }
                printf("%-9s P=%3d %-28s %7.3f ms  %6.2f TB/s\n", uncached ? "uncached" : "ordinary", P, what, best, bytes / best * 1e-9);
                fflush(stdout);
            };
            time("linear", [&] { hipLaunchKernelGGL(linear, dim3(512), dim3(256), 0, 0, d, bytes); });
            time("linear, 2048 workgroups", [&] { hipLaunchKernelGGL(linear, dim3(2048), dim3(256), 0, 0, d, bytes); });
            for(int run : {128, 256})
            {
                char name[64];
                snprintf(name, sizeof name, "planes, %d-byte runs", run);
                time(name, [&] { hipLaunchKernelGGL(planes, dim3(512), dim3(256), 0, 0, d, W, H, P, run); });
            }
            time("pairs of 128-byte tiles, row by row", [&] { hipLaunchKernelGGL(pairs<0>, dim3(512), dim3(256), 0, 0, d, W, H, P); });
            time("pairs, tile after tile", [&] { hipLaunchKernelGGL(pairs<1>, dim3(512), dim3(256), 0, 0, d, W, H, P); });
            time("pairs, tile, 1 us, tile", [&] { hipLaunchKernelGGL(pairs<2>, dim3(512), dim3(256), 0, 0, d, W, H, P); });
            CK(hipFree(d));
        }
    return 0;
}
