// Round 4: what does the memory system give the ALL-FOCUS read pattern?  Config 5 (15x15 images of 3840x2160 RGBA8), the per-pixel warp
// sx = (int)fma(f, ox, x), sy = (int)fma(f, oy, y) with f from a focus map, reads only (a xor per lane keeps them alive), no LDS, no arithmetic:
// the ceiling for any kernel that gathers its samples one by one.  Variants: one dword per lane and sample (what blend_persist<allfocus>
// requests), 16 bytes per lane where a lane's four pixels share the shift (else four dwords), both at 4/8 waves per SIMD.
// usage: probe_gather <map.bin: H*W bytes> <offsets.bin: 225 x float2> focus range     (tools/probe_gather.sh writes the inputs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while(0)
constexpr int W = 3840, H = 2160, N = 225;
typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ uint32_t xcd_contig(uint32_t b, uint32_t nb) { const uint32_t xcd = b & 7u, idx = b >> 3, q = nb >> 3, rem = nb & 7u; return xcd * q + min(xcd, rem) + idx; }

// one wave = 64 pixels of a row (PX4 = false) or 256 pixels (PX4 = true: lane = 4 consecutive pixels)
template <bool PX4>
__global__ void __launch_bounds__(256) gather(const uint32_t *__restrict__ grid, const uint8_t *__restrict__ map, const float2 *__restrict__ offs, float focus, float range,
                                              uint32_t *__restrict__ sink, int n_wtiles, int g0, int g1)
{
    const int lane = threadIdx.x & 63;
    const int per_row = W / (PX4 ? 256 : 64);
    uint32_t acc = 0;
    for(int wt = xcd_contig(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6); wt < n_wtiles; wt += gridDim.x * 4)
    {
        const int y = wt / per_row, x0 = (wt - y * per_row) * (PX4 ? 256 : 64);
        if constexpr(!PX4)
        {
            const int x = x0 + lane;
            const float f = __builtin_fmaf(float(map[(size_t)y * W + x]) / 255.0f, range, focus), xf = float(x), yf = float(y);
#pragma unroll 8
            for(int g = g0; g < g1; g++)
            {
                const float2 o = offs[g];
                const int sx = min(max(int(__builtin_fmaf(f, o.x, xf)), 0), W - 1), sy = min(max(int(__builtin_fmaf(f, o.y, yf)), 0), H - 1);
                acc ^= grid[(size_t)g * W * H + (size_t)sy * W + sx];
            }
        }
        else
        {
            const int x = x0 + 4 * lane;
            const uint32_t m4 = *reinterpret_cast<const uint32_t *>(map + (size_t)y * W + x);
            float f[4];
#pragma unroll
            for(int i = 0; i < 4; i++)
                f[i] = __builtin_fmaf(float((m4 >> (8 * i)) & 0xffu) / 255.0f, range, focus);
            const float yf = float(y);
#pragma unroll 4
            for(int g = g0; g < g1; g++)
            {
                const float2 o = offs[g];
                int sx[4], sy[4];
#pragma unroll
                for(int i = 0; i < 4; i++)
                {
                    sx[i] = min(max(int(__builtin_fmaf(f[i], o.x, float(x + i))), 0), W - 1);
                    sy[i] = min(max(int(__builtin_fmaf(f[i], o.y, yf)), 0), H - 1);
                }
                const uint32_t *plane = grid + (size_t)g * W * H;
                const bool run = sx[1] == sx[0] + 1 && sx[2] == sx[0] + 2 && sx[3] == sx[0] + 3 && sy[1] == sy[0] && sy[2] == sy[0] && sy[3] == sy[0];
                if(run)
                {
                    const u32x4a v = *reinterpret_cast<const u32x4a *>(plane + (size_t)sy[0] * W + sx[0]);
                    acc ^= v.x ^ v.y ^ v.z ^ v.w;
                }
                else
                {
#pragma unroll
                    for(int i = 0; i < 4; i++)
                        acc ^= plane[(size_t)sy[i] * W + sx[i]];
                }
            }
        }
    }
    if(acc == 0x12345678u)
        sink[threadIdx.x] = acc;
}

int main(int argc, char **argv)
{
    if(argc < 5) { printf("usage: probe_gather map.bin offsets.bin focus range\n"); return 1; }
    std::vector<uint8_t> hmap((size_t)W * H);
    std::vector<float2> hoffs(N);
    FILE *f = fopen(argv[1], "rb"); if(!f || fread(hmap.data(), 1, hmap.size(), f) != hmap.size()) { printf("cannot read the map\n"); return 1; } fclose(f);
    f = fopen(argv[2], "rb"); if(!f || fread(hoffs.data(), sizeof(float2), N, f) != (size_t)N) { printf("cannot read the offsets\n"); return 1; } fclose(f);
    const float focus = atof(argv[3]), range = atof(argv[4]);
    uint32_t *grid, *sink; uint8_t *map; float2 *offs;
    CK(hipMalloc((void **)&grid, (size_t)N * W * H * 4)); CK(hipMemset(grid, 0x5a, (size_t)N * W * H * 4));
    CK(hipMalloc((void **)&map, hmap.size())); CK(hipMalloc((void **)&offs, N * sizeof(float2))); CK(hipMalloc((void **)&sink, 4096));
    CK(hipMemcpy(offs, hoffs.data(), N * sizeof(float2), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<uint8_t> cmap(hmap.size(), 128);
    for(int which = 0; which < 2; which++)
    {
        CK(hipMemcpy(map, which ? cmap.data() : hmap.data(), hmap.size(), hipMemcpyHostToDevice));
        for(int px4 = 0; px4 < 2; px4++)
            for(int blocks_per_cu : {2, 4, 8})
            {
                const int n_wtiles = H * (W / (px4 ? 256 : 64));
                const int grid_blocks = 256 * blocks_per_cu;
                float best = 1e9f;
                for(int rep = 0; rep < 4; rep++)
                {
                    CK(hipEventRecord(e0));
                    if(px4) hipLaunchKernelGGL(gather<true>, dim3(grid_blocks), dim3(256), 0, 0, grid, map, offs, focus, range, sink, n_wtiles, 0, N);
                    else hipLaunchKernelGGL(gather<false>, dim3(grid_blocks), dim3(256), 0, 0, grid, map, offs, focus, range, sink, n_wtiles, 0, N);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
                }
                printf("%-9s map, %s, %d waves per SIMD: %7.3f ms  (%.2f TB/s of samples)\n", which ? "constant" : "estimated", px4 ? "16 B per lane where 4 pixels share the shift" : "one dword per lane and sample",
                       blocks_per_cu, best, (double)N * W * H * 4 / best * 1e-9);
                fflush(stdout);
            }
    }
    return 0;
}
