// ablate_out.hip — what does the memory system give the OUTPUT side when the views carry no alpha?  (config 2: 8x8 @1920x1080,
// 64 views.)  Every probe reads the planar, alpha-free input copy exactly as blend_planar does (192 byte-plane runs of 128 B per
// 128-pixel tile, from the shift copy that makes the run dword aligned) and writes the 64 views of the tile in one of:
//   rgba      64 RGBA planes, a wave = 32 pixels x 64 views, 4 B per lane (what blend_planar's epilogue issues: 128 B per half-wave)
//   planar    192 byte planes [view][channel], a wave = 128 pixels x 16 views, 8 B per lane (16 lanes = 128 B of one plane row)
//   rgb24     64 packed-RGB planes (3 B/px), a wave = 128 pixels x 16 views, 24 B per lane (16 lanes = 384 B of one view row)
// No arithmetic: the floor of each layout for the kernel that has to produce it.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while(0)

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));

constexpr int W = 1920, H = 1080, NIMG = 64, NV = 64, TPX = 128;
constexpr size_t PLANE = (size_t)W * H;

struct Offs { int ox[64], oy[64]; };

__device__ __forceinline__ uint32_t xcd_contig(uint32_t b, uint32_t nb)
{
    const uint32_t xcd = b & 7u, idx = b >> 3, q = nb >> 3, rem = nb & 7u;
    return xcd * q + min(xcd, rem) + idx;
}
// XCD x owns one contiguous range of the tile sequence for the whole launch and walks it in order
__device__ __forceinline__ void xcd_range(uint32_t b, uint32_t nb, uint32_t n_tiles, uint32_t &first, uint32_t &end, uint32_t &step)
{
    const uint32_t xcd = b & 7u, idx = b >> 3;
    const uint32_t lo = (uint64_t)n_tiles * xcd / 8u, hi = (uint64_t)n_tiles * (xcd + 1u) / 8u;
    step = (nb + 7u - xcd) / 8u; // blocks of this XCD
    first = lo + idx;
    end = hi;
}

enum { OUT_RGBA = 0, OUT_PLANAR = 1, OUT_RGB24 = 2, OUT_RGB24_X3 = 3, OUT_NONE = 4, OUT_PLANAR64 = 5 };

// persistent: grid = 2 x CUs x ..., tiles t, t+G, ...   MAP: 0 = batches of G tiles split over the XCDs (blend_planar today), 1 = xcd_range
// wider tiles: runs of TW bytes per (image, channel) and per (view, channel) plane row — does the memory system prefer longer runs?
template <int TW, bool WRITE>
__global__ void __launch_bounds__(256) k_wide(const uint8_t *__restrict__ planar, uint8_t *__restrict__ views, const Offs offs, const int pitch, const int padx,
                                              const int tiles_x, const int n_tiles)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t shift_stride = (size_t)H * pitch;
    constexpr int LPR = TW / 16;       // lanes per run
    constexpr int RPI = 64 / LPR;      // runs per load instruction
    constexpr int NLOAD = 192 / RPI / 4; // load instructions per wave and tile
    for(uint32_t t = xcd_contig(blockIdx.x, gridDim.x); t < (uint32_t)n_tiles; t += gridDim.x)
    {
        const int y = t / tiles_x, x0 = (t - y * tiles_x) * TW;
        uint32_t acc[4] = {uint32_t(lane), 1u, 2u, 3u};
#pragma unroll
        for(int j = 0; j < NLOAD; j++)
        {
            const int run = (wave + 4 * j) * RPI + lane / LPR; // 0..191 = channel-major: c = run / 64, g = run % 64
            const int c = run >> 6, g = run & 63;
            const int sy = min(max(y + offs.oy[g], 0), H - 1);
            const int start = x0 + offs.ox[g] + padx, k = start & 3;
            const uint8_t *src = planar + (((size_t)g * 3 + c) * 4 + k) * shift_stride + (size_t)sy * pitch + (start - k) + 16 * (lane % LPR);
            const u32x4_a4 v = *reinterpret_cast<const u32x4_a4 *>(src);
            acc[0] ^= v.x; acc[1] ^= v.y; acc[2] ^= v.z; acc[3] ^= v.w;
        }
        if(WRITE)
        {
            // wave w: views 16w … 16w+15; a lane stores 8 bytes; TW/8 lanes per plane row → 64·8/TW views per instruction
            constexpr int LPV = TW / 8, VPI = 64 / LPV;
#pragma unroll
            for(int i = 0; i < 16 / VPI; i++)
#pragma unroll
                for(int ch = 0; ch < 3; ch++)
                {
                    const int v = 16 * wave + i * VPI + lane / LPV;
                    u32x2 *p = reinterpret_cast<u32x2 *>(views + ((size_t)v * 3 + ch) * PLANE + (size_t)y * W + x0 + 8 * (lane % LPV));
                    const u32x2 val = {acc[0] + i, acc[1] + ch};
                    __builtin_nontemporal_store(val, p);
                }
        }
        else if((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u)
            views[t] = 1;
    }
}

template <int OUT, bool NT, int MAP>
__global__ void __launch_bounds__(256) k_tile(const uint8_t *__restrict__ planar, uint8_t *__restrict__ views, const Offs offs, const int pitch, const int padx,
                                              const int tiles_x, const int n_tiles)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t shift_stride = (size_t)H * pitch;
    uint32_t t, t_end, t_step;
    if(MAP == 0)
    {
        t = xcd_contig(blockIdx.x, gridDim.x);
        t_end = n_tiles;
        t_step = gridDim.x;
    }
    else
        xcd_range(blockIdx.x, gridDim.x, n_tiles, t, t_end, t_step);
    for(; t < t_end; t += t_step)
    {
        const int y = t / tiles_x, x0 = (t - y * tiles_x) * TPX;
        uint32_t acc[4] = {uint32_t(lane), 1u, 2u, 3u};
        // read: 24 pieces of 8 runs x 128 B; wave w takes pieces w, w+4, ...
#pragma unroll
        for(int j = 0; j < 6; j++)
        {
            const int p = wave + 4 * j, c = p >> 3, g = 8 * (p & 7) + (lane >> 3);
            const int sy = min(max(y + offs.oy[g], 0), H - 1);
            const int start = x0 + offs.ox[g] + padx, k = start & 3;
            const uint8_t *src = planar + (((size_t)g * 3 + c) * 4 + k) * shift_stride + (size_t)sy * pitch + (start - k) + 16 * (lane & 7);
            const u32x4_a4 v = *reinterpret_cast<const u32x4_a4 *>(src);
            acc[0] ^= v.x; acc[1] ^= v.y; acc[2] ^= v.z; acc[3] ^= v.w;
        }
        if(OUT == OUT_RGBA)
        {
            const int r = lane & 31, h = lane >> 5;
#pragma unroll 8
            for(int i = 0; i < 32; i++)
            {
                const int v = 2 * i + h;
                uint32_t *p = reinterpret_cast<uint32_t *>(views) + (size_t)v * PLANE + (size_t)y * W + x0 + wave * 32 + r;
                if(NT) __builtin_nontemporal_store(acc[0] + i, p); else *p = acc[0] + i;
            }
        }
        else if(OUT == OUT_PLANAR)
        {
            const int c = lane & 15, kg = lane >> 4;
#pragma unroll
            for(int i = 0; i < 4; i++)
#pragma unroll
                for(int ch = 0; ch < 3; ch++)
                {
                    const int v = 16 * wave + 4 * kg + i;
                    u32x2 *p = reinterpret_cast<u32x2 *>(views + ((size_t)v * 3 + ch) * PLANE + (size_t)y * W + x0 + 8 * c);
                    const u32x2 val = {acc[0] + i, acc[1] + ch};
                    if(NT) __builtin_nontemporal_store(val, p); else *p = val;
                }
        }
        else if(OUT == OUT_PLANAR64)
        {
            // a wave = 64 pixels x 32 views: 4 bytes per lane, 16 lanes = 64 B of one plane row (half a line), four rows per instruction
            const int c = lane & 15, kg = lane >> 4, ph = wave & 1, vh = wave >> 1;
#pragma unroll
            for(int vg = 0; vg < 2; vg++)
#pragma unroll
                for(int i = 0; i < 4; i++)
#pragma unroll
                    for(int ch = 0; ch < 3; ch++)
                    {
                        const int v = 32 * vh + 16 * vg + 4 * kg + i;
                        uint32_t *p = reinterpret_cast<uint32_t *>(views + ((size_t)v * 3 + ch) * PLANE + (size_t)y * W + x0 + 64 * ph + 4 * c);
                        if(NT) __builtin_nontemporal_store(acc[0] + i + vg, p); else *p = acc[0] + i + vg;
                    }
        }
        else if(OUT == OUT_RGB24)
        {
            const int c = lane & 15, kg = lane >> 4;
#pragma unroll
            for(int i = 0; i < 4; i++)
#pragma unroll
                for(int s = 0; s < 3; s++)
                {
                    const int v = 16 * wave + 4 * kg + i;
                    u32x2 *p = reinterpret_cast<u32x2 *>(views + (size_t)v * 3 * PLANE + ((size_t)y * W + x0) * 3 + 24 * c + 8 * s);
                    const u32x2 val = {acc[0] + i, acc[1] + s};
                    if(NT) __builtin_nontemporal_store(val, p); else *p = val;
                }
        }
        else if(OUT == OUT_RGB24_X3)
        {
            const int c = lane & 15, kg = lane >> 4;
#pragma unroll
            for(int i = 0; i < 4; i++)
#pragma unroll
                for(int s = 0; s < 2; s++)
                {
                    const int v = 16 * wave + 4 * kg + i;
                    uint32_t *p = reinterpret_cast<uint32_t *>(views + (size_t)v * 3 * PLANE + ((size_t)y * W + x0) * 3 + 24 * c + 12 * s);
                    // three dword stores the compiler merges into one global_store_dwordx3
                    typedef uint32_t u32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));
                    const u32x3_a4 val = {acc[0] + i, acc[1] + s, acc[2]};
                    if(NT) __builtin_nontemporal_store(val, reinterpret_cast<u32x3_a4 *>(p)); else *reinterpret_cast<u32x3_a4 *>(p) = val;
                }
        }
        else if((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u)
            views[t] = 1;
    }
}

template <typename F>
float time_it(F f, int runs = 15)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for(int i = 0; i < 5; i++) f();
    CK(hipDeviceSynchronize());
    std::vector<float> t;
    for(int i = 0; i < runs; i++) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

static const uint8_t *g_planar; static uint8_t *g_views; static Offs g_o; static int g_pitch, g_padx, g_cus;

template <int OUT, bool NT, int MAP>
float run(const char *name, int wgs_per_cu, bool quiet = false)
{
    const int tiles_x = W / TPX, n_tiles = tiles_x * H;
    const int grid = std::min(n_tiles, wgs_per_cu * g_cus);
    float ms = time_it([&] { hipLaunchKernelGGL((k_tile<OUT, NT, MAP>), dim3(grid), dim3(256), 0, 0, g_planar, g_views, g_o, g_pitch, g_padx, tiles_x, n_tiles); });
    const double out_b = OUT == OUT_RGBA ? 4.0 : (OUT == OUT_NONE ? 0.0 : 3.0);
    const double moved = 1.0 * W * H * (3.0 * NIMG + out_b * NV);
    if(!quiet)
        printf("%-46s wg/cu %d  %8.1f us  %7.0f GB/s moved (%.0f MB)  B_alg rate %6.0f GB/s\n", name, wgs_per_cu, ms * 1e3, moved / ms / 1e6, moved / 1e6,
               4.0 * W * H * (NIMG + NV) / ms / 1e6);
    return ms;
}

template <int TW, bool WRITE>
float run_wide(const char *name, int wgs_per_cu)
{
    const int tiles_x = W / TW, n_tiles = tiles_x * H; // 1920 = 15·128 = 7.5·256: the ragged half tile is dropped (a bandwidth probe)
    const int grid = std::min(n_tiles, wgs_per_cu * g_cus);
    float ms = time_it([&] { hipLaunchKernelGGL((k_wide<TW, WRITE>), dim3(grid), dim3(256), 0, 0, g_planar, g_views, g_o, g_pitch, g_padx, tiles_x, n_tiles); });
    const double moved = 1.0 * tiles_x * TW * H * (3.0 * NIMG + (WRITE ? 3.0 * NV : 0.0));
    printf("%-46s wg/cu %d  %8.1f us  %7.0f GB/s moved (%.0f MB)\n", name, wgs_per_cu, ms * 1e3, moved / ms / 1e6, moved / 1e6);
    return ms;
}

int main(int argc, char **argv)
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    g_cus = prop.multiProcessorCount;
    for(int col = 0; col < 8; col++) for(int row = 0; row < 8; row++) { int g = col * 8 + row; g_o.ox[g] = (int)lroundf((3.5f - col) / 8 * 1920 * 0.23f); g_o.oy[g] = (int)lroundf((3.5f - row) / 8 * 1080 * 0.997f * 0.23f); }
    int reach = 0;
    for(int g = 0; g < 64; g++) reach = std::max(reach, std::abs(g_o.ox[g]));
    g_padx = (reach + 128 + 3) / 4 * 4;
    g_pitch = (W + 2 * g_padx + 15) / 16 * 16;
    const size_t planar_bytes = (size_t)NIMG * 12 * H * g_pitch;
    uint8_t *planar;
    CK(hipMalloc(&planar, planar_bytes)); CK(hipMalloc(&g_views, PLANE * 4 * NV));
    CK(hipMemset(planar, 1, planar_bytes)); CK(hipMemset(g_views, 2, PLANE * 4 * NV));
    g_planar = planar;
    if(argc > 1 && std::string(argv[1]) == "p64")
    {
        // half-line stores (a wave = 64 pixels x 32 views) against whole-line stores, in ordinary and in uncached device memory
        uint8_t *cached = g_views, *uncached = nullptr;
        CK(hipExtMallocWithFlags(reinterpret_cast<void **>(&uncached), PLANE * 4 * NV, hipDeviceMallocUncached));
        for(int round = 0; round < 3; round++)
            for(int mem = 0; mem < 2; mem++)
            {
                g_views = mem ? uncached : cached;
                printf("-- round %d, views in %s memory\n", round, mem ? "uncached" : "ordinary");
                run<OUT_PLANAR, true, 0>("planar out, 128 B per row and instruction, nt", 2);
                run<OUT_PLANAR64, true, 0>("planar out, 64 B per row and instruction, nt", 2);
                run<OUT_PLANAR64, false, 0>("planar out, 64 B per row and instruction, plain", 2);
            }
        return 0;
    }
    printf("planar copy %.2f GB, pitch %d, padx %d, %d CUs\n", planar_bytes / 1e9, g_pitch, g_padx, g_cus);
    for(int round = 0; round < 2; round++)
        for(int wpc : {2, 4, 8})
        {
            run_wide<128, false>("runs 128 B, read only", wpc);
            run_wide<256, false>("runs 256 B, read only", wpc);
            run_wide<512, false>("runs 512 B, read only", wpc);
            run_wide<128, true>("runs 128 B, read + planar write", wpc);
            run_wide<256, true>("runs 256 B, read + planar write", wpc);
            run_wide<512, true>("runs 512 B, read + planar write", wpc);
        }
    for(int round = 0; round < 1; round++)
    {
        printf("-- round %d\n", round);
        for(int wpc : {2, 4, 8})
        {
            run<OUT_NONE, true, 0>("read only", wpc);
            run<OUT_RGBA, true, 0>("rgba out, nt", wpc);
            run<OUT_RGBA, true, 1>("rgba out, nt, xcd ranges", wpc);
            run<OUT_PLANAR, true, 0>("planar out, nt", wpc);
            run<OUT_PLANAR, false, 0>("planar out, plain stores", wpc);
            run<OUT_PLANAR, true, 1>("planar out, nt, xcd ranges", wpc);
            run<OUT_RGB24, true, 0>("rgb24 out (3 x dwordx2 per lane), nt", wpc);
            run<OUT_RGB24, false, 0>("rgb24 out (3 x dwordx2 per lane), plain", wpc);
            run<OUT_RGB24_X3, true, 0>("rgb24 out (2 x dwordx3 per lane), nt", wpc);
            run<OUT_RGB24, true, 1>("rgb24 out (3 x dwordx2), nt, xcd ranges", wpc);
        }
    }
    return 0;
}
