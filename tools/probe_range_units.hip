// probe (round 5): the unit costs behind a range pass (focus_factored.hpp) whose samples sit in LDS as u16 lanes.
//   A  ds_read_b32 / b64 / b128 at byte-misaligned LDS addresses: legal?  which bytes come back?
//   B  … and their throughput, aligned against misaligned (256 threads, several workgroups per CU)
//   C  issue rate of v_pk_minimum3_f16 / v_pk_maximum3_f16 and of v_perm_b32
//   D  LDS-DMA (global_load_lds_dwordx4) of patches that hit in L2: bytes per clock and CU
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe_range_units tools/probe_range_units.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t lds_addr_of(const void *p)
{
    return uint32_t(uintptr_t((__attribute__((address_space(3))) void *)p));
}

// A: one wave; lds[i] = byte pattern; lane reads W bytes at 16·lane + mis
__global__ void check(uint8_t *out, int mis, int width)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[2048];
    const int lane = threadIdx.x;
    for(int i = lane; i < 2048; i += 64)
        lds[i] = uint8_t((i * 7 + (i >> 8)) & 255);
    __syncthreads();
    const uint32_t a = lds_addr_of(lds) + 16u * lane + mis;
    u32x4 v = {0, 0, 0, 0};
    if(width == 4)
    {
        uint32_t r;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
        v.x = r;
    }
    else if(width == 8)
    {
        u32x2 r;
        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
        v.x = r.x, v.y = r.y;
    }
    else
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    reinterpret_cast<u32x4 *>(out)[lane] = v;
}

// B: throughput.  Every lane reads `width` bytes at stride·lane + mis + row offsets; 16 reads in flight per iteration
template <int WIDTH>
__global__ void __launch_bounds__(256) lds_rate(uint32_t *sink, int iters, int mis, int stride)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[48 * 1024];
    for(int i = threadIdx.x; i < 48 * 1024 / 4; i += 256)
        reinterpret_cast<uint32_t *>(lds)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t a = lds_addr_of(lds) + uint32_t(stride) * lane + mis + wave * 8192u;
    uint32_t acc = 0;
    for(int it = 0; it < iters; it++)
    {
        if constexpr(WIDTH == 4)
        {
            uint32_t r[16];
#pragma unroll
            for(int j = 0; j < 16; j++)
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[j]) : "v"(a), "n"(j * 512) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for(int j = 0; j < 16; j++)
                acc ^= r[j];
        }
        else if constexpr(WIDTH == 8)
        {
            u32x2 r[16];
#pragma unroll
            for(int j = 0; j < 16; j++)
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[j]) : "v"(a), "n"(j * 512) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for(int j = 0; j < 16; j++)
                acc ^= r[j].x ^ r[j].y;
        }
        else
        {
            u32x4 r[8];
#pragma unroll
            for(int j = 0; j < 8; j++)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[j]) : "v"(a), "n"(j * 1024) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for(int j = 0; j < 8; j++)
                acc ^= r[j].x ^ r[j].y ^ r[j].z ^ r[j].w;
        }
    }
    if(acc == 0x12345678u)
        sink[0] = acc;
}

// C: VALU issue rate: 24 independent accumulators, min3 / max3 chains (KIND 0), v_perm_b32 (KIND 1), v_pk_min_u16 (KIND 2)
template <int KIND>
__global__ void __launch_bounds__(256) valu_rate(uint32_t *sink, int iters, uint32_t seed)
{
    uint32_t acc[24];
#pragma unroll
    for(int j = 0; j < 24; j++)
        acc[j] = (seed * (j + 1) + threadIdx.x) & 0x00ff00ffu;
    uint32_t a = (seed ^ threadIdx.x) & 0x00ff00ffu, b = (seed * 3 + threadIdx.x) & 0x00ff00ffu;
    for(int it = 0; it < iters; it++)
    {
#pragma unroll
        for(int j = 0; j < 24; j++)
        {
            if constexpr(KIND == 0)
            {
                if(j & 1)
                    asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(acc[j]) : "v"(a), "v"(b));
                else
                    asm volatile("v_pk_minimum3_f16 %0, %0, %1, %2" : "+v"(acc[j]) : "v"(a), "v"(b));
            }
            else if constexpr(KIND == 1)
                asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(acc[j]) : "v"(a), "v"(b));
            else
                asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(acc[j]) : "v"(a));
        }
    }
    uint32_t x = 0;
#pragma unroll
    for(int j = 0; j < 24; j++)
        x ^= acc[j];
    if(x == 0x12345678u)
        sink[0] = x;
}

// D: LDS-DMA of patches that hit in L2.  A workgroup fetches `pieces` 1-KB pieces per round (wave w: pieces w, w + 4, …) from a region of
// `region` bytes that all workgroups share (offset by the workgroup and round, wrapped), waits, and goes on.
__device__ __forceinline__ void dma16_s(const uint8_t *base, uint32_t voff, uint32_t lds)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(base), "v"(voff), "s"(lds) : "memory");
}
__global__ void __launch_bounds__(256) dma_rate(const uint8_t *src, uint32_t region, int rounds, int pieces, int row_bytes, int pitch, uint32_t *sink)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[48 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t base = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
    for(int r = 0; r < rounds; r++)
    {
        const uint32_t origin = (uint32_t(blockIdx.x) * 7919u * 64u + uint32_t(r) * 104729u * 16u) % (region - uint32_t(pitch) * 64u - 65536u);
        for(int p = wave; p < pieces; p += 4)
        {
            // piece p, lane l: bytes 1024p + 16l of a patch whose rows are row_bytes long in LDS and `pitch` apart in memory
            const uint32_t b = 1024u * p + 16u * lane;
            const uint32_t voff = (b / uint32_t(row_bytes)) * uint32_t(pitch) + b % uint32_t(row_bytes);
            dma16_s(src + origin, voff, base + 1024u * p);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if(threadIdx.x == 0)
        sink[blockIdx.x] = lds[blockIdx.x & 1023];
}


// E: typed buffer loads: buffer_load_format_d16_xyzw through a descriptor with DATA_FORMAT 8_8_8_8 / NUM_FORMAT UINT — does the texture
// path unpack an RGBA8 pixel into four u16 (two VGPRs) for free?
typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4s make_rsrc(const void *base, uint32_t bytes, uint32_t word3)
{
    const uint64_t p = reinterpret_cast<uint64_t>(base);
    u32x4s r;
    r.x = __builtin_amdgcn_readfirstlane(uint32_t(p));
    r.y = __builtin_amdgcn_readfirstlane(uint32_t(p >> 32) & 0xffffu); // stride 0
    r.z = __builtin_amdgcn_readfirstlane(bytes);
    r.w = __builtin_amdgcn_readfirstlane(word3);
    return r;
}
__global__ void typed_check(const uint8_t *src, uint32_t *out, uint32_t word3, uint32_t soff)
{
    const int lane = threadIdx.x;
    const u32x4s rs = make_rsrc(src, 1u << 20, word3);
    u32x2 v;
    const uint32_t voff = 4u * lane;
    asm volatile("buffer_load_format_d16_xyzw %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(voff), "s"(rs), "s"(soff) : "memory");
    out[2 * lane] = v.x;
    out[2 * lane + 1] = v.y;
}
// rate: every lane loads one pixel per instruction, 16 in flight, rows of a patch 4224 bytes apart, from an L2-resident region
template <int KIND>
__global__ void __launch_bounds__(256) typed_rate(const uint8_t *src, uint32_t region, int iters, uint32_t word3, uint32_t *sink)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u32x4s rs = make_rsrc(src, region, word3);
    uint32_t acc = 0;
    const uint32_t voff = 4u * lane;
    for(int it = 0; it < iters; it++)
    {
        const uint32_t origin = ((uint32_t(blockIdx.x) * 7919u + uint32_t(it) * 104729u + wave * 977u) * 64u) % (region - 4224u * 20u);
        if constexpr(KIND == 0)
        {
            u32x2 v[16];
#pragma unroll
            for(int j = 0; j < 16; j++)
                asm volatile("buffer_load_format_d16_xyzw %0, %1, %2, %3 offen" : "=v"(v[j]) : "v"(voff), "s"(rs), "s"(origin + 4224u * j) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for(int j = 0; j < 16; j++)
                acc ^= v[j].x ^ v[j].y;
        }
        else
        {
            uint32_t v[16];
#pragma unroll
            for(int j = 0; j < 16; j++)
                asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(v[j]) : "v"(voff), "s"(rs), "s"(origin + 4224u * j) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for(int j = 0; j < 16; j++)
                acc ^= v[j];
        }
    }
    if(acc == 0x12345678u)
        sink[0] = acc;
}

template <class F>
float time_ms(F f, int reps = 5)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    float best = 1e30f;
    for(int i = 0; i < reps; i++)
    {
        hipEventRecord(e0);
        f();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    uint8_t *d_out;
    hipMalloc(&d_out, 4096);
    std::vector<uint8_t> h(1024);
    auto pat = [](int i) { return uint8_t((i * 7 + (i >> 8)) & 255); };
    printf("A: misaligned LDS reads (mismatching lanes of 64)\n");
    for(int width : {4, 8, 16})
        for(int mis = 0; mis < 16; mis++)
        {
            hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, d_out, mis, width);
            if(hipDeviceSynchronize() != hipSuccess)
            {
                printf("   width %2d mis %2d: FAULT %s\n", width, mis, hipGetErrorString(hipGetLastError()));
                return 1;
            }
            hipMemcpy(h.data(), d_out, 1024, hipMemcpyDeviceToHost);
            int bad = 0;
            for(int l = 0; l < 64; l++)
            {
                bool ok = true;
                for(int b = 0; b < width; b++)
                    ok = ok && h[16 * l + b] == pat(16 * l + mis + b);
                bad += !ok;
            }
            printf("   width %2d mis %2d: %2d lanes differ%s", width, mis, bad, bad ? "  lane 1 got:" : "\n");
            if(bad)
            {
                for(int b = 0; b < width; b++)
                    printf(" %02x", h[16 + b]);
                printf("  want:");
                for(int b = 0; b < width; b++)
                    printf(" %02x", pat(16 + mis + b));
                printf("\n");
            }
        }
    uint32_t *sink;
    hipMalloc(&sink, 4 * 65536);
    const double clk = prop.clockRate * 1e3; // Hz (peak)
    printf("B: LDS read throughput, bytes per clock (at the peak clock %.2f GHz) and CU; workgroups per CU 1 / 2 / 3\n", clk * 1e-9);
    for(int width : {4, 8, 16})
        for(int stride : {width, 16})
            for(int mis : {0, 2, 4, 6, 8})
            {
                if(stride == 16 && width == 16 && mis == 0)
                    continue;
                printf("   width %2d lane stride %2d mis %d:", width, stride, mis);
                for(int wgs = 1; wgs <= 3; wgs++)
                {
                    const int iters = 4000;
                    float ms = time_ms([&] {
                        if(width == 4)
                            hipLaunchKernelGGL(lds_rate<4>, dim3(cus * wgs), dim3(256), 0, 0, sink, iters, mis, stride);
                        else if(width == 8)
                            hipLaunchKernelGGL(lds_rate<8>, dim3(cus * wgs), dim3(256), 0, 0, sink, iters, mis, stride);
                        else
                            hipLaunchKernelGGL(lds_rate<16>, dim3(cus * wgs), dim3(256), 0, 0, sink, iters, mis, stride);
                    });
                    const double bytes = double(iters) * (width == 16 ? 8 : 16) * width * 256.0 * wgs; // per CU
                    printf("  %6.1f", bytes / (ms * 1e-3 * clk));
                }
                printf("\n");
            }
    printf("C: VALU issue: cycles per wave-instruction and SIMD (2 workgroups per CU = 2 waves per SIMD)\n");
    for(int kind = 0; kind < 3; kind++)
    {
        const int iters = 20000;
        float ms = time_ms([&] {
            if(kind == 0)
                hipLaunchKernelGGL(valu_rate<0>, dim3(cus * 2), dim3(256), 0, 0, sink, iters, 12345u);
            else if(kind == 1)
                hipLaunchKernelGGL(valu_rate<1>, dim3(cus * 2), dim3(256), 0, 0, sink, iters, 12345u);
            else
                hipLaunchKernelGGL(valu_rate<2>, dim3(cus * 2), dim3(256), 0, 0, sink, iters, 12345u);
        });
        const double instr_per_simd = double(iters) * 24 * 2; // two waves per SIMD
        printf("   %s: %.2f cycles (peak clock)\n", kind == 0 ? "v_pk_minimum3/maximum3_f16" : kind == 1 ? "v_perm_b32" : "v_pk_min_u16", ms * 1e-3 * clk / instr_per_simd);
    }
    printf("D: LDS-DMA from an L2 / MALL resident region: bytes per clock and CU (peak clock), TB/s chip-wide\n");
    uint8_t *src;
    const uint32_t region_max = 256u << 20;
    hipMalloc(&src, region_max);
    hipMemset(src, 1, region_max);
    for(uint32_t region : {2u << 20, 16u << 20, 256u << 20})
        for(int wgs : {1, 2, 3})
            for(int row_bytes : {1024, 1136, 2048})
            {
                const int pieces = 32, rounds = 400;
                const int pitch = region < (8u << 20) ? 4224 : 33024; // (64 rows of the patch + 64 KB must fit the region)
                float ms = time_ms([&] { hipLaunchKernelGGL(dma_rate, dim3(cus * wgs), dim3(256), 0, 0, src, region, rounds, pieces, row_bytes, pitch, sink); });
                const double bytes_cu = double(rounds) * pieces * 1024.0 * wgs;
                printf("   region %3u MB, %d WG/CU, patch rows of %4d B: %6.1f B/clk/CU  %5.1f TB/s\n", region >> 20, wgs, row_bytes, bytes_cu / (ms * 1e-3 * clk),
                       bytes_cu * cus / (ms * 1e-3) * 1e-12);
            }
    printf("E: typed buffer loads (buffer_load_format_d16_xyzw, descriptor word3 = 8_8_8_8 UINT)\n");
    {
        std::vector<uint8_t> hb(1 << 20);
        for(size_t i = 0; i < hb.size(); i++)
            hb[i] = uint8_t((i * 37 + (i >> 9) * 11) & 255);
        hipMemcpy(src, hb.data(), hb.size(), hipMemcpyHostToDevice);
        uint32_t *d32;
        hipMalloc(&d32, 4 * 128);
        for(uint32_t word3 : {0x54FACu, 0x00054FACu | (1u << 24)})
            for(uint32_t soff : {0u, 4u * 1000u, 4u * 1000u + 1u, 4u * 1000u + 2u})
            {
                hipLaunchKernelGGL(typed_check, dim3(1), dim3(64), 0, 0, src, d32, word3, soff);
                if(hipDeviceSynchronize() != hipSuccess)
                {
                    printf("   word3 %08x soffset %u: FAULT\n", word3, soff);
                    return 1;
                }
                uint32_t ho[128];
                hipMemcpy(ho, d32, sizeof(ho), hipMemcpyDeviceToHost);
                int bad = 0;
                for(int l = 0; l < 64; l++)
                {
                    const uint8_t *p = &hb[soff + 4 * l];
                    bad += ho[2 * l] != (uint32_t(p[0]) | uint32_t(p[1]) << 16) || ho[2 * l + 1] != (uint32_t(p[2]) | uint32_t(p[3]) << 16);
                }
                printf("   word3 %08x soffset %5u: %2d lanes differ; lane 1 got %08x %08x, bytes %02x %02x %02x %02x\n", word3, soff, bad, ho[2], ho[3], hb[soff + 4], hb[soff + 5],
                       hb[soff + 6], hb[soff + 7]);
            }
        hipMemset(src, 1, region_max);
        for(uint32_t region : {2u << 20, 64u << 20})
            for(int kind = 0; kind < 2; kind++)
                for(int wgs : {1, 2, 3})
                {
                    const int iters = 2000;
                    float ms = time_ms([&] {
                        if(kind == 0)
                            hipLaunchKernelGGL(typed_rate<0>, dim3(cus * wgs), dim3(256), 0, 0, src, region, iters, 0x54FACu, sink);
                        else
                            hipLaunchKernelGGL(typed_rate<1>, dim3(cus * wgs), dim3(256), 0, 0, src, region, iters, 0x00020000u, sink);
                    });
                    const double px_cu = double(iters) * 16 * 256.0 * wgs;
                    printf("   %s, region %2u MB, %d WG/CU: %5.2f pixels per clock and CU (%5.1f B/clk of memory)\n", kind == 0 ? "format_d16_xyzw" : "buffer_load_dword", region >> 20, wgs,
                           px_cu / (ms * 1e-3 * clk), 4 * px_cu / (ms * 1e-3 * clk));
                }
    }
    return 0;
}
