// probe (round 3): LDS-DMA (global_load_lds_dwordx4) from BYTE-aligned global addresses — is it legal, where do the bytes land, and what
// does it cost?  The derived planar input copy keeps four byte-shifted copies of every plane (12 B per pixel·image) only because the
// DMA sources were assumed to need dword alignment; if a run that starts at an arbitrary byte can be fetched directly, ONE copy (3 B)
// serves.   build: hipcc --offload-arch=gfx950 -O3 -o tools/probe_ldsdma_bytes tools/probe_ldsdma_bytes.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void dma16(const void *gptr, uint32_t lds_addr)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(__builtin_amdgcn_readfirstlane(lds_addr)) : "memory");
}

// correctness: one wave, lanes read 16 B each at src + shift + 16·lane (8 lanes = one 128-byte run, as the blend kernels read)
__global__ void check(const uint8_t *src, uint8_t *dst, int shift)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[1024];
    const int lane = threadIdx.x;
    for(int i = lane; i < 256; i += 64)
        reinterpret_cast<uint32_t *>(lds)[i] = 0xdeadbeefu;
    __syncthreads();
    dma16(src + shift + 16 * lane, uint32_t(uintptr_t((__attribute__((address_space(3))) void *)lds)));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for(int i = lane; i < 1024; i += 64)
        dst[i] = lds[i];
}

// bandwidth: persistent workgroups of four waves; every wave fetches 1-KB pieces = eight 128-byte runs, each run from its own "plane row"
// (rows `pitch` bytes apart, planes far apart — the access pattern of blend_p3), runs starting at byte x0 + shift; no arithmetic.
__global__ void __launch_bounds__(256, 2) stream(const uint8_t *src, size_t plane_bytes, int pitch, int rows, int planes, int shift, uint32_t *sink)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[3 * 24 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t base = uint32_t(uintptr_t((__attribute__((address_space(3))) void *)lds));
    const int tiles_x = pitch / 128 - 1, n_tiles = tiles_x * rows;
    int buf = 0;
    for(int t = blockIdx.x; t < n_tiles; t += gridDim.x)
    {
        const int y = t / tiles_x, x0 = (t - y * tiles_x) * 128;
        // 24 pieces per tile (64 images × 3 channels / 8 rows per piece): wave w takes pieces w, w + 4, …
        for(int p = wave; p < 24; p += 4)
        {
            const int plane = (p * 8 + (lane >> 3)) % planes;
            const uint8_t *g = src + (size_t)plane * plane_bytes + (size_t)y * pitch + x0 + shift + 16 * (lane & 7);
            dma16(g, base + uint32_t(buf * 24 + p) * 1024u);
        }
        buf = buf == 2 ? 0 : buf + 1;
        if(buf == 0)
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if(threadIdx.x == 0)
        sink[blockIdx.x] = lds[blockIdx.x & 1023];
}

int main()
{
    const int N = 1 << 16;
    std::vector<uint8_t> h(N);
    for(int i = 0; i < N; i++)
        h[i] = uint8_t((i * 7 + (i >> 8)) & 0xff);
    uint8_t *s, *d;
    hipMalloc(&s, N);
    hipMalloc(&d, 1024);
    hipMemcpy(s, h.data(), N, hipMemcpyHostToDevice);
    std::vector<uint8_t> o(1024);
    for(int shift = 0; shift < 8; shift++)
    {
        hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, s, d, shift);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(o.data(), d, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for(int i = 0; i < 1024; i++)
            bad += o[i] != h[shift + i];
        printf("check: byte shift %d: err=%d, %d of 1024 bytes differ from src[shift + i]  (first: %02x %02x %02x %02x | want %02x %02x %02x %02x)\n", shift, (int)e,
               bad, o[0], o[1], o[2], o[3], h[shift], h[shift + 1], h[shift + 2], h[shift + 3]);
    }
    // 192 planes of 1080 rows × 2304 bytes (config 2's planar copy has pitch 2336): 478 MB per pass
    const int pitch = 2304, rows = 1080, planes = 192;
    const size_t plane_bytes = (size_t)pitch * rows + 4096;
    uint8_t *big;
    uint32_t *sink;
    hipMalloc(&big, plane_bytes * planes + 4096);
    hipMalloc(&sink, 4096 * 4);
    hipMemset(big, 1, plane_bytes * planes + 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double bytes = double(pitch / 128 - 1) * rows * 24.0 * 1024.0;
    for(int round = 0; round < 2; round++)
        for(int shift = 0; shift < 5; shift++)
        {
            for(int i = 0; i < 3; i++)
                hipLaunchKernelGGL(stream, dim3(512), dim3(256), 0, 0, big, plane_bytes, pitch, rows, planes, shift, sink);
            hipEventRecord(e0);
            for(int i = 0; i < 10; i++)
                hipLaunchKernelGGL(stream, dim3(512), dim3(256), 0, 0, big, plane_bytes, pitch, rows, planes, shift, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("stream: byte shift %d: %.1f us per pass of %.0f MB = %.0f GB/s\n", shift, ms * 100.0, bytes / 1e6, bytes / (ms / 10.0) / 1e6);
        }
    return 0;
}
