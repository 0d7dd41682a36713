// probe: does LDS-DMA (global_load_lds) of 4/16 bytes per lane work from dword-aligned (not 16-B aligned) global addresses,
// and where do the bytes land in LDS?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

template <int SIZE>
__global__ void k(const uint32_t *src, uint32_t *dst, int shift_dw, int per_lane_stride_dw)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[64 * 4 * 2];
    const int lane = threadIdx.x;
    for(int i = lane; i < 64 * 4 * 2; i += 64) lds[i] = 0xdeadbeef;
    __syncthreads();
    const uint32_t *p = src + shift_dw + lane * per_lane_stride_dw;
    if constexpr(SIZE == 4) __builtin_amdgcn_global_load_lds(p, (__attribute__((address_space(3))) void *)lds, 4, 0, 0); else __builtin_amdgcn_global_load_lds(p, (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for(int i = lane; i < 64 * 4 * 2; i += 64) dst[i] = lds[i];
}

int main()
{
    const int N = 4096;
    std::vector<uint32_t> h(N);
    for(int i = 0; i < N; i++) h[i] = i;
    uint32_t *s, *d;
    hipMalloc(&s, N * 4); hipMalloc(&d, 512 * 4);
    hipMemcpy(s, h.data(), N * 4, hipMemcpyHostToDevice);
    std::vector<uint32_t> o(512);
    for(int size : {4, 16})
        for(int shift : {0, 1, 2, 3, 5})
        {
            int stride = size / 4;
            if(size == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, s, d, shift, stride);
            else hipLaunchKernelGGL(k<16>, dim3(1), dim3(64), 0, 0, s, d, shift, stride);
            hipError_t e = hipDeviceSynchronize();
            hipMemcpy(o.data(), d, 512 * 4, hipMemcpyDeviceToHost);
            int bad = 0;
            for(int i = 0; i < 64 * stride; i++) if(o[i] != (uint32_t)(shift + i)) bad++;
            printf("size %2d shift %d dwords: err=%d bad=%d  first: %u %u %u %u %u %u\n", size, shift, (int)e, bad, o[0], o[1], o[2], o[3], o[4], o[5]);
        }
    // gather form: lanes 0-31 from one row, lanes 32-63 from another, each shifted
    return 0;
}
