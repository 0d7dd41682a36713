// Round 4, VERDICT r3 item 2: does a range freed as hipDeviceMallocUncached memory misbehave when the runtime recycles it for an ordinary
// allocation?  Cycle: uncached alloc -> kernel fills it with non-temporal stores -> free -> ordinary alloc of a similar size (same range if the
// runtime recycles it) -> a kernel writes a pattern with plain stores -> another kernel (all CUs) reads it back and counts mismatches, and the
// host reads it with hipMemcpy.  Prints whether the range was recycled and the mismatch counts.
// build: hipcc -O2 --offload-arch=gfx950 -o tools/probe_uncached_recycle tools/probe_uncached_recycle.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while(0)
__global__ void fill_nt(uint32_t *p, size_t n, uint32_t seed) { for(size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(uint32_t(i) * 2654435761u + seed, p + i); }
__global__ void fill_plain(uint32_t *p, size_t n, uint32_t seed) { for(size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = uint32_t(i) * 2246822519u + seed; }
__global__ void check(const uint32_t *p, size_t n, uint32_t seed, unsigned long long *bad) { unsigned long long b = 0; for(size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b += p[i] != uint32_t(i) * 2246822519u + seed; if(b) atomicAdd(bad, b); }
int main()
{
    unsigned long long *bad; CK(hipMalloc((void **)&bad, 8));
    const size_t sizes[] = {size_t(64) << 20, size_t(398) << 20, size_t(1) << 30};
    for(int round = 0; round < 6; round++)
        for(size_t bytes : sizes)
        {
            const size_t n = bytes / 4;
            uint32_t *u = nullptr, *o = nullptr;
            CK(hipExtMallocWithFlags((void **)&u, bytes, hipDeviceMallocUncached));
            fill_nt<<<2048, 256>>>(u, n, 17u + round);
            CK(hipDeviceSynchronize());
            CK(hipFree(u));
            CK(hipMalloc((void **)&o, bytes - 4096 * (round & 1)));
            const size_t n2 = (bytes - 4096 * (round & 1)) / 4;
            std::vector<uint32_t> host(n2);
            for(size_t i = 0; i < n2; i++) host[i] = uint32_t(i) * 2246822519u + 99u + round;
            // host -> device copy (as the library uploads inputs and parameters), then a device check; then a kernel fill and a host check
            CK(hipMemcpy(o, host.data(), n2 * 4, hipMemcpyHostToDevice));
            CK(hipMemset(bad, 0, 8));
            check<<<2048, 256>>>(o, n2, 99u + round, bad);
            unsigned long long b1 = 0; CK(hipMemcpy(&b1, bad, 8, hipMemcpyDeviceToHost));
            fill_plain<<<2048, 256>>>(o, n2, 5u + round);
            CK(hipMemset(bad, 0, 8));
            check<<<2048, 256>>>(o, n2, 5u + round, bad);
            unsigned long long b2 = 0; CK(hipMemcpy(&b2, bad, 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(host.data(), o, n2 * 4, hipMemcpyDeviceToHost));
            size_t b3 = 0; for(size_t i = 0; i < n2; i++) b3 += host[i] != uint32_t(i) * 2246822519u + 5u + round;
            printf("round %d %5zu MB: recycled %s; mismatches after H2D+device check %llu, after kernel fill+device check %llu, host read-back %zu\n", round, bytes >> 20,
                   (void *)o == (void *)u ? "SAME range" : "other range", b1, b2, b3);
            CK(hipFree(o));
        }
    return 0;
}
