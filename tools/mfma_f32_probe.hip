// probe: sustained rate of v_mfma_f32_32x32x2_f32 streams shaped like the STD kernel's inner loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while(0)

// MODE 0: 6 independent accumulators, operands fixed registers
// MODE 1: like the kernel: B operand produced by v_cvt_f32_ubyte from a dword right before each MFMA pair
template <int MODE, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k(const uint32_t *in, float *out, int iters)
{
    f32x16 acc[6];
    for(int i = 0; i < 6; i++) for(int e = 0; e < 16; e++) acc[i][e] = 0.f;
    float w0 = in[threadIdx.x] * 1e-9f, w1 = in[threadIdx.x + 64] * 1e-9f;
    uint32_t p = in[threadIdx.x + 128];
    for(int it = 0; it < iters; it++)
    {
#pragma unroll
        for(int q = 0; q < 8; q++)
        {
            if(MODE == 0)
            {
                float b = __builtin_bit_cast(float, p);
#pragma unroll
                for(int c = 0; c < 3; c++)
                {
                    acc[2 * c] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, b, acc[2 * c], 0, 0, 0);
                    acc[2 * c + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, b, acc[2 * c + 1], 0, 0, 0);
                }
            }
            else
            {
                uint32_t pp = p + q + it;
                asm volatile("" : "+v"(pp));
#pragma unroll
                for(int c = 0; c < 3; c++)
                {
                    float b = static_cast<float>((pp >> (8 * c)) & 0xffu);
                    acc[2 * c] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, b, acc[2 * c], 0, 0, 0);
                    acc[2 * c + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, b, acc[2 * c + 1], 0, 0, 0);
                }
            }
        }
    }
    float s = 0;
    for(int i = 0; i < 6; i++) for(int e = 0; e < 16; e++) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int WAVES>
void run(const char *name, const uint32_t *in, float *out, int blocks)
{
    const int iters = 2000;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<MODE, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, in, out, 10);
    CK(hipDeviceSynchronize());
    std::vector<float> t;
    for(int r = 0; r < 5; r++) { CK(hipEventRecord(a)); hipLaunchKernelGGL((k<MODE, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, in, out, iters); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
    std::sort(t.begin(), t.end());
    double mfma_per_wave = (double)iters * 48;
    double flops = mfma_per_wave * 2.0 * 32 * 32 * 2 * blocks * WAVES;
    printf("%-40s %8.3f ms  %7.1f TFLOP/s  (%.1f cycles/MFMA/SIMD @2.4GHz)\n", name, t[2], flops / t[2] / 1e9,
           t[2] * 1e-3 * 2.4e9 / (mfma_per_wave * blocks * WAVES / 1024.0));
}

int main()
{
    uint32_t *in; float *out;
    CK(hipMalloc(&in, 4096)); CK(hipMalloc(&out, 4 * 1024 * 1024)); CK(hipMemset(in, 1, 4096));
    run<0, 4>("plain, 1 wave/SIMD (256 WG x 4 waves)", in, out, 256);
    run<0, 4>("plain, 2 waves/SIMD", in, out, 512);
    run<1, 4>("cvt before each pair, 1 wave/SIMD", in, out, 256);
    run<1, 4>("cvt before each pair, 2 waves/SIMD", in, out, 512);
    run<1, 4>("cvt before each pair, 4 waves/SIMD", in, out, 1024);
    return 0;
}
