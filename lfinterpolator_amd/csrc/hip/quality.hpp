// quality.hpp — PSNR / SSIM of a rendered view against a reference image, reduced on the device.
//
// Replaces scripts/imageQualityMetrics.sh (reference scripts/imageQualityMetrics.sh:1-12: ffmpeg's psnr and ssim filters on two
// PNGs; scripts/compareDirs.sh loops it over directories) — the offline check the reference's authors run on rendered views.
// ffmpeg is not part of this product, so the definitions are fixed here (and restated in numpy by the tests):
//   PSNR  per colour channel c: 10·log10(255² / MSE_c), MSE_c over all pixels; "all" from the mean of the three MSEs;
//   SSIM  per colour channel: the mean over all 8×8 windows at a stride of 4 pixels (the window set of ffmpeg's ssim filter) of
//         ((2·μa·μb + C1)(2·σab + C2)) / ((μa² + μb² + C1)(σa² + σb² + C2)), C1 = (0.01·255)², C2 = (0.03·255)², with the
//         window's biased moments (sums over its 64 pixels); "all" = the mean of the three channels.  Alpha is ignored.
// Squared errors are summed exactly in 64-bit integers; window SSIMs are summed in fp64 (atomic order may differ in the last
// bits between runs).
#pragma once

#include "lfi_device.hpp"

namespace lfi {

struct QualitySums
{
    unsigned long long sq_err[3]; // Σ (a − b)² per channel
    double ssim[3];               // Σ over windows
    unsigned long long windows;
};

// one thread per 8×8 window (stride 4) for SSIM; the same thread adds the squared error of the window's top-left 4×4 block
// (blocks tile the image; right / bottom remainders are handled by the edge threads)
__global__ void __launch_bounds__(256) quality_reduce(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, const int W, const int H,
                                                      QualitySums *__restrict__ out)
{
    const int bx = blockIdx.x * 16 + (threadIdx.x & 15), by = blockIdx.y * 16 + (threadIdx.x >> 4); // 4×4-block coordinates
    const int x0 = bx * 4, y0 = by * 4;
    unsigned long long se[3] = {0ull, 0ull, 0ull};
    double ss[3] = {0.0, 0.0, 0.0};
    unsigned long long nwin = 0ull;
    if(x0 < W && y0 < H)
    {
        // squared error of this thread's 4×4 block (clipped at the image border)
        for(int y = y0; y < min(y0 + 4, H); y++)
            for(int x = x0; x < min(x0 + 4, W); x++)
            {
                const uint32_t pa = a[(size_t)y * W + x], pb = b[(size_t)y * W + x];
#pragma unroll
                for(int c = 0; c < 3; c++)
                {
                    const int d = int((pa >> (8 * c)) & 0xffu) - int((pb >> (8 * c)) & 0xffu);
                    se[c] += (unsigned long long)(d * d);
                }
            }
        // SSIM of the 8×8 window whose top-left corner is this block, if it fits
        if(x0 + 8 <= W && y0 + 8 <= H)
        {
            uint32_t s1[3] = {0, 0, 0}, s2[3] = {0, 0, 0}, s11[3] = {0, 0, 0}, s22[3] = {0, 0, 0}, s12[3] = {0, 0, 0};
            for(int y = y0; y < y0 + 8; y++)
                for(int x = x0; x < x0 + 8; x++)
                {
                    const uint32_t pa = a[(size_t)y * W + x], pb = b[(size_t)y * W + x];
#pragma unroll
                    for(int c = 0; c < 3; c++)
                    {
                        const uint32_t va = (pa >> (8 * c)) & 0xffu, vb = (pb >> (8 * c)) & 0xffu;
                        s1[c] += va;
                        s2[c] += vb;
                        s11[c] += va * va;
                        s22[c] += vb * vb;
                        s12[c] += va * vb;
                    }
                }
            const double C1 = 0.01 * 255.0 * 0.01 * 255.0, C2 = 0.03 * 255.0 * 0.03 * 255.0;
#pragma unroll
            for(int c = 0; c < 3; c++)
            {
                const double mu1 = s1[c] / 64.0, mu2 = s2[c] / 64.0;
                const double var1 = s11[c] / 64.0 - mu1 * mu1, var2 = s22[c] / 64.0 - mu2 * mu2, cov = s12[c] / 64.0 - mu1 * mu2;
                ss[c] = ((2.0 * mu1 * mu2 + C1) * (2.0 * cov + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (var1 + var2 + C2));
            }
            nwin = 1ull;
        }
    }
    // wave reduction, then one atomic per wave and quantity
#pragma unroll
    for(int off = 32; off > 0; off >>= 1)
    {
#pragma unroll
        for(int c = 0; c < 3; c++)
        {
            se[c] += __shfl_down(se[c], off);
            ss[c] += __shfl_down(ss[c], off);
        }
        nwin += __shfl_down(nwin, off);
    }
    if((threadIdx.x & 63) == 0)
    {
#pragma unroll
        for(int c = 0; c < 3; c++)
        {
            atomicAdd(&out->sq_err[c], se[c]);
            atomicAdd(&out->ssim[c], ss[c]);
        }
        atomicAdd(&out->windows, nwin);
    }
}

} // namespace lfi
