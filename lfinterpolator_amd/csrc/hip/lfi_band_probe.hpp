// lfi_band_probe.hpp — the band method's MEASURED bound, checked on the device a context runs on (round 5).
//
// STD on more than 64 images (blend_stdx, blend_stdxa, blend_afs: Standard::process, reference src/kernels.cu:289-343, by fp16 matrix-core
// sums + the exact fmaf chain inside a band around x.5) sizes its band with N·2^-17 for the matrix pipe's accumulation error — a quarter
// ulp(512) per addend, a property MEASURED on gfx950 (lfi_device.hpp: std_accumulation_bound), not one that arithmetic guarantees.  So the
// first such launch on a device runs the measurement itself: chains of both MFMA shapes the kernels use, fed as the kernels feed them
// (weights × 2^15 as the A operand, pixel bytes as fp16 subnormals as B), over adversarial operand families (dominant + tiny, random
// exponents over 24 binades, ascending / descending magnitudes, sums near 510, … — the families of
// tests/test_gpu_parity.py::test_mfma_f16_accumulation_error_bound), exact sums in int64 on the host.  If any sum errs by more than the
// budget, every later launch on that device takes the ANALYTIC band (N·2^-15: true of any accumulator that keeps 24 bits; same bytes, more
// sums recomputed) and lfi_std_band_info says so.  One probe per device and process (≈ 1 ms); LFI_FLAG_STD_BAND_PROBE_FAIL makes a context
// behave as if its device had failed it (tests).
#pragma once

#include <chrono>
#include <mutex>

#include "lfi_context.hpp"
#include "blend_ten.hpp"

namespace {

struct BandProbe
{
    bool done = false, ok = false;
    float worst = 0.0f; // largest |error| seen, as a fraction of the budget K·2^-26 (on acc = S·2^-9)
    float ms = 0.0f;    // what the probe cost (host wall clock, once)
    int sums = 0;
};

constexpr int LFI_PROBE_MAX_DEVICES = 64;
BandProbe g_band_probe[LFI_PROBE_MAX_DEVICES];
std::mutex g_band_probe_mutex;

struct ProbeRng
{
    uint64_t s;
    uint64_t next()
    {
        s ^= s >> 12, s ^= s << 25, s ^= s >> 27;
        return s * 0x2545F4914F6CDD1Dull;
    }
    double uniform() { return double(next() >> 11) * 0x1p-53; }
    int below(int n) { return int(next() % uint64_t(n)); }
};

// one view's K weights in [0, 2), summing to at most 2 (the kernels' preconditions), as fp16 bit patterns of weight × 2^15
void probe_weight_row(ProbeRng &rng, const int K, const int kind, uint16_t *a_bits)
{
    const double full = 2047.0 / 2048.0;
    std::vector<double> w(K, 0.0);
    auto scale_to = [&](const double limit) {
        double sum = 0;
        for(double x : w)
            sum += x;
        if(sum > limit)
            for(double &x : w)
                x *= limit / sum;
    };
    switch(kind)
    {
        case 0: // one dominant weight, the rest 10 … 24 binades below with full mantissas
            for(double &x : w)
                x = full * std::ldexp(1.0, -(10 + rng.below(15)));
            w[rng.below(K)] = full;
            break;
        case 1: // a convex combination (what generateWeights produces, -s 7)
        {
            double sum = 0;
            for(double &x : w)
                sum += x = std::pow(rng.uniform(), 7.0);
            for(double &x : w)
                x /= sum;
            break;
        }
        case 2: // equal weights: every addend in one binade, carries ripple through the whole sum
            for(double &x : w)
                x = 1.0 / K;
            break;
        case 3: // sums up to 510: the top of the band's validity range
        {
            double sum = 0;
            for(double &x : w)
                sum += x = rng.uniform();
            for(double &x : w)
                x *= 1.999 / sum;
            break;
        }
        case 4: // a random exponent per addend over 24 binades, random mantissas
            for(double &x : w)
                x = (1.0 + rng.below(1024) / 1024.0) * std::ldexp(1.0, -1 - rng.below(25));
            scale_to(1.99);
            break;
        case 5: // magnitudes ascending …
        case 6: // … and descending
            for(double &x : w)
                x = full * std::ldexp(1.0, -(1 + rng.below(24)));
            std::sort(w.begin(), w.end());
            if(kind == 6)
                std::reverse(w.begin(), w.end());
            scale_to(1.99);
            break;
        case 7: // alternating large / tiny
            for(int k = 0; k < K; k++)
                w[k] = (k & 1) ? full * 0x1p-24 : full / K;
            break;
        case 8: // a geometric decay, repeated
            for(int k = 0; k < K; k++)
                w[k] = full * std::ldexp(1.0, -1 - (k % 24));
            scale_to(1.99);
            break;
        default: // just below powers of two, random binades near the top
            for(double &x : w)
                x = (1.0 - 0x1p-11) * std::ldexp(1.0, -(1 + rng.below(7)));
            scale_to(1.99);
            break;
    }
    double sum16 = 0;
    std::vector<_Float16> h(K);
    for(int k = 0; k < K; k++)
        sum16 += double(h[k] = static_cast<_Float16>(w[k]));
    const double down = sum16 > 2.0 ? 0.5 : 1.0; // the fp16 rounding pushed the sum over the precondition: one binade down (exact)
    for(int k = 0; k < K; k++)
        a_bits[k] = __builtin_bit_cast(uint16_t, static_cast<_Float16>(double(h[k]) * down * 32768.0)); // exact for weights in [0, 2)
}

void probe_pixel_column(ProbeRng &rng, const int K, const int kind, uint16_t *b_bits, const int stride)
{
    for(int k = 0; k < K; k++)
    {
        int v;
        switch(kind)
        {
            case 0: v = 255; break;
            case 1: v = rng.below(256); break;
            case 2: v = k == 0 ? 255 : (rng.below(256) | 1); break;
            case 3: v = (k & 1) ? 1 : 255; break;
            case 4: v = 1 << rng.below(8); break;
            default: v = rng.uniform() < 0.1 ? 255 : rng.below(4); break;
        }
        b_bits[(size_t)k * stride] = uint16_t(v); // the byte as an fp16 subnormal's mantissa
    }
}

// Runs the measurement on the context's device (once per device), returns its record.
int run_band_probe(lfi_ctx *c, const BandProbe **out)
{
    const int dev = c->device;
    if(dev < 0 || dev >= LFI_PROBE_MAX_DEVICES)
        return fail(c, LFI_EINVAL, "device index beyond the band probe's table");
    std::lock_guard<std::mutex> lock(g_band_probe_mutex);
    BandProbe &p = g_band_probe[dev];
    *out = &p;
    if(p.done)
        return LFI_OK;
    const auto t0 = std::chrono::steady_clock::now();
    constexpr int SETS = 6, KS[2] = {64, 256}; // per chain depth and MFMA shape: 2 × 2 × 6 sets of 1,024 sums
    constexpr size_t SET_WORDS = 2 * 32 * 256;  // A [32][K] + B [K][32], K ≤ 256
    const int n_sets = 2 * 2 * SETS;
    std::vector<uint16_t> host((size_t)n_sets * SET_WORDS, 0);
    std::vector<float> got((size_t)n_sets * 1024);
    ProbeRng rng{0x9E3779B97F4A7C15ull ^ uint64_t(dev + 1)};
    for(int s = 0; s < n_sets; s++)
    {
        const int K = KS[(s / SETS) & 1];
        uint16_t *a = host.data() + (size_t)s * SET_WORDS, *b = a + 32 * 256;
        for(int i = 0; i < 32; i++)
            probe_weight_row(rng, K, (s % SETS == 0) ? i % 10 : rng.below(10), a + (size_t)i * K);
        for(int j = 0; j < 32; j++)
            probe_pixel_column(rng, K, (s % SETS == 0) ? j % 6 : rng.below(6), b + j, 32);
    }
    uint8_t *d = nullptr;
    const size_t in_bytes = host.size() * sizeof(uint16_t), out_bytes = got.size() * sizeof(float);
    LFI_HIP(c, hipMalloc(reinterpret_cast<void **>(&d), in_bytes + out_bytes));
    hipError_t e = hipMemcpyAsync(d, host.data(), in_bytes, hipMemcpyHostToDevice, c->stream);
    for(int s = 0; s < n_sets && e == hipSuccess; s++)
    {
        const int K = KS[(s / SETS) & 1], shape = s / (2 * SETS);
        const uint16_t *da = reinterpret_cast<const uint16_t *>(d) + (size_t)s * SET_WORDS, *db = da + 32 * 256;
        float *dc = reinterpret_cast<float *>(d + in_bytes) + (size_t)s * 1024;
        if(shape == 0)
            hipLaunchKernelGGL(lfi::probe_mfma_f16_chain<0>, dim3(1), dim3(64), 0, c->stream, da, db, K, dc);
        else
            hipLaunchKernelGGL(lfi::probe_mfma_f16_chain<1>, dim3(1), dim3(64), 0, c->stream, da, db, K, dc);
        e = hipGetLastError();
    }
    if(e == hipSuccess)
        e = hipMemcpyAsync(got.data(), d + in_bytes, out_bytes, hipMemcpyDeviceToHost, c->stream);
    if(e == hipSuccess)
        e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    LFI_HIP(c, e);
    // exact sums: every fp16 is an integer multiple of 2^-24; A ≤ 65504 (< 2^41 units), B ≤ 255 units, ≤ 256 terms: < 2^57 in units of 2^-48
    double worst = 0.0;
    bool representable = true;
    for(int s = 0; s < n_sets; s++)
    {
        const int K = KS[(s / SETS) & 1];
        const uint16_t *a = host.data() + (size_t)s * SET_WORDS, *b = a + 32 * 256;
        const double bound = K * 0x1p-26; // the kernels' N·2^-17 on S is N·2^-26 on acc = S·2^-9
        for(int i = 0; i < 32; i++)
            for(int j = 0; j < 32; j++)
            {
                int64_t exact = 0;
                for(int k = 0; k < K; k++)
                    exact += (int64_t)std::llround(double(__builtin_bit_cast(_Float16, a[(size_t)i * K + k])) * 0x1p24) * (int64_t)b[(size_t)k * 32 + j];
                const double g = double(got[(size_t)s * 1024 + i * 32 + j]) * 0x1p48;
                const int64_t gi = std::llround(g);
                representable = representable && double(gi) == g;
                worst = std::max(worst, std::fabs(double(gi - exact)) * 0x1p-48 / bound);
            }
    }
    p.worst = float(worst);
    p.ok = representable && worst <= 1.0;
    p.sums = n_sets * 1024;
    p.ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    p.done = true;
    return LFI_OK;
}

// Must this STD launch over more than 64 images take the analytic band although the caller did not ask for it?
int std_band_forced_analytic(lfi_ctx *c, bool *forced)
{
    *forced = false;
    if(c->flags & LFI_FLAG_STD_ANALYTIC_BAND)
        return LFI_OK;
    const BandProbe *p = nullptr;
    if(int rc = run_band_probe(c, &p))
        return rc;
    *forced = !p->ok || (c->flags & LFI_FLAG_STD_BAND_PROBE_FAIL);
    return LFI_OK;
}

} // namespace
