// On-device input pipeline: the train-time transform chain of the reference's runners
//   RandomHorizontalFlip -> RandomCrop(S, padding) -> ToTensor -> x + 0.001 * randn -> Normalize   (`main/train_cifar.py:31-38`)
//   RandomCrop(224) -> RandomHorizontalFlip -> ToTensor -> x + 0.001 * randn -> Normalize           (`main/train_clip_imagenet.py:27-36`)
// and the per-batch device Normalize of `training/ad_trainer.py:413-425`, as ONE HBM-bound gather kernel over a uint8 NHWC
// image set that stays resident in HBM (CIFAR-10's 50 000 training images are 150 MB of the 288 GB): per step the host
// sends only (image index, crop origin, flip) per sample; the PIL work in DataLoader workers (`ad_trainer.py:103,385`)
// disappears.  ColorJitter(0.01) is not reproduced (SURVEY.md N1 lists crop / flip / noise / normalize).
// The noise is counter-based: element e of batch slot b draws from splitmix64(seed * 2^40 + b * 2^18 + e) by Box-Muller,
// so a step is reproducible from (seed, crop parameters) and restatable on the CPU (oracle/augment.py).
#include "common.h"

namespace {

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    unsigned long long z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// one thread per output pixel (img, y, x): 3 source bytes -> 3 floats in the 3 NCHW planes (coalesced along x)
__global__ __launch_bounds__(256) void augment_kernel(const uint8_t* __restrict__ src, const int32_t* __restrict__ params,
                                                      const float* __restrict__ mean, const float* __restrict__ stdv,
                                                      float* __restrict__ out, int n, int Hs, int Ws, int Ho, int Wo,
                                                      int flip_first, float noise_std, unsigned long long seed) {
    const size_t total = (size_t)n * Ho * Wo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho), b = (int)(i / ((size_t)Wo * Ho));
        const int idx = params[b * 4 + 0], top = params[b * 4 + 1], left = params[b * 4 + 2], flip = params[b * 4 + 3];
        const int sy = top + y;
        int sx;
        if (flip_first) sx = flip ? Ws - 1 - (left + x) : left + x;          // flip the source, then crop
        else sx = left + (flip ? Wo - 1 - x : x);                            // crop, then flip the crop
        float v[3] = {0.f, 0.f, 0.f};                                        // RandomCrop pads with 0
        if (sy >= 0 && sy < Hs && sx >= 0 && sx < Ws) {
            const uint8_t* p = src + (((size_t)idx * Hs + sy) * Ws + sx) * 3;
            v[0] = (float)p[0]; v[1] = (float)p[1]; v[2] = (float)p[2];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float a = v[c] / 255.0f;                                         // ToTensor
            if (noise_std > 0.f) {
                const unsigned long long e = ((unsigned long long)c * Ho + y) * Wo + x;
                const unsigned long long z = splitmix64((seed << 40) + ((unsigned long long)b << 18) + e);
                const float u1 = (float)((z >> 40) + 1ull) * (1.0f / 16777216.0f);          // (0, 1]
                const float u2 = (float)((z >> 16) & 0xFFFFFFull) * (1.0f / 16777216.0f);   // [0, 1)
                a += noise_std * sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
            }
            if (mean) a = (a - mean[c]) / stdv[c];
            out[(((size_t)b * 3 + c) * Ho + y) * Wo + x] = a;
        }
    }
}

}  // namespace

extern "C" int eoe_augment_batch(const uint8_t* src, int64_t n_src, int Hs, int Ws, const int32_t* params, const float* mean,
                                 const float* stdv, float* out, int n, int Ho, int Wo, int flip_first, float noise_std,
                                 uint64_t seed, void* stream) {
    EOE_CHECK_ARG(src && params && out && n_src > 0 && n > 0 && Hs > 0 && Ws > 0 && Ho > 0 && Wo > 0, "augment_batch: bad args");
    EOE_CHECK_ARG((mean == nullptr) == (stdv == nullptr), "augment_batch: mean/std must both be given or both NULL");
    EOE_CHECK_ARG(n < (1 << 22) && (size_t)3 * Ho * Wo < (1u << 18) && seed < (1ull << 24) && noise_std >= 0.f,
                  "augment_batch: n < 2^22, 3*Ho*Wo < 2^18, seed < 2^24 (the counter layout of the noise generator)");
    ProfScope ps("augment_batch", 0, 3.0 * n * Ho * Wo + 12.0 * n * Ho * Wo, stream);
    size_t g = ((size_t)n * Ho * Wo + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(augment_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, src, params, mean, stdv, out, n, Hs, Ws,
                       Ho, Wo, flip_first, noise_std, (unsigned long long)seed);
    EOE_CHECK_LAUNCH("augment_batch");
    return 0;
}
