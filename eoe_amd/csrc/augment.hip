// On-device input pipeline: the train-time transform chain of the reference's runners
//   RandomHorizontalFlip -> RandomCrop(S, padding) -> ToTensor -> x + 0.001 * randn -> Normalize   (`main/train_cifar.py:31-38`)
//   RandomCrop(224) -> RandomHorizontalFlip -> ToTensor -> x + 0.001 * randn -> Normalize           (`main/train_clip_imagenet.py:27-36`)
// and the per-batch device Normalize of `training/ad_trainer.py:413-425`, as ONE HBM-bound gather kernel over a uint8 NHWC
// image set that stays resident in HBM (CIFAR-10's 50 000 training images are 150 MB of the 288 GB): per step the host
// sends only (image index, crop origin, flip) per sample; the PIL work in DataLoader workers (`ad_trainer.py:103,385`)
// disappears.  Resize (bilinear / bicubic with Pillow's antialiasing) and ColorJitter are the kernels at the end of this file.
// The noise is counter-based: element e of batch slot b draws from splitmix64(seed * 2^40 + b * 2^18 + e) by Box-Muller,
// so a step is reproducible from (seed, crop parameters) and restatable on the CPU (oracle/augment.py).
#include "common.h"
#include <math.h>
#include <stdlib.h>

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


// ------------------------------------------------------------------------------------------------------------------------
// Resize and ColorJitter (main/train_imagenet.py:30-31, main/train_clip_imagenet.py:28-29, main/train_cifar.py:32; CLIP's own
// preprocessing, clip_official/clip/clip.py:58-65).  In the reference these run on PIL images inside DataLoader workers:
// torchvision hands them to Pillow, so the arithmetic restated here is Pillow's 8-bit integer one and the results are equal to
// Pillow's byte for byte (tests/golden g14):
//   Image.resize            separable, horizontal then vertical on uint8, filter support stretched by the down-scaling factor
//                           (antialias), weights normalised in double and rounded to 22-bit fixed point (libImaging/Resample.c)
//   ImageEnhance.*          Image.blend(degenerate, image, factor): black / rounded mean gray / gray image (libImaging/Blend.c)
//   hue                     8-bit RGB -> HSV -> h + uint8(255 * factor) -> RGB (libImaging/Convert.c)
// ------------------------------------------------------------------------------------------------------------------------
namespace {

constexpr int RESIZE_PRECISION_BITS = 32 - 8 - 2;

double filter_bilinear(double x) { if (x < 0.0) x = -x; return x < 1.0 ? 1.0 - x : 0.0; }
double filter_bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// one pass along `axis_len`: src [outer, axis_in, inner] -> dst [outer, axis_out, inner], uint8; thread per output byte
__global__ __launch_bounds__(256) void resize_pass_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                          const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk,
                                                          int ksize, size_t outer, int axis_in, int axis_out, int inner) {
    const size_t total = outer * axis_out * inner;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int in = (int)(i % inner);
        const int xx = (int)((i / inner) % axis_out);
        const size_t o = i / ((size_t)inner * axis_out);
        const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
        const uint8_t* p = src + (o * axis_in + xmin) * inner + in;
        const int32_t* k = kk + (size_t)xx * ksize;
        int ss = 1 << (RESIZE_PRECISION_BITS - 1);
        for (int x = 0; x < cnt; ++x) ss += (int)p[(size_t)x * inner] * k[x];
        ss >>= RESIZE_PRECISION_BITS;
        dst[i] = (uint8_t)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
    }
}

// Pillow's C code is compiled without fused multiply-adds; hipcc contracts a * b + c by default -- also through __fmul_rn /
// __fadd_rn, which are plain operators in HIP's headers -- and a blend that lands within one ulp of an integer then truncates
// to the neighbouring byte (saturation 0.99 on a gray level of 100: 100 - 99.00000095 = 0.99999905 -> 0 instead of 1)
#pragma clang fp contract(off)
// single IEEE operations compiled under the pragma above (HIP's __fmul_rn / __fadd_rn are header inlines that carry the default
// `contract` flag and still fuse)
__device__ __forceinline__ float add_(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_(float a, float b) { return a - b; }
__device__ __forceinline__ float mul_(float a, float b) { return a * b; }
__device__ __forceinline__ float div_(float a, float b) { return a / b; }

__device__ __forceinline__ int gray_l(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

// Blend.c: in1 + alpha * (in2 - in1) as separate float multiply and add (no fused multiply-add), truncated
__device__ __forceinline__ int blend_u8(int deg, int v, float alpha) {
    const float t = add_((float)deg, mul_(alpha, (float)(v - deg)));
    if (alpha >= 0.f && alpha <= 1.f) return (int)t & 255;
    return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}

__device__ __forceinline__ void rgb2hsv_u8(int r, int g, int b, int& uh, int& us, int& uv) {
    const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
    uv = maxc;
    if (minc == maxc) { uh = 0; us = 0; return; }
    const float cr = (float)(maxc - minc);
    const float s = div_(cr, (float)maxc);
    const float rc = div_((float)(maxc - r), cr), gc = div_((float)(maxc - g), cr), bc = div_((float)(maxc - b), cr);
    float h;
    if (r == maxc) h = sub_(bc, gc);
    else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
    else h = (float)(4.0 + (double)gc - (double)rc);
    h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
    int ih = (int)((double)h * 255.0), is = (int)((double)s * 255.0);
    uh = ih < 0 ? 0 : (ih > 255 ? 255 : ih);
    us = is < 0 ? 0 : (is > 255 ? 255 : is);
}

__device__ __forceinline__ void hsv2rgb_u8(int h, int s, int v, int& r, int& g, int& b) {
    if (s == 0) { r = g = b = v; return; }
    const float hf = div_(mul_((float)h, 6.0f), 255.0f);
    const int i = (int)floorf(hf);
    const float f = sub_(hf, (float)i);
    const float fs = div_((float)s, 255.0f);
    const float vf = (float)v;
    const int p = (int)floor((double)mul_(vf, sub_(1.0f, fs)) + 0.5);
    const int q = (int)floor((double)mul_(vf, sub_(1.0f, mul_(fs, f))) + 0.5);
    const int t = (int)floor((double)mul_(vf, sub_(1.0f, mul_(fs, sub_(1.0f, f)))) + 0.5);
    switch (i % 6) {
        case 0: r = v; g = t; b = p; break;
        case 1: r = q; g = v; b = p; break;
        case 2: r = p; g = v; b = t; break;
        case 3: r = p; g = q; b = v; break;
        case 4: r = t; g = p; b = v; break;
        default: r = v; g = p; b = q; break;
    }
    r = min(max(r, 0), 255); g = min(max(g, 0), 255); b = min(max(b, 0), 255);
}

// the ops of one image, in its order, applied to one pixel; `upto` = stop in front of the op with this code (4 = apply all)
__device__ __forceinline__ void jitter_pixel(int& r, int& g, int& b, const float* f, const int* order, int gray_mean, int upto) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int op = order[j];
        if (op == upto) return;
        if (op == 0) { r = blend_u8(0, r, f[0]); g = blend_u8(0, g, f[0]); b = blend_u8(0, b, f[0]); }
        else if (op == 1) { r = blend_u8(gray_mean, r, f[1]); g = blend_u8(gray_mean, g, f[1]); b = blend_u8(gray_mean, b, f[1]); }
        else if (op == 2) { const int l = gray_l(r, g, b); r = blend_u8(l, r, f[2]); g = blend_u8(l, g, f[2]); b = blend_u8(l, b, f[2]); }
        else if (op == 3) {
            int h, s, v;
            rgb2hsv_u8(r, g, b, h, s, v);
            h = (h + ((int)(f[3] * 255.0f) & 255)) & 255;
            hsv2rgb_u8(h, s, v, r, g, b);
        }
    }
}

// MODE 0: one workgroup per batch slot sums the gray level of its image as it is just before the contrast op -> gray_mean[slot]
// MODE 1: one thread per pixel applies all ops
template <int MODE>
__global__ __launch_bounds__(256) void jitter_kernel(const uint8_t* __restrict__ src, const int32_t* __restrict__ idx,
                                                     const float* __restrict__ factors, const int32_t* __restrict__ order,
                                                     int32_t* __restrict__ gray_mean, uint8_t* __restrict__ dst, int n, int HW) {
    if (MODE == 0) {
        __shared__ unsigned long long part[256];
        const int slot = blockIdx.x;
        const float f[4] = {factors[slot * 4], factors[slot * 4 + 1], factors[slot * 4 + 2], factors[slot * 4 + 3]};
        const int ord[4] = {order[slot * 4], order[slot * 4 + 1], order[slot * 4 + 2], order[slot * 4 + 3]};
        const uint8_t* p = src + (size_t)idx[slot] * HW * 3;
        unsigned long long acc = 0;
        for (int i = threadIdx.x; i < HW; i += blockDim.x) {
            int r = p[i * 3], g = p[i * 3 + 1], b = p[i * 3 + 2];
            jitter_pixel(r, g, b, f, ord, 0, 1);                 // everything in front of the contrast op
            acc += (unsigned long long)gray_l(r, g, b);
        }
        part[threadIdx.x] = acc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
            __syncthreads();
        }
        // ImageEnhance.Contrast: int(mean + 0.5) of the L image
        if (threadIdx.x == 0) gray_mean[slot] = (int)((2 * part[0] + (unsigned long long)HW) / (2ull * HW));
    } else {
        const size_t total = (size_t)n * HW;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
            const int slot = (int)(i / HW), px = (int)(i % HW);
            const float f[4] = {factors[slot * 4], factors[slot * 4 + 1], factors[slot * 4 + 2], factors[slot * 4 + 3]};
            const int ord[4] = {order[slot * 4], order[slot * 4 + 1], order[slot * 4 + 2], order[slot * 4 + 3]};
            const uint8_t* p = src + ((size_t)idx[slot] * HW + px) * 3;
            int r = p[0], g = p[1], b = p[2];
            jitter_pixel(r, g, b, f, ord, gray_mean[slot], 4);
            uint8_t* d = dst + i * 3;
            d[0] = (uint8_t)r; d[1] = (uint8_t)g; d[2] = (uint8_t)b;
        }
    }
}

}  // namespace

extern "C" int eoe_resize_coeffs(int in_size, int out_size, int filter, int32_t* bounds, int32_t* kk, int ksize_cap, int* ksize_out) {
    EOE_CHECK_ARG(in_size > 0 && out_size > 0 && (filter == EOE_RESIZE_BILINEAR || filter == EOE_RESIZE_BICUBIC),
                  "resize_coeffs: bad arguments");
    double (*fn)(double) = filter == EOE_RESIZE_BILINEAR ? filter_bilinear : filter_bicubic;
    const double support0 = filter == EOE_RESIZE_BILINEAR ? 1.0 : 2.0;
    const double scale = (double)in_size / (double)out_size;
    double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = support0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    if (ksize_out) *ksize_out = ksize;
    if (!bounds || !kk) return 0;                       // size query
    EOE_CHECK_ARG(ksize_cap >= ksize, "resize_coeffs: kk holds %d taps per output, %d needed", ksize_cap, ksize);
    const double ss = 1.0 / filterscale;
    double* w = (double*)malloc(sizeof(double) * ksize);
    if (!w) return eoe_set_error(EOE_ERR_LAUNCH, "resize_coeffs: out of memory");
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) { w[x] = fn((x + xmin - center + 0.5) * ss); ww += w[x]; }
        for (int x = 0; x < xmax; ++x) {
            if (ww != 0.0) w[x] /= ww;
            kk[(size_t)xx * ksize_cap + x] = w[x] < 0 ? (int32_t)(-0.5 + w[x] * (1 << RESIZE_PRECISION_BITS))
                                                      : (int32_t)(0.5 + w[x] * (1 << RESIZE_PRECISION_BITS));
        }
        for (int x = xmax; x < ksize_cap; ++x) kk[(size_t)xx * ksize_cap + x] = 0;
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    free(w);
    return 0;
}

extern "C" int eoe_resize_pass_u8(const uint8_t* src, uint8_t* dst, const int32_t* bounds, const int32_t* kk, int ksize, int64_t outer,
                                  int axis_in, int axis_out, int inner, void* stream) {
    EOE_CHECK_ARG(src && dst && bounds && kk && ksize > 0 && outer > 0 && axis_in > 0 && axis_out > 0 && inner > 0, "resize_pass: bad args");
    const size_t total = (size_t)outer * axis_out * inner;
    ProfScope ps("resize_pass", 0, (double)outer * axis_in * inner + (double)total, stream);
    size_t g = (total + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(resize_pass_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, src, dst, bounds, kk, ksize, (size_t)outer,
                       axis_in, axis_out, inner);
    EOE_CHECK_LAUNCH("resize_pass");
    return 0;
}

extern "C" int eoe_color_jitter_u8(const uint8_t* src, int64_t n_src, const int32_t* idx, const float* factors, const int32_t* order,
                                   int32_t* gray_mean_scratch, uint8_t* dst, int n, int H, int W, void* stream) {
    EOE_CHECK_ARG(src && idx && factors && order && gray_mean_scratch && dst && n_src > 0 && n > 0 && H > 0 && W > 0 &&
                  (size_t)H * W < (1u << 30), "color_jitter: bad args");
    ProfScope ps("color_jitter", 0, 3.0 * 3.0 * n * H * W, stream);
    hipLaunchKernelGGL(jitter_kernel<0>, dim3(n), dim3(256), 0, (hipStream_t)stream, src, idx, factors, order, gray_mean_scratch, dst, n, H * W);
    EOE_CHECK_LAUNCH("color_jitter_mean");
    size_t g = ((size_t)n * H * W + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(jitter_kernel<1>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, src, idx, factors, order, gray_mean_scratch, dst, n,
                       H * W);
    EOE_CHECK_LAUNCH("color_jitter_apply");
    return 0;
}
