// Parity mode (SURVEY.md section 7 "Hard parts", section 8d "Parity run"): the convolutions and linear layers of the BatchNorm
// encoders (cnn.py:73-86, resnet.py:85-149) in plain fp32 -- fp32 operands straight from the fp32 activations and the fp32 master
// weights, one fp32 FMA per product in a fixed k order, no 16-bit operand rounding anywhere.  It exists to separate "the fast
// path's 16-bit operand rounding" from "an implementation difference" in the K-step trajectory tests: the fast path (16-bit MFMA
// operands, gemm.hip) deviates from the reference by what 11-bit operands cost, this path by fp32 summation order only, like any
// two fp32 BLAS libraries.  It is a correctness instrument, not a performance path: a 64x64x16 register-tiled SGEMM on the vector
// ALUs (several TFLOP/s), selected with eoe_amd.set_parity_mode(True).
//
// One kernel, three addressings of the same implicit GEMM  C[M,N] = sum_k A(m,k) * B(n,k):
//   FWD   : m = output pixel, n = cout, k = (ky,kx,ci)      A = x gathered (zero padding; optional per-channel Normalize on an
//           NCHW image, ad_trainer.py:413-425), B = w[cout,cin,kh,kw], C = y [n*Ho*Wo, cout] (+ bias)
//   DGRAD : m = input pixel,  n = cin,  k = (ky,kx,co)      A = dy gathered at ((iy+pad-ky)/s, (ix+pad-kx)/s) where divisible,
//           B = w, C = dx [n*H*W, cin] (optionally += : the residual junction's shortcut gradient)
//   WGRAD : m = cout, n = (ci,ky,kx) in the weight's own OIHW order, k = output pixel, split over gridDim.z slabs that a second
//           kernel sums in slab order (no atomics: bitwise reproducible)
#include "common.h"

namespace {

struct PGeo {
    int n, H, W, C, kh, kw, stride, pad, Ho, Wo, cout;
    int nchw;                 // x is an NCHW image (first layer) instead of an NHWC activation
};

enum { P_FWD = 0, P_DGRAD = 1, P_WGRAD = 2 };

__device__ __forceinline__ float load_x(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ stdv,
                                        const PGeo& g, int img, int iy, int ix, int c) {
    if ((unsigned)iy >= (unsigned)g.H || (unsigned)ix >= (unsigned)g.W) return 0.f;       // zero padding (after Normalize)
    const size_t off = g.nchw ? (((size_t)img * g.C + c) * g.H + iy) * g.W + ix : (((size_t)img * g.H + iy) * g.W + ix) * g.C + c;
    float v = x[off];
    if (mean) v = (v - mean[c]) / stdv[c];                                                  // transformations.py:126-138
    return v;
}

template <int MODE>
__global__ __launch_bounds__(256) void conv_f32_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ stdv, const float* __restrict__ w,
                                                       const float* __restrict__ dy, const float* __restrict__ bias,
                                                       float* __restrict__ out, PGeo g, int M, int N, int K, int k_per_slab,
                                                       int accumulate) {
    __shared__ float As[16][65];
    __shared__ float Bs[16][65];
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int k_lo = blockIdx.z * k_per_slab, k_hi = min(K, k_lo + k_per_slab);
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const int taps = g.kh * g.kw;
    for (int k0 = k_lo; k0 < k_hi; k0 += 16) {
        // stage: 256 threads x 4 elements of A and of B; consecutive threads take consecutive k (the contiguous axis of NHWC)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = t + 256 * i, kk = e & 15, r = e >> 4;
            const int k = k0 + kk;
            float a = 0.f, b = 0.f;
            if (k < k_hi) {
                if (MODE == P_FWD) {
                    const int tap = k / g.C, c = k - tap * g.C, ky = tap / g.kw, kx = tap - ky * g.kw;
                    const int m = m0 + r;
                    if (m < M) {
                        const int img = m / (g.Ho * g.Wo), rem = m - img * (g.Ho * g.Wo), oy = rem / g.Wo, ox = rem - oy * g.Wo;
                        a = load_x(x, mean, stdv, g, img, oy * g.stride - g.pad + ky, ox * g.stride - g.pad + kx, c);
                    }
                    const int co = n0 + r;
                    if (co < N) b = w[(((size_t)co * g.C + c) * g.kh + ky) * g.kw + kx];
                } else if (MODE == P_DGRAD) {
                    const int tap = k / g.cout, co = k - tap * g.cout, ky = tap / g.kw, kx = tap - ky * g.kw;
                    const int m = m0 + r;
                    if (m < M) {
                        const int img = m / (g.H * g.W), rem = m - img * (g.H * g.W), iy = rem / g.W, ix = rem - iy * g.W;
                        const int ny = iy + g.pad - ky, nx = ix + g.pad - kx;
                        if (ny >= 0 && nx >= 0 && ny % g.stride == 0 && nx % g.stride == 0) {
                            const int oy = ny / g.stride, ox = nx / g.stride;
                            if (oy < g.Ho && ox < g.Wo) a = dy[(((size_t)img * g.Ho + oy) * g.Wo + ox) * g.cout + co];
                        }
                    }
                    const int ci = n0 + r;
                    if (ci < N) b = w[(((size_t)co * g.C + ci) * g.kh + ky) * g.kw + kx];
                } else {
                    // k = output pixel; A = dy[k][co]; B = x gathered for weight element j = (ci, ky, kx)
                    const int img = k / (g.Ho * g.Wo), rem = k - img * (g.Ho * g.Wo), oy = rem / g.Wo, ox = rem - oy * g.Wo;
                    const int co = m0 + r;
                    if (co < M) a = dy[(size_t)k * g.cout + co];
                    const int j = n0 + r;
                    if (j < N) {
                        const int ci = j / taps, tap = j - ci * taps, ky = tap / g.kw, kx = tap - ky * g.kw;
                        b = load_x(x, mean, stdv, g, img, oy * g.stride - g.pad + ky, ox * g.stride - g.pad + kx, ci);
                    }
                }
            }
            As[kk][r] = a;
            Bs[kk][r] = b;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; b[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
    float* dst = out + (MODE == P_WGRAD ? (size_t)blockIdx.z * M * N : 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nn = n0 + tx * 4 + j;
            if (nn >= N) continue;
            float v = acc[i][j];
            if (MODE == P_FWD && bias) v += bias[nn];
            const size_t o = (size_t)m * N + nn;
            dst[o] = (MODE == P_DGRAD && accumulate) ? dst[o] + v : v;
        }
    }
}

// dw[e] = sum over slabs in slab order
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ slabs, float* __restrict__ out, size_t count, int S) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    float s = 0.f;
    for (int z = 0; z < S; ++z) s += slabs[(size_t)z * count + e];
    out[e] = s;
}

// x fp32 -> (hi, lo) is not needed here: operands stay fp32.

int fill_geo(const char* who, const eoe_conv_geometry* geo, int cout, int nchw, PGeo& g) {
    EOE_CHECK_ARG(geo && geo->n > 0 && geo->H > 0 && geo->W > 0 && geo->C > 0 && geo->kh > 0 && geo->kw > 0 && geo->stride > 0 &&
                  geo->pad >= 0 && cout > 0, "%s: bad geometry", who);
    const int Ho = (geo->H + 2 * geo->pad - geo->kh) / geo->stride + 1, Wo = (geo->W + 2 * geo->pad - geo->kw) / geo->stride + 1;
    EOE_CHECK_ARG(Ho == geo->Ho && Wo == geo->Wo && Ho >= 1 && Wo >= 1, "%s: Ho/Wo = %d/%d do not match the geometry (%d/%d)", who,
                  geo->Ho, geo->Wo, Ho, Wo);
    EOE_CHECK_ARG((size_t)geo->n * geo->H * geo->W < 0x7fffffffull && (size_t)geo->kh * geo->kw * (geo->C > cout ? geo->C : cout) < 0x7fffffffull,
                  "%s: too large for 32-bit indexing", who);
    g = PGeo{geo->n, geo->H, geo->W, geo->C, geo->kh, geo->kw, geo->stride, geo->pad, Ho, Wo, cout, nchw};
    return 0;
}

}  // namespace

extern "C" int eoe_conv_f32_fwd(const float* x, int x_nchw, const float* mean, const float* stdv, const float* w, const float* bias,
                                float* y, const eoe_conv_geometry* geo, int cout, void* stream) {
    EOE_CHECK_ARG(x && w && y, "conv_f32_fwd: null pointer");
    EOE_CHECK_ARG((mean == nullptr) == (stdv == nullptr), "conv_f32_fwd: mean/std must both be given or both NULL");
    PGeo g;
    EOE_TRY(fill_geo("conv_f32_fwd", geo, cout, x_nchw, g));
    const int M = g.n * g.Ho * g.Wo, N = cout, K = g.kh * g.kw * g.C;
    ProfScope ps("conv_f32_fwd", 2.0 * M * N * K, 4.0 * ((double)g.n * g.H * g.W * g.C + (double)N * K + (double)M * N), stream);
    hipLaunchKernelGGL((conv_f32_kernel<P_FWD>), dim3((N + 63) / 64, (M + 63) / 64, 1), dim3(256), 0, (hipStream_t)stream, x, mean, stdv,
                       w, (const float*)nullptr, bias, y, g, M, N, K, K, 0);
    EOE_CHECK_LAUNCH("conv_f32_fwd");
    return 0;
}

extern "C" int eoe_conv_f32_dgrad(const float* dy, const float* w, float* dx, const eoe_conv_geometry* geo, int cout, int accumulate,
                                  void* stream) {
    EOE_CHECK_ARG(dy && w && dx, "conv_f32_dgrad: null pointer");
    PGeo g;
    EOE_TRY(fill_geo("conv_f32_dgrad", geo, cout, 0, g));
    const int M = g.n * g.H * g.W, N = g.C, K = g.kh * g.kw * cout;
    ProfScope ps("conv_f32_dgrad", 2.0 * M * N * K, 4.0 * ((double)g.n * g.Ho * g.Wo * cout + (double)N * K + (double)M * N), stream);
    hipLaunchKernelGGL((conv_f32_kernel<P_DGRAD>), dim3((N + 63) / 64, (M + 63) / 64, 1), dim3(256), 0, (hipStream_t)stream,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, w, dy, (const float*)nullptr, dx, g, M, N, K, K,
                       accumulate);
    EOE_CHECK_LAUNCH("conv_f32_dgrad");
    return 0;
}

extern "C" size_t eoe_conv_f32_wgrad_workspace(const eoe_conv_geometry* geo, int cout) {
    if (!geo) return 0;
    const size_t count = (size_t)cout * geo->C * geo->kh * geo->kw;
    return count * sizeof(float) * 64;             // at most 64 slabs
}

extern "C" int eoe_conv_f32_wgrad(const float* x, int x_nchw, const float* mean, const float* stdv, const float* dy, float* dw,
                                  const eoe_conv_geometry* geo, int cout, void* workspace, size_t workspace_bytes, void* stream) {
    EOE_CHECK_ARG(x && dy && dw && workspace, "conv_f32_wgrad: null pointer");
    EOE_CHECK_ARG((mean == nullptr) == (stdv == nullptr), "conv_f32_wgrad: mean/std must both be given or both NULL");
    PGeo g;
    EOE_TRY(fill_geo("conv_f32_wgrad", geo, cout, x_nchw, g));
    const int M = cout, N = g.C * g.kh * g.kw, K = g.n * g.Ho * g.Wo;
    const int tiles = ((M + 63) / 64) * ((N + 63) / 64);
    int S = (1024 + tiles - 1) / tiles;            // about four workgroups per CU
    if (S > 64) S = 64;
    int per = ((K + S - 1) / S + 15) / 16 * 16;
    S = (K + per - 1) / per;
    const size_t count = (size_t)M * N;
    EOE_CHECK_ARG(workspace_bytes >= count * sizeof(float) * S, "conv_f32_wgrad: workspace of %zu bytes, need %zu", workspace_bytes,
                  count * sizeof(float) * S);
    ProfScope ps("conv_f32_wgrad", 2.0 * M * N * K, 4.0 * ((double)g.n * g.H * g.W * g.C + (double)K * M + (double)M * N), stream);
    hipLaunchKernelGGL((conv_f32_kernel<P_WGRAD>), dim3((N + 63) / 64, (M + 63) / 64, S), dim3(256), 0, (hipStream_t)stream, x, mean, stdv,
                       (const float*)nullptr, dy, (const float*)nullptr, (float*)workspace, g, M, N, K, per, 0);
    EOE_CHECK_LAUNCH("conv_f32_wgrad");
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, dw,
                       count, S);
    EOE_CHECK_LAUNCH("conv_f32_wgrad_sum");
    return 0;
}
