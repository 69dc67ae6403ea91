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

int g_parity_flags = 0;      // option "parity_flags": bit 0 = the fp32 VALU kernels instead of the fp32 MFMA ones (A/B, tests)

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

// ------------------------------------------------------------------------------------------------ fp32 on the matrix cores (round 3)
// The same three implicit GEMMs on v_mfma_f32_16x16x4_f32: fp32 operands in, fp32 accumulate, every product exact in fp32 and summed
// in a fixed k order (gfx950 has no tf32 / xf32 path: this instruction runs at the fp32 vector rate, 64 flop/clk/SIMD = 157 TF peak --
// 16x below the 16-bit MFMA but ~20x above the scalar-gather VALU kernel above, which spent its time in per-element index arithmetic).
// 128 x BN x 16 tiles (BN = 128 or 64), four waves of 64 x BN/2, operands staged through a double-buffered k-major LDS image
// (As[k][m], Bs[k][n]: a fragment read is 16 consecutive floats of one k row), ONE barrier per k-tile, the next k-tile's global loads
// in flight under the current MFMAs.  Activations are fetched as float4 along their contiguous axis (NHWC channels: the k axis of
// forward / dgrad, the m / n axis of wgrad) with one pixel decode per thread per kernel (forward / dgrad: a thread stages the same two
// rows for every k-tile) or per k-tile (wgrad, multiply-shift divisions); the weights (small, L2-resident, OIHW) by scalar loads.
//   requirements: forward C % 16 == 0 and an NHWC activation; dgrad cout % 16 == 0; wgrad C % 4 == 0 and cout % 4 == 0, NHWC.
//   Anything else (the 3-channel NCHW stem with its fused Normalize) keeps the kernel above.
// The N-side fragment is the MFMA's A operand, the M-side its B operand: lane (lr = lane & 15, lg = lane >> 4) then holds output row
// m = lr and the 4 CONSECUTIVE columns n = 4*lg .. 4*lg + 3 of each 16 x 16 tile -- 16-byte stores along the contiguous axis.
// S2 (dgrad of a stride-2 convolution over an even map): an input pixel (iy, ix) only meets the taps with ky = iy + pad, kx = ix + pad
// (mod 2); gathering all kh x kw taps for every pixel runs 4x the useful MFMAs on zeros (measured: 15-20 TF against 66-76 TF for the
// stride-1 layers).  Instead the four parity classes (iy & 1, ix & 1) are four GEMMs over a quarter of the pixels each, with the class's
// own tap list (3x3, pad 1: 1 / 2 / 2 / 4 taps; 1x1: one class has the tap, three write zeros) -- blockIdx.z = class * nslab + slab.  The
// valid taps come in the same (ky, kx, channel) order as before: bitwise the same sums.
// Split k (forward / dgrad with few output tiles and a long reduction: the small maps of the 32 x 32 WideResNet, the FC layers): nslab > 1
// slabs of the reduction write partial outputs [slab][rows][N] (`out` = the workspace), summed in slab order by slab_sum_out_kernel.
// 16 bytes at p[idx] if `ok`, else zeros -- as an UNCONDITIONAL load (from a 16-byte block of zeros when !ok; p may then be anything), so that
// a batch of them is issued back to back instead of one load -> wait -> use group per `if`
__device__ __forceinline__ f32x4 load4_if(const float* __restrict__ p, size_t idx, bool ok) {
    alignas(16) static __device__ const float zero4[4] = {0.f, 0.f, 0.f, 0.f};
    return *(const f32x4*)(ok ? p + idx : zero4);
}
template <int MODE, int BN, int WV = 0, bool S2 = false, bool WK = false>      // WK: forward / dgrad weights from the k-major packed copy `wk` (eoe_conv_f32_pack_weights); WV (wgrad variants): 0 = both sides float4, 1 = the x side element-wise (OIHW order), 2 = roles exchanged, 3 = both
__global__ __launch_bounds__(256, (WK && BN == 128) ? 4 : 1) void conv_f32_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ dy, const float* __restrict__ bias,
                                                            float* __restrict__ out, PGeo g, int M, int N, int K, int k_per_slab,
                                                            int accumulate, FDiv dHoWo, FDiv dWo, FDiv dHW, FDiv dW, FDiv dC,
                                                            const float* __restrict__ mean = nullptr, const float* __restrict__ stdv = nullptr,
                                                            FDiv dtaps = FDiv(1), FDiv dkw = FDiv(1), int nslab = 1,
                                                            const float* __restrict__ wk = nullptr, int wk_rows = 0) {
    constexpr bool b_oihw = (WV == 1 || WV == 3), swap = (WV == 2 || WV == 3);
    // BK: 16-deep k-tiles; the narrow weight-gradient tiles (BN = 64: half the MFMAs per k-tile for the same barrier and staging overhead)
    // take 32 -- their reduction runs over millions of pixels anyway.  AV = float4 per thread and operand and k-tile
    constexpr int BM = 128, BK = (MODE == P_WGRAD && BN == 64) ? 32 : 16, LDA = BM + 4, LDB = BN + 4, NI = BN / 32, NBV = BN / 16;
    constexpr int AV = BM * BK / 4 / 256;
    __shared__ __attribute__((aligned(16))) float As[2][BK][LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lr = lane & 15, lg = lane >> 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    int slab = (int)blockIdx.z;
    int s2_py = 0, s2_px = 0, s2_ky0 = 0, s2_kx0 = 0, s2_ntx = 1;
    if (S2) {
        const int cls = (int)blockIdx.z / nslab;
        slab = (int)blockIdx.z - cls * nslab;
        s2_py = cls >> 1; s2_px = cls & 1;
        s2_ky0 = (s2_py + g.pad) & 1; s2_kx0 = (s2_px + g.pad) & 1;
        const int nty = g.kh > s2_ky0 ? (g.kh - s2_ky0 + 1) >> 1 : 0;
        s2_ntx = g.kw > s2_kx0 ? (g.kw - s2_kx0 + 1) >> 1 : 0;
        K = nty * s2_ntx * g.cout;                         // this class's reduction length
        k_per_slab = ((K + nslab - 1) / nslab + 15) / 16 * 16;
    }
    const int k_lo = slab * k_per_slab, k_hi = min(K, k_lo + k_per_slab);
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * (BN / 2);
    const int taps = g.kh * g.kw;
    f32x4 acc[4][NI];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- per-thread staging coordinates
    // forward / dgrad: A = 128 rows x 16 k as 2 float4 per thread (row ar[i], k quad akq[i]); the row's pixel is decoded once
    int a_img[AV], a_y[AV], a_x[AV], a_r[AV], a_kq[AV];
    bool a_ok[AV];
#pragma unroll
    for (int i = 0; i < AV; ++i) {
        const int e = t + 256 * i;
        if (MODE == P_WGRAD) { a_r[i] = e >> 5; a_kq[i] = e & 31; a_ok[i] = true; a_img[i] = a_y[i] = a_x[i] = 0; continue; }     // (kk, m quad)
        a_r[i] = e >> 2; a_kq[i] = e & 3;
        const int m = m0 + a_r[i];
        a_ok[i] = m < M;
        unsigned img = 0, rem = 0, yy = 0, xx = 0;
        if (a_ok[i]) {
            if (MODE == P_FWD) { dHoWo.divmod((unsigned)m, img, rem); dWo.divmod(rem, yy, xx); }
            else { dHW.divmod((unsigned)m, img, rem); dW.divmod(rem, yy, xx); }
        }
        a_img[i] = (int)img;
        if (S2) { yy = 2 * yy + s2_py; xx = 2 * xx + s2_px; }       // (dHW, dW divide by the class's quarter map)
        a_y[i] = MODE == P_FWD ? (int)yy * g.stride - g.pad : (int)yy + g.pad;
        a_x[i] = MODE == P_FWD ? (int)xx * g.stride - g.pad : (int)xx + g.pad;
    }
    // wgrad: the x-side column quad of this thread (the same for every k-tile): packed (channel << 16 | ky << 8 | kx) per element
    // (b_oihw: four separate (ci, ky, kx); else one entry, four consecutive channels of one tap); -1 = past the last column
    bool wg_ok[AV];
    int wg_dec[AV][4];
#pragma unroll
    for (int i = 0; i < AV; ++i) wg_ok[i] = false;
    if (MODE == P_WGRAD) {
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int q4 = a_kq[i] * 4;
            const int nn = swap ? m0 + q4 : n0 + q4, lim = swap ? M : N;
            wg_ok[i] = nn < lim && (swap || q4 < BN);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) wg_dec[i][jj] = -1;
            if (!wg_ok[i]) continue;
            if (b_oihw) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    if (nn + jj >= lim) break;
                    unsigned ci, tap, ky, kx;
                    dtaps.divmod((unsigned)(nn + jj), ci, tap); dkw.divmod(tap, ky, kx);
                    wg_dec[i][jj] = (int)((ci << 16) | (ky << 8) | kx);
                }
            } else {
                unsigned tap, ci;
                dC.divmod((unsigned)nn, tap, ci);
                const int ky = (int)tap / g.kw, kx = (int)tap - ky * g.kw;
                wg_dec[i][0] = (int)((ci << 16) | ((unsigned)ky << 8) | (unsigned)kx);
            }
        }
    }
    f32x4 ra[AV];
    float rb[NBV];
    f32x4 rbv[AV];
    f32x4 rbq[NBV / 4];                               // WK: the weight tile as BN / 4 float4 along n per k row, NBV / 4 per thread
    auto load_tile = [&](int k0) {
        if (MODE == P_WGRAD) {
            // k = output pixel.  A(m = co, k) = dy[k][co]: float4 along m.  B(n' = tap * C + ci, k) = x at the pixel's tap window: float4
            // along ci.  `swap` (cout <= 64: the 128-row side would be half empty): the roles exchanged, C^T[n'][co] is computed
            // the column quad a thread gathers is the same for every k-tile: its (tap, channel) decode is hoisted (wg_*), only the pixel moves
            auto gather_x = [&](int k, int i) -> f32x4 {
                f32x4 vb = {0.f, 0.f, 0.f, 0.f};
                if (!wg_ok[i]) return vb;
                unsigned img, rem, oy, ox;
                dHoWo.divmod((unsigned)k, img, rem); dWo.divmod(rem, oy, ox);
                const int by = (int)oy * g.stride - g.pad, bx = (int)ox * g.stride - g.pad;
                if (b_oihw) {
                    // a layer whose input the float4 fetch cannot serve (the NCHW image with its fused Normalize, C % 4 != 0): element by
                    // element, n = (ci, ky, kx) in the weight's own order; the dy side stays float4
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int d = wg_dec[i][jj];
                        if (d < 0) break;
                        vb[jj] = load_x(x, mean, stdv, g, (int)img, by + ((d >> 8) & 0xff), bx + (d & 0xff), d >> 16);
                    }
                    return vb;
                }
                const int d = wg_dec[i][0];
                const int iy = by + ((d >> 8) & 0xff), ix = bx + (d & 0xff);
                if ((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W)
                    vb = *(const f32x4*)(x + (((size_t)img * g.H + iy) * g.W + ix) * g.C + (d >> 16));
                return vb;
            };
#pragma unroll
            for (int i = 0; i < AV; ++i) {
                const int kk = a_r[i], q4 = a_kq[i] * 4, k = k0 + kk;
                f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
                if (k < k_hi) {                             // (measured: the unconditional form of these -- load4_if -- is 5 % slower here)
                    if (!swap) {
                        if (m0 + q4 < M) va = *(const f32x4*)(dy + (size_t)k * g.cout + m0 + q4);
                        if (q4 < BN) vb = gather_x(k, i);
                    } else {
                        va = gather_x(k, i);
                        if (q4 < BN && n0 + q4 < N) vb = *(const f32x4*)(dy + (size_t)k * g.cout + n0 + q4);
                    }
                }
                ra[i] = va; rbv[i] = vb;
            }
            return;
        }
        const int tap = k0 / (MODE == P_FWD ? g.C : g.cout), c0 = k0 - tap * (MODE == P_FWD ? g.C : g.cout);
        int ky = tap / g.kw, kx = tap - ky * g.kw;
        if (S2) { const int ty = tap / s2_ntx; ky = s2_ky0 + 2 * ty; kx = s2_kx0 + 2 * (tap - ty * s2_ntx); }
        // forward over a 4-channel map (the 3-channel image packed to NHWC4, eoe_pack_image_nhwc4): a 16-deep k-tile is 4 taps x 4 channels,
        // the thread's float4 is the whole pixel of tap (k0 / 4 + kq)
        const bool c4 = MODE == P_FWD && g.C == 4;
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            // (every fetch unconditional -- load4_if: a pixel outside the map reads the tensor's first 16 bytes and is zeroed.  Behind `if (inside)`
            //  hipcc waits for each load where it is issued, and the k-tile's loads no longer fly under the MFMAs)
            f32x4 v;
            if (MODE == P_FWD && c4) {
                const int tp = (k0 >> 2) + a_kq[i];
                const int ty = tp / g.kw, iy = a_y[i] + ty, ix = a_x[i] + (tp - ty * g.kw);
                v = load4_if(x, (((size_t)a_img[i] * g.H + iy) * g.W + ix) * 4,
                             a_ok[i] && tp < taps && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W);
            } else if (MODE == P_FWD) {
                const int iy = a_y[i] + ky, ix = a_x[i] + kx;
                v = load4_if(x, (((size_t)a_img[i] * g.H + iy) * g.W + ix) * g.C + c0 + a_kq[i] * 4,
                             a_ok[i] && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W);
            } else {
                const int ny = a_y[i] - ky, nx = a_x[i] - kx;                       // (iy + pad - ky, ix + pad - kx)
                int oy = ny, ox = nx;
                bool okd = a_ok[i] && ny >= 0 && nx >= 0;
                if (g.stride != 1) { oy = ny / g.stride; ox = nx / g.stride; okd = okd && (oy * g.stride == ny) && (ox * g.stride == nx); }
                v = load4_if(dy, (((size_t)a_img[i] * g.Ho + oy) * g.Wo + ox) * g.cout + c0 + a_kq[i] * 4, okd && oy < g.Ho && ox < g.Wo);
            }
            ra[i] = v;
        }
        if (WK) {
            // k-major packed weights wk[k row][N] (forward: row = tap * C + c; dgrad: row = tap * cout + co, tap = ky * kw + kx of the FULL
            // kernel): a thread fetches 4 consecutive n of one k row -- one 16-byte load and one 16-byte LDS store instead of four scalar
            // loads gathered with a stride of kh * kw floats and four scalar stores
#pragma unroll
            for (int i = 0; i < NBV / 4; ++i) {
                const int e = t + 256 * i, kk = e / (BN / 4), nq = (e - kk * (BN / 4)) * 4, nn = n0 + nq;
                const int krow = (MODE == P_DGRAD && S2) ? (ky * g.kw + kx) * g.cout + c0 + kk : k0 + kk;
                rbq[i] = load4_if(wk, (size_t)krow * N + nn, nn < N && krow < wk_rows);
            }
            return;
        }
        // weights: BN rows x 16 k scalars, NBV per thread; element (n, kk): forward w[n][c0+kk][ky][kx], dgrad w[c0+kk][n][ky][kx]
#pragma unroll
        for (int i = 0; i < NBV; ++i) {
            const int e = t + 256 * i, kk = e & 15, r = e >> 4, nn = n0 + r;
            size_t wi;
            bool wok = nn < N;
            if (MODE == P_FWD && c4) {
                const int tp = (k0 >> 2) + (kk >> 2), ty = tp / g.kw;
                wok = wok && tp < taps;
                wi = (((size_t)nn * 4 + (kk & 3)) * g.kh + ty) * g.kw + (tp - ty * g.kw);
            } else {
                wi = MODE == P_FWD ? (((size_t)nn * g.C + c0 + kk) * g.kh + ky) * g.kw + kx
                                   : (((size_t)(c0 + kk) * g.C + nn) * g.kh + ky) * g.kw + kx;
            }
            const float b = w[wok ? wi : (size_t)0];
            rb[i] = wok ? b : 0.f;
        }
    };
    auto store_tile = [&](int buf) {
        if (MODE == P_WGRAD) {
#pragma unroll
            for (int i = 0; i < AV; ++i) {
                *(f32x4*)&As[buf][a_r[i]][a_kq[i] * 4] = ra[i];
                if (a_kq[i] * 4 < BN) *(f32x4*)&Bs[buf][a_r[i]][a_kq[i] * 4] = rbv[i];
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < AV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) As[buf][a_kq[i] * 4 + j][a_r[i]] = ra[i][j];
        if (WK) {
#pragma unroll
            for (int i = 0; i < NBV / 4; ++i) {
                const int e = t + 256 * i, kk = e / (BN / 4), nq = (e - kk * (BN / 4)) * 4;
                *(f32x4*)&Bs[buf][kk][nq] = rbq[i];
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NBV; ++i) {
            const int e = t + 256 * i;
            Bs[buf][e & 15][e >> 4] = rb[i];
        }
    };

    const int nkt = (k_hi - k_lo + BK - 1) / BK;
    if (nkt > 0) {
        load_tile(k_lo);
        store_tile(0);
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nkt) load_tile(k_lo + (kt + 1) * BK);
#pragma unroll
            for (int ks = 0; ks < BK / 4; ++ks) {
                float am[4], bn[NI];
#pragma unroll
                for (int i = 0; i < 4; ++i) am[i] = As[buf][ks * 4 + lg][wm0 + i * 16 + lr];
#pragma unroll
                for (int j = 0; j < NI; ++j) bn[j] = Bs[buf][ks * 4 + lg][wn0 + j * 16 + lr];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bn[j], am[i], acc[i][j], 0, 0, 0);
            }
            if (kt + 1 < nkt) store_tile(buf ^ 1);
            __syncthreads();
        }
    }
    // ---- epilogue: lane holds row m = lr, columns 4*lg .. 4*lg+3 of each tile
    // split k (nslab > 1): slab partials [slab][rows][N]; bias / accumulate are applied by the slab sum.  S2: the class's row -> its pixel
    const size_t rows_all = S2 ? (size_t)g.n * g.H * g.W : (size_t)M;
    float* dst = out + (MODE == P_WGRAD ? (size_t)blockIdx.z * M * N : (nslab > 1 ? (size_t)slab * rows_all * N : 0));
    const bool vec = (N & 3) == 0, plain = nslab > 1;
    // the bias once per column group, ahead of the stores (a load between two stores waits for the first store to drain: the tile's 4 x NI
    // stores went out one at a time)
    f32x4 bv[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int nn = n0 + wn0 + j * 16 + lg * 4;
        bv[j] = load4_if(bias, (size_t)nn, MODE == P_FWD && bias != nullptr && !plain && vec && nn < N);
    }
    const bool acc_dst = MODE == P_DGRAD && accumulate && !plain;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm0 + i * 16 + lr;
        if (m >= M) continue;
        size_t row = (size_t)m;
        if (S2) {
            unsigned img, rem, yy, xx;
            dHW.divmod((unsigned)m, img, rem); dW.divmod(rem, yy, xx);
            row = ((size_t)img * g.H + 2 * yy + s2_py) * g.W + 2 * xx + s2_px;
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int nn = n0 + wn0 + j * 16 + lg * 4;
            if (nn >= N) continue;
            f32x4 v = acc[i][j];
            float* d = dst + row * N + nn;
            if (vec) {
                v += bv[j];
                if (acc_dst) v += *(const f32x4*)d;
                *(f32x4*)d = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (nn + r >= N) break;
                    float s1 = v[r];
                    if (MODE == P_FWD && bias && !plain) s1 += bias[nn + r];
                    d[r] = (MODE == P_DGRAD && accumulate && !plain) ? d[r] + s1 : s1;
                }
            }
        }
    }
}

// The same tiles for the shapes the float4 fetches cannot serve -- the 3-channel NCHW image of the first layer with its fused
// Normalize (ad_trainer.py:413-425), channel counts that are not multiples of 4 / 16: operands fetched element by element, but with the
// index arithmetic hoisted -- a thread stages the same k column (forward: one (tap, channel) decode per k-tile; wgrad: one pixel decode)
// and the same eight rows (decoded once per kernel) for the whole launch.  Forward: k = tap * C + c as above; wgrad: n = (ci, ky, kx) in
// the weight's own OIHW order (no remap afterwards), k = output pixel.
template <int MODE, int BN>
__global__ __launch_bounds__(256) void conv_f32_mfma_scalar_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                                   const float* __restrict__ stdv, const float* __restrict__ w,
                                                                   const float* __restrict__ dy, const float* __restrict__ bias,
                                                                   float* __restrict__ out, PGeo g, int M, int N, int K, int k_per_slab,
                                                                   FDiv dHoWo, FDiv dWo, FDiv dC, FDiv dkw, FDiv dtaps) {
    constexpr int BM = 128, BK = 16, LDA = BM + 4, LDB = BN + 4, NI = BN / 32, NBV = BN / 16;
    __shared__ float As[2][BK][LDA];
    __shared__ float Bs[2][BK][LDB];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lr = lane & 15, lg = lane >> 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int k_lo = blockIdx.z * k_per_slab, k_hi = min(K, k_lo + k_per_slab);
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * (BN / 2);
    const int kk = t & 15, r0 = t >> 4;                     // this thread's k column of every k-tile; its rows are r0 + 16 i
    f32x4 acc[4][NI];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // forward: rows = output pixels (decoded once); wgrad: B rows = weight elements j = (ci, ky, kx) (decoded once)
    int a_img[8], a_y[8], a_x[8];
    int b_ci[NBV], b_ky[NBV], b_kx[NBV];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a_img[i] = -1; a_y[i] = a_x[i] = 0;
        if (MODE == P_FWD) {
            const int m = m0 + r0 + 16 * i;
            if (m < M) {
                unsigned img, rem, oy, ox;
                dHoWo.divmod((unsigned)m, img, rem); dWo.divmod(rem, oy, ox);
                a_img[i] = (int)img; a_y[i] = (int)oy * g.stride - g.pad; a_x[i] = (int)ox * g.stride - g.pad;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NBV; ++i) {
        b_ci[i] = -1; b_ky[i] = b_kx[i] = 0;
        if (MODE == P_WGRAD) {
            const int j = n0 + r0 + 16 * i;
            if (j < N) {
                unsigned ci, tap, ky, kx;
                dtaps.divmod((unsigned)j, ci, tap); dkw.divmod(tap, ky, kx);
                b_ci[i] = (int)ci; b_ky[i] = (int)ky; b_kx[i] = (int)kx;
            }
        }
    }
    float ra[8], rb[NBV];
    auto load_tile = [&](int k0) {
        const int k = k0 + kk;
        const bool kok = k < k_hi;
        if (MODE == P_FWD) {
            unsigned tap = 0, c = 0, ky = 0, kx = 0;
            if (kok) { dC.divmod((unsigned)k, tap, c); dkw.divmod(tap, ky, kx); }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                ra[i] = (kok && a_img[i] >= 0) ? load_x(x, mean, stdv, g, a_img[i], a_y[i] + (int)ky, a_x[i] + (int)kx, (int)c) : 0.f;
#pragma unroll
            for (int i = 0; i < NBV; ++i) {
                const int nn = n0 + r0 + 16 * i;
                rb[i] = (kok && nn < N) ? w[(((size_t)nn * g.C + c) * g.kh + ky) * g.kw + kx] : 0.f;
            }
        } else {
            unsigned img = 0, rem = 0, oy = 0, ox = 0;
            if (kok) { dHoWo.divmod((unsigned)k, img, rem); dWo.divmod(rem, oy, ox); }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int co = m0 + r0 + 16 * i;
                ra[i] = (kok && co < M) ? dy[(size_t)k * g.cout + co] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < NBV; ++i)
                rb[i] = (kok && b_ci[i] >= 0) ? load_x(x, mean, stdv, g, (int)img, (int)oy * g.stride - g.pad + b_ky[i],
                                                       (int)ox * g.stride - g.pad + b_kx[i], b_ci[i]) : 0.f;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) As[buf][kk][r0 + 16 * i] = ra[i];
#pragma unroll
        for (int i = 0; i < NBV; ++i) Bs[buf][kk][r0 + 16 * i] = rb[i];
    };
    const int nkt = (k_hi - k_lo + BK - 1) / BK;
    if (nkt > 0) {
        load_tile(k_lo);
        store_tile(0);
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nkt) load_tile(k_lo + (kt + 1) * BK);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                float am[4], bn[NI];
#pragma unroll
                for (int i = 0; i < 4; ++i) am[i] = As[buf][ks * 4 + lg][wm0 + i * 16 + lr];
#pragma unroll
                for (int j = 0; j < NI; ++j) bn[j] = Bs[buf][ks * 4 + lg][wn0 + j * 16 + lr];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bn[j], am[i], acc[i][j], 0, 0, 0);
            }
            if (kt + 1 < nkt) store_tile(buf ^ 1);
            __syncthreads();
        }
    }
    float* dst = out + (MODE == P_WGRAD ? (size_t)blockIdx.z * M * N : 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm0 + i * 16 + lr;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nn = n0 + wn0 + j * 16 + lg * 4 + r;
                if (nn >= N) continue;
                float v = acc[i][j][r];
                if (MODE == P_FWD && bias) v += bias[nn];
                dst[(size_t)m * N + nn] = v;
            }
        }
    }
}

// The slab sums of the weight gradients.  Up to 1024 slabs per element: one thread walking them serially is a chain of a thousand dependent
// L2 round trips (the 3 -> 32 first layer of CNN32: 0.3 of its 0.38 ms), so an element gets ZL = 16 lanes -- lane j adds slabs j, j + 16, ...
// in ascending order (8 loads in flight), and the 16 lane sums are added in lane order: a fixed order, in double (the split costs no accuracy
// against one long fp32 sum).  src(e) maps an output element to its place inside a slab:
//   LAYOUT 0: the same place;  1: out [co][ci][tap] <- slab [co][tap * C + ci];  2: out [co][ci][tap] <- slab [tap * C + ci][co];
//   3: out [co][n] <- slab [n][co]
template <int LAYOUT>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, size_t count, int S, int cout,
                                                          int C, int taps) {
    constexpr int ZL = 16;
    __shared__ double part[ZL][16];
    const int el = threadIdx.x & 15, j = threadIdx.x >> 4;
    const size_t e = (size_t)blockIdx.x * 16 + el;
    double s = 0.0;
    if (e < count) {
        size_t src = e;
        if (LAYOUT == 1 || LAYOUT == 2) {
            const int co = (int)(e / ((size_t)C * taps)), r = (int)(e - (size_t)co * C * taps), ci = r / taps, tap = r - ci * taps;
            src = LAYOUT == 2 ? ((size_t)tap * C + ci) * cout + co : (size_t)co * C * taps + (size_t)tap * C + ci;
        } else if (LAYOUT == 3) {
            const size_t N = count / (size_t)cout;
            const int co = (int)(e / N);
            src = (e - (size_t)co * N) * cout + co;
        }
        const float* p = slabs + src;
        int z = j;
#pragma unroll 1
        for (; z + 7 * ZL < S; z += 8 * ZL) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(z + u * ZL) * count];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)v[u];
        }
        for (; z < S; z += ZL) s += (double)p[(size_t)z * count];
    }
    part[j][el] = s;
    __syncthreads();
    if (j == 0 && e < count) {
        double t = 0.0;
#pragma unroll
        for (int u = 0; u < ZL; ++u) t += part[u][el];
        out[e] = (float)t;
    }
}
#define EOE_SLAB_REDUCE(LAYOUT, slabs, out, count, S, cout, C, taps, stream)                                                          \
    hipLaunchKernelGGL((slab_reduce_kernel<LAYOUT>), dim3((unsigned)(((count) + 15) / 16)), dim3(256), 0, (hipStream_t)(stream), slabs, out, \
                       (size_t)(count), S, cout, C, taps)

// the 3-channel NCHW image, normalised (ad_trainer.py:413-425: (x - mean) / std per channel), as an fp32 NHWC4 map (4th channel 0): the
// first layer's forward and weight gradient then fetch whole pixels as float4 instead of gathering and normalising element by element
// (each image element is used kh * kw times by either: the stem ran at 31 / 15 TF against 70-95 TF for the other layers)
__global__ __launch_bounds__(256) void pack_image_nhwc4_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                               const float* __restrict__ stdv, float* __restrict__ out, size_t pixels, size_t HW) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= pixels) return;
    const size_t img = e / HW, p = e - img * HW;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float t = x[(img * 3 + c) * HW + p];
        if (mean) t = (t - mean[c]) / stdv[c];
        v[c] = t;
    }
    *(f32x4*)(out + e * 4) = v;
}

// forward / dgrad split k: out[e] = (accumulate ? out[e] : 0) + bias[e % N] + sum over slabs in slab order
__global__ __launch_bounds__(256) void slab_sum_out_kernel(const float* __restrict__ slabs, float* __restrict__ out, const float* __restrict__ bias,
                                                           size_t count4, int N, int S, int accumulate) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count4) return;
    double s[4] = {0.0, 0.0, 0.0, 0.0};              // in double: the split costs no accuracy against one long fp32 sum
    for (int z = 0; z < S; ++z) {
        const f32x4 v = *(const f32x4*)(slabs + ((size_t)z * count4 + e) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] += (double)v[r];
    }
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (float)s[r];
    if (bias) o += *(const f32x4*)(bias + (e * 4) % (size_t)N);
    if (accumulate) o += *(const f32x4*)(out + e * 4);
    *(f32x4*)(out + e * 4) = o;
}

// how many k slabs a forward / dgrad launch of `tiles` output tiles (x `classes`) over a reduction of K should use: none when the launch
// fills the chip or the reduction is short; else about two workgroups per CU, >= 128 k per slab, within the workspace
int fwd_slabs(int tiles, int K, size_t out_elems, size_t ws_bytes, const void* ws) {
    if (!ws || (((uintptr_t)ws) & 15) || (out_elems & 3) || tiles >= 128 || K < 512) return 1;
    int S = (512 + tiles - 1) / tiles;
    if (S > K / 128) S = K / 128;
    const size_t fit = ws_bytes / (out_elems * sizeof(float));
    if ((size_t)S > fit) S = (int)fit;
    return S < 2 ? 1 : S;
}

#define MFMA_DIVS(g)                                                                                                          \
    const FDiv dHoWo((unsigned)((g).Ho * (g).Wo)), dWo((unsigned)(g).Wo), dHW((unsigned)((g).H * (g).W)), dW((unsigned)(g).W), \
        dC((unsigned)(g).C)

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

// k-major fp32 copies of a convolution weight [cout][cin][kh][kw]: wf[(tap * cin + ci)][cout] (forward) and wd[(tap * cout + co)][cin]
// (dgrad), tap = ky * kw + kx -- what the forward / dgrad kernels stage as 16-byte rows (conv_f32_mfma_kernel, WK)
__global__ __launch_bounds__(256) void conv_f32_pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wd,
                                                                    int cout, int cin, int taps) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x, count = (size_t)cout * cin * taps;
    if (e >= count) return;
    // e enumerates the OUTPUT of the forward copy: (tap, ci, co) with co fastest (coalesced stores; the gather side is small and cached)
    const int co = (int)(e % cout), ci = (int)((e / cout) % cin), tap = (int)(e / ((size_t)cout * cin));
    const float v = w[((size_t)co * cin + ci) * taps + tap];
    if (wf) wf[e] = v;
    if (wd) wd[((size_t)tap * cout + co) * cin + ci] = v;
}

extern "C" int eoe_conv_f32_pack_weights(const float* w, float* wf, float* wd, int cout, int cin, int kh, int kw, void* stream) {
    EOE_CHECK_ARG(w && (wf || wd) && cout > 0 && cin > 0 && kh > 0 && kw > 0, "conv_f32_pack_weights: bad arguments");
    const size_t count = (size_t)cout * cin * kh * kw;
    hipLaunchKernelGGL(conv_f32_pack_weights_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, wf, wd, cout, cin,
                       kh * kw);
    EOE_CHECK_LAUNCH("conv_f32_pack_weights");
    return 0;
}

extern "C" int eoe_pack_image_nhwc4(const float* x_nchw, const float* mean, const float* stdv, float* out_nhwc4, int n, int H, int W, void* stream) {
    EOE_CHECK_ARG(x_nchw && out_nhwc4 && n > 0 && H > 0 && W > 0, "pack_image_nhwc4: bad arguments");
    EOE_CHECK_ARG((mean == nullptr) == (stdv == nullptr), "pack_image_nhwc4: mean/std must both be given or both NULL");
    const size_t pixels = (size_t)n * H * W;
    ProfScope ps("pack_image_nhwc4", 0, 28.0 * pixels, stream);
    hipLaunchKernelGGL(pack_image_nhwc4_kernel, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x_nchw, mean, stdv,
                       out_nhwc4, pixels, (size_t)H * W);
    EOE_CHECK_LAUNCH("pack_image_nhwc4");
    return 0;
}

extern "C" int eoe_conv_f32_fwd(const float* x, int x_nchw, const float* mean, const float* stdv, const float* w, const float* bias,
                                float* y, const eoe_conv_geometry* geo, int cout, void* workspace, size_t workspace_bytes, const float* w_kmajor,
                                void* stream) {
    EOE_CHECK_ARG(x && w && y, "conv_f32_fwd: null pointer");
    EOE_CHECK_ARG((mean == nullptr) == (stdv == nullptr), "conv_f32_fwd: mean/std must both be given or both NULL");
    PGeo g;
    EOE_TRY(fill_geo("conv_f32_fwd", geo, cout, x_nchw, g));
    const int M = g.n * g.Ho * g.Wo, N = cout, K = g.kh * g.kw * g.C;
    ProfScope ps("conv_f32_fwd", 2.0 * M * N * K, 4.0 * ((double)g.n * g.H * g.W * g.C + (double)N * K + (double)M * N), stream);
    if (!x_nchw && !mean && ((g.C % 16) == 0 || g.C == 4) && !(g_parity_flags & 1)) {   // fp32 MFMA (parity_flags bit 0: the VALU kernel, A/B)
        MFMA_DIVS(g);
        const int gx = N <= 64 ? (N + 63) / 64 : (N + 127) / 128, gy = (M + 127) / 128;
        const int S = (N & 3) ? 1 : fwd_slabs(gx * gy, K, (size_t)M * N, workspace_bytes, workspace);
        const int per = S > 1 ? ((K + S - 1) / S + 15) / 16 * 16 : K;
        float* dst = S > 1 ? (float*)workspace : y;
        const bool wkp = w_kmajor != nullptr && (N & 3) == 0 && !(g_parity_flags & 8);       // parity_flags bit 3: the scalar weight gather (A/B)
#define EOE_FWD_LAUNCH(BNV, WKV)                                                                                                        \
        hipLaunchKernelGGL((conv_f32_mfma_kernel<P_FWD, BNV, 0, false, WKV>), dim3(gx, gy, S), dim3(256), 0, (hipStream_t)stream,       \
                           x, w, (const float*)nullptr, bias, dst, g, M, N, K, per, 0, dHoWo, dWo, dHW, dW, dC,                          \
                           (const float*)nullptr, (const float*)nullptr, FDiv(1), FDiv(1), S, w_kmajor, K)
        if (N <= 64) { if (wkp) EOE_FWD_LAUNCH(64, true); else EOE_FWD_LAUNCH(64, false); }
        else { if (wkp) EOE_FWD_LAUNCH(128, true); else EOE_FWD_LAUNCH(128, false); }
#undef EOE_FWD_LAUNCH
        EOE_CHECK_LAUNCH("conv_f32_fwd (mfma)");
        if (S > 1) {
            const size_t count4 = (size_t)M * N / 4;
            hipLaunchKernelGGL(slab_sum_out_kernel, dim3((unsigned)((count4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                               (const float*)workspace, y, bias, count4, N, S, 0);
            EOE_CHECK_LAUNCH("conv_f32_fwd_sum");
        }
        return 0;
    }
    if (!(g_parity_flags & 1)) {                    // fp32 MFMA, element-wise fetches (NCHW image + Normalize, odd channel counts)
        const FDiv sHoWo((unsigned)(g.Ho * g.Wo)), sWo((unsigned)g.Wo), sC((unsigned)g.C), skw((unsigned)g.kw), staps((unsigned)(g.kh * g.kw));
        if (N <= 64) hipLaunchKernelGGL((conv_f32_mfma_scalar_kernel<P_FWD, 64>), dim3((N + 63) / 64, (M + 127) / 128, 1), dim3(256), 0,
                                        (hipStream_t)stream, x, mean, stdv, w, (const float*)nullptr, bias, y, g, M, N, K, K, sHoWo, sWo, sC, skw, staps);
        else hipLaunchKernelGGL((conv_f32_mfma_scalar_kernel<P_FWD, 128>), dim3((N + 127) / 128, (M + 127) / 128, 1), dim3(256), 0,
                                (hipStream_t)stream, x, mean, stdv, w, (const float*)nullptr, bias, y, g, M, N, K, K, sHoWo, sWo, sC, skw, staps);
        EOE_CHECK_LAUNCH("conv_f32_fwd (mfma, scalar)");
        return 0;
    }
    hipLaunchKernelGGL((conv_f32_kernel<P_FWD>), dim3((N + 63) / 64, (M + 63) / 64, 1), dim3(256), 0, (hipStream_t)stream, x, mean, stdv,
                       w, (const float*)nullptr, bias, y, g, M, N, K, K, 0);
    EOE_CHECK_LAUNCH("conv_f32_fwd");
    return 0;
}

extern "C" int eoe_conv_f32_dgrad(const float* dy, const float* w, float* dx, const eoe_conv_geometry* geo, int cout, int accumulate,
                                  void* workspace, size_t workspace_bytes, const float* w_kmajor, void* stream) {
    EOE_CHECK_ARG(dy && w && dx, "conv_f32_dgrad: null pointer");
    PGeo g;
    EOE_TRY(fill_geo("conv_f32_dgrad", geo, cout, 0, g));
    const int M = g.n * g.H * g.W, N = g.C, K = g.kh * g.kw * cout;
    ProfScope ps("conv_f32_dgrad", 2.0 * M * N * K, 4.0 * ((double)g.n * g.Ho * g.Wo * cout + (double)N * K + (double)M * N), stream);
    if ((cout % 16) == 0 && !(g_parity_flags & 1)) {
        MFMA_DIVS(g);
        const int gx = N <= 64 ? (N + 63) / 64 : (N + 127) / 128;
        if (g.stride == 2 && (g.H % 2) == 0 && (g.W % 2) == 0 && !(g_parity_flags & 2)) {
            // the four parity classes of the input pixels, each with its own tap list (conv_f32_mfma_kernel, S2); parity_flags bit 1: off
            const int Mc = g.n * (g.H / 2) * (g.W / 2), gy = (Mc + 127) / 128;
            const FDiv cHW((unsigned)((g.H / 2) * (g.W / 2))), cW((unsigned)(g.W / 2));
            const int kmax = ((g.kh + 1) / 2) * ((g.kw + 1) / 2) * cout;            // the longest class
            const int S = (N & 3) ? 1 : fwd_slabs(gx * gy * 4, kmax, (size_t)M * N, workspace_bytes, workspace);
            float* dst = S > 1 ? (float*)workspace : dx;
            const bool wkp = w_kmajor != nullptr && (N & 3) == 0 && !(g_parity_flags & 8);
#define EOE_DG2_LAUNCH(BNV, WKV)                                                                                                        \
            hipLaunchKernelGGL((conv_f32_mfma_kernel<P_DGRAD, BNV, 0, true, WKV>), dim3(gx, gy, 4 * S), dim3(256), 0, (hipStream_t)stream,  \
                               (const float*)nullptr, w, dy, (const float*)nullptr, dst, g, Mc, N, K, K, accumulate, dHoWo, dWo, cHW, cW, dC, \
                               (const float*)nullptr, (const float*)nullptr, FDiv(1), FDiv(1), S, w_kmajor, K)
            if (N <= 64) { if (wkp) EOE_DG2_LAUNCH(64, true); else EOE_DG2_LAUNCH(64, false); }
            else { if (wkp) EOE_DG2_LAUNCH(128, true); else EOE_DG2_LAUNCH(128, false); }
#undef EOE_DG2_LAUNCH
            EOE_CHECK_LAUNCH("conv_f32_dgrad (mfma, stride-2 classes)");
            if (S > 1) {
                const size_t count4 = (size_t)M * N / 4;
                hipLaunchKernelGGL(slab_sum_out_kernel, dim3((unsigned)((count4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                                   (const float*)workspace, dx, (const float*)nullptr, count4, N, S, accumulate);
                EOE_CHECK_LAUNCH("conv_f32_dgrad_sum");
            }
            return 0;
        }
        const int gy = (M + 127) / 128;
        const int S = (N & 3) ? 1 : fwd_slabs(gx * gy, K, (size_t)M * N, workspace_bytes, workspace);
        const int per = S > 1 ? ((K + S - 1) / S + 15) / 16 * 16 : K;
        float* dst = S > 1 ? (float*)workspace : dx;
        const bool wkp = w_kmajor != nullptr && (N & 3) == 0 && !(g_parity_flags & 8);
#define EOE_DG_LAUNCH(BNV, WKV)                                                                                                         \
        hipLaunchKernelGGL((conv_f32_mfma_kernel<P_DGRAD, BNV, 0, false, WKV>), dim3(gx, gy, S), dim3(256), 0, (hipStream_t)stream,     \
                           (const float*)nullptr, w, dy, (const float*)nullptr, dst, g, M, N, K, per, accumulate, dHoWo, dWo, dHW, dW, dC, \
                           (const float*)nullptr, (const float*)nullptr, FDiv(1), FDiv(1), S, w_kmajor, K)
        if (N <= 64) { if (wkp) EOE_DG_LAUNCH(64, true); else EOE_DG_LAUNCH(64, false); }
        else { if (wkp) EOE_DG_LAUNCH(128, true); else EOE_DG_LAUNCH(128, false); }
#undef EOE_DG_LAUNCH
        EOE_CHECK_LAUNCH("conv_f32_dgrad (mfma)");
        if (S > 1) {
            const size_t count4 = (size_t)M * N / 4;
            hipLaunchKernelGGL(slab_sum_out_kernel, dim3((unsigned)((count4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                               (const float*)workspace, dx, (const float*)nullptr, count4, N, S, accumulate);
            EOE_CHECK_LAUNCH("conv_f32_dgrad_sum");
        }
        return 0;
    }
    hipLaunchKernelGGL((conv_f32_kernel<P_DGRAD>), dim3((N + 63) / 64, (M + 63) / 64, 1), dim3(256), 0, (hipStream_t)stream,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, w, dy, (const float*)nullptr, dx, g, M, N, K, K,
                       accumulate);
    EOE_CHECK_LAUNCH("conv_f32_dgrad");
    return 0;
}

extern "C" size_t eoe_conv_f32_wgrad_workspace(const eoe_conv_geometry* geo, int cout) {
    if (!geo) return 0;
    const size_t count = (size_t)cout * geo->C * geo->kh * geo->kw;
    // at most 64 slabs; small weights (the first layers: few output tiles, millions of pixels to reduce over) up to 1024 within 16 MB
    size_t slabs = 64;
    if (count * sizeof(float) * slabs < (16u << 20)) slabs = (16u << 20) / (count * sizeof(float));
    if (slabs > 1024) slabs = 1024;
    return count * sizeof(float) * slabs;
}

extern "C" int eoe_conv_f32_wgrad(const float* x, int x_nchw, const float* mean, const float* stdv, const float* dy, float* dw,
                                  const eoe_conv_geometry* geo, int cout, void* workspace, size_t workspace_bytes, void* stream) {
    EOE_CHECK_ARG(x && dy && dw && workspace, "conv_f32_wgrad: null pointer");
    EOE_CHECK_ARG((mean == nullptr) == (stdv == nullptr), "conv_f32_wgrad: mean/std must both be given or both NULL");
    PGeo g;
    EOE_TRY(fill_geo("conv_f32_wgrad", geo, cout, x_nchw, g));
    const int M = cout, N = g.C * g.kh * g.kw, K = g.n * g.Ho * g.Wo;
    if (!x_nchw && !mean && (g.C % 4) == 0 && (cout % 4) == 0 && !(g_parity_flags & 1)) {
        // fp32 MFMA: partial sums [S][cout][tap * C + ci] over S pixel slabs, summed in slab order into the OIHW gradient.  cout <= 64:
        // the roles of the two sides exchanged ([S][tap * C + ci][cout]: the 128-row side of the tile is the long one)
        const int swap = (M <= 64 && N > 64) ? 1 : 0;
        const int Mk = swap ? N : M, Nk = swap ? M : N;
        const bool narrow = Nk <= 64;
        const int tiles = ((Mk + 127) / 128) * (narrow ? (Nk + 63) / 64 : (Nk + 127) / 128);
        const size_t count = (size_t)M * N;
        // about four workgroups per CU (25-34 KB of LDS each: several are co-resident, and 5 tiles x 64 slabs = 320 workgroups left
        // a quarter of the CUs with twice the work of the rest), at least 8 k-tiles per slab, within the workspace
        int S = (1024 + tiles - 1) / tiles;
        const size_t s_fit = workspace_bytes / (count * sizeof(float));
        if ((size_t)S > s_fit) S = (int)s_fit;
        if (S > K / 128) S = K / 128;
        if (S > 1024) S = 1024;
        if (S < 1) S = 1;
        int per = ((K + S - 1) / S + 15) / 16 * 16;
        S = (K + per - 1) / per;
        EOE_CHECK_ARG(workspace_bytes >= count * sizeof(float) * S, "conv_f32_wgrad: workspace of %zu bytes, need %zu", workspace_bytes,
                      count * sizeof(float) * S);
        ProfScope ps("conv_f32_wgrad", 2.0 * M * N * K, 4.0 * ((double)g.n * g.H * g.W * g.C + (double)K * M + (double)M * N), stream);
        MFMA_DIVS(g);
#define EOE_WG_LAUNCH(BNV, WVV)                                                                                                          \
        hipLaunchKernelGGL((conv_f32_mfma_kernel<P_WGRAD, BNV, WVV>), dim3((Nk + BNV - 1) / BNV, (Mk + 127) / 128, S), dim3(256), 0,       \
                           (hipStream_t)stream, x, (const float*)nullptr, dy, (const float*)nullptr, (float*)workspace, g, Mk, Nk, K, per, 0, \
                           dHoWo, dWo, dHW, dW, dC)
        if (swap) { if (narrow) EOE_WG_LAUNCH(64, 2); else EOE_WG_LAUNCH(128, 2); }
        else { if (narrow) EOE_WG_LAUNCH(64, 0); else EOE_WG_LAUNCH(128, 0); }
#undef EOE_WG_LAUNCH
        EOE_CHECK_LAUNCH("conv_f32_wgrad (mfma)");
        if (swap) EOE_SLAB_REDUCE(2, (const float*)workspace, dw, count, S, cout, g.C, g.kh * g.kw, stream);
        else EOE_SLAB_REDUCE(1, (const float*)workspace, dw, count, S, cout, g.C, g.kh * g.kw, stream);
        EOE_CHECK_LAUNCH("conv_f32_wgrad_sum");
        return 0;
    }
    if (!(g_parity_flags & 1) && (cout % 4) == 0) {
        // dy side float4, x side element by element in OIHW order (the stem: NCHW image + Normalize, 3 channels)
        const bool narrow = N <= 64, swap1 = M <= 64 && N > 64;
        const int tiles = swap1 ? ((N + 127) / 128) * ((M + 63) / 64) : ((M + 127) / 128) * (narrow ? (N + 63) / 64 : (N + 127) / 128);
        const size_t count = (size_t)M * N;
        const int s_max = (int)(workspace_bytes / (count * sizeof(float)));
        int S = (2048 + tiles - 1) / tiles;
        if (S > s_max) S = s_max;
        if (S > 1024) S = 1024;
        if (S < 1) S = 1;
        int per = ((K + S - 1) / S + 15) / 16 * 16;
        S = (K + per - 1) / per;
        EOE_CHECK_ARG(workspace_bytes >= count * sizeof(float) * S, "conv_f32_wgrad: workspace of %zu bytes, need %zu", workspace_bytes,
                      count * sizeof(float) * S);
        ProfScope ps("conv_f32_wgrad", 2.0 * M * N * K, 4.0 * ((double)g.n * g.H * g.W * g.C + (double)K * M + (double)M * N), stream);
        MFMA_DIVS(g);
        const FDiv staps((unsigned)(g.kh * g.kw)), skw((unsigned)g.kw);
        if (swap1) {
            // cout <= 64 (the stem): the roles exchanged -- the (ci, ky, kx) columns fill the 128-row side, partials [S][N][cout]
            hipLaunchKernelGGL((conv_f32_mfma_kernel<P_WGRAD, 64, 3>), dim3((M + 63) / 64, (N + 127) / 128, S), dim3(256), 0, (hipStream_t)stream,
                               x, (const float*)nullptr, dy, (const float*)nullptr, (float*)workspace, g, N, M, K, per, 0, dHoWo, dWo, dHW, dW, dC,
                               mean, stdv, staps, skw);
        } else if (narrow) hipLaunchKernelGGL((conv_f32_mfma_kernel<P_WGRAD, 64, 1>), dim3((N + 63) / 64, (M + 127) / 128, S), dim3(256), 0, (hipStream_t)stream,
                                       x, (const float*)nullptr, dy, (const float*)nullptr, (float*)workspace, g, M, N, K, per, 0, dHoWo, dWo, dHW, dW, dC,
                                       mean, stdv, staps, skw);
        else hipLaunchKernelGGL((conv_f32_mfma_kernel<P_WGRAD, 128, 1>), dim3((N + 127) / 128, (M + 127) / 128, S), dim3(256), 0, (hipStream_t)stream,
                                x, (const float*)nullptr, dy, (const float*)nullptr, (float*)workspace, g, M, N, K, per, 0, dHoWo, dWo, dHW, dW, dC,
                                mean, stdv, staps, skw);
        EOE_CHECK_LAUNCH("conv_f32_wgrad (mfma, dy float4)");
        if (swap1) EOE_SLAB_REDUCE(3, (const float*)workspace, dw, count, S, M, 1, 1, stream);
        else EOE_SLAB_REDUCE(0, (const float*)workspace, dw, count, S, M, 1, 1, stream);
        EOE_CHECK_LAUNCH("conv_f32_wgrad_sum");
        return 0;
    }
    if (!(g_parity_flags & 1)) {
        const bool narrow = N <= 64;
        const int tiles = ((M + 127) / 128) * (narrow ? (N + 63) / 64 : (N + 127) / 128);
        int S = (1024 + tiles - 1) / tiles;
        if (S > 64) S = 64;
        int per = ((K + S - 1) / S + 15) / 16 * 16;
        S = (K + per - 1) / per;
        const size_t count = (size_t)M * N;
        EOE_CHECK_ARG(workspace_bytes >= count * sizeof(float) * S, "conv_f32_wgrad: workspace of %zu bytes, need %zu", workspace_bytes,
                      count * sizeof(float) * S);
        ProfScope ps("conv_f32_wgrad", 2.0 * M * N * K, 4.0 * ((double)g.n * g.H * g.W * g.C + (double)K * M + (double)M * N), stream);
        const FDiv sHoWo((unsigned)(g.Ho * g.Wo)), sWo((unsigned)g.Wo), sC((unsigned)g.C), skw((unsigned)g.kw), staps((unsigned)(g.kh * g.kw));
        if (narrow) hipLaunchKernelGGL((conv_f32_mfma_scalar_kernel<P_WGRAD, 64>), dim3((N + 63) / 64, (M + 127) / 128, S), dim3(256), 0,
                                       (hipStream_t)stream, x, mean, stdv, (const float*)nullptr, dy, (const float*)nullptr, (float*)workspace, g,
                                       M, N, K, per, sHoWo, sWo, sC, skw, staps);
        else hipLaunchKernelGGL((conv_f32_mfma_scalar_kernel<P_WGRAD, 128>), dim3((N + 127) / 128, (M + 127) / 128, S), dim3(256), 0,
                                (hipStream_t)stream, x, mean, stdv, (const float*)nullptr, dy, (const float*)nullptr, (float*)workspace, g,
                                M, N, K, per, sHoWo, sWo, sC, skw, staps);
        EOE_CHECK_LAUNCH("conv_f32_wgrad (mfma, scalar)");
        EOE_SLAB_REDUCE(0, (const float*)workspace, dw, count, S, M, 1, 1, stream);
        EOE_CHECK_LAUNCH("conv_f32_wgrad_sum");
        return 0;
    }
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
    EOE_SLAB_REDUCE(0, (const float*)workspace, dw, count, S, M, 1, 1, stream);
    EOE_CHECK_LAUNCH("conv_f32_wgrad_sum");
    return 0;
}
