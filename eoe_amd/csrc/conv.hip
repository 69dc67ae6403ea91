// CNN backbone kernels (reference `src/eoe/models/cnn.py:44-86` CNN32 and `src/eoe/models/resnet.py:25-152` WideResNet):
// convolutions (any kernel / stride / padding) as im2col + the MFMA GEMMs of gemm.hip / gemm_tn.hip, BatchNorm (training statistics) + LeakyReLU/ReLU + 2x2 MaxPool
// fused into one apply kernel per direction.  Activations between layers are NHWC 16-bit; pre-BatchNorm conv
// outputs are fp32 [N*H*W, C] (the GEMM's fp32 epilogue) so that batch statistics are taken in fp32.
// All of these are HBM-bound (or launch-latency-bound at 32x32): coalesced 16-byte accesses along the channel dim.
#include "common.h"

namespace {

struct Geo { int kh, kw, stride, pad, Ho, Wo; };   // convolution geometry (square stride / padding)

// ---------------------------------------------------------------------------------------------- im2col
// image layer: x fp32 NCHW [n,cin,H,W] (+ optional per-channel normalise) -> patches [n*H*W, Kp] 16-bit,
// column = (ky*5+kx)*cin + c, zero padded to Kp
template <typename T>
__global__ __launch_bounds__(256) void im2col_img_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                         const float* __restrict__ stdv, T* __restrict__ out, int n,
                                                         int cin, int H, int W, int Kp, Geo g) {
    const size_t total = (size_t)n * g.Ho * g.Wo * Kp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % Kp);
        const size_t row = i / Kp;
        float v = 0.f;
        if (col < g.kh * g.kw * cin) {
            const int tap = col / cin, c = col % cin;
            const int w = (int)(row % g.Wo), h = (int)((row / g.Wo) % g.Ho), img = (int)(row / ((size_t)g.Wo * g.Ho));
            const int hh = h * g.stride + tap / g.kw - g.pad, ww = w * g.stride + tap % g.kw - g.pad;
            if (hh >= 0 && hh < H && ww >= 0 && ww < W) {
                v = x[(((size_t)img * cin + c) * H + hh) * W + ww];
                if (mean) v = (v - mean[c]) / stdv[c];
            }
        }
        out[i] = (T)v;
    }
}

// hidden layers: x NHWC [n,H,W,C] (C multiple of 8; 16-bit, or fp32 if XF32) -> patches [n*H*W, Kp], one 16-byte
// output chunk per thread
template <typename T, bool XF32>
__global__ __launch_bounds__(256) void im2col_nhwc_kernel(const void* __restrict__ xv, T* __restrict__ out, int n, int H,
                                                          int W, int C, int Kp, Geo g) {
    const int cpr = Kp / 8;                       // 16-B chunks per output row
    const int cc = C / 8;                         // chunks per tap
    const size_t total = (size_t)n * g.Ho * g.Wo * cpr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cpr);
        const size_t row = i / cpr;
        u32x4 v = {0u, 0u, 0u, 0u};
        const int tap = ch / cc;
        if (tap < g.kh * g.kw) {
            const int c8 = ch % cc;
            const int w = (int)(row % g.Wo), h = (int)((row / g.Wo) % g.Ho), img = (int)(row / ((size_t)g.Wo * g.Ho));
            const int hh = h * g.stride + tap / g.kw - g.pad, ww = w * g.stride + tap % g.kw - g.pad;
            if (hh >= 0 && hh < H && ww >= 0 && ww < W) {
                const size_t o = (((size_t)img * H + hh) * W + ww) * C + c8 * 8;
                if (XF32) {
                    const f32x4 a = *(const f32x4*)((const float*)xv + o), b = *(const f32x4*)((const float*)xv + o + 4);
                    const float t[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
                    v = pack8<T>(t);
                } else {
                    v = *(const u32x4*)((const T*)xv + o);
                }
            }
        }
        *(u32x4*)(out + row * Kp + (size_t)ch * 8) = v;
    }
}

// gradient wrt the layer input: dx fp32 NHWC [n,H,W,C] = gather over the kh*kw taps of dpatches 16-bit [n*Ho*Wo, Kp].
// S = the stride as a compile-time constant (1 or 2; 0 = run-time value): the tap loop tests divisibility by it.
template <typename T, int S>
__global__ __launch_bounds__(256) void col2im_kernel(const T* __restrict__ dp, float* __restrict__ dx, int n, int H, int W,
                                                     int C, int Kp, Geo g, QuadDecode dec, int accumulate) {
    const unsigned total = (unsigned)n * H * W * (C / 4);
    const int st = S ? S : g.stride;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        unsigned c4, pix, w, h, img;
        dec(i, c4, pix, w, h, img);
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < g.kh; ++ky) {
            // output position (ho, wo) whose tap (ky, kx) reads this pixel: ho*stride + ky - pad == h
            const int th = (int)h + g.pad - ky;
            if (th < 0 || th % st) continue;
            const int hh = th / st;
            if (hh >= g.Ho) continue;
            for (int kx = 0; kx < g.kw; ++kx) {
                const int tw = (int)w + g.pad - kx;
                if (tw < 0 || tw % st) continue;
                const int ww = tw / st;
                if (ww >= g.Wo) continue;
                float t[4];
                unpack4<T>(*(const u32x2*)(dp + (((size_t)img * g.Ho + hh) * g.Wo + ww) * Kp + (ky * g.kw + kx) * C + c4 * 4), t);
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] += t[r];
            }
        }
        f32x4 o = {a[0], a[1], a[2], a[3]};
        if (accumulate) o += *(const f32x4*)(dx + (size_t)pix * C + c4 * 4);
        *(f32x4*)(dx + (size_t)pix * C + c4 * 4) = o;
    }
}

// ---------------------------------------------------------------------------------------------- weights
// w fp32 [cout, cin, 5, 5] -> w16 [cout, Kp] (column = tap*cin + c) and w16t [Kp, cout]
// a batch of conv weights in one launch (the 19 block convolutions of the WideResNet re-pack their weights after every optimiser
// step): 64 x 64 tiles, job found through prefix sums
constexpr int PACK_MAX_JOBS = 32;
struct PackBatch {
    const float* w[PACK_MAX_JOBS];
    void* w16[PACK_MAX_JOBS];
    void* w16t[PACK_MAX_JOBS];
    void* w16d[PACK_MAX_JOBS];
    int cout[PACK_MAX_JOBS], cin[PACK_MAX_JOBS], cpad[PACK_MAX_JOBS], Kp[PACK_MAX_JOBS], taps[PACK_MAX_JOBS];
    int block_start[PACK_MAX_JOBS + 1];
    int count;
};
// A workgroup owns a 64 (output channels) x 64 (patch columns) tile: it reads w and writes w16 with the column index on the lanes,
// turns the tile through LDS, and writes the two transposed copies with the OUTPUT CHANNEL on the lanes -- both have cout as their
// contiguous dimension (with one element per thread in (o, col) order those were 2-byte writes cout elements apart: 300 us for
// the 11 M weights of the WideResNet).
template <typename T>
__global__ __launch_bounds__(256) void conv_pack_multi_kernel(PackBatch b) {
    __shared__ float tile[64][65];
    int j = 0;
    while (j + 1 < b.count && (int)blockIdx.x >= b.block_start[j + 1]) ++j;
    const float* __restrict__ w = b.w[j];
    T* __restrict__ w16 = (T*)b.w16[j];
    T* __restrict__ w16t = (T*)b.w16t[j];
    T* __restrict__ w16d = (T*)b.w16d[j];
    const int cout = b.cout[j], cin = b.cin[j], cpad = b.cpad[j], Kp = b.Kp[j], taps = b.taps[j];
    const int tiles_c = (Kp + 63) >> 6, t = blockIdx.x - b.block_start[j];
    const int o0 = (t / tiles_c) * 64, col0 = (t % tiles_c) * 64;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    {
        const int col = col0 + lx, tap = col / cpad, c = col - tap * cpad;
        const bool real = col < Kp && tap < taps && c < cin;
        float v[16];                               // all 16 loads of the thread in flight (small weights: latency-bound)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + r * 4 + ly;
            v[r] = (o < cout && real) ? w[((size_t)o * cin + c) * taps + tap] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ol = r * 4 + ly, o = o0 + ol;
            if (o < cout && col < Kp) w16[(size_t)o * Kp + col] = (T)v[r];
            tile[ol][lx] = v[r];
        }
    }
    __syncthreads();
    if (!w16t && !w16d) return;
    const int o = o0 + lx;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int cl = r * 4 + ly, col = col0 + cl;
        if (o >= cout || col >= Kp) continue;
        const T v = (T)tile[lx][cl];
        if (w16t) w16t[(size_t)col * cout + o] = v;
        const int tap = col / cpad, c = col - tap * cpad;
        // dgrad operand of a stride-1 convolution: [cin, (taps reversed) x cout] (dx = conv of dy with the flipped kernel)
        if (w16d && tap < taps && c < cin) w16d[((size_t)c * taps + (taps - 1 - tap)) * cout + o] = v;
    }
}
// g fp32 [cout, Kp] (or transposed: [taps*cin, cout]) -> dw fp32 [cout, cin, kh, kw] (+= if accumulate)
__global__ __launch_bounds__(256) void conv_unpack_kernel(const float* __restrict__ g, float* __restrict__ dw, int cout, int cin,
                                                          int cpad, int Kp, int taps, int transposed, int accumulate) {
    const int total = cout * cin * taps;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int tap = i % taps, c = (i / taps) % cin, o = i / (taps * cin);
        const float v = transposed ? g[(size_t)(tap * cpad + c) * cout + o] : g[(size_t)o * Kp + tap * cpad + c];
        dw[i] = accumulate ? dw[i] + v : v;
    }
}

// ---------------------------------------------------------------------------------------------- packed 3-channel first layer
// image fp32 NCHW [n,3,H,W] (+ per-channel normalise) -> 16-bit [n, Hp, Wp, 4] with the conv's zero padding made physical
// (pixel (h, w) lands at (h + pad, w + pad); channel 3 and the border are 0): the operand of the GATHER == 2 GEMMs
template <typename T, int CP>
__global__ __launch_bounds__(256) void stem_pack_image_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                              const float* __restrict__ stdv, T* __restrict__ out, int n, int H, int W,
                                                              int Hp, int Wp, int pad) {
    const size_t total = (size_t)n * Hp * Wp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int wp = (int)(i % Wp), hp = (int)((i / Wp) % Hp), img = (int)(i / ((size_t)Wp * Hp));
        const int h = hp - pad, w = wp - pad;
        float v[3] = {0.f, 0.f, 0.f};
        if (h >= 0 && h < H && w >= 0 && w < W) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                v[c] = x[(((size_t)img * 3 + c) * H + h) * W + w];
                if (mean) v[c] = (v[c] - mean[c]) / stdv[c];
            }
        }
        *(u32x2*)(out + i * CP) = pack4<T>(v[0], v[1], v[2], 0.f);
        if (CP == 8) *(u32x2*)(out + i * CP + 4) = (u32x2){0u, 0u};
    }
}
// w fp32 [cout,3,kh,kw] -> 16-bit [cout, K], K = ceil(kh/2)*64, column = ky*32 + kx*4 + c (zero where ky >= kh, kx >= kw, c = 3)
template <typename T>
__global__ __launch_bounds__(256) void stem_pack_weight_kernel(const float* __restrict__ w, T* __restrict__ w16, int cout, int kh, int kw, int K) {
    const int total = cout * K;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int o = i / K, col = i % K;
        const int ky = col >> 5, kx = (col >> 2) & 7, c = col & 3;
        float v = 0.f;
        if (ky < kh && kx < kw && c < 3) v = w[(((size_t)o * 3 + c) * kh + ky) * kw + kx];
        w16[i] = (T)v;
    }
}
// gT fp32 [K, cout] -> dw fp32 [cout,3,kh,kw]
__global__ __launch_bounds__(256) void stem_unpack_kernel(const float* __restrict__ g, float* __restrict__ dw, int cout, int kh, int kw) {
    const int total = cout * 3 * kh * kw;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int kx = i % kw, ky = (i / kw) % kh, c = (i / (kw * kh)) % 3, o = i / (3 * kh * kw);
        dw[i] = g[(size_t)(ky * 32 + kx * 4 + c) * cout + o];
    }
}

// ---------------------------------------------------------------------------------------------- BatchNorm
// column sums of y and y^2 over M rows, y fp32 [M, C]: workgroup (bx, by) writes its partial sums (accumulated in double)
// to row by of part[gridDim.y][2C] -- no atomics (device-scope float atomics from 8 XCDs cost ~80 us per call here,
// more than the streaming pass itself), no zero-init; bn_finalize_kernel adds the rows up.  VEC = 4: a thread owns 4 adjacent channels (16-B loads), cpb
// thread-columns x rpb row lanes per workgroup; VEC = 1 for C not a multiple of 4 (the 1-channel gate BatchNorm).
template <int VEC>
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ y, float* __restrict__ part, int M, int C, int cpb) {
    __shared__ double red[2][256][VEC];
    const int rpb = 256 / cpb;
    const int c = (blockIdx.x * cpb + threadIdx.x % cpb) * VEC;
    const int rl = threadIdx.x / cpb;
    double s[VEC], q[VEC];
#pragma unroll
    for (int r = 0; r < VEC; ++r) { s[r] = 0.0; q[r] = 0.0; }
    if (c < C) {
        int row = blockIdx.y * rpb + rl;
        const int step = gridDim.y * rpb;
        if (VEC == 4) {
            // four independent 16-B loads in flight per thread (one outstanding load per thread cannot cover HBM latency)
            for (; row + 3 * step < M; row += 4 * step) {
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *(const f32x4*)(y + (size_t)(row + u * step) * C + c);
#pragma unroll
                for (int r = 0; r < VEC; ++r) {
                    s[r] += (double)((v[0][r] + v[1][r]) + (v[2][r] + v[3][r]));
                    q[r] += (double)v[0][r] * v[0][r] + (double)v[1][r] * v[1][r] + (double)v[2][r] * v[2][r] + (double)v[3][r] * v[3][r];
                }
            }
        }
        for (; row < M; row += step) {
            if (VEC == 4) {
                const f32x4 v = *(const f32x4*)(y + (size_t)row * C + c);
#pragma unroll
                for (int r = 0; r < VEC; ++r) { s[r] += v[r]; q[r] += (double)v[r] * v[r]; }
            } else {
                const float v = y[(size_t)row * C + c];
                s[0] += v;
                q[0] += (double)v * v;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < VEC; ++r) { red[0][threadIdx.x][r] = s[r]; red[1][threadIdx.x][r] = q[r]; }
    __syncthreads();
    if (rl == 0 && c < C) {
        for (int k = 1; k < rpb; ++k)
#pragma unroll
            for (int r = 0; r < VEC; ++r) { s[r] += red[0][threadIdx.x + k * cpb][r]; q[r] += red[1][threadIdx.x + k * cpb][r]; }
        float* row = part + (size_t)blockIdx.y * 2 * C;
#pragma unroll
        for (int r = 0; r < VEC; ++r) {
            row[c + r] = (float)s[r];
            row[C + c + r] = (float)q[r];
        }
    }
}

// out[i] = sum_p part[p][i], i < n: 16 columns x 64 row lanes per workgroup, four independent loads in flight per thread
// (P <= 1024 rows -> 16 loads each; the 64-columns x 16-lanes version spent 20 us in 64 dependent L2 round trips), fixed
// summation order.  Optional BatchNorm parameter gradients straight from the sums: p0[i] (i < C), p1[i - C] (i >= C).
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ part, float* __restrict__ out, int P, int n,
                                                               float* __restrict__ p0, float* __restrict__ p1, int C, int accumulate) {
    __shared__ float l[64][17];
    const int col = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + col;
    float s = 0.f;
    if (i < n) {
        int p = lane;
        for (; p + 192 < P; p += 256) {
            const float a = part[(size_t)p * n + i], b = part[(size_t)(p + 64) * n + i], c = part[(size_t)(p + 128) * n + i],
                        d = part[(size_t)(p + 192) * n + i];
            s += (a + b) + (c + d);
        }
        for (; p < P; p += 64) s += part[(size_t)p * n + i];
    }
    l[lane][col] = s;
    __syncthreads();
    if (lane == 0 && i < n) {
        for (int k = 1; k < 64; ++k) s += l[k][col];
        out[i] = s;
        if (p0) {
            float* dst = i < C ? p0 + i : p1 + (i - C);
            *dst = accumulate ? *dst + s : s;
        }
    }
}

// out[p][w] = sum of rows p, p + P, p + 2P, ... of in[R][width] (fixed order): R partial rows -> P
__global__ __launch_bounds__(256) void fold_partials_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int width, int P) {
    const int p = blockIdx.x;
    for (int w = threadIdx.x; w < width; w += blockDim.x) {
        float s0 = 0.f, s1 = 0.f;
        int r = p;
        for (; r + P < R; r += 2 * P) { s0 += in[(size_t)r * width + w]; s1 += in[(size_t)(r + P) * width + w]; }
        if (r < R) s0 += in[(size_t)r * width + w];
        out[(size_t)p * width + w] = s0 + s1;
    }
}

// sums -> stats[0..C) = mean, stats[C..2C) = rstd; running buffers updated as nn.BatchNorm does (momentum 0.1,
// unbiased variance for the running estimate)
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ part, int P, float* __restrict__ stats,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var,
                                                           int64_t* __restrict__ nbt, int M, int C, float eps, float momentum) {
    __shared__ double l[2][64][17];
    const int col = threadIdx.x & 15, lane = threadIdx.x >> 4;          // 16 channels x 64 row lanes
    const int c = blockIdx.x * 16 + col;
    double s = 0.0, q = 0.0;
    if (c < C) {
        int p = lane;
        for (; p + 64 < P; p += 128) {
            const float s0 = part[(size_t)p * 2 * C + c], q0 = part[(size_t)p * 2 * C + C + c];
            const float s1 = part[(size_t)(p + 64) * 2 * C + c], q1 = part[(size_t)(p + 64) * 2 * C + C + c];
            s += (double)s0 + (double)s1;
            q += (double)q0 + (double)q1;
        }
        for (; p < P; p += 64) {
            s += part[(size_t)p * 2 * C + c];
            q += part[(size_t)p * 2 * C + C + c];
        }
    }
    l[0][lane][col] = s;
    l[1][lane][col] = q;
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) nbt[0] += 1;
    if (lane != 0 || c >= C) return;
    for (int k = 1; k < 64; ++k) { s += l[0][k][col]; q += l[1][k][col]; }
    const double mean = s / M;
    double var = q / M - mean * mean;
    if (var < 0) var = 0;
    stats[c] = (float)mean;
    stats[C + c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * M / (M > 1 ? M - 1 : 1));
    }
}
// eval mode: stats from the running buffers
__global__ __launch_bounds__(256) void bn_running_stats_kernel(const float* __restrict__ rm, const float* __restrict__ rv,
                                                               float* __restrict__ stats, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    stats[c] = rm[c];
    stats[C + c] = 1.0f / sqrtf(rv[c] + eps);
}

// synchronised BatchNorm, forward: partial rows -> sums[0..C) = sum y, sums[C..2C) = sum y^2, sums[2C] = M as doubles (the
// all-reduce hook then adds the ranks up), and statistics from such sums
__global__ __launch_bounds__(1024) void bn_sums_kernel(const float* __restrict__ part, int P, double* __restrict__ sums, int M, int C) {
    __shared__ double l[2][64][17];
    const int col = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + col;
    double s = 0.0, q = 0.0;
    if (c < C)
        for (int p = lane; p < P; p += 64) {
            s += part[(size_t)p * 2 * C + c];
            q += part[(size_t)p * 2 * C + C + c];
        }
    l[0][lane][col] = s;
    l[1][lane][col] = q;
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) sums[2 * C] = (double)M;
    if (lane != 0 || c >= C) return;
    for (int k = 1; k < 64; ++k) { s += l[0][k][col]; q += l[1][k][col]; }
    sums[c] = s;
    sums[C + c] = q;
}
__global__ __launch_bounds__(256) void bn_finalize_sums_kernel(const double* __restrict__ sums, float* __restrict__ stats,
                                                               float* __restrict__ running_mean, float* __restrict__ running_var,
                                                               int64_t* __restrict__ nbt, int C, float eps, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) nbt[0] += 1;
    if (c >= C) return;
    const double M = sums[2 * C];
    const double mean = sums[c] / M;
    double var = sums[C + c] / M - mean * mean;
    if (var < 0) var = 0;
    stats[c] = (float)mean;
    stats[C + c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * M / (M > 1 ? M - 1 : 1));
    }
}
__global__ void set_float_kernel(float* p, float v) { *p = v; }

__device__ __forceinline__ float lrelu(float z, float slope) { return z > 0.f ? z : slope * z; }

// four channels of the convolution output y at element index idx: y is fp32, or (YT = the 16-bit compute type, EOE_Y16 in the entry
// points' dtype argument) the 16-bit copy the conv GEMM wrote instead -- half the bytes of the three passes that read it
template <typename YT>
__device__ __forceinline__ f32x4 load_y4(const void* __restrict__ y, size_t idx) {
    if constexpr (sizeof(YT) == 4) {
        return *(const f32x4*)((const float*)y + idx);
    } else {
        float t[4];
        unpack4<YT>(*(const u32x2*)((const YT*)y + idx), t);
        return (f32x4){t[0], t[1], t[2], t[3]};
    }
}

// out = maxpool_P(leaky_relu(bn(y))); y fp32 [n,H,W,C] ; out 16-bit [n,H/P,W/P,C], or (nchw_flat) [n, C*(H/P)*(W/P)] in the
// reference's NCHW flatten order (cnn.py:83), or fp32 if out_f32.  P in {1,2}.  4 channels per thread.
template <typename T, typename YT>
__global__ __launch_bounds__(256) void bn_act_pool_fwd_kernel(const void* __restrict__ y, const float* __restrict__ stats,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              void* __restrict__ out, T* __restrict__ out16, int n, int H, int W,
                                                              int C, int P, int nchw_flat, int out_f32, float slope,
                                                              QuadDecode dec) {
    const int Ho = H / P, Wo = W / P, cc = C / 4;
    const unsigned total = (unsigned)n * Ho * Wo * cc;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        unsigned c4, opu, wo, ho, img;
        dec(i, c4, opu, wo, ho, img);
        const int c = (int)c4 * 4;
        const size_t op = opu;
        const f32x4 mu = *(const f32x4*)(stats + c), rs = *(const f32x4*)(stats + C + c);
        f32x4 g = {1.f, 1.f, 1.f, 1.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (gamma) { g = *(const f32x4*)(gamma + c); b = *(const f32x4*)(beta + c); }
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int dy = 0; dy < P; ++dy)
            for (int dx = 0; dx < P; ++dx) {
                const f32x4 v = load_y4<YT>(y, (((size_t)img * H + ho * P + dy) * W + wo * P + dx) * C + c);
#pragma unroll
                for (int r = 0; r < 4; ++r) best[r] = fmaxf(best[r], lrelu((v[r] - mu[r]) * rs[r] * g[r] + b[r], slope));
            }
        if (nchw_flat) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t o = ((size_t)img * C + c + r) * Ho * Wo + (size_t)ho * Wo + wo;
                if (out_f32) ((float*)out)[o] = best[r];
                else ((T*)out)[o] = (T)best[r];
            }
        } else if (out_f32) {
            *(f32x4*)((float*)out + op * C + c) = (f32x4){best[0], best[1], best[2], best[3]};
            if (out16) *(u32x2*)(out16 + op * C + c) = pack4<T>(best[0], best[1], best[2], best[3]);
        } else {
            *(u32x2*)((T*)out + op * C + c) = pack4<T>(best[0], best[1], best[2], best[3]);
        }
    }
}

// upstream gradient of one pooled output element (dout fp32, NHWC or NCHW-flat)
__device__ __forceinline__ float load_dout(const float* dout, int nchw_flat, int img, int ho, int wo, int c, int Ho, int Wo, int C) {
    return nchw_flat ? dout[((size_t)img * C + c) * Ho * Wo + (size_t)ho * Wo + wo]
                     : dout[(((size_t)img * Ho + ho) * Wo + wo) * C + c];
}

// Workgroup-level, FIXED-ORDER sum of the per-thread channel-quad accumulators of the BatchNorm backward reduce passes
// (a thread keeps one channel quad for its whole grid-stride loop): thread t holds quad ((blockIdx.x*256 + t) % cc); every
// output value adds its contributors t0, t0 + cc, t0 + 2cc, ... in that order -- no atomics, so the training step is bitwise
// reproducible.  stage: 256 x 8 floats of LDS (the kernels' dynamic LDS, >= 2*C floats, is separate); row: this workgroup's
// partial row [2*C] in global memory.
__device__ __forceinline__ void block_channel_sums(const float (&acc0)[4], const float (&acc1)[4], float* /*unused dynamic lds*/,
                                                   float* __restrict__ row, int C, size_t total) {
    __shared__ float stage[256][8];
    const int cc = C / 4;
    const size_t first = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = first < total;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        stage[threadIdx.x][r] = live ? acc0[r] : 0.f;
        stage[threadIdx.x][4 + r] = live ? acc1[r] : 0.f;
    }
    __syncthreads();
    const int base = (int)(((size_t)blockIdx.x * blockDim.x) % cc);          // quad of thread 0
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
        const int which = i / C, ch = i - which * C, quad = ch >> 2, r = ch & 3;
        int t0 = quad - base;
        if (t0 < 0) t0 += cc;
        float s = 0.f;
        for (int t = t0; t < 256; t += cc) s += stage[t][which * 4 + r];
        row[i] = s;
    }
}

// the gradient at the BatchNorm output of one (pooled position, channel quad): xh / z of the P x P window, the winning tap per channel
// and the upstream gradient (one 16-byte load in the NHWC layout)
template <int P, typename YT>
struct BnBwdElem {
    f32x4 xh[P * P], z[P * P];
    int arg[4];
    float d[4];
    unsigned opu, wo, ho, img;
    int c;
    __device__ __forceinline__ void load(unsigned i, const void* __restrict__ y, const float* __restrict__ dout, const f32x4& mu, const f32x4& rs,
                                         const f32x4& g, const f32x4& b, int nchw_flat, int H, int W, int C, float slope, const QuadDecode& dec) {
        unsigned c4;
        dec(i, c4, opu, wo, ho, img);
        c = (int)c4 * 4;
        const int Ho = H / P, Wo = W / P;
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        arg[0] = arg[1] = arg[2] = arg[3] = 0;
#pragma unroll
        for (int k = 0; k < P * P; ++k) {
            const int dy_ = k / P, dx_ = k % P;
            const f32x4 v = load_y4<YT>(y, (((size_t)img * H + ho * P + dy_) * W + wo * P + dx_) * C + c);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xh[k][r] = (v[r] - mu[r]) * rs[r];
                z[k][r] = xh[k][r] * g[r] + b[r];
                const float a = lrelu(z[k][r], slope);
                if (a > best[r]) { best[r] = a; arg[r] = k; }      // first maximum wins, as max_pool2d does
            }
        }
        if (!nchw_flat) {
            const f32x4 dv = *(const f32x4*)(dout + (((size_t)img * Ho + ho) * Wo + wo) * C + c);
            d[0] = dv[0]; d[1] = dv[1]; d[2] = dv[2]; d[3] = dv[3];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) d[r] = load_dout(dout, 1, img, ho, wo, c + r, Ho, Wo, C);
        }
    }
    // sum contributions of channel r: gz = gradient at the winning tap, gz * xhat there
    __device__ __forceinline__ void sums(int r, float slope, float& gz, float& gx) const {
        gz = 0.f;
        float xk = 0.f;
#pragma unroll
        for (int kk = 0; kk < P * P; ++kk)
            if (kk == arg[r]) { gz = d[r] * (z[kk][r] > 0.f ? 1.f : slope); xk = xh[kk][r]; }
        gx = gz * xk;
    }
};

// backward pass 1: per-channel sums of g and g*xhat, where g is the gradient at the BatchNorm OUTPUT (un-pooled through
// the first maximum of each window, times LeakyReLU').  One thread per (pooled position, 4 channels); LDS + atomics.
// mode 0 = reduce into red[0..C)=sum g, red[C..2C)=sum g*xhat;  mode 1 = write dy (16-bit [n*H*W, C]) using those sums.
template <typename T, int MODE, int P, typename YT>
__global__ __launch_bounds__(256) void bn_act_pool_bwd_kernel(const void* __restrict__ y, const float* __restrict__ stats,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ dout, float* __restrict__ red,
                                                              void* __restrict__ dy, int dy_f32, int n, int H, int W, int C,
                                                              int nchw_flat, int use_batch_stats, float slope, QuadDecode dec) {
    extern __shared__ float lds[];                // MODE 0: [2][C] partial sums
    const int Ho = H / P, Wo = W / P, cc = C / 4;
    const unsigned total = (unsigned)n * Ho * Wo * cc;
    // use_batch_stats == 2: synchronised BatchNorm, the all-reduced row count sits behind the sums
    const float invM = (MODE == 1 && use_batch_stats == 2) ? 1.0f / red[2 * C] : 1.0f / ((float)n * H * W);
    if (MODE == 0) {
        for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) lds[i] = 0.f;
        __syncthreads();
    }
    // MODE 0: the launch keeps gridDim.x * 256 a multiple of C/4, so a thread sees ONE channel quad for its whole loop
    // and sums in registers; LDS / global atomics only once per thread / workgroup at the end
    float acc0[4] = {0.f, 0.f, 0.f, 0.f}, acc1[4] = {0.f, 0.f, 0.f, 0.f};
    const unsigned first = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    // (block_channel_sums must be reached through ONE call site by every thread of the workgroup: inlined at two sites its static LDS
    //  staging array was two arrays, and the idle threads of a small problem met the busy ones at the barrier with different ones)
    // the thread's channel quad (and its per-channel constants) never change
    const int c_own = (int)(first % (unsigned)cc) * 4;
    const f32x4 mu = *(const f32x4*)(stats + c_own), rs = *(const f32x4*)(stats + C + c_own);
    f32x4 g = {1.f, 1.f, 1.f, 1.f}, b = {0.f, 0.f, 0.f, 0.f};
    if (gamma) { g = *(const f32x4*)(gamma + c_own); b = *(const f32x4*)(beta + c_own); }
    const bool fixed_quad = (stride % (unsigned)cc) == 0;       // always so in MODE 0 (the launch arranges it); MODE 1 reloads otherwise
    if (MODE == 0) {
        // two positions per iteration: twice the loads in flight (this pass ran at 3.7 TB/s with one), same summation order
        unsigned i = first;
        for (; i < total && i + stride < total && i + stride > i; i += 2 * stride) {
            BnBwdElem<P, YT> e0, e1;
            e0.load(i, y, dout, mu, rs, g, b, nchw_flat, H, W, C, slope, dec);
            e1.load(i + stride, y, dout, mu, rs, g, b, nchw_flat, H, W, C, slope, dec);
#pragma unroll
            for (int r = 0; r < 4; ++r) { float gz, gx; e0.sums(r, slope, gz, gx); acc0[r] += gz; acc1[r] += gx; }
#pragma unroll
            for (int r = 0; r < 4; ++r) { float gz, gx; e1.sums(r, slope, gz, gx); acc0[r] += gz; acc1[r] += gx; }
        }
        for (; i < total; i += stride) {
            BnBwdElem<P, YT> e0;
            e0.load(i, y, dout, mu, rs, g, b, nchw_flat, H, W, C, slope, dec);
#pragma unroll
            for (int r = 0; r < 4; ++r) { float gz, gx; e0.sums(r, slope, gz, gx); acc0[r] += gz; acc1[r] += gx; }
            if (i + stride < i) break;
        }
        block_channel_sums(acc0, acc1, lds, red + (size_t)(1 + blockIdx.x) * 2 * C, C, total);
        return;
    }
    for (unsigned i = first; i < total; i += stride) {
        f32x4 mu_ = mu, rs_ = rs, g_ = g, b_ = b;
        if (!fixed_quad) {
            const int c = (int)(i % (unsigned)cc) * 4;
            mu_ = *(const f32x4*)(stats + c); rs_ = *(const f32x4*)(stats + C + c);
            if (gamma) { g_ = *(const f32x4*)(gamma + c); b_ = *(const f32x4*)(beta + c); }
        }
        BnBwdElem<P, YT> e;
        e.load(i, y, dout, mu_, rs_, g_, b_, nchw_flat, H, W, C, slope, dec);
        const int c = e.c;
        const f32x4 s1 = *(const f32x4*)(red + c), s2 = *(const f32x4*)(red + C + c);
#pragma unroll
        for (int k = 0; k < P * P; ++k) {
            const int dy_ = k / P, dx_ = k % P;
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float gz = (k == e.arg[r]) ? e.d[r] * (e.z[k][r] > 0.f ? 1.f : slope) : 0.f;
                // training: dy = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat)); eval (running stats): gamma*rstd*g
                o[r] = use_batch_stats ? g_[r] * rs_[r] * (gz - s1[r] * invM - e.xh[k][r] * s2[r] * invM) : g_[r] * rs_[r] * gz;
            }
            const size_t oo = (((size_t)e.img * H + e.ho * P + dy_) * W + e.wo * P + dx_) * C + c;
            if (dy_f32) *(f32x4*)((float*)dy + oo) = (f32x4){o[0], o[1], o[2], o[3]};
            else *(u32x2*)((T*)dy + oo) = pack4<T>(o[0], o[1], o[2], o[3]);
        }
        if (i + stride < i) break;
    }
}

// ---------------------------------------------------------------------------------------------- BN + act + overlapping MaxPool
// out = maxpool_{k,stride,pad}(act(bn(y))) in one pass over y (the stem of resnet.py:93-96: the 112x112x64 activation is
// never written); idx = winning tap (first maximum), out16 = optional 16-bit copy
template <typename T, typename YT>
__global__ __launch_bounds__(256) void bn_act_maxpool_fwd_kernel(const void* __restrict__ y, const float* __restrict__ stats,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float* __restrict__ out, T* __restrict__ out16,
                                                                 uint8_t* __restrict__ idx, int n, int H, int W, int C, int k,
                                                                 int stride, int pad, int Ho, int Wo, float slope, QuadDecode dec) {
    const int cc = C / 4;
    const unsigned total = (unsigned)n * Ho * Wo * cc;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        unsigned c4, opu, wou, hou, img;
        dec(i, c4, opu, wou, hou, img);
        const int c = (int)c4 * 4, wo = (int)wou, ho = (int)hou;
        const size_t op = opu;
        const f32x4 mu = *(const f32x4*)(stats + c), rs = *(const f32x4*)(stats + C + c);
        f32x4 g = {1.f, 1.f, 1.f, 1.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (gamma) { g = *(const f32x4*)(gamma + c); b = *(const f32x4*)(beta + c); }
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int arg[4] = {0, 0, 0, 0};
        for (int ky = 0; ky < k; ++ky) {
            const int h = ho * stride + ky - pad;
            if (h < 0 || h >= H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int w = wo * stride + kx - pad;
                if (w < 0 || w >= W) continue;
                const f32x4 v = load_y4<YT>(y, (((size_t)img * H + h) * W + w) * C + c);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = lrelu((v[r] - mu[r]) * rs[r] * g[r] + b[r], slope);
                    if (a > best[r]) { best[r] = a; arg[r] = ky * k + kx; }      // taps in increasing order: first maximum wins
                }
            }
        }
        *(f32x4*)(out + op * C + c) = (f32x4){best[0], best[1], best[2], best[3]};
        if (out16) *(u32x2*)(out16 + op * C + c) = pack4<T>(best[0], best[1], best[2], best[3]);
        *(uint32_t*)(idx + op * C + c) = (uint32_t)arg[0] | ((uint32_t)arg[1] << 8) | ((uint32_t)arg[2] << 16) | ((uint32_t)arg[3] << 24);
    }
}
// backward, one thread per (pixel of y, 4 channels): gradient at the BN output = sum of the windows this pixel won (gather
// form, as maxpool_bwd_kernel) times act'.  MODE 0 = per-channel sums of g and g*xhat into this workgroup's partial row;
// MODE 1 = dy (16-bit) from those sums.
// S = the pooling stride as a compile-time constant (1 or 2; 0 = use the run-time value): the window enumeration below
// divides by it twice per pixel
template <typename T, int MODE, int S, typename YT>
__global__ __launch_bounds__(256) void bn_act_maxpool_bwd_kernel(const void* __restrict__ y, const float* __restrict__ stats,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const float* __restrict__ dout, const uint8_t* __restrict__ idx,
                                                                 float* __restrict__ red, T* __restrict__ dy, int n, int H, int W,
                                                                 int C, int k, int stride, int pad, int Ho, int Wo,
                                                                 int use_batch_stats, float slope, QuadDecode dec) {
    extern __shared__ float lds[];
    const int cc = C / 4;
    const unsigned total = (unsigned)n * H * W * cc;
    const float invM = (MODE == 1 && use_batch_stats == 2) ? 1.0f / red[2 * C] : 1.0f / ((float)n * H * W);
    if (MODE == 0) {
        for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) lds[i] = 0.f;
        __syncthreads();
    }
    float acc0[4] = {0.f, 0.f, 0.f, 0.f}, acc1[4] = {0.f, 0.f, 0.f, 0.f};
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        unsigned c4, ipu, wu, hu, img;
        dec(i, c4, ipu, wu, hu, img);
        const int c = (int)c4 * 4, w = (int)wu, h = (int)hu;
        const size_t ip = ipu;
        const f32x4 mu = *(const f32x4*)(stats + c), rs = *(const f32x4*)(stats + C + c);
        f32x4 g = {1.f, 1.f, 1.f, 1.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (gamma) { g = *(const f32x4*)(gamma + c); b = *(const f32x4*)(beta + c); }
        const f32x4 v = load_y4<YT>(y, ip * C + c);
        float gs[4] = {0.f, 0.f, 0.f, 0.f};
        {   // the windows (ho, wo) that contain this pixel: ho*st <= h+pad <= ho*st + k-1
            const int st = S ? S : stride;
            const int hp = h + pad, wp = w + pad;
            const int ho_hi = min(hp / st, Ho - 1), ho_lo = max(0, (hp - k + st) / st);
            const int wo_hi = min(wp / st, Wo - 1), wo_lo = max(0, (wp - k + st) / st);
            for (int ho = ho_lo; ho <= ho_hi; ++ho)
                for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                    const int tap = (hp - ho * st) * k + (wp - wo * st);
                    const size_t o = (((size_t)img * Ho + ho) * Wo + wo) * C + c;
                    const uint32_t a = *(const uint32_t*)(idx + o);
                    const f32x4 d = *(const f32x4*)(dout + o);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if ((int)((a >> (8 * r)) & 255u) == tap) gs[r] += d[r];
                }
        }
        float o4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float xh = (v[r] - mu[r]) * rs[r];
            const float gz = gs[r] * (xh * g[r] + b[r] > 0.f ? 1.f : slope);
            if (MODE == 0) {
                acc0[r] += gz;
                acc1[r] += gz * xh;
            } else {
                const float s1 = red[c + r], s2 = red[C + c + r];
                o4[r] = use_batch_stats ? g[r] * rs[r] * (gz - s1 * invM - xh * s2 * invM) : g[r] * rs[r] * gz;
            }
        }
        if (MODE == 1) {
            if constexpr (sizeof(T) == 4) *(f32x4*)(dy + ip * C + c) = f32x4{o4[0], o4[1], o4[2], o4[3]};      // parity mode: fp32 dY
            else *(u32x2*)(dy + ip * C + c) = pack4<T>(o4[0], o4[1], o4[2], o4[3]);
        }
    }
    if (MODE == 0) block_channel_sums(acc0, acc1, lds, red + (size_t)(1 + blockIdx.x) * 2 * C, C, total);
}

// The stem's shape (3x3 / stride 2 / pad 1 windows over an even H x W map), one thread per 2x2 PIXEL BLOCK and 4 channels: the four
// pixels of block (a, b) only ever belong to the four windows (a | a+1, b | b+1), so their winner indices and upstream gradients are
// loaded once (4 index words + 4 gradient vectors per thread) instead of once per pixel and window (9 of each for the same four pixels
// in the generic kernel above, which ran at 2.4-2.7 TB/s on the 822 MB stem activation); same arithmetic per pixel.
template <typename T, int MODE, typename YT>
__global__ __launch_bounds__(256) void bn_act_maxpool_bwd_s2k3_kernel(const void* __restrict__ y, const float* __restrict__ stats,
                                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                      const float* __restrict__ dout, const uint8_t* __restrict__ idx,
                                                                      float* __restrict__ red, T* __restrict__ dy, int n, int H, int W,
                                                                      int C, int Ho, int Wo, int use_batch_stats, float slope,
                                                                      QuadDecode dec) {
    extern __shared__ float lds[];
    const int cc = C / 4, Hb = H / 2, Wb = W / 2;
    const unsigned total = (unsigned)n * Hb * Wb * cc;
    const float invM = (MODE == 1 && use_batch_stats == 2) ? 1.0f / red[2 * C] : 1.0f / ((float)n * H * W);
    if (MODE == 0) {
        for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) lds[i] = 0.f;
        __syncthreads();
    }
    float acc0[4] = {0.f, 0.f, 0.f, 0.f}, acc1[4] = {0.f, 0.f, 0.f, 0.f};
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        unsigned c4, bpu, bu, au, img;
        dec(i, c4, bpu, bu, au, img);
        const int c = (int)c4 * 4, a = (int)au, b = (int)bu;
        const f32x4 mu = *(const f32x4*)(stats + c), rs = *(const f32x4*)(stats + C + c);
        f32x4 g = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
        if (gamma) { g = *(const f32x4*)(gamma + c); be = *(const f32x4*)(beta + c); }
        // windows (a + wy, b + wx): winner index (one byte per channel) and upstream gradient; a window off the map contributes nothing
        uint32_t wi[2][2];
        f32x4 wd[2][2];
#pragma unroll
        for (int wy = 0; wy < 2; ++wy)
#pragma unroll
            for (int wx = 0; wx < 2; ++wx) {
                const int ho = a + wy, wo = b + wx;
                if (ho < Ho && wo < Wo) {
                    const size_t o = (((size_t)img * Ho + ho) * Wo + wo) * C + c;
                    wi[wy][wx] = *(const uint32_t*)(idx + o);
                    wd[wy][wx] = *(const f32x4*)(dout + o);
                } else {
                    wi[wy][wx] = 0xffffffffu;              // tap 255 never matches
                    wd[wy][wx] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                const int h = 2 * a + py, w = 2 * b + px;
                const size_t ip = ((size_t)img * H + h) * W + w;
                const f32x4 v = load_y4<YT>(y, ip * C + c);
                // pixel (2a + py, 2b + px) sits at tap (py + 1 - 2 wy, px + 1 - 2 wx) of window (a + wy, b + wx): py = 0 -> wy = 0 only
                float gs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int wy = 0; wy <= py; ++wy)
#pragma unroll
                    for (int wx = 0; wx <= px; ++wx) {
                        const int tap = (py + 1 - 2 * wy) * 3 + (px + 1 - 2 * wx);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if ((int)((wi[wy][wx] >> (8 * r)) & 255u) == tap) gs[r] += wd[wy][wx][r];
                    }
                float o4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float xh = (v[r] - mu[r]) * rs[r];
                    const float gz = gs[r] * (xh * g[r] + be[r] > 0.f ? 1.f : slope);
                    if (MODE == 0) {
                        acc0[r] += gz;
                        acc1[r] += gz * xh;
                    } else {
                        const float s1 = red[c + r], s2 = red[C + c + r];
                        o4[r] = use_batch_stats ? g[r] * rs[r] * (gz - s1 * invM - xh * s2 * invM) : g[r] * rs[r] * gz;
                    }
                }
                if (MODE == 1) {
                    if constexpr (sizeof(T) == 4) *(f32x4*)(dy + ip * C + c) = f32x4{o4[0], o4[1], o4[2], o4[3]};
                    else *(u32x2*)(dy + ip * C + c) = pack4<T>(o4[0], o4[1], o4[2], o4[3]);
                }
            }
    }
    if (MODE == 0) block_channel_sums(acc0, acc1, lds, red + (size_t)(1 + blockIdx.x) * 2 * C, C, total);
}

// out[c] = sum over the rows of an fp32 matrix [rows, C] (C % 4 == 0): the bias gradient of a convolution in the exact-fp32 mode.  The
// BatchNorm reduce passes' scheme: a thread keeps ONE channel quad for its whole grid-stride loop (two rows in flight), the workgroup's
// 256 threads are added in a fixed order into its partial row, reduce_partials_kernel adds the rows -- no atomics.  (eoe_cast_colsum, built
// for the ViT's 768-wide matrices, used 8 of its 64 column lanes on a 32-channel map and walked 256 rows serially per thread: 0.18 of
// CNN32's 1.75 ms step in that mode.)
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ x, float* __restrict__ red, int rows, int C) {
    extern __shared__ float lds[];
    const int cc = C / 4;
    const unsigned total = (unsigned)rows * cc;
    float acc0[4] = {0.f, 0.f, 0.f, 0.f}, acc1[4] = {0.f, 0.f, 0.f, 0.f};
    const unsigned first = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    unsigned i = first;
    for (; i < total && i + stride < total && i + stride > i; i += 2 * stride) {
        const f32x4 a = *(const f32x4*)(x + (size_t)i * 4), b = *(const f32x4*)(x + (size_t)(i + stride) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc0[r] += a[r];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc0[r] += b[r];
    }
    for (; i < total; i += stride) {
        const f32x4 a = *(const f32x4*)(x + (size_t)i * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc0[r] += a[r];
        if (i + stride < i) break;
    }
    block_channel_sums(acc0, acc1, lds, red + (size_t)(1 + blockIdx.x) * 2 * C, C, total);
}

int grid_for(size_t total) {
    size_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

// flat (pixel, channel-quad) indices are 32-bit in the streaming kernels (FDiv needs x < 2^31)
#define EOE_CHECK_IDX(total, who) EOE_CHECK_ARG((size_t)(total) < 0x7fffffffull, "%s: tensor too large for 32-bit indexing", who)

#define DISPATCH_T(dtype, ...)                                  \
    do {                                                        \
        if ((dtype) == EOE_F16) { typedef f16_t T; __VA_ARGS__; } \
        else if ((dtype) == EOE_BF16) { typedef bf16_t T; __VA_ARGS__; } \
        else return eoe_set_error(EOE_ERR_ARG, "bad dtype %d", (int)(dtype)); \
    } while (0)

// T = the 16-bit compute type, YT = the type of the convolution output y the BatchNorm kernels read: fp32, or T with EOE_Y16
#define DISPATCH_TY(dtype, y16, ...)                                                       \
    do {                                                                                   \
        if ((dtype) == EOE_F16) { typedef f16_t T; if (y16) { typedef f16_t YT; __VA_ARGS__; } else { typedef float YT; __VA_ARGS__; } }     \
        else if ((dtype) == EOE_BF16) { typedef bf16_t T; if (y16) { typedef bf16_t YT; __VA_ARGS__; } else { typedef float YT; __VA_ARGS__; } } \
        else return eoe_set_error(EOE_ERR_ARG, "bad dtype %d", (int)(dtype));              \
    } while (0)

static int check_geo(const char* who, int H, int W, int kh, int kw, int stride, int pad, Geo& g) {
    EOE_CHECK_ARG(kh >= 1 && kw >= 1 && stride >= 1 && pad >= 0, "%s: bad geometry", who);
    g.kh = kh; g.kw = kw; g.stride = stride; g.pad = pad;
    g.Ho = (H + 2 * pad - kh) / stride + 1;
    g.Wo = (W + 2 * pad - kw) / stride + 1;
    EOE_CHECK_ARG(g.Ho >= 1 && g.Wo >= 1, "%s: empty output", who);
    return 0;
}

extern "C" int eoe_im2col(const void* x, int x_kind, const float* mean, const float* stdv, void* out, int n, int cin,
                          int H, int W, int kh, int kw, int stride, int pad, int Kp, int dtype, void* stream) {
    EOE_CHECK_ARG(x && out && n > 0 && cin > 0 && H > 0 && W > 0, "im2col: bad args");
    Geo g;
    EOE_TRY(check_geo("im2col", H, W, kh, kw, stride, pad, g));
    EOE_CHECK_ARG(Kp >= kh * kw * cin && Kp % 64 == 0, "im2col: Kp = %d must be a multiple of 64 and >= kh*kw*cin", Kp);
    EOE_CHECK_ARG((mean == nullptr) == (stdv == nullptr), "im2col: mean/std must both be given or both NULL");
    const size_t rows = (size_t)n * g.Ho * g.Wo;
    ProfScope ps("im2col", 0, 2.0 * rows * Kp, stream);
    if (x_kind == 1) {
        DISPATCH_T(dtype, hipLaunchKernelGGL((im2col_img_kernel<T>), dim3(grid_for(rows * Kp)), dim3(256), 0,
                                             (hipStream_t)stream, (const float*)x, mean, stdv, (T*)out, n, cin, H, W, Kp, g));
    } else {
        EOE_CHECK_ARG(cin % 8 == 0, "im2col: hidden layers need cin %% 8 == 0");
        EOE_CHECK_ARG(x_kind == 0 || x_kind == 2, "im2col: x_kind must be 0 (16-bit NHWC), 1 (fp32 NCHW image) or 2 (fp32 NHWC)");
        if (x_kind == 2) {
            DISPATCH_T(dtype, hipLaunchKernelGGL((im2col_nhwc_kernel<T, true>), dim3(grid_for(rows * Kp / 8)), dim3(256), 0,
                                                 (hipStream_t)stream, x, (T*)out, n, H, W, cin, Kp, g));
        } else {
            DISPATCH_T(dtype, hipLaunchKernelGGL((im2col_nhwc_kernel<T, false>), dim3(grid_for(rows * Kp / 8)), dim3(256), 0,
                                                 (hipStream_t)stream, x, (T*)out, n, H, W, cin, Kp, g));
        }
    }
    EOE_CHECK_LAUNCH("im2col");
    return 0;
}

extern "C" int eoe_col2im(const void* dpatches, float* dx, int n, int C, int H, int W, int kh, int kw, int stride, int pad,
                          int Kp, int dtype, int accumulate, void* stream) {
    EOE_CHECK_ARG(dpatches && dx && n > 0 && C % 4 == 0 && Kp >= kh * kw * C, "col2im: bad args");
    Geo g;
    EOE_TRY(check_geo("col2im", H, W, kh, kw, stride, pad, g));
    EOE_CHECK_IDX((size_t)n * H * W * C / 4, "col2im");
    ProfScope ps("col2im", 0, 2.0 * n * g.Ho * g.Wo * kh * kw * C + 4.0 * n * H * W * C, stream);
    const QuadDecode dec(C / 4, W, H);
#define EOE_C2I(SS)                                                                                                      \
    DISPATCH_T(dtype, hipLaunchKernelGGL((col2im_kernel<T, SS>), dim3(grid_for((size_t)n * H * W * C / 4)), dim3(256), 0, \
                                         (hipStream_t)stream, (const T*)dpatches, dx, n, H, W, C, Kp, g, dec, accumulate))
    if (stride == 2) { EOE_C2I(2); } else if (stride == 1) { EOE_C2I(1); } else { EOE_C2I(0); }
#undef EOE_C2I
    EOE_CHECK_LAUNCH("col2im");
    return 0;
}

extern "C" int eoe_conv_pack_weight_multi(const eoe_conv_pack_job* jobs, int count, int dtype, void* stream) {
    EOE_CHECK_ARG(jobs && count > 0, "conv_pack_weight_multi: bad args");
    for (int first = 0; first < count; first += PACK_MAX_JOBS) {
        PackBatch b;
        b.count = count - first < PACK_MAX_JOBS ? count - first : PACK_MAX_JOBS;
        int blocks = 0;
        for (int i = 0; i < b.count; ++i) {
            const eoe_conv_pack_job& j = jobs[first + i];
            EOE_CHECK_ARG(j.w && j.w16 && j.cout > 0 && j.cin > 0 && j.cpad >= j.cin && j.Kp >= j.kh * j.kw * j.cpad,
                          "conv_pack_weight_multi: job %d: bad args", first + i);
            EOE_CHECK_IDX((size_t)j.cout * j.Kp, "conv_pack_weight_multi");
            b.w[i] = j.w; b.w16[i] = j.w16; b.w16t[i] = j.w16t; b.w16d[i] = j.w16d;
            b.cout[i] = j.cout; b.cin[i] = j.cin; b.cpad[i] = j.cpad; b.Kp[i] = j.Kp; b.taps[i] = j.kh * j.kw;
            b.block_start[i] = blocks;
            blocks += cdiv(j.cout, 64) * cdiv(j.Kp, 64);
        }
        b.block_start[b.count] = blocks;
        DISPATCH_T(dtype, hipLaunchKernelGGL((conv_pack_multi_kernel<T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, b));
        EOE_CHECK_LAUNCH("conv_pack_weight_multi");
    }
    return 0;
}

extern "C" int eoe_conv_pack_weight(const float* w, void* w16, void* w16t, void* w16d, int cout, int cin, int cpad, int kh, int kw,
                                    int Kp, int dtype, void* stream) {
    eoe_conv_pack_job job = {w, w16, w16t, w16d, cout, cin, cpad, kh, kw, Kp};
    return eoe_conv_pack_weight_multi(&job, 1, dtype, stream);
}

extern "C" int eoe_conv_unpack_wgrad(const float* g, float* dw, int cout, int cin, int cpad, int kh, int kw, int Kp, int transposed,
                                     int accumulate, void* stream) {
    EOE_CHECK_ARG(g && dw && cout > 0 && cin > 0 && cpad >= cin && Kp >= kh * kw * cpad, "conv_unpack_wgrad: bad args");
    hipLaunchKernelGGL(conv_unpack_kernel, dim3(grid_for((size_t)cout * cin * kh * kw)), dim3(256), 0, (hipStream_t)stream, g, dw,
                       cout, cin, cpad, Kp, kh * kw, transposed, accumulate);
    EOE_CHECK_LAUNCH("conv_unpack_wgrad");
    return 0;
}

extern "C" int eoe_stem_pack_image(const float* x, const float* mean, const float* stdv, void* out, int n, int H, int W, int Hp,
                                   int Wp, int pad, int cpad, int dtype, void* stream) {
    EOE_CHECK_ARG(x && out && n > 0 && H > 0 && W > 0 && pad >= 0 && Hp >= H + pad && Wp >= W + pad && (cpad == 4 || cpad == 8),
                  "stem_pack_image: bad args");
    EOE_CHECK_ARG((mean == nullptr) == (stdv == nullptr), "stem_pack_image: mean/std must both be given or both NULL");
    ProfScope ps("stem_pack", 0, 12.0 * n * H * W + 8.0 * n * Hp * Wp, stream);
    if (cpad == 4) {
        DISPATCH_T(dtype, hipLaunchKernelGGL((stem_pack_image_kernel<T, 4>), dim3(grid_for((size_t)n * Hp * Wp)), dim3(256), 0,
                                             (hipStream_t)stream, x, mean, stdv, (T*)out, n, H, W, Hp, Wp, pad));
    } else {
        DISPATCH_T(dtype, hipLaunchKernelGGL((stem_pack_image_kernel<T, 8>), dim3(grid_for((size_t)n * Hp * Wp)), dim3(256), 0,
                                             (hipStream_t)stream, x, mean, stdv, (T*)out, n, H, W, Hp, Wp, pad));
    }
    EOE_CHECK_LAUNCH("stem_pack_image");
    return 0;
}

extern "C" int eoe_stem_pack_weight(const float* w, void* w16, int cout, int kh, int kw, int dtype, void* stream) {
    EOE_CHECK_ARG(w && w16 && cout > 0 && kh > 0 && kw > 0 && kw <= 8, "stem_pack_weight: bad args");
    const int K = (kh + 1) / 2 * 64;
    DISPATCH_T(dtype, hipLaunchKernelGGL((stem_pack_weight_kernel<T>), dim3(grid_for((size_t)cout * K)), dim3(256), 0,
                                         (hipStream_t)stream, w, (T*)w16, cout, kh, kw, K));
    EOE_CHECK_LAUNCH("stem_pack_weight");
    return 0;
}

extern "C" int eoe_stem_unpack_wgrad(const float* g, float* dw, int cout, int kh, int kw, void* stream) {
    EOE_CHECK_ARG(g && dw && cout > 0 && kh > 0 && kw > 0 && kw <= 8, "stem_unpack_wgrad: bad args");
    hipLaunchKernelGGL(stem_unpack_kernel, dim3(grid_for((size_t)cout * 3 * kh * kw)), dim3(256), 0, (hipStream_t)stream, g, dw, cout,
                       kh, kw);
    EOE_CHECK_LAUNCH("stem_unpack_wgrad");
    return 0;
}

// ---- synchronised BatchNorm: a registered all-reduce is called at every training-mode reduction point (include/eoe_hip.h)
static eoe_allreduce_fn g_bn_sync_fn = nullptr;
static void* g_bn_sync_user = nullptr;
extern "C" int eoe_set_bn_sync(eoe_allreduce_fn fn, void* user) {
    g_bn_sync_fn = fn;
    g_bn_sync_user = fn ? user : nullptr;
    return 0;
}
bool eoe_bn_sync_active() { return g_bn_sync_fn != nullptr; }
void* eoe_bn_sync_user() { return g_bn_sync_fn ? g_bn_sync_user : nullptr; }
int eoe_bn_sync_allreduce(void* buf, int64_t count, int is_f64, void* stream) {
    if (!g_bn_sync_fn) return eoe_set_error(EOE_ERR_ARG, "bn sync: no hook registered");
    if (g_bn_sync_fn(g_bn_sync_user, buf, count, is_f64, stream) != 0) return eoe_set_error(EOE_ERR_LAUNCH, "bn sync: the all-reduce hook failed");
    return 0;
}

// partial rows [P][2C] -> stats (+ running buffers); with the hook: -> double sums behind the rows -> all-reduce -> stats
static int bn_finalize_rows(const float* rows, int P, float* sums_scratch, float* stats, float* running_mean, float* running_var,
                            int64_t* nbt, int M, int C, float eps, float momentum, hipStream_t s) {
    if (!g_bn_sync_fn) {
        hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 16)), dim3(1024), 0, s, rows, P, stats, running_mean, running_var, nbt, M, C,
                           eps, momentum);
        EOE_CHECK_LAUNCH("bn_finalize");
        return 0;
    }
    double* dsum = (double*)(sums_scratch + (size_t)EOE_BN_PARTIALS * 2 * C);        // the 3 spare rows of EOE_BN_SCRATCH: 3C doubles
    hipLaunchKernelGGL(bn_sums_kernel, dim3(cdiv(C, 16)), dim3(1024), 0, s, rows, P, dsum, M, C);
    EOE_CHECK_LAUNCH("bn_sums");
    EOE_TRY(eoe_bn_sync_allreduce(dsum, 2 * (int64_t)C + 1, 1, (void*)s));
    hipLaunchKernelGGL(bn_finalize_sums_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, (const double*)dsum, stats, running_mean, running_var,
                       nbt, C, eps, momentum);
    EOE_CHECK_LAUNCH("bn_finalize_sums");
    return 0;
}

extern "C" int eoe_bn_stats(const float* y, float* sums_scratch, float* stats, float* running_mean, float* running_var,
                            int64_t* num_batches_tracked, int M, int C, float eps, float momentum, int training, void* stream) {
    EOE_CHECK_ARG(stats && M > 0 && C > 0, "bn_stats: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (!training) {
        EOE_CHECK_ARG(running_mean && running_var, "bn_stats: eval mode needs the running buffers");
        hipLaunchKernelGGL(bn_running_stats_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, running_mean, running_var, stats, C, eps);
        EOE_CHECK_LAUNCH("bn_running_stats");
        return 0;
    }
    EOE_CHECK_ARG(y && sums_scratch, "bn_stats: bad args");
    const int vec = (C % 4 == 0) ? 4 : 1;
    const int cols = C / vec;                     // thread-columns needed
    int cpb = 1;
    while (cpb < 64 && cpb < cols) cpb *= 2;      // power of two <= 64 so that it divides 256
    ProfScope ps("bn_stats", 0, 4.0 * M * C, stream);
    const int rpb = 256 / cpb;
    int gy = cdiv(M, rpb * 16);
    const int gx = cdiv(cols, cpb);
    if (gy * gx > 1024) gy = 1024 / gx;
    if (gy > 512) gy = 512;
    if (gy < 1) gy = 1;
    if (vec == 4) hipLaunchKernelGGL(bn_stats_kernel<4>, dim3(gx, gy), dim3(256), 0, s, y, sums_scratch, M, C, cpb);
    else hipLaunchKernelGGL(bn_stats_kernel<1>, dim3(gx, gy), dim3(256), 0, s, y, sums_scratch, M, C, cpb);
    EOE_CHECK_LAUNCH("bn_stats");
    return bn_finalize_rows(sums_scratch, gy, sums_scratch, stats, running_mean, running_var, num_batches_tracked, M, C, eps, momentum, s);
}

extern "C" int eoe_bn_stats_partials(const float* part, int R, float* sums_scratch, float* stats, float* running_mean,
                                     float* running_var, int64_t* num_batches_tracked, int M, int C, float eps, float momentum,
                                     void* stream) {
    EOE_CHECK_ARG(part && sums_scratch && stats && R > 0 && M > 0 && C > 0, "bn_stats_partials: bad args");
    hipStream_t s = (hipStream_t)stream;
    const float* rows = part;
    int P = R;
    if (R > EOE_BN_PARTIALS) {          // fold R rows into EOE_BN_PARTIALS (a streaming pass over the partial buffer)
        P = EOE_BN_PARTIALS;
        hipLaunchKernelGGL(fold_partials_kernel, dim3(P), dim3(256), 0, s, part, sums_scratch, R, 2 * C, P);
        EOE_CHECK_LAUNCH("bn_fold_partials");
        rows = sums_scratch;
    }
    return bn_finalize_rows(rows, P, sums_scratch, stats, running_mean, running_var, num_batches_tracked, M, C, eps, momentum, s);
}

extern "C" int eoe_bn_act_pool_fwd(const void* y, const float* stats, const float* gamma, const float* beta, void* out,
                                   void* out16, int n, int H, int W, int C, int pool, int nchw_flat, int out_f32, float slope,
                                   int dtype, void* stream) {
    const int y16 = dtype & EOE_Y16;
    dtype &= ~EOE_Y16;
    EOE_CHECK_ARG(!out16 || (out_f32 && !nchw_flat), "bn_act_pool_fwd: the extra 16-bit copy goes with an fp32 NHWC output");
    EOE_CHECK_ARG(y && stats && out && n > 0 && C % 4 == 0, "bn_act_pool_fwd: bad args");
    EOE_CHECK_ARG((pool == 1 || pool == 2) && H % pool == 0 && W % pool == 0, "bn_act_pool: pool must be 1 or 2 and divide H, W");
    EOE_CHECK_ARG((gamma == nullptr) == (beta == nullptr), "bn_act_pool: gamma/beta must both be given or both NULL");
    EOE_CHECK_IDX((size_t)n * H * W * C / 4, "bn_act_pool_fwd");
    ProfScope ps("bn_act_pool_fwd", 0, (y16 ? 2.0 : 4.0) * n * H * W * C + 2.0 * n * H * W * C / (pool * pool), stream);
    const QuadDecode dec(C / 4, W / pool, H / pool);
    DISPATCH_TY(dtype, y16, hipLaunchKernelGGL((bn_act_pool_fwd_kernel<T, YT>), dim3(grid_for((size_t)n * (H / pool) * (W / pool) * C / 4)),
                                         dim3(256), 0, (hipStream_t)stream, y, stats, gamma, beta, out, (T*)out16, n, H, W, C,
                                         pool, nchw_flat, out_f32, slope, dec));
    EOE_CHECK_LAUNCH("bn_act_pool_fwd");
    return 0;
}

extern "C" int eoe_bn_act_pool_bwd(const void* y, const float* stats, const float* gamma, const float* beta, const float* dout,
                                   float* red_scratch, void* dy, int dy_f32, float* dgamma, float* dbeta, int n, int H, int W,
                                   int C, int pool, int nchw_flat, int training, int accumulate, float slope, int dtype,
                                   void* stream) {
    const int y16 = dtype & EOE_Y16;
    dtype &= ~EOE_Y16;
    EOE_CHECK_ARG(y && stats && dout && red_scratch && dy && n > 0 && C % 4 == 0, "bn_act_pool_bwd: bad args");
    EOE_CHECK_ARG((pool == 1 || pool == 2) && H % pool == 0 && W % pool == 0, "bn_act_pool: pool must be 1 or 2 and divide H, W");
    EOE_CHECK_ARG((gamma == nullptr) == (beta == nullptr) && (dgamma == nullptr) == (dbeta == nullptr), "bn_act_pool_bwd: gamma/beta pairs");
    EOE_CHECK_ARG(C <= 4096, "bn_act_pool_bwd: C too large");
    hipStream_t s = (hipStream_t)stream;
    EOE_CHECK_IDX((size_t)n * H * W * C / 4, "bn_act_pool_bwd");
    ProfScope ps("bn_act_pool_bwd", 0, 2 * (y16 ? 2.0 : 4.0) * n * H * W * C + 2.0 * n * H * W * C, stream);
    const QuadDecode dec(C / 4, W / pool, H / pool);
    const int grid = grid_for((size_t)n * (H / pool) * (W / pool) * C / 4);
    int g0 = grid > EOE_BN_PARTIALS ? EOE_BN_PARTIALS : grid;
    {   // reduce pass: gridDim.x * 256 must be a multiple of C/4 (one channel quad per thread)
        int cc = C / 4, a = cc, b = 256;
        while (b) { const int t = a % b; a = b; b = t; }
        const int q = cc / a;                     // cc / gcd(cc, 256)
        EOE_CHECK_ARG(q <= EOE_BN_PARTIALS, "bn_act_pool_bwd: C = %d not supported", C);
        g0 = g0 / q * q;
        if (g0 < q) g0 = q;
    }
#define EOE_BNB(MODE, PP, GRID, LDS)                                                                                  \
    DISPATCH_TY(dtype, y16, hipLaunchKernelGGL((bn_act_pool_bwd_kernel<T, MODE, PP, YT>), dim3(GRID), dim3(256), LDS, s, y, stats, gamma, \
                                         beta, dout, red_scratch, dy, dy_f32, n, H, W, C, nchw_flat, training, slope, dec))
    if (pool == 1) { EOE_BNB(0, 1, g0, 2 * C * sizeof(float)); } else { EOE_BNB(0, 2, g0, 2 * C * sizeof(float)); }
    EOE_CHECK_LAUNCH("bn_act_pool_bwd_reduce");
    // + dbeta = sum g, dgamma = sum g*xhat
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(2 * C, 16)), dim3(1024), 0, s, (const float*)(red_scratch + 2 * C), red_scratch, g0,
                       2 * C, dbeta, dgamma, C, accumulate);
    EOE_CHECK_LAUNCH("bn_act_pool_bwd_reduce2");
    if (training && g_bn_sync_fn) {      // (sum g, sum g*xhat, rows) summed over the ranks; dgamma / dbeta above stay this rank's
        hipLaunchKernelGGL(set_float_kernel, dim3(1), dim3(1), 0, s, red_scratch + 2 * C, (float)n * H * W);
        EOE_TRY(eoe_bn_sync_allreduce(red_scratch, 2 * (int64_t)C + 1, 0, stream));
        training = 2;
    }
    if (pool == 1) { EOE_BNB(1, 1, grid, 0); } else { EOE_BNB(1, 2, grid, 0); }
    EOE_CHECK_LAUNCH("bn_act_pool_bwd_apply");
#undef EOE_BNB
    return 0;
}

extern "C" int eoe_colsum_f32(const float* x, float* out, float* red_scratch, int rows, int C, int accumulate, void* stream) {
    EOE_CHECK_ARG(x && out && red_scratch && rows > 0 && C > 0 && C % 4 == 0 && C <= 4096, "colsum_f32: bad args");
    EOE_CHECK_IDX((size_t)rows * C / 4, "colsum_f32");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("colsum_f32", 0, 4.0 * rows * C, stream);
    const int grid = grid_for((size_t)rows * C / 4);
    int g0 = grid > EOE_BN_PARTIALS ? EOE_BN_PARTIALS : grid;
    {   // gridDim.x * 256 must be a multiple of C/4 (one channel quad per thread)
        int cc = C / 4, a = cc, b = 256;
        while (b) { const int t = a % b; a = b; b = t; }
        const int q = cc / a;
        EOE_CHECK_ARG(q <= EOE_BN_PARTIALS, "colsum_f32: C = %d not supported", C);
        g0 = g0 / q * q;
        if (g0 < q) g0 = q;
    }
    hipLaunchKernelGGL(colsum_f32_kernel, dim3(g0), dim3(256), 2 * C * sizeof(float), s, x, red_scratch, rows, C);
    EOE_CHECK_LAUNCH("colsum_f32");
    // partial rows [g0][2C] behind the first 2C floats; the second half of every row is zero and goes to a spare row of the scratch
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(2 * C, 16)), dim3(1024), 0, s, (const float*)(red_scratch + 2 * C), red_scratch, g0,
                       2 * C, out, red_scratch + (size_t)(EOE_BN_PARTIALS + 2) * 2 * C, C, accumulate);
    EOE_CHECK_LAUNCH("colsum_f32_reduce");
    return 0;
}

extern "C" int eoe_bn_act_maxpool_fwd(const void* y, const float* stats, const float* gamma, const float* beta, float* out,
                                      void* out16, uint8_t* idx, int n, int H, int W, int C, int k, int stride, int pad, float slope,
                                      int dtype, void* stream) {
    const int y16 = dtype & EOE_Y16;
    dtype &= ~EOE_Y16;
    EOE_CHECK_ARG(y && stats && out && idx && n > 0 && C % 4 == 0 && k >= 1 && k * k <= 255 && stride >= 1 && pad >= 0 && 2 * pad <= k,
                  "bn_act_maxpool_fwd: bad args");
    EOE_CHECK_ARG((gamma == nullptr) == (beta == nullptr), "bn_act_maxpool: gamma/beta must both be given or both NULL");
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    EOE_CHECK_ARG(Ho >= 1 && Wo >= 1, "bn_act_maxpool_fwd: empty output");
    EOE_CHECK_IDX((size_t)n * H * W * C / 4, "bn_act_maxpool_fwd");
    ProfScope ps("bn_act_maxpool_fwd", 0, (y16 ? 2.0 : 4.0) * n * H * W * C + 7.0 * n * Ho * Wo * C, stream);
    const QuadDecode dec(C / 4, Wo, Ho);
    DISPATCH_TY(dtype, y16, hipLaunchKernelGGL((bn_act_maxpool_fwd_kernel<T, YT>), dim3(grid_for((size_t)n * Ho * Wo * C / 4)), dim3(256), 0,
                                         (hipStream_t)stream, y, stats, gamma, beta, out, (T*)out16, idx, n, H, W, C, k, stride, pad, Ho,
                                         Wo, slope, dec));
    EOE_CHECK_LAUNCH("bn_act_maxpool_fwd");
    return 0;
}

extern "C" int eoe_bn_act_maxpool_bwd(const void* y, const float* stats, const float* gamma, const float* beta, const float* dout,
                                      const uint8_t* idx, float* red_scratch, void* dy, float* dgamma, float* dbeta, int n, int H,
                                      int W, int C, int k, int stride, int pad, int training, float slope, int dtype, void* stream) {
    const int y16 = dtype & EOE_Y16;
    dtype &= ~EOE_Y16;
    EOE_CHECK_ARG(!(y16 && dtype == EOE_F32), "bn_act_maxpool_bwd: EOE_Y16 goes with a 16-bit dtype");
    EOE_CHECK_ARG(y && stats && dout && idx && red_scratch && dy && n > 0 && C % 4 == 0 && C <= 4096 && k >= 1 && stride >= 1 && pad >= 0,
                  "bn_act_maxpool_bwd: bad args");
    EOE_CHECK_ARG((gamma == nullptr) == (beta == nullptr) && (dgamma == nullptr) == (dbeta == nullptr), "bn_act_maxpool_bwd: gamma/beta pairs");
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    hipStream_t s = (hipStream_t)stream;
    EOE_CHECK_IDX((size_t)n * H * W * C / 4, "bn_act_maxpool_bwd");
    ProfScope ps("bn_act_maxpool_bwd", 0, 2 * (y16 ? 2.0 : 4.0) * n * H * W * C + 2.0 * n * H * W * C + 2 * 5.0 * n * Ho * Wo * C, stream);
    const QuadDecode dec(C / 4, W, H);
    const int grid = grid_for((size_t)n * H * W * C / 4);
    int g0 = grid > EOE_BN_PARTIALS ? EOE_BN_PARTIALS : grid;
    {
        int cc = C / 4, a = cc, b = 256;
        while (b) { const int t = a % b; a = b; b = t; }
        const int q = cc / a;
        EOE_CHECK_ARG(q <= EOE_BN_PARTIALS, "bn_act_maxpool_bwd: C = %d not supported", C);
        g0 = g0 / q * q;
        if (g0 < q) g0 = q;
    }
#define EOE_BMP_T(MODE, SS, GRID, LDS)                                                                                        \
    hipLaunchKernelGGL((bn_act_maxpool_bwd_kernel<T, MODE, SS, YT>), dim3(GRID), dim3(256), LDS, s, y, stats, gamma, beta, dout, idx, \
                       red_scratch, (T*)dy, n, H, W, C, k, stride, pad, Ho, Wo, training, slope, dec)
#define EOE_BMP(MODE, SS, GRID, LDS)                                                                                          \
    do {                                                                                                                      \
        if (dtype == EOE_F32) { typedef float T; typedef float YT; EOE_BMP_T(MODE, SS, GRID, LDS); }      /* parity mode: dY in fp32 */ \
        else DISPATCH_TY(dtype, y16, EOE_BMP_T(MODE, SS, GRID, LDS));                                                         \
    } while (0)
    // the stem's geometry (resnet.py:95: MaxPool2d(3, 2, 1) over an even map): one thread per 2x2 pixel block
    const bool s2k3 = k == 3 && stride == 2 && pad == 1 && H % 2 == 0 && W % 2 == 0 && Ho == H / 2 && Wo == W / 2;
    const QuadDecode dec2(C / 4, W / 2, H / 2);
    const int grid2 = grid_for((size_t)n * (H / 2) * (W / 2) * C / 4);
    int g2 = grid2 > EOE_BN_PARTIALS ? EOE_BN_PARTIALS : grid2;
    {
        int cc = C / 4, a = cc, b = 256;
        while (b) { const int t = a % b; a = b; b = t; }
        const int q = cc / a;
        g2 = g2 / q * q;
        if (g2 < q) g2 = q;
    }
#define EOE_BMP2_T(MODE, GRID, LDS)                                                                                           \
    hipLaunchKernelGGL((bn_act_maxpool_bwd_s2k3_kernel<T, MODE, YT>), dim3(GRID), dim3(256), LDS, s, y, stats, gamma, beta, dout, idx, \
                       red_scratch, (T*)dy, n, H, W, C, Ho, Wo, training, slope, dec2)
#define EOE_BMP2(MODE, GRID, LDS)                                                                                             \
    do {                                                                                                                      \
        if (dtype == EOE_F32) { typedef float T; typedef float YT; EOE_BMP2_T(MODE, GRID, LDS); }                             \
        else DISPATCH_TY(dtype, y16, EOE_BMP2_T(MODE, GRID, LDS));                                                            \
    } while (0)
    if (s2k3) { g0 = g2; EOE_BMP2(0, g2, 2 * C * sizeof(float)); }
    else if (stride == 2) { EOE_BMP(0, 2, g0, 2 * C * sizeof(float)); } else if (stride == 1) { EOE_BMP(0, 1, g0, 2 * C * sizeof(float)); }
    else { EOE_BMP(0, 0, g0, 2 * C * sizeof(float)); }
    EOE_CHECK_LAUNCH("bn_act_maxpool_bwd_reduce");
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(2 * C, 16)), dim3(1024), 0, s, (const float*)(red_scratch + 2 * C), red_scratch, g0,
                       2 * C, dbeta, dgamma, C, 0);
    EOE_CHECK_LAUNCH("bn_act_maxpool_bwd_reduce2");
    if (training && g_bn_sync_fn) {
        hipLaunchKernelGGL(set_float_kernel, dim3(1), dim3(1), 0, s, red_scratch + 2 * C, (float)n * H * W);
        EOE_TRY(eoe_bn_sync_allreduce(red_scratch, 2 * (int64_t)C + 1, 0, stream));
        training = 2;
    }
    if (s2k3) { EOE_BMP2(1, grid2, 0); }
    else if (stride == 2) { EOE_BMP(1, 2, grid, 0); } else if (stride == 1) { EOE_BMP(1, 1, grid, 0); } else { EOE_BMP(1, 0, grid, 0); }
#undef EOE_BMP
#undef EOE_BMP_T
#undef EOE_BMP2
#undef EOE_BMP2_T
    EOE_CHECK_LAUNCH("bn_act_maxpool_bwd_apply");
    return 0;
}
