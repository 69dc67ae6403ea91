// HBM-bound kernels of the hot path: casts/transposes of the master weights, patch extraction + normalise,
// token assembly + ln_pre, LayerNorm forward/backward, column sums (bias gradients), objectives, fused Adam.
// One wavefront (64 lanes) per row for the row-wise ops, float4 / 8-byte vector accesses, shuffle reductions.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------ cast/transpose
template <typename T>
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, size_t n) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (; i + 3 < n; i += stride) {
        f32x4 v = *(const f32x4*)(src + i);
        *(u32x2*)(dst + i) = pack4<T>(v[0], v[1], v[2], v[3]);
    }
    // tail (n % 4) handled by the first threads of block 0
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        size_t j = (n & ~(size_t)3) + threadIdx.x;
        dst[j] = (T)src[j];
    }
}

// 64x64 tile transpose through LDS: reads fp32 rows coalesced, writes the straight 16-bit copy and the
// transposed 16-bit copy, both coalesced.
template <typename T>
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, T* __restrict__ dst, T* __restrict__ dst_t,
                                      int rows, int cols) {
    __shared__ float tile[64][65];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 256 threads: 4 rows per pass
    for (int r = ty; r < 64; r += 4) {
        const int gr = r0 + r, gc = c0 + tx;
        float v = 0.f;
        if (gr < rows && gc < cols) {
            v = src[(size_t)gr * cols + gc];
            if (dst) dst[(size_t)gr * cols + gc] = (T)v;
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    if (dst_t) {
        for (int c = ty; c < 64; c += 4) {
            const int gc = c0 + c, gr = r0 + tx;
            if (gc < cols && gr < rows) dst_t[(size_t)gc * rows + gr] = (T)tile[tx][c];
        }
    }
}

// vectorised variant (rows, cols multiples of 4): float4 reads, 8-byte writes on both copies
template <typename T>
__global__ __launch_bounds__(256) void cast_transpose_vec_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                                                 T* __restrict__ dst_t, int rows, int cols) {
    __shared__ T tile[64][68];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int sub = threadIdx.x >> 4, q4 = (threadIdx.x & 15) * 4;
    // the four loads of a thread first, unconditionally (an out-of-range piece re-reads the matrix's first 16 bytes and is zeroed): behind
    // `if (in range)` each load was issued only after the previous piece's store had drained (s_waitcnt vmcnt(0) between them)
    f32x4 v4[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int gr = r0 + p * 16 + sub, gc = c0 + q4;
        const bool ok = gr < rows && gc < cols;
        v4[p] = *(const f32x4*)(src + (ok ? (size_t)gr * cols + gc : (size_t)0));
        if (!ok) v4[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = p * 16 + sub, gr = r0 + r, gc = c0 + q4;
        const f32x4 v = v4[p];
        if (dst && gr < rows && gc < cols) *(u32x2*)(dst + (size_t)gr * cols + gc) = pack4<T>(v[0], v[1], v[2], v[3]);
        tile[r][q4 + 0] = (T)v[0]; tile[r][q4 + 1] = (T)v[1]; tile[r][q4 + 2] = (T)v[2]; tile[r][q4 + 3] = (T)v[3];
    }
    __syncthreads();
    if (dst_t) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int c = p * 16 + sub, gc = c0 + c, gr = r0 + q4;
            if (gc < cols && gr < rows) {
                typename T16<T>::v4 o;
                o[0] = tile[q4 + 0][c]; o[1] = tile[q4 + 1][c]; o[2] = tile[q4 + 2][c]; o[3] = tile[q4 + 3][c];
                *(u32x2*)(dst_t + (size_t)gc * rows + gr) = __builtin_bit_cast(u32x2, o);
            }
        }
    }
}

// the same for a batch of matrices in ONE launch (the 48 weight matrices of a 12-block ViT are each too small to fill the chip:
// 51 launches of 6 us against 0.14 ms of HBM time): workgroup -> (job, 64x64 tile) through the batch's tile prefix sums
constexpr int CT_MAX_JOBS = 56;
struct CtBatch {
    const float* src[CT_MAX_JOBS];
    void* dst[CT_MAX_JOBS];
    void* dst_t[CT_MAX_JOBS];
    int rows[CT_MAX_JOBS], cols[CT_MAX_JOBS];
    int tile_start[CT_MAX_JOBS + 1];
    int count;
};
template <typename T>
__global__ __launch_bounds__(256) void cast_transpose_multi_kernel(CtBatch b) {
    __shared__ T tile[64][68];
    int j = 0;
    while (j + 1 < b.count && (int)blockIdx.x >= b.tile_start[j + 1]) ++j;          // uniform scan over <= 56 entries
    const int t = blockIdx.x - b.tile_start[j];
    const int rows = b.rows[j], cols = b.cols[j], tc = (cols + 63) >> 6;
    const float* __restrict__ src = b.src[j];
    T* __restrict__ dst = (T*)b.dst[j];
    T* __restrict__ dst_t = (T*)b.dst_t[j];
    const int c0 = (t % tc) * 64, r0 = (t / tc) * 64;
    const int sub = threadIdx.x >> 4, q4 = (threadIdx.x & 15) * 4;
    // the four loads of a thread first, unconditionally (an out-of-range piece re-reads the matrix's first 16 bytes and is zeroed): behind
    // `if (in range)` each load was issued only after the previous piece's store had drained (s_waitcnt vmcnt(0) between them)
    f32x4 v4[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int gr = r0 + p * 16 + sub, gc = c0 + q4;
        const bool ok = gr < rows && gc < cols;
        v4[p] = *(const f32x4*)(src + (ok ? (size_t)gr * cols + gc : (size_t)0));
        if (!ok) v4[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = p * 16 + sub, gr = r0 + r, gc = c0 + q4;
        const f32x4 v = v4[p];
        if (dst && gr < rows && gc < cols) *(u32x2*)(dst + (size_t)gr * cols + gc) = pack4<T>(v[0], v[1], v[2], v[3]);
        tile[r][q4 + 0] = (T)v[0]; tile[r][q4 + 1] = (T)v[1]; tile[r][q4 + 2] = (T)v[2]; tile[r][q4 + 3] = (T)v[3];
    }
    __syncthreads();
    if (dst_t) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int c = p * 16 + sub, gc = c0 + c, gr = r0 + q4;
            if (gc < cols && gr < rows) {
                typename T16<T>::v4 o;
                o[0] = tile[q4 + 0][c]; o[1] = tile[q4 + 1][c]; o[2] = tile[q4 + 2][c]; o[3] = tile[q4 + 3][c];
                *(u32x2*)(dst_t + (size_t)gc * rows + gr) = __builtin_bit_cast(u32x2, o);
            }
        }
    }
}

// fp32 [rows, cols] -> 16-bit copy + column sums (bias gradient of the layer whose dY this is), one pass
template <typename T>
__global__ __launch_bounds__(256) void cast_colsum_kernel(const float* __restrict__ x, T* __restrict__ dst,
                                                          float* __restrict__ out, float* __restrict__ part, int rows, int cols,
                                                          int blocked) {
    __shared__ float red[4][256];
    const int cg = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + cg * 4;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < cols) {
        for (int r = blockIdx.y * 4 + rl; r < rows; r += gridDim.y * 4) {
            const f32x4 v = *(const f32x4*)(x + (size_t)r * cols + c);
            *(u32x2*)(dst + (size_t)r * cols + c) = pack4<T>(v[0], v[1], v[2], v[3]);
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] += v[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[rl][cg * 4 + k] = a[k];
    __syncthreads();
    const int cc = blockIdx.x * 256 + threadIdx.x;
    if (cc >= cols) return;
    const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (part) part[eoe_part_index(cc, blockIdx.y, gridDim.y, cols, blocked)] = t;     // this workgroup row's partial sums
    else atomicAdd(out + cc, t);
}

// zero up to EOE_ZERO_MAX small fp32 buffers in one launch (gradient accumulators filled by atomics)
struct ZeroArgs { float* p[EOE_ZERO_MAX]; int n[EOE_ZERO_MAX]; int count; };
__global__ __launch_bounds__(256) void zero_multi_kernel(ZeroArgs a) {
    const int b = blockIdx.y;
    if (b >= a.count) return;
    float* p = a.p[b];
    const int n = a.n[b];
    if ((((uintptr_t)p) & 15) == 0) {             // 16-byte stores over the aligned body (large buffers: the head's 39 MB dx)
        const int n4 = n >> 2;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) ((f32x4*)p)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int i = (n4 << 2) + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0.f;
    } else {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------ patchify
// one block per (image, patch-row py): reads 3 * p rows of res floats (coalesced), writes g patches.
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                const float* __restrict__ stdv, T* __restrict__ out, int res, int p) {
    const int g = res / p;
    const int n = blockIdx.x / g, py = blockIdx.x % g;
    const int kdim = 3 * p * p;
    float mu[3] = {0.f, 0.f, 0.f}, is[3] = {1.f, 1.f, 1.f};
    if (mean) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { mu[c] = mean[c]; is[c] = 1.0f / stdv[c]; }
    }
    // a thread's pieces are 16-byte quads i = t, t + 256, ... of the (channel, kernel row, x) order; (c, ky, xq) of the first one by division,
    // of the following ones by stepping (256 quads = sq rows + rq quads): no division inside the loop.  Four pieces per pass, their loads
    // first (one load -> wait -> store per pass left a single request per thread in flight).
    const int rq_row = res >> 2;                        // quads per image row
    const int nq = 3 * p * rq_row;                      // quads of this patch row
    const int sq = 256 / rq_row, rq = 256 - sq * rq_row;
    int i = threadIdx.x;
    int c = i / (p * rq_row), rem = i - c * (p * rq_row);
    int ky = rem / rq_row, xq = rem - ky * rq_row;
    const float* img = x + (size_t)n * 3 * res * res + (size_t)py * p * res;
    T* orow = out + (size_t)(n * g + py) * g * kdim;
    constexpr int U = 4;
    while (i < nq) {
        f32x4 v[U];
        int cc[U], oo[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = i < nq;
            const int xx = xq * 4;
            v[u] = *(const f32x4*)(img + (ok ? ((size_t)c * res + ky) * res + xx : (size_t)0));
            const int px = xx / p, kx = xx - px * p;
            cc[u] = c;
            oo[u] = ok ? px * kdim + c * p * p + ky * p + kx : -1;
            i += 256;
            ky += sq; xq += rq;
            if (xq >= rq_row) { xq -= rq_row; ++ky; }
            while (ky >= p) { ky -= p; ++c; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (oo[u] < 0) continue;
            const float m = cc[u] == 0 ? mu[0] : (cc[u] == 1 ? mu[1] : mu[2]), sc = cc[u] == 0 ? is[0] : (cc[u] == 1 ? is[1] : is[2]);
            f32x4 w = v[u];
            if (mean) {
#pragma unroll
                for (int r = 0; r < 4; ++r) w[r] = (w[r] - m) * sc;
            }
            *(u32x2*)(orow + oo[u]) = pack4<T>(w[0], w[1], w[2], w[3]);
        }
    }
}

// ------------------------------------------------------------------------------------------ LayerNorm
// one wavefront per row; D <= 64 * 4 * LN_MAXV
constexpr int LN_MAXV = 4;   // float4 per lane -> D <= 1024

struct RowStats { float mean, rstd; };

template <int NV>
__device__ __forceinline__ RowStats row_stats(const f32x4 (&v)[NV], int D, int lane, float eps) {
    constexpr int nv = NV;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = v[i][r] - mean;
                q += d * d;
            }
        }
    const float var = wave_sum(q) / D;
    return {mean, 1.0f / sqrtf(var + eps)};
}

template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, void* __restrict__ y, float* __restrict__ stats,
                                     int rows, int D, float eps, int out_f32) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    constexpr int nv = NV;    // float4 groups per lane (D = 256 * NV)
    f32x4 v[NV], gv[NV], bv[NV];
    const float* xr = x + (size_t)row * ldx;
    // gamma / beta with the row (round 4): loaded where they are used, each pair sat behind s_waitcnt vmcnt(0) -- which also waits for the
    // previous piece's store -- i.e. NV dependent round trips after the reduction
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (i < nv) {
            v[i] = *(const f32x4*)(xr + i * 256 + lane * 4);
            gv[i] = *(const f32x4*)(gamma + i * 256 + lane * 4);
            bv[i] = *(const f32x4*)(beta + i * 256 + lane * 4);
        }
    const RowStats st = row_stats<NV>(v, D, lane, eps);
    if (stats && lane == 0) {
        stats[row * 2] = st.mean;
        stats[row * 2 + 1] = st.rstd;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (i < nv) {
            const int c = i * 256 + lane * 4;
            const f32x4 g = gv[i], b = bv[i];
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (v[i][r] - st.mean) * st.rstd * g[r] + b[r];
            if (out_f32)
                *(f32x4*)((float*)y + (size_t)row * D + c) = (f32x4){o[0], o[1], o[2], o[3]};
            else
                *(u32x2*)((T*)y + (size_t)row * D + c) = pack4<T>(o[0], o[1], o[2], o[3]);
        }
}

// backward: rows are strided over the grid; each wave accumulates dgamma/dbeta partials in registers,
// the block combines its 4 waves through LDS and issues one fp32 atomic per column.
// (round 4: `dy_f32` and `dres` are template parameters.  As run-time branches around the loads they made hipcc emit every conditional load as
//  load -> s_waitcnt vmcnt(0) -> use -- and vmcnt(0) also waits for the row's earlier STORES: a row was ~7 dependent HBM round trips.  Now the
//  row's 3 NV + 1 loads are issued back to back ahead of the first use.)
template <typename T, int NV, bool DYF32, bool DRES>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(NV <= 3 ? 4 : 2, NV <= 3 ? 4 : 2))) void layernorm_bwd_kernel(const void* __restrict__ dy, const float* __restrict__ x, int ldx,
                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                     const float* __restrict__ dres, float* __restrict__ dx_out, int ld_out,
                                     T* __restrict__ dx16, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                     float* __restrict__ dxsum, float* __restrict__ part, int rows, int D) {
    // 8 waves per workgroup (16 waves per CU at 2 workgroups: enough loads in flight for an HBM-bound kernel)
    constexpr int NW = 8;
    extern __shared__ float red_raw[];               // [3][NW][256*NV]
    float (*red)[NW][256 * NV] = (float (*)[NW][256 * NV])red_raw;
    // the wave index as a scalar: the row, and with it every row pointer, lives in SGPRs and a lane's address is one 32-bit offset
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr int nv = NV;
    f32x4 ag[NV], ab[NV], ax[NV], gm[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        ag[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        ab[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        ax[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        gm[i] = *(const f32x4*)(gamma + i * 256 + lane * 4);
    }
    for (int row = blockIdx.x * NW + wave; row < rows; row += gridDim.x * NW) {
        f32x4 df[NV], xv[NV], rv[NV];
        u32x2 dh[NV];
        const f32x2 st = *(const f32x2*)(stats + row * 2);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 256 + lane * 4;
            if (DYF32) df[i] = *(const f32x4*)((const float*)dy + (size_t)row * D + c);
            else dh[i] = *(const u32x2*)((const T*)dy + (size_t)row * D + c);
            xv[i] = *(const f32x4*)(x + (size_t)row * ldx + c);
            if (DRES) rv[i] = *(const f32x4*)(dres + (size_t)row * ld_out + c);
        }
        const float mean = st[0], rstd = st[1];
        // two passes over the row's registers (d and x stay, g = d * gamma and xhat are formed twice: 24 registers less than keeping them, which
        // is what holds the kernel at 128 = two workgroups per CU)
        auto dval = [&](int i) -> f32x4 {
            if (DYF32) return df[i];
            float t[4];
            unpack4<T>(dh[i], t);
            return (f32x4){t[0], t[1], t[2], t[3]};
        };
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const f32x4 d = dval(i);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float xh = (xv[i][r] - mean) * rstd, g = d[r] * gm[i][r];
                ab[i][r] += d[r];
                ag[i][r] += d[r] * xh;
                s1 += g;
                s2 += g * xh;
            }
        }
        s1 = wave_sum(s1) / D;
        s2 = wave_sum(s2) / D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 256 + lane * 4;
            const f32x4 d = dval(i);
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // the two contractions spelled out (g - s1 as ONE fma of d * gamma, then -s2 * xhat into it): which of these hipcc fuses on its
                // own changed with the surrounding code (the round-3 kernel fused both here and 2 of the 12 sums of the first pass), and dx
                // moves by an ulp in a sixth of its elements with it -- enough to move the single-batch AUC of tests/test_gpu_parity_big.py
                // by a few of its 16 384 pairs (6.7e-4 / 1.04e-3 / 7e-4 for three such builds)
                const float xh = (xv[i][r] - mean) * rstd;
                o[r] = rstd * __builtin_fmaf(-s2, xh, __builtin_fmaf(d[r], gm[i][r], -s1));
            }
            if (DRES) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] += rv[i][r];
            }
            *(f32x4*)(dx_out + (size_t)row * ld_out + c) = (f32x4){o[0], o[1], o[2], o[3]};
            if (dx16) *(u32x2*)(dx16 + (size_t)row * D + c) = pack4<T>(o[0], o[1], o[2], o[3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) ax[i][r] += o[r];
        }
    }
    if (!dgamma && !dxsum) return;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (i < nv) {
            // 16-byte stores (four scalar stores per lane at a 16-B lane stride were a 4-way bank conflict each).
            // (round 2: software-pipelining the wave's rows -- next row's loads ahead of this row's reductions -- raised the
            //  kernel to 174 VGPRs = one workgroup per CU and measured the same 0.87 ms / step; reverted)
            *(f32x4*)&red[0][wave][i * 256 + lane * 4] = ag[i];
            *(f32x4*)&red[1][wave][i * 256 + lane * 4] = ab[i];
            *(f32x4*)&red[2][wave][i * 256 + lane * 4] = ax[i];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { a0 += red[0][w][c]; a1 += red[1][w][c]; a2 += red[2][w][c]; }
        if (part) {                                  // this workgroup's partial row (width 3D, blocked layout); a finish kernel adds the rows up
            part[eoe_part_index(c, blockIdx.x, gridDim.x, 3 * D, 1)] = a0;
            part[eoe_part_index(D + c, blockIdx.x, gridDim.x, 3 * D, 1)] = a1;
            part[eoe_part_index(2 * D + c, blockIdx.x, gridDim.x, 3 * D, 1)] = a2;
            continue;
        }
        if (dgamma) {
            atomicAdd(dgamma + c, a0);
            atomicAdd(dbeta + c, a1);
        }
        if (dxsum) atomicAdd(dxsum + c, a2);
    }
}
// all deferred finish reductions of a block in one launch (EoeRedJobs, common.h): a workgroup owns 64 columns (256-B row segments)
// = 16 thread-columns of one float4 x 64 row lanes, so that a thread's <= 8 rows (R <= 512) are ONE batch of independent 16-B loads:
// the partial rows were written hundreds of microseconds earlier and come back from HBM, and with four dependent batches per
// thread this kernel took 23 us (as long as the five launches it replaces).  Fixed summation order; N and seg multiples of 4.
template <typename JOBS>
__global__ __launch_bounds__(1024) void multi_reduce_kernel(JOBS jobs) {
    __shared__ f32x4 l[64][17];
    int j = 0;
    if (sizeof(JOBS) > sizeof(EoeRedJobs)) {          // a tower's table: binary search over the prefix sums
        int lo = 0, hi = jobs.count - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if ((int)blockIdx.x >= jobs.tile_start[mid]) lo = mid; else hi = mid - 1;
        }
        j = lo;
    } else {
        while (j + 1 < jobs.count && (int)blockIdx.x >= jobs.tile_start[j + 1]) ++j;
    }
    const EoeRedJob jb = jobs.job[j];
    const float* __restrict__ part = jb.part;
    const int cq = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const int i = (blockIdx.x - jobs.tile_start[j]) * 64 + cq * 4, n = jb.N, P = jb.R;
    // row p of this thread's column quad: blocked layout = contiguous R x 256 B per 64-column block
    // (round 4) the job's pointer comes out of a by-value struct array: hipcc treats it as a generic pointer (flat loads), and the ragged tail's
    // `row < P ? load : 0` compiled to one load -> wait per row -- R = 160 (the GEMM's column sums) and R = 256 (attention) ran 3 and 4 dependent
    // round trips.  Global address space spelled out; a row past P re-reads the thread's first row and is zeroed: ONE batch of eight loads
    typedef const __attribute__((address_space(1))) f32x4 gf32x4;
    const float* base = jb.blocked ? part + (size_t)(i >> 6) * P * 64 + (i & 63) : part + i;
    const size_t pitch = jb.blocked ? 64 : (size_t)n;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n) {
        for (int p = lane; p < P; p += 512) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = p + 64 * u;
                v[u] = *(gf32x4*)(base + (size_t)(q < P ? q : p) * pitch);
            }
#pragma unroll
            for (int u = 1; u < 8; ++u)
                if (p + 64 * u >= P) v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
    }
    l[lane][cq] = s;
    __syncthreads();
    if (lane >= 8) return;
    // 64 lanes -> 8 -> 1, in a fixed order
    f32x4 t = l[lane][cq];
    for (int k = 1; k < 8; ++k) t += l[lane + 8 * k][cq];
    __syncthreads();
    l[lane][cq] = t;
    __syncthreads();
    if (lane != 0 || i >= n) return;
    for (int k = 1; k < 8; ++k) t += l[k][cq];
    const int which = i / jb.seg, c = i - which * jb.seg;          // seg % 4 == 0: the four columns share a segment
    float* dst = jb.out[which];
    if (!dst) return;
    if (!jobs.overwrite) t += *(const f32x4*)(dst + c);
    *(f32x4*)(dst + c) = t;
}

// ------------------------------------------------------------------------------------------ embed + ln_pre
// one wavefront per token row: x0 = (l == 0 ? cls : tok[n*(L-1) + l-1]) + pos[l];  y = LN(x0)
template <int NV>
__global__ __launch_bounds__(256) void embed_lnpre_fwd_kernel(const float* __restrict__ tok, const float* __restrict__ cls,
                                       const float* __restrict__ pos, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, float* __restrict__ x0,
                                       float* __restrict__ y, float* __restrict__ stats, int n, int L, int D,
                                       float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n * L) return;
    const int img = row / L, l = row % L;
    constexpr int nv = NV;
    const float* src = (l == 0) ? cls : tok + ((size_t)img * (L - 1) + (l - 1)) * D;
    f32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (i < nv) {
            const int c = i * 256 + lane * 4;
            v[i] = *(const f32x4*)(src + c) + *(const f32x4*)(pos + (size_t)l * D + c);
            *(f32x4*)(x0 + (size_t)row * D + c) = v[i];
        }
    const RowStats st = row_stats<NV>(v, D, lane, eps);
    if (lane == 0) {
        stats[row * 2] = st.mean;
        stats[row * 2 + 1] = st.rstd;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (i < nv) {
            const int c = i * 256 + lane * 4;
            const f32x4 g = *(const f32x4*)(gamma + c), b = *(const f32x4*)(beta + c);
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (v[i][r] - st.mean) * st.rstd * g[r] + b[r];
            *(f32x4*)(y + (size_t)row * D + c) = o;
        }
}

// backward: one block per token position l (grid = L), waves stride over the images; dpos[l] and (l == 0)
// dcls are complete sums over the batch -> plain stores/adds without atomics; dgamma/dbeta via atomics.
template <typename T, int NV>
__global__ __launch_bounds__(256) void embed_lnpre_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x0,
                                       const float* __restrict__ stats, const float* __restrict__ gamma,
                                       T* __restrict__ dtok, float* __restrict__ dcls, float* __restrict__ dpos,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ part, int n, int L,
                                       int D) {
    __shared__ float red[3][4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l = blockIdx.x;
    constexpr int nv = NV;
    f32x4 ag[NV], ab[NV], ap[NV], gm[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        ag[i] = ab[i] = ap[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (i < nv) gm[i] = *(const f32x4*)(gamma + i * 256 + lane * 4);
    }
    for (int img = wave; img < n; img += 4) {
        const size_t row = (size_t)img * L + l;
        const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
        f32x4 g[NV], xh[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (i < nv) {
                const int c = i * 256 + lane * 4;
                const f32x4 d = *(const f32x4*)(dy + row * D + c);
                const f32x4 xv = *(const f32x4*)(x0 + row * D + c);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    xh[i][r] = (xv[r] - mean) * rstd;
                    ab[i][r] += d[r];
                    ag[i][r] += d[r] * xh[i][r];
                    g[i][r] = d[r] * gm[i][r];
                    s1 += g[i][r];
                    s2 += g[i][r] * xh[i][r];
                }
            }
        s1 = wave_sum(s1) / D;
        s2 = wave_sum(s2) / D;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (i < nv) {
                const int c = i * 256 + lane * 4;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o[r] = rstd * (g[i][r] - s1 - xh[i][r] * s2);
                    ap[i][r] += o[r];
                }
                if (l > 0)
                    *(u32x2*)(dtok + ((size_t)img * (L - 1) + (l - 1)) * D + c) = pack4<T>(o[0], o[1], o[2], o[3]);
            }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (i < nv) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                red[0][wave][i * 256 + lane * 4 + r] = ag[i][r];
                red[1][wave][i * 256 + lane * 4 + r] = ab[i][r];
                red[2][wave][i * 256 + lane * 4 + r] = ap[i][r];
            }
        }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        const float sg = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
        const float sb = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
        const float sp = red[2][0][c] + red[2][1][c] + red[2][2][c] + red[2][3][c];
        if (part) {                                  // this token position's partial row (width 2D, blocked layout); a finish kernel adds the L rows up
            part[eoe_part_index(c, l, L, 2 * D, 1)] = sg;
            part[eoe_part_index(D + c, l, L, 2 * D, 1)] = sb;
        } else {
            atomicAdd(dgamma + c, sg);
            atomicAdd(dbeta + c, sb);
        }
        dpos[(size_t)l * D + c] += sp;
        if (l == 0) dcls[c] += sp;
    }
}

// ------------------------------------------------------------------------------------------ column sums
// block = 256 threads = cpb column-groups (4 cols = 8 B each; cpb = min(64, cols/4) rounded up to a power of two, so narrow
// matrices -- the conv-bias gradients of CNN32, 32..128 columns -- still use every thread) x 256/cpb row lanes; grid.x over
// column groups, grid.y strides rows; combine through LDS then one atomic per column per block.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int ldx, float* __restrict__ out, int rows, int cols,
                                                     int cpb) {
    __shared__ f32x4 red[256];
    const int rpb = 256 / cpb;
    const int tc = threadIdx.x % cpb, rl = threadIdx.x / cpb;
    const int c = (blockIdx.x * cpb + tc) * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (c < cols) {
        for (int r = blockIdx.y * rpb + rl; r < rows; r += gridDim.y * rpb) {
            float t[4];
            unpack4<T>(*(const u32x2*)(x + (size_t)r * ldx + c), t);
            a += (f32x4){t[0], t[1], t[2], t[3]};
        }
    }
    red[threadIdx.x] = a;
    __syncthreads();
    if (rl == 0) {
        for (int k = 1; k < rpb; ++k) a += red[threadIdx.x + k * cpb];
        red[tc] = a;                              // row 0 of the LDS image: only this thread read / writes slot tc
    }
    __syncthreads();
    // one atomic per column, consecutive lanes on consecutive addresses (a 4-atomics-per-lane epilogue measured 50 % slower)
    const int cc = blockIdx.x * cpb * 4 + threadIdx.x;
    if ((int)threadIdx.x < cpb * 4 && cc < cols) atomicAdd(out + cc, ((const float*)red)[threadIdx.x]);
}

// deterministic variant: workgroup (bx, by) writes its partial column sums to row by of part[gridDim.y][cols]; a second
// kernel adds the rows (no atomics, no memset: bitwise reproducible bias gradients, and nothing but kernels in a captured graph)
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, int ldx, float* __restrict__ part, int rows, int cols,
                                                             int cpb) {
    __shared__ f32x4 red[256];
    const int rpb = 256 / cpb;
    const int tc = threadIdx.x % cpb, rl = threadIdx.x / cpb;
    const int c = (blockIdx.x * cpb + tc) * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (c < cols) {
        for (int r = blockIdx.y * rpb + rl; r < rows; r += gridDim.y * rpb) {
            float t[4];
            unpack4<T>(*(const u32x2*)(x + (size_t)r * ldx + c), t);
            a += (f32x4){t[0], t[1], t[2], t[3]};
        }
    }
    red[threadIdx.x] = a;
    __syncthreads();
    if (rl == 0 && c < cols) {
        for (int k = 1; k < rpb; ++k) a += red[threadIdx.x + k * cpb];
        *(f32x4*)(part + (size_t)blockIdx.y * cols + c) = a;
    }
}
__global__ __launch_bounds__(1024) void colsum_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int P, int n) {
    __shared__ float l[1024];
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), lane = threadIdx.x >> 6;
    float s = 0.f;
    if (i < n)
        for (int p = lane; p < P; p += 16) s += part[(size_t)p * n + i];
    l[threadIdx.x] = s;
    __syncthreads();
    if (lane == 0 && i < n) {
        for (int k = 1; k < 16; ++k) s += l[threadIdx.x + 64 * k];
        out[i] = s;
    }
}

// ------------------------------------------------------------------------------------------ objectives
// HSC rows: one wavefront per sample
__global__ __launch_bounds__(256) void hsc_rows_kernel(const float* __restrict__ f, const int64_t* __restrict__ labels, int64_t nominal,
                                float* __restrict__ scores, float* __restrict__ dists, float* __restrict__ losses,
                                int n, int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    float ss = 0.f;
    for (int c = lane; c < d; c += 64) {
        const float v = f[(size_t)row * d + c];
        ss += v * v;
    }
    ss = wave_sum(ss);
    if (lane == 0) {
        const float nrm = sqrtf(ss);                     // hsc.py:13  norm(f)**2 + 1
        const float dist = sqrtf(nrm * nrm + 1.0f) - 1.0f;
        const float sc = 1.0f - expf(-dist);
        if (scores) scores[row] = sc;
        if (dists) dists[row] = dist;
        if (losses) losses[row] = (labels[row] == nominal) ? dist : -logf(sc + 1e-9f);
    }
}

// deterministic fixed-order sum of n floats by one block -> out[0] = scale * sum
__global__ __launch_bounds__(256) void sum_scale_kernel(const float* __restrict__ v, float* __restrict__ out, int n, float scale) {
    __shared__ float red[16];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) a += v[i];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
        out[0] = s * scale;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void hsc_bwd_kernel(const float* __restrict__ f, const int64_t* __restrict__ labels, int64_t nominal,
                               const float* __restrict__ gscale, float* __restrict__ df, T* __restrict__ df16, int n,
                               int d, float inv_count) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    float ss = 0.f;
    for (int c = lane; c < d; c += 64) {
        const float v = f[(size_t)row * d + c];
        ss += v * v;
    }
    ss = wave_sum(ss);
    const float root = sqrtf(ss + 1.0f);
    const float dist = root - 1.0f;
    const float e = expf(-dist);
    const float dl_dd = (labels[row] == nominal) ? 1.0f : -e / (1.0f - e + 1e-9f);
    const float coef = dl_dd / root * inv_count * (gscale ? gscale[0] : 1.0f);
    for (int c = lane; c < d; c += 64) {
        const float g = f[(size_t)row * d + c] * coef;
        if (df) df[(size_t)row * d + c] = g;
        if (df16) df16[(size_t)row * d + c] = (T)g;
    }
}

// the other row objectives of the TRAINER registry, one wavefront per sample:
//   KIND 1 = DSAD  (dsad.py:17-21):  d = |f|^2,        loss = d if nominal else 1 / (d + 1e-9)
//   KIND 2 = DSVDD (dsvdd.py:24-27): d = |f - c|^2,    loss = score = d for every sample
template <int KIND>
__global__ __launch_bounds__(256) void rowobj_fwd_kernel(const float* __restrict__ f, const float* __restrict__ center,
                                                         const int64_t* __restrict__ labels, int64_t nominal,
                                                         float* __restrict__ losses, int n, int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    float ss = 0.f;
    for (int c = lane; c < d; c += 64) {
        float v = f[(size_t)row * d + c];
        if (KIND == 2) v -= center[c];
        ss += v * v;
    }
    ss = wave_sum(ss);
    if (lane == 0) losses[row] = (KIND == 2 || labels[row] == nominal) ? ss : 1.0f / (ss + 1e-9f);
}
template <int KIND>
__global__ __launch_bounds__(256) void rowobj_bwd_kernel(const float* __restrict__ f, const float* __restrict__ center,
                                                         const int64_t* __restrict__ labels, int64_t nominal,
                                                         const float* __restrict__ gscale, float* __restrict__ df, int n, int d,
                                                         float inv_count) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    float coef = 2.0f;
    if (KIND == 1 && labels[row] != nominal) {
        float ss = 0.f;
        for (int c = lane; c < d; c += 64) {
            const float v = f[(size_t)row * d + c];
            ss += v * v;
        }
        ss = wave_sum(ss);
        const float t = ss + 1e-9f;
        coef = -2.0f / (t * t);
    }
    coef *= inv_count * (gscale ? gscale[0] : 1.0f);
    for (int c = lane; c < d; c += 64) {
        float v = f[(size_t)row * d + c];
        if (KIND == 2) v -= center[c];
        df[(size_t)row * d + c] = v * coef;
    }
}

// ------------------------------------------------------------------------------------------ rank metrics on the device
// AUC (tie-averaged rank statistic = trapezoidal ROC area, ad_trainer.py:452-455,517-519) and average precision (:520-521) from
// EXACT integer pair counts -- no sort: for every positive i
//     wins2_i = 2 #{j negative: s_j < s_i} + #{j negative: s_j == s_i}            AUC = sum_i wins2_i / (2 P N)
//     prec_i  = #{j positive: s_j >= s_i} / #{j: s_j >= s_i}                      AP  = sum_i prec_i / P
// (precision at a positive's own threshold; tied positives share it, which is sklearn's step-wise sum over distinct thresholds).
// Thread i walks all j through LDS tiles: n^2 compares (n = 10^4 test scores: 10^8), partial sums per workgroup, fixed-order finish.
__global__ __launch_bounds__(256) void rank_pairs_kernel(const float* __restrict__ s, const int64_t* __restrict__ labels, int64_t positive,
                                                         unsigned long long* __restrict__ part_wins, double* __restrict__ part_prec,
                                                         unsigned long long* __restrict__ part_pos, int n) {
    __shared__ float ts[256];
    __shared__ int tp[256];
    __shared__ unsigned long long rw[256];
    __shared__ double rp[256];
    __shared__ unsigned long long rc[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n;
    const float si = live ? s[i] : 0.f;
    const bool pi = live && labels[i] == positive;
    unsigned lt_neg = 0, eq_neg = 0, ge_pos = 0, ge_all = 0;
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int j = j0 + threadIdx.x;
        ts[threadIdx.x] = j < n ? s[j] : 0.f;
        tp[threadIdx.x] = j < n ? (labels[j] == positive ? 1 : 0) : -1;
        __syncthreads();
        const int m = n - j0 < 256 ? n - j0 : 256;
        for (int k = 0; k < m; ++k) {
            const float sj = ts[k];
            const int pj = tp[k];
            lt_neg += (pj == 0 && sj < si);
            eq_neg += (pj == 0 && sj == si);
            ge_pos += (pj == 1 && sj >= si);
            ge_all += (sj >= si);
        }
        __syncthreads();
    }
    rw[threadIdx.x] = pi ? 2ull * lt_neg + eq_neg : 0ull;
    rp[threadIdx.x] = pi ? (double)ge_pos / (double)ge_all : 0.0;
    rc[threadIdx.x] = pi ? 1ull : 0ull;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long w = 0, c = 0;
        double p = 0.0;
        for (int k = 0; k < 256; ++k) { w += rw[k]; p += rp[k]; c += rc[k]; }
        part_wins[blockIdx.x] = w;
        part_prec[blockIdx.x] = p;
        part_pos[blockIdx.x] = c;
    }
}
__global__ void rank_finish_kernel(const unsigned long long* __restrict__ part_wins, const double* __restrict__ part_prec,
                                   const unsigned long long* __restrict__ part_pos, int nb, int n, double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned long long w = 0, P = 0;
    double p = 0.0;
    for (int b = 0; b < nb; ++b) { w += part_wins[b]; p += part_prec[b]; P += part_pos[b]; }
    const unsigned long long N = (unsigned long long)n - P;
    const double nan = __builtin_nan("");
    out[0] = (P == 0 || N == 0) ? nan : (double)w / (2.0 * (double)P * (double)N);
    out[1] = P == 0 ? nan : p / (double)P;
}

// CLIP text-prompt objective (training/clip.py:66-103): one wavefront per sample.  l_j = 100 * <f / |f|, t_j>, j < T <= 64;
//   loss_i = -(log_softmax l)[pick],  pick = T-1 (anomalous), 0 (nominal, one_vs_rest) or argmax_{j < T-1} (nominal,
//   leave_one_out); samples whose label is neither contribute 0 (clip.py:89-91);  score_i = softmax(l)[T-1].
// The logits live one per lane (lane j holds l_j after the T wave reductions).
__device__ __forceinline__ float clip_logits(const float* __restrict__ f, const float* __restrict__ text, int row, int d, int T,
                                             int lane, float& inv_norm) {
    float ss = 0.f;
    for (int c = lane; c < d; c += 64) { const float v = f[(size_t)row * d + c]; ss += v * v; }
    inv_norm = 1.0f / sqrtf(wave_sum(ss));
    float mine = -INFINITY;
    for (int j = 0; j < T; ++j) {
        float dot = 0.f;
        for (int c = lane; c < d; c += 64) dot += f[(size_t)row * d + c] * text[(size_t)j * d + c];
        dot = wave_sum(dot);
        if (lane == j) mine = 100.0f * dot * inv_norm;
    }
    return mine;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = fmaxf(v, __shfl_xor(v, s, 64));
    return v;
}
// lane j: p = softmax(l)_j, returns the picked column (uniform); lse through the reference argument
__device__ __forceinline__ int clip_pick(float l, int lane, int T, int64_t label, int64_t nominal, int leave_one_out, float& p, float& lse) {
    const float m = wave_max(l);
    const float e = lane < T ? __expf(l - m) : 0.f;
    const float se = wave_sum(e);
    p = e / se;
    lse = m + __logf(se);
    if (label == 1 - nominal) return T - 1;
    if (label != nominal) return -1;
    if (!leave_one_out) return 0;
    // first maximum over j < T-1 (torch.max returns the first index among equal maxima on CPU; ties are measure-zero here)
    const float cand = lane < T - 1 ? l : -INFINITY;
    const float mx = wave_max(cand);
    const unsigned long long hit = __ballot(cand == mx && lane < T - 1);
    return __ffsll((long long)hit) - 1;
}
__global__ __launch_bounds__(256) void clip_rows_kernel(const float* __restrict__ f, const float* __restrict__ text,
                                                        const int64_t* __restrict__ labels, int64_t nominal, int leave_one_out,
                                                        float* __restrict__ scores, float* __restrict__ losses, int n, int d, int T) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    float inv_norm;
    const float l = clip_logits(f, text, row, d, T, lane, inv_norm);
    float p, lse;
    const int pick = clip_pick(l, lane, T, labels ? labels[row] : nominal, nominal, leave_one_out, p, lse);
    if (scores && lane == T - 1) scores[row] = p;
    if (losses) {
        const float lp = __shfl(l, pick < 0 ? 0 : pick, 64);
        if (lane == 0) losses[row] = pick < 0 ? 0.f : lse - lp;
    }
}
// df = coef * (100 * sum_j g_j t_j - fhat * sum_j g_j l_j) / |f|,  g_j = softmax_j - [j == pick]
__global__ __launch_bounds__(256) void clip_bwd_kernel(const float* __restrict__ f, const float* __restrict__ text,
                                                       const int64_t* __restrict__ labels, int64_t nominal, int leave_one_out,
                                                       const float* __restrict__ gscale, float* __restrict__ df, int n, int d, int T,
                                                       float inv_count) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    float inv_norm;
    const float l = clip_logits(f, text, row, d, T, lane, inv_norm);
    float p, lse;
    const int pick = clip_pick(l, lane, T, labels[row], nominal, leave_one_out, p, lse);
    const float coef = inv_count * (gscale ? gscale[0] : 1.0f);
    float g = (pick < 0 || lane >= T) ? 0.f : (p - (lane == pick ? 1.f : 0.f));
    const float gl = wave_sum(lane < T ? g * l : 0.f);
    for (int c = lane; c < d; c += 64) {
        float acc = 0.f;
        for (int j = 0; j < T; ++j) acc += __shfl(g, j, 64) * text[(size_t)j * d + c];
        const float fh = f[(size_t)row * d + c] * inv_norm;
        df[(size_t)row * d + c] = coef * (100.0f * acc - fh * gl) * inv_norm;
    }
}

// focal loss on logits (focal.py:11-24): b = bce(x, y), pt = clamp(exp(-b), eps, 1 - eps), loss = (1 - pt)^gamma * b
__global__ __launch_bounds__(256) void focal_rows_kernel(const float* __restrict__ x, const int64_t* __restrict__ labels, int64_t nominal,
                                                         float* __restrict__ scores, float* __restrict__ losses, int n, float gamma,
                                                         float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i], y = (float)labels[i];
    const float s = 1.0f / (1.0f + expf(-v));
    if (scores) scores[i] = (nominal == 0) ? s : 1.0f - s;
    if (losses) {
        const float b = fmaxf(v, 0.f) - v * y + log1pf(expf(-fabsf(v)));
        const float pt = fminf(fmaxf(expf(-b), eps), 1.0f - eps);
        losses[i] = powf(1.0f - pt, gamma) * b;
    }
}
__global__ __launch_bounds__(256) void focal_bwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ labels,
                                                        const float* __restrict__ gscale, float* __restrict__ dx, int n,
                                                        float inv_count, float gamma, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i], y = (float)labels[i];
    const float b = fmaxf(v, 0.f) - v * y + log1pf(expf(-fabsf(v)));
    const float db = 1.0f / (1.0f + expf(-v)) - y;
    const float raw = expf(-b);
    const float pt = fminf(fmaxf(raw, eps), 1.0f - eps);
    float g = powf(1.0f - pt, gamma) * db;
    if (raw >= eps && raw <= 1.0f - eps) g += gamma * powf(1.0f - pt, gamma - 1.0f) * pt * db * b;   // the clamp's derivative is 0 outside
    dx[i] = g * inv_count * (gscale ? gscale[0] : 1.0f);
}

__global__ __launch_bounds__(256) void bce_rows_kernel(const float* __restrict__ x, const int64_t* __restrict__ labels, int64_t nominal,
                                float* __restrict__ scores, float* __restrict__ losses, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i], y = (float)labels[i];
    const float s = 1.0f / (1.0f + expf(-v));
    if (scores) scores[i] = (nominal == 0) ? s : 1.0f - s;
    if (losses) losses[i] = fmaxf(v, 0.f) - v * y + log1pf(expf(-fabsf(v)));
}

__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ labels,
                               const float* __restrict__ gscale, float* __restrict__ dx, int n, float inv_count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s = 1.0f / (1.0f + expf(-x[i]));
    dx[i] = (s - (float)labels[i]) * inv_count * (gscale ? gscale[0] : 1.0f);
}

// ------------------------------------------------------------------------------------------ narrow linear head
// y[M,N] = x[M,K] w[N,K]^T + b for N <= 8 (the 1-wide classification head of CustomNet(clf=True),
// custom_base.py:25-26): exact fp32, one wavefront per row.  Latency-bound by construction.
constexpr int SMALL_N = 8;
__global__ __launch_bounds__(256) void linear_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               int M, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float acc[SMALL_N];
#pragma unroll
    for (int n = 0; n < SMALL_N; ++n) acc[n] = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float xv = x[(size_t)row * K + k];
#pragma unroll
        for (int n = 0; n < SMALL_N; ++n)
            if (n < N) acc[n] += xv * w[(size_t)n * K + k];
    }
#pragma unroll
    for (int n = 0; n < SMALL_N; ++n)
        if (n < N) {
            const float t = wave_sum(acc[n]);
            if (lane == 0) y[(size_t)row * N + n] = t + (bias ? bias[n] : 0.f);
        }
}

__global__ __launch_bounds__(256) void linear_small_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                              float* __restrict__ dx, int M, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float d[SMALL_N];
#pragma unroll
    for (int n = 0; n < SMALL_N; ++n) d[n] = (n < N) ? dy[(size_t)row * N + n] : 0.f;
    for (int k = lane; k < K; k += 64) {
        float a = 0.f;
#pragma unroll
        for (int n = 0; n < SMALL_N; ++n)
            if (n < N) a += d[n] * w[(size_t)n * K + k];
        dx[(size_t)row * K + k] = a;
    }
}

// dw[n,k] += sum_m dy[m,n] x[m,k] (thread per k, rows strided over grid.y), db[n] += sum_m dy[m,n]
__global__ __launch_bounds__(256) void linear_small_dw_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ dw, float* __restrict__ db, int M, int N,
                                                              int K) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    float acc[SMALL_N], accb[SMALL_N];
#pragma unroll
    for (int n = 0; n < SMALL_N; ++n) { acc[n] = 0.f; accb[n] = 0.f; }
    for (int m = blockIdx.y; m < M; m += gridDim.y) {
        const float xv = (k < K) ? x[(size_t)m * K + k] : 0.f;
#pragma unroll
        for (int n = 0; n < SMALL_N; ++n)
            if (n < N) {
                const float d = dy[(size_t)m * N + n];
                acc[n] += d * xv;
                accb[n] += d;
            }
    }
    if (k < K) {
#pragma unroll
        for (int n = 0; n < SMALL_N; ++n)
            if (n < N) atomicAdd(dw + (size_t)n * K + k, acc[n]);
    }
    if (db && k == 0) {
#pragma unroll
        for (int n = 0; n < SMALL_N; ++n)
            if (n < N) atomicAdd(db + n, accb[n]);
    }
}

// ------------------------------------------------------------------------------------------ Adam
// fused multi-tensor SGD with momentum / Nesterov (torch.optim.SGD, dampening 0; ad_trainer.py:380-381): g += wd * p;
// buf = momentum * buf + g (buf starts at 0, which reproduces torch's "first step: buf = g"); p -= lr * (nesterov ? g + momentum * buf : buf).
// Same chunk tables as the Adam kernel (m_off = the momentum buffer, v_off unused).
// Chunks flagged EOE_CHUNK_FP16 (the "fp16-weights mode", clip/model.py:371-392 convert_weights: CLIP's convolution / linear / attention /
// projection parameters are fp16 tensors on a GPU and torch.optim.SGD updates them in fp16): the fp32 storage holds fp16-representable
// values and every elementwise op of torch's update rounds its result to fp16, op by op as torch's kernels do (fp32 arithmetic inside an
// op, one rounding at its end): g = r(g + wd p); buf = r(momentum buf); buf = r(buf + g); g = r(g + momentum buf); p = r(p - lr g).
// round an fp32 op result to fp16 (nearest even), as torch stores the result of an op on half tensors.  The empty asm keeps the fp32 value
// materialised: without it the compiler folds fma + conversion into v_fma_mixlo_f16 -- ONE rounding from the exact sum, where torch's
// kernels round to fp32 first (a sum that lands on an fp16 tie in fp32 then goes to even: 1.2 % of the elements differed per step)
__device__ __forceinline__ float r16(float x) {
    asm volatile("" : "+v"(x));
    return (float)(_Float16)x;
}
__global__ __launch_bounds__(256) void sgd_multi_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                        const eoe_adam_chunk* __restrict__ chunks, float lr, float momentum, float wd,
                                                        int nesterov, float grad_scale_inv, const int* __restrict__ skip) {
    if (skip && *skip) return;                      // a non-finite gradient was found this step (eoe_grads_nonfinite): no update at all
    const eoe_adam_chunk ck = chunks[blockIdx.x];
    float* pp = p + ck.p_off;
    const float* gg = g + ck.g_off;
    float* bb = buf + ck.m_off;
    if (ck.group & EOE_CHUNK_FP16) {
        for (int i = threadIdx.x; i < ck.n; i += blockDim.x) {
            const float pv = pp[i];
            float gv = r16(gg[i] * grad_scale_inv);                                  // the fp16 gradient autograd hands torch's optimiser
            if (wd != 0.f) gv = r16(__fmaf_rn(wd, pv, gv));                          // grad.add(param, alpha=wd): a + alpha * b in fp32, contracted
            float step = gv;
            if (momentum != 0.f) {
                float bv = r16(__fmul_rn(bb[i], momentum));                          // buf.mul_(momentum)
                bv = r16(__fadd_rn(bv, gv));                                         // buf.add_(grad, alpha=1 - dampening)
                bb[i] = bv;
                step = nesterov ? r16(__fmaf_rn(momentum, bv, gv)) : bv;             // grad.add(buf, alpha=momentum)
            }
            pp[i] = r16(__fmaf_rn(-lr, step, pv));                                   // param.add_(grad, alpha=-lr)
        }
        return;
    }
    for (int i = threadIdx.x; i < ck.n; i += blockDim.x) {
        const float pv = pp[i];
        float gv = gg[i] * grad_scale_inv;
        if (wd != 0.f) gv = gv + wd * pv;
        float step = gv;
        if (momentum != 0.f) {
            const float bv = momentum * bb[i] + gv;
            bb[i] = bv;
            step = nesterov ? gv + momentum * bv : bv;
        }
        pp[i] = pv - lr * step;
    }
}

// one element of torch.optim.Adam's update (shared by the chunk and the tile kernel: the same expression, the same contractions)
struct AdamK { float step_size, bc2_sqrt, ginv, beta1, beta2, eps, wd; };
__device__ __forceinline__ float adam_upd(const AdamK& k, float pv, float gv, float& mv, float& vvv) {
    gv *= k.ginv;
    if (k.wd != 0.f) gv = gv + k.wd * pv;
    mv = mv + (gv - mv) * (1.0f - k.beta1);
    vvv = vvv * k.beta2 + (1.0f - k.beta2) * gv * gv;
    const float denom = sqrtf(vvv) / k.bc2_sqrt + k.eps;
    return pv - k.step_size * (mv / denom);
}

// eoe_adam_tiles: one workgroup per 64 x 64 tile of a 2-D weight.  Thread (sub = t >> 4, q4 = 4 (t & 15)) owns the four 16-byte pieces
// (row 16 i + sub, columns q4 .. q4 + 3): 16 loads in flight (p, g, m, v), the update, p / m / v and the [rows, cols] 16-bit piece stored
// from the registers, the transposed copy through a padded LDS tile (as cast_transpose_multi_kernel).
template <typename T>
__global__ __launch_bounds__(256) void adam_tiles_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, const eoe_adam_tile* __restrict__ tiles, eoe_adam_scalars sc,
                                                         float beta1, float beta2, float eps, float wd, const int* __restrict__ skip) {
    if (skip && *skip) return;                      // (the copies then stay those of the unchanged weight)
    __shared__ T tile[64][68];
    const eoe_adam_tile tl = tiles[blockIdx.x];
    const AdamK k = {sc.step_size[tl.group & (EOE_ADAM_GROUPS - 1)], sc.bc2_sqrt[tl.group & (EOE_ADAM_GROUPS - 1)], sc.grad_scale_inv,
                     beta1, beta2, eps, wd};
    float* pp = p + tl.p_off;
    const float* gg = g + tl.g_off;
    float* mm = m + tl.m_off;
    float* vv = v + tl.v_off;
    T* dst = (T*)tl.d16;
    T* dst_t = (T*)tl.d16_t;
    const int rows = tl.rows, cols = tl.cols, tc = (cols + 63) >> 6;
    const int c0 = (tl.tile % tc) * 64, r0 = (tl.tile / tc) * 64;
    const int sub = threadIdx.x >> 4, q4 = (threadIdx.x & 15) * 4;
    f32x4 pv[4], gv[4], mv[4], vx[4];
    size_t at[4];
    bool ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {                   // (a piece outside the matrix re-reads the matrix's first 16 bytes and is dropped)
        const int gr = r0 + i * 16 + sub, gc = c0 + q4;
        ok[i] = gr < rows && gc < cols;
        at[i] = ok[i] ? (size_t)gr * cols + gc : (size_t)0;
        pv[i] = *(const f32x4*)(pp + at[i]);
        gv[i] = *(const f32x4*)(gg + at[i]);
        mv[i] = *(const f32x4*)(mm + at[i]);
        vx[i] = *(const f32x4*)(vv + at[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = mv[i][r], b = vx[i][r];
            pv[i][r] = adam_upd(k, pv[i][r], gv[i][r], a, b);
            mv[i][r] = a;
            vx[i][r] = b;
        }
        const int rr = i * 16 + sub;
        if (ok[i]) {
            *(f32x4*)(pp + at[i]) = pv[i];
            *(f32x4*)(mm + at[i]) = mv[i];
            *(f32x4*)(vv + at[i]) = vx[i];
            if (dst) *(u32x2*)(dst + at[i]) = pack4<T>(pv[i][0], pv[i][1], pv[i][2], pv[i][3]);
        }
        tile[rr][q4 + 0] = (T)pv[i][0]; tile[rr][q4 + 1] = (T)pv[i][1]; tile[rr][q4 + 2] = (T)pv[i][2]; tile[rr][q4 + 3] = (T)pv[i][3];
    }
    if (!dst_t) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = i * 16 + sub, gc = c0 + c, gr = r0 + q4;
        if (gc < cols && gr < rows) {
            typename T16<T>::v4 o;
            o[0] = tile[q4 + 0][c]; o[1] = tile[q4 + 1][c]; o[2] = tile[q4 + 2][c]; o[3] = tile[q4 + 3][c];
            *(u32x2*)(dst_t + (size_t)gc * rows + gr) = __builtin_bit_cast(u32x2, o);
        }
    }
}

// one block per chunk of <= EOE_ADAM_CHUNK elements; float4 accesses (chunk offsets are multiples of 4 for
// 16-B aligned parameter starts; scalar path otherwise).
template <typename T>
__global__ __launch_bounds__(256) void adam_multi_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                  float* __restrict__ v, const eoe_adam_chunk* __restrict__ chunks,
                                  eoe_adam_scalars sc, float beta1, float beta2, float eps, float wd,
                                  T* __restrict__ shadow, const int* __restrict__ skip) {
    if (skip && *skip) return;                      // a non-finite gradient was found this step (eoe_grads_nonfinite): no update at all
    const eoe_adam_chunk ck = chunks[blockIdx.x];
    const float step_size = sc.step_size[ck.group & (EOE_ADAM_GROUPS - 1)];
    const float bc2_sqrt = sc.bc2_sqrt[ck.group & (EOE_ADAM_GROUPS - 1)];
    float* pp = p + ck.p_off;
    const float* gg = g + ck.g_off;
    float* mm = m + ck.m_off;
    float* vv = v + ck.v_off;
    T* sh = shadow ? shadow + ck.p_off : nullptr;
    const bool aligned = (((ck.p_off | ck.g_off | ck.m_off | ck.v_off) & 3) == 0);
    // (gradients arrive multiplied by the loss scale, a power of two: the division back is exact)
    const AdamK k = {step_size, bc2_sqrt, sc.grad_scale_inv, beta1, beta2, eps, wd};
    auto upd = [&](float pv, float gv, float& mv, float& vvv) -> float { return adam_upd(k, pv, gv, mv, vvv); };
    if (aligned) {
        const int n4 = ck.n >> 2;
        for (int i = threadIdx.x; i < n4; i += blockDim.x) {
            f32x4 pv = *(f32x4*)(pp + i * 4), gv = *(const f32x4*)(gg + i * 4);
            f32x4 mv = *(f32x4*)(mm + i * 4), vx = *(f32x4*)(vv + i * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = mv[r], b = vx[r];
                pv[r] = upd(pv[r], gv[r], a, b);
                mv[r] = a;
                vx[r] = b;
            }
            *(f32x4*)(pp + i * 4) = pv;
            *(f32x4*)(mm + i * 4) = mv;
            *(f32x4*)(vv + i * 4) = vx;
            if (sh) *(u32x2*)(sh + i * 4) = pack4<T>(pv[0], pv[1], pv[2], pv[3]);
        }
        for (int i = (n4 << 2) + threadIdx.x; i < ck.n; i += blockDim.x) {
            float a = mm[i], b = vv[i];
            const float np = upd(pp[i], gg[i], a, b);
            pp[i] = np; mm[i] = a; vv[i] = b;
            if (sh) sh[i] = (T)np;
        }
    } else {
        for (int i = threadIdx.x; i < ck.n; i += blockDim.x) {
            float a = mm[i], b = vv[i];
            const float np = upd(pp[i], gg[i], a, b);
            pp[i] = np; mm[i] = a; vv[i] = b;
            if (sh) sh[i] = (T)np;
        }
    }
}

// Non-finite guard of a scaled fp16 step (the reference's numerical-failure policy is the per-epoch NaN check of ad_trainer.py:448-449;
// the gradient scale this build adds for fp16 needs its own): one streaming pass over the gradients the optimiser is about to apply,
// through the SAME chunk table.  state = {flag of even steps, flag of odd steps, steps skipped so far, steps checked so far}: a chunk
// with an inf / NaN raises flag[parity]; eoe_adam_multi / eoe_sgd_multi given &state[parity] then leave p, m, v untouched.  Block 0 also
// retires the OTHER flag (the previous step's, final since that step's optimiser launch has completed in stream order) into the
// skipped-steps count, so nothing is ever reset from the host and nothing needs a host synchronisation.
__global__ __launch_bounds__(256) void grads_nonfinite_kernel(const float* __restrict__ g, const eoe_adam_chunk* __restrict__ chunks,
                                                              int* __restrict__ state, int parity, int first) {
    if (first && blockIdx.x == 0 && threadIdx.x == 0) {
        if (state[1 - parity]) state[2] += 1;
        state[1 - parity] = 0;
        state[3] += 1;
    }
    const eoe_adam_chunk ck = chunks[blockIdx.x];
    const float* gg = g + ck.g_off;
    unsigned bad = 0;
    auto test = [&](float x) { bad |= ((__float_as_uint(x) & 0x7f800000u) == 0x7f800000u) ? 1u : 0u; };
    if ((ck.g_off & 3) == 0) {
        const int n4 = ck.n >> 2;
        int i = threadIdx.x;
        for (; i + 3 * (int)blockDim.x < n4; i += 4 * blockDim.x) {          // four independent loads per thread in flight
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const f32x4*)(gg + (size_t)(i + u * (int)blockDim.x) * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) { test(v[u][0]); test(v[u][1]); test(v[u][2]); test(v[u][3]); }
        }
        for (; i < n4; i += blockDim.x) {
            const f32x4 v = *(const f32x4*)(gg + i * 4);
            test(v[0]); test(v[1]); test(v[2]); test(v[3]);
        }
        for (int i = (n4 << 2) + threadIdx.x; i < ck.n; i += blockDim.x) test(gg[i]);
    } else {
        for (int i = threadIdx.x; i < ck.n; i += blockDim.x) test(gg[i]);
    }
    if (__any(bad != 0) && (threadIdx.x & 63) == 0) state[parity] = 1;       // same value from every writer: plain stores
}

}  // namespace

#define DISPATCH_NV(D, ...)                                      \
    do {                                                         \
        switch ((D) / 256) {                                     \
            case 1: { constexpr int NV = 1; __VA_ARGS__; } break; \
            case 2: { constexpr int NV = 2; __VA_ARGS__; } break; \
            case 3: { constexpr int NV = 3; __VA_ARGS__; } break; \
            default: { constexpr int NV = 4; __VA_ARGS__; } break; \
        }                                                        \
    } while (0)

#define DISPATCH_T(dtype, ...)                                  \
    do {                                                        \
        if ((dtype) == EOE_F16) { typedef f16_t T; __VA_ARGS__; } \
        else if ((dtype) == EOE_BF16) { typedef bf16_t T; __VA_ARGS__; } \
        else return eoe_set_error(EOE_ERR_ARG, "bad dtype %d", (int)(dtype)); \
    } while (0)

extern "C" int eoe_cast(const float* src, void* dst, size_t n, int dtype, void* stream) {
    EOE_CHECK_ARG(src && dst, "cast: null pointer");
    ProfScope ps("cast", 0, 6.0 * n, stream);
    if (n == 0) return 0;
    int grid = (int)((n / 4 + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    DISPATCH_T(dtype, hipLaunchKernelGGL((cast_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (T*)dst, n));
    EOE_CHECK_LAUNCH("cast");
    return 0;
}

extern "C" int eoe_cast_transpose(const float* src, void* dst, void* dst_t, int rows, int cols, int dtype,
                                  void* stream) {
    EOE_CHECK_ARG(src && (dst || dst_t) && rows > 0 && cols > 0, "cast_transpose: bad args");
    ProfScope ps("cast_transpose", 0, ((dst ? 2.0 : 0.0) + (dst_t ? 2.0 : 0.0) + 4.0) * rows * cols, stream);
    dim3 grid(cdiv(cols, 64), cdiv(rows, 64));
    if ((rows & 3) == 0 && (cols & 3) == 0) {
        DISPATCH_T(dtype, hipLaunchKernelGGL((cast_transpose_vec_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, src,
                                             (T*)dst, (T*)dst_t, rows, cols));
    } else {
        DISPATCH_T(dtype, hipLaunchKernelGGL((cast_transpose_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, src,
                                             (T*)dst, (T*)dst_t, rows, cols));
    }
    EOE_CHECK_LAUNCH("cast_transpose");
    return 0;
}

extern "C" int eoe_cast_transpose_multi(const eoe_cast_job* jobs, int count, int dtype, void* stream) {
    EOE_CHECK_ARG(jobs && count > 0, "cast_transpose_multi: bad args");
    double bytes = 0;
    for (int i = 0; i < count; ++i) {
        const eoe_cast_job& j = jobs[i];
        EOE_CHECK_ARG(j.src && (j.dst || j.dst_t) && j.rows > 0 && j.cols > 0 && (j.rows & 3) == 0 && (j.cols & 3) == 0,
                      "cast_transpose_multi: job %d: null pointers or rows / cols not multiples of 4", i);
        bytes += ((j.dst ? 2.0 : 0.0) + (j.dst_t ? 2.0 : 0.0) + 4.0) * j.rows * j.cols;
    }
    ProfScope ps("cast_transpose", 0, bytes, stream);
    for (int first = 0; first < count; first += CT_MAX_JOBS) {
        CtBatch b;
        b.count = count - first < CT_MAX_JOBS ? count - first : CT_MAX_JOBS;
        int tiles = 0;
        for (int i = 0; i < b.count; ++i) {
            const eoe_cast_job& j = jobs[first + i];
            b.src[i] = j.src; b.dst[i] = j.dst; b.dst_t[i] = j.dst_t; b.rows[i] = j.rows; b.cols[i] = j.cols;
            b.tile_start[i] = tiles;
            tiles += cdiv(j.rows, 64) * cdiv(j.cols, 64);
        }
        b.tile_start[b.count] = tiles;
        DISPATCH_T(dtype, hipLaunchKernelGGL((cast_transpose_multi_kernel<T>), dim3(tiles), dim3(256), 0, (hipStream_t)stream, b));
        EOE_CHECK_LAUNCH("cast_transpose_multi");
    }
    return 0;
}

extern "C" int eoe_patchify(const float* x, const float* mean, const float* stdv, void* out, int n, int res,
                            int patch, int dtype, void* stream) {
    EOE_CHECK_ARG(x && out && n > 0, "patchify: bad args");
    EOE_CHECK_ARG(res % patch == 0 && patch % 4 == 0, "patchify: res %% patch != 0 or patch %% 4 != 0");
    EOE_CHECK_ARG((mean == nullptr) == (stdv == nullptr), "patchify: mean/std must both be given or both NULL");
    ProfScope ps("patchify", 0, 6.0 * n * 3 * res * res, stream);
    const int g = res / patch;
    DISPATCH_T(dtype, hipLaunchKernelGGL((patchify_kernel<T>), dim3(n * g), dim3(256), 0, (hipStream_t)stream, x, mean,
                                         stdv, (T*)out, res, patch));
    EOE_CHECK_LAUNCH("patchify");
    return 0;
}

extern "C" int eoe_layernorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, void* y,
                                 float* stats, int rows, int D, float eps, int dtype, int out_f32, void* stream) {
    EOE_CHECK_ARG(x && gamma && beta && y && rows > 0, "layernorm_fwd: bad args");
    EOE_CHECK_ARG(D % 256 == 0 && D <= 1024 && ldx % 4 == 0, "layernorm: D must be a multiple of 256, <= 1024");
    ProfScope ps("layernorm_fwd", 0, (4.0 + (out_f32 ? 4.0 : 2.0)) * rows * D, stream);
    DISPATCH_T(dtype, DISPATCH_NV(D, hipLaunchKernelGGL((layernorm_fwd_kernel<T, NV>), dim3(cdiv(rows, 4)), dim3(256), 0,
                                         (hipStream_t)stream, x, ldx, gamma, beta, y, stats, rows, D, eps, out_f32)));
    EOE_CHECK_LAUNCH("layernorm_fwd");
    return 0;
}

extern "C" int eoe_layernorm_bwd(const void* dy, int dy_f32, const float* x, int ldx, const float* stats,
                                 const float* gamma, const float* dres, float* dx_out, int ld_out, void* dx16,
                                 float* dgamma, float* dbeta, float* dxsum, float* red_scratch, int rows, int D, int dtype,
                                 void* stream) {
    EOE_CHECK_ARG(dy && x && stats && gamma && dx_out && rows > 0, "layernorm_bwd: bad args");
    EOE_CHECK_ARG(D % 256 == 0 && D <= 1024 && ldx % 4 == 0 && ld_out % 4 == 0, "layernorm: D must be a multiple of 256, <= 1024");
    EOE_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "layernorm_bwd: dgamma/dbeta must both be given or both NULL");
    ProfScope ps("layernorm_bwd", 0, ((dy_f32 ? 4.0 : 2.0) + 4.0 + (dres ? 4.0 : 0.0) + 4.0 + (dx16 ? 2.0 : 0.0)) * rows * D, stream);
    int grid = cdiv(rows, 8);
    if (grid > EOE_LN_PARTIALS) grid = EOE_LN_PARTIALS;
    float* part = (dgamma || dxsum) ? red_scratch : nullptr;
    // (the attribute once per instantiation, not per launch)
#define EOE_LNB_LAUNCH(F, R)                                                                                                           \
    { static bool once = (hipFuncSetAttribute((const void*)layernorm_bwd_kernel<T, NV, F, R>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 8 * 256 * NV * 4), true); (void)once; } \
    hipLaunchKernelGGL((layernorm_bwd_kernel<T, NV, F, R>), dim3(grid), dim3(512), 3 * 8 * 256 * NV * 4, (hipStream_t)stream, dy, x, ldx, stats, gamma, \
                       dres, dx_out, ld_out, (T*)dx16, dgamma, dbeta, dxsum, part, rows, D)
    DISPATCH_T(dtype, DISPATCH_NV(D, {
        if (dy_f32) { if (dres) { EOE_LNB_LAUNCH(true, true); } else { EOE_LNB_LAUNCH(true, false); } }
        else { if (dres) { EOE_LNB_LAUNCH(false, true); } else { EOE_LNB_LAUNCH(false, false); } }
    }));
#undef EOE_LNB_LAUNCH
    EOE_CHECK_LAUNCH("layernorm_bwd");
    if (part) EOE_TRY(eoe_finish_reduce(part, grid, 3 * D, D, dgamma, dbeta, dxsum, 1, stream));
    return 0;
}

thread_local EoeRedJobs* eoe_tls_defer = nullptr;

bool eoe_defer_reduce(const float* part, int R, int N, int seg, float* o0, float* o1, float* o2, int blocked) {
    EoeRedJobs* j = eoe_tls_defer;
    if (!j || j->count >= 6 || (N & 3) || (seg & 3)) return false;
    EoeRedJob& r = j->job[j->count];
    r.part = part; r.R = R; r.N = N; r.seg = seg; r.blocked = blocked; r.out[0] = o0; r.out[1] = o1; r.out[2] = o2;
    j->tile_start[j->count + 1] = j->tile_start[j->count] + (N + 63) / 64;
    j->count += 1;
    return true;
}

int eoe_finish_reduce(const float* part, int R, int N, int seg, float* o0, float* o1, float* o2, int blocked, void* stream) {
    if (eoe_defer_reduce(part, R, N, seg, o0, o1, o2, blocked)) return 0;
    EOE_CHECK_ARG(!eoe_tls_defer && (N & 3) == 0 && (seg & 3) == 0, "finish_reduce: batch full or widths not multiples of 4");
    EoeRedJobs one;
    one.count = 0; one.tile_start[0] = 0; one.overwrite = 0;
    eoe_tls_defer = &one;
    eoe_defer_reduce(part, R, N, seg, o0, o1, o2, blocked);
    eoe_tls_defer = nullptr;
    return eoe_flush_reduce(&one, stream);
}

int eoe_flush_reduce(EoeRedJobs* jobs, void* stream) {
    if (!jobs || jobs->count == 0) return 0;
    double bytes = 0;
    for (int j = 0; j < jobs->count; ++j) bytes += 4.0 * jobs->job[j].R * jobs->job[j].N;
    ProfScope ps("finish_reduce", 0, bytes, stream);
    hipLaunchKernelGGL(multi_reduce_kernel<EoeRedJobs>, dim3(jobs->tile_start[jobs->count]), dim3(1024), 0, (hipStream_t)stream, *jobs);
    EOE_CHECK_LAUNCH("multi_reduce");
    jobs->count = 0;
    return 0;
}

// a block's jobs go to the caller's table instead of a launch of their own (eoe_vit_block_bwd_args.red_table); a full table is flushed first
int eoe_red_table_append(eoe_red_table* t, EoeRedJobs* jobs, void* stream) {
    if (!jobs || jobs->count == 0) return 0;
    if (t->count < 0 || t->count > EOE_RED_TABLE_MAX) return eoe_set_error(EOE_ERR_ARG, "red_table: bad count %d", t->count);
    if (t->count + jobs->count > EOE_RED_TABLE_MAX) EOE_TRY(eoe_red_table_flush(t, stream));
    if (t->count == 0) t->overwrite = jobs->overwrite;
    else if (t->overwrite != jobs->overwrite) {        // mixed accumulate modes: keep them in separate launches
        EOE_TRY(eoe_red_table_flush(t, stream));
        t->overwrite = jobs->overwrite;
    }
    for (int j = 0; j < jobs->count; ++j) {
        const EoeRedJob& s = jobs->job[j];
        eoe_red_job& d = t->job[t->count++];
        d.part = s.part; d.R = s.R; d.N = s.N; d.seg = s.seg; d.blocked = s.blocked; d.out[0] = s.out[0]; d.out[1] = s.out[1]; d.out[2] = s.out[2];
    }
    jobs->count = 0;
    return 0;
}

extern "C" int eoe_red_table_flush(eoe_red_table* t, void* stream) {
    EOE_CHECK_ARG(t != nullptr && t->count >= 0 && t->count <= EOE_RED_TABLE_MAX, "eoe_red_table_flush: bad table");
    for (int first = 0; first < t->count; first += 64) {
        EoeRedJobsBig big;
        big.count = t->count - first < 64 ? t->count - first : 64;
        big.overwrite = t->overwrite;
        big.tile_start[0] = 0;
        double bytes = 0;
        for (int j = 0; j < big.count; ++j) {
            const eoe_red_job& s = t->job[first + j];
            EOE_CHECK_ARG(s.part && s.R > 0 && s.N > 0 && (s.N & 3) == 0 && s.seg > 0 && (s.seg & 3) == 0, "eoe_red_table_flush: bad job %d", first + j);
            EoeRedJob& d = big.job[j];
            d.part = s.part; d.R = s.R; d.N = s.N; d.seg = s.seg; d.blocked = s.blocked; d.out[0] = s.out[0]; d.out[1] = s.out[1]; d.out[2] = s.out[2];
            big.tile_start[j + 1] = big.tile_start[j] + (s.N + 63) / 64;
            bytes += 4.0 * s.R * s.N;
        }
        ProfScope ps("finish_reduce", 0, bytes, stream);
        hipLaunchKernelGGL(multi_reduce_kernel<EoeRedJobsBig>, dim3(big.tile_start[big.count]), dim3(1024), 0, (hipStream_t)stream, big);
        EOE_CHECK_LAUNCH("multi_reduce (table)");
    }
    t->count = 0;
    return 0;
}

extern "C" int eoe_embed_lnpre_fwd(const float* tok, const float* cls, const float* pos, const float* gamma,
                                   const float* beta, float* x0, float* y, float* stats, int n, int L, int D,
                                   float eps, void* stream) {
    EOE_CHECK_ARG(tok && cls && pos && gamma && beta && x0 && y && stats && n > 0 && L > 1, "embed_lnpre_fwd: bad args");
    EOE_CHECK_ARG(D % 256 == 0 && D <= 1024, "embed_lnpre: D must be a multiple of 256, <= 1024");
    DISPATCH_NV(D, hipLaunchKernelGGL((embed_lnpre_fwd_kernel<NV>), dim3(cdiv(n * L, 4)), dim3(256), 0, (hipStream_t)stream, tok, cls, pos,
                       gamma, beta, x0, y, stats, n, L, D, eps));
    EOE_CHECK_LAUNCH("embed_lnpre_fwd");
    return 0;
}

extern "C" int eoe_embed_lnpre_bwd(const float* dy, const float* x0, const float* stats, const float* gamma,
                                   void* dtok, float* dcls, float* dpos, float* dgamma, float* dbeta, float* scratch, int n,
                                   int L, int D, int dtype, void* stream) {
    EOE_CHECK_ARG(dy && x0 && stats && gamma && dtok && dcls && dpos && dgamma && dbeta && n > 0 && L > 1,
                  "embed_lnpre_bwd: bad args");
    EOE_CHECK_ARG(D % 256 == 0 && D <= 1024, "embed_lnpre: D must be a multiple of 256, <= 1024");
    DISPATCH_T(dtype, DISPATCH_NV(D, hipLaunchKernelGGL((embed_lnpre_bwd_kernel<T, NV>), dim3(L), dim3(256), 0, (hipStream_t)stream, dy, x0,
                                         stats, gamma, (T*)dtok, dcls, dpos, dgamma, dbeta, scratch, n, L, D)));
    EOE_CHECK_LAUNCH("embed_lnpre_bwd");
    if (scratch) EOE_TRY(eoe_finish_reduce(scratch, L, 2 * D, D, dgamma, dbeta, nullptr, 1, stream));
    return 0;
}

extern "C" int eoe_colsum(const void* x, int ldx, float* out, int rows, int cols, int dtype, int accumulate,
                          void* stream) {
    EOE_CHECK_ARG(x && out && rows > 0 && cols > 0, "colsum: bad args");
    EOE_CHECK_ARG(cols % 4 == 0 && ldx % 4 == 0, "colsum: cols and ldx must be multiples of 4");
    ProfScope ps("colsum", 0, 2.0 * rows * cols, stream);
    if (!accumulate) {
        if (hipMemsetAsync(out, 0, (size_t)cols * sizeof(float), (hipStream_t)stream) != hipSuccess)
            return eoe_set_error(EOE_ERR_LAUNCH, "colsum: memset failed");
    }
    int cpb = 1;
    while (cpb < 64 && cpb < cols / 4) cpb *= 2;
    const int rpb = 256 / cpb, gx = cdiv(cols / 4, cpb);
    int gy = cdiv(rows, rpb * 16);
    if (gy > 128) gy = 128;
    if (gx * gy > 1024) gy = 1024 / gx;
    if (gy < 1) gy = 1;
    DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_kernel<T>), dim3(gx, gy), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)x, ldx, out, rows, cols, cpb));
    EOE_CHECK_LAUNCH("colsum");
    return 0;
}

extern "C" int eoe_colsum_det(const void* x, int ldx, float* out, float* scratch, int rows, int cols, int dtype, void* stream) {
    EOE_CHECK_ARG(x && out && scratch && rows > 0 && cols > 0, "colsum_det: bad args");
    EOE_CHECK_ARG(cols % 4 == 0 && ldx % 4 == 0, "colsum_det: cols and ldx must be multiples of 4");
    ProfScope ps("colsum", 0, 2.0 * rows * cols, stream);
    int cpb = 1;
    while (cpb < 64 && cpb < cols / 4) cpb *= 2;
    const int rpb = 256 / cpb, gx = cdiv(cols / 4, cpb);
    int gy = cdiv(rows, rpb * 16);
    if (gy > EOE_COLSUM_PARTIALS) gy = EOE_COLSUM_PARTIALS;
    if (gx * gy > 1024) gy = 1024 / gx;
    if (gy < 1) gy = 1;
    DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_partial_kernel<T>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const T*)x, ldx,
                                         scratch, rows, cols, cpb));
    EOE_CHECK_LAUNCH("colsum_partial");
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3(cdiv(cols, 64)), dim3(1024), 0, (hipStream_t)stream, (const float*)scratch, out, gy, cols);
    EOE_CHECK_LAUNCH("colsum_reduce");
    return 0;
}

extern "C" int eoe_linear_small_fwd(const float* x, const float* w, const float* bias, float* y, int M, int N, int K,
                                    void* stream) {
    EOE_CHECK_ARG(x && w && y && M > 0 && K > 0, "linear_small_fwd: bad args");
    EOE_CHECK_ARG(N >= 1 && N <= SMALL_N, "linear_small: N = %d not in [1, %d]", N, SMALL_N);
    hipLaunchKernelGGL(linear_small_fwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, M, N, K);
    EOE_CHECK_LAUNCH("linear_small_fwd");
    return 0;
}

extern "C" int eoe_linear_small_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db,
                                    int M, int N, int K, int accumulate, void* stream) {
    EOE_CHECK_ARG(x && w && dy && M > 0 && K > 0, "linear_small_bwd: bad args");
    EOE_CHECK_ARG(N >= 1 && N <= SMALL_N, "linear_small: N = %d not in [1, %d]", N, SMALL_N);
    hipStream_t s = (hipStream_t)stream;
    if (dx) {
        hipLaunchKernelGGL(linear_small_dx_kernel, dim3(cdiv(M, 4)), dim3(256), 0, s, dy, w, dx, M, N, K);
        EOE_CHECK_LAUNCH("linear_small_dx");
    }
    if (dw) {
        if (!accumulate) {
            if (hipMemsetAsync(dw, 0, (size_t)N * K * sizeof(float), s) != hipSuccess ||
                (db && hipMemsetAsync(db, 0, (size_t)N * sizeof(float), s) != hipSuccess))
                return eoe_set_error(EOE_ERR_LAUNCH, "linear_small_bwd: memset failed");
        }
        int gy = cdiv(M, 32);
        if (gy > 64) gy = 64;
        hipLaunchKernelGGL(linear_small_dw_kernel, dim3(cdiv(K, 256), gy), dim3(256), 0, s, x, dy, dw, db, M, N, K);
        EOE_CHECK_LAUNCH("linear_small_dw");
    }
    return 0;
}

extern "C" int eoe_zero_multi(float* const* ptrs, const int* counts, int n, void* stream) {
    EOE_CHECK_ARG(ptrs && counts && n >= 1 && n <= EOE_ZERO_MAX, "zero_multi: n %d not in [1, %d]", n, EOE_ZERO_MAX);
    ZeroArgs a;
    int mx = 0;
    for (int i = 0; i < n; ++i) {
        EOE_CHECK_ARG(ptrs[i] && counts[i] >= 0, "zero_multi: bad buffer %d", i);
        a.p[i] = ptrs[i]; a.n[i] = counts[i];
        if (counts[i] > mx) mx = counts[i];
    }
    a.count = n;
    int gx = cdiv(mx, 1024);                       // one float4 per thread per pass
    if (gx > 2048) gx = 2048;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(zero_multi_kernel, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, a);
    EOE_CHECK_LAUNCH("zero_multi");
    return 0;
}

extern "C" int eoe_cast_colsum(const float* x, void* dst, float* out, float* scratch, int rows, int cols, int dtype, int accumulate,
                               void* stream) {
    EOE_CHECK_ARG(x && dst && out && rows > 0 && cols > 0, "cast_colsum: bad args");
    EOE_CHECK_ARG(cols % 4 == 0, "cast_colsum: cols must be a multiple of 4");
    ProfScope ps("cast_colsum", 0, 6.0 * rows * cols, stream);
    int gy = cdiv(rows, 4 * 8);
    if (gy > EOE_CAST_COLSUM_PARTIALS) gy = EOE_CAST_COLSUM_PARTIALS;
    // with a scratch: per-workgroup-row partial sums + a fixed-order finish (deferred into the caller's batch when one is open);
    // a deferred finish follows the batch's overwrite / accumulate mode, so nothing is zeroed here
    const bool deferred_mode = scratch && eoe_tls_defer;
    const int blocked = (cols & 63) == 0;
    if (!accumulate && !deferred_mode) {
        if (hipMemsetAsync(out, 0, (size_t)cols * sizeof(float), (hipStream_t)stream) != hipSuccess)
            return eoe_set_error(EOE_ERR_LAUNCH, "cast_colsum: memset failed");
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL((cast_colsum_kernel<T>), dim3(cdiv(cols, 256), gy), dim3(256), 0,
                                         (hipStream_t)stream, x, (T*)dst, out, scratch, rows, cols, blocked));
    EOE_CHECK_LAUNCH("cast_colsum");
    if (scratch) EOE_TRY(eoe_finish_reduce(scratch, gy, cols, cols, out, nullptr, nullptr, blocked, stream));
    return 0;
}

extern "C" int eoe_hsc_fwd(const float* f, const int64_t* labels, int64_t nominal_label, float* loss, float* scores,
                           float* dists, float* losses, int n, int d, float inv_count, void* stream) {
    EOE_CHECK_ARG(f && labels && n > 0 && d > 0, "hsc_fwd: bad args");
    EOE_CHECK_ARG(!loss || losses, "hsc_fwd: the loss needs the per-sample `losses` buffer");
    hipLaunchKernelGGL(hsc_rows_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, f, labels, nominal_label,
                       scores, dists, losses, n, d);
    EOE_CHECK_LAUNCH("hsc_rows");
    if (loss) {
        hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, losses, loss, n, inv_count);
        EOE_CHECK_LAUNCH("hsc_sum");
    }
    return 0;
}

extern "C" int eoe_hsc_score(const float* f, float* scores, int n, int d, void* stream) {
    EOE_CHECK_ARG(f && scores && n > 0 && d > 0, "hsc_score: bad args");
    hipLaunchKernelGGL(hsc_rows_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, f, (const int64_t*)nullptr,
                       (int64_t)0, scores, (float*)nullptr, (float*)nullptr, n, d);
    EOE_CHECK_LAUNCH("hsc_score");
    return 0;
}

extern "C" int eoe_hsc_bwd(const float* f, const int64_t* labels, int64_t nominal_label, const float* gscale,
                           float* df, void* df16, int n, int d, float inv_count, int dtype, void* stream) {
    EOE_CHECK_ARG(f && labels && (df || df16) && n > 0 && d > 0, "hsc_bwd: bad args");
    DISPATCH_T(dtype, hipLaunchKernelGGL((hsc_bwd_kernel<T>), dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, f,
                                         labels, nominal_label, gscale, df, (T*)df16, n, d, inv_count));
    EOE_CHECK_LAUNCH("hsc_bwd");
    return 0;
}

extern "C" int eoe_bce_fwd(const float* x, const int64_t* labels, int64_t nominal_label, float* loss, float* scores,
                           float* losses, int n, float inv_count, void* stream) {
    EOE_CHECK_ARG(x && labels && n > 0, "bce_fwd: bad args");
    EOE_CHECK_ARG(!loss || losses, "bce_fwd: the loss needs the per-sample `losses` buffer");
    hipLaunchKernelGGL(bce_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, labels, nominal_label,
                       scores, losses, n);
    EOE_CHECK_LAUNCH("bce_rows");
    if (loss) {
        hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, losses, loss, n, inv_count);
        EOE_CHECK_LAUNCH("bce_sum");
    }
    return 0;
}

extern "C" int eoe_bce_bwd(const float* x, const int64_t* labels, const float* gscale, float* dx, int n,
                           float inv_count, void* stream) {
    EOE_CHECK_ARG(x && labels && dx && n > 0, "bce_bwd: bad args");
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, labels, gscale, dx, n,
                       inv_count);
    EOE_CHECK_LAUNCH("bce_bwd");
    return 0;
}

extern "C" int eoe_dsad_fwd(const float* f, const int64_t* labels, int64_t nominal_label, float* loss, float* losses, int n, int d,
                            float inv_count, void* stream) {
    EOE_CHECK_ARG(f && labels && losses && n > 0 && d > 0, "dsad_fwd: bad args");
    hipLaunchKernelGGL(rowobj_fwd_kernel<1>, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, f, (const float*)nullptr, labels,
                       nominal_label, losses, n, d);
    EOE_CHECK_LAUNCH("dsad_rows");
    if (loss) {
        hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, losses, loss, n, inv_count);
        EOE_CHECK_LAUNCH("dsad_sum");
    }
    return 0;
}

extern "C" int eoe_dsad_bwd(const float* f, const int64_t* labels, int64_t nominal_label, const float* gscale, float* df, int n,
                            int d, float inv_count, void* stream) {
    EOE_CHECK_ARG(f && labels && df && n > 0 && d > 0, "dsad_bwd: bad args");
    hipLaunchKernelGGL(rowobj_bwd_kernel<1>, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, f, (const float*)nullptr, labels,
                       nominal_label, gscale, df, n, d, inv_count);
    EOE_CHECK_LAUNCH("dsad_bwd");
    return 0;
}

extern "C" int eoe_dsvdd_fwd(const float* f, const float* center, float* loss, float* dists, int n, int d, float inv_count,
                             void* stream) {
    EOE_CHECK_ARG(f && center && dists && n > 0 && d > 0, "dsvdd_fwd: bad args");
    hipLaunchKernelGGL(rowobj_fwd_kernel<2>, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, f, center, (const int64_t*)nullptr,
                       (int64_t)0, dists, n, d);
    EOE_CHECK_LAUNCH("dsvdd_rows");
    if (loss) {
        hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, dists, loss, n, inv_count);
        EOE_CHECK_LAUNCH("dsvdd_sum");
    }
    return 0;
}

extern "C" int eoe_dsvdd_bwd(const float* f, const float* center, const float* gscale, float* df, int n, int d, float inv_count,
                             void* stream) {
    EOE_CHECK_ARG(f && center && df && n > 0 && d > 0, "dsvdd_bwd: bad args");
    hipLaunchKernelGGL(rowobj_bwd_kernel<2>, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, f, center, (const int64_t*)nullptr,
                       (int64_t)0, gscale, df, n, d, inv_count);
    EOE_CHECK_LAUNCH("dsvdd_bwd");
    return 0;
}

extern "C" int eoe_focal_fwd(const float* x, const int64_t* labels, int64_t nominal_label, float* loss, float* scores,
                             float* losses, int n, float inv_count, float gamma, float eps, void* stream) {
    EOE_CHECK_ARG(x && labels && n > 0 && gamma >= 1.0f && eps > 0.f && eps < 0.5f, "focal_fwd: bad args");
    EOE_CHECK_ARG(!loss || losses, "focal_fwd: the loss needs the per-sample `losses` buffer");
    hipLaunchKernelGGL(focal_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, labels, nominal_label, scores,
                       losses, n, gamma, eps);
    EOE_CHECK_LAUNCH("focal_rows");
    if (loss) {
        hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, losses, loss, n, inv_count);
        EOE_CHECK_LAUNCH("focal_sum");
    }
    return 0;
}

extern "C" int eoe_focal_bwd(const float* x, const int64_t* labels, const float* gscale, float* dx, int n, float inv_count,
                             float gamma, float eps, void* stream) {
    EOE_CHECK_ARG(x && labels && dx && n > 0 && gamma >= 1.0f && eps > 0.f && eps < 0.5f, "focal_bwd: bad args");
    hipLaunchKernelGGL(focal_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, labels, gscale, dx, n, inv_count,
                       gamma, eps);
    EOE_CHECK_LAUNCH("focal_bwd");
    return 0;
}

extern "C" int eoe_auc_ap(const float* scores, const int64_t* labels, int64_t positive_label, double* out, void* scratch, int n,
                          void* stream) {
    EOE_CHECK_ARG(scores && labels && out && scratch && n > 0, "auc_ap: bad args");
    const int nb = cdiv(n, 256);
    unsigned long long* pw = (unsigned long long*)scratch;
    double* pp = (double*)(pw + nb);
    unsigned long long* pc = (unsigned long long*)(pp + nb);
    hipLaunchKernelGGL(rank_pairs_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, scores, labels, positive_label, pw, pp, pc, n);
    EOE_CHECK_LAUNCH("auc_ap_pairs");
    hipLaunchKernelGGL(rank_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned long long*)pw, (const double*)pp,
                       (const unsigned long long*)pc, nb, n, out);
    EOE_CHECK_LAUNCH("auc_ap_finish");
    return 0;
}

extern "C" int eoe_clip_fwd(const float* f, const float* text, const int64_t* labels, int64_t nominal_label, int leave_one_out,
                            float* loss, float* scores, float* losses, int n, int d, int T, float inv_count, void* stream) {
    EOE_CHECK_ARG(f && text && labels && n > 0 && d > 0 && T >= 2 && T <= 64, "clip_fwd: bad args (2 <= T <= 64)");
    EOE_CHECK_ARG(!loss || losses, "clip_fwd: the loss needs the per-sample `losses` buffer");
    hipLaunchKernelGGL(clip_rows_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, f, text, labels, nominal_label, leave_one_out,
                       scores, losses, n, d, T);
    EOE_CHECK_LAUNCH("clip_rows");
    if (loss) {
        hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, losses, loss, n, inv_count);
        EOE_CHECK_LAUNCH("clip_sum");
    }
    return 0;
}

extern "C" int eoe_clip_bwd(const float* f, const float* text, const int64_t* labels, int64_t nominal_label, int leave_one_out,
                            const float* gscale, float* df, int n, int d, int T, float inv_count, void* stream) {
    EOE_CHECK_ARG(f && text && labels && df && n > 0 && d > 0 && T >= 2 && T <= 64, "clip_bwd: bad args (2 <= T <= 64)");
    hipLaunchKernelGGL(clip_bwd_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, f, text, labels, nominal_label, leave_one_out,
                       gscale, df, n, d, T, inv_count);
    EOE_CHECK_LAUNCH("clip_bwd");
    return 0;
}

extern "C" int eoe_clip_score(const float* f, const float* text, float* scores, int n, int d, int T, void* stream) {
    EOE_CHECK_ARG(f && text && scores && n > 0 && d > 0 && T >= 2 && T <= 64, "clip_score: bad args (2 <= T <= 64)");
    hipLaunchKernelGGL(clip_rows_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, f, text, (const int64_t*)nullptr, (int64_t)0, 0,
                       scores, (float*)nullptr, n, d, T);
    EOE_CHECK_LAUNCH("clip_score");
    return 0;
}

extern "C" int eoe_grads_nonfinite(const float* g, const eoe_adam_chunk* chunks, int n_chunks, int32_t* state, int parity, int first,
                                  void* stream) {
    EOE_CHECK_ARG(g && chunks && state && n_chunks > 0 && (parity == 0 || parity == 1), "grads_nonfinite: bad args");
    ProfScope ps("grads_nonfinite", 0, 4.0 * n_chunks * EOE_ADAM_CHUNK, stream);
    hipLaunchKernelGGL(grads_nonfinite_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, g, chunks, (int*)state, parity, first);
    EOE_CHECK_LAUNCH("grads_nonfinite");
    return 0;
}

extern "C" int eoe_sgd_multi(float* p, const float* g, float* buf, const eoe_adam_chunk* chunks, int n_chunks, float lr,
                             float momentum, float weight_decay, int nesterov, float grad_scale_inv, const int32_t* skip_flag, void* stream) {
    EOE_CHECK_ARG(p && g && chunks && n_chunks > 0 && (buf || momentum == 0.f), "sgd_multi: bad args");
    ProfScope ps("sgd_multi", 0, 20.0 * n_chunks * EOE_ADAM_CHUNK, stream);
    hipLaunchKernelGGL(sgd_multi_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, p, g, buf, chunks, lr, momentum, weight_decay,
                       nesterov, grad_scale_inv > 0.f ? grad_scale_inv : 1.0f, (const int*)skip_flag);
    EOE_CHECK_LAUNCH("sgd_multi");
    return 0;
}

extern "C" int eoe_adam_tiles(float* p, const float* g, float* m, float* v, const eoe_adam_tile* tiles, int n_tiles,
                              const eoe_adam_scalars* scalars, float beta1, float beta2, float eps, float weight_decay, int dtype,
                              const int32_t* skip_flag, void* stream) {
    EOE_CHECK_ARG(p && g && m && v && tiles && scalars && n_tiles > 0, "adam_tiles: bad args");
    // per tile: p, g, m, v read (16 B), p, m, v written (12 B), the two 16-bit copies written (4 B)
    ProfScope ps("adam_tiles", 0, 32.0 * n_tiles * 4096, stream);
    eoe_adam_scalars sc = *scalars;
    if (!(sc.grad_scale_inv > 0.f)) sc.grad_scale_inv = 1.0f;
    DISPATCH_T(dtype, hipLaunchKernelGGL((adam_tiles_kernel<T>), dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, p, g, m, v, tiles, sc,
                                         beta1, beta2, eps, weight_decay, (const int*)skip_flag));
    EOE_CHECK_LAUNCH("adam_tiles");
    return 0;
}

extern "C" int eoe_adam_multi(float* p, const float* g, float* m, float* v, const eoe_adam_chunk* chunks,
                              int n_chunks, const eoe_adam_scalars* scalars, float beta1, float beta2, float eps,
                              float weight_decay, void* shadow16, int dtype, const int32_t* skip_flag, void* stream) {
    EOE_CHECK_ARG(p && g && m && v && chunks && scalars && n_chunks > 0, "adam_multi: bad args");
    ProfScope ps("adam_multi", 0, 28.0 * n_chunks * EOE_ADAM_CHUNK, stream);
    if (!shadow16) dtype = EOE_BF16;
    eoe_adam_scalars sc = *scalars;
    if (!(sc.grad_scale_inv > 0.f)) sc.grad_scale_inv = 1.0f;        // 0 (a zero-initialised struct) = no scaling
    DISPATCH_T(dtype, hipLaunchKernelGGL((adam_multi_kernel<T>), dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, p, g,
                                         m, v, chunks, sc, beta1, beta2, eps, weight_decay, (T*)shadow16, (const int*)skip_flag));
    EOE_CHECK_LAUNCH("adam_multi");
    return 0;
}
