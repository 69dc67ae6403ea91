// Shared pieces of the NT GEMM kernels (gemm.hip, gemm256.hip): the kernel argument block, the two epilogues and the waits.
#pragma once
#include "common.h"

namespace {

constexpr int BM = 256, BN = 128, BK = 64, NSTAGE = 3;
#ifndef EOE_NT_DEFAULT_FLAGS
#define EOE_NT_DEFAULT_FLAGS 1     // bit 0 (fast epilogue): 551 vs 669 us per ViT layer; bit 1 (asm fragment reads): no gain here (574 vs 573 us)
#endif
constexpr int A_BYTES = BM * BK * 2;              // 32 KiB
constexpr int B_BYTES = BN * BK * 2;              // 16 KiB
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;    // 48 KiB
constexpr int SMEM_BYTES = NSTAGE * STAGE_BYTES;  // 144 KiB: one 8-wave workgroup per CU

struct GemmP {
    const void* A; const void* B; void* C; const float* bias; const void* aux; void* aux_out; float* colsum;
    float* colsum_part;            // column sums through per-wave partial rows (eoe_part_index layout) + a finish kernel (no atomics)
    int colsum_blocked;            // layout of the partial rows: N % 64 == 0 -> blocked (BatchNorm statistics: always plain [R][2][N])
    int colsum_sq;                 // partial rows are [.][2][N]: sums and sums of squares (BatchNorm statistics); no finish kernel
    int M, N, K, lda, ldb, ldc, ldaux, out_f32, accumulate;
    int split_k;                   // eoe_gemm_args.split_k (a hint: launch_nt128_splitk)
    // stream-K form of the eight-wave 256 x 256 kernel (gemm_w8.hip): partial accumulator slots and ticket / flag words in the CALLER's
    // workspace (eoe_gemm_args.sk_workspace), sk_rounds data-parallel rounds in front of the stream-K part, sk_flags bit 0 = stream-K first
    float* sk_part; int* sk_sync; int sk_rounds, sk_flags;
    float alpha;
    unsigned bytesA, bytesB;
    unsigned long long* stamp;     // diagnostics (EOE_GEMM_STAMP=1): per-workgroup s_memtime stamps, else NULL
    int dbg;                       // diagnostics (EOE_GEMM_DEBUG): bit 1 = no LDS-DMA inside the k-loop, bit 2 = no fragment reads there (gemm256.hip; results wrong)
    // implicit patch matrix (GATHER kernels): A = 16-bit NHWC tensor [n, gH, gW, gC]; row m = output pixel (img, ho, wo) of
    // a gHo x gWo grid, column k = (ky*gkw + kx)*gC + c -> element (img, ho*gstride - gpad + ky, wo*gstride - gpad + kx, c)
    int gH, gW, gC, gWo, gHoWo, gkw, gstride, gpad;
    // GATHER == 2 (3-channel first layer): A = physically zero-padded 16-bit [n, gH, gW, 4] image; a 128-B k-tile = 2 kernel
    // rows x 8 pixels x 4 channels, i.e. 16-B piece q of k-tile kt = pixels (2*(q&3), +1) of kernel row 2*kt + (q>>2)
    int gkstep;       // bytes between consecutive k-tiles = 2 * gW * 8
    // GATHER == 3 (narrow layers, gC in {8, 16, 32}): a 64-deep k-tile spans 64/gC taps, so every 16-B piece decodes its
    // own tap: k = kt*64 + 8*piece, tap = k >> glog2c, (ky, kx) = divmod(tap, gkw) by multiply-shift; taps >= gkh*gkw (the
    // zero padding of K up to a multiple of 64) and taps in the image's zero padding read as zeros
    int gkh, glog2c;
    unsigned gmagic;  // ceil(65536 / gkw)
};

// General (slow) epilogue: straight from the MFMA accumulator layout (lane = output row within a 16-row band, 4
// consecutive columns per 16x16 tile) -- 8/16-byte pieces, 16 rows per store instruction.  Handles any N / ldc.
template <typename T, int EPI, int NI, int MI = 4>
__device__ __forceinline__ void epilogue_generic(const GemmP& p, f32x4 (&acc)[MI][NI], int m_base, int n_base, int lane) {
    const int lr = lane & 15, lg = lane >> 4;
    const bool vec_ok = ((p.ldc & 3) == 0) && ((p.N & 3) == 0);
    f32x4 cs[NI];                      // per-ni column sums of this lane's rows (bias gradient of the producer)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) cs[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m_base + mi * 16 + lr;
        if (m >= p.M) continue;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n_base + ni * 16 + lg * 4;
            if (n >= p.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[mi][ni][r] * p.alpha;
            const int nvalid = (p.N - n) < 4 ? (p.N - n) : 4;
            if (p.bias) {
                if (vec_ok) {
                    const f32x4 bv = *(const f32x4*)(p.bias + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += bv[r];
                } else {
                    for (int r = 0; r < nvalid; ++r) v[r] += p.bias[n + r];
                }
            }
            const size_t off = (size_t)m * p.ldc + n;
            if (EPI == EOE_EPI_GELU) {
                // pre-activation (16-bit, saved for backward) and activation
                T* pre = (T*)p.aux_out;
                T* act = (T*)p.C;
                if (vec_ok) {
                    // the saved pre-activation is what backward differentiates: activate its ROUNDED value
                    u32x2 pk = pack4<T>(v[0], v[1], v[2], v[3]);
                    float pr[4];
                    unpack4<T>(pk, pr);
                    if (p.aux_out) *(u32x2*)(pre + off) = pk;
                    *(u32x2*)(act + off) = pack4<T>(quick_gelu_f(pr[0]), quick_gelu_f(pr[1]),
                                                    quick_gelu_f(pr[2]), quick_gelu_f(pr[3]));
                } else {
                    for (int r = 0; r < nvalid; ++r) {
                        T q = (T)v[r];
                        if (p.aux_out) pre[off + r] = q;
                        act[off + r] = (T)quick_gelu_f((float)q);
                    }
                }
                continue;
            }
            if (EPI == EOE_EPI_RESIDUAL) {
                const float* res = (const float*)p.aux + (size_t)m * p.ldaux + n;
                if (vec_ok && (p.ldaux & 3) == 0) {
                    f32x4 rv = *(const f32x4*)res;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += rv[r];
                } else {
                    for (int r = 0; r < nvalid; ++r) v[r] += res[r];
                }
            }
            if (EPI == EOE_EPI_GELU_BWD) {
                const T* pre = (const T*)p.aux + (size_t)m * p.ldaux + n;
                if (vec_ok && (p.ldaux & 3) == 0) {
                    float pr[4];
                    unpack4<T>(*(const u32x2*)pre, pr);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= quick_gelu_grad_f(pr[r]);
                } else {
                    for (int r = 0; r < nvalid; ++r) v[r] *= quick_gelu_grad_f((float)pre[r]);
                }
            }
            if (EPI != EOE_EPI_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) cs[ni][r] += v[r];
            }
            if (p.out_f32) {
                float* c = (float*)p.C + off;
                if (vec_ok) {
                    f32x4 o = {v[0], v[1], v[2], v[3]};
                    if (p.accumulate) {
                        f32x4 old = *(f32x4*)c;
                        o += old;
                    }
                    *(f32x4*)c = o;
                } else {
                    for (int r = 0; r < nvalid; ++r) c[r] = p.accumulate ? c[r] + v[r] : v[r];
                }
            } else {
                T* c = (T*)p.C + off;
                if (vec_ok) {
                    *(u32x2*)c = pack4<T>(v[0], v[1], v[2], v[3]);
                } else {
                    for (int r = 0; r < nvalid; ++r) c[r] = (T)v[r];
                }
            }
        }
    }
    if (EPI != EOE_EPI_GELU && p.colsum) {
        // rows of this wave's 64x64 sub-tile live on the 16 lanes sharing lane>>4: xor-reduce over lane&15
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = cs[ni][r];
                t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64);
                const int n = n_base + ni * 16 + lg * 4 + r;
                if (lr != 0 || n >= p.N) continue;
                if (p.colsum_part) {
                    if (m_base < p.M)
                        p.colsum_part[eoe_part_index(n, m_base / (16 * MI), (p.M + 16 * MI - 1) / (16 * MI), p.N, p.colsum_blocked)] = t;
                }
                else atomicAdd(p.colsum + n, t);
            }
        }
    }
}

// Fast epilogue (N % 16 == 0, 16-byte aligned rows).  The generic form above is store-ISSUE-bound: one store
// instruction writes 16 rows x 32 B, and it cost 15-21 k cycles per 256x128 tile -- as long as the 12 k-iterations of a
// K = 768 GEMM.  Here every 16-row band of the wave's 64 x (16*NI) sub-tile is transposed through a 4-KiB LDS scratch
// (the wave's OWN 4-KiB piece of the ring slot that was just consumed: only this wave's later LDS-DMA writes there, in
// program order after these reads), so that a lane holds 16 consecutive columns of one row: bias / residual / saved
// pre-activation are read, and the result is written, as 64-byte (fp32) or 32-byte (16-bit) contiguous pieces, 4 lanes
// per row = whole 128/256-byte lines per row.
// (round 2, measured and rejected: alternating the bands between two 4-KiB halves so that a band's LDS writes need not wait for the
//  previous band's reads -- layer total 494.0 / 491.2 vs 490.9 / 492.2 us on one box: the band round trips are not the epilogue's cost)
// the fast form's precondition, checked by the host for kernels that have no generic fallback compiled in
static inline bool epilogue_fast_ok(const GemmP& p) {
    return ((p.N & 15) == 0) && ((p.ldc & 7) == 0) && ((p.ldaux & 7) == 0) && ((((uintptr_t)p.C) & 15) == 0);
}

template <typename T, int EPI, int NI, int MI = 4, bool GENERIC_FALLBACK = true>
__device__ __forceinline__ void epilogue(const GemmP& p, f32x4 (&acc)[MI][NI], int m_base, int n_base, int lane, char* scr) {
    if (GENERIC_FALLBACK) {
        const bool fast = ((p.N & 15) == 0) && ((p.ldc & 7) == 0) && ((p.ldaux & 7) == 0) && ((((uintptr_t)p.C) & 15) == 0);
        if (!fast) {
            epilogue_generic<T, EPI, NI, MI>(p, acc, m_base, n_base, lane);
            return;
        }
    }
    const int lr = lane & 15, lg = lane >> 4;
    const int rrow = lane >> 2, rq = lane & 3;                 // read side: row of the band, 16-column group
    const int n = n_base + rq * 16;
    const bool col_ok = (rq < NI) && (n < p.N);
    float cs[16], cs2[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) { cs[c] = 0.f; cs2[c] = 0.f; }
    float bias[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) bias[c] = 0.f;
    if (p.bias && col_ok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bv = *(const f32x4*)(p.bias + n + q * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) bias[q * 4 + r] = bv[r];
        }
    }
    // per-band operands (the saved pre-activation of GELU' x dY: 32 B per lane and band; the fp32 residual: 64 B) are requested for ALL
    // bands before the first transposition: inside the band loop each request sat behind the band's LDS round trip and its latency was
    // exposed once per band (the GELU' epilogue cost 9.8 k cycles per tile against 5.4 k for the plain one)
    u32x4 pre_[EPI == EOE_EPI_GELU_BWD ? MI : 1][2];
    f32x4 res_[EPI == EOE_EPI_RESIDUAL ? MI : 1][4];
    if (EPI == EOE_EPI_GELU_BWD || EPI == EOE_EPI_RESIDUAL) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = m_base + mi * 16 + rrow;
            const bool ok = m < p.M && col_ok;
            if (EPI == EOE_EPI_GELU_BWD) {
                const T* pre = (const T*)p.aux + (size_t)m * p.ldaux + n;
                pre_[mi][0] = ok ? *(const u32x4*)pre : (u32x4){0u, 0u, 0u, 0u};
                pre_[mi][1] = ok ? *(const u32x4*)(pre + 8) : (u32x4){0u, 0u, 0u, 0u};
            } else {
                const float* res = (const float*)p.aux + (size_t)m * p.ldaux + n;
#pragma unroll
                for (int q = 0; q < 4; ++q) res_[mi][q] = ok ? *(const f32x4*)(res + q * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        // accumulator layout -> LDS [16 rows][64 cols] fp32, 16-B chunk index XOR row (bank spread)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) *(f32x4*)(scr + lr * 256 + (((ni * 4 + lg) ^ lr) << 4)) = acc[mi][ni];
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 t = *(const f32x4*)(scr + rrow * 256 + (((rq * 4 + q) ^ rrow) << 4));
#pragma unroll
            for (int r = 0; r < 4; ++r) v[q * 4 + r] = t[r] * p.alpha + bias[q * 4 + r];
        }
        const int m = m_base + mi * 16 + rrow;
        if (m >= p.M || !col_ok) continue;
        const size_t off = (size_t)m * p.ldc + n;
        if (EPI == EOE_EPI_GELU) {
            T* pre = (T*)p.aux_out + off;
            T* act = (T*)p.C + off;
            float pr[16], ac[16];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const u32x4 pk = pack8<T>(v + 8 * h);          // the saved pre-activation is what backward differentiates:
                unpack8<T>(pk, pr + 8 * h);                    // activate its ROUNDED value
                if (p.aux_out) *(u32x4*)(pre + 8 * h) = pk;    // (forward-only callers do not keep it: same activation either way)
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) ac[c] = quick_gelu_f(pr[c]);
            *(u32x4*)(act) = pack8<T>(ac);
            *(u32x4*)(act + 8) = pack8<T>(ac + 8);
            continue;
        }
        if (EPI == EOE_EPI_RESIDUAL) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 rv = res_[EPI == EOE_EPI_RESIDUAL ? mi : 0][q];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[q * 4 + r] += rv[r];
            }
        }
        if (EPI == EOE_EPI_GELU_BWD) {
            float pr[16];
            unpack8<T>(pre_[EPI == EOE_EPI_GELU_BWD ? mi : 0][0], pr);
            unpack8<T>(pre_[EPI == EOE_EPI_GELU_BWD ? mi : 0][1], pr + 8);
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] *= quick_gelu_grad_f(pr[c]);
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) cs[c] += v[c];
        if (p.colsum_sq) {
#pragma unroll
            for (int c = 0; c < 16; ++c) cs2[c] += v[c] * v[c];
        }
        if (p.out_f32) {
            float* c = (float*)p.C + off;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 o = {v[q * 4], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]};
                if (p.accumulate) o += *(const f32x4*)(c + q * 4);
                *(f32x4*)(c + q * 4) = o;
            }
        } else {
            T* c = (T*)p.C + off;
            *(u32x4*)(c) = pack8<T>(v);
            *(u32x4*)(c + 8) = pack8<T>(v + 8);
        }
    }
    if (EPI != EOE_EPI_GELU && p.colsum_part) {
        // partial-row form: the lanes' 16-column sums go through the wave's LDS scratch ([16 rows][64 cols], the band
        // transposition's swizzle), lane j then adds column j over the 16 rows and stores ONE float of this wave's partial row
        // (64 cross-lane shuffles + 64 scattered stores in the atomic form below cost more than a separate pass over C)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *(f32x4*)(scr + rrow * 256 + (((rq * 4 + q) ^ rrow) << 4)) = (f32x4){cs[q * 4], cs[q * 4 + 1], cs[q * 4 + 2], cs[q * 4 + 3]};
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += *(const float*)(scr + r * 256 + ((((lane >> 2) ^ r) << 4) | ((lane & 3) << 2)));
        const int nn = n_base + lane;
        const bool st_ok = lane < 16 * NI && nn < p.N && m_base < p.M;
        if (!p.colsum_sq) {
            if (st_ok) p.colsum_part[eoe_part_index(nn, m_base / (16 * MI), (p.M + 16 * MI - 1) / (16 * MI), p.N, p.colsum_blocked)] = t;
        } else {
            // BatchNorm statistics: row [2][N] = (sum, sum of squares) of this wave's 16*MI output rows
            if (st_ok) p.colsum_part[((size_t)(m_base / (16 * MI)) * 2) * p.N + nn] = t;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(f32x4*)(scr + rrow * 256 + (((rq * 4 + q) ^ rrow) << 4)) = (f32x4){cs2[q * 4], cs2[q * 4 + 1], cs2[q * 4 + 2], cs2[q * 4 + 3]};
            float t2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) t2 += *(const float*)(scr + r * 256 + ((((lane >> 2) ^ r) << 4) | ((lane & 3) << 2)));
            if (st_ok) p.colsum_part[((size_t)(m_base / (16 * MI)) * 2 + 1) * p.N + nn] = t2;
        }
    } else if (EPI != EOE_EPI_GELU && p.colsum) {
        // the 16 rows of a band sit on lanes with equal lane&3: xor-reduce over lane>>2, then 4 lanes x 16 atomics
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float t = cs[c];
            t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
            if (rrow == 0 && col_ok) atomicAdd(p.colsum + n + c, t);
        }
    }
}

// Direct epilogue (round 3; gemm256.hip).  The LDS-transposed form above costs a workgroup that has nothing else resident on its CU
// an LDS round trip per 16-row band, a ring slot to transpose in (the k-tile that would have been staged there is deferred past the
// epilogue) and a full drain of the memory counters before the next tile.  Here the B tile's rows are PERMUTED at staging so that a
// lane's 16 accumulator values of a band (tiles ni = 0..3 x r = 0..3; lane = (lr = lane & 15: output row, lg = lane >> 4)) are runs of
// consecutive output columns, and so that the four lanes lg = 0..3 of a row write ADJACENT 16-byte pieces with each store instruction
// (a store instruction then covers 16 rows x 64 contiguous bytes; 64 scattered 16-byte pieces per instruction ran the epilogue at
// ~14 B/clk per CU):
//   16-bit C:  value c = 4*ni + r  ->  column 32*(c >> 3) + 8*lg + (c & 7)      two 16-byte stores per lane and band
//   fp32 C:    value c = 4*ni + r  ->  column 16*ni + 4*lg + r                  the MFMA's own layout: four 16-byte stores
// eoe_direct_row() is the matching row permutation (LDS row 16*ni + i of a 64-row group <- weight row ...); no LDS is used here.  Every
// memory operation is a buffer builtin the compiler counts itself (its vmcnt(N) for a load leaves exactly the younger stores in flight
// -- the LDS-DMA pieces, which it cannot see, are all OLDER than anything issued here); per-band operands (the fp32 residual, the saved
// pre-activation) are requested one band ahead, in source order.  Every wave issues the same number of stores per tile whatever its
// position (epilogue_direct_stores; rows / columns outside C carry the out-of-range offset): the caller's first wait of the next tile
// counts them.
__host__ __device__ __forceinline__ int eoe_direct_row(int x /* LDS row within its 64-row group: 16*ni + i */, int out_f32) {
    const int ni = x >> 4, i = x & 15;
    return out_f32 ? x : ((ni >> 1) * 32 + (i >> 2) * 8 + (ni & 1) * 4 + (i & 3));
}
__device__ __forceinline__ void store_b128(__amdgpu_buffer_rsrc_t r, unsigned voff, u32x4 d) {
    __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)voff, 0, 0);
}
// stores one wave issues per (16*MI) x 128 tile (two 64-column halves)
__host__ __device__ __forceinline__ int epilogue_direct_stores(const GemmP& p, int epi, int MI) {
    const int per_band = (epi == EOE_EPI_GELU) ? (p.aux_out ? 4 : 2) : (p.out_f32 ? 4 : 2);
    return 2 * MI * per_band + ((epi != EOE_EPI_GELU && p.colsum_part) ? 8 : 0);
}
// the combinations the direct form implements (the launcher sends anything else to the other kernels)
static inline bool epilogue_direct_ok(const GemmP& p, int epi) {
    if (!epilogue_fast_ok(p) || p.colsum_sq) return false;
    if (epi == EOE_EPI_GELU || epi == EOE_EPI_GELU_BWD) return !p.out_f32;
    if (epi == EOE_EPI_RESIDUAL) return p.out_f32 != 0;
    return true;
}

template <typename T, int EPI, int MI>
__device__ __forceinline__ void epilogue_direct(const GemmP& p, f32x4 (&acc)[2][MI][4], int m_base, int n_base, int lane,
                                                __amdgpu_buffer_rsrc_t rc, __amdgpu_buffer_rsrc_t rpre, __amdgpu_buffer_rsrc_t raux,
                                                __amdgpu_buffer_rsrc_t rpart) {
    const int lr = lane & 15, lg = lane >> 4;
    constexpr int NB = 2 * MI;                                 // bands: b = half * MI + mi
    constexpr bool F32 = (EPI == EOE_EPI_RESIDUAL);            // RESIDUAL is fp32 in, fp32 out; GELU / GELU_BWD are 16-bit out
    const bool f32 = F32 || (EPI == EOE_EPI_NONE && p.out_f32);
    // column of value c (relative to the half's first column) and the start of run q: fp32 -> 4 runs of 4, 16-bit -> 2 runs of 8
    auto col_of = [&](int c) -> int { return f32 ? (c >> 2) * 16 + lg * 4 + (c & 3) : (c >> 3) * 32 + lg * 8 + (c & 7); };
    constexpr int NL = (EPI == EOE_EPI_RESIDUAL) ? 4 : (EPI == EOE_EPI_GELU_BWD ? 2 : 0);      // loads per band
    float bias[2][16];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int c = 0; c < 16; ++c) bias[h][c] = 0.f;
        if (p.bias) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n_base + h * 64 + col_of(q * 4);
                if (n < p.N) {
                    const f32x4 bv = *(const f32x4*)(p.bias + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) bias[h][q * 4 + r] = bv[r];
                }
            }
        }
    }
    // byte offset of run q of the lane's values of band b in a matrix of leading dimension ld, element size es; rows / runs outside the
    // matrix: the out-of-range offset (N % 16 == 0: a run of 4 or 8 columns is all in or all out).  Runs are 64 bytes apart in both
    // layouts (fp32: 16 columns x 4 B; 16-bit: 32 columns x 2 B)
    auto run_off = [&](int b, int q, unsigned ld, unsigned es) -> unsigned {
        const int h = b / MI, mi = b - h * MI;
        const int m = m_base + mi * 16 + lr, n = n_base + h * 64 + (f32 ? q * 16 + lg * 4 : q * 32 + lg * 8);
        return (m < p.M && n < p.N) ? (unsigned)(((size_t)m * ld + n) * es) : EOE_OOB;
    };
    u32x4 ld[2][NL > 0 ? NL : 1];
    auto issue_loads = [&](int b, int buf) {
#pragma unroll
        for (int q = 0; q < NL; ++q)
            ld[buf][q] = __builtin_amdgcn_raw_buffer_load_b128(raux, (int)run_off(b, q, (unsigned)p.ldaux, F32 ? 4u : 2u), 0, 0);
    };
    if (NL > 0) issue_loads(0, 0);
    float cs[2][16];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int c = 0; c < 16; ++c) cs[h][c] = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int h = b / MI, mi = b - h * MI, buf = b & 1;
        const int m = m_base + mi * 16 + lr;
        const bool row_ok = m < p.M;
        if (NL > 0 && b + 1 < NB) issue_loads(b + 1, buf ^ 1);
        float v[16];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[ni * 4 + r] = acc[h][mi][ni][r] * p.alpha + bias[h][ni * 4 + r];
        unsigned off[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) off[q] = run_off(b, q, (unsigned)p.ldc, f32 ? 4u : 2u);      // (16-bit layout: runs 0 and 1 only)
        if (EPI == EOE_EPI_GELU) {
            float pr[16], ac[16];
            u32x4 pk[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                pk[hh] = pack8<T>(v + 8 * hh);                 // the saved pre-activation is what backward differentiates:
                unpack8<T>(pk[hh], pr + 8 * hh);               // activate its ROUNDED value
            }
            if (p.aux_out) {
                store_b128(rpre, off[0], pk[0]);
                store_b128(rpre, off[1], pk[1]);
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) ac[c] = quick_gelu_f(pr[c]);
            store_b128(rc, off[0], pack8<T>(ac));
            store_b128(rc, off[1], pack8<T>(ac + 8));
            continue;
        }
        if (EPI == EOE_EPI_RESIDUAL) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // (whole-vector bit cast: __builtin_bit_cast of ONE ELEMENT of an ext_vector lvalue reads element 0 whatever the index)
                const f32x4 rv = __builtin_bit_cast(f32x4, ld[buf][NL > 3 ? q : 0]);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[q * 4 + r] += rv[r];
            }
        }
        if (EPI == EOE_EPI_GELU_BWD) {
            float pr[16];
            unpack8<T>(ld[buf][0], pr);
            unpack8<T>(ld[buf][NL > 1 ? 1 : 0], pr + 8);
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] *= quick_gelu_grad_f(pr[c]);
        }
        if (row_ok) {                                          // (columns outside N: their sums are never stored)
#pragma unroll
            for (int c = 0; c < 16; ++c) cs[h][c] += v[c];
        }
        if (f32) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 o = {v[q * 4], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]};
                if (p.accumulate)                              // (not on the ViT path: the load waits for everything older)
                    o += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rc, (int)off[q], 0, 0));
                store_b128(rc, off[q], __builtin_bit_cast(u32x4, o));
            }
        } else {
            store_b128(rc, off[0], pack8<T>(v));
            store_b128(rc, off[1], pack8<T>(v + 8));
        }
    }
    if (EPI != EOE_EPI_GELU && (p.colsum_part || p.colsum)) {
        // column sums of this wave's 16*MI rows: the 16 lanes lr = 0..15 of a column group are one DPP row -- xor-reduce in a fixed
        // order, lane lr == 0 stores its runs of the wave's partial row (always 4 stores per lane and half, out-of-range for the other
        // lanes; 16-bit layout: 2 runs of 8 columns = 2 x 32 B, fp32 layout: 4 runs of 4 = 4 x 16 B)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                float t = cs[h][c];
                t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64);
                cs[h][c] = t;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n_base + h * 64 + col_of(q * 4);
                const bool st_ok = lr == 0 && n < p.N && m_base < p.M;
                const f32x4 t4 = {cs[h][q * 4], cs[h][q * 4 + 1], cs[h][q * 4 + 2], cs[h][q * 4 + 3]};
                if (p.colsum_part) {
                    const unsigned po = st_ok ? (unsigned)(eoe_part_index(n, m_base / (16 * MI), (p.M + 16 * MI - 1) / (16 * MI), p.N, p.colsum_blocked) * 4) : EOE_OOB;
                    store_b128(rpart, po, __builtin_bit_cast(u32x4, t4));
                } else if (st_ok) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) atomicAdd(p.colsum + n + r, t4[r]);
                }
            }
        }
    }
}

// the epilogue-only arguments are re-read from the kernarg segment at the point of use (behind an opaque pointer), so
// that they do not occupy ~40 SGPRs across the k-loop (the loop otherwise carries SGPR spills and extra waits)
__device__ __forceinline__ void load_epilogue_args(GemmP& ep, const GemmP& p) {
#if defined(__HIP_DEVICE_COMPILE__)
    // read through a CONSTANT-address-space pointer: scalar loads (s_load_dwordx16: SMEM, counted by lgkmcnt, a couple of hundred
    // cycles).  Through a generic pointer (round 1-2: memcpy from `(const void*)kp`) the struct came back by per-lane vector loads,
    // i.e. behind every LDS-DMA piece still in flight (vmcnt is in issue order): one to two thousand cycles at the top of every epilogue
    typedef const __attribute__((address_space(4))) GemmP* kernarg_ptr;
    const __attribute__((address_space(4))) char* kp =
        (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    ep = *(kernarg_ptr)kp;
#else
    ep = p;
#endif
}

#define EOE_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define EOE_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

static inline int num_cus() {
    static const int ncu = [] { int d = 0, n = 256; hipDeviceProp_t pr; if (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess) n = pr.multiProcessorCount; return n; }();
    return ncu;
}

static int finish_colsum(const GemmP& p, int epi, int mi, hipStream_t s) {
    if (!p.colsum_part || !p.colsum || p.colsum_sq || epi == EOE_EPI_GELU) return 0;
    return eoe_finish_reduce(p.colsum_part, cdiv(p.M, 16 * mi), p.N, p.N, p.colsum, nullptr, nullptr, p.colsum_blocked, s);
}

}  // namespace
