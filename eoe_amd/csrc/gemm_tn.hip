// Weight-gradient GEMM for gfx950:  C[M,N] (fp32) = A[T,M]^T . B[T,N]   (A = dY, B = X, reduction over T tokens).
//
// Design for the wgrad regime (few output tiles, very long reduction: ViT-B/32 has M,N in {768,2304,3072} and
// T = 12800): 256x128 output tiles, 8 wavefronts (4x2, 64x64 each, two per SIMD so they cover each other's LDS
// and barrier stalls), ONE workgroup per CU, and a GROUPED launch: up to EOE_TN_MAX_GROUP independent problems
// (the four weight gradients of a transformer block: 54 + 18 + 72 + 72 = 216 tiles) share one grid, so the chip
// is filled by plain data parallelism over output tiles -- every tile runs the whole reduction, there is no
// split-K, no atomics and no zero-init for that case, and the result is bitwise reproducible.  Small stand-alone
// problems can still split T across workgroups (fp32 atomic accumulation).
//
// Operands are staged global -> LDS by bounds-checked LDS-DMA (buffer_load ... lds, 16 B per lane; out-of-range
// lanes read 0 = zero padding of the ragged T / M / N tails).  The reduction dimension is the row (strided)
// dimension of both operands, so MFMA fragments are read with the hardware-transposing ds_read_b64_tr_b16 from
// LDS images [64 t][cols] whose 32-B granules are XOR-swizzled by (t&3)|((t>>3)&1)<<2 (applied on the SOURCE
// address: LDS-DMA writes linearly), conflict-free for both 4-row halves of a k-step.
// Workgroup ids are remapped so that each XCD (private L2) gets a contiguous run of tiles, which share A/B panels.
#include "common.h"
#include <map>
#include <mutex>
#include <utility>
#include <stdlib.h>

int g_tn_flags = 0;     // option "tn_flags": reserved for A/B experiments (none active)

namespace {

constexpr int BM = 256, BN = 128, BK = 64;
constexpr int A_BYTES = BK * BM * 2;              // 32 KiB
constexpr int B_BYTES = BK * BN * 2;              // 16 KiB
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;    // 48 KiB
constexpr int NSTAGE = 3;
constexpr int SMEM_BYTES = NSTAGE * STAGE_BYTES;  // 144 KiB: one 8-wave workgroup per CU
constexpr int NWAVES = 8;

struct TnProblem {
    const void* A; const void* B; float* C;
    int M, N, lda, ldb, ldc, tiles_n, tile_start;
    unsigned bytesA, bytesB;
};
struct TnGroup {
    TnProblem p[EOE_TN_MAX_GROUP];
    int count, T, total_tiles, splits, t_per_split, accumulate;
    float alpha;
    // GATHER (single-problem groups): geometry of the implicit patch matrix, multiply-shift constants for / (Ho*Wo), / Wo
    int gH, gW, gC, gWo, gHoWo, gkw, gstride, gpad;
    unsigned long long mHoWo, mWo;
    float* part;      // split reduction without atomics (optional workspace): split s of problem i writes a dense [M][N] block
    long long part_stride[EOE_TN_MAX_GROUP];   // element offset of problem i's partial block in `part`
    // stream-K (sk_wgs > 0): see gemm_tn_grouped_kernel
    int sk_wgs;
    float* sk_part;   // [sk_wgs][BM*BN] partial accumulators of the ranges that do not end their tile
    int* sk_flags;    // [sk_wgs] = sk_epoch once that workgroup's partial of THIS launch is visible (never reset: the next launch waits
                      // for its own epoch, so a time-out or an aborted launch cannot leave a flag that a later launch would trust)
    int sk_epoch;     // per-(device, stream) launch counter, never 0
};

__device__ __forceinline__ int swz(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

template <typename T>
__device__ __forceinline__ typename T16<T>::v8 tr_frag(const char* base, int off_lo, int row_bytes) {
    i16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4v*)(base + off_lo));
    i16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4v*)(base + off_lo + 4 * row_bytes));
    i16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(typename T16<T>::v8, r);
}

// Fragment read without the compiler's LDS-DMA bookkeeping.  With the ds_read_tr intrinsic the waitcnt pass cannot tell the
// transposed LDS reads from the ring slots the in-flight LDS-DMA is filling and puts `s_waitcnt vmcnt(0)` in front of them:
// the k-tile that was just requested (two tiles ahead) must land before this iteration may read the current one -- the
// whole HBM/L2 latency is exposed every k-tile (the NT kernel's plain ds_read_b128 do not get that wait).  As inline asm
// the reads are opaque; the kernel's own counted vmcnt + barrier protocol already orders them against the DMA, and their
// completion is awaited with explicit lgkmcnt(0) before the MFMAs that consume them.
template <typename T, int HI_OFF>
__device__ __forceinline__ typename T16<T>::v8 tr_frag_asm(unsigned lds_addr) {
    i16x4v lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(lds_addr) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(lds_addr), "n"(HI_OFF) : "memory");
    i16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(typename T16<T>::v8, r);
}

// (An L2-prefetcher ninth wavefront touching the operand lines of k-tile kt+6 was measured and rejected: 356 vs 304 us
//  stand-alone, 4.58 vs 3.9 ms in the step.)
// GATHER: the A operand is a convolution's patch matrix [T = n*Ho*Wo pixels, M = kh*kw*C] that is never materialised:
// each lane's column (tap, channel) is fixed for the whole kernel, its row (pixel) changes every k-tile and is decoded
// with two multiply-shift divisions; pieces in the zero padding get the out-of-range offset (zero fill).
__device__ __forceinline__ int div_magic(int x, unsigned long long m) { return (int)(((unsigned long long)(unsigned)x * m) >> 40); }

template <typename T, int GATHER, int RD = 1>
__global__ __launch_bounds__(512, 2) void gemm_tn_grouped_kernel(TnGroup g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Stream-K (g.sk_wgs > 0; round 2).  A group of 216 equal tiles on 256 CUs leaves 16 % of the chip idle whatever the tile shape (one
    // round, the largest tile sets the time).  Instead the (tile, k-tile) space of the whole group is cut into gridDim.x = #CUs equal
    // ranges, one per workgroup.  A range is shorter than a tile's reduction, so it is at most the END of one tile followed by the START
    // of the next: up to two segments.  A segment that does not end its tile stores its accumulators as a partial (128 KiB, fragment
    // order) and publishes a flag; the workgroup whose segment ends the tile adds the one or two partials in a fixed order and runs the
    // epilogue.  No atomics on data, no second kernel, bitwise reproducible; extra traffic 2 x 128 KiB per split point.  The grid is
    // exactly one workgroup per CU (144 KiB of LDS each), so every range is resident while its successor waits for it; the wait is
    // bounded all the same and a time-out poisons the tile with NaN instead of hanging the GPU or passing silently.  Flags carry the
    // launch's epoch (TnGroup::sk_epoch), so they are never reset and nothing an earlier launch left behind can be mistaken for a
    // partial of this one.
    const int split = g.sk_wgs ? 0 : (int)blockIdx.x / g.total_tiles;
    const int nk_tile = (g.T + BK - 1) / BK;
    const long long sk_total = (long long)g.total_tiles * nk_tile;
    int seg_gt[2], seg_tb[2], seg_te[2], nseg = 1;
    // range index: XCD-remapped like the tiles of the plain launch, so the 32 workgroups of an XCD walk neighbouring tiles (shared A / B
    // slabs stay in its L2: without this every CU streams its own 10 MB of operands, 2.5 GB per launch) and a range's predecessor sits
    // on the same XCD (its partial is read at the same-XCD rate) except at the 7 chunk starts
    const int rid = g.sk_wgs ? xcd_remap((int)blockIdx.x, g.sk_wgs) : 0;
    if (g.sk_wgs) {
        const long long u0 = sk_total * (long long)rid / g.sk_wgs, u1 = sk_total * ((long long)rid + 1) / g.sk_wgs;
        const int t0 = (int)(u0 / nk_tile);
        const long long e0 = (long long)(t0 + 1) * nk_tile;
        seg_gt[0] = t0;
        seg_tb[0] = (int)(u0 - (long long)t0 * nk_tile) * BK;
        seg_te[0] = (u1 < e0) ? (int)(u1 - (long long)t0 * nk_tile) * BK : g.T;
        seg_gt[1] = t0 + 1; seg_tb[1] = 0; seg_te[1] = 0;
        if (u1 > e0) { nseg = 2; seg_te[1] = min(g.T, (int)(u1 - e0) * BK); }
    } else {
        seg_gt[0] = xcd_remap((int)blockIdx.x % g.total_tiles, g.total_tiles);
        seg_tb[0] = split * g.t_per_split;
        seg_te[0] = min(g.T, seg_tb[0] + g.t_per_split);
        seg_gt[1] = 0; seg_tb[1] = 0; seg_te[1] = 0;
    }
    // the segment that STARTS a tile goes first: its partial is what the next workgroup's fix-up waits for (the other order chains every
    // workgroup behind its predecessor: 7 ms instead of 0.2)
#pragma nounroll
  for (int si = 0; si < nseg; ++si) {
    const int seg = nseg - 1 - si;
    if (si) __syncthreads();                     // the previous segment's last ring reads are behind every wave
    const int gt = seg_gt[seg];
    int pi = 0;
#pragma unroll
    for (int i = 1; i < EOE_TN_MAX_GROUP; ++i)
        if (i < g.count && gt >= g.p[i].tile_start) pi = i;
    const TnProblem& P = g.p[pi];
    const int lt = gt - P.tile_start;
    const int m0 = (lt / P.tiles_n) * BM, n0 = (lt % P.tiles_n) * BN;
    const int t_begin = seg_tb[seg];
    const int t_end = seg_te[seg];

    __amdgpu_buffer_rsrc_t ra = make_rsrc(P.A, P.bytesA);
    __amdgpu_buffer_rsrc_t rb = make_rsrc(P.B, P.bytesB);
    const int lda = P.lda, ldb = P.ldb;

    // staging.  A image: 64 rows x 512 B -> 32 wave-loads (2 rows each), 4 per wave;
    //           B image: 64 rows x 256 B -> 16 wave-loads (4 rows each), 2 per wave.
    int rowA[4], rowB[2];
    unsigned colA[4], colB[2];
    int gky[4], gkx[4];                                              // GATHER: tap offset (ky - pad, kx - pad) of the lane's column
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int wl = wave * 4 + j;
        const int row = wl * 2 + (lane >> 5), s = lane & 31;
        const int c16 = (((s >> 1) ^ swz(row)) << 1) | (s & 1);      // swz touches the low 3 granule bits only
        rowA[j] = row;
        const int c = m0 + c16 * 8;
        if (GATHER == 2) {
            // packed 3-channel first layer: column = ky*32 + px*4 + ch over a zero-padded [n, gH, gW, 4] image
            const int q8 = c >> 3;                                   // 16-B piece: kernel row q8 >> 2, pixels 2*(q8 & 3), +1
            colA[j] = (c < P.M) ? (unsigned)((((q8 >> 2) * g.gW + 2 * (q8 & 3)) * 4) * 2) : EOE_OOB;
        } else if (GATHER == 1) {
            const int tap = c / g.gC, ch = c - tap * g.gC;
            const int ky = tap / g.gkw;
            gky[j] = ky - g.gpad;
            gkx[j] = tap - ky * g.gkw - g.gpad;
            colA[j] = (c < P.M) ? (unsigned)(ch * 2) : EOE_OOB;
        } else {
            colA[j] = (c < P.M) ? (unsigned)(c * 2) : EOE_OOB;
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int wl = wave * 2 + j;
        const int row = wl * 4 + (lane >> 4), s = lane & 15;
        const int c16 = (((s >> 1) ^ swz(row)) << 1) | (s & 1);
        rowB[j] = row;
        const int c = n0 + c16 * 8;
        colB[j] = (c < P.N) ? (unsigned)(c * 2) : EOE_OOB;
    }
    auto stage = [&](int buf, int t0) {
        char* sa = smem + buf * STAGE_BYTES;
        char* sb = sa + A_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = t0 + rowA[j];
            unsigned o = EOE_OOB;
            if (GATHER == 2) {
                if (t < t_end && colA[j] != EOE_OOB) {
                    const int img = div_magic(t, g.mHoWo), rem = t - img * g.gHoWo;
                    const int ho = div_magic(rem, g.mWo), wo = rem - ho * g.gWo;
                    o = (unsigned)((((img * g.gH + ho * g.gstride) * g.gW + wo * g.gstride) * 4) * 2) + colA[j];
                }
            } else if (GATHER == 1) {
                if (t < t_end && colA[j] != EOE_OOB) {
                    const int img = div_magic(t, g.mHoWo), rem = t - img * g.gHoWo;
                    const int ho = div_magic(rem, g.mWo), wo = rem - ho * g.gWo;
                    const int hh = ho * g.gstride + gky[j], ww = wo * g.gstride + gkx[j];
                    if ((unsigned)hh < (unsigned)g.gH && (unsigned)ww < (unsigned)g.gW)
                        o = (unsigned)(((img * g.gH + hh) * g.gW + ww) * g.gC * 2) + colA[j];
                }
            } else {
                o = (t < t_end && colA[j] != EOE_OOB) ? (unsigned)((size_t)t * lda * 2) + colA[j] : EOE_OOB;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(sa + (wave * 4 + j) * 1024), 16, o, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int t = t0 + rowB[j];
            const unsigned o = (t < t_end && colB[j] != EOE_OOB) ? (unsigned)((size_t)t * ldb * 2) + colB[j] : EOE_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_t*)(sb + (wave * 2 + j) * 1024), 16, o, 0, 0, 0);
        }
    };

    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    const int lr = lane & 15, lg = lane >> 4;
    // transposed read: lane 4q+p of a 16-lane group addresses row q of the group's 4-row block, columns 4p..4p+3
    const int q = lr >> 2, pp = lr & 3;
    const int f = q | ((lg & 1) << 2);                 // swz(row) for row = 32*ks + 8*lg + 4*h + q
    const int rowsel = 8 * lg + q;
    int offA[4], offB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        offA[i] = rowsel * (BM * 2) + ((((wm0 >> 4) + i) ^ f) << 5) + pp * 8;
        offB[i] = rowsel * (BN * 2) + ((((wn0 >> 4) + i) ^ f) << 5) + pp * 8;
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // main loop: 3-stage LDS ring, LDS-DMA two k-tiles ahead behind a counted vmcnt, one raw s_barrier per k-tile,
    // fragments double-buffered in registers (same schedule and hazard argument as gemm_nt_kernel in gemm.hip)
    typedef typename T16<T>::v8 V8;
#define EOE_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
    // RD = 1: asm reads (see tr_frag_asm); the ring starts at LDS address 0 (the kernel's only LDS object)
#define EOE_READ(XA, WB, base, ks)                                                         \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                        \
        if (RD) {                                                                          \
            const unsigned b_ = (unsigned)((base) - smem);                                 \
            XA[i] = tr_frag_asm<T, 4 * BM * 2>(b_ + (ks) * 32 * (BM * 2) + offA[i]);       \
            WB[i] = tr_frag_asm<T, 4 * BN * 2>(b_ + A_BYTES + (ks) * 32 * (BN * 2) + offB[i]); \
        } else {                                                                           \
            XA[i] = tr_frag<T>((base), (ks) * 32 * (BM * 2) + offA[i], BM * 2);            \
            WB[i] = tr_frag<T>((base) + A_BYTES, (ks) * 32 * (BN * 2) + offB[i], BN * 2);  \
        }                                                                                  \
    }
    // the asm reads complete asynchronously and the compiler does not know: wait for them IN an asm statement that also
    // "rewrites" the fragment registers, so that no MFMA consuming them can be scheduled above the wait
#define EOE_LANDED(XA, WB)                                                                                       \
    do {                                                                                                         \
        if (RD) asm volatile("s_waitcnt lgkmcnt(0)"                                                              \
                             : "+v"(XA[0]), "+v"(XA[1]), "+v"(XA[2]), "+v"(XA[3]), "+v"(WB[0]), "+v"(WB[1]), "+v"(WB[2]), \
                               "+v"(WB[3])                                                                       \
                             :: "memory");                                                                       \
    } while (0)
    // D = (B-tile fragment as the A operand) x (A-tile fragment as the B operand): the lane holds 4 consecutive n
    // (output columns) of one output row m
#define EOE_MFMA(XA, WB)                                                                   \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                       \
        _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = T16<T>::mfma16(WB[ni], XA[mi], acc[mi][ni]);

    const int nk = (t_end - t_begin + BK - 1) / BK;
    if (nk > 0) {
        V8 xa0[4], wb0[4], xa1[4], wb1[4];
        stage(0, t_begin);
        if (nk > 1) {
            stage(1, t_begin + BK);
            EOE_WAIT_VM(6);
        } else {
            EOE_WAIT_VM(0);
        }
        __builtin_amdgcn_s_barrier();
        EOE_READ(xa0, wb0, smem, 0);
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const int nxt = (cur == NSTAGE - 1) ? 0 : cur + 1;
            const int nx2 = (nxt == NSTAGE - 1) ? 0 : nxt + 1;
            const char* sc = smem + cur * STAGE_BYTES;
            if (kt + 2 < nk) stage(nx2, t_begin + (kt + 2) * BK);
            EOE_LANDED(xa0, wb0);               // F0 (read during the previous MFMA half) has landed
            EOE_READ(xa1, wb1, sc, 1);
            EOE_MFMA(xa0, wb0);
            if (kt + 2 < nk) { EOE_WAIT_VM(6); } else { EOE_WAIT_VM(0); }
            if (RD) { EOE_LANDED(xa1, wb1); } else { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
            __builtin_amdgcn_s_barrier();
            const char* sn = smem + nxt * STAGE_BYTES;
            EOE_READ(xa0, wb0, sn, 0);          // unconditional: the last one reads a stale slot and is discarded
            EOE_MFMA(xa1, wb1);
            cur = nxt;
        }
        // the last (stale, discarded) asm read is still in flight and the compiler does not know: without this wait it may
        // hand those registers to the epilogue while the LDS unit is about to write them
        EOE_LANDED(xa0, wb0);
    }
#undef EOE_READ
#undef EOE_MFMA
#undef EOE_LANDED

    if (g.sk_wgs) {
        const long long W = g.sk_wgs;
        // (no second __shared__ object: the asm fragment reads assume the ring at LDS address 0; the time-out word borrows the first
        //  bytes of the ring, idle by now -- every DMA piece and fragment read of this workgroup's last segment has landed)
        volatile int* sk_timeout = (volatile int*)smem;
        if (t_end < g.T) {                       // contributor: partial accumulators in fragment order (16 B per lane, coalesced)
            float* mine = g.sk_part + (size_t)rid * (BM * BN);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    *(f32x4*)(mine + ((size_t)((wave * 16 + mi * 4 + ni) * 64 + lane)) * 4) = acc[mi][ni];
            // publish (cdna_hip_programming.md, Guideline 16): every storing wave drains its stores, the workgroup meets, ONE lane
            // releases at agent scope and raises the flag with a relaxed agent-scope store (a __threadfence() per thread here
            // writes the XCD's L2 back 512 times per workgroup: 360 us instead of 180)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(g.sk_flags + rid, g.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            continue;
        }
        if (t_begin > 0) {                       // owner of a split tile: add the earlier ranges' partials, nearest first
            const long long tile_u0 = (long long)gt * nk_tile;
            for (int cc = rid - 1; cc >= 0; --cc) {
                const long long cu0 = sk_total * cc / W, cu1 = sk_total * (cc + 1) / W;
                if (cu1 <= tile_u0) break;
                if (tid == 0) {
                    // relaxed polls of the one word (an acquiring poll would invalidate the caches on every iteration), bounded: a
                    // protocol error must not hang the GPU (the result is then wrong and the tests say so); ONE acquire after the match.
                    // The flag carries this launch's epoch: whatever an earlier launch left there (a contributor that arrived after
                    // its consumer had timed out, a launch that was aborted) is not mistaken for this launch's partial.
                    int spins = 0;
                    while (__hip_atomic_load(g.sk_flags + cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != g.sk_epoch && ++spins < (1 << 22))
                        __builtin_amdgcn_s_sleep(8);
                    *sk_timeout = spins >= (1 << 22) ? 1 : 0;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();
                const bool timed_out = *sk_timeout != 0;
                __syncthreads();                 // every thread has read the word before thread 0 may rewrite it for the next partial
                if (timed_out) {                 // loud, not silent: the tile becomes NaN (no flag to clean up: epochs)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f32x4){NAN, NAN, NAN, NAN};
                    break;
                }
                const float* theirs = g.sk_part + (size_t)cc * (BM * BN);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        acc[mi][ni] += *(const f32x4*)(theirs + ((size_t)((wave * 16 + mi * 4 + ni) * 64 + lane)) * 4);
                if (cu0 <= tile_u0) break;
            }
        }
    }
    if (g.part) {
        // per-split partial result [M][N] (dense), summed by tn_reduce_kernel
        float* base = g.part + g.part_stride[pi] + (size_t)split * P.M * P.N;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = m0 + wm0 + mi * 16 + lr;
            if (m >= P.M) continue;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = n0 + wn0 + ni * 16 + lg * 4;
                if (n >= P.N) continue;            // N % 8 == 0: a 4-column piece is entirely in or out
                *(f32x4*)(base + (size_t)m * P.N + n) = acc[mi][ni];
            }
        }
        continue;
    }
    const bool atomic = g.splits > 1;
    const bool vec = ((P.ldc & 3) == 0) && ((P.N & 3) == 0) && !atomic;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm0 + mi * 16 + lr;
        if (m >= P.M) continue;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wn0 + ni * 16 + lg * 4;
            if (n >= P.N) continue;
            float* c = P.C + (size_t)m * P.ldc + n;
            f32x4 v = acc[mi][ni] * g.alpha;
            if (vec) {
                if (g.accumulate) v += *(const f32x4*)c;
                *(f32x4*)c = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (n + r >= P.N) break;
                    if (atomic) atomicAdd(c + r, v[r]);
                    else c[r] = g.accumulate ? c[r] + v[r] : v[r];
                }
            }
        }
    }
  }   // segments
}

// C[m][n] = alpha * sum_s part[s][m][n] (+ C if accumulate).  A workgroup owns 64 column quads; its 4 wavefronts take the
// splits s = w, w+4, ... with four independent 16-B loads in flight each and are combined through LDS in a fixed order (one
// thread walking all the splits serially was a chain of up to 85 dependent loads: 18 us average, 62 us worst case).
// unpack_cin > 0: the [M = taps * cin][N = cout] matrix is a convolution's transposed weight gradient and C is that weight's own
// [cout][cin][taps] layout -- written here directly (the former separate eoe_conv_unpack_wgrad pass: 20 launches per WideResNet step)
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ part, float* __restrict__ C, int M, int N, int ldc,
                                                        int splits, float alpha, int accumulate, int unpack_cin, int unpack_taps) {
    __shared__ f32x4 red[4][64];
    const int q = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + q;
    const int n4 = N / 4;
    const bool valid = i < M * n4;
    const int m = valid ? i / n4 : 0, n = valid ? (i - m * n4) * 4 : 0;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
        const float* p = part + (size_t)m * N + n;
        const size_t st = (size_t)M * N;
        int k = w;
        for (; k + 12 < splits; k += 16) {
            const f32x4 a = *(const f32x4*)(p + k * st), b = *(const f32x4*)(p + (k + 4) * st);
            const f32x4 c = *(const f32x4*)(p + (k + 8) * st), d = *(const f32x4*)(p + (k + 12) * st);
            s += (a + b) + (c + d);
        }
        for (; k < splits; k += 4) s += *(const f32x4*)(p + k * st);
    }
    red[w][q] = s;
    __syncthreads();
    if (w != 0 || !valid) return;
    s = ((red[0][q] + red[1][q]) + (red[2][q] + red[3][q])) * alpha;
    if (unpack_cin) {
        const int tap = m / unpack_cin, ch = m - tap * unpack_cin;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* d = C + ((size_t)(n + r) * unpack_cin + ch) * unpack_taps + tap;
            *d = accumulate ? *d + s[r] : s[r];
        }
        return;
    }
    float* c = C + (size_t)m * ldc + n;
    if ((ldc & 3) == 0) {
        if (accumulate) s += *(const f32x4*)c;
        *(f32x4*)c = s;
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = accumulate ? c[r] + s[r] : s[r];
    }
}

int check_problem(const eoe_gemm_args* a, int T, int dtype) {
    EOE_CHECK_ARG(a->A && a->B && a->C, "gemm_tn: null operand");
    EOE_CHECK_ARG(a->M > 0 && a->N > 0 && a->K == T, "gemm_tn: bad shape %d %d %d (group T = %d)", a->M, a->N, a->K, T);
    EOE_CHECK_ARG(a->dtype == dtype, "gemm_tn: mixed dtypes in a group");
    EOE_CHECK_ARG((a->lda % 8) == 0 && (a->ldb % 8) == 0, "gemm_tn: lda/ldb must be multiples of 8 (16-B rows)");
    EOE_CHECK_ARG((((uintptr_t)a->A | (uintptr_t)a->B) & 15) == 0, "gemm_tn: A/B must be 16-B aligned");
    EOE_CHECK_ARG((a->M % 8) == 0 && (a->N % 8) == 0, "gemm_tn: M, N must be multiples of 8");
    EOE_CHECK_ARG((a->gather || a->lda >= a->M) && a->ldb >= a->N && a->ldc >= a->N, "gemm_tn: leading dims too small");
    EOE_CHECK_ARG(a->epilogue == EOE_EPI_NONE && a->out_f32 && !a->bias, "gemm_tn: only plain fp32 output is supported");
    return 0;
}

}  // namespace

// stream-K flag words: one block per (device, stream), owned by the library, zeroed once.  A launch publishes and waits for its own
// EPOCH (a per-block launch counter, never 0), so the words need no cleaning: a launch -- whatever happened to the ones before it --
// never trusts a flag it did not write itself.  Not used while the stream is being captured: a replayed launch would reuse its
// captured epoch and find the previous replay's flags (that launch then runs without stream-K).
struct SkFlags { int* flags; int epoch; };
static bool streamk_flags(hipStream_t s, int n, int** flags, int* epoch) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, SkFlags> table;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return false;
    std::lock_guard<std::mutex> lk(mu);
    auto it = table.find({dev, s});
    if (it == table.end()) {
        int* p = nullptr;
        if (hipMalloc(&p, (size_t)n * sizeof(int)) != hipSuccess) return false;
        if (hipMemset(p, 0, (size_t)n * sizeof(int)) != hipSuccess) { (void)hipFree(p); return false; }
        it = table.insert({{dev, s}, SkFlags{p, 0}}).first;
    }
    if (flags) {                                   // a launch: take the next epoch
        it->second.epoch = it->second.epoch >= 0x7ffffff0 ? 1 : it->second.epoch + 1;
        *flags = it->second.flags;
        *epoch = it->second.epoch;
    }
    return true;
}

static int tn_num_cus() {
    static const int ncu = [] { int d = 0, n = 256; hipDeviceProp_t pr; if (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess) n = pr.multiProcessorCount; return n; }();
    return ncu;
}

// Stream-K precondition, in one place (the launch below and vit.cpp's choice of the LayerNorm-1 side stream both ask): a plain
// group (no implicit patch matrix, no split) whose tiles fill between 5/8 and all of the CUs -- every one of the #CUs ranges must be
// RESIDENT at once (one 144-KiB workgroup per CU), because a range waits for its predecessor's partial inside the launch --, a
// reduction of at least 32 k-tiles, a 16-byte aligned workspace of EOE_TN_STREAMK_WORKSPACE_BYTES(#CUs), an eager (not captured)
// stream, and "tn_flags" bit 1 clear.
static bool streamk_applies(int tiles, int splits, int gather, int T, const void* ws, int64_t ws_bytes, hipStream_t s) {
    const int ncu = tn_num_cus();
    const size_t need = (size_t)ncu * (BM * BN * sizeof(float));
    if ((g_tn_flags & 2) || gather != 0 || splits != 1 || tiles >= ncu || tiles * 8 < ncu * 5 || cdiv(T, BK) < 32) return false;
    if (!ws || (size_t)ws_bytes < need || (((uintptr_t)ws) & 15) != 0) return false;
    return streamk_flags(s, ncu, nullptr, nullptr);
}

int eoe_tn256_splits(const eoe_gemm_args* args, int count, int64_t ws_bytes, const void* ws);      // gemm_tn256.hip
int eoe_launch_tn256(const eoe_gemm_args* args, int count, int splits, hipStream_t s);

// for vit.cpp: will eoe_gemm_tn_grouped(args, count, stream) fill every CU by itself (stream-K)?  (The wide-tile kernel's aligned
// slices leave CUs free -- 216 of 256 for a ViT block -- so LayerNorm-1 backward is worth running next to it.)
bool eoe_tn_streamk_would_run(const eoe_gemm_args* args, int count, void* stream) {
    if (!args || count < 1) return false;
    if (eoe_tn256_splits(args, count, args[0].workspace_bytes, args[0].workspace) > 0) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream, &st) == hipSuccess && st == hipStreamCaptureStatusNone) return false;
    }
    int tiles = 0;
    for (int i = 0; i < count; ++i) tiles += cdiv(args[i].M, BM) * cdiv(args[i].N, BN);
    const int ncu = tn_num_cus();
    const int T = args[0].K;
    const bool would_split = tiles * 8 < ncu * 5;
    return streamk_applies(tiles, would_split ? 2 : 1, args[0].gather, T, args[0].workspace, args[0].workspace_bytes, (hipStream_t)stream);
}

extern "C" int eoe_gemm_tn_grouped(const eoe_gemm_args* args, int count, void* stream) {
    EOE_CHECK_ARG(args && count >= 1 && count <= EOE_TN_MAX_GROUP, "gemm_tn_grouped: count %d not in [1, %d]", count,
                  EOE_TN_MAX_GROUP);
    const int T = args[0].K, dtype = args[0].dtype;
    EOE_CHECK_ARG(dtype == EOE_F16 || dtype == EOE_BF16, "gemm_tn: bad dtype %d", dtype);
    EOE_CHECK_ARG(T > 0, "gemm_tn: empty reduction");
    TnGroup g;
    memset(&g, 0, sizeof(g));
    int tiles = 0;
    double flops = 0, bytes = 0;
    for (int i = 0; i < count; ++i) {
        const eoe_gemm_args* a = &args[i];
        EOE_TRY(check_problem(a, T, dtype));
        EOE_CHECK_ARG(a->accumulate == args[0].accumulate && a->alpha == args[0].alpha,
                      "gemm_tn_grouped: accumulate / alpha must agree within a group");
        TnProblem& p = g.p[i];
        p.A = a->A; p.B = a->B; p.C = (float*)a->C; p.M = a->M; p.N = a->N; p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc;
        p.tiles_n = cdiv(a->N, BN);
        p.tile_start = tiles;
        tiles += cdiv(a->M, BM) * p.tiles_n;
        size_t ba = ((size_t)(T - 1) * a->lda + a->M) * 2;
        const size_t bb = ((size_t)(T - 1) * a->ldb + a->N) * 2;
        if (a->gather) {
            const eoe_conv_geometry& q = a->geo;
            EOE_CHECK_ARG(count == 1, "gemm_tn: an implicit patch matrix cannot be grouped");
            EOE_CHECK_ARG(q.n > 0 && q.H > 0 && q.W > 0 && q.C > 0 && q.kh > 0 && q.kw > 0 && q.stride > 0 && q.pad >= 0 &&
                          q.Ho > 0 && q.Wo > 0 && q.Ho * q.Wo < 65536 && T == q.n * q.Ho * q.Wo && T < (1 << 24), "gemm_tn: bad conv geometry");
            if (a->gather == 2) {
                EOE_CHECK_ARG(q.C == 4 && q.kw <= 8 && q.stride % 2 == 0 && q.pad == 0 && q.W % 2 == 0 && (q.Ho - 1) * q.stride + q.kh <= q.H &&
                              (q.Wo - 1) * q.stride + 8 <= q.W && a->M == (q.kh + 1) / 2 * 64, "gemm_tn: bad packed first-layer geometry");
            } else {
                EOE_CHECK_ARG(q.C % 8 == 0 && a->M == q.kh * q.kw * q.C, "gemm_tn: conv geometry does not match M = %d", a->M);
            }
            ba = (size_t)q.n * q.H * q.W * q.C * 2;
            g.gH = q.H; g.gW = q.W; g.gC = q.C; g.gWo = q.Wo; g.gHoWo = q.Ho * q.Wo; g.gkw = q.kw; g.gstride = q.stride; g.gpad = q.pad;
            g.mHoWo = (1ull << 40) / (unsigned)g.gHoWo + 1;      // exact for x * d < 2^40 (x < 2^24, d < 2^16)
            g.mWo = (1ull << 40) / (unsigned)g.gWo + 1;
        }
        EOE_CHECK_ARG(ba < 0x7fffffffull && bb < 0x7fffffffull, "gemm_tn: operand larger than 2 GiB");
        p.bytesA = (unsigned)ba; p.bytesB = (unsigned)bb;
        flops += 2.0 * a->M * a->N * T;
        bytes += 2.0 * ((double)T * a->M + (double)T * a->N) + 4.0 * a->M * a->N;
    }
    // wide tiles (gemm_tn256.hip): 256x256 output tiles, the reduction in aligned slices that meet inside the launch -- the ViT block's
    // four weight gradients; "tn_flags" bit 2 switches it off (A/B)
    if (const int s256 = eoe_tn256_splits(args, count, args[0].workspace_bytes, args[0].workspace)) {
        ProfScope ps256("gemm_tn", flops, bytes, stream);
        const int rc = eoe_launch_tn256(args, count, s256, (hipStream_t)stream);
        if (rc >= 0) return rc;                       // < 0: not available on this stream (being captured): the kernel below
    }
    static const int dbg = getenv("EOE_GEMM_DEBUG") ? atoi(getenv("EOE_GEMM_DEBUG")) : 0;
    if (dbg & 1) for (int i = 0; i < count; ++i) { g.p[i].bytesA = 0; g.p[i].bytesB = 0; }
    g.count = count; g.T = T; g.total_tiles = tiles; g.accumulate = args[0].accumulate; g.alpha = args[0].alpha;
    // split the reduction only when the whole group leaves most of the chip idle
    // (as many splits as fill the CUs exactly once: tiles * splits <= #CUs -- a power-of-two rule left e.g. 18 tiles x 16
    //  splits = 288 workgroups on 256 CUs, i.e. a second, nearly empty round)
    static const int ncu = [] { int d = 0, n = 256; hipDeviceProp_t pr; if (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess) n = pr.multiProcessorCount; return n; }();
    int splits = 1;
    if (tiles * 8 < ncu * 5) {
        splits = ncu / tiles;
        if (splits > T / 512) splits = T / 512;
        if (splits > 256) splits = 256;
        if (splits < 1) splits = 1;
    }
    int t_per = cdiv(cdiv(T, splits), BK) * BK;
    splits = cdiv(T, t_per);
    g.splits = splits; g.t_per_split = t_per;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("gemm_tn", flops, bytes, stream);
    g.part = nullptr;
    const bool unpack = args[0].unpack_dw != 0;
    if (unpack) {
        EOE_CHECK_ARG(count == 1 && args[0].gather == 1 && args[0].workspace && args[0].M == args[0].geo.kh * args[0].geo.kw * args[0].geo.C,
                      "gemm_tn: unpack_dw needs a single implicit-convolution problem and a workspace");
        EOE_CHECK_ARG((size_t)splits * args[0].M * args[0].N * sizeof(float) <= (size_t)args[0].workspace_bytes &&
                      (((uintptr_t)args[0].workspace) & 15) == 0, "gemm_tn: unpack_dw: workspace too small");
    }
    if ((splits > 1 || unpack) && args[0].workspace) {
        size_t need = 0;
        for (int i = 0; i < count; ++i) {
            g.part_stride[i] = (long long)need;
            need += (size_t)splits * args[i].M * args[i].N;
        }
        if (need * sizeof(float) <= (size_t)args[0].workspace_bytes && (((uintptr_t)args[0].workspace) & 15) == 0)
            g.part = (float*)args[0].workspace;
    }
    if (splits > 1 && !g.accumulate && !g.part) {
        for (int i = 0; i < count; ++i) {
            EOE_CHECK_ARG(args[i].ldc == args[i].N, "gemm_tn: split reduction needs a dense C");
            if (hipMemsetAsync(args[i].C, 0, (size_t)args[i].M * args[i].N * sizeof(float), s) != hipSuccess)
                return eoe_set_error(EOE_ERR_LAUNCH, "gemm_tn: memset failed");
        }
    }
    const int gather = args[0].gather;
    // stream-K (streamk_applies): the ViT block's four wgrads (216 tiles) run as #CUs equal k-ranges instead (kernel comment)
    int grid = tiles * splits;
    if (streamk_applies(tiles, splits, gather, T, args[0].workspace, args[0].workspace_bytes, s)) {
        int* flags = nullptr;
        int epoch = 0;
        if (streamk_flags(s, ncu, &flags, &epoch)) {
            g.sk_wgs = ncu;
            g.sk_part = (float*)args[0].workspace;
            g.sk_flags = flags;
            g.sk_epoch = epoch;
            grid = ncu;
        }
    }
#define EOE_TN_LAUNCH_RD(TT, GG, RR)                                                                                       \
    do {                                                                                                                   \
        static bool once = (hipFuncSetAttribute((const void*)gemm_tn_grouped_kernel<TT, GG, RR>,                           \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES), true);            \
        (void)once;                                                                                                        \
        hipLaunchKernelGGL((gemm_tn_grouped_kernel<TT, GG, RR>), dim3(grid), dim3(512), SMEM_BYTES, s, g);       \
    } while (0)
    // option "tn_flags" bit 0 = 1 selects the intrinsic fragment reads (A/B of tr_frag_asm, plain kernel only)
#define EOE_TN_LAUNCH(TT, GG)                                                                                              \
    do {                                                                                                                   \
        if (GG == 0 && (g_tn_flags & 1)) EOE_TN_LAUNCH_RD(TT, GG, 0); else EOE_TN_LAUNCH_RD(TT, GG, 1);                    \
    } while (0)
    if (dtype == EOE_F16) {
        if (gather == 2) EOE_TN_LAUNCH(f16_t, 2); else if (gather) EOE_TN_LAUNCH(f16_t, 1); else EOE_TN_LAUNCH(f16_t, 0);
    } else {
        if (gather == 2) EOE_TN_LAUNCH(bf16_t, 2); else if (gather) EOE_TN_LAUNCH(bf16_t, 1); else EOE_TN_LAUNCH(bf16_t, 0);
    }
#undef EOE_TN_LAUNCH
#undef EOE_TN_LAUNCH_RD
    EOE_CHECK_LAUNCH("gemm_tn_grouped");
    if (g.part) {
        for (int i = 0; i < count; ++i) {
            hipLaunchKernelGGL(tn_reduce_kernel, dim3(cdiv(args[i].M * (args[i].N / 4), 64)), dim3(256), 0, s,
                               (const float*)(g.part + g.part_stride[i]), (float*)args[i].C, args[i].M, args[i].N, args[i].ldc, splits,
                               g.alpha, g.accumulate, unpack ? args[i].geo.C : 0, unpack ? args[i].geo.kh * args[i].geo.kw : 0);
        }
        EOE_CHECK_LAUNCH("gemm_tn_reduce");
    }
    return 0;
}

extern "C" int eoe_gemm_tn(const eoe_gemm_args* a, void* stream) {
    EOE_CHECK_ARG(a != nullptr, "gemm_tn: null args");
    return eoe_gemm_tn_grouped(a, 1, stream);
}
