// Weight-gradient GEMM, wide tiles:  C[M,N] (fp32) = A[T,M]^T . B[T,N]  with 256 x 256 output tiles, one wave per SIMD.
//
// Why (round 3): the 256x128 kernel of gemm_tn.hip moves 48 KiB of operands per 64-deep k-tile for 1024 cycles of MFMA issue per
// SIMD -- the CU's 64 B/clk vector-memory path is as busy as the matrix pipes, and the block's four weight gradients cost 784 MB of
// fabric traffic per launch against 343 MB of operands (2.3x).  A 256x256 tile needs 64 KiB per k-tile for 2048 cycles of MFMAs (half
// the bytes per flop), and four wavefronts of 128x128 (the whole 512-entry register file each: 256 accumulator registers) read 0.25
// LDS fragments per MFMA instead of 0.5.  Main loop = gemm256.hip's (inline-asm MFMAs with the accumulators tied in place, fragment
// reads and LDS-DMA pieces woven between them, counted vmcnt, one raw barrier per k-tile, 2-stage 128 KiB ring), fragments by the
// hardware-transposing ds_read_b64_tr_b16 from gemm_tn.hip's granule-swizzled [64 t][256 cols] images.
//
// Decomposition: the ViT block's four wgrads are 108 such tiles -- too few for 256 CUs, and stream-K ranges (gemm_tn.hip) start at
// unrelated k offsets, so workgroups that share an operand panel never read it at the same time and nothing is shared through the
// XCD's L2.  Here the reduction is cut into S ALIGNED slices (S = 2 for the block: 216 workgroups, each XCD holding 27 neighbouring
// tiles of ONE slice = 9 + 3 shared panels per k-tile instead of 54).  The S workgroups of a tile meet inside the launch: whoever
// finishes its main loop LAST (an arrival ticket per tile) becomes the tile's owner, the others store their accumulators (fragment
// order, 256 KiB) and raise a flag; the owner adds them in slice order -- its own registers standing in at its own position, so the
// sum does not depend on who came last: bitwise reproducible, no atomics on data -- and writes C.  The owner only ever waits for
// workgroups that have already arrived, i.e. are resident and a few microseconds from done: no co-residency assumption, safe next to
// RCCL's kernels.  Flags carry the launch's epoch; the arrival counters alternate between two arrays, the idle one being cleared by
// the running launch for the next.
#include "common.h"
#include <map>
#include <mutex>
#include <utility>
#include <stdlib.h>

extern int g_tn_flags;
extern unsigned long long* g_stamp_buf;       // gemm.hip (eoe_debug_gemm_stamps copies it out)
int g_tn256_launches = 0;      // diagnostics: eoe_get_option("tn256_launches")

namespace {

constexpr int BK = 64;
// (round 4) The ring holds FOUR stages of one 32-deep k-step each instead of two 64-deep k-tiles.  Same 128 KiB, but three stages (96 KiB)
// are in flight while the fourth is read, against one of two (64 KiB): the k-loop is bound by the LATENCY of the staged operands -- a CU
// receives (bytes in flight) / (~2 us under load), measured 2.0 us per 64-deep k-tile against 1.37 us of MFMA work -- not by their bandwidth.
constexpr int KS = 32;                            // rows of t per stage
constexpr int IMG_BYTES = KS * 256 * 2;           // 16 KiB: [32 t][256 cols] 16-bit
constexpr int STAGE = 2 * IMG_BYTES;              // 32 KiB
constexpr int NST = 4;
constexpr int SMEM = NST * STAGE;                 // 128 KiB: one workgroup per CU
// NW = waves per workgroup.  4: one wave per SIMD, 128 x 128 of the tile each (256 accumulator registers).  8 (round 4): two waves per SIMD,
// 128 x 64 each (128 accumulators + 128 other registers) -- a wave issues one instruction at a time and the fragment reads, LDS-DMA pieces
// and scalar work of a k-step need about as many issue cycles as its MFMAs leave free (stamps: 3196 cycles per k-tile for 2048 of matrix
// work); instructions of DIFFERENT waves of a SIMD issue side by side (gemm_w8.hip's header).  Same products in the same k order per
// accumulator, same slice meeting: same bits.
constexpr int ROWB = 512;                         // bytes per image row

struct P256 {
    const void* A; const void* B; float* C;
    int M, N, lda, ldb, ldc, tiles_n, tile_start;
    unsigned bytesA, bytesB;
};
struct G256 {
    P256 p[EOE_TN_MAX_GROUP];
    int count, T, total_tiles, splits, nk_tile, accumulate;
    float alpha;
    float* part;          // [total_tiles * splits][256 * 256] partial accumulators, fragment order
    int* flags;           // [total_tiles * splits] = epoch once that slice's partial is visible
    int* arrive;          // [total_tiles] arrival tickets of THIS launch
    int* arrive_next;     // [total_tiles] cleared by this launch for the next one
    int epoch;
    unsigned long long* stamp;   // diagnostics (EOE_GEMM_STAMP=1): per workgroup 8 words, see the kernel; else NULL
};

__device__ __forceinline__ int swz(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

template <typename T> __device__ __forceinline__ void mfma_inplace(f32x4& c, typename T16<T>::v8 a, typename T16<T>::v8 b);
template <> __device__ __forceinline__ void mfma_inplace<f16_t>(f32x4& c, f16x8 a, f16x8 b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <> __device__ __forceinline__ void mfma_inplace<bf16_t>(f32x4& c, bf16x8 a, bf16x8 b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

template <typename T>
__device__ __forceinline__ typename T16<T>::v8 join(i16x4v lo, i16x4v hi) {
    i16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(typename T16<T>::v8, r);
}

#define EOE_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <typename T, int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void gemm_tn256_kernel(G256 g) {
    static_assert(NW == 4 || NW == 8, "waves");
    constexpr int NI = NW == 8 ? 4 : 8;                    // 16-column groups of the B image per wave: 128 x (16 NI) of the tile
    constexpr int PIW = 16 / NW;                           // wave-loads (2 rows) per wave, image and stage
    constexpr int NMF = 8 * NI;                            // MFMAs per cluster
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int items = g.total_tiles * g.splits;
    if (blockIdx.x == 0)                                   // the next launch's arrival counters (nobody touches them in this one)
        for (int i = tid; i < g.total_tiles; i += 64 * NW) g.arrive_next[i] = 0;
    // item order: slice-major, XCD-remapped -> the workgroups of an XCD hold neighbouring tiles of one k-slice
    const int item = xcd_remap((int)blockIdx.x, items);
    const int slice = item / g.total_tiles, gt = item - slice * g.total_tiles;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < EOE_TN_MAX_GROUP; ++i)
        if (i < g.count && gt >= g.p[i].tile_start) pi = i;
    const P256& P = g.p[pi];
    const int lt = gt - P.tile_start;
    const int m0 = (lt / P.tiles_n) * 256, n0 = (lt % P.tiles_n) * 256;
    const int kt0 = (int)(((long long)g.nk_tile * slice) / g.splits), kt1 = (int)(((long long)g.nk_tile * (slice + 1)) / g.splits);
    const int nk = 2 * (kt1 - kt0);                        // 32-deep k-steps of this slice
    const int t_begin = kt0 * BK;

    __amdgpu_buffer_rsrc_t ra = make_rsrc(P.A, P.bytesA);
    __amdgpu_buffer_rsrc_t rb = make_rsrc(P.B, P.bytesB);
    const int lda = P.lda, ldb = P.ldb;

    // staging: a stage's image is 32 rows x 512 B = 16 wave-loads of 2 rows, PIW per wave; lane -> (row, 16-B slot holding granule-swizzled chunk)
    unsigned baseA[PIW], baseB[PIW];
    const int row_in = lane >> 5, s = lane & 31;
#pragma unroll
    for (int j = 0; j < PIW; ++j) {
        const int row = (wave * PIW + j) * 2 + row_in;
        const int c16 = (((s >> 1) ^ swz(row)) << 1) | (s & 1);
        const int ca = m0 + c16 * 8, cb = n0 + c16 * 8;
        baseA[j] = (ca < P.M) ? (unsigned)(((size_t)row * lda + ca) * 2) : EOE_OOB;
        baseB[j] = (cb < P.N) ? (unsigned)(((size_t)row * ldb + cb) * 2) : EOE_OOB;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_void_t*)smem);
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(wave);
#define EOE_DMA16(rsrc, lds_addr, voff)                                                                               \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"                            \
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory")
    int st_kt = 0, st_slot = 0;                            // staging cursor (k-tile of this slice, ring slot)
    // piece j of this wave: j < 8 -> the A image, else the B image.  No per-piece predicate: rows past T lie beyond the buffer
    // descriptor's size (an offset >= num_records reads as zeros: the ragged last k-tile pads itself), columns past M / N carry the
    // out-of-range base, and a cursor past the slice's last k-tile (`st_off_*` = out of range) fills a dead slot with zeros
    unsigned st_off_a = 0, st_off_b = 0;                   // byte offset of the cursor's k-tile in A / B (wave-uniform)
    auto stage_cursor = [&]() {
        const int t0 = t_begin + st_kt * KS;
        const bool live = st_kt < nk;
        st_off_a = live ? (unsigned)((size_t)t0 * lda * 2) : EOE_OOB;
        st_off_b = live ? (unsigned)((size_t)t0 * ldb * 2) : EOE_OOB;
    };
    auto stage_piece = [&](int j) {                        // j < PIW: the A image, else the B image
        const unsigned sa = lds0 + (unsigned)st_slot * STAGE;
        const int jj = j < PIW ? j : j - PIW;
        if (j < PIW) {
            const unsigned la = sa + (wave_u * PIW + jj) * 1024u, vo = baseA[jj] + st_off_a;
            EOE_DMA16(ra, la, vo);
        } else {
            const unsigned lb = sa + IMG_BYTES + (wave_u * PIW + jj) * 1024u, vo = baseB[jj] + st_off_b;
            EOE_DMA16(rb, lb, vo);
        }
    };
    auto stage_issue = [&]() {
        stage_cursor();
#pragma unroll
        for (int j = 0; j < 2 * PIW; ++j) stage_piece(j);
    };
    auto stage_advance = [&]() { st_slot = (st_slot + 1) & (NST - 1); ++st_kt; };

    // fragments: lane 4q+p of a 16-lane group addresses row q of the group's 4-row block, columns 4p..4p+3 (transposed read)
    const int wm0 = NW == 8 ? (wave >> 2) * 128 : (wave >> 1) * 128, wn0 = NW == 8 ? (wave & 3) * 64 : (wave & 1) * 128;
    const int lr = lane & 15, lg = lane >> 4;
    const int q = lr >> 2, pp = lr & 3;
    const int f = q | ((lg & 1) << 2);                     // swz(row) for row = 32*ks + 8*lg + 4*h + q
    const int rowsel = 8 * lg + q;
    unsigned offA[8], offB[NI];
#pragma unroll
    for (int i = 0; i < 8; ++i) offA[i] = (unsigned)(rowsel * ROWB + ((((wm0 >> 4) + i) ^ f) << 5) + pp * 8);
#pragma unroll
    for (int i = 0; i < NI; ++i) offB[i] = (unsigned)(IMG_BYTES + rowsel * ROWB + ((((wn0 >> 4) + i) ^ f) << 5) + pp * 8);
    typedef typename T16<T>::v8 V8;

    f32x4 acc[8][NI];                                      // [m 16-col group of the A image][n 16-col group of the B image]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // one fragment = two 8-byte transposed reads (rows r and r + 4); kept as halves until they have landed
#define EOE_TRREAD(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF))
    // wait for every outstanding LDS read; names all 32 halves of one fragment set so that no use is scheduled above the wait
#define EOE_LANDED(LA, HA, LB, HB)                                                                                           \
    do {                                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(LA[0]), "+v"(LA[1]), "+v"(LA[2]), "+v"(LA[3]), "+v"(LA[4]), "+v"(LA[5]), "+v"(LA[6]), \
                     "+v"(LA[7]), "+v"(HA[0]), "+v"(HA[1]), "+v"(HA[2]), "+v"(HA[3]), "+v"(HA[4]), "+v"(HA[5]), "+v"(HA[6]), "+v"(HA[7]) :: "memory"); \
        asm volatile("" : "+v"(LB[0]), "+v"(LB[1]), "+v"(LB[2]), "+v"(LB[3]), "+v"(LB[NI > 4 ? 4 : 0]), "+v"(LB[NI > 4 ? 5 : 1]), "+v"(LB[NI > 4 ? 6 : 2]), \
                     "+v"(LB[NI > 4 ? 7 : 3]), "+v"(HB[0]), "+v"(HB[1]), "+v"(HB[2]), "+v"(HB[3]), "+v"(HB[NI > 4 ? 4 : 0]), "+v"(HB[NI > 4 ? 5 : 1]), \
                     "+v"(HB[NI > 4 ? 6 : 2]), "+v"(HB[NI > 4 ? 7 : 3]) :: "memory");                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
    } while (0)
    // the 64 MFMAs of one 32-deep k-step on fragment set X, with woven between them (a) the 32 half-reads of the OTHER set R from
    // ring slot `rbase`, k-step `rks` -- one behind every second MFMA --, and (b) (DMA) this wave's 16 LDS-DMA pieces of the next
    // k-tile behind the (wave + 1)-th quarter of the cluster (the four SIMDs share one LDS-DMA path: staggered, not a burst)
#define EOE_CLUSTER(XLA, XHA, XLB, XHB, RLA, RHA, RLB, RHB, rbase, DMA, PH)                                                      \
    do {                                                                                                                     \
        const unsigned rb_ = (unsigned)(rbase);                                                                              \
        _Pragma("unroll") for (int mi = 0; mi < 8; ++mi) {                                                                   \
            const V8 xa_ = join<T>(XLA[mi], XHA[mi]);                                                                        \
            _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) {                                                              \
                const int idx = mi * NI + ni;                                                                                \
                mfma_inplace<T>(acc[mi][ni], join<T>(XLB[ni], XHB[ni]), xa_);                                                \
                /* the 16 + 2 NI half-reads of the other set: one behind every second MFMA (NI = 8), behind each of the first 24 (NI = 4) */ \
                if (NI == 8 ? (idx & 1) == 0 : idx < 16 + 2 * NI) {                                                          \
                    const int j = NI == 8 ? idx >> 1 : idx, fr = j >> 1;                                                     \
                    if (fr < 8) {                                                                                            \
                        if (j & 1) EOE_TRREAD(RHA[fr < 8 ? fr : 0], rb_ + offA[fr < 8 ? fr : 0], 4 * ROWB);                  \
                        else EOE_TRREAD(RLA[fr < 8 ? fr : 0], rb_ + offA[fr < 8 ? fr : 0], 0);                               \
                    } else {                                                                                                 \
                        if (j & 1) EOE_TRREAD(RHB[fr >= 8 ? fr - 8 : 0], rb_ + offB[fr >= 8 ? fr - 8 : 0], 4 * ROWB);        \
                        else EOE_TRREAD(RLB[fr >= 8 ? fr - 8 : 0], rb_ + offB[fr >= 8 ? fr - 8 : 0], 0);                     \
                    }                                                                                                        \
                }                                                                                                            \
                /* this wave's 2 PIW pieces of the k-step four ahead, one behind every 8th MFMA (spread: the SIMDs share one LDS-DMA path) */ \
                if ((DMA) && (idx & 7) == (PH) && (idx >> 3) < 2 * PIW) stage_piece(idx >> 3);                               \
            }                                                                                                                \
        }                                                                                                                    \
    } while (0)

    i16x4v la0[8], ha0[8], lb0[NI], hb0[NI], la1[8], ha1[8], lb1[NI], hb1[NI];
    // counted wait: everything but the pieces of the two youngest stages has landed (2 PIW pieces per wave and stage)
#define EOE_WAIT_2STAGES() do { if (PIW == 2) { EOE_WAIT_VM(8); } else { EOE_WAIT_VM(16); } } while (0)
    if (nk > 0) {                                          // nk is even and >= 48 (eoe_tn256_splits: >= 24 k-tiles per slice)
#pragma unroll 1
        for (int i = 0; i < NST; ++i) { stage_issue(); stage_advance(); }
        if (PIW == 2) { EOE_WAIT_VM(12); } else { EOE_WAIT_VM(24); }       // k-step 0 landed; k-steps 1..3 may stay in flight
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            EOE_TRREAD(la0[i], offA[i], 0);
            EOE_TRREAD(ha0[i], offA[i], 4 * ROWB);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            EOE_TRREAD(lb0[i], offB[i], 0);
            EOE_TRREAD(hb0[i], offB[i], 4 * ROWB);
        }
        int cur = 0;                                       // ring stage of the k-step being multiplied
        // diagnostics: wave 0's cycles [waiting for the staged k-step and its fragments], [at the barrier], [in the clusters]
        unsigned long long* stp = g.stamp ? g.stamp + (size_t)blockIdx.x * 8 : nullptr;
        unsigned long long t_a = 0, t_b = 0, t_c = 0, c_first = 0, c_bar = 0, c_second = 0, t_loop0 = 0;
        if (stp) { t_c = __builtin_amdgcn_s_memtime(); t_loop0 = t_c; }
        // k-step s (fragments in set s & 1, read from stage s % 4 during the previous cluster):
        //   wait: stage s + 1 has landed (the pieces of s + 2, s + 3 stay in flight) and this wave's reads of stage s are complete;
        //   barrier: ... for every wave -- stage s + 1 may be read, stage s may be refilled;
        //   cluster: the MFMAs of k-step s + the reads of k-step s + 1 + the pieces of k-step s + 4 into stage s.
        // Past the slice's last k-step the pieces carry the out-of-range offset (nothing fetched, a dead stage zero-filled) and the reads
        // fetch a stale stage and are discarded: the counts stay constant.
#define EOE_KSTEP(XLA, XHA, XLB, XHB, RLA, RHA, RLB, RHB)                                                                    \
        do {                                                                                                                 \
            EOE_WAIT_2STAGES();                                                                                              \
            EOE_LANDED(XLA, XHA, XLB, XHB);                                                                                  \
            if (stp) { t_a = __builtin_amdgcn_s_memtime(); c_first += t_a - t_c; }                                           \
            __builtin_amdgcn_s_barrier();                                                                                    \
            if (stp) { t_b = __builtin_amdgcn_s_memtime(); c_bar += t_b - t_a; }                                             \
            stage_cursor();                                                                                                  \
            if (dma_phase & 1) asm volatile("s_nop 15" ::: "memory");                                                        \
            if (dma_phase & 2) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                                            \
            EOE_CLUSTER(XLA, XHA, XLB, XHB, RLA, RHA, RLB, RHB, (unsigned)((cur + 1) & (NST - 1)) * STAGE, true, 7);         \
            stage_advance();                                                                                                 \
            if (stp) { t_c = __builtin_amdgcn_s_memtime(); c_second += t_c - t_b; }                                          \
            cur = (cur + 1) & (NST - 1);                                                                                     \
        } while (0)
        const int dma_phase = (int)(wave_u & 3);           // the waves of a SIMD pair / of the CU a little apart behind the barrier
#pragma unroll 1
        for (int ks = 0; ks < nk; ks += 2) {
            EOE_KSTEP(la0, ha0, lb0, hb0, la1, ha1, lb1, hb1);
            EOE_KSTEP(la1, ha1, lb1, hb1, la0, ha0, lb0, hb0);
        }
        EOE_LANDED(la0, ha0, lb0, hb0);                    // the discarded reads of the last cluster
#undef EOE_KSTEP
        if (stp && lane == 0) {
            unsigned long long* o = stp + (wave & 1) * 4;
            if (wave < 2) { o[0] = c_first; o[1] = c_bar; o[2] = c_second; o[3] = t_c - t_loop0; }
        }
    }
#undef EOE_WAIT_2STAGES
    EOE_WAIT_VM(0);                                        // the (dead) LDS-DMA pieces of the last iterations have landed: the ring is reused below
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the last MFMAs' results have landed before the accumulators are read
#undef EOE_CLUSTER
#undef EOE_LANDED
#undef EOE_TRREAD
#undef EOE_DMA16

    // ---- the tile's S workgroups meet: the last to arrive owns the tile
    __syncthreads();                                       // every wave is done with the ring (the words below borrow its first bytes)
    volatile int* wg_word = (volatile int*)smem;
    const int S = g.splits;
    int role_owner = 1;
    if (S > 1) {
        if (tid == 0) wg_word[0] = __hip_atomic_fetch_add(g.arrive + gt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        role_owner = (wg_word[0] == S - 1) ? 1 : 0;
        __syncthreads();
    }
    const size_t frag_base = (size_t)(wave * 8 * NI) * 64 * 4;  // this wave's 8 NI accumulator quads x 64 lanes x 4 floats
    if (!role_owner) {
        float* mine = g.part + (size_t)(gt * S + slice) * (256 * 256) + frag_base;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) *(f32x4*)(mine + ((size_t)(mi * NI + ni) * 64 + lane) * 4) = acc[mi][ni];
        // publish (cdna_hip_programming.md, Guideline 16): every storing wave drains, the workgroup meets, ONE lane releases at agent
        // scope and raises the flag with a relaxed agent-scope store
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(g.flags + gt * S + slice, g.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    if (S > 1) {
        // owner: the sum over the slices in SLICE ORDER, whoever owns the tile (fp32 addition is commutative, not associative: the order
        // of the additions is what must not depend on arrival).  S = 2: own + other == other + own bit for bit, straight from the
        // registers.  S > 2: the owner parks its own accumulators in its slot too and adds all S slots from memory, first to last.
        bool poisoned = false;
        if (S > 2) {
            float* mine = g.part + (size_t)(gt * S + slice) * (256 * 256) + frag_base;
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    *(f32x4*)(mine + ((size_t)(mi * NI + ni) * 64 + lane) * 4) = acc[mi][ni];
                    acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (read back below behind the acquire of the first foreign slot)
        }
        for (int sidx = 0; sidx < S; ++sidx) {
            if (sidx == slice && S == 2) continue;
            if (sidx != slice) {
                if (tid == 0) {
                    int spins = 0;
                    while (__hip_atomic_load(g.flags + gt * S + sidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != g.epoch && ++spins < (1 << 22))
                        __builtin_amdgcn_s_sleep(8);
                    wg_word[0] = spins >= (1 << 22) ? 1 : 0;
                }
                __syncthreads();
                const bool timed_out = wg_word[0] != 0;
                __syncthreads();
                if (timed_out) { poisoned = true; break; }
            }
            if (tid == 0) {                                  // ONE acquire per slot read (own slot included: this CU's L1 may hold an old line)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            const float* theirs = g.part + (size_t)(gt * S + sidx) * (256 * 256) + frag_base;
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = acc[mi][ni] + *(const f32x4*)(theirs + ((size_t)(mi * NI + ni) * 64 + lane) * 4);
        }
        if (poisoned) {                                    // loud, not silent
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){NAN, NAN, NAN, NAN};
        }
    }
    // epilogue: D = (B fragment as the A operand) x (A fragment as the B operand): the lane holds 4 consecutive n of one output row m
    const bool vec = ((P.ldc & 3) == 0) && ((P.N & 3) == 0);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int m = m0 + wm0 + mi * 16 + lr;
        if (m >= P.M) continue;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
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
                    c[r] = g.accumulate ? c[r] + v[r] : v[r];
                }
            }
        }
    }
}

struct SyncState { int* flags; int* arrive[2]; int epoch; };
// per (device, stream): flag words + two arrival-counter arrays, zeroed once; every launch takes the next epoch
bool sync_state(hipStream_t s, int n, G256& g) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, SyncState> table;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;      // a replayed launch would reuse its captured epoch
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return false;
    std::lock_guard<std::mutex> lk(mu);
    auto it = table.find({dev, s});
    if (it == table.end()) {
        int* p = nullptr;
        if (hipMalloc(&p, (size_t)3 * n * sizeof(int)) != hipSuccess) return false;
        if (hipMemset(p, 0, (size_t)3 * n * sizeof(int)) != hipSuccess) { (void)hipFree(p); return false; }
        it = table.insert({{dev, s}, SyncState{p, {p + n, p + 2 * n}, 0}}).first;
    }
    SyncState& ss = it->second;
    ss.epoch = ss.epoch >= 0x7ffffff0 ? 1 : ss.epoch + 1;
    g.flags = ss.flags;
    g.arrive = ss.arrive[ss.epoch & 1];
    g.arrive_next = ss.arrive[(ss.epoch + 1) & 1];
    g.epoch = ss.epoch;
    return true;
}

int num_cus256() {
    static const int ncu = [] { int d = 0, n = 256; hipDeviceProp_t pr; if (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess) n = pr.multiProcessorCount; return n; }();
    return ncu;
}

}  // namespace

// the slicing the wide-tile kernel would use for this group, or 0 if the group is not for it: plain problems (no implicit patch
// matrix), a long reduction, and S <= 4 aligned slices of >= 24 k-tiles that put at least 5/8 of the CUs to work
int eoe_tn256_splits(const eoe_gemm_args* args, int count, int64_t ws_bytes, const void* ws) {
    if ((g_tn_flags & 4) || !args || count < 1 || args[0].gather || args[0].unpack_dw) return 0;
    const int T = args[0].K, nk = (T + BK - 1) / BK, ncu = num_cus256();
    int tiles = 0;
    for (int i = 0; i < count; ++i) {
        if (args[i].M < 256 || args[i].N < 256) return 0;            // narrow outputs keep the 256x128 kernel
        tiles += cdiv(args[i].M, 256) * cdiv(args[i].N, 256);
    }
    if (tiles > ncu) return 0;
    int best = 0;
    for (int S = 1; S <= 4; ++S)
        if (tiles * S <= ncu && nk / S >= 24) best = S;
    if (!best || tiles * best * 8 < ncu * 5) return 0;
    if (best > 1 && (!ws || (size_t)ws_bytes < (size_t)tiles * best * 256 * 256 * sizeof(float) || (((uintptr_t)ws) & 15) != 0)) return 0;
    return best;
}

int eoe_launch_tn256(const eoe_gemm_args* args, int count, int splits, hipStream_t s) {
    G256 g;
    memset(&g, 0, sizeof(g));
    const int T = args[0].K;
    int tiles = 0;
    for (int i = 0; i < count; ++i) {
        const eoe_gemm_args* a = &args[i];
        P256& p = g.p[i];
        p.A = a->A; p.B = a->B; p.C = (float*)a->C; p.M = a->M; p.N = a->N; p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc;
        p.tiles_n = cdiv(a->N, 256);
        p.tile_start = tiles;
        tiles += cdiv(a->M, 256) * p.tiles_n;
        const size_t ba = ((size_t)(T - 1) * a->lda + a->M) * 2, bb = ((size_t)(T - 1) * a->ldb + a->N) * 2;
        EOE_CHECK_ARG(ba < 0x7fffffffull && bb < 0x7fffffffull, "gemm_tn: operand larger than 2 GiB");
        p.bytesA = (unsigned)ba; p.bytesB = (unsigned)bb;
    }
    static const int dbg = getenv("EOE_GEMM_DEBUG") ? atoi(getenv("EOE_GEMM_DEBUG")) : 0;      // diagnostics: bit 0 = every operand load
    if (dbg & 1) for (int i = 0; i < count; ++i) { g.p[i].bytesA = 0; g.p[i].bytesB = 0; }     // out of range (nothing fetched; results wrong)
    static const int stampon = getenv("EOE_GEMM_STAMP") ? atoi(getenv("EOE_GEMM_STAMP")) : 0;
    if (stampon) {
        static unsigned long long* buf = [] { void* b = nullptr; (void)hipMalloc(&b, 4096 * 16 * 8); return (unsigned long long*)b; }();
        g.stamp = buf;
        g_stamp_buf = buf;
    }
    g.count = count; g.T = T; g.total_tiles = tiles; g.splits = splits; g.nk_tile = cdiv(T, BK);
    g.accumulate = args[0].accumulate; g.alpha = args[0].alpha;
    g.part = (float*)args[0].workspace;
    // (a single slice needs no meeting, but the kernel clears arrive_next either way.)  Captured stream: -1, the caller falls back
    // to the 256x128 kernel
    if (!sync_state(s, num_cus256(), g)) return -1;
    const int dtype = args[0].dtype;
    const bool w8 = !(g_tn_flags & 8);                     // tn_flags bit 3 = 8: the one-wave-per-SIMD form (A/B)
#define EOE_TN256_LAUNCH(TT, NW_)                                                                                                  \
    do {                                                                                                                          \
        static bool once = (hipFuncSetAttribute((const void*)gemm_tn256_kernel<TT, NW_>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM), true); \
        (void)once;                                                                                                               \
        hipLaunchKernelGGL((gemm_tn256_kernel<TT, NW_>), dim3(tiles * splits), dim3(64 * NW_), SMEM, s, g);                      \
    } while (0)
    if (dtype == EOE_F16) { if (w8) EOE_TN256_LAUNCH(f16_t, 8); else EOE_TN256_LAUNCH(f16_t, 4); }
    else { if (w8) EOE_TN256_LAUNCH(bf16_t, 8); else EOE_TN256_LAUNCH(bf16_t, 4); }
#undef EOE_TN256_LAUNCH
    EOE_CHECK_LAUNCH("gemm_tn256");
    ++g_tn256_launches;
    return 0;
}
