// The one-wave-per-SIMD NT GEMM kernel (its own translation unit: 256 accumulator registers per lane make it slow to compile).
#include "gemm_common.h"

namespace {

// ------------------------------------------------------------------------------------------------ NT, (32*MI) x 256 tiles, one wave per SIMD
// Fourth tile shape: (32*MI) x 256 x 64 tiles, FOUR wavefronts (2x2), each (16*MI) x 128 -- one wave per SIMD with the whole
// 512-entry register file (up to 256 accumulator registers + the fragments), ONE persistent workgroup per CU.  Why: the 64x64
// wave tiles of the kernels above read 16 KB of LDS per 32 MFMAs (0.5 ds_read_b128 per MFMA); with two waves per SIMD the LDS
// array and the per-k-tile barrier skew, not the matrix pipe, set the k-tile time (~1800 cycles against 1024 of MFMA issue).  An
// 80x128 wave tile reads 13 KB per 40 MFMAs (0.33 per MFMA), a 128x128 one 16 KB per 64 (0.25), halves the operand bytes that
// cross L2 per flop, and a k-tile holds 1280 / 2048 cycles of back-to-back MFMAs behind ONE barrier.  MI = 5 (160 x 256) tiles
// the ViT's M = 12800 without waste: N = 768 -> 240 tiles = one round over the 256 CUs, N = 2304 -> 720 = 3 rounds at 94 %,
// N = 3072 -> 960 = 4 rounds at 94 %.  Persistent grid with cross-tile prefetch (the staging cursor runs two k-tiles ahead across
// tile boundaries), 2-stage ring (2 x 52 / 64 KB) + 16 KB epilogue scratch.  Same LDS image / swizzle / fragment layout /
// epilogues as the other kernels.
constexpr int b256_bytes() { return 256 * BK * 2; }                               // 32 KiB
constexpr int a256_bytes(int MI) { return 32 * MI * BK * 2; }                      // 20 / 32 KiB
constexpr int stage256_bytes(int MI) { return a256_bytes(MI) + b256_bytes(); }     // 52 / 64 KiB
// ring depth: 3 stages where they fit the 160 KiB (MI = 5: 156 KiB), else 2 (MI = 8: 128 KiB).  The epilogue needs no LDS
// (epilogue_direct, gemm_common.h)
constexpr int nstage256(int MI) { return 3 * stage256_bytes(MI) <= 160 * 1024 ? 3 : 2; }
// + 4 KiB behind the ring (and scratch): the target of LDS-DMA pieces that are issued unconditionally but have nothing to fetch
// (out-of-range offset -> zero fill), one KiB per wave -- cheaper than a branch around every piece
constexpr int spare256_off(int MI) { return nstage256(MI) * stage256_bytes(MI); }
constexpr int smem256_bytes(int MI) { return spare256_off(MI) + 4096; }
static_assert(smem256_bytes(5) <= 160 * 1024 && smem256_bytes(8) <= 160 * 1024, "LDS");

template <typename T> __device__ __forceinline__ void mfma_inplace(f32x4& c, typename T16<T>::v8 a, typename T16<T>::v8 b);
template <> __device__ __forceinline__ void mfma_inplace<f16_t>(f32x4& c, f16x8 a, f16x8 b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <> __device__ __forceinline__ void mfma_inplace<bf16_t>(f32x4& c, bf16x8 a, bf16x8 b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

template <typename T, int EPI, int MI>
__global__ __launch_bounds__(256, 1) void gemm_nt256_kernel(GemmP p) {
    constexpr int BMT = 32 * MI, A_B = a256_bytes(MI), STAGE = stage256_bytes(MI), NST = nstage256(MI), PER = MI + 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (p.N + 255) / 256;
    const int total_tiles = tiles_n * ((p.M + BMT - 1) / BMT);
    const int G = gridDim.x;
    const int my_tiles = (total_tiles - (int)blockIdx.x + G - 1) / G;
    const int nk = p.K / BK;
    const int iters = my_tiles * nk;
    if (iters <= 0) return;

    __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.bytesA);
    __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.bytesB);
    unsigned offA[8], offB[8];
    static_assert(MI <= 8, "offA");
    int st_tile = 0, st_kt = 0, st_slot = 0;
    auto tile_origin = [&](int seq, int& m0, int& n0) {
        const int r = xcd_remap((int)blockIdx.x + seq * G, total_tiles);
        m0 = (r / tiles_n) * BMT;
        n0 = (r % tiles_n) * 256;
    };
    auto set_offsets = [&](int t) {
        int m0, n0;
        tile_origin(t, m0, n0);
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            const int row = (wave * MI + j) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            offA[j] = (m0 + row < p.M) ? (unsigned)(((size_t)(m0 + row) * p.lda + c * 8) * 2) : EOE_OOB;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // the rows of each 64-row group permuted (eoe_direct_row) so that a lane's accumulator values are runs of consecutive output
            // columns and the lanes of a row write adjacent 16-byte pieces (epilogue_direct).  A permutation of whole rows: the LDS
            // image, its swizzle and the fragment reads are unchanged
            const int row = (wave * 8 + j) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            const int grow = (row & ~63) + eoe_direct_row(row & 63, p.out_f32);
            offB[j] = (n0 + grow < p.N) ? (unsigned)(((size_t)(n0 + grow) * p.ldb + c * 8) * 2) : EOE_OOB;
        }
    };
    // LDS-DMA as inline asm (buffer_load_dwordx4 ... offen lds; M0 = the wave-uniform LDS byte address of the 1-KiB piece, written in
    // the same statement that uses it): issued through the builtin, every LDS read the compiler can see -- and every asm
    // statement -- is preceded by s_waitcnt vmcnt(0) while a piece is in flight (the waitcnt pass cannot prove that the read
    // does not touch the slot being filled), which would shorten the DMA's lead from a whole k-tile to half of one.  The
    // pieces are awaited by hand: EOE_WAIT_VM(0) in front of the barrier that precedes the first read of the slot.
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_void_t*)smem);
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(wave);
#define EOE_DMA16(rsrc, lds_addr, voff)                                                                               \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"                            \
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory")
    // piece j of the staging cursor's k-tile (j < MI: the A image, else the B image).  `live` (wave-uniform): a piece with nothing to
    // fetch keeps its place in the instruction stream and in the vmcnt count -- out-of-range offset (zero fill), written to this wave's
    // KiB of the spare area
    const unsigned spare = lds0 + (unsigned)spare256_off(MI) + wave_u * 1024u;
    auto stage_piece = [&](int j, bool live) {
        const unsigned sa = lds0 + (unsigned)st_slot * STAGE;
        const unsigned k0 = live ? (unsigned)st_kt * (BK * 2u) : EOE_OOB;
        if (j < MI) {
            const unsigned la = live ? sa + (wave_u * MI + j) * 1024u : spare, vo = offA[j < MI ? j : 0] + k0;
            EOE_DMA16(ra, la, vo);
        } else {
            const unsigned lb = live ? sa + A_B + (wave_u * 8 + (j - MI)) * 1024u : spare, vo = offB[j >= MI ? j - MI : 0] + k0;
            EOE_DMA16(rb, lb, vo);
        }
    };
    auto stage_issue = [&]() {                     // the whole k-tile in one burst (prologue, and the k-tile deferred past an epilogue)
#pragma unroll
        for (int j = 0; j < PER; ++j) stage_piece(j, true);
    };
    auto stage_advance = [&]() {                   // the staging cursor moves on (every wave, after its pieces were issued)
        st_slot = (st_slot == NST - 1) ? 0 : st_slot + 1;
        if (++st_kt == nk) {
            st_kt = 0;
            st_tile += 1;
            if (st_tile < my_tiles) set_offsets(st_tile);
        }
    };
    auto stage_next = [&]() { stage_issue(); stage_advance(); };

    const int wm0 = (wave >> 1) * (16 * MI), wn0 = (wave & 1) * 128;
    const int lr = lane & 15, lg = lane >> 4;
    const int sw = (lr >> 1) & 7;
    const int fragA = (wm0 + lr) * 128, fragB = A_B + (wn0 + lr) * 128;
    const int ch0 = ((0 + lg) ^ sw) * 16, ch1 = ((4 + lg) ^ sw) * 16;
    typedef typename T16<T>::v8 V8;

    // Fragment reads are inline asm too: hipcc's waitcnt pass cannot tell a compiler-issued ds_read from the ring slot an
    // in-flight LDS-DMA is filling and puts s_waitcnt vmcnt(0) in front of the first read of every k-step, i.e. the k-tile
    // requested half an iteration ago must land before the current one may be read.  Their completion is awaited by
    // EOE_LANDED256: an lgkmcnt(0) that names every fragment register "+v", so that no use can be scheduled above it; each
    // set is awaited before the loop's back edge / before its MFMAs, so the compiler never touches a register in flight.
#define EOE_READ256(XA, WB, base, ks)                                                                            \
    do {                                                                                                         \
        const unsigned a_ = (unsigned)(size_t)((base) - smem) + fragA + ((ks) ? ch1 : ch0);                      \
        const unsigned b_ = (unsigned)(size_t)((base) - smem) + fragB + ((ks) ? ch1 : ch0);                      \
        _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                           \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(XA[i]) : "v"(a_), "i"(i * 2048));     \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                            \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(WB[i]) : "v"(b_), "i"(i * 2048));     \
    } while (0)
#define EOE_LANDED256(XA, WB)                                                                                    \
    do {                                                                                                         \
        if (MI == 5) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(XA[0]), "+v"(XA[1]), "+v"(XA[2]), "+v"(XA[3]), "+v"(XA[4]), \
                                  "+v"(WB[0]), "+v"(WB[1]), "+v"(WB[2]), "+v"(WB[3]), "+v"(WB[4]), "+v"(WB[5]), "+v"(WB[6]), "+v"(WB[7]) :: "memory"); \
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(XA[0]), "+v"(XA[1]), "+v"(XA[2]), "+v"(XA[3]), "+v"(XA[4 < MI ? 4 : 0]), \
                          "+v"(XA[5 < MI ? 5 : 0]), "+v"(XA[6 < MI ? 6 : 0]), "+v"(XA[7 < MI ? 7 : 0]),                      \
                          "+v"(WB[0]), "+v"(WB[1]), "+v"(WB[2]), "+v"(WB[3]), "+v"(WB[4]), "+v"(WB[5]), "+v"(WB[6]), "+v"(WB[7]) :: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    } while (0)
    // The MFMAs are inline asm with the accumulator tied in place ("+a": an AGPR quad that is both C and D).  Through the
    // builtin, with 160 accumulator registers carried around a loop that updates each of them twice, the register allocator
    // gave the first update a different destination and moved accumulators between AGPRs with ~6 v_accvgpr_* copies per MFMA.
    // Hazards: an MFMA taking the previous MFMA's D whole as its C needs no wait states; the fragments come from compiler-issued
    // ds_reads (the compiler waits for them before the statement that names them); the only other reader of the accumulators
    // is the epilogue, fenced by EOE_MFMA_DRAIN below.
#define EOE_MFMA256(XA, WB)                                               \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                     \
        _Pragma("unroll") for (int ni = 0; ni < 8; ++ni)                  \
            mfma_inplace<T>(acc[ni >> 2][mi][ni & 3], WB[ni], XA[mi]);
    // The cluster the k-loop uses: the MI * 8 MFMAs on (XA, WB) with, woven between them,
    //   * the MI + 8 fragment reads of the OTHER register set (RA, RB) from `rbase`, one after every RSTEP-th MFMA: a read
    //     issues in the shadow of the MFMA in front of it, and the four waves' LDS traffic is spread over the cluster instead
    //     of arriving in one burst behind the barrier;
    //   * (DMA = true) this wave's LDS-DMA pieces of the next k-tile behind the (wave_u + 1)-th quarter of the cluster.  The four
    //     SIMDs of a CU share ONE address / LDS-DMA path (a 1 KiB piece ~ 16 cycles of it): when every wave issued its 13 pieces
    //     right behind the barrier, the four of them queued for ~830 cycles with all matrix pipes idle (measured: 17 us of the
    //     73 us of the K = 3072 GEMM went away when the pieces were not issued at all); staggered, one wave issues while the
    //     other three SIMDs keep multiplying.
    //   * (round 3) the LDS-DMA pieces [P_LO, P_HI) of the staging cursor's k-tile, SPREAD evenly over the cluster, the four waves one
    //     MFMA apart (s_nop behind the barrier).  The four SIMDs of a CU share ONE LDS-DMA path (a 1 KiB piece ~ 16 cycles of it) and a
    //     wave sits at a piece until the path takes it: issued as bursts of 13 / 16 (round 2: one wave's burst per quarter of the
    //     cluster) the bursts of the four waves ran into each other and a piece cost its wave ~97 cycles -- 1550 cycles per k-tile with
    //     the matrix pipe idle (in-kernel stamps of the same scheme in gemm_tn256.hip: second half 2577 -> 1524 cycles); one piece per
    //     wave per 4-6 MFMAs costs ~30.  With the 3-stage ring the pieces of k-tile it+3 are spread over the second half of iteration
    //     it AND the first half of iteration it+1; with 2 stages over the second half only.
    constexpr int NMF = MI * 8, RSTEP = NMF / (MI + 8);
    constexpr int P1 = (NST == 3) ? (PER + 1) / 2 : PER;          // pieces issued in the second half; the rest in the next first half
#define EOE_CLUSTER256(XA, WB, RA, RB, rbase, rks, P_LO, P_HI, LIVE)                                            \
    do {                                                                                                         \
        const unsigned ra_ = (unsigned)(size_t)((rbase) - smem) + fragA + ((rks) ? ch1 : ch0);                   \
        const unsigned rb_ = (unsigned)(size_t)((rbase) - smem) + fragB + ((rks) ? ch1 : ch0);                   \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                        \
            _Pragma("unroll") for (int ni = 0; ni < 8; ++ni) {                                                   \
                const int idx = mi * 8 + ni;                                                                     \
                mfma_inplace<T>(acc[ni >> 2][mi][ni & 3], WB[ni], XA[mi]);                                       \
                if (idx % RSTEP == 0 && idx / RSTEP < MI + 8) {                                                  \
                    const int j = idx / RSTEP;                                                                   \
                    if (j < MI) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(RA[j < MI ? j : 0]) : "v"(ra_), "i"((j < MI ? j : 0) * 2048)); \
                    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(RB[j >= MI ? j - MI : 0]) : "v"(rb_), "i"((j >= MI ? j - MI : 0) * 2048)); \
                }                                                                                                \
                _Pragma("unroll") for (int sl = 0; sl < (P_HI) - (P_LO); ++sl)                                   \
                    if (idx == ((sl + 1) * NMF) / ((P_HI) - (P_LO)) - 1) stage_piece((P_LO) + sl, (LIVE));       \
            }                                                                                                    \
    } while (0)
#define EOE_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 15" ::: "memory")

    // counted waits: every k-tile is PER = MI + 8 LDS-DMA instructions per wave, nothing else is outstanding inside a tile's
    // k-loop (the epilogue's stores are drained with everything else at the top of each tile)
#define EOE_WAIT_GROUPS(g)                                                                             \
    do {                                                                                               \
        if ((g) <= 0) { EOE_WAIT_VM(0); }                                                              \
        else if ((g) == 1) { if (PER == 13) { EOE_WAIT_VM(13); } else { EOE_WAIT_VM(16); } }           \
        else { if (PER == 13) { EOE_WAIT_VM(26); } else { EOE_WAIT_VM(32); } }                         \
    } while (0)
    static_assert(PER == 13 || PER == 16, "EOE_WAIT_GROUPS knows MI = 5 and MI = 8");
    // the first wait of a tile that follows an epilogue: the epilogue's E stores (inline asm, counted like every vector-memory operation,
    // in issue order) sit between the pieces that must have landed and the younger ones that may stay in flight -- leave them in flight
    // too: vmcnt((NST - 2) * PER + E), capped at the counter's 63
#define EOE_WAIT_AFTER_EPILOGUE(E)                                                                     \
    do {                                                                                               \
        const int n_ = (NST - 2) * PER + (E);                                                          \
        if (n_ >= 63) { EOE_WAIT_VM(63); }                                                             \
        else if (n_ >= 61) { EOE_WAIT_VM(61); }                                                        \
        else if (n_ >= 53) { EOE_WAIT_VM(53); }                                                        \
        else if (n_ >= 41) { EOE_WAIT_VM(41); }                                                        \
        else if (n_ >= 40) { EOE_WAIT_VM(40); }                                                        \
        else if (n_ >= 33) { EOE_WAIT_VM(33); }                                                        \
        else if (n_ >= 32) { EOE_WAIT_VM(32); }                                                        \
        else if (n_ >= 20) { EOE_WAIT_VM(20); }                                                        \
        else { EOE_WAIT_GROUPS(NST - 2); }                                                             \
    } while (0)
    V8 xa0[MI], wb0[8], xa1[MI], wb1[8];
    set_offsets(0);
    const int pre = iters < NST ? iters : NST;
    for (int i = 0; i < pre; ++i) stage_next();
    EOE_WAIT_GROUPS(pre - 1);                       // k-tile 0 landed; the later ones may stay in flight
    __builtin_amdgcn_s_barrier();
    EOE_READ256(xa0, wb0, smem, 0);
    EOE_LANDED256(xa0, wb0);
    int cur = 0, it = 0;
    // diagnostics (EOE_GEMM_STAMP=1): wave 0's cycles per workgroup -- [0] kernel entry -> first operands landed, [1] sum over the
    // k-tiles of (first half + DMA wait), [2] at the barrier, [3] second half, [4] epilogues (issue only), [5] whole kernel, [6] tiles
    unsigned long long* stp = (p.stamp && wave == 0) ? p.stamp + (size_t)blockIdx.x * 16 : nullptr;
    unsigned long long t_a = 0, t_b = 0, t_c = 0, c_first = 0, c_bar = 0, c_second = 0, c_epi = 0, t_entry = 0;
    if (stp) { t_entry = __builtin_amdgcn_s_memtime(); }
    bool part = false;                             // the staging cursor's k-tile has its first P1 pieces issued, the rest are due
    int stores_in_flight = 0;                      // the previous tile's epilogue stores (per wave), for the first wait of this tile
    // two nested loops (tile, k-tile) over the flattened iteration space `it`: the accumulators live from their zero
    // initialisation to the tile's epilogue and are not carried around the outer loop
    if (stp) { t_c = __builtin_amdgcn_s_memtime(); if (lane == 0) stp[0] = t_c - t_entry; }
    for (int c_tile = 0; c_tile < my_tiles; ++c_tile) {
        f32x4 acc[2][MI][4];                        // [64-column half][16-row tile][16-column tile]
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[h][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // (no drain here: the epilogue's stores are inline asm and its few loads were consumed there, so the compiler's waitcnt pass has
        //  nothing pending to merge into the loop header)
        for (int kt = 0; kt < nk; ++kt, ++it) {
            const int nxt = (cur == NST - 1) ? 0 : cur + 1;
            const char* sc = smem + cur * STAGE;
            const char* sn = smem + nxt * STAGE;
            // first half: MFMA(F0) with the reads of F1 (k-step 1 of this k-tile) and -- 3-stage ring -- the second part of the pieces
            // the previous second half began (dead pieces if it began none)
            EOE_CLUSTER256(xa0, wb0, xa1, wb1, sc, 1, P1, PER, part);
            if (P1 < PER && part) { stage_advance(); part = false; }
            // k-tile it+1 has landed; the k-tile staged after it (3 stages) stays in flight.  Every iteration issues exactly PER
            // pieces, live or dead, so the count does not depend on where in the sequence we are
            if (kt == 0 && stores_in_flight) { EOE_WAIT_AFTER_EPILOGUE(stores_in_flight); } else { EOE_WAIT_GROUPS(NST - 2); }
            EOE_LANDED256(xa1, wb1);               // this wave's reads of slot `cur` are complete
            if (stp) { t_a = __builtin_amdgcn_s_memtime(); c_first += t_a - t_c; }
            __builtin_amdgcn_s_barrier();
            if (stp) { t_b = __builtin_amdgcn_s_memtime(); c_bar += t_b - t_a; }
            // the four waves one MFMA (16 cycles = one piece on the shared LDS-DMA path) apart until the next barrier
            if (wave_u & 1) asm volatile("s_nop 15" ::: "memory");
            if (wave_u & 2) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            // second half: MFMA(F1) with the reads of the next k-tile's F0 and the first P1 pieces of k-tile it+NST (into the slot just
            // consumed; dead pieces past the last k-tile)
            const bool dma = it + NST < iters && !(p.dbg & 2);
            EOE_CLUSTER256(xa1, wb1, xa0, wb0, sn, 0, 0, P1, dma);
            if (P1 == PER) { if (dma) stage_advance(); } else { part = dma; }
            EOE_LANDED256(xa0, wb0);               // before the back edge: no fragment register is in flight across it
            if (stp) { t_c = __builtin_amdgcn_s_memtime(); c_second += t_c - t_b; }
            cur = nxt;
        }
        // epilogue while the next tile's first k-tiles are in flight (their fragments of k-step 0 are already in xa0 / wb0); no LDS, no
        // drain: the stores stay in flight under the next tile's first MFMAs
        EOE_MFMA_DRAIN();                           // the last MFMAs' results must have landed before the accumulators are read
        int m0, n0;
        tile_origin(c_tile, m0, n0);
        GemmP ep;
        load_epilogue_args(ep, p);
        // buffer descriptors for the asm stores / bounds-checked loads (the epilogue arguments come through vector registers:
        // readfirstlane makes pointers and sizes provably wave-uniform, i.e. SGPR operands)
        auto uni_ptr = [](const void* q) -> const void* {
            const unsigned long long v = (unsigned long long)(uintptr_t)q;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
            return (const void*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
        };
        auto uni_u32 = [](size_t b) -> unsigned { return (unsigned)__builtin_amdgcn_readfirstlane((unsigned)b); };
        const size_t esz = ep.out_f32 ? 4 : 2;
        const size_t c_elems = ((size_t)ep.M - 1) * ep.ldc + ep.N, a_elems = ((size_t)ep.M - 1) * ep.ldaux + ep.N;
        __amdgpu_buffer_rsrc_t rc = make_rsrc(uni_ptr(ep.C), uni_u32(c_elems * esz));
        __amdgpu_buffer_rsrc_t rpre = make_rsrc(uni_ptr(ep.aux_out ? ep.aux_out : ep.C), uni_u32(ep.aux_out ? c_elems * 2 : 0));
        __amdgpu_buffer_rsrc_t raux = make_rsrc(uni_ptr(ep.aux ? ep.aux : ep.C), uni_u32(ep.aux ? a_elems * (EPI == EOE_EPI_RESIDUAL ? 4 : 2) : 0));
        const size_t part_rows = (size_t)((ep.M + 16 * MI - 1) / (16 * MI));
        __amdgpu_buffer_rsrc_t rpart = make_rsrc(uni_ptr(ep.colsum_part ? (const void*)ep.colsum_part : ep.C),
                                                 uni_u32(ep.colsum_part ? part_rows * (size_t)ep.N * 4 : 0));
        asm volatile("s_nop 4" ::: "memory");      // SGPRs fresh from readfirstlane -> a buffer instruction's descriptor (section 5.7, item 2)
        epilogue_direct<T, EPI, MI>(ep, acc, m0 + wm0, n0 + wn0, lane, rc, rpre, raux, rpart);
        stores_in_flight = epilogue_direct_stores(ep, EPI, MI);
        if (stp) { const unsigned long long t_e = __builtin_amdgcn_s_memtime(); c_epi += t_e - t_c; t_c = t_e; }
    }
    if (stp && lane == 0) { stp[1] = c_first; stp[2] = c_bar; stp[3] = c_second; stp[4] = c_epi; stp[5] = t_c - t_entry; stp[6] = (unsigned long long)my_tiles; }
#undef EOE_WAIT_AFTER_EPILOGUE
#undef EOE_WAIT_GROUPS
#undef EOE_DMA16
#undef EOE_READ256
#undef EOE_LANDED256
#undef EOE_MFMA256
#undef EOE_CLUSTER256
#undef EOE_MFMA_DRAIN
}


extern int g_nt_flags;

template <typename T, int MI>
int launch_nt256(const GemmP& p, int epi, hipStream_t s) {
    const int tiles = cdiv(p.M, 32 * MI) * cdiv(p.N, 256);
    const int ncu = num_cus();
    const int grid = tiles < ncu ? tiles : ncu;
#define EOE_NT256_CASE(E)                                                                   \
    case E:                                                                                 \
        { static bool once = (hipFuncSetAttribute((const void*)gemm_nt256_kernel<T, E, MI>, hipFuncAttributeMaxDynamicSharedMemorySize, smem256_bytes(MI)), true); (void)once; } \
        hipLaunchKernelGGL((gemm_nt256_kernel<T, E, MI>), dim3(grid), dim3(256), smem256_bytes(MI), s, p); \
        break;
    switch (epi) {
        EOE_NT256_CASE(EOE_EPI_NONE)
        EOE_NT256_CASE(EOE_EPI_GELU)
        EOE_NT256_CASE(EOE_EPI_RESIDUAL)
        EOE_NT256_CASE(EOE_EPI_GELU_BWD)
        default: return eoe_set_error(EOE_ERR_ARG, "gemm_nt: unknown epilogue %d", epi);
    }
#undef EOE_NT256_CASE
    EOE_CHECK_LAUNCH("gemm_nt256");
    return finish_colsum(p, epi, MI, s);
}

}  // namespace

int eoe_launch_nt256(const void* gemm_p, int dtype, int epi, int mi, hipStream_t s) {
    const GemmP& p = *(const GemmP*)gemm_p;
#ifdef EOE_DEV256          // development builds: one instantiation family only (compile time)
    return launch_nt256<f16_t, 5>(p, epi, s);
#else
    if (dtype == EOE_F16) return mi == 8 ? launch_nt256<f16_t, 8>(p, epi, s) : launch_nt256<f16_t, 5>(p, epi, s);
    return mi == 8 ? launch_nt256<bf16_t, 8>(p, epi, s) : launch_nt256<bf16_t, 5>(p, epi, s);
#endif
}
