// NT GEMM on 256 x 256 x 64 tiles with EIGHT waves (two per SIMD, 128 x 64 of the tile each), one persistent workgroup per CU (round 4).
//
// Why this shape (measured this round, tools/ntp_check.py with EOE_GEMM_DEBUG = 1 / 4 / 8 / 12 and tools/ntp_stamps.py):
//   * a CU takes in its operands at ~47 GB/s at best -- the rate of the L2 -> LDS path under an MFMA loop, found with every tile staging the
//     SAME (L2-resident) panels; with loads that fetch nothing the producer / consumer kernel (gemm_pc.hip) runs the in-projection at
//     1135 TF, with L2-resident panels at 856, from HBM at 770.  A kernel's k-loop therefore runs at (flop per staged byte) x ~47 GB/s per CU:
//     160 x 128 tiles (71 flop/B) sit AT that bound, 128 x 256 (85) barely above it, 256 x 256 (128) halves the bytes per flop;
//   * one wave per SIMD cannot issue fast enough (gemm_pc.hip's header): two waves per SIMD are needed, and 256 registers per wave then
//     allow 128 accumulators -- 128 x 64 per wave, eight waves for a 256 x 256 tile.
// Structure: 2 x 64 KB LDS ring filled by LDS-DMA (every wave stages 4 A + 4 B pieces per k-tile, woven into the second half of the
// iteration), ONE workgroup barrier per k-tile, fragments double-buffered in registers, accumulators in literal AGPRs a[0:127] (quad
// e = 4 mi + ni -> a[4e : 4e+3]; reserved by a clobber list on every MFMA statement: gemm_ntp.hip's finding), epilogue in the open
// straight from the accumulators (B's rows permuted at staging, eoe_direct_row: 16-byte row-contiguous stores), the next tile's first two
// k-tiles in flight under it.  The tile's bias row reaches LDS by one extra piece of wave 0 (two alternating slots).
// Same products in the same k order as every other NT kernel, same epilogue arithmetic: bitwise the same results.
#include "gemm_common.h"
#include <type_traits>

int eoe_nt_flags();             // gemm.hip: the nt_flags option

namespace {

constexpr int W8_A_BYTES = 256 * BK * 2;                                               // 32 KiB
constexpr int W8_STAGE_BYTES = 2 * W8_A_BYTES;                                         // 64 KiB
constexpr int W8_NST = 2;
constexpr int W8_BIAS_OFF = W8_NST * W8_STAGE_BYTES;                                   // behind the ring: 2 slots x 1 KiB
constexpr int W8_SYNC_OFF = W8_BIAS_OFF + 2 * 1024;                                   // stream-K: ticket / wait words (64 B)
constexpr int W8_SMEM_BYTES = W8_SYNC_OFF + 64;
static_assert(W8_SMEM_BYTES <= 160 * 1024, "LDS");

#define W8_A10(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"
#define W8_AGPRS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", W8_A10(1), W8_A10(2), W8_A10(3), W8_A10(4), W8_A10(5), W8_A10(6), \
    W8_A10(7), W8_A10(8), W8_A10(9), W8_A10(10), W8_A10(11), "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
#define W8_MFMA_INPLACE(MNEM, E, A, B) asm volatile(MNEM " a[%c0:%c1], %2, %3, a[%c0:%c1]" :: "i"(4 * (E)), "i"(4 * (E) + 3), "v"(A), "v"(B) : W8_AGPRS)
#define W8_MFMA_FIRST(MNEM, E, A, B) asm volatile(MNEM " a[%c0:%c1], %2, %3, 0" :: "i"(4 * (E)), "i"(4 * (E) + 3), "v"(A), "v"(B) : W8_AGPRS)

// cache policy of the output stores: 0 = default, 2 = nt (non-temporal).  Stand-alone the streaming hint is worth 2 us of 55 (the outputs, 59 -
// 157 MB per launch, then do not evict the operand panels the other workgroups are about to re-read: PMC, profiles/r4/gemm_pmc_*.txt).  In
// the training step it LOSES: the in-projection's output is what the attention kernel reads next, and with the hint that kernel finds nothing in
// L2 / the Infinity Cache -- attn_fwd 0.299 -> 0.236 ms per step without it against +0.03 for the GEMMs (tools/cflags_ab.sh -DW8_ST_AUX=2,
// three interleaved pairs: 10.116 / 10.085 ms per step).  Default 0 since the end of round 4.
#ifndef W8_ST_AUX
#define W8_ST_AUX 0
#endif
template <int N> __device__ __forceinline__ void w8_wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// EPI: EOE_EPI_NONE (16-bit C = alpha acc + bias) or EOE_EPI_GELU (pre -> aux_out, C = QuickGELU of the rounded pre)
template <typename T, int EPI, bool SK>
__global__ __launch_bounds__(512, 2) void gemm_w8_kernel(GemmP p) {
    static_assert(EPI == EOE_EPI_NONE || EPI == EOE_EPI_GELU, "epilogues with a second input are not built yet");
    constexpr int A_B = W8_A_BYTES, STAGE = W8_STAGE_BYTES, NST = W8_NST;
    constexpr int PER = 8;                             // LDS-DMA pieces per wave and k-tile: 4 of the A image, 4 of the B image
    constexpr int NMF = 32;                            // MFMAs per cluster (one 32-deep k-step of a wave's 128 x 64) = accumulator quads
    constexpr int NB = 8;                              // bands of a wave's tile = its 16-row tiles
    constexpr int ES = (EPI == EOE_EPI_GELU) ? 2 : 1;  // stores per half band
    constexpr int L = (EPI == EOE_EPI_GELU) ? 66 : 13; // micro-operations per half band (epi_op below)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    asm volatile("" ::: W8_AGPRS);                     // the kernel descriptor allocates a0..a127
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = p.N >> 8;                      // N % 256 == 0 (host)
    const int total_tiles = tiles_n * ((p.M + 255) >> 8);
    const int G = gridDim.x;
    const int nk = p.K / BK;                           // >= 2 (host; stream-K: >= 4)
    // ---- the workgroup's segments.  Data-parallel form (SK false): tiles b, b + G, ... each over the whole of K.  Stream-K form (the file's
    // header): sk_rounds full rounds of such tiles, then this workgroup's range [sk_lo, sk_hi) of its XCD's (tile, k-tile) space
    const int xcd = (int)blockIdx.x & 7, wq = (int)blockIdx.x >> 3, Wn = G >> 3;
    const int sk_R = SK ? p.sk_rounds : 0;
    const int sk_Ix = SK ? ((total_tiles - sk_R * G - xcd + 7) >> 3) * nk : 0;      // this XCD's stream-K tiles x k-tiles
    auto sk_bound = [&](int i) -> int {                // first (tile, k-tile) index of workgroup i of the XCD: never 1 k-tile from a tile edge
        int v = (int)(((long long)i * sk_Ix) / Wn);
        const int m = v % nk;
        if (m == 1) v -= 1;
        else if (m == nk - 1) v += 1;
        return v;
    };
    auto sk_wg_of = [&](int v) -> int {                // the workgroup (of this XCD) whose range holds index v
        int g = (int)(((long long)v * Wn) / (sk_Ix > 0 ? sk_Ix : 1));
        if (g > Wn - 1) g = Wn - 1;
        while (g + 1 < Wn && sk_bound(g + 1) <= v) ++g;
        while (g > 0 && sk_bound(g) > v) --g;
        return g;
    };
    const int sk_lo = SK ? sk_bound(wq) : 0, sk_hi = SK ? sk_bound(wq + 1) : 0;
    const int sk_t0 = sk_lo / nk;
    const int n_sk = sk_hi > sk_lo ? (sk_hi - 1) / nk - sk_t0 + 1 : 0;
    const int my_tiles = SK ? sk_R + n_sk : (total_tiles - (int)blockIdx.x + G - 1) / G;      // segments
    const int iters = SK ? sk_R * nk + (sk_hi - sk_lo) : my_tiles * nk;
    if (iters <= 0) return;
    const bool sk_first = SK && (p.sk_flags & 1);      // the stream-K segments before the data-parallel tiles
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_void_t*)smem);
    // segment j: virtual block id (-> tile), k-tiles [kb, ke); js = its index among the stream-K segments or -1
    auto seg_get = [&](int j, int& vb, int& kb, int& ke, int& js) {
        js = -1;
        if (SK) {
            const bool is_sk = sk_first ? (j < n_sk) : (j >= sk_R);
            if (is_sk) {
                js = sk_first ? j : j - sk_R;
                const int t = sk_t0 + js;
                kb = js == 0 ? sk_lo - sk_t0 * nk : 0;
                ke = sk_hi - t * nk < nk ? sk_hi - t * nk : nk;
                vb = sk_R * G + t * 8 + xcd;
                return;
            }
            if (sk_first) j -= n_sk;
        }
        vb = (int)blockIdx.x + j * G;
        kb = 0;
        ke = nk;
    };
    auto tile_of = [&](int vb, int& m0, int& n0) {
        const int r = xcd_remap(vb, total_tiles);
        m0 = (r / tiles_n) * 256;
        n0 = (r % tiles_n) * 256;
    };

    // ------------------------------------------------------------------------------------------------ staging
    __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.bytesA);
    __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.bytesB);
    __amdgpu_buffer_rsrc_t rbias = make_rsrc(p.bias ? (const void*)p.bias : p.C, p.bias ? (unsigned)p.N * 4u : 0u);
    // a piece is 8 rows x 128 B; lane -> (row lane >> 3, 16-byte slot lane & 7 holding chunk slot ^ ((row >> 1) & 7)).  Wave w stages rows
    // 32 w .. 32 w + 31 of both images (pieces j = 0..3); the swizzle term of piece j depends on j's parity only -> two per-lane offsets
    // per operand (VGPRs) + a uniform offset per piece (the instruction's soffset).  B rows are permuted inside each 64-row group
    // (eoe_direct_row, 16-bit C): image row 64 g + 32 (w & 1) + 8 j + l8 holds weight row 64 g + 32 (w & 1) + fj(j) + gl(lane)
    const int l8 = lane >> 3;
    unsigned voffA[2], voffB[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int c = (lane & 7) ^ ((par * 4 + (lane >> 4)) & 7);
        voffA[par] = (unsigned)((l8 * p.lda + c * 8) * 2);
        voffB[par] = (unsigned)((((lane >> 5) * 8 + (l8 & 3)) * p.ldb + c * 8) * 2);
    }
    int st_tile = 0, st_kt = 0, st_kb = 0, st_kend = nk, st_slot = 0;
    unsigned sA_base = 0, sB_base = 0, bias_vo = EOE_OOB;
    int rows_left = 0;
    auto set_offsets = [&](int t) {
        int m0, n0, vb, js;
        seg_get(t, vb, st_kb, st_kend, js);
        st_kt = st_kb;
        tile_of(vb, m0, n0);
        if (p.dbg & 4) m0 = 0;                         // diagnostics (EOE_GEMM_DEBUG=4 / 8): every tile stages the first A / B panel
        if (p.dbg & 8) n0 = 0;
        const int ra0 = m0 + (int)wave_u * 32;
        sA_base = (unsigned)ra0 * (unsigned)p.lda * 2u;
        rows_left = p.M - ra0;
        sB_base = (unsigned)(n0 + (int)wave_u * 32) * (unsigned)p.ldb * 2u;
        bias_vo = (unsigned)((n0 + lane * 4) * 4);      // wave 0: the tile's 256 bias values = 64 lanes x 16 B
    };
#define W8_DMA16(rsrc, lds_addr, voff, soff)                                                                          \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"                           \
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory")
    // piece j (0..3: the A image, 4..7: the B image) of the staging cursor's k-tile.  Rows of A at or past M: out-of-range offset (zeros)
    // `live` false (past the workgroup's last k-tile): the piece keeps its place in the instruction stream; it fetches nothing and lands
    // (zeros) in the slot that would have been refilled, which nobody reads any more
    auto stage_piece = [&](int j, bool live) {
        const unsigned sa = lds0 + (unsigned)st_slot * STAGE;
        const unsigned k0 = (unsigned)st_kt * (BK * 2u);
        if (j < 4) {
            const unsigned la = sa + (wave_u * 4 + j) * 1024u;
            const unsigned vo = (live && l8 + j * 8 < rows_left) ? voffA[j & 1] : EOE_OOB;
            const unsigned so = sA_base + (unsigned)(j * 8) * (unsigned)p.lda * 2u + k0;
            W8_DMA16(ra, la, vo, so);
        } else {
            const int jb = j - 4;
            const unsigned lb = sa + A_B + (wave_u * 4 + jb) * 1024u;
            const unsigned so = sB_base + (unsigned)((jb & 1) * 16 + ((jb >> 1) & 1) * 4) * (unsigned)p.ldb * 2u + k0;
            const unsigned vo = live ? voffB[jb & 1] : EOE_OOB;
            W8_DMA16(rb, lb, vo, so);
        }
    };
    // before the first piece of a tile's first k-tile: the tile's bias row (wave 0; into the slot of the tile's parity)
    auto stage_bias = [&](bool live) {
        if (live && wave_u == 0 && st_kt == st_kb) {
            const unsigned la = lds0 + (unsigned)W8_BIAS_OFF + (unsigned)(st_tile & 1) * 1024u;
            W8_DMA16(rbias, la, bias_vo, 0);
        }
    };
    auto stage_advance = [&]() {
        st_slot ^= 1;
        if (++st_kt == st_kend) {
            st_tile += 1;
            if (st_tile < my_tiles) set_offsets(st_tile);
        }
    };

    // ------------------------------------------------------------------------------------------------ fragments
    const int wm0 = (int)(wave_u >> 2) * 128, wn0 = (int)(wave_u & 3) * 64;
    const int lr = lane & 15, lg = lane >> 4;
    const int sw = (lr >> 1) & 7;
    const int fragA = (wm0 + lr) * 128, fragB = A_B + (wn0 + lr) * 128;
    const int ch0 = ((0 + lg) ^ sw) * 16, ch1 = ((4 + lg) ^ sw) * 16;
    typedef typename T16<T>::v8 V8;
    // ------------------------------------------------------------------------------------------------ the epilogue's pieces
    const unsigned c_bytes = (unsigned)((((size_t)p.M - 1) * p.ldc + p.N) * 2);
    __amdgpu_buffer_rsrc_t rc = make_rsrc(p.C, c_bytes);
    __amdgpu_buffer_rsrc_t rpre = make_rsrc(p.aux_out ? p.aux_out : p.C, p.aux_out ? c_bytes : 0u);
    const float alpha = p.alpha;
    const int ldc2 = p.ldc * 2;
    const unsigned lane_off = (unsigned)(lr * ldc2 + lg * 16);      // this lane's row and 8-column run inside a band's 16 x 32 block (bytes)
    // ------------------------------------------------------------------------------------------------ one cluster
    // FIRST: acc = W X (first k-step of a tile), else acc += W X.  32 MFMAs, mi-major (8 rows of 4).  The A fragments are SINGLE-buffered:
    // row mi's fragment is dead after its 4 MFMAs and is re-read for the NEXT k-step right behind them, 28 MFMAs before it is needed
    // again; the 4 B fragments are double-buffered and re-read behind the first four MFMAs (96 fragment registers did not fit beside the
    // staging and epilogue state: 55 spilled).  LDS reads return in order, so the waits are counted: in front of row 0 the 7 youngest reads
    // (rows 1..7 of the previous cluster) may be in flight, in front of row mi >= 1 the 11 youngest (7 - mi of the previous cluster, this
    // cluster's 4 B reads and mi A reads).  STAGEP: this wave's 8 LDS-DMA pieces of k-tile kt + 2, one behind every fourth MFMA.
#define W8_CLUSTER(FIRST, STAGEP, LIVE, READS, WBC, WBN, rbase, rks)                                                          \
    do {                                                                                                         \
        const unsigned ra_ = (unsigned)(size_t)((rbase) - smem) + fragA + ((rks) ? ch1 : ch0);                   \
        const unsigned rb_ = (unsigned)(size_t)((rbase) - smem) + fragB + ((rks) ? ch1 : ch0);                   \
        _Pragma("unroll") for (int mi = 0; mi < 8; ++mi) {                                                       \
            if (mi == 0) asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(xa[0]), "+v"(WBC[0]), "+v"(WBC[1]), "+v"(WBC[2]), "+v"(WBC[3]) :: "memory"); \
            else asm volatile("s_waitcnt lgkmcnt(11)" : "+v"(xa[mi]) :: "memory");                              \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) {                                                   \
                const int e_ = mi * 4 + ni;                                                                      \
                if (std::is_same<T, f16_t>::value) {                                                             \
                    if (FIRST) W8_MFMA_FIRST("v_mfma_f32_16x16x32_f16", e_, WBC[ni], xa[mi]);                    \
                    else W8_MFMA_INPLACE("v_mfma_f32_16x16x32_f16", e_, WBC[ni], xa[mi]);                        \
                } else {                                                                                         \
                    if (FIRST) W8_MFMA_FIRST("v_mfma_f32_16x16x32_bf16", e_, WBC[ni], xa[mi]);                   \
                    else W8_MFMA_INPLACE("v_mfma_f32_16x16x32_bf16", e_, WBC[ni], xa[mi]);                       \
                }                                                                                                \
                if ((READS) && mi == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(WBN[ni]) : "v"(rb_), "i"(ni * 2048)); \
            }                                                                                                    \
            if (READS) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xa[mi]) : "v"(ra_), "i"(mi * 2048)); \
            if (STAGEP) stage_piece(mi, LIVE);                                                                     \
        }                                                                                                        \
    } while (0)
    // one k-tile iteration (2-stage ring): k-step 0 (+ the reads of k-step 1 from the same slot); every read of the slot and k-tile it + 1
    // have landed (its pieces were issued one iteration ago; WAITN: the stores of an epilogue in between stay in flight), barrier; k-step 1
    // + the reads of the next k-tile's k-step 0 + the pieces of k-tile it + 2 into the slot just consumed
#define W8_ITER(FIRST, LASTK)                                                                                    \
    do {                                                                                                         \
        const char* sc = smem + cur * STAGE;                                                                     \
        const char* sn = smem + (cur ^ 1) * STAGE;                                                               \
        W8_CLUSTER(FIRST, false, false, true, wb0, wb1, sc, 1);                                                  \
        if ((FIRST) && c_tile > 0) w8_wait_vm<NB * 2 * ES>(); else w8_wait_vm<0>();                              \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3]), "+v"(xa[4]), "+v"(xa[5]), "+v"(xa[6]), \
                     "+v"(xa[7]), "+v"(wb1[0]), "+v"(wb1[1]), "+v"(wb1[2]), "+v"(wb1[3]) :: "memory");           \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        __builtin_amdgcn_s_barrier();                                                                            \
        const bool live_ = it + 2 < iters;                                                                       \
        stage_bias(live_);                                                                                       \
        W8_CLUSTER(false, true, live_, !(LASTK), wb1, wb0, sn, 0);                                               \
        if (live_) stage_advance();                                                                              \
        cur ^= 1;                                                                                                \
        ++it;                                                                                                    \
    } while (0)

    V8 xa[8], wb0[4], wb1[4];
    int cur = 0, it = 0;
    // prologue: k-tiles 0 and 1 (iters >= 2)
    set_offsets(0);
#pragma unroll 1
    for (int i = 0; i < 2; ++i) {
        stage_bias(true);
#pragma unroll
        for (int j = 0; j < PER; ++j) stage_piece(j, true);
        stage_advance();
    }
    w8_wait_vm<PER>();                                 // k-tile 0 (and the first bias row) landed
    __builtin_amdgcn_s_barrier();
    // the fragments of a tile's first k-step (slot `cur`).  Not prefetched under the previous tile's epilogue: 48 live registers there made
    // the compiler spill the epilogue's state
#define W8_FIRST_FRAGS()                                                                                         \
    do {                                                                                                         \
        const unsigned a_ = (unsigned)(cur * STAGE) + fragA + ch0, b_ = (unsigned)(cur * STAGE) + fragB + ch0;   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wb0[i]) : "v"(b_), "i"(i * 2048)); \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xa[i]) : "v"(a_), "i"(i * 2048)); \
    } while (0)
    // ---- stream-K: the partial-accumulator slots (fragment order: quad e of wave w = 1 KiB at (32 w + e) KiB / 1024; 256 KiB per slot; two
    // slots per workgroup: its first stream-K segment and its last) and the routines that move a wave's 32 quads
    __amdgpu_buffer_rsrc_t rpart = make_rsrc(SK ? (const void*)p.sk_part : p.C, SK ? (unsigned)(2 * G) * 262144u : 0u);
    const unsigned part_voff = (unsigned)lane * 16u;
    auto part_store = [&](int slot) {                  // straight from the accumulation registers
        const int so = slot * 262144 + (int)wave_u * 32768;
#pragma unroll
        for (int e = 0; e < 32; ++e)
            asm volatile("buffer_store_dwordx4 a[%c0:%c1], %2, %3, %4 offen sc1" :: "i"(4 * e), "i"(4 * e + 3), "v"(part_voff), "s"(rpart), "s"(so + e * 1024) : "memory");
    };
    auto part_load = [&](int slot) {                   // accumulators = the slot
        const int so = slot * 262144 + (int)wave_u * 32768;
#pragma unroll
        for (int e = 0; e < 32; ++e)
            asm volatile("buffer_load_dwordx4 a[%c0:%c1], %2, %3, %4 offen sc1" :: "i"(4 * e), "i"(4 * e + 3), "v"(part_voff), "s"(rpart), "s"(so + e * 1024) : "memory", W8_AGPRS);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory", W8_AGPRS);
    };
    auto part_add = [&](int slot) {                    // accumulators += the slot: 8 quads in flight (the compiler counts these loads itself:
        const int so = slot * 262144 + (int)wave_u * 32768;       // every memory operation it cannot see is older)
        constexpr int DEPTH = 8;
        f32x4 ld[DEPTH];
#pragma unroll
        for (int e = 0; e < DEPTH; ++e) ld[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rpart, (int)part_voff, so + e * 1024, 16));
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const f32x4 v = ld[e % DEPTH];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a;
                asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(a) : "i"(4 * e + r));
                a += v[r];
                asm volatile("v_accvgpr_write_b32 a[%c0], %1" :: "i"(4 * e + r), "v"(a) : W8_AGPRS);
            }
            if (e + DEPTH < 32) ld[e % DEPTH] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rpart, (int)part_voff, so + (e + DEPTH) * 1024, 16));
        }
    };
    for (int c_tile = 0; c_tile < my_tiles; ++c_tile) {
        int c_vb, c_kb, c_ke, c_js;
        seg_get(c_tile, c_vb, c_kb, c_ke, c_js);
        // behind an epilogue its NB * 2 * ES stores are younger than the pieces of this k-tile + 1: they stay in flight
        W8_FIRST_FRAGS();
        W8_ITER(true, false);
        _Pragma("unroll 1") for (int kt = c_kb + 1; kt < c_ke - 1; ++kt) W8_ITER(false, false);
        W8_ITER(false, true);
        // ---- epilogue (the next tile's first k-step is already in xa0 / wb0, its first two k-tiles staged or in flight)
        int m0, n0;
        tile_of(c_vb, m0, n0);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");        // the last MFMAs' results have landed
        __builtin_amdgcn_sched_barrier(0);
        if (SK && c_ke - c_kb < nk) {
            // ---- a split tile: its S segments (consecutive workgroups of this XCD) meet.  Whoever arrives LAST owns the tile (gemm_tn256.hip's
            // rule: the owner only ever waits for workgroups that have arrived, i.e. are resident and a few microseconds from done -- no
            // co-residency assumption); the others leave their accumulators in their slot and raise its flag.  The owner adds the segments in
            // SEGMENT order whoever it is (me <= 1: own + p0 == p0 + own, then the rest; me >= 2: its own accumulators go through its slot as
            // well): bitwise reproducible.  The owner leaves the ticket counter and the flags zeroed for the next launch.
            const int t = sk_t0 + c_js;
            const int w_first = sk_wg_of(t * nk), w_last = sk_wg_of(t * nk + nk - 1);
            const int S = w_last - w_first + 1, me = wq - w_first;
            int* arrive = p.sk_sync + (t * 8 + xcd);
            int* flags = p.sk_sync + 1024;
            const int my_slot = (int)blockIdx.x * 2 + (c_js == 0 ? 0 : 1);
            auto slot_of = [&](int sgm) -> int {       // segment sgm of this tile: workgroup w_first + sgm of the XCD; its first segment iff its range starts here
                const int wc = w_first + sgm;
                return (wc * 8 + xcd) * 2 + ((sgm > 0 || sk_bound(wc) >= t * nk) ? 0 : 1);
            };
            volatile int* sw = (volatile int*)(smem + W8_SYNC_OFF);
            if (tid == 0) sw[0] = __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const bool owner = __builtin_amdgcn_readfirstlane(sw[0]) == S - 1;
            if (!owner) {
                part_store(my_slot);
                // publish (cdna_hip_programming.md, Guideline 16, R1 with write-through payload): the slot's stores carry sc1, every storing
                // wave drains, the workgroup meets, ONE lane raises the flag with an agent-scope (sc1) store.  No release fence: it would write
                // back the XCD's whole L2 -- megabytes of freshly written output tiles (measured: in-projection 56 -> 72 us with it)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(flags + my_slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                continue;
            }
            if (me >= 2) {
                part_store(my_slot);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (tid == 0) {
                __hip_atomic_store(arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int bad = 0;
                for (int sgm = 0; sgm < S; ++sgm) {
                    if (sgm == me) continue;
                    int* f = flags + slot_of(sgm);
                    int spins = 0;
                    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1 && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(4);
                    if (spins >= (1 << 22)) bad = 1;
                    __hip_atomic_store(f, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                sw[1] = bad;                           // (no acquire fence: EVERY load of a slot below is an sc1 load -- the guide's table, row 1)
            }
            __syncthreads();
            const bool poisoned = __builtin_amdgcn_readfirstlane(sw[1]) != 0;
            if (me >= 2) part_load(slot_of(0));
            for (int sgm = me >= 2 ? 1 : 0; sgm < S; ++sgm) {
                if (me < 2 && sgm == me) continue;
                part_add(sgm == me ? my_slot : slot_of(sgm));
            }
            if (poisoned) {                             // loud, not silent
                const float qnan = __builtin_nanf("");
#pragma unroll
                for (int e = 0; e < 128; ++e) asm volatile("v_accvgpr_write_b32 a[%c0], %1" :: "i"(e), "v"(qnan) : W8_AGPRS);
            }
            asm volatile("s_nop 7" ::: "memory");
        }
        // (everything below lives inside the epilogue: declared at kernel scope the half-written vectors and the branch-assigned cv[] were
        //  carried around the k-loop -- 16 registers too many for the 128 a wave has beside its accumulators)
        float bias16[16];                                  // bias of the lane's 16 columns of the wave's 64
        float cv[8], xv[8], wv[8];
        V8 hv = {}, av = {};
#pragma unroll
        for (int c = 0; c < 8; ++c) cv[c] = 0.f;
        // the accumulators of half band q (16-column tiles 2q, 2q + 1) of band b -> cv[0..7]: a uniform branch per band
        auto load_cv = [&](const int band, const int q) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (band == b) {
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(cv[4 * e + r]) : "i"(4 * (b * 4 + 2 * q + e) + r));
                }
            }
        };
        auto load_bias16 = [&](unsigned slot) {
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const f32x4 bv = *(const f32x4*)(smem + slot + (unsigned)((wn0 + qq * 32 + lg * 8 + e * 4) * 4));
#pragma unroll
                    for (int r = 0; r < 4; ++r) bias16[qq * 8 + e * 4 + r] = bv[r];
                }
        };
        unsigned e_voff = EOE_OOB;
        int e_soff = 0;
        // micro-operation k of half band q (gemm_ntp.hip's list: phases of 8 independent operations)
        constexpr float C1702 = -1.702f * 1.4426950408889634f;
        auto epi_op = [&](const int q, const int k) {
            if (k < 8) {
                xv[k] = cv[k] * alpha + bias16[8 * q + k];
                asm volatile("" : "+v"(xv[k]));           // round to fp32 HERE, then to 16 bits, as the other kernels' epilogues do: fused into
                return;                                    // v_fma_mixlo_f16 (one rounding from the exact sum) 1 in 2e5 results differs by an ulp
            }
            if (k < 12) { const int j = k - 8; hv[2 * j] = (T)xv[2 * j]; hv[2 * j + 1] = (T)xv[2 * j + 1]; return; }
            if (k == 12) {
                if (EPI == EOE_EPI_GELU) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hv), rpre, (int)e_voff, e_soff + q * 64, W8_ST_AUX);
                else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hv), rc, (int)e_voff, e_soff + q * 64, W8_ST_AUX);
                return;
            }
            if (EPI != EOE_EPI_GELU) return;
            if (k < 21) { const int j = k - 13; xv[j] = (float)hv[j]; return; }       // the ROUNDED pre-activation is what is activated
            if (k < 29) { const int j = k - 21; wv[j] = C1702 * xv[j]; return; }
            if (k < 37) { const int j = k - 29; wv[j] = __builtin_amdgcn_exp2f(wv[j]); return; }
            if (k < 45) { const int j = k - 37; wv[j] = 1.0f + wv[j]; return; }
            if (k < 53) { const int j = k - 45; wv[j] = __builtin_amdgcn_rcpf(wv[j]); return; }
            if (k < 61) { const int j = k - 53; wv[j] = xv[j] * wv[j]; return; }
            if (k < 65) { const int j = k - 61; av[2 * j] = (T)wv[2 * j]; av[2 * j + 1] = (T)wv[2 * j + 1]; return; }
            if (k == 65) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, av), rc, (int)e_voff, e_soff + q * 64, W8_ST_AUX);
        };

        load_bias16((unsigned)W8_BIAS_OFF + (unsigned)(c_tile & 1) * 1024u);
        _Pragma("unroll 1") for (int band = 0; band < NB; ++band) {
            const int mrow = m0 + wm0 + band * 16;
            e_voff = (mrow + lr < p.M) ? lane_off : EOE_OOB;
            e_soff = mrow * ldc2 + (n0 + wn0) * 2;
            load_cv(band, 0);
#pragma unroll
            for (int k = 0; k < L; ++k) epi_op(0, k);
            load_cv(band, 1);
#pragma unroll
            for (int k = 0; k < L; ++k) epi_op(1, k);
        }
    }
#undef W8_ITER
#undef W8_FIRST_FRAGS
#undef W8_CLUSTER
#undef W8_DMA16
}

// The stream-K plan of a launch, or rounds = -1 for the data-parallel form.  OPT-IN (nt_flags bit 20 = 1048576): measured, it loses on every ViT
// shape -- a split costs its 256 KiB fp32 partial tile twice over the fabric, ~0.1 us of the whole chip's memory time per split, 256 splits per
// launch (profiles/r5/streamk_nt.txt).  Full rounds of tiles stay data-parallel; the fractional last round
// (with nt_flags bit 21: the last full round too -- "two-tile" stream-K) is cut along k over all workgroups, XCD by XCD.  Preconditions: the
// caller's workspace, one workgroup per CU on a multiple of 8 CUs, >= 4 k-tiles per tile, and every XCD's share >= 4 k-tiles per workgroup
// (segments are then >= 2 k-tiles long, which the kernel's pipeline needs)
int w8_sk_rounds(const GemmP& p, int tiles, int ncu) {
    const int g_nt_flags = eoe_nt_flags();
    if (!(g_nt_flags & 1048576) || !p.sk_part || !p.sk_sync || (ncu & 7) || ncu > 256 || tiles <= 0) return -1;
    const int nk = p.K / BK;
    if (nk < 4 || tiles % ncu == 0) return -1;
    int R = tiles / ncu;
    if ((g_nt_flags & 2097152) && R >= 1) R -= 1;
    const int tsk = tiles - R * ncu;
    if (tsk < 8 || (long)(tsk / 8) * nk < 4L * (ncu / 8)) return -1;
    return R;
}

template <typename T>
int launch_w8(const GemmP& p0, int epi, hipStream_t s) {
    GemmP p = p0;
    const int tiles = cdiv(p.M, 256) * (p.N / 256);
    const int ncu = num_cus();
    const int R = w8_sk_rounds(p, tiles, ncu);
    const bool sk = R >= 0;
    const int grid = sk ? ncu : (tiles < ncu ? tiles : ncu);
    p.sk_rounds = sk ? R : 0;
    p.sk_flags = (eoe_nt_flags() & 4194304) ? 1 : 0;
#define EOE_W8_LAUNCH(E, SK_)                                                               \
    do {                                                                                    \
        static bool once = (hipFuncSetAttribute((const void*)gemm_w8_kernel<T, E, SK_>, hipFuncAttributeMaxDynamicSharedMemorySize, W8_SMEM_BYTES), true); (void)once; \
        hipLaunchKernelGGL((gemm_w8_kernel<T, E, SK_>), dim3(grid), dim3(512), W8_SMEM_BYTES, s, p); \
    } while (0)
#define EOE_W8_CASE(E)                                                                      \
    case E:                                                                                 \
        if (sk) EOE_W8_LAUNCH(E, true); else EOE_W8_LAUNCH(E, false);                       \
        break;
    switch (epi) {
        EOE_W8_CASE(EOE_EPI_NONE)
        EOE_W8_CASE(EOE_EPI_GELU)
        default: return eoe_set_error(EOE_ERR_ARG, "gemm_w8: epilogue %d is not built", epi);
    }
#undef EOE_W8_CASE
#undef EOE_W8_LAUNCH
    EOE_CHECK_LAUNCH("gemm_w8");
    return 0;
}

}  // namespace

// what the kernel is built for (the launcher in gemm.hip sends everything else to the other kernels)
bool eoe_w8_applies(const void* gemm_p, int epi) {
    const GemmP& p = *(const GemmP*)gemm_p;
    if (epi != EOE_EPI_NONE && epi != EOE_EPI_GELU) return false;
    if (p.out_f32 || p.accumulate || p.colsum || p.colsum_part || p.colsum_sq || p.split_k) return false;
    if ((p.N & 255) || (p.K % BK) || p.K / BK < 2 || p.M < 256) return false;
    if (!epilogue_fast_ok(p)) return false;
    const size_t c_bytes = (((size_t)p.M - 1) * p.ldc + p.N) * 2 + (size_t)256 * p.ldc * 2;           // + one tile of rows: soffset of a ragged last tile
    const size_t a_reach = ((size_t)p.M + 256) * p.lda * 2 + (size_t)p.K * 2;                          // soffset + voffset of a staged row past M
    return c_bytes < 0x7fffffffull && a_reach < 0x7fffffffull;
}

// would the launch run in the stream-K form (the dispatcher's rule in gemm.hip asks)
bool eoe_w8_streamk(const void* gemm_p) {
    const GemmP& p = *(const GemmP*)gemm_p;
    return w8_sk_rounds(p, cdiv(p.M, 256) * (p.N / 256), num_cus()) >= 0;
}

int eoe_launch_w8(const void* gemm_p, int dtype, int epi, hipStream_t s) {
    const GemmP& p = *(const GemmP*)gemm_p;
#ifdef EOE_W8_DEV          // development builds: one instantiation family only (compile time)
    return launch_w8<f16_t>(p, epi, s);
#else
    return dtype == EOE_F16 ? launch_w8<f16_t>(p, epi, s) : launch_w8<bf16_t>(p, epi, s);
#endif
}
