// NT GEMM with the work of a CU split by ROLE (round 4): 128 x 256 x 64 tiles, one persistent 8-wave workgroup per CU = two waves per SIMD --
// a CONSUMER (waves 0-3: 64 x 128 of the tile each; nothing but MFMAs and the LDS reads of their fragments in the k-loop, then the tile's
// epilogue) and a PRODUCER (waves 4-7: every LDS-DMA piece of the 3-stage ring, their address arithmetic, the waits on the memory counter).
//
// Why.  A wave issues ONE instruction at a time, and every instruction holds its issue for >= 4 cycles (an MFMA 16x16x32 for 8 of its 16).
// With one wave per SIMD (gemm256.hip; gemm_ntp.hip, measured this round with in-kernel stamps, tools/ntp_stamps.py) the 40 MFMAs of a
// 32-deep k-step leave 320 issue cycles for everything else, and the 13 fragment reads (~12 cycles each), the 7 LDS-DMA pieces (~46 each,
// with their M0 / offset arithmetic) and the loop's scalar work need ~650: a k-step took 1300 cycles for 640 of matrix work, and epilogue
// arithmetic woven between the MFMAs ADDED its issue time instead of hiding in their shadow.  Instructions of DIFFERENT waves of a SIMD do
// issue side by side (matrix / vector-memory / LDS / scalar ports), which is what the 8-wave and two-workgroup kernels live on -- but
// there both waves of a SIMD run the same stream, are blocked at the same pieces and reach the epilogue together.  Here the consumer's
// stream is 32 MFMAs + 12 LDS reads per k-step (~400 of 512 issue cycles) and is never blocked by a memory instruction.
//   ring: 3 x 48 KB.  One workgroup barrier per k-tile (in the middle of the consumer's iteration): behind barrier j every consumer has
//   read slot j % 3 for the last time and the producers have waited for k-tile j + 1 (counted vmcnt: k-tile j + 2 stays in flight); the
//   producers then refill slot j % 3 with k-tile j + 3.  Every iteration stages exactly 12 pieces per producer wave (dead ones past the
//   last k-tile land in a slot nobody reads), so the wait is a constant.
//   accumulators: literal AGPRs a[0:127] (quad e = (h * 4 + mi) * 4 + ni4 -> a[4e : 4e+3]), reserved by a clobber list on every MFMA
//   statement (gemm_ntp.hip's finding: declared as "+a" operands the allocator copies them around; left unnamed it parks values in them).
//   epilogue: in the open, by the consumers, band by band (16 rows x 64 columns per lane) straight from the accumulators -- B's rows are
//   permuted at staging (eoe_direct_row) so that a lane holds runs of consecutive output columns: 16-byte stores, no LDS round trip.  The
//   tile's bias row comes through LDS (a 13th piece of the producer in the tile's first k-tile, two alternating slots).  Meanwhile the
//   producers have the next tile's first three k-tiles in flight.
// Same products in the same k order as every other NT kernel, same epilogue arithmetic: bitwise the same results.
#include "gemm_common.h"
#include <type_traits>

namespace {

constexpr int PC_MI = 4;                                                               // 16-row MFMA tiles per consumer wave: 128-row tiles
constexpr int PC_A_BYTES = 32 * PC_MI * BK * 2;                                        // 16 KiB
constexpr int PC_STAGE_BYTES = PC_A_BYTES + 256 * BK * 2;                              // 48 KiB
constexpr int PC_NST = 3;
constexpr int PC_BIAS_OFF = PC_NST * PC_STAGE_BYTES;                                   // behind the ring: 2 slots x 4 consumers x 1 KiB (512 B used)
constexpr int PC_SMEM_BYTES = PC_BIAS_OFF + 2 * 4096;
static_assert(PC_SMEM_BYTES <= 160 * 1024, "LDS");

#define PC_A10(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"
#define PC_AGPRS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", PC_A10(1), PC_A10(2), PC_A10(3), PC_A10(4), PC_A10(5), PC_A10(6), \
    PC_A10(7), PC_A10(8), PC_A10(9), PC_A10(10), PC_A10(11), "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
#define PC_MFMA_INPLACE(MNEM, E, A, B) asm volatile(MNEM " a[%c0:%c1], %2, %3, a[%c0:%c1]" :: "i"(4 * (E)), "i"(4 * (E) + 3), "v"(A), "v"(B) : PC_AGPRS)
#define PC_MFMA_FIRST(MNEM, E, A, B) asm volatile(MNEM " a[%c0:%c1], %2, %3, 0" :: "i"(4 * (E)), "i"(4 * (E) + 3), "v"(A), "v"(B) : PC_AGPRS)

template <int N> __device__ __forceinline__ void pc_wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// EPI: EOE_EPI_NONE (16-bit C = alpha acc + bias) or EOE_EPI_GELU (pre -> aux_out, C = QuickGELU of the rounded pre)
template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_pc_kernel(GemmP p) {
    static_assert(EPI == EOE_EPI_NONE || EPI == EOE_EPI_GELU, "epilogues with a second input are not built yet");
    constexpr int MI = PC_MI, BMT = 32 * MI, A_B = PC_A_BYTES, STAGE = PC_STAGE_BYTES, NST = PC_NST, PER = MI + 8;
    constexpr int NMF = MI * 8;                        // MFMAs per cluster (one 32-deep k-step of a consumer's tile) = accumulator quads
    constexpr int NB = 2 * MI;                         // bands of a consumer's tile: b = h * MI + mi (h: 64-column half, mi: 16-row tile)
    constexpr int L = (EPI == EOE_EPI_GELU) ? 66 : 13; // micro-operations per half band (epi_op below)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    asm volatile("" ::: PC_AGPRS);                     // the kernel descriptor allocates a0..a127
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = p.N >> 8;                      // N % 256 == 0 (host)
    const int total_tiles = tiles_n * ((p.M + BMT - 1) / BMT);
    const int G = gridDim.x;
    const int my_tiles = (total_tiles - (int)blockIdx.x + G - 1) / G;
    const int nk = p.K / BK;                           // >= 3 (host)
    const int iters = my_tiles * nk;
    if (iters <= 0) return;
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_void_t*)smem);
    auto tile_origin = [&](int seq, int& m0, int& n0) {
        const int r = xcd_remap((int)blockIdx.x + seq * G, total_tiles);
        m0 = (r / tiles_n) * BMT;
        n0 = (r % tiles_n) * 256;
    };

    if (wave_u >= 4) {
        // ============================================================================================ producer
        const unsigned pw = wave_u - 4;
        __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.bytesA);
        __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.bytesB);
        __amdgpu_buffer_rsrc_t rbias = make_rsrc(p.bias ? (const void*)p.bias : p.C, p.bias ? (unsigned)p.N * 4u : 0u);
        // staging addresses (gemm_ntp.hip): a piece is 8 rows x 128 B; lane -> (row lane >> 3, 16-byte slot lane & 7 holding chunk
        // slot ^ ((row >> 1) & 7)); the swizzle term of piece j depends on j's parity only -> two per-lane offsets per operand + a uniform
        // offset per piece (soffset).  B rows are permuted inside each 64-row group (eoe_direct_row, 16-bit C): row (wave, j, lane) ->
        // 64 wave + fj(j) + gl(lane)
        const int l8 = lane >> 3;
        unsigned voffA[2], voffB[2];
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int c = (lane & 7) ^ ((par * 4 + (lane >> 4)) & 7);
            voffA[par] = (unsigned)((l8 * p.lda + c * 8) * 2);
            voffB[par] = (unsigned)((((lane >> 5) * 8 + (l8 & 3)) * p.ldb + c * 8) * 2);
        }
        int st_tile = 0, st_kt = 0, st_slot = 0;
        unsigned sA_base = 0, sB_base = 0, bias_vo = EOE_OOB;
        int rows_left = 0;
        auto set_offsets = [&](int t) {
            int m0, n0;
            tile_origin(t, m0, n0);
            if (p.dbg & 4) m0 = 0;                     // diagnostics (EOE_GEMM_DEBUG=4): every tile stages the first A panel (L2-resident)
            if (p.dbg & 8) n0 = 0;                     // diagnostics (EOE_GEMM_DEBUG=8): ... and the first B panel
            const int ra0 = m0 + (int)pw * MI * 8;
            sA_base = (unsigned)ra0 * (unsigned)p.lda * 2u;
            rows_left = p.M - ra0;
            sB_base = (unsigned)(n0 + (int)pw * 64) * (unsigned)p.ldb * 2u;
            // the bias of consumer pw's 128 columns (wn0 = (pw & 1) * 128): 32 lanes x 16 B
            bias_vo = lane < 32 ? (unsigned)((n0 + (int)(pw & 1) * 128 + lane * 4) * 4) : EOE_OOB;
        };
#define PC_DMA16(rsrc, lds_addr, voff, soff)                                                                          \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"                           \
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory")
        // the 12 pieces of the staging cursor's k-tile (4 of the A image, 8 of the B image), then the cursor moves on.  Past the workgroup's
        // last k-tile: dead pieces (out-of-range offset: zeros) into the slot that would have been refilled -- nobody reads it any more
        auto stage_ktile = [&](bool live) {
            const unsigned sa = lds0 + (unsigned)st_slot * STAGE;
            const unsigned k0 = (unsigned)st_kt * (BK * 2u);
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                const int pj = ((int)pw * MI + j) & 1;
                const unsigned la = sa + (pw * MI + j) * 1024u;
                const unsigned vo = (live && l8 + j * 8 < rows_left) ? (pj ? voffA[1] : voffA[0]) : EOE_OOB;
                const unsigned so = sA_base + (unsigned)(j * 8) * (unsigned)p.lda * 2u + k0;
                PC_DMA16(ra, la, vo, so);
            }
#pragma unroll
            for (int jb = 0; jb < 8; ++jb) {
                const unsigned lb = sa + A_B + (pw * 8 + jb) * 1024u;
                const unsigned vo = live ? voffB[jb & 1] : EOE_OOB;
                const unsigned so = sB_base + (unsigned)((jb >> 2) * 32 + (jb & 1) * 16 + ((jb >> 1) & 1) * 4) * (unsigned)p.ldb * 2u + k0;
                PC_DMA16(rb, lb, vo, so);
            }
            if (live) {
                st_slot = (st_slot == NST - 1) ? 0 : st_slot + 1;
                if (++st_kt == nk) {
                    st_kt = 0;
                    st_tile += 1;
                    if (st_tile < my_tiles) set_offsets(st_tile);
                }
            }
        };
        // the bias piece of the staging cursor's tile: issued IN FRONT of that tile's first k-tile (older than its 12 pieces: the counted
        // waits below then cover it), into the slot of the tile's parity
        auto stage_bias = [&]() {
            const unsigned la = lds0 + (unsigned)PC_BIAS_OFF + (unsigned)(st_tile & 1) * 4096u + pw * 1024u;
            PC_DMA16(rbias, la, bias_vo, 0);
        };
        set_offsets(0);
#pragma unroll 1
        for (int i = 0; i < NST; ++i) {                // iters >= nk >= 3
            if (st_kt == 0) stage_bias();
            stage_ktile(true);
        }
        pc_wait_vm<2 * PER>();                         // k-tile 0 (and the first bias row) landed; k-tiles 1, 2 may stay in flight
        __builtin_amdgcn_s_barrier();                  // the consumers start
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
            pc_wait_vm<PER>();                         // k-tile it + 1 landed; the 12 youngest (k-tile it + 2, live or dead) may stay in flight
            __builtin_amdgcn_s_barrier();              // ... and every consumer is done with slot it % 3
            const bool live = it + NST < iters;
            if (live && st_kt == 0) stage_bias();
            stage_ktile(live);
        }
        pc_wait_vm<0>();                               // dead pieces target LDS: none may be in flight when the workgroup's LDS is released
        return;
#undef PC_DMA16
    }

    // ================================================================================================ consumer
    const unsigned cw = wave_u;
    const unsigned c_bytes = (unsigned)((((size_t)p.M - 1) * p.ldc + p.N) * 2);
    __amdgpu_buffer_rsrc_t rc = make_rsrc(p.C, c_bytes);
    __amdgpu_buffer_rsrc_t rpre = make_rsrc(p.aux_out ? p.aux_out : p.C, p.aux_out ? c_bytes : 0u);
    const int wm0 = (int)(cw >> 1) * (16 * MI), wn0 = (int)(cw & 1) * 128;
    const int lr = lane & 15, lg = lane >> 4;
    const int sw = (lr >> 1) & 7;
    const int fragA = (wm0 + lr) * 128, fragB = A_B + (wn0 + lr) * 128;
    const int ch0 = ((0 + lg) ^ sw) * 16, ch1 = ((4 + lg) ^ sw) * 16;
    typedef typename T16<T>::v8 V8;

    // ---- L2 prefetch.  The k-loop is bound by the LATENCY of the staged operands: three stages hold at most two k-tiles (96 KB) in flight,
    // and A comes from HBM (2+ us under load: measured 1.14 us per k-tile against 0.75 with loads that fetch nothing).  The consumers never
    // wait on their memory counter inside the k-loop, so they can touch the lines of the k-tile PF_AHEAD ahead of the staging cursor -- one
    // dword per 128-byte line of this wave's quarter of the A image (32 rows) and of the B image (64 rows) -- and the producers' pieces
    // then hit L2.  The loads land in one register that nothing reads (kept allocated to the end of the kernel: a pending load may write
    // it at any time).
#ifndef PC_PF_AHEAD
#define PC_PF_AHEAD 0
#endif
    __amdgpu_buffer_rsrc_t pfa = make_rsrc(p.A, p.bytesA);
    __amdgpu_buffer_rsrc_t pfb = make_rsrc(p.B, p.bytesB);
    const unsigned pf_voffA = lane < 32 ? (unsigned)(lane * p.lda * 2) : EOE_OOB;
    const unsigned pf_voffB = (unsigned)(lane * p.ldb * 2);
    unsigned pf_dummy = 0;
    int pf_tile = 0, pf_kt = 0;
    unsigned pf_sA = 0, pf_sB = 0;
    auto pf_set_tile = [&](int t) {
        int m0, n0;
        tile_origin(t, m0, n0);
        pf_sA = (unsigned)(m0 + (int)cw * 32) * (unsigned)p.lda * 2u;
        pf_sB = (unsigned)(n0 + (int)cw * 64) * (unsigned)p.ldb * 2u;
    };
    auto pf_touch = [&]() {                            // the k-tile under the prefetch cursor, then the cursor moves on
        if (PC_PF_AHEAD <= 0 || pf_tile >= my_tiles) return;
        const unsigned k0 = (unsigned)pf_kt * (BK * 2u);
        asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "+v"(pf_dummy) : "v"(pf_voffA), "s"(pfa), "s"(pf_sA + k0) : "memory");
        asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "+v"(pf_dummy) : "v"(pf_voffB), "s"(pfb), "s"(pf_sB + k0) : "memory");
        if (++pf_kt == nk) {
            pf_kt = 0;
            pf_tile += 1;
            if (pf_tile < my_tiles) pf_set_tile(pf_tile);
        }
    };

#define PC_LANDED(XA, WB)                                                                                        \
    do {                                                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(XA[0]), "+v"(XA[1]), "+v"(XA[2]), "+v"(XA[3]),               \
                     "+v"(WB[0]), "+v"(WB[1]), "+v"(WB[2]), "+v"(WB[3]), "+v"(WB[4]), "+v"(WB[5]), "+v"(WB[6]), "+v"(WB[7]) :: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    } while (0)
    static_assert(MI == 4, "PC_LANDED names four A fragments");

    // ------------------------------------------------------------------------------------------------ the epilogue's pieces
    const float alpha = p.alpha;
    const int ldc2 = p.ldc * 2;
    const unsigned lane_off = (unsigned)(lr * ldc2 + lg * 16);      // this lane's row and 8-column run inside a band's 16 x 32 block (bytes)
    float bias16[16];                                  // bias of the lane's 16 columns of the current 64-column half
    float cv[8], xv[8], wv[8];
    V8 hv, av;
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
    auto load_bias16 = [&](unsigned slot, int h) {
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const f32x4 bv = *(const f32x4*)(smem + slot + (unsigned)((h * 64 + qq * 32 + lg * 8 + e * 4) * 4));
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
            if (EPI == EOE_EPI_GELU) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hv), rpre, (int)e_voff, e_soff + q * 64, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hv), rc, (int)e_voff, e_soff + q * 64, 0);
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
        if (k == 65) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, av), rc, (int)e_voff, e_soff + q * 64, 0);
    };

    // ------------------------------------------------------------------------------------------------ one cluster
    // FIRST: acc = W X (first k-step of a tile), else acc += W X.  Woven between the 32 MFMAs: the 12 fragment reads of the other register
    // set, one after every second MFMA
    constexpr int RSTEP = NMF / (MI + 8);
#define PC_CLUSTER(FIRST, XA, WB, RA, RB, rbase, rks)                                                            \
    do {                                                                                                         \
        const unsigned ra_ = (unsigned)(size_t)((rbase) - smem) + fragA + ((rks) ? ch1 : ch0);                   \
        const unsigned rb_ = (unsigned)(size_t)((rbase) - smem) + fragB + ((rks) ? ch1 : ch0);                   \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                        \
            _Pragma("unroll") for (int ni = 0; ni < 8; ++ni) {                                                   \
                const int idx = mi * 8 + ni;                                                                     \
                const int e_ = ((ni >> 2) * MI + mi) * 4 + (ni & 3);                                             \
                if (std::is_same<T, f16_t>::value) {                                                             \
                    if (FIRST) PC_MFMA_FIRST("v_mfma_f32_16x16x32_f16", e_, WB[ni], XA[mi]);                     \
                    else PC_MFMA_INPLACE("v_mfma_f32_16x16x32_f16", e_, WB[ni], XA[mi]);                         \
                } else {                                                                                         \
                    if (FIRST) PC_MFMA_FIRST("v_mfma_f32_16x16x32_bf16", e_, WB[ni], XA[mi]);                    \
                    else PC_MFMA_INPLACE("v_mfma_f32_16x16x32_bf16", e_, WB[ni], XA[mi]);                        \
                }                                                                                                \
                if (idx % RSTEP == 0 && idx / RSTEP < MI + 8) {                                                  \
                    const int j = idx / RSTEP;                                                                   \
                    if (j < MI) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(RA[j < MI ? j : 0]) : "v"(ra_), "i"((j < MI ? j : 0) * 2048)); \
                    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(RB[j >= MI ? j - MI : 0]) : "v"(rb_), "i"((j >= MI ? j - MI : 0) * 2048)); \
                }                                                                                                \
            }                                                                                                    \
    } while (0)
    // one k-tile iteration: k-step 0 from the registers (+ the reads of k-step 1 from the same slot), the workgroup's barrier (k-tile it + 1
    // has landed; slot `cur` goes back to the producers), k-step 1 (+ the reads of the next k-tile's k-step 0)
#define PC_ITER(FIRST)                                                                                           \
    do {                                                                                                         \
        const int nxt = (cur == NST - 1) ? 0 : cur + 1;                                                          \
        const char* sc = smem + cur * STAGE;                                                                     \
        const char* sn = smem + nxt * STAGE;                                                                     \
        PC_CLUSTER(FIRST, xa0, wb0, xa1, wb1, sc, 1);                                                            \
        PC_LANDED(xa1, wb1);                                                                                     \
        __builtin_amdgcn_s_barrier();                                                                            \
        pf_touch();                                                                                              \
        PC_CLUSTER(false, xa1, wb1, xa0, wb0, sn, 0);                                                            \
        PC_LANDED(xa0, wb0);                                                                                     \
        cur = nxt;                                                                                               \
    } while (0)

    V8 xa0[MI], wb0[8], xa1[MI], wb1[8];
    int cur = 0;
    // the prefetch cursor starts behind the prologue's k-tiles and runs PF_AHEAD k-tiles ahead of the staging cursor
    pf_set_tile(0);
    pf_kt = NST;
    if (pf_kt >= nk) { pf_kt -= nk; pf_tile = 1; if (pf_tile < my_tiles) pf_set_tile(1); }     // (nk >= 3)
#pragma unroll 1
    for (int i = 0; i < PC_PF_AHEAD; ++i) pf_touch();
    __builtin_amdgcn_s_barrier();                      // k-tile 0 has landed
    {
        const unsigned a_ = (unsigned)fragA + ch0, b_ = (unsigned)fragB + ch0;
#pragma unroll
        for (int i = 0; i < MI; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xa0[i]) : "v"(a_), "i"(i * 2048));
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wb0[i]) : "v"(b_), "i"(i * 2048));
    }
    PC_LANDED(xa0, wb0);
    for (int c_tile = 0; c_tile < my_tiles; ++c_tile) {
        PC_ITER(true);
        _Pragma("unroll 1") for (int kt = 1; kt < nk; ++kt) PC_ITER(false);
        // ---- epilogue (the next tile's first k-step is already in xa0 / wb0; the producers have its first k-tiles in flight)
        int m0, n0;
        tile_origin(c_tile, m0, n0);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");        // the last MFMAs' results have landed
        __builtin_amdgcn_sched_barrier(0);
        const unsigned bias_slot = (unsigned)PC_BIAS_OFF + (unsigned)(c_tile & 1) * 4096u + cw * 1024u;
        _Pragma("unroll 1") for (int band = 0; band < NB; ++band) {
            const int h = band >= MI ? 1 : 0, mi = band - h * MI;
            if (mi == 0) load_bias16(bias_slot, h);
            const int mrow = m0 + wm0 + mi * 16;
            e_voff = (mrow + lr < p.M) ? lane_off : EOE_OOB;
            e_soff = mrow * ldc2 + (n0 + wn0 + h * 64) * 2;
            load_cv(band, 0);
#pragma unroll
            for (int k = 0; k < L; ++k) epi_op(0, k);
            load_cv(band, 1);
#pragma unroll
            for (int k = 0; k < L; ++k) epi_op(1, k);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf_dummy) :: "memory");       // no load may still be on its way to a register of a finished wave
#undef PC_ITER
#undef PC_CLUSTER
#undef PC_LANDED
}

template <typename T>
int launch_pc(const GemmP& p, int epi, hipStream_t s) {
    const int tiles = cdiv(p.M, 32 * PC_MI) * (p.N / 256);
    const int ncu = num_cus();
    const int grid = tiles < ncu ? tiles : ncu;
#define EOE_PC_CASE(E)                                                                      \
    case E:                                                                                 \
        { static bool once = (hipFuncSetAttribute((const void*)gemm_pc_kernel<T, E>, hipFuncAttributeMaxDynamicSharedMemorySize, PC_SMEM_BYTES), true); (void)once; } \
        hipLaunchKernelGGL((gemm_pc_kernel<T, E>), dim3(grid), dim3(512), PC_SMEM_BYTES, s, p); \
        break;
    switch (epi) {
        EOE_PC_CASE(EOE_EPI_NONE)
        EOE_PC_CASE(EOE_EPI_GELU)
        default: return eoe_set_error(EOE_ERR_ARG, "gemm_pc: epilogue %d is not built", epi);
    }
#undef EOE_PC_CASE
    EOE_CHECK_LAUNCH("gemm_pc");
    return 0;
}

}  // namespace

// what the kernel is built for (the launcher in gemm.hip sends everything else to the other kernels)
bool eoe_pc_applies(const void* gemm_p, int epi) {
    const GemmP& p = *(const GemmP*)gemm_p;
    if (epi != EOE_EPI_NONE && epi != EOE_EPI_GELU) return false;
    if (p.out_f32 || p.accumulate || p.colsum || p.colsum_part || p.colsum_sq || p.split_k) return false;
    if ((p.N & 255) || (p.K % BK) || p.K / BK < 3 || p.M < 32 * PC_MI) return false;
    if (!epilogue_fast_ok(p)) return false;
    const size_t c_bytes = (((size_t)p.M - 1) * p.ldc + p.N) * 2 + (size_t)32 * PC_MI * p.ldc * 2;   // + one tile of rows: soffset of a ragged last tile
    const size_t a_reach = ((size_t)p.M + 32 * PC_MI) * p.lda * 2 + (size_t)p.K * 2;                    // soffset + voffset of a staged row past M
    return c_bytes < 0x7fffffffull && a_reach < 0x7fffffffull;
}

int eoe_launch_pc(const void* gemm_p, int dtype, int epi, hipStream_t s) {
    const GemmP& p = *(const GemmP*)gemm_p;
#ifdef EOE_PC_DEV          // development builds: one instantiation family only (compile time)
    return launch_pc<f16_t>(p, epi, s);
#else
    return dtype == EOE_F16 ? launch_pc<f16_t>(p, epi, s) : launch_pc<bf16_t>(p, epi, s);
#endif
}
