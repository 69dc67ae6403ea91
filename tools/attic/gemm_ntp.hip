// NT GEMM with the epilogue of tile t drained UNDER the MFMAs of tile t + 1 (round 4).  One persistent workgroup per CU, one wave per SIMD
// (gemm256.hip's geometry: 128 x 256 x 64 tiles, four waves of 64 x 128, 3-stage LDS ring filled by LDS-DMA, asm k-loop).
//
// Why: on the K = 768 shapes with wide outputs (c_fc forward: two 16-bit outputs, GELU' x dY, the
// in-projection) a tile's epilogue -- convert, activation, 40-80 stores per wave -- costs a third to all of its 12-iteration main loop, and
// with one workgroup per CU nothing runs beside it; two workgroups per CU run in lockstep and lose the same time.  Here the last 32-deep
// k-step of a tile writes its MFMA results into a second register set (`hold`: D != C), the next tile starts with C = 0 into `acc`, and
// the held tile is converted and stored band by band (16 rows x 64 columns per lane: one band per k-tile iteration of the next tile) in
// the issue slots the matrix pipe leaves free: an MFMA 16x16x32 occupies the SIMD's issue for 8 of its 16 cycles, the band's VALU
// instructions are placed two per MFMA (sched_barrier between the steps), its stores between them.  Nothing of the epilogue is on the
// critical path except the last tile of a workgroup (drained after the loop).
//   registers: the accumulators are LITERAL registers -- quad e of the running tile is a[4e : 4e+3], quad e of the held tile a[128 + 4e : ...]:
//   all 256 AGPRs.  Through "+a" operands the register allocator split the two 128-register values around the loops that do not touch
//   them, kept parts of `hold` in VGPRs and scratch and copied them into place in front of every asm statement declared to read them.
//   Named registers are invisible to it; what makes this safe: ntp_reserve_agprs() clobbers a0..a255 once, so the kernel descriptor
//   allocates them; the file is compiled with -mllvm -amdgpu-spill-vgpr-to-agpr=0; and the build is audited -- no AGPR outside the asm
//   statements of this file (tests/test_cpu_host.py::test_ntp_kernel_keeps_out_of_the_accumulators).
//   memory counter: every k-tile iteration issues exactly PER LDS-DMA pieces (live or dead) and, in the iterations that carry the drain, ES
//   stores per cluster at fixed places in the instruction stream, so the counted waits are compile-time constants (vmcnt counts loads,
//   stores and LDS-DMA together, in issue order).
//   bias: the tile's 256 bias values reach LDS by one LDS-DMA piece per wave (its own 128 columns) and are read into 16 registers per
//   64-column half when the drain reaches that half.
// Same products in the same k order as every other NT kernel, same epilogue arithmetic: bitwise the same results.
#include "gemm_common.h"
#include <type_traits>

namespace {

// MI = 16-row MFMA tiles per wave along M: (32 MI) x 256 tiles.  MI = 4: acc + hold = all 256 AGPRs.  MI = 5 (160 x 256: 98 flop per byte staged
// against 85 -- a CU takes in ~70 GB/s from L2, i.e. ~71 flop per byte are needed to feed its matrix pipes): 32 of the 40 accumulator quads
// in AGPRs as for MI = 4, the last 8 (acc and hold: C and D of an MFMA share one register file) in VGPRs as ordinary tied operands
constexpr int ntp_a_bytes(int MI) { return 32 * MI * BK * 2; }                         // 16 / 20 KiB
constexpr int ntp_stage_bytes(int MI) { return ntp_a_bytes(MI) + 256 * BK * 2; }       // 48 / 52 KiB
constexpr int NTP_NST = 3;
constexpr int ntp_bias_off(int MI) { return NTP_NST * ntp_stage_bytes(MI); }           // behind the ring: one KiB per wave (512 B used)
constexpr int ntp_smem_bytes(int MI) { return ntp_bias_off(MI) + 4096; }
static_assert(ntp_smem_bytes(5) <= 160 * 1024, "LDS");
constexpr int NTP_QA = 32;                                                             // accumulator quads (acc + hold) that live in AGPRs

// The three MFMA forms on accumulator quad E (and held quad H): in place; first k-step of a tile (C = 0); last k-step (C = the acc quad,
// D = the hold quad: the finished tile moves aside).  E, H: constants after unrolling ("i" operands spliced into the register text)
#define EOE_NTP_MFMA_INPLACE(MNEM, E, A, B) asm volatile(MNEM " a[%c0:%c1], %2, %3, a[%c0:%c1]" :: "i"(4 * (E)), "i"(4 * (E) + 3), "v"(A), "v"(B) : NTP_AGPRS)
#define EOE_NTP_MFMA_FIRST(MNEM, E, A, B) asm volatile(MNEM " a[%c0:%c1], %2, %3, 0" :: "i"(4 * (E)), "i"(4 * (E) + 3), "v"(A), "v"(B) : NTP_AGPRS)
#define EOE_NTP_MFMA_LAST(MNEM, E, H, A, B) \
    asm volatile(MNEM " a[%c0:%c1], %4, %5, a[%c2:%c3]" :: "i"(4 * (H)), "i"(4 * (H) + 3), "i"(4 * (E)), "i"(4 * (E) + 3), "v"(A), "v"(B) : NTP_AGPRS)
// the same on a VGPR-resident quad (tied operands: one value updated in place; `first` does not read it, `last` writes the held quad)
#define EOE_NTP_MFMA_INPLACE_V(MNEM, C, A, B) asm volatile(MNEM " %0, %1, %2, %0" : "+v"(C) : "v"(A), "v"(B))
#define EOE_NTP_MFMA_FIRST_V(MNEM, C, A, B) asm volatile(MNEM " %0, %1, %2, 0" : "+v"(C) : "v"(A), "v"(B))
#define EOE_NTP_MFMA_LAST_V(MNEM, C, H, A, B) asm volatile(MNEM " %0, %2, %3, %1" : "+v"(H) : "v"(C), "v"(A), "v"(B))

// every accumulator register, as a clobber list: on ntp_reserve_agprs() (the kernel descriptor then allocates all 256) and on EVERY MFMA
// statement -- under VGPR pressure the compiler parks values in AGPRs it believes free (seen: a0..a3, with -amdgpu-spill-vgpr-to-agpr=0),
// and a register it must assume overwritten by each of the 64-80 MFMAs of an iteration is of no use to it
#define NTP_A10(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"
#define NTP_AGPRS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", NTP_A10(1), NTP_A10(2), NTP_A10(3), NTP_A10(4), NTP_A10(5), NTP_A10(6), \
    NTP_A10(7), NTP_A10(8), NTP_A10(9), NTP_A10(10), NTP_A10(11), NTP_A10(12), NTP_A10(13), NTP_A10(14), NTP_A10(15), NTP_A10(16), NTP_A10(17),        \
    NTP_A10(18), NTP_A10(19), NTP_A10(20), NTP_A10(21), NTP_A10(22), NTP_A10(23), NTP_A10(24), "a250", "a251", "a252", "a253", "a254", "a255"
__device__ __forceinline__ void ntp_reserve_agprs() { asm volatile("" ::: NTP_AGPRS); }

template <int N> __device__ __forceinline__ void ntp_wait_vm() {
    static_assert(N >= 0, "vmcnt");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N > 63 ? 63 : N) : "memory");
}

// EPI: EOE_EPI_NONE (16-bit C = alpha acc + bias) or EOE_EPI_GELU (pre -> aux_out, C = QuickGELU of the rounded pre)
template <typename T, int EPI, int MI>
__global__ __launch_bounds__(256, 1) void gemm_ntp_kernel(GemmP p) {
    static_assert(EPI == EOE_EPI_NONE || EPI == EOE_EPI_GELU, "epilogues with a second input are not built yet");
    static_assert(MI == 4 || MI == 5, "MI");
    constexpr int BMT = 32 * MI, A_B = ntp_a_bytes(MI), STAGE = ntp_stage_bytes(MI), NST = NTP_NST, PER = MI + 8;
    constexpr int P1 = (PER + 1) / 2;                  // pieces of a k-tile issued in a second half; the rest in the next first half
    constexpr int NMF = MI * 8;                        // MFMAs per cluster (one 32-deep k-step of the wave's tile) = accumulator quads
    constexpr int NB = 2 * MI;                         // bands of a wave's tile: b = h * MI + mi (h: 64-column half, mi: 16-row tile)
    constexpr int ES = (EPI == EOE_EPI_GELU) ? 2 : 1;  // stores per half band = per cluster in the iterations that carry the drain
    constexpr int L = (EPI == EOE_EPI_GELU) ? 66 : 13; // micro-operations per half band (epi_op below)
    constexpr int QA = NTP_QA, QV = NMF - QA;         // quads in AGPRs: acc a[4e : 4e+3], hold a[128 + 4e : ...]; quads e >= QA in VGPRs
    static_assert(QA == 32 && QV >= 0 && QV <= 8, "quads");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ntp_reserve_agprs();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = p.N >> 8;                      // N % 256 == 0 (host)
    const int total_tiles = tiles_n * ((p.M + BMT - 1) / BMT);
    const int G = gridDim.x;
    const int my_tiles = (total_tiles - (int)blockIdx.x + G - 1) / G;
    const int nk = p.K / BK;                           // >= NB + 2 (host)
    const int iters = my_tiles * nk;
    if (iters <= 0) return;

    __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.bytesA);
    __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.bytesB);
    const unsigned c_bytes = (unsigned)((((size_t)p.M - 1) * p.ldc + p.N) * 2);
    __amdgpu_buffer_rsrc_t rc = make_rsrc(p.C, c_bytes);
    __amdgpu_buffer_rsrc_t rpre = make_rsrc(p.aux_out ? p.aux_out : p.C, p.aux_out ? c_bytes : 0u);
    __amdgpu_buffer_rsrc_t rbias = make_rsrc(p.bias ? (const void*)p.bias : p.C, p.bias ? (unsigned)p.N * 4u : 0u);

    const unsigned lds0 = (unsigned)(uintptr_t)((lds_void_t*)smem);
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(wave);
    // staging addresses: a piece is 8 rows x 128 B; lane -> (row lane >> 3, 16-byte slot lane & 7 holding chunk slot ^ ((row >> 1) & 7)).  The
    // swizzle term of piece j depends on j's parity only, so two per-lane offsets per operand (VGPRs) + a uniform offset per piece (the
    // instruction's soffset) address everything: 5 registers instead of the MI + 8 per-piece offsets.  B rows are permuted inside each
    // 64-row group (eoe_direct_row, 16-bit C) so that a lane's accumulator values are runs of 8 consecutive output columns and the four
    // lanes of a row write adjacent 16-byte pieces: row (wave, j, lane) -> 64 wave + fj(j) + gl(lane)
    const int l8 = lane >> 3;
    unsigned voffA[2], voffB[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int c = (lane & 7) ^ ((par * 4 + (lane >> 4)) & 7);
        voffA[par] = (unsigned)((l8 * p.lda + c * 8) * 2);
        voffB[par] = (unsigned)((((lane >> 5) * 8 + (l8 & 3)) * p.ldb + c * 8) * 2);
    }
    int st_tile = 0, st_kt = 0, st_slot = 0;
    unsigned sA_base = 0, sB_base = 0;                 // byte offsets of the staging tile's first row of this wave (uniform)
    int rows_left = 0;                                 // rows of A below M from this wave's first staged row on
    auto tile_origin = [&](int seq, int& m0, int& n0) {
        const int r = xcd_remap((int)blockIdx.x + seq * G, total_tiles);
        m0 = (r / tiles_n) * BMT;
        n0 = (r % tiles_n) * 256;
    };
    auto set_offsets = [&](int t) {
        int m0, n0;
        tile_origin(t, m0, n0);
        const int ra0 = m0 + (int)wave_u * MI * 8;
        sA_base = (unsigned)ra0 * (unsigned)p.lda * 2u;
        rows_left = p.M - ra0;
        sB_base = (unsigned)(n0 + (int)wave_u * 64) * (unsigned)p.ldb * 2u;
    };
#define EOE_DMA16(rsrc, lds_addr, voff, soff)                                                                         \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"                           \
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory")
    // piece j of the staging cursor's k-tile (j < MI: the A image, else the B image).  A piece with nothing to fetch (`live` false: past
    // the workgroup's last k-tile) keeps its place in the instruction stream and in the vmcnt count; it lands (zeros) in the slot of the
    // third-last k-tile, which nothing reads any more.  Rows of A at or past M: out-of-range offset (zero fill)
    auto stage_piece = [&](int j, bool live) {
        const unsigned sa = lds0 + (unsigned)st_slot * STAGE;
        const unsigned k0 = (unsigned)st_kt * (BK * 2u);
        if (j < MI) {
            const int pj = ((int)wave_u * MI + j) & 1;
            const unsigned la = sa + (wave_u * MI + j) * 1024u;
            const unsigned vo = (live && l8 + j * 8 < rows_left) ? (pj ? voffA[1] : voffA[0]) : EOE_OOB;
            const unsigned so = sA_base + (unsigned)(j * 8) * (unsigned)p.lda * 2u + k0;
            EOE_DMA16(ra, la, vo, so);
        } else {
            const int jb = j - MI;
            const unsigned lb = sa + A_B + (wave_u * 8 + jb) * 1024u;
            const unsigned vo = live ? voffB[jb & 1] : EOE_OOB;
            const unsigned so = sB_base + (unsigned)((jb >> 2) * 32 + (jb & 1) * 16 + ((jb >> 1) & 1) * 4) * (unsigned)p.ldb * 2u + k0;
            EOE_DMA16(rb, lb, vo, so);
        }
    };
    auto stage_advance = [&]() {
        st_slot = (st_slot == NST - 1) ? 0 : st_slot + 1;
        if (++st_kt == nk) {
            st_kt = 0;
            st_tile += 1;
            if (st_tile < my_tiles) set_offsets(st_tile);
        }
    };

    const int wm0 = (int)(wave_u >> 1) * (16 * MI), wn0 = (int)(wave_u & 1) * 128;
    const int lr = lane & 15, lg = lane >> 4;
    const int sw = (lr >> 1) & 7;
    const int fragA = (wm0 + lr) * 128, fragB = A_B + (wn0 + lr) * 128;
    const int ch0 = ((0 + lg) ^ sw) * 16, ch1 = ((4 + lg) ^ sw) * 16;
    typedef typename T16<T>::v8 V8;

#define EOE_NTP_LANDED(XA, WB)                                                                                   \
    do {                                                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(XA[0]), "+v"(XA[1]), "+v"(XA[2]), "+v"(XA[3]), "+v"(XA[MI - 1]),        \
                     "+v"(WB[0]), "+v"(WB[1]), "+v"(WB[2]), "+v"(WB[3]), "+v"(WB[4]), "+v"(WB[5]), "+v"(WB[6]), "+v"(WB[7]) :: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    } while (0)

    // ------------------------------------------------------------------------------------------------ the drain of the held tile
    // accumulator quad e = (h * MI + mi) * 4 + ni4 (h: 64-column half, mi: 16-row tile, ni4: 16-column tile of the half): a[4e : 4e+3];
    // the same quad of the held tile: a[4 (NMF + e) : ...]
    const float alpha = p.alpha;
    const int ldc2 = p.ldc * 2;
    const unsigned lane_off = (unsigned)(lr * ldc2 + lg * 16);      // this lane's row and 8-column run inside a band's 16 x 32 block (bytes)
    int pm0 = 0, pn0 = 0;                              // origin of the held tile
    int band = 0;                                      // band being drained (uniform)
    bool have_held = false;                            // a finished tile sits in `hold`
    float bias16[16];                                  // bias of the lane's 16 columns of the current 64-column half
    float cv[8], xv[8], wv[8];
    V8 hv, av;
#pragma unroll
    for (int c = 0; c < 16; ++c) bias16[c] = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) cv[c] = 0.f;
    const unsigned bias_lds = (unsigned)ntp_bias_off(MI) + wave_u * 1024u;
    f32x4 vacc[QV > 0 ? QV : 1], vhold[QV > 0 ? QV : 1];   // the VGPR-resident quads (MI = 5)
#pragma unroll
    for (int i = 0; i < (QV > 0 ? QV : 1); ++i) { vacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; vhold[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    // the held values of half band q (16-column tiles 2q, 2q + 1) of band `band` -> cv[0..7]: a uniform branch per band
    auto load_cv = [&](const int q) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (band == b) {
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int hq = b * 4 + 2 * q + e;
                        if (hq < QA) asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(cv[4 * e + r]) : "i"(4 * (QA + hq) + r));
                        else asm volatile("v_mov_b32 %0, %1" : "=v"(cv[4 * e + r]) : "v"(vhold[hq >= QA ? hq - QA : 0][r]));
                    }
            }
        }
    };
    // bias of the half the drain enters (band 0 and band MI): 16 values = the two 8-column runs of the lane
    auto load_bias16 = [&](int h) {
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const f32x4 bv = *(const f32x4*)(smem + bias_lds + (unsigned)((h * 64 + qq * 32 + lg * 8 + e * 4) * 4));
#pragma unroll
                for (int r = 0; r < 4; ++r) bias16[qq * 8 + e * 4 + r] = bv[r];
            }
    };
    unsigned e_voff = EOE_OOB;
    int e_soff = 0;
    auto band_setup = [&](bool on) {                   // store addresses of band `band` of the held tile (nothing held: out of range)
        const int h = band >= MI ? 1 : 0, mi = band - h * MI;
        const int mrow = pm0 + wm0 + mi * 16;
        e_voff = (on && mrow + lr < p.M) ? lane_off : EOE_OOB;
        e_soff = mrow * ldc2 + (pn0 + wn0 + h * 64) * 2;
    };
    // micro-operation k of half band q (constants after unrolling).  Phases of 8 independent operations: a dependent operation is 8 apart.
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
    // MODE 0: acc += W X;  1: acc = W X (first k-step of a tile);  2: hold = acc + W X (last k-step of a tile).
    // Woven between the NMF MFMAs: the MI + 8 fragment reads of the other register set (one after every RSTEP-th MFMA), the LDS-DMA pieces
    // [PLO, PHI) of the staging cursor's k-tile, and (ACT) the micro-operations of half band Q of the held tile -- all at fixed places.
    constexpr int RSTEP = NMF / (MI + 8);
#define EOE_NTP_MFMA(MNEM, MODE, E, A, B)                                                                        \
    do {                                                                                                         \
        if ((E) < QA) {                                                                                          \
            if ((MODE) == 0) EOE_NTP_MFMA_INPLACE(MNEM, E, A, B);                                                \
            else if ((MODE) == 1) EOE_NTP_MFMA_FIRST(MNEM, E, A, B);                                             \
            else EOE_NTP_MFMA_LAST(MNEM, E, QA + (E), A, B);                                                     \
        } else {                                                                                                 \
            if ((MODE) == 0) EOE_NTP_MFMA_INPLACE_V(MNEM, vacc[(E) >= QA ? (E) - QA : 0], A, B);                 \
            else if ((MODE) == 1) EOE_NTP_MFMA_FIRST_V(MNEM, vacc[(E) >= QA ? (E) - QA : 0], A, B);              \
            else EOE_NTP_MFMA_LAST_V(MNEM, vacc[(E) >= QA ? (E) - QA : 0], vhold[(E) >= QA ? (E) - QA : 0], A, B); \
        }                                                                                                        \
    } while (0)
#define EOE_NTP_CLUSTER(MODE, ACT, Q, XA, WB, RA, RB, rbase, rks, PLO, PHI, LIVE)                                \
    do {                                                                                                         \
        const unsigned ra_ = (unsigned)(size_t)((rbase) - smem) + fragA + ((rks) ? ch1 : ch0);                   \
        const unsigned rb_ = (unsigned)(size_t)((rbase) - smem) + fragB + ((rks) ? ch1 : ch0);                   \
        if ((ACT) && have_held) load_cv(Q);                                                                      \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                        \
            _Pragma("unroll") for (int ni = 0; ni < 8; ++ni) {                                                   \
                const int idx = mi * 8 + ni;                                                                     \
                const int e_ = ((ni >> 2) * MI + mi) * 4 + (ni & 3);                                             \
                if (std::is_same<T, f16_t>::value) EOE_NTP_MFMA("v_mfma_f32_16x16x32_f16", MODE, e_, WB[ni], XA[mi]); \
                else EOE_NTP_MFMA("v_mfma_f32_16x16x32_bf16", MODE, e_, WB[ni], XA[mi]);                         \
                if (idx % RSTEP == 0 && idx / RSTEP < MI + 8) {                                                  \
                    const int j = idx / RSTEP;                                                                   \
                    if (j < MI) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(RA[j < MI ? j : 0]) : "v"(ra_), "i"((j < MI ? j : 0) * 2048)); \
                    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(RB[j >= MI ? j - MI : 0]) : "v"(rb_), "i"((j >= MI ? j - MI : 0) * 2048)); \
                }                                                                                                \
                _Pragma("unroll") for (int sl = 0; sl < (PHI) - (PLO); ++sl)                                     \
                    if (idx == ((sl + 1) * NMF) / ((PHI) - (PLO)) - 1) stage_piece((PLO) + sl, (LIVE));          \
                if (ACT) {                                                                                       \
                    _Pragma("unroll") for (int k = 0; k < L; ++k)                                                \
                        if ((k * NMF) / L == idx) epi_op(Q, k);                                                  \
                }                                                                                                \
                __builtin_amdgcn_sched_barrier(0);                                                               \
            }                                                                                                    \
    } while (0)

    V8 xa0[MI], wb0[8], xa1[MI], wb1[8];
    int cur = 0, it = 0;
    // diagnostics (EOE_GEMM_STAMP=1): wave 0's cycles per workgroup -- [0] prologue, [1] sum of first halves, [2] DMA wait + fragment wait,
    // [3] at the barrier, [4] second halves, [5] drain in the open, [6] whole kernel, [7] tiles
    unsigned long long* stp = (p.stamp && wave == 0) ? p.stamp + (size_t)blockIdx.x * 16 : nullptr;
    unsigned long long t_0 = 0, t_a = 0, t_b = 0, t_c = 0, t_d = 0, c_a = 0, c_w = 0, c_bar = 0, c_b = 0;
    if (stp) t_0 = __builtin_amdgcn_s_memtime();
    bool part = false;                                 // the staging cursor's k-tile has its first P1 pieces issued, the rest are due
    // one k-tile iteration.  FIRST / LAST: of its tile;  ACT: carries the drain's micro-operations and ES stores per cluster (band = kt < NB;
    // in a workgroup's first tile they run on nothing and store out of range: no second set of bodies);  PREV: the previous iteration was
    // an ACT one;  wx (run time, uniform): the iteration behind the last ACT one, which also issues the tile's bias piece
    //   wait in the middle: k-tile it + 1 landed.  Younger than its last piece (issued in the previous iteration's first half): the P1
    //   pieces + ES stores of the previous second half, the PER - P1 pieces + ES stores of this first half
#define EOE_NTP_ITER(FIRST, LAST, ACT, PREV, wx)                                                                 \
    do {                                                                                                         \
        const int nxt = (cur == NST - 1) ? 0 : cur + 1;                                                          \
        const char* sc = smem + cur * STAGE;                                                                     \
        const char* sn = smem + nxt * STAGE;                                                                     \
        if (stp) t_a = __builtin_amdgcn_s_memtime();                                                             \
        EOE_NTP_CLUSTER((FIRST) ? 1 : 0, ACT, 0, xa0, wb0, xa1, wb1, sc, 1, P1, PER, part);                      \
        if (part) { stage_advance(); part = false; }                                                             \
        if (stp) t_b = __builtin_amdgcn_s_memtime();                                                             \
        if (wx) ntp_wait_vm<PER + ES + 1>(); else ntp_wait_vm<PER + ((PREV) ? ES : 0) + ((ACT) ? ES : 0)>();     \
        EOE_NTP_LANDED(xa1, wb1);                                                                                \
        if (stp) t_c = __builtin_amdgcn_s_memtime();                                                             \
        __builtin_amdgcn_s_barrier();                                                                            \
        if (stp) t_d = __builtin_amdgcn_s_memtime();                                                             \
        if (wave_u & 1) asm volatile("s_nop 15" ::: "memory");                                                   \
        if (wave_u & 2) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                                       \
        const bool dma_ = it + NST < iters;                                                                      \
        EOE_NTP_CLUSTER((LAST) ? 2 : 0, ACT, 1, xa1, wb1, xa0, wb0, sn, 0, 0, P1, dma_);                         \
        part = dma_;                                                                                             \
        EOE_NTP_LANDED(xa0, wb0);                                                                                \
        if (stp) { c_a += t_b - t_a; c_w += t_c - t_b; c_bar += t_d - t_c; c_b += __builtin_amdgcn_s_memtime() - t_d; } \
        cur = nxt;                                                                                               \
        ++it;                                                                                                    \
    } while (0)

    // ------------------------------------------------------------------------------------------------ prologue
    // k-tiles 0 and 1 whole, the first P1 pieces of k-tile 2: the state every iteration starts from (iters >= 3)
    set_offsets(0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < PER; ++j) stage_piece(j, true);
        stage_advance();
    }
#pragma unroll
    for (int j = 0; j < P1; ++j) stage_piece(j, true);
    part = true;
    ntp_wait_vm<PER + P1>();                           // k-tile 0 landed
    __builtin_amdgcn_s_barrier();
    {
        const unsigned a_ = (unsigned)fragA + ch0, b_ = (unsigned)fragB + ch0;
#pragma unroll
        for (int i = 0; i < MI; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xa0[i]) : "v"(a_), "i"(i * 2048));
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wb0[i]) : "v"(b_), "i"(i * 2048));
    }
    EOE_NTP_LANDED(xa0, wb0);
    if (stp && lane == 0) stp[0] = __builtin_amdgcn_s_memtime() - t_0;

    // ------------------------------------------------------------------------------------------------ tiles
    for (int c_tile = 0; c_tile < my_tiles; ++c_tile) {
        int m0, n0;
        tile_origin(c_tile, m0, n0);
        // k-tiles 0 .. NB - 1: one band of the held tile each
        band = 0;
        if (have_held) load_bias16(0);
        band_setup(have_held);
        EOE_NTP_ITER(true, false, true, false, false);
        _Pragma("unroll 1") for (int kt = 1; kt < NB; ++kt) {
            band = kt;
            if (have_held && kt == MI) load_bias16(1);
            band_setup(have_held);
            EOE_NTP_ITER(false, false, true, true, false);
        }
        // k-tile NB: this tile's bias row goes to LDS (one piece per wave: its 128 columns = 32 lanes x 16 B; behind the drain's last
        // reads of the previous row, in this wave's program order)
        _Pragma("unroll 1") for (int kt = NB; kt < nk - 1; ++kt) {
            if (kt == NB) {
                const unsigned vo = lane < 32 ? (unsigned)((n0 + wn0 + lane * 4) * 4) : EOE_OOB;
                const unsigned la = lds0 + bias_lds;
                EOE_DMA16(rbias, la, vo, 0);
            }
            EOE_NTP_ITER(false, false, false, false, kt == NB);
        }
        // last k-tile: its second k-step leaves the tile in `hold`
        EOE_NTP_ITER(false, true, false, false, false);
        pm0 = m0;
        pn0 = n0;
        have_held = true;
    }
    // ------------------------------------------------------------------------------------------------ the last tile: drained in the open
    if (stp) t_a = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");            // the last MFMAs' results have landed in `hold`
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // incl. this wave's bias piece
    __builtin_amdgcn_sched_barrier(0);
    _Pragma("unroll 1") for (band = 0; band < NB; ++band) {
        if (band == 0) load_bias16(0);
        if (band == MI) load_bias16(1);
        band_setup(true);
        load_cv(0);
#pragma unroll
        for (int k = 0; k < L; ++k) epi_op(0, k);
        load_cv(1);
#pragma unroll
        for (int k = 0; k < L; ++k) epi_op(1, k);
    }
    if (stp && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_e = __builtin_amdgcn_s_memtime();
        stp[1] = c_a; stp[2] = c_w; stp[3] = c_bar; stp[4] = c_b; stp[5] = t_e - t_a; stp[6] = t_e - t_0; stp[7] = (unsigned long long)my_tiles;
    }
#undef EOE_NTP_ITER
#undef EOE_NTP_CLUSTER
#undef EOE_NTP_MFMA
#undef EOE_NTP_LANDED
#undef EOE_DMA16
}

template <typename T, int MI>
int launch_ntp(const GemmP& p, int epi, hipStream_t s) {
    const int tiles = cdiv(p.M, 32 * MI) * (p.N / 256);
    const int ncu = num_cus();
    const int grid = tiles < ncu ? tiles : ncu;
#define EOE_NTP_CASE(E)                                                                     \
    case E:                                                                                 \
        { static bool once = (hipFuncSetAttribute((const void*)gemm_ntp_kernel<T, E, MI>, hipFuncAttributeMaxDynamicSharedMemorySize, ntp_smem_bytes(MI)), true); (void)once; } \
        hipLaunchKernelGGL((gemm_ntp_kernel<T, E, MI>), dim3(grid), dim3(256), ntp_smem_bytes(MI), s, p); \
        break;
    switch (epi) {
        EOE_NTP_CASE(EOE_EPI_NONE)
        EOE_NTP_CASE(EOE_EPI_GELU)
        default: return eoe_set_error(EOE_ERR_ARG, "gemm_ntp: epilogue %d is not built", epi);
    }
#undef EOE_NTP_CASE
    EOE_CHECK_LAUNCH("gemm_ntp");
    return 0;
}

}  // namespace

// what the kernel is built for (the launcher in gemm.hip sends everything else to the other kernels)
bool eoe_ntp_applies(const void* gemm_p, int epi, int mi) {
    const GemmP& p = *(const GemmP*)gemm_p;
    if (epi != EOE_EPI_NONE && epi != EOE_EPI_GELU) return false;
    if (p.out_f32 || p.accumulate || p.colsum || p.colsum_part || p.colsum_sq || p.split_k) return false;
    if ((p.N & 255) || (p.K % BK) || p.K / BK < 2 * mi + 2 || p.M < 32 * mi) return false;
    if (!epilogue_fast_ok(p)) return false;
    const size_t c_bytes = (((size_t)p.M - 1) * p.ldc + p.N) * 2 + (size_t)32 * mi * p.ldc * 2;   // + one tile of rows: soffset of a ragged last tile
    return c_bytes < 0x7fffffffull;
}

int eoe_launch_ntp(const void* gemm_p, int dtype, int epi, int mi, hipStream_t s) {
    const GemmP& p = *(const GemmP*)gemm_p;
#ifdef EOE_NTP_DEV         // development builds: one instantiation family only (compile time)
    return mi == 4 ? launch_ntp<f16_t, 4>(p, epi, s) : launch_ntp<f16_t, 5>(p, epi, s);
#else
    if (dtype == EOE_F16) return mi == 4 ? launch_ntp<f16_t, 4>(p, epi, s) : launch_ntp<f16_t, 5>(p, epi, s);
    return mi == 4 ? launch_ntp<bf16_t, 4>(p, epi, s) : launch_ntp<bf16_t, 5>(p, epi, s);
#endif
}
