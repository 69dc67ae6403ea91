// MFMA GEMM (NT) for gfx950.  Three tile shapes, chosen per problem by launch_nt:
//   gemm_nt128_kernel  128x128x64 or 160x128x64 (whichever needs fewer rounds over the 2 x #CU workgroup slots), 4 wavefronts
//                      (2x2, 64x64 or 80x64 each), 2-stage ring, two non-persistent workgroups per CU -- the default;
//   gemm_nt_kernel     256x128x64 (256x96 / 256x64 where they quantise or fit better), 8 wavefronts (4x2, 64x64 each; two per
//                      SIMD), one persistent workgroup per CU, 3-stage ring -- large square problems and the implicit-GEMM
//                      convolutions (GATHER modes);
//   gemm_nt64_kernel   256x64x64, 4 wavefronts stacked along M (64x64 each), 2-stage ring, two workgroups per CU -- N <= 64
//                      (the 64-channel convolutions, with or without gather).
// All: v_mfma_f32_16x16x32_{f16,bf16}, operands staged global -> LDS with bounds-checked LDS-DMA (buffer_load ... lds,
// 16 B per lane; an out-of-range lane reads 0, which is the zero padding of ragged M / N tails), and -- GATHER modes -- the A
// operand of a convolution fetched piece by piece from the NHWC activation (implicit GEMM, no im2col).
//
//   NT:  C[M,N] = A[M,K] . B[N,K]^T   both operands K-contiguous; LDS image [128 rows][64 k] with the 16-B chunk
//        index XOR-swizzled by (row>>1)&7 (applied on the SOURCE address, LDS-DMA writes linearly), fragments
//        by ds_read_b128, conflict-free.
//   TN:  C[M,N] = A[T,M]^T . B[T,N]   reduction over rows (wgrad); LDS image [64 t][128 cols] with the 32-B
//        granule XOR-swizzled by (t&3)|((t>>3)&1)<<2, fragments by ds_read_b64_tr_b16 (hardware transpose).
//
// The MFMA is issued with the weight-side fragment as the A operand so that each lane ends up holding 4
// consecutive output COLUMNS of one output row (D: col = lane&15 -> m, row = 4*(lane>>4)+r -> n): epilogue
// loads/stores are 8-B (16-bit) or 16-B (fp32) vectors along the contiguous dimension.
#include "gemm_common.h"
#include <stdlib.h>
#include <map>
#include <mutex>
#include <type_traits>

unsigned long long* g_stamp_buf = nullptr;
extern int g_tn_flags;     // gemm_tn.hip
extern int g_vit_side_stream;   // vit.cpp
extern int g_attn_flags;        // attention.hip
extern int g_tn256_launches;    // gemm_tn256.hip
extern int g_parity_flags;      // parity.hip
int eoe_launch_nt256(const void* gemm_p, int dtype, int epi, int mi, hipStream_t s);   // gemm256.hip
bool eoe_w8_applies(const void* gemm_p, int epi);                                       // gemm_w8.hip
int eoe_launch_w8(const void* gemm_p, int dtype, int epi, hipStream_t s);
bool eoe_w8_streamk(const void* gemm_p);                                                // would run in its stream-K form (caller's workspace)

namespace {
extern int g_nt_flags;

// ------------------------------------------------------------------------------------------------ NT
// main loop: 3-stage LDS ring filled by LDS-DMA two k-tiles ahead (counted vmcnt: the newest tile stays in
// flight across the barrier), ONE raw s_barrier per k-tile, MFMA fragments double-buffered in registers so the
// ds_reads of the next 32-deep k-step run under the 16 MFMAs of the current one.
//   iteration kt:  stage(kt+2) | F1 = frags(kt, ks1) | MFMA(F0) | vmcnt -> tile kt+1 landed, lgkmcnt(0), barrier |
//                  F0 = frags(kt+1, ks0) | MFMA(F1)
//   WAR: stage(kt+2) overwrites the buffer of tile kt-1, whose last reads (its F1) completed before every wave
//        passed the barrier of iteration kt-1.   RAW: each wave waits for its own DMA pieces of tile kt+1 before
//        the barrier; the reads of that tile come after the barrier.

// NI = 16-column MFMA tiles per wave along N: the workgroup tile is 256 x (32*NI).  NI = 4 (256x128) is the default;
// NI = 3 (256x96) is chosen when it quantises better over the CUs (e.g. N = 768, M = 12800: 400 tiles instead of 300
// -> 2 rounds of 3/4-size tiles).  The LDS image keeps the 128-row B slot; rows >= 32*NI are never fetched.
// FLAGS (tuning switches, A/B-able in one binary when built with -DEOE_AB): bit 0 = LDS-transposed fast epilogue,
// bit 1 = fragment reads as inline asm with explicit waits
// GATHER: the A operand is a convolution's patch matrix that is never materialised -- every 16-B LDS-DMA piece (8
// channels of one tap of one output pixel) is fetched from the NHWC activation with its own address; taps that fall into
// the zero padding get the out-of-range offset and arrive as zeros.  gC % 64 == 0, so a 64-deep k-tile is one tap.
template <typename T, int EPI, int NI, int FLAGS, int GATHER = 0>
__global__ __launch_bounds__(512, 2) void gemm_nt_kernel(GemmP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int BNI = 32 * NI;
    const int tiles_n = (p.N + BNI - 1) / BNI;
    const int total_tiles = tiles_n * ((p.M + BM - 1) / BM);
    const int G = gridDim.x;                       // persistent: this workgroup runs a strided sequence of tiles
    const int my_tiles = (total_tiles - (int)blockIdx.x + G - 1) / G;     // tiles b, b+G, b+2G, ... (XCD-remapped)
    const int nk = p.K / BK;
    const int iters = my_tiles * nk;               // flattened (tile, k-tile) iteration space of this workgroup

    __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.bytesA);
    __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.bytesB);

    // staging cursor: the LDS-DMA ring runs two k-tiles ahead of the MFMAs ACROSS tile boundaries, so the next tile's
    // first operands are already in flight while this tile's epilogue stores drain.
    // a wave-load (1 KiB) covers 8 tile rows of 128 B; lane -> (row, 16-B slot); the slot holds logical chunk
    // slot ^ ((row>>1)&7).  A image: 256 rows = 32 wave-loads (4 per wave); B image: 128 rows = 16 (2 per wave)
    unsigned offA[4], offB[2];
    int gh[4], gw[4];                              // GATHER: top-left input coordinate of each staged row's window
    int gw3[4];                                    // GATHER == 3: element offset of the lane's 16-B piece within a k-tile
    int g_ky = 0, g_kx = 0, g_c0 = 0;              // GATHER: tap / channel offset of the k-tile being staged (uniform)
    int st_tile = 0, st_kt = 0, st_slot = 0;       // st_tile / c_tile count this workgroup's tiles (0 .. my_tiles)
    // (tried and rejected, measured interleaved on one device: giving each XCD a band of whole tile rows swept
    //  column-major to keep A panels L2-resident -- 587 vs 551 us per layer stand-alone, 10.6 vs 8.2 ms in the step)
    auto tile_origin = [&](int seq, int& m0, int& n0) {
        const int r = xcd_remap((int)blockIdx.x + seq * G, total_tiles);
        m0 = (r / tiles_n) * BM;
        n0 = (r % tiles_n) * BNI;
    };
    auto set_offsets = [&](int t) {
        int m0, n0;
        tile_origin(t, m0, n0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (wave * 4 + j) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            const int ga = m0 + row;
            if (GATHER == 2) {
                if (ga < p.M) {
                    const int img = ga / p.gHoWo, rem = ga - img * p.gHoWo;
                    const int ho = rem / p.gWo, wo = rem - ho * p.gWo;
                    offA[j] = (unsigned)((((img * p.gH + ho * p.gstride + (c >> 2)) * p.gW + wo * p.gstride + 2 * (c & 3)) * 4) * 2);
                } else {
                    offA[j] = EOE_OOB;
                }
            } else if (GATHER == 1 || GATHER == 3) {
                if (GATHER == 3) gw3[j] = c * 8;
                if (ga < p.M) {
                    const int img = ga / p.gHoWo, rem = ga - img * p.gHoWo;
                    const int ho = rem / p.gWo, wo = rem - ho * p.gWo;
                    gh[j] = ho * p.gstride - p.gpad;
                    gw[j] = wo * p.gstride - p.gpad;
                    offA[j] = (unsigned)((((img * p.gH + gh[j]) * p.gW + gw[j]) * p.gC + (GATHER == 3 ? 0 : c * 8)) * 2);   // wraps for h < 0: only used when valid
                } else {
                    gh[j] = -(1 << 24);
                    gw[j] = 0;
                    offA[j] = 0;
                }
            } else {
                offA[j] = (ga < p.M) ? (unsigned)(((size_t)ga * p.lda + c * 8) * 2) : EOE_OOB;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (wave * 2 + j) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            const int gb = n0 + row;
            offB[j] = (row < BNI && gb < p.N) ? (unsigned)(((size_t)gb * p.ldb + c * 8) * 2) : EOE_OOB;
        }
    };
    auto stage_next = [&]() {                      // 6 LDS-DMA instructions per wave per k-tile
        char* sa = smem + st_slot * STAGE_BYTES;
        char* sb = sa + A_BYTES;
        const unsigned k0 = (unsigned)st_kt * (BK * 2u);
        const unsigned kA = (GATHER == 2) ? (unsigned)(st_kt * p.gkstep) : k0;
        if (GATHER == 1) {
            const unsigned delta = (unsigned)(((g_ky * p.gW + g_kx) * p.gC + g_c0) * 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = (unsigned)(gh[j] + g_ky) < (unsigned)p.gH && (unsigned)(gw[j] + g_kx) < (unsigned)p.gW;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(sa + (wave * 4 + j) * 1024), 16,
                                                         ok ? offA[j] + delta : EOE_OOB, 0, 0, 0);
            }
            g_c0 += BK;
            if (g_c0 == p.gC) {
                g_c0 = 0;
                if (++g_kx == p.gkw) { g_kx = 0; ++g_ky; }
            }
        } else if (GATHER == 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kg = st_kt * BK + gw3[j];
                const int tap = kg >> p.glog2c, ch = kg & (p.gC - 1);
                const int ky = (int)(((unsigned)tap * p.gmagic) >> 16), kx = tap - ky * p.gkw;
                const bool ok = ky < p.gkh && (unsigned)(gh[j] + ky) < (unsigned)p.gH && (unsigned)(gw[j] + kx) < (unsigned)p.gW;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(sa + (wave * 4 + j) * 1024), 16,
                                                         ok ? offA[j] + (unsigned)(((ky * p.gW + kx) * p.gC + ch) * 2) : EOE_OOB, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(sa + (wave * 4 + j) * 1024), 16, offA[j] + kA, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_t*)(sb + (wave * 2 + j) * 1024), 16, offB[j] + k0, 0, 0, 0);
        st_slot = (st_slot == NSTAGE - 1) ? 0 : st_slot + 1;
        if (++st_kt == nk) {
            st_kt = 0;
            g_ky = 0; g_kx = 0; g_c0 = 0;
            st_tile += 1;
            if (st_tile < my_tiles) set_offsets(st_tile);
        }
    };

    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * (16 * NI);
    const int lr = lane & 15, lg = lane >> 4;
    // B rows of this wave start at wn0 = 0 or 16*NI: (row>>1)&7 of row = wn0 + 16 i + lr depends on wn0 when NI is odd
    const int swA = (lr >> 1) & 7, swB = ((wn0 + lr) >> 1) & 7;
    const int fragA = (wm0 + lr) * 128, fragB = A_BYTES + (wn0 + lr) * 128;
    const int chA0 = ((0 + lg) ^ swA) * 16, chA1 = ((4 + lg) ^ swA) * 16;
    const int chB0 = ((0 + lg) ^ swB) * 16, chB1 = ((4 + lg) ^ swB) * 16;
    typedef typename T16<T>::v8 V8;

    f32x4 acc[4][NI];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // FLAGS bit 1: fragment reads as inline asm (opaque to the compiler's waitcnt bookkeeping; completion awaited by
    // EOE_LANDED, an lgkmcnt(0) that also "rewrites" the fragment registers so no MFMA can be scheduled above it) -- the
    // A/B of what fixed the wgrad kernel (gemm_tn.hip, tr_frag_asm)
#define EOE_READ(XA, WB, base, ks)                                                        \
    if (FLAGS & 2) {                                                                      \
        const unsigned a_ = (unsigned)((base) - smem) + fragA + ((ks) ? chA1 : chA0);     \
        const unsigned b_ = (unsigned)((base) - smem) + fragB + ((ks) ? chB1 : chB0);     \
        asm volatile("ds_read_b128 %0, %1" : "=v"(XA[0]) : "v"(a_) : "memory");           \
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(XA[1]) : "v"(a_) : "memory"); \
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(XA[2]) : "v"(a_) : "memory"); \
        asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(XA[3]) : "v"(a_) : "memory"); \
        asm volatile("ds_read_b128 %0, %1" : "=v"(WB[0]) : "v"(b_) : "memory");           \
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(WB[1]) : "v"(b_) : "memory"); \
        if (NI > 2) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(WB[2]) : "v"(b_) : "memory"); \
        if (NI > 3) asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(WB[3 < NI ? 3 : 0]) : "v"(b_) : "memory"); \
    } else {                                                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                     \
            XA[i] = *(const V8*)((base) + fragA + i * 2048 + ((ks) ? chA1 : chA0));       \
        _Pragma("unroll") for (int i = 0; i < NI; ++i)                                    \
            WB[i] = *(const V8*)((base) + fragB + i * 2048 + ((ks) ? chB1 : chB0));       \
    }
#define EOE_LANDED(XA, WB)                                                                \
    do {                                                                                  \
        if (NI == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(XA[0]), "+v"(XA[1]), "+v"(XA[2]), "+v"(XA[3]), "+v"(WB[0]), \
                                  "+v"(WB[1]), "+v"(WB[2 < NI ? 2 : 0]), "+v"(WB[3 < NI ? 3 : 0]) :: "memory");           \
        else if (NI == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(XA[0]), "+v"(XA[1]), "+v"(XA[2]), "+v"(XA[3]),        \
                                       "+v"(WB[0]), "+v"(WB[1]), "+v"(WB[2 < NI ? 2 : 0]) :: "memory");                    \
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(XA[0]), "+v"(XA[1]), "+v"(XA[2]), "+v"(XA[3]), "+v"(WB[0]),        \
                          "+v"(WB[1]) :: "memory");                                                                         \
    } while (0)
#define EOE_MFMA(XA, WB)                                                  \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                      \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = T16<T>::mfma16(WB[ni], XA[mi], acc[mi][ni]);

    if (iters <= 0) return;
    unsigned long long* stp = p.stamp ? p.stamp + (size_t)blockIdx.x * 16 : nullptr;
    int sti = 0;
    unsigned long long acc_vm = 0, acc_bar = 0;    // diagnostics: cycles this wave spent in the DMA wait / at the barrier
#define EOE_STAMP() do { if (stp && tid == 0 && sti < 12) stp[sti++] = __builtin_amdgcn_s_memtime(); } while (0)
    EOE_STAMP();                                   // [0] kernel entry
    V8 xa0[4], wb0[NI], xa1[4], wb1[NI];
    set_offsets(st_tile);
    stage_next();
    if (iters > 1) {
        stage_next();
        EOE_WAIT_VM(6);
    } else {
        EOE_WAIT_VM(0);
    }
    __builtin_amdgcn_s_barrier();
    EOE_STAMP();                                   // [1] first operands landed
    EOE_READ(xa0, wb0, smem, 0);
    int cur = 0;                                   // ring slot of the k-tile being multiplied
    int c_tile = 0, c_kt = 0;
    for (int it = 0; it < iters; ++it) {
        const int nxt = (cur == NSTAGE - 1) ? 0 : cur + 1;
        const char* sc = smem + cur * STAGE_BYTES;
        if (it + 2 < iters) stage_next();
        if (FLAGS & 2) { EOE_LANDED(xa0, wb0); }
        EOE_READ(xa1, wb1, sc, 1);
        EOE_MFMA(xa0, wb0);
        unsigned long long tq0 = 0, tq1 = 0;
        if (stp) tq0 = __builtin_amdgcn_s_memtime();
        if (it + 2 < iters) { EOE_WAIT_VM(6); } else { EOE_WAIT_VM(0); }
        if (FLAGS & 2) { EOE_LANDED(xa1, wb1); } else { EOE_WAIT_LGKM0(); }
        if (stp) tq1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        if (stp) { const unsigned long long tq2 = __builtin_amdgcn_s_memtime(); acc_vm += tq1 - tq0; acc_bar += tq2 - tq1; }
        {   // unconditional (the last iteration reads a stale ring slot and discards it): keeps the compiler's
            // lgkmcnt bookkeeping exact, so MFMA(F1) does not wait for these reads
            const char* sn = smem + nxt * STAGE_BYTES;
            EOE_READ(xa0, wb0, sn, 0);
        }
        EOE_MFMA(xa1, wb1);
        cur = nxt;
        if (++c_kt == nk) {                        // tile finished: epilogue while the next tile's DMA is in flight
            int m0, n0;
            tile_origin(c_tile, m0, n0);
            EOE_STAMP();                           // [2+2i] main loop of tile i done
            if (FLAGS & 2) { EOE_LANDED(xa0, wb0); }   // asm reads in flight are invisible to the compiler: none may be
                                                       // outstanding when the epilogue starts reusing registers
            GemmP ep;
            load_epilogue_args(ep, p);
            if (FLAGS & 1) epilogue<T, EPI, NI>(ep, acc, m0 + wm0, n0 + wn0, lane, (char*)sc + wave * 4096);
            else epilogue_generic<T, EPI, NI, 4>(ep, acc, m0 + wm0, n0 + wn0, lane);
            if (stp) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            EOE_STAMP();                           // [3+2i] epilogue of tile i issued and drained
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            c_kt = 0;
            c_tile += 1;
        }
    }
#undef EOE_READ
#undef EOE_MFMA
#undef EOE_LANDED
    if (stp && lane == 0 && wave < 2) { stp[12 + wave * 2 - 0] = acc_vm; stp[13 + wave * 2 - 0] = acc_bar; }
}

// ------------------------------------------------------------------------------------------------ NT, 128x128 tiles
// Second tile shape: 128x128x64 tiles, FOUR wavefronts (2x2, 64x64 each), a 2-stage 64
// KiB ring -> TWO workgroups per CU.  The 256x128 kernel leaves the MFMA pipe idle while its one workgroup per CU is in a
// prologue, at the per-k-tile barrier (~25 % of a wave's time) or in the epilogue (6-18 k cycles per tile, HBM-saturating
// for the fp32 residual outputs while every workgroup is in that phase together); with two independent workgroups per CU
// each SIMD holds one wave of each, so one workgroup's stalls are the other's issue slots.  Non-persistent on purpose: one tile
// per workgroup, dispatched as slots free up, which de-synchronises the two workgroups of a CU (a persistent tile loop with
// cross-tile prefetch kept them in lockstep and measured 5-13 % slower on the heavy-epilogue shapes).  Tiles XCD-remapped.  Same LDS image / swizzle / fragment layout / epilogues as gemm_nt_kernel.
// (round 2, measured and rejected: s_setprio 2 for the MFMA loop and 0 for the epilogue, or the reverse -- the co-resident workgroup's
//  epilogue VALU against this one's MFMA issue: layer total 473.2 / 475.3 vs 474.4 us, i.e. nothing)
// MI = 16-row MFMA tiles per wave along M: the workgroup tile is (32*MI) x 128.  MI = 4 (128x128) is the default; MI = 5
// (160x128) is chosen when it quantises better over the 2 x #CU workgroup slots (N = 768, M = 12800: 480 tiles in one round
// instead of 600 in two).
constexpr int B128_BYTES = 128 * BK * 2;                                          // 16 KiB
constexpr int a128_bytes(int MI) { return 32 * MI * BK * 2; }                     // 16 / 20 KiB
constexpr int stage128_bytes(int MI) { return a128_bytes(MI) + B128_BYTES; }      // 32 / 36 KiB
constexpr int smem128_bytes(int MI) { return 2 * stage128_bytes(MI); }            // 64 / 72 KiB: two workgroups per CU

// GATHER = 1: A is a convolution's implicit patch matrix (C % 64 == 0: one tap per k-tile), as in gemm_nt64_kernel below -- the layers with
// cout >= 128 ran on the persistent 256x128 kernel, whose 784 tiles at M = 200704 (28 x 28 maps) are four rounds over 256 CUs for 3.06 rounds
// of work; 160-row tiles on the 2 x #CU slots quantise at 98 % there.
template <typename T, int EPI, int MI, int GATHER = 0>
__global__ __launch_bounds__(256, 2) void gemm_nt128_kernel(GemmP p) {
    constexpr int BM = 32 * MI, A128_BYTES = a128_bytes(MI), STAGE128_BYTES = stage128_bytes(MI);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NI = 4;
    const int tiles_n = (p.N + 127) / 128;
    const int total_tiles = tiles_n * ((p.M + BM - 1) / BM);
    const int r = xcd_remap((int)blockIdx.x, total_tiles);
    // (xcd_halves -- the column-half order of the 160 x 256 kernel below -- measured slightly worse here: N = 768 is six tile columns, c_proj
    //  forward 0.865 -> 0.879 ms per step)
    const int mt = r / tiles_n, nt = r % tiles_n;
    const int m0 = mt * BM, n0 = nt * 128;
    // split k (launch_nt128_splitk: small M, long K): gridDim.y equal k-ranges, workgroup y writes its fp32 partial tile to slice y of C
    const int nk = (p.K / BK) / (int)gridDim.y;
    const unsigned kbase = (unsigned)blockIdx.y * (unsigned)nk * (BK * 2u);

    __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.bytesA);
    __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.bytesB);
    // a wave-load covers 8 rows: MI per wave for A (BM rows), 4 per wave for B (128 rows);
    // lane -> (row, 16-B slot holding chunk slot ^ ((row>>1)&7))
    // offA has a FIXED extent: with `unsigned offA[MI]` feeding the LDS-DMA builtin hipcc (ROCm 7.2) silently drops the HOST
    // stub of every instantiation of this kernel and the library fails to load with an undefined symbol
    unsigned offA[5], offB[4];
    static_assert(MI <= 5, "offA");
    int gh[5], gw[5];                               // GATHER: top-left input coordinate of each staged row's window
    int g_ky = 0, g_kx = 0, g_c0 = 0;               // GATHER: tap / channel offset of the next k-tile to be staged (uniform)
#pragma unroll
    for (int j = 0; j < MI; ++j) {
        const int row = (wave * MI + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int ga = m0 + row;
        if (GATHER == 1) {
            if (ga < p.M) {
                const int img = ga / p.gHoWo, rem = ga - img * p.gHoWo;
                const int ho = rem / p.gWo, wo = rem - ho * p.gWo;
                gh[j] = ho * p.gstride - p.gpad;
                gw[j] = wo * p.gstride - p.gpad;
                offA[j] = (unsigned)((((img * p.gH + gh[j]) * p.gW + gw[j]) * p.gC + c * 8) * 2);   // wraps for h < 0: only used when valid
            } else {
                gh[j] = -(1 << 24);
                gw[j] = 0;
                offA[j] = 0;
            }
        } else {
            gh[j] = 0; gw[j] = 0;
            offA[j] = (ga < p.M) ? (unsigned)(((size_t)ga * p.lda + c * 8) * 2) : EOE_OOB;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wave * 4 + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        offB[j] = (n0 + row < p.N) ? (unsigned)(((size_t)(n0 + row) * p.ldb + c * 8) * 2) : EOE_OOB;
    }
    // MI + 4 LDS-DMA instructions per wave per k-tile
#define EOE_STAGE128(slot, kt)                                                                                              \
    do {                                                                                                                    \
        char* sa_ = smem + (slot) * STAGE128_BYTES;                                                                         \
        char* sb_ = sa_ + A128_BYTES;                                                                                       \
        const unsigned k0_ = kbase + (unsigned)(kt) * (BK * 2u);                                                            \
        if (GATHER == 1) {                                                                                                  \
            /* k-tiles are staged in order 0, 1, 2, ...: the tap cursor advances with them */                                 \
            const unsigned delta_ = (unsigned)(((g_ky * p.gW + g_kx) * p.gC + g_c0) * 2);                                   \
            _Pragma("unroll") for (int j = 0; j < MI; ++j) {                                                                \
                const bool ok_ = (unsigned)(gh[j] + g_ky) < (unsigned)p.gH && (unsigned)(gw[j] + g_kx) < (unsigned)p.gW;    \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(sa_ + (wave * MI + j) * 1024), 16,               \
                                                         ok_ ? offA[j] + delta_ : EOE_OOB, 0, 0, 0);                        \
            }                                                                                                               \
            g_c0 += BK;                                                                                                     \
            if (g_c0 == p.gC) {                                                                                             \
                g_c0 = 0;                                                                                                   \
                if (++g_kx == p.gkw) { g_kx = 0; ++g_ky; }                                                                  \
            }                                                                                                               \
        } else {                                                                                                            \
            _Pragma("unroll") for (int j = 0; j < MI; ++j)                                                                  \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(sa_ + (wave * MI + j) * 1024), 16, offA[j] + k0_, 0, 0, 0); \
        }                                                                                                                   \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                       \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_t*)(sb_ + (wave * 4 + j) * 1024), 16, offB[j] + k0_, 0, 0, 0); \
    } while (0)

    const int wm0 = (wave >> 1) * (16 * MI), wn0 = (wave & 1) * 64;
    const int lr = lane & 15, lg = lane >> 4;
    const int sw = (lr >> 1) & 7;                        // wm0, wn0 are multiples of 16: (row>>1)&7 depends on lr only
    const int fragA = (wm0 + lr) * 128, fragB = A128_BYTES + (wn0 + lr) * 128;
    const int ch0 = ((0 + lg) ^ sw) * 16, ch1 = ((4 + lg) ^ sw) * 16;
    typedef typename T16<T>::v8 V8;

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#define EOE_READ128(XA, WB, base, ks)                                                     \
    _Pragma("unroll") for (int i = 0; i < MI; ++i)                                        \
        XA[i] = *(const V8*)((base) + fragA + i * 2048 + ((ks) ? ch1 : ch0));             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                         \
        WB[i] = *(const V8*)((base) + fragB + i * 2048 + ((ks) ? ch1 : ch0));
#define EOE_MFMA128(XA, WB)                                               \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                     \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = T16<T>::mfma16(WB[ni], XA[mi], acc[mi][ni]);

    // 2-stage ring with register double-buffered fragments (the schedule of gemm_nt_kernel with one stage less):
    //   iteration kt:  F1 = frags(kt, ks1) | MFMA(F0) | vmcnt(0): tile kt+1 landed, lgkmcnt(0), barrier |
    //                  stage(kt+2) into the slot just consumed | F0 = frags(kt+1, ks0) | MFMA(F1)
    //   WAR: stage(kt+2) overwrites slot kt&1 after the barrier behind which every wave's reads of it (F0 in iteration kt-1,
    //        F1 in this one) have completed.   RAW: the DMA of tile kt+1 was issued one iteration ago; each wave waits for its
    //        own pieces before the barrier, the reads come after it.
    // diagnostics (EOE_GEMM_STAMP=1): 8 words per workgroup = entry, operands landed, main loop done, exit, HW_ID, XCC_ID
    unsigned long long* stp = (p.stamp && blockIdx.x < 8192) ? p.stamp + (size_t)blockIdx.x * 8 : nullptr;
    if (stp && tid == 0) {
        stp[0] = __builtin_amdgcn_s_memtime();
        stp[4] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        stp[5] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
    V8 xa0[MI], wb0[4], xa1[MI], wb1[4];
    EOE_STAGE128(0, 0);
    if (nk > 1) EOE_STAGE128(1, 1);
    if (nk > 1) {                                  // tile 0 landed, tile 1 (MI + 4 loads per wave) may stay in flight
        if (MI == 5) { EOE_WAIT_VM(9); } else { EOE_WAIT_VM(8); }
    } else {
        EOE_WAIT_VM(0);
    }
    __builtin_amdgcn_s_barrier();
    if (stp && tid == 0) stp[1] = __builtin_amdgcn_s_memtime();
    EOE_READ128(xa0, wb0, smem, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const char* sc = smem + (kt & 1) * STAGE128_BYTES;
        const char* sn = smem + ((kt + 1) & 1) * STAGE128_BYTES;
        EOE_READ128(xa1, wb1, sc, 1);
        EOE_MFMA128(xa0, wb0);
        if (stp) {                                 // diagnostics: cycles wave 0 spends waiting for the DMA / at the barrier
            const unsigned long long t_a = __builtin_amdgcn_s_memtime();
            EOE_WAIT_VM(0);
            EOE_WAIT_LGKM0();
            const unsigned long long t_b = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            if (tid == 0) { stp[6] += t_b - t_a; stp[7] += __builtin_amdgcn_s_memtime() - t_b; }
        } else {
            EOE_WAIT_VM(0);
            EOE_WAIT_LGKM0();
            __builtin_amdgcn_s_barrier();
        }
        if (kt + 2 < nk) EOE_STAGE128(kt & 1, kt + 2);
        EOE_READ128(xa0, wb0, sn, 0);          // unconditional (the last one reads a stale slot and is discarded)
        EOE_MFMA128(xa1, wb1);
    }
    EOE_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();               // every wave is done reading the ring before the epilogue's scratch use
#undef EOE_READ128
#undef EOE_MFMA128
#undef EOE_STAGE128
    if (stp && tid == 0) stp[2] = __builtin_amdgcn_s_memtime();
    GemmP ep;
    load_epilogue_args(ep, p);
    if (gridDim.y > 1) ep.C = (char*)ep.C + (size_t)blockIdx.y * (size_t)ep.M * ep.ldc * sizeof(float);      // split k: slice y
    // scratch: this wave's own 4 KiB of slot 0 (all reads of the ring are behind the final barrier)
    epilogue<T, EPI, NI, MI>(ep, acc, m0 + wm0, n0 + wn0, lane, smem + wave * 4096);
    if (stp && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // diagnostics only: the stores of wave 0 have left
        stp[3] = __builtin_amdgcn_s_memtime();
    }
}

// ------------------------------------------------------------------------------------------------ NT, 160x256x32 tiles, two workgroups per CU
// Fourth tile shape (round 3; nt_flags bit 12 = 4096 switches it off).  Why: the 160x128 kernel above is LDS-BANDWIDTH bound -- its 80x64
// wave tiles read 9 KB of fragments per 20 MFMAs, 112 B/clk per CU, and the LDS-DMA of the same k-tile writes 57 B/clk more: 169 B/clk asked
// of a 128 B/clk LDS.  Here a wave owns 80x128 (13 KB per 40 MFMAs: 81 B/clk + 41 B/clk of DMA); to keep TWO workgroups on a CU the k-tile is 32
// deep (3 x 26 KB of LDS per workgroup) and a wave stays inside 256 registers (160 accumulators, single-buffered fragments).  LDS image of a
// k-tile: 16-row blocks of 1 KB (one LDS-DMA instruction each), row r of a block at r * 64 B, its four 16-byte k-chunks XOR-ed with (r >> 2) & 3:
// a fragment read (16 rows x one chunk) then touches 16 different 16-byte bank groups.  (rocprofv3 still counts SQ_LDS_BANK_CONFLICT = 5 M
// cycles per launch for this image; the 160x128 kernel's 128-byte-row image rebuilt from pairs of row blocks counts 0 and measured 0.04 ms
// per step SLOWER, two builds interleaved on one box -- kept as is.)  Same products in the same k order as the other kernels: same bits.
// MI = 5 only.
constexpr int BKW = 32;
constexpr int AW_BYTES = 160 * BKW * 2;              // 10 KiB: ten 16-row blocks
constexpr int BW_BYTES = 256 * BKW * 2;              // 16 KiB: sixteen blocks
constexpr int STAGEW_BYTES = AW_BYTES + BW_BYTES;    // 26 KiB
constexpr int NSTAGEW = 3;
constexpr int SMEMW_BYTES = NSTAGEW * STAGEW_BYTES;  // 78 KiB: two workgroups per CU (156 of 160 KiB)

template <typename T, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt128w_kernel(GemmP p) {
    constexpr int MI = 5, BM = 160;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = (p.N + 255) / 256;
    const int total_tiles = tiles_n * ((p.M + BM - 1) / BM);
    const int r = xcd_remap((int)blockIdx.x, total_tiles);
    int mt, nt;
    xcd_halves(r, total_tiles, tiles_n, !(p.dbg & 32), mt, nt);
    const int m0 = mt * BM, n0 = nt * 256;
    const int nk = p.K / BKW;

    __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.bytesA);
    __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.bytesB);
    // one LDS-DMA instruction = one 16-row block: lane -> (row lane >> 2, chunk (lane & 3) ^ ((row >> 2) & 3)).  A: blocks wave, wave + 4,
    // wave + 8 (< 10); B: blocks wave, wave + 4, wave + 8, wave + 12
    // (round 5) the XOR term is (0 - (row >> 2)) & 3 = {0, 3, 2, 1}, not (row >> 2) & 3: a ds_read_b128 is served in the lane groups
    // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS), i.e. rows 0-3 and 12-15 of chunk lg with rows 4-11 of chunk lg ^ 1 --
    // with the plain term rows a and a + 4 (and a + 8, a + 12) met on the same banks two ways (SQ_LDS_BANK_CONFLICT 5.0 M cycles per launch);
    // nt_flags bit 23 = 8388608 restores the old image (A/B)
    const int swz_old = (p.dbg & 64) ? 1 : 0;
    auto swz = [&](int row) -> int { const int x = (row >> 2) & 3; return swz_old ? x : ((0 - x) & 3); };
    const int srow = lane >> 2, schunk = (lane & 3) ^ swz(srow);
    unsigned offA[3], offB[4];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int ga = m0 + (wave + 4 * t) * 16 + srow;
        offA[t] = (wave + 4 * t < 2 * MI && ga < p.M) ? (unsigned)(((size_t)ga * p.lda + schunk * 8) * 2) : EOE_OOB;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int gb = n0 + (wave + 4 * t) * 16 + srow;
        offB[t] = (gb < p.N) ? (unsigned)(((size_t)gb * p.ldb + schunk * 8) * 2) : EOE_OOB;
    }
    const bool a3 = wave + 8 < 2 * MI;               // waves 0, 1 stage three A blocks, waves 2, 3 two (uniform)
#define EOE_STAGEW(slot, kt)                                                                                                \
    do {                                                                                                                    \
        char* sa_ = smem + (slot) * STAGEW_BYTES;                                                                           \
        char* sb_ = sa_ + AW_BYTES;                                                                                         \
        const unsigned k0_ = (unsigned)(kt) * (BKW * 2u);                                                                   \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(sa_ + wave * 1024), 16, offA[0] + k0_, 0, 0, 0);         \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(sa_ + (wave + 4) * 1024), 16, offA[1] + k0_, 0, 0, 0);   \
        if (a3) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(sa_ + (wave + 8) * 1024), 16, offA[2] + k0_, 0, 0, 0); \
        _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                                       \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_t*)(sb_ + (wave + 4 * t) * 1024), 16, offB[t] + k0_, 0, 0, 0); \
    } while (0)

    const int wm0 = (wave >> 1) * (16 * MI), wn0 = (wave & 1) * 128;
    const int lr = lane & 15, lg = lane >> 4;
    const int fo = lr * 64 + ((lg ^ swz(lr)) * 16);                              // this lane's piece of a 16-row block
    const int fragA = (wave >> 1) * MI * 1024 + fo, fragB = AW_BYTES + (wave & 1) * 8 * 1024 + fo;
    typedef typename T16<T>::v8 V8;

    f32x4 accL[MI][4], accR[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            accL[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            accR[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    // (The same weaving in the 2-stage 160x128 kernel above -- pieces of tile kt + 2 among the second half's MFMAs -- measured 10.49 -> 10.95 ms
    //  per step: with one tile of lead the later issue is paid in full at the next wait.  Here the third stage pays for it.)
    // 3-stage ring.  Tile kt is multiplied from slot kt % 3 while tile kt + 1 lands and the pieces of tile kt + 2 are ISSUED BETWEEN the MFMAs of
    // this iteration (into the slot iteration kt - 1 freed): behind the barrier, in a burst, an LDS-DMA instruction costs the issuing wave ~100
    // cycles against ~30 spread out (gemm256's stamps), seven of them against a 640-cycle MFMA block.
    // A wave issues 7 (waves 0, 1) or 6 loads per tile.
#define EOE_PIECE_A(t, slot_, kt_)                                                                                            \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(smem + (slot_) * STAGEW_BYTES + (wave + 4 * (t)) * 1024), 16,       \
                                             offA[t] + (unsigned)(kt_) * (BKW * 2u), 0, 0, 0)
#define EOE_PIECE_B(t, slot_, kt_)                                                                                            \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_t*)(smem + (slot_) * STAGEW_BYTES + AW_BYTES + (wave + 4 * (t)) * 1024), 16, \
                                             offB[t] + (unsigned)(kt_) * (BKW * 2u), 0, 0, 0)
    EOE_STAGEW(0, 0);
    if (nk > 1) {
        EOE_STAGEW(1, 1);
        if (a3) { EOE_WAIT_VM(7); } else { EOE_WAIT_VM(6); }        // tile 0 landed, tile 1 may stay in flight
    } else {
        EOE_WAIT_VM(0);
    }
    __builtin_amdgcn_s_barrier();
    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const char* sc = smem + slot * STAGEW_BYTES;
        const int fslot = slot == 0 ? 2 : slot - 1;           // the slot iteration kt - 1 read (kt = 0: the third, untouched one)
        const bool more = kt + 2 < nk;
        V8 xa[MI], wl[4], wr[4];
#pragma unroll
        for (int i = 0; i < MI; ++i) xa[i] = *(const V8*)(sc + fragA + i * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) wl[i] = *(const V8*)(sc + fragB + i * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) wr[i] = *(const V8*)(sc + fragB + (4 + i) * 1024);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) accL[mi][ni] = T16<T>::mfma16(wl[ni], xa[mi], accL[mi][ni]);
            if (more) {                                       // one piece per four MFMAs
                if (mi == 0) EOE_PIECE_A(0, fslot, kt + 2);
                if (mi == 1) EOE_PIECE_A(1, fslot, kt + 2);
                if (mi == 2 && a3) EOE_PIECE_A(2, fslot, kt + 2);
                if (mi == 3) EOE_PIECE_B(0, fslot, kt + 2);
                if (mi == 4) EOE_PIECE_B(1, fslot, kt + 2);
            }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) accR[mi][ni] = T16<T>::mfma16(wr[ni], xa[mi], accR[mi][ni]);
            if (more) {
                if (mi == 0) EOE_PIECE_B(2, fslot, kt + 2);
                if (mi == 1) EOE_PIECE_B(3, fslot, kt + 2);
            }
        }
        // tile kt + 1 landed (this wave's pieces; those of tile kt + 2, just issued, may stay in flight), every wave done reading this slot
        if (more) { if (a3) { EOE_WAIT_VM(7); } else { EOE_WAIT_VM(6); } } else { EOE_WAIT_VM(0); }
        EOE_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
        slot = slot == NSTAGEW - 1 ? 0 : slot + 1;
    }
#undef EOE_PIECE_A
#undef EOE_PIECE_B
#undef EOE_STAGEW
    GemmP ep;
    load_epilogue_args(ep, p);
    // two 64-column halves through the 160x128 kernel's epilogue; scratch: this wave's own 4 KiB of slot 0 (all ring reads are behind the last barrier)
    epilogue<T, EPI, 4, MI>(ep, accL, m0 + wm0, n0 + wn0, lane, smem + wave * 4096);
    epilogue<T, EPI, 4, MI>(ep, accR, m0 + wm0, n0 + wn0 + 64, lane, smem + wave * 4096);
}

// ------------------------------------------------------------------------------------------------ NT, N <= 64
// Third tile shape, for cout = 64 layers (plain epilogue): 256x64x64, four waves stacked along M (64x64 each -- the same
// 16 MFMAs per 8 fragment reads as the other kernels; the 256x64 tile of the persistent kernel gives a wave 64x32 = 8 MFMAs
// per 6 reads and ran the 64-channel WideResNet convolutions at 355 TF), 2-stage 80 KB ring, two non-persistent workgroups
// per CU, the schedule of gemm_nt128_kernel.  GATHER = 1: A is the implicit patch matrix (C % 64 == 0: one tap per k-tile).
constexpr int A64_BYTES = 256 * BK * 2;             // 32 KiB
constexpr int B64_BYTES = 64 * BK * 2;              // 8 KiB
constexpr int STAGE64_BYTES = A64_BYTES + B64_BYTES;
constexpr int SMEM64_BYTES = 2 * STAGE64_BYTES;     // 80 KiB: two workgroups fill the CU's 160 KiB

template <typename T, int GATHER>
__global__ __launch_bounds__(256, 2) void gemm_nt64_kernel(GemmP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NI = 4;
    const int total_tiles = (p.M + 255) / 256;
    const int m0 = xcd_remap((int)blockIdx.x, total_tiles) * 256;
    const int nk = p.K / BK;

    __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.bytesA);
    __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.bytesB);
    // a wave-load covers 8 rows: 8 per wave for A (256 rows), 2 per wave for B (64 rows)
    unsigned offA[8], offB[2];
    int gh[8], gw[8];
    int g_ky = 0, g_kx = 0, g_c0 = 0;              // GATHER: tap / channel offset of the next k-tile to be staged (uniform)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = (wave * 8 + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int ga = m0 + row;
        if (GATHER == 1) {
            if (ga < p.M) {
                const int img = ga / p.gHoWo, rem = ga - img * p.gHoWo;
                const int ho = rem / p.gWo, wo = rem - ho * p.gWo;
                gh[j] = ho * p.gstride - p.gpad;
                gw[j] = wo * p.gstride - p.gpad;
                offA[j] = (unsigned)((((img * p.gH + gh[j]) * p.gW + gw[j]) * p.gC + c * 8) * 2);   // wraps for h < 0: only used when valid
            } else {
                gh[j] = -(1 << 24);
                gw[j] = 0;
                offA[j] = 0;
            }
        } else if (GATHER == 2) {
            // the stem over the zero-padded NHWC4 image (packed k axis: a k-tile = two kernel rows x 8 pixels x 4 channels; gemm_nt_kernel)
            gh[j] = 0; gw[j] = 0;
            if (ga < p.M) {
                const int img = ga / p.gHoWo, rem = ga - img * p.gHoWo;
                const int ho = rem / p.gWo, wo = rem - ho * p.gWo;
                offA[j] = (unsigned)((((img * p.gH + ho * p.gstride + (c >> 2)) * p.gW + wo * p.gstride + 2 * (c & 3)) * 4) * 2);
            } else {
                offA[j] = EOE_OOB;
            }
        } else {
            gh[j] = 0; gw[j] = 0;
            offA[j] = (ga < p.M) ? (unsigned)(((size_t)ga * p.lda + c * 8) * 2) : EOE_OOB;
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        offB[j] = (row < p.N) ? (unsigned)(((size_t)row * p.ldb + c * 8) * 2) : EOE_OOB;
    }
    // 10 LDS-DMA instructions per wave per k-tile; k-tiles are staged in order 0, 1, 2, ... (the tap cursor advances with them)
#define EOE_STAGE64(slot, kt)                                                                                                \
    do {                                                                                                                     \
        char* sa_ = smem + (slot) * STAGE64_BYTES;                                                                           \
        char* sb_ = sa_ + A64_BYTES;                                                                                         \
        const unsigned k0_ = (unsigned)(kt) * (BK * 2u);                                                                     \
        if (GATHER == 1) {                                                                                                   \
            const unsigned delta_ = (unsigned)(((g_ky * p.gW + g_kx) * p.gC + g_c0) * 2);                                    \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                                  \
                const bool ok_ = (unsigned)(gh[j] + g_ky) < (unsigned)p.gH && (unsigned)(gw[j] + g_kx) < (unsigned)p.gW;     \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(sa_ + (wave * 8 + j) * 1024), 16,                 \
                                                         ok_ ? offA[j] + delta_ : EOE_OOB, 0, 0, 0);                         \
            }                                                                                                                \
            g_c0 += BK;                                                                                                      \
            if (g_c0 == p.gC) {                                                                                              \
                g_c0 = 0;                                                                                                    \
                if (++g_kx == p.gkw) { g_kx = 0; ++g_ky; }                                                                   \
            }                                                                                                                \
        } else {                                                                                                             \
            const unsigned kA_ = (GATHER == 2) ? (unsigned)((kt) * p.gkstep) : k0_;                                          \
            _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                                    \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(sa_ + (wave * 8 + j) * 1024), 16, offA[j] + kA_, 0, 0, 0); \
        }                                                                                                                    \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                        \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_t*)(sb_ + (wave * 2 + j) * 1024), 16, offB[j] + k0_, 0, 0, 0); \
    } while (0)

    const int wm0 = wave * 64;
    const int lr = lane & 15, lg = lane >> 4;
    const int sw = (lr >> 1) & 7;
    const int fragA = (wm0 + lr) * 128, fragB = A64_BYTES + lr * 128;
    const int ch0 = ((0 + lg) ^ sw) * 16, ch1 = ((4 + lg) ^ sw) * 16;
    typedef typename T16<T>::v8 V8;

    f32x4 acc[4][NI];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#define EOE_READ64(XA, WB, base, ks)                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                       \
        XA[i] = *(const V8*)((base) + fragA + i * 2048 + ((ks) ? ch1 : ch0));             \
        WB[i] = *(const V8*)((base) + fragB + i * 2048 + ((ks) ? ch1 : ch0));             \
    }
#define EOE_MFMA64(XA, WB)                                                \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                      \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = T16<T>::mfma16(WB[ni], XA[mi], acc[mi][ni]);

    // schedule, WAR / RAW argument: gemm_nt128_kernel
    V8 xa0[4], wb0[4], xa1[4], wb1[4];
    EOE_STAGE64(0, 0);
    if (nk > 1) EOE_STAGE64(1, 1);
    if (nk > 1) { EOE_WAIT_VM(10); } else { EOE_WAIT_VM(0); }
    __builtin_amdgcn_s_barrier();
    EOE_READ64(xa0, wb0, smem, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const char* sc = smem + (kt & 1) * STAGE64_BYTES;
        const char* sn = smem + ((kt + 1) & 1) * STAGE64_BYTES;
        EOE_READ64(xa1, wb1, sc, 1);
        EOE_MFMA64(xa0, wb0);
        EOE_WAIT_VM(0);
        EOE_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) EOE_STAGE64(kt & 1, kt + 2);
        EOE_READ64(xa0, wb0, sn, 0);           // unconditional (the last one reads a stale slot and is discarded)
        EOE_MFMA64(xa1, wb1);
    }
    EOE_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();               // every wave is done reading the ring before the epilogue's scratch use
#undef EOE_READ64
#undef EOE_MFMA64
#undef EOE_STAGE64
    GemmP ep;
    load_epilogue_args(ep, p);
    epilogue<T, EOE_EPI_NONE, NI, 4>(ep, acc, m0 + wm0, 0, lane, smem + wave * 4096);
}

template <typename T, int GATHER>
int launch_nt64(const GemmP& p, hipStream_t s) {
    static bool once = (hipFuncSetAttribute((const void*)gemm_nt64_kernel<T, GATHER>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM64_BYTES), true);
    (void)once;
    hipLaunchKernelGGL((gemm_nt64_kernel<T, GATHER>), dim3(cdiv(p.M, 256)), dim3(256), SMEM64_BYTES, s, p);
    EOE_CHECK_LAUNCH("gemm_nt64");
    return 0;
}

// the implicit-convolution form (plain epilogue, optional BatchNorm statistics for MI = 4: 64 output rows per wave = the partial rows' unit)
template <typename T, int MI>
int launch_nt128_gather(const GemmP& p, hipStream_t s) {
    const int tiles = cdiv(p.M, 32 * MI) * cdiv(p.N, 128);
    static bool once = (hipFuncSetAttribute((const void*)gemm_nt128_kernel<T, EOE_EPI_NONE, MI, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem128_bytes(MI)), true);
    (void)once;
    hipLaunchKernelGGL((gemm_nt128_kernel<T, EOE_EPI_NONE, MI, 1>), dim3(tiles), dim3(256), smem128_bytes(MI), s, p);
    EOE_CHECK_LAUNCH("gemm_nt128 (gather)");
    return finish_colsum(p, EOE_EPI_NONE, MI, s);
}

template <typename T, int MI>
int launch_nt128(const GemmP& p, int epi, hipStream_t s) {
    const int tiles = cdiv(p.M, 32 * MI) * cdiv(p.N, 128);
#define EOE_NT128_CASE(E)                                                                   \
    case E:                                                                                 \
        { static bool once = (hipFuncSetAttribute((const void*)gemm_nt128_kernel<T, E, MI>, hipFuncAttributeMaxDynamicSharedMemorySize, smem128_bytes(MI)), true); (void)once; } \
        hipLaunchKernelGGL((gemm_nt128_kernel<T, E, MI>), dim3(tiles), dim3(256), smem128_bytes(MI), s, p); \
        break;
    switch (epi) {
        EOE_NT128_CASE(EOE_EPI_NONE)
        EOE_NT128_CASE(EOE_EPI_GELU)
        EOE_NT128_CASE(EOE_EPI_RESIDUAL)
        EOE_NT128_CASE(EOE_EPI_GELU_BWD)
        default: return eoe_set_error(EOE_ERR_ARG, "gemm_nt: unknown epilogue %d", epi);
    }
#undef EOE_NT128_CASE
    EOE_CHECK_LAUNCH("gemm_nt128");
    return finish_colsum(p, epi, MI, s);
}

template <typename T>
int launch_nt128w(const GemmP& p, int epi, hipStream_t s) {
    const int tiles = cdiv(p.M, 160) * cdiv(p.N, 256);
#define EOE_NT128W_CASE(E)                                                                  \
    case E:                                                                                 \
        { static bool once = (hipFuncSetAttribute((const void*)gemm_nt128w_kernel<T, E>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEMW_BYTES), true); (void)once; } \
        hipLaunchKernelGGL((gemm_nt128w_kernel<T, E>), dim3(tiles), dim3(256), SMEMW_BYTES, s, p); \
        break;
    switch (epi) {
        EOE_NT128W_CASE(EOE_EPI_NONE)
        EOE_NT128W_CASE(EOE_EPI_GELU)
        EOE_NT128W_CASE(EOE_EPI_RESIDUAL)
        EOE_NT128W_CASE(EOE_EPI_GELU_BWD)
        default: return eoe_set_error(EOE_ERR_ARG, "gemm_nt: unknown epilogue %d", epi);
    }
#undef EOE_NT128W_CASE
    EOE_CHECK_LAUNCH("gemm_nt128w");
    return finish_colsum(p, epi, 5, s);
}

// 128- or 160-row tiles: whichever needs less (rounds over the 2 x #CU workgroup slots) x (rows per tile)
template <typename T>
int launch_nt128_auto(const GemmP& p, int epi, hipStream_t s) {
    const int slots = 2 * num_cus();
    const long t4 = (long)cdiv(p.M, 128) * cdiv(p.N, 128), t5 = (long)cdiv(p.M, 160) * cdiv(p.N, 128);
    const long c4 = (t4 + slots - 1) / slots * 128, c5 = (t5 + slots - 1) / slots * 160;
    // (a tie goes to the 160-row tiles: N = 3072 at M = 12800 is 5 x 128 = 4 x 160 rounds x rows, and the in-step timing of
    //  tools/nt_shapes_sweep.sh has GELU' x dY at 1.226 against 1.260 ms per step with them -- fewer, larger tiles per slot)
    const bool five = (g_nt_flags & 32) || (!(g_nt_flags & 16) && c5 <= c4);
    return five ? launch_nt128<T, 5>(p, epi, s) : launch_nt128<T, 4>(p, epi, s);
}

// ---- split k for small M behind a long K, on request (eoe_gemm_args.split_k; round 3: the last ViT block on its class-token rows: 256 x 768 x
// 3072 is 12 tiles of 48 k-tiles each -- 50 us of one latency chain per CU on 12 CUs).  Not chosen by shape alone: a block's result must not
// depend on the batch size (test_full_batch_properties), and the split depends on K only.  S equal k-ranges per tile (grid.y), fp32 partial tiles in the caller's workspace, then one
// elementwise kernel that adds the S partials IN ORDER (reproducible) and applies alpha / bias / residual / the output type.
template <typename T>
__global__ __launch_bounds__(256) void nt_splitk_finish_kernel(const float* __restrict__ part, int S, int M, int N, float alpha,
                                                               const float* __restrict__ bias, const float* __restrict__ res, int ldres,
                                                               void* __restrict__ C, int ldc, int out_f32) {
    const int nq = N / 4;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * nq) return;
    const int m = i / nq, n = (i - m * nq) * 4;
    f32x4 v = *(const f32x4*)(part + (size_t)m * N + n);
    for (int s2 = 1; s2 < S; ++s2) v += *(const f32x4*)(part + ((size_t)s2 * M + m) * N + n);
    v *= alpha;
    if (bias) v += *(const f32x4*)(bias + n);
    if (res) v += *(const f32x4*)(res + (size_t)m * ldres + n);
    if (out_f32) *(f32x4*)((float*)C + (size_t)m * ldc + n) = v;
    else *(u32x2*)((T*)C + (size_t)m * ldc + n) = pack4<T>(v[0], v[1], v[2], v[3]);
}

// (round 5) the partial tiles live in the CALLER's workspace -- the slot area of eoe_gemm_args.sk_workspace -- and no longer in a library-owned
// buffer grown with hipMalloc / hipStreamSynchronize / hipFree: nothing the caller's allocator, a graph capture or a second process cannot see
// returns -1 when the split form does not apply (the caller then takes the plain launch)
template <typename T>
int launch_nt128_splitk(const GemmP& p, int epi, hipStream_t s) {
    const int nk = p.K / BK;
    if ((g_nt_flags & 8192) || !p.split_k || p.M > 1024 || nk < 24 || (epi != EOE_EPI_NONE && epi != EOE_EPI_RESIDUAL) || p.colsum || p.colsum_sq || p.accumulate ||
        (p.N & 15) || (p.ldc & 3) || (epi == EOE_EPI_RESIDUAL && (p.ldaux & 3)))
        return -1;
    int S = 8;
    while (S > 1 && (nk % S) != 0) S >>= 1;
    if (S < 2 || nk / S < 3) return -1;
    float* part = p.sk_part;                          // 2 x #CUs slots of 256 x 256 floats
    if (!part || (size_t)S * p.M * p.N > (size_t)2 * num_cus() * 65536) return -1;
    GemmP q = p;
    q.C = part; q.ldc = p.N; q.out_f32 = 1; q.bias = nullptr; q.aux = nullptr; q.alpha = 1.0f;
    const int tiles = cdiv(p.M, 128) * cdiv(p.N, 128);
    { static bool once = (hipFuncSetAttribute((const void*)gemm_nt128_kernel<T, EOE_EPI_NONE, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, smem128_bytes(4)), true); (void)once; }
    hipLaunchKernelGGL((gemm_nt128_kernel<T, EOE_EPI_NONE, 4>), dim3(tiles, S), dim3(256), smem128_bytes(4), s, q);
    EOE_CHECK_LAUNCH("gemm_nt128 (split k)");
    hipLaunchKernelGGL((nt_splitk_finish_kernel<T>), dim3(cdiv(p.M * (p.N / 4), 256)), dim3(256), 0, s, (const float*)part, S, p.M, p.N, p.alpha,
                       p.bias, epi == EOE_EPI_RESIDUAL ? (const float*)p.aux : nullptr, p.ldaux, p.C, p.ldc, p.out_f32);
    EOE_CHECK_LAUNCH("nt_splitk_finish");
    return 0;
}

// (round 2) Delaying the workgroup in the odd wave slot of each SIMD (HW_REG_HW_ID[3:0]) by half a k-tile at kernel entry, so that the
// two workgroups of a CU would issue their LDS-DMA / fragment reads next to each other's MFMA clusters: null (layer total 592 / 601 us
// against 605 us in the same process, within run-to-run spread).
// (round 2) Weaving the LDS-DMA pieces of the next k-tile between the MFMAs of the second half (unconditional pieces, out-of-range offset
// beyond the last k-tile, sched_group_barrier 2 MFMA : 1 piece : 1 fragment read) instead of issuing them in one burst behind the
// barrier -- what gave the one-wave kernel of gemm256.hip 19 %: null here (layer total 476.9 against 477.4 us interleaved in one
// process, bitwise identical results; in the step 11.70 against 11.65-11.79 ms).  With two workgroups per CU the burst of one is
// already covered by the MFMAs of the other.
// Measured on the two-workgroup kernel (tools/gemm128_timeline.py, s_memtime stamps + HW_ID per workgroup): the two workgroups of a CU
// run in LOCKSTEP for the whole launch (they start, finish and are replaced together), so the MFMA loops of a CU cover only 72 % of its
// time on the GELU GEMM (91 % on the plain K = 768 one).  De-phasing them (the second arrival of the first round sits out half a tile
// period, found through a per-CU atomic counter) raises the coverage to 94 % and makes every GEMM SLOWER (fc forward 100 -> 108 us,
// QKV 57 -> 61 us): a workgroup's MFMA loop is a latency chain that does not run faster alone, and the neighbour's epilogue takes
// issue slots from it.  Two loops side by side are the better packing; kept as is.  Inside the loop wave 0 spends < 2 % waiting for
// the LDS-DMA and 13-21 % at the per-k-tile barrier (skew between the four waves); s_setprio(1) around the MFMA clusters: null (+-1 %).
//
// Variants measured and rejected (interleaved A/B on one device with tools/gemm_ab.py, layer total of the 8 forward +
// dgrad GEMMs of a ViT-B/32 block at M = 12800; the kept kernel = 546 us stand-alone, 8.1 ms/step in the full step):
//   * sched_barrier pinning "8 fragment reads, then 16 MFMAs" per half iteration ........ 613 us (-11 %)
//   * BK = 32, five-stage ring (LDS-DMA 4 barrier intervals ahead), one barrier / 16 MFMAs  743 us, 10.6 ms/step
//   * BK = 32, two 8-wave workgroups per CU (no fragment double-buffering) ................ 772 us
//   * four waves of 128x64 (12 fragment reads / 32 MFMAs), BK = 32, two workgroups per CU .. 835 us
//   * per-XCD bands of whole tile rows swept column-major (A panels L2-resident) .......... 587 us, 10.6 ms/step
//   * non-persistent grid (one tile per workgroup, prologue exposed) ..................... within noise of persistent

int g_nt_flags = EOE_NT_DEFAULT_FLAGS;

template <typename T, int NI, int FLAGS>
int launch_nt_f(const GemmP& p, int epi, int grid, hipStream_t s) {
#define EOE_NT_CASE(E)                                                                      \
    case E:                                                                                 \
        { static bool once = (hipFuncSetAttribute((const void*)gemm_nt_kernel<T, E, NI, FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES), true); (void)once; } \
        hipLaunchKernelGGL((gemm_nt_kernel<T, E, NI, FLAGS>), dim3(grid), dim3(512), SMEM_BYTES, s, p); \
        break;
    switch (epi) {
        EOE_NT_CASE(EOE_EPI_NONE)
        EOE_NT_CASE(EOE_EPI_GELU)
        EOE_NT_CASE(EOE_EPI_RESIDUAL)
        EOE_NT_CASE(EOE_EPI_GELU_BWD)
        default: return eoe_set_error(EOE_ERR_ARG, "gemm_nt: unknown epilogue %d", epi);
    }
#undef EOE_NT_CASE
    EOE_CHECK_LAUNCH("gemm_nt");
    return finish_colsum(p, epi, 4, s);
}

// convolution variants (plain epilogue only): implicit patch matrix and / or the 256x64 tile for cout = 64
template <typename T, int NI, int GATHER>
int launch_nt_conv(const GemmP& p, int grid, hipStream_t s) {
    static bool once = (hipFuncSetAttribute((const void*)gemm_nt_kernel<T, EOE_EPI_NONE, NI, EOE_NT_DEFAULT_FLAGS, GATHER>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES), true);
    (void)once;
    hipLaunchKernelGGL((gemm_nt_kernel<T, EOE_EPI_NONE, NI, EOE_NT_DEFAULT_FLAGS, GATHER>), dim3(grid), dim3(512), SMEM_BYTES, s, p);
    EOE_CHECK_LAUNCH("gemm_nt");
    return finish_colsum(p, EOE_EPI_NONE, 4, s);
}

template <typename T, int NI>
int launch_nt_ni(const GemmP& p, int epi, int grid, hipStream_t s) {
#ifdef EOE_AB
    switch (g_nt_flags & 3) {
        case 0: return launch_nt_f<T, NI, 0>(p, epi, grid, s);
        case 1: return launch_nt_f<T, NI, 1>(p, epi, grid, s);
        case 2: return launch_nt_f<T, NI, 2>(p, epi, grid, s);
        default: return launch_nt_f<T, NI, 3>(p, epi, grid, s);
    }
#else
    return launch_nt_f<T, NI, EOE_NT_DEFAULT_FLAGS>(p, epi, grid, s);
#endif
}

template <typename T>
int launch_nt(const GemmP& p, int epi, int gather, hipStream_t s) {
    const int ncu = num_cus();
    if (epi == EOE_EPI_NONE && p.N <= 64 && (gather == 0 || gather == 1 || gather == 2) && !(g_nt_flags & 64)) {
        // cout <= 64: four 64x64 waves per 256x64 tile, two workgroups per CU (nt_flags bit 6: the persistent 256x64 kernel instead)
        const int rc = gather == 2 ? launch_nt64<T, 2>(p, s) : (gather ? launch_nt64<T, 1>(p, s) : launch_nt64<T, 0>(p, s));
        return rc ? rc : finish_colsum(p, EOE_EPI_NONE, 4, s);
    }
    if (gather == 1 && epi == EOE_EPI_NONE && p.N >= 128 && !(g_nt_flags & 2048)) {
        // cout >= 128, one tap per k-tile: the two-workgroup kernel (nt_flags bit 11 = 2048: the persistent 256x128 kernel, A/B).  160-row
        // tiles where they need fewer (rounds x rows) -- not with BatchNorm statistics in the epilogue, whose partial rows are per 64 rows
        const int slots = 2 * ncu;
        const long t4 = (long)cdiv(p.M, 128) * cdiv(p.N, 128), t5 = (long)cdiv(p.M, 160) * cdiv(p.N, 128);
        const long c4 = (t4 + slots - 1) / slots * 128, c5 = (t5 + slots - 1) / slots * 160;
        if (!p.colsum_sq && c5 < c4) return launch_nt128_gather<T, 5>(p, s);
        return launch_nt128_gather<T, 4>(p, s);
    }
    if (gather || (epi == EOE_EPI_NONE && p.N <= 64)) {
        const bool narrow = p.N <= 64;               // 256x64 tiles: no MFMA / LDS work on columns that do not exist
        const int tiles = cdiv(p.M, BM) * cdiv(p.N, narrow ? 64 : 128);
        const int grid = tiles < ncu ? tiles : ncu;
        if (gather == 2) return narrow ? launch_nt_conv<T, 2, 2>(p, grid, s) : launch_nt_conv<T, 4, 2>(p, grid, s);
        if (gather == 3) return narrow ? launch_nt_conv<T, 2, 3>(p, grid, s) : launch_nt_conv<T, 4, 3>(p, grid, s);
        if (gather) return narrow ? launch_nt_conv<T, 2, 1>(p, grid, s) : launch_nt_conv<T, 4, 1>(p, grid, s);
        return launch_nt_conv<T, 2, 0>(p, grid, s);
    }
    // Tile shape (tools/gemm_bench.py, M = 12800): the two-workgroup kernel (128- or 160-row tiles, launch_nt128_auto) wins on
    // every ViT shape -- most where the epilogue is heavy against a short K loop (GELU' x dY 116 -> 94 us, fp32 residual
    // 42 -> 32 us, GELU forward 107 -> 96 us once its epilogue math was cheap) and, with 160-row tiles, on the N = 768 shapes
    // whose 600 128x128 tiles need two rounds over the 512 workgroup slots (480 tiles of 160x128: one round; K = 3072:
    // 83 -> 68 us = 887 TF, K = 2304: 61 -> 50 us = 916 TF).  The persistent 256-row kernel keeps large square problems
    // (4096^3: 1046 vs 984 TF).  nt_flags: bit 3 forces the two-workgroup kernel, bit 2 forbids it, bit 4 / 5 force 128 / 160 rows.
    // the one-wave-per-SIMD kernel (gemm256.hip) has only the LDS-transposed epilogue compiled in
    if ((g_nt_flags & (128 | 256)) && epilogue_direct_ok(p, epi))
        return eoe_launch_nt256(&p, std::is_same<T, f16_t>::value ? EOE_F16 : EOE_BF16, epi, (g_nt_flags & 256) ? 8 : 5, s);
    const bool big = p.K >= 4096 && p.N >= 2048;
    // The one-wave-per-SIMD kernel (gemm256.hip; 160x256 or 256x256 tiles by the cost model below) for every large NT GEMM: OPT-IN
    // (nt_flags bit 9 = 512).  Stand-alone at M = 12800 (tools/gemm_bench.py, the same GEMM back to back, operands hot in the
    // Infinity Cache) it wins on most shapes -- N=768 K=3072 59.8 us against 67-69 (1010 TF), + residual 74 / 81-91, N=3072 K=768
    // GELU 91-93 / 99-103, N=2304 K=768 54.3 / 57.5, patch embedding 67.5 / 72-81, 4096^3 1312 TF / 1044 TF -- but INSIDE the training
    // step, where every GEMM meets its weights cold (340 MB of 16-bit copies per step do not stay in the 256 MB cache), it loses on
    // seven of the eight shapes (per-shape hipEvents in one process on one box, EOE_PROF_SHAPES=1: qkv 63.5 / 61.4 us, dX of fc
    // 64.2 / 61.9, c_proj 86.5 / 81.7, GELU' x dY 114 / 103; only c_fc forward 102 / 105.6 wins; step 12.35 against 12.13 ms): one
    // wave per SIMD behind a 2-k-tile LDS-DMA lead cannot hide an HBM miss the way two independent workgroups per CU do.
    // Large square problems (below) keep it: there the 256x256 tile's halved operand traffic decides.
    // Per-shape in-step timing (round 3, tools/nt_shapes_ab.sh: EOE_PROF_SHAPES=1, both kernels interleaved on one box, ms per step over the 12
    // layers): the one-wave kernel wins where the epilogue is the GELU pair of outputs behind a short K loop -- c_fc forward 1.167 against
    // 1.260 --, ties on the K >= 2304 shapes (-0.001 ... -0.010) and loses on QKV (+0.135), GELU' x dY (+0.057) and the two K = 768 -> 768
    // shapes (+0.056, +0.020).  So: c_fc forward only (nt_flags bit 10 = 1024 turns the rule off).
    if (gather == 0 && p.split_k) {          // asked for: small M behind a long K (nt_flags bit 13 = 8192 switches it off)
        const int rc = launch_nt128_splitk<T>(p, epi, s);
        if (rc >= 0) return rc;
    }
    // the eight-wave 256 x 256 kernel (gemm_w8.hip; round 4) for the plain / GELU 16-bit shapes whose 256 x 256 tiles fill the CUs to >= 85 %
    // in every round (M = 12 800: N = 2304 -> 450 tiles = 2 rounds at 88 %: the in-projection, 55 against 65 us, same bits; N = 3072 -> 600
    // = 3 rounds at 78 %: stays on the 160 x 256 x 32 kernel below).  nt_flags bit 17 = 131072 switches it off, bit 18 = 262144 forces it
    if (!(g_nt_flags & (131072 | 4 | 8 | 512)) && gather == 0 && p.M >= 2048 && eoe_w8_applies(&p, epi)) {
        const long tiles = (long)cdiv(p.M, 256) * (p.N / 256);
        // (round 5) opt-in, nt_flags bit 20 = 1048576: with the caller's stream-K workspace the fractional last round is cut along k over all
        // CUs (bit 21: the last full round joins the stream-K part; bit 22: stream-K part first).  Measured slower than the rules below on every
        // ViT shape (gemm_w8.hip, w8_sk_rounds)
        if ((g_nt_flags & 262144) || eoe_w8_streamk(&p) || tiles * 100 >= ((tiles + ncu - 1) / ncu) * ncu * 85) 
            return eoe_launch_w8(&p, std::is_same<T, f16_t>::value ? EOE_F16 : EOE_BF16, epi, s);
    }
    // the 160x256x32 two-workgroup kernel on the wide-N shapes (c_fc forward, GELU' x dY: 10.64 -> 10.54 ms per step, three interleaved
    // pairs, same bits; nt_flags bit 12 = 4096 switches it off)
    if (!(g_nt_flags & (4096 | 4 | 8 | 512)) && gather == 0 && !p.colsum_sq && (p.N % 256) == 0 && p.N >= 2048 && (p.K % 32) == 0 && p.M >= 2048) {
        // only where its tiles fill the 2 x #CU workgroup slots to >= 90 % in every round (N = 3072 at M = 12 800: 960 of 1024; N = 2304: 720)
        const long tiles = (long)cdiv(p.M, 160) * (p.N / 256), slots = 2L * ncu;
        if (tiles * 10 >= ((tiles + slots - 1) / slots) * slots * 9) return launch_nt128w<T>(p, epi, s);
    }
    const bool gelu_wide = epi == EOE_EPI_GELU && p.N >= 2048 && p.K <= 1024 && !(g_nt_flags & 1024);
    if (((g_nt_flags & 512) || gelu_wide) && !(g_nt_flags & (4 | 8)) && epilogue_direct_ok(p, epi) && p.M >= 2048 && p.N >= 512) {
        // time of a launch in units of one 16-row x 256-column x 64-deep slab of MFMAs: rounds over the CUs x (rows per tile x
        // (k-tiles + 2 for the pipeline fill) + 30 for a tile's epilogue and hand-over)
        const int nk = p.K / BK;
        long best = 0;
        int mi_best = 5;
        for (int mi : {5, 8}) {
            const long tiles = (long)cdiv(p.M, 32 * mi) * cdiv(p.N, 256);
            const long cost = ((tiles + ncu - 1) / ncu) * ((long)mi * (nk + 2) + 30);
            if (!best || cost < best) { best = cost; mi_best = mi; }
        }
        return eoe_launch_nt256(&p, std::is_same<T, f16_t>::value ? EOE_F16 : EOE_BF16, epi, mi_best, s);
    }
    if (big && !(g_nt_flags & (4 | 8)) && epilogue_direct_ok(p, epi) && p.M >= 2048)
        return eoe_launch_nt256(&p, std::is_same<T, f16_t>::value ? EOE_F16 : EOE_BF16, epi, 8, s);
    if (!p.colsum_sq && ((g_nt_flags & 8) || (!big && !(g_nt_flags & 4)))) return launch_nt128_auto<T>(p, epi, s);
    // tile width: 256x128 unless 256x96 needs >= 10 % fewer (rounds x width) units over the CUs
    const int t4 = cdiv(p.M, BM) * cdiv(p.N, 128), t3 = cdiv(p.M, BM) * cdiv(p.N, 96);
    const int c4 = cdiv(t4, ncu) * 4, c3 = cdiv(t3, ncu) * 3;
    const int force = (g_nt_flags >> 4) & 7;          // option bits 4-6: 0 = heuristic, 3 / 4 = force the tile width
    const bool use3 = force ? (force == 3) : (c3 * 10 <= c4 * 9);
    const int tiles = use3 ? t3 : t4;
    const int grid = tiles < ncu ? tiles : ncu;      // persistent: one 8-wave workgroup per CU
    return use3 ? launch_nt_ni<T, 3>(p, epi, grid, s) : launch_nt_ni<T, 4>(p, epi, grid, s);
}

int fill_params(const eoe_gemm_args* a, GemmP& p) {
    EOE_CHECK_ARG(a != nullptr, "gemm: null args");
    EOE_CHECK_ARG(a->A && a->B && a->C, "gemm: null operand");
    EOE_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0, "gemm: bad shape %d %d %d", a->M, a->N, a->K);
    EOE_CHECK_ARG(a->dtype == EOE_F16 || a->dtype == EOE_BF16, "gemm: bad dtype %d", a->dtype);
    EOE_CHECK_ARG((a->lda % 8) == 0 && (a->ldb % 8) == 0, "gemm: lda/ldb must be multiples of 8 (16-B rows)");
    EOE_CHECK_ARG((((uintptr_t)a->A | (uintptr_t)a->B) & 15) == 0, "gemm: A/B must be 16-B aligned");
    EOE_CHECK_ARG(!a->accumulate || a->out_f32, "gemm: accumulate needs an fp32 C");
    p.A = a->A; p.B = a->B; p.C = a->C; p.bias = a->bias; p.aux = a->aux; p.aux_out = a->aux_out; p.colsum = a->colsum;
    p.colsum_part = nullptr; p.colsum_sq = 0; p.colsum_blocked = 0;
    p.M = a->M; p.N = a->N; p.K = a->K; p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc; p.ldaux = a->ldaux;
    p.out_f32 = a->out_f32; p.accumulate = a->accumulate; p.alpha = a->alpha; p.split_k = a->split_k;
    // the stream-K workspace (gemm_w8.hip): 8 KiB of ticket / flag words, then the partial-accumulator slots
    p.sk_part = nullptr; p.sk_sync = nullptr; p.sk_rounds = 0; p.sk_flags = 0;
    if (a->sk_workspace && a->sk_workspace_bytes >= (int64_t)EOE_NT_STREAMK_WORKSPACE_BYTES(num_cus()) && (((uintptr_t)a->sk_workspace) & 15) == 0) {
        p.sk_sync = (int*)a->sk_workspace;
        p.sk_part = (float*)((char*)a->sk_workspace + 8192);
    }
    EOE_CHECK_ARG((a->K % BK) == 0, "gemm_nt: K=%d must be a multiple of %d", a->K, BK);
    EOE_CHECK_ARG(a->ldb >= a->K, "gemm_nt: leading dims smaller than K");
    size_t ba = ((size_t)(a->M - 1) * a->lda + a->K) * 2;
    const size_t bb = ((size_t)(a->N - 1) * a->ldb + a->K) * 2;
    p.gH = p.gW = p.gC = p.gWo = p.gHoWo = p.gkw = p.gstride = p.gpad = p.gkstep = p.gkh = p.glog2c = 0;
    p.gmagic = 0;
    if (a->gather == 2) {
        const eoe_conv_geometry& g = a->geo;
        EOE_CHECK_ARG(a->epilogue == EOE_EPI_NONE, "gemm_nt: an implicit patch matrix supports the plain epilogue only");
        EOE_CHECK_ARG(g.n > 0 && g.C == 4 && g.kh > 0 && g.kw > 0 && g.kw <= 8 && g.stride > 0 && g.stride % 2 == 0 && g.pad == 0 && g.Ho > 0 &&
                      g.Wo > 0 && g.W % 2 == 0, "gemm_nt: bad packed first-layer geometry");
        EOE_CHECK_ARG((g.Ho - 1) * g.stride + g.kh <= g.H && (g.Wo - 1) * g.stride + 8 <= g.W && a->M == g.n * g.Ho * g.Wo &&
                      a->K == (g.kh + 1) / 2 * BK, "gemm_nt: packed geometry does not match M = %d, K = %d (image must be padded)", a->M, a->K);
        ba = (size_t)g.n * g.H * g.W * 4 * 2;
        p.gH = g.H; p.gW = g.W; p.gC = 4; p.gWo = g.Wo; p.gHoWo = g.Ho * g.Wo; p.gkw = g.kw; p.gstride = g.stride;
        p.gkstep = 2 * g.W * 8;
    } else if (a->gather) {
        const eoe_conv_geometry& g = a->geo;
        EOE_CHECK_ARG(a->epilogue == EOE_EPI_NONE, "gemm_nt: an implicit patch matrix supports the plain epilogue only");
        const bool narrowc = g.C == 8 || g.C == 16 || g.C == 32;
        EOE_CHECK_ARG(g.n > 0 && g.H > 0 && g.W > 0 && g.C > 0 && (g.C % BK == 0 || narrowc) && g.kh > 0 && g.kw > 0 && g.stride > 0 &&
                      g.pad >= 0 && g.Ho > 0 && g.Wo > 0, "gemm_nt: bad conv geometry (C = %d must be 8, 16, 32 or a multiple of %d)", g.C, BK);
        EOE_CHECK_ARG(a->M == g.n * g.Ho * g.Wo && (narrowc ? a->K >= g.kh * g.kw * g.C : a->K == g.kh * g.kw * g.C),
                      "gemm_nt: conv geometry does not match M = %d, K = %d", a->M, a->K);
        if (narrowc) {
            p.gkh = g.kh;
            p.glog2c = g.C == 8 ? 3 : (g.C == 16 ? 4 : 5);
            p.gmagic = (65536u + (unsigned)g.kw - 1) / (unsigned)g.kw;
            EOE_CHECK_ARG((a->K >> p.glog2c) < 4096, "gemm_nt: too many taps");
        }
        ba = (size_t)g.n * g.H * g.W * g.C * 2;
        p.gH = g.H; p.gW = g.W; p.gC = g.C; p.gWo = g.Wo; p.gHoWo = g.Ho * g.Wo; p.gkw = g.kw; p.gstride = g.stride; p.gpad = g.pad;
    } else {
        EOE_CHECK_ARG(a->lda >= a->K, "gemm_nt: leading dims smaller than K");
    }
    EOE_CHECK_ARG(ba < 0x7fffffffull && bb < 0x7fffffffull, "gemm: operand larger than 2 GiB");
    p.bytesA = (unsigned)ba; p.bytesB = (unsigned)bb;
    // diagnostics only: EOE_GEMM_DEBUG=1 makes every operand load out of range (zero-filled, nothing fetched), which
    // times the LDS/MFMA/epilogue side of the kernel alone (results are then wrong by construction)
    static const int dbg = getenv("EOE_GEMM_DEBUG") ? atoi(getenv("EOE_GEMM_DEBUG")) : 0;
    if (dbg & 1) { p.bytesA = 0; p.bytesB = 0; }
    p.dbg = dbg | ((g_nt_flags & 524288) ? 32 : 0) | ((g_nt_flags & 8388608) ? 64 : 0);       // nt_flags bit 19: xcd_halves off; bit 23: nt128w's old LDS image (A/B)
    p.stamp = nullptr;
    static const int stampon = getenv("EOE_GEMM_STAMP") ? atoi(getenv("EOE_GEMM_STAMP")) : 0;
    if (stampon) {
        static unsigned long long* buf = [] { void* b = nullptr; (void)hipMalloc(&b, 4096 * 16 * 8); return (unsigned long long*)b; }();
        (void)hipMemsetAsync(buf, 0, 4096 * 16 * 8, 0);
        p.stamp = buf;
        g_stamp_buf = buf;
    }
    if (a->epilogue == EOE_EPI_GELU) EOE_CHECK_ARG(!a->out_f32, "gemm: the GELU epilogue needs a 16-bit C");      // aux_out optional
    if (a->epilogue == EOE_EPI_RESIDUAL) EOE_CHECK_ARG(a->aux && a->out_f32, "gemm: RESIDUAL epilogue needs aux, fp32 C");
    if (a->epilogue == EOE_EPI_GELU_BWD) EOE_CHECK_ARG(a->aux, "gemm: GELU_BWD epilogue needs aux");
    return 0;
}

}  // namespace

int eoe_nt_flags() { return g_nt_flags; }          // for gemm_w8.hip's launcher

extern "C" int eoe_gemm_nt(const eoe_gemm_args* a, void* stream) {
    GemmP p;
    EOE_TRY(fill_params(a, p));
    const int osz = a->out_f32 ? 4 : 2;
    // diagnostics (EOE_PROF_SHAPES=1): one profile row per (shape, epilogue) instead of one for all NT GEMMs
    static const bool by_shape = getenv("EOE_PROF_SHAPES") != nullptr;
    char pname[32] = "gemm_nt";
    if (by_shape) snprintf(pname, sizeof(pname), "nt_%dx%dx%d_e%d%s", a->M, a->N, a->K, a->epilogue, a->colsum ? "c" : "");
    ProfScope ps(pname, 2.0 * a->M * a->N * a->K,
                 2.0 * ((a->gather ? (double)a->geo.n * a->geo.H * a->geo.W * a->geo.C : (double)a->M * a->K) + (double)a->N * a->K) +
                     (double)osz * a->M * a->N *
                     (a->epilogue == EOE_EPI_GELU ? 2 : 1) + (a->epilogue == EOE_EPI_RESIDUAL ? 4.0 * a->M * a->N : 0.0) +
                     (a->epilogue == EOE_EPI_GELU_BWD ? 2.0 * a->M * a->N : 0.0), stream);
    const int mode = (a->gather == 1 && p.gkh) ? 3 : a->gather;      // narrow-channel geometry -> per-piece tap decoding
    // fused column sums: through partial rows in the workspace (one row per 64 / 80 output rows) when it is large enough
    if (a->colsum && a->workspace && a->workspace_bytes >= (int64_t)EOE_NT_COLSUM_WORKSPACE_BYTES(a->M, a->N))
        p.colsum_part = (float*)a->workspace, p.colsum_blocked = ((a->N & 63) == 0 && (a->N & 3) == 0) ? 1 : 0;
    if (a->colstats) {
        // per-64-row partial sums and sums of squares of the fp32 result (BatchNorm batch statistics without a pass over C)
        EOE_CHECK_ARG(a->epilogue != EOE_EPI_GELU && !a->colsum, "gemm_nt: colstats goes with neither the GELU epilogue nor colsum");
        EOE_CHECK_ARG(a->workspace && a->workspace_bytes >= 2 * (int64_t)EOE_NT_COLSUM_WORKSPACE_BYTES(a->M, a->N),
                      "gemm_nt: colstats needs a workspace of 2 * EOE_NT_COLSUM_WORKSPACE_BYTES(M, N)");
        EOE_CHECK_ARG((a->N & 15) == 0 && (a->ldc & 7) == 0 && (a->ldaux & 7) == 0 && (((uintptr_t)a->C) & 15) == 0,
                      "gemm_nt: colstats needs N %% 16 == 0 and 16-byte aligned rows of C");
        p.colsum_part = (float*)a->workspace;
        p.colsum_sq = 1;
        p.colsum_blocked = 0;
    }
    return a->dtype == EOE_F16 ? launch_nt<f16_t>(p, a->epilogue, mode, (hipStream_t)stream)
                               : launch_nt<bf16_t>(p, a->epilogue, mode, (hipStream_t)stream);
}


// diagnostics: copies the per-workgroup s_memtime stamps of the last gemm_nt launch (EOE_GEMM_STAMP=1) to the host
extern "C" int eoe_debug_gemm_stamps(unsigned long long* out, int n_words) {
    if (!g_stamp_buf || !out) return eoe_set_error(EOE_ERR_ARG, "gemm stamps are not enabled (EOE_GEMM_STAMP=1)");
    if (hipDeviceSynchronize() != hipSuccess) return eoe_set_error(EOE_ERR_LAUNCH, "sync failed");
    if (hipMemcpy(out, g_stamp_buf, (size_t)n_words * 8, hipMemcpyDeviceToHost) != hipSuccess)
        return eoe_set_error(EOE_ERR_LAUNCH, "copy failed");
    return 0;
}

// tuning switches (see gemm_nt_kernel FLAGS; bits 4-6 force the tile width); all four variants exist only in -DEOE_AB builds
extern "C" int eoe_set_option(const char* name, int value) {
    if (name && !strcmp(name, "nt_flags")) { g_nt_flags = value; return 0; }
    if (name && !strcmp(name, "tn_flags")) { g_tn_flags = value; return 0; }
    if (name && !strcmp(name, "vit_side_stream")) { g_vit_side_stream = value; return 0; }
    if (name && !strcmp(name, "attn_flags")) { g_attn_flags = value; return 0; }
    if (name && !strcmp(name, "parity_flags")) { g_parity_flags = value; return 0; }
    return eoe_set_error(EOE_ERR_ARG, "unknown option");
}

extern "C" int eoe_get_option(const char* name, int* value) {
    EOE_CHECK_ARG(name && value, "eoe_get_option: null pointer");
    if (!strcmp(name, "nt_flags")) { *value = g_nt_flags; return 0; }
    if (!strcmp(name, "tn_flags")) { *value = g_tn_flags; return 0; }
    if (!strcmp(name, "vit_side_stream")) { *value = g_vit_side_stream; return 0; }
    if (!strcmp(name, "attn_flags")) { *value = g_attn_flags; return 0; }
    if (!strcmp(name, "tn256_launches")) { *value = g_tn256_launches; return 0; }       // read-only diagnostics
    if (!strcmp(name, "parity_flags")) { *value = g_parity_flags; return 0; }
    return eoe_set_error(EOE_ERR_ARG, "unknown option");
}
