// Multi-head self-attention for short sequences (L <= 64 tokens, head dim 64; ViT-B/32: L = 50, 12 heads),
// one wavefront per (image, head), everything in registers + one small LDS image per operand that needs a
// transposed (k-strided) read.  v_mfma_f32_16x16x32_{f16,bf16}; scores are computed TRANSPOSED
// (S^T = K Q^T: key on the accumulator row = register index, query on the lane), so
//   * the softmax row reduction is 16 in-lane values + two cross-lane steps (xor 16, xor 32),
//   * the probabilities are already the B operand of the next product (O^T = V^T P^T sums over the key =
//     accumulator-row index: no lane movement, no LDS) -- with the k order inside a 32-key step permuted to
//     (j<4: key 32s+4g+j, j>=4: key 32s+16+4g+j-4), which the other operand reproduces by choosing which 4-row
//     blocks its ds_read_b64_tr_b16 transposed reads fetch.
// Backward recomputes the probabilities (L*L is tiny) in both orientations: lane = query for dQ, and
// lane = key (operands swapped) for dV and dK, so that every product sums over an accumulator-row index.
#include "common.h"

int g_attn_flags = 0;            // eoe_set_option("attn_flags", bit 0: the one-wave backward kernel)

namespace {

constexpr int ROWB = 160;            // LDS row pitch in bytes (64 x 16-bit + pad): conflict-free transposed reads
constexpr int TILEB = 64 * ROWB;     // one 64 x 64 operand image
// (round 5) The 16-byte chunks of ODD rows are stored pairwise swapped (in-row byte offset ^ 16).  With the hardware's lane groups (a
// ds_read_b64_tr_b16 is served in two groups of 32 lanes, banks (a / 4) mod 64) the interleaved transposing read tfrag_il -- dword 40 row + 8 p +
// 2 tc over rows 0-7 x p 0-3 -- put 32 lanes on 16 banks FOUR ways (8 LDS cycles for an ideal 2; attn_bwd4: SQ_LDS_BANK_CONFLICT 3.56 M cycles
// per launch); with the swap two ways, the floor for 8-byte pieces of 16-byte-aligned rows (searched by simulation over every GF(2)-linear
// 3-bit function of the row, pitches 128 / 160 / 192); the ds_read_b128 fragment reads and the plain transposing reads stay conflict-free.
// attn_bwd 0.480 -> 0.465 ms per step, step 10.256 -> 10.222 (three interleaved pairs, tools/cflags_ab.sh -DEOE_ATTN_NO_RSWZ).  Also measured: an
// unpadded 128-byte pitch with the 3-bit term ((row >> 1 & 1) * 3 | row & 4) -- the same conflict profile, 28 KB of LDS per attn_bwd4 workgroup --
// and five workgroups per CU (amdgpu_waves_per_eu(5, 5): 96 registers, 4 spilled): attn_bwd 0.465 -> 0.509 ms per step; not kept.
#ifdef EOE_ATTN_NO_RSWZ          // A/B builds (tools/cflags_ab.sh): the round-4 image
__device__ __forceinline__ int rswz(int) { return 0; }
#else
__device__ __forceinline__ int rswz(int row) { return (row & 1) << 4; }
#endif

template <typename T> using V8 = typename T16<T>::v8;

template <typename T>
__device__ __forceinline__ V8<T> zero8() {
    i16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_bit_cast(V8<T>, z);
}

// stage rows [0, L) of a 64-wide head slice into an LDS image, zero rows [L, 64)
template <typename T>
__device__ __forceinline__ void stage_tile(char* lds, const T* src, int ld, int L, int lane) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + (lane >> 3), ch = lane & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < L) v = *(const u32x4*)(src + (size_t)row * ld + ch * 8);
        *(u32x4*)(lds + row * ROWB + ((ch * 16) ^ rswz(row))) = v;
    }
}

// row-major fragment (operand element j <-> column 8*(lane>>4)+j of k-step ks) straight from global memory
template <typename T>
__device__ __forceinline__ V8<T> gfrag(const T* src, int ld, int L, int t, int ks, int lane) {
    const int row = t * 16 + (lane & 15);
    if (row >= L) return zero8<T>();
    return __builtin_bit_cast(V8<T>, *(const u32x4*)(src + (size_t)row * ld + (ks * 4 + (lane >> 4)) * 8));
}
// The loads above sit behind `row < L` branches, and hipcc turns every one of them into load -> s_waitcnt vmcnt(0) -> use: a wave's
// operand fetch became a chain of 8-11 dependent HBM round trips (round 4: attn_fwd's ISA showed eight load/wait/ds_write groups for V alone).
// The forms below read through a buffer resource that covers the (image, head) slab; a row >= L gets an out-of-range offset, which reads
// zeros, so the loads are unconditional and a whole batch is in flight before the first wait.
__device__ __forceinline__ u32x4 bload16(__amdgpu_buffer_rsrc_t r, int row, int L, unsigned pitchb, unsigned cb) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(row < L ? (unsigned)row * pitchb + cb : EOE_OOB), 0, 0));
}
// gfrag through the resource: cb0 = byte offset of the operand's head slice inside a row
template <typename T>
__device__ __forceinline__ V8<T> bfrag(__amdgpu_buffer_rsrc_t r, unsigned pitchb, unsigned cb0, int L, int t, int ks, int lane) {
    return __builtin_bit_cast(V8<T>, bload16(r, t * 16 + (lane & 15), L, pitchb, cb0 + (unsigned)(ks * 4 + (lane >> 4)) * 16u));
}
// the same fragment from an LDS image
template <typename T>
__device__ __forceinline__ V8<T> lfrag(const char* lds, int t, int ks, int lane) {
    const int row = t * 16 + (lane & 15);
    return __builtin_bit_cast(V8<T>, *(const u32x4*)(lds + row * ROWB + (((ks * 4 + (lane >> 4)) * 16) ^ rswz(row))));
}
// transposed fragment: lane <-> column 16*tc + (lane&15); element j <-> row perm(s, lane>>4, j) (see header)
template <typename T>
__device__ __forceinline__ V8<T> tfrag(const char* lds, int tc, int s, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const char* a = lds + (32 * s + 4 * g + q) * ROWB + (((16 * tc + 4 * p) * 2) ^ rswz(4 * g + q));
    i16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4v*)(a));
    i16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4v*)(a + 16 * ROWB));
    i16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(V8<T>, r);
}
// the same with the COLUMNS of the four tiles tc = 0..3 interleaved: lane <-> column 16*((lane&15)>>2) + 4*tc + (lane&3).  As the A operand of a
// transposed product (rows of the result = these columns) it leaves lane (lane&15, g = lane>>4) of result tile tc with rows
// 16*g + 4*tc + 0..3 -- over the four tiles 16 CONSECUTIVE rows per lane: a lane's share of an output row is one 32-byte run
// (attn_bwd4_kernel's dQ / dK / dV stores) instead of four 8-byte pieces 32 bytes apart.  Only the address each lane hands the transposing read
// changes.
template <typename T>
__device__ __forceinline__ V8<T> tfrag_il(const char* lds, int tc, int s, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const char* a = lds + (32 * s + 4 * g + q) * ROWB + (((16 * p + 4 * tc) * 2) ^ rswz(4 * g + q));
    i16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4v*)(a));
    i16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4v*)(a + 16 * ROWB));
    i16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(V8<T>, r);
}
// two accumulator tiles (rows 32s+0..15 and 32s+16..31 of a transposed product) as the next B operand
template <typename T>
__device__ __forceinline__ V8<T> acc_as_operand(const f32x4& lo, const f32x4& hi) {
    V8<T> r;
    r[0] = (T)lo[0]; r[1] = (T)lo[1]; r[2] = (T)lo[2]; r[3] = (T)lo[3];
    r[4] = (T)hi[0]; r[5] = (T)hi[1]; r[6] = (T)hi[2]; r[7] = (T)hi[3];
    return r;
}

__device__ __forceinline__ float group_max(float v) {   // over the 4 lanes sharing lane&15
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// transposed scores -> normalised probabilities in place; returns per-tq (max, sum) through mx/sm
template <typename T>
__device__ __forceinline__ void softmax_T(f32x4 (&s)[4][4], int L, float scale, int lane, float (&mx)[4],
                                          float (&sm)[4]) {
    const int lg = lane >> 4;
#pragma unroll
    for (int tq = 0; tq < 4; ++tq) {
        float m = -INFINITY;
#pragma unroll
        for (int tk = 0; tk < 4; ++tk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * tk + 4 * lg + r;
                const float v = (key < L) ? s[tk][tq][r] * scale : -INFINITY;
                s[tk][tq][r] = v;
                m = fmaxf(m, v);
            }
        m = group_max(m);
        float sum = 0.f;
#pragma unroll
        for (int tk = 0; tk < 4; ++tk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(s[tk][tq][r] - m);
                s[tk][tq][r] = e;
                sum += e;
            }
        sum = group_sum(sum);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int tk = 0; tk < 4; ++tk)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[tk][tq][r] *= inv;
        mx[tq] = m;
        sm[tq] = sum;
    }
}

// ------------------------------------------------------------------------------------------------ forward
template <typename T>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out, int L,
                                                      int heads, float scale) {
    __shared__ __attribute__((aligned(16))) char vs[TILEB];
    const int lane = threadIdx.x;
    const int img = blockIdx.x / heads, h = blockIdx.x % heads;
    const int D = heads * 64, ld = 3 * D;
    const T* qp = qkv + (size_t)img * L * ld + h * 64;
    const T* kp = qp + D;
    const T* vp = qp + 2 * D;

    // one resource over this (image, head)'s Q | K | V slices: rows of 3 D elements, the last one ending behind V's 64 columns
    const unsigned pitchb = (unsigned)ld * 2u, kcb = (unsigned)D * 2u, vcb = (unsigned)D * 4u;
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(qp, ((unsigned)(L - 1) * (unsigned)ld + 2u * D + 64u) * 2u);
    (void)kp; (void)vp;
    // all 24 loads of the wave back to back: V (for the LDS image), then the K / Q fragments of both k-steps
    u32x4 vv[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) vv[it] = bload16(rq, it * 8 + (lane >> 3), L, pitchb, vcb + (unsigned)(lane & 7) * 16u);
    V8<T> kf[2][4], qf[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            kf[ks][t] = bfrag<T>(rq, pitchb, kcb, L, t, ks, lane);
            qf[ks][t] = bfrag<T>(rq, pitchb, 0u, L, t, ks, lane);
        }
    __builtin_amdgcn_sched_barrier(0);          // (left alone the scheduler weaves the first MFMAs, and their waits, in among the loads)
#pragma unroll
    for (int it = 0; it < 8; ++it) *(u32x4*)(vs + (it * 8 + (lane >> 3)) * ROWB + (((lane & 7) * 16) ^ rswz(it * 8 + (lane >> 3)))) = vv[it];
    __builtin_amdgcn_sched_barrier(0);

    f32x4 s[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int tk = 0; tk < 4; ++tk)
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) s[tk][tq] = T16<T>::mfma16(kf[ks][tk], qf[ks][tq], s[tk][tq]);
    float mx[4], sm[4];
    softmax_T<T>(s, L, scale, lane, mx, sm);

    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // O^T tile td, register r of lane (lr, lg) = d 16 lg + 4 td + r (tfrag_il): over the four tiles a lane holds 16 consecutive d of its
    // query's row -- two 16-byte stores, and the four lanes of a query write its whole 128-byte head slice (round 4; before: sixteen 8-byte
    // pieces per lane).  Same products in the same order.
    const int lr = lane & 15, lg = lane >> 4;
    u32x2 pk[4][4];
#pragma unroll
    for (int td = 0; td < 4; ++td) {
        f32x4 o[4];
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) o[tq] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const V8<T> vf = tfrag_il<T>(vs, td, st, lane);
#pragma unroll
            for (int tq = 0; tq < 4; ++tq)
                o[tq] = T16<T>::mfma16(vf, acc_as_operand<T>(s[2 * st][tq], s[2 * st + 1][tq]), o[tq]);
        }
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) pk[tq][td] = pack4<T>(o[tq][0], o[tq][1], o[tq][2], o[tq][3]);
    }
#pragma unroll
    for (int tq = 0; tq < 4; ++tq) {
        const int query = 16 * tq + lr;
        if (query < L) {
            T* d = out + ((size_t)img * L + query) * D + h * 64 + 16 * lg;
            *(u32x4*)d = (u32x4){pk[tq][0][0], pk[tq][0][1], pk[tq][1][0], pk[tq][1][1]};
            *(u32x4*)(d + 8) = (u32x4){pk[tq][2][0], pk[tq][2][1], pk[tq][3][0], pk[tq][3][1]};
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
// sum over the L tokens of a [token = 16 t + lr][4 columns] accumulator set: in-lane over t, then over the 16 lanes sharing lane>>4
__device__ __forceinline__ f32x4 token_sum(const f32x4 (&o)[4], int L, int lr) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (16 * i + lr < L) t += o[i];
    // the 16 lanes are one DPP row: quad xor 1, quad xor 2, half-row mirror (quads 0<->1, 2<->3 hold equal sums by then), row
    // mirror -- four VALU instructions per value instead of four ds_bpermute round trips
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v = t[r];
        v += dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]
        v += dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]
        v += dpp_mov<0x141>(v);     // row_half_mirror
        v += dpp_mov<0x140>(v);     // row_mirror
        t[r] = v;
    }
    return t;
}
// (round 2, measured and rejected: TWO waves per (image, head) -- wave 0 = phase A / dQ, wave 1 = phase B / dK, dV, sharing the three
//  LDS images, the row statistics handed over behind one workgroup barrier; bitwise identical results, but 256 VGPRs + 88 B of
//  scratch per lane = four workgroups per CU, and 0.89 ms per step against 0.85: the kernel is not a per-wave latency chain that
//  more waves would hide -- the fragment-shaped 8-byte stores of dQ / dK / dV and the row-strided loads bound it.)
template <typename T>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                      T* __restrict__ dqkv, float* __restrict__ bias_part, int L, int heads,
                                                      float scale) {
    __shared__ __attribute__((aligned(16))) char smem[3 * TILEB + 2 * 64 * 4];
    char* qs = smem;
    char* ks_ = smem + TILEB;
    char* dos = smem + 2 * TILEB;
    float* lse_s = (float*)(smem + 3 * TILEB);
    float* dlt_s = lse_s + 64;

    const int lane = threadIdx.x, lr = lane & 15, lg = lane >> 4;
    const int img = blockIdx.x / heads, h = blockIdx.x % heads;
    const int D = heads * 64, ld = 3 * D;
    const T* qp = qkv + (size_t)img * L * ld + h * 64;
    const T* kp = qp + D;
    const T* vp = qp + 2 * D;
    const T* dop = dout + (size_t)img * L * D + h * 64;
    T* dqp = dqkv + (size_t)img * L * ld + h * 64;
    // optional: this (image, head)'s column sums of dQ | dK | dV (the in_proj bias gradient) -> partial row `img` of width 3D in
    // the blocked layout (eoe_part_index): a head's 64 columns of dQ / dK / dV are column blocks h, heads + h, 2 heads + h
    const int n_img = gridDim.x / heads;
    float* bpart = bias_part ? bias_part + ((size_t)h * n_img + img) * 64 : nullptr;
    const size_t bseg = (size_t)heads * n_img * 64;          // from the dQ block to the dK block to the dV block

    // V is only ever an MFMA operand in fragment layout: its 8 fragments are loaded straight from global memory ONCE, up front
    // (both phases use the same ones; loading them inside the phases exposed two more global round trips per wave)
    V8<T> vfr[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int t = 0; t < 4; ++t) vfr[ks][t] = gfrag<T>(vp, ld, L, t, ks, lane);
    stage_tile<T>(qs, qp, ld, L, lane);
    stage_tile<T>(ks_, kp, ld, L, lane);
    stage_tile<T>(dos, dop, D, L, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- phase A: lane = query.  S^T = K Q^T, dP^T = V dO^T
    {
        f32x4 s[4][4], dp[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                dp[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            V8<T> kf[4], qf[4], vf[4], df[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                kf[t] = lfrag<T>(ks_, t, ks, lane);
                qf[t] = lfrag<T>(qs, t, ks, lane);
                vf[t] = vfr[ks][t];
                df[t] = lfrag<T>(dos, t, ks, lane);
            }
#pragma unroll
            for (int tk = 0; tk < 4; ++tk)
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) {
                    s[tk][tq] = T16<T>::mfma16(kf[tk], qf[tq], s[tk][tq]);
                    dp[tk][tq] = T16<T>::mfma16(vf[tk], df[tq], dp[tk][tq]);
                }
        }
        float mx[4], sm[4];
        softmax_T<T>(s, L, scale, lane, mx, sm);
        // delta[q] = sum_key P dP ;  dS^T = P^T (dP^T - delta) * scale   (written over dp)
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
            float dl = 0.f;
#pragma unroll
            for (int tk = 0; tk < 4; ++tk)
#pragma unroll
                for (int r = 0; r < 4; ++r) dl += s[tk][tq][r] * dp[tk][tq][r];
            dl = group_sum(dl);
#pragma unroll
            for (int tk = 0; tk < 4; ++tk)
#pragma unroll
                for (int r = 0; r < 4; ++r) dp[tk][tq][r] = s[tk][tq][r] * (dp[tk][tq][r] - dl) * scale;
            if (lg == 0) {
                lse_s[16 * tq + lr] = mx[tq] + __logf(sm[tq]);
                dlt_s[16 * tq + lr] = dl;
            }
        }
        // dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]
#pragma unroll
        for (int td = 0; td < 4; ++td) {
            f32x4 o[4];
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) o[tq] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const V8<T> kt = tfrag<T>(ks_, td, st, lane);
#pragma unroll
                for (int tq = 0; tq < 4; ++tq)
                    o[tq] = T16<T>::mfma16(kt, acc_as_operand<T>(dp[2 * st][tq], dp[2 * st + 1][tq]), o[tq]);
            }
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) {
                const int query = 16 * tq + lr;
                if (query < L)
                    *(u32x2*)(dqp + (size_t)query * ld + 16 * td + 4 * lg) = pack4<T>(o[tq][0], o[tq][1], o[tq][2], o[tq][3]);
            }
            if (bpart) {
                const f32x4 cs = token_sum(o, L, lr);
                if (lr == 0) *(f32x4*)(bpart + 16 * td + 4 * lg) = cs;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- phase B: lane = key (operands swapped).  S = Q K^T, dP = dO V^T; rows (registers) = queries
    {
        f32x4 s[4][4], dp[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                dp[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            V8<T> kf[4], qf[4], vf[4], df[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                kf[t] = lfrag<T>(ks_, t, ks, lane);
                qf[t] = lfrag<T>(qs, t, ks, lane);
                vf[t] = vfr[ks][t];
                df[t] = lfrag<T>(dos, t, ks, lane);
            }
#pragma unroll
            for (int tq = 0; tq < 4; ++tq)
#pragma unroll
                for (int tk = 0; tk < 4; ++tk) {
                    s[tq][tk] = T16<T>::mfma16(qf[tq], kf[tk], s[tq][tk]);
                    dp[tq][tk] = T16<T>::mfma16(df[tq], vf[tk], dp[tq][tk]);
                }
        }
        // P = exp(S*scale - lse[q]);  dS = P (dP - delta[q]) * scale
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
            const f32x4 lse = *(const f32x4*)(lse_s + 16 * tq + 4 * lg);
            const f32x4 dl = *(const f32x4*)(dlt_s + 16 * tq + 4 * lg);
#pragma unroll
            for (int tk = 0; tk < 4; ++tk)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __expf(s[tq][tk][r] * scale - lse[r]);
                    s[tq][tk][r] = p;
                    dp[tq][tk][r] = p * (dp[tq][tk][r] - dl[r]) * scale;
                }
        }
        // dV^T[d][key] = sum_q dO^T[d][q] P[q][key] ;  dK^T[d][key] = sum_q Q^T[d][q] dS[q][key]
#pragma unroll
        for (int td = 0; td < 4; ++td) {
            f32x4 ov[4], ok[4];
#pragma unroll
            for (int tk = 0; tk < 4; ++tk) {
                ov[tk] = (f32x4){0.f, 0.f, 0.f, 0.f};
                ok[tk] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const V8<T> dot = tfrag<T>(dos, td, st, lane);
                const V8<T> qt = tfrag<T>(qs, td, st, lane);
#pragma unroll
                for (int tk = 0; tk < 4; ++tk) {
                    ov[tk] = T16<T>::mfma16(dot, acc_as_operand<T>(s[2 * st][tk], s[2 * st + 1][tk]), ov[tk]);
                    ok[tk] = T16<T>::mfma16(qt, acc_as_operand<T>(dp[2 * st][tk], dp[2 * st + 1][tk]), ok[tk]);
                }
            }
#pragma unroll
            for (int tk = 0; tk < 4; ++tk) {
                const int key = 16 * tk + lr;
                if (key < L) {
                    *(u32x2*)(dqp + (size_t)key * ld + 2 * D + 16 * td + 4 * lg) = pack4<T>(ov[tk][0], ov[tk][1], ov[tk][2], ov[tk][3]);
                    *(u32x2*)(dqp + (size_t)key * ld + D + 16 * td + 4 * lg) = pack4<T>(ok[tk][0], ok[tk][1], ok[tk][2], ok[tk][3]);
                }
            }
            if (bpart) {
                const f32x4 cv = token_sum(ov, L, lr), ck = token_sum(ok, L, lr);
                if (lr == 0) {
                    *(f32x4*)(bpart + 2 * bseg + 16 * td + 4 * lg) = cv;
                    *(f32x4*)(bpart + bseg + 16 * td + 4 * lg) = ck;
                }
            }
        }
    }
}


// ---- four waves per (image, head) (round 2).  The one-wave kernel above keeps 5 waves on a CU (31 KB of LDS each), 1.25 per SIMD:
// its ~14 k issue cycles per wave sit inside ~55 k cycles of exposed global / LDS / store latency.  Here the same seven products are
// split by 16-row slabs: wave w owns queries 16w..16w+15 in phase A (S^T, dP^T columns of its queries, all keys -> row statistics, dS^T,
// dQ) and keys 16w..16w+15 in phase B (S, dP rows of all queries x its keys -> dV, dK), the three LDS images are shared, the row
// statistics cross between the phases behind one workgroup barrier.  (Round 3, measured and rejected: the dQ / dK / dV slabs through a
// per-wave [16][64] LDS tile so that every global store is a 16-byte piece of a whole 128-byte row segment instead of 8 bytes per lane --
// 48 -> 37.5 us stand-alone without the bias sums, but 9 KB more LDS per workgroup (5 -> 4 workgroups per CU) and in the training step
// the kernel went from 51-53 to 55.7 us per layer: its time is latency covered by occupancy, not store issue.)  A wave holds a quarter of the accumulators (<= 128 registers),
// so 4 workgroups = 16 waves fit a CU and one workgroup's load / store latency is another's issue time.  Same arithmetic per element
// as the one-wave kernel (same products, same k order); only the bias column sums associate differently (per-wave partials summed in
// wave order).
// column sums over the 16 tokens of one accumulator tile (token = lane & 15): the DPP row reduction of token_sum
__device__ __forceinline__ f32x4 slab_sum(const f32x4& o, bool valid) {
    f32x4 t = valid ? o : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v = t[r];
        v += dpp_mov<0xB1>(v);
        v += dpp_mov<0x4E>(v);
        v += dpp_mov<0x141>(v);
        v += dpp_mov<0x140>(v);
        t[r] = v;
    }
    return t;
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd4_kernel(const T* __restrict__ qkv, const T* __restrict__ dout, T* __restrict__ dqkv,
                                                        float* __restrict__ bias_part, int L, int heads, float scale) {
    __shared__ __attribute__((aligned(16))) char smem[3 * TILEB + 2 * 64 * 4 + 4 * 192 * 4];
    char* qs = smem;
    char* ks_ = smem + TILEB;
    char* dos = smem + 2 * TILEB;
    float* lse_s = (float*)(smem + 3 * TILEB);
    float* dlt_s = lse_s + 64;
    float* red = dlt_s + 64;                                  // [4 waves][dQ | dK | dV column sums, 64 each]

    const int lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int img = blockIdx.x / heads, h = blockIdx.x % heads;
    const int D = heads * 64, ld = 3 * D;
    const T* qp = qkv + (size_t)img * L * ld + h * 64;
    const T* kp = qp + D;
    const T* vp = qp + 2 * D;
    const T* dop = dout + (size_t)img * L * D + h * 64;
    T* dqp = dqkv + (size_t)img * L * ld + h * 64;
    const int n_img = gridDim.x / heads;
    float* bpart = bias_part ? bias_part + ((size_t)h * n_img + img) * 64 : nullptr;
    const size_t bseg = (size_t)heads * n_img * 64;
    const int mine = 16 * w + lr;                             // this lane's query (phase A) / key (phase B)
    float* myred = red + w * 192;

    // V only ever is a row-major MFMA operand: fragments straight from global memory (phase A: all keys; phase B: this wave's keys)
    // (all 16 loads of the wave unconditional and back to back -- see bload16; the slabs for the LDS images first: the barrier waits for them)
    const unsigned pitchb = (unsigned)ld * 2u, kcb = (unsigned)D * 2u, vcb = (unsigned)D * 4u;
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(qp, ((unsigned)(L - 1) * (unsigned)ld + 2u * D + 64u) * 2u);
    const __amdgpu_buffer_rsrc_t rdo = make_rsrc(dop, ((unsigned)(L - 1) * (unsigned)D + 64u) * 2u);
    (void)kp; (void)vp;
    u32x4 sq[2], sk[2], sd[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int row = 16 * w + it * 8 + (lane >> 3);
        const unsigned cb = (unsigned)(lane & 7) * 16u;
        sq[it] = bload16(rq, row, L, pitchb, cb);
        sk[it] = bload16(rq, row, L, pitchb, kcb + cb);
        sd[it] = bload16(rdo, row, L, (unsigned)D * 2u, cb);
    }
    V8<T> vfr[2][4], vfw[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int t = 0; t < 4; ++t) vfr[ks][t] = bfrag<T>(rq, pitchb, vcb, L, t, ks, lane);
        vfw[ks] = bfrag<T>(rq, pitchb, vcb, L, w, ks, lane);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int o = (16 * w + it * 8 + (lane >> 3)) * ROWB + (((lane & 7) * 16) ^ rswz(it * 8 + (lane >> 3)));
        *(u32x4*)(qs + o) = sq[it];
        *(u32x4*)(ks_ + o) = sk[it];
        *(u32x4*)(dos + o) = sd[it];
    }
    if (bpart) {
        // The K and V thirds of the in-projection bias gradient need no sums over dK / dV (round 4; they were 2 x 64 DPP operations per wave):
        //   sum_key dV[key][d] = sum_q (sum_key P[q][key]) dO[q][d] = sum_q dO[q][d]      -- a softmax row sums to one,
        //   sum_key dK[key][d] = sum_q (sum_key dS[q][key]) Q[q][d] = 0                   -- a softmax-backward row sums to zero
        // (a key bias does not reach the output at all; the reference's fp32 autograd leaves rounding noise there).  The dO column sums
        // come from the slab this wave has just staged: lane (row lane >> 3, 8 columns from 8 * (lane & 7)), rows >= L read as zeros.
        float cv[8];
        {
            float a0[8], a1[8];
            unpack8<T>(sd[0], a0);
            unpack8<T>(sd[1], a1);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = a0[j] + a1[j];
                v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
                cv[j] = v;
            }
        }
        if (lane < 8) {
            *(f32x4*)(myred + 128 + 8 * lane) = (f32x4){cv[0], cv[1], cv[2], cv[3]};
            *(f32x4*)(myred + 128 + 8 * lane + 4) = (f32x4){cv[4], cv[5], cv[6], cv[7]};
        } else if (lane < 24) {
            *(f32x4*)(myred + 64 + 4 * (lane - 8)) = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();

    // ---- phase A: lane = query of this wave's slab.  S^T = K Q^T, dP^T = V dO^T
    {
        f32x4 s[4], dp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dp[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const V8<T> qf = lfrag<T>(qs, w, ks, lane), df = lfrag<T>(dos, w, ks, lane);
#pragma unroll
            for (int tk = 0; tk < 4; ++tk) {
                s[tk] = T16<T>::mfma16(lfrag<T>(ks_, tk, ks, lane), qf, s[tk]);
                dp[tk] = T16<T>::mfma16(vfr[ks][tk], df, dp[tk]);
            }
        }
        // softmax over the keys of this lane's query: 16 in-lane values + the 4 lanes sharing lane & 15 (softmax_T, one query tile)
        float m = -INFINITY;
#pragma unroll
        for (int tk = 0; tk < 4; ++tk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * tk + 4 * lg + r;
                const float v = (key < L) ? s[tk][r] * scale : -INFINITY;
                s[tk][r] = v;
                m = fmaxf(m, v);
            }
        m = group_max(m);
        float sum = 0.f;
#pragma unroll
        for (int tk = 0; tk < 4; ++tk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(s[tk][r] - m);
                s[tk][r] = e;
                sum += e;
            }
        sum = group_sum(sum);
        const float inv = 1.0f / sum;
        float dl = 0.f;
#pragma unroll
        for (int tk = 0; tk < 4; ++tk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[tk][r] *= inv;
                dl += s[tk][r] * dp[tk][r];
            }
        dl = group_sum(dl);
#pragma unroll
        for (int tk = 0; tk < 4; ++tk)
#pragma unroll
            for (int r = 0; r < 4; ++r) dp[tk][r] = s[tk][r] * (dp[tk][r] - dl) * scale;
        if (lg == 0) {
            lse_s[mine] = m + __logf(sum);
            dlt_s[mine] = dl;
        }
        // dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]; tile td, register r of lane (lr, lg) = d 16 lg + 4 td + r (tfrag_il): the lane's
        // four tiles are 16 consecutive d of its query's row -- two 16-byte stores
        u32x2 pk[4];
#pragma unroll
        for (int td = 0; td < 4; ++td) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 2; ++st)
                o = T16<T>::mfma16(tfrag_il<T>(ks_, td, st, lane), acc_as_operand<T>(dp[2 * st], dp[2 * st + 1]), o);
            pk[td] = pack4<T>(o[0], o[1], o[2], o[3]);
            if (bpart) {
                const f32x4 cs = slab_sum(o, mine < L);
                if (lr == 0) *(f32x4*)(myred + 16 * lg + 4 * td) = cs;
            }
        }
        // phase B's V fragments have long landed; without this use the wait for them sits BEHIND the stores below, and vmcnt cannot tell a
        // load from a store that was issued later: phase B would start with s_waitcnt vmcnt(0), i.e. after the dQ stores have drained
        asm volatile("" : "+v"(vfw[0]), "+v"(vfw[1]));
        if (mine < L) {
            T* d = dqp + (size_t)mine * ld + 16 * lg;
            *(u32x4*)d = (u32x4){pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
            *(u32x4*)(d + 8) = (u32x4){pk[2][0], pk[2][1], pk[3][0], pk[3][1]};
        }
    }
    __syncthreads();

    // ---- phase B: lane = key of this wave's slab (operands swapped).  S = Q K^T, dP = dO V^T; rows (registers) = queries
    {
        f32x4 s[4], dp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dp[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const V8<T> kf = lfrag<T>(ks_, w, ks, lane);
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) {
                s[tq] = T16<T>::mfma16(lfrag<T>(qs, tq, ks, lane), kf, s[tq]);
                dp[tq] = T16<T>::mfma16(lfrag<T>(dos, tq, ks, lane), vfw[ks], dp[tq]);
            }
        }
        // P = exp(S*scale - lse[q]);  dS = P (dP - delta[q]) * scale
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
            const f32x4 lse = *(const f32x4*)(lse_s + 16 * tq + 4 * lg);
            const f32x4 dl = *(const f32x4*)(dlt_s + 16 * tq + 4 * lg);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(s[tq][r] * scale - lse[r]);
                s[tq][r] = p;
                dp[tq][r] = p * (dp[tq][r] - dl[r]) * scale;
            }
        }
        // dV^T[d][key] = sum_q dO^T[d][q] P[q][key] ;  dK^T[d][key] = sum_q Q^T[d][q] dS[q][key]
        u32x2 pv[4], pk[4];
#pragma unroll
        for (int td = 0; td < 4; ++td) {
            f32x4 ov = {0.f, 0.f, 0.f, 0.f}, ok = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                ov = T16<T>::mfma16(tfrag_il<T>(dos, td, st, lane), acc_as_operand<T>(s[2 * st], s[2 * st + 1]), ov);
                ok = T16<T>::mfma16(tfrag_il<T>(qs, td, st, lane), acc_as_operand<T>(dp[2 * st], dp[2 * st + 1]), ok);
            }
            pv[td] = pack4<T>(ov[0], ov[1], ov[2], ov[3]);
            pk[td] = pack4<T>(ok[0], ok[1], ok[2], ok[3]);
        }
        if (mine < L) {
            T* dv = dqp + (size_t)mine * ld + 2 * D + 16 * lg;
            T* dk = dqp + (size_t)mine * ld + D + 16 * lg;
            *(u32x4*)dv = (u32x4){pv[0][0], pv[0][1], pv[1][0], pv[1][1]};
            *(u32x4*)(dv + 8) = (u32x4){pv[2][0], pv[2][1], pv[3][0], pv[3][1]};
            *(u32x4*)dk = (u32x4){pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
            *(u32x4*)(dk + 8) = (u32x4){pk[2][0], pk[2][1], pk[3][0], pk[3][1]};
        }
    }
    if (bpart) {
        __syncthreads();
        const int t = threadIdx.x;
        if (t < 192) bpart[(size_t)(t >> 6) * bseg + (t & 63)] = (red[t] + red[192 + t]) + (red[384 + t] + red[576 + t]);
    }
}

}  // namespace

extern "C" int eoe_attn_fwd(const void* qkv, void* out, int n, int L, int heads, int dtype, void* stream) {
    EOE_CHECK_ARG(qkv && out && n > 0 && heads > 0, "attn_fwd: bad args");
    EOE_CHECK_ARG(L >= 1 && L <= 64, "attn: sequence length %d not in [1, 64]", L);
    const float scale = 0.125f;   // 1/sqrt(64)
    ProfScope ps("attn_fwd", 4.0 * n * heads * (double)L * L * 64, 2.0 * (double)n * L * heads * 64 * 4, stream);
    if (dtype == EOE_F16)
        hipLaunchKernelGGL((attn_fwd_kernel<f16_t>), dim3(n * heads), dim3(64), 0, (hipStream_t)stream,
                           (const f16_t*)qkv, (f16_t*)out, L, heads, scale);
    else if (dtype == EOE_BF16)
        hipLaunchKernelGGL((attn_fwd_kernel<bf16_t>), dim3(n * heads), dim3(64), 0, (hipStream_t)stream,
                           (const bf16_t*)qkv, (bf16_t*)out, L, heads, scale);
    else
        return eoe_set_error(EOE_ERR_ARG, "attn_fwd: bad dtype %d", dtype);
    EOE_CHECK_LAUNCH("attn_fwd");
    return 0;
}

extern "C" int eoe_attn_bwd(const void* qkv, const void* dout, void* dqkv, float* dbias, float* bias_scratch, int n, int L,
                            int heads, int dtype, void* stream) {
    EOE_CHECK_ARG(qkv && dout && dqkv && n > 0 && heads > 0, "attn_bwd: bad args");
    EOE_CHECK_ARG((dbias == nullptr) == (bias_scratch == nullptr), "attn_bwd: dbias and bias_scratch go together");
    EOE_CHECK_ARG(L >= 1 && L <= 64, "attn: sequence length %d not in [1, 64]", L);
    const float scale = 0.125f;
    ProfScope ps("attn_bwd", 14.0 * n * heads * (double)L * L * 64, 2.0 * (double)n * L * heads * 64 * 7, stream);
    const bool one_wave = g_attn_flags & 1;     // A/B switch: the round-1 kernel
    if (dtype == EOE_F16 && one_wave)
        hipLaunchKernelGGL((attn_bwd_kernel<f16_t>), dim3(n * heads), dim3(64), 0, (hipStream_t)stream,
                           (const f16_t*)qkv, (const f16_t*)dout, (f16_t*)dqkv, bias_scratch, L, heads, scale);
    else if (dtype == EOE_BF16 && one_wave)
        hipLaunchKernelGGL((attn_bwd_kernel<bf16_t>), dim3(n * heads), dim3(64), 0, (hipStream_t)stream,
                           (const bf16_t*)qkv, (const bf16_t*)dout, (bf16_t*)dqkv, bias_scratch, L, heads, scale);
    else if (dtype == EOE_F16)
        hipLaunchKernelGGL((attn_bwd4_kernel<f16_t>), dim3(n * heads), dim3(256), 0, (hipStream_t)stream,
                           (const f16_t*)qkv, (const f16_t*)dout, (f16_t*)dqkv, bias_scratch, L, heads, scale);
    else if (dtype == EOE_BF16)
        hipLaunchKernelGGL((attn_bwd4_kernel<bf16_t>), dim3(n * heads), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)qkv, (const bf16_t*)dout, (bf16_t*)dqkv, bias_scratch, L, heads, scale);
    else
        return eoe_set_error(EOE_ERR_ARG, "attn_bwd: bad dtype %d", dtype);
    EOE_CHECK_LAUNCH("attn_bwd");
    if (dbias) EOE_TRY(eoe_finish_reduce(bias_scratch, n, 3 * heads * 64, 3 * heads * 64, dbias, nullptr, nullptr, 1, stream));
    return 0;
}
