// Shared device/host helpers for the eoe_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/eoe_hip.h"

// ---------------------------------------------------------------------------------------------------
// error convention: 0 = ok, non-zero = error, message via eoe_last_error()
// ---------------------------------------------------------------------------------------------------
extern thread_local char g_eoe_err[512];
int eoe_set_error(int code, const char* fmt, ...);

#define EOE_CHECK_ARG(cond, ...)                         \
    do {                                                 \
        if (!(cond)) return eoe_set_error(EOE_ERR_ARG, __VA_ARGS__); \
    } while (0)

#define EOE_CHECK_LAUNCH(name)                                                            \
    do {                                                                                  \
        hipError_t e__ = hipGetLastError();                                               \
        if (e__ != hipSuccess)                                                            \
            return eoe_set_error(EOE_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

#define EOE_TRY(expr)              \
    do {                           \
        int rc__ = (expr);         \
        if (rc__ != 0) return rc__; \
    } while (0)

// in-library kernel timing (api.cpp): a scope brackets the launches of one entry point with hipEvents
bool eoe_prof_active();
int eoe_prof_begin(const char* name, double flops, double bytes, hipStream_t s);
void eoe_prof_finish(int idx, hipStream_t s);
// Deferred finish of partial-row column reductions: out[c / seg][c % seg] += sum_r part[r][c] for c < N.  The host wrappers that
// normally launch their own tiny finish kernel (LayerNorm backward, the GEMM's fused column sums, the attention backward's bias sums)
// append a job here instead while `eoe_tls_defer` is set; eoe_vit_block_bwd then finishes all jobs of a block in ONE launch
// (each finish kernel is a dependent 4-5 us bubble in the stream: 4 per block otherwise).
struct EoeRedJob {
    const float* part;
    int R, N, seg;            // R partial rows of width N; output segment length (N for a single output)
    int blocked;              // layout of `part`: 0 = [R][N];  1 = [N/64][R][64] (eoe_part_index)
    float* out[3];            // segment s = c / seg is added into out[s] (NULL = skip)
};
// Where producer row r puts its partial sum of column c.  Blocked layout (N % 64 == 0): all R partial values of a 64-column block
// are contiguous (R x 256 B), so the finish kernel's workgroup streams one contiguous range; with the plain [R][N] layout it reads
// 256-byte pieces 4N bytes apart -- from HBM (the rows are hundreds of microseconds old) that ran at 0.7 TB/s.
__host__ __device__ __forceinline__ size_t eoe_part_index(int c, int r, int R, int N, int blocked) {
    return blocked ? ((size_t)(c >> 6) * R + r) * 64 + (c & 63) : (size_t)r * N + c;
}
struct EoeRedJobs {
    EoeRedJob job[6];
    int tile_start[7];        // prefix sums of ceil(N / 64) workgroups per job
    int count;
    int overwrite;            // 1: out = sum (no zero-initialised accumulators needed), 0: out += sum
};
// the same for a whole tower's jobs, 64 per launch (eoe_red_table_flush; 3.4 KB of kernel arguments)
struct EoeRedJobsBig {
    EoeRedJob job[64];
    int tile_start[65];
    int count;
    int overwrite;
};
extern thread_local EoeRedJobs* eoe_tls_defer;
// appends to the deferred list if one is active (returns true), else returns false and the caller launches its own finish
bool eoe_defer_reduce(const float* part, int R, int N, int seg, float* o0, float* o1, float* o2, int blocked);
int eoe_flush_reduce(EoeRedJobs* jobs, void* stream);      // elementwise.hip
bool eoe_bn_sync_active();                                  // conv.hip: synchronised BatchNorm hook (eoe_set_bn_sync)
int eoe_bn_sync_allreduce(void* buf, int64_t count, int is_f64, void* stream);
void* eoe_bn_sync_user();                                    // the registered hook's user pointer (NULL when none)
// appends to the open batch, or (none open) finishes this one job right away with `out += sum`
int eoe_finish_reduce(const float* part, int R, int N, int seg, float* o0, float* o1, float* o2, int blocked, void* stream);

struct ProfScope {
    int idx; hipStream_t s;
    ProfScope(const char* name, double flops, double bytes, void* stream) : idx(-1), s((hipStream_t)stream) {
        if (eoe_prof_active()) idx = eoe_prof_begin(name, flops, bytes, s);
    }
    ~ProfScope() { if (idx >= 0) eoe_prof_finish(idx, s); }
};

// ---------------------------------------------------------------------------------------------------
// 16-bit element types: one code path templated on the storage/MFMA type
// ---------------------------------------------------------------------------------------------------
typedef _Float16 f16_t;
typedef __bf16 bf16_t;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef short i16x8 __attribute__((ext_vector_type(8)));
typedef short i16x4v __attribute__((__vector_size__(4 * sizeof(short))));   // type of the tr16 builtin
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <typename T> struct T16;
template <> struct T16<f16_t> {
    typedef f16x8 v8;
    typedef f16x4 v4;
    static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct T16<bf16_t> {
    typedef bf16x8 v8;
    typedef bf16x4 v4;
    static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};

template <typename T> __device__ __forceinline__ float to_f32(T x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x) { return (T)x; }   // RNE (v_cvt_*)

// pack 4 floats into 4 x 16-bit (8 bytes)
template <typename T> __device__ __forceinline__ u32x2 pack4(float a, float b, float c, float d) {
    typename T16<T>::v4 v;
    v[0] = (T)a; v[1] = (T)b; v[2] = (T)c; v[3] = (T)d;
    return __builtin_bit_cast(u32x2, v);
}
template <typename T> __device__ __forceinline__ void unpack4(u32x2 u, float* o) {
    typename T16<T>::v4 v = __builtin_bit_cast(typename T16<T>::v4, u);
    o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3];
}
template <typename T> __device__ __forceinline__ void unpack8(u32x4 u, float* o) {
    typename T16<T>::v8 v = __builtin_bit_cast(typename T16<T>::v8, u);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
}
template <typename T> __device__ __forceinline__ u32x4 pack8(const float* o) {
    typename T16<T>::v8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (T)o[i];
    return __builtin_bit_cast(u32x4, v);
}

// ---------------------------------------------------------------------------------------------------
// wavefront (64 lanes) reductions
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// buffer resource (raw, bounds-checked: an out-of-range offset reads 0)
typedef __attribute__((address_space(3))) void lds_void_t;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
#define EOE_OOB 0x80000000u

// QuickGELU x*sigmoid(1.702x) (custom_clip.py:68-70) and its derivative.  v_exp_f32 + v_rcp_f32 (1 ulp each) instead of expf and
// an IEEE division: 5 VALU instructions per element instead of 15 -- the GEMM epilogues that apply these to 64 outputs
// per lane are VALU-bound.  exp2 overflow (x << 0) gives rcp(inf) = 0, the same limit as the division.
__device__ __forceinline__ float sigmoid1702_f(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * x));
}
__device__ __forceinline__ float quick_gelu_f(float x) { return x * sigmoid1702_f(x); }
__device__ __forceinline__ float quick_gelu_grad_f(float x) {
    float s = sigmoid1702_f(x);
    return s * (1.0f + 1.702f * x * (1.0f - s));
}

// Exact unsigned division by a launch-time constant (Granlund-Montgomery round-up multiplier), for x < 2^31: three VALU
// instructions instead of the ~30 (32-bit) / ~100 (64-bit) of a hardware-less integer division.  The streaming kernels decode
// (pixel, channel-quad) from a flat index several times per 16-B load; with real divisions they are VALU-bound, not HBM-bound.
struct FDiv {
    unsigned mp, l, d;
    FDiv() = default;
    __host__ __device__ explicit FDiv(unsigned dd) : d(dd) {
        l = 0;
        while ((1ull << l) < dd) ++l;
        mp = (unsigned)((((1ull << 32) * ((1ull << l) - dd)) / dd) + 1);
    }
    __device__ __forceinline__ unsigned div(unsigned x) const { return (__umulhi(x, mp) + x) >> l; }
    __device__ __forceinline__ void divmod(unsigned x, unsigned& q, unsigned& r) const { q = div(x); r = x - q * d; }
};
// flat index over [img][h][w][channel-quad] -> its coordinates
struct QuadDecode {
    FDiv cc, W, H;
    QuadDecode() = default;
    QuadDecode(int cc_, int W_, int H_) : cc((unsigned)cc_), W((unsigned)W_), H((unsigned)H_) {}
    __device__ __forceinline__ void operator()(unsigned i, unsigned& c4, unsigned& pix, unsigned& w, unsigned& h, unsigned& img) const {
        cc.divmod(i, pix, c4);
        unsigned t;
        W.divmod(pix, t, w);
        H.divmod(t, img, h);
    }
};

// bijective XCD-aware remap of a 1-D block id (blocks b and b+8 share an XCD): every XCD gets a contiguous
// chunk of the logical tile order, so neighbouring tiles hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    int q = nblocks >> 3, r = nblocks & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// (tile row, tile column) of logical tile r = xcd_remap(...) (round 4).  With the plain row-major order an XCD's chunk is a band of whole tile rows
// and the ~64 workgroups it has resident touch EVERY tile column at once, i.e. the whole weight matrix -- 4.7 MB for the ViT's MLP against 4 MB of
// L2 per XCD: PMC showed the weights fetched 2.4 times per XCD (c_fc forward: 111 MB from the fabric for 24 MB of operands).  Where the chunk is
// whole rows and the column count is even, the chunk's LEFT half of the columns goes first for all its rows, then the right half: a resident round
// works on half the weights.  c_fc forward 0.937 -> 0.906 ms per step, step -0.04 ms (tools/nt_shapes_ab.sh 524288 0); quarters measured worse.
__device__ __forceinline__ void xcd_halves(int r, int total, int tiles_n, bool on, int& mt, int& nt) {
    mt = r / tiles_n;
    nt = r - mt * tiles_n;
    if (!on || (total & 7) || (tiles_n & 1) || ((total >> 3) % tiles_n)) return;
    const int q = total >> 3, x = r / q, l = r - x * q, rows = q / tiles_n, hn = tiles_n >> 1, per = rows * hn;
    const int half = l / per, ll = l - half * per;
    mt = x * rows + ll / hn;
    nt = half * hn + ll % hn;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
