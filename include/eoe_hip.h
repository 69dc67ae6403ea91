/* eoe_hip.h -- C ABI of libeoe_hip.so: the MI355X (gfx950) kernels of the outlier-exposure AD training hot path.
 *
 * The reference (liznerski/eoe) is pure Python and has no FFI; its boundary for this path is the duck-typed
 * torch.nn.Module / ADTrainer-hook interface.  Every entry point below names the
 * reference call it replaces (path:line relative to /root/reference).  Conventions:
 *   - plain pointers and sizes only; the caller (PyTorch) owns every buffer, including workspaces;
 *   - every call takes the HIP stream to launch on (`void* stream` = hipStream_t);
 *   - return 0 on success, non-zero on error; the message is in the thread-local eoe_last_error();
 *   - "16-bit" buffers hold IEEE fp16 (EOE_F16) or bfloat16 (EOE_BF16), chosen per call with `dtype`;
 *     accumulation, LayerNorm statistics, the residual stream, losses, gradients of parameters and the
 *     optimiser state are fp32;
 *   - token tensors are batch-major:  row = image * L + token  (the reference's LND permute,
 *     clip/model.py:227,229, is value-preserving).
 */
#ifndef EOE_HIP_H
#define EOE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EOE_ABI_VERSION 5

enum { EOE_OK = 0, EOE_ERR_ARG = 1, EOE_ERR_LAUNCH = 2, EOE_ERR_UNSUPPORTED = 3 };
enum { EOE_F16 = 1, EOE_BF16 = 2, EOE_F32 = 3 /* only where an entry point says so */ };

int eoe_abi_version(void);
/* sizeof of an argument struct as compiled into the library: 0 eoe_gemm_args, 1 eoe_conv_geometry, 2 eoe_adam_chunk,
 * 3 eoe_adam_scalars, 4/5 eoe_vit_block_fwd/bwd_args, 6/7 eoe_cgate(_bwd)_args, 8/9 eoe_sgate(_bwd)_args, 10 eoe_adam_tile, 11 eoe_red_table;
 * -1 otherwise */
int eoe_struct_size(int which);
const char* eoe_last_error(void);

/* ------------------------------------------------------------------------------------------------------
 * GEMM (MFMA 16x16x32, LDS-staged 128x128x64 tiles, fp32 accumulate)
 *   replaces: every nn.Linear / F.linear / conv1-as-GEMM / `x @ proj` on the path
 *   (clip/model.py:171-178,220,233-234; custom_base.py:25-26,47-48) and their autograd backward
 *   (ad_trainer.py:431).
 *   NT:  C[M,N] = A[M,K] . B[N,K]^T            (forward: B = weight;  dgrad: B = transposed weight copy)
 *   TN:  C[M,N] = A[T,M]^T . B[T,N]            (wgrad: A = dY, B = X, reduction over the T tokens)
 * ---------------------------------------------------------------------------------------------------- */
enum {
    EOE_EPI_NONE = 0,      /* C = acc (+ bias)                                                         */
    EOE_EPI_GELU = 1,      /* pre = acc + bias -> aux_out (16-bit; NULL: not kept);  C = pre * sigmoid(1.702 pre) of the ROUNDED pre */
    EOE_EPI_RESIDUAL = 2,  /* C(fp32) = acc + bias + aux(fp32 [M,N], leading dim ldaux)                */
    EOE_EPI_GELU_BWD = 3   /* C = acc * d/dpre[pre sigmoid(1.702 pre)],  pre = aux (16-bit [M,N])      */
};

/* geometry of a convolution whose patch matrix is never materialised ("implicit GEMM": the GEMM kernel fetches every
 * 16-byte piece -- 8 channels of one tap of one output pixel -- from the 16-bit NHWC activation [n, H, W, C] straight into
 * its LDS stage; taps in the zero padding arrive as zeros).  Patch row = output pixel (img, ho, wo) of the Ho x Wo grid,
 * patch column = (ky*kw + kx)*C + c  ->  element (img, ho*stride - pad + ky, wo*stride - pad + kx, c). */
typedef struct {
    int32_t n, H, W, C;           /* gathered tensor */
    int32_t kh, kw, stride, pad;
    int32_t Ho, Wo;               /* grid of patch rows */
} eoe_conv_geometry;

typedef struct {
    const void* A;      /* 16-bit */
    const void* B;      /* 16-bit */
    void* C;            /* 16-bit, or fp32 if out_f32 */
    const float* bias;  /* [N] fp32 or NULL */
    const void* aux;    /* see epilogue */
    void* aux_out;      /* see epilogue */
    float* colsum;      /* optional (NT only): colsum[n] += sum_m C[m,n] of the epilogue result -- fp32 atomics, or, with a
                         * workspace of >= EOE_NT_COLSUM_WORKSPACE_BYTES(M, N), per-wave partial rows added up by a second kernel
                         * in a fixed order (bitwise reproducible; the atomics cost +45 us on a 12800 x 3072 output) */
    int32_t M, N, K;    /* for TN, K is the reduction length T */
    int32_t lda, ldb, ldc, ldaux;   /* leading dimensions in elements */
    int32_t dtype;      /* EOE_F16 | EOE_BF16 */
    int32_t epilogue;   /* EOE_EPI_* */
    int32_t out_f32;    /* C is fp32 */
    int32_t accumulate; /* C += result (C must be fp32) */
    float alpha;        /* result scale applied to acc before bias/epilogue (1.0 = none) */
    void* workspace;    /* optional (TN): scratch for the per-split partial results of a split reduction; with
                         * workspace_bytes >= EOE_TN_WORKSPACE_BYTES the splits are summed by a second kernel instead of fp32
                         * atomics (faster, and bitwise reproducible).  With >= EOE_TN_STREAMK_WORKSPACE_BYTES(#CUs) a group
                         * whose tiles fill most but not all of the CUs once (216 of 256) runs as #CUs equal k-ranges (stream-K;
                         * the first problem's workspace is the one used) */
    int64_t workspace_bytes;
    int32_t gather;     /* 1: A is the NHWC tensor of `geo` and stands for its patch matrix (lda ignored).
                         *    NT: [M = n*Ho*Wo, K = kh*kw*C], C % 64 == 0 (one tap per 64-deep k-tile), or C in {8,16,32} with K
                         *        = kh*kw*C rounded up to a multiple of 64 (every 16-byte piece decodes its own tap); plain epilogue
                         *        (conv forward; stride-1 dgrad)
                         *    TN: [T = n*Ho*Wo, M = kh*kw*C], C % 8 == 0 (conv wgrad, transposed: C[kh*kw*C, cout])
                         * 2: A is the zero-padded NHWC4 image of eoe_stem_pack_image (geo.C = 4, geo.pad = 0, geo.H/W = Hp/Wp),
                         *    K (NT) / M (TN) = ceil(kh/2)*64 */
    eoe_conv_geometry geo;
    int32_t colstats;   /* NT only: the workspace (>= 2 * EOE_NT_COLSUM_WORKSPACE_BYTES(M, N)) receives, per 64 output rows, the column
                         * sums and sums of squares of the fp32 epilogue result: [ceil(M/64)][2][N] floats -- BatchNorm batch
                         * statistics without a pass over C (eoe_bn_stats_partials reduces them); N % 16 == 0 */
    int32_t unpack_dw;  /* TN with gather == 1 only: C is the convolution's weight gradient in the reference's own layout
                         * [cout][cin][kh][kw] (not the [kh*kw*cin][cout] matrix): the launch goes through the workspace's partial
                         * tiles and the reduce kernel writes that order directly -- no eoe_conv_unpack_wgrad pass.  Needs the
                         * workspace (>= splits * M * N * 4 bytes, EOE_TN_WORKSPACE_BYTES always suffices for the shapes here) */
    int32_t split_k;    /* NT only, a hint: != 0 asks for the split form of a small-M problem behind a long K (M <= 1024, K >= 1536, plain or
                         * fp32-residual epilogue, no column sums): up to 8 equal k-ranges per output tile in one launch, fp32 partial tiles in the
                         * slot area of `sk_workspace` (needed: without it the hint is ignored), summed in a fixed order by a second kernel.  The split depends on K only, so a row's result
                         * does not depend on M (eoe_vit_block_fwd / _bwd ask for it on the class-token-only block's n x 768 x 3072 products:
                         * 12 tiles of 48 k-tiles are one 50-us latency chain on 12 CUs).  Ignored where it does not apply. */
    void* sk_workspace; /* NT only, optional: >= EOE_NT_STREAMK_WORKSPACE_BYTES(#CUs) bytes, 16-byte aligned, ZEROED ONCE by the caller before its
                         * first use (every launch leaves the ticket / flag words at its head zeroed again), used by one stream at a time.
                         * With it the eight-wave 256 x 256 kernel runs in its stream-K form where the tiles do not fill the CUs a whole
                         * number of times: full rounds of tiles data-parallel, the fractional last round cut along k over all CUs, fp32
                         * partial accumulators through this workspace, added in segment order by the tile's last-arriving workgroup
                         * (bitwise reproducible; no atomics on data).  A row's bits then depend on M (where the cut falls) at the level of
                         * fp32 summation order.  NULL: data-parallel tiles only. */
    int64_t sk_workspace_bytes;
} eoe_gemm_args;
/* head: 8 KiB of ticket / flag words; then two 256 x 256 fp32 partial tiles per CU */
#define EOE_NT_STREAMK_WORKSPACE_BYTES(cus) ((size_t)8192 + (size_t)(cus) * 2 * (256 * 256 * 4))
#define EOE_NT_COLSUM_WORKSPACE_BYTES(M, N) ((size_t)(((M) + 63) / 64) * (size_t)(N) * 4)

int eoe_gemm_nt(const eoe_gemm_args* args, void* stream);
int eoe_gemm_tn(const eoe_gemm_args* args, void* stream);
/* up to EOE_TN_MAX_GROUP independent TN problems with the same reduction length T, dtype, accumulate and alpha in
 * ONE launch (the four weight gradients of a transformer block fill the chip together: no split-K, no atomics) */
#define EOE_TN_MAX_GROUP 4
#define EOE_TN_WORKSPACE_BYTES (512ll * 256 * 128 * 4)   /* (#CUs rounded up) x one 256x128 fp32 tile */
#define EOE_TN_STREAMK_WORKSPACE_BYTES(cus) ((size_t)(cus) * (256 * 256 * 4))   /* one partial fp32 tile (up to 256 x 256) per CU */
int eoe_gemm_tn_grouped(const eoe_gemm_args* args, int count, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * element-wise / reduction kernels (HBM-bound; wavefront-shuffle reductions)
 * ---------------------------------------------------------------------------------------------------- */

/* fp32 [rows, cols] -> 16-bit copy [rows, cols] and (if dst_t != NULL) transposed 16-bit copy [cols, rows].
 * Produces the MFMA operand copies of the fp32 master weights once per optimiser step. */
int eoe_cast_transpose(const float* src, void* dst, void* dst_t, int rows, int cols, int dtype, void* stream);
/* the same for `count` matrices (rows, cols multiples of 4) in one launch per 56 jobs: the per-step refresh of all 16-bit
 * weight copies of a model (each matrix alone is too small to fill the chip) */
typedef struct {
    const float* src;   /* fp32 [rows, cols] */
    void* dst;          /* 16-bit [rows, cols] or NULL */
    void* dst_t;        /* 16-bit [cols, rows] or NULL */
    int32_t rows, cols;
} eoe_cast_job;
int eoe_cast_transpose_multi(const eoe_cast_job* jobs, int count, int dtype, void* stream);

/* per-channel affine + patch extraction: x fp32 [n,3,res,res] (NCHW, as the reference feeds it,
 * ad_trainer.py:411-429) -> 16-bit patch matrix [n*(res/p)^2, 3*p*p], column = c*p*p + ky*p + kx
 * (the flattened conv1 weight order, clip/model.py:207,220).  mean/std NULL = already normalised
 * (replaces transformations.py:126-138 applied at ad_trainer.py:413-425). */
int eoe_patchify(const float* x, const float* mean, const float* std, void* out, int n, int res, int patch,
                 int dtype, void* stream);

/* token assembly + ln_pre (clip/model.py:223-225): tok fp32 [n*(L-1), D] patch embeddings,
 * cls [D], pos [L, D] -> x0 fp32 [n*L, D] = cat(cls, tok) + pos (kept for backward),
 * y fp32 [n*L, D] = LayerNorm(x0) (the residual stream entering block 0), stats [n*L, 2] = (mean, rstd). */
int eoe_embed_lnpre_fwd(const float* tok, const float* cls, const float* pos, const float* gamma,
                        const float* beta, float* x0, float* y, float* stats, int n, int L, int D, float eps,
                        void* stream);
/* backward of the above: dy fp32 [n*L, D] -> dtok 16-bit [n*(L-1), D] (operand of the conv1 wgrad GEMM),
 * dcls[D] +=, dpos[L,D] +=, dgamma[D] +=, dbeta[D] += (caller zeroes or accumulates).  scratch: optional, L * 2 * D floats ->
 * dgamma / dbeta through per-token-position partial rows + a fixed-order finish kernel instead of fp32 atomics. */
int eoe_embed_lnpre_bwd(const float* dy, const float* x0, const float* stats, const float* gamma, void* dtok,
                        float* dcls, float* dpos, float* dgamma, float* dbeta, float* scratch, int n, int L, int D, int dtype,
                        void* stream);

/* LayerNorm over the last dim (clip/model.py:153-159, fp32 statistics, biased variance):
 * x fp32 [rows, D] (row stride ldx elements) -> y 16-bit [rows, D] (GEMM operand) or fp32 if out_f32,
 * stats [rows, 2]. */
int eoe_layernorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, void* y, float* stats,
                      int rows, int D, float eps, int dtype, int out_f32, void* stream);
/* dx_out[r, :] (fp32, row stride ld_out) = (dres ? dres[r, :] : 0) + LN'(dy)[r, :];  dx16 (optional) = 16-bit
 * copy of dx_out (the next GEMM's operand); dgamma/dbeta += column sums of dy*xhat / dy;  dxsum (optional) +=
 * column sums of dx_out (the bias gradient of the layer that produced the LayerNorm input's residual branch).
 * dy is 16-bit, or fp32 if dy_f32.  red_scratch: optional, EOE_LN_SCRATCH(D) floats -- the column sums then go through
 * per-workgroup partial rows and a fixed-order reduce kernel (no atomics: bitwise reproducible, and 9 us faster per call at
 * 12800 x 768) and are still ADDED to dgamma / dbeta / dxsum; NULL = fp32 atomics. */
#define EOE_LN_PARTIALS 512
#define EOE_LN_SCRATCH(D) ((size_t)EOE_LN_PARTIALS * 3 * (D))
int eoe_layernorm_bwd(const void* dy, int dy_f32, const float* x, int ldx, const float* stats,
                      const float* gamma, const float* dres, float* dx_out, int ld_out, void* dx16,
                      float* dgamma, float* dbeta, float* dxsum, float* red_scratch, int rows, int D, int dtype,
                      void* stream);

/* out[c] (+)= sum_r x[r, c]  -- bias gradients (x 16-bit [rows, cols], row stride ldx). */
int eoe_colsum(const void* x, int ldx, float* out, int rows, int cols, int dtype, int accumulate, void* stream);
/* the same sums without atomics or memset (out is overwritten): per-workgroup partial rows in `scratch`
 * (EOE_COLSUM_PARTIALS * cols floats) + a reduce kernel -- bitwise reproducible, kernels only (graph capture) */
#define EOE_COLSUM_PARTIALS 256
int eoe_colsum_det(const void* x, int ldx, float* out, float* scratch, int rows, int cols, int dtype, void* stream);

/* dst = 16-bit copy of x fp32 [rows, cols] and out[c] (+)= sum_r x[r, c] in one pass.  scratch: optional, EOE_CAST_COLSUM_PARTIALS * cols
 * floats -> the sums go through per-workgroup partial rows + a fixed-order finish kernel instead of fp32 atomics */
#define EOE_CAST_COLSUM_PARTIALS 256
int eoe_cast_colsum(const float* x, void* dst, float* out, float* scratch, int rows, int cols, int dtype, int accumulate, void* stream);

/* narrow linear head, N <= 8 outputs, exact fp32 (the 1-wide `final_linear` of CustomNet(clf=True),
 * custom_base.py:25-26, used by the BCE objective): y[M,N] = x[M,K] w[N,K]^T + bias;
 * backward: dx (optional) = dy w, dw (+)= dy^T x, db (optional) (+)= column sums of dy. */
int eoe_linear_small_fwd(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, void* stream);
int eoe_linear_small_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int M, int N,
                         int K, int accumulate, void* stream);

/* zero up to EOE_ZERO_MAX fp32 buffers in one launch (accumulators that the kernels above fill with atomics) */
#define EOE_ZERO_MAX 12
int eoe_zero_multi(float* const* ptrs, const int* counts, int n, void* stream);

/* fp32 -> 16-bit copy of n contiguous elements */
int eoe_cast(const float* src, void* dst, size_t n, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * multi-head self-attention over short sequences (L <= 64, head dim 64), one wavefront per (image, head):
 *   replaces nn.MultiheadAttention(768, 12) inside ResidualAttentionBlock.attention
 *   (clip/model.py:171,181-183), without mask or dropout.
 * qkv 16-bit [n*L, 3*D] (q | k | v, each D = heads*64 wide), out 16-bit [n*L, D].
 * ---------------------------------------------------------------------------------------------------- */
int eoe_attn_fwd(const void* qkv, void* out, int n, int L, int heads, int dtype, void* stream);
/* dbias (optional, fp32 [3D]): += column sums of dqkv = the in_proj bias gradient, taken from the kernel's fp32 accumulators
 * through bias_scratch (fp32 [n, 3D], required with dbias) and a fixed-order reduce kernel -- no separate pass over dqkv */
int eoe_attn_bwd(const void* qkv, const void* dout, void* dqkv, float* dbias, float* bias_scratch, int n, int L, int heads,
                 int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * objectives (fused heads)
 * ---------------------------------------------------------------------------------------------------- */
/* HSC (hsc.py:12-21): f fp32 [n, d]; labels int64 [n]; per sample  dist = sqrt(|f|^2+1)-1,
 * score = 1-exp(-dist), loss_i = dist if label==nominal else -log(score+1e-9);
 * loss[0] = inv_count * sum_i loss_i  (inv_count = 1/n for the plain mean, 1/global_n under data parallel).
 * Any of scores / dists / losses may be NULL. */
int eoe_hsc_fwd(const float* f, const int64_t* labels, int64_t nominal_label, float* loss, float* scores,
                float* dists, float* losses, int n, int d, float inv_count, void* stream);
/* df fp32 [n,d] = gscale[0] * inv_count * dloss_i/df (gscale = upstream gradient of the scalar loss, device
 * pointer or NULL = 1);  df16 (optional) = 16-bit copy. */
int eoe_hsc_bwd(const float* f, const int64_t* labels, int64_t nominal_label, const float* gscale, float* df,
                void* df16, int n, int d, float inv_count, int dtype, void* stream);
/* score only (hsc.py:12-15; ad_trainer.py:434-436,505) */
int eoe_hsc_score(const float* f, float* scores, int n, int d, void* stream);

/* BCE with logits (bce.py:15-20): x fp32 [n] logits; loss[0] = inv_count * sum_i bce_i; scores = sigmoid(x)
 * (1 - sigmoid if nominal_label != 0). */
int eoe_bce_fwd(const float* x, const int64_t* labels, int64_t nominal_label, float* loss, float* scores,
                float* losses, int n, float inv_count, void* stream);
int eoe_bce_bwd(const float* x, const int64_t* labels, const float* gscale, float* dx, int n, float inv_count,
                void* stream);

/* ------------------------------------------------------------------------------------------------------
 * fused multi-tensor Adam (ad_trainer.py:383: torch.optim.Adam(params, lr, weight_decay=wdk, amsgrad=False))
 * One launch over a chunk table.  Per chunk: element offset into each of the four fp32 arenas
 * (p, g, m, v may also be distinct allocations: offsets are relative to the pointers given here) and a length.
 * L2-in-gradient weight decay, bias correction per step-count group (frozen parameters, whose grad is None,
 * are simply not in the table and keep their step count: SURVEY.md section 7 "Adam details").
 * shadow16 (optional, may be NULL): 16-bit copy of the updated parameter written at the same element offset.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct {
    int64_t p_off, g_off, m_off, v_off; /* element offsets */
    int32_t n;                          /* elements in this chunk (<= EOE_ADAM_CHUNK) */
    int32_t group;                      /* index into eoe_adam_scalars (parameters sharing a step count); | EOE_CHUNK_FP16, see eoe_sgd_multi */
} eoe_adam_chunk;
#define EOE_CHUNK_FP16 0x100            /* eoe_sgd_multi: this chunk's parameter is an fp16 tensor in the reference (clip/model.py:371-392) */
#define EOE_ADAM_CHUNK 8192
#define EOE_ADAM_GROUPS 4
typedef struct {                        /* per step-count group, computed on the host in double as torch does: */
    float step_size[EOE_ADAM_GROUPS];   /*   lr / (1 - beta1**step)                                            */
    float bc2_sqrt[EOE_ADAM_GROUPS];    /*   sqrt(1 - beta2**step)                                             */
    float grad_scale_inv;               /* the gradients are multiplied by this first (0 = 1): 1 / the loss scale of an fp16 run.
                                         * fp16's smallest subnormal is 6e-8: with 12 800 token rows sharing a mean loss the 16-bit
                                         * dY tensors of late training steps underflow, and the 10-step trajectory of the 12-layer
                                         * ViT drifts to 3e-3 off the fp32 reference; with the loss gradient scaled by 256 (the
                                         * `inv_count` argument of the loss kernels) it stays within 1.2e-4 */
} eoe_adam_scalars;

/* fused multi-tensor SGD with momentum / Nesterov = torch.optim.SGD(..., dampening=0) as constructed for CLIP models
 * (ad_trainer.py:380-381: momentum 0.9, nesterov): the same chunk tables (m_off = momentum buffer, zero-initialised; v_off unused).
 * fp16-weights mode (clip/model.py:371-392 `convert_weights`, applied by build_model :430): chunks with
 * `group & EOE_CHUNK_FP16` are updated as torch updates an fp16 parameter with an fp16 gradient -- the fp32 storage keeps
 * fp16-representable values, every elementwise op of the update rounds to fp16 (5 roundings per element, in torch's op order) */
int eoe_sgd_multi(float* p, const float* g, float* buf, const eoe_adam_chunk* chunks /*device*/, int n_chunks, float lr,
                  float momentum, float weight_decay, int nesterov, float grad_scale_inv /* as in eoe_adam_scalars; <= 0 = 1 */,
                  const int32_t* skip_flag /* device, may be NULL: see eoe_grads_nonfinite */, void* stream);
int eoe_adam_multi(float* p, const float* g, float* m, float* v, const eoe_adam_chunk* chunks /*device*/,
                   int n_chunks, const eoe_adam_scalars* scalars /*host*/, float beta1, float beta2, float eps,
                   float weight_decay, void* shadow16, int dtype, const int32_t* skip_flag /* device, may be NULL */, void* stream);
/* The same update for 2-D weights, walked by 64 x 64 tiles, which ALSO writes the two 16-bit MFMA operand copies of the new weight --
 * [rows, cols] and its transpose [cols, rows] -- that the next forward / backward multiply with (round 4).  Until then the copies were
 * made by a pass of their own before the next forward (eoe_cast_transpose_multi: the 85 M weights of the 12 blocks read again as fp32,
 * 0.14 ms per step); here they leave from the registers the update is in, through an LDS transposition.  p, m, v get exactly what
 * eoe_adam_multi writes (the same update function).  One table row per tile; rows, cols multiples of 4.  New relative to the
 * reference (torch.optim.Adam has no such copies: autocast re-casts the weights in every forward). */
typedef struct {
    int64_t p_off, g_off, m_off, v_off; /* element offsets of the MATRIX (not of the tile) from the four base pointers */
    void* d16;                          /* [rows, cols] 16-bit copy (may be NULL) */
    void* d16_t;                        /* [cols, rows] 16-bit transposed copy (may be NULL) */
    int32_t rows, cols;
    int32_t tile;                       /* this row's tile: (tile / ceil(cols / 64), tile % ceil(cols / 64)) */
    int32_t group;                      /* as in eoe_adam_chunk */
} eoe_adam_tile;
int eoe_adam_tiles(float* p, const float* g, float* m, float* v, const eoe_adam_tile* tiles /*device*/, int n_tiles,
                   const eoe_adam_scalars* scalars /*host*/, float beta1, float beta2, float eps, float weight_decay, int dtype,
                   const int32_t* skip_flag /* device, may be NULL */, void* stream);
/* Non-finite guard for the scaled fp16 step.  The reference's numerical-failure policy is the per-epoch NaN check with retry
 * (ad_trainer.py:257-280, 448-449); the loss-gradient scale this build adds for fp16 can overflow the 16-bit backward chain, so the
 * optimisers can be told to drop such a step whole.  One streaming pass over the gradients of the chunk table the optimiser is about to
 * apply.  `state` = 4 device ints {flag of even steps, flag of odd steps, steps skipped so far, steps checked so far}, zero-initialised
 * by the caller once; `parity` alternates 0 / 1 from step to step; `first` != 0 on the first chunk table of a step (several parameter
 * groups = several calls with the same parity): that call retires the previous step's flag into the skipped count.  An inf / NaN raises
 * state[parity]; pass &state[parity] as `skip_flag` to eoe_adam_multi / eoe_sgd_multi of the same step: with the flag up they touch
 * nothing (parameters, moments, 16-bit shadows).  No host synchronisation anywhere; the host reads state[2] when it cares to. */
int eoe_grads_nonfinite(const float* g, const eoe_adam_chunk* chunks /*device*/, int n_chunks, int32_t* state /*device*/, int parity,
                        int first, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * fused ViT residual block (clip/model.py:167-188 ResidualAttentionBlock.forward and its backward):
 * one call launches the whole kernel chain of a block on `stream` (no host round trips in between).
 * ---------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n, L, D, heads, dtype;   /* rows M = n*L; hidden = 4*D */
    float eps;
    /* parameters: fp32 vectors, 16-bit matrices (row-major [out, in]) and their transposes [in, out] */
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *b_in, *b_out, *b_fc, *b_proj;
    const void *w_in, *w_out, *w_fc, *w_proj;           /* [3D,D] [D,D] [4D,D] [D,4D]   */
    const void *w_in_t, *w_out_t, *w_fc_t, *w_proj_t;   /* [D,3D] [D,D] [D,4D] [4D,D]   */
    /* activations (caller-allocated; saved for backward) */
    const float* x_in;   /* fp32 [M,D] residual stream in */
    float* x_mid;        /* fp32 [M,D] after attention */
    float* x_out;        /* fp32 [M,D] residual stream out */
    void *xn1, *qkv, *att, *xn2, *hpre, *hact;   /* 16-bit [M,D] [M,3D] [M,D] [M,D] [M,4D] [M,4D] */
    float *stats1, *stats2;                      /* fp32 [M,2] */
    /* != 0: the caller reads only the class-token rows (row i*L of image i) of x_out -- the LAST block of the vision tower, whose output
     * goes through x[:, 0, :] to ln_post (clip/model.py:231-232).  The rows nobody reads are not computed: the out-projection, LayerNorm-2
     * and the MLP run on the n class-token rows, and x_mid, x_out, xn2, hpre, hact, stats2 hold [n, ...] (dense, image order) instead of
     * [n*L, ...]; xn1, qkv, att keep all rows (keys and values of every token feed the class token's attention).  Backward: dx_out is
     * fp32 [n, D]; dx_in and every parameter gradient are what the full computation gives with zero dx_out on the other rows (the
     * weight gradients to fp32 summation order: the zero rows are left out of the sums). */
    int32_t cls_only;
    /* optional: the stream-K workspace of the block's NT GEMMs (eoe_gemm_args.sk_workspace: zeroed once by the caller, one stream at a time;
     * forward and backward use the same one).  NULL: data-parallel tiles only. */
    void* nt_sk_workspace;
    int64_t nt_sk_workspace_bytes;
} eoe_vit_block_fwd_args;

/* deferred finish reductions: out[c / seg][c % seg] (+)= sum over r < R of part[r][c], c < N (blocked: eoe's [N/64][R][64] partial-row layout) */
typedef struct {
    const float* part;
    int32_t R, N, seg, blocked;
    float* out[3];
} eoe_red_job;
#define EOE_RED_TABLE_MAX 128
typedef struct eoe_red_table {
    eoe_red_job job[EOE_RED_TABLE_MAX];
    int32_t count;
    int32_t overwrite;          /* 1: out = sum, 0: out += sum (set by the first appender; all jobs of a table agree) */
} eoe_red_table;
/* launches the table's jobs (64 per launch) on `stream` and empties it */
int eoe_red_table_flush(eoe_red_table* table, void* stream);

typedef struct {
    eoe_vit_block_fwd_args f;   /* same parameters and saved activations as the forward */
    const float* dx_out;        /* fp32 [M,D] gradient wrt x_out ([n,D] with f.cls_only) */
    float* dx_in;               /* fp32 [M,D] gradient wrt x_in */
    /* parameter gradients, fp32, overwritten (or += if accumulate) */
    float *g_ln1_g, *g_ln1_b, *g_ln2_g, *g_ln2_b, *g_b_in, *g_b_out, *g_b_fc, *g_b_proj;
    float *g_w_in, *g_w_out, *g_w_fc, *g_w_proj;
    int32_t accumulate;
    /* scratch (caller-allocated, 16-bit unless noted): */
    void *d16_a;      /* [M,D]   16-bit copy of dx_out (dY of c_proj)          */
    void *d16_b;      /* [M,D]   dgrad outputs (d xn2, d att, d xn1 in turn)   */
    void *d16_c;      /* [M,D]   16-bit copy of dx_mid (dY of out_proj)        */
    void *dh;         /* [M,4D]  */
    void *dqkv;       /* [M,3D]  */
    float* dx_mid;    /* fp32 [M,D] */
    float* red_scratch; /* optional: EOE_VIT_RED_SCRATCH(n, L, D) floats -> LayerNorm-parameter and bias column sums through
                         * partial rows and fixed-order reduce kernels instead of fp32 atomics / separate passes */
    void* tn_workspace; /* optional: EOE_TN_STREAMK_WORKSPACE_BYTES(#CUs) bytes -> the grouped wgrad launch balances its 216 tiles
                         * over all CUs (eoe_gemm_args.workspace) */
    int64_t tn_workspace_bytes;
    /* Hand-over between consecutive blocks of the backward sweep (all optional; round 3).  The producer of a block's dx_out is the NEXT
     * block's LayerNorm-1 backward, which can emit what this block's first step (eoe_cast_colsum: the 16-bit copy of dx_out = dY of
     * c_proj, and its column sums = the gradient of c_proj.bias) would otherwise compute in a pass of its own:
     *   next_d16       out: 16-bit copy [n*L, D] of THIS call's dx_in, for the block that runs next in backward
     *   in_d16         in:  that copy of THIS call's dx_out, written by the previous call (a different buffer than next_d16: the
     *                       previous call's weight-gradient GEMM still reads it while this call's LayerNorm writes)
     *   in_red_scratch in:  the previous call's red_scratch (must differ from this call's: alternate two): the column sums of dx_out
     *                       sit in its LayerNorm-1 partial rows and are finished into g_b_proj by this call's one finish kernel */
    void* next_d16;
    const void* in_d16;
    const float* in_red_scratch;
    /* != 0: the four weight gradients are launched on the library's side stream and this call returns without ordering `stream` behind
     * them: the launch (216 one-per-CU workgroups, ~213 us, 40 CUs idle) then runs under the NEXT block's kernels.  The caller keeps
     * every buffer that launch reads -- dh, dqkv, d16_c, the dY of c_proj (d16_a or in_d16), the saved activations xn1 / xn2 / att / hact --
     * untouched until a later eoe_vit_block_bwd on the same stream has returned (alternate two sets of scratch; that call orders the stream
     * behind the previous launch at its own fork point) or eoe_vit_side_join(stream) was called; the weight gradients themselves may only
     * be read after eoe_vit_side_join.  Ignored (synchronous launch) while the stream is being captured or without red_scratch.
     * 2 (round 5): as 1, but a call orders the stream only behind the launch BEFORE the previous call's (the previous one is often still
     * running at this call's fork point: waiting for it stalled the compute stream ~12 us per block).  The caller then rotates THREE sets of
     * those buffers and keeps a call's buffers untouched until TWO later calls on the stream have returned (or eoe_vit_side_join). */
    int32_t async_wgrad;
    /* optional (round 5): a HOST table the block appends its finish reductions to (the five or six partial-row column sums behind its
     * bias and LayerNorm-parameter gradients) instead of launching its own finish kernel: the caller flushes the table once, after the
     * last block of the backward sweep (eoe_red_table_flush), in one or two launches for the whole tower instead of one 17-us launch per
     * block.  The caller then keeps every block's red_scratch untouched until the flush (one scratch per block, not two alternating), and
     * reads the bias / LayerNorm gradients only behind it.  NULL: the block finishes its own, as before. */
    struct eoe_red_table* red_table;
} eoe_vit_block_bwd_args;

/* floats: partial rows of the fc dgrad GEMM's fused column sums [ceil(n*L/64)][4D] + of the two LayerNorm backwards + the attention
 * backward's per-image in_proj bias sums [n][3D] (separate pieces: one kernel finishes all four at the end of the block) */
#define EOE_VIT_RED_SCRATCH(n, L, D) \
    ((size_t)(((size_t)(n) * (L) + 63) / 64) * 4 * (D) + 2 * EOE_LN_SCRATCH(D) + (size_t)(n) * 3 * (D) + (size_t)EOE_CAST_COLSUM_PARTIALS * (D))
int eoe_vit_block_fwd(const eoe_vit_block_fwd_args* a, void* stream);
int eoe_vit_block_bwd(const eoe_vit_block_bwd_args* a, void* stream);
int eoe_vit_side_join(void* stream);      /* see eoe_vit_block_bwd_args.async_wgrad */

/* ------------------------------------------------------------------------------------------------------
 * CNN backbones (cnn.py:44-86 CNN32; resnet.py:25-152 WideResNet): convolution = im2col + eoe_gemm_nt (forward),
 * eoe_gemm_tn (wgrad), eoe_gemm_nt + col2im (dgrad); BatchNorm(train statistics) + (Leaky)ReLU + MaxPool2 fused.
 * Activations between layers: 16-bit NHWC; conv outputs before BatchNorm: fp32 [n*H*W, C].
 * ---------------------------------------------------------------------------------------------------- */
/* patches [n*Ho*Wo, Kp] 16-bit, column = (ky*kw+kx)*cin + c, zero padded to Kp (multiple of 64); Ho = (H+2*pad-kh)/stride+1.
 * x_kind: 1 = the fp32 NCHW input image batch (optional per-channel normalise as ad_trainer.py:413-425),
 * 0 = a 16-bit NHWC activation, 2 = an fp32 NHWC activation. */
int eoe_im2col(const void* x, int x_kind, const float* mean, const float* std, void* out, int n, int cin, int H, int W,
               int kh, int kw, int stride, int pad, int Kp, int dtype, void* stream);
/* dx fp32 NHWC [n,H,W,C] = transpose of im2col applied to dpatches 16-bit [n*Ho*Wo, Kp] */
int eoe_col2im(const void* dpatches, float* dx, int n, int C, int H, int W, int kh, int kw, int stride, int pad, int Kp,
               int dtype, int accumulate /* dx += (a shortcut gradient already in dx) */, void* stream);
/* conv weight fp32 [cout,cin,kh,kw] -> 16-bit [cout,Kp] (patch column order), optionally its transpose [Kp,cout] and the
 * operand of the implicit stride-1 dgrad [cin, (kh*kw reversed) x cout] (dx = conv of dy with the flipped kernel);
 * and the inverse reorder of the fp32 weight gradient [cout,Kp] (or, transposed, [kh*kw*cpad, cout]) -> [cout,cin,kh,kw].
 * cpad >= cin = channels per tap in the column order (8 for the channel-padded NHWC8 image of a 3-channel first layer). */
typedef struct {      /* one eoe_conv_pack_weight call */
    const float* w;
    void *w16, *w16t, *w16d;
    int32_t cout, cin, cpad, kh, kw, Kp;
} eoe_conv_pack_job;
/* `count` weight packs in one launch per 32 jobs: the per-step refresh of all 16-bit conv weight copies of a model */
int eoe_conv_pack_weight_multi(const eoe_conv_pack_job* jobs, int count, int dtype, void* stream);
int eoe_conv_pack_weight(const float* w, void* w16, void* w16t, void* w16d, int cout, int cin, int cpad, int kh, int kw,
                         int Kp, int dtype, void* stream);
int eoe_conv_unpack_wgrad(const float* g, float* dw, int cout, int cin, int cpad, int kh, int kw, int Kp, int transposed,
                          int accumulate, void* stream);
/* 3-channel first layer without a materialised patch matrix (gather = 2 of eoe_gemm_args): the image batch fp32 NCHW
 * [n,3,H,W] (optional per-channel normalise, ad_trainer.py:413-425) becomes a 16-bit NHWC4 tensor [n,Hp,Wp,4] with the
 * convolution's zero padding made physical (pixel (h,w) at (h+pad, w+pad)); the GEMM's k axis is ky*32 + kx*4 + c with
 * K = ceil(kh/2)*64 (kw <= 8, even stride), weights packed to match, wgrad [K, cout] unpacked to [cout,3,kh,kw].
 * Needs Hp >= (Ho-1)*stride + 2*ceil(kh/2), Wp >= (Wo-1)*stride + 8, Wp even. */
int eoe_stem_pack_image(const float* x, const float* mean, const float* std, void* out, int n, int H, int W, int Hp, int Wp,
                        int pad, int cpad /* 4, or 8 = NHWC8 for gather 1 with C = 8 (any stride; pad 0: no physical border) */,
                        int dtype, void* stream);
int eoe_stem_pack_weight(const float* w, void* w16, int cout, int kh, int kw, int dtype, void* stream);
int eoe_stem_unpack_wgrad(const float* g, float* dw, int cout, int kh, int kw, void* stream);

/* batch statistics of y fp32 [M,C]: stats[0..C) = mean, stats[C..2C) = 1/sqrt(var+eps) (biased var); training updates
 * running_mean/var (momentum, unbiased var) and num_batches_tracked as nn.BatchNorm does (cnn.py:57-66); eval reads
 * the running buffers.  sums_scratch: EOE_BN_SCRATCH(C) floats (per-workgroup partial sums: no atomics). */
#define EOE_BN_PARTIALS 1024
#define EOE_BN_SCRATCH(C) ((EOE_BN_PARTIALS + 3) * 2 * (C))
/* Synchronised BatchNorm for data-parallel training (the reference is single-device: its BatchNorm sees the whole batch,
 * cnn.py:57-66, resnet.py:37-41, cbam.py:74).  With a hook registered, every training-mode BatchNorm reduction point of this
 * library -- (sum, sum of squares, row count) of the forward statistics as 2C+1 doubles, (sum g, sum g*xhat, row count) of the
 * backward as 2C+1 floats -- is handed to `fn` for an in-place SUM over the ranks, on `stream`, before it is used; the row count
 * travels with the sums, so ranks may hold different numbers of rows.  fn == NULL switches back to per-rank statistics.
 * Process-wide, like eoe_set_option; not capturable in a HIP graph.  Returns 0 / the hook returns 0 on success. */
typedef int (*eoe_allreduce_fn)(void* user, void* buf, int64_t count, int is_f64, void* stream);
int eoe_set_bn_sync(eoe_allreduce_fn fn, void* user);
/* the same statistics from the partial rows [R][2][C] an eoe_gemm_nt call with `colstats` left in its workspace (R = ceil(M/64)) */
int eoe_bn_stats_partials(const float* part, int R, float* sums_scratch, float* stats, float* running_mean, float* running_var,
                          int64_t* num_batches_tracked, int M, int C, float eps, float momentum, void* stream);
int eoe_bn_stats(const float* y, float* sums_scratch, float* stats, float* running_mean, float* running_var,
                 int64_t* num_batches_tracked, int M, int C, float eps, float momentum, int training, void* stream);
/* EOE_Y16, OR-ed into the `dtype` argument of the four eoe_bn_act_* entry points below: y is not fp32 but the 16-bit (that dtype) matrix
 * an eoe_gemm_nt call with out_f32 = 0 and `colstats` wrote -- the batch statistics still come from the GEMM's fp32 accumulators; the
 * three passes that read y (forward, backward reduce, backward apply) move half the bytes.  16-bit fast mode only. */
#define EOE_Y16 0x100
/* out = maxpool_{pool}(act(bn(y))), act(z) = z > 0 ? z : slope*z (slope 0.01 = LeakyReLU of cnn.py, 0 = ReLU of
 * resnet.py, 1 = identity); y fp32 [n,H,W,C] (16-bit with EOE_Y16); out 16-bit NHWC (or fp32 if out_f32; or the reference's NCHW flatten
 * order [n, C*(H/p)*(W/p)] if nchw_flat, cnn.py:83); out16 (optional, with an fp32 NHWC out): a 16-bit copy of out, the
 * operand of the next convolution's implicit GEMM */
int eoe_bn_act_pool_fwd(const void* y, const float* stats, const float* gamma, const float* beta, void* out, void* out16,
                        int n, int H, int W, int C, int pool, int nchw_flat, int out_f32, float slope, int dtype, void* stream);
/* backward of the above: dout fp32 (layout of `out`) -> dy [n*H*W, C] 16-bit (dY operand of the conv wgrad/dgrad), or
 * fp32 if dy_f32; dgamma, dbeta.  red_scratch: EOE_BN_SCRATCH(C) floats. */
/* out[c] (+= if accumulate) = sum over the rows of an fp32 matrix [rows, C], C % 4 == 0: a convolution's bias gradient in the exact-fp32
 * mode; red_scratch: EOE_BN_SCRATCH(C) floats; fixed summation order, no atomics */
int eoe_colsum_f32(const float* x, float* out, float* red_scratch, int rows, int C, int accumulate, void* stream);
int eoe_bn_act_pool_bwd(const void* y, const float* stats, const float* gamma, const float* beta, const float* dout,
                        float* red_scratch, void* dy, int dy_f32, float* dgamma, float* dbeta, int n, int H, int W, int C,
                        int pool, int nchw_flat, int training, int accumulate, float slope, int dtype, void* stream);

/* BatchNorm + activation + OVERLAPPING MaxPool2d(k, stride, pad) in one pass (the stem of resnet.py:93-96: the
 * 112x112x64 post-ReLU activation is never written): out fp32 [n,Ho,Wo,C], optional 16-bit copy, idx = winning tap;
 * backward: dout fp32 [n,Ho,Wo,C] -> dy 16-bit [n*H*W, C] (gather form; fp32 with dtype = EOE_F32), dgamma, dbeta;
 * red_scratch EOE_BN_SCRATCH(C). */
int eoe_bn_act_maxpool_fwd(const void* y, const float* stats, const float* gamma, const float* beta, float* out, void* out16,
                           uint8_t* idx, int n, int H, int W, int C, int k, int stride, int pad, float slope, int dtype,
                           void* stream);
int eoe_bn_act_maxpool_bwd(const void* y, const float* stats, const float* gamma, const float* beta, const float* dout,
                           const uint8_t* idx, float* red_scratch, void* dy, float* dgamma, float* dbeta, int n, int H, int W,
                           int C, int k, int stride, int pad, int training, float slope, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Resize and ColorJitter of the input pipeline (N1; main/train_imagenet.py:30-31 Resize(256), main/train_clip_imagenet.py:28-29
 * Resize((256,256)) + ColorJitter(0.01 x 4), main/train_cifar.py:32, CLIP's clip_official/clip/clip.py:58-65 bicubic Resize(224)).
 * The reference runs them on PIL images (torchvision -> Pillow); these reproduce Pillow's 8-bit arithmetic byte for byte.
 *   eoe_resize_coeffs   HOST helper: Pillow's antialiased filter taps for one axis (Resample.c), 22-bit fixed point.
 *                       bounds int32 [out_size][2] = (first source index, taps); kk int32 [out_size][ksize_cap].
 *                       bounds == NULL or kk == NULL: only *ksize_out (taps per output) is returned.
 *   eoe_resize_pass_u8  one separable pass over uint8 data viewed as [outer, axis_in, inner] -> [outer, axis_out, inner]
 *                       (bounds / kk on the DEVICE): horizontal pass outer = n*H, inner = 3; vertical pass outer = n,
 *                       inner = Wo*3.  Image.resize = horizontal, then vertical, each skipped when the size is unchanged.
 *   eoe_color_jitter_u8 dst[slot] = ColorJitter of src[idx[slot]] (uint8 NHWC [., H, W, 3]): factors fp32 [n][4] =
 *                       (brightness, contrast, saturation around 1; hue around 0) as torchvision samples them, order int32
 *                       [n][4] = the permutation in which the four ops (0..3) are applied; an entry outside 0..3 skips.
 *                       gray_mean_scratch: n int32.
 * ---------------------------------------------------------------------------------------------------- */
enum { EOE_RESIZE_BILINEAR = 2, EOE_RESIZE_BICUBIC = 3 };   /* PIL.Image.BILINEAR / BICUBIC */
int eoe_resize_coeffs(int in_size, int out_size, int filter, int32_t* bounds, int32_t* kk, int ksize_cap, int* ksize_out);
int eoe_resize_pass_u8(const uint8_t* src, uint8_t* dst, const int32_t* bounds, const int32_t* kk, int ksize, int64_t outer,
                       int axis_in, int axis_out, int inner, void* stream);
int eoe_color_jitter_u8(const uint8_t* src, int64_t n_src, const int32_t* idx, const float* factors, const int32_t* order,
                        int32_t* gray_mean_scratch, uint8_t* dst, int n, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Data-parallel exchange (new -- the reference is single-device, main/__init__.py:110-114): gradient
 * SUM all-reduce and score / label all-gather over RCCL on xGMI, one process per GPU.  RCCL is bound at run time (the librccl
 * already in the process, else the system one); without it these return EOE_ERR_UNSUPPORTED and nothing else is affected.
 * Rank 0 makes the id (eoe_comm_unique_id) and hands its EOE_COMM_ID_BYTES bytes to the other ranks by any out-of-band channel
 * (the launcher's store); every rank then calls eoe_comm_init.  The `_async` calls run on the communicator's own side stream
 * behind everything `after_stream` holds at the time of the call (overlap with the rest of backward); eoe_comm_join makes
 * `stream` wait for all collectives issued so far.  Buffers are caller-owned device memory.
 * dtype: EOE_F32 | EOE_F16 | EOE_BF16 | EOE_COMM_I64.  algo: EOE_COMM_ALGO_RING = RCCL's all-reduce; EOE_COMM_ALGO_RS_AG =
 * in-place reduce-scatter + all-gather (count %% world == 0, else the former): on a fully connected xGMI node every rank
 * exchanges 1/world of the buffer with every peer at once instead of walking a ring.
 * ---------------------------------------------------------------------------------------------------- */
#define EOE_COMM_ID_BYTES 128
enum { EOE_COMM_I64 = 8 };
enum { EOE_COMM_ALGO_RING = 0, EOE_COMM_ALGO_RS_AG = 1 };
typedef struct eoe_comm* eoe_comm_t;
int eoe_comm_unique_id(void* id_out);
int eoe_comm_init(const void* id, int rank, int world, int device, eoe_comm_t* out);
int eoe_comm_destroy(eoe_comm_t comm);
int eoe_comm_info(eoe_comm_t comm, int* rank, int* world);
int eoe_comm_allreduce_sum_async(eoe_comm_t comm, void* buf, int64_t count, int dtype, int algo, void* after_stream);
int eoe_comm_allgather_async(eoe_comm_t comm, const void* send, void* recv, int64_t send_count, int dtype, void* after_stream);
int eoe_comm_join(eoe_comm_t comm, void* stream);
/* synchronised BatchNorm over this communicator without leaving the library: registers (enable != 0) or clears an eoe_set_bn_sync
 * hook that sums the BatchNorm reduction buffers with ncclAllReduce IN the stream the BatchNorm kernels run on (no side stream: the
 * very next kernel consumes the sums).  The sums use a SECOND RCCL communicator over the same ranks (bucket all-reduces of the first one
 * may be in flight on the side stream); `bn_id` is its id -- EOE_COMM_ID_BYTES bytes from eoe_comm_unique_id() on rank 0, identical on
 * every rank, handed over like the first one's (NULL once the communicator has it, and with enable == 0).  The library allocates no
 * device memory and synchronises nothing here.  The communicator must outlive the registration. */
int eoe_comm_sync_bn(eoe_comm_t comm, int enable, const void* bn_id);

/* ------------------------------------------------------------------------------------------------------
 * Parity mode (SURVEY.md section 7 "Hard parts", section 8d "Parity run"): the convolutions / linear layers of the BatchNorm
 * encoders (cnn.py:73-86, resnet.py:85-149) in plain fp32 -- fp32 activations and fp32 master weights as operands, one fp32 FMA
 * per product in a fixed order, no 16-bit rounding.  Since round 3 on the fp32 matrix cores (v_mfma_f32_16x16x4_f32; the
 * register-tiled SGEMM on the vector ALUs stays selectable: option "parity_flags" bit 0): the conformant mode of BASELINE configs 1-3
 * and the instrument that tells the fast path's 16-bit operand rounding from an implementation difference.
 *   x: fp32 NHWC [n,H,W,C], or the fp32 NCHW image batch if x_nchw (then optional per-channel Normalize (x - mean) / std,
 *   ad_trainer.py:413-425); w: [cout, C, kh, kw] (the nn.Conv2d parameter itself; a Linear layer is H = W = kh = kw = 1);
 *   y / dy: fp32 [n*Ho*Wo, cout]; dx: fp32 NHWC (+= if accumulate); dw: [cout, C, kh, kw] overwritten (slab sums in fixed
 *   order: bitwise reproducible).  geo.C = input channels; geo.Ho / geo.Wo must match the geometry.
 *   workspace (forward / dgrad; may be NULL / 0): with few output tiles and a long reduction (small maps, FC layers) the reduction
 *   is cut into slabs whose partial outputs ([slab][rows][cout] floats) go here and are summed in slab order; a workspace of
 *   k x the output size allows k slabs.  A stride-2 dgrad over an even map runs as four parity-class GEMMs (no work on taps a pixel
 *   never meets); "parity_flags" bit 1 turns that off.
 * ---------------------------------------------------------------------------------------------------- */
/* the 3-channel NCHW image batch, optionally normalised ((x - mean) / std, ad_trainer.py:413-425), as an fp32 NHWC map with 4 channels
 * (the 4th = 0): with it (geo.C = 4, weights zero-padded to [cout, 4, kh, kw]) the first layer runs through the float4 paths of
 * eoe_conv_f32_fwd / _wgrad instead of element-wise gathers */
int eoe_pack_image_nhwc4(const float* x_nchw, const float* mean, const float* stdv, float* out_nhwc4, int n, int H, int W, void* stream);
/* w_kmajor (forward / dgrad; may be NULL): the k-major fp32 copy of w made by eoe_conv_f32_pack_weights (forward: wf, dgrad: wd) -- the
 * kernels then stage the weight tile with 16-byte loads and stores instead of scalar gathers with a stride of kh * kw floats (+5-10 %) */
int eoe_conv_f32_pack_weights(const float* w /* [cout, cin, kh, kw] */, float* wf /* [kh*kw*cin, cout] or NULL */,
                              float* wd /* [kh*kw*cout, cin] or NULL */, int cout, int cin, int kh, int kw, void* stream);
int eoe_conv_f32_fwd(const float* x, int x_nchw, const float* mean, const float* stdv, const float* w, const float* bias, float* y,
                     const eoe_conv_geometry* geo, int cout, void* workspace, size_t workspace_bytes, const float* w_kmajor, void* stream);
int eoe_conv_f32_dgrad(const float* dy, const float* w, float* dx, const eoe_conv_geometry* geo, int cout, int accumulate,
                       void* workspace, size_t workspace_bytes, const float* w_kmajor, void* stream);
size_t eoe_conv_f32_wgrad_workspace(const eoe_conv_geometry* geo, int cout);
int eoe_conv_f32_wgrad(const float* x, int x_nchw, const float* mean, const float* stdv, const float* dy, float* dw,
                       const eoe_conv_geometry* geo, int cout, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * WideResNet + CBAM (resnet.py:85-109,130-149; cbam.py:31-107).  All activations fp32 NHWC.  The 7x7/2 stem,
 * the 3x3 and the 1x1/2 downsample convolutions are eoe_im2col + eoe_gemm_nt (+ eoe_bn_stats / eoe_bn_act_pool_*
 * with slope 0 = ReLU or 1 = none); the pieces below are the HBM-bound rest of a BasicBlock.
 * ---------------------------------------------------------------------------------------------------- */
/* nn.MaxPool2d(k, stride, pad) (resnet.py:96): out [n,Ho,Wo,C]; idx [n,Ho,Wo,C] = winning tap ky*k+kx (first maximum) */
int eoe_maxpool_fwd(const float* x, float* out, void* out16 /* optional 16-bit copy */, uint8_t* idx, int n, int H, int W, int C,
                    int k, int stride, int pad, int dtype, void* stream);
int eoe_maxpool_bwd(const float* dout, const uint8_t* idx, float* dx, int n, int H, int W, int C, int k, int stride, int pad,
                    void* stream);

/* ChannelGate (cbam.py:31-66, pool_types avg+max): out = x * sigmoid(mlp(avgpool x) + mlp(maxpool x)),
 * mlp = Linear(C, Ch) -> ReLU -> Linear(Ch, C).  The forward fills the saved tensors the backward reads. */
typedef struct {
    const float* x;       /* [n, HW, C] */
    float* out;           /* [n, HW, C] (forward only) */
    const float* w1;      /* mlp.1.weight [Ch, C] */
    const float* b1;      /* mlp.1.bias   [Ch]    */
    const float* w2;      /* mlp.3.weight [C, Ch] */
    const float* b2;      /* mlp.3.bias   [C]     */
    float* pooled;        /* [n, 2, C]  avg, max       (saved) */
    int* argmax;          /* [n, C]     position of the max (saved) */
    float* hidden;        /* [n, 2, Ch] post-ReLU      (saved) */
    float* scale;         /* [n, C]     sigmoid output (saved) */
    int n, HW, C, Ch;
} eoe_cgate_args;
typedef struct {
    eoe_cgate_args f;
    const float* dout;    /* [n, HW, C] */
    float* dx;            /* [n, HW, C] */
    float* dscale;        /* [n, C]     scratch */
    float* dpooled;       /* [n, 2, C]  scratch */
    float* dhidden;       /* [n, 2, Ch] scratch */
    float *dw1, *db1, *dw2, *db2;      /* written (not accumulated) */
} eoe_cgate_bwd_args;
int eoe_cgate_fwd(const eoe_cgate_args* a, void* stream);
int eoe_cgate_bwd(const eoe_cgate_bwd_args* a, void* stream);

/* SpatialGate (cbam.py:76-92): out = x * sigmoid(bn(conv7x7([max_c x, mean_c x]))), conv 2->1 without bias, pad 3,
 * BatchNorm2d(1) with the module's eps / momentum (cbam.py:14: momentum 0.01) and running buffers. */
typedef struct {
    const float* x;       /* [n, H, W, C] */
    float* out;           /* [n, H, W, C] (forward only) */
    const float* w;       /* spatial.conv.weight [1, 2, 7, 7] */
    const float* gamma;   /* spatial.bn.weight [1] or NULL */
    const float* beta;    /* spatial.bn.bias   [1] or NULL */
    float* running_mean;  /* [1] or NULL */
    float* running_var;   /* [1] or NULL */
    int64_t* num_batches_tracked;   /* [1] or NULL */
    float* comp;          /* [n, H, W, 2] channel max, channel mean (saved) */
    int* argmax;          /* [n, H, W]    channel of the max        (saved) */
    float* z;             /* [n, H, W]    conv output               (saved) */
    float* stats;         /* [2]          mean, rstd used           (saved) */
    float* scale;         /* [n, H, W]    sigmoid output            (saved) */
    float* sums;          /* EOE_BN_SCRATCH(1) floats scratch */
    int n, H, W, C;
    float eps, momentum;
    int training;         /* 1: batch statistics (+ running update), 0: running statistics */
    const float* res;     /* optional (forward): residual [n, H, W, C]; then out = relu(x * scale + res), the block's junction
                           * (resnet.py:143-147) fused into the gate's last pass */
    void* out16;          /* optional with res: 16-bit copy of out (operand of the next convolution) */
    int dtype;            /* EOE_F16 | EOE_BF16 of out16 */
} eoe_sgate_args;
#define EOE_SGATE_PARTIALS 512
#define EOE_SGATE_RED (2 + 2 * EOE_SGATE_PARTIALS)
typedef struct {
    eoe_sgate_args f;
    const float* dout;    /* [n, H, W, C] */
    float* dx;            /* [n, H, W, C] */
    float* dscale;        /* [n, H, W]    scratch */
    float* dcomp;         /* [n, H, W, 2] scratch */
    float* red;           /* [EOE_SGATE_RED] scratch: per-workgroup partial sums of the 1-channel BatchNorm backward */
    float* dw;            /* [1, 2, 7, 7] written */
    float* dgamma;        /* [1] or NULL */
    float* dbeta;         /* [1] or NULL */
    float* wpart;         /* [n, 98] scratch: per-image partial sums of dw (added up in a fixed order: no atomics) */
} eoe_sgate_bwd_args;
int eoe_sgate_fwd(const eoe_sgate_args* a, void* stream);
int eoe_sgate_bwd(const eoe_sgate_bwd_args* a, void* stream);
/* A BasicBlock's tail out = relu(SpatialGate(ChannelGate(x)) + res) (resnet.py:143-147, cbam.py:100-106) as ONE unit: the channel-gated
 * tensor is never written (every consumer multiplies x by the channel scale on the fly -- the same fp32 product), the junction's ReLU mask
 * is applied where the gradient is first read, and the spatial gate's input gradient feeds the channel gate's reduction and its final
 * pass without being stored: 22 + 32 bytes per activation element instead of 30 + 44.  cg / cb: the channel gate's argument blocks
 * (cg->out unused); sg / sb: the spatial gate's (sg->x, sb->dout, sb->dx unused; sg->res, sg->out, sg->out16 required / as in eoe_sgate_fwd).
 * backward: dout = gradient at the block's output, out = that output (the mask); g (written) = the residual branch's gradient,
 * cb->dx = the gradient of x; cb->dscale must hold EOE_CBAM_DSCALE_SLICES slices of [n, C] and sb->dcomp FOUR floats per pixel here (scale, dcomp0, dcomp1, argmax bits: one 16-byte record). */
#define EOE_CBAM_DSCALE_SLICES 8
int eoe_cbam_junction_fwd(const eoe_cgate_args* cg, const eoe_sgate_args* sg, void* stream);
int eoe_cbam_junction_bwd(const eoe_cgate_bwd_args* cb, const eoe_sgate_bwd_args* sb, const float* dout, const float* out, float* g,
                          void* stream);

/* out = relu(a + b) (resnet.py:146-147); g = dout * [out > 0] (the gradient of both summands) */
int eoe_add_relu_fwd(const float* a, const float* b, float* out, void* out16 /* optional 16-bit copy */, int dtype,
                     int64_t count, void* stream);
int eoe_relu_bwd(const float* dout, const float* out, float* g, int64_t count, void* stream);
/* nn.AvgPool2d over the whole HW grid (resnet.py:38,104): pooled_scratch [n,2,C] receives (mean, max), the mean is
 * pooled_scratch[:,0,:]; backward dx[n,hw,c] = dout[n,c] / HW */
int eoe_avgpool_fwd(const float* x, float* pooled_scratch, int* argmax_scratch, int n, int HW, int C, void* stream);
int eoe_avgpool_bwd(const float* dout, float* dx, int n, int HW, int C, void* stream);

/* the other objectives of the reference's TRAINER registry (training/__init__.py:8-11), same
 * conventions as eoe_hsc_* / eoe_bce_* (loss[0] = inv_count * sum of the per-sample losses; gscale = upstream gradient):
 *   DSAD  (dsad.py:17-21):  loss_i = |f|^2 if label == nominal else 1 / (|f|^2 + 1e-9);  score = the HSC score (eoe_hsc_score)
 *   DSVDD (dsvdd.py:24-27): loss_i = score_i = |f - center|^2 (center fp32 [d], from prepare_metric dsvdd.py:10-22)
 *   focal (focal.py:11-36): b = bce_with_logits(x, y), pt = clamp(exp(-b), eps, 1 - eps), loss_i = (1 - pt)^gamma * b;
 *                           scores = sigmoid(x) (1 - sigmoid if nominal_label != 0) */
/* Epoch-tail metrics on the device (ad_trainer.py:452-455,517-521): out[0] = ROC AUC (tie-averaged rank
 * statistic = sklearn's trapezoidal area), out[1] = average precision (sklearn's step-wise sum), both fp64, NaN when a class is
 * missing; scores fp32 [n], labels int64 [n] (positive_label = the anomalous label, 1); computed from exact integer pair counts
 * (n^2 compares, no sort).  scratch: EOE_AUC_SCRATCH_BYTES(n) bytes. */
#define EOE_AUC_SCRATCH_BYTES(n) ((size_t)(((n) + 255) / 256) * 24)
int eoe_auc_ap(const float* scores, const int64_t* labels, int64_t positive_label, double* out, void* scratch, int n, void* stream);

/* CLIP text-prompt objective (training/clip.py:66-103): f fp32 [n, d] image features, text fp32 [T, d]
 * (2 <= T <= 64; the frozen, l2-normalised text features prepare_metric returns, clip.py:50-64), l = 100 * f/|f| . text^T;
 *   loss_i = -log_softmax(l)[pick]: pick = T-1 for the anomalous label (1 - nominal), 0 for the nominal label (one_vs_rest) or
 *   the arg max over j < T-1 (leave_one_out); other labels contribute 0;  scores = softmax(l)[:, T-1] (any of scores / loss NULL).
 * eoe_clip_score: the same score for text rows the caller has l2-normalised (compute_anomaly_score, clip.py:66-79). */
int eoe_clip_fwd(const float* f, const float* text, const int64_t* labels, int64_t nominal_label, int leave_one_out, float* loss,
                 float* scores, float* losses, int n, int d, int T, float inv_count, void* stream);
int eoe_clip_bwd(const float* f, const float* text, const int64_t* labels, int64_t nominal_label, int leave_one_out,
                 const float* gscale, float* df, int n, int d, int T, float inv_count, void* stream);
int eoe_clip_score(const float* f, const float* text, float* scores, int n, int d, int T, void* stream);
int eoe_dsad_fwd(const float* f, const int64_t* labels, int64_t nominal_label, float* loss, float* losses, int n, int d,
                 float inv_count, void* stream);
int eoe_dsad_bwd(const float* f, const int64_t* labels, int64_t nominal_label, const float* gscale, float* df, int n, int d,
                 float inv_count, void* stream);
int eoe_dsvdd_fwd(const float* f, const float* center, float* loss, float* dists, int n, int d, float inv_count, void* stream);
int eoe_dsvdd_bwd(const float* f, const float* center, const float* gscale, float* df, int n, int d, float inv_count,
                  void* stream);
int eoe_focal_fwd(const float* x, const int64_t* labels, int64_t nominal_label, float* loss, float* scores, float* losses,
                  int n, float inv_count, float gamma, float eps, void* stream);
int eoe_focal_bwd(const float* x, const int64_t* labels, const float* gscale, float* dx, int n, float inv_count, float gamma,
                  float eps, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * on-device input pipeline: gather + RandomCrop (zero padding) + RandomHorizontalFlip +
 * ToTensor + Gaussian noise + per-channel Normalize in one pass over a uint8 NHWC image set resident in HBM
 * (main/train_cifar.py:31-38, main/train_clip_imagenet.py:27-36, training/ad_trainer.py:413-425).
 *   src    uint8 [n_src, Hs, Ws, 3];   params int32 [n, 4] = (source index, crop top, crop left, flip) per batch slot
 *          (top / left may be negative or reach past the image: RandomCrop(padding) with fill 0)
 *   out    fp32 NCHW [n, 3, Ho, Wo] = ((src / 255 + noise_std * N(0,1)) - mean[c]) / std[c]   (mean/std NULL: none)
 *   flip_first 1: flip the source then crop (CIFAR order), 0: crop then flip (CLIP order)
 *   noise  element e = (c*Ho + y)*Wo + x of slot b draws Box-Muller from splitmix64(seed*2^40 + b*2^18 + e)
 * ---------------------------------------------------------------------------------------------------- */
int eoe_augment_batch(const uint8_t* src, int64_t n_src, int Hs, int Ws, const int32_t* params, const float* mean,
                      const float* std, float* out, int n, int Ho, int Wo, int flip_first, float noise_std, uint64_t seed,
                      void* stream);

/* ------------------------------------------------------------------------------------------------------
 * in-library kernel timing (used by bench.py for the roofline line): while enabled, every entry point brackets
 * its kernel launches with hipEvents on the stream it launches on and records the algorithmic flops / bytes.
 * eoe_prof_collect synchronises the recorded events and aggregates them per kernel name.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct {
    char name[32];
    int64_t launches;
    double total_ms;   /* sum of per-launch durations */
    double flops;      /* sum of algorithmic floating-point operations (2 per MAC) */
    double bytes;      /* sum of algorithmic HBM bytes (inputs read once + outputs written once) */
} eoe_prof_entry;
int eoe_prof_enable(int on);

/* Box calibration probes (bench.py prints them in its line so that numbers from different boxes can be compared; not on the product path):
 * a bare v_mfma_f32_16x16x32_f16 loop -- `blocks` workgroups of 4 waves, 8 independent accumulators each, `iters` rounds of 8 MFMAs per wave
 * (2 * 16*16*32 * 8 * iters * 4 * blocks FLOP), out: blocks * 256 floats or NULL -- and a 16-byte-per-lane streaming copy. */
int eoe_probe_mfma_f16(float* out, int iters, int blocks, void* stream);
int eoe_probe_copy(void* dst, const void* src, int64_t bytes, void* stream);

/* tuning switches for A/B measurements inside one process ("nt_flags": see gemm.hip) */
int eoe_set_option(const char* name, int value);
int eoe_get_option(const char* name, int* value);    /* the current value (callers that change a switch restore what they found) */
/* diagnostics only (EOE_GEMM_STAMP=1): in-kernel s_memtime stamps of the last eoe_gemm_nt launch, 16 words per workgroup */
int eoe_debug_gemm_stamps(unsigned long long* out, int n_words);
int eoe_prof_collect(eoe_prof_entry* out, int max_entries, int* n_out);

#ifdef __cplusplus
}
#endif
#endif /* EOE_HIP_H */
