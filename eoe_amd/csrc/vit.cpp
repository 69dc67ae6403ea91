// Fused launch chains for one ViT residual block (clip/model.py:167-188 and its backward): the host side only
// enqueues kernels on the caller's stream; there is no host synchronisation and no allocation in here.
#include "common.h"
#include <mutex>
#include <stdlib.h>

#define TRY(expr)                   \
    do {                            \
        int rc__ = (expr);          \
        if (rc__ != 0) return rc__; \
    } while (0)

extern int g_tn_flags;          // gemm_tn.hip
bool eoe_tn_streamk_would_run(const eoe_gemm_args* args, int count, void* stream);   // gemm_tn.hip: the stream-K precondition
int eoe_red_table_append(eoe_red_table* t, EoeRedJobs* jobs, void* stream);          // elementwise.hip
int g_vit_side_stream = 1;      // eoe_set_option("vit_side_stream", 0|1)

namespace {

// Low-priority side stream of the block backward: the grouped wgrad GEMM runs 216 full-CU workgroups and leaves 40 of the 256 CUs
// idle for its whole duration (~213 us); LayerNorm-1's backward (HBM-bound, ~31 us on the whole chip, independent of the wgrad) is
// launched on this stream next to it and fills those CUs (measured: 12 of its 31 us hidden, -0.14 ms per step).  Eager launches only.
struct SideStream {
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr, wgrad_done[2] = {nullptr, nullptr};
    bool ok = false;
    // asynchronous weight-gradient launches of the current backward sweep: launch number `seq` records wgrad_done[seq & 1]; `recorded[e]`: that
    // event stands for a launch no compute stream has been ordered behind yet
    unsigned seq = 0;
    bool recorded[2] = {false, false};
};
// per-device handle table (the only mutable state this file keeps), created once per device under a mutex
SideStream* side_stream(hipStream_t main) {
    static SideStream per_dev[16];
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    SideStream& ss = per_dev[dev];
    // not inside a stream capture: a captured fork / join made the replayed ViT step 19 ms instead of 12
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(main, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;
    if (!ss.ok) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);          // lo = numerically largest = lowest priority
        if (hipStreamCreateWithPriority(&ss.s, hipStreamNonBlocking, lo) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ss.join, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ss.wgrad_done[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ss.wgrad_done[1], hipEventDisableTiming) != hipSuccess)
            return nullptr;
        ss.ok = true;
    }
    return &ss;
}

eoe_gemm_args gemm(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, int lda, int ldb,
                   int ldc, int dtype) {
    eoe_gemm_args g = {};
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.aux = nullptr; g.aux_out = nullptr; g.colsum = nullptr;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldaux = 0;
    g.dtype = dtype; g.epilogue = EOE_EPI_NONE; g.out_f32 = 0; g.accumulate = 0; g.alpha = 1.0f;
    return g;
}
// the same with the block's stream-K workspace attached
eoe_gemm_args gemm(const eoe_vit_block_fwd_args* a, const void* A, const void* B, void* C, const float* bias, int M, int N, int K, int lda,
                   int ldb, int ldc, int dtype) {
    eoe_gemm_args g = gemm(A, B, C, bias, M, N, K, lda, ldb, ldc, dtype);
    g.sk_workspace = a->nt_sk_workspace; g.sk_workspace_bytes = a->nt_sk_workspace ? a->nt_sk_workspace_bytes : 0;
    return g;
}

int check_fwd(const eoe_vit_block_fwd_args* a) {
    if (!a) return eoe_set_error(EOE_ERR_ARG, "vit_block: null args");
    if (a->n <= 0 || a->L <= 0 || a->L > 64 || a->heads <= 0 || a->D != a->heads * 64)
        return eoe_set_error(EOE_ERR_ARG, "vit_block: unsupported shape n=%d L=%d D=%d heads=%d (need D = 64*heads, L <= 64)",
                             a->n, a->L, a->D, a->heads);
    if (!a->ln1_g || !a->ln1_b || !a->ln2_g || !a->ln2_b || !a->b_in || !a->b_out || !a->b_fc || !a->b_proj || !a->w_in ||
        !a->w_out || !a->w_fc || !a->w_proj || !a->x_in || !a->x_mid || !a->x_out || !a->xn1 || !a->qkv || !a->att ||
        !a->xn2 || !a->hact || !a->stats1 || !a->stats2)      // hpre is optional: forward-only callers pass NULL
        return eoe_set_error(EOE_ERR_ARG, "vit_block: null pointer in arguments");
    return 0;
}

}  // namespace

extern "C" int eoe_vit_block_fwd(const eoe_vit_block_fwd_args* a, void* stream) {
    TRY(check_fwd(a));
    const int M = a->n * a->L, D = a->D, H = 4 * a->D, dt = a->dtype;
    // x_mid = x_in + out_proj(attn(ln_1(x_in)))
    TRY(eoe_layernorm_fwd(a->x_in, D, a->ln1_g, a->ln1_b, a->xn1, a->stats1, M, D, a->eps, dt, 0, stream));
    eoe_gemm_args g = gemm(a, a->xn1, a->w_in, a->qkv, a->b_in, M, 3 * D, D, D, D, 3 * D, dt);
    TRY(eoe_gemm_nt(&g, stream));
    TRY(eoe_attn_fwd(a->qkv, a->att, a->n, a->L, a->heads, dt, stream));
    // (cls_only: from here on only the class-token rows -- row i*L of image i, gathered by the row strides of the GEMM's A operand
    //  and of its residual input; everything downstream is a dense [n, ...] matrix)
    const int Mo = a->cls_only ? a->n : M, ldrow = a->cls_only ? a->L * D : D;
    g = gemm(a, a->att, a->w_out, a->x_mid, a->b_out, Mo, D, D, ldrow, D, D, dt);
    g.epilogue = EOE_EPI_RESIDUAL; g.aux = a->x_in; g.ldaux = ldrow; g.out_f32 = 1;
    TRY(eoe_gemm_nt(&g, stream));
    // x_out = x_mid + c_proj(quick_gelu(c_fc(ln_2(x_mid))))
    TRY(eoe_layernorm_fwd(a->x_mid, D, a->ln2_g, a->ln2_b, a->xn2, a->stats2, Mo, D, a->eps, dt, 0, stream));
    g = gemm(a, a->xn2, a->w_fc, a->hact, a->b_fc, Mo, H, D, D, D, H, dt);
    g.epilogue = EOE_EPI_GELU; g.aux_out = a->hpre;
    TRY(eoe_gemm_nt(&g, stream));
    g = gemm(a, a->hact, a->w_proj, a->x_out, a->b_proj, Mo, D, H, H, H, D, dt);
    g.epilogue = EOE_EPI_RESIDUAL; g.aux = a->x_mid; g.ldaux = D; g.out_f32 = 1;
    g.split_k = a->cls_only ? 1 : 0;          // n x D x 4D: twelve tiles behind 48 k-tiles each (eoe_hip.h, eoe_gemm_args.split_k)
    TRY(eoe_gemm_nt(&g, stream));
    return 0;
}

extern "C" int eoe_vit_block_bwd(const eoe_vit_block_bwd_args* b, void* stream) {
    if (!b) return eoe_set_error(EOE_ERR_ARG, "vit_block_bwd: null args");
    const eoe_vit_block_fwd_args* a = &b->f;
    TRY(check_fwd(a));
    if (!a->hpre || !a->w_in_t || !a->w_out_t || !a->w_fc_t || !a->w_proj_t || !b->dx_out || !b->dx_in || !b->g_ln1_g || !b->g_ln1_b ||
        !b->g_ln2_g || !b->g_ln2_b || !b->g_b_in || !b->g_b_out || !b->g_b_fc || !b->g_b_proj || !b->g_w_in || !b->g_w_out ||
        !b->g_w_fc || !b->g_w_proj || !b->d16_a || !b->d16_b || !b->d16_c || !b->dh || !b->dqkv || !b->dx_mid)
        return eoe_set_error(EOE_ERR_ARG, "vit_block_bwd: null pointer in arguments");
    const int M = a->n * a->L, D = a->D, H = 4 * a->D, dt = a->dtype, acc = b->accumulate;
    hipStream_t s = (hipStream_t)stream;
    // cls_only (eoe_hip.h): dx_out and the MLP / out-projection chain live on the n class-token rows (dense [n, ...] matrices); the
    // attention, the in-projection and LayerNorm-1 see every row, with zeros where the full computation has zero gradients
    const bool cls = a->cls_only != 0;
    const int Mo = cls ? a->n : M, ldrow = cls ? a->L * D : D;
    if (cls && a->L < 4) return eoe_set_error(EOE_ERR_ARG, "vit_block_bwd: cls_only needs L >= 4");
    if (!acc && !b->red_scratch) {
        // without the reduction scratch these gradients are accumulated with fp32 atomics: one zeroing launch
        float* zp[8] = {b->g_ln1_g, b->g_ln1_b, b->g_ln2_g, b->g_ln2_b, b->g_b_fc, b->g_b_out, b->g_b_proj, b->g_b_in};
        const int zn[8] = {D, D, D, D, H, D, D, 3 * D};
        TRY(eoe_zero_multi(zp, zn, 8, stream));
    }
    // the five partial-row column reductions of the block (db_proj, db_fc, LayerNorm-2 parameters + db_out, db_in, LayerNorm-1
    // parameters) each get their own piece of the scratch and are finished together by ONE kernel at the end of the block, which
    // overwrites (or, with `accumulate`, adds to) the eight small gradient vectors: no atomics, no zeroing launch
    EoeRedJobs jobs;
    jobs.count = 0; jobs.tile_start[0] = 0; jobs.overwrite = acc ? 0 : 1;
    struct DeferGuard {
        explicit DeferGuard(EoeRedJobs* j) { eoe_tls_defer = j; }
        ~DeferGuard() { eoe_tls_defer = nullptr; }
    } guard(b->red_scratch ? &jobs : nullptr);
    float* red_fc = b->red_scratch;
    float* red_ln2 = red_fc ? red_fc + (size_t)((M + 63) / 64) * H : nullptr;
    float* red_attn = red_fc ? red_ln2 + EOE_LN_SCRATCH(D) : nullptr;
    float* red_ln1 = red_fc ? red_attn + (size_t)a->n * 3 * D : nullptr;
    float* red_cast = red_fc ? red_ln1 + EOE_LN_SCRATCH(D) : nullptr;
    eoe_gemm_args g, w[4];
    // ---- MLP branch:  x_out = x_mid + c_proj(gelu(c_fc(ln_2(x_mid))))
    // bias gradients are column sums of the dY tensors: fused into the kernels that produce them (fp32 atomics)
    // dY of c_proj (the 16-bit copy of dx_out) + db_proj (its column sums): handed over by the previous call's LayerNorm-1 backward when
    // that call was given `next_d16` (see eoe_hip.h) -- its partial rows are finished by this call's finish kernel --, else one pass here
    const bool handed = !cls && b->in_d16 && b->in_red_scratch && b->red_scratch && b->in_red_scratch != b->red_scratch;
    const void* dy_proj = handed ? b->in_d16 : b->d16_a;
    if (handed) {
        const float* in_ln1 = b->in_red_scratch + (size_t)((M + 63) / 64) * H + EOE_LN_SCRATCH(D) + (size_t)a->n * 3 * D;
        int rows = (M + 7) / 8;
        if (rows > EOE_LN_PARTIALS) rows = EOE_LN_PARTIALS;
        if (!eoe_defer_reduce(in_ln1, rows, 3 * D, D, nullptr, nullptr, b->g_b_proj, 1))
            return eoe_set_error(EOE_ERR_ARG, "vit_block_bwd: could not queue the handed-over column sums");
    } else {
        TRY(eoe_cast_colsum(b->dx_out, b->d16_a, b->g_b_proj, red_cast, Mo, D, dt, 1, stream));
    }
    g = gemm(a, dy_proj, a->w_proj_t, b->dh, nullptr, Mo, H, D, D, D, H, dt);                 // d hact, then * gelu'(hpre)
    g.epilogue = EOE_EPI_GELU_BWD; g.aux = a->hpre; g.ldaux = H;
    if (b->red_scratch) {
        // db_fc = column sums of dh, from the GEMM's epilogue through per-wave partial rows in the scratch (with fp32 atomics
        // instead, the fused sums cost +45 us -- more than a separate 16-us pass over dh)
        g.colsum = b->g_b_fc; g.workspace = red_fc; g.workspace_bytes = (int64_t)EOE_NT_COLSUM_WORKSPACE_BYTES(Mo, H);
        TRY(eoe_gemm_nt(&g, stream));
    } else {
        TRY(eoe_gemm_nt(&g, stream));
        TRY(eoe_colsum(b->dh, H, b->g_b_fc, Mo, H, dt, 1, stream));
    }
    g = gemm(a, b->dh, a->w_fc_t, b->d16_b, nullptr, Mo, D, H, H, H, D, dt);                   // d xn2
    g.split_k = cls ? 1 : 0;
    TRY(eoe_gemm_nt(&g, stream));
    // (cls_only: dx_mid of the class-token rows goes to a dense [n, D] piece behind the dY copy in d16_a -- L >= 4 leaves the room --
    //  and is scattered into the zeroed full dx_mid LayerNorm-1 backward reads; d att likewise through the GEMM's row stride)
    float* dx_mid_o = cls ? (float*)((char*)b->d16_a + (((size_t)Mo * D * 2 + 255) & ~(size_t)255)) : b->dx_mid;
    TRY(eoe_layernorm_bwd(b->d16_b, 0, a->x_mid, D, a->stats2, a->ln2_g, b->dx_out, dx_mid_o, D, b->d16_c, b->g_ln2_g,
                          b->g_ln2_b, b->g_b_out, red_ln2, Mo, D, dt, stream));       // + db_out = colsum(dx_mid)
    if (cls) {
        if (hipMemsetAsync(b->dx_mid, 0, (size_t)M * D * sizeof(float), s) != hipSuccess ||
            hipMemcpy2DAsync(b->dx_mid, (size_t)ldrow * sizeof(float), dx_mid_o, (size_t)D * sizeof(float), (size_t)D * sizeof(float), Mo,
                             hipMemcpyDeviceToDevice, s) != hipSuccess ||
            hipMemsetAsync(b->d16_b, 0, (size_t)M * D * 2, s) != hipSuccess)
            return eoe_set_error(EOE_ERR_LAUNCH, "vit_block_bwd: zero-fill / scatter of the class-token rows failed");
    }
    // ---- attention branch:  x_mid = x_in + out_proj(attn(ln_1(x_in)))
    g = gemm(a, b->d16_c, a->w_out_t, b->d16_b, nullptr, Mo, D, D, D, D, ldrow, dt);           // d att
    TRY(eoe_gemm_nt(&g, stream));
    // + db_in = column sums of dqkv, from the attention kernel's accumulators when the scratch is there
    TRY(eoe_attn_bwd(a->qkv, b->d16_b, b->dqkv, red_attn ? b->g_b_in : nullptr, red_attn, a->n, a->L, a->heads, dt, stream));
    g = gemm(a, b->dqkv, a->w_in_t, b->d16_b, nullptr, M, D, 3 * D, 3 * D, 3 * D, D, dt);      // d xn1
    TRY(eoe_gemm_nt(&g, stream));
    if (!b->red_scratch) TRY(eoe_colsum(b->dqkv, 3 * D, b->g_b_in, M, 3 * D, dt, 1, stream));
    // ---- the four weight gradients of the block in one grouped launch (every dY and X is still live)
    w[0] = gemm(b->dh, a->xn2, b->g_w_fc, nullptr, H, D, Mo, H, D, D, dt);                   // dW_fc[4D,D]   = dh^T xn2
    w[1] = gemm(dy_proj, a->hact, b->g_w_proj, nullptr, D, H, Mo, D, H, H, dt);             // dW_proj[D,4D] = dY^T hact
    w[2] = gemm(b->dqkv, a->xn1, b->g_w_in, nullptr, 3 * D, D, M, 3 * D, D, D, dt);         // dW_in[3D,D]   = dqkv^T xn1
    w[3] = gemm(b->d16_c, a->att, b->g_w_out, nullptr, D, D, Mo, D, ldrow, D, dt);          // dW_out[D,D]   = dmid^T att
    for (int i = 0; i < 4; ++i) { w[i].out_f32 = 1; w[i].accumulate = acc; }
    w[0].workspace = b->tn_workspace; w[0].workspace_bytes = b->tn_workspace ? b->tn_workspace_bytes : 0;     // stream-K partials
    // cls_only: two reduction lengths (n rows for the MLP and the out-projection, n*L for the in-projection) = two launches, in order on
    // one stream (they share the workspace)
    eoe_gemm_args wc[3] = {w[0], w[1], w[3]};
    w[2].workspace = cls ? w[0].workspace : nullptr; w[2].workspace_bytes = cls ? w[0].workspace_bytes : 0;
    auto launch_wgrads = [&](void* st) -> int {
        if (!cls) return eoe_gemm_tn_grouped(w, 4, st);
        TRY(eoe_gemm_tn_grouped(wc, 3, st));
        return eoe_gemm_tn_grouped(&w[2], 1, st);
    };
    // LayerNorm-1 backward next to the wgrad launch on a second stream pays only while that launch leaves CUs idle (216 tiles on 256
    // CUs); with the stream-K workspace the launch fills every CU itself and the side stream is left out (same step time, one
    // stream, capturable)
    // (asked of the launch itself: small or ragged batches, captured streams and "tn_flags" bit 1 all run the plain 216-tile form)
    const bool streamk = !cls && eoe_tn_streamk_would_run(w, 4, stream);
    SideStream* ss = (g_vit_side_stream == 2 || (g_vit_side_stream && (!streamk || b->async_wgrad))) && b->red_scratch ? side_stream(s) : nullptr;
    // Asynchronous weight gradients (b->async_wgrad; round 3).  Nothing in the backward sweep needs a block's weight gradients, and the
    // grouped wgrad launch leaves 40 of the 256 CUs idle for its ~213 us (216 one-per-CU workgroups): launched on the side stream, it runs
    // UNDER the next block's chain (LayerNorm-1 backward here, then that block's dgrad GEMMs on the free CUs) instead of in front of it --
    // measured 11.65 -> 11.24 ms per step.  Contract with the caller (eoe_hip.h): every buffer the launch reads (dh, dqkv, d16_c, the dY of
    // c_proj, the saved activations) stays untouched until a later call on this stream has passed its own fork point -- there the compute
    // stream is ordered behind the previous launch, long finished by then -- or eoe_vit_side_join() was called.
    if (ss && b->async_wgrad) {
        // async_wgrad == 1: this call orders the stream behind the PREVIOUS call's launch (the caller alternates two sets of the buffers a launch
        // reads).  async_wgrad >= 2 (round 5): behind the launch BEFORE the previous one only (three sets) -- the previous launch (~200 us beside
        // this call's ~340-us chain) is often still running when this call reaches its fork point, and the wait stalled the compute stream for
        // ~12 us per block (profiles/r4/timeline.txt); two launches back is long finished.
        const unsigned e_prev = (ss->seq + 1) & 1, e_old = ss->seq & 1;       // launch seq - 1 / launch seq - 2 (the event this call re-records)
        if (ss->recorded[e_old]) {
            if (hipStreamWaitEvent(s, ss->wgrad_done[e_old], 0) != hipSuccess) return eoe_set_error(EOE_ERR_LAUNCH, "vit_block_bwd: stream wait failed");
            ss->recorded[e_old] = false;
        }
        if (b->async_wgrad < 2 && ss->recorded[e_prev]) {
            if (hipStreamWaitEvent(s, ss->wgrad_done[e_prev], 0) != hipSuccess) return eoe_set_error(EOE_ERR_LAUNCH, "vit_block_bwd: stream wait failed");
            ss->recorded[e_prev] = false;
        }
        if (hipEventRecord(ss->fork, s) != hipSuccess || hipStreamWaitEvent(ss->s, ss->fork, 0) != hipSuccess)
            return eoe_set_error(EOE_ERR_LAUNCH, "vit_block_bwd: fork failed");
        TRY(launch_wgrads((void*)ss->s));
        if (hipEventRecord(ss->wgrad_done[e_old], ss->s) != hipSuccess) return eoe_set_error(EOE_ERR_LAUNCH, "vit_block_bwd: event record failed");
        ss->recorded[e_old] = true;
        ss->seq += 1;
        TRY(eoe_layernorm_bwd(b->d16_b, 0, a->x_in, D, a->stats1, a->ln1_g, b->dx_mid, b->dx_in, D, b->next_d16, b->g_ln1_g,
                              b->g_ln1_b, nullptr, red_ln1, M, D, dt, stream));
        TRY(b->red_table && b->red_scratch ? eoe_red_table_append(b->red_table, &jobs, stream) : eoe_flush_reduce(b->red_scratch ? &jobs : nullptr, stream));
        return 0;
    }
    if (ss) {
        // fork before the wgrad launch (LayerNorm-1 backward depends on d xn1 and dx_mid only), join before the finish kernel
        if (hipEventRecord(ss->fork, s) != hipSuccess) return eoe_set_error(EOE_ERR_LAUNCH, "vit_block_bwd: event record failed");
        TRY(launch_wgrads(stream));
        if (hipStreamWaitEvent(ss->s, ss->fork, 0) != hipSuccess) return eoe_set_error(EOE_ERR_LAUNCH, "vit_block_bwd: stream wait failed");
        TRY(eoe_layernorm_bwd(b->d16_b, 0, a->x_in, D, a->stats1, a->ln1_g, b->dx_mid, b->dx_in, D, b->next_d16, b->g_ln1_g,
                              b->g_ln1_b, nullptr, red_ln1, M, D, dt, (void*)ss->s));
        if (hipEventRecord(ss->join, ss->s) != hipSuccess || hipStreamWaitEvent(s, ss->join, 0) != hipSuccess)
            return eoe_set_error(EOE_ERR_LAUNCH, "vit_block_bwd: join failed");
    } else {
        TRY(launch_wgrads(stream));
        TRY(eoe_layernorm_bwd(b->d16_b, 0, a->x_in, D, a->stats1, a->ln1_g, b->dx_mid, b->dx_in, D, b->next_d16, b->g_ln1_g,
                              b->g_ln1_b, nullptr, red_ln1, M, D, dt, stream));
    }
    TRY(b->red_table && b->red_scratch ? eoe_red_table_append(b->red_table, &jobs, stream) : eoe_flush_reduce(b->red_scratch ? &jobs : nullptr, stream));
    return 0;
}

// orders `stream` behind the weight-gradient launch an asynchronous eoe_vit_block_bwd left on the side stream (no-op if there is none):
// call it before anything reads the weight gradients or reuses the buffers that launch reads (see eoe_vit_block_bwd_args.async_wgrad)
extern "C" int eoe_vit_side_join(void* stream) {
    hipStream_t s = (hipStream_t)stream;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return 0;      // never used while capturing
    SideStream* ss = side_stream(s);
    if (!ss) return 0;
    for (int e = 0; e < 2; ++e) {
        if (!ss->recorded[e]) continue;
        if (hipStreamWaitEvent(s, ss->wgrad_done[e], 0) != hipSuccess) return eoe_set_error(EOE_ERR_LAUNCH, "vit_side_join: stream wait failed");
        ss->recorded[e] = false;
    }
    return 0;
}
