// bt_decoder_forward: one GPT-NeoX stack (block decoder or token decoder) over M = B*q_len rows, as a fixed sequence of
// kernel launches on one stream — the unit the Python host captures into a hipGraph per block step.
//
// Per layer (parallel residual, HF GPTNeoXLayer):
//   LN1|LN2(x) -> qkv = LN1 Wqkv^T + b with RoPE + KV append fused into the GEMM epilogue -> attention -> x += att Wd^T + bd (in place, fused epilogue)
//   -> h = gelu(LN2 W1^T + b1) (fused epilogue) -> x += h W2^T + b2 (in place, fused epilogue)
// 6 launches per layer (7 when the fused QKV epilogue does not apply) instead of the reference's ~25 ATen kernels; no torch.cat cache copy.
//
// Activation workspace for M rows (elements): [ln1|att: M*H][ln2: M*H][qkv: M*3H][h4: M*I].
#include "bt_common.h"

using namespace bt;

static int g_few_rows_max = 16;             // rows up to which a stack takes the fused single-launch projections (the
                                            // entry points accept <= 64; beyond ~16 rows split-K + reduce is faster)
extern "C" void bt_debug_few_rows_max(int n) { g_few_rows_max = n; }
static bool g_no_few_rows_fusion = false;   // test hook: run 1..4-row stacks through the unfused launch sequence (A/B, parity)
extern "C" void bt_debug_no_few_rows_fusion(int on) { g_no_few_rows_fusion = on != 0; }

static int64_t ws_elems(const bt_decoder_dims* d, int64_t M) {
    return M * ((int64_t)d->hidden * 5 + d->inter);
}
// split-K scratch for skinny GEMMs (M <= 256 rows): up to 16 fp32 partial slabs of [M, inter]
static int64_t splitk_bytes(const bt_decoder_dims* d, int64_t M) {
    return M <= 256 ? 16 * M * (int64_t)d->inter * 4 : 0;
}

extern "C" int64_t bt_decoder_workspace_bytes(const bt_decoder_dims* dims, int dtype, int64_t M) {
    if (!dims || M < 0) return -1;
    return ws_elems(dims, M) * (int64_t)dtype_size(dtype) + splitk_bytes(dims, M) + 512;
}

extern "C" int bt_decoder_forward(int dtype, const bt_decoder_dims* dims, const bt_decoder_weights* w, int64_t B,
                                  int32_t q_len, void* x, const bt_attn_args* attn, int32_t first_out_row, void* out,
                                  void* ws, int64_t ws_bytes, bt_stream stream) {
    BT_CHECK_ARG(dims && w && w->layers_host && attn && x && out && ws, "bt_decoder_forward: null pointer");
    BT_CHECK_ARG(dtype == BT_F32 || dtype == BT_BF16, "bt_decoder_forward: bad dtype %d", dtype);
    BT_CHECK_ARG(B >= 0 && q_len > 0 && first_out_row >= 0 && first_out_row < q_len, "bt_decoder_forward: bad shape");
    BT_CHECK_ARG(dims->hidden == dims->heads * dims->head_dim, "bt_decoder_forward: hidden != heads*head_dim");
    if (B == 0) return BT_OK;
    const int64_t M = B * q_len;
    const int64_t H = dims->hidden, I = dims->inter;
    const size_t es = dtype_size(dtype);
    // 256-byte align the workspace base
    uintptr_t base = (reinterpret_cast<uintptr_t>(ws) + 255u) & ~(uintptr_t)255u;
    const int64_t avail = ws_bytes - (int64_t)(base - reinterpret_cast<uintptr_t>(ws));
    // the activation buffers are mandatory; the split-K scratch is an optimisation (a workspace sized for a larger M, e.g.
    // the token decoder's B*E prefix rows reused for its B-row steps, simply runs the skinny GEMMs without split-K)
    if (avail < ws_elems(dims, M) * (int64_t)es + 256) {
        set_error("bt_decoder_forward: workspace too small: %lld bytes for M=%lld (need %lld)", (long long)ws_bytes,
                  (long long)M, (long long)(ws_elems(dims, M) * (int64_t)es + 512));
        return BT_ERR_WORKSPACE;
    }
    char* p = reinterpret_cast<char*>(base);
    void* ln1 = p;              // also the attention output
    void* ln2 = p + M * H * es;
    void* qkv = p + 2 * M * H * es;
    void* h4 = p + 5 * M * H * es;
    void* att = ln1;
    char* sk = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p + ws_elems(dims, M) * es) + 255u) & ~(uintptr_t)255u);
    int64_t sk_bytes = splitk_bytes(dims, M);
    if (sk_bytes && (sk + sk_bytes) > (reinterpret_cast<char*>(ws) + ws_bytes)) sk_bytes = 0;
    void* scratch = sk_bytes ? sk : nullptr;

    // When the caller consumes only the LAST row of every sequence (prefill -> context embedding; token-decoder step 0),
    // the last layer's attention output / dense / MLP are needed for that row alone: the other rows of that layer only
    // contribute their K,V (cache) — skip 9/12 of the last layer's GEMM flops and its attention, exactly.
    const bool tail_only = (q_len > 1) && (first_out_row == q_len - 1);
    // the fused small-batch path: same conditions as bt_ln_linear / bt_ln_qkv_rope_append accept (checked up front so a
    // layer never runs half fused).  With so few rows the last-layer shortcut for tail-only callers is not worth a
    // second launch sequence: every layer runs on every row and the final LayerNorm picks the rows the caller consumes.
    const bool few_rows = dtype == BT_BF16 && M <= g_few_rows_max && (M > 4 ? (H % 32 == 0 && I % 32 == 0) : (H % 512 == 0 && H <= 2048)) &&
                          dims->rot == 32 &&
                          dims->head_dim % 32 == 0 && ((reinterpret_cast<uintptr_t>(x) & 15u) == 0) && !g_no_few_rows_fusion;
    for (int l = 0; l < dims->layers; ++l) {
        const bt_layer_weights& lw = w->layers_host[l];
        char* kc = reinterpret_cast<char*>(attn->k_cache) + (int64_t)l * attn->layer_stride * es;
        char* vc = reinterpret_cast<char*>(attn->v_cache) + (int64_t)l * attn->layer_stride * es;
        const bool last = tail_only && (l == dims->layers - 1);
        if (few_rows) {
            // 1..4 rows (small-batch decode, launch-bound): LayerNorms, RoPE and the KV append ride inside the weight-
            // streaming GEMVs, and the two residual-stream projections share one -- 4 launches per layer instead of 7.  The MLP's first projection reads the layer INPUT
            // (parallel residual), so it runs before the attention output projection updates x in place.
            BT_TRY(bt_ln_qkv_rope_append(dtype, B, q_len, dims->heads, dims->head_dim, dims->rot, x, H, lw.ln1_w, lw.ln1_b,
                                         dims->ln_eps, lw.qkv_w, lw.qkv_b, qkv, 3 * H, kc, vc, attn->n_max, attn->pos0_dev,
                                         attn->pos0_host, attn->pos_table, attn->cos_tab, attn->sin_tab, stream));
            BT_TRY(bt_attention(dtype, attn->mode, B, q_len, dims->heads, dims->head_dim, qkv, 3 * H, kc, vc, attn->n_max,
                                attn->pos0_dev, attn->pos0_host, attn->mask, attn->mask_ld, att, H, attn->n_split,
                                attn->split_ws, attn->split_ws_bytes, stream));
            BT_TRY(bt_ln_linear(dtype, M, I, H, x, H, lw.ln2_w, lw.ln2_b, dims->ln_eps, lw.h4_w, lw.h4_b, nullptr, 0, h4, I,
                                BT_ACT_GELU, stream));
            // x += mlp_down(h4) + attn_dense(att): both residual-stream projections in one launch (K = I, then K = H)
            if (M > 4 || I % 2048 == 0) {
                BT_TRY(bt_linear2(dtype, M, H, I, h4, I, lw.h_w, lw.h_b, H, att, H, lw.dense_w, lw.dense_b, x, H, x, H, stream));
            } else {
                BT_TRY(bt_linear_ws(dtype, M, H, H, att, H, lw.dense_w, lw.dense_b, x, H, x, H, BT_ACT_NONE, scratch, sk_bytes, stream));
                BT_TRY(bt_linear_ws(dtype, M, H, I, h4, I, lw.h_w, lw.h_b, x, H, x, H, BT_ACT_NONE, scratch, sk_bytes, stream));
            }
            continue;
        }
        if (!last) {
            BT_TRY(bt_layernorm(dtype, M, H, x, H, lw.ln1_w, lw.ln1_b, ln1, H, lw.ln2_w, lw.ln2_b, ln2, H, dims->ln_eps,
                                stream));
        } else {
            // LN1 for every row (QKV needs it), LN2 only for the consumed rows (compacted into ln2[0..B))
            BT_TRY(bt_layernorm(dtype, M, H, x, H, lw.ln1_w, lw.ln1_b, ln1, H, nullptr, nullptr, nullptr, 0, dims->ln_eps,
                                stream));
            const char* xl = reinterpret_cast<const char*>(x) + (int64_t)(q_len - 1) * H * es;
            BT_TRY(bt_layernorm(dtype, B, H, xl, (int64_t)q_len * H, lw.ln2_w, lw.ln2_b, ln2, H, nullptr, nullptr, nullptr,
                                0, dims->ln_eps, stream));
        }
        BT_TRY(bt_qkv_rope_append(dtype, B, q_len, dims->heads, dims->head_dim, dims->rot, ln1, H, lw.qkv_w, lw.qkv_b, qkv,
                                  3 * H, kc, vc, attn->n_max, attn->pos0_dev, attn->pos0_host, attn->pos_table,
                                  attn->cos_tab, attn->sin_tab, scratch, sk_bytes, stream));
        if (!last) {
            BT_TRY(bt_attention(dtype, attn->mode, B, q_len, dims->heads, dims->head_dim, qkv, 3 * H, kc, vc,
                                attn->n_max, attn->pos0_dev, attn->pos0_host, attn->mask, attn->mask_ld, att, H,
                                attn->n_split, attn->split_ws, attn->split_ws_bytes, stream));
            BT_TRY(bt_linear_ws(dtype, M, H, H, att, H, lw.dense_w, lw.dense_b, x, H, x, H, BT_ACT_NONE, scratch, sk_bytes, stream));
            BT_TRY(bt_linear_ws(dtype, M, I, H, ln2, H, lw.h4_w, lw.h4_b, nullptr, 0, h4, I, BT_ACT_GELU, scratch, sk_bytes, stream));
            BT_TRY(bt_linear_ws(dtype, M, H, I, h4, I, lw.h_w, lw.h_b, x, H, x, H, BT_ACT_NONE, scratch, sk_bytes, stream));
        } else {
            // one query row per sequence: row (q_len - 1) at cache position pos0 + q_len - 1; row-wise kernel, q_len = 1
            const char* ql = reinterpret_cast<const char*>(qkv) + (int64_t)(q_len - 1) * 3 * H * es;
            char* xl = reinterpret_cast<char*>(x) + (int64_t)(q_len - 1) * H * es;
            int32_t split = attn->mode == 0 ? attn->n_split : 1;
            // the device-side start position (if any) counts from the first new row: shift it by q_len - 1 on the host
            // side only when it is a host value; a device scalar cannot be shifted here, so fall back to the full path
            if (attn->pos0_dev != nullptr) {
                BT_TRY(bt_attention(dtype, attn->mode, B, q_len, dims->heads, dims->head_dim, qkv, 3 * H, kc, vc,
                                    attn->n_max, attn->pos0_dev, attn->pos0_host, attn->mask, attn->mask_ld, att, H,
                                    attn->n_split, attn->split_ws, attn->split_ws_bytes, stream));
                const char* al = reinterpret_cast<const char*>(att) + (int64_t)(q_len - 1) * H * es;
                BT_TRY(bt_linear_ws(dtype, B, H, H, al, (int64_t)q_len * H, lw.dense_w, lw.dense_b, xl, (int64_t)q_len * H,
                                    xl, (int64_t)q_len * H, BT_ACT_NONE, scratch, sk_bytes, stream));
            } else {
                BT_TRY(bt_attention(dtype, 0, B, 1, dims->heads, dims->head_dim, ql, (int64_t)q_len * 3 * H, kc, vc,
                                    attn->n_max, nullptr, attn->pos0_host + q_len - 1, attn->mask, attn->mask_ld, att, H,
                                    split, attn->split_ws, attn->split_ws_bytes, stream));
                BT_TRY(bt_linear_ws(dtype, B, H, H, att, H, lw.dense_w, lw.dense_b, xl, (int64_t)q_len * H, xl,
                                    (int64_t)q_len * H, BT_ACT_NONE, scratch, sk_bytes, stream));
            }
            BT_TRY(bt_linear_ws(dtype, B, I, H, ln2, H, lw.h4_w, lw.h4_b, nullptr, 0, h4, I, BT_ACT_GELU, scratch, sk_bytes, stream));
            BT_TRY(bt_linear_ws(dtype, B, H, I, h4, I, lw.h_w, lw.h_b, xl, (int64_t)q_len * H, xl, (int64_t)q_len * H,
                                BT_ACT_NONE, scratch, sk_bytes, stream));
        }
    }
    // final_layer_norm of the rows the caller consumes
    const int count = q_len - first_out_row;
    if (first_out_row == 0) {
        BT_TRY(bt_layernorm(dtype, M, H, x, H, w->lnf_w, w->lnf_b, out, H, nullptr, nullptr, nullptr, 0, dims->ln_eps,
                            stream));
    } else {
        for (int t = 0; t < count; ++t) {
            const char* xin = reinterpret_cast<const char*>(x) + (int64_t)(first_out_row + t) * H * es;
            char* o = reinterpret_cast<char*>(out) + (int64_t)t * H * es;
            BT_TRY(bt_layernorm(dtype, B, H, xin, (int64_t)q_len * H, w->lnf_w, w->lnf_b, o, (int64_t)count * H, nullptr,
                                nullptr, nullptr, 0, dims->ln_eps, stream));
        }
    }
    return BT_OK;
}
