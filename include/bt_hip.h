/*
 * bt_hip.h — C-ABI of libbt_hip.so, the MI355X (gfx950) kernel library behind the Block Transformer decode path.
 *
 * This is the drop-in boundary B2 of SURVEY.md section 8(b).  The reference (itsnamgyu/block-transformer) has no
 * native code: its decode arithmetic is issued through PyTorch/HF `transformers==4.39.3` GPT-NeoX modules and
 * flash-attn.  Each entry point below therefore cites the reference *Python* interface whose arithmetic it
 * replaces (paths relative to the reference repo root).  The Python host side (block_transformer_amd/model/...)
 * keeps the reference's module API and binds these symbols with ctypes.
 *
 * Contract (every entry point):
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless the name ends in `_host`;
 *   - never allocates or frees device memory (the caller passes workspaces), never synchronises the stream,
 *     no internal threads, no process-global mutable dispatch state (kernel selection is a pure function of the
 *     arguments; the A/B and ablation hooks `bt_debug_*` exist only in the separate test build libbt_hip_dbg.so,
 *     compiled with -DBT_DEBUG_HOOKS); re-entrant per (device, stream).  The one opt-in exception is the bench-only
 *     kernel-family timer `bt_prof_enable` (off by default; not thread-safe while enabled);
 *   - enqueues its kernels on `stream` (a hipStream_t passed as void*; NULL = the null stream) and returns
 *     BT_OK (0) or a negative bt_status; `bt_last_error()` returns a thread-local message;
 *   - safe inside hipStreamBeginCapture / hipGraph capture (no host reads of device memory): values that change
 *     between graph replays (number of cached global positions) are read from device scalars.
 *
 * Data layout in HBM (dtype T = float or bf16, selected by `dtype`):
 *   activations      x[M, H]                 row-major, explicit row stride (elements)
 *   fused QKV rows   qkv[M, 3H]              HF GPT-NeoX per-head interleaved: col = head*3d + {q:0..d-1,k:d..2d-1,v:2d..3d-1}
 *   weights          W[N, K]                 HF nn.Linear layout (out_features, in_features), row-major, unchanged
 *   global KV cache  K,V[layer][B][heads][Nmax][d]   one (k,v) per L-token block; K stored post-RoPE
 *   local  KV cache  K,V[layer][B][heads][E+L-1][d]  token-decoder window; "reset" = start writing at row 0 again
 *   block mask       mask[B, mask_ld] uint8  1 = key visible (reference block_attention_mask)
 *   RoPE tables      cos,sin[max_pos, rot/2] float32, angle = pos * base^(-2j/rot)
 *   token ids        int64 (torch.long)
 */
#ifndef BT_HIP_H
#define BT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BT_VERSION 310 /* 0.3.0: bt_attn_args / bt_engine_desc start with struct_size (NOT binary compatible with 0.2.x: a binder
                          * compiled against an older header is rejected by the size check instead of having fields read past the
                          * end of its struct); stream-K GEMM path (bt_linear2_ws, bt_stream_k_scratch_bytes); bt_ceiling_*
                          * 0.3.1: bt_attn_args.err_flag (the struct grew: the size check rejects 0.3.0 binders), BT_DEV_ERR_HANDOFF_TIMEOUT,
                          * bt_stream_k_reset / bt_stream_k_status; the shared object exports bt_* symbols only */

typedef void* bt_stream; /* hipStream_t */

enum bt_dtype { BT_F32 = 0, BT_BF16 = 1 };
enum bt_act { BT_ACT_NONE = 0, BT_ACT_GELU = 1 /* exact erf GELU, HF hidden_act="gelu" */ };
enum bt_status {
    BT_OK = 0,
    BT_ERR_ARG = -1,         /* bad shape / null pointer / misalignment */
    BT_ERR_UNSUPPORTED = -2, /* dtype or head_dim not instantiated */
    BT_ERR_WORKSPACE = -3,   /* workspace too small */
    BT_ERR_HIP = -4          /* a HIP runtime call failed (message has hipGetErrorString) */
};

/* Shape of one GPT-NeoX stack (HF GPTNeoXConfig fields the decode path reads). */
typedef struct bt_decoder_dims {
    int32_t hidden;   /* H */
    int32_t heads;
    int32_t head_dim; /* d = H / heads: 32, 64 or 128 */
    int32_t rot;      /* rotary_ndims = int(d * rotary_pct) */
    int32_t inter;    /* intermediate_size */
    int32_t layers;
    float ln_eps;
} bt_decoder_dims;

/* One layer's parameters, HF names: input_layernorm, post_attention_layernorm, attention.query_key_value,
 * attention.dense, mlp.dense_h_to_4h, mlp.dense_4h_to_h (weight [out,in], bias [out]). */
typedef struct bt_layer_weights {
    const void *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    const void *qkv_w, *qkv_b;
    const void *dense_w, *dense_b;
    const void *h4_w, *h4_b;
    const void *h_w, *h_b;
    /* ---- since 0.3.0, optional (all six or none; bf16): LayerNorm folded into the two GEMMs that consume it --
     *   qkv_wf = qkv_w * diag(ln1_w),  qkv_s[n] = sum_k qkv_wf[n, k],  qkv_c = qkv_w . ln1_b + qkv_b  (s, c: float32 [N]); h4_* likewise
     * with ln2.  Honoured by the DEBUG build only: the experiment (row statistics + a correction rstd * (acc - mean * s) + c in the
     * GEMM epilogues instead of the dual LayerNorm launch) measured no net gain (profiles/r5_ab_ln_fused.jsonl); the product
     * library ignores the six fields -- leave them NULL. */
    const void *qkv_wf; const float *qkv_s, *qkv_c;
    const void *h4_wf;  const float *h4_s,  *h4_c;
} bt_layer_weights;

typedef struct bt_decoder_weights {
    const bt_layer_weights* layers_host; /* HOST array of `dims.layers` entries (device pointers inside) */
    const void *lnf_w, *lnf_b;           /* final_layer_norm */
} bt_decoder_weights;

/* ------------------------------------------------------------------------------------------------------------ */
int bt_version(void);
const char* bt_last_error(void);
/* Number of bytes of activation workspace bt_decoder_forward needs for M rows. */
int64_t bt_decoder_workspace_bytes(const bt_decoder_dims* dims, int dtype, int64_t M);

/* ---- single ops (each replaces one ATen/cuBLAS/flash-attn call pattern of SURVEY.md section 2.2) -------------- */

/* out[M,N] = act(A[M,K] . W[N,K]^T + bias[N]) + residual[M,N]     (C3, C7, C8, C9, C10, C11)
 * Replaces nn.Linear / Conv1d(k=1) calls inside HF GPTNeoXAttention/GPTNeoXMLP (reference call sites
 * model/block_decoder/gpt_neo_x.py:100-111, model/token_decoder/gpt_neo_x.py:59-74) and
 * BaseTokenDecoder.expand_block_embeddings (model/token_decoder/base.py:118-139).
 * bias/residual may be NULL.  bf16: MFMA kernel when K % 64 == 0 and rows are 16-byte aligned. */
int bt_linear(int dtype, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* W,
              const void* bias, const void* residual, int64_t ldr, void* out, int64_t ldo, int act,
              bt_stream stream);

/* bt_linear with caller-provided scratch (fp32, >= 2*M*N*4 bytes to be useful): lets skinny problems (M <~ 256 rows, the
 * weight-streaming regime of small-batch decode) split K across workgroups so that all 256 CUs stream W; the partials are
 * reduced by a second kernel that applies the same epilogue.  Identical results contract as bt_linear. */
int bt_linear_ws(int dtype, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* W, const void* bias,
                 const void* residual, int64_t ldr, void* out, int64_t ldo, int act, void* scratch,
                 int64_t scratch_bytes, bt_stream stream);

/* LayerNorm of the same row with one or two affine sets (parallel residual reads LN1(x) and LN2(x)):  (C2)
 * out1 = LN(x)*w1+b1, out2 = LN(x)*w2+b2 (out2/w2/b2 may be NULL).  HF GPTNeoXLayer.input_layernorm /
 * post_attention_layernorm / GPTNeoXModel.final_layer_norm. */
int bt_layernorm(int dtype, int64_t M, int64_t H, const void* x, int64_t ldx, const void* w1, const void* b1,
                 void* out1, int64_t ldo1, const void* w2, const void* b2, void* out2, int64_t ldo2, float eps,
                 bt_stream stream);

/* out[r, :] = table[ids[r*ids_stride], :] (width elements per row); if row_zero != NULL and row_zero[r / rows_per_seq] != 0
 * the row is zero-filled.  With ids[B*L] and width h_e the output viewed as [B, L*h_e] IS the concat projection:
 * LookupEmbedder.forward (model/embedder/lookup.py:71-87) + BaseEmbedder.project_hidden_state concat branch
 * (model/embedder/base.py:79-89); with width H it is GPTNeoXTokenDecoder.embed_input_ids
 * (model/token_decoder/gpt_neo_x.py:52-53).  (C1) */
int bt_embed_rows(int dtype, int64_t rows, int64_t width, const int64_t* ids, int64_t ids_stride, int64_t vocab,
                  const void* table, void* out, int64_t ldo, const int32_t* row_zero, int32_t rows_per_seq,
                  bt_stream stream);

/* Partial RoPE on q and k of `q_len` new rows per sequence, in place in qkv, and append of (k_rot, v) to a KV cache:
 *   cache row = pos0 + t (t < q_len), RoPE angle position = pos_table ? pos_table[pos0+t] : pos0 + t,
 *   pos0 = pos0_dev ? *pos0_dev : pos0_host.
 * Replaces HF apply_rotary_pos_emb + the torch.cat KV growth (C4, C5).  qkv rows are ordered [b][t]. */
int bt_rope_kv_append(int dtype, int64_t B, int32_t q_len, int32_t heads, int32_t head_dim, int32_t rot, void* qkv,
                      int64_t ld_qkv, void* k_cache, void* v_cache, int64_t n_max, const int32_t* pos0_dev,
                      int32_t pos0_host, const int32_t* pos_table, const float* cos_tab, const float* sin_tab,
                      bt_stream stream);

/* Fused form of the two calls above for the GPT-NeoX query_key_value projection of a layer:
 *   qkv = x[M, H] . Wqkv[3H, H]^T + b ; RoPE on q (written to qkv, k/v columns of qkv are NOT written) and k ;
 *   (k_rot, v) appended to the cache rows pos0 + t  —  RoPE and the append run in the GEMM epilogue (bf16, rot == 32,
 *   head_dim % 32 == 0); any other configuration runs bt_linear + bt_rope_kv_append (then all of qkv is written).
 * Replaces HF GPTNeoXAttention.query_key_value + apply_rotary_pos_emb + torch.cat cache growth (C3, C4, C5). */
int bt_qkv_rope_append(int dtype, int64_t B, int32_t q_len, int32_t heads, int32_t head_dim, int32_t rot, const void* x,
                       int64_t ldx, const void* qkv_w, const void* qkv_b, void* qkv, int64_t ld_qkv, void* k_cache,
                       void* v_cache, int64_t n_max, const int32_t* pos0_dev, int32_t pos0_host,
                       const int32_t* pos_table, const float* cos_tab, const float* sin_tab, void* scratch,
                       int64_t scratch_bytes, bt_stream stream);   /* scratch: optional split-K scratch, see bt_linear_ws */

/* Small-batch decode (bf16; 1..4 rows: weight-streaming GEMV kernel, 5..64 rows: MFMA weight streamer): the same
 * projections as ONE launch each with the neighbouring elementwise ops folded in (small-batch decode is launch-bound:
 * ~5 us per dependent launch).
 *   bt_ln_linear:          out = act(LayerNorm(x; ln_w, ln_b, eps) . W^T + bias) + residual
 *                          = HF GPTNeoXLayer.{input,post_attention}_layernorm + the nn.Linear that follows it
 *                          (1..4 rows: K % 512 == 0, K <= 2048; 5..64 rows: K % 32 == 0, N % 4 == 0)
 *   bt_ln_qkv_rope_append: bt_qkv_rope_append with the layer's input LayerNorm in front (ln_w == NULL: no LayerNorm) and
 *                          RoPE + KV append done by the same kernel (rot == 32, head_dim % 32 == 0; 1..4 rows: hidden % 512 == 0)
 * All three return BT_ERR_UNSUPPORTED (nothing launched) for any other configuration; the caller then uses bt_layernorm +
 * bt_linear / bt_qkv_rope_append (bt_decoder_forward does exactly that, and takes this route up to 16 rows). */
int bt_ln_linear(int dtype, int64_t M, int64_t N, int64_t K, const void* x, int64_t ldx, const void* ln_w, const void* ln_b,
                 float eps, const void* W, const void* bias, const void* residual, int64_t ldr, void* out, int64_t ldo,
                 int act, bt_stream stream);
/*   bt_linear2:            out = x1 . W1^T + bias1 + x2 . W2^T + bias2 + residual   (1..4 rows: K1 % 2048 == 0, K2 % 512 == 0;
 *                          5..64 rows: K % 32 == 0, N % 4 == 0; > 64 rows: one MFMA GEMM over K1 + K2, K % 64 == 0,
 *                          N % 8 == 0, both biases present; BT_ERR_UNSUPPORTED otherwise): the two
 *                          projections that feed the residual stream of a parallel-residual layer (attention.dense and
 *                          mlp.dense_4h_to_h) as one launch */
int bt_linear2(int dtype, int64_t M, int64_t N, int64_t K1, const void* x1, int64_t ldx1, const void* W1, const void* bias1,
               int64_t K2, const void* x2, int64_t ldx2, const void* W2, const void* bias2, const void* residual,
               int64_t ldr, void* out, int64_t ldo, bt_stream stream);
/* bt_linear2 with caller-provided scratch: above 256 rows, with >= bt_stream_k_scratch_bytes() bytes (16-byte aligned) and full
 * 256 x 256 output tiles, the two projections run as ONE stream-K MFMA GEMM over K1 + K2 (items of a tile's K range dealt out to
 * all CUs; a tile shared by two workgroups is handed over through fp32 slabs + flag words inside `scratch`: no second launch, x read
 * and written once).  Any other case behaves like bt_linear2.  bt_linear_ws / bt_qkv_rope_append take the same route with that much
 * scratch when their 256 x 256 tiles do not fill whole rounds of CUs (M = 4096 QKV: 384 tiles).  Results are not bit-identical to
 * the two-launch form (one bf16 rounding of the sum instead of two, fp32 summation split at the hand-off). */
int64_t bt_stream_k_scratch_bytes(void);
/* The hand-off of the stream-K path cannot deadlock, but it CAN time out: the workgroup that finishes a shared tile polls its
 * partner's flag a bounded number of times (~1 s) -- every workgroup of the launch is resident on an otherwise idle GPU, not
 * necessarily beside another process's kernels.  A consumer that gives up does NOT add the (unwritten) slab: it sets a sticky
 * word inside `scratch`, and the results of that launch are incomplete.  The sticky word survives the per-launch clearing of the
 * flag words; it is how a caller of bt_linear_ws / bt_linear2_ws / bt_qkv_rope_append learns about it:
 *   bt_stream_k_reset   zero the flag words AND the sticky word -- a one-workgroup kernel, not a memset (once after allocating `scratch`, whose bytes are otherwise
 *                       undefined, and to acknowledge a reported time-out);
 *   bt_stream_k_status  enqueue a one-thread kernel that ORs BT_DEV_ERR_HANDOFF_TIMEOUT into *err_flag (device int32) when the
 *                       sticky word of `scratch` is set; the host reads *err_flag whenever it synchronises anyway.
 * bt_decoder_forward does both itself around every pass that may take the path (bt_attn_args.err_flag), and so do the
 * bt_*_enqueue entry points (bt_engine_desc.err_flag).  A scratch smaller than bt_stream_k_scratch_bytes() is ignored (BT_OK). */
int bt_stream_k_reset(void* scratch, int64_t scratch_bytes, bt_stream stream);
int bt_stream_k_status(const void* scratch, int64_t scratch_bytes, int32_t* err_flag, bt_stream stream);
int bt_linear2_ws(int dtype, int64_t M, int64_t N, int64_t K1, const void* x1, int64_t ldx1, const void* W1, const void* bias1,
                  int64_t K2, const void* x2, int64_t ldx2, const void* W2, const void* bias2, const void* residual,
                  int64_t ldr, void* out, int64_t ldo, void* scratch, int64_t scratch_bytes, bt_stream stream);
int bt_ln_qkv_rope_append(int dtype, int64_t B, int32_t q_len, int32_t heads, int32_t head_dim, int32_t rot, const void* x,
                          int64_t ldx, const void* ln_w, const void* ln_b, float eps, const void* qkv_w, const void* qkv_b,
                          void* qkv, int64_t ld_qkv, void* k_cache, void* v_cache, int64_t n_max, const int32_t* pos0_dev,
                          int32_t pos0_host, const int32_t* pos_table, const float* cos_tab, const float* sin_tab,
                          bt_stream stream);

/* Causal softmax attention of q_len (already rotated) query rows per sequence over the cache:
 *   query t sees keys j <= pos0 + t with mask[b, j] != 0 (mask may be NULL);  out[b*q_len+t, head*d + :]
 * mode: 0 = row-wise streaming kernel (decode: q_len = 1 global step, or the token decoder's <= E+L-1 window);
 *       1 = MFMA flash kernel (bf16 prefill).  Replaces flash-attn fwd / eager baddbmm+softmax+matmul (C6).
 * `ws`/`ws_bytes`: scratch for split-N partials (may be NULL when n_split == 1).  Rows whose keys are all masked
 * produce zeros (the reference leaves them unspecified: finfo.min arithmetic). */
/* The split-N factor bt_attention(mode 0) will actually use for a hint: raised until one split's score row fits in LDS
 * (about 15.8k keys per split at head_dim 128).  Callers size `ws` as B*heads*q_len*n_split*(head_dim+2) floats when > 1. */
int32_t bt_attention_split_count(int32_t head_dim, int64_t n_max, int32_t n_split_hint);
int bt_attention(int dtype, int mode, int64_t B, int32_t q_len, int32_t heads, int32_t head_dim, const void* qkv,
                 int64_t ld_qkv, const void* k_cache, const void* v_cache, int64_t n_max, const int32_t* pos0_dev,
                 int32_t pos0_host, const uint8_t* mask, int64_t mask_ld, void* out, int64_t ldo, int32_t n_split,
                 void* ws, int64_t ws_bytes, bt_stream stream);

/* ids_out[r*ids_stride] = argmax_j logits[r, j] (first maximal index, torch.argmax semantics).  HF greedy_search (C12). */
int bt_argmax(int dtype, int64_t M, int64_t V, const void* logits, int64_t ld, int64_t* ids_out, int64_t ids_stride,
              bt_stream stream);

/* Greedy token step in one call: logits = hid[M, K] . W[V, K]^T (token_decoder.embed_out, no bias;
 * model/token_decoder/gpt_neo_x.py:73-74, base.py:349) and ids_out[r*ids_stride] = argmax_j logits[r, j] (HF greedy_search, C11 +
 * C12).  When the persistent MFMA GEMM takes the shape (bf16, >= ~192 output tiles of 256 x 256) its epilogue reduces every
 * 64-column slab of a row to one (value, column) pair -- over the bf16-rounded logits, lowest column on ties: the token
 * bt_linear + bt_argmax produce, bit for bit -- and a second small kernel reduces the pairs; the [M, V] logits matrix is then
 * written only if store_logits != 0.  `ws`: >= bt_logits_argmax_workspace_bytes(M, V) bytes (16-byte aligned) for the pairs;
 * without it, or for any other shape / dtype, the call is bt_linear into `logits` followed by bt_argmax.  `logits` must
 * always be a valid [M, ld_logits] buffer. */
int64_t bt_logits_argmax_workspace_bytes(int64_t M, int64_t V);
int bt_logits_argmax(int dtype, int64_t M, int64_t V, int64_t K, const void* hid, int64_t ldh, const void* W, void* logits,
                     int64_t ld_logits, int32_t store_logits, int64_t* ids_out, int64_t ids_stride, void* ws, int64_t ws_bytes,
                     bt_stream stream);

/* HF sample(): logits/temperature -> top-k filter -> softmax -> multinomial, one draw per row with a counter-based
 * RNG keyed by (seed, *step_dev, row) -- deterministic for a given key: the kept set is compacted in vocabulary order
 * (every score above the k-th largest + the lowest-index ties, at most 1024 candidates; top_k in (1024, V) is
 * refused with BT_ERR_UNSUPPORTED; top_k <= 0 or >= V = no filter).  Distributional parity only.  (C12) */
int bt_topk_sample(int dtype, int64_t M, int64_t V, const void* logits, int64_t ld, float temperature, int32_t top_k,
                   uint64_t seed, const int32_t* step_dev, int64_t row0, int64_t* ids_out, int64_t ids_stride,
                   bt_stream stream);   /* row0: RNG row index of logits row 0 (request shards of one batch draw the same
                                           streams as the whole batch would: key = (seed, *step_dev, row0 + r)) */

/* ---- composite: one GPT-NeoX stack over M = B*q_len rows (what GPTNeoXModel.forward does for inputs_embeds) ---- */
typedef struct bt_attn_args {
    uint32_t struct_size;    /* = sizeof(bt_attn_args) of the header the caller was compiled against; every entry point rejects
                                a value that is not its own (BT_ERR_ARG): structs grew between versions */
    void* k_cache;           /* [layers][B_alloc][heads][n_max][d]; points at this call's first sequence */
    void* v_cache;
    int64_t layer_stride;    /* elements between consecutive layers (= B_alloc*heads*n_max*d) */
    int64_t n_max;
    const int32_t* pos0_dev; /* device scalar: first cache row written by this call (NULL -> pos0_host) */
    int32_t pos0_host;
    const int32_t* pos_table; /* optional RoPE position indirection (token decoder policy table) */
    const uint8_t* mask;      /* [B, mask_ld] or NULL */
    int64_t mask_ld;
    const float* cos_tab;
    const float* sin_tab;
    int32_t mode;    /* bt_attention mode */
    int32_t n_split; /* split-N factor for mode 0 */
    void* split_ws;
    int64_t split_ws_bytes;
    const int32_t* seq_skip; /* optional [B]: != 0 marks a finished sequence whose decode-step attention (mode 0, q_len = 1) is
                                skipped (zeros are written; its tokens are forced to pad anyway) -- the reference compacts such
                                rows away, model/block_transformer.py:381-398 */
    /* ---- optional second stream for the parallel-residual branch (all three or none); honoured by the DEBUG build only since
     * 0.3.0 (measured -0.2 ... -0.3 % end to end: the product library runs one stream whatever these hold) ----
     * GPT-NeoX's parallel residual makes mlp.dense_h_to_4h independent of the attention (both read LayerNorms of the same layer
     * input; HF GPTNeoXLayer, use_parallel_residual).  With these set, stacks of > 256 bf16 rows enqueue that GEMM on `aux_stream`
     * (forked from and joined back into `stream` with the two events, capture-safe: hipEventRecord / hipStreamWaitEvent only), so it
     * can fill the CUs the QKV projection's last partial round of tiles and the attention kernel leave idle.  NULL = one stream. */
    bt_stream aux_stream;
    void* ev_fork;           /* hipEvent_t, created by the caller (hipEventDisableTiming is enough) */
    void* ev_join;
    /* ---- since 0.3.1 ---- */
    int32_t* err_flag;       /* device int32 or NULL: sticky BT_DEV_ERR_* bits of this pass -- today BT_DEV_ERR_HANDOFF_TIMEOUT of the
                                stream-K GEMMs (see bt_stream_k_status); with NULL a time-out goes unreported */
} bt_attn_args;

/* x[M,H] (row stride H) -> NL x {LN1|LN2, QKV, RoPE+append, attention, dense(+x), h_to_4h+GELU, 4h_to_h(+)} ->
 * final LN of rows (row % q_len) >= first_out_row into out[B*(q_len-first_out_row), H].
 * x is overwritten (residual stream).  GPTNeoXBlockDecoder.forward (model/block_decoder/gpt_neo_x.py:71-126) and
 * GPTNeoXTokenDecoder.model_forward (model/token_decoder/gpt_neo_x.py:55-71). */
int bt_decoder_forward(int dtype, const bt_decoder_dims* dims, const bt_decoder_weights* w, int64_t B, int32_t q_len,
                       void* x, const bt_attn_args* attn, int32_t first_out_row, void* out, void* ws,
                       int64_t ws_bytes, bt_stream stream);

/* ---- roll-over bookkeeping (BlockTransformer.generate, model/block_transformer.py:381-426) -------------------- */
/* Finish one generated block on device: apply the HF greedy_search EOS/pad rules to raw tokens raw[B, L]
 * (first `count` columns valid; rewritten in place with the final tokens), write them to out_ids[b, *blk_dev, :],
 * copy finished[] to finished_prev[] (the reference's decode_mask of this iteration: rows that get re-embedded),
 * update finished[b], and append the block mask column mask[b, *blk_dev] = !finished[b] (mask may be NULL).
 * gen_eos < 0 disables EOS handling (benchmark protocol, measure_generation_time.py:90). */
int bt_finish_block(int64_t B, int32_t L, int32_t count, int64_t* raw, int64_t* out_ids, int64_t nb_max,
                    const int32_t* blk_dev, int32_t* finished, int32_t* finished_prev, uint8_t* mask,
                    int64_t mask_ld, int64_t config_pad, int64_t gen_eos, int64_t gen_pad, bt_stream stream);
/* *counter += delta (device scalar bump inside a captured graph). */
int bt_bump(int32_t* counter, int32_t delta, bt_stream stream);

/* ---- the two-level block step as C entry points (BlockTransformer.generate loop body, model/block_transformer.py:370-426) ----
 * Everything one (batch, max_blocks) problem needs, as raw device pointers: a binder in any language fills this once,
 * then runs   bt_prefill_enqueue (per chunk of sequences)  ->  { bt_block_step_enqueue } x n_blocks   on a stream,
 * or captures bt_block_step_enqueue into a hipGraph (bt_graph_*) and replays it: every value that changes between
 * replays lives in device memory (n_cur, finished[], mask, sample_step).  No entry point allocates, synchronises or
 * reads device memory on the host.  All activation buffers are dtype `dtype`, row-major, contiguous. */
typedef struct bt_engine_desc {
    uint32_t struct_size;    /* = sizeof(bt_engine_desc), checked like bt_attn_args.struct_size */
    int32_t dtype;           /* bt_dtype of weights, activations and KV caches */
    int32_t L;               /* block_length */
    int32_t E;               /* token_decoder.n_expanded_emb (prefix rows per block) */
    int32_t vocab;
    int64_t B;               /* sequences in this engine */
    bt_decoder_dims bd_dims; /* block decoder (global) stack */
    bt_decoder_weights bd_w;
    bt_decoder_dims td_dims; /* token decoder (local) stack */
    bt_decoder_weights td_w;
    const void* emb_table;   /* embedder.embeddings.weight [vocab, emb_width], emb_width * L == bd_dims.hidden */
    int32_t emb_width;
    int32_t do_sample;       /* 0 greedy (bt_logits_argmax), 1 temperature + top-k multinomial (bt_topk_sample) */
    const void* exp_w;       /* token_decoder.expansion_layer.weight [E * H_td, H_bd] (Conv1d k=1 == Linear) */
    const void* exp_b;       /* [E * H_td] or NULL */
    const void* embed_in;    /* token_decoder.gpt_neox.embed_in.weight [vocab, H_td] */
    const void* embed_out;   /* token_decoder.embed_out.weight [vocab, H_td] */
    /* global KV cache K,V[layers][B][heads][g_nmax][d] + block-decoder RoPE tables */
    void* gk; void* gv; int64_t g_layer_stride; int64_t g_nmax; const float* g_cos; const float* g_sin;
    /* local KV cache K,V[layers][B][heads][l_nmax = E+L-1][d] + token-decoder RoPE tables and position policy table */
    void* lk; void* lv; int64_t l_layer_stride; int64_t l_nmax; const float* l_cos; const float* l_sin;
    const int32_t* l_pos_table;
    /* activations */
    void* x_blk;             /* [B, H_bd]  embedded block (block-decoder step input; overwritten) */
    void* ctx;               /* [B, H_bd]  context embedding = block-decoder output of the newest block */
    void* x_pre;             /* [B*E, H_td] expanded prefix rows */
    void* x_tok;             /* [B, H_td]  embed_in(previous token) */
    void* hid;               /* [B, H_td]  token-decoder output row */
    void* logits;            /* [B, vocab] */
    int64_t* raw;            /* [B, L] tokens of the block being decoded */
    int64_t* out_ids;        /* [B, nb_max, L] all blocks (prompt + generated) */
    int64_t nb_max;
    int32_t* n_cur;          /* device scalar: cached global positions == index of the block being decoded */
    int32_t* finished;       /* [B] */
    int32_t* finished_prev;  /* [B] */
    uint8_t* mask;           /* [B, mask_ld] block attention mask, or NULL when no key is ever masked (EOS disabled, no padding) */
    int64_t mask_ld;
    void* ws_tok; int64_t ws_tok_bytes;     /* >= bt_decoder_workspace_bytes(td_dims, dtype, B*E) */
    void* ws_step; int64_t ws_step_bytes;   /* >= bt_decoder_workspace_bytes(bd_dims, dtype, B) */
    void* exp_scratch; int64_t exp_scratch_bytes; /* optional split-K scratch of the expansion GEMM (may be NULL) */
    int32_t n_split;         /* split-N factor of the global attention step (1 = off) */
    int32_t top_k;
    void* split_ws; int64_t split_ws_bytes;
    float temperature;
    int32_t skip_logits;     /* != 0: greedy steps need not materialise `logits` (HF generate's output_scores = False): the logits
                                GEMM's epilogue hands bt_logits_argmax's pair reduction the row maxima instead */
    int64_t sample_row0;     /* RNG row index of this engine's sequence 0 (shards of one batch; 0 otherwise) */
    uint64_t seed;
    int32_t* sample_step;    /* device counter of sampling draws (RNG stream position); required when do_sample */
    int64_t config_pad;      /* token_decoder.config.pad_token_id (fills the unfilled tail of a partial block) */
    int64_t gen_eos;         /* generation_config.eos_token_id, < 0 = EOS handling disabled */
    int64_t gen_pad;         /* pad written after EOS */
    /* ---- since 0.2.1 (compaction of finished sequences; overflow guard) ---- */
    const int64_t* row_map;  /* [B] or NULL (identity): ORIGINAL batch row of engine row b after bt_compact_enqueue -- the row of
                                out_ids bt_finish_block writes and the RNG stream bt_topk_sample draws from (key = seed, step,
                                sample_row0 + row_map[b]); < 0 marks a padding row of a compacted engine (nothing is written) */
    bt_stream aux_stream;    /* optional second stream + two events for the parallel-residual branch of large stacks: handed to */
    void* ev_fork;           /* bt_decoder_forward through bt_attn_args (see there); all three or none */
    void* ev_join;
    int32_t* err_flag;       /* device int32 or NULL: sticky error bits set by the device-side guards (BT_DEV_ERR_*); the host reads
                                it whenever it synchronises anyway.  With it set, bt_roll_over_enqueue clamps *n_cur to g_nmax - 1
                                before anything is appended (a binder replaying one graph too many overwrites the last cache row
                                and gets the flag instead of writing past the cache) */
} bt_engine_desc;

enum bt_dev_err {
    BT_DEV_ERR_CACHE_FULL = 1, /* bt_roll_over_enqueue found *n_cur >= g_nmax */
    BT_DEV_ERR_BLOCK_RANGE = 2, /* bt_finish_block found *n_cur >= nb_max (out_ids) or >= mask_ld (mask): nothing was written */
    BT_DEV_ERR_HANDOFF_TIMEOUT = 4 /* a stream-K GEMM's consumer workgroup gave up waiting for its partner's partial tile: the
                                      activations of that pass (and every token derived from them) are incomplete -- discard them */
};

/* Prefill of sequences [b0, b0+nseq): embed their nb prompt blocks (block_ids [B, nb, L] int64 on device), run the block
 * decoder over them (causal, mode 1 = MFMA flash for bf16 when nb > 1) writing global KV rows 0..nb-1, and leave the last
 * hidden row of each sequence in ctx[b0 ..].  x: scratch [nseq*nb, H_bd]; ws: >= bt_decoder_workspace_bytes(bd, nseq*nb).
 * The caller sets *n_cur = nb, out_ids[:, :nb] = block_ids and the mask columns.  (block_transformer.py:355-380, first iteration) */
int bt_prefill_enqueue(const bt_engine_desc* e, const int64_t* block_ids, int32_t nb, int64_t b0, int64_t nseq, void* x,
                       void* ws, int64_t ws_bytes, bt_stream stream);
/* Token steps t in [t_begin, t_end) of the block being decoded (t_begin == 0 also runs the expansion and the E prefix
 * rows): token_decoder.generate (model/token_decoder/base.py:399-453 = L cached forwards + greedy/sample).  When
 * finish_count > 0 the block is closed afterwards with bt_finish_block(count = finish_count). */
int bt_token_block_enqueue(const bt_engine_desc* e, int32_t t_begin, int32_t t_end, int32_t finish_count, bt_stream stream);
/* Roll-over: embedder lookup+concat of the block just finished (zero rows for sequences finished before it), block-decoder
 * step at global cache row *n_cur (RoPE + append + attention over *n_cur + 1 positions) -> ctx, then *n_cur += 1.
 * (block_transformer.py:370-380,419-426) */
int bt_roll_over_enqueue(const bt_engine_desc* e, bt_stream stream);
/* One full block step = bt_token_block_enqueue(e, 0, count, count) followed, when roll != 0, by bt_roll_over_enqueue:
 * the unit that is hipGraph-captured. */
int bt_block_step_enqueue(const bt_engine_desc* e, int32_t count, int32_t roll, bt_stream stream);

/* Compaction of finished sequences between a token block and the roll-over (reference model/block_transformer.py:381-398,421:
 * only decode_mask rows are decoded; inference_demo.py:62-83 runs exactly this mode -- sampling, EOS on).  `live_rows` =
 * device array of n_live ASCENDING row indices of `src` that continue.  `dst` describes the engine that carries on: the SAME
 * global KV storage as src (gk / gv / g_layer_stride / g_nmax equal), dst->B in [n_live, src->B], and its OWN small state
 * buffers (mask, raw, finished, finished_prev, n_cur, row_map -- distinct from src's; all required except mask).
 *   - global K/V rows of sequence live_rows[i] move to sequence slot i, positions [0, *src->n_cur), layer by layer and head by
 *     head, IN PLACE: live_rows[i] >= i and ascending, so every workgroup walks i upward over its own byte range of a slot and
 *     never reads a slot it has already overwritten -- no temporary, no extra HBM;
 *   - dst state rows i < n_live: mask / raw rows of live_rows[i], finished = finished_prev = 0,
 *     row_map[i] = src->row_map ? src->row_map[live_rows[i]] : live_rows[i]  (compaction composes);
 *     rows n_live <= i < dst->B are padding: finished = finished_prev = 1, row_map = -1, mask row 1, raw 0;
 *   - *dst->n_cur = *src->n_cur.
 * Local KV, ctx, x_blk and the activation buffers carry no state across the roll-over: dst may share src's. */
int bt_compact_enqueue(const bt_engine_desc* src, const bt_engine_desc* dst, const int64_t* live_rows, int64_t n_live,
                       bt_stream stream);

/* ---- hipGraph helpers (capture the block step once, replay per block) ------------------------------------------ */
typedef struct bt_graph bt_graph;
int bt_graph_begin_capture(bt_stream stream);
int bt_graph_end_capture(bt_stream stream, bt_graph** out);
int bt_graph_launch(bt_graph* g, bt_stream stream);
int bt_graph_destroy(bt_graph* g);

/* ---- timing helpers for bench.py (HIP events on the launch stream) ---------------------------------------------- */
typedef struct bt_event bt_event;
int bt_event_create(bt_event** out);
int bt_event_record(bt_event* e, bt_stream stream);
int bt_event_elapsed_ms(bt_event* start, bt_event* stop, float* ms_out); /* synchronises on `stop` */
int bt_event_destroy(bt_event* e);

/* ---- built-in per-kernel-family timer (bench.py roofline leg; eager launches only, not inside graph capture) ----------
 * When enabled every op brackets its kernel(s) with a hipEvent pair on the launch stream; bt_prof_collect()
 * synchronises and returns accumulated milliseconds and launch counts per family. */
enum bt_prof_family {
    BT_PROF_LINEAR_MFMA = 0, BT_PROF_LINEAR_GENERIC = 1, BT_PROF_LAYERNORM = 2, BT_PROF_EMBED = 3, BT_PROF_ROPE = 4,
    BT_PROF_ATTN_ROWS = 5, BT_PROF_ATTN_FLASH = 6, BT_PROF_ARGMAX = 7, BT_PROF_SAMPLE = 8, BT_PROF_FINISH = 9,
    BT_PROF_NFAM = 10
};
int bt_prof_enable(int on);                                          /* clears previous records */
int bt_prof_collect(double* ms_out, int64_t* count_out, int n_fam); /* n_fam = BT_PROF_NFAM */
/* Per-launch records of the same timer, in launch order: start (ms after the first record's start), duration (ms) and
 * family of every bracketed launch since bt_prof_enable(1).  capacity = 0 only returns the count in *n_out.  Does not
 * clear the records (bt_prof_collect / bt_prof_enable do).  Replaces the role of torch.profiler's export_chrome_trace in
 * the reference's harness (measure_generation_time.py:150-155): the harness counterpart writes these as trace events. */
int bt_prof_trace(double* start_ms, double* dur_ms, int32_t* family, int64_t capacity, int64_t* n_out);

/* ---- empirical ceilings of the box for bench.py's `ceilings` object (the reference's harness measures its
 * numbers in-run the same way, measure_generation_time.py:143-161) -- measurement kernels, not on the decode path ----------------
 * bt_ceiling_copy: dst[0, bytes) = src with 16 bytes per lane (flags bit 0: non-temporal loads / stores): HBM rate = 2 * bytes / t.
 * bt_ceiling_mfma: `workgroups` x 4 waves each issue iters * 64 v_mfma_f32_16x16x32_bf16 on the 16-byte operand words found in
 *   `data` (>= workgroups * 256 * 128 bytes; register operands, no memory traffic in the loop); bt_ceiling_mfma_flops() is the flop count of such a launch;
 *   `out`: workgroups * 256 floats (keeps the results alive). */
int bt_ceiling_copy(void* dst, const void* src, int64_t bytes, int32_t flags, bt_stream stream);
int64_t bt_ceiling_mfma_flops(int32_t workgroups, int32_t iters);
int bt_ceiling_mfma(const void* data, int64_t data_bytes, void* out, int64_t out_bytes, int32_t workgroups, int32_t iters,
                    bt_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* BT_HIP_H */
