// Shared device/host helpers for libbt_hip.so (gfx950 only; wave = 64 lanes).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/bt_hip.h"

namespace bt {

// ---- error plumbing --------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define BT_CHECK_ARG(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            bt::set_error(__VA_ARGS__);         \
            return BT_ERR_ARG;                  \
        }                                       \
    } while (0)

#define BT_LAUNCH_CHECK(what)                               \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return bt::hip_fail(e__, what); \
    } while (0)

#define BT_TRY(expr)                 \
    do {                             \
        int rc__ = (expr);           \
        if (rc__ != BT_OK) return rc__; \
    } while (0)

// ---- element types ---------------------------------------------------------------------------------------------
struct bf16 {
    uint16_t x;
};

__host__ __device__ inline float bf16_to_f32(uint16_t h) {
    union {
        uint32_t u;
        float f;
    } c;
    c.u = ((uint32_t)h) << 16;
    return c.f;
}

// round-to-nearest-even, NaN preserved (same as torch's float -> bfloat16); branch-free (a divergent NaN branch per value
// showed up as ~100 exec-mask save/restore pairs in the flash / GEMM epilogues)
__host__ __device__ inline uint16_t f32_to_bf16(float f) {
    union {
        uint32_t u;
        float f;
    } c;
    c.f = f;
    const uint32_t u = c.u;
    const uint32_t rounded = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    const uint32_t nan = (u >> 16) | 0x0040u;
    return (uint16_t)(((u & 0x7fffffffu) > 0x7f800000u) ? nan : rounded);
}

template <typename T>
struct Elem;
template <>
struct Elem<float> {
    static constexpr int VEC = 4;  // elements per 16-byte vector
    __device__ static inline float load(const float* p) { return *p; }
    __device__ static inline void store(float* p, float v) { *p = v; }
};
template <>
struct Elem<bf16> {
    static constexpr int VEC = 8;
    __device__ static inline float load(const bf16* p) { return bf16_to_f32(p->x); }
    __device__ static inline void store(bf16* p, float v) { p->x = f32_to_bf16(v); }
};

// 16-byte vector load/store of Elem<T>::VEC elements to/from fp32 registers (pointer must be 16-byte aligned)
template <typename T>
__device__ inline void load_vec(const T* p, float (&v)[Elem<T>::VEC]);
template <>
__device__ inline void load_vec<float>(const float* p, float (&v)[4]) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <>
__device__ inline void load_vec<bf16>(const bf16* p, float (&v)[8]) {
    uint4 t = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xffff0000u);
    v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xffff0000u);
}
// streaming (non-temporal) variants: read-once K/V streams should not displace reusable lines in L2 / MALL
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
template <typename T>
__device__ inline void load_vec_nt(const T* p, float (&v)[Elem<T>::VEC]);
template <>
__device__ inline void load_vec_nt<float>(const float* p, float (&v)[4]) {
    const u32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    v[0] = __uint_as_float(t.x); v[1] = __uint_as_float(t.y); v[2] = __uint_as_float(t.z); v[3] = __uint_as_float(t.w);
}
template <>
__device__ inline void load_vec_nt<bf16>(const bf16* p, float (&v)[8]) {
    const u32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xffff0000u);
    v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xffff0000u);
}
template <typename T>
__device__ inline void store_vec(T* p, const float (&v)[Elem<T>::VEC]);
template <>
__device__ inline void store_vec<float>(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
// two fp32 -> packed bf16x2 (RNE) in ONE instruction: gfx950 v_cvt_pk_bf16_f32 (no clang builtin; operands are VGPRs)
__device__ inline uint32_t pack_bf16x2(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
template <>
__device__ inline void store_vec<bf16>(bf16* p, const float (&v)[8]) {
    uint4 t;
    t.x = pack_bf16x2(v[0], v[1]); t.y = pack_bf16x2(v[2], v[3]);
    t.z = pack_bf16x2(v[4], v[5]); t.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = t;
}

// ---- wave / block reductions (wave64) ----------------------------------------------------------------------------
template <int WIDTH = 64>
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int WIDTH = 64>
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int o = WIDTH / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// 16-lane butterfly sum on the DPP path (no LDS round trips): stage for stage the partners of wave_sum<16>'s xor 8 / 4 / 2 / 1 --
// row_ror:8 IS lane ^ 8 inside a 16-lane row; after it lanes j and j ^ 8 hold the same value, so row_ror:4 delivers the value
// lane ^ 4 holds; quad_perm gives ^ 2 and ^ 1 exactly -- and fp32 addition commutes, so the result is bit-identical to
// wave_sum<16> at ~1/10 of its latency (4 ds_bpermute round trips -> 4 VALU instructions).
__device__ inline float row16_sum_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    return v;
}

// ---- per-family kernel timer (core.hip) ------------------------------------------------------------------------
extern bool g_prof_on;
void prof_push(int family, hipEvent_t a, hipEvent_t b);
hipEvent_t prof_event();
struct ProfScope {
    int fam;
    hipStream_t s;
    hipEvent_t a = nullptr;
    ProfScope(int family, hipStream_t stream) : fam(family), s(stream) {
        if (g_prof_on) {
            a = prof_event();
            (void)hipEventRecord(a, s);
        }
    }
    ~ProfScope() {
        if (a) {
            hipEvent_t b = prof_event();
            (void)hipEventRecord(b, s);
            prof_push(fam, a, b);
        }
    }
};

int attention_impl(int dtype, int mode, int64_t B, int32_t q_len, int32_t heads, int32_t head_dim, const void* qkv,
                   int64_t ld_qkv, const void* k_cache, const void* v_cache, int64_t n_max, const int32_t* pos0_dev,
                   int32_t pos0_host, const uint8_t* mask, int64_t mask_ld, void* out, int64_t ldo, int32_t n_split, void* ws,
                   int64_t ws_bytes, const int32_t* seq_skip, bt_stream stream);

// ---- engine-side forms of bt_finish_block / bt_topk_sample (row_map of a compacted engine, device error flag) and the roll-over guard (misc.hip)
int finish_block_impl(int64_t B, int32_t L, int32_t count, int64_t* raw, int64_t* out_ids, int64_t nb_max, const int32_t* blk_dev,
                      int32_t* finished, int32_t* finished_prev, uint8_t* mask, int64_t mask_ld, int64_t config_pad, int64_t gen_eos,
                      int64_t gen_pad, const int64_t* row_map, int32_t* err_flag, bt_stream stream);
int topk_sample_impl(int dtype, int64_t M, int64_t V, const void* logits, int64_t ld, float temperature, int32_t top_k, uint64_t seed,
                     const int32_t* step_dev, int64_t row0, const int64_t* row_map, int64_t* ids_out, int64_t ids_stride,
                     bt_stream stream);
int guard_ncur(int32_t* n_cur, int64_t n_max, int32_t* err_flag, bt_stream stream);

// ---- <= 4 rows: QKV (LN1, RoPE, append) and h->4h (LN2, GELU) GEMVs of one layer as ONE launch (gemm.hip) ----------------------
bool few_rows_dual_ok(int64_t M, int64_t H, int64_t I, int head_dim, int rot);
int few_rows_qkv_h4(hipStream_t s, int64_t B, int q_len, int heads, int head_dim, const void* x, int64_t ldx, const void* ln1_w,
                    const void* ln1_b, const void* ln2_w, const void* ln2_b, float eps, const void* qkv_w, const void* qkv_b,
                    void* qkv, int64_t ld_qkv, void* kc, void* vc, int64_t n_max, const int32_t* pos0_dev, int pos0_host,
                    const int32_t* pos_table, const float* cos_tab, const float* sin_tab, const void* h4_w, const void* h4_b,
                    void* h4, int64_t I);

// ---- <= 4 rows, token-decoder local window: attention computed inside the residual-stream GEMV (gemm.hip, round 4) --------------
bool few_rows_attn_res_ok(int64_t M, int q_len, int heads, int head_dim, int64_t H, int64_t I, int pos0_host, int64_t n_max);
int few_rows_attn_res(hipStream_t s, int64_t M, int q_len, int heads, int64_t I, const void* h4, const void* h_w, const void* h_b,
                      const void* qkv, int64_t ld_qkv, const void* kc, const void* vc, int64_t n_max, int pos0_host,
                      const void* dense_w, const void* dense_b, void* x, int64_t ldx);

// ---- mid-batch (3 .. 512 rows) decoder-layer pieces (gemm.hip), used by bt_decoder_forward (engine.hip) -------------------
bool mid_rows_ok(int64_t M, int64_t H, int64_t I, int head_dim, int rot);
int64_t mid_slab_bytes(int64_t M, int64_t H, int64_t I);
bool mid_tall_layer_ok(int64_t H, int64_t I, const void* qkv_w, const void* h4_w, const void* h_w, const void* dense_w);   // > 256 rows
int mid_qkv(hipStream_t s, int64_t B, int q_len, int heads, int head_dim, const void* x, int64_t ldx, const void* w,
            const void* bias, void* qkv, int64_t ld_qkv, void* kc, void* vc, int64_t n_max, const int32_t* pos0_dev, int pos0_host,
            const int32_t* pos_table, const float* cos_tab, const float* sin_tab, void* slabs, int64_t slab_bytes, int* counters);
int mid_qkv_gelu(hipStream_t s, int64_t B, int q_len, int heads, int head_dim, const void* x1, const void* qkv_w, const void* qkv_b,
                 void* qkv, int64_t ld_qkv, void* kc, void* vc, int64_t n_max, const int32_t* pos0_dev, int pos0_host,
                 const int32_t* pos_table, const float* cos_tab, const float* sin_tab, const void* x2, const void* h4_w,
                 const void* h4_b, void* h4, int64_t I, void* slabs, int64_t slab_bytes, int* counters, bool* done, void* att,
                 int64_t ld_att, bool* attn_done);   // att != NULL: the caller allows the local-window attention inside the reduce launch
int mid_gelu(hipStream_t s, int64_t M, int64_t N, int64_t K, const void* x, int64_t ldx, const void* w, const void* bias, void* out,
             int64_t ldo, void* slabs, int64_t slab_bytes, int* counters);
constexpr int64_t MID_COUNTER_BYTES = 4096;   // one int per 64 x 64 tile of a <= 256-row GEMM with N <= 16384
int mid_res_ln(hipStream_t s, int64_t M, int64_t N, int64_t K1, const void* x1, int64_t ldx1, const void* w1, const void* b1,
               int64_t K2, const void* x2, int64_t ldx2, const void* w2, const void* b2, void* x, int64_t ldx, const void* ln1_w,
               const void* ln1_b, void* ln1_out, const void* ln2_w, const void* ln2_b, void* ln2_out, int64_t ld_ln, float eps,
               void* slabs, int64_t slab_bytes);

// ---- stream-K items GEMM (gemm.hip, round 5): scratch sizing / flag clearing, and the engine-side forms of the entry points that
// may dispatch to it (`sk_clean`: the flag words of `scratch` are known to be zero -- bt_decoder_forward clears them once per pass)
int64_t sk_scratch_bytes();
bool sk_linear2_ok(int64_t M, int64_t N, int64_t K1, int64_t K2, int64_t scratch_bytes);
int sk_clear_flags(void* scratch, int64_t scratch_bytes, hipStream_t s);                              // hand-off flags only (per launch)
int tall_status(const int* counters, int32_t* err_flag, hipStream_t s, uint32_t tag = 0);          // fused mid-batch launch: sticky time-out word -> err_flag
int zero_words(void* p, int words, hipStream_t s);                                                    // one-workgroup clearing kernel (no memset nodes in captured graphs)
int sk_reset(void* scratch, int64_t scratch_bytes, hipStream_t s);                                    // flags + sticky time-out word
int sk_status(const void* scratch, int64_t scratch_bytes, int32_t* err_flag, hipStream_t s, uint32_t tag = 0);          // sticky word -> BT_DEV_ERR_HANDOFF_TIMEOUT
int linear_ws_impl(int dtype, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* W, const void* bias,
                   const void* residual, int64_t ldr, void* out, int64_t ldo, int act, void* scratch, int64_t scratch_bytes,
                   bool sk_clean, bt_stream stream);
int qkv_rope_append_impl(int dtype, int64_t B, int32_t q_len, int32_t heads, int32_t head_dim, int32_t rot, const void* x, int64_t ldx,
                         const void* qkv_w, const void* qkv_b, void* qkv, int64_t ld_qkv, void* k_cache, void* v_cache, int64_t n_max,
                         const int32_t* pos0_dev, int32_t pos0_host, const int32_t* pos_table, const float* cos_tab,
                         const float* sin_tab, void* scratch, int64_t scratch_bytes, bool sk_clean, bt_stream stream,
                         const void* ln_stats = nullptr, const float* ln_s = nullptr, const float* ln_c = nullptr);   // folded LayerNorm: x = raw rows, qkv_w = folded weight
#ifdef BT_DEBUG_HOOKS   // folded-LayerNorm experiment (debug library only)
int rowstats_impl(int dtype, int64_t M, int64_t H, const void* x, int64_t ldx, float eps, void* stats, bt_stream stream);
bool lnfold_shape_ok(int64_t M, int64_t N, int64_t K);
#endif
int linear2_impl(int dtype, int64_t M, int64_t N, int64_t K1, const void* x1, int64_t ldx1, const void* W1, const void* bias1, int64_t K2,
                 const void* x2, int64_t ldx2, const void* W2, const void* bias2, const void* residual, int64_t ldr, void* out,
                 int64_t ldo, void* scratch, int64_t scratch_bytes, bool sk_clean, bt_stream stream);

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline size_t dtype_size(int dtype) { return dtype == BT_BF16 ? 2 : 4; }

}  // namespace bt
