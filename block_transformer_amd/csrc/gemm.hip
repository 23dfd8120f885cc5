// bt_linear / bt_linear_ws / bt_qkv_rope_append: out[M,N] = act(A[M,K] . W[N,K]^T + bias) + residual  (HF nn.Linear layout:
// both operands K-contiguous), bf16 on MFMA, fp32 on a plain tiled kernel (the parity path).
//
// Kernel family (dispatch in linear_impl; every bf16 kernel streams A and W tiles HBM/L2 -> LDS with 16-byte
// global_load_lds, XOR-swizzles the 16-byte slots of an LDS row so ds_read_b128 fragment reads are conflict-free, feeds the
// WEIGHT tile as the MFMA "A" operand -- so a lane ends up with 4 consecutive output columns of one row -- and remaps
// block ids XCD-aware with grouping along M so an XCD's L2 sees a compact set of A / W panels):
//  * gemm_bf16_mfma256pp_kernel<MODE, Mma16>  PRODUCTION for grids >= ~240 tiles: persistent 256x256x64 tiles, 8 waves, half-
//      tile LDS ring, two staggered wave groups, per-wave epilogue (see the block comment at the kernel).  MODE = plain /
//      GELU / residual / fused QKV (RoPE + KV-cache append in the epilogue).  Mma32 = 32x32x16 MFMA A/B variant.
//  * gemm_bf16_mfma256p2_kernel               one 256x256 tile per workgroup, same ring (fallback: N % 8 != 0, K < 256; A/B).
//  * gemm_bf16_mfma256_kernel<STAGGER, ABL>   the earlier four-phase ring + timing ablations (A/B probes only).
//  * gemm_bf16_mfma_kernel<TBM, TBN, SPLITK>  128x128 / 128x64 / 64x64 tiles, 4 waves, double-buffered: grids that would
//      leave CUs idle with 256^2 tiles (decode GEMMs at M = 4096 with N = 2048 / 6144, small batches); SPLITK + 
//      splitk_reduce_kernel for the weight-streaming regime (M <~ 256 rows).
//  * gemm_generic_kernel<T>                   any dtype / shape / stride: 64x64x16 LDS tile, fp32 FMA accumulation (fp32
//      parity path with erff GELU; bf16 shapes with K % 64 != 0).
//
// Roofline: MFMA-bound for M >~ 300 rows (2*M*N*K flop vs 2*(M*K + N*K + M*N) bytes); HBM-bound on W below that.
#include "bt_common.h"
#include <type_traits>

#include <algorithm>

namespace bt {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

// raw (untyped, unbounded) buffer descriptor whose base is a wave-uniform pointer: offsets are 32-bit, so the base is the
// origin of the tile being written, not of the tensor
__device__ inline __amdgpu_buffer_rsrc_t tile_rsrc(const char* base_uniform) {
    const uint64_t a = reinterpret_cast<uint64_t>(base_uniform);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}

__device__ inline float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// GELU for the bf16 MFMA epilogues.  erff costs ~45 VALU instructions per value (measured: a K = 2048 GEMM with GELU ran
// 33 % slower than without), so the bf16 path evaluates  gelu(x) = x * Phi(x)  through the scaled normal tail
//     Phi(-a) = exp(-a^2 / 2) * R(a),   a = |x|,   R = degree-11 polynomial in u = a/3 - 1 (relative minimax fit on [0, 6])
//     gelu(x) = x * Phi(-a)          (x < 0)        = x * (1 - Phi(-a))   (x >= 0)
// which keeps RELATIVE accuracy in the negative tail (no 1 + erf cancellation).  In fp32 arithmetic the result is within
// 8.5e-6 relative of the exact erf GELU for x > -6 (|gelu| < 6e-9 beyond, absolute error < 8e-9): 1/230 of a bf16
// half-ulp (round 6: degree 12 -> 11, polynomial error 2.2e-6 -> 7.7e-6 against ~3e-6 of fp32 evaluation noise -- on every bf16
// input in [-9, 9] both round to the correctly rounded bf16 value; a degree-10 fit, 2.7e-5, would flip 0.03 % of them, degree 9 0.5 %).
// The fp32 parity path (gemm_generic_kernel) keeps erff.  Two values per call so the polynomial runs on the packed-fp32 pipe
// (v_pk_fma_f32): 28 instruction slots per PAIR of values (13 v_pk_fma + 3 v_pk_mul + 2 v_min + 2 v_max + 2 quarter-rate v_exp),
// round 5: 35.
typedef float v2f __attribute__((ext_vector_type(2)));

// min(|x|, 6) and max(x, 0) as single instructions: the builtins (fminf / fmaxf / fabsf) make hipcc quiet a possible signalling NaN
// first -- a v_max_f32 x, x in front of every min / max -- and materialise |x| with a v_and: 8 of the 35 VALU slots a pair of values
// cost (ISA of the round-5 epilogue: 14 v_pk_fma + 3 v_pk_mul + 6 v_max + 2 v_min + 2 v_and + 2 quarter-rate v_exp).  Plain asm (not
// volatile: the scheduler may move it); 6.0 is not an inline constant of the VOP3 encoding -> one SGPR for the whole kernel.
__device__ inline float min_abs_f32(float x, float hi) {
    float r;
    asm("v_min_f32_e64 %0, |%1|, %2" : "=v"(r) : "v"(x), "s"(hi));
    return r;
}
__device__ inline float relu_f32(float x) {
    float r;
    asm("v_max_f32_e32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}

__device__ inline v2f gelu_tail2(v2f x) {
    const v2f a = {min_abs_f32(x.x, 6.0f), min_abs_f32(x.y, 6.0f)};
    const v2f u = a * (1.0f / 3.0f) - 1.0f;
    v2f r = u * -3.976902133e-03f + 8.081740986e-03f;
    r = r * u + -2.753396143e-03f;
    r = r * u + 1.759199353e-03f;
    r = r * u + -1.408788524e-02f;
    r = r * u + 2.432763856e-02f;
    r = r * u + -3.125067898e-02f;
    r = r * u + 4.423156669e-02f;
    r = r * u + -6.233500973e-02f;
    r = r * u + 8.247426398e-02f;
    r = r * u + -1.032048037e-01f;
    r = r * u + 1.215130848e-01f;
    const v2f t = x * x * -0.72134752044448170368f;     // -x^2 / 2 * log2(e)  (unclamped: exp2 underflows to 0 beyond |x| ~ 14)
    const v2f e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    const v2f q = e * r;                                 // Phi(-|x|)
    // x * Phi(x) = relu(x) - |x| * Phi(-|x|): the same value as selecting q or 1 - q by sign and multiplying (identical for x < 0,
    // x - x q for x >= 0), in one packed FMA instead of two compares, two selects, a subtraction and a multiply.  The clamped |x| is
    // used for the product: identical bits for |x| <= 6, and beyond that the term is < 1e-8 either way (q < 2e-9)
    const v2f relu = {relu_f32(x.x), relu_f32(x.y)};
    return relu - a * q;
}

#ifdef BT_DEBUG_HOOKS   // the round-5 form (degree 12, builtin min / max / abs: 35 slots per pair) -- A/B reference of the round-6 rewrite (variant 30)
__device__ inline v2f gelu_tail2_r5(v2f x) {
    const v2f ax = {__builtin_fabsf(x.x), __builtin_fabsf(x.y)};
    const v2f a = {__builtin_fminf(ax.x, 6.0f), __builtin_fminf(ax.y, 6.0f)};
    const v2f u = a * (1.0f / 3.0f) - 1.0f;
    v2f r = u * 2.359363747e-03f + -4.802452166e-03f;
    r = r * u + 1.242561118e-03f;
    r = r * u + -5.446036305e-04f;
    r = r * u + 9.139311313e-03f;
    r = r * u + -1.622305936e-02f;
    r = r * u + 2.069297082e-02f;
    r = r * u + -3.035567151e-02f;
    r = r * u + 4.502980965e-02f;
    r = r * u + -6.248621239e-02f;
    r = r * u + 8.241136925e-02f;
    r = r * u + -1.031977833e-01f;
    r = r * u + 1.215138655e-01f;
    const v2f t = ax * ax * -0.72134752044448170368f;
    const v2f e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    const v2f q = e * r;
    const v2f relu = {__builtin_fmaxf(x.x, 0.0f), __builtin_fmaxf(x.y, 0.0f)};
    return relu - ax * q;
}
__device__ inline void gelu4_r5(float (&v)[4]) {
    const v2f lo = gelu_tail2_r5(v2f{v[0], v[1]}), hi = gelu_tail2_r5(v2f{v[2], v[3]});
    v[0] = lo.x; v[1] = lo.y; v[2] = hi.x; v[3] = hi.y;
}
#endif

__device__ inline void gelu4(float (&v)[4]) {
    const v2f lo = gelu_tail2(v2f{v[0], v[1]}), hi = gelu_tail2(v2f{v[2], v[3]});
    v[0] = lo.x; v[1] = lo.y; v[2] = hi.x; v[3] = hi.y;
}

// ---- shared epilogue ---------------------------------------------------------------------------------------------------
// Optional fused "QKV mode" (bt_qkv_rope_append): the GEMM is the GPT-NeoX query_key_value projection; instead of
// writing [M, 3H] and running a separate RoPE/append pass, the epilogue rotates q and k in registers (the rotate-half
// partner of column e < 16 is column e + 16 = the NEXT 16-column fragment, same lane, same register index) and scatters
// k (rotated) and v straight into the KV cache; only q goes to the qkv buffer.  Requires head_dim % 32 == 0, rot == 32.
struct QkvEpi {
    bf16* kc;
    bf16* vc;
    int64_t n_max;
    const int32_t* pos0_dev;
    const int32_t* pos_table;
    const float* cos_tab;
    const float* sin_tab;
    int32_t pos0_host;
    int32_t q_len;
    int32_t heads;
    int32_t head_dim;
    int32_t enabled;
};

struct EpiArgs {
    const bf16* bias;
    const bf16* bias2;   // optional second bias row (two projections summed by one GEMM); LDS-staged epilogues only
    const bf16* res;
    int64_t ldr;
    bf16* out;
    int64_t ldo;
    int M, N, act;
    QkvEpi q;
    uint2* amax;      // PP_ARGMAX: per (row, 64-column slab) {bits of the largest bf16-rounded value, its column}; [M][amax_ld]
    int amax_ld;
    int slab_wt;      // split-K slab kernels: fp32 partial tiles stored write-through (sc1) instead of left dirty in L2 for the end-of-kernel release
    // LayerNorm folded into the GEMM (round 5; persistent kernel, EABL bit 32): the GEMM runs on the RAW rows x with the weight
    // W' = W diag(gamma), and the epilogue turns  acc = W' x  into  W (gamma (x - mu) rstd + beta) + b  =  rstd (acc - mu s) + c
    // with  s[n] = sum_k W'[n, k]  and  c[n] = (W beta)[n] + b[n]:
    const float2* ln_stats;   // [M] (mu, rstd) of the rows (bt_rowstats), or NULL
    const float* ln_s;        // [N]
    const float* ln_c;        // [N]
};

__device__ inline void epi_store4(const EpiArgs& e, int m, int n, float (&v)[4], bool vec_ok) {
    if (vec_ok) {
        if (e.bias) {
            const uint2 b2 = *reinterpret_cast<const uint2*>(e.bias + n);
            v[0] += __uint_as_float(b2.x << 16); v[1] += __uint_as_float(b2.x & 0xffff0000u);
            v[2] += __uint_as_float(b2.y << 16); v[3] += __uint_as_float(b2.y & 0xffff0000u);
        }
        if (e.act == BT_ACT_GELU) {
            gelu4(v);
        }
        if (e.res) {
            const uint2 r2 = *reinterpret_cast<const uint2*>(e.res + (int64_t)m * e.ldr + n);
            v[0] += __uint_as_float(r2.x << 16); v[1] += __uint_as_float(r2.x & 0xffff0000u);
            v[2] += __uint_as_float(r2.y << 16); v[3] += __uint_as_float(r2.y & 0xffff0000u);
        }
        uint2 o2;
        o2.x = pack_bf16x2(v[0], v[1]);
        o2.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>(e.out + (int64_t)m * e.ldo + n) = o2;
    } else {
        for (int r = 0; r < 4; ++r) {
            if (n + r >= e.N) break;
            float x = v[r];
            if (e.bias) x += bf16_to_f32(e.bias[n + r].x);
            if (e.act == BT_ACT_GELU) x = gelu_tail2(v2f{x, x}).x;
            if (e.res) x += bf16_to_f32(e.res[(int64_t)m * e.ldr + n + r].x);
            e.out[(int64_t)m * e.ldo + n + r].x = f32_to_bf16(x);
        }
    }
}

// one output row m, two adjacent 16-column fragments starting at the 32-aligned column n_lo - (lane >> 4) * 4
__device__ inline void epilogue_pair(const EpiArgs& e, int m, int n_lo, const f32x4_t& a_lo, const f32x4_t& a_hi,
                                     bool vec_ok) {
    if (m >= e.M) return;
    float lo[4] = {a_lo[0], a_lo[1], a_lo[2], a_lo[3]};
    float hi[4] = {a_hi[0], a_hi[1], a_hi[2], a_hi[3]};
    if (!e.q.enabled) {
        if (n_lo < e.N) epi_store4(e, m, n_lo, lo, vec_ok);
        if (n_lo + 16 < e.N) epi_store4(e, m, n_lo + 16, hi, vec_ok);
        return;
    }
    // QKV mode (N = heads * 3 * D, D % 32 == 0, rot == 32; N % 4 == 0 holds)
    const int D = e.q.head_dim;
    const int head = n_lo / (3 * D);
    const int within = n_lo - head * 3 * D;
    const int part = within / D;              // 0 q, 1 k, 2 v
    const int el = within - part * D;         // element of the lo fragment; the hi fragment holds el + 16
    if (e.bias) {
        const uint2 b2 = *reinterpret_cast<const uint2*>(e.bias + n_lo);
        const uint2 b3 = *reinterpret_cast<const uint2*>(e.bias + n_lo + 16);
        lo[0] += __uint_as_float(b2.x << 16); lo[1] += __uint_as_float(b2.x & 0xffff0000u);
        lo[2] += __uint_as_float(b2.y << 16); lo[3] += __uint_as_float(b2.y & 0xffff0000u);
        hi[0] += __uint_as_float(b3.x << 16); hi[1] += __uint_as_float(b3.x & 0xffff0000u);
        hi[2] += __uint_as_float(b3.y << 16); hi[3] += __uint_as_float(b3.y & 0xffff0000u);
    }
    const int64_t b = m / e.q.q_len;
    const int t = m - (int)b * e.q.q_len;
    const int pos0 = e.q.pos0_dev ? *e.q.pos0_dev : e.q.pos0_host;
    if (part < 2 && el < 16) {  // first 32 dims of q / k: rotate pairs (el + r, el + r + 16)
        const int rpos = e.q.pos_table ? e.q.pos_table[pos0 + t] : pos0 + t;
        const float4 c4 = *reinterpret_cast<const float4*>(e.q.cos_tab + (int64_t)rpos * 16 + el);
        const float4 s4 = *reinterpret_cast<const float4*>(e.q.sin_tab + (int64_t)rpos * 16 + el);
        const float c[4] = {c4.x, c4.y, c4.z, c4.w}, sn[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x1 = lo[r], x2 = hi[r];
            lo[r] = x1 * c[r] - x2 * sn[r];
            hi[r] = x2 * c[r] + x1 * sn[r];
        }
    }
    uint2 o_lo, o_hi;
    o_lo.x = pack_bf16x2(lo[0], lo[1]); o_lo.y = pack_bf16x2(lo[2], lo[3]);
    o_hi.x = pack_bf16x2(hi[0], hi[1]); o_hi.y = pack_bf16x2(hi[2], hi[3]);
    if (part == 0) {
        bf16* o = e.out + (int64_t)m * e.ldo + n_lo;
        *reinterpret_cast<uint2*>(o) = o_lo;
        *reinterpret_cast<uint2*>(o + 16) = o_hi;
    } else {
        bf16* cache = part == 1 ? e.q.kc : e.q.vc;
        bf16* o = cache + ((b * e.q.heads + head) * e.q.n_max + pos0 + t) * D + el;
        *reinterpret_cast<uint2*>(o) = o_lo;
        *reinterpret_cast<uint2*>(o + 16) = o_hi;
    }
}

// ---- LDS-staged epilogue (MFMA kernels) -----------------------------------------------------------------------------------
// The accumulator layout gives a lane 4 consecutive columns of one row: storing from registers is 8 bytes per lane in
// 32-byte row segments, which the memory system handles at < 2 TB/s (measured: 28 % of a K = 2048 GEMM).  Instead the
// block's output tile goes through LDS (free after the K-loop): phase 1 writes bf16(acc + bias [GELU] [RoPE]) into a padded
// row-major tile, phase 2 reads it back 16 bytes per lane along rows and issues fully coalesced 16-byte global stores
// (adding the residual on the way; QKV mode scatters k / v chunks to the cache rows instead).
template <int COLS>
struct CTile {
    static constexpr int PITCH = COLS * 2 + 16;  // bytes; +16 breaks the power-of-two row stride (ds_write_b64 conflicts)
};

template <int COLS, bool BIAS = true>
__device__ inline void epi_to_lds(const EpiArgs& e, char* ctile, int ml, int nl, int m, int n_lo, const f32x4_t& a_lo,
                                  const f32x4_t& a_hi) {
    float lo[4] = {a_lo[0], a_lo[1], a_lo[2], a_lo[3]};
    float hi[4] = {a_hi[0], a_hi[1], a_hi[2], a_hi[3]};
    if (BIAS) {
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const bf16* bp = which ? e.bias2 : e.bias;
            if (!bp) continue;
            if (n_lo < e.N) {
                const uint2 b2 = *reinterpret_cast<const uint2*>(bp + n_lo);
                lo[0] += __uint_as_float(b2.x << 16); lo[1] += __uint_as_float(b2.x & 0xffff0000u);
                lo[2] += __uint_as_float(b2.y << 16); lo[3] += __uint_as_float(b2.y & 0xffff0000u);
            }
            if (n_lo + 16 < e.N) {
                const uint2 b3 = *reinterpret_cast<const uint2*>(bp + n_lo + 16);
                hi[0] += __uint_as_float(b3.x << 16); hi[1] += __uint_as_float(b3.x & 0xffff0000u);
                hi[2] += __uint_as_float(b3.y << 16); hi[3] += __uint_as_float(b3.y & 0xffff0000u);
            }
        }
    }
    if (e.act == BT_ACT_GELU) {
        gelu4(lo);
        gelu4(hi);
    }
    if (e.q.enabled && m < e.M) {
        const int D = e.q.head_dim;
        const int head = n_lo / (3 * D);
        const int within = n_lo - head * 3 * D;
        const int part = within / D;
        const int el = within - part * D;
        if (part < 2 && el < 16) {
            const int64_t b = m / e.q.q_len;
            const int t = m - (int)b * e.q.q_len;
            const int pos0 = e.q.pos0_dev ? *e.q.pos0_dev : e.q.pos0_host;
            const int rpos = e.q.pos_table ? e.q.pos_table[pos0 + t] : pos0 + t;
            const float4 c4 = *reinterpret_cast<const float4*>(e.q.cos_tab + (int64_t)rpos * 16 + el);
            const float4 s4 = *reinterpret_cast<const float4*>(e.q.sin_tab + (int64_t)rpos * 16 + el);
            const float c[4] = {c4.x, c4.y, c4.z, c4.w}, sn[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x1 = lo[r], x2 = hi[r];
                lo[r] = x1 * c[r] - x2 * sn[r];
                hi[r] = x2 * c[r] + x1 * sn[r];
            }
        }
    }
    uint2 o_lo, o_hi;
    o_lo.x = pack_bf16x2(lo[0], lo[1]); o_lo.y = pack_bf16x2(lo[2], lo[3]);
    o_hi.x = pack_bf16x2(hi[0], hi[1]); o_hi.y = pack_bf16x2(hi[2], hi[3]);
    char* p = ctile + ml * CTile<COLS>::PITCH + nl * 2;
    *reinterpret_cast<uint2*>(p) = o_lo;
    *reinterpret_cast<uint2*>(p + 32) = o_hi;
}

__device__ inline uint4 add_bf16x8(uint4 a, uint4 b) {
    uint4 r;
    r.x = pack_bf16x2(__uint_as_float(a.x << 16) + __uint_as_float(b.x << 16),
                      __uint_as_float(a.x & 0xffff0000u) + __uint_as_float(b.x & 0xffff0000u));
    r.y = pack_bf16x2(__uint_as_float(a.y << 16) + __uint_as_float(b.y << 16),
                      __uint_as_float(a.y & 0xffff0000u) + __uint_as_float(b.y & 0xffff0000u));
    r.z = pack_bf16x2(__uint_as_float(a.z << 16) + __uint_as_float(b.z << 16),
                      __uint_as_float(a.z & 0xffff0000u) + __uint_as_float(b.z & 0xffff0000u));
    r.w = pack_bf16x2(__uint_as_float(a.w << 16) + __uint_as_float(b.w << 16),
                      __uint_as_float(a.w & 0xffff0000u) + __uint_as_float(b.w & 0xffff0000u));
    return r;
}

// phase 2: ROWS x COLS tile (bf16, padded pitch) -> global, 16 bytes per lane along rows.  Requires N % 8 == 0.
template <int ROWS, int COLS, int NT>
__device__ inline void store_tile_from_lds(const EpiArgs& e, const char* ctile, int m0, int n0, int tid) {
    constexpr int CPR = COLS / 8;
    for (int idx = tid; idx < ROWS * CPR; idx += NT) {
        const int r = idx / CPR, c = idx - r * CPR;
        const int m = m0 + r, n = n0 + c * 8;
        if (m >= e.M || n >= e.N) continue;
        uint4 v = *reinterpret_cast<const uint4*>(ctile + r * CTile<COLS>::PITCH + c * 16);
        if (!e.q.enabled) {
            if (e.res) v = add_bf16x8(v, *reinterpret_cast<const uint4*>(e.res + (int64_t)m * e.ldr + n));
            *reinterpret_cast<uint4*>(e.out + (int64_t)m * e.ldo + n) = v;
        } else {
            const int D = e.q.head_dim;
            const int head = n / (3 * D);
            const int within = n - head * 3 * D;
            const int part = within / D;
            const int el = within - part * D;
            if (part == 0) {
                *reinterpret_cast<uint4*>(e.out + (int64_t)m * e.ldo + n) = v;
            } else {
                const int64_t b = m / e.q.q_len;
                const int t = m - (int)b * e.q.q_len;
                const int pos0 = e.q.pos0_dev ? *e.q.pos0_dev : e.q.pos0_host;
                bf16* cache = part == 1 ? e.q.kc : e.q.vc;
                *reinterpret_cast<uint4*>(cache + ((b * e.q.heads + head) * e.q.n_max + pos0 + t) * D + el) = v;
            }
        }
    }
}

struct LnPro {
    const bf16* w;
    const bf16* b;
    float eps;
};

// optional second (x2, W2) operand pair accumulated onto the same outputs: out = x . W^T + x2 . W2^T + bias + bias2 [+ res]
// (the two residual-stream projections of a parallel-residual layer -- attention output and MLP down -- in one launch)
struct Seg2 {
    const bf16* x;
    const bf16* w;
    const bf16* bias;
    int64_t ldx, ldw;
    int K;
};

// the token decoder's local-window attention as the PRODUCER of a GEMV's second operand (gemv_body<..., ATT>): the window is so
// small (<= 12 cached rows of 256 B per head) that every workgroup of the residual-stream GEMV recomputes all heads from L2
// instead of waiting for a separate attention launch
struct LocalAtt {
    const bf16* qkv;   // [M, ld_qkv]: rotated q of the new rows
    const bf16* kc;    // local cache K,V[B][heads][n_max][128] of this layer
    const bf16* vc;
    int64_t ld_qkv, n_max;
    int pos0, q_len, heads;
    float scale;
};

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int GROUP_M = 8;

// logical tile id for this workgroup: bijective XCD remap (cdna guide T1) then GROUP_M grouping
__device__ inline void tile_coords(int tiles_m, int tiles_n, int& tm, int& tn) {
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int group_sz = GROUP_M * tiles_n;
    const int g = id / group_sz;
    const int first_m = g * GROUP_M;
    const int gm = min(tiles_m - first_m, GROUP_M);
    const int in_g = id - g * group_sz;
    tm = first_m + in_g % gm;
    tn = in_g / gm;
}

// issue the LDS-DMA for one ROWS-row x 64-col bf16 tile: each wave-instruction moves 8 rows of 128 B
template <int ROWS>
__device__ inline void stage_tile(const bf16* __restrict__ src, int64_t ld, int row0, int row_max, int k0, char* lds_tile,
                                  int wave, int lane) {
    constexpr int NI = ROWS / 32;  // instructions per wave (4 waves)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int rbase = (wave * NI + i) * 8;
        const int row = rbase + (lane >> 3);
        const int gslot = (lane & 7) ^ (row & 7);
        int grow = row0 + row;
        grow = grow < row_max ? grow : row_max - 1;
        const bf16* g = src + (int64_t)grow * ld + k0 + gslot * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(lds_tile + rbase * 128), 16, 0, 0);
    }
}

__device__ inline bf16x8_t read_frag(const char* lds_tile, int row_base, int kk, int lane) {
    const int row = row_base + (lane & 15);
    const int slot = (kk * 4 + (lane >> 4)) ^ (row & 7);
    return *reinterpret_cast<const bf16x8_t*>(lds_tile + row * 128 + slot * 16);
}

// TBM x TBN x 64 tile, 4 waves (2 x 2); 128x128 for full grids, 128x64 / 64x64 when the grid would not fill 256 CUs
// SPLITK: blockIdx.y selects a K slice; the fp32 partial tile goes to `partial[slice][M][N]` and splitk_reduce_kernel
// applies the epilogue.  This is the weight-streaming ("skinny") regime: M <= ~256 rows, where one pass over W is the cost
// and a tile-per-CU grid would leave most CUs idle (M=1, N=2048, K=8192: 32 tiles -> 256 workgroups with 8 slices).
__device__ inline void cbarrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// NS: depth of the LDS ring.  NS == 2 is the classic double buffer (load t+1 while computing t, drain every step); NS >= 3
// keeps NS - 2 further K-tiles of LDS-DMA in flight ACROSS the per-step barrier (counted s_waitcnt vmcnt, raw s_barrier), which
// is what a latency-bound small grid needs: with one or two workgroups per CU nothing else hides the ~1-2 us L2/HBM round trip.
//   RAW: before a wave reads tile t it has waited for its own DMA share of tile t (in-order vmcnt: at most the DMAs of the
//        younger tiles stay outstanding) and passed the barrier every other wave reaches after doing the same;
//   WAR: tile t + NS - 1 goes into the buffer tile t - 1 was read from, and is issued after that barrier, i.e. after every
//        wave has issued all MFMAs of step t - 1 (which consumed its ds_reads of that buffer).
template <int N>
__device__ inline void wait_vmcnt_n() {
    static_assert(N >= 0 && N <= 48 && N % 2 == 0, "unsupported vmcnt immediate");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// FUSE (with SPLITK): the split-K reduction happens INSIDE the launch -- every slice publishes its fp32 slab, the slice that
// arrives last at the tile's counter sums all slabs in slice order (bit-identical to splitk_reduce_kernel, whatever the
// arrival order) and runs the normal fused epilogue (bias / GELU / residual / RoPE + KV append); with one slice there is no
// slab at all.  Publication follows the agent-scope recipe of the CDNA guide (section 5, "in-launch split-K reduction"):
// plain slab stores -> every wave s_waitcnt vmcnt(0) -> barrier -> lane 0: release fence, vmcnt(0), relaxed agent
// fetch_add; the last arriver: lane 0 acquire fence -> barrier -> plain loads.  `counters` (one int per tile) must be zero at
// launch; the last arriver puts its counter back to zero (agent-scope store), bt_decoder_forward also clears them per pass.
// The body is a device function of the tile coordinates (and of the LDS ring the kernel owns) so that TWO problems of the same
// tile shape can share one launch (gemm_bf16_mfma_dual_kernel below).
template <int TBM, int TBN, bool SPLITK, int NS, bool FUSE>
__device__ __forceinline__ void gemm_mfma_body(char* smem, const int tm, const int tn, const bf16* __restrict__ A, int64_t lda,
                                               const bf16* __restrict__ W, int64_t ldw, const EpiArgs& ep, int K, int tiles_n,
                                               float* __restrict__ partial, const Seg2& s2, int slices1,
                                               int* __restrict__ counters, const bool wt) {
    const int M = ep.M, N = ep.N;
    constexpr int A_BYTES = TBM * BK * 2, W_BYTES = TBN * BK * 2, STAGE = A_BYTES + W_BYTES;
    constexpr int FM = TBM / 32, FN = TBN / 32;  // 16x16 fragments per wave along m / n
    constexpr int IPT = TBM / 32 + TBN / 32;     // LDS-DMA instructions per wave and K-tile
    static_assert(NS >= 2 && NS <= 4, "ring depth");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = tm * TBM, n0 = tn * TBN;
    if (SPLITK) {  // this workgroup's K slice (K here is the slice length; the operands are advanced to the slice start).
        // Slices >= slices1 belong to the optional second operand pair (out = A.W^T + A2.W2^T: the two residual-stream
        // projections of a parallel-residual layer summed by the reduce kernel); a slice never straddles the two.
        const int sl = blockIdx.y;
        if (sl >= slices1) {
            A = s2.x + (int64_t)(sl - slices1) * K;
            W = s2.w + (int64_t)(sl - slices1) * K;
            lda = s2.ldx;
            ldw = s2.ldw;
        } else {
            A += (int64_t)sl * K;
            W += (int64_t)sl * K;
        }
    }

    f32x4_t acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // K-tiles: the first operand pair's, then (not SPLITK: one launch over both K ranges) the optional second pair's --
    // out = A.W^T + A2.W2^T, the two residual-stream projections of a parallel-residual layer as ONE GEMM over K = K1 + K2
    const int nk1 = K / BK;
    const int nk = nk1 + ((!SPLITK && s2.K > 0) ? s2.K / BK : 0);
    auto stage_pair = [&](int ti, char* dst) {
        if (!SPLITK && ti >= nk1) {
            stage_tile<TBM>(s2.x, s2.ldx, m0, M, (ti - nk1) * BK, dst, wave, lane);
            stage_tile<TBN>(s2.w, s2.ldw, n0, N, (ti - nk1) * BK, dst + A_BYTES, wave, lane);
        } else {
            stage_tile<TBM>(A, lda, m0, M, ti * BK, dst, wave, lane);
            stage_tile<TBN>(W, ldw, n0, N, ti * BK, dst + A_BYTES, wave, lane);
        }
    };
    if constexpr (NS == 2) {
        stage_pair(0, smem);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else {
#pragma unroll
        for (int s0 = 0; s0 < NS - 1; ++s0)
            if (s0 < nk) stage_pair(s0, smem + s0 * STAGE);
    }

    for (int t = 0; t < nk; ++t) {
        char* a_cur;
        if constexpr (NS == 2) {
            const int cur = t & 1;
            a_cur = smem + cur * STAGE;
            if (t + 1 < nk) stage_pair(t + 1, smem + (cur ^ 1) * STAGE);
        } else {
            // tiles issued after tile t so far: min(nk - 1, t + NS - 2) - t  (each IPT DMA instructions of this wave)
            const int younger = min(nk - 1, t + NS - 2) - t;
            if (younger >= 2) wait_vmcnt_n<2 * IPT>();
            else if (younger == 1) wait_vmcnt_n<IPT>();
            else wait_vmcnt_n<0>();
            cbarrier();
            a_cur = smem + (t % NS) * STAGE;
            const int tn_ = t + NS - 1;
            if (tn_ < nk) stage_pair(tn_, smem + (tn_ % NS) * STAGE);
        }
        char* w_cur = a_cur + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t wf[FN], af[FM];
#pragma unroll
            for (int i = 0; i < FN; ++i) wf[i] = read_frag(w_cur, wn * (TBN / 2) + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < FM; ++j) af[j] = read_frag(a_cur, wm * (TBM / 2) + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
        }
        if constexpr (NS == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    if constexpr (NS > 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing is in flight any more (tail waits reached 0); fence for the LDS reuse
        __syncthreads();
    }

    if (SPLITK && !(FUSE && gridDim.y == 1)) {  // fp32 partials: lane owns 4 consecutive columns of one row -> one 16-byte store per fragment
        float* slab = partial + (int64_t)blockIdx.y * M * N;
        // (descriptor-addressed: slab origin of this slice in SGPRs, 32-bit lane offsets -- a few MB of slab per slice on every split-K path)
        const __amdgpu_buffer_rsrc_t srs = tile_rsrc(reinterpret_cast<const char*>(slab));
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            const int m = m0 + wm * (TBM / 2) + j * 16 + (lane & 15);
            if (m >= M) continue;
#pragma unroll
            for (int i = 0; i < FN; ++i) {
                const int n = n0 + wn * (TBN / 2) + i * 16 + (lane >> 4) * 4;
                if (n >= N) continue;
                const f32x4_t v = acc[i][j];
                const u32x4_t u = u32x4_t{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                const uint32_t off = ((uint32_t)m * (uint32_t)N + (uint32_t)n) * 4u;
                if (wt) __builtin_amdgcn_raw_buffer_store_b128(u, srs, off, 0, 16 /* sc1 */);
                else __builtin_amdgcn_raw_buffer_store_b128(u, srs, off, 0, 0);
            }
        }
        if (!FUSE) return;
        // ---- arrive at the tile's counter; the last slice to arrive reduces ----
        const int S = gridDim.y;
        int* cnt = counters + (tm * tiles_n + tn);
        int* flag = reinterpret_cast<int*>(smem);          // the ring is idle: every DMA landed, every ds_read is done
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's slab stores have left the CU
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // restated after buffer_wbl2 where the compiler cannot drop it
            const int ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = (ticket == S - 1) ? 1 : 0;
        }
        __syncthreads();
        const bool last = *flag != 0;
        __syncthreads();                                    // everyone has read the flag before smem is reused below
        if (!last) return;
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        }
        __syncthreads();
        // sum the slabs in slice order (own contribution from its slab too: identical values, fixed order -> deterministic)
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            const int m = m0 + wm * (TBM / 2) + j * 16 + (lane & 15);
#pragma unroll
            for (int i = 0; i < FN; ++i) {
                const int n = n0 + wn * (TBN / 2) + i * 16 + (lane >> 4) * 4;
                f32x4_t sum = f32x4_t{0.f, 0.f, 0.f, 0.f};
                if (m < M && n < N) {
                    for (int sl = 0; sl < S; ++sl) {
                        const f32x4_t p = *reinterpret_cast<const f32x4_t*>(partial + ((int64_t)sl * M + m) * N + n);
                        sum[0] += p[0]; sum[1] += p[1]; sum[2] += p[2]; sum[3] += p[3];
                    }
                }
                acc[i][j] = sum;
            }
        }
    }
    // epilogue: D[i = n][j = m]; lane holds column m = lane & 15 and rows n = (lane >> 4) * 4 + r
    if ((N & 7) == 0) {
        // the last K-step ended with vmcnt(0) + barrier: every DMA has landed and every ds_read is done -> LDS is free
        static_assert(TBM * CTile<TBN>::PITCH <= NS * STAGE, "output tile must fit in the staging LDS");
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            const int ml = wm * (TBM / 2) + j * 16 + (lane & 15);
#pragma unroll
            for (int i = 0; i < FN; i += 2) {
                const int nl = wn * (TBN / 2) + i * 16 + (lane >> 4) * 4;
                epi_to_lds<TBN>(ep, smem, ml, nl, m0 + ml, n0 + nl, acc[i][j], acc[i + 1][j]);
            }
        }
        __syncthreads();
        store_tile_from_lds<TBM, TBN, 256>(ep, smem, m0, n0, tid);
        return;
    }
    const bool vec_ok = ((N & 3) == 0);
#pragma unroll
    for (int j = 0; j < FM; ++j) {
        const int m = m0 + wm * (TBM / 2) + j * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < FN; i += 2) {
            const int n = n0 + wn * (TBN / 2) + i * 16 + (lane >> 4) * 4;
            epilogue_pair(ep, m, n, acc[i][j], acc[i + 1][j], vec_ok);
        }
    }
}

template <int TBM, int TBN, bool SPLITK = false, int NS = 2, bool FUSE = false>
__global__ __launch_bounds__(256) void gemm_bf16_mfma_kernel(const bf16* __restrict__ A, int64_t lda,
                                                            const bf16* __restrict__ W, int64_t ldw, const EpiArgs ep,
                                                            int K, int tiles_m, int tiles_n,
                                                            float* __restrict__ partial = nullptr, const Seg2 s2 = Seg2{},
                                                            int slices1 = 1 << 30, int* __restrict__ counters = nullptr) {
    __shared__ __attribute__((aligned(16))) char smem[NS * (TBM + TBN) * BK * 2];
    int tm, tn;
    tile_coords(tiles_m, tiles_n, tm, tn);
    gemm_mfma_body<TBM, TBN, SPLITK, NS, FUSE>(smem, tm, tn, A, lda, W, ldw, ep, K, tiles_n, partial, s2, slices1, counters,
                                               ep.slab_wt != 0);
}

// Two problems with the same M, K and K slicing in ONE launch (round 4; the mid-batch decoder layer: the QKV projection reads
// LN1(x), the MLP up-projection LN2(x) of the same x, and with the parallel residual neither depends on the attention): tile
// columns [0, tiles_n1) belong to problem 1, the rest to problem 2; each keeps its own operands, epilogue arguments and slab
// region.  One launch ramps up and drains once and gives the K-slice choice 224 instead of 96 / 128 tiles to fill the chip with.
struct GemmSide {
    const bf16* A;
    const bf16* W;
    int64_t lda, ldw;
    float* partial;
    int* counters;
};
struct DualArgs {
    GemmSide g[2];
    EpiArgs ep[2];
    int wt;          // slab stores write-through (EpiArgs::slab_wt; a scalar here: the body reads it before it indexes the arrays)
};
// (out of line: with the body inlined into a kernel that picks its operands at run time this compiler's simplifycfg pass crashes;
//  one call per workgroup costs nothing against a ~10 us tile)
__device__ __attribute__((noinline)) void gemm_mfma_body_64x64_split4(char* smem, int tm, int tn, const bf16* A, int64_t lda,
                                                                     const bf16* W, int64_t ldw, const EpiArgs& ep, int K,
                                                                     int tiles_n, float* partial, int* counters, bool wt) {
    gemm_mfma_body<64, 64, true, 4, false>(smem, tm, tn, A, lda, W, ldw, ep, K, tiles_n, partial, Seg2{}, 1 << 30, counters, wt);
}
__global__ __launch_bounds__(256) void gemm_bf16_mfma_dual_kernel(const DualArgs a, int K, int tiles_m, int tiles_n1, int tiles_n2) {
    __shared__ __attribute__((aligned(16))) char smem[4 * (64 + 64) * BK * 2];
    int tm, tn;
    tile_coords(tiles_m, tiles_n1 + tiles_n2, tm, tn);
    // ONE inlined body; the side is picked by a workgroup-uniform INDEX into the kernel-argument arrays (scalar loads at a computed
    // offset).  Conditional struct copies or two inlined bodies crash this compiler's simplifycfg.
    const int side = tn >= tiles_n1 ? 1 : 0;
    const GemmSide& g = a.g[side];
    gemm_mfma_body_64x64_split4(smem, tm, tn - side * tiles_n1, g.A, g.lda, g.W, g.ldw, a.ep[side], K, side ? tiles_n2 : tiles_n1,
                                g.partial, g.counters, a.wt != 0);
}

// sum the K-slice partials and apply the shared epilogue (bias, GELU, residual): 8 columns per thread, 16-byte accesses
__device__ __forceinline__ void splitk_reduce_body(const int64_t idx, const float* __restrict__ partial, int slices, const EpiArgs& ep) {
    const int M = ep.M, N = ep.N;
    const int cpr = N / 8;
    if (idx >= (int64_t)M * cpr) return;
    const int m = (int)(idx / cpr), n = (int)(idx - (int64_t)m * cpr) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < slices; ++s) {
        const float* p = partial + ((int64_t)s * M + m) * N + n;
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
        v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
    if (ep.bias) {
        float bv[8];
        load_vec<bf16>(ep.bias + n, bv);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += bv[i];
    }
    if (ep.act == BT_ACT_GELU) {
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            const v2f g2 = gelu_tail2(v2f{v[i], v[i + 1]});
            v[i] = g2.x;
            v[i + 1] = g2.y;
        }
    }
    if (ep.res) {
        float rv[8];
        load_vec<bf16>(ep.res + (int64_t)m * ep.ldr + n, rv);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += rv[i];
    }
    store_vec<bf16>(ep.out + (int64_t)m * ep.ldo + n, v);
}
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int slices, const EpiArgs ep) {
    splitk_reduce_body((int64_t)blockIdx.x * 256 + threadIdx.x, partial, slices, ep);
}

// ---- fused reduce kernels of the mid-batch path (3 .. 512 rows, bt_decoder_forward) ----------------------------------------
// The split-K GEMM above leaves fp32 slabs [slices][M][N]; these kernels sum them and do everything the layer needs next at
// that point, so that no separate RoPE / KV-append / LayerNorm launch remains (a 17..256-row decode step is bound by the
// number of dependent launches, ~5-10 us each, not by bytes or flops).
// (slice order, whatever the count: the in-launch reductions and every reduce kernel produce the same bits.  Four slices' loads are
//  issued before the first add -- with 16 slices of the 256 x 128 slab kernel a one-load-at-a-time loop is sixteen dependent round trips)
__device__ inline void sum_slabs8(const float* __restrict__ partial, int slices, int64_t slab, int64_t off, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
    int s = 0;
    for (; s + 4 <= slices; s += 4) {
        float4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* p = partial + (int64_t)(s + u) * slab + off;
            a[u] = *reinterpret_cast<const float4*>(p);
            b[u] = *reinterpret_cast<const float4*>(p + 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[0] += a[u].x; v[1] += a[u].y; v[2] += a[u].z; v[3] += a[u].w;
            v[4] += b[u].x; v[5] += b[u].y; v[6] += b[u].z; v[7] += b[u].w;
        }
    }
    for (; s < slices; ++s) {
        const float* p = partial + (int64_t)s * slab + off;
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
        v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
}

// QKV: + bias, partial RoPE (rot == 32: pairs (e, e + 16) of the first 32 dims of q and k), q -> qkv buffer, (k_rot, v) -> cache
// rows pos0 + t.  One thread = one row x 8 columns (one 16-byte output); a thread in a rotated zone also sums its partner's 8.
__device__ __forceinline__ void mid_reduce_qkv_body(const int64_t idx, const float* __restrict__ partial, int slices, const EpiArgs& ep) {
    const int M = ep.M, N = ep.N;
    const int cpr = N / 8;
    if (idx >= (int64_t)M * cpr) return;
    const int m = (int)(idx / cpr), n = (int)(idx - (int64_t)m * cpr) * 8;
    const int64_t slab = (int64_t)M * N;
    float v[8];
    sum_slabs8(partial, slices, slab, (int64_t)m * N + n, v);
    if (ep.bias) {
        float bv[8];
        load_vec<bf16>(ep.bias + n, bv);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += bv[i];
    }
    const int D = ep.q.head_dim;
    const int head = n / (3 * D);
    const int within = n - head * 3 * D;
    const int part = within / D;              // 0 q, 1 k, 2 v
    const int el = within - part * D;         // multiple of 8
    const int64_t b = m / ep.q.q_len;
    const int t = m - (int)b * ep.q.q_len;
    const int pos0 = ep.q.pos0_dev ? *ep.q.pos0_dev : ep.q.pos0_host;
    if (part < 2 && el < 32) {
        const bool lo = el < 16;
        const int np = lo ? n + 16 : n - 16;
        float w[8];
        sum_slabs8(partial, slices, slab, (int64_t)m * N + np, w);
        if (ep.bias) {
            float bv[8];
            load_vec<bf16>(ep.bias + np, bv);
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] += bv[i];
        }
        const int rpos = ep.q.pos_table ? ep.q.pos_table[pos0 + t] : pos0 + t;
        const int e0 = lo ? el : el - 16;
        const float* ct = ep.q.cos_tab + (int64_t)rpos * 16 + e0;
        const float* st = ep.q.sin_tab + (int64_t)rpos * 16 + e0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float c = ct[i], sn = st[i];
            // lo: own = x1, partner = x2 -> x1 c - x2 s ; hi: own = x2, partner = x1 -> x2 c + x1 s
            v[i] = lo ? v[i] * c - w[i] * sn : v[i] * c + w[i] * sn;
        }
    }
    bf16* dst = part == 0 ? ep.out + (int64_t)m * ep.ldo + n
                          : (part == 1 ? ep.q.kc : ep.q.vc) + ((b * ep.q.heads + head) * ep.q.n_max + pos0 + t) * D + el;
    store_vec<bf16>(dst, v);
}
__global__ __launch_bounds__(256) void mid_reduce_qkv_kernel(const float* __restrict__ partial, int slices, const EpiArgs ep) {
    mid_reduce_qkv_body((int64_t)blockIdx.x * 256 + threadIdx.x, partial, slices, ep);
}
// both reductions behind the dual GEMM in one launch: workgroups [0, nb1) finish the QKV slabs, the rest the h->4h slabs
__global__ __launch_bounds__(256) void mid_reduce_qkv_gelu_kernel(const float* __restrict__ p1, const EpiArgs ep1, int nb1,
                                                                 const float* __restrict__ p2, const EpiArgs ep2, int slices) {
    if ((int)blockIdx.x < nb1) mid_reduce_qkv_body((int64_t)blockIdx.x * 256 + threadIdx.x, p1, slices, ep1);
    else splitk_reduce_body((int64_t)(blockIdx.x - nb1) * 256 + threadIdx.x, p2, slices, ep2);
}

// The same launch with the token decoder's local-window attention folded in (round 4): a WAVE per (sequence, head) sums the q / k / v
// slab columns of its head (lanes 0-15 q, 16-31 k, 32-47 v: one 8-column piece each), adds the bias, rotates (the rotate-half partner
// piece sits two lanes away), appends (k, v) to the cache rows pos0 + t -- value for value what mid_reduce_qkv_body writes -- and
// then attends: lanes 0-15 hold the query row exactly as local_attend_kernel lays it out (16 lanes x 8 dims), the cached rows
// (< pos0, written by earlier launches) are loaded 16 B per lane, the rows of this call come over from lanes 16-47 by shuffle, the
// score is a 16-lane DPP butterfly and the <= KMAX-key softmax runs sequentially in registers.  The q rows are not written at all
// and no attention launch follows: a 17..64-row token-decoder layer is 4 launches.  (Block decoder: global cache -> stays separate.)
template <int KMAX>
__device__ __forceinline__ void mid_reduce_qkv_attn_body(const int item, const float* __restrict__ partial, int slices, const EpiArgs& ep,
                                                         bf16* __restrict__ att, int64_t ld_att, float scale) {
    constexpr int D = 128;
    const int lane = threadIdx.x & 63;
    const int heads = ep.q.heads, q_len = ep.q.q_len, N = ep.N;
    const int n_seq = ep.M / q_len;
    if (item >= n_seq * heads) return;
    const int b = item / heads, head = item - b * heads;
    const int part = lane >> 4, c = lane & 15;                  // part 3 (lanes 48-63) only takes part in the shuffles
    const int pos0 = ep.q.pos0_host;
    const int64_t slab = (int64_t)ep.M * N;
    const int n = head * 3 * D + (part < 3 ? part : 0) * D + c * 8;
    const bf16* kbase = ep.q.kc + ((int64_t)b * heads + head) * ep.q.n_max * D + c * 8;
    const bf16* vbase = ep.q.vc + ((int64_t)b * heads + head) * ep.q.n_max * D + c * 8;
    // cached rows of earlier calls (lanes 0-15 use them; every lane loads: clamped, L2-resident)
    uint4 kq[KMAX - 1], vq[KMAX - 1];
#pragma unroll
    for (int r = 0; r < KMAX - 1; ++r) {
        const int rr = r < pos0 ? r : (pos0 > 0 ? pos0 - 1 : 0);
        kq[r] = *reinterpret_cast<const uint4*>(kbase + (int64_t)rr * D);
        vq[r] = *reinterpret_cast<const uint4*>(vbase + (int64_t)rr * D);
        // rows this call has not reached yet get weight exactly 0 below -- but 0 x (never-written memory, possibly NaN / Inf) is not
        // 0: they must not enter the sums as data (at pos0 = 0 the clamped row is the one this very wave is about to write)
        if (r >= pos0) kq[r] = vq[r] = uint4{0, 0, 0, 0};
    }
    float knew[2][8], vnew[2][8];                               // this call's rows, as seen by lanes 0-15 (bf16-rounded)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (t >= q_len) break;
        const int m = b * q_len + t;
        float v[8];
        sum_slabs8(partial, slices, slab, (int64_t)m * N + n, v);
        if (ep.bias) {
            float bv[8];
            load_vec<bf16>(ep.bias + n, bv);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += bv[i];
        }
        // partial RoPE (rot == 32): pieces 0, 1 (dims 0-15) pair with pieces 2, 3 (dims 16-31) of q and k
        float w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = __shfl_xor(v[i], 2, 64);
        if (part < 2 && c < 4) {
            const bool lo = c < 2;
            const int rpos = ep.q.pos_table ? ep.q.pos_table[pos0 + t] : pos0 + t;
            const int e0 = (lo ? c : c - 2) * 8;
            const float* ct = ep.q.cos_tab + (int64_t)rpos * 16 + e0;
            const float* st = ep.q.sin_tab + (int64_t)rpos * 16 + e0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float cs = ct[i], sn = st[i];
                v[i] = lo ? v[i] * cs - w[i] * sn : v[i] * cs + w[i] * sn;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = bf16_to_f32(f32_to_bf16(v[i]));   // what the unfused path stores and reads back
        if (part == 1 || part == 2)
            store_vec<bf16>((part == 1 ? ep.q.kc : ep.q.vc) + (((int64_t)b * heads + head) * ep.q.n_max + pos0 + t) * D + c * 8, v);
        // hand k / v of this row to lanes 0-15
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            knew[t][i] = __shfl(v[i], c + 16, 64);
            vnew[t][i] = __shfl(v[i], c + 32, 64);
        }
        // ---- attention of row t over pos0 + t + 1 keys (meaningful in lanes 0-15; the others compute along harmlessly) ----
        float q[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) q[i] = v[i] * scale;
        float sc[KMAX + 1];
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < KMAX - 1; ++r) {
            const uint32_t kw[4] = {kq[r].x, kq[r].y, kq[r].z, kq[r].w};
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                d = fmaf(q[2 * i], __uint_as_float(kw[i] << 16), d);
                d = fmaf(q[2 * i + 1], __uint_as_float(kw[i] & 0xffff0000u), d);
            }
            d = row16_sum_dpp(d);
            sc[r] = r < pos0 ? d : -INFINITY;
            mx = fmaxf(mx, sc[r]);
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) d = fmaf(q[i], knew[tt <= t ? tt : t][i], d);
            d = row16_sum_dpp(d);
            sc[KMAX - 1 + tt] = tt <= t ? d : -INFINITY;
            mx = fmaxf(mx, sc[KMAX - 1 + tt]);
        }
        float l = 0.f, acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
        for (int r = 0; r < KMAX - 1; ++r) {
            const float p = __expf(sc[r] - mx);                  // masked rows: exp(-inf) = 0
            l += p;
            const uint32_t vw[4] = {vq[r].x, vq[r].y, vq[r].z, vq[r].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[2 * i] = fmaf(p, __uint_as_float(vw[i] << 16), acc[2 * i]);
                acc[2 * i + 1] = fmaf(p, __uint_as_float(vw[i] & 0xffff0000u), acc[2 * i + 1]);
            }
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const float p = __expf(sc[KMAX - 1 + tt] - mx);
            l += p;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fmaf(p, vnew[tt <= t ? tt : t][i], acc[i]);
        }
        if (part == 0) {
            const float inv = 1.0f / l;                          // the row's own key is always visible: l >= 1
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] *= inv;
            store_vec<bf16>(att + (int64_t)m * ld_att + head * D + c * 8, acc);
        }
    }
}
template <int KMAX>
__global__ __launch_bounds__(256) void mid_reduce_qkv_attn_gelu_kernel(const float* __restrict__ p1, const EpiArgs ep1, int nb1,
                                                                      const float* __restrict__ p2, const EpiArgs ep2, int slices,
                                                                      bf16* __restrict__ att, int64_t ld_att, float scale) {
    // (alternating the two kinds of workgroup instead -- dependent round trips next to a stream -- measured 16.7 -> 24.9 us per launch at
    //  256 rows, -1.5 ... -2.4 % end to end at B = 64 ... 256: profiles/r5_ab_mid_reduce_interleave_rejected.jsonl)
    const int b = (int)blockIdx.x;
    const int kind = b < nb1 ? 0 : 1, idx = b < nb1 ? b : b - nb1;
    if (kind == 0) mid_reduce_qkv_attn_body<KMAX>(idx * 4 + (int)(threadIdx.x >> 6), p1, slices, ep1, att, ld_att, scale);
    else splitk_reduce_body((int64_t)idx * 256 + threadIdx.x, p2, slices, ep2);
}

// Residual stream: x[m, :] += sum(slabs) + bias + bias2 (in place, rounded to bf16 like the unfused epilogues), then the NEXT
// layer's two LayerNorms of the new row (ln1 / ln2; skipped when ln.w == nullptr).  One workgroup per row, 8 columns per
// thread and 2048-column chunk, the row stays in registers (H <= 8192).
struct Ln2Out {
    const bf16 *w1, *b1, *w2, *b2;
    bf16 *o1, *o2;
    int64_t ldo1, ldo2;
    float eps;
};

__global__ __launch_bounds__(256) void mid_reduce_res_ln_kernel(const float* __restrict__ partial, int slices, const EpiArgs ep,
                                                               const bf16* __restrict__ bias2, const Ln2Out ln) {
    __shared__ float red[2][4];
    const int N = ep.N, m = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t slab = (int64_t)ep.M * N;
    constexpr int MAXC = 4;
    float x[MAXC][8];
    float sum = 0.f;
#pragma unroll
    for (int ch = 0; ch < MAXC; ++ch) {
        const int n = ch * 2048 + tid * 8;
        if (n < N) {
            float v[8];
            sum_slabs8(partial, slices, slab, (int64_t)m * N + n, v);
            float bv[8];
            if (ep.bias) {
                load_vec<bf16>(ep.bias + n, bv);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += bv[i];
            }
            if (bias2) {
                load_vec<bf16>(bias2 + n, bv);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += bv[i];
            }
            if (ep.res) {
                load_vec<bf16>(ep.res + (int64_t)m * ep.ldr + n, bv);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += bv[i];
            }
            store_vec<bf16>(ep.out + (int64_t)m * ep.ldo + n, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                x[ch][i] = bf16_to_f32(f32_to_bf16(v[i]));   // LayerNorm sees the stored (bf16) row, as bt_layernorm would
                sum += x[ch][i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[ch][i] = 0.f;
        }
    }
    if (!ln.w1) return;
    sum = wave_sum(sum);
    if (lane == 0) red[0][wave] = sum;
    __syncthreads();
    const float mean = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)N;
    float sq = 0.f;
#pragma unroll
    for (int ch = 0; ch < MAXC; ++ch) {
        if (ch * 2048 + tid * 8 < N) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float d = x[ch][i] - mean;
                sq = fmaf(d, d, sq);
            }
        }
    }
    sq = wave_sum(sq);
    if (lane == 0) red[1][wave] = sq;
    __syncthreads();
    const float rstd = rsqrtf((red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)N + ln.eps);
#pragma unroll
    for (int ch = 0; ch < MAXC; ++ch) {
        const int n = ch * 2048 + tid * 8;
        if (n < N) {
            float g[8], bb[8], r[8];
            load_vec<bf16>(ln.w1 + n, g);
            load_vec<bf16>(ln.b1 + n, bb);
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = fmaf((x[ch][i] - mean) * rstd, g[i], bb[i]);
            store_vec<bf16>(ln.o1 + (int64_t)m * ln.ldo1 + n, r);
            if (ln.o2) {
                load_vec<bf16>(ln.w2 + n, g);
                load_vec<bf16>(ln.b2 + n, bb);
#pragma unroll
                for (int i = 0; i < 8; ++i) r[i] = fmaf((x[ch][i] - mean) * rstd, g[i], bb[i]);
                store_vec<bf16>(ln.o2 + (int64_t)m * ln.ldo2 + n, r);
            }
        }
    }
}

// ---- 256x256 deep-pipelined kernel ---------------------------------------------------------------------------------------
// For grids that fill the chip with one 8-wave workgroup per CU.  Arithmetic intensity per workgroup doubles vs the 128^2
// tile (128 flop per L2->LDS byte), which takes the kernel off the L2-bandwidth roof the small tile sits under.
//
// K-loop = phases of 16 MFMAs per wave (one 64x32 quadrant of the wave's 128x64 output, K = 64).  LDS holds two K-tile
// buffers of four 16 KiB HALF-tiles [A rows 0-127 | A rows 128-255 | W rows 0-127 | W rows 128-255]; a wave's 128 output
// rows are 64 rows of each A half and its 64 output columns 32 rows of each W half, so a half-tile is last read in a
// known phase and can be refilled by LDS-DMA long before the buffer as a whole is free:
//     phase p of K-tile t:  p0 reads A0,W0 -> MFMA(A0,W0)   p1 reads W1 -> MFMA(A0,W1)
//                           p2 reads A1    -> MFMA(A1,W1)   p3 (no reads) -> MFMA(A1,W0)        (both W halves stay in VGPRs)
// Half-tile loads are issued one per phase, PD = 6 phases ahead of the phase that first needs them, in consumption order
// A0,W0,W1,A1.  Each phase is  [s_waitcnt vmcnt(counted) ; s_barrier ; issue 1 half-tile DMA ; ds_reads ; 16 MFMA]:
//   RAW: the data a phase reads was waited for (counted vmcnt, never 0 in steady state: 4 half-tiles = 8 DMA instructions
//        stay in flight) by EVERY wave before the barrier that opens the phase;
//   WAR: a half-tile slot is refilled >= 2 barriers after its last ds_read (reads complete before the MFMAs that consume
//        them, i.e. before the reading wave reaches the next barrier).
constexpr int BM2 = 256, BN2 = 256;
constexpr int HALF_BYTES = 128 * BK * 2;   // 16 KiB
constexpr int PD = 6;                      // prefetch distance in phases
constexpr int GROUP_M2 = 4;

template <int N>
__device__ inline void wait_vmcnt() {
    if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// half-tile load s (global sequence number): K-tile t = s / 4, kind = s % 4 in consumption order A0, W0, W1, A1
__device__ inline void stage_half(const bf16* __restrict__ A, int64_t lda, const bf16* __restrict__ W, int64_t ldw,
                                  int m0, int n0, int M, int N, int s, char* smem, int wave, int lane) {
    const int t = s >> 2, kind = s & 3;
    const bool is_a = (kind == 0) || (kind == 3);
    const int half = (kind == 2 || kind == 3) ? 1 : 0;
    const bf16* src = is_a ? A : W;
    const int64_t ld = is_a ? lda : ldw;
    const int row0 = (is_a ? m0 : n0) + half * 128;
    const int row_max = is_a ? M : N;
    char* slot = smem + (t & 1) * 4 * HALF_BYTES + ((is_a ? 0 : 2) + half) * HALF_BYTES;
    const int k0 = t * BK;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rbase = wave * 16 + i * 8;
        const int row = rbase + (lane >> 3);
        const int gslot = (lane & 7) ^ (row & 7);
        int grow = row0 + row;
        grow = grow < row_max ? grow : row_max - 1;
        const bf16* g = src + (int64_t)grow * ld + k0 + gslot * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(slot + rbase * 128), 16, 0, 0);
    }
}

// wait until every half-tile load that phase g reads has landed (this wave's DMA share), leaving the younger ones in flight
__device__ inline void wait_for_phase(int g, int total) {
    if (g >= total) return;  // past the last phase: nothing to wait for (total = 4 * nk phases == loads)
    const int p = g & 3;
    const int need = (g & ~3) + (p == 0 ? 1 : (p == 1 ? 2 : 3));
    const int issued = min(g - 1 + PD, total - 1);
    const int allow = issued - need;  // half-tile loads that may stay in flight (2 DMA instructions each)
    if (allow >= 4) wait_vmcnt<8>();
    else if (allow == 3) wait_vmcnt<6>();
    else if (allow == 2) wait_vmcnt<4>();
    else if (allow == 1) wait_vmcnt<2>();
    else wait_vmcnt<0>();
}

#ifdef BT_DEBUG_HOOKS   // four-phase ring + timing ablations: A/B probes only, not in the product library
// STAGGER: waves 4-7 run half a phase behind waves 0-3 (two barriers per phase): while one wave of a SIMD is in its
// 16-MFMA cluster (s_setprio 1) its partner issues the DMA and ds_reads of its own phase, so the matrix pipe does not
// idle during LDS latency.  Group 1 therefore waits for phase g+1's data before the barrier that opens group 0's phase
// g+1 (every wave must have waited for its DMA share before ANY wave reads).
// ABL (timing experiments only, results are wrong for 1 and 2): 1 = no DMA in the loop, 2 = no ds_reads in the loop,
// 4 = ds_reads before the DMA issue, 8 = DMA issued in the middle of the MFMA cluster, 16 = no MFMAs
template <bool STAGGER, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_bf16_mfma256_kernel(const bf16* __restrict__ A, int64_t lda,
                                                               const bf16* __restrict__ W, int64_t ldw, const EpiArgs ep,
                                                               int K, int tiles_m, int tiles_n) {
    const int M = ep.M, N = ep.N;
    // 128 KiB ring; the epilogue reuses it as a padded 256 x 256 bf16 output tile (132 KiB)
    __shared__ __attribute__((aligned(16))) char smem[BM2 * CTile<BN2>::PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const bool grp1 = STAGGER && (__builtin_amdgcn_readfirstlane(tid >> 8) != 0);
    // tile coordinates: bijective XCD remap, then GROUP_M2 grouping
    int tm, tn;
    {
        const int nwg = tiles_m * tiles_n;
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        const int group_sz = GROUP_M2 * tiles_n;
        const int g = id / group_sz;
        const int first_m = g * GROUP_M2;
        const int gm = min(tiles_m - first_m, GROUP_M2);
        const int in_g = id - g * group_sz;
        tm = first_m + in_g % gm;
        tn = in_g / gm;
    }
    const int m0 = tm * BM2, n0 = tn * BN2;

    f32x4_t acc[8][4];  // [A half * 4 + m-frag][W half * 2 + n-frag]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    const int total = 4 * nk;  // half-tile loads == phases
#pragma unroll
    for (int s = 0; s < PD; ++s)
        if (s < total) stage_half(A, lda, W, ldw, m0, n0, M, N, s, smem, wave, lane);

    bf16x8_t af[4][2];      // current A half: [m-frag][kk]
    bf16x8_t wf[2][2][2];   // both W halves:  [half][n-frag][kk]

    if (grp1) {
        wait_for_phase(0, total);
        cbarrier();
    }
    for (int t = 0; t < nk; ++t) {
        const char* buf = smem + (t & 1) * 4 * HALF_BYTES;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int g = 4 * t + p;
            if (!grp1) wait_for_phase(g, total);
            cbarrier();
            if (!(ABL & (1 | 4 | 8)) && g + PD < total) stage_half(A, lda, W, ldw, m0, n0, M, N, g + PD, smem, wave, lane);
            if (ABL & 2) {
                if (t == 0 && p == 0) {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            wf[0][j][kk] = read_frag(buf + 2 * HALF_BYTES, wn * 32 + j * 16, kk, lane);
                            wf[1][j][kk] = read_frag(buf + 3 * HALF_BYTES, wn * 32 + j * 16, kk, lane);
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) af[i][kk] = read_frag(buf, wm * 64 + i * 16, kk, lane);
                    }
                }
            } else if (p == 0) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) wf[0][j][kk] = read_frag(buf + 2 * HALF_BYTES, wn * 32 + j * 16, kk, lane);
#pragma unroll
                    for (int i = 0; i < 4; ++i) af[i][kk] = read_frag(buf, wm * 64 + i * 16, kk, lane);
                }
            } else if (p == 1) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < 2; ++j) wf[1][j][kk] = read_frag(buf + 3 * HALF_BYTES, wn * 32 + j * 16, kk, lane);
            } else if (p == 2) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i) af[i][kk] = read_frag(buf + HALF_BYTES, wm * 64 + i * 16, kk, lane);
            }
            if ((ABL & 4) && g + PD < total) stage_half(A, lda, W, ldw, m0, n0, M, N, g + PD, smem, wave, lane);
            if (STAGGER) {
                if (grp1) wait_for_phase(g + 1, total);
                cbarrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            const int ah = (p >= 2) ? 1 : 0;
            const int wh = (p == 1 || p == 2) ? 1 : 0;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2 && !(ABL & 16); ++kk) {   // ABL & 16: no MFMAs (operand-delivery rate of the ring alone)
                if ((ABL & 8) && kk == 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (g + PD < total) stage_half(A, lda, W, ldw, m0, n0, M, N, g + PD, smem, wave, lane);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[ah * 4 + i][wh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            wf[wh][j][kk], af[i][kk], acc[ah * 4 + i][wh * 2 + j], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
        }
    }
    if (STAGGER && !grp1) cbarrier();

    if ((N & 7) == 0) {
        __syncthreads();  // all waves are past their last ds_reads; every DMA was waited for (vmcnt(0) in the tail phases)
#pragma unroll
        for (int ai = 0; ai < 8; ++ai) {
            const int ml = (ai >> 2) * 128 + wm * 64 + (ai & 3) * 16 + (lane & 15);
#pragma unroll
            for (int wh = 0; wh < 2; ++wh) {
                const int nl = wh * 128 + wn * 32 + (lane >> 4) * 4;
                epi_to_lds<BN2>(ep, smem, ml, nl, m0 + ml, n0 + nl, acc[ai][wh * 2], acc[ai][wh * 2 + 1]);
            }
        }
        __syncthreads();
        store_tile_from_lds<BM2, BN2, 512>(ep, smem, m0, n0, tid);
        return;
    }
#pragma unroll
    for (int ai = 0; ai < 8; ++ai) {
        const int m = m0 + (ai >> 2) * 128 + wm * 64 + (ai & 3) * 16 + (lane & 15);
#pragma unroll
        for (int wh = 0; wh < 2; ++wh) {
            const int n = n0 + wh * 128 + wn * 32 + (lane >> 4) * 4;
            epilogue_pair(ep, m, n, acc[ai][wh * 2], acc[ai][wh * 2 + 1], true);
        }
    }
}

#endif  // BT_DEBUG_HOOKS

// ---- 256x256, two 32-MFMA phases per K-tile ---------------------------------------------------------------------------------
// Same tile, LDS ring and fragment mapping as gemm_bf16_mfma256_kernel, but a phase is 32 MFMAs (half of the wave's output
// x K = 64) so a K-tile costs 4 barriers instead of 8:
//     phase 2t  : reads A0, W0, W1 -> MFMA(A0,W0), MFMA(A0,W1)        phase 2t+1: reads A1 -> MFMA(A1,W1), MFMA(A1,W0)
// Two half-tile loads are issued per phase, PD2 = 6 loads ahead (consumption order A0,W0,W1,A1).
//   RAW: before the barrier that opens phase g every wave has waited for the loads that phase reads;
//   WAR: load s+8 reuses the slot of load s and is issued >= 1 barrier after the last read of s (checked by simulation,
//        tests/test_gemm_ring_schedule.py).
constexpr int PD2 = 6;

__device__ inline void wait_for_phase2(int g, int total_loads) {
    // loads phase g needs: all of K-tile t's A0,W0,W1 (s <= 4t+2) for even g, plus A1 (s <= 4t+3) for odd g
    const int t = g >> 1;
    if (4 * t >= total_loads) return;
    const int need = 4 * t + ((g & 1) ? 3 : 2);
    const int issued = min(2 * (g - 1) + 1 + PD2, total_loads - 1);   // through phase g-1 (prologue: s < PD2)
    const int allow = issued - need;
    if (allow >= 4) wait_vmcnt<8>();
    else if (allow == 3) wait_vmcnt<6>();
    else if (allow == 2) wait_vmcnt<4>();
    else if (allow == 1) wait_vmcnt<2>();
    else wait_vmcnt<0>();
}

template <bool STAGGER, bool NOSTORE = false>
__global__ __launch_bounds__(512) void gemm_bf16_mfma256p2_kernel(const bf16* __restrict__ A, int64_t lda,
                                                                 const bf16* __restrict__ W, int64_t ldw, const EpiArgs ep,
                                                                 int K, int tiles_m, int tiles_n) {
    const int M = ep.M, N = ep.N;
    // 128 KiB ring; the epilogue reuses it as a padded 256 x 256 bf16 output tile (132 KiB)
    __shared__ __attribute__((aligned(16))) char smem[BM2 * CTile<BN2>::PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const bool grp1 = STAGGER && (__builtin_amdgcn_readfirstlane(tid >> 8) != 0);
    int tm, tn;
    {
        const int nwg = tiles_m * tiles_n;
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        const int group_sz = GROUP_M2 * tiles_n;
        const int g = id / group_sz;
        const int first_m = g * GROUP_M2;
        const int gm = min(tiles_m - first_m, GROUP_M2);
        const int in_g = id - g * group_sz;
        tm = first_m + in_g % gm;
        tn = in_g / gm;
    }
    const int m0 = tm * BM2, n0 = tn * BN2;

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    const int total = 4 * nk;      // half-tile loads
    const int phases = 2 * nk;
#pragma unroll
    for (int s = 0; s < PD2; ++s)
        if (s < total) stage_half(A, lda, W, ldw, m0, n0, M, N, s, smem, wave, lane);

    bf16x8_t af[4][2];
    bf16x8_t wf[2][2][2];

    if (grp1) {
        wait_for_phase2(0, total);
        cbarrier();
    }
    for (int t = 0; t < nk; ++t) {
        const char* buf = smem + (t & 1) * 4 * HALF_BYTES;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int g = 2 * t + p;
            if (!grp1) wait_for_phase2(g, total);
            cbarrier();
            if (p == 0) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        wf[0][j][kk] = read_frag(buf + 2 * HALF_BYTES, wn * 32 + j * 16, kk, lane);
                        wf[1][j][kk] = read_frag(buf + 3 * HALF_BYTES, wn * 32 + j * 16, kk, lane);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) af[i][kk] = read_frag(buf, wm * 64 + i * 16, kk, lane);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i) af[i][kk] = read_frag(buf + HALF_BYTES, wm * 64 + i * 16, kk, lane);
            }
            // two half-tile DMAs per phase, after the ds_reads were issued (their issue cost hides LDS latency)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int s = 2 * g + e + PD2;
                if (s < total) stage_half(A, lda, W, ldw, m0, n0, M, N, s, smem, wave, lane);
            }
            if (STAGGER) {
                if (grp1) wait_for_phase2(g + 1, total);
                cbarrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            const int ah = p;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int wh = (p == 0) ? h : 1 - h;   // A0: W0 then W1 ; A1: W1 then W0
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[ah * 4 + i][wh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                wf[wh][j][kk], af[i][kk], acc[ah * 4 + i][wh * 2 + j], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
        }
    }
    (void)phases;
    if (STAGGER && !grp1) cbarrier();
    if (NOSTORE) {  // timing ablation: keep the accumulators live, skip the epilogue
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }

    if ((N & 7) == 0) {
        __syncthreads();  // all waves are past their last ds_reads; every DMA was waited for (vmcnt(0) in the tail phases)
#pragma unroll
        for (int ai = 0; ai < 8; ++ai) {
            const int ml = (ai >> 2) * 128 + wm * 64 + (ai & 3) * 16 + (lane & 15);
#pragma unroll
            for (int wh = 0; wh < 2; ++wh) {
                const int nl = wh * 128 + wn * 32 + (lane >> 4) * 4;
                epi_to_lds<BN2>(ep, smem, ml, nl, m0 + ml, n0 + nl, acc[ai][wh * 2], acc[ai][wh * 2 + 1]);
            }
        }
        __syncthreads();
        store_tile_from_lds<BM2, BN2, 512>(ep, smem, m0, n0, tid);
        return;
    }
#pragma unroll
    for (int ai = 0; ai < 8; ++ai) {
        const int m = m0 + (ai >> 2) * 128 + wm * 64 + (ai & 3) * 16 + (lane & 15);
#pragma unroll
        for (int wh = 0; wh < 2; ++wh) {
            const int n = n0 + wh * 128 + wn * 32 + (lane >> 4) * 4;
            epilogue_pair(ep, m, n, acc[ai][wh * 2], acc[ai][wh * 2 + 1], true);
        }
    }
}

// ---- persistent 256x256 kernel (production path for chip-filling grids) ----------------------------------------------------
// The two-phase ring above, made persistent: min(tiles, #CU) workgroups each walk tiles id = round * G + remap(block), and
// the half-tile load stream simply continues across tile boundaries (the load cursor moves to the next tile's A / W panels
// two K-tiles before the MFMAs get there).  What this removes is the per-tile bubble of the one-tile-per-workgroup kernels:
// prologue latency + a 32 MiB chip-wide store burst that had to drain before the next workgroup could start (measured:
// 18 % of a K = 2048 GEMM).  Here the epilogue is per WAVE and needs no block barrier:
//   * a wave owns 128 rows x 64 CONTIGUOUS columns (wn selects the W half and the 64-row slab inside it), so its output
//     rows are full 128-byte lines;
//   * per 16-row fragment: registers -> a private 16 x 64 bf16 LDS patch (2.25 KiB per wave, next to the ring) -> two
//     16-byte-per-lane coalesced global stores per fragment = 16 store instructions per tile, fire and forget;
//   * the bias is folded into the accumulator init (DMA'd to a 1 KiB LDS row by wave 0 together with the tile's first
//     loads), so the epilogue issues no loads unless there is a residual / RoPE table to read.
// vmcnt accounting: loads and stores retire in order on gfx9-family vmcnt, so the stores sit in the same queue as the
// ring DMAs.  For the two phases after an epilogue the needed DMAs are OLDER than the 16 stores, and the counted waits add
// 16 (only when the previous tile was a full tile: every wave then issued >= 16 stores; on edge tiles the smaller count is
// used, which is merely conservative).  From the third phase on the needed DMAs are younger than the stores and the
// stores must have been acknowledged by L2 -- they get ~2 phases (~1 us) to do so, which holds once the CUs have drifted
// out of lockstep.  Any other VMEM op a wave issues (bias DMA, residual / RoPE loads) only makes its waits stricter.
constexpr int EPI_PITCH = CTile<64>::PITCH;                  // 144 B: 16 rows x (128 B + 16 pad)
constexpr int PP_EPI_OFF = 8 * HALF_BYTES;                   // after the 128 KiB ring
constexpr int PP_EPI_WAVE = 16 * EPI_PITCH;
constexpr int PP_BIAS_OFF = PP_EPI_OFF + 8 * 2560;   // 8 wave patches (16x16 MFMA: 2304 B, 32x32: 2560 B)
constexpr int PP_SMEM = PP_BIAS_OFF + 2 * 1024;
constexpr int PP_STORES = 16;                                // store instructions per wave per full tile
static_assert(PP_SMEM <= 160 * 1024, "persistent GEMM LDS budget");

template <int N>
__device__ inline void wait_vmcnt_pp() {
    if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if constexpr (N == 22) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
    else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else wait_vmcnt<N>();
}

// as wait_for_phase2, over the workgroup's whole load stream; stores_younger: a full tile's epilogue (PP_STORES store
// instructions) was issued after the youngest load this phase needs
__device__ inline void wait_for_phase_pp(int g, int total_loads, bool stores_younger) {
    const int t = g >> 1;
    if (4 * t >= total_loads) return;
    const int need = 4 * t + ((g & 1) ? 3 : 2);
    const int issued = min(2 * (g - 1) + 1 + PD2, total_loads - 1);
    const int allow = issued - need;
    if (stores_younger) {
        static_assert(PP_STORES == 16, "vmcnt immediates below assume 16 stores");
        if (allow >= 4) wait_vmcnt_pp<24>();
        else if (allow == 3) wait_vmcnt_pp<22>();
        else if (allow == 2) wait_vmcnt_pp<20>();
        else if (allow == 1) wait_vmcnt_pp<18>();
        else wait_vmcnt_pp<16>();
    } else {
        if (allow >= 4) wait_vmcnt<8>();
        else if (allow == 3) wait_vmcnt<6>();
        else if (allow == 2) wait_vmcnt<4>();
        else if (allow == 1) wait_vmcnt<2>();
        else wait_vmcnt<0>();
    }
}

// tile of (round, this workgroup): XCD remap inside the round (each XCD's L2 sees a contiguous id range), then GROUP_M2
__device__ inline bool pp_tile(int round, int tiles_m, int tiles_n, int& m0, int& n0) {
    const int G = gridDim.x;
    const int nwg = tiles_m * tiles_n;
    const int base = round * G;
    const int S = min(G, nwg - base);
    if (S <= 0) return false;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, local = bid >> 3, q = S >> 3, r = S & 7;
    if (local >= q + (xcd < r ? 1 : 0)) return false;
    const int id = base + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    const int group_sz = GROUP_M2 * tiles_n;
    const int g = id / group_sz;
    const int first_m = g * GROUP_M2;
    const int gm = min(tiles_m - first_m, GROUP_M2);
    const int in_g = id - g * group_sz;
    m0 = (first_m + in_g % gm) * BM2;
    n0 = (in_g / gm) * BN2;
    return true;
}

__device__ inline uint32_t lds_addr(const char* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

// 16-byte-per-lane LDS-DMA, global address = SGPR base + 32-bit VGPR offset, LDS address = lds_base (M0) + lane * 16
__device__ inline void glds16_sbase(const char* base_uniform, uint32_t voff, uint32_t lds_base_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(voff), "s"(base_uniform), "s"(lds_base_uniform)
                 : "memory");
}

// same with a per-lane 64-bit global address (edge tiles, bias row).  The persistent kernel issues every LDS-DMA through
// these two helpers so the compiler never emits its own M0 writes in that kernel (it does not know the asm changes M0).
__device__ inline void glds16_vaddr(const void* gptr, uint32_t lds_base_uniform) {
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gptr), "s"(lds_base_uniform) : "memory");
}

// one half-tile (kind in consumption order A0, W0, W1, A1) of the K-tile at k0 into ring buffer `parity`.
// Addressing is (wave-uniform 64-bit base) + (32-bit lane offset): the lane part -- row (lane >> 3) of an 8-row group and
// the swizzled 16-byte slot -- is the same for every half-tile of an operand, so the whole load stream costs two VGPRs
// (voff_a / voff_w) instead of a 64-bit pointer per DMA.  Tiles cut by the matrix edge clamp rows per lane (rare path).
__device__ inline void stage_half_at(const bf16* __restrict__ A, int64_t lda, const bf16* __restrict__ W, int64_t ldw,
                                     int m0, int n0, int M, int N, int k0, int parity, int kind, char* smem, int wave,
                                     int lane, uint32_t voff_a, uint32_t voff_w) {
    const bool is_a = (kind == 0) || (kind == 3);
    const int half = (kind == 2 || kind == 3) ? 1 : 0;
    const char* src = reinterpret_cast<const char*>(is_a ? A : W);
    const int64_t ld = is_a ? lda : ldw;
    const int row0 = (is_a ? m0 : n0) + half * 128 + wave * 16;
    const int row_max = is_a ? M : N;
    const uint32_t voff = is_a ? voff_a : voff_w;
    char* slot = smem + parity * 4 * HALF_BYTES + ((is_a ? 0 : 2) + half) * HALF_BYTES + wave * 16 * 128;
    if (row0 + 16 <= row_max) {
        const char* base = src + ((int64_t)row0 * ld + k0) * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16_sbase(base + i * 8 * ld * 2, voff, lds_addr(slot + i * 8 * 128));
    } else {
        int l = lane;
        asm volatile("" : "+v"(l));   // opaque: keep this path's lane-derived address parts out of registers until needed
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int grow = row0 + i * 8 + (l >> 3);
            grow = grow < row_max ? grow : row_max - 1;
            const int gslot = (l & 7) ^ ((l >> 3) & 7);
            glds16_vaddr(src + ((int64_t)grow * ld + k0 + gslot * 8) * 2, lds_addr(slot + i * 8 * 128));
        }
    }
}


// Per-wave epilogue of the persistent kernel: the wave's 128 rows x 64 columns (8 fragments of 16 rows), bias already in
// the accumulators.  Sweep A issues EVERY load the epilogue needs (RoPE table rows / residual chunks) before sweep B
// issues the first store: with in-order vmcnt retirement a load waited for after a store would wait for that store too.
enum { PP_PLAIN = 0, PP_GELU = 1, PP_RES = 2, PP_QKV = 3, PP_ARGMAX = 4 };   // epilogue flavour (bias is orthogonal: accumulator init)
// PP_ARGMAX (bt_logits_argmax, round 3): the logits GEMM of a greedy token step.  Every wave reduces its 128 x 64 block to one
// (value, column) pair per row -- over the bf16-ROUNDED values, lowest column on ties, exactly what bt_argmax sees in a
// materialised logits row -- and writes 8 bytes per row; the logits tile itself is only stored when e.out != nullptr.  A
// [4096, 50304] bf16 logits matrix is 412 MB written and read back per token step; the partials are 26 MB.



// (b, t) = (m / q_len, m % q_len) for the rows of one tile: q_len == 1 (decode) is the identity; for q_len >= 256 a tile's 256 rows
// cross at most one sequence boundary, so the tile's first row is divided once (scalar) and lanes add / wrap; otherwise divide.
// (The per-lane 32-bit division is ~30 VALU instructions; the QKV epilogue needs it for 24 addresses per lane and tile.)
struct RowSplit {
    int q_len, b0, t0, m0;
    bool wrap1;
    __device__ inline RowSplit(int q_len_, int cm0) : q_len(q_len_), m0(cm0) {
        wrap1 = q_len_ >= BM2;
        b0 = wrap1 ? __builtin_amdgcn_readfirstlane(cm0 / q_len_) : 0;
        t0 = wrap1 ? __builtin_amdgcn_readfirstlane(cm0 - b0 * q_len_) : 0;
    }
    __device__ inline void split(int m, int& b, int& t) const {
        if (q_len == 1) {
            b = m;
            t = 0;
        } else if (wrap1) {
            t = t0 + (m - m0);
            b = b0;
            if (t >= q_len) {
                t -= q_len;
                b += 1;
            }
        } else {
            b = m / q_len;
            t = m - b * q_len;
        }
    }
};

// EABL (debug timing ablations, wrong results): 1 = no global stores, 2 = no LDS transposes (the packed registers are stored as is);
// 16 (correct results, A/B): QKV mode with the generic per-lane K / V addresses
// EABL 256 (timeline variant): s_memtime stamps at LDS byte address tl + 8 * event (tl = 0: none) -- events 10 .. 12 inside the epilogue
template <int EABL>
__device__ inline void epi_stamp(uint32_t tl, int ev) {
    if constexpr ((EABL & 256) != 0) {
        if (tl != 0) {
            uint64_t tm;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm) : : "memory");
            const uint32_t a = tl + (uint32_t)ev * 8u;
            asm volatile("ds_write_b64 %0, %1" : : "v"(a), "v"(tm) : "memory");
        }
    }
}

template <int MODE, int EABL = 0>
__device__ inline void pp_epilogue(const EpiArgs& e, int pos0, char* stage, f32x4_t (&acc)[8][4], int cm0, int n_wave, int wm,
                                   int lane, uint32_t tl = 0) {
    const int M = e.M, N = e.N;
    constexpr bool qkv = MODE == PP_QKV;
    if constexpr ((EABL & 32) != 0) {
        // folded LayerNorm (EpiArgs.ln_*): acc <- rstd[m] * (acc - mu[m] * s[n]) + c[n].  The lane holds output rows m = 16-row fragment
        // ai, row (lane & 15), and for every 16-column fragment jf the 4 columns (lane >> 4) * 4 + r.  All 24 loads first (8 row
        // statistics, 2 x 4 column vectors), then 2 FMAs per value:  acc * a + (b * s + c)  with  a = rstd,  b = -rstd * mu.
        float2 st[8];
        float4 sv[4], cv[4];
#pragma unroll
        for (int ai = 0; ai < 8; ++ai) {
            int m = cm0 + (ai >> 2) * 128 + wm * 64 + (ai & 3) * 16 + (lane & 15);
            m = m < M ? m : M - 1;
            st[ai] = e.ln_stats[m];
        }
#pragma unroll
        for (int jf = 0; jf < 4; ++jf) {
            int n = n_wave + jf * 16 + (lane >> 4) * 4;
            n = n <= N - 4 ? n : N - 4;
            sv[jf] = *reinterpret_cast<const float4*>(e.ln_s + n);
            cv[jf] = *reinterpret_cast<const float4*>(e.ln_c + n);
        }
#pragma unroll
        for (int ai = 0; ai < 8; ++ai) {
            const float a = st[ai].y, b = -st[ai].y * st[ai].x;
#pragma unroll
            for (int jf = 0; jf < 4; ++jf) {
                const float sj[4] = {sv[jf].x, sv[jf].y, sv[jf].z, sv[jf].w};
                const float cj[4] = {cv[jf].x, cv[jf].y, cv[jf].z, cv[jf].w};
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[ai][jf][r] = fmaf(acc[ai][jf][r], a, fmaf(b, sj[r], cj[r]));
            }
        }
    }
    // ---- sweep A ----
    if (qkv) {
        const int D = e.q.head_dim;
        bool rot[2];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int n_grp = n_wave + pr * 32;           // 32-aligned column group (wave-uniform)
            const int head = n_grp / (3 * D);
            const int within = n_grp - head * 3 * D;
            const int part = within / D;                  // 0 q, 1 k, 2 v
            rot[pr] = part < 2 && within == part * D && n_grp < N;   // the group holds dims 0..31 of a q / k head
        }
        if (rot[0] || rot[1]) {
            // table rows of the lane's 8 output rows (independent loads first, then the dependent cos / sin rows)
            int rpos[8];
            const RowSplit rs(e.q.q_len, cm0);
#pragma unroll
            for (int ai = 0; ai < 8; ++ai) {
                int m = cm0 + (ai >> 2) * 128 + wm * 64 + (ai & 3) * 16 + (lane & 15);
                m = m < M ? m : M - 1;
                int b_, t_;
                rs.split(m, b_, t_);
                rpos[ai] = pos0 + t_;
            }
            if (e.q.pos_table) {
#pragma unroll
                for (int ai = 0; ai < 8; ++ai) rpos[ai] = e.q.pos_table[rpos[ai]];
            }
            if constexpr ((EABL & 64) == 0) {
                // round 6: the cos / sin rows of all 8 fragments in flight together -- one exposed load latency per rotated column group
                // instead of two (64 transient registers: the K-loop's fragment registers are dead here): +1.2 ... 1.7 % on the prefill
                // and both decode QKV shapes, bit-identical (profiles/r6_ppqkv_rope_one_batch_new_boundary.jsonl; EABL 64 = variant 29
                // of the debug build keeps the round-5 two-batch form as the A/B reference)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    if (!rot[pr]) continue;
                    float4 c[8], sn[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int64_t off = (int64_t)rpos[i] * 16 + (lane >> 4) * 4;
                        c[i] = *reinterpret_cast<const float4*>(e.q.cos_tab + off);
                        sn[i] = *reinterpret_cast<const float4*>(e.q.sin_tab + off);
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float cc[4] = {c[i].x, c[i].y, c[i].z, c[i].w};
                        const float ss[4] = {sn[i].x, sn[i].y, sn[i].z, sn[i].w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float x1 = acc[i][pr * 2][r], x2 = acc[i][pr * 2 + 1][r];
                            acc[i][pr * 2][r] = x1 * cc[r] - x2 * ss[r];
                            acc[i][pr * 2 + 1][r] = x2 * cc[r] + x1 * ss[r];
                        }
                    }
                }
            } else
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                if (!rot[pr]) continue;
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    float4 c[4], sn[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int64_t off = (int64_t)rpos[hb * 4 + i] * 16 + (lane >> 4) * 4;
                        c[i] = *reinterpret_cast<const float4*>(e.q.cos_tab + off);
                        sn[i] = *reinterpret_cast<const float4*>(e.q.sin_tab + off);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float cc[4] = {c[i].x, c[i].y, c[i].z, c[i].w};
                        const float ss[4] = {sn[i].x, sn[i].y, sn[i].z, sn[i].w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float x1 = acc[hb * 4 + i][pr * 2][r], x2 = acc[hb * 4 + i][pr * 2 + 1][r];
                            acc[hb * 4 + i][pr * 2][r] = x1 * cc[r] - x2 * ss[r];
                            acc[hb * 4 + i][pr * 2 + 1][r] = x2 * cc[r] + x1 * ss[r];
                        }
                    }
                }
            }
        }
    }
    // PP_RES: the residual tile is read in the STORE layout (rows of 128 B: 16 bytes per lane, 8 lanes per row, the same bytes the
    // lane writes later -- an in-place x += ... is safe) and joins the bf16-rounded result after the transposition, exactly like
    // the small-tile kernels' LDS-staged epilogue (store_tile_from_lds).  Round 3: the accumulator-layout version (32 gathers of
    // 8 bytes per lane in 32-byte row segments) cost ~7 us per tile on top of the plain epilogue; these are 16 full-line loads, all in
    // flight before the first store (the K-loop's fragment registers are dead here).
    uint4 rres[8][2];
    if (MODE == PP_RES) {
        const __amdgpu_buffer_rsrc_t res_rsrc = tile_rsrc(reinterpret_cast<const char*>(e.res) + ((int64_t)cm0 * e.ldr + n_wave) * 2);
        const uint32_t res_voff = ((uint32_t)(lane >> 3) * (uint32_t)e.ldr + (uint32_t)(lane & 7) * 8u) * 2u;
#pragma unroll
        for (int ai = 0; ai < 8; ++ai) {
            const int row_l = (ai >> 2) * 128 + wm * 64 + (ai & 3) * 16;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int m = cm0 + row_l + h * 8 + (lane >> 3);
                const int n = n_wave + (lane & 7) * 8;
                u32x4_t r4 = u32x4_t{0, 0, 0, 0};
                if (m < M && n < N)
                    r4 = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, res_voff, (uint32_t)(row_l + h * 8) * (uint32_t)e.ldr * 2u, 0);
                rres[ai][h] = uint4{r4.x, r4.y, r4.z, r4.w};
            }
        }
    }
    asm volatile("" ::: "memory");
    epi_stamp<EABL>(tl, 10);     // sweep A done (RoPE applied / residual loads issued)
    // ---- sweep B ----
    const uint32_t patch_w = lds_addr(stage) + (lane & 15) * EPI_PITCH + (lane >> 4) * 8;
    const uint32_t patch_r = lds_addr(stage) + (lane >> 3) * EPI_PITCH + (lane & 7) * 16;
    // plain output rows: buffer stores addressed as (tile-origin descriptor) + (uniform row offset in an SGPR) + (ONE lane offset
    // VGPR).  A vector-memory instruction costs the CU's memory pipe ~16 cycles per VGPR operand it reads (round-3 store
    // ablations: a tile's 128 stores of 4 data + 2 address registers drain at ~10 B/clk and stall the load ring behind them);
    // the descriptor form reads 4 + 1.
    const __amdgpu_buffer_rsrc_t out_rsrc = tile_rsrc(reinterpret_cast<const char*>(e.out) + ((int64_t)cm0 * e.ldo + n_wave) * 2);
    const uint32_t out_voff = ((uint32_t)(lane >> 3) * (uint32_t)e.ldo + (uint32_t)(lane & 7) * 8u) * 2u;
    // QKV mode, K / V columns: the same descriptor form into the cache.  A wave's 64 columns lie inside one head's q, k or v part
    // (head_dim % 64 == 0), so head / part are scalars, and the cache row of output row r of the tile is an affine function of r
    // for q_len == 1 (one sequence per row: stride = one sequence of the cache), for q_len >= 256 (consecutive positions of one
    // sequence, + one jump where the tile crosses into the next sequence) and for power-of-two q_len (shift / mask).  The generic
    // form below (two integer divisions and a 64-bit address per lane and store: ~80 VALU instructions x 16 stores per wave
    // and tile) made the fused epilogue cost +15 % of a prefill QKV GEMM; it stays for cut tiles, other q_len and > 2 GiB spans.
    int kv_mode = 0;                       // 0 generic, 1 q_len == 1, 2 q_len >= 256 (one wrap), 3 power-of-two q_len
    int part_u = 0, kv_shift = 0, kv_thr0 = 0;
    uint32_t kv_voff = 0, kv_row = 0, kv_seq = 0, kv_delta = 0;
    __amdgpu_buffer_rsrc_t kv_rsrc = out_rsrc;
    if (qkv) {
        const int D = e.q.head_dim, q_len = e.q.q_len;
        const int64_t seq_bytes = (int64_t)e.q.heads * e.q.n_max * D * 2;
        const bool pow2 = (q_len & (q_len - 1)) == 0;
        if (!(EABL & 16) && D % 64 == 0 && cm0 + BM2 <= M && n_wave + 64 <= N && seq_bytes * (q_len >= BM2 ? 1 : BM2) < (int64_t)0x7ff00000 &&
            (q_len == 1 || q_len >= BM2 || pow2)) {
            const int head = n_wave / (3 * D);
            const int within = n_wave - head * 3 * D;
            part_u = within / D;
            const int el0 = within - part_u * D;
            const RowSplit rs(q_len, cm0);
            int64_t b_base = cm0, t_base = 0;
            if (q_len == 1) {
                kv_mode = 1;
                kv_row = (uint32_t)seq_bytes;
            } else if (q_len >= BM2) {
                kv_mode = 2;
                b_base = rs.b0;
                t_base = rs.t0;
                kv_row = (uint32_t)(D * 2);
                kv_delta = (uint32_t)(seq_bytes - (int64_t)q_len * D * 2);
                kv_thr0 = q_len - rs.t0;             // tile rows >= kv_thr0 belong to the next sequence
            } else {
                kv_mode = 3;
                kv_shift = 31 - __builtin_clz(q_len);
                b_base = cm0 >> kv_shift;            // cm0 is a multiple of 256 >= q_len: t_base = 0
                kv_row = (uint32_t)(D * 2);
                kv_seq = (uint32_t)seq_bytes;
            }
            if (part_u != 0) {
                const bf16* cache = part_u == 1 ? e.q.kc : e.q.vc;
                kv_rsrc = tile_rsrc(reinterpret_cast<const char*>(cache) +
                                    (((b_base * e.q.heads + head) * e.q.n_max + pos0 + t_base) * D + el0) * 2);
                kv_voff = (uint32_t)(lane >> 3) * kv_row + (uint32_t)(lane & 7) * 16u;
            }
        }
    }
    epi_stamp<EABL>(tl, 11);     // store addressing set up
#pragma unroll
    for (int ai = 0; ai < 8; ++ai) {
        if (ai == 4) epi_stamp<EABL>(tl, 12);     // half of the fragments converted / transposed / stored
        // EABL 512 (round-6 experiment, variant 97): the two waves of a SIMD (one of each group) take turns at raised priority, fragment by
        // fragment -- with equal priority the older wave (group 0) wins the VALU, finishes its epilogue first and waits for group 1's
        if constexpr ((EABL & 512) != 0) {
            if ((ai & 1) == wm) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        const int row_l = (ai >> 2) * 128 + wm * 64 + (ai & 3) * 16;
        uint2 o[4];
#pragma unroll
        for (int jf = 0; jf < 4; ++jf) {
            float v[4] = {acc[ai][jf][0], acc[ai][jf][1], acc[ai][jf][2], acc[ai][jf][3]};
#ifdef BT_DEBUG_HOOKS
            if (MODE == PP_GELU && (EABL & 128) != 0) gelu4_r5(v);
            else
#endif
            if (MODE == PP_GELU) gelu4(v);
            o[jf].x = pack_bf16x2(v[0], v[1]);
            o[jf].y = pack_bf16x2(v[2], v[3]);
        }
        if (MODE == PP_ARGMAX) {
            // the lane's 16 columns of output row (lane & 15) in increasing column order, then the 4 lanes that share the row
            const int mrow = cm0 + row_l + (lane & 15);
            float best = -INFINITY;
            int bidx = 0x7fffffff;
#pragma unroll
            for (int jf = 0; jf < 4; ++jf) {
                const int n = n_wave + jf * 16 + (lane >> 4) * 4;
                const float v[4] = {__uint_as_float(o[jf].x << 16), __uint_as_float(o[jf].x & 0xffff0000u),
                                    __uint_as_float(o[jf].y << 16), __uint_as_float(o[jf].y & 0xffff0000u)};
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < N && (v[r] > best || bidx == 0x7fffffff)) { best = v[r]; bidx = n + r; }
            }
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {
                const float ov = __shfl_xor(best, off, 64);
                const int oi = __shfl_xor(bidx, off, 64);
                if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
            }
            if ((lane >> 4) == 0 && mrow < M && n_wave < N)
                e.amax[(int64_t)mrow * e.amax_ld + (n_wave >> 6)] = uint2{__float_as_uint(best), (uint32_t)bidx};
            if (e.out == nullptr) continue;
        }
        // registers -> wave-private patch -> registers, transposed to 16 B per lane along rows (inline asm: the compiler
        // must not see LDS traffic here, it would fence it against the in-flight LDS-DMA with vmcnt(0))
        uint4 v0, v1;
        if (EABL & 2) {
            v0 = uint4{o[0].x, o[0].y, o[1].x, o[1].y};
            v1 = uint4{o[2].x, o[2].y, o[3].x, o[3].y};
        } else {
            asm volatile("ds_write_b64 %2, %4\n\tds_write_b64 %2, %5 offset:32\n\t"
                         "ds_write_b64 %2, %6 offset:64\n\tds_write_b64 %2, %7 offset:96\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:%8\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1)
                         : "v"(patch_w), "v"(patch_r), "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "n"(8 * EPI_PITCH)
                         : "memory");
        }
        if ((EABL & 1) || ((EABL & 8) && ai > 0)) {   // EABL 8: only the first fragment's stores (2 per wave and tile)
            asm volatile("" ::"v"(v0.x), "v"(v0.y), "v"(v0.z), "v"(v0.w), "v"(v1.x), "v"(v1.y), "v"(v1.z), "v"(v1.w));
            continue;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if ((EABL & 4) && h == 1) {                // EABL 4: half of the stores (8 per wave and tile)
                asm volatile("" ::"v"(v1.x), "v"(v1.y), "v"(v1.z), "v"(v1.w));
                continue;
            }
            uint4 v = h ? v1 : v0;
            if (MODE == PP_RES) v = add_bf16x8(v, rres[ai][h]);
            const int m = cm0 + row_l + h * 8 + (lane >> 3);
            const int n = n_wave + (lane & 7) * 8;
            if (m < M && n < N) {
                bf16* dst;
                if (!qkv) {
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{v.x, v.y, v.z, v.w}, out_rsrc, out_voff,
                                                           (uint32_t)(row_l + h * 8) * (uint32_t)e.ldo * 2u, 0);
                    continue;
                } else if (kv_mode != 0) {
                    const uint32_t r0 = (uint32_t)(row_l + h * 8);          // first tile row of this store (scalar)
                    if (part_u == 0 || (EABL & 1024) != 0) {                 // q columns: rows of the qkv buffer  (EABL 1024, variant 98: timing
                                                                             //  experiment, WRONG results -- k / v go there too)
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{v.x, v.y, v.z, v.w}, out_rsrc, out_voff, r0 * (uint32_t)e.ldo * 2u, 0);
                    } else if (kv_mode == 3) {
                        const uint32_t r = r0 + (uint32_t)(lane >> 3);
                        const uint32_t vo = (r >> kv_shift) * kv_seq + (r & ((1u << kv_shift) - 1u)) * kv_row + (uint32_t)(lane & 7) * 16u;
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{v.x, v.y, v.z, v.w}, kv_rsrc, vo, 0, 0);
                    } else {
                        uint32_t vo = kv_voff;
                        if (kv_mode == 2 && (int)(r0 + (uint32_t)(lane >> 3)) >= kv_thr0) vo += kv_delta;
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{v.x, v.y, v.z, v.w}, kv_rsrc, vo, r0 * kv_row, 0);
                    }
                    continue;
                } else {
                    const int D = e.q.head_dim;
                    const int head = n / (3 * D);
                    const int within = n - head * 3 * D;
                    const int part = within / D;
                    const int el = within - part * D;
                    int bi, t;
                    RowSplit(e.q.q_len, cm0).split(m, bi, t);
                    const int64_t b = bi;
                    bf16* cache = part == 1 ? e.q.kc : e.q.vc;
                    dst = part == 0 ? e.out + (int64_t)m * e.ldo + n
                                    : cache + ((b * e.q.heads + head) * e.q.n_max + pos0 + t) * D + el;
                }
                *reinterpret_cast<uint4*>(dst) = v;
            }
        }
    }
    if constexpr ((EABL & 512) != 0) __builtin_amdgcn_s_setprio(0);
}

// ---- MFMA shape policies of the persistent kernel -----------------------------------------------------------------------
// Mma16: v_mfma_f32_16x16x32_bf16, wave tile = 8 x 4 fragments (production).  Mma32: v_mfma_f32_32x32x16_bf16, wave tile =
// 4 x 2 blocks, four K = 16 steps per K-tile; same LDS image and bytes read per K-tile (a 32-row fragment read is the
// 16-row one with row = lane & 31 and 16-byte slot = 2 * kstep + (lane >> 5)).  The 32x32 form has the higher
// micro-benchmark floor (guide: 2382 vs 2075 TF) but measured 5-8 % SLOWER in this two-waves-per-SIMD ping-pong schedule
// (profiles/r1_gemm_ab_mma32_vs_mma16.jsonl), so it stays a test / A-B variant (bt_debug_gemm_variant(14)).
typedef float f32x16_t __attribute__((ext_vector_type(16)));

__device__ inline bf16x8_t read_frag32(const char* lds_tile, int row_base, int ks, int lane) {
    const int row = row_base + (lane & 31);
    const int slot = (ks * 2 + (lane >> 5)) ^ (row & 7);
    return *reinterpret_cast<const bf16x8_t*>(lds_tile + row * 128 + slot * 16);
}

__device__ inline f32x4_t bf16x4_to_f32(uint64_t b) {
    const uint32_t lo = (uint32_t)b, hi = (uint32_t)(b >> 32);
    return f32x4_t{__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u), __uint_as_float(hi << 16),
                   __uint_as_float(hi & 0xffff0000u)};
}

struct Mma16 {
    typedef f32x4_t Acc;
    static constexpr int MI = 4, NB = 4, KS = 2;
    static constexpr int PATCH_BYTES = PP_EPI_WAVE;
    __device__ static inline bf16x8_t read_a(const char* half, int wm, int i, int ks, int lane) {
        return read_frag(half, wm * 64 + i * 16, ks, lane);
    }
    __device__ static inline bf16x8_t read_w(const char* half, int wrow, int nb, int ks, int lane) {
        return read_frag(half, wrow + nb * 16, ks, lane);
    }
    __device__ static inline Acc mma(const bf16x8_t& w, const bf16x8_t& a, const Acc& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, a, c, 0, 0, 0);
    }
    // accumulators <- bias row in LDS (bias_row = LDS byte address of the tile's 256-column bf16 row), or zero
    // has_bias2: the second half of the 1 KiB row holds a second bias (bt_linear2) -> start at the sum of the two
    __device__ static inline void init(Acc (&acc)[8][4], bool has_bias, bool has_bias2, uint32_t bias_row, int wcol, int lane) {
        if (has_bias) {
            // inline asm: a compiler-visible LDS read of a DMA-written region would get a vmcnt(0) in front
            uint64_t b[4];
            const uint32_t ba = bias_row + (wcol + (lane >> 4) * 4) * 2;
            asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:32\n\t"
                         "ds_read_b64 %2, %4 offset:64\n\tds_read_b64 %3, %4 offset:96\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3])
                         : "v"(ba)
                         : "memory");
            f32x4_t bv[4];
#pragma unroll
            for (int jf = 0; jf < 4; ++jf) bv[jf] = bf16x4_to_f32(b[jf]);
            if (has_bias2) {
                asm volatile("ds_read_b64 %0, %4 offset:512\n\tds_read_b64 %1, %4 offset:544\n\t"
                             "ds_read_b64 %2, %4 offset:576\n\tds_read_b64 %3, %4 offset:608\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3])
                             : "v"(ba)
                             : "memory");
#pragma unroll
                for (int jf = 0; jf < 4; ++jf) bv[jf] += bf16x4_to_f32(b[jf]);
            }
#pragma unroll
            for (int jf = 0; jf < 4; ++jf) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i][jf] = bv[jf];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int jf = 0; jf < 4; ++jf) acc[i][jf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
    }
    template <int MODE, int EABL = 0>
    __device__ static inline void epilogue(const EpiArgs& e, int pos0, char* stage, Acc (&acc)[8][4], int cm0, int n_wave,
                                           int wm, int lane, uint32_t tl = 0) {
        pp_epilogue<MODE, EABL>(e, pos0, stage, acc, cm0, n_wave, wm, lane, tl);
    }
};

#ifdef BT_DEBUG_HOOKS   // round 5: the 32x32 MFMA policy (measured 5-8 % slower) and the first-generation persistent
                        // kernel (A/B reference of the second generation; its two-operand-pair form measured 4-7 % slower than two
                        // launches) live in the debug library only -- the product binary carries what runs
// 32x32 blocks: lane holds row m = lane & 31 of the block and, per q = 0..3, the 4 consecutive columns 8q + 4(lane >> 5) + r
constexpr int EPI32_PITCH = 80;                  // 32 columns x 2 B + 16 pad
constexpr int PP_EPI32_WAVE = 32 * EPI32_PITCH;  // 2.5 KiB per wave
static_assert(PP_EPI_OFF + 8 * PP_EPI32_WAVE <= PP_BIAS_OFF && PP_EPI_OFF + 8 * PP_EPI_WAVE <= PP_BIAS_OFF, "wave patches overlap the bias rows");

template <int MODE>
__device__ inline void pp32_epilogue(const EpiArgs& e, int pos0, char* stage, f32x16_t (&acc)[4][2], int cm0, int n_wave, int wm,
                                     int lane) {
    const int M = e.M, N = e.N;
    constexpr bool qkv = MODE == PP_QKV;
    const int lrow = lane & 31, lcol = (lane >> 5) * 4;
    // ---- sweep A: every load of the epilogue, before its first store ----
    if (qkv) {
        const int D = e.q.head_dim;
        bool rot[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int n_grp = n_wave + nb * 32;
            const int head = n_grp / (3 * D);
            const int within = n_grp - head * 3 * D;
            const int part = within / D;
            rot[nb] = part < 2 && within == part * D && n_grp < N;   // dims 0..31 of a q / k head
        }
        if (rot[0] || rot[1]) {
            int rpos[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                int m = cm0 + (mb >> 1) * 128 + wm * 64 + (mb & 1) * 32 + lrow;
                m = m < M ? m : M - 1;
                rpos[mb] = pos0 + m % e.q.q_len;
            }
            if (e.q.pos_table) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) rpos[mb] = e.q.pos_table[rpos[mb]];
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                if (!rot[nb]) continue;
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    float4 c[2][2], sn[2][2];   // [m-block][q]: table columns 8q + lcol .. + 3
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int64_t off = (int64_t)rpos[hb * 2 + i] * 16 + q * 8 + lcol;
                            c[i][q] = *reinterpret_cast<const float4*>(e.q.cos_tab + off);
                            sn[i][q] = *reinterpret_cast<const float4*>(e.q.sin_tab + off);
                        }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const float cc[4] = {c[i][q].x, c[i][q].y, c[i][q].z, c[i][q].w};
                            const float ss[4] = {sn[i][q].x, sn[i][q].y, sn[i][q].z, sn[i][q].w};
                            f32x16_t& a = acc[hb * 2 + i][nb];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {   // column 8q + lcol + r pairs with the same + 16 = register q + 2
                                const float x1 = a[q * 4 + r], x2 = a[(q + 2) * 4 + r];
                                a[q * 4 + r] = x1 * cc[r] - x2 * ss[r];
                                a[(q + 2) * 4 + r] = x2 * cc[r] + x1 * ss[r];
                            }
                        }
                }
            }
        }
    } else if (MODE == PP_RES) {   // residual joins the fp32 accumulators
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            uint2 rv[2][2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = cm0 + hb * 128 + wm * 64 + i * 32 + lrow;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = n_wave + nb * 32 + q * 8 + lcol;
                        rv[i][nb][q] = (m < M && n < N) ? *reinterpret_cast<const uint2*>(e.res + (int64_t)m * e.ldr + n)
                                                       : uint2{0, 0};
                    }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x16_t& a = acc[hb * 2 + i][nb];
                        a[q * 4 + 0] += __uint_as_float(rv[i][nb][q].x << 16);
                        a[q * 4 + 1] += __uint_as_float(rv[i][nb][q].x & 0xffff0000u);
                        a[q * 4 + 2] += __uint_as_float(rv[i][nb][q].y << 16);
                        a[q * 4 + 3] += __uint_as_float(rv[i][nb][q].y & 0xffff0000u);
                    }
        }
    }
    asm volatile("" ::: "memory");
    // ---- sweep B: per 32 x 32 block: registers -> wave-private LDS patch -> 16 B per lane along rows -> 2 stores ----
    const uint32_t patch_w = lds_addr(stage) + lrow * EPI32_PITCH + lcol * 2;
    const uint32_t patch_r = lds_addr(stage) + (lane >> 2) * EPI32_PITCH + (lane & 3) * 16;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const int row_l = (mb >> 1) * 128 + wm * 64 + (mb & 1) * 32;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            uint2 o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4] = {acc[mb][nb][q * 4], acc[mb][nb][q * 4 + 1], acc[mb][nb][q * 4 + 2], acc[mb][nb][q * 4 + 3]};
                if (MODE == PP_GELU) gelu4(v);
                o[q].x = pack_bf16x2(v[0], v[1]);
                o[q].y = pack_bf16x2(v[2], v[3]);
            }
            uint4 v0, v1;
            asm volatile("ds_write_b64 %2, %4\n\tds_write_b64 %2, %5 offset:16\n\t"
                         "ds_write_b64 %2, %6 offset:32\n\tds_write_b64 %2, %7 offset:48\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:%8\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1)
                         : "v"(patch_w), "v"(patch_r), "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "n"(16 * EPI32_PITCH)
                         : "memory");
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint4 v = h ? v1 : v0;
                const int m = cm0 + row_l + h * 16 + (lane >> 2);
                const int n = n_wave + nb * 32 + (lane & 3) * 8;
                if (m < M && n < N) {
                    bf16* dst;
                    if (!qkv) {
                        dst = e.out + (int64_t)m * e.ldo + n;
                    } else {
                        const int D = e.q.head_dim;
                        const int head = n / (3 * D);
                        const int within = n - head * 3 * D;
                        const int part = within / D;
                        const int el = within - part * D;
                        const int64_t b = m / e.q.q_len;
                        const int t = m - (int)b * e.q.q_len;
                        bf16* cache = part == 1 ? e.q.kc : e.q.vc;
                        dst = part == 0 ? e.out + (int64_t)m * e.ldo + n
                                        : cache + ((b * e.q.heads + head) * e.q.n_max + pos0 + t) * D + el;
                    }
                    *reinterpret_cast<uint4*>(dst) = v;
                }
            }
        }
    }
}

struct Mma32 {
    typedef f32x16_t Acc;
    static constexpr int MI = 2, NB = 2, KS = 4;
    static constexpr int PATCH_BYTES = PP_EPI32_WAVE;
    __device__ static inline bf16x8_t read_a(const char* half, int wm, int i, int ks, int lane) {
        return read_frag32(half, wm * 64 + i * 32, ks, lane);
    }
    __device__ static inline bf16x8_t read_w(const char* half, int wrow, int nb, int ks, int lane) {
        return read_frag32(half, wrow + nb * 32, ks, lane);
    }
    __device__ static inline Acc mma(const bf16x8_t& w, const bf16x8_t& a, const Acc& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a, c, 0, 0, 0);
    }
    __device__ static inline void init(Acc (&acc)[4][2], bool has_bias, bool /*has_bias2: Mma16 only*/, uint32_t bias_row, int wcol, int lane) {
        if (has_bias) {
            uint64_t b[8];   // [nb][q]: columns nb * 32 + 8q + 4 (lane >> 5) .. + 3
            const uint32_t ba = bias_row + (wcol + (lane >> 5) * 4) * 2;
            asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:16\n\t"
                         "ds_read_b64 %2, %8 offset:32\n\tds_read_b64 %3, %8 offset:48\n\t"
                         "ds_read_b64 %4, %8 offset:64\n\tds_read_b64 %5, %8 offset:80\n\t"
                         "ds_read_b64 %6, %8 offset:96\n\tds_read_b64 %7, %8 offset:112\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]), "=&v"(b[4]), "=&v"(b[5]), "=&v"(b[6]),
                           "=&v"(b[7])
                         : "v"(ba)
                         : "memory");
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                f32x16_t bv;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4_t x = bf16x4_to_f32(b[nb * 4 + q]);
                    bv[q * 4] = x[0]; bv[q * 4 + 1] = x[1]; bv[q * 4 + 2] = x[2]; bv[q * 4 + 3] = x[3];
                }
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) acc[mb][nb] = bv;
            }
        } else {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
        }
    }
    template <int MODE>
    __device__ static inline void epilogue(const EpiArgs& e, int pos0, char* stage, Acc (&acc)[4][2], int cm0, int n_wave,
                                           int wm, int lane) {
        pp32_epilogue<MODE>(e, pos0, stage, acc, cm0, n_wave, wm, lane);
    }
};

struct LoadCursor {   // wave-uniform: the K-tile whose half-tiles are being issued
    int m0, n0, t, j, x;
    int kt, rot;          // physical K-tile (== t unless the K order is rotated, see PP_FLAG_KROT) and the tile's rotation
    bool valid;
};

// ABL (debug build only; 1..16 are timing ablations with WRONG results): 1 = no LDS-DMA after the prologue, 2 = no fragment
// ds_reads after the first phase, 4 = no epilogue, 8 = every K-tile reads K-tile 0 (all operand loads hit in L2), 16 = no MFMAs.
// 32 = K-order rotation (correct results): each workgroup walks the K-tiles of a tile in groups of 8 starting at a
//   tile-dependent offset ((tn + 2 tm) & 7), so the workgroups of an XCD that share an A / W panel do not all request the
//   same lines at the same moment (the first toucher pays the fabric latency, the others then hit in L2).  The sum over K
//   is order-independent up to fp32 rounding; the order is a pure function of the tile.  Needs K % 512 == 0.

template <int MODE, typename P, bool SEG2 = false, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_bf16_mfma256pp_kernel(const bf16* __restrict__ A, int64_t lda,
                                                                 const bf16* __restrict__ W, int64_t ldw, const EpiArgs ep,
                                                                 int K, int tiles_m, int tiles_n, const Seg2 s2) {
    const int M = ep.M, N = ep.N;
    __shared__ __attribute__((aligned(16))) char smem[PP_SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int wh = wn >> 1;                      // W half this wave multiplies
    const int wrow = (wn & 1) * 64;              // its 64-row slab inside the half == 64 contiguous output columns
    const int wcol = wh * 128 + wrow;            // column offset of the wave's output inside the tile
    const bool grp1 = wave >= 4;
    char* stage = smem + PP_EPI_OFF + wave * P::PATCH_BYTES;
    const uint32_t lane_slot = (uint32_t)(((lane & 7) ^ ((lane >> 3) & 7)) * 16);
    const uint32_t voff_a = (uint32_t)(lane >> 3) * (uint32_t)(lda * 2) + lane_slot;
    const uint32_t voff_w = (uint32_t)(lane >> 3) * (uint32_t)(ldw * 2) + lane_slot;
    // optional second operand pair (bt_linear2): K-tiles nk1 .. nk-1 stream s2.x / s2.w into the same accumulators
    const uint32_t voff_a2 = (uint32_t)(lane >> 3) * (uint32_t)(s2.ldx * 2) + lane_slot;
    const uint32_t voff_w2 = (uint32_t)(lane >> 3) * (uint32_t)(s2.ldw * 2) + lane_slot;

    const int nk1 = K / BK;
    const int nk = nk1 + (SEG2 ? s2.K / BK : 0);
    // tiles of this workgroup: rounds 0 .. nt-1 (all workgroups have the full rounds; the last round may be partial)
    const int nwg = tiles_m * tiles_n;
    const int G = gridDim.x;
    int nt = nwg / G;
    {
        int mm, nn;
        if (nwg % G != 0 && pp_tile(nt, tiles_m, tiles_n, mm, nn)) ++nt;
    }
    const int total = 4 * nk * nt;

    const bool krot = !SEG2 && (ABL & 32) && (nk1 & 7) == 0;
    LoadCursor lc;
    lc.t = 0; lc.j = 0; lc.x = 0;
    lc.valid = pp_tile(0, tiles_m, tiles_n, lc.m0, lc.n0);
    lc.rot = krot ? (((lc.n0 >> 8) + 2 * (lc.m0 >> 8)) & 7) : 0;
    lc.kt = lc.rot;
    auto issue = [&](int kind) {
        if ((ABL & 1) && lc.x >= 2) return;
        if (lc.valid) {
            if (!SEG2 || lc.t < nk1)
                stage_half_at(A, lda, W, ldw, lc.m0, lc.n0, M, N, ((ABL & 8) ? 0 : lc.kt) * BK, lc.x & 1, kind, smem, wave, lane, voff_a, voff_w);
            else
                stage_half_at(s2.x, s2.ldx, s2.w, s2.ldw, lc.m0, lc.n0, M, N, (lc.t - nk1) * BK, lc.x & 1, kind, smem, wave,
                              lane, voff_a2, voff_w2);
        }
    };
    // wave 0: bias[n0 .. n0+255] of the cursor's tile -> LDS row (j & 1), before that tile's loads.  The row is 1 KiB = 64
    // lanes x 16 B but the tile's bias is 512 B: lanes 32..63 fetch the second pair's bias (or the same bytes again)
    const bf16* bias_hi = (SEG2 && s2.bias) ? s2.bias : ep.bias;
    auto issue_bias = [&]() {
        if (wave == 0 && ep.bias && lc.valid) {
            int col = lc.n0 + (lane & 31) * 8;
            col = col <= N - 8 ? col : N - 8;
            glds16_vaddr((lane < 32 ? ep.bias : bias_hi) + col, lds_addr(smem + PP_BIAS_OFF + (lc.j & 1) * 1024));
        }
    };
    auto advance = [&]() {
        ++lc.x;
        if (++lc.t == nk) {
            lc.t = 0;
            ++lc.j;
            lc.valid = lc.j < nt && pp_tile(lc.j, tiles_m, tiles_n, lc.m0, lc.n0);
            if (krot) lc.rot = ((lc.n0 >> 8) + 2 * (lc.m0 >> 8)) & 7;
        }
        lc.kt = krot ? ((lc.t & ~7) | ((lc.t + lc.rot) & 7)) : lc.t;
    };
    // prologue: K-tile 0 (4 half-tiles) + K-tile 1 (A0, W0) = PD2 loads
    issue_bias();
    issue(0); issue(1); issue(2); issue(3);
    advance();
    if (lc.t == 0) issue_bias();
    issue(0); issue(1);

    typename P::Acc acc[2 * P::MI][P::NB];   // [A half * MI + m-fragment][n-fragment]
    bf16x8_t af[P::MI][P::KS];
    bf16x8_t wf[P::NB][P::KS];
    // fused-QKV mode: the cache write position is constant for the launch -> one scalar, read before the pipeline starts
    int pos0 = 0;
    if (MODE == PP_QKV) pos0 = __builtin_amdgcn_readfirstlane(ep.q.pos0_dev ? *ep.q.pos0_dev : ep.q.pos0_host);

    if (grp1) {
        wait_for_phase_pp(0, total, false);
        cbarrier();
    }
    int g = 0;
    int cm0 = 0, cn0 = 0;
    bool prev_full = false;
    for (int j = 0; j < nt; ++j) {
        pp_tile(j, tiles_m, tiles_n, cm0, cn0);
        for (int t = 0; t < nk; ++t) {
            const char* buf = smem + (g & 2) * 2 * HALF_BYTES;   // ring buffer of K-tile g >> 1
#pragma unroll
            for (int p = 0; p < 2; ++p, ++g) {
                if (!grp1) wait_for_phase_pp(g, total, prev_full && t == 0);
                cbarrier();
                const bool do_reads = !(ABL & 2) || g == 0;
                if (p == 0) {
                    // new tile: accumulators start at the bias (LDS row j & 1, landed before this barrier)
                    if (t == 0) P::init(acc, ep.bias != nullptr, SEG2 && s2.bias != nullptr, lds_addr(smem + PP_BIAS_OFF + (j & 1) * 1024), wcol, lane);
                    if (do_reads) {
#pragma unroll
                        for (int ks = 0; ks < P::KS; ++ks) {
#pragma unroll
                            for (int nb = 0; nb < P::NB; ++nb)
                                wf[nb][ks] = P::read_w(buf + (2 + wh) * HALF_BYTES, wrow, nb, ks, lane);
#pragma unroll
                            for (int i = 0; i < P::MI; ++i) af[i][ks] = P::read_a(buf, wm, i, ks, lane);
                        }
                    }
                    issue(2); issue(3);
                    advance();
                } else {
                    if (!(ABL & 2)) {
#pragma unroll
                        for (int ks = 0; ks < P::KS; ++ks)
#pragma unroll
                            for (int i = 0; i < P::MI; ++i) af[i][ks] = P::read_a(buf + HALF_BYTES, wm, i, ks, lane);
                    }
                    if (lc.t == 0) issue_bias();
                    issue(0); issue(1);
                }
                if (grp1) wait_for_phase_pp(g + 1, total, prev_full && t == 0 && p == 0);
                cbarrier();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                if (!(ABL & 16)) {
#pragma unroll
                    for (int ks = 0; ks < P::KS; ++ks)
#pragma unroll
                        for (int i = 0; i < P::MI; ++i)
#pragma unroll
                            for (int nb = 0; nb < P::NB; ++nb)
                                acc[p * P::MI + i][nb] = P::mma(wf[nb][ks], af[i][ks], acc[p * P::MI + i][nb]);
                } else {
#pragma unroll
                    for (int ks = 0; ks < P::KS; ++ks) {
#pragma unroll
                        for (int i = 0; i < P::MI; ++i) asm volatile("" ::"v"(af[i][ks]));
#pragma unroll
                        for (int nb = 0; nb < P::NB; ++nb) asm volatile("" ::"v"(wf[nb][ks]));
                    }
                }
                __builtin_amdgcn_s_setprio(0);
            }
        }
        if (ABL & 4) {   // timing ablation: keep the accumulators live, no epilogue (no stores: prev_full stays false)
#pragma unroll
            for (int i = 0; i < 2 * P::MI; ++i)
#pragma unroll
                for (int nb = 0; nb < P::NB; ++nb) asm volatile("" ::"v"(acc[i][nb][0]));   // (one element keeps the MFMA chain live)
            continue;
        }
        // the epilogue's lane-derived addresses must not be hoisted out of the tile loop (they would sit in VGPRs through
        // the K-loop, which has none to spare): hand it an opaque copy of the lane id
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        P::template epilogue<MODE>(ep, pos0, stage, acc, cm0, cn0 + wcol, wm, lane_e);
        prev_full = (cm0 + BM2 <= M) && (cn0 + BN2 <= N);
    }
    if (!grp1) cbarrier();
}

#endif  // BT_DEBUG_HOOKS (Mma32, first-generation persistent kernel)

// ---- persistent kernel, second generation of the K-loop (round 3) --------------------------------------------------------------
// Same tile walk, LDS ring, fragment mapping, MFMA order, per-wave epilogue and -- phase for phase -- the same vmcnt values as
// gemm_bf16_mfma256pp_kernel; what changes is the SCALAR code around them.  The round-3 timing ablations
// (profiles/r3_pp_v1_ablation.jsonl) showed the MFMA + barrier skeleton of that kernel at 1.63 PF and the full kernel at 1.07-1.25 PF,
// and its ISA showed why: every phase ran a ~30-instruction SALU branch tree to pick its s_waitcnt immediate BETWEEN the MFMA
// cluster and the barrier the partner wave group is parked at (pure matrix-pipe idle time, four times per K-tile), and each
// half-tile issue recomputed a 64-bit row address with s_mul_hi / s_mul chains (~25 SALU per half-tile).  Here:
//   * the K-loop is split by position in the load stream: K-tile 0 of a tile (the only place the store-count waits apply),
//     the steady state (constant immediates: vmcnt(6) opens phase 0, vmcnt(8) phase 1 -- what wait_for_phase_pp evaluates to
//     whenever the issue count is not clipped by the end of the stream; tests/test_gemm_ring_schedule.py checks the
//     equality) and the last two K-tiles of the workgroup's LAST tile (the generic function, as before);
//   * a full tile's half-tiles are addressed as (SGPR base of the tile's half) + (per-lane 32-bit offset that advances by
//     128 B per K-tile): one s_add into M0 + the DMA per instruction, four v_add per K-tile, no multiplies; tiles cut by the
//     matrix edge keep the per-lane clamped path (stage_half_at).
//   * tile boundary: a tile's 128 output stores (16 per wave) occupy the CU's vector-memory pipe for ~5 us (~16 cycles per VGPR
//     operand of a VMEM instruction; profiles/r3_pp2_store_count_scaling.jsonl: the cost scales with the NUMBER of stores, whatever their
//     destination), and everything issued behind them queues.  So the two half-tiles the next tile's first phase would issue
//     (W1, A1 of its K-tile 1) go out BEFORE the epilogue's stores, behind one extra barrier per tile (their slots were last read
//     in the tile's final phase): the ring then holds two full K-tiles = four phases of MFMA work that do not depend on anything
//     queued behind the stores, and the store-count waits (+ PP_STORES) extend over K-tiles 0 and 1.
// Bit-identical results to the first-generation kernel (same operand order, same accumulation order).
constexpr int PP2_WAIT_P0 = 6, PP2_WAIT_P1 = 8;   // steady-state vmcnt immediates (+ PP_STORES in K-tiles 0 / 1 behind a full tile)

template <int OFF>
__device__ inline void glds16_sbase_off(const char* base_uniform, uint32_t voff, uint32_t lds_base_uniform) {
    // M0 = LDS base + OFF in one SALU op; the s_nop covers the SALU-writes-M0 -> LDS-DMA wait state
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(voff), "s"(base_uniform), "s"(lds_base_uniform), "n"(OFF)
                 : "memory", "scc");
}

// (Measured and rejected, round 3, profiles/r3_ppab_dmac.jsonl: issuing the second half-tile of each phase in the middle of the
//  phase's MFMA cluster -- shorter load segment, two DMA issues among the MFMAs -- is 3-8 % SLOWER on every shape: the MFMA
//  cluster, not the load segment, is the critical resource of a phase.)
template <int MODE, int ABL = 0, bool LNF = false>
__global__ __launch_bounds__(512) void gemm_bf16_mfma256pp2_kernel(const bf16* __restrict__ A, int64_t lda,
                                                                  const bf16* __restrict__ W, int64_t ldw, const EpiArgs ep,
                                                                  int K, int tiles_m, int tiles_n) {
    typedef Mma16 P;
    const int M = ep.M, N = ep.N;
    // ABL bit 131072 (round 6, variant 28: the tile-boundary TIMELINE): every wave stamps s_memtime at ten points of each of its first
    // twelve tiles into 8 KiB of LDS behind the kernel's own (inline asm: the compiler must not see LDS traffic), and workgroups 0 .. 3
    // dump them to ep.amax (unused outside the ARGMAX mode) at the end.  Correct results; tools/kernel_probe.py pptl reads the buffer.
    constexpr bool TL = (ABL & 131072) != 0;
    constexpr int TL_EV = 13, TL_TILES = 9;
    // ABL bit 1048576 (variant 31 / 32 with the timeline): the round-5 tile boundary -- group 1's extra barrier in FRONT of its refill, i.e.
    // the two groups' epilogues one after the other (A/B reference of the round-6 boundary, bit-identical results)
    constexpr bool SERIAL_EPI = (ABL & 1048576) != 0;
    __shared__ __attribute__((aligned(16))) char smem[PP_SMEM + (TL ? 8192 : 0)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int wh = wn >> 1;
    const int wrow = (wn & 1) * 64;
    const int wcol = wh * 128 + wrow;
    const bool grp1 = wave >= 4;
    char* stage = smem + PP_EPI_OFF + wave * P::PATCH_BYTES;
    const uint32_t tl_lds = lds_addr(smem) + (uint32_t)PP_SMEM + (uint32_t)wave * 1024u;
    auto stamp = [&](int j, int ev) {
        if constexpr (TL) {
            if (j < TL_TILES) {
                uint64_t tm;
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm) : : "memory");
                const uint32_t a = tl_lds + (uint32_t)(j * TL_EV + ev) * 8u;
                asm volatile("ds_write_b64 %0, %1" : : "v"(a), "v"(tm) : "memory");
            }
        }
    };
    if constexpr (TL) {     // slot 117: kernel entry (118: exit, 119: s_memrealtime at entry -- 100 MHz, comparable across workgroups)
        uint64_t tm, rt;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm), "=s"(rt) : : "memory");
        asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:16" : : "v"(tl_lds + 117u * 8u), "v"(tm), "v"(rt) : "memory");
    }
    const uint32_t lane_slot = (uint32_t)(((lane & 7) ^ ((lane >> 3) & 7)) * 16);
    // lane part of a DMA address inside its 16-row group: row (lane >> 3) of rows 0..7 / 8..15, swizzled 16-byte slot
    const uint32_t voff_a_0 = (uint32_t)(lane >> 3) * (uint32_t)(lda * 2) + lane_slot;
    const uint32_t voff_w_0 = (uint32_t)(lane >> 3) * (uint32_t)(ldw * 2) + lane_slot;
    const uint32_t row8_a = (uint32_t)(8 * lda * 2), row8_w = (uint32_t)(8 * ldw * 2);
    uint32_t voff_a = voff_a_0, voff_w = voff_w_0;          // + 128 B per K-tile of the load cursor
    const uint32_t ring_lds = lds_addr(smem) + (uint32_t)wave * 2048u;

    const int nk = K / BK;
    const int nwg = tiles_m * tiles_n;
    const int G = gridDim.x;
    int nt = nwg / G;
    {
        int mm, nn;
        if (nwg % G != 0 && pp_tile(nt, tiles_m, tiles_n, mm, nn)) ++nt;
    }
    const int total = 4 * nk * nt;

    // ---- load cursor (wave-uniform) ----
    int lm0 = 0, ln0 = 0, lt = 0, lj = 0, lx = 0;
    bool lvalid = pp_tile(0, tiles_m, tiles_n, lm0, ln0);
    bool lfull = false;
    const char *pa0 = nullptr, *pa1 = nullptr, *pw0 = nullptr, *pw1 = nullptr;
    auto enter_tile = [&]() {   // the cursor stands at K-tile 0 of tile (lm0, ln0)
        lfull = (lm0 + BM2 <= M) && (ln0 + BN2 <= N);
        pa0 = reinterpret_cast<const char*>(A) + (int64_t)(lm0 + wave * 16) * lda * 2;
        pa1 = pa0 + (int64_t)128 * lda * 2;
        pw0 = reinterpret_cast<const char*>(W) + (int64_t)(ln0 + wave * 16) * ldw * 2;
        pw1 = pw0 + (int64_t)128 * ldw * 2;
        voff_a = voff_a_0;
        voff_w = voff_w_0;
    };
    if (lvalid) enter_tile();
    // half-tile `kind` (consumption order A0, W0, W1, A1) of the cursor's K-tile
    auto issue1 = [&](auto kc, bool check_valid) {
        if ((ABL & 1) && lx >= 2) return;
        if (check_valid && !lvalid) return;
        constexpr int kind = decltype(kc)::value;
        if (__builtin_expect(lfull, 1)) {
            const uint32_t ring = ring_lds + (uint32_t)(lx & 1) * (4u * HALF_BYTES);
            constexpr bool is_a = (kind == 0) || (kind == 3);
            constexpr int half = (kind >= 2) ? 1 : 0;
            constexpr int off = ((is_a ? 0 : 2) + half) * HALF_BYTES;
            const char* base = is_a ? (half ? pa1 : pa0) : (half ? pw1 : pw0);
            const uint32_t v0 = is_a ? voff_a : voff_w;
            glds16_sbase_off<off>(base, v0, ring);
            glds16_sbase_off<off + 1024>(base, v0 + (is_a ? row8_a : row8_w), ring);
        } else {   // tile cut by the matrix edge: per-lane clamped rows
            stage_half_at(A, lda, W, ldw, lm0, ln0, M, N, lt * BK, lx & 1, kind, smem, wave, lane, voff_a_0, voff_w_0);
        }
    };
    auto issue2 = [&](auto k0c, auto k1c, bool check_valid) {
        issue1(k0c, check_valid);
        issue1(k1c, check_valid);
    };
    auto issue_bias = [&]() {
        if (wave == 0 && ep.bias && lvalid) {
            int col = ln0 + (lane & 31) * 8;
            col = col <= N - 8 ? col : N - 8;
            glds16_vaddr(ep.bias + col, lds_addr(smem + PP_BIAS_OFF + (lj & 1) * 1024));
        }
    };
    auto advance = [&]() {
        ++lx;
        voff_a += 128;
        voff_w += 128;
        if (++lt == nk) {
            lt = 0;
            ++lj;
            lvalid = lj < nt && pp_tile(lj, tiles_m, tiles_n, lm0, ln0);
            if (lvalid) enter_tile();
        }
    };
    // steady state inside a full tile (fast K-tiles, below): the cursor's tile is the compute tile, full, and does not end within the
    // next two K-tiles -> no edge / validity / wrap test, and the ring stage RING is a compile-time constant of the unrolled loop
    auto issue1_fast = [&](auto kc, auto rc) {
        if (ABL & 1) return;
        constexpr int kind = decltype(kc)::value, RING = decltype(rc)::value;
        constexpr bool is_a = (kind == 0) || (kind == 3);
        constexpr int half = (kind >= 2) ? 1 : 0;
        constexpr int off = RING * 4 * HALF_BYTES + ((is_a ? 0 : 2) + half) * HALF_BYTES;
        const char* base = is_a ? (half ? pa1 : pa0) : (half ? pw1 : pw0);
        const uint32_t v0 = is_a ? voff_a : voff_w;
        glds16_sbase_off<off>(base, v0, ring_lds);
        glds16_sbase_off<off + 1024>(base, v0 + (is_a ? row8_a : row8_w), ring_lds);
    };
    auto advance_fast = [&]() {   // the K step goes into the scalar base pointers: the lane offsets of the DMAs stay loop-invariant
        ++lx;
        ++lt;
        pa0 += 128;   // (SALU instead of VALU adds: same speed, six registers less -- profiles/r3_ppab_k_step_sgpr_vs_valu.jsonl)
        pa1 += 128;
        pw0 += 128;
        pw1 += 128;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using IDyn = std::integral_constant<int, -1>;
    // prologue: K-tile 0 (4 half-tiles) + K-tile 1 (A0, W0) = PD2 loads
    issue_bias();
    issue2(I0{}, I1{}, true);
    issue2(I2{}, I3{}, true);
    advance();
    if (lt == 0) issue_bias();
    issue2(I0{}, I1{}, true);

    typename P::Acc acc[2 * P::MI][P::NB];
    bf16x8_t af[P::MI][P::KS];
    bf16x8_t wf[P::NB][P::KS];
    int pos0 = 0;
    if (MODE == PP_QKV) pos0 = __builtin_amdgcn_readfirstlane(ep.q.pos0_dev ? *ep.q.pos0_dev : ep.q.pos0_host);

    if (grp1) {
        wait_for_phase_pp(0, total, false);
        cbarrier();
    }
    int g = 0;
    int cm0 = 0, cn0 = 0;
    bool prev_full = false;   // the previous tile's epilogue issued PP_STORES stores per wave (a full tile) ...
    bool pre_issued = false;  // ... behind the half-tiles phase (0, 0) of this tile would have issued
    // one phase.  PM: 0 = K-tiles 0 and 1 of a tile (store-count waits behind a full tile), 1 = steady state, 2 = generic (stream tail)
    // ABL bit 2048 (experiment, variant 19; rejected: 6-12 % slower, profiles/r3_ppab_cluster_tail_after_barrier.jsonl): the last four
    // MFMAs of a cluster are issued AFTER the barrier that hands the matrix pipe to the other wave group (at the top of the wave's
    // next phase), so that group starts while this one still has work queued.  The held-back MFMAs then queue behind the other
    // group's on the same pipe and delay this wave's fragment reads -- the off-turn path is the one that must stay short
    constexpr bool TAIL = (ABL & 2048) != 0;
    // ABL bit 32768 (experiment, variant 26): in fast K-tiles the second half-tile of a phase is issued inside the MFMA cluster
    constexpr bool DMAC = (ABL & 32768) != 0;
    bool tail_pending = false;
    auto run_tail = [&](auto ppc) {   // the held-back MFMAs of the previous phase (its p = ppc): i = MI - 1, ks = KS - 1, every nb
        constexpr int pp = decltype(ppc)::value;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int nb = 0; nb < P::NB; ++nb)
            acc[pp * P::MI + P::MI - 1][nb] = P::mma(wf[nb][P::KS - 1], af[P::MI - 1][P::KS - 1], acc[pp * P::MI + P::MI - 1][nb]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // ST >= 0: fast K-tile (PM == 1 only) whose ring stage is the compile-time constant ST -- LDS addresses become loop-invariant
    // registers + immediates, the loads go out unconditionally
    auto phase = [&](auto pc, auto pmc, int t, int j, auto stc) {
        constexpr int p = decltype(pc)::value, PM = decltype(pmc)::value, ST = decltype(stc)::value;
        static_assert(ST < 0 || PM == 1, "fast K-tiles are steady-state K-tiles");
        if constexpr (TL && ST < 0) {
            if (p == 0 && t == 0) stamp(j, 0);
            if (p == 0 && t == nk - 1) stamp(j, 5);
        }
        if (!grp1) {
            if constexpr (PM == 2) {
                wait_for_phase_pp(g, total, false);
            } else if constexpr (PM == 0) {
                if (prev_full) wait_vmcnt_pp<(p ? PP2_WAIT_P1 : PP2_WAIT_P0) + PP_STORES>();
                else wait_vmcnt<(p ? PP2_WAIT_P1 : PP2_WAIT_P0)>();
            } else {
                wait_vmcnt<(p ? PP2_WAIT_P1 : PP2_WAIT_P0)>();
            }
        }
        const char* buf = smem + (ST >= 0 ? ST * 4 * HALF_BYTES : (g & 2) * 2 * HALF_BYTES);
        const bool do_reads = !(ABL & 2) || g == 0;
        cbarrier();
        if constexpr (TAIL) {
            if (ST >= 0 || tail_pending) run_tail(std::integral_constant<int, 1 - p>{});
            tail_pending = true;
        }
        if (ABL & 16384) __builtin_amdgcn_s_setprio(2);   // experiment (variant 25): the off-turn (reads + DMA issue) at raised priority
        if constexpr ((ABL & 4096) != 0 && ST >= 0) {
            // experiment (variant 23): the four DMA instructions spread among the fragment reads instead of behind them
            constexpr int STc = ST >= 0 ? ST : 0;
            if constexpr (p == 0) {
#pragma unroll
                for (int nb = 0; nb < P::NB; ++nb) wf[nb][0] = P::read_w(buf + (2 + wh) * HALF_BYTES, wrow, nb, 0, lane);
                issue1_fast(I2{}, std::integral_constant<int, 1 - STc>{});
#pragma unroll
                for (int i = 0; i < P::MI; ++i) af[i][0] = P::read_a(buf, wm, i, 0, lane);
#pragma unroll
                for (int nb = 0; nb < P::NB; ++nb) wf[nb][1] = P::read_w(buf + (2 + wh) * HALF_BYTES, wrow, nb, 1, lane);
                issue1_fast(I3{}, std::integral_constant<int, 1 - STc>{});
                advance_fast();
#pragma unroll
                for (int i = 0; i < P::MI; ++i) af[i][1] = P::read_a(buf, wm, i, 1, lane);
            } else {
#pragma unroll
                for (int i = 0; i < P::MI; ++i) af[i][0] = P::read_a(buf + HALF_BYTES, wm, i, 0, lane);
                issue1_fast(I0{}, std::integral_constant<int, STc>{});
#pragma unroll
                for (int i = 0; i < P::MI; ++i) af[i][1] = P::read_a(buf + HALF_BYTES, wm, i, 1, lane);
                issue1_fast(I1{}, std::integral_constant<int, STc>{});
            }
        } else if constexpr (p == 0) {
            if constexpr (PM == 0 && !(ABL & 128)) {
                if (t == 0) P::init(acc, ep.bias != nullptr, false, lds_addr(smem + PP_BIAS_OFF + (j & 1) * 1024), wcol, lane);
            }
            if (do_reads) {
#pragma unroll
                for (int ks = 0; ks < P::KS; ++ks) {
#pragma unroll
                    for (int nb = 0; nb < P::NB; ++nb) wf[nb][ks] = P::read_w(buf + (2 + wh) * HALF_BYTES, wrow, nb, ks, lane);
#pragma unroll
                    for (int i = 0; i < P::MI; ++i) af[i][ks] = P::read_a(buf, wm, i, ks, lane);
                }
            }
            if constexpr (ST >= 0) {     // the cursor stands one K-tile ahead: the other ring stage
                issue1_fast(I2{}, std::integral_constant<int, 1 - ST>{});
                if constexpr (!DMAC) {
                    issue1_fast(I3{}, std::integral_constant<int, 1 - ST>{});
                    advance_fast();
                }
            } else if (PM == 0 && t == 0 && pre_issued) {
                pre_issued = false;          // issued at the tile boundary, ahead of the previous tile's stores
            } else {
                issue2(I2{}, I3{}, PM == 2);
                advance();
            }
        } else {
            if (!(ABL & 2)) {
#pragma unroll
                for (int ks = 0; ks < P::KS; ++ks)
#pragma unroll
                    for (int i = 0; i < P::MI; ++i) af[i][ks] = P::read_a(buf + HALF_BYTES, wm, i, ks, lane);
            }
            if constexpr (ST >= 0) {     // two K-tiles ahead: this K-tile's own stage
                issue1_fast(I0{}, std::integral_constant<int, ST>{});
                if constexpr (!DMAC) issue1_fast(I1{}, std::integral_constant<int, ST>{});
            } else {
                if (lt == 0) issue_bias();
                issue2(I0{}, I1{}, PM == 2);
            }
        }
        if (grp1) {   // group 1 waits for the data of phase g + 1 before the barrier that opens it for group 0
            if constexpr (PM == 2) {
                wait_for_phase_pp(g + 1, total, false);
            } else if constexpr (PM == 0) {
                // phase g + 1 is (0, 1), (1, 0) or (1, 1): its half-tiles are older than the previous tile's stores; (2, 0) is not
                if (prev_full && !(t == 1 && p == 1)) wait_vmcnt_pp<(p == 0 ? PP2_WAIT_P1 : PP2_WAIT_P0) + PP_STORES>();
                else wait_vmcnt<(p == 0 ? PP2_WAIT_P1 : PP2_WAIT_P0)>();
            } else {
                wait_vmcnt<(p == 0 ? PP2_WAIT_P1 : PP2_WAIT_P0) - ((DMAC && ST >= 0) ? 2 : 0)>();
            }
        }
        if (ABL & 16384) __builtin_amdgcn_s_setprio(0);
        cbarrier();
        if constexpr (TL && ST < 0) {
            if (t == 0) stamp(j, p == 0 ? 1 : 3);
            if (t == 1 && p == 0) stamp(j, 4);
            if (t == nk - 1 && p == 1) stamp(j, 6);
        }
        __builtin_amdgcn_sched_barrier(0);
        // (no s_setprio around the cluster: raising the MFMA-issuing wave's priority delays the OTHER group's fragment reads and
        //  DMA issue on the same SIMD, and that off-turn path is what bounds a phase -- +1-5 % without it on every shape and
        //  epilogue, profiles/r3_ppab_no_setprio.jsonl; ABL 8192 = the round-2 setting, 16384 = priority for the off-turn instead)
        if (ABL & 8192) __builtin_amdgcn_s_setprio(1);
        if (!(ABL & 16)) {
#pragma unroll
            for (int ks = 0; ks < P::KS; ++ks)
#pragma unroll
                for (int i = 0; i < P::MI; ++i)
#pragma unroll
                    for (int nb = 0; nb < P::NB; ++nb)
                        if (!(TAIL && ks == P::KS - 1 && i == P::MI - 1)) {
                            if constexpr (DMAC && ST >= 0) {
                                if (ks == 0 && i == 2 && nb == 0) {   // before the 9th MFMA of the cluster
                                    constexpr int STc = ST >= 0 ? ST : 0;
                                    if constexpr (p == 0) {
                                        issue1_fast(I3{}, std::integral_constant<int, 1 - STc>{});
                                        advance_fast();
                                    } else {
                                        issue1_fast(I1{}, std::integral_constant<int, STc>{});
                                    }
                                }
                            }
                            acc[p * P::MI + i][nb] = P::mma(wf[nb][ks], af[i][ks], acc[p * P::MI + i][nb]);
                        }
        } else {
#pragma unroll
            for (int ks = 0; ks < P::KS; ++ks) {
#pragma unroll
                for (int i = 0; i < P::MI; ++i) asm volatile("" ::"v"(af[i][ks]));
#pragma unroll
                for (int nb = 0; nb < P::NB; ++nb) asm volatile("" ::"v"(wf[nb][ks]));
            }
        }
        if (ABL & 8192) __builtin_amdgcn_s_setprio(0);
        if constexpr (TL && ST < 0) {
            __builtin_amdgcn_sched_barrier(0);
            if (t == 0 && p == 0) stamp(j, 2);
            if (t == nk - 1 && p == 1) stamp(j, 7);
        }
        ++g;
    };
    for (int j = 0; j < nt; ++j) {
        pp_tile(j, tiles_m, tiles_n, cm0, cn0);
        // the generic waits are only needed where the end of the load stream clips the issue count: the last two K-tiles of
        // the workgroup's last tile (nk >= 4 is a launch precondition, so K-tile 0 is never among them)
        const int t_gen = (j == nt - 1) ? nk - 2 : nk;
        phase(I0{}, I0{}, 0, j, IDyn{});
        phase(I1{}, I0{}, 0, j, IDyn{});
        phase(I0{}, I0{}, 1, j, IDyn{});
        phase(I1{}, I0{}, 1, j, IDyn{});
        int t = 2;
        if (!(ABL & 1024) && (cm0 + BM2 <= M) && (cn0 + BN2 <= N)) {
            // fast K-tiles: t <= nk - 3, i.e. everything the K-tile issues (K-tiles t + 1, t + 2 of THIS tile) lies before the tile's
            // end.  Unrolled by two so that the ring stage of a K-tile is a constant; one generic K-tile first if stage 1 is next
            if ((g & 2) && t <= nk - 3) {
                phase(I0{}, I1{}, t, j, IDyn{});
                phase(I1{}, I1{}, t, j, IDyn{});
                ++t;
            }
            for (; t + 1 <= nk - 3; t += 2) {
                phase(I0{}, I1{}, t, j, I0{});
                phase(I1{}, I1{}, t, j, I0{});
                phase(I0{}, I1{}, t + 1, j, I1{});
                phase(I1{}, I1{}, t + 1, j, I1{});
            }
        }
        for (; t < t_gen; ++t) {
            phase(I0{}, I1{}, t, j, IDyn{});
            phase(I1{}, I1{}, t, j, IDyn{});
        }
        for (; t < nk; ++t) {
            phase(I0{}, I2{}, t, j, IDyn{});
            phase(I1{}, I2{}, t, j, IDyn{});
        }
        if constexpr (TAIL) {   // the tile's last cluster is complete before the epilogue
            run_tail(I1{});
            tail_pending = false;
        }
        if (ABL & 4) {
#pragma unroll
            for (int i = 0; i < 2 * P::MI; ++i)
#pragma unroll
                for (int nb = 0; nb < P::NB; ++nb) asm volatile("" ::"v"(acc[i][nb][0]));
            continue;
        }
        prev_full = (cm0 + BM2 <= M) && (cn0 + BN2 <= N) && !(ABL & (32 | 256 | 512));
        if (MODE == PP_ARGMAX && ep.out == nullptr) prev_full = false;   // 8 partial stores per wave instead of >= PP_STORES
        if (j + 1 < nt) {
            // tile boundary: W1 / A1 of the next tile's K-tile 1 go out ahead of this tile's stores.  Their slots were last read in
            // the final phase: one barrier after those reads were issued, the margin every other refill of the ring has.
            // Round 6 (profiles/r6_pp_timeline.jsonl): the two wave groups run one barrier apart, so group 0's barrier here pairs
            // with group 1's SECOND barrier of its final phase -- and group 1's own boundary barrier used to pair with group 0's first
            // barrier of the NEXT tile, which group 0 reaches only after its epilogue; group 0 then waited at its second barrier for
            // group 1's epilogue: the two epilogues ran one after the other (per tile 2 x 3.9 k cycles plain, 2 x 8.9 k QKV, 2 x 13 k
            // GELU with the matrix pipe idle).  Group 1 needs no barrier in front of its refill -- every wave that passed group 1's
            // second barrier of the final phase has issued its final reads, and group 0's were issued a barrier earlier -- so its
            // extra barrier (the count must match group 0's: the stagger) moves BEHIND its epilogue: both groups now run their
            // epilogues at the same time, group 1's starting when its last MFMA cluster is issued.
            if (!grp1 || SERIAL_EPI) cbarrier();
            issue2(I2{}, I3{}, false);
            advance();
            pre_issued = true;
        } else {
            prev_full = false;
        }
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        stamp(j, 8);
        P::template epilogue<MODE, ((ABL >> 5) & 3) | ((ABL >> 6) & 12) | ((ABL >> 12) & 16) | (LNF ? 32 : 0) | ((ABL >> 12) & (64 | 128)) | ((ABL >> 9) & 256) | ((ABL >> 13) & 512) | ((ABL >> 13) & 1024)>(
            ep, pos0, stage, acc, cm0, cn0 + wcol, wm, lane_e, (TL && j < TL_TILES) ? tl_lds + (uint32_t)(j * TL_EV) * 8u : 0u);
        stamp(j, 9);
        if (j + 1 < nt && grp1 && !SERIAL_EPI) cbarrier();
    }
    if (!grp1) cbarrier();
    if constexpr (TL) {   // every wave dumps its own 1 KiB of stamps (it wrote them itself: no barrier needed)
        {
            uint64_t tm, rt;
            asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm), "=s"(rt) : : "memory");   // (stores acknowledged)
            asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:16" : : "v"(tl_lds + 118u * 8u), "v"(tm), "v"(rt) : "memory");
        }
        if ((blockIdx.x < 4 || blockIdx.x >= gridDim.x - 4) && ep.amax != nullptr && MODE != PP_ARGMAX) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int slot = blockIdx.x < 4 ? (int)blockIdx.x : 4 + (int)(blockIdx.x - (gridDim.x - 4));   // first four and last four workgroups
            uint64_t* dst = reinterpret_cast<uint64_t*>(ep.amax) + ((int64_t)slot * 8 + wave) * 128;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int idx = lane + h * 64;
                uint64_t v;
                const uint32_t a = tl_lds + (uint32_t)idx * 8u;
                asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
                if (idx < TL_EV * TL_TILES + 4) dst[idx] = v;
            }
        }
    }
}

// ---- persistent kernel, stream-K items (round 5): the decode-shape GEMMs whose 256 x 256 tiles do not fill whole rounds ------------
// The persistent kernel above gives a workgroup WHOLE tiles.  At M = 4096 that quantises badly exactly where the decode step
// spends its GEMM time: the QKV projection is 384 tiles = 1.5 rounds (2 tile-times for 1.5 of work), and the two
// projections onto the residual stream (N = 2048: 128 tiles) cannot use it at all -- they run on the one-tile 256 x 128 kernel
// (16 MFMAs per phase instead of 32: 0.88 / 1.13 PF) as two launches.  Here the unit of work is an ITEM = (output tile, nk K-tiles
// of its K range): a tile is C items ("chunks"), the T * C items are dealt out tile-major in equal contiguous runs of q = T * C / G
// to the G resident workgroups (classic stream-K), and consecutive items of one tile CHAIN (the accumulators carry over: no
// epilogue, no init between them).  A run that starts or ends inside a tile shares that tile with the neighbouring workgroup:
//   * the workgroup holding the tile's FIRST chunks ends its chain early -> it PUBLISHES the fp32 accumulators (bias included) to a
//     slab and raises a flag;  the workgroup holding the LAST chunks starts from zero, CONSUMES the slab (adds it) and runs the
//     normal fused epilogue (bias / residual / RoPE + append) -- one tile is never shared by more than two workgroups (host check);
//   * a workgroup walks its chains in REVERSE tile order: its publish chain (the tail of its run) FIRST, its consume chain (the head)
//     LAST, so in a 1.5-round problem the partials are published one item-time into the launch and consumed at its very end: the
//     hand-off is off the critical path (the consumer never waits);
//   * hand-off per WAVE, no workgroup barrier (the two wave groups run half a phase apart; a barrier "for everyone" does not exist
//     in this kernel): wave w of the consumer owns exactly the accumulator fragments wave w of the producer owns.  Producer wave:
//     32 write-through (sc1) 16-byte stores per lane = its 128 x 64 fp32 block, s_waitcnt vmcnt(0), lane 0 stores the wave's flag
//     (relaxed, agent scope).  Consumer wave: lane 0 polls that one word (relaxed, s_sleep, bounded), one agent-scope acquire,
//     plain loads, add, flag back to zero.  (CDNA guide, Guideline 16 recipe R1.)  Virtual workgroup ids are XCD-contiguous, so a
//     pair shares an L2 -- a speed bonus only.
// A second operand pair (bt_linear2: attention.dense + mlp.dense_4h_to_h as ONE GEMM over K = H + I) is just a different base
// pointer for the items whose chunk lies in the second K range: nothing of it lives in the K-loop (the first generation's two-pair
// K-loop spilled scalars there and measured 4-7 % slower).
// The K-loop, ring, waits and tile-boundary protocol are gemm_bf16_mfma256pp2_kernel's, item for tile (every item has nk K-tiles;
// full tiles only: M % 256 == N % 256 == 0); an item that ends without stores leaves prev_full = false, a publish drains its
// stores itself.  tests/test_gemm_ring_schedule.py runs the v2 hazard model over such item sequences.
struct SkArgs {
    const bf16* A2;      // second operand pair (items whose chunk starts at K-tile >= nk1), or unused when nk1 covers all of K
    int64_t lda2;
    const bf16* W2;
    int64_t ldw2;
    const bf16* bias2;   // second bias (the accumulators of a tile's first chunk start at bias + bias2)
    int nk1;             // K-tiles of the first pair (a multiple of nk)
    int C;               // chunks per output tile
    int q;               // items per workgroup
    int nk;              // K-tiles per item (>= 4)
    float* slabs;        // [G][8 waves][32][64 lanes][4] fp32: slab vw = partial tile of virtual workgroup vw's consume chain
    uint32_t* flags;     // [G][8] one word per (consumer workgroup, wave), zero at launch and left zero; word [8 G] = STICKY time-out word
                         // (SK_TIMEOUT_MAGIC once any consumer gave up; cleared by sk_reset only, never by the per-launch sk_clear_flags)
    int dbg_drop_vw;     // debug build: the publishing workgroup with this virtual id never raises its flags (-1: none) -> its consumer
    uint32_t dbg_spin;   //              times out after dbg_spin polls (0: SK_SPIN_LIMIT); the product build ignores both
    uint32_t* dbg_diag;  // debug build (bt_debug_sk_diag_buffer): [0] = count, then 8 words per recorded time-out; never cleared by the library
};
constexpr int SK_SLAB_FLOATS = 8 * 32 * 64 * 4;          // 256 KiB per workgroup
constexpr uint32_t SK_SPIN_LIMIT = 1u << 20;              // polls (x ~1 us) before a consumer gives up and flags the launch
constexpr uint32_t SK_TIMEOUT_MAGIC = 0x534b544fu;        // "SKTO": value of the sticky word after a time-out

// flag words of the hand-off (wave-uniform addresses): relaxed agent-scope accesses = `sc1` loads / stores (CDNA guide, visibility
// table).  Inline asm with its own zero offset register: the compiler's form kept that zero in a VGPR across the K-loop, spilled it,
// and reloaded it from scratch (+ vmcnt(0)) on every poll.
__device__ inline uint32_t sk_flag_load(const uint32_t* p) {      // every lane reads the same word
    uint32_t v;
    // s_nop 4 first: the pointer may sit in SGPRs the compiler has just written with v_readlane / v_readfirstlane (a spill reload),
    // and a VALU-written SGPR needs 5 wait states before a VMEM instruction reads it as its base -- hipcc pads that hazard for its
    // own instructions, never inside an asm string (CDNA guide 5.7 item 2).  Without it the load went to (stale high half : low
    // half) -- a memory access fault on every launch with a hand-off (profiles/r5_sk_bisect.md).
    asm volatile("s_nop 4\n\tv_mov_b32 %0, 0\n\tglobal_load_dword %0, %0, %1 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "s"(p) : "memory");
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ inline void sk_flag_store(uint32_t* p, uint32_t value) {   // every lane stores the same value to the same word
    uint32_t z, d;
    asm volatile("s_nop 4\n\tv_mov_b32 %0, 0\n\tv_mov_b32 %1, %3\n\tglobal_store_dword %0, %1, %2 sc1\n\ts_nop 1"
                 : "=&v"(z), "=&v"(d) : "s"(p), "s"(value) : "memory");
}

struct SkWalk {          // wave-uniform: the items of this workgroup's run [lo, lo + q), chains in reverse tile order
    int tile, c, cs, ce; // current tile, current chunk, first / end chunk of the current chain
    int m0, n0;
    __device__ inline void coords(int tiles_m, int tiles_n) {
        const int group_sz = GROUP_M2 * tiles_n;
        const int g = tile / group_sz;
        const int first_m = g * GROUP_M2;
        const int gm = min(tiles_m - first_m, GROUP_M2);
        const int in_g = tile - g * group_sz;
        m0 = (first_m + in_g % gm) * BM2;
        n0 = (in_g / gm) * BN2;
    }
    __device__ inline void start(int lo, int hi, int C, int tiles_m, int tiles_n) {
        tile = (hi - 1) / C;
        cs = max(lo - tile * C, 0);
        ce = hi - tile * C;
        c = cs;
        coords(tiles_m, tiles_n);
    }
    __device__ inline void next(int lo, int C, int tiles_m, int tiles_n) {   // (the caller counts items: never called past the run)
        if (++c == ce) {
            --tile;
            ce = C;
            cs = max(lo - tile * C, 0);
            c = cs;
            coords(tiles_m, tiles_n);
        }
    }
    __device__ inline bool chain_first() const { return c == cs; }
    __device__ inline bool chain_last() const { return c == ce - 1; }
};

template <int MODE, bool SERIAL_EPI = false>   // SERIAL_EPI: the round-5 item boundary (debug-build A/B reference, bt_debug_serial_epilogues)
__global__ __launch_bounds__(512) void gemm_bf16_mfma256sk_kernel(const bf16* __restrict__ A, int64_t lda,
                                                                 const bf16* __restrict__ W, int64_t ldw, const EpiArgs ep,
                                                                 int tiles_m, int tiles_n, const SkArgs sk) {
    typedef Mma16 P;
    __shared__ __attribute__((aligned(16))) char smem[PP_SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int wh = wn >> 1;
    const int wrow = (wn & 1) * 64;
    const int wcol = wh * 128 + wrow;
    const bool grp1 = wave >= 4;
    char* stage = smem + PP_EPI_OFF + wave * P::PATCH_BYTES;
    uint32_t voff_a = 0, voff_w = 0, row8_a = 0, row8_w = 0;   // of the load cursor's item (its operand pair's row pitch)
    const uint32_t ring_lds = lds_addr(smem) + (uint32_t)wave * 2048u;

    const int nk = sk.nk, nt = sk.q, C = sk.C;
    const int total = 4 * nk * nt;
    const int G = gridDim.x;
    const int vw = (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3);   // XCD-contiguous virtual id (G % 8 == 0)
    const int lo = vw * nt;

    // ---- load cursor (wave-uniform) ----
    SkWalk lw;
    lw.start(lo, lo + nt, C, tiles_m, tiles_n);
    int lt = 0, lj = 0, lx = 0;
    bool lvalid = true, lbias = false;
    const char *pa0 = nullptr, *pa1 = nullptr, *pw0 = nullptr, *pw1 = nullptr;
    auto enter_item = [&]() {   // the cursor stands at K-tile 0 of item lw
        const int kt0 = lw.c * nk;
        const bool second = kt0 >= sk.nk1;
        const bf16* Ap = second ? sk.A2 : A;
        const bf16* Wp = second ? sk.W2 : W;
        const int64_t la = second ? sk.lda2 : lda, lwd = second ? sk.ldw2 : ldw;
        const int64_t kb = (int64_t)(second ? kt0 - sk.nk1 : kt0) * (BK * 2);
        pa0 = reinterpret_cast<const char*>(Ap) + (int64_t)(lw.m0 + wave * 16) * la * 2 + kb;
        pa1 = pa0 + (int64_t)128 * la * 2;
        pw0 = reinterpret_cast<const char*>(Wp) + (int64_t)(lw.n0 + wave * 16) * lwd * 2 + kb;
        pw1 = pw0 + (int64_t)128 * lwd * 2;
        // lane part of a DMA address, recomputed from the lane id per item (volatile asm: a value the compiler could hoist would sit
        // in two VGPRs through the K-loop, which has none to spare -- it spilled them to scratch, and a scratch reload is a VMEM op
        // the compiler fences with vmcnt(0), i.e. a drain of the whole DMA ring)
        uint32_t ln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        const uint32_t lane_slot = ((ln & 7u) ^ ((ln >> 3) & 7u)) * 16u;
        const uint32_t lane_row = ln >> 3;
        voff_a = lane_row * (uint32_t)(la * 2) + lane_slot;
        voff_w = lane_row * (uint32_t)(lwd * 2) + lane_slot;
        row8_a = (uint32_t)(8 * la * 2);
        row8_w = (uint32_t)(8 * lwd * 2);
        lbias = lw.chain_first() && lw.c == 0 && ep.bias != nullptr;
    };
    enter_item();
    auto issue1 = [&](auto kc, bool check_valid) {
        if (check_valid && !lvalid) return;
        constexpr int kind = decltype(kc)::value;
        const uint32_t ring = ring_lds + (uint32_t)(lx & 1) * (4u * HALF_BYTES);
        constexpr bool is_a = (kind == 0) || (kind == 3);
        constexpr int half = (kind >= 2) ? 1 : 0;
        constexpr int off = ((is_a ? 0 : 2) + half) * HALF_BYTES;
        const char* base = is_a ? (half ? pa1 : pa0) : (half ? pw1 : pw0);
        const uint32_t v0 = is_a ? voff_a : voff_w;
        glds16_sbase_off<off>(base, v0, ring);
        glds16_sbase_off<off + 1024>(base, v0 + (is_a ? row8_a : row8_w), ring);
    };
    auto issue2 = [&](auto k0c, auto k1c, bool check_valid) {
        issue1(k0c, check_valid);
        issue1(k1c, check_valid);
    };
    // wave 0: the tile's bias row(s) -> LDS row (lj & 1), only for an item that starts its tile (chunk 0): lanes 0..31 fetch
    // bias[n0 .. n0 + 255], lanes 32..63 the second pair's bias (or the same bytes again)
    const bf16* bias_hi = sk.bias2 ? sk.bias2 : ep.bias;
    auto issue_bias = [&]() {
        if (wave == 0 && lbias && lvalid)
            glds16_vaddr((lane < 32 ? ep.bias : bias_hi) + lw.n0 + (lane & 31) * 8, lds_addr(smem + PP_BIAS_OFF + (lj & 1) * 1024));
    };
    auto advance = [&]() {
        ++lx;
        voff_a += 128;
        voff_w += 128;
        if (++lt == nk) {
            lt = 0;
            ++lj;
            lvalid = lj < nt;
            if (lvalid) {
                lw.next(lo, C, tiles_m, tiles_n);
                enter_item();
            }
        }
    };
    auto issue1_fast = [&](auto kc, auto rc) {
        constexpr int kind = decltype(kc)::value, RING = decltype(rc)::value;
        constexpr bool is_a = (kind == 0) || (kind == 3);
        constexpr int half = (kind >= 2) ? 1 : 0;
        constexpr int off = RING * 4 * HALF_BYTES + ((is_a ? 0 : 2) + half) * HALF_BYTES;
        const char* base = is_a ? (half ? pa1 : pa0) : (half ? pw1 : pw0);
        const uint32_t v0 = is_a ? voff_a : voff_w;
        glds16_sbase_off<off>(base, v0, ring_lds);
        glds16_sbase_off<off + 1024>(base, v0 + (is_a ? row8_a : row8_w), ring_lds);
    };
    auto advance_fast = [&]() {
        ++lx;
        ++lt;
        pa0 += 128;
        pa1 += 128;
        pw0 += 128;
        pw1 += 128;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using IDyn = std::integral_constant<int, -1>;
    // prologue: K-tile 0 (4 half-tiles) + K-tile 1 (A0, W0) = PD2 loads  (nk >= 4: K-tile 1 belongs to the same item)
    issue_bias();
    issue2(I0{}, I1{}, true);
    issue2(I2{}, I3{}, true);
    advance();
    issue2(I0{}, I1{}, true);

    typename P::Acc acc[2 * P::MI][P::NB];
    bf16x8_t af[P::MI][P::KS];
    bf16x8_t wf[P::NB][P::KS];
    int pos0 = 0;
    if (MODE == PP_QKV) pos0 = __builtin_amdgcn_readfirstlane(ep.q.pos0_dev ? *ep.q.pos0_dev : ep.q.pos0_host);

    if (grp1) {
        wait_for_phase_pp(0, total, false);
        cbarrier();
    }
    int g = 0;
    bool prev_full = false;
    bool pre_issued = false;
    // ---- compute cursor ----
    SkWalk cw;
    cw.start(lo, lo + nt, C, tiles_m, tiles_n);
    bool c_init = true, c_bias = false;      // of the compute cursor's item: start a chain (accumulators <- bias rows, or zero)
    auto phase = [&](auto pc, auto pmc, int t, int j, auto stc) {
        constexpr int p = decltype(pc)::value, PM = decltype(pmc)::value, ST = decltype(stc)::value;
        static_assert(ST < 0 || PM == 1, "fast K-tiles are steady-state K-tiles");
        if (!grp1) {
            if constexpr (PM == 2) {
                wait_for_phase_pp(g, total, false);
            } else if constexpr (PM == 0) {
                if (prev_full) wait_vmcnt_pp<(p ? PP2_WAIT_P1 : PP2_WAIT_P0) + PP_STORES>();
                else wait_vmcnt<(p ? PP2_WAIT_P1 : PP2_WAIT_P0)>();
            } else {
                wait_vmcnt<(p ? PP2_WAIT_P1 : PP2_WAIT_P0)>();
            }
        }
        const char* buf = smem + (ST >= 0 ? ST * 4 * HALF_BYTES : (g & 2) * 2 * HALF_BYTES);
        cbarrier();
        if constexpr (p == 0) {
            if constexpr (PM == 0) {
                if (t == 0 && c_init)
                    P::init(acc, c_bias, c_bias && sk.bias2 != nullptr, lds_addr(smem + PP_BIAS_OFF + (j & 1) * 1024), wcol, lane);
            }
#pragma unroll
            for (int ks = 0; ks < P::KS; ++ks) {
#pragma unroll
                for (int nb = 0; nb < P::NB; ++nb) wf[nb][ks] = P::read_w(buf + (2 + wh) * HALF_BYTES, wrow, nb, ks, lane);
#pragma unroll
                for (int i = 0; i < P::MI; ++i) af[i][ks] = P::read_a(buf, wm, i, ks, lane);
            }
            if constexpr (ST >= 0) {     // the cursor stands one K-tile ahead: the other ring stage
                issue1_fast(I2{}, std::integral_constant<int, 1 - ST>{});
                issue1_fast(I3{}, std::integral_constant<int, 1 - ST>{});
                advance_fast();
            } else if (PM == 0 && t == 0 && pre_issued) {
                pre_issued = false;          // issued at the item boundary
            } else {
                issue2(I2{}, I3{}, PM == 2);
                advance();
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < P::KS; ++ks)
#pragma unroll
                for (int i = 0; i < P::MI; ++i) af[i][ks] = P::read_a(buf + HALF_BYTES, wm, i, ks, lane);
            if constexpr (ST >= 0) {     // two K-tiles ahead: this K-tile's own stage
                issue1_fast(I0{}, std::integral_constant<int, ST>{});
                issue1_fast(I1{}, std::integral_constant<int, ST>{});
            } else {
                if (lt == 0) issue_bias();
                issue2(I0{}, I1{}, PM == 2);
            }
        }
        if (grp1) {   // group 1 waits for the data of phase g + 1 before the barrier that opens it for group 0
            if constexpr (PM == 2) {
                wait_for_phase_pp(g + 1, total, false);
            } else if constexpr (PM == 0) {
                if (prev_full && !(t == 1 && p == 1)) wait_vmcnt_pp<(p == 0 ? PP2_WAIT_P1 : PP2_WAIT_P0) + PP_STORES>();
                else wait_vmcnt<(p == 0 ? PP2_WAIT_P1 : PP2_WAIT_P0)>();
            } else {
                wait_vmcnt<(p == 0 ? PP2_WAIT_P1 : PP2_WAIT_P0)>();
            }
        }
        cbarrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < P::KS; ++ks)
#pragma unroll
            for (int i = 0; i < P::MI; ++i)
#pragma unroll
                for (int nb = 0; nb < P::NB; ++nb) acc[p * P::MI + i][nb] = P::mma(wf[nb][ks], af[i][ks], acc[p * P::MI + i][nb]);
        ++g;
    };
    for (int j = 0; j < nt; ++j) {
        c_init = cw.chain_first();
        c_bias = c_init && cw.c == 0 && ep.bias != nullptr;
        const int t_gen = (j == nt - 1) ? nk - 2 : nk;
        phase(I0{}, I0{}, 0, j, IDyn{});
        phase(I1{}, I0{}, 0, j, IDyn{});
        phase(I0{}, I0{}, 1, j, IDyn{});
        phase(I1{}, I0{}, 1, j, IDyn{});
        int t = 2;
        {
            if ((g & 2) && t <= nk - 3) {
                phase(I0{}, I1{}, t, j, IDyn{});
                phase(I1{}, I1{}, t, j, IDyn{});
                ++t;
            }
            for (; t + 1 <= nk - 3; t += 2) {
                phase(I0{}, I1{}, t, j, I0{});
                phase(I1{}, I1{}, t, j, I0{});
                phase(I0{}, I1{}, t + 1, j, I1{});
                phase(I1{}, I1{}, t + 1, j, I1{});
            }
        }
        for (; t < t_gen; ++t) {
            phase(I0{}, I1{}, t, j, IDyn{});
            phase(I1{}, I1{}, t, j, IDyn{});
        }
        for (; t < nk; ++t) {
            phase(I0{}, I2{}, t, j, IDyn{});
            phase(I1{}, I2{}, t, j, IDyn{});
        }
        const bool last = cw.chain_last();
        const bool publish = last && cw.ce != C;          // the chain ends inside its tile: the neighbour finishes the tile
        const bool consume = last && !publish && cw.cs != 0;   // the chain began inside its tile: add the neighbour's partial first
        prev_full = last && !publish;                      // a normal epilogue follows: >= PP_STORES stores per wave
        if (MODE == PP_ARGMAX && ep.out == nullptr) prev_full = false;
        const bool more = j + 1 < nt;
        if (more) {
            // item boundary: W1 / A1 of the next item's K-tile 1 go out ahead of whatever this item stores (see the v2 kernel).  Round 6:
            // group 1's extra barrier sits BEHIND whatever it does at the boundary (epilogue, publish, consume), so that the two wave
            // groups do it at the same time instead of one after the other -- see the v2 kernel's tile boundary
            if (!grp1 || SERIAL_EPI) cbarrier();
            issue2(I2{}, I3{}, false);
            advance();
            pre_issued = true;
        } else {
            prev_full = false;
        }
        const int cm0 = cw.m0, cn0 = cw.n0;
        if (more) cw.next(lo, C, tiles_m, tiles_n);
        if (!last) {                                       // the next item continues this tile in the same accumulators
            if (more && grp1 && !SERIAL_EPI) cbarrier();
            continue;
        }
        if (publish) {
            // this wave's 128 x 64 fp32 block -> slab of the consumer (the next virtual workgroup), write-through, then its flag
            float* slab = sk.slabs + ((int64_t)(vw + 1) * 8 + wave) * (32 * 64 * 4);
            const __amdgpu_buffer_rsrc_t srs = tile_rsrc(reinterpret_cast<const char*>(slab));
            const uint32_t svoff = (uint32_t)lane * 16u;
#pragma unroll
            for (int i = 0; i < 2 * P::MI; ++i)
#pragma unroll
                for (int nb = 0; nb < P::NB; ++nb) {
                    const f32x4_t v = acc[i][nb];
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]),
                                                                   __float_as_uint(v[3])},
                                                           srs, svoff, (uint32_t)((i * P::NB + nb) * 1024), 16 /* sc1: write-through */);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every store of this wave has left the CU (and the loads issued before them)
#ifdef BT_DEBUG_HOOKS
            if (vw != sk.dbg_drop_vw)                          // test hook: a producer that never arrives
#endif
            sk_flag_store(sk.flags + (int64_t)(vw + 1) * 8 + wave, 1u);
            if (more && grp1 && !SERIAL_EPI) cbarrier();
            continue;
        }
        if (consume) {
            uint32_t* flag = sk.flags + (int64_t)vw * 8 + wave;
            uint32_t seen = 0;
#ifdef BT_DEBUG_HOOKS
            const uint32_t spin_limit = sk.dbg_spin ? sk.dbg_spin : SK_SPIN_LIMIT;
#else
            constexpr uint32_t spin_limit = SK_SPIN_LIMIT;
#endif
            for (uint32_t spins = 0; spins < spin_limit; ++spins) {
                seen = sk_flag_load(flag);
                if (seen != 0) break;
                __builtin_amdgcn_s_sleep(8);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (seen == 0) {
                // The producer never arrived.  All G workgroups are resident when the launch has the GPU to itself, so this takes
                // another process's kernels, a profiler or a hung partner.  The slab is NOT added (its bytes are undefined): the tile
                // is left incomplete and the launch is flagged through the sticky word, which only sk_reset clears; bt_decoder_forward /
                // bt_stream_k_status turn it into BT_DEV_ERR_HANDOFF_TIMEOUT and the host discards the pass.  The flag word stays as
                // it is -- a producer that arrives late leaves it at 1, and whoever runs next on this scratch starts from sk_reset.
                sk_flag_store(sk.flags + (int64_t)G * 8, SK_TIMEOUT_MAGIC);
                sk_flag_store(sk.flags + (int64_t)G * 8 + 1, (uint32_t)vw);            // who gave up (last writer wins): diagnostics only
                sk_flag_store(sk.flags + (int64_t)G * 8 + 2, (uint32_t)wave | ((uint32_t)MODE << 8) | ((uint32_t)nt << 16));
                sk_flag_store(sk.flags + (int64_t)G * 8 + 3, (uint32_t)tiles_m | ((uint32_t)tiles_n << 16));
#ifdef BT_DEBUG_HOOKS
                if (sk.dbg_diag != nullptr && lane == 0) {
                    const uint32_t slot = atomicAdd(sk.dbg_diag, 1u);
                    if (slot < 64) {
                        uint32_t* d = sk.dbg_diag + 8 + slot * 8;
                        d[0] = (uint32_t)vw; d[1] = (uint32_t)wave; d[2] = (uint32_t)MODE; d[3] = (uint32_t)nt;
                        d[4] = (uint32_t)ep.M; d[5] = (uint32_t)ep.N; d[6] = (uint32_t)sk.nk; d[7] = (uint32_t)sk.C;
                    }
                }
#endif
            } else {
            const float* slab = sk.slabs + ((int64_t)vw * 8 + wave) * (32 * 64 * 4);
            const __amdgpu_buffer_rsrc_t srs = tile_rsrc(reinterpret_cast<const char*>(slab));
            uint32_t svoff = (uint32_t)lane * 16u;
            // two batches of 16 loads (64 transient registers = the K-loop's dead fragment registers); the scheduling barriers keep
            // the compiler from hoisting all 32 loads in front of the first add (128 transient registers: it spilled)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x4_t pv[P::MI][P::NB];
#pragma unroll
                for (int i = 0; i < P::MI; ++i)
#pragma unroll
                    for (int nb = 0; nb < P::NB; ++nb)
                        pv[i][nb] = __builtin_amdgcn_raw_buffer_load_b128(srs, svoff, (uint32_t)(((h * P::MI + i) * P::NB + nb) * 1024), 0);
#pragma unroll
                for (int i = 0; i < P::MI; ++i)
#pragma unroll
                    for (int nb = 0; nb < P::NB; ++nb) {
                        acc[h * P::MI + i][nb][0] += __uint_as_float(pv[i][nb].x);
                        acc[h * P::MI + i][nb][1] += __uint_as_float(pv[i][nb].y);
                        acc[h * P::MI + i][nb][2] += __uint_as_float(pv[i][nb].z);
                        acc[h * P::MI + i][nb][3] += __uint_as_float(pv[i][nb].w);
                    }
                // the second batch's address "depends" on a sum of the first: its loads cannot be issued ahead of the first adds
                asm volatile("" : "+v"(svoff) : "v"(acc[h * P::MI + P::MI - 1][P::NB - 1][3]) : "memory");
            }
            sk_flag_store(flag, 0u);   // ready for the next launch
            }
        }
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        P::template epilogue<MODE>(ep, pos0, stage, acc, cm0, cn0 + wcol, wm, lane_e);
        if (more && grp1 && !SERIAL_EPI) cbarrier();
    }
    if (!grp1) cbarrier();
}

// ---- 256 x 128 tiles, one per workgroup: the decode projections onto the residual stream at M = 4096 (N = 2048) ----------------------
// attention.dense (K = 2048) and mlp.dense_4h_to_h (K = 8192) of a 4096-row decode step are 16 x 8 = 128 tiles of 256 x 256 -- half
// of the chip for the persistent kernel -- and ran on the 4-wave 128 x 128 kernel at 0.78 / 0.96 PF (0.81-0.85 of the vendor
// library, profiles/r3_vendor_hipblaslt.jsonl).  This kernel gives every CU exactly one 256 x 128 tile of such a problem with the
// persistent kernel's machinery cut down to what one tile needs:
//   * K-tile = three 16 KiB half-tiles [A rows 0-127 | A rows 128-255 | W rows 0-127], ring of two K-tiles (96 KiB);
//   * 8 waves = two staggered groups; a wave owns 64 rows of each A half x 32 columns (64 accumulator registers);
//     phase 0 reads A0 + W and runs 16 MFMAs, phase 1 reads A1 and runs 16;
//   * half-tile s (consumption order A0, W, A1) is issued 5 ahead: A1 of K-tile t+1 in phase (t, 0), A0 and W of t+2 in (t, 1); three
//     half-tiles = 6 DMA instructions stay in flight at every wait of the steady state (vmcnt(6)), the last two K-tiles
//     drain (vmcnt(0));  RAW / WAR as in the persistent kernel (every wave waits for its share before the barrier that opens
//     the reading phase; a slot is refilled one barrier after its last read was issued): tests/test_gemm_ring_schedule.py;
//   * epilogue after the loop (nothing is in flight, the ring is free): bias in fp32, fragment -> wave-private LDS patch -> 16
//     bytes per lane along rows, residual joined in the store layout, descriptor-addressed loads / stores.
// Requires M % 256 == 0, N % 128 == 0, K % 64 == 0, K >= 256 (full tiles only: no clamping code).
constexpr int T21_STAGE = 3 * HALF_BYTES;          // one K-tile of the ring
constexpr int T21_SMEM = 2 * T21_STAGE;
constexpr int T21_PATCH_PITCH = 80;                // 32 columns x 2 B + 16 pad
constexpr int T21_WAIT = 6;

// SAFE: a ring slot is refilled by LDS-DMA one barrier after its last fragment read was ISSUED; group 1's reads of a slot
// are issued right before the barrier after which group 0 issues that slot's refill, and nothing but latency (a DMA's L2 round
// trip >> a queued ds_read) orders them.  With SAFE group 1 drains its LDS queue (s_waitcnt lgkmcnt(0)) before those two barriers
// -- the MFMAs behind them need the fragments anyway, so the wait only moves in front of the barrier.  Cost: profiles/r4_t21_safe_ab.jsonl.
template <bool RES, bool PRIO = true, bool SAFE = true>
__global__ __launch_bounds__(512) void gemm_bf16_mfma256x128_kernel(const bf16* __restrict__ A, int64_t lda,
                                                                   const bf16* __restrict__ W, int64_t ldw, const EpiArgs ep,
                                                                   int K, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) char smem[T21_SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const bool grp1 = wave >= 4;
    int tm, tn;
    {
        const int nwg = tiles_m * tiles_n;
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        const int group_sz = GROUP_M2 * tiles_n;
        const int g = id / group_sz;
        const int first_m = g * GROUP_M2;
        const int gm = min(tiles_m - first_m, GROUP_M2);
        const int in_g = id - g * group_sz;
        tm = first_m + in_g % gm;
        tn = in_g / gm;
    }
    const int m0 = tm * 256, n0 = tn * 128;
    const uint32_t lane_slot = (uint32_t)(((lane & 7) ^ ((lane >> 3) & 7)) * 16);
    uint32_t voff_a = (uint32_t)(lane >> 3) * (uint32_t)(lda * 2) + lane_slot;
    uint32_t voff_w = (uint32_t)(lane >> 3) * (uint32_t)(ldw * 2) + lane_slot;
    const uint32_t row8_a = (uint32_t)(8 * lda * 2), row8_w = (uint32_t)(8 * ldw * 2);
    const char* pa0 = reinterpret_cast<const char*>(A) + (int64_t)(m0 + wave * 16) * lda * 2;
    const char* pa1 = pa0 + (int64_t)128 * lda * 2;
    const char* pw = reinterpret_cast<const char*>(W) + (int64_t)(n0 + wave * 16) * ldw * 2;
    const uint32_t ring_lds = lds_addr(smem) + (uint32_t)wave * 2048u;
    const int nk = K / BK;
    // load cursor: K-tile lt, ring buffer lt & 1; the lane offsets advance with it
    int lt = 0;
    auto issue_a = [&](auto hc) {
        constexpr int half = decltype(hc)::value;
        const uint32_t ring = ring_lds + (uint32_t)(lt & 1) * (uint32_t)T21_STAGE;
        glds16_sbase_off<half * HALF_BYTES>(half ? pa1 : pa0, voff_a, ring);
        glds16_sbase_off<half * HALF_BYTES + 1024>(half ? pa1 : pa0, voff_a + row8_a, ring);
    };
    auto issue_w = [&]() {
        const uint32_t ring = ring_lds + (uint32_t)(lt & 1) * (uint32_t)T21_STAGE;
        glds16_sbase_off<2 * HALF_BYTES>(pw, voff_w, ring);
        glds16_sbase_off<2 * HALF_BYTES + 1024>(pw, voff_w + row8_w, ring);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // prologue: K-tile 0 (A0, W, A1) and A0, W of K-tile 1 -- in consumption order
    issue_a(I0{});
    issue_w();
    issue_a(I1{});
    ++lt;
    voff_a += 128;
    voff_w += 128;
    if (nk > 1) {
        issue_a(I0{});
        issue_w();
    }
    // (the cursor now stands at K-tile 1 with A1 outstanding)

    f32x4_t acc[8][2];   // [A half * 4 + m-fragment][n-fragment]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t af[4][2], wf[2][2];

    auto wait = [&](bool steady) {
        if (steady) wait_vmcnt<T21_WAIT>();
        else wait_vmcnt<0>();
    };
    if (grp1) {
        wait(nk >= 3);
        cbarrier();
    }
    for (int t = 0; t < nk; ++t) {
        const bool steady = t + 3 <= nk;            // K-tiles 0 .. nk-3: three younger half-tiles are in flight at every wait
        const char* buf = smem + (t & 1) * T21_STAGE;
        // ---- phase 0: A0, W ----
        if (!grp1) wait(steady);
        cbarrier();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) wf[nb][ks] = read_frag(buf + 2 * HALF_BYTES, wn * 32 + nb * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i][ks] = read_frag(buf, wm * 64 + i * 16, ks, lane);
        }
        if (t + 1 < nk) issue_a(I1{});              // A1 of K-tile t + 1 (cursor at t + 1)
        if (grp1) wait(steady);
        if (SAFE && grp1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // A0 / W reads done before group 0 may refill them
        cbarrier();
        __builtin_amdgcn_sched_barrier(0);
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[i][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nb][ks], af[i][ks], acc[i][nb], 0, 0, 0);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        // ---- phase 1: A1 ----
        if (!grp1) wait(steady);
        cbarrier();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i][ks] = read_frag(buf + HALF_BYTES, wm * 64 + i * 16, ks, lane);
        ++lt;                                         // cursor -> K-tile t + 2
        voff_a += 128;
        voff_w += 128;
        if (t + 2 < nk) {
            issue_a(I0{});
            issue_w();
        }
        if (grp1) wait(t + 4 <= nk);                // (its wait for phase 0 of K-tile t + 1: steady while t + 1 <= nk - 3)
        if (SAFE && grp1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // A1 reads done before group 0 may refill the slot
        cbarrier();
        __builtin_amdgcn_sched_barrier(0);
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[4 + i][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nb][ks], af[i][ks], acc[4 + i][nb], 0, 0, 0);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    }
    if (!grp1) cbarrier();
    // ---- epilogue ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                  // every wave is past its last fragment read: the ring is free
    char* patch = smem + wave * (16 * T21_PATCH_PITCH);
    const int n_wave = n0 + wn * 32;
    float bias_v[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (ep.bias) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const uint2 b2 = *reinterpret_cast<const uint2*>(ep.bias + n_wave + nb * 16 + (lane >> 4) * 4);
            bias_v[nb][0] = __uint_as_float(b2.x << 16); bias_v[nb][1] = __uint_as_float(b2.x & 0xffff0000u);
            bias_v[nb][2] = __uint_as_float(b2.y << 16); bias_v[nb][3] = __uint_as_float(b2.y & 0xffff0000u);
        }
    }
    const __amdgpu_buffer_rsrc_t out_rsrc = tile_rsrc(reinterpret_cast<const char*>(ep.out) + ((int64_t)m0 * ep.ldo + n_wave) * 2);
    const uint32_t out_voff = ((uint32_t)(lane >> 2) * (uint32_t)ep.ldo + (uint32_t)(lane & 3) * 8u) * 2u;
    uint4 rres[8];
    if (RES) {
        const __amdgpu_buffer_rsrc_t res_rsrc = tile_rsrc(reinterpret_cast<const char*>(ep.res) + ((int64_t)m0 * ep.ldr + n_wave) * 2);
        const uint32_t res_voff = ((uint32_t)(lane >> 2) * (uint32_t)ep.ldr + (uint32_t)(lane & 3) * 8u) * 2u;
#pragma unroll
        for (int ai = 0; ai < 8; ++ai) {
            const int row_l = (ai >> 2) * 128 + wm * 64 + (ai & 3) * 16;
            const u32x4_t r4 = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, res_voff, (uint32_t)row_l * (uint32_t)ep.ldr * 2u, 0);
            rres[ai] = uint4{r4.x, r4.y, r4.z, r4.w};
        }
    }
#pragma unroll
    for (int ai = 0; ai < 8; ++ai) {
        const int row_l = (ai >> 2) * 128 + wm * 64 + (ai & 3) * 16;
        // accumulator layout: row (lane & 15), columns nb * 16 + (lane >> 4) * 4 .. + 3  ->  patch[row][col]
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            uint2 o;
            o.x = pack_bf16x2(acc[ai][nb][0] + bias_v[nb][0], acc[ai][nb][1] + bias_v[nb][1]);
            o.y = pack_bf16x2(acc[ai][nb][2] + bias_v[nb][2], acc[ai][nb][3] + bias_v[nb][3]);
            *reinterpret_cast<uint2*>(patch + (lane & 15) * T21_PATCH_PITCH + (nb * 16 + (lane >> 4) * 4) * 2) = o;
        }
        // (wave-private patch: the LDS pipeline executes a wave's writes and reads in order)
        uint4 v = *reinterpret_cast<const uint4*>(patch + (lane >> 2) * T21_PATCH_PITCH + (lane & 3) * 16);
        if (RES) v = add_bf16x8(v, rres[ai]);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{v.x, v.y, v.z, v.w}, out_rsrc, out_voff,
                                               (uint32_t)row_l * (uint32_t)ep.ldo * 2u, 0);
    }
}

// ---- 256 x 128 tiles cut along K into fp32 slabs: the mid-batch projections at 97 .. 512 rows (round 5) ----------------------------
// At 256 rows the 64 x 64 slab kernel above takes 22 us for each of the QKV and h -> 4h launches (0.29 PF, 1.1-1.5 TB/s of weights;
// profiles/r4_bench_b256_kernel_stats.md) and 30 us for the residual launch; above 256 rows the stack used to fall to the large-M launch
// sequence, whose 256 x 256 tiles leave most CUs idle at 257 .. 512 rows.  This kernel is the one-tile 256 x 128 kernel's tile and wave layout (8 waves = two groups half a
// K-tile apart, a wave owns 64 rows of each A half x 32 columns, phase 0 = A0 + W, phase 1 = A1) with what this regime needs:
//   * the whole stack is ONE row tile (two at 257 .. 512 rows): rows >= M are clamped in the LDS-DMA addresses (their products are never
//     stored);
//   * a K range per workgroup: workgroup = (column tile, slice); slice s of a pair with nk K-tiles and S slices takes K-tiles
//     [s nk / S, (s + 1) nk / S) -- slices need not be equal, so 16 column tiles x 16 slices cover the residual projections' 128 + 32
//     K-tiles with 256 workgroups; slices >= s1 belong to the optional second operand pair;
//   * up to two problems per launch (the QKV projection and the MLP up-projection of a parallel-residual layer);
//   * the two operand streams are issued by DIFFERENT waves: group 0 streams A (activations: L2 hits, two K-tiles deep), group 1 streams
//     W (weights: HBM, every byte used by one workgroup only) TS_WD K-tiles deep with the nt policy.  vmcnt retires in order per wave,
//     so a wave that issued both could not wait for the next A tile without draining the deep W stream.  Measured against the first
//     version (the one-tile kernel's shared stream, five half-tiles ahead = 27 KiB of weights in flight per CU): 26.9 -> 25.8 us for the
//     QKV + h -> 4h launch at 256 rows, 22.0 -> 22.2 us for the residual launch (profiles/r5_bench_b256_kernel_stats_v{1,2}_*.md) -- the
//     K-loop was not the bound: the two launches differ by 5 K-tiles and 3.6 us, i.e. 0.72 us per K-tile (5 TB/s of weights inside the
//     loop) on top of ~14 us that do not depend on K (cold start of 224-256 workgroups, 29-34 MB of slab stores, the drain);
//   * the epilogue = 16-byte write-through stores of the fp32 fragments into slab[slice][M][N] -- the format the mid-batch reduce
//     kernels read.
// Synchronisation (t = K-tile, barriers counted per group; group 1 runs one barrier behind group 0, as in the one-tile kernel):
//   RAW  a wave waits for its own DMAs of a tile before a barrier that precedes every read of it: A0(t) / A1(t) by group 0 before the
//        barrier that opens its phase 0 / 1 of t; W(t) by group 1 before the barrier that closes its phase 1 of t - 1 (= the one group 0
//        opens phase 0 of t with);
//   WAR  A0(t + 2) is issued in group 0's phase 1 of t and A1(t + 1) in its phase 0 of t, W(t + TS_WD) in group 1's phase 1 of t: every
//        read of the slot's previous tenant was issued (group 1: completed, lgkmcnt(0)) at least one barrier earlier.
//   tests/test_gemm_ring_schedule.py (slab_model) replays both groups' streams for every K-tile count.
// Requires N % 128 == 0, K % 64 == 0, >= 4 K-tiles per slice (checked by the host).
struct TallSide {
    const bf16* A;
    const bf16* W;
    const bf16* A2;
    const bf16* W2;
    int64_t lda, ldw, lda2, ldw2;
    float* partial;
    int N, nk1, nk2, s1, s2;
};
struct TallArgs {
    TallSide g[2];
};
constexpr int TS_WD = 5;                                    // depth of the W ring (K-tiles)
constexpr int TS_A_STAGE = 2 * HALF_BYTES;                  // [A rows 0-127 | A rows 128-255] of one K-tile
constexpr int TS_SMEM = 2 * TS_A_STAGE + TS_WD * HALF_BYTES;   // 64 + 80 = 144 KiB

template <int OFF>
__device__ inline void glds16_sbase_off_nt(const char* base_uniform, uint32_t voff, uint32_t lds_base_uniform) {
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt"
                 :
                 : "v"(voff), "s"(base_uniform), "s"(lds_base_uniform), "n"(OFF)
                 : "memory", "scc");
}

// s_waitcnt vmcnt(4 * tiles): a W tile (or an A half-tile) is four DMA instructions of the issuing wave
__device__ inline void ts_wait_tiles(int tiles) {
    if (tiles >= 4) wait_vmcnt_n<16>();
    else if (tiles == 3) wait_vmcnt_n<12>();
    else if (tiles == 2) wait_vmcnt_n<8>();
    else if (tiles == 1) wait_vmcnt_n<4>();
    else wait_vmcnt_n<0>();
}

// ---- round 6: the QKV + h->4h launch of the 97 .. 512-row path with its epilogues FUSED (no reduce launch behind it) -----------------------
// Up to 256 rows a column tile's K range is cut in two (224 workgroups for the 1.2B layer); the workgroup with the FIRST K-tiles publishes
// its fp32 tile exactly as the slab kernel stores it (write-through) and raises one flag word per tile, the workgroup with the LAST
// K-tiles (a few more of them, so that it never waits) adds that tile to its accumulators and runs the epilogue the reduce launch used to:
// bias, RoPE on the first 32 dims of q / k (rot == 32: the partner column of e < 16 is e + 16 = the wave's other column fragment, same
// lane, same register), q -> qkv buffer, (k, v) -> cache rows pos0 + t, GELU for the h->4h side, bf16 stores.  257 .. 512 rows: one K
// range per tile (two row tiles), the epilogue runs on the accumulators directly.  The hand-off is the stream-K kernel's recipe with one
// flag per TILE (workgroup barrier on either side): write-through slab stores, s_waitcnt vmcnt(0), barrier, relaxed agent-scope flag store;
// bounded relaxed polls + s_sleep, barrier, agent-scope acquire, plain loads.  A consumer that gives up does not add the tile and stores
// SK_TIMEOUT_MAGIC into flags[TF_STICKY] (bt_decoder_forward reports it as BT_DEV_ERR_HANDOFF_TIMEOUT).
struct TallFuse {
    EpiArgs ep[2];       // side 0: QKV (ep.q.enabled), side 1: h->4h (ep.act == GELU)
    uint32_t* flags;     // one word per (side, tile), zero at launch and left zero; the caller's mid-batch counter words
    int first_nk;        // K-tiles of the publishing workgroup when a tile's K range is cut in two (the rest go to the finishing one)
    int half_mode;       // 129 .. 256 rows: a workgroup owns ONE 128-row half of a column tile over the whole K (grid = 2 x tiles, the two
                         // halves of a tile 8 block ids apart = on one XCD, so the W tile they both stream is fetched into that L2 once);
                         // its K-loop has one phase per K-tile (no A1 half) -- one K range, no hand-off
    uint64_t* tl;        // debug build only (bt_debug_timeline_buffer, else null): 16 s_memtime stamps per wave, [workgroup][wave][16]
    int dbg;             // debug build only, half mode, TIMING ONLY (wrong results): 1 = W addresses repeat after 4 K-tiles (W from L2), 2 = no A
                         // loads behind the first four, 4 = no MFMAs, 8 = nt W loads, 16 = no W loads behind the first TS_WD
};
constexpr int TF_STICKY = 1023;      // word of the 4 KiB counter region that holds the sticky time-out value

__device__ inline void tall_fused_epilogue(const EpiArgs& e, f32x4_t (&acc)[8][2], int m0, int nw, int wm, int lane, bool half_only = false) {
    const int M = e.M;
    const bool qkv = e.q.enabled != 0;
    const int cl = (lane >> 4) * 4;                       // the lane's 4 consecutive columns inside a 16-column fragment
    float bv[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (e.bias) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const uint2 b = *reinterpret_cast<const uint2*>(e.bias + nw + nb * 16 + cl);
            bv[nb][0] = __uint_as_float(b.x << 16); bv[nb][1] = __uint_as_float(b.x & 0xffff0000u);
            bv[nb][2] = __uint_as_float(b.y << 16); bv[nb][3] = __uint_as_float(b.y & 0xffff0000u);
        }
    }
    // the wave's 32 columns lie inside one head's q, k or v part (head_dim % 32 == 0)
    int head = 0, part = 0, el0 = 0, pos0 = 0, q_len = 1;
    bool rot = false;
    if (qkv) {
        const int D = e.q.head_dim;
        head = nw / (3 * D);
        const int within = nw - head * 3 * D;
        part = within / D;
        el0 = within - part * D;
        rot = part < 2 && el0 == 0;
        pos0 = __builtin_amdgcn_readfirstlane(e.q.pos0_dev ? *e.q.pos0_dev : e.q.pos0_host);
        q_len = e.q.q_len;
    }
    int rb[8], rt[8];                                      // (sequence, position inside the call) of the lane's row of every fragment
    float4 cs[8], sn[8];
#pragma unroll
    for (int ai = 0; ai < 8; ++ai) {
        int m = m0 + (ai >> 2) * 128 + wm * 64 + (ai & 3) * 16 + (lane & 15);
        m = m < M ? m : M - 1;
        if (q_len == 1) { rb[ai] = m; rt[ai] = 0; }
        else if (q_len == 2) { rb[ai] = m >> 1; rt[ai] = m & 1; }
        else { rb[ai] = m / q_len; rt[ai] = m - rb[ai] * q_len; }
    }
    if (rot) {
#pragma unroll
        for (int ai = 0; ai < 8; ++ai) {
            const int rpos = e.q.pos_table ? e.q.pos_table[pos0 + rt[ai]] : pos0 + rt[ai];
            cs[ai] = *reinterpret_cast<const float4*>(e.q.cos_tab + (int64_t)rpos * 16 + cl);
            sn[ai] = *reinterpret_cast<const float4*>(e.q.sin_tab + (int64_t)rpos * 16 + cl);
        }
    }
#pragma unroll
    for (int ai = 0; ai < 8; ++ai) {
        if (half_only && ai >= 4) break;                  // (a workgroup that owns one 128-row half has no A1 accumulators)
        const int m = m0 + (ai >> 2) * 128 + wm * 64 + (ai & 3) * 16 + (lane & 15);
        float v[2][4];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[nb][r] = acc[ai][nb][r] + bv[nb][r];
        if (rot) {
            const float c[4] = {cs[ai].x, cs[ai].y, cs[ai].z, cs[ai].w}, sv[4] = {sn[ai].x, sn[ai].y, sn[ai].z, sn[ai].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x1 = v[0][r], x2 = v[1][r];
                v[0][r] = x1 * c[r] - x2 * sv[r];
                v[1][r] = x2 * c[r] + x1 * sv[r];
            }
        }
        if (e.act == BT_ACT_GELU) {
            gelu4(v[0]);
            gelu4(v[1]);
        }
        if (m < M) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const uint2 o = uint2{pack_bf16x2(v[nb][0], v[nb][1]), pack_bf16x2(v[nb][2], v[nb][3])};
                bf16* dst;
                if (!qkv || part == 0) dst = e.out + (int64_t)m * e.ldo + nw + nb * 16 + cl;
                else
                    dst = (part == 1 ? e.q.kc : e.q.vc) +
                          (((int64_t)rb[ai] * e.q.heads + head) * e.q.n_max + pos0 + rt[ai]) * e.q.head_dim + el0 + nb * 16 + cl;
                *reinterpret_cast<uint2*>(dst) = o;
            }
        }
    }
}

// WT: the slab stores are write-through (sc1).  With plain stores the 29-34 MB of slabs sit dirty in the eight L2s until the end-of-kernel
// release writes them back -- after the last workgroup, on the critical path of the reduce launch that follows; written through they
// drain while the other workgroups still compute (guide "publish-large" / "splitk-seam": 64 KB fp32 slabs per workgroup at M = 256).
template <bool WT, bool FUSE = false>
__global__ __launch_bounds__(512) void gemm_bf16_mfma256x128_slab_kernel(const TallArgs a, int M, int wg_side1, const TallFuse tf = TallFuse{}) {
    static_assert(TS_WD >= 2 && TS_WD <= 5, "ts_wait_tiles covers up to four W tiles in flight behind the awaited one");
    __shared__ __attribute__((aligned(16))) char smem[TS_SMEM];
#ifdef BT_DEBUG_HOOKS
    // timeline of the fused launch (tools/bench_knobs.py, BT_MID_TL=1): 0 entry, 1 prologue issued, 2 first K-tile opens, 3 / 4 / 5 a quarter /
    // half / three quarters of the K-tiles done, 6 K-loop done, 7 epilogue starts, 8 its last store issued, 9 s_memrealtime at entry
    uint64_t tstamp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bool tl_on = false;
    if constexpr (FUSE) tl_on = tf.tl != nullptr;
#define TS_STAMP(i) do { if constexpr (FUSE) { if (tl_on) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tstamp[i]) : : "memory"); } } while (0)
#define TS_STAMP_AT(t, nk) do { if constexpr (FUSE) { if (tl_on) { if ((t) == (nk) / 4) TS_STAMP(3); else if ((t) == (nk) / 2) TS_STAMP(4); else if ((t) == (nk) * 3 / 4) TS_STAMP(5); } } } while (0)
    if constexpr (FUSE) { if (tl_on) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tstamp[0]), "=s"(tstamp[9]) : : "memory"); }
#else
#define TS_STAMP(i) do {} while (0)
#define TS_STAMP_AT(t, nk) do {} while (0)
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const bool grp1 = wave >= 4;
    const int wq = wave & 3;                 // this wave's quarter of the tile its group streams
    // side, column tile and slice: workgroup-uniform index into the kernel-argument array (scalar loads at a computed offset)
    bool half_mode = false;
    int lin = (int)blockIdx.x, half = 0;
    if constexpr (FUSE) {
        half_mode = tf.half_mode != 0;
        if (half_mode) {     // block b = 8 q + x: half = q & 1, column tile (q >> 1) * 8 + x -- the two halves of a tile share x = the XCD
            const int q = (int)blockIdx.x >> 3, x = (int)blockIdx.x & 7;
            half = q & 1;
            lin = (q >> 1) * 8 + x;
        }
    }
    const int side = lin >= wg_side1 ? 1 : 0;
    const TallSide& g = a.g[side];
    const int id = lin - side * wg_side1;
    const int S = g.s1 + g.s2;
    const int slice = id % S, tile = id / S;   // consecutive ids (= XCDs) take different slices: an XCD's L2 holds S / 8 of the A columns
    const int tiles_n = g.N >> 7;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;   // 257 .. 512 rows: two row tiles
    const int m0 = tm * 256 + half * 128;
    const bool second = slice >= g.s1;
    const int sidx = second ? slice - g.s1 : slice, scnt = second ? g.s2 : g.s1, nk_all = second ? g.nk2 : g.nk1;
    int kb = sidx * nk_all / scnt, ke = (sidx + 1) * nk_all / scnt;   // (slices x K-tiles << 2^31)
    if constexpr (FUSE) {     // two K ranges: [0, first_nk) is published, [first_nk, nk_all) finishes the tile (host: S <= 2, one operand pair)
        if (S == 2) {
            kb = slice == 0 ? 0 : tf.first_nk;
            ke = slice == 0 ? tf.first_nk : nk_all;
        }
    }
    const int nk = ke - kb;
    const bf16* A = second ? g.A2 : g.A;
    const bf16* W = second ? g.W2 : g.W;
    const int64_t lda = second ? g.lda2 : g.lda, ldw = second ? g.ldw2 : g.ldw;
    const int N = g.N, n0 = tn * 128;

    const uint32_t lane_slot = (uint32_t)(((lane & 7) ^ ((lane >> 3) & 7)) * 16);
    // group 0: rows h * 128 + wq * 32 + j * 8 + (lane >> 3) of A, clamped to the stack; group 1: rows n0 + wq * 32 + j * 8 + (lane >> 3) of W
    uint32_t va[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            va[h][j] = (uint32_t)min(m0 + h * 128 + wq * 32 + j * 8 + (lane >> 3), M - 1) * (uint32_t)(lda * 2) + lane_slot;
    const uint32_t vw = (uint32_t)(lane >> 3) * (uint32_t)(ldw * 2) + lane_slot;
    const uint32_t row8_w = (uint32_t)(8 * ldw * 2);
    const char* pa = reinterpret_cast<const char*>(A) + (int64_t)kb * (BK * 2);
    const char* pw = reinterpret_cast<const char*>(W) + ((int64_t)(n0 + wq * 32) * ldw + (int64_t)kb * BK) * 2;
    const uint32_t a_lds = lds_addr(smem) + (uint32_t)wq * 4096u;
    const uint32_t w_lds = lds_addr(smem) + (uint32_t)(2 * TS_A_STAGE) + (uint32_t)wq * 4096u;
    auto issue_a = [&](auto hc, int kt) {    // group 0: half-tile `half` of K-tile kt -> A ring stage kt & 1
        constexpr int half = decltype(hc)::value;
        const uint32_t ring = a_lds + (uint32_t)(kt & 1) * (uint32_t)TS_A_STAGE;
        const uint32_t koff = (uint32_t)kt * 128u;
        glds16_sbase_off<half * HALF_BYTES>(pa, va[half][0] + koff, ring);
        glds16_sbase_off<half * HALF_BYTES + 1024>(pa, va[half][1] + koff, ring);
        glds16_sbase_off<half * HALF_BYTES + 2048>(pa, va[half][2] + koff, ring);
        glds16_sbase_off<half * HALF_BYTES + 3072>(pa, va[half][3] + koff, ring);
    };
#ifdef BT_DEBUG_HOOKS
    int dbg = 0;
    if constexpr (FUSE) dbg = tf.dbg;
#else
    constexpr int dbg = 0;
#endif
    int w_slot = 0;                          // group 1: ring slot of the next W tile to issue (= its K-tile mod TS_WD)
    auto issue_w = [&](int kt) {
        const uint32_t ring = w_lds + (uint32_t)w_slot * (uint32_t)HALF_BYTES;
        const uint32_t v0 = vw + (uint32_t)kt * 128u;
        glds16_sbase_off_nt<0>(pw, v0, ring);
        glds16_sbase_off_nt<1024>(pw, v0 + row8_w, ring);
        glds16_sbase_off_nt<2048>(pw, v0 + 2 * row8_w, ring);
        glds16_sbase_off_nt<3072>(pw, v0 + 3 * row8_w, ring);
        w_slot = w_slot + 1 == TS_WD ? 0 : w_slot + 1;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    auto issue_ah = [&](int kt) {            // half mode, group 0: the workgroup's 128 rows of K-tile kt -> A ring slot kt & 3 (four 16 KiB slots)
        const uint32_t ring = a_lds + (uint32_t)(kt & 3) * (uint32_t)HALF_BYTES;
        const uint32_t koff = (uint32_t)kt * 128u;
        if ((dbg & 2) && kt >= 4) return;
        glds16_sbase_off<0>(pa, va[0][0] + koff, ring);
        glds16_sbase_off<1024>(pa, va[0][1] + koff, ring);
        glds16_sbase_off<2048>(pa, va[0][2] + koff, ring);
        glds16_sbase_off<3072>(pa, va[0][3] + koff, ring);
    };
    auto issue_wh = [&](int kt) {            // half mode, group 1: W(kt) with plain loads -- the tile's other half streams the same W tile on the same XCD
        const uint32_t ring = w_lds + (uint32_t)w_slot * (uint32_t)HALF_BYTES;
        const uint32_t v0 = vw + (uint32_t)((dbg & 1) ? (kt & 3) : kt) * 128u;
        if ((dbg & 16) && kt >= TS_WD) {
        } else if (dbg & 8) {
            glds16_sbase_off_nt<0>(pw, v0, ring);
            glds16_sbase_off_nt<1024>(pw, v0 + row8_w, ring);
            glds16_sbase_off_nt<2048>(pw, v0 + 2 * row8_w, ring);
            glds16_sbase_off_nt<3072>(pw, v0 + 3 * row8_w, ring);
        } else {
            glds16_sbase_off<0>(pw, v0, ring);
            glds16_sbase_off<1024>(pw, v0 + row8_w, ring);
            glds16_sbase_off<2048>(pw, v0 + 2 * row8_w, ring);
            glds16_sbase_off<3072>(pw, v0 + 3 * row8_w, ring);
        }
        w_slot = w_slot + 1 == TS_WD ? 0 : w_slot + 1;
    };
    // prologue: group 0 -> A0(0), A1(0), A0(1) (half mode: A(0 .. 2)); group 1 -> W(0 .. TS_WD - 1)
    if (!grp1) {
        if (half_mode) {
            for (int kt = 0; kt < 3 && kt < nk; ++kt) issue_ah(kt);
        } else {
            issue_a(I0{}, 0);
            issue_a(I1{}, 0);
            if (nk > 1) issue_a(I0{}, 1);
        }
    } else {
        for (int kt = 0; kt < TS_WD && kt < nk; ++kt) {
            if (half_mode) issue_wh(kt);
            else issue_w(kt);
        }
    }

    TS_STAMP(1);
    f32x4_t acc[8][2];   // [A half * 4 + m-fragment][n-fragment]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t af[4][2], wf[2][2];

    if (grp1) {
        ts_wait_tiles(min(TS_WD, nk) - 1);   // W(0) has landed
        cbarrier();
    }
    TS_STAMP(2);
    int r_slot = 0;                          // ring slot of W(t)
    if (!half_mode) {
        for (int t = 0; t < nk; ++t) {
            TS_STAMP_AT(t, nk);
            const char* abuf = smem + (t & 1) * TS_A_STAGE;
            const char* wbuf = smem + 2 * TS_A_STAGE + r_slot * HALF_BYTES;
            r_slot = r_slot + 1 == TS_WD ? 0 : r_slot + 1;
            // ---- phase 0: A0, W ----
            if (!grp1) ts_wait_tiles(t + 1 < nk ? 2 : 1);   // A0(t) landed; A1(t) and A0(t + 1) may be in flight
            cbarrier();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) wf[nb][ks] = read_frag(wbuf, wn * 32 + nb * 16, ks, lane);
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i][ks] = read_frag(abuf, wm * 64 + i * 16, ks, lane);
            }
            if (!grp1 && t + 1 < nk) issue_a(I1{}, t + 1);
            if (grp1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // A0 / W reads done before their slots may be refilled
            cbarrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        acc[i][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nb][ks], af[i][ks], acc[i][nb], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            // ---- phase 1: A1 ----
            if (!grp1) ts_wait_tiles(t + 1 < nk ? 2 : 0);        // A1(t) landed; A0(t + 1) and A1(t + 1) may be in flight
            cbarrier();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i][ks] = read_frag(abuf + HALF_BYTES, wm * 64 + i * 16, ks, lane);
            if (!grp1 && t + 2 < nk) issue_a(I0{}, t + 2);
            if (grp1) {
                if (t + TS_WD < nk) issue_w(t + TS_WD);
                ts_wait_tiles(min(t + TS_WD, nk - 1) - (t + 1));   // W(t + 1) landed (its wait for phase 0 of K-tile t + 1)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // A1 reads done before group 0 may refill the slot
            }
            cbarrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        acc[4 + i][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nb][ks], af[i][ks], acc[4 + i][nb], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
    } else {
        // ---- half mode: ONE phase per K-tile (two barriers instead of four).  Group 1 runs one barrier behind group 0, so one group's
        // fragment reads overlap the other's MFMAs.  A ring = four 16 KiB slots (the [A0 | A1] stages of the full form), A(t + 3)
        // issued by group 0 behind the barrier that opens its reads of t (both groups' reads of t - 1 are complete there: group 0's fed
        // its MFMAs of t - 1, group 1 drained lgkmcnt in front of the same barrier); W(t + TS_WD) issued by group 1 behind the barrier
        // that closes its reads of t (group 0 passed it after issuing its MFMAs of t).  RAW: A(t) waited for by group 0 in front of
        // the barrier that opens its reads of t, W(t + 1) by group 1 in front of the barrier that closes its reads of t (= the one
        // group 0 opens its reads of t + 1 with).
        // Measured at 256 rows (profiles/r6_mid_half_mode_ablations.md): this form 31.1 us per launch; the full loop's four barriers
        // with an empty phase 1: 34.5; one barrier per K-tile with the fragments double-buffered in registers: 33.5; the refills
        // issued piecewise inside the MFMA block: 38.6.  Timing-only ablations put the K-tile at ~1600 cycles against 512 of MFMA per
        // SIMD: W served from L2 alone (an eighth of the HBM traffic) changes nothing, no W loads -18 %, no MFMAs -26 %, neither 1080
        // -- the per-CU load path (32 KiB per K-tile in 1 KiB LDS-DMA pieces) and the barrier-coupled phases bound it, not HBM.
        for (int t = 0; t < nk; ++t) {
            TS_STAMP_AT(t, nk);
            const char* abuf = smem + (t & 3) * HALF_BYTES;
            const char* wbuf = smem + 2 * TS_A_STAGE + r_slot * HALF_BYTES;
            r_slot = r_slot + 1 == TS_WD ? 0 : r_slot + 1;
            if (!grp1) ts_wait_tiles(min(t + 2, nk - 1) - t);   // A(t) landed; A(t + 1), A(t + 2) may be in flight
            cbarrier();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) wf[nb][ks] = read_frag(wbuf, wn * 32 + nb * 16, ks, lane);
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i][ks] = read_frag(abuf, wm * 64 + i * 16, ks, lane);
            }
            if (!grp1 && t + 3 < nk) issue_ah(t + 3);
            if (grp1) {
                ts_wait_tiles(max(min(t + TS_WD - 1, nk - 1) - (t + 1), 0));   // W(t + 1) landed; W(t + 2 .. t + TS_WD - 1) may be in flight
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // A / W reads done before their slots may be refilled
            }
            cbarrier();
            if (grp1 && t + TS_WD < nk) issue_wh(t + TS_WD);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            if (!(dbg & 4)) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb)
                            acc[i][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nb][ks], af[i][ks], acc[i][nb], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
        }
    }
    TS_STAMP(6);
    if (!grp1) cbarrier();
    TS_STAMP(7);
#ifdef BT_DEBUG_HOOKS
    auto tl_dump = [&]() {
        if constexpr (FUSE) {
            if (tl_on) {
                TS_STAMP(8);
                if (lane == 0) {
                    uint64_t* o = tf.tl + ((int64_t)blockIdx.x * 8 + wave) * 16;
#pragma unroll
                    for (int i = 0; i < 10; ++i) o[i] = tstamp[i];
                }
            }
        }
    };
#else
    auto tl_dump = [&]() {};
#endif
    if constexpr (FUSE) {
        const EpiArgs& ep = tf.ep[side];
        uint32_t* flag = tf.flags + (side ? (wg_side1 / S) : 0) + tile;          // one word per (side, tile)
        if (S == 2 && slice == 1) {
            // finishing workgroup: wait for the partner's tile (wave 0 polls, everybody passes the barrier behind it), add it
            uint32_t seen = 1;
            if (wave == 0) {
                seen = 0;
                for (uint32_t spins = 0; spins < SK_SPIN_LIMIT; ++spins) {
                    seen = sk_flag_load(flag);
                    if (seen != 0) break;
                    __builtin_amdgcn_s_sleep(8);
                }
                if (seen == 0) sk_flag_store(tf.flags + TF_STICKY, SK_TIMEOUT_MAGIC);   // gave up: the tile stays incomplete, the pass is flagged
                else sk_flag_store(flag, 0u);                                         // ready for the next launch
            }
            cbarrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            // (every wave looks at the sticky word instead of sharing wave 0's verdict through LDS: it is only ever non-zero after a give-up)
            const bool gave_up = sk_flag_load(tf.flags + TF_STICKY) == SK_TIMEOUT_MAGIC;
            if (!gave_up) {
                const __amdgpu_buffer_rsrc_t prs = tile_rsrc(reinterpret_cast<const char*>(g.partial + n0 + wn * 32));
                const uint32_t pvoff = ((uint32_t)(lane & 15) * (uint32_t)N + (uint32_t)(lane >> 4) * 4u) * 4u;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    u32x4_t pv[4][2];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        int row0 = m0 + h * 128 + wm * 64 + i * 16;
                        row0 = row0 + (lane & 15) < M ? row0 : 0;           // rows beyond the stack: any valid address (their sums are never stored)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb)
                            pv[i][nb] = __builtin_amdgcn_raw_buffer_load_b128(prs, pvoff, ((uint32_t)row0 * (uint32_t)N + (uint32_t)nb * 16u) * 4u, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) {
                            acc[h * 4 + i][nb][0] += __uint_as_float(pv[i][nb].x);
                            acc[h * 4 + i][nb][1] += __uint_as_float(pv[i][nb].y);
                            acc[h * 4 + i][nb][2] += __uint_as_float(pv[i][nb].z);
                            acc[h * 4 + i][nb][3] += __uint_as_float(pv[i][nb].w);
                        }
                }
            }
            tall_fused_epilogue(ep, acc, m0, n0 + wn * 32, wm, lane);
            tl_dump();
            return;
        }
        if (S == 1) {
            tall_fused_epilogue(ep, acc, m0, n0 + wn * 32, wm, lane, half_mode);
            tl_dump();
            return;
        }
        // S == 2, slice 0: publish below (slab region of slice 0), then the flag
    }
    // ---- slab stores: accumulator fragment = row (lane & 15), columns nb * 16 + (lane >> 4) * 4 .. + 3 ----
    const __amdgpu_buffer_rsrc_t srs = tile_rsrc(reinterpret_cast<const char*>(g.partial + ((int64_t)slice * M) * N + n0 + wn * 32));
    const uint32_t svoff = ((uint32_t)(lane & 15) * (uint32_t)N + (uint32_t)(lane >> 4) * 4u) * 4u;
#pragma unroll
    for (int ai = 0; ai < 8; ++ai) {
        const int row0 = m0 + (ai >> 2) * 128 + wm * 64 + (ai & 3) * 16;
        if (row0 + (lane & 15) < M) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const f32x4_t v = acc[ai][nb];
                __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]),
                                                               __float_as_uint(v[3])},
                                                       srs, svoff, ((uint32_t)row0 * (uint32_t)N + (uint32_t)nb * 16u) * 4u, (WT || FUSE) ? 16 : 0);
            }
        }
    }
    if constexpr (FUSE) {     // publisher: every wave's stores have left the CU, then ONE flag for the tile
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        cbarrier();
        if (wave == 0) sk_flag_store(tf.flags + (side ? (wg_side1 / S) : 0) + tile, 1u);
    }
}

#undef TS_STAMP
#undef TS_STAMP_AT

// ---- round 6 EXPERIMENT (debug build only, bt_debug_mid_tall_fuse(5)): the fused QKV + h->4h launch at 129 .. 256 rows, STREAM form -------------
// Measured SLOWER than the half-mode loop above (35.0 us per launch against 31.1 at 256 rows; with A eight K-tiles ahead as well: 36.6),
// oracle-clean (tests/test_model_gpu.py::test_mid_rows_fused_launch_at_1p2b_width_vs_oracle).  Kept as the A/B that rules out two
// explanations of the half mode's 0.75 us per K-tile -- LDS-DMA issue cost / barrier count (none of the former, half of the latter here)
// and prefetch depth (4 -> 8 K-tiles changes nothing): either form moves 32 KiB per K-tile into a CU in 0.75-0.9 us
// (profiles/r6_mid_half_mode_ablations.md; the cause is not isolated).
// The kernel does the same work per workgroup -- one 128-row half of a 128-column tile over the whole K, same grid, same epilogue -- with
//   * W never in LDS: a wave streams ITS fragments global -> registers (two 16-byte loads per lane and K-tile, MS_PW K-tiles ahead), so the
//     weight stream needs no barrier, no LDS-DMA issue and no ring slot;
//   * the K-tile's two 32-wide k-steps split between the wave pair (kh = wave >> 2): a wave owns 128 rows x 32 columns at ONE k-step
//     (8 A fragments + 2 W fragments -> 16 MFMAs), so no W fragment is loaded twice and a wave reads half of the A tile;
//   * A through LDS by plain loads + ds_write_b128 (MS_PA K-tiles ahead in registers, double-buffered in LDS): every vmcnt is the
//     compiler's, counted exactly; ONE barrier per K-tile (behind it: A(t + 1) registers -> the other LDS buffer, fragment reads of
//     A(t), the refills, the MFMAs);
//   * at the end the pair exchanges halves through LDS (kh = 0 keeps rows 0-63 of the half, kh = 1 rows 64-127, each adds the partner's
//     partial sums) and every wave runs tall_fused_epilogue on 64 rows x 32 columns exactly as in half mode.
// Hazards: the LDS buffer (t + 1) & 1 was last read for A(t - 1); every wave drains lgkmcnt in front of barrier B(t), the writes of
// A(t + 1) come behind it and are drained in front of B(t + 1), the reads of A(t + 1) come behind that.
// Requires nk % MS_PW == 0, nk >= 2 MS_PW, N % 128 == 0, one K range per tile (host: mid_qkv_gelu).
#ifdef BT_DEBUG_HOOKS
constexpr int MS_PW = 8;      // W fragments in flight (K-tiles)
constexpr int MS_PA = 4;      // A row pieces in flight in registers (K-tiles; 8 measured no faster: 36.6 vs 35.0 us)
constexpr int MS_SMEM = 65536;

__global__ __launch_bounds__(512) void gemm_bf16_mid_stream_kernel(const TallArgs a, int M, int wg_side1, const TallFuse tf) {
    static_assert(MS_PW % MS_PA == 0 && MS_PW % 2 == 0, "static register / buffer indices inside one unrolled group");
    __shared__ __attribute__((aligned(16))) char smem[MS_SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 2, wn = wave & 3;
    const int q = (int)blockIdx.x >> 3, x = (int)blockIdx.x & 7;     // block b = 8 q + x: half = q & 1, column tile (q >> 1) * 8 + x
    const int half = q & 1, lin = (q >> 1) * 8 + x;
    const int side = lin >= wg_side1 ? 1 : 0;
    const TallSide& g = a.g[side];
    const int tile = lin - side * wg_side1;
    const int n0 = tile * 128, m0 = half * 128;
    const int nk = g.nk1;
    const int64_t lda = g.lda, ldw = g.ldw;
    // A: wave w stages rows w * 16 .. + 15 of the half (two 8-row pieces), lane = (row in piece, 16-byte chunk); LDS image = the
    // XOR-swizzled 128-byte rows read_frag expects
    const char* pa[2];
    uint32_t la[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = wave * 16 + j * 8 + (lane >> 3);
        pa[j] = reinterpret_cast<const char*>(g.A) + (int64_t)min(m0 + r, M - 1) * lda * 2 + (lane & 7) * 16;
        la[j] = (uint32_t)(r * 128 + (((lane & 7) ^ (r & 7)) << 4));
    }
    // W: rows n0 + wn * 32 + nb * 16 + (lane & 15), bytes kt * 128 + kh * 64 + (lane >> 4) * 16
    const char* pw[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
        pw[nb] = reinterpret_cast<const char*>(g.W) + (int64_t)(n0 + wn * 32 + nb * 16 + (lane & 15)) * ldw * 2 + kh * 64 + (lane >> 4) * 16;
    auto load_a = [&](int kt, bf16x8_t (&r)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) r[j] = *reinterpret_cast<const bf16x8_t*>(pa[j] + (int64_t)kt * 128);
    };
    auto load_w = [&](int kt, bf16x8_t (&r)[2]) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) r[nb] = *reinterpret_cast<const bf16x8_t*>(pw[nb] + (int64_t)kt * 128);   // (plain: the tile's other half reads the same lines on this XCD)
    };
    auto write_a = [&](int buf, const bf16x8_t (&r)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<bf16x8_t*>(smem + buf * 16384 + la[j]) = r[j];
    };

    bf16x8_t ar[MS_PA][2], wr[MS_PW][2];
    f32x4_t acc[8][2];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // prologue: A(0) -> LDS buffer 0; A(1 .. MS_PA) and W(0 .. MS_PW - 1) in registers
    load_a(0, ar[0]);
#pragma unroll
    for (int k = 0; k < MS_PW; ++k) load_w(k, wr[k]);
    write_a(0, ar[0]);
#pragma unroll
    for (int k = 1; k <= MS_PA; ++k) load_a(k, ar[k % MS_PA]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    auto group = [&](int t0, auto last_c) {
        constexpr bool LAST = decltype(last_c)::value;
        auto step = [&](auto sc) {
            constexpr int st = decltype(sc)::value;
            const int t = t0 + st;
            cbarrier();                                                       // B(t): A(t) is in buffer t & 1; every read of A(t - 1) is done
            if constexpr (!LAST || st + 1 < MS_PW) write_a((st + 1) & 1, ar[(st + 1) % MS_PA]);   // A(t + 1) -> the other buffer
            bf16x8_t af[8];
            const char* abuf = smem + (st & 1) * 16384;
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = read_frag(abuf, i * 16, kh, lane);
            if constexpr (!LAST || st + 1 + MS_PA < MS_PW) load_a(t + 1 + MS_PA, ar[(st + 1) % MS_PA]);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) acc[i][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[st][nb], af[i], acc[i][nb], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!LAST) load_w(t + MS_PW, wr[st]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the writes of A(t + 1) (and the reads of A(t)) are done before B(t + 1)
        };
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        step(std::integral_constant<int, 4>{});
        step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 6>{});
        step(std::integral_constant<int, 7>{});
    };
    static_assert(MS_PW == 8, "the group body is written out for eight steps");
    int t0 = 0;
    for (; t0 + MS_PW < nk; t0 += MS_PW) group(t0, std::false_type{});
    group(t0, std::true_type{});

    // ---- the pair exchanges halves: kh = 0 finishes rows 0-63 of the half, kh = 1 rows 64-127 ----
    cbarrier();                                                               // every read of the A buffers is done: smem is free
    {
        char* mine = smem + wave * 8192;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const f32x4_t v = kh ? acc[i][nb] : acc[4 + i][nb];           // the rows this wave gives away
                *reinterpret_cast<f32x4_t*>(mine + (i * 2 + nb) * 1024 + lane * 16) = v;
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    cbarrier();
    f32x4_t fin[8][2];
    {
        const char* theirs = smem + (wave ^ 4) * 8192;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const f32x4_t p = *reinterpret_cast<const f32x4_t*>(theirs + (i * 2 + nb) * 1024 + lane * 16);
                const f32x4_t k = kh ? acc[4 + i][nb] : acc[i][nb];
                fin[i][nb] = k + p;
                fin[4 + i][nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
    }
    tall_fused_epilogue(tf.ep[side], fin, m0, n0 + wn * 32, kh, lane, true);
}
#endif

// ---- GEMV for a handful of rows (M <= 4): the weight-streaming regime of small-batch decode ------------------------------
// out[m, n] = epi(sum_k x[m, k] * W[n, k]).  One pass over W at HBM speed is the whole cost (N*K*2 bytes; x is a few KiB), so
// the kernel is a pure streamer: a wave owns RW rows of W, every lane keeps 4 x 16-byte loads per row in flight (a 2048-wide
// K chunk = 4 coalesced 1 KiB wave-loads per row), the x chunk sits in LDS (MR x 4 KiB, read as ds_read_b128 and reused
// for the wave's RW rows), products accumulate in fp32 per lane, and one butterfly per (row, m) finishes the dot product.
// Bias / GELU / residual are applied by the lane that stores the element.  Replaces the split-K MFMA kernel + reduce kernel
// pair for M <= 4 (two launches, fp32 partial round trip): B = 1 decode 700 -> 1084 tok/s end to end.
constexpr int GV_KC = 2048;   // K chunk staged in LDS

// 16-byte weight load; NT = non-temporal (streamed once per forward: do not displace reusable lines in L2 / MALL; measured on
// this part for a decode layer: issued -> landed -18 %, guide "nt-weights")
template <bool NT>
__device__ inline uint4 load_w16(const bf16* p) {
    if constexpr (NT) {
        const u32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
        return uint4{t.x, t.y, t.z, t.w};
    } else {
        return *reinterpret_cast<const uint4*>(p);
    }
}

__device__ inline float dot8_bf16(const uint4& w, const uint4& x, float acc) {
    acc = fmaf(__uint_as_float(w.x << 16), __uint_as_float(x.x << 16), acc);
    acc = fmaf(__uint_as_float(w.x & 0xffff0000u), __uint_as_float(x.x & 0xffff0000u), acc);
    acc = fmaf(__uint_as_float(w.y << 16), __uint_as_float(x.y << 16), acc);
    acc = fmaf(__uint_as_float(w.y & 0xffff0000u), __uint_as_float(x.y & 0xffff0000u), acc);
    acc = fmaf(__uint_as_float(w.z << 16), __uint_as_float(x.z << 16), acc);
    acc = fmaf(__uint_as_float(w.z & 0xffff0000u), __uint_as_float(x.z & 0xffff0000u), acc);
    acc = fmaf(__uint_as_float(w.w << 16), __uint_as_float(x.w << 16), acc);
    acc = fmaf(__uint_as_float(w.w & 0xffff0000u), __uint_as_float(x.w & 0xffff0000u), acc);
    return acc;
}

// LN: the staged x rows are LayerNorm-ed in LDS first (gamma / beta / eps in `ln`; needs K <= GV_KC so a row's statistics
// see the whole row): removes the separate LayerNorm launch in front of the QKV and h->4h projections.  QKV: the wave owns
// the row PAIR (n, n + 16) of a 32-column group, so the lane that finishes row m can rotate the pair (RoPE) and scatter
// q to the qkv buffer / k, v to their KV-cache rows itself: removes the RoPE + append launch.  Small-batch decode is
// launch-bound (~5 us per dependent launch): 7 -> 5 launches per layer.

// local-window attention of the (sequence, head) items of a <= 2-row step into LDS rows xs_att[m][head * 128 + ..] (bf16, the
// layout the GEMV stages its x rows in).  One wave per item with local_attend_kernel's lane mapping and arithmetic (keep in sync:
// tests/test_model_gpu.py::test_few_rows_fused_attention_is_bit_identical compares the two).  The items of a wave (it = j * 4 + wave)
// are spread over the GEMV's first-segment K chunks, <= ATT_HPW per chunk: `issue` goes out right BEFORE the chunk's weight-row
// loads (L2 hits retire first: the wait leaves the HBM stream in flight), `finish` runs in their shadow.
constexpr int ATT_HPW = 2;
template <int U>
struct LocalAttRegs {
    uint4 kq[ATT_HPW][U], vq[ATT_HPW][U], qq[ATT_HPW][2];
};

template <int U>
__device__ __forceinline__ void local_att_issue(const LocalAtt& la, int items, int j0, int j1, int lane, int wave, LocalAttRegs<U>& r) {
    constexpr int D = 128;
    const int c = lane & 15, sub = lane >> 4;
    const int nk_all = la.pos0 + la.q_len;
#pragma unroll
    for (int h = 0; h < ATT_HPW; ++h) {
        int it = (j0 + h < j1 ? j0 + h : j0) * 4 + wave;
        it = it < items ? it : items - 1;
        const int b = it / la.heads, head = it - b * la.heads;
        const bf16* kb = la.kc + ((int64_t)b * la.heads + head) * la.n_max * D + c * 8;
        const bf16* vb = la.vc + ((int64_t)b * la.heads + head) * la.n_max * D + c * 8;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int row = u * 4 + sub;
            row = row < nk_all ? row : nk_all - 1;
            r.kq[h][u] = *reinterpret_cast<const uint4*>(kb + (int64_t)row * D);
            r.vq[h][u] = *reinterpret_cast<const uint4*>(vb + (int64_t)row * D);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int tt = t < la.q_len ? t : la.q_len - 1;
            r.qq[h][t] = *reinterpret_cast<const uint4*>(la.qkv + ((int64_t)b * la.q_len + tt) * la.ld_qkv + (int64_t)head * 3 * D + c * 8);
        }
    }
}

template <int U>
__device__ __forceinline__ void local_att_finish(const LocalAtt& la, int items, int j0, int j1, int lane, int wave,
                                                 const LocalAttRegs<U>& r, char* xs_att) {
    constexpr int D = 128;
    const int c = lane & 15, sub = lane >> 4;
#pragma unroll
    for (int h = 0; h < ATT_HPW; ++h) {
        if (j0 + h >= j1) break;
        const int it = (j0 + h) * 4 + wave;
        if (it >= items) break;
        const int b = it / la.heads, head = it - b * la.heads;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t >= la.q_len) break;
            const int nk = la.pos0 + t + 1;
            float q[8];
            {
                const uint32_t w[4] = {r.qq[h][t].x, r.qq[h][t].y, r.qq[h][t].z, r.qq[h][t].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    q[2 * i] = __uint_as_float(w[i] << 16) * la.scale;
                    q[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u) * la.scale;
                }
            }
            float sc[U];
            float mx = -INFINITY;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t w[4] = {r.kq[h][u].x, r.kq[h][u].y, r.kq[h][u].z, r.kq[h][u].w};
                float d = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    d = fmaf(q[2 * i], __uint_as_float(w[i] << 16), d);
                    d = fmaf(q[2 * i + 1], __uint_as_float(w[i] & 0xffff0000u), d);
                }
                d = row16_sum_dpp(d);
                const int row = u * 4 + sub;
                sc[u] = row < nk ? d : -INFINITY;
                mx = fmaxf(mx, sc[u]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float l = 0.f;
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float p = (mx == -INFINITY) ? 0.f : __expf(sc[u] - mx);
                l += p;
                const uint32_t w[4] = {r.vq[h][u].x, r.vq[h][u].y, r.vq[h][u].z, r.vq[h][u].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[2 * i] = fmaf(p, __uint_as_float(w[i] << 16), acc[2 * i]);
                    acc[2 * i + 1] = fmaf(p, __uint_as_float(w[i] & 0xffff0000u), acc[2 * i + 1]);
                }
            }
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] += __shfl_xor(acc[i], 16, 64);
                acc[i] += __shfl_xor(acc[i], 32, 64);
            }
            if (sub == 0) {
                const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] *= inv;
                store_vec<bf16>(reinterpret_cast<bf16*>(xs_att) + (b * la.q_len + t) * GV_KC + head * D + c * 8, acc);
            }
        }
    }
}

// The body is a device function of the workgroup index so that two GEMVs reading the same x can share ONE launch
// (gemv_qkv_h4_kernel below): workgroups [0, nq) run one instantiation, the rest the other.
// ATT = U > 0 (round 4): the second operand pair's x rows (the attention output) are not read from memory but computed here, by
// every workgroup, from the local KV window (see LocalAtt; U = row batches of 4 keys, 2 or 3): wave w takes the (sequence, head)
// items w, w + 4, ... with exactly local_attend_kernel's lane mapping and arithmetic (attention.hip -- 16 lanes x 16 B per row, the
// same fmaf / shuffle order, bf16 rounding of the result), so the fused layer is bit-identical to attention launch + GEMV.  Its
// loads are issued right behind the first two chunks of W rows and it is evaluated in their shadow; the result sits in LDS until
// the last chunk consumes it: the layer loses a dependent launch (~5 us at batch 1) for ~1 us of prologue.
template <int MR, int RW, bool LN = false, bool QKV = false, int ATT = 0>
__device__ __forceinline__ void gemv_body(const int bid, const bf16* __restrict__ X, int64_t ldx, const bf16* __restrict__ W,
                                          int64_t ldw, const EpiArgs& ep, int K, const LnPro& ln, const Seg2& s2,
                                          const LocalAtt& la = LocalAtt{}) {
    static_assert(!QKV || RW == 2, "QKV mode works on row pairs");
    static_assert(ATT == 0 || (!LN && !QKV && RW == 1), "the attention producer feeds the residual-stream GEMV");
    __shared__ __attribute__((aligned(16))) char xs[MR * GV_KC * 2];
    __shared__ __attribute__((aligned(16))) char xs_att[ATT ? MR * GV_KC * 2 : 16];
    __shared__ float red[2][4][MR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = ep.M, N = ep.N;
    int rows[RW];
    if (QKV) {   // pair p = (32-column group p >> 4, column p & 15 of it) and its rotate-half partner 16 columns further
        const int p = bid * 4 + wave;
        rows[0] = (p >> 4) * 32 + (p & 15);
        rows[RW - 1] = rows[0] + 16;
    } else {
        const int n_blk = bid * (4 * RW);
#pragma unroll
        for (int r = 0; r < RW; ++r) rows[r] = n_blk + wave + 4 * r;   // consecutive rows go to different waves
    }
    float acc[RW][MR];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int m = 0; m < MR; ++m) acc[r][m] = 0.f;

    const int k_all = K + ((LN || QKV) ? 0 : s2.K);   // K chunks of the first operand pair, then of the second
    // Load pipeline.  A wave keeps one chunk of its W rows in registers (RW x 4 x 16 bytes per lane).  With several chunks
    // (the residual-stream projections: K = 8192 + 2048 -> five) a single buffer pays a full HBM latency per chunk with
    // only 8 waves per CU in flight (measured 3.6 TB/s); PIPE double-buffers: the NEXT chunk's rows are issued before this
    // chunk is consumed.  Within a chunk the x pieces are loaded FIRST and the W rows right behind them: vmcnt retires in
    // order, so waiting for the (older, L2-resident) x pieces leaves the W stream in flight, and the LDS staging and the
    // LayerNorm prologue run in its shadow.  Summation order is unchanged -> bit-identical results.
    constexpr bool PIPE = !LN && !QKV && RW < 4;
    constexpr int NBUF = PIPE ? 2 : 1;
    uint4 wv[NBUF][RW][GV_KC / 512];
    // `live` = false (no next chunk): the same number of loads is issued from a harmless L2-resident address (the first x
    // row) -- a branch around the prefetch would make the compiler merge both paths' vmcnt states and wait for the prefetch
    auto load_w = [&](auto BUF, int kg_req, bool live) {
        constexpr int buf = decltype(BUF)::value;
        const int kg = live ? kg_req : 0;
        const bool second = kg >= K;                   // (K is a multiple of GV_KC whenever a second segment follows)
        const int kc = second ? kg - K : kg;
        const bf16* Ws = second ? s2.w : W;
        const int64_t ldws = second ? s2.ldw : ldw;
        const int kw = min(GV_KC, (second ? s2.K : K) - kc);   // multiple of 512
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int n = rows[r] < N ? rows[r] : N - 1;
            const bf16* wp = (live ? Ws + (int64_t)n * ldws : X) + kc + lane * 8;
            // every load is issued unconditionally (pieces past a short last chunk re-read its last valid piece and are
            // never multiplied): no branch around a load, so the compiler's vmcnt bookkeeping stays exact
#pragma unroll
            for (int j = 0; j < GV_KC / 512; ++j)
                wv[buf][r][j] = load_w16<(RW < 4)>(wp + min(j * 512, kw - 512));   // RW == 4: the logits matrix, re-read every token -> cacheable
        }
    };
    auto chunk = [&](auto BUF, int kg) {
        constexpr int buf = decltype(BUF)::value;
        const bool second = kg >= K;
        const int kc = second ? kg - K : kg;
        const bf16* Xs = second ? s2.x : X;
        const int64_t ldxs = second ? s2.ldx : ldx;
        const int kw = min(GV_KC, (second ? s2.K : K) - kc);
        // thread t owns the 16-byte piece t of every x row of the chunk (kw / 8 <= 256 pieces per row)
        const bool own = tid < kw / 8;
        const bool from_att = ATT && second;           // this chunk's x rows were produced into xs_att by the prologue below
        uint4 xv[MR];
#pragma unroll
        for (int m = 0; m < MR; ++m)   // unconditional (clamped) loads, see load_w; rows >= M are zeroed when staged
            xv[m] = *reinterpret_cast<const uint4*>((from_att ? X : Xs + kc) + (int64_t)min(m, M - 1) * (from_att ? ldx : ldxs) +
                                                    (own ? tid : 0) * 8);
        // ATT: this chunk's share of the wave's attention items -- loads BEFORE the weight rows, arithmetic behind them
        LocalAttRegs<(ATT > 0 ? ATT : 1)> ar;
        int aj0 = 0, aj1 = 0, a_items = 0;
        if constexpr (ATT > 0) {
            if (!second) {
                a_items = (M / la.q_len) * la.heads;
                const int ipw = (a_items + 3) / 4, nfc = K / GV_KC;
                const int ipc = (ipw + nfc - 1) / nfc;       // <= ATT_HPW (few_rows_attn_res_ok)
                aj0 = min((kg / GV_KC) * ipc, ipw);
                aj1 = min(aj0 + ipc, ipw);
                if (aj1 > aj0) local_att_issue<ATT>(la, a_items, aj0, aj1, lane, wave, ar);
                if (kg == 0) {   // rows >= M (and columns >= heads * 128) of the staged operand are zero: the GEMV multiplies all MR rows
                    for (int i = tid * 8; i < MR * GV_KC; i += 256 * 8)
                        if (i >= M * GV_KC || (i % GV_KC) >= la.heads * 128) *reinterpret_cast<uint4*>(xs_att + i * 2) = uint4{0, 0, 0, 0};
                }
            }
        }
        if (!PIPE || kg == 0) load_w(BUF, kg, true);                                    // this chunk's rows (first chunk / no pipeline)
        if (PIPE) load_w(std::integral_constant<int, (buf + 1) % NBUF>{}, kg + GV_KC, kg + GV_KC < k_all);   // next chunk's rows
        if constexpr (ATT > 0) {
            if (!second && aj1 > aj0) local_att_finish<ATT>(la, a_items, aj0, aj1, lane, wave, ar, xs_att);
        }
        __syncthreads();                               // previous chunk fully consumed
        if (own && !from_att) {
#pragma unroll
            for (int m = 0; m < MR; ++m)
                *reinterpret_cast<uint4*>(xs + (m * GV_KC + tid * 8) * 2) = m < M ? xv[m] : uint4{0, 0, 0, 0};
        }
        __syncthreads();
        const char* xsrc = from_att ? xs_att : xs;
        if (LN) {   // K <= GV_KC: the chunk is the whole row.  Thread t owns the 16-byte piece t of every row.
            float v[MR][8];
            float part[MR];
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                part[m] = 0.f;
                if (own) {
                    load_vec<bf16>(reinterpret_cast<const bf16*>(xs + (m * GV_KC + tid * 8) * 2), v[m]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) part[m] += v[m][i];
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[m][i] = 0.f;
                }
                part[m] = wave_sum(part[m]);
                if (lane == 0) red[0][wave][m] = part[m];
            }
            __syncthreads();
            float mean[MR];
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                mean[m] = (red[0][0][m] + red[0][1][m] + red[0][2][m] + red[0][3][m]) / (float)kw;
                float sq = 0.f;
                if (own) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float d = v[m][i] - mean[m];
                        sq = fmaf(d, d, sq);
                    }
                }
                sq = wave_sum(sq);
                if (lane == 0) red[1][wave][m] = sq;
            }
            __syncthreads();
            if (own) {
                float gam[8], bet[8];
                load_vec<bf16>(ln.w + tid * 8, gam);
                load_vec<bf16>(ln.b + tid * 8, bet);
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    const float rstd = rsqrtf((red[1][0][m] + red[1][1][m] + red[1][2][m] + red[1][3][m]) / (float)kw + ln.eps);
                    float y[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) y[i] = (v[m][i] - mean[m]) * rstd * gam[i] + bet[i];
                    store_vec<bf16>(reinterpret_cast<bf16*>(xs + (m * GV_KC + tid * 8) * 2), y);   // bf16, as bt_layernorm writes it
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < GV_KC / 512; ++j) {
            if (j * 512 < kw) {
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    const uint4 xq = *reinterpret_cast<const uint4*>(xsrc + (m * GV_KC + j * 512 + lane * 8) * 2);
#pragma unroll
                    for (int r = 0; r < RW; ++r) acc[r][m] = dot8_bf16(wv[buf][r][j], xq, acc[r][m]);
                }
            }
        }
    };
    for (int kg = 0; kg < k_all; kg += NBUF * GV_KC) {
        chunk(std::integral_constant<int, 0>{}, kg);
        if (PIPE && kg + GV_KC < k_all) chunk(std::integral_constant<int, NBUF - 1>{}, kg + GV_KC);
    }
    // finish the dot products; afterwards every lane holds every sum
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int m = 0; m < MR; ++m) acc[r][m] = wave_sum(acc[r][m]);
    if (QKV) {   // lane m finishes row m: bias, rotate the (n, n + 16) pair when it lies in dims 0..31 of a q / k head, scatter
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int m = 0; m < MR; ++m)
            if (lane == m) {
                v0 = acc[0][m];
                v1 = acc[RW - 1][m];
            }
        const int n0 = rows[0], n1 = rows[RW - 1];
        if (lane < M && n1 < N) {
            const int m = lane;
            if (ep.bias) {
                v0 += bf16_to_f32(ep.bias[n0].x);
                v1 += bf16_to_f32(ep.bias[n1].x);
            }
            const int D = ep.q.head_dim;
            const int head = n0 / (3 * D);
            const int within = n0 - head * 3 * D;
            const int part = within / D;              // 0 q, 1 k, 2 v
            const int el = within - part * D;         // n1 holds el + 16 of the same part (D % 32 == 0)
            const int64_t b = m / ep.q.q_len;
            const int t = m - (int)b * ep.q.q_len;
            const int pos0 = ep.q.pos0_dev ? *ep.q.pos0_dev : ep.q.pos0_host;
            if (part < 2 && el < 16) {
                const int rpos = ep.q.pos_table ? ep.q.pos_table[pos0 + t] : pos0 + t;
                const float c = ep.q.cos_tab[(int64_t)rpos * 16 + el], sn = ep.q.sin_tab[(int64_t)rpos * 16 + el];
                const float x1 = v0, x2 = v1;
                v0 = x1 * c - x2 * sn;
                v1 = x2 * c + x1 * sn;
            }
            bf16* dst = part == 0 ? ep.out + (int64_t)m * ep.ldo + n0
                                  : (part == 1 ? ep.q.kc : ep.q.vc) + ((b * ep.q.heads + head) * ep.q.n_max + pos0 + t) * D + el;
            dst[0].x = f32_to_bf16(v0);
            dst[16].x = f32_to_bf16(v1);
        }
        return;
    }
    // lane (r * MR + m) stores element (m, rows[r])
    float mine = 0.f;
    int my_r = 0, my_m = 0;
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int m = 0; m < MR; ++m)
            if (lane == r * MR + m) {
                mine = acc[r][m];
                my_r = r;
                my_m = m;
            }
    if (lane < RW * MR) {
        int n = 0;
#pragma unroll
        for (int r = 0; r < RW; ++r)
            if (my_r == r) n = rows[r];
        if (my_m < M && n < N) {
            float v = mine;
            if (ep.bias) v += bf16_to_f32(ep.bias[n].x);
            if (s2.bias) v += bf16_to_f32(s2.bias[n].x);
            if (ep.act == BT_ACT_GELU) v = gelu_tail2(v2f{v, v}).x;
            if (ep.res) v += bf16_to_f32(ep.res[(int64_t)my_m * ep.ldr + n].x);
            ep.out[(int64_t)my_m * ep.ldo + n].x = f32_to_bf16(v);
        }
    }
}

template <int MR, int RW, bool LN = false, bool QKV = false>
__global__ __launch_bounds__(256) void gemv_bf16_kernel(const bf16* __restrict__ X, int64_t ldx, const bf16* __restrict__ W,
                                                       int64_t ldw, const EpiArgs ep, int K, const LnPro ln,
                                                       const Seg2 s2 = Seg2{}) {
    gemv_body<MR, RW, LN, QKV>(blockIdx.x, X, ldx, W, ldw, ep, K, ln, s2);
}

#ifdef BT_DEBUG_HOOKS   // measured, not adopted (see g_few_attn): A/B and bit-identity test only
// x += h4 . W2^T + b2 + attention(local window) . Wd^T + bd : the attention output never exists in memory (gemv_body<..., ATT = U>)
template <int MR, int U>
__global__ __launch_bounds__(256) void gemv_attn_res_kernel(const bf16* __restrict__ X, int64_t ldx, const bf16* __restrict__ W,
                                                           int64_t ldw, const EpiArgs ep, int K, const Seg2 s2, const LocalAtt la) {
    gemv_body<MR, 1, false, false, U>(blockIdx.x, X, ldx, W, ldw, ep, K, LnPro{}, s2, la);
}
#endif

// Parallel residual: the layer's QKV projection (LN1 prologue, RoPE + KV append) and its MLP up-projection (LN2 prologue, GELU)
// both read the SAME x and do not depend on each other -> one launch streams both weight matrices (58 MB at 1.2B dims)
// instead of two dependent ~9 us launches; the decoder layer is then 3 launches for <= 4 rows:
// [QKV | h->4h] -> attention -> [4h->h + attention.dense + residual].
template <int MR>
__global__ __launch_bounds__(256) void gemv_qkv_h4_kernel(const bf16* __restrict__ X, int64_t ldx, const bf16* __restrict__ Wq,
                                                         const EpiArgs epq, const LnPro ln1, int nq, const bf16* __restrict__ Wh,
                                                         const EpiArgs eph, const LnPro ln2, int K) {
    if ((int)blockIdx.x < nq) gemv_body<MR, 2, true, true>(blockIdx.x, X, ldx, Wq, K, epq, K, ln1, Seg2{});
    else gemv_body<MR, 2, true, false>(blockIdx.x - nq, X, ldx, Wh, K, eph, K, ln2, Seg2{});
}

// ---- persistent form of the [QKV | h->4h] GEMV launch (round 4) ---------------------------------------------------------------
// gemv_qkv_h4_kernel starts 1792 workgroups of 32 KiB of weights each (1.2B dims), and every one of them stages x and runs the
// LayerNorm prologue (an L2 round trip for x, three block reductions, gamma / beta loads: ~3 us) in front of ~1 us of streaming: seven
// prologues per CU.  Here ~2 workgroups per CU stay resident: x is staged and normalised ONCE per workgroup, then the workgroup walks
// its row groups with the next group's weight rows in flight (second register buffer) while the current group is multiplied, reduced
// and written -- no barrier after the prologue, the waves free-run.  Row-for-row the arithmetic of gemv_body<MR, 2, true, QKV>
// (same piece ownership, same LayerNorm, same dot / butterfly order, same epilogues): bit-identical results
// (tests/test_model_gpu.py::test_few_rows_persistent_gemv_is_bit_identical).
template <int MR, bool QKV>
__device__ __forceinline__ void gemv_ln_persist_body(const int bid0, const int stride, const int n_groups,
                                                     const bf16* __restrict__ X, int64_t ldx, const bf16* __restrict__ W, int64_t ldw,
                                                     const EpiArgs& ep, int K, const LnPro& ln) {
    constexpr int RW = 2;
    __shared__ __attribute__((aligned(16))) char xs[MR * GV_KC * 2];
    __shared__ float red[2][4][MR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = ep.M, N = ep.N;
    const int kw = K;                                  // K <= GV_KC, multiple of 512: the one chunk is the whole row
    const bool own = tid < kw / 8;
    auto rows_of = [&](int bid, int (&rows)[RW]) {
        if (QKV) {
            const int p = bid * 4 + wave;
            rows[0] = (p >> 4) * 32 + (p & 15);
            rows[1] = rows[0] + 16;
        } else {
            const int n_blk = bid * (4 * RW);
            rows[0] = n_blk + wave;
            rows[1] = n_blk + wave + 4;
        }
    };
    uint4 wv[2][RW][GV_KC / 512];
    auto load_w = [&](auto BUF, int bid, bool live) {   // unconditional loads (see gemv_body::load_w)
        constexpr int buf = decltype(BUF)::value;
        int rows[RW];
        rows_of(live ? bid : 0, rows);
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int n = rows[r] < N ? rows[r] : N - 1;
            const bf16* wp = (live ? W + (int64_t)n * ldw : X) + lane * 8;
#pragma unroll
            for (int j = 0; j < GV_KC / 512; ++j) wv[buf][r][j] = load_w16<true>(wp + min(j * 512, kw - 512));
        }
    };
    // ---- prologue: x pieces first, the first group's rows right behind them, LayerNorm in their shadow ----
    uint4 xv[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) xv[m] = *reinterpret_cast<const uint4*>(X + (int64_t)min(m, M - 1) * ldx + (own ? tid : 0) * 8);
    load_w(std::integral_constant<int, 0>{}, bid0, bid0 < n_groups);
    if (own) {
#pragma unroll
        for (int m = 0; m < MR; ++m) *reinterpret_cast<uint4*>(xs + (m * GV_KC + tid * 8) * 2) = m < M ? xv[m] : uint4{0, 0, 0, 0};
    }
    __syncthreads();
    {
        float v[MR][8];
        float part[MR];
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            part[m] = 0.f;
            if (own) {
                load_vec<bf16>(reinterpret_cast<const bf16*>(xs + (m * GV_KC + tid * 8) * 2), v[m]);
#pragma unroll
                for (int i = 0; i < 8; ++i) part[m] += v[m][i];
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[m][i] = 0.f;
            }
            part[m] = wave_sum(part[m]);
            if (lane == 0) red[0][wave][m] = part[m];
        }
        __syncthreads();
        float mean[MR];
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            mean[m] = (red[0][0][m] + red[0][1][m] + red[0][2][m] + red[0][3][m]) / (float)kw;
            float sq = 0.f;
            if (own) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float d = v[m][i] - mean[m];
                    sq = fmaf(d, d, sq);
                }
            }
            sq = wave_sum(sq);
            if (lane == 0) red[1][wave][m] = sq;
        }
        __syncthreads();
        if (own) {
            float gam[8], bet[8];
            load_vec<bf16>(ln.w + tid * 8, gam);
            load_vec<bf16>(ln.b + tid * 8, bet);
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                const float rstd = rsqrtf((red[1][0][m] + red[1][1][m] + red[1][2][m] + red[1][3][m]) / (float)kw + ln.eps);
                float y[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) y[i] = (v[m][i] - mean[m]) * rstd * gam[i] + bet[i];
                store_vec<bf16>(reinterpret_cast<bf16*>(xs + (m * GV_KC + tid * 8) * 2), y);
            }
        }
        __syncthreads();
    }
    // ---- row groups: xs is read-only from here on, no barrier ----
    auto group = [&](auto BUF, int bid) {
        constexpr int buf = decltype(BUF)::value;
        load_w(std::integral_constant<int, 1 - buf>{}, bid + stride, bid + stride < n_groups);   // next group's rows
        int rows[RW];
        rows_of(bid, rows);
        float acc[RW][MR];
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int m = 0; m < MR; ++m) acc[r][m] = 0.f;
#pragma unroll
        for (int j = 0; j < GV_KC / 512; ++j) {
            if (j * 512 < kw) {
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    const uint4 xq = *reinterpret_cast<const uint4*>(xs + (m * GV_KC + j * 512 + lane * 8) * 2);
#pragma unroll
                    for (int r = 0; r < RW; ++r) acc[r][m] = dot8_bf16(wv[buf][r][j], xq, acc[r][m]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int m = 0; m < MR; ++m) acc[r][m] = wave_sum(acc[r][m]);
        if (QKV) {
            float v0 = 0.f, v1 = 0.f;
#pragma unroll
            for (int m = 0; m < MR; ++m)
                if (lane == m) {
                    v0 = acc[0][m];
                    v1 = acc[RW - 1][m];
                }
            const int n0 = rows[0], n1 = rows[RW - 1];
            if (lane < M && n1 < N) {
                const int m = lane;
                if (ep.bias) {
                    v0 += bf16_to_f32(ep.bias[n0].x);
                    v1 += bf16_to_f32(ep.bias[n1].x);
                }
                const int D = ep.q.head_dim;
                const int head = n0 / (3 * D);
                const int within = n0 - head * 3 * D;
                const int part = within / D;
                const int el = within - part * D;
                const int64_t b = m / ep.q.q_len;
                const int t = m - (int)b * ep.q.q_len;
                const int pos0 = ep.q.pos0_dev ? *ep.q.pos0_dev : ep.q.pos0_host;
                if (part < 2 && el < 16) {
                    const int rpos = ep.q.pos_table ? ep.q.pos_table[pos0 + t] : pos0 + t;
                    const float c = ep.q.cos_tab[(int64_t)rpos * 16 + el], sn = ep.q.sin_tab[(int64_t)rpos * 16 + el];
                    const float x1 = v0, x2 = v1;
                    v0 = x1 * c - x2 * sn;
                    v1 = x2 * c + x1 * sn;
                }
                bf16* dst = part == 0 ? ep.out + (int64_t)m * ep.ldo + n0
                                      : (part == 1 ? ep.q.kc : ep.q.vc) + ((b * ep.q.heads + head) * ep.q.n_max + pos0 + t) * D + el;
                dst[0].x = f32_to_bf16(v0);
                dst[16].x = f32_to_bf16(v1);
            }
        } else {
            float mine = 0.f;
            int my_r = 0, my_m = 0;
#pragma unroll
            for (int r = 0; r < RW; ++r)
#pragma unroll
                for (int m = 0; m < MR; ++m)
                    if (lane == r * MR + m) {
                        mine = acc[r][m];
                        my_r = r;
                        my_m = m;
                    }
            if (lane < RW * MR) {
                const int n = my_r == 0 ? rows[0] : rows[1];
                if (my_m < M && n < N) {
                    float v = mine;
                    if (ep.bias) v += bf16_to_f32(ep.bias[n].x);
                    if (ep.act == BT_ACT_GELU) v = gelu_tail2(v2f{v, v}).x;
                    if (ep.res) v += bf16_to_f32(ep.res[(int64_t)my_m * ep.ldr + n].x);
                    ep.out[(int64_t)my_m * ep.ldo + n].x = f32_to_bf16(v);
                }
            }
        }
    };
    for (int bid = bid0; bid < n_groups; bid += 2 * stride) {
        group(std::integral_constant<int, 0>{}, bid);
        if (bid + stride < n_groups) group(std::integral_constant<int, 1>{}, bid + stride);
    }
}

// workgroups [0, gq) walk the nq QKV row groups, the others the nh h->4h row groups
template <int MR>
__global__ __launch_bounds__(256) void gemv_qkv_h4_persist_kernel(const bf16* __restrict__ X, int64_t ldx, const bf16* __restrict__ Wq,
                                                                 const EpiArgs epq, const LnPro ln1, int nq, int gq,
                                                                 const bf16* __restrict__ Wh, const EpiArgs eph, const LnPro ln2,
                                                                 int nh, int K) {
    if ((int)blockIdx.x < gq) gemv_ln_persist_body<MR, true>(blockIdx.x, gq, nq, X, ldx, Wq, K, epq, K, ln1);
    else gemv_ln_persist_body<MR, false>(blockIdx.x - gq, gridDim.x - gq, nh, X, ldx, Wh, K, eph, K, ln2);
}

// ---- MFMA weight streamer for 5 .. 64 rows (mid-batch decode) ----------------------------------------------------------------
// Same contract as gemv_bf16_kernel (one launch, W streamed once at HBM speed, LayerNorm / RoPE+append / second operand
// pair folded in), but the products run on the matrix pipe -- at 8+ rows the GEMV's per-row VALU work is slower than the
// weight stream.  A workgroup (8 waves) owns 32 rows of W over the whole K; wave w multiplies K slice w: per 32-wide
// k-step it loads two W fragments (rows n0 .. n0+15 / +16 .. +31; lane = row (lane & 15), 16-byte k-slot (lane >> 4) --
// 64 contiguous bytes per row) and MT x fragments straight from global memory (x is a few KiB per row and lives in L2)
// into MFMA operand registers, no LDS.  The 8 partial 32 x M tiles are summed through LDS, and wave mt finishes the 16
// columns m = 16 mt .. of both row tiles (so the rotate-half partner n + 16 is in the same lane).  LN: per-row mean / rstd
// are computed first (wave per row), x fragments are normalised in registers as they arrive.
constexpr int SK_ROWS = 32;

template <int MT>
struct SkinnyLds {
    float red[8][2][MT][64][4];
    float stats[MT * 16][2];
};

template <int MT, bool LN, bool QKV>
__device__ __forceinline__ void skinny_body(const int bid, SkinnyLds<MT>& lds, const bf16* __restrict__ X, int64_t ldx,
                                            const bf16* __restrict__ W, int64_t ldw, const EpiArgs& ep, int K, const LnPro& ln,
                                            const Seg2& s2) {
    auto& red = lds.red;
    auto& stats = lds.stats;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = ep.M, N = ep.N;
    const int n0 = bid * SK_ROWS;
    const int l15 = lane & 15, g = lane >> 4;

    if (LN) {   // a wave per row: two passes over the row (K * 2 bytes, L2-resident)
        for (int m = wave; m < M; m += 8) {
            const bf16* xr = X + (int64_t)m * ldx;
            float sum = 0.f;
            for (int c = lane * 8; c < K; c += 512) {
                float v[8];
                load_vec<bf16>(xr + c, v);
#pragma unroll
                for (int i = 0; i < 8; ++i) sum += v[i];
            }
            const float mean = wave_sum(sum) / (float)K;
            float sq = 0.f;
            for (int c = lane * 8; c < K; c += 512) {
                float v[8];
                load_vec<bf16>(xr + c, v);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float d = v[i] - mean;
                    sq = fmaf(d, d, sq);
                }
            }
            const float rstd = rsqrtf(wave_sum(sq) / (float)K + ln.eps);
            if (lane == 0) {
                stats[m][0] = mean;
                stats[m][1] = rstd;
            }
        }
        __syncthreads();
    }
    float mean[MT], rstd[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + l15;
        mean[mt] = (LN && m < M) ? stats[m][0] : 0.f;
        rstd[mt] = (LN && m < M) ? stats[m][1] : 0.f;
    }

    f32x4_t acc[2][MT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // k-steps of the first operand pair, then of the second; wave w takes a contiguous range of them
    const int nks1 = K / 32, nks = nks1 + ((LN || QKV) ? 0 : s2.K / 32);
    const int per = (nks + 7) / 8;
    const int ks_lo = wave * per, ks_hi = min(nks, ks_lo + per);
    int wr[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = n0 + t * 16 + l15;
        wr[t] = n < N ? n : N - 1;
    }
    constexpr int U = 2;
    for (int ks = ks_lo; ks < ks_hi; ks += U) {
        uint4 wfr[U][2], xfr[U][MT], gam[U], bet[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kq = min(ks + u, ks_hi - 1);           // a repeated last step contributes nothing (see below)
            const bool second = kq >= nks1;
            const int k0 = (second ? kq - nks1 : kq) * 32 + g * 8;
            const bf16* Ws = second ? s2.w : W;
            const bf16* Xs = second ? s2.x : X;
            const int64_t ldws = second ? s2.ldw : ldw, ldxs = second ? s2.ldx : ldx;
#pragma unroll
            // (a runtime non-temporal / default select here made hipcc branch around every load: B = 8 6.7k -> 5.2k tok/s; the
            // streamer keeps default-policy loads, the GEMV picks the policy at compile time)
            for (int t = 0; t < 2; ++t) wfr[u][t] = load_w16<false>(Ws + (int64_t)wr[t] * ldws + k0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = mt * 16 + l15;
                xfr[u][mt] = (m < M && ks + u < ks_hi) ? *reinterpret_cast<const uint4*>(Xs + (int64_t)m * ldxs + k0)
                                                        : uint4{0, 0, 0, 0};
            }
            if (LN) {
                gam[u] = *reinterpret_cast<const uint4*>(ln.w + k0);
                bet[u] = *reinterpret_cast<const uint4*>(ln.b + k0);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                uint4 xv = xfr[u][mt];
                if (LN) {   // (x - mean) * rstd * gamma + beta, rounded to bf16 as bt_layernorm writes it
                    const int m = mt * 16 + l15;
                    const uint32_t xi[4] = {xv.x, xv.y, xv.z, xv.w};
                    const uint32_t gi[4] = {gam[u].x, gam[u].y, gam[u].z, gam[u].w};
                    const uint32_t bi[4] = {bet[u].x, bet[u].y, bet[u].z, bet[u].w};
                    uint32_t yo[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float a = (__uint_as_float(xi[i] << 16) - mean[mt]) * rstd[mt] * __uint_as_float(gi[i] << 16) +
                                        __uint_as_float(bi[i] << 16);
                        const float b = (__uint_as_float(xi[i] & 0xffff0000u) - mean[mt]) * rstd[mt] *
                                            __uint_as_float(gi[i] & 0xffff0000u) +
                                        __uint_as_float(bi[i] & 0xffff0000u);
                        yo[i] = pack_bf16x2(a, b);
                    }
                    const bool live = m < M && ks + u < ks_hi;
                    xv = live ? uint4{yo[0], yo[1], yo[2], yo[3]} : uint4{0, 0, 0, 0};
                }
                union {
                    uint4 q;
                    bf16x8_t v;
                } xb, wb0, wb1;
                xb.q = xv;
                wb0.q = wfr[u][0];
                wb1.q = wfr[u][1];
                acc[0][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb0.v, xb.v, acc[0][mt], 0, 0, 0);
                acc[1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb1.v, xb.v, acc[1][mt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<f32x4_t*>(red[wave][t][mt][lane]) = acc[t][mt];
    __syncthreads();
    if (wave >= MT) return;
    const int mt = wave;
    f32x4_t v[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        v[t] = *reinterpret_cast<const f32x4_t*>(red[0][t][mt][lane]);
#pragma unroll
        for (int w = 1; w < 8; ++w) {
            const f32x4_t p = *reinterpret_cast<const f32x4_t*>(red[w][t][mt][lane]);
            v[t][0] += p[0]; v[t][1] += p[1]; v[t][2] += p[2]; v[t][3] += p[3];
        }
    }
    // lane: row m of x, columns n0 + t * 16 + g * 4 .. + 3 of the output
    const int m = mt * 16 + l15;
    if (m >= M) return;
    float o[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = n0 + t * 16 + g * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[t][r] = v[t][r];
        if (n < N) {
            if (ep.bias) {
                const uint2 b2 = *reinterpret_cast<const uint2*>(ep.bias + n);
                o[t][0] += __uint_as_float(b2.x << 16); o[t][1] += __uint_as_float(b2.x & 0xffff0000u);
                o[t][2] += __uint_as_float(b2.y << 16); o[t][3] += __uint_as_float(b2.y & 0xffff0000u);
            }
            if (s2.bias) {
                const uint2 b2 = *reinterpret_cast<const uint2*>(s2.bias + n);
                o[t][0] += __uint_as_float(b2.x << 16); o[t][1] += __uint_as_float(b2.x & 0xffff0000u);
                o[t][2] += __uint_as_float(b2.y << 16); o[t][3] += __uint_as_float(b2.y & 0xffff0000u);
            }
        }
    }
    if (QKV) {
        const int D = ep.q.head_dim;
        const int head = n0 / (3 * D);
        const int within = n0 - head * 3 * D;
        const int part = within / D;              // the 32-column group n0 .. n0+31 lies in one part (D % 32 == 0)
        const int el0 = within - part * D;
        const int64_t b = m / ep.q.q_len;
        const int tq = m - (int)b * ep.q.q_len;
        const int pos0 = ep.q.pos0_dev ? *ep.q.pos0_dev : ep.q.pos0_host;
        if (part < 2 && el0 == 0) {               // dims 0..31 of a q / k head: column e pairs with e + 16 (the other row tile)
            const int rpos = ep.q.pos_table ? ep.q.pos_table[pos0 + tq] : pos0 + tq;
            const float4 c4 = *reinterpret_cast<const float4*>(ep.q.cos_tab + (int64_t)rpos * 16 + g * 4);
            const float4 s4 = *reinterpret_cast<const float4*>(ep.q.sin_tab + (int64_t)rpos * 16 + g * 4);
            const float c[4] = {c4.x, c4.y, c4.z, c4.w}, sn[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x1 = o[0][r], x2 = o[1][r];
                o[0][r] = x1 * c[r] - x2 * sn[r];
                o[1][r] = x2 * c[r] + x1 * sn[r];
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int n = n0 + t * 16 + g * 4;
            if (n >= N) continue;
            const int el = el0 + t * 16 + g * 4;
            bf16* dst = part == 0 ? ep.out + (int64_t)m * ep.ldo + n
                                  : (part == 1 ? ep.q.kc : ep.q.vc) + ((b * ep.q.heads + head) * ep.q.n_max + pos0 + tq) * D + el;
            uint2 pk;
            pk.x = pack_bf16x2(o[t][0], o[t][1]);
            pk.y = pack_bf16x2(o[t][2], o[t][3]);
            *reinterpret_cast<uint2*>(dst) = pk;
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = n0 + t * 16 + g * 4;
        if (n >= N) continue;
        if (ep.act == BT_ACT_GELU) gelu4(o[t]);
        if (ep.res) {
            const uint2 r2 = *reinterpret_cast<const uint2*>(ep.res + (int64_t)m * ep.ldr + n);
            o[t][0] += __uint_as_float(r2.x << 16); o[t][1] += __uint_as_float(r2.x & 0xffff0000u);
            o[t][2] += __uint_as_float(r2.y << 16); o[t][3] += __uint_as_float(r2.y & 0xffff0000u);
        }
        uint2 pk;
        pk.x = pack_bf16x2(o[t][0], o[t][1]);
        pk.y = pack_bf16x2(o[t][2], o[t][3]);
        *reinterpret_cast<uint2*>(ep.out + (int64_t)m * ep.ldo + n) = pk;
    }
}

template <int MT, bool LN, bool QKV>
__global__ __launch_bounds__(512) void skinny_mfma_kernel(const bf16* __restrict__ X, int64_t ldx, const bf16* __restrict__ W,
                                                         int64_t ldw, const EpiArgs ep, int K, const LnPro ln, const Seg2 s2) {
    __shared__ __attribute__((aligned(16))) SkinnyLds<MT> lds;
    skinny_body<MT, LN, QKV>(blockIdx.x, lds, X, ldx, W, ldw, ep, K, ln, s2);
}

// 5 .. 16 rows: the layer's QKV (LN1, RoPE, append) and h->4h (LN2, GELU) projections as ONE launch, like gemv_qkv_h4_kernel
template <int MT>
__global__ __launch_bounds__(512) void skinny_qkv_h4_kernel(const bf16* __restrict__ X, int64_t ldx, const bf16* __restrict__ Wq,
                                                           const EpiArgs epq, const LnPro ln1, int nq,
                                                           const bf16* __restrict__ Wh, const EpiArgs eph, const LnPro ln2, int K) {
    __shared__ __attribute__((aligned(16))) SkinnyLds<MT> lds;
    if ((int)blockIdx.x < nq) skinny_body<MT, true, true>(blockIdx.x, lds, X, ldx, Wq, K, epq, K, ln1, Seg2{});
    else skinny_body<MT, true, false>(blockIdx.x - nq, lds, X, ldx, Wh, K, eph, K, ln2, Seg2{});
}

// launch helper: MT = ceil(M / 16) in 1..4
template <bool LN, bool QKV>
static void launch_skinny(hipStream_t s, int64_t M, int64_t N, const bf16* X, int64_t ldx, const bf16* W, int64_t ldw,
                          const EpiArgs& ep, int64_t K, const LnPro& ln, const Seg2& s2) {
    const dim3 grid((unsigned)ceil_div(N, SK_ROWS));
    const int mt = (int)ceil_div(M, 16);
    if (mt <= 1) hipLaunchKernelGGL((skinny_mfma_kernel<1, LN, QKV>), grid, dim3(512), 0, s, X, ldx, W, ldw, ep, (int)K, ln, s2);
    else if (mt == 2) hipLaunchKernelGGL((skinny_mfma_kernel<2, LN, QKV>), grid, dim3(512), 0, s, X, ldx, W, ldw, ep, (int)K, ln, s2);
    else if (mt == 3) hipLaunchKernelGGL((skinny_mfma_kernel<3, LN, QKV>), grid, dim3(512), 0, s, X, ldx, W, ldw, ep, (int)K, ln, s2);
    else hipLaunchKernelGGL((skinny_mfma_kernel<4, LN, QKV>), grid, dim3(512), 0, s, X, ldx, W, ldw, ep, (int)K, ln, s2);
}

// common preconditions of the streamer (beyond M <= 64): 8-byte output / bias / residual groups, 16-byte operand pieces
static bool skinny_ok(int64_t M, int64_t N, int64_t K, const void* x, int64_t ldx, const void* W, const void* bias,
                      const void* residual, int64_t ldr, const void* out, int64_t ldo) {
    return M >= 1 && M <= 64 && N % 4 == 0 && K % 32 == 0 && ldx % 8 == 0 && aligned16(x) && aligned16(W) && ldo % 4 == 0 &&
           (reinterpret_cast<uintptr_t>(out) & 7u) == 0 && (!bias || (reinterpret_cast<uintptr_t>(bias) & 7u) == 0) &&
           (!residual || (ldr % 4 == 0 && (reinterpret_cast<uintptr_t>(residual) & 7u) == 0));
}

// ---- generic tiled kernel (fp32 parity path; any M, N, K, strides) --------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gemm_generic_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ W,
                                                          int64_t ldw, const T* __restrict__ bias,
                                                          const T* res, int64_t ldr, T* out,
                                                          int64_t ldo, int M, int N, int K, int act) {
    __shared__ float As[16][65];
    __shared__ float Ws[16][65];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * 256;
            const int r = idx >> 4, kk = idx & 15;
            const int k = k0 + kk;
            float a = 0.f, w = 0.f;
            if (k < K) {
                if (m0 + r < M) a = Elem<T>::load(A + (int64_t)(m0 + r) * lda + k);
                if (n0 + r < N) w = Elem<T>::load(W + (int64_t)(n0 + r) * ldw + k);
            }
            As[kk][r] = a;
            Ws[kk][r] = w;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = Ws[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], w[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= N) continue;
            float x = acc[i][j];
            if (bias) x += Elem<T>::load(bias + n);
            if (act == BT_ACT_GELU) x = gelu_erf(x);
            if (res) x += Elem<T>::load(res + (int64_t)m * ldr + n);
            Elem<T>::store(out + (int64_t)m * ldo + n, x);
        }
    }
}

static int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            n = v;
        else
            n = 256;
    }
    return n;
}

// Dispatch knobs.  In the product library they are compile-time constants (no process-global mutable state); the debug
// build (-DBT_DEBUG_HOOKS -> libbt_hip_dbg.so, loaded by tests and tools/kernel_probe.py only) exposes bt_debug_* setters and
// additionally compiles the A/B and timing-ablation kernels.
#ifdef BT_DEBUG_HOOKS
#define BT_KNOB static
#else
#define BT_KNOB static constexpr
#endif
BT_KNOB bool g_force_generic = false;
BT_KNOB int g_skinny_rows = 16;   // bt_linear takes the MFMA weight streamer for 5 .. g_skinny_rows rows (kernel: <= 64; measured
                                  // B = 8: +21 %, 16: +4 %, 32: -20 %, 64: -30 % vs split-K + reduce -> 16)
BT_KNOB int g_max_wg = 0;         // cap the persistent kernel's grid so small problems walk several tiles per workgroup
BT_KNOB bool g_no_qkv_fusion = false;
// mid-batch QKV / h->4h GEMMs: 0 = slabs + separate reduce kernel always; 1 (default) = when ONE K slice already gives every CU a
// workgroup (M >~ 200) the GEMM runs its fused epilogue directly, no slab, no reduce launch; 2 = in-launch (last-arriver) reduction
// also for several slices -- measured SLOWER (B = 64: 42.7k -> 36.5k tok/s): every slice pays an agent-scope release fence over
// its freshly dirtied 16 KiB slab (guide: ~6.5 us) on a ~10 us GEMM; kept for A/B only
BT_KNOB int g_mid_fuse = 1;
BT_KNOB bool g_mid_force2 = true;    // two K slices for the dual launch also where one fills the chip (65..224 rows); false: A/B
BT_KNOB int g_mid_tile = 0;       // fused one-slice QKV / h->4h GEMMs at >= 128 rows: 0 = 64 x 64 tiles, 1 = 128 x 64, 2 = 128 x 128 (A/B)
BT_KNOB bool g_mid_attn = true;   // ... and the token decoder's local-window attention inside that reduce launch (false: A/B)
BT_KNOB bool g_mid_dual = true;   // 17..256 rows: QKV and h->4h split-K GEMMs of a layer in one launch, their reductions in one (false: A/B)
BT_KNOB bool g_mid_rows = true;   // 17..256-row stacks take the slab + fused-reduce path (false: A/B against the unfused launches)
BT_KNOB bool g_mid_wt = false;    // split-K slab stores of the 64 x 64 kernels write-through: measured -0.8 ... -2.6 % at B = 32 ... 128
                                  // (profiles/r5_ab_mid_write_through_slabs.jsonl; their slabs are a few MB and stay in L2) -> off (true: A/B)
BT_KNOB int g_mid_tall = 1;       // 97..256 rows: the slab GEMMs run on 256 x 128 tiles (0: the 64 x 64 kernels, A/B; 2: from 65 rows, A/B)
BT_KNOB int g_mid_rows_max = 512;  // ... and take stacks of up to this many rows (two row tiles; 256: the round-4 limit, A/B)
BT_KNOB int g_mid_tall_wgs = 256; // ... with about this many workgroups per launch (A/B)
BT_KNOB int g_mid_tall_wgs_res = 256; // ... (the launch with two operand pairs: the residual projections)
BT_KNOB bool g_mid_tall_wt = true; // ... and write-through slab stores (false: plain stores, A/B)
BT_KNOB int g_mid_tall_fuse = 1;  // round 6: the QKV + h->4h launch of that path with FUSED epilogues, no reduce launch.  1: where a tile has one K range
                                  // (257 .. 512 rows: B = 384 / 512 +3.6 %); 2 (debug A/B): also up to 256 rows, with the hand-off between a tile's two
                                  // K ranges -- measured -3 % (B = 256) / -10 % (B = 128): the reduce launch it removes also carried the token
                                  // decoder's local attention, which becomes a launch of its own (profiles/r6_ab_mid_fused_epilogues.jsonl); 0: off
// <= 2 rows, local window: the attention computed inside the residual-stream GEMV.  Built and A/B-measured in round 4
// (profiles/r4_ab_few_attn.jsonl, same box, interleaved): B = 1 1695 vs 1725 tok/s, B = 2 2950 vs 2921 -- a wash: the per-workgroup
// attention arithmetic (latency chains of ~28 cross-lane ops per item, redundantly in 512 workgroups) costs what the dependent
// launch cost.  Off; the kernel and its bit-identity test live in the debug library only.
BT_KNOB bool g_few_attn = false;
BT_KNOB bool g_few_persist = true;   // <= 2 rows: the [QKV | h->4h] launch as ~2 resident workgroups per CU walking the row groups (false: A/B)
BT_KNOB bool g_few_dual = true;   // <= 4 rows: QKV and h->4h GEMVs of a layer in one launch (false: two launches, A/B)
BT_KNOB int g_small_stages = 0;   // LDS ring depth of the 128^2 / 128x64 / 64x64 kernels: 0 = per-tile default, 2..4 forced (A/B)
// stream-K items kernel (gemm_bf16_mfma256sk_kernel, round 5) for full-tile problems whose 256 x 256 tiles leave the persistent
// kernel's last round under 90 % full, and for bt_linear2 above 256 rows: 1 = on (needs the caller's scratch: slabs + flags), 0 = A/B
BT_KNOB int g_sk = 1;
#ifdef BT_DEBUG_HOOKS
static int g_mid_tall_dbg = 0;       // bt_debug_mid_tall_dbg: timing-only ablations of the half-mode fused launch (TallFuse::dbg)
static void* g_timeline = nullptr;   // bt_debug_timeline_buffer: 4 workgroups x 8 waves x 128 stamps of 8 bytes (variant 28)
#endif
#ifdef BT_DEBUG_HOOKS
static uint32_t* g_sk_diag = nullptr;
#endif
BT_KNOB bool g_serial_epi = false;   // debug build (bt_debug_serial_epilogues): the round-5 tile / item boundary of the persistent and stream-K kernels
BT_KNOB int g_sk_drop_vw = -1;    // debug build (bt_debug_stream_k_drop): this publishing workgroup never raises its flags ...
BT_KNOB int g_sk_spin = 0;        // ... and consumers give up after this many polls (0: SK_SPIN_LIMIT) -> the time-out path under test
BT_KNOB int g_gemm_variant = 0;   // 0 auto; 1 force 128^2; 2/3 256^2 four-phase ring (plain/staggered); 4-8 timing ablations; 9/10 256^2
// two-phase ring, one tile per workgroup (staggered/plain); 14 persistent on 32x32x16 MFMA; default: persistent two-phase ring.

}  // namespace bt

using namespace bt;

// LayerNorm folded into a GEMM (EpiArgs.ln_*): the caller passes the raw rows as A, the folded weight W diag(gamma) as W, no bias
struct LnFold {
    const float2* stats;   // [M] (mu, rstd): bt_rowstats
    const float* s;        // [N] row sums of the folded weight
    const float* c;        // [N] W beta + b
};
// the persistent 256 x 256 kernel's own dispatch rule (the only kernel with the folded-LayerNorm epilogue)
static bool pp2_takes(int64_t M, int64_t N, int64_t K) {
    const int64_t tiles256 = ceil_div(M, BM2) * ceil_div(N, BN2);
    return (K % BK == 0) && K / BK >= 4 && (N % 8 == 0) &&
           tiles256 >= 192 && (tiles256 % 256 == 0 || tiles256 >= 1024 || (tiles256 % 256) >= 96) &&
           4 * M * N >= 3 * tiles256 * (int64_t)(BM2 * BN2);
}

// ---- stream-K items plan (gemm_bf16_mfma256sk_kernel) --------------------------------------------------------------------------
// T output tiles x C chunks of nk K-tiles, dealt tile-major in runs of q to G workgroups.  Picks the LARGEST nk (fewest item
// boundaries) for which the items divide evenly, an item never straddles the two operand pairs, and no tile is shared by more than
// two workgroups (the kernel's hand-off is pairwise).  ok = false: the problem stays on the other kernels.
struct SkPlan {
    bool ok;
    int nk, C, q, G;
};
static SkPlan sk_plan_search(int64_t M, int64_t N, int64_t K1, int64_t K2, int G);
// The search is O(T * nkt / 4) host work and the engine asks for the same few shapes up to three times per layer of every eager pass
// (ADVICE r5): a small per-thread memo of the last shapes (pure function of its arguments: no dispatch state, nothing shared)
static SkPlan sk_plan(int64_t M, int64_t N, int64_t K1, int64_t K2, int G) {
    struct Key { int64_t M, N, K1, K2; int G; SkPlan p; bool used; };
    static thread_local Key memo[8] = {};
    static thread_local int next = 0;
    for (const Key& k : memo)
        if (k.used && k.M == M && k.N == N && k.K1 == K1 && k.K2 == K2 && k.G == G) return k.p;
    const SkPlan p = sk_plan_search(M, N, K1, K2, G);
    memo[next] = Key{M, N, K1, K2, G, p, true};
    next = (next + 1) & 7;
    return p;
}
static SkPlan sk_plan_search(int64_t M, int64_t N, int64_t K1, int64_t K2, int G) {
    SkPlan p{false, 0, 0, 0, G};
    if (G < 8 || G % 8 != 0 || M % BM2 != 0 || N % BN2 != 0 || K1 % BK != 0 || K2 % BK != 0 || K1 <= 0) return p;
    const int64_t T = (M / BM2) * (N / BN2);
    const int nk1 = (int)(K1 / BK), nkt = (int)((K1 + K2) / BK);
    if (T < G / 2 || T > (1 << 20)) return p;
    for (int C = 1; C <= nkt / 4; ++C) {
        if (nkt % C != 0 || (T * C) % G != 0) continue;
        const int nk = nkt / C;
        if (nk < 4 || nk1 % nk != 0) continue;
        const int64_t q = T * C / G;
        if (q < 1 || q > (1 << 20)) continue;
        bool pairwise = true;
        for (int64_t t = 0; t < T && pairwise; ++t) pairwise = ((t * C + C - 1) / q - (t * C) / q) <= 1;
        if (!pairwise) continue;
        p.ok = true;
        p.nk = nk;
        p.C = C;
        p.q = (int)q;
        return p;
    }
    return p;
}
// [G slabs][G * 8 flag words][32 words = one 128-byte line of its own for the sticky time-out word]: the region ends on a line boundary, so
// the word never shares a cache line with whatever lies behind the scratch (round 6: in the token decoder's workspace that is the
// activation area of the 2 B-row pass -- written by other XCDs' GEMM epilogues, i.e. a line some L2 may still hold from before the reset)
constexpr int SK_TAIL_WORDS = 32;
static int64_t sk_ws_bytes(int G) { return (int64_t)G * SK_SLAB_FLOATS * 4 + ((int64_t)G * 8 + SK_TAIL_WORDS) * 4; }
namespace bt {
// scratch layout of the stream-K kernel: [G slabs of 256 KiB][G * 8 flag words][sticky time-out word + 31 spare = its own 128-byte line]; bt_decoder_forward
// sizes its scratch for it (engine.hip), zeroes flags + sticky word once per pass with sk_reset() and reports the sticky word at the end
// of the pass with sk_status(); the kernels leave the flag words zero
int64_t sk_scratch_bytes() { return sk_ws_bytes(num_cus()); }
// does bt_linear2 (M rows, N outputs, operand pairs of K1 and K2) run as ONE stream-K GEMM with this much scratch?
bool sk_linear2_ok(int64_t M, int64_t N, int64_t K1, int64_t K2, int64_t scratch_bytes) {
    const int G = num_cus();
    return g_sk != 0 && M > 256 && scratch_bytes >= sk_ws_bytes(G) && sk_plan(M, N, K1, K2, G).ok;
}
// Clearing is a KERNEL of this library, not hipMemsetAsync (round 6): inside replayed hipGraphs the words behind the 8 KiB of flags were
// found holding stale bytes at the end of a pass whose first node was a memset over them (profiles/r6_handoff_false_positive.md); a
// one-workgroup kernel is something the library controls in eager and captured launches alike.
__global__ __launch_bounds__(256) void sk_zero_kernel(uint32_t* words, int n) {
    for (int i = threadIdx.x; i < n; i += 256) __hip_atomic_store(words + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static int sk_zero(void* scratch, int64_t scratch_bytes, hipStream_t s, int words) {
    const int G = num_cus();
    if (!scratch || scratch_bytes < sk_ws_bytes(G)) return BT_OK;
    hipLaunchKernelGGL(sk_zero_kernel, dim3(1), dim3(256), 0, s,
                       reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(scratch) + (int64_t)G * SK_SLAB_FLOATS * 4), words);
    BT_LAUNCH_CHECK("stream-K flag clear launch");
    return BT_OK;
}
// the fused 256 x 128 launch's sticky time-out word (TF_STICKY of the mid-batch counter words) -> BT_DEV_ERR_HANDOFF_TIMEOUT
int tall_status(const int* counters, int32_t* err_flag, hipStream_t s, uint32_t tag);
// the same kernel for the mid-batch path's tile counters (engine.hip): no memset node is left in a captured block step
int zero_words(void* p, int words, hipStream_t s) {
    hipLaunchKernelGGL(sk_zero_kernel, dim3(1), dim3(256), 0, s, reinterpret_cast<uint32_t*>(p), words);
    BT_LAUNCH_CHECK("zero_words launch");
    return BT_OK;
}
// per launch (callers without sk_clean): the G * 8 hand-off flags only -- an unread time-out (the sticky word behind them) survives
int sk_clear_flags(void* scratch, int64_t scratch_bytes, hipStream_t s) { return sk_zero(scratch, scratch_bytes, s, num_cus() * 8); }
// flags AND the sticky word: a fresh scratch (undefined bytes), the start of a bt_decoder_forward pass (which reports the word itself
// at its end), or the caller's acknowledgement of a reported time-out
int sk_reset(void* scratch, int64_t scratch_bytes, hipStream_t s) { return sk_zero(scratch, scratch_bytes, s, num_cus() * 8 + SK_TAIL_WORDS); }
__global__ void sk_status_kernel(const uint32_t* sticky, int32_t* err_flag, uint32_t* diag, uint32_t tag) {
    // agent-scope acquire + an agent-scope load: the word is written by the clearing kernel (some XCD) and by the GEMMs' sc1 stores, and
    // must not be served from a line this XCD's L2 happens to hold from an earlier pass
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const uint32_t v = __hip_atomic_load(sticky, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (exactly the value a consumer that gave up stores: bytes nobody cleared are not a time-out)
    if (v == SK_TIMEOUT_MAGIC) atomicOr(err_flag, (int32_t)BT_DEV_ERR_HANDOFF_TIMEOUT);
    if (v != 0 && diag != nullptr) {     // debug build: what the word held and which pass saw it (tag = rows | q_len << 24 of the pass)
        const uint32_t slot = atomicAdd(diag + 1, 1u);
        if (slot < 8) {
            uint32_t* d = diag + 8 + 64 * 8 + slot * 8;
            d[0] = tag; d[1] = sticky[0]; d[2] = sticky[1]; d[3] = sticky[2]; d[4] = sticky[3]; d[5] = sticky[-1];
            d[6] = (uint32_t)reinterpret_cast<uint64_t>(sticky); d[7] = (uint32_t)(reinterpret_cast<uint64_t>(sticky) >> 32);
        }
    }
}
int sk_status(const void* scratch, int64_t scratch_bytes, int32_t* err_flag, hipStream_t s, uint32_t tag) {
    const int G = num_cus();
    if (!scratch || !err_flag || scratch_bytes < sk_ws_bytes(G)) return BT_OK;
    const uint32_t* sticky = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(scratch) + (int64_t)G * SK_SLAB_FLOATS * 4) + (int64_t)G * 8;
#ifdef BT_DEBUG_HOOKS
    uint32_t* diag = g_sk_diag;
#else
    uint32_t* diag = nullptr;
#endif
    hipLaunchKernelGGL(sk_status_kernel, dim3(1), dim3(1), 0, s, sticky, err_flag, diag, tag);
    BT_LAUNCH_CHECK("stream-K status launch");
    return BT_OK;
}
int tall_status(const int* counters, int32_t* err_flag, hipStream_t s, uint32_t tag) {
    if (!counters || !err_flag) return BT_OK;
#ifdef BT_DEBUG_HOOKS
    uint32_t* diag = g_sk_diag;
#else
    uint32_t* diag = nullptr;
#endif
    hipLaunchKernelGGL(sk_status_kernel, dim3(1), dim3(1), 0, s, reinterpret_cast<const uint32_t*>(counters) + TF_STICKY, err_flag, diag, tag);
    BT_LAUNCH_CHECK("mid-batch hand-off status launch");
    return BT_OK;
}
}  // namespace bt

extern "C" int bt_stream_k_reset(void* scratch, int64_t scratch_bytes, bt_stream stream) {
    BT_CHECK_ARG(scratch_bytes >= 0 && (scratch || scratch_bytes == 0), "bt_stream_k_reset: bad scratch");
    return bt::sk_reset(scratch, scratch_bytes, (hipStream_t)stream);
}
extern "C" int bt_stream_k_status(const void* scratch, int64_t scratch_bytes, int32_t* err_flag, bt_stream stream) {
    BT_CHECK_ARG(scratch_bytes >= 0 && (scratch || scratch_bytes == 0) && err_flag, "bt_stream_k_status: null pointer");
    return bt::sk_status(scratch, scratch_bytes, err_flag, (hipStream_t)stream, 0u);
}

// amax (bt_logits_argmax): when the persistent kernel takes the problem, run its PP_ARGMAX epilogue (per-slab argmax partials
// into amax[M][amax_ld]; the logits tile is stored only if store_out) and report it through *amax_done
static int linear_impl(int dtype, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* W,
                       const void* bias, const void* residual, int64_t ldr, void* out, int64_t ldo, int act,
                       const QkvEpi* qkv, bool* fused_out, bt_stream stream, void* scratch = nullptr,
                       int64_t scratch_bytes = 0, const Seg2* seg2 = nullptr, uint2* amax = nullptr, int amax_ld = 0,
                       bool store_out = true, bool* amax_done = nullptr, bool sk_clean = false, const LnFold* lnf = nullptr) {
    // seg2 (bt_linear2, > 64 rows): a second (x2, W2, bias2) operand pair accumulated into the same output tiles; only
    // the persistent 256^2 kernel and the non-split small-tile family take it -> BT_ERR_UNSUPPORTED otherwise
    BT_CHECK_ARG(dtype == BT_F32 || dtype == BT_BF16, "bt_linear: bad dtype %d", dtype);
    BT_CHECK_ARG(M >= 0 && N > 0 && K > 0, "bt_linear: bad shape M=%lld N=%lld K=%lld", (long long)M, (long long)N,
                 (long long)K);
    BT_CHECK_ARG(A && W && out, "bt_linear: null pointer");
    BT_CHECK_ARG(lda >= K && ldo >= N && (!residual || ldr >= N), "bt_linear: row stride smaller than row");
    BT_CHECK_ARG(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "bt_linear: dimension exceeds int32");
    BT_CHECK_ARG(act == BT_ACT_NONE || act == BT_ACT_GELU, "bt_linear: bad activation %d", act);
    if (fused_out) *fused_out = false;
    if (M == 0) return BT_OK;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == BT_BF16) {
        const bool mfma_ok = !g_force_generic && (K % BK == 0) && (lda % 8 == 0) && aligned16(A) && aligned16(W) &&
                             (ldo % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 7u) == 0) &&
                             (!residual || ((ldr % 4 == 0) && (reinterpret_cast<uintptr_t>(residual) & 7u) == 0)) &&
                             (!bias || (reinterpret_cast<uintptr_t>(bias) & 7u) == 0);
        EpiArgs ep{};
        ep.bias = (const bf16*)bias;
        ep.res = (const bf16*)residual;
        ep.ldr = ldr;
        ep.out = (bf16*)out;
        ep.ldo = ldo;
        ep.M = (int)M;
        ep.N = (int)N;
        ep.act = act;
        ep.q = QkvEpi{};
        Seg2 s2{};
        if (seg2) {
            const bool seg_ok = mfma_ok && !qkv && bias && seg2->bias && aligned16(bias) && aligned16(seg2->bias) &&
                                seg2->K > 0 && seg2->K % BK == 0 && seg2->ldx % 8 == 0 && seg2->ldw % 8 == 0 &&
                                aligned16(seg2->x) && aligned16(seg2->w) && (N % 8 == 0) && (ldo % 8 == 0) && aligned16(out) &&
                                (!residual || (ldr % 8 == 0 && aligned16(residual)));
            if (!seg_ok) {
                set_error("bt_linear2: > 64 rows need bf16, both biases, K1 %% 64 == K2 %% 64 == 0, N %% 8 == 0 and 16-byte aligned operands");
                return BT_ERR_UNSUPPORTED;
            }
            s2 = *seg2;
            ep.bias2 = s2.bias;   // small-tile family: the epilogue adds both; the persistent kernel starts from their sum
        }
        // fused QKV epilogue: only on the MFMA kernels, rot == 32 (checked by the caller), 32-aligned head_dim
        if (qkv && mfma_ok && !g_no_qkv_fusion) {
            ep.q = *qkv;
            ep.q.enabled = 1;
            if (fused_out) *fused_out = true;
        }
        // stream-K items kernel: full 256 x 256 tiles whose count leaves the persistent kernel's last round under 90 % full
        // (M = 4096 QKV: 384 tiles = 1.5 rounds), and every two-pair problem (bt_linear2 above 256 rows: attention.dense +
        // dense_4h_to_h as ONE K = H + I GEMM; M = 4096: 128 tiles x 10 chunks over 256 workgroups).  Needs the caller's scratch.
#ifndef BT_DEBUG_HOOKS
        if (lnf) {   // (measured: no net gain, profiles/r5_ab_ln_fused.jsonl -> the folded-LayerNorm epilogue is a debug-build experiment)
            set_error("bt_linear (folded LayerNorm): only in the debug build of the library");
            return BT_ERR_UNSUPPORTED;
        }
#endif
        if (lnf) {
            // folded LayerNorm: only the persistent kernel has that epilogue -- refuse anything it would not take, before any launch
            const bool ok = mfma_ok && !seg2 && !residual && !bias && !amax && g_gemm_variant == 0 && pp2_takes(M, N, K) && (ldo % 8 == 0) &&
                            aligned16(out) && lnf->stats && lnf->s && lnf->c && aligned16(lnf->s) && aligned16(lnf->c) &&
                            (reinterpret_cast<uintptr_t>(lnf->stats) & 7u) == 0 && N % 4 == 0 && (!qkv || ep.q.enabled);
            if (!ok) {
                set_error("bt_linear (folded LayerNorm): bf16, no bias / residual, and a shape the persistent 256 x 256 kernel takes "
                          "(>= 192 tiles, last round >= 96 tiles) are required");
                return BT_ERR_UNSUPPORTED;
            }
            ep.ln_stats = lnf->stats;
            ep.ln_s = lnf->s;
            ep.ln_c = lnf->c;
        }
        if (!lnf && g_sk && mfma_ok && g_gemm_variant == 0 && scratch && aligned16(scratch) && M > 256 && (ldo % 8 == 0) && aligned16(out) &&
            (!residual || (ldr % 8 == 0 && aligned16(residual))) && (!bias || aligned16(bias)) && !amax &&
            (act == BT_ACT_NONE || !residual)) {
            const int G = num_cus();
            const int64_t T = (M / BM2) * (N / BN2);
            // (at least one full round plus a remainder: problems below one round keep the one-tile kernels).  g_sk == 1: only where
            // the persistent kernel's own dispatch rule would refuse the shape (last round under 96 tiles: those ran on the 128 x 128
            // kernel -- M = 6144 QKV 181 -> 159 us, profiles/r5_decode_shapes_sk.jsonl); where it takes the shape (M = 4096 QKV: 384
            // tiles, last round half full) its half-empty last round runs faster than a full one and the hand-off costs what the
            // better balance saves (103 vs 101 us, same file) -> g_sk >= 2 (A/B) includes those
            const bool pp_takes = T >= 192 && (T % 256 == 0 || T >= 1024 || (T % 256) >= 96);
            const bool ragged = T > G && (T % G) != 0 && (double)T / (double)(ceil_div(T, G) * G) < 0.9 && (g_sk >= 2 || !pp_takes);
            const SkPlan pl = (seg2 || ragged) && scratch_bytes >= sk_ws_bytes(G) ? sk_plan(M, N, K, s2.K, G) : SkPlan{false, 0, 0, 0, G};
            if (pl.ok) {
                ProfScope ps(BT_PROF_LINEAR_MFMA, s);
                SkArgs sk{};
                sk.A2 = s2.x;
                sk.lda2 = s2.ldx;
                sk.W2 = s2.w;
                sk.ldw2 = s2.ldw;
                sk.bias2 = s2.bias;
                sk.nk1 = (int)(K / BK);
                sk.C = pl.C;
                sk.q = pl.q;
                sk.nk = pl.nk;
                sk.slabs = reinterpret_cast<float*>(scratch);
                sk.flags = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(scratch) + (int64_t)G * SK_SLAB_FLOATS * 4);
                sk.dbg_drop_vw = g_sk_drop_vw;
                sk.dbg_spin = (uint32_t)g_sk_spin;
#ifdef BT_DEBUG_HOOKS
                sk.dbg_diag = g_sk_diag;
#endif
                if (!sk_clean) BT_TRY(bt::sk_clear_flags(scratch, scratch_bytes, s));
                const int tiles_m = (int)(M / BM2), tiles_n = (int)(N / BN2);
#ifdef BT_DEBUG_HOOKS
#define BT_LAUNCH_SK(MODE_)                                                                                                  \
    do {                                                                                                                     \
        if (g_serial_epi)                                                                                                    \
            hipLaunchKernelGGL((gemm_bf16_mfma256sk_kernel<MODE_, true>), dim3(G), dim3(512), 0, s, (const bf16*)A, lda, (const bf16*)W, K, \
                               ep, tiles_m, tiles_n, sk);                                                                    \
        else                                                                                                                 \
            hipLaunchKernelGGL((gemm_bf16_mfma256sk_kernel<MODE_>), dim3(G), dim3(512), 0, s, (const bf16*)A, lda, (const bf16*)W, K, ep, \
                               tiles_m, tiles_n, sk);                                                                        \
    } while (0)
#else
#define BT_LAUNCH_SK(MODE_)                                                                                                  \
    hipLaunchKernelGGL((gemm_bf16_mfma256sk_kernel<MODE_>), dim3(G), dim3(512), 0, s, (const bf16*)A, lda, (const bf16*)W, K, ep, \
                       tiles_m, tiles_n, sk)
#endif
                if (ep.q.enabled) BT_LAUNCH_SK(PP_QKV);
                else if (act == BT_ACT_GELU) BT_LAUNCH_SK(PP_GELU);
                else if (residual) BT_LAUNCH_SK(PP_RES);
                else BT_LAUNCH_SK(PP_PLAIN);
#undef BT_LAUNCH_SK
                BT_LAUNCH_CHECK("bt_linear stream-K launch");
                return BT_OK;
            }
        }
        const int64_t tiles256 = ceil_div(M, BM2) * ceil_div(N, BN2);
        // big-tile kernel: one 8-wave workgroup per CU -> needs >= ~3/4 of a wave of 256 tiles and little tail quantisation
        // (round 3, profiles/r3_decode_shapes.jsonl + r3_pp_v1_vs_v2.jsonl: M = 4096, N = 6144 = 384 tiles runs 94 us on the persistent
        //  kernel vs 130 us on the 128^2 kernel, M = 2048, N = 6144 = 192 tiles 57 vs 70 us; 128 tiles lose: 54 vs 44 us)
        const bool pp_ok = (N % 8 == 0) && (ldo % 8 == 0) && aligned16(out) && (K + s2.K) / BK >= 4 &&
                           (!residual || (ldr % 8 == 0 && aligned16(residual))) && (!bias || aligned16(bias));
        const bool use256 = mfma_ok && g_gemm_variant != 1 && (N % 4 == 0) &&
                            (g_gemm_variant >= 2 || (tiles256 >= 192 && (tiles256 % 256 == 0 || tiles256 >= 1024 ||
                                                                         (tiles256 % 256) >= 96) &&
                                                     // ... and the 256 x 256 tiles must be mostly real work (M = 64 rows against the
                                                     // vocabulary is 197 tiles too, three quarters of them padding)
                                                     4 * M * N >= 3 * tiles256 * (int64_t)(BM2 * BN2))) &&
                            (!seg2 || (pp_ok && (g_gemm_variant == 0 || g_gemm_variant == 14)));
        if (use256) {
            ProfScope ps(BT_PROF_LINEAR_MFMA, s);
            const int tiles_m = (int)ceil_div(M, BM2), tiles_n = (int)ceil_div(N, BN2);
#define BT_LAUNCH_256(KERN_)                                                                                        \
    hipLaunchKernelGGL((KERN_), dim3(tiles_m * tiles_n), dim3(512), 0, s, (const bf16*)A, lda, (const bf16*)W, K, ep, \
                       (int)K, tiles_m, tiles_n)
#ifdef BT_DEBUG_HOOKS
            bool dbg_done = true;
            switch (g_gemm_variant) {
                case 2: BT_LAUNCH_256((gemm_bf16_mfma256_kernel<false, 0>)); break;
                case 3: BT_LAUNCH_256((gemm_bf16_mfma256_kernel<true, 0>)); break;
                case 4: BT_LAUNCH_256((gemm_bf16_mfma256_kernel<true, 1>)); break;   // timing ablation: no DMA
                case 5: BT_LAUNCH_256((gemm_bf16_mfma256_kernel<true, 2>)); break;   // timing ablation: no ds_reads
                case 6: BT_LAUNCH_256((gemm_bf16_mfma256_kernel<true, 4>)); break;   // ds_reads before the DMA issue
                case 7: BT_LAUNCH_256((gemm_bf16_mfma256_kernel<true, 8>)); break;   // DMA inside the MFMA cluster
                case 8: BT_LAUNCH_256((gemm_bf16_mfma256_kernel<true, 3>)); break;   // timing ablation: MFMA+barriers
                case 20: BT_LAUNCH_256((gemm_bf16_mfma256_kernel<true, 16>)); break;      // timing ablation: loads + ds_reads, no MFMA
                case 21: BT_LAUNCH_256((gemm_bf16_mfma256_kernel<true, 18>)); break;      // timing ablation: LDS-DMA stream only
                case 10: BT_LAUNCH_256((gemm_bf16_mfma256p2_kernel<false>)); break;
                case 11: BT_LAUNCH_256((gemm_bf16_mfma256p2_kernel<true, true>)); break;  // timing ablation: no epilogue
                case 9: BT_LAUNCH_256((gemm_bf16_mfma256p2_kernel<true>)); break;    // one tile per workgroup
                default: dbg_done = false; break;
            }
            if (!dbg_done)
#endif
            {
                {
                    if (pp_ok) {   // production: persistent, per-wave epilogue
                        const int n_wg = (int)std::min<int64_t>((int64_t)tiles_m * tiles_n, g_max_wg > 0 ? g_max_wg : num_cus());
#ifdef BT_DEBUG_HOOKS   /* A/B: 32x32x16 MFMA blocks (measured 5-8 % slower in this schedule) */
#define BT_PP_MMA32_AB(MODE_)                                                                                       \
    if (g_gemm_variant == 14)                                                                                       \
        hipLaunchKernelGGL((gemm_bf16_mfma256pp_kernel<MODE_, Mma32>), dim3(n_wg), dim3(512), 0, s, (const bf16*)A,   \
                           lda, (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n, Seg2{});                           \
    else
#else
#define BT_PP_MMA32_AB(MODE_)
#endif
#ifdef BT_DEBUG_HOOKS   /* timing ablations of the production kernel (wrong results): variants 30 + ABL bits */
#define BT_PP_ABL(MODE_, ABL_)                                                                                      \
    hipLaunchKernelGGL((gemm_bf16_mfma256pp_kernel<MODE_, Mma16, false, ABL_>), dim3(n_wg), dim3(512), 0, s,          \
                       (const bf16*)A, lda, (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n, Seg2{})
#define BT_PP_ABL_AB(MODE_)                                                                                         \
    if (g_gemm_variant >= 31 && g_gemm_variant <= 93) {                                                             \
        switch (g_gemm_variant - 30) {                                                                              \
            case 1: BT_PP_ABL(MODE_, 1); break;    /* no DMA */                                                     \
            case 2: BT_PP_ABL(MODE_, 2); break;    /* no ds_reads */                                                \
            case 3: BT_PP_ABL(MODE_, 3); break;    /* MFMA + barriers (+ epilogue) */                               \
            case 4: BT_PP_ABL(MODE_, 4); break;    /* no epilogue */                                                \
            case 7: BT_PP_ABL(MODE_, 7); break;    /* MFMA + barriers only */                                       \
            case 8: BT_PP_ABL(MODE_, 8); break;    /* all operand loads hit L2 */                                   \
            case 12: BT_PP_ABL(MODE_, 12); break;  /* L2 hits + no epilogue */                                      \
            case 16: BT_PP_ABL(MODE_, 16); break;  /* no MFMA */                                                    \
            case 18: BT_PP_ABL(MODE_, 18); break;  /* DMA ring only (+ epilogue) */                                 \
            case 22: BT_PP_ABL(MODE_, 22); break;  /* DMA ring only */                                              \
            case 30: BT_PP_ABL(MODE_, 30); break;  /* ring only, L2 hits */                                         \
            case 32: BT_PP_ABL(MODE_, 32); break;  /* K-order rotation (correct results) */                         \
            case 36: BT_PP_ABL(MODE_, 36); break;  /* K-order rotation, no epilogue */                              \
            case 54: BT_PP_ABL(MODE_, 54); break;  /* K-order rotation, ring only */                                \
            default: BT_PP_ABL(MODE_, 0); break;                                                                    \
        }                                                                                                           \
    } else
#else
#define BT_PP_ABL_AB(MODE_)
#endif
#ifdef BT_DEBUG_HOOKS   /* A/B: 16 = first-generation K-loop (scalar wait trees, per-half-tile address math) */
#define BT_PP_V1_AB(MODE_)                                                                                          \
    if (g_gemm_variant == 16)                                                                                       \
        hipLaunchKernelGGL((gemm_bf16_mfma256pp_kernel<MODE_, Mma16>), dim3(n_wg), dim3(512), 0, s, (const bf16*)A,   \
                           lda, (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n, Seg2{});                           \
    else if (g_gemm_variant == 18) /* second generation without the fast (static ring stage) K-tiles */             \
        hipLaunchKernelGGL((gemm_bf16_mfma256pp2_kernel<MODE_, 1024>), dim3(n_wg), dim3(512), 0, s, (const bf16*)A,   \
                           lda, (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n);                                   \
    else if (g_gemm_variant == 23) /* experiment: DMA instructions spread among the fragment reads */               \
        hipLaunchKernelGGL((gemm_bf16_mfma256pp2_kernel<MODE_, 4096>), dim3(n_wg), dim3(512), 0, s, (const bf16*)A,   \
                           lda, (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n);                                   \
    else if (g_gemm_variant == 24) /* A/B: s_setprio 1 around the MFMA cluster (round 2) */                          \
        hipLaunchKernelGGL((gemm_bf16_mfma256pp2_kernel<MODE_, 8192>), dim3(n_wg), dim3(512), 0, s, (const bf16*)A,   \
                           lda, (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n);                                   \
    else if (g_gemm_variant == 25) /* experiment: raised priority for the off-turn instead of the MFMA cluster */    \
        hipLaunchKernelGGL((gemm_bf16_mfma256pp2_kernel<MODE_, 16384>), dim3(n_wg), dim3(512), 0, s, (const bf16*)A,  \
                           lda, (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n);                                   \
    else if (g_gemm_variant == 26) /* experiment: second half-tile of a fast phase issued inside the MFMA cluster */  \
        hipLaunchKernelGGL((gemm_bf16_mfma256pp2_kernel<MODE_, 32768>), dim3(n_wg), dim3(512), 0, s, (const bf16*)A,  \
                           lda, (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n);                                   \
    else if (g_gemm_variant == 27) /* A/B: QKV epilogue with the generic (per-lane division) K / V addresses */      \
        hipLaunchKernelGGL((gemm_bf16_mfma256pp2_kernel<MODE_, 65536>), dim3(n_wg), dim3(512), 0, s, (const bf16*)A,  \
                           lda, (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n);                                   \
    else if (g_gemm_variant == 28) /* round 6: tile-boundary timeline (stamps -> g_timeline) */                       \
        { ep.amax = (MODE_ == PP_ARGMAX) ? ep.amax : (uint2*)g_timeline;                                             \
        hipLaunchKernelGGL((gemm_bf16_mfma256pp2_kernel<MODE_, 131072>), dim3(n_wg), dim3(512), 0, s, (const bf16*)A,  \
                           lda, (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n); }                                 \
    else if (g_gemm_variant == 29) /* round 6 A/B: QKV epilogue with the round-5 RoPE loads (two batches of four fragments) */ \
        hipLaunchKernelGGL((gemm_bf16_mfma256pp2_kernel<MODE_, 262144>), dim3(n_wg), dim3(512), 0, s, (const bf16*)A,  \
                           lda, (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n);                                   \
    else if (g_gemm_variant == 94 || (g_gemm_variant == 0 && g_serial_epi)) /* round 6 A/B: the round-5 tile boundary (the two wave groups' epilogues serialised); bt_debug_serial_epilogues */ \
        hipLaunchKernelGGL((gemm_bf16_mfma256pp2_kernel<MODE_, 1048576>), dim3(n_wg), dim3(512), 0, s, (const bf16*)A,  \
                           lda, (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n);                                   \
    else if (g_gemm_variant == 98) /* round 6 timing experiment (wrong results): QKV mode with k / v stored into the qkv buffer */ \
        hipLaunchKernelGGL((gemm_bf16_mfma256pp2_kernel<MODE_, 8388608>), dim3(n_wg), dim3(512), 0, s, (const bf16*)A,  \
                           lda, (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n);                                   \
    else if (g_gemm_variant == 97) /* round 6 experiment: the two waves of a SIMD alternate at raised priority through the epilogue */ \
        hipLaunchKernelGGL((gemm_bf16_mfma256pp2_kernel<MODE_, 4194304>), dim3(n_wg), dim3(512), 0, s, (const bf16*)A,  \
                           lda, (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n);                                   \
    else if (g_gemm_variant == 95) /* ... with the timeline stamps */                                                \
        { ep.amax = (MODE_ == PP_ARGMAX) ? ep.amax : (uint2*)g_timeline;                                             \
        hipLaunchKernelGGL((gemm_bf16_mfma256pp2_kernel<MODE_, 1048576 + 131072>), dim3(n_wg), dim3(512), 0, s, (const bf16*)A,  \
                           lda, (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n); }                                 \
    else if (g_gemm_variant == 30) /* round 6 A/B: the round-5 GELU (degree 12, builtin min / max) */                  \
        hipLaunchKernelGGL((gemm_bf16_mfma256pp2_kernel<MODE_, 524288>), dim3(n_wg), dim3(512), 0, s, (const bf16*)A,  \
                           lda, (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n);                                   \
    else if (g_gemm_variant == 19) /* experiment: cluster tail after the hand-over barrier */                       \
        hipLaunchKernelGGL((gemm_bf16_mfma256pp2_kernel<MODE_, 2048>), dim3(n_wg), dim3(512), 0, s, (const bf16*)A,   \
                           lda, (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n);                                   \
    else
#define BT_PP2_ABL(MODE_, ABL_)                                                                                     \
    hipLaunchKernelGGL((gemm_bf16_mfma256pp2_kernel<MODE_, ABL_>), dim3(n_wg), dim3(512), 0, s, (const bf16*)A, lda,  \
                       (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n)
#define BT_PP2_ABL_AB(MODE_)                                                                                        \
    if (g_gemm_variant >= 101 && g_gemm_variant <= 700) {                                                           \
        switch (g_gemm_variant - 100) {                                                                             \
            case 1: BT_PP2_ABL(MODE_, 1); break;    /* no DMA */                                                    \
            case 2: BT_PP2_ABL(MODE_, 2); break;    /* no ds_reads */                                               \
            case 4: BT_PP2_ABL(MODE_, 4); break;    /* no epilogue */                                               \
            case 7: BT_PP2_ABL(MODE_, 7); break;    /* MFMA + barriers only */                                      \
            case 22: BT_PP2_ABL(MODE_, 22); break;  /* DMA ring only */                                             \
            case 32: BT_PP2_ABL(MODE_, 32); break;  /* epilogue without global stores */                            \
            case 64: BT_PP2_ABL(MODE_, 64); break;  /* epilogue without LDS transposes */                           \
            case 96: BT_PP2_ABL(MODE_, 96); break;  /* epilogue = cvt only */                                       \
            case 128: BT_PP2_ABL(MODE_, 128); break; /* no accumulator init from the bias row */                    \
            case 256: BT_PP2_ABL(MODE_, 256); break; /* half of the stores */                                       \
            case 512: BT_PP2_ABL(MODE_, 512); break; /* two stores per wave and tile */                             \
            default: BT_PP2_ABL(MODE_, 0); break;                                                                   \
        }                                                                                                           \
    } else
#else
#define BT_PP_V1_AB(MODE_)
#define BT_PP2_ABL_AB(MODE_)
#endif
#define BT_LAUNCH_PP(MODE_)                                                                                           \
    do {                                                                                                              \
        BT_PP_MMA32_AB(MODE_)                                                                                       \
        BT_PP_V1_AB(MODE_)                                                                                          \
        hipLaunchKernelGGL((gemm_bf16_mfma256pp2_kernel<MODE_>), dim3(n_wg), dim3(512), 0, s, (const bf16*)A, lda,      \
                           (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n);                                          \
    } while (0)
#ifdef BT_DEBUG_HOOKS
#define BT_LAUNCH_PP2(MODE_)                                                                                          \
    hipLaunchKernelGGL((gemm_bf16_mfma256pp_kernel<MODE_, Mma16, true>), dim3(n_wg), dim3(512), 0, s, (const bf16*)A,   \
                       lda, (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n, s2)
#else   // (unreachable in the product library: linear2_impl never sends a two-pair problem to the 256 x 256 dispatch)
#define BT_LAUNCH_PP2(MODE_) do { set_error("bt_linear2: two-pair 256 x 256 kernel is a debug-build A/B kernel"); return BT_ERR_UNSUPPORTED; } while (0)
#endif
                        // (the engine never combines these: QKV has no activation / residual, GELU has no residual)
#ifdef BT_DEBUG_HOOKS
#define BT_LAUNCH_PP_LNF(MODE_)                                                                                       \
    hipLaunchKernelGGL((gemm_bf16_mfma256pp2_kernel<MODE_, 0, true>), dim3(n_wg), dim3(512), 0, s, (const bf16*)A, lda, \
                       (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n)
#else
#define BT_LAUNCH_PP_LNF(MODE_) do { } while (0)   /* unreachable: lnf is refused above in the product build */
#endif
                        if (lnf && ep.q.enabled) BT_LAUNCH_PP_LNF(PP_QKV);
                        else if (lnf && act == BT_ACT_GELU) BT_LAUNCH_PP_LNF(PP_GELU);
                        else if (lnf) BT_LAUNCH_PP_LNF(PP_PLAIN);
                        else if (seg2 && residual) BT_LAUNCH_PP2(PP_RES);
                        else if (seg2) BT_LAUNCH_PP2(PP_PLAIN);
                        else if (ep.q.enabled) BT_LAUNCH_PP(PP_QKV);
                        else if (act == BT_ACT_GELU && !residual) BT_LAUNCH_PP(PP_GELU);
                        else if (act == BT_ACT_NONE && residual) BT_LAUNCH_PP(PP_RES);
                        else if (act == BT_ACT_NONE && amax && g_gemm_variant == 0) {
                            ep.amax = amax;
                            ep.amax_ld = amax_ld;
                            if (!store_out) ep.out = nullptr;
                            BT_LAUNCH_PP(PP_ARGMAX);
                            if (amax_done) *amax_done = true;
                        }
                        else if (act == BT_ACT_NONE) {   // (timing ablations are instantiated for the plain mode only)
                            BT_PP_ABL_AB(PP_PLAIN)
                            BT_PP2_ABL_AB(PP_PLAIN)
                            BT_LAUNCH_PP(PP_PLAIN);
                        }
                        else BT_LAUNCH_256((gemm_bf16_mfma256p2_kernel<true>));   // GELU + residual: one-tile kernel
#undef BT_LAUNCH_PP
#undef BT_PP_V1_AB
#undef BT_PP2_ABL
#undef BT_PP2_ABL_AB
#undef BT_PP_ABL_AB
#undef BT_LAUNCH_PP2
#undef BT_LAUNCH_PP_LNF
#undef BT_PP_MMA32_AB
                    } else {
                        BT_LAUNCH_256((gemm_bf16_mfma256p2_kernel<true>));
                    }
                }
            }
#undef BT_LAUNCH_256
        } else if (mfma_ok && !seg2 && !ep.q.enabled && act == BT_ACT_NONE && g_gemm_variant == 0 && M % 256 == 0 && N % 128 == 0 &&
                   K / BK >= 4 && (M / 256) * (N / 128) > 192 && (M / 256) * (N / 128) <= 256 && (ldo % 8 == 0) && aligned16(out) &&
                   (!residual || (ldr % 8 == 0 && aligned16(residual))) && (K % 8 == 0)) {
            // one 256 x 128 tile per CU: the decode projections onto the residual stream (M = 4096, N = 2048: 256 tiles)
            ProfScope ps(BT_PROF_LINEAR_MFMA, s);
            const int tiles_m = (int)(M / 256), tiles_n = (int)(N / 128);
#define BT_LAUNCH_T21(RES_, PRIO_)                                                                                           \
    hipLaunchKernelGGL((gemm_bf16_mfma256x128_kernel<RES_, PRIO_>), dim3(tiles_m * tiles_n), dim3(512), 0, s, (const bf16*)A, lda,  \
                       (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n)
#ifdef BT_DEBUG_HOOKS   /* A/B: without s_setprio around the MFMA clusters (bt_debug_small_stages(9); the knob is otherwise unused here).
                           Unlike the persistent kernel this one is not faster without it: 0 .. -2 %, profiles/r3_t21_setprio.jsonl */
            if (g_small_stages == 9) {
                if (residual) BT_LAUNCH_T21(true, false);
                else BT_LAUNCH_T21(false, false);
            } else if (g_small_stages == 8) {   /* A/B: round-3 form without group 1's lgkmcnt(0) in front of the refill barriers */
                if (residual)
                    hipLaunchKernelGGL((gemm_bf16_mfma256x128_kernel<true, true, false>), dim3(tiles_m * tiles_n), dim3(512), 0, s,
                                       (const bf16*)A, lda, (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n);
                else
                    hipLaunchKernelGGL((gemm_bf16_mfma256x128_kernel<false, true, false>), dim3(tiles_m * tiles_n), dim3(512), 0, s,
                                       (const bf16*)A, lda, (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n);
            } else
#endif
            if (residual) BT_LAUNCH_T21(true, true);
            else BT_LAUNCH_T21(false, true);
#undef BT_LAUNCH_T21
        } else if (!seg2 && M <= 4 && !ep.q.enabled && g_gemm_variant != 1 && g_gemm_variant != 22 && (K % 512 == 0) && (lda % 8 == 0) &&
                   (K % 8 == 0) && aligned16(A) && aligned16(W) && !g_force_generic) {
            // a handful of rows: pure weight streaming (one launch, no fp32 partials).  Measured (bench.py, hipGraph replay):
            // B = 1 decode 700 -> 1084 tok/s; at M = 8 the per-row VALU work catches up with the split-K MFMA path, so M <= 4.
            ProfScope ps(BT_PROF_LINEAR_MFMA, s);
            const int mr = M <= 1 ? 1 : (M <= 2 ? 2 : 4);
            // rows per wave: keep >= ~512 workgroups in flight while giving a wave several rows' loads to overlap
            const int rw = N >= 32768 ? 4 : (N >= 4096 ? 2 : 1);
#define BT_LAUNCH_GEMV(MR_, RW_)                                                                                       \
    hipLaunchKernelGGL((gemv_bf16_kernel<MR_, RW_>), dim3((unsigned)ceil_div(N, 4 * RW_)), dim3(256), 0, s, (const bf16*)A, \
                       lda, (const bf16*)W, K, ep, (int)K, LnPro{})
#define BT_GEMV_RW(MR_)                                   \
    do {                                                  \
        if (rw == 4) BT_LAUNCH_GEMV(MR_, 4);              \
        else if (rw == 2) BT_LAUNCH_GEMV(MR_, 2);         \
        else BT_LAUNCH_GEMV(MR_, 1);                      \
    } while (0)
            if (mr == 1) BT_GEMV_RW(1);
            else if (mr == 2) BT_GEMV_RW(2);
            else BT_GEMV_RW(4);
#undef BT_GEMV_RW
#undef BT_LAUNCH_GEMV
        } else if (!seg2 && M <= g_skinny_rows && M <= 64 && !ep.q.enabled && g_gemm_variant != 1 && g_gemm_variant != 22 && !g_force_generic &&
                   skinny_ok(M, N, K, A, lda, W, bias, residual, ldr, out, ldo)) {
            // 5 .. 64 rows: one pass over W on the matrix pipe, one launch (no fp32 partials, no reduce kernel)
            ProfScope ps(BT_PROF_LINEAR_MFMA, s);
            launch_skinny<false, false>(s, M, N, (const bf16*)A, lda, (const bf16*)W, K, ep, K, LnPro{}, Seg2{});
        } else if (!seg2 && mfma_ok && scratch && !ep.q.enabled && g_gemm_variant != 1 && (N % 8 == 0) && (ldo % 8 == 0) &&
                   aligned16(out) && (!residual || (ldr % 8 == 0 && aligned16(residual))) && (!bias || aligned16(bias)) &&
                   aligned16(scratch) && ceil_div(M, 64) * ceil_div(N, 64) < 256) {
            // skinny regime: even 64x64 tiles leave CUs idle -> split K until >= ~256 workgroups stream W concurrently
            ProfScope ps(BT_PROF_LINEAR_MFMA, s);
            const int tiles_m = (int)ceil_div(M, 64), tiles_n = (int)ceil_div(N, 64);
            const int nk = (int)(K / BK);
            int slices = 1;
            // a scratch of the stream-K size carries that kernel's flag words behind its slabs: the partials stay in front of them
            // (a later sk_clean launch of the same pass assumes the words are zero -- ADVICE r5)
            const int64_t room = scratch_bytes >= sk_ws_bytes(num_cus()) ? (int64_t)num_cus() * SK_SLAB_FLOATS * 4 : scratch_bytes;
            while (slices < 16 && tiles_m * tiles_n * slices < 256 && nk % (slices * 2) == 0 && nk / (slices * 2) >= 2 &&
                   (int64_t)(slices * 2) * M * N * 4 <= room)
                slices *= 2;
            if (slices > 1) {
                if (g_small_stages == 2)
                    hipLaunchKernelGGL((gemm_bf16_mfma_kernel<64, 64, true, 2>), dim3(tiles_m * tiles_n, slices), dim3(256), 0, s,
                                       (const bf16*)A, lda, (const bf16*)W, K, ep, (int)(K / slices), tiles_m, tiles_n,
                                       (float*)scratch);
                else
                    hipLaunchKernelGGL((gemm_bf16_mfma_kernel<64, 64, true, 4>), dim3(tiles_m * tiles_n, slices), dim3(256), 0, s,
                                       (const bf16*)A, lda, (const bf16*)W, K, ep, (int)(K / slices), tiles_m, tiles_n,
                                       (float*)scratch);
                BT_LAUNCH_CHECK("bt_linear split-K launch");
                hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)ceil_div(M * (N / 8), 256)), dim3(256), 0, s,
                                   (const float*)scratch, slices, ep);
            } else if (g_small_stages == 2) {
                hipLaunchKernelGGL((gemm_bf16_mfma_kernel<64, 64, false, 2>), dim3(tiles_m * tiles_n), dim3(256), 0, s,
                                   (const bf16*)A, lda, (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n, nullptr);
            } else {
                hipLaunchKernelGGL((gemm_bf16_mfma_kernel<64, 64, false, 4>), dim3(tiles_m * tiles_n), dim3(256), 0, s,
                                   (const bf16*)A, lda, (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n, nullptr);
            }
        } else if (mfma_ok) {
            ProfScope ps(BT_PROF_LINEAR_MFMA, s);
            // pick the largest tile that still gives every CU a workgroup (256 CUs): 128x128 -> 128x64 -> 64x64
            int tbm = 128, tbn = 128;
            if (g_gemm_variant != 1) {
                if (ceil_div(M, 128) * ceil_div(N, 128) < 256) tbn = 64;
                if (ceil_div(M, 128) * ceil_div(N, 64) < 256 && tbn == 64) tbm = 64;
            }
            const int tiles_m = (int)ceil_div(M, tbm), tiles_n = (int)ceil_div(N, tbn);
            BT_CHECK_ARG((int64_t)tiles_m * tiles_n < (1ll << 31), "bt_linear: grid too large");
#define BT_LAUNCH_GEMM(TM_, TN_, NS_)                                                                                   \
    hipLaunchKernelGGL((gemm_bf16_mfma_kernel<TM_, TN_, false, NS_>), dim3(tiles_m * tiles_n), dim3(256), 0, s,             \
                       (const bf16*)A, lda, (const bf16*)W, K, ep, (int)K, tiles_m, tiles_n, nullptr, s2)
            // ring depth per tile: 64x64 x 4 stages and 128x64 x 3 stages still fit two workgroups per CU (64 / 72 KiB)
            // measured (profiles/r2_ring_ab.jsonl): 64x64 x 4 stages 1.2-1.5x faster at K >= 2048, 128x64 x 3 stages +7 %;
            // 128x128 x 3 stages (one workgroup per CU instead of two) 1.4x SLOWER -> the 128^2 tile keeps the double buffer
            const int ns = g_small_stages ? g_small_stages : (tbm == 64 ? 4 : (tbn == 64 ? 3 : 2));
            if (tbm == 128 && tbn == 128) { if (ns == 2) BT_LAUNCH_GEMM(128, 128, 2); else BT_LAUNCH_GEMM(128, 128, 3); }
            else if (tbm == 128) { if (ns == 2) BT_LAUNCH_GEMM(128, 64, 2); else BT_LAUNCH_GEMM(128, 64, 3); }
            else { if (ns == 2) BT_LAUNCH_GEMM(64, 64, 2); else BT_LAUNCH_GEMM(64, 64, 4); }
#undef BT_LAUNCH_GEMM
        } else {
            ProfScope ps(BT_PROF_LINEAR_GENERIC, s);
            dim3 grid((unsigned)ceil_div(N, 64), (unsigned)ceil_div(M, 64));
            BT_CHECK_ARG(grid.y <= 65535u, "bt_linear: M too large for the generic kernel");
            hipLaunchKernelGGL(gemm_generic_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)A, lda, (const bf16*)W,
                               K, (const bf16*)bias, (const bf16*)residual, ldr, (bf16*)out, ldo, (int)M, (int)N,
                               (int)K, act);
        }
    } else {
        ProfScope ps(BT_PROF_LINEAR_GENERIC, s);
        dim3 grid((unsigned)ceil_div(N, 64), (unsigned)ceil_div(M, 64));
        BT_CHECK_ARG(grid.y <= 65535u, "bt_linear: M too large for the generic kernel");
        hipLaunchKernelGGL(gemm_generic_kernel<float>, grid, dim3(256), 0, s, (const float*)A, lda, (const float*)W, K,
                           (const float*)bias, (const float*)residual, ldr, (float*)out, ldo, (int)M, (int)N, (int)K,
                           act);
    }
    BT_LAUNCH_CHECK("bt_linear launch");
    return BT_OK;
}

extern "C" int bt_linear(int dtype, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* W,
                         const void* bias, const void* residual, int64_t ldr, void* out, int64_t ldo, int act,
                         bt_stream stream) {
    return linear_impl(dtype, M, N, K, A, lda, W, bias, residual, ldr, out, ldo, act, nullptr, nullptr, stream);
}

// ---- bt_logits_argmax: greedy token step = logits GEMM + argmax --------------------------------------------------------------
// one wave per row: (value, column) partials of the row's 64-column slabs -> the row's argmax (larger value wins, lowest column on ties)
__global__ __launch_bounds__(256) void argmax_partials_kernel(const uint2* __restrict__ part, int ld, int slabs, int M,
                                                             int64_t* __restrict__ ids_out, int64_t ids_stride) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    for (int c = lane; c < slabs; c += 64) {
        const uint2 p = part[(int64_t)m * ld + c];
        const float v = __uint_as_float(p.x);
        const int i = (int)p.y;
        if (i != 0x7fffffff && (v > best || bidx == 0x7fffffff || (v == best && i < bidx))) { best = v; bidx = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        if (oi != 0x7fffffff && (ov > best || bidx == 0x7fffffff || (ov == best && oi < bidx))) { best = ov; bidx = oi; }
    }
    if (lane == 0) ids_out[(int64_t)m * ids_stride] = bidx == 0x7fffffff ? 0 : bidx;
}

extern "C" int64_t bt_logits_argmax_workspace_bytes(int64_t M, int64_t V) {
    return M < 0 || V <= 0 ? -1 : M * ceil_div(V, 64) * (int64_t)sizeof(uint2);
}

extern "C" int bt_logits_argmax(int dtype, int64_t M, int64_t V, int64_t K, const void* hid, int64_t ldh, const void* W,
                                void* logits, int64_t ld_logits, int32_t store_logits, int64_t* ids_out, int64_t ids_stride,
                                void* ws, int64_t ws_bytes, bt_stream stream) {
    BT_CHECK_ARG(logits && ids_out, "bt_logits_argmax: null pointer (the logits buffer is required: shapes the fused kernel does not "
                                    "take run bt_linear + bt_argmax through it)");
    bool fused = false;
    uint2* part = nullptr;
    if (dtype == BT_BF16 && ws && aligned16(ws) && ws_bytes >= bt_logits_argmax_workspace_bytes(M, V) && V < (1ll << 31))
        part = reinterpret_cast<uint2*>(ws);
    BT_TRY(linear_impl(dtype, M, V, K, hid, ldh, W, nullptr, nullptr, 0, logits, ld_logits, BT_ACT_NONE, nullptr, nullptr, stream,
                       nullptr, 0, nullptr, part, (int)ceil_div(V, 64), store_logits != 0, &fused));
    if (M == 0) return BT_OK;
    if (!fused) return bt_argmax(dtype, M, V, logits, ld_logits, ids_out, ids_stride, stream);
    ProfScope ps(BT_PROF_ARGMAX, (hipStream_t)stream);
    hipLaunchKernelGGL(argmax_partials_kernel, dim3((unsigned)ceil_div(M, 4)), dim3(256), 0, (hipStream_t)stream, part,
                       (int)ceil_div(V, 64), (int)ceil_div(V, 64), (int)M, ids_out, ids_stride);
    BT_LAUNCH_CHECK("bt_logits_argmax reduce launch");
    return BT_OK;
}

extern "C" int bt_linear_ws(int dtype, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* W,
                            const void* bias, const void* residual, int64_t ldr, void* out, int64_t ldo, int act,
                            void* scratch, int64_t scratch_bytes, bt_stream stream) {
    return linear_impl(dtype, M, N, K, A, lda, W, bias, residual, ldr, out, ldo, act, nullptr, nullptr, stream, scratch,
                       scratch_bytes);
}
namespace bt {
// engine-side forms (engine.hip): the same calls with `sk_clean` = the stream-K flag words of `scratch` are known to be zero
// (bt_decoder_forward clears them once per pass; the kernels leave them zero) -> no memset in front of every launch
int linear_ws_impl(int dtype, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* W, const void* bias,
                   const void* residual, int64_t ldr, void* out, int64_t ldo, int act, void* scratch, int64_t scratch_bytes,
                   bool sk_clean, bt_stream stream) {
    return linear_impl(dtype, M, N, K, A, lda, W, bias, residual, ldr, out, ldo, act, nullptr, nullptr, stream, scratch, scratch_bytes,
                       nullptr, nullptr, 0, true, nullptr, sk_clean);
}
}  // namespace bt

extern "C" int bt_qkv_rope_append(int dtype, int64_t B, int32_t q_len, int32_t heads, int32_t head_dim, int32_t rot,
                                  const void* x, int64_t ldx, const void* qkv_w, const void* qkv_b, void* qkv,
                                  int64_t ld_qkv, void* k_cache, void* v_cache, int64_t n_max, const int32_t* pos0_dev,
                                  int32_t pos0_host, const int32_t* pos_table, const float* cos_tab, const float* sin_tab,
                                  void* scratch, int64_t scratch_bytes, bt_stream stream) {
    return bt::qkv_rope_append_impl(dtype, B, q_len, heads, head_dim, rot, x, ldx, qkv_w, qkv_b, qkv, ld_qkv, k_cache, v_cache, n_max,
                                    pos0_dev, pos0_host, pos_table, cos_tab, sin_tab, scratch, scratch_bytes, false, stream, nullptr,
                                    nullptr, nullptr);
}

// ---- LayerNorm folded into the consuming GEMM (round 5) ---------------------------------------------------------------------------
// out = act( LayerNorm(x; gamma, beta) . W^T + b )  computed as  act( rstd (x . W'^T - mu s) + c )  on the RAW rows x:
//   W' = W diag(gamma) (bf16, [N, K]),  s[n] = sum_k W'[n, k],  c[n] = (W beta)[n] + b[n]  (fp32, built once per weight by the caller:
//   block_transformer_amd/runtime.py),  stats[m] = (mu, rstd) of row m (bt_rowstats).  The stand-alone LayerNorm launch of a large
//   stack read M H and wrote 2 M H elements; bt_rowstats reads M H and writes 8 M bytes.  Only where the persistent 256 x 256
//   kernel takes the shape (bt_lnfold_shape_ok); BT_ERR_UNSUPPORTED otherwise, nothing launched.
// Built and measured in round 5 (profiles/r5_ab_ln_fused.jsonl, one setting per process): the LayerNorm family of a generate goes
// 95 -> 41 ms, the GEMM family 2851 -> 2908 ms (24 dependent loads + 256 FMAs per lane at the head of every tile's epilogue, with the
// next tile's operand stream waiting behind them) -- 125.7k tok/s either way.  So the mechanism lives in the DEBUG library only
// (bt_debug_* entry points, the engine's bt_debug_ln_fold knob); the product keeps the dual LayerNorm launch.
#ifdef BT_DEBUG_HOOKS
namespace bt {
bool lnfold_shape_ok(int64_t M, int64_t N, int64_t K) { return M > 256 && pp2_takes(M, N, K) && g_gemm_variant == 0; }
}
extern "C" int32_t bt_debug_lnfold_shape_ok(int64_t M, int64_t N, int64_t K) { return bt::lnfold_shape_ok(M, N, K) ? 1 : 0; }

extern "C" int bt_debug_linear_lnfold(int dtype, int64_t M, int64_t N, int64_t K, const void* x, int64_t ldx, const void* stats,
                                const void* w_folded, const float* s_vec, const float* c_vec, void* out, int64_t ldo, int act,
                                bt_stream stream) {
    BT_CHECK_ARG(stats && s_vec && c_vec, "bt_debug_linear_lnfold: null pointer");
    LnFold lnf{reinterpret_cast<const float2*>(stats), s_vec, c_vec};
    return linear_impl(dtype, M, N, K, x, ldx, w_folded, nullptr, nullptr, 0, out, ldo, act, nullptr, nullptr, stream, nullptr, 0, nullptr,
                       nullptr, 0, true, nullptr, false, &lnf);
}
#endif  // BT_DEBUG_HOOKS (folded LayerNorm entry points)

int bt::qkv_rope_append_impl(int dtype, int64_t B, int32_t q_len, int32_t heads, int32_t head_dim, int32_t rot, const void* x,
                             int64_t ldx, const void* qkv_w, const void* qkv_b, void* qkv, int64_t ld_qkv, void* k_cache,
                             void* v_cache, int64_t n_max, const int32_t* pos0_dev, int32_t pos0_host, const int32_t* pos_table,
                             const float* cos_tab, const float* sin_tab, void* scratch, int64_t scratch_bytes, bool sk_clean,
                             bt_stream stream, const void* ln_stats, const float* ln_s, const float* ln_c) {
    BT_CHECK_ARG(B >= 0 && q_len > 0 && heads > 0 && head_dim > 0, "bt_qkv_rope_append: bad shape");
    BT_CHECK_ARG(x && qkv_w && qkv && k_cache && v_cache && cos_tab && sin_tab, "bt_qkv_rope_append: null pointer");
    BT_CHECK_ARG(pos0_dev || (pos0_host >= 0 && pos0_host + q_len <= n_max), "bt_qkv_rope_append: cache overflow");
    const int64_t H = (int64_t)heads * head_dim;
    const int64_t M = B * q_len;
    QkvEpi q{};
    q.kc = (bf16*)k_cache;
    q.vc = (bf16*)v_cache;
    q.n_max = n_max;
    q.pos0_dev = pos0_dev;
    q.pos_table = pos_table;
    q.cos_tab = cos_tab;
    q.sin_tab = sin_tab;
    q.pos0_host = pos0_host;
    q.q_len = q_len;
    q.heads = heads;
    q.head_dim = head_dim;
    const bool fusable = dtype == BT_BF16 && rot == 32 && head_dim % 32 == 0 && ld_qkv % 4 == 0 &&
                         (reinterpret_cast<uintptr_t>(k_cache) & 7u) == 0 && (reinterpret_cast<uintptr_t>(v_cache) & 7u) == 0 &&
                         (reinterpret_cast<uintptr_t>(cos_tab) & 15u) == 0 && (reinterpret_cast<uintptr_t>(sin_tab) & 15u) == 0;
    bool fused = false;
    // skinny problems stream W faster with split-K (an unfused RoPE pass over a few rows costs nothing)
    const bool skinny = scratch != nullptr && ceil_div(M, 64) * ceil_div(3 * H, 64) < 256;
    LnFold lnf{reinterpret_cast<const float2*>(ln_stats), ln_s, ln_c};
    if (ln_stats && !(fusable && !skinny)) {
        set_error("bt_qkv_rope_append (folded LayerNorm): needs the fused RoPE + append epilogue (bf16, rot == 32, head_dim %% 32 == 0)");
        return BT_ERR_UNSUPPORTED;
    }
    BT_TRY(linear_impl(dtype, M, 3 * H, H, x, ldx, qkv_w, qkv_b, nullptr, 0, qkv, ld_qkv, BT_ACT_NONE,
                       (fusable && !skinny) ? &q : nullptr, &fused, stream, scratch, scratch_bytes, nullptr, nullptr, 0, true, nullptr,
                       sk_clean, ln_stats ? &lnf : nullptr));
    if (!fused)
        BT_TRY(bt_rope_kv_append(dtype, B, q_len, heads, head_dim, rot, qkv, ld_qkv, k_cache, v_cache, n_max, pos0_dev,
                                 pos0_host, pos_table, cos_tab, sin_tab, stream));
    return BT_OK;
}

// ---- fused small-batch entry points (M <= 4 rows; see gemv_bf16_kernel) -------------------------------------------------
static bool gemv_ln_ok(int dtype, int64_t M, int64_t N, int64_t K, const void* x, int64_t ldx, const void* W,
                       const void* ln_w, const void* ln_b) {
    return dtype == BT_BF16 && M >= 1 && M <= 4 && K % 512 == 0 && K <= GV_KC && ldx % 8 == 0 && aligned16(x) && aligned16(W) &&
           aligned16(ln_w) && aligned16(ln_b) && !g_force_generic && g_gemm_variant != 22 && N > 0;
}

extern "C" int bt_ln_linear(int dtype, int64_t M, int64_t N, int64_t K, const void* x, int64_t ldx, const void* ln_w,
                            const void* ln_b, float eps, const void* W, const void* bias, const void* residual, int64_t ldr,
                            void* out, int64_t ldo, int act, bt_stream stream) {
    BT_CHECK_ARG(x && ln_w && ln_b && W && out, "bt_ln_linear: null pointer");
    BT_CHECK_ARG(act == BT_ACT_NONE || act == BT_ACT_GELU, "bt_ln_linear: bad activation %d", act);
    const bool gemv = gemv_ln_ok(dtype, M, N, K, x, ldx, W, ln_w, ln_b);
    const bool skinny = !gemv && dtype == BT_BF16 && M > 4 && skinny_ok(M, N, K, x, ldx, W, bias, residual, ldr, out, ldo) &&
                        aligned16(ln_w) && aligned16(ln_b) && !g_force_generic && g_gemm_variant != 22;
    if (!gemv && !skinny) {
        set_error("bt_ln_linear: only bf16 with 1 <= M <= 4 (K %% 512 == 0, K <= %d) or 5 <= M <= 64 (K %% 32 == 0, N %% 4 == 0), "
                  "16-byte aligned operands (M=%lld K=%lld)", GV_KC, (long long)M, (long long)K);
        return BT_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    EpiArgs ep{};
    ep.bias = (const bf16*)bias;
    ep.res = (const bf16*)residual;
    ep.ldr = ldr;
    ep.out = (bf16*)out;
    ep.ldo = ldo;
    ep.M = (int)M;
    ep.N = (int)N;
    ep.act = act;
    ep.q = QkvEpi{};
    const LnPro ln{(const bf16*)ln_w, (const bf16*)ln_b, eps};
    ProfScope ps(BT_PROF_LINEAR_MFMA, s);
    if (skinny) {
        launch_skinny<true, false>(s, M, N, (const bf16*)x, ldx, (const bf16*)W, K, ep, K, ln, Seg2{});
        BT_LAUNCH_CHECK("bt_ln_linear launch");
        return BT_OK;
    }
    const int rw = N >= 32768 ? 4 : (N >= 4096 ? 2 : 1);
#define BT_LAUNCH_GEMV_LN(MR_, RW_)                                                                                        \
    hipLaunchKernelGGL((gemv_bf16_kernel<MR_, RW_, true, false>), dim3((unsigned)ceil_div(N, 4 * RW_)), dim3(256), 0, s,       \
                       (const bf16*)x, ldx, (const bf16*)W, K, ep, (int)K, ln)
#define BT_GEMV_LN_RW(MR_)                                   \
    do {                                                     \
        if (rw == 4) BT_LAUNCH_GEMV_LN(MR_, 4);              \
        else if (rw == 2) BT_LAUNCH_GEMV_LN(MR_, 2);         \
        else BT_LAUNCH_GEMV_LN(MR_, 1);                      \
    } while (0)
    if (M <= 1) BT_GEMV_LN_RW(1);
    else if (M <= 2) BT_GEMV_LN_RW(2);
    else BT_GEMV_LN_RW(4);
#undef BT_GEMV_LN_RW
#undef BT_LAUNCH_GEMV_LN
    BT_LAUNCH_CHECK("bt_ln_linear launch");
    return BT_OK;
}

extern "C" int bt_ln_qkv_rope_append(int dtype, int64_t B, int32_t q_len, int32_t heads, int32_t head_dim, int32_t rot,
                                     const void* x, int64_t ldx, const void* ln_w, const void* ln_b, float eps,
                                     const void* qkv_w, const void* qkv_b, void* qkv, int64_t ld_qkv, void* k_cache,
                                     void* v_cache, int64_t n_max, const int32_t* pos0_dev, int32_t pos0_host,
                                     const int32_t* pos_table, const float* cos_tab, const float* sin_tab, bt_stream stream) {
    BT_CHECK_ARG(B >= 0 && q_len > 0 && heads > 0 && head_dim > 0, "bt_ln_qkv_rope_append: bad shape");
    BT_CHECK_ARG(x && qkv_w && qkv && k_cache && v_cache && cos_tab && sin_tab, "bt_ln_qkv_rope_append: null pointer");
    BT_CHECK_ARG(pos0_dev || (pos0_host >= 0 && pos0_host + q_len <= n_max), "bt_ln_qkv_rope_append: cache overflow");
    const int64_t H = (int64_t)heads * head_dim;
    const int64_t M = B * q_len;
    // ln_w == NULL: no LayerNorm prologue (x is already normalised) -- RoPE / append fusion only
    const bool ln_on = ln_w != nullptr;
    const bool common = dtype == BT_BF16 && ldx % 8 == 0 && aligned16(x) && aligned16(qkv_w) &&
                        (!ln_on || (ln_b && aligned16(ln_w) && aligned16(ln_b))) && rot == 32 && head_dim % 32 == 0 &&
                        !g_force_generic && g_gemm_variant != 22;
    const bool ok = common && M >= 1 && M <= 4 && H % 512 == 0 && (!ln_on || H <= GV_KC);
    const bool skinny = common && !ok && M > 4 && M <= 64 && H % 32 == 0 && ld_qkv % 4 == 0 &&
                        (reinterpret_cast<uintptr_t>(qkv) & 7u) == 0 && (reinterpret_cast<uintptr_t>(k_cache) & 7u) == 0 &&
                        (reinterpret_cast<uintptr_t>(v_cache) & 7u) == 0 && (!qkv_b || (reinterpret_cast<uintptr_t>(qkv_b) & 7u) == 0) &&
                        (reinterpret_cast<uintptr_t>(cos_tab) & 15u) == 0 && (reinterpret_cast<uintptr_t>(sin_tab) & 15u) == 0;
    if (!ok && !skinny) {
        set_error("bt_ln_qkv_rope_append: only bf16, rot == 32, head_dim %% 32 == 0, and 1 <= rows <= 4 (hidden %% 512 == 0, <= %d "
                  "with LayerNorm) or 5 <= rows <= 64", GV_KC);
        return BT_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    EpiArgs ep{};
    ep.bias = (const bf16*)qkv_b;
    ep.res = nullptr;
    ep.ldr = 0;
    ep.out = (bf16*)qkv;
    ep.ldo = ld_qkv;
    ep.M = (int)M;
    ep.N = (int)(3 * H);
    ep.act = BT_ACT_NONE;
    ep.q = QkvEpi{};
    ep.q.kc = (bf16*)k_cache;
    ep.q.vc = (bf16*)v_cache;
    ep.q.n_max = n_max;
    ep.q.pos0_dev = pos0_dev;
    ep.q.pos_table = pos_table;
    ep.q.cos_tab = cos_tab;
    ep.q.sin_tab = sin_tab;
    ep.q.pos0_host = pos0_host;
    ep.q.q_len = q_len;
    ep.q.heads = heads;
    ep.q.head_dim = head_dim;
    ep.q.enabled = 1;
    const LnPro ln{(const bf16*)ln_w, (const bf16*)ln_b, eps};
    ProfScope ps(BT_PROF_LINEAR_MFMA, s);
    if (skinny) {
        if (ln_on) launch_skinny<true, true>(s, M, 3 * H, (const bf16*)x, ldx, (const bf16*)qkv_w, H, ep, H, ln, Seg2{});
        else launch_skinny<false, true>(s, M, 3 * H, (const bf16*)x, ldx, (const bf16*)qkv_w, H, ep, H, ln, Seg2{});
        BT_LAUNCH_CHECK("bt_ln_qkv_rope_append launch");
        return BT_OK;
    }
    const unsigned grid = (unsigned)ceil_div(3 * H / 2, 4);   // one wave per (n, n + 16) pair, 4 pairs per workgroup
#define BT_LAUNCH_GEMV_QKV(MR_)                                                                                       \
    do {                                                                                                              \
        if (ln_on)                                                                                                    \
            hipLaunchKernelGGL((gemv_bf16_kernel<MR_, 2, true, true>), dim3(grid), dim3(256), 0, s, (const bf16*)x, ldx, \
                               (const bf16*)qkv_w, H, ep, (int)H, ln);                                                \
        else                                                                                                          \
            hipLaunchKernelGGL((gemv_bf16_kernel<MR_, 2, false, true>), dim3(grid), dim3(256), 0, s, (const bf16*)x, ldx, \
                               (const bf16*)qkv_w, H, ep, (int)H, ln);                                                \
    } while (0)
    if (M <= 1) BT_LAUNCH_GEMV_QKV(1);
    else if (M <= 2) BT_LAUNCH_GEMV_QKV(2);
    else BT_LAUNCH_GEMV_QKV(4);
#undef BT_LAUNCH_GEMV_QKV
    BT_LAUNCH_CHECK("bt_ln_qkv_rope_append launch");
    return BT_OK;
}

extern "C" int bt_linear2(int dtype, int64_t M, int64_t N, int64_t K1, const void* x1, int64_t ldx1, const void* W1,
                          const void* bias1, int64_t K2, const void* x2, int64_t ldx2, const void* W2, const void* bias2,
                          const void* residual, int64_t ldr, void* out, int64_t ldo, bt_stream stream) {
    return bt::linear2_impl(dtype, M, N, K1, x1, ldx1, W1, bias1, K2, x2, ldx2, W2, bias2, residual, ldr, out, ldo, nullptr, 0, false,
                            stream);
}

extern "C" int bt_linear2_ws(int dtype, int64_t M, int64_t N, int64_t K1, const void* x1, int64_t ldx1, const void* W1,
                             const void* bias1, int64_t K2, const void* x2, int64_t ldx2, const void* W2, const void* bias2,
                             const void* residual, int64_t ldr, void* out, int64_t ldo, void* scratch, int64_t scratch_bytes,
                             bt_stream stream) {
    return bt::linear2_impl(dtype, M, N, K1, x1, ldx1, W1, bias1, K2, x2, ldx2, W2, bias2, residual, ldr, out, ldo, scratch,
                            scratch_bytes, false, stream);
}

extern "C" int64_t bt_stream_k_scratch_bytes(void) { return bt::sk_scratch_bytes(); }

int bt::linear2_impl(int dtype, int64_t M, int64_t N, int64_t K1, const void* x1, int64_t ldx1, const void* W1, const void* bias1,
                     int64_t K2, const void* x2, int64_t ldx2, const void* W2, const void* bias2, const void* residual, int64_t ldr,
                     void* out, int64_t ldo, void* scratch, int64_t scratch_bytes, bool sk_clean, bt_stream stream) {
    BT_CHECK_ARG(x1 && W1 && x2 && W2 && out, "bt_linear2: null pointer");
    if (dtype == BT_BF16 && M > 64) {
        // large stacks: one MFMA GEMM over K = K1 + K2, both biases and the residual in its epilogue -- the residual stream is read
        // and written once instead of twice.  Product library: the stream-K items kernel when the caller's scratch and the shape
        // allow it (sk_linear2_ok), the small-tile family with its second operand pair below 192 tiles of 256 x 256, and otherwise
        // simply the two projections one after the other (same sums, one more bf16 rounding of the residual stream).  The debug
        // build keeps the first-generation two-pair persistent kernel for the A/B that retired it (profiles/r2_probe_lin2.jsonl).
#ifndef BT_DEBUG_HOOKS
        const bool sk_takes = scratch && sk_linear2_ok(M, N, K1, K2, scratch_bytes) && aligned16(scratch);
        const int64_t tiles256 = ceil_div(M, BM2) * ceil_div(N, BN2);
        if (!sk_takes && tiles256 >= 192) {
            const bool seg_ok = bias1 && bias2 && aligned16(bias1) && aligned16(bias2) && K1 % BK == 0 && K2 > 0 && K2 % BK == 0 &&
                                ldx1 % 8 == 0 && ldx2 % 8 == 0 && aligned16(x1) && aligned16(x2) && aligned16(W1) && aligned16(W2) &&
                                (N % 8 == 0) && (ldo % 8 == 0) && aligned16(out) && (!residual || (ldr % 8 == 0 && aligned16(residual)));
            if (!seg_ok) {
                set_error("bt_linear2: > 64 rows need bf16, both biases, K1 %% 64 == K2 %% 64 == 0, N %% 8 == 0 and 16-byte aligned operands");
                return BT_ERR_UNSUPPORTED;
            }
            BT_TRY(linear_impl(dtype, M, N, K2, x2, ldx2, W2, bias2, residual, ldr, out, ldo, BT_ACT_NONE, nullptr, nullptr, stream));
            return linear_impl(dtype, M, N, K1, x1, ldx1, W1, bias1, out, ldo, out, ldo, BT_ACT_NONE, nullptr, nullptr, stream);
        }
#endif
        Seg2 seg;
        seg.x = (const bf16*)x2;
        seg.w = (const bf16*)W2;
        seg.bias = (const bf16*)bias2;
        seg.ldx = ldx2;
        seg.ldw = K2;
        seg.K = (int)K2;
        return linear_impl(dtype, M, N, K1, x1, ldx1, W1, bias1, residual, ldr, out, ldo, BT_ACT_NONE, nullptr, nullptr, stream,
                           scratch, scratch_bytes, &seg, nullptr, 0, true, nullptr, sk_clean);
    }
    const bool ok = dtype == BT_BF16 && M >= 1 && M <= 4 && N > 0 && K1 % GV_KC == 0 && K2 % 512 == 0 && K1 > 0 && K2 > 0 &&
                    ldx1 % 8 == 0 && ldx2 % 8 == 0 && aligned16(x1) && aligned16(x2) && aligned16(W1) && aligned16(W2) &&
                    !g_force_generic && g_gemm_variant != 22;
    const bool skinny = !ok && dtype == BT_BF16 && M > 4 && K2 > 0 && K2 % 32 == 0 && ldx2 % 8 == 0 && aligned16(x2) && aligned16(W2) &&
                        skinny_ok(M, N, K1, x1, ldx1, W1, bias1, residual, ldr, out, ldo) &&
                        (!bias2 || (reinterpret_cast<uintptr_t>(bias2) & 7u) == 0) && !g_force_generic && g_gemm_variant != 22;
    if (!ok && !skinny) {
        set_error("bt_linear2: only bf16 with 1 <= M <= 4 (K1 %% %d == 0, K2 %% 512 == 0), 5 <= M <= 64 (K %% 32 == 0, N %% 4 == 0) or "
                  "M > 64 (K %% 64 == 0, N %% 8 == 0, both biases), 16-byte aligned operands", GV_KC);
        return BT_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    EpiArgs ep{};
    ep.bias = (const bf16*)bias1;
    ep.res = (const bf16*)residual;
    ep.ldr = ldr;
    ep.out = (bf16*)out;
    ep.ldo = ldo;
    ep.M = (int)M;
    ep.N = (int)N;
    ep.act = BT_ACT_NONE;
    ep.q = QkvEpi{};
    Seg2 s2;
    s2.x = (const bf16*)x2;
    s2.w = (const bf16*)W2;
    s2.bias = (const bf16*)bias2;
    s2.ldx = ldx2;
    s2.ldw = K2;
    s2.K = (int)K2;
    ProfScope ps(BT_PROF_LINEAR_MFMA, s);
    if (skinny) {
        launch_skinny<false, false>(s, M, N, (const bf16*)x1, ldx1, (const bf16*)W1, K1, ep, K1, LnPro{}, s2);
        BT_LAUNCH_CHECK("bt_linear2 launch");
        return BT_OK;
    }
    const int rw = N >= 32768 ? 4 : (N >= 4096 ? 2 : 1);
#define BT_LAUNCH_GEMV2(MR_, RW_)                                                                                        \
    hipLaunchKernelGGL((gemv_bf16_kernel<MR_, RW_>), dim3((unsigned)ceil_div(N, 4 * RW_)), dim3(256), 0, s, (const bf16*)x1, \
                       ldx1, (const bf16*)W1, K1, ep, (int)K1, LnPro{}, s2)
#define BT_GEMV2_RW(MR_)                                   \
    do {                                                   \
        if (rw == 4) BT_LAUNCH_GEMV2(MR_, 4);              \
        else if (rw == 2) BT_LAUNCH_GEMV2(MR_, 2);         \
        else BT_LAUNCH_GEMV2(MR_, 1);                      \
    } while (0)
    if (M <= 1) BT_GEMV2_RW(1);
    else if (M <= 2) BT_GEMV2_RW(2);
    else BT_GEMV2_RW(4);
#undef BT_GEMV2_RW
#undef BT_LAUNCH_GEMV2
    BT_LAUNCH_CHECK("bt_linear2 launch");
    return BT_OK;
}

// ---- <= 4 rows: the layer's two x-consuming projections as one launch (gemv_qkv_h4_kernel); called by bt_decoder_forward -----
namespace bt {

bool few_rows_dual_ok(int64_t M, int64_t H, int64_t I, int head_dim, int rot) {
    if (rot != 32 || head_dim % 32 != 0 || g_force_generic || g_gemm_variant == 22 || !g_few_dual) return false;
    if (M >= 1 && M <= 4) return H % 512 == 0 && H <= GV_KC && I % 8 == 0;
    return M >= 5 && M <= 64 && H % 32 == 0 && I % 32 == 0;     // MFMA weight streamer (bt_decoder_forward uses it up to 16 rows)
}

int few_rows_qkv_h4(hipStream_t s, int64_t B, int q_len, int heads, int head_dim, const void* x, int64_t ldx, const void* ln1_w,
                    const void* ln1_b, const void* ln2_w, const void* ln2_b, float eps, const void* qkv_w, const void* qkv_b,
                    void* qkv, int64_t ld_qkv, void* kc, void* vc, int64_t n_max, const int32_t* pos0_dev, int pos0_host,
                    const int32_t* pos_table, const float* cos_tab, const float* sin_tab, const void* h4_w, const void* h4_b,
                    void* h4, int64_t I) {
    const int64_t H = (int64_t)heads * head_dim, M = B * q_len;
    EpiArgs epq{};
    epq.bias = (const bf16*)qkv_b;
    epq.out = (bf16*)qkv;
    epq.ldo = ld_qkv;
    epq.M = (int)M;
    epq.N = (int)(3 * H);
    epq.q.kc = (bf16*)kc;
    epq.q.vc = (bf16*)vc;
    epq.q.n_max = n_max;
    epq.q.pos0_dev = pos0_dev;
    epq.q.pos_table = pos_table;
    epq.q.cos_tab = cos_tab;
    epq.q.sin_tab = sin_tab;
    epq.q.pos0_host = pos0_host;
    epq.q.q_len = q_len;
    epq.q.heads = heads;
    epq.q.head_dim = head_dim;
    epq.q.enabled = 1;
    EpiArgs eph{};
    eph.bias = (const bf16*)h4_b;
    eph.out = (bf16*)h4;
    eph.ldo = I;
    eph.M = (int)M;
    eph.N = (int)I;
    eph.act = BT_ACT_GELU;
    const LnPro l1{(const bf16*)ln1_w, (const bf16*)ln1_b, eps}, l2{(const bf16*)ln2_w, (const bf16*)ln2_b, eps};
    ProfScope ps(BT_PROF_LINEAR_MFMA, s);
    if (M > 4) {   // 5 .. 64 rows: MFMA weight streamer, 32 weight rows per workgroup
        const int nq = (int)ceil_div(3 * H, SK_ROWS), nh = (int)ceil_div(I, SK_ROWS);
        const int mt = (int)ceil_div(M, 16);
#define BT_LAUNCH_SDUAL(MT_)                                                                                                  \
    hipLaunchKernelGGL((skinny_qkv_h4_kernel<MT_>), dim3(nq + nh), dim3(512), 0, s, (const bf16*)x, ldx, (const bf16*)qkv_w, epq, \
                       l1, nq, (const bf16*)h4_w, eph, l2, (int)H)
        if (mt <= 1) BT_LAUNCH_SDUAL(1);
        else if (mt == 2) BT_LAUNCH_SDUAL(2);
        else if (mt == 3) BT_LAUNCH_SDUAL(3);
        else BT_LAUNCH_SDUAL(4);
#undef BT_LAUNCH_SDUAL
        BT_LAUNCH_CHECK("fused QKV + h->4h streamer launch");
        return BT_OK;
    }
    const int nq = (int)ceil_div(3 * H / 2, 4);        // one wave per (n, n + 16) pair, 4 pairs per workgroup
    const int nh = (int)ceil_div(I, 4 * 2);            // RW = 2 rows per wave, 4 waves
    if (g_few_persist && M <= 2) {
        // ~2 resident workgroups per CU walk the row groups (x staged and normalised once per workgroup): groups per workgroup
        // chosen so that both sides divide evenly where they can (1.2B dims: 768 + 1024 groups -> 192 + 256 workgroups x 4)
        const int per = (int)std::max<int64_t>(1, ceil_div((int64_t)nq + nh, 2 * (int64_t)num_cus()));
        const int gq = (int)ceil_div(nq, per), gh = (int)ceil_div(nh, per);
#define BT_LAUNCH_PDUAL(MR_)                                                                                                         \
    hipLaunchKernelGGL((gemv_qkv_h4_persist_kernel<MR_>), dim3(gq + gh), dim3(256), 0, s, (const bf16*)x, ldx, (const bf16*)qkv_w, epq, \
                       l1, nq, gq, (const bf16*)h4_w, eph, l2, nh, (int)H)
        if (M <= 1) BT_LAUNCH_PDUAL(1);
        else BT_LAUNCH_PDUAL(2);
#undef BT_LAUNCH_PDUAL
        BT_LAUNCH_CHECK("persistent fused QKV + h->4h GEMV launch");
        return BT_OK;
    }
#define BT_LAUNCH_DUAL(MR_)                                                                                                  \
    hipLaunchKernelGGL((gemv_qkv_h4_kernel<MR_>), dim3(nq + nh), dim3(256), 0, s, (const bf16*)x, ldx, (const bf16*)qkv_w, epq, \
                       l1, nq, (const bf16*)h4_w, eph, l2, (int)H)
    if (M <= 1) BT_LAUNCH_DUAL(1);
    else if (M <= 2) BT_LAUNCH_DUAL(2);
    else BT_LAUNCH_DUAL(4);
#undef BT_LAUNCH_DUAL
    BT_LAUNCH_CHECK("fused QKV + h->4h GEMV launch");
    return BT_OK;
}

#ifndef BT_DEBUG_HOOKS
bool few_rows_attn_res_ok(int64_t, int, int, int, int64_t, int64_t, int, int64_t) { return false; }
int few_rows_attn_res(hipStream_t, int64_t, int, int, int64_t, const void*, const void*, const void*, const void*, int64_t, const void*,
                      const void*, int64_t, int, const void*, const void*, void*, int64_t) {
    set_error("fused local attention + residual GEMV: debug library only");
    return BT_ERR_UNSUPPORTED;
}
#else
bool few_rows_attn_res_ok(int64_t M, int q_len, int heads, int head_dim, int64_t H, int64_t I, int pos0_host, int64_t n_max) {
    if (!(g_few_attn && !g_force_generic && g_gemm_variant != 22 && M >= 1 && M <= 2 && q_len >= 1 && q_len <= 2 && M % q_len == 0 &&
          head_dim == 128 && H == (int64_t)heads * 128 && H % 512 == 0 && H <= GV_KC && I % GV_KC == 0 && pos0_host >= 0 &&
          pos0_host + q_len <= 12 && n_max <= 64))
        return false;
    // a wave's (sequence, head) items are spread over the first segment's K chunks, at most ATT_HPW per chunk
    const int64_t ipw = ceil_div((M / q_len) * heads, 4), nfc = I / GV_KC;
    return ceil_div(ipw, nfc) <= ATT_HPW;
}

int few_rows_attn_res(hipStream_t s, int64_t M, int q_len, int heads, int64_t I, const void* h4, const void* h_w, const void* h_b,
                      const void* qkv, int64_t ld_qkv, const void* kc, const void* vc, int64_t n_max, int pos0_host,
                      const void* dense_w, const void* dense_b, void* x, int64_t ldx) {
    const int64_t H = (int64_t)heads * 128;
    EpiArgs ep{};
    ep.bias = (const bf16*)h_b;
    ep.res = (const bf16*)x;
    ep.ldr = ldx;
    ep.out = (bf16*)x;
    ep.ldo = ldx;
    ep.M = (int)M;
    ep.N = (int)H;
    ep.act = BT_ACT_NONE;
    ep.q = QkvEpi{};
    Seg2 s2;
    s2.x = nullptr;                       // produced in LDS
    s2.w = (const bf16*)dense_w;
    s2.bias = (const bf16*)dense_b;
    s2.ldx = H;
    s2.ldw = H;
    s2.K = (int)H;
    LocalAtt la;
    la.qkv = (const bf16*)qkv;
    la.kc = (const bf16*)kc;
    la.vc = (const bf16*)vc;
    la.ld_qkv = ld_qkv;
    la.n_max = n_max;
    la.pos0 = pos0_host;
    la.q_len = q_len;
    la.heads = heads;
    la.scale = 1.0f / sqrtf(128.0f);
    ProfScope ps(BT_PROF_LINEAR_MFMA, s);
    const unsigned grid = (unsigned)ceil_div(H, 4);
#define BT_LAUNCH_AR(MR_, U_)                                                                                                    \
    hipLaunchKernelGGL((gemv_attn_res_kernel<MR_, U_>), dim3(grid), dim3(256), 0, s, (const bf16*)h4, I, (const bf16*)h_w, I, ep, \
                       (int)I, s2, la)
    const bool u2 = pos0_host + q_len <= 8;
    if (M <= 1) { if (u2) BT_LAUNCH_AR(1, 2); else BT_LAUNCH_AR(1, 3); }
    else { if (u2) BT_LAUNCH_AR(2, 2); else BT_LAUNCH_AR(2, 3); }
#undef BT_LAUNCH_AR
    BT_LAUNCH_CHECK("fused local attention + residual GEMV launch");
    return BT_OK;
}
#endif  // BT_DEBUG_HOOKS

}  // namespace bt

// ---- mid-batch path (3 .. 512 rows): split-K slabs + fused reduces; called by bt_decoder_forward (engine.hip) ------------------
namespace bt {

// K-slice length: the largest of {8192 .. 256} that divides both operand pairs' K and still gives every CU a workgroup
static int mid_slice_len(int64_t tiles, int64_t K1, int64_t K2) {
    int best = 0;
    for (int ks = 8192; ks >= 256; ks >>= 1) {
        if (K1 % ks != 0 || (K2 > 0 && K2 % ks != 0)) continue;
        if (best != 0 && ks < 512) break;   // slices shorter than 8 K-steps are all prologue: only when nothing longer divides K
        best = ks;
        if (tiles * (K1 / ks + (K2 > 0 ? K2 / ks : 0)) >= 256) break;   // one workgroup per CU (two fit) is enough to stream W
    }
    return best;
}

static bool mid_tall_rows(int64_t M);
static void tall_slices(int tiles, int nk1, int nk2, int* s1, int* s2);

bool mid_rows_ok(int64_t M, int64_t H, int64_t I, int head_dim, int rot) {
    // (bt_decoder_forward sends stacks of <= g_few_rows_max rows to the weight-streaming GEMV / MFMA-streamer launches first)
    // (257 .. 512 rows, round 5: only with the 256 x 128 slab kernel -- the 64 x 64 kernels lose to the large-M launches there)
    return M >= 1 && (M <= 256 || (mid_tall_rows(M) && M <= 512)) && rot == 32 && head_dim % 32 == 0 && H % 256 == 0 && I % 256 == 0 &&
           H <= 8192 && !g_force_generic && g_gemm_variant == 0 && g_mid_rows;
}

int64_t mid_slab_bytes(int64_t M, int64_t H, int64_t I) {
    // worst case over the three GEMMs of a layer: slices x M x N fp32 (slice counts from mid_slice_len)
    int64_t worst = 0;
    const int64_t tm = ceil_div(M, 64);
    const int64_t shapes[3][3] = {{3 * H, H, 0}, {I, H, 0}, {H, I, H}};
    for (auto& sh : shapes) {
        const int ks = mid_slice_len(tm * ceil_div(sh[0], 64), sh[1], sh[2]);
        if (ks == 0) return -1;
        const int64_t sl = sh[1] / ks + (sh[2] ? sh[2] / ks : 0);
        worst = std::max(worst, sl * M * sh[0] * 4);
    }
    {   // the dual QKV + h->4h launch keeps both problems' slabs at once
        const int ks = mid_slice_len(tm * (ceil_div(3 * H, 64) + ceil_div(I, 64)), H, 0);
        if (ks > 0) worst = std::max(worst, std::max<int64_t>(H / ks, 2) * M * (3 * H + I) * 4);   // (two slices are forced up to 224 rows)
    }
    if (mid_tall_rows(M) && H % 128 == 0 && I % 128 == 0) {   // 256 x 128 tiles: K ranges sized to fill the chip
        int s1 = 0, s2 = 0;
        const int tmr = (int)ceil_div(M, 256);
        tall_slices(tmr * (int)(H / 128), (int)(I / 64), (int)(H / 64), &s1, &s2);
        worst = std::max(worst, (int64_t)(s1 + s2) * M * H * 4);
        tall_slices(tmr * (int)((3 * H + I) / 128), (int)(H / 64), 0, &s1, &s2);
        worst = std::max(worst, (int64_t)s1 * M * (3 * H + I) * 4);
    }
    return worst;
}

// ---- 97 .. 512 rows: 256 x 128 tiles, K ranges per workgroup (gemm_bf16_mfma256x128_slab_kernel) ----
static bool mid_tall_rows(int64_t M) { return g_mid_tall != 0 && M <= g_mid_rows_max && M > (g_mid_tall >= 2 ? 64 : 96); }

// slices of the (up to two) operand pairs of one problem: about g_mid_tall_wgs workgroups in all, >= 4 K-tiles per slice, the second
// pair's share in proportion to its K
static void tall_slices(int tiles, int nk1, int nk2, int* s1, int* s2) {
    const int total = std::max(1, (nk2 > 0 ? g_mid_tall_wgs_res : g_mid_tall_wgs) / std::max(tiles, 1));
    if (nk2 <= 0) {
        *s1 = std::max(1, std::min(total, nk1 / 4));
        *s2 = 0;
        return;
    }
    int a = (int)((int64_t)total * nk1 / (nk1 + nk2));
    a = std::max(1, std::min(a, std::max(1, nk1 / 4)));
    int b = std::max(1, std::min(total - a, std::max(1, nk2 / 4)));
    *s1 = a;
    *s2 = b;
}

static bool tall_operand_ok(const void* x, int64_t ld, int64_t K) {
    return x && aligned16(x) && ld % 8 == 0 && K % 64 == 0 && K >= 256 && ld * 2 * 512 < ((int64_t)1 << 31);
}

// 257 .. 512 rows are on the mid-batch path ONLY with the 256 x 128 slab kernel (the 64 x 64 slab kernels lose to the large-M launch
// sequence there, and their dual launch would need more tile counters than MID_COUNTER_BYTES holds): the per-call `tall` predicates of
// mid_qkv_gelu / mid_res_ln restated for one layer's weights, so that bt_decoder_forward decides ONCE, before the first launch, and a
// stack whose weights fail them (an unaligned view, a row pitch beyond the 32-bit lane offsets) takes the large-M sequence
bool mid_tall_layer_ok(int64_t H, int64_t I, const void* qkv_w, const void* h4_w, const void* h_w, const void* dense_w) {
    return H % 128 == 0 && I % 128 == 0 && (3 * H) % 128 == 0 && tall_operand_ok(qkv_w, H, H) && tall_operand_ok(h4_w, H, H) &&
           tall_operand_ok(h_w, I, I) && tall_operand_ok(dense_w, H, H);
}

static int tall_launch(hipStream_t s, int64_t M, const TallSide& g0, const TallSide* g1, const TallFuse* fuse = nullptr) {
    TallArgs ta{};
    ta.g[0] = g0;
    ta.g[1] = g1 ? *g1 : g0;
    const int tiles_m = (int)ceil_div(M, 256);
    const int wg0 = tiles_m * (g0.N / 128) * (g0.s1 + g0.s2);
    const int wg1 = g1 ? tiles_m * (g1->N / 128) * (g1->s1 + g1->s2) : 0;
    if (fuse) {
        const int grid = fuse->half_mode ? 2 * (wg0 + wg1) : wg0 + wg1;
#ifdef BT_DEBUG_HOOKS
        if (fuse->half_mode == 2) {
            hipLaunchKernelGGL(gemm_bf16_mid_stream_kernel, dim3(grid), dim3(512), 0, s, ta, (int)M, wg0, *fuse);
            BT_LAUNCH_CHECK("mid-batch stream-form fused GEMM launch");
            return BT_OK;
        }
#endif
        hipLaunchKernelGGL((gemm_bf16_mfma256x128_slab_kernel<true, true>), dim3(grid), dim3(512), 0, s, ta, (int)M, wg0, *fuse);
        BT_LAUNCH_CHECK("mid-batch 256 x 128 fused GEMM launch");
        return BT_OK;
    }
    if (g_mid_tall_wt)
        hipLaunchKernelGGL(gemm_bf16_mfma256x128_slab_kernel<true>, dim3(wg0 + wg1), dim3(512), 0, s, ta, (int)M, wg0);
    else
        hipLaunchKernelGGL(gemm_bf16_mfma256x128_slab_kernel<false>, dim3(wg0 + wg1), dim3(512), 0, s, ta, (int)M, wg0);
    BT_LAUNCH_CHECK("mid-batch 256 x 128 slab GEMM launch");
    return BT_OK;
}

static bool mid_fuse_now(int64_t M, int64_t N, int64_t K) {
    if (g_mid_fuse == 2) return true;
    return g_mid_fuse == 1 && mid_slice_len(ceil_div(M, 64) * ceil_div(N, 64), K, 0) == K;
}

// slabs[slices][M][N] = A . W^T (K1)  [+ A2 . W2^T (K2)]   -- returns the slice count through *slices.
// fused_ep != nullptr: in-launch reduction + that epilogue (no slabs at all when one slice is enough); `counters` = one zeroed
// int per 64 x 64 tile.
static int mid_gemm(hipStream_t s, int64_t M, int64_t N, const void* A, int64_t lda, const void* W, int64_t K1, const void* A2,
                    int64_t lda2, const void* W2, int64_t K2, void* slabs, int64_t slab_bytes, int* slices,
                    const EpiArgs* fused_ep = nullptr, int* counters = nullptr) {
    const int tiles_m = (int)ceil_div(M, 64), tiles_n = (int)ceil_div(N, 64);
    const int ks = mid_slice_len((int64_t)tiles_m * tiles_n, K1, K2);
    BT_CHECK_ARG(ks > 0, "mid-batch GEMM: K = %lld / %lld has no common slice length", (long long)K1, (long long)K2);
    const int sl1 = (int)(K1 / ks), sl2 = K2 > 0 ? (int)(K2 / ks) : 0;
    if ((int64_t)(sl1 + sl2) * M * N * 4 > slab_bytes) {
        set_error("mid-batch GEMM: slab workspace too small (%lld bytes for %d slices of %lld x %lld)", (long long)slab_bytes,
                  sl1 + sl2, (long long)M, (long long)N);
        return BT_ERR_WORKSPACE;
    }
    EpiArgs ep{};
    ep.M = (int)M;
    ep.N = (int)N;
    ep.slab_wt = g_mid_wt ? 1 : 0;
    EpiArgs fep{};
    if (fused_ep) {
        fep = *fused_ep;
        fep.slab_wt = ep.slab_wt;
    }
    Seg2 s2{};
    s2.x = (const bf16*)A2;
    s2.w = (const bf16*)W2;
    s2.ldx = lda2;
    s2.ldw = K2;
    s2.K = (int)K2;
    ProfScope ps(BT_PROF_LINEAR_MFMA, s);
    if (fused_ep && sl1 + sl2 == 1 && K2 == 0 && M >= 128 && g_mid_tile != 0) {
        // one K slice, >= 128 rows: a taller tile halves the L2 -> LDS traffic per flop of the 64 x 64 tile (A/B knob, round 4)
        const int tm2 = (int)ceil_div(M, 128);
        if (g_mid_tile == 1)
            hipLaunchKernelGGL((gemm_bf16_mfma_kernel<128, 64, false, 3>), dim3(tm2 * tiles_n), dim3(256), 0, s, (const bf16*)A, lda,
                               (const bf16*)W, K1, *fused_ep, (int)K1, tm2, tiles_n, nullptr, Seg2{});
        else
            hipLaunchKernelGGL((gemm_bf16_mfma_kernel<128, 128, false, 2>), dim3(tm2 * (int)ceil_div(N, 128)), dim3(256), 0, s,
                               (const bf16*)A, lda, (const bf16*)W, K1, *fused_ep, (int)K1, tm2, (int)ceil_div(N, 128), nullptr, Seg2{});
        BT_LAUNCH_CHECK("mid-batch GEMM launch (tall tile)");
        *slices = 1;
        return BT_OK;
    }
    if (fused_ep)
        hipLaunchKernelGGL((gemm_bf16_mfma_kernel<64, 64, true, 4, true>), dim3(tiles_m * tiles_n, sl1 + sl2), dim3(256), 0, s,
                           (const bf16*)A, lda, (const bf16*)W, K1, fep, ks, tiles_m, tiles_n, (float*)slabs, s2, sl1,
                           counters);
    else
        hipLaunchKernelGGL((gemm_bf16_mfma_kernel<64, 64, true, 4>), dim3(tiles_m * tiles_n, sl1 + sl2), dim3(256), 0, s,
                           (const bf16*)A, lda, (const bf16*)W, K1, ep, ks, tiles_m, tiles_n, (float*)slabs, s2, sl1);
    BT_LAUNCH_CHECK("mid-batch GEMM launch");
    *slices = sl1 + sl2;
    return BT_OK;
}

int mid_qkv(hipStream_t s, int64_t B, int q_len, int heads, int head_dim, const void* x, int64_t ldx, const void* w,
            const void* bias, void* qkv, int64_t ld_qkv, void* kc, void* vc, int64_t n_max, const int32_t* pos0_dev, int pos0_host,
            const int32_t* pos_table, const float* cos_tab, const float* sin_tab, void* slabs, int64_t slab_bytes, int* counters) {
    const int64_t H = (int64_t)heads * head_dim, M = B * q_len;
    int slices = 0;
    EpiArgs ep{};
    ep.bias = (const bf16*)bias;
    ep.out = (bf16*)qkv;
    ep.ldo = ld_qkv;
    ep.M = (int)M;
    ep.N = (int)(3 * H);
    ep.q.kc = (bf16*)kc;
    ep.q.vc = (bf16*)vc;
    ep.q.n_max = n_max;
    ep.q.pos0_dev = pos0_dev;
    ep.q.pos_table = pos_table;
    ep.q.cos_tab = cos_tab;
    ep.q.sin_tab = sin_tab;
    ep.q.pos0_host = pos0_host;
    ep.q.q_len = q_len;
    ep.q.heads = heads;
    ep.q.head_dim = head_dim;
    ep.q.enabled = 1;
    if (counters && mid_fuse_now(M, 3 * H, H))   // reduction, bias, RoPE and the scatter all inside the GEMM launch
        return mid_gemm(s, M, 3 * H, x, ldx, w, H, nullptr, 0, nullptr, 0, slabs, slab_bytes, &slices, &ep, counters);
    BT_TRY(mid_gemm(s, M, 3 * H, x, ldx, w, H, nullptr, 0, nullptr, 0, slabs, slab_bytes, &slices));
    ProfScope ps(BT_PROF_ROPE, s);
    hipLaunchKernelGGL(mid_reduce_qkv_kernel, dim3((unsigned)ceil_div(M * (3 * H / 8), 256)), dim3(256), 0, s,
                       (const float*)slabs, slices, ep);
    BT_LAUNCH_CHECK("mid-batch QKV reduce launch");
    return BT_OK;
}

// QKV projection (LN1 rows) and MLP up-projection (LN2 rows) of a layer as ONE split-K launch + ONE reduce launch
int mid_qkv_gelu(hipStream_t s, int64_t B, int q_len, int heads, int head_dim, const void* x1, const void* qkv_w, const void* qkv_b,
                 void* qkv, int64_t ld_qkv, void* kc, void* vc, int64_t n_max, const int32_t* pos0_dev, int pos0_host,
                 const int32_t* pos_table, const float* cos_tab, const float* sin_tab, const void* x2, const void* h4_w,
                 const void* h4_b, void* h4, int64_t I, void* slabs, int64_t slab_bytes, int* counters, bool* done, void* att,
                 int64_t ld_att, bool* attn_done) {
    const int64_t H = (int64_t)heads * head_dim, M = B * q_len, N1 = 3 * H, N2 = I, K = H;
    *done = false;
    if (attn_done) *attn_done = false;
    if (!g_mid_dual || g_mid_fuse == 2 || N1 % 64 != 0 || N2 % 64 != 0) return BT_OK;
    const int tiles_m = (int)ceil_div(M, 64), t1 = (int)(N1 / 64), t2 = (int)(N2 / 64);
    // 97 .. 512 rows: both problems on 256 x 128 tiles, two K ranges each up to 256 rows (112 column tiles x 2 = 224 workgroups for the
    // 1.2B layer), one above (two row tiles)
    const bool tall = mid_tall_rows(M) && N1 % 128 == 0 && N2 % 128 == 0 && tall_operand_ok(x1, H, K) && tall_operand_ok(x2, H, K) &&
                      tall_operand_ok(qkv_w, K, K) && tall_operand_ok(h4_w, K, K);
    int tall_s = 0, unused = 0;
    if (tall) tall_slices((int)(ceil_div(M, 256) * ((N1 + N2) / 128)), (int)(K / 64), 0, &tall_s, &unused);
    const int ks = tall ? (int)K : mid_slice_len((int64_t)tiles_m * (t1 + t2), K, 0);
    if (ks <= 0) return BT_OK;
    int slices = tall ? tall_s : (int)(K / ks);
    int ks_use = ks;
    // one K slice already fills the chip (M >~ 100).  A shared launch with both FUSED epilogues only lengthens the critical path there
    // (measured: B = 128 62.2k -> 59.3k, B = 256 78.7k -> 72.3k tok/s, profiles/r4_bench_dual_fused_mode_b*.json), but the 4-launch
    // structure itself still pays up to ~200 rows when the shared launch is cut into TWO K slices (slabs + the merged reduce with the
    // attention folded in): B = 96 +6.8 %, 128 +5.7 %, 192 +3.1 %, 256 -0.2 % (profiles/r4_ab_mid_force2.jsonl) -> forced up to 224 rows
    if (!tall && slices < 2 && g_mid_force2 && M <= 224 && K % 128 == 0) {
        slices = 2;
        ks_use = (int)(K / 2);
    }
    if ((slices < 2 && !tall) || g_mid_fuse == 0) return BT_OK;
    if ((int64_t)slices * M * (N1 + N2) * 4 > slab_bytes) return BT_OK;
    EpiArgs e1{};
    e1.bias = (const bf16*)qkv_b;
    e1.out = (bf16*)qkv;
    e1.ldo = ld_qkv;
    e1.M = (int)M;
    e1.N = (int)N1;
    e1.q.kc = (bf16*)kc;
    e1.q.vc = (bf16*)vc;
    e1.q.n_max = n_max;
    e1.q.pos0_dev = pos0_dev;
    e1.q.pos_table = pos_table;
    e1.q.cos_tab = cos_tab;
    e1.q.sin_tab = sin_tab;
    e1.q.pos0_host = pos0_host;
    e1.q.q_len = q_len;
    e1.q.heads = heads;
    e1.q.head_dim = head_dim;
    e1.q.enabled = 1;
    EpiArgs e2{};
    e2.bias = (const bf16*)h4_b;
    e2.out = (bf16*)h4;
    e2.ldo = I;
    e2.M = (int)M;
    e2.N = (int)N2;
    e2.act = BT_ACT_GELU;
    e1.slab_wt = e2.slab_wt = g_mid_wt ? 1 : 0;
    float* p1 = (float*)slabs;
    float* p2 = p1 + (int64_t)slices * M * N1;
    DualArgs da;
    da.g[0] = GemmSide{(const bf16*)x1, (const bf16*)qkv_w, H, K, p1, counters};
    da.g[1] = GemmSide{(const bf16*)x2, (const bf16*)h4_w, H, K, p2, counters ? counters + tiles_m * t1 : nullptr};
    da.ep[0] = e1;
    da.ep[1] = e2;
    da.wt = g_mid_wt ? 1 : 0;
    if (tall) {
        ProfScope ps(BT_PROF_LINEAR_MFMA, s);
        TallSide g0{}, g1{};
        g0.A = (const bf16*)x1; g0.W = (const bf16*)qkv_w; g0.lda = H; g0.ldw = K; g0.partial = p1; g0.N = (int)N1;
        g0.nk1 = (int)(K / 64); g0.s1 = slices;
        g1.A = (const bf16*)x2; g1.W = (const bf16*)h4_w; g1.lda = H; g1.ldw = K; g1.partial = p2; g1.N = (int)N2;
        g1.nk1 = (int)(K / 64); g1.s1 = slices;
        // round 6: epilogues fused into this launch (hand-off between the two K ranges of a tile, or none above 256 rows): no reduce
        // launch; the attention of either decoder follows as its own launch (the caller sees attn_done == false)
        const int nk = (int)(K / 64);
        const int64_t n_tiles = ceil_div(M, 256) * ((N1 + N2) / 128);
        // 129 .. 256 rows (one row tile, two K ranges in the slab form): with fused epilogues each workgroup takes ONE 128-row half of a
        // column tile over the whole K instead -- same 224 workgroups, no hand-off (g_mid_tall_fuse == 3: off, A/B)
        const bool halves = slices == 2 && M > 128 && M <= 256 && n_tiles % 8 == 0 && g_mid_tall_fuse != 3 && g_mid_tall_fuse != 2;   // (5, debug build: the stream-form experiment)
        if (g_mid_tall_fuse && counters && (slices == 1 || halves || (slices == 2 && g_mid_tall_fuse == 2 && nk >= 8)) && n_tiles < TF_STICKY && head_dim % 32 == 0 &&
            ld_qkv % 4 == 0 && I % 4 == 0 && (reinterpret_cast<uintptr_t>(qkv) & 7u) == 0 && (reinterpret_cast<uintptr_t>(h4) & 7u) == 0 &&
            (reinterpret_cast<uintptr_t>(kc) & 7u) == 0 && (reinterpret_cast<uintptr_t>(vc) & 7u) == 0 &&
            (!qkv_b || (reinterpret_cast<uintptr_t>(qkv_b) & 7u) == 0) && (!h4_b || (reinterpret_cast<uintptr_t>(h4_b) & 7u) == 0)) {
            TallFuse tf{};
            tf.ep[0] = e1;
            tf.ep[1] = e2;
            tf.flags = reinterpret_cast<uint32_t*>(counters);
            tf.first_nk = std::max(1, nk / 2 - 2);         // the finishing workgroup runs ~4 K-tiles longer: the partner's tile is there when it looks
            if (halves) {
                tf.half_mode = 1;
#ifdef BT_DEBUG_HOOKS       // g_mid_tall_fuse == 5: the stream-form experiment (W global -> registers) where its unrolled groups fit
                if (g_mid_tall_fuse == 5 && nk % MS_PW == 0 && nk >= 2 * MS_PW && aligned16(x1) && aligned16(x2) && aligned16(qkv_w) && aligned16(h4_w) &&
                    H % 8 == 0 && K % 8 == 0)
                    tf.half_mode = 2;
#endif
                g0.s1 = g1.s1 = 1;                          // one K range per (tile, half)
            }
#ifdef BT_DEBUG_HOOKS
            tf.tl = reinterpret_cast<uint64_t*>(g_timeline);
            tf.dbg = g_mid_tall_dbg;
#endif
            BT_TRY(tall_launch(s, M, g0, &g1, &tf));
            *done = true;
            return BT_OK;
        }
        BT_TRY(tall_launch(s, M, g0, &g1));
    } else {
        ProfScope ps(BT_PROF_LINEAR_MFMA, s);
        // slab mode: plain stores of the fp32 partial tiles (the epilogue arguments only carry M and N)
        hipLaunchKernelGGL(gemm_bf16_mfma_dual_kernel, dim3(tiles_m * (t1 + t2), slices), dim3(256), 0, s, da, ks_use, tiles_m, t1, t2);
        BT_LAUNCH_CHECK("mid-batch dual GEMM launch");
    }
    *done = true;
    const int nb2 = (int)ceil_div(M * (N2 / 8), 256);
    // token decoder (local window at a host-known position, no mask): the attention rides in the same launch
    if (att && attn_done && g_mid_attn && !pos0_dev && head_dim == 128 && q_len <= 2 && pos0_host >= 0 && pos0_host + q_len <= 12 &&
        n_max >= pos0_host + q_len && aligned16(kc) && aligned16(vc) && aligned16(att) && ld_att % 8 == 0) {
        ProfScope ps(BT_PROF_ATTN_ROWS, s);
        const int nb1 = (int)ceil_div(B * heads, 4);
        const float scale = 1.0f / sqrtf(128.0f);
        if (pos0_host + q_len <= 8)
            hipLaunchKernelGGL((mid_reduce_qkv_attn_gelu_kernel<8>), dim3(nb1 + nb2), dim3(256), 0, s, p1, e1, nb1, p2, e2, slices,
                               (bf16*)att, ld_att, scale);
        else
            hipLaunchKernelGGL((mid_reduce_qkv_attn_gelu_kernel<12>), dim3(nb1 + nb2), dim3(256), 0, s, p1, e1, nb1, p2, e2, slices,
                               (bf16*)att, ld_att, scale);
        BT_LAUNCH_CHECK("mid-batch QKV + attention + GELU reduce launch");
        *attn_done = true;
        return BT_OK;
    }
    ProfScope ps(BT_PROF_ROPE, s);
    const int nb1 = (int)ceil_div(M * (N1 / 8), 256);
    hipLaunchKernelGGL(mid_reduce_qkv_gelu_kernel, dim3(nb1 + nb2), dim3(256), 0, s, p1, e1, nb1, p2, e2, slices);
    BT_LAUNCH_CHECK("mid-batch QKV + GELU reduce launch");
    return BT_OK;
}

int mid_gelu(hipStream_t s, int64_t M, int64_t N, int64_t K, const void* x, int64_t ldx, const void* w, const void* bias, void* out,
             int64_t ldo, void* slabs, int64_t slab_bytes, int* counters) {
    int slices = 0;
    EpiArgs ep{};
    ep.bias = (const bf16*)bias;
    ep.out = (bf16*)out;
    ep.ldo = ldo;
    ep.M = (int)M;
    ep.N = (int)N;
    ep.act = BT_ACT_GELU;
    if (counters && mid_fuse_now(M, N, K))
        return mid_gemm(s, M, N, x, ldx, w, K, nullptr, 0, nullptr, 0, slabs, slab_bytes, &slices, &ep, counters);
    BT_TRY(mid_gemm(s, M, N, x, ldx, w, K, nullptr, 0, nullptr, 0, slabs, slab_bytes, &slices));
    ProfScope ps(BT_PROF_LINEAR_MFMA, s);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)ceil_div(M * (N / 8), 256)), dim3(256), 0, s, (const float*)slabs,
                       slices, ep);
    BT_LAUNCH_CHECK("mid-batch GELU reduce launch");
    return BT_OK;
}

// x += x1 . W1^T + b1 + x2 . W2^T + b2 ; then (optionally) the next LayerNorm pair of the new rows
int mid_res_ln(hipStream_t s, int64_t M, int64_t N, int64_t K1, const void* x1, int64_t ldx1, const void* w1, const void* b1,
               int64_t K2, const void* x2, int64_t ldx2, const void* w2, const void* b2, void* x, int64_t ldx, const void* ln1_w,
               const void* ln1_b, void* ln1_out, const void* ln2_w, const void* ln2_b, void* ln2_out, int64_t ld_ln, float eps,
               void* slabs, int64_t slab_bytes) {
    int slices = 0;
    int ts1 = 0, ts2 = 0;
    const bool tall = mid_tall_rows(M) && N % 128 == 0 && tall_operand_ok(x1, ldx1, K1) && tall_operand_ok(w1, K1, K1) &&
                      (K2 == 0 || (tall_operand_ok(x2, ldx2, K2) && tall_operand_ok(w2, K2, K2)));
    if (tall) tall_slices((int)(ceil_div(M, 256) * (N / 128)), (int)(K1 / 64), (int)(K2 / 64), &ts1, &ts2);
    if (tall && (int64_t)(ts1 + ts2) * M * N * 4 <= slab_bytes) {
        ProfScope ps(BT_PROF_LINEAR_MFMA, s);
        TallSide g0{};
        g0.A = (const bf16*)x1; g0.W = (const bf16*)w1; g0.lda = ldx1; g0.ldw = K1; g0.nk1 = (int)(K1 / 64); g0.s1 = ts1;
        g0.A2 = (const bf16*)x2; g0.W2 = (const bf16*)w2; g0.lda2 = ldx2; g0.ldw2 = K2; g0.nk2 = (int)(K2 / 64); g0.s2 = ts2;
        g0.partial = (float*)slabs; g0.N = (int)N;
        BT_TRY(tall_launch(s, M, g0, nullptr));
        slices = ts1 + ts2;
    } else {
        BT_TRY(mid_gemm(s, M, N, x1, ldx1, w1, K1, x2, ldx2, w2, K2, slabs, slab_bytes, &slices));
    }
    EpiArgs ep{};
    ep.bias = (const bf16*)b1;
    ep.res = (const bf16*)x;
    ep.ldr = ldx;
    ep.out = (bf16*)x;
    ep.ldo = ldx;
    ep.M = (int)M;
    ep.N = (int)N;
    Ln2Out ln{};
    ln.w1 = (const bf16*)ln1_w;
    ln.b1 = (const bf16*)ln1_b;
    ln.o1 = (bf16*)ln1_out;
    ln.w2 = (const bf16*)ln2_w;
    ln.b2 = (const bf16*)ln2_b;
    ln.o2 = (bf16*)ln2_out;
    ln.ldo1 = ln.ldo2 = ld_ln;
    ln.eps = eps;
    ProfScope ps(BT_PROF_LAYERNORM, s);
    hipLaunchKernelGGL(mid_reduce_res_ln_kernel, dim3((unsigned)M), dim3(256), 0, s, (const float*)slabs, slices, ep,
                       (const bf16*)b2, ln);
    BT_LAUNCH_CHECK("mid-batch residual + LayerNorm reduce launch");
    return BT_OK;
}

}  // namespace bt

#ifdef BT_DEBUG_HOOKS
// test / A-B hooks (libbt_hip_dbg.so only): force the generic kernel for bf16, pick a kernel variant, cap the persistent grid
extern "C" void bt_debug_force_generic_gemm(int on) { g_force_generic = on != 0; }
extern "C" void bt_debug_gemm_variant(int v) { g_gemm_variant = v; }
extern "C" void bt_debug_stream_k(int on) { g_sk = on; }
extern "C" void bt_debug_stream_k_drop(int vw, int spin) { g_sk_drop_vw = vw; g_sk_spin = spin; }
extern "C" void bt_debug_timeline_buffer(void* buf) { g_timeline = buf; }
extern "C" void bt_debug_serial_epilogues(int on) { g_serial_epi = on != 0; }
extern "C" void bt_debug_sk_diag_buffer(void* buf) { g_sk_diag = (uint32_t*)buf; }
extern "C" void bt_debug_gemm_max_wg(int n) { g_max_wg = n; }
extern "C" void bt_debug_skinny_rows(int n) { g_skinny_rows = n; }
extern "C" void bt_debug_small_stages(int n) { g_small_stages = n; }
extern "C" void bt_debug_mid_rows(int on) { g_mid_rows = on != 0; }
extern "C" void bt_debug_mid_tall(int mode) { g_mid_tall = mode; }
extern "C" void bt_debug_mid_tall_wgs(int wgs) { g_mid_tall_wgs = wgs > 0 ? wgs : 256; }
extern "C" void bt_debug_mid_tall_wt(int on) { g_mid_tall_wt = on != 0; }
extern "C" void bt_debug_mid_tall_fuse(int mode) { g_mid_tall_fuse = mode; }
extern "C" void bt_debug_mid_tall_dbg(int bits) { g_mid_tall_dbg = bits; }
extern "C" void bt_debug_mid_wt(int on) { g_mid_wt = on != 0; }
extern "C" void bt_debug_mid_rows_max(int rows) { g_mid_rows_max = rows; }
extern "C" void bt_debug_mid_tall_wgs_res(int wgs) { g_mid_tall_wgs_res = wgs > 0 ? wgs : 256; }
extern "C" void bt_debug_few_dual(int on) { g_few_dual = on != 0; }
extern "C" void bt_debug_few_persist(int on) { g_few_persist = on != 0; }
extern "C" void bt_debug_few_attn(int on) { g_few_attn = on != 0; }
extern "C" void bt_debug_mid_fuse(int mode) { g_mid_fuse = mode; }
extern "C" void bt_debug_mid_dual(int on) { g_mid_dual = on != 0; }
extern "C" void bt_debug_mid_attn(int on) { g_mid_attn = on != 0; }
extern "C" void bt_debug_mid_tile(int v) { g_mid_tile = v; }
extern "C" void bt_debug_mid_force2(int on) { g_mid_force2 = on != 0; }
extern "C" void bt_debug_no_qkv_fusion(int on) { g_no_qkv_fusion = on != 0; }
#endif
