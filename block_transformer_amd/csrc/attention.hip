// Attention over the Block Transformer's two KV caches.
//
//  rope_kv_append_kernel<T>   partial RoPE on q (in place) and k, append (k_rot, v) to a cache [B][heads][n_max][d].
//                             Replaces HF apply_rotary_pos_emb + torch.cat cache growth: the cache is preallocated and a
//                             step writes ONE row per (b, head) instead of re-copying n rows (C4/C5).
//  attend_rows_kernel<T,D,NW> row-wise streaming attention: one workgroup per (b, head, query row) [x split].  This is
//                             the GLOBAL decode step (q = 1 against n cached block positions): each (b, head) slab is a
//                             contiguous n x d stream read with 16-byte loads, 16 lanes per 128-dim bf16 row (4 rows per
//                             wave-instruction), scores reduced with in-wave shuffles into an LDS score row, softmax with
//                             wave shuffle + LDS block reduce, then the V stream with per-lane fp32 accumulators.
//                             HBM-bound: algorithmic bytes = 2 * n * d * sizeof(T) per (b, head) per layer.  MFMA does not
//                             apply: with q = 1 and no KV sharing across heads every K/V byte is used for exactly one
//                             multiply-add (arithmetic intensity 1 flop/byte).  The same kernel with NW = 1 serves the
//                             token decoder's local window (<= E+L-1 keys), whose K/V rows sit in L2/LDS-sized tiles.
//  flash_prefill_kernel<D>    bf16 MFMA flash attention for the block-decoder prefill (q_len = n): S^T = K Q^T and
//                             O = P V with v_mfma_f32_16x16x32_bf16, online softmax in registers (the swapped product
//                             leaves a query's scores in one lane column, so the row max/sum are 2 shuffles), V tile
//                             transposed through LDS.
//  attn_combine_kernel<T>     merges split-N partials (flash-decoding style) when B*heads is too small to fill 256 CUs.
#include "bt_common.h"

#include <algorithm>

namespace bt {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// ---- RoPE + KV append ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rope_kv_append_kernel(T* __restrict__ qkv, int64_t ld_qkv, T* __restrict__ kc,
                                                            T* __restrict__ vc, int64_t n_max, int64_t rows, int q_len,
                                                            int heads, int D, int rot, const int32_t* __restrict__ pos0_dev,
                                                            int pos0_host, const int32_t* __restrict__ pos_table,
                                                            const float* __restrict__ cos_tab,
                                                            const float* __restrict__ sin_tab) {
    constexpr int V = Elem<T>::VEC;
    const int cpr = heads * 3 * D / V;  // 16-byte chunks per qkv row
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= rows * cpr) return;
    const int64_t row = gid / cpr;
    const int c = (int)(gid - row * cpr);
    const int col = c * V;
    const int head = col / (3 * D);
    const int within = col - head * 3 * D;
    const int part = within / D;  // 0 q, 1 k, 2 v
    const int e0 = within - part * D;
    const int64_t b = row / q_len;
    const int t = (int)(row - b * q_len);
    const int pos0 = pos0_dev ? *pos0_dev : pos0_host;
    const int cache_row = pos0 + t;
    const int rpos = pos_table ? pos_table[pos0 + t] : pos0 + t;
    const int half = rot >> 1;
    T* src = qkv + row * ld_qkv + col;
    const int64_t cidx = ((b * heads + head) * n_max + cache_row) * D + e0;
    if (part == 2) {
        *reinterpret_cast<uint4*>(vc + cidx) = *reinterpret_cast<const uint4*>(src);
        return;
    }
    float x[V];
    if (e0 >= rot) {  // pass-through dims
        if (part == 1) *reinterpret_cast<uint4*>(kc + cidx) = *reinterpret_cast<const uint4*>(src);
        return;
    }
    const float* cs = cos_tab + (int64_t)rpos * half;
    const float* sn = sin_tab + (int64_t)rpos * half;
    if (half < V) {
        // the whole rotary span [0, rot) lies inside this chunk (rot <= V)
        load_vec<T>(src, x);
        float y[V];
#pragma unroll
        for (int i = 0; i < V; ++i) y[i] = x[i];
        for (int j = 0; j < half; ++j) {
            const float c_ = cs[j], s_ = sn[j];
            y[j] = x[j] * c_ - x[j + half] * s_;
            y[j + half] = x[j + half] * c_ + x[j] * s_;
        }
        if (part == 0) store_vec<T>(src, y);
        else store_vec<T>(kc + cidx, y);
        return;
    }
    // half >= V: chunk e0 in [0, half) owns the pair (e0, e0 + half); chunks in [half, rot) are written by their owner
    if (e0 >= half) return;
    float x2[V], y1[V], y2[V];
    load_vec<T>(src, x);
    load_vec<T>(src + half, x2);
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const float c_ = cs[e0 + i], s_ = sn[e0 + i];
        y1[i] = x[i] * c_ - x2[i] * s_;
        y2[i] = x2[i] * c_ + x[i] * s_;
    }
    if (part == 0) {
        store_vec<T>(src, y1);
        store_vec<T>(src + half, y2);
    } else {
        store_vec<T>(kc + cidx, y1);
        store_vec<T>(kc + cidx + half, y2);
    }
}

// ---- row-wise streaming attention ------------------------------------------------------------------------------------------
template <int NW>
__device__ inline float block_max(float v, float* red, int wave, int lane) {
    v = wave_max(v);
    if (NW == 1) return v;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) r = fmaxf(r, red[w]);
    return r;
}
template <int NW>
__device__ inline float block_sum(float v, float* red, int wave, int lane) {
    v = wave_sum(v);
    if (NW == 1) return v;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) r += red[w];
    return r;
}

// PERSIST: a fixed grid of workgroups walks the (sequence, head, query row) items (total_items of them) instead of one
// workgroup per item: the kernel then holds a bounded number of wave slots / registers per CU for its whole run time, so a
// GEMM launched on another stream can stay co-resident with it (half-batch pipeline: this kernel is HBM-bound and needs
// bytes in flight, not occupancy).
template <typename T, int D, int NW, bool NT = false, bool PERSIST = false>
__global__ __launch_bounds__(NW * 64) void attend_rows_kernel(const T* __restrict__ qkv, int64_t ld_qkv,
                                                             const T* __restrict__ kc, const T* __restrict__ vc,
                                                             int64_t n_max, const int32_t* __restrict__ pos0_dev,
                                                             int pos0_host, const uint8_t* __restrict__ mask,
                                                             int64_t mask_ld, T* __restrict__ out, int64_t ldo,
                                                             float* __restrict__ ws, int q_len, int heads, int n_split,
                                                             float scale, int64_t total_items = 0,
                                                             const int32_t* __restrict__ seq_skip = nullptr) {
    constexpr int V = Elem<T>::VEC;
    constexpr int LPR = D / V;     // lanes per K/V row
    constexpr int RPW = 64 / LPR;  // rows per wave-instruction
    constexpr int U = NW >= 16 ? 8 : 4;   // independent 16-byte loads in flight per lane (the 16-wave form covers 512 keys per sweep)
    constexpr int RED = NW > 4 ? NW : 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* red = smem;               // [NW]
    float* ored = smem + RED;        // [NW][D]  cross-wave output reduce
    float* sc = smem + RED + NW * D; // [cap] scores / probabilities of this split

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane % LPR, sub = lane / LPR;
    for (int64_t idx = blockIdx.x; PERSIST ? idx < total_items : idx == (int64_t)blockIdx.x; idx += gridDim.x) {
    if (PERSIST && idx != (int64_t)blockIdx.x) __syncthreads();   // the previous item's LDS rows are free again
    const int t = (int)(idx % q_len);
    const int head = (int)((idx / q_len) % heads);
    const int64_t b = idx / ((int64_t)q_len * heads);
    const int split = blockIdx.y;
    if (seq_skip && seq_skip[b] != 0) {
        // a finished sequence (BlockTransformer.generate: its next tokens are forced to pad whatever the context is; the
        // reference compacts such rows away, :381-398): do not stream its KV cache, leave zeros
        if (n_split == 1) {
            if (wave == 0 && sub == 0) {
                float z[V];
#pragma unroll
                for (int i = 0; i < V; ++i) z[i] = 0.f;
                store_vec<T>(out + (b * q_len + t) * ldo + (int64_t)head * D + c * V, z);
            }
        } else if (wave == 0 && sub == 0) {
            float* w = ws + (idx * n_split + split) * (D + 2);
            if (c == 0) {
                w[0] = -INFINITY;
                w[1] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < V; ++i) w[2 + c * V + i] = 0.f;
        }
        continue;
    }
    const int pos0 = pos0_dev ? *pos0_dev : pos0_host;
    const int nk = pos0 + t + 1;
    const int per = (nk + n_split - 1) / n_split;
    const int k0 = split * per;
    const int k1 = min(nk, k0 + per);
    const int cnt = max(k1 - k0, 0);

    float q[V];
    load_vec<T>(qkv + (b * q_len + t) * ld_qkv + (int64_t)head * 3 * D + c * V, q);
#pragma unroll
    for (int i = 0; i < V; ++i) q[i] *= scale;
    const T* kbase = kc + (b * heads + head) * n_max * D + c * V;
    const T* vbase = vc + (b * heads + head) * n_max * D + c * V;
    const uint8_t* mrow = mask ? mask + b * mask_ld : nullptr;

    // local-window variant (NW == 1, a handful of keys): the first batch of V rows does not depend on the scores -- issue
    // it together with the K rows so the whole kernel is one memory round trip instead of two
    float vpre[U][V];
    if (NW == 1) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int r = k0 + u * RPW + sub;
            r = r < k1 ? r : k1 - 1;
            r = r < 0 ? 0 : r;
            load_vec<T>(vbase + (int64_t)r * D, vpre[u]);
        }
    }
    // pass 1: scores
    for (int rb = k0 + wave * RPW; rb < k1; rb += NW * RPW * U) {
        float kv[U][V];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int r = rb + u * NW * RPW + sub;
            r = r < k1 ? r : k1 - 1;
            if (NT) load_vec_nt<T>(kbase + (int64_t)r * D, kv[u]);
            else load_vec<T>(kbase + (int64_t)r * D, kv[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < V; ++i) d = fmaf(q[i], kv[u][i], d);
            d = wave_sum<LPR>(d);
            const int r = rb + u * NW * RPW + sub;
            if (c == 0 && r < k1) sc[r - k0] = (mrow && mrow[r] == 0) ? -INFINITY : d;
        }
    }
    __syncthreads();
    float m = -INFINITY;
    for (int i = tid; i < cnt; i += NW * 64) m = fmaxf(m, sc[i]);
    m = block_max<NW>(m, red, wave, lane);
    float l = 0.f;
    for (int i = tid; i < cnt; i += NW * 64) {
        const float p = (m == -INFINITY) ? 0.f : __expf(sc[i] - m);
        sc[i] = p;
        l += p;
    }
    l = block_sum<NW>(l, red, wave, lane);
    __syncthreads();

    // pass 2: o = sum_r p[r] * V[r]
    float acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = 0.f;
    for (int rb = k0 + wave * RPW; rb < k1; rb += NW * RPW * U) {
        float vv[U][V];
        float p[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int r = rb + u * NW * RPW + sub;
            const bool ok = r < k1;
            r = ok ? r : k1 - 1;
            if (NW == 1 && rb == k0) {
#pragma unroll
                for (int i = 0; i < V; ++i) vv[u][i] = vpre[u][i];
            } else if (NT) load_vec_nt<T>(vbase + (int64_t)r * D, vv[u]);
            else load_vec<T>(vbase + (int64_t)r * D, vv[u]);
            p[u] = ok ? sc[r - k0] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < V; ++i) acc[i] = fmaf(p[u], vv[u][i], acc[i]);
    }
    // reduce over the RPW row groups of the wave (lane bits >= log2(LPR)), then across waves
#pragma unroll
    for (int i = 0; i < V; ++i) {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) acc[i] += __shfl_xor(acc[i], o, 64);
    }
    if (NW > 1) {
        if (sub == 0) {
#pragma unroll
            for (int i = 0; i < V; ++i) ored[wave * D + c * V + i] = acc[i];
        }
        __syncthreads();
        if (wave == 0 && sub == 0) {
#pragma unroll
            for (int i = 0; i < V; ++i) {
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) s += ored[w * D + c * V + i];
                acc[i] = s;
            }
        }
    }
    if (wave == 0 && sub == 0) {
        if (n_split == 1) {
            const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
            for (int i = 0; i < V; ++i) acc[i] *= inv;
            store_vec<T>(out + (b * q_len + t) * ldo + (int64_t)head * D + c * V, acc);
        } else {
            float* w = ws + (idx * n_split + split) * (D + 2);
            if (c == 0) {
                w[0] = m;
                w[1] = l;
            }
#pragma unroll
            for (int i = 0; i < V; ++i) w[2 + c * V + i] = acc[i];
        }
    }
    }   // item loop
}

// ---- local-window attention (token decoder: <= E + L - 1 keys, <= 2 query rows per call) ------------------------------------
// The north-star "local path": one WAVE per (sequence, head) covers ALL query rows of the call; the whole K/V window (<= 12 rows
// of 256 B at d = 128) is fetched once, 16 lanes x 16 B per row, 4 rows per wave-instruction, every load of the item issued
// before the first use (one memory round trip); the softmax row lives in registers and is reduced with wave shuffles only
// (no LDS, no barrier): scores by a 16-lane butterfly per key, max / sum across the four 16-lane row groups by xor-16 / xor-32.
// With <= 12 keys an LDS-resident K/V tile would be written once and read once -- registers ARE the resident tile.
// HPW = (sequence, head) items per wave, processed back to back with all their loads in flight together: the kernel is
// latency-bound (tens of thousands of tiny items), not bandwidth-bound.
template <int D, int HPW, int U>
__global__ __launch_bounds__(256) void local_attend_kernel(const bf16* __restrict__ qkv, int64_t ld_qkv, const bf16* __restrict__ kc,
                                                          const bf16* __restrict__ vc, int64_t n_max, int pos0,
                                                          const uint8_t* __restrict__ mask, int64_t mask_ld,
                                                          bf16* __restrict__ out, int64_t ldo, int q_len, int heads,
                                                          int64_t items, float scale) {
    static_assert(D == 128, "one 16-lane group per 256-byte row");
    // U = row batches of 4 keys this instantiation fetches: 2 (window <= 8 rows: L = 4) or 3 (<= 12 rows: L = 8)
    constexpr int QMAX = 2;                    // query rows per call (E = 2 prefix rows at step 0, 1 afterwards)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, sub = lane >> 4;
    const int64_t first = ((int64_t)blockIdx.x * 4 + wave) * HPW;
    const int nk_all = pos0 + q_len;           // keys visible to the last query row
    uint4 kq[HPW][U], vq[HPW][U], qq[HPW][QMAX];
    // ---- every load of every item first ----
#pragma unroll
    for (int h = 0; h < HPW; ++h) {
        int64_t it = first + h;
        it = it < items ? it : items - 1;
        const int64_t b = it / heads;
        const int head = (int)(it - b * heads);
        const bf16* kb = kc + (b * heads + head) * n_max * D + c * 8;
        const bf16* vb = vc + (b * heads + head) * n_max * D + c * 8;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int r = u * 4 + sub;
            r = r < nk_all ? r : nk_all - 1;
            kq[h][u] = *reinterpret_cast<const uint4*>(kb + (int64_t)r * D);
            vq[h][u] = *reinterpret_cast<const uint4*>(vb + (int64_t)r * D);
        }
#pragma unroll
        for (int t = 0; t < QMAX; ++t) {
            const int tt = t < q_len ? t : q_len - 1;
            qq[h][t] = *reinterpret_cast<const uint4*>(qkv + (b * q_len + tt) * ld_qkv + (int64_t)head * 3 * D + c * 8);
        }
    }
#pragma unroll
    for (int h = 0; h < HPW; ++h) {
        const int64_t it = first + h;
        if (it >= items) break;
        const int64_t b = it / heads;
        const int head = (int)(it - b * heads);
        const uint8_t* mrow = mask ? mask + b * mask_ld : nullptr;
#pragma unroll
        for (int t = 0; t < QMAX; ++t) {
            if (t >= q_len) break;
            const int nk = pos0 + t + 1;
            float q[8];
            {
                const uint32_t w[4] = {qq[h][t].x, qq[h][t].y, qq[h][t].z, qq[h][t].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    q[2 * i] = __uint_as_float(w[i] << 16) * scale;
                    q[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u) * scale;
                }
            }
            float sc[U];
            float mx = -INFINITY;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t w[4] = {kq[h][u].x, kq[h][u].y, kq[h][u].z, kq[h][u].w};
                float d = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    d = fmaf(q[2 * i], __uint_as_float(w[i] << 16), d);
                    d = fmaf(q[2 * i + 1], __uint_as_float(w[i] & 0xffff0000u), d);
                }
                d = row16_sum_dpp(d);                                  // all 16 lanes of the row group hold the score (== wave_sum<16>, bit for bit)
                const int r = u * 4 + sub;
                const bool vis = r < nk && (mrow == nullptr || mrow[r] != 0);
                sc[u] = vis ? d : -INFINITY;
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
                const uint32_t w[4] = {vq[h][u].x, vq[h][u].y, vq[h][u].z, vq[h][u].w};
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
                store_vec<bf16>(out + (b * q_len + t) * ldo + (int64_t)head * D + c * 8, acc);
            }
        }
    }
}

template <typename T>
__global__ void attn_combine_kernel(const float* __restrict__ ws, T* __restrict__ out, int64_t ldo, int q_len, int heads,
                                    int D, int n_split) {
    const int64_t idx = blockIdx.x;
    const int t = (int)(idx % q_len);
    const int head = (int)((idx / q_len) % heads);
    const int64_t b = idx / ((int64_t)q_len * heads);
    const float* w = ws + idx * n_split * (D + 2);
    float m = -INFINITY;
    for (int s = 0; s < n_split; ++s) m = fmaxf(m, w[s * (D + 2)]);
    for (int e = threadIdx.x; e < D; e += blockDim.x) {
        float l = 0.f, o = 0.f;
        for (int s = 0; s < n_split; ++s) {
            const float ms = w[s * (D + 2)];
            const float f = (ms == -INFINITY) ? 0.f : __expf(ms - m);
            l += w[s * (D + 2) + 1] * f;
            o += w[s * (D + 2) + 2 + e] * f;
        }
        Elem<T>::store(out + (b * q_len + t) * ldo + (int64_t)head * D + e, l > 0.f ? o / l : 0.f);
    }
}

// ---- bf16 MFMA flash prefill ---------------------------------------------------------------------------------------------
// block = 4 waves x (QF = 2 fragments of 16 query rows) = 128 query rows; key tiles of 64.
constexpr int FQ_ROWS = 128, FK_TILE = 64, VT_PITCH = 72;  // Vt row pitch in bf16 elements (64 keys + 8 pad)

#ifdef BT_DEBUG_HOOKS   // v1 flash prefill (register/LDS-transposed V): A/B baseline only, not in the product library
template <int D>
__global__ __launch_bounds__(256) void flash_prefill_kernel(const bf16* __restrict__ qkv, int64_t ld_qkv,
                                                           const bf16* __restrict__ kc, const bf16* __restrict__ vc,
                                                           int64_t n_max, const int32_t* __restrict__ pos0_dev,
                                                           int pos0_host, const uint8_t* __restrict__ mask,
                                                           int64_t mask_ld, bf16* __restrict__ out, int64_t ldo,
                                                           int q_len, int heads, float scale) {
    constexpr int KK = D / 32;  // k-steps of the QK^T product
    constexpr int DF = D / 16;  // output d-fragments
    constexpr int QF = 2;
    __shared__ __attribute__((aligned(16))) uint16_t vt[D * VT_PITCH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int qb = blockIdx.x, head = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int pos0 = pos0_dev ? *pos0_dev : pos0_host;
    const int qbase = qb * FQ_ROWS + wave * (QF * 16);

    // Q fragments (B operand of S^T = K Q^T): lane (j = q row = l15, k group g) holds Q[row][kk*32 + g*8 .. +8)
    bf16x8_t qf[QF][KK];
    int qrow[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        qrow[f] = qbase + f * 16 + l15;
        const int r = qrow[f] < q_len ? qrow[f] : q_len - 1;
        const bf16* qp = qkv + (b * q_len + r) * ld_qkv + (int64_t)head * 3 * D + g * 8;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) qf[f][kk] = *reinterpret_cast<const bf16x8_t*>(qp + kk * 32);
    }
    f32x4_t oacc[QF][DF];
#pragma unroll
    for (int f = 0; f < QF; ++f)
#pragma unroll
        for (int d = 0; d < DF; ++d) oacc[f][d] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float mrun[QF], lrun[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        mrun[f] = -INFINITY;
        lrun[f] = 0.f;
    }

    const bf16* kbase = kc + (b * heads + head) * n_max * D;
    const bf16* vbase = vc + (b * heads + head) * n_max * D;
    const uint8_t* mrow = mask ? mask + b * mask_ld : nullptr;
    const int last_q = min(q_len, (qb + 1) * FQ_ROWS) - 1;
    const int n_keys = pos0 + last_q + 1;
    const int n_tiles = (n_keys + FK_TILE - 1) / FK_TILE;

    for (int kt = 0; kt < n_tiles; ++kt) {
        const int key0 = kt * FK_TILE;
        __syncthreads();  // previous tile's Vt reads are done
        // stage V^T: chunk = (key, 8 dims)
#pragma unroll
        for (int i = 0; i < (FK_TILE * D / 8) / 256; ++i) {
            const int ch = tid + i * 256;
            const int key = ch / (D / 8), dc = ch % (D / 8);
            const int kr = key0 + key;
            // rows >= n_keys were never written (uninitialised cache memory): P is 0 there, but 0 * NaN = NaN
            uint4 v = make_uint4(0, 0, 0, 0);
            if (kr < n_keys) v = *reinterpret_cast<const uint4*>(vbase + (int64_t)kr * D + dc * 8);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                vt[(dc * 8 + 2 * e) * VT_PITCH + key] = (uint16_t)(w[e] & 0xffffu);
                vt[(dc * 8 + 2 * e + 1) * VT_PITCH + key] = (uint16_t)(w[e] >> 16);
            }
        }
        __syncthreads();

        // S^T fragments: sacc[f][kf][r] = score(key = key0 + kf*16 + g*4 + r, q = qrow[f])
        f32x4_t sacc[QF][4];
#pragma unroll
        for (int f = 0; f < QF; ++f)
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) sacc[f][kf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) {
            int kr = key0 + kf * 16 + l15;
            kr = kr < n_max ? kr : (int)n_max - 1;
            const bf16* kp = kbase + (int64_t)kr * D + g * 8;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const bf16x8_t kfr = *reinterpret_cast<const bf16x8_t*>(kp + kk * 32);
#pragma unroll
                for (int f = 0; f < QF; ++f)
                    sacc[f][kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[f][kk], sacc[f][kf], 0, 0, 0);
            }
        }
        // key visibility for this lane's 16 keys (same for both q fragments except the causal bound)
        bool kvis[4][4];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = key0 + kf * 16 + g * 4 + r;
                kvis[kf][r] = (mrow == nullptr) || (key < n_keys && mrow[key] != 0);
            }
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            const int qpos = pos0 + qrow[f];
            float tmax = -INFINITY;
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = key0 + kf * 16 + g * 4 + r;
                    const float s = (key <= qpos && kvis[kf][r]) ? sacc[f][kf][r] * scale : -INFINITY;
                    sacc[f][kf][r] = s;
                    tmax = fmaxf(tmax, s);
                }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float mnew = fmaxf(mrun[f], tmax);
            const float alpha = (mnew == -INFINITY) ? 1.0f : __expf(mrun[f] - mnew);
            float psum = 0.f;
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = (mnew == -INFINITY) ? 0.f : __expf(sacc[f][kf][r] - mnew);
                    sacc[f][kf][r] = p;
                    psum += p;
                }
            lrun[f] = lrun[f] * alpha + psum;  // per-lane partial (this lane's keys); reduced over g at the end
            mrun[f] = mnew;
            // rescale O rows: row (g*4 + r) of fragment f belongs to the lane column l15' = g*4 + r
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = __shfl(alpha, g * 4 + r, 64);
#pragma unroll
                for (int d = 0; d < DF; ++d) oacc[f][d][r] *= a;
            }
        }
        // O += P V : two 32-key steps; key slot (g, e): e < 4 -> key (2ks)*16 + g*4 + e ; e >= 4 -> (2ks+1)*16 + g*4 + e-4
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t pf[QF];
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                union {
                    uint32_t u[4];
                    bf16x8_t v;
                } pk;
                pk.u[0] = pack_bf16x2(sacc[f][2 * ks][0], sacc[f][2 * ks][1]);
                pk.u[1] = pack_bf16x2(sacc[f][2 * ks][2], sacc[f][2 * ks][3]);
                pk.u[2] = pack_bf16x2(sacc[f][2 * ks + 1][0], sacc[f][2 * ks + 1][1]);
                pk.u[3] = pack_bf16x2(sacc[f][2 * ks + 1][2], sacc[f][2 * ks + 1][3]);
                pf[f] = pk.v;
            }
#pragma unroll
            for (int d = 0; d < DF; ++d) {
                union {
                    uint2 h[2];
                    bf16x8_t v;
                } vf;
                const uint16_t* vrow = vt + (d * 16 + l15) * VT_PITCH + g * 4;
                vf.h[0] = *reinterpret_cast<const uint2*>(vrow + (2 * ks) * 16);
                vf.h[1] = *reinterpret_cast<const uint2*>(vrow + (2 * ks + 1) * 16);
#pragma unroll
                for (int f = 0; f < QF; ++f)
                    oacc[f][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[f], vf.v, oacc[f][d], 0, 0, 0);
            }
        }
    }
    // epilogue: normalise and store.  oacc[f][d][r] = O[q = f*16 + g*4 + r][dim = d*16 + l15]
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        float l = lrun[f];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = __shfl(inv, g * 4 + r, 64);
            const int row = qbase + f * 16 + g * 4 + r;
            if (row < q_len) {
                bf16* op = out + (b * q_len + row) * ldo + (int64_t)head * D + l15;
#pragma unroll
                for (int d = 0; d < DF; ++d) op[d * 16].x = f32_to_bf16(oacc[f][d][r] * a);
            }
        }
    }
}
#endif  // BT_DEBUG_HOOKS

// ---- bf16 MFMA flash prefill, v2 ------------------------------------------------------------------------------------------
// Same math and fragment mapping as flash_prefill_kernel, different data movement:
//   * K and V tiles (64 keys x D) are streamed HBM/L2 -> LDS with 16-byte global_load_lds (no VGPR round trip), double
//     buffered: tile t+1 is in flight while tile t is computed; one barrier per tile;
//   * K fragments (MFMA A operand of S^T = K Q^T) are ds_read_b128 from rows XOR-swizzled over the row's 16-byte slots;
//   * V is kept ROW-MAJOR in LDS and consumed through ds_read_b64_tr_b16 (hardware 4x16 transpose read: lane c of a
//     16-lane group receives V[k0..k0+3][c] — semantics probed on gfx950, profiles/r1_ds_read_tr_b16_semantics_probe.txt),
//     32-byte segments XOR-swizzled with (row & 7) so the four rows a group touches land on distinct banks;
//   * waves whose 32 query rows all precede a key tile skip its MFMAs (causal), but still take part in staging.
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

template <int D, int NW = 4>
__device__ inline void flash_stage_tile(const bf16* __restrict__ kbase, const bf16* __restrict__ vbase, int key0,
                                        int n_keys, char* kbuf, char* vbuf, int wave, int lane) {
    constexpr int ROWB = D * 2;            // bytes per row
    constexpr int SLOTS = ROWB / 16;       // 16-byte slots per row (16 / 8 / 4)
    constexpr int RPI = 1024 / ROWB;       // rows per wave-instruction (4 / 8 / 16)
    constexpr int NI = 64 / RPI / NW;      // instructions per wave per operand
    static_assert(NI >= 1, "more waves than 1 KiB pieces in a K tile");
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int rbase = (wave * NI + i) * RPI;
        const int row = rbase + lane / SLOTS;
        const int p = lane % SLOTS;
        int kr = key0 + row;
        // rows >= n_keys were never written (uninitialised cache memory; 0 * NaN = NaN in P.V): re-read the last visible
        // row instead — its score is masked to -inf, so the duplicate gets weight exactly 0
        kr = kr < n_keys ? kr : n_keys - 1;
        const int ks = p ^ (row & (SLOTS - 1));                       // K: slot swizzle
        const int vs = p ^ (((row & 7) << 1) & (SLOTS - 1));          // V: 32-byte segment swizzle
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kbase + (int64_t)kr * D + ks * 8),
                                         (__attribute__((address_space(3))) void*)(kbuf + rbase * ROWB), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vbase + (int64_t)kr * D + vs * 8),
                                         (__attribute__((address_space(3))) void*)(vbuf + rbase * ROWB), 16, 0, 0);
    }
}

// NW waves x 32 query rows per workgroup, ST-deep ring of 64-key K/V tiles (LDS-DMA, ST-1 tiles in flight).
//   <D, 4, 2>  128 query rows, double buffer, two workgroups per CU: production.
//   <D, 8, 4>  256 query rows (K/V tiles loaded once per 256 rows), three tiles in flight, one workgroup per CU: measured
//              the SAME speed (287-307 TF at n = 512), i.e. the tile loop is not waiting for its loads -- it is
//              instruction-issue / dependency bound (~450 non-MFMA instructions and ~30 waits per 64 MFMAs, two waves
//              per SIMD at 240 VGPRs).  Kept as test variant 2.
//   OPT bits (A/B in the debug library): 1 = the first ST-1 K/V tiles are put in flight BEFORE the Q fragment loads (their
//   latencies overlap instead of adding up: a 512-key prompt has only 2..8 tiles per workgroup, so the prologue is a large
//   share of its time); 2 = query blocks are dispatched heaviest-first (causal: block qb has 2 qb + 2 tiles); 4 = s_setprio
//   around the MFMA clusters (the two workgroups of a CU are at independent phases).
constexpr int FLASH_OPT = 95;
//   QF_ = 16-row query fragments per wave: <D, 8, 2, OPT, 1> = the same 128-row block on 8 waves of 16 rows -- half the
//   accumulator registers per wave, so four waves per SIMD instead of two hide the tile loop's LDS / MFMA latencies.
template <int D, int NW, int ST, int OPT = FLASH_OPT, int QF_ = 2>
__global__ __launch_bounds__(NW * 64, NW * QF_ == 8 ? (QF_ == 1 ? 4 : 2) : 1) void flash_prefill_v2_kernel(const bf16* __restrict__ qkv, int64_t ld_qkv,
                                                              const bf16* __restrict__ kc, const bf16* __restrict__ vc,
                                                              int64_t n_max, const int32_t* __restrict__ pos0_dev,
                                                              int pos0_host, const uint8_t* __restrict__ mask,
                                                              int64_t mask_ld, bf16* __restrict__ out, int64_t ldo,
                                                              int q_len, int heads, float scale) {
    constexpr int KK = D / 32, DF = D / 16, QF = QF_;
    // OPT bit 8: accumulate O TRANSPOSED (O^T = V^T P^T: the same two operand registers, swapped).  D[i = dim][j = query]
    // puts query row l15 in the lane that also holds that row's softmax statistics, so the per-tile rescale needs no
    // cross-lane broadcast (8 of the loop's 12 ds_bpermute round trips) and the epilogue packs four consecutive dims
    constexpr bool OT = (OPT & 8) != 0;
    constexpr bool FS = (OPT & 16) != 0;   // bit 16: softmax scale folded into the exponent FMA (no separate multiply per score)
    constexpr bool LZ = (OPT & 32) != 0;   // bit 32: skip the O rescale on tiles where no running max of the wave moved
    constexpr int ROWB = D * 2, SLOTS = ROWB / 16;
    constexpr int TILE_B = FK_TILE * ROWB;  // bytes of one K (or V) tile
    constexpr int FQ = NW * QF * 16;                                     // query rows per workgroup
    constexpr int NI = 64 / (1024 / ROWB) / NW;                          // DMA instructions per wave, operand and tile
    __shared__ __attribute__((aligned(16))) char smem[ST * 2 * TILE_B];  // [stage][K | V]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    int qb, head;
    int64_t b;
    if (OPT & 64) {
        // 1-D grid, XCD-aware: workgroup w runs on XCD w % 8 (round-robin dispatch) and each XCD has its own L2, so the
        // query blocks of one (sequence, head) -- which all stream the SAME K/V rows -- get ids 8 apart: same XCD, dispatched
        // together, the K/V tiles come from HBM once instead of once per query block (a 512-key prompt: 640 -> 256 KB)
        const int nq = (q_len + FQ - 1) / FQ;
        const int w = blockIdx.x, ng = gridDim.x / nq, full = ng / 8;
        int grp, qi;
        if (w < full * 8 * nq) {
            const int c = w / (8 * nq), r = w - c * (8 * nq);
            grp = c * 8 + (r & 7);
            qi = r >> 3;
        } else {   // the last ng % 8 groups
            const int rem = ng - full * 8, r = w - full * 8 * nq;
            grp = full * 8 + r % rem;
            qi = r / rem;
        }
        qb = (OPT & 2) ? nq - 1 - qi : qi;
        head = grp % heads;
        b = grp / heads;
    } else {
        qb = (OPT & 2) ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
        head = blockIdx.y;
        b = blockIdx.z;
    }
    const int pos0 = pos0_dev ? *pos0_dev : pos0_host;
    const int qbase = qb * FQ + wave * (QF * 16);

    const bf16* kbase = kc + (b * heads + head) * n_max * D;
    const bf16* vbase = vc + (b * heads + head) * n_max * D;
    const uint8_t* mrow = mask ? mask + b * mask_ld : nullptr;
    const int last_q = min(q_len, (qb + 1) * FQ) - 1;
    const int n_keys = pos0 + last_q + 1;
    const int n_tiles = (n_keys + FK_TILE - 1) / FK_TILE;
    const int wave_last_pos = pos0 + min(qbase + QF * 16, q_len) - 1;  // last query position this wave owns
    const float sl2 = scale * 1.44269504088896340736f;                 // softmax in base 2

    if (OPT & 1) {   // tiles 0 .. ST-2 into the ring first; the Q loads below are younger (the first in-loop wait covers both)
#pragma unroll
        for (int x = 0; x < ST - 1; ++x)
            if (x < n_tiles)
                flash_stage_tile<D, NW>(kbase, vbase, x * FK_TILE, n_keys, smem + x * 2 * TILE_B, smem + x * 2 * TILE_B + TILE_B,
                                        wave, lane);
    }
    bf16x8_t qf[QF][KK];
    int qrow[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        qrow[f] = qbase + f * 16 + l15;
        const int r = qrow[f] < q_len ? qrow[f] : q_len - 1;
        const bf16* qp = qkv + (b * q_len + r) * ld_qkv + (int64_t)head * 3 * D + g * 8;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) qf[f][kk] = *reinterpret_cast<const bf16x8_t*>(qp + kk * 32);
    }
    f32x4_t oacc[QF][DF];
#pragma unroll
    for (int f = 0; f < QF; ++f)
#pragma unroll
        for (int d = 0; d < DF; ++d) oacc[f][d] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float mrun[QF], lrun[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        mrun[f] = -INFINITY;
        lrun[f] = 0.f;
    }

    // lane-constant LDS offsets of the fragment reads (the XOR swizzles make them non-affine in kk / d, so they are
    // tabulated once instead of being recomputed for every tile: ~350 VALU instructions per tile and wave)
    int kaddr[KK], vaddr[DF];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) kaddr[kk] = l15 * ROWB + (((kk * 4 + g) ^ (l15 & (SLOTS - 1))) * 16);
    {
        const int rl = g * 4 + (l15 >> 2);                            // row inside a 16-row group of the transpose read
        const int swz = (((rl & 7) << 1) & (SLOTS - 1)) << 4;
#pragma unroll
        for (int d = 0; d < DF; ++d) vaddr[d] = rl * ROWB + (((d * 16 + (l15 & 3) * 4) * 2) ^ swz);
    }
    if (!(OPT & 1)) {
        // the Q fragments are loaded first (older than every DMA), then tiles 0 .. ST-2 go into the ring
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int x = 0; x < ST - 1; ++x)
            if (x < n_tiles)
                flash_stage_tile<D, NW>(kbase, vbase, x * FK_TILE, n_keys, smem + x * 2 * TILE_B, smem + x * 2 * TILE_B + TILE_B,
                                        wave, lane);
    } else if (ST > 2) {
        // deeper rings leave younger tiles in flight at the first in-loop wait: the Q loads (youngest) must have landed by then
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    for (int kt = 0; kt < n_tiles; ++kt) {
        const int key0 = kt * FK_TILE;
        const char* kb = smem + (kt % ST) * 2 * TILE_B;
        const char* vb = kb + TILE_B;
        {   // this wave's share of tile kt has landed; the (up to ST-2) younger tiles stay in flight: 2*NI DMAs each
            const int younger = min(ST - 2, n_tiles - 1 - kt);
            if (younger >= 2) {
                if (NI == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else if (NI == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else if (younger == 1) {
                if (NI == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (NI == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __syncthreads();   // tile kt visible to everybody; everybody is done with tile kt-1 -> its stage can be refilled
        if (kt + ST - 1 < n_tiles) {
            char* nb = smem + ((kt + ST - 1) % ST) * 2 * TILE_B;
            flash_stage_tile<D, NW>(kbase, vbase, key0 + (ST - 1) * FK_TILE, n_keys, nb, nb + TILE_B, wave, lane);
        }
        if (key0 <= wave_last_pos) {  // wave-uniform: some query of this wave can see this tile
            f32x4_t sacc[QF][4];
#pragma unroll
            for (int f = 0; f < QF; ++f)
#pragma unroll
                for (int kf = 0; kf < 4; ++kf) sacc[f][kf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (OPT & 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) {
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    const bf16x8_t kfr = *reinterpret_cast<const bf16x8_t*>(kb + kaddr[kk] + kf * 16 * ROWB);
#pragma unroll
                    for (int f = 0; f < QF; ++f)
                        sacc[f][kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[f][kk], sacc[f][kf], 0, 0, 0);
                }
            }
            if (OPT & 4) __builtin_amdgcn_s_setprio(0);
            // scores -> probabilities (online softmax, base 2: `sl2` = scale * log2(e) is folded into one multiply).
            // Interior tiles -- every key of the tile visible to every query row of this wave, no padding mask, no key past
            // n_keys -- skip the visibility compares and selects (most tiles of a causal prefill).
            const bool need_mask = (mrow != nullptr) || (key0 + FK_TILE > n_keys) || (key0 + FK_TILE - 1 > pos0 + qbase);
            if (need_mask) {
                bool kvis[4][4];
#pragma unroll
                for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = key0 + kf * 16 + g * 4 + r;
                        kvis[kf][r] = (key < n_keys) && (mrow == nullptr || mrow[key] != 0);
                    }
#pragma unroll
                for (int f = 0; f < QF; ++f) {
                    const int qpos = pos0 + qrow[f];
#pragma unroll
                    for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int key = key0 + kf * 16 + g * 4 + r;
                            sacc[f][kf][r] = (key <= qpos && kvis[kf][r]) ? (FS ? sacc[f][kf][r] : sacc[f][kf][r] * sl2) : -INFINITY;
                        }
                }
            } else if (!FS) {
#pragma unroll
                for (int f = 0; f < QF; ++f)
#pragma unroll
                    for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sacc[f][kf][r] *= sl2;
            }
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                float tmax = -INFINITY;
#pragma unroll
                for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, sacc[f][kf][r]);
                tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float mnew = fmaxf(mrun[f], tmax);
                const bool dead = (mnew == -INFINITY);          // no visible key so far for this query row
                // FS: the running max stays in RAW score units and the softmax scale rides in the exponent's FMA
                // (exp2(s * sl2 - m * sl2)): one multiply per score less, and one rounding less
                const float msub = dead ? 0.f : (FS ? -mnew * sl2 : mnew);
                const float alpha = dead ? 1.0f : __builtin_amdgcn_exp2f(FS ? (mrun[f] - mnew) * sl2 : mrun[f] - mnew);
                float psum = 0.f;
#pragma unroll
                for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // exp2(-inf) = 0 for masked keys
                        const float pv = __builtin_amdgcn_exp2f(FS ? fmaf(sacc[f][kf][r], sl2, msub) : sacc[f][kf][r] - msub);
                        sacc[f][kf][r] = pv;
                        psum += pv;
                    }
                lrun[f] = lrun[f] * alpha + psum;
                mrun[f] = mnew;
                if (OT) {   // O^T accumulators: the lane's column IS query row l15 -> its own alpha, no broadcast
                    // (and when no row of the wave raised its running max -- alpha == 1 everywhere -- nothing to do)
                    if (!LZ || __builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                        for (int d = 0; d < DF; ++d)
#pragma unroll
                            for (int r = 0; r < 4; ++r) oacc[f][d][r] *= alpha;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float a = __shfl(alpha, g * 4 + r, 64);
#pragma unroll
                        for (int d = 0; d < DF; ++d) oacc[f][d][r] *= a;
                    }
                }
            }
            // O += P V.  key slot (g, e): e < 4 -> tile key (2ks)*16 + g*4 + e ; e >= 4 -> (2ks+1)*16 + g*4 + (e-4)
            if (OPT & 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t pf[QF];
#pragma unroll
                for (int f = 0; f < QF; ++f) {
                    union {
                        uint32_t u[4];
                        bf16x8_t v;
                    } pk;
                    pk.u[0] = pack_bf16x2(sacc[f][2 * ks][0], sacc[f][2 * ks][1]);
                    pk.u[1] = pack_bf16x2(sacc[f][2 * ks][2], sacc[f][2 * ks][3]);
                    pk.u[2] = pack_bf16x2(sacc[f][2 * ks + 1][0], sacc[f][2 * ks + 1][1]);
                    pk.u[3] = pack_bf16x2(sacc[f][2 * ks + 1][2], sacc[f][2 * ks + 1][3]);
                    pf[f] = pk.v;
                }
                // transpose-read addresses: lane i of a 16-lane group supplies row kb + (i >> 2), columns (i & 3) * 4
                // (rows (2 ks) * 16 + rl and + 16; the row's swizzle depends on rl & 7 only)
#pragma unroll
                for (int d = 0; d < DF; ++d) {
                    const int a1 = vaddr[d] + (2 * ks) * 16 * ROWB;
                    const int a2 = a1 + 16 * ROWB;
                    union {
                        s16x4_t h[2];
                        bf16x8_t v;
                    } vf;
                    vf.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vb + a1));
                    vf.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vb + a2));
#pragma unroll
                    for (int f = 0; f < QF; ++f)
                        oacc[f][d] = OT ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf[f], oacc[f][d], 0, 0, 0)
                                        : __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[f], vf.v, oacc[f][d], 0, 0, 0);
                }
            }
            if (OPT & 4) __builtin_amdgcn_s_setprio(0);
        }
    }
    // epilogue: normalise, stage the FQ x D output tile through LDS (K/V buffers are free now) and store whole rows with
    // coalesced 16-byte accesses (a lane owns single bf16 values of 4 rows: direct stores would be 2 bytes per lane)
    constexpr int OP = ROWB + 16;  // padded LDS pitch of the output tile
    static_assert(FQ * OP <= ST * 2 * TILE_B, "output tile must fit in the K/V staging LDS");
    __syncthreads();
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        float l = lrun[f];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        if (OT) {   // lane = query row l15, four consecutive dims per fragment: 8-byte LDS stores
            char* orow = smem + (wave * (QF * 16) + f * 16 + l15) * OP + g * 8;
#pragma unroll
            for (int d = 0; d < DF; ++d) {
                uint2 o2;
                o2.x = pack_bf16x2(oacc[f][d][0] * inv, oacc[f][d][1] * inv);
                o2.y = pack_bf16x2(oacc[f][d][2] * inv, oacc[f][d][3] * inv);
                *reinterpret_cast<uint2*>(orow + d * 32) = o2;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = __shfl(inv, g * 4 + r, 64);
                const int rl = wave * (QF * 16) + f * 16 + g * 4 + r;  // row inside the block's tile
                uint16_t* orow = reinterpret_cast<uint16_t*>(smem + rl * OP) + l15;
#pragma unroll
                for (int d = 0; d < DF; ++d) orow[d * 16] = f32_to_bf16(oacc[f][d][r] * a);
            }
        }
    }
    __syncthreads();
    constexpr int CPR = ROWB / 16;
    for (int idx = tid; idx < FQ * CPR; idx += NW * 64) {
        const int rl = idx / CPR, c = idx - rl * CPR;
        const int row = qb * FQ + rl;
        if (row < q_len)
            *reinterpret_cast<uint4*>(out + (b * q_len + row) * ldo + (int64_t)head * D + c * 8) =
                *reinterpret_cast<const uint4*>(smem + rl * OP + c * 16);
    }
}

// K/V tile staging for the persistent kernel: wave-uniform 64-bit base in SGPRs + one 32-bit byte offset per lane (the v2
// helper's per-lane 64-bit addresses cost the persistent kernel its last registers: 29 spills, reloaded inside the tile loop).
// Every LDS-DMA of that kernel goes through this helper, so the compiler never has its own idea of M0 there.
template <int D, int NW = 4>
__device__ inline void flash_stage_tile_s(const bf16* __restrict__ kbase, const bf16* __restrict__ vbase, int key0, int n_keys,
                                          char* kbuf, char* vbuf, int wave, int lane) {
    constexpr int ROWB = D * 2;
    constexpr int SLOTS = ROWB / 16;
    constexpr int RPI = 1024 / ROWB;
    constexpr int NI = 64 / RPI / NW;
    static_assert(NI >= 1, "more waves than 1 KiB pieces in a K tile");
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int rbase = (wave * NI + i) * RPI;
        const int row = rbase + lane / SLOTS;
        const int p = lane % SLOTS;
        int kr = key0 + row;
        kr = kr < n_keys ? kr : n_keys - 1;          // never-written rows: re-read the last visible one (weight exactly 0)
        const uint32_t ko = (uint32_t)(kr * ROWB + ((p ^ (row & (SLOTS - 1))) << 4));
        const uint32_t vo = (uint32_t)(kr * ROWB + ((p ^ (((row & 7) << 1) & (SLOTS - 1))) << 4));
        const uint32_t kl = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(kbuf + rbase * ROWB);
        const uint32_t vl = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(vbuf + rbase * ROWB);
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(ko), "s"(kbase), "s"(kl) : "memory");
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(vo), "s"(vbase), "s"(vl) : "memory");
    }
}

// ---- bf16 MFMA flash prefill, v3: persistent workgroups ------------------------------------------------------------------
// Same tile math as flash_prefill_v2_kernel<D, 4, 2> with its measured options fixed (transposed O accumulators, scale in the
// exponent FMA, s_setprio), but a workgroup does not end with its 128-row query block: it walks a list of (sequence, head,
// query block) items.  Why: a 512-key prompt gives a workgroup only 2..8 key tiles, and `profiles/r2_probe_flash_lengths.jsonl`
// shows what that costs -- the SAME kernel runs 914 TF at n = 8192, 739 at 2048, 402 at 512, 168 at 128: ~10 us per
// workgroup go to dispatch, the exposed Q / first-tile load latency and the store tail, not to tiles.  Here
//   * the NEXT item's first K/V tile is put in flight (LDS-DMA into the ring stage the last tile does not use) while the
//     current item's last tile is computed, and its Q fragments are loaded as soon as the last S = K Q^T has issued, so an
//     item starts with its operands already there;
//   * the normalised output tile is staged in the last tile's ring stage (128 rows x D bf16 = exactly one K|V stage, 16-byte
//     chunks XOR-swizzled by row instead of a padded pitch) and leaves as coalesced 16-byte stores under the next item;
//   * items are assigned XCD-aware like v2's 1-D grid (workgroup p works for XCD p % 8 on groups with group % 8 == XCD, the
//     query blocks of one group run at the same time on that XCD's workgroups and share its L2), and a workgroup's position
//     in the per-XCD list changes per round so that it alternates heavy and light causal blocks (2 .. 8 tiles).
template <int D>
__global__ __launch_bounds__(256, 2) void flash_prefill_v3_kernel(const bf16* __restrict__ qkv, int64_t ld_qkv,
                                                                 const bf16* __restrict__ kc, const bf16* __restrict__ vc,
                                                                 int64_t n_max, const int32_t* __restrict__ pos0_dev,
                                                                 int pos0_host, const uint8_t* __restrict__ mask,
                                                                 int64_t mask_ld, bf16* __restrict__ out, int64_t ldo,
                                                                 int q_len, int heads, float scale, int n_groups) {
    constexpr int NW = 4, QF = 2, KK = D / 32, DF = D / 16;
    constexpr int ROWB = D * 2, SLOTS = ROWB / 16;
    constexpr int TILE_B = FK_TILE * ROWB;
    constexpr int FQ = NW * QF * 16;
    static_assert(FQ * ROWB == 2 * TILE_B, "the output tile must be exactly one K|V ring stage");
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * TILE_B];  // [stage][K | V]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int pos0 = pos0_dev ? *pos0_dev : pos0_host;
    const float sl2 = scale * 1.44269504088896340736f;                 // softmax in base 2
    const int nq = (q_len + FQ - 1) / FQ;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, S = gridDim.x >> 3;
    const int lim = ((n_groups - xcd + 7) >> 3) * nq;                  // items of this XCD: its groups x query blocks

    // item of round k for this workgroup (false: none left -- and then none in any later round either)
    auto item_of = [&](int k, int& qb, int& head, int64_t& b) -> bool {
        // position inside the round's S consecutive per-XCD items: when whole groups fit (S % nq == 0) odd rounds mirror the
        // query-block index, so a workgroup alternates heavy / light causal blocks (8, 2, 8, 2 | 6, 4, 6, 4 tiles: equal work
        // every two rounds -- the workgroups sharing a group's K/V through L2 stay in step); otherwise rotate by one per round
        const int sl = (S % nq == 0) ? ((k & 1) ? slot - slot % nq + (nq - 1 - slot % nq) : slot) : (slot + k) % S;
        const int j = k * S + sl;
        if (j >= lim) return false;
        const int gi = j / nq, qi = j - gi * nq;
        const int grp = gi * 8 + xcd;
        qb = nq - 1 - qi;                                              // heaviest causal block of a group first
        head = grp % heads;
        b = grp / heads;
        return true;
    };
    auto load_q = [&](int qb, int head, int64_t b, bf16x8_t (&qf)[QF][KK]) {
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            const int row = qb * FQ + wave * (QF * 16) + f * 16 + l15;
            const int r = row < q_len ? row : q_len - 1;
            const bf16* qp = qkv + (b * q_len + r) * ld_qkv + (int64_t)head * 3 * D + g * 8;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) qf[f][kk] = *reinterpret_cast<const bf16x8_t*>(qp + kk * 32);
        }
    };
    auto n_keys_of = [&](int qb) { return pos0 + min(q_len, (qb + 1) * FQ); };

    // lane-constant LDS offsets of the fragment reads (see v2)
    int kaddr[KK], vaddr[DF];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) kaddr[kk] = l15 * ROWB + (((kk * 4 + g) ^ (l15 & (SLOTS - 1))) * 16);
    {
        const int rl = g * 4 + (l15 >> 2);
        const int swz = (((rl & 7) << 1) & (SLOTS - 1)) << 4;
#pragma unroll
        for (int d = 0; d < DF; ++d) vaddr[d] = rl * ROWB + (((d * 16 + (l15 & 3) * 4) * 2) ^ swz);
    }

    int qb, head;
    int64_t b;
    int k = 0;
    if (!item_of(0, qb, head, b)) return;
    constexpr int ST_FULL = FQ * SLOTS / (NW * 64);   // output store instructions per thread of a full 128-row block
    int pend_stores = 0;   // store instructions this wave issued in the previous item's epilogue (ST_FULL or "unknown" = 0)
    int gt = 0;   // key tiles this workgroup has consumed so far: tile kt of the current item lives in ring stage (gt + kt) & 1
    bf16x8_t qf[QF][KK];
    {   // first item: tile 0 into the ring, then the Q fragments (younger: the first in-loop wait covers both)
        const bf16* kbase0 = kc + (b * heads + head) * n_max * D;
        const bf16* vbase0 = vc + (b * heads + head) * n_max * D;
        flash_stage_tile_s<D, NW>(kbase0, vbase0, 0, n_keys_of(qb), smem, smem + TILE_B, wave, lane);
        load_q(qb, head, b, qf);
    }
    while (true) {
        int nqb = 0, nhead = 0;
        int64_t nb = 0;
        const bool has_next = item_of(k + 1, nqb, nhead, nb);
        const int qbase = qb * FQ + wave * (QF * 16);
        const bf16* kbase = kc + (b * heads + head) * n_max * D;
        const bf16* vbase = vc + (b * heads + head) * n_max * D;
        const uint8_t* mrow = mask ? mask + b * mask_ld : nullptr;
        const int n_keys = n_keys_of(qb);
        const int n_tiles = (n_keys + FK_TILE - 1) / FK_TILE;
        const int wave_last_pos = pos0 + min(qbase + QF * 16, q_len) - 1;  // last query position this wave owns
        int qrow[QF];
#pragma unroll
        for (int f = 0; f < QF; ++f) qrow[f] = qbase + f * 16 + l15;
        f32x4_t oacc[QF][DF];   // O^T accumulators: oacc[f][d][r] = O[query f*16 + l15][dim d*16 + g*4 + r]
#pragma unroll
        for (int f = 0; f < QF; ++f)
#pragma unroll
            for (int d = 0; d < DF; ++d) oacc[f][d] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        float mrun[QF], lrun[QF];   // running max in RAW score units, running sum
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            mrun[f] = -INFINITY;
            lrun[f] = 0.f;
        }

        for (int kt = 0; kt < n_tiles; ++kt) {
            const int key0 = kt * FK_TILE;
            const char* kb = smem + ((gt + kt) & 1) * 2 * TILE_B;
            const char* vb = kb + TILE_B;
            // this wave's share of tile kt has landed.  On an item's first tile that is the prefetched tile and the Q fragments,
            // both OLDER than the previous item's output stores (vmcnt retires in order, and gfx9 counts stores in it): wait
            // for everything but those stores instead of paying their acknowledge latency once per item
            if (kt == 0 && pend_stores == ST_FULL) {
                if (ST_FULL == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (ST_FULL == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();   // tile kt visible to everybody; everybody is done with the other stage -> it can be refilled
            {
                char* nbuf = smem + ((gt + kt + 1) & 1) * 2 * TILE_B;
                if (kt + 1 < n_tiles) {
                    flash_stage_tile_s<D, NW>(kbase, vbase, key0 + FK_TILE, n_keys, nbuf, nbuf + TILE_B, wave, lane);
                } else if (has_next) {   // the next item's first tile rides under this item's last one
                    const bf16* kn = kc + (nb * heads + nhead) * n_max * D;
                    const bf16* vn = vc + (nb * heads + nhead) * n_max * D;
                    flash_stage_tile_s<D, NW>(kn, vn, 0, n_keys_of(nqb), nbuf, nbuf + TILE_B, wave, lane);
                }
            }
            const bool visible = key0 <= wave_last_pos;   // wave-uniform: some query of this wave can see this tile
            f32x4_t sacc[QF][4];
            if (visible) {
#pragma unroll
                for (int f = 0; f < QF; ++f)
#pragma unroll
                    for (int kf = 0; kf < 4; ++kf) sacc[f][kf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kf = 0; kf < 4; ++kf) {
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk) {
                        const bf16x8_t kfr = *reinterpret_cast<const bf16x8_t*>(kb + kaddr[kk] + kf * 16 * ROWB);
#pragma unroll
                        for (int f = 0; f < QF; ++f)
                            sacc[f][kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[f][kk], sacc[f][kf], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_s_setprio(0);
            }
            // the current Q fragments have issued their last MFMA: the next item's go in flight under softmax, PV and the epilogue
            if (kt == n_tiles - 1 && has_next) load_q(nqb, nhead, nb, qf);
            if (visible) {
                const bool need_mask = (mrow != nullptr) || (key0 + FK_TILE > n_keys) || (key0 + FK_TILE - 1 > pos0 + qbase);
                if (need_mask && mrow == nullptr && key0 + FK_TILE <= n_keys) {
                    // causal diagonal of an unpadded batch (the common masked tile): compares and selects only
#pragma unroll
                    for (int f = 0; f < QF; ++f) {
                        const int qpos = pos0 + qrow[f];
#pragma unroll
                        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                sacc[f][kf][r] = (key0 + kf * 16 + g * 4 + r <= qpos) ? sacc[f][kf][r] : -INFINITY;
                    }
                } else if (need_mask) {
                    // padding mask / ragged end: the 16 mask bytes of the lane's keys are loaded up front (clamped index, all
                    // in flight together, one wait) instead of one dependent byte load per key
                    bool kvis[4][4];
                    if (mrow != nullptr) {
                        uint8_t mb[4][4];
#pragma unroll
                        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                            for (int r = 0; r < 4; ++r) mb[kf][r] = mrow[min(key0 + kf * 16 + g * 4 + r, n_keys - 1)];
#pragma unroll
                        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                            for (int r = 0; r < 4; ++r) kvis[kf][r] = (key0 + kf * 16 + g * 4 + r < n_keys) && mb[kf][r] != 0;
                    } else {
#pragma unroll
                        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                            for (int r = 0; r < 4; ++r) kvis[kf][r] = key0 + kf * 16 + g * 4 + r < n_keys;
                    }
#pragma unroll
                    for (int f = 0; f < QF; ++f) {
                        const int qpos = pos0 + qrow[f];
#pragma unroll
                        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int key = key0 + kf * 16 + g * 4 + r;
                                sacc[f][kf][r] = (key <= qpos && kvis[kf][r]) ? sacc[f][kf][r] : -INFINITY;
                            }
                    }
                }
#pragma unroll
                for (int f = 0; f < QF; ++f) {
                    float tmax = -INFINITY;
#pragma unroll
                    for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                        for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, sacc[f][kf][r]);
                    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
                    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                    const float mnew = fmaxf(mrun[f], tmax);
                    const bool dead = (mnew == -INFINITY);          // no visible key so far for this query row
                    const float msub = dead ? 0.f : -mnew * sl2;
                    const float alpha = dead ? 1.0f : __builtin_amdgcn_exp2f((mrun[f] - mnew) * sl2);
                    float psum = 0.f;
#pragma unroll
                    for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[f][kf][r], sl2, msub));   // exp2(-inf) = 0: masked keys
                            sacc[f][kf][r] = pv;
                            psum += pv;
                        }
                    lrun[f] = lrun[f] * alpha + psum;
                    mrun[f] = mnew;
#pragma unroll
                    for (int d = 0; d < DF; ++d)
#pragma unroll
                        for (int r = 0; r < 4; ++r) oacc[f][d][r] *= alpha;   // the lane's column IS query row l15: its own alpha
                }
                // O^T += V^T P^T.  key slot (g, e): e < 4 -> tile key (2ks)*16 + g*4 + e ; e >= 4 -> (2ks+1)*16 + g*4 + (e-4)
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8_t pf[QF];
#pragma unroll
                    for (int f = 0; f < QF; ++f) {
                        union {
                            uint32_t u[4];
                            bf16x8_t v;
                        } pk;
                        pk.u[0] = pack_bf16x2(sacc[f][2 * ks][0], sacc[f][2 * ks][1]);
                        pk.u[1] = pack_bf16x2(sacc[f][2 * ks][2], sacc[f][2 * ks][3]);
                        pk.u[2] = pack_bf16x2(sacc[f][2 * ks + 1][0], sacc[f][2 * ks + 1][1]);
                        pk.u[3] = pack_bf16x2(sacc[f][2 * ks + 1][2], sacc[f][2 * ks + 1][3]);
                        pf[f] = pk.v;
                    }
#pragma unroll
                    for (int d = 0; d < DF; ++d) {
                        const int a1 = vaddr[d] + (2 * ks) * 16 * ROWB;
                        const int a2 = a1 + 16 * ROWB;
                        union {
                            s16x4_t h[2];
                            bf16x8_t v;
                        } vf;
                        vf.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vb + a1));
                        vf.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vb + a2));
#pragma unroll
                        for (int f = 0; f < QF; ++f)
                            oacc[f][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf[f], oacc[f][d], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_s_setprio(0);
            }
        }
        // epilogue: normalise, stage the 128 x D tile in the LAST tile's ring stage (the other one is receiving the next
        // item's first tile), 16-byte chunks swizzled by row; then whole rows leave as coalesced 16-byte stores
        char* ost = smem + ((gt + n_tiles - 1) & 1) * 2 * TILE_B;
        __syncthreads();   // every wave is done reading the last tile's K / V
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            float l = lrun[f];
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
            const float inv = l > 0.f ? 1.0f / l : 0.f;
            const int rl = wave * (QF * 16) + f * 16 + l15;
            char* orow = ost + rl * ROWB + (g & 1) * 8;
#pragma unroll
            for (int d = 0; d < DF; ++d) {
                uint2 o2;
                o2.x = pack_bf16x2(oacc[f][d][0] * inv, oacc[f][d][1] * inv);
                o2.y = pack_bf16x2(oacc[f][d][2] * inv, oacc[f][d][3] * inv);
                const int c = d * 2 + (g >> 1);
                *reinterpret_cast<uint2*>(orow + ((c ^ (rl & (SLOTS - 1))) << 4)) = o2;
            }
        }
        __syncthreads();
        for (int idx = tid; idx < FQ * SLOTS; idx += NW * 64) {
            const int rl = idx / SLOTS, c = idx - rl * SLOTS;
            const int row = qb * FQ + rl;
            if (row < q_len)
                *reinterpret_cast<uint4*>(out + (b * q_len + row) * ldo + (int64_t)head * D + c * 8) =
                    *reinterpret_cast<const uint4*>(ost + rl * ROWB + ((c ^ (rl & (SLOTS - 1))) << 4));
        }
        if (!has_next) break;
        pend_stores = ((qb + 1) * FQ <= q_len) ? ST_FULL : 0;   // a block cut by q_len skips stores per wave: wait for all
        gt += n_tiles;
        qb = nqb;
        head = nhead;
        b = nb;
        ++k;
    }
}

// ---- host dispatch ------------------------------------------------------------------------------------------------------
#ifdef BT_DEBUG_HOOKS
static int g_attn_nt = 0;   // A/B hook (debug library only): non-temporal K/V loads in the global (NW = 4) streaming kernel
static int g_attn_persist = 0;   // A/B hook: > 0 = persistent global-attention kernel with that many workgroups per CU
static int g_local_kernel = 1;   // A/B hook: 0 = the token-decoder window runs on the generic row-wise kernel (NW = 1)
#else
static constexpr int g_attn_persist = 0;
static constexpr int g_local_kernel = 1;
#endif
#ifdef BT_DEBUG_HOOKS
static int g_attn_wide = 1;      // A/B hook: 0 = tiny batches keep split-N + combine for the global step
extern "C" void bt_debug_attn_wide(int on) { g_attn_wide = on; }
#else
static constexpr int g_attn_wide = 1;
#endif

template <typename T, int D>
static int launch_attend(int64_t B, int q_len, int heads, const void* qkv, int64_t ld_qkv, const void* kc, const void* vc,
                         int64_t n_max, const int32_t* pos0_dev, int pos0_host, const uint8_t* mask, int64_t mask_ld,
                         void* out, int64_t ldo, int n_split, float* ws, hipStream_t s, const int32_t* seq_skip = nullptr) {
    const float scale = 1.0f / sqrtf((float)D);
    ProfScope ps(BT_PROF_ATTN_ROWS, s);
    const int64_t cap = ceil_div(n_max, n_split);
    const dim3 grid((unsigned)(B * heads * q_len), (unsigned)n_split);
    if constexpr (D == 128 && sizeof(T) == 2) {
        // token-decoder local window: the dedicated wave-per-(sequence, head) kernel (host-known position, <= 12 keys, <= 2 rows)
        if (g_local_kernel && n_split == 1 && pos0_dev == nullptr && pos0_host + q_len <= 12 && q_len <= 2 && n_max <= 64) {
            const int64_t items = B * heads;
            constexpr int HPW = 2;
            if (pos0_host + q_len <= 8)
                hipLaunchKernelGGL((local_attend_kernel<128, HPW, 2>), dim3((unsigned)ceil_div(items, 4 * HPW)), dim3(256), 0, s,
                                   (const bf16*)qkv, ld_qkv, (const bf16*)kc, (const bf16*)vc, n_max, pos0_host, mask, mask_ld,
                                   (bf16*)out, ldo, q_len, heads, items, scale);
            else
                hipLaunchKernelGGL((local_attend_kernel<128, HPW, 3>), dim3((unsigned)ceil_div(items, 4 * HPW)), dim3(256), 0, s,
                                   (const bf16*)qkv, ld_qkv, (const bf16*)kc, (const bf16*)vc, n_max, pos0_host, mask, mask_ld,
                                   (bf16*)out, ldo, q_len, heads, items, scale);
            BT_LAUNCH_CHECK("bt_attention(local window) launch");
            return BT_OK;
        }
    }
    // a handful of (sequence, head) items against a long cache (batch 1-4 block-decoder steps): split-N over 8-16 small workgroups
    // plus a combine launch is two dependent launches of ~7 + ~5 us; ONE 16-wave workgroup per item sweeps 512 keys per pass with
    // 8 loads in flight per lane and combines in LDS (round 4; the split request and its workspace are then simply not used)
    if (g_attn_wide && n_split > 1 && sizeof(T) == 2 && (int64_t)B * heads * q_len <= 64 && !seq_skip &&
        (16 + 16 * D + n_max) * sizeof(float) <= 64 * 1024) {
        const size_t shm = (16 + 16 * D + n_max) * sizeof(float);
        hipLaunchKernelGGL((attend_rows_kernel<T, D, 16>), dim3((unsigned)(B * heads * q_len), 1), dim3(1024), shm, s, (const T*)qkv,
                           ld_qkv, (const T*)kc, (const T*)vc, n_max, pos0_dev, pos0_host, mask, mask_ld, (T*)out, ldo, ws, q_len,
                           heads, 1, scale);
        BT_LAUNCH_CHECK("bt_attention(rows, 16 waves) launch");
        return BT_OK;
    }
    if (n_max <= 64 && n_split == 1) {
        const size_t shm = (4 + 1 * D + cap) * sizeof(float);
        hipLaunchKernelGGL((attend_rows_kernel<T, D, 1>), grid, dim3(64), shm, s, (const T*)qkv, ld_qkv, (const T*)kc,
                           (const T*)vc, n_max, pos0_dev, pos0_host, mask, mask_ld, (T*)out, ldo, ws, q_len, heads,
                           n_split, scale);
    } else {
        const size_t shm = (4 + 4 * D + cap) * sizeof(float);
        if (shm > 64 * 1024) {
            set_error("bt_attention: %lld keys per split need %zu B of LDS; raise n_split", (long long)cap, shm);
            return BT_ERR_ARG;
        }
        if (g_attn_persist > 0 && n_split == 1) {   // bounded-footprint variant: g_attn_persist workgroups per CU walk the items
            const int64_t items = B * heads * q_len;
            int dev = 0, cus = 256;
            (void)hipGetDevice(&dev);
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            const dim3 pgrid((unsigned)std::min<int64_t>(items, (int64_t)g_attn_persist * cus), 1);
            hipLaunchKernelGGL((attend_rows_kernel<T, D, 4, false, true>), pgrid, dim3(256), shm, s, (const T*)qkv, ld_qkv,
                               (const T*)kc, (const T*)vc, n_max, pos0_dev, pos0_host, mask, mask_ld, (T*)out, ldo, ws, q_len,
                               heads, n_split, scale, items);
        } else
#ifdef BT_DEBUG_HOOKS
        if (g_attn_nt)
            hipLaunchKernelGGL((attend_rows_kernel<T, D, 4, true>), grid, dim3(256), shm, s, (const T*)qkv, ld_qkv,
                               (const T*)kc, (const T*)vc, n_max, pos0_dev, pos0_host, mask, mask_ld, (T*)out, ldo, ws, q_len,
                               heads, n_split, scale);
        else
#endif
            hipLaunchKernelGGL((attend_rows_kernel<T, D, 4>), grid, dim3(256), shm, s, (const T*)qkv, ld_qkv, (const T*)kc,
                               (const T*)vc, n_max, pos0_dev, pos0_host, mask, mask_ld, (T*)out, ldo, ws, q_len, heads,
                               n_split, scale, (int64_t)0, seq_skip);
    }
    BT_LAUNCH_CHECK("bt_attention(rows) launch");
    if (n_split > 1) {
        hipLaunchKernelGGL(attn_combine_kernel<T>, dim3((unsigned)(B * heads * q_len)), dim3(D < 64 ? 64 : D), 0, s, ws,
                           (T*)out, ldo, q_len, heads, D, n_split);
        BT_LAUNCH_CHECK("bt_attention(combine) launch");
    }
    return BT_OK;
}

#ifdef BT_DEBUG_HOOKS
static int g_flash_variant = 0;
#else
static constexpr int g_flash_variant = 0;
#endif
                                 // 0 = v2 <4 waves, double buffer> (LDS-DMA + transpose reads), 2 = v2 <8 waves, 4-deep ring>,
                                 // 1 = v1 (register/LDS-transposed V)

#define BT_FLASH_OPT(OPT_)                                                                                              \
    hipLaunchKernelGGL((flash_prefill_v2_kernel<D, 4, 2, OPT_>), ((OPT_) & 64) ? grid1 : grid, dim3(256), 0, s, (const bf16*)qkv, ld_qkv, \
                       (const bf16*)kc, (const bf16*)vc, n_max, pos0_dev, pos0_host, mask, mask_ld, (bf16*)out, ldo, q_len, \
                       heads, scale)
static int flash_num_cus() {
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

template <int D>
static int launch_flash(int64_t B, int q_len, int heads, const void* qkv, int64_t ld_qkv, const void* kc, const void* vc,
                        int64_t n_max, const int32_t* pos0_dev, int pos0_host, const uint8_t* mask, int64_t mask_ld,
                        void* out, int64_t ldo, hipStream_t s) {
    const float scale = 1.0f / sqrtf((float)D);
    ProfScope ps(BT_PROF_ATTN_FLASH, s);
    const dim3 grid((unsigned)ceil_div(q_len, FQ_ROWS), (unsigned)heads, (unsigned)B);
    const int64_t n_wg = ceil_div(q_len, FQ_ROWS) * heads * B;
    BT_CHECK_ARG(n_wg < (1ll << 31), "bt_attention: prefill grid too large");
    const dim3 grid1((unsigned)n_wg);   // 1-D, XCD-aware mapping (OPT bit 64)
    if (g_flash_variant == 0 && ceil_div(q_len, FQ_ROWS) <= 8) {
        // production for prompts of <= 1024 block positions (2 .. 16 key tiles per item: per-item fixed costs dominate; longer
        // prompts keep the one-workgroup-per-block grid, whose dynamic dispatch balances 2 .. 32+-tile items better: n = 2048
        // 733 vs 686 TF): persistent workgroups, two per CU; 8 x S of them, S = slots per XCD
        const int64_t n_groups = (int64_t)heads * B;
        BT_CHECK_ARG(n_groups < (1ll << 31), "bt_attention: too many (sequence, head) groups");
        const int64_t per_xcd = ceil_div(n_groups, 8) * ceil_div(q_len, FQ_ROWS);
        const int S = (int)std::min<int64_t>(std::max(1, 2 * flash_num_cus() / 8), per_xcd);
        hipLaunchKernelGGL((flash_prefill_v3_kernel<D>), dim3(8 * S), dim3(256), 0, s, (const bf16*)qkv, ld_qkv, (const bf16*)kc,
                           (const bf16*)vc, n_max, pos0_dev, pos0_host, mask, mask_ld, (bf16*)out, ldo, q_len, heads, scale,
                           (int)n_groups);
        BT_LAUNCH_CHECK("bt_attention(flash) launch");
        return BT_OK;
    }
#ifdef BT_DEBUG_HOOKS
    if (g_flash_variant == 1)
        hipLaunchKernelGGL(flash_prefill_kernel<D>, grid, dim3(256), 0, s, (const bf16*)qkv, ld_qkv, (const bf16*)kc,
                           (const bf16*)vc, n_max, pos0_dev, pos0_host, mask, mask_ld, (bf16*)out, ldo, q_len, heads,
                           scale);
    else
#endif
#ifdef BT_DEBUG_HOOKS
    if ((g_flash_variant >= 3 && g_flash_variant <= 7) || (g_flash_variant >= 9 && g_flash_variant <= 13)) {   // A/B of the OPT bits: 3 -> 0 (round-1 kernel), 4 -> 1, 5 -> 3, 6 -> 7, 7 -> 15
        switch (g_flash_variant) {
            case 3: BT_FLASH_OPT(0); break;
            case 9: BT_FLASH_OPT(31); break;    // + scale folded into the exponent FMA
            case 12: BT_FLASH_OPT(95); break;   // 31 + XCD-aware 1-D grid
            case 13: BT_FLASH_OPT(95); break;   // the non-persistent kernel the production v3 grew out of (A/B)
            case 10: BT_FLASH_OPT(63); break;   // + lazy rescale
            case 11: BT_FLASH_OPT(59); break;   // 63 without setprio
            case 4: BT_FLASH_OPT(1); break;
            case 5: BT_FLASH_OPT(3); break;
            case 6: BT_FLASH_OPT(7); break;
            default: BT_FLASH_OPT(15); break;   // 7: transposed O accumulators (bit 8) + setprio
        }
    } else
#endif
    if ((g_flash_variant != 2 && g_flash_variant != 8) || D < 64)   // production: 4 waves, double buffer, two workgroups per CU
        hipLaunchKernelGGL((flash_prefill_v2_kernel<D, 4, 2>), (FLASH_OPT & 64) ? grid1 : grid, dim3(256), 0, s, (const bf16*)qkv, ld_qkv,
                           (const bf16*)kc, (const bf16*)vc, n_max, pos0_dev, pos0_host, mask, mask_ld, (bf16*)out, ldo, q_len,
                           heads, scale);
#ifdef BT_DEBUG_HOOKS
    else if constexpr (D >= 64) {
        if (g_flash_variant == 8) {   // 8 waves x 16 query rows, double buffer, two workgroups per CU (four waves per SIMD)
            hipLaunchKernelGGL((flash_prefill_v2_kernel<D, 8, 2, FLASH_OPT, 1>), (FLASH_OPT & 64) ? grid1 : grid, dim3(512), 0, s, (const bf16*)qkv, ld_qkv,
                               (const bf16*)kc, (const bf16*)vc, n_max, pos0_dev, pos0_host, mask, mask_ld, (bf16*)out, ldo,
                               q_len, heads, scale);
            BT_LAUNCH_CHECK("bt_attention(flash) launch");
            return BT_OK;
        }
        // variant 2 (A/B): 8 waves, 4-deep ring -- same speed: the loop is issue-bound, not load-bound
        const dim3 grid8 = (FLASH_OPT & 64) ? dim3((unsigned)(ceil_div(q_len, 256) * heads * B))
                                            : dim3((unsigned)ceil_div(q_len, 256), (unsigned)heads, (unsigned)B);
        hipLaunchKernelGGL((flash_prefill_v2_kernel<D, 8, 4>), grid8, dim3(512), 0, s, (const bf16*)qkv, ld_qkv,
                           (const bf16*)kc, (const bf16*)vc, n_max, pos0_dev, pos0_host, mask, mask_ld, (bf16*)out, ldo, q_len,
                           heads, scale);
    }
#endif
    BT_LAUNCH_CHECK("bt_attention(flash) launch");
    return BT_OK;
}

}  // namespace bt

using namespace bt;

extern "C" int bt_rope_kv_append(int dtype, int64_t B, int32_t q_len, int32_t heads, int32_t head_dim, int32_t rot,
                                 void* qkv, int64_t ld_qkv, void* k_cache, void* v_cache, int64_t n_max,
                                 const int32_t* pos0_dev, int32_t pos0_host, const int32_t* pos_table,
                                 const float* cos_tab, const float* sin_tab, bt_stream stream) {
    BT_CHECK_ARG(dtype == BT_F32 || dtype == BT_BF16, "bt_rope_kv_append: bad dtype %d", dtype);
    BT_CHECK_ARG(B >= 0 && q_len > 0 && heads > 0 && head_dim > 0, "bt_rope_kv_append: bad shape");
    BT_CHECK_ARG(qkv && k_cache && v_cache && cos_tab && sin_tab, "bt_rope_kv_append: null pointer");
    const int V = dtype == BT_BF16 ? 8 : 4;
    const int half = rot / 2;
    BT_CHECK_ARG(rot >= 2 && rot % 2 == 0 && rot <= head_dim, "bt_rope_kv_append: bad rotary_ndims %d", rot);
    BT_CHECK_ARG(head_dim % V == 0 && ld_qkv % V == 0 && aligned16(qkv) && aligned16(k_cache) && aligned16(v_cache),
                 "bt_rope_kv_append: head_dim/stride/pointers must be 16-byte granular");
    BT_CHECK_ARG((half % V == 0) || (V % rot == 0), "bt_rope_kv_append: rotary_ndims %d not supported for this dtype", rot);
    BT_CHECK_ARG(ld_qkv >= (int64_t)3 * heads * head_dim, "bt_rope_kv_append: qkv row stride too small");
    BT_CHECK_ARG(pos0_dev || (pos0_host >= 0 && pos0_host + q_len <= n_max), "bt_rope_kv_append: cache overflow");
    if (B == 0) return BT_OK;
    const int64_t rows = B * q_len;
    const int64_t total = rows * (heads * 3 * head_dim / V);
    const unsigned grid = (unsigned)ceil_div(total, 256);
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(BT_PROF_ROPE, s);
    if (dtype == BT_BF16)
        hipLaunchKernelGGL(rope_kv_append_kernel<bf16>, dim3(grid), dim3(256), 0, s, (bf16*)qkv, ld_qkv, (bf16*)k_cache,
                           (bf16*)v_cache, n_max, rows, q_len, heads, head_dim, rot, pos0_dev, pos0_host, pos_table,
                           cos_tab, sin_tab);
    else
        hipLaunchKernelGGL(rope_kv_append_kernel<float>, dim3(grid), dim3(256), 0, s, (float*)qkv, ld_qkv,
                           (float*)k_cache, (float*)v_cache, n_max, rows, q_len, heads, head_dim, rot, pos0_dev,
                           pos0_host, pos_table, cos_tab, sin_tab);
    BT_LAUNCH_CHECK("bt_rope_kv_append launch");
    return BT_OK;
}

// effective split-N factor of the row-wise kernel: the hint, raised until one split's score row fits the 64 KiB LDS budget
extern "C" int32_t bt_attention_split_count(int32_t head_dim, int64_t n_max, int32_t n_split_hint) {
    int32_t n_split = n_split_hint < 1 ? 1 : n_split_hint;
    while ((4 + 4 * (int64_t)head_dim + bt::ceil_div(n_max, n_split)) * 4 > 64 * 1024) ++n_split;
    return n_split;
}

namespace bt {
int attention_impl(int dtype, int mode, int64_t B, int32_t q_len, int32_t heads, int32_t head_dim, const void* qkv,
                   int64_t ld_qkv, const void* k_cache, const void* v_cache, int64_t n_max, const int32_t* pos0_dev,
                   int32_t pos0_host, const uint8_t* mask, int64_t mask_ld, void* out, int64_t ldo, int32_t n_split, void* ws,
                   int64_t ws_bytes, const int32_t* seq_skip, bt_stream stream);
}

extern "C" int bt_attention(int dtype, int mode, int64_t B, int32_t q_len, int32_t heads, int32_t head_dim,
                            const void* qkv, int64_t ld_qkv, const void* k_cache, const void* v_cache, int64_t n_max,
                            const int32_t* pos0_dev, int32_t pos0_host, const uint8_t* mask, int64_t mask_ld, void* out,
                            int64_t ldo, int32_t n_split, void* ws, int64_t ws_bytes, bt_stream stream) {
    return bt::attention_impl(dtype, mode, B, q_len, heads, head_dim, qkv, ld_qkv, k_cache, v_cache, n_max, pos0_dev, pos0_host,
                              mask, mask_ld, out, ldo, n_split, ws, ws_bytes, nullptr, stream);
}

// seq_skip[b] != 0: sequence b is finished -- the row-wise kernel (global decode step, n_max > 64) leaves zeros for it instead of
// streaming its cache (bt_decoder_forward passes bt_attn_args.seq_skip)
int bt::attention_impl(int dtype, int mode, int64_t B, int32_t q_len, int32_t heads, int32_t head_dim, const void* qkv,
                       int64_t ld_qkv, const void* k_cache, const void* v_cache, int64_t n_max, const int32_t* pos0_dev,
                       int32_t pos0_host, const uint8_t* mask, int64_t mask_ld, void* out, int64_t ldo, int32_t n_split,
                       void* ws, int64_t ws_bytes, const int32_t* seq_skip, bt_stream stream) {
    BT_CHECK_ARG(dtype == BT_F32 || dtype == BT_BF16, "bt_attention: bad dtype %d", dtype);
    BT_CHECK_ARG(mode == 0 || mode == 1, "bt_attention: bad mode %d", mode);
    BT_CHECK_ARG(B >= 0 && q_len > 0 && heads > 0, "bt_attention: bad shape");
    BT_CHECK_ARG(qkv && k_cache && v_cache && out, "bt_attention: null pointer");
    BT_CHECK_ARG(head_dim == 32 || head_dim == 64 || head_dim == 128, "bt_attention: head_dim %d not instantiated",
                 head_dim);
    const int V = dtype == BT_BF16 ? 8 : 4;
    BT_CHECK_ARG(ld_qkv % V == 0 && ldo % V == 0 && aligned16(qkv) && aligned16(out) && aligned16(k_cache) &&
                     aligned16(v_cache),
                 "bt_attention: strides/pointers must be 16-byte granular");
    BT_CHECK_ARG(pos0_dev || (pos0_host >= 0 && pos0_host + q_len <= n_max), "bt_attention: cache overflow");
    BT_CHECK_ARG(!mask || mask_ld >= (pos0_dev ? 1 : pos0_host + q_len), "bt_attention: mask row too short");
    if (B == 0) return BT_OK;
    BT_CHECK_ARG(B * heads * q_len < (1ll << 31), "bt_attention: grid too large");
    hipStream_t s = (hipStream_t)stream;
    if (mode == 1 && dtype == BT_BF16) {
        BT_CHECK_ARG(B < 65536 && heads < 65536, "bt_attention: flash grid too large");
        switch (head_dim) {
            case 32: return launch_flash<32>(B, q_len, heads, qkv, ld_qkv, k_cache, v_cache, n_max, pos0_dev, pos0_host, mask, mask_ld, out, ldo, s);
            case 64: return launch_flash<64>(B, q_len, heads, qkv, ld_qkv, k_cache, v_cache, n_max, pos0_dev, pos0_host, mask, mask_ld, out, ldo, s);
            default: return launch_flash<128>(B, q_len, heads, qkv, ld_qkv, k_cache, v_cache, n_max, pos0_dev, pos0_host, mask, mask_ld, out, ldo, s);
        }
    }
    n_split = bt_attention_split_count(head_dim, n_max, n_split);
    if (n_split > 1) {
        const int64_t need = B * heads * q_len * n_split * (head_dim + 2) * (int64_t)sizeof(float);
        if (!ws || ws_bytes < need) {
            set_error("bt_attention: split workspace too small (%lld < %lld bytes)", (long long)ws_bytes, (long long)need);
            return BT_ERR_WORKSPACE;
        }
    }
#define BT_ATT(T, D_) launch_attend<T, D_>(B, q_len, heads, qkv, ld_qkv, k_cache, v_cache, n_max, pos0_dev, pos0_host, mask, mask_ld, out, ldo, n_split, (float*)ws, s, seq_skip)
    if (dtype == BT_BF16) {
        switch (head_dim) {
            case 32: return BT_ATT(bf16, 32);
            case 64: return BT_ATT(bf16, 64);
            default: return BT_ATT(bf16, 128);
        }
    }
    switch (head_dim) {
        case 32: return BT_ATT(float, 32);
        case 64: return BT_ATT(float, 64);
        default: return BT_ATT(float, 128);
    }
#undef BT_ATT
}

#ifdef BT_DEBUG_HOOKS
extern "C" void bt_debug_flash_variant(int v) { bt::g_flash_variant = v; }
extern "C" void bt_debug_attn_nt(int v) { bt::g_attn_nt = v; }
extern "C" void bt_debug_attn_persist(int v) { bt::g_attn_persist = v; }
extern "C" void bt_debug_local_kernel(int v) { bt::g_local_kernel = v; }
#endif
