// Row gather (embedder lookup+concat, embed_in), greedy argmax, top-k/temperature sampler and the roll-over bookkeeping
// kernels.  All HBM-bound byte/index work: coalesced 16-byte accesses, one wave or one workgroup per row.
#include "bt_common.h"

#include <algorithm>

namespace bt {

// ---- bt_embed_rows ----------------------------------------------------------------------------------------------------
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void embed_rows_kernel(const int64_t* __restrict__ ids, int64_t ids_stride, int64_t vocab,
                                                        const T* __restrict__ table, T* __restrict__ out, int64_t ldo,
                                                        int64_t rows, int width, const int32_t* __restrict__ row_zero,
                                                        int rows_per_seq) {
    constexpr int V = Elem<T>::VEC;
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const bool zero = row_zero != nullptr && row_zero[r / rows_per_seq] != 0;
    int64_t id = ids[r * ids_stride];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // memory safety; torch would device-assert
    const T* src = table + id * (int64_t)width;
    T* dst = out + r * ldo;
    if (VEC) {
        for (int c = lane * V; c < width; c += 64 * V) {
            uint4 v = zero ? make_uint4(0, 0, 0, 0) : *reinterpret_cast<const uint4*>(src + c);
            *reinterpret_cast<uint4*>(dst + c) = v;
        }
    } else {
        for (int c = lane; c < width; c += 64) Elem<T>::store(dst + c, zero ? 0.f : Elem<T>::load(src + c));
    }
}

// ---- bt_argmax -----------------------------------------------------------------------------------------------------
// NT threads per row: 256 when there are rows enough to fill the chip, 1024 for the few rows of small-batch decode, where the kernel is
// one workgroup's chain of dependent 16-byte loads (V = 50304: 25 round trips with 256 threads -- 11 us measured at batch 1 -- 7 with 1024)
template <typename T, bool VEC, int NT>
__global__ __launch_bounds__(NT) void argmax_kernel(const T* __restrict__ logits, int64_t ld, int V_,
                                                   int64_t* __restrict__ ids_out, int64_t ids_stride) {
    constexpr int V = Elem<T>::VEC;
    constexpr int NWV = NT / 64;
    __shared__ float s_val[NWV];
    __shared__ int s_idx[NWV];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* row = logits + (int64_t)blockIdx.x * ld;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    if (VEC) {
        for (int c = tid * V; c < V_; c += NT * V) {
            float v[V];
            load_vec<T>(row + c, v);
#pragma unroll
            for (int i = 0; i < V; ++i)
                if (v[i] > best || bidx == 0x7fffffff) { best = v[i]; bidx = c + i; }
        }
    } else {
        for (int c = tid; c < V_; c += NT) {
            const float v = Elem<T>::load(row + c);
            if (v > best || bidx == 0x7fffffff) { best = v; bidx = c; }
        }
    }
    // wave reduce: larger value wins, ties -> smaller index (torch.argmax returns the first maximal index)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    if (lane == 0) { s_val[wave] = best; s_idx[wave] = bidx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < NWV; ++w)
            if (s_val[w] > best || (s_val[w] == best && s_idx[w] < bidx)) { best = s_val[w]; bidx = s_idx[w]; }
        ids_out[(int64_t)blockIdx.x * ids_stride] = bidx == 0x7fffffff ? 0 : bidx;
    }
}

// ---- bt_topk_sample ----------------------------------------------------------------------------------------------------
__device__ inline uint32_t float_key(float x) {  // order-preserving float -> uint
    const uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline uint64_t splitmix64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

constexpr int MAX_CAND = 1024;

template <typename T>
__global__ __launch_bounds__(256) void topk_sample_kernel(const T* __restrict__ logits, int64_t ld, int V_, float inv_temp,
                                                         int top_k, uint64_t seed, const int32_t* __restrict__ step_dev,
                                                         int64_t row0, const int64_t* __restrict__ row_map,
                                                         int64_t* __restrict__ ids_out, int64_t ids_stride) {
    __shared__ int hist[256];
    __shared__ uint32_t s_prefix, s_mask;
    __shared__ int s_k;
    __shared__ float c_val[MAX_CAND];
    __shared__ int c_idx[MAX_CAND];
    __shared__ float red[4];
    __shared__ int w_gt[4], w_eq[4];
    __shared__ float s_part[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* row = logits + (int64_t)blockIdx.x * ld;
    const uint64_t rnd = splitmix64(splitmix64(seed ^ (uint64_t)(step_dev ? *step_dev : 0) * 0x100000001b3ull) +
                                    (uint64_t)(row0 + (row_map ? row_map[blockIdx.x] : (int64_t)blockIdx.x)));
    const float u01 = (float)((rnd >> 40) & 0xffffff) * (1.0f / 16777216.0f);

    if (top_k > 0 && top_k < V_) {
        // radix select of the top_k-th largest key, 8 bits per pass
        if (tid == 0) { s_prefix = 0; s_mask = 0; s_k = top_k; }
        for (int pass = 3; pass >= 0; --pass) {
            hist[tid] = 0;
            __syncthreads();
            const uint32_t prefix = s_prefix, mask = s_mask;
            for (int c = tid; c < V_; c += 256) {
                const uint32_t key = float_key(Elem<T>::load(row + c) * inv_temp);
                if ((key & mask) == prefix) atomicAdd(&hist[(key >> (8 * pass)) & 255], 1);
            }
            __syncthreads();
            if (tid == 0) {
                int k = s_k, cum = 0, b = 255;
                for (; b > 0; --b) {
                    if (cum + hist[b] >= k) break;
                    cum += hist[b];
                }
                s_k = k - cum;
                s_prefix = prefix | ((uint32_t)b << (8 * pass));
                s_mask = mask | (0xffu << (8 * pass));
            }
            __syncthreads();
        }
        // HF TopKLogitsWarper keeps every score >= the k-th largest (ties included).  Candidates are compacted in
        // VOCABULARY ORDER (wave w owns a contiguous quarter of the row; ballot prefix inside the wave, per-wave bases from
        // a counting pass), so the kept set and the draw are deterministic for a given (seed, step, row).  At most
        // MAX_CAND candidates: every score strictly above the threshold (fewer than top_k <= MAX_CAND of them) and the
        // lowest-index ties at the threshold up to the capacity.
        const uint32_t thr = s_prefix;
        const int per_wave = (V_ + 3) / 4;
        const int c_lo = wave * per_wave, c_hi = min(V_, c_lo + per_wave);
        int n_gt = 0, n_eq = 0;
        for (int c0 = c_lo; c0 < c_hi; c0 += 64) {
            const int c = c0 + lane;
            const uint32_t key = c < c_hi ? float_key(Elem<T>::load(row + c) * inv_temp) : 0u;
            n_gt += __popcll(__ballot(c < c_hi && key > thr));
            n_eq += __popcll(__ballot(c < c_hi && key == thr));
        }
        if (lane == 0) { w_gt[wave] = n_gt; w_eq[wave] = n_eq; }
        __syncthreads();
        int tot_gt = 0, tot_eq = 0, gt_before = 0, eq_before = 0;
        for (int w = 0; w < 4; ++w) {
            if (w < wave) { gt_before += w_gt[w]; eq_before += w_eq[w]; }
            tot_gt += w_gt[w];
            tot_eq += w_eq[w];
        }
        const int eq_allow = MAX_CAND - tot_gt;   // >= 1: tot_gt < top_k <= MAX_CAND (checked on the host)
        const uint64_t below = (1ull << lane) - 1ull;
        for (int c0 = c_lo; c0 < c_hi; c0 += 64) {
            const int c = c0 + lane;
            const float v = c < c_hi ? Elem<T>::load(row + c) * inv_temp : 0.f;
            const uint32_t key = c < c_hi ? float_key(v) : 0u;
            const bool gt = c < c_hi && key > thr, eq = c < c_hi && key == thr;
            const uint64_t bg = __ballot(gt), be = __ballot(eq);
            const int my_gt = gt_before + __popcll(bg & below), my_eq = eq_before + __popcll(be & below);
            int slot = -1;
            if (gt) slot = my_gt + min(my_eq, eq_allow);
            else if (eq && my_eq < eq_allow) slot = my_gt + my_eq;
            if (slot >= 0) { c_val[slot] = v; c_idx[slot] = c; }
            gt_before += __popcll(bg);
            eq_before += __popcll(be);
        }
        __syncthreads();
        const int n = tot_gt + min(tot_eq, eq_allow);
        // softmax over the candidates (block-parallel) + inverse-CDF draw in vocabulary order
        float mx = -INFINITY;
        for (int i = tid; i < n; i += 256) mx = fmaxf(mx, c_val[i]);
        mx = wave_max(mx);
        if (lane == 0) red[wave] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        for (int i = tid; i < n; i += 256) c_val[i] = expf(c_val[i] - mx);
        __syncthreads();
        if (tid == 0) {
            float tot = 0.f;
            for (int i = 0; i < n; ++i) tot += c_val[i];
            const float target = u01 * tot;
            float cum = 0.f;
            int chosen = n > 0 ? c_idx[n - 1] : 0;
            for (int i = 0; i < n; ++i) {
                cum += c_val[i];
                if (cum > target) { chosen = c_idx[i]; break; }
            }
            ids_out[(int64_t)blockIdx.x * ids_stride] = chosen;
        }
        return;
    }
    // no top-k filter: full softmax, chunked inverse CDF
    float mx = -INFINITY;
    for (int c = tid; c < V_; c += 256) mx = fmaxf(mx, Elem<T>::load(row + c) * inv_temp);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const int chunk = (V_ + 255) / 256;
    const int c0 = tid * chunk, c1 = min(V_, c0 + chunk);
    float part = 0.f;
    for (int c = c0; c < c1; ++c) part += expf(Elem<T>::load(row + c) * inv_temp - mx);
    s_part[tid] = part;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.f;
        for (int i = 0; i < 256; ++i) tot += s_part[i];
        const float target = u01 * tot;
        float cum = 0.f;
        int t = 0;
        for (; t < 255; ++t) {
            if (cum + s_part[t] > target) break;
            cum += s_part[t];
        }
        const int a0 = t * chunk, a1 = min(V_, a0 + chunk);
        int chosen = a1 > a0 ? a1 - 1 : V_ - 1;
        for (int c = a0; c < a1; ++c) {
            cum += expf(Elem<T>::load(row + c) * inv_temp - mx);
            if (cum > target) { chosen = c; break; }
        }
        ids_out[(int64_t)blockIdx.x * ids_stride] = chosen;
    }
}

// ---- bt_finish_block ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void finish_block_kernel(int64_t B, int L, int count, int64_t* __restrict__ raw,
                                                           int64_t* __restrict__ out_ids, int64_t nb_max,
                                                           const int32_t* __restrict__ blk_dev,
                                                           int32_t* __restrict__ finished,
                                                           int32_t* __restrict__ finished_prev,
                                                           uint8_t* __restrict__ mask, int64_t mask_ld,
                                                           int64_t config_pad, int64_t gen_eos, int64_t gen_pad,
                                                           const int64_t* __restrict__ row_map, int32_t* __restrict__ err_flag) {
    __shared__ int s_stop;
    const int blk = *blk_dev;
    // device-side guard (a binder replaying the block step past the end of its buffers): flag it, write nothing
    if (blk < 0 || blk >= nb_max || (mask && blk >= mask_ld)) {
        if (threadIdx.x == 0 && err_flag) atomicOr(err_flag, BT_DEV_ERR_BLOCK_RANGE);
        return;
    }
    if (threadIdx.x == 0) s_stop = -1;
    __syncthreads();
    // HF greedy_search stops right after the first step at which every active row has produced EOS (Q6)
    if (gen_eos >= 0) {
        for (int64_t b = threadIdx.x; b < B; b += blockDim.x) {
            if (finished[b]) continue;
            int e = count;
            for (int t = 0; t < count; ++t)
                if (raw[b * L + t] == gen_eos) { e = t; break; }
            atomicMax(&s_stop, e);
        }
    }
    __syncthreads();
    const int ncols = (gen_eos < 0 || s_stop < 0) ? count : min(s_stop + 1, count);
    for (int64_t b = threadIdx.x; b < B; b += blockDim.x) {
        bool fin = finished[b] != 0;
        if (finished_prev) finished_prev[b] = fin ? 1 : 0;
        // row of out_ids: the sequence's ORIGINAL batch row once the engine has been compacted (< 0: padding row)
        const int64_t orow = row_map ? row_map[b] : b;
        int64_t* __restrict__ dst = out_ids + (orow * nb_max + blk) * L;
        if (fin) {
            for (int t = 0; t < L; ++t) {
                raw[b * L + t] = config_pad;
                if (orow >= 0) dst[t] = config_pad;
            }
        } else {
            bool seen = false, any_eos = false;
            for (int t = 0; t < L; ++t) {
                int64_t tok;
                if (t < ncols) {
                    tok = seen ? gen_pad : raw[b * L + t];
                    if (gen_eos >= 0 && tok == gen_eos) seen = true;
                } else {
                    tok = config_pad;
                }
                any_eos = any_eos || (gen_eos >= 0 && tok == gen_eos);
                raw[b * L + t] = tok;
                if (orow >= 0) dst[t] = tok;
            }
            fin = any_eos;  // unfinished *= next_tokens.ne(eos).all(-1)   (block_transformer.py:411-412)
            finished[b] = fin ? 1 : 0;
        }
        if (mask) mask[b * mask_ld + blk] = fin ? 0 : 1;
    }
}

__global__ void bump_kernel(int32_t* c, int32_t d) { *c += d; }

// bt_roll_over_enqueue's guard: the block-decoder step appends at global cache row *n_cur
__global__ void guard_ncur_kernel(int32_t* n_cur, int32_t n_max, int32_t* err_flag) {
    if (*n_cur >= n_max || *n_cur < 0) {
        atomicOr(err_flag, BT_DEV_ERR_CACHE_FULL);
        *n_cur = n_max - 1;
    }
}

// ---- bt_compact_enqueue ------------------------------------------------------------------------------------------------------
// K/V rows of sequence live[i] -> sequence slot i, in place.  grid.x = 4 KiB chunks of one (sequence, head) slab's first
// *n_cur positions (contiguous: [n_max][d] per head), grid.y = (layer, head, K|V).  A thread owns ONE 16-byte offset of the slab and
// walks i upward: slot i is written after slot live[i] >= i was read, and later sources live[i'] > live[i] >= i are never a slot
// this thread has written -- so U loads go out back to back before their stores (U x 16 B per thread in flight), no barrier.
constexpr int COMPACT_U = 8;
__global__ __launch_bounds__(256) void compact_kv_kernel(char* __restrict__ k, char* __restrict__ v, int64_t layer_stride_b,
                                                         int64_t seq_stride_b, int64_t head_stride_b, int heads, int row_bytes,
                                                         const int64_t* __restrict__ live, int n_live,
                                                         const int32_t* __restrict__ n_cur_dev, int n_max) {
    const int n = min(max(*n_cur_dev, 0), n_max);
    const int64_t span = (int64_t)n * row_bytes;
    const int64_t off = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (off >= span) return;
    const int y = blockIdx.y;
    const int lh = y >> 1;
    char* base = ((y & 1) ? v : k) + (int64_t)(lh / heads) * layer_stride_b + (int64_t)(lh % heads) * head_stride_b + off;
    int i = 0;
    while (i < n_live && live[i] == i) ++i;   // rows already in place (wave-uniform scalar loads)
    for (; i < n_live; i += COMPACT_U) {
        uint4 t[COMPACT_U];
#pragma unroll
        for (int u = 0; u < COMPACT_U; ++u)
            if (i + u < n_live) t[u] = *reinterpret_cast<const uint4*>(base + live[i + u] * seq_stride_b);
#pragma unroll
        for (int u = 0; u < COMPACT_U; ++u)
            if (i + u < n_live) *reinterpret_cast<uint4*>(base + (int64_t)(i + u) * seq_stride_b) = t[u];
    }
}

// the small per-sequence state of the compacted engine (its own buffers: a plain gather); one workgroup per dst row
__global__ __launch_bounds__(64) void compact_state_kernel(const int64_t* __restrict__ live, int n_live, int L,
                                                          const uint8_t* __restrict__ s_mask, int64_t s_mask_ld,
                                                          uint8_t* __restrict__ d_mask, int64_t d_mask_ld,
                                                          const int64_t* __restrict__ s_raw, int64_t* __restrict__ d_raw,
                                                          int32_t* __restrict__ d_fin, int32_t* __restrict__ d_fin_prev,
                                                          const int64_t* __restrict__ s_map, int64_t* __restrict__ d_map,
                                                          const int32_t* __restrict__ s_ncur, int32_t* __restrict__ d_ncur) {
    const int i = blockIdx.x, tid = threadIdx.x;
    const bool real = i < n_live;
    const int64_t src = real ? live[i] : 0;
    if (d_mask) {
        const int64_t w = min(s_mask_ld, d_mask_ld);
        for (int64_t c = tid; c < d_mask_ld; c += 64)
            d_mask[i * d_mask_ld + c] = (real && s_mask && c < w) ? s_mask[src * s_mask_ld + c] : (uint8_t)1;
    }
    for (int t = tid; t < L; t += 64) d_raw[(int64_t)i * L + t] = real ? s_raw[src * L + t] : 0;
    if (tid == 0) {
        d_fin[i] = real ? 0 : 1;
        d_fin_prev[i] = real ? 0 : 1;
        d_map[i] = real ? (s_map ? s_map[src] : src) : -1;
        if (i == 0) *d_ncur = *s_ncur;
    }
}

#ifdef BT_DEBUG_HOOKS
// debug probe of ds_read_b64_tr_b16: lds[i] = in[i] (4096 shorts); lane l reads at element offset addr[l]; out[l*4 + j]
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__global__ void tr_probe_kernel(const short* in, const int* addr, short* out) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
    __syncthreads();
    const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4_t*)(lds + addr[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
#endif  // BT_DEBUG_HOOKS

template <typename T>
static int launch_embed(int64_t rows, int64_t width, const int64_t* ids, int64_t ids_stride, int64_t vocab,
                        const void* table, void* out,
                        int64_t ldo, const int32_t* row_zero, int rows_per_seq, hipStream_t s) {
    constexpr int V = Elem<T>::VEC;
    const bool vec = (width % V == 0) && (ldo % V == 0) && aligned16(table) && aligned16(out);
    const unsigned grid = (unsigned)ceil_div(rows, 4);
    ProfScope ps(BT_PROF_EMBED, s);
    if (vec)
        hipLaunchKernelGGL((embed_rows_kernel<T, true>), dim3(grid), dim3(256), 0, s, ids, ids_stride, vocab, (const T*)table,
                           (T*)out, ldo, rows, (int)width, row_zero, rows_per_seq);
    else
        hipLaunchKernelGGL((embed_rows_kernel<T, false>), dim3(grid), dim3(256), 0, s, ids, ids_stride, vocab, (const T*)table,
                           (T*)out, ldo, rows, (int)width, row_zero, rows_per_seq);
    BT_LAUNCH_CHECK("bt_embed_rows launch");
    return BT_OK;
}

template <typename T>
static int launch_argmax(int64_t M, int64_t V_, const void* logits, int64_t ld, int64_t* ids_out, int64_t ids_stride,
                         hipStream_t s) {
    constexpr int V = Elem<T>::VEC;
    const bool vec = (V_ % V == 0) && (ld % V == 0) && aligned16(logits);
    ProfScope ps(BT_PROF_ARGMAX, s);
    const bool wide = M <= 256;   // few rows: one row's scan is the critical path -> 1024 threads per row
#define BT_LAUNCH_ARGMAX(VEC_, NT_)                                                                                     \
    hipLaunchKernelGGL((argmax_kernel<T, VEC_, NT_>), dim3((unsigned)M), dim3(NT_), 0, s, (const T*)logits, ld, (int)V_, ids_out, \
                       ids_stride)
    if (vec) { if (wide) BT_LAUNCH_ARGMAX(true, 1024); else BT_LAUNCH_ARGMAX(true, 256); }
    else { if (wide) BT_LAUNCH_ARGMAX(false, 1024); else BT_LAUNCH_ARGMAX(false, 256); }
#undef BT_LAUNCH_ARGMAX
    BT_LAUNCH_CHECK("bt_argmax launch");
    return BT_OK;
}

}  // namespace bt

using namespace bt;

extern "C" int bt_embed_rows(int dtype, int64_t rows, int64_t width, const int64_t* ids, int64_t ids_stride,
                             int64_t vocab, const void* table, void* out, int64_t ldo, const int32_t* row_zero,
                             int32_t rows_per_seq, bt_stream stream) {
    BT_CHECK_ARG(dtype == BT_F32 || dtype == BT_BF16, "bt_embed_rows: bad dtype %d", dtype);
    BT_CHECK_ARG(rows >= 0 && width > 0 && width < (1ll << 31) && vocab > 0, "bt_embed_rows: bad shape");
    BT_CHECK_ARG(ids && table && out && ldo >= width, "bt_embed_rows: null pointer or short row stride");
    BT_CHECK_ARG(!row_zero || rows_per_seq > 0, "bt_embed_rows: rows_per_seq must be > 0 with row_zero");
    if (rows == 0) return BT_OK;
    if (dtype == BT_BF16)
        return launch_embed<bf16>(rows, width, ids, ids_stride, vocab, table, out, ldo, row_zero, rows_per_seq, (hipStream_t)stream);
    return launch_embed<float>(rows, width, ids, ids_stride, vocab, table, out, ldo, row_zero, rows_per_seq, (hipStream_t)stream);
}

extern "C" int bt_argmax(int dtype, int64_t M, int64_t V, const void* logits, int64_t ld, int64_t* ids_out,
                         int64_t ids_stride, bt_stream stream) {
    BT_CHECK_ARG(dtype == BT_F32 || dtype == BT_BF16, "bt_argmax: bad dtype %d", dtype);
    BT_CHECK_ARG(M >= 0 && V > 0 && V < (1ll << 31) && ld >= V && M < (1ll << 31), "bt_argmax: bad shape");
    BT_CHECK_ARG(logits && ids_out, "bt_argmax: null pointer");
    if (M == 0) return BT_OK;
    if (dtype == BT_BF16) return launch_argmax<bf16>(M, V, logits, ld, ids_out, ids_stride, (hipStream_t)stream);
    return launch_argmax<float>(M, V, logits, ld, ids_out, ids_stride, (hipStream_t)stream);
}

int bt::topk_sample_impl(int dtype, int64_t M, int64_t V, const void* logits, int64_t ld, float temperature, int32_t top_k,
                         uint64_t seed, const int32_t* step_dev, int64_t row0, const int64_t* row_map, int64_t* ids_out,
                         int64_t ids_stride, bt_stream stream) {
    BT_CHECK_ARG(dtype == BT_F32 || dtype == BT_BF16, "bt_topk_sample: bad dtype %d", dtype);
    BT_CHECK_ARG(M >= 0 && V > 0 && V < (1ll << 31) && ld >= V && M < (1ll << 31), "bt_topk_sample: bad shape");
    BT_CHECK_ARG(logits && ids_out, "bt_topk_sample: null pointer");
    BT_CHECK_ARG(temperature > 0.f, "bt_topk_sample: temperature must be > 0");
    if (top_k > MAX_CAND && top_k < V) {   // the candidate buffer holds MAX_CAND scores: refuse rather than drop candidates
        set_error("bt_topk_sample: top_k = %d exceeds the %d-candidate capacity (use top_k <= %d, or 0 / >= vocab for no filter)",
                  top_k, MAX_CAND, MAX_CAND);
        return BT_ERR_UNSUPPORTED;
    }
    if (M == 0) return BT_OK;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(BT_PROF_SAMPLE, s);
    if (dtype == BT_BF16)
        hipLaunchKernelGGL(topk_sample_kernel<bf16>, dim3((unsigned)M), dim3(256), 0, s, (const bf16*)logits, ld, (int)V,
                           1.0f / temperature, top_k, seed, step_dev, row0, row_map, ids_out, ids_stride);
    else
        hipLaunchKernelGGL(topk_sample_kernel<float>, dim3((unsigned)M), dim3(256), 0, s, (const float*)logits, ld,
                           (int)V, 1.0f / temperature, top_k, seed, step_dev, row0, row_map, ids_out, ids_stride);
    BT_LAUNCH_CHECK("bt_topk_sample launch");
    return BT_OK;
}

extern "C" int bt_topk_sample(int dtype, int64_t M, int64_t V, const void* logits, int64_t ld, float temperature,
                              int32_t top_k, uint64_t seed, const int32_t* step_dev, int64_t row0, int64_t* ids_out,
                              int64_t ids_stride, bt_stream stream) {
    return topk_sample_impl(dtype, M, V, logits, ld, temperature, top_k, seed, step_dev, row0, nullptr, ids_out, ids_stride, stream);
}

int bt::finish_block_impl(int64_t B, int32_t L, int32_t count, int64_t* raw, int64_t* out_ids, int64_t nb_max,
                          const int32_t* blk_dev, int32_t* finished, int32_t* finished_prev, uint8_t* mask, int64_t mask_ld,
                          int64_t config_pad, int64_t gen_eos, int64_t gen_pad, const int64_t* row_map, int32_t* err_flag,
                          bt_stream stream) {
    BT_CHECK_ARG(B >= 0 && L > 0 && count >= 0 && count <= L, "bt_finish_block: bad shape B=%lld L=%d count=%d",
                 (long long)B, L, count);
    BT_CHECK_ARG(raw && out_ids && blk_dev && finished, "bt_finish_block: null pointer");
    if (B == 0) return BT_OK;
    ProfScope ps(BT_PROF_FINISH, (hipStream_t)stream);
    hipLaunchKernelGGL(finish_block_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, B, L, count, raw, out_ids,
                       nb_max, blk_dev, finished, finished_prev, mask, mask_ld, config_pad, gen_eos, gen_pad, row_map, err_flag);
    BT_LAUNCH_CHECK("bt_finish_block launch");
    return BT_OK;
}

extern "C" int bt_finish_block(int64_t B, int32_t L, int32_t count, int64_t* raw, int64_t* out_ids, int64_t nb_max,
                               const int32_t* blk_dev, int32_t* finished, int32_t* finished_prev, uint8_t* mask,
                               int64_t mask_ld, int64_t config_pad, int64_t gen_eos, int64_t gen_pad,
                               bt_stream stream) {
    return finish_block_impl(B, L, count, raw, out_ids, nb_max, blk_dev, finished, finished_prev, mask, mask_ld, config_pad,
                             gen_eos, gen_pad, nullptr, nullptr, stream);
}

int bt::guard_ncur(int32_t* n_cur, int64_t n_max, int32_t* err_flag, bt_stream stream) {
    hipLaunchKernelGGL(guard_ncur_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, n_cur, (int32_t)n_max, err_flag);
    BT_LAUNCH_CHECK("bt_roll_over_enqueue guard launch");
    return BT_OK;
}

extern "C" int bt_compact_enqueue(const bt_engine_desc* src, const bt_engine_desc* dst, const int64_t* live_rows, int64_t n_live,
                                  bt_stream stream) {
    BT_CHECK_ARG(src && dst && live_rows, "bt_compact_enqueue: null pointer");
    BT_CHECK_ARG(src->struct_size == sizeof(bt_engine_desc) && dst->struct_size == sizeof(bt_engine_desc),
                 "bt_compact_enqueue: bt_engine_desc.struct_size %u / %u != %zu -- the caller was built against another bt_hip.h", src->struct_size,
                 dst->struct_size, sizeof(bt_engine_desc));
    BT_CHECK_ARG(n_live >= 0 && n_live <= dst->B && dst->B <= src->B && n_live < (1ll << 31), "bt_compact_enqueue: bad row counts "
                 "(n_live %lld, dst B %lld, src B %lld)", (long long)n_live, (long long)dst->B, (long long)src->B);
    BT_CHECK_ARG(src->gk && src->gv && src->gk == dst->gk && src->gv == dst->gv && src->g_layer_stride == dst->g_layer_stride &&
                     src->g_nmax == dst->g_nmax && src->dtype == dst->dtype && src->L == dst->L,
                 "bt_compact_enqueue: dst must describe the same global KV storage as src");
    BT_CHECK_ARG(src->n_cur && src->raw && dst->n_cur && dst->raw && dst->finished && dst->finished_prev && dst->row_map,
                 "bt_compact_enqueue: null state buffer (n_cur / raw / finished / finished_prev / row_map)");
    BT_CHECK_ARG(dst->raw != src->raw && dst->finished != src->finished && dst->n_cur != src->n_cur &&
                     (!dst->mask || dst->mask != src->mask) && dst->row_map != src->row_map,
                 "bt_compact_enqueue: dst's state buffers must be distinct from src's (only the KV storage is compacted in place)");
    if (dst->B == 0) return BT_OK;
    hipStream_t s = (hipStream_t)stream;
    const bt_decoder_dims& d = src->bd_dims;
    const int64_t es = (int64_t)dtype_size(src->dtype);
    const int64_t row_bytes = (int64_t)d.head_dim * es;
    BT_CHECK_ARG(row_bytes % 16 == 0 && aligned16(src->gk) && aligned16(src->gv), "bt_compact_enqueue: K/V rows must be 16-byte multiples");
    if (n_live > 0) {
        const int64_t head_stride_b = src->g_nmax * row_bytes, seq_stride_b = (int64_t)d.heads * head_stride_b;
        const unsigned gx = (unsigned)ceil_div(head_stride_b, 4096);
        ProfScope ps(BT_PROF_EMBED, s);
        hipLaunchKernelGGL(compact_kv_kernel, dim3(gx, (unsigned)(d.layers * d.heads * 2)), dim3(256), 0, s, (char*)src->gk,
                           (char*)src->gv, src->g_layer_stride * es, seq_stride_b, head_stride_b, d.heads, (int)row_bytes, live_rows,
                           (int)n_live, src->n_cur, (int)src->g_nmax);
        BT_LAUNCH_CHECK("bt_compact_enqueue KV launch");
    }
    hipLaunchKernelGGL(compact_state_kernel, dim3((unsigned)dst->B), dim3(64), 0, s, live_rows, (int)n_live, (int)src->L, src->mask,
                       src->mask_ld, dst->mask, dst->mask_ld, src->raw, dst->raw, dst->finished, dst->finished_prev, src->row_map,
                       const_cast<int64_t*>(dst->row_map), src->n_cur, dst->n_cur);
    BT_LAUNCH_CHECK("bt_compact_enqueue state launch");
    return BT_OK;
}

extern "C" int bt_bump(int32_t* counter, int32_t delta, bt_stream stream) {
    BT_CHECK_ARG(counter, "bt_bump: null pointer");
    hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter, delta);
    BT_LAUNCH_CHECK("bt_bump launch");
    return BT_OK;
}

// ---- empirical ceilings of the box (bench.py `ceilings`, SURVEY.md section 8d: "measure empirical ceilings on the box ... and report
// against both"; the reference's harness measures its run's own numbers in-run, measure_generation_time.py:143-161) -------------------
// bt_ceiling_copy: dst[0 .. bytes) = src, 16 bytes per lane, four independent loads in flight per lane and iteration, grid-stride;
// flags bit 0 = non-temporal loads and stores.  HBM rate = 2 * bytes / time (bytes read + bytes written).
template <bool NT>
__global__ __launch_bounds__(256) void ceiling_copy_kernel(const u32x4_t* __restrict__ src, u32x4_t* __restrict__ dst, int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * 1024;
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += stride) {
        u32x4_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = i + u * 256;
            const int64_t jc = j < n16 ? j : n16 - 1;   // unconditional (clamped) loads: all four go out before the first store
            v[u] = NT ? __builtin_nontemporal_load(src + jc) : src[jc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = i + u * 256;
            if (j < n16) {
                if (NT) __builtin_nontemporal_store(v[u], dst + j);
                else dst[j] = v[u];
            }
        }
    }
}

extern "C" int bt_ceiling_copy(void* dst, const void* src, int64_t bytes, int32_t flags, bt_stream stream) {
    BT_CHECK_ARG(dst && src && bytes > 0 && bytes % 16 == 0 && aligned16(dst) && aligned16(src), "bt_ceiling_copy: 16-byte multiples only");
    const int64_t n16 = bytes / 16;
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(n16, 1024), 256 * 32);
    if (flags & 1)
        hipLaunchKernelGGL(ceiling_copy_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const u32x4_t*)src, (u32x4_t*)dst, n16);
    else
        hipLaunchKernelGGL(ceiling_copy_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const u32x4_t*)src, (u32x4_t*)dst, n16);
    BT_LAUNCH_CHECK("bt_ceiling_copy launch");
    return BT_OK;
}

// bt_ceiling_mfma: a bare v_mfma_f32_16x16x32_bf16 loop on the caller's (random) bf16 data, operands in registers, 16 independent
// accumulators per wave (no dependent-issue stalls), four operand pairings rotated through the unrolled body so consecutive MFMAs
// toggle their inputs like a GEMM's do.  No LDS, no memory traffic inside the loop: what it reaches is the matrix pipe's rate at the
// clock the part SUSTAINS under that load.  Each wave issues iters * 64 MFMAs of 16 * 16 * 32 * 2 flop; grid = workgroups of 4 waves.
typedef short ceil_bf16x8_t __attribute__((ext_vector_type(8)));
typedef float ceil_f32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void ceiling_mfma_kernel(const u32x4_t* __restrict__ data, float* __restrict__ out, int iters) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    ceil_bf16x8_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(ceil_bf16x8_t, data[gid * 8 + i]);
        b[i] = __builtin_bit_cast(ceil_bf16x8_t, data[gid * 8 + 4 + i]);
    }
    ceil_f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = ceil_f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    // (inline asm: with the builtin this compiler permutes the 16 accumulators through ~80 v_accvgpr moves per trip)
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(a[(i + r) & 3]), "v"(b[(j + r) & 3]));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results before the VALU reads below (inline asm is not hazard-checked)
    ceil_f32x4_t sum = ceil_f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) sum += acc[i][j];
    out[gid] = sum[0] + sum[1] + sum[2] + sum[3];
}

extern "C" int64_t bt_ceiling_mfma_flops(int32_t workgroups, int32_t iters) {
    return (int64_t)workgroups * 4 * (int64_t)iters * 64 * (16 * 16 * 32 * 2);
}

extern "C" int bt_ceiling_mfma(const void* data, int64_t data_bytes, void* out, int64_t out_bytes, int32_t workgroups, int32_t iters,
                               bt_stream stream) {
    BT_CHECK_ARG(data && out && aligned16(data) && workgroups > 0 && iters > 0 && data_bytes >= (int64_t)workgroups * 256 * 128 &&
                     out_bytes >= (int64_t)workgroups * 256 * 4,
                 "bt_ceiling_mfma: data (workgroups * 256 * 128 bytes, 16-byte aligned) and out (workgroups * 256 floats) required");
    hipLaunchKernelGGL(ceiling_mfma_kernel, dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream, (const u32x4_t*)data,
                       (float*)out, iters);
    BT_LAUNCH_CHECK("bt_ceiling_mfma launch");
    return BT_OK;
}

#ifdef BT_DEBUG_HOOKS
// ---- round 6 probe: what would a grid-wide barrier cost a persistent <= 4-row decoder LAYER? -------------------------
// One launch of `workgroups` resident workgroups x 256 threads runs `rounds` barrier rounds; between two barriers every workgroup
// streams `bytes_per_wg` bytes of `stream_src` (16 B per lane, non-temporal; 0 = nothing) and publishes one 128-byte record that the
// next round re-reads -- the shape of "GEMV phase -> hand its rows to every workgroup -> next phase".  mode 0: ONE device-scope counter
// (monotonic, relaxed sc1 polls + s_sleep, release fence before the arrive, acquire fence after); mode 1: XCD-hierarchical -- a counter
// per XCD (workgroup i sits on XCD i % 8 in a one-workgroup-per-CU launch), the last arriver of an XCD arrives at the top counter, the
// last of those bumps the top generation, every XCD leader then bumps its XCD's generation and the other workgroups poll that word
// (their own L2).  state: >= 64 uint32 words per XCD slot + 64 for the top, zeroed by the caller; ticks = s_memrealtime (100 MHz) of
// workgroup 0 around the loop.
__global__ __launch_bounds__(256) void grid_barrier_probe_kernel(uint32_t* state, uint32_t* records, const uint4* stream_src, int64_t bytes_per_wg,
                                                                 int rounds, int mode, uint64_t* ticks, float* sink) {
    const int G = gridDim.x, bid = blockIdx.x, tid = threadIdx.x;
    const int xcd = bid & 7, n_x = (G >> 3) + ((G & 7) > xcd ? 1 : 0);
    uint32_t* top_cnt = state;              // [0] counter, [32] generation
    uint32_t* x_cnt = state + 64 * (1 + xcd);
    float acc = 0.f;
    uint64_t t0 = 0;
    if (bid == 0 && tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < rounds; ++r) {
        // phase work: stream this workgroup's slice, re-read the previous round's record of the "next" workgroup, publish its own
        const int64_t n16 = bytes_per_wg / 16;
        const uint4* src = stream_src + (int64_t)bid * n16;
        for (int64_t i = tid; i < n16; i += 256) {
            typedef unsigned int pu32x4_t __attribute__((ext_vector_type(4)));
            const pu32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const pu32x4_t*>(src + i));
            acc += __uint_as_float(v.x ^ v.y ^ v.z ^ v.w);
        }
        if (tid < 32) {
            const uint32_t seen = __hip_atomic_load(records + (int64_t)((bid + 1) % G) * 32 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc += (float)seen;
            __hip_atomic_store(records + (int64_t)bid * 32 + tid, (uint32_t)(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const uint32_t gen = (uint32_t)(r + 1);
            if (mode == 0) {
                __hip_atomic_fetch_add(top_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(top_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen * (uint32_t)G) __builtin_amdgcn_s_sleep(2);
            } else {
                const uint32_t old = __hip_atomic_fetch_add(x_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == gen * (uint32_t)n_x - 1u) {             // last arriver of this XCD: arrive at the top, wait for its generation
                    const uint32_t told = __hip_atomic_fetch_add(top_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (told == gen * 8u - 1u) __hip_atomic_store(top_cnt + 32, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    while (__hip_atomic_load(top_cnt + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) __builtin_amdgcn_s_sleep(1);
                    __hip_atomic_store(x_cnt + 32, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    while (__hip_atomic_load(x_cnt + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) __builtin_amdgcn_s_sleep(1);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    if (bid == 0 && tid == 0) {
        ticks[0] = __builtin_amdgcn_s_memrealtime() - t0;
    }
    if (acc == 12345.678f) sink[bid] = acc;
}

extern "C" int bt_debug_grid_barrier_probe(void* state, void* records, const void* stream_src, int64_t bytes_per_wg, int32_t workgroups,
                                           int32_t rounds, int32_t mode, void* ticks, void* sink, bt_stream stream) {
    BT_CHECK_ARG(state && records && ticks && sink && workgroups >= 8 && rounds > 0 && bytes_per_wg % 16 == 0, "bt_debug_grid_barrier_probe: bad argument");
    hipLaunchKernelGGL(grid_barrier_probe_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, (uint32_t*)state, (uint32_t*)records,
                       (const uint4*)stream_src, bytes_per_wg, rounds, mode, (uint64_t*)ticks, (float*)sink);
    BT_LAUNCH_CHECK("grid barrier probe launch");
    return BT_OK;
}
// the launch-per-phase counterpart: one phase (the same streaming + record work, no barrier) per launch; the caller captures a chain of them
__global__ __launch_bounds__(256) void phase_probe_kernel(uint32_t* records, const uint4* stream_src, int64_t bytes_per_wg, int r, float* sink) {
    const int G = gridDim.x, bid = blockIdx.x, tid = threadIdx.x;
    float acc = 0.f;
    const int64_t n16 = bytes_per_wg / 16;
    const uint4* src = stream_src + (int64_t)bid * n16;
    for (int64_t i = tid; i < n16; i += 256) {
        typedef unsigned int pu32x4_t __attribute__((ext_vector_type(4)));
        const pu32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const pu32x4_t*>(src + i));
        acc += __uint_as_float(v.x ^ v.y ^ v.z ^ v.w);
    }
    if (tid < 32) {
        acc += (float)records[(int64_t)((bid + 1) % G) * 32 + tid];
        records[(int64_t)bid * 32 + tid] = (uint32_t)(r + 1);
    }
    if (acc == 12345.678f) sink[bid] = acc;
}
extern "C" int bt_debug_phase_probe(void* records, const void* stream_src, int64_t bytes_per_wg, int32_t workgroups, int32_t r, void* sink,
                                    bt_stream stream) {
    hipLaunchKernelGGL(phase_probe_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, (uint32_t*)records, (const uint4*)stream_src,
                       bytes_per_wg, r, (float*)sink);
    BT_LAUNCH_CHECK("phase probe launch");
    return BT_OK;
}

// round 6 probe (profiles/r6_handoff_false_positive.md): a bare hipMemsetAsync of `bytes` at `p`, to be captured into a hipGraph by the caller
extern "C" int bt_debug_memset(void* p, int32_t value, int64_t bytes, bt_stream stream) {
    hipError_t e = hipMemsetAsync(p, value, (size_t)bytes, (hipStream_t)stream);
    if (e != hipSuccess) return bt::hip_fail(e, "hipMemsetAsync(probe)");
    return BT_OK;
}

extern "C" int bt_debug_tr_probe(const void* in, const void* addr, void* out, bt_stream stream) {
    hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const short*)in, (const int*)addr,
                       (short*)out);
    BT_LAUNCH_CHECK("tr_probe launch");
    return BT_OK;
}
#endif
