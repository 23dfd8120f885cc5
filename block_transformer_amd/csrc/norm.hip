// bt_layernorm: LayerNorm of a row with one or two affine sets.
//
// GPT-NeoX's parallel residual feeds LN1(x) to attention and LN2(x) to the MLP *of the same x*: one pass computes the
// row statistics once and writes both normalised rows (C2 of SURVEY.md 2.2).  One wave64 per row, 4 rows per
// workgroup; 16-byte vector loads; statistics in fp32 with the two-pass (mean, then centred variance) form that
// torch.nn.functional.layer_norm uses.  HBM-bound: reads M*H, writes (1 or 2)*M*H elements; the row (<= 16 KiB) is
// re-read from L1/L2 for the second and third pass.
#include "bt_common.h"

namespace bt {

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ w1,
                                                       const T* __restrict__ b1, T* __restrict__ o1, int64_t ldo1,
                                                       const T* __restrict__ w2, const T* __restrict__ b2,
                                                       T* __restrict__ o2, int64_t ldo2, int64_t M, int H, float eps) {
    constexpr int V = Elem<T>::VEC;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const T* xr = x + row * ldx;
    float sum = 0.f;
    if (VEC) {
        for (int c = lane * V; c < H; c += 64 * V) {
            float v[V];
            load_vec<T>(xr + c, v);
#pragma unroll
            for (int i = 0; i < V; ++i) sum += v[i];
        }
    } else {
        for (int c = lane; c < H; c += 64) sum += Elem<T>::load(xr + c);
    }
    const float mean = wave_sum(sum) / (float)H;
    float sq = 0.f;
    if (VEC) {
        for (int c = lane * V; c < H; c += 64 * V) {
            float v[V];
            load_vec<T>(xr + c, v);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float d = v[i] - mean;
                sq = fmaf(d, d, sq);
            }
        }
    } else {
        for (int c = lane; c < H; c += 64) {
            const float d = Elem<T>::load(xr + c) - mean;
            sq = fmaf(d, d, sq);
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)H + eps);
    T* o1r = o1 + row * ldo1;
    T* o2r = o2 ? o2 + row * ldo2 : nullptr;
    if (VEC) {
        for (int c = lane * V; c < H; c += 64 * V) {
            float v[V], g[V], b[V], r[V];
            load_vec<T>(xr + c, v);
            load_vec<T>(w1 + c, g);
            load_vec<T>(b1 + c, b);
#pragma unroll
            for (int i = 0; i < V; ++i) r[i] = fmaf((v[i] - mean) * rstd, g[i], b[i]);
            store_vec<T>(o1r + c, r);
            if (o2r) {
                load_vec<T>(w2 + c, g);
                load_vec<T>(b2 + c, b);
#pragma unroll
                for (int i = 0; i < V; ++i) r[i] = fmaf((v[i] - mean) * rstd, g[i], b[i]);
                store_vec<T>(o2r + c, r);
            }
        }
    } else {
        for (int c = lane; c < H; c += 64) {
            const float n = (Elem<T>::load(xr + c) - mean) * rstd;
            Elem<T>::store(o1r + c, fmaf(n, Elem<T>::load(w1 + c), Elem<T>::load(b1 + c)));
            if (o2r) Elem<T>::store(o2r + c, fmaf(n, Elem<T>::load(w2 + c), Elem<T>::load(b2 + c)));
        }
    }
}

// Register-resident variant for H <= NV * 64 * VEC (vector-aligned rows): a lane loads its NV 16-byte pieces of the row ONCE,
// all in flight together; the statistics and both outputs come from registers.  The streaming kernel above pays three
// dependent load latencies per row (HBM, then two L1/L2 re-reads) with one 1 KiB wave-load in flight at a time; here the
// row costs one latency.  Same per-lane summation order and the same two-pass statistics -> bit-identical results.
template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm_reg_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ w1,
                                                           const T* __restrict__ b1, T* __restrict__ o1, int64_t ldo1,
                                                           const T* __restrict__ w2, const T* __restrict__ b2,
                                                           T* __restrict__ o2, int64_t ldo2, int64_t M, int H, float eps) {
    constexpr int V = Elem<T>::VEC;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const T* xr = x + row * ldx;
    float v[NV][V];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * V;
        if (c < H) load_vec<T>(xr + c, v[i]);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if ((i * 64 + lane) * V < H) {
#pragma unroll
            for (int j = 0; j < V; ++j) sum += v[i][j];
        }
    const float mean = wave_sum(sum) / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if ((i * 64 + lane) * V < H) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float d = v[i][j] - mean;
                sq = fmaf(d, d, sq);
            }
        }
    const float rstd = rsqrtf(wave_sum(sq) / (float)H + eps);
    T* o1r = o1 + row * ldo1;
    T* o2r = o2 ? o2 + row * ldo2 : nullptr;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * V;
        if (c < H) {
            float g[V], b[V], r[V];
            load_vec<T>(w1 + c, g);
            load_vec<T>(b1 + c, b);
#pragma unroll
            for (int j = 0; j < V; ++j) r[j] = fmaf((v[i][j] - mean) * rstd, g[j], b[j]);
            store_vec<T>(o1r + c, r);
            if (o2r) {
                load_vec<T>(w2 + c, g);
                load_vec<T>(b2 + c, b);
#pragma unroll
                for (int j = 0; j < V; ++j) r[j] = fmaf((v[i][j] - mean) * rstd, g[j], b[j]);
                store_vec<T>(o2r + c, r);
            }
        }
    }
}

#ifdef BT_DEBUG_HOOKS   // the folded-LayerNorm experiment (gemm.hip, profiles/r5_ab_ln_fused.jsonl): debug library only
// Row statistics only (round 5: LayerNorm folded into the consuming GEMMs, gemm.hip EpiArgs.ln_*): stats[row] = (mean, rstd) with
// exactly layernorm_reg_kernel's arithmetic (same per-lane summation order, two-pass variance) -- reads M*H elements, writes 8 M
// bytes instead of 2 M H elements.
template <typename T, int NV>
__global__ __launch_bounds__(256) void rowstats_reg_kernel(const T* __restrict__ x, int64_t ldx, float2* __restrict__ stats, int64_t M,
                                                          int H, float eps) {
    constexpr int V = Elem<T>::VEC;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const T* xr = x + row * ldx;
    float v[NV][V];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * V;
        if (c < H) load_vec<T>(xr + c, v[i]);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if ((i * 64 + lane) * V < H) {
#pragma unroll
            for (int j = 0; j < V; ++j) sum += v[i][j];
        }
    const float mean = wave_sum(sum) / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if ((i * 64 + lane) * V < H) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float d = v[i][j] - mean;
                sq = fmaf(d, d, sq);
            }
        }
    const float rstd = rsqrtf(wave_sum(sq) / (float)H + eps);
    if (lane == 0) stats[row] = float2{mean, rstd};
}

int rowstats_impl(int dtype, int64_t M, int64_t H, const void* x, int64_t ldx, float eps, void* stats, bt_stream stream) {
    BT_CHECK_ARG(dtype == BT_BF16, "bt_debug_rowstats: bf16 only");
    BT_CHECK_ARG(M >= 0 && H > 0 && H % 8 == 0 && H <= 8 * 64 * 8 && ldx >= H && ldx % 8 == 0 && x && stats && aligned16(x) &&
                     (reinterpret_cast<uintptr_t>(stats) & 7u) == 0,
                 "bt_debug_rowstats: rows of <= 4096 bf16 elements in 16-byte units, 8-byte aligned stats");
    if (M == 0) return BT_OK;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(BT_PROF_LAYERNORM, s);
    const unsigned grid = (unsigned)ceil_div(M, 4);
    const int nv = (int)ceil_div(H, (int64_t)64 * 8);
#define BT_RS(NV_) hipLaunchKernelGGL((rowstats_reg_kernel<bf16, NV_>), dim3(grid), dim3(256), 0, s, (const bf16*)x, ldx, (float2*)stats, M, (int)H, eps)
    if (nv <= 1) BT_RS(1);
    else if (nv <= 2) BT_RS(2);
    else if (nv <= 4) BT_RS(4);
    else BT_RS(8);
#undef BT_RS
    BT_LAUNCH_CHECK("bt_debug_rowstats launch");
    return BT_OK;
}

#endif  // BT_DEBUG_HOOKS (rowstats)

#ifdef BT_DEBUG_HOOKS
static bool g_ln_streaming = false;   // A/B (debug library only): force the three-pass streaming kernel
#else
static constexpr bool g_ln_streaming = false;
#endif

template <typename T>
static int launch_ln(int64_t M, int64_t H, const void* x, int64_t ldx, const void* w1, const void* b1, void* o1,
                     int64_t ldo1, const void* w2, const void* b2, void* o2, int64_t ldo2, float eps, hipStream_t s) {
    constexpr int V = Elem<T>::VEC;
    bool vec = (H % V == 0) && (ldx % V == 0) && (ldo1 % V == 0) && aligned16(x) && aligned16(w1) && aligned16(b1) &&
               aligned16(o1);
    if (o2) vec = vec && (ldo2 % V == 0) && aligned16(w2) && aligned16(b2) && aligned16(o2);
    const unsigned grid = (unsigned)ceil_div(M, 4);
    ProfScope ps(BT_PROF_LAYERNORM, s);
    const int nv = (int)ceil_div(H, (int64_t)64 * V);
#define BT_LN_REG(NV_)                                                                                                  \
    hipLaunchKernelGGL((layernorm_reg_kernel<T, NV_>), dim3(grid), dim3(256), 0, s, (const T*)x, ldx, (const T*)w1,     \
                       (const T*)b1, (T*)o1, ldo1, (const T*)w2, (const T*)b2, (T*)o2, ldo2, M, (int)H, eps)
    if (vec && nv <= 8 && !g_ln_streaming) {
        if (nv <= 1) BT_LN_REG(1);
        else if (nv <= 2) BT_LN_REG(2);
        else if (nv <= 4) BT_LN_REG(4);
        else BT_LN_REG(8);
    } else if (vec)
        hipLaunchKernelGGL((layernorm_kernel<T, true>), dim3(grid), dim3(256), 0, s, (const T*)x, ldx, (const T*)w1,
                           (const T*)b1, (T*)o1, ldo1, (const T*)w2, (const T*)b2, (T*)o2, ldo2, M, (int)H, eps);
    else
        hipLaunchKernelGGL((layernorm_kernel<T, false>), dim3(grid), dim3(256), 0, s, (const T*)x, ldx, (const T*)w1,
                           (const T*)b1, (T*)o1, ldo1, (const T*)w2, (const T*)b2, (T*)o2, ldo2, M, (int)H, eps);
#undef BT_LN_REG
    BT_LAUNCH_CHECK("bt_layernorm launch");
    return BT_OK;
}

}  // namespace bt

using namespace bt;

extern "C" int bt_layernorm(int dtype, int64_t M, int64_t H, const void* x, int64_t ldx, const void* w1, const void* b1,
                            void* out1, int64_t ldo1, const void* w2, const void* b2, void* out2, int64_t ldo2,
                            float eps, bt_stream stream) {
    BT_CHECK_ARG(dtype == BT_F32 || dtype == BT_BF16, "bt_layernorm: bad dtype %d", dtype);
    BT_CHECK_ARG(M >= 0 && H > 0 && H < (1ll << 31), "bt_layernorm: bad shape M=%lld H=%lld", (long long)M, (long long)H);
    BT_CHECK_ARG(x && w1 && b1 && out1, "bt_layernorm: null pointer");
    BT_CHECK_ARG(!out2 || (w2 && b2), "bt_layernorm: out2 given without w2/b2");
    BT_CHECK_ARG(ldx >= H && ldo1 >= H && (!out2 || ldo2 >= H), "bt_layernorm: row stride smaller than row");
    if (M == 0) return BT_OK;
    if (dtype == BT_BF16)
        return launch_ln<bf16>(M, H, x, ldx, w1, b1, out1, ldo1, w2, b2, out2, ldo2, eps, (hipStream_t)stream);
    return launch_ln<float>(M, H, x, ldx, w1, b1, out1, ldo1, w2, b2, out2, ldo2, eps, (hipStream_t)stream);
}

#ifdef BT_DEBUG_HOOKS
extern "C" int bt_debug_rowstats(int dtype, int64_t M, int64_t H, const void* x, int64_t ldx, float eps, void* stats, bt_stream stream) {
    return bt::rowstats_impl(dtype, M, H, x, ldx, eps, stats, stream);
}
extern "C" void bt_debug_ln_streaming(int on) { bt::g_ln_streaming = on != 0; }
#endif
