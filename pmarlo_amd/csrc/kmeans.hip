// k-means assignment (and, further down, the fit kernels).
//
// Layout.  Frames are rows of X [n, ld] (f32 or f64) in HBM; a lane owns R
// frames and keeps their (optionally whitened) coordinates in registers as
// fp64.  Centres [k, d] fp64 are staged into LDS in tiles together with their
// squared norms; every lane walks the tile with broadcast LDS reads
// (ds_read_b128, all lanes on one address) and a D-long fp64 FMA chain per
// centre, so the inner loop is v_fma_f64-bound (2*k*d flop/frame against
// d*s bytes/frame: compute-bound for k >~ 40).
//
// Arithmetic is pinned so that the labels are bit-reproducible on the CPU:
//   dot  = fma(z[d-1], c[d-1], ... fma(z[0], c[0], +0.0))
//   dist = fma(-2.0, dot, csq)      csq = fma chain of c[f]*c[f]
//   argmin with strict '<' over ascending centre index.
#include "common.h"

#include <algorithm>
#include <cstdlib>

namespace {

constexpr int kThreads = 256;

template <typename T>
__device__ __forceinline__ double load_as_f64(const T* p) { return (double)(*p); }

// ---------------------------------------------------------------------------
// fit: Lloyd iterations = (assign + accumulate) kernel, then a tiny update kernel.
//
// Accumulation is in 64-bit FIXED POINT: z * 2^e is rounded to an integer and added
// with integer atomics (LDS per workgroup, then global).  Integer addition commutes,
// so centre sums are bitwise independent of scheduling and of the number of GPUs the
// frames are spread over (shards all-reduce the int64 sums exactly) -- the property
// the reference tests as "same seed -> same labels"
// (tests/perf/test_msm_clustering_perf.py:241-258).  e is chosen from max|z| so that
// n_total * max|z| * 2^e < 2^62; the quantisation (<= 2^-e, ~1e-11 for O(1) data at
// 1e6 frames) is far below the k-means tolerance.
// ---------------------------------------------------------------------------
struct FitState {  // device-resident control block (doubles for easy host readback)
    double scale;      // 2^e
    double inv_scale;  // 2^-e
    double absmax;
    double shift2;     // sum ||c_new - c_old||^2 of the last update
    double tol2;       // stop when shift2 <= tol2
    double done;       // 1.0 once converged: later accumulate/update launches are no-ops
    double n_iter;
    double inertia;
};

__device__ __forceinline__ long long to_fixed(double v, double scale) { return __double2ll_rn(v * scale); }

template <typename T>
__global__ __launch_bounds__(kThreads) void absmax_kernel(const T* __restrict__ x, int64_t n, int d, int64_t ld,
                                                         const double* __restrict__ mean,
                                                         const double* __restrict__ stdv,
                                                         unsigned long long* __restrict__ out_bits) {
    double m = 0.0;
    const int64_t total = n * d;
    const int64_t step = (int64_t)gridDim.x * kThreads;
    if (ld == d && !mean) {  // contiguous, no whitening: a flat stream of 16-byte loads, 4 in flight per lane
        constexpr int V = 16 / (int)sizeof(T);
        struct alignas(16) Pack { T v[V]; };
        const bool aligned = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
        const int64_t npack = aligned ? total / V : 0;
        const Pack* xp = reinterpret_cast<const Pack*>(x);
        int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
        for (; i + 3 * step < npack; i += 4 * step) {
            Pack q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = xp[i + u * step];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < V; ++j) m = fmax(m, fabs((double)q[u].v[j]));  // fmax drops NaN operands
        }
        for (; i < npack; i += step) {
            const Pack q = xp[i];
#pragma unroll
            for (int j = 0; j < V; ++j) m = fmax(m, fabs((double)q.v[j]));
        }
        for (int64_t e = npack * V + (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += step)
            m = fmax(m, fabs((double)x[e]));
    } else {
        for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += step) {
            const int64_t r = i / d;
            const int f = (int)(i - r * d);
            double v = (double)x[r * ld + f];
            if (mean) v = (v - mean[f]) / stdv[f];
            m = fmax(m, fabs(v));
        }
    }
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off, 64));
    // one atomic per workgroup, and only when it can raise the running maximum: thousands of
    // same-address atomics would otherwise serialise in L2 and dominate the pass.
    // Non-negative doubles order like their bit patterns.
    __shared__ double wave_max[kThreads / 64];
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kThreads / 64; ++w) m = fmax(m, wave_max[w]);
        const unsigned long long bits = (unsigned long long)__double_as_longlong(m);
        if (bits > __atomic_load_n(out_bits, __ATOMIC_RELAXED)) atomicMax(out_bits, bits);
    }
}

__device__ __forceinline__ void fit_scale_body(const unsigned long long* __restrict__ absmax_bits, double n_total,
                                               double tol2, FitState* __restrict__ st) {
    double amax = __longlong_as_double((long long)*absmax_bits);
    if (!(amax > 0.0)) amax = 1.0;
    // 2^e * n_total * amax < 2^62
    int e = 61 - (int)ceil(log2(n_total * amax));
    if (e > 60) e = 60;
    if (e < -900) e = -900;
    st->scale = ldexp(1.0, e);
    st->inv_scale = ldexp(1.0, -e);
    st->absmax = amax;
    st->shift2 = 0.0;
    st->tol2 = tol2;
    st->done = 0.0;
    st->n_iter = 0.0;
    st->inertia = 0.0;
}

__global__ void fit_scale_kernel(const unsigned long long* __restrict__ absmax_bits, double n_total, double tol2,
                                 FitState* __restrict__ st) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    fit_scale_body(absmax_bits, n_total, tol2, st);
}

// centres[j] = whitened frame floor((j + u_j) * n / k), u_j = hash(seed, j) in [0, 1)
template <typename T>
__global__ void init_centers_kernel(const T* __restrict__ x, int64_t n, int d, int64_t ld,
                                    const double* __restrict__ mean, const double* __restrict__ stdv, int k,
                                    unsigned long long seed, double* __restrict__ centers,
                                    const unsigned long long* __restrict__ absmax_bits, double n_total, double tol2,
                                    FitState* __restrict__ st) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && st) fit_scale_body(absmax_bits, n_total, tol2, st);     // (fit_scale_kernel's job, one launch less)
    if (i >= k * d) return;
    const int j = i / d, f = i - j * d;
    unsigned long long h = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(j + 1);  // splitmix64
    h = (h ^ (h >> 30)) * 0xBF58476D1CE4E5B9ull;
    h = (h ^ (h >> 27)) * 0x94D049BB133111EBull;
    h ^= h >> 31;
    const double u = (double)(h >> 11) * (1.0 / 9007199254740992.0);
    int64_t t = (int64_t)(((double)j + u) * ((double)n / (double)k));
    if (t >= n) t = n - 1;
    double v = (double)x[t * ld + f];
    if (mean) v = (v - mean[f]) / stdv[f];
    centers[i] = v;
}

// centres <- sums / counts (empty clusters keep their centre); shift2 = sum ||delta||^2;
// done <- shift2 <= tol2.  One workgroup; sums/counts are cleared for the next iteration.
__global__ __launch_bounds__(1024) void kmeans_update_kernel(unsigned long long* __restrict__ sums,
                                                            unsigned long long* __restrict__ counts, int k, int d,
                                                            double* __restrict__ centers, FitState* __restrict__ st,
                                                            int clear) {
    __shared__ double red[16];
    if (st->done != 0.0) return;
    const double inv_scale = st->inv_scale;
    double acc = 0.0;
    constexpr int CH = 4;       // elements in flight per thread: the loads of a chunk before the arithmetic of any
    for (int i0 = threadIdx.x; i0 < k * d; i0 += 1024 * CH) {
        long long cnt[CH], sm[CH];
        double old[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int i = i0 + c * 1024;
            cnt[c] = 0;
            if (i < k * d) {
                cnt[c] = (long long)counts[i / d];
                sm[c] = (long long)sums[i];
                old[c] = centers[i];
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c)
            if (cnt[c] > 0) {
                const double c_new = (double)sm[c] * inv_scale / (double)cnt[c];
                const double dlt = c_new - old[c];
                acc = fma(dlt, dlt, acc);
                centers[i0 + c * 1024] = c_new;
            }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (clear) {
        for (int i = threadIdx.x; i < k * d; i += blockDim.x) sums[i] = 0ull;
        for (int i = threadIdx.x; i < k; i += blockDim.x) counts[i] = 0ull;
    }
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
        st->shift2 = t;
        st->n_iter += 1.0;
        if (t <= st->tol2) st->done = 1.0;
    }
}

// The same update for wide tables (k d above kUpdateWide: C5 has 512 k elements, 387 us in one workgroup): whole
// centres dealt out to workgroups, each leaving its share of shift2 (summed as in the one-workgroup kernel) in
// `partial`; a finishing launch adds the shares in workgroup order and closes the iteration.
constexpr int kUpdateWide = 16384;
__global__ __launch_bounds__(1024) void kmeans_update_wide_kernel(unsigned long long* __restrict__ sums,
                                                                 unsigned long long* __restrict__ counts, int k, int d,
                                                                 int centres_per_block, double* __restrict__ centers,
                                                                 const FitState* __restrict__ st, int clear,
                                                                 double* __restrict__ partial) {
    __shared__ double red[16];
    if (st->done != 0.0) return;
    const double inv_scale = st->inv_scale;
    const int j0 = blockIdx.x * centres_per_block, j1 = min(k, j0 + centres_per_block);
    const int64_t e0 = (int64_t)j0 * d, e1 = (int64_t)j1 * d;
    double acc = 0.0;
    for (int64_t i = e0 + threadIdx.x; i < e1; i += blockDim.x) {
        const int j = (int)(i / d);
        const long long cnt = (long long)counts[j];
        if (cnt > 0) {
            const double c_new = (double)(long long)sums[i] * inv_scale / (double)cnt;
            const double dlt = c_new - centers[i];
            acc = fma(dlt, dlt, acc);
            centers[i] = c_new;
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (clear) {
        for (int64_t i = e0 + threadIdx.x; i < e1; i += blockDim.x) sums[i] = 0ull;
        for (int j = j0 + threadIdx.x; j < j1; j += blockDim.x) counts[j] = 0ull;
    }
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
        partial[blockIdx.x] = t;
    }
}
__global__ void kmeans_update_finish_kernel(const double* __restrict__ partial, int nb, FitState* __restrict__ st) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || st->done != 0.0) return;
    double t = 0.0;
    for (int i = 0; i < nb; ++i) t += partial[i];
    st->shift2 = t;
    st->n_iter += 1.0;
    if (t <= st->tol2) st->done = 1.0;
}

// inertia = sum(mindist) with a fixed-order two-level reduction
__global__ __launch_bounds__(1024) void sum_partial_kernel(const double* __restrict__ v, int64_t n,
                                                          double* __restrict__ partial) {
    __shared__ double red[16];
    double acc = 0.0;
    const int64_t chunk = (n + gridDim.x - 1) / gridDim.x;
    const int64_t a = (int64_t)blockIdx.x * chunk, b = min(a + chunk, n);
    for (int64_t i = a + threadIdx.x; i < b; i += blockDim.x) acc += v[i];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += red[i];
        partial[blockIdx.x] = t;
    }
}
__global__ void sum_final_kernel(const double* __restrict__ partial, int nb, double* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < nb; ++i) t += partial[i];
        *out = t;
    }
}

// ---------------------------------------------------------------------------
// Matrix-core path: the distance table is the GEMM  dot[c][t] = sum_f C[c][f] z[t][f].
// One v_mfma_f64_16x16x4_f64 covers 16 centres x 16 frames x 4 features:
//   A (centres): lane l holds C[c0 + (l & 15)][4s + (l >> 4)]      (from the LDS tile)
//   B (frames) : lane l holds z[t0 + (l & 15)][4s + (l >> 4)]      (registers, loaded once)
//   D          : lane l, reg r = dot[c0 + (l >> 4) + 4r][t0 + (l & 15)]
// A wave keeps NF groups of 16 frames resident and walks all centre tiles, so each
// A fragment feeds NF MFMAs.  With the accumulator starting at +0.0 and the k-steps in
// ascending feature order the hardware's chain fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0,C))))
// is exactly the pinned ascending-feature FMA chain of the oracle (padding features multiply
// 0 by 0), so labels stay bit-reproducible.
//
// Score.  The arg-min runs on  m = dot - |c|^2/2  (maximised): dist = fma(-2, dot, |c|^2) equals
// -2 m EXACTLY (scaling by 2 commutes with rounding), so the order, the ties and the reported
// distance are those of the pinned formula.  When d is not a multiple of 4 the spare k-slot
// carries z = 1, c = -|c|^2/2, i.e. the MFMA chain itself ends with fma(1, -|c|^2/2, dot) = m and
// the VALU does no arithmetic at all per pair.
//
// fp64 VALU instructions and fp64 MFMAs do NOT overlap on gfx950 (tools/probe/
// kmeans_loop_probe.hip: time = MFMA time + VALU time at any interleaving), so the
// per-tile epilogue is kept minimal (see the tile loop): a max tree, one compare and the id of
// the winning tile PAIR; the winner inside that pair is recovered once per frame group by
// re-scoring its candidates with the same FMA chain from the LDS tile.
// ---------------------------------------------------------------------------
typedef double v4f64 __attribute__((ext_vector_type(4)));
// 8 waves share one LDS centre tile (+ the fixed-point accumulators): two waves per SIMD even
// when tile + accumulators take ~100 KB.
// workgroup size of the matrix-core kernel: 16 waves (4 per SIMD, <= 128 VGPRs, two frame groups
// per wave) while the frame fits 16 features, 8 waves (<= 256 VGPRs) for wider frames
constexpr int kMTNarrow = 1024, kMTWide = 512;

// one v_max_f64: fmax() would first canonicalise both operands (two more VALU ops each);
// NaNs lose against numbers here too.  The caller pads the MFMA -> VALU hazard itself.
__device__ __forceinline__ double max_f64(double a, double b) {
    double r;
    asm volatile("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Diagnostic build only (tools/probe/kmeans_stamp_probe.hip): per-phase cycle stamps of wave 0.
#ifdef MSM_KM_STAMPS
__device__ unsigned long long g_km_stamps[8];
#define KSTAMP(i)                                                                          \
    do {                                                                                   \
        unsigned long long t__;                                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");       \
        kacc__[i] += t__ - klast__;                                                        \
        klast__ = t__;                                                                     \
    } while (0)
#define KSTAMP_VM(i) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); KSTAMP(i); } while (0)
#define KSTAMP_INIT unsigned long long kacc__[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long klast__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(klast__)::"memory");
#define KSTAMP_FLUSH if (threadIdx.x == 0) { for (int i__ = 0; i__ < 8; ++i__) atomicAdd(&g_km_stamps[i__], kacc__[i__]); }
#else
#define KSTAMP(i)
#define KSTAMP_VM(i)
#define KSTAMP_INIT
#define KSTAMP_FLUSH
#endif

#include "kmeans_filter.h"

// |c_j|^2 / 2 as the ascending-feature FMA chain (multi-chunk launches: once per launch instead of
// once per workgroup and chunk)
__global__ void chalf_kernel(const double* __restrict__ centers, int k, int d, double* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    double a = 0.0;
    for (int f = 0; f < d; ++f) {
        const double c = centers[(size_t)j * d + f];
        a = fma(c, c, a);
    }
    out[j] = 0.5 * a;
}

// Chunked launches: the LDS image of every 16-centre tile (k-step major, fold slot, one padding
// double -- exactly what the tile loop reads) is built ONCE per launch in global memory, so that
// staging a chunk is a straight 16-byte-vector copy.  img[t][TS], half[k16] (+inf for padding).
template <int KS, bool FOLD>
__global__ void pack_tiles_kernel(const double* __restrict__ centers, int k, int d, double* __restrict__ img,
                                  double* __restrict__ half) {
    constexpr int TS = KS * 64 + 1;
    const int t = blockIdx.x;                    // tile
    __shared__ double h[16];
    const int lane = threadIdx.x;                // 64 threads
    if (lane < 16) {
        const int j = t * 16 + lane;
        double a = 0.0;
        if (j < k)
            for (int f = 0; f < d; ++f) {
                const double c = centers[(size_t)j * d + f];
                a = fma(c, c, a);
            }
        h[lane] = j < k ? 0.5 * a : __builtin_inf();
        half[j] = h[lane];
    }
    __syncthreads();
    const int fold_s = d >> 2, fold_g = d & 3;
    for (int rem = lane; rem < TS; rem += 64) {
        double v = 0.0;
        if (rem < KS * 64) {
            const int s = rem >> 6, gg = (rem >> 4) & 3, jj = rem & 15;
            const int j = t * 16 + jj, f = 4 * s + gg;
            if (j < k && f < d) v = centers[(size_t)j * d + f];
            if (FOLD && s == fold_s && gg == fold_g) v = -h[jj];
        }
        img[(size_t)t * TS + rem] = v;
    }
}

// MULTI: the centres do not fit one LDS tile and are staged chunk by chunk (k > tile_k).
template <typename T, int KS, int NF, int kMT, bool ACCUM, bool FOLD, bool MULTI>
__global__ __launch_bounds__(kMT, kMT == 1024 ? 4 : 2) void kmeans_mfma_kernel(
    const T* __restrict__ x, int64_t n, int d, int64_t ld, const double* __restrict__ centers, int k,
    const double* __restrict__ mean, const double* __restrict__ stdv, int tile_k /* multiple of 16 */,
    int32_t* __restrict__ labels, double* __restrict__ mindist, const FitState* __restrict__ st,
    unsigned long long* __restrict__ sums, unsigned long long* __restrict__ counts, int lds_acc,
    const double* __restrict__ chalf_g /* [k] when MULTI or KS > 8 */,
    const double* __restrict__ tile_img /* packed tile images when MULTI */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // centre tile, k-step major so that a wave's A read is one contiguous 512-byte run:
    // cs[(j / 16) * TS + s * 64 + g * 16 + (j % 16)] = C[k0 + j][4s + g].  The tile stride TS
    // is padded by one double: the winner-recovery step reads tiles chosen per lane, and a
    // stride of 2 (mod 64) dwords spreads 32 different tiles over 32 different bank pairs
    // (an unpadded 1536-byte stride put every tile on the same banks: 32-way conflicts).
    constexpr int TS = KS * 64 + 1;
    if constexpr (ACCUM) {
        if (st->done != 0.0) return;
    }
    double* cs = reinterpret_cast<double*>(smem_raw);
    double* chalf = cs + (size_t)(tile_k / 16) * TS;  // [tile_k]  |c|^2 / 2  (+inf for padding centres)
    unsigned long long* lsum = reinterpret_cast<unsigned long long*>(chalf + tile_k);  // [k][d] when lds_acc
    unsigned long long* lcnt = lsum + (size_t)k * d;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int j16 = lane & 15, g = lane >> 4;
    constexpr bool fold = FOLD;              // (d & 3) != 0: spare k-slot at feature index d carries -|c|^2/2
    const int fold_s = d >> 2, fold_g = d & 3;
    const double scale = ACCUM ? st->scale : 0.0;
    if constexpr (ACCUM) {
        if (lds_acc)
            for (int i = tid; i < k * (d + 1); i += kMT) lsum[i] = 0ull;
    }
    const int64_t frames_per_wave = 16 * NF;
    const int64_t n_units = (n + frames_per_wave - 1) / frames_per_wave;
    const int waves_per_block = kMT / 64;
    // stage centre tile [k0, k0 + kt): coordinates, then |c|^2/2 (ascending-feature FMA chain),
    // then (fold) the spare slot
    auto stage_tile = [&](int k0, int kt, int kt16) {
        if constexpr (MULTI) {
            // straight copy of the pre-packed images (pack_tiles_kernel): 16-byte vectors, four in
            // flight per thread; the chunk starts at a multiple of 16 centres
            const int n_doubles = (kt16 / 16) * TS;
            const double* src = tile_img + (size_t)(k0 / 16) * TS;
            const bool al = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(cs)) & 15) == 0;
            const int nvec = al ? n_doubles / 2 : 0;
            const double2* s2 = reinterpret_cast<const double2*>(src);
            double2* d2 = reinterpret_cast<double2*>(cs);
            int i = tid;
            for (; i + 3 * kMT < nvec; i += 4 * kMT) {
                const double2 a = s2[i], b = s2[i + kMT], c = s2[i + 2 * kMT], e = s2[i + 3 * kMT];
                d2[i] = a; d2[i + kMT] = b; d2[i + 2 * kMT] = c; d2[i + 3 * kMT] = e;
            }
            for (; i < nvec; i += kMT) d2[i] = s2[i];
            for (int q = 2 * nvec + tid; q < n_doubles; q += kMT) cs[q] = src[q];
            for (int j = tid; j < kt16; j += kMT) chalf[j] = chalf_g[k0 + j];
            __syncthreads();
            return;
        }
        // 8 independent loads in flight per thread
        const int total = kt16 * 4 * KS;
        for (int i0 = tid; i0 < total; i0 += 8 * kMT) {
            double v[8];
            int dst[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = min(i0 + q * kMT, total - 1);
                const int jt = i / (KS * 64);
                const int rem = i - jt * (KS * 64);
                const int s = rem >> 6, gg = (rem >> 4) & 3, jj = rem & 15;
                const int j = jt * 16 + jj, f = 4 * s + gg;
                const bool ok = j < kt && f < d;
                v[q] = centers[(size_t)(k0 + (ok ? j : 0)) * d + (ok ? f : 0)];
                if (!ok) v[q] = 0.0;
                dst[q] = jt * TS + rem;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (i0 + q * kMT < total) cs[dst[q]] = v[q];
        }
        __syncthreads();
        for (int j = tid; j < kt16; j += kMT) {
            double* cj = cs + (j >> 4) * TS + (j & 15);
            double h = __builtin_inf();  // padding centres can never win
            if (j < kt) {
                double a = 0.0;
                for (int f = 0; f < d; ++f) {
                    const double c = cj[(f >> 2) * 64 + (f & 3) * 16];
                    a = fma(c, c, a);
                }
                h = 0.5 * a;
            }
            chalf[j] = h;
            if (fold) cj[fold_s * 64 + fold_g * 16] = -h;
        }
        __syncthreads();
    };
    constexpr bool single_tile = !MULTI;  // the usual case: the tile is staged once per workgroup
    // winner recovery from the LDS tile (narrow frames, one chunk) or from the global centre table
    constexpr bool kGlobalRecovery = MULTI || KS > 8;
    __shared__ int unit_ctr;
    KSTAMP_INIT
    if (tid == 0) unit_ctr = kMT / 64;  // groups 0 .. waves-1 of the block's range are taken statically
    if (single_tile) stage_tile(0, k, (k + 15) & ~15);
    else __syncthreads();
    KSTAMP(0);
    // all waves of a block walk the centre tiles together (shared LDS tile), each on its own frames.
    // The raw coordinates of the NEXT frame group are requested before the tile loop of the
    // current one (clamped addresses, no branches): every wave of a block reaches its loads at the
    // same moment, so without the prefetch the whole CU would sit out each HBM round trip.
    // prefetch only in the 8-wave configuration with registers to spare (4 waves per SIMD hide
    // the load latency by themselves; wide frames need the registers for the frame)
    constexpr bool kPrefetch = KS > 4 && KS <= 16;
    T raw[kPrefetch ? NF : 1][kPrefetch ? KS : 1];
    auto fetch = [&](int64_t unit) {
        if constexpr (!kPrefetch) return;
#pragma unroll
        for (int u = 0; u < NF; ++u) {
            int64_t t = unit * frames_per_wave + 16 * u + j16;
            t = t < n ? t : n - 1;
            const T* row = x + t * ld;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int f = 4 * s + g;
                raw[kPrefetch ? u : 0][kPrefetch ? s : 0] = row[f < d ? f : d - 1];
            }
        }
    };
    // Frame groups are handed out dynamically inside a workgroup (single-tile case): the two waves
    // of a SIMD do not progress at the same rate (oldest-first arbitration), and with a static
    // split the favoured waves idle at the final barrier for ~15% of the kernel.  Each workgroup
    // owns a contiguous range of groups; a wave takes the next one from an LDS counter and
    // prefetches it.  The multi-tile case keeps the static stride (it needs block-wide barriers).
    const int wave = tid >> 6;
    const int64_t unit_stride = (int64_t)gridDim.x * waves_per_block;
    const int64_t units_per_block = (n_units + gridDim.x - 1) / gridDim.x;
    const int64_t u_begin = single_tile ? (int64_t)blockIdx.x * units_per_block : 0;
    const int64_t u_end = single_tile ? min(n_units, u_begin + units_per_block) : n_units;
    int64_t cur = single_tile ? u_begin + wave : (int64_t)blockIdx.x * waves_per_block + wave;
    fetch(cur);
    for (;;) {
        // static mode: the exit test is block-uniform (every wave takes part in the barriers)
        if (single_tile ? cur >= u_end : cur - wave >= n_units) break;
        const int64_t unit = cur;
        int64_t nxt;
        if (single_tile) {
            int t = 0;
            if (lane == 0) t = atomicAdd(&unit_ctr, 1);
            nxt = u_begin + __builtin_amdgcn_readfirstlane(t);
        } else {
            nxt = cur + unit_stride;
        }
        double zb[NF][KS];
        int64_t fidx[NF];
        bool fok[NF];
#pragma unroll
        for (int u = 0; u < NF; ++u) {
            fidx[u] = unit * frames_per_wave + 16 * u + j16;
            fok[u] = unit < n_units && fidx[u] < n;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int f = 4 * s + g;
                double v;
                if constexpr (kPrefetch) v = (double)raw[u][s];
                else {
                    const int64_t t = fidx[u] < n ? fidx[u] : n - 1;
                    v = (double)x[t * ld + (f < d ? f : d - 1)];
                }
                if (mean) v = (v - mean[f < d ? f : d - 1]) / stdv[f < d ? f : d - 1];
                if (!(fok[u] && f < d)) v = 0.0;
                if (fold && s == fold_s && g == fold_g) v = 1.0;
                zb[u][s] = v;
            }
        }
        fetch(nxt);
        cur = nxt;
        KSTAMP(1);
        // Final winner of every frame group: score (replicated on the frame's 4 lanes) and index.
        double bm[NF];
        int bidx[NF];
#pragma unroll
        for (int u = 0; u < NF; ++u) { bm[u] = -__builtin_inf(); bidx[u] = 0; }

        // running maximum and the tile pair it came from.  MULTI: carried across the chunks, the pair
        // is remembered as (global index of its first tile) * 2 + (second tile == first tile)
        double best[NF];
        int bpair[NF];
#pragma unroll
        for (int u = 0; u < NF; ++u) { best[u] = -__builtin_inf(); bpair[u] = 0; }
        for (int k0 = 0; k0 < k; k0 += tile_k) {
            const int kt = min(tile_k, k - k0);
            const int kt16 = (kt + 15) & ~15;
            if (!single_tile) {
                __syncthreads();  // previous chunk fully consumed
                stage_tile(k0, kt, kt16);
                KSTAMP(5);
            }
            const int n_tiles = kt16 / 16;
            // ---- tile loop, two 16-centre tiles per trip.  Per lane and frame group only the
            // running maximum over everything seen so far and the PAIR it came from are kept:
            // 22 VALU instructions per pair of wave-tiles (every VALU issue costs matrix-pipe time
            // on this chip, tools/probe/valu_mix_probe.hip), no arithmetic on the scores.
            for (int jt = 0; jt < n_tiles; jt += 2) {
                const int jb = min(jt + 1, n_tiles - 1);  // odd tile count: the last tile twice
                v4f64 acca[NF], accb[NF];
#pragma unroll
                for (int u = 0; u < NF; ++u) { acca[u] = (v4f64){0.0, 0.0, 0.0, 0.0}; accb[u] = acca[u]; }
                if constexpr (KS <= 4) {
                    // all of tile a, then all of tile b (the max tree below relies on tile a being
                    // complete when tile b's last MFMA has issued); A fragments stream from LDS
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        const double afa = cs[jt * TS + s * 64 + lane];
#pragma unroll
                        for (int u = 0; u < NF; ++u)
                            acca[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(afa, zb[u][s], acca[u], 0, 0, 0);
                    }
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        const double afb = cs[jb * TS + s * 64 + lane];
#pragma unroll
                        for (int u = 0; u < NF; ++u)
                            accb[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(afb, zb[u][s], accb[u], 0, 0, 0);
                    }
                } else {
                    // wide frames keep one or two frame groups per wave: the two tiles are interleaved
                    // (2 NF independent accumulator chains instead of NF) and the A fragments of the
                    // next k-step are requested before the MFMAs of this one
                    double fa = cs[jt * TS + lane], fb = cs[jb * TS + lane];
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        double na = fa, nb = fb;
                        if (s + 1 < KS) {
                            na = cs[jt * TS + (s + 1) * 64 + lane];
                            nb = cs[jb * TS + (s + 1) * 64 + lane];
                        }
#pragma unroll
                        for (int u = 0; u < NF; ++u) {
                            acca[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, zb[u][s], acca[u], 0, 0, 0);
                            accb[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb, zb[u][s], accb[u], 0, 0, 0);
                        }
                        fa = na;
                        fb = nb;
                    }
                }
                if constexpr (!fold) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double ha = chalf[jt * 16 + g + 4 * r], hb = chalf[jb * 16 + g + 4 * r];
#pragma unroll
                        for (int u = 0; u < NF; ++u) { acca[u][r] -= ha; accb[u][r] -= hb; }
                    }
                }
                // MFMA -> VALU read needs 18 wait states; hipcc pads for its own instructions but
                // not for the inline-asm v_max_f64 below (tile a finished long ago, tile b has not)
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (KS > 4) asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");  // both tiles just issued
                double ma[NF], mb[NF];
#pragma unroll
                for (int u = 0; u < NF; ++u) ma[u] = max_f64(max_f64(acca[u][0], acca[u][1]), max_f64(acca[u][2], acca[u][3]));
                asm volatile("s_nop 7" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < NF; ++u) {
                    mb[u] = max_f64(max_f64(accb[u][0], accb[u][1]), max_f64(accb[u][2], accb[u][3]));
                    const double m = max_f64(ma[u], mb[u]);
                    const bool better = m > best[u];  // strict: the first pair keeps ties
                    best[u] = better ? m : best[u];
                    const int code = kGlobalRecovery ? (((k0 >> 4) + jt) << 1) | (jb == jt ? 1 : 0) : jt;
                    bpair[u] = better ? code : bpair[u];
                }
            }
            KSTAMP(2);
            if constexpr (!kGlobalRecovery) {
            // ---- winner of this chunk, per frame.  M = max over the frame's 4 lanes; the lane(s)
            // holding M name the pair; that pair's 8 candidate centres of the lane (2 tiles x 4
            // slots) are re-scored with the SAME chain, two per lane, and the lowest index with
            // score == M wins.  Several lanes holding M (duplicate centres, exact ties) are walked
            // one after the other -- one extra trip of the loop per additional tied lane.
#pragma unroll
            for (int u = 0; u < NF; ++u) {
                double M = best[u];
                M = fmax(M, __shfl_xor(M, 16, 64));
                M = fmax(M, __shfl_xor(M, 32, 64));
                const unsigned long long tied = __ballot(best[u] == M) >> j16;  // bits 0,16,32,48: lanes g = 0..3
                unsigned pending = (unsigned)(tied & 1) | (unsigned)((tied >> 15) & 2) | (unsigned)((tied >> 30) & 4) |
                                   (unsigned)((tied >> 45) & 8);
                int found = 0x7fffffff;
                while (__any(pending != 0)) {
                    const int gs = pending ? __builtin_ctz(pending) : 0;  // candidate-owning lane of this frame
                    const int pj = __shfl(bpair[u], j16 + 16 * gs, 64);
                    const int ta = pj, tb = min(pj + 1, n_tiles - 1);
                    // this lane scores slot r = g of tiles ta and tb for owner gs: centre row gs + 4g
                    const int crow = gs + 4 * g;
                    const double* ca = cs + ta * TS + crow;
                    const double* cb = cs + tb * TS + crow;
                    double da = 0.0, db = 0.0;
#pragma unroll
                    for (int s = 0; s < KS; ++s)
#pragma unroll
                        for (int gp = 0; gp < 4; ++gp) {
                            const double zv = __shfl(zb[u][s], j16 + 16 * gp, 64);
                            da = fma(ca[s * 64 + gp * 16], zv, da);
                            db = fma(cb[s * 64 + gp * 16], zv, db);
                        }
                    if constexpr (!fold) {
                        da -= chalf[ta * 16 + crow];
                        db -= chalf[tb * 16 + crow];
                    }
                    int cand = 0x7fffffff;
                    if (pending) {
                        if (db == M) cand = k0 + tb * 16 + crow;
                        if (da == M) cand = k0 + ta * 16 + crow;  // ta <= tb: the lower index last
                    }
                    cand = min(cand, __shfl_xor(cand, 16, 64));
                    cand = min(cand, __shfl_xor(cand, 32, 64));
                    found = min(found, cand);
                    pending &= pending - 1;
                }
                if (found < k) { bm[u] = M; bidx[u] = found; }
            }
            }  // LDS recovery
        }
        if constexpr (kGlobalRecovery) {
            // ---- winner over ALL chunks: the LDS tile has moved on, so the candidates of the winning
            // pair are re-scored from the centre table in global memory (L2), and the frame is re-read
            // instead of being gathered from registers: one plain loop over the features per tied lane.
#pragma unroll
            for (int u = 0; u < NF; ++u) {
                double M = best[u];
                M = fmax(M, __shfl_xor(M, 16, 64));
                M = fmax(M, __shfl_xor(M, 32, 64));
                const unsigned long long tied = __ballot(best[u] == M) >> j16;
                unsigned pending = (unsigned)(tied & 1) | (unsigned)((tied >> 15) & 2) | (unsigned)((tied >> 30) & 4) |
                                   (unsigned)((tied >> 45) & 8);
                int found = 0x7fffffff;
                const T* row = x + (fok[u] ? fidx[u] : 0) * ld;
                while (__any(pending != 0)) {
                    const int gs = pending ? __builtin_ctz(pending) : 0;
                    const int pj = __shfl(bpair[u], j16 + 16 * gs, 64);
                    const int ta = pj >> 1, tb = ta + ((pj & 1) ? 0 : 1);
                    const int crow = gs + 4 * g;
                    const int ia = ta * 16 + crow, ib = tb * 16 + crow;      // global centre indices
                    const double* ca = centers + (size_t)min(ia, k - 1) * d;
                    const double* cb = centers + (size_t)min(ib, k - 1) * d;
                    double da = 0.0, db = 0.0;
                    int f = 0;
                    for (; f + 8 <= d; f += 8) {   // loads of 8 features in flight, FMAs in feature order
                        double v[8], va[8], vb[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            v[q] = load_as_f64(row + f + q);
                            va[q] = ca[f + q];
                            vb[q] = cb[f + q];
                        }
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            double z = v[q];
                            if (mean) z = (z - mean[f + q]) / stdv[f + q];
                            da = fma(va[q], z, da);
                            db = fma(vb[q], z, db);
                        }
                    }
                    for (; f < d; ++f) {
                        double v = load_as_f64(row + f);
                        if (mean) v = (v - mean[f]) / stdv[f];
                        da = fma(ca[f], v, da);
                        db = fma(cb[f], v, db);
                    }
                    // fold: the spare k-slot contributes fma(-h, 1, .) = . - h; otherwise the VALU subtracts h
                    da -= chalf_g[min(ia, k - 1)];
                    db -= chalf_g[min(ib, k - 1)];
                    int cand = 0x7fffffff;
                    if (pending) {
                        if (ib < k && db == M) cand = ib;
                        if (ia < k && da == M) cand = ia;  // ia <= ib: the lower index last
                    }
                    cand = min(cand, __shfl_xor(cand, 16, 64));
                    cand = min(cand, __shfl_xor(cand, 32, 64));
                    found = min(found, cand);
                    pending &= pending - 1;
                }
                if (found < k) { bm[u] = M; bidx[u] = found; }
            }
        }
        KSTAMP(3);
#pragma unroll
        for (int u = 0; u < NF; ++u) {
            if (!fok[u]) continue;
            if constexpr (ACCUM) {
                // delta mode (labels != NULL): only frames that changed centre since the last pass move their
                // contribution (kmeans_filter.h has the same rule); the four g lanes of a frame read the old label
                // before lane g == 0 replaces it
                const int old = labels ? labels[fidx[u]] : -1;
                if (!labels || old != bidx[u]) {
                    unsigned long long* base = lds_acc ? lsum : sums;
                    unsigned long long* srow = base + (size_t)bidx[u] * d;
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        const int f = 4 * s + g;
                        if (f < d) {
                            const unsigned long long fx = (unsigned long long)to_fixed(zb[u][s], scale);
                            atomicAdd(&srow[f], fx);
                            if (old >= 0) atomicAdd(&base[(size_t)old * d + f], 0ull - fx);
                        }
                    }
                    if (g == 0) {
                        atomicAdd((lds_acc ? lcnt : counts) + bidx[u], 1ull);
                        if (old >= 0) atomicAdd((lds_acc ? lcnt : counts) + old, ~0ull);
                        if (labels) labels[fidx[u]] = bidx[u];
                    }
                }
            } else {
                if (g == 0) {
                    labels[fidx[u]] = bidx[u];
                    if (mindist) {  // |z|^2 as the ascending-feature chain of the oracle
                        const T* row = x + fidx[u] * ld;
                        double zsq = 0.0;
                        for (int f = 0; f < d; ++f) {
                            double v = load_as_f64(row + f);
                            if (mean) v = (v - mean[f]) / stdv[f];
                            zsq = fma(v, v, zsq);
                        }
                        const double md = -2.0 * bm[u] + zsq;  // -2 m == fma(-2, dot, |c|^2) exactly
                        mindist[fidx[u]] = md > 0.0 ? md : 0.0;
                    }
                }
            }
        }
        KSTAMP(4);
    }
    KSTAMP(7);
    if constexpr (ACCUM) {
        if (lds_acc) {
            __syncthreads();
            KSTAMP(5);
            for (int i = tid; i < k * d; i += kMT)
                if (lsum[i]) atomicAdd(&sums[i], lsum[i]);
            for (int i = tid; i < k; i += kMT)
                if (lcnt[i]) atomicAdd(&counts[i], lcnt[i]);
        }
    }
    KSTAMP(6);
    KSTAMP_FLUSH
}

template <typename T, int KS, bool ACCUM>
msm_status launch_mfma(msm_ctx* ctx, const T* x, int64_t n, int d, int64_t ld, const double* centers, int k,
                       const double* mean, const double* stdv, int32_t* labels, double* mindist, const FitState* st,
                       unsigned long long* sums, unsigned long long* counts) {
    constexpr int NF = KS <= 16 ? 2 : 1;
    constexpr int kMT = KS <= 4 ? kMTNarrow : kMTWide;
    constexpr int TS = KS * 64 + 1;  // doubles per 16-centre tile (see the kernel)
    const size_t acc_bytes = ACCUM ? (size_t)k * (d + 1) * sizeof(unsigned long long) : 0;
    const int lds_acc = ACCUM && acc_bytes <= 64 * 1024;
    const size_t tile_budget = 150 * 1024 - (lds_acc ? acc_bytes : 0);  // one 8-wave workgroup per CU
    int tile_k = (int)(tile_budget / ((TS + 16) * sizeof(double))) * 16;
    const int k16 = (k + 15) & ~15;
    if (tile_k > k16) tile_k = k16;
    if (tile_k < 16) tile_k = 16;
    // chunked launches: an even number of tiles per chunk keeps every chunk of the packed image
    // (odd tile stride TS) 16-byte aligned for the vector copy
    if (tile_k < k16 && tile_k >= 32) tile_k &= ~31;
    const size_t lds = (size_t)(tile_k / 16) * (TS + 16) * sizeof(double) + (lds_acc ? acc_bytes : 0);
    const int64_t n_units = (n + 16 * NF - 1) / (16 * NF);
    const int waves = kMT / 64;
    const int grid = (int)std::min<int64_t>((n_units + waves - 1) / waves, (int64_t)ctx->n_cu);
    const bool multi = k > tile_k;
    const bool foldm = (d & 3) != 0;
    auto kern = multi ? (foldm ? kmeans_mfma_kernel<T, KS, NF, kMT, ACCUM, true, true>
                               : kmeans_mfma_kernel<T, KS, NF, kMT, ACCUM, false, true>)
                      : (foldm ? kmeans_mfma_kernel<T, KS, NF, kMT, ACCUM, true, false>
                               : kmeans_mfma_kernel<T, KS, NF, kMT, ACCUM, false, false>);
    double* chalf_g = nullptr;
    double* tile_img = nullptr;
    if (multi) {   // packed tile images + half-norms, once per launch (the main scratch may hold the
                   // caller's member sums, hence the auxiliary buffer)
        const size_t n_tiles_all = (size_t)k16 / 16;
        const size_t img_doubles = (n_tiles_all * TS + 1) & ~(size_t)1;
        msm_status rs = msm_reserve_aux(ctx, (img_doubles + k16) * sizeof(double));
        if (rs != MSM_OK) return rs;
        tile_img = (double*)ctx->aux;
        chalf_g = tile_img + img_doubles;
        if (foldm)
            hipLaunchKernelGGL((pack_tiles_kernel<KS, true>), dim3((unsigned)n_tiles_all), dim3(64), 0, ctx->stream, centers,
                               k, d, tile_img, chalf_g);
        else
            hipLaunchKernelGGL((pack_tiles_kernel<KS, false>), dim3((unsigned)n_tiles_all), dim3(64), 0, ctx->stream, centers,
                               k, d, tile_img, chalf_g);
    } else if (KS > 8) {   // the global winner recovery reads the half-norms from a table
        msm_status rs = msm_reserve_aux(ctx, (size_t)k * sizeof(double));
        if (rs != MSM_OK) return rs;
        chalf_g = (double*)ctx->aux;
        hipLaunchKernelGGL(chalf_kernel, dim3((k + 255) / 256), dim3(256), 0, ctx->stream, centers, k, d, chalf_g);
    }
    if (lds > 48 * 1024)
        MSM_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kMT), lds, ctx->stream, x, n, d, ld, centers, k, mean, stdv, tile_k,
                       labels, mindist, st, sums, counts, lds_acc, (const double*)chalf_g, (const double*)tile_img);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

// one v_mfma_f32_16x16x32_bf16 per workgroup on operands given element by element (msm_mfma_bf16_probe)
__global__ void mfma_bf16_probe_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B,
                                       const float* __restrict__ C, float* __restrict__ D) {
    const int l = threadIdx.x, i = l & 15, q = l >> 4;
    A += (size_t)blockIdx.x * 512;
    B += (size_t)blockIdx.x * 512;
    C += (size_t)blockIdx.x * 256;
    D += (size_t)blockIdx.x * 256;
    v8bf a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = __builtin_bit_cast(__bf16, A[i * 32 + 8 * q + j]);      // A[row i][k = 8 q + j]
        b[j] = __builtin_bit_cast(__bf16, B[(8 * q + j) * 16 + i]);    // B[k = 8 q + j][column i]
    }
    v4f32 c;
    for (int r = 0; r < 4; ++r) c[r] = C[(4 * q + r) * 16 + i];
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + i] = c[r];
}

// ---- certified bf16 filter path (kmeans_filter.h): d <= 10 and centres + images fit the LDS -------------
bool filter_enabled() {
    static const bool on = [] {
        const char* e = getenv("MSM_KMEANS_FILTER");   // MSM_KMEANS_FILTER=0: all-fp64 kernel everywhere (A/B timing)
        return !(e && e[0] == '0');
    }();
    return on;
}

// whole units of 64 rows (the filter kernel loads a unit without clamping)
size_t filter_image_bytes(int64_t n, int d) { return (size_t)((n + 63) & ~(int64_t)63) * 16 * filter_rowq(filter_nm(d)); }

template <typename T>
msm_status launch_pack(msm_ctx* ctx, const T* x, int64_t n, int d, int64_t ld, const double* mean, const double* stdv,
                       uint4* image) {
    const unsigned grid = (unsigned)((n + 255) / 256);
    const int dense = ld == d && ((uintptr_t)x & 15) == 0;
#define MSM_PACK_CASE(D)                                                                                              \
    case D:                                                                                                           \
        hipLaunchKernelGGL((kmeans_pack_kernel<T, D>), dim3(grid), dim3(256), 0, ctx->stream, x, n, ld, mean, stdv, image, \
                           dense);                                                                                    \
        break
    switch (d) {
        MSM_PACK_CASE(1); MSM_PACK_CASE(2); MSM_PACK_CASE(3); MSM_PACK_CASE(4); MSM_PACK_CASE(5);
        MSM_PACK_CASE(6); MSM_PACK_CASE(7); MSM_PACK_CASE(8); MSM_PACK_CASE(9); MSM_PACK_CASE(10);
        default: return msm_fail(ctx, MSM_ERR_UNSUPPORTED, "k-means frame images need d <= %d (got %d)", kFilterMaxD, d);
    }
#undef MSM_PACK_CASE
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status filter_stats_buffer(msm_ctx* ctx, unsigned long long** out) {
    // one device word counting the frames that took the exhaustive scan (diagnostics, msm_kmeans_filter_scanned)
    static_assert(sizeof(unsigned long long) == 8, "");
    if (!ctx->km_stats) {
        if (ctx->capturing) { *out = nullptr; return MSM_OK; }
        MSM_HIP(ctx, hipMalloc(&ctx->km_stats, 16));       // [frames scanned u64 | arrival ticket of the fused update u32 | pad]
        MSM_HIP(ctx, hipMemsetAsync(ctx->km_stats, 0, 16, ctx->stream));
    }
    *out = (unsigned long long*)ctx->km_stats;
    return MSM_OK;
}

template <typename T, int NM, bool ACCUM>
msm_status launch_filter_nm(msm_ctx* ctx, const T* x, int64_t n, int d, int64_t ld, const double* centers, int k,
                            const double* mean, const double* stdv, const uint4* image, int32_t* labels, double* mindist,
                            const FitState* st, unsigned long long* sums, unsigned long long* counts, double* upd_centers) {
    constexpr int NF = 4;
    const size_t lds = filter_lds_bytes(k, d, ACCUM);
    // (the centre tables are built by every workgroup of the kernel for itself: no staging launch)
    msm_status rs;
    const int64_t n_units = (n + 16 * NF - 1) / (16 * NF);
    constexpr int kWaves = filter_waves(NM);
    const int grid = (int)std::min<int64_t>((n_units + kWaves - 1) / kWaves, (int64_t)ctx->n_cu);
    unsigned long long* stats = nullptr;
    rs = filter_stats_buffer(ctx, &stats);
    if (rs != MSM_OK) return rs;
    auto kern = mean ? kmeans_filter_kernel<T, NM, NF, ACCUM, true> : kmeans_filter_kernel<T, NM, NF, ACCUM, false>;
    if (lds > 48 * 1024)
        MSM_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * kWaves), lds, ctx->stream, x, n, d, ld, k, mean, stdv, image, centers,
                       labels, mindist, st, sums, counts, stats, stats ? upd_centers : nullptr,
                       stats ? (unsigned int*)(stats + 1) : nullptr);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

template <typename T, bool ACCUM>
msm_status launch_filter(msm_ctx* ctx, const T* x, int64_t n, int d, int64_t ld, const double* centers, int k,
                         const double* mean, const double* stdv, const uint4* image, int32_t* labels, double* mindist,
                         const FitState* st, unsigned long long* sums, unsigned long long* counts, double* upd_centers) {
    if (filter_nm(d) == 1)
        return launch_filter_nm<T, 1, ACCUM>(ctx, x, n, d, ld, centers, k, mean, stdv, image, labels, mindist, st, sums,
                                             counts, upd_centers);
    return launch_filter_nm<T, 2, ACCUM>(ctx, x, n, d, ld, centers, k, mean, stdv, image, labels, mindist, st, sums, counts,
                                         upd_centers);
}

bool filter_fits(int k, int d, bool accum) { return filter_enabled() && d <= kFilterMaxD && filter_lds_bytes(k, d, accum) != 0; }

// image: the frames' bf16 images (msm_kmeans_pack) or NULL (built here into the context's buffer)
template <typename T, bool ACCUM>
msm_status dispatch_mfma(msm_ctx* ctx, const T* x, int64_t n, int d, int64_t ld, const double* centers, int k,
                         const double* mean, const double* stdv, int32_t* labels, double* mindist, const FitState* st,
                         unsigned long long* sums, unsigned long long* counts, const void* image = nullptr,
                         double* upd_centers = nullptr, bool* folded = nullptr) {
    // upd_centers: close the Lloyd iteration inside the accumulate launch (its last workgroup); *folded tells the caller
    // whether that happened (filter path, small table, not under a capture without the ticket word) or the separate update
    // launch is still due
    if (folded) *folded = false;
    if (filter_fits(k, d, ACCUM)) {
        if (!image) {
            msm_status rs = msm_reserve_km_image(ctx, filter_image_bytes(n, d));
            if (rs != MSM_OK) return rs;
            rs = launch_pack<T>(ctx, x, n, d, ld, mean, stdv, (uint4*)ctx->km_image);
            if (rs != MSM_OK) return rs;
            image = ctx->km_image;
        }
        if (upd_centers && !(ACCUM && (int64_t)k * d <= kUpdateWide && (ctx->km_stats || !ctx->capturing))) upd_centers = nullptr;
        if (folded) *folded = upd_centers != nullptr;
        return launch_filter<T, ACCUM>(ctx, x, n, d, ld, centers, k, mean, stdv, (const uint4*)image, labels, mindist, st,
                                       sums, counts, upd_centers);
    }
#define MSM_MFMA_CASE(KSV) \
    if (d <= 4 * KSV)      \
        return launch_mfma<T, KSV, ACCUM>(ctx, x, n, d, ld, centers, k, mean, stdv, labels, mindist, st, sums, counts)
    MSM_MFMA_CASE(1);
    MSM_MFMA_CASE(2);
    MSM_MFMA_CASE(3);
    MSM_MFMA_CASE(4);
    MSM_MFMA_CASE(6);
    MSM_MFMA_CASE(8);
    MSM_MFMA_CASE(12);
    MSM_MFMA_CASE(16);
    MSM_MFMA_CASE(24);
    MSM_MFMA_CASE(32);
    MSM_MFMA_CASE(48);
    MSM_MFMA_CASE(64);
#undef MSM_MFMA_CASE
    return msm_fail(ctx, MSM_ERR_UNSUPPORTED, "k-means: d=%d > 256 not supported", d);
}

}  // namespace

extern "C" {

static msm_status kmeans_assign_impl(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int d, int64_t ld,
                                     const double* d_centers, int k, const double* d_mean, const double* d_std,
                                     const void* d_image, int32_t* d_labels, double* d_mindist) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 0 && d >= 1 && k >= 1, "msm_kmeans_assign: need n >= 0, d >= 1, k >= 1");
    MSM_REQUIRE(ctx, ld >= d, "msm_kmeans_assign: ld (%lld) < d (%d)", (long long)ld, d);
    MSM_REQUIRE(ctx, (d_mean == nullptr) == (d_std == nullptr),
                "msm_kmeans_assign: mean and std must both be given or both be NULL");
    MSM_REQUIRE(ctx, dtype == MSM_F32 || dtype == MSM_F64, "msm_kmeans_assign: bad dtype");
    if (n == 0) return MSM_OK;
    MSM_REQUIRE(ctx, d_x && d_centers && d_labels, "msm_kmeans_assign: NULL pointer");
    if (dtype == MSM_F32)
        return dispatch_mfma<float, false>(ctx, (const float*)d_x, n, d, ld, d_centers, k, d_mean, d_std, d_labels,
                                           d_mindist, nullptr, nullptr, nullptr, d_image);
    return dispatch_mfma<double, false>(ctx, (const double*)d_x, n, d, ld, d_centers, k, d_mean, d_std, d_labels,
                                        d_mindist, nullptr, nullptr, nullptr, d_image);
}

msm_status msm_kmeans_assign(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int d, int64_t ld,
                             const double* d_centers, int k, const double* d_mean, const double* d_std,
                             int32_t* d_labels, double* d_mindist) {
    return kmeans_assign_impl(ctx, d_x, dtype, n, d, ld, d_centers, k, d_mean, d_std, nullptr, d_labels, d_mindist);
}

msm_status msm_kmeans_assign_packed(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int d, int64_t ld,
                                    const double* d_centers, int k, const double* d_mean, const double* d_std,
                                    const void* d_image, int32_t* d_labels, double* d_mindist) {
    return kmeans_assign_impl(ctx, d_x, dtype, n, d, ld, d_centers, k, d_mean, d_std, d_image, d_labels, d_mindist);
}

msm_status msm_kmeans_image_bytes(int64_t n, int d, size_t* out_bytes) {
    if (!out_bytes || n < 0 || d < 1) return MSM_ERR_INVALID;
    *out_bytes = d <= kFilterMaxD ? filter_image_bytes(n, d) : 0;
    return MSM_OK;
}

msm_status msm_kmeans_pack(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int d, int64_t ld,
                           const double* d_mean, const double* d_std, void* d_image) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 0 && d >= 1 && d <= kFilterMaxD && ld >= d, "msm_kmeans_pack: need 1 <= d <= %d, ld >= d",
                kFilterMaxD);
    MSM_REQUIRE(ctx, (d_mean == nullptr) == (d_std == nullptr), "msm_kmeans_pack: mean/std must come together");
    MSM_REQUIRE(ctx, dtype == MSM_F32 || dtype == MSM_F64, "msm_kmeans_pack: bad dtype");
    if (n == 0) return MSM_OK;
    MSM_REQUIRE(ctx, d_x && d_image, "msm_kmeans_pack: NULL pointer");
    MSM_REQUIRE(ctx, ((uintptr_t)d_image & 15) == 0, "msm_kmeans_pack: the image must be 16-byte aligned");
    if (dtype == MSM_F32) return launch_pack<float>(ctx, (const float*)d_x, n, d, ld, d_mean, d_std, (uint4*)d_image);
    return launch_pack<double>(ctx, (const double*)d_x, n, d, ld, d_mean, d_std, (uint4*)d_image);
}

msm_status msm_kmeans_filter_scanned(msm_ctx* ctx, uint64_t* h_out, int reset) {
    if (!ctx || !h_out) return MSM_ERR_INVALID;
    *h_out = 0;
    if (!ctx->km_stats) return MSM_OK;
    MSM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    MSM_HIP(ctx, hipMemcpy(h_out, ctx->km_stats, 8, hipMemcpyDeviceToHost));
    if (reset) MSM_HIP(ctx, hipMemset(ctx->km_stats, 0, 8));
    return MSM_OK;
}

msm_status msm_mfma_bf16_probe(msm_ctx* ctx, const uint16_t* h_a, const uint16_t* h_b, const float* h_c, float* h_d,
                               int n_tiles) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n_tiles >= 1 && h_a && h_b && h_c && h_d, "msm_mfma_bf16_probe: bad arguments");
    const size_t na = (size_t)n_tiles * 512 * sizeof(uint16_t), nc = (size_t)n_tiles * 256 * sizeof(float);
    msm_status rs = msm_reserve_scratch(ctx, 2 * na + 2 * nc);
    if (rs != MSM_OK) return rs;
    unsigned char* base = (unsigned char*)ctx->scratch;
    uint16_t* d_a = (uint16_t*)base;
    uint16_t* d_b = (uint16_t*)(base + na);
    float* d_c = (float*)(base + 2 * na);
    float* d_d = (float*)(base + 2 * na + nc);
    MSM_HIP(ctx, hipMemcpyAsync(d_a, h_a, na, hipMemcpyHostToDevice, ctx->stream));
    MSM_HIP(ctx, hipMemcpyAsync(d_b, h_b, na, hipMemcpyHostToDevice, ctx->stream));
    MSM_HIP(ctx, hipMemcpyAsync(d_c, h_c, nc, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(mfma_bf16_probe_kernel, dim3((unsigned)n_tiles), dim3(64), 0, ctx->stream, d_a, d_b, d_c, d_d);
    MSM_CHECK_LAUNCH(ctx);
    MSM_HIP(ctx, hipMemcpyAsync(h_d, d_d, nc, hipMemcpyDeviceToHost, ctx->stream));
    MSM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MSM_OK;
}

msm_status msm_kmeans_fit_begin(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int d, int64_t ld,
                                const double* d_mean, const double* d_std, int k, uint64_t seed, int init_centers,
                                double n_total, double tol2, double* d_centers, double* d_state, int absmax_ready) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 1 && d >= 1 && d <= 256 && k >= 1 && ld >= d, "msm_kmeans_fit_begin: bad shape");
    MSM_REQUIRE(ctx, !absmax_ready || !d_mean, "msm_kmeans_fit_begin: a precomputed max |x| excludes whitening");
    MSM_REQUIRE(ctx, !init_centers || n >= k, "msm_kmeans_fit_begin: fewer frames (%lld) than centres (%d)",
                (long long)n, k);
    MSM_REQUIRE(ctx, (d_mean == nullptr) == (d_std == nullptr), "msm_kmeans_fit_begin: mean/std must come together");
    MSM_REQUIRE(ctx, d_x && d_centers && d_state, "msm_kmeans_fit_begin: NULL pointer");
    MSM_REQUIRE(ctx, dtype == MSM_F32 || dtype == MSM_F64, "msm_kmeans_fit_begin: bad dtype");
    MSM_REQUIRE(ctx, n_total >= (double)n && tol2 >= 0.0, "msm_kmeans_fit_begin: bad n_total / tol2");
    // d_state doubles as scratch for the absmax bits (slot 2 = absmax)
    unsigned long long* bits = (unsigned long long*)(d_state + 2);
    if (!absmax_ready) {   // otherwise msm_project (d_absmax = d_state + 2) left it there while writing d_x
        MSM_HIP(ctx, hipMemsetAsync(bits, 0, sizeof(unsigned long long), ctx->stream));
        const int grid = (int)std::min<int64_t>((n * d + kThreads * 8 - 1) / (kThreads * 8), (int64_t)ctx->n_cu * 8);
        if (dtype == MSM_F32)
            hipLaunchKernelGGL(absmax_kernel<float>, dim3(grid), dim3(kThreads), 0, ctx->stream, (const float*)d_x, n, d,
                               ld, d_mean, d_std, bits);
        else
            hipLaunchKernelGGL(absmax_kernel<double>, dim3(grid), dim3(kThreads), 0, ctx->stream, (const double*)d_x, n,
                               d, ld, d_mean, d_std, bits);
        MSM_CHECK_LAUNCH(ctx);
    }
    if (!init_centers) {
        hipLaunchKernelGGL(fit_scale_kernel, dim3(1), dim3(64), 0, ctx->stream, bits, n_total, tol2, (FitState*)d_state);
        MSM_CHECK_LAUNCH(ctx);
    }
    if (init_centers) {
        const int g2 = msm_ceil_div((int64_t)k * d, 256);
        if (dtype == MSM_F32)
            hipLaunchKernelGGL(init_centers_kernel<float>, dim3(g2), dim3(256), 0, ctx->stream, (const float*)d_x, n, d,
                               ld, d_mean, d_std, k, (unsigned long long)seed, d_centers, bits, n_total, tol2,
                               (FitState*)d_state);
        else
            hipLaunchKernelGGL(init_centers_kernel<double>, dim3(g2), dim3(256), 0, ctx->stream, (const double*)d_x, n,
                               d, ld, d_mean, d_std, k, (unsigned long long)seed, d_centers, bits, n_total, tol2,
                               (FitState*)d_state);
        MSM_CHECK_LAUNCH(ctx);
    }
    return MSM_OK;
}

static msm_status kmeans_accumulate_impl(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int d, int64_t ld,
                                         const double* d_centers, int k, const double* d_mean, const double* d_std,
                                         const void* d_image, const double* d_state, int64_t* d_sums, int64_t* d_counts,
                                         int32_t* d_prev_labels = nullptr, double* d_update_centers = nullptr,
                                         bool* folded = nullptr) {
    if (folded) *folded = false;
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 0 && d >= 1 && k >= 1 && ld >= d, "msm_kmeans_accumulate: bad shape");
    MSM_REQUIRE(ctx, (d_mean == nullptr) == (d_std == nullptr), "msm_kmeans_accumulate: mean/std must come together");
    MSM_REQUIRE(ctx, dtype == MSM_F32 || dtype == MSM_F64, "msm_kmeans_accumulate: bad dtype");
    if (n == 0) return MSM_OK;
    MSM_REQUIRE(ctx, d_x && d_centers && d_state && d_sums && d_counts, "msm_kmeans_accumulate: NULL pointer");
    if (dtype == MSM_F32)
        return dispatch_mfma<float, true>(ctx, (const float*)d_x, n, d, ld, d_centers, k, d_mean, d_std, d_prev_labels,
                                          nullptr, (const FitState*)d_state, (unsigned long long*)d_sums,
                                          (unsigned long long*)d_counts, d_image, d_update_centers, folded);
    return dispatch_mfma<double, true>(ctx, (const double*)d_x, n, d, ld, d_centers, k, d_mean, d_std, d_prev_labels,
                                       nullptr, (const FitState*)d_state, (unsigned long long*)d_sums,
                                       (unsigned long long*)d_counts, d_image, d_update_centers, folded);
}

// One Lloyd iteration: member sums of the frames under `d_centers` (following the frames that change centre when
// d_prev_labels is given), then centres <- sums / counts, shift and convergence flag -- in ONE launch where the filter
// kernel runs (its last workgroup closes the iteration), else msm_kmeans_accumulate* + msm_kmeans_update.
msm_status msm_kmeans_lloyd_pass(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int d, int64_t ld,
                                 double* d_centers, int k, const double* d_mean, const double* d_std, const void* d_image,
                                 double* d_state, int32_t* d_prev_labels, int64_t* d_sums, int64_t* d_counts) {
    bool folded = false;
    msm_status rs = kmeans_accumulate_impl(ctx, d_x, dtype, n, d, ld, d_centers, k, d_mean, d_std, d_image, d_state, d_sums,
                                           d_counts, d_prev_labels, d_centers, &folded);
    if (rs != MSM_OK || folded) return rs;
    return msm_kmeans_update(ctx, d_sums, d_counts, k, d, d_centers, d_state, 0);
}

msm_status msm_kmeans_accumulate_delta(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int d, int64_t ld,
                                       const double* d_centers, int k, const double* d_mean, const double* d_std,
                                       const void* d_image, const double* d_state, int32_t* d_prev_labels,
                                       int64_t* d_sums, int64_t* d_counts) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, d_prev_labels || n == 0, "msm_kmeans_accumulate_delta: NULL label buffer");
    return kmeans_accumulate_impl(ctx, d_x, dtype, n, d, ld, d_centers, k, d_mean, d_std, d_image, d_state, d_sums,
                                  d_counts, d_prev_labels);
}

msm_status msm_kmeans_accumulate(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int d, int64_t ld,
                                 const double* d_centers, int k, const double* d_mean, const double* d_std,
                                 const double* d_state, int64_t* d_sums, int64_t* d_counts) {
    return kmeans_accumulate_impl(ctx, d_x, dtype, n, d, ld, d_centers, k, d_mean, d_std, nullptr, d_state, d_sums,
                                  d_counts);
}

msm_status msm_kmeans_accumulate_packed(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int d, int64_t ld,
                                        const double* d_centers, int k, const double* d_mean, const double* d_std,
                                        const void* d_image, const double* d_state, int64_t* d_sums,
                                        int64_t* d_counts) {
    return kmeans_accumulate_impl(ctx, d_x, dtype, n, d, ld, d_centers, k, d_mean, d_std, d_image, d_state, d_sums,
                                  d_counts);
}

msm_status msm_kmeans_update(msm_ctx* ctx, int64_t* d_sums, int64_t* d_counts, int k, int d, double* d_centers,
                             double* d_state, int clear) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, k >= 1 && d >= 1, "msm_kmeans_update: bad shape");
    MSM_REQUIRE(ctx, d_sums && d_counts && d_centers && d_state, "msm_kmeans_update: NULL pointer");
    if ((int64_t)k * d > kUpdateWide) {
        const int nb = std::min(k, ctx->n_cu);
        const int cpb = (k + nb - 1) / nb;
        const int nblocks = (k + cpb - 1) / cpb;
        msm_status rs = msm_reserve_aux(ctx, (size_t)nblocks * sizeof(double));
        if (rs != MSM_OK) return rs;
        hipLaunchKernelGGL(kmeans_update_wide_kernel, dim3(nblocks), dim3(1024), 0, ctx->stream, (unsigned long long*)d_sums,
                           (unsigned long long*)d_counts, k, d, cpb, d_centers, (const FitState*)d_state, clear,
                           (double*)ctx->aux);
        MSM_CHECK_LAUNCH(ctx);
        hipLaunchKernelGGL(kmeans_update_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, (const double*)ctx->aux, nblocks,
                           (FitState*)d_state);
        MSM_CHECK_LAUNCH(ctx);
        return MSM_OK;
    }
    hipLaunchKernelGGL(kmeans_update_kernel, dim3(1), dim3(1024), 0, ctx->stream, (unsigned long long*)d_sums,
                       (unsigned long long*)d_counts, k, d, d_centers, (FitState*)d_state, clear);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_kmeans_fit(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int d, int64_t ld,
                          const double* d_mean, const double* d_std, int k, uint64_t seed, int init_centers,
                          int max_iter, double tol2, double* d_centers, double* d_state) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, max_iter >= 0, "msm_kmeans_fit: max_iter must be >= 0");
    msm_status rs = msm_kmeans_fit_begin(ctx, d_x, dtype, n, d, ld, d_mean, d_std, k, seed, init_centers, (double)n,
                                         tol2, d_centers, d_state, 0);
    if (rs != MSM_OK) return rs;
    const size_t acc_bytes = ((size_t)k * (d + 1) * sizeof(unsigned long long) + 15) & ~(size_t)15;
    rs = msm_reserve_scratch(ctx, acc_bytes + (size_t)n * sizeof(int32_t));
    if (rs != MSM_OK) return rs;
    int64_t* sums = (int64_t*)ctx->scratch;
    int64_t* counts = sums + (size_t)k * d;
    // the member sums persist over the iterations and follow the frames that change centre (integer sums: the
    // bits of a full re-accumulation); prev = the centre each frame is booked under, -1 before the first pass
    int32_t* prev = (int32_t*)((char*)ctx->scratch + acc_bytes);
    MSM_HIP(ctx, hipMemsetAsync(sums, 0, acc_bytes, ctx->stream));
    MSM_HIP(ctx, hipMemsetAsync(prev, 0xFF, (size_t)n * sizeof(int32_t), ctx->stream));
    // the frames' bf16 images are built once and serve every iteration (kmeans_filter.h)
    const void* image = nullptr;
    if (max_iter > 0 && filter_fits(k, d, true)) {
        rs = msm_reserve_km_image(ctx, filter_image_bytes(n, d));
        if (rs != MSM_OK) return rs;
        rs = msm_kmeans_pack(ctx, d_x, dtype, n, d, ld, d_mean, d_std, ctx->km_image);
        if (rs != MSM_OK) return rs;
        image = ctx->km_image;
    }
    for (int it = 0; it < max_iter; ++it) {
        rs = msm_kmeans_lloyd_pass(ctx, d_x, dtype, n, d, ld, d_centers, k, d_mean, d_std, image, d_state, prev, sums, counts);
        if (rs != MSM_OK) return rs;
    }
    return MSM_OK;
}

msm_status msm_sum_f64(msm_ctx* ctx, const double* d_v, int64_t n, double* d_out) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 0 && d_out && (d_v || n == 0), "msm_sum_f64: bad arguments");
    const int nb = (int)std::min<int64_t>(256, std::max<int64_t>(1, n / 4096));
    msm_status rs = msm_reserve_scratch(ctx, (size_t)nb * sizeof(double));
    if (rs != MSM_OK) return rs;
    hipLaunchKernelGGL(sum_partial_kernel, dim3(nb), dim3(1024), 0, ctx->stream, d_v, n, (double*)ctx->scratch);
    MSM_CHECK_LAUNCH(ctx);
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(64), 0, ctx->stream, (const double*)ctx->scratch, nb, d_out);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

}  // extern "C"
