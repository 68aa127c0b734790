// k-means++ seeding on the device (Arthur & Vassilvitskii: every new centre is a frame drawn with probability
// proportional to its squared distance D^2 to the nearest centre already chosen).  deeptime's KMeans, which the
// reference constructs, starts from init_strategy = 'kmeans++' (S/markov_state_model/clustering.py:322-361), sklearn's
// KMeans of _KMeansDiscretizer from 'k-means++' (S/analysis/discretize.py:458-469).  Their random streams cannot be
// reproduced; what is reproduced is the algorithm, in a form a CPU restatement repeats bit for bit
// (oracle/npport.kmeans_plusplus):
//
//   * D^2_t = sum over ascending f of (z_tf - c_f)^2 with a separate multiplication and addition per feature
//     (no fma: numpy has none), min over the chosen centres;
//   * the draw works on INTEGER weights w_t = rint(D^2_t 2^e), 2^e from max |z| so that the total stays below 2^62:
//     integer sums are exact in any order, so the block sums, their scan and the pick cannot depend on the
//     schedule;
//   * the j-th uniform is splitmix64(seed ^ "k++kmean", j) and the draw is r = floor(h W / 2^64) (the high half of
//     a 64 x 64-bit product): the frame with  cumsum(w)[t] > r  first;  W = 0 (every frame sits on a centre): frame
//     floor(h n / 2^64).
//
// Two launches per centre: `kpp_update_kernel` (all frames: D^2 against the newest centre, running minimum, block
// sums of the weights) and `kpp_pick_kernel` (one workgroup: scan of the block sums, scan inside the block, copy of
// the frame).  C3 (1 M x 10, k = 500): ~1000 launches, 48 GB of traffic, ~15 ms -- a start-up cost of the operator
// API's fit, not of the bench step (which keeps its seeded stratified draw, as round 1 defined the workload).
#include "common.h"

namespace {

constexpr int kBlock = 1024;   // frames per block sum

__device__ __forceinline__ unsigned long long kpp_hash(unsigned long long seed, unsigned long long j) {
    unsigned long long h = (seed ^ 0x6B2B2B6B6D65616Eull) + 0x9E3779B97F4A7C15ull * (j + 1);   // splitmix64
    h = (h ^ (h >> 30)) * 0xBF58476D1CE4E5B9ull;
    h = (h ^ (h >> 27)) * 0x94D049BB133111EBull;
    return h ^ (h >> 31);
}

// 2^e with n_total * (4 d absmax^2) * 2^e < 2^62 (frexp: the same exponent on every platform)
__device__ __forceinline__ double kpp_scale(double absmax, double n_total, int d) {
    double v = (n_total * 4.0 * (double)d) * absmax * absmax;
    if (!(v > 0.0)) v = 1.0;
    int ex;
    (void)frexp(v, &ex);   // v = m 2^ex, 0.5 <= m < 1
    int e = 61 - ex;
    e = e > 60 ? 60 : (e < -900 ? -900 : e);
    return ldexp(1.0, e);
}

template <typename T>
__device__ __forceinline__ double kpp_coord(const T* __restrict__ x, int64_t t, int64_t ld, int f,
                                            const double* __restrict__ mean, const double* __restrict__ stdv) {
    double v = (double)x[t * ld + f];
    if (mean) v = (v - mean[f]) / stdv[f];
    return v;
}

template <typename T>
__global__ void kpp_first_kernel(const T* __restrict__ x, int64_t n, int d, int64_t ld, const double* __restrict__ mean,
                                 const double* __restrict__ stdv, unsigned long long seed, double* __restrict__ centers,
                                 long long* __restrict__ picked) {
    const int64_t t = (int64_t)__umul64hi(kpp_hash(seed, 0), (unsigned long long)n);
    for (int f = threadIdx.x; f < d; f += blockDim.x) centers[f] = kpp_coord(x, t, ld, f, mean, stdv);
    if (threadIdx.x == 0) picked[0] = t;
}

// running minimum of D^2 against centre `j_new` and the block sums of the integer weights
template <typename T>
__global__ __launch_bounds__(kBlock) void kpp_update_kernel(const T* __restrict__ x, int64_t n, int d, int64_t ld,
                                                           const double* __restrict__ mean, const double* __restrict__ stdv,
                                                           const double* __restrict__ centers, int j_new, int first,
                                                           const double* __restrict__ state, double n_total,
                                                           double* __restrict__ mind, long long* __restrict__ block_sums) {
    __shared__ long long red[kBlock / 64];
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const double scale = kpp_scale(state[2], n_total, d);
    long long w = 0;
    if (t < n) {
        const double* c = centers + (size_t)j_new * d;
        double a = 0.0;
        for (int f = 0; f < d; ++f) {
            const double diff = __dsub_rn(kpp_coord(x, t, ld, f, mean, stdv), c[f]);
            a = __dadd_rn(a, __dmul_rn(diff, diff));
        }
        double m = first ? a : mind[t];
        if (!first && a < m) m = a;
        if (!(m == m)) m = 0.0;            // NaN coordinates: weight 0, never drawn
        mind[t] = m;
        w = __double2ll_rn(m * scale);
    }
    for (int off = 32; off > 0; off >>= 1) w += __shfl_down(w, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long s = 0;
        for (int i = 0; i < kBlock / 64; ++i) s += red[i];
        block_sums[blockIdx.x] = s;
    }
}

// exclusive scan of 1024 integers in the LDS (Hillis-Steele on two buffers); returns the total
__device__ __forceinline__ long long kpp_scan(long long v, long long* buf0, long long* buf1, long long* excl) {
    const int i = threadIdx.x;
    buf0[i] = v;
    __syncthreads();
    long long* src = buf0;
    long long* dst = buf1;
    for (int off = 1; off < kBlock; off <<= 1) {
        dst[i] = src[i] + (i >= off ? src[i - off] : 0);
        __syncthreads();
        long long* tmp = src;
        src = dst;
        dst = tmp;
    }
    *excl = src[i] - v;
    const long long total = src[kBlock - 1];
    __syncthreads();
    return total;
}

// one workgroup: draw frame t with cumsum(w)[t] > r, r = floor(h W / 2^64), and make it centre j
template <typename T>
__global__ __launch_bounds__(kBlock) void kpp_pick_kernel(const T* __restrict__ x, int64_t n, int d, int64_t ld,
                                                         const double* __restrict__ mean, const double* __restrict__ stdv,
                                                         const long long* __restrict__ block_sums, int64_t n_blocks,
                                                         const double* __restrict__ mind, const double* __restrict__ state,
                                                         double n_total, unsigned long long seed, int j,
                                                         double* __restrict__ centers, long long* __restrict__ picked) {
    __shared__ long long b0[kBlock], b1[kBlock];
    __shared__ long long sh_block, sh_rest, sh_t;
    const int i = threadIdx.x;
    const unsigned long long h = kpp_hash(seed, (unsigned long long)j);
    // ---- level 1: chunks of consecutive block sums per thread, scanned
    const int64_t per = (n_blocks + kBlock - 1) / kBlock;
    const int64_t lo = (int64_t)i * per, hi = lo + per < n_blocks ? lo + per : n_blocks;
    long long mine = 0;
    for (int64_t b = lo; b < hi; ++b) mine += block_sums[b];
    long long excl;
    const long long W = kpp_scan(mine, b0, b1, &excl);
    if (i == 0) {
        sh_block = -1;
        sh_t = -1;
    }
    __syncthreads();
    if (W <= 0) {
        if (i == 0) sh_t = (long long)__umul64hi(h, (unsigned long long)n);
    } else {
        const long long r = (long long)__umul64hi(h, (unsigned long long)W);
        if (mine > 0 && excl <= r && r < excl + mine) {      // exactly one thread: its chunk holds the draw
            long long acc = excl;
            for (int64_t b = lo; b < hi; ++b) {
                const long long s = block_sums[b];
                if (r < acc + s) {
                    sh_block = b;
                    sh_rest = r - acc;
                    break;
                }
                acc += s;
            }
        }
    }
    __syncthreads();
    if (W > 0) {
        // ---- level 2: the weights of that block, scanned
        const int64_t t = sh_block * kBlock + i;
        const double scale = kpp_scale(state[2], n_total, d);
        const long long w = t < n ? __double2ll_rn(mind[t] * scale) : 0;
        long long ex2;
        (void)kpp_scan(w, b0, b1, &ex2);
        if (w > 0 && ex2 <= sh_rest && sh_rest < ex2 + w) sh_t = t;
        __syncthreads();
    }
    const int64_t tp = sh_t;
    for (int f = i; f < d; f += kBlock) centers[(size_t)j * d + f] = kpp_coord(x, tp, ld, f, mean, stdv);
    if (i == 0) picked[j] = tp;
}

template <typename T>
msm_status run_plusplus(msm_ctx* ctx, const T* x, int64_t n, int d, int64_t ld, const double* mean, const double* stdv,
                        int k, uint64_t seed, double n_total, double* centers, const double* state, int64_t* picked) {
    const int64_t nb = (n + kBlock - 1) / kBlock;
    const size_t mind_bytes = ((size_t)n * sizeof(double) + 15) & ~(size_t)15;
    const size_t sums_bytes = ((size_t)nb * sizeof(long long) + 15) & ~(size_t)15;
    msm_status rs = msm_reserve_scratch(ctx, mind_bytes + sums_bytes + (picked ? 0 : (size_t)k * sizeof(long long)));
    if (rs != MSM_OK) return rs;
    double* mind = (double*)ctx->scratch;
    long long* sums = (long long*)((char*)ctx->scratch + mind_bytes);
    long long* pk = picked ? (long long*)picked : (long long*)((char*)ctx->scratch + mind_bytes + sums_bytes);
    hipLaunchKernelGGL(kpp_first_kernel<T>, dim3(1), dim3(256), 0, ctx->stream, x, n, d, ld, mean, stdv,
                       (unsigned long long)seed, centers, pk);
    MSM_CHECK_LAUNCH(ctx);
    for (int j = 1; j < k; ++j) {
        hipLaunchKernelGGL(kpp_update_kernel<T>, dim3((unsigned)nb), dim3(kBlock), 0, ctx->stream, x, n, d, ld, mean, stdv,
                           (const double*)centers, j - 1, j == 1 ? 1 : 0, state, n_total, mind, sums);
        hipLaunchKernelGGL(kpp_pick_kernel<T>, dim3(1), dim3(kBlock), 0, ctx->stream, x, n, d, ld, mean, stdv,
                           (const long long*)sums, nb, (const double*)mind, state, n_total, (unsigned long long)seed, j,
                           centers, pk);
    }
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

}  // namespace

extern "C" msm_status msm_kmeans_init_plusplus(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int d, int64_t ld,
                                               const double* d_mean, const double* d_std, int k, uint64_t seed,
                                               double n_total, double* d_centers, const double* d_state,
                                               int64_t* d_picked) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 1 && d >= 1 && k >= 1 && ld >= d, "msm_kmeans_init_plusplus: bad shape");
    MSM_REQUIRE(ctx, n >= k, "msm_kmeans_init_plusplus: fewer frames (%lld) than centres (%d)", (long long)n, k);
    MSM_REQUIRE(ctx, (d_mean == nullptr) == (d_std == nullptr), "msm_kmeans_init_plusplus: mean/std must come together");
    MSM_REQUIRE(ctx, d_x && d_centers && d_state, "msm_kmeans_init_plusplus: NULL pointer");
    MSM_REQUIRE(ctx, dtype == MSM_F32 || dtype == MSM_F64, "msm_kmeans_init_plusplus: bad dtype");
    MSM_REQUIRE(ctx, n_total >= (double)n, "msm_kmeans_init_plusplus: n_total < n");
    if (dtype == MSM_F32)
        return run_plusplus<float>(ctx, (const float*)d_x, n, d, ld, d_mean, d_std, k, seed, n_total, d_centers, d_state, d_picked);
    return run_plusplus<double>(ctx, (const double*)d_x, n, d, ld, d_mean, d_std, k, seed, n_total, d_centers, d_state, d_picked);
}
