// Mean silhouette coefficient of a clustering (model selection for n_states = "auto").
//
// Reference: _auto_select_n_states (S/markov_state_model/clustering.py:156-233) scores k = 4..20 with
// sklearn.metrics.silhouette_score: s_i = (b_i - a_i) / max(a_i, b_i), a_i = mean Euclidean distance to
// the other members of i's cluster, b_i = smallest mean distance to another cluster, s_i = 0 for
// singleton clusters; the score is the mean of s_i.
//
// O(n^2 d) pair distances: the points arrive SORTED BY CLUSTER (offsets[k+1]), every thread owns one
// query point and walks the clusters in order, so the per-cluster distance sums sit in registers with
// static indices; the reference points stream through LDS tiles shared by the workgroup.
#include "common.h"

namespace {

constexpr int kST = 256;
constexpr int kMaxK = 32;
constexpr int kTile = 64;

struct SilOffsets { int64_t off[kMaxK + 1]; };

__global__ __launch_bounds__(kST) void silhouette_kernel(const double* __restrict__ x, int64_t n, int d, int64_t ld, int k,
                                                         SilOffsets so, double* __restrict__ s_out) {
    extern __shared__ double tile[];  // [kTile][d]
    const int64_t i = (int64_t)blockIdx.x * kST + threadIdx.x;
    const bool live = i < n;
    const double* xi = x + (live ? i : 0) * ld;
    double acc[kMaxK];
#pragma unroll
    for (int c = 0; c < kMaxK; ++c) acc[c] = 0.0;
#pragma unroll
    for (int c = 0; c < kMaxK; ++c) {
        if (c < k) {
            double a = 0.0;
            for (int64_t j0 = so.off[c]; j0 < so.off[c + 1]; j0 += kTile) {
                const int cnt = (int)min((int64_t)kTile, so.off[c + 1] - j0);
                __syncthreads();
                for (int e = threadIdx.x; e < cnt * d; e += kST) tile[e] = x[(j0 + e / d) * ld + e % d];
                __syncthreads();
                for (int j = 0; j < cnt; ++j) {
                    double d2 = 0.0;
                    const double* xj = tile + j * d;
                    for (int f = 0; f < d; ++f) {
                        const double df = xi[f] - xj[f];
                        d2 = fma(df, df, d2);
                    }
                    a += sqrt(d2);
                }
            }
            acc[c] = a;
        }
    }
    if (!live) return;
    int ci = 0;
#pragma unroll
    for (int c = 0; c < kMaxK; ++c)
        if (c < k && i >= so.off[c] && i < so.off[c + 1]) ci = c;
    double a_i = 0.0, b_i = __builtin_inf();
    int64_t n_ci = 1;
#pragma unroll
    for (int c = 0; c < kMaxK; ++c) {
        if (c < k) {
            const int64_t nc = so.off[c + 1] - so.off[c];
            if (c == ci) { n_ci = nc; a_i = nc > 1 ? acc[c] / (double)(nc - 1) : 0.0; }
            else if (nc > 0) b_i = fmin(b_i, acc[c] / (double)nc);
        }
    }
    double s = 0.0;
    if (n_ci > 1 && b_i < __builtin_inf()) {
        const double den = fmax(a_i, b_i);
        s = den > 0.0 ? (b_i - a_i) / den : 0.0;
    }
    s_out[i] = s;
}

__global__ __launch_bounds__(1024) void mean_kernel(const double* __restrict__ v, int64_t n, double* __restrict__ out) {
    __shared__ double red[16];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) acc += v[i];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red[w];
        out[0] = t / (double)n;
    }
}

}  // namespace

extern "C" {

msm_status msm_silhouette(msm_ctx* ctx, const double* d_x, int64_t n, int d, int64_t ld, const int64_t* h_offsets, int k,
                          double* d_samples, double* d_score) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 2 && d >= 1 && d <= 128 && ld >= d, "msm_silhouette: need n >= 2 and 1 <= d <= 128");
    MSM_REQUIRE(ctx, k >= 2 && k <= kMaxK, "msm_silhouette: need 2 <= k <= %d clusters", kMaxK);
    MSM_REQUIRE(ctx, d_x && h_offsets && d_samples && d_score, "msm_silhouette: NULL pointer");
    SilOffsets so;
    for (int c = 0; c <= kMaxK; ++c) so.off[c] = c <= k ? h_offsets[c] : h_offsets[k];
    MSM_REQUIRE(ctx, so.off[0] == 0 && so.off[k] == n, "msm_silhouette: offsets must run from 0 to n");
    for (int c = 0; c < k; ++c) MSM_REQUIRE(ctx, so.off[c + 1] >= so.off[c], "msm_silhouette: offsets must be non-decreasing");
    const int grid = (int)((n + kST - 1) / kST);
    hipLaunchKernelGGL(silhouette_kernel, dim3(grid), dim3(kST), (size_t)kTile * d * sizeof(double), ctx->stream, d_x, n, d,
                       ld, k, so, d_samples);
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_samples, n, d_score);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

}  // extern "C"
