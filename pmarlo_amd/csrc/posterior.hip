// Posterior samples of the transition matrix for the implied-timescale confidence intervals.
//
// Reference: ITSMixin._its_compute_for_single_lag (S/markov_state_model/_its.py:272-357) draws
// n_samples matrices from deeptime's BayesianMSM and summarises their spectra
// (_summarize_its_stats :543-668).  deeptime's sampler is third-party C++ that is absent here and
// its random stream cannot be reproduced; this file samples the closed-form posterior of the
// estimator the engine uses (msm_transition_matrix mode 1: T = rownorm(C_active + alpha),
// reversible = False): row i ~ Dirichlet(C_active[i, :] + alpha), independent rows, so the
// posterior mean is exactly the point estimate.
//
// A Dirichlet row is a vector of gamma variates divided by its sum.  Gamma variates come from
// Marsaglia & Tsang's squeeze method (ACM TOMS 26, 2000) for shape >= 1 and the boost
// G(a) = G(a + 1) U^(1/a) below; everything is carried as logarithms, so that the alpha-only cells
// (shape 1e-3: U^1000) cannot underflow a whole row to 0/0.  Random numbers are Philox4x32-10
// (Salmon et al., SC'11) keyed by the seed with counter (column, row, sample, attempt): a cell's
// variate does not depend on the launch geometry, the batch it is drawn in, or any other cell.
#include "common.h"

namespace {

constexpr int kPT = 256;

struct Philox {
    uint32_t k0, k1;
    __device__ __forceinline__ void round(uint32_t (&c)[4], uint32_t a, uint32_t b) const {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ a;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ b;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    __device__ __forceinline__ void operator()(uint32_t (&c)[4]) const {
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            round(c, a, b);
            a += 0x9E3779B9u;
            b += 0xBB67AE85u;
        }
    }
};

// 64 random bits -> double in (0, 1): 53 bits, never 0 or 1
__device__ __forceinline__ double unit_open(uint32_t hi, uint32_t lo) {
    const uint64_t v = (((uint64_t)hi << 32) | lo) >> 11;
    return ((double)v + 0.5) * 1.1102230246251565e-16;  // 2^-53
}

// log of a Gamma(shape, 1) variate; cell identity = (col, row, sample)
__device__ double log_gamma_variate(const Philox& rng, double shape, uint32_t col, uint32_t row, uint32_t sample) {
    const bool boost = shape < 1.0;
    const double a = boost ? shape + 1.0 : shape;
    const double d = a - 1.0 / 3.0;
    const double c = 1.0 / sqrt(9.0 * d);
    double lg = 0.0;
    uint32_t attempt = 0;
    for (;;) {
        uint32_t r[4] = {col, row, sample, 2 * attempt};
        rng(r);
        const double u1 = unit_open(r[0], r[1]), u2 = unit_open(r[2], r[3]);
        uint32_t q[4] = {col, row, sample, 2 * attempt + 1};
        rng(q);
        const double u3 = unit_open(q[0], q[1]), u4 = unit_open(q[2], q[3]);
        ++attempt;
        const double x = sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);  // Box-Muller
        const double t = 1.0 + c * x;
        if (t <= 0.0) continue;
        const double v = t * t * t;
        const double x2 = x * x;
        const double lv = log(v);
        if (u3 < 1.0 - 0.0331 * x2 * x2 || log(u3) < 0.5 * x2 + d * (1.0 - v + lv) || attempt >= 64) {
            lg = log(d) + lv;
            if (boost) lg += log(u4) / shape;
            break;
        }
    }
    return lg;
}

__device__ __forceinline__ double block_reduce(double v, double* sh, bool is_max) {
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    for (int s = kPT / 2; s > 0; s >>= 1) {
        if (tid < s) sh[tid] = is_max ? fmax(sh[tid], sh[tid + s]) : sh[tid] + sh[tid + s];
        __syncthreads();
    }
    const double out = sh[0];
    __syncthreads();
    return out;
}

// one workgroup per (packed row, sample)
template <typename CT>
__global__ __launch_bounds__(kPT) void sample_rows_kernel(const CT* __restrict__ counts, int k,
                                                          const int32_t* __restrict__ active,
                                                          const int32_t* __restrict__ n_active, double alpha,
                                                          Philox rng, uint32_t first_sample,
                                                          double* __restrict__ T, int64_t t_stride, int ld) {
    __shared__ double sh[kPT];
    const int n = *n_active;
    const int row = blockIdx.x;
    if (row >= n) return;
    const uint32_t sample = first_sample + blockIdx.y;
    double* out = T + (int64_t)blockIdx.y * t_stride + (int64_t)row * ld;
    const CT* crow = counts + (int64_t)active[row] * k;
    double mx = -INFINITY;
    for (int j = threadIdx.x; j < n; j += kPT) {
        const double shape = (double)crow[active[j]] + alpha;
        double lg = -INFINITY;
        if (shape > 0.0) lg = log_gamma_variate(rng, shape, (uint32_t)j, (uint32_t)row, sample);
        out[j] = lg;
        mx = fmax(mx, lg);
    }
    mx = block_reduce(mx, sh, true);
    double sum = 0.0;
    for (int j = threadIdx.x; j < n; j += kPT) {
        const double e = mx == -INFINITY ? 0.0 : exp(out[j] - mx);
        out[j] = e;
        sum += e;
    }
    sum = block_reduce(sum, sh, false);
    // a row without any positive shape (alpha = 0 and no counts) stays a self-loop, as rownorm leaves it
    for (int j = threadIdx.x; j < n; j += kPT) out[j] = sum > 0.0 ? out[j] / sum : (j == row ? 1.0 : 0.0);
}

__global__ void philox_kat_kernel(Philox rng, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t* out) {
    uint32_t c[4] = {c0, c1, c2, c3};
    rng(c);
    for (int i = 0; i < 4; ++i) out[i] = c[i];
}

}  // namespace

extern "C" {

msm_status msm_sample_transition_matrices(msm_ctx* ctx, const void* d_counts, int counts_are_f64, int k,
                                          const int32_t* d_active, const int32_t* d_n_active, double alpha,
                                          uint64_t seed, int first_sample, int n_samples, double* d_T,
                                          int64_t t_stride, int ld) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, d_counts && d_active && d_n_active && d_T, "msm_sample_transition_matrices: null pointer");
    MSM_REQUIRE(ctx, k >= 1 && ld >= k && t_stride >= (int64_t)k * ld, "msm_sample_transition_matrices: bad shape");
    MSM_REQUIRE(ctx, n_samples >= 0 && n_samples <= 65535 && first_sample >= 0,
                "msm_sample_transition_matrices: 0 <= n_samples <= 65535 per call");
    MSM_REQUIRE(ctx, alpha >= 0.0, "msm_sample_transition_matrices: alpha must be >= 0");
    if (n_samples == 0) return MSM_OK;
    const Philox rng{(uint32_t)seed, (uint32_t)(seed >> 32)};
    const dim3 grid((unsigned)k, (unsigned)n_samples);
    if (counts_are_f64)
        sample_rows_kernel<double><<<grid, kPT, 0, ctx->stream>>>((const double*)d_counts, k, d_active, d_n_active, alpha,
                                                                   rng, (uint32_t)first_sample, d_T, t_stride, ld);
    else
        sample_rows_kernel<long long><<<grid, kPT, 0, ctx->stream>>>((const long long*)d_counts, k, d_active, d_n_active,
                                                                      alpha, rng, (uint32_t)first_sample, d_T, t_stride, ld);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_philox4x32(msm_ctx* ctx, uint64_t key, const uint32_t counter[4], uint32_t* d_out) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, counter && d_out, "msm_philox4x32: null pointer");
    philox_kat_kernel<<<1, 1, 0, ctx->stream>>>(Philox{(uint32_t)key, (uint32_t)(key >> 32)}, counter[0], counter[1],
                                                counter[2], counter[3], d_out);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

}  // extern "C"
