// Chapman-Kolmogorov test on the device: P(tau)^f against P(f tau)  (SURVEY.md section 8f rank 2).
//
// Reference: ck_error (S/validation/ck_rule.py:36-47: RMS of matrix_power(P, f) - P_f),
// _multinomial_rms_se (:50-63), _ck_on_trajs (S/markov_state_model/ck_runner.py:155-176: MSE of the
// same difference).  The matrix power is a chain of fp64 matrix-core GEMMs; one
// v_mfma_f64_16x16x4_f64 per 16 x 16 output tile and 4 inner indices, i.e. every output element is
// the plain ascending-k FMA chain (bit-reproducible, restated in oracle/msm_oracle.c).
#include "common.h"

namespace {

typedef double v4f64 __attribute__((ext_vector_type(4)));

// C (m x n) = A (m x k) . B (k x n), row-major; one wave per 16 x 16 tile of C.
// A fragment: lane l holds A[row0 + (l & 15)][k0 + (l >> 4)]; B: B[k0 + (l >> 4)][col0 + (l & 15)];
// D: lane l, reg r = C[row0 + (l >> 4) + 4 r][col0 + (l & 15)].
__global__ __launch_bounds__(256) void gemm_f64_kernel(const double* __restrict__ A, int64_t lda,
                                                       const double* __restrict__ B, int64_t ldb,
                                                       double* __restrict__ C, int64_t ldc, int m, int n, int k) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles_n = (n + 15) / 16;
    const int tile = blockIdx.x * 4 + wave;
    const int tr = tile / tiles_n, tc = tile - tr * tiles_n;
    if (tr * 16 >= m) return;
    const int row0 = tr * 16, col0 = tc * 16;
    const int ar = min(row0 + (lane & 15), m - 1);    // clamped: rows / columns past the edge are
    const int bc = min(col0 + (lane & 15), n - 1);    // computed on valid data and never stored
    const int kk = lane >> 4;
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    const double* ap = A + (size_t)ar * lda + kk;
    const double* bp = B + (size_t)kk * ldb + bc;
    int k0 = 0;
    for (; k0 + 16 <= k; k0 += 16) {  // 8 loads in flight per lane
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = ap[k0 + 4 * u];
            b[u] = bp[(size_t)(k0 + 4 * u) * ldb];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
    }
    for (; k0 < k; k0 += 4) {
        const bool ok = k0 + kk < k;      // tail: inner indices past k contribute 0 * 0
        const double a = ok ? ap[k0] : 0.0;
        const double b = ok ? bp[(size_t)k0 * ldb] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    const int c = col0 + (lane & 15);
    if (c < n) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + (lane >> 4) + 4 * r;
            if (row < m) C[(size_t)row * ldc + c] = acc[r];
        }
    }
}

// fixed-order block reduction (same tree for every launch)
__device__ double block_sum_1024(double v, double* red) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;  // valid in thread 0
}

// out[0] = mean((P - Q)^2)
__global__ __launch_bounds__(1024) void mse_kernel(const double* __restrict__ P, int64_t ldp, const double* __restrict__ Q,
                                                   int64_t ldq, int n, double* __restrict__ out) {
    __shared__ double red[16];
    double acc = 0.0;
    for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
        const int i = e / n, j = e - i * n;
        const double d = P[(size_t)i * ldp + j] - Q[(size_t)i * ldq + j];
        acc = fma(d, d, acc);
    }
    const double t = block_sum_1024(acc, red);
    if (threadIdx.x == 0) out[0] = t / ((double)n * (double)n);
}

// out = { sum |P - Q|, sum |Q|, sum (P - Q)^2 } over n x m entries, fixed summation order
__global__ __launch_bounds__(1024) void diff_norms_kernel(const double* __restrict__ P, int64_t ldp,
                                                          const double* __restrict__ Q, int64_t ldq, int n, int m,
                                                          double* __restrict__ out) {
    __shared__ double red[16];
    double l1 = 0.0, ref = 0.0, l2 = 0.0;
    for (int64_t e = threadIdx.x; e < (int64_t)n * m; e += blockDim.x) {
        const int i = (int)(e / m), j = (int)(e - (int64_t)i * m);
        const double q = Q[(size_t)i * ldq + j];
        const double d = P[(size_t)i * ldp + j] - q;
        l1 += fabs(d);
        ref += fabs(q);
        l2 = fma(d, d, l2);
    }
    l1 = block_sum_1024(l1, red);
    __syncthreads();
    ref = block_sum_1024(ref, red);
    __syncthreads();
    l2 = block_sum_1024(l2, red);
    if (threadIdx.x == 0) { out[0] = l1; out[1] = ref; out[2] = l2; }
}

// out[0] = sqrt(mean_i( sum_j p_ij (1 - p_ij) / N_i / n )),  N_i <= 0 or non-finite -> 1
__global__ __launch_bounds__(1024) void multinomial_se_kernel(const double* __restrict__ P, int64_t ldp,
                                                              const double* __restrict__ rowcounts, int n,
                                                              double* __restrict__ out) {
    __shared__ double red[16];
    double acc = 0.0;
    for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
        const int i = e / n, j = e - i * n;
        double Ni = rowcounts[i];
        if (!(Ni > 0.0) || !(Ni < 1.0e300)) Ni = 1.0;
        const double p = P[(size_t)i * ldp + j];
        acc += p * (1.0 - p) / Ni;
    }
    const double t = block_sum_1024(acc, red);
    if (threadIdx.x == 0) out[0] = sqrt(t / ((double)n * (double)n));
}

msm_status launch_gemm(msm_ctx* ctx, const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc,
                       int m, int n, int k) {
    const int tiles = ((m + 15) / 16) * ((n + 15) / 16);
    hipLaunchKernelGGL(gemm_f64_kernel, dim3((tiles + 3) / 4), dim3(256), 0, ctx->stream, A, lda, B, ldb, C, ldc, m, n, k);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

}  // namespace

extern "C" {

msm_status msm_gemm_f64(msm_ctx* ctx, int m, int n, int k, const double* d_A, int64_t lda, const double* d_B,
                        int64_t ldb, double* d_C, int64_t ldc) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, m >= 0 && n >= 0 && k >= 0, "msm_gemm_f64: negative dimension");
    MSM_REQUIRE(ctx, lda >= k && ldb >= n && ldc >= n, "msm_gemm_f64: bad leading dimension");
    if (m == 0 || n == 0) return MSM_OK;
    MSM_REQUIRE(ctx, d_C && (k == 0 || (d_A && d_B)), "msm_gemm_f64: NULL pointer");
    MSM_REQUIRE(ctx, d_C != d_A && d_C != d_B, "msm_gemm_f64: the output must not alias an input");
    if (k == 0) {
        MSM_HIP(ctx, hipMemset2DAsync(d_C, (size_t)ldc * sizeof(double), 0, (size_t)n * sizeof(double), (size_t)m,
                                      ctx->stream));
        return MSM_OK;
    }
    return launch_gemm(ctx, d_A, lda, d_B, ldb, d_C, ldc, m, n, k);
}

msm_status msm_ck_test(msm_ctx* ctx, const double* d_T1, int64_t ld1, const double* d_Tk, int64_t tk_stride, int64_t ldk,
                       int n, const int32_t* h_factors, int n_factors, const double* d_rowcounts, int64_t rc_stride,
                       double* d_mse, double* d_noise) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 1 && ld1 >= n && ldk >= n && n_factors >= 0, "msm_ck_test: bad shape");
    if (n_factors == 0) return MSM_OK;
    MSM_REQUIRE(ctx, d_T1 && d_Tk && h_factors && d_mse, "msm_ck_test: NULL pointer");
    MSM_REQUIRE(ctx, (d_rowcounts == nullptr) == (d_noise == nullptr), "msm_ck_test: rowcounts and noise come together");
    int fmax = 0;
    for (int i = 0; i < n_factors; ++i) {
        MSM_REQUIRE(ctx, h_factors[i] >= 1, "msm_ck_test: factors must be >= 1 (got %d)", h_factors[i]);
        fmax = std::max(fmax, (int)h_factors[i]);
    }
    // powers P, P^2, ..., P^fmax by repeated right-multiplication (numpy's matrix_power squares
    // instead; the two agree to rounding, the oracle restates this chain)
    const size_t mat = (size_t)n * n;
    msm_status rs = msm_reserve_scratch(ctx, 2 * mat * sizeof(double));
    if (rs != MSM_OK) return rs;
    double* bufs[2] = {(double*)ctx->scratch, (double*)ctx->scratch + mat};
    const double* cur = d_T1;
    int64_t cur_ld = ld1;
    for (int f = 1; f <= fmax; ++f) {
        if (f > 1) {
            double* dst = bufs[f & 1];
            rs = launch_gemm(ctx, cur, cur_ld, d_T1, ld1, dst, n, n, n, n);
            if (rs != MSM_OK) return rs;
            cur = dst;
            cur_ld = n;
        }
        for (int i = 0; i < n_factors; ++i) {
            if (h_factors[i] != f) continue;
            const double* Tk = d_Tk + (size_t)i * tk_stride;
            hipLaunchKernelGGL(mse_kernel, dim3(1), dim3(1024), 0, ctx->stream, cur, cur_ld, Tk, ldk, n, d_mse + i);
            MSM_CHECK_LAUNCH(ctx);
            if (d_noise) {
                hipLaunchKernelGGL(multinomial_se_kernel, dim3(1), dim3(1024), 0, ctx->stream, Tk, ldk,
                                   d_rowcounts + (size_t)i * rc_stride, n, d_noise + i);
                MSM_CHECK_LAUNCH(ctx);
            }
        }
    }
    return MSM_OK;
}

msm_status msm_diff_norms(msm_ctx* ctx, const double* d_P, int64_t ldp, const double* d_Q, int64_t ldq, int n, int m,
                          double* d_out) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, d_P && d_Q && d_out, "msm_diff_norms: null pointer");
    MSM_REQUIRE(ctx, n >= 1 && m >= 1 && ldp >= m && ldq >= m, "msm_diff_norms: bad shape");
    hipLaunchKernelGGL(diff_norms_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_P, ldp, d_Q, ldq, n, m, d_out);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

}  // extern "C"
