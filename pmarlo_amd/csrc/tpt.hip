// Dense solves on the k x k transition matrix: committors, reactive flux, macrostate lumping and
// mean first-passage times  (SURVEY.md section 8f rank 4).
//
// Reference: TPTMixin (S/markov_state_model/_tpt.py:39-160, 255-347) delegates to deeptime 0.4.5
// (absent here; its published dense algorithm is restated: committor systems with identity rows
// on source / sink, f_ij = pi_i q-_i T_ij q+_j, net flux, total flux, rate, mfpt);
// lump_micro_to_macro_T / compute_macro_populations / compute_macro_mfpt
// (S/markov_state_model/_msm_utils.py:103-160).
//
// msm_solve_f64 is Gaussian elimination with partial pivoting by ONE workgroup (right-looking:
// per column a pivot search, a row swap, a scaling and a rank-1 update of the trailing block and
// of the right-hand sides, each fully parallel); fixed arithmetic order -> reproducible.
#include "common.h"

namespace {

constexpr int kLT = 1024;

struct PivotShared {
    double val[kLT / 64];
    int idx[kLT / 64];
    int piv;
    int singular;
};

// A (n x n, lda) is overwritten by its LU factors, B (n x nrhs, ldb) by the solution.
__global__ __launch_bounds__(kLT) void solve_kernel(double* __restrict__ A, int64_t lda, double* __restrict__ B,
                                                    int64_t ldb, int n, int nrhs, int* __restrict__ info) {
    __shared__ PivotShared sh;
    const int tid = threadIdx.x;
    if (tid == 0) sh.singular = 0;
    __syncthreads();
    for (int k = 0; k < n; ++k) {
        // pivot: first row of maximal |A[i][k]|, i >= k
        double best = -1.0;
        int bi = k;
        for (int i = k + tid; i < n; i += kLT) {
            const double v = fabs(A[(size_t)i * lda + k]);
            if (v > best) { best = v; bi = i; }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_down(best, off, 64);
            const int oi = __shfl_down(bi, off, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if ((tid & 63) == 0) { sh.val[tid >> 6] = best; sh.idx[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < kLT / 64; ++w)
                if (sh.val[w] > best || (sh.val[w] == best && sh.idx[w] < bi)) { best = sh.val[w]; bi = sh.idx[w]; }
            sh.piv = bi;
            if (!(best > 0.0)) sh.singular = k + 1;
        }
        __syncthreads();
        if (sh.singular) break;
        const int p = sh.piv;
        if (p != k) {
            for (int j = tid; j < n + nrhs; j += kLT) {
                double* a = j < n ? &A[(size_t)k * lda + j] : &B[(size_t)k * ldb + (j - n)];
                double* b = j < n ? &A[(size_t)p * lda + j] : &B[(size_t)p * ldb + (j - n)];
                const double t = *a; *a = *b; *b = t;
            }
            __syncthreads();
        }
        const double pivot = A[(size_t)k * lda + k];
        for (int i = k + 1 + tid; i < n; i += kLT) A[(size_t)i * lda + k] /= pivot;
        __syncthreads();
        const int rows = n - k - 1, cols = n - k - 1 + nrhs;
        for (int64_t e = tid; e < (int64_t)rows * cols; e += kLT) {
            const int a = (int)(e / cols), c = (int)(e - (int64_t)a * cols);
            const int i = k + 1 + a;
            const double l = A[(size_t)i * lda + k];
            if (c < n - k - 1) {
                const int j = k + 1 + c;
                A[(size_t)i * lda + j] = fma(-l, A[(size_t)k * lda + j], A[(size_t)i * lda + j]);
            } else {
                const int j = c - (n - k - 1);
                B[(size_t)i * ldb + j] = fma(-l, B[(size_t)k * ldb + j], B[(size_t)i * ldb + j]);
            }
        }
        __syncthreads();
    }
    if (sh.singular) {
        if (tid == 0) *info = sh.singular;
        return;
    }
    // back substitution, column-oriented: x_k = b_k / u_kk, then b_i -= u_ik x_k for i < k
    for (int k = n - 1; k >= 0; --k) {
        const double ukk = A[(size_t)k * lda + k];
        for (int j = tid; j < nrhs; j += kLT) B[(size_t)k * ldb + j] /= ukk;
        __syncthreads();
        for (int64_t e = tid; e < (int64_t)k * nrhs; e += kLT) {
            const int i = (int)(e / nrhs), j = (int)(e - (int64_t)i * nrhs);
            B[(size_t)i * ldb + j] = fma(-A[(size_t)i * lda + k], B[(size_t)k * ldb + j], B[(size_t)i * ldb + j]);
        }
        __syncthreads();
    }
    if (tid == 0) *info = 0;
}

// committor systems.  role[i]: 0 intermediate, 1 source (A), 2 sink (B).
//   forward : W = T - I on intermediate rows, identity rows on A and B, r = 1 on B
//   backward: W_ij = pi_j T_ji / pi_i - delta_ij on intermediate rows, r = 1 on A
__global__ void committor_system_kernel(const double* __restrict__ T, int64_t ldt, const double* __restrict__ pi,
                                        const int* __restrict__ role, int n, int backward, double* __restrict__ W,
                                        double* __restrict__ r) {
    const int i = blockIdx.x;
    const int ro = role[i];
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        double v;
        if (ro != 0) v = i == j ? 1.0 : 0.0;
        else if (!backward) v = T[(size_t)i * ldt + j] - (i == j ? 1.0 : 0.0);
        else v = pi[j] * T[(size_t)j * ldt + i] / pi[i] - (i == j ? 1.0 : 0.0);
        W[(size_t)i * n + j] = v;
    }
    if (threadIdx.x == 0) r[i] = (backward ? ro == 1 : ro == 2) ? 1.0 : 0.0;
}

// gross flux f_ij = pi_i q-_i T_ij q+_j (i != j), net flux max(0, f_ij - f_ji)
__global__ void flux_kernel(const double* __restrict__ T, int64_t ldt, const double* __restrict__ pi,
                            const double* __restrict__ qm, const double* __restrict__ qp, int n, double* __restrict__ gross,
                            double* __restrict__ net) {
    const int i = blockIdx.x;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const double fij = i == j ? 0.0 : pi[i] * qm[i] * T[(size_t)i * ldt + j] * qp[j];
        const double fji = i == j ? 0.0 : pi[j] * qm[j] * T[(size_t)j * ldt + i] * qp[i];
        gross[(size_t)i * n + j] = fij;
        net[(size_t)i * n + j] = fmax(0.0, fij - fji);
    }
}

// out = {total flux = sum_{i in A, j not in A} f_ij,  sum_i pi_i q-_i,  rate,  mfpt}
__global__ __launch_bounds__(1024) void flux_totals_kernel(const double* __restrict__ gross, const double* __restrict__ pi,
                                                           const double* __restrict__ qm, const int* __restrict__ role,
                                                           int n, double* __restrict__ out) {
    __shared__ double red[16];
    __shared__ double tot[2];
    double f = 0.0, z = 0.0;
    for (int64_t e = threadIdx.x; e < (int64_t)n * n; e += blockDim.x) {
        const int i = (int)(e / n), j = (int)(e - (int64_t)i * n);
        if (role[i] == 1 && role[j] != 1) f += gross[e];
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) z = fma(pi[i], qm[i], z);
    for (int which = 0; which < 2; ++which) {
        double v = which == 0 ? f : z;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
            tot[which] = t;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = tot[0];
        out[1] = tot[1];
        out[2] = tot[0] / tot[1];
        out[3] = tot[1] / tot[0];
    }
}

// lumping: rowflux[i][B] = pi_i sum_{j in B} T_ij (one workgroup per micro row, fixed order), then
// F[A][B] = sum_{i in A} rowflux[i][B] in ascending i; T_macro = F / rowsum (zero rows stay zero);
// pi_macro[A] = sum_{i in A} pi_i, renormalised.
__global__ __launch_bounds__(256) void lump_rows_kernel(const double* __restrict__ T, int64_t ldt, const double* __restrict__ pi,
                                                        const int* __restrict__ macro, int n, int n_macro,
                                                        double* __restrict__ rowflux) {
    extern __shared__ double part[];  // [waves][n_macro]
    const int i = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    for (int b = lane; b < n_macro; b += 64) part[wave * n_macro + b] = 0.0;
    __syncthreads();
    // each wave walks its columns in ascending order; lanes of a wave own disjoint macro bins per pass
    for (int B = 0; B < n_macro; ++B) {
        double acc = 0.0;
        for (int j = wave * 64 + lane; j < n; j += waves * 64)
            if (macro[j] == B) acc += T[(size_t)i * ldt + j];
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if (lane == 0) part[wave * n_macro + B] = acc;
    }
    __syncthreads();
    for (int B = threadIdx.x; B < n_macro; B += blockDim.x) {
        double t = 0.0;
        for (int w = 0; w < waves; ++w) t += part[w * n_macro + B];
        rowflux[(size_t)i * n_macro + B] = pi[i] * t;
    }
}

__global__ void lump_finish_kernel(const double* __restrict__ rowflux, const double* __restrict__ pi,
                                   const int* __restrict__ macro, int n, int n_macro, double* __restrict__ T_macro,
                                   double* __restrict__ pi_macro) {
    __shared__ double psum;
    const int A = blockIdx.x;
    for (int B = threadIdx.x; B < n_macro; B += blockDim.x) {
        double f = 0.0;
        for (int i = 0; i < n; ++i)
            if (macro[i] == A) f += rowflux[(size_t)i * n_macro + B];
        T_macro[(size_t)A * n_macro + B] = f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double rs = 0.0;
        for (int B = 0; B < n_macro; ++B) rs += T_macro[(size_t)A * n_macro + B];
        psum = rs == 0.0 ? 1.0 : rs;
        double p = 0.0;
        for (int i = 0; i < n; ++i)
            if (macro[i] == A) p += pi[i];
        pi_macro[A] = p;
    }
    __syncthreads();
    for (int B = threadIdx.x; B < n_macro; B += blockDim.x) T_macro[(size_t)A * n_macro + B] /= psum;
}

__global__ void normalise_vec_kernel(double* v, int n) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += v[i];
    if (s > 0.0)
        for (int i = 0; i < n; ++i) v[i] /= s;
}

// MFPT to target j: (I - Q) t = 1 with row / column j removed; systems for all targets side by side
__global__ void mfpt_system_kernel(const double* __restrict__ T, int64_t ldt, int n, double* __restrict__ A,
                                   double* __restrict__ b) {
    const int target = blockIdx.x, m = n - 1;
    double* At = A + (size_t)target * m * m;
    for (int e = threadIdx.x; e < m * m; e += blockDim.x) {
        const int a = e / m, c = e - a * m;
        const int i = a + (a >= target), j = c + (c >= target);
        At[e] = (a == c ? 1.0 : 0.0) - T[(size_t)i * ldt + j];
    }
    for (int a = threadIdx.x; a < m; a += blockDim.x) b[(size_t)target * m + a] = 1.0;
}

}  // namespace

extern "C" {

msm_status msm_solve_f64(msm_ctx* ctx, int n, int nrhs, double* d_A, int64_t lda, double* d_B, int64_t ldb,
                         int32_t* d_info) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 1 && nrhs >= 1 && lda >= n && ldb >= nrhs, "msm_solve_f64: bad shape");
    MSM_REQUIRE(ctx, d_A && d_B && d_info, "msm_solve_f64: NULL pointer");
    hipLaunchKernelGGL(solve_kernel, dim3(1), dim3(kLT), 0, ctx->stream, d_A, lda, d_B, ldb, n, nrhs, d_info);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_reactive_flux(msm_ctx* ctx, const double* d_T, int64_t ldt, const double* d_pi, const int32_t* d_role,
                             int n, double* d_qplus, double* d_qminus, double* d_gross, double* d_net, double* d_totals,
                             int32_t* d_info) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 2 && ldt >= n, "msm_reactive_flux: bad shape");
    MSM_REQUIRE(ctx, d_T && d_pi && d_role && d_qplus && d_qminus && d_info, "msm_reactive_flux: NULL pointer");
    MSM_REQUIRE(ctx, (d_gross == nullptr) == (d_net == nullptr) && (d_totals == nullptr || d_gross != nullptr),
                "msm_reactive_flux: gross, net (and totals) come together");
    const size_t mat = (size_t)n * n;
    msm_status rs = msm_reserve_scratch(ctx, mat * sizeof(double));
    if (rs != MSM_OK) return rs;
    double* W = (double*)ctx->scratch;
    for (int backward = 0; backward < 2; ++backward) {
        double* q = backward ? d_qminus : d_qplus;
        hipLaunchKernelGGL(committor_system_kernel, dim3(n), dim3(256), 0, ctx->stream, d_T, ldt, d_pi, d_role, n,
                           backward, W, q);
        hipLaunchKernelGGL(solve_kernel, dim3(1), dim3(kLT), 0, ctx->stream, W, (int64_t)n, q, (int64_t)1, n, 1,
                           d_info + backward);
        MSM_CHECK_LAUNCH(ctx);
    }
    if (d_gross) {
        hipLaunchKernelGGL(flux_kernel, dim3(n), dim3(256), 0, ctx->stream, d_T, ldt, d_pi, d_qminus, d_qplus, n, d_gross,
                           d_net);
        if (d_totals)
            hipLaunchKernelGGL(flux_totals_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_gross, d_pi, d_qminus, d_role, n,
                               d_totals);
        MSM_CHECK_LAUNCH(ctx);
    }
    return MSM_OK;
}

msm_status msm_lump_macro(msm_ctx* ctx, const double* d_T, int64_t ldt, const double* d_pi, const int32_t* d_macro, int n,
                          int n_macro, double* d_T_macro, double* d_pi_macro) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 1 && n_macro >= 1 && n_macro <= 4096 && ldt >= n, "msm_lump_macro: bad shape");
    MSM_REQUIRE(ctx, d_T && d_pi && d_macro && d_T_macro && d_pi_macro, "msm_lump_macro: NULL pointer");
    msm_status rs = msm_reserve_scratch(ctx, (size_t)n * n_macro * sizeof(double));
    if (rs != MSM_OK) return rs;
    double* rowflux = (double*)ctx->scratch;
    hipLaunchKernelGGL(lump_rows_kernel, dim3(n), dim3(256), (size_t)4 * n_macro * sizeof(double), ctx->stream, d_T, ldt,
                       d_pi, d_macro, n, n_macro, rowflux);
    hipLaunchKernelGGL(lump_finish_kernel, dim3(n_macro), dim3(64), 0, ctx->stream, rowflux, d_pi, d_macro, n, n_macro,
                       d_T_macro, d_pi_macro);
    hipLaunchKernelGGL(normalise_vec_kernel, dim3(1), dim3(64), 0, ctx->stream, d_pi_macro, n_macro);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_macro_mfpt(msm_ctx* ctx, const double* d_T, int64_t ldt, int n, double* d_mfpt, int32_t* d_info) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 2 && n <= 1024 && ldt >= n && d_T && d_mfpt && d_info, "msm_macro_mfpt: bad arguments");
    const int m = n - 1;
    msm_status rs = msm_reserve_scratch(ctx, ((size_t)n * m * m + (size_t)n * m) * sizeof(double));
    if (rs != MSM_OK) return rs;
    double* A = (double*)ctx->scratch;
    double* b = A + (size_t)n * m * m;
    hipLaunchKernelGGL(mfpt_system_kernel, dim3(n), dim3(256), 0, ctx->stream, d_T, ldt, n, A, b);
    for (int t = 0; t < n; ++t)
        hipLaunchKernelGGL(solve_kernel, dim3(1), dim3(kLT), 0, ctx->stream, A + (size_t)t * m * m, (int64_t)m,
                           b + (size_t)t * m, (int64_t)1, m, 1, d_info + t);
    MSM_CHECK_LAUNCH(ctx);
    // scatter: mfpt[i][t] = x_t[i - (i > t)], mfpt[t][t] = 0  (tiny: done with 2-D copies per target)
    MSM_HIP(ctx, hipMemsetAsync(d_mfpt, 0, (size_t)n * n * sizeof(double), ctx->stream));
    for (int t = 0; t < n; ++t) {
        if (t > 0)
            MSM_HIP(ctx, hipMemcpy2DAsync(d_mfpt + t, (size_t)n * sizeof(double), b + (size_t)t * m, sizeof(double),
                                          sizeof(double), (size_t)t, hipMemcpyDeviceToDevice, ctx->stream));
        if (t < n - 1)
            MSM_HIP(ctx, hipMemcpy2DAsync(d_mfpt + (size_t)(t + 1) * n + t, (size_t)n * sizeof(double),
                                          b + (size_t)t * m + t, sizeof(double), sizeof(double), (size_t)(n - 1 - t),
                                          hipMemcpyDeviceToDevice, ctx->stream));
    }
    return MSM_OK;
}

}  // extern "C"
