// Reversible maximum-likelihood transition matrix.
//
// Reference: _fit_msm_deeptime (S/markov_state_model/_msm_utils.py:210-262) and the ITS / lag
// selectors (S/markov_state_model/ck_its_selector.py:395-401) call deeptime's
// MaximumLikelihoodMSM(reversible=True).  deeptime 0.4.5 is absent here (parity unpinned); this
// restates its published dense estimator (Prinz et al., J. Chem. Phys. 134, 174105 (2011), eq. 42;
// Trendelkamp-Schroer et al., J. Chem. Phys. 143, 174101 (2015), algorithm 1): with c_i the row
// sums of C, the symmetric flux matrix x_ij = pi_i T_ij is the fixed point of
//     x_ij <- (c_ij + c_ji) / (c_i / x_i + c_j / x_j),      x_i = sum_j x_ij,
// iterated on the row sums alone (the update of x_i only needs the vector x), renormalised to
// sum 1 each step, until max_i |x_i - x_i'| / ((x_i + x_i') / 2) <= maxerr; then
// T_ij = x_ij / x_i and pi = x.
//
// One launch per iteration: the state is a k-vector, every workgroup re-derives the
// normalisation of the previous iterate in a fixed order (bit-identical across workgroups) and
// produces the unnormalised new row sums of its rows, one wave per row.
#include "common.h"

namespace {

constexpr int kRT = 256;
constexpr int kRowsPerBlock = kRT / 64;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// fixed-order sum of x[0..n) by one workgroup (identical in every workgroup)
__device__ double block_total(const double* __restrict__ x, int n, double* sh) {
    double part = 0.0;
    for (int i = threadIdx.x; i < n; i += kRT) part += x[i];
    sh[threadIdx.x] = part;
    __syncthreads();
    for (int s = kRT / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    const double tot = sh[0];
    __syncthreads();
    return tot;
}

__global__ __launch_bounds__(kRT) void symmetrise_kernel(const double* __restrict__ C, int n, int ld,
                                                         double* __restrict__ C2, double* __restrict__ c,
                                                         double* __restrict__ x0) {
    // C2 = C + C' (packed n x n, read once with a stride, then every iteration streams rows);
    // c_i = sum_j C_ij;  x0_i = sum_j C2_ij  (start: X = C + C', normalised by the iteration)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * kRowsPerBlock + wave;
    if (i >= n) return;
    double a = 0.0, b = 0.0;
    for (int j = lane; j < n; j += 64) {
        const double cij = C[(size_t)i * ld + j];
        const double s = cij + C[(size_t)j * ld + i];
        C2[(size_t)i * n + j] = s;
        a += cij;
        b += s;
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0) { c[i] = a; x0[i] = b; }
}

__global__ __launch_bounds__(kRT) void iterate_kernel(const double* __restrict__ C2, int n,
                                                      const double* __restrict__ c,
                                                      const double* __restrict__ x_prev,
                                                      double* __restrict__ x_new) {
    extern __shared__ double v[];  // c_j / x_j of the normalised previous iterate
    __shared__ double red[kRT];
    const double tot = block_total(x_prev, n, red);
    for (int j = threadIdx.x; j < n; j += kRT) {
        const double xj = x_prev[j] / tot;
        v[j] = xj > 0.0 ? c[j] / xj : 0.0;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * kRowsPerBlock + wave;
    if (i >= n) return;
    const double vi = v[i];
    double acc = 0.0;
    for (int j = lane; j < n; j += 64) {
        const double c2 = C2[(size_t)i * n + j];
        const double den = vi + v[j];
        if (c2 > 0.0 && den > 0.0) acc += c2 / den;
    }
    acc = wave_sum(acc);
    if (lane == 0) x_new[i] = acc;
}

// err = max_i |a_i - b_i| / ((a_i + b_i) / 2) on the normalised vectors
__global__ __launch_bounds__(kRT) void error_kernel(const double* __restrict__ xa, const double* __restrict__ xb, int n,
                                                    double* __restrict__ err) {
    __shared__ double red[kRT];
    const double ta = block_total(xa, n, red), tb = block_total(xb, n, red);
    double worst = 0.0;
    for (int i = threadIdx.x; i < n; i += kRT) {
        const double a = xa[i] / ta, b = xb[i] / tb;
        const double mid = 0.5 * (a + b);
        if (mid > 0.0) worst = fmax(worst, fabs(a - b) / mid);
        if (!(a == a) || !(b == b)) worst = INFINITY;
    }
    red[threadIdx.x] = worst;
    __syncthreads();
    for (int s = kRT / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) *err = red[0];
}

__global__ __launch_bounds__(kRT) void finish_kernel(const double* __restrict__ C2, int n,
                                                     const double* __restrict__ c, const double* __restrict__ x,
                                                     double* __restrict__ T, int ldt, double* __restrict__ pi) {
    extern __shared__ double v[];
    __shared__ double red[kRT];
    const double tot = block_total(x, n, red);
    for (int j = threadIdx.x; j < n; j += kRT) {
        const double xj = x[j] / tot;
        v[j] = xj > 0.0 ? c[j] / xj : 0.0;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * kRowsPerBlock + wave;
    if (i >= n) return;
    const double vi = v[i];
    double acc = 0.0;
    for (int j = lane; j < n; j += 64) {
        const double c2 = C2[(size_t)i * n + j];
        const double den = vi + v[j];
        const double f = (c2 > 0.0 && den > 0.0) ? c2 / den : 0.0;
        T[(size_t)i * ldt + j] = f;
        acc += f;
    }
    acc = wave_sum(acc);
    for (int j = lane; j < n; j += 64) {
        // a state without any flux keeps a self-loop (cannot happen on a connected count matrix)
        T[(size_t)i * ldt + j] = acc > 0.0 ? T[(size_t)i * ldt + j] / acc : (j == i ? 1.0 : 0.0);
    }
    if (lane == 0 && pi) pi[i] = x[i] / tot;
}

}  // namespace

extern "C" {

msm_status msm_reversible_mle(msm_ctx* ctx, const double* d_counts, int n, int ld, double maxerr, int maxiter,
                              double* d_T, int ldt, double* d_pi, int* h_iterations, double* h_err) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, d_counts && d_T, "msm_reversible_mle: null pointer");
    MSM_REQUIRE(ctx, n >= 1 && ld >= n && ldt >= n, "msm_reversible_mle: bad shape");
    MSM_REQUIRE(ctx, maxerr > 0.0 && maxiter >= 1, "msm_reversible_mle: need maxerr > 0 and maxiter >= 1");
    MSM_REQUIRE(ctx, (size_t)n * sizeof(double) <= 96 * 1024, "msm_reversible_mle: n = %d exceeds the LDS table", n);
    if (ctx->capturing) return msm_fail(ctx, MSM_ERR_UNSUPPORTED, "msm_reversible_mle polls the host: not capturable");
    msm_status rs = msm_reserve_scratch(ctx, ((size_t)n * n + 3 * (size_t)n + 2) * sizeof(double));
    if (rs != MSM_OK) return rs;
    double* C2 = (double*)ctx->scratch;
    double* c = C2 + (size_t)n * n;
    double* xa = c + n;
    double* xb = xa + n;
    double* d_err = xb + n;
    const size_t lds = (size_t)n * sizeof(double);
    if (lds > 48 * 1024) {
        MSM_HIP(ctx, hipFuncSetAttribute((const void*)iterate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MSM_HIP(ctx, hipFuncSetAttribute((const void*)finish_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const dim3 grid((unsigned)msm_ceil_div(n, kRowsPerBlock));
    symmetrise_kernel<<<grid, kRT, 0, ctx->stream>>>(d_counts, n, ld, C2, c, xa);
    MSM_CHECK_LAUNCH(ctx);
    int it = 0;
    double err = INFINITY;
    int chunk = 32;   // iterations between convergence checks (grows: late iterations change little)
    while (it < maxiter && err > maxerr) {
        const int todo = std::min(chunk, maxiter - it);
        for (int q = 0; q < todo; ++q) {
            iterate_kernel<<<grid, kRT, lds, ctx->stream>>>(C2, n, c, xa, xb);
            std::swap(xa, xb);
        }
        MSM_CHECK_LAUNCH(ctx);
        it += todo;
        error_kernel<<<1, kRT, 0, ctx->stream>>>(xa, xb, n, d_err);   // last iterate against the one before
        MSM_HIP(ctx, hipMemcpyAsync(&err, d_err, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        MSM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (!(err <= 4.0))   // the measure is bounded by 2; error_kernel reports inf for NaN iterates
            return msm_fail(ctx, MSM_ERR_INVALID, "msm_reversible_mle: the iteration produced NaN (negative counts?)");
        if (chunk < 1024) chunk *= 2;
    }
    finish_kernel<<<grid, kRT, lds, ctx->stream>>>(C2, n, c, xa, d_T, ldt, d_pi);
    MSM_CHECK_LAUNCH(ctx);
    if (h_iterations) *h_iterations = it;
    if (h_err) *h_err = err;
    return MSM_OK;
}

}  // extern "C"
