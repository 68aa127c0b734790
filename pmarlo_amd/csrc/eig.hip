// Dense symmetric eigensolver (parallel cyclic Jacobi, one workgroup) and the
// TICA solve built on it.  The matrices are F x F with F <= 256: this stage is
// latency-bound, not bandwidth- or flop-bound, so the design
// goal is "stay on the device, one launch": no host round trip sits between
// the covariance pass and the projection pass.
//
// Jacobi: the n(n-1)/2 pivots of a sweep are visited in n-1 rounds of n/2
// disjoint pairs (round-robin tournament).  Within a round all rotations are
// independent: compute (c, s) per pair, rotate rows, then rotate columns of A and
// V.  A and V live in LDS (odd row stride, conflict-free column walks) whenever
// 2 n (n+1) doubles fit; otherwise in global scratch.
#include "common.h"

namespace {

constexpr int kEigThreads = 1024;
constexpr int kMaxPairs = 128;  // n <= 256

struct JacobiShared {
    double c[kMaxPairs], s[kMaxPairs];
    int p[kMaxPairs], q[kMaxPairs];
    double red[kEigThreads / 64];
    double bc[4];
    int ibc[4];
};

__device__ __forceinline__ double block_sum(double v, JacobiShared* sh) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh->red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh->red[i];
        sh->bc[0] = t;
    }
    __syncthreads();
    return sh->bc[0];
}

// A (n x n, row stride ld, symmetric) -> diagonal; V -> eigenvectors in columns.
// Returns the number of sweeps used (uniform across the block).
__device__ int jacobi_eigh(double* A, double* V, int n, int ld, JacobiShared* sh, int max_sweeps) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < n * n; i += nt) V[(i / n) * ld + (i % n)] = (i / n == i % n) ? 1.0 : 0.0;
    __syncthreads();
    if (n < 2) return 0;
    const int npad = n + (n & 1);
    const int m = npad / 2;
    int sweep = 0;
    for (; sweep < max_sweeps; ++sweep) {
        double off = 0.0, dia = 0.0;
        for (int i = tid; i < n * n; i += nt) {
            const int r = i / n, c = i % n;
            const double v = A[r * ld + c];
            if (r == c) dia = fma(v, v, dia); else off = fma(v, v, off);
        }
        off = block_sum(off, sh);
        dia = block_sum(dia, sh);
        // converged when ||off||_F <= n*eps*||A||_F: rounding of the rotations themselves
        // re-pollutes the zeroed entries at that level, and the eigenvalue error left
        // is second order in it.
        const double tol = (double)n * 2.220446049250313e-16;
        if (off <= tol * tol * (dia + off) || off == 0.0) break;
        for (int round = 0; round < npad - 1; ++round) {
            if (tid < m) {
                int a, b;
                if (tid == 0) { a = npad - 1; b = round; }
                else { a = (round + tid) % (npad - 1); b = (round - tid + (npad - 1)) % (npad - 1); }
                const int p = min(a, b), q = max(a, b);
                double c = 1.0, s = 0.0;
                if (q < n) {
                    const double apq = A[p * ld + q];
                    if (apq != 0.0) {
                        const double app = A[p * ld + p], aqq = A[q * ld + q];
                        const double theta = (aqq - app) / (2.0 * apq);
                        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(fma(theta, theta, 1.0)));
                        c = 1.0 / sqrt(fma(t, t, 1.0));
                        s = t * c;
                    }
                }
                sh->c[tid] = c; sh->s[tid] = s; sh->p[tid] = p; sh->q[tid] = q;
            }
            __syncthreads();
            // rows: A <- J' A
            for (int e = tid; e < m * n; e += nt) {
                const int i = e / n, j = e - i * n;
                const int q = sh->q[i];
                if (q >= n) continue;
                const int p = sh->p[i];
                const double c = sh->c[i], s = sh->s[i];
                const double ap = A[p * ld + j], aq = A[q * ld + j];
                A[p * ld + j] = c * ap - s * aq;
                A[q * ld + j] = s * ap + c * aq;
            }
            __syncthreads();
            // columns: A <- A J, V <- V J
            for (int e = tid; e < 2 * m * n; e += nt) {
                const int which = e / (m * n);
                const int e2 = e - which * m * n;
                const int i = e2 / n, r = e2 - i * n;
                const int q = sh->q[i];
                if (q >= n) continue;
                const int p = sh->p[i];
                const double c = sh->c[i], s = sh->s[i];
                double* M = which ? V : A;
                const double mp = M[r * ld + p], mq = M[r * ld + q];
                M[r * ld + p] = c * mp - s * mq;
                M[r * ld + q] = s * mp + c * mq;
            }
            __syncthreads();
            if (tid < m && sh->q[tid] < n) {  // the pivot is annihilated exactly
                A[sh->p[tid] * ld + sh->q[tid]] = 0.0;
                A[sh->q[tid] * ld + sh->p[tid]] = 0.0;
            }
            __syncthreads();
        }
    }
    return sweep;
}

// order[j] = index of the j-th largest |ev| (stable)
__device__ void sort_desc_abs(const double* ev, int n, int* order) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double a = fabs(ev[i]);
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const double b = fabs(ev[j]);
            rank += (b > a) || (b == a && j < i);
        }
        order[rank] = i;
    }
    __syncthreads();
}

// column j of M (n rows, stride ld) *= sign of its largest-magnitude entry (first occurrence)
__device__ void canonical_signs(double* M, int n, int ncols, int ld) {
    for (int j = threadIdx.x; j < ncols; j += blockDim.x) {
        double best = -1.0, sgn = 1.0;
        for (int i = 0; i < n; ++i) {
            const double v = M[i * ld + j];
            if (fabs(v) > best) { best = fabs(v); sgn = v < 0.0 ? -1.0 : 1.0; }
        }
        if (sgn < 0.0)
            for (int i = 0; i < n; ++i) M[i * ld + j] = -M[i * ld + j];
    }
    __syncthreads();
}

struct TicaWork {  // global scratch, each n*ld doubles unless noted
    double *C00, *C0t, *A, *V, *L, *tmp, *ev /*n*/, *mean /*n*/;
    int* order;  // n
};

// moments = [M00 F*F][M0t F*F][sx F][sy F][T]   (centred by shift, unscaled)
// scale   = per-feature divisor applied to the centred data (NULL -> 1)
__global__ __launch_bounds__(kEigThreads) void tica_solve_kernel(
    const double* __restrict__ mom, const double* __restrict__ scale, int n, int ld, double epsilon, int kinetic_map,
    TicaWork wk, int use_lds, double* __restrict__ out_eig, double* __restrict__ out_W, double* __restrict__ out_mean,
    int* __restrict__ out_rank) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ JacobiShared sh;
    const int tid = threadIdx.x, nt = blockDim.x;
    double* A = use_lds ? reinterpret_cast<double*>(smem_raw) : wk.A;
    double* V = use_lds ? A + (size_t)n * ld : wk.V;

    const double* M00 = mom;
    const double* M0t = mom + (size_t)n * n;
    const double* sx = M0t + (size_t)n * n;
    const double* sy = sx + n;
    const double T = sy[n];
    const double w = 2.0 * T;
    if (!(T > 0.0)) {
        if (tid == 0) *out_rank = 0;
        for (int i = tid; i < n; i += nt) { out_eig[i] = 0.0; out_mean[i] = 0.0; }
        for (int i = tid; i < n * n; i += nt) out_W[i] = 0.0;
        return;
    }
    for (int i = tid; i < n; i += nt) {
        const double is = scale ? 1.0 / scale[i] : 1.0;
        wk.mean[i] = (sx[i] + sy[i]) / w * is;
        out_mean[i] = wk.mean[i];
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e - i * n;
        const double isi = scale ? 1.0 / scale[i] : 1.0, isj = scale ? 1.0 / scale[j] : 1.0;
        const double mm = wk.mean[i] * wk.mean[j];
        const double c00 = 0.5 * (M00[e] + M00[j * n + i]) / w * isi * isj - mm;
        const double c0t = (M0t[e] + M0t[j * n + i]) / w * isi * isj - mm;
        wk.C00[i * ld + j] = c00;
        wk.C0t[i * ld + j] = c0t;
        A[i * ld + j] = c00;
    }
    __syncthreads();

    // ---- spd_inv_split(C00): eigh, sort by |ev| desc, cut at epsilon, canonical signs ----
    jacobi_eigh(A, V, n, ld, &sh, 40);
    for (int i = tid; i < n; i += nt) wk.ev[i] = A[i * ld + i];
    __syncthreads();
    sort_desc_abs(wk.ev, n, wk.order);
    if (tid == 0) {
        double evmin = wk.ev[0];
        for (int i = 1; i < n; ++i) evmin = fmin(evmin, wk.ev[i]);
        double eps = epsilon;
        if (evmin < 0.0) eps = fmax(eps, -evmin + 1e-16);
        int rank = 0;
        for (int i = 0; i < n; ++i) rank += fabs(wk.ev[i]) >= eps;
        sh.ibc[0] = rank;
        *out_rank = rank;
    }
    __syncthreads();
    const int rank = sh.ibc[0];
    if (rank == 0) {
        for (int i = tid; i < n; i += nt) out_eig[i] = 0.0;
        for (int i = tid; i < n * n; i += nt) out_W[i] = 0.0;
        return;
    }
    for (int e = tid; e < n * rank; e += nt) {
        const int i = e / rank, j = e - i * rank;
        wk.L[i * ld + j] = V[i * ld + wk.order[j]];
    }
    __syncthreads();
    canonical_signs(wk.L, n, rank, ld);
    for (int e = tid; e < n * rank; e += nt) {
        const int i = e / rank, j = e - i * rank;
        wk.L[i * ld + j] /= sqrt(wk.ev[wk.order[j]]);
    }
    __syncthreads();
    // ---- Ct = L' C0t L ----
    for (int e = tid; e < n * rank; e += nt) {
        const int i = e / rank, j = e - i * rank;
        double a = 0.0;
        for (int k = 0; k < n; ++k) a = fma(wk.C0t[i * ld + k], wk.L[k * ld + j], a);
        wk.tmp[i * ld + j] = a;
    }
    __syncthreads();
    for (int e = tid; e < rank * rank; e += nt) {
        const int i = e / rank, j = e - i * rank;
        double a = 0.0;
        for (int k = 0; k < n; ++k) a = fma(wk.L[k * ld + i], wk.tmp[k * ld + j], a);
        wk.C00[i * ld + j] = a;  // C00 no longer needed: holds Ct
    }
    __syncthreads();
    for (int e = tid; e < rank * rank; e += nt) {
        const int i = e / rank, j = e - i * rank;
        A[i * ld + j] = 0.5 * (wk.C00[i * ld + j] + wk.C00[j * ld + i]);
    }
    __syncthreads();
    jacobi_eigh(A, V, rank, ld, &sh, 40);
    for (int i = tid; i < rank; i += nt) wk.ev[i] = A[i * ld + i];
    __syncthreads();
    sort_desc_abs(wk.ev, rank, wk.order);
    // ---- R = L Rt (sorted), canonical signs, kinetic map ----
    for (int e = tid; e < n * rank; e += nt) {
        const int i = e / rank, j = e - i * rank;
        const int src = wk.order[j];
        double a = 0.0;
        for (int k = 0; k < rank; ++k) a = fma(wk.L[i * ld + k], V[k * ld + src], a);
        wk.tmp[i * ld + j] = a;
    }
    __syncthreads();
    canonical_signs(wk.tmp, n, rank, ld);
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e - i * n;
        double v = 0.0;
        if (j < rank) {
            v = wk.tmp[i * ld + j];
            if (kinetic_map) v *= wk.ev[wk.order[j]];
        }
        out_W[e] = v;
    }
    for (int j = tid; j < n; j += nt) out_eig[j] = j < rank ? wk.ev[wk.order[j]] : 0.0;
}

// Plain symmetric eigendecomposition (ascending eigenvalues), for tests and the
// reversible MSM path.
__global__ __launch_bounds__(kEigThreads) void eigh_kernel(const double* __restrict__ Ain, int n, int ld,
                                                          double* gA, double* gV, int use_lds, int* order,
                                                          double* __restrict__ out_w, double* __restrict__ out_v,
                                                          int* __restrict__ out_sweeps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ JacobiShared sh;
    const int tid = threadIdx.x, nt = blockDim.x;
    double* A = use_lds ? reinterpret_cast<double*>(smem_raw) : gA;
    double* V = use_lds ? A + (size_t)n * ld : gV;
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e - i * n;
        A[i * ld + j] = 0.5 * (Ain[e] + Ain[j * n + i]);
    }
    __syncthreads();
    const int sweeps = jacobi_eigh(A, V, n, ld, &sh, 40);
    // ascending order by value
    for (int i = tid; i < n; i += nt) {
        const double a = A[i * ld + i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const double b = A[j * ld + j];
            rank += (b < a) || (b == a && j < i);
        }
        order[rank] = i;
    }
    __syncthreads();
    for (int j = tid; j < n; j += nt) out_w[j] = A[order[j] * ld + order[j]];
    if (out_v)
        for (int e = tid; e < n * n; e += nt) {
            const int i = e / n, j = e - i * n;
            out_v[e] = V[i * ld + order[j]];
        }
    if (tid == 0 && out_sweeps) *out_sweeps = sweeps;
}

size_t jacobi_lds_bytes(int n, int ld) { return (size_t)2 * n * ld * sizeof(double); }

}  // namespace

extern "C" {

msm_status msm_tica_solve(msm_ctx* ctx, const double* d_moments, const double* d_scale, int F, double epsilon,
                          int kinetic_map, double* d_eigvals, double* d_coeffs, double* d_mean, int* d_rank) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, F >= 1 && F <= 2 * kMaxPairs, "msm_tica_solve: need 1 <= F <= %d (got %d)", 2 * kMaxPairs, F);
    MSM_REQUIRE(ctx, epsilon >= 0.0, "msm_tica_solve: epsilon must be >= 0");
    MSM_REQUIRE(ctx, d_moments && d_eigvals && d_coeffs && d_mean && d_rank, "msm_tica_solve: NULL pointer");
    const int ld = F | 1;  // odd stride
    const size_t mat = (size_t)F * ld;
    const size_t need = (6 * mat + 2 * F) * sizeof(double) + (size_t)F * sizeof(int) + 64;
    msm_status rs = msm_reserve_scratch(ctx, need);
    if (rs != MSM_OK) return rs;
    double* base = (double*)ctx->scratch;
    TicaWork wk;
    wk.C00 = base; wk.C0t = base + mat; wk.A = base + 2 * mat; wk.V = base + 3 * mat; wk.L = base + 4 * mat;
    wk.tmp = base + 5 * mat; wk.ev = base + 6 * mat; wk.mean = wk.ev + F; wk.order = (int*)(wk.mean + F);
    const size_t lds = jacobi_lds_bytes(F, ld);
    const int use_lds = lds <= 140 * 1024;
    if (use_lds && lds > 48 * 1024)
        MSM_HIP(ctx, hipFuncSetAttribute((const void*)tica_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds));
    hipLaunchKernelGGL(tica_solve_kernel, dim3(1), dim3(kEigThreads), use_lds ? lds : 0, ctx->stream, d_moments,
                       d_scale, F, ld, epsilon, kinetic_map, wk, use_lds, d_eigvals, d_coeffs, d_mean, d_rank);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_eigh(msm_ctx* ctx, const double* d_a, int n, double* d_w, double* d_v, int* d_sweeps) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 1 && n <= 2 * kMaxPairs, "msm_eigh: need 1 <= n <= %d (got %d)", 2 * kMaxPairs, n);
    MSM_REQUIRE(ctx, d_a && d_w, "msm_eigh: NULL pointer");
    const int ld = n | 1;
    const size_t mat = (size_t)n * ld;
    msm_status rs = msm_reserve_scratch(ctx, 2 * mat * sizeof(double) + (size_t)n * sizeof(int) + 64);
    if (rs != MSM_OK) return rs;
    double* gA = (double*)ctx->scratch;
    double* gV = gA + mat;
    int* order = (int*)(gV + mat);
    const size_t lds = jacobi_lds_bytes(n, ld);
    const int use_lds = lds <= 140 * 1024;
    if (use_lds && lds > 48 * 1024)
        MSM_HIP(ctx, hipFuncSetAttribute((const void*)eigh_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(eigh_kernel, dim3(1), dim3(kEigThreads), use_lds ? lds : 0, ctx->stream, d_a, n, ld, gA, gV,
                       use_lds, order, d_w, d_v, d_sweeps);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

}  // extern "C"
