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

// Short-latency fp64 reciprocal / reciprocal square root: hardware seed plus two
// Newton steps (the IEEE division / sqrt expansions are 3-4x longer dependent chains,
// and the rotation set-up is the serial part of every Jacobi round).
__device__ __forceinline__ double nr_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}
__device__ __forceinline__ double nr_rsqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    r = r * fma(-h * r, r, 1.5);
    r = r * fma(-h * r, r, 1.5);
    return r;
}

// Jacobi rotation annihilating a_pq:  t = sgn(a) b / (|a| + sqrt(a^2 + b^2)) with
// a = (a_qq - a_pp)/2, b = a_pq;  c = 1/sqrt(1 + t^2), s = t c  (so c^2 + s^2 = 1 to
// rounding, which is what keeps V orthogonal).
__device__ __forceinline__ void jacobi_rotation(double app, double aqq, double apq, double& c, double& s) {
    c = 1.0; s = 0.0;
    if (apq == 0.0) return;
    const double a = 0.5 * (aqq - app);
    const double h2 = fma(a, a, apq * apq);
    if (!(h2 > 1e-300) || !(h2 < 1e300)) return;  // degenerate scale: skip this pivot
    const double h = h2 * nr_rsqrt(h2);
    const double t = (a >= 0.0 ? apq : -apq) * nr_rcp(fabs(a) + h);
    c = nr_rsqrt(fma(t, t, 1.0));
    s = t * c;
}

// A (n x n, row stride ld, symmetric) -> diagonal; V -> eigenvectors in columns.
// Returns the number of sweeps used (uniform across the block).
//
// Work split: a wave owns pivots i = wave, wave + n_waves, ... of the round and its
// lanes walk the row (then column) index, so (p, q, c, s) are wave-uniform values kept
// in registers between the two phases: no parameter exchange, two barriers per round.
// A wave may form its rotation right after the previous round's barrier because rows
// p, q (which hold a_pp, a_qq, a_pq) are touched by no other pivot of the round.
__device__ int jacobi_eigh(double* A, double* V, int n, int ld, JacobiShared* sh, int max_sweeps) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, n_waves = nt >> 6;
    for (int i = tid; i < n * n; i += nt) V[(i / n) * ld + (i % n)] = (i / n == i % n) ? 1.0 : 0.0;
    __syncthreads();
    if (n < 2) return 0;
    const int npad = n + (n & 1);
    const int m = npad / 2;
    int sweep = 0;
    for (; sweep < max_sweeps; ++sweep) {
        double off = 0.0, dia = 0.0;
        for (int r = wave; r < n; r += n_waves)
            for (int c = lane; c < n; c += 64) {
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
            // rows: A <- J' A   (rotation parameters parked in LDS for the column phase)
#pragma unroll 1
            for (int i = wave; i < m; i += n_waves) {
                int a, b;
                if (i == 0) { a = npad - 1; b = round; }
                else { a = (round + i) % (npad - 1); b = (round - i + (npad - 1)) % (npad - 1); }
                const int p = min(a, b), q = max(a, b);
                if (q >= n) continue;  // padding pivot (odd n)
                double c, s;
                jacobi_rotation(A[p * ld + p], A[q * ld + q], A[p * ld + q], c, s);
                if (lane == 0) { sh->c[i] = c; sh->s[i] = s; }
                double* rp = A + p * ld;
                double* rq = A + q * ld;
                for (int j = lane; j < n; j += 64) {
                    const double ap = rp[j], aq = rq[j];
                    rp[j] = c * ap - s * aq;
                    rq[j] = s * ap + c * aq;
                }
            }
            __syncthreads();
            // columns: A <- A J, V <- V J; the pivot is annihilated exactly
#pragma unroll 1
            for (int i = wave; i < m; i += n_waves) {
                int a, b;
                if (i == 0) { a = npad - 1; b = round; }
                else { a = (round + i) % (npad - 1); b = (round - i + (npad - 1)) % (npad - 1); }
                const int p = min(a, b), q = max(a, b);
                if (q >= n) continue;
                const double c = sh->c[i], s = sh->s[i];
                for (int r = lane; r < n; r += 64) {
                    const double ap = A[r * ld + p], aq = A[r * ld + q];
                    const double vp = V[r * ld + p], vq = V[r * ld + q];
                    double np_ = c * ap - s * aq, nq_ = s * ap + c * aq;
                    if (r == p) nq_ = 0.0;
                    if (r == q) np_ = 0.0;
                    A[r * ld + p] = np_;
                    A[r * ld + q] = nq_;
                    V[r * ld + p] = c * vp - s * vq;
                    V[r * ld + q] = s * vp + c * vq;
                }
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

struct TicaWork {  // global scratch: four n*ld matrices, then ev[n], mean[n], isc[n], order[n]
    double *A, *V, *B1, *B2, *ev, *mean, *isc;
    int* order;
};

// C[i][j] = sum_k opA(i,k) * B[k][j]   (opA = A or A'), i < rows, j < cols, k < inner
__device__ void small_mm(double* C, const double* A, bool transA, const double* B, int rows, int cols, int inner,
                         int ld) {
    for (int e = threadIdx.x; e < rows * cols; e += blockDim.x) {
        const int i = e / cols, j = e - i * cols;
        double a = 0.0;
        if (transA)
            for (int k = 0; k < inner; ++k) a = fma(A[k * ld + i], B[k * ld + j], a);
        else
            for (int k = 0; k < inner; ++k) a = fma(A[i * ld + k], B[k * ld + j], a);
        C[i * ld + j] = a;
    }
    __syncthreads();
}

// moments = [M00 F*F][M0t F*F][sx F][sy F][T]   (centred by shift, unscaled)
// scale   = per-feature divisor applied to the centred data (NULL -> 1)
// lds_mats: how many of {A, V, B1, B2} live in LDS (4, 2 or 0)
__global__ __launch_bounds__(kEigThreads) void tica_solve_kernel(
    const double* __restrict__ mom, const double* __restrict__ scale, int n, int ld, double epsilon, int kinetic_map,
    TicaWork wk, int lds_mats, double* __restrict__ out_eig, double* __restrict__ out_W, double* __restrict__ out_mean,
    int* __restrict__ out_rank) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ JacobiShared sh;
    const int tid = threadIdx.x, nt = blockDim.x;
    double* lds = reinterpret_cast<double*>(smem_raw);
    const size_t mat = (size_t)n * ld;
    double* A = lds_mats >= 2 ? lds : wk.A;
    double* V = lds_mats >= 2 ? lds + mat : wk.V;
    double* B1 = lds_mats >= 4 ? lds + 2 * mat : wk.B1;
    double* B2 = lds_mats >= 4 ? lds + 3 * mat : wk.B2;

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
        wk.isc[i] = is;
        wk.mean[i] = (sx[i] + sy[i]) / w * is;
        out_mean[i] = wk.mean[i];
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e - i * n;
        const double ss = wk.isc[i] * wk.isc[j];
        const double mm = wk.mean[i] * wk.mean[j];
        A[i * ld + j] = 0.5 * (M00[e] + M00[j * n + i]) / w * ss - mm;   // C00
        B1[i * ld + j] = (M0t[e] + M0t[j * n + i]) / w * ss - mm;        // C0t
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
    // L (in B2) = V[:, order[:rank]] with canonical signs, columns scaled by 1/sqrt(s)
    for (int e = tid; e < n * rank; e += nt) {
        const int i = e / rank, j = e - i * rank;
        B2[i * ld + j] = V[i * ld + wk.order[j]];
    }
    __syncthreads();
    canonical_signs(B2, n, rank, ld);
    for (int e = tid; e < n * rank; e += nt) {
        const int i = e / rank, j = e - i * rank;
        B2[i * ld + j] /= sqrt(wk.ev[wk.order[j]]);
    }
    __syncthreads();
    // ---- Ct = L' C0t L: A <- C0t L, V <- L' A, A <- sym(V) ----
    small_mm(A, B1, false, B2, n, rank, n, ld);
    small_mm(V, B2, true, A, rank, rank, n, ld);
    for (int e = tid; e < rank * rank; e += nt) {
        const int i = e / rank, j = e - i * rank;
        A[i * ld + j] = 0.5 * (V[i * ld + j] + V[j * ld + i]);
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
        for (int k = 0; k < rank; ++k) a = fma(B2[i * ld + k], V[k * ld + src], a);
        B1[i * ld + j] = a;
    }
    __syncthreads();
    canonical_signs(B1, n, rank, ld);
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e - i * n;
        double v = 0.0;
        if (j < rank) {
            v = B1[i * ld + j];
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
    const size_t need = (4 * mat + 3 * F) * sizeof(double) + (size_t)F * sizeof(int) + 64;
    msm_status rs = msm_reserve_scratch(ctx, need);
    if (rs != MSM_OK) return rs;
    double* base = (double*)ctx->scratch;
    TicaWork wk;
    wk.A = base; wk.V = base + mat; wk.B1 = base + 2 * mat; wk.B2 = base + 3 * mat;
    wk.ev = base + 4 * mat; wk.mean = wk.ev + F; wk.isc = wk.mean + F; wk.order = (int*)(wk.isc + F);
    const size_t lds_budget = 150 * 1024;
    int lds_mats = 0;
    if (4 * mat * sizeof(double) <= lds_budget) lds_mats = 4;
    else if (2 * mat * sizeof(double) <= lds_budget) lds_mats = 2;
    const size_t lds = (size_t)lds_mats * mat * sizeof(double);
    if (lds > 48 * 1024)
        MSM_HIP(ctx, hipFuncSetAttribute((const void*)tica_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds));
    hipLaunchKernelGGL(tica_solve_kernel, dim3(1), dim3(kEigThreads), lds, ctx->stream, d_moments, d_scale, F, ld,
                       epsilon, kinetic_map, wk, lds_mats, d_eigvals, d_coeffs, d_mean, d_rank);
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
