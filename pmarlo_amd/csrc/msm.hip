// Counts -> transition matrix -> stationary distribution / leading eigenvalues /
// implied timescales, all on the device.
//
// The matrices are k x k (k <= a few thousand): tiny against HBM, so every stage
// is latency-bound (SURVEY.md section 8d "row-normalise, pi, top-m eigenvalues").  Design:
// one workgroup per matrix, many matrices per launch (the lag scan batches its
// lags over blockIdx.x), no host round trips inside a solve.
//
// Spectrum: block power ("subspace") iteration on T' with Cholesky-QR
// re-orthonormalisation, Rayleigh-Ritz on the p x p projected matrix, whose
// non-symmetric eigenproblem is solved by elmhes/hqr (small_eig.h).  The Ritz
// vector of the eigenvalue nearest 1 gives the stationary distribution.
#include "common.h"
#include "small_eig.h"

namespace {

constexpr int kThreads = 256;
constexpr int kSolveThreads = 1024;
constexpr int kMaxP = 32;

template <typename CT>
__device__ __forceinline__ double cnt_as_f64(CT v) { return (double)v; }

// rowsum / colsum of the k x k count matrix (one block per row for rows; atomics-free
// column sums through a second pass with the transposed walk)
template <typename CT>
__global__ __launch_bounds__(kThreads) void rowcol_sums_kernel(const CT* __restrict__ C, int k,
                                                              double* __restrict__ rowsum,
                                                              double* __restrict__ colsum) {
    __shared__ double red[kThreads / 64];
    const int i = blockIdx.x;  // state index: row i and column i
    double r = 0.0, c = 0.0;
    for (int j = threadIdx.x; j < k; j += kThreads) {
        r += cnt_as_f64(C[(size_t)i * k + j]);
        c += cnt_as_f64(C[(size_t)j * k + i]);
    }
    for (int off = 32; off > 0; off >>= 1) { r += __shfl_down(r, off, 64); c += __shfl_down(c, off, 64); }
    __shared__ double red2[kThreads / 64];
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = r; red2[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tr = 0.0, tc = 0.0;
        for (int w = 0; w < kThreads / 64; ++w) { tr += red[w]; tc += red2[w]; }
        rowsum[i] = tr;
        colsum[i] = tc;
    }
}

// active = states with rowsum + colsum > eps, in ascending order (np.where); ka = count
__global__ __launch_bounds__(1024) void active_set_kernel(const double* __restrict__ rowsum,
                                                         const double* __restrict__ colsum, int k, double eps,
                                                         int use_all, int* __restrict__ active,
                                                         int* __restrict__ inv_map, int* __restrict__ ka_out) {
    __shared__ int scan[1024];
    __shared__ int base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < k; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        const int flag = (i < k) && (use_all || (rowsum[i] + colsum[i] > eps));
        scan[threadIdx.x] = flag;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
            const int v = threadIdx.x >= off ? scan[threadIdx.x - off] : 0;
            __syncthreads();
            scan[threadIdx.x] += v;
            __syncthreads();
        }
        if (i < k) {
            if (flag) { const int pos = base + scan[threadIdx.x] - 1; active[pos] = i; inv_map[i] = pos; }
            else inv_map[i] = -1;
        }
        __syncthreads();
        if (threadIdx.x == 1023) base += scan[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *ka_out = base;
}

// mode 0: T[i][j] = C[i][j] / rowsum[i] (zero rows stay 0), full k x k          (discretize.py:678-682)
// mode 1: T[a][b] = (C[act a][act b] + alpha) / (rowsum[act a] + ka*alpha), packed ka x ka, ld = k
//                                                     (msm_utils.py:129-167 + _estimation.py:158-188)
template <typename CT>
__global__ __launch_bounds__(kThreads) void build_T_kernel(const CT* __restrict__ C, int k, int mode, double alpha,
                                                          const double* __restrict__ rowsum,
                                                          const int* __restrict__ active,
                                                          const int* __restrict__ ka_ptr, double* __restrict__ T) {
    const int a = blockIdx.x;
    if (mode == 0) {
        const double rs = rowsum[a];
        for (int j = threadIdx.x; j < k; j += kThreads)
            T[(size_t)a * k + j] = rs > 0.0 ? cnt_as_f64(C[(size_t)a * k + j]) / rs : 0.0;
        return;
    }
    const int ka = *ka_ptr;
    if (a >= ka) return;
    const int ia = active[a];
    const double denom = rowsum[ia] + (double)ka * alpha;
    for (int b = threadIdx.x; b < ka; b += kThreads)
        T[(size_t)a * k + b] = (cnt_as_f64(C[(size_t)ia * k + active[b]]) + alpha) / denom;
}

__global__ __launch_bounds__(1024) void diag_mass_kernel(const double* __restrict__ T, int k, double* __restrict__ out) {
    __shared__ double red[16];
    double t = 0.0;
    for (int i = threadIdx.x; i < k; i += 1024) t += T[(size_t)i * k + i];
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < 16; ++w) s += red[w];
        *out = k > 0 ? s / (double)k : __builtin_nan("");
    }
}

// T_full = I with the active block; pi_full = 0 outside the active set (_estimation.py:174-181)
__global__ __launch_bounds__(kThreads) void embed_full_kernel(const double* __restrict__ T_act,
                                                             const double* __restrict__ pi_act,
                                                             const int* __restrict__ inv_map, int k,
                                                             double* __restrict__ T_full,
                                                             double* __restrict__ pi_full) {
    const int i = blockIdx.x;
    const int a = inv_map[i];
    for (int j = threadIdx.x; j < k; j += kThreads) {
        const int b = inv_map[j];
        double v = (i == j) ? 1.0 : 0.0;
        if (a >= 0) v = b >= 0 ? T_act[(size_t)a * k + b] : 0.0;
        T_full[(size_t)i * k + j] = v;
    }
    if (threadIdx.x == 0 && pi_full) pi_full[i] = a >= 0 && pi_act ? pi_act[a] : 0.0;
}

// ---------------------------------------------------------------------------
// spectrum of a packed row-stochastic matrix T (n x n, row stride ld)
// ---------------------------------------------------------------------------
struct SpecArgs {
    const double* T;      // [batch] matrices, stride t_stride
    size_t t_stride;
    int ld;
    const int* n_ptr;     // [batch] (or NULL -> n_fixed)
    int n_fixed;
    int p;                // subspace size (<= kMaxP)
    int n_iter;
    int init;             // 1: (re)initialise the basis
    unsigned long long seed;
    double* Z;            // [batch][n_max * p] basis (persists between calls)
    double* W;            // [batch][n_max * p]
    size_t zw_stride;
    double* ritz;         // [batch][4 * kMaxP]: re | im | previous re | previous im   (sorted by |.| desc)
    double* pi;           // [batch][n_max] or NULL
    size_t pi_stride;
    double* change;       // [batch] max relative change of the top `n_watch` Ritz values over the last check gap
    int n_watch;
    int check_gap;
    int* status;          // [batch] 0 ok, else hqr failure index
    // implied timescales (optional)
    int n_its;            // 0: skip
    const double* lags;   // [batch]
    double* its_eig;      // [batch][n_its]
    double* its_ts;       // [batch][n_its]
};

struct SpecShared {
    double G[kMaxP * kMaxP];
    double R[kMaxP * kMaxP];
    double H[kMaxP * kMaxP];
    double Hw[kMaxP * kMaxP + 2 * kMaxP * kMaxP];  // hqr copy + eigenvector work
    double wr[kMaxP], wi[kMaxP], y[kMaxP];
    double red[kSolveThreads / 64];
    double bc;
    int order[kMaxP];
    int status;
};

__device__ double spec_block_sum(double v, SpecShared* sh) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh->red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh->red[i];
        sh->bc = t;
    }
    __syncthreads();
    return sh->bc;
}

// W = T' Z   (W[j][c] = sum_i T[i][j] Z[i][c]); T reads are coalesced over j.
__device__ void spec_apply_Tt(const double* __restrict__ T, int ld, int n, int p, const double* __restrict__ Z,
                              double* __restrict__ W, double* lds_part /* [groups][64][p] unused for groups == 1 */) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    // each wave owns column tiles of 64 states; within the tile lanes = j
    for (int j0 = wave * 64; j0 < n; j0 += n_waves * 64) {
        const int j = j0 + lane;
        for (int c0 = 0; c0 < p; c0 += 8) {
            double acc[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = 0.0;
            if (j < n) {
                for (int i = 0; i < n; ++i) {
                    const double t = T[(size_t)i * ld + j];
                    const double* zr = Z + (size_t)i * p + c0;
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        if (c0 + c < p) acc[c] = fma(t, zr[c], acc[c]);
                }
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (c0 + c < p) W[(size_t)j * p + c0 + c] = acc[c];
            }
        }
    }
    (void)lds_part;
    __syncthreads();
}

// M[a][b] = sum_j A[j][a] B[j][b]  (p x p), every thread block-strided over j
__device__ void spec_gram(const double* __restrict__ A, const double* __restrict__ B, int n, int p, double* M,
                          SpecShared* sh) {
    for (int a = 0; a < p; ++a)
        for (int b = 0; b < p; ++b) {
            double acc = 0.0;
            for (int j = threadIdx.x; j < n; j += blockDim.x) acc = fma(A[(size_t)j * p + a], B[(size_t)j * p + b], acc);
            acc = spec_block_sum(acc, sh);
            if (threadIdx.x == 0) M[a * p + b] = acc;
        }
    __syncthreads();
}

// faster Gram: thread (pair, group) partial sums, then fixed-order reduce through LDS scratch in W tail
__device__ void spec_gram_fast(const double* __restrict__ A, const double* __restrict__ B, int n, int p, double* M,
                               double* scratch /* >= blockDim.x doubles, global */) {
    const int pairs = p * p;
    const int groups = blockDim.x / pairs;  // >= 1 because p <= 32 and blockDim = 1024
    const int tid = threadIdx.x;
    const int pr = tid % pairs, g = tid / pairs;
    double acc = 0.0;
    if (g < groups) {
        const int a = pr / p, b = pr - a * p;
        for (int j = g; j < n; j += groups) acc = fma(A[(size_t)j * p + a], B[(size_t)j * p + b], acc);
        scratch[tid] = acc;
    }
    __syncthreads();
    if (tid < pairs) {
        double t = 0.0;
        for (int gg = 0; gg < groups; ++gg) t += scratch[gg * pairs + tid];
        M[tid] = t;
    }
    __syncthreads();
}

// Cholesky G = R'R (upper R) by one wave, lanes over columns; then Z = W R^-1 row-wise.
__device__ void spec_cholqr(double* G, double* R, int p, const double* __restrict__ W, double* __restrict__ Z, int n) {
    const int tid = threadIdx.x;
    if (tid < 64) {
        const int j = tid;
        for (int c = 0; c < p; ++c) {
            // R[c][j] for j >= c
            double v = 0.0;
            if (j < p && j >= c) {
                v = G[c * p + j];
                for (int m = 0; m < c; ++m) v = fma(-R[m * p + c], R[m * p + j], v);
            }
            double diag = __shfl(v, c, 64);
            if (!(diag > 1e-300)) diag = 1e-300;  // rank-deficient basis: keep going, column dies out
            const double rcc = sqrt(diag);
            if (j < p) R[c * p + j] = j > c ? v / rcc : (j == c ? rcc : 0.0);
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += blockDim.x) {
        double z[kMaxP];
#pragma unroll 1
        for (int c = 0; c < p; ++c) {
            double v = W[(size_t)i * p + c];
            for (int m = 0; m < c; ++m) v = fma(-z[m], R[m * p + c], v);
            z[c] = v / R[c * p + c];
            Z[(size_t)i * p + c] = z[c];
        }
    }
    __syncthreads();
}

__device__ void spec_ritz(SpecShared* sh, int p, double* out_re, double* out_im) {
    // thread 0: eigenvalues of a copy of H, sorted by descending magnitude
    if (threadIdx.x == 0) {
        for (int i = 0; i < p * p; ++i) sh->Hw[i] = sh->H[i];
        const int rc = small_eig::eigenvalues(sh->Hw, p, p, sh->wr, sh->wi);
        if (rc) sh->status = rc;
        for (int i = 0; i < p; ++i) {
            const double a = sh->wr[i] * sh->wr[i] + sh->wi[i] * sh->wi[i];
            int rank = 0;
            for (int j = 0; j < p; ++j) {
                const double b = sh->wr[j] * sh->wr[j] + sh->wi[j] * sh->wi[j];
                rank += (b > a) || (b == a && (sh->wr[j] > sh->wr[i] || (sh->wr[j] == sh->wr[i] && j < i)));
            }
            sh->order[rank] = i;
        }
        for (int r = 0; r < p; ++r) { out_re[r] = sh->wr[sh->order[r]]; out_im[r] = sh->wi[sh->order[r]]; }
    }
    __syncthreads();
}

__global__ __launch_bounds__(kSolveThreads) void spectrum_kernel(SpecArgs ar) {
    __shared__ SpecShared sh;
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int n = ar.n_ptr ? ar.n_ptr[b] : ar.n_fixed;
    const double* T = ar.T + (size_t)b * ar.t_stride;
    double* Z = ar.Z + (size_t)b * ar.zw_stride;
    double* W = ar.W + (size_t)b * ar.zw_stride;
    double* ritz = ar.ritz + (size_t)b * 4 * kMaxP;
    if (tid == 0) sh.status = 0;
    __syncthreads();
    if (n <= 0) {
        if (tid == 0) { ar.change[b] = 0.0; ar.status[b] = 0; }
        for (int i = tid; i < 4 * kMaxP; i += blockDim.x) ritz[i] = 0.0;
        for (int i = tid; i < ar.n_its; i += blockDim.x) {
            ar.its_eig[(size_t)b * ar.n_its + i] = __builtin_nan("");
            ar.its_ts[(size_t)b * ar.n_its + i] = __builtin_nan("");
        }
        return;
    }
    const int p = min(ar.p, n);
    if (ar.init) {
        for (int e = tid; e < n * p; e += blockDim.x) {
            const int i = e / p, c = e - i * p;
            double v;
            if (c == 0) v = 1.0;
            else {
                unsigned long long h = ar.seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(e + 1);
                h = (h ^ (h >> 30)) * 0xBF58476D1CE4E5B9ull;
                h = (h ^ (h >> 27)) * 0x94D049BB133111EBull;
                h ^= h >> 31;
                v = (double)(h >> 11) * (2.0 / 9007199254740992.0) - 1.0;
            }
            W[e] = v;
        }
        __syncthreads();
        spec_gram_fast(W, W, n, p, sh.G, Z);  // Z is free scratch here (n*p >= p*p*groups? use safe path below)
        spec_cholqr(sh.G, sh.R, p, W, Z, n);
    }
    double* scratch = W + (size_t)n * p;  // tail of the W allocation (>= kSolveThreads doubles)
    for (int it = 0; it < ar.n_iter; ++it) {
        const bool check_prev = (it == ar.n_iter - 1 - ar.check_gap);
        spec_apply_Tt(T, ar.ld, n, p, Z, W, nullptr);
        if (check_prev) {
            spec_gram_fast(Z, W, n, p, sh.H, scratch);
            spec_ritz(&sh, p, ritz + 2 * kMaxP, ritz + 3 * kMaxP);
        }
        spec_gram_fast(W, W, n, p, sh.G, scratch);
        spec_cholqr(sh.G, sh.R, p, W, Z, n);
    }
    // Rayleigh-Ritz on the final basis
    spec_apply_Tt(T, ar.ld, n, p, Z, W, nullptr);
    spec_gram_fast(Z, W, n, p, sh.H, scratch);
    spec_ritz(&sh, p, ritz, ritz + kMaxP);
    for (int i = p + tid; i < kMaxP; i += blockDim.x) { ritz[i] = 0.0; ritz[kMaxP + i] = 0.0; }
    __syncthreads();
    // Convergence measure: the true residual ||T'x - theta x|| / ||x|| of every watched REAL
    // Ritz pair (x = Z y, T'x = W y); for complex values, the change over the last check gap.
    // (Changes of Ritz values alone stagnate on clustered spectra and would stop too early.)
    {
        const int nw = min(ar.n_watch, p);
        double worst = 0.0;
        for (int wv = 0; wv < nw; ++wv) {
            const double th_re = ritz[wv], th_im = ritz[kMaxP + wv];
            if (th_im != 0.0) {
                const double dr = th_re - ritz[2 * kMaxP + wv], di = th_im - ritz[3 * kMaxP + wv];
                const double mag = sqrt(th_re * th_re + th_im * th_im);
                const double ch = (ar.n_iter > ar.check_gap) ? sqrt(dr * dr + di * di) / fmax(mag, 1e-300) : 1.0;
                worst = fmax(worst, ch);
                continue;
            }
            if (tid == 0) small_eig::eigenvector(sh.H, p, p, th_re, sh.y, sh.Hw);
            __syncthreads();
            double rn = 0.0, xn = 0.0;
            for (int i = tid; i < n; i += blockDim.x) {
                double xv = 0.0, tv = 0.0;
                for (int c = 0; c < p; ++c) {
                    xv = fma(Z[(size_t)i * p + c], sh.y[c], xv);
                    tv = fma(W[(size_t)i * p + c], sh.y[c], tv);
                }
                const double r = tv - th_re * xv;
                rn = fma(r, r, rn);
                xn = fma(xv, xv, xn);
            }
            rn = spec_block_sum(rn, &sh);
            xn = spec_block_sum(xn, &sh);
            worst = fmax(worst, sqrt(rn / fmax(xn, 1e-300)));
        }
        if (tid == 0) ar.change[b] = worst;
    }
    __syncthreads();
    // stationary distribution: Ritz vector of the eigenvalue nearest 1
    if (ar.pi) {
        if (tid == 0) {
            int best = 0;
            double bd = 1e300;
            for (int i = 0; i < p; ++i) {
                const double dr = sh.wr[i] - 1.0, di = sh.wi[i];
                const double dd = dr * dr + di * di;
                if (dd < bd) { bd = dd; best = i; }
            }
            small_eig::eigenvector(sh.H, p, p, sh.wr[best], sh.y, sh.Hw);
        }
        __syncthreads();
        double* pi = ar.pi + (size_t)b * ar.pi_stride;
        double part = 0.0;
        for (int i = tid; i < n; i += blockDim.x) {
            double v = 0.0;
            for (int c = 0; c < p; ++c) v = fma(Z[(size_t)i * p + c], sh.y[c], v);
            pi[i] = v;
            part += v;
        }
        const double tot = spec_block_sum(part, &sh);
        for (int i = tid; i < n; i += blockDim.x) pi[i] = pi[i] / tot;
    }
    // implied timescales (_its.py:543-604 on one matrix; utils.py:17-57)
    if (ar.n_its > 0 && tid == 0) {
        const int n_its = ar.n_its;
        const int kreq = min(n_its + 1, p);
        // top kreq by magnitude are ritz[0..kreq); re-sort by descending real part
        int idx[kMaxP];
        for (int i = 0; i < kreq; ++i) {
            int rank = 0;
            for (int j = 0; j < kreq; ++j) rank += (ritz[j] > ritz[i]) || (ritz[j] == ritz[i] && j < i);
            idx[rank] = i;
        }
        const double lag = ar.lags ? ar.lags[b] : 1.0;
        for (int i = 0; i < n_its; ++i) {
            double ev = __builtin_nan(""), ts = __builtin_nan("");
            if (i + 1 < kreq) {
                ev = fabs(ritz[idx[i + 1]]);
                ev = fmin(fmax(ev, 1e-12), 1.0 - 1e-12);
                ts = -fmax(1.0, lag) / log(ev);
            }
            ar.its_eig[(size_t)b * n_its + i] = ev;
            ar.its_ts[(size_t)b * n_its + i] = ts;
        }
    }
    if (tid == 0) ar.status[b] = sh.status;
}

}  // namespace

extern "C" {

msm_status msm_transition_matrix(msm_ctx* ctx, const void* d_counts, int counts_are_f64, int k, int mode,
                                 double alpha, double epsilon, double* d_T, int32_t* d_active, int32_t* d_inv_map,
                                 int32_t* d_n_active, double* d_rowsum, double* d_diag_mass) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, k >= 1, "msm_transition_matrix: k must be >= 1");
    MSM_REQUIRE(ctx, mode == 0 || mode == 1, "msm_transition_matrix: mode must be 0 or 1");
    MSM_REQUIRE(ctx, d_counts && d_T && d_rowsum, "msm_transition_matrix: NULL pointer");
    MSM_REQUIRE(ctx, mode == 0 || (d_active && d_inv_map && d_n_active),
                "msm_transition_matrix: mode 1 needs d_active, d_inv_map and d_n_active");
    msm_status rs = msm_reserve_scratch(ctx, (size_t)k * sizeof(double));
    if (rs != MSM_OK) return rs;
    double* colsum = (double*)ctx->scratch;
    if (counts_are_f64)
        hipLaunchKernelGGL(rowcol_sums_kernel<double>, dim3(k), dim3(kThreads), 0, ctx->stream, (const double*)d_counts, k,
                           d_rowsum, colsum);
    else
        hipLaunchKernelGGL(rowcol_sums_kernel<long long>, dim3(k), dim3(kThreads), 0, ctx->stream,
                           (const long long*)d_counts, k, d_rowsum, colsum);
    MSM_CHECK_LAUNCH(ctx);
    if (mode == 1) {
        hipLaunchKernelGGL(active_set_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_rowsum, colsum, k, epsilon, 0,
                           d_active, d_inv_map, d_n_active);
        MSM_CHECK_LAUNCH(ctx);
        MSM_HIP(ctx, hipMemsetAsync(d_T, 0, (size_t)k * k * sizeof(double), ctx->stream));
    }
    if (counts_are_f64)
        hipLaunchKernelGGL(build_T_kernel<double>, dim3(k), dim3(kThreads), 0, ctx->stream, (const double*)d_counts, k,
                           mode, alpha, d_rowsum, d_active, d_n_active, d_T);
    else
        hipLaunchKernelGGL(build_T_kernel<long long>, dim3(k), dim3(kThreads), 0, ctx->stream,
                           (const long long*)d_counts, k, mode, alpha, d_rowsum, d_active, d_n_active, d_T);
    MSM_CHECK_LAUNCH(ctx);
    if (d_diag_mass && mode == 0) {
        hipLaunchKernelGGL(diag_mass_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_T, k, d_diag_mass);
        MSM_CHECK_LAUNCH(ctx);
    }
    return MSM_OK;
}

msm_status msm_embed_full(msm_ctx* ctx, const double* d_T_active, const double* d_pi_active,
                          const int32_t* d_inv_map, int k, double* d_T_full, double* d_pi_full) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, k >= 1 && d_T_active && d_inv_map && d_T_full, "msm_embed_full: bad arguments");
    hipLaunchKernelGGL(embed_full_kernel, dim3(k), dim3(kThreads), 0, ctx->stream, d_T_active, d_pi_active, d_inv_map,
                       k, d_T_full, d_pi_full);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

size_t msm_spectrum_workspace_bytes(int n_max, int p, int batch) {
    const size_t per = 2 * ((size_t)n_max * p + kSolveThreads) * sizeof(double) + 4 * kMaxP * sizeof(double);
    return per * (size_t)batch + 256;
}

msm_status msm_spectrum(msm_ctx* ctx, const double* d_T, int64_t t_stride, int ld, const int32_t* d_n, int n_max,
                        int batch, int p, int n_iter, int init, uint64_t seed, int n_watch, void* d_workspace,
                        double* d_ritz, double* d_pi, int64_t pi_stride, double* d_change, int32_t* d_status,
                        int n_its, const double* d_lags, double* d_its_eig, double* d_its_ts) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n_max >= 1 && batch >= 1 && ld >= n_max, "msm_spectrum: bad shape");
    MSM_REQUIRE(ctx, p >= 1 && p <= kMaxP, "msm_spectrum: need 1 <= p <= %d", kMaxP);
    MSM_REQUIRE(ctx, n_iter >= 0 && n_its >= 0 && n_its < kMaxP, "msm_spectrum: bad n_iter / n_its");
    MSM_REQUIRE(ctx, d_T && d_workspace && d_ritz && d_change && d_status, "msm_spectrum: NULL pointer");
    MSM_REQUIRE(ctx, n_its == 0 || (d_its_eig && d_its_ts), "msm_spectrum: its outputs missing");
    SpecArgs ar;
    ar.T = d_T; ar.t_stride = (size_t)t_stride; ar.ld = ld; ar.n_ptr = d_n; ar.n_fixed = n_max;
    ar.p = p; ar.n_iter = n_iter; ar.init = init; ar.seed = seed;
    const size_t zw = (size_t)n_max * p + kSolveThreads;
    ar.Z = (double*)d_workspace;
    ar.W = ar.Z + zw * batch;
    ar.zw_stride = zw;
    ar.ritz = d_ritz; ar.pi = d_pi; ar.pi_stride = (size_t)pi_stride; ar.change = d_change;
    ar.n_watch = n_watch; ar.check_gap = n_iter >= 8 ? 4 : (n_iter > 1 ? 1 : 0);
    ar.status = d_status; ar.n_its = n_its; ar.lags = d_lags; ar.its_eig = d_its_eig; ar.its_ts = d_its_ts;
    hipLaunchKernelGGL(spectrum_kernel, dim3(batch), dim3(kSolveThreads), 0, ctx->stream, ar);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

}  // extern "C"
