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

#include <algorithm>
#include <cstdio>
#include <cstdlib>

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

// mode 0 in ONE launch: row sums (the additions of rowcol_sums_kernel, in its order), T = C / row sum (build_T_kernel's
// expression) and trace(T) / k (diag_mass_kernel's additions over its 1024 threads, by the workgroup that finishes last:
// the diagonal entries travel as device-scope stores, every workgroup waits for the acknowledgement of its own before it
// takes a ticket).  Same bits as the three launches it replaces.
template <typename CT>
__global__ __launch_bounds__(kThreads) void row_normalise_kernel(const CT* __restrict__ C, int k, double* __restrict__ rowsum,
                                                                double* __restrict__ T, double* tdiag,
                                                                unsigned int* __restrict__ ticket,
                                                                double* __restrict__ diag_out) {
    __shared__ double red[16];
    __shared__ double rs_sh;
    __shared__ int is_last;
    const int i = blockIdx.x;
    double r = 0.0;
    for (int j = threadIdx.x; j < k; j += kThreads) r += cnt_as_f64(C[(size_t)i * k + j]);
    for (int off = 32; off > 0; off >>= 1) r += __shfl_down(r, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = r;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tr = 0.0;
        for (int w = 0; w < kThreads / 64; ++w) tr += red[w];
        rowsum[i] = tr;
        rs_sh = tr;
    }
    __syncthreads();
    const double rs = rs_sh;
    for (int j = threadIdx.x; j < k; j += kThreads)
        T[(size_t)i * k + j] = rs > 0.0 ? cnt_as_f64(C[(size_t)i * k + j]) / rs : 0.0;
    if (!diag_out) return;
    if (threadIdx.x == 0) {
        const double tii = rs > 0.0 ? cnt_as_f64(C[(size_t)i * k + i]) / rs : 0.0;
        __hip_atomic_store(&tdiag[i], tii, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = t == gridDim.x - 1;
        if (is_last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!is_last) return;
    for (int v0 = 0; v0 < 1024; v0 += kThreads) {          // diag_mass_kernel's 1024 threads, kThreads at a time
        const int v = v0 + threadIdx.x;
        double t = 0.0;
        for (int q = v; q < k; q += 1024) t += __hip_atomic_load(&tdiag[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
        if ((v & 63) == 0) red[v >> 6] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < 16; ++w) s += red[w];
        *diag_out = k > 0 ? s / (double)k : __builtin_nan("");
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
    double freeze_tol;    // > 0: a matrix whose `change` (from the previous call) is <= this is left as it is
    int check_gap;
    int* status;          // [batch] 0 ok, else hqr failure index
    // implied timescales (optional)
    int n_its;            // 0: skip
    const double* lags;   // [batch]
    double* its_eig;      // [batch][n_its]
    double* its_ts;       // [batch][n_its]
    // leading left Ritz vectors (optional)
    double* vecs;         // [batch][n_vecs][n_fixed]
    int n_vecs;
    const int* persist_error;   // set by spec_persist_kernel when a group barrier timed out (NULL: not used)
    int splits;           // row splits of T per product launch, 1 .. kSpecSplits (set by the host from the batch size)
};

// Batches converge unevenly (a lag scan, posterior samples): once a matrix has met the caller's
// tolerance, later calls skip it.  `change` is only written by the last kernel of a call, so every
// kernel of the next call reads the same value; a (re)start ignores it.
__device__ __forceinline__ bool spec_frozen(const SpecArgs& ar, int b) {
    return ar.freeze_tol > 0.0 && !ar.init && ar.change[b] <= ar.freeze_tol;
}


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

__device__ double spec_block_max(double v, SpecShared* sh) {
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh->red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = -INFINITY;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t = fmax(t, sh->red[i]);
        sh->bc = t;
    }
    __syncthreads();
    return sh->bc;
}

// Gram matrix M = A'B (p x p, operands n x p with row stride p) on the fp64 matrix cores.  The scalar version (a thread
// per entry walking all rows) read every operand element p times from the LDS: 8 us at n = 200, p = 32, LDS-bandwidth
// bound.  Here a wave takes a 16 x 16 tile of M and a share of the rows (16 rows per trip: four v_mfma_f64_16x16x4
// whose operand reads go out together); the shares of a tile are added in share order.  scratch: LDS, >= 3 kMaxP^2 doubles.
typedef double spec_v4f64 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void spec_gram_fast(const double* __restrict__ A, const double* __restrict__ B, int n, int p,
                                               double* M, double* scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int tp = (p + 15) >> 4, tiles = tp * tp;
    const int shares = max(1, min(nw / tiles, (3 * kMaxP * kMaxP) / (tiles * 256)));   // row shares per tile
    const int trips = (n + 15) >> 4;
    const int trips_per_share = (trips + shares - 1) / shares;
    if (wave < tiles * shares) {
        const int t = wave % tiles, sh_id = wave / tiles;
        const int a0 = (t / tp) * 16, b0 = (t - (t / tp) * tp) * 16;
        const bool aok = a0 + j < p, bok = b0 + j < p;
        const int ac = aok ? a0 + j : 0, bc = bok ? b0 + j : 0;
        spec_v4f64 acc = {0.0, 0.0, 0.0, 0.0};
        const int k_begin = sh_id * trips_per_share * 16, k_end = min(n, k_begin + trips_per_share * 16);
        for (int k0 = k_begin; k0 < k_end; k0 += 16) {
            double av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + 4 * u + g;
                const bool kok = k < k_end;
                const int kk = kok ? k : 0;
                av[u] = A[kk * p + ac];
                bv[u] = B[kk * p + bc];
                if (!(aok && kok)) av[u] = 0.0;
                if (!(bok && kok)) bv[u] = 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
        }
        double* part = scratch + (size_t)(sh_id * tiles + t) * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r) part[(g + 4 * r) * 16 + j] = acc[r];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < p * p; e += blockDim.x) {
        const int a = e / p, b = e - a * p;
        const int t = (a >> 4) * tp + (b >> 4), off = (a & 15) * 16 + (b & 15);
        double v = 0.0;
        for (int s2 = 0; s2 < shares; ++s2) v += scratch[(size_t)(s2 * tiles + t) * 256 + off];
        M[e] = v;
    }
    __syncthreads();
}

__device__ void spec_ritz(SpecShared* sh, int p, double* out_re, double* out_im) {
    // wave 0: eigenvalues of a copy of H (lanes share the row / column updates); lane 0 sorts them by
    // descending magnitude
    if (threadIdx.x < 64) {
        for (int i = threadIdx.x; i < p * p; i += 64) sh->Hw[i] = sh->H[i];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int rc = small_eig::eigenvalues_wave(sh->Hw, p, p, sh->wr, sh->wi);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (rc && threadIdx.x == 0) sh->status = rc;
    }
    if (threadIdx.x == 0) {
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

// ---- multi-workgroup driver --------------------------------------------------------------
// One subspace-iteration step is  W = T'Z  (the only O(n^2 p) part: spread over the chip) followed
// by a light single-workgroup step per matrix (sum of the row-split partials, and every few
// steps Gram + Cholesky-QR; Rayleigh-Ritz / residuals / pi / implied timescales at the end).
// T is stochastic (|lambda| <= 1) so the basis needs no rescaling between orthogonalisations.
constexpr int kSpecSplits = 16;       // most row splits of T per apply launch (fixed-order partial sums); a large batch
                                      // fills the chip without splitting and takes 1: a sixteenth of the partial-product
                                      // traffic (C4: 4 GB written and read back per iteration with 16)
constexpr int kApplyThreads = 256;    // 4 waves = 4 tiles of 64 columns

// partial[s][j][c] = sum_{i in split s} T[i][j] Z[i][c]
template <int PC>
__global__ __launch_bounds__(kApplyThreads) void spec_apply_kernel(SpecArgs ar, const double* __restrict__ Zin_all,
                                                                  double* __restrict__ partial_all, size_t part_stride) {
    const int b = blockIdx.z;
    const int n = ar.n_ptr ? ar.n_ptr[b] : ar.n_fixed;
    if (n <= 0 || spec_frozen(ar, b)) return;
    const int p = min(ar.p, n);
    const double* T = ar.T + (size_t)b * ar.t_stride;
    const double* Z = Zin_all + (size_t)b * ar.zw_stride;
    double* part = partial_all + (size_t)b * part_stride + (size_t)blockIdx.y * ((size_t)ar.n_fixed * ar.p);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = (blockIdx.x * 4 + wave) * 64 + lane;
    const int rows = (n + ar.splits - 1) / ar.splits;
    const int i0 = blockIdx.y * rows, i1 = min(n, i0 + rows);
    double acc[PC];
#pragma unroll
    for (int c = 0; c < PC; ++c) acc[c] = 0.0;
    const int jc = j < n ? j : n - 1;
    int i = i0;
    for (; i + 7 < i1; i += 8) {  // 8 independent T loads in flight per lane
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = T[(size_t)(i + u) * ar.ld + jc];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double* zr = Z + (size_t)(i + u) * p;  // wave-uniform: scalar loads
#pragma unroll
            for (int c = 0; c < PC; ++c)
                if (c < p) acc[c] = fma(t[u], zr[c], acc[c]);
        }
    }
    for (; i < i1; ++i) {
        const double t = T[(size_t)i * ar.ld + jc];
        const double* zr = Z + (size_t)i * p;
#pragma unroll
        for (int c = 0; c < PC; ++c)
            if (c < p) acc[c] = fma(t, zr[c], acc[c]);
    }
    if (j < n) {
#pragma unroll
        for (int c = 0; c < PC; ++c)
            if (c < p) part[(size_t)j * p + c] = acc[c];
    }
}

// Diagnostic build only (tools/probe/spec_stamp_probe.hip)
#ifdef MSM_SPEC_STAMPS
__device__ unsigned long long g_spec_stamps[8];
#define SSTAMP(i) do { unsigned long long t__ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) g_spec_stamps[i] += t__ - slast__; slast__ = t__; } while (0)
#define SSTAMP_INIT unsigned long long slast__ = __builtin_amdgcn_s_memtime();
#else
#define SSTAMP(i)
#define SSTAMP_INIT
#endif

// mode bits
constexpr int kStepInit = 1;     // seeded basis -> orthonormal Z (no partials involved)
constexpr int kStepOrtho = 2;    // Z <- orth(W)
constexpr int kStepTwice = 4;    // with kStepOrtho: a second Cholesky-QR pass over Z (the basis the Rayleigh-Ritz step gets
                                 // when the iterations ran on a power of T: one pass leaves ~cond(W)^2 eps of non-orthogonality)
constexpr int kStepFinish = 8;   // Rayleigh-Ritz, residuals, pi, implied timescales

// Z = W R^-1 for upper-triangular R (p x p in LDS): R^-1 by one wave (lane = column), then a
// p-term dot product per element.
// Cholesky factor R (upper, G = R'R) of the p x p Gram matrix (both in LDS, row stride p) by ONE wave: lane j keeps
// column j of R in registers (compile-time indices through template recursion -- `#pragma unroll` with a break left the
// arrays in scratch memory), the pivot-row elements come by v_readlane, and the dependent chain holds only the
// multiply-adds, the square root and the division: 13 instead of 21 us at p = 32, 3.7 instead of 5.5 us at p = 12
// (tools/run/spec_stamps.sh).  Same operations in the same order as the LDS walk it replaces: the same bits.
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)b, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int C>
__device__ __forceinline__ void chol_step(double (&r)[kMaxP], const double* G, double* R, int p, int j) {
    if constexpr (C < kMaxP) {
        if (C < p) {
            double v = (j < p && j >= C) ? G[C * p + j] : 0.0;
#pragma unroll
            for (int m = 0; m < C; ++m) v = fma(-readlane_f64(r[m], C), r[m], v);
            double diag = readlane_f64(v, C);
            if (!(diag > 1e-300)) diag = 1e-300;
            const double rcc = sqrt(diag);
            r[C] = j > C ? v / rcc : (j == C ? rcc : 0.0);
            if (j < p) R[C * p + j] = r[C];
            chol_step<C + 1>(r, G, R, p, j);
        }
    }
}
__device__ __forceinline__ void chol_wave(const double* G, double* R, int p) {
    double r[kMaxP];
    chol_step<0>(r, G, R, p, threadIdx.x & 63);
}

// R^-1 (upper) by one wave, lane c = column c by back substitution, from a copy of R padded with zeros (ones on the
// diagonal) to PM x PM: no guards inside the sums, so every LDS read has a compile-time address and leaves the dependent
// chain (the walk over the compact copy paid an LDS round trip per term: 19 us at p = 32).  The padded terms add
// fma(0, x, a) = a: the bits of the compact walk.
template <int PM, int M>
__device__ __forceinline__ void rinv_pad_step(double (&x)[PM], const double* Rp, int c) {
    if constexpr (M >= 0) {
        double a = 0.0;
#pragma unroll
        for (int l = M + 1; l < PM; ++l) a = fma(Rp[M * PM + l], x[l], a);      // x[l] = 0 for l > c
        const double num = c == M ? 1.0 : -a;
        x[M] = M <= c ? num / Rp[M * PM + M] : 0.0;
        rinv_pad_step<PM, M - 1>(x, Rp, c);
    }
}
template <int PM>
__device__ __forceinline__ void rinv_wave_pad(const double* R, double* Rpad, double* Rinv, int p) {
    const int c = threadIdx.x & 63;
    for (int e = c; e < PM * PM; e += 64) {
        const int m = e / PM, l = e - m * PM;
        Rpad[e] = (m < p && l < p) ? R[m * p + l] : (m == l ? 1.0 : 0.0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double x[PM];
    rinv_pad_step<PM, PM - 1>(x, Rpad, c);
    if (c < p) {
#pragma unroll
        for (int m = 0; m < PM; ++m)
            if (m < p) Rinv[m * p + c] = x[m];
    }
}
// Rpad: LDS work space of kMaxP^2 doubles (the Gram matrix is no longer needed when this runs)
__device__ __forceinline__ void rinv_wave(const double* R, double* Rpad, double* Rinv, int p) {
    if (p <= 16) rinv_wave_pad<16>(R, Rpad, Rinv, p);
    else rinv_wave_pad<kMaxP>(R, Rpad, Rinv, p);
}

// Z = W Rinv (n x p times the upper-triangular p x p in LDS) on the matrix cores: a wave takes 16 rows and all their
// columns (two accumulator tiles at most), so W and Z may be the same array.  The scalar version kept one row per
// thread and walked R^-1 element by element behind LDS latency: 35 us at n = 200, p = 32 with a fifth of the threads busy.
__device__ __forceinline__ void spec_times_rinv(const double* Rinv, int p, const double* W, double* Z, int n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int tp = (p + 15) >> 4;     // 1 or 2
    for (int rb = wave; rb * 16 < n; rb += nw) {
        const int i0 = rb * 16;
        const bool rok = i0 + j < n;
        const int ri = rok ? i0 + j : 0;
        spec_v4f64 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        for (int k0 = 0; k0 < p; k0 += 16) {
            double av[4], b0[4], b1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + 4 * u + g;
                const bool kok = k < p;
                const int kk = kok ? k : 0;
                av[u] = W[ri * p + kk];
                if (!(rok && kok)) av[u] = 0.0;
                b0[u] = (kok && j < p) ? Rinv[kk * p + (j < p ? j : 0)] : 0.0;
                b1[u] = (kok && 16 + j < p) ? Rinv[kk * p + (16 + j < p ? 16 + j : 0)] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], b0[u], acc0, 0, 0, 0);
                if (tp > 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], b1[u], acc1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = i0 + g + 4 * r;
            if (row < n) {
                if (j < p) Z[row * p + j] = acc0[r];
                if (16 + j < p) Z[row * p + 16 + j] = acc1[r];
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void spec_apply_rinv(SpecShared* sh, int p, const double* W, double* Z, int n) {
    double* Rinv = sh->Hw;  // p x p
    if (threadIdx.x < 64) rinv_wave(sh->R, sh->G, Rinv, p);
    __syncthreads();
    spec_times_rinv(Rinv, p, W, Z, n);
}

__device__ void spec_cholesky(SpecShared* sh, int p) {
    if (threadIdx.x < 64) chol_wave(sh->G, sh->R, p);
    __syncthreads();
}


// ---- persistent driver: the subspace iterations of a matrix inside ONE launch ------------------------------
// The loop above is two latency-bound launches per iteration (~77 us at n = 500: 3.9 ms for 50 iterations, 1.5 s for
// the 5000 matrices of a lag scan with posterior samples).  Here a GROUP of G workgroups owns a matrix for all its
// iterations: member r keeps columns J_r of T (= rows of T') in its LDS, so  W[J_r, :] = T[:, J_r]' Z  needs no
// partial sums across workgroups; per iteration the members exchange their rows of W and the p x p Gram partials
// through the L2 behind ONE barrier of the group; every member then sums the Gram partials in member order,
// factorises the same p x p matrix and orthonormalises all of W itself (identical bits on all members, no broadcast).
// Groups are resident together (cooperative launch) and placed on ONE XCD each (workgroup ids that differ by
// multiples of 8 share an XCD: the barriers and the exchanged rows stay in that XCD's L2); a group works through the
// matrices q, q + groups, ... of the batch (used for small batches only: see msm_spectrum).  A barrier that is not met
// within ~1 s raises an error flag instead of spinning for ever.
struct PersistArgs {
    double* gram;          // [groups][2][G][kMaxP * kMaxP] Gram partials of odd / even steps
    unsigned* counters;    // [groups] arrivals (monotonic)
    int* error;            // [1]
    int G, groups, cols, batch, n_iter;
    int twice;             // second Cholesky-QR pass over the final basis (iterations on a power of T)
    double* bufA;          // [batch] basis in / out (zw_stride apart)
    double* bufB;          // [batch] the other basis buffer
};

// The members of a group share ONE L2 (same XCD), so what a barrier must add to the arrival counter is: my stores have
// left the CU (s_waitcnt: the vector L1 writes through) and, after the wait, my CU's L1 holds nothing stale
// (buffer_inv sc1).  An agent-scope release would also write the L2 back to memory (buffer_wbl2) for readers on other
// XCDs: ~20 us per barrier, none of which this exchange needs.
__device__ __forceinline__ void group_barrier(unsigned* ctr, unsigned G, unsigned& epoch, int* error) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // every wave: its own stores have left the CU
    __syncthreads();
    if (threadIdx.x == 0) {
        epoch += 1;
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * G) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 40000000u) {
                atomicExch(error, 1);
                break;
            }
        }
        asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

struct PersistShared {     // (SpecShared carries 49 KB of Rayleigh-Ritz work space this kernel has no use for)
    double G[kMaxP * kMaxP];
    double R[kMaxP * kMaxP];
    double Rinv[kMaxP * kMaxP];
};

__device__ __forceinline__ void persist_cholesky(PersistShared* sh, int p) {
    if (threadIdx.x < 64) chol_wave(sh->G, sh->R, p);
    __syncthreads();
}

__global__ __launch_bounds__(kSolveThreads) void spec_persist_kernel(SpecArgs ar, PersistArgs pa) {
    extern __shared__ __attribute__((aligned(16))) double plds[];
    __shared__ PersistShared sh;
    const int tid = threadIdx.x;
    // workgroup id -> (group q, member r): ids that differ by multiples of 8 share an XCD, so group q lives on XCD
    // q % 8 and its members are the ids  ((q / 8) G + r) 8 + q % 8
    const int xcd = blockIdx.x % 8, t = blockIdx.x / 8;
    const int q = (t / pa.G) * 8 + xcd, r = t % pa.G;
    if (q >= pa.groups) return;       // ids of the padded grid without a group
    unsigned* ctr = pa.counters + q;
    unsigned epoch = 0;
    const int n_max = ar.n_fixed;
    const int ldt = pa.cols | 1, ldz = ar.p | 1;         // odd row strides: the lanes of a wave read 32 different rows
    double* Tl = plds;                                   // [n][ldt]
    double* Zl = Tl + (size_t)n_max * ldt;               // [n][ldz]
    double* Wl = Zl + (size_t)n_max * ldz;               // [cols][p]
    for (int b = q; b < pa.batch; b += pa.groups) {
        const int n = ar.n_ptr ? ar.n_ptr[b] : ar.n_fixed;
        if (n <= 0 || spec_frozen(ar, b)) continue;      // the same decision on every member
        const int p = min(ar.p, n);
        if (p == n) continue;       // the basis spans the whole space: nothing to iterate (see spec_step_kernel)
        const double* T = ar.T + (size_t)b * ar.t_stride;
        const int j0 = r * pa.cols, nj = max(0, min(n, j0 + pa.cols) - j0);
        // this member's columns of T
        for (int e = tid; e < n * pa.cols; e += kSolveThreads) {
            const int i = e / pa.cols, jj = e - i * pa.cols;
            Tl[i * ldt + jj] = jj < nj ? T[(size_t)i * ar.ld + j0 + jj] : 0.0;
        }
        double* zbuf = pa.bufA + (size_t)b * ar.zw_stride;          // orthonormal basis in, orthonormal basis out
        double* wbuf[2] = {pa.bufB + (size_t)b * ar.zw_stride, zbuf};   // the exchanged (un-normalised) W of odd / even steps
        double* gram = pa.gram + (size_t)q * 2 * pa.G * (kMaxP * kMaxP);
        for (int e = tid; e < n * p; e += kSolveThreads) {
            const int i = e / p;
            Zl[i * ldz + (e - i * p)] = zbuf[e];
        }
        __syncthreads();
        // One barrier per iteration: the members exchange their rows of W = T'Z (NOT yet orthonormal) and their Gram
        // partials together; behind the barrier every member has all of W and the whole Gram matrix, factorises it and
        // forms the orthonormal Z = W R^-1 for ALL rows itself (n p^2 flops: nothing) before its next product.
        for (int it = 0; it < pa.n_iter; ++it) {
            // W[jj][c] = sum_i T[i][j0 + jj] Z[i][c]: tiles of 2 x 4 outputs; the 32 lanes of a half wave take the rows
            // i = lane, lane + 32, ... of one tile and add up by shuffles in a fixed order
            {
                const int tj = (pa.cols + 1) / 2, tc = (p + 3) / 4, tiles = tj * tc;
                const int sp = tid & 31;
                for (int tile = tid >> 5; tile < tiles; tile += kSolveThreads / 32) {
                    double acc[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
                    const int jj0 = (tile / tc) * 2, c0 = (tile % tc) * 4;
                    const bool two = jj0 + 1 < pa.cols;
                    for (int i = sp; i < n; i += 32) {
                        const double t0 = Tl[i * ldt + jj0], t1 = two ? Tl[i * ldt + jj0 + 1] : 0.0;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const double z = c0 + c < p ? Zl[i * ldz + c0 + c] : 0.0;
                            acc[0][c] = fma(t0, z, acc[0][c]);
                            acc[1][c] = fma(t1, z, acc[1][c]);
                        }
                    }
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            double v = acc[a][c];
#pragma unroll
                            for (int off = 1; off < 32; off <<= 1) v += __shfl_xor(v, off, 64);
                            const int jj = jj0 + a, cc = c0 + c;
                            if (sp == 0 && jj < pa.cols && cc < p) Wl[jj * p + cc] = jj < nj ? v : 0.0;
                        }
                }
            }
            __syncthreads();
            double* wx = wbuf[it & 1];
            double* gx = gram + (size_t)(it & 1) * pa.G * (kMaxP * kMaxP);
            for (int e = tid; e < nj * p; e += kSolveThreads) wx[(size_t)j0 * p + e] = Wl[e];
            for (int e = tid; e < p * p; e += kSolveThreads) {
                const int a = e / p, c = e - a * p;
                double v = 0.0;
                for (int jj = 0; jj < nj; ++jj) v = fma(Wl[jj * p + a], Wl[jj * p + c], v);
                gx[(size_t)r * (kMaxP * kMaxP) + e] = v;
            }
            group_barrier(ctr, pa.G, epoch, pa.error);
            for (int e = tid; e < n * p; e += kSolveThreads) {
                const int i = e / p;
                Zl[i * ldz + (e - i * p)] = wx[e];
            }
            for (int e = tid; e < p * p; e += kSolveThreads) {
                double v = 0.0;
                for (int m = 0; m < pa.G; ++m) v += gx[(size_t)m * (kMaxP * kMaxP) + e];   // member order: the same bits everywhere
                sh.G[e] = v;
            }
            __syncthreads();
          for (int pass = 0; pass < ((pa.twice && it == pa.n_iter - 1) ? 2 : 1); ++pass) {
            if (pass == 1) {            // Gram matrix of the basis just formed: every member has all of it
                for (int e = tid; e < p * p; e += kSolveThreads) {
                    const int a = e / p, c = e - a * p;
                    double v = 0.0;
                    for (int i = 0; i < n; ++i) v = fma(Zl[i * ldz + a], Zl[i * ldz + c], v);
                    sh.G[e] = v;
                }
                __syncthreads();
            }
            persist_cholesky(&sh, p);
            double* Rinv = sh.Rinv;     // R^-1 by one wave (lane = column)
            if (tid < 64) rinv_wave(sh.R, sh.G, Rinv, p);
            __syncthreads();
            // Z = W R^-1, row by row in place (a thread owns a row: no hazard)
            for (int i = tid; i < n; i += kSolveThreads) {
                double w[kMaxP];
#pragma unroll
                for (int m = 0; m < kMaxP; ++m) w[m] = m < p ? Zl[i * ldz + m] : 0.0;
                for (int c = 0; c < p; ++c) {
                    double v = 0.0;
                    for (int m = 0; m <= c; ++m) v = fma(w[m], Rinv[m * p + c], v);
                    Zl[i * ldz + c] = v;
                }
            }
            __syncthreads();
          }
        }
        // the finishing launch expects the orthonormal basis in bufA; the last exchange may still be read from there
        group_barrier(ctr, pa.G, epoch, pa.error);
        for (int e = tid; e < nj * p; e += kSolveThreads) {
            const int jj = e / p;
            zbuf[(size_t)j0 * p + e] = Zl[(j0 + jj) * ldz + (e - jj * p)];
        }
        group_barrier(ctr, pa.G, epoch, pa.error);   // nobody overwrites Tl / the Gram partials of a matrix still in use
    }
}

// Z <- Z R^-1 in place (a wave reads all of its 16 rows before it writes them)
__device__ __forceinline__ void spec_apply_rinv_inplace(SpecShared* sh, int p, double* Z, int n) { spec_apply_rinv(sh, p, Z, Z, n); }

// Complex Ritz pairs theta = a +- i b get a true residual like the real ones (their change between two launches was
// the measure before: a solve with a complex pair among the watched values could not finish in one launch).  A pair whose
// imaginary part is at rounding level relative to its size keeps the change measure (the partner vector divides by b).
__device__ __forceinline__ bool spec_complex_ok(double a, double b) { return fabs(b) > 1e-7 * fmax(1.0, fabs(a)); }
// v = (a u - H u) / b by one wave (all lanes call it; u, v: p doubles in LDS, v must not alias u)
__device__ __forceinline__ void spec_complex_partner(const double* H, int p, double a, double b, const double* u, double* v) {
    const int lane = threadIdx.x & 63;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double hv = 0.0;
    if (lane < p)
        for (int c = 0; c < p; ++c) hv = fma(H[lane * p + c], u[c], hv);
    const double out = lane < p ? (a * u[lane] - hv) / b : 0.0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < p) v[lane] = out;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Z: current (orthonormal unless between orthogonalisations) basis; Wb: the other buffer.
template <bool lds_w>
__global__ __launch_bounds__(kSolveThreads) void spec_step_kernel(SpecArgs ar, int mode, double* __restrict__ Zall,
                                                                 double* __restrict__ Wall,
                                                                 const double* __restrict__ partial_all,
                                                                 size_t part_stride) {
    __shared__ SpecShared sh;
    extern __shared__ __attribute__((aligned(16))) double w_lds[];  // n_fixed * p doubles when lds_w
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int n = ar.n_ptr ? ar.n_ptr[b] : ar.n_fixed;
    double* Z = Zall + (size_t)b * ar.zw_stride;
    double* Wg = Wall + (size_t)b * ar.zw_stride;
    // A lone workgroup pays ~2 us for every dependent trip to memory another XCD has just
    // written: W = T'Z is summed into LDS in one round of loads and stays there for the Gram
    // matrix, the triangular solve and the residuals.
    double* W = lds_w ? w_lds : Wg;
    double* ritz = ar.ritz + (size_t)b * 4 * kMaxP;
    if (spec_frozen(ar, b)) return;   // uniform over the workgroup; outputs of the previous call stand
    if (tid == 0) sh.status = 0;
    __syncthreads();
    if (n <= 0) {
        if (mode & kStepFinish) {
            if (tid == 0) { ar.change[b] = 0.0; ar.status[b] = 0; }
            for (int i = tid; i < 4 * kMaxP; i += blockDim.x) ritz[i] = 0.0;
            for (int i = tid; i < ar.n_its; i += blockDim.x) {
                ar.its_eig[(size_t)b * ar.n_its + i] = __builtin_nan("");
                ar.its_ts[(size_t)b * ar.n_its + i] = __builtin_nan("");
            }
        }
        return;
    }
    const int p = min(ar.p, n);
    // A basis of the WHOLE space (p = n: small matrices) needs no iterations -- the Rayleigh-Ritz step on any orthonormal
    // basis of it returns the eigenvalues of T -- and must not get them: T'Z of a singular T is rank-deficient and its
    // Cholesky-QR breaks down (NaN Ritz values, "hqr did not converge").  The seeded basis stays as it is until the finish.
    if (p == n && !(mode & (kStepInit | kStepFinish))) return;
    if (mode & kStepInit) {
        for (int e = tid; e < n * p; e += blockDim.x) {
            const int c = e % p;
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
        spec_gram_fast(W, W, n, p, sh.G, sh.Hw);
        spec_cholesky(&sh, p);
        spec_apply_rinv(&sh, p, W, Z, n);
        return;
    }
    SSTAMP_INIT
    // W = sum of the row-split partials, fixed order
    {
        const double* part = partial_all + (size_t)b * part_stride;
        const size_t split_stride = (size_t)ar.n_fixed * ar.p;
        int e = tid;
        if (ar.splits == kSpecSplits) {
            for (; e + (int)blockDim.x < n * p; e += 2 * blockDim.x) {  // 2 x kSpecSplits loads in flight
                double v0[kSpecSplits], v1[kSpecSplits];
#pragma unroll
                for (int sp = 0; sp < kSpecSplits; ++sp) {
                    v0[sp] = part[sp * split_stride + e];
                    v1[sp] = part[sp * split_stride + e + blockDim.x];
                }
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int sp = 0; sp < kSpecSplits; ++sp) { a0 += v0[sp]; a1 += v1[sp]; }
                W[e] = a0;
                W[e + blockDim.x] = a1;
            }
            for (; e < n * p; e += blockDim.x) {
                double v = 0.0;
#pragma unroll
                for (int sp = 0; sp < kSpecSplits; ++sp) v += part[sp * split_stride + e];
                W[e] = v;
            }
        } else {
            // few splits (large batches: one): eight elements in flight per thread
            for (; e + 7 * (int)blockDim.x < n * p; e += 8 * blockDim.x) {
                double a[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) a[q] = part[e + q * blockDim.x];
                for (int sp = 1; sp < ar.splits; ++sp)
#pragma unroll
                    for (int q = 0; q < 8; ++q) a[q] += part[sp * split_stride + e + q * blockDim.x];
#pragma unroll
                for (int q = 0; q < 8; ++q) W[e + q * blockDim.x] = a[q];
            }
            for (; e < n * p; e += blockDim.x) {
                double v = part[e];
                for (int sp = 1; sp < ar.splits; ++sp) v += part[sp * split_stride + e];
                W[e] = v;
            }
        }
        __syncthreads();
        if (lds_w && !(mode & (kStepOrtho | kStepFinish))) {  // plain power step: the next apply reads global
            for (int q = tid; q < n * p; q += blockDim.x) Wg[q] = W[q];
        }
    }
    SSTAMP(0);
    if (mode & kStepOrtho) {
        spec_gram_fast(W, W, n, p, sh.G, sh.Hw);
        SSTAMP(1);
        spec_cholesky(&sh, p);
        SSTAMP(2);
        spec_apply_rinv(&sh, p, W, Z, n);
        SSTAMP(3);
        if (mode & kStepTwice) {
            spec_gram_fast(Z, Z, n, p, sh.G, sh.Hw);
            spec_cholesky(&sh, p);
            spec_apply_rinv_inplace(&sh, p, Z, n);
        }
    }
    if (!(mode & kStepFinish)) {
        if (tid == 0 && sh.status) ar.status[b] = sh.status;
        return;
    }
    // ---- Rayleigh-Ritz on the (orthonormal) basis Z with W = T'Z.  The values the previous call left
    // become the "previous" ones (complex pairs are judged by their change across calls: one small
    // nonsymmetric eigensolve per call instead of two).
    for (int i = tid; i < 2 * kMaxP; i += blockDim.x) ritz[2 * kMaxP + i] = ritz[i];
    __syncthreads();
    spec_gram_fast(Z, W, n, p, sh.H, sh.Hw);
    spec_ritz(&sh, p, ritz, ritz + kMaxP);
    for (int i = p + tid; i < kMaxP; i += blockDim.x) { ritz[i] = 0.0; ritz[kMaxP + i] = 0.0; }
    __syncthreads();
    // Convergence measure: the true residual ||T'x - theta x|| / ||x|| of every watched REAL
    // Ritz pair (x = Z y, T'x = W y); for complex values, the change since the last check.
    // (Changes of Ritz values alone stagnate on clustered spectra and would stop too early.)
    // The Ritz vectors these steps need (one per watched real Ritz value, one for the stationary distribution) are
    // inverse iterations of ~50 us each by ONE wave: when their work space fits they run side by side on separate
    // waves (7 vectors at p = 12: 0.43 -> 0.1 ms for this launch); wider subspaces take them one after the other.
    const int nw = min(ar.n_watch, p);
    const int per = p * p + 3 * p;                          // [y p | work p p + 2 p] of one inverse iteration
    const bool side_by_side = (nw + 1) * per <= 3 * kMaxP * kMaxP && nw + 1 <= (int)(blockDim.x >> 6);
    int pi_id = 0;
    {
        double bd = 1e300;
        for (int i = 0; i < p; ++i) {
            const double dr = sh.wr[i] - 1.0, di = sh.wi[i];
            const double dd = dr * dr + di * di;
            if (dd < bd) { bd = dd; pi_id = i; }
        }
    }
    if (side_by_side) {
        const int w = tid >> 6;
        if (w < nw && ritz[kMaxP + w] == 0.0)
            small_eig::eigenvector_wave(sh.H, p, p, ritz[w], sh.Hw + w * per, sh.Hw + w * per + p);
        else if (w < nw && spec_complex_ok(ritz[w], ritz[kMaxP + w])) {
            double* u = sh.Hw + w * per;
            small_eig::eigenvector_wave(sh.H, p, p, ritz[w], u, u + p, ritz[kMaxP + w]);
            spec_complex_partner(sh.H, p, ritz[w], ritz[kMaxP + w], u, u + p);      // v into the (now free) work space
        }
        else if (w == nw && ar.pi)
            small_eig::eigenvector_wave(sh.H, p, p, sh.wr[pi_id], sh.Hw + w * per, sh.Hw + w * per + p);
        __syncthreads();
    }
    {
        double worst = 0.0;
        for (int wv = 0; wv < nw; ++wv) {
            const double th_re = ritz[wv], th_im = ritz[kMaxP + wv];
            if (th_im != 0.0) {
                if (spec_complex_ok(th_re, th_im)) {
                    // true residual of the complex pair x = Z (u + i v):  T'x - theta x  in real arithmetic
                    const double* uv = sh.Hw + wv * per;       // [u p | v p]
                    if (!side_by_side) {
                        if (tid < 64) {
                            small_eig::eigenvector_wave(sh.H, p, p, th_re, sh.y, sh.Hw + p, th_im);
                            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            for (int c = tid; c < p; c += 64) sh.Hw[c] = sh.y[c];
                            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            spec_complex_partner(sh.H, p, th_re, th_im, sh.Hw, sh.Hw + p);
                        }
                        __syncthreads();
                        uv = sh.Hw;
                    }
                    double rn = 0.0, xn = 0.0;
                    for (int i = tid; i < n; i += blockDim.x) {
                        double xr = 0.0, xi = 0.0, tr = 0.0, ti = 0.0;
                        for (int c = 0; c < p; ++c) {
                            const double z = Z[(size_t)i * p + c], w_ = W[(size_t)i * p + c];
                            xr = fma(z, uv[c], xr);
                            xi = fma(z, uv[p + c], xi);
                            tr = fma(w_, uv[c], tr);
                            ti = fma(w_, uv[p + c], ti);
                        }
                        const double rr = tr - (th_re * xr - th_im * xi), ri = ti - (th_im * xr + th_re * xi);
                        rn = fma(rr, rr, fma(ri, ri, rn));
                        xn = fma(xr, xr, fma(xi, xi, xn));
                    }
                    rn = spec_block_sum(rn, &sh);
                    xn = spec_block_sum(xn, &sh);
                    // (the plane comes from the SQUARED shifted matrix: with another eigenvalue very close to the pair
                    // its residual floors near eps / gap^2; the change between launches then still decides)
                    double res = sqrt(rn / fmax(xn, 1e-300));
                    if (!ar.init) {
                        const double dr = th_re - ritz[2 * kMaxP + wv], di = th_im - ritz[3 * kMaxP + wv];
                        res = fmin(res, sqrt(dr * dr + di * di) / fmax(sqrt(th_re * th_re + th_im * th_im), 1e-300));
                    }
                    worst = fmax(worst, res);
                    continue;
                }
                const double dr = th_re - ritz[2 * kMaxP + wv], di = th_im - ritz[3 * kMaxP + wv];
                const double mag = sqrt(th_re * th_re + th_im * th_im);
                const double ch = ar.init ? 1.0 : sqrt(dr * dr + di * di) / fmax(mag, 1e-300);   // no history yet
                worst = fmax(worst, ch);
                continue;
            }
            const double* yv = sh.y;
            if (side_by_side) yv = sh.Hw + wv * per;
            else {
                if (tid < 64) small_eig::eigenvector_wave(sh.H, p, p, th_re, sh.y, sh.Hw);
                __syncthreads();
            }
            double rn = 0.0, xn = 0.0;
            for (int i = tid; i < n; i += blockDim.x) {
                double xv = 0.0, tv = 0.0;
                for (int c = 0; c < p; ++c) {
                    xv = fma(Z[(size_t)i * p + c], yv[c], xv);
                    tv = fma(W[(size_t)i * p + c], yv[c], tv);
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
        const double* yv = sh.y;
        if (side_by_side) yv = sh.Hw + nw * per;
        else {
            if (tid < 64) small_eig::eigenvector_wave(sh.H, p, p, sh.wr[pi_id], sh.y, sh.Hw);
            __syncthreads();
        }
        double* pi = ar.pi + (size_t)b * ar.pi_stride;
        double part = 0.0;
        for (int i = tid; i < n; i += blockDim.x) {
            double v = 0.0;
            for (int c = 0; c < p; ++c) v = fma(Z[(size_t)i * p + c], yv[c], v);
            pi[i] = v;
            part += v;
        }
        const double tot = spec_block_sum(part, &sh);
        for (int i = tid; i < n; i += blockDim.x) pi[i] = pi[i] / tot;
    }
    __syncthreads();
    // leading left eigenvectors (x' T = theta x'), Ritz values by descending magnitude (the order
    // deeptime's eigenvectors() returns); unit 2-norm, the component of largest magnitude (lowest
    // index on ties) positive; NaN for a complex pair or beyond the subspace
    if (ar.vecs && ar.n_vecs > 0) {
        if (tid == 0) {
            for (int i = 0; i < p; ++i) {
                const double mi = sh.wr[i] * sh.wr[i] + sh.wi[i] * sh.wi[i];
                int rank = 0;
                for (int j = 0; j < p; ++j) {
                    const double mj = sh.wr[j] * sh.wr[j] + sh.wi[j] * sh.wi[j];
                    rank += (mj > mi) || (mj == mi && j < i);
                }
                sh.order[rank] = i;
            }
        }
        __syncthreads();
        for (int q = 0; q < ar.n_vecs; ++q) {
            double* out = ar.vecs + ((size_t)b * ar.n_vecs + q) * ar.n_fixed;
            const int id = q < p ? sh.order[q] : -1;
            if (id < 0 || sh.wi[id] != 0.0) {
                for (int i = tid; i < n; i += blockDim.x) out[i] = __builtin_nan("");
                continue;
            }
            __syncthreads();
            if (tid < 64) small_eig::eigenvector_wave(sh.H, p, p, sh.wr[id], sh.y, sh.Hw);
            __syncthreads();
            double nn = 0.0, big = 0.0;
            for (int i = tid; i < n; i += blockDim.x) {
                double v = 0.0;
                for (int c = 0; c < p; ++c) v = fma(Z[(size_t)i * p + c], sh.y[c], v);
                out[i] = v;
                nn = fma(v, v, nn);
                big = fmax(big, fabs(v));
            }
            nn = spec_block_sum(nn, &sh);
            big = spec_block_max(big, &sh);
            double first = -(double)n;                       // -(lowest index that attains the maximum)
            for (int i = tid; i < n; i += blockDim.x)
                if (fabs(out[i]) == big) { first = -(double)i; break; }
            first = spec_block_max(first, &sh);
            const int lead = min(n - 1, max(0, (int)(-first)));
            const double scale = (out[lead] < 0.0 ? -1.0 : 1.0) / sqrt(fmax(nn, 1e-300));
            __syncthreads();                                 // every thread has read out[lead]
            for (int i = tid; i < n; i += blockDim.x) out[i] *= scale;
        }
        __syncthreads();
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
    if (tid == 0 && sh.status) ar.status[b] = sh.status;
    if (tid == 0 && ar.persist_error && *ar.persist_error) ar.status[b] = 777;   // a group barrier of the persistent launch timed out
}


// ---- powers of the transition matrices (the iteration operator of msm_spectrum) -----------------------------------
// C_b = A_b A_b for a batch of packed square matrices of order n_b <= n_fixed (row stride ld), on the fp64 matrix
// cores: a 64 x 64 tile of C per workgroup, 16 columns of A / rows of A staged through the LDS per trip, wave w owns
// the rows 16 w .. 16 w + 15 of the tile (four accumulator tiles).  Entries outside the n_b x n_b block are written
// as zeros.  A subspace iteration with T^4 needs a quarter of the iterations of one with T (its convergence ratio is
// the fourth power) and the squarings cost 4 n^3 flop per matrix at matrix-core rate: C4's 5000 matrices of order
// 200 take ~4 ms, a k = 500 matrix 0.1 ms.
constexpr int kSqTile = 64, kSqK = 16;
__global__ __launch_bounds__(256) void square_batch_kernel(const double* __restrict__ A, size_t stride, int ld,
                                                         const int* __restrict__ n_ptr, int n_fixed,
                                                         double* __restrict__ C) {
    __shared__ double As[kSqTile][kSqK + 1];
    __shared__ double Bs[kSqK][kSqTile + 1];
    const int b = blockIdx.z;
    const int n = n_ptr ? n_ptr[b] : n_fixed;
    const int r0 = blockIdx.y * kSqTile, c0 = blockIdx.x * kSqTile;
    const double* Ab = A + (size_t)b * stride;
    double* Cb = C + (size_t)b * stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    typedef double v4f64 __attribute__((ext_vector_type(4)));
    v4f64 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = v4f64{0.0, 0.0, 0.0, 0.0};
    const bool live = r0 < n && c0 < n;        // a tile outside the block: zeros
    const int arow = tid >> 2, ak = (tid & 3) * 4;         // A tile: 64 rows x 16 k, four doubles per thread
    const int bk = tid >> 4, bc = (tid & 15) * 4;          // B tile: 16 k x 64 columns
    for (int k0 = 0; live && k0 < n; k0 += kSqK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + arow, kk = k0 + ak + i;
            As[arow][ak + i] = (r < n && kk < n) ? Ab[(size_t)r * ld + kk] : 0.0;
            const int kr = k0 + bk, c = c0 + bc + i;
            Bs[bk][bc + i] = (kr < n && c < n) ? Ab[(size_t)kr * ld + c] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double a = As[16 * wave + j][4 * u + g];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
                acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bs[4 * u + g][16 * ct + j], acc[ct], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = r0 + 16 * wave + g + 4 * r, col = c0 + 16 * ct + j;
            if (row < n_fixed && col < n_fixed) Cb[(size_t)row * ld + col] = (row < n && col < n) ? acc[ct][r] : 0.0;
        }
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
    if (mode == 0 && ctx->km_stats) {        // one launch (the scratch holds the diagonal instead of the column sums)
        unsigned int* ticket = (unsigned int*)ctx->km_stats + 3;
        if (counts_are_f64)
            hipLaunchKernelGGL(row_normalise_kernel<double>, dim3(k), dim3(kThreads), 0, ctx->stream, (const double*)d_counts, k,
                               d_rowsum, d_T, colsum, ticket, d_diag_mass);
        else
            hipLaunchKernelGGL(row_normalise_kernel<long long>, dim3(k), dim3(kThreads), 0, ctx->stream,
                               (const long long*)d_counts, k, d_rowsum, d_T, colsum, ticket, d_diag_mass);
        MSM_CHECK_LAUNCH(ctx);
        return MSM_OK;
    }
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

static bool spec_persist_enabled() {
    static const bool on = [] {
        const char* e = getenv("MSM_SPEC_PERSIST");   // MSM_SPEC_PERSIST=0: the launch-per-iteration loop (A/B timing)
        return !(e && e[0] == '0');
    }();
    return on;
}

size_t msm_spectrum_workspace_bytes(int n_max, int p, int batch) {
    // two basis buffers (+ reduction scratch tail) and the row-split partial products
    const size_t per = (2 * ((size_t)n_max * p + kSolveThreads) + (size_t)kSpecSplits * n_max * p) * sizeof(double);
    return per * (size_t)batch + 256;
}

msm_status msm_matrix_power(msm_ctx* ctx, const double* d_T, int64_t t_stride, int ld, const int32_t* d_n, int n_max,
                            int batch, int n_squarings, double* d_scratch, double* d_out) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n_max >= 1 && batch >= 1 && ld >= n_max && t_stride >= (int64_t)n_max * ld - (ld - n_max),
                "msm_matrix_power: bad shape");
    MSM_REQUIRE(ctx, n_squarings >= 1 && n_squarings <= 6, "msm_matrix_power: need 1 <= n_squarings <= 6");
    MSM_REQUIRE(ctx, d_T && d_out && (n_squarings == 1 || d_scratch), "msm_matrix_power: NULL pointer");
    MSM_REQUIRE(ctx, batch <= 65535, "msm_matrix_power: batch too large for one launch");
    const unsigned tiles = (unsigned)((n_max + kSqTile - 1) / kSqTile);
    const double* src = d_T;
    for (int sq = 0; sq < n_squarings; ++sq) {
        // the last squaring lands in d_out; the ones before alternate between the two buffers
        double* dst = ((n_squarings - 1 - sq) % 2 == 0) ? d_out : d_scratch;
        hipLaunchKernelGGL(square_batch_kernel, dim3(tiles, tiles, (unsigned)batch), dim3(256), 0, ctx->stream, src,
                           (size_t)t_stride, ld, d_n, n_max, dst);
        MSM_CHECK_LAUNCH(ctx);
        src = dst;
    }
    return MSM_OK;
}

static msm_status spectrum_impl(msm_ctx* ctx, const double* d_T, const double* d_iter, int64_t t_stride, int ld,
                                const int32_t* d_n, int n_max,
                        int batch, int p, int n_iter, int init, uint64_t seed, int n_watch, void* d_workspace,
                        double* d_ritz, double* d_pi, int64_t pi_stride, double* d_change, int32_t* d_status,
                        int n_its, const double* d_lags, double* d_its_eig, double* d_its_ts, double freeze_tol,
                        double* d_vecs, int n_vecs) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n_vecs >= 0 && (n_vecs == 0 || d_vecs), "msm_spectrum: d_vecs missing");
    MSM_REQUIRE(ctx, n_max >= 1 && batch >= 1 && ld >= n_max, "msm_spectrum: bad shape");
    MSM_REQUIRE(ctx, p >= 1 && p <= kMaxP, "msm_spectrum: need 1 <= p <= %d", kMaxP);
    MSM_REQUIRE(ctx, n_iter >= 0 && n_its >= 0 && n_its < kMaxP, "msm_spectrum: bad n_iter / n_its");
    MSM_REQUIRE(ctx, d_T && d_workspace && d_ritz && d_change && d_status, "msm_spectrum: NULL pointer");
    MSM_REQUIRE(ctx, n_its == 0 || (d_its_eig && d_its_ts), "msm_spectrum: its outputs missing");
    SpecArgs ar;
    ar.T = d_T; ar.t_stride = (size_t)t_stride; ar.ld = ld; ar.n_ptr = d_n; ar.n_fixed = n_max;
    ar.p = p; ar.n_iter = n_iter; ar.init = init; ar.seed = seed;
    const size_t zw = (size_t)n_max * p + kSolveThreads;
    double* bufA = (double*)d_workspace;              // the orthonormal basis lives here between calls
    double* bufB = bufA + zw * batch;
    double* partial = bufB + zw * batch;
    const size_t part_stride = (size_t)kSpecSplits * n_max * p;
    ar.Z = bufA; ar.W = bufB;
    ar.zw_stride = zw;
    ar.ritz = d_ritz; ar.pi = d_pi; ar.pi_stride = (size_t)pi_stride; ar.change = d_change;
    ar.n_watch = n_watch;
    ar.freeze_tol = freeze_tol;
    ar.vecs = d_vecs; ar.n_vecs = n_vecs;
    ar.status = d_status; ar.n_its = n_its; ar.lags = d_lags; ar.its_eig = d_its_eig; ar.its_ts = d_its_ts;
    ar.persist_error = nullptr;
    {   // enough workgroups for the product to fill the chip four times over, no more splits than that needs
        const int64_t per_split = (int64_t)((n_max + 255) / 256) * batch;
        ar.splits = (int)std::max<int64_t>(1, std::min<int64_t>(kSpecSplits, (4 * (int64_t)ctx->n_cu + per_split - 1) / per_split));
    }
    // orthogonalise every kOrthoEvery applications and always after the last one; compare the
    // complex Ritz values against those right after an earlier orthogonalisation.  Every step:
    // Cholesky-QR squares the condition number of W, and a metastable T damps the fast directions
    // by lambda^q -- already q = 6 un-orthogonalised applications broke the factorisation.
    constexpr int kOrthoEvery = 1;
    ar.check_gap = n_iter;
    // the ITERATIONS may run on a power of T (d_iter, msm_matrix_power: same invariant subspaces, convergence ratio to
    // that power); the Rayleigh-Ritz step, the residuals and everything reported come from T itself
    SpecArgs ar_it = ar;
    if (d_iter) ar_it.T = d_iter;
    auto apply = [&](const double* zin, const SpecArgs& aa) {
        const dim3 grid((unsigned)((n_max + 255) / 256), (unsigned)aa.splits, (unsigned)batch);
        if (p <= 8) hipLaunchKernelGGL(spec_apply_kernel<8>, grid, dim3(kApplyThreads), 0, ctx->stream, aa, zin, partial, part_stride);
        else if (p <= 16) hipLaunchKernelGGL(spec_apply_kernel<16>, grid, dim3(kApplyThreads), 0, ctx->stream, aa, zin, partial, part_stride);
        else if (p <= 24) hipLaunchKernelGGL(spec_apply_kernel<24>, grid, dim3(kApplyThreads), 0, ctx->stream, aa, zin, partial, part_stride);
        else hipLaunchKernelGGL(spec_apply_kernel<32>, grid, dim3(kApplyThreads), 0, ctx->stream, aa, zin, partial, part_stride);
    };
    const size_t w_bytes = (size_t)n_max * p * sizeof(double);
    const bool lds_w = w_bytes <= 96 * 1024;
    if (lds_w && w_bytes > 12 * 1024)
        MSM_HIP(ctx, hipFuncSetAttribute((const void*)spec_step_kernel<true>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)w_bytes));
    auto step = [&](int mode, double* z, double* w) {
        if (lds_w && !(mode & kStepInit))
            hipLaunchKernelGGL(spec_step_kernel<true>, dim3(batch), dim3(kSolveThreads), w_bytes, ctx->stream, ar, mode,
                               z, w, partial, part_stride);
        else
            hipLaunchKernelGGL(spec_step_kernel<false>, dim3(batch), dim3(kSolveThreads), 0, ctx->stream, ar, mode, z,
                               w, partial, part_stride);
    };
    if (init || !(freeze_tol > 0.0))   // frozen matrices keep their status
        MSM_HIP(ctx, hipMemsetAsync(d_status, 0, sizeof(int32_t) * batch, ctx->stream));
    if (init) step(kStepInit, bufA, bufB);
    MSM_CHECK_LAUNCH(ctx);
    // ---- the iterations: one persistent launch when the shape allows (see spec_persist_kernel)
    bool persisted = false;
    if (n_iter > 0 && spec_persist_enabled()) {
        const size_t budget = (size_t)(160 - 24 - 4) * 1024 / sizeof(double);   // LDS less PersistShared and slack
        const size_t fixed = (size_t)n_max * (p | 1);
        int cols = budget > fixed + (size_t)8 * (n_max + p + 2) ? (int)((budget - fixed) / (size_t)(n_max + p + 2)) : 0;
        cols = std::min(cols, n_max);
        auto fits = [&](int c) { return (size_t)n_max * (c | 1) + fixed + (size_t)c * p <= budget; };
        while (cols >= 8 && !fits(cols)) --cols;
        if (getenv("MSM_SPEC_DEBUG")) fprintf(stderr, "msm_spectrum: n=%d p=%d first cols=%d\n", n_max, p, cols);
        if (cols >= 8) {
            const int G = (n_max + cols - 1) / cols;
            while (cols > 1 && (n_max + cols - 2) / (cols - 1) == G && fits(cols - 1)) --cols;   // even shares: the fewest columns that keep G
            // a group lives on one XCD (32 CUs): G <= 32; groups per XCD = 32 / G, eight XCDs
            const int per_xcd = G <= 32 ? 32 / G : 0;
            // Only when every matrix gets a group at once: a large batch (a lag scan with posterior samples) already
            // fills the chip with independent single-workgroup steps, and a group per matrix would serialise it.
            const int groups = batch <= 8 * per_xcd ? batch : 0;
            if (groups >= 1) {
                const int slots = (groups + 7) / 8;          // group q = slot * 8 + xcd
                const size_t gram_bytes = (size_t)groups * 2 * G * kMaxP * kMaxP * sizeof(double);
                const size_t need = gram_bytes + (size_t)groups * sizeof(unsigned) + 64;
                msm_status rs = msm_reserve_aux(ctx, need);
                if (rs != MSM_OK) return rs;
                PersistArgs pa;
                pa.gram = (double*)ctx->aux;
                pa.counters = (unsigned*)((char*)ctx->aux + gram_bytes);
                pa.error = (int*)(pa.counters + groups);
                pa.G = G; pa.groups = groups; pa.cols = cols; pa.batch = batch; pa.n_iter = n_iter;
                pa.twice = d_iter ? 1 : 0;
                pa.bufA = bufA; pa.bufB = bufB;
                MSM_HIP(ctx, hipMemsetAsync(pa.counters, 0, (size_t)groups * sizeof(unsigned) + sizeof(int), ctx->stream));
                const size_t lds = ((size_t)n_max * (cols | 1) + (size_t)n_max * (p | 1) + (size_t)cols * p) * sizeof(double);
                MSM_HIP(ctx, hipFuncSetAttribute((const void*)spec_persist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)lds));
                // Residency: the barriers need every workgroup of the grid on a CU at once.  The grid is at most one
                // workgroup per CU (checked here against the occupancy the runtime reports), the stream is in order, and a
                // barrier that is not met within ~1 s raises the error flag instead of spinning on; a plain launch is
                // used because rocprofv3 crashes at exit of a process that made a cooperative launch (MSM_SPEC_COOP=1
                // asks for hipLaunchCooperativeKernel all the same).
                const unsigned grid = (unsigned)(slots * G * 8);
                int per_cu = 0;
                hipError_t le = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)spec_persist_kernel,
                                                                             kSolveThreads, lds);
                if (le == hipSuccess && (per_cu < 1 || grid > (unsigned)ctx->n_cu)) le = hipErrorCooperativeLaunchTooLarge;
                if (le == hipSuccess) {
                    const char* coop = getenv("MSM_SPEC_COOP");
                    if (coop && coop[0] == '1') {
                        void* kargs[] = {(void*)&ar_it, (void*)&pa};
                        le = hipLaunchCooperativeKernel((const void*)spec_persist_kernel, dim3(grid), dim3(kSolveThreads), kargs,
                                                        (unsigned)lds, ctx->stream);
                    } else {
                        hipLaunchKernelGGL(spec_persist_kernel, dim3(grid), dim3(kSolveThreads), lds, ctx->stream, ar_it, pa);
                        le = hipGetLastError();
                    }
                }
                if (getenv("MSM_SPEC_DEBUG"))
                    fprintf(stderr, "msm_spectrum: persistent launch n=%d p=%d cols=%d G=%d groups=%d lds=%zu -> %s\n", n_max, p,
                            cols, G, groups, lds, hipGetErrorString(le));
                if (le == hipSuccess) {
                    persisted = true;
                    ar.persist_error = pa.error;
                } else {
                    (void)hipGetLastError();     // not resident together on this device / in this state: the launch loop below
                }
            }
        }
    }
    // (Tried: the iterations of a large batch sub-batch by sub-batch, so that a sub-batch's matrices stay in the 256 MB
    // memory-side cache between iterations -- the product reads 1.6 GB per iteration at C4.  Slower at every sub-batch
    // size: 623 / 592 / 567 ms at 256 / 512 / 1024 matrices against 531 ms for whole-batch launches.)
    // invariant at the top of an iteration: the current basis is in `cur`
    double* cur = bufA;
    double* other = bufB;
    for (int it = 0; it < (persisted ? 0 : n_iter); ++it) {
        apply(cur, ar_it);
        const bool ortho = (it % kOrthoEvery) == kOrthoEvery - 1 || it == n_iter - 1;
        const int mode = ortho ? (kStepOrtho | (d_iter && it == n_iter - 1 ? kStepTwice : 0)) : 0;
        step(mode, cur, other);   // sum -> other; ortho: orth(other) -> cur
        if (!ortho) std::swap(cur, other);
        MSM_CHECK_LAUNCH(ctx);
    }
    if (cur != bufA) {  // keep the persistent basis in bufA
        MSM_HIP(ctx, hipMemcpyAsync(bufA, cur, zw * batch * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        cur = bufA; other = bufB;
    }
    apply(cur, ar);
    step(kStepFinish, cur, other);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_spectrum(msm_ctx* ctx, const double* d_T, int64_t t_stride, int ld, const int32_t* d_n, int n_max,
                        int batch, int p, int n_iter, int init, uint64_t seed, int n_watch, void* d_workspace,
                        double* d_ritz, double* d_pi, int64_t pi_stride, double* d_change, int32_t* d_status,
                        int n_its, const double* d_lags, double* d_its_eig, double* d_its_ts, double freeze_tol,
                        double* d_vecs, int n_vecs) {
    return spectrum_impl(ctx, d_T, nullptr, t_stride, ld, d_n, n_max, batch, p, n_iter, init, seed, n_watch, d_workspace,
                         d_ritz, d_pi, pi_stride, d_change, d_status, n_its, d_lags, d_its_eig, d_its_ts, freeze_tol, d_vecs,
                         n_vecs);
}

msm_status msm_spectrum_powered(msm_ctx* ctx, const double* d_T, const double* d_T_power, int64_t t_stride, int ld,
                                const int32_t* d_n, int n_max, int batch, int p, int n_iter, int init, uint64_t seed,
                                int n_watch, void* d_workspace, double* d_ritz, double* d_pi, int64_t pi_stride,
                                double* d_change, int32_t* d_status, int n_its, const double* d_lags, double* d_its_eig,
                                double* d_its_ts, double freeze_tol, double* d_vecs, int n_vecs) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, d_T_power, "msm_spectrum_powered: d_T_power missing");
    return spectrum_impl(ctx, d_T, d_T_power, t_stride, ld, d_n, n_max, batch, p, n_iter, init, seed, n_watch, d_workspace,
                         d_ritz, d_pi, pi_stride, d_change, d_status, n_its, d_lags, d_its_eig, d_its_ts, freeze_tol, d_vecs,
                         n_vecs);
}

}  // extern "C"
