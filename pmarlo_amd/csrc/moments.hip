// Column moments (mean / std) and the fused standardise+project kernel.
//
// Both are single passes over X [n, ld] (f32 or f64) and HBM-bound:
//   moments : read F*s bytes per frame, 2F flop
//   project : read F*s, write d*8 bytes per frame, 2*F*d flop (d << F)
// Lanes span consecutive features of a row so every wave load is a contiguous
// run of the row-major matrix.  Reductions are two-stage (per-workgroup slab in
// scratch, then one fixed-order pass), so results are run-to-run reproducible.
#include "common.h"

namespace {

constexpr int kThreads = 256;

template <typename T>
__device__ __forceinline__ double ld_f64(const T* p) { return (double)(*p); }

// ---------------------------------------------------------------------------
// moments: per column  cnt = #finite-or-inf (non-NaN), S1 = sum(x - shift), S2 = sum((x - shift)^2)
// shift = caller-provided vector or row 0 of X (NaN -> 0).
// partial layout: [block][3][F]
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kThreads) void moments_partial_kernel(
    const T* __restrict__ x, int64_t n, int F, int64_t ld, const double* __restrict__ shift_in,
    int tf, int64_t rows_per_block, double* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* red = reinterpret_cast<double*>(smem_raw);  // [3][kThreads]
    const int tid = threadIdx.x;
    const int fx = tid % tf;
    const int ry = tid / tf;
    const int rp = kThreads / tf;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(r0 + rows_per_block, n);
    double* out = partial + (size_t)blockIdx.x * 3 * F;
    for (int f0 = 0; f0 < F; f0 += tf) {
        const int f = f0 + fx;
        double cnt = 0.0, s1 = 0.0, s2 = 0.0;
        if (f < F) {
            double sh;
            if (shift_in) sh = shift_in[f];
            else { sh = ld_f64(x + f); if (sh != sh) sh = 0.0; }
            int64_t r = r0 + ry;
            // 4 independent loads in flight per lane
            for (; r + 3 * rp < r1; r += 4 * rp) {
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = ld_f64(x + (r + (int64_t)u * rp) * ld + f);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (v[u] == v[u]) { const double dlt = v[u] - sh; cnt += 1.0; s1 += dlt; s2 = fma(dlt, dlt, s2); }
                }
            }
            for (; r < r1; r += rp) {
                const double v = ld_f64(x + r * ld + f);
                if (v == v) { const double dlt = v - sh; cnt += 1.0; s1 += dlt; s2 = fma(dlt, dlt, s2); }
            }
        }
        red[tid] = cnt; red[kThreads + tid] = s1; red[2 * kThreads + tid] = s2;
        __syncthreads();
        if (ry == 0 && f < F) {
            for (int y = 1; y < rp; ++y) {  // fixed order
                cnt += red[y * tf + fx]; s1 += red[kThreads + y * tf + fx]; s2 += red[2 * kThreads + y * tf + fx];
            }
            out[f] = cnt; out[F + f] = s1; out[2 * F + f] = s2;
        }
        __syncthreads();
    }
}

// sums[3][F] = fixed-order sum over blocks; also records the shift used.
// 1024 threads = 16 block-groups x 64 outputs: group g adds blocks g, g+16, ... (coalesced
// over the outputs), then the 16 partial sums are added in group order.
template <typename T>
__global__ __launch_bounds__(1024) void moments_reduce_kernel(const double* __restrict__ partial, int n_blocks, int F,
                                                             const T* __restrict__ x,
                                                             const double* __restrict__ shift_in,
                                                             double* __restrict__ sums, double* __restrict__ shift_out) {
    __shared__ double red[16][64];
    const int io = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + io;
    double acc = 0.0;
    if (i < 3 * F)
        for (int b = g; b < n_blocks; b += 16) acc += partial[(size_t)b * 3 * F + i];
    red[g][io] = acc;
    __syncthreads();
    if (g == 0 && i < 3 * F) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += red[k][io];
        sums[i] = t;
    }
    if (g == 1 && i < F && shift_out) {
        double sh;
        if (shift_in) sh = shift_in[i];
        else { sh = ld_f64(x + i); if (sh != sh) sh = 0.0; }
        shift_out[i] = sh;
    }
}

// mean = shift + S1/cnt ; var = (S2 - S1^2/cnt)/(cnt - ddof) ; std = sqrt(max(var, 0))
__global__ void moments_finalize_kernel(const double* __restrict__ sums, const double* __restrict__ shift, int F,
                                        int ddof, double* __restrict__ mean, double* __restrict__ stdv,
                                        double* __restrict__ count) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const double cnt = sums[f], s1 = sums[F + f], s2 = sums[2 * F + f];
    double m = 0.0, sd = 0.0;
    if (cnt > 0.0) {
        m = shift[f] + s1 / cnt;
        const double denom = cnt - (double)ddof;
        if (denom > 0.0) {
            double var = (s2 - s1 * s1 / cnt) / denom;
            sd = var > 0.0 ? sqrt(var) : 0.0;
        } else {
            sd = __builtin_nan("");
        }
    }
    mean[f] = m;
    stdv[f] = sd;
    if (count) count[f] = cnt;
}

// mean / divisor of reduction._preprocess from raw sums, entirely on the device:
//   mean  = shift + S1/cnt
//   sigma = sqrt((S2 - S1^2/cnt) / n_rows)      (NaNs were imputed with the mean: they add 0
//           to the centred square sum but count in the divisor), sigma < 10 eps -> 1
__global__ void standardise_params_kernel(const double* __restrict__ sums, const double* __restrict__ shift, int F,
                                          double n_rows, int with_std, double* __restrict__ mean,
                                          double* __restrict__ scale, double* __restrict__ inv_scale) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const double cnt = sums[f], s1 = sums[F + f], s2 = sums[2 * F + f];
    double m = 0.0, sd = 1.0;
    if (cnt > 0.0) {
        m = shift[f] + s1 / cnt;
        if (with_std) {
            const double var = (s2 - s1 * s1 / cnt) / n_rows;
            sd = var > 0.0 ? sqrt(var) : 0.0;
            if (sd < 10.0 * 2.220446049250313e-16) sd = 1.0;
        }
    }
    mean[f] = m;
    scale[f] = sd;
    inv_scale[f] = 1.0 / sd;
}

// ---------------------------------------------------------------------------
// project: Y[t][c] = sum_f ((x[t][f] - mu[f]) * inv_sigma[f] - m[f]) * W[f][c]
// NaN inputs are imputed to the column mean (z = 0), as reduction._preprocess does.
// Tile = 64 frames; the standardised tile sits in LDS as fp64 [64][FT+1]
// (odd row stride -> conflict-free column walks), W as [FT][d].
// ---------------------------------------------------------------------------
constexpr int kProjFrames = 64;
constexpr int kProjFT = 64;  // feature chunk

// Running max |y| of everything a workgroup stored, folded into *bits (the IEEE bit pattern of a
// non-negative double orders like the integer): one atomic per workgroup, and only when it can raise
// the maximum.  Feeds the fixed-point scale of the k-means fit without a second pass over Y.
__device__ __forceinline__ void publish_absmax(double m, unsigned long long* __restrict__ bits, double* red) {
    if (!bits) return;
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kThreads / 64; ++w) m = fmax(m, red[w]);
        const unsigned long long b = (unsigned long long)__double_as_longlong(m);
        if (b > *bits) atomicMax(bits, b);
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void project_kernel(
    const T* __restrict__ x, int64_t n, int F, int64_t ld, const double* __restrict__ mu,
    const double* __restrict__ inv_sigma, const double* __restrict__ m2, const double* __restrict__ W, int d,
    int64_t ldw, double* __restrict__ y, int64_t ldy, unsigned long long* __restrict__ absmax_bits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double amax = 0.0;
    double* zs = reinterpret_cast<double*>(smem_raw);          // [64][FT+1]
    double* ws = zs + kProjFrames * (kProjFT + 1);              // [FT][d]
    const int tid = threadIdx.x;
    const int r = tid & 63;
    const int cg = tid >> 6;  // wave id: column group, wave-uniform
    const int64_t n_tiles = (n + kProjFrames - 1) / kProjFrames;
    constexpr int kMaxCols = 16;  // columns per thread (d <= 64)
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t t0 = tile * kProjFrames;
        const int rows = (int)min<int64_t>(kProjFrames, n - t0);
        double acc[kMaxCols];
#pragma unroll
        for (int c = 0; c < kMaxCols; ++c) acc[c] = 0.0;
        for (int f0 = 0; f0 < F; f0 += kProjFT) {
            const int fc = min(kProjFT, F - f0);
            __syncthreads();
            for (int i = tid; i < rows * fc; i += kThreads) {
                const int rr = i / fc, ff = i - rr * fc;
                double v = ld_f64(x + (t0 + rr) * ld + f0 + ff);
                double z = (v - mu[f0 + ff]) * inv_sigma[f0 + ff];
                if (v != v) z = 0.0;
                if (m2) z -= m2[f0 + ff];
                zs[rr * (kProjFT + 1) + ff] = z;
            }
            for (int i = tid; i < fc * d; i += kThreads) {
                const int ff = i / d, c = i - ff * d;
                ws[i] = W[(size_t)(f0 + ff) * ldw + c];
            }
            __syncthreads();
            if (r < rows) {
                const double* zr = zs + r * (kProjFT + 1);
#pragma unroll
                for (int ci = 0; ci < kMaxCols; ++ci) {
                    const int c = cg + 4 * ci;
                    if (c < d) {
                        double a = acc[ci];
                        for (int ff = 0; ff < fc; ++ff) a = fma(zr[ff], ws[ff * d + c], a);
                        acc[ci] = a;
                    }
                }
            }
        }
        if (r < rows) {
#pragma unroll
            for (int ci = 0; ci < kMaxCols; ++ci) {
                const int c = cg + 4 * ci;
                if (c < d) {
                    y[(t0 + r) * ldy + c] = acc[ci];
                    amax = fmax(amax, fabs(acc[ci]));
                }
            }
        }
    }
    publish_absmax(amax, absmax_bits, zs);
}

// ---------------------------------------------------------------------------
// project on the fp64 matrix cores (d <= 16).
//   Y' (16 x 16 frames) = W'' (16 x F) . Z' (F x 16 frames),  W'[f][c] = inv_sigma[f] * W[f][c]
//   Z = x - mu (NaN -> 0), accumulator initialised with -sum_f m2[f] * W[f][c].
// One MFMA = 4 features x (16 output columns x 16 frames).  The contraction order over
// features is free, so k-step s = 4q + i is given features {16q + 4g + i : g = 0..3}: lane
// (frame j = l & 15, g = l >> 4) then needs features 16q + 4g .. 16q + 4g + 3, one 16-byte
// load per q.  Every 256-byte row is consumed whole across the four q loads.
// ---------------------------------------------------------------------------
typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int kPChunk = 64;  // features per pipelined chunk (4 x 16-byte loads per lane)

// W' (zero-padded to 16 columns, row stride 16) and mu live in LDS: the A operand of every MFMA
// is one ds_read_b64, which keeps the kernel at <= 128 VGPRs (4 waves per SIMD) for any F.
// VEC path: the raw 16-byte loads of the NEXT (frame group, chunk) are issued before the MFMAs
// of the current one, so HBM latency overlaps the matrix work inside a wave as well.
// FINITE (vector path): the caller vouches for X without NaNs (msm_project_finite; msm_column_moments' count tells) and the
// NaN test -- three of the five vector instructions per element, all of them matrix-pipe time next to fp64 matrix
// instructions -- is left out.
template <typename T, bool VEC, bool FINITE = false>
__global__ __launch_bounds__(kThreads) void project_mfma_kernel(
    const T* __restrict__ x, int64_t n, int F, int64_t ld, const double* __restrict__ mu,
    const double* __restrict__ inv_sigma, const double* __restrict__ m2, const double* __restrict__ W, int d,
    int64_t ldw, double* __restrict__ y, int64_t ldy, unsigned long long* __restrict__ absmax_bits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char proj_smem[];
    double amax = 0.0;
    // W' as [F16 / 4][16 columns][4 features]: the four A operands a lane needs per 16-byte load of X are two 16-byte
    // LDS reads (the scalar path below indexes the same layout)
    double* wl = reinterpret_cast<double*>(proj_smem);
    const int F16 = (F + 15) & ~15;
    double* mul = wl + (size_t)F16 * 16;  // [F16]
    for (int idx = threadIdx.x; idx < F16 * 16; idx += kThreads) {
        const int f = idx >> 4, c = idx & 15;
        wl[(f >> 2) * 64 + c * 4 + (f & 3)] = (f < F && c < d) ? inv_sigma[f] * W[(size_t)f * ldw + c] : 0.0;
    }
    for (int f = threadIdx.x; f < F16; f += kThreads) mul[f] = f < F ? mu[f] : 0.0;
    __syncthreads();
    // accumulator start of output column c: -(m2 . W'[:, c]), once per workgroup (every lane used to walk the F
    // rows of W itself: 8 F global loads per lane before the first frame)
    double* a0 = mul + F16;   // [16]
    if (threadIdx.x < 16) {
        double a = 0.0;
        if (m2 && threadIdx.x < d)
            for (int f = 0; f < F; ++f) a = fma(m2[f], W[(size_t)f * ldw + threadIdx.x], a);
        a0[threadIdx.x] = -a;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4;
    const int64_t wave_id = ((int64_t)blockIdx.x * kThreads + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * kThreads) >> 6;
    const int64_t n_groups = (n + 15) / 16;
    const int n_chunks = (F16 + kPChunk - 1) / kPChunk;

    // accumulator start: -(m2 . W[:, c]) for this lane's 4 output columns c = g + 4r
    v4f64 acc0;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc0[r] = a0[g + 4 * r];
    auto store = [&](int64_t grp, const v4f64& acc) {
        const int64_t t = grp * 16 + j;
        if (t < n) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = g + 4 * r;
                if (c < d) {
                    y[t * ldy + c] = acc[r];
                    amax = fmax(amax, fabs(acc[r]));
                }
            }
        }
    };

    if constexpr (VEC) {
        using VT = T __attribute__((ext_vector_type(4)));
        // feature offsets past the end (F16 not a multiple of 64) are clamped onto valid memory;
        // their W' rows are read as the zero rows of a clamped LDS index instead
        auto load_raw = [&](int64_t grp, int ch, VT (&raw)[4]) {
            int64_t t = grp * 16 + j;
            t = t < n ? t : n - 1;
            const T* row = x + t * ld;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int f = ch * kPChunk + 16 * q + 4 * g;
                f = f < F16 ? f : F16 - 4;
                raw[q] = *reinterpret_cast<const VT*>(row + f);
            }
        };
        int64_t grp = wave_id;
        int ch = 0;
        VT cur[4];
        if (grp < n_groups) load_raw(grp, 0, cur);
        v4f64 acc = acc0;
        while (grp < n_groups) {
            int64_t ngrp = grp;
            int nch = ch + 1;
            if (nch == n_chunks) { nch = 0; ngrp = grp + n_waves; }
            VT nxt[4];
            if (ngrp < n_groups) load_raw(ngrp, nch, nxt);
            else {
#pragma unroll
                for (int q = 0; q < 4; ++q) nxt[q] = cur[q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int fq = ch * kPChunk + 16 * q + 4 * g;
                const bool live = fq < F16;
                const int fb = live ? fq : 0;
                const double2* mp = reinterpret_cast<const double2*>(mul + fb);
                const double2* wp = reinterpret_cast<const double2*>(wl + (fb >> 2) * 64 + j * 4);
                const double2 m01 = mp[0], m23 = mp[1], w01 = wp[0], w23 = wp[1];
                const double mv[4] = {m01.x, m01.y, m23.x, m23.y};
                const double wv[4] = {w01.x, w01.y, w23.x, w23.y};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double xv = (double)cur[q][i];
                    double z = xv - mv[i];
                    if constexpr (!FINITE) {
                        if (!(xv == xv)) z = 0.0;
                    }
                    const double w = live ? wv[i] : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(w, z, acc, 0, 0, 0);
                }
            }
            if (nch == 0) {
                store(grp, acc);
                acc = acc0;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
            grp = ngrp;
            ch = nch;
        }
    } else {
        for (int64_t grp = wave_id; grp < n_groups; grp += n_waves) {
            const int64_t t = grp * 16 + j;
            const bool tok = t < n;
            const T* row = x + (tok ? t : 0) * ld;
            v4f64 acc = acc0;
            for (int f0 = 0; f0 < F16; f0 += 16) {
                double zv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int f = f0 + 4 * g + i;
                    zv[i] = (tok && f < F) ? (double)row[f] : 0.0;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int f = f0 + 4 * g + i;
                    double z = zv[i] - mul[f];
                    if (!(zv[i] == zv[i]) || !tok || f >= F) z = 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wl[(f >> 2) * 64 + j * 4 + (f & 3)], z, acc, 0, 0, 0);
                }
            }
            store(grp, acc);
        }
    }
    publish_absmax(amax, absmax_bits, wl);
}

int pick_tf(int F) {
    int tf = 1;
    while (tf < F && tf < kThreads) tf <<= 1;
    return tf;
}

template <typename T>
msm_status moments_partial_impl(msm_ctx* ctx, const T* x, int64_t n, int F, int64_t ld, const double* d_shift,
                                double* d_sums, double* d_shift_out) {
    const int tf = pick_tf(F);
    const int rp = kThreads / tf;
    int blocks = (int)std::min<int64_t>((int64_t)ctx->n_cu * 4, std::max<int64_t>(1, n / (rp * 16)));
    const int64_t rows_per_block = (n + blocks - 1) / blocks;
    blocks = (int)((n + rows_per_block - 1) / rows_per_block);
    msm_status rs = msm_reserve_scratch(ctx, (size_t)blocks * 3 * F * sizeof(double));
    if (rs != MSM_OK) return rs;
    double* partial = (double*)ctx->scratch;
    hipLaunchKernelGGL(moments_partial_kernel<T>, dim3(blocks), dim3(kThreads), 3 * kThreads * sizeof(double),
                       ctx->stream, x, n, F, ld, d_shift, tf, rows_per_block, partial);
    MSM_CHECK_LAUNCH(ctx);
    hipLaunchKernelGGL(moments_reduce_kernel<T>, dim3(msm_ceil_div(3 * F, 64)), dim3(1024), 0, ctx->stream, partial,
                       blocks, F, x, d_shift, d_sums, d_shift_out);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

}  // namespace

// column minima / maxima over the FINITE entries, the number of non-finite entries and of rows that are finite
// throughout: what validate_features reports besides mean and std (S/analysis/validation.py:89-172).  One thread per
// row; minima / maxima through the order-preserving 64-bit image of the doubles (LDS per workgroup, then global).
namespace {
__device__ __forceinline__ unsigned long long ordered_bits(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return b ^ ((b >> 63) ? ~0ull : 0x8000000000000000ull);
}
template <typename T>
__global__ __launch_bounds__(256) void column_minmax_kernel(const T* __restrict__ x, int64_t n, int F, int64_t ld,
                                                           unsigned long long* __restrict__ gmin,
                                                           unsigned long long* __restrict__ gmax,
                                                           unsigned long long* __restrict__ counters) {
    extern __shared__ unsigned long long smm[];   // [F] minima, [F] maxima
    unsigned long long* lmin = smm;
    unsigned long long* lmax = smm + F;
    for (int f = threadIdx.x; f < F; f += 256) {
        lmin[f] = ~0ull;
        lmax[f] = 0ull;
    }
    __syncthreads();
    unsigned long long bad_entries = 0, good_rows = 0;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
        const T* row = x + t * ld;
        bool all_ok = true;
        for (int f = 0; f < F; ++f) {
            const double v = (double)row[f];
            if (fabs(v) <= 1.7976931348623157e308) {     // finite
                const unsigned long long key = ordered_bits(v);
                if (key < lmin[f]) atomicMin(&lmin[f], key);
                if (key > lmax[f]) atomicMax(&lmax[f], key);
            } else {
                all_ok = false;
                ++bad_entries;
            }
        }
        good_rows += all_ok ? 1 : 0;
    }
    for (int off = 32; off > 0; off >>= 1) {
        bad_entries += __shfl_down(bad_entries, off, 64);
        good_rows += __shfl_down(good_rows, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (bad_entries) atomicAdd(&counters[0], bad_entries);
        if (good_rows) atomicAdd(&counters[1], good_rows);
    }
    __syncthreads();
    for (int f = threadIdx.x; f < F; f += 256) {
        if (lmin[f] != ~0ull) atomicMin(&gmin[f], lmin[f]);
        if (lmax[f] != 0ull) atomicMax(&gmax[f], lmax[f]);
    }
}
__global__ void column_minmax_finish_kernel(const unsigned long long* __restrict__ gmin, const unsigned long long* __restrict__ gmax,
                                            int F, double* __restrict__ out_min, double* __restrict__ out_max) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    auto back = [](unsigned long long k) {
        const unsigned long long b = k ^ ((k >> 63) ? 0x8000000000000000ull : ~0ull);
        return __longlong_as_double((long long)b);
    };
    const double nan = __longlong_as_double(0x7FF8000000000000ll);
    out_min[f] = gmin[f] == ~0ull ? nan : back(gmin[f]);
    out_max[f] = gmax[f] == 0ull ? nan : back(gmax[f]);
}
}  // namespace

extern "C" {

msm_status msm_column_moments_partial(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int F,
                                      int64_t ld, const double* d_shift, double* d_sums, double* d_shift_out) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 1 && F >= 1 && ld >= F, "msm_column_moments_partial: need n >= 1, F >= 1, ld >= F");
    MSM_REQUIRE(ctx, d_x && d_sums, "msm_column_moments_partial: NULL pointer");
    MSM_REQUIRE(ctx, dtype == MSM_F32 || dtype == MSM_F64, "msm_column_moments_partial: bad dtype");
    if (dtype == MSM_F32)
        return moments_partial_impl<float>(ctx, (const float*)d_x, n, F, ld, d_shift, d_sums, d_shift_out);
    return moments_partial_impl<double>(ctx, (const double*)d_x, n, F, ld, d_shift, d_sums, d_shift_out);
}

msm_status msm_moments_finalize(msm_ctx* ctx, const double* d_sums, const double* d_shift, int F, int ddof,
                                double* d_mean, double* d_std, double* d_count) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, F >= 1 && ddof >= 0, "msm_moments_finalize: need F >= 1, ddof >= 0");
    MSM_REQUIRE(ctx, d_sums && d_shift && d_mean && d_std, "msm_moments_finalize: NULL pointer");
    hipLaunchKernelGGL(moments_finalize_kernel, dim3(msm_ceil_div(F, 256)), dim3(256), 0, ctx->stream, d_sums, d_shift,
                       F, ddof, d_mean, d_std, d_count);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_standardise_params(msm_ctx* ctx, const double* d_sums, const double* d_shift, int F, double n_rows,
                                  int with_std, double* d_mean, double* d_scale, double* d_inv_scale) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, F >= 1 && n_rows >= 1.0, "msm_standardise_params: need F >= 1, n_rows >= 1");
    MSM_REQUIRE(ctx, d_sums && d_shift && d_mean && d_scale && d_inv_scale, "msm_standardise_params: NULL pointer");
    hipLaunchKernelGGL(standardise_params_kernel, dim3(msm_ceil_div(F, 256)), dim3(256), 0, ctx->stream, d_sums, d_shift,
                       F, n_rows, with_std, d_mean, d_scale, d_inv_scale);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_column_minmax(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int F, int64_t ld, double* d_min,
                             double* d_max, int64_t* d_counts) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 0 && F >= 1 && ld >= F, "msm_column_minmax: bad shape");
    MSM_REQUIRE(ctx, dtype == MSM_F32 || dtype == MSM_F64, "msm_column_minmax: bad dtype");
    MSM_REQUIRE(ctx, d_min && d_max && d_counts && (d_x || n == 0), "msm_column_minmax: NULL pointer");
    msm_status rs = msm_reserve_scratch(ctx, (size_t)2 * F * sizeof(unsigned long long));
    if (rs != MSM_OK) return rs;
    unsigned long long* gmin = (unsigned long long*)ctx->scratch;
    unsigned long long* gmax = gmin + F;
    MSM_HIP(ctx, hipMemsetAsync(gmin, 0xFF, (size_t)F * sizeof(unsigned long long), ctx->stream));
    MSM_HIP(ctx, hipMemsetAsync(gmax, 0, (size_t)F * sizeof(unsigned long long), ctx->stream));
    MSM_HIP(ctx, hipMemsetAsync(d_counts, 0, 2 * sizeof(int64_t), ctx->stream));
    if (n > 0) {
        const int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)ctx->n_cu * 8);
        const size_t lds = (size_t)2 * F * sizeof(unsigned long long);
        if (dtype == MSM_F32)
            hipLaunchKernelGGL(column_minmax_kernel<float>, dim3(grid), dim3(256), lds, ctx->stream, (const float*)d_x, n, F, ld,
                               gmin, gmax, (unsigned long long*)d_counts);
        else
            hipLaunchKernelGGL(column_minmax_kernel<double>, dim3(grid), dim3(256), lds, ctx->stream, (const double*)d_x, n, F, ld,
                               gmin, gmax, (unsigned long long*)d_counts);
        MSM_CHECK_LAUNCH(ctx);
    }
    hipLaunchKernelGGL(column_minmax_finish_kernel, dim3((F + 63) / 64), dim3(64), 0, ctx->stream, gmin, gmax, F, d_min, d_max);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_column_moments(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int F, int64_t ld, int ddof,
                              double* d_mean, double* d_std, double* d_count) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, F >= 1, "msm_column_moments: F must be >= 1");
    // workspace after the per-block partials: [3F sums][F shift]
    const size_t tail = (size_t)4 * F * sizeof(double);
    const size_t head = (size_t)ctx->n_cu * 8 * 3 * F * sizeof(double);
    msm_status rs = msm_reserve_scratch(ctx, head + tail);
    if (rs != MSM_OK) return rs;
    double* sums = (double*)((char*)ctx->scratch + head);
    double* shift = sums + 3 * F;
    rs = msm_column_moments_partial(ctx, d_x, dtype, n, F, ld, nullptr, sums, shift);
    if (rs != MSM_OK) return rs;
    return msm_moments_finalize(ctx, sums, shift, F, ddof, d_mean, d_std, d_count);
}

static msm_status project_impl(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int F, int64_t ld,
                               const double* d_mu, const double* d_inv_sigma, const double* d_mean2, const double* d_w, int d,
                               int64_t ldw, double* d_y, int64_t ldy, double* d_absmax, bool finite) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 0 && F >= 1 && d >= 1 && d <= 64, "msm_project: need n >= 0, F >= 1, 1 <= d <= 64");
    unsigned long long* bits = reinterpret_cast<unsigned long long*>(d_absmax);
    if (bits) MSM_HIP(ctx, hipMemsetAsync(bits, 0, sizeof(unsigned long long), ctx->stream));
    MSM_REQUIRE(ctx, ld >= F && ldw >= d && ldy >= d, "msm_project: bad leading dimension");
    MSM_REQUIRE(ctx, dtype == MSM_F32 || dtype == MSM_F64, "msm_project: bad dtype");
    if (n == 0) return MSM_OK;
    MSM_REQUIRE(ctx, d_x && d_mu && d_inv_sigma && d_w && d_y, "msm_project: NULL pointer");
    if (d <= 16 && ((size_t)((F + 15) & ~15) * 17 + 16) * sizeof(double) <= 48 * 1024) {
        const int64_t n_groups = (n + 15) / 16;
        const size_t esz = dtype == MSM_F32 ? 4 : 8;
        const int F16 = (F + 15) & ~15;
        const size_t plds = ((size_t)F16 * 17 + 16) * sizeof(double);
        const bool vec = (F % 16 == 0) && (ld % 4 == 0) && (((uintptr_t)d_x) % (4 * esz) == 0);
        // one wave of workgroups: exactly as many as are resident at once (a grid of 8 per CU ran a second, thin round
        // behind the 6 per CU the registers allow)
#define MSM_PROJ(T, V, FIN)                                                                                    \
        do {                                                                                                   \
            int per_cu = 0;                                                                                    \
            MSM_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, project_mfma_kernel<T, V, FIN>, kThreads, plds)); \
            const int grid = (int)std::min<int64_t>((n_groups + 3) / 4, (int64_t)ctx->n_cu * std::max(per_cu, 1)); \
            hipLaunchKernelGGL((project_mfma_kernel<T, V, FIN>), dim3(grid), dim3(kThreads), plds, ctx->stream, (const T*)d_x, n, \
                               F, ld, d_mu, d_inv_sigma, d_mean2, d_w, d, ldw, d_y, ldy, bits);                \
        } while (0)
        if (dtype == MSM_F32) { if (vec && finite) MSM_PROJ(float, true, true); else if (vec) MSM_PROJ(float, true, false); else MSM_PROJ(float, false, false); }
        else { if (vec && finite) MSM_PROJ(double, true, true); else if (vec) MSM_PROJ(double, true, false); else MSM_PROJ(double, false, false); }
#undef MSM_PROJ
        MSM_CHECK_LAUNCH(ctx);
        return MSM_OK;
    }
    const size_t lds = ((size_t)kProjFrames * (kProjFT + 1) + (size_t)kProjFT * d) * sizeof(double);
    const int64_t n_tiles = (n + kProjFrames - 1) / kProjFrames;
    const int grid = (int)std::min<int64_t>(n_tiles, (int64_t)ctx->n_cu * 8);
    if (lds > 48 * 1024) {
        MSM_HIP(ctx, hipFuncSetAttribute((const void*)project_kernel<float>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MSM_HIP(ctx, hipFuncSetAttribute((const void*)project_kernel<double>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (dtype == MSM_F32)
        hipLaunchKernelGGL(project_kernel<float>, dim3(grid), dim3(kThreads), lds, ctx->stream, (const float*)d_x, n, F,
                           ld, d_mu, d_inv_sigma, d_mean2, d_w, d, ldw, d_y, ldy, bits);
    else
        hipLaunchKernelGGL(project_kernel<double>, dim3(grid), dim3(kThreads), lds, ctx->stream, (const double*)d_x, n,
                           F, ld, d_mu, d_inv_sigma, d_mean2, d_w, d, ldw, d_y, ldy, bits);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_project(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int F, int64_t ld,
                       const double* d_mu, const double* d_inv_sigma, const double* d_mean2, const double* d_w, int d,
                       int64_t ldw, double* d_y, int64_t ldy, double* d_absmax) {
    return project_impl(ctx, d_x, dtype, n, F, ld, d_mu, d_inv_sigma, d_mean2, d_w, d, ldw, d_y, ldy, d_absmax, false);
}

msm_status msm_project_finite(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int F, int64_t ld,
                              const double* d_mu, const double* d_inv_sigma, const double* d_mean2, const double* d_w,
                              int d, int64_t ldw, double* d_y, int64_t ldy, double* d_absmax) {
    return project_impl(ctx, d_x, dtype, n, F, ld, d_mu, d_inv_sigma, d_mean2, d_w, d, ldw, d_y, ldy, d_absmax, true);
}

}  // extern "C"
