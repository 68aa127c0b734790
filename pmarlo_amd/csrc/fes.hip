// Free-energy surfaces from frame-weighted collective variables  (SURVEY.md section 8f rank 3).
//
// Reference: compute_weighted_fes and helpers (S/analysis/fes.py:91-114 weights, :142-173
// bandwidth, :176-238 Gaussian KDE on the bin centres, :241-292 weighted 2-D histogram with
// sparse-bin smoothing, :570-599 F = -kT ln p - min).
//
//  * weighted_stats:  per-coordinate weighted mean / variance (two passes, like np.average twice),
//    min / max, sum w, sum w^2 -- fixed-order tree reductions.
//  * hist2d:  np.histogram2d semantics (bin = searchsorted(edges, v, 'right') - 1, the last edge
//    belongs to the last bin, values outside are dropped) against the SAME edge values the host
//    hands in; LDS bins, integer atomics: counts are exact, weights go through 2^e fixed point so
//    the sums do not depend on scheduling.
//  * kde2d:  density[i][j] = sum_k ex_i(k) ey_j(k) w_k is a GEMM whose contraction index is the
//    frame: v_mfma_f64_16x16x4_f64 with the Gaussian factors evaluated on the fly (8 exps feed 16
//    matrix instructions per 4 frames and lane); per-workgroup slabs, fixed-order reduction.
#include "common.h"

namespace {

constexpr int kT = 256;
typedef double v4f64 __attribute__((ext_vector_type(4)));

__device__ double block_sum(double v, double* red) {  // valid in thread 0; blockDim.x <= 1024
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}

// ---------------------------------------------------------------------------------------------
// weighted statistics of one strided coordinate.  partial[b] = {sw, sw2, swx, min, max} then a
// second launch for the variance around the global mean.  w == NULL -> unit weights.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void wstats_pass1(const double* __restrict__ x, int64_t stride, int64_t n,
                                                     const double* __restrict__ w, double* __restrict__ partial) {
    __shared__ double red[16];
    double sw = 0.0, sw2 = 0.0, swx = 0.0, mn = __builtin_inf(), mx = -__builtin_inf();
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * per, i1 = min(n, i0 + per);
    for (int64_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const double v = x[i * stride];
        const double wi = w ? w[i] : 1.0;
        sw += wi;
        sw2 = fma(wi, wi, sw2);
        swx = fma(wi, v, swx);
        mn = fmin(mn, v);
        mx = fmax(mx, v);
    }
    const double a = block_sum(sw, red), b = block_sum(sw2, red), c = block_sum(swx, red);
    for (int off = 32; off > 0; off >>= 1) {
        mn = fmin(mn, __shfl_down(mn, off, 64));
        mx = fmax(mx, __shfl_down(mx, off, 64));
    }
    __shared__ double rmn[16], rmx[16];
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { rmn[threadIdx.x >> 6] = mn; rmx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) { mn = fmin(mn, rmn[i]); mx = fmax(mx, rmx[i]); }
        double* p = partial + 5 * blockIdx.x;
        p[0] = a; p[1] = b; p[2] = c; p[3] = mn; p[4] = mx;
    }
}

__global__ void wstats_mid(const double* __restrict__ partial, int nb, double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double sw = 0.0, sw2 = 0.0, swx = 0.0, mn = __builtin_inf(), mx = -__builtin_inf();
    for (int b = 0; b < nb; ++b) {
        const double* p = partial + 5 * b;
        sw += p[0]; sw2 += p[1]; swx += p[2];
        mn = fmin(mn, p[3]); mx = fmax(mx, p[4]);
    }
    out[0] = sw; out[1] = sw2; out[2] = swx / sw; out[4] = mn; out[5] = mx;
}

__global__ __launch_bounds__(1024) void wstats_pass2(const double* __restrict__ x, int64_t stride, int64_t n,
                                                     const double* __restrict__ w, const double* __restrict__ out,
                                                     double* __restrict__ partial) {
    __shared__ double red[16];
    const double mean = out[2];
    double acc = 0.0;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * per, i1 = min(n, i0 + per);
    for (int64_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const double d = x[i * stride] - mean;
        acc = fma(w ? w[i] : 1.0, d * d, acc);
    }
    const double t = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ void wstats_final(const double* __restrict__ partial, int nb, double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int b = 0; b < nb; ++b) s += partial[b];
    out[3] = s / out[0];
}

// ---------------------------------------------------------------------------------------------
// 2-D histogram
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int edge_bin(const double* e, int nb, double inv_d, double v) {
    // searchsorted(e, v, 'right') - 1, last edge inclusive; -1 when outside / NaN
    if (!(v >= e[0]) || !(v <= e[nb])) return -1;
    int i = (int)((v - e[0]) * inv_d);
    i = i < 0 ? 0 : (i > nb - 1 ? nb - 1 : i);
    while (i > 0 && v < e[i]) --i;
    while (i < nb - 1 && v >= e[i + 1]) ++i;
    return i;
}

template <bool WEIGHTED, bool LDS_BINS>
__global__ __launch_bounds__(kT) void hist2d_kernel(const double* __restrict__ x, int64_t sx, const double* __restrict__ y,
                                                    int64_t sy, int64_t n, const double* __restrict__ w, double scale,
                                                    const double* __restrict__ xe, int nx, const double* __restrict__ ye,
                                                    int ny, unsigned long long* __restrict__ bins) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* lxe = reinterpret_cast<double*>(smem);
    double* lye = lxe + nx + 1;
    unsigned long long* lb = reinterpret_cast<unsigned long long*>(lye + ny + 1);
    for (int i = threadIdx.x; i <= nx; i += kT) lxe[i] = xe[i];
    for (int i = threadIdx.x; i <= ny; i += kT) lye[i] = ye[i];
    if (LDS_BINS)
        for (int i = threadIdx.x; i < nx * ny; i += kT) lb[i] = 0ull;
    __syncthreads();
    const double inv_dx = nx / (lxe[nx] - lxe[0]), inv_dy = ny / (lye[ny] - lye[0]);
    for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kT) {
        const int ix = edge_bin(lxe, nx, inv_dx, x[i * sx]);
        const int iy = edge_bin(lye, ny, inv_dy, y[i * sy]);
        if (ix < 0 || iy < 0) continue;
        const unsigned long long inc = WEIGHTED ? (unsigned long long)__double2ll_rn(w[i] * scale) : 1ull;
        atomicAdd((LDS_BINS ? lb : bins) + ix * ny + iy, inc);
    }
    if (LDS_BINS) {
        __syncthreads();
        for (int i = threadIdx.x; i < nx * ny; i += kT)
            if (lb[i]) atomicAdd(&bins[i], lb[i]);
    }
}

__global__ void hist_to_f64(const unsigned long long* __restrict__ bins, int n, double inv_scale, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (double)(long long)bins[i] * inv_scale;
}

// 3 x 3 neighbour mean (centre excluded, edges replicated) fills bins below min_count
// (S/analysis/fes.py:270-292); out != in
__global__ void smooth_kernel(const double* __restrict__ h, int nx, int ny, double min_count, double* __restrict__ out,
                              unsigned int* __restrict__ n_smoothed) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nx * ny) return;
    const int i = e / ny, j = e - i * ny;
    const double c = h[e];
    double v = c;
    if (c < min_count) {
        double tot = 0.0;
        for (int di = -1; di <= 1; ++di)
            for (int dj = -1; dj <= 1; ++dj) {
                const int ii = min(max(i + di, 0), nx - 1), jj = min(max(j + dj, 0), ny - 1);
                tot += h[ii * ny + jj];
            }
        const double nm = (tot - c) / 8.0;
        const double target = fmax(nm, min_count);
        if (nm > 0.0 && target > c) { v = target; atomicAdd(n_smoothed, 1u); }
    }
    out[e] = v;
}

__global__ __launch_bounds__(1024) void scale_to_total(double* __restrict__ h, int n, double want_total) {
    __shared__ double red[16];
    __shared__ double f;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += h[i];
    const double t = block_sum(acc, red);
    if (threadIdx.x == 0) f = t > 0.0 ? want_total / t : 1.0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) h[i] *= f;
}

// F = -kT ln(h / total) - min;  status bit 0: non-finite entry, bit 1: total <= 0, bit 2: entry <= 0
__global__ __launch_bounds__(1024) void fes_finalize_kernel(const double* __restrict__ h, int n, double kT_,
                                                            double* __restrict__ F, int* __restrict__ status) {
    __shared__ double red[16];
    __shared__ double total, fmin_s;
    __shared__ int st;
    if (threadIdx.x == 0) st = 0;
    __syncthreads();
    double acc = 0.0;
    int bad = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double v = h[i];
        if (!(v == v) || v > 1.0e300 || v < -1.0e300) bad |= 1;
        if (v <= 0.0) bad |= 4;
        acc += v;
    }
    const double t = block_sum(acc, red);
    if (bad) atomicOr(&st, bad);
    if (threadIdx.x == 0) total = t;
    __syncthreads();
    if (threadIdx.x == 0 && !(total > 0.0 && total < 1.0e300)) st |= 2;
    __syncthreads();
    if (st) {
        if (threadIdx.x == 0) *status = st;
        return;
    }
    double mn = __builtin_inf();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double f = -kT_ * log(h[i] / total);
        F[i] = f;
        mn = fmin(mn, f);
    }
    for (int off = 32; off > 0; off >>= 1) mn = fmin(mn, __shfl_down(mn, off, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mn;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) mn = fmin(mn, red[i]);
        fmin_s = mn;
        *status = 0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) F[i] -= fmin_s;
}

// ---------------------------------------------------------------------------------------------
// Gaussian KDE on a 64 x 64 block of bin centres (rows i0.., columns j0..), all frames.
// MFMA operands per 4 frames: A lane l = ex_{16 rb + (l & 15)}(frame k0 + (l >> 4)), B likewise
// ey * w; D reg r of tile (rb, cb) = density[16 rb + (l >> 4) + 4 r][16 cb + (l & 15)].
// ---------------------------------------------------------------------------------------------
// ((a + pi) mod 2 pi) - pi with numpy's sign convention for the remainder (result in [-pi, pi))
__device__ __forceinline__ double wrap_angle(double a) {
    const double two_pi = 6.283185307179586, pi = 3.141592653589793;
    double m = fmod(a + pi, two_pi);
    if (m < 0.0) m += two_pi;
    return m - pi;
}

__global__ __launch_bounds__(256, 2) void kde2d_kernel(const double* __restrict__ x, int64_t sx, const double* __restrict__ y,
                                                       int64_t sy, int64_t n, const double* __restrict__ w, double w_scale,
                                                       const double* __restrict__ xcen, int nx,
                                                       const double* __restrict__ ycen, int ny, double inv_bwx,
                                                       double inv_bwy, int periodic, double* __restrict__ slabs) {
    __shared__ double acc_lds[64 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.z * 64;
    double xc[4], yc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {  // centres past the grid are clamped: computed, never stored
        const int i = min(i0 + 16 * b + (lane & 15), nx - 1), j = min(j0 + 16 * b + (lane & 15), ny - 1);
        xc[b] = xcen[i];
        yc[b] = ycen[j];
    }
    v4f64 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (v4f64){0.0, 0.0, 0.0, 0.0};
    const int64_t n_groups = (n + 3) / 4;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + wave, n_waves = (int64_t)gridDim.x * 4;
    for (int64_t grp = wave_id; grp < n_groups; grp += n_waves) {
        const int64_t k = grp * 4 + (lane >> 4);
        const bool ok = k < n;
        const int64_t kc = ok ? k : n - 1;
        const double xv = x[kc * sx], yv = y[kc * sy];
        const double wv = ok ? (w ? w[kc] * w_scale : w_scale) : 0.0;
        double ea[4], eb[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            double dx = xc[b] - xv, dy = yc[b] - yv;
            if (periodic & 1) dx = wrap_angle(dx);     // toroidal coordinate: nearest image
            if (periodic & 2) dy = wrap_angle(dy);
            const double u = dx * inv_bwx, v = dy * inv_bwy;
            ea[b] = exp(-0.5 * (u * u));
            eb[b] = exp(-0.5 * (v * v)) * wv;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ea[a], eb[b], acc[a][b], 0, 0, 0);
    }
    // the four waves add their blocks one after the other (fixed order), then one slab per workgroup
    for (int turn = 0; turn < 4; ++turn) {
        if (wave == turn) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * a + (lane >> 4) + 4 * r, col = 16 * b + (lane & 15);
                        if (turn == 0) acc_lds[row * 64 + col] = acc[a][b][r];
                        else acc_lds[row * 64 + col] += acc[a][b][r];
                    }
        }
        __syncthreads();
    }
    double* slab = slabs + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4096;
    for (int e = threadIdx.x; e < 4096; e += 256) slab[e] = acc_lds[e];
}

__global__ void kde_reduce_kernel(const double* __restrict__ slabs, int n_slabs, int nby, int nbz, int nx, int ny,
                                  double normaliser, double* __restrict__ density) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nx * ny) return;
    const int i = e / ny, j = e - i * ny;
    const int by = i / 64, bz = j / 64;
    const double* s = slabs + (size_t)(bz * nby + by) * n_slabs * 4096 + (i % 64) * 64 + (j % 64);
    double t = 0.0;
    for (int b = 0; b < n_slabs; ++b) t += s[(size_t)b * 4096];
    density[e] = t * normaliser;
    (void)nbz;
}

}  // namespace

namespace {
// out[t] = table[idx[t]] (0 for an index outside [0, m)): per-frame weights pi[state] of the MSM-reweighted FES
__global__ __launch_bounds__(256) void gather_kernel(const double* __restrict__ table, int m, const int32_t* __restrict__ idx,
                                                     int64_t n, double* __restrict__ out) {
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
        const int i = idx[t];
        out[t] = (unsigned)i < (unsigned)m ? table[i] : 0.0;
    }
}

// mode 1: np.clip(x, lo, hi);  mode 2: ((x - lo) % (hi - lo)) + lo with numpy's remainder (result in [lo, hi))
__global__ __launch_bounds__(256) void clip_or_wrap_kernel(const double* __restrict__ x, int64_t stride, int64_t n,
                                                           double lo, double hi, int mode, double* __restrict__ out) {
    const double span = hi - lo;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
        const double v = x[t * stride];
        double r;
        if (mode == 1) {
            r = fmin(fmax(v, lo), hi);
        } else {
            r = fmod(v - lo, span);
            if (r != 0.0) { if (r < 0.0) r += span; } else r = 0.0;
            r += lo;
        }
        out[t] = r;
    }
}
}  // namespace

extern "C" {

msm_status msm_weighted_stats(msm_ctx* ctx, const double* d_x, int64_t stride, int64_t n, const double* d_w,
                              double* d_out6) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 1 && stride >= 1 && d_x && d_out6, "msm_weighted_stats: bad arguments");
    const int nb = (int)std::min<int64_t>((n + 4095) / 4096, (int64_t)ctx->n_cu * 2);
    msm_status rs = msm_reserve_scratch(ctx, (size_t)nb * 6 * sizeof(double));
    if (rs != MSM_OK) return rs;
    double* part = (double*)ctx->scratch;
    hipLaunchKernelGGL(wstats_pass1, dim3(nb), dim3(1024), 0, ctx->stream, d_x, stride, n, d_w, part);
    hipLaunchKernelGGL(wstats_mid, dim3(1), dim3(64), 0, ctx->stream, part, nb, d_out6);
    hipLaunchKernelGGL(wstats_pass2, dim3(nb), dim3(1024), 0, ctx->stream, d_x, stride, n, d_w, d_out6, part + 5 * nb);
    hipLaunchKernelGGL(wstats_final, dim3(1), dim3(64), 0, ctx->stream, part + 5 * nb, nb, d_out6);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_hist2d(msm_ctx* ctx, const double* d_x, int64_t sx, const double* d_y, int64_t sy, int64_t n,
                      const double* d_w, double w_absmax, const double* d_xedges, int nx, const double* d_yedges, int ny,
                      double* d_hist) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 0 && nx >= 1 && ny >= 1 && (int64_t)nx * ny <= (1 << 24), "msm_hist2d: bad shape");
    MSM_REQUIRE(ctx, d_xedges && d_yedges && d_hist && (n == 0 || (d_x && d_y)), "msm_hist2d: NULL pointer");
    MSM_REQUIRE(ctx, !d_w || (w_absmax > 0.0 && w_absmax < 1e300), "msm_hist2d: w_absmax must bound |w| (> 0)");
    const int cells = nx * ny;
    msm_status rs = msm_reserve_scratch(ctx, (size_t)cells * sizeof(unsigned long long));
    if (rs != MSM_OK) return rs;
    unsigned long long* bins = (unsigned long long*)ctx->scratch;
    MSM_HIP(ctx, hipMemsetAsync(bins, 0, (size_t)cells * sizeof(unsigned long long), ctx->stream));
    double scale = 1.0;
    if (d_w) {  // 2^e with n * w_absmax * 2^e < 2^62
        int e = 61 - (int)std::ceil(std::log2(std::max(1.0, (double)n) * w_absmax));
        e = std::min(e, 1000);
        scale = std::ldexp(1.0, e);
    }
    if (n > 0) {
        const size_t edge_bytes = (size_t)(nx + ny + 2) * sizeof(double);
        const bool lds_bins = (size_t)cells * 8 + edge_bytes <= 64 * 1024;
        const size_t lds = edge_bytes + (lds_bins ? (size_t)cells * 8 : 0);
        const int grid = (int)std::min<int64_t>((n + kT * 8 - 1) / (kT * 8), (int64_t)ctx->n_cu * 4);
#define MSM_H2D(W, L)                                                                                              \
        hipLaunchKernelGGL((hist2d_kernel<W, L>), dim3(grid), dim3(kT), lds, ctx->stream, d_x, sx, d_y, sy, n, d_w, \
                           scale, d_xedges, nx, d_yedges, ny, bins)
        if (d_w) { if (lds_bins) MSM_H2D(true, true); else MSM_H2D(true, false); }
        else { if (lds_bins) MSM_H2D(false, true); else MSM_H2D(false, false); }
#undef MSM_H2D
        MSM_CHECK_LAUNCH(ctx);
    }
    hipLaunchKernelGGL(hist_to_f64, dim3((cells + 255) / 256), dim3(256), 0, ctx->stream, bins, cells, 1.0 / scale, d_hist);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_smooth_sparse_bins(msm_ctx* ctx, const double* d_hist, int nx, int ny, double min_count, double* d_out,
                                  int32_t* d_n_smoothed) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, nx >= 1 && ny >= 1 && d_hist && d_out && d_n_smoothed && d_out != d_hist,
                "msm_smooth_sparse_bins: bad arguments");
    MSM_HIP(ctx, hipMemsetAsync(d_n_smoothed, 0, sizeof(int32_t), ctx->stream));
    const int cells = nx * ny;
    hipLaunchKernelGGL(smooth_kernel, dim3((cells + 255) / 256), dim3(256), 0, ctx->stream, d_hist, nx, ny, min_count,
                       d_out, (unsigned int*)d_n_smoothed);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_scale_to_total(msm_ctx* ctx, double* d_v, int n, double total) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 1 && d_v && total > 0.0, "msm_scale_to_total: bad arguments");
    hipLaunchKernelGGL(scale_to_total, dim3(1), dim3(1024), 0, ctx->stream, d_v, n, total);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_fes_finalize(msm_ctx* ctx, const double* d_hist, int n_cells, double kT_, double* d_F, int32_t* d_status) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n_cells >= 1 && d_hist && d_F && d_status && kT_ > 0.0, "msm_fes_finalize: bad arguments");
    hipLaunchKernelGGL(fes_finalize_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_hist, n_cells, kT_, d_F, d_status);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_kde2d(msm_ctx* ctx, const double* d_x, int64_t sx, const double* d_y, int64_t sy, int64_t n,
                     const double* d_w, double w_scale, const double* d_xcenters, int nx, const double* d_ycenters, int ny,
                     double bw_x, double bw_y, int periodic, double* d_density) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 1 && nx >= 1 && ny >= 1 && nx <= 4096 && ny <= 4096, "msm_kde2d: bad shape");
    MSM_REQUIRE(ctx, periodic >= 0 && periodic <= 3, "msm_kde2d: periodic is a 2-bit mask (1 = x, 2 = y)");
    MSM_REQUIRE(ctx, d_x && d_y && d_density && d_xcenters && d_ycenters && bw_x > 0.0 && bw_y > 0.0,
                "msm_kde2d: bad arguments");
    const int nby = (nx + 63) / 64, nbz = (ny + 63) / 64;
    const int64_t n_groups = (n + 3) / 4;
    int gx = (int)std::min<int64_t>((n_groups + 3) / 4, std::max(1, ctx->n_cu * 2 / (nby * nbz)));
    gx = std::max(gx, 1);
    const size_t slab_bytes = (size_t)gx * nby * nbz * 4096 * sizeof(double);
    msm_status rs = msm_reserve_scratch(ctx, slab_bytes);
    if (rs != MSM_OK) return rs;
    double* slabs = (double*)ctx->scratch;
    hipLaunchKernelGGL(kde2d_kernel, dim3(gx, nby, nbz), dim3(256), 0, ctx->stream, d_x, sx, d_y, sy, n, d_w, w_scale,
                       d_xcenters, nx, d_ycenters, ny, 1.0 / bw_x, 1.0 / bw_y, periodic, slabs);
    MSM_CHECK_LAUNCH(ctx);
    const double normaliser = 1.0 / (2.0 * 3.14159265358979323846 * bw_x * bw_y);
    hipLaunchKernelGGL(kde_reduce_kernel, dim3((nx * ny + 255) / 256), dim3(256), 0, ctx->stream, slabs, gx, nby, nbz, nx,
                       ny, normaliser, d_density);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_clip_or_wrap(msm_ctx* ctx, const double* d_x, int64_t stride, int64_t n, double lo, double hi, int mode,
                            double* d_out) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, d_x && d_out && n >= 0 && stride >= 1, "msm_clip_or_wrap: bad arguments");
    MSM_REQUIRE(ctx, (mode == 1 || mode == 2) && lo < hi, "msm_clip_or_wrap: mode 1 = clip, 2 = wrap; need lo < hi");
    if (n == 0) return MSM_OK;
    const int blocks = (int)std::min<int64_t>(std::max<int64_t>(1, (n + 1023) / 1024), (int64_t)ctx->n_cu * 8);
    hipLaunchKernelGGL(clip_or_wrap_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_x, stride, n, lo, hi, mode, d_out);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_gather_f64(msm_ctx* ctx, const double* d_table, int m, const int32_t* d_idx, int64_t n, double* d_out) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, d_table && d_idx && d_out && m >= 1 && n >= 0, "msm_gather_f64: bad arguments");
    if (n == 0) return MSM_OK;
    const int blocks = (int)std::min<int64_t>(std::max<int64_t>(1, (n + 1023) / 1024), (int64_t)ctx->n_cu * 8);
    hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_table, m, d_idx, n, d_out);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

}  // extern "C"
