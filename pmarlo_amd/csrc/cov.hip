// Time-lagged covariance moments on the fp64 matrix cores (v_mfma_f64_16x16x4_f64).
//
// This is the one dense contraction of the path: for centred frames z_t
//     M00 = sum_t w_t z_t z_t'          w_t = [t in X0] + [t in Yt]   (symmetric)
//     M0t = sum_{t in X0} z_t z_{t+lag}'
//     sx  = sum_{t in X0} z_t,  sy = sum_{t in Yt} z_t,  T = #pairs
// where X0 = frames with a partner lag frames later in the same segment and
// Yt = the partners.  These are the raw moments of deeptime's reversible
// Covariance estimator (what TICA.fit consumes; oracle/npport.py:lagged_moments),
// kept un-normalised so that shards all-reduce by plain summation.
//
// Mapping.  The contraction index is the frame axis: one MFMA consumes 4 frames
// (K = 4) and a 16x16 feature tile.  Lane l holds feature (l & 15) of frame
// (l >> 4) for every 16-feature tile, which is both the A and the B fragment
// layout, so z_t serves as A for every product and as B for M00; the lagged
// frame is the B operand of M0t.  A wave owns a contiguous run of frames and
// ALL tiles (F <= 64: 16 M0t + 10 upper M00 tiles = 208 accumulator VGPRs), so
// X is read from HBM exactly once (the lagged re-read hits L2).  Per-wave
// accumulators are summed across the 4 waves of a workgroup through LDS in a
// fixed order, written as one slab per workgroup, and a second kernel adds the
// slabs in block order: bitwise reproducible, no atomics.
//
// Roofline: 3*F^2 flop per frame (2F^2 for M0t + F^2 for the symmetric half of
// M00) against F*s bytes -> MFMA-bound for F >~ 27.
#include "common.h"

#include <cstring>

namespace {

typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kUnroll = 2;  // k-groups (of 4 frames) per pipelined iteration

struct FrameTab {
    int n;    // segments longer than lag
    int lag;
    int64_t start[MSM_SEG_INLINE];
    int64_t stop[MSM_SEG_INLINE];
    int64_t prefix[MSM_SEG_INLINE + 1];  // dense frame index prefix
    int64_t total;                       // frames in those segments
    int64_t pairs;                       // T
};

struct FrameInfo {
    int64_t t;
    bool valid, in_x0, in_yt;
};

__device__ __forceinline__ FrameInfo locate_frame(const FrameTab& ft, int64_t q, int64_t q_end) {
    FrameInfo fi;
    fi.valid = q < q_end;
    const int64_t qq = fi.valid ? q : 0;
    int s = 0;
#pragma unroll 1
    while (s + 1 < ft.n && qq >= ft.prefix[s + 1]) ++s;
    fi.t = ft.start[s] + (qq - ft.prefix[s]);
    fi.in_x0 = fi.valid && (fi.t + ft.lag < ft.stop[s]);
    fi.in_yt = fi.valid && (fi.t - ft.lag >= ft.start[s]);
    return fi;
}

template <typename T>
__device__ __forceinline__ double to_f64(T v) { return (double)v; }

template <int NT>
struct CovShape {
    static constexpr int kSym = NT * (NT + 1) / 2;
    static constexpr int kTiles = NT * NT + kSym;
    static constexpr int kSlab = kTiles * 256 + 2 * NT * 16;  // doubles per workgroup slab
};

// upper-triangular tile index -> position
template <int NT>
__host__ __device__ constexpr int sym_index(int ti, int tj) { return ti * NT - ti * (ti - 1) / 2 + (tj - ti); }

template <typename T, int NT>
__global__ __launch_bounds__(kThreads, 1) void cov_fused_kernel(const T* __restrict__ x, int F, int64_t ld, FrameTab ft,
                                                               const double* __restrict__ mu,
                                                               int64_t frames_per_wave, double* __restrict__ slabs) {
    using S = CovShape<NT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* red = reinterpret_cast<double*>(smem_raw);  // [kTiles*256] then [kWaves][2][NT][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int fi_ = lane & 15;
    const int kk = lane >> 4;
    const int lag = ft.lag;

    v4f64 acc0t[NT][NT];
    v4f64 acc00[S::kSym];
    double sx[NT], sy[NT], shift[NT];
    bool fok[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
#pragma unroll
        for (int b = 0; b < NT; ++b) acc0t[a][b] = (v4f64){0.0, 0.0, 0.0, 0.0};
        sx[a] = sy[a] = 0.0;
        const int f = 16 * a + fi_;
        fok[a] = f < F;
        shift[a] = fok[a] ? mu[f] : 0.0;
    }
#pragma unroll
    for (int a = 0; a < S::kSym; ++a) acc00[a] = (v4f64){0.0, 0.0, 0.0, 0.0};

    const int64_t gw = (int64_t)blockIdx.x * kWaves + wave;
    const int64_t q_begin = gw * frames_per_wave;
    const int64_t q_end = min(q_begin + frames_per_wave, ft.total);

    // raw values of the group being prefetched
    T rx[kUnroll][NT], ry[kUnroll][NT];
    FrameInfo info[kUnroll];

    auto issue_loads = [&](int64_t q0) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            info[u] = locate_frame(ft, q0 + 4 * u + kk, q_end);
            const T* px = x + info[u].t * ld + fi_;
            const T* py = x + (info[u].t + (info[u].in_x0 ? lag : 0)) * ld + fi_;
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                rx[u][a] = (info[u].valid && fok[a]) ? px[16 * a] : (T)0;
                ry[u][a] = (info[u].in_x0 && fok[a]) ? py[16 * a] : (T)0;
            }
        }
    };

    if (q_begin < q_end) issue_loads(q_begin);
    for (int64_t q0 = q_begin; q0 < q_end; q0 += 4 * kUnroll) {
        // consume the prefetched group into fp64 operands
        double za[kUnroll][NT], zb[kUnroll][NT], zy[kUnroll][NT];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const double w = (info[u].in_x0 ? 1.0 : 0.0) + (info[u].in_yt ? 1.0 : 0.0);
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                double vx = to_f64(rx[u][a]);
                double vy = to_f64(ry[u][a]);
                double cx = vx - shift[a];
                double cy = vy - shift[a];
                if (!(vx == vx) || !info[u].valid || !fok[a]) cx = 0.0;  // NaN -> column mean
                if (!(vy == vy) || !info[u].in_x0 || !fok[a]) cy = 0.0;
                za[u][a] = cx;
                zy[u][a] = cy;
                zb[u][a] = w * cx;
                sx[a] += info[u].in_x0 ? cx : 0.0;
                sy[a] += info[u].in_yt ? cx : 0.0;
            }
        }
        // next group's loads fly while the matrix cores work
        if (q0 + 4 * kUnroll < q_end) issue_loads(q0 + 4 * kUnroll);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
            for (int a = 0; a < NT; ++a) {
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    acc0t[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(za[u][a], zy[u][b], acc0t[a][b], 0, 0, 0);
#pragma unroll
                for (int b = a; b < NT; ++b)
                    acc00[sym_index<NT>(a, b)] =
                        __builtin_amdgcn_mfma_f64_16x16x4f64(za[u][a], zb[u][b], acc00[sym_index<NT>(a, b)], 0, 0, 0);
            }
        }
    }

    // ---- workgroup reduction in fixed wave order, then one slab per workgroup ----
    double* sums = red + S::kTiles * 256;  // [kWaves][2][NT][64]
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        sums[((wave * 2 + 0) * NT + a) * 64 + lane] = sx[a];
        sums[((wave * 2 + 1) * NT + a) * 64 + lane] = sy[a];
    }
    for (int w = 0; w < kWaves; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int idx = ((a * NT + b) * 4 + r) * 64 + lane;
                        red[idx] = (w == 0 ? 0.0 : red[idx]) + acc0t[a][b][r];
                    }
#pragma unroll
            for (int s = 0; s < S::kSym; ++s)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int idx = ((NT * NT + s) * 4 + r) * 64 + lane;
                    red[idx] = (w == 0 ? 0.0 : red[idx]) + acc00[s][r];
                }
        }
        __syncthreads();
    }
    double* slab = slabs + (size_t)blockIdx.x * S::kSlab;
    for (int i = tid; i < S::kTiles * 256; i += kThreads) slab[i] = red[i];
    // column sums: element e = which*NT*16 + a*16 + f ; add over waves and the 4 frame groups
    if (tid < 2 * NT * 16) {
        const int which = tid / (NT * 16);
        const int a = (tid / 16) % NT;
        const int f = tid % 16;
        double acc = 0.0;
        for (int w = 0; w < kWaves; ++w)
            for (int g = 0; g < 4; ++g) acc += sums[((w * 2 + which) * NT + a) * 64 + g * 16 + f];
        slab[S::kTiles * 256 + tid] = acc;
    }
}

// Adds the slabs in block order and scatters tiles into the moment block
//   out = [M00 F*F][M0t F*F][sx F][sy F][T]      (raw, centred by `mu`, unscaled)
template <int NT>
__global__ void cov_reduce_kernel(const double* __restrict__ slabs, int n_slabs, int F, double pairs,
                                  double* __restrict__ out) {
    using S = CovShape<NT>;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= S::kSlab) return;
    double acc = 0.0;
    for (int b = 0; b < n_slabs; ++b) acc += slabs[(size_t)b * S::kSlab + e];
    double* M00 = out;
    double* M0t = out + (size_t)F * F;
    double* sxy = M0t + (size_t)F * F;
    if (e < S::kTiles * 256) {
        const int tile = e >> 8;
        const int r = (e >> 6) & 3;
        const int lane = e & 63;
        const int row_in = (lane >> 4) + 4 * r;  // f64 MFMA C/D layout
        const int col_in = lane & 15;
        if (tile < NT * NT) {
            const int row = 16 * (tile / NT) + row_in, col = 16 * (tile % NT) + col_in;
            if (row < F && col < F) M0t[(size_t)row * F + col] = acc;
        } else {
            int s = tile - NT * NT, ti = 0;
            while (s >= NT - ti) { s -= NT - ti; ++ti; }
            const int tj = ti + s;
            const int row = 16 * ti + row_in, col = 16 * tj + col_in;
            if (row < F && col < F) {
                M00[(size_t)row * F + col] = acc;
                if (ti != tj) M00[(size_t)col * F + row] = acc;
            }
        }
    } else {
        const int i = e - S::kTiles * 256;
        const int which = i / (NT * 16);
        const int f = i % (NT * 16);
        if (f < F) sxy[(size_t)which * F + f] = acc;
    }
    if (e == 0) sxy[2 * (size_t)F] = pairs;
}

msm_status build_frametab(msm_ctx* ctx, int64_t n, const int64_t* h_start, const int64_t* h_stop, int n_seg, int lag,
                          FrameTab* out) {
    MSM_REQUIRE(ctx, lag >= 1, "lag must be >= 1 (got %d)", lag);
    FrameTab ft;
    memset(&ft, 0, sizeof(ft));
    ft.lag = lag;
    const int64_t one_a = 0, one_b = n;
    if (n_seg == 0) { h_start = &one_a; h_stop = &one_b; n_seg = 1; }
    MSM_REQUIRE(ctx, h_start && h_stop, "segment arrays are NULL");
    for (int s = 0; s < n_seg; ++s) {
        const int64_t a = h_start[s] < 0 ? 0 : h_start[s];
        const int64_t b = h_stop[s] > n ? n : h_stop[s];
        if (b - a <= lag) continue;
        if (ft.n == MSM_SEG_INLINE)
            return msm_fail(ctx, MSM_ERR_UNSUPPORTED, "msm_lagged_moments: more than %d segments per call; "
                            "accumulate over several calls", MSM_SEG_INLINE);
        ft.start[ft.n] = a;
        ft.stop[ft.n] = b;
        ft.prefix[ft.n + 1] = ft.prefix[ft.n] + (b - a);
        ft.pairs += (b - a) - lag;
        ++ft.n;
    }
    ft.total = ft.prefix[ft.n];
    *out = ft;
    return MSM_OK;
}

template <typename T, int NT>
msm_status launch_cov(msm_ctx* ctx, const T* x, int F, int64_t ld, const FrameTab& ft, const double* mu,
                      double* d_out) {
    using S = CovShape<NT>;
    int blocks = ctx->n_cu;  // one 4-wave workgroup per CU: each wave owns a SIMD's matrix core
    const int64_t groups = (ft.total + 4 * kUnroll - 1) / (4 * kUnroll);
    const int64_t min_groups_per_wave = 4;
    if ((int64_t)blocks * kWaves * min_groups_per_wave > groups)
        blocks = (int)std::max<int64_t>(1, groups / (kWaves * min_groups_per_wave));
    int64_t fpw = (ft.total + (int64_t)blocks * kWaves - 1) / ((int64_t)blocks * kWaves);
    fpw = (fpw + 4 * kUnroll - 1) / (4 * kUnroll) * (4 * kUnroll);
    blocks = (int)((ft.total + fpw * kWaves - 1) / (fpw * kWaves));
    msm_status rs = msm_reserve_scratch(ctx, (size_t)blocks * S::kSlab * sizeof(double));
    if (rs != MSM_OK) return rs;
    const size_t lds = ((size_t)S::kTiles * 256 + (size_t)kWaves * 2 * NT * 64) * sizeof(double);
    auto kern = cov_fused_kernel<T, NT>;
    if (lds > 64 * 1024)
        MSM_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(kThreads), lds, ctx->stream, x, F, ld, ft, mu, fpw,
                       (double*)ctx->scratch);
    MSM_CHECK_LAUNCH(ctx);
    hipLaunchKernelGGL(cov_reduce_kernel<NT>, dim3(msm_ceil_div(S::kSlab, 256)), dim3(256), 0, ctx->stream,
                       (const double*)ctx->scratch, blocks, F, (double)ft.pairs, d_out);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

template <typename T>
msm_status dispatch_cov(msm_ctx* ctx, const T* x, int F, int64_t ld, const FrameTab& ft, const double* mu,
                        double* d_out) {
    if (F <= 16) return launch_cov<T, 1>(ctx, x, F, ld, ft, mu, d_out);
    if (F <= 32) return launch_cov<T, 2>(ctx, x, F, ld, ft, mu, d_out);
    if (F <= 48) return launch_cov<T, 3>(ctx, x, F, ld, ft, mu, d_out);
    if (F <= 64) return launch_cov<T, 4>(ctx, x, F, ld, ft, mu, d_out);
    return msm_fail(ctx, MSM_ERR_UNSUPPORTED, "msm_lagged_moments: F=%d > 64 not supported yet", F);
}

}  // namespace

extern "C" {

msm_status msm_lagged_moments(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int F, int64_t ld,
                              const int64_t* h_seg_start, const int64_t* h_seg_stop, int n_seg, int lag,
                              const double* d_shift, double* d_moments) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 0 && F >= 1 && ld >= F, "msm_lagged_moments: need n >= 0, F >= 1, ld >= F");
    MSM_REQUIRE(ctx, dtype == MSM_F32 || dtype == MSM_F64, "msm_lagged_moments: bad dtype");
    MSM_REQUIRE(ctx, d_shift && d_moments && (d_x || n == 0), "msm_lagged_moments: NULL pointer");
    FrameTab ft;
    msm_status rs = build_frametab(ctx, n, h_seg_start, h_seg_stop, n_seg, lag, &ft);
    if (rs != MSM_OK) return rs;
    if (ft.total == 0) {
        MSM_HIP(ctx, hipMemsetAsync(d_moments, 0, ((size_t)2 * F * F + 2 * F + 1) * sizeof(double), ctx->stream));
        return MSM_OK;
    }
    if (dtype == MSM_F32) return dispatch_cov<float>(ctx, (const float*)d_x, F, ld, ft, d_shift, d_moments);
    return dispatch_cov<double>(ctx, (const double*)d_x, F, ld, ft, d_shift, d_moments);
}

}  // extern "C"
