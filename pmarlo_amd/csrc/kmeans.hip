// k-means assignment (and, further down, the fit kernels).
//
// Layout.  Frames are rows of X [n, ld] (f32 or f64) in HBM; a lane owns R
// frames and keeps their (optionally whitened) coordinates in registers as
// fp64.  Centres [k, d] fp64 are staged into LDS in tiles together with their
// squared norms; every lane walks the tile with broadcast LDS reads
// (ds_read_b128, all lanes on one address) and a D-long fp64 FMA chain per
// centre, so the inner loop is v_fma_f64-bound (2*k*d flop/frame against
// d*s bytes/frame: compute-bound for k >~ 40).
//
// Arithmetic is pinned so that the labels are bit-reproducible on the CPU:
//   dot  = fma(z[d-1], c[d-1], ... fma(z[0], c[0], +0.0))
//   dist = fma(-2.0, dot, csq)      csq = fma chain of c[f]*c[f]
//   argmin with strict '<' over ascending centre index.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kCentreLdsBudget = 48 * 1024;  // bytes of centre tile (+ norms) per workgroup

template <typename T>
__device__ __forceinline__ double load_as_f64(const T* p) { return (double)(*p); }

// D = padded feature count (compile time), R = frames per lane.
template <typename T, int D, int R>
__global__ __launch_bounds__(kThreads) void kmeans_assign_kernel(
    const T* __restrict__ x, int64_t n, int d, int64_t ld, const double* __restrict__ centers, int k,
    const double* __restrict__ mean, const double* __restrict__ stdv, int tile_k,
    int32_t* __restrict__ labels, double* __restrict__ mindist) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* cs = reinterpret_cast<double*>(smem_raw);  // [tile_k][D]
    double* csq = cs + (size_t)tile_k * D;              // [tile_k]

    const int tid = threadIdx.x;
    const int64_t frames_per_block = (int64_t)kThreads * R;
    const int64_t n_blocks = (n + frames_per_block - 1) / frames_per_block;

    for (int64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        double z[R][D];
        double zsq[R];
        int64_t fidx[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            fidx[r] = blk * frames_per_block + (int64_t)r * kThreads + tid;
            const bool ok = fidx[r] < n;
            const T* row = x + (ok ? fidx[r] : 0) * ld;
            double acc = 0.0;
#pragma unroll
            for (int f = 0; f < D; ++f) {
                double v = 0.0;
                if (f < d) {
                    v = load_as_f64(row + f);
                    if (mean) v = (v - mean[f]) / stdv[f];
                }
                z[r][f] = v;
                acc = fma(v, v, acc);
            }
            zsq[r] = acc;
        }
        double best[R];
        int bidx[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { best[r] = __builtin_inf(); bidx[r] = 0; }

        for (int k0 = 0; k0 < k; k0 += tile_k) {
            const int kt = min(tile_k, k - k0);
            __syncthreads();  // previous tile fully consumed
            for (int i = tid; i < kt * D; i += kThreads) {
                const int j = i / D, f = i - j * D;
                cs[i] = f < d ? centers[(size_t)(k0 + j) * d + f] : 0.0;
            }
            __syncthreads();
            for (int j = tid; j < kt; j += kThreads) {
                double a = 0.0;
#pragma unroll
                for (int f = 0; f < D; ++f) a = fma(cs[j * D + f], cs[j * D + f], a);
                csq[j] = a;
            }
            __syncthreads();
#pragma unroll 2
            for (int j = 0; j < kt; ++j) {
                const double* c = cs + j * D;
                double dot[R];
#pragma unroll
                for (int r = 0; r < R; ++r) dot[r] = 0.0;
#pragma unroll
                for (int f = 0; f < D; ++f) {
                    const double cf = c[f];
#pragma unroll
                    for (int r = 0; r < R; ++r) dot[r] = fma(z[r][f], cf, dot[r]);
                }
                const double cq = csq[j];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double dist = fma(-2.0, dot[r], cq);
                    if (dist < best[r]) { best[r] = dist; bidx[r] = k0 + j; }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (fidx[r] < n) {
                labels[fidx[r]] = bidx[r];
                if (mindist) {
                    const double m = best[r] + zsq[r];
                    mindist[fidx[r]] = m > 0.0 ? m : 0.0;
                }
            }
        }
    }
}

template <typename T, int D, int R>
msm_status launch_assign(msm_ctx* ctx, const T* x, int64_t n, int d, int64_t ld, const double* centers,
                         int k, const double* mean, const double* stdv, int32_t* labels, double* mindist) {
    int tile_k = kCentreLdsBudget / ((D + 1) * (int)sizeof(double));
    if (tile_k > k) tile_k = k;
    const size_t lds = (size_t)tile_k * (D + 1) * sizeof(double);
    const int64_t frames_per_block = (int64_t)kThreads * R;
    const int64_t n_blocks = (n + frames_per_block - 1) / frames_per_block;
    const int grid = (int)std::min<int64_t>(n_blocks, (int64_t)ctx->n_cu * 2);
    hipLaunchKernelGGL((kmeans_assign_kernel<T, D, R>), dim3(grid), dim3(kThreads), lds, ctx->stream, x, n, d, ld,
                       centers, k, mean, stdv, tile_k, labels, mindist);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

template <typename T>
msm_status dispatch_assign(msm_ctx* ctx, const T* x, int64_t n, int d, int64_t ld, const double* centers,
                           int k, const double* mean, const double* stdv, int32_t* labels, double* mindist) {
#define MSM_ASSIGN_CASE(DP, RR) \
    if (d <= DP) return launch_assign<T, DP, RR>(ctx, x, n, d, ld, centers, k, mean, stdv, labels, mindist)
    MSM_ASSIGN_CASE(2, 4);
    MSM_ASSIGN_CASE(4, 4);
    MSM_ASSIGN_CASE(6, 4);
    MSM_ASSIGN_CASE(8, 4);
    MSM_ASSIGN_CASE(10, 4);
    MSM_ASSIGN_CASE(12, 4);
    MSM_ASSIGN_CASE(16, 4);
    MSM_ASSIGN_CASE(24, 2);
    MSM_ASSIGN_CASE(32, 2);
    MSM_ASSIGN_CASE(48, 1);
    MSM_ASSIGN_CASE(64, 1);
#undef MSM_ASSIGN_CASE
    return msm_fail(ctx, MSM_ERR_UNSUPPORTED, "msm_kmeans_assign: d=%d > 64 not supported yet", d);
}

}  // namespace

extern "C" {

msm_status msm_kmeans_assign(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int d, int64_t ld,
                             const double* d_centers, int k, const double* d_mean, const double* d_std,
                             int32_t* d_labels, double* d_mindist) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 0 && d >= 1 && k >= 1, "msm_kmeans_assign: need n >= 0, d >= 1, k >= 1");
    MSM_REQUIRE(ctx, ld >= d, "msm_kmeans_assign: ld (%lld) < d (%d)", (long long)ld, d);
    MSM_REQUIRE(ctx, (d_mean == nullptr) == (d_std == nullptr),
                "msm_kmeans_assign: mean and std must both be given or both be NULL");
    MSM_REQUIRE(ctx, dtype == MSM_F32 || dtype == MSM_F64, "msm_kmeans_assign: bad dtype");
    if (n == 0) return MSM_OK;
    MSM_REQUIRE(ctx, d_x && d_centers && d_labels, "msm_kmeans_assign: NULL pointer");
    if (dtype == MSM_F32)
        return dispatch_assign<float>(ctx, (const float*)d_x, n, d, ld, d_centers, k, d_mean, d_std, d_labels,
                                      d_mindist);
    return dispatch_assign<double>(ctx, (const double*)d_x, n, d, ld, d_centers, k, d_mean, d_std, d_labels,
                                   d_mindist);
}

}  // extern "C"
