// Regular-grid microstates (cluster_mode = "grid").
//
// Reference: _GridDiscretizer (S/analysis/discretize.py:517-593): per dimension
// idx = clip(np.digitize(x, edges) - 1, 0, bins - 1) on edges = linspace(min, max, bins + 1); the
// state of a frame is the order of FIRST APPEARANCE of its index combination in the data (a Python
// dict filled frame by frame).  On the device: (1) flat cell id per frame, (2) first frame index of
// every cell (integer atomicMin: exact, order-free), (3) relabel through the cell -> state table
// the host derives from (2).
#include "common.h"

namespace {

constexpr int kGT = 256;

// np.digitize(v, e) - 1 clipped to [0, bins - 1] for increasing e: the number of edges <= v, minus one.
// NaN sorts past the last edge in numpy (-> bins - 1 after the clip).
__device__ __forceinline__ int digitize_clip(const double* e, int bins, double inv_w, double v) {
    if (!(v == v)) return bins - 1;
    if (v < e[0]) return 0;
    if (v >= e[bins]) return bins - 1;
    int i = (int)((v - e[0]) * inv_w);
    i = i < 0 ? 0 : (i > bins - 1 ? bins - 1 : i);
    while (i > 0 && v < e[i]) --i;
    while (i < bins - 1 && v >= e[i + 1]) ++i;
    return i;
}

template <typename T>
__global__ __launch_bounds__(kGT) void grid_cells_kernel(const T* __restrict__ x, int64_t n, int F, int64_t ld,
                                                         const double* __restrict__ edges, int bins,
                                                         int32_t* __restrict__ flat) {
    extern __shared__ double le[];  // [F][bins + 1]
    for (int i = threadIdx.x; i < F * (bins + 1); i += kGT) le[i] = edges[i];
    __syncthreads();
    for (int64_t t = (int64_t)blockIdx.x * kGT + threadIdx.x; t < n; t += (int64_t)gridDim.x * kGT) {
        const T* row = x + t * ld;
        int cell = 0;
        for (int f = 0; f < F; ++f) {
            const double* e = le + f * (bins + 1);
            const double inv_w = bins / (e[bins] - e[0]);
            cell = cell * bins + digitize_clip(e, bins, inv_w, (double)row[f]);
        }
        flat[t] = cell;
    }
}

__global__ __launch_bounds__(kGT) void first_occurrence_kernel(const int32_t* __restrict__ flat, int64_t n, int n_cells,
                                                               unsigned long long* __restrict__ first) {
    for (int64_t t = (int64_t)blockIdx.x * kGT + threadIdx.x; t < n; t += (int64_t)gridDim.x * kGT) {
        const int c = flat[t];
        if ((unsigned)c < (unsigned)n_cells && (unsigned long long)t < first[c]) atomicMin(&first[c], (unsigned long long)t);
    }
}

__global__ __launch_bounds__(kGT) void relabel_kernel(const int32_t* __restrict__ flat, int64_t n, const int32_t* __restrict__ map,
                                                      int n_cells, int32_t* __restrict__ labels) {
    for (int64_t t = (int64_t)blockIdx.x * kGT + threadIdx.x; t < n; t += (int64_t)gridDim.x * kGT) {
        const int c = flat[t];
        labels[t] = (unsigned)c < (unsigned)n_cells ? map[c] : -1;
    }
}

int grid_for(const msm_ctx* ctx, int64_t n) {
    return (int)std::min<int64_t>(std::max<int64_t>(1, (n + kGT * 4 - 1) / (kGT * 4)), (int64_t)ctx->n_cu * 8);
}

}  // namespace

extern "C" {

msm_status msm_grid_cells(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int F, int64_t ld,
                          const double* d_edges, int bins, int32_t* d_flat) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 0 && F >= 1 && bins >= 1 && ld >= F, "msm_grid_cells: bad shape");
    MSM_REQUIRE(ctx, dtype == MSM_F32 || dtype == MSM_F64, "msm_grid_cells: bad dtype");
    double cells = 1.0;
    for (int f = 0; f < F; ++f) cells *= bins;
    MSM_REQUIRE(ctx, cells <= 2147483647.0, "msm_grid_cells: %d bins in %d dimensions overflow the cell index", bins, F);
    const size_t lds = (size_t)F * (bins + 1) * sizeof(double);
    MSM_REQUIRE(ctx, lds <= 64 * 1024, "msm_grid_cells: edge table too large");
    if (n == 0) return MSM_OK;
    MSM_REQUIRE(ctx, d_x && d_edges && d_flat, "msm_grid_cells: NULL pointer");
    if (dtype == MSM_F32)
        hipLaunchKernelGGL(grid_cells_kernel<float>, dim3(grid_for(ctx, n)), dim3(kGT), lds, ctx->stream, (const float*)d_x, n,
                           F, ld, d_edges, bins, d_flat);
    else
        hipLaunchKernelGGL(grid_cells_kernel<double>, dim3(grid_for(ctx, n)), dim3(kGT), lds, ctx->stream, (const double*)d_x,
                           n, F, ld, d_edges, bins, d_flat);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_first_occurrence(msm_ctx* ctx, const int32_t* d_flat, int64_t n, int n_cells, int64_t* d_first) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 0 && n_cells >= 1 && d_first && (n == 0 || d_flat), "msm_first_occurrence: bad arguments");
    MSM_HIP(ctx, hipMemsetAsync(d_first, 0xFF, (size_t)n_cells * sizeof(int64_t), ctx->stream));   // = -1 = "never"
    if (n == 0) return MSM_OK;
    hipLaunchKernelGGL(first_occurrence_kernel, dim3(grid_for(ctx, n)), dim3(kGT), 0, ctx->stream, d_flat, n, n_cells,
                       (unsigned long long*)d_first);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_relabel(msm_ctx* ctx, const int32_t* d_flat, int64_t n, const int32_t* d_map, int n_cells,
                       int32_t* d_labels) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 0 && n_cells >= 1 && (n == 0 || (d_flat && d_map && d_labels)), "msm_relabel: bad arguments");
    if (n == 0) return MSM_OK;
    hipLaunchKernelGGL(relabel_kernel, dim3(grid_for(ctx, n)), dim3(kGT), 0, ctx->stream, d_flat, n, d_map, n_cells, d_labels);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

}  // extern "C"
