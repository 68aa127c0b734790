// Shared internals of libmsmhip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/msmhip.h"

struct msm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool capturing = false;
    int n_cu = 256;
    std::string err;
    // device scratch for per-workgroup partial slabs; grows monotonically and
    // is only (re)allocated outside stream capture.
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
    // second, small scratch for tables a kernel needs WHILE the main scratch is lent to its caller
    // (half-norms of the k-means centres during msm_kmeans_fit, whose member sums live in `scratch`)
    void* aux = nullptr;
    size_t aux_bytes = 0;
    // bf16 frame images of the k-means filter when the caller did not supply one (msm_kmeans_pack)
    void* km_image = nullptr;
    size_t km_image_bytes = 0;
    void* km_stats = nullptr;   // device u64: frames that took the filter's exhaustive scan
    // pinned host staging for small tables (segment lists with > MSM_SEG_INLINE entries)
    void* pinned = nullptr;
    size_t pinned_bytes = 0;
    void* dtab = nullptr;
    size_t dtab_bytes = 0;
};

struct msm_event {
    hipEvent_t ev;
};
struct msm_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

msm_status msm_fail(msm_ctx* ctx, msm_status st, const char* fmt, ...);
// Ensure ctx->scratch holds at least `bytes`; fails during capture if growth is needed.
msm_status msm_reserve_scratch(msm_ctx* ctx, size_t bytes);
msm_status msm_reserve_aux(msm_ctx* ctx, size_t bytes);
msm_status msm_reserve_km_image(msm_ctx* ctx, size_t bytes);

#define MSM_HIP(ctx, call)                                                              \
    do {                                                                                \
        hipError_t e__ = (call);                                                        \
        if (e__ != hipSuccess)                                                          \
            return msm_fail((ctx), e__ == hipErrorOutOfMemory ? MSM_ERR_NOMEM : MSM_ERR_HIP, \
                            "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__),     \
                            __FILE__, __LINE__);                                        \
    } while (0)

#define MSM_CHECK_LAUNCH(ctx)                                                           \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess)                                                          \
            return msm_fail((ctx), MSM_ERR_HIP, "kernel launch failed: %s (%s:%d)",     \
                            hipGetErrorString(e__), __FILE__, __LINE__);                \
    } while (0)

#define MSM_REQUIRE(ctx, cond, ...)                                                     \
    do {                                                                                \
        if (!(cond)) return msm_fail((ctx), MSM_ERR_INVALID, __VA_ARGS__);              \
    } while (0)

// ---------------------------------------------------------------------------
// Trajectory segments.  A "pair" is a frame index t whose partner t+lag lies in
// the same segment; pairs are numbered globally in segment order so that a
// kernel can walk a dense range of pair ids with coalesced label/feature loads.
// Up to MSM_SEG_INLINE segments travel by value in the kernel arguments (safe
// under hipGraph capture); longer lists go through a device table.
// ---------------------------------------------------------------------------
#define MSM_SEG_INLINE 16

struct SegTab {
    int n;                                // number of segments with >= 1 pair
    int stride;                           // step between pair starts
    int lag;
    int use_table;                        // 0: inline arrays, 1: device table
    int64_t start[MSM_SEG_INLINE];        // first frame of the segment
    int64_t prefix[MSM_SEG_INLINE + 1];   // pair-id prefix sum
    const int64_t* d_start;               // device table variant
    const int64_t* d_prefix;
    int64_t total_pairs;
};

// Builds the table for `lag`/`stride` from host segment bounds (clipped to [0,n]).
msm_status msm_build_segtab(msm_ctx* ctx, int64_t n, const int64_t* h_start,
                            const int64_t* h_stop, int n_seg, int lag, int stride,
                            SegTab* out, int table_slot);

__device__ __forceinline__ int64_t seg_pair_to_frame(const SegTab& st, int64_t p) {
    if (!st.use_table) {
        int s = 0;
#pragma unroll 1
        while (s + 1 < st.n && p >= st.prefix[s + 1]) ++s;
        return st.start[s] + (p - st.prefix[s]) * (int64_t)st.stride;
    }
    int lo = 0, hi = st.n - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (st.d_prefix[mid] <= p) lo = mid; else hi = mid - 1;
    }
    return st.d_start[lo] + (p - st.d_prefix[lo]) * (int64_t)st.stride;
}

static inline int msm_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
