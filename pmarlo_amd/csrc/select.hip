// Exact order statistics of a strided fp64 column by radix selection.
//
// Reference: the quantile rules behind the free-energy grids (scipy.stats.mstats.mquantiles and
// scipy.stats.iqr in generate_2d_fes, S/markov_state_model/free_energy.py:494-590) and any median need
// x_(r), the r-th smallest sample.  Sorting N samples for a handful of ranks is wasteful: the IEEE bit
// pattern of a double, with the sign bit flipped for positive and all bits flipped for negative values,
// orders like an unsigned integer, so the r-th key is found digit by digit: one histogram pass per 12-bit
// digit over the elements that share the prefix found so far (LDS-privatised bins, integer atomics), the
// host picks the bucket that holds rank r and descends.  Six passes per rank, exact, order independent.
#include <algorithm>
#include <cstring>

#include "common.h"

namespace {

constexpr int kST = 256;
constexpr int kDigitBits = 12;
constexpr int kBins = 1 << kDigitBits;

__device__ __forceinline__ unsigned long long sortable_key(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

__global__ __launch_bounds__(kST) void digit_histogram_kernel(const double* __restrict__ x, int64_t stride, int64_t n,
                                                              unsigned long long prefix, int prefix_bits, int width,
                                                              unsigned long long* __restrict__ hist) {
    __shared__ unsigned int bins[kBins];
    for (int i = threadIdx.x; i < kBins; i += kST) bins[i] = 0u;
    __syncthreads();
    const int shift = 64 - prefix_bits - width;
    const unsigned long long mask = (1ull << width) - 1ull;
    for (int64_t t = (int64_t)blockIdx.x * kST + threadIdx.x; t < n; t += (int64_t)gridDim.x * kST) {
        const unsigned long long key = sortable_key(x[t * stride]);
        if (prefix_bits == 0 || (key >> (64 - prefix_bits)) == prefix) atomicAdd(&bins[(key >> shift) & mask], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kBins; i += kST)
        if (bins[i]) atomicAdd(&hist[i], (unsigned long long)bins[i]);
}

}  // namespace

extern "C" {

msm_status msm_order_statistics(msm_ctx* ctx, const double* d_x, int64_t stride, int64_t n, const int64_t* h_ranks,
                                int n_ranks, double* h_out) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, d_x && h_ranks && h_out && n >= 1 && stride >= 1 && n_ranks >= 0, "msm_order_statistics: bad arguments");
    if (ctx->capturing) return msm_fail(ctx, MSM_ERR_UNSUPPORTED, "msm_order_statistics polls the host: not capturable");
    msm_status rs = msm_reserve_scratch(ctx, kBins * sizeof(unsigned long long));
    if (rs != MSM_OK) return rs;
    unsigned long long* d_hist = (unsigned long long*)ctx->scratch;
    std::vector<unsigned long long> hist(kBins);
    const int blocks = (int)std::min<int64_t>(std::max<int64_t>(1, msm_ceil_div(n, kST * 8)), (int64_t)ctx->n_cu * 4);
    for (int q = 0; q < n_ranks; ++q) {
        MSM_REQUIRE(ctx, h_ranks[q] >= 0 && h_ranks[q] < n, "msm_order_statistics: rank %lld outside [0, %lld)",
                    (long long)h_ranks[q], (long long)n);
        unsigned long long prefix = 0, want = (unsigned long long)h_ranks[q];
        int prefix_bits = 0;
        while (prefix_bits < 64) {
            const int width = std::min(kDigitBits, 64 - prefix_bits);
            MSM_HIP(ctx, hipMemsetAsync(d_hist, 0, kBins * sizeof(unsigned long long), ctx->stream));
            hipLaunchKernelGGL(digit_histogram_kernel, dim3(blocks), dim3(kST), 0, ctx->stream, d_x, stride, n, prefix,
                               prefix_bits, width, d_hist);
            MSM_CHECK_LAUNCH(ctx);
            MSM_HIP(ctx, hipMemcpyAsync(hist.data(), d_hist, kBins * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                                        ctx->stream));
            MSM_HIP(ctx, hipStreamSynchronize(ctx->stream));
            unsigned long long seen = 0;
            int b = 0;
            const int n_b = 1 << width;
            for (; b < n_b; ++b) {
                if (seen + hist[b] > want) break;
                seen += hist[b];
            }
            if (b == n_b) return msm_fail(ctx, MSM_ERR_HIP, "msm_order_statistics: histogram does not cover the rank");
            want -= seen;
            prefix = (prefix << width) | (unsigned long long)b;
            prefix_bits += width;
        }
        const unsigned long long bits = (prefix >> 63) ? (prefix & 0x7FFFFFFFFFFFFFFFull) : ~prefix;
        double v;
        static_assert(sizeof(v) == sizeof(bits), "double is 64 bits");
        memcpy(&v, &bits, sizeof(v));
        h_out[q] = v;
    }
    return MSM_OK;
}

}  // extern "C"
