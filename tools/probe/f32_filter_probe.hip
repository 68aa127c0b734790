// How fast could an fp32 matrix-core FILTER pass of the k-means distance table run (exact fp64
// refinement of near-ties not included)?  Same tile shape as the production loop: 16 centres x 16
// frames x 4 features per MFMA, 3 k-steps, NF frame groups per wave, two tiles per trip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float v4f32 __attribute__((ext_vector_type(4)));
typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int NT = 256, KS = 3;

template <int VAR, int NF, int THREADS>
__global__ __launch_bounds__(THREADS) void k(const float* __restrict__ in, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    __shared__ float cs[32 * KS * 64];
    for (int i = threadIdx.x; i < 32 * KS * 64; i += THREADS) cs[i] = in[i % 4096];
    __syncthreads();
    float b[NF][KS];
    for (int u = 0; u < NF; ++u) for (int s = 0; s < KS; ++s) b[u][s] = in[(3 + u * 3 + s) * 64 + lane];
    float best1[NF], best2[NF]; int bp[NF];
    for (int u = 0; u < NF; ++u) { best1[u] = -1e30f; best2[u] = -1e30f; bp[u] = 0; }
    for (int jt = 0; jt < NT; jt += 2) {
        v4f32 acca[NF], accb[NF];
#pragma unroll
        for (int u = 0; u < NF; ++u) { acca[u] = (v4f32){0, 0, 0, 0}; accb[u] = acca[u]; }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float fa = cs[((jt & 31) * KS + s) * 64 + lane];
#pragma unroll
            for (int u = 0; u < NF; ++u) acca[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, b[u][s], acca[u], 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float fb = cs[(((jt + 1) & 31) * KS + s) * 64 + lane];
#pragma unroll
            for (int u = 0; u < NF; ++u) accb[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb, b[u][s], accb[u], 0, 0, 0);
        }
        if constexpr (VAR == 0) {
            asm volatile("" ::"v"(acca[0]), "v"(accb[0]), "v"(acca[NF - 1]), "v"(accb[NF - 1]));
        } else {
            // pair maximum + top-2 tracking of pair maxima (what a certified filter needs)
#pragma unroll
            for (int u = 0; u < NF; ++u) {
                const float ma = fmaxf(fmaxf(acca[u][0], acca[u][1]), fmaxf(acca[u][2], acca[u][3]));
                const float mb = fmaxf(fmaxf(accb[u][0], accb[u][1]), fmaxf(accb[u][2], accb[u][3]));
                const float m = fmaxf(ma, mb);
                const bool better = m > best1[u];
                const float second = better ? best1[u] : m;
                best2[u] = fmaxf(best2[u], second);
                best1[u] = better ? m : best1[u];
                bp[u] = better ? jt : bp[u];
            }
        }
    }
    float t = 0;
    for (int u = 0; u < NF; ++u) t += best1[u] + best2[u] + bp[u];
    out[blockIdx.x * THREADS + threadIdx.x] = t;
}

template <int VAR, int NF, int THREADS>
int run(const char* name, const float* in, float* out) {
    const int blocks = 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<VAR, NF, THREADS><<<blocks, THREADS>>>(in, out);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) k<VAR, NF, THREADS><<<blocks, THREADS>>>(in, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double flops = (double)blocks * (THREADS / 64) * NT * NF * KS * 2048.0;
    printf("%-40s NF %d waves/SIMD %d  %.3f ms  %.1f TFLOP/s (fp32 matrix peak 157.3)\n", name, NF, THREADS / 256, ms, flops / ms / 1e9);
    return 0;
}

int main() {
    float *in, *out;
    std::vector<float> h(64 * 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = ((i * 2654435761u) % 1000) * 0.001f - 0.5f;
    CK(hipMalloc(&in, h.size() * 4)); CK(hipMalloc(&out, 256 * 1024 * 4));
    CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    run<0, 2, 1024>("f32 MFMA only", in, out);
    run<1, 2, 1024>("f32 MFMA + pair max + top-2 tracking", in, out);
    run<0, 4, 1024>("f32 MFMA only", in, out);
    run<1, 4, 1024>("f32 MFMA + pair max + top-2 tracking", in, out);
    run<0, 4, 512>("f32 MFMA only", in, out);
    run<1, 4, 512>("f32 MFMA + pair max + top-2 tracking", in, out);
    return 0;
}
