// Which structure of the k-means tile loop keeps the fp64 matrix core busy?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int KS = 3, NF = 4, NT = 32;   // 32 tiles of 16 centres, like k = 500

template <int VAR>
__global__ __launch_bounds__(512, 2) void loop_kernel(const double* __restrict__ cin, const double* __restrict__ yin,
                                                      int units, int* __restrict__ out) {
    __shared__ double cs[NT * KS * 64];
    __shared__ double csq[NT * 16];
    for (int i = threadIdx.x; i < NT * KS * 64; i += 512) cs[i] = cin[i];
    for (int i = threadIdx.x; i < NT * 16; i += 512) csq[i] = 1.0 + 0.001 * i;
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 4;
    int total = 0;
    for (int unit = 0; unit < units; ++unit) {
        double zb[NF][KS];
        for (int u = 0; u < NF; ++u) for (int s = 0; s < KS; ++s) zb[u][s] = yin[((unit * 7 + u) * KS + s) * 64 + lane];
        double best[NF]; int bidx[NF];
        for (int u = 0; u < NF; ++u) { best[u] = 1e300; bidx[u] = 0; }
        auto mfma_tile = [&](int jt, v4f64 (&acc)[NF]) {
            double af[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) af[s] = cs[(jt * KS + s) * 64 + lane];
#pragma unroll
            for (int u = 0; u < NF; ++u) acc[u] = (v4f64){0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int u = 0; u < NF; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[s], zb[u][s], acc[u], 0, 0, 0);
        };
        auto epi_tile = [&](int jt, const v4f64 (&acc)[NF]) {
            double cq[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) cq[r] = csq[jt * 16 + g + 4 * r];
#pragma unroll
            for (int u = 0; u < NF; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double dist = fma(-2.0, acc[u][r], cq[r]);
                    if (dist < best[u]) { best[u] = dist; bidx[u] = jt * 16 + g + 4 * r; }
                }
        };
        if constexpr (VAR == 0) {           // clustered: MFMAs then epilogue
            for (int jt = 0; jt < NT; ++jt) { v4f64 acc[NF]; mfma_tile(jt, acc); epi_tile(jt, acc); }
        } else if constexpr (VAR == 1 || VAR == 2) {  // pipelined (1: with sched hints, 2: without)
            v4f64 accA[NF], accB[NF];
            mfma_tile(0, accA);
            for (int jt = 0; jt < NT; jt += 2) {
                mfma_tile(jt + 1 < NT ? jt + 1 : NT - 1, accB);
                epi_tile(jt, accA);
                if constexpr (VAR == 1) {
#pragma unroll
                    for (int i = 0; i < NF * KS; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 7, 0); }
                }
                mfma_tile(jt + 2 < NT ? jt + 2 : NT - 1, accA);
                epi_tile(jt + 1, accB);
                if constexpr (VAR == 1) {
#pragma unroll
                    for (int i = 0; i < NF * KS; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 7, 0); }
                }
            }
        } else if constexpr (VAR == 3) {    // MFMA only
            v4f64 acc[NF];
            for (int jt = 0; jt < NT; ++jt) { mfma_tile(jt, acc); asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3])); }
            best[0] = acc[0][0];
        } else if constexpr (VAR == 4) {    // epilogue only
            v4f64 acc[NF];
            for (int u = 0; u < NF; ++u) acc[u] = (v4f64){zb[u][0], zb[u][1], zb[u][2], zb[u][0]};
            for (int jt = 0; jt < NT; ++jt) { epi_tile(jt, acc); acc[jt & 3][jt & 3] += 1e-3; }
        } else if constexpr (VAR == 6) {    // clustered, deferred index: per tile only min-of-4 + tile id
            int btile[NF];
            for (int u = 0; u < NF; ++u) btile[u] = 0;
            for (int jt = 0; jt < NT; ++jt) {
                v4f64 acc[NF]; mfma_tile(jt, acc);
                double cq[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) cq[r] = csq[jt * 16 + g + 4 * r];
#pragma unroll
                for (int u = 0; u < NF; ++u) {
                    const double d0 = fma(-2.0, acc[u][0], cq[0]), d1 = fma(-2.0, acc[u][1], cq[1]);
                    const double d2 = fma(-2.0, acc[u][2], cq[2]), d3 = fma(-2.0, acc[u][3], cq[3]);
                    const double m = fmin(fmin(d0, d1), fmin(d2, d3));
                    if (m < best[u]) { best[u] = m; btile[u] = jt; }
                }
            }
            for (int u = 0; u < NF; ++u) bidx[u] = btile[u];
        } else if constexpr (VAR == 5) {    // clustered, epilogue without the index bookkeeping (min only)
            for (int jt = 0; jt < NT; ++jt) {
                v4f64 acc[NF]; mfma_tile(jt, acc);
                for (int u = 0; u < NF; ++u) for (int r = 0; r < 4; ++r) best[u] = fmin(best[u], fma(-2.0, acc[u][r], csq[jt * 16 + g + 4 * r]));
            }
        }
        for (int u = 0; u < NF; ++u) total += bidx[u] + (best[u] < 0.5);
    }
    out[blockIdx.x * 512 + threadIdx.x] = total;
}

template <int VAR>
int run(const char* name, const double* c, const double* y, int* out) {
    const int units = 8, blocks = 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    loop_kernel<VAR><<<blocks, 512>>>(c, y, units, out);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) loop_kernel<VAR><<<blocks, 512>>>(c, y, units, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double mfmas = (double)blocks * 8 * units * NT * NF * KS;
    printf("%-34s %.3f ms  MFMA-equivalent rate %.1f TFLOP/s (%.0f%% of 78.6)\n", name, ms, mfmas * 2048 / ms / 1e9, mfmas * 2048 / ms / 1e9 / 78.6 * 100);
    return 0;
}

int main() {
    double *c, *y; int* out;
    std::vector<double> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = ((i * 2654435761u) % 1000) * 0.001 - 0.5;
    CK(hipMalloc(&c, h.size() * 8)); CK(hipMalloc(&y, h.size() * 8)); CK(hipMalloc(&out, 256 * 512 * 4));
    CK(hipMemcpy(c, h.data(), h.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(y, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    run<0>("V0 clustered (current)", c, y, out);
    run<1>("V1 pipelined + sched hints", c, y, out);
    run<2>("V2 pipelined, no hints", c, y, out);
    run<3>("V3 MFMA only", c, y, out);
    run<4>("V4 arg-min epilogue only", c, y, out);
    run<5>("V5 clustered, min without index", c, y, out);
    run<6>("V6 clustered, deferred index", c, y, out);
    return 0;
}
