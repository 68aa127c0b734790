// What does a VALU instruction cost next to v_mfma_f64_16x16x4_f64?  One "tile" = 12 MFMAs
// (4 independent accumulator chains x 3 k-steps) followed by a block of VALU work of one kind.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int NT = 256;

template <int VAR, int THREADS>
__global__ __launch_bounds__(THREADS) void k(const double* __restrict__ in, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    __shared__ double cs[16 * 3 * 64];
    for (int i = threadIdx.x; i < 16 * 3 * 64; i += THREADS) cs[i] = in[i % 4096];
    __syncthreads();
    double a[3], b[4][3];
    for (int u = 0; u < 4; ++u) for (int s = 0; s < 3; ++s) b[u][s] = in[(3 + u * 3 + s) * 64 + lane];
    double best[4][4]; int bt[4][4];
    long long kbest[4][4];
    for (int u = 0; u < 4; ++u) for (int r = 0; r < 4; ++r) { best[u][r] = -1e300; bt[u][r] = 0; kbest[u][r] = (long long)0x8000000000000000ull; }
    double x[16]; int w[64];
    for (int i = 0; i < 16; ++i) x[i] = in[(20 + i) * 64 + lane];
    for (int i = 0; i < 64; ++i) w[i] = lane * 7 + i;
    for (int jt = 0; jt < NT; ++jt) {
        v4f64 acc[4];
#pragma unroll
        for (int s = 0; s < 3; ++s) a[s] = cs[((jt & 15) * 3 + s) * 64 + lane];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = (v4f64){0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[u][s], acc[u], 0, 0, 0);
        if constexpr (VAR == 0) {
            asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
        } else if constexpr (VAR == 1) {  // fp64 compare + selects on the MFMA results
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool better = acc[u][r] > best[u][r];
                    best[u][r] = better ? acc[u][r] : best[u][r];
                    bt[u][r] = better ? jt : bt[u][r];
                }
        } else if constexpr (VAR == 2) {  // order-preserving integer key + int64 compare
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long bits = __double_as_longlong(acc[u][r]);
                    const long long key = bits ^ ((bits >> 63) & 0x7fffffffffffffffll);
                    const bool better = key > kbest[u][r];
                    kbest[u][r] = better ? key : kbest[u][r];
                    bt[u][r] = better ? jt : bt[u][r];
                }
        } else {
            asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
            if constexpr (VAR == 3) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_cmp_gt_f64 vcc, %0, %1" ::"v"(x[i]), "v"(x[(i + 1) & 15]) : "vcc");
            } else if constexpr (VAR == 4) {
#pragma unroll
                for (int i = 0; i < 48; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(w[i]) : "v"(w[63 - i % 8]));
            } else if constexpr (VAR == 5) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_cmp_gt_i64 vcc, %0, %1" ::"v"(x[i]), "v"(x[(i + 1) & 15]) : "vcc");
            } else if constexpr (VAR == 6) {
#pragma unroll
                for (int i = 0; i < 64; ++i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(w[i]) : "v"(w[(i + 5) & 63]));
            } else if constexpr (VAR == 7) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x[i]) : "v"(x[(i + 3) & 15]));
            } else if constexpr (VAR == 8) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_cmp_gt_u32 vcc, %0, %1" ::"v"(w[i]), "v"(w[i + 1]) : "vcc");
            } else if constexpr (VAR == 9) {  // 4 fp64 ops only
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_cmp_gt_f64 vcc, %0, %1" ::"v"(x[i]), "v"(x[(i + 1) & 15]) : "vcc");
            }
        }
    }
    double t = 0;
    for (int u = 0; u < 4; ++u) for (int r = 0; r < 4; ++r) t += best[u][r] + bt[u][r] + (double)kbest[u][r];
    for (int i = 0; i < 16; ++i) t += x[i];
    for (int i = 0; i < 64; ++i) t += w[i];
    out[blockIdx.x * THREADS + threadIdx.x] = t;
}

template <int VAR, int THREADS>
int run(const char* name, const double* in, double* out) {
    const int blocks = 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<VAR, THREADS><<<blocks, THREADS>>>(in, out);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) k<VAR, THREADS><<<blocks, THREADS>>>(in, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double waves_per_simd = THREADS / 256.0;
    const double cyc_per_tile = ms * 1e-3 * 2.4e9 / (NT * waves_per_simd);  // SIMD cycles per wave-tile
    printf("%-44s waves/SIMD %.0f  %.3f ms  %.0f cycles per wave-tile (MFMA pipe floor 768)\n", name, waves_per_simd, ms, cyc_per_tile);
    return 0;
}

// software-pipelined: MFMAs of tile t+1 are issued BEFORE the arg-max of tile t
template <int VAR, int THREADS>
__global__ __launch_bounds__(THREADS) void kp(const double* __restrict__ in, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    __shared__ double cs[16 * 3 * 64];
    for (int i = threadIdx.x; i < 16 * 3 * 64; i += THREADS) cs[i] = in[i % 4096];
    __syncthreads();
    double b[4][3];
    for (int u = 0; u < 4; ++u) for (int s = 0; s < 3; ++s) b[u][s] = in[(3 + u * 3 + s) * 64 + lane];
    double best[4][4]; int bt[4][4];
    for (int u = 0; u < 4; ++u) for (int r = 0; r < 4; ++r) { best[u][r] = -1e300; bt[u][r] = 0; }
    auto mfma_tile = [&](int jt, v4f64 (&acc)[4]) {
        double a[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) a[s] = cs[((jt & 15) * 3 + s) * 64 + lane];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = (v4f64){0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[u][s], acc[u], 0, 0, 0);
    };
    auto epi = [&](int jt, const v4f64 (&acc)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool better = acc[u][r] > best[u][r];
                best[u][r] = better ? acc[u][r] : best[u][r];
                bt[u][r] = better ? jt : bt[u][r];
            }
    };
    v4f64 accA[4], accB[4];
    mfma_tile(0, accA);
    for (int jt = 0; jt < NT; jt += 2) {
        mfma_tile(jt + 1, accB);
        if constexpr (VAR == 1) __builtin_amdgcn_sched_barrier(0);
        epi(jt, accA);
        if constexpr (VAR == 1) __builtin_amdgcn_sched_barrier(0);
        mfma_tile(jt + 2, accA);
        if constexpr (VAR == 1) __builtin_amdgcn_sched_barrier(0);
        epi(jt + 1, accB);
        if constexpr (VAR == 1) __builtin_amdgcn_sched_barrier(0);
    }
    double t = 0;
    for (int u = 0; u < 4; ++u) for (int r = 0; r < 4; ++r) t += best[u][r] + bt[u][r] + accA[u][r];
    out[blockIdx.x * THREADS + threadIdx.x] = t;
}

template <int VAR, int THREADS>
int runp(const char* name, const double* in, double* out) {
    const int blocks = 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    kp<VAR, THREADS><<<blocks, THREADS>>>(in, out);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) kp<VAR, THREADS><<<blocks, THREADS>>>(in, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double waves_per_simd = THREADS / 256.0;
    printf("%-44s waves/SIMD %.0f  %.3f ms  %.0f cycles per wave-tile\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / (NT * waves_per_simd));
    return 0;
}
#define ALL(V, NAME) run<V, 256>(NAME, in, out); run<V, 512>(NAME, in, out); run<V, 1024>(NAME, in, out);
int main() {
    double *in, *out;
    std::vector<double> h(64 * 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = ((i * 2654435761u) % 1000) * 0.001 - 0.5;
    CK(hipMalloc(&in, h.size() * 8)); CK(hipMalloc(&out, 256 * 1024 * 8));
    CK(hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    runp<0, 256>("P0 pipelined epilogue", in, out); runp<0, 512>("P0 pipelined epilogue", in, out); runp<0, 1024>("P0 pipelined epilogue", in, out);
    runp<1, 256>("P1 pipelined + sched barriers", in, out); runp<1, 512>("P1 pipelined + sched barriers", in, out); runp<1, 1024>("P1 pipelined + sched barriers", in, out);
    ALL(0, "V0 MFMA only");
    ALL(1, "V1 + 16x(cmp_f64 + 3 cndmask) on results");
    ALL(3, "V3 + 16 independent v_cmp_gt_f64");
    ALL(4, "V4 + 48 independent v_cndmask_b32");
    return 0;
}
