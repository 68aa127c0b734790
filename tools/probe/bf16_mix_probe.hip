// How do v_mfma_f32_16x16x32_bf16, fp32 VALU work and LDS reads share a SIMD?  One iteration = NM matrix instructions
// (pairs that accumulate, as the k-means filter issues them), NV VALU instructions (v_max3_f32 on the accumulators when
// DEP, on private registers otherwise) and NL ds_read_b128 with a wait in front of the matrix instructions.
// One workgroup per CU, W waves per SIMD; prints SIMD cycles per wave-iteration next to the two floors
// (matrix pipe: 16 cycles per instruction; VALU: 4 cycles per instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v4f32 __attribute__((ext_vector_type(4)));
constexpr int NT = 2000;

template <int NMF, int NV, int NL, bool DEP, int THREADS>
__global__ __launch_bounds__(THREADS) void k(const uint4* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ cyc) {
    const int lane = threadIdx.x & 63;
    __shared__ uint4 cs[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += THREADS) cs[i] = in[i];
    __syncthreads();
    v8bf b[4][2];
    for (int u = 0; u < 4; ++u)
        for (int m = 0; m < 2; ++m) b[u][m] = __builtin_bit_cast(v8bf, in[(u * 2 + m) * 64 + lane]);
    v8bf a[4];
    for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(v8bf, in[(8 + i) * 64 + lane]);
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)(lane + i);
    float best[4] = {-1e30f, -1e30f, -1e30f, -1e30f};
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int jt = 0; jt < NT; ++jt) {
        if constexpr (NL > 0) {
#pragma unroll
            for (int i = 0; i < NL; ++i) a[i & 3] = __builtin_bit_cast(v8bf, cs[((jt * 4 + i) & 63) * 64 + lane]);
        }
        v4f32 acc[8];
        if constexpr (NMF > 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[2 * u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[u][0], (v4f32){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                acc[2 * u + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[u][0], (v4f32){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            }
            if constexpr (NMF > 8) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[2 * u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[u][1], acc[2 * u], 0, 0, 0);
                    acc[2 * u + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[3], b[u][1], acc[2 * u + 1], 0, 0, 0);
                }
            }
        }
        if constexpr (DEP && NMF > 0) {
            // the filter's top-two bookkeeping shape: 4 v_max3 per frame group on the accumulators, repeated to NV
#pragma unroll
            for (int r = 0; r < NV / 16; ++r)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float m = __builtin_fmaxf(__builtin_fmaxf(best[u], acc[2 * u][0]), acc[2 * u][1]);
                    m = __builtin_fmaxf(__builtin_fmaxf(m, acc[2 * u][2]), acc[2 * u][3]);
                    m = __builtin_fmaxf(__builtin_fmaxf(m, acc[2 * u + 1][0]), acc[2 * u + 1][1]);
                    m = __builtin_fmaxf(__builtin_fmaxf(m, acc[2 * u + 1][2]), acc[2 * u + 1][3]);
                    best[u] = m;
                }
        } else {
            if constexpr (NMF > 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(acc[i]));
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i & 7]) : "v"(x[(i + 3) & 7]), "v"(x[(i + 5) & 7]));
        }
    }
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float t = 0;
    for (int i = 0; i < 8; ++i) t += x[i];
    for (int u = 0; u < 4; ++u) t += best[u];
    for (int i = 0; i < 4; ++i) t += (float)a[i][0];
    out[blockIdx.x * THREADS + threadIdx.x] = t;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// the 32 x 32 x 16 shape: NMF instructions of 32 cycles each (same flops per cycle), 16 accumulators per lane
typedef float v16f32 __attribute__((ext_vector_type(16)));
template <int NMF, int NV, int NL, int THREADS>
__global__ __launch_bounds__(THREADS) void k32(const uint4* __restrict__ in, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    __shared__ uint4 cs[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += THREADS) cs[i] = in[i];
    __syncthreads();
    v8bf b[2][4];
    for (int u = 0; u < 2; ++u)
        for (int m = 0; m < 4; ++m) b[u][m] = __builtin_bit_cast(v8bf, in[(u * 4 + m) * 64 + lane]);
    v8bf a[4];
    for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(v8bf, in[(8 + i) * 64 + lane]);
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)(lane + i);
    for (int jt = 0; jt < NT; ++jt) {
#pragma unroll
        for (int i = 0; i < NL; ++i) a[i & 3] = __builtin_bit_cast(v8bf, cs[((jt * 4 + i) & 63) * 64 + lane]);
        v16f32 acc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[u][0], (v16f32){0.f}, 0, 0, 0);
#pragma unroll
            for (int m = 1; m < NMF / 2; ++m) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m & 3], b[u][m & 3], acc[u], 0, 0, 0);
        }
        asm volatile("" ::"v"(acc[0]), "v"(acc[1]));
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i & 7]) : "v"(x[(i + 3) & 7]), "v"(x[(i + 5) & 7]));
    }
    float t = 0;
    for (int i = 0; i < 8; ++i) t += x[i];
    for (int i = 0; i < 4; ++i) t += (float)a[i][0];
    out[blockIdx.x * THREADS + threadIdx.x] = t;
}
template <int NMF, int NV, int NL, int THREADS>
int run32(const uint4* in, float* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k32<NMF, NV, NL, THREADS><<<256, THREADS>>>(in, out);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) k32<NMF, NV, NL, THREADS><<<256, THREADS>>>(in, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    const int w = THREADS / 256;
    printf("mfma32 %2d valu %3d (private) lds %d  waves/SIMD %d: %7.3f ms  %6.0f ns per wave-iteration per SIMD (= %5.0f cycles at 2.4 GHz); floors: matrix %4d, valu %4d cycles\n",
           NMF, NV, NL, w, ms, ms * 1e6 / (NT * w), ms * 1e-3 * 2.4e9 / (NT * w), NMF * 32, (NV + NMF) * 4);
    return 0;
}

template <int NMF, int NV, int NL, bool DEP, int THREADS>
int run(const uint4* in, float* out, unsigned long long* cyc) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<NMF, NV, NL, DEP, THREADS><<<256, THREADS>>>(in, out, cyc);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) k<NMF, NV, NL, DEP, THREADS><<<256, THREADS>>>(in, out, cyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const int w = THREADS / 256;
    // s_memtime ticks at 100 MHz on this part: the shader clock follows from the event time
    printf("mfma %2d valu %3d (%s) lds %d  waves/SIMD %d: %7.3f ms  %6.0f ns per wave-iteration per SIMD (= %5.0f cycles at 2.4 GHz); floors: matrix %4d, valu %4d cycles; memtime ticks %llu\n",
           NMF, NV, DEP ? "on acc" : "private", NL, w, ms, ms * 1e6 / (NT * w), ms * 1e-3 * 2.4e9 / (NT * w), NMF * 16, (NV + NMF) * 4, c);
    return 0;
}

#define RUNW(NMF, NV, NL, DEP)                                  \
    if (run<NMF, NV, NL, DEP, 256>(in, out, cyc)) return 1;     \
    if (run<NMF, NV, NL, DEP, 512>(in, out, cyc)) return 1;     \
    if (run<NMF, NV, NL, DEP, 768>(in, out, cyc)) return 1;     \
    if (run<NMF, NV, NL, DEP, 1024>(in, out, cyc)) return 1;

int main() {
    uint4* in; float* out; unsigned long long* cyc;
    CK(hipMalloc(&in, 64 * 64 * 16)); CK(hipMalloc(&out, 256 * 1024 * 4)); CK(hipMalloc(&cyc, 8));
    std::vector<unsigned short> h(64 * 64 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3F80 + (i * 7) % 64);   // bf16 numbers near 1
    CK(hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    RUNW(16, 0, 4, false)
    RUNW(8, 0, 4, false)
    if (run32<8, 0, 4, 512>(in, out)) return 1;
    if (run32<8, 0, 4, 1024>(in, out)) return 1;
    if (run32<8, 32, 4, 512>(in, out)) return 1;
    if (run32<8, 32, 4, 1024>(in, out)) return 1;
    if (run32<8, 64, 4, 512>(in, out)) return 1;
    if (run32<8, 64, 4, 1024>(in, out)) return 1;
    RUNW(0, 32, 0, false)
    RUNW(0, 64, 0, false)
    RUNW(16, 32, 0, false)
    RUNW(16, 64, 0, false)
    RUNW(16, 32, 0, true)
    RUNW(16, 32, 4, true)
    RUNW(16, 64, 4, false)
    RUNW(8, 32, 0, false)
    RUNW(8, 64, 0, false)
    return 0;
}
