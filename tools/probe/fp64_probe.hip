// fp64 VALU / MFMA issue-rate and latency probe (single wave and full chip).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s\n", hipGetErrorString(e)); return 1; } } while (0)
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int ILP>
__global__ void fma_ilp(double* out, int iters, unsigned long long* cyc) {
    double a[ILP];
    for (int i = 0; i < ILP; ++i) a[i] = out[threadIdx.x] + i;
    const double b = 1.0000001, c = 1e-9;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) a[i] = fma(a[i], b, c);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < ILP; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int ILP>
__global__ void mfma_ilp(double* out, int iters, unsigned long long* cyc) {
    v4f64 acc[ILP];
    for (int i = 0; i < ILP; ++i) acc[i] = (v4f64){0, 0, 0, 0};
    double a = out[threadIdx.x], b = 1.0 + threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < ILP; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename K>
int run(const char* name, K kern, int ilp, int blocks, int threads, int iters, double flop_per_inst, double* d, unsigned long long* c) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    kern<<<blocks, threads>>>(d, iters, c);  // warm
    CK(hipEventRecord(e0));
    kern<<<blocks, threads>>>(d, iters, c);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h; CK(hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost));
    double insts = (double)iters * ilp;
    double total_flop = insts * flop_per_inst * blocks * (threads / 64);
    printf("%-10s ilp=%2d grid=%4dx%4d: %.2f cycles/inst/wave, %.3f ms, %.2f TFLOP/s\n", name, ilp, blocks, threads,
           (double)h / insts, ms, total_flop / ms / 1e9);
    return 0;
}

int main() {
    double* d; unsigned long long* c;
    CK(hipMalloc(&d, 1024 * 2048 * 8)); CK(hipMalloc(&c, 16)); CK(hipMemset(d, 0, 1024 * 2048 * 8));
    const int it = 200000;
    // single wave: latency vs ILP
    run("fma64", fma_ilp<1>, 1, 1, 64, it, 128, d, c);
    run("fma64", fma_ilp<2>, 2, 1, 64, it, 128, d, c);
    run("fma64", fma_ilp<4>, 4, 1, 64, it, 128, d, c);
    run("fma64", fma_ilp<8>, 8, 1, 64, it, 128, d, c);
    run("fma64", fma_ilp<16>, 16, 1, 64, it, 128, d, c);
    // full chip
    run("fma64", fma_ilp<8>, 8, 1024, 256, it / 4, 128, d, c);
    run("fma64", fma_ilp<8>, 8, 2048, 256, it / 4, 128, d, c);
    run("fma64", fma_ilp<16>, 16, 2048, 256, it / 4, 128, d, c);
    run("mfma64", mfma_ilp<1>, 1, 1, 64, it, 2048, d, c);
    run("mfma64", mfma_ilp<2>, 2, 1, 64, it, 2048, d, c);
    run("mfma64", mfma_ilp<4>, 4, 1, 64, it, 2048, d, c);
    run("mfma64", mfma_ilp<8>, 8, 1, 64, it, 2048, d, c);
    run("mfma64", mfma_ilp<8>, 8, 256, 256, it / 4, 2048, d, c);
    run("mfma64", mfma_ilp<8>, 8, 512, 256, it / 4, 2048, d, c);
    run("mfma64", mfma_ilp<4>, 4, 1024, 256, it / 4, 2048, d, c);
    return 0;
}
