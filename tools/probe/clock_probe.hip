// Single-workgroup latency probe: what clock does a lone CU run at, what does a barrier cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void dep_fma(double* out, int iters, unsigned long long* cyc) {
    double a = out[threadIdx.x], b = 1.0000001;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) a = fma(a, b, 1e-9);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}

__global__ void barrier_loop(double* out, int iters, unsigned long long* cyc) {
    __shared__ double sh[1024];
    double a = out[threadIdx.x];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        sh[threadIdx.x] = a;
        __syncthreads();
        a += sh[(threadIdx.x + 1) % blockDim.x];
        __syncthreads();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    double* d; unsigned long long* c;
    hipMalloc(&d, 1024 * 8); hipMalloc(&c, 16);
    hipMemset(d, 0, 1024 * 8);
    unsigned long long h[2];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        dep_fma<<<1, 64>>>(d, 1000000, c);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
        printf("dep_fma 1M iters: %.3f ms, %llu shader cycles (%.2f cyc/fma), realtime ticks %llu (100MHz) -> clock %.2f GHz\n",
               ms, h[0], h[0] / 1e6, h[1], h[0] / (h[1] * 10.0));
    }
    for (int threads : {64, 256, 1024}) {
        hipEventRecord(e0);
        barrier_loop<<<1, threads>>>(d, 10000, c);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, c, 8, hipMemcpyDeviceToHost);
        printf("barrier_loop %4d threads: %.3f ms, %.1f cycles per (write,sync,read,sync) iteration\n", threads, ms, h[0] / 1e4);
    }
    return 0;
}
