// Cost of __syncthreads() by workgroup size (one workgroup alone on the chip), and of a barrier that the waves reach
// at different times (wave 0 does `work` dependent fp64 FMAs first).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned long long now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }
__global__ void k(double* out, unsigned long long* ticks, int iters, int work, int workers) {
    __shared__ double sh[1024];
    double x = out[threadIdx.x];
    const int wave = threadIdx.x >> 6;
    const unsigned long long t0 = now();
    for (int i = 0; i < iters; ++i) {
        if (wave < workers) {
            for (int w = 0; w < work; ++w) x = fma(x, 0.999, 1.5);
            sh[threadIdx.x] = x;
        }
        __syncthreads();
        if (wave < workers) x += sh[(threadIdx.x + 64) & 1023];
        __syncthreads();
    }
    const unsigned long long t1 = now();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
int main() {
    double* out; unsigned long long* tk;
    hipMalloc(&out, 1024 * 8); hipMalloc(&tk, 8); hipMemset(out, 0, 1024 * 8);
    for (int threads : {128, 256, 512, 1024})
        for (int workers : {1, 4, 16})
            for (int work : {0, 32}) {
                if (workers * 64 > threads) continue;
                k<<<1, threads>>>(out, tk, 1000, work, workers);
                k<<<1, threads>>>(out, tk, 1000, work, workers);
                hipDeviceSynchronize();
                unsigned long long t; hipMemcpy(&t, tk, 8, hipMemcpyDeviceToHost);
                printf("%4d threads, %2d working waves, %2d dependent FMAs + LDS write/read per round: %7.1f cycles per round (two barriers)\n", threads, workers, work, t / 1000.0);
            }
    return 0;
}
