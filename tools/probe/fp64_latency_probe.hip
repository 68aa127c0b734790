// Latency of dependent fp64 operations (one wave, and 16 waves of one workgroup): cycles per op from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned long long now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }
template <int MODE>
__global__ void chain(double* out, unsigned long long* ticks, double a, double b) {
    double x = out[threadIdx.x];
    const unsigned long long t0 = now();
#pragma unroll 1
    for (int i = 0; i < 256; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0) x = fma(x, a, b);
            if (MODE == 1) x = x * a + b;                       // mul + add (contract off)
            if (MODE == 2) x = __builtin_amdgcn_rcp(x) + b;     // rcp + add
            if (MODE == 3) x = __builtin_amdgcn_rsq(x) + b;
            if (MODE == 4) { float f = (float)x; f = fmaf(f, (float)a, (float)b); x = (double)f; }  // cvt round trip + fp32 fma
            if (MODE == 5) x = 1.0 / x + b;                     // IEEE division
            if (MODE == 6) x = sqrt(x) + b;
        }
    }
    const unsigned long long t1 = now();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
template <int MODE>
void run(const char* name, int threads, int ops_per_iter) {
    double* out; unsigned long long* tk;
    hipMalloc(&out, 1024 * 8); hipMalloc(&tk, 8);
    hipMemset(out, 0x3f, 1024 * 8);
    chain<MODE><<<1, threads>>>(out, tk, 0.999, 1.5);
    chain<MODE><<<1, threads>>>(out, tk, 0.999, 1.5);
    hipDeviceSynchronize();
    unsigned long long t; hipMemcpy(&t, tk, 8, hipMemcpyDeviceToHost);
    printf("%-34s %4d threads: %7.1f ticks per step (%d dependent ops per step)\n", name, threads, t / 4096.0, ops_per_iter);
    hipFree(out); hipFree(tk);
}
int main() {
    for (int th : {64, 1024}) {
        run<0>("v_fma_f64", th, 1);
        run<1>("v_mul_f64 + v_add_f64", th, 2);
        run<2>("v_rcp_f64 + v_add_f64", th, 2);
        run<3>("v_rsq_f64 + v_add_f64", th, 2);
        run<4>("cvt f64->f32, v_fma_f32, cvt back", th, 3);
        run<5>("IEEE 1/x + add", th, 2);
        run<6>("IEEE sqrt + add", th, 2);
    }
    return 0;
}
