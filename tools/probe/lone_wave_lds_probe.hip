// What one wave ALONE on a CU (the other 15 waves of its workgroup asleep at a barrier) pays for LDS traffic:
// cycles per wave-instruction for straight-line streams of reads / writes / read-modify-write, and the price of a
// taken scalar branch.  Explains the design of the barrier-free Householder wave in eig.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned long long now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }
constexpr int LD = 65;
template <int MODE>
__global__ __launch_bounds__(1024) void k(double* out, unsigned long long* ticks, int n, double s) {
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 2 * 64 * LD; i += blockDim.x) lds[i] = 1.0 + 1e-3 * i;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        double* At = lds + lane;
        double acc0 = 0.0, acc1 = 0.0;
        const unsigned long long t0 = now();
#pragma unroll 1
        for (int rep = 0; rep < 16; ++rep) {
            if (MODE == 0) {   // 64 x ds_read_b64 (lane-contiguous), straight line, FMA each
#pragma unroll
                for (int j = 0; j < 64; j += 2) { acc0 = fma(At[j * LD], s, acc0); acc1 = fma(At[(j + 1) * LD], s, acc1); }
            }
            if (MODE == 1) {   // 64 x (ds_read_b64 lane-contiguous + ds_read_b64 broadcast)
#pragma unroll
                for (int j = 0; j < 64; j += 2) { acc0 = fma(At[j * LD], lds[64 * LD + j], acc0); acc1 = fma(At[(j + 1) * LD], lds[64 * LD + j + 1], acc1); }
            }
            if (MODE == 2) {   // 64 x ds_write_b64
#pragma unroll
                for (int j = 0; j < 64; ++j) At[j * LD] = s + j;
            }
            if (MODE == 3) {   // 64 x (read b64 + broadcast b128 + 2 fma + write b64): the rank-2 update, straight line
                const double2* vw = reinterpret_cast<const double2*>(lds + 64 * LD + 64);
#pragma unroll
                for (int j = 0; j < 64; ++j) { const double2 b = vw[j]; At[j * LD] = fma(-s, b.y, fma(-acc0, b.x, At[j * LD])); }
            }
            if (MODE == 4) {   // MODE 1 in a rolled loop of 4 rows per trip (one taken branch per 4 rows)
#pragma unroll 1
                for (int j = 0; j < n; j += 4) {
#pragma unroll
                    for (int u = 0; u < 4; u += 2) { acc0 = fma(At[(j + u) * LD], lds[64 * LD + j + u], acc0); acc1 = fma(At[(j + u + 1) * LD], lds[64 * LD + j + u + 1], acc1); }
                }
            }
            if (MODE == 5) {   // 64 x ds_read_b128 of the lane's own ROW (stride 66 doubles: 16-byte aligned), FMA x2
                const double2* row = reinterpret_cast<const double2*>(lds + lane * 66);
#pragma unroll
                for (int j = 0; j < 32; ++j) { const double2 a = row[j]; acc0 = fma(a.x, s, acc0); acc1 = fma(a.y, s, acc1); }
            }
            if (MODE == 6) {   // empty rolled loop: 64 taken branches + 1 v_add each
#pragma unroll 1
                for (int j = 0; j < n; ++j) { acc0 += s; asm volatile(""); }
            }
            if (MODE == 7) {   // straight line: 64 x v_fma_f64 independent pairs (VALU issue reference)
#pragma unroll
                for (int j = 0; j < 64; j += 2) { acc0 = fma(acc0, s, s); acc1 = fma(acc1, s, s); }
            }
        }
        const unsigned long long t1 = now();
        out[lane] = acc0 + acc1;
        if (lane == 0) ticks[0] = t1 - t0;
    }
    __syncthreads();
}
template <int MODE>
void run(const char* name, int ops) {
    double* out; unsigned long long* tk;
    hipMalloc(&out, 1024 * 8); hipMalloc(&tk, 8);
    const size_t lds = 2 * 64 * 66 * 8 + 4096;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k<MODE><<<1, 1024, lds>>>(out, tk, 64, 0.999);
    k<MODE><<<1, 1024, lds>>>(out, tk, 64, 0.999);
    hipDeviceSynchronize();
    unsigned long long t; hipMemcpy(&t, tk, 8, hipMemcpyDeviceToHost);
    printf("%-86s %8.1f cycles per pass of 64 rows, %5.1f per LDS instruction (%d)\n", name, t / 16.0, ops ? t / 16.0 / ops : 0.0, ops);
    hipFree(out); hipFree(tk);
}
int main() {
    run<0>("64 ds_read_b64 + fma, straight line", 64);
    run<1>("64 (ds_read_b64 + broadcast ds_read_b64 + fma), straight line", 128);
    run<2>("64 ds_write_b64, straight line", 64);
    run<3>("64 (read b64 + broadcast b128 + 2 fma + write b64), straight line", 192);
    run<4>("as row 2 but a rolled loop, 4 rows per trip", 128);
    run<5>("32 ds_read_b128 of the lane's own row (stride 66) + 2 fma", 32);
    run<6>("64 trips of an empty rolled loop (taken branch + v_add)", 0);
    run<7>("64 v_fma_f64, two chains (VALU reference)", 0);
    return 0;
}
