// Diagnostic: where does a Jacobi round spend its cycles?  (thread 0's view)
#define MSM_JACOBI_STAMPS 1
#include "../../pmarlo_amd/csrc/eig.hip"
#include "../../pmarlo_amd/csrc/ctx.hip"
#include <vector>
#include <cstdio>
#include <cmath>
int main() {
    const int n = 64;
    msm_ctx* ctx; if (msm_ctx_create(0, nullptr, &ctx)) return 1;
    std::vector<double> A(n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i * n + j] = (i == j ? 2.0 : 0.0) + sin(0.37 * (i + 1) * (j + 1)) * 0.1 + sin(0.37 * (j + 1) * (i + 1)) * 0.1;
    double *dA, *dw, *dv; int* ds;
    hipMalloc(&dA, n * n * 8); hipMalloc(&dw, n * 8); hipMalloc(&dv, n * n * 8); hipMalloc(&ds, 4);
    hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
    unsigned long long z[8] = {0};
    for (int rep = 0; rep < 2; ++rep) {
        hipMemcpyToSymbol(HIP_SYMBOL(g_jacobi_stamps), z, sizeof(z));
        msm_eigh(ctx, dA, n, dw, dv, ds); msm_sync(ctx);
    }
    unsigned long long st[8]; int sweeps;
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_jacobi_stamps), sizeof(st));
    hipMemcpy(&sweeps, ds, 4, hipMemcpyDeviceToHost);
    const double rounds = sweeps * (n - 1.0);
    printf("sweeps %d rounds %.0f\n", sweeps, rounds);
    const char* names[] = {"loop-top(prev barrier tail)", "conv check (per sweep)", "phaseA rotations", "barrier1", "phaseB 2x2 blocks", "phaseB V", "barrier2"};
    for (int i = 0; i < 7; ++i) printf("%-30s total %10llu cycles   per round %8.1f\n", names[i], st[i], st[i] / rounds);
    return 0;
}
