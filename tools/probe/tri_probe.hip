// Diagnostic: phases of the tridiagonal eigensolver inside msm_eigh (thread 0's view, shader cycles)
#define MSM_TRI_STAMPS 1
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
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpyToSymbol(HIP_SYMBOL(g_tri_stamps), z, sizeof(z));
        hipEventRecord(e0, ctx->stream);
        msm_eigh(ctx, dA, n, dw, dv, ds);
        hipEventRecord(e1, ctx->stream); msm_sync(ctx); hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long st[8]; int sweeps;
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_tri_stamps), sizeof(st));
    hipMemcpy(&sweeps, ds, 4, hipMemcpyDeviceToHost);
    const char* names[] = {"init Q", "reflector (wave 0) + barrier", "A v, Q v + barrier", "rank-2 / rank-1 updates + barrier", "extract d, e, bounds", "multisection (15 rounds)", "twisted factorisation vectors", "orthogonality check + Z = Q X"};
    printf("msm_eigh n=%d: %.3f ms, sweeps %d (0 = tridiagonal path)\n", n, ms, sweeps);
    unsigned long long tot = 0; for (int i = 0; i < 8; ++i) tot += st[i];
    for (int i = 0; i < 8; ++i) printf("%-38s %10llu cycles  %5.1f%%\n", names[i], st[i], 100.0 * st[i] / tot);
    printf("total %llu cycles -> %.2f GHz\n", tot, tot / (ms * 1e6));
    return 0;
}
