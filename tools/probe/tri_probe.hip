// Diagnostic: phases of the tridiagonal eigensolver inside msm_eigh and of msm_tica_solve (thread 0's view, shader
// cycles).  Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/probe/tri_probe.hip -o ...
#define MSM_TRI_STAMPS 1
#include "../../pmarlo_amd/csrc/eig.hip"
#include "../../pmarlo_amd/csrc/ctx.hip"
#include <vector>
#include <cstdio>
#include <cmath>
static const char* names[24] = {"init Q", "Householder columns", "barrier", "extract d, e, bounds",
                                "multisection", "twisted factorisation vectors", "residual + orthogonality check", "Z = Q X",
                                "tica: covariances from moments", "tica: LDL' + inverse (or eigen path)", "tica: Ct = L' C0t L",
                                "tica: eigensolve (sum of the rows above)", "tica: sort, R = L Z, signs, output", "  ldl: pivots, reciprocals", "  ldl: updates", "  ldl: barrier",
                                "  hh: loop top", "  hh: barrier 2, rank-2 update + next reflector", "  hh: partial p = A v", "  hh: barrier 1", "  hh: sum partials, p.v, w", "  hh: row k+1 published", "", ""};
static void report(const char* what, float ms) {
    unsigned long long st[24];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_tri_stamps), sizeof(st));
    printf("%s: %.3f ms\n", what, ms);
    unsigned long long tot = 0;
    for (int i = 0; i < 8; ++i) tot += st[i];
    unsigned long long ktot = 0;
    for (int i = 8; i < 16; ++i) ktot += st[i];
    for (int i = 0; i < 24; ++i)
        if (st[i]) printf("  %-44s %10llu cycles\n", names[i], st[i]);
    printf("  solver total %llu cycles; kernel total %llu cycles -> %.2f GHz\n", tot, ktot, (ktot ? ktot : tot) / (ms * 1e6));
}
int main() {
    const int n = 64;
    msm_ctx* ctx; if (msm_ctx_create(0, nullptr, &ctx)) return 1;
    std::vector<double> A(n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i * n + j] = (i == j ? 2.0 : 0.0) + sin(0.37 * (i + 1) * (j + 1)) * 0.1 + sin(0.37 * (j + 1) * (i + 1)) * 0.1;
    double *dA, *dw, *dv; int* ds;
    hipMalloc(&dA, n * n * 8); hipMalloc(&dw, n * 8); hipMalloc(&dv, n * n * 8); hipMalloc(&ds, 4);
    hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
    unsigned long long z[24] = {0};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpyToSymbol(HIP_SYMBOL(g_tri_stamps), z, sizeof(z));
        hipEventRecord(e0, ctx->stream);
        msm_eigh(ctx, dA, n, dw, dv, ds);
        hipEventRecord(e1, ctx->stream); msm_sync(ctx); hipEventElapsedTime(&ms, e0, e1);
    }
    int sweeps;
    hipMemcpy(&sweeps, ds, 4, hipMemcpyDeviceToHost);
    printf("sweeps %d (0 = tridiagonal path)\n", sweeps);
    report("msm_eigh n=64", ms);
    // residual check on the host
    {
        std::vector<double> w(n), V(n * n);
        hipMemcpy(w.data(), dw, n * 8, hipMemcpyDeviceToHost);
        hipMemcpy(V.data(), dv, n * n * 8, hipMemcpyDeviceToHost);
        double worst = 0.0, orth = 0.0;
        for (int c = 0; c < n; ++c)
            for (int i = 0; i < n; ++i) {
                double r = -w[c] * V[i * n + c];
                for (int k = 0; k < n; ++k) r += A[i * n + k] * V[k * n + c];
                worst = fmax(worst, fabs(r));
            }
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < n; ++b) {
                double g = a == b ? -1.0 : 0.0;
                for (int k = 0; k < n; ++k) g += V[k * n + a] * V[k * n + b];
                orth = fmax(orth, fabs(g));
            }
        printf("  |A v - w v|_max %.3e   |V'V - I|_max %.3e   w[0] %.15g w[n-1] %.15g\n", worst, orth, w[0], w[n - 1]);
    }
    // TICA solve on synthetic moments: X = AR(1)-like covariance, C0t = 0.5 decay
    {
        const int F = 64;
        std::vector<double> mom(2 * F * F + 2 * F + 1, 0.0);
        const double T = 1000.0;
        for (int i = 0; i < F; ++i)
            for (int j = 0; j < F; ++j) {
                const double c00 = exp(-0.3 * abs(i - j)) + (i == j ? 0.5 : 0.0);
                const double c0t = 0.6 * exp(-0.35 * abs(i - j)) * cos(0.05 * (i + j));
                mom[i * F + j] = 2.0 * T * c00;
                mom[F * F + i * F + j] = T * c0t;
            }
        mom[2 * F * F + 2 * F] = T;
        double *dm, *de, *dW, *dmean; int* dr;
        hipMalloc(&dm, mom.size() * 8); hipMalloc(&de, F * 8); hipMalloc(&dW, F * F * 8); hipMalloc(&dmean, F * 8); hipMalloc(&dr, 4);
        hipMemcpy(dm, mom.data(), mom.size() * 8, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 3; ++rep) {
            hipMemcpyToSymbol(HIP_SYMBOL(g_tri_stamps), z, sizeof(z));
            hipEventRecord(e0, ctx->stream);
            msm_tica_solve(ctx, dm, nullptr, F, 1e-6, 1, de, dW, dmean, dr);
            hipEventRecord(e1, ctx->stream); msm_sync(ctx); hipEventElapsedTime(&ms, e0, e1);
        }
        int rank; hipMemcpy(&rank, dr, 4, hipMemcpyDeviceToHost);
        std::vector<double> ev(F); hipMemcpy(ev.data(), de, F * 8, hipMemcpyDeviceToHost);
        printf("rank %d  eig[0] %.15g eig[1] %.15g\n", rank, ev[0], ev[1]);
        report("msm_tica_solve F=64", ms);
    }
    return 0;
}
