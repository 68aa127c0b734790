// Diagnostic: phases of the single-workgroup orthogonalisation step of msm_spectrum
#define MSM_SPEC_STAMPS 1
#include "../../pmarlo_amd/csrc/msm.hip"
#include "../../pmarlo_amd/csrc/ctx.hip"
#include <vector>
#include <cstdio>
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 500, p = argc > 2 ? atoi(argv[2]) : 12;
    msm_ctx* ctx; if (msm_ctx_create(0, nullptr, &ctx)) return 1;
    std::vector<double> T((size_t)n * n);
    unsigned long long sd = 1;
    for (int i = 0; i < n; ++i) { double rs = 0; for (int j = 0; j < n; ++j) { sd = sd * 6364136223846793005ull + 1442695040888963407ull; double v = (sd >> 11) * (1.0 / 9007199254740992.0) + (i / (n / 5 + 1) == j / (n / 5 + 1) ? 20.0 : 0.0); T[(size_t)i * n + j] = v; rs += v; } for (int j = 0; j < n; ++j) T[(size_t)i * n + j] /= rs; }
    double *dT, *ritz, *pi, *chg, *ie, *it; int* st; void* ws;
    hipMalloc(&dT, T.size() * 8); hipMemcpy(dT, T.data(), T.size() * 8, hipMemcpyHostToDevice);
    hipMalloc(&ws, msm_spectrum_workspace_bytes(n, p, 1)); hipMalloc(&ritz, 128 * 8); hipMalloc(&pi, n * 8); hipMalloc(&chg, 8); hipMalloc(&st, 4); hipMalloc(&ie, 64); hipMalloc(&it, 64);
    unsigned long long z[8] = {0};
    const int iters = 40;
    for (int rep = 0; rep < 2; ++rep) {
        hipMemcpyToSymbol(HIP_SYMBOL(g_spec_stamps), z, sizeof(z));
        if (msm_spectrum(ctx, dT, (int64_t)n * n, n, nullptr, n, 1, p, iters, 1, 0, p < 7 ? p : 6, ws, ritz, pi, n, chg, st, 5, nullptr, ie, it, 0.0, nullptr, 0)) { printf("%s\n", msm_last_error(ctx)); return 1; }
        msm_sync(ctx);
    }
    unsigned long long s[8]; hipMemcpyFromSymbol(s, HIP_SYMBOL(g_spec_stamps), sizeof(s));
    const char* nm[] = {"sum partials", "gram W'W", "cholesky", "R^-1 + Z = W R^-1"};
    for (int i = 0; i < 4; ++i) printf("%-20s %10.0f ticks (10 ns) per step\n", nm[i], s[i] / (double)(iters + 1));
    double c; hipMemcpy(&c, chg, 8, hipMemcpyDeviceToHost); printf("residual %g\n", c);
    return 0;
}
