// Diagnostic: where does the bf16-filter k-means kernel spend its cycles?  (wave 0 of every block; the stamp
// build also drains vmcnt before the stamps that follow loads, so load latency lands in its own bucket)
#define MSM_KM_STAMPS 1
#include "../../pmarlo_amd/csrc/kmeans.hip"
#include "../../pmarlo_amd/csrc/ctx.hip"
#include <vector>
#include <cstdio>
#include <cmath>
#include <cstdlib>
int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000; const int d = argc > 2 ? atoi(argv[2]) : 10, k = argc > 3 ? atoi(argv[3]) : 500;
    msm_ctx* ctx; if (msm_ctx_create(0, nullptr, &ctx)) return 1;
    std::vector<double> Y(n * d);
    unsigned long long sd = 12345;
    // two slow coordinates with a wide spread, the rest narrow: the shape of TICA output
    for (int64_t t = 0; t < n; ++t)
        for (int f = 0; f < d; ++f) {
            sd = sd * 6364136223846793005ull + 1442695040888963407ull;
            const double u = (sd >> 11) * (1.0 / 9007199254740992.0) - 0.5;
            Y[t * d + f] = u * (f < 2 ? 4.0 : 0.05);
        }
    double *dY, *dC, *dS; int64_t *dsum, *dcnt; void* img;
    size_t ib = 0; msm_kmeans_image_bytes(n, d, &ib);
    hipMalloc(&dY, n * d * 8); hipMalloc(&dC, k * d * 8); hipMalloc(&dS, 64); hipMalloc(&dsum, k * d * 8); hipMalloc(&dcnt, k * 8); hipMalloc(&img, ib);
    hipMemcpy(dY, Y.data(), n * d * 8, hipMemcpyHostToDevice);
    if (msm_kmeans_fit_begin(ctx, dY, MSM_F64, n, d, d, nullptr, nullptr, k, 7, 1, (double)n, 0.0, dC, dS, 0)) { printf("%s\n", msm_last_error(ctx)); return 1; }
    if (msm_kmeans_pack(ctx, dY, MSM_F64, n, d, d, nullptr, nullptr, img)) { printf("%s\n", msm_last_error(ctx)); return 1; }
    hipMemset(dsum, 0, k * d * 8); hipMemset(dcnt, 0, k * 8);
    unsigned long long z[8] = {0};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpyToSymbol(HIP_SYMBOL(g_km_stamps), z, sizeof(z));
        hipEventRecord(e0, ctx->stream);
        if (msm_kmeans_accumulate_packed(ctx, dY, MSM_F64, n, d, d, dC, k, nullptr, nullptr, img, dS, dsum, dcnt)) { printf("%s\n", msm_last_error(ctx)); return 1; }
        hipEventRecord(e1, ctx->stream);
        msm_sync(ctx);
        hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long st[8];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_km_stamps), sizeof(st));
    uint64_t scanned = 0; msm_kmeans_filter_scanned(ctx, &scanned, 0);
    const char* names[] = {"centre tables built into the LDS", "image loads (drained)", "tile loop", "next loads out, cross-lane: holder, pair, R", "coordinates in, candidate pick", "commit (LDS atomics, labels)", "step 4 (fp32 scan of all centres, band in fp64)", "exit + flush"};
    unsigned long long tot = 0; for (int i = 0; i < 8; ++i) tot += st[i];
    printf("n=%lld d=%d k=%d kernel %.3f ms (stamp build); scanned %llu frames in 3 passes; stamps are sums over the blocks (wave 0)\n", (long long)n, d, k, ms, (unsigned long long)scanned);
    for (int i = 0; i < 8; ++i) printf("%-28s %12llu ticks  %5.1f%%  per block %.0f\n", names[i], st[i], 100.0 * st[i] / tot, st[i] / 256.0);
    printf("total per block %.0f ticks -> %.3f ms kernel => tick rate %.1f MHz\n", tot / 256.0, ms, tot / 256.0 / ms / 1e3);
    return 0;
}
