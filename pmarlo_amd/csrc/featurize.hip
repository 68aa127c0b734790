// Per-frame featurizers: pair distances, three-body angles, dihedrals (+ cos/sin).
//
// HBM-bound: read 12*A bytes of coordinates per frame, write 4 bytes per feature
// (SURVEY.md section 8d).  Threads are laid out feature-fastest so the output rows are
// written as contiguous runs; the (few hundred bytes of) coordinates of a frame are
// shared by all its features through L1.  Arithmetic is fp32 in the operation order
// of the reference's in-repo extractor (S/features/deeptica/ts_feature_extractor.py:
// 423-500, no PBC): d = sqrt(max(|v|^2, eps)); angle = acos(clamp(v1.v2/(|v1||v2|)));
// dihedral = atan2((c0 x c1).b1/|b1|, c0.c1) with c0 = b0 x b1, c1 = b1 x b2 normalised.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr float kEps = 1.0e-12f;

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ V3 scale(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }

__global__ __launch_bounds__(kThreads) void distances_kernel(const float* __restrict__ xyz, int64_t n, int A,
                                                            const int* __restrict__ pairs, int P,
                                                            float* __restrict__ out, int64_t ld, int col_off) {
    const int64_t total = n * P;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
        const int64_t t = e / P;
        const int f = (int)(e - t * P);
        const float* fr = xyz + t * A * 3;
        const V3 v = sub(ld3(fr + 3 * pairs[2 * f + 1]), ld3(fr + 3 * pairs[2 * f]));
        out[t * ld + col_off + f] = sqrtf(fmaxf(dot(v, v), kEps));
    }
}

__global__ __launch_bounds__(kThreads) void angles_kernel(const float* __restrict__ xyz, int64_t n, int A,
                                                         const int* __restrict__ trip, int Tn,
                                                         float* __restrict__ out, int64_t ld, int col_off) {
    const int64_t total = n * Tn;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
        const int64_t t = e / Tn;
        const int f = (int)(e - t * Tn);
        const float* fr = xyz + t * A * 3;
        const V3 pj = ld3(fr + 3 * trip[3 * f + 1]);
        const V3 v1 = sub(ld3(fr + 3 * trip[3 * f]), pj);
        const V3 v2 = sub(ld3(fr + 3 * trip[3 * f + 2]), pj);
        const float n1 = sqrtf(fmaxf(dot(v1, v1), kEps));
        const float n2 = sqrtf(fmaxf(dot(v2, v2), kEps));
        float c = dot(v1, v2) / (n1 * n2);
        c = fminf(fmaxf(c, -1.0f), 1.0f);
        out[t * ld + col_off + f] = acosf(c);
    }
}

// mode 0: angle in (-pi, pi];  mode 1: [cos, sin] adjacent per angle (api/features.py:138-180);
// mode 2: [cos block | sin block] (markov_state_model/_features.py:131-142)
__global__ __launch_bounds__(kThreads) void dihedrals_kernel(const float* __restrict__ xyz, int64_t n, int A,
                                                            const int* __restrict__ quads, int Q, int mode,
                                                            float* __restrict__ out, int64_t ld, int col_off) {
    const int64_t total = n * Q;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
        const int64_t t = e / Q;
        const int f = (int)(e - t * Q);
        const float* fr = xyz + t * A * 3;
        const V3 p0 = ld3(fr + 3 * quads[4 * f]), p1 = ld3(fr + 3 * quads[4 * f + 1]);
        const V3 p2 = ld3(fr + 3 * quads[4 * f + 2]), p3 = ld3(fr + 3 * quads[4 * f + 3]);
        const V3 b0 = sub(p1, p0), b1 = sub(p2, p1), b2 = sub(p3, p2);
        V3 c0 = cross(b0, b1), c1 = cross(b1, b2);
        const float n0 = sqrtf(fmaxf(dot(c0, c0), kEps));
        const float n1 = sqrtf(fmaxf(dot(c1, c1), kEps));
        const float nb = sqrtf(fmaxf(dot(b1, b1), kEps));
        c0 = scale(c0, n0);
        c1 = scale(c1, n1);
        const V3 b1u = scale(b1, nb);
        const float x = dot(c0, c1);
        const float y = dot(cross(c0, c1), b1u);
        const bool ok = (fabsf(x) + fabsf(y)) >= kEps;
        float ang = ok ? atan2f(y, x) : 0.0f;
        if (ang <= -3.14159265358979323846f) ang += 6.28318530717958647692f;  // builtins.py:11-14
        float* row = out + t * ld + col_off;
        if (mode == 0) row[f] = ang;
        else if (mode == 1) { row[2 * f] = cosf(ang); row[2 * f + 1] = sinf(ang); }
        else { row[f] = cosf(ang); row[Q + f] = sinf(ang); }
    }
}

// contact indicator of one atom pair per column: 1.0 when the distance is <= rcut, else 0.0; a NaN
// distance (missing coordinates) counts as "no contact" (S/features/builtins.py:252-275)
__global__ __launch_bounds__(kThreads) void contacts_kernel(const float* __restrict__ xyz, int64_t n, int A,
                                                           const int* __restrict__ pairs, int P, float rcut,
                                                           float* __restrict__ out, int64_t ld, int col_off) {
    const int64_t total = n * P;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
        const int64_t t = e / P;
        const int f = (int)(e - t * P);
        const float* fr = xyz + t * A * 3;
        const V3 v = sub(ld3(fr + 3 * pairs[2 * f + 1]), ld3(fr + 3 * pairs[2 * f]));
        const float dist = sqrtf(fmaxf(dot(v, v), kEps));
        out[t * ld + col_off + f] = dist <= rcut ? 1.0f : 0.0f;   // NaN compares false
    }
}

// radius of gyration with unit masses (mdtraj.compute_rg as called at S/features/builtins.py:98):
// sqrt(mean_a |r_a - mean_a r_a|^2); one wave per frame, fp64 accumulation of the fp32 coordinates
__global__ __launch_bounds__(kThreads) void rg_kernel(const float* __restrict__ xyz, int64_t n, int A,
                                                     float* __restrict__ out, int64_t ld, int col_off) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * kThreads + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * kThreads) >> 6;
    for (int64_t t = wave; t < n; t += n_waves) {
        const float* fr = xyz + t * A * 3;
        double sx = 0.0, sy = 0.0, sz = 0.0;
        for (int a = lane; a < A; a += 64) { sx += fr[3 * a]; sy += fr[3 * a + 1]; sz += fr[3 * a + 2]; }
        for (int off = 32; off > 0; off >>= 1) {
            sx += __shfl_xor(sx, off, 64); sy += __shfl_xor(sy, off, 64); sz += __shfl_xor(sz, off, 64);
        }
        const double mx = sx / A, my = sy / A, mz = sz / A;
        double q = 0.0;
        for (int a = lane; a < A; a += 64) {
            const double dx = fr[3 * a] - mx, dy = fr[3 * a + 1] - my, dz = fr[3 * a + 2] - mz;
            q += dx * dx + dy * dy + dz * dz;
        }
        for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
        if (lane == 0) out[t * ld + col_off] = (float)sqrt(q / A);
    }
}

int grid_for(const msm_ctx* ctx, int64_t total) {
    return (int)std::min<int64_t>(std::max<int64_t>(1, (total + kThreads - 1) / kThreads), (int64_t)ctx->n_cu * 16);
}

msm_status check_common(msm_ctx* ctx, const char* who, const float* xyz, int64_t n, int A, const int32_t* idx, int m,
                        float* out, int64_t ld, int col_off, int width) {
    MSM_REQUIRE(ctx, n >= 0 && A >= 1 && m >= 0, "%s: need n >= 0, A >= 1, count >= 0", who);
    MSM_REQUIRE(ctx, col_off >= 0 && ld >= col_off + (int64_t)width, "%s: output columns [%d, %d) exceed ld=%lld", who,
                col_off, col_off + width, (long long)ld);
    MSM_REQUIRE(ctx, (n == 0 || m == 0) || (xyz && idx && out), "%s: NULL pointer", who);
    return MSM_OK;
}

}  // namespace

extern "C" {

msm_status msm_featurize_distances(msm_ctx* ctx, const float* d_xyz, int64_t n, int A, const int32_t* d_pairs, int P,
                                   float* d_out, int64_t ld, int col_off) {
    if (!ctx) return MSM_ERR_INVALID;
    msm_status rs = check_common(ctx, "msm_featurize_distances", d_xyz, n, A, d_pairs, P, d_out, ld, col_off, P);
    if (rs != MSM_OK || n == 0 || P == 0) return rs;
    hipLaunchKernelGGL(distances_kernel, dim3(grid_for(ctx, n * P)), dim3(kThreads), 0, ctx->stream, d_xyz, n, A,
                       d_pairs, P, d_out, ld, col_off);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_featurize_contacts(msm_ctx* ctx, const float* d_xyz, int64_t n, int A, const int32_t* d_pairs, int P,
                                  float rcut, float* d_out, int64_t ld, int col_off) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, rcut > 0.0f, "msm_featurize_contacts: rcut must be positive");
    msm_status rs = check_common(ctx, "msm_featurize_contacts", d_xyz, n, A, d_pairs, P, d_out, ld, col_off, P);
    if (rs != MSM_OK || n == 0 || P == 0) return rs;
    hipLaunchKernelGGL(contacts_kernel, dim3(grid_for(ctx, n * P)), dim3(kThreads), 0, ctx->stream, d_xyz, n, A, d_pairs,
                       P, rcut, d_out, ld, col_off);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_featurize_rg(msm_ctx* ctx, const float* d_xyz, int64_t n, int A, float* d_out, int64_t ld, int col_off) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 0 && A >= 1 && col_off >= 0 && ld >= col_off + 1, "msm_featurize_rg: bad shape");
    if (n == 0) return MSM_OK;
    MSM_REQUIRE(ctx, d_xyz && d_out, "msm_featurize_rg: NULL pointer");
    hipLaunchKernelGGL(rg_kernel, dim3(grid_for(ctx, n * 64)), dim3(kThreads), 0, ctx->stream, d_xyz, n, A, d_out, ld,
                       col_off);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_featurize_angles(msm_ctx* ctx, const float* d_xyz, int64_t n, int A, const int32_t* d_triplets, int Tn,
                                float* d_out, int64_t ld, int col_off) {
    if (!ctx) return MSM_ERR_INVALID;
    msm_status rs = check_common(ctx, "msm_featurize_angles", d_xyz, n, A, d_triplets, Tn, d_out, ld, col_off, Tn);
    if (rs != MSM_OK || n == 0 || Tn == 0) return rs;
    hipLaunchKernelGGL(angles_kernel, dim3(grid_for(ctx, n * Tn)), dim3(kThreads), 0, ctx->stream, d_xyz, n, A,
                       d_triplets, Tn, d_out, ld, col_off);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_featurize_dihedrals(msm_ctx* ctx, const float* d_xyz, int64_t n, int A, const int32_t* d_quads, int Q,
                                   int mode, float* d_out, int64_t ld, int col_off) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, mode >= 0 && mode <= 2, "msm_featurize_dihedrals: mode must be 0, 1 or 2");
    msm_status rs = check_common(ctx, "msm_featurize_dihedrals", d_xyz, n, A, d_quads, Q, d_out, ld, col_off,
                                 mode == 0 ? Q : 2 * Q);
    if (rs != MSM_OK || n == 0 || Q == 0) return rs;
    hipLaunchKernelGGL(dihedrals_kernel, dim3(grid_for(ctx, n * Q)), dim3(kThreads), 0, ctx->stream, d_xyz, n, A,
                       d_quads, Q, mode, d_out, ld, col_off);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

}  // extern "C"
