// Structure features whose reference implementation is an mdtraj algorithm (S/features/builtins.py:171-250: "sasa",
// "hbonds_count", "ssfrac"): Shrake-Rupley solvent accessible surface, Baker-Hubbard hydrogen-bond presence and the
// Kabsch-Sander secondary structure assignment (DSSP), each one frame at a time, frames side by side.
// mdtraj itself is absent from the build container: the kernels restate its published algorithms (mdtraj 1.10:
// geometry/src/sasa.cpp, geometry/hbond.py, geometry/src/dssp.cpp) in their operation order; the oracle
// (oracle/npport.py) restates them once more in numpy.
#include "common.h"

namespace {

constexpr int kSThreads = 256;
constexpr int kSWaves = kSThreads / 64;
constexpr int kMaxNbr = 384;   // neighbours of one atom within R_i + R_j (probe included); ~90 in a folded protein

struct F3 { float x, y, z; };
__device__ __forceinline__ F3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ float dot3(F3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }   // (x x + y y) + z z

// ---------------------------------------------------------------------------------------------------------
// Shrake-Rupley.  One wave per (frame, atom): neighbours j with |r_i - r_j|^2 < (R_i + R_j)^2 are compacted into the
// wave's LDS list (position and R_j^2), then the lanes share the sphere points: point k sits at r_i + R_i p_k and
// counts unless it lies inside a neighbour's sphere.  area_i = ((4 pi / P) R_i) R_i n_accessible, all in fp32 as in
// mdtraj's sasa.cpp (whose starting the scan at the neighbour that blocked the previous point only saves time).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kSThreads) void sasa_kernel(const float* __restrict__ xyz, int64_t n, int A,
                                                        const float* __restrict__ radii,
                                                        const float* __restrict__ points, int P,
                                                        float* __restrict__ out, int* __restrict__ overflow) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sasa_smem[];
    float4* nbr = reinterpret_cast<float4*>(sasa_smem);                 // [kSWaves][kMaxNbr]
    float* pts = reinterpret_cast<float*>(nbr + kSWaves * kMaxNbr);     // [P][3]
    for (int i = threadIdx.x; i < 3 * P; i += kSThreads) pts[i] = points[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4* mine = nbr + wave * kMaxNbr;
    const float constant = (float)(4.0 * 3.14159265358979323846 / (double)P);
    const int64_t items = n * (int64_t)A;
    const int64_t wid = (int64_t)blockIdx.x * kSWaves + wave, nw = (int64_t)gridDim.x * kSWaves;
    for (int64_t it = wid; it < items; it += nw) {
        const int64_t t = it / A;
        const int i = (int)(it - t * A);
        const float* fr = xyz + t * (int64_t)A * 3;
        const F3 ri = ld3(fr + 3 * i);
        const float Ri = radii[i];
        int cnt = 0;
        for (int j0 = 0; j0 < A; j0 += 64) {
            const int j = j0 + lane;
            bool hit = false;
            F3 rj = {0.f, 0.f, 0.f};
            float Rj = 0.f;
            if (j < A && j != i) {
                rj = ld3(fr + 3 * j);
                Rj = radii[j];
                const F3 d = {ri.x - rj.x, ri.y - rj.y, ri.z - rj.z};
                const float cut = Ri + Rj;
                hit = dot3(d) < cut * cut;
            }
            const unsigned long long m = __ballot(hit);
            if (hit) {
                const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
                if (pos < kMaxNbr) mine[pos] = make_float4(rj.x, rj.y, rj.z, Rj * Rj);
            }
            cnt += __popcll(m);
        }
        if (cnt > kMaxNbr) {
            if (lane == 0) atomicExch(overflow, 1);
            cnt = kMaxNbr;
        }
        int acc = 0;
        for (int k = lane; k < P; k += 64) {
            const F3 c = {ri.x + Ri * pts[3 * k], ri.y + Ri * pts[3 * k + 1], ri.z + Ri * pts[3 * k + 2]};
            bool open = true;
            for (int q = 0; q < cnt; ++q) {
                const float4 nb = mine[q];
                const F3 d = {c.x - nb.x, c.y - nb.y, c.z - nb.z};
                if (dot3(d) < nb.w) { open = false; break; }
            }
            acc += open ? 1 : 0;
        }
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) out[it] = constant * Ri * Ri * (float)acc;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Baker-Hubbard presence (mdtraj/geometry/hbond.py baker_hubbard + _compute_bounded_geometry): for every
// (donor, hydrogen, acceptor) triplet the number of frames with |H - A| < dcut and angle(D, H, A) > acut, the
// angle by the law of cosines on the three fp32 distances as mdtraj forms it.  Counting per triplet and frame
// chunk, one atomic per (chunk, triplet).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dist3(const float* a, const float* b) {
    const F3 d = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    return sqrtf(dot3(d));
}
__global__ __launch_bounds__(kSThreads) void hbond_presence_kernel(const float* __restrict__ xyz, int64_t n, int A,
                                                                  const int32_t* __restrict__ trip, int Tn, float dcut,
                                                                  float acut, int frames_per_block,
                                                                  unsigned long long* __restrict__ counts) {
    const int64_t t0 = (int64_t)blockIdx.y * frames_per_block;
    const int64_t t1 = min(n, t0 + frames_per_block);
    for (int q = blockIdx.x * kSThreads + threadIdx.x; q < Tn; q += gridDim.x * kSThreads) {
        const int d = trip[3 * q], h = trip[3 * q + 1], a = trip[3 * q + 2];
        unsigned c = 0;
        for (int64_t t = t0; t < t1; ++t) {
            const float* fr = xyz + t * (int64_t)A * 3;
            const float b = dist3(fr + 3 * h, fr + 3 * a);           // H ... A
            if (!(b < dcut)) continue;
            const float aa = dist3(fr + 3 * d, fr + 3 * h);          // D - H
            const float cc = dist3(fr + 3 * a, fr + 3 * d);          // A ... D
            float cosv = (aa * aa + b * b - cc * cc) / (2.0f * aa * b);
            cosv = fminf(fmaxf(cosv, -1.0f), 1.0f);
            if (acosf(cosv) > acut) ++c;
        }
        if (c) atomicAdd(&counts[q], (unsigned long long)c);
    }
}

// ---------------------------------------------------------------------------------------------------------
// DSSP (Kabsch & Sander 1983 as implemented by mdtraj/geometry/src/dssp.cpp, a port of DSSP 2.0).  One LANE per
// frame runs the whole assignment for its frame; per-frame work arrays live in global scratch, laid out
// [array][residue][frame] so that the lanes of a wave touch neighbouring addresses.
//   backbone int32 [R][4]: atom indices of N, CA, C, O (-1: residue without a full backbone: never bonded, 'coil')
//   chain    int32 [R],  proline uint8 [R]
//   codes    uint8 [n][R]: 0 loop, 1 alpha helix H, 2 bridge B, 3 strand E, 4 3-10 helix G, 5 pi helix I, 6 turn T, 7 bend S
// ---------------------------------------------------------------------------------------------------------
enum : unsigned char { SS_LOOP = 0, SS_H = 1, SS_B = 2, SS_E = 3, SS_G = 4, SS_I = 5, SS_T = 6, SS_S = 7 };
enum : unsigned char { HX_NONE = 0, HX_START = 1, HX_END = 2, HX_BOTH = 3, HX_MID = 4 };

struct DsspWork {
    // all indexed [r * n + t] (t = frame)
    int* acc0; int* acc1;          // the two best acceptors of residue r's N-H (-1: none)
    float* e0; float* e1;          // their energies
    unsigned char* hflag;          // [3][R][n]: helix flags of stride 3, 4, 5
    unsigned char* bend;           // [R][n]
    unsigned char* ss;             // [R][n]
    int* lad;                      // ladders: [6][R][n]: i_first, i_last, j_first, j_last, type, alive
    float* hpos;                   // [3][R][n]: amide hydrogen positions (Angstrom)
};

__device__ __forceinline__ bool dssp_bond(const DsspWork& w, int64_t n, int64_t t, int donor, int acceptor) {
    // TestBond(donor, acceptor): the donor's N-H ... O of `acceptor`, one of its two best with E < -0.5
    const int64_t o = (int64_t)donor * n + t;
    return (w.acc0[o] == acceptor && w.e0[o] < -0.5f) || (w.acc1[o] == acceptor && w.e1[o] < -0.5f);
}

__global__ __launch_bounds__(64) void dssp_kernel(const float* __restrict__ xyz, int64_t n, int A,
                                                 const int32_t* __restrict__ bb, const int32_t* __restrict__ chain,
                                                 const unsigned char* __restrict__ proline, int R, DsspWork w,
                                                 unsigned char* __restrict__ codes) {
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= n) return;
    const float* fr = xyz + t * (int64_t)A * 3;
    auto at = [&](int r, int which) -> F3 {   // Angstrom, as DSSP works
        const int a = bb[4 * r + which];
        return {fr[3 * a] * 10.0f, fr[3 * a + 1] * 10.0f, fr[3 * a + 2] * 10.0f};
    };
    auto full = [&](int r) { return bb[4 * r] >= 0 && bb[4 * r + 1] >= 0 && bb[4 * r + 2] >= 0 && bb[4 * r + 3] >= 0; };
    auto dist = [&](F3 a, F3 b) { const F3 d = {a.x - b.x, a.y - b.y, a.z - b.z}; return sqrtf(dot3(d)); };
    // chain break between residues a <= b: different chains, a residue without backbone, or a C(i) - N(i+1) gap > 2.5 A
    auto no_break = [&](int a, int b) {
        for (int r = a; r < b; ++r) {
            if (chain[r] != chain[r + 1] || !full(r) || !full(r + 1)) return false;
            if (dist(at(r, 2), at(r + 1, 0)) > 2.5f) return false;
        }
        return full(a) && full(b);
    };
    auto IX = [&](int r) { return (int64_t)r * n + t; };
    // ---- hydrogen positions and H-bond energies: two best acceptors per donor
    for (int r = 0; r < R; ++r) { w.acc0[IX(r)] = -1; w.acc1[IX(r)] = -1; w.e0[IX(r)] = 0.0f; w.e1[IX(r)] = 0.0f; }
    for (int r = 0; r < R; ++r) {
        if (!full(r)) continue;
        F3 N = at(r, 0);
        if (r > 0 && !proline[r] && full(r - 1) && chain[r - 1] == chain[r]) {
            const F3 pc = at(r - 1, 2), po = at(r - 1, 3);
            const float len = dist(pc, po);
            N.x += (pc.x - po.x) / len; N.y += (pc.y - po.y) / len; N.z += (pc.z - po.z) / len;
        }
        w.hpos[((int64_t)0 * R + r) * n + t] = N.x;
        w.hpos[((int64_t)1 * R + r) * n + t] = N.y;
        w.hpos[((int64_t)2 * R + r) * n + t] = N.z;
    }
    auto hpos = [&](int r) -> F3 {
        return {w.hpos[((int64_t)0 * R + r) * n + t], w.hpos[((int64_t)1 * R + r) * n + t], w.hpos[((int64_t)2 * R + r) * n + t]};
    };
    auto energy = [&](int donor, int acceptor) {
        float res = 0.0f;
        if (!proline[donor]) {
            const F3 H = hpos(donor), N = at(donor, 0), O = at(acceptor, 3), Cc = at(acceptor, 2);
            const float dHO = dist(H, O), dHC = dist(H, Cc), dNC = dist(N, Cc), dNO = dist(N, O);
            if (dHO < 0.5f || dHC < 0.5f || dNC < 0.5f || dNO < 0.5f) res = -9.9f;
            else res = -27.888f / dHO + 27.888f / dHC - 27.888f / dNC + 27.888f / dNO;
            res = roundf(res * 1000.0f) / 1000.0f;   // DSSP compatibility mode
            if (res < -9.9f) res = -9.9f;
        }
        const int64_t o = IX(donor);
        if (res < w.e0[o]) { w.acc1[o] = w.acc0[o]; w.e1[o] = w.e0[o]; w.acc0[o] = acceptor; w.e0[o] = res; }
        else if (res < w.e1[o]) { w.acc1[o] = acceptor; w.e1[o] = res; }
    };
    for (int i = 0; i + 1 < R; ++i) {
        if (!full(i)) continue;
        const F3 cai = at(i, 1);
        for (int j = i + 1; j < R; ++j) {
            if (!full(j)) continue;
            if (dist(cai, at(j, 1)) < 9.0f) {
                energy(i, j);
                if (j != i + 1) energy(j, i);
            }
        }
    }
    // ---- beta bridges -> ladders (sheets need not be numbered for the codes)
    for (int r = 0; r < R; ++r) w.ss[IX(r)] = SS_LOOP;
    int n_lad = 0;
    auto L = [&](int field, int l) -> int& { return w.lad[((int64_t)field * R + l) * n + t]; };
    for (int i = 1; i + 4 < R; ++i) {
        for (int j = i + 3; j + 1 < R; ++j) {
            // TestBridge(i, j): a = i-1, b = i, c = i+1; d = j-1, e = j, f = j+1
            int type = 0;   // 1 parallel, 2 antiparallel
            if (no_break(i - 1, i + 1) && no_break(j - 1, j + 1)) {
                if ((dssp_bond(w, n, t, i + 1, j) && dssp_bond(w, n, t, j, i - 1)) ||
                    (dssp_bond(w, n, t, j + 1, i) && dssp_bond(w, n, t, i, j - 1)))
                    type = 1;
                else if ((dssp_bond(w, n, t, i + 1, j - 1) && dssp_bond(w, n, t, j + 1, i - 1)) ||
                         (dssp_bond(w, n, t, j, i) && dssp_bond(w, n, t, i, j)))
                    type = 2;
            }
            if (!type) continue;
            bool found = false;
            for (int l = 0; l < n_lad && !found; ++l) {
                if (L(4, l) != type || i != L(1, l) + 1) continue;
                if (type == 1 && L(3, l) + 1 == j) { L(1, l) = i; L(3, l) = j; found = true; }
                else if (type == 2 && L(2, l) - 1 == j) { L(1, l) = i; L(2, l) = j; found = true; }
            }
            if (!found && n_lad < R) {
                L(0, n_lad) = i; L(1, n_lad) = i; L(2, n_lad) = j; L(3, n_lad) = j; L(4, n_lad) = type; L(5, n_lad) = 1;
                ++n_lad;
            }
        }
    }
    // sort by (i_first, j_first ...): the construction above already yields ascending i_first; equal starts keep
    // their order of creation (ascending j), which is DSSP's operator< on the first elements
    // ---- join ladders across bulges
    for (int a = 0; a < n_lad; ++a) {
        if (!L(5, a)) continue;
        for (int b = a + 1; b < n_lad; ++b) {
            if (!L(5, b)) continue;
            const int ibi = L(0, a), iei = L(1, a), jbi = L(2, a), jei = L(3, a);
            const int ibj = L(0, b), iej = L(1, b), jbj = L(2, b), jej = L(3, b);
            if (L(4, a) != L(4, b) || !no_break(min(ibi, ibj), max(iei, iej)) || !no_break(min(jbi, jbj), max(jei, jej)) ||
                ibj - iei >= 6 || (iei >= ibj && ibi <= iej))
                continue;
            bool bulge;
            if (L(4, a) == 1) bulge = (jbj - jei < 6 && ibj - iei < 3) || (jbj - jei < 3);
            else bulge = (jbi - jej < 6 && ibj - iei < 3) || (jbi - jej < 3);
            if (bulge) {
                L(1, a) = max(iei, iej); L(0, a) = min(ibi, ibj);
                L(2, a) = min(jbi, jbj); L(3, a) = max(jei, jej);
                L(5, b) = 0;
            }
        }
    }
    for (int l = 0; l < n_lad; ++l) {
        if (!L(5, l)) continue;
        const unsigned char code = (L(1, l) - L(0, l) >= 1 || L(3, l) - L(2, l) >= 1) ? SS_E : SS_B;
        for (int r = L(0, l); r <= L(1, l); ++r) if (w.ss[IX(r)] != SS_E) w.ss[IX(r)] = code;
        for (int r = L(2, l); r <= L(3, l); ++r) if (w.ss[IX(r)] != SS_E) w.ss[IX(r)] = code;
    }
    // ---- turns, bends, helices
    for (int s = 0; s < 3; ++s)
        for (int r = 0; r < R; ++r) w.hflag[((int64_t)s * R + r) * n + t] = HX_NONE;
    auto HF = [&](int s, int r) -> unsigned char& { return w.hflag[((int64_t)s * R + r) * n + t]; };
    for (int s = 0; s < 3; ++s) {
        const int stride = s + 3;
        for (int i = 0; i + stride < R; ++i) {
            if (no_break(i, i + stride) && dssp_bond(w, n, t, i + stride, i)) {
                HF(s, i + stride) = HF(s, i + stride) == HX_START ? HX_BOTH : HX_END;
                for (int j = i + 1; j < i + stride; ++j)
                    if (HF(s, j) == HX_NONE) HF(s, j) = HX_MID;
                HF(s, i) = HF(s, i) == HX_END ? HX_BOTH : HX_START;
            }
        }
    }
    for (int r = 0; r < R; ++r) w.bend[IX(r)] = 0;
    for (int i = 2; i + 2 < R; ++i) {
        if (!no_break(i - 2, i + 2)) continue;
        const F3 a = at(i - 2, 1), b = at(i, 1), c = at(i + 2, 1);
        const F3 u = {b.x - a.x, b.y - a.y, b.z - a.z}, v = {c.x - b.x, c.y - b.y, c.z - b.z};
        const float ck = (u.x * v.x + u.y * v.y + u.z * v.z) / (sqrtf(dot3(u)) * sqrtf(dot3(v)));
        const float kappa = acosf(fminf(fmaxf(ck, -1.0f), 1.0f)) * 57.29577951308232f;
        w.bend[IX(i)] = kappa > 70.0f ? 1 : 0;
    }
    auto is_start = [&](int r, int s) { const unsigned char f = HF(s, r); return f == HX_START || f == HX_BOTH; };
    for (int i = 1; i + 4 < R; ++i)
        if (is_start(i, 1) && is_start(i - 1, 1))
            for (int j = i; j <= i + 3; ++j) w.ss[IX(j)] = SS_H;
    for (int i = 1; i + 3 < R; ++i) {
        if (is_start(i, 0) && is_start(i - 1, 0)) {
            bool empty = true;
            for (int j = i; empty && j <= i + 2; ++j) empty = w.ss[IX(j)] == SS_LOOP || w.ss[IX(j)] == SS_G;
            if (empty) for (int j = i; j <= i + 2; ++j) w.ss[IX(j)] = SS_G;
        }
    }
    for (int i = 1; i + 5 < R; ++i) {
        if (is_start(i, 2) && is_start(i - 1, 2)) {
            bool empty = true;
            for (int j = i; empty && j <= i + 4; ++j) empty = w.ss[IX(j)] == SS_LOOP || w.ss[IX(j)] == SS_I;
            if (empty) for (int j = i; j <= i + 4; ++j) w.ss[IX(j)] = SS_I;
        }
    }
    for (int i = 1; i + 1 < R; ++i) {
        if (w.ss[IX(i)] != SS_LOOP) continue;
        bool turn = false;
        for (int s = 0; s < 3 && !turn; ++s)
            for (int k = 1; k < s + 3 && !turn; ++k) turn = i >= k && is_start(i - k, s);
        if (turn) w.ss[IX(i)] = SS_T;
        else if (w.bend[IX(i)]) w.ss[IX(i)] = SS_S;
    }
    for (int r = 0; r < R; ++r) codes[t * (int64_t)R + r] = w.ss[IX(r)];
}

}  // namespace

extern "C" {

msm_status msm_featurize_sasa(msm_ctx* ctx, const float* d_xyz, int64_t n, int A, const float* d_radii,
                              const float* d_points, int P, float* d_out) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 0 && A >= 1 && P >= 1 && P <= 4096, "msm_featurize_sasa: need n >= 0, A >= 1, 1 <= points <= 4096");
    if (n == 0) return MSM_OK;
    MSM_REQUIRE(ctx, d_xyz && d_radii && d_points && d_out, "msm_featurize_sasa: NULL pointer");
    msm_status rs = msm_reserve_aux(ctx, 64);
    if (rs != MSM_OK) return rs;
    int* flag = (int*)ctx->aux;
    MSM_HIP(ctx, hipMemsetAsync(flag, 0, sizeof(int), ctx->stream));
    const size_t lds = (size_t)kSWaves * kMaxNbr * sizeof(float4) + (size_t)3 * P * sizeof(float);
    const int64_t items = n * (int64_t)A;
    const int grid = (int)std::min<int64_t>((items + kSWaves - 1) / kSWaves, (int64_t)ctx->n_cu * 8);
    if (lds > 48 * 1024)
        MSM_HIP(ctx, hipFuncSetAttribute((const void*)sasa_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(sasa_kernel, dim3(grid), dim3(kSThreads), lds, ctx->stream, d_xyz, n, A, d_radii, d_points, P, d_out,
                       flag);
    MSM_CHECK_LAUNCH(ctx);
    int h_flag = 0;
    MSM_HIP(ctx, hipMemcpyAsync(&h_flag, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MSM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h_flag)
        return msm_fail(ctx, MSM_ERR_UNSUPPORTED, "msm_featurize_sasa: an atom has more than %d neighbours within R_i + R_j",
                        kMaxNbr);
    return MSM_OK;
}

msm_status msm_hbond_presence(msm_ctx* ctx, const float* d_xyz, int64_t n, int A, const int32_t* d_triplets, int Tn,
                              float dist_cutoff, float angle_cutoff, uint64_t* d_counts) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 0 && A >= 1 && Tn >= 0, "msm_hbond_presence: need n >= 0, A >= 1, triplets >= 0");
    if (Tn == 0) return MSM_OK;
    MSM_REQUIRE(ctx, d_triplets && d_counts, "msm_hbond_presence: NULL pointer");
    MSM_HIP(ctx, hipMemsetAsync(d_counts, 0, (size_t)Tn * sizeof(uint64_t), ctx->stream));
    if (n == 0) return MSM_OK;
    MSM_REQUIRE(ctx, d_xyz, "msm_hbond_presence: NULL pointer");
    const int gx = (int)std::min<int64_t>((Tn + kSThreads - 1) / kSThreads, 1024);
    // enough frame chunks to fill the chip, at least 16 frames each
    int64_t chunks = std::max<int64_t>(1, (int64_t)ctx->n_cu * 8 / gx);
    chunks = std::min<int64_t>(chunks, std::max<int64_t>(1, n / 16));
    chunks = std::min<int64_t>(chunks, 65535);
    const int fpb = (int)((n + chunks - 1) / chunks);
    const int gy = (int)((n + fpb - 1) / fpb);
    hipLaunchKernelGGL(hbond_presence_kernel, dim3(gx, gy), dim3(kSThreads), 0, ctx->stream, d_xyz, n, A, d_triplets, Tn,
                       dist_cutoff, angle_cutoff, fpb, (unsigned long long*)d_counts);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_dssp(msm_ctx* ctx, const float* d_xyz, int64_t n, int A, const int32_t* d_backbone,
                    const int32_t* d_chain, const uint8_t* d_proline, int R, uint8_t* d_codes) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 0 && A >= 1 && R >= 0, "msm_dssp: need n >= 0, A >= 1, residues >= 0");
    if (n == 0 || R == 0) return MSM_OK;
    MSM_REQUIRE(ctx, d_xyz && d_backbone && d_chain && d_proline && d_codes, "msm_dssp: NULL pointer");
    // work arrays, frames in chunks so that the scratch stays bounded
    const size_t per_frame = (size_t)R * (2 * sizeof(int) + 5 * sizeof(float) + 3 + 1 + 1 + 6 * sizeof(int)) + 128;
    const int64_t chunk = std::max<int64_t>(64, std::min<int64_t>(n, (int64_t)((256u << 20) / per_frame) / 64 * 64));
    msm_status rs = msm_reserve_scratch(ctx, (size_t)chunk * per_frame + 256);
    if (rs != MSM_OK) return rs;
    for (int64_t t0 = 0; t0 < n; t0 += chunk) {
        const int64_t m = std::min<int64_t>(chunk, n - t0);
        char* p = (char*)ctx->scratch;
        auto take = [&](size_t bytes) { char* q = p; p += (bytes + 15) & ~(size_t)15; return q; };
        DsspWork w;
        w.acc0 = (int*)take((size_t)R * m * sizeof(int));
        w.acc1 = (int*)take((size_t)R * m * sizeof(int));
        w.e0 = (float*)take((size_t)R * m * sizeof(float));
        w.e1 = (float*)take((size_t)R * m * sizeof(float));
        w.lad = (int*)take((size_t)6 * R * m * sizeof(int));
        w.hpos = (float*)take((size_t)3 * R * m * sizeof(float));
        w.hflag = (unsigned char*)take((size_t)3 * R * m);
        w.bend = (unsigned char*)take((size_t)R * m);
        w.ss = (unsigned char*)take((size_t)R * m);
        hipLaunchKernelGGL(dssp_kernel, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, ctx->stream,
                           d_xyz + t0 * (int64_t)A * 3, m, A, d_backbone, d_chain, d_proline, R, w, d_codes + t0 * (int64_t)R);
        MSM_CHECK_LAUNCH(ctx);
    }
    return MSM_OK;
}

}  // extern "C"
