// Dense symmetric eigensolver (parallel cyclic Jacobi, one workgroup) and the
// TICA solve built on it.  The matrices are F x F with F <= 256: this stage is
// latency-bound, not bandwidth- or flop-bound, so the design
// goal is "stay on the device, one launch": no host round trip sits between
// the covariance pass and the projection pass.
//
// Jacobi: the n(n-1)/2 pivots of a sweep are visited in n-1 rounds of n/2
// disjoint pairs (round-robin tournament).  Within a round all rotations are
// independent: compute (c, s) per pair, rotate rows, then rotate columns of A and
// V.  A and V live in LDS (odd row stride, conflict-free column walks) whenever
// 2 n (n+1) doubles fit; otherwise in global scratch.
#include "common.h"

// Diagnostic build only (tools/probe/jacobi_probe.hip): per-phase cycle stamps.
#ifdef MSM_JACOBI_STAMPS
__device__ unsigned long long g_jacobi_stamps[8];
#define JSTAMP(i)                                                                          \
    do {                                                                                   \
        unsigned long long t__;                                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");       \
        acc__[i] += t__ - last__;                                                          \
        last__ = t__;                                                                      \
    } while (0)
#define JSTAMP_INIT unsigned long long acc__[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long last__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last__)::"memory");
#define JSTAMP_FLUSH if (threadIdx.x == 0) { for (int i__ = 0; i__ < 8; ++i__) g_jacobi_stamps[i__] += acc__[i__]; }
#else
#define JSTAMP(i)
#define JSTAMP_INIT
#define JSTAMP_FLUSH
#endif

namespace {

constexpr int kEigThreads = 1024;
constexpr int kMaxPairs = 128;  // n <= 256

struct JacobiShared {
    double2 cs[kMaxPairs];  // (c, s) of pivot i: one 16-byte LDS read
    int p[kMaxPairs], q[kMaxPairs];
    int colw[kMaxPairs];  // p | q << 16            (even-n fast path: one read instead of the
    int roww[kMaxPairs];  // p * ld | (q * ld) << 16   tournament arithmetic + row multiplies)
    // second buffer of the pipelined path (n <= 64): rotations of round r+1 are formed by wave 0
    // while the other waves still apply round r
    double2 cs2[2][32];
    int colw2[2][32], roww2[2][32];
    double red[kEigThreads / 64];
    double bc[4];
    int ibc[4];
};

__device__ __forceinline__ double block_sum(double v, JacobiShared* sh) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh->red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh->red[i];
        sh->bc[0] = t;
    }
    __syncthreads();
    return sh->bc[0];
}

// Short-latency fp64 reciprocal / reciprocal square root: hardware seed plus two
// Newton steps (the IEEE division / sqrt expansions are 3-4x longer dependent chains,
// and the rotation set-up is the serial part of every Jacobi round).
__device__ __forceinline__ double nr_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}
__device__ __forceinline__ double nr_rsqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    r = r * fma(-h * r, r, 1.5);
    r = r * fma(-h * r, r, 1.5);
    return r;
}

// Jacobi rotation annihilating a_pq:  t = sgn(a) b / (|a| + sqrt(a^2 + b^2)) with
// a = (a_qq - a_pp)/2, b = a_pq;  c = 1/sqrt(1 + t^2), s = t c  (so c^2 + s^2 = 1 to
// rounding, which is what keeps V orthogonal).
__device__ __forceinline__ void jacobi_rotation(double app, double aqq, double apq, double& c, double& s) {
    c = 1.0; s = 0.0;
    if (apq == 0.0) return;
    const double a = 0.5 * (aqq - app);
    const double h2 = fma(a, a, apq * apq);
    if (!(h2 > 1e-300) || !(h2 < 1e300)) return;  // degenerate scale: skip this pivot
    // h and 1/(|a|+h) only steer the angle: one Newton step (~1e-13 relative) is plenty, the
    // pivot is zeroed explicitly; c below carries orthogonality and gets the full two steps.
    double rh = __builtin_amdgcn_rsq(h2);
    rh = rh * fma(-0.5 * h2 * rh, rh, 1.5);
    const double h = h2 * rh;
    const double den = fabs(a) + h;
    double rd = __builtin_amdgcn_rcp(den);
    rd = fma(rd, fma(-den, rd, 1.0), rd);
    const double t = (a >= 0.0 ? apq : -apq) * rd;
    c = nr_rsqrt(fma(t, t, 1.0));
    s = t * c;
}

// Round-robin tournament: pivot i of round r pairs indices (p, q); index npad-1 is the
// padding player when n is odd (it only ever appears as q).  Pure integer arithmetic (cheaper than an LDS table: the
// update phase is bound by LDS instruction issue).
// (p, q) is NOT ordered: p = round + i and q = round - i (mod npad-1) are runs of consecutive
// indices over i, so a wave's column accesses A[.][p_i] / A[.][q_i] fall on consecutive LDS
// banks (ordering the pair by min/max scrambled them: bank conflicts in the update phase).
__device__ __forceinline__ void pivot_pair(int round, int i, int npad, int& p, int& q) {
    if (i == 0) { p = round; q = npad - 1; return; }
    p = round + i; if (p >= npad - 1) p -= npad - 1;
    q = round - i; if (q < 0) q += npad - 1;
}

// A (n x n, row stride ld, symmetric) -> diagonal; V -> eigenvectors in columns.
// Returns the number of sweeps used (uniform across the block).
//
// One round = two phases, two barriers:
//  (1) lane i of the first waves forms the rotation of pivot i -- all m = n/2 serial
//      fp64 chains of the round run side by side (a lone dependent fp64 op costs ~44
//      cycles on this chip, so the chain, not the flop count, sets the pace);
//  (2) A <- J'AJ in ONE pass: the pivots partition the indices, so A splits into m x m
//      disjoint 2x2 blocks (rows of pivot a, columns of pivot b) and block (a, b) needs
//      only its own four entries and the two rotations -- no intermediate "rows done"
//      barrier.  V <- VJ rides in the same phase.
__device__ __forceinline__ int jacobi_eigh_pipelined(double* A, double* V, int n, int ld, JacobiShared* sh,
                                                     int max_sweeps);

__device__ __forceinline__ int jacobi_eigh(double* A, double* V, int n, int ld, JacobiShared* sh, int max_sweeps) {
    const int tid = threadIdx.x, nt = blockDim.x;
    {
        const int mh = n / 2;
        if ((n & 1) == 0 && mh >= 4 && mh <= 32 && mh * mh <= nt && mh * n <= 2 * nt && n * ld < 65536 && nt >= 64)
            return jacobi_eigh_pipelined(A, V, n, ld, sh, max_sweeps);
    }
    for (int i = tid; i < n * n; i += nt) V[(i / n) * ld + (i % n)] = (i / n == i % n) ? 1.0 : 0.0;
    __syncthreads();
    if (n < 2) return 0;
    const int npad = n + (n & 1);
    const int m = npad / 2;
    int sweep = 0;
    JSTAMP_INIT
    for (; sweep < max_sweeps; ++sweep) {
        JSTAMP(0);
        double off = 0.0, dia = 0.0;
        for (int i = tid; i < n * n; i += nt) {
            const int r = i / n, c = i - r * n;
            const double v = A[r * ld + c];
            if (r == c) dia = fma(v, v, dia); else off = fma(v, v, off);
        }
        off = block_sum(off, sh);
        dia = block_sum(dia, sh);
        // converged when ||off||_F <= n*eps*||A||_F: rounding of the rotations themselves
        // re-pollutes the zeroed entries at that level, and the eigenvalue error left
        // is second order in it.
        const double tol = (double)n * 2.220446049250313e-16;
        if (off <= tol * tol * (dia + off) || off == 0.0) break;
        JSTAMP(1);
        // fast path: one 2x2 block and at most two V items per thread, indices fixed for the sweep
        const bool fast = (m * m <= nt) && (m * n <= 2 * nt);
        // n even: nobody sits out, so the fast path needs no padding logic at all (n = 64 with 1024
        // threads: every thread owns exactly one 2x2 block and two V items)
        const bool fast_even = fast && (n & 1) == 0 && n * ld < 65536;
        const int my_ia = tid / m, my_ib = tid - my_ia * m;
        const bool has_blk = tid < m * m;
        const int v_ib0 = tid / n, v_r0 = tid - v_ib0 * n;
        const int v_ib1 = (tid + nt) / n, v_r1 = (tid + nt) - v_ib1 * n;
        const bool has_v0 = tid < m * n, has_v1 = tid + nt < m * n;
        // clamped copies keep every fast-path load in range for the threads without work
        const int my_ia_c = has_blk ? my_ia : 0, my_ib_c = has_blk ? my_ib : 0;
        const int v_ib0_c = has_v0 ? v_ib0 : 0, v_ib1_c = has_v1 ? v_ib1 : 0;
        const int v_r1_c = has_v1 ? v_r1 : 0;
        for (int round = 0; round < npad - 1; ++round) {
            if (tid < m) {
                const int i = tid;
                int p, q;
                pivot_pair(round, i, npad, p, q);
                double c = 1.0, s = 0.0;
                if (q < n) jacobi_rotation(A[p * ld + p], A[q * ld + q], A[p * ld + q], c, s);
                sh->cs[i] = make_double2(c, s); sh->p[i] = p; sh->q[i] = q;
                sh->colw[i] = p | (q << 16);
                sh->roww[i] = (p * ld) | ((q * ld) << 16);
            }
            JSTAMP(2);
            __syncthreads();
            JSTAMP(3);
            if (fast_even) {
                // ~70 instructions per thread (the phase is bound by instruction issue: 16 waves
                // on 4 SIMDs).  All loads first, clamped indices for idle threads, stores guarded.
                const int wr = sh->roww[my_ia_c], wc = sh->colw[my_ib_c];
                const int w0 = sh->colw[v_ib0_c], w1 = sh->colw[v_ib1_c];
                const double2 ra = sh->cs[my_ia_c], rb = sh->cs[my_ib_c], r0 = sh->cs[v_ib0_c], r1 = sh->cs[v_ib1_c];
                double* Ap = A + (wr & 0xffff);
                double* Aq = A + (wr >> 16);
                const int pb = wc & 0xffff, qb = wc >> 16;
                double* V0 = V + v_r0 * ld;
                double* V1 = V + v_r1_c * ld;
                const int vp0 = w0 & 0xffff, vq0 = w0 >> 16, vp1 = w1 & 0xffff, vq1 = w1 >> 16;
                const double app = Ap[pb], apq = Ap[qb], aqp = Aq[pb], aqq = Aq[qb];
                const double x0p = V0[vp0], x0q = V0[vq0], x1p = V1[vp1], x1q = V1[vq1];
                const double rpp = fma(-ra.y, aqp, ra.x * app), rpq = fma(-ra.y, aqq, ra.x * apq);
                const double rqp = fma(ra.y, app, ra.x * aqp), rqq = fma(ra.y, apq, ra.x * aqq);
                const double npp = fma(-rb.y, rpq, rb.x * rpp), nqq = fma(rb.y, rqp, rb.x * rqq);
                double npq = fma(rb.y, rpp, rb.x * rpq), nqp = fma(-rb.y, rqq, rb.x * rqp);
                if (my_ia_c == my_ib_c) { npq = 0.0; nqp = 0.0; }  // the pivot is annihilated exactly
                const double y0p = fma(-r0.y, x0q, r0.x * x0p), y0q = fma(r0.y, x0p, r0.x * x0q);
                const double y1p = fma(-r1.y, x1q, r1.x * x1p), y1q = fma(r1.y, x1p, r1.x * x1q);
                if (has_blk) { Ap[pb] = npp; Ap[qb] = npq; Aq[pb] = nqp; Aq[qb] = nqq; }
                if (has_v0) { V0[vp0] = y0p; V0[vq0] = y0q; }
                if (has_v1) { V1[vp1] = y1p; V1[vq1] = y1q; }
                JSTAMP(4);
                JSTAMP(5);
            } else if (fast) {
                // Branch-free: every load uses an in-range (clamped) address and is issued
                // before the first use; only the stores are predicated.  (Predicated LOADS
                // made the compiler fence each one with s_waitcnt: ~20 exposed LDS latencies.)
                int pa, qa, pb, qb, vp0, vq0, vp1, vq1;
                pivot_pair(round, my_ia_c, npad, pa, qa);
                pivot_pair(round, my_ib_c, npad, pb, qb);
                pivot_pair(round, v_ib0_c, npad, vp0, vq0);
                pivot_pair(round, v_ib1_c, npad, vp1, vq1);
                const bool row_real = qa < n, col_real = qb < n, v0 = has_v0 && vq0 < n, v1 = has_v1 && vq1 < n;
                const int qa_c = row_real ? qa : pa, qb_c = col_real ? qb : pb;
                const int vq0_c = vq0 < n ? vq0 : vp0, vq1_c = vq1 < n ? vq1 : vp1;
                const double2 ra = sh->cs[my_ia_c], rb = sh->cs[my_ib_c], r0 = sh->cs[v_ib0_c], r1 = sh->cs[v_ib1_c];
                const double app = A[pa * ld + pb], apq = A[pa * ld + qb_c];
                const double aqp = A[qa_c * ld + pb], aqq = A[qa_c * ld + qb_c];
                const double x0p = V[v_r0 * ld + vp0], x0q = V[v_r0 * ld + vq0_c];
                const double x1p = V[v_r1_c * ld + vp1], x1q = V[v_r1_c * ld + vq1_c];
                // the index sitting out (odd n) is not rotated
                const double ca = row_real ? ra.x : 1.0, sa = row_real ? ra.y : 0.0;
                const double cb = col_real ? rb.x : 1.0, sb = col_real ? rb.y : 0.0;
                const double rpp = fma(-sa, aqp, ca * app), rpq = fma(-sa, aqq, ca * apq);
                const double rqp = fma(sa, app, ca * aqp), rqq = fma(sa, apq, ca * aqq);
                double npp = fma(-sb, rpq, cb * rpp), npq = fma(sb, rpp, cb * rpq);
                double nqp = fma(-sb, rqq, cb * rqp), nqq = fma(sb, rqp, cb * rqq);
                if (my_ia_c == my_ib_c) { npq = 0.0; nqp = 0.0; }  // the pivot is annihilated exactly
                const double y0p = fma(-r0.y, x0q, r0.x * x0p), y0q = fma(r0.y, x0p, r0.x * x0q);
                const double y1p = fma(-r1.y, x1q, r1.x * x1p), y1q = fma(r1.y, x1p, r1.x * x1q);
                if (has_blk) A[pa * ld + pb] = npp;
                if (has_blk && col_real) A[pa * ld + qb] = npq;
                if (has_blk && row_real) A[qa * ld + pb] = nqp;
                if (has_blk && row_real && col_real) A[qa * ld + qb] = nqq;
                if (v0) { V[v_r0 * ld + vp0] = y0p; V[v_r0 * ld + vq0] = y0q; }
                if (v1) { V[v_r1_c * ld + vp1] = y1p; V[v_r1_c * ld + vq1] = y1q; }
                JSTAMP(4);
                JSTAMP(5);
            } else {
                for (int blk = tid; blk < m * m; blk += nt) {
                    const int ia = blk / m, ib = blk - ia * m;
                    const int pa = sh->p[ia], qa = sh->q[ia], pb = sh->p[ib], qb = sh->q[ib];
                    const bool row_real = qa < n, col_real = qb < n;  // odd n: one index sits out each round
                    if (!row_real && !col_real) continue;
                    const double ca = sh->cs[ia].x, sa = sh->cs[ia].y, cb = sh->cs[ib].x, sb = sh->cs[ib].y;
                    if (!col_real) {          // unpaired column pb: rows of pivot a only
                        const double ap = A[pa * ld + pb], aq = A[qa * ld + pb];
                        A[pa * ld + pb] = fma(-sa, aq, ca * ap);
                        A[qa * ld + pb] = fma(sa, ap, ca * aq);
                        continue;
                    }
                    if (!row_real) {          // unpaired row pa: columns of pivot b only
                        const double ap = A[pa * ld + pb], aq = A[pa * ld + qb];
                        A[pa * ld + pb] = fma(-sb, aq, cb * ap);
                        A[pa * ld + qb] = fma(sb, ap, cb * aq);
                        continue;
                    }
                    const double app = A[pa * ld + pb], apq = A[pa * ld + qb];
                    const double aqp = A[qa * ld + pb], aqq = A[qa * ld + qb];
                    const double rpp = fma(-sa, aqp, ca * app), rpq = fma(-sa, aqq, ca * apq);
                    const double rqp = fma(sa, app, ca * aqp), rqq = fma(sa, apq, ca * aqq);
                    double npp = fma(-sb, rpq, cb * rpp), npq = fma(sb, rpp, cb * rpq);
                    double nqp = fma(-sb, rqq, cb * rqp), nqq = fma(sb, rqp, cb * rqq);
                    if (ia == ib) { npq = 0.0; nqp = 0.0; }  // the pivot is annihilated exactly
                    A[pa * ld + pb] = npp; A[pa * ld + qb] = npq;
                    A[qa * ld + pb] = nqp; A[qa * ld + qb] = nqq;
                }
                JSTAMP(4);
                for (int e = tid; e < m * n; e += nt) {
                    const int ib = e / n, r = e - ib * n;
                    const int pb = sh->p[ib], qb = sh->q[ib];
                    if (qb >= n) continue;
                    const double cb = sh->cs[ib].x, sb = sh->cs[ib].y;
                    const double vp = V[r * ld + pb], vq = V[r * ld + qb];
                    V[r * ld + pb] = fma(-sb, vq, cb * vp);
                    V[r * ld + qb] = fma(sb, vp, cb * vq);
                }
                JSTAMP(5);
            }
            __syncthreads();
            JSTAMP(6);
        }
    }
    JSTAMP_FLUSH
    return sweep;
}

// Pipelined variant for even n <= 64 (m = n/2 pivots, 2m <= 64 lanes of wave 0).
//
// In pair-index space the entries the NEXT round's rotations need sit in a fixed set of 2m
// blocks: the m diagonal blocks (all diagonal entries) and, for next pivot j, block
// (j+1, j-1) (block (1,0) for j = 0, (2,0) for j = 1, (m-1, m-2) for j = m-1) -- the
// tournament only rotates the ring.  Wave 0 updates exactly those blocks first, reads its own
// results back (LDS operations of one wave execute in order) and forms the rotations of round
// r+1 into the other table while waves 1.. apply round r to the remaining blocks and to V:
// the ~700-cycle serial rotation chain leaves the critical path and one of the two barriers
// per round goes away.
__device__ __forceinline__ int jacobi_eigh_pipelined(double* A, double* V, int n, int ld, JacobiShared* sh,
                                                     int max_sweeps) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < n * n; i += nt) V[(i / n) * ld + (i % n)] = (i / n == i % n) ? 1.0 : 0.0;
    const int m = n / 2, ring = n - 1;
    auto form_rotations = [&](int round, int buf) {  // lanes 0..m-1 of wave 0
        int p, q;
        pivot_pair(round, tid, n, p, q);
        double c, s;
        jacobi_rotation(A[p * ld + p], A[q * ld + q], A[p * ld + q], c, s);
        sh->cs2[buf][tid] = make_double2(c, s);
        sh->colw2[buf][tid] = p | (q << 16);
        sh->roww2[buf][tid] = (p * ld) | ((q * ld) << 16);
    };
    auto update_block = [&](int ia, int ib, int buf, bool live) {
        const int wr = sh->roww2[buf][ia], wc = sh->colw2[buf][ib];
        const double2 ra = sh->cs2[buf][ia], rb = sh->cs2[buf][ib];
        double* Ap = A + (wr & 0xffff);
        double* Aq = A + (wr >> 16);
        const int pb = wc & 0xffff, qb = wc >> 16;
        const double app = Ap[pb], apq = Ap[qb], aqp = Aq[pb], aqq = Aq[qb];
        const double rpp = fma(-ra.y, aqp, ra.x * app), rpq = fma(-ra.y, aqq, ra.x * apq);
        const double rqp = fma(ra.y, app, ra.x * aqp), rqq = fma(ra.y, apq, ra.x * aqq);
        const double npp = fma(-rb.y, rpq, rb.x * rpp), nqq = fma(rb.y, rqp, rb.x * rqq);
        double npq = fma(rb.y, rpp, rb.x * rpq), nqp = fma(-rb.y, rqq, rb.x * rqp);
        if (ia == ib) { npq = 0.0; nqp = 0.0; }  // the pivot is annihilated exactly
        if (live) { Ap[pb] = npp; Ap[qb] = npq; Aq[pb] = nqp; Aq[qb] = nqq; }
    };
    // static roles
    const int my_ia = tid / m, my_ib = tid - my_ia * m;
    const bool priority = (my_ib == my_ia) || (my_ia >= 2 && my_ib == my_ia - 2) || (my_ia == 1 && my_ib == 0) ||
                          (my_ia == m - 1 && my_ib == m - 2);
    const bool has_blk = tid < m * m && !priority;
    const int my_ia_c = tid < m * m ? my_ia : 0, my_ib_c = tid < m * m ? my_ib : 0;
    // wave 0: lane l < m -> diagonal block l; lane m + j -> the block holding next pivot j
    int pr_ia = 0, pr_ib = 0;
    const bool has_pr = tid < 2 * m;
    if (tid < m) { pr_ia = tid; pr_ib = tid; }
    else if (tid < 2 * m) {
        const int j = tid - m;
        if (j == 0) { pr_ia = 1; pr_ib = 0; }
        else if (j == 1) { pr_ia = 2; pr_ib = 0; }
        else if (j == m - 1) { pr_ia = m - 1; pr_ib = m - 2; }
        else { pr_ia = j + 1; pr_ib = j - 1; }
    }
    const int v_ib0 = tid / n, v_r0 = tid - v_ib0 * n;
    const int v_ib1 = (tid + nt) / n, v_r1 = (tid + nt) - v_ib1 * n;
    const bool has_v0 = tid < m * n, has_v1 = tid + nt < m * n;
    const int v_ib0_c = has_v0 ? v_ib0 : 0, v_ib1_c = has_v1 ? v_ib1 : 0, v_r1_c = has_v1 ? v_r1 : 0;
    __syncthreads();
    int buf = 0, round = 0;
    if (tid < m) form_rotations(0, 0);
    __syncthreads();
    int sweep = 0;
    for (; sweep < max_sweeps; ++sweep) {
        double off = 0.0, dia = 0.0;
        for (int i = tid; i < n * n; i += nt) {
            const int r = i / n, c = i - r * n;
            const double v = A[r * ld + c];
            if (r == c) dia = fma(v, v, dia); else off = fma(v, v, off);
        }
        off = block_sum(off, sh);
        dia = block_sum(dia, sh);
        const double tol = (double)n * 2.220446049250313e-16;  // see jacobi_eigh
        if (off <= tol * tol * (dia + off) || off == 0.0) break;
        for (int rr = 0; rr < ring; ++rr) {
            const int next_round = round + 1 == ring ? 0 : round + 1;
            if (tid < 64) {  // wave 0 (uniform branch)
                update_block(pr_ia, pr_ib, buf, has_pr);
                if (tid < m) form_rotations(next_round, buf ^ 1);
            }
            update_block(my_ia_c, my_ib_c, buf, has_blk);
            {
                const int w0 = sh->colw2[buf][v_ib0_c], w1 = sh->colw2[buf][v_ib1_c];
                const double2 r0 = sh->cs2[buf][v_ib0_c], r1 = sh->cs2[buf][v_ib1_c];
                double* V0 = V + v_r0 * ld;
                double* V1 = V + v_r1_c * ld;
                const int vp0 = w0 & 0xffff, vq0 = w0 >> 16, vp1 = w1 & 0xffff, vq1 = w1 >> 16;
                const double x0p = V0[vp0], x0q = V0[vq0], x1p = V1[vp1], x1q = V1[vq1];
                const double y0p = fma(-r0.y, x0q, r0.x * x0p), y0q = fma(r0.y, x0p, r0.x * x0q);
                const double y1p = fma(-r1.y, x1q, r1.x * x1p), y1q = fma(r1.y, x1p, r1.x * x1q);
                if (has_v0) { V0[vp0] = y0p; V0[vq0] = y0q; }
                if (has_v1) { V1[vp1] = y1p; V1[vq1] = y1q; }
            }
            __syncthreads();
            buf ^= 1;
            round = next_round;
        }
    }
    return sweep;
}

// order[j] = index of the j-th largest |ev| (stable)
__device__ void sort_desc_abs(const double* ev, int n, int* order) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double a = fabs(ev[i]);
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const double b = fabs(ev[j]);
            rank += (b > a) || (b == a && j < i);
        }
        order[rank] = i;
    }
    __syncthreads();
}

// column j of M (n rows, stride ld) *= sign of its largest-magnitude entry (first occurrence)
__device__ void canonical_signs(double* M, int n, int ncols, int ld) {
    for (int j = threadIdx.x; j < ncols; j += blockDim.x) {
        double best = -1.0, sgn = 1.0;
        for (int i = 0; i < n; ++i) {
            const double v = M[i * ld + j];
            if (fabs(v) > best) { best = fabs(v); sgn = v < 0.0 ? -1.0 : 1.0; }
        }
        if (sgn < 0.0)
            for (int i = 0; i < n; ++i) M[i * ld + j] = -M[i * ld + j];
    }
    __syncthreads();
}

// Two factorisations in lockstep (same barriers): P = chol(Mp) is only a positive-definiteness
// probe, Q = chol(Mq) is the factor that is used.  Returns false as soon as Mp loses a pivot
// (Mq is then unfinished and must not be used).
__device__ bool cholesky_lower_pair(double* Mp, double* Mq, int n, int ld, JacobiShared* sh) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int j = 0; j < n; ++j) {
        if (tid < 2) {
            double* M = tid == 0 ? Mp : Mq;
            const double dj = M[j * ld + j];
            sh->ibc[1 + tid] = dj > 0.0 ? 1 : 0;
            if (dj > 0.0) M[j * ld + j] = sqrt(dj);
        }
        __syncthreads();
        if (!sh->ibc[1] || !sh->ibc[2]) return false;
        const double pjj = Mp[j * ld + j], qjj = Mq[j * ld + j];
        for (int i = j + 1 + tid; i < n; i += nt) {
            Mp[i * ld + j] /= pjj;
            Mq[i * ld + j] /= qjj;
        }
        __syncthreads();
        const int m = n - j - 1;
        for (int e = tid; e < m * m; e += nt) {
            const int a = e / m, b = e - a * m;  // trailing (j+1+a, j+1+b), lower part only
            if (b <= a) {
                const int ra = (j + 1 + a) * ld, rb = (j + 1 + b) * ld;
                Mp[ra + j + 1 + b] -= Mp[ra + j] * Mp[rb + j];
                Mq[ra + j + 1 + b] -= Mq[ra + j] * Mq[rb + j];
            }
        }
        __syncthreads();
    }
    return true;
}

// X = G^-1 for lower-triangular G; X lower.  Right-looking elimination on [G | I]: per pivot k
// one scaling of row k and one rank-1 update of the rows below, both fully parallel -- n short
// steps instead of per-column chains of ~n^2/2 dependent FMAs (a dependent fp64 FMA costs ~44
// cycles here, a workgroup barrier ~100).  X must not alias G.
__device__ void lower_inverse(const double* G, double* X, int n, int ld) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, c = e - i * n;
        X[i * ld + c] = i == c ? 1.0 : 0.0;
    }
    __syncthreads();
    for (int k = 0; k < n; ++k) {
        const double gkk = G[k * ld + k];
        for (int c = tid; c <= k; c += nt) X[k * ld + c] /= gkk;
        __syncthreads();
        const int rows = n - k - 1, cols = k + 1;
        for (int e = tid; e < rows * cols; e += nt) {
            const int a = e / cols, c = e - a * cols;
            const int i = k + 1 + a;
            X[i * ld + c] = fma(-G[i * ld + k], X[k * ld + c], X[i * ld + c]);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// Symmetric eigensolver for n <= 64 without the ~600 barrier-bound Jacobi rounds:
//   (1) Householder tridiagonalisation A = Q T Q' with A and Q' held in the registers of eight waves (householder_phase
//       below: rows dealt out four ways, lane = column, two barriers per column);
//   (2) all eigenvalues of T at once by multisection on Sturm counts: 8 lanes per eigenvalue (two waves per SIMD: a
//       lone wave only gets every other issue slot), 8 trial points per round, 20 rounds; the count uses the product
//       recurrence p_j = (d_j - x) p_{j-1} - e_{j-1}^2 p_{j-2} (no division), sign changes shifted into a bit mask,
//       rescaled every 8 rows;
//   (3) one eigenvector of T per lane from the twisted factorisation of T - lambda I (forward and backward
//       pivots, twist at the smallest |gamma|: Parlett & Dhillon), i.e. one exact inverse-iteration step;
//   (4) Z = Q X on the matrix cores, the 1/|x| factors folded into the operand.
// Eigenvectors of T belonging to eigenvalues closer than ~1e-9 |T| come out of (3) short of orthogonal, and
// exactly repeated eigenvalues give repeated vectors: X'X is checked and the caller falls back to Jacobi on its
// copy of A when the defect exceeds `orth_tol` (or a residual |T x - lambda x| is large).
// ---------------------------------------------------------------------------------------------------
#ifdef MSM_TRI_STAMPS
__device__ unsigned long long g_tri_stamps[24];
#define TSTAMP(i) do { if (threadIdx.x == 0) { unsigned long long t__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory"); g_tri_stamps[i] += t__ - tlast__; tlast__ = t__; } } while (0)
#define TSTAMP_INIT unsigned long long tlast__ = 0; if (threadIdx.x == 0) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast__)::"memory"); }
#define KSTAMP(i) do { if (threadIdx.x == 0) { unsigned long long t__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory"); g_tri_stamps[i] += t__ - klast__; klast__ = t__; } } while (0)
#define KSTAMP_INIT unsigned long long klast__ = 0; if (threadIdx.x == 0) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(klast__)::"memory"); }
#ifdef MSM_TRI_STAMPS_FINE
#define LSTAMP(i) TSTAMP(i)
#define LSTAMP_INIT TSTAMP_INIT
#else
#define LSTAMP(i)
#define LSTAMP_INIT
#endif
#else
#define LSTAMP(i)
#define LSTAMP_INIT
#define TSTAMP(i)
#define TSTAMP_INIT
#define KSTAMP(i)
#define KSTAMP_INIT
#endif
constexpr int kTriMax = 64;
constexpr int kTriLd = 65;   // row stride of every matrix handed to tridiag_eigh (a constant: row offsets become immediates)
struct TriShared {
    double2 de[kTriMax + 8];   // {d_j, e_{j-1}^2} of the scaled matrix; rows n.. are {8, 0} (no sign change)
    double d[kTriMax], e[kTriMax], lam[kTriMax], praw[kTriMax], uq[kTriMax], inv[kTriMax];
    double scale, gl, gu;
    double red[kEigThreads / 64];
    int bad;
    double part[16][kTriMax];  // partial products of the tridiagonalisation, one row per wave (the LDL' phase uses rows 0-7)
    union {                    // the two register-resident phases never overlap
        double2 vw[8][kTriMax];                 // tridiagonalisation: {v_j, w_j} of the column in flight, one copy per A wave
        alignas(16) double lv[16][kTriMax];     // LDL': multipliers of the column in flight, one copy per working wave
    };
    double vpub[2][kTriMax];   // tridiagonalisation: the reflector of column k for the Q' waves, slot k & 1
    double taupub[2];
};

typedef double tri_v4f64 __attribute__((ext_vector_type(4)));

// 64-lane sum on the VALU (row DPP moves, then the two row swaps of gfx950): every lane gets the total, bit for
// bit the same in every wave that feeds it the same numbers
template <int CTRL>
__device__ __forceinline__ double mov_dpp64(double x) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// the bare v_max_f64 / v_min_f64 (operands never NaN where these are used): fmax() on a value that came through a DPP
// move costs an extra canonicalising v_max_f64 per call
__device__ __forceinline__ double hw_max_f64(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double hw_min_f64(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
typedef unsigned tri_v2u32 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double sum8_dpp(double x) {   // sum over aligned groups of 8 lanes
    x += mov_dpp64<0xB1>(x);    // quad_perm [1,0,3,2]
    x += mov_dpp64<0x4E>(x);    // quad_perm [2,3,0,1]
    x += mov_dpp64<0x141>(x);   // row_half_mirror
    return x;
}
__device__ __forceinline__ double wave_sum_all(double x) {
    x = sum8_dpp(x);
    x += mov_dpp64<0x140>(x);   // row_mirror: all 16 lanes of a row hold the row sum
    {
        const long long b = __double_as_longlong(x);
        const tri_v2u32 rl = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
        const tri_v2u32 rh = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
        x = __longlong_as_double(((long long)rh[0] << 32) | rl[0]) + __longlong_as_double(((long long)rh[1] << 32) | rl[1]);
    }
    {
        const long long b = __double_as_longlong(x);
        const tri_v2u32 rl = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
        const tri_v2u32 rh = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
        x = __longlong_as_double(((long long)rh[0] << 32) | rl[0]) + __longlong_as_double(((long long)rh[1] << 32) | rl[1]);
    }
    return x;
}

// C[i][j] = sum_k opA(i,k) opB(k,j) on the matrix cores (v_mfma_f64_16x16x4_f64; one 16 x 16 tile of C per wave
// and trip), i < rows, j < cols, k < inner; opA = A or A', opB = B or B'.  bscale (optional) multiplies column j of
// opB.  C must not alias A or B.  A 64^3 product costs what it costs on the VALU (the fp64 rates are equal) but
// reads each operand from the LDS once per tile instead of once per multiply.
template <bool TA, bool TB>
__device__ __forceinline__ void mfma_mm(double* C, const double* A, const double* B, int rows, int cols, int inner, int ld,
                        const double* bscale = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int tr = (rows + 15) >> 4, tc = (cols + 15) >> 4;
    for (int t = wave; t < tr * tc; t += nw) {
        const int i0 = (t / tc) * 16, c0 = (t - (t / tc) * tc) * 16;
        const bool aok = i0 + j < rows, bok = c0 + j < cols;
        const int ai = aok ? i0 + j : 0, bj = bok ? c0 + j : 0;
        const double bs = bscale ? bscale[bj] : 1.0;
        tri_v4f64 acc = {0.0, 0.0, 0.0, 0.0};
        for (int k0 = 0; k0 < inner; k0 += 16) {   // four instructions per trip: their operand reads go out together
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool kok = k0 + 4 * u + g < inner;
                const int k = kok ? k0 + 4 * u + g : 0;
                a[u] = TA ? A[k * ld + ai] : A[ai * ld + k];
                b[u] = (TB ? B[bj * ld + k] : B[k * ld + bj]) * bs;
                if (!(aok && kok)) a[u] = 0.0;
                if (!(bok && kok)) b[u] = 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = i0 + g + 4 * r, col = c0 + j;
            if (row < rows && col < cols) C[row * ld + col] = acc[r];
        }
    }
    __syncthreads();
}

// # eigenvalues of the (scaled) tridiagonal matrix below x: sign changes of p_{-1} = 1, p_0, ..., p_{n-1}.  The sign
// bits are shifted into a mask as the rows go by (one v_alignbit per row) and the changes counted at the end,
// popc(S ^ (S >> 1)) per 32 rows: per row 4 vector instructions (sub, mul, fma, alignbit).  An exact zero counts as
// positive: it only occurs when x is an eigenvalue of a leading block, and a bracket spoilt by it fails the residual
// check.  Eight rows at a time: their {d, e^2} pairs come out of the LDS in one burst.
__device__ __forceinline__ int sturm_count(const TriShared* ts, int nblocks, double x) {
    double pp = 0.0, p = 1.0;
    unsigned bits = 0, w0 = 0;
    for (int b = 0; b < nblocks; ++b) {
        // the rows are re-read from the LDS on every call: kept in registers across the rounds of the caller they
        // would take 2 x 128 registers, and the compiler spills to scratch to try
        asm volatile("" ::: "memory");
        double2 r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = ts->de[8 * b + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double pn = fma(r[u].x - x, p, -(r[u].y * pp));
            bits = __builtin_amdgcn_alignbit(bits, (unsigned)__double2hiint(pn), 31);   // (bits << 1) | sign(pn)
            pp = p;
            p = pn;
        }
        if (b & 1) {
            // keep the pair in range: the matrix is scaled to |T| <= 1, so sixteen rows grow it by 2.5^16 at most and
            // cannot shrink it below the normal range unless x sits within 1e-19 of sixteen diagonal entries in a row
            const int ex = max(__builtin_amdgcn_frexp_exp(p), __builtin_amdgcn_frexp_exp(pp));
            p = ldexp(p, -ex);
            pp = ldexp(pp, -ex);
        }
        if (b == 3) { w0 = bits; bits = 0; }
    }
    if (nblocks <= 4) return __popc(bits ^ (bits >> 1));   // right-aligned: the zeros above stand for p_{-1} > 0
    // rows 32.. sit right-aligned in `bits`; the row above the first of them is row 31 = bit 0 of w0
    const int r1 = 8 * nblocks - 32;
    return __popc(w0 ^ (w0 >> 1)) + __popc(bits ^ ((bits >> 1) | ((w0 & 1u) << (r1 - 1))));
}

// Tridiagonalisation A = Q T Q' with the matrices held in REGISTERS, every wave of the workgroup at work, two
// barriers per column.
// Measured on this chip (tools/probe/lone_wave_lds_probe.hip, tri_probe.hip): a wave issues about one instruction per
// 10 cycles whatever the instruction and however many waves share its SIMD (a SIMD takes one per ~4 cycles between
// all its waves), pays ~37 cycles per taken branch and 12-19 per LDS instruction, so one barrier-free wave sweeping an
// LDS-resident matrix needed ~5000 cycles per column and sixteen waves sweeping it between barriers were no faster.
// Here nothing is swept and the per-wave instruction chain of a column is kept short:
//   waves 0-7 hold the rows j = w (mod 8) of A for the whole reduction, waves 8-15 the same rows of Q'
//   (QT[j][r] = Q[r][j]): 8 doubles per lane, indexed by compile-time constants only (lane = column);
//   A waves, column k (the reflector of column 0 is formed ahead of the loop):
//   (1) partial product p = A22 v over the own rows (the v_j come back as broadcast reads of the wave's own LDS copy);
//       barrier 1;
//   (2) the eight partials are summed in a fixed order, w = tau p - (tau^2 p.v / 2) v; the owner of row k + 1 puts the
//       new value of that row in the LDS at once; barrier 2;
//   (3) a_j -= v_j w + w_j v on the own rows and, in the same straight-line stretch, the reflector of column k + 1
//       from row k + 1, every A wave for itself (same inputs, same instructions, same bits): x = A[k+1][k+2:],
//       H = I - tau v v', v_{k+2} = 1; wave 0 also leaves v and tau in the LDS for the Q' waves.  A column that is
//       already reduced gives tau = 0 and everything downstream of it is arithmetic with zeros: no branches.
//   Q' waves run half a column behind and never form a reflector: between barrier 1 and 2 of column k they read v_k and
//   write their partial u = Q v; between barrier 2 and barrier 1 of the next column they sum the partials and apply
//   q_j -= tau v_j u.  They add nothing to the A waves' chain, which is what a column costs.
// Rows up to k carry v_j = w_j = 0; groups of four own rows that lie entirely there are skipped.
__device__ __forceinline__ double bcast_lane(double x, int j) {   // lane j's value to everybody; j uniform
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)b, j), hi = __builtin_amdgcn_readlane((int)(b >> 32), j);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <typename F>
__device__ __forceinline__ void hh_for_groups(int sub, int k, F&& body) {
    // own rows sub + 8 t, t = 4 g + u; group g is dead once its last row sub + 32 g + 24 <= k
#pragma unroll
    for (int g = 0; g < 2; ++g)
        if (sub + 32 * g + 24 > k) body(g);
}
__device__ __forceinline__ void householder_phase(double* A, double* QT, int n, TriShared* ts) {
    static_assert(kEigThreads == 1024, "householder_phase deals the rows of A and Q' over sixteen waves");
    constexpr int ld = kTriLd;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool in = lane < n;
    const bool is_a = wave < 8;
#ifdef MSM_TRI_NOQ   // diagnostic build only (tools/probe/tri_probe.hip): what the tridiagonalisation costs without its Q waves
    const bool is_q = false;
#else
    const bool is_q = !is_a;
#endif
    const int sub = wave & 7;
    double* part = ts->part[wave];
    const double* parts = ts->part[is_a ? 0 : 8];
    double2* vw = ts->vw[sub];        // A waves: this wave's own copy of {v_j, w_j}
    double m[8];                      // rows sub + 8 t of A or Q', column `lane`
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int j = sub + 8 * t;
        m[t] = (in && j < n) ? (is_a ? A[j * ld + lane] : (j == lane ? 1.0 : 0.0)) : 0.0;
    }
    // Q' side of column c: partial u = Q v over the own rows / sum of the partials and the update
    auto q_partial = [&](int c) {
        if (__builtin_amdgcn_readfirstlane(ts->taupub[c & 1] != 0.0)) {
            const double* vp = ts->vpub[c & 1];
            double s0 = 0.0, s1 = 0.0;
            hh_for_groups(sub, c, [&](int g) {
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = vp[sub + 32 * g + 8 * u];
                s0 = fma(m[4 * g], v[0], s0);
                s1 = fma(m[4 * g + 1], v[1], s1);
                s0 = fma(m[4 * g + 2], v[2], s0);
                s1 = fma(m[4 * g + 3], v[3], s1);
            });
            part[lane] = s0 + s1;
        }
    };
    auto sum_parts = [&]() {
        return (((parts[lane] + parts[kTriMax + lane]) + (parts[2 * kTriMax + lane] + parts[3 * kTriMax + lane])) +
                ((parts[4 * kTriMax + lane] + parts[5 * kTriMax + lane]) + (parts[6 * kTriMax + lane] + parts[7 * kTriMax + lane])));
    };
    auto q_finish = [&](int c) {
        const double tau_c = ts->taupub[c & 1];
        if (__builtin_amdgcn_readfirstlane(tau_c != 0.0)) {
            const double* vp = ts->vpub[c & 1];
            const double tu = -tau_c * sum_parts();
            hh_for_groups(sub, c, [&](int g) {
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = vp[sub + 32 * g + 8 * u];
#pragma unroll
                for (int u = 0; u < 4; ++u) m[4 * g + u] = fma(tu, v[u], m[4 * g + u]);
            });
        }
    };
    // reflector of column c from row c in the LDS (straight-line: a column that is already reduced gives tau = 0, v = e_{c+1},
    // and everything downstream of it is a no-op); wave 0 also records d_c, e_c and leaves v, tau for the Q' waves
    auto reflector = [&](int c, double& vi, double& tau) {
        const double* rowc = A + c * ld;
        double xi = rowc[in ? lane : 0];
        const double alpha = rowc[c + 1];
        if (wave == 0 && lane == c) ts->d[c] = xi;
        const bool actc = in && lane > c;
        if (!actc) xi = 0.0;
        const double sigma = wave_sum_all(lane > c + 1 ? xi * xi : 0.0);
        const bool nz = sigma != 0.0;   // the same in every lane
        const double h2 = fma(alpha, alpha, sigma);
        const double rs = nr_rsqrt(nz ? h2 : 1.0);
        const double beta = nz ? -copysign(h2 * rs, alpha) : alpha;
        tau = nz ? (beta - alpha) * -copysign(rs, alpha) : 0.0;   // (beta - alpha) / beta
        const double scal = nr_rcp(nz ? alpha - beta : 1.0);
        vi = actc ? (lane == c + 1 ? 1.0 : (nz ? xi * scal : 0.0)) : 0.0;
        if (wave == 0) {
            ts->vpub[c & 1][lane] = vi;
            if (lane == 0) { ts->e[c] = beta; ts->taupub[c & 1] = tau; }
        }
        vw[lane].x = vi;
    };
    LSTAMP_INIT
    double vi = 0.0, tau = 0.0;
    if (is_a && n > 2) reflector(0, vi, tau);
    for (int k = 0; k + 2 < n; ++k) {
        LSTAMP(16);
        const bool act = in && lane > k;
        if (is_a) {
            // partial product p = A22 v over the own rows
            double s0 = 0.0, s1 = 0.0;
            hh_for_groups(sub, k, [&](int g) {
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = vw[sub + 32 * g + 8 * u].x;
                s0 = fma(m[4 * g], v[0], s0);
                s1 = fma(m[4 * g + 1], v[1], s1);
                s0 = fma(m[4 * g + 2], v[2], s0);
                s1 = fma(m[4 * g + 3], v[3], s1);
            });
            part[lane] = s0 + s1;
            LSTAMP(18);
        } else if (is_q && k > 0) {
            q_finish(k - 1);
        }
        __syncthreads();
        LSTAMP(19);
        double wi = 0.0;
        if (is_a) {
            const double p = sum_parts();
            const double pv = wave_sum_all(act ? p * vi : 0.0);
            const double al = -0.5 * tau * (tau * pv);
            wi = act ? fma(tau, p, al * vi) : 0.0;
            vw[lane].y = wi;
            LSTAMP(20);
            // the owner of row k + 1 leaves its new value in the LDS ahead of the rest of the update: everybody's next
            // reflector hangs on it (the same two fmas again below: the same bits)
#pragma unroll
            for (int t2 = 0; t2 < 8; ++t2)
                if (sub + 8 * t2 == k + 1) {
                    const double2 bk = vw[k + 1];
                    if (in) A[(k + 1) * ld + lane] = fma(-wi, bk.x, fma(-vi, bk.y, m[t2]));
                }
        } else if (is_q) {
            q_partial(k);
        }
        LSTAMP(21);
        __syncthreads();
        if (is_a) {
            // rank-2 update of the own rows (rows up to k carry v_j = w_j = 0: untouched) and, in the same straight-line
            // stretch, the reflector of the next column: two independent chains for the scheduler to interleave
            double2 bb[8];
#pragma unroll
            for (int t2 = 0; t2 < 8; ++t2) bb[t2] = vw[sub + 8 * t2];
            const double vk = vi, wk = wi;
            if (k + 3 < n) reflector(k + 1, vi, tau);
#pragma unroll
            for (int t2 = 0; t2 < 8; ++t2) m[t2] = fma(-wk, bb[t2].x, fma(-vk, bb[t2].y, m[t2]));
            LSTAMP(17);
        }
    }
    if (is_q && n > 2) q_finish(n - 3);
    // back to the LDS: Q' whole, of A the last two rows (the 2 x 2 block the reduction leaves)
    if (is_q && in) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (sub + 8 * t < n) QT[(sub + 8 * t) * ld + lane] = m[t];
    }
    if (is_a && in) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (sub + 8 * t >= n - 2 && sub + 8 * t < n) A[(sub + 8 * t) * ld + lane] = m[t];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ts->d[n - 2] = A[(n - 2) * ld + n - 2];
        ts->e[n - 2] = A[(n - 1) * ld + n - 2];
        ts->d[n - 1] = A[(n - 1) * ld + n - 1];
        ts->e[n - 1] = 0.0;
    }
}

// Whitening of a full-rank C00 in registers, the pattern of householder_phase: an LDL' elimination with X = L^-1 carried
// along, one barrier per column.  Waves 0-7 hold the rows i = w (mod 8) of C00, waves 8-15 the same rows of X (= identity
// at the start); lane = column; 8 doubles per lane, compile-time indices only.
// Column j: the owner of row j has left it in the LDS.  The trailing matrix is symmetric, so that row is also column j:
// lane i of (row j) / d_j IS the multiplier l_i; every wave parks the multipliers in its own LDS copy, ordered so that
// those of its eight rows come back as four 16-byte broadcast reads.  Row i of C00 or X then takes one fma (l_i = 0 up
// to row j: finished rows stay); the owner of row j + 1 leaves it in the LDS for the next column.  A wave issues about
// one instruction per 10 cycles, so the column's chain is kept short by dealing the rows over all sixteen waves.
// On success W = (D^-1/2 L^-1)' (upper triangular, W' C00 W = I) is written to `W`; false (uniformly) as soon as a pivot
// is not positive.  C is left as it was.  blockDim.x == 1024.
// mode 1: C00 and X (the caller certifies full rank from W, see tica_solve_kernel);
// mode 2: the probe C00 - epsilon I alone (waves 0-7), nothing is written to W.
__device__ __forceinline__ bool ldl_whiten_registers(const double* C, double* W, int n, double epsilon, TriShared* ts,
                                                     int mode) {
    static_assert(kEigThreads == 1024, "ldl_whiten_registers deals the rows over sixteen waves");
    constexpr int ld = kTriLd;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool in = lane < n;
    // 0: C00, 1: X, 2: probe, 3: nothing
    const int what = mode == 2 ? (wave < 8 ? 2 : 3) : (wave < 8 ? 0 : 1);
    const int src = mode == 2 ? 2 : 0;   // the matrix the multipliers come from: its published rows sit in part[par + src]
    const int sub = wave & 7;
    double m[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int i = sub + 8 * t;
        if (what == 1) m[t] = lane == i ? 1.0 : 0.0;
        else m[t] = (what < 3 && in && i < n) ? C[i * ld + lane] - (what == 2 && lane == i ? epsilon : 0.0) : 0.0;
    }
    if (sub == 0 && what < 3) ts->part[what][lane] = m[0];   // row 0 for column 0
    __syncthreads();
    LSTAMP_INIT
    for (int j = 0; j < n; ++j) {
        const int par = (j & 1) * 4;
        if (what < 3) {
            // row j of the matrix the multipliers come from and of the matrix this wave updates
            const double rowM = ts->part[par + src][lane];
            const double row = what == 1 ? ts->part[par + 1][lane] : rowM;
            const double d = bcast_lane(rowM, j);
            LSTAMP(13);
            if (!(d > 0.0)) return false;   // uniform over the workgroup (see below for the idle waves)
            {
                // multipliers of the live rows (zero up to row j: finished rows stay as they are), dealt out so that
                // the eight of this wave's rows lie side by side in its LDS copy: l_i at (i mod 8) * 8 + i / 8
                const double lvec = lane > j ? rowM * nr_rcp(d) : 0.0;
                double* lv = ts->lv[wave];
                lv[(lane & 7) * 8 + (lane >> 3)] = lvec;
                const double2* mine = reinterpret_cast<const double2*>(lv + sub * 8);
#pragma unroll
                for (int t2 = 0; t2 < 4; ++t2) {
                    const double2 l = mine[t2];
                    m[2 * t2] = fma(-l.x, row, m[2 * t2]);
                    m[2 * t2 + 1] = fma(-l.y, row, m[2 * t2 + 1]);
                }
                if (((j + 1 - sub) & 7) == 0) {   // this wave owns row j + 1: leave it for the next column
                    double* dst = ts->part[(par ^ 4) + what] + lane;
                    switch ((j + 1 - sub) >> 3) {
                        case 0: *dst = m[0]; break;   case 1: *dst = m[1]; break;   case 2: *dst = m[2]; break;
                        case 3: *dst = m[3]; break;   case 4: *dst = m[4]; break;   case 5: *dst = m[5]; break;
                        case 6: *dst = m[6]; break;   case 7: *dst = m[7]; break;   default: break;
                    }
                }
            }
        } else {   // waves without rows: the same pivot, the same decision
            const double dP = ts->part[par + src][j];
            if (!(dP > 0.0)) return false;
        }
        LSTAMP(14);
        __syncthreads();
        LSTAMP(15);
    }
    // d_i sits in lane i of row i of the C00 waves (row i was final after column i - 1): hand D^-1/2 to the X waves
    if (what == 0) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (lane == sub + 8 * t && lane < n) ts->inv[lane] = nr_rsqrt(m[t]);
    }
    __syncthreads();
    if (what == 1 && in) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int i = sub + 8 * t;
            if (i < n) W[lane * ld + i] = m[t] * ts->inv[i];   // X is lower triangular: zeros above the diagonal of W'
        }
    }
    __syncthreads();
    return true;
}

// A: symmetric n x n (destroyed; on success its columns hold the eigenvectors, ascending eigenvalues in
// ts->lam).  Q (ends up holding Q'), X: n x n work matrices (same stride).  Returns false (uniformly) when the result must not be
// used; the caller then solves its own copy of A by Jacobi.  blockDim.x == 1024.
__device__ __forceinline__ bool tridiag_eigh(double* A, double* Q, double* X, int n, int ld, TriShared* ts, double orth_tol) {
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6;
    if (n == 1) {
        if (tid == 0) { ts->lam[0] = A[0]; A[0] = 1.0; }
        __syncthreads();
        return true;
    }
    if (ld != kTriLd) return false;
    TSTAMP_INIT
    if (tid == 0) ts->bad = 0;
    __syncthreads();
    TSTAMP(0);
    // ---- (1) tridiagonalisation
    householder_phase(A, Q, n, ts);
    TSTAMP(1);
    __syncthreads();
    TSTAMP(2);
    {   // scale to |T| <= 1 by a power of two (exact), Gershgorin bounds of the scaled matrix: wave-parallel
        const double dl = lane < n ? ts->d[lane] : 0.0, el = lane < n ? ts->e[lane] : 0.0;
        double an = fmax(fabs(dl), fabs(el));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) an = fmax(an, __shfl_xor(an, off, 64));
        double sc = 1.0;
        if (an > 0.0 && an < 1e300) { int ex; (void)frexp(an, &ex); sc = ldexp(1.0, -ex); }
        const double ds = dl * sc, es = el * sc;
        const double em = lane > 0 && lane < n ? ts->e[lane - 1] * sc : 0.0;
        const double rad = fabs(em) + (lane + 1 < n ? fabs(es) : 0.0);
        double gl = lane < n ? ds - rad : 1e300, gu = lane < n ? ds + rad : -1e300;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            gl = fmin(gl, __shfl_xor(gl, off, 64));
            gu = fmax(gu, __shfl_xor(gu, off, 64));
        }
        __syncthreads();   // every wave has read the unscaled d, e
        if (wave == 0) {
            if (lane < n) { ts->d[lane] = ds; ts->e[lane] = es; ts->de[lane] = make_double2(ds, em * em); }
            else ts->de[lane] = make_double2(8.0, 0.0);
            if (lane < 8) ts->de[kTriMax + lane] = make_double2(8.0, 0.0);
            if (lane == 0) {
                if (!(an < 1e300)) ts->bad = 1;   // NaN / inf input
                ts->scale = sc;
                const double bn = fmax(fabs(gl), fabs(gu));
                const double pad = 4.0 * n * 2.220446049250313e-16 * bn + 1e-290;
                ts->gl = gl - pad;
                ts->gu = gu + pad;
            }
        }
    }
    __syncthreads();
    TSTAMP(3);
    if (ts->bad) return false;
    // ---- (2) eigenvalues: eigenvalue i = tid / 8 is bracketed by 8 lanes (the first eight waves, two per SIMD).
    // 20 rounds: 9^20 = 2^63.4, the brackets end at rounding level (2^55.7 left 4e-12 of orthogonality defect on a
    // cond-1e4 matrix)
    if (tid < 8 * kTriMax && wave * 8 < n) {
        const int i = tid >> 3, t = tid & 7;
        const int nblocks = (n + 7) >> 3;
        double lo = ts->gl, hi = ts->gu;
        const double frac = (double)(t + 1) * (1.0 / 9.0);
        for (int it = 0; it < 20; ++it) {
            const double x = fma(hi - lo, frac, lo);
            const int c = sturm_count(ts, nblocks, x);
            double nlo = c <= i ? x : lo, nhi = c > i ? x : hi;
            nlo = hw_max_f64(nlo, mov_dpp64<0xB1>(nlo)); nhi = hw_min_f64(nhi, mov_dpp64<0xB1>(nhi));
            nlo = hw_max_f64(nlo, mov_dpp64<0x4E>(nlo)); nhi = hw_min_f64(nhi, mov_dpp64<0x4E>(nhi));
            nlo = hw_max_f64(nlo, mov_dpp64<0x141>(nlo)); nhi = hw_min_f64(nhi, mov_dpp64<0x141>(nhi));
            lo = nlo;
            hi = nhi;
        }
        if (i < n && t == 0) ts->lam[i] = 0.5 * (lo + hi);
    }
    __syncthreads();
    TSTAMP(4);
    // ---- (3) eigenvectors of T, eigenvalue i = lane: wave 0 runs the forward pivots q_j (into X[:, i]), wave 1
    // the backward pivots r_j (their reciprocals into A[:, i]); gamma_j = q_j - e_j^2 / r_{j+1} is smallest at the
    // twist k; z_k = 1, z_j = -(e_j / q_j) z_{j+1} above it (wave 0), z_{j+1} = -(e_j / r_{j+1}) z_j below (wave 1).
    // Every loop takes its rows eight at a time so that the LDS reads of a block are in flight together.
    const double tiny = 1e-290;
    double* Xc = X + (lane < n ? lane : 0);
    double* Ac = A + (lane < n ? lane : 0);
    const double lamb = ts->lam[lane < n ? lane : 0];
    if (wave == 0 && lane < n) {
        double q = ts->de[0].x - lamb;
        Xc[0] = q;
        for (int j0 = 1; j0 < n; j0 += 8) {
            double2 r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) r[u] = ts->de[j0 + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (j0 + u < n) {
                    if (fabs(q) < tiny) q = -tiny;
                    q = fma(-r[u].y, nr_rcp(q), r[u].x - lamb);
                    Xc[(j0 + u) * ld] = q;
                }
            }
        }
    } else if (wave == 1 && lane < n) {
        double r = ts->de[n - 1].x - lamb;
        if (fabs(r) < tiny) r = -tiny;
        double ir = nr_rcp(r);
        Ac[(n - 1) * ld] = ir;
        for (int j0 = n - 2; j0 >= 0; j0 -= 8) {
            double dj[8], e2[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 - u >= 0 ? j0 - u : 0;
                dj[u] = ts->de[j].x;
                e2[u] = ts->de[j + 1].y;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (j0 - u >= 0) {
                    r = fma(-e2[u], ir, dj[u] - lamb);
                    if (fabs(r) < tiny) r = -tiny;
                    ir = nr_rcp(r);
                    Ac[(j0 - u) * ld] = ir;
                }
            }
        }
    }
    __syncthreads();
    int kk = n - 1;
    if (wave < 2 && lane < n) {
        double gbest = fabs(Xc[(n - 1) * ld]);   // gamma_{n-1} = q_{n-1}
        for (int j0 = n - 2; j0 >= 0; j0 -= 8) {
            double g[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 - u >= 0 ? j0 - u : 0;
                g[u] = fabs(fma(-ts->de[j + 1].y, Ac[(j + 1) * ld], Xc[j * ld]));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j0 - u >= 0 && g[u] <= gbest) { gbest = g[u]; kk = j0 - u; }   // ties: the smallest index, in both waves alike
        }
    }
    __syncthreads();   // both waves have read every q before the z overwrite them
    if (wave == 0 && lane < n) {
        double z = 1.0, nrm2 = 1.0;
        for (int j0 = n - 2; j0 >= 0; j0 -= 8) {
            double f[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 - u >= 0 ? j0 - u : 0;
                double qq = Xc[j * ld];
                if (fabs(qq) < tiny) qq = -tiny;
                f[u] = -(ts->e[j] * nr_rcp(qq));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (j0 - u >= 0 && j0 - u < kk) {
                    z = f[u] * z;
                    Xc[(j0 - u) * ld] = z;
                    nrm2 = fma(z, z, nrm2);
                }
            }
        }
        Xc[kk * ld] = 1.0;
        ts->praw[lane] = nrm2;
    } else if (wave == 1 && lane < n) {
        double z = 1.0, nrm2 = 0.0;
        for (int j0 = 0; j0 + 1 < n; j0 += 8) {
            double f[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u + 1 < n ? j0 + u : 0;
                f[u] = -(ts->e[j] * Ac[(j + 1) * ld]);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (j0 + u + 1 < n && j0 + u >= kk) {
                    z = f[u] * z;
                    Xc[(j0 + u + 1) * ld] = z;
                    nrm2 = fma(z, z, nrm2);
                }
            }
        }
        ts->uq[lane] = nrm2;
    }
    __syncthreads();
    if (tid < n) {
        const double nrm2 = ts->praw[tid] + ts->uq[tid];
        const bool ok = nrm2 > 0.0 && nrm2 < 1e300;
        ts->inv[tid] = nr_rsqrt(ok ? nrm2 : 1.0);
        if (!ok) ts->bad = 1;
    }
    __syncthreads();
    TSTAMP(5);
    // ---- residuals |T x - lambda x|_inf of the normalised vectors and orthogonality of X (clustered / repeated
    // eigenvalues), every thread / wave its share; then (4) Z = Q X into A
    {
        double worst = 0.0;
        bool bad = false;
        for (int e0 = tid; e0 < n * n; e0 += nt) {
            const int j = e0 / n, i = e0 - j * n;
            const double x0 = X[j * ld + i];
            const double xm = j > 0 ? ts->e[j - 1] * X[(j - 1) * ld + i] : 0.0;
            const double xp = j + 1 < n ? ts->e[j] * X[(j + 1) * ld + i] : 0.0;
            const double tx = fma(ts->d[j] - ts->lam[i], x0, xm + xp) * ts->inv[i];
            bad = bad || !(fabs(tx) <= 1e-10);   // scaled matrix: |T| <= 1
        }
        const int jj = lane & 15, g = lane >> 4;
        const int tiles = (n + 15) >> 4;
        for (int t = wave; t < tiles * tiles; t += nt >> 6) {
            const int a0 = (t / tiles) * 16, b0 = (t - (t / tiles) * tiles) * 16;
            if (b0 > a0) continue;   // X'X is symmetric
            const bool aok = a0 + jj < n, bok = b0 + jj < n;
            const int ai = aok ? a0 + jj : 0, bi = bok ? b0 + jj : 0;
            const double sa = ts->inv[ai], sb = ts->inv[bi];
            tri_v4f64 acc = {0.0, 0.0, 0.0, 0.0};
            for (int k0 = 0; k0 < n; k0 += 16) {
                double a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool kok = k0 + 4 * u + g < n;
                    const int k = kok ? k0 + 4 * u + g : 0;
                    a[u] = X[k * ld + ai] * sa;
                    b[u] = X[k * ld + bi] * sb;
                    if (!(aok && kok)) a[u] = 0.0;
                    if (!(bok && kok)) b[u] = 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = a0 + g + 4 * r, col = b0 + jj;
                if (row < n && col < n) worst = fmax(worst, fabs(acc[r] - (row == col ? 1.0 : 0.0)));
            }
        }
        if (bad) worst = 1e300;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) worst = fmax(worst, __shfl_xor(worst, off, 64));
        if (lane == 0) ts->red[wave] = worst;
        __syncthreads();
        worst = 0.0;
        for (int w = 0; w < nt / 64; ++w) worst = fmax(worst, ts->red[w]);
        TSTAMP(6);
        if (ts->bad || !(worst <= orth_tol)) return false;
    }
    mfma_mm<true, false>(A, Q, X, n, n, n, ld, ts->inv);   // Q holds Q'
    if (tid < n) ts->lam[tid] /= ts->scale;
    __syncthreads();
    TSTAMP(7);
    return true;
}

struct TicaWork {  // global scratch: four n*ld matrices, then ev[n], mean[n], isc[n], order[n]
    double *A, *V, *B1, *B2, *ev, *mean, *isc;
    int* order;
};

// moments = [M00 F*F][M0t F*F][sx F][sy F][T]   (centred by shift, unscaled)
// scale   = per-feature divisor applied to the centred data (NULL -> 1)
// lds_mats: how many of {A, V, B1, B2} live in LDS (4, 2 or 0)
// (template, not a runtime flag: a pointer that may be LDS or global forces slow FLAT accesses)
template <int lds_mats>
__global__ __launch_bounds__(kEigThreads) void tica_solve_kernel(
    const double* __restrict__ mom, const double* __restrict__ scale, int n, int ld, double epsilon, int kinetic_map,
    TicaWork wk, double* __restrict__ out_eig, double* __restrict__ out_W, double* __restrict__ out_mean,
    int* __restrict__ out_rank) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ JacobiShared sh;
    __shared__ TriShared ts;
    const int tid = threadIdx.x, nt = blockDim.x;
    double* lds = reinterpret_cast<double*>(smem_raw);
    const size_t mat = (size_t)n * ld;
    double *A, *V, *B1, *B2;
    if constexpr (lds_mats >= 2) { A = lds; V = lds + mat; } else { A = wk.A; V = wk.V; }
    if constexpr (lds_mats >= 4) { B1 = lds + 2 * mat; B2 = lds + 3 * mat; } else { B1 = wk.B1; B2 = wk.B2; }

    const double* M00 = mom;
    const double* M0t = mom + (size_t)n * n;
    const double* sx = M0t + (size_t)n * n;
    const double* sy = sx + n;
    const double T = sy[n];
    const double w = 2.0 * T;
    KSTAMP_INIT
    if (!(T > 0.0)) {
        if (tid == 0) *out_rank = 0;
        for (int i = tid; i < n; i += nt) { out_eig[i] = 0.0; out_mean[i] = 0.0; }
        for (int i = tid; i < n * n; i += nt) out_W[i] = 0.0;
        return;
    }
    // 1 / scale and the means: in the LDS when they fit the solver's arrays (they are read 2 n times each below)
    double* iscp = (lds_mats == 4 && n <= kTriMax) ? ts.lam : wk.isc;
    double* meanp = (lds_mats == 4 && n <= kTriMax) ? ts.inv : wk.mean;
    for (int i = tid; i < n; i += nt) {
        const double is = scale ? 1.0 / scale[i] : 1.0;
        iscp[i] = is;
        const double m = (sx[i] + sy[i]) / w * is;
        meanp[i] = m;
        out_mean[i] = m;
    }
    __syncthreads();
    auto build_cov = [&]() {
        for (int e = tid; e < n * n; e += nt) {
            const int i = e / n, j = e - i * n;
            const double ss = iscp[i] * iscp[j];
            const double mm = meanp[i] * meanp[j];
            A[i * ld + j] = 0.5 * (M00[e] + M00[j * n + i]) / w * ss - mm;   // C00
            B1[i * ld + j] = (M0t[e] + M0t[j * n + i]) / w * ss - mm;        // C0t
        }
        __syncthreads();
    };
    build_cov();
    KSTAMP(8);

    // ---- whitening L with L' C00 L = I -------------------------------------------------
    // deeptime's spd_inv_split keeps the eigen-directions of C00 with |s| >= epsilon.  When
    // ALL of them qualify (C00 - epsilon I positive definite: tested by a factorisation attempt)
    // the TICA eigenpairs do not depend on which whitening is used -- they solve
    // C0t r = lambda C00 r -- so L = chol(C00)^-T replaces the first eigensolve.
    // Rank-deficient C00 takes the eigen path below, as before.
    constexpr bool kFused = lds_mats == 4;   // fused LDL' + inverse (one barrier per column) and the tridiagonal solver
    bool full_rank;
    if (kFused && n <= kTriMax) {
        // B2 = whitening W (upper triangular, W' C00 W = I); C00 stays in A.  C00^-1 = W W', so lambda_min(C00) >=
        // 1 / trace(W W') = 1 / ||W||_F^2: when that already clears epsilon every eigen-direction is kept and the
        // elimination of the probe C00 - epsilon I (a third of the phase's instructions) is not needed
        full_rank = ldl_whiten_registers(A, B2, n, epsilon, &ts, 1);
        if (full_rank) {
            double s = 0.0;
            for (int e = tid; e < n * n; e += nt) {
                const double wv = B2[(e / n) * ld + (e % n)];
                s = fma(wv, wv, s);
            }
            s = block_sum(s, &sh);
            if (!(s * epsilon * (1.0 + 1e-9) <= 1.0)) full_rank = ldl_whiten_registers(A, nullptr, n, epsilon, &ts, 2);
        }
    } else {
        for (int e = tid; e < n * n; e += nt) {
            const int i = e / n, j = e - i * n;
            B2[i * ld + j] = A[i * ld + j] - (i == j ? epsilon : 0.0);
            V[i * ld + j] = A[i * ld + j];
        }
        __syncthreads();
        full_rank = cholesky_lower_pair(B2, V, n, ld, &sh);   // probe on C00 - eps I, factor C00 = G G'
        if (full_rank) {
            lower_inverse(V, A, n, ld);         // A = G^-1 (lower)
            for (int e = tid; e < n * n; e += nt) {
                const int i = e / n, j = e - i * n;
                B2[i * ld + j] = j >= i ? A[j * ld + i] : 0.0;  // L = G^-T (upper)
            }
            __syncthreads();
        } else {
            __syncthreads();
            build_cov();
        }
    }
    KSTAMP(9);
    int rank;
    if (full_rank) {
        rank = n;
        if (tid == 0) *out_rank = n;
    } else {
        // ---- spd_inv_split(C00): eigh, sort by |ev| desc, cut at epsilon, canonical signs ----
        jacobi_eigh(A, V, n, ld, &sh, 40);
        for (int i = tid; i < n; i += nt) wk.ev[i] = A[i * ld + i];
        __syncthreads();
        sort_desc_abs(wk.ev, n, wk.order);
        if (tid == 0) {
            double evmin = wk.ev[0];
            for (int i = 1; i < n; ++i) evmin = fmin(evmin, wk.ev[i]);
            double eps = epsilon;
            if (evmin < 0.0) eps = fmax(eps, -evmin + 1e-16);
            int rank = 0;
            for (int i = 0; i < n; ++i) rank += fabs(wk.ev[i]) >= eps;
            sh.ibc[0] = rank;
            *out_rank = rank;
        }
        __syncthreads();
        rank = sh.ibc[0];
        if (rank == 0) {
            for (int i = tid; i < n; i += nt) out_eig[i] = 0.0;
            for (int i = tid; i < n * n; i += nt) out_W[i] = 0.0;
            return;
        }
        // L (in B2) = V[:, order[:rank]] with canonical signs, columns scaled by 1/sqrt(s)
        for (int e = tid; e < n * rank; e += nt) {
            const int i = e / rank, j = e - i * rank;
            B2[i * ld + j] = V[i * ld + wk.order[j]];
        }
        __syncthreads();
        canonical_signs(B2, n, rank, ld);
        for (int e = tid; e < n * rank; e += nt) {
            const int i = e / rank, j = e - i * rank;
            B2[i * ld + j] /= sqrt(wk.ev[wk.order[j]]);
        }
        __syncthreads();
    }
    // ---- Ct = L' C0t L: A <- C0t L, V <- L' A, A <- sym(V) ----
    mfma_mm<false, false>(A, B1, B2, n, rank, n, ld);
    mfma_mm<true, false>(V, B2, A, rank, rank, n, ld);
    for (int e = tid; e < rank * rank; e += nt) {
        const int i = e / rank, j = e - i * rank;
        A[i * ld + j] = 0.5 * (V[i * ld + j] + V[j * ld + i]);
    }
    __syncthreads();
    KSTAMP(10);
    // eigenpairs of the whitened C0t: tridiagonal solver (fallback: Jacobi on the saved copy)
    const double* Vec = V;
    bool fast = false;
    if (kFused && rank <= kTriMax) {
        for (int e = tid; e < rank * rank; e += nt) wk.A[e] = A[(e / rank) * ld + (e % rank)];
        __syncthreads();
        fast = tridiag_eigh(A, V, B1, rank, ld, &ts, 1e-11);
        if (fast) {
            for (int i = tid; i < rank; i += nt) wk.ev[i] = ts.lam[i];
            Vec = A;
        } else {
            __syncthreads();
            for (int e = tid; e < rank * rank; e += nt) A[(e / rank) * ld + (e % rank)] = wk.A[e];
        }
        __syncthreads();
    }
    if (!fast) {
        jacobi_eigh(A, V, rank, ld, &sh, 40);
        for (int i = tid; i < rank; i += nt) wk.ev[i] = A[i * ld + i];
    }
    __syncthreads();
    KSTAMP(11);
    if (kFused && n <= kTriMax) {
        // ---- the same tail with everything small in the LDS: eigenvalues, their order (|ev| descending, stable), the
        // sign of every column of R = L Rt (its largest-magnitude entry, first occurrence, made positive); the signs
        // and the kinetic-map factors are applied on the way out instead of in place
        double* evl = ts.praw;      // eigenvalues in solver order
        double* sgn = ts.uq;        // +-1 per solver column
        int* ord = sh.p;            // ord[j] = solver column of the j-th largest |ev|
        if (tid < rank) evl[tid] = wk.ev[tid];
        __syncthreads();
        if (tid < rank) {
            const double a = fabs(evl[tid]);
            int r = 0;
            for (int j = 0; j < rank; ++j) {
                const double b = fabs(evl[j]);
                r += (b > a) || (b == a && j < tid);
            }
            ord[r] = tid;
        }
        mfma_mm<false, false>(B1, B2, Vec, n, rank, rank, ld);   // columns still in solver order (barrier inside)
        {
            const int lane = tid & 63, wave = tid >> 6;
            for (int j = wave; j < rank; j += nt >> 6) {          // one wave per column, lane = row
                double best = lane < n ? fabs(B1[lane * ld + j]) : -1.0;
                int bi = lane;
                double val = lane < n ? B1[lane * ld + j] : 0.0;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const double ob = __shfl_xor(best, off, 64), ov = __shfl_xor(val, off, 64);
                    const int oi = __shfl_xor(bi, off, 64);
                    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; val = ov; }
                }
                if (lane == 0) sgn[j] = val < 0.0 ? -1.0 : 1.0;
            }
        }
        __syncthreads();
        for (int e = tid; e < n * n; e += nt) {
            const int i = e / n, j = e - i * n;
            double v = 0.0;
            if (j < rank) {
                const int c = ord[j];
                v = B1[i * ld + c] * sgn[c];
                if (kinetic_map) v *= evl[c];
            }
            out_W[e] = v;
        }
        for (int j = tid; j < n; j += nt) out_eig[j] = j < rank ? evl[ord[j]] : 0.0;
        KSTAMP(12);
        return;
    }
    sort_desc_abs(wk.ev, rank, wk.order);
    // ---- R = L Rt (sorted), canonical signs, kinetic map ----
    mfma_mm<false, false>(B1, B2, Vec, n, rank, rank, ld);   // columns still in solver order
    canonical_signs(B1, n, rank, ld);
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e - i * n;
        double v = 0.0;
        if (j < rank) {
            v = B1[i * ld + wk.order[j]];
            if (kinetic_map) v *= wk.ev[wk.order[j]];
        }
        out_W[e] = v;
    }
    for (int j = tid; j < n; j += nt) out_eig[j] = j < rank ? wk.ev[wk.order[j]] : 0.0;
    KSTAMP(12);
}

// Plain symmetric eigendecomposition (ascending eigenvalues), for tests and the
// reversible MSM path.
template <bool use_lds>
__global__ __launch_bounds__(kEigThreads) void eigh_kernel(const double* __restrict__ Ain, int n, int ld,
                                                          double* gA, double* gV, int* order,
                                                          double* __restrict__ out_w, double* __restrict__ out_v,
                                                          int* __restrict__ out_sweeps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ JacobiShared sh;
    const int tid = threadIdx.x, nt = blockDim.x;
    double *A, *V;
    if constexpr (use_lds) { A = reinterpret_cast<double*>(smem_raw); V = A + (size_t)n * ld; }
    else { A = gA; V = gV; }
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e - i * n;
        A[i * ld + j] = 0.5 * (Ain[e] + Ain[j * n + i]);
    }
    __syncthreads();
    if constexpr (use_lds) {
        if (n <= kTriMax) {   // three LDS matrices were requested for this size (jacobi_lds_bytes)
            __shared__ TriShared ts;
            double* X = V + (size_t)n * ld;
            for (int e = tid; e < n * n; e += nt) gA[e] = A[(e / n) * ld + (e % n)];
            __syncthreads();
            if (tridiag_eigh(A, V, X, n, ld, &ts, 1e-12)) {
                for (int j = tid; j < n; j += nt) out_w[j] = ts.lam[j];
                if (out_v)
                    for (int e = tid; e < n * n; e += nt) out_v[e] = A[(e / n) * ld + (e % n)];
                if (tid == 0 && out_sweeps) *out_sweeps = 0;
                return;
            }
            __syncthreads();
            for (int e = tid; e < n * n; e += nt) A[(e / n) * ld + (e % n)] = gA[e];
            __syncthreads();
        }
    }
    const int sweeps = jacobi_eigh(A, V, n, ld, &sh, 40);
    // ascending order by value
    for (int i = tid; i < n; i += nt) {
        const double a = A[i * ld + i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const double b = A[j * ld + j];
            rank += (b < a) || (b == a && j < i);
        }
        order[rank] = i;
    }
    __syncthreads();
    for (int j = tid; j < n; j += nt) out_w[j] = A[order[j] * ld + order[j]];
    if (out_v)
        for (int e = tid; e < n * n; e += nt) {
            const int i = e / n, j = e - i * n;
            out_v[e] = V[i * ld + order[j]];
        }
    if (tid == 0 && out_sweeps) *out_sweeps = sweeps;
}

// Eigenvalue estimator of the reference's DeepTICA trainer (_estimate_top_eigenvalues,
// S/features/deeptica/core/trainer_api.py:632-656) from ONE-SIDED raw moments (msm_lagged_moments_onesided):
//   C0 = (M00 - sx sx'/T) / max(1, T-1),  Ct = (M0t - sx sy'/T) / max(1, T-1)      (separate means of y_t, y_tau)
//   eigh(sym C0), eigenvalues clipped at `clip`; S = V diag(w^-1/2) V';  eigvalsh(sym(S Ct S')), descending.
// Four n x n matrices P0..P3 (LDS when they fit, else global), eigensolves by tridiag_eigh with the Jacobi fallback.
template <bool use_lds>
__global__ __launch_bounds__(kEigThreads) void onesided_eig_kernel(const double* __restrict__ mom, int n, int ld,
                                                                  double clip, double* gP, double* gbak, double* gw,
                                                                  int* order, double* __restrict__ out_eig) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ JacobiShared sh;
    __shared__ TriShared ts;
    const int tid = threadIdx.x, nt = blockDim.x;
    const size_t mat = (size_t)n * ld;
    double* P0 = use_lds ? reinterpret_cast<double*>(smem_raw) : gP;
    double *P1 = P0 + mat, *P2 = P0 + 2 * mat, *P3 = P0 + 3 * mat;
    const double* M00 = mom;
    const double* M0t = mom + (size_t)n * n;
    const double* sx = M0t + (size_t)n * n;
    const double* sy = sx + n;
    const double T = sy[n];
    if (!(T > 0.0)) {
        for (int i = tid; i < n; i += nt) out_eig[i] = 0.0;
        return;
    }
    const double den = T - 1.0 > 1.0 ? T - 1.0 : 1.0;
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e - i * n;
        const double c0 = 0.5 * (M00[e] + M00[j * n + i]) - sx[i] * sx[j] / T;
        P0[i * ld + j] = c0 / den;
        P1[i * ld + j] = (M0t[e] - sx[i] * sy[j] / T) / den;
    }
    __syncthreads();
    // eigenpairs of the symmetric matrix in P0 (work P2, P3): eigenvalues to gw, eigenvectors to *vec
    auto eigh = [&](const double*& vec) {
        bool fast = false;
        if (use_lds && n <= kTriMax) {
            for (int e = tid; e < n * n; e += nt) gbak[e] = P0[(e / n) * ld + (e % n)];
            __syncthreads();
            fast = tridiag_eigh(P0, P2, P3, n, ld, &ts, 1e-12);
            if (fast) {
                for (int i = tid; i < n; i += nt) gw[i] = ts.lam[i];
                vec = P0;
            } else {
                __syncthreads();
                for (int e = tid; e < n * n; e += nt) P0[(e / n) * ld + (e % n)] = gbak[e];
            }
            __syncthreads();
        }
        if (!fast) {
            jacobi_eigh(P0, P2, n, ld, &sh, 40);
            for (int i = tid; i < n; i += nt) gw[i] = P0[i * ld + i];
            vec = P2;
            __syncthreads();
        }
    };
    const double* vec = nullptr;
    eigh(vec);
    // P3 = V diag(w^-1/4): S = P3 P3'
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e - i * n;
        const double wj = gw[j] > clip ? gw[j] : clip;
        P3[i * ld + j] = vec[i * ld + j] / sqrt(sqrt(wj));
    }
    __syncthreads();
    mfma_mm<false, true>(P0, P3, P3, n, n, n, ld);   // S -> P0
    mfma_mm<false, false>(P2, P0, P1, n, n, n, ld);   // S Ct
    mfma_mm<false, false>(P3, P2, P0, n, n, n, ld);   // (S Ct) S'   (S is symmetric)
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e - i * n;
        P0[i * ld + j] = 0.5 * (P3[i * ld + j] + P3[j * ld + i]);
    }
    __syncthreads();
    eigh(vec);
    // descending order
    for (int i = tid; i < n; i += nt) {
        const double a = gw[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const double b = gw[j];
            rank += (b > a) || (b == a && j < i);
        }
        order[rank] = i;
    }
    __syncthreads();
    for (int j = tid; j < n; j += nt) out_eig[j] = gw[order[j]];
}

size_t jacobi_lds_bytes(int n, int ld) { return (size_t)(n <= kTriMax ? 3 : 2) * n * ld * sizeof(double); }

}  // namespace

extern "C" {

msm_status msm_tica_solve(msm_ctx* ctx, const double* d_moments, const double* d_scale, int F, double epsilon,
                          int kinetic_map, double* d_eigvals, double* d_coeffs, double* d_mean, int* d_rank) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, F >= 1 && F <= 2 * kMaxPairs, "msm_tica_solve: need 1 <= F <= %d (got %d)", 2 * kMaxPairs, F);
    MSM_REQUIRE(ctx, epsilon >= 0.0, "msm_tica_solve: epsilon must be >= 0");
    MSM_REQUIRE(ctx, d_moments && d_eigvals && d_coeffs && d_mean && d_rank, "msm_tica_solve: NULL pointer");
    const int ld = F <= kTriMax ? kTriLd : (F | 1);  // odd stride; the fixed one of the tridiagonal solver for F <= 64
    const size_t mat = (size_t)F * ld;
    const size_t need = (4 * mat + 3 * F) * sizeof(double) + (size_t)F * sizeof(int) + 64;
    msm_status rs = msm_reserve_scratch(ctx, need);
    if (rs != MSM_OK) return rs;
    double* base = (double*)ctx->scratch;
    TicaWork wk;
    wk.A = base; wk.V = base + mat; wk.B1 = base + 2 * mat; wk.B2 = base + 3 * mat;
    wk.ev = base + 4 * mat; wk.mean = wk.ev + F; wk.isc = wk.mean + F; wk.order = (int*)(wk.isc + F);
    const size_t lds_budget = 150 * 1024;
    int lds_mats = 0;
    if (4 * mat * sizeof(double) <= lds_budget) lds_mats = 4;
    else if (2 * mat * sizeof(double) <= lds_budget) lds_mats = 2;
    const size_t lds = (size_t)lds_mats * mat * sizeof(double);
    auto kern = lds_mats == 4 ? tica_solve_kernel<4> : (lds_mats == 2 ? tica_solve_kernel<2> : tica_solve_kernel<0>);
    if (lds > 48 * 1024)
        MSM_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(1), dim3(kEigThreads), lds, ctx->stream, d_moments, d_scale, F, ld, epsilon,
                       kinetic_map, wk, d_eigvals, d_coeffs, d_mean, d_rank);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_onesided_tica_eigenvalues(msm_ctx* ctx, const double* d_moments, int F, double clip, double* d_eigvals) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, F >= 1 && F <= 2 * kMaxPairs, "msm_onesided_tica_eigenvalues: need 1 <= F <= %d (got %d)",
                2 * kMaxPairs, F);
    MSM_REQUIRE(ctx, d_moments && d_eigvals && clip > 0.0, "msm_onesided_tica_eigenvalues: bad arguments");
    const int ld = F <= kTriMax ? kTriLd : (F | 1);
    const size_t mat = (size_t)F * ld;
    const size_t lds = 4 * mat * sizeof(double);
    const bool use_lds = lds <= 140 * 1024;
    msm_status rs = msm_reserve_scratch(ctx, ((use_lds ? 0 : 4 * mat) + (size_t)F * F + F) * sizeof(double) +
                                                 (size_t)F * sizeof(int) + 64);
    if (rs != MSM_OK) return rs;
    double* gP = (double*)ctx->scratch;
    double* gbak = gP + (use_lds ? 0 : 4 * mat);
    double* gw = gbak + (size_t)F * F;
    int* order = (int*)(gw + F);
    auto kern = use_lds ? onesided_eig_kernel<true> : onesided_eig_kernel<false>;
    if (use_lds && lds > 48 * 1024)
        MSM_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(1), dim3(kEigThreads), use_lds ? lds : 0, ctx->stream, d_moments, F, ld, clip, gP, gbak,
                       gw, order, d_eigvals);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_eigh(msm_ctx* ctx, const double* d_a, int n, double* d_w, double* d_v, int* d_sweeps) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 1 && n <= 2 * kMaxPairs, "msm_eigh: need 1 <= n <= %d (got %d)", 2 * kMaxPairs, n);
    MSM_REQUIRE(ctx, d_a && d_w, "msm_eigh: NULL pointer");
    const int ld = n <= kTriMax ? kTriLd : (n | 1);
    const size_t mat = (size_t)n * ld;
    msm_status rs = msm_reserve_scratch(ctx, 2 * mat * sizeof(double) + (size_t)n * sizeof(int) + 64);
    if (rs != MSM_OK) return rs;
    double* gA = (double*)ctx->scratch;
    double* gV = gA + mat;
    int* order = (int*)(gV + mat);
    const size_t lds = jacobi_lds_bytes(n, ld);
    const int use_lds = lds <= 140 * 1024;
    auto kern = use_lds ? eigh_kernel<true> : eigh_kernel<false>;
    if (use_lds && lds > 48 * 1024)
        MSM_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(1), dim3(kEigThreads), use_lds ? lds : 0, ctx->stream, d_a, n, ld, gA, gV, order,
                       d_w, d_v, d_sweeps);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

}  // extern "C"
