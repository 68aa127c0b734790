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

// In-place Cholesky M = G G' (lower triangle of M becomes G).  Returns false (uniformly) as
// soon as a pivot is not positive.  Block-parallel right-looking form, 3 barriers per column.
__device__ bool cholesky_lower(double* M, int n, int ld, JacobiShared* sh) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int j = 0; j < n; ++j) {
        if (tid == 0) {
            const double dj = M[j * ld + j];
            sh->ibc[1] = dj > 0.0 ? 1 : 0;
            if (dj > 0.0) M[j * ld + j] = sqrt(dj);
        }
        __syncthreads();
        if (!sh->ibc[1]) return false;
        const double gjj = M[j * ld + j];
        for (int i = j + 1 + tid; i < n; i += nt) M[i * ld + j] /= gjj;
        __syncthreads();
        const int m = n - j - 1;
        for (int e = tid; e < m * m; e += nt) {
            const int a = e / m, b = e - a * m;  // trailing (j+1+a, j+1+b), lower part only
            if (b <= a) M[(j + 1 + a) * ld + (j + 1 + b)] -= M[(j + 1 + a) * ld + j] * M[(j + 1 + b) * ld + j];
        }
        __syncthreads();
    }
    return true;
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

struct TicaWork {  // global scratch: four n*ld matrices, then ev[n], mean[n], isc[n], order[n]
    double *A, *V, *B1, *B2, *ev, *mean, *isc;
    int* order;
};

// C[i][j] = sum_k opA(i,k) * B[k][j]   (opA = A or A'), i < rows, j < cols, k < inner
__device__ void small_mm(double* C, const double* A, bool transA, const double* B, int rows, int cols, int inner,
                         int ld) {
    // four output elements per trip: four independent FMA chains hide the fp64 latency
    const int total = rows * cols, nt = blockDim.x;
    for (int e0 = threadIdx.x; e0 < total; e0 += 4 * nt) {
        int ia[4], ja[4];
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = min(e0 + u * nt, total - 1);
            ia[u] = e / cols;
            ja[u] = e - ia[u] * cols;
        }
        for (int k = 0; k < inner; ++k) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double av = transA ? A[k * ld + ia[u]] : A[ia[u] * ld + k];
                acc[u] = fma(av, B[k * ld + ja[u]], acc[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (e0 + u * nt < total) C[ia[u] * ld + ja[u]] = acc[u];
    }
    __syncthreads();
}

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
    if (!(T > 0.0)) {
        if (tid == 0) *out_rank = 0;
        for (int i = tid; i < n; i += nt) { out_eig[i] = 0.0; out_mean[i] = 0.0; }
        for (int i = tid; i < n * n; i += nt) out_W[i] = 0.0;
        return;
    }
    for (int i = tid; i < n; i += nt) {
        const double is = scale ? 1.0 / scale[i] : 1.0;
        wk.isc[i] = is;
        wk.mean[i] = (sx[i] + sy[i]) / w * is;
        out_mean[i] = wk.mean[i];
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e - i * n;
        const double ss = wk.isc[i] * wk.isc[j];
        const double mm = wk.mean[i] * wk.mean[j];
        A[i * ld + j] = 0.5 * (M00[e] + M00[j * n + i]) / w * ss - mm;   // C00
        B1[i * ld + j] = (M0t[e] + M0t[j * n + i]) / w * ss - mm;        // C0t
    }
    __syncthreads();

    // ---- whitening L with L' C00 L = I -------------------------------------------------
    // deeptime's spd_inv_split keeps the eigen-directions of C00 with |s| >= epsilon.  When
    // ALL of them qualify (C00 - epsilon I positive definite: tested by a Cholesky attempt)
    // the TICA eigenpairs do not depend on which whitening is used -- they solve
    // C0t r = lambda C00 r -- so L = chol(C00)^-T replaces the first Jacobi eigensolve
    // (~15x cheaper).  Rank-deficient C00 takes the eigen path below, as before.
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e - i * n;
        B2[i * ld + j] = A[i * ld + j] - (i == j ? epsilon : 0.0);
        V[i * ld + j] = A[i * ld + j];
    }
    __syncthreads();
    const bool full_rank = cholesky_lower_pair(B2, V, n, ld, &sh);   // probe on C00 - eps I, factor C00 = G G'
    int rank;
    if (full_rank) {
        lower_inverse(V, A, n, ld);         // A = G^-1 (lower)
        for (int e = tid; e < n * n; e += nt) {
            const int i = e / n, j = e - i * n;
            B2[i * ld + j] = j >= i ? A[j * ld + i] : 0.0;  // L = G^-T (upper)
        }
        __syncthreads();
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
    small_mm(A, B1, false, B2, n, rank, n, ld);
    small_mm(V, B2, true, A, rank, rank, n, ld);
    for (int e = tid; e < rank * rank; e += nt) {
        const int i = e / rank, j = e - i * rank;
        A[i * ld + j] = 0.5 * (V[i * ld + j] + V[j * ld + i]);
    }
    __syncthreads();
    jacobi_eigh(A, V, rank, ld, &sh, 40);
    for (int i = tid; i < rank; i += nt) wk.ev[i] = A[i * ld + i];
    __syncthreads();
    sort_desc_abs(wk.ev, rank, wk.order);
    // ---- R = L Rt (sorted), canonical signs, kinetic map ----
    for (int e = tid; e < n * rank; e += nt) {
        const int i = e / rank, j = e - i * rank;
        const int src = wk.order[j];
        double a = 0.0;
        for (int k = 0; k < rank; ++k) a = fma(B2[i * ld + k], V[k * ld + src], a);
        B1[i * ld + j] = a;
    }
    __syncthreads();
    canonical_signs(B1, n, rank, ld);
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e - i * n;
        double v = 0.0;
        if (j < rank) {
            v = B1[i * ld + j];
            if (kinetic_map) v *= wk.ev[wk.order[j]];
        }
        out_W[e] = v;
    }
    for (int j = tid; j < n; j += nt) out_eig[j] = j < rank ? wk.ev[wk.order[j]] : 0.0;
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

size_t jacobi_lds_bytes(int n, int ld) { return (size_t)2 * n * ld * sizeof(double); }

}  // namespace

extern "C" {

msm_status msm_tica_solve(msm_ctx* ctx, const double* d_moments, const double* d_scale, int F, double epsilon,
                          int kinetic_map, double* d_eigvals, double* d_coeffs, double* d_mean, int* d_rank) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, F >= 1 && F <= 2 * kMaxPairs, "msm_tica_solve: need 1 <= F <= %d (got %d)", 2 * kMaxPairs, F);
    MSM_REQUIRE(ctx, epsilon >= 0.0, "msm_tica_solve: epsilon must be >= 0");
    MSM_REQUIRE(ctx, d_moments && d_eigvals && d_coeffs && d_mean && d_rank, "msm_tica_solve: NULL pointer");
    const int ld = F | 1;  // odd stride
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

msm_status msm_eigh(msm_ctx* ctx, const double* d_a, int n, double* d_w, double* d_v, int* d_sweeps) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 1 && n <= 2 * kMaxPairs, "msm_eigh: need 1 <= n <= %d (got %d)", 2 * kMaxPairs, n);
    MSM_REQUIRE(ctx, d_a && d_w, "msm_eigh: NULL pointer");
    const int ld = n | 1;
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
