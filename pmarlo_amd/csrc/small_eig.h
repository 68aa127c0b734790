// Eigenvalues (and one eigenvector) of a small dense real NON-symmetric matrix:
// elimination to Hessenberg form + Francis double-shift QR (the classical EISPACK
// elmhes / hqr pair).  Used on the p x p Rayleigh-Ritz matrix of the subspace
// iteration (p <= 32), i.e. a latency-bound epilogue; plain scalar code that
// compiles for host (unit-tested on the CPU) and device.
#pragma once
#include <math.h>

#ifdef __HIPCC__
#define SE_HD __host__ __device__
#else
#define SE_HD
#endif

namespace small_eig {

SE_HD inline double sign_of(double a, double b) { return b >= 0.0 ? fabs(a) : -fabs(a); }

// a: n x n row-major with row stride ld (destroyed).  Returns 0 on success, >0 = the
// index at which 60 QR iterations did not converge.
SE_HD inline int eigenvalues(double* a, int n, int ld, double* wr, double* wi) {
#define A_(i, j) a[(i) * ld + (j)]
    // ---- elmhes ----
    for (int m = 1; m < n - 1; ++m) {
        double x = 0.0;
        int i = m;
        for (int j = m; j < n; ++j)
            if (fabs(A_(j, m - 1)) > fabs(x)) { x = A_(j, m - 1); i = j; }
        if (i != m) {
            for (int j = m - 1; j < n; ++j) { double t = A_(i, j); A_(i, j) = A_(m, j); A_(m, j) = t; }
            for (int j = 0; j < n; ++j) { double t = A_(j, i); A_(j, i) = A_(j, m); A_(j, m) = t; }
        }
        if (x != 0.0) {
            for (i = m + 1; i < n; ++i) {
                double y = A_(i, m - 1);
                if (y != 0.0) {
                    y /= x;
                    A_(i, m - 1) = y;
                    for (int j = m; j < n; ++j) A_(i, j) -= y * A_(m, j);
                    for (int j = 0; j < n; ++j) A_(j, m) += y * A_(j, i);
                }
            }
        }
    }
    for (int i = 2; i < n; ++i)
        for (int j = 0; j < i - 1; ++j) A_(i, j) = 0.0;
    // ---- hqr ----
    double anorm = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = (i > 0 ? i - 1 : 0); j < n; ++j) anorm += fabs(A_(i, j));
    int nn = n - 1;
    double t = 0.0, p = 0.0, q = 0.0, r = 0.0;
    while (nn >= 0) {
        int its = 0, l;
        do {
            for (l = nn; l >= 1; --l) {
                double s = fabs(A_(l - 1, l - 1)) + fabs(A_(l, l));
                if (s == 0.0) s = anorm;
                if (fabs(A_(l, l - 1)) + s == s) { A_(l, l - 1) = 0.0; break; }
            }
            double x = A_(nn, nn);
            if (l == nn) {
                wr[nn] = x + t;
                wi[nn--] = 0.0;
            } else {
                double y = A_(nn - 1, nn - 1);
                double w = A_(nn, nn - 1) * A_(nn - 1, nn);
                if (l == nn - 1) {
                    p = 0.5 * (y - x);
                    q = p * p + w;
                    double z = sqrt(fabs(q));
                    x += t;
                    if (q >= 0.0) {
                        z = p + sign_of(z, p);
                        wr[nn - 1] = wr[nn] = x + z;
                        if (z != 0.0) wr[nn] = x - w / z;
                        wi[nn - 1] = wi[nn] = 0.0;
                    } else {
                        wr[nn - 1] = wr[nn] = x + p;
                        wi[nn] = z;
                        wi[nn - 1] = -z;
                    }
                    nn -= 2;
                } else {
                    if (its == 60) return nn + 1;
                    if (its == 10 || its == 20 || its == 30 || its == 40) {
                        t += x;
                        for (int i = 0; i <= nn; ++i) A_(i, i) -= x;
                        const double s = fabs(A_(nn, nn - 1)) + fabs(A_(nn - 1, nn - 2));
                        y = x = 0.75 * s;
                        w = -0.4375 * s * s;
                    }
                    ++its;
                    int m;
                    for (m = nn - 2; m >= l; --m) {
                        const double z = A_(m, m);
                        r = x - z;
                        double s = y - z;
                        p = (r * s - w) / A_(m + 1, m) + A_(m, m + 1);
                        q = A_(m + 1, m + 1) - z - r - s;
                        r = A_(m + 2, m + 1);
                        s = fabs(p) + fabs(q) + fabs(r);
                        p /= s; q /= s; r /= s;
                        if (m == l) break;
                        const double u = fabs(A_(m, m - 1)) * (fabs(q) + fabs(r));
                        const double v = fabs(p) * (fabs(A_(m - 1, m - 1)) + fabs(z) + fabs(A_(m + 1, m + 1)));
                        if (u + v == v) break;
                    }
                    for (int i = m + 2; i <= nn; ++i) {
                        A_(i, i - 2) = 0.0;
                        if (i != m + 2) A_(i, i - 3) = 0.0;
                    }
                    for (int k = m; k <= nn - 1; ++k) {
                        if (k != m) {
                            p = A_(k, k - 1);
                            q = A_(k + 1, k - 1);
                            r = 0.0;
                            if (k != nn - 1) r = A_(k + 2, k - 1);
                            if ((x = fabs(p) + fabs(q) + fabs(r)) != 0.0) { p /= x; q /= x; r /= x; }
                        }
                        const double s = sign_of(sqrt(p * p + q * q + r * r), p);
                        if (s != 0.0) {
                            if (k == m) {
                                if (l != m) A_(k, k - 1) = -A_(k, k - 1);
                            } else {
                                A_(k, k - 1) = -s * x;
                            }
                            p += s;
                            x = p / s;
                            y = q / s;
                            const double z = r / s;
                            q /= p;
                            r /= p;
                            for (int j = k; j <= nn; ++j) {
                                p = A_(k, j) + q * A_(k + 1, j);
                                if (k != nn - 1) { p += r * A_(k + 2, j); A_(k + 2, j) -= p * z; }
                                A_(k + 1, j) -= p * y;
                                A_(k, j) -= p * x;
                            }
                            const int mmin = nn < k + 3 ? nn : k + 3;
                            for (int i = l; i <= mmin; ++i) {
                                p = x * A_(i, k) + y * A_(i, k + 1);
                                if (k != nn - 1) { p += z * A_(i, k + 2); A_(i, k + 2) -= p * r; }
                                A_(i, k + 1) -= p * q;
                                A_(i, k) -= p;
                            }
                        }
                    }
                }
            }
        } while (l < nn - 1);
    }
#undef A_
    return 0;
}

#ifdef __HIPCC__
// The same algorithm executed by ONE 64-lane wave (n <= 64): the scalar control flow runs redundantly on
// every lane, the O(n) row / column updates of the elimination and of every Francis step are spread over
// the lanes (one element each).  Every matrix element sees the same operations in the same order as in
// eigenvalues(), so the results agree with the scalar routine; the matrix sits in LDS, and a scalar
// thread pays an LDS round trip per element (8 ms for n = 32), the wave one per O(n) update.
// All 64 lanes must call it with the same arguments; wr / wi are written by lane 0.
#define SE_WAVE_SYNC() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier()
__device__ inline int eigenvalues_wave(volatile double* a, int n, int ld, double* wr, double* wi) {
#define A_(i, j) a[(i) * ld + (j)]
    const int lane = threadIdx.x & 63;
    // ---- elmhes ----
    for (int m = 1; m < n - 1; ++m) {
        // pivot: the first row j >= m with the largest |A(j, m-1)| (lane = candidate row)
        double x = (lane >= m && lane < n) ? A_(lane, m - 1) : 0.0;
        double big = fabs(x);
        for (int off = 32; off > 0; off >>= 1) big = fmax(big, __shfl_xor(big, off, 64));
        int i = m;
        if (big > 0.0) {
            const unsigned long long hits = __ballot(lane >= m && lane < n && fabs(x) == big);
            i = __builtin_ctzll(hits);
        }
        x = big > 0.0 ? __shfl(x, i, 64) : 0.0;
        if (i != m) {
            SE_WAVE_SYNC();
            if (lane >= m - 1 && lane < n) { const double t = A_(i, lane); A_(i, lane) = A_(m, lane); A_(m, lane) = t; }
            SE_WAVE_SYNC();
            if (lane < n) { const double t = A_(lane, i); A_(lane, i) = A_(lane, m); A_(lane, m) = t; }
            SE_WAVE_SYNC();
        }
        if (x != 0.0) {
            for (i = m + 1; i < n; ++i) {
                double y = A_(i, m - 1);
                if (y != 0.0) {
                    y /= x;
                    SE_WAVE_SYNC();
                    if (lane == 0) A_(i, m - 1) = y;
                    if (lane >= m && lane < n) A_(i, lane) -= y * A_(m, lane);
                    SE_WAVE_SYNC();
                    if (lane < n) A_(lane, m) += y * A_(lane, i);
                    SE_WAVE_SYNC();
                }
            }
        }
    }
    SE_WAVE_SYNC();
    for (int i = 2; i < n; ++i)
        if (lane < i - 1) A_(i, lane) = 0.0;
    SE_WAVE_SYNC();
    // ---- hqr ----
    double anorm = 0.0;   // only a fall-back scale for the deflation test: summation order is free
    if (lane < n)
        for (int i = 0; i <= (lane + 1 < n ? lane + 1 : n - 1); ++i) anorm += fabs(A_(i, lane));
    for (int off = 32; off > 0; off >>= 1) anorm += __shfl_xor(anorm, off, 64);
    int nn = n - 1;
    double t = 0.0, p = 0.0, q = 0.0, r = 0.0;
    while (nn >= 0) {
        int its = 0, l;
        do {
            {   // largest l in [1, nn] with a negligible subdiagonal element (lane = candidate l), else 0
                bool small = false;
                if (lane >= 1 && lane <= nn) {
                    double s = fabs(A_(lane - 1, lane - 1)) + fabs(A_(lane, lane));
                    if (s == 0.0) s = anorm;
                    small = fabs(A_(lane, lane - 1)) + s == s;
                }
                const unsigned long long hits = __ballot(small);
                l = hits ? 63 - __builtin_clzll(hits) : 0;
                SE_WAVE_SYNC();
                if (l >= 1 && lane == 0) A_(l, l - 1) = 0.0;
                SE_WAVE_SYNC();
            }
            double x = A_(nn, nn);
            if (l == nn) {
                if (lane == 0) { wr[nn] = x + t; wi[nn] = 0.0; }
                --nn;
            } else {
                double y = A_(nn - 1, nn - 1);
                double w = A_(nn, nn - 1) * A_(nn - 1, nn);
                if (l == nn - 1) {
                    p = 0.5 * (y - x);
                    q = p * p + w;
                    double z = sqrt(fabs(q));
                    x += t;
                    if (lane == 0) {
                        if (q >= 0.0) {
                            z = p + sign_of(z, p);
                            wr[nn - 1] = wr[nn] = x + z;
                            if (z != 0.0) wr[nn] = x - w / z;
                            wi[nn - 1] = wi[nn] = 0.0;
                        } else {
                            wr[nn - 1] = wr[nn] = x + p;
                            wi[nn] = z;
                            wi[nn - 1] = -z;
                        }
                    }
                    nn -= 2;
                } else {
                    if (its == 60) return nn + 1;
                    if (its == 10 || its == 20 || its == 30 || its == 40) {
                        t += x;
                        SE_WAVE_SYNC();
                        if (lane <= nn) A_(lane, lane) -= x;
                        SE_WAVE_SYNC();
                        const double s = fabs(A_(nn, nn - 1)) + fabs(A_(nn - 1, nn - 2));
                        y = x = 0.75 * s;
                        w = -0.4375 * s * s;
                    }
                    ++its;
                    int m;
                    {   // largest m in [l, nn - 2] whose two consecutive subdiagonals are small (lane = candidate m)
                        bool stop = false;
                        double pm = 0.0, qm = 0.0, rm = 0.0;
                        if (lane >= l && lane <= nn - 2) {
                            const double z = A_(lane, lane);
                            rm = x - z;
                            double s = y - z;
                            pm = (rm * s - w) / A_(lane + 1, lane) + A_(lane, lane + 1);
                            qm = A_(lane + 1, lane + 1) - z - rm - s;
                            rm = A_(lane + 2, lane + 1);
                            s = fabs(pm) + fabs(qm) + fabs(rm);
                            pm /= s; qm /= s; rm /= s;
                            stop = lane == l;
                            if (!stop) {
                                const double u = fabs(A_(lane, lane - 1)) * (fabs(qm) + fabs(rm));
                                const double v = fabs(pm) * (fabs(A_(lane - 1, lane - 1)) + fabs(z) + fabs(A_(lane + 1, lane + 1)));
                                stop = u + v == v;
                            }
                        }
                        const unsigned long long hits = __ballot(stop);
                        m = 63 - __builtin_clzll(hits);          // lane l always votes: hits != 0
                        p = __shfl(pm, m, 64);
                        q = __shfl(qm, m, 64);
                        r = __shfl(rm, m, 64);
                    }
                    SE_WAVE_SYNC();
                    {
                        const int i = m + 2 + lane;
                        if (i <= nn) {
                            A_(i, i - 2) = 0.0;
                            if (i != m + 2) A_(i, i - 3) = 0.0;
                        }
                    }
                    SE_WAVE_SYNC();
                    for (int k = m; k <= nn - 1; ++k) {
                        if (k != m) {
                            p = A_(k, k - 1);
                            q = A_(k + 1, k - 1);
                            r = 0.0;
                            if (k != nn - 1) r = A_(k + 2, k - 1);
                            if ((x = fabs(p) + fabs(q) + fabs(r)) != 0.0) { p /= x; q /= x; r /= x; }
                        }
                        const double s = sign_of(sqrt(p * p + q * q + r * r), p);
                        if (s != 0.0) {
                            SE_WAVE_SYNC();
                            if (lane == 0) {
                                if (k == m) {
                                    if (l != m) A_(k, k - 1) = -A_(k, k - 1);
                                } else {
                                    A_(k, k - 1) = -s * x;
                                }
                            }
                            p += s;
                            x = p / s;
                            y = q / s;
                            const double z = r / s;
                            q /= p;
                            r /= p;
                            SE_WAVE_SYNC();
                            {
                                const int j = k + lane;
                                if (j <= nn) {
                                    double pp = A_(k, j) + q * A_(k + 1, j);
                                    if (k != nn - 1) { pp += r * A_(k + 2, j); A_(k + 2, j) -= pp * z; }
                                    A_(k + 1, j) -= pp * y;
                                    A_(k, j) -= pp * x;
                                }
                            }
                            SE_WAVE_SYNC();
                            const int mmin = nn < k + 3 ? nn : k + 3;
                            {
                                const int i = l + lane;
                                if (i <= mmin) {
                                    double pp = x * A_(i, k) + y * A_(i, k + 1);
                                    if (k != nn - 1) { pp += z * A_(i, k + 2); A_(i, k + 2) -= pp * r; }
                                    A_(i, k + 1) -= pp * q;
                                    A_(i, k) -= pp;
                                }
                            }
                            SE_WAVE_SYNC();
                        }
                    }
                }
            }
        } while (l < nn - 1);
    }
#undef A_
    return 0;
}
#undef SE_WAVE_SYNC
#endif  // __HIPCC__

// Null vector of (h - theta I) by two steps of inverse iteration with partial-pivot
// Gaussian elimination.  h: n x n (row stride ld, preserved), work: n*n + n doubles.
SE_HD inline void eigenvector(const double* h, int n, int ld, double theta, double* y, double* work) {
    double* m = work;  // n x n
    double* b = work + n * n;
    int piv[64];
    double scale = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            m[i * n + j] = h[i * ld + j] - (i == j ? theta : 0.0);
            scale = fmax(scale, fabs(h[i * ld + j]));
        }
    const double tiny = 2.3e-16 * (scale > 0.0 ? scale : 1.0);
    // LU with partial pivoting, in place
    for (int c = 0; c < n; ++c) {
        int pr = c;
        for (int i = c + 1; i < n; ++i)
            if (fabs(m[i * n + c]) > fabs(m[pr * n + c])) pr = i;
        piv[c] = pr;
        if (pr != c)
            for (int j = 0; j < n; ++j) { double t = m[c * n + j]; m[c * n + j] = m[pr * n + j]; m[pr * n + j] = t; }
        if (fabs(m[c * n + c]) < tiny) m[c * n + c] = tiny;  // singular by construction
        for (int i = c + 1; i < n; ++i) {
            const double f = m[i * n + c] / m[c * n + c];
            m[i * n + c] = f;
            for (int j = c + 1; j < n; ++j) m[i * n + j] -= f * m[c * n + j];
        }
    }
    for (int i = 0; i < n; ++i) y[i] = 1.0;
    for (int it = 0; it < 3; ++it) {
        for (int i = 0; i < n; ++i) b[i] = y[i];
        for (int c = 0; c < n; ++c) {
            if (piv[c] != c) { double t = b[c]; b[c] = b[piv[c]]; b[piv[c]] = t; }
            for (int i = c + 1; i < n; ++i) b[i] -= m[i * n + c] * b[c];
        }
        for (int i = n - 1; i >= 0; --i) {
            double s = b[i];
            for (int j = i + 1; j < n; ++j) s -= m[i * n + j] * y[j];
            y[i] = s / m[i * n + i];
        }
        double nrm = 0.0;
        for (int i = 0; i < n; ++i) nrm = fmax(nrm, fabs(y[i]));
        if (nrm > 0.0)
            for (int i = 0; i < n; ++i) y[i] /= nrm;
    }
}

#ifdef __HIPCC__
// eigenvector() executed by one 64-lane wave (n <= 32; lane = row): the same LU with partial pivoting and
// three steps of inverse iteration; the back substitution is column oriented (each y_i is broadcast and
// removed from the rows above), so sums are taken in a different order than in the scalar routine (last
// digits may differ).  work: n*n + 2n doubles.  All lanes call it with the same arguments.
// theta_im != 0: a vector u of the REAL invariant plane of the complex pair theta +- i theta_im, as the null vector of
// (H - theta)^2 + theta_im^2 = H^2 - 2 theta H + |.|^2 (real arithmetic throughout); the partner is
// v = (theta u - H u) / theta_im, so that H (u + i v) = (theta + i theta_im)(u + i v).
__device__ inline void eigenvector_wave(const double* h, int n, int ld, double theta, volatile double* y,
                                        volatile double* work, double theta_im = 0.0) {
#define SE_SYNC() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier()
    const int lane = threadIdx.x & 63;
    volatile double* m = work;            // n x n
    volatile double* b = work + n * n;    // n
    volatile double* piv = b + n;         // n (row indices kept as doubles)
    double scale = 0.0;
    for (int e = lane; e < n * n; e += 64) {
        const int i = e / n, j = e - i * n;
        double v = h[i * ld + j];
        if (theta_im != 0.0) {
            double q = 0.0;
            for (int k = 0; k < n; ++k) q = fma(h[i * ld + k], h[k * ld + j], q);
            v = q - 2.0 * theta * v + (i == j ? theta * theta + theta_im * theta_im : 0.0);
            m[e] = v;
        } else {
            m[e] = v - (i == j ? theta : 0.0);
        }
        scale = fmax(scale, fabs(v));
    }
    for (int off = 32; off > 0; off >>= 1) scale = fmax(scale, __shfl_xor(scale, off, 64));
    const double tiny = 2.3e-16 * (scale > 0.0 ? scale : 1.0);
    SE_SYNC();
    for (int c = 0; c < n; ++c) {
        const double mine = (lane >= c && lane < n) ? fabs(m[lane * n + c]) : -1.0;
        double big = mine;
        for (int off = 32; off > 0; off >>= 1) big = fmax(big, __shfl_xor(big, off, 64));
        const int pr = __builtin_ctzll(__ballot(mine == big));     // first row attaining the maximum
        if (lane == 0) piv[c] = (double)pr;
        if (pr != c && lane < n) { const double t = m[c * n + lane]; m[c * n + lane] = m[pr * n + lane]; m[pr * n + lane] = t; }
        SE_SYNC();
        if (lane == 0 && fabs(m[c * n + c]) < tiny) m[c * n + c] = tiny;  // singular by construction
        SE_SYNC();
        if (lane > c && lane < n) {
            const double f = m[lane * n + c] / m[c * n + c];
            m[lane * n + c] = f;
            for (int j = c + 1; j < n; ++j) m[lane * n + j] -= f * m[c * n + j];
        }
        SE_SYNC();
    }
    if (lane < n) y[lane] = 1.0;
    SE_SYNC();
    for (int it = 0; it < 3; ++it) {
        if (lane < n) b[lane] = y[lane];
        SE_SYNC();
        for (int c = 0; c < n; ++c) {
            const int pr = (int)piv[c];
            if (pr != c && lane == 0) { const double t = b[c]; b[c] = b[pr]; b[pr] = t; }
            SE_SYNC();
            if (lane > c && lane < n) b[lane] -= m[lane * n + c] * b[c];
            SE_SYNC();
        }
        for (int i = n - 1; i >= 0; --i) {
            if (lane == 0) y[i] = b[i] / m[i * n + i];
            SE_SYNC();
            if (lane < i) b[lane] -= m[lane * n + i] * y[i];
            SE_SYNC();
        }
        double nrm = lane < n ? fabs(y[lane]) : 0.0;
        for (int off = 32; off > 0; off >>= 1) nrm = fmax(nrm, __shfl_xor(nrm, off, 64));
        if (nrm > 0.0 && lane < n) y[lane] /= nrm;
        SE_SYNC();
    }
#undef SE_SYNC
}
#endif  // __HIPCC__

}  // namespace small_eig
