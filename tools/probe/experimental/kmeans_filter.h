// EXPERIMENT (not part of libmsmhip.so): fp32 matrix-core FILTER + fp64 REFINEMENT for the k-means
// distance table.  Written to be included by pmarlo_amd/csrc/kmeans.hip inside its anonymous namespace
// (uses FitState, to_fixed, load_as_f64, the KSTAMP macros) and dispatched from launch_mfma for KS <= 4,
// single-tile launches.
//
// Round-1 result (MI355X, C3 k-means pass: 1 M frames, d = 10, k = 500): bit-exact against the oracle
// on the whole k-means / configuration test set (labels, ties, distances, the Lloyd fit), but NOT
// faster than the all-fp64 kernel: 0.30 ms per pass on the bench data against 0.23 ms (0.23 vs 0.265 on
// uniform random data).  The fp32 tile loop itself is 1.65x faster; the gain is eaten by (i) the 0.7 %
// of frames that cannot be certified -- one such frame sends its whole 16-frame group through the
// exhaustive fp64 scan (10 % of the groups) -- and (ii) the heavier per-frame prologue / refinement.
// To pay off it needs the uncertified frames deferred to a compacted second pass of the fp64 kernel and
// a cheaper certification step.
//
// Idea.  The label of a frame is the arg-max of m_j = x.c_j - |c_j|^2/2 over the k centres, in
// the pinned fp64 arithmetic.  A cheap fp32 pass (v_mfma_f32_16x16x4_f32, twice the rate of the
// fp64 matrix instruction) cannot decide near-ties, but it can CERTIFY most frames: with
//   E  >=  | fp32 score - exact score |   for every centre of this frame,
//   M1  =  largest fp32 score, found in tile pair P*;   R = largest fp32 score outside P*,
// M1 - R > 2E implies that the fp64 arg-max lies inside P*, and re-scoring the 32 centres of P* with
// the fp64 chain gives exactly the label, tie-break and minimum distance of the all-fp64 kernel.
// Frames that are not certified take the exhaustive fp64 scan of the LDS tile (lane-parallel).
//
// Error bound.  Inputs are rounded to fp32 (relative 2^-24 each), the 4 KS products are summed in
// fp32 by the matrix core (<= 4 KS + 2 roundings of partial sums bounded by S = sum |x_f c_f| + h),
// so |error| <= (4 KS + 6) 2^-24 S (1 + o(1)), and S <= |x| |c|_max + h_max (Cauchy-Schwarz).  E uses
// the factor 2 (4 KS + 6) 2^-24: a 2x margin over the bound.
#pragma once

typedef float v4f32 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float max_f32(float a, float b) {
    float r;
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// LDS: cs64 [tiles][TS] | chalf [k16] | cs32 [tiles][KS*64] floats | (ACCUM) lsum [k][d], lcnt [k]
template <typename T, int KS, int NF, bool ACCUM, bool FOLD>
__global__ __launch_bounds__(1024, 4) void kmeans_filter_kernel(
    const T* __restrict__ x, int64_t n, int d, int64_t ld, const double* __restrict__ centers, int k,
    const double* __restrict__ mean, const double* __restrict__ stdv, int32_t* __restrict__ labels,
    double* __restrict__ mindist, const FitState* __restrict__ st, unsigned long long* __restrict__ sums,
    unsigned long long* __restrict__ counts, int lds_acc, unsigned long long* __restrict__ n_fallback) {
    constexpr int kMT = 1024;
    constexpr int TS = KS * 64 + 1;
    constexpr int TS32 = KS * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if constexpr (ACCUM) {
        if (st->done != 0.0) return;
    }
    const int k16 = (k + 15) & ~15, n_tiles = k16 / 16;
    double* cs = reinterpret_cast<double*>(smem_raw);
    double* chalf = cs + (size_t)n_tiles * TS;
    float* cs32 = reinterpret_cast<float*>(chalf + k16);
    unsigned long long* lsum = reinterpret_cast<unsigned long long*>(cs32 + (size_t)n_tiles * TS32 + (n_tiles * TS32 & 1));
    unsigned long long* lcnt = lsum + (size_t)k * d;
    __shared__ double hmax_s;
    __shared__ double red[16];
    __shared__ int unit_ctr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j16 = lane & 15, g = lane >> 4;
    const int fold_s = d >> 2, fold_g = d & 3;
    const double scale = ACCUM ? st->scale : 0.0;
    if constexpr (ACCUM) {
        if (lds_acc)
            for (int i = tid; i < k * (d + 1); i += kMT) lsum[i] = 0ull;
    }
    if (tid == 0) unit_ctr = kMT / 64;
    KSTAMP_INIT
    // ---- stage the fp64 tile (as the fp64 kernel does), the half-norms, then the fp32 image
    for (int i = tid; i < k16 * 4 * KS; i += kMT) {
        const int jt = i / (KS * 64);
        const int rem = i - jt * (KS * 64);
        const int s = rem >> 6, gg = (rem >> 4) & 3, jj = rem & 15;
        const int j = jt * 16 + jj, f = 4 * s + gg;
        cs[jt * TS + rem] = (j < k && f < d) ? centers[(size_t)j * d + f] : 0.0;
    }
    __syncthreads();
    double hm = 0.0;
    for (int j = tid; j < k16; j += kMT) {
        double* cj = cs + (j >> 4) * TS + (j & 15);
        double h = __builtin_inf();
        if (j < k) {
            double a = 0.0;
            for (int f = 0; f < d; ++f) {
                const double c = cj[(f >> 2) * 64 + (f & 3) * 16];
                a = fma(c, c, a);
            }
            h = 0.5 * a;
            hm = fmax(hm, h);
        }
        chalf[j] = h;
        if (FOLD) cj[fold_s * 64 + fold_g * 16] = -h;
    }
    for (int off = 32; off > 0; off >>= 1) hm = fmax(hm, __shfl_down(hm, off, 64));
    if (lane == 0) red[wave] = hm;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < kMT / 64; ++w) t = fmax(t, red[w]);
        hmax_s = t;
    }
    for (int i = tid; i < n_tiles * TS32; i += kMT) {
        const int jt = i / TS32, rem = i - jt * TS32;
        cs32[i] = (float)cs[jt * TS + rem];   // -inf of a padded fold slot stays -inf
    }
    __syncthreads();
    KSTAMP(0);
    const double hmax = hmax_s, cmax = sqrt(2.0 * hmax);
    const double kappa = 2.0 * (4 * KS + 6) * 5.9604644775390625e-08;   // 2^-24

    const int64_t frames_per_wave = 16 * NF;
    const int64_t n_units = (n + frames_per_wave - 1) / frames_per_wave;
    const int64_t units_per_block = (n_units + gridDim.x - 1) / gridDim.x;
    const int64_t u_begin = (int64_t)blockIdx.x * units_per_block;
    const int64_t u_end = min(n_units, u_begin + units_per_block);
    unsigned long long my_fallbacks = 0;
    for (int64_t unit = u_begin + wave; unit < u_end;) {
        int nt = 0;
        if (lane == 0) nt = atomicAdd(&unit_ctr, 1);
        const int64_t nxt = u_begin + __builtin_amdgcn_readfirstlane(nt);
        double zb[NF][KS];
        float zf[NF][KS];
        int64_t fidx[NF];
        bool fok[NF];
        double xn2[NF];
#pragma unroll
        for (int u = 0; u < NF; ++u) {
            fidx[u] = unit * frames_per_wave + 16 * u + j16;
            fok[u] = fidx[u] < n;
            const T* row = x + (fok[u] ? fidx[u] : n - 1) * ld;
            double q = 0.0;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int f = 4 * s + g;
                double v = 0.0;
                if (f < d) {
                    v = load_as_f64(row + f);
                    if (mean) v = (v - mean[f]) / stdv[f];
                    if (!fok[u]) v = 0.0;
                }
                q = fma(v, v, q);
                if (FOLD && s == fold_s && g == fold_g) v = 1.0;
                zb[u][s] = v;
                zf[u][s] = (float)v;
            }
            q += __shfl_xor(q, 16, 64);
            q += __shfl_xor(q, 32, 64);
            xn2[u] = q;
        }
        KSTAMP(1);
        // ---- fp32 filter pass: pair maxima with top-2 tracking
        float b1[NF], b2[NF];
        int bp[NF];
#pragma unroll
        for (int u = 0; u < NF; ++u) { b1[u] = -__builtin_inff(); b2[u] = -__builtin_inff(); bp[u] = 0; }
        for (int jt = 0; jt < n_tiles; jt += 2) {
            const int jb = min(jt + 1, n_tiles - 1);
            v4f32 acca[NF], accb[NF];
#pragma unroll
            for (int u = 0; u < NF; ++u) { acca[u] = (v4f32){0.f, 0.f, 0.f, 0.f}; accb[u] = acca[u]; }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float fa = cs32[jt * TS32 + s * 64 + lane];
#pragma unroll
                for (int u = 0; u < NF; ++u) acca[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, zf[u][s], acca[u], 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float fb = cs32[jb * TS32 + s * 64 + lane];
#pragma unroll
                for (int u = 0; u < NF; ++u) accb[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb, zf[u][s], accb[u], 0, 0, 0);
            }
            if constexpr (!FOLD) {   // D rows of the fp32 instruction: centre 4 g + r
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float ha = (float)chalf[jt * 16 + 4 * g + r], hb = (float)chalf[jb * 16 + 4 * g + r];
#pragma unroll
                    for (int u = 0; u < NF; ++u) { acca[u][r] -= ha; accb[u][r] -= hb; }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15" ::: "memory");   // MFMA -> VALU hazard of the inline-asm maxima
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NF; ++u) {
                const float ma = max_f32(max_f32(acca[u][0], acca[u][1]), max_f32(acca[u][2], acca[u][3]));
                const float mb = jb != jt ? max_f32(max_f32(accb[u][0], accb[u][1]), max_f32(accb[u][2], accb[u][3])) : ma;
                const float m = max_f32(ma, mb);
                const bool better = m > b1[u];
                const float sec = better ? b1[u] : m;
                b2[u] = max_f32(b2[u], sec);
                b1[u] = better ? m : b1[u];
                bp[u] = better ? jt : bp[u];
            }
        }
        KSTAMP(2);
        // ---- certification and fp64 refinement
        double bm[NF];
        int bidx[NF];
#pragma unroll
        for (int u = 0; u < NF; ++u) {
            float M1 = b1[u];
            M1 = fmaxf(M1, __shfl_xor(M1, 16, 64));
            M1 = fmaxf(M1, __shfl_xor(M1, 32, 64));
            const unsigned long long own = __ballot(b1[u] == M1) >> j16;
            const unsigned holders = (unsigned)(own & 1) | (unsigned)((own >> 15) & 2) | (unsigned)((own >> 30) & 4) |
                                     (unsigned)((own >> 45) & 8);
            const int gs = holders ? __builtin_ctz(holders) : 0;
            const int pstar = __shfl(bp[u], j16 + 16 * gs, 64);
            // R bounds the fp32 score of every centre that is NOT one of lane gs's 8 candidates (its
            // 4 accumulator rows in the two tiles of its winning pair)
            float R = g == gs ? b2[u] : b1[u];
            R = fmaxf(R, __shfl_xor(R, 16, 64));
            R = fmaxf(R, __shfl_xor(R, 32, 64));
            const double E = kappa * fma(sqrt(xn2[u]), cmax, hmax);
            const bool certified = (double)M1 - (double)R > 2.0 * E;   // false for NaN / inf scores too
            double best = -__builtin_inf();
            int bi = 0x7fffffff;
            if (__all(certified)) {
                // the usual case, all 16 frames of the group certified: the 8 candidates are re-scored
                // two per lane (row 4 gs + g of both tiles) -- the fp64 kernel's recovery step
                const int ta = pstar, tb = min(pstar + 1, n_tiles - 1);
                const int crow = 4 * gs + g;
                const double* ca = cs + ta * TS + crow;
                const double* cb = cs + tb * TS + crow;
                double da = 0.0, db = 0.0;
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int gp = 0; gp < 4; ++gp) {
                        const double z = __shfl(zb[u][s], j16 + 16 * gp, 64);
                        da = fma(ca[s * 64 + gp * 16], z, da);
                        db = fma(cb[s * 64 + gp * 16], z, db);
                    }
                const int ia = ta * 16 + crow, ib = tb * 16 + crow;
                if constexpr (!FOLD) { da -= chalf[ia]; db -= chalf[ib]; }
                if (ia < k) { best = da; bi = ia; }
                if (ib < k && ib != ia && (db > best || (db == best && ib < bi))) { best = db; bi = ib; }
            } else {
                // some frame of the group is not certified: every frame of the group takes the
                // exhaustive fp64 scan (rare: ~0.2 % of the frames are ambiguous)
                if (!certified && g == 0 && fok[u]) ++my_fallbacks;
                double zv[4 * KS];
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int gp = 0; gp < 4; ++gp) zv[4 * s + gp] = __shfl(zb[u][s], j16 + 16 * gp, 64);
                for (int t = 0; t < n_tiles; ++t) {
                    const double* ct = cs + t * TS + g;
                    double dacc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int s = 0; s < KS; ++s)
#pragma unroll
                        for (int gp = 0; gp < 4; ++gp) {
                            const double* cf = ct + s * 64 + gp * 16;
#pragma unroll
                            for (int r = 0; r < 4; ++r) dacc[r] = fma(cf[4 * r], zv[4 * s + gp], dacc[r]);
                        }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = t * 16 + g + 4 * r;
                        double m = dacc[r];
                        if constexpr (!FOLD) m -= chalf[c];
                        if (c < k && (m > best || (m == best && c < bi))) { best = m; bi = c; }
                    }
                }
            }
#pragma unroll
            for (int off = 16; off < 64; off <<= 1) {
                const double ob = __shfl_xor(best, off, 64);
                const int oi = __shfl_xor(bi, off, 64);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            bm[u] = best;
            bidx[u] = bi < k ? bi : 0;
        }
        KSTAMP(3);
#pragma unroll
        for (int u = 0; u < NF; ++u) {
            if (!fok[u]) continue;
            if constexpr (ACCUM) {
                unsigned long long* srow = (lds_acc ? lsum : sums) + (size_t)bidx[u] * d;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int f = 4 * s + g;
                    if (f < d) atomicAdd(&srow[f], (unsigned long long)to_fixed(zb[u][s], scale));
                }
                if (g == 0) atomicAdd((lds_acc ? lcnt : counts) + bidx[u], 1ull);
            } else {
                if (g == 0) {
                    labels[fidx[u]] = bidx[u];
                    if (mindist) {
                        const T* row = x + fidx[u] * ld;
                        double zsq = 0.0;
                        for (int f = 0; f < d; ++f) {
                            double v = load_as_f64(row + f);
                            if (mean) v = (v - mean[f]) / stdv[f];
                            zsq = fma(v, v, zsq);
                        }
                        const double md = -2.0 * bm[u] + zsq;
                        mindist[fidx[u]] = md > 0.0 ? md : 0.0;
                    }
                }
            }
        }
        KSTAMP(4);
        unit = nxt;
    }
    KSTAMP(7);
    if (n_fallback && my_fallbacks) atomicAdd(n_fallback, my_fallbacks);
    if constexpr (ACCUM) {
        if (lds_acc) {
            __syncthreads();
            for (int i = tid; i < k * d; i += kMT)
                if (lsum[i]) atomicAdd(&sums[i], lsum[i]);
            for (int i = tid; i < k; i += kMT)
                if (lcnt[i]) atomicAdd(&counts[i], lcnt[i]);
        }
    }
    KSTAMP(6);
    KSTAMP_FLUSH
}
