// Certified bf16 matrix-core FILTER + pinned fp64 refinement for the k-means score table (d <= 10).
// Included by kmeans.hip inside its anonymous namespace (uses FitState, to_fixed, load_as_f64).
//
// The label of a frame is the arg-max over the k centres of the PINNED fp64 score
//     m_j = fl(dot_j - h_j),  dot_j = ascending-feature fma chain of c_j[f] * z[f],  h_j = |c_j|^2 / 2
// (strict '>' over ascending j; kmeans.hip states why this is sklearn's arg-min).  An all-fp64 scan costs
// three v_mfma_f64_16x16x4 (64 cycles each) per 16 centres x 16 frames.  This kernel gets the same labels,
// bit for bit, from bf16 matrix instructions at 1/6 of that matrix-pipe time:
//
// 1. FILTER.  Every coordinate is split into three bf16 numbers (v = vh + vm + vl + r, |r| <= 2^-24 |v|:
//    a bf16 triple holds an fp32 exactly).  One K = 32 instruction pair (v_mfma_f32_16x16x32_bf16 twice, 64
//    product slots) evaluates, per centre and frame,
//        u_j = sum_f (ch xh + cm xh + ch xm + cm xm + cl xh + ch xl)  -  (1 - kappa) h_j  +  kappa |c_j| |x|
//    i.e. the score plus kappa (|x||c_j| + h_j) >= kappa S_j,  S_j = sum_f |x_f c_jf| + h_j.
//    u_j is an UPPER BOUND of the pinned score m_j:
//      dropped split terms   <= 4.01 x 2^-24 S_j
//      accumulation          <= 68.7 x 2^-24 S_j: the instruction aligns its 32 products and C to the largest
//                               exponent, keeps 24 bits below it, truncates, adds, and rounds once (measured on
//                               MI355X, tools/probe/bf16_filter_probe.hip: terms below 2^-24 of the largest one
//                               vanish, equal terms at 2^-24 survive, the order of the slots does not matter;
//                               worst observed error 11.7 x 2^-24 of the largest term against the bound 33)
//      fp64 chain of m_j     <= 12 x 2^-53 S_j
//    and kappa = 80 x 2^-24 covers their sum with 7 x 2^-24 S_j to spare.  With d <= 4 everything fits ONE
//    instruction (28 slots): accumulation <= 34.3 x 2^-24 S_j, kappa = 44 x 2^-24.
// 2. Per lane (4 accumulator rows of a frame) only the largest PAIR maximum of u, the runner-up pair maximum and
//    the pair index are tracked (max3 tree, med3, max, compare, select: 8 VALU per 8 scores).
// 3. REFINEMENT.  The 8 centres of the winning lane's winning pair are re-scored with the pinned fp64 chain; jw =
//    their arg-max (lowest index on ties), m_jw its score.  R = the largest u outside those 8 centres (runner-up
//    pair of the winning lane, best pairs of the frame's other three lanes).  If m_jw > R, then every centre
//    outside the 8 has m_j <= u_j <= R < m_jw: the label is jw, exactly as the all-fp64 scan gives it.
// 4. Otherwise (about 1 % of the frames: near-ties across candidate sets, duplicate centres, NaN / out-of-range
//    input) the wave scans all k centres for that frame with the pinned fp64 chain, one lane per centre.
//
// Range guard: the bounds assume no fp32 overflow / underflow inside the filter, so frames or centres with a
// non-zero coordinate outside [1e-14, 1e18] in magnitude (or NaN) are sent to step 4.
#pragma once

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v4f32 __attribute__((ext_vector_type(4)));

constexpr int kFilterMaxD = 10;
constexpr double kFilterLo = 1e-14, kFilterHi = 1e18;

__host__ __device__ constexpr int filter_nm(int d) { return 6 * d + 4 <= 32 ? 1 : 2; }   // K = 32 instructions per tile
__host__ __device__ constexpr double filter_kappa(int nm) { return (nm == 1 ? 44.0 : 80.0) * 5.9604644775390625e-08; }

__device__ __forceinline__ unsigned bf16_rn(float f) {            // round to nearest even, finite input
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float bf16_f32(unsigned h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ unsigned bf16_up(float f) {            // smallest bf16 >= f, f >= 0 finite
    unsigned h = bf16_rn(f);
    if (bf16_f32(h) < f) ++h;
    return h;
}
__device__ __forceinline__ void bf16_split3(double v, unsigned (&out)[3]) {
    float r = (float)v;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        out[t] = bf16_rn(r);
        r -= bf16_f32(out[t]);
    }
}
// product term t of slot t*d + f pairs these split parts (0 = high, 1 = middle, 2 = low)
__device__ __forceinline__ int filter_part_c(int t) { return t == 1 || t == 3 ? 1 : (t == 4 ? 2 : 0); }
__device__ __forceinline__ int filter_part_x(int t) { return t == 2 || t == 3 ? 1 : (t == 5 ? 2 : 0); }

// max3 / med3 are written with compiler-visible builtins (hipcc then pads the MFMA -> VALU read hazard itself;
// it does not inside inline asm).  fmaxf(fmaxf(a, b), c) becomes ONE v_max3_f32 with no canonicalising v_max x, x
// when `a` is already the result of a VALU maximum, which is why the pair maximum below starts from b2.
__device__ __forceinline__ float max3_f32(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float med3_f32(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }
// the bare v_max_f32 (operands are never NaN where this is used): fmaxf() on a loop-carried value costs an extra
// canonicalising v_max per call, and med3(a, b, +inf) is folded back into it
__device__ __forceinline__ float hw_max_f32(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ---------------------------------------------------------------------------------------------------------
// Frame images: row t = the B operand of frame t, 32 * NM bf16 slots in slot order (lane (j, q) of instruction m
// loads the 16 bytes at 64 m + 16 q): slot t*d + f = the x part of term t, slots 6d .. 6d+2 = 1.0 (they meet the
// three parts of -(1 - kappa) h_j), the LAST slot = |x| rounded up (it meets kappa |c_j|), +inf when the frame
// fails the range guard.  One thread per frame.
// ---------------------------------------------------------------------------------------------------------
// Both sides of the kernel go through the LDS so that the memory system only ever sees consecutive lanes on
// consecutive 16-byte pieces: a lane reading its own 8 D-byte row or writing its own 64 NM-byte image row touched 64
// cache lines per instruction (61 us for 208 MB at C3).  `dense`: rows back to back (ld == D) on a 16-byte boundary.
template <typename T, int D>
__global__ __launch_bounds__(256) void kmeans_pack_kernel(const T* __restrict__ x, int64_t n, int64_t ld,
                                                         const double* __restrict__ mean,
                                                         const double* __restrict__ stdv, uint4* __restrict__ image,
                                                         int dense) {
    constexpr int NM = filter_nm(D);
    constexpr int kOutStride = 4 * NM + 1;              // uint4 per staged image row: one of padding (bank spread)
    constexpr int kInBytes = 256 * D * (int)sizeof(T);  // a workgroup's rows, back to back
    constexpr int kOutBytes = 256 * kOutStride * 16;
    __shared__ __attribute__((aligned(16))) unsigned char stage[kOutBytes > kInBytes ? kOutBytes : kInBytes];
    const int64_t t0 = (int64_t)blockIdx.x * 256;
    const int64_t t = t0 + threadIdx.x;
    const int rows = (int)(n - t0 < 256 ? n - t0 : 256);
    double v[D];
    if (dense) {
        // rows * D * sizeof(T) bytes from x + t0 * D, 16 bytes per lane and trip (the tail of the last workgroup by element)
        const unsigned char* src = reinterpret_cast<const unsigned char*>(x + t0 * D);
        const int nbytes = rows * D * (int)sizeof(T);
        for (int b = threadIdx.x * 16; b + 16 <= nbytes; b += 256 * 16)
            *reinterpret_cast<uint4*>(stage + b) = *reinterpret_cast<const uint4*>(src + b);
        if (threadIdx.x < (nbytes & 15) / (int)sizeof(T)) {
            const int e = (nbytes & ~15) / (int)sizeof(T) + threadIdx.x;
            reinterpret_cast<T*>(stage)[e] = reinterpret_cast<const T*>(src)[e];
        }
        __syncthreads();
        const T* row = reinterpret_cast<const T*>(stage) + threadIdx.x * D;
#pragma unroll
        for (int f = 0; f < D; ++f) v[f] = threadIdx.x < rows ? (double)row[f] : 0.0;
        __syncthreads();   // the staging area is reused for the images
    } else {
        const T* row = x + (t < n ? t : n - 1) * ld;
#pragma unroll
        for (int f = 0; f < D; ++f) v[f] = load_as_f64(row + f);
    }
    unsigned part[D][3];
    double q = 0.0;
    bool ok = true;
#pragma unroll
    for (int f = 0; f < D; ++f) {
        if (mean) v[f] = (v[f] - mean[f]) / stdv[f];
        const double a = fabs(v[f]);
        ok = ok && (v[f] == 0.0 || (a >= kFilterLo && a <= kFilterHi));   // false for NaN
    }
#pragma unroll
    for (int f = 0; f < D; ++f) {
        const double w = ok ? v[f] : 0.0;
        q = fma(w, w, q);
        bf16_split3(w, part[f]);
    }
    const unsigned xn = ok ? bf16_up((float)(sqrt(q) * (1.0 + 9.5367431640625e-07))) : 0x7F80u;
    unsigned slots[32 * NM];
#pragma unroll
    for (int s = 0; s < 32 * NM; ++s) {
        unsigned val = 0;
        if (s < 6 * D) val = part[s % D][s / D == 2 || s / D == 3 ? 1 : (s / D == 5 ? 2 : 0)];
        else if (s < 6 * D + 3) val = 0x3F80u;            // 1.0 against the three parts of -(1 - kappa) h_j
        else if (s == 32 * NM - 1) val = xn;              // |x| rounded up against kappa |c_j|
        slots[s] = val;
    }
    uint4* mine = reinterpret_cast<uint4*>(stage) + threadIdx.x * kOutStride;
#pragma unroll
    for (int c = 0; c < 4 * NM; ++c)
        mine[c] = make_uint4(slots[8 * c + 0] | (slots[8 * c + 1] << 16), slots[8 * c + 2] | (slots[8 * c + 3] << 16),
                             slots[8 * c + 4] | (slots[8 * c + 5] << 16), slots[8 * c + 6] | (slots[8 * c + 7] << 16));
    __syncthreads();
    uint4* dst = image + t0 * (4 * NM);
    for (int c = threadIdx.x; c < rows * 4 * NM; c += 256)
        dst[c] = reinterpret_cast<const uint4*>(stage)[(c / (4 * NM)) * kOutStride + (c % (4 * NM))];
}

// ---------------------------------------------------------------------------------------------------------
// Centre side, once per launch and workgroup (filter_stage_tile, one wave per 16-centre tile) into the kernel's LDS:
//   img   [n_tiles][NM][64] uint4      A operands, lane-major per instruction
//   cs64  [n_tiles * 16][DP + 2] f64   centre coordinates zero-padded to DP features, then h_j (+inf for padding
//                                      rows), then one pad double (rows are 16-byte aligned): refinement and scan
//   flag  int [n_tiles]                a centre of the tile failed the range guard: every frame takes the exhaustive scan
// (Each workgroup builds the tables for itself: as fast as copying 113 KB of ready tables from a staging launch, and
// one launch less per pass.)  DP = 4 (NM = 1) or 10 (NM = 2): the fp64
// chains run over DP features with zeros beyond d, which leaves every partial sum unchanged.
// ---------------------------------------------------------------------------------------------------------
template <int NM>
struct FilterShape {
    static constexpr int DP = NM == 1 ? 4 : 10;
    static constexpr int D1 = DP + 2;
    static constexpr int kTileBytes = NM * 1024 + 16 * D1 * 8;    // image + table rows of one 16-centre tile
};

// One wave stages one 16-centre tile (simg: NM * 64 * 8 shorts of LDS of its own) in two steps, so that a wave with
// several tiles has the global loads of all of them in flight before it builds the first (the build is ~300
// instructions; one load round trip under load is as long):
//   filter_stage_fetch: the tile's 16 x d coordinates, one coalesced load per 64 elements (element e = row e / d,
//                       feature e % d), at most kStageRegs per lane;
//   filter_stage_build: bf16 triples into the A operands, coordinates into the table rows; the 16 row lanes then read
//                       their row back from the LDS for h_j, the range guard and the kappa slots (same chain, same bits).
// `cs_g` and `flag` may be global or LDS (in-order within the wave either way); the table rows go to cs_g + j * D1.
constexpr int kStageRegs = (16 * kFilterMaxD + 63) / 64;
// i / d for 0 <= i < 256, 1 <= d <= 10 without the integer-division sequence: (i + 0.5) / d is at least 0.05 away from
// every integer, far outside fp32 rounding
__device__ __forceinline__ int stage_row(int i, int d) { return (int)(((float)i + 0.5f) * (1.0f / (float)d)); }
__device__ __forceinline__ void filter_stage_fetch(int tile, int lane, const double* __restrict__ centers, int k, int d,
                                                   double (&c)[kStageRegs]) {
#pragma unroll
    for (int u = 0; u < kStageRegs; ++u) {
        const int i = lane + 64 * u;
        const int r = stage_row(i, d), j = tile * 16 + r;
        c[u] = (i < 16 * d && j < k) ? centers[(size_t)tile * 16 * d + i] : 0.0;
    }
}
template <int NM>
__device__ __forceinline__ void filter_stage_build(int tile, int lane, unsigned short* simg, const double (&c)[kStageRegs],
                                                   int k, int d, uint4* __restrict__ img_g, double* __restrict__ cs_g,
                                                   int* __restrict__ flag) {
    using S = FilterShape<NM>;
    constexpr double kappa = filter_kappa(NM);
    for (int i = lane; i < NM * 64 * 4; i += 64) reinterpret_cast<unsigned*>(simg)[i] = 0u;
    for (int i = lane; i < 16 * S::D1; i += 64) cs_g[(size_t)tile * 16 * S::D1 + i] = 0.0;   // pads beyond d, h and pad slots
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // one wave, LDS in order: only the compiler must not reorder
    auto slot_addr = [&](int i, int sl) {      // bf16 element of tile row i, slot sl
        const int m = sl >> 5, qq = (sl & 31) >> 3, e = sl & 7;
        return ((m * 64) + qq * 16 + i) * 8 + e;
    };
#pragma unroll
    for (int u = 0; u < kStageRegs; ++u) {
        const int i = lane + 64 * u;
        const int r = stage_row(i, d), f = i - r * d, j = tile * 16 + r;
        if (i < 16 * d && j < k) {
            unsigned cp[3];
            bf16_split3(c[u], cp);
#pragma unroll
            for (int t = 0; t < 6; ++t) simg[slot_addr(r, t * d + f)] = (unsigned short)cp[filter_part_c(t)];
            cs_g[(size_t)j * S::D1 + f] = c[u];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    int bad_row = 0;
    if (lane < 16) {
        const int j = tile * 16 + lane;
        double* row = cs_g + (size_t)j * S::D1;
        double h = __builtin_inf();
        if (j < k) {
            double a = 0.0;
            bool ok = true;
#pragma unroll
            for (int f = 0; f < S::DP; ++f) {
                const double cf = row[f];   // zero beyond d
                a = fma(cf, cf, a);
                const double ac = fabs(cf);
                ok = ok && (cf == 0.0 || (ac >= kFilterLo && ac <= kFilterHi));
            }
            h = 0.5 * a;
            if (!ok) bad_row = 1;
            unsigned hs[3];
            bf16_split3(-(h - kappa * h), hs);
            simg[slot_addr(lane, 6 * d)] = (unsigned short)hs[0];
            simg[slot_addr(lane, 6 * d + 1)] = (unsigned short)hs[1];
            simg[slot_addr(lane, 6 * d + 2)] = (unsigned short)hs[2];
            simg[slot_addr(lane, 32 * NM - 1)] =
                (unsigned short)bf16_up((float)(kappa * sqrt(a) * (1.0 + 9.5367431640625e-07)));
        } else {
            simg[slot_addr(lane, 6 * d)] = 0xFF7F;   // -3.4e38 x 1.0: a padding centre never holds a maximum
        }
        row[S::DP] = h;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (img_g)   // NULL: `simg` is the image's final place (the main kernel stages into its own LDS)
        for (int i = lane; i < NM * 64; i += 64) img_g[(size_t)tile * NM * 64 + i] = reinterpret_cast<const uint4*>(simg)[i];
    const bool any_bad = __any(bad_row != 0);      // every staging rewrites its flag
    if (lane == 0) flag[tile] = any_bad ? 1 : 0;
}
template <int NM>
__device__ __forceinline__ void filter_stage_tile(int tile, int lane, unsigned short* simg, const double* centers, int k,
                                                  int d, uint4* __restrict__ img_g, double* __restrict__ cs_g,
                                                  int* __restrict__ flag) {
    double c[kStageRegs];
    filter_stage_fetch(tile, lane, centers, k, d, c);
    filter_stage_build<NM>(tile, lane, simg, c, k, d, img_g, cs_g, flag);
}

// cross-row butterflies over the 4 lanes (j, j + 16, j + 32, j + 48) that share a frame: v_permlane16_swap /
// v_permlane32_swap exchange whole rows of 16 lanes on the VALU (no LDS round trip as ds_bpermute would take)
typedef unsigned v2u32 __attribute__((ext_vector_type(2)));
template <typename Op>
__device__ __forceinline__ unsigned xrow_reduce_u32(unsigned x, Op op) {
    v2u32 r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    x = op(r[0], r[1]);
    r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return op(r[0], r[1]);
}
__device__ __forceinline__ float xrow_max_f32(float x) {   // operands are never NaN here: med3(a, b, +inf) = max(a, b)
    return __uint_as_float(xrow_reduce_u32(__float_as_uint(x), [](unsigned a, unsigned b) {
        return __float_as_uint(__builtin_amdgcn_fmed3f(__uint_as_float(a), __uint_as_float(b), __builtin_inff()));
    }));
}
__device__ __forceinline__ int xrow_min_i32(int x) {
    return (int)xrow_reduce_u32((unsigned)x, [](unsigned a, unsigned b) { return (unsigned)min((int)a, (int)b); });
}
// arg-max of (score, index) with the lower index on equal scores
__device__ __forceinline__ void xrow_argmax_f64(double& best, int& bi) {
#pragma unroll
    for (int step = 0; step < 2; ++step) {
        const unsigned lo = (unsigned)__double_as_longlong(best), hi = (unsigned)(__double_as_longlong(best) >> 32);
        v2u32 rl, rh, ri;
        if (step == 0) {
            rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
            rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
            ri = __builtin_amdgcn_permlane16_swap((unsigned)bi, (unsigned)bi, false, false);
        } else {
            rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
            rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
            ri = __builtin_amdgcn_permlane32_swap((unsigned)bi, (unsigned)bi, false, false);
        }
        const double a = __longlong_as_double((long long)(((unsigned long long)rh[0] << 32) | rl[0]));
        const double b = __longlong_as_double((long long)(((unsigned long long)rh[1] << 32) | rl[1]));
        const int ia = (int)ri[0], ib = (int)ri[1];
        const bool take_b = b > a || (b == a && ib < ia);
        best = take_b ? b : a;
        bi = take_b ? ib : ia;
    }
}

// ---------------------------------------------------------------------------------------------------------
// LDS of the main kernel: img | cs64 (copies of the staged tables) | lsum [k][d], lcnt [k] u64 (ACCUM)
// ---------------------------------------------------------------------------------------------------------
template <typename T, int NM, int NF, bool ACCUM, bool WHITEN>
__global__ __launch_bounds__(1024, 4) void kmeans_filter_kernel(
    const T* __restrict__ x, int64_t n, int d, int64_t ld, int k, const double* __restrict__ mean,
    const double* __restrict__ stdv, const uint4* __restrict__ image, const double* __restrict__ centers,
    int32_t* __restrict__ labels,
    double* __restrict__ mindist, const FitState* __restrict__ st, unsigned long long* __restrict__ sums,
    unsigned long long* __restrict__ counts, unsigned long long* __restrict__ n_scanned, int stagger) {
    using S = FilterShape<NM>;
    constexpr int kMT = 1024, DP = S::DP, D1 = S::D1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if constexpr (ACCUM) {
        if (st->done != 0.0) return;
    }
    const int k16 = (k + 15) & ~15;
    const int n_tiles = ((k16 / 16) + 1) & ~1;            // even: the loop takes tile pairs
    uint4* img = reinterpret_cast<uint4*>(smem_raw);
    double* cs64 = reinterpret_cast<double*>(img + (size_t)n_tiles * NM * 64);
    unsigned long long* lsum = reinterpret_cast<unsigned long long*>(cs64 + (size_t)n_tiles * 16 * D1);
    unsigned long long* lcnt = lsum + (size_t)k * d;
    __shared__ int unit_ctr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j16 = lane & 15, q = lane >> 4;
    const double scale = ACCUM ? st->scale : 0.0;
    if (tid == 0) unit_ctr = kMT / 64;
    KSTAMP_INIT
    if constexpr (ACCUM) {
        for (int i = tid; i < k * (d + 1); i += kMT) lsum[i] = 0ull;
    }
    // ---- centre tables, built by every workgroup for itself: wave w stages tiles w, w + 16, ... straight into the
    // LDS (a separate staging launch + 113 KB of copies per workgroup did the same for 4.7 us more per pass)
    __shared__ int tile_flag[64];
    {
        // padding tiles too (their rows carry the -inf sentinel); two tiles per trip, both fetched before either is built
#ifdef MSM_KMF_DIAG_NOSTAGE   // timing experiment only: tables left unbuilt (wrong results)
        for (int t = wave; t < 0; t += 2 * (kMT / 64)) {
#else
        for (int t = wave; t < n_tiles; t += 2 * (kMT / 64)) {
#endif
            const int t2 = t + kMT / 64;
            double c0[kStageRegs], c1[kStageRegs];
            filter_stage_fetch(t, lane, centers, k, d, c0);
            if (t2 < n_tiles) filter_stage_fetch(t2, lane, centers, k, d, c1);
            filter_stage_build<NM>(t, lane, reinterpret_cast<unsigned short*>(img + (size_t)t * NM * 64), c0, k, d, nullptr, cs64,
                                   tile_flag);
            if (t2 < n_tiles)
                filter_stage_build<NM>(t2, lane, reinterpret_cast<unsigned short*>(img + (size_t)t2 * NM * 64), c1, k, d, nullptr,
                                       cs64, tile_flag);
        }
    }
    __syncthreads();
    int bad_tiles = 0;
    for (int t = 0; t < n_tiles; ++t) bad_tiles |= tile_flag[t];
    const bool all_scan = bad_tiles != 0;
    __syncthreads();
    KSTAMP(0);

    const int64_t frames_per_wave = 16 * NF;
    const int64_t n_units = (n + frames_per_wave - 1) / frames_per_wave;
    const int64_t units_per_block = (n_units + gridDim.x - 1) / gridDim.x;
    const int64_t u_begin = (int64_t)blockIdx.x * units_per_block;
    const int64_t u_end = min(n_units, u_begin + units_per_block);
    unsigned long long my_scans = 0;
    // 16-byte loads of a frame's coordinates: fp64 rows of exactly DP features on 16-byte boundaries
    const bool vec_rows = sizeof(T) == 8 && d == DP && ((ld * sizeof(T)) & 15) == 0 && (((uintptr_t)x) & 15) == 0;

    // pinned fp64 score of table row `crow` for the frame whose coordinates are z[]: the ascending-feature chain
    auto score = [&](const double* crow, const double (&z)[DP]) {
        double a = 0.0;
        const double2* c2 = reinterpret_cast<const double2*>(crow);
#pragma unroll
        for (int f2 = 0; f2 < DP / 2; ++f2) {
            const double2 c = c2[f2];
            a = fma(c.x, z[2 * f2], a);
            a = fma(c.y, z[2 * f2 + 1], a);
        }
        return a - crow[DP];
    };
    auto load_frame = [&](int64_t t, double (&z)[DP], auto vec_tag) {
        const T* row = x + t * ld;
        if constexpr (decltype(vec_tag)::value) {
            const double2* r2 = reinterpret_cast<const double2*>(row);
#pragma unroll
            for (int f2 = 0; f2 < DP / 2; ++f2) {
                const double2 v = r2[f2];
                z[2 * f2] = v.x;
                z[2 * f2 + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int f = 0; f < DP; ++f) {
                const double v = load_as_f64(row + (f < d ? f : d - 1));
                z[f] = f < d ? v : 0.0;
            }
        }
        if constexpr (WHITEN) {
#pragma unroll
            for (int f = 0; f < DP; ++f) {
                const int fc = f < d ? f : d - 1;
                const double w = (z[f] - mean[fc]) / stdv[fc];
                z[f] = f < d ? w : 0.0;
            }
        }
    };
    auto write_label = [&](int64_t t, int bidx, double bm, const double (&z)[DP]) {
        labels[t] = bidx;
        if (mindist) {
            double zsq = 0.0;
#pragma unroll
            for (int f = 0; f < DP; ++f) zsq = fma(z[f], z[f], zsq);
            const double md = -2.0 * bm + zsq;
            mindist[t] = md > 0.0 ? md : 0.0;
        }
    };

    // The four waves of a SIMD (w, w + 4, w + 8, w + 12) would otherwise run their tile loops together and their
    // refinements together, leaving the matrix pipe idle half of the time: start them a fraction of a unit apart
    for (int i = 0; i < (wave >> 2) * stagger; ++i) __builtin_amdgcn_s_sleep(8);
    for (int64_t unit = u_begin + wave; unit < u_end;) {
        int nt = 0;
        if (lane == 0) nt = atomicAdd(&unit_ctr, 1);
        const int64_t nxt = u_begin + __builtin_amdgcn_readfirstlane(nt);
        int64_t fidx[NF];
        v8bf b[NF][NM];
#pragma unroll
        for (int u = 0; u < NF; ++u) {
            fidx[u] = unit * frames_per_wave + 16 * u + j16;
            const int64_t t = fidx[u] < n ? fidx[u] : n - 1;
#pragma unroll
            for (int m = 0; m < NM; ++m) b[u][m] = __builtin_bit_cast(v8bf, image[t * (4 * NM) + 4 * m + q]);
        }
        KSTAMP_VM(1);
        // ---- filter: pair maxima of the upper bounds, top two per lane
        float b1[NF], b2[NF];
        int bp[NF];
#pragma unroll
        for (int u = 0; u < NF; ++u) { b1[u] = -__builtin_inff(); b2[u] = -__builtin_inff(); bp[u] = 0; }
#ifdef MSM_KMF_DIAG_NOTILE   // timing experiments only (tools/build_variant.sh): wrong results
        for (int jt = 0; jt < 2; jt += 2) {
#else
        for (int jt = 0; jt < n_tiles; jt += 2) {
#endif
            v8bf aa[NM], ab[NM];
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                aa[m] = __builtin_bit_cast(v8bf, img[((jt + 0) * NM + m) * 64 + lane]);
                ab[m] = __builtin_bit_cast(v8bf, img[((jt + 1) * NM + m) * 64 + lane]);
            }
            v4f32 acca[NF], accb[NF];
#pragma unroll
            for (int u = 0; u < NF; ++u) {
                acca[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aa[0], b[u][0], (v4f32){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                accb[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[0], b[u][0], (v4f32){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            }
            if constexpr (NM == 2) {
#pragma unroll
                for (int u = 0; u < NF; ++u) {
                    acca[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aa[1], b[u][1], acca[u], 0, 0, 0);
                    accb[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[1], b[u][1], accb[u], 0, 0, 0);
                }
            }
#ifdef MSM_KMF_DIAG_NOTOP2
#pragma unroll
            for (int u = 0; u < NF; ++u) asm volatile("" ::"v"(acca[u]), "v"(accb[u]));
            if (jt == 0)
#endif
#pragma unroll
            for (int u = 0; u < NF; ++u) {
                // m = max(b2, pair maximum): b2 <= b1, so the top-two update below is the same as with the bare
                // pair maximum, and every v_max3 takes an already canonical first operand
                float m = max3_f32(b2[u], acca[u][0], acca[u][1]);
                m = max3_f32(m, acca[u][2], acca[u][3]);
                m = max3_f32(m, accb[u][0], accb[u][1]);
                m = max3_f32(m, accb[u][2], accb[u][3]);
                const bool better = m > b1[u];
                b2[u] = med3_f32(b1[u], b2[u], m);
                b1[u] = hw_max_f32(b1[u], m);
                bp[u] = better ? jt : bp[u];
            }
        }
        KSTAMP(2);
        // ---- the winning lane, its pair and the bound R on everything outside its 8 candidates, per frame
        // lane (q, j16) keeps the values of group u = q, the frame it refines below (selected here, one group at a
        // time: picked out of per-group arrays afterwards, the arrays were indexed by q and went to scratch memory --
        // 64 bytes per frame of extra HBM writes)
        int cd = 0x7fffffff;
        float Ru = __builtin_inff();
#pragma unroll
        for (int u = 0; u < NF; ++u) {
            const float M1 = xrow_max_f32(b1[u]);
            // lowest holder lane q and its pair, in one minimum: (q << 16) | pair
            const int code_u = xrow_min_i32(b1[u] == M1 ? ((q << 16) | bp[u]) : 0x7fffffff);
            const int gs = code_u >> 16;
            float r = q == gs ? b2[u] : b1[u];
            // range guard of the frame: the last slot of its image row (lane q = 3 of the last instruction) is +inf;
            // an infinite R refuses the certificate
            if (q == 3 && __builtin_bit_cast(unsigned short, b[u][NM - 1][7]) == 0x7F80) r = __builtin_inff();
            const float R_u = xrow_max_f32(r);
            cd = q == u ? code_u : cd;
            Ru = q == u ? R_u : Ru;
        }
        KSTAMP(3);
        // ---- refinement, one frame per lane: lane (q, j16) takes frame j16 of group u = q -- its own eight
        // candidates (rows 4 gs .. 4 gs + 3 of both tiles of the winning pair) scored one after the other with the
        // pinned fp64 chain, no cross-lane arg-max, one pass over the unit instead of one per group (the four lanes
        // of a frame used to score two rows each and settle the winner by butterflies, 16 frames at a time).
        // (one copy of the code per row-load flavour: chosen inside, the two load sequences met in front of the
        // scoring chain and the wait counts there fell back to draining everything in flight)
        static_assert(NF <= 4, "one frame group per lane quarter");
        auto refine_unit = [&](auto vec_tag) {
            const int64_t f0 = unit * frames_per_wave + lane;   // = fidx[q]
            const bool fok = q < NF && f0 < n;
            const int64_t fr = fok ? f0 : n - 1;
            // delta mode: the previous label goes out BEFORE the coordinates -- loads return in order
            int old = -1;
            if constexpr (ACCUM) {
                if (labels) old = labels[fr];
            }
            double z[DP];
            load_frame(fr, z, vec_tag);
            const int gs = cd >> 16, pstar = cd & 0xffff;
            const int base = min(pstar * 16 + 4 * gs, n_tiles * 16 - 20);   // (the clamp only meets the "no candidate" code)
            double best = score(cs64 + (size_t)base * D1, z);
            int bi = base;
#pragma unroll
            for (int c = 1; c < 8; ++c) {   // ascending row index: ties keep the lower one
                const int row = base + (c & 3) + 16 * (c >> 2);
                const double s = score(cs64 + (size_t)row * D1, z);
                if (s > best) { best = s; bi = row; }
            }
            const bool certified = !all_scan && cd != 0x7fffffff && bi < k && best > (double)Ru;   // false for NaN
            unsigned long long todo = __ballot(fok && !certified);   // one bit per frame
            KSTAMP(4);
            if (fok && certified) {
                if constexpr (ACCUM) {
                    // delta mode (labels != NULL): the sums follow the frames that CHANGED centre since the last pass
                    // (integer sums: the same bits as a full re-accumulation); else every frame is added
                    if (!labels || old != bi) {
#pragma unroll
                        for (int f = 0; f < DP; ++f) {
                            if (f < d) {
                                const unsigned long long fx = (unsigned long long)to_fixed(z[f], scale);
                                atomicAdd(&lsum[(size_t)bi * d + f], fx);
                                if (old >= 0) atomicAdd(&lsum[(size_t)old * d + f], 0ull - fx);
                            }
                        }
                        atomicAdd(&lcnt[bi], 1ull);
                        if (old >= 0) atomicAdd(&lcnt[old], ~0ull);
                        if (labels) labels[f0] = bi;
                    }
                } else {
                    write_label(f0, bi, best, z);
                }
            }
            KSTAMP(5);
            // ---- step 4: the wave scans all centres for each frame left over, one lane per centre
            while (todo) {
                const int jf = __builtin_ctzll(todo);
                todo &= todo - 1;
                const int64_t t = ((int64_t)__builtin_amdgcn_readlane((int)(f0 >> 32), jf) << 32) |
                                  (unsigned)__builtin_amdgcn_readlane((int)f0, jf);
                double zz[DP];
                load_frame(t, zz, vec_tag);
                double sbest = -__builtin_inf();
                int sbi = 0x7fffffff;
                for (int c = lane; c < k; c += 64) {
                    const double s = score(cs64 + (size_t)c * D1, zz);
                    if (s > sbest) { sbest = s; sbi = c; }   // ascending c per lane: the first maximum stays
                }
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const double ob = __shfl_xor(sbest, off, 64);
                    const int oi = __shfl_xor(sbi, off, 64);
                    if (ob > sbest || (ob == sbest && oi < sbi)) { sbest = ob; sbi = oi; }
                }
                if (sbi >= k) { sbi = 0; sbest = -__builtin_inf(); }   // every score NaN: label 0, as the fp64 kernel
                if constexpr (ACCUM) {
                    const int old_s = labels ? labels[t] : -1;
                    if (!labels || old_s != sbi) {
                        double v = zz[0];                                   // lane f adds feature f
#pragma unroll
                        for (int f = 1; f < DP; ++f) v = lane == f ? zz[f] : v;
                        if (lane < d) {
                            const unsigned long long fx = (unsigned long long)to_fixed(v, scale);
                            atomicAdd(&lsum[(size_t)sbi * d + lane], fx);
                            if (old_s >= 0) atomicAdd(&lsum[(size_t)old_s * d + lane], 0ull - fx);
                        }
                        if (lane == 0) {
                            atomicAdd(&lcnt[sbi], 1ull);
                            if (old_s >= 0) atomicAdd(&lcnt[old_s], ~0ull);
                            if (labels) labels[t] = sbi;
                        }
                    }
                } else {
                    if (lane == 0) write_label(t, sbi, sbest, zz);
                }
                ++my_scans;
            }
            KSTAMP(6);
        };
#ifndef MSM_KMF_DIAG_NOREFINE
        if (vec_rows) refine_unit(std::true_type{});
        else refine_unit(std::false_type{});
#else
        if (lane == 0 && cd == 12345) labels[0] = (int)Ru;
#endif
        unit = nxt;
    }
    KSTAMP(7);
    if (n_scanned && lane == 0 && my_scans) atomicAdd(n_scanned, my_scans);
#ifndef MSM_KMF_DIAG_NOFLUSH
    if constexpr (ACCUM) {
        __syncthreads();
        for (int i = tid; i < k * d; i += kMT)
            if (lsum[i]) atomicAdd(&sums[i], lsum[i]);
        for (int i = tid; i < k; i += kMT)
            if (lcnt[i]) atomicAdd(&counts[i], lcnt[i]);
    }
#endif
    KSTAMP_FLUSH
}

// LDS bytes of the filter kernel; 0 when the shape does not fit (the fp64 kernel runs instead).  The member
// sums of an accumulate pass must fit the LDS too.
static inline size_t filter_lds_bytes(int k, int d, bool accum) {
    if (d > kFilterMaxD) return 0;
    const int nm = filter_nm(d);
    const int k16 = (k + 15) & ~15;
    const int n_tiles = ((k16 / 16) + 1) & ~1;
    const size_t tile_bytes = nm == 1 ? FilterShape<1>::kTileBytes : FilterShape<2>::kTileBytes;
    const size_t total = (size_t)n_tiles * tile_bytes + (accum ? (size_t)k * (d + 1) * sizeof(unsigned long long) : 0);
    const size_t cap = 160 * 1024 - 256;   // static __shared__ words of the kernel
    return total <= cap ? total : 0;
}
