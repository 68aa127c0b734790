// Certified bf16 matrix-core FILTER + certified fp32 candidate pick for the k-means score table (d <= 10).
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
//                               MI355X, tools/probe/bf16_filter_probe.hip and tests/test_gpu_mfma_rule.py: terms
//                               below 2^-24 of the largest one vanish, equal terms at 2^-24 survive; the last bits
//                               depend on the order of the slots (groups of products are added), the bound holds
//                               for every order; worst observed error 11.7 x 2^-24 of the largest term against
//                               the bound 33)
//      fp64 chain of m_j     <= 12 x 2^-53 S_j
//    and kappa = 80 x 2^-24 covers their sum with 7 x 2^-24 S_j to spare.  With d <= 4 everything fits ONE
//    instruction (28 slots): accumulation <= 34.3 x 2^-24 S_j, kappa = 44 x 2^-24.
// 2. Per lane (4 accumulator rows of a frame) only the largest PAIR maximum of u, the runner-up pair maximum and
//    the pair index are tracked (max3 tree, med3, max, compare, select: 8 VALU per 8 scores).
// 3. CANDIDATE PICK.  The 8 centres of the winning lane's winning pair are scored in fp32 from an fp32 copy of the
//    centre table (s_j: the same chain in fp32; |s_j - exact_j| <= 14.1 x 2^-24 S_j, and S_j <= |z||c_j| + h_j),
//    E = 20 x 2^-24 max_j (max(|z|, 1) |c_j| + h_j) + 1e-30.  With jw = arg-max s_j and R = the largest u outside the 8
//    (runner-up pair of the winning lane, best pairs of the frame's other three lanes): if s_jw - E exceeds
//    every other s_j + E AND R, then m_jw > m_j for every other centre: the label is jw, exactly as the
//    all-fp64 scan gives it, and no fp64 arithmetic was needed.  (The pinned distance, when asked for, is the fp64
//    chain of that one centre.)
// 4. Otherwise (about 0.2 % of the frames: near-ties, duplicate centres, NaN / out-of-range input) the wave scores ALL
//    centres for that frame in fp32 from the same LDS table, lane l rows l, l + 64, ...: no centre with
//    s_j < max s - 2 E can hold the pinned maximum, so when one centre is left it is the label; else the pinned fp64
//    scores of the few rows in that band decide (lowest index on ties, as the all-fp64 scan).  Only frames or tables
//    that fail the range guard take the plain scan of all centres in fp64 (rows from global memory).
//
// Range: coordinates with 0 < |v| < 1e-14 are LEFT OUT of the bf16 images (their products could underflow inside
// the matrix instruction) and the bound pays for them: a frame adds (sum of its tiny |x_f|) / kappa to its |x|
// slot, a centre adds the sum of its tiny |c_jf| to its kappa |c_j| slot -- either covers the dropped products.
// Frames or centres with |v| > 1e18, inf or NaN fail the guard: the frame (for a centre: every frame) takes step 4.
//
// Schedule.  One workgroup of kFilterWaves waves per CU builds the centre tables in its LDS and then takes units of
// 64 frames: tile loop (matrix pipe + top-two bookkeeping), the loads of the next unit's images and of this unit's
// coordinates, the cross-lane step, the candidate pick, the commit, step 4 for what is left.  (Step 4 as a scan of all
// centres in fp64, rows from global memory, cost 25-30 us per pass for 0.2 % of the frames, in place or queued for the
// end of the workgroup's units alike: ~10 us of a wave per frame, and the slowest workgroup has 20 of them.  Scoring
// the eight candidates in fp64 before step 4, one lane and one row at a time, halves the frames that reach step 4 and
// still loses 6 us per pass: eight dependent trips to the L2 per occurrence.)  What the SIMD can do (tools/probe/bf16_mix_probe.hip): a matrix instruction keeps the
// matrix pipe for 16 cycles and the VALU port for 8, a VALU instruction the port for 4; the tile loop (16 + 34 per
// iteration) is balanced between the two.  Tried and dropped: scoring the candidates of unit i - 1 inside the tile
// loop of unit i (same time: the port is the limit either way, and the state costs 12 of 16 waves).
#pragma once

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v4f32 __attribute__((ext_vector_type(4)));

// waves per workgroup (= per CU): the two-instruction shapes need ~150 VGPRs to keep the next unit's images in flight
// without spilling (a kernel with ANY scratch memory ran 25 us longer per pass), the one-instruction shapes fit 128
#ifdef MSM_KMF_WAVES
__host__ __device__ constexpr int filter_waves(int) { return MSM_KMF_WAVES; }
#else
__host__ __device__ constexpr int filter_waves(int nm) { return nm == 1 ? 16 : 12; }
#endif
constexpr int kFilterMaxD = 10;
constexpr double kFilterTiny = 1e-14, kFilterHi = 1e18;
constexpr float kPickEps = 20.0f * 5.9604644775390625e-08f;   // 20 x 2^-24
constexpr float kPickFloor = 1e-30f;
constexpr float kUp20 = 1.0f + 9.5367431640625e-07f;          // 1 + 2^-20

__host__ __device__ constexpr int filter_nm(int d) { return 6 * d + 4 <= 32 ? 1 : 2; }   // K = 32 instructions per tile
__host__ __device__ constexpr int filter_rowq(int nm) { return nm == 1 ? 4 : 5; }         // uint4 per frame image
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
// product term t pairs these split parts (0 = high, 1 = middle, 2 = low)
__device__ __forceinline__ int filter_part_c(int t) { return t == 1 || t == 3 ? 1 : (t == 4 ? 2 : 0); }
__device__ __forceinline__ int filter_part_x(int t) { return t == 2 || t == 3 ? 1 : (t == 5 ? 2 : 0); }

// K slot of product term t, feature f.  NM = 1: t d + f, then three slots for h and slot 31 for the norms.
// NM = 2: the slots are ordered so that a lane's 8 slots (quarter q of instruction m) need ONE 16-byte piece of
// a 80-byte frame image  [xh0..7 | xm0..7 | xl0..7 | xh8 xh9 xh8 xh9 xh8 xh9 xm8 xm9 | xm8 xm9 xl8 xl9 1 1 1 |x|]:
//   m = 0: q0 = ch xh, q1 = cm xh, q2 = cl xh (all three read piece 0), q3 = ch xm (piece 1)
//   m = 1: q0 = cm xm (piece 1), q1 = ch xl (piece 2), q2 = features 8, 9 of terms 0, 1, 4, 2 (piece 3),
//          q3 = features 8, 9 of terms 3, 5, then -(1 - kappa) h_j in three parts and kappa |c_j| (piece 4)
// (the expanded image was 128 bytes per frame and pass; the error bound of the instruction holds for any slot order).
template <int NM>
__host__ __device__ constexpr int filter_slot(int t, int f, int d) {
    if (NM == 1) return t * d + f;
    constexpr int lo8[6] = {0, 8, 24, 32, 16, 40};      // term -> first slot of features 0..7
    constexpr int hi2[6] = {48, 50, 54, 56, 52, 58};    // term -> first slot of features 8, 9
    return f < 8 ? lo8[t] + f : hi2[t] + (f - 8);
}
template <int NM>
__host__ __device__ constexpr int filter_slot_h(int d) { return NM == 1 ? 6 * d : 60; }   // three slots, then ...
template <int NM>
__host__ __device__ constexpr int filter_slot_norm() { return 32 * NM - 1; }
// piece of the frame image read by quarter q of instruction m
template <int NM>
__device__ __forceinline__ int filter_piece(int q, int m) {
    if (NM == 1) return q;
    return m == 0 ? (q < 3 ? 0 : 1) : q + 1;
}

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
// Frame images: row t = the B operands of frame t (filter_rowq uint4: 64 bytes for d <= 4, 80 bytes for d <= 10),
// layout as filter_slot; the |x| slot is rounded up (it meets kappa |c_j|) and is +inf when the frame fails the
// range guard.  One thread per frame.
// ---------------------------------------------------------------------------------------------------------
// Both sides of the kernel go through the LDS so that the memory system only ever sees consecutive lanes on
// consecutive 16-byte pieces: a lane reading its own 8 D-byte row or writing its own image row touched 64
// cache lines per instruction.  `dense`: rows back to back (ld == D) on a 16-byte boundary.
template <typename T, int D>
__global__ __launch_bounds__(256) void kmeans_pack_kernel(const T* __restrict__ x, int64_t n, int64_t ld,
                                                         const double* __restrict__ mean,
                                                         const double* __restrict__ stdv, uint4* __restrict__ image,
                                                         int dense) {
    constexpr int NM = filter_nm(D);
    constexpr int RQ = filter_rowq(NM);
    constexpr int kOutStride = 5;                       // uint4 per staged image row (odd: bank spread)
    constexpr int kInBytes = 256 * D * (int)sizeof(T);  // a workgroup's rows, back to back
    constexpr int kOutBytes = 256 * kOutStride * 16;
    __shared__ __attribute__((aligned(16))) unsigned char stage[kOutBytes > kInBytes ? kOutBytes : kInBytes];
    const int64_t t0 = (int64_t)blockIdx.x * 256;
    const int64_t t = t0 + threadIdx.x;
    const int rows = (int)(n - t0 < 256 ? n - t0 : 256);
    // the image is padded to whole units of 64 rows: rows beyond n are written as frames that fail the range guard
    const int64_t n_pad = (n + 63) & ~(int64_t)63;
    const int rows_out = (int)(n_pad - t0 < 256 ? n_pad - t0 : 256);
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
    double q = 0.0, tiny = 0.0;
    bool ok = t < n;
#pragma unroll
    for (int f = 0; f < D; ++f) {
        if (mean) v[f] = (v[f] - mean[f]) / stdv[f];
        ok = ok && fabs(v[f]) <= kFilterHi;   // false for NaN
    }
#pragma unroll
    for (int f = 0; f < D; ++f) {
        const double a = fabs(v[f]);
        const bool keep = ok && a >= kFilterTiny;
        if (ok && !keep) tiny += a;            // left out of the image, paid for in the |x| slot
        const double w = keep ? v[f] : 0.0;
        q = fma(w, w, q);
        bf16_split3(w, part[f]);
    }
    const unsigned xn = ok ? bf16_up((float)((sqrt(q) + tiny / filter_kappa(NM)) * (1.0 + 9.5367431640625e-07))) : 0x7F80u;
    unsigned slots[8 * RQ];
#pragma unroll
    for (int s = 0; s < 8 * RQ; ++s) slots[s] = 0;
    if constexpr (NM == 1) {
#pragma unroll
        for (int tt = 0; tt < 6; ++tt)
#pragma unroll
            for (int f = 0; f < D; ++f) slots[tt * D + f] = part[f][tt == 2 || tt == 3 ? 1 : (tt == 5 ? 2 : 0)];
        slots[6 * D] = slots[6 * D + 1] = slots[6 * D + 2] = 0x3F80u;   // 1.0 against the three parts of -(1 - kappa) h_j
        slots[31] = xn;                                                  // |x| rounded up against kappa |c_j|
    } else {
#pragma unroll
        for (int f = 0; f < D; ++f) {
            if (f < 8) {
                slots[f] = part[f][0];
                slots[8 + f] = part[f][1];
                slots[16 + f] = part[f][2];
            } else {
                const int g = f - 8;
                slots[24 + g] = slots[26 + g] = slots[28 + g] = part[f][0];
                slots[30 + g] = part[f][1];
                slots[32 + g] = part[f][1];
                slots[34 + g] = part[f][2];
            }
        }
        slots[36] = slots[37] = slots[38] = 0x3F80u;
        slots[39] = xn;
    }
    uint4* mine = reinterpret_cast<uint4*>(stage) + threadIdx.x * kOutStride;
#pragma unroll
    for (int c = 0; c < RQ; ++c)
        mine[c] = make_uint4(slots[8 * c + 0] | (slots[8 * c + 1] << 16), slots[8 * c + 2] | (slots[8 * c + 3] << 16),
                             slots[8 * c + 4] | (slots[8 * c + 5] << 16), slots[8 * c + 6] | (slots[8 * c + 7] << 16));
    __syncthreads();
    uint4* dst = image + t0 * RQ;
    for (int c = threadIdx.x; c < rows_out * RQ; c += 256)
        dst[c] = reinterpret_cast<const uint4*>(stage)[(c / RQ) * kOutStride + (c % RQ)];
}

// ---------------------------------------------------------------------------------------------------------
// Centre side, once per launch and workgroup (one wave per 16-centre tile) into the kernel's LDS:
//   img   [n_tiles][NM][64] uint4       A operands, lane-major per instruction
//   tab   [n_tiles * 16][RF] f32        fp32 centre coordinates zero-padded to DP features, then h_j, then |c_j|
//                                       rounded up (padding rows: 0, 3e38, -3e38), rows 16-byte aligned: step 3
//   hs    [n_tiles * 16] f64            h_j (+inf for padding rows): step 4 and the pinned distance
//   any_bad                             a centre failed the range guard: every frame takes the exhaustive scan
// DP = 4 (NM = 1) or 10 (NM = 2).
// ---------------------------------------------------------------------------------------------------------
template <int NM>
struct FilterShape {
    static constexpr int DP = NM == 1 ? 4 : 10;
    static constexpr int RF = NM == 1 ? 8 : 12;                          // floats per table row
    static constexpr int kTileBytes = NM * 1024 + 16 * RF * 4 + 16 * 8;  // image + table rows + h of one 16-centre tile
};

// One wave stages one 16-centre tile in two steps, so that a wave with several tiles has the global loads of all of
// them in flight before it builds the first (the build is ~300 instructions; one load round trip under load is as long):
//   filter_stage_fetch: the tile's 16 x d coordinates, one coalesced load per 64 elements (element e = row e / d,
//                       feature e % d), at most kStageRegs per lane;
//   filter_stage_build: bf16 triples into the A operands, fp32 coordinates into the table rows; the 16 row lanes then
//                       fetch their row once more for h_j, the range guard and the kappa slots (the fp64 chain over
//                       ascending features: the bits of every other h_j in the library).
constexpr int kStageRegs = (16 * kFilterMaxD + 63) / 64;
// i / d for 0 <= i < 256, 1 <= d <= 10 without the integer-division sequence: (i + 0.5) / d is at least 0.05 away from
// every integer, far outside fp32 rounding
__device__ __forceinline__ int stage_row(int i, int d) { return (int)(((float)i + 0.5f) * (1.0f / (float)d)); }
template <int DP>
__device__ __forceinline__ void filter_stage_fetch(int tile, int lane, const double* __restrict__ centers, int k, int d,
                                                   double (&c)[kStageRegs], double (&rc)[DP]) {
#pragma unroll
    for (int u = 0; u < kStageRegs; ++u) {
        const int i = lane + 64 * u;
        const int r = stage_row(i, d), j = tile * 16 + r;
        c[u] = (i < 16 * d && j < k) ? centers[(size_t)tile * 16 * d + i] : 0.0;
    }
    // the 16 row lanes: their own row once more, feature by feature (zeros beyond d leave every chain unchanged)
    const int jr = tile * 16 + lane;
#pragma unroll
    for (int f = 0; f < DP; ++f) rc[f] = (lane < 16 && jr < k && f < d) ? centers[(size_t)jr * d + f] : 0.0;
}
template <int NM>
__device__ __forceinline__ void filter_stage_build(int tile, int lane, unsigned short* simg, const double (&c)[kStageRegs],
                                                   const double (&rc)[FilterShape<NM>::DP], int k, int d,
                                                   float* __restrict__ tab, double* __restrict__ hs, int* __restrict__ any_bad) {
    using S = FilterShape<NM>;
    constexpr double kappa = filter_kappa(NM);
    for (int i = lane; i < NM * 64 * 4; i += 64) reinterpret_cast<unsigned*>(simg)[i] = 0u;
    for (int i = lane; i < 16 * S::RF; i += 64) tab[(size_t)tile * 16 * S::RF + i] = 0.0f;   // pads beyond d
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
            tab[(size_t)j * S::RF + f] = (float)c[u];
            if (fabs(c[u]) >= kFilterTiny) {            // tiny coordinates stay out of the image (see the header)
                unsigned cp[3];
                bf16_split3(c[u], cp);
#pragma unroll
                for (int t = 0; t < 6; ++t) simg[slot_addr(r, filter_slot<NM>(t, f, d))] = (unsigned short)cp[filter_part_c(t)];
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    int bad_row = 0;
    if (lane < 16) {
        const int j = tile * 16 + lane;
        float* row = tab + (size_t)j * S::RF;
        double h = __builtin_inf();
        if (j < k) {
            double a = 0.0, akeep = 0.0, tiny = 0.0;
            bool ok = true;
#pragma unroll
            for (int f = 0; f < S::DP; ++f) {
                const double cf = rc[f];   // zero beyond d
                a = fma(cf, cf, a);
                const double ac = fabs(cf);
                ok = ok && ac <= kFilterHi;              // false for NaN
                if (ac >= kFilterTiny) akeep = fma(cf, cf, akeep);
                else tiny += ac;
            }
            h = 0.5 * a;
            if (!ok) bad_row = 1;
            unsigned hsp[3];
            bf16_split3(-(h - kappa * h), hsp);
            const int sh = filter_slot_h<NM>(d);
            simg[slot_addr(lane, sh)] = (unsigned short)hsp[0];
            simg[slot_addr(lane, sh + 1)] = (unsigned short)hsp[1];
            simg[slot_addr(lane, sh + 2)] = (unsigned short)hsp[2];
            simg[slot_addr(lane, filter_slot_norm<NM>())] =
                (unsigned short)bf16_up((float)((kappa * sqrt(akeep) + tiny) * (1.0 + 9.5367431640625e-07)));
            row[S::DP] = (float)h;
            row[S::DP + 1] = (float)(sqrt(a) * (1.0 + 9.5367431640625e-07));
        } else {
            simg[slot_addr(lane, filter_slot_h<NM>(d))] = 0xFF7F;   // -3.4e38 x 1.0: a padding centre never holds a maximum
            row[S::DP] = 3.0e38f;        // score -3e38 ...
            row[S::DP + 1] = -3.0e38f;   // ... and an S bound <= 0 (the pick multiplies this by max(|z|, 1))
        }
        hs[j] = h;
    }
    const bool bad = __any(bad_row != 0);
    if (bad && lane == 0) atomicOr(any_bad, 1);
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
__device__ __forceinline__ float xrow_max_f32(float x) {   // operands are never NaN here
    return __uint_as_float(xrow_reduce_u32(__float_as_uint(x), [](unsigned a, unsigned b) {
        return __float_as_uint(hw_max_f32(__uint_as_float(a), __uint_as_float(b)));
    }));
}
__device__ __forceinline__ int xrow_min_i32(int x) {
    return (int)xrow_reduce_u32((unsigned)x, [](unsigned a, unsigned b) { return (unsigned)min((int)a, (int)b); });
}
// reductions over the whole wave on the VALU: four DPP steps inside the rows of 16 lanes (quad_perm [1,0,3,2],
// quad_perm [2,3,0,1], row_half_mirror, row_mirror), then the two row swaps -- every lane ends with the result
// (a butterfly of __shfl_xor is six dependent trips through the LDS crossbar, ~1 us)
template <typename Op>
__device__ __forceinline__ unsigned row_reduce_u32(unsigned x, Op op) {
    x = op(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false));
    x = op(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, false));
    x = op(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, false));
    x = op(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, false));
    return x;
}
__device__ __forceinline__ float wave_max_f32(float x) {   // operands are never NaN here
    const unsigned r = row_reduce_u32(__float_as_uint(x), [](unsigned a, unsigned b) {
        return __float_as_uint(hw_max_f32(__uint_as_float(a), __uint_as_float(b)));
    });
    return xrow_max_f32(__uint_as_float(r));
}
__device__ __forceinline__ int wave_min_i32(int x) {
    return xrow_min_i32((int)row_reduce_u32((unsigned)x, [](unsigned a, unsigned b) { return (unsigned)min((int)a, (int)b); }));
}
__device__ __forceinline__ double wave_max_f64(double x) {   // NaN operands lose (v_max_f64)
    auto mx = [](double a, double b) {
        double r;
        asm volatile("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    };
    auto dpp = [&](double v, auto ctrl) {
        const long long bits = __double_as_longlong(v);
        const int lo = __builtin_amdgcn_update_dpp(0, (int)bits, decltype(ctrl)::value, 0xF, 0xF, false);
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), decltype(ctrl)::value, 0xF, 0xF, false);
        return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
    };
    x = mx(x, dpp(x, std::integral_constant<int, 0xB1>{}));
    x = mx(x, dpp(x, std::integral_constant<int, 0x4E>{}));
    x = mx(x, dpp(x, std::integral_constant<int, 0x141>{}));
    x = mx(x, dpp(x, std::integral_constant<int, 0x140>{}));
#pragma unroll
    for (int step = 0; step < 2; ++step) {
        const unsigned lo = (unsigned)__double_as_longlong(x), hi = (unsigned)(__double_as_longlong(x) >> 32);
        v2u32 rl, rh;
        if (step == 0) {
            rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
            rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        } else {
            rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
            rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        }
        x = mx(__longlong_as_double((long long)(((unsigned long long)rh[0] << 32) | rl[0])),
               __longlong_as_double((long long)(((unsigned long long)rh[1] << 32) | rl[1])));
    }
    return x;
}

template <typename T, int DP, bool WHITEN>
__device__ __forceinline__ void filter_load_frame(const T* __restrict__ x, int64_t ld, int d, bool vec_rows,
                                                  const double* __restrict__ mean, const double* __restrict__ stdv,
                                                  int64_t t, double (&z)[DP]) {
    const T* row = x + t * ld;
    if (vec_rows) {
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
}

// ---------------------------------------------------------------------------------------------------------
// LDS of the main kernel: img | tab | hs | lsum [k][d], lcnt [k] u64 (ACCUM)
// centres <- sums / counts, shift2, n_iter, done: what kmeans_update_kernel (kmeans.hip) does, by one workgroup of MT
// threads, with the additions of shift2 in the order of that kernel's 1024 threads (virtual thread v takes the
// elements v, v + 1024, ...; 64 consecutive virtual threads add up by the same shuffles; the sixteen partial sums in
// order), so the two ways of closing an iteration give the same bits.  sums / counts were last written by other
// workgroups' atomics: read them with device-scope loads.
template <int MT>
__device__ __forceinline__ void filter_close_iteration(const unsigned long long* sums, const unsigned long long* counts, int k,
                                                       int d, double* centers, FitState* st) {
    __shared__ double red[16];
    const int tid = threadIdx.x;
    const double inv_scale = st->inv_scale;
    for (int v0 = 0; v0 < 1024; v0 += MT) {
        const int v = v0 + tid;
        if (v < 1024) {                 // whole waves: MT and 1024 are multiples of 64
            double acc = 0.0;
            constexpr int CH = 4;       // elements in flight per thread: the loads of a chunk before the arithmetic of any
            for (int i0 = v; i0 < k * d; i0 += 1024 * CH) {
                long long cnt[CH], sm[CH];
                double old[CH];
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const int i = i0 + c * 1024;
                    cnt[c] = 0;
                    if (i < k * d) {
                        cnt[c] = (long long)__hip_atomic_load(&counts[i / d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        sm[c] = (long long)__hip_atomic_load(&sums[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        old[c] = centers[i];
                    }
                }
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    if (cnt[c] > 0) {
                        const double c_new = (double)sm[c] * inv_scale / (double)cnt[c];
                        const double dlt = c_new - old[c];
                        acc = fma(dlt, dlt, acc);
                        centers[i0 + c * 1024] = c_new;
                    }
            }
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
            if ((v & 63) == 0) red[v >> 6] = acc;
        }
    }
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += red[i];
        st->shift2 = t;
        st->n_iter += 1.0;
        if (t <= st->tol2) st->done = 1.0;
    }
}

// ---------------------------------------------------------------------------------------------------------
template <typename T, int NM, int NF, bool ACCUM, bool WHITEN>
__global__ __launch_bounds__(64 * filter_waves(NM)) void kmeans_filter_kernel(
    const T* __restrict__ x, int64_t n, int d, int64_t ld, int k, const double* __restrict__ mean,
    const double* __restrict__ stdv, const uint4* __restrict__ image, const double* __restrict__ centers,
    int32_t* __restrict__ labels,
    double* __restrict__ mindist, const FitState* __restrict__ st, unsigned long long* __restrict__ sums,
    unsigned long long* __restrict__ counts, unsigned long long* __restrict__ n_scanned, double* upd_centers,
    unsigned int* __restrict__ ticket) {
    using S = FilterShape<NM>;
    constexpr int kFilterWaves = filter_waves(NM);
    constexpr int kMT = 64 * kFilterWaves, DP = S::DP, RF = S::RF, RQ = filter_rowq(NM);
    static_assert(NF == 4, "one frame group per lane quarter: a unit is 64 frames, one per lane in the candidate pick");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if constexpr (ACCUM) {
        if (st->done != 0.0) return;
    }
    const int k16 = (k + 15) & ~15;
    const int n_tiles = ((k16 / 16) + 1) & ~1;            // even: the loop takes tile pairs
    uint4* img = reinterpret_cast<uint4*>(smem_raw);
    float* tab = reinterpret_cast<float*>(img + (size_t)n_tiles * NM * 64);
    double* hs = reinterpret_cast<double*>(tab + (size_t)n_tiles * 16 * RF);
    unsigned long long* lsum = reinterpret_cast<unsigned long long*>(hs + (size_t)n_tiles * 16);
    unsigned long long* lcnt = lsum + (ACCUM ? (size_t)k * d : 0);
    __shared__ int unit_ctr;
    __shared__ int any_bad;
    // (the wave number through readfirstlane: the compiler then knows that unit numbers are uniform and addresses the
    // image and coordinate loads as scalar base + lane offset + immediate, not with a 64-bit register pair per load)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j16 = lane & 15, q = lane >> 4;
    const double scale = ACCUM ? st->scale : 0.0;
    if (tid == 0) {
        unit_ctr = kFilterWaves;
        any_bad = 0;
    }
    KSTAMP_INIT
    if constexpr (ACCUM) {
        for (int i = tid; i < k * (d + 1); i += kMT) lsum[i] = 0ull;
    }
    __syncthreads();
    constexpr int kUnit = 16 * NF;   // 64 frames
    const int64_t n_units = (n + kUnit - 1) / kUnit;
    const int64_t units_per_block = (n_units + gridDim.x - 1) / gridDim.x;
    const int64_t u_begin = (int64_t)blockIdx.x * units_per_block;
    const int64_t u_end = min(n_units, u_begin + units_per_block);
    // image pieces of this lane: quarter q of instruction m reads piece filter_piece(q, m) of its frame's row
    int voff[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) voff[m] = j16 * RQ + filter_piece<NM>(q, m);

    int64_t unit = u_begin + wave;
    v8bf b[NF][NM];
    // (the image holds whole units of rows, msm_kmeans_image_bytes: no clamping at the end of the shard, and the loads
    // are scalar base + lane offset + immediate)
    auto load_images = [&](int64_t un) {
        const uint4* ub = image + un * (kUnit * RQ);
#pragma unroll
        for (int u = 0; u < NF; ++u)
#pragma unroll
            for (int m = 0; m < NM; ++m) b[u][m] = __builtin_bit_cast(v8bf, ub[voff[m] + u * 16 * RQ]);
    };
    if (unit < u_end) load_images(unit);   // the first unit's images travel while the tables are built

    // ---- centre tables, built by every workgroup for itself: wave w stages tiles w, w + W, ... straight into the
    // LDS (a separate staging launch + 113 KB of copies per workgroup did the same for 4.7 us more per pass);
    // padding tiles too (their rows carry the -3.4e38 sentinel); two tiles per trip, both fetched before either is built
    for (int t = wave; t < n_tiles; t += 2 * kFilterWaves) {
        const int t2 = t + kFilterWaves;
        double c0[kStageRegs], c1[kStageRegs], r0[DP], r1[DP];
        filter_stage_fetch<DP>(t, lane, centers, k, d, c0, r0);
        if (t2 < n_tiles) filter_stage_fetch<DP>(t2, lane, centers, k, d, c1, r1);
        filter_stage_build<NM>(t, lane, reinterpret_cast<unsigned short*>(img + (size_t)t * NM * 64), c0, r0, k, d, tab, hs,
                               &any_bad);
        if (t2 < n_tiles)
            filter_stage_build<NM>(t2, lane, reinterpret_cast<unsigned short*>(img + (size_t)t2 * NM * 64), c1, r1, k, d, tab, hs,
                                   &any_bad);
    }
    __syncthreads();
    const bool all_scan = any_bad != 0;
    KSTAMP(0);

    unsigned long long my_scans = 0;
    // 16-byte loads of a frame's coordinates: fp64 rows of exactly DP features on 16-byte boundaries
    const bool vec_rows = sizeof(T) == 8 && d == DP && ((ld * sizeof(T)) & 15) == 0 && (((uintptr_t)x) & 15) == 0;
    const int iters = n_tiles / 2;
    constexpr int kNone = 0x7fffffff;

    auto load_frame = [&](int64_t t, double (&z)[DP]) { filter_load_frame<T, DP, WHITEN>(x, ld, d, vec_rows, mean, stdv, t, z); };
    // pinned fp64 score of centre c for the frame z: the ascending-feature chain; `crow` = the centre's coordinates
    auto score64 = [&](const double* crow, int c, const double (&z)[DP]) {
        double a = 0.0;
#pragma unroll
        for (int f = 0; f < DP; ++f)
            if (f < d) a = fma(crow[f], z[f], a);
        return a - hs[c];
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
    // ---- what a scan leaves behind: the label (delta mode: the move of the frame's contribution) or label + distance
    auto commit_scan = [&](int64_t t, int sbi, double sbest, const double (&zz)[DP], int old_s) {
        if constexpr (ACCUM) {
            if (!labels || old_s != sbi) {
                // lane f adds feature f (written out per feature: picked by lane number, the coordinates became an
                // array in scratch memory)
#pragma unroll
                for (int f = 0; f < DP; ++f) {
                    if (f < d && lane == f) {
                        const unsigned long long fx = (unsigned long long)to_fixed(zz[f], scale);
                        atomicAdd(&lsum[(size_t)sbi * d + f], fx);
                        if (old_s >= 0) atomicAdd(&lsum[(size_t)old_s * d + f], 0ull - fx);
                    }
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
    };
    // ---- the plain form of step 4: the pinned fp64 scores of ALL centres, lane l takes centres l, l + 64, ..., rows from
    // the global table (40 KB, L2-resident), one round trip per 64 centres: only for frames (or centre tables) outside
    // the range the fp32 scores are certified for, and for bands too crowded for the bookkeeping of the fp32 scan
    auto scan_frame = [&](int64_t t) {
        double zz[DP];
        load_frame(t, zz);
        double sbest = -__builtin_inf();
        int sbi = kNone;
        for (int c = lane; c < k; c += 64) {
            const double sc = score64(centers + (size_t)c * d, c, zz);
            if (sc > sbest) { sbest = sc; sbi = c; }   // ascending c per lane: the first maximum stays
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double ob = __shfl_xor(sbest, off, 64);
            const int oi = __shfl_xor(sbi, off, 64);
            if (ob > sbest || (ob == sbest && oi < sbi)) { sbest = ob; sbi = oi; }
        }
        if (sbi >= k) { sbi = 0; sbest = -__builtin_inf(); }   // every score NaN: label 0, as the fp64 kernel
        int old_s = -1;
        if constexpr (ACCUM) {
            if (labels) old_s = labels[t];
        }
        commit_scan(t, sbi, sbest, zz, old_s);
#ifdef MSM_KMF_DIAG_COUNT   // diagnostic builds count one kind of event: 1 = plain scans, 2 = bands resolved in fp64
        if (MSM_KMF_DIAG_COUNT == 1) ++my_scans;
#else
        ++my_scans;
#endif
    };

    while (unit < u_end) {
        int nt = 0;
        if (lane == 0) nt = atomicAdd(&unit_ctr, 1);
        const int64_t nxt = u_begin + __builtin_amdgcn_readfirstlane(nt);
        KSTAMP_VM(1);
        // ---- filter: pair maxima of the upper bounds, top two per lane
        float b1[NF], b2[NF];
        int bp[NF];
#pragma unroll
        for (int u = 0; u < NF; ++u) { b1[u] = -__builtin_inff(); b2[u] = -__builtin_inff(); bp[u] = 0; }
#ifdef MSM_KMF_DIAG_NOTILE   // timing experiments only (tools/build_variant.sh): wrong results
        for (int it = 0; it < (iters < 9 ? iters : 9); ++it) {
#else
        for (int it = 0; it < iters; ++it) {
#endif
            const int jt = 2 * it;
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
        // the range guard of this lane's frames (the |x| slot: last element of the last piece, lane quarter 3), before
        // the images of the next unit replace them
        bool guard[NF];
#pragma unroll
        for (int u = 0; u < NF; ++u) guard[u] = q == 3 && __builtin_bit_cast(unsigned short, b[u][NM - 1][7]) == 0x7F80;
        // ---- the loads of the next round go out now: the next unit's images, this unit's coordinates (lane = frame:
        // lane (q, j16) takes frame j16 of group q)
        if (nxt < u_end) load_images(nxt);
        const int64_t f0 = unit * kUnit + lane;
        const bool fok = f0 < n;
        const int64_t fr = fok ? f0 : n - 1;
        // delta mode: the previous label goes out BEFORE the coordinates -- loads return in order
        int old = -1;
        if constexpr (ACCUM) {
            if (labels) old = labels[fr];
        }
        double z[DP];
        load_frame(fr, z);
        // ---- the winning lane, its pair and the bound R on everything outside its 8 candidates, per frame
        // lane (q, j16) keeps the values of group u = q, the frame it refines (selected here, one group at a time:
        // picked out of per-group arrays afterwards, the arrays were indexed by q and went to scratch memory)
        int cd = kNone;
        float Ru = __builtin_inff();
#ifdef MSM_KMF_DIAG_NOCROSS
        cd = bp[0] | (q << 16);
        Ru = b2[0] + b1[1] + b1[2] + b1[3] + b2[1] + b2[2] + b2[3] + (guard[0] | guard[1] | guard[2] | guard[3] ? 1.f : 0.f) + (float)(bp[1] + bp[2] + bp[3]);
#else
#pragma unroll
        for (int u = 0; u < NF; ++u) {
            const float M1 = xrow_max_f32(b1[u]);
            // lowest holder lane q and its pair, in one minimum: (q << 16) | pair
            const int code_u = xrow_min_i32(b1[u] == M1 ? ((q << 16) | bp[u]) : kNone);
            const int gs = code_u >> 16;
            float r = q == gs ? b2[u] : b1[u];
            if (guard[u]) r = __builtin_inff();     // an infinite R refuses the certificate
            const float R_u = xrow_max_f32(r);
            cd = q == u ? code_u : cd;
            Ru = q == u ? R_u : Ru;
        }
#endif
        KSTAMP(3);
        // ---- step 3, one frame per lane: its eight candidates (rows 4 gs .. 4 gs + 3 of both tiles of the winning
        // pair) scored in fp32 one after the other: best and second-best score, the index of the best, the largest S bound
        const int gs = cd >> 16, pstar = cd & 0xffff;
        const int base = min(pstar * 16 + 4 * gs, n_tiles * 16 - 20);   // (the clamp only meets the "no candidate" code)
        float pz[DP];
        float qz = 0.f;
#pragma unroll
        for (int f = 0; f < DP; ++f) {
            pz[f] = (float)z[f];
            qz = __builtin_fmaf(pz[f], pz[f], qz);
        }
        const bool z_bad = !(qz < __builtin_inff());   // NaN or overflow: the bare v_max / v_med3 below drop NaN operands
        // max(|z|, 1) rounded up: the S bound of a row is |c_j| max(|z|, 1) + h_j (padding rows: (1 - this) 3e38 <= 0)
        const float zn = hw_max_f32(__builtin_amdgcn_sqrtf(qz) * kUp20, 1.0f);
        float s1 = -__builtin_inff(), s2 = -__builtin_inff(), sbmax = 0.f;
        int i1 = 0;
#ifndef MSM_KMF_DIAG_NOPICK
        const float* p_row = tab + (size_t)base * RF;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float4* r4 = reinterpret_cast<const float4*>(p_row + ((c & 3) + 16 * (c >> 2)) * RF);
            float a = 0.f, h, cn;
            if constexpr (NM == 1) {
                const float4 r0 = r4[0], r1 = r4[1];
                a = __builtin_fmaf(r0.x, pz[0], a);
                a = __builtin_fmaf(r0.y, pz[1], a);
                a = __builtin_fmaf(r0.z, pz[2], a);
                a = __builtin_fmaf(r0.w, pz[3], a);
                h = r1.x;
                cn = r1.y;
            } else {
                const float4 r0 = r4[0], r1 = r4[1], r2 = r4[2];
                a = __builtin_fmaf(r0.x, pz[0], a);
                a = __builtin_fmaf(r0.y, pz[1], a);
                a = __builtin_fmaf(r0.z, pz[2], a);
                a = __builtin_fmaf(r0.w, pz[3], a);
                a = __builtin_fmaf(r1.x, pz[4], a);
                a = __builtin_fmaf(r1.y, pz[5], a);
                a = __builtin_fmaf(r1.z, pz[6], a);
                a = __builtin_fmaf(r1.w, pz[7], a);
                a = __builtin_fmaf(r2.x, pz[8], a);
                a = __builtin_fmaf(r2.y, pz[9], a);
                h = r2.z;
                cn = r2.w;
            }
            const float s = a - h;
            sbmax = hw_max_f32(sbmax, __builtin_fmaf(zn, cn, h));
            const bool better = s > s1;             // ties keep the earlier candidate; s2 = s1 then refuses the certificate
            s2 = med3_f32(s1, s2, s);
            s1 = hw_max_f32(s1, s);
            i1 = better ? c : i1;
        }
#endif
        // one error bound for the eight: E >= E_j.  s1 - s2 > 2 E puts s_jw - E_jw above every other s_j + E_j
        const float e = __builtin_fmaf(kPickEps, sbmax, kPickFloor);
        const int jw = base + (i1 & 3) + 16 * (i1 >> 2);
        const bool certified = !all_scan && !z_bad && cd != kNone && jw < k && s1 - s2 > 2.0f * e && s1 - e > Ru;
        const int lab = jw;
#if defined(MSM_KMF_DIAG_NOSCAN) || defined(MSM_KMF_DIAG_NOQUEUE)
        unsigned long long todo = 0;
#else
        unsigned long long todo = __ballot(fok && !certified);   // one bit per frame
#endif
        KSTAMP(4);
#ifdef MSM_KMF_DIAG_NOCOMMIT
        if (fok && certified && cd == 12345) {
#else
        if (fok && certified) {
#endif
            if constexpr (ACCUM) {
                // delta mode (labels != NULL): the sums follow the frames that CHANGED centre since the last pass
                // (integer sums: the same bits as a full re-accumulation); else every frame is added
                if (!labels || old != lab) {
#pragma unroll
                    for (int f = 0; f < DP; ++f) {
                        if (f < d) {
                            const unsigned long long fx = (unsigned long long)to_fixed(z[f], scale);
                            atomicAdd(&lsum[(size_t)lab * d + f], fx);
                            if (old >= 0) atomicAdd(&lsum[(size_t)old * d + f], 0ull - fx);
                        }
                    }
                    atomicAdd(&lcnt[lab], 1ull);
                    if (old >= 0) atomicAdd(&lcnt[old], ~0ull);
                    if (labels) labels[f0] = lab;
                }
            } else {
                // the pinned distance, when asked for: the fp64 chain of the one winning centre (row from the global table)
                if (mindist) write_label(f0, lab, score64(centers + (size_t)lab * d, lab, z), z);
                else labels[f0] = lab;
            }
        }
        KSTAMP(5);
        // ---- step 4 for the frames without a certificate, one after the other, by the whole wave
        while (todo) {
            const int jf = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int64_t t = ((int64_t)__builtin_amdgcn_readlane((int)(f0 >> 32), jf) << 32) |
                              (unsigned)__builtin_amdgcn_readlane((int)f0, jf);
            const bool plain = all_scan || __builtin_amdgcn_readlane((int)(z_bad || !(Ru < __builtin_inff())), jf) != 0;
            if (plain) {
                scan_frame(t);
                continue;
            }
            // (a) fp32 scores of ALL centres from the LDS table, lane l takes rows l, l + 64, ...: per lane the best two rows
            // and the third-best score; over the wave the best score, the largest S bound and from them the band
            // [top - 2 E, top] outside of which no centre can hold the pinned maximum
            float zf[DP];
#pragma unroll
            for (int f = 0; f < DP; ++f) zf[f] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pz[f]), jf));
            const float znj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(zn), jf));
            float t1 = -__builtin_inff(), t2 = -__builtin_inff(), t3 = -__builtin_inff(), sbl = 0.f;
            int c1 = 0, c2 = 0;
            for (int c = lane; c < n_tiles * 16; c += 64) {
                const float4* r4 = reinterpret_cast<const float4*>(tab + (size_t)c * RF);
                float a = 0.f, h, cn;
                if constexpr (NM == 1) {
                    const float4 r0 = r4[0], r1 = r4[1];
                    a = __builtin_fmaf(r0.x, zf[0], a);
                    a = __builtin_fmaf(r0.y, zf[1], a);
                    a = __builtin_fmaf(r0.z, zf[2], a);
                    a = __builtin_fmaf(r0.w, zf[3], a);
                    h = r1.x;
                    cn = r1.y;
                } else {
                    const float4 r0 = r4[0], r1 = r4[1], r2 = r4[2];
                    a = __builtin_fmaf(r0.x, zf[0], a);
                    a = __builtin_fmaf(r0.y, zf[1], a);
                    a = __builtin_fmaf(r0.z, zf[2], a);
                    a = __builtin_fmaf(r0.w, zf[3], a);
                    a = __builtin_fmaf(r1.x, zf[4], a);
                    a = __builtin_fmaf(r1.y, zf[5], a);
                    a = __builtin_fmaf(r1.z, zf[6], a);
                    a = __builtin_fmaf(r1.w, zf[7], a);
                    a = __builtin_fmaf(r2.x, zf[8], a);
                    a = __builtin_fmaf(r2.y, zf[9], a);
                    h = r2.z;
                    cn = r2.w;
                }
                const float sc = a - h;
                sbl = hw_max_f32(sbl, __builtin_fmaf(znj, cn, h));
                if (sc > t1) { t3 = t2; t2 = t1; c2 = c1; t1 = sc; c1 = c; }
                else if (sc > t2) { t3 = t2; t2 = sc; c2 = c; }
                else t3 = hw_max_f32(t3, sc);
            }
            // the fp64 rows of this lane's two best centres go out now: (b) wants them one round trip later, for the
            // few lanes inside the band
            double rowa[DP], rowb[DP];
            {
                const double* ca = centers + (size_t)(c1 < k ? c1 : 0) * d;
                const double* cb = centers + (size_t)(c2 < k ? c2 : 0) * d;
#pragma unroll
                for (int f = 0; f < DP; ++f) {
                    rowa[f] = f < d ? ca[f] : 0.0;
                    rowb[f] = f < d ? cb[f] : 0.0;
                }
            }
            const float top = wave_max_f32(t1), sbw = wave_max_f32(sbl);
            const float band = top - 2.0f * __builtin_fmaf(kPickEps, sbw, kPickFloor);
            const unsigned long long in1 = __ballot(t1 >= band);
            if (__any(t3 >= band) || in1 == 0ull) {   // three of one lane's rows in the band (or NaN): the plain scan decides
                scan_frame(t);
                continue;
            }
            // the frame's fp64 coordinates and its previous label are in lane jf's registers
            double zz[DP];
#pragma unroll
            for (int f = 0; f < DP; ++f) {
                const long long bits = __double_as_longlong(z[f]);
                zz[f] = __longlong_as_double((long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(bits >> 32), jf) << 32) |
                                                         (unsigned)__builtin_amdgcn_readlane((int)bits, jf)));
            }
            const int old_t = __builtin_amdgcn_readlane(old, jf);
            int sbi;
            double sbest = 0.0;
            const bool two = __any(t2 >= band) || __builtin_popcountll(in1) > 1;
            if (!two && !(mindist && !ACCUM)) {
                sbi = __builtin_amdgcn_readlane(c1, __builtin_ctzll(in1));   // one centre in the band: it is the pinned arg-max
            } else {
                // (b) the pinned fp64 scores of the rows in the band decide (rows from the global table)
                double m1 = -__builtin_inf();
                int i1b = kNone;
                auto chain = [&](const double (&row)[DP], int c) {   // zeros beyond d leave the chain unchanged
                    double a = 0.0;
#pragma unroll
                    for (int f = 0; f < DP; ++f) a = fma(row[f], zz[f], a);
                    return a - hs[c];
                };
                if (t1 >= band && c1 < k) { m1 = chain(rowa, c1); i1b = c1; }
                if (t2 >= band && c2 < k) {
                    const double m2 = chain(rowb, c2);
                    if (m2 > m1 || (m2 == m1 && c2 < i1b)) { m1 = m2; i1b = c2; }
                }
                sbest = wave_max_f64(m1);
                sbi = wave_min_i32(m1 == sbest ? i1b : kNone);    // lowest index on ties, as the all-fp64 scan
                if (sbi >= k) {   // cannot happen for finite input; the plain scan has the rule for it
                    scan_frame(t);
                    continue;
                }
            }
            commit_scan(t, sbi, sbest, zz, old_t);
#ifdef MSM_KMF_DIAG_COUNT
            if (MSM_KMF_DIAG_COUNT == 2 && two) ++my_scans;
#else
            ++my_scans;
#endif
        }
        KSTAMP(6);
        unit = nxt;
    }
    KSTAMP(7);
    if (n_scanned && lane == 0 && my_scans) atomicAdd(n_scanned, my_scans);
    if constexpr (ACCUM) {
        __syncthreads();
        for (int i = tid; i < k * d; i += kMT)
            if (lsum[i]) atomicAdd(&sums[i], lsum[i]);
        for (int i = tid; i < k; i += kMT)
            if (lcnt[i]) atomicAdd(&counts[i], lcnt[i]);
        // The workgroup that finishes LAST closes the Lloyd iteration (centres <- sums / counts, shift, convergence flag)
        // instead of a one-workgroup launch of its own after this one: every other workgroup has then added its member
        // sums and read the old centres for the last time.
        if (upd_centers) {
            // Ordering without a device-wide fence (__threadfence() = buffer_wbl2: a write-back of the XCD's whole L2 by
            // every workgroup, +45 us per launch, measured): the member sums travel as atomics, which are resolved at
            // the device's coherence point, and every wave waits for the acknowledgement of its own (vmcnt) before the
            // workgroup takes its ticket; the closing workgroup reads them with device-scope loads, and nothing in
            // this launch has read sums / counts before, so no cache holds an older copy.
            __shared__ int is_last;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                is_last = t == gridDim.x - 1;
                if (is_last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            }
            __syncthreads();
            if (is_last) filter_close_iteration<kMT>(sums, counts, k, d, upd_centers, const_cast<FitState*>(st));
        }
    }
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
    const size_t cap = 160 * 1024 - 64;   // static __shared__ words of the kernel (two ints)
    return total <= cap ? total : 0;
}
