// Time-lagged covariance moments on the fp64 matrix cores (v_mfma_f64_16x16x4_f64).
//
// This is the one dense contraction of the path: for centred frames z_t
//     M00 = sum_t w_t z_t z_t'          w_t = [t in X0] + [t in Yt]   (symmetric)
//     M0t = sum_{t in X0} z_t z_{t+lag}'
//     sx  = sum_{t in X0} z_t,  sy = sum_{t in Yt} z_t,  T = #pairs
// where X0 = frames with a partner lag frames later in the same segment and
// Yt = the partners.  These are the raw moments of deeptime's reversible
// Covariance estimator (what TICA.fit consumes; oracle/npport.py:lagged_moments),
// kept un-normalised so that shards all-reduce by plain summation.
//
// Mapping.  The contraction index is the frame axis: one MFMA consumes 4 frames
// (K = 4) and a 16x16 feature tile.  Lane l holds feature (l & 15) of frame
// (l >> 4) for every 16-feature tile, which is both the A and the B fragment
// layout, so z_t serves as A for every product and as B for M00; the lagged
// frame is the B operand of M0t.  A wave owns a contiguous run of frames and
// ALL tiles (F <= 64: 16 M0t + 10 upper M00 tiles = 208 accumulator VGPRs), so
// X is read from HBM exactly once (the lagged re-read hits L2).  Per-wave
// accumulators are summed across the 4 waves of a workgroup through LDS in a
// fixed order, written as one slab per workgroup, and a second kernel adds the
// slabs in block order: bitwise reproducible, no atomics.
//
// Roofline: 3*F^2 flop per frame (2F^2 for M0t + F^2 for the symmetric half of
// M00) against F*s bytes -> MFMA-bound for F >~ 27.
#include "common.h"

#include <cstring>

namespace {

typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kUnroll = 1;  // k-groups (of 4 frames) per pipelined iteration (segments padded to kGroup)

struct FrameTab {
    int n;    // segments longer than lag
    int lag;
    int64_t start[MSM_SEG_INLINE];
    int64_t stop[MSM_SEG_INLINE];
    // Dense frame index prefix.  Every segment is padded to a multiple of kGroup
    // frames so that a pipelined group never straddles two segments: the segment
    // of a group is wave-uniform and lives in SGPRs.  Padding frames are masked.
    int64_t prefix[MSM_SEG_INLINE + 1];
    int64_t total;  // padded frames in those segments
    int64_t pairs;  // T
    double wy_w;    // weight of the Yt frames in M00: 1 (reversible estimator) or 0 (one-sided: M00 = sum over X0)
};
constexpr int kGroup = 4 * kUnroll;

template <typename T>
__device__ __forceinline__ double to_f64(T v) { return (double)v; }

template <int NT>
struct CovShape {
    static constexpr int kSym = NT * (NT + 1) / 2;
    static constexpr int kTiles = NT * NT + kSym;
    static constexpr int kSlab = kTiles * 256 + 2 * NT * 16;  // doubles per workgroup slab
};

// upper-triangular tile index -> position
template <int NT>
__host__ __device__ constexpr int sym_index(int ti, int tj) { return ti * NT - ti * (ti - 1) / 2 + (tj - ti); }

// Tile ownership.  With F > 32 the 16 + 10 (or 9 + 6) accumulator tiles do not fit the
// 256 registers a wave may use at two waves per SIMD, and a lone wave per SIMD cannot
// overlap its fp64 VALU operand preparation with its own in-order MFMA stream.  So the
// tiles are split between two waves that walk the SAME frames (waves w and w+4 of a
// 512-thread workgroup share a SIMD): while one wave's MFMAs occupy the matrix core the
// other converts/centres its next operands.  Both halves load the operands (L1 hits).
template <int NT>
struct TileSplit {
    static constexpr int kSym = NT * (NT + 1) / 2;
    static constexpr int kTotal = NT * NT + kSym;
    static constexpr int kRows0 = (NT + 1) / 2;                   // M0t tile rows of half 0
    static constexpr int kSym0 = (kTotal + 1) / 2 - kRows0 * NT;  // first kSym0 M00 tiles -> half 0
    // HALF < 0: the wave owns everything
    template <int HALF> static constexpr bool own_0t(int a) { return HALF < 0 || (HALF == 0) == (a < kRows0); }
    template <int HALF> static constexpr bool own_00(int sidx) { return HALF < 0 || (HALF == 0) == (sidx < kSym0); }
};

// Symmetric flavour (reversible estimator only): the wave pair accumulates S = sum wx (zx + zy)(zx + zy)' and M00, both
// symmetric -- 2 x kSym tiles instead of NT^2 + kSym (20 instead of 26 at F = 64) -- and the finishing kernel forms
// (M0t + M0t') / 2 = (S - M00) / 2, which is all the reversible TICA ever reads of M0t.  S lives in the M0t tile slots
// (a, b), a <= b; half 0 owns S, half 1 owns M00 and both column sums.
template <int NT>
struct TileSplitSym {
    template <int HALF> static constexpr bool own_0t(int) { return HALF <= 0; }
    template <int HALF> static constexpr bool own_00(int) { return HALF != 0; }
};

// Feature <-> (tile, row) map.  The contraction does not care which 16 features form a
// "tile", so tile a holds features {NT*i + a : i = 0..15}: lane i then needs features
// NT*i .. NT*i + NT-1 of its frame, ONE contiguous load of NT elements (16 bytes for
// fp32 at F = 64), and 16 lanes cover a whole 64-feature row.
template <typename T, int NT, bool VEC, int HALF, bool FINITE, bool SYM = false>
__device__ __forceinline__ void cov_wave_body(const T* __restrict__ x, int F, int64_t ld, const FrameTab& ft,
                                              const double* __restrict__ mu, int64_t frames_per_wave,
                                              int frame_wave, int n_frame_waves, double* red, int red_wave,
                                              volatile int* prog = nullptr) {
    using S = CovShape<NT>;
    using TS = std::conditional_t<SYM, TileSplitSym<NT>, TileSplit<NT>>;
    const int lane = threadIdx.x & 63;
    const int fi_ = lane & 15;
    const int kk = lane >> 4;
    const int lag = ft.lag;

    v4f64 acc0t[NT][NT];
    v4f64 acc00[S::kSym];
    double ssum[NT], shift[NT];  // HALF 0 sums X0 (sx), HALF 1 sums Yt (sy); HALF < 0 keeps both
    double ssum2[NT];
    double sint[NT];             // SYM: column sums over interior groups (they count for sx and sy alike)
    bool fok[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
#pragma unroll
        for (int b = 0; b < NT; ++b) acc0t[a][b] = (v4f64){0.0, 0.0, 0.0, 0.0};
        ssum[a] = ssum2[a] = sint[a] = 0.0;
        const int f = NT * fi_ + a;
        fok[a] = f < F;
        shift[a] = fok[a] ? mu[f] : 0.0;
    }
#pragma unroll
    for (int a = 0; a < S::kSym; ++a) acc00[a] = (v4f64){0.0, 0.0, 0.0, 0.0};

    const int64_t gw = (int64_t)blockIdx.x * n_frame_waves + frame_wave;
    const int64_t q_begin = gw * frames_per_wave;  // multiple of kGroup, wave-uniform
    const int64_t q_end = min(q_begin + frames_per_wave, ft.total);

    using VT = T __attribute__((ext_vector_type(NT)));
    const double w_int = 1.0 + ft.wy_w;          // weight of an interior frame in M00: 2 (reversible) or 1 (one-sided)
    const double inv_w_int = 1.0 / w_int;        // 0.5 or 1: exact
    // Validity never masks the A operand: an out-of-segment lane re-reads the segment's
    // last frame (finite data) and its B operands carry weight 0, so it adds exact zeros.
    auto load_group = [&](int64_t t, int64_t s_start, int64_t s_stop, T (&rx)[NT], T (&ry)[NT], double& wx,
                          double& wy) {
        wx = (t + lag < s_stop) ? 1.0 : 0.0;
        wy = (t < s_stop && t - lag >= s_start) ? 1.0 : 0.0;
        const int64_t tx = min(t, s_stop - 1);
        const int64_t ty = min(t + lag, s_stop - 1);
        const T* px = x + tx * ld + NT * fi_;
        const T* py = x + ty * ld + NT * fi_;
        if constexpr (VEC) {
            const VT vx = *reinterpret_cast<const VT*>(px);
            const VT vy = *reinterpret_cast<const VT*>(py);
#pragma unroll
            for (int a = 0; a < NT; ++a) { rx[a] = vx[a]; ry[a] = vy[a]; }
        } else {
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                rx[a] = fok[a] ? px[a] : (T)0;
                ry[a] = fok[a] ? py[a] : (T)0;
            }
        }
    };

    // The two waves that share a frame run (HALF 0 / 1: the same frames, different tiles) read the same rows of X.
    // Left alone they drift apart by more than the CU's share of the L2 holds, and every row is then fetched from
    // memory twice (2.0x the algorithmic bytes in the round-1 FETCH_SIZE profile).  A rendezvous every 8 frame groups
    // through two LDS words keeps them within 16 groups (16 KB) of each other: the second reader hits L1 / L2.
    // (the words are addressed as LDS explicitly: through the generic pointer they became flat loads, and a flat
    // load waits for vmcnt(0) -- it drained the whole prefetch ring at every rendezvous)
    typedef __attribute__((address_space(3))) volatile int* lds_words;
    lds_words progl = (lds_words)prog;
    int my_groups = 0;
    auto rendezvous = [&]() {
#ifndef MSM_COV_NO_RDV
        if constexpr (HALF >= 0) {
            if ((my_groups & 7) == 0) {
                if (lane == 0) progl[2 * red_wave + HALF] = my_groups;
                while (progl[2 * red_wave + (1 - HALF)] + 8 < my_groups) __builtin_amdgcn_s_sleep(1);
            }
            ++my_groups;
        }
#endif
    };
    int seg = 0;  // wave-uniform
    int64_t q0 = q_begin;
    while (q0 < q_end) {
        while (seg + 1 < ft.n && q0 >= ft.prefix[seg + 1]) ++seg;
        const int64_t s_start = ft.start[seg], s_stop = ft.stop[seg];
        const int64_t q_hi = min(q_end, ft.prefix[seg + 1]);   // this wave's share of the segment
        // Frame groups in the interior of a segment (all four frames have a partner lag frames later AND are
        // partners themselves) need no weights and no clamped addresses.  fp64 vector instructions do not overlap
        // fp64 matrix instructions on this chip (SQ_VALU_MFMA_COEXEC_CYCLES = 0 for this kernel, profiles/r02_pmc.md),
        // so every vector instruction in the loop is matrix-pipe time lost: interior groups take 9 per feature pair
        // (convert and centre both rows, one add for the column sum) and none for addresses -- the rows of a group
        // come from one wave-uniform base plus a per-lane offset that never changes -- in a loop of their own, three
        // groups in flight.  M00 is accumulated at 1 / (1 + wy_w) of its weight (the weight of an interior frame, a
        // power of two) and scaled back when the wave hands in its tiles: exact, the bits are those of the weighted sum.
        const int64_t len = s_stop - s_start;
        const int64_t o_b = q0 - ft.prefix[seg], o_e = q_hi - ft.prefix[seg];   // wave-uniform group offsets, step 4
        // M0t tiles take sa (x) sb, M00 tiles za (x) zb; sa = za, sb = zy in the plain flavour, sa = zx + zy in the
        // symmetric one (upper tiles only)
        auto mfma_block = [&](const double (&za)[NT], const double (&zb)[NT], const double (&sa)[NT],
                              const double (&sb)[NT]) {
#pragma unroll
            for (int a = 0; a < NT; ++a) {
#pragma unroll
                for (int b = SYM ? a : 0; b < NT; ++b)
                    if (TS::template own_0t<HALF>(a))
                        acc0t[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[a], sb[b], acc0t[a][b], 0, 0, 0);
#pragma unroll
                for (int b = a; b < NT; ++b)
                    if (TS::template own_00<HALF>(sym_index<NT>(a, b)))
                        acc00[sym_index<NT>(a, b)] = __builtin_amdgcn_mfma_f64_16x16x4f64(
                            za[a], zb[b], acc00[sym_index<NT>(a, b)], 0, 0, 0);
            }
        };
        // groups [oa, ob) with weights and clamped rows: one group of look-ahead (segment heads and tails only)
        auto run_general = [&](int64_t oa, int64_t ob) {
            if (oa >= ob) return;
            T rx[NT], ry[NT];
            double wx, wy;
            int64_t tt = s_start + oa + kk;
            load_group(tt, s_start, s_stop, rx, ry, wx, wy);
            for (int64_t og = oa; og < ob; og += 4) {
                rendezvous();
                double za[NT], zb[NT], zy[NT], sv[NT];
                const double w = fma(ft.wy_w, wy, wx) * inv_w_int;
#pragma unroll
                for (int a = 0; a < NT; ++a) {
                    const double vx = to_f64(rx[a]);
                    const double vy = to_f64(ry[a]);
                    double cx = vx - shift[a];
                    double cy = vy - shift[a];
                    if constexpr (!FINITE) {  // NaN -> column mean
                        if (!(vx == vx)) cx = 0.0;
                        if (!(vy == vy)) cy = 0.0;
                    }
                    if constexpr (!VEC) {
                        if (!fok[a]) { cx = 0.0; cy = 0.0; }
                    }
                    za[a] = cx;
                    zb[a] = w * cx;
                    if constexpr (SYM) {
                        sv[a] = cx + cy;
                        zy[a] = wx * sv[a];
                        if constexpr (HALF != 0) {
                            ssum[a] = fma(wx, cx, ssum[a]);
                            ssum2[a] = fma(wy, cx, ssum2[a]);
                        }
                    } else {
                        sv[a] = cx;
                        zy[a] = wx * cy;
                        if constexpr (HALF <= 0) ssum[a] = fma(wx, cx, ssum[a]);
                        if constexpr (HALF == 1) ssum[a] = fma(wy, cx, ssum[a]);
                        if constexpr (HALF < 0) ssum2[a] = fma(wy, cx, ssum2[a]);
                    }
                }
                tt += 4;
                if (og + 4 < ob) load_group(tt, s_start, s_stop, rx, ry, wx, wy);
                mfma_block(za, zb, sv, zy);
            }
        };
        int64_t o_i0 = o_b, o_i1 = o_b;   // interior run [o_i0, o_i1) on the wave's own grid of groups
        if constexpr (VEC) {
            if (o_b < lag) o_i0 = o_b + ((lag - o_b + 3) / 4) * 4;
            const int64_t lim = len - lag - 3;   // first offset whose last frame has no partner
            o_i1 = lim > o_i0 ? o_i0 + ((lim - o_i0 + 3) / 4) * 4 : o_i0;
            o_i0 = min(o_i0, o_e);
            o_i1 = max(min(o_i1, o_e), o_i0);
        }
        run_general(o_b, o_i0);
        if constexpr (VEC) {
#ifdef MSM_COV_DEPTH
            constexpr int kDepth = MSM_COV_DEPTH;
#else
            constexpr int kDepth = 3;
#endif
            // A ring of kDepth groups in flight with a FIXED slot per position in the unrolled trip: no exit and no
            // conditional load inside the steady loop, so the slots are never copied around and each conversion waits
            // for its own (oldest) load only.  (With an exit test after every group the compiler rotated the slots
            // through copies behind s_waitcnt vmcnt(0): the ring was one trip deep whatever kDepth said.)  Rows are
            // addressed as wave-uniform base + a 32-bit lane offset that never changes: the scalar unit advances the
            // base, the vector unit does nothing for addresses.
            const int n_groups = (int)((o_i1 - o_i0) >> 2);
            VT rx[kDepth], ry[kDepth];
            const char* sbase = reinterpret_cast<const char*>(x + (s_start + o_i0) * ld);
            const uint32_t voff = (uint32_t)(((int64_t)kk * ld + NT * fi_) * (int64_t)sizeof(T));   // host: 4 ld sizeof(T) < 2^31
            const int64_t step_b = 4 * ld * (int64_t)sizeof(T), lag_b = (int64_t)lag * ld * (int64_t)sizeof(T);
            auto load_slot = [&](int gi, int s) {
                const char* p = sbase + gi * step_b;
                uint32_t off = voff;
                // opaque: keeps the zero-extension next to the load (scalar base + 32-bit lane offset addressing)
                asm volatile("" : "+v"(off));
                rx[s] = *reinterpret_cast<const VT*>(p + off);
                ry[s] = *reinterpret_cast<const VT*>(p + lag_b + off);
            };
            auto group = [&](int s, int next, bool more) {
                rendezvous();
                double za[NT], zy[NT];
#pragma unroll
                for (int a = 0; a < NT; ++a) {
#ifdef MSM_COV_DIAG_NOVALU   // timing experiment only: operands without conversion or centring (wrong results)
                    if constexpr (sizeof(T) == 4) {
                        za[a] = __hiloint2double(__float_as_int((float)rx[s][a]), __float_as_int((float)ry[s][(a + 1) % NT]));
                        zy[a] = __hiloint2double(__float_as_int((float)ry[s][a]), __float_as_int((float)rx[s][(a + 1) % NT]));
                        continue;
                    }
#endif
                    const double vx = to_f64(rx[s][a]);
                    const double vy = to_f64(ry[s][a]);
                    double cx = vx - shift[a];
                    double cy = vy - shift[a];
                    if constexpr (!FINITE) {  // NaN -> column mean
                        if (!(vx == vx)) cx = 0.0;
                        if (!(vy == vy)) cy = 0.0;
                    }
                    za[a] = cx;
                    if constexpr (SYM) {
                        zy[a] = cx + cy;
                        if constexpr (HALF != 0) sint[a] += cx;
                    } else {
                        zy[a] = cy;
                        ssum[a] += cx;
                        if constexpr (HALF < 0) ssum2[a] += cx;
                    }
                }
                // the slot is free only now: a load placed above its conversions lands in other registers and is copied
                // into the slot behind a full wait.  The empty asm takes every converted value as an input and the
                // load's lane offset passes through one after it, so the order holds.
#pragma unroll
                for (int a = 0; a < NT; ++a) {
                    if constexpr (!SYM || HALF != 0) asm volatile("" ::"v"(za[a]));
                    if constexpr (!SYM || HALF <= 0) asm volatile("" ::"v"(zy[a]));
                }
#ifndef MSM_COV_DIAG_NOLOAD   // timing experiment only: the ring is never refilled (wrong results)
                if (more) load_slot(next, s);
#endif
                if constexpr (SYM) mfma_block(za, za, zy, zy);
                else mfma_block(za, za, za, zy);
            };
            if (n_groups >= 2 * kDepth) {
                // (unconditional loads on the way in: with loads under branches ahead of the loop the compiler's
                // wait-count bookkeeping falls back to "wait for everything" at the loop head)
#pragma unroll
                for (int s = 0; s < kDepth; ++s) load_slot(s, s);
                int g = 0;
                for (; g + 2 * kDepth <= n_groups; g += kDepth) {
#pragma unroll
                    for (int s = 0; s < kDepth; ++s) group(s, g + kDepth + s, true);
                }
#pragma unroll
                for (int i = 0; i < 2 * kDepth - 1; ++i)   // fewer than 2 kDepth groups are left
                    if (g + i < n_groups) group(i % kDepth, g + i + kDepth, g + i + kDepth < n_groups);
            } else {
                run_general(o_i0, o_i1);   // a handful of groups: the weighted path gives the same bits (all weights 1)
            }
        }
        run_general(o_i1, o_e);
        q0 = q_hi;
    }

    if constexpr (HALF >= 0) {
        if (lane == 0) progl[2 * red_wave + HALF] = 0x7ffffff0;   // done: never hold the partner back
    }
    // ---- workgroup reduction: waves of a half add their tiles in wave order ----
    double* sums = red + S::kTiles * 256;  // [4][2][NT][64]
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        if constexpr (SYM) {
            if constexpr (HALF != 0) {
                sums[((red_wave * 2 + 0) * NT + a) * 64 + lane] = ssum[a] + sint[a];
                sums[((red_wave * 2 + 1) * NT + a) * 64 + lane] = ssum2[a] + sint[a];
            }
        } else {
            if constexpr (HALF <= 0) sums[((red_wave * 2 + 0) * NT + a) * 64 + lane] = ssum[a];
            if constexpr (HALF == 1) sums[((red_wave * 2 + 1) * NT + a) * 64 + lane] = ssum[a];
            if constexpr (HALF < 0) sums[((red_wave * 2 + 1) * NT + a) * 64 + lane] = ssum2[a];
        }
    }
    for (int w = 0; w < 4; ++w) {
        if (red_wave == w) {
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    if (TS::template own_0t<HALF>(a)) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int idx = ((a * NT + b) * 4 + r) * 64 + lane;
                            red[idx] = (w == 0 ? 0.0 : red[idx]) + acc0t[a][b][r];
                        }
                    }
#pragma unroll
            for (int s = 0; s < S::kSym; ++s)
                if (TS::template own_00<HALF>(s)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int idx = ((NT * NT + s) * 4 + r) * 64 + lane;
                        red[idx] = (w == 0 ? 0.0 : red[idx]) + w_int * acc00[s][r];
                    }
                }
        }
        __syncthreads();
    }
}

template <typename T, int NT, bool VEC, bool SPLIT, bool FINITE, bool SYM = false>
__global__ __launch_bounds__(SPLIT ? 512 : 256, SPLIT ? 2 : 1) void cov_fused_kernel(
    const T* __restrict__ x, int F, int64_t ld, FrameTab ft, const double* __restrict__ mu, int64_t frames_per_wave,
    double* __restrict__ slabs) {
    using S = CovShape<NT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* red = reinterpret_cast<double*>(smem_raw);  // [kTiles*256] then [4][2][NT][64]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (SPLIT) {
        __shared__ int prog[8];   // progress of the 4 wave pairs (frame groups done), see cov_wave_body
        if (tid < 8) prog[tid] = 0;
        __syncthreads();
        if (wave < 4) cov_wave_body<T, NT, VEC, 0, FINITE, SYM>(x, F, ld, ft, mu, frames_per_wave, wave, 4, red, wave, prog);
        else cov_wave_body<T, NT, VEC, 1, FINITE, SYM>(x, F, ld, ft, mu, frames_per_wave, wave - 4, 4, red, wave - 4, prog);
    } else {
        cov_wave_body<T, NT, VEC, -1, FINITE, SYM>(x, F, ld, ft, mu, frames_per_wave, wave, 4, red, wave);
    }
    double* slab = slabs + (size_t)blockIdx.x * S::kSlab;
    for (int i = tid; i < S::kTiles * 256; i += blockDim.x) slab[i] = red[i];
    // column sums: element e = which*NT*16 + a*16 + i ; add over the 4 frame-waves and the 4 frame groups
    const double* sums = red + S::kTiles * 256;
    if (tid < 2 * NT * 16) {
        const int which = tid / (NT * 16);
        const int a = (tid / 16) % NT;
        const int f = tid % 16;
        double acc = 0.0;
        for (int w = 0; w < 4; ++w)
            for (int g = 0; g < 4; ++g) acc += sums[((w * 2 + which) * NT + a) * 64 + g * 16 + f];
        slab[S::kTiles * 256 + tid] = acc;
    }
}

// Adds the slabs and scatters tiles into the moment block
//   out = [M00 F*F][M0t F*F][sx F][sy F][T]      (raw, centred by `mu`, unscaled)
// 1024 threads = 16 slab-groups x 64 elements; group g adds slabs g, g+16, ... and the 16
// partial sums are added in group order (fixed order -> bitwise reproducible).
// SYM: the M0t tile slots hold S (upper tiles): element e of M0t tile (a, b), a <= b, also adds up its M00 partner
// (same tile position in the symmetric list, same register and lane) and writes (S - M00) / 2 to both mirror places.
template <int NT, bool SYM = false>
__global__ __launch_bounds__(1024) void cov_reduce_kernel(const double* __restrict__ slabs, int n_slabs, int F,
                                                         double pairs, double* __restrict__ out) {
    using S = CovShape<NT>;
    __shared__ double red[16][64];
    __shared__ double red2[SYM ? 16 : 1][64];
    const int io = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + io;
    double part = 0.0, part2 = 0.0;
    int e2 = -1;   // SYM: the M00 element under an S element
    if constexpr (SYM) {
        if (e < NT * NT * 256) {
            const int tile = e >> 8, a = tile / NT, b = tile % NT;
            if (a <= b) e2 = (NT * NT + sym_index<NT>(a, b)) * 256 + (e & 255);
        }
    }
    if (e < S::kSlab)
        for (int b = g; b < n_slabs; b += 16) {
            part += slabs[(size_t)b * S::kSlab + e];
            if (SYM && e2 >= 0) part2 += slabs[(size_t)b * S::kSlab + e2];
        }
    red[g][io] = part;
    if constexpr (SYM) red2[g][io] = part2;
    __syncthreads();
    if (g != 0 || e >= S::kSlab) return;
    double acc = 0.0, acc2 = 0.0;
    for (int k = 0; k < 16; ++k) acc += red[k][io];
    if constexpr (SYM)
        for (int k = 0; k < 16; ++k) acc2 += red2[k][io];
    double* M00 = out;
    double* M0t = out + (size_t)F * F;
    double* sxy = M0t + (size_t)F * F;
    if (e < S::kTiles * 256) {
        const int tile = e >> 8;
        const int r = (e >> 6) & 3;
        const int lane = e & 63;
        const int row_in = (lane >> 4) + 4 * r;  // f64 MFMA C/D layout
        const int col_in = lane & 15;
        // feature of (tile a, row i) is NT*i + a (see cov_fused_kernel)
        if (tile < NT * NT) {
            const int row = NT * row_in + tile / NT, col = NT * col_in + tile % NT;
            if constexpr (SYM) {
                if (e2 >= 0 && row < F && col < F) {
                    const double h = 0.5 * (acc - acc2);
                    M0t[(size_t)row * F + col] = h;
                    if (tile / NT != tile % NT) M0t[(size_t)col * F + row] = h;
                }
            } else {
                if (row < F && col < F) M0t[(size_t)row * F + col] = acc;
            }
        } else {
            int s = tile - NT * NT, ti = 0;
            while (s >= NT - ti) { s -= NT - ti; ++ti; }
            const int tj = ti + s;
            const int row = NT * row_in + ti, col = NT * col_in + tj;
            if (row < F && col < F) {
                M00[(size_t)row * F + col] = acc;
                if (ti != tj) M00[(size_t)col * F + row] = acc;
            }
        }
    } else {
        const int i = e - S::kTiles * 256;
        const int which = i / (NT * 16);
        const int a = (i / 16) % NT, fi = i % 16;
        const int f = NT * fi + a;
        if (f < F) sxy[(size_t)which * F + f] = acc;
    }
    if (e == 0) sxy[2 * (size_t)F] = pairs;
}

// ---------------------------------------------------------------------------
// F > 64: the F x F outputs are cut into 64 x 64 feature blocks; a workgroup computes ONE
// block (16 tiles, 128 accumulator VGPRs per wave: two waves per SIMD without splitting) of
// either M0t (bi, bj) or the upper triangle of M00 (bi <= bj) over its frame chunk.
// grid = (frame chunks, block tasks).  Every task streams two 64-feature column blocks of X,
// so X is re-read from L2 / Infinity Cache by the tasks that share a frame chunk.
// ---------------------------------------------------------------------------
constexpr int kBlkSlab = 16 * 256 + 2 * 64;  // doubles: 16 tiles + sx/sy of the 64 features

template <typename T, bool VEC, bool FINITE>
__global__ __launch_bounds__(256, 2) void cov_block_kernel(const T* __restrict__ x, int F, int64_t ld, FrameTab ft,
                                                           const double* __restrict__ mu, int64_t frames_per_wave,
                                                           int n_fb, double* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* red = reinterpret_cast<double*>(smem_raw);  // [16*256] + [4][2][4][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi_ = lane & 15, kk = lane >> 4;
    const int lag = ft.lag;
    // decode the block task
    const int task = blockIdx.y;
    int bi, bj;
    bool is_00;
    if (task < n_fb * n_fb) { is_00 = false; bi = task / n_fb; bj = task - bi * n_fb; }
    else {
        is_00 = true;
        int s = task - n_fb * n_fb;
        bi = 0;
        while (s >= n_fb - bi) { s -= n_fb - bi; ++bi; }
        bj = bi + s;
    }
    const bool diag = is_00 && bi == bj;
    v4f64 acc[4][4];
    double sx[4], sy[4], sha[4], shb[4];
    bool foka[4], fokb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (v4f64){0.0, 0.0, 0.0, 0.0};
        sx[a] = sy[a] = 0.0;
        const int fa = 64 * bi + 4 * fi_ + a, fb = 64 * bj + 4 * fi_ + a;
        foka[a] = fa < F; fokb[a] = fb < F;
        sha[a] = foka[a] ? mu[fa] : 0.0;
        shb[a] = fokb[a] ? mu[fb] : 0.0;
    }
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
    const int64_t q_begin = gw * frames_per_wave;
    const int64_t q_end = min(q_begin + frames_per_wave, ft.total);
    using VT = T __attribute__((ext_vector_type(4)));
    auto load_group = [&](int64_t t, int64_t s_start, int64_t s_stop, T (&ra)[4], T (&rb)[4], double& wx, double& wy) {
        wx = (t + lag < s_stop) ? 1.0 : 0.0;
        wy = (t < s_stop && t - lag >= s_start) ? 1.0 : 0.0;
        const int64_t ta = min(t, s_stop - 1);
        const int64_t tb = is_00 ? ta : min(t + lag, s_stop - 1);
        const T* pa = x + ta * ld + 64 * bi + 4 * fi_;
        const T* pb = x + tb * ld + 64 * bj + 4 * fi_;
        if constexpr (VEC) {
            const VT va = *reinterpret_cast<const VT*>(pa);
            const VT vb = *reinterpret_cast<const VT*>(pb);
#pragma unroll
            for (int a = 0; a < 4; ++a) { ra[a] = va[a]; rb[a] = vb[a]; }
        } else {
#pragma unroll
            for (int a = 0; a < 4; ++a) { ra[a] = foka[a] ? pa[a] : (T)0; rb[a] = fokb[a] ? pb[a] : (T)0; }
        }
    };
    int seg = 0;
    int64_t q0 = q_begin;
    while (q0 < q_end) {
        while (seg + 1 < ft.n && q0 >= ft.prefix[seg + 1]) ++seg;
        const int64_t s_start = ft.start[seg], s_stop = ft.stop[seg];
        const int64_t q_hi = min(q_end, ft.prefix[seg + 1]);
        int64_t t = s_start + (q0 - ft.prefix[seg]) + kk;
        T ra[4], rb[4];
        double wx, wy;
        load_group(t, s_start, s_stop, ra, rb, wx, wy);
        for (; q0 < q_hi; q0 += 4) {
            double za[4], zv[4];
            const double wb = is_00 ? fma(ft.wy_w, wy, wx) : wx;  // B-side weight: [X0]+[Yt] for M00, [X0] for M0t
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const double va = to_f64(ra[a]), vb = to_f64(rb[a]);
                double ca = va - sha[a], cb = vb - shb[a];
                if constexpr (!FINITE) {
                    if (!(va == va)) ca = 0.0;
                    if (!(vb == vb)) cb = 0.0;
                }
                if constexpr (!VEC) {
                    if (!foka[a]) ca = 0.0;
                    if (!fokb[a]) cb = 0.0;
                }
                za[a] = ca;
                zv[a] = wb * cb;
                if (diag) { sx[a] = fma(wx, ca, sx[a]); sy[a] = fma(wy, ca, sy[a]); }
            }
            t += 4;
            if (q0 + 4 < q_hi) load_group(t, s_start, s_stop, ra, rb, wx, wy);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(za[a], zv[b], acc[a][b], 0, 0, 0);
        }
    }
    double* sums = red + 16 * 256;  // [4][2][4][64]
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        sums[((wave * 2 + 0) * 4 + a) * 64 + lane] = sx[a];
        sums[((wave * 2 + 1) * 4 + a) * 64 + lane] = sy[a];
    }
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int idx = ((a * 4 + b) * 4 + r) * 64 + lane;
                        red[idx] = (w == 0 ? 0.0 : red[idx]) + acc[a][b][r];
                    }
        }
        __syncthreads();
    }
    double* slab = slabs + ((size_t)task * gridDim.x + blockIdx.x) * kBlkSlab;
    for (int i = tid; i < 16 * 256; i += 256) slab[i] = red[i];
    if (tid < 128) {
        const int which = tid >> 6, a = (tid >> 4) & 3, f = tid & 15;
        double accs = 0.0;
        for (int w = 0; w < 4; ++w)
            for (int g = 0; g < 4; ++g) accs += sums[((w * 2 + which) * 4 + a) * 64 + g * 16 + f];
        slab[16 * 256 + tid] = accs;
    }
}

__global__ __launch_bounds__(1024) void cov_block_reduce_kernel(const double* __restrict__ slabs, int n_chunks, int F,
                                                               int n_fb, double pairs, double* __restrict__ out) {
    __shared__ double red[16][64];
    const int io = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int task = blockIdx.y;
    const int e = blockIdx.x * 64 + io;
    double part = 0.0;
    if (e < kBlkSlab)
        for (int b = g; b < n_chunks; b += 16) part += slabs[((size_t)task * n_chunks + b) * kBlkSlab + e];
    red[g][io] = part;
    __syncthreads();
    if (g != 0 || e >= kBlkSlab) return;
    double acc = 0.0;
    for (int q = 0; q < 16; ++q) acc += red[q][io];
    int bi, bj;
    bool is_00;
    if (task < n_fb * n_fb) { is_00 = false; bi = task / n_fb; bj = task - bi * n_fb; }
    else {
        is_00 = true;
        int s = task - n_fb * n_fb;
        bi = 0;
        while (s >= n_fb - bi) { s -= n_fb - bi; ++bi; }
        bj = bi + s;
    }
    double* M00 = out;
    double* M0t = out + (size_t)F * F;
    double* sxy = M0t + (size_t)F * F;
    if (e < 16 * 256) {
        const int tile = e >> 8, r = (e >> 6) & 3, lane = e & 63;
        const int a = tile >> 2, b = tile & 3;
        const int row = 64 * bi + 4 * ((lane >> 4) + 4 * r) + a;   // f64 MFMA C/D layout + feature map
        const int col = 64 * bj + 4 * (lane & 15) + b;
        if (row < F && col < F) {
            if (!is_00) M0t[(size_t)row * F + col] = acc;
            else {
                M00[(size_t)row * F + col] = acc;
                if (bi != bj) M00[(size_t)col * F + row] = acc;
            }
        }
    } else if (is_00 && bi == bj) {
        const int i = e - 16 * 256;
        const int which = i >> 6, a = (i >> 4) & 3, fi = i & 15;
        const int f = 64 * bi + 4 * fi + a;
        if (f < F) sxy[(size_t)which * F + f] = acc;
    }
    if (task == 0 && e == 0) sxy[2 * (size_t)F] = pairs;
}

template <typename T>
msm_status launch_cov_blocked(msm_ctx* ctx, const T* x, int F, int64_t ld, const FrameTab& ft, const double* mu,
                              bool finite, double* d_out) {
    const int n_fb = (F + 63) / 64;
    const int n_tasks = n_fb * n_fb + n_fb * (n_fb + 1) / 2;
    int chunks = std::max(1, ctx->n_cu * 2 / n_tasks);
    const int64_t groups = ft.total / kGroup;
    if ((int64_t)chunks * 4 * 4 > groups) chunks = (int)std::max<int64_t>(1, groups / 16);
    int64_t fpw = (ft.total + (int64_t)chunks * 4 - 1) / ((int64_t)chunks * 4);
    fpw = (fpw + kGroup - 1) / kGroup * kGroup;
    chunks = (int)((ft.total + fpw * 4 - 1) / (fpw * 4));
    msm_status rs = msm_reserve_scratch(ctx, (size_t)n_tasks * chunks * kBlkSlab * sizeof(double));
    if (rs != MSM_OK) return rs;
    const size_t lds = ((size_t)16 * 256 + 4 * 2 * 4 * 64) * sizeof(double);
    const bool vec = (F % 64 == 0) && (ld % 4 == 0) && (((uintptr_t)x) % (4 * sizeof(T)) == 0);
    auto kern = vec ? (finite ? cov_block_kernel<T, true, true> : cov_block_kernel<T, true, false>)
                    : (finite ? cov_block_kernel<T, false, true> : cov_block_kernel<T, false, false>);
    MSM_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(chunks, n_tasks), dim3(256), lds, ctx->stream, x, F, ld, ft, mu, fpw, n_fb,
                       (double*)ctx->scratch);
    MSM_CHECK_LAUNCH(ctx);
    hipLaunchKernelGGL(cov_block_reduce_kernel, dim3(msm_ceil_div(kBlkSlab, 64), n_tasks), dim3(1024), 0, ctx->stream,
                       (const double*)ctx->scratch, chunks, F, n_fb, (double)ft.pairs, d_out);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status build_frametab(msm_ctx* ctx, int64_t n, const int64_t* h_start, const int64_t* h_stop, int n_seg, int lag,
                          bool one_sided, FrameTab* out) {
    MSM_REQUIRE(ctx, lag >= 0, "lag must be >= 0 (got %d)", lag);   // 0: instantaneous covariance (PCA)
    FrameTab ft;
    memset(&ft, 0, sizeof(ft));
    ft.lag = lag;
    ft.wy_w = one_sided ? 0.0 : 1.0;
    const int64_t one_a = 0, one_b = n;
    if (n_seg == 0) { h_start = &one_a; h_stop = &one_b; n_seg = 1; }
    MSM_REQUIRE(ctx, h_start && h_stop, "segment arrays are NULL");
    for (int s = 0; s < n_seg; ++s) {
        const int64_t a = h_start[s] < 0 ? 0 : h_start[s];
        const int64_t b = h_stop[s] > n ? n : h_stop[s];
        if (b - a <= lag) continue;
        if (ft.n == MSM_SEG_INLINE)
            return msm_fail(ctx, MSM_ERR_UNSUPPORTED, "msm_lagged_moments: more than %d segments per call; "
                            "accumulate over several calls", MSM_SEG_INLINE);
        ft.start[ft.n] = a;
        ft.stop[ft.n] = b;
        ft.prefix[ft.n + 1] = ft.prefix[ft.n] + ((b - a) + kGroup - 1) / kGroup * kGroup;
        ft.pairs += (b - a) - lag;
        ++ft.n;
    }
    ft.total = ft.prefix[ft.n];
    *out = ft;
    return MSM_OK;
}

// M0t <- (M0t + M0t') / 2 in place (the symmetric flavour on the shapes without a kernel of their own)
__global__ __launch_bounds__(256) void symmetrise_m0t_kernel(double* __restrict__ m0t, int F) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)F * F) return;
    const int i = (int)(e / F), j = (int)(e % F);
    if (i >= j) return;
    const double h = 0.5 * (m0t[(size_t)i * F + j] + m0t[(size_t)j * F + i]);
    m0t[(size_t)i * F + j] = h;
    m0t[(size_t)j * F + i] = h;
}

template <typename T, int NT>
msm_status launch_cov(msm_ctx* ctx, const T* x, int F, int64_t ld, const FrameTab& ft, const double* mu,
                      bool finite, bool sym, double* d_out) {
    using S = CovShape<NT>;
    int blocks = ctx->n_cu;  // one 4-wave workgroup per CU: each wave owns a SIMD's matrix core
    const int64_t groups = ft.total / kGroup;
    const int64_t min_groups_per_wave = 4;
    if ((int64_t)blocks * kWaves * min_groups_per_wave > groups)
        blocks = (int)std::max<int64_t>(1, groups / (kWaves * min_groups_per_wave));
    int64_t fpw = (ft.total + (int64_t)blocks * kWaves - 1) / ((int64_t)blocks * kWaves);
    fpw = (fpw + kGroup - 1) / kGroup * kGroup;
    blocks = (int)((ft.total + fpw * kWaves - 1) / (fpw * kWaves));
    msm_status rs = msm_reserve_scratch(ctx, (size_t)blocks * S::kSlab * sizeof(double));
    if (rs != MSM_OK) return rs;
    const size_t lds = ((size_t)S::kTiles * 256 + (size_t)kWaves * 2 * NT * 64) * sizeof(double);
    // vector path: every lane's NT features exist and its NT-element load is aligned
    const bool vec = (NT != 3) && (F == 16 * NT) && (ld % NT == 0) && (((uintptr_t)x) % (NT * sizeof(T)) == 0) &&
                     (ld * 4 * (int64_t)sizeof(T) < (int64_t(1) << 31));   // the kernel's 32-bit lane offsets
    constexpr bool kSplit = NT >= 3;
    constexpr bool kHasSym = kSplit;   // the symmetric flavour pays where the tiles are split over a wave pair
    auto kern = vec ? (finite ? cov_fused_kernel<T, NT, true, kSplit, true> : cov_fused_kernel<T, NT, true, kSplit, false>)
                    : (finite ? cov_fused_kernel<T, NT, false, kSplit, true> : cov_fused_kernel<T, NT, false, kSplit, false>);
    if constexpr (kHasSym) {
        if (sym)
            kern = vec ? (finite ? cov_fused_kernel<T, NT, true, kSplit, true, true> : cov_fused_kernel<T, NT, true, kSplit, false, true>)
                       : (finite ? cov_fused_kernel<T, NT, false, kSplit, true, true> : cov_fused_kernel<T, NT, false, kSplit, false, true>);
    }
    if (lds > 48 * 1024)
        MSM_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(kSplit ? 512 : 256), lds, ctx->stream, x, F, ld, ft, mu, fpw,
                       (double*)ctx->scratch);
    MSM_CHECK_LAUNCH(ctx);
    if (sym && kHasSym) {
        hipLaunchKernelGGL((cov_reduce_kernel<NT, kHasSym>), dim3(msm_ceil_div(S::kSlab, 64)), dim3(1024), 0, ctx->stream,
                           (const double*)ctx->scratch, blocks, F, (double)ft.pairs, d_out);
    } else {
        hipLaunchKernelGGL(cov_reduce_kernel<NT>, dim3(msm_ceil_div(S::kSlab, 64)), dim3(1024), 0, ctx->stream,
                           (const double*)ctx->scratch, blocks, F, (double)ft.pairs, d_out);
        if (sym) {
            MSM_CHECK_LAUNCH(ctx);
            hipLaunchKernelGGL(symmetrise_m0t_kernel, dim3(msm_ceil_div(F * F, 256)), dim3(256), 0, ctx->stream,
                               d_out + (size_t)F * F, F);
        }
    }
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

template <typename T>
msm_status dispatch_cov(msm_ctx* ctx, const T* x, int F, int64_t ld, const FrameTab& ft, const double* mu,
                        bool finite, bool sym, double* d_out) {
    if (F <= 16) return launch_cov<T, 1>(ctx, x, F, ld, ft, mu, finite, sym, d_out);
    if (F <= 32) return launch_cov<T, 2>(ctx, x, F, ld, ft, mu, finite, sym, d_out);
    if (F <= 48) return launch_cov<T, 3>(ctx, x, F, ld, ft, mu, finite, sym, d_out);
    if (F <= 64) return launch_cov<T, 4>(ctx, x, F, ld, ft, mu, finite, sym, d_out);
    msm_status rs = launch_cov_blocked<T>(ctx, x, F, ld, ft, mu, finite, d_out);
    if (rs == MSM_OK && sym) {
        hipLaunchKernelGGL(symmetrise_m0t_kernel, dim3(msm_ceil_div((int64_t)F * F, 256)), dim3(256), 0, ctx->stream,
                           d_out + (size_t)F * F, F);
        MSM_CHECK_LAUNCH(ctx);
    }
    return rs;
}

}  // namespace

namespace {

struct EdgeSegs {
    int n;
    int lag;
    int64_t start[MSM_SEG_INLINE];
    int64_t stop[MSM_SEG_INLINE];
};

// Column sums over ALL frames of the segments from the lagged moments of the same shift:
// X0 = [s, e - lag) and Yt = [s + lag, e) cover every frame twice except the first / last `lag`
// frames of a segment, which this kernel adds (segments no longer than the lag lie in both edges
// entirely): 2 S1 = sx + sy + first + last, 2 S2 = diag(M00) + first2 + last2.
// One workgroup; thread f owns feature f (F <= 1024 per pass), at most 2 lag frames per segment.
template <typename T>
__global__ __launch_bounds__(256) void moments_from_lagged_kernel(const T* __restrict__ x, int F, int64_t ld,
                                                                  EdgeSegs sg, const double* __restrict__ shift,
                                                                  const double* __restrict__ mom,
                                                                  double* __restrict__ sums) {
    double frames = 0.0;
    for (int q = 0; q < sg.n; ++q) frames += (double)(sg.stop[q] - sg.start[q]);
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        const double sh = shift[f];
        double e1 = 0.0, e2 = 0.0;
        for (int q = 0; q < sg.n; ++q) {
            const int64_t s = sg.start[q], e = sg.stop[q];
            const int64_t head = min<int64_t>(s + sg.lag, e), tail = max<int64_t>(e - sg.lag, s);
            const int n_head = (int)(head - s), n_edge = n_head + (int)(e - tail);
#pragma unroll 8
            for (int q2 = 0; q2 < n_edge; ++q2) {   // independent loads: several rows in flight per thread
                const int64_t t = q2 < n_head ? s + q2 : tail + (q2 - n_head);
                const double z = (double)x[t * ld + f] - sh;
                e1 += z;
                e2 = fma(z, z, e2);
            }
        }
        const double sx = mom[(size_t)2 * F * F + f], sy = mom[(size_t)2 * F * F + F + f];
        sums[f] = frames;
        sums[F + f] = 0.5 * (sx + sy + e1);
        sums[2 * F + f] = 0.5 * (mom[(size_t)f * F + f] + e2);
    }
}

}  // namespace

extern "C" {

static msm_status lagged_moments_impl(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int F, int64_t ld,
                                      const int64_t* h_seg_start, const int64_t* h_seg_stop, int n_seg, int lag,
                                      const double* d_shift, int assume_finite, bool one_sided, bool sym,
                                      double* d_moments) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 0 && F >= 1 && ld >= F, "msm_lagged_moments: need n >= 0, F >= 1, ld >= F");
    MSM_REQUIRE(ctx, dtype == MSM_F32 || dtype == MSM_F64, "msm_lagged_moments: bad dtype");
    MSM_REQUIRE(ctx, d_shift && d_moments && (d_x || n == 0), "msm_lagged_moments: NULL pointer");
    FrameTab ft;
    msm_status rs = build_frametab(ctx, n, h_seg_start, h_seg_stop, n_seg, lag, one_sided, &ft);
    if (rs != MSM_OK) return rs;
    if (ft.total == 0) {
        MSM_HIP(ctx, hipMemsetAsync(d_moments, 0, ((size_t)2 * F * F + 2 * F + 1) * sizeof(double), ctx->stream));
        return MSM_OK;
    }
    if (dtype == MSM_F32)
        return dispatch_cov<float>(ctx, (const float*)d_x, F, ld, ft, d_shift, assume_finite != 0, sym, d_moments);
    return dispatch_cov<double>(ctx, (const double*)d_x, F, ld, ft, d_shift, assume_finite != 0, sym, d_moments);
}

msm_status msm_lagged_moments(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int F, int64_t ld,
                              const int64_t* h_seg_start, const int64_t* h_seg_stop, int n_seg, int lag,
                              const double* d_shift, int assume_finite, double* d_moments) {
    return lagged_moments_impl(ctx, d_x, dtype, n, F, ld, h_seg_start, h_seg_stop, n_seg, lag, d_shift, assume_finite,
                               false, false, d_moments);
}

msm_status msm_lagged_moments_reversible(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int F, int64_t ld,
                                         const int64_t* h_seg_start, const int64_t* h_seg_stop, int n_seg, int lag,
                                         const double* d_shift, int assume_finite, double* d_moments) {
    return lagged_moments_impl(ctx, d_x, dtype, n, F, ld, h_seg_start, h_seg_stop, n_seg, lag, d_shift, assume_finite,
                               false, true, d_moments);
}

msm_status msm_lagged_moments_onesided(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int F, int64_t ld,
                                       const int64_t* h_seg_start, const int64_t* h_seg_stop, int n_seg, int lag,
                                       const double* d_shift, int assume_finite, double* d_moments) {
    return lagged_moments_impl(ctx, d_x, dtype, n, F, ld, h_seg_start, h_seg_stop, n_seg, lag, d_shift, assume_finite,
                               true, false, d_moments);
}

msm_status msm_moments_from_lagged(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int F, int64_t ld,
                                   const int64_t* h_seg_start, const int64_t* h_seg_stop, int n_seg, int lag,
                                   const double* d_shift, const double* d_moments, double* d_sums) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 0 && F >= 1 && ld >= F && lag >= 1, "msm_moments_from_lagged: bad shape / lag");
    MSM_REQUIRE(ctx, dtype == MSM_F32 || dtype == MSM_F64, "msm_moments_from_lagged: bad dtype");
    MSM_REQUIRE(ctx, d_shift && d_moments && d_sums && (d_x || n == 0), "msm_moments_from_lagged: NULL pointer");
    MSM_REQUIRE(ctx, n_seg >= 0 && n_seg <= MSM_SEG_INLINE && (n_seg == 0 || (h_seg_start && h_seg_stop)),
                "msm_moments_from_lagged: at most %d segments per call", MSM_SEG_INLINE);
    EdgeSegs sg;
    sg.n = 0;
    sg.lag = lag;
    if (n_seg == 0) {
        if (n > 0) { sg.start[0] = 0; sg.stop[0] = n; sg.n = 1; }
    } else {
        for (int q = 0; q < n_seg; ++q) {
            const int64_t a = std::max<int64_t>(0, h_seg_start[q]), b = std::min<int64_t>(n, h_seg_stop[q]);
            if (b > a) { sg.start[sg.n] = a; sg.stop[sg.n] = b; ++sg.n; }
        }
    }
    if (dtype == MSM_F32)
        hipLaunchKernelGGL(moments_from_lagged_kernel<float>, dim3(1), dim3(256), 0, ctx->stream, (const float*)d_x, F, ld,
                           sg, d_shift, d_moments, d_sums);
    else
        hipLaunchKernelGGL(moments_from_lagged_kernel<double>, dim3(1), dim3(256), 0, ctx->stream, (const double*)d_x, F,
                           ld, sg, d_shift, d_moments, d_sums);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

}  // extern "C"
