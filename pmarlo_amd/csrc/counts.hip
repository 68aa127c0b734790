// Lag-tau transition counts: LDS-binned atomics, row-block privatised.
//
// Layout.  labels int32 [n] in HBM (4 B/frame is the whole algorithmic read;
// the tau-shifted read hits L2).  The k x k matrix is split into row blocks of
// `rows` source states so that rows*k 4-byte bins fit the LDS budget; the grid
// is (frame chunks) x (row blocks) [x lags].  A workgroup streams its chunk of
// pair ids with coalesced loads, bins the pairs whose source state falls into
// its row block with LDS atomics, and flushes only the non-zero bins into the
// int64 matrix with global atomics.  Integer adds commute, so the result is
// bit-exact and independent of scheduling.  When even 16 rows do not fit (very
// large k) the kernel degenerates to direct global atomics.  Unweighted counts with
// at least as many pairs as bins (the bench shape) take the two-pass bucket path
// further down instead: no global atomics at all.
#include "common.h"

#include <cstdlib>

namespace {

constexpr int kThreads = 1024;  // 16 waves per CU hide the label-load latency (one 128 KB workgroup per CU)
constexpr int kLdsBudgetSmall = 64 * 1024;   // 2 workgroups / CU
constexpr int kLdsBudgetLarge = 128 * 1024;  // 1 workgroup / CU
constexpr int kMaxRowBlocks = 16;

template <bool WEIGHTED>
struct BinT { using lds_t = unsigned int; using out_t = unsigned long long; };
template <>
struct BinT<true> { using lds_t = double; using out_t = double; };

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// One same-address global atomic per WORKGROUP: thousands of per-wave atomics on one counter
// serialise in L2 and used to cost more than the whole binning pass.
__device__ __forceinline__ void block_add_u64(unsigned long long v, unsigned long long* dst) {
    __shared__ unsigned long long wave_part[kThreads / 64];
    v = wave_sum_u64(v);
    if ((threadIdx.x & 63) == 0) wave_part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < kThreads / 64; ++w) t += wave_part[w];
        if (t) atomicAdd(dst, t);
    }
}

// One lag.  grid = (chunks, row_blocks).
template <bool WEIGHTED>
__global__ __launch_bounds__(kThreads) void count_lds_kernel(
    const int32_t* __restrict__ labels, const double* __restrict__ weights, SegTab st, int k,
    int rows, int64_t pairs_per_chunk, typename BinT<WEIGHTED>::out_t* __restrict__ counts,
    unsigned long long* __restrict__ pairs_out) {
    using lds_t = typename BinT<WEIGHTED>::lds_t;
    using out_t = typename BinT<WEIGHTED>::out_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lds_t* bins = reinterpret_cast<lds_t*>(smem_raw);

    const int tid = threadIdx.x;
    const int r0 = blockIdx.y * rows;
    const int nrows = min(rows, k - r0);
    const int nbins = nrows * k;
    for (int i = tid; i < nbins; i += kThreads) bins[i] = (lds_t)0;
    __syncthreads();

    const int64_t p0 = (int64_t)blockIdx.x * pairs_per_chunk;
    const int64_t p1 = min(p0 + pairs_per_chunk, st.total_pairs);
    const int lag = st.lag;
    unsigned long long local_pairs = 0;
    auto bin_pair = [&](int a, int b, int64_t t) {
        const int ra = a - r0;
        if ((unsigned)b < (unsigned)k && (unsigned)ra < (unsigned)nrows) {
            if constexpr (WEIGHTED) atomicAdd(&bins[ra * k + b], weights[t]);
            else atomicAdd(&bins[ra * k + b], 1u);
            ++local_pairs;
        }
    };
#ifdef MSM_COUNTS_UNROLL
    constexpr int kU = MSM_COUNTS_UNROLL;
#else
    constexpr int kU = 4;
#endif
    int64_t p = p0 + tid;
    for (; p + (kU - 1) * kThreads < p1; p += kU * kThreads) {  // 2 kU independent loads in flight per lane
        int64_t t[kU];
        int a[kU], b[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) t[u] = seg_pair_to_frame(st, p + u * kThreads);
#pragma unroll
        for (int u = 0; u < kU; ++u) { a[u] = labels[t[u]]; b[u] = labels[t[u] + lag]; }
#pragma unroll
        for (int u = 0; u < kU; ++u) bin_pair(a[u], b[u], t[u]);
    }
    for (; p < p1; p += kThreads) {
        const int64_t t = seg_pair_to_frame(st, p);
        bin_pair(labels[t], labels[t + lag], t);
    }
    __syncthreads();
    out_t* dst = counts + (size_t)r0 * k;
#ifndef MSM_COUNTS_DIAG_NOFLUSH   // timing experiment only
    for (int i = tid; i < nbins; i += kThreads) {
        const lds_t v = bins[i];
        if (v != (lds_t)0) atomicAdd(&dst[i], (out_t)v);
    }
#endif
    if (pairs_out) block_add_u64(local_pairs, pairs_out);
}


// ---- bucket path (unweighted, dense regime: at least as many pairs as bins) --------------------------------------------
// The LDS-privatised kernel above still ends in one 64-bit global atomic per non-empty (workgroup, bin), and at C3
// (1 M pairs, 250 k bins) that is most of its time and most of its HBM traffic: a device-scope atomic is resolved behind
// the L2s (64 B of traffic each, ~16 G atomics/s over the whole chip, measured), and so are atomics into per-XCD private
// copies of the matrix (tried: 60 us random labels, 1.4 ms when every pair hits one bin).  This path has NO global
// atomics on the matrix.
//   Pass 1: a workgroup counts the DISTINCT pairs of its chunk in an LDS hash table (a trajectory dwells: a chunk of
//   consecutive frames holds few distinct pairs), groups them by bucket of source rows and writes them out as 16-bit
//   elements with its table of group offsets.  Element stream of a group: a key alone (the pair was seen once) or a key
//   with the `multi` flag followed by a count element.
//   Pass 2: workgroup b OWNS the rows of bucket b, gathers its groups from all chunks, bins them in LDS and stores its
//   rows of the int64 matrix once.
// Integer adds only: exact and independent of scheduling.  HBM traffic at C3 with no repeated pairs at all: labels 4 MB
// + elements 2 MB out and in + offsets 0.5 MB + matrix 2 MB = 10.5 MB (the privatised kernel: 26 MB).
constexpr int kBucketsMax = 256;            // pass-2 workgroups (one per CU)
constexpr int kChunkMax = 4096;             // pairs per pass-1 workgroup (hash table of 2 x chunk slots, 8 B each)
constexpr int kBinCopiesBytes = 64 * 1024;  // LDS for the (wave-private) copies of a bucket's bins; R k <= 16384 keys
constexpr unsigned kElemMulti = 0x8000u;    // key element: a count element follows
constexpr unsigned kElemCount = 0x4000u;    // count element: count - 2 in the low bits (count <= kChunkMax)
constexpr unsigned kSlotEmpty = 0xFFFFFFFFu;

// exclusive prefix sum over the workgroup (kThreads values); total returned to every thread
__device__ __forceinline__ unsigned block_excl_scan_u32(unsigned v, unsigned* total) {
    __shared__ unsigned wave_tot[kThreads / 64 + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    __syncthreads();                       // wave_tot of a previous call is no longer read
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned run = 0;
        for (int w = 0; w < kThreads / 64; ++w) { const unsigned t = wave_tot[w]; wave_tot[w] = run; run += t; }
        wave_tot[kThreads / 64] = run;
    }
    __syncthreads();
    *total = wave_tot[kThreads / 64];
    return wave_tot[wave] + inc - v;
}

// Pass 1.  grid = chunks.  elems: [chunks][chunk] uint16, offs: [chunks][buckets + 1] uint32 (a coalesced row per chunk;
// the transposed layout cost a 32-byte write transaction per 4-byte offset).  slots: power of two >= 2 chunk.
__global__ __launch_bounds__(kThreads) void count_bucket_scatter_kernel(const int32_t* __restrict__ labels, SegTab st, int k,
                                                                       int R, int buckets, int chunk, int slots,
                                                                       unsigned short* __restrict__ elems,
                                                                       unsigned int* __restrict__ offs,
                                                                       unsigned int* __restrict__ valid_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned int* tkey = reinterpret_cast<unsigned int*>(smem_raw);                  // [slots] pair id a k + b
    unsigned int* tcnt = tkey + slots;                                               // [slots] times seen
    unsigned int* hist = tcnt + slots;                                               // [kBucketsMax] then the write cursors
    unsigned int* start = hist + kBucketsMax;                                        // [kBucketsMax + 1]
    unsigned short* out = reinterpret_cast<unsigned short*>(start + kBucketsMax + 1 + 1);   // [chunk] elements by bucket
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < slots; i += kThreads) { tkey[i] = kSlotEmpty; tcnt[i] = 0; }
    if (tid < kBucketsMax) hist[tid] = 0;
    __syncthreads();
    const int64_t p0 = (int64_t)blockIdx.x * chunk;
    const int cnt = (int)max((int64_t)0, min((int64_t)chunk, st.total_pairs - p0));
    const int lag = st.lag;
    const unsigned mask = (unsigned)slots - 1u;
    unsigned local_pairs = 0;
    for (int i0 = 0; i0 < cnt; i0 += kThreads) {       // whole waves iterate together (wave-wide run detection below)
        const int i = i0 + tid;
        unsigned id = kSlotEmpty;
        if (i < cnt) {
            const int64_t t = seg_pair_to_frame(st, p0 + i);
            const int a = labels[t], b = labels[t + lag];
            if ((unsigned)a < (unsigned)k && (unsigned)b < (unsigned)k) { id = (unsigned)a * (unsigned)k + (unsigned)b; ++local_pairs; }
        }
        // neighbouring lanes with the same pair insert once: the first lane of a run carries the run length
        const unsigned prev = __shfl_up(id, 1, 64);
        const bool head = lane == 0 || id != prev;
        const unsigned long long heads = __ballot(head);
        if (head && id != kSlotEmpty) {
            const unsigned long long above = (heads >> lane) >> 1;
            const unsigned run = above ? (unsigned)__ffsll((long long)above) : (unsigned)(64 - lane);
            unsigned h = (id * 2654435761u) >> 7 & mask;
            for (;;) {      // at most `cnt` distinct ids in >= 2 cnt slots: an empty slot is always found
                const unsigned was = atomicCAS(&tkey[h], kSlotEmpty, id);
                if (was == kSlotEmpty || was == id) break;
                h = (h + 1) & mask;
            }
            atomicAdd(&tcnt[h], run);
        }
    }
    __syncthreads();
    const unsigned rk = (unsigned)R * (unsigned)k;
    for (int sidx = tid; sidx < slots; sidx += kThreads) {
        const unsigned id = tkey[sidx];
        if (id != kSlotEmpty) atomicAdd(&hist[id / rk], tcnt[sidx] > 1 ? 2u : 1u);
    }
    __syncthreads();
    unsigned total;
    const unsigned mine = tid < buckets ? hist[tid] : 0u;
    const unsigned ex = block_excl_scan_u32(mine, &total);
    if (tid < buckets) { start[tid] = ex; hist[tid] = ex; }
    if (tid == buckets) start[tid] = total;
    unsigned n_valid;
    (void)block_excl_scan_u32(local_pairs, &n_valid);      // (also the barrier between `start` and its readers)
    for (int sidx = tid; sidx < slots; sidx += kThreads) {
        const unsigned id = tkey[sidx];
        if (id == kSlotEmpty) continue;
        const unsigned bk = id / rk, key = id - bk * rk, c = tcnt[sidx];
        const unsigned pos = atomicAdd(&hist[bk], c > 1 ? 2u : 1u);
        if (c > 1) {
            out[pos] = (unsigned short)(key | kElemMulti);
            out[pos + 1] = (unsigned short)(kElemCount | (c - 2));
        } else {
            out[pos] = (unsigned short)key;
        }
    }
    __syncthreads();
    // (the order inside a group depends on the scheduling of the LDS atomics; the counts do not)
    unsigned int* dst = reinterpret_cast<unsigned int*>(elems + (size_t)blockIdx.x * chunk);     // chunk is even
    const unsigned int* src = reinterpret_cast<const unsigned int*>(out);
    for (int i = tid; i < (int)((total + 1) >> 1); i += kThreads) dst[i] = src[i];
    if (tid <= buckets) offs[(size_t)blockIdx.x * (buckets + 1) + tid] = start[tid];     // one coalesced row per chunk
    if (tid == 0) valid_out[blockIdx.x] = n_valid;
}

// Pass 2.  grid = buckets rounded up to a multiple of 8; workgroup b owns the rows [b R, b R + R) of the matrix.
__global__ __launch_bounds__(kThreads) void count_bucket_bin_kernel(const unsigned short* __restrict__ elems,
                                                                   const unsigned int* __restrict__ offs, int chunks,
                                                                   int chunk, int k, int R, int n_buckets, int copies,
                                                                   unsigned long long* __restrict__ counts,
                                                                   const unsigned int* __restrict__ valid,
                                                                   unsigned long long* __restrict__ pairs_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int nb = R * k;                                                  // bins of one copy
    unsigned int* bins = reinterpret_cast<unsigned int*>(smem_raw);        // [copies][nb]
    unsigned int* pref = bins + (size_t)copies * nb;                       // [kThreads + 1] group prefix of this tile
    unsigned int* base = pref + kThreads + 1;                              // [kThreads] first element of the group in `elems`
    const int tid = threadIdx.x;
    // Workgroup ids that differ by multiples of 8 share an XCD (and its L2): give each XCD a CONTIGUOUS range of buckets.
    // The groups of neighbouring buckets lie next to each other in a chunk's element row and are shorter than a cache
    // line, so a line is then fetched from HBM by one XCD instead of by up to four (10.5 -> read-side MB in r03_pmc_hbm.md).
    const int per_xcd = (n_buckets + 7) / 8;       // grid = 8 per_xcd
    const int b = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (b >= n_buckets) return;          // (whole workgroup; the ids of a short last range)
    for (int i = tid; i < copies * nb; i += kThreads) bins[i] = 0;
    unsigned int* my_bins = bins + (size_t)((tid >> 6) & (copies - 1)) * nb;   // copies is a power of two
    for (int w0 = 0; w0 < chunks; w0 += kThreads) {
        const int w = w0 + tid;
        unsigned s = 0, len = 0;
        if (w < chunks) {
            s = offs[(size_t)w * (n_buckets + 1) + b];
            len = offs[(size_t)w * (n_buckets + 1) + b + 1] - s;
        }
        unsigned total;
        const unsigned ex = block_excl_scan_u32(len, &total);      // (its barriers also cover the zeroing above)
        pref[tid] = ex;
        base[tid] = (unsigned)w * (unsigned)chunk + s;             // chunks * chunk < 2^32 (launch_counts checks)
        __syncthreads();
        const int nseg = min(kThreads, chunks - w0);
        constexpr int kU = 4;                                      // independent searches and loads in flight per lane
        for (unsigned e0 = 0; e0 < total; e0 += kU * kThreads) {
            unsigned key[kU], add[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const unsigned e = e0 + u * kThreads + tid;
                key[u] = 0;
                add[u] = 0;
                if (e < total) {
                    int lo = 0, hi = nseg - 1;                     // last group with pref <= e
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        if (pref[mid] <= e) lo = mid; else hi = mid - 1;
                    }
                    const unsigned short* at = elems + (size_t)base[lo] + (e - pref[lo]);
                    const unsigned v = at[0];
                    if (!(v & kElemCount)) {                        // a count element is consumed with its key
                        key[u] = v & (kElemCount - 1u);
                        add[u] = (v & kElemMulti) ? (at[1] & (kElemCount - 1u)) + 2u : 1u;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < kU; ++u)
                if (add[u]) atomicAdd(&my_bins[key[u]], add[u]);
        }
        __syncthreads();
    }
    const int r0 = b * R;
    const int nrows = min(R, k - r0);
    unsigned long long* dst = counts + (size_t)r0 * k;
    for (int i = tid; i < nrows * k; i += kThreads) {
        unsigned long long v = 0;
        for (int c = 0; c < copies; ++c) v += bins[(size_t)c * nb + i];
        dst[i] = v;
    }
    if (b == 0 && pairs_out) {         // number of counted pairs: the chunks' valid counts, no atomics, no memset
        unsigned long long v = 0;
        for (int w = tid; w < chunks; w += kThreads) v += valid[w];
        __shared__ unsigned long long wave_part[kThreads / 64];
        v = wave_sum_u64(v);
        if ((tid & 63) == 0) wave_part[tid >> 6] = v;
        __syncthreads();
        if (tid == 0) {
            unsigned long long t = 0;
            for (int w = 0; w < kThreads / 64; ++w) t += wave_part[w];
            *pairs_out = t;
        }
    }
}

// Very large k: no privatisation, one global atomic per pair.
template <bool WEIGHTED>
__global__ __launch_bounds__(kThreads) void count_global_kernel(
    const int32_t* __restrict__ labels, const double* __restrict__ weights, SegTab st, int k,
    typename BinT<WEIGHTED>::out_t* __restrict__ counts, unsigned long long* __restrict__ pairs_out) {
    using out_t = typename BinT<WEIGHTED>::out_t;
    const int lag = st.lag;
    unsigned long long local_pairs = 0;
    const int64_t step = (int64_t)gridDim.x * kThreads;
    for (int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x; p < st.total_pairs; p += step) {
        const int64_t t = seg_pair_to_frame(st, p);
        const int a = labels[t];
        const int b = labels[t + lag];
        if ((unsigned)a < (unsigned)k && (unsigned)b < (unsigned)k) {
            if constexpr (WEIGHTED) atomicAdd(&counts[(size_t)a * k + b], weights[t]);
            else atomicAdd(&counts[(size_t)a * k + b], (out_t)1);
            ++local_pairs;
        }
    }
    if (pairs_out) block_add_u64(local_pairs, pairs_out);
}

__global__ __launch_bounds__(kThreads) void state_counts_kernel(
    const int32_t* __restrict__ labels, int64_t n, int k, unsigned long long* __restrict__ visits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned int* bins = reinterpret_cast<unsigned int*>(smem_raw);
    const bool use_lds = k <= 16384;
    if (use_lds) {
        for (int i = threadIdx.x; i < k; i += kThreads) bins[i] = 0;
        __syncthreads();
    }
    const int64_t step = (int64_t)gridDim.x * kThreads;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += step) {
        const int a = labels[i];
        if ((unsigned)a < (unsigned)k) {
            if (use_lds) atomicAdd(&bins[a], 1u);
            else atomicAdd(&visits[a], 1ull);
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < k; i += kThreads)
            if (bins[i]) atomicAdd(&visits[i], (unsigned long long)bins[i]);
    }
}

bool count_buckets_enabled() {
    static const bool on = [] {
        const char* e = getenv("MSM_COUNTS_BUCKETS");   // MSM_COUNTS_BUCKETS=0: the privatised kernel everywhere (A/B timing)
        return !(e && e[0] == '0');
    }();
    return on;
}

struct CountPlan {
    bool global_path;
    int rows, row_blocks, chunks;
    size_t lds_bytes;
    int64_t pairs_per_chunk;
};

CountPlan plan_counts(const msm_ctx* ctx, int k, int64_t total_pairs, size_t bin_bytes) {
    CountPlan pl{};
    const size_t row_bytes = (size_t)k * bin_bytes;
    size_t budget = (row_bytes * k <= (size_t)kLdsBudgetSmall) ? kLdsBudgetSmall : kLdsBudgetLarge;
    int rows = (int)(budget / row_bytes);
    if (rows > k) rows = k;
#ifdef MSM_COUNTS_FORCE_GLOBAL   // timing experiment (tools/build_variant.sh)
    rows = 0;
#endif
    if (rows < 1 || msm_ceil_div(k, rows) > kMaxRowBlocks) {
        pl.global_path = true;
        pl.chunks = (int)std::min<int64_t>(std::max<int64_t>(1, msm_ceil_div(total_pairs, kThreads * 8)),
                                           (int64_t)ctx->n_cu * 8);
        return pl;
    }
    pl.global_path = false;
    pl.row_blocks = msm_ceil_div(k, rows);
    pl.rows = msm_ceil_div(k, pl.row_blocks);  // balance the blocks
    pl.lds_bytes = (size_t)pl.rows * row_bytes;
    const int wg_per_cu = 1;  // 1024-thread workgroups
    int chunks = std::max(1, ctx->n_cu * wg_per_cu / pl.row_blocks);
    // keep at least ~2k pairs per workgroup so the LDS flush amortises
    const int64_t max_chunks = std::max<int64_t>(1, total_pairs / 2048);
    if (chunks > max_chunks) chunks = (int)max_chunks;
    pl.chunks = chunks;
    pl.pairs_per_chunk = (total_pairs + chunks - 1) / chunks;
    return pl;
}

template <bool WEIGHTED>
msm_status launch_counts(msm_ctx* ctx, const int32_t* d_labels, const double* d_weights,
                         const SegTab& st, int k, void* d_counts, int64_t* d_pairs) {
    using out_t = typename BinT<WEIGHTED>::out_t;
    using lds_t = typename BinT<WEIGHTED>::lds_t;
    if constexpr (!WEIGHTED) {
        // dense regime, a bucket of rows fits the LDS: two passes without global atomics (see count_bucket_*)
        const int R = msm_ceil_div(k, kBucketsMax), buckets = msm_ceil_div(k, R);
        const size_t bin_bytes = (size_t)R * k * sizeof(unsigned int);
        if (count_buckets_enabled() && bin_bytes <= (size_t)kBinCopiesBytes && st.total_pairs >= (int64_t)k * k &&
            st.total_pairs < ((int64_t)1 << 31)) {
            int chunk = (int)std::min<int64_t>(kChunkMax, std::max<int64_t>(2048, msm_ceil_div(st.total_pairs, ctx->n_cu)));
            if (const char* e = getenv("MSM_COUNTS_CHUNK")) chunk = std::max(64, std::min(kChunkMax, atoi(e)));   // tests: many chunks at small n
            chunk = (chunk + 63) & ~63;
            const int chunks = msm_ceil_div(st.total_pairs, chunk);
            int slots = 128;
            while (slots < 2 * chunk) slots *= 2;
            const size_t elem_bytes = ((size_t)chunks * chunk * sizeof(unsigned short) + 255) & ~(size_t)255;
            const size_t need = elem_bytes + (size_t)(buckets + 2) * chunks * sizeof(unsigned int);
            if (!(ctx->capturing && need > ctx->aux_bytes)) {      // (a capture cannot grow the scratch: privatised kernel)
                msm_status rs = msm_reserve_aux(ctx, need);
                if (rs != MSM_OK) return rs;
                unsigned short* elems = (unsigned short*)ctx->aux;
                unsigned int* offs = (unsigned int*)((char*)ctx->aux + elem_bytes);
                unsigned int* valid = offs + (size_t)(buckets + 1) * chunks;
                const size_t lds1 = (size_t)slots * 8 + (size_t)(2 * kBucketsMax + 2) * sizeof(unsigned int) + (size_t)chunk * 2;
                if (lds1 > 48 * 1024)
                    MSM_HIP(ctx, hipFuncSetAttribute((const void*)count_bucket_scatter_kernel,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
                hipLaunchKernelGGL(count_bucket_scatter_kernel, dim3(chunks), dim3(kThreads), lds1, ctx->stream, d_labels, st,
                                   k, R, buckets, chunk, slots, elems, offs, valid);
                int copies = 1;
                while (copies < kThreads / 64 && (size_t)2 * copies * bin_bytes <= (size_t)kBinCopiesBytes) copies *= 2;
                const size_t lds2 = (size_t)copies * bin_bytes + (size_t)(2 * kThreads + 1) * sizeof(unsigned int);
                if (lds2 > 48 * 1024)
                    MSM_HIP(ctx, hipFuncSetAttribute((const void*)count_bucket_bin_kernel,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
                hipLaunchKernelGGL(count_bucket_bin_kernel, dim3(8 * ((buckets + 7) / 8)), dim3(kThreads), lds2, ctx->stream,
                                   (const unsigned short*)elems, (const unsigned int*)offs, chunks, chunk, k, R, buckets, copies,
                                   (unsigned long long*)d_counts, (const unsigned int*)valid, (unsigned long long*)d_pairs);
                MSM_CHECK_LAUNCH(ctx);
                return MSM_OK;
            }
        }
    }
    MSM_HIP(ctx, hipMemsetAsync(d_counts, 0, (size_t)k * k * sizeof(out_t), ctx->stream));
    if (d_pairs) MSM_HIP(ctx, hipMemsetAsync(d_pairs, 0, sizeof(int64_t), ctx->stream));
    if (st.total_pairs == 0) return MSM_OK;
    CountPlan pl = plan_counts(ctx, k, st.total_pairs, sizeof(lds_t));
    if (pl.global_path) {
        hipLaunchKernelGGL(count_global_kernel<WEIGHTED>, dim3(pl.chunks), dim3(kThreads), 0, ctx->stream,
                           d_labels, d_weights, st, k, (out_t*)d_counts, (unsigned long long*)d_pairs);
    } else {
        auto kern = count_lds_kernel<WEIGHTED>;
        if (pl.lds_bytes > 64 * 1024)
            MSM_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)pl.lds_bytes));
        hipLaunchKernelGGL(kern, dim3(pl.chunks, pl.row_blocks), dim3(kThreads), pl.lds_bytes, ctx->stream,
                           d_labels, d_weights, st, k, pl.rows, pl.pairs_per_chunk, (out_t*)d_counts,
                           (unsigned long long*)d_pairs);
    }
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

}  // namespace

namespace {

// Dwell times: a run starts at frame t when label[t] is a state and differs from label[t-1] (negative
// labels act as separators between trajectories); the starting thread walks to the end of its run.
// Per state: min / max / sum of the run lengths and the number of runs (64-bit integer atomics: order
// free, exact); every run is also appended to a list for the medians.
__global__ __launch_bounds__(256) void run_lengths_kernel(const int32_t* __restrict__ labels, int64_t n, int k,
                                                          unsigned long long* __restrict__ stats,
                                                          int32_t* __restrict__ run_state,
                                                          long long* __restrict__ run_len, long long cap,
                                                          unsigned long long* __restrict__ n_runs) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int s = labels[t];
        if (s < 0 || s >= k) continue;
        if (t > 0 && labels[t - 1] == s) continue;
        int64_t u = t + 1;
        while (u < n && labels[u] == s) ++u;
        const unsigned long long len = (unsigned long long)(u - t);
        atomicMin(&stats[s], len);
        atomicMax(&stats[(size_t)k + s], len);
        atomicAdd(&stats[2 * (size_t)k + s], len);
        atomicAdd(&stats[3 * (size_t)k + s], 1ull);
        const unsigned long long slot = atomicAdd(n_runs, 1ull);
        if ((long long)slot < cap) { run_state[slot] = s; run_len[slot] = (long long)len; }
    }
}

}  // namespace

extern "C" {

msm_status msm_count_transitions(msm_ctx* ctx, const int32_t* d_labels, int64_t n,
                                 const int64_t* h_seg_start, const int64_t* h_seg_stop, int n_seg,
                                 int lag, int stride, int k, int64_t* d_counts, int64_t* d_pairs) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 0 && k >= 1, "msm_count_transitions: need n >= 0 and k >= 1 (n=%lld k=%d)",
                (long long)n, k);
    MSM_REQUIRE(ctx, k <= 46340, "msm_count_transitions: k=%d too large", k);
    MSM_REQUIRE(ctx, d_counts && (d_labels || n == 0), "msm_count_transitions: NULL pointer");
    SegTab st;
    msm_status rs = msm_build_segtab(ctx, n, h_seg_start, h_seg_stop, n_seg, lag, stride, &st, 0);
    if (rs != MSM_OK) return rs;
    return launch_counts<false>(ctx, d_labels, nullptr, st, k, d_counts, d_pairs);
}

msm_status msm_count_transitions_weighted(msm_ctx* ctx, const int32_t* d_labels,
                                          const double* d_weights, int64_t n,
                                          const int64_t* h_seg_start, const int64_t* h_seg_stop,
                                          int n_seg, int lag, int stride, int k, double* d_counts,
                                          int64_t* d_pairs) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 0 && k >= 1, "msm_count_transitions_weighted: need n >= 0 and k >= 1");
    MSM_REQUIRE(ctx, k <= 46340, "msm_count_transitions_weighted: k=%d too large", k);
    MSM_REQUIRE(ctx, d_counts && ((d_labels && d_weights) || n == 0),
                "msm_count_transitions_weighted: NULL pointer");
    SegTab st;
    msm_status rs = msm_build_segtab(ctx, n, h_seg_start, h_seg_stop, n_seg, lag, stride, &st, 0);
    if (rs != MSM_OK) return rs;
    return launch_counts<true>(ctx, d_labels, d_weights, st, k, d_counts, d_pairs);
}

msm_status msm_count_transitions_lagscan(msm_ctx* ctx, const int32_t* d_labels, int64_t n,
                                         const int64_t* h_seg_start, const int64_t* h_seg_stop,
                                         int n_seg, const int32_t* h_lags, int n_lag, int k,
                                         int64_t* d_counts, int64_t* d_pairs) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n_lag >= 0 && (h_lags || n_lag == 0), "msm_count_transitions_lagscan: bad lag list");
    MSM_REQUIRE(ctx, n >= 0 && k >= 1 && k <= 46340, "msm_count_transitions_lagscan: bad n/k");
    MSM_REQUIRE(ctx, d_counts && (d_labels || n == 0), "msm_count_transitions_lagscan: NULL pointer");
    // One launch per lag into its own k*k slice: the labels stay L2-resident
    // across lags, and under msm_graph_begin/end the launches fuse into one graph.
    for (int l = 0; l < n_lag; ++l) {
        SegTab st;
        msm_status rs = msm_build_segtab(ctx, n, h_seg_start, h_seg_stop, n_seg, h_lags[l], 1, &st, l);
        if (rs != MSM_OK) return rs;
        rs = launch_counts<false>(ctx, d_labels, nullptr, st, k, d_counts + (size_t)l * k * k,
                                  d_pairs ? d_pairs + l : nullptr);
        if (rs != MSM_OK) return rs;
    }
    return MSM_OK;
}

msm_status msm_state_counts(msm_ctx* ctx, const int32_t* d_labels, int64_t n, int k, int64_t* d_visits) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 0 && k >= 1, "msm_state_counts: need n >= 0 and k >= 1");
    MSM_REQUIRE(ctx, d_visits && (d_labels || n == 0), "msm_state_counts: NULL pointer");
    MSM_HIP(ctx, hipMemsetAsync(d_visits, 0, (size_t)k * sizeof(int64_t), ctx->stream));
    if (n == 0) return MSM_OK;
    const int blocks = (int)std::min<int64_t>(std::max<int64_t>(1, msm_ceil_div(n, kThreads * 16)),
                                              (int64_t)ctx->n_cu * 2);
    const size_t lds = k <= 16384 ? (size_t)k * sizeof(unsigned int) : 0;
    hipLaunchKernelGGL(state_counts_kernel, dim3(blocks), dim3(kThreads), lds, ctx->stream, d_labels, n, k,
                       (unsigned long long*)d_visits);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

msm_status msm_run_lengths(msm_ctx* ctx, const int32_t* d_labels, int64_t n, int k, int64_t* d_stats,
                           int32_t* d_run_state, int64_t* d_run_len, int64_t capacity, int64_t* d_n_runs) {
    if (!ctx) return MSM_ERR_INVALID;
    MSM_REQUIRE(ctx, n >= 0 && k >= 1 && capacity >= 0, "msm_run_lengths: bad shape");
    MSM_REQUIRE(ctx, d_stats && d_n_runs && (d_labels || n == 0) && (capacity == 0 || (d_run_state && d_run_len)),
                "msm_run_lengths: NULL pointer");
    MSM_HIP(ctx, hipMemsetAsync(d_stats, 0xFF, (size_t)k * sizeof(int64_t), ctx->stream));            // min: all ones
    MSM_HIP(ctx, hipMemsetAsync(d_stats + k, 0, (size_t)3 * k * sizeof(int64_t), ctx->stream));
    MSM_HIP(ctx, hipMemsetAsync(d_n_runs, 0, sizeof(int64_t), ctx->stream));
    if (n == 0) return MSM_OK;
    const int blocks = (int)std::min<int64_t>(std::max<int64_t>(1, msm_ceil_div(n, 256 * 4)), (int64_t)ctx->n_cu * 8);
    hipLaunchKernelGGL(run_lengths_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_labels, n, k,
                       (unsigned long long*)d_stats, d_run_state, (long long*)d_run_len, (long long)capacity,
                       (unsigned long long*)d_n_runs);
    MSM_CHECK_LAUNCH(ctx);
    return MSM_OK;
}

}  // extern "C"
