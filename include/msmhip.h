/*
 * msmhip.h -- C ABI of libmsmhip.so, the MI355X (gfx950) engine behind the
 * pmarlo featurize -> TICA -> k-means -> transition-matrix -> ITS path.
 *
 * The reference (pmarlo, pure Python) has no FFI for this path: the seam is
 * "Python callables taking/returning numpy arrays" (SURVEY.md section 8b).  Each
 * entry point below therefore names the reference callable whose arithmetic it
 * replaces (file:line relative to the pmarlo tree, src/pmarlo = S/).  The
 * ctypes binding a pmarlo maintainer would add is shown in INTEGRATION.md and
 * implemented in pmarlo_amd/_lib.py.
 *
 * Conventions
 *   - every function returns an msm_status; 0 is success.  msm_last_error()
 *     gives the message for the last failure on that context.
 *   - pointers named d_* are DEVICE pointers (hipMalloc'ed, e.g. msm_malloc or
 *     a torch tensor's data_ptr()); pointers named h_* are HOST pointers.
 *   - matrices are row-major; `ld` is the row stride in ELEMENTS.
 *   - all work is enqueued on the context's HIP stream; nothing synchronises
 *     unless documented (msm_sync, msm_memcpy_d2h, functions with h_ outputs).
 *   - no torch / C++ types cross this boundary.
 */
#ifndef MSMHIP_H
#define MSMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct msm_ctx msm_ctx;
typedef struct msm_event msm_event;
typedef struct msm_graph msm_graph;

typedef enum {
    MSM_OK = 0,
    MSM_ERR_INVALID = 1,     /* bad argument            -> ValueError   */
    MSM_ERR_HIP = 2,         /* HIP runtime failure     -> RuntimeError */
    MSM_ERR_NOMEM = 3,       /* allocation failure      -> MemoryError  */
    MSM_ERR_UNSUPPORTED = 4, /* shape outside kernel limits -> NotImplementedError */
    MSM_ERR_NOCONV = 5       /* iterative solver did not converge -> RuntimeError */
} msm_status;

/* element type of a feature matrix handed to the engine */
typedef enum { MSM_F32 = 0, MSM_F64 = 1 } msm_dtype;

/* ------------------------------------------------------------------ */
/* context, memory, timing                                              */
/* ------------------------------------------------------------------ */

/* Create a context on HIP device `device`.  `hip_stream` may be NULL (the
 * context then owns a new non-blocking stream) or an existing hipStream_t
 * (e.g. torch.cuda.current_stream().cuda_stream) which is borrowed. */
msm_status msm_ctx_create(int device, void* hip_stream, msm_ctx** out);
void msm_ctx_destroy(msm_ctx* ctx);
const char* msm_last_error(const msm_ctx* ctx);
/* "gfx950" etc.; number of compute units; bytes of device memory */
msm_status msm_device_info(msm_ctx* ctx, char* arch, size_t arch_len,
                           int* n_cu, size_t* total_mem);
const char* msm_version(void);

msm_status msm_malloc(msm_ctx* ctx, size_t bytes, void** d_out);
msm_status msm_free(msm_ctx* ctx, void* d_ptr);
msm_status msm_memcpy_h2d(msm_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
msm_status msm_memcpy_d2h(msm_ctx* ctx, void* h_dst, const void* d_src, size_t bytes); /* syncs */
msm_status msm_memcpy_d2d(msm_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);
msm_status msm_memset(msm_ctx* ctx, void* d_dst, int value, size_t bytes);
msm_status msm_sync(msm_ctx* ctx);

/* HIP events on the context's stream (bench.py times kernels with these). */
msm_status msm_event_create(msm_ctx* ctx, msm_event** out);
void msm_event_destroy(msm_event* ev);
msm_status msm_event_record(msm_ctx* ctx, msm_event* ev);
msm_status msm_event_elapsed_ms(msm_event* start, msm_event* stop, float* h_ms); /* syncs on stop */

/* Stream capture: everything enqueued between begin and end becomes one
 * hipGraph that msm_graph_launch replays (launch-bound inner loops). */
msm_status msm_graph_begin(msm_ctx* ctx);
msm_status msm_graph_end(msm_ctx* ctx, msm_graph** out);
msm_status msm_graph_launch(msm_ctx* ctx, msm_graph* g);
void msm_graph_destroy(msm_graph* g);

/* ------------------------------------------------------------------ */
/* featurizers                                                          */
/* ------------------------------------------------------------------ */

/* Per-frame geometry on d_xyz float32 [n, A, 3] (nm), written as float32 columns
 * [col_off, col_off + width) of d_out [n, ld].  fp32 arithmetic in the operation
 * order of the reference's in-repo extractor (S/features/deeptica/
 * ts_feature_extractor.py:423-500, use_pbc = False), which is also what stands in
 * for mdtraj.compute_distances / compute_dihedrals behind featurize_trajectory
 * (S/features/featurize.py:41-62), PhiPsiFeature / DistanceFeature / AngleFeature /
 * DihedralFeature (S/features/builtins.py:42-86, 281-395) and
 * FeaturesMixin._compute_phi_psi_features (S/markov_state_model/_features.py:131-142).
 *   distances: d_pairs int32 [P, 2]      d = sqrt(max(|x_j - x_i|^2, 1e-12))
 *   angles   : d_triplets int32 [T, 3]   acos(clamp(v1.v2 / (|v1||v2|), -1, 1)), vertex = middle atom
 *   dihedrals: d_quads int32 [Q, 4]      atan2 form, wrapped to (-pi, pi] (builtins.py:11-14)
 *       mode 0: Q columns of radians
 *       mode 1: 2Q columns [cos_0, sin_0, cos_1, sin_1, ...]  (trig_expand_periodic,
 *               S/api/features.py:138-180)
 *       mode 2: 2Q columns [cos_0..cos_{Q-1} | sin_0..sin_{Q-1}] (_features.py:131-142) */
msm_status msm_featurize_distances(msm_ctx* ctx, const float* d_xyz, int64_t n, int A,
                                   const int32_t* d_pairs, int P, float* d_out, int64_t ld, int col_off);
/* contacts: 1.0 where the pair distance is <= rcut (nm), else 0.0 (NaN -> 0); Rg: radius of gyration
 * with unit masses, one column (S/features/builtins.py:89-108, 252-275). */
msm_status msm_featurize_contacts(msm_ctx* ctx, const float* d_xyz, int64_t n, int A,
                                  const int32_t* d_pairs, int P, float rcut, float* d_out,
                                  int64_t ld, int col_off);
msm_status msm_featurize_rg(msm_ctx* ctx, const float* d_xyz, int64_t n, int A, float* d_out,
                            int64_t ld, int col_off);
msm_status msm_featurize_angles(msm_ctx* ctx, const float* d_xyz, int64_t n, int A,
                                const int32_t* d_triplets, int Tn, float* d_out, int64_t ld, int col_off);
msm_status msm_featurize_dihedrals(msm_ctx* ctx, const float* d_xyz, int64_t n, int A,
                                   const int32_t* d_quads, int Q, int mode, float* d_out, int64_t ld,
                                   int col_off);

/* Structure features whose reference implementation is an mdtraj algorithm (S/features/builtins.py:171-250: the
 * built-ins "sasa", "hbonds_count", "ssfrac", all three in the default feature set of S/api/features.py:378).
 *   sasa: Shrake-Rupley accessible area per atom, float32 [n, A] (nm^2), as mdtraj.shrake_rupley computes it
 *         (geometry/src/sasa.cpp) before its per-residue sums: d_radii float32 [A] = atomic radius + probe radius,
 *         d_points float32 [P, 3] = the unit sphere points (golden spiral), fp32 throughout.
 *   hbond presence: for every (donor, hydrogen, acceptor) row of d_triplets int32 [Tn, 3] the number of frames with
 *         |H - A| < dist_cutoff and angle(D, H, A) > angle_cutoff (radians), the two criteria of
 *         mdtraj.baker_hubbard (geometry/hbond.py); the frequency threshold is the caller's.
 *   dssp: Kabsch-Sander secondary structure per frame and residue (mdtraj.compute_dssp, geometry/src/dssp.cpp):
 *         d_backbone int32 [R, 4] = atom indices of N, CA, C, O of each protein residue, d_chain int32 [R],
 *         d_proline uint8 [R]; d_codes uint8 [n, R]: 0 loop, 1 H, 2 B, 3 E, 4 G, 5 I, 6 T, 7 S. */
msm_status msm_featurize_sasa(msm_ctx* ctx, const float* d_xyz, int64_t n, int A, const float* d_radii,
                              const float* d_points, int P, float* d_out);
msm_status msm_hbond_presence(msm_ctx* ctx, const float* d_xyz, int64_t n, int A, const int32_t* d_triplets,
                              int Tn, float dist_cutoff, float angle_cutoff, uint64_t* d_counts);
msm_status msm_dssp(msm_ctx* ctx, const float* d_xyz, int64_t n, int A, const int32_t* d_backbone,
                    const int32_t* d_chain, const uint8_t* d_proline, int R, uint8_t* d_codes);

/* ------------------------------------------------------------------ */
/* lag-tau transition counts                                            */
/* ------------------------------------------------------------------ */

/* Unweighted lag-tau transition counts over trajectory segments.
 * Replaces pmarlo.analysis.discretize._weighted_counts (S/analysis/
 * discretize.py:609-645, weights=None) and, with stride=1, the deeptime
 * "sliding" count of EstimationMixin._count_transitions_deeptime
 * (S/markov_state_model/_estimation.py:116-156) and
 * debug_export._build_transition_counts (S/analysis/debug_export.py:385-409).
 *
 * For every segment [start, stop) with stop-start > lag and every
 * t = start, start+stride, ... < stop-lag: if labels[t] and labels[t+lag] are
 * both in [0, k) then counts[labels[t]][labels[t+lag]] += 1.
 * (The reference keeps pairs with both labels >= 0; labels >= k cannot occur
 * there because k = max label + 1.  Here they are skipped, never written.)
 *
 * d_labels  int32 [n]           h_seg_start/h_seg_stop  int64 [n_seg] (clipped to [0,n])
 * d_counts  int64 [k*k]  OVERWRITTEN   d_pairs  int64 [1] OVERWRITTEN (may be NULL)
 */
msm_status msm_count_transitions(msm_ctx* ctx, const int32_t* d_labels, int64_t n,
                                 const int64_t* h_seg_start, const int64_t* h_seg_stop,
                                 int n_seg, int lag, int stride, int k,
                                 int64_t* d_counts, int64_t* d_pairs);

/* Weighted variant: counts[src][dst] += w[t] (weight of the STARTING frame,
 * S/analysis/discretize.py:633-641).  d_counts is float64 [k*k]. The sum is
 * accumulated with fp64 atomics, so it equals the reference up to summation
 * order (exact whenever the weights are dyadic / integer valued). */
msm_status msm_count_transitions_weighted(msm_ctx* ctx, const int32_t* d_labels,
                                          const double* d_weights, int64_t n,
                                          const int64_t* h_seg_start, const int64_t* h_seg_stop,
                                          int n_seg, int lag, int stride, int k,
                                          double* d_counts, int64_t* d_pairs);

/* Lag scan: counts for n_lag lags in one launch (ITS scan,
 * S/markov_state_model/_its.py:525-541 called once per lag).
 * h_lags int32 [n_lag]; d_counts int64 [n_lag*k*k]; d_pairs int64 [n_lag]. */
msm_status msm_count_transitions_lagscan(msm_ctx* ctx, const int32_t* d_labels, int64_t n,
                                         const int64_t* h_seg_start, const int64_t* h_seg_stop,
                                         int n_seg, const int32_t* h_lags, int n_lag, int k,
                                         int64_t* d_counts, int64_t* d_pairs);

/* Visits per state: np.bincount(labels[0<=l<k], minlength=k)
 * (S/analysis/discretize.py:648-667, weights=None).  d_visits int64 [k]. */
msm_status msm_state_counts(msm_ctx* ctx, const int32_t* d_labels, int64_t n, int k,
                            int64_t* d_visits);

/* Dwell times (runs of one state) of a label sequence in which negative labels separate trajectories
 * (_compute_dwell_times, S/analysis/debug_export.py:447-530, after its removal of unassigned frames):
 * d_stats int64 [4][k] = {min, max, sum, number} of the run lengths per state (min = -1 for a state
 * without runs); every run is also appended, in no particular order, to d_run_state int32 / d_run_len
 * int64 [capacity] (for the medians; capacity >= number of runs, at most n), *d_n_runs = runs found. */
msm_status msm_run_lengths(msm_ctx* ctx, const int32_t* d_labels, int64_t n, int k, int64_t* d_stats,
                           int32_t* d_run_state, int64_t* d_run_len, int64_t capacity, int64_t* d_n_runs);

/* ------------------------------------------------------------------ */
/* standardisation moments, time-lagged covariance, TICA               */
/* ------------------------------------------------------------------ */

/* Column moments of X [n, ld].  Replaces the statistics half of
 * reduction._preprocess (S/markov_state_model/reduction.py:13-40: SimpleImputer
 * mean + StandardScaler, ddof = 0) and of _KMeansDiscretizer.fit
 * (S/analysis/discretize.py:446-447: np.mean / np.std(ddof=1)).
 * NaN entries are skipped (the reference imputes them with the column mean,
 * which leaves mean and the centred sum of squares unchanged except for the
 * divisor, see msm_moments_finalize).
 *
 * _partial writes raw sums d_sums = [cnt F][S1 F][S2 F] with S1 = sum(x-shift),
 * S2 = sum((x-shift)^2); d_shift NULL means "row 0 of X" (NaN -> 0) and the
 * shift used is returned in d_shift_out [F] (may be NULL).  Shards that share
 * one shift vector all-reduce d_sums by plain summation.
 * _finalize: mean = shift + S1/cnt, std = sqrt((S2 - S1^2/cnt)/(cnt - ddof)).
 * msm_column_moments = partial + finalize.  d_count [F] f64 may be NULL. */
msm_status msm_column_moments_partial(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int F,
                                      int64_t ld, const double* d_shift, double* d_sums,
                                      double* d_shift_out);
msm_status msm_moments_finalize(msm_ctx* ctx, const double* d_sums, const double* d_shift, int F,
                                int ddof, double* d_mean, double* d_std, double* d_count);
/* The remaining column statistics of validate_features (S/analysis/validation.py:89-172): d_min / d_max [F] over the
 * finite entries of every column (NaN for a column without any), d_counts int64 [2] = {non-finite entries, rows that
 * are finite throughout}.  With msm_column_moments this replaces four host passes over the matrix. */
msm_status msm_column_minmax(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int F, int64_t ld,
                             double* d_min, double* d_max, int64_t* d_counts);
msm_status msm_column_moments(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int F,
                              int64_t ld, int ddof, double* d_mean, double* d_std, double* d_count);

/* mean / divisor / reciprocal divisor of reduction._preprocess (reduction.py:13-40) from the
 * raw sums, on the device (no host round trip between the moments and covariance passes):
 * mean = shift + S1/cnt; sigma = sqrt((S2 - S1^2/cnt)/n_rows) -- NaNs imputed with the
 * column mean count in n_rows (the total row count over all shards) -- and sigma < 10 eps
 * -> 1 (sklearn's zero-scale rule); with_std == 0 gives sigma = 1 (scale=False). */
msm_status msm_standardise_params(msm_ctx* ctx, const double* d_sums, const double* d_shift, int F,
                                  double n_rows, int with_std, double* d_mean, double* d_scale,
                                  double* d_inv_scale);

/* Raw time-lagged moments on the fp64 matrix cores.  Replaces the covariance
 * accumulation inside deeptime TICA.fit as called by reduction.tica_reduce
 * (S/markov_state_model/reduction.py:103-109) and FeaturesMixin._maybe_apply_tica
 * (S/markov_state_model/_features.py:194-202): per segment x = X[:-lag],
 * y = X[lag:] (pairs never cross segments), reversible estimator.
 * With z = x - shift (NaN -> 0, i.e. imputed to the column mean):
 *   d_moments = [M00 F*F][M0t F*F][sx F][sy F][T]   (2F^2 + 2F + 1 doubles)
 *   M00 = sum_{X0} z z' + sum_{Yt} z z',  M0t = sum_pairs z_t z_{t+lag}',
 *   sx / sy = column sums over X0 / Yt, T = number of pairs.
 * Un-normalised on purpose: shards all-reduce d_moments by summation
 * (they must share d_shift).  At most 16 segments per call.  F <= 64 runs the fused
 * single-pass kernel; larger F the 64 x 64 block-task kernel.
 * assume_finite != 0 skips the NaN test (callers know from msm_column_moments'
 * d_count whether X holds NaNs; with NaNs present it must be 0).
 * lag == 0 gives the instantaneous second moments (M00 = 2 sum z z', T = n): the covariance
 * behind pca_reduce (S/markov_state_model/reduction.py:43-74). */
msm_status msm_lagged_moments(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int F, int64_t ld,
                              const int64_t* h_seg_start, const int64_t* h_seg_stop, int n_seg, int lag,
                              const double* d_shift, int assume_finite, double* d_moments);

/* The same pass with the M0t block SYMMETRISED: the slot holds (M0t + M0t') / 2, everything else as above.  That is
 * all the reversible estimator reads of M0t (deeptime forms C0t = (X'Y + Y'X) / 2T), and it is cheaper: for
 * 32 < F <= 64 the kernel accumulates S = sum_pairs (z_t + z_{t+lag})(z_t + z_{t+lag})' and M00, both symmetric
 * (2 x F(F+16)/2 multiply-adds per frame instead of F^2 + F(F+16)/2), and finishes with (S - M00) / 2; other shapes
 * run the plain pass and symmetrise in place.  Still additive over shards. */
msm_status msm_lagged_moments_reversible(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int F, int64_t ld,
                                         const int64_t* h_seg_start, const int64_t* h_seg_stop, int n_seg, int lag,
                                         const double* d_shift, int assume_finite, double* d_moments);

/* The same pass with ONE-SIDED second moments: M00 = sum over X0 only (M0t, sx, sy, T as above).
 * What the reference's in-repo TICA eigenvalue estimator needs (separate means and covariance of
 * y_t: _estimate_top_eigenvalues, S/features/deeptica/core/trainer_api.py:641-646). */
msm_status msm_lagged_moments_onesided(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int F,
                                       int64_t ld, const int64_t* h_seg_start, const int64_t* h_seg_stop,
                                       int n_seg, int lag, const double* d_shift, int assume_finite,
                                       double* d_moments);

/* _estimate_top_eigenvalues (S/features/deeptica/core/trainer_api.py:632-656) from one-sided moments:
 *   C0 = (M00 - sx sx'/T)/max(1, T-1), Ct = (M0t - sx sy'/T)/max(1, T-1);
 *   eigh(sym C0) with eigenvalues clipped at `clip` (the reference's NUMERIC_MIN_POSITIVE = 1e-12);
 *   S = C0^-1/2;  d_eigvals [F] = eigvalsh(sym(S Ct S')) in DESCENDING order (the caller takes the top n_out).
 * One launch, no host round trip; F <= 256. */
msm_status msm_onesided_tica_eigenvalues(msm_ctx* ctx, const double* d_moments, int F, double clip,
                                         double* d_eigvals);

/* The raw column sums msm_column_moments_partial would give ([cnt F][S1 F][S2 F] about d_shift, over every
 * frame of the segments) derived from the lagged moments of the SAME shift and segments plus the first
 * and last `lag` frames of each segment: 2 S1 = sx + sy + edges, 2 S2 = diag(M00) + edges^2.  Saves the
 * separate standardisation pass over X when the covariance pass runs anyway.  Finite data only (the
 * lagged moments impute NaN as 0 about the shift, which is the column mean only in the two-pass order);
 * at most 16 segments, lag >= 1.  Additive across shards like the inputs. */
msm_status msm_moments_from_lagged(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int F, int64_t ld,
                                   const int64_t* h_seg_start, const int64_t* h_seg_stop, int n_seg, int lag,
                                   const double* d_shift, const double* d_moments, double* d_sums);

/* TICA solve on the device (deeptime 0.4.5 TICA._decomposition semantics):
 *   mean = (sx+sy)/(2T); C00 = M00/(2T) - mean mean'; C0t = (M0t+M0t')/(2T) - mean mean'
 *   (all divided by d_scale[i]*d_scale[j] when d_scale != NULL, i.e. covariances of
 *   the standardised data (x-shift)/scale);
 *   C00 = V S V' (Jacobi), keep |s| >= epsilon, canonical signs, L = V S^-1/2;
 *   eigh(L' C0t L), sort by descending magnitude, R = L R', canonical signs,
 *   kinetic_map != 0: column i of R scaled by eigenvalue i.
 * Outputs: d_eigvals [F] (0 beyond rank), d_coeffs [F, F] row-major with column i
 * = i-th TICA component (0 beyond rank), d_mean [F] = symmetric mean in
 * standardised coordinates, d_rank int32 [1].  One launch, no host sync. */
msm_status msm_tica_solve(msm_ctx* ctx, const double* d_moments, const double* d_scale, int F,
                          double epsilon, int kinetic_map, double* d_eigvals, double* d_coeffs,
                          double* d_mean, int* d_rank);

/* Y[t][c] = sum_f (((x[t][f] - mu[f]) * inv_sigma[f]) - mean2[f]) * W[f][c], fp64 FMA chain
 * over ascending f.  Replaces model.transform(X_prep) of tica_reduce
 * (S/markov_state_model/reduction.py:109).  NaN -> 0 after centring.
 * d_mean2 may be NULL.  W is [F, ldw] (first d columns used), Y is [n, ldy] f64.
 * d_absmax (f64 [1], may be NULL) receives max |Y| from the same pass: pointing it at slot 2 of a
 * k-means fit state and calling msm_kmeans_fit_begin with absmax_ready = 1 saves that call's own
 * pass over Y. */
msm_status msm_project(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int F, int64_t ld,
                       const double* d_mu, const double* d_inv_sigma, const double* d_mean2,
                       const double* d_w, int d, int64_t ldw, double* d_y, int64_t ldy, double* d_absmax);

/* The same projection for X known to hold no NaN (msm_column_moments' d_count tells): the NaN test per element is
 * left out, same results on finite data.  With NaNs present the plain entry must be used. */
msm_status msm_project_finite(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int F, int64_t ld,
                              const double* d_mu, const double* d_inv_sigma, const double* d_mean2,
                              const double* d_w, int d, int64_t ldw, double* d_y, int64_t ldy, double* d_absmax);

/* Symmetric eigendecomposition by parallel cyclic Jacobi (n <= 256), ascending
 * eigenvalues d_w [n], eigenvectors in the columns of d_v [n, n] (may be NULL).
 * Replaces np.linalg.eigh on the path (e.g. _estimate_top_eigenvalues,
 * S/features/deeptica/core/trainer_api.py:646-651). d_sweeps int32 [1] may be NULL. */
msm_status msm_eigh(msm_ctx* ctx, const double* d_a, int n, double* d_w, double* d_v, int* d_sweeps);

/* ------------------------------------------------------------------ */
/* k-means                                                              */
/* ------------------------------------------------------------------ */

/* Assign every frame to its nearest centre.
 * Replaces _KMeansDiscretizer.transform (S/analysis/discretize.py:471-494:
 * whiten with (x-mean)/std_safe, then sklearn predict) and the deeptime
 * model.transform of cluster_microstates (S/markov_state_model/
 * clustering.py:608-609).
 *
 *   z      = d_mean ? (x - mean[f]) / std[f] : x        (fp64, IEEE division)
 *   dot_j  = fma chain over f = 0..d-1 of z[f]*c[j][f], starting from +0.0
 *   dist_j = fma(-2.0, dot_j, |c_j|^2)      |c_j|^2 = fma chain of c[j][f]^2
 *   label  = first j with minimal dist_j (strict <, ties -> lowest index)
 *
 * which is sklearn's argmin_j(|c_j|^2 - 2 x.c_j) (lloyd_iter_chunked_dense)
 * with the summation order pinned; oracle/msm_oracle.c restates it with C fma().
 *
 * d_x  [n, ld] dtype   d_centers f64 [k, d]   d_mean/d_std f64 [d] or NULL   (1 <= d <= 256;
 * wider frames return MSM_ERR_UNSUPPORTED)
 * d_labels int32 [n]   d_mindist f64 [n] or NULL (dist_j + |z|^2 >= 0 clipped,
 * i.e. the squared distance to the chosen centre, for inertia)
 */
msm_status msm_kmeans_assign(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int d,
                             int64_t ld, const double* d_centers, int k,
                             const double* d_mean, const double* d_std,
                             int32_t* d_labels, double* d_mindist);

/* ---- k-means fit (Lloyd) ------------------------------------------------
 * Replaces the estimator.fit of _KMeansDiscretizer.fit (S/analysis/discretize.py:
 * 458-469, sklearn KMeans / MiniBatchKMeans) and of cluster_microstates
 * (S/markov_state_model/clustering.py:605-609, deeptime KMeans.fit_fetch).
 * Bit-level parity with those RNG-driven fits is not attainable (SURVEY.md hard
 * part 2): the contract is (a) assignment parity given identical centres
 * (msm_kmeans_assign), (b) inertia no worse than the reference's, (c) run-to-run
 * and shard-count independence of the result for a fixed seed.
 *
 * Initial centres: k frames drawn by stratified sampling along the time axis
 * (frame floor((j + u_j) n / k), u_j = splitmix64(seed, j)), whitened like the data.
 * Iteration: assign (same arithmetic as msm_kmeans_assign) + accumulate member sums
 * in 64-bit fixed point (integer atomics: order independent, exact across shards),
 * then centres = sums / counts; empty clusters keep their centre.
 *
 * d_state: 8 doubles on the device = {scale, inv_scale, absmax, shift2, tol2, done,
 * n_iter, inertia}.  Iterations after convergence (shift2 <= tol2) are no-ops, so a
 * fixed launch sequence (or a captured graph) needs no host round trip.
 *
 * msm_kmeans_fit        = begin + max_iter x (accumulate + update) on one device.
 * msm_kmeans_fit_begin  computes the fixed-point scale from max|z| and n_total (the
 *                       frame count over ALL shards) and, if init_centers != 0, the
 *                       initial centres.  Shards then all-reduce MIN of state[0].
 *                       absmax_ready != 0: state[2] already holds max|x| (msm_project's d_absmax);
 *                       only without whitening (d_mean == NULL).
 * msm_kmeans_accumulate adds this shard's member sums / counts into d_sums int64 [k*d],
 *                       d_counts int64 [k] (caller zeroes them; all-reduce SUM across shards).
 * msm_kmeans_update     centres <- sums/counts, shift2, done, n_iter; clear != 0 zeroes
 *                       the accumulators for the next iteration. */
msm_status msm_kmeans_fit(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int d, int64_t ld,
                          const double* d_mean, const double* d_std, int k, uint64_t seed,
                          int init_centers, int max_iter, double tol2, double* d_centers,
                          double* d_state);
/* k-means++ seeding (deeptime KMeans init_strategy = 'kmeans++', S/markov_state_model/clustering.py:322-361;
 * sklearn KMeans init = 'k-means++', S/analysis/discretize.py:458-469): after msm_kmeans_fit_begin (which leaves
 * max |z| in d_state[2]) this replaces the k centres by frames drawn with probability proportional to the squared
 * distance to the nearest centre already drawn.  Integer weights and splitmix64 draws (csrc/kmeanspp.hip): the
 * frames drawn are a function of the data and the seed alone, and oracle/npport.kmeans_plusplus repeats them bit
 * for bit.  d_picked (int64 [k], may be NULL) receives the frame numbers. */
msm_status msm_kmeans_init_plusplus(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int d, int64_t ld,
                                    const double* d_mean, const double* d_std, int k, uint64_t seed, double n_total,
                                    double* d_centers, const double* d_state, int64_t* d_picked);
msm_status msm_kmeans_fit_begin(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int d,
                                int64_t ld, const double* d_mean, const double* d_std, int k,
                                uint64_t seed, int init_centers, double n_total, double tol2,
                                double* d_centers, double* d_state, int absmax_ready);
msm_status msm_kmeans_accumulate(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int d,
                                 int64_t ld, const double* d_centers, int k, const double* d_mean,
                                 const double* d_std, const double* d_state, int64_t* d_sums,
                                 int64_t* d_counts);
msm_status msm_kmeans_update(msm_ctx* ctx, int64_t* d_sums, int64_t* d_counts, int k, int d,
                             double* d_centers, double* d_state, int clear);

/* Frame images for the certified bf16 filter (pmarlo_amd/csrc/kmeans_filter.h).  For d <= 10 and a centre
 * table that fits the LDS (k <= ~750 at d = 10) msm_kmeans_assign / _accumulate / _fit take the labels from a
 * bf16 matrix-core pass over three-way bf16 splits of the coordinates: an upper bound of every pinned fp64 score,
 * the winner's candidates re-scored with the pinned fp64 chain and accepted only when they beat every bound
 * outside the candidate set; the remaining ~1 % of the frames take an exhaustive fp64 scan.  Labels, ties,
 * distances and member sums are those of the all-fp64 arithmetic stated at msm_kmeans_assign, bit for bit
 * (same reference lines: S/analysis/discretize.py:471-494, S/markov_state_model/clustering.py:608-609).
 *
 * The image of a frame (its split coordinates in matrix-operand order, 64 or 128 bytes) depends only on the
 * frame and the whitening, so a caller that runs many passes over the same frames (Lloyd iterations) builds
 * it once with msm_kmeans_pack and hands it to the _packed entry points; the plain entry points build it per
 * call in a buffer of the context.  d_image == NULL or a shape outside the filter's range: same as the plain
 * call.  msm_kmeans_image_bytes gives the size (0 for d > 10); the buffer must be 16-byte aligned.
 * msm_kmeans_filter_scanned reports (and optionally clears) the number of frames that took the exhaustive
 * scan since the context was created: diagnostics only. */
msm_status msm_kmeans_image_bytes(int64_t n, int d, size_t* out_bytes);
msm_status msm_kmeans_pack(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int d, int64_t ld,
                           const double* d_mean, const double* d_std, void* d_image);
msm_status msm_kmeans_assign_packed(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int d,
                                    int64_t ld, const double* d_centers, int k, const double* d_mean,
                                    const double* d_std, const void* d_image, int32_t* d_labels,
                                    double* d_mindist);
msm_status msm_kmeans_accumulate_packed(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int d,
                                        int64_t ld, const double* d_centers, int k, const double* d_mean,
                                        const double* d_std, const void* d_image, const double* d_state,
                                        int64_t* d_sums, int64_t* d_counts);
/* Incremental form of the accumulate pass: d_prev_labels int32 [n] holds the centre each frame is booked under
 * (-1: none yet) and is updated in place; a frame whose centre did not change is left alone, one that moved is
 * subtracted from its old centre's sums / count and added to the new one's.  d_sums / d_counts must therefore
 * PERSIST between calls (msm_kmeans_update with clear = 0).  The sums are 64-bit fixed-point integers, so they hold
 * the bits of a full re-accumulation; after the first passes of a Lloyd run only a few per cent of the frames move.
 * d_image may be NULL. */
msm_status msm_kmeans_accumulate_delta(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int d,
                                       int64_t ld, const double* d_centers, int k, const double* d_mean,
                                       const double* d_std, const void* d_image, const double* d_state,
                                       int32_t* d_prev_labels, int64_t* d_sums, int64_t* d_counts);
/* One whole Lloyd iteration: the accumulate pass above (incremental when d_prev_labels is given, which then needs
 * persistent d_sums / d_counts) followed by msm_kmeans_update(clear = 0) on d_centers / d_state -- in ONE launch where
 * the filter kernel runs: the workgroup that finishes last closes the iteration, with the bits of the separate update
 * kernel.  For a single shard; with several shards the sums are reduced between the two halves
 * (msm_kmeans_accumulate_delta, msm_allreduce_i64_from, msm_kmeans_update).
 * Reference: one iteration of deeptime KMeans.fit / sklearn KMeans behind cluster_microstates
 * (S/markov_state_model/clustering.py:322-361) and _KMeansDiscretizer.fit (S/analysis/discretize.py:458-469). */
msm_status msm_kmeans_lloyd_pass(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int d, int64_t ld,
                                 double* d_centers, int k, const double* d_mean, const double* d_std,
                                 const void* d_image, double* d_state, int32_t* d_prev_labels, int64_t* d_sums,
                                 int64_t* d_counts);
msm_status msm_kmeans_filter_scanned(msm_ctx* ctx, uint64_t* h_out, int reset);
/* Hardware rule the filter's certificate rests on (kmeans_filter.h, step 1): n_tiles independent
 * v_mfma_f32_16x16x32_bf16 instructions, D = A B + C with A [16][32], B [32][16] bf16 (row major, raw 16-bit
 * patterns) and C, D [16][16] f32, all HOST arrays.  tests/test_gpu_mfma_rule.py checks the accumulation
 * bound, the irrelevance of the slot order and the treatment of small terms on every box the suite runs on. */
msm_status msm_mfma_bf16_probe(msm_ctx* ctx, const uint16_t* h_a, const uint16_t* h_b, const float* h_c,
                               float* h_d, int n_tiles);

/* d_out[0] = sum of d_v[0..n) with a fixed-order two-level reduction (inertia =
 * sum of msm_kmeans_assign's d_mindist; clustering.py:391-392). */
msm_status msm_sum_f64(msm_ctx* ctx, const double* d_v, int64_t n, double* d_out);

/* ------------------------------------------------------------------ */
/* MSM estimation: transition matrix, stationary distribution, ITS      */
/* ------------------------------------------------------------------ */

/* Count matrix -> transition matrix.  d_counts is int64 [k*k] (counts_are_f64 = 0)
 * or float64 [k*k].  d_rowsum f64 [k] receives the row sums of the counts.
 *  mode 0: T = C / rowsum, all-zero rows stay zero, full k x k
 *          (_normalise_counts, S/analysis/discretize.py:678-682); d_diag_mass (may be
 *          NULL) = trace(T)/k (discretize.py:1059-1062).
 *  mode 1: ensure_connected_counts + MaximumLikelihoodMSM(reversible=False)
 *          (S/utils/msm_utils.py:129-167, S/markov_state_model/_estimation.py:158-188):
 *          active = states with rowsum + colsum > epsilon (ascending), C_active + alpha on
 *          every cell, T_active = row-normalised.  T_active is written PACKED into the
 *          top-left n_active x n_active corner of d_T with row stride k; d_active int32 [k]
 *          lists the active states, d_inv_map int32 [k] maps state -> packed index or -1,
 *          d_n_active int32 [1]. */
msm_status msm_transition_matrix(msm_ctx* ctx, const void* d_counts, int counts_are_f64, int k, int mode,
                                 double alpha, double epsilon, double* d_T, int32_t* d_active,
                                 int32_t* d_inv_map, int32_t* d_n_active, double* d_rowsum,
                                 double* d_diag_mass);

/* T_full = identity with the active block of a mode-1 result embedded, pi_full = 0
 * outside the active set (_estimation.py:174-181).  d_pi_active / d_pi_full may be NULL. */
msm_status msm_embed_full(msm_ctx* ctx, const double* d_T_active, const double* d_pi_active,
                          const int32_t* d_inv_map, int k, double* d_T_full, double* d_pi_full);

/* Leading spectrum of `batch` row-stochastic matrices (matrix b at d_T + b*t_stride, row
 * stride ld, order d_n[b] or n_max when d_n is NULL) by subspace iteration on T' with
 * Rayleigh-Ritz (one workgroup per matrix; the lag scan batches its lags).
 * Replaces deeptime stationary_distribution + eigenvalues(T, k) as used by
 * _finalize_transition_and_stationary and _summarize_its_stats
 * (S/markov_state_model/_its.py:543-604), and utils.safe_timescales (utils.py:17-57).
 *   p        subspace size (<= 32; use n_its + 1 + guard vectors)
 *   n_iter   iterations this call; init != 0 (re)starts from a seeded basis, init == 0
 *            continues from the basis left in d_workspace by the previous call
 *   d_ritz   f64 [batch][128] = {re[32] | im[32] | previous call's re[32] | im[32]},
 *            Ritz values sorted by descending magnitude (the caller keeps the buffer between calls)
 *   d_change f64 [batch]: convergence measure of the top n_watch Ritz values -- the residual
 *            ||T'x - theta x|| / ||x|| of every Ritz pair (complex pairs in real arithmetic through the
 *            real invariant plane; for them the smaller of the residual and the relative change of
 *            the value since the previous call counts) -- the caller relaunches with init = 0 until
 *            it is small
 *   d_pi     f64 [batch][pi_stride] stationary distribution (sum 1), or NULL
 *   n_its>0: d_its_eig / d_its_ts f64 [batch][n_its]: the top (n_its+1) values re-sorted by
 *            descending real part, first dropped, |real part| clipped to [1e-12, 1-1e-12],
 *            t = -max(1, lag)/ln(.), lag = d_lags[b]; NaN padding when fewer exist
 *   d_status int32 [batch]: 0, or the hqr failure index
 *   freeze_tol > 0 with init == 0: matrices whose d_change (left by the previous call) is already
 *            <= freeze_tol are skipped and keep all their outputs (uneven batches: lag scans,
 *            posterior samples); 0 iterates every matrix
 *   d_vecs   f64 [batch][n_vecs][n_max] or NULL: left eigenvectors (x'T = theta x') of the Ritz values
 *            in descending magnitude, unit 2-norm, largest-magnitude component positive; NaN rows for
 *            complex pairs.
 *            For a reversible T the right eigenvectors are x / pi (PCCA+ input)
 * d_workspace must hold msm_spectrum_workspace_bytes(n_max, p, batch) bytes. */
size_t msm_spectrum_workspace_bytes(int n_max, int p, int batch);
msm_status msm_spectrum(msm_ctx* ctx, const double* d_T, int64_t t_stride, int ld, const int32_t* d_n,
                        int n_max, int batch, int p, int n_iter, int init, uint64_t seed, int n_watch,
                        void* d_workspace, double* d_ritz, double* d_pi, int64_t pi_stride,
                        double* d_change, int32_t* d_status, int n_its, const double* d_lags,
                        double* d_its_eig, double* d_its_ts, double freeze_tol, double* d_vecs, int n_vecs);
/* d_out <- T^(2^n_squarings) for the same batch layout (fp64 matrix cores; entries outside the d_n[b] x d_n[b] block
 * zero).  d_scratch: a second buffer of the batch's size (may be NULL for one squaring).  msm_spectrum_powered is
 * msm_spectrum with the ITERATIONS run on such a power d_T_power (same invariant subspaces, the convergence ratio
 * raised to that power: a quarter of the iterations with T^4); the Rayleigh-Ritz values, residuals, pi, implied
 * timescales and vectors are those of d_T itself.  Same reference operators as msm_spectrum. */
msm_status msm_matrix_power(msm_ctx* ctx, const double* d_T, int64_t t_stride, int ld, const int32_t* d_n, int n_max,
                            int batch, int n_squarings, double* d_scratch, double* d_out);
msm_status msm_spectrum_powered(msm_ctx* ctx, const double* d_T, const double* d_T_power, int64_t t_stride, int ld,
                                const int32_t* d_n, int n_max, int batch, int p, int n_iter, int init, uint64_t seed,
                                int n_watch, void* d_workspace, double* d_ritz, double* d_pi, int64_t pi_stride,
                                double* d_change, int32_t* d_status, int n_its, const double* d_lags,
                                double* d_its_eig, double* d_its_ts, double freeze_tol, double* d_vecs, int n_vecs);

/* Reversible maximum-likelihood estimate: deeptime's MaximumLikelihoodMSM(reversible=True) as
 * called by _fit_msm_deeptime (S/markov_state_model/_msm_utils.py:210-262) and the lag selector
 * (S/markov_state_model/ck_its_selector.py:395-401), restated from the published fixed-point
 * iteration x_ij <- (c_ij + c_ji) / (c_i/x_i + c_j/x_j) on the row sums x_i, normalised every step,
 * until max_i |x_i - x_i'| / ((x_i + x_i')/2) <= maxerr (deeptime: 1e-8) or maxiter (1e6).
 * d_counts f64 [n, ld] must be a connected count matrix (e.g. ensure_connected_counts' result);
 * d_T f64 [n, ldt] row-stochastic with pi_i T_ij = pi_j T_ji, d_pi f64 [n] (may be NULL).
 * The convergence test polls the host (every 32, 64, ... 1024 iterations): the call synchronises the
 * stream and cannot be captured.  *h_iterations / *h_err (host, may be NULL) report what was run. */
msm_status msm_reversible_mle(msm_ctx* ctx, const double* d_counts, int n, int ld, double maxerr, int maxiter,
                              double* d_T, int ldt, double* d_pi, int* h_iterations, double* h_err);

/* Posterior samples of a mode-1 estimate for the implied-timescale confidence intervals
 * (ITSMixin._its_compute_for_single_lag, S/markov_state_model/_its.py:272-357, which asks
 * deeptime's BayesianMSM for n_samples matrices; that sampler's stream is not reproducible, so
 * this draws from the closed-form posterior of the estimator msm_transition_matrix implements):
 * row i of every sample ~ Dirichlet(C[active[i], active[:]] + alpha), rows independent.
 * d_counts / d_active / d_n_active as for msm_transition_matrix mode 1.  Sample s (numbered
 * first_sample + s) is written packed (n_active x n_active, row stride ld) at d_T + s*t_stride:
 * the layout msm_spectrum's batch takes.  Variates are Philox4x32-10 keyed by `seed` with counter
 * (column, row, sample number, attempt): a cell does not depend on the batch it is drawn in. */
msm_status msm_sample_transition_matrices(msm_ctx* ctx, const void* d_counts, int counts_are_f64, int k,
                                          const int32_t* d_active, const int32_t* d_n_active, double alpha,
                                          uint64_t seed, int first_sample, int n_samples, double* d_T,
                                          int64_t t_stride, int ld);

/* One Philox4x32-10 block (known-answer test of the generator): d_out uint32 [4]. */
msm_status msm_philox4x32(msm_ctx* ctx, uint64_t key, const uint32_t counter[4], uint32_t* d_out);

/* ---- Chapman-Kolmogorov test ---------------------------------------------
 * msm_gemm_f64: C (m x n) = A (m x k) . B (k x n), row-major fp64 on the matrix cores; every
 * output element is the ascending-k FMA chain from +0 (bit-reproducible).  C must not alias
 * A or B.  Building block of the matrix powers below.
 *
 * msm_ck_test replaces the numerics of ck_error / _multinomial_rms_se / decide_ck
 * (S/validation/ck_rule.py:36-63, 71-117) and of _ck_on_trajs (S/markov_state_model/
 * ck_runner.py:155-176): for every factor f = h_factors[i]
 *   d_mse[i]   = mean((T1^f - Tk[i])^2)                         (RMS error = sqrt of it)
 *   d_noise[i] = sqrt(mean_i(sum_j p_ij (1 - p_ij) / N_i / n)),  p = Tk[i], N = row counts of
 *                the lag-f*tau count matrix (N_i <= 0 or non-finite -> 1); skipped when
 *                d_rowcounts / d_noise are NULL
 * d_T1 f64 [n, ld1]; d_Tk f64 [n_factors][tk_stride] with row stride ldk; d_rowcounts f64
 * [n_factors][rc_stride].  T1^f is formed by f-1 right-multiplications with T1. */
msm_status msm_gemm_f64(msm_ctx* ctx, int m, int n, int k, const double* d_A, int64_t lda,
                        const double* d_B, int64_t ldb, double* d_C, int64_t ldc);
msm_status msm_ck_test(msm_ctx* ctx, const double* d_T1, int64_t ld1, const double* d_Tk,
                       int64_t tk_stride, int64_t ldk, int n, const int32_t* h_factors, int n_factors,
                       const double* d_rowcounts, int64_t rc_stride, double* d_mse, double* d_noise);

/* d_out f64 [3] = { sum |P - Q|, sum |Q|, sum (P - Q)^2 } over the n x m entries (fixed order).
 * The relative L1 error of the lag selector's CK test is out[0] / out[1]
 * (_compute_ck_error, S/markov_state_model/ck_its_selector.py:211-226). */
msm_status msm_diff_norms(msm_ctx* ctx, const double* d_P, int64_t ldp, const double* d_Q, int64_t ldq, int n, int m,
                          double* d_out);

/* Exact order statistics: h_out[q] = the h_ranks[q]-th smallest (0-based) of the n strided samples, by radix
 * selection on the order-preserving integer image of the doubles (six 12-bit histogram passes per rank; no
 * sort).  Feeds the quantile rules of the free-energy grids (mquantiles / iqr in generate_2d_fes,
 * S/markov_state_model/free_energy.py:494-590) and medians.  Samples must not be NaN.  Polls the host
 * between passes: synchronises the stream, cannot be captured. */
msm_status msm_order_statistics(msm_ctx* ctx, const double* d_x, int64_t stride, int64_t n, const int64_t* h_ranks,
                                int n_ranks, double* h_out);

/* ---- free-energy surfaces (S/analysis/fes.py) -----------------------------
 * msm_weighted_stats: d_out6 = {sum w, sum w^2, weighted mean, weighted variance (around that
 *   mean, / sum w), min, max} of the strided coordinate x[i * stride]; d_w NULL = unit weights
 *   (np.average twice, _compute_bandwidth :142-173; data range of np.histogram2d).
 * msm_hist2d: np.histogram2d(x, y, bins=[nx, ny], weights=w) on the given edges (:241-249):
 *   bin = searchsorted(edges, v, 'right') - 1, last edge inclusive, values outside or NaN
 *   dropped.  d_hist f64 [nx, ny].  Weighted sums run in 2^e fixed point (e from n * w_absmax,
 *   w_absmax >= max |w|): independent of scheduling, exact to 2^-e per frame.
 * msm_smooth_sparse_bins: bins < min_count are raised to max(mean of the 8 neighbours with
 *   replicated edges, min_count) when that mean is > 0 (:270-292); d_out != d_hist;
 *   *d_n_smoothed = number of changed bins.  msm_scale_to_total rescales to a given sum
 *   (the reference restores the raw total only when something was smoothed, :254-258).
 * msm_fes_finalize: F = -kT ln(h / sum h) - min (:570-599).  *d_status: 0 ok, bit 0 non-finite
 *   entry, bit 1 total <= 0, bit 2 entry <= 0 (the reference raises for each).
 * msm_kde2d: density[i][j] = 1/(2 pi bw_x bw_y) sum_k w_k w_scale exp(-((xc_i - x_k)/bw_x)^2/2)
 *   exp(-((yc_j - y_k)/bw_y)^2/2) on the matrix cores (:176-238); d_w NULL = all weights
 *   w_scale.  periodic bit 0 / bit 1: the x / y difference is wrapped to [-pi, pi) first (wrapped
 *   Gaussian on the torus: periodic_kde_2d, S/markov_state_model/free_energy.py:321-360). */
msm_status msm_weighted_stats(msm_ctx* ctx, const double* d_x, int64_t stride, int64_t n,
                              const double* d_w, double* d_out6);
msm_status msm_hist2d(msm_ctx* ctx, const double* d_x, int64_t sx, const double* d_y, int64_t sy,
                      int64_t n, const double* d_w, double w_absmax, const double* d_xedges, int nx,
                      const double* d_yedges, int ny, double* d_hist);
msm_status msm_smooth_sparse_bins(msm_ctx* ctx, const double* d_hist, int nx, int ny,
                                  double min_count, double* d_out, int32_t* d_n_smoothed);
msm_status msm_scale_to_total(msm_ctx* ctx, double* d_v, int n, double total);
msm_status msm_fes_finalize(msm_ctx* ctx, const double* d_hist, int n_cells, double kT,
                            double* d_F, int32_t* d_status);
msm_status msm_kde2d(msm_ctx* ctx, const double* d_x, int64_t sx, const double* d_y, int64_t sy,
                     int64_t n, const double* d_w, double w_scale, const double* d_xcenters, int nx,
                     const double* d_ycenters, int ny, double bw_x, double bw_y, int periodic, double* d_density);

/* d_out[t] = np.clip(x[t], lo, hi) (mode 1) or ((x[t] - lo) % (hi - lo)) + lo with numpy's remainder (mode 2):
 * the sample preparation of generate_2d_fes (S/markov_state_model/free_energy.py:494-556). */
msm_status msm_clip_or_wrap(msm_ctx* ctx, const double* d_x, int64_t stride, int64_t n, double lo, double hi, int mode,
                            double* d_out);

/* d_out[t] = d_table[d_idx[t]], 0 for an index outside [0, m): the frame weights pi[state] of the MSM-reweighted
 * free-energy surface (FESCalculator.calculate_fes, S/markov_state_model/free_energy.py:1011). */
msm_status msm_gather_f64(msm_ctx* ctx, const double* d_table, int m, const int32_t* d_idx, int64_t n, double* d_out);

/* ---- dense solves on T: committors, reactive flux, lumping, MFPT ---------------------------
 * msm_solve_f64: A X = B by Gaussian elimination with partial pivoting (first maximal pivot),
 *   one workgroup; d_A [n, lda] is overwritten by the factors, d_B [n, ldb] by X.
 *   *d_info = 0, or j + 1 when column j has no non-zero pivot (singular).
 * msm_reactive_flux replaces the numerics TPTMixin gets from deeptime (S/markov_state_model/
 *   _tpt.py:39-160, 255-347): d_role[i] = 0 intermediate, 1 source A, 2 sink B;
 *   q+ : (T - I) q = 0 on intermediates, q = 0 on A, 1 on B;   q- : the same for the time-reversed
 *   chain pi_j T_ji / pi_i with 1 on A, 0 on B;   gross f_ij = pi_i q-_i T_ij q+_j (i != j);
 *   net = max(0, f_ij - f_ji);   d_totals = {F = sum_{i in A, j not in A} f_ij, Z = sum_i pi_i q-_i,
 *   rate F / Z, mfpt Z / F}.  d_info int32 [2] (forward, backward solve).  d_gross / d_net /
 *   d_totals may be NULL together (committors only).
 * msm_lump_macro: lump_micro_to_macro_T + compute_macro_populations (S/markov_state_model/
 *   _msm_utils.py:103-135): F_AB = sum_{i in A, j in B} pi_i T_ij, T_macro = F / rowsum (zero rows
 *   stay zero), pi_macro[A] = sum_{i in A} pi_i renormalised.
 * msm_macro_mfpt: compute_macro_mfpt (:138-160): for every target j solve (I - Q_j) t = 1 with
 *   row / column j removed; d_mfpt f64 [n, n] (column j = times into j, diagonal 0); d_info [n]. */
msm_status msm_solve_f64(msm_ctx* ctx, int n, int nrhs, double* d_A, int64_t lda, double* d_B,
                         int64_t ldb, int32_t* d_info);
msm_status msm_reactive_flux(msm_ctx* ctx, const double* d_T, int64_t ldt, const double* d_pi,
                             const int32_t* d_role, int n, double* d_qplus, double* d_qminus,
                             double* d_gross, double* d_net, double* d_totals, int32_t* d_info);
msm_status msm_lump_macro(msm_ctx* ctx, const double* d_T, int64_t ldt, const double* d_pi,
                          const int32_t* d_macro, int n, int n_macro, double* d_T_macro,
                          double* d_pi_macro);
msm_status msm_macro_mfpt(msm_ctx* ctx, const double* d_T, int64_t ldt, int n, double* d_mfpt,
                          int32_t* d_info);

/* ---- silhouette score (n_states = "auto") ---------------------------------------------------
 * Replaces sklearn.metrics.silhouette_score in _auto_select_n_states (S/markov_state_model/
 * clustering.py:156-233).  d_x f64 [n, ld]: the points SORTED BY CLUSTER, cluster c = rows
 * [h_offsets[c], h_offsets[c+1]); 2 <= k <= 32.  d_samples f64 [n]: s_i = (b_i - a_i) / max(a_i,
 * b_i) (0 for singleton clusters); *d_score = mean s_i. */
msm_status msm_silhouette(msm_ctx* ctx, const double* d_x, int64_t n, int d, int64_t ld,
                          const int64_t* h_offsets, int k, double* d_samples, double* d_score);

/* ---- regular-grid microstates (cluster_mode = "grid") -----------------------------------------
 * Replaces _GridDiscretizer._compute_indices / transform (S/analysis/discretize.py:552-583).
 * msm_grid_cells: d_flat[t] = sum_f idx_f * bins^(F-1-f), idx_f = clip(np.digitize(x[t][f],
 *   edges_f) - 1, 0, bins - 1); d_edges f64 [F][bins + 1] increasing; NaN -> last bin (as numpy).
 * msm_first_occurrence: d_first[c] = smallest frame index t with d_flat[t] == c, or -1 (the
 *   reference numbers states by order of first appearance).
 * msm_relabel: d_labels[t] = d_map[d_flat[t]]. */
msm_status msm_grid_cells(msm_ctx* ctx, const void* d_x, msm_dtype dtype, int64_t n, int F, int64_t ld,
                          const double* d_edges, int bins, int32_t* d_flat);
msm_status msm_first_occurrence(msm_ctx* ctx, const int32_t* d_flat, int64_t n, int n_cells,
                                int64_t* d_first);
msm_status msm_relabel(msm_ctx* ctx, const int32_t* d_flat, int64_t n, const int32_t* d_map, int n_cells,
                       int32_t* d_labels);

/* ------------------------------------------------------------------ */
/* exchange steps of the sharded path (RCCL over xGMI)                  */
/* ------------------------------------------------------------------ */

/* One process per GPU, one trajectory shard per process (the reference has no
 * multi-device code: lag pairs never cross a shard -- S/analysis/discretize.py:625-632 segments,
 * S/markov_state_model/_features.py:216-227 per-trajectory frame dropping -- so only the SMALL
 * per-shard results are summed).  The collectives run on the context's stream, in place, on device
 * buffers; librccl is dlopen'ed by the first msm_comm_unique_id / msm_comm_init.
 *
 *   msm_comm_unique_id     rank 0 draws the 128-byte RCCL id; the caller carries it to the other ranks
 *                          (file, environment, any store: pmarlo_amd/dist.py uses a file)
 *   msm_comm_init          collective over `world` processes (ncclCommInitRank)
 *   msm_allreduce_i64      exact integer sum: k-means member sums / counts, transition counts, the
 *                          batched L x k x k lag-scan counts (one call)
 *   msm_allreduce_f64      fp64 sum in RANK ORDER (all-gather + fixed-order add): moment blocks; the same
 *                          bits on every rank and for every algorithm RCCL may choose
 *   msm_allreduce_min_f64  / _max_f64: the fixed-point scale of the Lloyd sums; timing maxima
 *   msm_broadcast          bytes from `root`: the shared shift vector, the initial centres
 *   msm_comm_info          rank, world and the number of collectives issued so far
 *   msm_rcp_f64            dst[i] = 1 / src[i] on the stream (2^-e after the MIN of 2^e: exact) */
typedef struct msm_comm msm_comm;
#define MSM_COMM_ID_BYTES 128
msm_status msm_comm_unique_id(void* out_id, size_t bytes);
msm_status msm_comm_init(msm_ctx* ctx, int rank, int world, const void* id, size_t id_bytes, msm_comm** out);
void msm_comm_destroy(msm_comm* comm);
msm_status msm_comm_info(msm_comm* comm, int* rank, int* world, uint64_t* n_collectives);
msm_status msm_allreduce_i64(msm_comm* comm, int64_t* d_buf, size_t count);
/* the same sum out of place: d_dst = sum over ranks of d_src (the persistent local member sums of the Lloyd passes
 * are reduced straight into the exchange buffer, no copy in front of the collective) */
msm_status msm_allreduce_i64_from(msm_comm* comm, const int64_t* d_src, int64_t* d_dst, size_t count);
msm_status msm_allreduce_f64(msm_comm* comm, double* d_buf, size_t count);
msm_status msm_allreduce_min_f64(msm_comm* comm, double* d_buf, size_t count);
msm_status msm_allreduce_max_f64(msm_comm* comm, double* d_buf, size_t count);
msm_status msm_broadcast(msm_comm* comm, void* d_buf, size_t bytes, int root);
msm_status msm_rcp_f64(msm_ctx* ctx, const double* d_src, size_t n, double* d_dst);

#ifdef __cplusplus
}
#endif
#endif /* MSMHIP_H */
