// brov2_kernels.h -- host-side launch entry points of the HIP kernels (internal; the public
// boundary is include/brov2.h).  All pointers are device pointers; all launches are
// asynchronous on `st`.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "brov2_device.h"
#include "brov2_fast.h"

namespace brov {

// done / seq: when done != nullptr every row's thread stores seq to done[row] (system scope, release) after its results
hipError_t launch_rhs(hipStream_t st, const DevParams& p, int model, int64_t B, const double* x, const double* u,
                      double* lag, double* xd, unsigned long long* done = nullptr, unsigned long long seq = 0);
hipError_t launch_thruster_forces(hipStream_t st, const DevParams& p, int64_t B, const double* u, double* lag, double* tau,
                                  unsigned long long* done = nullptr, unsigned long long seq = 0);
hipError_t launch_rollout(hipStream_t st, const FastParams* d_fp, int model, int integ, int lag_mode, int layout, int64_t B,
                          int64_t T, double dt, const double* x0, const double* U, double* lag, double* traj,
                          int64_t stride, double* xT, int btu_staging);
hipError_t launch_window_endpoint(hipStream_t st, const FastParams* d_fp, int model, int integ, int64_t N, int64_t H, double dt,
                                  const double* X, const double* U, int carry_lag, const double* d_phi9,
                                  double* d_resp, double* d_start, double* d_se, double* d_total);
hipError_t launch_fill_controls(hipStream_t st, int layout, int dist, int64_t B, int64_t T, int nu, uint64_t seed,
                                int64_t b0, int64_t T_total, const double* scale8, double* U);

int probe_xcd_round_robin(hipStream_t st);

// ---- EDMDc -----------------------------------------------------------------------------
// Lifted row layout (device native, "Z rows"): [rbf_0 .. rbf_{kp-1} | x_0..x_{n-1} u_0..u_{r-1} 0..]
// with kp = k rounded up to 16 and the tail block (n + r) rounded up to 16; width = kp + tailp.
struct EdmdcShape {
    int n, r, k;
    int kp;      // k padded to a multiple of 16
    int tailp;   // (n + r) padded to a multiple of 16
    int width;   // kp + tailp  (doubles per lifted row)
    int d, p;    // n + k, n + k + r
    int xplus;   // 1: the padding of the tail is wide enough for x_{t+1} (n + r + n <= tailp); lift_rows writes it there and the
                 //    Gram takes the x part of Y from the row's own last tile instead of from a 33rd Y tile of the next row
};
inline EdmdcShape edmdc_shape(int n, int r, int k) {
    EdmdcShape s;
    s.n = n; s.r = r; s.k = k;
    s.kp = (k + 15) / 16 * 16;
    s.tailp = (n + r + 15) / 16 * 16;
    s.width = s.kp + s.tailp;
    s.d = n + k; s.p = n + k + r;
    s.xplus = (n + r + n <= s.tailp) ? 1 : 0;
    return s;
}
// Reference-order lift Z[N][n+k] = [x, rbf]  (edmdc_lift C entry point)
hipError_t launch_lift_ref(hipStream_t st, int64_t N, int n, int k, double gamma, const double* X, const double* C, double* Z);
// multistep_rmse by linearity: se[w] = |X[w + H] - (RHt^T phi(x_w) + sum_t Gt[t]^T u_{w+t})|^2 for w < nw; RHt [n + k][n], Gt [H][r][n]; xhat [nw][n] optional
hipError_t launch_linear_windows(hipStream_t st, int64_t nw, int n, int r, int k, int64_t H, double gamma, const double* X, const double* U,
                                 const double* C, const double* RHt, const double* Gt, double* se, double* xhat);
// Device-native lifted rows for `rows` consecutive state rows starting at global row `row0`
// (bag structure: state row index = b * xs + t, t in [0, L]; input row = b * us + t, t < L).
// Rows >= total_rows (and gap rows t > L) are written as zeros with weight 0.
// pairflag != nullptr: ragged bags -- X / U hold the bags' rows one after the other (U row-aligned with X), pairflag[g] = 1 when rows
// (g, g + 1) belong to one bag (launch_bag_pairflags); pass xs > total_rows and L = xs - 1.
hipError_t launch_lift_rows_total(hipStream_t st, const EdmdcShape& s, double gamma, const double* C,
                                  int64_t row0, int64_t rows, int64_t total_rows, int64_t L, int64_t xs, int64_t us,
                                  const double* X, const double* U, double* Zrows, double* wrow, const unsigned char* pairflag = nullptr);
// pairflag[g] = 1 for every row g < total_rows that has a successor in its own bag: bag b = rows [offsets[b], offsets[b + 1])
hipError_t launch_bag_pairflags(hipStream_t st, int64_t nbags, const int64_t* d_offsets, int64_t total_rows, unsigned char* pairflag);
// Gram task table (device copy owned by the ctx) and sizes.
// partial: [ntasks][nslab][24 tiles][64 lanes][4] doubles.
// mode 0: tasks of G^T[G|Y]; mode 1: tasks of W^T Y (edmdc_pinv_apply)
size_t gram_partial_doubles(const EdmdcShape& s, int mode, int* ntasks_out, int* nslab_out);
size_t gram_task_bytes(const EdmdcShape& s, int mode);
hipError_t upload_gram_tasks(hipStream_t st, const EdmdcShape& s, int mode, void* d_tasks, size_t cap_bytes, int* ntasks);
// Gram of one chunk: pairs (row, row+1) for row in [0, npairs) of Zrows (4*ceil(npairs/4)+1 rows lifted),
// weight wrow[row]; accumulated into `partial` in place when `accumulate`.
// Arows: rows the A operand is cut from (== Zrows for the Gram; the rows of W = G P^T for edmdc_pinv_apply)
hipError_t launch_gram_chunk_tasks(hipStream_t st, const EdmdcShape& s, int ntasks, const void* d_tasks, int64_t npairs,
                                   const double* Arows, const double* Zrows, const double* wrow, double* partial, int accumulate);
// Wrows[rows][width] = Zrows . PdT  (PdT [width][width] = P^T in device feature order)
// (PdT needs 8 extra rows of allocation: the tuned kernel prefetches two K-steps past the last feature; simple = the round-2 kernel)
hipError_t launch_rows_times_pt(hipStream_t st, const EdmdcShape& s, int64_t rows, const double* Zrows, const double* PdT, double* Wrows,
                                int simple = 0);
void wrows_decomposition(const EdmdcShape& s, int* items_per_unit, int* tiles_wanted_per_unit);
int edmdc_dev_to_ref_feature(const EdmdcShape& s, int f);
// Sum partials over slabs (fixed order) and scatter into reference-order GtG [p][p], GtY [p][d].
hipError_t launch_gram_finish_tasks(hipStream_t st, const EdmdcShape& s, int ntasks, const void* d_tasks, const double* partial,
                                    int accumulate_out, double* GtG, double* GtY);

// ---- lifted propagation (propagate.hip) ---------------------------------------------------
struct PropShape {
    int n, r, k, d, p;
    int dpad;      // d padded to 48 (feature rows of Zt / columns of ABt)
    int ksteps;    // ceil(p / 4)
    int ppad;      // 4 * ksteps (rows of ABt)
    int64_t nw;    // windows (or simulated trajectories)
    int64_t nwp;   // nw padded to 128
    int zrows;     // rows of a Zt buffer: max(dpad, ppad) -- lifted state rows, then the r input rows of the current step
};
PropShape prop_shape(int n, int r, int k, int64_t nw);
hipError_t launch_lift_t(hipStream_t st, const PropShape& s, double gamma, int64_t xstride, const double* X, const double* C, double* Zt);
hipError_t launch_transpose(hipStream_t st, int64_t rows, int64_t cols, const double* src, int64_t lds_, double* dst, int64_t ldd);
hipError_t launch_set_input_rows(hipStream_t st, const PropShape& s, const double* U, int64_t ldu, double* Zt);
int64_t prop_window_blocks(const PropShape& s);
hipError_t launch_propagate(hipStream_t st, const PropShape& s, const double* ABt, const double* Zin, const double* Unext, int64_t ldu, double* Zout,
                            int64_t wb0 = 0, int64_t nwb = -1);
hipError_t launch_endpoint_se(hipStream_t st, const PropShape& s, int64_t xstride, const double* Xref, const double* Zt, double* se, double* xhat);
hipError_t launch_extract_state(hipStream_t st, const PropShape& s, int64_t T1, int64_t t, const double* Zt, double* Xp);
hipError_t launch_useq_t(hipStream_t st, const PropShape& s, int64_t T, const double* Us, double* Ust);
int window_scan_chunk();
hipError_t launch_sum(hipStream_t st, int64_t n, const double* v, double* out);

// ---- column statistics (colstats.hip) -------------------------------------------------------
// partial [colstats_blocks(N)][16]: per-block sums of x (squares = false) or of (x - d_shift)^2 (squares = true) per column
int colstats_blocks(int64_t N);
hipError_t launch_colstats(hipStream_t st, int64_t N, int n, const double* X, int64_t xstride, const double* d_shift, bool squares,
                           double* d_partial);

// copy by the shader cores, 16-byte words (both pointers 16-byte aligned, bytes a multiple of 16); src or dst may be pinned host memory
hipError_t launch_copy_bytes(hipStream_t st, void* dst, const void* src, size_t bytes);

// ---- k-means (kmeans.hip) ------------------------------------------------------------------
int kmeans_blocks(int64_t N, int n, int k, bool scalar_records);
int kmeans_epochs(int64_t N, int n, int k, bool scalar_records);
size_t kmeans_partial_words(int64_t N, int n, int k, bool scalar_records);      // u64 words of the E-step's partial tables
size_t kmeans_red_words(int n, int k);                                          // int64 words of the totals (+ 2 tail words)
bool kmeans_reads_through_perm(int n, int k, bool scalar_records);
hipError_t launch_kmeans_c2(hipStream_t st, int n, int k, const double* C, double* c2);
hipError_t launch_kmeans_range(hipStream_t st, int64_t N, int n, const double* X, int64_t xstride, const double* mean, unsigned long long* rng);
hipError_t launch_kmeans_scale(hipStream_t st, int n, const unsigned long long* rng, double* fix, double* prm);
// distance bounds of the sorted loop (kmeans.hip: kmeans_bounds_kernel): per-position bounds moved by the centres' shifts; the positions
// whose bounds fail go to `list` (tiles padded to whole waves with ~position), their number to nlist[0]
// nlist: [0] the list's entries (one-region form) | [KM_NL_TICKET] tickets drawn by the E-step | [KM_NL_FRONT], [KM_NL_BACK] entries of the two
// regions (two-region form; [0] stays 0) -- every counter in its own 64-byte line: atomics on one line are served one after the other
// (83 per microsecond, tools/attic/atomic_ticket_probe.hip), and 2 442 tiles reserving their piece of the list are 29 us of that
constexpr int KM_NL_TICKET = 16, KM_NL_FRONT = 32, KM_NL_BACK = 48;
struct KmBounds {
    float* ub = nullptr;            // [N] >= distance to the own centre
    float* lb = nullptr;            // [N] <= distance to every other centre
    const float* shiftc = nullptr;  // [k + kmeans_bounds_tail()] from launch_kmeans_average
    const float* mvd = nullptr;     // [k][4] from launch_kmeans_cdist: distances to the centres that moved most
    const float* rw2 = nullptr;     // [k] from launch_kmeans_cdist, or nullptr: squared radius beyond which a centre's samples make expensive passes
                                    //     (given: the list lies in two regions, the tiles with such samples in front; nlist[2] = their entries)
    int* list = nullptr;            // [kmeans_bounds_list_words(N)]
    int* nlist = nullptr;           // [64]: counters of the list and of the E-step's tickets, one per 64-byte line (kmeans.hip: KM_NL_*)
    double beta = -1.0;             // >= 0: the prefix of a pass that leaves bounds is cut at 2 (1 + beta) u (default KM_BND_BETA)
    bool use_list = false;          // the E-step visits the list only and its partials are CHANGES (launch_kmeans_mstep: delta)
};
size_t kmeans_bounds_list_words(int64_t N);
hipError_t launch_kmeans_bounds(hipStream_t st, int64_t N, int k, const int* labels, const KmBounds& b, const double* prm);
// E-step; Dc != nullptr selects the candidate-filtered form (Dc [k][k rounded up to 256] floats from launch_kmeans_cdist); prm [4], fix [32] from launch_kmeans_scale
hipError_t launch_kmeans_assign(hipStream_t st, int64_t N, int n, int k, const double* X, int64_t xstride, const double* mean,
                                const double* c2, int* labels, unsigned long long* partial, double* block_inertia, int* block_changed,
                                const float* Dc, const double* prm, const double* fix, float* d2out, bool scalar_records, const int* perm = nullptr,
                                const unsigned long long* Nk = nullptr, const float* Pf = nullptr, const KmBounds* bounds = nullptr);
// Dc [k][kp] (kp = k rounded up to 256); Nk (optional, kp <= 512): every row once more sorted, as keys (distance bits << 16 | centre index)
// Pf (optional, with Nk; n <= 13): the sorted rows once more as float pair records [k][kp / 2][32] for kmeans_assign_pk_kernel
struct KmMstepArgs;
hipError_t launch_kmeans_cdist(hipStream_t st, int n, int k, const double* c2, float* Dc, unsigned long long* Nk = nullptr, float* Pf = nullptr,
                               const float* shiftc = nullptr, float* mvd = nullptr, float* rw2 = nullptr, int pf_pairs = 0,
                               const KmMstepArgs* tail = nullptr);
int kmeans_bounds_tail();
int kmeans_lds_pf_pairs();          // pair records per row the LDS / DPP kernel's screening can reach (pf_pairs of launch_kmeans_cdist when it is Pf's only reader)
// E-step with packed-fp32 screening of the candidates (kmeans.hip, third form): labels, scores and member sums as the other kernels'
int kmeans_pk_blocks(int64_t N, int n, int k);
int kmeans_pk_epochs(int64_t N, int n, int k);
bool kmeans_pk_supported(int n, int k);
hipError_t launch_kmeans_assign_pk(hipStream_t st, int64_t N, int n, int k, const double* X, int64_t xstride, const double* mean, const double* c2,
                                   int* labels, unsigned long long* partial, double* block_inertia, int* block_changed, const double* prm,
                                   const double* fix, float* d2out, const int* perm, const unsigned long long* Nk, const float* Pf);
hipError_t launch_kmeans_average(hipStream_t st, int n, int k, const long long* red, const double* fix, const double* Cold, double* Cnew,
                                 double* c2, double* stats, double* prm, int mode, float* shiftc = nullptr, int* nlist = nullptr);
// the M-step as one launch (kmeans.hip: kmeans_mstep_kernel): phases 1 = sums, 2 = centres + tail (from `red`), 3 = both
struct KmMstepArgs {
    int nparts = 0, nblocks = 0, n = 0, k = 0;
    const unsigned long long* partial = nullptr;
    const double* block_inertia = nullptr;
    const int* block_changed = nullptr;
    long long* red = nullptr;
    long long* tot = nullptr;
    int delta = 0;
    int* nlist = nullptr;               // the counter of the bounds list (read when delta, zeroed for the next list)
    const double* fix = nullptr;
    const double* Cold = nullptr;
    double* Cnew = nullptr;
    double* Ct = nullptr;
    double* stats = nullptr;
    double* prm = nullptr;
    float* shiftc = nullptr;
    float* mvd = nullptr;
    double* scratch = nullptr;          // [kmeans_mstep_scratch_doubles(k)], zeroed once before the loop (its ticket word)
    double* hstats = nullptr;           // pinned, device-mapped [5] or nullptr: four statistics, then `seq`
    double seq = 0.0;
    bool tail_deferred = false;         // the one-block global part (sum of shifts, empty clusters, movers, the host's statistics) is not launched here:
                                        // the caller hands these arguments to the launch_kmeans_cdist that follows (`tail`), which runs it as one more block
};
size_t kmeans_mstep_scratch_doubles(int k);
hipError_t launch_kmeans_mstep(hipStream_t st, const KmMstepArgs& a, int phases);
hipError_t launch_kmeans_reloc_dist(hipStream_t st, int64_t N, int n, const double* X, int64_t xstride, const double* mean, const double* Cold,
                                    const int* labels, const int* perm, double* dist_row, int* lab_row);


// width of the label field in the sort keys of the Lloyd loop's sample order (sortperm.hip); kmeans.hip checks its own limit against it
constexpr int KM_SORT_LABEL_BITS = 10;
constexpr int KM_SORT_LABEL_MAX = 1 << KM_SORT_LABEL_BITS;
// sample order of the Lloyd loop (sortperm.hip): sort by (label, distance to the centre), gather rows / labels / permutation
size_t kmeans_sort_temp_bytes(int64_t N);
hipError_t launch_kmeans_resort(hipStream_t st, int64_t N, const int* labels_old, int* labels_new, const int* perm_old, int* perm_new,
                                const float* d2, unsigned* keys_in, unsigned* keys_out, unsigned* vals_in, unsigned* vals_out, void* temp,
                                size_t temp_bytes, float* d2_new = nullptr, const float* ub_old = nullptr, float* ub_new = nullptr,
                                const float* lb_old = nullptr, float* lb_new = nullptr);
hipError_t launch_kmeans_unpermute(hipStream_t st, int64_t N, const int* perm, const int* labels_sorted, int* labels_out);

int kmeanspp_chunks(int64_t N);
size_t kmeanspp_sum_doubles(int64_t N);
size_t kmeanspp_state_bytes();
hipError_t launch_kmeanspp(hipStream_t st, int64_t N, int n, int k, int L, const double* X, int64_t xstride, const double* mean,
                           long long first, const double* u, double* Xt, double* xsq, double* closest, double* S,
                           void* state, double* C, long long* indices, float* Xf, void* rowbuf = nullptr);
// scratch of the seeding's row level (a ball, the largest closest and the sum of closest per row of 16 samples): bytes, 256-aligned pieces
size_t kmeanspp_row_bytes(int64_t N, int n);

size_t kmeanspp_shard_doubles(int world);
hipError_t launch_kmeanspp_sharded(hipStream_t st, int64_t N, int n, int k, int L, const double* X, int64_t xstride, const double* mean,
                                   long long first, const double* u, double* Xt, double* xsq, double* closest, double* S,
                                   void* state, double* C, long long* indices, float* Xf, int world, int rank, long long row0,
                                   double* shard, int (*exch)(void*, void*, int64_t, int), void* user, int* comm_failed, void* rowbuf = nullptr);

}  // namespace brov
