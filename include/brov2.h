/*
 * brov2.h -- C ABI of libbrov2.so: MI355X (gfx950) batched BlueROV2 Fossen dynamics
 * rollouts and Koopman-EDMDc lift / Gram build.
 *
 * The reference (ViktorNfa/bluerov2_dynamics) is pure Python and has no FFI layer; its
 * boundary for this path is the Python object API that training scripts import.  Each
 * entry point below names the reference interface it stands behind (paths relative to the
 * reference checkout).  The Python classes in bluerov2_dynamics_amd/{fossen,Koopman}/ keep
 * the reference's names/signatures and call these functions through ctypes
 * (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - every array is IEEE fp64, C-contiguous; sizes are element counts, not bytes;
 *   - functions return BROV_OK (0) or a negative brov_status; brov_last_error(ctx) gives
 *     a message owned by the ctx.  Nothing aborts, NaN/inf propagate like in the reference;
 *   - a ctx is bound to one device, is not thread-safe, and launches on one HIP stream
 *     (brov_set_stream; default: the null stream);
 *   - "_dev" functions take DEVICE pointers and are asynchronous on the ctx stream;
 *     the others take HOST pointers, copy in/out and return when the result is on the host.
 *
 * Models (brov_model):
 *   BROV_THRUSTER_EULER  nx=12 [x y z phi theta psi u v w p q r], nu=8 thruster commands
 *                        in [-1,1], 8x3 thruster-lag state   (fossen/BlueROV2.py)
 *   BROV_WRENCH_EULER    nx=12, nu=6 body wrench, no lag       (fossen/BlueROV2_thrust.py)
 *   BROV_WRENCH_QUAT     nx=13 [x y z qw qx qy qz u v w p q r], nu=6; the quaternion is
 *                        re-normalised after every integrator step (fossen/BlueROV2_wrench.py,
 *                        training/train_tank_brov2_wrench_quat.py:258-263)
 */
#ifndef BROV2_H
#define BROV2_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BROV2_ABI_VERSION 1

#if defined(BROV2_BUILDING)
#define BROV_API __attribute__((visibility("default")))
#else
#define BROV_API
#endif

typedef struct brov_ctx brov_ctx;

typedef enum brov_status {
    BROV_OK = 0,
    BROV_ERR_ARG = -1,     /* bad argument (NULL, negative size, unknown enum) */
    BROV_ERR_HIP = -2,     /* a HIP runtime call failed (message has hipGetErrorString) */
    BROV_ERR_NOMEM = -3,   /* device or host allocation failed */
    BROV_ERR_NODEVICE = -4,/* no gfx950 device visible (no GPU at all, or a GPU of another architecture) */
    BROV_ERR_COMM = -5     /* RCCL missing or an RCCL call failed (brov_comm_last_error) */
} brov_status;

typedef enum brov_model {
    BROV_THRUSTER_EULER = 0, BROV_WRENCH_EULER = 1, BROV_WRENCH_QUAT = 2,
    /* learned double-integrator baseline of the comparison scripts (gains from brov_set_di_gains):
     * nx/nu = 12/8, 12/6, 13/6; rollouts and window errors only */
    BROV_DI_THRUSTER_EULER = 3, BROV_DI_WRENCH_EULER = 4, BROV_DI_WRENCH_QUAT = 5
} brov_model;
typedef enum brov_integrator { BROV_EULER = 0, BROV_RK4 = 1 } brov_integrator;
/* PER_CALL = the reference: the lag filters advance on every dynamics() call, i.e. 4x per RK4
 * step (training/train_tank_brov2_rk4.py:386-391 calling fossen/BlueROV2.py:258).
 * PER_STEP = lag advanced once per step, thrust frozen over the RK4 stages (not the reference). */
typedef enum brov_lag_mode { BROV_LAG_PER_CALL = 0, BROV_LAG_PER_STEP = 1 } brov_lag_mode;
/* BTU: U[B][T][nu], traj[B][T/stride+1][nx]  (what the reference's callers hold, row-major)
 * TUB: U[T][nu][B], traj[T/stride+1][nx][B]  (time-major struct-of-arrays, device native) */
/* TPB: as TUB with channel PAIRS interleaved -- U[T][ceil(nu/2)][B][2], traj[T/stride+1][ceil(nx/2)][B][2] (an odd
 *      channel count is padded with one unused element): 16-byte accesses per lane, the fastest rollout layout */
typedef enum brov_layout { BROV_LAYOUT_BTU = 0, BROV_LAYOUT_TUB = 1, BROV_LAYOUT_TPB = 2 } brov_layout;
typedef enum brov_dist { BROV_DIST_IID_UNIFORM = 0, BROV_DIST_AR1 = 1 } brov_dist;

/* Vehicle parameters; brov_default_params() fills the reference's values
 * (fossen/BlueROV2.py:79-140,172-232,245-257,476-480). Signs follow the reference's attributes. */
typedef struct brov_params {
    double rho, g, m, volume;
    double xb, yb, zb;          /* centre of buoyancy (CG at origin) */
    double Ix, Iy, Iz;
    double added_mass[6];       /* Xu_dot Yv_dot Zw_dot Kp_dot Mq_dot Nr_dot (negative) */
    double lin_damp[6];         /* Xu Yv Zw Kp Mq Nr (negative or -0.0) */
    double quad_damp[6];        /* Xu_abs ... Nr_abs (negative) */
    double current[3];          /* NED current speed, default 0 */
    double thr_r[8][3];         /* thruster positions (body) */
    double thr_dir[8][3];       /* thruster directions (body) */
    double thrust_poly[5];      /* F_cmd = c0 V + c1 V^3 + c2 V^5 + c3 V^7 + c4 V^9 */
    double lag_Ac[9], lag_Bc[3], lag_Cc[3]; /* continuous thruster lag (row-major) */
} brov_params;

/* ---- context ------------------------------------------------------------------------- */
BROV_API int brov_abi_version(void);
BROV_API int brov_create(int device_id, brov_ctx** out);
BROV_API void brov_destroy(brov_ctx* ctx);
BROV_API const char* brov_last_error(const brov_ctx* ctx);
/* 1 if a device whose hipDeviceProp_t.gcnArchName is `gcn_arch_name` can run this library ("gfx950[:features]"), else 0.
 * brov_create applies it to the selected device and returns BROV_ERR_NODEVICE otherwise.  Host only. */
BROV_API int brov_arch_is_supported(const char* gcn_arch_name);
BROV_API int brov_device_arch(const brov_ctx* ctx, char* buf, size_t cap);   /* gcnArchName of the ctx's device */
/* 1 if workgroups b and b+8 were seen to share an XCD when the ctx was created (the assumption behind the XCD-aware block
 * numbering of the Gram / propagation kernels; speed only), 0 if not, -1 if the probe could not run. */
BROV_API int brov_xcd_round_robin(const brov_ctx* ctx);
BROV_API int brov_set_stream(brov_ctx* ctx, void* hip_stream);     /* hipStream_t; NULL = null stream */
BROV_API int brov_sync(brov_ctx* ctx);                             /* hipStreamSynchronize(ctx stream) */
/* HIP-event timing of the kernels launched by the most recent call on this ctx
 * (sum over its launches, ms).  Enable first; reading waits for the stop event. */
BROV_API int brov_set_timing(brov_ctx* ctx, int enabled);
BROV_API int brov_last_kernel_ms(brov_ctx* ctx, float* ms);

/* ---- parameters  (replaces BlueROV2.__init__, fossen/BlueROV2.py:79-157) -------------- */
BROV_API void brov_default_params(brov_params* p);
BROV_API int brov_set_params(brov_ctx* ctx, const brov_params* p);
BROV_API int brov_get_params(const brov_ctx* ctx, brov_params* p);
/* Minv diagonal (6) and the 6x8 allocation matrix tau = T F  (fossen/BlueROV2.py:126,265-278).
 * Host only; p = NULL means the default parameters. */
BROV_API int brov_get_derived(const brov_params* p, double Minv6[6], double alloc6x8[48]);
/* ZOH discretisation of the thruster lag: replaces ThrusterLag._discretise
 * (fossen/BlueROV2.py:490-496 -> scipy.signal.cont2discrete(method="zoh")).  Host only. */
BROV_API int brov_discretise_lag(const brov_params* p, double dt, double Ad[9], double Bd[3]);

BROV_API int brov_model_nx(int model);
BROV_API int brov_model_nu(int model);

/* ---- device memory helpers (so that callers need no HIP/torch of their own) -----------
 * brov_free keeps blocks of up to 256 MB (1 GB in total per ctx) for the next brov_malloc of a similar size; brov_destroy releases them.
 * The copies are synchronous (done on return) and accept pageable host memory: between 64 KB and 16 MB they are staged through two
 * pinned blocks owned by the ctx instead of handing the caller's pages to the runtime. */
BROV_API int brov_malloc(brov_ctx* ctx, size_t bytes, void** dptr);
BROV_API int brov_free(brov_ctx* ctx, void* dptr);
BROV_API int brov_memcpy_h2d(brov_ctx* ctx, void* dst, const void* src, size_t bytes);
BROV_API int brov_memcpy_d2h(brov_ctx* ctx, void* dst, const void* src, size_t bytes);
BROV_API int brov_memset(brov_ctx* ctx, void* dst, int value, size_t bytes);
/* Free and total bytes of the ctx's device (hipMemGetInfo): lets a caller without HIP / torch decide whether an optional buffer
 * (edmdc_lift_cache) fits. */
BROV_API int brov_mem_info(brov_ctx* ctx, size_t* free_bytes, size_t* total_bytes);
/* A list of host arrays into ONE device buffer -- what KoopmanEDMDc.fit_multi's np.vstack of its trajectory list becomes
 * (Koopman/koopmanEDMDc.py:125,140-142).  Bag b = bag_rows[b] rows of `cols` contiguous doubles at bag_ptrs[b]; it is written to
 * d_dst + dst_rows[b] * cols.  The destinations must ascend without overlap (holes are allowed; small ones are zero-filled, larger ones
 * left untouched).  Packed through two pinned staging blocks by a few host threads while the previous block is in flight; the data is
 * in place when the call returns. */
BROV_API int brov_upload_bags(brov_ctx* ctx, int64_t nbags, const double* const* bag_ptrs, const int64_t* bag_rows,
                              const int64_t* dst_rows, int cols, double* d_dst);

/* ---- Fossen RHS / rollouts ------------------------------------------------------------ */
/* Batched dynamics(): xdot[b] = f(x[b], u[b]).  Replaces BlueROV2.dynamics
 * (fossen/BlueROV2.py:357-400, BlueROV2_thrust.py:235-282, BlueROV2_wrench.py:322-367).
 * lag_io [B][8][3] is advanced one sample in place (thruster model; the reference's side
 * effect on self.thruster_lags); NULL = zero lag state, not returned. */
BROV_API int brov_rhs(brov_ctx* ctx, int model, int64_t B, const double* x, const double* u, double dt,
             double* lag_io, double* xdot);
/* Replaces BlueROV2.compute_thruster_forces (fossen/BlueROV2.py:265-278): tau [B][6]. */
BROV_API int brov_thruster_forces(brov_ctx* ctx, int64_t B, const double* u, double dt, double* lag_io, double* tau);

/* simulate_physics over a batch: training/train_tank_brov2_full_comparison.py:453-466 (Euler),
 * training/train_tank_brov2_rk4.py:375-396 (RK4), ..._wrench_quat.py:249-266 (quaternion).
 * x0 [B][nx]; U and traj per `layout`; lag_io [B][8][3] in/out or NULL (zero start);
 * traj holds every traj_stride-th state including x0 (T/stride+1 states) or is NULL;
 * xT [B][nx] (final state) or NULL. */
BROV_API int brov_rollout(brov_ctx* ctx, int model, int integrator, int lag_mode, int layout,
                 int64_t B, int64_t T, double dt, const double* x0, const double* U,
                 double* lag_io, double* traj, int64_t traj_stride, double* xT);
/* Gains of the double-integrator models: dv = u K_lin, dw = u K_ang with K_lin, K_ang [nu][3] row-major, nu = 8 or 6
 * (estimate_di_gains + simulate_double_integrator, training/train_tank_brov2_full_comparison.py:510-573,
 * RK4 form ..._rk4.py:461-525, wrench / quaternion forms ..._wrench_comp.py:293-341, ..._wrench_quat.py:324-372). */
BROV_API int brov_set_di_gains(brov_ctx* ctx, int nu, const double* K_lin, const double* K_ang);
/* How BROV_LAYOUT_BTU rollouts (stride 1 or no trajectory) move data: 0 = auto (LDS-staged tiles
 * for the memory-bound cases -- Euler, wrench models -- and lane-per-row accesses for the
 * instruction-bound thruster RK4 kernel; measured in DESIGN.md), 1 = always LDS-staged, 2 = never. */
BROV_API int brov_set_btu_staging(brov_ctx* ctx, int mode);
/* Which kernel runs the thruster-model rollouts: 0 = the two-wave kernel (thrust half and body half of a step on two waves that share
 * a SIMD; default), 1 = the whole step in one lane (the kernel of the wrench / double-integrator models; the independent second
 * implementation of the parity tests).  Same trajectories to rounding. */
BROV_API int brov_set_rollout_variant(brov_ctx* ctx, int variant);
BROV_API int brov_rollout_dev(brov_ctx* ctx, int model, int integrator, int lag_mode, int layout,
                     int64_t B, int64_t T, double dt, const double* d_x0, const double* d_U,
                     double* d_lag_io, double* d_traj, int64_t traj_stride, double* d_xT);

/* multistep_rmse_endpoint_physics: training/train_tank_brov2_full_comparison.py:469-487,
 * ..._rk4.py:399-417, ..._wrench_comp.py, ..._wrench_quat.py:279-297.
 * X [N][nx], U [N][nu]; windows k = 0..N-H-1 start at X[k], apply U[k..k+H-1], and are scored
 * against X[k+H].  carry_lag=1 is the reference: ONE vehicle object serves all windows, so the
 * thruster-lag state left by window k-1 is the initial lag state of window k.
 * se_total = sum_k |x_end - X[k+H]|^2 ; per_window [N-H] optional (NULL).
 * rmse = sqrt(se_total / ((N-H) * nx)). */
BROV_API int brov_window_endpoint_se(brov_ctx* ctx, int model, int integrator, int64_t N, int64_t H, double dt,
                            const double* X, const double* U, int carry_lag,
                            double* se_total, double* per_window);
BROV_API int brov_window_endpoint_se_dev(brov_ctx* ctx, int model, int integrator, int64_t N, int64_t H, double dt,
                                const double* d_X, const double* d_U, int carry_lag,
                                double* d_se_total, double* d_per_window /* [N-H], required */);

/* Synthetic control sequences on device (benchmarks):
 * counter-based splitmix64 stream, value for (trajectory b0+b, step t, channel j) independent
 * of layout and of how trajectories are sharded.  scale[nu] multiplies each channel (NULL = 1). */
BROV_API int brov_fill_controls_dev(brov_ctx* ctx, int layout, int dist, int64_t B, int64_t T, int nu,
                           uint64_t seed, int64_t b0, int64_t T_total, const double* scale_host,
                           double* d_U);

/* ---- Koopman EDMDc --------------------------------------------------------------------- */
/* phi(x) = [x, exp(-gamma(|x|^2 + |c|^2 - 2 x.c))]: KoopmanEDMDc._lift / _rbf_mat
 * (Koopman/koopmanEDMDc.py:41-48,221-236).  X [N][n], C [k][n] -> Z [N][n+k]. */
BROV_API int edmdc_lift(brov_ctx* ctx, int64_t N, int n, int k, double gamma, const double* X, const double* C, double* Z);

/* Normal-equation blocks of fit / fit_multi (Koopman/koopmanEDMDc.py:89-97,129-147):
 *   G = [phi(x_t), u_t], Y = phi(x_{t+1});  GtG [p][p] = G^T G, GtY [p][d] = G^T Y,
 *   d = n + k, p = d + r, feature order exactly the reference's ([x, rbf..., u]).
 * Data: nbags trajectories ("bags"); bag b holds states X[b*x_bag_stride + t], t = 0..L,
 * and inputs U[b*u_bag_stride + t], t = 0..L-1, i.e. L pairs per bag and no pair crosses
 * a bag.  fit(X,U) is nbags=1, L=N-1.  accumulate=1 adds to the GtG/GtY passed in. */
BROV_API int edmdc_gram(brov_ctx* ctx, int n, int r, int k, double gamma, const double* C,
               int64_t nbags, int64_t L, int64_t x_bag_stride, int64_t u_bag_stride,
               const double* X, const double* U, int accumulate, double* GtG, double* GtY);
BROV_API int edmdc_gram_dev(brov_ctx* ctx, int n, int r, int k, double gamma, const double* d_C,
                   int64_t nbags, int64_t L, int64_t x_bag_stride, int64_t u_bag_stride,
                   const double* d_X, const double* d_U, int accumulate, double* d_GtG, double* d_GtY);
/* The same normal-equation blocks for a RAGGED list of trajectories -- fit_multi(X_list, U_list), Koopman/koopmanEDMDc.py:113-152:
 * "Each (X, U) is a bag/rollout.  We never create cross-bag transitions"; a bag with fewer than 2 states contributes nothing (:131-132),
 * bag b contributes the pairs (X_b[t], U_b[t], X_b[t+1]), t = 0 .. len(X_b) - 2 (:133-138).
 * Data: the bags' rows one after the other.  X [rows][n] = np.vstack(X_list) (:125 -- also what the reference clusters for its centres),
 * U [rows][r] ROW-ALIGNED with X (U_b[t] next to X_b[t]; the row next to the last state of a bag is never read -- the reference drops it
 * with U[:-1]), bag b = rows bag_offsets[b] .. bag_offsets[b + 1] - 1, bag_offsets [nbags + 1] on the HOST in both forms (metadata: it is
 * validated there -- bag_offsets[0] = 0, non-decreasing -- and rows = bag_offsets[nbags]).  Empty bags and one-row bags are allowed.
 * One call lifts every state once, whatever the number of bags: no per-bag launch, allocation or copy. */
BROV_API int edmdc_gram_ragged(brov_ctx* ctx, int n, int r, int k, double gamma, const double* C,
                               int64_t nbags, const int64_t* bag_offsets, const double* X, const double* U,
                               int accumulate, double* GtG, double* GtY);
BROV_API int edmdc_gram_ragged_dev(brov_ctx* ctx, int n, int r, int k, double gamma, const double* d_C,
                                   int64_t nbags, const int64_t* bag_offsets, const double* d_X, const double* d_U,
                                   int accumulate, double* d_GtG, double* d_GtY /* NULL: G^T G alone */);
/* Rows lifted per chunk by edmdc_gram* (workspace = rows * padded_width * 8 bytes). */
BROV_API int edmdc_set_chunk_rows(brov_ctx* ctx, int64_t rows);

/* H-step propagation in lifted space + endpoint squared error: KoopmanEDMDc.multistep_rmse /
 * evaluate (Koopman/koopmanEDMDc.py:157-200).  X [N][n], U [N-1][r] or longer (only rows 0..N-2 are read, as in the
 * reference, which accepts len(U) == len(X) - 1), A [d][d], B [d][r];
 * se_total = sum over k < N-H and the n state coordinates of (X[k+H] - x_hat)^2;
 * rmse = sqrt(se_total / ((N-H) * n)).  xhat_end [N-H][n] optional (NULL). */
BROV_API int edmdc_multistep_se(brov_ctx* ctx, int n, int r, int k, double gamma, const double* C,
                       const double* A, const double* B, int64_t N, int64_t H,
                       const double* X, const double* U, double* se_total, double* xhat_end);
/* The same scores by linearity (opt-in; KoopmanEDMDc.multistep_rmse(..., method="linear")): the H-step prediction is
 *   x_hat[w] = (E A^H) phi(x_w) + sum_{t<H} (E A^(H-1-t) B) u_{w+t},   E = first n rows of the identity,
 * so the caller passes RHt [n+k][n] = (E A^H)^T and Gt [H][r][n] with Gt[t] = (E A^(H-1-t) B)^T (host, H products of n x d by d x d) and
 * the device evaluates every window in one pass -- no H-step recurrence.  Same outputs and argument rules as edmdc_multistep_se;
 * agrees with it to the rounding of the explicit powers of A (<= 1e-9 in the RMSE on the fixtures at H = 1 / 10 / 100). */
BROV_API int edmdc_multistep_se_linear(brov_ctx* ctx, int n, int r, int k, double gamma, const double* C,
                       const double* RHt, const double* Gt, int64_t N, int64_t H,
                       const double* X, const double* U, double* se_total, double* xhat_end);
/* KoopmanEDMDc.simulate (Koopman/koopmanEDMDc.py:202-216), batched over nb start states:
 * x0 [nb][n], U_seq [nb][T][r] -> X_pred [nb][T+1][n]. */
BROV_API int edmdc_simulate(brov_ctx* ctx, int n, int r, int k, double gamma, const double* C,
                   const double* A, const double* B, int64_t nb, int64_t T,
                   const double* x0, const double* U_seq, double* X_pred);

/* Lloyd iterations of k-means on device: the E/M loop of scikit-learn's KMeans(algorithm="lloyd") that
 * KoopmanEDMDc.fit / fit_multi run to place the RBF centres (Koopman/koopmanEDMDc.py:85,126).  The host layer
 * obtains the k-means++ initialisation from scikit-learn and passes it in C_io; the final centres come back
 * in C_io.  X [N][n] (row stride x_stride doubles), mean [n] = column means to subtract (NULL = none; centres
 * are then in the centred frame), stop when no label changes, when the summed squared centre shift is
 * <= tol_abs, or after max_iter iterations.  labels [N] int32 (host version: optional), inertia = sum of
 * squared distances to the final centres, n_iter = iterations run.
 * Empty clusters follow scikit-learn 1.7.2 (sklearn/cluster/_k_means_common.pyx): `_relocate_empty_clusters_dense` moves
 * each of them to one of the n_empty samples farthest from their own centre (that sample leaves its cluster's sum), unless
 * all those distances are zero; `_average_centers` then gives a cluster that is still empty the row of the biggest cluster
 * (its mean, or -- in front of it in index order -- its member SUM, as the in-place loop of that function leaves it).
 * The member sums are exact integer sums of the samples in 2^-48 fixed point per coordinate range, so the centres are the
 * same bits from run to run, for every edmdc_set_kmeans_variant, and for any sharding over ranks. */
BROV_API int edmdc_kmeans_lloyd(brov_ctx* ctx, int64_t N, int n, int k, const double* X, const double* mean,
                                double* C_io, int max_iter, double tol_abs, int32_t* labels, double* inertia, int* n_iter);
BROV_API int edmdc_kmeans_lloyd_dev(brov_ctx* ctx, int64_t N, int n, int k, const double* d_X, int64_t x_stride,
                                    const double* mean_host, double* d_C_io, int max_iter, double tol_abs,
                                    int32_t* d_labels, double* inertia, int* n_iter);

/* Column means and (population) variances of d_X [N][n] (row stride x_stride doubles, n <= 16) into HOST arrays mean_host [n],
 * var_host [n] (either may be NULL): what scikit-learn's KMeans takes from `X.mean(axis=0)` and `np.var(X, axis=0)` before the loop
 * above (sklearn/cluster/_kmeans.py: centring and `_tolerance`; reached from Koopman/koopmanEDMDc.py:85,126) -- mean_host is
 * edmdc_kmeans_lloyd_dev's `mean_host`, tol * mean(var_host) its tol_abs.  Two passes over X, fixed summation order: the values
 * depend on (N, n, x_stride) only.  Synchronous (the results are on the host on return). */
BROV_API int edmdc_col_stats_dev(brov_ctx* ctx, int64_t N, int n, const double* d_X, int64_t x_stride, double* mean_host,
                                 double* var_host);

/* k-means++ seeding on device: scikit-learn 1.7.2 `_kmeans_plusplus` with unit sample weights, i.e. the initialisation
 * of the KMeans(n_clusters, n_init="auto", random_state=0) that KoopmanEDMDc.fit constructs
 * (Koopman/koopmanEDMDc.py:85,126).  The random numbers are the caller's, in the order scikit-learn draws them from
 * its RandomState: first_index = random_state.choice(N, p=uniform), then uniforms[(c-1)*n_trials + j] =
 * random_state.uniform(size=n_trials)[j] for centre c = 1..k-1, n_trials = 2 + int(log(k)).  d_X [N][n] (row stride
 * x_stride), mean_host [n] subtracted from every row (NULL = none).  Outputs: d_C [k][n] (centred frame, device),
 * indices_host [k] = rows of X chosen (optional).  Same candidates as scikit-learn unless a drawn value falls within
 * rounding (~1e-13 relative) of a running-sum boundary. */
BROV_API int edmdc_kmeanspp_dev(brov_ctx* ctx, int64_t N, int n, int k, const double* d_X, int64_t x_stride,
                                const double* mean_host, int64_t first_index, int n_trials, const double* uniforms_host,
                                double* d_C, int64_t* indices_host);

/* KoopmanEDMDc.fit evaluates M = pinv(G^T G + ridge I) @ G.T @ Y left to right, i.e. (P G^T) Y
 * (Koopman/koopmanEDMDc.py:97), whereas fit_multi forms P (G^T Y) (:147).  The two differ by the conditioning of the Gram
 * (1e-6 in the H = 100 RMSE at the class defaults k = 200, ridge = 1e-8).  This entry point reproduces fit()'s order on
 * the device: P [p][p] is the host's pinv of the regularised Gram (from edmdc_gram), rows W = G P^T are formed chunk by
 * chunk (fp64 MFMA) and M [p][d] = W^T Y is accumulated over the same pairs as edmdc_gram.  Data layout as edmdc_gram. */
BROV_API int edmdc_pinv_apply(brov_ctx* ctx, int n, int r, int k, double gamma, const double* C,
                              int64_t nbags, int64_t L, int64_t x_bag_stride, int64_t u_bag_stride,
                              const double* X, const double* U, const double* P, double* M);
BROV_API int edmdc_pinv_apply_dev(brov_ctx* ctx, int n, int r, int k, double gamma, const double* d_C,
                                  int64_t nbags, int64_t L, int64_t x_bag_stride, int64_t u_bag_stride,
                                  const double* d_X, const double* d_U, const double* P_host, double* d_M);

/* edmdc_pinv_apply for the ragged bag list of edmdc_gram_ragged (fit()'s product order over fit_multi's data: what a caller gets who
 * wants the better-conditioned association on a trajectory list). */
BROV_API int edmdc_pinv_apply_ragged(brov_ctx* ctx, int n, int r, int k, double gamma, const double* C,
                                     int64_t nbags, const int64_t* bag_offsets, const double* X, const double* U,
                                     const double* P, double* M);
BROV_API int edmdc_pinv_apply_ragged_dev(brov_ctx* ctx, int n, int r, int k, double gamma, const double* d_C,
                                         int64_t nbags, const int64_t* bag_offsets, const double* d_X, const double* d_U,
                                         const double* P_host, double* d_M);

/* Work decomposition of edmdc_gram / edmdc_gram_dev for a shape (no device work; for roofline accounting): the normal
 * equations G^T[G|Y] (Koopman/koopmanEDMDc.py:129-147) are computed as ntasks blocks of 4 x 6 tiles of 16 x 16 per slab of
 * rows, nslabs slabs per chunk; a task executes 24 x 16 x 16 x 2 = 12 288 flop per sample whether a tile is wanted or not. */
BROV_API int edmdc_gram_decomposition(int n, int r, int k, int* ntasks, int* nslabs);
/* The same for edmdc_gram_dev called with d_GtY = NULL: G^T G alone, the Gram pass of KoopmanEDMDc.fit (which never forms G^T Y,
 * Koopman/koopmanEDMDc.py:89-97). */
BROV_API int edmdc_gtg_decomposition(int n, int r, int k, int* ntasks, int* nslabs);
/* The same for edmdc_pinv_apply(_dev), fit()'s own product order (Koopman/koopmanEDMDc.py:97): the rows of W = G P^T are
 * formed per unit of 192 rows by `wrows_items_per_192_rows` blocks of 4 x 6 tiles (`wrows_tiles_wanted` of their tile products
 * are wanted; a tile product is 16 x 16 x 16 x 2 flop per 16 rows = 512 flop per row), and W^T Y is accumulated by `wty_tasks`
 * blocks of 4 x 6 tiles per slab of rows, `wty_slabs` slabs per chunk (12 288 flop per sample and task). */
BROV_API int edmdc_apply_decomposition(int n, int r, int k, int* wrows_items_per_192_rows, int* wrows_tiles_wanted, int* wty_tasks, int* wty_slabs);
/* Lifted-row cache for the fit() sequence edmdc_gram_dev -> (host pinv) -> edmdc_pinv_apply_dev.  The caller lends the ctx a
 * device buffer (16-byte aligned; rows x (padded width + 1) x 8 bytes plus 8 rows of padding per chunk: 45.7 GB for 1e7 states at
 * k = 512 -- what 288 GB of HBM are for).  The next edmdc_gram_dev whose lifted chunks fit lifts them straight into it, and an
 * edmdc_pinv_apply_dev with the SAME pointers, shape, gamma and bag layout reads them back instead of lifting again.  The caller
 * promises not to modify X, U or C between the two calls and keeps the buffer alive until it is withdrawn (d_buffer = NULL).
 * Any other edmdc_gram_dev call invalidates the cached rows. */
BROV_API int edmdc_lift_cache(brov_ctx* ctx, void* d_buffer, size_t bytes);
/* Which kernel forms the rows of W in edmdc_pinv_apply(_dev): 0 = the tuned one (default), 1 = the plain one-row-tile-per-wave
 * form (kept as an independent second implementation for the parity tests; BROV2_APPLY_SIMPLE=1 selects it at brov_create). */
BROV_API int edmdc_set_apply_variant(brov_ctx* ctx, int variant);
/* Lloyd's loop in edmdc_kmeans_lloyd(_dev) -- a mask of three bits; every setting gives the same labels and the same centres, bit for bit
 * (integer member sums), so the non-default ones are the independent second implementations the parity tests compare against:
 *   0      default: E-step with the per-wave candidate filter (triangle inequality over the centre-centre distances) on a sample order
 *          kept sorted by (label, distance to the centre) -- a permutation, re-sorted as labels move; candidates screened in packed fp32
 *          in the frame of the wave's reference centre, exact fp64 only for the certified pair; Hamerly-style distance bounds let an
 *          E-step visit only the samples whose label could change (sorted order: >= 2^18 samples, n <= 14, k <= 512, or k <= 1024 for
 *          n = 12 / 13; bounds and screening: n = 12 / 13, k <= 512; anything else falls back to the plainer forms by itself);
 *   + 1    full scan over all k centres in the caller's order: no filter, no sorting, no bounds;
 *   + 2    candidate filter in the caller's order (no sorted sample order, hence no screening and no bounds);   (1 + 2 is refused)
 *   + 4    distance bounds off: every E-step visits every sample.
 * A library built with -DBROV2_EXPERIMENTS=1 (tools/, A/B measurements; brov_experiments_build() = 1) accepts further bits that swap
 * single stages for their earlier forms (csrc/capi.hip: KMV_*) and reads its tuning knobs from the environment; the default build
 * refuses those bits and reads no environment variable except BROV2_QUIET (silences the one note brov_create may print about the XCD
 * placement probe) and BROV2_RCCL_LIBRARY (below).
 * Changelog: up to round 4 `+ 4` selected the scalar-record kernel and `+ 8 .. + 256` were accepted by every build; since round 5 the
 * bits mean what is listed above and a caller passing the old values to a default build gets BROV_ERR_ARG. */
BROV_API int edmdc_set_kmeans_variant(brov_ctx* ctx, int variant);
/* When the distance bounds take over: an E-step visits only the samples whose bounds fail once at most `rate` of all labels changed in
 * the iteration before (default 0.03; 1 = from the first sorted iteration on, 0 = never).  Speed only: labels and centres do not depend
 * on it (tests/test_gpu_parity.py::test_lloyd_distance_bounds_change_nothing runs both ends). */
BROV_API int edmdc_set_kmeans_bounds_rate(brov_ctx* ctx, double rate);
BROV_API int brov_experiments_build(void);                          /* 1 for a -DBROV2_EXPERIMENTS=1 library */
/* Which samples `_relocate_empty_clusters_dense` moves its empty clusters to is
 * `np.argpartition(distances, -n_empty)[:-n_empty-1:-1]` (sklearn/cluster/_k_means_common.pyx): the selection algorithm decides the
 * order of the n_empty largest (which empty cluster gets which row) and the winner among equal distances -- and NumPy has two of them:
 * its own introselect (numpy/_core/src/npysort/selection.cpp) wherever no SIMD kernel is dispatched, x86-simd-sort's argselect on x86
 * hosts with AVX-512 / AVX2, with different answers.  Without a callback (fn = NULL: plain-C callers) the library applies a restatement
 * of the FORMER -- NumPy's introselect, step for step, pinned by tests/golden/farselect.npz (np.argpartition with its dispatch
 * disabled) and exported as edmdc_far_select_numpy.  The Python layer installs a callback that calls np.argpartition on the host at hand,
 * so that the rows are scikit-learn's on that host whatever it dispatches.  distances [N] (host), rows of the caller's X -- in a sharded
 * run (edmdc_set_kmeans_shard) the distances of ALL ranks' rows in global row order, N = n_global; far_rows_out [n_empty]; return 0 on
 * success. */
typedef int (*brov_far_select_fn)(void* user, const double* distances, int64_t N, int n_empty, int64_t* far_rows_out);
BROV_API int edmdc_set_kmeans_far_select(brov_ctx* ctx, brov_far_select_fn fn, void* user);
/* The library's own rule, host only: far_rows_out[q] = np.argpartition(distances, -n_empty)[N - 1 - q] as NumPy's introselect leaves it
 * (NaN = farthest).  1 <= n_empty <= N. */
BROV_API int edmdc_far_select_numpy(const double* distances, int64_t N, int n_empty, int64_t* far_rows_out);
/* relocations of empty clusters during the last edmdc_kmeans_lloyd(_dev) call (iterations in which at least one took place) */
BROV_API int edmdc_kmeans_relocations(brov_ctx* ctx);
/* How the last edmdc_kmeans_lloyd(_dev) call ran (speed only; none of it changes a result): info[0] = iterations with a relocation,
 * info[1] = re-sorts of the sample order, info[2] = iteration of the first re-sort (0 = none), info[3] = E-steps that visited only the
 * samples whose distance bounds failed.  Ranks of a sharded run report the same numbers. */
BROV_API int edmdc_kmeans_loop_info(brov_ctx* ctx, int info[4]);
/* Sharded Lloyd: every rank calls edmdc_kmeans_lloyd_dev on its own rows with the same initial centres; `fn` is called on the
 * ctx stream's behalf with a DEVICE buffer that has to be combined over all ranks in place before work queued later on the ctx
 * stream reads it: op 0 = sum of `count` int64, op 1 = maximum of `count` uint64.  One call with op 1 (16 words) before the loop,
 * one with op 0 (2 k (n + 1) + 2 words, 53 KB at k = 512) per iteration.  Integer sums: every rank gets the same centres, and
 * they are the centres of the unsharded run bit for bit.  fn = NULL: single rank.  Return 0 on success. */
typedef int (*brov_allreduce_fn)(void* user, void* d_buf, int64_t count, int op);
BROV_API int edmdc_set_kmeans_allreduce(brov_ctx* ctx, brov_allreduce_fn fn, void* user);
/* This rank's place in a sharded k-means: ranks hold contiguous shards of the rows in rank order, row_offset = global index of this
 * rank's first row, n_global = rows over all ranks (defaults: rank 0 of 1).  With an exchange installed (edmdc_set_kmeans_allreduce
 * or edmdc_kmeans_use_comm):
 *   edmdc_kmeanspp_dev seeds over ALL rows -- first_index and the returned indices are global, k <= n_global, every rank passes the
 *     same first_index and uniforms (drawn for n_global rows) and ends with the same centres; two small exchanges per centre (the
 *     ranks' partial potentials, the rows of the next candidates);
 *   edmdc_kmeans_lloyd_dev relocates empty clusters to the farthest rows of the WHOLE set: the ranks' distances are put side by side in
 *     global row order (one all-reduce of n_global words; this path runs when a cluster runs empty, i.e. hardly ever) and every rank applies
 *     the same selection -- the callback of edmdc_set_kmeans_far_select when one is installed, the library's rule otherwise -- to the same
 *     array, then the owner of each chosen row sends its label and coordinates (one small exchange per row): the rows, and with them the
 *     centres, of the unsharded run under the same rule. */
BROV_API int edmdc_set_kmeans_shard(brov_ctx* ctx, int rank, int world, int64_t row_offset, int64_t n_global);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ----------------------------------------
 * Rollouts shard over trajectories with no communication; the sharded EDMDc fit has exactly one exchange: the sum over
 * ranks of the local [GtG | GtY] blocks (4.5 MB at k = 512), after which every rank solves the same p x p system.
 * librccl is bound with dlopen at first use (a copy already mapped by the process, e.g. PyTorch's, is reused; with
 * BROV2_RCCL_LIBRARY set that path is the ONLY candidate -- when it cannot be opened brov_comm_available() is 0 and the
 * brov_comm_* calls return BROV_ERR_COMM, the rest of the library keeps working).  Rank 0 calls brov_comm_unique_id and hands the 128 bytes to the other ranks out of band
 * (file, socket, MPI, torch.distributed store ...); every rank then calls brov_comm_init_rank (collective, blocks until
 * all nranks have joined).  A communicator is bound to one device and is independent of any brov_ctx. */
#define BROV_COMM_ID_BYTES 128
typedef struct brov_comm brov_comm;
BROV_API int brov_comm_available(void);                               /* 1 if librccl could be bound */
BROV_API int brov_comm_unique_id(unsigned char id[BROV_COMM_ID_BYTES]);
BROV_API int brov_comm_init_rank(int device_id, const unsigned char id[BROV_COMM_ID_BYTES], int nranks, int rank, brov_comm** out);
BROV_API void brov_comm_destroy(brov_comm* comm);
BROV_API int brov_comm_nranks(const brov_comm* comm);
BROV_API int brov_comm_rank(const brov_comm* comm);
BROV_API const char* brov_comm_last_error(const brov_comm* comm);
/* In-place sum over all ranks of d_GtG (n_gtg doubles) and d_GtY (n_gty doubles), issued as ONE grouped RCCL all-reduce on
 * `hip_stream` (hipStream_t, NULL = null stream; asynchronous: order later work on the same stream or synchronise it). */
BROV_API int edmdc_gram_allreduce_dev(brov_comm* comm, double* d_GtG, int64_t n_gtg, double* d_GtY, int64_t n_gty, void* hip_stream);
/* In-place all-reduce of `count` 64-bit words on `hip_stream` (asynchronous): op 0 = sum of int64, op 1 = maximum of uint64. */
BROV_API int brov_comm_allreduce_words(brov_comm* comm, void* d_buf, int64_t count, int op, void* hip_stream);
/* Sharded Lloyd without torch: edmdc_kmeans_lloyd_dev on this ctx exchanges its member sums through `comm` (RCCL on the ctx stream);
 * the same as edmdc_set_kmeans_allreduce with a callback that calls brov_comm_allreduce_words.  comm = NULL: single rank again. */
BROV_API int edmdc_kmeans_use_comm(brov_ctx* ctx, brov_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* BROV2_H */
