// capi.hip -- the C ABI of libbrov2.so (include/brov2.h): context, parameters, host-side
// ZOH discretisation, host<->device staging around the kernels of rollout.hip / edmdc.hip /
// controls.hip / propagate.hip.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <ctime>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/brov2.h"
#include "brov2_kernels.h"

using namespace brov;

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
struct brov_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    brov_params params;
    std::string err;
    // timing
    bool timing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    // grow-only scratch arena (device)
    char* scratch = nullptr;
    size_t scratch_cap = 0;
    // derived-parameter cache (per dt)
    bool dp_valid = false;
    double dp_dt = 0.0;
    DevParams dp;
    FastParams fp;
    FastParams* d_fp = nullptr;   // device copy read by the time-loop kernels through the constant address space
    FastParams* d_fp_di = nullptr;    // same struct for the double-integrator models: Tm holds the gains [K_lin | K_ang]^T
    bool di_set = false;
    // EDMDc
    int btu_staging = 0;
    int single_lane = 0;          // 1: never use the two-wave rollout kernel (A/B measurements: BROV2_ROLLOUT_SINGLE_LANE=1)
    int64_t chunk_rows = (int64_t)1 << 20;
    // lifted-row cache of fit(): edmdc_gram_dev lifts its chunks straight into it, edmdc_pinv_apply_dev with the same arguments
    // reads them back instead of lifting again (edmdc_lift_cache; the caller promises not to touch X / U / C in between)
    char* lift_cache = nullptr;   // caller-owned device buffer (edmdc_lift_cache), nullptr = off
    size_t lift_cache_cap = 0;
    bool lift_cache_valid = false;
    struct LiftKey { const void *X, *U, *C; int n, r, k; double gamma; int64_t nbags, L, xs, us, chunk; uint64_t bag_hash; } lift_key = {};
    int kmeans_variant = 0;       // edmdc_set_kmeans_variant: KMV_* bits below
    double km_bounds_rate = 0.03; // edmdc_set_kmeans_bounds_rate
    int apply_variant = 0;        // edmdc_pinv_apply: 0 = wrows_kernel (tuned), 1 = the round-2 kernel (second implementation of the tests)
    void* d_tasks[3] = {nullptr, nullptr, nullptr};      // Gram task tables: [0] G^T[G|Y], [1] W^T Y (edmdc_pinv_apply), [2] G^T G alone
    EdmdcShape task_shape[3] = {};
    int ntasks[3] = {0, 0, 0};
    // persistent EDMDc workspaces (separate from the per-call arena so that accumulate works across calls)
    double* d_partial = nullptr;
    size_t partial_cap = 0;
    size_t tasks_cap[3] = {0, 0, 0};  // bytes behind d_tasks[]
    hipEvent_t ev_handover = nullptr; // orders the work queued on the previous stream before the next one (brov_set_stream)
    hipStream_t side[3] = {nullptr, nullptr, nullptr};   // extra streams of edmdc_multistep_se (window groups advance independently), created on demand
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    double* h_io = nullptr;           // pinned, device-mapped staging of the per-call entry points (brov_rhs / brov_thruster_forces with a
    double* d_io = nullptr;           // handful of vehicles): the kernel reads and writes host memory, no copy calls; d_io = its device alias
    unsigned long long io_seq = 0;    // sequence number of the last per-call launch (completion flags at the end of the staging block)
    std::vector<std::pair<void*, size_t>> pool;   // freed brov_malloc blocks kept for the next request (brov_malloc / brov_free)
    std::unordered_map<void*, size_t> live;       // blocks handed out by brov_malloc: their sizes
    size_t pool_bytes = 0;
    char* h_stage[2] = {nullptr, nullptr};        // pinned staging blocks of brov_upload_bags (created on first use)
    hipEvent_t ev_stage[2] = {nullptr, nullptr};  // "the DMA that read block i has finished"
    int upload_threads = 0;                       // host threads packing a block (0 = not probed yet)
    double* h_stats = nullptr;        // pinned, device-mapped: the M-step stores an iteration's statistics there while the next E-step is queued
    double km_seq = 0.0;              // number of the last M-step launched (wait_stats)
    double* d_stats_map = nullptr;    // its device alias
    brov_far_select_fn far_select = nullptr;      // rows an empty cluster is relocated to (edmdc_set_kmeans_far_select); nullptr = descending selection
    void* far_select_user = nullptr;
    brov_allreduce_fn km_allreduce = nullptr;     // sharded Lloyd (edmdc_set_kmeans_allreduce): sums / maxima over ranks, stream-ordered
    void* km_allreduce_user = nullptr;
    brov_comm* km_comm = nullptr;                 // edmdc_kmeans_use_comm: the exchanges go to this communicator
    long long km_row_offset = 0;                  // global index of this rank's first row (edmdc_set_kmeans_shard)
    int km_rank = 0, km_world = 1;                // this rank's place in a sharded k-means
    long long km_n_global = 0;                    // rows over all ranks
    int kmeans_relocations = 0;       // relocations of empty clusters in the last edmdc_kmeans_lloyd(_dev) call
    int km_info[4] = {0, 0, 0, 0};    // of the last Lloyd call: [1] re-sorts of the sample order, [2] iteration of the first one, [3] E-steps in list form
    int prop_groups = 2;              // window groups of edmdc_multistep_se, 1..4 (BROV2_PROP_GROUPS; 1 = everything on the ctx stream)
    int xcd_round_robin = -1;         // -1 not probed, 0 no, 1 yes: blockIdx % 8 groups blocks by XCD (speed only)
    char arch[64] = {0};
};

// edmdc_set_kmeans_variant.  The public surface is three bits (include/brov2.h); the others select the independent second
// implementations of single stages and exist only in a -DBROV2_EXPERIMENTS=1 build (tools/, A/B runs, the cross-checks of the test
// suite when it is pointed at such a library): same labels, same centres, bit for bit, whatever is set.
enum : int {
    KMV_FULL_SCAN = 1,          // E-step over all k centres (no candidate filter), the caller's order
    KMV_UNSORTED = 2,           // candidate filter in the caller's order (no sorted sample order)
    KMV_NO_BOUNDS = 4,          // distance bounds off: every E-step visits every sample
    KMV_PUBLIC = 7,
    KMV_PP_UNSCREENED = 8,      // seeding: every row through its fp64 distances (no float screening)
    KMV_MASK_FILTER = 16,       // the round-3 form of the candidate filter alone (label groups, masks over all centres)
    KMV_PP_SHARD_KERNELS = 32,  // a single rank's seeding through the kernels of the sharded run
    KMV_PK_STANDALONE = 64,     // sorted loop: the stand-alone packed-fp32 kernel (the form k = 513..1024 always takes)
    KMV_NO_SCREENING = 128,     // packed-fp32 screening off: fp64 evaluation of every candidate
    KMV_SCALAR_RECORDS = 256,   // centre records through scalar registers (the kernel k > 1024 / n = 15 always take)
#ifdef BROV2_EXPERIMENTS
    KMV_ACCEPTED = 511,
#else
    KMV_ACCEPTED = KMV_PUBLIC,
#endif
};

// host <-> device copies of the ctx (defined with the device memory helpers below): through the ctx's pinned blocks, never by
// handing pageable memory to the runtime in the size range it would pin in place
static hipError_t h2d_copy(brov_ctx* c, void* dst, const void* src, size_t bytes);
static hipError_t d2h_copy(brov_ctx* c, void* dst, const void* src, size_t bytes);

namespace {

int fail(brov_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}
int hip_fail(brov_ctx* c, hipError_t e, const char* what) {
    return fail(c, e == hipErrorOutOfMemory ? BROV_ERR_NOMEM : BROV_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPCK(ctx, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return hip_fail((ctx), e__, #call); } while (0)

// Makes the ctx's device current for the duration of one API call and puts the caller's device back afterwards:
// the process-wide "current device" belongs to the caller (PyTorch reads it for its own allocations and launches).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(const brov_ctx* c) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != c->device) { (void)hipSetDevice(c->device); switched = true; }
    }
    ~DeviceGuard() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// RAII: records the HIP events that bracket the kernels of one API call
struct CallTimer {
    brov_ctx* c;
    explicit CallTimer(brov_ctx* ctx) : c(ctx) {
        c->timed = false;
        if (c->timing) (void)hipEventRecord(c->ev0, c->stream);
    }
    ~CallTimer() {
        if (c->timing) { (void)hipEventRecord(c->ev1, c->stream); c->timed = true; }
    }
};

// bump allocator over the ctx scratch arena; grows (after a stream sync) when too small
struct Arena {
    brov_ctx* c;
    size_t off = 0;
    std::vector<size_t> wants;
    explicit Arena(brov_ctx* ctx) : c(ctx) {}
    static size_t al(size_t b) { return (b + 255) & ~(size_t)255; }
    int reserve(size_t total) {
        total = al(total) + 4096;
        if (total <= c->scratch_cap) return BROV_OK;
        (void)hipStreamSynchronize(c->stream);
        if (c->scratch) (void)hipFree(c->scratch);
        c->scratch = nullptr;
        c->scratch_cap = 0;
        hipError_t e = hipMalloc((void**)&c->scratch, total);
        if (e != hipSuccess) return hip_fail(c, e, "hipMalloc(scratch)");
        c->scratch_cap = total;
        return BROV_OK;
    }
    template <typename T> T* take(size_t count) {
        T* p = reinterpret_cast<T*>(c->scratch + off);
        off += al(count * sizeof(T));
        return p;
    }
};

// The lifted-row cache (edmdc_lift_cache) is keyed by the DEVICE addresses of X / U / C.  The host entry points edmdc_gram and
// edmdc_pinv_apply stage their arrays in allocations of their own and free them on return: a cache armed with those addresses
// would be found valid by the next host call of the same shape -- hipMalloc hands the same addresses out again -- and that call
// would read the lifted rows of the PREVIOUS data.  Only a direct _dev call (whose buffers the caller owns and promises to leave
// alone) may leave the cache armed: the host wrappers disarm it on entry and on every way out.
struct LiftCacheDisarm {
    brov_ctx* c;
    explicit LiftCacheDisarm(brov_ctx* ctx) : c(ctx) { c->lift_cache_valid = false; }
    ~LiftCacheDisarm() { c->lift_cache_valid = false; }
};

// ... and the same holds for buffers that come from brov_malloc (the Python host paths stage their lists there): freeing or
// overwriting one through this library while it is a key of the armed cache disarms it.
static void lift_cache_touch(brov_ctx* c, const void* p) {
    if (c->lift_cache_valid && p && (p == c->lift_key.X || p == c->lift_key.U || p == c->lift_key.C)) c->lift_cache_valid = false;
}

// ---- small dense helpers (host, fp64) -------------------------------------------------------
void matmul(int n, const double* A, const double* B, double* C) {
    std::vector<double> t(n * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double a = 0.0;
            for (int k = 0; k < n; ++k) a += A[i * n + k] * B[k * n + j];
            t[i * n + j] = a;
        }
    std::memcpy(C, t.data(), sizeof(double) * n * n);
}
// solve P X = Q (n x n), partial pivoting, in place on copies
bool solve(int n, const double* P, const double* Q, double* X) {
    std::vector<double> a(P, P + n * n), b(Q, Q + n * n);
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r) if (std::fabs(a[r * n + c]) > std::fabs(a[piv * n + c])) piv = r;
        if (a[piv * n + c] == 0.0) return false;
        if (piv != c) for (int j = 0; j < n; ++j) { std::swap(a[c * n + j], a[piv * n + j]); std::swap(b[c * n + j], b[piv * n + j]); }
        for (int r = c + 1; r < n; ++r) {
            const double f = a[r * n + c] / a[c * n + c];
            if (f == 0.0) continue;
            for (int j = c; j < n; ++j) a[r * n + j] -= f * a[c * n + j];
            for (int j = 0; j < n; ++j) b[r * n + j] -= f * b[c * n + j];
        }
    }
    for (int j = 0; j < n; ++j)
        for (int r = n - 1; r >= 0; --r) {
            double s = b[r * n + j];
            for (int k = r + 1; k < n; ++k) s -= a[r * n + k] * X[k * n + j];
            X[r * n + j] = s / a[r * n + r];
        }
    return true;
}
// Matrix exponential, scaling and squaring with the [13/13] Pade approximant (Higham 2005).
bool expm_pade13(int n, const double* Ain, double* E) {
    static const double b[14] = {64764752532480000.0, 32382376266240000.0, 7771770303897600.0, 1187353796428800.0,
                                 129060195264000.0, 10559470521600.0, 670442572800.0, 33522128640.0,
                                 1323241920.0, 40840800.0, 960960.0, 16380.0, 182.0, 1.0};
    const int nn = n * n;
    std::vector<double> A(Ain, Ain + nn), A2(nn), A4(nn), A6(nn), U(nn), V(nn), T1(nn), T2(nn);
    double norm1 = 0.0;
    for (int j = 0; j < n; ++j) { double s = 0.0; for (int i = 0; i < n; ++i) s += std::fabs(A[i * n + j]); norm1 = std::fmax(norm1, s); }
    int s = 0;
    const double theta13 = 5.371920351148152;
    if (norm1 > theta13) s = (int)std::ceil(std::log2(norm1 / theta13));
    if (s < 0) s = 0;
    const double sc = std::ldexp(1.0, -s);
    for (auto& v : A) v *= sc;
    matmul(n, A.data(), A.data(), A2.data());
    matmul(n, A2.data(), A2.data(), A4.data());
    matmul(n, A4.data(), A2.data(), A6.data());
    for (int i = 0; i < nn; ++i) T1[i] = b[13] * A6[i] + b[11] * A4[i] + b[9] * A2[i];
    matmul(n, A6.data(), T1.data(), T2.data());
    for (int i = 0; i < nn; ++i) T2[i] += b[7] * A6[i] + b[5] * A4[i] + b[3] * A2[i];
    for (int i = 0; i < n; ++i) T2[i * n + i] += b[1];
    matmul(n, A.data(), T2.data(), U.data());
    for (int i = 0; i < nn; ++i) T1[i] = b[12] * A6[i] + b[10] * A4[i] + b[8] * A2[i];
    matmul(n, A6.data(), T1.data(), V.data());
    for (int i = 0; i < nn; ++i) V[i] += b[6] * A6[i] + b[4] * A4[i] + b[2] * A2[i];
    for (int i = 0; i < n; ++i) V[i * n + i] += b[0];
    for (int i = 0; i < nn; ++i) { T1[i] = V[i] - U[i]; T2[i] = V[i] + U[i]; }
    if (!solve(n, T1.data(), T2.data(), E)) return false;
    for (int q = 0; q < s; ++q) matmul(n, E, E, E);
    return true;
}

bool discretise(const brov_params& p, double dt, double Ad[9], double Bd[3]) {
    // ZOH: expm(dt * [[Ac, Bc], [0, 0]]) -> top rows  (scipy.signal.cont2discrete(method="zoh"))
    double M[16] = {0}, E[16];
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) M[4 * i + j] = p.lag_Ac[3 * i + j] * dt; M[4 * i + 3] = p.lag_Bc[i] * dt; }
    if (!expm_pade13(4, M, E)) return false;
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) Ad[3 * i + j] = E[4 * i + j]; Bd[i] = E[4 * i + 3]; }
    return true;
}

void alloc_matrix(const brov_params& p, double T[6][8]) {
    for (int i = 0; i < 8; ++i) {
        const double* r = p.thr_r[i];
        const double* d = p.thr_dir[i];
        T[0][i] = d[0]; T[1][i] = d[1]; T[2][i] = d[2];
        T[3][i] = r[1] * d[2] - r[2] * d[1];
        T[4][i] = r[2] * d[0] - r[0] * d[2];
        T[5][i] = r[0] * d[1] - r[1] * d[0];
    }
}

bool derive(const brov_params& p, double dt, DevParams& o) {
    std::memset(&o, 0, sizeof o);
    const double md[6] = {p.m - p.added_mass[0], p.m - p.added_mass[1], p.m - p.added_mass[2],
                          p.Ix - p.added_mass[3], p.Iy - p.added_mass[4], p.Iz - p.added_mass[5]};
    for (int i = 0; i < 6; ++i) {
        o.md[i] = md[i];
        o.minv[i] = 1.0 / md[i];
        o.dl[i] = -p.lin_damp[i] + 0.0;   // -(-0.0) = +0.0
        o.dq[i] = -p.quad_damp[i];
    }
    const double W = p.m * p.g, B = p.rho * p.g * p.volume;
    o.WmB = W - B;
    o.xbB = p.xb * B; o.ybB = p.yb * B; o.zbB = p.zb * B;
    o.has_current = 0;
    for (int i = 0; i < 3; ++i) { o.cur[i] = p.current[i]; if (p.current[i] != 0.0) o.has_current = 1; }
    alloc_matrix(p, o.alloc);
    for (int i = 0; i < 5; ++i) o.poly[i] = p.thrust_poly[i];
    double Ad[9], Bd[3];
    if (!discretise(p, dt, Ad, Bd)) return false;
    // powers: A^s, b_s = (I + A + ... + A^(s-1)) Bd, c_s = Cc A^s, d_s = Cc b_s
    double As[9], bs[3];
    std::memcpy(As, Ad, sizeof As);
    std::memcpy(bs, Bd, sizeof bs);
    for (int s = 0; s < 4; ++s) {
        if (s > 0) {
            double An[9], bn[3];
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j) An[3 * i + j] = Ad[3 * i] * As[j] + Ad[3 * i + 1] * As[3 + j] + Ad[3 * i + 2] * As[6 + j];
                bn[i] = Ad[3 * i] * bs[0] + Ad[3 * i + 1] * bs[1] + Ad[3 * i + 2] * bs[2] + Bd[i];
            }
            std::memcpy(As, An, sizeof As);
            std::memcpy(bs, bn, sizeof bs);
        }
        for (int j = 0; j < 9; ++j) o.lag_A[s][j] = As[j];
        for (int j = 0; j < 3; ++j) {
            o.lag_b[s][j] = bs[j];
            o.lag_c[s][j] = p.lag_Cc[0] * As[j] + p.lag_Cc[1] * As[3 + j] + p.lag_Cc[2] * As[6 + j];
        }
        o.lag_d[s] = p.lag_Cc[0] * bs[0] + p.lag_Cc[1] * bs[1] + p.lag_Cc[2] * bs[2];
    }
    return true;
}

// time-loop form of the constants (brov2_fast.h): everything pre-multiplied by Minv
void derive_fast(const brov_params& p, const DevParams& d, FastParams& f) {
    std::memset(&f, 0, sizeof f);
    const double* md = d.md;
    const double* mi = d.minv;
    f.E[0] = mi[0] * md[2];  f.E[1] = mi[0] * md[1];
    f.E[2] = mi[1] * md[0];  f.E[3] = mi[1] * md[2];
    f.E[4] = mi[2] * md[1];  f.E[5] = mi[2] * md[0];
    f.E[6] = mi[3] * (md[2] - md[1]);  f.E[7] = mi[3] * (md[5] - md[4]);
    f.E[8] = mi[4] * (md[0] - md[2]);  f.E[9] = mi[4] * (md[3] - md[5]);
    f.E[10] = mi[5] * (md[1] - md[0]); f.E[11] = mi[5] * (md[4] - md[3]);
    for (int i = 0; i < 6; ++i) { f.da[i] = mi[i] * d.dl[i]; f.db[i] = mi[i] * d.dq[i]; f.minv[i] = mi[i]; }
    f.G[0] = mi[0] * d.WmB; f.G[1] = mi[1] * d.WmB; f.G[2] = mi[2] * d.WmB;
    f.G[3] = mi[3] * d.zbB; f.G[4] = mi[4] * d.zbB;
    f.XY[0] = mi[3] * d.ybB; f.XY[1] = mi[4] * d.xbB; f.XY[2] = mi[5] * d.xbB; f.XY[3] = mi[5] * d.ybB;
    f.has_xy = (p.xb != 0.0 || p.yb != 0.0) ? 1 : 0;
    f.has_current = d.has_current;
    for (int i = 0; i < 3; ++i) f.cur[i] = d.cur[i];
    f.tm_dense = 0;
    for (int k = 0; k < 6; ++k)
        for (int i = 0; i < 8; ++i) {
            f.Tm[k][i] = mi[k] * d.alloc[k][i];
            const bool expect_zero = (k == 2) ? (i < 4) : ((k == 0 || k == 1 || k == 5) ? (i >= 4) : false);
            if (expect_zero && d.alloc[k][i] != 0.0) f.tm_dense = 1;
        }
    for (int i = 0; i < 5; ++i) f.poly[i] = d.poly[i];
    for (int s = 0; s < 4; ++s) { for (int j = 0; j < 3; ++j) f.lc[s][j] = d.lag_c[s][j]; f.ld[s] = d.lag_d[s]; }
    for (int j = 0; j < 9; ++j) { f.A1[j] = d.lag_A[0][j]; f.A4[j] = d.lag_A[3][j]; }
    for (int j = 0; j < 3; ++j) { f.b1[j] = d.lag_b[0][j]; f.b4[j] = d.lag_b[3][j]; }
    // observer basis w = O z, rows of O = Cc Ad^1..3 (LagZ::to_observer): constants in long double, O^-1 by the adjugate
    long double O[9], Oi[9];
    for (int r = 0; r < 3; ++r) for (int j = 0; j < 3; ++j) O[3 * r + j] = d.lag_c[r][j];
    const long double det = O[0] * (O[4] * O[8] - O[5] * O[7]) - O[1] * (O[3] * O[8] - O[5] * O[6]) + O[2] * (O[3] * O[7] - O[4] * O[6]);
    long double nO = 0, nOi = 0;
    f.obs_bad = 1;
    if (det != 0.0L && std::isfinite((double)det)) {
        Oi[0] = (O[4] * O[8] - O[5] * O[7]) / det; Oi[1] = (O[2] * O[7] - O[1] * O[8]) / det; Oi[2] = (O[1] * O[5] - O[2] * O[4]) / det;
        Oi[3] = (O[5] * O[6] - O[3] * O[8]) / det; Oi[4] = (O[0] * O[8] - O[2] * O[6]) / det; Oi[5] = (O[2] * O[3] - O[0] * O[5]) / det;
        Oi[6] = (O[3] * O[7] - O[4] * O[6]) / det; Oi[7] = (O[1] * O[6] - O[0] * O[7]) / det; Oi[8] = (O[0] * O[4] - O[1] * O[3]) / det;
        for (int j = 0; j < 9; ++j) { nO += O[j] * O[j]; nOi += Oi[j] * Oi[j]; }
        const long double cond = sqrtl(nO) * sqrtl(nOi);           // Frobenius condition number
        if (std::isfinite((double)cond) && cond < 1.0e4L) {
            f.obs_bad = 0;
            long double OA[9];                                       // O A4
            for (int r = 0; r < 3; ++r) for (int j = 0; j < 3; ++j) {
                long double a = 0;
                for (int m = 0; m < 3; ++m) a += O[3 * r + m] * (long double)d.lag_A[3][3 * m + j];
                OA[3 * r + j] = a;
            }
            for (int r = 0; r < 3; ++r) {
                long double b4 = 0, g = 0;
                for (int j = 0; j < 3; ++j) {
                    long double a = 0;
                    for (int m = 0; m < 3; ++m) a += OA[3 * r + m] * Oi[3 * m + j];
                    f.Aw4[3 * r + j] = (double)a;
                    b4 += O[3 * r + j] * (long double)d.lag_b[3][j];
                    g += O[3 * r + j] * (long double)d.lag_b[0][j];
                }
                f.bw4[r] = (double)b4;
                f.g1[r] = (double)g;
            }
            for (int j = 0; j < 3; ++j) {
                long double a = 0;
                for (int m = 0; m < 3; ++m) a += (long double)d.lag_c[3][m] * Oi[3 * m + j];
                f.al4[j] = (double)a;
            }
            for (int j = 0; j < 9; ++j) f.Ob[j] = (double)O[j];
        }
    }
}

int get_dp(brov_ctx* c, double dt, const DevParams** out) {
    if (!(dt > 0.0) || !std::isfinite(dt)) return fail(c, BROV_ERR_ARG, "dt must be finite and > 0");
    if (!c->dp_valid || c->dp_dt != dt) {
        if (!derive(c->params, dt, c->dp)) return fail(c, BROV_ERR_ARG, "thruster-lag discretisation failed (singular Pade system)");
        derive_fast(c->params, c->dp, c->fp);
        if (!c->d_fp) HIPCK(c, hipMalloc((void**)&c->d_fp, sizeof(FastParams)));
        // stream ordered: kernels already queued keep the old constants; c->fp outlives the copy (sync below)
        HIPCK(c, h2d_copy(c, c->d_fp, &c->fp, sizeof(FastParams)));
        HIPCK(c, hipStreamSynchronize(c->stream));
        c->dp_dt = dt;
        c->dp_valid = true;
    }
    *out = &c->dp;
    return BROV_OK;
}

bool model_ok(int m) { return m >= BROV_THRUSTER_EULER && m <= BROV_DI_WRENCH_QUAT; }
bool model_is_di_h(int m) { return m >= BROV_DI_THRUSTER_EULER; }
int NX(int m) { return (m == BROV_WRENCH_QUAT || m == BROV_DI_WRENCH_QUAT) ? 13 : 12; }
int NU(int m) { return (m == BROV_THRUSTER_EULER || m == BROV_DI_THRUSTER_EULER) ? 8 : 6; }

}  // namespace

// ------------------------------------------------------------------------------------------
extern "C" {

int brov_abi_version(void) { return BROV2_ABI_VERSION; }
int brov_model_nx(int model) { return model_ok(model) ? NX(model) : BROV_ERR_ARG; }
int brov_model_nu(int model) { return model_ok(model) ? NU(model) : BROV_ERR_ARG; }

void brov_default_params(brov_params* p) {
    if (!p) return;
    std::memset(p, 0, sizeof *p);
    // fossen/BlueROV2.py:81-99
    p->rho = 1000.0; p->g = 9.82; p->m = 13.5; p->volume = 0.0134;
    p->xb = 0.0; p->yb = 0.0; p->zb = -0.01;
    p->Ix = 0.26; p->Iy = 0.23; p->Iz = 0.37;
    // :111-116, :129-140
    const double am[6] = {-6.36, -7.12, -18.68, -0.189, -0.135, -0.222};
    const double dl[6] = {-13.7, -0.0, -33.0, -0.0, -0.8, -0.0};
    const double dq[6] = {-141.0, -217.0, -190.0, -1.19, -0.47, -1.5};
    for (int i = 0; i < 6; ++i) { p->added_mass[i] = am[i]; p->lin_damp[i] = dl[i]; p->quad_damp[i] = dq[i]; }
    // :172-232 thruster geometry: r_i = Rz(a_i) r_base (angles as printed there), directions Rz(b_i) e or -z
    const double r1234[3] = {0.156, 0.111, 0.085}, r5678[3] = {0.12, 0.218, 0.0};
    const double e[3] = {1.0 / std::sqrt(2.0), -1.0 / std::sqrt(2.0), 0.0};
    const double ar[8] = {0.0, 5.05, 1.91, M_PI, 0.0, 4.15, 1.01, M_PI};
    const double ae[4] = {0.0, M_PI / 2, 3 * M_PI / 2, M_PI};
    for (int i = 0; i < 8; ++i) {
        const double* rb = i < 4 ? r1234 : r5678;
        const double s = std::sin(ar[i]), c = std::cos(ar[i]);
        p->thr_r[i][0] = c * rb[0] - s * rb[1];
        p->thr_r[i][1] = s * rb[0] + c * rb[1];
        p->thr_r[i][2] = rb[2];
        if (i < 4) {
            const double se = std::sin(ae[i]), ce = std::cos(ae[i]);
            p->thr_dir[i][0] = ce * e[0] - se * e[1];
            p->thr_dir[i][1] = se * e[0] + ce * e[1];
            p->thr_dir[i][2] = 0.0;
        } else {
            p->thr_dir[i][0] = 0.0; p->thr_dir[i][1] = 0.0; p->thr_dir[i][2] = -1.0;
        }
    }
    // :257 thrust curve, :476-480 lag
    const double poly[5] = {8.9, 176.0, -404.1, 389.9, -140.3};
    for (int i = 0; i < 5; ++i) p->thrust_poly[i] = poly[i];
    const double Ac[9] = {-89.0, -72.33, -26.54, 128.0, 0.0, 0.0, 0.0, 32.0, 0.0};
    const double Bc[3] = {8.0, 0.0, 0.0}, Cc[3] = {0.0, 5.992, 3.317};
    for (int i = 0; i < 9; ++i) p->lag_Ac[i] = Ac[i];
    for (int i = 0; i < 3; ++i) { p->lag_Bc[i] = Bc[i]; p->lag_Cc[i] = Cc[i]; }
}

int brov_arch_is_supported(const char* gcn_arch_name) {
    // the library carries gfx950 code objects only ("gfx950:sramecc+:xnack-" is what hipDeviceProp_t.gcnArchName reads on MI355X)
    return gcn_arch_name && std::strncmp(gcn_arch_name, "gfx950", 6) == 0 && (gcn_arch_name[6] == '\0' || gcn_arch_name[6] == ':');
}

int brov_create(int device_id, brov_ctx** out) {
    if (!out) return BROV_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return BROV_ERR_NODEVICE;
    if (device_id < 0 || device_id >= ndev) return BROV_ERR_ARG;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return BROV_ERR_HIP;
    if (!brov_arch_is_supported(prop.gcnArchName)) return BROV_ERR_NODEVICE;     // a GPU, but not one this library has code for
    brov_ctx* c = new (std::nothrow) brov_ctx();
    if (!c) return BROV_ERR_NOMEM;
    c->device = device_id;
    std::snprintf(c->arch, sizeof c->arch, "%s", prop.gcnArchName);
    brov_default_params(&c->params);
    DeviceGuard g(c);               // the caller's current device is restored on return
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_handover, hipEventDisableTiming) != hipSuccess) {
        if (c->ev0) (void)hipEventDestroy(c->ev0);
        if (c->ev1) (void)hipEventDestroy(c->ev1);
        if (c->ev_handover) (void)hipEventDestroy(c->ev_handover);
        delete c;
        return BROV_ERR_HIP;
    }
    // gram_kernel / propagate_kernel group work items by blockIdx % 8 so that blocks sharing an XCD share rows through its
    // L2.  That placement is observed behaviour, not a HIP guarantee: probe it once and remember (speed only, never correctness).
    c->xcd_round_robin = probe_xcd_round_robin(nullptr);
#ifdef BROV2_EXPERIMENTS
    // run-time knobs of the experiments build (tools/): the default build reads no environment variable but BROV2_QUIET (below) and
    // BROV2_RCCL_LIBRARY (comm.hip), both described in include/brov2.h
    if (const char* e = std::getenv("BROV2_ROLLOUT_SINGLE_LANE")) c->single_lane = (e[0] == '1');
    if (const char* e = std::getenv("BROV2_APPLY_SIMPLE")) c->apply_variant = (e[0] == '1');
    if (const char* e = std::getenv("BROV2_KMEANS_VARIANT")) { const int v = std::atoi(e); if (v >= 0 && v <= KMV_ACCEPTED && (v & 3) != 3) c->kmeans_variant = v; }
    if (const char* e = std::getenv("BROV2_PROP_GROUPS")) { const int g = std::atoi(e); if (g >= 1 && g <= 4) c->prop_groups = g; }
    if (const char* e = std::getenv("BROV2_KM_BOUNDS_RATE")) c->km_bounds_rate = std::atof(e);
#endif
    if (c->xcd_round_robin != 1 && std::getenv("BROV2_QUIET") == nullptr)
        std::fprintf(stderr, "[libbrov2] note: workgroups are not dealt round-robin over the XCDs on device %d (probe=%d); "
                             "the XCD-aware block mappings lose their L2 sharing (results unaffected)\n", device_id, c->xcd_round_robin);
    *out = c;
    return BROV_OK;
}

static void pool_release(brov_ctx* c);

void brov_destroy(brov_ctx* c) {
    if (!c) return;
    DeviceGuard g(c);
    (void)hipStreamSynchronize(c->stream);
    pool_release(c);
    if (c->scratch) (void)hipFree(c->scratch);
    for (int m = 0; m < 3; ++m) if (c->d_tasks[m]) (void)hipFree(c->d_tasks[m]);
    if (c->d_fp) (void)hipFree(c->d_fp);
    if (c->d_fp_di) (void)hipFree(c->d_fp_di);
    if (c->d_partial) (void)hipFree(c->d_partial);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_handover) (void)hipEventDestroy(c->ev_handover);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->h_stats) (void)hipHostFree(c->h_stats);
    if (c->h_io) (void)hipHostFree(c->h_io);
    for (int i = 0; i < 2; ++i) {
        if (c->h_stage[i]) (void)hipHostFree(c->h_stage[i]);
        if (c->ev_stage[i]) (void)hipEventDestroy(c->ev_stage[i]);
    }
    for (int i = 0; i < 3; ++i) {
        if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
        if (c->side[i]) (void)hipStreamDestroy(c->side[i]);
    }
    delete c;
}

int brov_device_arch(const brov_ctx* c, char* buf, size_t cap) {
    if (!c || !buf || cap == 0) return BROV_ERR_ARG;
    std::snprintf(buf, cap, "%s", c->arch);
    return BROV_OK;
}
int brov_xcd_round_robin(const brov_ctx* c) { return c ? c->xcd_round_robin : BROV_ERR_ARG; }

const char* brov_last_error(const brov_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

int brov_set_stream(brov_ctx* c, void* s) {
    if (!c) return BROV_ERR_ARG;
    hipStream_t ns = reinterpret_cast<hipStream_t>(s);
    if (ns == c->stream) return BROV_OK;
    // The scratch arena, the Gram partials and the constant buffers are per ctx: work still queued on the old stream may be
    // using them, so the new stream waits for it (event hand-over).  If the old handle is no longer valid (a destroyed torch
    // stream), fall back to a device-wide sync.
    DeviceGuard g(c);
    if (hipEventRecord(c->ev_handover, c->stream) != hipSuccess || hipStreamWaitEvent(ns, c->ev_handover, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();
    }
    c->stream = ns;
    return BROV_OK;
}
int brov_sync(brov_ctx* c) {
    if (!c) return BROV_ERR_ARG;
    DeviceGuard g(c);
    HIPCK(c, hipStreamSynchronize(c->stream));
    return BROV_OK;
}
int brov_set_timing(brov_ctx* c, int enabled) {
    if (!c) return BROV_ERR_ARG;
    c->timing = enabled != 0;
    c->timed = false;
    return BROV_OK;
}
int brov_last_kernel_ms(brov_ctx* c, float* ms) {
    if (!c || !ms) return BROV_ERR_ARG;
    if (!c->timed) return fail(c, BROV_ERR_ARG, "no timed call (brov_set_timing(ctx,1) first)");
    DeviceGuard g(c);
    HIPCK(c, hipEventSynchronize(c->ev1));
    HIPCK(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return BROV_OK;
}

int brov_set_params(brov_ctx* c, const brov_params* p) {
    if (!c || !p) return BROV_ERR_ARG;
    const double md[6] = {p->m - p->added_mass[0], p->m - p->added_mass[1], p->m - p->added_mass[2],
                          p->Ix - p->added_mass[3], p->Iy - p->added_mass[4], p->Iz - p->added_mass[5]};
    for (int i = 0; i < 6; ++i)
        if (!(md[i] != 0.0) || !std::isfinite(md[i])) return fail(c, BROV_ERR_ARG, "singular mass matrix");
    c->params = *p;
    c->dp_valid = false;
    return BROV_OK;
}
int brov_get_params(const brov_ctx* c, brov_params* p) {
    if (!c || !p) return BROV_ERR_ARG;
    *p = c->params;
    return BROV_OK;
}
int brov_get_derived(const brov_params* pp, double Minv6[6], double alloc6x8[48]) {
    if (!Minv6 || !alloc6x8) return BROV_ERR_ARG;
    brov_params def;
    if (!pp) { brov_default_params(&def); pp = &def; }
    const brov_params& p = *pp;
    const double md[6] = {p.m - p.added_mass[0], p.m - p.added_mass[1], p.m - p.added_mass[2],
                          p.Ix - p.added_mass[3], p.Iy - p.added_mass[4], p.Iz - p.added_mass[5]};
    for (int i = 0; i < 6; ++i) Minv6[i] = 1.0 / md[i];
    double T[6][8];
    alloc_matrix(p, T);
    std::memcpy(alloc6x8, T, sizeof T);
    return BROV_OK;
}
int brov_discretise_lag(const brov_params* p, double dt, double Ad[9], double Bd[3]) {
    if (!Ad || !Bd || !(dt > 0.0) || !std::isfinite(dt)) return BROV_ERR_ARG;
    brov_params def;
    if (!p) { brov_default_params(&def); p = &def; }
    return discretise(*p, dt, Ad, Bd) ? BROV_OK : BROV_ERR_ARG;
}

// ---- device memory helpers ------------------------------------------------------------------
// brov_malloc / brov_free keep freed blocks in a small per-ctx pool (like any caching allocator): a fit() of the drop-in class
// allocates and releases the same handful of buffers on every call (hipMalloc + hipFree of 4 MB: 0.3 ms a pair, six pairs per fit at
// the recorded size; a pooled pair: 0.02 ms).  A request takes the smallest pooled block of at least its size and at most twice its
// size (+ 1 MB); blocks above POOL_BLOCK_MAX or beyond POOL_TOTAL_MAX in total go back to the driver at once.
constexpr size_t POOL_BLOCK_MAX = (size_t)256 << 20;
constexpr size_t POOL_TOTAL_MAX = (size_t)1 << 30;

static void pool_release(brov_ctx* c) {
    for (auto& b : c->pool) (void)hipFree(b.first);
    c->pool.clear();
    c->pool_bytes = 0;
}

int brov_malloc(brov_ctx* c, size_t bytes, void** dptr) {
    if (!c || !dptr) return BROV_ERR_ARG;
    DeviceGuard g(c);
    *dptr = nullptr;
    if (bytes == 0) return BROV_OK;
    const size_t want = (bytes + 255) & ~(size_t)255;
    int best = -1;
    for (int i = 0; i < (int)c->pool.size(); ++i)
        if (c->pool[i].second >= want && c->pool[i].second <= 2 * want + ((size_t)1 << 20) && (best < 0 || c->pool[i].second < c->pool[best].second)) best = i;
    if (best >= 0) {
        *dptr = c->pool[best].first;
        c->live[*dptr] = c->pool[best].second;
        c->pool_bytes -= c->pool[best].second;
        c->pool.erase(c->pool.begin() + best);
        return BROV_OK;
    }
    hipError_t e = hipMalloc(dptr, want);
    if (e == hipErrorOutOfMemory && !c->pool.empty()) {        // give the pooled blocks back and try once more
        (void)hipGetLastError();
        (void)hipStreamSynchronize(c->stream);
        pool_release(c);
        e = hipMalloc(dptr, want);
    }
    if (e != hipSuccess) { *dptr = nullptr; return hip_fail(c, e, "hipMalloc"); }
    c->live[*dptr] = want;
    return BROV_OK;
}
int brov_free(brov_ctx* c, void* dptr) {
    if (!c) return BROV_ERR_ARG;
    DeviceGuard g(c);
    lift_cache_touch(c, dptr);
    if (!dptr) return BROV_OK;
    auto it = c->live.find(dptr);
    if (it == c->live.end()) { HIPCK(c, hipFree(dptr)); return BROV_OK; }        // not one of ours (never handed out by brov_malloc)
    const size_t sz = it->second;
    c->live.erase(it);
    // whoever frees a block may still have work queued on it: the next owner's work is ordered behind it only on the same stream
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (sz <= POOL_BLOCK_MAX && c->pool_bytes + sz <= POOL_TOTAL_MAX) {
        c->pool.emplace_back(dptr, sz);
        c->pool_bytes += sz;
        return BROV_OK;
    }
    HIPCK(c, hipFree(dptr));
    return BROV_OK;
}
// ---- host <-> device copies through the ctx's own pinned blocks -----------------------------------------------------------------
// A hipMemcpyAsync from / to PAGEABLE host memory of more than a few pages is served by registering that memory with the driver (a
// "userptr" mapping) for the DMA.  When the memory is released shortly afterwards -- a NumPy temporary, a std::vector of this file --
// its unmapping invalidates the registration, and the kernel driver reacts by EVICTING every queue of the process and restoring them
// some milliseconds later: the next kernel or DMA of the process, whatever it is, waits 12-28 ms (tools/time_first_fit.py,
// profiles/r06_fit_time.txt: a third of a warm KoopmanEDMDc.fit() at the reference's recorded size; gone with single-threaded BLAS only
// because the allocator then stops unmapping; not an SDMA, allocation or CPU-quota effect -- each was ruled out by experiment).
// So no pageable pointer of this size range reaches the runtime: copies between 64 KB and 16 MB go through the two pinned blocks the
// ctx keeps anyway (created once, on first use) -- pack + one DMA, or DMA + unpack.  Smaller ones take the runtime's own small-copy
// buffer; larger uploads are left to the runtime (pinned in place: 55 GB/s measured), larger downloads stream through the two blocks
// alternately.
constexpr size_t STAGE_BLOCK = (size_t)32 << 20;        // bytes per pinned staging block (== UPLOAD_BLOCK below)
constexpr size_t STAGE_MIN = (size_t)64 << 10;
constexpr size_t STAGE_DIRECT_UP = (size_t)16 << 20;

static hipError_t ensure_stage(brov_ctx* c) {
    for (int i = 0; i < 2; ++i) {
        if (!c->h_stage[i]) { hipError_t e = hipHostMalloc((void**)&c->h_stage[i], STAGE_BLOCK, hipHostMallocDefault); if (e != hipSuccess) return e; }
        if (!c->ev_stage[i]) { hipError_t e = hipEventCreateWithFlags(&c->ev_stage[i], hipEventDisableTiming); if (e != hipSuccess) return e; }
    }
    return hipSuccess;
}

// memcpy with a few threads once it is worth starting them (>= 4 MB); a process that cannot start threads copies alone
static void par_memcpy(brov_ctx* c, char* dst, const char* src, size_t bytes);

#define HIPTRY(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return e__; } while (0)
// Host -> device on the ctx stream.  Staged sizes are in place on return; the others are queued like a plain hipMemcpyAsync from pageable
// memory (the runtime has consumed the source when the call returns).
static hipError_t h2d_copy(brov_ctx* c, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return hipSuccess;
    if (bytes < STAGE_MIN || bytes >= STAGE_DIRECT_UP) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream);
    HIPTRY(ensure_stage(c));
    par_memcpy(c, c->h_stage[0], static_cast<const char*>(src), bytes);          // bytes < 16 MB: one block
    HIPTRY(hipMemcpyAsync(dst, c->h_stage[0], bytes, hipMemcpyHostToDevice, c->stream));
    return hipStreamSynchronize(c->stream);                                       // the block is free again
}
// Device -> host on the ctx stream.  Staged sizes are on the host on return; small ones are queued (the caller synchronises, as before).
static hipError_t d2h_copy(brov_ctx* c, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return hipSuccess;
    if (bytes < STAGE_MIN) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream);
    HIPTRY(ensure_stage(c));
    // block i is filled by the DMA engine while block i - 1 is unpacked: at most two copies in flight
    size_t off = 0, out = 0, len[2] = {0, 0};
    int nissued = 0, nunpacked = 0;
    while (out < bytes) {
        while (off < bytes && nissued - nunpacked < 2) {
            const int s_ = nissued & 1;
            len[s_] = std::min(STAGE_BLOCK, bytes - off);
            HIPTRY(hipMemcpyAsync(c->h_stage[s_], static_cast<const char*>(src) + off, len[s_], hipMemcpyDeviceToHost, c->stream));
            HIPTRY(hipEventRecord(c->ev_stage[s_], c->stream));
            off += len[s_];
            ++nissued;
        }
        const int s_ = nunpacked & 1;
        HIPTRY(hipEventSynchronize(c->ev_stage[s_]));
        par_memcpy(c, static_cast<char*>(dst) + out, c->h_stage[s_], len[s_]);
        out += len[s_];
        ++nunpacked;
    }
    return hipSuccess;
}
#undef HIPTRY

int brov_memcpy_h2d(brov_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c || (bytes && (!dst || !src))) return BROV_ERR_ARG;
    DeviceGuard g(c);
    lift_cache_touch(c, dst);
    HIPCK(c, h2d_copy(c, dst, src, bytes));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return BROV_OK;
}
int brov_memcpy_d2h(brov_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c || (bytes && (!dst || !src))) return BROV_ERR_ARG;
    DeviceGuard g(c);
    HIPCK(c, d2h_copy(c, dst, src, bytes));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return BROV_OK;
}
int brov_memset(brov_ctx* c, void* dst, int value, size_t bytes) {
    if (!c || (bytes && !dst)) return BROV_ERR_ARG;
    DeviceGuard g(c);
    lift_cache_touch(c, dst);
    HIPCK(c, hipMemsetAsync(dst, value, bytes, c->stream));
    return BROV_OK;
}

int brov_mem_info(brov_ctx* c, size_t* free_bytes, size_t* total_bytes) {
    if (!c || !free_bytes || !total_bytes) return BROV_ERR_ARG;
    DeviceGuard g(c);
    HIPCK(c, hipMemGetInfo(free_bytes, total_bytes));
    return BROV_OK;
}

// ---- a list of host arrays into one device buffer (fit_multi's X_list / U_list) ------------------------------------------------
// The reference stacks its trajectory list on the host (np.vstack, Koopman/koopmanEDMDc.py:125,140-142).  Here every bag goes
// straight to its place in ONE device buffer: a few host threads pack the bags into a pinned block while the DMA engine moves the
// block before it (two blocks in flight), so the list costs about one pass of host memcpy instead of a stack + a pageable upload.
constexpr size_t UPLOAD_BLOCK = STAGE_BLOCK;            // bytes per pinned staging block
constexpr size_t UPLOAD_GAP_MAX = 4096;                 // a hole of at most this many bytes between two bags' destinations is zero-filled
constexpr size_t UPLOAD_DIRECT_MIN = (size_t)16 << 20;  // a contiguous run this long skips the staging block (the runtime pins it in place)

static int upload_thread_count() {
    unsigned hc = std::thread::hardware_concurrency();
    int n = hc ? (int)hc : 1;
    // containers: the cgroup quota, not the host's core count, is what this process may use
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        long long per = 0;
        if (std::fscanf(f, "%63s %lld", q, &per) == 2 && std::strcmp(q, "max") != 0 && per > 0) {
            const long long qq = std::atoll(q);
            if (qq > 0) { const int lim = (int)((qq + per / 2) / per); if (lim >= 1 && lim < n) n = lim; }
        }
        std::fclose(f);
    }
    if (n > 6) n = 6;                                   // the copy is memory-bound: more threads add nothing
    return n < 1 ? 1 : n;
}

static void par_memcpy(brov_ctx* c, char* dst, const char* src, size_t bytes) {
    if (c->upload_threads == 0) c->upload_threads = upload_thread_count();
    const int nt = (bytes >= ((size_t)4 << 20)) ? c->upload_threads : 1;
    if (nt <= 1) { std::memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    const size_t per = ((bytes + nt - 1) / nt + 63) & ~(size_t)63;
    size_t started = per;
    try {
        for (int t = 1; t < nt; ++t) {
            const size_t lo = std::min(bytes, per * t), hi = std::min(bytes, per * (t + 1));
            th.emplace_back([=]() { if (hi > lo) std::memcpy(dst + lo, src + lo, hi - lo); });
            started = hi;
        }
    } catch (...) {
        c->upload_threads = 1;
    }
    std::memcpy(dst, src, std::min(bytes, per));
    if (started < bytes) std::memcpy(dst + started, src + started, bytes - started);
    for (auto& t : th) t.join();
}

int brov_upload_bags(brov_ctx* c, int64_t nbags, const double* const* bag_ptrs, const int64_t* bag_rows, const int64_t* dst_rows,
                     int cols, double* d_dst) {
    if (!c || nbags < 0 || cols < 1 || (nbags && (!bag_ptrs || !bag_rows || !dst_rows || !d_dst)))
        return fail(c, BROV_ERR_ARG, "brov_upload_bags: bad argument");
    struct Piece { const char* src; size_t bytes; size_t dst; };        // dst = byte offset in d_dst
    std::vector<Piece> pieces;
    pieces.reserve((size_t)nbags);
    const size_t rowb = (size_t)cols * 8;
    for (int64_t b = 0; b < nbags; ++b) {
        if (bag_rows[b] < 0 || dst_rows[b] < 0) return fail(c, BROV_ERR_ARG, "brov_upload_bags: negative row count or offset");
        if (bag_rows[b] == 0) continue;
        if (!bag_ptrs[b]) return fail(c, BROV_ERR_ARG, "brov_upload_bags: null bag");
        const Piece pc = {reinterpret_cast<const char*>(bag_ptrs[b]), (size_t)bag_rows[b] * rowb, (size_t)dst_rows[b] * rowb};
        if (!pieces.empty()) {
            Piece& q = pieces.back();
            if (pc.dst < q.dst + q.bytes) return fail(c, BROV_ERR_ARG, "brov_upload_bags: destinations must ascend without overlap");
            if (q.src + q.bytes == pc.src && q.dst + q.bytes == pc.dst) { q.bytes += pc.bytes; continue; }   // views of one array
        }
        pieces.push_back(pc);
    }
    if (pieces.empty()) return BROV_OK;
    DeviceGuard g(c);
    lift_cache_touch(c, d_dst);
    HIPCK(c, ensure_stage(c));
    if (c->upload_threads == 0) c->upload_threads = upload_thread_count();
    // a block = consecutive (parts of) pieces whose destinations are contiguous up to small holes: one DMA per block
    struct Part { const char* src; size_t bytes; size_t at; };          // at = offset in the staging block; src == nullptr: zeros
    std::vector<Part> parts;
    size_t ip = 0, done_in_piece = 0;
    int slot = 0;
    bool used[2] = {false, false};
    while (ip < pieces.size()) {
        if (done_in_piece == 0 && pieces[ip].bytes >= UPLOAD_DIRECT_MIN) {     // e.g. a list of views that tile one big array
            HIPCK(c, hipMemcpyAsync(reinterpret_cast<char*>(d_dst) + pieces[ip].dst, pieces[ip].src, pieces[ip].bytes, hipMemcpyHostToDevice, c->stream));
            ++ip;
            continue;
        }
        parts.clear();
        const size_t dst0 = pieces[ip].dst + done_in_piece;
        size_t fill = 0;
        while (ip < pieces.size() && fill < UPLOAD_BLOCK) {
            const Piece& pc = pieces[ip];
            if (done_in_piece == 0 && pc.bytes >= UPLOAD_DIRECT_MIN && fill > 0) break;     // goes direct, after this block
            const size_t want_at = pc.dst + done_in_piece - dst0;      // where this piece's next byte belongs in the block
            if (want_at > fill) {                                       // a hole before it
                if (want_at - fill > UPLOAD_GAP_MAX || want_at >= UPLOAD_BLOCK) break;
                parts.push_back({nullptr, want_at - fill, fill});
                fill = want_at;
            }
            const size_t take = std::min(pc.bytes - done_in_piece, UPLOAD_BLOCK - fill);
            parts.push_back({pc.src + done_in_piece, take, fill});
            fill += take;
            done_in_piece += take;
            if (done_in_piece == pc.bytes) { ++ip; done_in_piece = 0; }
        }
        char* stage = c->h_stage[slot];
        if (used[slot]) HIPCK(c, hipEventSynchronize(c->ev_stage[slot]));      // the DMA that last read this block
        // pack: the parts are cut into equal byte ranges, one per thread
        const int nt = (fill >= ((size_t)4 << 20)) ? c->upload_threads : 1;
        auto pack = [&](size_t lo, size_t hi) {
            for (const Part& pt : parts) {
                const size_t a0 = std::max(lo, pt.at), a1 = std::min(hi, pt.at + pt.bytes);
                if (a0 >= a1) continue;
                if (pt.src) std::memcpy(stage + a0, pt.src + (a0 - pt.at), a1 - a0);
                else std::memset(stage + a0, 0, a1 - a0);
            }
        };
        if (nt <= 1) pack(0, fill);
        else {
            std::vector<std::thread> th;
            const size_t per = ((fill + nt - 1) / nt + 63) & ~(size_t)63;
            size_t started = per;                      // bytes [0, started) are this thread's and the helpers'
            try {
                for (int t = 1; t < nt; ++t) {
                    th.emplace_back(pack, std::min(fill, per * t), std::min(fill, per * (t + 1)));
                    started = std::min(fill, per * (t + 1));
                }
            } catch (...) {                            // no more threads to be had (a process limit): this thread packs the rest itself
                c->upload_threads = 1;
            }
            pack(0, std::min(fill, per));
            if (started < fill) pack(started, fill);
            for (auto& t : th) t.join();
        }
        HIPCK(c, hipMemcpyAsync(reinterpret_cast<char*>(d_dst) + dst0, stage, fill, hipMemcpyHostToDevice, c->stream));
        HIPCK(c, hipEventRecord(c->ev_stage[slot], c->stream));
        used[slot] = true;
        slot ^= 1;
    }
    HIPCK(c, hipStreamSynchronize(c->stream));      // the staging blocks may be reused by the next call; the data is in place on return
    return BROV_OK;
}

// ---- RHS ---------------------------------------------------------------------------------------
// The reference's scripts call dynamics() once per vehicle and step (fossen/BlueROV2.py:357-400 from the loops at
// training/train_tank_brov2_full_comparison.py:453-466): one launch whose cost is latency.  Up to SMALL_B vehicles go through a
// pinned, device-mapped staging block: the host packs the operands, the kernel reads and writes that block over the bus,
// the host unpacks -- one launch and one synchronisation, no copy calls (five of them cost more than the launch).
constexpr int64_t SMALL_B = 16;
constexpr size_t IO_FLAGS = SMALL_B * (13 + 8 + 24 + 13);          // offset (in doubles) of the SMALL_B completion flags
constexpr size_t IO_DOUBLES = IO_FLAGS + SMALL_B + 64;

// Wait for the per-call kernel: every row's thread stores the call's sequence number to its flag after its results (release,
// system scope), the host spins on the flags (acquire) -- 3-4 us less than hipStreamSynchronize.  The spin is bounded by the
// CLOCK, not by an iteration count (`pause` costs 10-140 cycles depending on the CPU): after 2 ms without the flags -- long
// prior work on the stream, a faulted kernel, a debugger -- the stream is synchronised instead and its status reported.  A ctx
// bound to a caller's stream (brov_set_stream) never spins: other work may be queued in front of the launch.
static int wait_flags(brov_ctx* c, int64_t B, unsigned long long seq, const char* what) {
    volatile unsigned long long* f = reinterpret_cast<volatile unsigned long long*>(c->h_io + IO_FLAGS);
    if (c->stream != nullptr) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return hip_fail(c, e, what);
        for (int64_t b = 0; b < B; ++b)
            if (__atomic_load_n(const_cast<unsigned long long*>(&f[b]), __ATOMIC_ACQUIRE) != seq) return fail(c, BROV_ERR_HIP, what);
        return BROV_OK;
    }
    timespec t0{};
    bool timed = false;
    for (int64_t b = 0; b < B; ++b) {
        long spins = 0;
        while (__atomic_load_n(const_cast<unsigned long long*>(&f[b]), __ATOMIC_ACQUIRE) != seq) {
            bool give_up = false;
            if ((++spins & 0x3FF) == 0) {                 // look at the clock every 1024 polls
                timespec t1{};
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if (!timed) { t0 = t1; timed = true; }
                give_up = (t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > 2000000L;
            }
            if (give_up) {
                hipError_t e = hipStreamSynchronize(c->stream);
                if (e != hipSuccess) return hip_fail(c, e, what);
                if (__atomic_load_n(const_cast<unsigned long long*>(&f[b]), __ATOMIC_ACQUIRE) != seq) return fail(c, BROV_ERR_HIP, what);
                break;
            }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
    }
    // the runtime retires its per-launch bookkeeping when it is asked about the stream: do that now and then
    if ((seq & 0x3FF) == 0) { hipError_t e = hipStreamSynchronize(c->stream); if (e != hipSuccess) return hip_fail(c, e, what); }
    return BROV_OK;
}
// Wait for one M-step of the Lloyd loop: its tail stores the iteration's four statistics and then `seq` into the pinned block
// (release, system scope); the host polls (acquire).  An event in the stream would do, and cost the kernel behind it 6 us of a
// 0.44 ms iteration (kernel trace, round 5).  Bounded by the clock: half a second without the number -- a faulted kernel, a
// debugger -- and the stream is synchronised instead and its status reported.
static int wait_stats(brov_ctx* c, double seq) {
    unsigned long long want;
    std::memcpy(&want, &seq, 8);
    const unsigned long long* w = reinterpret_cast<const unsigned long long*>(c->h_stats + 4);
    timespec t0{};
    bool timed = false;
    long spins = 0;
    while (__atomic_load_n(w, __ATOMIC_ACQUIRE) != want) {
        if ((++spins & 0x3FF) == 0) {
            timespec t1{};
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if (!timed) { t0 = t1; timed = true; }
            if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > 500000000L) {
                hipError_t e = hipStreamSynchronize(c->stream);
                if (e != hipSuccess) return hip_fail(c, e, "edmdc_kmeans_lloyd");
                if (__atomic_load_n(w, __ATOMIC_ACQUIRE) != want) return fail(c, BROV_ERR_HIP, "edmdc_kmeans_lloyd: the M-step's statistics never arrived");
                break;
            }
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    return BROV_OK;
}
static int ensure_io(brov_ctx* c) {
    if (c->h_io) return BROV_OK;
    HIPCK(c, hipHostMalloc((void**)&c->h_io, IO_DOUBLES * sizeof(double), hipHostMallocMapped));
    std::memset(c->h_io, 0, IO_DOUBLES * sizeof(double));
    hipError_t e = hipHostGetDevicePointer((void**)&c->d_io, c->h_io, 0);
    if (e != hipSuccess) { (void)hipHostFree(c->h_io); c->h_io = nullptr; return hip_fail(c, e, "hipHostGetDevicePointer"); }
    return BROV_OK;
}

int brov_rhs(brov_ctx* c, int model, int64_t B, const double* x, const double* u, double dt, double* lag_io, double* xdot) {
    if (!c || !model_ok(model) || model_is_di_h(model) || B < 0 || (B && (!x || !u || !xdot)))
        return fail(c, BROV_ERR_ARG, "brov_rhs: bad argument (the double-integrator models have rollouts only)");
    if (B == 0) return BROV_OK;
    DeviceGuard g(c);
    const DevParams* dp;
    int rc = get_dp(c, dt, &dp);
    if (rc) return rc;
    const int nx = NX(model), nu = NU(model);
    const bool lag = lag_io && model == BROV_THRUSTER_EULER;
    if (B <= SMALL_B) {
        rc = ensure_io(c);
        if (rc) return rc;
        const size_t ox = 0, ou = ox + B * nx, ol = ou + B * nu, od = ol + (lag ? B * 24 : 0);
        std::memcpy(c->h_io + ox, x, B * nx * 8);
        std::memcpy(c->h_io + ou, u, B * nu * 8);
        if (lag) std::memcpy(c->h_io + ol, lag_io, B * 24 * 8);
        const bool spin = !c->timing;               // with kernel timing on, the events need the stream synchronised anyway
        const unsigned long long seq = ++c->io_seq;
        {
            CallTimer t(c);
            HIPCK(c, launch_rhs(c->stream, *dp, model, B, c->d_io + ox, c->d_io + ou, lag ? c->d_io + ol : nullptr, c->d_io + od,
                                spin ? reinterpret_cast<unsigned long long*>(c->d_io + IO_FLAGS) : nullptr, seq));
        }
        if (spin) { rc = wait_flags(c, B, seq, "brov_rhs"); if (rc) return rc; }
        else HIPCK(c, hipStreamSynchronize(c->stream));
        std::memcpy(xdot, c->h_io + od, B * nx * 8);
        if (lag) std::memcpy(lag_io, c->h_io + ol, B * 24 * 8);
        return BROV_OK;
    }
    Arena a(c);
    rc = a.reserve(Arena::al(B * nx * 8) * 2 + Arena::al(B * nu * 8) + Arena::al(B * 24 * 8));
    if (rc) return rc;
    double* dx = a.take<double>(B * nx);
    double* du = a.take<double>(B * nu);
    double* dxd = a.take<double>(B * nx);
    double* dl = lag ? a.take<double>(B * 24) : nullptr;
    HIPCK(c, h2d_copy(c, dx, x, B * nx * 8));
    HIPCK(c, h2d_copy(c, du, u, B * nu * 8));
    if (lag) HIPCK(c, h2d_copy(c, dl, lag_io, B * 24 * 8));
    {
        CallTimer t(c);
        HIPCK(c, launch_rhs(c->stream, *dp, model, B, dx, du, dl, dxd));
    }
    HIPCK(c, d2h_copy(c, xdot, dxd, B * nx * 8));
    if (lag) HIPCK(c, d2h_copy(c, lag_io, dl, B * 24 * 8));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return BROV_OK;
}

int brov_thruster_forces(brov_ctx* c, int64_t B, const double* u, double dt, double* lag_io, double* tau) {
    if (!c || B < 0 || (B && (!u || !tau || !lag_io))) return fail(c, BROV_ERR_ARG, "brov_thruster_forces: bad argument");
    if (B == 0) return BROV_OK;
    DeviceGuard g(c);
    const DevParams* dp;
    int rc = get_dp(c, dt, &dp);
    if (rc) return rc;
    if (B <= SMALL_B) {
        rc = ensure_io(c);
        if (rc) return rc;
        const size_t ou = 0, ol = ou + B * 8, ot = ol + B * 24;
        std::memcpy(c->h_io + ou, u, B * 8 * 8);
        std::memcpy(c->h_io + ol, lag_io, B * 24 * 8);
        const bool spin = !c->timing;
        const unsigned long long seq = ++c->io_seq;
        {
            CallTimer t(c);
            HIPCK(c, launch_thruster_forces(c->stream, *dp, B, c->d_io + ou, c->d_io + ol, c->d_io + ot,
                                            spin ? reinterpret_cast<unsigned long long*>(c->d_io + IO_FLAGS) : nullptr, seq));
        }
        if (spin) { rc = wait_flags(c, B, seq, "brov_thruster_forces"); if (rc) return rc; }
        else HIPCK(c, hipStreamSynchronize(c->stream));
        std::memcpy(tau, c->h_io + ot, B * 6 * 8);
        std::memcpy(lag_io, c->h_io + ol, B * 24 * 8);
        return BROV_OK;
    }
    Arena a(c);
    rc = a.reserve(Arena::al(B * 8 * 8) + Arena::al(B * 24 * 8) + Arena::al(B * 6 * 8));
    if (rc) return rc;
    double* du = a.take<double>(B * 8);
    double* dl = a.take<double>(B * 24);
    double* dt_ = a.take<double>(B * 6);
    HIPCK(c, h2d_copy(c, du, u, B * 8 * 8));
    HIPCK(c, h2d_copy(c, dl, lag_io, B * 24 * 8));
    {
        CallTimer t(c);
        HIPCK(c, launch_thruster_forces(c->stream, *dp, B, du, dl, dt_));
    }
    HIPCK(c, d2h_copy(c, tau, dt_, B * 6 * 8));
    HIPCK(c, d2h_copy(c, lag_io, dl, B * 24 * 8));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return BROV_OK;
}

// ---- rollout -------------------------------------------------------------------------------------
static int rollout_args_ok(brov_ctx* c, int model, int integ, int lag_mode, int layout, int64_t B, int64_t T, int64_t stride,
                           const void* x0, const void* U, const void* traj) {
    if (!c) return BROV_ERR_ARG;
    if (!model_ok(model) || (integ != BROV_EULER && integ != BROV_RK4) || (lag_mode != BROV_LAG_PER_CALL && lag_mode != BROV_LAG_PER_STEP) ||
        (layout != BROV_LAYOUT_BTU && layout != BROV_LAYOUT_TUB && layout != BROV_LAYOUT_TPB) || B < 0 || T < 0)
        return fail(c, BROV_ERR_ARG, "brov_rollout: bad enum or negative size");
    if (B && (!x0 || (T && !U))) return fail(c, BROV_ERR_ARG, "brov_rollout: NULL input");
    if (traj && stride < 1) return fail(c, BROV_ERR_ARG, "brov_rollout: traj_stride must be >= 1");
    return BROV_OK;
}

int brov_set_di_gains(brov_ctx* c, int nu, const double* K_lin, const double* K_ang) {
    if (!c || (nu != 6 && nu != 8) || !K_lin || !K_ang) return fail(c, BROV_ERR_ARG, "brov_set_di_gains: nu must be 6 or 8");
    DeviceGuard g(c);
    FastParams f;
    std::memset(&f, 0, sizeof f);
    for (int i = 0; i < nu; ++i)
        for (int k = 0; k < 3; ++k) { f.Tm[k][i] = K_lin[3 * i + k]; f.Tm[3 + k][i] = K_ang[3 * i + k]; }
    if (!c->d_fp_di) HIPCK(c, hipMalloc((void**)&c->d_fp_di, sizeof(FastParams)));
    HIPCK(c, h2d_copy(c, c->d_fp_di, &f, sizeof f));
    HIPCK(c, hipStreamSynchronize(c->stream));
    c->di_set = true;
    return BROV_OK;
}

int brov_set_btu_staging(brov_ctx* c, int mode) {
    if (!c || mode < 0 || mode > 2) return fail(c, BROV_ERR_ARG, "brov_set_btu_staging: mode must be 0, 1 or 2");
    c->btu_staging = mode;
    return BROV_OK;
}

int brov_rollout_dev(brov_ctx* c, int model, int integ, int lag_mode, int layout, int64_t B, int64_t T, double dt,
                     const double* d_x0, const double* d_U, double* d_lag_io, double* d_traj, int64_t stride, double* d_xT) {
    int rc = rollout_args_ok(c, model, integ, lag_mode, layout, B, T, stride, d_x0, d_U, d_traj);
    if (rc) return rc;
    if (B == 0) return BROV_OK;
    DeviceGuard g(c);
    const DevParams* dp;
    rc = get_dp(c, dt, &dp);
    if (rc) return rc;
    if (model_is_di_h(model) && !c->di_set) return fail(c, BROV_ERR_ARG, "double-integrator model: call brov_set_di_gains first");
    CallTimer t(c);
    HIPCK(c, launch_rollout(c->stream, model_is_di_h(model) ? c->d_fp_di : c->d_fp, model, integ, lag_mode, layout, B, T, dt, d_x0, d_U,
                            d_lag_io, d_traj, d_traj ? stride : 1, d_xT,
                            c->btu_staging | ((c->fp.has_current || c->fp.has_xy || c->fp.tm_dense || c->fp.obs_bad) ? 4 : 0) |
                                (c->single_lane ? 8 : 0)));
    return BROV_OK;
}

int brov_rollout(brov_ctx* c, int model, int integ, int lag_mode, int layout, int64_t B, int64_t T, double dt,
                 const double* x0, const double* U, double* lag_io, double* traj, int64_t stride, double* xT) {
    int rc = rollout_args_ok(c, model, integ, lag_mode, layout, B, T, stride, x0, U, traj);
    if (rc) return rc;
    if (B == 0) return BROV_OK;
    DeviceGuard g(c);
    const int nx = NX(model), nu = NU(model);
    const bool lag = lag_io && model == BROV_THRUSTER_EULER;
    const int64_t rows = traj ? T / stride + 1 : 0;
    const int nuw = layout == BROV_LAYOUT_TPB ? (nu + 1) / 2 * 2 : nu;     // TPB pads odd channel counts to pairs
    const int nxw = layout == BROV_LAYOUT_TPB ? (nx + 1) / 2 * 2 : nx;
    Arena a(c);
    rc = a.reserve(Arena::al(B * nx * 8) * 2 + Arena::al((size_t)B * T * nuw * 8) + Arena::al(B * 24 * 8) + Arena::al((size_t)B * rows * nxw * 8));
    if (rc) return rc;
    double* dx0 = a.take<double>(B * nx);
    double* dU = a.take<double>((size_t)B * T * nuw);
    double* dxT = a.take<double>(B * nx);
    double* dl = lag ? a.take<double>(B * 24) : nullptr;
    double* dtr = traj ? a.take<double>((size_t)B * rows * nxw) : nullptr;
    HIPCK(c, h2d_copy(c, dx0, x0, B * nx * 8));
    if (T) HIPCK(c, h2d_copy(c, dU, U, (size_t)B * T * nuw * 8));
    if (lag) HIPCK(c, h2d_copy(c, dl, lag_io, B * 24 * 8));
    rc = brov_rollout_dev(c, model, integ, lag_mode, layout, B, T, dt, dx0, dU, dl, dtr, stride, dxT);
    if (rc) return rc;
    if (xT) HIPCK(c, d2h_copy(c, xT, dxT, B * nx * 8));
    if (lag) HIPCK(c, d2h_copy(c, lag_io, dl, B * 24 * 8));
    if (traj) HIPCK(c, d2h_copy(c, traj, dtr, (size_t)B * rows * nxw * 8));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return BROV_OK;
}

// ---- sliding-window endpoint error ----------------------------------------------------------------
// Phi = Ad^(samples per window), by repeated squaring on the host
static void lag_window_phi(const DevParams& dp, int64_t samples, double Phi[9]) {
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Bq[9];
    std::memcpy(Bq, dp.lag_A[0], sizeof Bq);
    while (samples > 0) {
        if (samples & 1) matmul(3, R, Bq, R);
        matmul(3, Bq, Bq, Bq);
        samples >>= 1;
    }
    std::memcpy(Phi, R, sizeof R);
}

static int window_dev_impl(brov_ctx* c, int model, int integ, int64_t N, int64_t H, double dt, const double* dX, const double* dU,
                           int carry, double* d_total, double* d_se, Arena& a) {
    const DevParams* dp;
    int rc = get_dp(c, dt, &dp);
    if (rc) return rc;
    const int64_t nwin = N - H;
    double *d_resp = nullptr, *d_start = nullptr, *d_phi = nullptr;
    if (model == BROV_THRUSTER_EULER && carry) {
        d_resp = a.take<double>(nwin * 18);
        d_start = a.take<double>((nwin + (nwin + window_scan_chunk() - 1) / window_scan_chunk()) * 18);   // start states + chunk states
        d_phi = a.take<double>(32);
        double Phi[18];                             // Phi, Phi^chunk (blocked scan of rollout.hip)
        const int64_t spw = H * (integ == BROV_RK4 ? 4 : 1);
        lag_window_phi(*dp, spw, Phi);
        lag_window_phi(*dp, spw * window_scan_chunk(), Phi + 9);
        HIPCK(c, h2d_copy(c, d_phi, Phi, sizeof Phi));
        HIPCK(c, hipStreamSynchronize(c->stream));   // Phi is a stack temporary
    }
    if (model_is_di_h(model) && !c->di_set) return fail(c, BROV_ERR_ARG, "double-integrator model: call brov_set_di_gains first");
    CallTimer t(c);
    HIPCK(c, launch_window_endpoint(c->stream, model_is_di_h(model) ? c->d_fp_di : c->d_fp, model, integ, N, H, dt, dX, dU, carry, d_phi,
                                    d_resp, d_start, d_se, d_total));
    return BROV_OK;
}

int brov_window_endpoint_se_dev(brov_ctx* c, int model, int integ, int64_t N, int64_t H, double dt, const double* d_X,
                                const double* d_U, int carry_lag, double* d_se_total, double* d_per_window) {
    if (!c || !model_ok(model) || (integ != BROV_EULER && integ != BROV_RK4) || N < 0 || H < 0 || !d_se_total)
        return fail(c, BROV_ERR_ARG, "brov_window_endpoint_se_dev: bad argument");
    DeviceGuard g(c);
    const int64_t nwin = N - H;
    if (nwin <= 0) { HIPCK(c, hipMemsetAsync(d_se_total, 0, 8, c->stream)); return BROV_OK; }
    if (!d_X || !d_U || !d_per_window) return fail(c, BROV_ERR_ARG, "brov_window_endpoint_se_dev: NULL array");
    Arena a(c);
    int rc = a.reserve(Arena::al(nwin * 24 * 8) * 2 + Arena::al((nwin / window_scan_chunk() + 2) * 18 * 8) + 4096);
    if (rc) return rc;
    return window_dev_impl(c, model, integ, N, H, dt, d_X, d_U, carry_lag, d_se_total, d_per_window, a);
}

int brov_window_endpoint_se(brov_ctx* c, int model, int integ, int64_t N, int64_t H, double dt, const double* X, const double* U,
                            int carry_lag, double* se_total, double* per_window) {
    if (!c || !model_ok(model) || (integ != BROV_EULER && integ != BROV_RK4) || N < 0 || H < 0 || !se_total)
        return fail(c, BROV_ERR_ARG, "brov_window_endpoint_se: bad argument");
    const int64_t nwin = N - H;
    if (nwin <= 0) { *se_total = 0.0; return BROV_OK; }
    if (!X || !U) return fail(c, BROV_ERR_ARG, "brov_window_endpoint_se: NULL array");
    DeviceGuard g(c);
    const int nx = NX(model), nu = NU(model);
    Arena a(c);
    int rc = a.reserve(Arena::al(N * nx * 8) + Arena::al(N * nu * 8) + Arena::al(nwin * 8) + Arena::al(nwin * 24 * 8) * 2 +
                       Arena::al((nwin / window_scan_chunk() + 2) * 18 * 8) + 4096);
    if (rc) return rc;
    double* dX = a.take<double>(N * nx);
    double* dU = a.take<double>(N * nu);
    double* dse = a.take<double>(nwin);
    double* dtot = a.take<double>(8);
    HIPCK(c, h2d_copy(c, dX, X, N * nx * 8));
    HIPCK(c, h2d_copy(c, dU, U, N * nu * 8));
    rc = window_dev_impl(c, model, integ, N, H, dt, dX, dU, carry_lag, dtot, dse, a);
    if (rc) return rc;
    HIPCK(c, d2h_copy(c, se_total, dtot, 8));
    if (per_window) HIPCK(c, d2h_copy(c, per_window, dse, nwin * 8));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return BROV_OK;
}

// ---- synthetic controls -------------------------------------------------------------------------------
int brov_fill_controls_dev(brov_ctx* c, int layout, int dist, int64_t B, int64_t T, int nu, uint64_t seed, int64_t b0,
                           int64_t T_total, const double* scale_host, double* d_U) {
    if (!c || (layout != BROV_LAYOUT_BTU && layout != BROV_LAYOUT_TUB && layout != BROV_LAYOUT_TPB) || (dist != BROV_DIST_IID_UNIFORM && dist != BROV_DIST_AR1) ||
        B < 0 || T < 0 || nu < 1 || nu > 8 || b0 < 0 || T_total < T || (B && T && !d_U))
        return fail(c, BROV_ERR_ARG, "brov_fill_controls_dev: bad argument");
    DeviceGuard g(c);
    CallTimer t(c);
    double sc[8];
    for (int j = 0; j < 8; ++j) sc[j] = (scale_host && j < nu) ? scale_host[j] : 1.0;
    HIPCK(c, launch_fill_controls(c->stream, layout, dist, B, T, nu, seed, b0, T_total, sc, d_U));
    return BROV_OK;
}

// ---- EDMDc -------------------------------------------------------------------------------------------
int edmdc_set_chunk_rows(brov_ctx* c, int64_t rows) {
    if (!c || rows < 64) return fail(c, BROV_ERR_ARG, "edmdc_set_chunk_rows: rows must be >= 64");
    c->chunk_rows = (rows + 3) / 4 * 4;
    return BROV_OK;
}
// the one shape predicate of the EDMDc entry points (the decomposition queries included: the sanitizer sweep of round 4 found them
// walking shapes -- r = 65 -- that the compute entry points refuse)
static bool edmdc_shape_supported(int n, int r, int k) {
    return !(n < 1 || n > 16 || r < 0 || r > 64 || k < 1 || k > 65535 * 16 || ((n + r + 15) / 16 * 16) > 256);
}

static int edmdc_shape_ok(brov_ctx* c, int n, int r, int k) {
    if (!c) return BROV_ERR_ARG;
    if (!edmdc_shape_supported(n, r, k))
        return fail(c, BROV_ERR_ARG, "edmdc: unsupported shape (need 1<=n<=16, 0<=r<=64, k>=1)");
    return BROV_OK;
}

int edmdc_lift(brov_ctx* c, int64_t N, int n, int k, double gamma, const double* X, const double* C, double* Z) {
    int rc = edmdc_shape_ok(c, n, 0, k);
    if (rc) return rc;
    if (N < 0 || (N && (!X || !Z)) || !C) return fail(c, BROV_ERR_ARG, "edmdc_lift: bad argument");
    if (N == 0) return BROV_OK;
    DeviceGuard g(c);
    Arena a(c);
    rc = a.reserve(Arena::al(N * n * 8) + Arena::al((size_t)k * n * 8) + Arena::al((size_t)N * (n + k) * 8));
    if (rc) return rc;
    double* dX = a.take<double>(N * n);
    double* dC = a.take<double>((size_t)k * n);
    double* dZ = a.take<double>((size_t)N * (n + k));
    HIPCK(c, h2d_copy(c, dX, X, N * n * 8));
    HIPCK(c, h2d_copy(c, dC, C, (size_t)k * n * 8));
    {
        CallTimer t(c);
        HIPCK(c, launch_lift_ref(c->stream, N, n, k, gamma, dX, dC, dZ));
    }
    HIPCK(c, d2h_copy(c, Z, dZ, (size_t)N * (n + k) * 8));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return BROV_OK;
}

static int ensure_tasks(brov_ctx* c, const EdmdcShape& s, int mode) {
    if (c->ntasks[mode] == 0 || c->task_shape[mode].n != s.n || c->task_shape[mode].r != s.r || c->task_shape[mode].k != s.k) {
        // the table grows with the shape (k = 512: 81 tasks, k = 1024: 295): sized from the task list, not a fixed cap
        const size_t need = gram_task_bytes(s, mode);
        if (need > c->tasks_cap[mode]) {
            HIPCK(c, hipStreamSynchronize(c->stream));
            if (c->d_tasks[mode]) (void)hipFree(c->d_tasks[mode]);
            c->d_tasks[mode] = nullptr; c->tasks_cap[mode] = 0; c->ntasks[mode] = 0;
            HIPCK(c, hipMalloc(&c->d_tasks[mode], need));
            c->tasks_cap[mode] = need;
        }
        int nt = 0;
        hipError_t e = upload_gram_tasks(c->stream, s, mode, c->d_tasks[mode], c->tasks_cap[mode], &nt);
        if (e != hipSuccess) return hip_fail(c, e, "upload_gram_tasks");
        HIPCK(c, hipStreamSynchronize(c->stream));   // source vector is a temporary
        c->ntasks[mode] = nt;
        c->task_shape[mode] = s;
    }
    return BROV_OK;
}

// one cache slot = the lifted rows of a chunk (+ 8 rows of padding: the Gram kernel prefetches past the end) and their pair weights
static size_t lift_slot_bytes(int64_t chunk, int width) { return ((size_t)(chunk + 8) * width + (size_t)(chunk + 8)) * 8; }

static int ensure_partial(brov_ctx* c, size_t pdoubles) {
    if (pdoubles * 8 > c->partial_cap) {
        HIPCK(c, hipStreamSynchronize(c->stream));
        if (c->d_partial) (void)hipFree(c->d_partial);
        c->d_partial = nullptr; c->partial_cap = 0;
        HIPCK(c, hipMalloc((void**)&c->d_partial, pdoubles * 8));
        c->partial_cap = pdoubles * 8;
    }
    return BROV_OK;
}

// How the rows of X / U split into bags.  Uniform: nbags bags of L pairs, bag b = state rows b xs .. b xs + L and input rows
// b us .. b us + L - 1 (edmdc_gram_dev).  Ragged: the bags' rows one after the other, bag b = rows offsets[b] .. offsets[b + 1] - 1 of X
// and of U (row-aligned: U's row at the last state of a bag is never read, the `U[:-1]` of Koopman/koopmanEDMDc.py:138) -- the trajectory
// list fit_multi takes (edmdc_gram_ragged_dev).  Either way a lifted chunk is `chunk` consecutive rows and a pair never crosses a bag.
struct BagLayout {
    int64_t nbags = 0, L = 0, xs = 0, us = 0;
    int64_t total_rows = 0;
    const int64_t* offsets_host = nullptr;      // ragged: [nbags + 1], validated by bag_layout_ragged
    uint64_t hash = 0;                          // ragged: fingerprint of the offsets (part of the lift-cache key)
    bool ragged() const { return offsets_host != nullptr; }
    size_t scratch_bytes() const { return ragged() ? Arena::al((size_t)(nbags + 1) * 8) + Arena::al((size_t)total_rows + 16) : 0; }
};

static void bag_layout_uniform(int64_t nbags, int64_t L, int64_t xs, int64_t us, BagLayout* bl) {
    if (nbags <= 1) { xs = L + 1; us = L; }
    bl->nbags = nbags; bl->L = L; bl->xs = xs; bl->us = us;
    bl->total_rows = nbags > 0 ? (nbags - 1) * xs + L + 1 : 0;
}

static int bag_layout_ragged(brov_ctx* c, int64_t nbags, const int64_t* off, BagLayout* bl, const char* who) {
    if (nbags < 0 || !off) return fail(c, BROV_ERR_ARG, std::string(who) + ": bad bag list");
    if (off[0] != 0) return fail(c, BROV_ERR_ARG, std::string(who) + ": bag_offsets[0] must be 0");
    uint64_t h = 1469598103934665603ull;
    for (int64_t b = 0; b <= nbags; ++b) {
        if (b && off[b] < off[b - 1]) return fail(c, BROV_ERR_ARG, std::string(who) + ": bag_offsets must not decrease");
        h = (h ^ (uint64_t)off[b]) * 1099511628211ull;
    }
    bl->nbags = nbags;
    bl->xs = (int64_t)1 << 62;                  // one "bag" longer than any data: every row below total_rows is a state row,
    bl->L = bl->xs - 1;                         // the pair flags say where the real bags end
    bl->us = 0;
    bl->total_rows = off[nbags];
    bl->offsets_host = off;
    bl->hash = h | 1;
    return BROV_OK;
}

// the pair flags of a ragged bag list in the call's arena (nullptr for a uniform layout)
static int bag_pairflags(brov_ctx* c, Arena& a, const BagLayout& bl, const unsigned char** out) {
    *out = nullptr;
    if (!bl.ragged()) return BROV_OK;
    int64_t* d_off = a.take<int64_t>((size_t)bl.nbags + 1);
    unsigned char* pf = a.take<unsigned char>((size_t)bl.total_rows + 16);
    HIPCK(c, h2d_copy(c, d_off, bl.offsets_host, (size_t)(bl.nbags + 1) * 8));
    HIPCK(c, launch_bag_pairflags(c->stream, bl.nbags, d_off, bl.total_rows, pf));
    HIPCK(c, hipStreamSynchronize(c->stream));  // the offsets are the caller's (host) array: it may go away when the call returns
    *out = pf;
    return BROV_OK;
}

static int gram_core(brov_ctx* c, int n, int r, int k, double gamma, const double* d_C, const BagLayout& bl,
                     const double* d_X, const double* d_U, int accumulate, double* d_GtG, double* d_GtY) {
    const int mode = d_GtY ? 0 : 2;                     // no G^T Y wanted (fit()'s Gram pass): the staircase over the G tiles alone
    DeviceGuard g(c);
    const EdmdcShape s = edmdc_shape(n, r, k);
    int rc = ensure_tasks(c, s, mode);
    if (rc) return rc;
    const int64_t L = bl.L, xs = bl.xs, us = bl.us;
    const int64_t total_rows = bl.total_rows;
    const int64_t total_pairs_rows = total_rows > 0 ? total_rows - 1 : 0;   // rows that can start a pair
    int nslab = 0, ntasks = 0;
    const size_t pdoubles = gram_partial_doubles(s, mode, &ntasks, &nslab);
    rc = ensure_partial(c, pdoubles);
    if (rc) return rc;
    int64_t chunk = c->chunk_rows;
    if (chunk > (total_pairs_rows + 3) / 4 * 4) chunk = (total_pairs_rows + 3) / 4 * 4;
    if (chunk < 4) chunk = 4;
    Arena a(c);
    rc = a.reserve(Arena::al((size_t)(chunk + 8) * s.width * 8) + Arena::al((chunk + 8) * 8) + bl.scratch_bytes());
    if (rc) return rc;
    double* dZ = a.take<double>((size_t)(chunk + 8) * s.width);
    double* dw = a.take<double>(chunk + 8);
    const unsigned char* pf = nullptr;
    rc = bag_pairflags(c, a, bl, &pf);
    if (rc) return rc;
    // lifted-row cache (edmdc_lift_cache): when every chunk of this call fits, lift straight into the cache slots
    c->lift_cache_valid = false;
    const int64_t nchunks = total_pairs_rows > 0 ? (total_pairs_rows + chunk - 1) / chunk : 0;
    const size_t slot = lift_slot_bytes(chunk, s.width);
    const bool caching = c->lift_cache && nchunks > 0 && (size_t)nchunks * slot <= c->lift_cache_cap;
    CallTimer t(c);
    if (total_pairs_rows == 0) {
        HIPCK(c, hipMemsetAsync(c->d_partial, 0, pdoubles * 8, c->stream));
    }
    int first = 1;
    int64_t ci = 0;
    for (int64_t r0 = 0; r0 < total_pairs_rows; r0 += chunk, ++ci) {
        const int64_t npairs = (total_pairs_rows - r0 < chunk) ? (total_pairs_rows - r0) : chunk;
        const int64_t rows_lift = (npairs + 3) / 4 * 4 + 1;
        double* z = caching ? reinterpret_cast<double*>(c->lift_cache + (size_t)ci * slot) : dZ;
        double* w = caching ? z + (size_t)(chunk + 8) * s.width : dw;
        HIPCK(c, launch_lift_rows_total(c->stream, s, gamma, d_C, r0, rows_lift, total_rows, L, xs, us, d_X, d_U, z, w, pf));
        HIPCK(c, launch_gram_chunk_tasks(c->stream, s, c->ntasks[mode], c->d_tasks[mode], npairs, z, z, w, c->d_partial, first ? 0 : 1));
        first = 0;
    }
    if (caching) {
        c->lift_key = {d_X, d_U, d_C, n, r, k, gamma, bl.nbags, L, xs, us, chunk, bl.hash};
        c->lift_cache_valid = true;
    }
    HIPCK(c, launch_gram_finish_tasks(c->stream, s, c->ntasks[mode], c->d_tasks[mode], c->d_partial, accumulate, d_GtG, d_GtY));
    return BROV_OK;
}

int edmdc_gram_dev(brov_ctx* c, int n, int r, int k, double gamma, const double* d_C, int64_t nbags, int64_t L,
                   int64_t xs, int64_t us, const double* d_X, const double* d_U, int accumulate, double* d_GtG, double* d_GtY) {
    int rc = edmdc_shape_ok(c, n, r, k);
    if (rc) return rc;
    if (nbags < 0 || L < 0 || !d_C || !d_GtG || (nbags && L && (!d_X || (r && !d_U))) || (nbags > 1 && (xs < L + 1 || us < L)))
        return fail(c, BROV_ERR_ARG, "edmdc_gram_dev: bad argument");
    BagLayout bl;
    bag_layout_uniform(nbags, L, xs, us, &bl);
    return gram_core(c, n, r, k, gamma, d_C, bl, d_X, d_U, accumulate, d_GtG, d_GtY);
}

int edmdc_gram_ragged_dev(brov_ctx* c, int n, int r, int k, double gamma, const double* d_C, int64_t nbags, const int64_t* bag_offsets,
                          const double* d_X, const double* d_U, int accumulate, double* d_GtG, double* d_GtY) {
    int rc = edmdc_shape_ok(c, n, r, k);
    if (rc) return rc;
    if (!d_C || !d_GtG) return fail(c, BROV_ERR_ARG, "edmdc_gram_ragged_dev: bad argument");
    BagLayout bl;
    rc = bag_layout_ragged(c, nbags, bag_offsets, &bl, "edmdc_gram_ragged_dev");
    if (rc) return rc;
    if (bl.total_rows > 1 && (!d_X || (r && !d_U))) return fail(c, BROV_ERR_ARG, "edmdc_gram_ragged_dev: bad argument");
    return gram_core(c, n, r, k, gamma, d_C, bl, d_X, d_U, accumulate, d_GtG, d_GtY);
}

int edmdc_gram_decomposition(int n, int r, int k, int* ntasks, int* nslabs) {
    if (!edmdc_shape_supported(n, r, k)) return BROV_ERR_ARG;
    (void)gram_partial_doubles(edmdc_shape(n, r, k), 0, ntasks, nslabs);
    return BROV_OK;
}

int edmdc_gtg_decomposition(int n, int r, int k, int* ntasks, int* nslabs) {
    if (!edmdc_shape_supported(n, r, k)) return BROV_ERR_ARG;
    (void)gram_partial_doubles(edmdc_shape(n, r, k), 2, ntasks, nslabs);
    return BROV_OK;
}

int edmdc_apply_decomposition(int n, int r, int k, int* wrows_items_per_192_rows, int* wrows_tiles_wanted, int* wty_tasks, int* wty_slabs) {
    if (!edmdc_shape_supported(n, r, k)) return BROV_ERR_ARG;
    const EdmdcShape s = edmdc_shape(n, r, k);
    wrows_decomposition(s, wrows_items_per_192_rows, wrows_tiles_wanted);
    (void)gram_partial_doubles(s, 1, wty_tasks, wty_slabs);
    return BROV_OK;
}

int edmdc_lift_cache(brov_ctx* c, void* d_buffer, size_t bytes) {
    if (!c || (d_buffer && (reinterpret_cast<uintptr_t>(d_buffer) & 15))) return fail(c, BROV_ERR_ARG, "edmdc_lift_cache: buffer must be 16-byte aligned");
    c->lift_cache = d_buffer ? static_cast<char*>(d_buffer) : nullptr;
    c->lift_cache_cap = d_buffer ? bytes : 0;
    c->lift_cache_valid = false;
    return BROV_OK;
}

int edmdc_set_kmeans_variant(brov_ctx* c, int variant) {
    if (!c || variant < 0 || (variant & ~KMV_ACCEPTED) != 0 || (variant & 3) == 3)
        return fail(c, BROV_ERR_ARG, "edmdc_set_kmeans_variant: variant must be 0, 1 (full scan) or 2 (no sorted order), optionally + 4 (distance bounds off)");
    c->kmeans_variant = variant;
    return BROV_OK;
}
int edmdc_set_kmeans_bounds_rate(brov_ctx* c, double rate) {
    if (!c || !(rate >= 0.0) || rate > 1.0) return fail(c, BROV_ERR_ARG, "edmdc_set_kmeans_bounds_rate: rate must lie in [0, 1]");
    c->km_bounds_rate = rate;
    return BROV_OK;
}
int brov_set_rollout_variant(brov_ctx* c, int variant) {
    if (!c || variant < 0 || variant > 1) return fail(c, BROV_ERR_ARG, "brov_set_rollout_variant: variant must be 0 or 1");
    c->single_lane = variant;
    return BROV_OK;
}
int brov_experiments_build(void) {
#ifdef BROV2_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

int edmdc_set_kmeans_far_select(brov_ctx* c, brov_far_select_fn fn, void* user) {
    if (!c) return BROV_ERR_ARG;
    c->far_select = fn;
    c->far_select_user = user;
    return BROV_OK;
}
int edmdc_kmeans_relocations(brov_ctx* c) { return c ? c->kmeans_relocations : 0; }
int edmdc_kmeans_loop_info(brov_ctx* c, int info[4]) {
    if (!c || !info) return BROV_ERR_ARG;
    info[0] = c->kmeans_relocations; info[1] = c->km_info[1]; info[2] = c->km_info[2]; info[3] = c->km_info[3];
    return BROV_OK;
}
static void far_select_default(const double* dist, int64_t N, int n_empty, int64_t* out);
int edmdc_far_select_numpy(const double* distances, int64_t N, int n_empty, int64_t* far_rows_out) {
    if (!distances || !far_rows_out || N < 1 || n_empty < 1 || n_empty > N) return BROV_ERR_ARG;
    far_select_default(distances, N, n_empty, far_rows_out);
    return BROV_OK;
}
int edmdc_set_kmeans_allreduce(brov_ctx* c, brov_allreduce_fn fn, void* user) {
    if (!c) return BROV_ERR_ARG;
    c->km_allreduce = fn;
    c->km_allreduce_user = user;
    return BROV_OK;
}

// the torch-free transport of the sharded Lloyd loop: the exchanges go straight to RCCL on the ctx stream (comm.hip)
static int km_allreduce_through_comm(void* user, void* d_buf, int64_t count, int op) {
    brov_ctx* c = static_cast<brov_ctx*>(user);
    return brov_comm_allreduce_words(c->km_comm, d_buf, count, op, c->stream);
}
int edmdc_set_kmeans_shard(brov_ctx* c, int rank, int world, int64_t row_offset, int64_t n_global) {
    if (!c || world < 1 || rank < 0 || rank >= world || row_offset < 0 || row_offset >= (1ll << 61) || n_global < 0)
        return fail(c, BROV_ERR_ARG, "edmdc_set_kmeans_shard: bad argument");
    c->km_rank = rank;
    c->km_world = world;
    c->km_row_offset = row_offset;
    c->km_n_global = n_global;
    return BROV_OK;
}
int edmdc_kmeans_use_comm(brov_ctx* c, brov_comm* comm) {
    if (!c) return BROV_ERR_ARG;
    c->km_comm = comm;
    c->km_allreduce = comm ? km_allreduce_through_comm : nullptr;
    c->km_allreduce_user = comm ? c : nullptr;
    return BROV_OK;
}

int edmdc_set_apply_variant(brov_ctx* c, int variant) {
    if (!c || variant < 0 || variant > 1) return fail(c, BROV_ERR_ARG, "edmdc_set_apply_variant: variant must be 0 or 1");
    c->apply_variant = variant;
    return BROV_OK;
}

// ---- fit()'s own association: M = (P G^T) Y  (Koopman/koopmanEDMDc.py:97) ------------------------------------------
static int apply_core(brov_ctx* c, int n, int r, int k, double gamma, const double* d_C, const BagLayout& bl,
                      const double* d_X, const double* d_U, const double* P_host, double* d_M) {
    int rc = BROV_OK;
    DeviceGuard g(c);
    const EdmdcShape s = edmdc_shape(n, r, k);
    rc = ensure_tasks(c, s, 1);
    if (rc) return rc;
    const int64_t nbags = bl.nbags, L = bl.L, xs = bl.xs, us = bl.us;
    const int64_t total_rows = bl.total_rows;
    const int64_t total_pairs_rows = total_rows > 0 ? total_rows - 1 : 0;
    int nslab = 0, ntasks = 0;
    const size_t pdoubles = gram_partial_doubles(s, 1, &ntasks, &nslab);
    rc = ensure_partial(c, pdoubles);
    if (rc) return rc;
    int64_t chunk = c->chunk_rows;
    if (chunk > (total_pairs_rows + 3) / 4 * 4) chunk = (total_pairs_rows + 3) / 4 * 4;
    if (chunk < 4) chunk = 4;
    const int W = s.width;
    Arena a(c);
    rc = a.reserve(2 * Arena::al((size_t)(chunk + 8) * W * 8) + Arena::al((chunk + 8) * 8) + Arena::al((size_t)(W + 8) * W * 8) + bl.scratch_bytes());
    if (rc) return rc;
    double* dZ = a.take<double>((size_t)(chunk + 8) * W);
    double* dWr = a.take<double>((size_t)(chunk + 8) * W);
    double* dw = a.take<double>(chunk + 8);
    double* dPt = a.take<double>((size_t)(W + 8) * W);      // 8 rows of padding: wrows_kernel prefetches two K-steps past the last feature
    const unsigned char* pf = nullptr;
    rc = bag_pairflags(c, a, bl, &pf);
    if (rc) return rc;
    {   // PdT[f][j] = P[ref(j)][ref(f)]: P^T permuted to the device feature order, zero for padding features
        std::vector<double> h((size_t)(W + 8) * W, 0.0);
        const int p = s.p;
        for (int f = 0; f < W; ++f) {
            const int rf = edmdc_dev_to_ref_feature(s, f);
            if (rf < 0) continue;
            for (int j = 0; j < W; ++j) {
                const int rj = edmdc_dev_to_ref_feature(s, j);
                if (rj >= 0) h[(size_t)f * W + j] = P_host[(size_t)rj * p + rf];
            }
        }
        HIPCK(c, h2d_copy(c, dPt, h.data(), h.size() * 8));
        HIPCK(c, hipStreamSynchronize(c->stream));
    }
    HIPCK(c, hipMemsetAsync(dWr, 0, (size_t)(chunk + 8) * W * 8, c->stream));   // the Gram kernel prefetches up to 8 rows past the chunk
    // the lifted rows edmdc_gram_dev left in the cache for exactly these arguments (edmdc_lift_cache): no second lift
    const brov_ctx::LiftKey& lk = c->lift_key;
    const bool cached = c->lift_cache_valid && lk.X == d_X && lk.U == d_U && lk.C == d_C && lk.n == n && lk.r == r && lk.k == k &&
                        lk.gamma == gamma && lk.nbags == nbags && lk.L == L && lk.xs == xs && lk.us == us && lk.chunk == chunk && lk.bag_hash == bl.hash;
    const size_t slot = lift_slot_bytes(chunk, W);
    CallTimer t(c);
    if (total_pairs_rows == 0) HIPCK(c, hipMemsetAsync(c->d_partial, 0, pdoubles * 8, c->stream));
    int first = 1;
    int64_t ci = 0;
    for (int64_t r0 = 0; r0 < total_pairs_rows; r0 += chunk, ++ci) {
        const int64_t npairs = (total_pairs_rows - r0 < chunk) ? (total_pairs_rows - r0) : chunk;
        const int64_t rows_lift = (npairs + 3) / 4 * 4 + 1;
        double* z = cached ? reinterpret_cast<double*>(c->lift_cache + (size_t)ci * slot) : dZ;
        double* w = cached ? z + (size_t)(chunk + 8) * W : dw;
        if (!cached) HIPCK(c, launch_lift_rows_total(c->stream, s, gamma, d_C, r0, rows_lift, total_rows, L, xs, us, d_X, d_U, z, w, pf));
        HIPCK(c, launch_rows_times_pt(c->stream, s, rows_lift, z, dPt, dWr, c->apply_variant));
        HIPCK(c, launch_gram_chunk_tasks(c->stream, s, c->ntasks[1], c->d_tasks[1], npairs, dWr, z, w, c->d_partial, first ? 0 : 1));
        first = 0;
    }
    HIPCK(c, launch_gram_finish_tasks(c->stream, s, c->ntasks[1], c->d_tasks[1], c->d_partial, 0, nullptr, d_M));
    return BROV_OK;
}

int edmdc_pinv_apply_dev(brov_ctx* c, int n, int r, int k, double gamma, const double* d_C, int64_t nbags, int64_t L,
                         int64_t xs, int64_t us, const double* d_X, const double* d_U, const double* P_host, double* d_M) {
    int rc = edmdc_shape_ok(c, n, r, k);
    if (rc) return rc;
    if (nbags < 0 || L < 0 || !d_C || !P_host || !d_M || (nbags && L && (!d_X || (r && !d_U))) || (nbags > 1 && (xs < L + 1 || us < L)))
        return fail(c, BROV_ERR_ARG, "edmdc_pinv_apply_dev: bad argument");
    BagLayout bl;
    bag_layout_uniform(nbags, L, xs, us, &bl);
    return apply_core(c, n, r, k, gamma, d_C, bl, d_X, d_U, P_host, d_M);
}

int edmdc_pinv_apply_ragged_dev(brov_ctx* c, int n, int r, int k, double gamma, const double* d_C, int64_t nbags, const int64_t* bag_offsets,
                                const double* d_X, const double* d_U, const double* P_host, double* d_M) {
    int rc = edmdc_shape_ok(c, n, r, k);
    if (rc) return rc;
    if (!d_C || !P_host || !d_M) return fail(c, BROV_ERR_ARG, "edmdc_pinv_apply_ragged_dev: bad argument");
    BagLayout bl;
    rc = bag_layout_ragged(c, nbags, bag_offsets, &bl, "edmdc_pinv_apply_ragged_dev");
    if (rc) return rc;
    if (bl.total_rows > 1 && (!d_X || (r && !d_U))) return fail(c, BROV_ERR_ARG, "edmdc_pinv_apply_ragged_dev: bad argument");
    return apply_core(c, n, r, k, gamma, d_C, bl, d_X, d_U, P_host, d_M);
}

// host forms: the arrays are staged in device allocations of their own for the duration of the call
static int apply_host(brov_ctx* c, int n, int r, int k, double gamma, const double* C, const BagLayout& bl, int64_t urows,
                      const double* X, const double* U, const double* P, double* M) {
    DeviceGuard g(c);
    LiftCacheDisarm disarm(c);
    const int64_t xrows = bl.total_rows;
    const int d = n + k, p = d + r;
    double *dX = nullptr, *dU = nullptr, *dC = nullptr, *dM = nullptr;
    auto cleanup = [&]() { (void)hipFree(dX); (void)hipFree(dU); (void)hipFree(dC); (void)hipFree(dM); };
#define HIPCK_CLEAN(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { cleanup(); return hip_fail(c, e__, #call); } } while (0)
    HIPCK_CLEAN(hipMalloc((void**)&dX, (size_t)(xrows > 0 ? xrows : 1) * n * 8));
    HIPCK_CLEAN(hipMalloc((void**)&dU, (size_t)(urows > 0 ? urows : 1) * (r > 0 ? r : 1) * 8));
    HIPCK_CLEAN(hipMalloc((void**)&dC, (size_t)k * n * 8));
    HIPCK_CLEAN(hipMalloc((void**)&dM, (size_t)p * d * 8));
    if (xrows) HIPCK_CLEAN(h2d_copy(c, dX, X, (size_t)xrows * n * 8));
    if (urows && r) HIPCK_CLEAN(h2d_copy(c, dU, U, (size_t)urows * r * 8));
    HIPCK_CLEAN(h2d_copy(c, dC, C, (size_t)k * n * 8));
    HIPCK_CLEAN(hipMemsetAsync(dM, 0, (size_t)p * d * 8, c->stream));
    int rc = apply_core(c, n, r, k, gamma, dC, bl, dX, dU, P, dM);
    if (rc) { cleanup(); return rc; }
    HIPCK_CLEAN(d2h_copy(c, M, dM, (size_t)p * d * 8));
    HIPCK_CLEAN(hipStreamSynchronize(c->stream));
    cleanup();
    return BROV_OK;
}

int edmdc_pinv_apply(brov_ctx* c, int n, int r, int k, double gamma, const double* C, int64_t nbags, int64_t L, int64_t xs, int64_t us,
                     const double* X, const double* U, const double* P, double* M) {
    int rc = edmdc_shape_ok(c, n, r, k);
    if (rc) return rc;
    if (nbags < 0 || L < 0 || !C || !P || !M || (nbags && L && (!X || (r && !U))) || (nbags > 1 && (xs < L + 1 || us < L)))
        return fail(c, BROV_ERR_ARG, "edmdc_pinv_apply: bad argument");
    BagLayout bl;
    bag_layout_uniform(nbags, L, xs, us, &bl);
    return apply_host(c, n, r, k, gamma, C, bl, nbags > 0 ? (nbags - 1) * bl.us + L : 0, X, U, P, M);
}

int edmdc_pinv_apply_ragged(brov_ctx* c, int n, int r, int k, double gamma, const double* C, int64_t nbags, const int64_t* bag_offsets,
                            const double* X, const double* U, const double* P, double* M) {
    int rc = edmdc_shape_ok(c, n, r, k);
    if (rc) return rc;
    if (!C || !P || !M) return fail(c, BROV_ERR_ARG, "edmdc_pinv_apply_ragged: bad argument");
    BagLayout bl;
    rc = bag_layout_ragged(c, nbags, bag_offsets, &bl, "edmdc_pinv_apply_ragged");
    if (rc) return rc;
    if (bl.total_rows > 1 && (!X || (r && !U))) return fail(c, BROV_ERR_ARG, "edmdc_pinv_apply_ragged: bad argument");
    return apply_host(c, n, r, k, gamma, C, bl, bl.total_rows, X, U, P, M);
}

static int gram_host(brov_ctx* c, int n, int r, int k, double gamma, const double* C, const BagLayout& bl, int64_t urows,
                     const double* X, const double* U, int accumulate, double* GtG, double* GtY) {
    DeviceGuard g(c);
    LiftCacheDisarm disarm(c);
    const int64_t xrows = bl.total_rows;
    const int d = n + k, p = d + r;
    // inputs live in plain device allocations (the arena is used by gram_core itself)
    double *dX = nullptr, *dU = nullptr, *dC = nullptr, *dG = nullptr, *dY = nullptr;
    auto cleanup = [&]() { (void)hipFree(dX); (void)hipFree(dU); (void)hipFree(dC); (void)hipFree(dG); (void)hipFree(dY); };
    HIPCK_CLEAN(hipMalloc((void**)&dX, (size_t)(xrows > 0 ? xrows : 1) * n * 8));
    HIPCK_CLEAN(hipMalloc((void**)&dU, (size_t)(urows > 0 ? urows : 1) * (r > 0 ? r : 1) * 8));
    HIPCK_CLEAN(hipMalloc((void**)&dC, (size_t)k * n * 8));
    HIPCK_CLEAN(hipMalloc((void**)&dG, (size_t)p * p * 8));
    HIPCK_CLEAN(hipMalloc((void**)&dY, (size_t)p * d * 8));
    if (xrows) HIPCK_CLEAN(h2d_copy(c, dX, X, (size_t)xrows * n * 8));
    if (urows && r) HIPCK_CLEAN(h2d_copy(c, dU, U, (size_t)urows * r * 8));
    HIPCK_CLEAN(h2d_copy(c, dC, C, (size_t)k * n * 8));
    if (accumulate) {
        HIPCK_CLEAN(h2d_copy(c, dG, GtG, (size_t)p * p * 8));
        HIPCK_CLEAN(h2d_copy(c, dY, GtY, (size_t)p * d * 8));
    } else {
        HIPCK_CLEAN(hipMemsetAsync(dG, 0, (size_t)p * p * 8, c->stream));
        HIPCK_CLEAN(hipMemsetAsync(dY, 0, (size_t)p * d * 8, c->stream));
    }
    int rc = gram_core(c, n, r, k, gamma, dC, bl, dX, dU, accumulate, dG, dY);
    if (rc) { cleanup(); return rc; }
    HIPCK_CLEAN(d2h_copy(c, GtG, dG, (size_t)p * p * 8));
    HIPCK_CLEAN(d2h_copy(c, GtY, dY, (size_t)p * d * 8));
    HIPCK_CLEAN(hipStreamSynchronize(c->stream));
    cleanup();
    return BROV_OK;
}

int edmdc_gram(brov_ctx* c, int n, int r, int k, double gamma, const double* C, int64_t nbags, int64_t L, int64_t xs, int64_t us,
               const double* X, const double* U, int accumulate, double* GtG, double* GtY) {
    int rc = edmdc_shape_ok(c, n, r, k);
    if (rc) return rc;
    if (nbags < 0 || L < 0 || !C || !GtG || !GtY || (nbags && L && (!X || (r && !U))) || (nbags > 1 && (xs < L + 1 || us < L)))
        return fail(c, BROV_ERR_ARG, "edmdc_gram: bad argument");
    BagLayout bl;
    bag_layout_uniform(nbags, L, xs, us, &bl);
    return gram_host(c, n, r, k, gamma, C, bl, nbags > 0 ? (nbags - 1) * bl.us + L : 0, X, U, accumulate, GtG, GtY);
}

int edmdc_gram_ragged(brov_ctx* c, int n, int r, int k, double gamma, const double* C, int64_t nbags, const int64_t* bag_offsets,
                      const double* X, const double* U, int accumulate, double* GtG, double* GtY) {
    int rc = edmdc_shape_ok(c, n, r, k);
    if (rc) return rc;
    if (!C || !GtG || !GtY) return fail(c, BROV_ERR_ARG, "edmdc_gram_ragged: bad argument");
    BagLayout bl;
    rc = bag_layout_ragged(c, nbags, bag_offsets, &bl, "edmdc_gram_ragged");
    if (rc) return rc;
    if (bl.total_rows > 1 && (!X || (r && !U))) return fail(c, BROV_ERR_ARG, "edmdc_gram_ragged: bad argument");
    return gram_host(c, n, r, k, gamma, C, bl, bl.total_rows, X, U, accumulate, GtG, GtY);
}

// ---- lifted propagation: evaluate / multistep_rmse / simulate ---------------------------------------
// builds ABt [ppad][dpad] = [A | B]^T (zero padded) from host A [d][d], B [d][r]
static int upload_ABt(brov_ctx* c, const PropShape& s, const double* A, const double* B, double* dABt) {
    std::vector<double> h((size_t)s.ppad * s.dpad, 0.0);
    for (int i = 0; i < s.d; ++i) {
        for (int j = 0; j < s.d; ++j) h[(size_t)j * s.dpad + i] = A[(size_t)i * s.d + j];
        for (int j = 0; j < s.r; ++j) h[(size_t)(s.d + j) * s.dpad + i] = B[(size_t)i * s.r + j];
    }
    HIPCK(c, h2d_copy(c, dABt, h.data(), h.size() * 8));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return BROV_OK;
}

int edmdc_multistep_se(brov_ctx* c, int n, int r, int k, double gamma, const double* C, const double* A, const double* B,
                       int64_t N, int64_t H, const double* X, const double* U, double* se_total, double* xhat_end) {
    int rc = edmdc_shape_ok(c, n, r, k);
    if (rc) return rc;
    if (N < 0 || H < 0 || !C || !A || !B || !se_total || (N && (!X || (r && !U)))) return fail(c, BROV_ERR_ARG, "edmdc_multistep_se: bad argument");
    const int64_t nw = N - H;
    if (nw <= 0) { *se_total = 0.0; return BROV_OK; }
    DeviceGuard g(c);
    const PropShape s = prop_shape(n, r, k, nw);
    const int rpad = s.ppad - s.d;
    const int64_t NUt = s.nwp + H + 16;
    Arena a(c);
    rc = a.reserve(Arena::al(N * n * 8) + Arena::al(N * (r ? r : 1) * 8) + Arena::al((size_t)k * n * 8) + Arena::al((size_t)s.ppad * s.dpad * 8) +
                   2 * Arena::al((size_t)s.zrows * s.nwp * 8) + Arena::al((size_t)rpad * NUt * 8) + Arena::al(nw * 8) + Arena::al(nw * n * 8) + 4096);
    if (rc) return rc;
    double* dX = a.take<double>(N * n);
    double* dU = a.take<double>(N * (r ? r : 1));
    double* dC = a.take<double>((size_t)k * n);
    double* dABt = a.take<double>((size_t)s.ppad * s.dpad);
    double* dZ0 = a.take<double>((size_t)s.zrows * s.nwp);
    double* dZ1 = a.take<double>((size_t)s.zrows * s.nwp);
    double* dUt = a.take<double>((size_t)rpad * NUt);
    double* dse = a.take<double>(nw);
    double* dxh = a.take<double>(nw * n);
    double* dtot = a.take<double>(8);
    HIPCK(c, h2d_copy(c, dX, X, N * n * 8));
    // the windows read input rows 0 .. N-2 only (window k, step t uses U[k+t], k+t <= N-2): like the reference's
    // multistep_rmse / evaluate, accept a U with N-1 rows and never touch row N-1 of the caller's buffer
    const int64_t urows = N - 1;
    if (r) {
        HIPCK(c, hipMemsetAsync(dU + urows * r, 0, (size_t)r * 8, c->stream));
        if (urows > 0) HIPCK(c, h2d_copy(c, dU, U, urows * r * 8));
    }
    HIPCK(c, h2d_copy(c, dC, C, (size_t)k * n * 8));
    rc = upload_ABt(c, s, A, B, dABt);
    if (rc) return rc;
    HIPCK(c, hipMemsetAsync(dUt, 0, (size_t)rpad * NUt * 8, c->stream));
    HIPCK(c, hipMemsetAsync(dZ0, 0, (size_t)s.zrows * s.nwp * 8, c->stream));   // padding rows are K rows of the step: must be finite
    HIPCK(c, hipMemsetAsync(dZ1, 0, (size_t)s.zrows * s.nwp * 8, c->stream));
    // window groups and their streams (created once per ctx, outside the timed region)
    const int64_t nwb = prop_window_blocks(s);
    const int G = (c->prop_groups > 1 && nwb >= 8 * c->prop_groups && H > 1) ? c->prop_groups : 1;
    if (G > 1 && !c->ev_fork) HIPCK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    for (int g = 1; g < G; ++g)
        if (!c->side[g - 1]) {
            HIPCK(c, hipStreamCreateWithFlags(&c->side[g - 1], hipStreamNonBlocking));
            HIPCK(c, hipEventCreateWithFlags(&c->ev_join[g - 1], hipEventDisableTiming));
        }
    // Any early return between the fork and the join below would leave side-stream kernels running on arena scratch that the
    // Arena destructor hands to the next call: this guard (declared after the Arena, so destroyed before it) waits for the
    // side streams on every path that did not reach the join.
    struct SideJoin {
        brov_ctx* c; int G; bool armed = false;
        ~SideJoin() { if (armed) for (int g = 1; g < G; ++g) if (c->side[g - 1]) (void)hipStreamSynchronize(c->side[g - 1]); }
    } side_join{c, G};
    {
        CallTimer t(c);
        HIPCK(c, launch_transpose(c->stream, N, r, dU, r, dUt, NUt));
        HIPCK(c, launch_lift_t(c->stream, s, gamma, n, dX, dC, dZ0));
        HIPCK(c, launch_set_input_rows(c->stream, s, dUt, NUt, dZ0));                 // u_0 of every window: U[w]
        double *zin = dZ0, *zout = dZ1;
        // A window block's step t+1 needs that block's step t only, so the recurrence runs in G groups of window blocks on G
        // streams: at G = 2 each launch is one round of waves (3 938 -> 2 x 1 969 for 2 048 slots at the recorded size) and
        // the ramp and tail of one group's launch are covered by the other's.
        if (G > 1) {
            HIPCK(c, hipEventRecord(c->ev_fork, c->stream));
            side_join.armed = true;
            for (int g = 1; g < G; ++g) HIPCK(c, hipStreamWaitEvent(c->side[g - 1], c->ev_fork, 0));
        }
        for (int64_t t = 0; t < H; ++t) {
            // window w at step t+1 reads U[w + t + 1]: the transposed input array shifted by t + 1
            const double* un = t + 1 < H ? dUt + t + 1 : nullptr;
            for (int g = 0; g < G; ++g) {
                const int64_t w0 = nwb * g / G, w1 = nwb * (g + 1) / G;
                HIPCK(c, launch_propagate(g == 0 ? c->stream : c->side[g - 1], s, dABt, zin, un, NUt, zout, w0, w1 - w0));
            }
            std::swap(zin, zout);
        }
        for (int g = 1; g < G; ++g) { HIPCK(c, hipEventRecord(c->ev_join[g - 1], c->side[g - 1])); HIPCK(c, hipStreamWaitEvent(c->stream, c->ev_join[g - 1], 0)); }
        side_join.armed = false;                     // joined: the ctx stream now orders everything behind the side streams
        HIPCK(c, launch_endpoint_se(c->stream, s, n, dX + H * n, zin, dse, xhat_end ? dxh : nullptr));
        HIPCK(c, launch_sum(c->stream, nw, dse, dtot));
    }
    HIPCK(c, d2h_copy(c, se_total, dtot, 8));
    if (xhat_end) HIPCK(c, d2h_copy(c, xhat_end, dxh, nw * n * 8));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return BROV_OK;
}

int edmdc_multistep_se_linear(brov_ctx* c, int n, int r, int k, double gamma, const double* C, const double* RHt, const double* Gt,
                              int64_t N, int64_t H, const double* X, const double* U, double* se_total, double* xhat_end) {
    int rc = edmdc_shape_ok(c, n, r, k);
    if (rc) return rc;
    if (N < 0 || H < 0 || !C || !RHt || (H && r && !Gt) || !se_total || (N && (!X || (r && !U)))) return fail(c, BROV_ERR_ARG, "edmdc_multistep_se_linear: bad argument");
    const int64_t nw = N - H;
    if (nw <= 0) { *se_total = 0.0; return BROV_OK; }
    DeviceGuard g(c);
    const int d = n + k;
    const int64_t urows = N - 1;                       // like edmdc_multistep_se: rows 0 .. N-2 of U are read, a shorter-by-one U is accepted
    Arena a(c);
    rc = a.reserve(Arena::al(N * n * 8) + Arena::al(N * (r ? r : 1) * 8) + Arena::al((size_t)k * n * 8) + Arena::al((size_t)d * n * 8) +
                   Arena::al((size_t)(H ? H : 1) * (r ? r : 1) * n * 8) + Arena::al(nw * 8) + Arena::al(nw * n * 8) + 4096);
    if (rc) return rc;
    double* dX = a.take<double>(N * n);
    double* dU = a.take<double>(N * (r ? r : 1));
    double* dC = a.take<double>((size_t)k * n);
    double* dR = a.take<double>((size_t)d * n);
    double* dG = a.take<double>((size_t)(H ? H : 1) * (r ? r : 1) * n);
    double* dse = a.take<double>(nw);
    double* dxh = a.take<double>(nw * n);
    double* dtot = a.take<double>(8);
    HIPCK(c, h2d_copy(c, dX, X, N * n * 8));
    if (r) {
        HIPCK(c, hipMemsetAsync(dU + urows * r, 0, (size_t)r * 8, c->stream));
        if (urows > 0) HIPCK(c, h2d_copy(c, dU, U, urows * r * 8));
    }
    HIPCK(c, h2d_copy(c, dC, C, (size_t)k * n * 8));
    HIPCK(c, h2d_copy(c, dR, RHt, (size_t)d * n * 8));
    if (H && r) HIPCK(c, h2d_copy(c, dG, Gt, (size_t)H * r * n * 8));
    {
        CallTimer t(c);
        HIPCK(c, launch_linear_windows(c->stream, nw, n, r, k, H, gamma, dX, dU, dC, dR, dG, dse, xhat_end ? dxh : nullptr));
        HIPCK(c, launch_sum(c->stream, nw, dse, dtot));
    }
    HIPCK(c, d2h_copy(c, se_total, dtot, 8));
    if (xhat_end) HIPCK(c, d2h_copy(c, xhat_end, dxh, nw * n * 8));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return BROV_OK;
}

int edmdc_simulate(brov_ctx* c, int n, int r, int k, double gamma, const double* C, const double* A, const double* B,
                   int64_t nb, int64_t T, const double* x0, const double* U_seq, double* X_pred) {
    int rc = edmdc_shape_ok(c, n, r, k);
    if (rc) return rc;
    if (nb < 0 || T < 0 || !C || !A || !B || (nb && (!x0 || !X_pred || (T && r && !U_seq)))) return fail(c, BROV_ERR_ARG, "edmdc_simulate: bad argument");
    if (nb == 0) return BROV_OK;
    DeviceGuard g(c);
    const PropShape s = prop_shape(n, r, k, nb);
    const int rpad = s.ppad - s.d;
    const int64_t ust_rows = (T > 0 ? T : 1) * (int64_t)rpad;   // [t][rpad][nbp]: step t uses rows t*r .. (reads up to rpad rows)
    Arena a(c);
    rc = a.reserve(Arena::al(nb * n * 8) + Arena::al((size_t)nb * T * (r ? r : 1) * 8 + 8) + Arena::al((size_t)k * n * 8) +
                   Arena::al((size_t)s.ppad * s.dpad * 8) + 2 * Arena::al((size_t)s.zrows * s.nwp * 8) +
                   Arena::al((size_t)(ust_rows + rpad) * s.nwp * 8) + Arena::al((size_t)nb * (T + 1) * n * 8) + 4096);
    if (rc) return rc;
    double* dx0 = a.take<double>(nb * n);
    double* dUs = a.take<double>((size_t)nb * T * (r ? r : 1) + 1);
    double* dC = a.take<double>((size_t)k * n);
    double* dABt = a.take<double>((size_t)s.ppad * s.dpad);
    double* dZ0 = a.take<double>((size_t)s.zrows * s.nwp);
    double* dZ1 = a.take<double>((size_t)s.zrows * s.nwp);
    double* dUst = a.take<double>((size_t)(ust_rows + rpad) * s.nwp);
    double* dXp = a.take<double>((size_t)nb * (T + 1) * n);
    HIPCK(c, h2d_copy(c, dx0, x0, nb * n * 8));
    if (T && r) HIPCK(c, h2d_copy(c, dUs, U_seq, (size_t)nb * T * r * 8));
    HIPCK(c, h2d_copy(c, dC, C, (size_t)k * n * 8));
    rc = upload_ABt(c, s, A, B, dABt);
    if (rc) return rc;
    HIPCK(c, hipMemsetAsync(dUst, 0, (size_t)(ust_rows + rpad) * s.nwp * 8, c->stream));
    HIPCK(c, hipMemsetAsync(dZ0, 0, (size_t)s.zrows * s.nwp * 8, c->stream));
    HIPCK(c, hipMemsetAsync(dZ1, 0, (size_t)s.zrows * s.nwp * 8, c->stream));
    {
        CallTimer t(c);
        HIPCK(c, launch_useq_t(c->stream, s, T, dUs, dUst));
        HIPCK(c, launch_lift_t(c->stream, s, gamma, n, dx0, dC, dZ0));
        HIPCK(c, launch_set_input_rows(c->stream, s, dUst, s.nwp, dZ0));                 // u_0
        double *zin = dZ0, *zout = dZ1;
        HIPCK(c, launch_extract_state(c->stream, s, T + 1, 0, zin, dXp));
        for (int64_t t = 0; t < T; ++t) {
            HIPCK(c, launch_propagate(c->stream, s, dABt, zin, t + 1 < T ? dUst + (size_t)(t + 1) * r * s.nwp : nullptr, s.nwp, zout));
            std::swap(zin, zout);
            HIPCK(c, launch_extract_state(c->stream, s, T + 1, t + 1, zin, dXp));
        }
    }
    HIPCK(c, d2h_copy(c, X_pred, dXp, (size_t)nb * (T + 1) * n * 8));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return BROV_OK;
}

// ---- k-means (Lloyd) --------------------------------------------------------------------------------------
#ifndef KM_SORT_MOVED
#define KM_SORT_MOVED 0.03       // re-sort thresholds of the loop's private sample order, see below
#endif
#ifndef KM_SORT_RATE
#define KM_SORT_RATE 0.2
#endif
// `_relocate_empty_clusters_dense`'s choice of rows is `np.argpartition(distances, -n_empty)[:-n_empty-1:-1]`
// (sklearn/cluster/_k_means_common.pyx): WHICH of several equal distances it returns, and in which order the n_empty largest come
// out (= which empty cluster gets which row), are properties of NumPy's selection algorithm.  NumPy has two: its own introselect
// (numpy/_core/src/npysort/selection.cpp: median-of-3 quickselect on the index array, median-of-medians-of-5 once the depth limit
// 2 floor(log2 N) is spent, an O(n kth) selection when kth is within 3 of the range's start, a plain maximum scan for kth = N - 1) --
// what runs wherever no SIMD kernel is dispatched -- and, on x86 hosts with AVX-512 / AVX2, x86-simd-sort's vectorised argselect,
// whose order differs (seen in this container: 7 largest of 1 000 values come out as rows 51, 423, 758 ... against 689, 243, 51 ...).
// The reference's result is therefore host-dependent.  This is the former, restated: the library's rule for plain-C callers and for
// sharded runs, pinned by tests/golden/farselect.npz (np.argpartition itself with its SIMD dispatch disabled,
// tools/gen_farselect_golden.py).  The Python layer may install a callback that calls np.argpartition on the host at hand instead
// (edmdc_set_kmeans_far_select): then the rows are scikit-learn's on THAT host, whatever it dispatches.
namespace npysel {
struct Sel {
    const double* v;
    int64_t* t;                                            // the index array being partitioned ("tosort")
    // NumPy's order for floating point: NaNs are the largest (npy::double_tag::less)
    static bool less(double a, double b) { return a < b || (b != b && a == a); }
    double at(int64_t i) const { return v[t[i]]; }
    void swap(int64_t i, int64_t j) { const int64_t x = t[i]; t[i] = t[j]; t[j] = x; }

    void dumb_select(int64_t base, int64_t num, int64_t kth) {
        for (int64_t i = 0; i <= kth; ++i) {
            int64_t minidx = i;
            double minval = at(base + i);
            for (int64_t k = i + 1; k < num; ++k)
                if (less(at(base + k), minval)) { minidx = k; minval = at(base + k); }
            swap(base + i, base + minidx);
        }
    }
    void median3_swap(int64_t low, int64_t mid, int64_t high) {
        if (less(at(high), at(mid))) swap(high, mid);
        if (less(at(high), at(low))) swap(high, low);
        if (less(at(low), at(mid))) swap(low, mid);        // the median goes to low ...
        swap(mid, low + 1);                                // ... the smallest of the three to low + 1
    }
    int64_t median5(int64_t b) {
        if (less(at(b + 1), at(b + 0))) swap(b + 1, b + 0);
        if (less(at(b + 4), at(b + 3))) swap(b + 4, b + 3);
        if (less(at(b + 3), at(b + 0))) swap(b + 3, b + 0);
        if (less(at(b + 4), at(b + 1))) swap(b + 4, b + 1);
        if (less(at(b + 2), at(b + 1))) swap(b + 2, b + 1);
        if (less(at(b + 3), at(b + 2))) return less(at(b + 3), at(b + 1)) ? 1 : 3;
        return 2;
    }
    void unguarded_partition(double pivot, int64_t* ll, int64_t* hh) {
        for (;;) {
            do { ++*ll; } while (less(at(*ll), pivot));
            do { --*hh; } while (less(pivot, at(*hh)));
            if (*hh < *ll) break;
            swap(*ll, *hh);
        }
    }
    int64_t median_of_median5(int64_t base, int64_t num) {
        const int64_t nmed = num / 5;
        for (int64_t i = 0, subleft = 0; i < nmed; ++i, subleft += 5) {
            const int64_t m = median5(base + subleft);
            swap(base + subleft + m, base + i);
        }
        if (nmed > 2) introselect(base, nmed, nmed / 2);
        return nmed / 2;
    }
    // partitions t[base .. base + num) so that position base + kth holds what a sort would put there
    void introselect(int64_t base, int64_t num, int64_t kth) {
        int64_t low = 0, high = num - 1;
        if (kth - low < 3) { dumb_select(base + low, high - low + 1, kth - low); return; }
        if (kth == num - 1) {                              // (NumPy: "useful to check if NaN present via partition(d, (x, -1))")
            int64_t maxidx = low;
            double maxval = at(base + low);
            for (int64_t k = low + 1; k < num; ++k)
                if (!less(at(base + k), maxval)) { maxidx = k; maxval = at(base + k); }
            swap(base + kth, base + maxidx);
            return;
        }
        int depth_limit = 0;
        for (uint64_t u = (uint64_t)num; u >>= 1;) ++depth_limit;
        depth_limit *= 2;
        for (; low + 1 < high;) {
            int64_t ll = low + 1, hh = high;
            if (depth_limit > 0 || hh - ll < 5) {
                median3_swap(base + low, base + low + (high - low) / 2, base + high);
            } else {
                const int64_t mid = ll + median_of_median5(base + ll, hh - ll);
                swap(base + mid, base + low);
                --ll;
                ++hh;
            }
            --depth_limit;
            int64_t al = base + ll, ah = base + hh;
            unguarded_partition(at(base + low), &al, &ah);
            ll = al - base; hh = ah - base;
            swap(base + low, base + hh);
            if (hh >= kth) high = hh - 1;
            if (hh <= kth) low = ll;
        }
        if (high == low + 1 && less(at(base + high), at(base + low))) swap(base + high, base + low);
    }
};
}  // namespace npysel

static void far_select_default(const double* dist, int64_t N, int n_empty, int64_t* out) {
    if (n_empty <= 0 || N <= 0) return;
    if (n_empty > N) n_empty = (int)N;
    std::vector<int64_t> idx((size_t)N);
    for (int64_t i = 0; i < N; ++i) idx[(size_t)i] = i;
    npysel::Sel s{dist, idx.data()};
    s.introselect(0, N, N - n_empty);                      // np.argpartition(distances, -n_empty)
    for (int q = 0; q < n_empty; ++q) out[q] = idx[(size_t)(N - 1 - q)];      // [:-n_empty-1:-1]
}

// The rare path of the M-step: stats[3] = n_empty > 0.  The E-step that was queued behind the M-step has returned at once (hold), so
// the labels and the centres `Cold` are still those the member sums were formed with.  Restates `_relocate_empty_clusters_dense`
// (sklearn/cluster/_k_means_common.pyx) followed by `_average_centers`; see kmeans.hip.  The relocation itself is integer
// arithmetic on the totals (host side, on a copy every rank holds identically): the far row's fixed-point coordinates leave its
// old cluster's sums -- exactly what the row had added -- and become the sums of the empty one.
// Sharded run: distances and labels stay on their ranks; the n_empty farthest rows of the WHOLE set are found one at a time with
// three small all-reduces each (largest distance; among its holders the lowest global row; the owner's label and coordinates) --
// the library's descending rule with rows counted globally (edmdc_set_kmeans_shard), so that every rank applies the same
// changes and the result is that of the unsharded run under the same rule.
static int kmeans_relocate(brov_ctx* c, int64_t N, int n, int k, const double* d_X, int64_t xstride, const double* mean_host, const double* mp,
                           const double* fix, const double* Cold, double* Cnew, double* c2, double* stats, double* prm, long long* red,
                           const int* Lc, const int* Pc) {
    double* d_dist = nullptr;
    int* d_lab = nullptr;
    long long* d_words = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_dist); (void)hipFree(d_lab); (void)hipFree(d_words); };
#define HIPCK_R(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { cleanup(); return hip_fail(c, e__, #call); } } while (0)
    HIPCK_R(hipMalloc((void**)&d_dist, (size_t)N * 8));
    HIPCK_R(hipMalloc((void**)&d_lab, (size_t)N * 4));
    HIPCK_R(hipMalloc((void**)&d_words, 64 * 8));
    HIPCK_R(launch_kmeans_reloc_dist(c->stream, N, n, d_X, xstride, mp, Cold, Lc, Pc, d_dist, d_lab));
    std::vector<double> dist((size_t)N);
    std::vector<int> lab((size_t)N);
    std::vector<long long> hred(kmeans_red_words(n, k));
    double hfix[32];
    HIPCK_R(d2h_copy(c, dist.data(), d_dist, (size_t)N * 8));
    HIPCK_R(d2h_copy(c, lab.data(), d_lab, (size_t)N * 4));
    HIPCK_R(d2h_copy(c, hred.data(), red, hred.size() * 8));
    HIPCK_R(d2h_copy(c, hfix, fix, sizeof hfix));
    HIPCK_R(hipStreamSynchronize(c->stream));
    const int np1 = n + 1;
    auto load = [&](int q, int j) { return ((__int128)hred[((size_t)q * np1 + j) * 2] << 42) + (__int128)hred[((size_t)q * np1 + j) * 2 + 1]; };
    auto store = [&](int q, int j, __int128 v) {
        hred[((size_t)q * np1 + j) * 2] = (long long)(v >> 42);
        hred[((size_t)q * np1 + j) * 2 + 1] = (long long)(v & (((__int128)1 << 42) - 1));
    };
    // words through the ranks (device buffer in, device buffer out: the callback's contract), synchronous: this path is rare
    auto exchange = [&](long long* w, int count, int op) -> int {
        if (!c->km_allreduce) return BROV_OK;
        if (h2d_copy(c, d_words, w, (size_t)count * 8) != hipSuccess) return BROV_ERR_HIP;
        if (c->km_allreduce(c->km_allreduce_user, d_words, count, op) != 0) return BROV_ERR_COMM;
        if (d2h_copy(c, w, d_words, (size_t)count * 8) != hipSuccess) return BROV_ERR_HIP;
        return hipStreamSynchronize(c->stream) == hipSuccess ? BROV_OK : BROV_ERR_HIP;
    };
#define EXCH(w, count, op) do { int rc__ = exchange((w), (count), (op)); if (rc__) { cleanup(); return fail(c, rc__, "edmdc_kmeans_lloyd: exchange of the relocation failed"); } } while (0)
    // empty_clusters = np.where(weight_in_clusters == 0)[0], ascending
    std::vector<int> new_ids;
    for (int q = 0; q < k; ++q)
        if (load(q, n) == 0) new_ids.push_back(q);
    const int n_empty = (int)new_ids.size();
    double dmax = 0.0;
    long long any_nan = 0;
    for (int64_t i = 0; i < N; ++i) { if (dist[i] != dist[i]) any_nan = 1; else if (dist[i] > dmax) dmax = dist[i]; }
    {
        long long w[2];
        std::memcpy(&w[0], &dmax, 8);                    // non-negative doubles order like their bit patterns
        w[1] = any_nan;
        EXCH(w, 2, 1);
        std::memcpy(&dmax, &w[0], 8);
        any_nan = w[1];
    }
    // one row: its quantised coordinates leave `oc` and become cluster `nc`
    auto apply = [&](int nc, int oc, const long long* q) {
        for (int j = 0; j < n; ++j) { store(oc, j, load(oc, j) - (__int128)q[j]); store(nc, j, (__int128)q[j]); }
        store(oc, n, load(oc, n) - 1);
        store(nc, n, (__int128)1);
    };
    auto quantise_row = [&](int64_t row, long long* q) -> int {
        double x[16];
        if (hipMemcpy(x, d_X + row * xstride, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        for (int j = 0; j < n; ++j) {
            const double xc = x[j] - (mean_host ? mean_host[j] : 0.0);
            q[j] = (xc - xc == 0.0) ? (long long)std::nearbyint(xc * hfix[j]) : 0ll;      // x s is exact (s = 2^m): the device's FMA rounds the same way
        }
        return 0;
    };
    // `if np.max(distances) == 0: return` -- more clusters than distinct samples: nothing to relocate to
    if (n_empty > 0 && (dmax != 0.0 || any_nan)) {
        if (!c->km_allreduce) {
            if (n_empty > N) { cleanup(); return fail(c, BROV_ERR_ARG, "edmdc_kmeans_lloyd: more empty clusters than samples"); }
            std::vector<int64_t> far((size_t)n_empty);
            if (c->far_select) {
                if (c->far_select(c->far_select_user, dist.data(), N, n_empty, far.data()) != 0) { cleanup(); return fail(c, BROV_ERR_ARG, "edmdc_kmeans_lloyd: the far-sample callback failed"); }
                for (int64_t r : far) if (r < 0 || r >= N) { cleanup(); return fail(c, BROV_ERR_ARG, "edmdc_kmeans_lloyd: the far-sample callback returned a row out of range"); }
            } else {
                far_select_default(dist.data(), N, n_empty, far.data());
            }
            for (int q = 0; q < n_empty; ++q) {
                long long qv[16];
                if (quantise_row(far[q], qv)) { cleanup(); return fail(c, BROV_ERR_HIP, "edmdc_kmeans_lloyd: reading a far row failed"); }
                apply(new_ids[q], lab[(size_t)far[q]], qv);
            }
        } else {
            // Sharded: the selection rule looks at the distances of ALL rows in global row order (np.argpartition's result depends on where
            // the values sit, not only on what they are), so the ranks' distance arrays are put side by side first -- one all-reduce (sum)
            // of N_global words in which every rank fills its own rows with the bit patterns and leaves zeros elsewhere (80 MB at 1e7
            // rows: this path runs when a cluster runs empty, i.e. hardly ever) -- and every rank then applies the same rule to the same
            // array: the rows of the unsharded run, callback or library rule alike.
            const int64_t Ng = c->km_n_global > 0 ? (int64_t)c->km_n_global : N;
            if (c->km_row_offset + N > Ng) { cleanup(); return fail(c, BROV_ERR_ARG, "edmdc_kmeans_lloyd: edmdc_set_kmeans_shard does not cover this rank's rows"); }
            if (n_empty > Ng) { cleanup(); return fail(c, BROV_ERR_ARG, "edmdc_kmeans_lloyd: more empty clusters than samples"); }
            std::vector<double> gdist((size_t)Ng, 0.0);
            {
                long long* d_all = nullptr;
                if (hipMalloc((void**)&d_all, (size_t)Ng * 8) != hipSuccess) { cleanup(); return fail(c, BROV_ERR_NOMEM, "edmdc_kmeans_lloyd: no memory for the gathered distances"); }
                bool ok = hipMemsetAsync(d_all, 0, (size_t)Ng * 8, c->stream) == hipSuccess &&
                          hipMemcpyAsync(d_all + c->km_row_offset, d_dist, (size_t)N * 8, hipMemcpyDeviceToDevice, c->stream) == hipSuccess;
                const int arc = ok ? c->km_allreduce(c->km_allreduce_user, d_all, Ng, 0) : 1;
                ok = ok && arc == 0 && d2h_copy(c, gdist.data(), d_all, (size_t)Ng * 8) == hipSuccess &&
                     hipStreamSynchronize(c->stream) == hipSuccess;
                (void)hipFree(d_all);
                if (!ok) { cleanup(); return fail(c, arc ? BROV_ERR_COMM : BROV_ERR_HIP, "edmdc_kmeans_lloyd: gathering the distances of the ranks failed"); }
            }
            std::vector<int64_t> far((size_t)n_empty);
            if (c->far_select) {
                if (c->far_select(c->far_select_user, gdist.data(), Ng, n_empty, far.data()) != 0) { cleanup(); return fail(c, BROV_ERR_ARG, "edmdc_kmeans_lloyd: the far-sample callback failed"); }
                for (int64_t r : far) if (r < 0 || r >= Ng) { cleanup(); return fail(c, BROV_ERR_ARG, "edmdc_kmeans_lloyd: the far-sample callback returned a row out of range"); }
            } else {
                far_select_default(gdist.data(), Ng, n_empty, far.data());
            }
            for (int q = 0; q < n_empty; ++q) {
                // the owner of the row sends its label and fixed-point coordinates (sum: the other ranks hold zeros)
                const int64_t lrow = far[q] - c->km_row_offset;
                const bool owner = lrow >= 0 && lrow < N;
                long long msg[18];
                for (long long& v : msg) v = 0;
                if (owner) {
                    if (quantise_row(lrow, msg)) { cleanup(); return fail(c, BROV_ERR_HIP, "edmdc_kmeans_lloyd: reading a far row failed"); }
                    msg[16] = lab[(size_t)lrow];
                    msg[17] = 1;
                }
                EXCH(msg, 18, 0);
                if (msg[17] != 1) { cleanup(); return fail(c, BROV_ERR_COMM, "edmdc_kmeans_lloyd: the ranks disagree about a relocated row"); }
                apply(new_ids[q], (int)msg[16], msg);
            }
        }
        HIPCK_R(h2d_copy(c, red, hred.data(), hred.size() * 8));
        HIPCK_R(hipStreamSynchronize(c->stream));
        ++c->kmeans_relocations;
    }
    HIPCK_R(launch_kmeans_average(c->stream, n, k, red, fix, Cold, Cnew, c2, stats, prm, 1));
    HIPCK_R(hipStreamSynchronize(c->stream));
#undef EXCH
#undef HIPCK_R
    cleanup();
    return BROV_OK;
}

int edmdc_kmeans_lloyd_dev(brov_ctx* c, int64_t N, int n, int k, const double* d_X, int64_t xstride, const double* mean_host,
                           double* d_C, int max_iter, double tol_abs, int32_t* d_labels, double* inertia, int* n_iter) {
    if (!c || N < 1 || n < 1 || n > 15 || k < 1 || !d_X || !d_C || !d_labels || max_iter < 1 || xstride < n || (size_t)k * (n + 1) * 8 > 150 * 1024)
        return fail(c, BROV_ERR_ARG, "edmdc_kmeans_lloyd_dev: bad argument (need 1<=n<=15, k*(n+1)*8 <= 150 KiB of LDS)");
    DeviceGuard g(c);
    const int variant = c->kmeans_variant & 3;
    const bool scalar_records = (c->kmeans_variant & KMV_SCALAR_RECORDS) != 0;     // the E-step kernel with scalar centre records (kmeans.hip)
    const int nb = kmeans_blocks(N, n, k, scalar_records);
    const int nparts = nb * kmeans_epochs(N, n, k, scalar_records);
    const size_t pwords = kmeans_partial_words(N, n, k, scalar_records), rwords = kmeans_red_words(n, k);
    Arena a(c);
    const bool filter = variant != 1 && k >= 64;       // below one mask word there is nothing to skip
    // sample order (sortperm.hip): the filter decides per wave of 64 consecutive samples, so the loop keeps a permutation that orders the
    // samples by (label, distance to the centre), re-sorted when enough labels have moved; the LDS / DPP kernel reads its rows
    // through it.  Worth it only at size.
    // k = 513 ... 1024 (n = 12 / 13): beyond the LDS / DPP kernel's table; the packed-fp32 kernel (kmeans_assign_pk_kernel) reads the same order
    // and the same sorted rows with only the member sums in the LDS.  + 128 (screening off) or + 4 keeps the scalar-record kernel there.
    const bool big = !scalar_records && !kmeans_reads_through_perm(n, k, false) && kmeans_pk_supported(n, k) && (c->kmeans_variant & (KMV_MASK_FILTER | KMV_NO_SCREENING)) == 0;
    const bool sorting = filter && variant == 0 && (kmeans_reads_through_perm(n, k, scalar_records) || big) && N >= ((int64_t)1 << 18) && N < ((int64_t)1 << 31);
    const size_t sort_tmp = sorting ? kmeans_sort_temp_bytes(N) : 0;
    // single-reference form of the filter (kmeans.hip, round 4): sorted rows of the centre distances for the LDS / DPP kernel; + 16 in the
    // k-means variant keeps the mask form of round 3 alone (the independent second implementation of the filter)
    const bool nbr = filter && (kmeans_reads_through_perm(n, k, scalar_records) || (big && sorting)) && (c->kmeans_variant & KMV_MASK_FILTER) == 0;
    const size_t kp_ = (size_t)((k + 255) & ~255);
    // third form of the E-step (kmeans.hip): candidates screened in packed fp32, exact arithmetic for the winner; for the loop's sorted
    // order.  + 64 in the k-means variant selects it.
    const bool pk = nbr && sorting && ((c->kmeans_variant & KMV_PK_STANDALONE) != 0 || big) && kmeans_pk_supported(n, k);
    // ... and the same screening as an evaluation path INSIDE the LDS / DPP kernel (its prefetching, its LDS-resident exact records): the
    // default for the sorted loop; + 128 switches it off (fp64 evaluation of every candidate, the form of round 3 and early round 4)
    const bool pk_lds = nbr && sorting && !pk && (c->kmeans_variant & KMV_NO_SCREENING) == 0 && kmeans_pk_supported(n, k) && kmeans_reads_through_perm(n, k, scalar_records);
    // distance bounds (kmeans.hip: kmeans_bounds_kernel): once few labels change per iteration, an E-step visits only the samples whose
    // bounds fail and the M-step adds their CHANGES to the totals it keeps; + 256 in the k-means variant (or BROV2_KM_BOUNDS=0) switches it off
    const bool bnd = pk_lds && (c->kmeans_variant & KMV_NO_BOUNDS) == 0 && kmeans_bounds_list_words(N) < ((size_t)1 << 31);      // (32-bit offsets into the list)
    const size_t lwords = bnd ? kmeans_bounds_list_words(N) : 0;
    const int pf_pairs = (pk_lds && !pk) ? kmeans_lds_pf_pairs() : 0;      // pair records per row worth building (0: all of them)
    const int nb_pk = pk ? kmeans_pk_blocks(N, n, k) : 0;
    const int nparts_pk = pk ? nb_pk * kmeans_pk_epochs(N, n, k) : 0;
    const size_t pwords_pk = (size_t)nparts_pk * k * (n + 1);
    const size_t pw_max = pwords > pwords_pk ? pwords : pwords_pk;
    const int nb_max = nb > nb_pk ? nb : nb_pk;
    int rc = a.reserve(Arena::al(pw_max * 8) + Arena::al(rwords * 8) + Arena::al(nb_max * 8) + Arena::al(nb_max * 4) + Arena::al((size_t)k * 16 * 8) +
                       ((pk || pk_lds) ? Arena::al(k * (kp_ / 2) * 32 * 4) : 0) + 256 +
                       2 * Arena::al((size_t)k * n * 8) + Arena::al(filter ? (size_t)k * ((k + 255) & ~255) * 4 : 8) +
                       (nbr ? Arena::al(k * kp_ * 8) : 0) +
                       (sorting ? 9 * Arena::al((size_t)N * 4) + Arena::al(sort_tmp + 256) : 0) +
                       (bnd ? 5 * Arena::al((size_t)N * 4) + Arena::al(lwords * 4) + Arena::al(rwords * 8) + 4 * Arena::al((size_t)k * 16 + 64) + 1024 : 0) +
                       Arena::al(kmeans_mstep_scratch_doubles(k) * 8) + 8192);
    if (rc) return rc;
    unsigned long long* partial = a.take<unsigned long long>(pw_max);
    long long* red = a.take<long long>(rwords);
    double* binert = a.take<double>(nb_max);
    float* Dc = a.take<float>(filter ? (size_t)k * ((k + 255) & ~255) : 1);
    unsigned long long* Nk = nbr ? a.take<unsigned long long>(k * kp_) : nullptr;      // sorted rows of the centre distances (keys)
    int* bchg = a.take<int>(nb_max);
    float* Pf = (pk || pk_lds) ? a.take<float>(k * (kp_ / 2) * 32) : nullptr;         // float pair records of the sorted rows
    double* c2 = a.take<double>((size_t)k * 16);       // packed centre table (kmeans.hip)
    double* Cb[2] = {a.take<double>((size_t)k * n), a.take<double>((size_t)k * n)};     // centres of this / the next iteration
    double* stats = a.take<double>(16);                // [0] squared shift, [1] inertia, [2] changed labels, [3] empty clusters; [4..11] = prm
    double* fix = a.take<double>(32);                  // fixed-point scales of the member sums
    unsigned long long* rng = a.take<unsigned long long>(16);
    double* dmean = a.take<double>(16);
    double* ms_scratch = a.take<double>(kmeans_mstep_scratch_doubles(k));      // the fused M-step's per-centre shifts / flags and its arrival ticket
    int *Ls[2] = {nullptr, nullptr}, *Ps[2] = {nullptr, nullptr};
    float* d2 = nullptr;
    float *d2s[2] = {nullptr, nullptr}, *ubs[2] = {nullptr, nullptr}, *lbs[2] = {nullptr, nullptr};      // per position: two copies each, swapped by a re-sort
    long long* tot = nullptr;
    float *shiftc = nullptr, *mvd = nullptr, *rw2 = nullptr;
    int *blist = nullptr, *nlist = nullptr;
    unsigned *kin = nullptr, *kout = nullptr, *vin = nullptr, *vout = nullptr;
    void* stmp = nullptr;
    if (sorting) {
        for (int q = 0; q < 2; ++q) { Ls[q] = a.take<int>(N); Ps[q] = a.take<int>(N); }
        d2 = a.take<float>(N);
        kin = a.take<unsigned>(N); kout = a.take<unsigned>(N); vin = a.take<unsigned>(N); vout = a.take<unsigned>(N);
        stmp = a.take<char>(sort_tmp + 256);
    }
    if (bnd) {
        d2s[0] = d2; d2s[1] = a.take<float>(N);
        for (int q = 0; q < 2; ++q) { ubs[q] = a.take<float>(N); lbs[q] = a.take<float>(N); }
        blist = a.take<int>(lwords);
        tot = a.take<long long>(rwords);
        shiftc = a.take<float>((size_t)k + kmeans_bounds_tail());
        mvd = a.take<float>((size_t)k * 4 + 4);
        rw2 = a.take<float>((size_t)k + 4);
        nlist = a.take<int>(64);
    }
    if (mean_host) {
        HIPCK(c, h2d_copy(c, dmean, mean_host, n * 8));
        HIPCK(c, hipStreamSynchronize(c->stream));
    }
    HIPCK(c, hipMemsetAsync(d_labels, 0xFF, N * sizeof(int32_t), c->stream));
    if (!c->h_stats) {
        HIPCK(c, hipHostMalloc((void**)&c->h_stats, 8 * sizeof(double), hipHostMallocMapped));
        std::memset(c->h_stats, 0, 8 * sizeof(double));
        HIPCK(c, hipHostGetDevicePointer((void**)&c->d_stats_map, c->h_stats, 0));      // the M-step stores the iteration's statistics there itself
    }
#ifdef BROV2_EXPERIMENTS      // the side stream serves the cdist_beside experiment alone (BROV2_KM_CDIST_BESIDE below): nothing of it in the default build
    if (!c->side[0]) HIPCK(c, hipStreamCreateWithFlags(&c->side[0], hipStreamNonBlocking));
    if (!c->ev_fork) HIPCK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    if (!c->ev_join[0]) HIPCK(c, hipEventCreateWithFlags(&c->ev_join[0], hipEventDisableTiming));
#endif
    HIPCK(c, hipMemsetAsync(ms_scratch, 0, kmeans_mstep_scratch_doubles(k) * 8, c->stream));
    CallTimer t(c);
    double* prm = stats + 4;
    const double* mp = mean_host ? dmean : nullptr;
    // one pass over the samples before the loop: the range of every coordinate (scales of the integer member sums) and max |x|^2
    // (margins of the candidate filter); a sharded run takes the maximum over its ranks here
    HIPCK(c, launch_kmeans_range(c->stream, N, n, d_X, xstride, mp, rng));
    if (c->km_allreduce && c->km_allreduce(c->km_allreduce_user, rng, 16, 1) != 0) return fail(c, BROV_ERR_COMM, "edmdc_kmeans_lloyd: all-reduce (max) failed");
    HIPCK(c, launch_kmeans_scale(c->stream, n, rng, fix, prm));
    HIPCK(c, hipMemcpyAsync(Cb[0], d_C, (size_t)k * n * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPCK(c, launch_kmeans_c2(c->stream, n, k, Cb[0], c2));
    // The E-step of iteration it+1 is queued before the host looks at the statistics of iteration it: if the loop goes on it
    // is the next E-step; if the shift criterion or max_iter ends it, it is the final E-step scikit-learn runs to make the
    // labels consistent with the last centres; under strict convergence the centres did not move and it rewrites the same
    // labels.  The read-back and the host's round trip hide behind it.  (With an empty cluster the queued E-step returns at
    // once -- hold -- and is queued again after the relocation.)
    bool strict = false;
    int it = 0;
    double hs[4] = {0, 0, 0, 0};
    // current sample order: labels per position (the caller's label array until the first sort)
    int* Lc = d_labels;
    const int* Pc = nullptr;                            // position -> row of the caller's X (nullptr = identity)
    int cur = 0, cc = 0;                                // cc: Cb[cc] = the centres the last E-step used
    bool want_sort = false;
    double moved = 0.0;                                 // labels changed since the last sort
    // (with the distance bounds an E-step visits a fraction of the samples while a sort still handles them all: later sorts -- 0.03: 169 ms per
    // 300 iterations at 1e7 x 12, k = 512; 0.06: 165; 0.1: 166.5; 0.2: 177.5)
    double sort_moved = bnd ? 2.0 * KM_SORT_MOVED : KM_SORT_MOVED, sort_rate = KM_SORT_RATE;
#ifdef BROV2_EXPERIMENTS
    if (const char* e = std::getenv("BROV2_KM_SORT_MOVED")) sort_moved = std::atof(e);      // (tools/time_lloyd_ab.py)
    if (const char* e = std::getenv("BROV2_KM_SORT_RATE")) sort_rate = std::atof(e);
#endif
    c->kmeans_relocations = 0;
    c->km_info[1] = c->km_info[2] = c->km_info[3] = 0;
    // the first E-step has no labels to start from: full scan; from then on the candidate filter (kmeans.hip) unless switched off
    HIPCK(c, launch_kmeans_assign(c->stream, N, n, k, d_X, xstride, mp, c2, Lc, partial, binert, bchg, nullptr, prm, fix, d2, scalar_records, nullptr));
    int e_nparts = nparts, e_nb = nb;                  // geometry of the E-step whose partials are waiting (the packed-fp32 kernel has its own)
    // distance bounds: bcur = the copies of d2 / ub / lb in the current order; bounds_valid: a full E-step has left bounds for every position
    // since the centres were last changed behind the loop's back (relocation); e_list: the E-step whose partials are waiting walked the list
    int bcur = 0;
    bool bounds_valid = false, e_list = false, any_list = false, use_list = false;
    const double bounds_rate = c->km_bounds_rate;       // list form once at most this share of the labels changed in the last summed iteration
    bool bounds_log = false;
    double bounds_beta = -1.0;
#ifdef BROV2_EXPERIMENTS
    bounds_log = std::getenv("BROV2_KM_BOUNDS_LOG") != nullptr;
    if (const char* e = std::getenv("BROV2_KM_BOUNDS_BETA")) bounds_beta = std::atof(e);      // (tools/time_lloyd_ab.py)
#endif
    const double n_all = (c->km_allreduce && c->km_n_global > 0) ? (double)c->km_n_global : (double)N;
    // (the centre distances on a side stream beside the bounds pass: measured and left off -- both kernels slow each other down, 75 + 69 us
    // against 31 + 61 one after the other, and with the two event hand-overs 300 iterations take 153 ms instead of 150.5)
    bool cdist_forked = false, cdist_beside = false;
#ifdef BROV2_EXPERIMENTS
    if (const char* e = std::getenv("BROV2_KM_CDIST_BESIDE")) cdist_beside = std::atoi(e) != 0;
#endif
    auto e_step = [&](bool filtered) -> hipError_t {
        if (filtered && pk) {
            if (cdist_forked) { cdist_forked = false; if (hipStreamWaitEvent(c->stream, c->ev_join[0], 0) != hipSuccess) return hipErrorUnknown; }
            e_nparts = nparts_pk; e_nb = nb_pk;
            return launch_kmeans_assign_pk(c->stream, N, n, k, d_X, xstride, mp, c2, Lc, partial, binert, bchg, prm, fix, d2, Pc, Nk, Pf);
        }
        e_nparts = nparts; e_nb = nb;
        KmBounds kb;
        const bool with_bounds = bnd && filtered && Pc != nullptr;
        e_list = false;
        if (with_bounds) {
            kb.ub = ubs[bcur]; kb.lb = lbs[bcur]; kb.beta = bounds_beta; kb.shiftc = shiftc; kb.mvd = mvd; kb.rw2 = rw2; kb.list = blist; kb.nlist = nlist;
            kb.use_list = use_list && bounds_valid;
            if (kb.use_list) {
                hipError_t e = launch_kmeans_bounds(c->stream, N, k, Lc, kb, prm);
                if (e != hipSuccess) return e;
                e_list = any_list = true;
                ++c->km_info[3];
            }
            bounds_valid = true;                        // (a full pass writes them all; a list pass keeps them)
        }
        if (cdist_forked) {                             // the centre distances were queued on the side stream (beside the bounds pass): join
            cdist_forked = false;
            if (hipStreamWaitEvent(c->stream, c->ev_join[0], 0) != hipSuccess) return hipErrorUnknown;
        }
        return launch_kmeans_assign(c->stream, N, n, k, d_X, xstride, mp, c2, Lc, partial, binert, bchg, filtered && filter ? Dc : nullptr, prm, fix,
                                    bnd ? d2s[bcur] : d2, scalar_records, filtered ? Pc : nullptr, filtered ? Nk : nullptr, filtered && pk_lds ? Pf : nullptr,
                                    with_bounds ? &kb : nullptr);
    };
    for (it = 1; it <= max_iter; ++it) {
        // the M-step: one launch (kmeans.hip: kmeans_mstep_kernel) -- sums, centres, the iteration's statistics straight into the host's
        // pinned block; a sharded run all-reduces the sums between its two halves
        KmMstepArgs ma;
        ma.nparts = e_nparts; ma.nblocks = e_nb; ma.n = n; ma.k = k; ma.partial = partial; ma.block_inertia = binert; ma.block_changed = bchg;
        ma.red = red; ma.tot = tot; ma.delta = e_list ? 1 : 0; ma.nlist = nlist; ma.fix = fix; ma.Cold = Cb[cc]; ma.Cnew = Cb[cc ^ 1]; ma.Ct = c2;
        ma.stats = stats; ma.prm = prm; ma.shiftc = shiftc; ma.mvd = mvd; ma.scratch = ms_scratch; ma.hstats = c->d_stats_map;
        ma.seq = (c->km_seq += 1.0);                  // (never repeats within a context: a stale block cannot pass for this iteration's)
        // the M-step's one-block global part rides on the centre-distance launch that follows (one launch less per iteration); not when
        // that launch goes to the side stream, where the bounds pass would not wait for it
        ma.tail_deferred = filter && !cdist_beside;
        if (c->km_allreduce) {
            HIPCK(c, launch_kmeans_mstep(c->stream, ma, 1));
            if (c->km_allreduce(c->km_allreduce_user, red, (int64_t)rwords, 0) != 0) return fail(c, BROV_ERR_COMM, "edmdc_kmeans_lloyd: all-reduce (sum) failed");
            HIPCK(c, launch_kmeans_mstep(c->stream, ma, 2));
        } else {
            HIPCK(c, launch_kmeans_mstep(c->stream, ma, 3));
        }
        if (want_sort) {
            // (label, distance) of the E-step that has just been summed up; labels and permutation move together
            if (bnd) {
                // (the bounds and the sort key's distance move with their samples; before the first sort there are no bounds yet)
                const bool have = bounds_valid;
                HIPCK(c, launch_kmeans_resort(c->stream, N, Lc, Ls[cur], Pc, Ps[cur], d2s[bcur], kin, kout, vin, vout, stmp, sort_tmp, d2s[bcur ^ 1],
                                              have ? ubs[bcur] : nullptr, have ? ubs[bcur ^ 1] : nullptr, have ? lbs[bcur] : nullptr, have ? lbs[bcur ^ 1] : nullptr));
                bcur ^= 1;
            } else {
                HIPCK(c, launch_kmeans_resort(c->stream, N, Lc, Ls[cur], Pc, Ps[cur], d2, kin, kout, vin, vout, stmp, sort_tmp));
            }
            Lc = Ls[cur]; Pc = Ps[cur];
            cur ^= 1;
            want_sort = false;
            moved = 0.0;
            if (c->km_info[1]++ == 0) c->km_info[2] = it;
        }
        // hs still holds the statistics of the iteration before this one (the host has not waited yet): few changed labels -> list form
        use_list = bnd && it >= 2 && hs[2] <= bounds_rate * n_all;
        if (filter) {
            // centre distances, sorted rows, pair records: 32 us of short dependent phases in k blocks, needed by the E-step but not by the
            // bounds pass (whose mover distances the M-step has left): on the side stream, beside it
            const bool beside = cdist_beside && bnd && use_list && bounds_valid && Pc != nullptr;
            hipStream_t cs_ = beside ? c->side[0] : c->stream;
            if (beside) {
                HIPCK(c, hipEventRecord(c->ev_fork, c->stream));
                HIPCK(c, hipStreamWaitEvent(c->side[0], c->ev_fork, 0));
            }
            HIPCK(c, launch_kmeans_cdist(cs_, n, k, c2, Dc, Nk, Pf, nullptr, nullptr, rw2, pf_pairs, ma.tail_deferred ? &ma : nullptr));
            if (beside) {
                HIPCK(c, hipEventRecord(c->ev_join[0], c->side[0]));
                cdist_forked = true;
            }
        }
        HIPCK(c, e_step(true));
        { const int rcw = wait_stats(c, ma.seq); if (rcw) return rcw; }
        for (int q = 0; q < 4; ++q) hs[q] = c->h_stats[q];
        if (hs[3] > 0.0) {
            // empty clusters: relocate (the queued E-step did nothing), average again, and queue the E-step again
            rc = kmeans_relocate(c, N, n, k, d_X, xstride, mean_host, mp, fix, Cb[cc], Cb[cc ^ 1], c2, stats, prm, red, Lc, Pc);
            if (rc) return rc;
            HIPCK(c, d2h_copy(c, c->h_stats, stats, sizeof hs));
            HIPCK(c, hipStreamSynchronize(c->stream));
            hs[0] = c->h_stats[0];
            if (filter) HIPCK(c, launch_kmeans_cdist(c->stream, n, k, c2, Dc, Nk, Pf, shiftc, mvd, rw2, pf_pairs));
            bounds_valid = false;                       // relocated centres jumped: a full E-step, whose sums start the totals afresh
            HIPCK(c, e_step(true));
        }
        cc ^= 1;
        if (bounds_log && e_list) {
            int hn[64] = {0};
            HIPCK(c, hipMemcpy(hn, nlist, sizeof hn, hipMemcpyDeviceToHost));
            std::fprintf(stderr, "[kmeans bounds] iteration %d: changed %.0f, next E-step walks %d of %lld positions (%d of them in front)\n", it, hs[2],
                         hn[0] + hn[KM_NL_FRONT] + hn[KM_NL_BACK], (long long)N, hn[KM_NL_FRONT]);
        }
        if (hs[2] == 0.0) { strict = true; break; }      // labels unchanged (sklearn's strict convergence)
        if (hs[0] <= tol_abs) break;
        if (sorting) {
            // hs[2] = labels changed by the E-step before the one just queued.  Sort once 1 % of the samples have changed label since
            // the last sort (a fresh order costs ~55 candidates per wave, the caller's ~130; a sort costs less than half an E-step),
            // but not while more than 1 % still change per iteration -- such an order is stale at once.  Thresholds from scans at
            // 10^7 x 12, k = 512 (tools/attic/run_lloyd_variants.sh): 314 ms per 300 iterations; 0.02 / 0.01: 322 ms; 0.005 / 0.005: 331 ms;
            // every iteration: 384 ms; the caller's order: 423 ms.  Round 4, single-reference filter (a moved label widens the wave's
            // radius a little instead of opening a second candidate set: the order decays more slowly): moved / rate 0.01 / 0.01: 305 ms;
            // 0.02 / 0.2: 290; 0.03 / 0.01: 294; 0.05 / 0.05: 288; 0.08 / 0.01: 299; 0.15 / 0.01: 314; 0.005 / 0.2: 326; never: 484
            // (mask form alone: 313 / - / 313 / 326 / 331 / 355 / - / 557).  An order made while a fifth of the labels still move
            // per iteration buys nothing (rate 0.2 or 1.0 at moved 0.01: 303-304 ms).  Once a wave with a few moved labels sums its main
            // label over the wave (kmeans.hip: KM2_SUM_MIN) an early order pays after all: 0.03 / 0.05: 274 ms; 0.03 / 0.2: 263.5; 0.03 / 1.0: 264;
            // 0.05 / 0.2: 264; 0.08 / 1.0: 273 (one box).
            // (hs[2] counts the changed labels of ALL ranks in a sharded run: both gates are shares of all rows -- round-4 advice: against
            // the local row count the rate gate was `world` times stricter and ranks with unequal shards sorted at different iterations)
            moved += hs[2];
            if (moved >= sort_moved * n_all && hs[2] <= sort_rate * n_all) want_sort = true;
        }
    }
    if (it > max_iter) it = max_iter;
    double in = hs[1];
    if (any_list) {
        // inertia over ALL samples: one plain pass with the final centres (the list form sums only what it visits; the labels it
        // rewrites are the ones already there)
        use_list = false;
        HIPCK(c, e_step(true));
        strict = false;
    }
    HIPCK(c, hipMemcpyAsync(d_C, Cb[cc], (size_t)k * n * 8, hipMemcpyDeviceToDevice, c->stream));
    if (Pc) HIPCK(c, launch_kmeans_unpermute(c->stream, N, Pc, Lc, d_labels));      // labels back in the caller's order
    if (!strict) {   // labels / inertia consistent with the final centres: the E-step already queued
        std::vector<double> hb(e_nb);
        HIPCK(c, d2h_copy(c, hb.data(), binert, e_nb * 8));
        HIPCK(c, hipStreamSynchronize(c->stream));
        in = 0.0;
        for (double v : hb) in += v;
    } else {
        HIPCK(c, hipStreamSynchronize(c->stream));
    }
    if (inertia) *inertia = in;
    if (n_iter) *n_iter = it;
    return BROV_OK;
}

// ---- k-means++ seeding (scikit-learn's algorithm, scikit-learn's random numbers) ----------------------------
int edmdc_kmeanspp_dev(brov_ctx* c, int64_t N, int n, int k, const double* d_X, int64_t xstride, const double* mean_host,
                       int64_t first_index, int n_trials, const double* uniforms_host, double* d_C, int64_t* indices_host) {
    const bool sharded = c && c->km_allreduce && c->km_world > 1;
    const int64_t Ng = sharded ? (int64_t)c->km_n_global : N;         // the first index and k refer to the rows of all ranks
    if (!c || N < 1 || n < 1 || n > 16 || k < 1 || k > Ng || !d_X || !d_C || xstride < n || first_index < 0 || first_index >= Ng ||
        n_trials < 1 || n_trials > 16 || (k > 1 && !uniforms_host) || N > 30000000)
        return fail(c, BROV_ERR_ARG, "edmdc_kmeanspp_dev: bad argument (need 1<=n<=16, 1<=n_trials<=16, k<=N<=3e7)");
    DeviceGuard g(c);
    const size_t nsum = kmeanspp_sum_doubles(N);
    const size_t nu = (size_t)(k > 1 ? k : 1) * n_trials;        // one spare row: the last round's (unused) draw pointer stays in bounds
    Arena a(c);
    // a float copy of the coordinates screens out the rows a round cannot affect (kmeans.hip, pp_round_kernel); + 8 in the k-means
    // variant: every row goes through the fp64 path
    const bool screening = (c->kmeans_variant & KMV_PP_UNSCREENED) == 0;
    const size_t nshard = kmeanspp_shard_doubles(sharded ? c->km_world : 1);
    int rc = a.reserve(Arena::al((size_t)N * n * 8) + 2 * Arena::al((size_t)N * 8) + Arena::al(nsum * 8) + Arena::al(nu * 8) + Arena::al(nshard * 8) +
                       Arena::al((size_t)k * 8) + Arena::al(kmeanspp_state_bytes()) + (screening ? Arena::al((size_t)N * n * 4) + Arena::al(kmeanspp_row_bytes(N, n)) : 0) + 8192);
    if (rc) return rc;
    double* Xt = a.take<double>((size_t)N * n);
    double* xsq = a.take<double>(N);
    double* closest = a.take<double>(N);
    double* dsum = a.take<double>(nsum);
    double* du = a.take<double>(nu);
    long long* dind = a.take<long long>(k);
    char* state = a.take<char>(kmeanspp_state_bytes());
    double* dmean = a.take<double>(16);
    float* Xf = screening ? a.take<float>((size_t)N * n) : nullptr;
    char* rowbuf = screening ? a.take<char>(kmeanspp_row_bytes(N, n)) : nullptr;      // row level of the screening (kmeans.hip: PPRows)
    double* dshard = a.take<double>(nshard);
    if (mean_host) HIPCK(c, h2d_copy(c, dmean, mean_host, n * 8));
    if (k > 1) HIPCK(c, h2d_copy(c, du, uniforms_host, (size_t)(k - 1) * n_trials * 8));
    HIPCK(c, hipStreamSynchronize(c->stream));          // the host buffers may be temporaries of the caller
    {
        CallTimer t(c);
        if (sharded || (c->kmeans_variant & KMV_PP_SHARD_KERNELS)) {
            // rows sharded over ranks (edmdc_set_kmeans_allreduce / edmdc_set_kmeans_shard): two small exchanges per centre; + 32 in the
            // k-means variant sends a single rank through the same kernels (the tests' check of that path against the one above)
            int comm_failed = 0;
            HIPCK(c, launch_kmeanspp_sharded(c->stream, N, n, k, n_trials, d_X, xstride, mean_host ? dmean : nullptr, (long long)first_index, du, Xt, xsq,
                                             closest, dsum, state, d_C, dind, Xf, sharded ? c->km_world : 1, sharded ? c->km_rank : 0,
                                             sharded ? c->km_row_offset : 0, dshard, c->km_allreduce, c->km_allreduce_user, &comm_failed, rowbuf));
            if (comm_failed) return fail(c, BROV_ERR_COMM, "edmdc_kmeanspp_dev: an exchange between the ranks failed");
        } else {
            HIPCK(c, launch_kmeanspp(c->stream, N, n, k, n_trials, d_X, xstride, mean_host ? dmean : nullptr, (long long)first_index, du, Xt, xsq,
                                     closest, dsum, state, d_C, dind, Xf, rowbuf));
        }
    }
    if (indices_host) {
        static_assert(sizeof(long long) == sizeof(int64_t), "index width");
        HIPCK(c, d2h_copy(c, indices_host, dind, (size_t)k * 8));
    }
    HIPCK(c, hipStreamSynchronize(c->stream));
    return BROV_OK;
}

int edmdc_col_stats_dev(brov_ctx* c, int64_t N, int n, const double* d_X, int64_t xstride, double* mean_host, double* var_host) {
    if (!c || N < 1 || n < 1 || n > 16 || !d_X || xstride < n) return fail(c, BROV_ERR_ARG, "edmdc_col_stats_dev: bad argument (1 <= n <= 16, x_stride >= n)");
    DeviceGuard g(c);
    const int nb = colstats_blocks(N);
    Arena a(c);
    int rc = a.reserve((size_t)nb * 16 * 8 + 16 * 8 + 1024);
    if (rc) return rc;
    double* d_part = a.take<double>((size_t)nb * 16);
    double* d_mean = a.take<double>(16);
    std::vector<double> part((size_t)nb * 16);
    double mean[16] = {0};
    auto total = [&](double* out) {            // the partial rows in block order
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int b = 0; b < nb; ++b) s += part[(size_t)b * 16 + j];
            out[j] = s / (double)N;
        }
    };
    HIPCK(c, launch_colstats(c->stream, N, n, d_X, xstride, nullptr, false, d_part));
    HIPCK(c, d2h_copy(c, part.data(), d_part, part.size() * 8));
    HIPCK(c, hipStreamSynchronize(c->stream));
    total(mean);
    if (mean_host) std::memcpy(mean_host, mean, (size_t)n * 8);
    if (var_host) {
        HIPCK(c, h2d_copy(c, d_mean, mean, 16 * 8));
        HIPCK(c, launch_colstats(c->stream, N, n, d_X, xstride, d_mean, true, d_part));
        HIPCK(c, d2h_copy(c, part.data(), d_part, part.size() * 8));
        HIPCK(c, hipStreamSynchronize(c->stream));
        total(var_host);
    }
    return BROV_OK;
}

int edmdc_kmeans_lloyd(brov_ctx* c, int64_t N, int n, int k, const double* X, const double* mean, double* C_io, int max_iter,
                       double tol_abs, int32_t* labels, double* inertia, int* n_iter) {
    if (!c || N < 1 || n < 1 || k < 1 || !X || !C_io) return fail(c, BROV_ERR_ARG, "edmdc_kmeans_lloyd: bad argument");
    DeviceGuard g(c);
    double *dX = nullptr, *dC = nullptr;
    int32_t* dL = nullptr;
    auto cleanup = [&]() { (void)hipFree(dX); (void)hipFree(dC); (void)hipFree(dL); };
    HIPCK_CLEAN(hipMalloc((void**)&dX, (size_t)N * n * 8));
    HIPCK_CLEAN(hipMalloc((void**)&dC, (size_t)k * n * 8));
    HIPCK_CLEAN(hipMalloc((void**)&dL, (size_t)N * 4));
    HIPCK_CLEAN(h2d_copy(c, dX, X, (size_t)N * n * 8));
    HIPCK_CLEAN(h2d_copy(c, dC, C_io, (size_t)k * n * 8));
    int rc = edmdc_kmeans_lloyd_dev(c, N, n, k, dX, n, mean, dC, max_iter, tol_abs, dL, inertia, n_iter);
    if (rc) { cleanup(); return rc; }
    HIPCK_CLEAN(d2h_copy(c, C_io, dC, (size_t)k * n * 8));
    if (labels) HIPCK_CLEAN(d2h_copy(c, labels, dL, (size_t)N * 4));
    HIPCK_CLEAN(hipStreamSynchronize(c->stream));
    cleanup();
    return BROV_OK;
}

}  // extern "C"
