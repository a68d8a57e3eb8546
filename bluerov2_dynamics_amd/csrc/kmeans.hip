// kmeans.hip -- Lloyd iterations on device for the RBF centres of KoopmanEDMDc.fit
// (Koopman/koopmanEDMDc.py:85,126 call sklearn.cluster.KMeans(n_clusters, n_init="auto", random_state=0)).
//
// scikit-learn stays the owner of the initialisation (k-means++, seeded) -- the host layer calls it --
// and this file restates what scikit-learn 1.7.2's `_kmeans_single_lloyd` iterates:
//   E-step: label_i = argmin_c (|c|^2 - 2 x_i.c)          (first minimum; |x_i|^2 is common to all c;
//           evaluated as argmax_c (x_i.c - |c|^2 / 2))
//   M-step: c <- mean of its members (an empty cluster keeps its centre: sklearn relocates it instead)
//   stop  : labels unchanged ("strict convergence") or sum |c_new - c_old|^2 <= tol, or max_iter
// on the mean-centred data (the host passes the column means), so that centres agree with
// scikit-learn's to rounding whenever no assignment is decided by the last bit.
//
// E-step kernel: lane = sample (its n coordinates in VGPRs), loop over centres whose coordinates arrive
// as wave-uniform scalar loads: 12 FMA + compare/select per (64 samples, centre).  Persistent 1024-thread blocks
// accumulate member sums and counts with LDS fp64 atomics and write one partial per block; a second
// kernel reduces the partials in block order and forms the new centres.
#include "brov2_kernels.h"
#include <cstdint>

namespace brov {

constexpr int KM_NMAX = 16;
constexpr int KM_BLOCKS = 512;        // persistent blocks (2 per CU: each holds a 53 KB LDS table of member sums at k = 512)
constexpr int KM_THREADS = 1024;      // 16 waves per block -> 8 waves per SIMD: the centre loop waits on its scalar loads once
                                      // per centre, and only other waves can fill that time (256-thread blocks: 4x slower)

typedef const double __attribute__((address_space(4)))* cdp;
// packed centre table: one 128-byte record per centre = n coordinates, |c|^2 / 2 in slot n, zeros (n <= 15)
constexpr int KM_CMAX = 15;
struct __attribute__((aligned(128))) Cen { double v[16]; };
typedef const Cen __attribute__((address_space(4)))* ccp;

template <int NS>
__global__ void __launch_bounds__(KM_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) kmeans_assign_kernel(int64_t N, int n, int k, const double* __restrict__ X, int64_t xstride,
                                                            const double* __restrict__ mean,
                                                            const double* __restrict__ Ct /* [k][16]: coordinates, |c|^2/2 at [n] */, int* __restrict__ labels,
                                                            double* __restrict__ partial /* [blocks][k][n+1] */,
                                                            double* __restrict__ block_inertia, int* __restrict__ block_changed) {
    extern __shared__ double sums[];                  // [k][n+1]: member sums and count
    const int np1 = n + 1;
    for (int i = threadIdx.x; i < k * np1; i += KM_THREADS) sums[i] = 0.0;
    __shared__ double sh_inertia[KM_THREADS / 64];
    __shared__ int sh_changed[KM_THREADS / 64];
    __syncthreads();
    const ccp T = (ccp)(unsigned long long)Ct;
    double inertia = 0.0;
    int changed = 0;
    for (int64_t base = (int64_t)blockIdx.x * KM_THREADS; base < N; base += (int64_t)gridDim.x * KM_THREADS) {
        const int64_t i = base + threadIdx.x;
        const bool live = i < N;
        const int64_t ii = live ? i : N - 1;
        double x[KM_NMAX], x2 = 0.0;
#pragma unroll
        for (int j = 0; j < KM_NMAX; ++j) {
            const bool on = NS > 0 ? (j < NS) : (j < n);
            x[j] = on ? X[ii * xstride + j] - (mean ? mean[j] : 0.0) : 0.0;
            x2 = fma(x[j], x[j], x2);
        }
        // argmin_c (|c|^2 - 2 x.c) as argmax_c (x.c - |c|^2 / 2): the half norm seeds the FMA chain, the running best is
        // one v_max, the index one compare + select -- 16 VALU instructions per centre instead of 19
        double best = -1.0e300;
        int bi = 0;
        auto eval = [&](const Cen& t, int c) {
            double dot = 0.0, dot1 = 0.0;                 // two chains: dependent fp64 FMAs do not issue back to back
            constexpr int NJ = NS > 0 ? NS : KM_CMAX;     // generic n: slots beyond n hold zeros (and the half norm, read below)
#pragma unroll
            for (int j = 0; j + 1 < NJ; j += 2) { dot = fma(x[j], t.v[j], dot); dot1 = fma(x[j + 1], t.v[j + 1], dot1); }
            if constexpr (NJ & 1) dot = fma(x[NJ - 1], t.v[NJ - 1], dot);
            const double hn = NS > 0 ? t.v[NS] : t.v[n];
            const double sc = fma(hn, -1.0, dot + dot1);
            bi = (sc <= best) ? bi : c;               // strict '>' to replace: the first extremum wins, like np.argmin
            asm("v_max_f64 %0, %1, %2" : "=v"(best) : "v"(best), "v"(sc));      // plain max: fmax() adds a canonicalising self-max
        };
        // two centres per trip: their four 64-byte scalar loads go out together and are waited for once
        int c = 0;
#pragma unroll 1
        for (; c + 1 < k; c += 2) {
            Cen a, b;
#pragma unroll
            for (int j = 0; j < 16; ++j) { a.v[j] = T[c].v[j]; b.v[j] = T[c + 1].v[j]; }
            eval(a, c);
            eval(b, c + 1);
        }
        if (c < k) {
            Cen a;
#pragma unroll
            for (int j = 0; j < 16; ++j) a.v[j] = T[c].v[j];
            eval(a, c);
        }
        if (live) {
            if (labels[i] != bi) ++changed;
            labels[i] = bi;
            inertia += fma(-2.0, best, x2);
            double* s = sums + bi * np1;
            for (int j = 0; j < n; ++j) atomicAdd(&s[j], x[j]);
            atomicAdd(&s[n], 1.0);
        }
    }
    // block reductions of inertia / changed
    for (int off = 32; off > 0; off >>= 1) {
        inertia += __shfl_down(inertia, off);
        changed += __shfl_down(changed, off);
    }
    if ((threadIdx.x & 63) == 0) { sh_inertia[threadIdx.x >> 6] = inertia; sh_changed[threadIdx.x >> 6] = changed; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double in = 0.0;
        int ch = 0;
        for (int w = 0; w < KM_THREADS / 64; ++w) { in += sh_inertia[w]; ch += sh_changed[w]; }
        block_inertia[blockIdx.x] = in;
        block_changed[blockIdx.x] = ch;
    }
    double* out = partial + (int64_t)blockIdx.x * k * np1;
    for (int i = threadIdx.x; i < k * np1; i += KM_THREADS) out[i] = sums[i];
}

// thread per (c, j): sum partials over blocks (fixed order), new centre, accumulate squared shift.
// stats[0] = sum of squared centre shifts, stats[1] = inertia, stats[2] = changed labels
__global__ void __launch_bounds__(256) kmeans_update_kernel(int nblocks, int n, int k, const double* __restrict__ partial,
                                                            const double* __restrict__ block_inertia, const int* __restrict__ block_changed,
                                                            double* __restrict__ C, double* __restrict__ Ct, double* __restrict__ stats) {
    const int np1 = n + 1;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);       // one wave per centre
    const int j = threadIdx.x & 63;
    double shift2 = 0.0;
    if (c < k) {
        double cnt = 0.0, sum = 0.0;
        for (int b = 0; b < nblocks; ++b) {
            const double* p = partial + ((int64_t)b * k + c) * np1;
            cnt += p[n];
            if (j < n) sum += p[j];
        }
        double nv = 0.0;
        if (j < n) {
            const double old = C[c * n + j];
            nv = cnt > 0.0 ? sum / cnt : old;
            C[c * n + j] = nv;
            const double dd = nv - old;
            shift2 = dd * dd;
        }
        double q = nv * nv;
        for (int off = 32; off > 0; off >>= 1) { q += __shfl_down(q, off); shift2 += __shfl_down(shift2, off); }
        q = __shfl(q, 0);
        if (j < 16) Ct[c * 16 + j] = j < n ? nv : (j == n ? 0.5 * q : 0.0);
        if (j == 0) atomicAdd(&stats[0], shift2);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double in = 0.0;
        long long ch = 0;
        for (int b = 0; b < nblocks; ++b) { in += block_inertia[b]; ch += block_changed[b]; }
        stats[1] = in;
        stats[2] = (double)ch;
    }
}

// packed table from the centres: Ct[c] = [coordinates | half squared norm | zeros]
__global__ void __launch_bounds__(256) kmeans_c2_kernel(int n, int k, const double* __restrict__ C, double* __restrict__ Ct) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= k) return;
    double s = 0.0;
    for (int j = 0; j < n; ++j) s = fma(C[c * n + j], C[c * n + j], s);
    for (int j = 0; j < 16; ++j) Ct[c * 16 + j] = j < n ? C[c * n + j] : (j == n ? 0.5 * s : 0.0);
}

// ---------------------------------------------------------------------------------------------------------
// k-means++ seeding on device: scikit-learn 1.7.2 `_kmeans_plusplus` (sklearn/cluster/_kmeans.py) with unit sample
// weights, which is what KMeans(n_init="auto", random_state=0).fit runs before Lloyd (Koopman/koopmanEDMDc.py:85,126).
// The random numbers stay scikit-learn's: the host draws them from numpy's RandomState in the order
// `_kmeans_plusplus` consumes them (one `choice` for the first centre, then `uniform(size=L)` per centre,
// L = 2 + int(log k)) and hands them over; everything that touches the N samples runs here:
//   closest_i = min_j d(x_i, c_j),  d(x, c) = max(0, (-2 x.c + |c|^2) + |x|^2)          (sklearn's _euclidean_distances)
//   candidates = searchsorted(cumsum(closest), u * pot)  (first index whose running sum reaches the value)
//   candidate with the smallest new potential sum_i min(closest_i, d(x_i, cand)) wins.
// The running sum is formed per 4096-sample chunk (tree) + a sequential pass over the chunk sums + a sequential pass
// inside the chunk that contains the value, not as one sequential np.cumsum: a candidate could differ from
// scikit-learn's only if a drawn value fell within ~1e-13 (relative) of a running-sum boundary.
// Per centre: pp_update_chunksum (1 pass over X), pp_pick (1 block), pp_candidates (1 pass over X), pp_select (1 block);
// nothing returns to the host until all k centres are chosen.
constexpr int PP_CHUNK = 4096;
constexpr int PP_LMAX = 16;           // trials per centre (k = 512: 8)
constexpr int PP_THREADS = 256;

struct PPState {                      // device-resident scalars of the seeding loop
    double pot;                       // current potential
    long long cand[PP_LMAX];          // candidate sample indices of this round
    long long last;                   // sample index of the centre chosen last
    int pad[2];
};

// The seeding makes 2 (k - 1) passes over the data, so the rows are first copied, centred, into a coordinate-major
// array Xt[j][i] (one strided read of X; every later access is a coalesced 512-byte wave load), with |x_i|^2 alongside
// (sklearn: row_norms(X, squared=True)).
template <int NS>
__global__ void __launch_bounds__(PP_THREADS) pp_transpose_kernel(int64_t N, int n, const double* __restrict__ X, int64_t xstride,
                                                                 const double* __restrict__ mean, double* __restrict__ Xt, double* __restrict__ xsq) {
    const int64_t i = (int64_t)blockIdx.x * PP_THREADS + threadIdx.x;
    if (i >= N) return;
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < KM_NMAX; ++j) {
        if (NS > 0 ? (j < NS) : (j < n)) {
            const double v = X[i * xstride + j] - (mean ? mean[j] : 0.0);
            Xt[(int64_t)j * N + i] = v;
            s += v * v;
        }
    }
    xsq[i] = s;
}

template <int NS>
__device__ __forceinline__ void pp_load_col(const double* __restrict__ Xt, int64_t N, int n, int64_t i, double x[KM_NMAX]) {
#pragma unroll
    for (int j = 0; j < KM_NMAX; ++j) x[j] = (NS > 0 ? (j < NS) : (j < n)) ? Xt[(int64_t)j * N + i] : 0.0;
}

// closest_i = min(closest_i, d(x_i, x_last))  (first = 1: closest_i = d) and the sum of every 4096-sample chunk.
// One block per chunk, 256 threads x 16 samples, fixed reduction tree.
template <int NS>
__global__ void __launch_bounds__(PP_THREADS) pp_update_chunksum_kernel(int64_t N, int n, const double* __restrict__ Xt,
                                                                       const double* __restrict__ xsq, const PPState* __restrict__ st, int first,
                                                                       double* __restrict__ closest, double* __restrict__ chunk_sum) {
    __shared__ double sh[PP_THREADS];
    const int64_t last = st->last;
    double c[KM_NMAX];
    pp_load_col<NS>(Xt, N, n, last, c);
    const double cc = xsq[last];
    double acc = 0.0;
    const int64_t base = (int64_t)blockIdx.x * PP_CHUNK;
#pragma unroll 1
    for (int q = 0; q < PP_CHUNK / PP_THREADS; ++q) {
        const int64_t i = base + q * PP_THREADS + threadIdx.x;
        if (i < N) {
            double x[KM_NMAX], dot = 0.0;
            pp_load_col<NS>(Xt, N, n, i, x);
#pragma unroll
            for (int j = 0; j < KM_NMAX; ++j) dot = fma(x[j], c[j], dot);
            double d = (-2.0 * dot + cc) + xsq[i];
            d = d > 0.0 ? d : 0.0;
            if (!first) { const double o = closest[i]; d = o < d ? o : d; }
            closest[i] = d;
            acc += d;
        }
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int off = PP_THREADS / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) chunk_sum[blockIdx.x] = sh[0];
}

// One block, one wave per trial: pot = sum of the chunk sums (round 0 only, else the winner's potential stands);
// value = u * pot; chunk = first chunk whose running sum reaches it; inside the chunk lane l owns 64 consecutive
// samples.  cand = first index with running sum >= value, clipped to N - 1 (np.searchsorted + np.clip).
__global__ void __launch_bounds__(64 * PP_LMAX) pp_pick_kernel(int64_t N, int nchunks, int L, const double* __restrict__ u,
                                                              const double* __restrict__ closest, const double* __restrict__ chunk_sum,
                                                              PPState* __restrict__ st, int set_pot) {
    extern __shared__ double prefix[];            // [nchunks + 1] exclusive running sums of the chunks
    __shared__ double seg[64 * PP_LMAX];
    {
        // running sums of the chunk sums, in chunk order: every thread adds up a contiguous segment, thread 0 chains the
        // segment totals, every thread then writes its segment's prefixes (fixed grouping -> same result every run)
        const int nt = blockDim.x, per = (nchunks + nt - 1) / nt;
        const int b0 = threadIdx.x * per, b1 = b0 + per < nchunks ? b0 + per : nchunks;
        double a = 0.0;
        for (int b = b0; b < b1; ++b) a += chunk_sum[b];
        seg[threadIdx.x] = a;
        __syncthreads();
        if (threadIdx.x == 0) {
            double run = 0.0;
            for (int t = 0; t < nt; ++t) { const double v = seg[t]; seg[t] = run; run += v; }
            prefix[nchunks] = run;
            if (set_pot) st->pot = run;
        }
        __syncthreads();
        double run = seg[threadIdx.x];
        for (int b = b0; b < b1; ++b) { prefix[b] = run; run += chunk_sum[b]; }
    }
    __syncthreads();
    const int trial = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (trial >= L) return;
    const double pot = set_pot ? prefix[nchunks] : st->pot;
    const double v = u[trial] * pot;
    // first chunk b with prefix[b + 1] >= v (binary search, wave-uniform)
    int lo = 0, hi = nchunks;                     // answer in [lo, hi]; hi = nchunks means "beyond the end"
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (prefix[mid + 1] >= v) hi = mid; else lo = mid + 1;
    }
    long long found = N - 1;
    if (lo < nchunks) {
        const int64_t base = (int64_t)lo * PP_CHUNK + lane * 64;
        double seg = 0.0;
        for (int q = 0; q < 64; ++q) { const int64_t i = base + q; if (i < N) seg += closest[i]; }
        // exclusive prefix of the 64 segment sums, in lane order
        double incl = seg;
        for (int off = 1; off < 64; off <<= 1) { const double t = __shfl_up(incl, off); if (lane >= off) incl += t; }
        const double start = prefix[lo] + (incl - seg);
        // the crossing segment: first lane whose inclusive running sum reaches v
        const unsigned long long reach = __ballot(prefix[lo] + incl >= v);
        if (reach) {
            const int owner = __ffsll((long long)reach) - 1;
            long long idx = -1;
            if (lane == owner) {
                double run = start;
                idx = base + 63 < N ? base + 63 : N - 1;
                for (int q = 0; q < 64; ++q) {
                    const int64_t i = base + q;
                    if (i >= N) break;
                    run += closest[i];
                    if (run >= v) { idx = i; break; }
                }
            }
            found = __shfl(idx, owner);
        } else {
            // rounding between the chunk's tree sum and its sequential sum: the value lies just past this chunk
            const long long nxt = (long long)(lo + 1) * PP_CHUNK;
            found = nxt < N ? nxt : N - 1;
        }
    }
    if (lane == 0) st->cand[trial] = found;
}

// partial[block][trial] = sum over the block's samples of min(closest_i, d(x_i, x_cand[trial]))
template <int NS>
__global__ void __launch_bounds__(PP_THREADS) pp_candidates_kernel(int64_t N, int n, int L, const double* __restrict__ Xt,
                                                                  const double* __restrict__ xsq, const double* __restrict__ closest,
                                                                  const PPState* __restrict__ st, double* __restrict__ partial) {
    // candidate rows in LDS: [trial][16 coordinates | norm | pad].  A compiler-level memory barrier in front of every
    // trial keeps their reads where they are used: as plain loop invariants the compiler hoisted all 16 x 17 of them
    // into registers (256 VGPRs + scratch, one wave per SIMD, 1.2 ms per pass over 1e7 rows).
    constexpr int CSW = KM_NMAX + 2;
    __shared__ double cs[PP_LMAX * CSW];
    __shared__ double red[PP_THREADS / 64][PP_LMAX];
    for (int e = threadIdx.x; e < L * CSW; e += PP_THREADS) {
        const int t = e / CSW, j = e % CSW;
        const int64_t ci = st->cand[t];
        cs[e] = j < KM_NMAX ? ((NS > 0 ? j < NS : j < n) ? Xt[(int64_t)j * N + ci] : 0.0) : (j == KM_NMAX ? xsq[ci] : 0.0);
    }
    __syncthreads();
    const double2* csv = reinterpret_cast<const double2*>(cs);
    double acc[PP_LMAX];
#pragma unroll
    for (int t = 0; t < PP_LMAX; ++t) acc[t] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * PP_THREADS + threadIdx.x; i < N; i += (int64_t)gridDim.x * PP_THREADS) {
        double x[KM_NMAX];
        pp_load_col<NS>(Xt, N, n, i, x);
        const double xx = xsq[i], old = closest[i];
#pragma unroll
        for (int t = 0; t < PP_LMAX; ++t) {
            if (t < L) {
                asm volatile("" ::: "memory");
                double dot = 0.0, dot1 = 0.0;
                constexpr int NJ = NS > 0 ? (NS + 1) / 2 * 2 : KM_NMAX;     // coordinates beyond n are zero on both sides
#pragma unroll
                for (int j = 0; j < NJ; j += 2) {
                    const double cx = csv[(t * CSW + j) / 2].x, cy = csv[(t * CSW + j) / 2].y;
                    dot = fma(x[j], cx, dot);
                    dot1 = fma(x[j + 1], cy, dot1);
                }
                const double cn = csv[(t * CSW + KM_NMAX) / 2].x;
                double d = (-2.0 * (dot + dot1) + cn) + xx;
                d = d > 0.0 ? d : 0.0;
                acc[t] += old < d ? old : d;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < PP_LMAX; ++t) {
        double a = acc[t];
        for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][t] = a;
    }
    __syncthreads();
    if (threadIdx.x < L) {
        double a = 0.0;
        for (int w = 0; w < PP_THREADS / 64; ++w) a += red[w][threadIdx.x];
        partial[(int64_t)blockIdx.x * PP_LMAX + threadIdx.x] = a;
    }
}

// winner = first trial with the smallest potential (np.argmin); records it as centre c.  The per-block partial
// potentials are summed in a fixed order: 16 threads per trial take every 16th block, then a fixed tree.
__global__ void __launch_bounds__(16 * PP_LMAX) pp_select_kernel(int nblocks, int n, int L, int c, const double* __restrict__ X, int64_t xstride,
                                                                const double* __restrict__ mean, const double* __restrict__ partial,
                                                                PPState* __restrict__ st, double* __restrict__ C, long long* __restrict__ indices) {
    __shared__ double part[PP_LMAX][16];
    __shared__ double pots[PP_LMAX];
    __shared__ long long win;
    const int t = threadIdx.x >> 4, l = threadIdx.x & 15;
    if (t < L) {
        double a = 0.0;
        for (int b = l; b < nblocks; b += 16) a += partial[(int64_t)b * PP_LMAX + t];
        part[t][l] = a;
    }
    __syncthreads();
    if (threadIdx.x < L) {
        const double* q = part[threadIdx.x];
        pots[threadIdx.x] = (((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]))) +
                            (((q[8] + q[9]) + (q[10] + q[11])) + ((q[12] + q[13]) + (q[14] + q[15])));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int best = 0;
        for (int q = 1; q < L; ++q) if (pots[q] < pots[best]) best = q;
        st->pot = pots[best];
        st->last = st->cand[best];
        indices[c] = st->cand[best];
        win = st->cand[best];
    }
    __syncthreads();
    if (threadIdx.x < n) C[(int64_t)c * n + threadIdx.x] = X[win * xstride + threadIdx.x] - (mean ? mean[threadIdx.x] : 0.0);
}

__global__ void pp_first_kernel(int n, long long first, const double* __restrict__ X, int64_t xstride, const double* __restrict__ mean,
                                PPState* __restrict__ st, double* __restrict__ C, long long* __restrict__ indices) {
    const int t = threadIdx.x;
    if (t == 0) { st->last = first; st->pot = 0.0; indices[0] = first; }
    if (t < n) C[t] = X[first * xstride + t] - (mean ? mean[t] : 0.0);
}

int kmeanspp_chunks(int64_t N) { return (int)((N + PP_CHUNK - 1) / PP_CHUNK); }
int kmeanspp_blocks(int64_t N) {
    const int64_t need = (N + PP_THREADS - 1) / PP_THREADS;
    return (int)(need < 2048 ? (need > 0 ? need : 1) : 2048);
}
size_t kmeanspp_state_bytes() { return sizeof(PPState); }

// the whole seeding loop, stream ordered; u: device [(k-1) * L] uniforms; Xt: device scratch [n][N]; C: device [k][n];
// indices: device [k] (int64)
hipError_t launch_kmeanspp(hipStream_t st, int64_t N, int n, int k, int L, const double* X, int64_t xstride, const double* mean,
                           long long first, const double* u, double* Xt, double* xsq, double* closest, double* chunk_sum, double* partial,
                           void* state, double* C, long long* indices) {
    if (n > KM_NMAX || L > PP_LMAX || L < 1) return hipErrorInvalidValue;
    const int nchunks = kmeanspp_chunks(N), nblk = kmeanspp_blocks(N);
    if ((size_t)(nchunks + 1) * 8 > 60 * 1024) return hipErrorInvalidValue;        // prefix table of pp_pick in LDS: N <= 3.1e7
    PPState* ps = reinterpret_cast<PPState*>(state);
    const unsigned nb = (unsigned)((N + PP_THREADS - 1) / PP_THREADS);
#define PP_DISPATCH(NS_) do { \
        hipLaunchKernelGGL(pp_transpose_kernel<NS_>, dim3(nb), dim3(PP_THREADS), 0, st, N, n, X, xstride, mean, Xt, xsq); \
        hipLaunchKernelGGL(pp_first_kernel, dim3(1), dim3(64), 0, st, n, first, X, xstride, mean, ps, C, indices); \
        for (int c = 1; c < k; ++c) { \
            hipLaunchKernelGGL(pp_update_chunksum_kernel<NS_>, dim3(nchunks), dim3(PP_THREADS), 0, st, N, n, Xt, xsq, ps, c == 1 ? 1 : 0, closest, chunk_sum); \
            hipLaunchKernelGGL(pp_pick_kernel, dim3(1), dim3(64 * L), (size_t)(nchunks + 1) * 8, st, N, nchunks, L, u + (size_t)(c - 1) * L, closest, chunk_sum, ps, c == 1 ? 1 : 0); \
            hipLaunchKernelGGL(pp_candidates_kernel<NS_>, dim3(nblk), dim3(PP_THREADS), 0, st, N, n, L, Xt, xsq, closest, ps, partial); \
            hipLaunchKernelGGL(pp_select_kernel, dim3(1), dim3(16 * PP_LMAX), 0, st, nblk, n, L, c, X, xstride, mean, partial, ps, C, indices); \
        } } while (0)
    if (n == 12) PP_DISPATCH(12); else if (n == 13) PP_DISPATCH(13); else PP_DISPATCH(0);
#undef PP_DISPATCH
    return hipGetLastError();
}

int kmeans_blocks(int64_t N);
size_t kmeans_workspace_doubles(int n, int k) { return (size_t)KM_BLOCKS * k * (n + 1) + KM_BLOCKS + (size_t)k * 16 + 8; }

hipError_t launch_kmeans_c2(hipStream_t st, int n, int k, const double* C, double* c2) {
    hipLaunchKernelGGL(kmeans_c2_kernel, dim3((k + 255) / 256), dim3(256), 0, st, n, k, C, c2);
    return hipGetLastError();
}

// one E-step (+ accumulation); c2 = the packed centre table [k][16] that launch_kmeans_c2 / launch_kmeans_update maintain
hipError_t launch_kmeans_assign(hipStream_t st, int64_t N, int n, int k, const double* X, int64_t xstride, const double* mean,
                                const double* C, const double* c2, int* labels, double* partial, double* block_inertia, int* block_changed) {
    (void)C;
    if (n > KM_CMAX || (reinterpret_cast<uintptr_t>(c2) & 127)) return hipErrorInvalidValue;
    const size_t lds = (size_t)k * (n + 1) * sizeof(double) ;
    if (lds > 150 * 1024) return hipErrorInvalidValue;
    const int blocks = kmeans_blocks(N);
#define KM_LAUNCH(NS_) do { \
        hipError_t e_ = hipFuncSetAttribute((const void*)kmeans_assign_kernel<NS_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e_ != hipSuccess) return e_; \
        hipLaunchKernelGGL(kmeans_assign_kernel<NS_>, dim3(blocks), dim3(KM_THREADS), lds, st, N, n, k, X, xstride, mean, c2, labels, partial, \
                           block_inertia, block_changed); } while (0)
    if (n == 12) KM_LAUNCH(12); else if (n == 13) KM_LAUNCH(13); else KM_LAUNCH(0);
#undef KM_LAUNCH
    return hipGetLastError();
}
int kmeans_blocks(int64_t N) {
    const int64_t need = (N + KM_THREADS - 1) / KM_THREADS;
    return need < KM_BLOCKS ? (int)(need > 0 ? need : 1) : KM_BLOCKS;
}
hipError_t launch_kmeans_update(hipStream_t st, int nblocks, int n, int k, const double* partial, const double* block_inertia,
                                const int* block_changed, double* C, double* c2, double* stats) {
    hipError_t e = hipMemsetAsync(stats, 0, 3 * sizeof(double), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kmeans_update_kernel, dim3((k + 3) / 4), dim3(256), 0, st, nblocks, n, k, partial, block_inertia, block_changed, C, c2, stats);
    return hipGetLastError();
}

}  // namespace brov
