// kmeans.hip -- Lloyd iterations on device for the RBF centres of KoopmanEDMDc.fit
// (Koopman/koopmanEDMDc.py:85,126 call sklearn.cluster.KMeans(n_clusters, n_init="auto", random_state=0)).
//
// scikit-learn stays the owner of the initialisation (k-means++, seeded) -- the host layer calls it --
// and this file restates what scikit-learn 1.7.2's `_kmeans_single_lloyd` iterates:
//   E-step: label_i = argmin_c (|c|^2 - 2 x_i.c)          (first minimum; |x_i|^2 is common to all c)
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

namespace brov {

constexpr int KM_NMAX = 16;
constexpr int KM_BLOCKS = 512;        // persistent blocks (2 per CU: each holds a 53 KB LDS table of member sums at k = 512)
constexpr int KM_THREADS = 1024;      // 16 waves per block -> 8 waves per SIMD: the centre loop waits on its scalar loads once
                                      // per centre, and only other waves can fill that time (256-thread blocks: 4x slower)

typedef const double __attribute__((address_space(4)))* cdp;

template <int NS>
__global__ void __launch_bounds__(KM_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) kmeans_assign_kernel(int64_t N, int n, int k, const double* __restrict__ X, int64_t xstride,
                                                            const double* __restrict__ mean, const double* __restrict__ C,
                                                            const double* __restrict__ c2, int* __restrict__ labels,
                                                            double* __restrict__ partial /* [blocks][k][n+1] */,
                                                            double* __restrict__ block_inertia, int* __restrict__ block_changed) {
    extern __shared__ double sums[];                  // [k][n+1]: member sums and count
    const int np1 = n + 1;
    for (int i = threadIdx.x; i < k * np1; i += KM_THREADS) sums[i] = 0.0;
    __shared__ double sh_inertia[KM_THREADS / 64];
    __shared__ int sh_changed[KM_THREADS / 64];
    __syncthreads();
    const cdp Cc = (cdp)(unsigned long long)C;
    const cdp c2c = (cdp)(unsigned long long)c2;
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
        double best = 1.0e300;
        int bi = 0;
        for (int c = 0; c < k; ++c) {
            const cdp cc = Cc + (int64_t)c * n;
            double dot = 0.0, dot1 = 0.0;                 // two chains: dependent fp64 FMAs do not issue back to back
            if constexpr (NS > 0) {
#pragma unroll
                for (int j = 0; j + 1 < NS; j += 2) { dot = fma(x[j], cc[j], dot); dot1 = fma(x[j + 1], cc[j + 1], dot1); }
                if constexpr (NS & 1) dot = fma(x[NS - 1], cc[NS - 1], dot);
            } else {
#pragma unroll
                for (int j = 0; j < KM_NMAX; ++j) if (j < n) dot = fma(x[j], cc[j], dot);
            }
            dot += dot1;
            const double d = fma(-2.0, dot, c2c[c]);
            if (d < best) { best = d; bi = c; }       // strict '<': first minimum wins, like argmin
        }
        if (live) {
            if (labels[i] != bi) ++changed;
            labels[i] = bi;
            inertia += best + x2;
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
                                                            double* __restrict__ C, double* __restrict__ c2, double* __restrict__ stats) {
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
        if (j == 0) { c2[c] = q; atomicAdd(&stats[0], shift2); }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double in = 0.0;
        long long ch = 0;
        for (int b = 0; b < nblocks; ++b) { in += block_inertia[b]; ch += block_changed[b]; }
        stats[1] = in;
        stats[2] = (double)ch;
    }
}

__global__ void __launch_bounds__(256) kmeans_c2_kernel(int n, int k, const double* __restrict__ C, double* __restrict__ c2) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= k) return;
    double s = 0.0;
    for (int j = 0; j < n; ++j) s = fma(C[c * n + j], C[c * n + j], s);
    c2[c] = s;
}

int kmeans_blocks(int64_t N);
size_t kmeans_workspace_doubles(int n, int k) { return (size_t)KM_BLOCKS * k * (n + 1) + KM_BLOCKS + k + 8; }

hipError_t launch_kmeans_c2(hipStream_t st, int n, int k, const double* C, double* c2) {
    hipLaunchKernelGGL(kmeans_c2_kernel, dim3((k + 255) / 256), dim3(256), 0, st, n, k, C, c2);
    return hipGetLastError();
}

// one E-step (+ accumulation); pass C/c2 as they stand
hipError_t launch_kmeans_assign(hipStream_t st, int64_t N, int n, int k, const double* X, int64_t xstride, const double* mean,
                                const double* C, const double* c2, int* labels, double* partial, double* block_inertia, int* block_changed) {
    if (n > KM_NMAX) return hipErrorInvalidValue;
    const size_t lds = (size_t)k * (n + 1) * sizeof(double) ;
    if (lds > 150 * 1024) return hipErrorInvalidValue;
    const int blocks = kmeans_blocks(N);
#define KM_LAUNCH(NS_) do { \
        hipError_t e_ = hipFuncSetAttribute((const void*)kmeans_assign_kernel<NS_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e_ != hipSuccess) return e_; \
        hipLaunchKernelGGL(kmeans_assign_kernel<NS_>, dim3(blocks), dim3(KM_THREADS), lds, st, N, n, k, X, xstride, mean, C, c2, labels, partial, \
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
