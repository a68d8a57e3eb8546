// colstats.hip -- column means and variances of a row-major sample matrix (round 6).
//
// scikit-learn's KMeans centres the samples (`X -= X.mean(axis=0)`) and scales its tolerance by `mean(var(X, axis=0))`
// (sklearn/cluster/_kmeans.py: _tolerance) before the Lloyd loop that KoopmanEDMDc.fit / fit_multi run
// (Koopman/koopmanEDMDc.py:85,126).  Up to round 5 the host layer took both from torch reductions; the drop-in classes no longer
// need torch (north_star: torch only on the PINc path), so the two reductions live here.  HBM-bound: one pass per statistic over
// N x n doubles, rows read whole (n <= 16 contiguous doubles per thread), per-thread register accumulators, a fixed-order block
// reduction, one partial row per block; the host adds the <= 1024 partial rows in index order.  The result depends on (N, n, stride)
// only -- not on the stream, the caller or the array library that owns the buffer.
#include <cstdint>
#include <hip/hip_runtime.h>
#include "brov2_kernels.h"

namespace brov {

constexpr int CS_THREADS = 256;
constexpr int CS_MAXN = 16;

// partial[b][j] = sum over this block's rows of (x[row][j] - shift[j])^(SQ ? 2 : 1); rows are dealt grid-stride
template <bool SQ>
__global__ void __launch_bounds__(CS_THREADS) colstats_kernel(int64_t N, int n, const double* __restrict__ X, int64_t xstride,
                                                              const double* __restrict__ shift, double* __restrict__ partial) {
    __shared__ double red[CS_THREADS / 64][CS_MAXN];
    double acc[CS_MAXN];
    double sh[CS_MAXN];
#pragma unroll
    for (int j = 0; j < CS_MAXN; ++j) { acc[j] = 0.0; sh[j] = (SQ && j < n) ? shift[j] : 0.0; }
    const int64_t step = (int64_t)gridDim.x * CS_THREADS;
    for (int64_t row = (int64_t)blockIdx.x * CS_THREADS + threadIdx.x; row < N; row += step) {
        const double* x = X + row * xstride;
#pragma unroll
        for (int j = 0; j < CS_MAXN; ++j)
            if (j < n) {
                const double v = x[j] - sh[j];
                acc[j] += SQ ? v * v : v;
            }
    }
    // wave reduction in a fixed butterfly order, then the four waves of the block in index order
#pragma unroll
    for (int j = 0; j < CS_MAXN; ++j) {
        double v = acc[j];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
        acc[j] = v;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
        for (int j = 0; j < CS_MAXN; ++j) red[wave][j] = acc[j];
    __syncthreads();
    if (threadIdx.x < CS_MAXN) {
        double v = 0.0;
        for (int w = 0; w < CS_THREADS / 64; ++w) v += red[w][threadIdx.x];
        partial[(int64_t)blockIdx.x * CS_MAXN + threadIdx.x] = v;
    }
}

int colstats_blocks(int64_t N) {
    const int64_t want = (N + CS_THREADS - 1) / CS_THREADS;
    return (int)(want < 1 ? 1 : (want > 1024 ? 1024 : want));       // 1024 blocks = four per CU
}

hipError_t launch_colstats(hipStream_t st, int64_t N, int n, const double* X, int64_t xstride, const double* d_shift, bool squares,
                           double* d_partial) {
    if (N < 1 || n < 1 || n > CS_MAXN || xstride < n) return hipErrorInvalidValue;
    const int nb = colstats_blocks(N);
    if (squares) hipLaunchKernelGGL(colstats_kernel<true>, dim3(nb), dim3(CS_THREADS), 0, st, N, n, X, xstride, d_shift, d_partial);
    else hipLaunchKernelGGL(colstats_kernel<false>, dim3(nb), dim3(CS_THREADS), 0, st, N, n, X, xstride, d_shift, d_partial);
    return hipGetLastError();
}

// ---- block copy by the shader cores --------------------------------------------------------------------------------------------------
// dst[i] = src[i] for 16-byte words; either side may be pinned, device-mapped HOST memory (the ctx's staging blocks): the copy then
// crosses the host link under the kernel's own loads / stores instead of going to an SDMA engine (capi.hip: copy_h2d / copy_d2h).
__global__ void __launch_bounds__(256) copy_words_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t nwords) {
    const int64_t step = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += step) dst[i] = src[i];
}

hipError_t launch_copy_bytes(hipStream_t st, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return hipSuccess;
    if ((bytes & 15) || (reinterpret_cast<uintptr_t>(dst) & 15) || (reinterpret_cast<uintptr_t>(src) & 15)) return hipErrorInvalidValue;
    const int64_t nwords = (int64_t)(bytes / 16);
    int64_t nb = (nwords + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)nb), dim3(256), 0, st, static_cast<const uint4*>(src), static_cast<uint4*>(dst), nwords);
    return hipGetLastError();
}

}  // namespace brov
