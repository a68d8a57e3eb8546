// sortperm.hip -- the sample order of the Lloyd loop (round 3).
//
// The candidate filter of the E-step (kmeans.hip) decides per WAVE: 64 consecutive samples share one union of candidate
// centres, and a group's radius is the largest distance of its lanes to their centre.  In trajectory order a wave holds 4.5
// label groups and evaluates 131 of 512 centres; with the samples ordered by (label, distance to the centre) a wave is one
// group of homogeneous radius and evaluates 55 -- the per-sample need (tools/attic/nbr_probe.py, tools/attic/sort_probe.py).  The order
// decays as labels change (0.3-1.4 % of the samples per iteration: 57 -> 71 -> 80 -> 86 centres one to four iterations after a
// sort early on, 55 -> 70 over six iterations late), so the loop re-sorts when enough labels have moved (capi.hip).
//
// The order is a permutation, not a copy: the E-step (kmeans_assign_lds_kernel) reads row perm[p] for position p -- its rows
// are prefetched a pass ahead and the kernel is bound by the vector ALU, so the scattered 96-byte reads cost nothing measurable
// (first form: a sorted private copy of the rows, 1.9 GB moved per re-sort; 330 -> 322 ms per 300 iterations without it, and a
// re-sort fell from 1.0 to 0.45 ms).  A re-sort = keys (label << 12 | the top 12 bits of the float distance^2: non-negative floats
// order like their bit patterns), rocPRIM's device radix sort of (key, position) pairs -- a plain library primitive, like a
// library GEMM --, and a gather of labels and permutation.  Labels and distances live per POSITION; the labels go back to the
// caller's order at the end.  Nothing here touches the arithmetic of the E / M steps: the labels are those of the unsorted loop
// bit for bit, the member sums differ by the order of their additions.
#include <cstdint>
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include "brov2_kernels.h"

namespace brov {

#ifndef KM_SORT_DBITS
#define KM_SORT_DBITS 12      // bits of the float distance^2 (below the sign: exponent + 4 of the mantissa) in the key, under the 10
                              // label bits: 22-bit keys sort in three radix passes instead of four (300 iterations: 300 -> 289 ms;
                              // 14 bits: 292 ms; 6 bits -- not even the whole exponent --: 412 ms)
#endif
static_assert(KM_SORT_LABEL_BITS + KM_SORT_DBITS <= 32, "sort keys are 32-bit");

__global__ void __launch_bounds__(256) sort_keys_kernel(int64_t N, const int* __restrict__ labels, const float* __restrict__ d2,
                                                        unsigned* __restrict__ keys, unsigned* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const unsigned lab = (unsigned)labels[i] & (unsigned)(KM_SORT_LABEL_MAX - 1);
    const float d = d2[i];
    const unsigned bits = d > 0.0f ? (__float_as_uint(d) & 0x7FFFFFFFu) : 0u;      // NaN / negative: front of its cluster
    keys[i] = (lab << KM_SORT_DBITS) | (bits >> (31 - KM_SORT_DBITS));
    vals[i] = (unsigned)i;
}

// position j of the new order = position src[j] of the old one: labels and the permutation to the caller's rows move, the rows
// themselves stay where they are (perm_old == nullptr: the old order is the caller's)
// (round 4, distance bounds: three per-position float arrays -- the sort key's distance and the two bounds -- move with the labels: an
// E-step that visits only the samples whose bounds failed does not rewrite the others')
__global__ void __launch_bounds__(256) sort_gather_index_kernel(int64_t N, const unsigned* __restrict__ src, const int* __restrict__ labels_old,
                                                                int* __restrict__ labels_new, const int* __restrict__ perm_old, int* __restrict__ perm_new,
                                                                const float* __restrict__ f0_old, float* __restrict__ f0_new,
                                                                const float* __restrict__ f1_old, float* __restrict__ f1_new,
                                                                const float* __restrict__ f2_old, float* __restrict__ f2_new) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const int64_t s = src[j];
    labels_new[j] = labels_old[s];
    perm_new[j] = perm_old ? perm_old[s] : (int)s;
    if (f0_new) f0_new[j] = f0_old[s];
    if (f1_new) f1_new[j] = f1_old[s];
    if (f2_new) f2_new[j] = f2_old[s];
}

__global__ void __launch_bounds__(256) sort_unpermute_kernel(int64_t N, const int* __restrict__ perm, const int* __restrict__ labels_sorted,
                                                             int* __restrict__ labels_out) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < N) labels_out[perm[j]] = labels_sorted[j];
}

size_t kmeans_sort_temp_bytes(int64_t N) {
    size_t bytes = 0;
    unsigned* p = nullptr;
    if (rocprim::radix_sort_pairs(nullptr, bytes, p, p, p, p, (size_t)N, 0, KM_SORT_LABEL_BITS + KM_SORT_DBITS, (hipStream_t)nullptr) != hipSuccess) return 0;
    return bytes;
}

hipError_t launch_kmeans_resort(hipStream_t st, int64_t N, const int* labels_old, int* labels_new, const int* perm_old, int* perm_new,
                                const float* d2, unsigned* keys_in, unsigned* keys_out, unsigned* vals_in, unsigned* vals_out, void* temp,
                                size_t temp_bytes, float* d2_new, const float* ub_old, float* ub_new, const float* lb_old, float* lb_new) {
    if (N <= 0) return hipSuccess;
    if (N >= ((int64_t)1 << 31)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sort_keys_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, N, labels_old, d2, keys_in, vals_in);
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)N, 0, KM_SORT_LABEL_BITS + KM_SORT_DBITS, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sort_gather_index_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, N, vals_out, labels_old, labels_new, perm_old, perm_new,
                       d2, d2_new, ub_old, ub_new, lb_old, lb_new);
    return hipGetLastError();
}

hipError_t launch_kmeans_unpermute(hipStream_t st, int64_t N, const int* perm, const int* labels_sorted, int* labels_out) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(sort_unpermute_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, N, perm, labels_sorted, labels_out);
    return hipGetLastError();
}

}  // namespace brov
