// kmeans.hip -- Lloyd iterations on device for the RBF centres of KoopmanEDMDc.fit
// (Koopman/koopmanEDMDc.py:85,126 call sklearn.cluster.KMeans(n_clusters, n_init="auto", random_state=0)).
//
// scikit-learn stays the owner of the initialisation (k-means++, seeded) -- the host layer calls it --
// and this file restates what scikit-learn 1.7.2's `_kmeans_single_lloyd` iterates:
//   E-step: label_i = argmin_c (|c|^2 - 2 x_i.c)          (first minimum; |x_i|^2 is common to all c;
//           evaluated as argmax_c (x_i.c - |c|^2 / 2))
//   M-step: c <- mean of its members; an empty cluster is relocated to the sample farthest from its centre
//           (`_relocate_empty_clusters_dense`), one that stays empty takes the centre of the biggest cluster (`_average_centers`)
//   stop  : labels unchanged ("strict convergence") or sum |c_new - c_old|^2 <= tol, or max_iter
// on the mean-centred data (the host passes the column means), so that centres agree with
// scikit-learn's to rounding whenever no assignment is decided by the last bit.
//
// E-step kernel: lane = sample (its n coordinates in VGPRs), loop over centres whose coordinates arrive
// as wave-uniform scalar loads: 12 FMA + compare/select per (64 samples, centre).  Persistent 1024-thread blocks
// accumulate member sums and counts in an LDS table and write one partial per block and epoch; a second
// kernel reduces the partials and a third forms the new centres.
//
// Round 4: the member sums are INTEGERS.  Every coordinate of every sample is turned into a 64-bit fixed-point number,
// q = round(x_j * s_j), s_j = 2^(48 - e_j) with 2^e_j > max_i |x_ij| (one pass over the data before the loop:
// kmeans_range_kernel), and all additions -- DPP partial sums inside a wave, LDS atomics, block partials, the 128-bit totals,
// an all-reduce over ranks -- are integer additions: associative, so the centres are the same bits whatever the arrival
// order of the atomics, the sample order (sorted or the caller's), the kernel (LDS / DPP or scalar records), the number of
// blocks or the number of GPUs the samples are sharded over.  (Rounds 2-3: fp64 LDS atomics in arrival order; two runs
// differed in the last bits of their centres.)  The quantum is 2^-48 of a coordinate's range (3.6e-15..7.1e-15 relative to
// max |x_j|; a sample errs by at most half of it, a mean of m samples by ~1/sqrt(12 m) of it) -- below what the order of
// scikit-learn's own fp64 additions leaves open.  x -> q is one FMA with the constant 1.5 * 2^52 (the integer sits in
// the mantissa; the constant's bit pattern is taken off once per flushed table, times the count).  A table is flushed every
// KM_EPOCH_PASSES passes (2^14 samples x 2^48 < 2^63).  A non-finite sample adds zeros and a poison flag above the count:
// its centre becomes NaN, as a floating-point sum would have made it.
//
// Round 3: the E-step with a candidate filter (kmeans_assign_kernel<NS, true>).  With scikit-learn's stopping rule the config-3
// centres need all 300 iterations (tol 1e-4 is not reached at N = 1e7, k = 512), so Lloyd is two thirds of a fit().  Exact
// pruning by the triangle inequality, decided per WAVE so that every lane keeps running the same instruction stream:
//   * a wave holds 64 consecutive samples; their labels of the previous iteration form a few groups (4.5 on trajectory data);
//   * for a group with label a:  u = max over its lanes of d(x, c_a) (one distance per lane, rounded up);  a centre c with
//     d(c_a, c) >= 2 u + margin is farther from every lane of the group than c_a is, by at least `margin`;
//   * the wave evaluates the union over its groups of { c : d(c_a, c) < 2 u_a + margin } -- 131 of 512 centres on average --
//     in increasing index order with the plain kernel's eval(), for all lanes (a superfluous candidate is harmless).
// The labels are those of the full scan bit for bit: a skipped centre's exact score is below the group centre's by more than
// margin^2 / 2 = 5e-13 R^2 while the FMA chain errs by < 1.1e-15 R^2 (R^2 = 2 max |x|^2), so it cannot be the computed argmax,
// and ties among the candidates resolve to the lowest index as before.  The centre-centre distances come from
// kmeans_cdist_kernel (k x k floats rounded down, rebuilt after every M-step; 1 MB at k = 512, L2 resident).  Waves with a stale label
// (first iteration), a non-finite sample or more than KM_GMAX label groups (shuffled data) take the full scan.
#include "brov2_kernels.h"
#include <cstdint>
#include <type_traits>
#include <utility>

namespace brov {

#ifndef KM_LIST_DYNAMIC
#define KM_LIST_DYNAMIC 1        // list-form E-step: the waves draw their passes from a device-wide counter (0: a fixed share per block, the round-4 form)
#endif
#ifndef KM_FP64_BOUNDS
#define KM_FP64_BOUNDS 1         // list-form E-step: the fp64 evaluation paths leave distance bounds too (0: NaN, the round-4 behaviour)
#endif
#ifndef KM_TWO_CHAINS
#define KM_TWO_CHAINS 0
#endif
constexpr int KM_NMAX = 16;
constexpr int KM_BLOCKS = 512;        // persistent blocks (2 per CU: each holds a 53 KB LDS table of member sums at k = 512)
constexpr int KM_THREADS = 1024;      // 16 waves per block -> 8 waves per SIMD: the centre loop waits on its scalar loads once
                                      // per centre, and only other waves can fill that time (256-thread blocks: 4x slower)

typedef const double __attribute__((address_space(4)))* cdp;
// packed centre table: one 128-byte record per centre = n coordinates, |c|^2 / 2 in slot n, zeros (n <= 15)
constexpr int KM_CMAX = 15;
struct __attribute__((aligned(128))) Cen { double v[16]; };
typedef const Cen __attribute__((address_space(4)))* ccp;

constexpr int KM_GMAX = 8;            // label groups per wave the candidate filter handles; more -> full scan
// wave-wide maximum of a 32-bit unsigned value (all 64 lanes active): four DPP steps inside the rows of 16, then the row maxima
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false));     // quad_perm [1,0,3,2]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false));     // quad_perm [2,3,0,1]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false));    // row_half_mirror
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false));    // row_mirror
    const unsigned r0 = __builtin_amdgcn_readlane((int)v, 0), r1 = __builtin_amdgcn_readlane((int)v, 16);
    const unsigned r2 = __builtin_amdgcn_readlane((int)v, 32), r3 = __builtin_amdgcn_readlane((int)v, 48);
    return max(max(r0, r1), max(r2, r3));
}

// ---- member sums in fixed point (header comment): x -> round(x s) as the mantissa of x s + 1.5 * 2^52 ---------------------------
constexpr int KM_FIX_BITS = 48;                        // |x_j s_j| < 2^48
constexpr int KM_EPOCH_PASSES = 16;                    // passes of KM_THREADS samples between two flushes of a block's table
constexpr double KM_MAGIC = 6755399441055744.0;        // 1.5 * 2^52: ulp 1 on [2^52, 2^53)
constexpr unsigned long long KM_MAGIC_BITS = 0x4338000000000000ull;
constexpr unsigned long long KM_POISON = 1ull << 40;   // added to the count by a non-finite sample (counts stay below 2^40)
static_assert((long long)KM_THREADS * KM_EPOCH_PASSES <= (1ll << (62 - KM_FIX_BITS)), "a block's table must not overflow 63 bits between flushes");
typedef unsigned long long u64;
typedef const double __attribute__((address_space(4)))* cdp_;
__device__ __forceinline__ u64 km_fix(double x, double s) { return (u64)__double_as_longlong(fma(x, s, KM_MAGIC)); }
// sum over the 8 lanes of half a DPP row, in every lane of it: integer additions (any order gives the same bits)
__device__ __forceinline__ u64 half_row_sum_u64(u64 v) {
    auto dpp = [](u64 x, auto ctrl) {
        const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)x, decltype(ctrl)::value, 0xF, 0xF, true);
        const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(x >> 32), decltype(ctrl)::value, 0xF, 0xF, true);
        return ((u64)hi << 32) | lo;
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});       // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});       // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});      // row_half_mirror
    return v;
}
// The sum over ALL 64 lanes for all NC coordinates of a sample at once, carries through VCC, the DPP operand fused into the
// additions: six steps (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: every lane holds its row's sum; row_bcast:15 into rows
// 1 and 3, row_bcast:31 into rows 2 and 3: lane 63 holds the wave's), 2 NC instructions each.  lo / hi: the halves of the NC
// fixed-point numbers.  (First form of round 4: three steps, sums over groups of 8 lanes, eight lanes sending the atomics of a
// coordinate to ONE LDS address -- SQ_LDS_ADDR_CONFLICT counted more LDS cycles in conflicts than in accesses, and the phase cost
// 0.13 ms of a 0.87 ms E-step; rows of 16 and four lanes: 288 -> 277 ms per 300 iterations; the whole wave and one lane: 272 ms.
// The compiler's own form of a step is two v_mov_b32_dpp + add + addc.)  A DPP read needs two wait states after a VALU write of
// its source: the leading s_nop covers the values computed just before, inside the block every register is re-read 2 NC - 1
// instructions after it was written.
#define KM_DPPADDM(ctrl, rm, L, H) "v_add_co_u32_dpp %" #L ", vcc, %" #L ", %" #L " " ctrl " row_mask:" rm " bank_mask:0xf\n\tv_addc_co_u32_dpp %" #H ", vcc, %" #H ", %" #H ", vcc " ctrl " row_mask:" rm " bank_mask:0xf\n\t"
#define KM_DPPSTEP12(ctrl, rm) KM_DPPADDM(ctrl, rm, 0, 1) KM_DPPADDM(ctrl, rm, 2, 3) KM_DPPADDM(ctrl, rm, 4, 5) KM_DPPADDM(ctrl, rm, 6, 7) KM_DPPADDM(ctrl, rm, 8, 9) \
    KM_DPPADDM(ctrl, rm, 10, 11) KM_DPPADDM(ctrl, rm, 12, 13) KM_DPPADDM(ctrl, rm, 14, 15) KM_DPPADDM(ctrl, rm, 16, 17) KM_DPPADDM(ctrl, rm, 18, 19) \
    KM_DPPADDM(ctrl, rm, 20, 21) KM_DPPADDM(ctrl, rm, 22, 23)
#define KM_DPPSTEP13(ctrl, rm) KM_DPPSTEP12(ctrl, rm) KM_DPPADDM(ctrl, rm, 24, 25)
#define KM_DPPALL(STEP) STEP("quad_perm:[1,0,3,2]", "0xf") STEP("quad_perm:[2,3,0,1]", "0xf") STEP("row_half_mirror", "0xf") STEP("row_mirror", "0xf") \
    STEP("row_bcast:15", "0xa") STEP("row_bcast:31", "0xc")
#define KM_Q2(a) "+v"(lo[a]), "+v"(hi[a])
__device__ __forceinline__ void wave_sum_u64x(unsigned (&lo)[12], unsigned (&hi)[12]) {
    asm volatile("s_nop 1\n\t" KM_DPPALL(KM_DPPSTEP12)
                 : KM_Q2(0), KM_Q2(1), KM_Q2(2), KM_Q2(3), KM_Q2(4), KM_Q2(5), KM_Q2(6), KM_Q2(7), KM_Q2(8), KM_Q2(9), KM_Q2(10), KM_Q2(11) : : "vcc");
}
__device__ __forceinline__ void wave_sum_u64x(unsigned (&lo)[13], unsigned (&hi)[13]) {
    asm volatile("s_nop 1\n\t" KM_DPPALL(KM_DPPSTEP13)
                 : KM_Q2(0), KM_Q2(1), KM_Q2(2), KM_Q2(3), KM_Q2(4), KM_Q2(5), KM_Q2(6), KM_Q2(7), KM_Q2(8), KM_Q2(9), KM_Q2(10), KM_Q2(11), KM_Q2(12) : : "vcc");
}
#undef KM_Q2
#undef KM_DPPALL
#undef KM_DPPSTEP13
#undef KM_DPPSTEP12
#undef KM_DPPADDM
// A block's table [k][n+1] goes out as partial[epoch][block]: the bit pattern of 1.5 * 2^52 is taken off the coordinate sums
// (count times, poisoned samples included: they added the pattern of a zero), so that a partial is a signed sum of integers.
__device__ __forceinline__ void km_flush(u64* sums, u64* __restrict__ partial, int ep, int k, int n, bool rezero) {
    const int np1 = n + 1;
    __syncthreads();
    u64* out = partial + ((int64_t)ep * gridDim.x + blockIdx.x) * k * np1;
    for (int c = threadIdx.x >> 4; c < k; c += (int)blockDim.x / 16) {
        const int j = threadIdx.x & 15;
        if (j <= n) {
            const u64 cnt = sums[c * np1 + n];
            const u64 v = sums[c * np1 + j];
            out[c * np1 + j] = j < n ? v - (cnt & (KM_POISON - 1)) * KM_MAGIC_BITS : v;
        }
    }
    if (rezero) {
        __syncthreads();
        for (int i = threadIdx.x; i < k * np1; i += (int)blockDim.x) sums[i] = 0ull;
        __syncthreads();
    }
}
__device__ __forceinline__ void km_zero_epochs(u64* __restrict__ partial, int ep0, int nep, int k, int n) {
    for (int ep = ep0; ep < nep; ++ep) {
        u64* out = partial + ((int64_t)ep * gridDim.x + blockIdx.x) * k * (n + 1);
        for (int i = threadIdx.x; i < k * (n + 1); i += (int)blockDim.x) out[i] = 0ull;
    }
}

#ifndef KM_BLOCKTIME
#define KM_BLOCKTIME 0           // experiment build: how evenly the list-form E-step's work falls on blocks and waves (tools/attic/lloyd_balance.py)
#endif
#if KM_BLOCKTIME
__device__ unsigned long long km_blk[8];       // list form: [0] sum of block durations, [1] blocks, [2] sum of the waves' loop times, [3] waves (100 MHz ticks)
#endif
#ifndef KM_PROFILE
#define KM_PROFILE 0
#endif
#if KM_PROFILE
__device__ unsigned long long km_prof[16];     // [0..4] phase ticks, [7] wave passes, [8] single-reference passes, [9] their candidates, [10] tie repeats, [11] mask-form passes, [12] their candidates, [13] full scans, [14] passes settled by the packed-fp32 screening, [15] their candidates
#define KM_STAMP(slot) do { const unsigned long long now_ = __builtin_readcyclecounter(); t_acc[slot] += now_ - t_prev; t_prev = now_; } while (0)
#else
#define KM_STAMP(slot) do { } while (0)
#endif

// prm (device): [0] = margin, [1] = eps2 (see the header comment; kmeans_scale_kernel), [2] != 0: a centre is not finite, [3] != 0: hold --
// the M-step found an empty cluster and the host has to relocate it before this E-step may run (it returns at once; kmeans_average_kernel);
// Dc: [k][k] centre distances; fix [32]: the fixed-point scales s_j of the member sums (and their reciprocals)
template <int NS, bool PRUNE>
__global__ void __launch_bounds__(KM_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) kmeans_assign_kernel(int64_t N, int n, int k, const double* __restrict__ X, int64_t xstride,
                                                            const double* __restrict__ mean,
                                                            const double* __restrict__ Ct /* [k][16]: coordinates, |c|^2/2 at [n] */, int* __restrict__ labels,
                                                            u64* __restrict__ partial /* [epochs][blocks][k][n+1] */, int nepochs,
                                                            double* __restrict__ block_inertia, int* __restrict__ block_changed,
                                                            const float* __restrict__ Dc,
                                                            const double* __restrict__ prm, float* __restrict__ d2out, const double* __restrict__ fix) {
    extern __shared__ u64 sums[];                     // [k][n+1]: member sums (fixed point) and count
    if (prm[3] != 0.0) return;                        // hold (block-uniform)
    const int np1 = n + 1;
    for (int i = threadIdx.x; i < k * np1; i += KM_THREADS) sums[i] = 0ull;
    __shared__ double sh_inertia[KM_THREADS / 64];
    __shared__ int sh_changed[KM_THREADS / 64];
    __syncthreads();
    const ccp T = (ccp)(unsigned long long)Ct;
    const cdp_ FS = (cdp_)(unsigned long long)fix;
    double inertia = 0.0;
    int changed = 0;
    double margin = 0.0, eps2 = 0.0;
    bool centres_finite = true;
    if constexpr (PRUNE) { margin = prm[0]; eps2 = prm[1]; centres_finite = prm[2] == 0.0; }
    const int lane = threadIdx.x & 63;
    int pass = 0, ep = 0;
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
            // one FMA chain seeded with -|c|^2/2: n instructions per centre.  (Eight waves per SIMD: the other waves fill the
            // slots a dependent chain leaves; the first version ran two chains and paid an add and an FMA to join them.)
            constexpr int NJ = NS > 0 ? NS : KM_CMAX;     // generic n: slots beyond n hold zeros (and the half norm, read below)
            const double hn = NS > 0 ? t.v[NS] : t.v[n];
#if KM_TWO_CHAINS
            double dot = fma(x[0], t.v[0], -hn), dot1 = x[1] * t.v[1];
#pragma unroll
            for (int j = 2; j + 1 < NJ; j += 2) { dot = fma(x[j], t.v[j], dot); dot1 = fma(x[j + 1], t.v[j + 1], dot1); }
            if constexpr (NJ & 1) dot = fma(x[NJ - 1], t.v[NJ - 1], dot);
            const double sc = dot + dot1;
#else
            double sc = fma(x[0], t.v[0], -hn);
#pragma unroll
            for (int j = 1; j < NJ; ++j) sc = fma(x[j], t.v[j], sc);
#endif
            // strict '>' to replace: the first extremum wins, like np.argmin.  (The wave-uniform index cannot be the scalar operand
            // of the select: v_cndmask reads its condition over the constant bus as well, and gfx9 allows one scalar source.)
            bi = (sc <= best) ? bi : c;
            asm("v_max_f64 %0, %1, %2" : "=v"(best) : "v"(best), "v"(sc));      // plain max: fmax() adds a canonicalising self-max
        };
        const int ol = labels[ii];                    // label of the previous iteration (-1 before the first)
        bool filtered = false;
        if constexpr (PRUNE) {
            // ---- candidate filter (see the header comment); every branch below is wave-uniform
            const bool usable = (unsigned)ol < (unsigned)k && (x2 - x2 == 0.0) && centres_finite;
            if (__ballot(!usable) == 0ull) {
                int ga[KM_GMAX];
                double gthr2[KM_GMAX];                     // squared thresholds: Dc holds SQUARED centre distances (no square roots)
                int ng = 0;
                unsigned long long remaining = ~0ull;
#pragma unroll
                for (int g = 0; g < KM_GMAX; ++g) {
                    if (remaining != 0ull) {
                        const int lead = __builtin_ctzll(remaining);
                        const int a = __builtin_amdgcn_readlane(ol, lead);
                        const bool mine = ol == a;
                        Cen ca;
#pragma unroll
                        for (int j = 0; j < 16; ++j) ca.v[j] = T[a].v[j];
                        constexpr int NJ = NS > 0 ? NS : KM_CMAX;
                        double sc = fma(x[0], ca.v[0], -(NS > 0 ? ca.v[NS] : ca.v[n]));
#pragma unroll
                        for (int j = 1; j < NJ; ++j) sc = fma(x[j], ca.v[j], sc);
                        const double d2 = fma(-2.0, sc, x2);                       // |x - c_a|^2 up to rounding
                        // the group's squared radius as a float rounded UP (non-negative floats order like their bit patterns)
                        float rf = mine ? (float)fmax(d2, 0.0) : 0.0f;
                        rf = rf * 1.0000005f + 1.0e-37f;
                        const unsigned rb = wave_max_u32(mine ? __float_as_uint(rf) : 0u);
                        const double u2 = (double)__uint_as_float(rb) + eps2;      // >= the true squared radius u^2 of the group
                        ga[g] = a;
                        // (2 u + m)^2 <= 4.004 u^2 + 1001 m^2 (Young's inequality): the skip rule d(c_a, c)^2 >= this implies
                        // d(c_a, c) >= 2 u + m
                        // (wave-uniform: pinned into scalar registers -- left in VGPRs the eight thresholds were spilled to scratch and
                        // re-read for every mask word)
                        const double t2 = fma(4.004, u2, 1001.0 * margin * margin);
                        gthr2[g] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(t2)), __builtin_amdgcn_readfirstlane(__double2loint(t2)));
                        ng = g + 1;
                        remaining &= ~__ballot(mine);
                    }
                }
                const int kw = (k + 63) >> 6;
                const int kp = (k + 255) & ~255;                  // row length of Dc: blocks of 256 centres
                // Candidate masks of one block of 256 centres: Dc holds the squared centre distances as floats rounded DOWN, permuted
                // so that a lane's 16-byte load brings the centres 256 blk + 64 q + lane, q = 0..3 -- the four ballots are the
                // four mask words of the block in natural bit order.  (Round 3, first form: doubles, one 8-byte load per group and
                // 64-centre word -- 36 dependent load rounds per wave at k = 512 against 9 now.)
                float tf[KM_GMAX];
#pragma unroll
                for (int g = 0; g < KM_GMAX; ++g) {
                    // threshold as a float rounded UP, pinned in a scalar register
                    const float t = g < ng ? fminf((float)(gthr2[g] * 1.0000001) + 1.0e-37f, 3.4028234e38f) : 0.0f;
                    tf[g] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(t)));
                }
                auto block_masks = [&](int blk, unsigned long long* m4) {
                    bool p0 = false, p1 = false, p2 = false, p3 = false;
#pragma unroll
                    for (int g = 0; g < KM_GMAX; ++g)
                        if (g < ng) {
                            const float4 v = *reinterpret_cast<const float4*>(Dc + (int64_t)ga[g] * kp + blk * 256 + lane * 4);
                            p0 = p0 || v.x < tf[g]; p1 = p1 || v.y < tf[g]; p2 = p2 || v.z < tf[g]; p3 = p3 || v.w < tf[g];
                        }
                    m4[0] = __ballot(p0); m4[1] = __ballot(p1); m4[2] = __ballot(p2); m4[3] = __ballot(p3);
                };
                // candidates of one word in increasing index order, two per trip (their scalar loads go out together)
                auto scan_word = [&](int w, unsigned long long mw) {
                    while (mw != 0ull) {
                        const int c0 = (w << 6) + __builtin_ctzll(mw);
                        mw &= mw - 1ull;
                        if (mw != 0ull) {
                            const int c1 = (w << 6) + __builtin_ctzll(mw);
                            mw &= mw - 1ull;
                            Cen a, b;
#pragma unroll
                            for (int j = 0; j < 16; ++j) { a.v[j] = T[c0].v[j]; b.v[j] = T[c1].v[j]; }
                            eval(a, c0);
                            eval(b, c1);
                        } else {
                            Cen a;
#pragma unroll
                            for (int j = 0; j < 16; ++j) a.v[j] = T[c0].v[j];
                            eval(a, c0);
                        }
                    }
                };
                if (remaining == 0ull) {
                    filtered = true;
                    if (kw <= 8) {
                        // k <= 512: all mask words first (the group table is dead before the first centre is evaluated: with both
                        // alive the scalar registers did not fit and the hot loop carried 7 v_readlane / v_writelane per centre)
                        unsigned long long mws[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
                        block_masks(0, mws);
                        if (kw > 4) {
                            asm volatile("" ::: "memory");        // one block's distance loads at a time (64 VGPRs)
                            block_masks(1, mws + 4);
                        }
                        // one copy of the evaluation loop (eight unrolled copies were 30 KB of code and slower than the interleaved
                        // form); the word is picked with scalar selects
#pragma unroll 1
                        for (int w = 0; w < kw; ++w) {
                            unsigned long long mw = 0ull;
#pragma unroll
                            for (int q = 0; q < 8; ++q) mw = (w == q) ? mws[q] : mw;
                            scan_word(w, mw);
                        }
                    } else {
#pragma unroll 1
                        for (int blk = 0; blk * 4 < kw; ++blk) {
                            unsigned long long m4[4];
                            block_masks(blk, m4);
#pragma unroll
                            for (int q = 0; q < 4; ++q) scan_word(blk * 4 + q, m4[q]);
                        }
                    }
                }
            }
        }
        if (!filtered) {
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
        }
        if (live) {
            if (ol != bi) ++changed;
            labels[i] = bi;
            const double dmin2 = fma(-2.0, best, x2);     // squared distance to the chosen centre (up to rounding)
            if (d2out) d2out[i] = (float)dmin2;            // sort key of the loop's sample order (sortperm.hip)
            inertia += dmin2;
        }
        // member sums, fixed point (header comment).  A wave whose 64 samples all went to ONE centre (the rule once the loop keeps
        // its samples sorted) would send 64 same-address atomics per coordinate through the LDS, one after the other: 13 x 64 LDS
        // cycles per wave, more than the filtered evaluation itself.  Such a wave adds its rows up inside groups of 8 lanes first
        // (DPP, three steps) and sends eight atomics per coordinate (the vector ALU is the scarcer unit: a fourth step costs more
        // than it saves).  A non-finite sample adds zeros and the poison flag.
        const bool bad = !(x2 - x2 == 0.0);
        if (__ballot(bad) != 0ull) {
#pragma unroll
            for (int j = 0; j < KM_NMAX; ++j) x[j] = bad ? 0.0 : x[j];
        }
        const int bi0 = __builtin_amdgcn_readfirstlane(bi);
        if (__ballot(!live || bi != bi0 || bad) == 0ull) {
            u64* s = sums + bi0 * np1;
            const bool leader = (lane & 7) == 0;
            u64 q[KM_NMAX];
#pragma unroll
            for (int j = 0; j < KM_NMAX; ++j)
                if (NS > 0 ? (j < NS) : (j < n)) q[j] = half_row_sum_u64(km_fix(x[j], FS[j]));
            if (leader) {
#pragma unroll
                for (int j = 0; j < KM_NMAX; ++j)
                    if (NS > 0 ? (j < NS) : (j < n)) atomicAdd(&s[j], q[j]);
                atomicAdd(&s[n], 8ull);
            }
        } else if (live) {
            u64* s = sums + bi * np1;
#pragma unroll
            for (int j = 0; j < KM_NMAX; ++j)
                if (NS > 0 ? (j < NS) : (j < n)) atomicAdd(&s[j], km_fix(x[j], FS[j]));
            atomicAdd(&s[n], bad ? 1ull + KM_POISON : 1ull);
        }
        if (++pass == KM_EPOCH_PASSES && base + (int64_t)gridDim.x * KM_THREADS < N) { km_flush(sums, partial, ep, k, n, true); ++ep; pass = 0; }
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
    km_flush(sums, partial, ep, k, n, false);
    km_zero_epochs(partial, ep + 1, nepochs, k, n);
}

// ---- E-step, second form: centre records from the LDS through DPP ---------------------------------------------------------------
// Phase timing of the kernel above on the config-3 data (tools/attic/lloyd_phase_profile.py): the evaluation of the candidates is a
// third to a half of a wave's pass; the rest is waiting -- for the wave's own rows (HBM, nothing else to do meanwhile), for the
// group centres' records (scalar loads, one after the other), for the rows of the distance table, and in the evaluation loop
// for the scalar loads of the records (two in flight per wave: 52 scalar registers).  Eight waves per SIMD were there to hide
// that and left 64 vector registers per lane, not enough to prefetch anything.  This form needs fewer waves:
//   * the packed centre table lives in the LDS (64 KB at k = 512, next to the 53 KB of member sums: one 1024-thread block per
//     CU, 4 waves per SIMD, 128 vector registers);
//   * a centre record reaches a wave as ONE ds_read_b64 -- lane l gets double l & 15 of the record, so every DPP row of 16 lanes
//     holds the whole record -- and the FMA takes its centre operand through `row_newbcast:j` (v_fmac_f64_dpp: the broadcast
//     costs nothing; tools/attic/dpp64_probe.hip checks rate and bits against the scalar-operand form).  A record in flight costs 2
//     vector registers instead of 26 scalar ones: KM2_DEPTH records are prefetched round the evaluation loop;
//   * the candidates of all mask words form one list in the LDS (written by the lanes that hold the mask bits: mbcnt ranks), padded
//     by repeating the last candidate (an equal score never replaces the best): the loop spends no scalar instructions on bit
//     scanning -- a wave issues one instruction every four cycles, and with four waves per SIMD that rate is the budget;
//   * the rows of the NEXT pass are loaded before the evaluation loop of this one (16-byte loads when the rows allow) -- through
//     the loop's sample permutation when there is one (sortperm.hip: row perm[p] for position p, the index loaded a pass earlier
//     still; labels and distances are per position) --, the distance-table rows of up to four label groups are requested together;
//   * the records are evaluated in pairs, two interleaved FMA chains (a dependent fp64 FMA does not issue back to back);
//   * the first / unfiltered E-step is the same schedule over all k centres.
// Same arithmetic as above, instruction for instruction (seed -|c|^2/2, then fma(x_j, c_j, .) in index order; candidates in
// increasing index order, first maximum wins): labels and scores are bit-identical to the scalar-record kernel.
constexpr int KM2_BLOCKS = 256;
constexpr int KM2_DEPTH = 4;         // records in flight round the evaluation loop: two pairs (the loop body is written for 4)
static_assert(KM2_DEPTH == 4, "the evaluation loop of kmeans_assign_lds_kernel is written for two pairs");
constexpr int KM2_LIST = 512 + 2 * KM2_DEPTH + 8;      // candidate list of a wave: 16-bit LDS offsets, padded
constexpr int KM2_KMAX = 512;        // 8 mask words
constexpr int KM_PK_NMAX = 13;        // coordinates a record of the packed-fp32 pair table holds (2 x 13 + 2 floats of 32)
constexpr int KM2_SUM_MIN = 32;       // lanes that must share a label for their member sums to go through a wave sum and one lane's atomics
constexpr int KM2_NBR_MAX = 256;     // candidates the single-reference form of the filter takes from a sorted row (128 fetched a pass ahead, 128 more on demand)
static_assert(KM2_KMAX * 128 <= 65536, "the candidate lists of kmeans_assign_lds_kernel hold record offsets (c << 7) in 16 bits");
static_assert(KM2_KMAX <= KM_SORT_LABEL_MAX, "the sort keys of the loop's sample order (sortperm.hip) hold the label in KM_SORT_LABEL_BITS bits");
constexpr int KM2_NMAX = 14;         // slot 15 of a record holds -|c|^2/2 (the seed of the DPP chain), slot n the positive half norm
typedef double v2d __attribute__((ext_vector_type(2)));

// Operand constraints of the multi-instruction asm blocks of this file (audited in round 5 after the early-clobber fault of round 4):
//   score_bcast / score2_bcast : outputs "=&v" -- the accumulators are written (seed) while `rec` and x[] are still to be read;
//   sub_bcast, wave_sum_u64x   : every register the block writes is an in/out operand ("+v": it holds a live input, so no other input
//                                can share it); wave_sum_u64x clobbers vcc and says so;
//   v_max_f64 one-liners       : a single instruction reads its sources before it writes -- plain "=v" is right;
//   pk_issue / pk_wait         : "=&s" destinations (three requests read one address pair one after the other); the window between the
//                                two statements is checked on the compiler's listing by tools/isa_sload_window.py (a CPU test).
// x.c - |c|^2/2 for the record held by the DPP rows of `rec`: the seed from slot 15, then fma(x_j, c_j, .) in index order -- one
// asm block (between separate asm statements the compiler pads with s_nop, and a wave issues one instruction per four cycles).
// s_nop 1: a DPP read needs two wait states after a VALU write of its source; the compiler does not see into the asm.
#define KM2_SEED "s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
#define KM2_F(j, op) "v_fmac_f64_dpp %0, %1, %" #op " row_newbcast:" #j " row_mask:0xf bank_mask:0xf\n\t"
#define KM2_F12 KM2_F(0, 2) KM2_F(1, 3) KM2_F(2, 4) KM2_F(3, 5) KM2_F(4, 6) KM2_F(5, 7) KM2_F(6, 8) KM2_F(7, 9) KM2_F(8, 10) KM2_F(9, 11) KM2_F(10, 12) KM2_F(11, 13)
__device__ __forceinline__ double score_bcast(double rec, const double (&x)[12]) {
    double sc;
    asm(KM2_SEED KM2_F12 : "=&v"(sc) : "v"(rec), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]),
        "v"(x[9]), "v"(x[10]), "v"(x[11]));
    return sc;
}
__device__ __forceinline__ double score_bcast(double rec, const double (&x)[13]) {
    double sc;
    asm(KM2_SEED KM2_F12 KM2_F(12, 14) : "=&v"(sc) : "v"(rec), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]),
        "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]), "v"(x[12]));
    return sc;
}
__device__ __forceinline__ double score_bcast(double rec, const double (&x)[15]) {
    double sc;
    asm(KM2_SEED KM2_F12 KM2_F(12, 14) KM2_F(13, 15) KM2_F(14, 16) : "=&v"(sc) : "v"(rec), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]),
        "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]), "v"(x[12]), "v"(x[13]), "v"(x[14]));
    return sc;
}
// two records at once, their chains interleaved: a dependent fp64 FMA cannot issue back to back, so one chain per wave fills
// half of the vector ALU's slots and it takes four waves in this loop at the same time to fill them all (tools/attic/dpp64_probe.hip:
// 80 / 51 / 45 ns per evaluation and SIMD with 1 / 2 / 4 waves) -- with passes as short as 60 candidates they rarely are
#define KM2_SEED2 "s_nop 1\n\tv_mov_b64_dpp %0, %2 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %1, %3 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
#define KM2_G(j, op) "v_fmac_f64_dpp %0, %2, %" #op " row_newbcast:" #j " row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %3, %" #op " row_newbcast:" #j " row_mask:0xf bank_mask:0xf\n\t"
#define KM2_G12 KM2_G(0, 4) KM2_G(1, 5) KM2_G(2, 6) KM2_G(3, 7) KM2_G(4, 8) KM2_G(5, 9) KM2_G(6, 10) KM2_G(7, 11) KM2_G(8, 12) KM2_G(9, 13) KM2_G(10, 14) KM2_G(11, 15)
__device__ __forceinline__ void score2_bcast(double ra, double rb, const double (&x)[12], double& sa, double& sb) {
    asm(KM2_SEED2 KM2_G12 : "=&v"(sa), "=&v"(sb) : "v"(ra), "v"(rb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]),
        "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]));
}
__device__ __forceinline__ void score2_bcast(double ra, double rb, const double (&x)[13], double& sa, double& sb) {
    asm(KM2_SEED2 KM2_G12 KM2_G(12, 16) : "=&v"(sa), "=&v"(sb) : "v"(ra), "v"(rb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]),
        "v"(x[7]), "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]), "v"(x[12]));
}
__device__ __forceinline__ void score2_bcast(double ra, double rb, const double (&x)[15], double& sa, double& sb) {
    asm(KM2_SEED2 KM2_G12 KM2_G(12, 16) KM2_G(13, 17) KM2_G(14, 18) : "=&v"(sa), "=&v"(sb) : "v"(ra), "v"(rb), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]),
        "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]), "v"(x[12]), "v"(x[13]), "v"(x[14]));
}
#undef KM2_G12
#undef KM2_G
#undef KM2_SEED2
#undef KM2_F12
#undef KM2_F
#undef KM2_SEED

// y_j = x_j - c_j for the record held by the DPP rows of `rec` (lane l: double l & 15): v_fmac_f64_dpp with the factor -1 -- the
// product is exact, so the one rounding is that of the subtraction
#define KM_SUBB(j) "v_fmac_f64_dpp %" #j ", %12, %13 row_newbcast:" #j " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void sub_bcast(double rec, double (&y)[12]) {
    const double m1 = -1.0;
    asm("s_nop 1\n\t" KM_SUBB(0) KM_SUBB(1) KM_SUBB(2) KM_SUBB(3) KM_SUBB(4) KM_SUBB(5) KM_SUBB(6) KM_SUBB(7) KM_SUBB(8) KM_SUBB(9) KM_SUBB(10) KM_SUBB(11)
        : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]), "+v"(y[8]), "+v"(y[9]), "+v"(y[10]), "+v"(y[11])
        : "v"(rec), "v"(m1));
}
#define KM_SUBC(j) "v_fmac_f64_dpp %" #j ", %13, %14 row_newbcast:" #j " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void sub_bcast(double rec, double (&y)[13]) {
    const double m1 = -1.0;
    asm("s_nop 1\n\t" KM_SUBC(0) KM_SUBC(1) KM_SUBC(2) KM_SUBC(3) KM_SUBC(4) KM_SUBC(5) KM_SUBC(6) KM_SUBC(7) KM_SUBC(8) KM_SUBC(9) KM_SUBC(10) KM_SUBC(11) KM_SUBC(12)
        : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]), "+v"(y[8]), "+v"(y[9]), "+v"(y[10]), "+v"(y[11]), "+v"(y[12])
        : "v"(rec), "v"(m1));
}
#undef KM_SUBB
#undef KM_SUBC
constexpr int PK_THREADS = 512;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef const float __attribute__((address_space(4)))* cfp_;
typedef const unsigned long long __attribute__((address_space(4)))* cu64p_;

typedef int pk16i __attribute__((ext_vector_type(16)));
typedef int pk8i __attribute__((ext_vector_type(8)));
typedef int pk4i __attribute__((ext_vector_type(4)));
struct PkRec { pk16i a; pk8i b; pk4i c; };           // one pair record in scalar registers: coordinates 0-7 | 8-11 | 12 and the two -|c|^2/2
__device__ __forceinline__ void pk_issue(PkRec& r, cfp_ p) {
    // (early-clobber outputs: the three requests read the address one after the other, none of their destinations may share its registers)
    asm volatile("s_load_dwordx16 %0, %3, 0x0\n\ts_load_dwordx8 %1, %3, 0x40\n\ts_load_dwordx4 %2, %3, 0x60" : "=&s"(r.a), "=&s"(r.b), "=&s"(r.c) : "s"(p));
}
__device__ __forceinline__ void pk_wait(PkRec& r) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r.a), "+s"(r.b), "+s"(r.c)); }
__device__ __forceinline__ v2f pk_pr(int lo, int hi) { v2f v; v[0] = __int_as_float(lo); v[1] = __int_as_float(hi); return v; }
// two candidates through the packed FMAs; fb / fs: the lane's best and second-best float score so far, bp: the pair the best is in
template <int NS>
__device__ __forceinline__ void pk_pair(const PkRec& r, const v2f (&xx)[NS], float& fb, float& fs, int& bp, int t) {
    v2f acc = pk_pr(r.c[2], r.c[3]);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_elementwise_fma(xx[j], pk_pr(r.a[2 * j], r.a[2 * j + 1]), acc);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_elementwise_fma(xx[8 + j], pk_pr(r.b[2 * j], r.b[2 * j + 1]), acc);
    if constexpr (NS == 13) acc = __builtin_elementwise_fma(xx[12], pk_pr(r.c[0], r.c[1]), acc);
    fs = __builtin_amdgcn_fmed3f(fb, acc[0], fs);
    const float b1 = fmaxf(fb, acc[0]);
    fs = __builtin_amdgcn_fmed3f(b1, acc[1], fs);
    const float b2 = fmaxf(b1, acc[1]);
    bp = b2 > fb ? t : bp;
    fb = b2;
}

// LIST (round 4, distance bounds): the pass walks the positions kmeans_bounds_kernel has listed (entry p, or ~p: a padding lane that
// loads p's row and counts for nothing) instead of all N, its member sums are the CHANGES (a sample that moves takes its
// fixed-point coordinates from the old cluster to the new one; kmeans_mstep_kernel adds them to the kept totals), and labels,
// sort keys and bounds are written per listed position.  ubo / lbo != nullptr (either form): the packed-fp32 path leaves the sample's
// bounds there (NaN from every other path: such a sample is evaluated again next time), and the prefix is cut at 2 (1 + beta) u
// instead of 2 u (tscale = (1 + beta)^2): what lies beyond it is then at least (1 + 2 beta) u from every lane -- a lower bound worth keeping.
template <int NS, bool LIST>
__global__ void __launch_bounds__(KM_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
kmeans_assign_lds_kernel(int64_t N, int n, int k, const double* __restrict__ X, int64_t xstride, const double* __restrict__ mean,
                         const double* __restrict__ Ct, int* __restrict__ labels, u64* __restrict__ partial, int nepochs,
                         double* __restrict__ block_inertia, int* __restrict__ block_changed,
                         const float* __restrict__ Dc, const double* __restrict__ prm, float* __restrict__ d2out,
                         const int* __restrict__ perm /* position -> row of X (nullptr: identity); labels and d2out are per position */,
                         const double* __restrict__ fix, const unsigned long long* __restrict__ Nk, const float* __restrict__ Pf,
                         const int* __restrict__ list, const int* __restrict__ nlist, float* __restrict__ ubo, float* __restrict__ lbo, double tscale,
                         long long list_cap /* > 0: the list lies in two regions (kmeans_bounds_kernel with rw2) */) {
    extern __shared__ double lds2[];                  // [k][16] packed centre records | [k][n+1] member sums (fixed point) and count | candidate lists
    if (prm[3] != 0.0) return;                        // hold: an empty cluster waits for its relocation (block-uniform)
#if KM_BLOCKTIME
    const unsigned long long bt0 = __builtin_amdgcn_s_memrealtime();
#endif
    double* tab = lds2;
    u64* sums = reinterpret_cast<u64*>(lds2 + k * 16);
    unsigned short* cand = reinterpret_cast<unsigned short*>(sums + k * (n + 1));      // [16 waves][KM2_LIST]
    const int np1 = n + 1;
    for (int i = threadIdx.x; i < k * 16; i += KM_THREADS) tab[i] = Ct[i];
    for (int i = threadIdx.x; i < k * np1; i += KM_THREADS) sums[i] = 0ull;
    __shared__ double sh_inertia[KM_THREADS / 64];
    __shared__ int sh_changed[KM_THREADS / 64];
    __syncthreads();
    constexpr int NX = NS > 0 ? NS : KM_CMAX;         // coordinates in registers (generic n: slots beyond n hold zeros)
    const cdp_ FS = (cdp_)(unsigned long long)fix;
    double inertia = 0.0;
    int changed = 0;
    int pass = 0, ep = 0;
    double margin = 0.0, eps2 = 0.0;
    bool centres_finite = true;
    if (Dc) { margin = prm[0]; eps2 = prm[1]; centres_finite = prm[2] == 0.0; }
    const bool pk_ok = Pf != nullptr && prm[5] != 0.0;      // packed-fp32 screening (its pair table given, the data's scale far from float's limits)
    const int lane = threadIdx.x & 63;
    const unsigned laneoff = (unsigned)(lane & 15) * 8u;
    auto record = [&](unsigned addr) { return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(tab) + addr); };
    double mm[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) mm[j] = (mean && (NS > 0 || j < n)) ? mean[j] : 0.0;
    const bool vec = NS > 0 && (NS & 1) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 && (xstride & 1) == 0;
    // M: the positions this launch walks (LIST: the entries of the list, whole waves); a slot beyond the end repeats the last one
    const int64_t M = LIST ? (int64_t)nlist[0] + nlist[KM_NL_FRONT] + nlist[KM_NL_BACK] : N;
    auto slot = [&](int64_t base) { const int64_t i = base + threadIdx.x; return i < M ? i : M - 1; };
    auto load_rows = [&](int64_t ii, int prow, double (&xr)[NX], int& lab) {
        const double* row = X + (perm ? (int64_t)prow : ii) * xstride;
        if (vec) {
#pragma unroll
            for (int j = 0; j + 1 < NX; j += 2) { const v2d v = *reinterpret_cast<const v2d*>(row + j); xr[j] = v[0]; xr[j + 1] = v[1]; }
        } else {
#pragma unroll
            for (int j = 0; j < NX; ++j) xr[j] = (NS > 0 || j < n) ? row[j] : 0.0;
        }
        lab = labels[ii];
    };
    const int kw = (k + 63) >> 6;
    const int kp = (k + 255) & ~255;
    const int64_t stride = (int64_t)gridDim.x * KM_THREADS;
    double xn[NX];
    int oln = -1;
    int64_t base = (int64_t)blockIdx.x * KM_THREADS;
    // ---- LIST, dynamic assignment (round 5).  With a fixed share per block a launch lasted 306 us where its mean block was busy for 229
    // and its mean wave for 193 (tools/attic/lloyd_balance.py): a listed pass costs anything between 40 and 512 candidates, and the wide ones
    // come in runs.  Now a wave's pass is a TICKET = 64 consecutive list entries (the list is whole waves).  The first three tickets of
    // every wave are fixed -- (q blocks + block) 16 + wave, q = 0 .. 2: the depth of the prefetch chain --, all later ones are drawn from
    // the device-wide counter nlist[KM_NL_TICKET] (zeroed by the M-step) in BATCHES of 16 per block: a 64-bit LDS word holds (first ticket << 8 |
    // handed out); the wave that finds it empty draws the next batch (one global atomic per 1 024 entries: a single address serves
    // 83 per microsecond, tools/attic/atomic_ticket_probe.hip), the others wait on the LDS word for the few microseconds that takes.
    // The member sums still must not see more than KM_EPOCH_PASSES x KM_THREADS samples between two flushes: a block draws at most
    // that many entries per EPOCH, then all its waves run dry, meet at a barrier, flush and start the next epoch -- at most as many
    // epochs as the fixed shares would have used (kmeans_mstep_kernel derives that number from the list's length), which is capacity
    // enough for the whole list: if tickets were left, every block would have used all of its own.  No result depends on who takes
    // which ticket: labels and bounds are per position, the member sums integers.
    constexpr bool DYN = LIST && KM_LIST_DYNAMIC;
    __shared__ unsigned long long s_tw;               // (first ticket of the block's batch << 16) | tickets handed out (16: none left)
    // the count field is 16 bits wide: every retry of a waiting wave adds one to it (at most 15 waves x 257 tries = 3 855 on top of the
    // 16 of a spent batch), which must never carry into the ticket above it (round-5 review: an 8-bit field could, after 240 retries)
    static_assert(16 + 15 * 257 < (1 << 16), "retries of the waiting waves must fit below the ticket bits");
    __shared__ int s_taken, s_more;
    const int T = (int)(M >> 6);                      // tickets; a value >= T is "none" (N < 2^31: 32 bits hold every ticket drawn)
    const int tk_dyn0 = (int)gridDim.x * (3 * (KM_THREADS / 64));
    constexpr int TK_BATCH = 16, TK_EPOCH = KM_EPOCH_PASSES * (KM_THREADS / 64);
    int* const tctr = LIST ? const_cast<int*>(nlist) + KM_NL_TICKET : nullptr;
    int used_static = 1;
    if constexpr (LIST) {
        const int64_t passes = (M + stride - 1) / stride;
        const int used = (int)((passes + KM_EPOCH_PASSES - 1) / KM_EPOCH_PASSES);
        used_static = used < nepochs ? (used > 1 ? used : 1) : nepochs;
    }
    // the list in two regions: tickets below TW in the front one, the others in [list_cap - (M - 64 TW), list_cap)
    const int TW = (DYN && list_cap > 0) ? nlist[KM_NL_FRONT] >> 6 : T;
    const int64_t tk_back = (DYN && list_cap > 0) ? (int64_t)list_cap - M : 0;
    auto tk_index = [&](int tk) -> int64_t { return (int64_t)tk * 64 + (tk < TW ? 0 : tk_back) + lane; };
    int tk_cur = T, tk_nx = T, tk_nx2 = T, tk_nx3 = T;       // this pass, the next two, and (drawn during a pass, used at its end) the one after
    auto take = [&]() -> int {                        // the next ticket of this wave (wave-uniform)
        int tk = 0;
        if (lane == 0) {
            const int none = 0x7ffffff0;
            for (int tries = 0;; ++tries) {
                const unsigned long long old = atomicAdd(&s_tw, 1ull);
                const unsigned cnt = (unsigned)(old & 0xFFFFull);
                if (cnt < (unsigned)TK_BATCH) { tk = (int)(old >> 16) + (int)cnt; break; }
                if (cnt == (unsigned)TK_BATCH) {
                    if (atomicAdd(&s_taken, TK_BATCH) + TK_BATCH > TK_EPOCH) {      // this epoch's share is drawn: closed until the flush
                        atomicExch(&s_tw, (unsigned long long)none << 16);
                        tk = none;
                    } else {
                        tk = tk_dyn0 + atomicAdd(tctr, TK_BATCH);
                        atomicExch(&s_tw, ((unsigned long long)tk << 16) | 1ull);
                    }
                    break;
                }
                // another wave of the block is fetching the next batch: a few microseconds.  Bounded all the same -- a wave that gives up
                // takes "none" and stops drawing; what it would have drawn stays in the counter for the others (never seen)
                int spins = 0;
                while ((*reinterpret_cast<volatile unsigned long long*>(&s_tw) & 0xFFFFull) >= (unsigned long long)TK_BATCH && ++spins < 4096) __builtin_amdgcn_s_sleep(2);
                if (tries >= 256) { tk = none; break; }
            }
        }
        return __builtin_amdgcn_readfirstlane(tk);
    };
    // sample order through `perm`: the row index of a pass is loaded one pass before its rows are (LIST: the list entry one pass
    // before that).  pos_cur / pos_nx / pos_nx2: positions of this pass, the next (rows in flight) and the one after; dead_*: padding
    int pnext = 0;
    int64_t pos_cur = 0, pos_nx = 0;
    int ent_nx2 = 0;                                  // LIST: the raw entry of the pass after the next
    bool dead_cur = false, dead_nx = false;
    auto entry = [&](int64_t b, int64_t& pos, bool& dead) {
        if constexpr (LIST) { const int e = list[slot(b)]; pos = (int64_t)(e < 0 ? ~e : e); dead = e < 0 || b + threadIdx.x >= M; }
        else { pos = slot(b); dead = b + threadIdx.x >= M; }
    };
    auto entry_t = [&](int tk, int64_t& pos, bool& dead) { const int e = list[tk_index(tk)]; pos = (int64_t)(e < 0 ? ~e : e); dead = e < 0; };
    int a_nx = -1;
    unsigned long long nk0 = 0ull, nk1 = 0ull;
    // reference centre of the single-reference filter for the pass whose rows are in flight, and the head of its sorted row:
    // the label most lanes carry (lane 0's, or the next one when fewer than half the lanes share it)
    auto reference_ahead = [&]() {
        a_nx = -1;
        if (!Nk) return;
        int a = __builtin_amdgcn_readfirstlane(oln);
        const unsigned long long same = __ballot(oln == a);
        if (__builtin_popcountll(same) < 32) a = __builtin_amdgcn_readlane(oln, __builtin_ctzll(~same));
        if ((unsigned)a >= (unsigned)k) return;
        a_nx = a;
        const unsigned long long* row = Nk + (int64_t)a * kp;
        nk0 = row[lane];
        nk1 = row[64 + lane];
    };
#if KM_PROFILE
    unsigned long long t_acc[16] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
#endif
    for (;;) {                                        // DYN: one trip per epoch of the block; otherwise a single trip
    if constexpr (DYN) {
        // the first three tickets of every wave: fixed in the first epoch, one draw for the whole block in a later one
        if (threadIdx.x == 0) {
            s_tw = (unsigned long long)TK_BATCH;
            s_taken = 3 * (KM_THREADS / 64);
            s_more = ep == 0 ? (int)blockIdx.x * (KM_THREADS / 64) : tk_dyn0 + atomicAdd(tctr, 3 * (KM_THREADS / 64));
        }
        __syncthreads();
        {
            const int per = ep == 0 ? (int)gridDim.x * (KM_THREADS / 64) : KM_THREADS / 64;
            tk_cur = __builtin_amdgcn_readfirstlane(s_more + (int)(threadIdx.x >> 6));       // (wave-uniform, and the compiler should know)
            tk_nx = tk_cur + per; tk_nx2 = tk_nx + per;
        }
        if (tk_cur < T) {
            entry_t(tk_cur, pos_cur, dead_cur);
            load_rows(pos_cur, perm[pos_cur], xn, oln);
            if (tk_nx < T) { entry_t(tk_nx, pos_nx, dead_nx); pnext = perm[pos_nx]; }
            if (tk_nx2 < T) ent_nx2 = list[tk_index(tk_nx2)];
            reference_ahead();
        }
    } else {
    if (base < M) {
        entry(base, pos_cur, dead_cur);
        load_rows(pos_cur, perm ? perm[pos_cur] : 0, xn, oln);
        if (base + stride < M) {
            entry(base + stride, pos_nx, dead_nx);
            if (perm) pnext = perm[pos_nx];
        }
        if constexpr (LIST) { if (base + 2 * stride < M) ent_nx2 = list[slot(base + 2 * stride)]; }
    }
    if (base < M) reference_ahead();
    }
// which of the passes ahead exist (DYN: each stage of the prefetch chain is guarded by its own ticket)
#define KM_HAS_NX (DYN ? tk_nx < T : base + stride < M)
#define KM_HAS_NX2 (DYN ? tk_nx2 < T : base + 2 * stride < M)
#define KM_HAS_NX3 (DYN ? tk_nx3 < T : base + 3 * stride < M)
    for (; DYN ? tk_cur < T : base < M; base += DYN ? 0 : stride) {      // (a wave's tickets only grow: "none" is never followed by a pass)
#if KM_PROFILE
        unsigned long long t_prev = __builtin_readcyclecounter();
#endif
        const int64_t i = pos_cur;                    // the position this lane works on (labels, sort keys and bounds are per position)
        const bool live = !dead_cur;
        double x[NX], x2 = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) { x[j] = xn[j] - mm[j]; x2 = fma(x[j], x[j], x2); }
        const int ol = oln;                           // label of the previous iteration (-1 before the first)
        const int a_ref = a_nx;
        const unsigned long long key0 = nk0, key1 = nk1;
#if KM_PROFILE
        asm volatile("; rows have arrived" :: "v"(x2), "v"(ol));
        KM_STAMP(0);
#endif
        unsigned long long mws[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
        unsigned short* lst = cand + (threadIdx.x >> 6) * KM2_LIST;
        // ---- candidate filter, first form (round 4): ONE reference centre per wave.  For any centre a, a centre c with
        // d(c_a, c) >= 2 u + margin, u = max over the wave's lanes of d(x, c_a), is farther from every lane than c_a is by at least
        // `margin` -- whatever labels the lanes carry: the old labels only pick a good reference (the label most lanes share; in the
        // loop's sorted order a wave is one cluster's samples at one radius, give or take the few whose label has moved since the
        // sort -- as a second label GROUP those cost a second candidate set, as lanes of this one a slightly larger u).  The candidates
        // are a PREFIX of row a of the sorted centre distances (kmeans_cdist_kernel: Nk): no masks, no compaction; the 128 keys at the
        // head of the row are requested a pass ahead, as soon as the next pass's labels have arrived.  They are
        // evaluated in the order of that row, not by index, so an exact tie between two candidates' scores could fall to the other
        // one than in the full scan (first of equal maxima by INDEX): every evaluation that equals the running best raises a flag,
        // and a flagged wave repeats the pass through the mask form below (index order).  Exact duplicates of a centre are neighbours
        // in the row in index order, but they do tie: such waves always take the second form.
        double u2_ref = 0.0;                          // the wave's squared radius about its reference centre (single-reference form)
        double d2_ref = 0.0, t2_ref = 0.0;            // this lane's squared distance to that centre; the squared radius the prefix was cut at
        // LIST (bounds from every path, round 5): what the selection used last leaves for the centres it did NOT hand to the evaluation --
        // they are at least sqrt(far_t2) from the centre this lane's squared distance far_d2 refers to (all k evaluated: nothing is left out)
        float far_t2 = 3.0e38f, far_d2 = 0.0f;
        auto select_by_nbr = [&](int& ncand) -> bool {
            if (a_ref < 0 || __ballot(!((x2 - x2 == 0.0) && centres_finite)) != 0ull) return false;
            const int a = a_ref;
            const double sc = score_bcast(record(((unsigned)a << 7) + laneoff), x);
            const double d2 = fma(-2.0, sc, x2);                           // |x - c_a|^2 up to rounding
            float rf = (float)fmax(d2, 0.0);                               // squared radius as a float rounded UP
            rf = rf * 1.0000005f + 1.0e-37f;
            const unsigned rb = wave_max_u32(__float_as_uint(rf));
            const double u2 = (double)__uint_as_float(rb) + eps2;          // >= the true squared radius u^2 of the wave about c_a
            u2_ref = u2;
            d2_ref = d2;
            const double t2 = fma(4.004 * tscale, u2, 1001.0 * margin * margin);    // >= (2 u + margin)^2 (tscale = 1)
            t2_ref = t2;
            const unsigned tb = __float_as_uint(fminf((float)(t2 * 1.0000001) + 1.0e-37f, 3.4028234e38f));     // non-negative floats order like their bits
            int cnt = __builtin_popcountll(__ballot((unsigned)(key0 >> 16) < tb)) + __builtin_popcountll(__ballot((unsigned)(key1 >> 16) < tb));
            lst[lane] = (unsigned short)(((unsigned)key0 & 0xFFFFu) << 7);
            lst[64 + lane] = (unsigned short)(((unsigned)key1 & 0xFFFFu) << 7);
            if (cnt == 128) {
                // a wide wave (the outermost samples of a cluster, a cluster boundary of the order): the next 128 keys of the row, on demand
                if (kp < 256 + 0) return false;
                const unsigned long long* row = Nk + (int64_t)a * kp;
                const unsigned long long key2 = row[128 + lane], key3 = row[192 + lane];
                cnt += __builtin_popcountll(__ballot((unsigned)(key2 >> 16) < tb)) + __builtin_popcountll(__ballot((unsigned)(key3 >> 16) < tb));
                if (cnt >= KM2_NBR_MAX) return false;                      // wider still (unsorted data): second form
                lst[128 + lane] = (unsigned short)(((unsigned)key2 & 0xFFFFu) << 7);
                lst[192 + lane] = (unsigned short)(((unsigned)key3 & 0xFFFFu) << 7);
            }
            const unsigned short last = lst[cnt - 1];                      // (cnt >= 1: c_a itself is at distance 0)
            if (lane < 2 * KM2_DEPTH + 2) lst[cnt + lane] = last;          // the tail repeats the last candidate
            ncand = cnt;
            if constexpr (LIST && KM_FP64_BOUNDS) { far_t2 = __uint_as_float(tb); far_d2 = (float)fmax(d2, 0.0) * 1.0000005f + 1.0e-37f; }
            return true;
        };
        // ---- second form (round 3): label groups, masks over all k centres, candidates in index order
        auto select_by_masks = [&]() -> bool {
            bool filtered = false;
            const bool usable = (unsigned)ol < (unsigned)k && (x2 - x2 == 0.0) && centres_finite;
            if (__ballot(!usable) == 0ull) {
                int ga[KM_GMAX] = {0, 0, 0, 0, 0, 0, 0, 0};
                float tf[KM_GMAX] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                int ng = 0;
                unsigned long long remaining = ~0ull;
#pragma unroll
                for (int g = 0; g < KM_GMAX; ++g) {
                    if (remaining != 0ull) {
                        const int lead = __builtin_ctzll(remaining);
                        const int a = __builtin_amdgcn_readlane(ol, lead);
                        const bool mine = ol == a;
                        const double sc = score_bcast(record(((unsigned)a << 7) + laneoff), x);
                        const double d2 = fma(-2.0, sc, x2);                       // |x - c_a|^2 up to rounding
                        float rf = mine ? (float)fmax(d2, 0.0) : 0.0f;             // squared radius as a float rounded UP
                        rf = rf * 1.0000005f + 1.0e-37f;
                        const unsigned rb = wave_max_u32(mine ? __float_as_uint(rf) : 0u);
                        const double u2 = (double)__uint_as_float(rb) + eps2;      // >= the true squared radius u^2 of the group
                        const double t2 = fma(4.004, u2, 1001.0 * margin * margin);   // >= (2 u + margin)^2
                        const float t = fminf((float)(t2 * 1.0000001) + 1.0e-37f, 3.4028234e38f);
                        ga[g] = a;
                        tf[g] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(t)));
                        ng = g + 1;
                        if constexpr (LIST && KM_FP64_BOUNDS) { if (mine) { far_t2 = tf[g]; far_d2 = rf; } }
                        remaining &= ~__ballot(mine);
                    }
                }
                KM_STAMP(1);
                if (remaining == 0ull) {
                    filtered = true;
                    // candidate masks: the distance-table rows of 2 or 4 groups (both blocks of 256 centres) are requested together;
                    // groups beyond ng repeat group 0 with threshold 0 (no float is below it)
                    auto masks = [&](auto NGc, auto TWOc, int g0) {
                        constexpr int NG_ = decltype(NGc)::value;
                        constexpr bool TWO_ = decltype(TWOc)::value;
                        bool p0 = false, p1 = false, p2 = false, p3 = false, p4 = false, p5 = false, p6 = false, p7 = false;
#pragma unroll
                        for (int g = 0; g < NG_; ++g) {
                            int a = ga[0];
                            float t = 0.0f;
#pragma unroll
                            for (int q = 0; q < KM_GMAX; ++q)
                                if (q == g0 + g && q < ng) { a = ga[q]; t = tf[q]; }
                            const float* row = Dc + (int64_t)a * kp + lane * 4;
                            const float4 v = *reinterpret_cast<const float4*>(row);
                            p0 = p0 || v.x < t; p1 = p1 || v.y < t; p2 = p2 || v.z < t; p3 = p3 || v.w < t;
                            if constexpr (TWO_) {
                                const float4 w = *reinterpret_cast<const float4*>(row + 256);
                                p4 = p4 || w.x < t; p5 = p5 || w.y < t; p6 = p6 || w.z < t; p7 = p7 || w.w < t;
                            }
                        }
                        mws[0] |= __ballot(p0); mws[1] |= __ballot(p1); mws[2] |= __ballot(p2); mws[3] |= __ballot(p3);
                        if constexpr (TWO_) { mws[4] |= __ballot(p4); mws[5] |= __ballot(p5); mws[6] |= __ballot(p6); mws[7] |= __ballot(p7); }
                    };
                    using std::integral_constant;
                    if (kw > 4) {
                        if (ng <= 2) masks(integral_constant<int, 2>{}, std::true_type{}, 0);
                        else {
                            masks(integral_constant<int, 4>{}, std::true_type{}, 0);
                            if (ng > 4) masks(integral_constant<int, 4>{}, std::true_type{}, 4);
                        }
                    } else {
                        if (ng <= 2) masks(integral_constant<int, 2>{}, std::false_type{}, 0);
                        else {
                            masks(integral_constant<int, 4>{}, std::false_type{}, 0);
                            if (ng > 4) masks(integral_constant<int, 4>{}, std::false_type{}, 4);
                        }
                    }
                }
            }
                    return filtered;
        };
        bool filtered = false, by_nbr = false;
        int ncand_nbr = 0;
        if (Nk) by_nbr = select_by_nbr(ncand_nbr);
        if (!by_nbr && Dc) filtered = select_by_masks();
        KM_STAMP(2);
        // rows of the next pass: in flight during the evaluation (which waits on the LDS only)
        if (KM_HAS_NX) {
            load_rows(pos_nx, pnext, xn, oln);
            if constexpr (LIST) {
                if (KM_HAS_NX2) pnext = perm[ent_nx2 < 0 ? ~ent_nx2 : ent_nx2];
            } else {
                if (perm && base + 2 * stride < M) pnext = perm[slot(base + 2 * stride)];
            }
        }
        // ---- evaluation in increasing index order, KM2_DEPTH records in flight
        double best = -1.0e300;
        unsigned baddr = laneoff;                     // LDS address of the best record so far: the index is baddr >> 7
        unsigned addr[KM2_DEPTH];
        double rec[KM2_DEPTH];
        unsigned long long tiem = 0ull;               // lanes that saw a score EQUAL to their running best (first form only)
        unsigned long long tq[KM2_DEPTH] = {0ull, 0ull, 0ull, 0ull};      // ... per position of the trip (the padded tail of a list must not count)
        // LIST: the runner-up's score is kept too (sec = the largest score among the evaluated candidates but the first best: a repeated
        // candidate -- the padded tail of a list, a duplicate centre -- can only RAISE it, i.e. lower the bound it gives)
        double sec = -1.0e300;
        auto eval2 = [&](int d, auto TIE) {          // candidates d and d + 1 of the ring, in this order
            double sa, sb;
            score2_bcast(rec[d], rec[d + 1], x, sa, sb);
            // strict '>' to replace: the first maximum wins, like np.argmin on the distances
            if constexpr (decltype(TIE)::value) tq[d] = __ballot(sa == best);
            baddr = (sa <= best) ? baddr : addr[d];
            if constexpr (LIST && KM_FP64_BOUNDS) { double lo_; asm("v_min_f64 %0, %1, %2" : "=v"(lo_) : "v"(best), "v"(sa)); asm("v_max_f64 %0, %1, %2" : "=v"(sec) : "v"(sec), "v"(lo_)); }
            asm("v_max_f64 %0, %1, %2" : "=v"(best) : "v"(best), "v"(sa));
            if constexpr (decltype(TIE)::value) tq[d + 1] = __ballot(sb == best);
            baddr = (sb <= best) ? baddr : addr[d + 1];
            if constexpr (LIST && KM_FP64_BOUNDS) { double lo_; asm("v_min_f64 %0, %1, %2" : "=v"(lo_) : "v"(best), "v"(sb)); asm("v_max_f64 %0, %1, %2" : "=v"(sec) : "v"(sec), "v"(lo_)); }
            asm("v_max_f64 %0, %1, %2" : "=v"(best) : "v"(best), "v"(sb));
        };
        auto compact = [&]() -> int {
            // the candidates of the eight mask words as one list of LDS record offsets (128 c as 16 bits: k <= 512), written by
            // the lanes that hold the bits; the tail repeats the last candidate (an equal score never replaces the best)
            int ncand = 0, lastc = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const unsigned long long m = mws[q];
                if (m != 0ull) {
                    const int rank = ncand + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    if ((m >> lane) & 1ull) lst[rank] = (unsigned short)((q * 64 + lane) << 7);
                    ncand += __builtin_popcountll(m);
                    lastc = q * 64 + 63 - __builtin_clzll(m);
                }
            }
            if (lane < 2 * KM2_DEPTH + 2) lst[ncand + lane] = (unsigned short)(lastc << 7);
            return ncand;
        };
        auto run_list = [&](int ncand, auto TIE) {
            // The trip evaluates pair A = candidates (4t, 4t + 1), then pair B = (4t + 2, 4t + 3), and ENDS with an evaluation: the
            // compiler waits for every LDS read at the loop head, so the reads issued last before it must be an evaluation old.
            //   head: records of B (offsets cn[2..3], read a trip ago) | evaluate A | records of the next A | evaluate B
            unsigned cn[KM2_DEPTH];
#pragma unroll
            for (int d = 0; d < 2; ++d) { addr[d] = (unsigned)lst[d] + laneoff; rec[d] = record(addr[d]); cn[d] = lst[KM2_DEPTH + d]; cn[2 + d] = lst[2 + d]; }
            const int trips = (ncand + KM2_DEPTH - 1) / KM2_DEPTH;
            const unsigned short* lp = lst + KM2_DEPTH + 2;
#pragma unroll 1
            for (int t = 0; t < trips; ++t, lp += KM2_DEPTH) {
#pragma unroll
                for (int e = 2; e < 4; ++e) { addr[e] = cn[e] + laneoff; rec[e] = record(addr[e]); cn[e] = lp[e - 2]; }
                __builtin_amdgcn_sched_barrier(0);
                eval2(0, TIE);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 2; ++e) { addr[e] = cn[e] + laneoff; rec[e] = record(addr[e]); cn[e] = lp[e + 2]; }
                __builtin_amdgcn_sched_barrier(0);
                eval2(2, TIE);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (decltype(TIE)::value) {
                    // positions of the last trip beyond the list repeat its last candidate: their equality with the best is no tie
                    const int rem = ncand - t * KM2_DEPTH;
                    tiem |= tq[0] | (rem > 1 ? tq[1] : 0ull) | (rem > 2 ? tq[2] : 0ull) | (rem > 3 ? tq[3] : 0ull);
                }
            }
        };
        auto run_all = [&]() {
            // all k centres, same schedule
            auto off = [&](int c) { return ((unsigned)min(c, k - 1) << 7) + laneoff; };
#pragma unroll
            for (int d = 0; d < 2; ++d) { addr[d] = off(d); rec[d] = record(addr[d]); }
            const int trips = (k + KM2_DEPTH - 1) / KM2_DEPTH;
#pragma unroll 1
            for (int t = 0; t < trips; ++t) {
#pragma unroll
                for (int e = 2; e < 4; ++e) { addr[e] = off(t * KM2_DEPTH + e); rec[e] = record(addr[e]); }
                __builtin_amdgcn_sched_barrier(0);
                eval2(0, std::false_type{});
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 2; ++e) { addr[e] = off((t + 1) * KM2_DEPTH + e); rec[e] = record(addr[e]); }
                __builtin_amdgcn_sched_barrier(0);
                eval2(2, std::false_type{});
                __builtin_amdgcn_sched_barrier(0);
            }
                };
        bool by_pk = false;
        float ubv = __builtin_nanf(""), lbv = __builtin_nanf("");      // the sample's new bounds (packed-fp32 path only)
        if constexpr (NS == 12 || NS == 13) {
            if (by_nbr && pk_ok) {
                // ---- packed-fp32 screening of the prefix (see kmeans_assign_pk_kernel for the argument): pair records through scalar
                // registers in the frame of the reference centre; certified waves evaluate in fp64 only the pair the best lies in
                const double rec_a = record(((unsigned)a_ref << 7) + laneoff);
                double yd[NX];
#pragma unroll
                for (int j = 0; j < NX; ++j) yd[j] = x[j];
                sub_bcast(rec_a, yd);
                v2f xx[NX];
#pragma unroll
                for (int j = 0; j < NX; ++j) { const float f = (float)yd[j]; xx[j][0] = f; xx[j][1] = f; }
                const float mf = (float)(8.0e-6 * u2_ref) * 1.001f + 1.0e-37f;
                float fb = -3.0e38f, fs = -3.0e38f;
                int bp = 0;
                const int npairs2 = (((ncand_nbr + 1) >> 1) + 1) & ~1;       // <= kmeans_lds_pf_pairs() (static_assert there): the records cdist built
                const cfp_ rows = (cfp_)(unsigned long long)Pf + (int64_t)a_ref * (kp >> 1) * 32;
                // Two records per round trip, worked on together (two independent FMA chains: no wait states between them).  With
                // four waves on a SIMD and a record that comes from the L2 (each pass walks another cluster's row: the scalar cache
                // is cold) a wave spends most of a trip waiting whatever the schedule; twice the work per trip is what counts --
                // one record ahead of the arithmetic (the form of kmeans_assign_pk_kernel) was 1.6 x slower here.
                PkRec recA, recB;
#pragma unroll 1
                for (int t = 0; t < npairs2; t += 2) {
                    pk_issue(recA, rows + t * 32);
                    pk_issue(recB, rows + (t + 1) * 32);
                    pk_wait(recA);
                    pk_wait(recB);
                    pk_pair<NX>(recA, xx, fb, fs, bp, t);
                    pk_pair<NX>(recB, xx, fb, fs, bp, t + 1);
                }
                if (__ballot(!(fb - fs > mf)) == 0ull) {
                    // the two members of pair bp: entries 2 bp and 2 bp + 1 of the wave's list (a member past the prefix repeats the last
                    // candidate there: it cannot be the argmax, see DESIGN f2), their records gathered from the LDS table
                    const unsigned o0 = lst[2 * bp], o1 = lst[2 * bp + 1];
                    auto gather = [&](unsigned off) {
                        const double* r = reinterpret_cast<const double*>(reinterpret_cast<const char*>(tab) + off);
                        double sc = fma(x[0], r[0], -r[NX]);
#pragma unroll
                        for (int j = 1; j < NX; ++j) sc = fma(x[j], r[j], sc);
                        return sc;
                    };
                    const double s0 = gather(o0), s1 = gather(o1);
                    const bool second = s1 > s0 || (s1 == s0 && o1 < o0);
                    best = second ? s1 : s0;
                    baddr = (second ? o1 : o0) + laneoff;
                    by_pk = true;
                    if (ubo) {
                        // bounds for kmeans_bounds_kernel.  Upper: the exact distance to the winner.  Lower, over every other centre:
                        //   the pair's other member -- its exact score is at hand (o1 == o0: there is none, the list's padding);
                        //   any other candidate of the scanned prefix: its float score is at most the runner-up's fs and errs by less
                        //     than mf, so in the reference centre's frame |x - c|^2 = |y|^2 - 2 S_c >= |y|^2 - 2 (fs + mf);
                        //   any centre beyond the prefix: at least sqrt(t2) from the reference centre, |y| of which the lane can cover.
                        const double dbest2 = fma(-2.0, best, x2);
                        const double dlose2 = o1 != o0 ? fma(-2.0, second ? s0 : s1, x2) : 1.0e300;
                        const double dscan2 = fma(-2.0, (double)fs + (double)mf, d2_ref);
                        const double l2 = fmin(dlose2, dscan2) - 4.0 * eps2;
                        const double lfar = sqrt(t2_ref) * 0.9999999 - sqrt(fmax(d2_ref, 0.0) + eps2) * 1.0000001;
                        const double lbd = fmin(l2 > 0.0 ? sqrt(l2) : 0.0, lfar);
                        ubv = (float)sqrt(fmax(dbest2, 0.0) + eps2) * 1.0000003f + 1.0e-37f;
                        lbv = (float)lbd;
                        lbv = lbv > 0.0f ? lbv * 0.9999997f : lbv;
                    }
                }
            }
        }
        if (by_pk) {
            // (label and score are the full scan's: nothing more to evaluate)
#if KM_PROFILE
            t_acc[14] += 1ull; t_acc[15] += (unsigned long long)ncand_nbr;
#endif
        } else if (by_nbr) {
            run_list(ncand_nbr, std::true_type{});
#if KM_PROFILE
            t_acc[8] += 1ull; t_acc[9] += (unsigned long long)ncand_nbr; t_acc[10] += tiem != 0ull ? 1ull : 0ull;
#endif
            if (tiem != 0ull) {                       // an exact tie somewhere: once more in index order
                best = -1.0e300;
                sec = -1.0e300;
                baddr = laneoff;
                if (select_by_masks()) run_list(compact(), std::false_type{}); else { far_t2 = 3.0e38f; far_d2 = 0.0f; run_all(); }
            }
        } else if (filtered) {
            const int nc_ = compact();
#if KM_PROFILE
            t_acc[11] += 1ull; t_acc[12] += (unsigned long long)nc_;
#endif
            run_list(nc_, std::false_type{});
        } else {
#if KM_PROFILE
            t_acc[13] += 1ull;
#endif
            far_t2 = 3.0e38f; far_d2 = 0.0f;
            run_all();
        }
        if constexpr (LIST && KM_FP64_BOUNDS) {
            if (!by_pk && ubo) {
                // bounds from the fp64 paths too (round 5; until then they left NaN, "evaluate again", and one far-out sample kept the 63
                // it shares a wave with on the list -- and in the widest, most expensive passes -- for good).  Upper: the exact distance
                // to the winner.  Lower: the runner-up among the evaluated candidates, exact; every centre that was not evaluated is at
                // least sqrt(far_t2) from the centre the selection measured this lane's far_d2 against.
                const double dbest2 = fma(-2.0, best, x2);
                const double l2 = fma(-2.0, sec, x2) - 4.0 * eps2;
                const double lfar = sqrt((double)far_t2) * 0.9999999 - sqrt((double)far_d2 + eps2) * 1.0000001;
                const double lbd = fmin(l2 > 0.0 ? sqrt(fmin(l2, 1.0e300)) : 0.0, lfar);
                ubv = (float)sqrt(fmax(dbest2, 0.0) + eps2) * 1.0000003f + 1.0e-37f;
                lbv = (float)fmin(lbd, 3.0e38);
                lbv = lbv > 0.0f ? lbv * 0.9999997f : lbv;
                if (!(x2 - x2 == 0.0)) { ubv = __builtin_nanf(""); lbv = __builtin_nanf(""); }      // a non-finite sample: as before
            }
        }
        const int bi = (int)(baddr >> 7);
        if (KM_HAS_NX) reference_ahead();             // the next pass's labels have arrived during the evaluation
        // DYN: the ticket three passes ahead, drawn as late as the prefetch chain allows (its list entry is asked for at the end of this
        // pass) -- what a wave holds when the counter runs dry is the launch's tail
        if constexpr (DYN) tk_nx3 = KM_HAS_NX2 ? take() : T;
        KM_STAMP(3);
        if (live) {
            if (ol != bi) ++changed;
            labels[i] = bi;
            const double dmin2 = fma(-2.0, best, x2);     // squared distance to the chosen centre (up to rounding)
            if (d2out) d2out[i] = (float)dmin2;            // sort key of the loop's sample order (sortperm.hip)
            inertia += dmin2;
            if (ubo) { ubo[i] = ubv; lbo[i] = lbv; }
        }
        // member sums, fixed point (see the kernel above).  Same-address atomics are what this phase costs (see wave_sum_u64x), so
        // when at least half of the wave's lanes went to one centre -- a sorted wave: all of them but the few whose label has moved
        // since the sort -- their coordinates are summed over the wave first, the other lanes adding zeros, and sent by ONE lane; the
        // rest -- lanes of other labels, of a non-finite sample (zeros and the poison flag) -- send their own atomics.  (Scans: the
        // same for up to 2 or 4 labels per pass that 8, 16 or 24 lanes share: 264-268 ms per 300 iterations against 261-264 for
        // this form; without it, every lane of a wave with a single moved label sent its own: 270 ms.)
        const bool bad = !(x2 - x2 == 0.0);
        const bool ok = live && !bad;
        bool sent = false;
        if constexpr (LIST) {
            // CHANGES of the member sums: a sample that moved leaves its old cluster and joins the new one (two's-complement words:
            // the block's partial table holds signed differences; kmeans_mstep_kernel adds them to the totals it keeps)
            sent = true;
            if (live && ol != bi && (unsigned)ol < (unsigned)k) {
                u64* sn = sums + bi * np1;
                u64* so = sums + ol * np1;
#pragma unroll
                for (int j = 0; j < NX; ++j) {
                    const u64 q = km_fix(bad ? 0.0 : x[j], FS[j]);
                    atomicAdd(&sn[j], q);
                    atomicAdd(&so[j], 0ull - q);
                }
                const u64 one = bad ? 1ull + KM_POISON : 1ull;
                atomicAdd(&sn[n], one);
                atomicAdd(&so[n], 0ull - one);
            }
        }
        if constexpr (!LIST && (NS == 12 || NS == 13)) {
            const unsigned long long okm = __ballot(ok);
            if (okm != 0ull) {
                const int lab = __builtin_amdgcn_readlane(bi, __builtin_ctzll(okm));
                const bool mine = ok && bi == lab;
                const int cnt = __builtin_popcountll(__ballot(mine));
                if (cnt >= KM2_SUM_MIN) {
                    unsigned qlo[NX], qhi[NX];
#pragma unroll
                    for (int j = 0; j < NX; ++j) { const u64 q = mine ? km_fix(x[j], FS[j]) : 0ull; qlo[j] = (unsigned)q; qhi[j] = (unsigned)(q >> 32); }
                    wave_sum_u64x(qlo, qhi);
                    if (lane == 63) {                 // one lane, one atomic per coordinate: no two lanes on an address
                        u64* s = sums + lab * np1;
#pragma unroll
                        for (int j = 0; j < NX; ++j) atomicAdd(&s[j], ((u64)qhi[j] << 32) | qlo[j]);
                        atomicAdd(&s[n], (u64)cnt);
                    }
                    sent = mine;
                }
            }
        }
        if (live && !sent) {
            u64* s = sums + bi * np1;
#pragma unroll
            for (int j = 0; j < NX; ++j)
                if (NS > 0 || j < n) atomicAdd(&s[j], km_fix(bad ? 0.0 : x[j], FS[j]));
            atomicAdd(&s[n], bad ? 1ull + KM_POISON : 1ull);
        }
#if KM_PROFILE
        KM_STAMP(4);
        t_acc[7] += 1ull;
#endif
        // the positions move up a pass
        pos_cur = pos_nx;
        dead_cur = dead_nx;
        if (KM_HAS_NX2) {
            if constexpr (LIST) {
                pos_nx = (int64_t)(ent_nx2 < 0 ? ~ent_nx2 : ent_nx2);
                dead_nx = ent_nx2 < 0 || (!DYN && base + 2 * stride + threadIdx.x >= M);
                if (KM_HAS_NX3) ent_nx2 = list[DYN ? tk_index(tk_nx3) : slot(base + 3 * stride)];
            } else {
                entry(base + 2 * stride, pos_nx, dead_nx);
            }
        }
        if constexpr (DYN) { tk_cur = tk_nx; tk_nx = tk_nx2; tk_nx2 = tk_nx3; }
        else { if (++pass == KM_EPOCH_PASSES && base + stride < M) { km_flush(sums, partial, ep, k, n, true); ++ep; pass = 0; } }
    }
#undef KM_HAS_NX
#undef KM_HAS_NX2
#undef KM_HAS_NX3
    if constexpr (!DYN) break;
    else {
        // the block's share of this epoch is drawn, or the list is: every wave has run dry
        __syncthreads();
        if (threadIdx.x == 0) s_more = (ep + 1 < used_static && tk_dyn0 + __hip_atomic_load(tctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < T) ? 1 : 0;
        __syncthreads();
        if (!s_more) break;
        km_flush(sums, partial, ep, k, n, true);
        ++ep;
    }
    }
#if KM_PROFILE
    if (threadIdx.x % 64 == 0) for (int q = 0; q < 16; ++q) atomicAdd(&km_prof[q], t_acc[q]);
#endif
#if KM_BLOCKTIME
    if (LIST && lane == 0) { atomicAdd(&km_blk[2], __builtin_amdgcn_s_memrealtime() - bt0); atomicAdd(&km_blk[3], 1ull); }
#endif
    for (int off = 32; off > 0; off >>= 1) {
        inertia += __shfl_down(inertia, off);
        changed += __shfl_down(changed, off);
    }
    if ((threadIdx.x & 63) == 0) { sh_inertia[threadIdx.x >> 6] = inertia; sh_changed[threadIdx.x >> 6] = changed; }
    __syncthreads();
#if KM_BLOCKTIME
    if (LIST && threadIdx.x == 0) { atomicAdd(&km_blk[0], __builtin_amdgcn_s_memrealtime() - bt0); atomicAdd(&km_blk[1], 1ull); }
#endif
    if (threadIdx.x == 0) {
        double in = 0.0;
        int ch = 0;
        for (int q = 0; q < KM_THREADS / 64; ++q) { in += sh_inertia[q]; ch += sh_changed[q]; }
        block_inertia[blockIdx.x] = in;
        block_changed[blockIdx.x] = ch;
    }
    km_flush(sums, partial, ep, k, n, false);
    if constexpr (LIST) {
        // only the epochs the longest-running block reaches: kmeans_mstep_kernel derives the same number from the list's length
        km_zero_epochs(partial, ep + 1, used_static, k, n);
    } else {
        km_zero_epochs(partial, ep + 1, nepochs, k, n);
    }
}

// ---- E-step, third form (round 4): candidates screened in PACKED fp32, the exact arithmetic only for the winner ------------------
// The evaluation is what an E-step costs (two thirds of a wave's pass in the LDS / DPP kernel: 17-19 fp64 issue slots per candidate
// and 64 samples), and v_pk_fma_f32 does two candidates per slot.  With the single-reference filter the candidates of a wave are a
// prefix of its reference centre's sorted row, so consecutive candidates can be PAIRED once per iteration: kmeans_cdist_kernel
// writes every row as float pair records (Pf), a wave streams the prefix through scalar registers -- the (c0_j, c1_j) pair is the
// scalar operand of the packed FMA, the sample's coordinate the vector one -- and keeps, per lane, the best and second-best float
// score and the pair the best came from: 20 slots per PAIR.  Then, per lane:
//   certified  (best - second > mf): the exact fp64 argmax lies in that pair -- every other candidate's float score is lower by more
//              than twice the screening error.  The screening works in the frame of the reference centre (y = x - c_a against
//              d = c - c_a: the exact score of c minus that of c_a), where every term is of the size of the wave's radius u: float
//              inputs 2^-24 relative each, thirteen float FMAs, |error| <= 16 * 2^-24 * (|d|^2 / 2 + |y| |d|) <= 3.8e-6 u^2, mf =
//              8e-6 u^2.  (In the data's own frame the same bound is 1.4e-6 max |x|^2 -- a hundred times the gap between a
//              sample's two best centres, and 28 % of the waves held a lane that could not be certified.)  The centres outside
//              the prefix are out by the triangle inequality as before.  Both members are evaluated in fp64 -- the full scan's own chain,
//              seed -|c|^2/2, fma(x_j, c_j, .) by index -- and the larger wins (equal: the lower index): label and score are the
//              full scan's bit for bit;
//   otherwise  (a near-tie of two centres, 1e-4 of the samples; any lane of the wave): the wave walks the same prefix in fp64,
//              first maximum by INDEX.
// A wave without a usable reference (first E-step, non-finite sample or centre) takes the full scan over all k centres.
// 512-thread blocks, three per CU (the member sums are the only LDS table; the exact records come from L1 / L2), six waves per
// SIMD: the pair records arrive by SMEM, which can only be waited for as a whole -- other waves fill the gap.
// k = 513 ... 1024 (the LDS / DPP kernel's table and offsets end at 512): the member sums take up to 106 KB, one block per CU -- of
// 1024 threads (TH), four waves per SIMD; the loop's sorted order, the sorted rows and the pair records are the same.
template <int NS, int TH>
__global__ void __launch_bounds__(TH) __attribute__((amdgpu_waves_per_eu(6, 6)))
kmeans_assign_pk_kernel(int64_t N, int n, int k, const double* __restrict__ X, int64_t xstride, const double* __restrict__ mean,
                        const double* __restrict__ Ct, int* __restrict__ labels, u64* __restrict__ partial, int nepochs,
                        double* __restrict__ block_inertia, int* __restrict__ block_changed, const double* __restrict__ prm,
                        float* __restrict__ d2out, const int* __restrict__ perm, const double* __restrict__ fix,
                        const unsigned long long* __restrict__ Nk, const float* __restrict__ Pf) {
    static_assert(NS == 12 || NS == 13, "the pair records hold 12 or 13 coordinates");
    extern __shared__ u64 sums[];                     // [k][n+1]: member sums (fixed point) and count
    if (prm[3] != 0.0) return;                        // hold (block-uniform)
    const int np1 = n + 1;
    for (int i = threadIdx.x; i < k * np1; i += TH) sums[i] = 0ull;
    __shared__ double sh_inertia[TH / 64];
    __shared__ int sh_changed[TH / 64];
    __syncthreads();
    const ccp T = (ccp)(unsigned long long)Ct;
    const cdp_ FS = (cdp_)(unsigned long long)fix;
    const cdp_ MM = (cdp_)(unsigned long long)mean;
    const cu64p_ NKs = (cu64p_)(unsigned long long)Nk;
    const cfp_ PFs = (cfp_)(unsigned long long)Pf;
    const double margin = prm[0], eps2 = prm[1];
    const bool centres_finite = prm[2] == 0.0;
    const bool pk_ok = prm[5] != 0.0;
    const int lane = threadIdx.x & 63;
    const int kp = (k + 255) & ~255;
#ifdef KM_PK_CONTIG
    // experiment: a block walks a CONTIGUOUS range of positions (consecutive passes stay in one cluster: its pair records stay in
    // the scalar cache) instead of striding through the whole order
    const int64_t stride = TH;
    const int64_t per_block = ((N + TH - 1) / TH + gridDim.x - 1) / gridDim.x * TH;
#else
    const int64_t stride = (int64_t)gridDim.x * TH;
#endif
    double inertia = 0.0;
    int changed = 0, pass = 0, ep = 0;
    auto position = [&](int64_t b) { const int64_t i = b + threadIdx.x; return i < N ? i : N - 1; };
    // the old labels and the loop's permutation a pass ahead: they give the next pass's reference centre and the head of its sorted
    // row (as in kmeans_assign_lds_kernel) and the addresses of its rows; the rows themselves are loaded where they are used -- six
    // waves per SIMD cover that latency, and twelve doubles in flight per lane would not fit beside the float copy of the sample
    int oln = -1, pnext = 0, a_nx = -1;
    unsigned long long nk0 = 0ull, nk1 = 0ull;
    double carn = 0.0;
    auto labels_ahead = [&](int64_t b) {
        const int64_t ii = position(b);
        oln = labels[ii];
        pnext = perm ? perm[ii] : 0;
    };
    auto reference_ahead = [&]() {
        a_nx = -1;
        int a = __builtin_amdgcn_readfirstlane(oln);
        const unsigned long long same = __ballot(oln == a);
        if (__builtin_popcountll(same) < 32) a = __builtin_amdgcn_readlane(oln, __builtin_ctzll(~same));
        if ((unsigned)a >= (unsigned)k) return;
        a_nx = a;
        const unsigned long long* row = Nk + (int64_t)a * kp;
        nk0 = row[lane];
        nk1 = row[64 + lane];
        carn = Ct[(int64_t)a * 16 + (lane & 15)];      // the reference centre's record, one double per lane: every DPP row holds it whole
    };
    // Blocks are dealt round-robin over the 8 XCDs (blockIdx % 8): the blocks of ONE XCD -- and with them the waves that share a
    // scalar cache -- take neighbouring positions of the sorted order, i.e. the same few clusters' pair records.
    const int64_t vblock = (gridDim.x & 7) == 0 ? (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : (int64_t)blockIdx.x;
#ifdef KM_PK_CONTIG
    int64_t base = vblock * per_block;
    const int64_t Nend = base + per_block < N ? base + per_block : N;
#else
    int64_t base = vblock * TH;
    const int64_t Nend = N;
#endif
    if (base < Nend) {
        labels_ahead(base);
        reference_ahead();
    }
    // the exact score of one centre for this lane's sample: the full scan's chain on a record gathered from the packed table
    auto score64 = [&](const double (&x)[NS], int c) {
        const double* r = Ct + (int64_t)c * 16;
        double sc = fma(x[0], r[0], -r[NS]);
#pragma unroll
        for (int j = 1; j < NS; ++j) sc = fma(x[j], r[j], sc);
        return sc;
    };
#if KM_PROFILE
    unsigned long long t_acc[16] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    unsigned long long tp_ = __builtin_readcyclecounter();
#endif
    for (; base < Nend; base += stride) {
        const int64_t i = base + threadIdx.x;
        const bool live = i < N;
        const double* xrow = X + (perm ? (int64_t)pnext : position(base)) * xstride;
        double x[NS], x2 = 0.0;
#pragma unroll
        for (int j = 0; j < NS; ++j) { x[j] = xrow[j] - (mean ? MM[j] : 0.0); x2 = fma(x[j], x[j], x2); }
        const int ol = oln, a = a_nx;
        const unsigned long long key0 = nk0, key1 = nk1;
        const double car = carn;
#if KM_PROFILE
        asm volatile("; x ready" :: "v"(x2));
        { const unsigned long long now_ = __builtin_readcyclecounter(); t_acc[0] += now_ - tp_; tp_ = now_; }
#endif
        if (base + stride < Nend) labels_ahead(base + stride);
        double best = -1.0e300;
        int bi = 0;
        bool done = false;
#if KM_PROFILE
        bool certified_ = false;
#endif
        const bool fast = pk_ok && a >= 0 && centres_finite && __ballot(!(x2 - x2 == 0.0)) == 0ull;
        int cnt = 0;
        if (fast) {
            // ---- the wave's radius about its reference centre -> the prefix of the sorted row (single-reference filter)
            // (the record came with the labels, a pass ahead, one double per lane; the centre's coordinates reach the arithmetic
            // through DPP row broadcasts as in kmeans_assign_lds_kernel -- a scalar load here was a round trip on the critical path)
            double sa;
            double yd[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) yd[j] = x[j];
            if constexpr (NS == 12) {
                sa = score_bcast(car, x);
                sub_bcast(car, yd);
            } else {
                Cen ca;
#pragma unroll
                for (int j = 0; j < 16; ++j) ca.v[j] = T[a].v[j];
                sa = fma(x[0], ca.v[0], -ca.v[NS]);
#pragma unroll
                for (int j = 1; j < NS; ++j) sa = fma(x[j], ca.v[j], sa);
#pragma unroll
                for (int j = 0; j < NS; ++j) yd[j] = x[j] - ca.v[j];
            }
            const double d2 = fma(-2.0, sa, x2);
            float rf = (float)fmax(d2, 0.0);
            rf = rf * 1.0000005f + 1.0e-37f;
            const unsigned rb = wave_max_u32(__float_as_uint(rf));
            const double u2 = (double)__uint_as_float(rb) + eps2;
            const double t2 = fma(4.004, u2, 1001.0 * margin * margin);
            const unsigned tb = __float_as_uint(fminf((float)(t2 * 1.0000001) + 1.0e-37f, 3.4028234e38f));
            cnt = __builtin_popcountll(__ballot((unsigned)(key0 >> 16) < tb)) + __builtin_popcountll(__ballot((unsigned)(key1 >> 16) < tb));
            for (int off = 128; cnt == off && off < kp; off += 128) {      // a wide wave: the next 128 keys of the row, on demand
                const unsigned long long* row = Nk + (int64_t)a * kp + off;
                cnt += __builtin_popcountll(__ballot((unsigned)(row[lane] >> 16) < tb)) + __builtin_popcountll(__ballot((unsigned)(row[64 + lane] >> 16) < tb));
            }
            // ---- packed fp32 screening: pairs of the prefix through scalar registers
            // the sample in the frame of the reference centre; two float scores further apart than mf are ordered like the exact
            // ones: a float score errs by <= 16 * 2^-24 * (|d|^2 / 2 + |y| |d|) <= 16 * 2^-24 * 4.01 u^2 (|y| <= u, |d| < 2 u + margin)
            v2f xx[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) { const float f = (float)yd[j]; xx[j][0] = f; xx[j][1] = f; }
            const float mf = (float)(8.0e-6 * u2) * 1.001f + 1.0e-37f;
            float fb = -3.0e38f, fs = -3.0e38f;
            int bp = 0;
            const int npairs = (cnt + 1) >> 1;
            const cfp_ rows = PFs + (int64_t)a * (kp >> 1) * 32;
            // The record of the NEXT pair is requested before this one is worked on: scalar loads come back in any order and can only
            // be waited for all together, so a record asked for where it is used stalls the wave for the whole round trip.  The
            // compiler cannot be told that a load is in flight, so requests and waits are asm statements tied to the record's
            // registers (the wait "rewrites" them: nothing may read them before it), with scheduling barriers between the four phases.
#if KM_PROFILE
            asm volatile("; loop starts" :: "v"(xx[0]), "s"(npairs));
            { const unsigned long long now_ = __builtin_readcyclecounter(); t_acc[1] += now_ - tp_; tp_ = now_; }
#endif
            PkRec recA, recB;
            pk_issue(recA, rows);
            pk_wait(recA);
            const int npairs2 = (npairs + 1) & ~1;             // (a pair past the prefix is a real candidate or the row's padding: harmless)
#pragma unroll 1
            for (int t = 0; t < npairs2; t += 2) {
                pk_issue(recB, rows + (t + 1) * 32);
                __builtin_amdgcn_sched_barrier(0);
                pk_pair<NS>(recA, xx, fb, fs, bp, t);
                __builtin_amdgcn_sched_barrier(0);
                pk_wait(recB);
                pk_issue(recA, rows + (t + 2 < (kp >> 1) ? t + 2 : t) * 32);
                __builtin_amdgcn_sched_barrier(0);
                pk_pair<NS>(recB, xx, fb, fs, bp, t + 1);
                __builtin_amdgcn_sched_barrier(0);
                pk_wait(recA);
            }
#if KM_PROFILE
            asm volatile("; loop done" :: "v"(fb), "v"(fs));
            { const unsigned long long now_ = __builtin_readcyclecounter(); t_acc[2] += now_ - tp_; tp_ = now_; }
#endif
            if (__ballot(!(fb - fs > mf)) == 0ull) {
                // ---- certified: the exact argmax is one of the two members of pair bp
                // the two centres of pair bp: keys 2 bp and 2 bp + 1 of the row -- lane l holds keys l and 64 + l of its head (no trip
                // to memory for them: a cross-lane read); beyond the head (a wide wave) they are fetched
                int i0, i1;
                {
                    const int q0 = 2 * bp, q1 = 2 * bp + 1;
                    const int lo0 = (int)((unsigned)key0 & 0xFFFFu), lo1 = (int)((unsigned)key1 & 0xFFFFu);
                    const int a0 = __builtin_amdgcn_ds_bpermute((q0 & 63) << 2, lo0), b0 = __builtin_amdgcn_ds_bpermute((q0 & 63) << 2, lo1);
                    const int a1 = __builtin_amdgcn_ds_bpermute((q1 & 63) << 2, lo0), b1 = __builtin_amdgcn_ds_bpermute((q1 & 63) << 2, lo1);
                    i0 = q0 < 64 ? a0 : b0;
                    i1 = q1 < 64 ? a1 : b1;
                    if (__ballot(q1 >= 128) != 0ull) {
                        if (q1 >= 128) {
                            const unsigned long long* kr = Nk + (int64_t)a * kp + q0;
                            i0 = (int)(kr[0] & 0xFFFFull);
                            i1 = (int)(kr[1] & 0xFFFFull);
                        }
                    }
                }
                const double s0 = i0 < k ? score64(x, i0) : -1.0e300;          // (an index >= k is the row's padding)
                const double s1 = i1 < k ? score64(x, i1) : -1.0e300;
                const bool second = s1 > s0 || (s1 == s0 && i1 < i0);
                best = second ? s1 : s0;
                bi = second ? i1 : i0;
                done = true;
#if KM_PROFILE
                certified_ = true;
#endif
            }
        }
        if (!done) {
            if (fast) {
                // ---- a near-tie somewhere in the wave: the same prefix in fp64, first maximum by index
#pragma unroll 1
                for (int t = 0; t < cnt; ++t) {
                    const int c = (int)(NKs[(int64_t)a * kp + t] & 0xFFFFull);
                    Cen r;
#pragma unroll
                    for (int j = 0; j < 16; ++j) r.v[j] = T[c].v[j];
                    double sc = fma(x[0], r.v[0], -r.v[NS]);
#pragma unroll
                    for (int j = 1; j < NS; ++j) sc = fma(x[j], r.v[j], sc);
                    const bool take = sc > best || (sc == best && c < bi);
                    best = take ? sc : best;
                    bi = take ? c : bi;
                }
            } else {
                // ---- no reference (first E-step), a non-finite sample or centre: the full scan
#pragma unroll 1
                for (int c = 0; c < k; ++c) {
                    Cen r;
#pragma unroll
                    for (int j = 0; j < 16; ++j) r.v[j] = T[c].v[j];
                    double sc = fma(x[0], r.v[0], -r.v[NS]);
#pragma unroll
                    for (int j = 1; j < NS; ++j) sc = fma(x[j], r.v[j], sc);
                    bi = (sc <= best) ? bi : c;
                    asm("v_max_f64 %0, %1, %2" : "=v"(best) : "v"(best), "v"(sc));
                }
            }
        }
#if KM_PROFILE
        asm volatile("; exact done" :: "v"(best), "v"(bi));
        { const unsigned long long now_ = __builtin_readcyclecounter(); t_acc[3] += now_ - tp_; tp_ = now_; }
        t_acc[7] += 1ull;
        if (fast) { t_acc[8] += 1ull; t_acc[9] += (unsigned long long)cnt; if (!certified_) t_acc[10] += 1ull; }
        else t_acc[13] += 1ull;
#endif
        if (base + stride < Nend) reference_ahead();
        if (live) {
            if (ol != bi) ++changed;
            labels[i] = bi;
            const double dmin2 = fma(-2.0, best, x2);
            if (d2out) d2out[i] = (float)dmin2;
            inertia += dmin2;
        }
        // member sums (see kmeans_assign_lds_kernel)
        const bool bad = !(x2 - x2 == 0.0);
        const bool ok = live && !bad;
        bool sent = false;
        {
            const unsigned long long okm = __ballot(ok);
            if (okm != 0ull) {
                const int lab = __builtin_amdgcn_readlane(bi, __builtin_ctzll(okm));
                const bool mine = ok && bi == lab;
                const int cntm = __builtin_popcountll(__ballot(mine));
                if (cntm >= KM2_SUM_MIN) {
                    unsigned qlo[NS], qhi[NS];
#pragma unroll
                    for (int j = 0; j < NS; ++j) { const u64 q = mine ? km_fix(x[j], FS[j]) : 0ull; qlo[j] = (unsigned)q; qhi[j] = (unsigned)(q >> 32); }
                    wave_sum_u64x(qlo, qhi);
                    if (lane == 63) {
                        u64* sp = sums + lab * np1;
#pragma unroll
                        for (int j = 0; j < NS; ++j) atomicAdd(&sp[j], ((u64)qhi[j] << 32) | qlo[j]);
                        atomicAdd(&sp[n], (u64)cntm);
                    }
                    sent = mine;
                }
            }
        }
        if (live && !sent) {
            u64* sp = sums + bi * np1;
#pragma unroll
            for (int j = 0; j < NS; ++j) atomicAdd(&sp[j], km_fix(bad ? 0.0 : x[j], FS[j]));
            atomicAdd(&sp[n], bad ? 1ull + KM_POISON : 1ull);
        }
#if KM_PROFILE
        { const unsigned long long now_ = __builtin_readcyclecounter(); t_acc[4] += now_ - tp_; tp_ = now_; }
#endif
        if (++pass == KM_EPOCH_PASSES && base + stride < Nend) { km_flush(sums, partial, ep, k, n, true); ++ep; pass = 0; }
    }
#if KM_PROFILE
    if (lane == 0) for (int q = 0; q < 16; ++q) atomicAdd(&km_prof[q], t_acc[q]);
#endif
    for (int off = 32; off > 0; off >>= 1) {
        inertia += __shfl_down(inertia, off);
        changed += __shfl_down(changed, off);
    }
    if (lane == 0) { sh_inertia[threadIdx.x >> 6] = inertia; sh_changed[threadIdx.x >> 6] = changed; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double in = 0.0;
        int ch = 0;
        for (int q = 0; q < TH / 64; ++q) { in += sh_inertia[q]; ch += sh_changed[q]; }
        block_inertia[blockIdx.x] = in;
        block_changed[blockIdx.x] = ch;
    }
    km_flush(sums, partial, ep, k, n, false);
    km_zero_epochs(partial, ep + 1, nepochs, k, n);
}

#if KM_PROFILE
}  // namespace brov
extern "C" __attribute__((visibility("default"))) int brov_debug_kmprof(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(brov::km_prof), 128) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(brov::km_prof), z, 128) != hipSuccess) return -1; }
    return 0;
}
namespace brov {
#endif
#if KM_BLOCKTIME
}  // namespace brov
extern "C" __attribute__((visibility("default"))) int brov_debug_kmblk(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(brov::km_blk), 64) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(brov::km_blk), z, 64) != hipSuccess) return -1; }
    return 0;
}
namespace brov {
#endif
// ---- M-step --------------------------------------------------------------------------------------------------------------------
// Totals of the member sums: 128-bit integers kept as two int64 limbs (hi, lo), value = hi * 2^42 + lo with 0 <= lo < 2^42 --
// limbs that an int64 SUM all-reduce over ranks adds without overflow (|total| < 2^(48 + 40)), after which the pair still
// stands for the exact total (lo may then exceed 2^42; km_load128 does not care).
constexpr int KM_LIMB = 42;
__device__ __forceinline__ __int128 km_load128(const long long* p) { return ((__int128)p[0] << KM_LIMB) + (__int128)p[1]; }
__device__ __forceinline__ void km_store128(long long* p, __int128 v) {
    p[0] = (long long)(v >> KM_LIMB);
    p[1] = (long long)(v & (((__int128)1 << KM_LIMB) - 1));
}
__device__ __forceinline__ double km_to_double(__int128 v) {
    // sign and magnitude (two's-complement halves would cancel: -5 = -2^64 + (2^64 - 5), and the second term rounds to 2^64)
    const bool neg = v < 0;
    const unsigned __int128 m = neg ? (unsigned __int128)(-v) : (unsigned __int128)v;
    const double d = (double)(u64)(m >> 64) * 18446744073709551616.0 + (double)(u64)m;      // one rounding below 2^64, two above: the same bits every time
    return neg ? -d : d;
}
// packed table row of a centre: [coordinates | |c|^2 / 2 at slot n | zeros | -|c|^2 / 2 at slot 15 when n <= 14]; the norm in
// coordinate order with FMAs -- the one formula of kmeans_c2_kernel and kmeans_average_kernel
__device__ __forceinline__ double km_pack_centre(int n, const double* __restrict__ c, double* __restrict__ row) {
    double q = 0.0;
    for (int j = 0; j < n; ++j) q = fma(c[j], c[j], q);
    for (int j = 0; j < 16; ++j) row[j] = j < n ? c[j] : (j == n ? 0.5 * q : ((j == 15 && n <= KM2_NMAX) ? -0.5 * q : 0.0));
    return q;
}

// (the sums of the M-step -- one 256-thread block per centre over the partials of all blocks and epochs -- are phase 1 of
// kmeans_mstep_kernel below)
constexpr int KM_BND_TOP = 4;                          // movers that kmeans_bounds_kernel takes apart (a float4 of centre distances per cluster)
constexpr int KM_BND_TAIL = 2 * KM_BND_TOP + 4;
// New centres from the totals, one block of 1024 threads (thread = centre): mean = total / s_j / count (scikit-learn multiplies by
// 1 / count: `_average_centers`), NaN for a poisoned cluster; stats[0] = sum of squared centre shifts (fixed tree), stats[2] =
// changed labels, stats[3] = empty clusters; the packed table Ct; prm[2] != 0 when a centre is not finite.
//   mode 0 (every iteration): an empty cluster keeps its old centre for now, and if there is one, prm[3] = 1 puts the E-step that
//           is already queued on hold -- the host relocates (kmeans_relocate_kernel) and calls mode 1;
//   mode 1 (after the relocation, or when there was nothing to relocate to): a cluster that is (still) empty takes the new centre
//           of the cluster with the largest count (the first of them: np.argmax), as `_average_centers` does; the hold is lifted.
__global__ void __launch_bounds__(1024) kmeans_average_kernel(int n, int k, const long long* __restrict__ red, const double* __restrict__ fix,
                                                              const double* __restrict__ Cold, double* __restrict__ Cnew, double* __restrict__ Ct,
                                                              double* __restrict__ stats, double* __restrict__ prm, int mode,
                                                              float* __restrict__ shiftc /* [k + KM_BND_TAIL] or nullptr: |c_new - c_old| of every centre, rounded UP; then the
                                                                 KM_BND_TOP largest, their centres' indices, and the largest of the rest (distance bounds) */,
                                                              int* __restrict__ nlist /* or nullptr: the counter of kmeans_bounds_kernel's list, zeroed here for the next one */) {
    const int np1 = n + 1;
    __shared__ double sh_d[16];
    __shared__ long long sh_cnt[16];
    __shared__ int sh_idx[16], sh_emp[16], sh_bad[16];
    __shared__ int s_arg;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // pass 1: the non-empty clusters; count of the empty ones; the biggest cluster
    long long bestc = -1;
    int besti = 0x7fffffff, empties = 0;
    for (int c = threadIdx.x; c < k; c += 1024) {
        const __int128 cw = km_load128(red + ((int64_t)c * np1 + n) * 2);
        const long long cnt = (long long)(cw & (KM_POISON - 1));
        const bool poisoned = (cw >> 40) != 0;
        if (cnt > 0 || poisoned) {
            for (int j = 0; j < n; ++j) {
                const double tot = km_to_double(km_load128(red + ((int64_t)c * np1 + j) * 2));
                Cnew[(int64_t)c * n + j] = poisoned ? __builtin_nan("") : (tot * fix[16 + j]) / (double)cnt;
            }
        } else {
            ++empties;
            for (int j = 0; j < n; ++j) Cnew[(int64_t)c * n + j] = Cold[(int64_t)c * n + j];
        }
        if (cnt > bestc) { bestc = cnt; besti = c; }       // c ascends within a thread: the first maximum stays
    }
    for (int off = 32; off > 0; off >>= 1) {
        const long long oc = __shfl_down(bestc, off);
        const int oi = __shfl_down(besti, off);
        if (oc > bestc || (oc == bestc && oi < besti)) { bestc = oc; besti = oi; }
        empties += __shfl_down(empties, off);
    }
    if (lane == 0) { sh_cnt[w] = bestc; sh_idx[w] = besti; sh_emp[w] = empties; }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long bc = -1;
        int bi = 0x7fffffff, em = 0;
        for (int q = 0; q < 16; ++q) {
            if (sh_cnt[q] > bc || (sh_cnt[q] == bc && sh_idx[q] < bi)) { bc = sh_cnt[q]; bi = sh_idx[q]; }
            em += sh_emp[q];
        }
        s_arg = bi;
        stats[3] = (double)em;
        stats[2] = (double)red[(int64_t)k * np1 * 2];
        prm[3] = (mode == 0 && em > 0) ? 1.0 : 0.0;
    }
    __syncthreads();
    // pass 2 (mode 1): clusters that are still empty move to the biggest cluster's new centre
    if (mode == 1) {
        const int arg = s_arg;
        for (int c = threadIdx.x; c < k; c += 1024) {
            const __int128 cw = km_load128(red + ((int64_t)c * np1 + n) * 2);
            if ((long long)(cw & (KM_POISON - 1)) <= 0 && (cw >> 40) == 0)
                // `_average_centers` walks the clusters in index order IN PLACE: an empty cluster behind the biggest one copies its
                // mean, one in front of it copies the row before it was scaled -- the SUM of the biggest cluster's members
                // (scikit-learn 1.7.2, _k_means_common.pyx: `centers[j, k] = centers[argmax_weight, k]` inside the averaging loop)
                for (int j = 0; j < n; ++j)
                    Cnew[(int64_t)c * n + j] = c > arg ? Cnew[(int64_t)arg * n + j]     // (arg is not empty: nobody writes its row here)
                                                       : km_to_double(km_load128(red + ((int64_t)arg * np1 + j) * 2)) * fix[16 + j];
        }
    }
    // pass 3: shifts, packed table, finiteness
    double shift2 = 0.0;
    int bad = 0;
    bool snan = false;
    for (int c = threadIdx.x; c < k; c += 1024) {
        double cc[KM_NMAX];
        for (int j = 0; j < KM_NMAX; ++j) cc[j] = j < n ? Cnew[(int64_t)c * n + j] : 0.0;
        double sh = 0.0;
        for (int j = 0; j < n; ++j) { const double dd = cc[j] - Cold[(int64_t)c * n + j]; sh = fma(dd, dd, sh); }
        shift2 += sh;
        if (shiftc) {
            // >= the true shift: sh carries n roundings of 2^-53; NaN (a poisoned centre) stays NaN and fails every bound test
            const float f = (float)sqrt(sh) * 1.000001f + 1.0e-37f;
            shiftc[c] = f;
            if (!(f == f)) snan = true;
        }
        const double q = km_pack_centre(n, cc, Ct + (int64_t)c * 16);
        if (!(q - q == 0.0)) bad = 1;
    }
    for (int off = 32; off > 0; off >>= 1) { shift2 += __shfl_down(shift2, off); bad |= __shfl_down(bad, off); }
    if (lane == 0) { sh_d[w] = shift2; sh_bad[w] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        int b = 0;
        for (int q = 0; q < 16; ++q) { t += sh_d[q]; b |= sh_bad[q]; }
        stats[0] = t;
        prm[2] = b ? 1.0 : 0.0;      // a non-finite centre (NaN / inf data): the candidate filter stands down
        if (nlist) { nlist[0] = 0; nlist[KM_NL_TICKET] = 0; nlist[KM_NL_FRONT] = 0; nlist[KM_NL_BACK] = 0; }
    }
    if (shiftc) {
        // the KM_BND_TOP largest shifts, whose centres they are, and the largest of the rest (kmeans_bounds_kernel): wave 0 picks them one
        // after the other from the values the block has just written (equal values: the lower index).  A NaN shift poisons them all.
        __shared__ int sh_nan[16];
        const unsigned long long nanm = __ballot(snan);
        if (lane == 0) sh_nan[w] = nanm != 0ull;
        __syncthreads();                               // (shiftc[0 .. k) is this block's own: visible behind the barrier)
        if (w == 0) {
            int an = 0;
            for (int q = 0; q < 16; ++q) an |= sh_nan[q];
            int picked[KM_BND_TOP + 1];
            float pv[KM_BND_TOP + 1];
#pragma unroll
            for (int r = 0; r <= KM_BND_TOP; ++r) {
                float bv = -1.0f;
                int bi = 0x7fffffff;
                for (int c = lane; c < k; c += 64) {
                    bool taken = false;
#pragma unroll
                    for (int q = 0; q < r; ++q) taken = taken || picked[q] == c;
                    const float v = shiftc[c];
                    if (!taken && v > bv) { bv = v; bi = c; }      // c ascends within a lane: the first of equal values stays
                }
                for (int off = 32; off > 0; off >>= 1) {
                    const float ov = __shfl_down(bv, off);
                    const int oi = __shfl_down(bi, off);
                    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
                }
                picked[r] = __builtin_amdgcn_readfirstlane(bi);
                pv[r] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(bv)));
                if (picked[r] == 0x7fffffff) { picked[r] = -1; pv[r] = 0.0f; }      // fewer centres than picks
            }
            if (lane == 0) {
                const float nanf_ = __builtin_nanf("");
#pragma unroll
                for (int r = 0; r < KM_BND_TOP; ++r) {
                    shiftc[k + r] = an ? nanf_ : pv[r];
                    shiftc[k + KM_BND_TOP + r] = __int_as_float(picked[r]);
                }
                shiftc[k + 2 * KM_BND_TOP] = an ? nanf_ : pv[KM_BND_TOP];
            }
        }
    }
}

// ---- the M-step as two launches -- k blocks, then one (round 5): the partial tables' sums, the new centre of every cluster in the block that has just
// summed it, and kmeans_average_kernel's global tail in the block that finishes last.  Per iteration the loop used to queue reduce (9 us)
// -> average (29 us: one block, a thread per centre walking its thirteen totals) -> a 32-byte copy to the host -> centre distances ->
// bounds -> E-step, ~100 us of launch boundaries and short serial kernels around 400 us of work.  Here: phase 1 (the block of centre c)
// = the 128-bit totals (any grouping of integer additions gives the same bits); phase 2 (same block) = mean, shift, packed table row -- the
// arithmetic of kmeans_average_kernel, pass 1 and 3, for one centre; tail (a second launch of ONE block) = the sum
// of the squared shifts IN THE ORDER kmeans_average_kernel adds them (the stopping rule compares it with the tolerance: a sharded run,
// which still needs its all-reduce between the two phases and therefore launches this kernel twice, must stop at the same iteration),
// the count of empty clusters, the movers of the distance bounds and every centre's distance to them (what kmeans_cdist_kernel used
// to add), and the four statistics the host waits for -- stored straight into its pinned, device-mapped block (no copy in the stream).
// phases: 1 = sums only (sharded: all-reduce `red` next), 2 = centres + tail from `red`, 3 = both.
//
// Phase 1 in detail: one 256-thread block per centre adds up the partials of all blocks and epochs (thread = (slot j, sub-range of the partials),
// eight loads in flight) as 128-bit integers -- any grouping gives the same total -- and leaves them in red [k][n+1][2].
// The second wave of block 0 sums the E-step's per-block statistics: inertia (fp64, lane l takes blocks l, l + 64, ...; fixed
// tree: the same bits for the same launch geometry) into stats[1], changed labels into the tail of red (an integer, so that it
// takes part in the all-reduce of a sharded run).
// `tot` (round 4, distance bounds): this rank's own totals, kept from iteration to iteration.  delta = 0: the partials are the sums
// over ALL samples (tot = their total); delta = 1: the partials are the CHANGES of an E-step that visited only the samples whose
// bounds failed -- what a sample that moved took from its old cluster and brought to its new one, integers like the sums themselves
// -- and tot += their total: the same 128-bit integers as a fresh summation, bit for bit.  red = tot either way.
struct KmMstep {
    // sums
    int nparts, nblocks, n, k;
    const u64* partial;
    const double* block_inertia;
    const int* block_changed;
    long long* red;
    long long* tot;
    int delta;
    const int* nlist_in;
    int span;
    // centres
    const double* fix;
    const double* Cold;
    double* Cnew;
    double* Ct;
    double* stats;
    double* prm;
    float* shiftc;                  // [k + KM_BND_TAIL] or nullptr
    float* mvd;                     // [k][KM_BND_TOP] or nullptr
    int* nlist;                     // zeroed for the next kmeans_bounds_kernel, or nullptr
    double* shift2;                 // [k] scratch: squared shift of every centre
    int* flags;                     // [k] scratch: bit 0 empty, bit 1 not finite
    unsigned* ticket;               // device-wide arrival counter (zero between launches)
    double* hstats;                 // pinned, device-mapped [5], or nullptr
    double seq;                     // what the tail stores behind the four statistics: the host waits for THIS M-step by polling it
};
// ---- the iteration's global part: a launch of its own, one block (phases = 4).  (First built as the tail of the same launch, run by
// the last block to take a device-wide ticket: 61 us per M-step against 38 for the two kernels of round 4 -- a release / acquire
// between blocks on different XCDs means L2 write-backs and invalidations, which a kernel boundary does once.)
// (round 5: where the centre distances are built next, the block runs as one more block of kmeans_cdist_kernel -- a launch of 12 us less
// per iteration; `tail_deferred` of launch_kmeans_mstep / `tail` of launch_kmeans_cdist)
__device__ __forceinline__ void km_mstep_tail(const KmMstep& a) {
    const int n = a.n, k = a.k, np1 = n + 1;
    __shared__ double sh_d[16];
    __shared__ int sh_em[4], sh_bad[4], sh_nan[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    {
        // the squared shifts in kmeans_average_kernel's order: its thread v (of 1024) adds its centres v, v + 1024, ...; a shuffle tree
        // over each of its 16 waves; the 16 wave sums one after the other
        for (int vw = w; vw < 16; vw += 4) {
            double v = 0.0;
            for (int cc = 64 * vw + lane; cc < k; cc += 1024) v += a.shift2[cc];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0) sh_d[vw] = v;
        }
        int em = 0, bad = 0, snan = 0;
        for (int cc = threadIdx.x; cc < k; cc += 256) {
            const int f = a.flags[cc];
            em += f & 1;
            bad |= (f >> 1) & 1;
            if (a.shiftc) { const float sv = a.shiftc[cc]; snan |= !(sv == sv); }
        }
        for (int off = 32; off > 0; off >>= 1) { em += __shfl_down(em, off); bad |= __shfl_down(bad, off); snan |= __shfl_down(snan, off); }
        if (lane == 0) { sh_em[w] = em; sh_bad[w] = bad; sh_nan[w] = snan; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tsum = 0.0;
        for (int q = 0; q < 16; ++q) tsum += sh_d[q];
        const int em = sh_em[0] + sh_em[1] + sh_em[2] + sh_em[3];
        const int bad = sh_bad[0] | sh_bad[1] | sh_bad[2] | sh_bad[3];
        a.stats[0] = tsum;
        a.stats[2] = (double)a.red[(int64_t)k * np1 * 2];
        a.stats[3] = (double)em;
        a.prm[3] = em > 0 ? 1.0 : 0.0;                  // hold: the queued E-step returns at once, the host relocates
        a.prm[2] = bad ? 1.0 : 0.0;                     // a non-finite centre (NaN / inf data): the candidate filter stands down
        if (a.nlist) { a.nlist[0] = 0; a.nlist[KM_NL_TICKET] = 0; a.nlist[KM_NL_FRONT] = 0; a.nlist[KM_NL_BACK] = 0; }
        if (a.hstats) {
            a.hstats[0] = tsum; a.hstats[1] = a.stats[1]; a.hstats[2] = a.stats[2]; a.hstats[3] = (double)em;
            __threadfence_system();
            __hip_atomic_store(a.hstats + 4, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);      // (no event in the stream: an event record cost the next kernel 6 us)
        }
    }
    if (a.shiftc) {
        // the KM_BND_TOP largest shifts, whose centres they are, and the largest of the rest (kmeans_bounds_kernel): wave 0 picks them one
        // after the other (equal values: the lower index).  A NaN shift poisons them all.
        __shared__ int s_pick[KM_BND_TOP];
        if (w == 0) {
            const int an = sh_nan[0] | sh_nan[1] | sh_nan[2] | sh_nan[3];
            int picked[KM_BND_TOP + 1];
            float pv[KM_BND_TOP + 1];
#pragma unroll
            for (int r = 0; r <= KM_BND_TOP; ++r) {
                float bv = -1.0f;
                int bi = 0x7fffffff;
                for (int cc = lane; cc < k; cc += 64) {
                    bool taken = false;
#pragma unroll
                    for (int q = 0; q < r; ++q) taken = taken || picked[q] == cc;
                    const float v = a.shiftc[cc];
                    if (!taken && v > bv) { bv = v; bi = cc; }      // cc ascends within a lane: the first of equal values stays
                }
                for (int off = 32; off > 0; off >>= 1) {
                    const float ov = __shfl_down(bv, off);
                    const int oi = __shfl_down(bi, off);
                    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
                }
                picked[r] = __builtin_amdgcn_readfirstlane(bi);
                pv[r] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(bv)));
                if (picked[r] == 0x7fffffff) { picked[r] = -1; pv[r] = 0.0f; }      // fewer centres than picks
            }
            if (lane == 0) {
                const float nanf_ = __builtin_nanf("");
#pragma unroll
                for (int r = 0; r < KM_BND_TOP; ++r) {
                    a.shiftc[k + r] = an ? nanf_ : pv[r];
                    a.shiftc[k + KM_BND_TOP + r] = __int_as_float(picked[r]);
                    s_pick[r] = picked[r];
                }
                a.shiftc[k + 2 * KM_BND_TOP] = an ? nanf_ : pv[KM_BND_TOP];
            }
        }
        __syncthreads();
        if (a.mvd) {
            // every centre's distance to the movers, rounded DOWN (kmeans_bounds_kernel: a mover far from a sample's own centre NOW): the
            // movers' rows in the LDS, a thread per centre with its own row in registers
            __shared__ double s_mv[KM_BND_TOP][KM_NMAX];
            if (threadIdx.x < KM_BND_TOP * KM_NMAX) {
                const int r = threadIdx.x / KM_NMAX, jj = threadIdx.x % KM_NMAX;
                const int cm = s_pick[r];
                s_mv[r][jj] = ((unsigned)cm < (unsigned)k && jj < n) ? a.Ct[cm * 16 + jj] : 0.0;
            }
            __syncthreads();
            for (int ca = threadIdx.x; ca < k; ca += 256) {
                double row[KM_NMAX];
#pragma unroll
                for (int jj = 0; jj < KM_NMAX; ++jj) row[jj] = jj < n ? a.Ct[ca * 16 + jj] : 0.0;
                float out[KM_BND_TOP];
#pragma unroll
                for (int r = 0; r < KM_BND_TOP; ++r) {
                    float v = __builtin_inff();         // no such mover: it constrains nothing
                    if ((unsigned)s_pick[r] < (unsigned)k) {
                        double q = 0.0;
#pragma unroll
                        for (int jj = 0; jj < KM_NMAX; ++jj)
                            if (jj < n) { const double d = row[jj] - s_mv[r][jj]; q = fma(d, d, q); }
                        v = (float)(sqrt(q) * 0.999999);
                        v = v > 0.0f ? v * 0.9999999f : v;
                    }
                    out[r] = v;
                }
                *reinterpret_cast<float4*>(a.mvd + ca * KM_BND_TOP) = make_float4(out[0], out[1], out[2], out[3]);
            }
        }
    }
}
__global__ void __launch_bounds__(256) kmeans_mstep_kernel(KmMstep a, int phases) {
    const int n = a.n, k = a.k, np1 = n + 1;
    const int c = blockIdx.x;
    const int j = threadIdx.x & 15, sr = threadIdx.x >> 4;
    __shared__ long long part[16][17][2];
    __shared__ double s_cc[KM_NMAX];
    __shared__ long long s_cw[2];
    __int128 t = 0;
    if (phases == 4) goto tail;
    if (phases & 1) {
        int nparts = a.nparts;
        if (a.nlist_in) {
            // a list-form E-step (span = its blocks x threads) fills and zeroes only the epochs its longest-running block reaches
            const int64_t passes = ((int64_t)a.nlist_in[0] + a.nlist_in[KM_NL_FRONT] + a.nlist_in[KM_NL_BACK] + a.span - 1) / a.span;
            const int64_t used = (passes + KM_EPOCH_PASSES - 1) / KM_EPOCH_PASSES;
            const int tables = a.nblocks * (int)(used > 1 ? used : 1);
            nparts = tables < nparts ? tables : nparts;
        }
        __int128 s = 0;
        if (j <= n) {
            for (int b0 = sr; b0 < nparts; b0 += 16 * 8) {
                long long v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { const int b = b0 + 16 * q; v[q] = b < nparts ? (long long)a.partial[((int64_t)b * k + c) * np1 + j] : 0ll; }
#pragma unroll
                for (int q = 0; q < 8; ++q) s += (__int128)v[q];
            }
        }
        part[sr][j][0] = (long long)(s >> 64);
        part[sr][j][1] = (long long)(u64)s;
        __syncthreads();
        if ((int)threadIdx.x <= n) {
            for (int q = 0; q < 16; ++q) t += ((__int128)part[q][threadIdx.x][0] << 64) + (__int128)(u64)part[q][threadIdx.x][1];
            if (a.tot) {
                if (a.delta) t += km_load128(a.tot + ((int64_t)c * np1 + threadIdx.x) * 2);
                km_store128(a.tot + ((int64_t)c * np1 + threadIdx.x) * 2, t);
            }
            km_store128(a.red + ((int64_t)c * np1 + threadIdx.x) * 2, t);
        }
        if (blockIdx.x == 0 && (threadIdx.x >> 6) == 1) {
            const int l = threadIdx.x & 63;
            double in = 0.0;
            long long ch = 0;
            for (int b = l; b < a.nblocks; b += 64) { in += a.block_inertia[b]; ch += a.block_changed[b]; }
            for (int off = 32; off > 0; off >>= 1) {
                in += __shfl_down(in, off);
                ch += __shfl_down(ch, off);
            }
            if (l == 0) { a.stats[1] = in; a.red[(int64_t)k * np1 * 2] = ch; a.red[(int64_t)k * np1 * 2 + 1] = 0; }
        }
    }
    if (!(phases & 2)) return;
    // ---- this centre: kmeans_average_kernel's pass 1 and pass 3 (mode 0)
    if (!(phases & 1) && (int)threadIdx.x <= n) t = km_load128(a.red + ((int64_t)c * np1 + threadIdx.x) * 2);     // (sharded: the all-reduced totals)
    if ((int)threadIdx.x == n) { s_cw[0] = (long long)(t >> 64); s_cw[1] = (long long)(u64)t; }
    __syncthreads();
    {
        const __int128 cw = ((__int128)s_cw[0] << 64) + (__int128)(u64)s_cw[1];
        const long long cnt = (long long)(cw & (KM_POISON - 1));
        const bool poisoned = (cw >> 40) != 0;
        const bool empty = !(cnt > 0 || poisoned);
        if ((int)threadIdx.x < n) {
            const int jj = threadIdx.x;
            double v;
            if (!empty) v = poisoned ? __builtin_nan("") : (km_to_double(t) * a.fix[16 + jj]) / (double)cnt;
            else v = a.Cold[(int64_t)c * n + jj];          // an empty cluster keeps its old centre for now (the host relocates)
            a.Cnew[(int64_t)c * n + jj] = v;
            s_cc[jj] = v;
        } else if (threadIdx.x < KM_NMAX) {
            s_cc[threadIdx.x] = 0.0;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double cc[KM_NMAX];
            for (int q = 0; q < KM_NMAX; ++q) cc[q] = s_cc[q];
            double sh = 0.0;
            for (int q = 0; q < n; ++q) { const double dd = cc[q] - a.Cold[(int64_t)c * n + q]; sh = fma(dd, dd, sh); }
            a.shift2[c] = sh;
            // >= the true shift: sh carries n roundings of 2^-53; NaN (a poisoned centre) stays NaN and fails every bound test
            if (a.shiftc) a.shiftc[c] = (float)sqrt(sh) * 1.000001f + 1.0e-37f;
            const double qn = km_pack_centre(n, cc, a.Ct + (int64_t)c * 16);
            a.flags[c] = (empty ? 1 : 0) | (!(qn - qn == 0.0) ? 2 : 0);
        }
    }
    return;
tail:
    km_mstep_tail(a);
}
static KmMstep km_mstep_args(const KmMstepArgs& h) {
    KmMstep a;
    a.nparts = h.nparts; a.nblocks = h.nblocks; a.n = h.n; a.k = h.k;
    a.partial = h.partial; a.block_inertia = h.block_inertia; a.block_changed = h.block_changed; a.red = h.red; a.tot = h.tot;
    a.delta = h.delta; a.nlist_in = h.delta ? h.nlist : nullptr; a.span = h.nblocks * KM_THREADS;
    a.fix = h.fix; a.Cold = h.Cold; a.Cnew = h.Cnew; a.Ct = h.Ct; a.stats = h.stats; a.prm = h.prm; a.shiftc = h.shiftc; a.mvd = h.mvd;
    a.nlist = h.nlist; a.shift2 = h.scratch; a.flags = reinterpret_cast<int*>(h.scratch + h.k); a.ticket = reinterpret_cast<unsigned*>(h.scratch + h.k) + h.k;
    a.hstats = h.hstats; a.seq = h.seq;
    return a;
}
hipError_t launch_kmeans_mstep(hipStream_t st, const KmMstepArgs& h, int phases) {
    const KmMstep a = km_mstep_args(h);
    if (phases & 3) hipLaunchKernelGGL(kmeans_mstep_kernel, dim3(h.k), dim3(256), 0, st, a, phases & 3);
    if ((phases & 2) && !h.tail_deferred) hipLaunchKernelGGL(kmeans_mstep_kernel, dim3(1), dim3(256), 0, st, a, 4);
    return hipGetLastError();
}
size_t kmeans_mstep_scratch_doubles(int k) { return (size_t)k + ((size_t)k + 2) / 2 + 2; }

// ---- before the loop: the range of every (centred) coordinate -> fixed-point scales; max |x|^2 -> margins of the candidate filter ----
// rng [16] (zeroed by the caller): bit patterns of non-negative doubles, atomicMax'd -- slots 0..n-1: max_i |x_ij - mean_j| over the finite
// values, slot 15: max_i |x_i - mean|^2 over the finite rows.  (u64 maxima: an all-reduce(MAX) over ranks of a sharded run keeps the meaning.)
__global__ void __launch_bounds__(256) kmeans_range_kernel(int64_t N, int n, const double* __restrict__ X, int64_t xstride, const double* __restrict__ mean,
                                                           u64* __restrict__ rng) {
    double m[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) m[j] = 0.0;
    // (rows of an even number of doubles on 16-byte boundaries are read as 16-byte pieces: with 8-byte loads a wave's request for
    // coordinate j touches 64 rows 96 bytes apart, and the pass took 1.6 ms over 0.96 GB)
    const bool vec = (n & 1) == 0 && (xstride & 1) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        double x2 = 0.0;
        double xr[KM_CMAX + 1];
        if (vec) {
#pragma unroll
            for (int j = 0; j < KM_CMAX; j += 2)
                if (j < n) { const v2d v = *reinterpret_cast<const v2d*>(X + i * xstride + j); xr[j] = v[0]; xr[j + 1] = v[1]; }
        } else {
#pragma unroll
            for (int j = 0; j < KM_CMAX; ++j)
                if (j < n) xr[j] = X[i * xstride + j];
        }
#pragma unroll
        for (int j = 0; j < KM_CMAX; ++j) {
            if (j < n) {
                const double x = xr[j] - (mean ? mean[j] : 0.0);
                x2 = fma(x, x, x2);
                const double ax = fabs(x);
                if (ax - ax == 0.0) m[j] = fmax(m[j], ax);
            }
        }
        if (x2 - x2 == 0.0) m[15] = fmax(m[15], x2);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        double v = m[j];
        for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off));
        if ((threadIdx.x & 63) == 0 && v > 0.0) atomicMax(&rng[j], (u64)__double_as_longlong(v));
    }
}
// fix[j] = s_j = 2^(48 - e_j), 2^e_j > max |x_j| (frexp), fix[16 + j] = 1 / s_j;  prm = {margin, eps2, 0, 0} from R^2 = 2 max |x|^2
// (centres are means of samples: |c| <= max |x|)
__global__ void kmeans_scale_kernel(int n, const u64* __restrict__ rng, double* __restrict__ fix, double* __restrict__ prm) {
    const int j = threadIdx.x;
    if (j < 16) {
        const double m = __longlong_as_double((long long)rng[j]);
        int e = 0;
        if (j < n && m > 0.0) (void)frexp(m, &e);
        int sh = KM_FIX_BITS - e;
        sh = sh > 1000 ? 1000 : (sh < -1000 ? -1000 : sh);
        fix[j] = ldexp(1.0, sh);
        fix[16 + j] = ldexp(1.0, -sh);
    }
    if (j == 0) {
        const double R2 = 2.0 * __longlong_as_double((long long)rng[15]);
        prm[0] = 1.0e-6 * sqrt(R2);                       // margin
        prm[1] = 1.0e-13 * R2;                            // eps2: 30 x the rounding of a computed squared distance
        prm[2] = 0.0;
        prm[3] = 0.0;
        // packed-fp32 screening (kmeans_assign_pk_kernel): a float score errs by <= 16 * 2^-24 * (|c|^2 / 2 + |x| |c|) <= 1.43e-6 M^2,
        // M^2 = max |x|^2 = R^2 / 2; two scores further apart than marginf = 4e-6 M^2 are ordered like their exact values.  The
        // screening is used only when M^2 is far from float's range limits.
        const double M2 = 0.5 * R2;
        prm[4] = 4.0e-6 * M2 * 1.000001;
        prm[5] = (M2 > 1.0e-24 && M2 < 1.0e24) ? 1.0 : 0.0;
    }
}

// ---- empty clusters: scikit-learn's `_relocate_empty_clusters_dense` (sklearn/cluster/_k_means_common.pyx) -------------------------
// distances = ((X - centers_old[labels])**2).sum(axis=1) in NumPy's arithmetic: every square rounded, then the pairwise sum of
// np.add.reduce over a contiguous axis (n < 8: left to right; else eight running sums over the blocks of 8, combined as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the remainder left to right) -- no FMA contraction.  Written per ROW of the caller's X
// (through the loop's permutation), with the label beside it: the host picks the n_empty farthest rows with NumPy's own argpartition
// (engine.py; the C library's fallback is a plain descending selection) and moves their fixed-point coordinates between the totals (capi.hip).
__global__ void __launch_bounds__(256) kmeans_reloc_dist_kernel(int64_t N, int n, const double* __restrict__ X, int64_t xstride,
                                                                const double* __restrict__ mean, const double* __restrict__ Cold,
                                                                const int* __restrict__ labels, const int* __restrict__ perm,
                                                                double* __restrict__ dist_row, int* __restrict__ lab_row) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= N) return;
    const int64_t row = perm ? (int64_t)perm[p] : p;
    const int lab = labels[p];
    double sq[KM_NMAX];
#pragma unroll
    for (int j = 0; j < KM_NMAX; ++j) {
        sq[j] = 0.0;
        if (j < n) {
            const double xc = __dsub_rn(X[row * xstride + j], mean ? mean[j] : 0.0);
            const double d = __dsub_rn(xc, Cold[(int64_t)lab * n + j]);
            sq[j] = __dmul_rn(d, d);
        }
    }
    double res;
    if (n < 8) {
        res = 0.0;
#pragma unroll
        for (int j = 0; j < 7; ++j)
            if (j < n) res = __dadd_rn(res, sq[j]);
    } else {
        res = __dadd_rn(__dadd_rn(__dadd_rn(sq[0], sq[1]), __dadd_rn(sq[2], sq[3])), __dadd_rn(__dadd_rn(sq[4], sq[5]), __dadd_rn(sq[6], sq[7])));
#pragma unroll
        for (int j = 8; j < KM_NMAX; ++j)
            if (j < n) res = __dadd_rn(res, sq[j]);
    }
    dist_row[row] = res;
    lab_row[row] = lab;
}
// squared centre-centre distances (difference form) for the candidate filter, one block per row, rebuilt after every M-step:
// floats rounded DOWN (a candidate test that errs, errs towards evaluating), row length kp = k rounded up to 256, permuted inside
// every block of 256 so that position 4 lane + q holds centre 64 q + lane (see block_masks); padding = +inf (never a candidate).
// Round 4: the same row once more SORTED, as 64-bit keys (bits of the distance << 16 | centre index: distances ascending, equal ones
// by index) -- the candidates of a wave whose reference centre is a are a PREFIX of row a of Nk (kmeans_assign_lds_kernel, single-
// reference filter), and one 8-byte load per lane brings the distance to test and the centre to evaluate; for kp <= 1024 (the
// range of the kernels that use it).  Bitonic sort in the LDS.
#ifndef KM_WIDE_CAND
#define KM_WIDE_CAND 128                 // a pass with this many candidates counts as expensive (kmeans_bounds_kernel: such tiles first)
#endif
#ifndef KM_BND_BETA
#define KM_BND_BETA 0.06                 // the prefix of a pass that leaves bounds is cut at 2 (1 + beta) u instead of 2 u
#endif
__global__ void __launch_bounds__(256) kmeans_cdist_kernel(int n, int k, const double* __restrict__ Ct, float* __restrict__ Dc,
                                                           unsigned long long* __restrict__ Nk, float* __restrict__ Pf,
                                                           const float* __restrict__ shiftc, float* __restrict__ mvd, float* __restrict__ rw2,
                                                           int pf_pairs /* > 0: only that many pair records per row are ever read */,
                                                           KmMstep tail, int has_tail) {
    if ((int)blockIdx.x == k) {                       // (has_tail: the grid has this one block more) the M-step's global part, see km_mstep_tail
        if (has_tail) km_mstep_tail(tail);
        return;
    }
    const int a = blockIdx.x;
    const int kp = (k + 255) & ~255;
    if (mvd && threadIdx.x < KM_BND_TOP) {
        // distance bounds: from this centre to the KM_BND_TOP centres that moved most (kmeans_average_kernel), rounded DOWN
        const int c = __float_as_int(shiftc[k + KM_BND_TOP + threadIdx.x]);
        float v = __builtin_inff();                    // no such mover: it constrains nothing
        if ((unsigned)c < (unsigned)k) {
            double q = 0.0;
            for (int j = 0; j < n; ++j) { const double d = Ct[a * 16 + j] - Ct[c * 16 + j]; q = fma(d, d, q); }
            v = (float)(sqrt(q) * 0.999999);
            v = v > 0.0f ? v * 0.9999999f : v;
        }
        mvd[a * KM_BND_TOP + threadIdx.x] = v;
    }
    __shared__ unsigned long long keys[1024];
    const bool sorting = Nk != nullptr && kp <= 1024;
    for (int c = threadIdx.x; c < kp; c += 256) {
        float v = __builtin_inff();
        if (c < k) {
            double s = 0.0;
            for (int j = 0; j < n; ++j) { const double d = Ct[a * 16 + j] - Ct[c * 16 + j]; s = fma(d, d, s); }
            v = fminf((float)(s * 0.9999999), 3.0e38f);
        }
        const int r = c & 255;
        Dc[(int64_t)a * kp + (c & ~255) + (r & 63) * 4 + (r >> 6)] = v;
        if (sorting) keys[c] = ((unsigned long long)__float_as_uint(v) << 16) | (unsigned)c;      // non-negative floats (NaN included: last) order like their bits
    }
    if (!sorting) return;
    const int ks = kp <= 256 ? 256 : (kp <= 512 ? 512 : 1024);      // the bitonic network wants a power of two (kp = 768): keys of all ones behind the row
    for (int c = kp + threadIdx.x; c < ks; c += 256) keys[c] = ~0ull;
    __syncthreads();
    for (int size = 2; size <= ks; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = threadIdx.x; i < (ks >> 1); i += 256) {
                const int lo = ((i / stride) * 2 * stride) + (i % stride), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const unsigned long long x = keys[lo], y = keys[hi];
                if ((x > y) == up) { keys[lo] = y; keys[hi] = x; }
            }
            __syncthreads();                          // (round 5: wave-level ordering for the stages with strides up to 64 -- 6 workgroup barriers instead of 45 -- changed nothing: 25.8 us either way)
        }
    for (int i = threadIdx.x; i < kp; i += 256) Nk[(int64_t)a * kp + i] = keys[i];
    // (round 5) the squared radius from which a wave about this centre needs KM_WIDE_CAND candidates or more: kmeans_bounds_kernel puts
    // the tiles that hold such samples in FRONT of its list, so that the expensive passes are not the last ones drawn (a hint, nothing more)
    if (rw2 && threadIdx.x == 0)
        rw2[a] = k > KM_WIDE_CAND ? __uint_as_float((unsigned)(keys[KM_WIDE_CAND] >> 16)) / (float)(4.004 * (1.0 + KM_BND_BETA) * (1.0 + KM_BND_BETA)) : __builtin_inff();
    if (!Pf) return;
    // the row once more as FLOAT PAIRS for the packed-fp32 screening (kmeans_assign_pk_kernel), in the frame of the row's own centre:
    // pair t = the row's candidates 2 t and 2 t + 1, one 128-byte record [d0_0 d1_0 d0_1 d1_1 ... | -h0 -h1 | pad], d = c - c_a (formed
    // in fp64), h = |c - c_a|^2 / 2 -- what a wave streams through scalar registers as the second operand of v_pk_fma_f32.  With
    // y = x - c_a the float score y.d - h is the exact score of c minus that of c_a, and its terms are of the size of the wave's
    // radius, not of the data: the screening's error shrinks with them.  A pair past the end of the row (k odd) carries "minus infinity".
    // (one thread per candidate of the row forms its half of a record in the LDS -- twelve differences and their norm --, then the
    // block writes the records out as 16-byte pieces: entry by entry, with two of 32 threads walking the norm, this took 40 us)
    __shared__ float recs[256 * 32];                  // 256 pairs at a time (kp = 1024: two rounds)
    // (the LDS / DPP kernel's screening reads the pairs of a prefix of at most KM2_NBR_MAX candidates: 128 of the 256 at k = 512)
    const int npb = pf_pairs > 0 ? min(kp >> 1, pf_pairs) : (kp >> 1);
    for (int p0 = 0; p0 < npb; p0 += 256) {
        const int np = min(256, npb - p0);
        __syncthreads();
        for (int e = threadIdx.x; e < np * 32; e += 256) recs[e] = 0.0f;
        __syncthreads();
        for (int tm = threadIdx.x; tm < 2 * np; tm += 256) {
            const int t = tm >> 1, m = tm & 1;
            const int c = (int)(keys[2 * p0 + tm] & 0xFFFFull);
            float hneg = -3.0e38f;
            if (c < k) {
                double h2 = 0.0;
                for (int j = 0; j < n; ++j) {
                    const double d = Ct[c * 16 + j] - Ct[a * 16 + j];
                    h2 = fma(d, d, h2);
                    recs[t * 32 + 2 * j + m] = (float)d;
                }
                hneg = -(float)(0.5 * h2);
            }
            recs[t * 32 + 2 * KM_PK_NMAX + m] = hneg;
        }
        __syncthreads();
        float4* out = reinterpret_cast<float4*>(Pf + ((int64_t)a * (kp >> 1) + p0) * 32);
        const float4* src = reinterpret_cast<const float4*>(recs);
        for (int e = threadIdx.x; e < np * 8; e += 256) out[e] = src[e];
    }
}

// ---- distance bounds (round 4): which samples of the sorted loop need an E-step at all ------------------------------------------------
// Per position: ub >= d(x, c_a) for the sample's centre a, lb <= d(x, c) for every other centre c -- both left by the last E-step that
// evaluated the sample (kmeans_assign_lds_kernel, packed-fp32 path: the exact distance to the winner; the loser of the certified
// pair, the float runner-up with the screening's error, and for the centres outside the scanned prefix the triangle inequality
// through the reference centre).  After an M-step every centre has moved by shift_c (kmeans_average_kernel):
//     ub' = ub + shift_a
//     lb' = min( lb - (largest shift but the KM_BND_TOP largest),  min over those movers t != a of  max(lb - shift_t, d(c_a, c_t) - ub') )
// (Hamerly's bounds with the few largest movers taken apart: such a centre has come closer by its own shift -- unless it is far from
// the sample's own centre NOW, and then it is at least d(c_a, c_t) - ub' away whatever it did.  The largest shift is a single outlying cluster's on the config-3 data, 1.5 % of a cluster radius per iteration
// at iteration 70 against a median of 0.2 %; restricting it to the K nearest centres of a -- the others held off by the triangle
// inequality through c_a -- was probed for K = 8 ... 256, alone and all at once, and buys nothing in 12 dimensions with 512
// centres: the 65th nearest centre of a cluster is hardly farther than its 2nd; tools/attic/hamerly_probe.py.)  While ub' + margin < lb' the sample's
// nearest centre is still a, by more than the rounding of the E-step's scores: its label -- the full scan's -- cannot change, it keeps
// its bounds and is skipped.  Everything else goes to the list the next E-step walks: position p, in position order within a tile of
// 2048 positions, every tile's piece padded to whole waves with ~p of its last entry (a lane that loads the same row and counts for
// nothing), tiles in the order their blocks finish -- which no result depends on: labels are per sample, the member sums integers.
// All float operations round away from "skip".  NaN anywhere (a fresh sample, a poisoned centre) fails the test.
#ifndef KM_BND_TILE_
#define KM_BND_TILE_ 2048       // positions per block of kmeans_bounds_kernel = the unit its list is padded and classified in (4096 / 1024 threads:
                                 // the same 54 us, but the "expensive" class is coarser: list-form launches 247 against 241 us; 8192: 66 us)
#endif
constexpr int KM_BND_TILE = KM_BND_TILE_;
#ifndef KM_BND_THREADS
#define KM_BND_THREADS 512       // one round per block (2 442 blocks of 256 threads and four rounds were 1.2 waves of resident blocks: 59.6 us)
#endif
constexpr int KM_BND_BT = KM_BND_THREADS;
static_assert(KM_BND_TILE % (4 * KM_BND_BT) == 0 && KM_BND_BT % 64 == 0 && KM_BND_BT <= 1024, "a thread takes four consecutive positions per round");
__global__ void __launch_bounds__(KM_BND_BT) kmeans_bounds_kernel(int64_t N, int k, const int* __restrict__ labels, float* __restrict__ ub, float* __restrict__ lb,
                                                            const float* __restrict__ shiftc /* [k + KM_BND_TAIL]: kmeans_average_kernel */,
                                                            const float* __restrict__ mvd /* [k][KM_BND_TOP]: kmeans_cdist_kernel */,
                                                            const double* __restrict__ prm, int* __restrict__ list, int* __restrict__ nlist,
                                                            const float* __restrict__ rw2 /* [k] or nullptr: kmeans_cdist_kernel */, long long cap) {
    if (prm[3] != 0.0) return;                        // hold: an empty cluster waits for its relocation
    float mv_s[KM_BND_TOP];
    int mv_c[KM_BND_TOP];
#pragma unroll
    for (int t = 0; t < KM_BND_TOP; ++t) { mv_s[t] = shiftc[k + t]; mv_c[t] = __float_as_int(shiftc[k + KM_BND_TOP + t]); }
    const float m_rest = shiftc[k + 2 * KM_BND_TOP];
    const bool poisoned = !(m_rest == m_rest);         // a NaN shift (kmeans_average_kernel sets them all): every sample is evaluated
    __shared__ int buf[KM_BND_TILE + 64];
    __shared__ int wcnt[KM_BND_BT / 64];
    __shared__ int s_off, s_wide;
    if (threadIdx.x == 0) s_wide = 0;                 // (the round loop's barriers order it before its readers)
    bool wide = false;
    const float margin = (float)prm[0] * 1.0001f + 1.0e-37f;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * KM_BND_TILE;
    int count = 0;
    // a thread takes four consecutive positions per round (16-byte loads and stores; the arrays are the arena's: aligned); the loads of
    // all four rounds are requested before the first is worked on
    constexpr int ROUNDS = KM_BND_TILE / (4 * KM_BND_BT);
    int A[ROUNDS][4];
    float U[ROUNDS][4], L[ROUNDS][4];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int64_t p0 = base + r * (4 * KM_BND_BT) + threadIdx.x * 4;
        if (p0 + 3 < N) {
            const int4 av = *reinterpret_cast<const int4*>(labels + p0);
            const float4 uv = *reinterpret_cast<const float4*>(ub + p0), lv = *reinterpret_cast<const float4*>(lb + p0);
            A[r][0] = av.x; A[r][1] = av.y; A[r][2] = av.z; A[r][3] = av.w;
            U[r][0] = uv.x; U[r][1] = uv.y; U[r][2] = uv.z; U[r][3] = uv.w;
            L[r][0] = lv.x; L[r][1] = lv.y; L[r][2] = lv.z; L[r][3] = lv.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool in = p0 + j < N;
                A[r][j] = in ? labels[p0 + j] : 0; U[r][j] = in ? ub[p0 + j] : 0.0f; L[r][j] = in ? lb[p0 + j] : 0.0f;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int64_t p0 = base + r * (4 * KM_BND_BT) + threadIdx.x * 4;
        int (&a4)[4] = A[r];
        float (&u4)[4] = U[r], (&l4)[4] = L[r];
        const bool whole = p0 + 3 < N;
        unsigned act = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (p0 + j >= N) continue;
            bool active = true;
            const int a = a4[j];
            if ((unsigned)a < (unsigned)k) {
                const float u = (u4[j] + shiftc[a]) * 1.0000003f + 1.0e-37f;
                // every centre but the movers gives way by at most m_rest; mover t by its own shift -- or it is far from c_a NOW
                float l = l4[j] - m_rest;
                const float4 dm = *reinterpret_cast<const float4*>(mvd + a * KM_BND_TOP);
                const float dmv[KM_BND_TOP] = {dm.x, dm.y, dm.z, dm.w};
#pragma unroll
                for (int t = 0; t < KM_BND_TOP; ++t)
                    if (a != mv_c[t]) l = fminf(l, fmaxf(l4[j] - mv_s[t], dmv[t] - u));
                l = l > 0.0f ? l * 0.9999997f : l;
                // (fminf / fmaxf drop a NaN operand: a sample without bounds, a poisoned shift must not slip through them)
                if (!poisoned && l4[j] == l4[j] && u + margin < l) { active = false; u4[j] = u; l4[j] = l; }
                else if (rw2) wide = wide || u * u >= rw2[a];
            }
            act |= active ? 1u << j : 0u;
        }
        if (whole) {
            if (act != 15u) {                          // (an active sample's bounds stay as they are: its E-step rewrites them)
                *reinterpret_cast<float4*>(ub + p0) = make_float4(u4[0], u4[1], u4[2], u4[3]);
                *reinterpret_cast<float4*>(lb + p0) = make_float4(l4[0], l4[1], l4[2], l4[3]);
            }
        } else {
            for (int j = 0; j < 4; ++j)
                if (p0 + j < N && !((act >> j) & 1u)) { ub[p0 + j] = u4[j]; lb[p0 + j] = l4[j]; }
        }
        // ranks in position order: an inclusive scan of the threads' counts over the wave, the waves' totals through the LDS
        const int cnt = __builtin_popcount(act);
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= off) incl += v; }
        if (lane == 63) wcnt[w] = incl;
        __syncthreads();
        int at = count + incl - cnt, tot = 0;
#pragma unroll
        for (int q = 0; q < KM_BND_BT / 64; ++q) { const int c = wcnt[q]; at += q < w ? c : 0; tot += c; }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((act >> j) & 1u) buf[at++] = (int)(p0 + j);
        count += tot;
        __syncthreads();
    }
    if (count == 0) return;                           // (block-uniform)
    const int padded = (count + 63) & ~63;
    if ((int)threadIdx.x < padded - count) buf[count + threadIdx.x] = ~buf[count - 1];
    if (wide) s_wide = 1;
    __syncthreads();
    // rw2: two regions -- the tiles with a sample beyond its centre's "expensive" radius from the front of the buffer, the others from its
    // end (kmeans_assign_lds_kernel<LIST> draws its tickets front region first)
    if (threadIdx.x == 0) {
        if (rw2) {
            s_off = s_wide ? atomicAdd(nlist + KM_NL_FRONT, padded) : (int)(cap - (long long)atomicAdd(nlist + KM_NL_BACK, padded) - padded);
        } else {
            s_off = atomicAdd(nlist, padded);
        }
    }
    __syncthreads();
    int* out = list + s_off;
    for (int e = threadIdx.x; e < padded; e += KM_BND_BT) out[e] = buf[e];
}

// packed table from the centres: Ct[c] = [coordinates | half squared norm | zeros | minus the half norm]
__global__ void __launch_bounds__(256) kmeans_c2_kernel(int n, int k, const double* __restrict__ C, double* __restrict__ Ct) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= k) return;
    double cc[KM_NMAX];
    for (int j = 0; j < KM_NMAX; ++j) cc[j] = j < n ? C[(int64_t)c * n + j] : 0.0;
    (void)km_pack_centre(n, cc, Ct + (int64_t)c * 16);
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
// scikit-learn's only if a drawn value fell within ~1e-13 (relative) of a running-sum boundary -- or if two candidates of a
// round tie in exact arithmetic (two mutually nearest uncovered points both drawn: pot - closest[a] - closest[b] + d(a, b)
// either way; seen at N = 50, k = 16): the winner is then decided by the summation order of the potentials, scikit-learn's
// own by that of a BLAS matrix-vector product (tests/stress_parity.py checks that every mismatch is such a tie).
//
// ONE pass over the samples per centre (the loop is bound by HBM: 112 bytes per sample and pass).  Round c (centre c is
// being chosen, its L candidates are known):
//   pp_round  (one block per chunk)   closest_i <- min(closest_i, d(x_i, centre c-1))  -- the update the previous round owed --
//                                     and S[t][chunk] = sum over the chunk of min(closest_i, d(x_i, cand_t)) for every trial t
//   pp_decide (one block per trial)   pot_t = sum of S[t][.]; winner = first smallest; centre c = its sample.  The candidates of
//                                     round c+1 are drawn from the running sum of min(closest_i, d(x_i, centre c)): its chunk
//                                     sums are S[winner][.], and inside the one chunk a drawn value falls into, the 4096 values
//                                     are recomputed on the fly -- so the updated closest[] is never needed before the next
//                                     pp_round writes it.
// (The first version made two passes per centre: update + chunk sums, then the candidates' potentials; 535 us per centre at
// N = 1e7 against 270 us now.)  Round 3: pp_round screens rows of 16 samples with a float copy of the coordinates before it
// loads their fp64 ones (see the kernel).  Nothing returns to the host until all k centres are chosen.
#ifndef PP_CHUNK_
#define PP_CHUNK_ 4096                // (2048: 62 instead of 52 us per late round at 1e7 rows -- a block's fixed phases, not its longest pass, set the pace)
#endif
constexpr int PP_CHUNK = PP_CHUNK_;
constexpr int PP_LMAX = 16;           // trials per centre (k = 512: 8)
constexpr int PP_THREADS = 256;
constexpr int PP_PERMAX = (30000000 / PP_CHUNK + 1023) / 1024 + 1;      // chunk sums a thread of pp_decide walks at the 3e7 rows the seeding accepts
constexpr int PD_THREADS = 1024;      // pp_decide: 4 samples of the chunk per thread
constexpr int PP_SROWS = PP_LMAX + 1; // rows of S: one per trial + the chunk sums of closest itself (used for the first draw)
constexpr int PP_CW = KM_NMAX + 2;      // a row of the sharded run's candidate table: 16 coordinates, |x|^2, the GLOBAL sample index (as a double)
constexpr int PP_SCREEN_FROM = 8;     // rounds before this one evaluate every row in fp64: with so few centres most rows are in reach of a candidate

struct PPState {                      // device-resident scalars of the seeding loop
    double pot;                       // current potential
    long long cand[2][PP_LMAX];       // candidate sample indices: round c reads [(c - 1) & 1], draws the next round's into [c & 1]
    long long last;                   // sample index of the centre chosen last
    unsigned long long xmax_bits;     // max |x_i|^2 (finite rows) as the bit pattern of a non-negative double: the scale of the screening margin
    // the rows of the round's points as pp_round_kernel stages them ([16 coordinates | |x|^2 | pad]): written by the kernel that chooses
    // them (pp_decide / pp_first), so that the 2 400 blocks of a round read them directly instead of through an index (one dependent
    // load less at the head of every block)
    double crow[2][PP_LMAX][PP_CW];
    double lastrow[PP_CW];
};

// |x|^2 accumulated in coordinate order with FMAs: pp_transpose stores it, pp_round recomputes it from the coordinates it has
// loaded anyway (8 of the 112 bytes per sample and pass) -- one function, so that the two agree to the bit
template <int NS>
__device__ __forceinline__ double pp_norm2(const double x[KM_NMAX]) {
    constexpr int NJ = NS > 0 ? NS : KM_NMAX;                // coordinates beyond n are zero
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) s = fma(x[j], x[j], s);
    return s;
}

// The seeding makes k - 1 passes over the data, so the rows are first copied, centred, into a coordinate-major
// array Xt[j][i] (one strided read of X; every later access is a coalesced 512-byte wave load), with |x_i|^2 alongside
// (sklearn: row_norms(X, squared=True)).
template <int NS>
__global__ void __launch_bounds__(PP_THREADS) pp_transpose_kernel(int64_t N, int n, const double* __restrict__ X, int64_t xstride,
                                                                 const double* __restrict__ mean, double* __restrict__ Xt, double* __restrict__ xsq,
                                                                 float* __restrict__ Xf /* float copy for the screening of pp_round, or nullptr */,
                                                                 PPState* __restrict__ st) {
    const int64_t i = (int64_t)blockIdx.x * PP_THREADS + threadIdx.x;
    double xx = 0.0;
    if (i < N) {
        double x[KM_NMAX], xr[KM_NMAX];
        // (16-byte pieces of the row where its layout allows: see kmeans_range_kernel)
        if (NS > 0 && (NS & 1) == 0 && (xstride & 1) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0) {
#pragma unroll
            for (int j = 0; j + 1 < KM_NMAX; j += 2)
                if (j < NS) { const v2d v = *reinterpret_cast<const v2d*>(X + i * xstride + j); xr[j] = v[0]; xr[j + 1] = v[1]; }
        } else {
#pragma unroll
            for (int j = 0; j < KM_NMAX; ++j)
                if (NS > 0 ? (j < NS) : (j < n)) xr[j] = X[i * xstride + j];
        }
#pragma unroll
        for (int j = 0; j < KM_NMAX; ++j) {
            x[j] = 0.0;
            if (NS > 0 ? (j < NS) : (j < n)) {
                x[j] = xr[j] - (mean ? mean[j] : 0.0);
                Xt[(int64_t)j * N + i] = x[j];
                if (Xf) Xf[(int64_t)j * N + i] = (float)x[j];
            }
        }
        xx = pp_norm2<NS>(x);
        xsq[i] = xx;
    }
    if (Xf) {
        xx = (xx - xx == 0.0) ? xx : 0.0;                 // a NaN / inf row is never screened out (its comparisons fail)
        for (int off = 32; off > 0; off >>= 1) xx = fmax(xx, __shfl_down(xx, off));
        if ((threadIdx.x & 63) == 0) atomicMax(&st->xmax_bits, (unsigned long long)__double_as_longlong(xx));
    }
}

template <int NS>
__device__ __forceinline__ void pp_load_col(const double* __restrict__ Xt, int64_t N, int n, int64_t i, double x[KM_NMAX]) {
#pragma unroll
    for (int j = 0; j < KM_NMAX; ++j) x[j] = (NS > 0 ? (j < NS) : (j < n)) ? Xt[(int64_t)j * N + i] : 0.0;
}

// d(x, c) = max(0, (-2 x.c + |c|^2) + |x|^2), the dot product accumulated in coordinate order: the one formula every kernel
// of the seeding uses, so that a value recomputed by pp_decide is bit-identical to what pp_round stores later
template <int NS, class Row>
__device__ __forceinline__ double pp_dist(const double x[KM_NMAX], Row&& crow, double cn, double xx) {
    constexpr int NJ = NS > 0 ? NS : KM_NMAX;                // coordinates beyond n are zero on both sides
    double dot = 0.0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) dot = fma(x[j], crow(j), dot);
    const double d = (-2.0 * dot + cn) + xx;
    return d > 0.0 ? d : 0.0;
}

// Round 5: a second, coarser level in front of the float screening.  On trajectory-ordered data 16 consecutive samples are
// neighbours, so every ROW of PP_ROW = 16 samples (one 128-byte line of each fp64 coordinate array) carries a ball -- centre (floats,
// coordinate-major), radius -- plus the largest sqrt(closest) of its samples and the row's sum of closest: 64 bytes per row = 4 bytes
// per sample.  If for every point p of the round   |p - centre| - radius >= max sqrt(closest) (1 + 5e-6) + 2.5e-6 R   then no sample
// of the row is in reach of any point: the owed update changes nothing, every min(closest, d) is closest, and the row contributes its
// stored sum to every potential WITHOUT its samples being touched (not even closest[]).  From round ~100 on that certifies 85 % of the
// rows of the config-3 data (tools/attic/kmeanspp_ball_probe.py; the per-sample float test: 97 %), so a round reads ~15 instead of ~60 bytes
// per sample.  Rows that fail go through the per-sample float screening and, where that fails too, the fp64 path, exactly as before.
// The chunk sums are formed the same way on every path -- a fixed tree over the 16 lanes of a row, then a fixed order over the 256
// rows of the chunk -- so that a certified row's stored sum IS what its samples would add: screened and unscreened runs agree bit
// for bit (an experiments build checks that: KMV_PP_UNSCREENED).
constexpr int PP_ROW = 16;
constexpr int PP_NROW = PP_CHUNK / PP_ROW;            // rows per chunk
struct PPRows {
    float* rc = nullptr;                              // [n][nrows] ball centres
    float* rr = nullptr;                              // [nrows] radius: >= |x_i - centre| for every sample of the row (fp64 against the stored centre, rounded up)
    float* rs = nullptr;                              // [nrows] >= sqrt(closest_i) for every sample of the row
    double* rsum = nullptr;                           // [nrows] sum over the row of closest_i (the tree of pp_row_sum)
};
size_t kmeanspp_row_bytes(int64_t N, int n) {
    const size_t nr = (size_t)((N + PP_ROW - 1) / PP_ROW);
    return ((size_t)n * nr * 4 + 255) / 256 * 256 + 2 * ((nr * 4 + 255) / 256 * 256) + (nr * 8 + 255) / 256 * 256;
}
static PPRows pp_rows_from(void* buf, int64_t N, int n) {
    PPRows r;
    if (!buf) return r;
    const size_t nr = (size_t)((N + PP_ROW - 1) / PP_ROW);
    char* p = static_cast<char*>(buf);
    r.rc = reinterpret_cast<float*>(p); p += ((size_t)n * nr * 4 + 255) / 256 * 256;
    r.rr = reinterpret_cast<float*>(p); p += (nr * 4 + 255) / 256 * 256;
    r.rs = reinterpret_cast<float*>(p); p += (nr * 4 + 255) / 256 * 256;
    r.rsum = reinterpret_cast<double*>(p);
    return r;
}
// the sum over the 16 lanes of a row, in every lane of it: four exchange stages in which partners add the same two numbers (the same
// bits in both) -- lane ^ 1, lane ^ 2, then the mirror inside each half row (i <-> 7 - i) and inside the row (i <-> 15 - i): DPP
// controls, i.e. two 32-bit moves and one addition per stage instead of two ds_bpermute round trips
__device__ __forceinline__ double pp_row_sum(double v) {
    auto stage = [](double x, auto ctrl) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(x);
        const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)b, decltype(ctrl)::value, 0xF, 0xF, true);
        const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(b >> 32), decltype(ctrl)::value, 0xF, 0xF, true);
        return x + __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    v = stage(v, std::integral_constant<int, 0xB1>{});       // quad_perm [1,0,3,2]
    v = stage(v, std::integral_constant<int, 0x4E>{});       // quad_perm [2,3,0,1]
    v = stage(v, std::integral_constant<int, 0x141>{});      // row_half_mirror
    v = stage(v, std::integral_constant<int, 0x140>{});      // row_mirror
    return v;
}
// ball of every row: centre = the mean of its samples (as floats), radius against that stored centre
template <int NS>
__global__ void __launch_bounds__(256) pp_rowball_kernel(int64_t N, int n, const double* __restrict__ Xt, PPRows rows) {
    const int64_t nrows = (N + PP_ROW - 1) / PP_ROW;
    const int64_t R = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (R >= nrows) return;
    const int64_t i0 = R * PP_ROW;
    const int cnt = (int)((N - i0) < PP_ROW ? (N - i0) : PP_ROW);
    constexpr int NJ = NS > 0 ? NS : KM_NMAX;
    // two sweeps over the row's 16 x n values (the second one hits the cache): centre, then squared distances to the STORED centre
    double d2[PP_ROW];
#pragma unroll
    for (int q = 0; q < PP_ROW; ++q) d2[q] = 0.0;
#pragma unroll 1
    for (int j = 0; j < NJ; ++j) {
        if (!(NS > 0 || j < n)) break;
        double x[PP_ROW], m = 0.0;
#pragma unroll
        for (int q = 0; q < PP_ROW; ++q) { x[q] = q < cnt ? Xt[(int64_t)j * N + i0 + q] : 0.0; m += x[q]; }
        m /= (double)cnt;
        const float cf = (m - m == 0.0) ? (float)m : 0.0f;      // (a NaN / inf sample: its distance below is not finite and the row is never certified)
        rows.rc[(int64_t)j * nrows + R] = cf;
#pragma unroll
        for (int q = 0; q < PP_ROW; ++q) { const double e = x[q] - (double)cf; d2[q] = fma(e, e, d2[q]); }
    }
    double r2 = 0.0;
    bool bad = false;
#pragma unroll
    for (int q = 0; q < PP_ROW; ++q) {
        if (q >= cnt) continue;
        if (!(d2[q] - d2[q] == 0.0)) bad = true;
        r2 = d2[q] > r2 ? d2[q] : r2;
    }
    rows.rr[R] = bad ? __builtin_nanf("") : (float)sqrt(r2) * 1.000001f + 1.0e-37f;
}

// upd: 0 = closest stands (round 1: it already holds d(., centre 0)); 1 = closest_i = d(x_i, x_last) (before round 1);
//      2 = closest_i = min(closest_i, d(x_i, x_last)).
// S[t][chunk] = sum over the chunk of min(closest_i, d(x_i, cand_t)), t < L;  S[PP_LMAX][chunk] = sum of closest_i.
// One block per 4096-sample chunk = 256 rows of 16 samples.  BT threads per block: 256 at size (HBM-bound, many blocks per CU); 1024
// when the whole sample set is a few chunks (64 rows per pass instead of 16: the exposed latency of a pass is a good part of a round
// there -- 36 658 rows: 41 -> 34 us per centre in round 3).
template <int NS, int BT, int LM /* trials the registers are sized for: 8 (k <= 1096) or PP_LMAX */>
__global__ void __launch_bounds__(BT) pp_round_kernel(int64_t N, int n, int L, int nchunks, const double* __restrict__ Xt,
                                                             const double* __restrict__ xsq, const PPState* __restrict__ st, int par, int upd,
                                                             double* __restrict__ closest, double* __restrict__ S,
                                                             const float* __restrict__ Xf /* [n][N] float copy of Xt, or nullptr: no screening */,
                                                             const double* __restrict__ crow /* sharded run: the round's rows [PP_LMAX + 1][PP_CW] (nullptr: from Xt through st) */,
                                                             PPRows rows /* rc == nullptr: no row level */) {
    // candidate rows in LDS: [trial][16 coordinates | norm | pad]; row L = the centre chosen last.  A compiler-level memory
    // barrier in front of every trial keeps their reads where they are used: as plain loop invariants the compiler hoisted
    // all 16 x 17 of them into registers (256 VGPRs + scratch, one wave per SIMD).
    constexpr int CSW = KM_NMAX + 2;
    __shared__ double cs[(PP_LMAX + 1) * CSW];
    __shared__ float csf[(PP_LMAX + 1) * KM_NMAX];    // the same rows as floats (screening)
    // what every row of the chunk adds to S[t][chunk], t < L; row L: to the sum of closest.  Dynamic ((L + 1) x 256 doubles: 18 KB at
    // k = 512 instead of 34 KB for the 16 trials the loop allows): the LDS is what limits the blocks per CU here.
    extern __shared__ double rowval_dyn[];
    double (*rowval)[PP_NROW] = reinterpret_cast<double (*)[PP_NROW]>(rowval_dyn);
    __shared__ unsigned short todo[PP_NROW];          // rows the ball test did not certify, in position order
    __shared__ int wcnt[4];
    for (int e = threadIdx.x; e < (L + 1) * CSW; e += BT) {
        const int t = e / CSW, j = e % CSW;
        if (crow) {
            // sharded run: a candidate lives on ONE rank; its row was exchanged (pp_decide_sh_kernel) -- slot KM_NMAX holds |x|^2
            cs[e] = j <= KM_NMAX ? crow[(t < L ? t : PP_LMAX) * PP_CW + j] : 0.0;
        } else {
            cs[e] = j <= KM_NMAX ? (t < L ? st->crow[par][t][j] : st->lastrow[j]) : 0.0;
        }
        if (j < KM_NMAX) csf[t * KM_NMAX + j] = (float)cs[e];
    }
    // Screening (rounds with candidates, once closest[] exists): most samples are far from all of the round's points -- its L
    // candidates and the centre chosen last --, and then the owed update changes nothing and every min(closest, d) is closest.  A
    // float copy of the coordinates (48 instead of 96 bytes per sample) certifies that: in float, difference form,
    //   sqrt(d_float) >= sqrt(closest) (1 + 5e-6) + 2.5e-6 R     (R^2 = 2 max |x|^2)
    // implies true d >= closest by more than (1e-6 R)^2, ten orders above the rounding of the fp64 formula (float inputs err by
    // 6e-8 R per coordinate, the float arithmetic by 1e-6 relative).  Decided per row of 16 lanes (one 128-byte line of every
    // fp64 coordinate array); a row that is not certified goes through the fp64 path below unchanged.
    const bool screening = Xf != nullptr && rows.rsum != nullptr && upd != 1 && L > 0;      // (a screened row adds its STORED sum: the row data is part of it)
    const bool balls = screening && rows.rc != nullptr;
    const int npts = upd == 2 ? L + 1 : L;            // row L (the last centre) only when its update is owed
    float rmarg = 0.0f;
    if (screening) rmarg = 2.5e-6f * (float)sqrt(2.0 * __longlong_as_double((long long)st->xmax_bits)) * 1.000001f + 1.0e-37f;
    __syncthreads();

    constexpr int NF = NS > 0 ? NS : KM_NMAX;
    const int64_t nrows = (N + PP_ROW - 1) / PP_ROW;
    const int64_t R0 = (int64_t)blockIdx.x * PP_NROW;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // ---- level 1: one thread per row of the chunk
    int ntodo;
    {
        bool open = false;                             // the row needs its samples looked at
        const int r = threadIdx.x;
        if (r < PP_NROW) {
            const int64_t R = R0 + r;
            if (R >= nrows) {
                for (int t = 0; t <= L; ++t) rowval[t][r] = 0.0;
            } else {
                open = true;
                if (balls) {
                    float m[NF];
#pragma unroll
                    for (int j = 0; j < NF; ++j) m[j] = (NS > 0 || j < n) ? rows.rc[(int64_t)j * nrows + R] : 0.0f;
                    // |x - p| >= |centre - p| - radius for every sample of the row; the float distance errs by 1e-6 relative, its inputs
                    // by 6e-8 R per coordinate: both inside the margin the per-sample test uses
                    const float need_d = rows.rs[R] * 1.000005f + rmarg + rows.rr[R];
                    const float thr = need_d * need_d * 1.0000006f;
                    float dmin = 3.0e38f;
#pragma unroll 1
                    for (int t = 0; t < npts; ++t) {
                        const float* row = csf + t * KM_NMAX;
                        float d = 0.0f;
#pragma unroll
                        for (int j = 0; j < NF; ++j) { const float e = m[j] - row[j]; d = fmaf(e, e, d); }
                        dmin = fminf(dmin, d);
                        if (!(d == d)) dmin = 0.0f;
                    }
                    if (dmin >= thr) {                 // (false for a NaN radius / bound)
                        open = false;
                        const double v = rows.rsum[R];
                        for (int t = 0; t < L; ++t) rowval[t][r] = v;
                        rowval[L][r] = v;
                    }
                }
            }
        }
        const unsigned long long ob = __ballot(open);
        if (wv < 4 && lane == 0) wcnt[wv] = __builtin_popcountll(ob);
        __syncthreads();
        if (r < PP_NROW && open) {
            int at = __builtin_popcountll(ob & ((1ull << lane) - 1ull));
            for (int q = 0; q < wv; ++q) at += wcnt[q];
            todo[at] = (unsigned short)r;
        }
        ntodo = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
    }
    // ---- level 2: the open rows through the per-sample float test, BT / 16 rows per step, a lane per sample; PP_UNR steps have
    // their loads in flight together (a block is a chain of short dependent phases: every step saved is ~2 us of exposed latency, and
    // 2 400 blocks go through the chip in a few rounds).  Rows in which a lane fails go on to the fp64 list.
    constexpr int RPP = BT / PP_ROW;
    constexpr int PP_UNR = BT >= 1024 ? 1 : 2;
    __shared__ unsigned short todo2[PP_NROW];
    __shared__ int s_n2;
    const int l16 = threadIdx.x & 15;
    int n2 = ntodo;
    const unsigned short* list2 = todo;                // without screening every open row takes the fp64 path
    if (screening) {
        if (threadIdx.x == 0) s_n2 = 0;
        __syncthreads();
#pragma unroll 1
        for (int q0 = 0; q0 < ntodo; q0 += RPP * PP_UNR) {
            float xf[PP_UNR][NF];
            double old[PP_UNR];
            int rr_[PP_UNR];
            bool val_[PP_UNR], act_[PP_UNR];
#pragma unroll
            for (int u = 0; u < PP_UNR; ++u) {
                const int slot = q0 + u * RPP + (threadIdx.x >> 4);
                act_[u] = slot < ntodo;
                rr_[u] = act_[u] ? todo[slot] : 0;
                const int64_t i = (R0 + rr_[u]) * PP_ROW + l16;
                val_[u] = act_[u] && i < N;
                old[u] = val_[u] ? closest[i] : 0.0;
#pragma unroll
                for (int j = 0; j < NF; ++j) xf[u][j] = (val_[u] && (NS > 0 || j < n)) ? Xf[(int64_t)j * N + i] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < PP_UNR; ++u) {
                bool certified = true;                // lanes without a sample do not ask for anything
                if (val_[u]) {
                    const float so = sqrtf((float)old[u] * 1.0000003f + 1.0e-37f) * 1.000005f + rmarg;
                    const float thr = so * so * 1.0000003f;
                    float dmin = 3.0e38f;
#pragma unroll 1
                    for (int t = 0; t < npts; ++t) {
                        const float* row = csf + t * KM_NMAX;
                        float d = 0.0f;
#pragma unroll
                        for (int j = 0; j < NF; ++j) { const float e = xf[u][j] - row[j]; d = fmaf(e, e, d); }
                        dmin = fminf(dmin, d);        // a NaN distance is ignored by fminf: guard below
                        if (!(d == d)) dmin = 0.0f;
                    }
                    certified = dmin >= thr;          // false for a NaN threshold
                }
                unsigned long long b = __ballot(!certified);
                b |= b >> 8; b |= b >> 4; b |= b >> 2; b |= b >> 1;        // bit 16 g = some lane of the wave's row g is not certified
                const bool rowneed = ((b >> (threadIdx.x & 48)) & 1ull) != 0ull;
                if (act_[u] && l16 == 0) {
                    if (rowneed) {
                        todo2[atomicAdd(&s_n2, 1)] = (unsigned short)rr_[u];     // (any order: a row's result goes to its own slot)
                    } else {
                        // every sample of the row certified: closest stands, and the stored sum is the row's sum
                        const double v = rows.rsum[R0 + rr_[u]];
                        for (int t = 0; t < L; ++t) rowval[t][rr_[u]] = v;
                        rowval[L][rr_[u]] = v;
                    }
                }
            }
        }
        __syncthreads();
        n2 = s_n2;
        list2 = todo2;
    }
    // ---- the fp64 path for the rows that are left, BT / 16 per pass: the owed update, the trials' values, the row's sums
    auto fp64_load = [&](int q0, int& r, bool& active, bool& valid, int64_t& i, double& old, double (&x)[KM_NMAX]) {
        const int slot = q0 + (threadIdx.x >> 4);
        active = slot < n2;
        r = active ? list2[slot] : 0;
        i = (R0 + r) * PP_ROW + l16;
        valid = active && i < N;
        old = (upd == 1 || !valid) ? 0.0 : closest[i];
#pragma unroll
        for (int j = 0; j < KM_NMAX; ++j) x[j] = (valid && (NS > 0 ? (j < NS) : (j < n))) ? Xt[(int64_t)j * N + i] : 0.0;
    };
    auto fp64_work = [&](int r, bool active, bool valid, int64_t i, double old, const double (&x)[KM_NMAX]) {
        const int64_t R = R0 + r;
        double v0 = old, vt[LM];
#pragma unroll
        for (int t = 0; t < LM; ++t) vt[t] = old;  // no sample: zeros
        if (valid) {
            const double xx = pp_norm2<NS>(x);
            if (upd) {
                asm volatile("" ::: "memory");
                const double* row = cs + L * CSW;
                const double d = pp_dist<NS>(x, [&](int j) { return row[j]; }, row[KM_NMAX], xx);
                if (upd == 1 || d < old) { old = d; closest[i] = d; }     // np.minimum(closest, d); unchanged values are not rewritten
            }
            v0 = old;
#pragma unroll
            for (int t = 0; t < LM; ++t) {
                vt[t] = old;
                if (t < L) {
                    asm volatile("" ::: "memory");
                    const double* row = cs + t * CSW;
                    const double d = pp_dist<NS>(x, [&](int j) { return row[j]; }, row[KM_NMAX], xx);
                    vt[t] = old < d ? old : d;
                }
            }
        }
        const double s0 = pp_row_sum(v0);
#pragma unroll
        for (int t = 0; t < LM; ++t)
            if (t < L) vt[t] = pp_row_sum(vt[t]);
        float mx = (float)v0;                          // the row's largest closest, for the ball test of later rounds (a NaN shows in s0)
        mx = fmaxf(mx, __shfl_xor(mx, 8)); mx = fmaxf(mx, __shfl_xor(mx, 4)); mx = fmaxf(mx, __shfl_xor(mx, 2)); mx = fmaxf(mx, __shfl_xor(mx, 1));
        if (active && l16 == 0) {
#pragma unroll
            for (int t = 0; t < LM; ++t)
                if (t < L) rowval[t][r] = vt[t];
            rowval[L][r] = s0;
            if (rows.rsum) {
                rows.rsum[R] = s0;
                // >= sqrt(closest_i): (float) rounds to nearest, the factors cover that and sqrtf; NaN stays NaN (never certified)
                rows.rs[R] = (s0 - s0 == 0.0) ? sqrtf(mx * 1.0000003f + 1.0e-37f) * 1.0000003f : __builtin_nanf("");
            }
        }
    };
#pragma unroll 1
    for (int q0 = 0; q0 < n2; q0 += RPP) {
        // (two passes with their loads in flight together were tried: 144 VGPRs, three blocks per CU instead of four, 76 against 58 us
        // per late round -- this kernel lives on the number of blocks a CU holds)
        int ra;
        bool aa, va;
        int64_t ia;
        double oa, xa[KM_NMAX];
        fp64_load(q0, ra, aa, va, ia, oa, xa);
        fp64_work(ra, aa, va, ia, oa, xa);
    }
    __syncthreads();
    // ---- the chunk's sums: lane l adds rows l, l + 64, l + 128, l + 192, then a fixed tree; a wave per trial
    for (int t = wv; t <= L; t += BT / 64) {
        const int tr = t < L ? t : PP_LMAX;
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < PP_NROW / 64; ++q) a += rowval[t][lane + 64 * q];
        for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
        if (lane == 0) S[(int64_t)tr * nchunks + blockIdx.x] = a;
    }
}

// running sums of a row of chunk sums, in chunk order, into prefix[0 .. nchunks] (prefix[nchunks] = the total): every thread adds up a
// contiguous segment; the segment totals are scanned in thread order (shuffle scan inside a wave, the 16 wave totals chained) --
// fixed grouping, same result every run.  All PD_THREADS threads of the block; ends with a barrier.
__device__ __forceinline__ void pp_chunk_prefix(int nchunks, const double* __restrict__ cs, double* prefix, double* wtot) {
    const int tid = threadIdx.x;
    const int per = (nchunks + PD_THREADS - 1) / PD_THREADS;
    const int b0 = tid * per < nchunks ? tid * per : nchunks, b1 = b0 + per < nchunks ? b0 + per : nchunks;
    // the thread's segment of chunk sums: loaded together (at most 9: the LDS bound on nchunks / 1024 threads), added in order
    constexpr int PERMAX = PP_PERMAX;
    double seg[PERMAX];
#pragma unroll
    for (int q = 0; q < PERMAX; ++q) seg[q] = b0 + q < b1 ? cs[b0 + q] : 0.0;
    double a = 0.0;
#pragma unroll
    for (int q = 0; q < PERMAX; ++q)
        if (b0 + q < b1) a += seg[q];
    const int lane = tid & 63, w = tid >> 6;
    double incl = a;
    for (int off = 1; off < 64; off <<= 1) { const double t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    double woff = 0.0;
    for (int q = 0; q < w; ++q) woff += wtot[q];
    double run = woff + (incl - a);
#pragma unroll
    for (int q = 0; q < PERMAX; ++q)
        if (b0 + q < b1) { prefix[b0 + q] = run; run += seg[q]; }
    if (tid == PD_THREADS - 1) prefix[nchunks] = run;
    __syncthreads();                              // (wtot is reused by pp_find_sample)
}

// The sample a drawn value v falls on: first chunk b with prefix[b + 1] >= v (binary search, block-uniform); inside it thread th owns
// 4 consecutive samples whose values min(closest_i, d(x_i, last centre)) are recomputed here; *found = first index whose running sum
// reaches v, clipped to N - 1 (np.searchsorted + np.clip).  cl / cn: the centre chosen last (coordinates, |c|^2) when have_last.
template <int NS>
__device__ __forceinline__ void pp_find_sample(int64_t N, int n, int nchunks, double v, const double* prefix, bool have_last, const double (&cl)[KM_NMAX],
                                               double cn, const double* __restrict__ Xt, const double* __restrict__ xsq,
                                               const double* __restrict__ closest, double* vals, double* wtot, int* wfirst, long long* found) {
    const int tid = threadIdx.x;
    int lo = 0, hi = nchunks;                     // answer in [lo, hi]; hi = nchunks means "beyond the end"
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (prefix[mid + 1] >= v) hi = mid; else lo = mid + 1;
    }
    if (tid == 0) *found = N - 1;
    if (lo < nchunks) {
        const int64_t base = (int64_t)lo * PP_CHUNK;
#pragma unroll
        for (int q = 0; q < PP_CHUNK / PD_THREADS; ++q) {
            const int sidx = q * PD_THREADS + tid;
            const int64_t i = base + sidx;
            double val = 0.0;
            if (i < N) {
                val = closest[i];
                if (have_last) {
                    double x[KM_NMAX];
                    pp_load_col<NS>(Xt, N, n, i, x);
                    const double d = pp_dist<NS>(x, [&](int j) { return cl[j]; }, cn, xsq[i]);
                    val = d < val ? d : val;
                }
            }
            vals[sidx] = val;
        }
        __syncthreads();
        constexpr int PER = PP_CHUNK / PD_THREADS;    // consecutive samples per thread
        double mine[PER], sg = 0.0;
#pragma unroll
        for (int q = 0; q < PER; ++q) { mine[q] = vals[tid * PER + q]; sg += mine[q]; }
        const int lane = tid & 63, w = tid >> 6;
        double incl = sg;
        for (int off = 1; off < 64; off <<= 1) { const double t = __shfl_up(incl, off); if (lane >= off) incl += t; }
        if (lane == 63) wtot[w] = incl;
        __syncthreads();
        double woff = 0.0;
        for (int q = 0; q < w; ++q) woff += wtot[q];
        incl += woff;
        const double start = prefix[lo] + (incl - sg);
        const unsigned long long reach = __ballot(prefix[lo] + incl >= v);
        if (lane == 0) wfirst[w] = reach ? __ffsll((long long)reach) - 1 : -1;
        __syncthreads();
        int ow = -1;
        for (int q = 0; q < PD_THREADS / 64; ++q) if (wfirst[q] >= 0) { ow = q; break; }
        if (ow >= 0) {
            if (w == ow && lane == wfirst[ow]) {
                const int64_t b4 = base + (int64_t)tid * PER;
                double run = start;
                long long idx = b4 + PER - 1 < N ? b4 + PER - 1 : N - 1;
                for (int q = 0; q < PER; ++q) {
                    const int64_t i = b4 + q;
                    if (i >= N) break;
                    run += mine[q];
                    if (run >= v) { idx = i; break; }
                }
                *found = idx;
            }
        } else if (tid == 0) {
            // rounding between the chunk's tree sum and its sequential sum: the value lies just past this chunk
            const long long nxt = (long long)(lo + 1) * PP_CHUNK;
            *found = nxt < N ? nxt : N - 1;
        }
    }
}

// potential of one trial: lane l adds every 64th chunk sum (eight loads in flight, additions in the order of the plain loop), then a
// fixed shuffle tree -- the one summation pp_decide_kernel and pp_tot_kernel share
__device__ __forceinline__ double pp_row_total(const double* __restrict__ Sr, int nchunks, int l) {
    double a = 0.0;
    for (int b0 = l; b0 < nchunks; b0 += 64 * 8) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int b = b0 + 64 * q; v[q] = b < nchunks ? Sr[b] : 0.0; }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (b0 + 64 * q < nchunks) a += v[q];
    }
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
    return a;
}

// One block per trial of the NEXT round (one block when nothing is drawn); every block repeats the cheap global part.
//   c > 0: winner = first trial with the smallest potential (np.argmin) among the candidates cand[(c-1)&1]; block 0 records it
//          as centre c.  Potentials: one wave per trial, a lane takes every 64th chunk sum, then a fixed tree.
//   draw : value = u[trial] * pot (pot = the winner's potential; before round 1 the sum of the chunk sums of closest);
//          chunk = first chunk whose running sum reaches it (running sums of S[winner][.]); inside the chunk thread th
//          owns 4 consecutive samples, whose values min(closest_i, d(x_i, centre c)) are recomputed here.
//          cand = first index with running sum >= value, clipped to N - 1 (np.searchsorted + np.clip).
template <int NS>
__global__ void __launch_bounds__(PD_THREADS) pp_decide_kernel(int64_t N, int n, int nchunks, int L, int c, int draw, const double* __restrict__ u,
                                                              const double* __restrict__ Xt, const double* __restrict__ xsq,
                                                              const double* __restrict__ closest, const double* __restrict__ S,
                                                              const double* __restrict__ X, int64_t xstride, const double* __restrict__ mean,
                                                              PPState* __restrict__ st, double* __restrict__ C, long long* __restrict__ indices) {
    extern __shared__ double dyn[];               // prefix[nchunks + 1] | vals[PP_CHUNK]
    double* prefix = dyn;
    double* vals = dyn + (nchunks + 1);
    __shared__ double pots[PP_LMAX];
    __shared__ double wtot[PD_THREADS / 64];
    __shared__ int wfirst[PD_THREADS / 64];
    __shared__ long long s_last, s_found;
    __shared__ double s_pot;
    __shared__ int s_row;
    const int tid = threadIdx.x;
    if (c > 0) {
        // one wave per trial: lane l adds every 64th chunk sum, then a fixed shuffle tree
        const int t = tid >> 6, l = tid & 63;
        if (t < L) {
            // (eight loads in flight at a time: as written first, every addition waited for its own load -- 38 L2 round trips one after
            // the other at N = 1e7, most of this kernel's 26 us)
            const double a = pp_row_total(S + (int64_t)t * nchunks, nchunks, l);
            if (l == 0) pots[t] = a;
        }
        __syncthreads();
        if (tid == 0) {
            int best = 0;
            for (int q = 1; q < L; ++q) if (pots[q] < pots[best]) best = q;
            const long long win = st->cand[(c - 1) & 1][best];
            s_row = best; s_last = win; s_pot = pots[best];
            if (blockIdx.x == 0) { st->pot = pots[best]; st->last = win; indices[c] = win; }
        }
        __syncthreads();
        if (blockIdx.x == 0 && tid < n) C[(int64_t)c * n + tid] = X[s_last * xstride + tid] - (mean ? mean[tid] : 0.0);
        if (blockIdx.x == 0 && tid < PP_CW)          // the next round's "centre chosen last", as pp_round_kernel stages it
            st->lastrow[tid] = tid < KM_NMAX ? ((NS > 0 ? tid < NS : tid < n) ? Xt[(int64_t)tid * N + s_last] : 0.0) : (tid == KM_NMAX ? xsq[s_last] : 0.0);
    } else {
        if (tid == 0) { s_row = PP_LMAX; s_last = -1; s_pot = 0.0; }
        __syncthreads();
    }
    if (!draw) return;
    pp_chunk_prefix(nchunks, S + (int64_t)s_row * nchunks, prefix, wtot);
    if (c == 0 && tid == 0) { s_pot = prefix[nchunks]; if (blockIdx.x == 0) st->pot = prefix[nchunks]; }
    __syncthreads();
    const int trial = blockIdx.x;
    const double v = u[trial] * s_pot;
    const long long last = s_last;
    double cl[KM_NMAX], cn = 0.0;
#pragma unroll
    for (int j = 0; j < KM_NMAX; ++j) cl[j] = 0.0;
    if (last >= 0) { pp_load_col<NS>(Xt, N, n, last, cl); cn = xsq[last]; }
    pp_find_sample<NS>(N, n, nchunks, v, prefix, last >= 0, cl, cn, Xt, xsq, closest, vals, wtot, wfirst, &s_found);
    __syncthreads();
    if (tid == 0) st->cand[c & 1][trial] = s_found;
    if (tid < PP_CW) {
        const long long f = s_found;
        st->crow[c & 1][trial][tid] = tid < KM_NMAX ? ((NS > 0 ? tid < NS : tid < n) ? Xt[(int64_t)tid * N + f] : 0.0) : (tid == KM_NMAX ? xsq[f] : 0.0);
    }
}

__global__ void pp_first_kernel(int n, int64_t N, long long first, const double* __restrict__ X, int64_t xstride, const double* __restrict__ mean,
                                const double* __restrict__ Xt, const double* __restrict__ xsq, PPState* __restrict__ st, double* __restrict__ C,
                                long long* __restrict__ indices) {
    const int t = threadIdx.x;
    if (t == 0) { st->last = first; st->pot = 0.0; indices[0] = first; }
    if (t < n) C[t] = X[first * xstride + t] - (mean ? mean[t] : 0.0);
    if (t < PP_CW) st->lastrow[t] = t < KM_NMAX ? (t < n ? Xt[(int64_t)t * N + first] : 0.0) : (t == KM_NMAX ? xsq[first] : 0.0);
}

int kmeanspp_chunks(int64_t N) { return (int)((N + PP_CHUNK - 1) / PP_CHUNK); }
size_t kmeanspp_sum_doubles(int64_t N) { return (size_t)PP_SROWS * kmeanspp_chunks(N); }
size_t kmeanspp_state_bytes() { return sizeof(PPState); }

// the whole seeding loop, stream ordered; u: device [(k-1) * L] uniforms; Xt: device scratch [n][N]; S: device scratch
// [kmeanspp_sum_doubles(N)]; C: device [k][n]; indices: device [k] (int64)
hipError_t launch_kmeanspp(hipStream_t st, int64_t N, int n, int k, int L, const double* X, int64_t xstride, const double* mean,
                           long long first, const double* u, double* Xt, double* xsq, double* closest, double* S,
                           void* state, double* C, long long* indices, float* Xf, void* rowbuf) {
    if (n > KM_NMAX || L > PP_LMAX || L < 1) return hipErrorInvalidValue;
    const int nchunks = kmeanspp_chunks(N);
    const size_t lds = ((size_t)nchunks + 1 + PP_CHUNK) * 8;      // prefix table + the chunk's values: N <= 3e7
    if (lds > 150 * 1024 || (nchunks + PD_THREADS - 1) / PD_THREADS > PP_PERMAX) return hipErrorInvalidValue;     // (pp_decide's register segment)
    PPState* ps = reinterpret_cast<PPState*>(state);
    hipError_t e0 = hipMemsetAsync(ps, 0, sizeof(PPState), st);
    if (e0 != hipSuccess) return e0;
    const unsigned nb = (unsigned)((N + PP_THREADS - 1) / PP_THREADS);
    const bool small = nchunks <= 256 && N > 2048;     // fewer chunks than CUs (and more than a few waves of samples): latency, not bandwidth
    const PPRows rows = pp_rows_from(Xf ? rowbuf : nullptr, N, n);
    const unsigned nrb = (unsigned)(((N + PP_ROW - 1) / PP_ROW + 255) / 256);
#define PP_DISPATCH(NS_) do { \
        hipError_t e_ = hipFuncSetAttribute((const void*)pp_decide_kernel<NS_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e_ != hipSuccess) return e_; \
        hipLaunchKernelGGL(pp_transpose_kernel<NS_>, dim3(nb), dim3(PP_THREADS), 0, st, N, n, X, xstride, mean, Xt, xsq, Xf, ps); \
        if (rows.rc) hipLaunchKernelGGL(pp_rowball_kernel<NS_>, dim3(nrb), dim3(256), 0, st, N, n, Xt, rows); \
        hipLaunchKernelGGL(pp_first_kernel, dim3(1), dim3(64), 0, st, n, N, first, X, xstride, mean, Xt, xsq, ps, C, indices); \
        if (k > 1) { \
            if (small) hipLaunchKernelGGL((pp_round_kernel<NS_, 1024, PP_LMAX>), dim3(nchunks), dim3(1024), PP_NROW * 8, st, N, n, 0, nchunks, Xt, xsq, ps, 0, 1, closest, S, Xf, nullptr, rows); \
            else hipLaunchKernelGGL((pp_round_kernel<NS_, PP_THREADS, 8>), dim3(nchunks), dim3(PP_THREADS), PP_NROW * 8, st, N, n, 0, nchunks, Xt, xsq, ps, 0, 1, closest, S, Xf, nullptr, rows); \
            hipLaunchKernelGGL(pp_decide_kernel<NS_>, dim3(L), dim3(PD_THREADS), lds, st, N, n, nchunks, L, 0, 1, u, Xt, xsq, closest, S, X, xstride, mean, ps, C, indices); \
        } \
        for (int c = 1; c < k; ++c) { \
            const int draw = c + 1 < k ? 1 : 0; \
            if (small) hipLaunchKernelGGL((pp_round_kernel<NS_, 1024, PP_LMAX>), dim3(nchunks), dim3(1024), (size_t)(L + 1) * PP_NROW * 8, st, N, n, L, nchunks, Xt, xsq, ps, (c - 1) & 1, c == 1 ? 0 : 2, closest, S, c >= PP_SCREEN_FROM ? Xf : nullptr, nullptr, rows); \
            else if (L <= 8) hipLaunchKernelGGL((pp_round_kernel<NS_, PP_THREADS, 8>), dim3(nchunks), dim3(PP_THREADS), (size_t)(L + 1) * PP_NROW * 8, st, N, n, L, nchunks, Xt, xsq, ps, (c - 1) & 1, c == 1 ? 0 : 2, closest, S, c >= PP_SCREEN_FROM ? Xf : nullptr, nullptr, rows); \
            else hipLaunchKernelGGL((pp_round_kernel<NS_, PP_THREADS, PP_LMAX>), dim3(nchunks), dim3(PP_THREADS), (size_t)(L + 1) * PP_NROW * 8, st, N, n, L, nchunks, Xt, xsq, ps, (c - 1) & 1, c == 1 ? 0 : 2, closest, S, c >= PP_SCREEN_FROM ? Xf : nullptr, nullptr, rows); \
            hipLaunchKernelGGL(pp_decide_kernel<NS_>, dim3(draw ? L : 1), dim3(PD_THREADS), lds, st, N, n, nchunks, L, c, draw, u + (size_t)c * L, Xt, xsq, closest, S, X, xstride, mean, ps, C, indices); \
        } } while (0)
    if (n == 12) PP_DISPATCH(12); else if (n == 13) PP_DISPATCH(13); else PP_DISPATCH(0);
#undef PP_DISPATCH
    return hipGetLastError();
}

// ---- the seeding over rows sharded across ranks (round 4) -----------------------------------------------------------------------
// `_kmeans_plusplus` walks ALL samples: the running sum the candidates are drawn from runs over the ranks' rows in rank order, and a
// candidate's potential is a sum over all of them.  Every rank keeps the passes over its own rows (pp_round_kernel); what crosses
// the ranks per round are two small tables, both as SUMs of 64-bit words in which one rank writes and the others hold zeros (the
// exchange of the sharded Lloyd loop: edmdc_set_kmeans_allreduce):
//   tot  [world][PP_SROWS]  rank r's sum over its chunks of S[t][.] for every trial t (and of closest itself);
//   crow [PP_LMAX + 1][PP_CW]  the rows of the next round's candidates -- coordinates, |x|^2, GLOBAL index -- each written by the rank
//                           that owns the sample the drawn value falls on; row PP_LMAX = the centre chosen last (rank 0 writes it).
// Every rank then takes the same decisions from the same numbers: potentials = sums of tot over the ranks in rank order, the first
// smallest wins; a drawn value u * pot is located first among the ranks (running sum of tot[.][winner]), then -- by the owner,
// with the value reduced by the ranks before it -- among its chunks and inside the chunk exactly as on one rank.  The running
// sum is thereby grouped by rank and chunk instead of by chunk alone: a candidate can differ from the one-rank run's (and from
// scikit-learn's) only where a drawn value falls within rounding of a running-sum boundary, the caveat the chunked sum carries anyway.
__global__ void __launch_bounds__(PD_THREADS) pp_tot_kernel(int nchunks, int L, const double* __restrict__ S, double* __restrict__ tot_mine) {
    const int l = threadIdx.x & 63;
    for (int t = threadIdx.x >> 6; t <= L; t += PD_THREADS / 64) {      // a wave per row: the trials t < L, then the row of closest itself
        const int row = t < L ? t : PP_LMAX;
        const double a = pp_row_total(S + (int64_t)row * nchunks, nchunks, l);
        if (l == 0) tot_mine[row] = a;
    }
}

// the first centre: its owner writes the row (others: zeros; the table is summed over the ranks afterwards)
template <int NS>
__global__ void pp_first_sh_kernel(int64_t N, int n, long long first_global, long long row0, const double* __restrict__ Xt,
                                   const double* __restrict__ xsq, double* __restrict__ crow) {
    const int j = threadIdx.x;
    if (j >= PP_CW) return;
    const long long li = first_global - row0;
    double v = 0.0;
    if (li >= 0 && li < N) v = j < KM_NMAX ? ((NS > 0 ? j < NS : j < n) ? Xt[(int64_t)j * N + li] : 0.0) : (j == KM_NMAX ? xsq[li] : (double)first_global);
    crow[PP_LMAX * PP_CW + j] = v;
}
__global__ void pp_record_kernel(int n, int c, const double* __restrict__ row, double* __restrict__ C, long long* __restrict__ indices) {
    const int j = threadIdx.x;
    if (j < n) C[(int64_t)c * n + j] = row[j];
    if (j == 0) indices[c] = (long long)row[KM_NMAX + 1];
}

// pp_decide_kernel for a sharded run; one block per trial of the next round (one block when nothing is drawn).
//   crow_cur: the rows of this round (candidates 0..L-1, the previous centre at PP_LMAX) -- identical on every rank
//   crow_nxt: zeroed by the caller; receives this rank's contributions to the next round's rows
template <int NS>
__global__ void __launch_bounds__(PD_THREADS) pp_decide_sh_kernel(int64_t N, int n, int nchunks, int L, int c, int draw, const double* __restrict__ u,
                                                                 const double* __restrict__ Xt, const double* __restrict__ xsq,
                                                                 const double* __restrict__ closest, const double* __restrict__ S,
                                                                 int world, int rank, long long row0, const double* __restrict__ tot,
                                                                 const double* __restrict__ crow_cur, double* __restrict__ crow_nxt,
                                                                 double* __restrict__ C, long long* __restrict__ indices) {
    extern __shared__ double dyn[];               // prefix[nchunks + 1] | vals[PP_CHUNK]
    double* prefix = dyn;
    double* vals = dyn + (nchunks + 1);
    __shared__ double wtot[PD_THREADS / 64];
    __shared__ int wfirst[PD_THREADS / 64];
    __shared__ long long s_found;
    __shared__ double s_pot, s_before;
    __shared__ int s_row, s_owner;
    const int tid = threadIdx.x, trial = blockIdx.x;
    if (tid == 0) {
        int row = PP_LMAX;
        if (c > 0) {
            double bestp = 0.0;
            for (int t = 0; t < L; ++t) {
                double p = 0.0;
                for (int r = 0; r < world; ++r) p += tot[r * PP_SROWS + t];
                if (t == 0 || p < bestp) { bestp = p; row = t; }         // np.argmin: the first smallest
            }
        }
        double pot = 0.0;
        for (int r = 0; r < world; ++r) pot += tot[r * PP_SROWS + row];
        s_row = row;
        s_pot = pot;
        // the rank a drawn value falls on: first rank whose running sum reaches it (the last one when rounding puts it past the end)
        const double v = draw ? u[trial] * pot : 0.0;
        double run = 0.0;
        int owner = world - 1;
        double before = 0.0;
        for (int r = 0; r < world; ++r) {
            const double nxt = run + tot[r * PP_SROWS + row];
            if (nxt >= v) { owner = r; before = run; break; }
            if (r == world - 1) before = run;
            run = nxt;
        }
        s_owner = owner;
        s_before = before;
    }
    __syncthreads();
    const int row = s_row;
    // the centre this round has chosen: the winner's row (c > 0; centre 0 was recorded by pp_record_kernel); it is the next round's
    // "last" row, which rank 0 alone contributes
    const double* win = crow_cur + (c > 0 ? row : PP_LMAX) * PP_CW;
    if (blockIdx.x == 0) {
        if (c > 0) {
            if (tid < n) C[(int64_t)c * n + tid] = win[tid];
            if (tid == 0) indices[c] = (long long)win[KM_NMAX + 1];
        }
        if (draw && rank == 0 && tid < PP_CW) crow_nxt[PP_LMAX * PP_CW + tid] = win[tid];
    }
    if (!draw) return;
    if (rank != s_owner) return;                  // (block-uniform) the table keeps this rank's zeros for the trial
    pp_chunk_prefix(nchunks, S + (int64_t)row * nchunks, prefix, wtot);
    const double v = u[trial] * s_pot - s_before;
    double cl[KM_NMAX];
#pragma unroll
    for (int j = 0; j < KM_NMAX; ++j) cl[j] = win[j];
    pp_find_sample<NS>(N, n, nchunks, v, prefix, c > 0, cl, win[KM_NMAX], Xt, xsq, closest, vals, wtot, wfirst, &s_found);
    __syncthreads();
    const long long li = s_found;
    if (tid < PP_CW)
        crow_nxt[trial * PP_CW + tid] = tid < KM_NMAX ? ((NS > 0 ? tid < NS : tid < n) ? Xt[(int64_t)tid * N + li] : 0.0)
                                                      : (tid == KM_NMAX ? xsq[li] : (double)(row0 + li));
}

size_t kmeanspp_shard_doubles(int world) { return (size_t)world * PP_SROWS + 2 * (size_t)(PP_LMAX + 1) * PP_CW + 16; }

// The sharded seeding loop, stream ordered.  exch(user, device words, count, op): the ranks' exchange (op 0: sum of int64 words,
// op 1: maximum of uint64 words); first: GLOBAL index of the first centre; u: the uniforms every rank draws identically; shard:
// device scratch [kmeanspp_shard_doubles(world)]; indices come back GLOBAL.
hipError_t launch_kmeanspp_sharded(hipStream_t st, int64_t N, int n, int k, int L, const double* X, int64_t xstride, const double* mean,
                                   long long first, const double* u, double* Xt, double* xsq, double* closest, double* S,
                                   void* state, double* C, long long* indices, float* Xf, int world, int rank, long long row0,
                                   double* shard, int (*exch)(void*, void*, int64_t, int), void* user, int* comm_failed, void* rowbuf) {
    if (n > KM_NMAX || L > PP_LMAX || L < 1 || world < 1 || rank < 0 || rank >= world || N < 1) return hipErrorInvalidValue;
    const PPRows rows = pp_rows_from(Xf ? rowbuf : nullptr, N, n);
    const unsigned nrb = (unsigned)(((N + PP_ROW - 1) / PP_ROW + 255) / 256);
    *comm_failed = 0;
    const int nchunks = kmeanspp_chunks(N);
    const size_t lds = ((size_t)nchunks + 1 + PP_CHUNK) * 8;
    if (lds > 150 * 1024 || (nchunks + PD_THREADS - 1) / PD_THREADS > PP_PERMAX) return hipErrorInvalidValue;
    PPState* ps = reinterpret_cast<PPState*>(state);
    double* tot = shard;
    double* crow[2] = {shard + (size_t)world * PP_SROWS, shard + (size_t)world * PP_SROWS + (size_t)(PP_LMAX + 1) * PP_CW};
    const size_t tot_bytes = (size_t)world * PP_SROWS * 8, crow_bytes = (size_t)(PP_LMAX + 1) * PP_CW * 8;
    auto exchange = [&](void* p, int64_t words, int op) -> bool {
        if (world == 1 || !exch) return true;
        if (exch(user, p, words, op) != 0) { *comm_failed = 1; return false; }
        return true;
    };
    hipError_t e0 = hipMemsetAsync(ps, 0, sizeof(PPState), st);
    if (e0 != hipSuccess) return e0;
    const unsigned nb = (unsigned)((N + PP_THREADS - 1) / PP_THREADS);
#define PPS_DISPATCH(NS_) do { \
        hipError_t e_ = hipFuncSetAttribute((const void*)pp_decide_sh_kernel<NS_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e_ != hipSuccess) return e_; \
        hipLaunchKernelGGL(pp_transpose_kernel<NS_>, dim3(nb), dim3(PP_THREADS), 0, st, N, n, X, xstride, mean, Xt, xsq, Xf, ps); \
        if (rows.rc) hipLaunchKernelGGL(pp_rowball_kernel<NS_>, dim3(nrb), dim3(256), 0, st, N, n, Xt, rows); \
        if (Xf && !exchange(&ps->xmax_bits, 1, 1)) return hipSuccess;        /* the scale of the screening margin: over all ranks */ \
        if ((e_ = hipMemsetAsync(crow[1], 0, crow_bytes, st)) != hipSuccess) return e_; \
        hipLaunchKernelGGL(pp_first_sh_kernel<NS_>, dim3(1), dim3(64), 0, st, N, n, first, row0, Xt, xsq, crow[1]); \
        if (!exchange(crow[1], (int64_t)(PP_LMAX + 1) * PP_CW, 0)) return hipSuccess; \
        hipLaunchKernelGGL(pp_record_kernel, dim3(1), dim3(64), 0, st, n, 0, crow[1] + PP_LMAX * PP_CW, C, indices); \
        for (int c = 0; c < k; ++c) { \
            if (k == 1) break; \
            const int draw = c + 1 < k ? 1 : 0; \
            const int Lc = c == 0 ? 0 : L; \
            const double* cur = crow[(c + 1) & 1];                       /* round c reads T[(c - 1) & 1], written by decide(c - 1) (c = 0: pp_first) */ \
            if (Lc <= 8) hipLaunchKernelGGL((pp_round_kernel<NS_, PP_THREADS, 8>), dim3(nchunks), dim3(PP_THREADS), (size_t)(Lc + 1) * PP_NROW * 8, st, N, n, Lc, nchunks, Xt, xsq, ps, 0, \
                               c == 0 ? 1 : (c == 1 ? 0 : 2), closest, S, (c >= PP_SCREEN_FROM) ? Xf : nullptr, cur, rows); \
            else hipLaunchKernelGGL((pp_round_kernel<NS_, PP_THREADS, PP_LMAX>), dim3(nchunks), dim3(PP_THREADS), (size_t)(Lc + 1) * PP_NROW * 8, st, N, n, Lc, nchunks, Xt, xsq, ps, 0, \
                               c == 0 ? 1 : (c == 1 ? 0 : 2), closest, S, (c >= PP_SCREEN_FROM) ? Xf : nullptr, cur, rows); \
            if ((e_ = hipMemsetAsync(tot, 0, tot_bytes, st)) != hipSuccess) return e_; \
            hipLaunchKernelGGL(pp_tot_kernel, dim3(1), dim3(PD_THREADS), 0, st, nchunks, Lc, S, tot + (size_t)rank * PP_SROWS); \
            if (!exchange(tot, (int64_t)world * PP_SROWS, 0)) return hipSuccess; \
            double* nxt = crow[c & 1]; \
            if ((e_ = hipMemsetAsync(nxt, 0, crow_bytes, st)) != hipSuccess) return e_; \
            hipLaunchKernelGGL(pp_decide_sh_kernel<NS_>, dim3(draw ? L : 1), dim3(PD_THREADS), lds, st, N, n, nchunks, Lc, c, draw, u + (size_t)c * L, Xt, xsq, \
                               closest, S, world, rank, row0, tot, cur, nxt, C, indices); \
            if (draw && !exchange(nxt, (int64_t)(PP_LMAX + 1) * PP_CW, 0)) return hipSuccess; \
        } } while (0)
    if (n == 12) PPS_DISPATCH(12); else if (n == 13) PPS_DISPATCH(13); else PPS_DISPATCH(0);
#undef PPS_DISPATCH
    return hipGetLastError();
}

int kmeans_blocks(int64_t N, int n, int k, bool scalar_records);
// epochs of a block's table: a flush every KM_EPOCH_PASSES passes
int kmeans_epochs(int64_t N, int n, int k, bool scalar_records) {
    const int blocks = kmeans_blocks(N, n, k, scalar_records);
    const int64_t passes = (N + (int64_t)blocks * KM_THREADS - 1) / ((int64_t)blocks * KM_THREADS);
    const int64_t ep = (passes + KM_EPOCH_PASSES - 1) / KM_EPOCH_PASSES;
    return (int)(ep > 0 ? ep : 1);
}
size_t kmeans_partial_words(int64_t N, int n, int k, bool scalar_records) {
    return (size_t)kmeans_blocks(N, n, k, scalar_records) * kmeans_epochs(N, n, k, scalar_records) * k * (n + 1);
}
size_t kmeans_red_words(int n, int k) { return (size_t)k * (n + 1) * 2 + 2; }

hipError_t launch_kmeans_c2(hipStream_t st, int n, int k, const double* C, double* c2) {
    hipLaunchKernelGGL(kmeans_c2_kernel, dim3((k + 255) / 256), dim3(256), 0, st, n, k, C, c2);
    return hipGetLastError();
}

// range of the coordinates: rng [16] u64 (maxima as bit patterns; zeroed here).  A sharded run all-reduces rng with MAX before
// launch_kmeans_scale turns it into fix [32] (scales of the fixed-point member sums) and prm [4] (margins of the candidate filter, flags).
hipError_t launch_kmeans_range(hipStream_t st, int64_t N, int n, const double* X, int64_t xstride, const double* mean, unsigned long long* rng) {
    if (n > KM_CMAX) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(rng, 0, 16 * sizeof(unsigned long long), st);
    if (e != hipSuccess) return e;
    const int64_t need = (N + 255) / 256;
    hipLaunchKernelGGL(kmeans_range_kernel, dim3((unsigned)(need < 2048 ? need : 2048)), dim3(256), 0, st, N, n, X, xstride, mean, rng);
    return hipGetLastError();
}
hipError_t launch_kmeans_scale(hipStream_t st, int n, const unsigned long long* rng, double* fix, double* prm) {
    hipLaunchKernelGGL(kmeans_scale_kernel, dim3(1), dim3(64), 0, st, n, rng, fix, prm);
    return hipGetLastError();
}

// one E-step (+ accumulation); c2 = the packed centre table [k][16] that launch_kmeans_c2 / launch_kmeans_average maintain.
// Dc != nullptr: the candidate-filtered form (Dc [k][k] from launch_kmeans_cdist).  prm [4] and fix [32] from launch_kmeans_scale.
// `scalar_records`: the kernel with the centre records in scalar registers (the only one for k > 512 or n = 15); otherwise the
// LDS / DPP kernel.  The block count -- the number of partial sums -- follows the kernel: kmeans_blocks(N, n, k, scalar_records).
static bool kmeans_lds_form(int n, int k, bool scalar_records) { return !scalar_records && n <= KM2_NMAX && k <= KM2_KMAX; }
bool kmeans_reads_through_perm(int n, int k, bool scalar_records) { return kmeans_lds_form(n, k, scalar_records); }
hipError_t launch_kmeans_assign(hipStream_t st, int64_t N, int n, int k, const double* X, int64_t xstride, const double* mean,
                                const double* c2, int* labels, unsigned long long* partial, double* block_inertia, int* block_changed,
                                const float* Dc, const double* prm, const double* fix, float* d2out, bool scalar_records, const int* perm,
                                const unsigned long long* Nk, const float* Pf, const KmBounds* bounds) {
    if (perm && !kmeans_lds_form(n, k, scalar_records)) return hipErrorInvalidValue;      // only the LDS / DPP kernel reads through a permutation
    if (Nk && (!Dc || !kmeans_lds_form(n, k, scalar_records))) return hipErrorInvalidValue;
    if (n > KM_CMAX || (reinterpret_cast<uintptr_t>(c2) & 127) || !prm || !fix) return hipErrorInvalidValue;
    // distance bounds: only where the packed-fp32 path can leave them (sorted loop, n = 12 / 13, LDS / DPP kernel)
    if (bounds && (!Pf || !Nk || !perm || (n != 12 && n != 13) || !bounds->ub || !bounds->lb || N >= ((int64_t)1 << 31))) return hipErrorInvalidValue;
    if (bounds && bounds->use_list && (!bounds->list || !bounds->nlist)) return hipErrorInvalidValue;
    const int blocks = kmeans_blocks(N, n, k, scalar_records);
    const int nep = kmeans_epochs(N, n, k, scalar_records);
    if (kmeans_lds_form(n, k, scalar_records)) {
        const size_t lds2 = (size_t)k * (16 + n + 1) * sizeof(double) + (KM_THREADS / 64) * KM2_LIST * sizeof(unsigned short);   // 133 KB at k = 512, n = 12
        const bool use_list = bounds && bounds->use_list;
        float* ubo = bounds ? bounds->ub : nullptr;
        float* lbo = bounds ? bounds->lb : nullptr;
        const double beta = bounds ? (bounds->beta >= 0.0 ? bounds->beta : KM_BND_BETA) : 0.0;
        const double tscale = (1.0 + beta) * (1.0 + beta);
#define KM2_LAUNCH(NS_, LIST_) do { \
        hipError_t e_ = hipFuncSetAttribute((const void*)kmeans_assign_lds_kernel<NS_, LIST_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); \
        if (e_ != hipSuccess) return e_; \
        hipLaunchKernelGGL((kmeans_assign_lds_kernel<NS_, LIST_>), dim3(blocks), dim3(KM_THREADS), lds2, st, N, n, k, X, xstride, mean, c2, labels, partial, nep, \
                           block_inertia, block_changed, Dc, prm, d2out, perm, fix, Nk, Nk ? Pf : nullptr, \
                           use_list ? bounds->list : nullptr, use_list ? bounds->nlist : nullptr, ubo, lbo, tscale, \
                           (long long)((KM_LIST_DYNAMIC && use_list && bounds->rw2) ? kmeans_bounds_list_words(N) : 0)); } while (0)
        if (use_list) { if (n == 12) KM2_LAUNCH(12, true); else KM2_LAUNCH(13, true); }
        else if (n == 12) KM2_LAUNCH(12, false); else if (n == 13) KM2_LAUNCH(13, false); else KM2_LAUNCH(0, false);
#undef KM2_LAUNCH
        return hipGetLastError();
    }
    const size_t lds = (size_t)k * (n + 1) * sizeof(double) ;
    if (lds > 150 * 1024) return hipErrorInvalidValue;
#define KM_LAUNCH(NS_, PR_) do { \
        hipError_t e_ = hipFuncSetAttribute((const void*)kmeans_assign_kernel<NS_, PR_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e_ != hipSuccess) return e_; \
        hipLaunchKernelGGL((kmeans_assign_kernel<NS_, PR_>), dim3(blocks), dim3(KM_THREADS), lds, st, N, n, k, X, xstride, mean, c2, labels, partial, nep, \
                           block_inertia, block_changed, Dc, prm, d2out, fix); } while (0)
    if (Dc) { if (n == 12) KM_LAUNCH(12, true); else if (n == 13) KM_LAUNCH(13, true); else KM_LAUNCH(0, true); }
    else { if (n == 12) KM_LAUNCH(12, false); else if (n == 13) KM_LAUNCH(13, false); else KM_LAUNCH(0, false); }
#undef KM_LAUNCH
    return hipGetLastError();
}
// the packed-fp32 form of the E-step (sorted order, single-reference filter): its own block geometry
constexpr int KM_PK_KMAX = 1024;      // labels in KM_SORT_LABEL_BITS bits, centre indices in the 16 low bits of a key
static_assert(KM_PK_KMAX <= KM_SORT_LABEL_MAX, "the sort keys of the loop's sample order hold the label in KM_SORT_LABEL_BITS bits");
// 512-thread blocks, three per CU, while their member sums fit three times (k <= 512 at n = 12 / 13); 1024-thread blocks, one per CU, above
static bool kmeans_pk_small(int n, int k) { return (size_t)k * (n + 1) * 8 <= 53 * 1024; }
int kmeans_pk_threads(int n, int k) { return kmeans_pk_small(n, k) ? PK_THREADS : 2 * PK_THREADS; }
int kmeans_pk_blocks(int64_t N, int n, int k) {
    const int th = kmeans_pk_threads(n, k), cap = kmeans_pk_small(n, k) ? 768 : 256;
    const int64_t need = (N + th - 1) / th;
    return need < cap ? (int)(need > 0 ? need : 1) : cap;
}
int kmeans_pk_epochs(int64_t N, int n, int k) {
    const int blocks = kmeans_pk_blocks(N, n, k), th = kmeans_pk_threads(n, k);
    const int64_t passes = (N + (int64_t)blocks * th - 1) / ((int64_t)blocks * th);
    const int64_t ep = (passes + KM_EPOCH_PASSES - 1) / KM_EPOCH_PASSES;
    return (int)(ep > 0 ? ep : 1);
}
bool kmeans_pk_supported(int n, int k) { return (n == 12 || n == 13) && k >= 64 && k <= KM_PK_KMAX && (size_t)k * (n + 1) * 8 <= 150 * 1024; }
hipError_t launch_kmeans_assign_pk(hipStream_t st, int64_t N, int n, int k, const double* X, int64_t xstride, const double* mean, const double* c2,
                                   int* labels, unsigned long long* partial, double* block_inertia, int* block_changed, const double* prm,
                                   const double* fix, float* d2out, const int* perm, const unsigned long long* Nk, const float* Pf) {
    if (!kmeans_pk_supported(n, k) || !Nk || !Pf || !prm || !fix || (reinterpret_cast<uintptr_t>(c2) & 127)) return hipErrorInvalidValue;
    const int blocks = kmeans_pk_blocks(N, n, k), nep = kmeans_pk_epochs(N, n, k);
    const size_t lds = (size_t)k * (n + 1) * sizeof(double);
#define PK_LAUNCH(NS_, TH_) do { \
        hipError_t e_ = hipFuncSetAttribute((const void*)kmeans_assign_pk_kernel<NS_, TH_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e_ != hipSuccess) return e_; \
        hipLaunchKernelGGL((kmeans_assign_pk_kernel<NS_, TH_>), dim3(blocks), dim3(TH_), lds, st, N, n, k, X, xstride, mean, c2, labels, partial, nep, \
                           block_inertia, block_changed, prm, d2out, perm, fix, Nk, Pf); } while (0)
    if (kmeans_pk_small(n, k)) { if (n == 12) PK_LAUNCH(12, PK_THREADS); else PK_LAUNCH(13, PK_THREADS); }
    else { if (n == 12) PK_LAUNCH(12, 2 * PK_THREADS); else PK_LAUNCH(13, 2 * PK_THREADS); }
#undef PK_LAUNCH
    return hipGetLastError();
}
hipError_t launch_kmeans_cdist(hipStream_t st, int n, int k, const double* c2, float* Dc, unsigned long long* Nk, float* Pf, const float* shiftc,
                               float* mvd, float* rw2, int pf_pairs, const KmMstepArgs* tail) {
    if (Pf && (!Nk || n > KM_PK_NMAX)) return hipErrorInvalidValue;
    if (mvd && !shiftc) return hipErrorInvalidValue;
    if (tail && tail->k != k) return hipErrorInvalidValue;
    KmMstep ta{};
    if (tail) ta = km_mstep_args(*tail);
    hipLaunchKernelGGL(kmeans_cdist_kernel, dim3(k + (tail ? 1 : 0)), dim3(256), 0, st, n, k, c2, Dc, Nk, Pf, shiftc, mvd, rw2, pf_pairs, ta, tail ? 1 : 0);
    return hipGetLastError();
}
int kmeans_bounds_tail() { return KM_BND_TAIL; }
// The prefix contract of Pf (round-5 review): launch_kmeans_cdist builds only this many pair records per row when the LDS / DPP kernel is
// their only reader, and that kernel's screening loop walks npairs2 = ((ceil(ncand / 2) + 1) & ~1) records with ncand <= KM2_NBR_MAX.
static_assert((((((KM2_NBR_MAX + 1) >> 1) + 1) & ~1)) <= KM2_NBR_MAX / 2, "the screening loop would read pair records that were never built");
int kmeans_lds_pf_pairs() { return KM2_NBR_MAX / 2; }
size_t kmeans_bounds_list_words(int64_t N) { return (size_t)((N + KM_BND_TILE - 1) / KM_BND_TILE) * (KM_BND_TILE + 64); }
hipError_t launch_kmeans_bounds(hipStream_t st, int64_t N, int k, const int* labels, const KmBounds& b, const double* prm) {
    if (!b.ub || !b.lb || !b.shiftc || !b.mvd || !b.list || !b.nlist || N >= ((int64_t)1 << 31)) return hipErrorInvalidValue;
    if (kmeans_bounds_list_words(N) >= ((size_t)1 << 31)) return hipErrorInvalidValue;      // (offsets into the list are 32-bit)
    // (b.nlist was zeroed by the M-step's launch_kmeans_average)
    hipLaunchKernelGGL(kmeans_bounds_kernel, dim3((unsigned)((N + KM_BND_TILE - 1) / KM_BND_TILE)), dim3(KM_BND_BT), 0, st, N, k, labels, b.ub, b.lb, b.shiftc,
                       b.mvd, prm, b.list, b.nlist, KM_LIST_DYNAMIC ? b.rw2 : nullptr, (long long)kmeans_bounds_list_words(N));
    return hipGetLastError();
}
int kmeans_blocks(int64_t N, int n, int k, bool scalar_records) {
    const int64_t need = (N + KM_THREADS - 1) / KM_THREADS;
    const int cap = kmeans_lds_form(n, k, scalar_records) ? KM2_BLOCKS : KM_BLOCKS;
    return need < cap ? (int)(need > 0 ? need : 1) : cap;
}
// M-step, second half: red -> Cnew [k][n], c2 (packed table), stats[0] = squared shift against Cold, [2] = changed labels, [3] = empty clusters,
// prm[2] (non-finite centre), prm[3] (hold); mode: see kmeans_average_kernel
hipError_t launch_kmeans_average(hipStream_t st, int n, int k, const long long* red, const double* fix, const double* Cold, double* Cnew,
                                 double* c2, double* stats, double* prm, int mode, float* shiftc, int* nlist) {
    hipLaunchKernelGGL(kmeans_average_kernel, dim3(1), dim3(1024), 0, st, n, k, red, fix, Cold, Cnew, c2, stats, prm, mode, shiftc, nlist);
    return hipGetLastError();
}
hipError_t launch_kmeans_reloc_dist(hipStream_t st, int64_t N, int n, const double* X, int64_t xstride, const double* mean, const double* Cold,
                                    const int* labels, const int* perm, double* dist_row, int* lab_row) {
    hipLaunchKernelGGL(kmeans_reloc_dist_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, N, n, X, xstride, mean, Cold, labels, perm,
                       dist_row, lab_row);
    return hipGetLastError();
}
}  // namespace brov
