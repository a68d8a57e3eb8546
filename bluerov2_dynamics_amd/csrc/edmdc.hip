// edmdc.hip -- Koopman EDMDc on gfx950: RBF lift (K4) and the G^T[G|Y] normal-equation
// blocks as an fp64 MFMA split-K reduction (K5).  Reference: Koopman/koopmanEDMDc.py:41-48
// (_rbf_mat), :221-236 (_lift), :89-97 / :129-147 (fit / fit_multi normal equations).
//
// Data flow per chunk of state rows (chunk sized by the host, default 2^20 rows):
//   lift_rows_kernel : X,U  ->  Zrows[row][W]   (W = kp + tailp doubles, e.g. 544)
//                               row = [ rbf_0..rbf_{kp-1} | x_0..x_{n-1} u_0..u_{r-1} x+_0..x+_{n-1} 0.. ]   (x+ = next state, when it fits)
//                               wrow[row] = 1 if (row, row+1) is a pair inside one bag else 0
//   gram_kernel      : partial[task][slab] += sum_{row in slab} w[row] Z[row]^T [ Z[row] | Z[row+1] ]
//   gram_finish      : fixed-order sum over slabs, scatter to reference feature order.
// phi(x_{t+1}) of pair t is phi(x_t) of pair t+1, so every state is lifted once and the Y
// operand of the Gram is simply the next row of Zrows.
//
// gram_kernel: one wave = one task = a 4x6 block of 16x16 output tiles (64x96 outputs,
// 96 fp64 accumulators per lane = 192 VGPRs; the whole kernel stays under 256 VGPRs so the
// MFMAs use their VGPR form -- with more accumulators hipcc shuttles them between AGPRs and
// VGPRs every iteration) over one K-slab of rows.  v_mfma_f64_16x16x4_f64 consumes 4 rows per
// instruction; per 4-row step a wave issues 10 operand loads (8 B per lane, 4 rows x 128 B per
// wave-load) for 24 MFMAs, so operands are read once per wave and all reuse is in registers.
// Two waves share a SIMD and cover each other's load latency.  Waves of one K-slab are placed on one XCD (blockIdx % 8) so the
// slab's rows are fetched from HBM once and shared through that XCD's L2.
#include <vector>
#include "brov2_kernels.h"

namespace brov {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int LIFT_NMAX = 16;   // max state dimension held in registers per centre
constexpr int GRAM_TA = 4;      // A tiles (rows of the output block) per task
constexpr int GRAM_TB = 6;      // B tiles (cols of the output block) per task

// ---------------------------------------------------------------------------------------
// K4: lift
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) center_norms_kernel(int n, int k, const double* __restrict__ C, double* __restrict__ c2) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= k) return;
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += C[c * n + j] * C[c * n + j];   // np.sum(C**2, axis=1)
    c2[c] = s;
}
hipError_t launch_center_norms(hipStream_t st, int n, int k, const double* C, double* c2) {
    hipLaunchKernelGGL(center_norms_kernel, dim3((k + 255) / 256), dim3(256), 0, st, n, k, C, c2);
    return hipGetLastError();
}

// exp(x): k = rint(x / ln 2), r = x - k ln2 (two-constant Cody-Waite), degree-13 Taylor on |r| <= ln2/2
// (truncation 4e-18), scaled by 2^k with v_ldexp_f64 (underflows to 0 like exp, NaN propagates).
// ~20 instructions against ~35 for the OCML routine; <= 1 ulp on the range the RBF lift uses (x <= ~0).
__device__ __forceinline__ double exp_fast(double x) {
    const double kf = rint(x * 1.44269504088896338700e+00);
    double r = fma(-kf, 6.93147180369123816490e-01, x);      // ln2_hi
    r = fma(-kf, 1.90821492927058770002e-10, r);             // ln2_lo
    double p = fma(r, 1.6059043836821614599e-10, 2.0876756987868098979e-09);   // 1/13!, 1/12!
    p = fma(r, p, 2.5052108385441718775e-08);    // 1/11!
    p = fma(r, p, 2.7557319223985890653e-07);    // 1/10!
    p = fma(r, p, 2.7557319223985892511e-06);    // 1/9!
    p = fma(r, p, 2.4801587301587301566e-05);    // 1/8!
    p = fma(r, p, 1.9841269841269841253e-04);    // 1/7!
    p = fma(r, p, 1.3888888888888889419e-03);    // 1/6!
    p = fma(r, p, 8.3333333333333332177e-03);    // 1/5!
    p = fma(r, p, 4.1666666666666664354e-02);    // 1/4!
    p = fma(r, p, 1.6666666666666665741e-01);    // 1/3!
    p = fma(r, p, 0.5);
    p = fma(r, p, 1.0);
    p = fma(r, p, 1.0);
    double kc = fmin(fmax(kf, -2200.0), 2200.0);             // keep the int conversion in range; ldexp saturates
    return ldexp(p, (int)kc);
}

// rbf value for the centre held by this lane, state row read through wave-uniform (scalar) loads.
// NS > 0: compile-time state dimension (straight-line code, merged scalar loads); NS = 0: runtime n.
// x2 = |x|^2 is the same for every lane; it is computed once per row and shared by the NC centres of a lane.
template <int NS, int NC>
__device__ __forceinline__ void rbf_row(int n, double gamma, const double* __restrict__ xrow, const double (*c)[LIFT_NMAX],
                                        const double* c2, double* out) {
    double x2 = 0.0, dot[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) dot[q] = 0.0;
    if constexpr (NS > 0) {
        double xr[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) xr[j] = xrow[j];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            x2 = fma(xr[j], xr[j], x2);
#pragma unroll
            for (int q = 0; q < NC; ++q) dot[q] = fma(xr[j], c[q][j], dot[q]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < LIFT_NMAX; ++j) {
            if (j < n) {
                const double xj = xrow[j];
                x2 = fma(xj, xj, x2);
#pragma unroll
                for (int q = 0; q < NC; ++q) dot[q] = fma(xj, c[q][j], dot[q]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NC; ++q) out[q] = exp_fast(-gamma * ((x2 + c2[q]) - 2.0 * dot[q]));   // Koopman/koopmanEDMDc.py:46-48
}
// the same for a state row already in registers (wave-uniform values: scalar registers)
template <int NS, int NC>
__device__ __forceinline__ void rbf_vals(double gamma, const double* xr, const double (*c)[LIFT_NMAX], const double* c2, double* out) {
    double x2 = 0.0, dot[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) dot[q] = 0.0;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        x2 = fma(xr[j], xr[j], x2);
#pragma unroll
        for (int q = 0; q < NC; ++q) dot[q] = fma(xr[j], c[q][j], dot[q]);
    }
#pragma unroll
    for (int q = 0; q < NC; ++q) out[q] = exp_fast(-gamma * ((x2 + c2[q]) - 2.0 * dot[q]));   // Koopman/koopmanEDMDc.py:46-48
}
template <int NS>
__device__ __forceinline__ double rbf_one(int n, double gamma, const double* __restrict__ xrow, const double* c, double c2) {
    double o;
    rbf_row<NS, 1>(n, gamma, xrow, reinterpret_cast<const double (*)[LIFT_NMAX]>(c), &c2, &o);
    return o;
}

// Reference-order lift: Z[N][n+k] = [x, rbf].  Block = 256 lanes = 256 centres, tile of 64 rows.
template <int NS>
__global__ void __launch_bounds__(256) lift_ref_kernel(int64_t N, int n, int k, double gamma, const double* __restrict__ X,
                                                       const double* __restrict__ C, double* __restrict__ Z) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int d = n + k;
    double cc[LIFT_NMAX], c2 = 0.0;
#pragma unroll
    for (int j = 0; j < LIFT_NMAX; ++j) { cc[j] = (j < n && c < k) ? C[(int64_t)c * n + j] : 0.0; c2 = fma(cc[j], cc[j], c2); }
    for (int64_t r = r0; r < r0 + 64 && r < N; ++r) {
        const double* xrow = X + r * n;
        if (c < k) Z[r * d + n + c] = rbf_one<NS>(n, gamma, xrow, cc, c2);
        if (blockIdx.y == 0 && (int)threadIdx.x < n) Z[r * d + threadIdx.x] = xrow[threadIdx.x];
    }
}
hipError_t launch_lift_ref(hipStream_t st, int64_t N, int n, int k, double gamma, const double* X, const double* C, double* Z) {
    if (N <= 0) return hipSuccess;
    if (n > LIFT_NMAX) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((N + 63) / 64), (unsigned)((k + 255) / 256));
    if (n == 12) hipLaunchKernelGGL(lift_ref_kernel<12>, grid, dim3(256), 0, st, N, n, k, gamma, X, C, Z);
    else if (n == 13) hipLaunchKernelGGL(lift_ref_kernel<13>, grid, dim3(256), 0, st, N, n, k, gamma, X, C, Z);
    else hipLaunchKernelGGL(lift_ref_kernel<0>, grid, dim3(256), 0, st, N, n, k, gamma, X, C, Z);
    return hipGetLastError();
}

// ---- multistep_rmse by linearity (opt-in, round 6) ---------------------------------------------------------------------------------
// The H-step prediction of KoopmanEDMDc.multistep_rmse (Koopman/koopmanEDMDc.py:172-200) is linear in the lifted start state and in the
// inputs of the window: with E = the first n rows of the identity,
//     x_hat[w] = (E A^H) phi(x_w) + sum_{t < H} (E A^(H-1-t) B) u_{w+t}.
// The host supplies RHt [d][n] = (E A^H)^T and Gt [H][r][n] = the transposed input coefficients (H small n x d x d products); one thread
// per window then needs k RBF values, (n + k) n + H r n fused multiply-adds and no H-step recurrence: 2.4 TFLOP of the default path's
// H GEMMs (propagate.hip) become ~1.5 GFLOP.  The coefficient rows and the centres are the same for every thread: scalar loads.
// Not the default: the powers of A are formed explicitly, which the reference never does -- results agree to rounding amplified by
// |A^j| (tested: <= 1e-9 in the RMSE at H = 1 / 10 / 100 on the fixtures).
template <int NS>
__global__ void __launch_bounds__(256) linear_window_kernel(int64_t nw, int n, int r, int k, int64_t H, double gamma,
                                                            const double* __restrict__ X, const double* __restrict__ U,
                                                            const double* __restrict__ C, const double* __restrict__ RHt,
                                                            const double* __restrict__ Gt, double* __restrict__ se, double* __restrict__ xhat) {
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= nw) return;
    const double* xrow = X + w * n;
    double acc[LIFT_NMAX];
#pragma unroll
    for (int i = 0; i < LIFT_NMAX; ++i) acc[i] = 0.0;
    for (int j = 0; j < n; ++j) {                          // the state part of phi
        const double xj = xrow[j];
#pragma unroll
        for (int i = 0; i < LIFT_NMAX; ++i) if (i < n) acc[i] = fma(RHt[(int64_t)j * n + i], xj, acc[i]);
    }
    for (int c = 0; c < k; ++c) {                          // the RBF part, one centre at a time (wave-uniform rows)
        double cc[LIFT_NMAX], c2 = 0.0;
#pragma unroll
        for (int j = 0; j < LIFT_NMAX; ++j) { cc[j] = j < n ? C[(int64_t)c * n + j] : 0.0; c2 = fma(cc[j], cc[j], c2); }
        const double v = rbf_one<NS>(n, gamma, xrow, cc, c2);
        const double* row = RHt + (int64_t)(n + c) * n;
#pragma unroll
        for (int i = 0; i < LIFT_NMAX; ++i) if (i < n) acc[i] = fma(row[i], v, acc[i]);
    }
    for (int64_t t = 0; t < H; ++t) {                      // the inputs of the window
        const double* u = U + (w + t) * r;
        const double* g = Gt + t * r * n;
        for (int j = 0; j < r; ++j) {
            const double uj = u[j];
#pragma unroll
            for (int i = 0; i < LIFT_NMAX; ++i) if (i < n) acc[i] = fma(g[j * n + i], uj, acc[i]);
        }
    }
    const double* xe = X + (w + H) * n;
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < LIFT_NMAX; ++i)
        if (i < n) {
            const double e = xe[i] - acc[i];
            s = fma(e, e, s);
            if (xhat) xhat[w * n + i] = acc[i];
        }
    se[w] = s;
}
hipError_t launch_linear_windows(hipStream_t st, int64_t nw, int n, int r, int k, int64_t H, double gamma, const double* X, const double* U,
                                 const double* C, const double* RHt, const double* Gt, double* se, double* xhat) {
    if (nw <= 0) return hipSuccess;
    if (n > LIFT_NMAX) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((nw + 255) / 256));
    if (n == 12) hipLaunchKernelGGL(linear_window_kernel<12>, grid, dim3(256), 0, st, nw, n, r, k, H, gamma, X, U, C, RHt, Gt, se, xhat);
    else if (n == 13) hipLaunchKernelGGL(linear_window_kernel<13>, grid, dim3(256), 0, st, nw, n, r, k, H, gamma, X, U, C, RHt, Gt, se, xhat);
    else hipLaunchKernelGGL(linear_window_kernel<0>, grid, dim3(256), 0, st, nw, n, r, k, H, gamma, X, U, C, RHt, Gt, se, xhat);
    return hipGetLastError();
}

// Device-native lifted rows.  grid.x = row tiles of 64, grid.y = groups of 512 centres (the tail columns: lift_tail_kernel);
// 128-thread blocks.  A lane owns FOUR adjacent centres (their 4 x 12 coordinates in VGPRs): |x|^2 and the row's state are
// paid once per four RBF values (40 instructions per value instead of 65 with two centres per lane), and a lane's
// results are 32 contiguous bytes (two 16-byte stores, 2 KiB contiguous per wave).  History of the loop "fetch a row's state,
// lift it, store it" per 2^20-row chunk (180 VALU instructions per row and wave = 0.8 ms of issue slots, the stores alone
// 0.8 ms at 5.4 TB/s, tools/attic/store_probe.hip): state through scalar loads 1.65 ms; the next row's state fetched while the
// current one is lifted, tail blocks four rows at a time 1.48 ms; the block's 64 state rows staged in LDS once 1.26 ms; the
// tail columns in a kernel of their own 1.07 + 0.09 ms.
#ifndef LIFT_NC_
#define LIFT_NC_ 4
#endif
constexpr int LIFT_NC = LIFT_NC_;              // centres per lane (even)
constexpr int LIFT_BLOCK = 512 / LIFT_NC;      // threads per block: 512 centres per block row
template <int NS>
__global__ void __launch_bounds__(LIFT_BLOCK) lift_rows_kernel(EdmdcShape s, double gamma, const double* __restrict__ C,
                                                        int64_t row0, int64_t rows, int64_t total_rows, int64_t L, int64_t xs_, int64_t us,
                                                        const double* __restrict__ X, const double* __restrict__ U,
                                                        double* __restrict__ Zrows, double* __restrict__ wrow) {
    const int n = s.n, k = s.k, W = s.width;
    constexpr int RT = 64;                             // rows per block
    constexpr int NC = LIFT_NC;
    const int64_t l0 = (int64_t)blockIdx.x * RT;       // local row index within the chunk buffer
    const int ngroups = (s.kp + LIFT_BLOCK * NC - 1) / (LIFT_BLOCK * NC);
    // (bag, step) of the first row: one 64-bit division per block, then incremental (a division per row
    // costs ~130 scalar instructions, three times the useful work of a row)
    const int64_t g0 = row0 + l0;
    int64_t bag = g0 / xs_, t = g0 - bag * xs_;
    const int64_t lend = (l0 + RT < rows) ? l0 + RT : rows;
    {
        const int c0 = (blockIdx.y * LIFT_BLOCK + threadIdx.x) * NC;       // first of this lane's four centres
        double cc[NC][LIFT_NMAX], c2[NC];
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const int c = c0 + q;
            c2[q] = 0.0;
#pragma unroll
            for (int j = 0; j < LIFT_NMAX; ++j) { cc[q][j] = (j < n && c < k) ? C[(int64_t)c * n + j] : 0.0; c2[q] = fma(cc[q][j], cc[q][j], c2[q]); }
        }
        const double* xp = X + (g0 < total_rows ? g0 : total_rows - 1) * n;
        double* zp = Zrows + l0 * W + c0;
        const bool store = c0 < s.kp;                  // kp is a multiple of 16: a lane's four columns are inside or outside together
        if constexpr (NS > 0) {
            // The block's 64 state rows are staged in LDS once (one exposed global-load latency per block) and every row is then
            // read back as broadcast ds_reads: nothing in the row loop waits on HBM any more.  With the rows arriving through
            // scalar loads -- even fetched one row ahead -- the loop waited on them 46 % of the time (SMEM can only be waited
            // for with lgkmcnt(0), so the fetch distance cannot exceed one row, and a miss under the kernel's own 3.5 TB/s of
            // stores takes longer than the 0.4 us a row's arithmetic lasts).
            __shared__ __attribute__((aligned(16))) double xs[RT][LIFT_NMAX];
            const bool partial_wave = __builtin_amdgcn_ballot_w64(c0 + NC > k) != 0;       // some lane of the wave holds padding centres
            for (int e = threadIdx.x; e < RT * NS; e += LIFT_BLOCK) {
                const int rr = e / NS, j = e - rr * NS;
                const int64_t g = g0 + rr;
                xs[rr][j] = X[(g < total_rows ? g : total_rows - 1) * n + j];      // clamp: rows past the end are masked
            }
            __syncthreads();
#pragma unroll 1
            for (int64_t l = l0; l < lend; ++l) {
                const int64_t g = row0 + l;
                double xr[NS], z[NC];
                const double2* xl = reinterpret_cast<const double2*>(xs[l - l0]);
#pragma unroll
                for (int j = 0; j + 1 < NS; j += 2) { const double2 v = xl[j / 2]; xr[j] = v.x; xr[j + 1] = v.y; }
                if constexpr (NS & 1) xr[NS - 1] = xs[l - l0][NS - 1];
                rbf_vals<NS, NC>(gamma, xr, cc, c2, z);
                const bool valid = g < total_rows && t <= L;                     // wave-uniform
                // rows past the end / gap rows and the padding centres (k <= c < kp) are written as zeros; both are rare
                // and wave-uniform tests (as per-value selects they were 8 of the row's 180 VALU instructions)
                if (!valid || partial_wave) {
                    asm volatile("; lift: zero rows / padding centres, rare" ::: "memory");      // keeps the block behind its branch
#pragma unroll
                    for (int q = 0; q < NC; ++q) z[q] = (valid && c0 + q < k) ? z[q] : 0.0;
                }
                if (store) {
#pragma unroll
                    for (int q = 0; q < NC; q += 2) *reinterpret_cast<double2*>(zp + q) = make_double2(z[q], z[q + 1]);
                }
                zp += W;
                if (++t == xs_) { t = 0; ++bag; }
            }
        } else {
#pragma unroll 1
            for (int64_t l = l0; l < lend; ++l) {
                const int64_t g = row0 + l;
                double z[NC];
                rbf_row<NS, NC>(n, gamma, xp, cc, c2, z);
                const bool valid = g < total_rows && t <= L;                     // wave-uniform
#pragma unroll
                for (int q = 0; q < NC; ++q) z[q] = (valid && c0 + q < k) ? z[q] : 0.0;
                if (store) {
#pragma unroll
                    for (int q = 0; q < NC; q += 2) *reinterpret_cast<double2*>(zp + q) = make_double2(z[q], z[q + 1]);
                }
                zp += W;
                if (g + 1 < total_rows) xp += n;                                 // clamp: rows past the end are masked
                if (++t == xs_) { t = 0; ++bag; }
            }
        }
    }
}

// tail columns [x | u | x_next (xplus shapes, pair rows only) | 0] and the pair weight, as a kernel of its own: inside
// lift_rows_kernel the tail blocks held wave slots at that kernel's 144 VGPRs while they waited on their loads (a quarter of
// the slot time of the launch for 6 % of its bytes).  A thread = one (row, column) of a group of LIFT_TAIL_BLOCK / tailp rows.
constexpr int LIFT_TAIL_BLOCK = 256;
__global__ void __launch_bounds__(LIFT_TAIL_BLOCK) lift_tail_kernel(EdmdcShape s, int64_t row0, int64_t rows, int64_t total_rows, int64_t L,
                                                                   int64_t xs_, int64_t us, const double* __restrict__ X,
                                                                   const double* __restrict__ U, double* __restrict__ Zrows,
                                                                   double* __restrict__ wrow, const unsigned char* __restrict__ pf) {
    // pf != nullptr: ragged bags (fit_multi's trajectory list, Koopman/koopmanEDMDc.py:129-138).  X and U are the bags' rows one after
    // the other, U row-aligned with X, pf[g] = 1 when rows (g, g + 1) are a pair of one bag; the host passes xs_ > total_rows and
    // L = xs_ - 1, so that every row below total_rows is a state row of "bag 0".
    const int n = s.n, W = s.width;
    constexpr int RT = 64;                             // rows per block
    const int64_t l0 = (int64_t)blockIdx.x * RT;
    const int64_t g0 = row0 + l0;
    const int64_t bag = g0 / xs_, t = g0 - bag * xs_;  // one 64-bit division per block
    const int64_t lend = (l0 + RT < rows) ? l0 + RT : rows;
    const int tp = s.tailp, rpp = tp <= LIFT_TAIL_BLOCK ? LIFT_TAIL_BLOCK / tp : 1;     // rows per pass
    const int jr = threadIdx.x / tp, j = threadIdx.x - jr * tp;
    for (int64_t lb = l0; lb < lend; lb += rpp) {
        const int64_t l = lb + jr;
        if (jr < rpp && l < lend) {
            const int64_t g = row0 + l;
            // (bag, step) of this row from the block's first row: at most a few wraps per 64 rows
            int64_t tt = t + (l - l0), bb = bag;
            while (tt >= xs_) { tt -= xs_; ++bb; }
            const bool pair = pf ? (g + 1 < total_rows && pf[g] != 0) : (tt < L);     // rows (g, g + 1) are x_t, x_{t+1} of one bag
            for (int jj = j; jj < tp; jj += (tp <= LIFT_TAIL_BLOCK ? tp : LIFT_TAIL_BLOCK)) {
                double v = 0.0;
                if (g < total_rows && tt <= L) {
                    if (jj < n) v = X[g * n + jj];
                    else if (jj < n + s.r) { if (pair) v = U[(pf ? g : bb * us + tt) * s.r + (jj - n)]; }
                    else if (s.xplus && jj < n + s.r + n && pair && g + 1 < total_rows) v = X[(g + 1) * n + (jj - n - s.r)];
                }
                Zrows[l * W + s.kp + jj] = v;
            }
            if (j == 0) wrow[l] = (g < total_rows && pair) ? 1.0 : 0.0;
        }
    }
}
hipError_t launch_lift_rows_total(hipStream_t st, const EdmdcShape& s, double gamma, const double* C,
                                  int64_t row0, int64_t rows, int64_t total_rows, int64_t L, int64_t xs, int64_t us,
                                  const double* X, const double* U, double* Zrows, double* wrow, const unsigned char* pairflag) {
    if (rows <= 0) return hipSuccess;
    if (s.n > LIFT_NMAX || s.tailp > 256 || xs < 2 || total_rows < 1) return hipErrorInvalidValue;
    const int ngroups = (s.kp + LIFT_BLOCK * LIFT_NC - 1) / (LIFT_BLOCK * LIFT_NC);
    const dim3 grid((unsigned)((rows + 63) / 64), (unsigned)ngroups);
    hipLaunchKernelGGL(lift_tail_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(LIFT_TAIL_BLOCK), 0, st, s, row0, rows, total_rows, L, xs, us, X, U, Zrows, wrow, pairflag);
    if (s.n == 12) hipLaunchKernelGGL(lift_rows_kernel<12>, grid, dim3(LIFT_BLOCK), 0, st, s, gamma, C, row0, rows, total_rows, L, xs, us, X, U, Zrows, wrow);
    else if (s.n == 13) hipLaunchKernelGGL(lift_rows_kernel<13>, grid, dim3(LIFT_BLOCK), 0, st, s, gamma, C, row0, rows, total_rows, L, xs, us, X, U, Zrows, wrow);
    else hipLaunchKernelGGL(lift_rows_kernel<0>, grid, dim3(LIFT_BLOCK), 0, st, s, gamma, C, row0, rows, total_rows, L, xs, us, X, U, Zrows, wrow);
    return hipGetLastError();
}

// Ragged bags (fit_multi's X_list / U_list, Koopman/koopmanEDMDc.py:129-138: "if len(X) < 2: continue", Z = lift(X[:-1]),
// Zp = lift(X[1:]), U[:-1]): one byte per row, 1 = (g, g + 1) is a pair.  Rows start as 1 (memset by the launcher); the last row of
// every non-empty bag is cleared.  A bag of one row is its own last row: no pair starts or ends there.
__global__ void __launch_bounds__(256) bag_pairflags_kernel(int64_t nbags, const int64_t* __restrict__ off, int64_t total_rows,
                                                            unsigned char* __restrict__ pf) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbags) return;
    const int64_t e = off[b + 1];
    if (e > off[b] && e >= 1 && e <= total_rows) pf[e - 1] = 0;
}
hipError_t launch_bag_pairflags(hipStream_t st, int64_t nbags, const int64_t* d_offsets, int64_t total_rows, unsigned char* pairflag) {
    if (total_rows <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(pairflag, 1, (size_t)total_rows, st);
    if (e != hipSuccess) return e;
    if (nbags > 0)
        hipLaunchKernelGGL(bag_pairflags_kernel, dim3((unsigned)((nbags + 255) / 256)), dim3(256), 0, st, nbags, d_offsets, total_rows, pairflag);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// K5: Gram
// ---------------------------------------------------------------------------------------
// One task = GRAM_TA A-tiles x GRAM_TB B-tiles of 16 features.  A tile code: t >= 0 -> columns [16 t, 16 t + 16) of row t of
// the K-step ("G tile": phi(x_t), x_t, u_t); (t | 0x10000) -> the same columns of row t+1 ("Y tile": phi(x_{t+1})); -1 = empty
// slot.  The A operand carries the pair weight.  `want` bit (a * GRAM_TB + b) marks the tile products the finish kernel
// keeps: every needed product is computed by exactly one task.
struct GramTask { int a[GRAM_TA]; int b[GRAM_TB]; unsigned want; };
constexpr unsigned GRAM_SWAP = 1u << 31;     // (mode 1) transposed block: A tiles are cut from the lifted rows, B tiles from the rows of W

// mode 0: G^T [G | Y] (fit_multi's normal equations); mode 1: W^T Y only (edmdc_pinv_apply: A operand = rows of W = G P^T);
// mode 2: G^T G only (fit()'s Gram pass).
// Mode 0 is a staircase: number the columns of [G | Y] 0 .. nt + nty - 1 (G tiles, then Y tiles); by symmetry row-tile i
// needs columns i .. nt + nty - 1 only.  The row-tiles are cut into bands of 4 or 6: a 4-band into 4 x 6 blocks (A = the band's
// rows, B = 6 columns), a 6-band into 6 x 4 blocks the other way round (A = 4 COLUMN tiles, weighted, G or Y; B = the band's
// six row tiles), because the weight may sit on either factor of w g y^T.  The band sizes are chosen to minimise the number
// of blocks (a small DP: a band that starts at row a0 has ncol - a0 columns, cut in sixes or in fours).
// s.xplus (the tail tile has room for x_{t+1}: n + r + n <= tailp): the x part of Y rides in the last G tile of the SAME row, so
// the Y tiles are the rbf tiles only.  k = 512: 34 + 32 = 66 columns, 75 tasks (seven 4-bands and a 6-band); with the x part
// of Y as a 33rd Y tile every band has an odd number of columns and none divides: 78 tasks.  1683 wanted tile products in
// 75 x 24 = 1800 slots; 71 would be perfect packing.  Every wave still runs the same instruction stream (4 + 6 + 1 loads,
// 24 MFMAs per K-step), which is what keeps the waves of a slab together in the L2.
static void build_gram_tasks(const EdmdcShape& s, std::vector<GramTask>& tasks, int mode) {
    const int nt = s.width / 16;                     // G tiles (mode 1: tiles of a row of W = G P^T)
    tasks.clear();
    static_assert(GRAM_TB == 6 && GRAM_TA == 4, "a 6-band is cut into TB x TA blocks");
    // mode 0: a staircase over the columns [G tiles | Y tiles]; mode 1: the full rectangle (W tiles) x (Y tiles), Y = the rbf
    // tiles + the x part of the NEXT row's tail tile; mode 2 (round 3): the staircase over the G tiles alone -- G^T G without
    // G^T Y, which is all KoopmanEDMDc.fit needs before its pinv (Koopman/koopmanEDMDc.py:97 never forms G^T Y): 595 tile products
    // instead of 1 683 at k = 512.  All three are cut by the same DP; only the column count of a band differs.
    const bool stair = mode != 1;
    const int nty = mode == 2 ? 0 : ((stair && s.xplus) ? s.kp / 16 : s.kp / 16 + (s.n + 15) / 16);
    const int ncol = stair ? nt + nty : nty;
    auto code = [&](int c) { return stair ? (c < nt ? c : ((c - nt) | 0x10000)) : (c | 0x10000); };     // column number -> tile code
    // Ragged ends.  A band whose column count does not divide by its block width ends in a partly empty block.  The last
    // columns are Y tiles, which every row-tile needs: up to three SHARED SETS of four of them (set g = columns
    // ncol-4(g+1) .. ncol-4g-1) can be left out of a band's own blocks and handed to shared transposed blocks
    // (A = the set's four Y tiles, B = six row-tiles taken across the bands that left the set out).  Band sizes and the sets
    // each band leaves out are chosen together by a DP over (first row of the band, rows so far in each shared set mod 6).
    // k = 512, mode 0: bands 4 4 4 4 4 4 4 6, sets left out {-, 01, 1, -, 01, 0, -, -}: 69 own + 2 + 2 shared = 73 tasks (75
    // without the shared sets, 78 with the x part of Y as a 33rd Y tile); 1 683 wanted tile products, 71 would be perfect.
    // k = 512, mode 1: 34 x 33 = 1 122 products; plain 4 x 6 blocks need 9 x 6 = 54 tasks (round 2), the DP 49 (bands
    // 4 4 4 4 4 4 4 6, six of the 4-bands leave set 0 to four shared blocks); 47 would be perfect packing.
    // Transposed blocks of mode 1 read their A tiles (Y) from the lifted rows and their B tiles from the rows of W: GRAM_SWAP.
    // (mode 2 has no Y columns: the sets are its last G columns, and a band may leave one out only if all its rows lie before the
    // set's first column -- then every product of the shared block is wanted, as with Y sets.  k = 512: 30 -> 28 tasks.)
    const int gmax = mode == 2 ? ncol / GRAM_TA - 1 : nty / GRAM_TA;
    const int G = gmax < 0 ? 0 : (gmax < 3 ? gmax : 3);
    const int NM = 1 << G, NR = 216;                  // masks, (r0, r1, r2) in base 6
    struct Choice { int size = 0, mask = 0, next_r = 0; };
    const int INF = 1 << 28;
    std::vector<int> cost((size_t)(nt + 1) * NR, INF);
    std::vector<Choice> pick((size_t)(nt + 1) * NR);
    for (int r = 0; r < NR; ++r) {                    // all rows placed: a started shared block counts as one more task
        const int r0 = r % 6, r1 = (r / 6) % 6, r2 = r / 36;
        cost[(size_t)nt * NR + r] = (r0 > 0) + (r1 > 0) + (r2 > 0);
    }
    for (int a0 = nt - 1; a0 >= 0; --a0)
        for (int r = 0; r < NR; ++r) {
            const int rem = nt - a0;
            const int rr[3] = {r % 6, (r / 6) % 6, r / 36};
            for (int size : {4, 6}) {
                int sz = size;
                if (size == 6 && rem < 6) continue;
                if (size == 4 && rem < 4) sz = rem;
                const int width = size == 6 ? GRAM_TA : GRAM_TB;
                for (int m = 0; m < NM; ++m) {
                    int pop = 0, extra = 0, nr[3] = {rr[0], rr[1], rr[2]};
                    bool allowed = true;
                    for (int g = 0; g < G; ++g) if ((m >> g) & 1) {
                        if (stair && a0 + sz > ncol - GRAM_TA * (g + 1)) allowed = false;       // a row of the band lies inside the set
                        ++pop; extra += (nr[g] + sz) / 6; nr[g] = (nr[g] + sz) % 6;
                    }
                    if (!allowed) continue;
                    const int cols = (stair ? ncol - a0 : ncol) - GRAM_TA * pop;
                    if (cols < (stair ? sz : 0)) continue;          // a staircase band keeps at least its diagonal block
                    const int nxt = nr[0] + 6 * nr[1] + 36 * nr[2];
                    const int c = (cols + width - 1) / width + extra + cost[(size_t)(a0 + sz) * NR + nxt];
                    if (c < cost[(size_t)a0 * NR + r]) { cost[(size_t)a0 * NR + r] = c; pick[(size_t)a0 * NR + r] = Choice{sz == size ? size : sz, m, nxt}; }
                }
            }
        }
    const unsigned swap = stair ? 0u : GRAM_SWAP;
    std::vector<int> shared[3];                       // row-tiles whose products with set g go to the shared blocks
    for (int a0 = 0, r = 0; a0 < nt;) {
        const Choice ch = pick[(size_t)a0 * NR + r];
        const bool six = ch.size == 6;
        std::vector<int> cols;                        // the band's own columns
        for (int c = stair ? a0 : 0; c < ncol; ++c) {
            const int g = (ncol - 1 - c) / GRAM_TA;   // shared set the column belongs to (if any)
            if (g < G && ((ch.mask >> g) & 1)) continue;
            cols.push_back(c);
        }
        for (int g = 0; g < G; ++g) if ((ch.mask >> g) & 1) for (int i = 0; i < ch.size; ++i) shared[g].push_back(a0 + i);
        const int width = six ? GRAM_TA : GRAM_TB;
        for (size_t c0 = 0; c0 < cols.size(); c0 += width) {
            GramTask t; t.want = 0;
            if (six) {
                for (int i = 0; i < GRAM_TA; ++i) t.a[i] = (c0 + i < cols.size()) ? code(cols[c0 + i]) : -1;       // A = column tiles (weighted)
                for (int j = 0; j < GRAM_TB; ++j) t.b[j] = a0 + j;                                                  // B = the band's row tiles
                for (int i = 0; i < GRAM_TA; ++i) for (int j = 0; j < GRAM_TB; ++j)
                    if (t.a[i] >= 0 && (!stair || cols[c0 + i] >= a0 + j)) t.want |= 1u << (i * GRAM_TB + j);
                t.want |= swap;
            } else {
                for (int i = 0; i < GRAM_TA; ++i) t.a[i] = (i < ch.size) ? a0 + i : -1;
                for (int j = 0; j < GRAM_TB; ++j) t.b[j] = (c0 + j < cols.size()) ? code(cols[c0 + j]) : -1;
                for (int i = 0; i < GRAM_TA; ++i) for (int j = 0; j < GRAM_TB; ++j)
                    if (t.a[i] >= 0 && t.b[j] >= 0 && (!stair || cols[c0 + j] >= a0 + i)) t.want |= 1u << (i * GRAM_TB + j);
            }
            tasks.push_back(t);
        }
        a0 += ch.size;
        r = ch.next_r;
    }
    for (int g = 0; g < G; ++g)
        for (size_t r0 = 0; r0 < shared[g].size(); r0 += GRAM_TB) {
            GramTask t; t.want = 0;
            for (int i = 0; i < GRAM_TA; ++i) t.a[i] = code(ncol - GRAM_TA * (g + 1) + i);                          // the set's four Y tiles (weighted)
            for (int j = 0; j < GRAM_TB; ++j) t.b[j] = (r0 + j < shared[g].size()) ? shared[g][r0 + j] : -1;
            for (int i = 0; i < GRAM_TA; ++i) for (int j = 0; j < GRAM_TB; ++j) if (t.b[j] >= 0) t.want |= 1u << (i * GRAM_TB + j);
            t.want |= swap;
            tasks.push_back(t);
        }
}

// K-slabs per chunk: as many as keep tasks x slabs within the 2048 wave slots of the chip at 2 waves/SIMD
// (k = 512: 73 tasks x 28 slabs = 2044 waves, one resident round, 99.8 % of the slots busy)
constexpr int GRAM_WAVE_SLOTS = 2048;
static int gram_nslab(int ntasks) {
    int ns = GRAM_WAVE_SLOTS / (ntasks > 0 ? ntasks : 1);
    if (ns < 1) ns = 1;
    if (ns > 256) ns = 256;
    return ns;
}

size_t gram_partial_doubles(const EdmdcShape& s, int mode, int* ntasks_out, int* nslab_out) {
    std::vector<GramTask> tasks;
    build_gram_tasks(s, tasks, mode);
    if (ntasks_out) *ntasks_out = (int)tasks.size();
    const int ns = gram_nslab((int)tasks.size());
    if (nslab_out) *nslab_out = ns;
    return tasks.size() * (size_t)ns * GRAM_TA * GRAM_TB * 256;
}

size_t gram_task_bytes(const EdmdcShape& s, int mode) {
    std::vector<GramTask> tasks;
    build_gram_tasks(s, tasks, mode);
    return tasks.size() * sizeof(GramTask);
}

// d_tasks: device copy of the task table (owned by the ctx, rebuilt when the shape changes)
hipError_t upload_gram_tasks(hipStream_t st, const EdmdcShape& s, int mode, void* d_tasks, size_t cap_bytes, int* ntasks) {
    std::vector<GramTask> tasks;
    build_gram_tasks(s, tasks, mode);
    *ntasks = (int)tasks.size();
    if (tasks.size() * sizeof(GramTask) > cap_bytes) return hipErrorInvalidValue;
    return hipMemcpyAsync(d_tasks, tasks.data(), tasks.size() * sizeof(GramTask), hipMemcpyHostToDevice, st);
}

template <bool SEPARATE_A>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
gram_kernel(int W, int ntasks, int nslab, int items_per_xcd, int64_t ksteps_total, const GramTask* __restrict__ tasks,
            const double* __restrict__ ZA, const double* __restrict__ Z, const double* __restrict__ wrow, double* __restrict__ partial,
            int accumulate) {
    // ZA: rows the A tiles are cut from (== Z for the Gram; the rows of W = G P^T for edmdc_pinv_apply), same width W
    // XCD-aware item mapping: blocks b and b+8 share an XCD (observed round-robin, speed only); items are
    // ordered slab-major and each XCD takes a contiguous run, so the ~80 task-waves of a slab share one L2.
    const int bid = blockIdx.x;
    const int item = (bid & 7) * items_per_xcd + (bid >> 3);
    if ((bid >> 3) >= items_per_xcd || item >= ntasks * nslab) return;
    const int slab = item / ntasks;
    const int task = item - slab * ntasks;
    const int lane = threadIdx.x;
    const int kq = lane >> 4, col = lane & 15;

    const int64_t per = (ksteps_total + nslab - 1) / nslab;
    const int64_t ks0 = (int64_t)slab * per;
    int64_t ks1 = ks0 + per;
    if (ks1 > ksteps_total) ks1 = ksteps_total;

    // per-lane BYTE offsets (unsigned 32-bit) relative to the first row of the current k-step: with a wave-uniform base
    // pointer the loads take the "SGPR base + 32-bit VGPR offset" form, one address register per operand instead of two
    unsigned aoff[GRAM_TA], boff[GRAM_TB];
#pragma unroll
    for (int a = 0; a < GRAM_TA; ++a) {
        const int tae = tasks[task].a[a];
        const int ta = tae < 0 ? 0 : tae;
        aoff[a] = 8u * (unsigned)((kq + ((ta >> 16) & 1)) * W + (ta & 0xFFFF) * 16 + col);
    }
#pragma unroll
    for (int b = 0; b < GRAM_TB; ++b) {
        const int tbe = tasks[task].b[b];
        const int tb = tbe < 0 ? 0 : tbe;
        boff[b] = 8u * (unsigned)((kq + ((tb >> 16) & 1)) * W + (tb & 0xFFFF) * 16 + col);
    }
    const unsigned woff = 8u * (unsigned)kq;
    v4d acc[GRAM_TA][GRAM_TB];
#pragma unroll
    for (int a = 0; a < GRAM_TA; ++a)
#pragma unroll
        for (int b = 0; b < GRAM_TB; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};

    if (ks0 < ks1) {
        // wave-uniform bases of the current 4 rows: zp for the B tiles, za for the A tiles (the Gram proper keeps a single row
        // pointer; a transposed block of W^T Y takes its A tiles -- Y -- from the lifted rows and its B tiles from the rows of W)
        const bool swapped = SEPARATE_A && (tasks[task].want & GRAM_SWAP) != 0;
        const double* zp = (swapped ? ZA : Z) + ks0 * 4 * W;
        const double* za = SEPARATE_A ? (swapped ? Z : ZA) + ks0 * 4 * W : zp;
        const double* wp = wrow + ks0 * 4;
        // Operands are double-buffered by hand: two register sets, loop unrolled by two, the 11 loads of K-step k+1 issued
        // BEFORE the 24 MFMAs of K-step k, so that they have that whole K-step to arrive.  Left to the compiler the loop is
        // rotated so that a step's rows are waited for right after they are asked for, and with two register sets it keeps
        // a 64-bit VGPR pointer per operand and set (loop strength reduction) and spills.  So the loads are written out in
        // their "SGPR base + 32-bit VGPR offset" form (one address register per operand, shared by both sets), which takes
        // them out of the compiler's s_waitcnt bookkeeping: the waits are explicit -- vmcnt(11) = "everything but the 11
        // loads just issued has landed" -- and sched_barriers pin the order of the four phases (guide section 5.7, form iii).
        // Loads run up to two K-steps past the slab (the row buffers are padded by 8 rows) and are drained before the epilogue.
        static_assert(GRAM_TA + GRAM_TB + 1 == 11, "the vmcnt immediates below count 11 loads per register set");
        double a0[GRAM_TA], b0[GRAM_TB], w0, a1[GRAM_TA], b1[GRAM_TB], w1;
        auto gload = [](const double* base, unsigned byte_off) {
            double v;
            asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v) : "v"(byte_off), "s"(base) : "memory");
            return v;
        };
        auto load = [&](double* an, double* bn, double& wn) {
            wn = gload(wp, woff);
#pragma unroll
            for (int a = 0; a < GRAM_TA; ++a) an[a] = gload(za, aoff[a]);
#pragma unroll
            for (int b = 0; b < GRAM_TB; ++b) bn[b] = gload(zp, boff[b]);
        };
        auto advance = [&]() { zp += 4 * W; if constexpr (SEPARATE_A) za += 4 * W; else za = zp; wp += 4; };
        auto mfma = [&](double* an, const double* bn, double wn) {
#pragma unroll
            for (int a = 0; a < GRAM_TA; ++a) {
                an[a] *= wn;                        // pair weight (0 at bag ends / padding), in place: no third register set
#pragma unroll
                for (int b = 0; b < GRAM_TB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(an[a], bn[b], acc[a][b], 0, 0, 0);
            }
        };
        load(a0, b0, w0);
        int64_t ks = ks0;
        for (; ks + 2 <= ks1; ks += 2) {
            advance();
            load(a1, b1, w1);                       // K-step ks + 1
            asm volatile("s_waitcnt vmcnt(11)" ::: "memory");      // set 0 has landed
            __builtin_amdgcn_sched_barrier(0);
            mfma(a0, b0, w0);
            __builtin_amdgcn_sched_barrier(0);
            advance();
            load(a0, b0, w0);                       // K-step ks + 2 (past the slab on the last trip: padded, never consumed)
            asm volatile("s_waitcnt vmcnt(11)" ::: "memory");      // set 1 has landed
            __builtin_amdgcn_sched_barrier(0);
            mfma(a1, b1, w1);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // drain: nothing may land in a register the epilogue reuses
        __builtin_amdgcn_sched_barrier(0);
        if (ks < ks1) mfma(a0, b0, w0);             // odd tail: set 0 holds the last K-step
    }
    // partial[(task * NSLAB + slab)][tile = a * TB + b][lane][4]
    double* out = partial + ((int64_t)task * nslab + slab) * (GRAM_TA * GRAM_TB * 256) + lane * 4;
    if (accumulate) {
#pragma unroll
        for (int a = 0; a < GRAM_TA; ++a)
#pragma unroll
            for (int b = 0; b < GRAM_TB; ++b) {
                v4d* o = reinterpret_cast<v4d*>(out + (a * GRAM_TB + b) * 256);
                *o = *o + acc[a][b];
            }
    } else {
#pragma unroll
        for (int a = 0; a < GRAM_TA; ++a)
#pragma unroll
            for (int b = 0; b < GRAM_TB; ++b) *reinterpret_cast<v4d*>(out + (a * GRAM_TB + b) * 256) = acc[a][b];
    }
}

hipError_t launch_gram_chunk_tasks(hipStream_t st, const EdmdcShape& s, int ntasks, const void* d_tasks, int64_t npairs,
                                   const double* Arows, const double* Zrows, const double* wrow, double* partial, int accumulate) {
    const int64_t ksteps = (npairs + 3) / 4;
    const int nslab = gram_nslab(ntasks);
    const int per_xcd = (ntasks * nslab + 7) / 8;
    if (Arows == Zrows)
        hipLaunchKernelGGL(gram_kernel<false>, dim3((unsigned)(8 * per_xcd)), dim3(64), 0, st, s.width, ntasks, nslab, per_xcd, ksteps,
                           reinterpret_cast<const GramTask*>(d_tasks), Arows, Zrows, wrow, partial, accumulate);
    else
        hipLaunchKernelGGL(gram_kernel<true>, dim3((unsigned)(8 * per_xcd)), dim3(64), 0, st, s.width, ntasks, nslab, per_xcd, ksteps,
                           reinterpret_cast<const GramTask*>(d_tasks), Arows, Zrows, wrow, partial, accumulate);
    return hipGetLastError();
}

// W rows for edmdc_pinv_apply: Wrows[row][j] = sum_f Zrows[row][f] PdT[f][j]  (PdT = P^T in device feature order, [W][W],
// zero rows / columns for padding features), i.e. row t of W is (P g_t)^T, the t-th column of P G^T
// (Koopman/koopmanEDMDc.py:97 evaluated left to right).
//
// rows_times_pt_simple_kernel (round 2, kept as the second implementation the GPU suite compares with): one wave = 16 rows x
// RXP_TB column tiles; v_mfma_f64_16x16x4_f64 with A[i][k] = Z[row0+i][f0+k], B[k][j] = PdT[f0+k][j0+j]; 18 loads for 17
// MFMAs per K-step, nothing prefetched, every wave re-reads its half of PdT for 16 rows.
constexpr int RXP_TB = 17;
__global__ void __launch_bounds__(64) rows_times_pt_simple_kernel(int W, int64_t rows, const double* __restrict__ Zrows,
                                                                  const double* __restrict__ PdT, double* __restrict__ Wrows) {
    const int lane = threadIdx.x, kq = lane >> 4, col = lane & 15;
    const int64_t row0 = (int64_t)blockIdx.x * 16;
    const int j0 = blockIdx.y * (RXP_TB * 16);
    v4d acc[RXP_TB];
#pragma unroll
    for (int b = 0; b < RXP_TB; ++b) acc[b] = (v4d){0.0, 0.0, 0.0, 0.0};
    int64_t ar = row0 + col;
    if (ar >= rows) ar = rows - 1;                        // clamp: the clamped rows are never stored
    const double* ap = Zrows + ar * W + kq;
    const double* bp = PdT + (int64_t)kq * W + j0 + col;
    for (int f0 = 0; f0 < W; f0 += 4) {
        const double av = ap[f0];
#pragma unroll
        for (int b = 0; b < RXP_TB; ++b) {
            const double bv = (j0 + b * 16 < W) ? bp[(int64_t)f0 * W + b * 16] : 0.0;      // wave-uniform predicate
            acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[b], 0, 0, 0);
        }
    }
#pragma unroll
    for (int b = 0; b < RXP_TB; ++b) {
        if (j0 + b * 16 >= W) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = row0 + kq + 4 * r;
            if (row < rows) Wrows[row * W + j0 + b * 16 + col] = acc[b][r];
        }
    }
}

// wrows_kernel (round 3): the Gram kernel's treatment for W = Z PdT.  One wave = one item = a 4 x 6 block of 16 x 16 output
// tiles (96 accumulators = 192 VGPRs), K = the W features in steps of 4, operands double-buffered by hand exactly as in
// gram_kernel (10 loads for 24 MFMAs per K-step, two register sets, explicit vmcnt waits).  The A and B operands of
// v_mfma_f64_16x16x4_f64 have the same lane layout (lane & 15 = the non-K index, lane >> 4 = K), so the two sides of a block
// are interchangeable and the nt column tiles of W are covered without a ragged end:
//   type A item: A side = 4 row tiles of Z (64 rows),  B side = 6 column tiles of PdT  -> D[row][col]
//   type B item: A side = the last nt % 6 (<= 4) column tiles of PdT, B side = 6 row tiles of Z (96 rows) -> D[col][row]
// A unit = 12 row tiles = 192 rows: 3 x (nt / 6) type A items + 2 type B items; k = 512: nt = 34 = 5 x 6 + 4 -> 17 items of 24
// tile products for the unit's 12 x 34 = 408 products: every MFMA issued is wanted (the round-2 kernel: 17 of 18 loads per 17
// MFMAs, no reuse of PdT across row tiles).  Both types run the same instruction stream; what differs are two base pointers
// with their K strides (Z: 4 doubles per K-step, PdT: 4 rows) and ten per-lane offsets, all fixed before the loop.
// The items of a unit are consecutive on one XCD (blockIdx % 8 = XCD, as for the Gram): the unit's rows are fetched from HBM
// once and shared through that L2, which also holds PdT (2.4 MB at k = 512).
// Loads run up to two K-steps past the last feature: Z rows are followed by the next row (the buffers are padded by 8 rows),
// PdT must be allocated with 8 extra rows.
constexpr int WR_UNIT_TILES = 12;
struct WrowsPlan { int nt, nA, remB, items_per_unit; };
static WrowsPlan wrows_plan(const EdmdcShape& s) {
    WrowsPlan p;
    p.nt = s.width / 16;
    const int rem = p.nt % GRAM_TB;
    p.nA = p.nt / GRAM_TB + (rem > GRAM_TA ? 1 : 0);            // a remainder of 5 stays a (5/6 full) type A column group
    p.remB = rem <= GRAM_TA ? rem : 0;
    p.items_per_unit = (WR_UNIT_TILES / GRAM_TA) * p.nA + (p.remB ? WR_UNIT_TILES / GRAM_TB : 0);
    return p;
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
wrows_kernel(int W, int64_t rows, int nt, int nA, int remB, int items_per_unit, int64_t items_per_xcd, int64_t nitems,
             const double* __restrict__ Zrows, const double* __restrict__ PdT, double* __restrict__ Wrows) {
    const int64_t bid = blockIdx.x;
    const int64_t idx = bid >> 3;
    const int64_t item = (bid & 7) * items_per_xcd + idx;
    if (idx >= items_per_xcd || item >= nitems) return;
    const int64_t unit = item / items_per_unit;
    const int sub = (int)(item - unit * items_per_unit);
    const int lane = threadIdx.x, kq = lane >> 4, col = lane & 15;
    const bool typeB = sub >= (WR_UNIT_TILES / GRAM_TA) * nA;
    int64_t row0;
    int j0, ncolt;                                         // first column of W, valid column tiles of this item
    if (!typeB) {
        const int rb = sub / nA, cg = sub - rb * nA;
        row0 = unit * (WR_UNIT_TILES * 16) + rb * (GRAM_TA * 16);
        j0 = cg * (GRAM_TB * 16);
        ncolt = nt - cg * GRAM_TB < GRAM_TB ? nt - cg * GRAM_TB : GRAM_TB;
    } else {
        const int rb = sub - (WR_UNIT_TILES / GRAM_TA) * nA;
        row0 = unit * (WR_UNIT_TILES * 16) + rb * (GRAM_TB * 16);
        j0 = (nt - remB) * 16;
        ncolt = remB;
    }
    if (row0 >= rows) return;
    // per-lane BYTE offsets from the two wave-uniform bases (Z: first row of the item; PdT: first column of the item)
    auto zoff = [&](int t) {
        int64_t rr = row0 + 16 * t + col;
        if (rr >= rows) rr = rows - 1;                     // clamp: never stored
        return 8u * (unsigned)((rr - row0) * W + kq);
    };
    auto poff = [&](int t) { return 8u * (unsigned)(kq * W + 16 * (t < ncolt ? t : 0) + col); };
    unsigned aoff[GRAM_TA], boff[GRAM_TB];
#pragma unroll
    for (int a = 0; a < GRAM_TA; ++a) aoff[a] = typeB ? poff(a) : zoff(a);
#pragma unroll
    for (int b = 0; b < GRAM_TB; ++b) boff[b] = typeB ? zoff(b) : poff(b);
    const double* zb = Zrows + row0 * W;
    const double* pbase = PdT + j0;
    const double* pa = typeB ? pbase : zb;                 // A-side base and its stride per K-step (doubles)
    const double* pb = typeB ? zb : pbase;
    const int64_t sa = typeB ? 4 * (int64_t)W : 4, sb = typeB ? 4 : 4 * (int64_t)W;
    v4d acc[GRAM_TA][GRAM_TB];
#pragma unroll
    for (int a = 0; a < GRAM_TA; ++a)
#pragma unroll
        for (int b = 0; b < GRAM_TB; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
    {
        static_assert(GRAM_TA + GRAM_TB == 10, "the vmcnt immediates below count 10 loads per register set");
        double a0[GRAM_TA], b0[GRAM_TB], a1[GRAM_TA], b1[GRAM_TB];
        auto gload = [](const double* base, unsigned byte_off) {
            double v;
            asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v) : "v"(byte_off), "s"(base) : "memory");
            return v;
        };
        auto load = [&](double* an, double* bn) {
#pragma unroll
            for (int a = 0; a < GRAM_TA; ++a) an[a] = gload(pa, aoff[a]);
#pragma unroll
            for (int b = 0; b < GRAM_TB; ++b) bn[b] = gload(pb, boff[b]);
        };
        auto advance = [&]() { pa += sa; pb += sb; };
        auto mfma = [&](const double* an, const double* bn) {
#pragma unroll
            for (int a = 0; a < GRAM_TA; ++a)
#pragma unroll
                for (int b = 0; b < GRAM_TB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(an[a], bn[b], acc[a][b], 0, 0, 0);
        };
        const int ksteps = W / 4;                          // W is a multiple of 16: an even number of K-steps
        load(a0, b0);
        for (int ks = 0; ks < ksteps; ks += 2) {
            advance();
            load(a1, b1);                                  // K-step ks + 1
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");      // set 0 has landed
            __builtin_amdgcn_sched_barrier(0);
            mfma(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            advance();
            load(a0, b0);                                  // K-step ks + 2 (past the last feature on the last trip: padded, never consumed)
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");      // set 1 has landed
            __builtin_amdgcn_sched_barrier(0);
            mfma(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // drain: nothing may land in a register the epilogue reuses
        __builtin_amdgcn_sched_barrier(0);
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: A-side index = (lane >> 4) + 4 reg, B-side index = lane & 15
    if (!typeB) {
#pragma unroll
        for (int a = 0; a < GRAM_TA; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = row0 + 16 * a + kq + 4 * r;
                if (row >= rows) continue;
                double* o = Wrows + row * W + j0 + col;
#pragma unroll
                for (int b = 0; b < GRAM_TB; ++b) if (b < ncolt) o[16 * b] = acc[a][b][r];
            }
    } else {
#pragma unroll
        for (int b = 0; b < GRAM_TB; ++b) {
            const int64_t row = row0 + 16 * b + col;
            if (row >= rows) continue;
            double* o = Wrows + row * W + j0 + kq;
#pragma unroll
            for (int a = 0; a < GRAM_TA; ++a) {
                if (a >= ncolt) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[16 * a + 4 * r] = acc[a][b][r];
            }
        }
    }
}

hipError_t launch_rows_times_pt(hipStream_t st, const EdmdcShape& s, int64_t rows, const double* Zrows, const double* PdT, double* Wrows,
                                int simple) {
    if (rows <= 0) return hipSuccess;
    const int ntile = s.width / 16;
    if (simple) {
        hipLaunchKernelGGL(rows_times_pt_simple_kernel, dim3((unsigned)((rows + 15) / 16), (unsigned)((ntile + RXP_TB - 1) / RXP_TB)), dim3(64), 0, st,
                           s.width, rows, Zrows, PdT, Wrows);
        return hipGetLastError();
    }
    const WrowsPlan p = wrows_plan(s);
    const int64_t units = (rows + WR_UNIT_TILES * 16 - 1) / (WR_UNIT_TILES * 16);
    const int64_t nitems = units * p.items_per_unit;
    const int64_t units_per_xcd = (units + 7) / 8;
    const int64_t items_per_xcd = units_per_xcd * p.items_per_unit;
    hipLaunchKernelGGL(wrows_kernel, dim3((unsigned)(8 * items_per_xcd)), dim3(64), 0, st, s.width, rows, p.nt, p.nA, p.remB, p.items_per_unit,
                       items_per_xcd, nitems, Zrows, PdT, Wrows);
    return hipGetLastError();
}
// MFMA work of the W rows per row of Z: executed tile products (of 16 x 16 x 16 x 2 / 16 = 512 flop per row) and wanted ones
void wrows_decomposition(const EdmdcShape& s, int* items_per_unit, int* tiles_wanted_per_unit) {
    const WrowsPlan p = wrows_plan(s);
    if (items_per_unit) *items_per_unit = p.items_per_unit;
    if (tiles_wanted_per_unit) *tiles_wanted_per_unit = WR_UNIT_TILES * p.nt;
}

// device feature f (column of Zrows) -> reference feature index of G = [x | rbf | u] (or -1 for padding)
int edmdc_dev_to_ref_feature(const EdmdcShape& s, int f) {
    if (f < s.k) return s.n + f;
    if (f < s.kp) return -1;
    const int j = f - s.kp;
    if (j < s.n) return j;
    if (j < s.n + s.r) return s.d + (j - s.n);
    return -1;
}

// Finish: one thread per (task, tile, lane, reg); sums the slabs in index order and scatters to the
// reference feature order: G = [x (n) | rbf (k) | u (r)], Y = [x (n) | rbf (k)].
// Column f of a tile -> reference index and side.  ytile: the tile is cut from row t+1 (everything in it that is not padding
// or u is a Y feature); a G tile of an xplus shape carries x_{t+1} behind u: Y features in a G tile.
__device__ __forceinline__ int dev_to_ref_feature(const EdmdcShape& s, int f, bool ytile, bool& yfeat) {
    yfeat = ytile;
    if (f < s.k) return s.n + f;
    if (f < s.kp) return -1;
    const int j = f - s.kp;
    if (j < s.n) return j;
    if (j < s.n + s.r) return ytile ? -1 : s.d + (j - s.n);
    if (!ytile && s.xplus && j < s.n + s.r + s.n) { yfeat = true; return j - s.n - s.r; }
    return -1;
}
__global__ void __launch_bounds__(256) gram_finish_kernel(EdmdcShape s, int ntasks, int nslab, const GramTask* __restrict__ tasks,
                                                          const double* __restrict__ partial, int accumulate_out,
                                                          double* __restrict__ GtG, double* __restrict__ GtY) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over ntasks * 24 tiles * 256
    const int64_t total = (int64_t)ntasks * GRAM_TA * GRAM_TB * 256;
    if (idx >= total) return;
    const int e = (int)(idx & 255), tile = (int)((idx >> 8) % (GRAM_TA * GRAM_TB)), task = (int)(idx / (GRAM_TA * GRAM_TB * 256));
    if (!((tasks[task].want >> tile) & 1u)) return;     // empty slot, or a product another task owns
    const int lane = e >> 2, reg = e & 3;
    const int a = tile / GRAM_TB, b = tile % GRAM_TB;
    const int tae = tasks[task].a[a], tbe = tasks[task].b[b];
    const bool ya = (tae >> 16) & 1, yb = (tbe >> 16) & 1;
    const int ta = tae & 0xFFFF, tb = tbe & 0xFFFF;
    const int fa = ta * 16 + (lane >> 4) + 4 * reg;     // C/D layout of v_mfma_f64_16x16x4_f64: row = (lane>>4) + 4 reg (A side)
    const int fb = tb * 16 + (lane & 15);               //                                          col = lane & 15 (B side)
    if (!ya && !yb && ta == tb && fb < fa) return;      // diagonal G x G tile: the upper half is written, and mirrored below
    bool fya, fyb;
    const int ra = dev_to_ref_feature(s, fa, ya, fya), rb = dev_to_ref_feature(s, fb, yb, fyb);
    if (ra < 0 || rb < 0 || (fya && fyb)) return;       // padding, or a Y x Y product nobody asked for
    const bool is_y = fya || fyb;
    if (is_y && !GtY) return;                           // G^T G alone (mode 2): the x+ columns riding in the tail tile are Y features
    const int ri = fya ? rb : ra, rj = fya ? ra : rb;   // row of the output = the G-side feature (A = Y feature: transposed product)
    double sum = 0.0;
    const double* pp = partial + (int64_t)task * nslab * (GRAM_TA * GRAM_TB * 256) + tile * 256 + e;
    for (int sl = 0; sl < nslab; ++sl) sum += pp[(int64_t)sl * (GRAM_TA * GRAM_TB * 256)];
    if (is_y) {
        double* o = GtY + (int64_t)ri * s.d + rj;
        *o = accumulate_out ? *o + sum : sum;
    } else {
        double* o = GtG + (int64_t)ri * s.p + rj;
        const double v = accumulate_out ? *o + sum : sum;
        *o = v;
        if (ri != rj) GtG[(int64_t)rj * s.p + ri] = v;
    }
}
hipError_t launch_gram_finish_tasks(hipStream_t st, const EdmdcShape& s, int ntasks, const void* d_tasks, const double* partial,
                                    int accumulate_out, double* GtG, double* GtY) {
    const int64_t total = (int64_t)ntasks * GRAM_TA * GRAM_TB * 256;
    hipLaunchKernelGGL(gram_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, s, ntasks, gram_nslab(ntasks),
                       reinterpret_cast<const GramTask*>(d_tasks), partial, accumulate_out, GtG, GtY);
    return hipGetLastError();
}

}  // namespace brov
