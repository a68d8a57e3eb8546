// propagate.hip -- lifted-space propagation for KoopmanEDMDc.evaluate / multistep_rmse / simulate
// (Koopman/koopmanEDMDc.py:157-216):  Z <- A Z + B u_t, H times, for all start windows at once.
//
// Layout: feature-major.  Zt[i][w] = feature i (reference order [x | rbf]) of window w, leading
// dimension nwp (windows padded to a multiple of 128).  One step is the fp64 GEMM
//     Zout[d x nw] = [A | B] [d x (d+r)]  .  [Zin ; U_t] [(d+r) x nw]
// with both MFMA operands K-major so every wave-load is 4 rows x 128 contiguous bytes:
//     A operand  ABt[j][i] = [A|B][i][j]         (transposed once per call)
//     B operand  Zin[j][w]: rows j < d the lifted state, rows d..d+r-1 the inputs of the step (one buffer).
// A wave owns 3 feature tiles x 8 window tiles (the 192-VGPR accumulator block of the Gram, turned: d = 524 or 525
// pads to 528 = 11 x 48 features instead of 576, and 45 723 windows give 358 x 11 = 3 938 waves = two nearly full
// rounds of the chip's 2 048 wave slots; the 4 x 6 shape needed 4 296 = two rounds and a 5 % third).
#include "brov2_kernels.h"

namespace brov {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d_a8 __attribute__((ext_vector_type(2), aligned(8)));   // 16-byte access that is only 8-byte aligned (shifted input windows)
constexpr int PTA = 3, PTB = 8;
constexpr int PNMAX = 16;

// Zt[i][w] for window w = lane: x rows then rbf rows.  grid.x over 256-window groups, grid.y over 64-centre groups (+1 for x rows)
__global__ void __launch_bounds__(256) lift_t_kernel(int64_t nw, int64_t nwp, int n, int k, double gamma, int64_t xstride,
                                                     const double* __restrict__ X, const double* __restrict__ C,
                                                     double* __restrict__ Zt) {
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= nwp) return;
    const bool live = w < nw;
    double x[PNMAX], x2 = 0.0;
#pragma unroll
    for (int j = 0; j < PNMAX; ++j) { x[j] = (live && j < n) ? X[w * xstride + j] : 0.0; x2 = fma(x[j], x[j], x2); }
    if (blockIdx.y == 0) {
#pragma unroll
        for (int j = 0; j < PNMAX; ++j) if (j < n) Zt[(int64_t)j * nwp + w] = x[j];
    }
    const int c0 = blockIdx.y * 64;
    for (int c = c0; c < c0 + 64 && c < k; ++c) {
        const double* cc = C + (int64_t)c * n;    // wave-uniform -> scalar loads
        double dot = 0.0, c2 = 0.0;
#pragma unroll
        for (int j = 0; j < PNMAX; ++j) if (j < n) { const double cj = cc[j]; dot = fma(x[j], cj, dot); c2 = fma(cj, cj, c2); }
        Zt[(int64_t)(n + c) * nwp + w] = live ? exp(-gamma * ((x2 + c2) - 2.0 * dot)) : 0.0;
    }
}

// dst[j][i] = src[i][j] for src [rows][cols] row-major, dst leading dimension ldd (zero padding is done by memset)
__global__ void __launch_bounds__(256) transpose_kernel(int64_t rows, int64_t cols, const double* __restrict__ src, int64_t lds_,
                                                        double* __restrict__ dst, int64_t ldd) {
    __shared__ double tile[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t r0 = (int64_t)blockIdx.y * 16, c0 = (int64_t)blockIdx.x * 16;
    if (r0 + ty < rows && c0 + tx < cols) tile[ty][tx] = src[(r0 + ty) * lds_ + c0 + tx];
    __syncthreads();
    if (c0 + ty < cols && r0 + tx < rows) dst[(c0 + ty) * ldd + r0 + tx] = tile[tx][ty];
}

// One propagation step.  One wave per block.
//  * The K dimension of the step is ONE buffer: rows 0..d-1 of Zin are the lifted state, rows d..d+r-1 the inputs u_t of
//    this step (written by the previous step's launch, see below, or by set_input_rows_kernel before the first), so the
//    B operand is read through a single row pointer.
//  * Window tiles are "virtual": tile b of a block holds the windows w0 + 8 c + b, c = 0..15, so the eight B-operand
//    values of a lane (k = lane >> 4, c = lane & 15) are 64 contiguous bytes (4 x 16-byte loads instead of 8 x 8), and
//    so are the eight results it stores per feature row.  Which windows share a tile is immaterial to the product.
//  * Blocks are numbered so that the 11 feature blocks of one window block run on the same XCD (blockIdx % 8 selects
//    the XCD, observed round-robin): a window block's columns of Zin are pulled into one L2 only.  With the natural
//    2-D grid every XCD read all of Zin.
//  * Operands are double-buffered by hand (two register sets, loop unrolled by two, no copies): the loads of K-step k+1 are
//    issued BEFORE the 24 MFMAs of K-step k, so they have a whole K-step (1 536 pipe cycles, twice that with the SIMD's
//    second wave interleaved) to arrive.  The loop body is one basic block with sched_barriers between its four phases;
//    the first version (one set rotated through copies, a pointer select per load) made the compiler wait for the next
//    step's rows in the middle of the current step's MFMAs -- a prefetch distance of ~20 MFMAs.
//  * The last feature block of a window block also writes the NEXT step's input rows into Zout (rows d..d+r-1) and leaves
//    its own padding rows (>= d) alone.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
propagate_kernel(int d, int r, int ksteps, int64_t ldab, const double* __restrict__ ABt, int64_t nwp, const double* __restrict__ Zin,
                 const double* __restrict__ Unext, int64_t ldu, double* __restrict__ Zout, int nfb, int64_t nitems, int64_t items_per_xcd,
                 int64_t wb0) {
    const int64_t bid = blockIdx.x;
    const int64_t item = (bid & 7) * items_per_xcd + (bid >> 3);
    if ((bid >> 3) >= items_per_xcd || item >= nitems) return;
    const int64_t wbl = item / nfb;
    const int fb = (int)(item - wbl * nfb);
    const int64_t wb = wb0 + wbl;                        // window blocks [wb0, wb0 + nitems / nfb) of this launch
    const int lane = threadIdx.x, kq = lane >> 4, col = lane & 15;
    const int64_t w0 = wb * (PTB * 16);
    const int i0 = fb * (PTA * 16);
    v4d acc[PTA][PTB];
#pragma unroll
    for (int a = 0; a < PTA; ++a)
#pragma unroll
        for (int b = 0; b < PTB; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
    // per-lane row pointers of K index j = 4 ks + kq: [A|B]^T row j and [Z; u] row j
    const double* ap = ABt + (int64_t)kq * ldab + i0 + col;
    const double* zp = Zin + (int64_t)kq * nwp + w0 + 8 * col;
    const int64_t astep = 4 * ldab, zstep = 4 * nwp;
    double a0[PTA], b0[PTB], a1[PTA], b1[PTB];
    auto load = [&](double* an, double* bn) {
#pragma unroll
        for (int a = 0; a < PTA; ++a) an[a] = ap[a * 16];
#pragma unroll
        for (int b = 0; b < PTB; b += 2) {
            const double2 v = *reinterpret_cast<const double2*>(zp + b);
            bn[b] = v.x; bn[b + 1] = v.y;
        }
    };
    // `go` = false re-reads the current rows (never consumed): no branch inside the loop body
    auto advance = [&](bool go) { ap += go ? astep : 0; zp += go ? zstep : 0; };
    auto mfma = [&](const double* av, const double* bv) {
#pragma unroll
        for (int a = 0; a < PTA; ++a)
#pragma unroll
            for (int b = 0; b < PTB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
    };
    load(a0, b0);
    int ks = 0;
    for (; ks + 2 <= ksteps; ks += 2) {
        advance(true);
        load(a1, b1);                                   // K-step ks + 1
        __builtin_amdgcn_sched_barrier(0);
        mfma(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        advance(ks + 2 < ksteps);
        load(a0, b0);                                   // K-step ks + 2
        __builtin_amdgcn_sched_barrier(0);
        mfma(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (ks < ksteps) mfma(a0, b0);                      // odd tail: set 0 holds the last K-step
    // C/D layout: row (feature) = (lane >> 4) + 4 reg, column c = lane & 15 -> windows w0 + 8 c + b
#pragma unroll
    for (int a = 0; a < PTA; ++a)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int row = i0 + a * 16 + kq + 4 * rg;
            if (row < d) {
                double* o = Zout + (int64_t)row * nwp + w0 + 8 * col;
#pragma unroll
                for (int b = 0; b < PTB; b += 2) *reinterpret_cast<double2*>(o + b) = make_double2(acc[a][b][rg], acc[a][b + 1][rg]);
            }
        }
    if (fb == nfb - 1 && Unext != nullptr) {            // input rows of the next step: r rows x 128 windows, 2 windows per lane
        for (int j = 0; j < r; ++j) {
            const v2d_a8 v = *reinterpret_cast<const v2d_a8*>(Unext + (int64_t)j * ldu + w0 + 2 * lane);
            *reinterpret_cast<double2*>(Zout + (int64_t)(d + j) * nwp + w0 + 2 * lane) = make_double2(v[0], v[1]);
        }
    }
}

// rows d..d+r-1 of Zt <- U rows (before the first propagation step)
__global__ void __launch_bounds__(256) set_input_rows_kernel(int d, int r, int64_t nwp, const double* __restrict__ U, int64_t ldu,
                                                             double* __restrict__ Zt) {
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= nwp) return;
    for (int j = 0; j < r; ++j) Zt[(int64_t)(d + j) * nwp + w] = U[(int64_t)j * ldu + w];
}

// se[w] = sum_i (Xref[w][i] - Zt[i][w])^2 ; optional xhat [nw][n]
__global__ void __launch_bounds__(256) endpoint_se_kernel(int64_t nw, int64_t nwp, int n, int64_t xstride, const double* __restrict__ Xref,
                                                          const double* __restrict__ Zt, double* __restrict__ se, double* __restrict__ xhat) {
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= nw) return;
    double e = 0.0;
    for (int i = 0; i < n; ++i) {
        const double z = Zt[(int64_t)i * nwp + w];
        const double dd = Xref[w * xstride + i] - z;
        e = fma(dd, dd, e);
        if (xhat) xhat[w * n + i] = z;
    }
    se[w] = e;
}

// X_pred[b][t][i] = Zt[i][b]
__global__ void __launch_bounds__(256) extract_state_kernel(int64_t nb, int64_t nwp, int n, int64_t T1, int64_t t,
                                                            const double* __restrict__ Zt, double* __restrict__ Xp) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= nb) return;
    for (int i = 0; i < n; ++i) Xp[(b * T1 + t) * n + i] = Zt[(int64_t)i * nwp + b];
}

// Ust[t][j][b] = U_seq[b][t][j]
__global__ void __launch_bounds__(256) useq_t_kernel(int64_t nb, int64_t nbp, int64_t T, int r, const double* __restrict__ Us, double* __restrict__ Ust) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t t = blockIdx.y;
    if (b >= nb) return;
    for (int j = 0; j < r; ++j) Ust[(t * r + j) * nbp + b] = Us[(b * T + t) * r + j];
}

// ---- launchers -------------------------------------------------------------------------------------
PropShape prop_shape(int n, int r, int k, int64_t nw) {
    PropShape s;
    s.n = n; s.r = r; s.k = k; s.d = n + k; s.p = n + k + r;
    s.dpad = (s.d + PTA * 16 - 1) / (PTA * 16) * (PTA * 16);
    s.ksteps = (s.p + 3) / 4;
    s.ppad = s.ksteps * 4;
    s.nw = nw;
    s.nwp = (nw + PTB * 16 - 1) / (PTB * 16) * (PTB * 16);
    if (s.nwp == 0) s.nwp = PTB * 16;
    s.zrows = s.dpad > s.ppad ? s.dpad : s.ppad;
    return s;
}

hipError_t launch_lift_t(hipStream_t st, const PropShape& s, double gamma, int64_t xstride, const double* X, const double* C, double* Zt) {
    if (s.n > PNMAX) return hipErrorInvalidValue;
    hipLaunchKernelGGL(lift_t_kernel, dim3((unsigned)((s.nwp + 255) / 256), (unsigned)((s.k + 63) / 64)), dim3(256), 0, st,
                       s.nw, s.nwp, s.n, s.k, gamma, xstride, X, C, Zt);
    return hipGetLastError();
}
hipError_t launch_transpose(hipStream_t st, int64_t rows, int64_t cols, const double* src, int64_t lds_, double* dst, int64_t ldd) {
    if (rows <= 0 || cols <= 0) return hipSuccess;
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((cols + 15) / 16), (unsigned)((rows + 15) / 16)), dim3(256), 0, st, rows, cols, src, lds_, dst, ldd);
    return hipGetLastError();
}
hipError_t launch_set_input_rows(hipStream_t st, const PropShape& s, const double* U, int64_t ldu, double* Zt) {
    if (s.r <= 0) return hipSuccess;
    hipLaunchKernelGGL(set_input_rows_kernel, dim3((unsigned)((s.nwp + 255) / 256)), dim3(256), 0, st, s.d, s.r, s.nwp, U, ldu, Zt);
    return hipGetLastError();
}
// Zin rows d..d+r-1 must hold this step's inputs; Unext (may be NULL) = the next step's input rows [r][ldu], written into Zout.
// [wb0, wb0 + nwb): the window blocks (128 windows each) this launch advances; nwb < 0 = all of them.
int64_t prop_window_blocks(const PropShape& s) { return s.nwp / (PTB * 16); }
hipError_t launch_propagate(hipStream_t st, const PropShape& s, const double* ABt, const double* Zin, const double* Unext, int64_t ldu, double* Zout,
                            int64_t wb0, int64_t nwb) {
    const int nfb = s.dpad / (PTA * 16);
    if (nwb < 0) { wb0 = 0; nwb = prop_window_blocks(s); }
    if (nwb == 0) return hipSuccess;
    const int64_t nitems = nwb * nfb;
    const int64_t per_xcd = (nitems + 7) / 8;
    hipLaunchKernelGGL(propagate_kernel, dim3((unsigned)(8 * per_xcd)), dim3(64), 0, st, s.d, s.r, s.ksteps, (int64_t)s.dpad, ABt, s.nwp, Zin, Unext,
                       ldu, Zout, nfb, nitems, per_xcd, wb0);
    return hipGetLastError();
}
hipError_t launch_endpoint_se(hipStream_t st, const PropShape& s, int64_t xstride, const double* Xref, const double* Zt, double* se, double* xhat) {
    if (s.nw <= 0) return hipSuccess;
    hipLaunchKernelGGL(endpoint_se_kernel, dim3((unsigned)((s.nw + 255) / 256)), dim3(256), 0, st, s.nw, s.nwp, s.n, xstride, Xref, Zt, se, xhat);
    return hipGetLastError();
}
hipError_t launch_extract_state(hipStream_t st, const PropShape& s, int64_t T1, int64_t t, const double* Zt, double* Xp) {
    hipLaunchKernelGGL(extract_state_kernel, dim3((unsigned)((s.nw + 255) / 256)), dim3(256), 0, st, s.nw, s.nwp, s.n, T1, t, Zt, Xp);
    return hipGetLastError();
}
hipError_t launch_useq_t(hipStream_t st, const PropShape& s, int64_t T, const double* Us, double* Ust) {
    if (T <= 0 || s.r <= 0) return hipSuccess;
    hipLaunchKernelGGL(useq_t_kernel, dim3((unsigned)((s.nw + 255) / 256), (unsigned)T), dim3(256), 0, st, s.nw, s.nwp, T, s.r, Us, Ust);
    return hipGetLastError();
}

}  // namespace brov
