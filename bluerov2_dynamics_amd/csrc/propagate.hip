// propagate.hip -- lifted-space propagation for KoopmanEDMDc.evaluate / multistep_rmse / simulate
// (Koopman/koopmanEDMDc.py:157-216):  Z <- A Z + B u_t, H times, for all start windows at once.
//
// Layout: feature-major.  Zt[i][w] = feature i (reference order [x | rbf]) of window w, leading
// dimension nwp (windows padded to a multiple of 96).  One step is the fp64 GEMM
//     Zout[d x nw] = [A | B] [d x (d+r)]  .  [Zin ; U_t] [(d+r) x nw]
// with both MFMA operands K-major so every wave-load is 4 rows x 128 contiguous bytes:
//     A operand  ABt[j][i] = [A|B][i][j]         (transposed once per call)
//     B operand  Zin[j][w] for j < d,  Ucur[(j-d)][w] for the input rows.
// A wave owns 4 feature tiles x 6 window tiles (same 192-VGPR accumulator block as the Gram).
#include "brov2_kernels.h"

namespace brov {

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int PTA = 4, PTB = 6;
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

// One propagation step.  grid.x = window blocks (96 windows), grid.y = feature blocks (64 features); 1 wave per block.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
propagate_kernel(int d, int ksteps, int64_t ldab, const double* __restrict__ ABt, int64_t nwp, const double* __restrict__ Zin,
                 const double* __restrict__ Ucur, int64_t ldu, double* __restrict__ Zout) {
    const int lane = threadIdx.x, kq = lane >> 4, col = lane & 15;
    const int64_t w0 = (int64_t)blockIdx.x * (PTB * 16);
    const int i0 = blockIdx.y * (PTA * 16);
    v4d acc[PTA][PTB];
#pragma unroll
    for (int a = 0; a < PTA; ++a)
#pragma unroll
        for (int b = 0; b < PTB; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int ks = 0; ks < ksteps; ++ks) {
        const int j = ks * 4 + kq;                               // K index of this lane
        const double* arow = ABt + (int64_t)j * ldab + i0 + col;
        const double* brow = (j < d) ? Zin + (int64_t)j * nwp + w0 + col : Ucur + (int64_t)(j - d) * ldu + w0 + col;
        double av[PTA], bv[PTB];
#pragma unroll
        for (int a = 0; a < PTA; ++a) av[a] = arow[a * 16];
#pragma unroll
        for (int b = 0; b < PTB; ++b) bv[b] = brow[b * 16];
#pragma unroll
        for (int a = 0; a < PTA; ++a)
#pragma unroll
            for (int b = 0; b < PTB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
    // C/D layout: row (feature) = (lane>>4) + 4 reg, col (window) = lane & 15
#pragma unroll
    for (int a = 0; a < PTA; ++a)
#pragma unroll
        for (int b = 0; b < PTB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Zout[(int64_t)(i0 + a * 16 + kq + 4 * r) * nwp + w0 + b * 16 + col] = acc[a][b][r];
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
hipError_t launch_propagate(hipStream_t st, const PropShape& s, const double* ABt, const double* Zin, const double* Ucur, int64_t ldu, double* Zout) {
    hipLaunchKernelGGL(propagate_kernel, dim3((unsigned)(s.nwp / (PTB * 16)), (unsigned)(s.dpad / (PTA * 16))), dim3(64), 0, st,
                       s.d, s.ksteps, (int64_t)s.dpad, ABt, s.nwp, Zin, Ucur, ldu, Zout);
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
