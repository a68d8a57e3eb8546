// controls.hip -- K7: synthetic control sequences generated on device.
//
// Counter-based (random access) splitmix64, so the value for (trajectory b, step t,
// channel j) does not depend on the layout or on how trajectories are sharded over GPUs:
//   counter = ((b0 + b) * T_total + t) * nu + j
//   bits    = mix(seed + (counter + 1) * 0x9E3779B97F4A7C15)
//   uniform = (bits >> 11) * 2^-53
// dist A: u = 2 uniform - 1 (bit exact with oracle/controls.py).
// dist B: AR(1) u_t = clip(0.98 u_{t-1} + 0.02 xi_t, -1, 1) with Box-Muller xi, the smooth
// thruster command of training/train_sim_brov2_koopmanEDMDc.py:160-164.
#include "brov2_kernels.h"

namespace brov {

__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t counter) {
    uint64_t z = seed + (counter + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double uniform01_at(uint64_t seed, uint64_t counter) {
    return (double)(splitmix64_at(seed, counter) >> 11) * 0x1.0p-53;
}

// Box-Muller normal of dist B from two uniforms of the counter stream: sqrt(-2 ln(1 - u1)) cos(2 pi u2) (oracle/controls.py).  1 - u1 is exact
// (u1 is a multiple of 2^-53 below 1), so log(1 - u1) is the oracle's log1p(-u1) to an ulp, and cospi(2 u2) is cos(2 pi u2) without the
// rounding of the product 2 pi u2 and without the large-argument branch of cos: the values agree with the NumPy oracle to ~6e-15 absolute
// (tested at 1e-12) for 0.7 of the instructions (round 6: the fill kernels are bound by fp64 VALU issue, profiles/r06_cfg4_pmc_summary.json).
__device__ __forceinline__ double box_muller(double u1, double u2) {
    return sqrt(-2.0 * log(1.0 - u1)) * cospi(2.0 * u2);
}

struct Scale8 { double s[8]; };

// one thread per (b, t); TUB stores are coalesced over b, BTU stores are one 8*nu-byte row per thread
template <int LAYOUT>
__global__ void __launch_bounds__(256) fill_iid_kernel(int64_t B, int64_t T, int nu, uint64_t seed, int64_t b0,
                                                       int64_t T_total, Scale8 sc, double* __restrict__ U) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    for (int64_t t = blockIdx.y; t < T; t += gridDim.y) {
        const uint64_t c0 = ((uint64_t)(b0 + b) * (uint64_t)T_total + (uint64_t)t) * (uint64_t)nu;
        for (int j = 0; j < nu; ++j) {
            const double v = (2.0 * uniform01_at(seed, c0 + j) - 1.0) * sc.s[j];
            if constexpr (LAYOUT == LAYOUT_BTU) U[(b * T + t) * nu + j] = v;
            else if constexpr (LAYOUT == LAYOUT_TUB) U[(t * nu + j) * B + b] = v;
            else U[((t * ((nu + 1) / 2) + j / 2) * B + b) * 2 + (j & 1)] = v;
        }
    }
}

// one thread per (b, j), sequential in t
template <int LAYOUT>
__global__ void __launch_bounds__(256) fill_ar1_kernel(int64_t B, int64_t T, int nu, uint64_t seed, int64_t b0,
                                                       int64_t T_total, Scale8 sc, double* __restrict__ U) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (b >= B) return;
    const uint64_t s2 = seed ^ 0xA5A5A5A5A5A5A5A5ull;
    double prev = 0.0;
    for (int64_t t = 0; t < T; ++t) {
        const uint64_t c = ((uint64_t)(b0 + b) * (uint64_t)T_total + (uint64_t)t) * (uint64_t)nu + (uint64_t)j;
        const double u1 = uniform01_at(s2, 2ull * c), u2 = uniform01_at(s2, 2ull * c + 1ull);
        const double xi = box_muller(u1, u2);
        prev = fmin(fmax(fma(0.98, prev, 0.02 * xi), -1.0), 1.0);
        const double v = prev * sc.s[j];
        if constexpr (LAYOUT == LAYOUT_BTU) U[(b * T + t) * nu + j] = v;
        else if constexpr (LAYOUT == LAYOUT_TUB) U[(t * nu + j) * B + b] = v;
        else U[((t * ((nu + 1) / 2) + j / 2) * B + b) * 2 + (j & 1)] = v;
    }
}

// The caller layout [B][T][nu] at size (config 4: 33.5 GB): with one thread per (trajectory, channel) and lanes over
// trajectories every 8-byte store of a wave hits its own cache line (0.3 TB/s, 112 ms for the block config 4's 28 ms rollout
// reads).  Here a lane still owns one (trajectory, channel) recurrence, but a wave = 64 / nu trajectories x nu channels, values
// are parked in LDS for FILL_TS steps and written out as 16-byte pieces along each trajectory's contiguous FILL_TS x nu x 8 bytes.
constexpr int FILL_TS = 16;
template <int NU>
__global__ void __launch_bounds__(256) fill_ar1_btu_kernel(int64_t B, int64_t T, uint64_t seed, int64_t b0, int64_t T_total, Scale8 sc,
                                                           double* __restrict__ U) {
    constexpr int TPW = 64 / NU;                       // trajectories per wave (8 or 10; lanes beyond TPW * NU idle)
    constexpr int SROW = 64 + 8;                       // doubles per parked step (64 lanes + padding: conflict-free 16-byte reads)
    __shared__ __attribute__((aligned(16))) double park[4][FILL_TS * SROW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bl = lane / NU, j = lane - bl * NU;
    const int64_t bw = ((int64_t)blockIdx.x * 4 + wave) * TPW;       // first trajectory of this wave
    const int64_t b = bw + bl;
    const bool live = bl < TPW && b < B;
    const uint64_t s2 = seed ^ 0xA5A5A5A5A5A5A5A5ull;
    double* pk = park[wave];
    double prev = 0.0;
    for (int64_t t0 = 0; t0 < T; t0 += FILL_TS) {
        const int ns = (int)(T - t0 < FILL_TS ? T - t0 : FILL_TS);
        for (int s = 0; s < ns; ++s) {
            const uint64_t c = ((uint64_t)(b0 + b) * (uint64_t)T_total + (uint64_t)(t0 + s)) * (uint64_t)NU + (uint64_t)j;
            const double u1 = uniform01_at(s2, 2ull * c), u2 = uniform01_at(s2, 2ull * c + 1ull);
            const double xi = box_muller(u1, u2);
            prev = fmin(fmax(fma(0.98, prev, 0.02 * xi), -1.0), 1.0);
            pk[s * SROW + lane] = prev * sc.s[j];
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // write-out: trajectory q of the wave owns ns * NU contiguous doubles; 16-byte pieces, lanes along the trajectory
        const int pieces = ns * NU / 2;                // per trajectory (NU even)
        for (int q = 0; q < TPW; ++q) {
            if (bw + q >= B) break;
            double* dst = U + ((bw + q) * T + t0) * NU;
            for (int c = lane; c < pieces; c += 64) {
                const int e = 2 * c, s = e / NU, jj = e - s * NU;
                const double2 v = *reinterpret_cast<const double2*>(pk + s * SROW + q * NU + jj);
                *reinterpret_cast<double2*>(dst + e) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    (void)live;
}

// dist A in the caller layout: one thread per (trajectory, step) row with lanes along the flattened [B*T] rows, so that a
// wave writes 64 x nu x 8 contiguous bytes
__global__ void __launch_bounds__(256) fill_iid_btu_kernel(int64_t B, int64_t T, int nu, uint64_t seed, int64_t b0, int64_t T_total,
                                                           Scale8 sc, double* __restrict__ U) {
    const int64_t rows = B * T;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = r / T, t = r - b * T;
        const uint64_t c0 = ((uint64_t)(b0 + b) * (uint64_t)T_total + (uint64_t)t) * (uint64_t)nu;
        for (int j = 0; j < nu; ++j) U[r * nu + j] = (2.0 * uniform01_at(seed, c0 + j) - 1.0) * sc.s[j];
    }
}

hipError_t launch_fill_controls(hipStream_t st, int layout, int dist, int64_t B, int64_t T, int nu, uint64_t seed,
                                int64_t b0, int64_t T_total, const double* scale8, double* U) {
    if (B <= 0 || T <= 0) return hipSuccess;
    Scale8 sc;
    for (int j = 0; j < 8; ++j) sc.s[j] = scale8 ? scale8[j] : 1.0;
    const unsigned gx = (unsigned)((B + 255) / 256);
    if (dist == 0) {
        const unsigned gy = (unsigned)(T < 4096 ? T : 4096);
        if (layout == LAYOUT_BTU) {
            const int64_t nb = (B * T + 255) / 256;
            hipLaunchKernelGGL(fill_iid_btu_kernel, dim3((unsigned)(nb < (1 << 20) ? nb : (1 << 20))), dim3(256), 0, st, B, T, nu, seed, b0, T_total, sc, U);
        } else if (layout == LAYOUT_TUB)
            hipLaunchKernelGGL(fill_iid_kernel<LAYOUT_TUB>, dim3(gx, gy), dim3(256), 0, st, B, T, nu, seed, b0, T_total, sc, U);
        else
            hipLaunchKernelGGL(fill_iid_kernel<LAYOUT_TPB>, dim3(gx, gy), dim3(256), 0, st, B, T, nu, seed, b0, T_total, sc, U);
    } else {
        if (layout == LAYOUT_BTU && nu == 8)
            hipLaunchKernelGGL(fill_ar1_btu_kernel<8>, dim3((unsigned)((B + 31) / 32)), dim3(256), 0, st, B, T, seed, b0, T_total, sc, U);
        else if (layout == LAYOUT_BTU && nu == 6)
            hipLaunchKernelGGL(fill_ar1_btu_kernel<6>, dim3((unsigned)((B + 39) / 40)), dim3(256), 0, st, B, T, seed, b0, T_total, sc, U);
        else if (layout == LAYOUT_BTU)
            hipLaunchKernelGGL(fill_ar1_kernel<LAYOUT_BTU>, dim3(gx, (unsigned)nu), dim3(256), 0, st, B, T, nu, seed, b0, T_total, sc, U);
        else if (layout == LAYOUT_TUB)
            hipLaunchKernelGGL(fill_ar1_kernel<LAYOUT_TUB>, dim3(gx, (unsigned)nu), dim3(256), 0, st, B, T, nu, seed, b0, T_total, sc, U);
        else
            hipLaunchKernelGGL(fill_ar1_kernel<LAYOUT_TPB>, dim3(gx, (unsigned)nu), dim3(256), 0, st, B, T, nu, seed, b0, T_total, sc, U);
    }
    return hipGetLastError();
}

// ---- XCD placement probe ------------------------------------------------------------------------------
// gram_kernel / propagate_kernel assume that blocks b and b + 8 land on the same XCD (round-robin dealing, observed on
// gfx950, not promised by HIP).  Every block records the XCC id it runs on (HW_REG_XCC_ID = hwreg 20, bits 3:0).
__global__ void __launch_bounds__(64) xcc_probe_kernel(int* __restrict__ out) {
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xF);
}
// 1: blockIdx % 8 groups blocks by XCD; 0: it does not; -1: the probe itself failed.  Synchronous (context creation only).
int probe_xcd_round_robin(hipStream_t st) {
    constexpr int NB = 256;
    int* d = nullptr;
    if (hipMalloc((void**)&d, NB * sizeof(int)) != hipSuccess) return -1;
    int h[NB];
    hipLaunchKernelGGL(xcc_probe_kernel, dim3(NB), dim3(64), 0, st, d);
    const bool ok = hipGetLastError() == hipSuccess && hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    if (!ok) return -1;
    for (int b = 8; b < NB; ++b) if (h[b] != h[b - 8]) return 0;
    for (int a = 0; a < 8; ++a) for (int b = a + 1; b < 8; ++b) if (h[a] == h[b]) return 0;
    return 1;
}

}  // namespace brov
