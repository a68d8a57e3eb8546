// Micro-benchmarks that anchor the fp64 rooflines on gfx950 (run on the GPU box):
//   1. v_fma_f64 issue rate, 1 / 2 / 4 waves per SIMD (independent chains)  -> fp64 VALU peak
//   2. v_mfma_f64_16x16x4_f64 back-to-back issue (independent accumulators) -> fp64 MFMA peak
//   3. MFMA + VALU fp64 co-issue from two waves on one SIMD
// hipcc --offload-arch=gfx950 -O3 tools/ubench_fp64.hip -o tools/ubench_fp64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ void fma_kernel(double* out, int iters, double a, double b) {
    double x[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) x[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) x[i] = fma(x[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ACCS>
__global__ void mfma_kernel(double* out, int iters, double a, double b) {
    v4d acc[ACCS];
#pragma unroll
    for (int i = 0; i < ACCS; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double av = a + threadIdx.x * 1e-6, bv = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACCS; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ACCS; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// waves with even wave-id do MFMA, odd do VALU fp64 (co-issue test)
__global__ void mixed_kernel(double* out, int iters, double a, double b) {
    const int wid = threadIdx.x >> 6;
    if (wid & 1) {
        double x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3 + i;
        for (int it = 0; it < iters * 16; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = fma(x[i], a, b);
        }
        double s = 0;
        for (int i = 0; i < 8; ++i) s += x[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        v4d acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = (v4d){0, 0, 0, 0};
        double av = a + threadIdx.x * 1e-6, bv = b;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
        }
        double s = 0;
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

template <typename F>
float time_ms(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", p.gcnArchName, cus, p.clockRate);
    double* out;
    hipMalloc(&out, sizeof(double) * 256 * 8 * 4096);
    const int iters = 20000;
    for (int wps : {1, 2, 4}) {           // waves per SIMD = blocks(256 thr) per CU
        const int blocks = cus * wps;
        float ms = time_ms([&] { hipLaunchKernelGGL(fma_kernel<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); });
        double flops = 2.0 * 8 * iters * 256.0 * blocks;
        printf("v_fma_f64   %d waves/SIMD, 8 chains : %8.3f ms  %7.2f TFLOP/s\n", wps, ms, flops / ms * 1e-9);
        ms = time_ms([&] { hipLaunchKernelGGL(fma_kernel<2>, dim3(blocks), dim3(256), 0, 0, out, iters * 4, 1.0000001, 1e-9); });
        printf("v_fma_f64   %d waves/SIMD, 2 chains : %8.3f ms  %7.2f TFLOP/s\n", wps, ms, 2.0 * 2 * iters * 4 * 256.0 * blocks / ms * 1e-9);
    }
    // half-populated waves: 32-thread blocks = one wave with lanes 32..63 masked off.  If the SIMD skipped the
    // empty half, the same number of wave-instructions would take half the time.
    for (int wps : {1, 2, 4}) {
        const int blocks = cus * 4 * wps;
        float ms = time_ms([&] { hipLaunchKernelGGL(fma_kernel<8>, dim3(blocks), dim3(32), 0, 0, out, iters, 1.0000001, 1e-9); });
        printf("v_fma_f64   %d HALF waves/SIMD (32 lanes), 8 chains : %8.3f ms  (%.2f TFLOP/s of useful lanes)\n", wps, ms,
               2.0 * 8 * iters * 32.0 * blocks / ms * 1e-9);
    }
    for (int wps : {1, 2}) {
        const int blocks = cus * wps;
        float ms = time_ms([&] { hipLaunchKernelGGL(mfma_kernel<8>, dim3(blocks), dim3(256), 0, 0, out, iters / 4, 1.0000001, 1e-9); });
        double flops = 2048.0 * 8 * (iters / 4) * 4.0 * blocks;
        printf("mfma_f64_16x16x4 %d waves/SIMD, 8 accs: %8.3f ms  %7.2f TFLOP/s\n", wps, ms, flops / ms * 1e-9);
        ms = time_ms([&] { hipLaunchKernelGGL(mfma_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, iters * 2, 1.0000001, 1e-9); });
        printf("mfma_f64_16x16x4 %d waves/SIMD, 1 acc : %8.3f ms  %7.2f TFLOP/s (dependent chain)\n", wps, ms, 2048.0 * iters * 2 * 4.0 * blocks / ms * 1e-9);
    }
    {
        const int blocks = cus;           // 512-thread blocks: 8 waves/CU = 2 per SIMD, one MFMA + one VALU wave
        float ms = time_ms([&] { hipLaunchKernelGGL(mixed_kernel, dim3(blocks), dim3(512), 0, 0, out, iters / 4, 1.0000001, 1e-9); });
        double mf = 2048.0 * 8 * (iters / 4) * 4.0 * blocks, vf = 2.0 * 8 * (iters / 4) * 16 * 256.0 * blocks;
        printf("mixed (MFMA wave + VALU wave per SIMD): %8.3f ms  mfma %7.2f + valu %7.2f TFLOP/s\n", ms, mf / ms * 1e-9, vf / ms * 1e-9);
    }
    hipFree(out);
    return 0;
}
