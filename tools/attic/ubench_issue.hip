// Single-wave fp64 issue rate vs dependent latency on gfx950 (1 wave per SIMD), straight-line bodies:
//   chains = C independent FMA chains, body of 256 FMAs per loop iteration (loop overhead < 2 %).
// cycles per FMA = kernel cycles / FMAs per wave.  hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.hip -o tools/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>

template <int C>
__global__ void chain_kernel(double* out, int iters, double a, double b) {
    double x[C];
#pragma unroll
    for (int i = 0; i < C; ++i) x[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 256 / C; ++r)
#pragma unroll
            for (int i = 0; i < C; ++i) x[i] = fma(x[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < C; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// mix: FMA chains interleaved with independent v_mov/v_and style integer VALU (does cheap VALU cost an fp64 slot?)
template <int C>
__global__ void mix_kernel(double* out, int iters, double a, double b) {
    double x[C];
    unsigned y[C];
#pragma unroll
    for (int i = 0; i < C; ++i) { x[i] = threadIdx.x * 1e-3 + i; y[i] = threadIdx.x + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 256 / C; ++r)
#pragma unroll
            for (int i = 0; i < C; ++i) { x[i] = fma(x[i], a, b); y[i] = (y[i] ^ 0x5bd1e995u) + (y[i] >> 3); }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < C; ++i) s += x[i] + y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> float time_ms(F f) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
#define RUN(K, C, WPS) do { int blocks = cus * WPS; float ms = time_ms([&] { hipLaunchKernelGGL(K<C>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); }); \
    double fmas = 256.0 * iters; printf("%-12s chains=%2d waves/SIMD=%d : %7.3f ms  -> %5.2f cycles per FMA per wave @2.4GHz (%5.2f per SIMD)\n", #K, C, WPS, ms, ms * 1e-3 * 2.4e9 / fmas, ms * 1e-3 * 2.4e9 / fmas / WPS); } while (0)
int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    double* out; (void)hipMalloc(&out, sizeof(double) * 256 * 8 * 4096);
    const int iters = 4000;
    RUN(chain_kernel, 1, 1); RUN(chain_kernel, 2, 1); RUN(chain_kernel, 4, 1); RUN(chain_kernel, 8, 1); RUN(chain_kernel, 16, 1); RUN(chain_kernel, 32, 1);
    RUN(chain_kernel, 1, 2); RUN(chain_kernel, 4, 2); RUN(chain_kernel, 16, 2); RUN(chain_kernel, 16, 4);
    RUN(mix_kernel, 8, 1); RUN(mix_kernel, 16, 1); RUN(mix_kernel, 16, 2);
    (void)hipFree(out);
    return 0;
}
