// Does HBM traffic slow down a compute-bound fp64 kernel on this box?  One wave per SIMD runs a fixed number of
// independent fp64 FMAs (16 chains); variants add, per 2048 FMAs, a 1 KiB streaming store and/or a 1 KiB streaming
// load per wave (the rollout kernel's ratio is ~780 FMA-slots per 10 KiB).  If the FMA time grows with traffic
// that the memory system absorbs easily (a few TB/s), the cause is not a queue or a latency but the chip's power
// budget moving from the cores to HBM.   hipcc --offload-arch=gfx950 -O3 tools/ubench_power.hip -o tools/ubench_power
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ST, int LD>      // KiB stored / loaded per wave per iteration
__global__ void __launch_bounds__(256) k(double* out, const double* in, double2* sink, int iters, double a, double b) {
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3 + i;
    const size_t gt = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 128; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = fma(x[i], a, b);
#pragma unroll
        for (int s = 0; s < ST; ++s) {
            typedef double v2d __attribute__((ext_vector_type(2)));
            v2d w; w[0] = x[s]; w[1] = x[s + 1];
            __builtin_nontemporal_store(w, reinterpret_cast<v2d*>(sink + ((size_t)(it * ST + s) * stride + gt)));
        }
#pragma unroll
        for (int s = 0; s < LD; ++s) {
            const double2 v = reinterpret_cast<const double2*>(in)[(size_t)(it * LD + s) * stride + gt];
            acc += v.x;
        }
    }
    double s = acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[gt] = s;
}
template <typename F> float time_ms(F f) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount;           // 1 wave per SIMD
    const int iters = 3000;
    const size_t lanes = (size_t)blocks * 256;
    double *out, *in; double2* sink;
    (void)hipMalloc(&out, lanes * 8);
    const size_t big = lanes * 16 * (size_t)iters * 6;   // up to 6 KiB per wave-iteration
    (void)hipMalloc(&sink, big); (void)hipMalloc(&in, big);
    (void)hipMemset(in, 0, big);
#define RUN(ST, LD) do { float ms = time_ms([&] { hipLaunchKernelGGL((k<ST, LD>), dim3(blocks), dim3(256), 0, 0, out, in, sink, iters, 1.0000001, 1e-9); }); \
    double fmas = 2048.0 * iters; double gbs = (double)(ST + LD) * 1024.0 * (blocks * 4.0) * iters / (ms * 1e-3) / 1e9; \
    printf("store %d KiB + load %d KiB per 2048 FMAs: %7.3f ms  %5.2f cycles/FMA @2.4GHz nominal  traffic %6.0f GB/s\n", ST, LD, ms, ms * 1e-3 * 2.4e9 / fmas, gbs); } while (0)
    RUN(0, 0); RUN(1, 0); RUN(2, 0); RUN(4, 0); RUN(6, 0); RUN(0, 2); RUN(0, 4); RUN(4, 2); RUN(6, 4); RUN(0, 0);
    return 0;
}
