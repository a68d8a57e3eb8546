#include <hip/hip_runtime.h>
#include <cstdio>
// store-only kernels in the geometry of lift_rows_kernel: rows of `pitch` doubles, a 128-thread block writes 512 doubles of each of RT consecutive rows
template <int MODE>
__global__ void __launch_bounds__(128) rows_kernel(double* __restrict__ Z, long rows, int pitch, int RT) {
    const long l0 = (long)blockIdx.x * RT;
    const int t = threadIdx.x;
    for (long l = l0; l < l0 + RT && l < rows; ++l) {
        double* zp = Z + l * pitch;
        if (MODE == 0) {            // lane: 4 adjacent doubles (two 16-B stores, 32-B lane stride)
            *reinterpret_cast<double2*>(zp + 4 * t) = make_double2(1.0, 2.0);
            *reinterpret_cast<double2*>(zp + 4 * t + 2) = make_double2(3.0, 4.0);
        } else if (MODE == 1) {     // each instruction 1 KiB contiguous per wave (2 KiB per block)
            *reinterpret_cast<double2*>(zp + 2 * t) = make_double2(1.0, 2.0);
            *reinterpret_cast<double2*>(zp + 256 + 2 * t) = make_double2(3.0, 4.0);
        } else {                    // 8-byte stores, 512 B contiguous per wave-instruction (round-1 lift)
            zp[t] = 1.0; zp[128 + t] = 2.0; zp[256 + t] = 3.0; zp[384 + t] = 4.0;
        }
    }
}
__global__ void fill_kernel(double2* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_double2(1.0, 2.0);
}
template <typename F> float ms_of(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize(); hipEventRecord(a); f(); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 2;
}
int main() {
    const long rows = 1 << 20;
    double* Z; hipMalloc(&Z, (size_t)rows * 4608 * 8);
    for (int pitch : {544, 512, 576}) {
        const double gb = (double)rows * 512 * 8 / 1e9;
        for (int RT : {64, 8, 1}) {
            float m0 = ms_of([&] { hipLaunchKernelGGL(rows_kernel<0>, dim3((rows + RT - 1) / RT), dim3(128), 0, 0, Z, rows, pitch, RT); });
            float m1 = ms_of([&] { hipLaunchKernelGGL(rows_kernel<1>, dim3((rows + RT - 1) / RT), dim3(128), 0, 0, Z, rows, pitch, RT); });
            float m2 = ms_of([&] { hipLaunchKernelGGL(rows_kernel<2>, dim3((rows + RT - 1) / RT), dim3(128), 0, 0, Z, rows, pitch, RT); });
            printf("pitch %4d RT %3d: 32B/lane %.3f ms (%.2f TB/s) | 1KiB/instr %.3f ms (%.2f TB/s) | 8B stores %.3f ms (%.2f TB/s)\n", pitch, RT, m0, gb / m0, m1, gb / m1, m2, gb / m2);
        }
    }
    float mf = ms_of([&] { hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, (double2*)Z, (size_t)rows * 512 / 2); });
    printf("fill of the same bytes: %.3f ms (%.2f TB/s)\n", mf, (double)rows * 512 * 8 / 1e9 / mf);
    return 0;
}
