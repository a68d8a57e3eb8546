// Known-bytes probe for the FETCH_SIZE / WRITE_SIZE counters (MI355X_MICROARCH.md: "other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern").  Every kernel reads (or writes) each byte of a 4 GiB buffer
// exactly once, far beyond the 256 MiB Infinity Cache; run under
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace ...   and   rocprofv3 --pmc WRITE_SIZE --kernel-trace ...
// and divide the counter (KiB) by 4 GiB: the factor to apply to that access pattern.
//   read16_kernel   16 B per lane, 1 KiB contiguous per wave-instruction (the streaming pattern the guide calibrated: 0.5)
//   read8row_kernel  8 B per lane: lane (kq, col) reads row kq, column col of a [rows][544] fp64 array, 16-column tiles --
//                    the operand loads of gram_kernel (4 rows x 128 B per wave-instruction)
//   read8_kernel     8 B per lane, 512 B contiguous per wave-instruction
//   write16_kernel  16 B per lane streaming stores
// hipcc --offload-arch=gfx950 -O3 tools/fetch_probe.hip -o /tmp/fetch_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void read16_kernel(const double2* __restrict__ p, size_t n, double* out) {
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const double2 v = p[i]; s += v.x + v.y; }
    if (s == 1.2345e300) out[0] = s;
}
__global__ void read8_kernel(const double* __restrict__ p, size_t n, double* out) {
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 1.2345e300) out[0] = s;
}
// one wave per block; wave w walks K-steps of 4 rows over its slab and reads all 34 tiles of each
__global__ void __launch_bounds__(64) read8row_kernel(const double* __restrict__ Z, size_t rows, int W, double* out) {
    const int lane = threadIdx.x, kq = lane >> 4, col = lane & 15;
    const size_t ksteps = rows / 4, per = (ksteps + gridDim.x - 1) / gridDim.x;
    const size_t k0 = blockIdx.x * per, k1 = (k0 + per < ksteps) ? k0 + per : ksteps;
    double s = 0;
    for (size_t k = k0; k < k1; ++k) {
        const double* r = Z + (k * 4 + kq) * (size_t)W + col;
        for (int t = 0; t < W / 16; ++t) s += r[t * 16];
    }
    if (s == 1.2345e300) out[0] = s;
}
__global__ void write16_kernel(double2* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_double2(1.0, 2.0);
}
int main() {
    const size_t bytes = (size_t)4 << 30, W = 544, rows = bytes / (W * 8) / 4 * 4;
    double *buf, *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) return 1;
    (void)hipMemset(buf, 0, bytes);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(read16_kernel, dim3(4096), dim3(256), 0, 0, (const double2*)buf, bytes / 16, out);
        hipLaunchKernelGGL(read8_kernel, dim3(4096), dim3(256), 0, 0, buf, bytes / 8, out);
        hipLaunchKernelGGL(read8row_kernel, dim3(8192), dim3(64), 0, 0, buf, rows, (int)W, out);
        hipLaunchKernelGGL(write16_kernel, dim3(4096), dim3(256), 0, 0, (double2*)buf, bytes / 16);
    }
    (void)hipDeviceSynchronize();
    printf("known bytes: read16 %zu read8 %zu read8row %zu write16 %zu\n", bytes, bytes, rows * W * 8, bytes);
    return 0;
}
