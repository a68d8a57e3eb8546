// How many tickets per microsecond can the waves of a full-chip launch draw from global counters?  (round 5: sizing the dynamic
// assignment of list passes in kmeans_assign_lds_kernel<LIST>)   hipcc --offload-arch=gfx950 -O3 tools/atomic_ticket_probe.hip -o /tmp/atp
//   one counter for the device | one per XCD (blockIdx % 8) | one per block ; each wave's lane 0 draws `draws` tickets one after the other
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(1024) draw_kernel(unsigned long long* ctr, int mode, int draws, unsigned long long* sink, int spacing_clocks) {
    const int lane = threadIdx.x & 63;
    unsigned long long* c = ctr + (mode == 0 ? 0 : mode == 1 ? (blockIdx.x % 8) * 32 : blockIdx.x * 32);
    unsigned long long acc = 0;
    for (int d = 0; d < draws; ++d) {
        if (lane == 0) acc += atomicAdd(c, 1ull);
        if (spacing_clocks > 0) { const unsigned long long t0 = __builtin_readcyclecounter(); while (__builtin_readcyclecounter() - t0 < (unsigned long long)spacing_clocks) __builtin_amdgcn_s_sleep(8); }
    }
    if (lane == 0 && acc == 0x12345) sink[0] = acc;
}
int main() {
    unsigned long long *ctr, *sink;
    hipMalloc(&ctr, 256 * 32 * 8); hipMalloc(&sink, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int spacing : {0, 4000, 16000}) for (int mode = 0; mode < 3; ++mode) for (int draws : {12, 48}) {
        hipMemset(ctr, 0, 256 * 32 * 8);
        hipLaunchKernelGGL(draw_kernel, dim3(256), dim3(1024), 0, 0, ctr, mode, draws, sink, spacing);
        hipDeviceSynchronize();
        hipMemset(ctr, 0, 256 * 32 * 8);
        hipEventRecord(e0);
        hipLaunchKernelGGL(draw_kernel, dim3(256), dim3(1024), 0, 0, ctr, mode, draws, sink, spacing);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double total = 256.0 * 16 * draws;
        printf("spacing %5d clk  mode %d (%s)  draws/wave %2d: %8.1f us  -> %7.1f tickets/us\n", spacing, mode,
               mode == 0 ? "one counter" : mode == 1 ? "per XCD    " : "per block  ", draws, ms * 1e3, total / (ms * 1e3));
    }
    return 0;
}
