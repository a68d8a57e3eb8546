// Which SIMD does each wave of a 512-thread workgroup run on?  rollout_pair_kernel puts the body half of a step on wave w and
// the thrust half on wave w + 4 and wants the two on ONE SIMD (speed only: the hand-over goes through LDS either way).
// HW_REG_HW_ID (hwreg 4) on gfx9-family parts: [3:0] wave slot, [5:4] SIMD, [11:8] CU, [15:13] SE.
// hipcc --offload-arch=gfx950 -O3 tools/probe_wave_simd.hip -o /tmp/probe_wave_simd && /tmp/probe_wave_simd
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(512) probe(unsigned* out) {
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (15 << 11));
}
int main() {
    const int nb = 256;
    unsigned *d, h[nb * 8];
    (void)hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(512), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int paired = 0, spread = 0;
    for (int b = 0; b < nb; ++b) {
        bool p = true, s = true;
        for (int w = 0; w < 4; ++w) {
            const unsigned s0 = (h[b * 8 + w] >> 4) & 3, s1 = (h[b * 8 + w + 4] >> 4) & 3;
            p = p && (s0 == s1);
            for (int v = 0; v < w; ++v) s = s && (((h[b * 8 + v] >> 4) & 3) != s0);
        }
        paired += p; spread += s;
    }
    printf("512-thread workgroups: %d of %d have waves w and w+4 on the same SIMD; %d of %d spread waves 0-3 over the four SIMDs\n", paired, nb, spread, nb);
    printf("block 0: SIMD of waves 0..7 =");
    for (int w = 0; w < 8; ++w) printf(" %u", (h[w] >> 4) & 3);
    printf("   CU ids:");
    for (int w = 0; w < 8; ++w) printf(" %u", (h[w] >> 8) & 15);
    printf("\n");
    return 0;
}
