// How fast can the memory system take the caller-layout trajectory stores of config 4?  B trajectories x (T + 1) rows of 96 bytes,
// [B][T+1][12] doubles; 65 536 trajectories are in flight at a time (one lane each, 1 024 waves on the chip) and every `TILE`
// steps each of them writes TILE x 96 contiguous bytes (16-byte pieces, lanes along the trajectory's bytes: what an LDS-staged
// write-out produces; TILE = 0: the lane-per-row form, six 16-byte stores per lane and step, 64 trajectories per instruction).
// Optionally each step also reads the trajectory's 64-byte control row.  No arithmetic: the ceiling of the access pattern.
//   hipcc --offload-arch=gfx950 -O3 tools/btu_store_probe.hip -o /tmp/btu_store_probe && /tmp/btu_store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));

template <int TILE, bool READ>
__global__ void __launch_bounds__(256) probe(double* __restrict__ traj, const double* __restrict__ U, long B, long T, double* sink) {
    const int lane = threadIdx.x & 63;
    const long w = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;          // wave = 64 trajectories
    const long b0 = w * 64;
    if (b0 >= B) return;
    double acc = 0.0;
    if constexpr (TILE == 0) {
        double* tp = traj + (b0 + lane) * (T + 1) * 12;
        const double* up = U + (b0 + lane) * T * 8;
        for (long t = 0; t < T; ++t) {
            if constexpr (READ) { for (int i = 0; i < 4; ++i) { const v2d v = *reinterpret_cast<const v2d*>(up + t * 8 + 2 * i); acc += v[0] + v[1]; } }
            for (int i = 0; i < 6; ++i) { v2d v; v[0] = (double)t; v[1] = acc; *reinterpret_cast<v2d*>(tp + (t + 1) * 12 + 2 * i) = v; }
        }
    } else {
        constexpr int PIECES = TILE * 6;                                   // 16-byte pieces per trajectory and visit
        for (long t0 = 0; t0 < T; t0 += TILE) {
            if constexpr (READ) {
                for (int s = 0; s < TILE; ++s) { const double* up = U + ((b0 + lane) * T + t0 + s) * 8; for (int i = 0; i < 4; ++i) { const v2d v = *reinterpret_cast<const v2d*>(up + 2 * i); acc += v[0] + v[1]; } }
            }
            for (int g = lane; g < 64 * PIECES; g += 64) {
                const int j = g / PIECES, c = g - j * PIECES;
                v2d v; v[0] = (double)t0; v[1] = acc;
                *reinterpret_cast<v2d*>(traj + ((b0 + j) * (T + 1) + t0 + 1) * 12 + 2 * c) = v;
            }
        }
    }
    if (acc == 123.456) sink[0] = acc;
}
template <typename F> float ms_of(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize(); hipEventRecord(a); f(); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 2;
}
int main() {
    const long B = 65536, T = 4096;                                        // one resident round of lanes; 25.8 GB of rows, 17.2 GB of controls
    double *traj, *U, *sink;
    hipMalloc(&traj, (size_t)B * (T + 1) * 96); hipMalloc(&U, (size_t)B * T * 64); hipMalloc(&sink, 64);
    hipMemset(U, 0, (size_t)B * T * 64);
    const double gw = (double)B * T * 96 / 1e9, gr = (double)B * T * 64 / 1e9;
#define RUN(TL) do { \
        float w_ = ms_of([&] { hipLaunchKernelGGL((probe<TL, false>), dim3(B / 256), dim3(256), 0, 0, traj, U, B, T, sink); }); \
        float r_ = ms_of([&] { hipLaunchKernelGGL((probe<TL, true>), dim3(B / 256), dim3(256), 0, 0, traj, U, B, T, sink); }); \
        printf("piece %5d B (TILE %2d): stores only %7.2f ms = %.2f TB/s | stores + 64-B control reads %7.2f ms = %.2f TB/s total\n", TL ? TL * 96 : 16, TL, w_, gw / w_, r_, (gw + gr) / r_); } while (0)
    RUN(0); RUN(1); RUN(2); RUN(4); RUN(8); RUN(16); RUN(32);
    return 0;
}
