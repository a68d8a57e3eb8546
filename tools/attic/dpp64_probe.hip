// Can a wave take the 13 doubles of a k-means centre record from ONE vector register pair (lane l of every row of 16 holds
// double l & 15 of the record) and feed them to the fp64 FMA through DPP row_newbcast -- `v_fmac_f64_dpp acc, rec, x_j
// row_newbcast:j` -- at the rate of the scalar-operand form `v_fma_f64 acc, s[..], x_j, acc`?  Checks the semantics (the two
// forms must give the same bits) and times both with the records coming from an L2-resident table (64 KB), 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/dpp64_probe.hip -o /tmp/dpp64_probe && /tmp/dpp64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

struct __attribute__((aligned(128))) Cen { double v[16]; };
typedef const Cen __attribute__((address_space(4)))* ccp;

template <int J> __device__ __forceinline__ void fmac_bcast(double& acc, double rec, double x) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(rec), "v"(x), "i"(J));
}
template <int J> __device__ __forceinline__ double mov_bcast(double rec) {
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(rec), "i"(J));
    return r;
}
template <size_t... J> __device__ __forceinline__ double score_dpp(double rec, const double (&x)[12], std::index_sequence<J...>) {
    double sc = mov_bcast<15>(rec);                     // slot 15 holds -|c|^2 / 2
    (fmac_bcast<(int)J>(sc, rec, x[J]), ...);
    return sc;
}

// MODE 0: scalar records (s_load), two per trip;  MODE 1: DPP records from global loads, DEPTH in flight
template <int MODE, int DEPTH>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8)))
probe(const double* __restrict__ X, const Cen* __restrict__ Tg, const int* __restrict__ list, int nlist, int reps, double* __restrict__ out, int* __restrict__ outi) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    double x[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) x[j] = X[i * 12 + j];
    double best = -1.0e300;
    int bi = 0;
    auto pick = [&](double sc, int c) {
        bi = (sc <= best) ? bi : c;
        asm("v_max_f64 %0, %1, %2" : "=v"(best) : "v"(best), "v"(sc));
    };
    // per-wave candidate list, wave-uniform (every wave starts at another offset so that the waves of a CU ask for different records)
    const int w = (int)(i >> 6);
    for (int r = 0; r < reps; ++r) {
        const int __attribute__((address_space(4)))* L = (const int __attribute__((address_space(4)))*)(unsigned long long)(list + __builtin_amdgcn_readfirstlane((w * 37 + r * 11) % 16) * nlist);
        if constexpr (MODE == 0) {
            const ccp T = (ccp)(unsigned long long)Tg;
#pragma unroll 1
            for (int q = 0; q + 1 < nlist; q += 2) {
                const int c0 = L[q], c1 = L[q + 1];
                Cen a, b;
#pragma unroll
                for (int j = 0; j < 16; ++j) { a.v[j] = T[c0].v[j]; b.v[j] = T[c1].v[j]; }
                double s0 = fma(x[0], a.v[0], -a.v[12]), s1 = fma(x[0], b.v[0], -b.v[12]);
#pragma unroll
                for (int j = 1; j < 12; ++j) s0 = fma(x[j], a.v[j], s0);
                pick(s0, c0);
#pragma unroll
                for (int j = 1; j < 12; ++j) s1 = fma(x[j], b.v[j], s1);
                pick(s1, c1);
            }
        } else {
            const double* Tl = reinterpret_cast<const double*>(Tg) + (lane & 15);
            double rec[DEPTH];
            int cc[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) { cc[d] = L[d]; rec[d] = Tl[cc[d] * 16]; }
#pragma unroll 1
            for (int q = DEPTH; q < nlist + DEPTH; q += DEPTH) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    const double sc = score_dpp(rec[d], x, std::make_index_sequence<12>{});
                    pick(sc, cc[d]);
                    const int qn = q + d < nlist ? q + d : nlist - 1;          // padding: the last candidate again (harmless)
                    cc[d] = L[qn];
                    rec[d] = Tl[cc[d] * 16];
                }
            }
        }
    }
    out[i] = best;
    outi[i] = bi;
}

template <typename F> float ms_of(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize(); hipEventRecord(a); f(); f(); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 3;
}

int main() {
    const int64_t N = 512 * 1024;                        // 512 blocks x 1024 threads: 8 waves per SIMD on 256 CUs
    const int k = 512, nlist = 128, reps = 64;
    std::vector<double> hX(N * 12); std::vector<Cen> hT(k); std::vector<int> hl(16 * nlist);
    srand(1);
    for (auto& v : hX) v = rand() / (double)RAND_MAX - 0.5;
    for (auto& t : hT) { double s = 0; for (int j = 0; j < 12; ++j) { t.v[j] = rand() / (double)RAND_MAX - 0.5; s += t.v[j] * t.v[j]; } t.v[12] = 0.5 * s; t.v[13] = t.v[14] = 0; t.v[15] = -0.5 * s; }
    for (int s = 0; s < 16; ++s) { int c = rand() % 4; for (int q = 0; q < nlist; ++q) { hl[s * nlist + q] = c; c += 1 + rand() % 6; if (c >= k) c = k - 1; } }
    double *X, *o0, *o1; Cen* T; int *l, *i0, *i1;
    hipMalloc(&X, N * 96); hipMalloc(&T, k * 128); hipMalloc(&l, hl.size() * 4); hipMalloc(&o0, N * 8); hipMalloc(&o1, N * 8); hipMalloc(&i0, N * 4); hipMalloc(&i1, N * 4);
    hipMemcpy(X, hX.data(), N * 96, hipMemcpyHostToDevice); hipMemcpy(T, hT.data(), k * 128, hipMemcpyHostToDevice); hipMemcpy(l, hl.data(), hl.size() * 4, hipMemcpyHostToDevice);
    const double evals = (double)(N / 64) * nlist * reps;              // (wave, centre) evaluations per launch
    auto report = [&](const char* name, float ms) {
        // 13 fp64 VALU instructions of the score + 2 of the selection per evaluation; 1024 SIMDs, 4 cycles per instruction
        printf("%-34s %7.3f ms  = %.1f ns per (wave, centre) per SIMD = %.0f cycles at 2.0 GHz (16 instr x 4 = 64)\n", name, ms, ms * 1e6 / (evals / 1024), ms * 1e6 / (evals / 1024) * 2.0);
    };
    float t0 = ms_of([&] { hipLaunchKernelGGL((probe<0, 2>), dim3(512), dim3(1024), 0, 0, X, T, l, nlist, reps, o0, i0); });
    report("scalar records, 2 per trip", t0);
#define RUN(D) do { float t_ = ms_of([&] { hipLaunchKernelGGL((probe<1, D>), dim3(512), dim3(1024), 0, 0, X, T, l, nlist, reps, o1, i1); }); \
        report("DPP records, depth " #D, t_); } while (0)
    RUN(2); RUN(4); RUN(8);
    // how many waves per SIMD does the dependent FMA chain need to fill the vector ALU?  256 blocks (one per CU) of 256 / 512 /
    // 1024 threads = 1 / 2 / 4 waves per SIMD; the time per (wave, centre) per SIMD should stay at the 8-wave figure if one wave is enough
    for (int tpb = 256; tpb <= 1024; tpb *= 2) {
        const double ev = (double)(256 * tpb / 64) * nlist * reps;
        float t_ = ms_of([&] { hipLaunchKernelGGL((probe<1, 4>), dim3(256), dim3(tpb), 0, 0, X, T, l, nlist, reps, o1, i1); });
        printf("DPP records, depth 4, %d wave(s) per SIMD: %7.3f ms = %.1f ns per (wave, centre) per SIMD\n", tpb / 256, t_, t_ * 1e6 / (ev / 1024));
        float t0_ = ms_of([&] { hipLaunchKernelGGL((probe<0, 2>), dim3(256), dim3(tpb), 0, 0, X, T, l, nlist, reps, o0, i0); });
        printf("scalar records,       %d wave(s) per SIMD: %7.3f ms = %.1f ns per (wave, centre) per SIMD\n", tpb / 256, t0_, t0_ * 1e6 / (ev / 1024));
    }
    hipLaunchKernelGGL((probe<0, 2>), dim3(512), dim3(1024), 0, 0, X, T, l, nlist, reps, o0, i0);
    hipLaunchKernelGGL((probe<1, 8>), dim3(512), dim3(1024), 0, 0, X, T, l, nlist, reps, o1, i1);
    std::vector<double> a(N), b(N); std::vector<int> ia(N), ib(N);
    hipMemcpy(a.data(), o0, N * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o1, N * 8, hipMemcpyDeviceToHost);
    hipMemcpy(ia.data(), i0, N * 4, hipMemcpyDeviceToHost); hipMemcpy(ib.data(), i1, N * 4, hipMemcpyDeviceToHost);
    int64_t bad = 0;
    for (int64_t q = 0; q < N; ++q) bad += (a[q] != b[q]) || (ia[q] != ib[q]);
    printf("scores and indices of the two forms differ in %lld of %lld samples\n", (long long)bad, (long long)N);
    return bad != 0;
}
