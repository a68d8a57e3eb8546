// rollout.hip -- time-loop kernels: batched RHS, Euler/RK4 rollouts, sliding-window
// endpoint error with the reference's carried thruster-lag state.
//
// Mapping: one wavefront lane = one trajectory (or one evaluation window); all state lives in VGPRs, nothing in scratch.
// Vehicle constants sit in a small device buffer (FastParams) read through the constant address space, i.e. as scalar
// loads.  Two forms of the rollout:
//   rollout_pair_kernel  thruster model, time-major layouts (the benchmark): every step split over the two waves of a SIMD
//                        -- thrust half and body half, LDS hand-over (K1p below);
//   rollout_kernel       every model, every layout: the whole step in one lane, control rows prefetched one step ahead;
//                        256-thread workgroups = one wave per SIMD at BASELINE config 2.
// One RK4 step of the thruster model executes ~760 fp64 VALU instructions per 64 trajectories (brov2_fast.h).
#include "brov2_device.h"
#include "brov2_fast.h"
#include "brov2_kernels.h"

namespace brov {

constexpr int TRIG_REFRESH = 64;   // steps between full sin/cos evaluations of the carried attitude trig (power of two)

#if BROV_CLOCK_STAMPS
// Diagnostic build only (tools/attic/clock_probe.py; never in the shipped library): the body wave of every workgroup stamps the
// shader clock (s_memtime) and the constant 100 MHz clock (s_memrealtime) around its time loop; the ratio is the clock the
// chip holds under this launch.  The stamps go to a buffer of their own, no output value is computed from them.
__device__ unsigned long long g_clock_stamps[4096][4];
extern "C" __attribute__((visibility("default"))) int brov_debug_clock_stamps(unsigned long long* host, int nblocks) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_clock_stamps), sizeof(unsigned long long) * 4 * nblocks, 0, hipMemcpyDeviceToHost);
}
#endif

// ---------------------------------------------------------------------------------------
// global <-> register movement for one row of NX/NU doubles
// ---------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void load_row(const double* __restrict__ src, double* r) {
    if constexpr (N % 2 == 0) {
        const double2* s2 = reinterpret_cast<const double2*>(src);
#pragma unroll
        for (int i = 0; i < N / 2; ++i) { double2 v = s2[i]; r[2 * i] = v.x; r[2 * i + 1] = v.y; }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) r[i] = src[i];
    }
}
template <int N>
__device__ __forceinline__ void store_row(double* __restrict__ dst, const double* r) {
    if constexpr (N % 2 == 0) {
        double2* d2 = reinterpret_cast<double2*>(dst);
#pragma unroll
        for (int i = 0; i < N / 2; ++i) d2[i] = make_double2(r[2 * i], r[2 * i + 1]);
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) dst[i] = r[i];
    }
}
template <int N>
__device__ __forceinline__ void load_soa(const double* __restrict__ src, int64_t ld, double* r) {
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = src[i * ld];
}
template <int N>
__device__ __forceinline__ void store_soa(double* __restrict__ dst, int64_t ld, const double* r) {
#pragma unroll
    for (int i = 0; i < N; ++i) dst[i * ld] = r[i];
}
// paired struct-of-arrays: element pair i of lane b lives at base[(i * B + b) * 2 .. +1]; `p` points at pair 0 of this lane
template <int N>
__device__ __forceinline__ void load_pairs(const double* __restrict__ p, int64_t ld2, double* r) {
#pragma unroll
    for (int i = 0; i < (N + 1) / 2; ++i) {
        // streamed once: nontemporal (slc) accesses keep the rows out of each other's way in L2 (-1 % kernel time)
        typedef double v2d __attribute__((ext_vector_type(2)));
        const v2d w = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p + i * ld2));
        const double2 v = make_double2(w[0], w[1]);
        r[2 * i] = v.x;
        if (2 * i + 1 < N) r[2 * i + 1] = v.y;
    }
}
template <int N>
__device__ __forceinline__ void store_pairs(double* __restrict__ p, int64_t ld2, const double* r) {
#pragma unroll
    for (int i = 0; i < (N + 1) / 2; ++i) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        v2d w; w[0] = r[2 * i]; w[1] = (2 * i + 1 < N) ? r[2 * i + 1] : 0.0;
        __builtin_nontemporal_store(w, reinterpret_cast<v2d*>(p + i * ld2));
    }
}

// ---------------------------------------------------------------------------------------
// one integrator step (training/train_tank_brov2_full_comparison.py:462-465 Euler,
// training/train_tank_brov2_rk4.py:385-394 RK4, ..._wrench_quat.py:258-263 renormalisation)
// ---------------------------------------------------------------------------------------
template <int MODEL, int INTEG, int LAGMODE>
__device__ __forceinline__ void integrate_step(const DevParams& p, double dt, double* x, const double* u, LagBank& lag) {
    constexpr int NX = Dims<MODEL>::NX;
    constexpr bool THR = (MODEL == MODEL_THRUSTER_EULER);
    double fcmd[8], F[8], tau[6];
    if constexpr (THR) {
#pragma unroll
        for (int i = 0; i < 8; ++i) fcmd[i] = thrust_poly(p, u[i]);
    } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) tau[i] = u[i];
    }
    if constexpr (INTEG == INTEG_EULER) {
        double k[NX];
        if constexpr (THR) { lag.forces_after(p, 1, fcmd, F); allocate(p, F, tau); }
        rhs_state<MODEL>(p, x, tau, k);
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = fma(dt, k[i], x[i]);
        if constexpr (THR) lag.advance(p, 1, fcmd);
    } else {
        double k[NX], acc[NX], xs[NX];
        const double h2 = 0.5 * dt;
        if constexpr (THR) { lag.forces_after(p, 1, fcmd, F); allocate(p, F, tau); }
        rhs_state<MODEL>(p, x, tau, k);
#pragma unroll
        for (int i = 0; i < NX; ++i) { acc[i] = k[i]; xs[i] = fma(h2, k[i], x[i]); }
        if constexpr (THR && LAGMODE == 0) { lag.forces_after(p, 2, fcmd, F); allocate(p, F, tau); }
        rhs_state<MODEL>(p, xs, tau, k);
#pragma unroll
        for (int i = 0; i < NX; ++i) { acc[i] = fma(2.0, k[i], acc[i]); xs[i] = fma(h2, k[i], x[i]); }
        if constexpr (THR && LAGMODE == 0) { lag.forces_after(p, 3, fcmd, F); allocate(p, F, tau); }
        rhs_state<MODEL>(p, xs, tau, k);
#pragma unroll
        for (int i = 0; i < NX; ++i) { acc[i] = fma(2.0, k[i], acc[i]); xs[i] = fma(dt, k[i], x[i]); }
        if constexpr (THR && LAGMODE == 0) { lag.forces_after(p, 4, fcmd, F); allocate(p, F, tau); }
        rhs_state<MODEL>(p, xs, tau, k);
        const double h6 = dt / 6.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = fma(h6, acc[i] + k[i], x[i]);
        if constexpr (THR) lag.advance(p, LAGMODE == 0 ? 4 : 1, fcmd);
    }
    if constexpr (MODEL == MODEL_WRENCH_QUAT) quat_normalize(x + 3);
}

// ---------------------------------------------------------------------------------------
// K2: batched dynamics() -- one RHS evaluation per row, lag advanced one sample
// ---------------------------------------------------------------------------------------
template <int MODEL>
__global__ void __launch_bounds__(256) rhs_kernel(DevParams p, int64_t B, const double* __restrict__ X,
                                                  const double* __restrict__ U, double* __restrict__ lag_io,
                                                  double* __restrict__ XD, unsigned long long* done, unsigned long long seq) {
    constexpr int NX = Dims<MODEL>::NX, NU = Dims<MODEL>::NU;
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double x[NX], u[NU], xd[NX], tau[6];
    load_row<NX>(X + b * NX, x);
    load_row<NU>(U + b * NU, u);
    if constexpr (MODEL == MODEL_THRUSTER_EULER) {
        LagBank lag;
        if (lag_io) load_row<24>(lag_io + b * 24, &lag.x[0][0]);
        else {
#pragma unroll
            for (int i = 0; i < 24; ++i) (&lag.x[0][0])[i] = 0.0;
        }
        double fcmd[8], F[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) fcmd[i] = thrust_poly(p, u[i]);
        lag.forces_after(p, 1, fcmd, F);
        allocate(p, F, tau);
        lag.advance(p, 1, fcmd);
        if (lag_io) store_row<24>(lag_io + b * 24, &lag.x[0][0]);
    } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) tau[i] = u[i];
    }
    rhs_state<MODEL>(p, x, tau, xd);
    store_row<NX>(XD + b * NX, xd);
    // per-call path (capi.hip: brov_rhs with a handful of vehicles): the results live in host memory and the host spins on
    // done[b] instead of synchronising the stream -- release at system scope, after the row's stores
    if (done) __hip_atomic_store(done + b, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// compute_thruster_forces (fossen/BlueROV2.py:265-278), batched
__global__ void __launch_bounds__(256) thruster_forces_kernel(DevParams p, int64_t B, const double* __restrict__ U,
                                                              double* __restrict__ lag_io, double* __restrict__ TAU,
                                                              unsigned long long* done, unsigned long long seq) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double u[8], fcmd[8], F[8], tau[6];
    LagBank lag;
    load_row<8>(U + b * 8, u);
    load_row<24>(lag_io + b * 24, &lag.x[0][0]);
#pragma unroll
    for (int i = 0; i < 8; ++i) fcmd[i] = thrust_poly(p, u[i]);
    lag.forces_after(p, 1, fcmd, F);
    allocate(p, F, tau);
    lag.advance(p, 1, fcmd);
    store_row<24>(lag_io + b * 24, &lag.x[0][0]);
    store_row<6>(TAU + b * 6, tau);
    if (done) __hip_atomic_store(done + b, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------------------
// K1: rollout.  U / traj layouts: BTU = [B][T][nu] / [B][rows][nx]; TUB = [T][nu][B] / [rows][nx][B].
// ---------------------------------------------------------------------------------------
template <int MODEL, int INTEG, int LAYOUT, int LAGMODE, bool TRACK, bool GENERIC>
__global__ void __launch_bounds__(256) rollout_kernel(const FastParams* __restrict__ pg, int64_t B, int64_t T, double dt,
                                                      const double* __restrict__ X0, const double* __restrict__ U,
                                                      double* __restrict__ lag_io, double* __restrict__ traj,
                                                      int64_t stride, double* __restrict__ XT) {
    constexpr int NX = Dims<MODEL>::NX, NU = Dims<MODEL>::NU;
    __shared__ double2 qt[4];
    init_quadrant_table(qt);
    __syncthreads();
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const CFP p = as_constant(pg);
    HotConsts h;
    load_hot(p, h);
    double x[NX];
    load_row<NX>(X0 + b * NX, x);
    LagZ lz;
    double Xl[8][3];
    if constexpr (MODEL == MODEL_THRUSTER_EULER) {
        if constexpr (TRACK) {
            load_row<24>(lag_io + b * 24, &Xl[0][0]);
            lz.from_thrusters(p, Xl);
        } else {
            lz.zero();
        }
        if constexpr (!GENERIC) lz.to_observer(p);
    }
    const int64_t rows = traj ? T / stride + 1 : 0;
    double* tp = nullptr;       // next trajectory row of this lane
    int64_t tstep = 0;          // distance between rows
    constexpr int NXP = (NX + 1) / 2, NUP = (NU + 1) / 2;
    if (traj) {
        if constexpr (LAYOUT == LAYOUT_BTU) { tp = traj + b * rows * NX; tstep = NX; }
        else if constexpr (LAYOUT == LAYOUT_TUB) { tp = traj + b; tstep = (int64_t)NX * B; }
        else { tp = traj + 2 * b; tstep = (int64_t)NXP * 2 * B; }
    }
    auto store_state = [&]() {
        if constexpr (LAYOUT == LAYOUT_BTU) store_row<NX>(tp, x);
        else if constexpr (LAYOUT == LAYOUT_TUB) store_soa<NX>(tp, B, x);
        else store_pairs<NX>(tp, 2 * B, x);
        tp += tstep;
    };
    const double* up;
    int64_t ustep;
    if constexpr (LAYOUT == LAYOUT_BTU) { up = U + b * T * NU; ustep = NU; }
    else if constexpr (LAYOUT == LAYOUT_TUB) { up = U + b; ustep = (int64_t)NU * B; }
    else { up = U + 2 * b; ustep = (int64_t)NUP * 2 * B; }

    double un[NU];
    if (T > 0) {
        if constexpr (LAYOUT == LAYOUT_BTU) load_row<NU>(up, un);
        else if constexpr (LAYOUT == LAYOUT_TUB) load_soa<NU>(up, B, un);
        else load_pairs<NU>(up, 2 * B, un);
    }
    // (storing the previous state at the top of the iteration, before the prefetch, measured 2 % slower)
    if (traj) store_state();
    int64_t countdown = stride;
    Trig tcarry;           // sin/cos of the attitude angles, carried from step to step (integrate_fast)
#if BROV_CLOCK_STAMPS
    const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int64_t t = 0; t < T; ++t) {
        double u[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) u[i] = un[i];
        // prefetch the next control row while this step computes -- unconditionally: the last step re-reads its own row
        // (never consumed), so `un` is defined by one load on every path and costs no copy between register files
        up += (t + 1 < T) ? ustep : 0;
        {
            if constexpr (LAYOUT == LAYOUT_BTU) load_row<NU>(up, un);
            else if constexpr (LAYOUT == LAYOUT_TUB) load_soa<NU>(up, B, un);
            else load_pairs<NU>(up, 2 * B, un);
        }
        step_fast<MODEL, INTEG, LAGMODE, TRACK, GENERIC>(h, p, dt, x, u, lz, Xl, qt, &tcarry, (t & (TRIG_REFRESH - 1)) == 0);
        if (traj && --countdown == 0) {
            countdown = stride;
            store_state();
        }
    }
#if BROV_CLOCK_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        g_clock_stamps[blockIdx.x][0] = st0; g_clock_stamps[blockIdx.x][1] = sr0;
        g_clock_stamps[blockIdx.x][2] = __builtin_amdgcn_s_memtime(); g_clock_stamps[blockIdx.x][3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    if (XT) store_row<NX>(XT + b * NX, x);
    if constexpr (MODEL == MODEL_THRUSTER_EULER && TRACK) store_row<24>(lag_io + b * 24, &Xl[0][0]);
}

// ---------------------------------------------------------------------------------------
// K1p: thruster-model rollouts with every step split over TWO waves of one SIMD (time-major layouts).
//
// A lone wave issues one instruction of any kind every ~4.3 clocks (tools/gen_ubench_valu.py: fp64 FMA, mul, add, s_mov,
// v_mov, s_nop all cost the same slot, dependent or not), and BASELINE config 2 has exactly one wave of trajectories per
// SIMD, so in rollout_kernel every scalar instruction, every register-file shuffle and every wait comes straight out of the
// fp64 issue stream.  With two waves on a SIMD the fp64 pipe is still one instruction per ~4.15 clocks in total, but scalar
// instructions, waits and half of the move traffic of one wave disappear behind the fp64 work of the other.  The thruster
// model's step splits cleanly into two such streams (brov2_fast.h):
//   THRUST wave (waves 4-7 of a 512-thread workgroup): control row -> thrust polynomial -> allocation -> lag bank ->
//       the four accelerations a_s = Minv tau_s the step's four dynamics() calls will see (quirk Q1); never sees the state.
//       ~200 fp64 instructions per step, holds the control prefetch and all allocation / lag constants.
//   BODY wave (waves 0-3): rigid-body right-hand sides + RK4 + trajectory stores; reads a_s, never sees a control.
//       ~590 fp64 instructions per step, no lag state, no controls: ~70 VGPRs fewer than the one-lane form, nothing spills.
// Wave w and wave w + 4 serve the same 64 trajectories and (workgroup waves are dealt round-robin over the four SIMDs) share
// a SIMD.  The thrust wave runs one step ahead through a two-slot LDS exchange (24 doubles per lane and step, 16-byte
// accesses, conflict-free), one workgroup barrier per step:
//       thrust: produce a(0);  for t: barrier; produce a(t+1) -> slot (t+1)&1
//       body  :                for t: barrier; consume a(t)   <- slot t&1, integrate, store
// Slot (t+1)&1 was last read during step t-1, i.e. before the barrier the body wave has just passed.  The barrier waits for
// LDS traffic only (s_waitcnt lgkmcnt(0)): trajectory stores and control loads stay in flight across it.
// ---------------------------------------------------------------------------------------
#ifndef BROV_PAIR_EXP
#define BROV_PAIR_EXP 0          // experiments (tools/build_variants.py): 1 body wave alone, 3 thrust wave alone, 4 pairwise LDS flags
#endif
#ifndef BROV_PAIR_RING
#define BROV_PAIR_RING 2         // exchange slots (the flag form can run the thrust wave further ahead)
#endif
#ifndef BROV_PAIR_STAGE
#define BROV_PAIR_STAGE 1        // LAYOUT_BTU: stored states go through LDS and leave in contiguous runs (0: six 16-byte stores per lane and step)
#endif
#ifndef BROV_PAIR_PRIO
#define BROV_PAIR_PRIO 0         // 1: s_setprio 3 for the body wave, 0 for the thrust wave
#endif
__device__ __forceinline__ void pair_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int INTEG, int LAYOUT, int LAGMODE, bool TRACK, bool GENERIC>
__global__ void __launch_bounds__(512) rollout_pair_kernel(const FastParams* __restrict__ pg, int64_t B, int64_t T, double dt,
                                                           const double* __restrict__ X0, const double* __restrict__ U,
                                                           double* __restrict__ lag_io, double* __restrict__ traj,
                                                           int64_t stride, double* __restrict__ XT) {
    // LAYOUT_BTU (the callers' layout, round 3): lane-per-row accesses -- the thrust wave reads its trajectory's 64-byte control
    // row with four 16-byte loads, the body wave writes its 96-byte state row with six 16-byte stores.  A wave-instruction then
    // touches 64 cache lines instead of 8, which costs address-pipeline cycles but no issue slots: the instruction counts are
    // those of the time-major layouts, and the thrust wave's loads fall into its slack.
    constexpr int MODEL = MODEL_THRUSTER_EULER;
    constexpr int NX = 12, NU = 8, NXP = 6, NUP = 4;
    constexpr int NS = (INTEG == INTEG_RK4) ? 4 : 1;          // dynamics() calls per step
    __shared__ double2 qt[4];
    __shared__ __attribute__((aligned(16))) double2 xch[BROV_PAIR_RING][4][NS * 3][64];     // [slot][pair][stage, channel pair][lane]
    // LAYOUT_BTU, every state stored (round 6): a body wave parks STG_S consecutive states of its 64 trajectories in LDS (row q = trajectory
    // q, STG_S x 96 bytes + 16 of padding: conflict-free both ways) and writes them out with lanes ALONG the trajectories -- a store
    // instruction then covers 1 KB in runs of STG_S x 96 contiguous bytes (11 cache lines) instead of one 16-byte piece in each of 64 lines
    constexpr bool STG = (LAYOUT == LAYOUT_BTU) && BROV_PAIR_STAGE;
    constexpr int STG_S = 2, STG_ROW = STG_S * NXP + 1, STG_PER = STG_S * NXP;               // double2 per parked row; pieces per trajectory and tile
    __shared__ __attribute__((aligned(16))) double2 stg[STG ? 4 * 64 * STG_ROW : 1];
#if BROV_PAIR_EXP == 4
    __shared__ int pflag[2][4];                               // [produced | consumed][pair]: steps done so far
    if (threadIdx.x < 8) (&pflag[0][0])[threadIdx.x] = 0;
#endif
    init_quadrant_table(qt);
    __syncthreads();
    const int wave = threadIdx.x >> 6, pair = wave & 3, lane = threadIdx.x & 63;
    const bool thrust = wave >= 4;                            // wave-uniform
#if BROV_PAIR_PRIO
    if (thrust) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(3);
#endif
    const int64_t b_raw = (int64_t)blockIdx.x * 256 + pair * 64 + lane;
    const bool live = b_raw < B;
    const int64_t b = live ? b_raw : B - 1;                   // dead lanes shadow the last trajectory, never store
    const CFP p = as_constant(pg);
    HotConstsResident h;
    load_hot(p, h);

    if (thrust) {
        LagZ lz;
        double Xl[8][3];
        if constexpr (TRACK) { load_row<24>(lag_io + b * 24, &Xl[0][0]); lz.from_thrusters(p, Xl); }
        else lz.zero();
        if constexpr (!GENERIC) lz.to_observer(p);
        const double* up;
        int64_t ustep;
        if constexpr (LAYOUT == LAYOUT_BTU) { up = U + b * T * NU; ustep = NU; }
        else if constexpr (LAYOUT == LAYOUT_TUB) { up = U + b; ustep = (int64_t)NU * B; }
        else { up = U + 2 * b; ustep = (int64_t)NUP * 2 * B; }
        double un[NU];
        int64_t tl = 0;                                       // step whose controls sit in `un`
        auto load_controls = [&]() {
            if constexpr (LAYOUT == LAYOUT_BTU) load_row<NU>(up, un);
            else if constexpr (LAYOUT == LAYOUT_TUB) load_soa<NU>(up, B, un);
            else load_pairs<NU>(up, 2 * B, un);
        };
        if (T > 0) load_controls();
        auto produce = [&](int slot) {
            double u[NU], fcmd[8], acmd[6], a[6];
#pragma unroll
            for (int i = 0; i < NU; ++i) u[i] = un[i];
            up += (tl + 1 < T) ? ustep : 0;                   // prefetch the next row (the last step re-reads its own, unused)
            ++tl;
            load_controls();
            const CFP pp = relaunder(p);
            command_accel<MODEL, !GENERIC>(pp, u, fcmd, acmd);
#pragma unroll
            for (int s = 1; s <= NS; ++s) {
                lag_stage_accel<LAGMODE, GENERIC>(h, pp, lz, s, acmd, a);
#pragma unroll
                for (int j = 0; j < 3; ++j) xch[slot][pair][(s - 1) * 3 + j][lane] = make_double2(a[2 * j], a[2 * j + 1]);
            }
            lag_step_advance<INTEG, LAGMODE, TRACK, GENERIC>(relaunder(p), lz, fcmd, acmd, Xl);
        };
#if BROV_PAIR_EXP == 1                                        // experiment: the body wave alone (results are garbage)
        for (int64_t t = 0; t < T; ++t) pair_barrier();
        (void)produce;
#elif BROV_PAIR_EXP == 4                                      // experiment: pairwise flags instead of the workgroup barrier
        {
            volatile int* prodf = &pflag[0][pair];
            volatile int* consf = &pflag[1][pair];
            int slot = 0;
            for (int64_t s = 0; s < T; ++s) {
                if (s >= BROV_PAIR_RING) {
                    const int need = (int)(s - BROV_PAIR_RING + 1);
                    int spins = 0;
                    while (*consf < need && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
                }
                asm volatile("" ::: "memory");
                produce(slot);
                asm volatile("" ::: "memory");
                *prodf = (int)(s + 1);
                slot = (slot + 1 == BROV_PAIR_RING) ? 0 : slot + 1;
            }
        }
#else
        if (T > 0) produce(0);
        for (int64_t t = 0; t < T; ++t) {
            pair_barrier();
            if (t + 1 < T) produce((int)((t + 1) & 1));
        }
#endif
        if constexpr (TRACK) { if (live) store_row<24>(lag_io + b * 24, &Xl[0][0]); }
    } else {
        double x[NX];
        load_row<NX>(X0 + b * NX, x);
        double* tp = nullptr;
        int64_t tstep = 0;
        if (traj) {
            if constexpr (LAYOUT == LAYOUT_BTU) { tp = traj + b * (T / stride + 1) * NX; tstep = NX; }
            else if constexpr (LAYOUT == LAYOUT_TUB) { tp = traj + b; tstep = (int64_t)NX * B; }
            else { tp = traj + 2 * b; tstep = (int64_t)NXP * 2 * B; }
        }
        // staged stores (LAYOUT_BTU, stride 1).  Write-out of a tile: STG_PER = 12 lanes per trajectory, five trajectories per instruction (lanes
        // 60-63 idle), 13 instructions for the wave's 64 trajectories -- lane l always handles piece l % 12 of trajectory 5 it + l / 12, so
        // its global offset and LDS index are one register each plus a wave-uniform term per instruction
        const bool staged = STG && traj != nullptr && stride == 1 && (uint64_t)(T + 1) * (NX * 8) * 64 < (1ull << 32);
        const int64_t bw = (int64_t)blockIdx.x * 256 + pair * 64;          // first trajectory of this wave
        constexpr int STG_TPI = 64 / STG_PER, STG_NI = (64 + STG_TPI - 1) / STG_TPI;       // trajectories per instruction, instructions per tile
        const int st_q = lane / STG_PER, st_c = lane - st_q * STG_PER;
        const unsigned st_rb = (unsigned)((T + 1) * (NX * 8));             // bytes of one trajectory
        const unsigned st_off = (unsigned)st_q * st_rb + (unsigned)st_c * 16u;
        const int st_lds = st_q * STG_ROW + st_c;
        const int st_live = (int)((B - bw) < 64 ? (B - bw) : 64);          // trajectories of this wave that exist (wave-uniform)
        int st_fill = 0;
        char* st_base = nullptr;                                          // wave-uniform: row `rows flushed so far` of the wave's first trajectory
        double2* const st_wave = stg + (STG ? pair * 64 * STG_ROW : 0);
        if constexpr (STG) {
            if (staged) {
                const uint64_t a = reinterpret_cast<uint64_t>(traj) + (uint64_t)bw * (uint64_t)st_rb;
                st_base = reinterpret_cast<char*>(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                                  (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(a & 0xFFFFFFFFull)));
            }
        }
        auto st_flush = [&](int ns) {
            if constexpr (STG) {
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (ns == STG_S) {
#pragma unroll
                    for (int it = 0; it < STG_NI; ++it) {
                        if (lane < STG_TPI * STG_PER && it * STG_TPI + st_q < st_live) {
                            const double2 v = st_wave[st_lds + it * STG_TPI * STG_ROW];
                            *reinterpret_cast<double2*>(st_base + (uint64_t)(it * STG_TPI) * st_rb + st_off) = v;
                        }
                    }
                } else {                                                  // the last, shorter tile of a trajectory with an odd number of rows
                    const int per = ns * NXP;
                    for (int i = lane; i < 64 * per; i += 64) {
                        const int q = i / per, c = i - q * per;
                        const double2 v = st_wave[q * STG_ROW + c];
                        if (q < st_live) *reinterpret_cast<double2*>(st_base + (uint64_t)q * st_rb + (uint64_t)c * 16u) = v;
                    }
                }
                st_base += (int64_t)ns * (NX * 8);
                st_fill = 0;
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        };
        auto store_state = [&]() {
            if constexpr (STG) {
                if (staged) {
                    double2* row = st_wave + lane * STG_ROW + st_fill * NXP;
#pragma unroll
                    for (int j = 0; j < NXP; ++j) row[j] = make_double2(x[2 * j], x[2 * j + 1]);
                    if (++st_fill == STG_S) st_flush(STG_S);
                    return;
                }
            }
            if (live) {
                if constexpr (LAYOUT == LAYOUT_BTU) store_row<NX>(tp, x);
                else if constexpr (LAYOUT == LAYOUT_TUB) store_soa<NX>(tp, B, x);
                else store_pairs<NX>(tp, 2 * B, x);
            }
            tp += tstep;
        };
        if (traj) store_state();
        int64_t countdown = stride;
        Trig tcarry;
#if BROV_CLOCK_STAMPS
        const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
#endif
#if BROV_PAIR_EXP == 4
        volatile int* prodf = &pflag[0][pair];
        volatile int* consf = &pflag[1][pair];
        int seen = 0, slot = 0;
#endif
        for (int64_t t = 0; t < T; ++t) {
#if BROV_PAIR_EXP == 3                                        // experiment: the thrust wave alone
            pair_barrier();
            const double2* src = &xch[t & 1][pair][0][lane];
            if (T >= 0) continue;
#elif BROV_PAIR_EXP == 4
            {
                int spins = 0;
                while (seen < (int)(t + 1) && ++spins < (1 << 22)) { seen = *prodf; if (seen < (int)(t + 1)) __builtin_amdgcn_s_sleep(1); }
            }
            asm volatile("" ::: "memory");
            const double2* src = &xch[slot][pair][0][lane];
            slot = (slot + 1 == BROV_PAIR_RING) ? 0 : slot + 1;
#else
            pair_barrier();
            const double2* src = &xch[t & 1][pair][0][lane];
#endif
            integrate_fast<MODEL, INTEG, GENERIC>(h, p, dt, x, [&](int s, double* a) {
#pragma unroll
                for (int j = 0; j < 3; ++j) { const double2 v = src[((s - 1) * 3 + j) * 64]; a[2 * j] = v.x; a[2 * j + 1] = v.y; }
            }, qt, &tcarry, (t & (TRIG_REFRESH - 1)) == 0);
#if BROV_PAIR_EXP == 4
            asm volatile("" ::: "memory");
            *consf = (int)(t + 1);                            // LDS requests of one wave are served in order: after the step's reads
            seen = *prodf;                                    // asked for now, looked at when the next step begins
#endif
            if (traj && --countdown == 0) {
                countdown = stride;
                store_state();
            }
        }
#if BROV_CLOCK_STAMPS
        if (wave == 0 && lane == 0 && blockIdx.x < 4096) {
            g_clock_stamps[blockIdx.x][0] = st0; g_clock_stamps[blockIdx.x][1] = sr0;
            g_clock_stamps[blockIdx.x][2] = __builtin_amdgcn_s_memtime(); g_clock_stamps[blockIdx.x][3] = __builtin_amdgcn_s_memrealtime();
        }
#endif
        if constexpr (STG) { if (staged && st_fill > 0) st_flush(st_fill); }
        if (XT && live) store_row<NX>(XT + b * NX, x);
    }
}

// ---------------------------------------------------------------------------------------
// K1b: rollout for the caller layout BTU (U[B][T][nu], traj[B][T+1][nx]) with LDS staging.
//
// Per trajectory the controls / states of consecutive steps are contiguous in memory but lanes
// are T*nu*8 bytes apart, so a lane-per-row access touches 64 cache lines per instruction.  Here
// each wave moves TILE steps at a time through its own LDS region:
//   in : global_load_lds_dwordx4 (LDS-DMA, no VGPRs, asynchronous, double buffered): one
//        wave-instruction copies 1 KiB = the next TILE*nu*8 contiguous bytes of 64/CH trajectories.
//        The DMA writes LDS linearly (slot = i*64 + lane), so the bank-conflict-free image is
//        obtained by permuting which 16-byte chunk of its trajectory a lane fetches
//        (chunk = c ^ ((j>>1)&7) for nu = 8) and reading with the same permutation.
//   out: every lane parks its state row in a padded LDS row; after TILE steps the wave writes the
//        tile out with 16-byte stores whose lanes walk along the trajectory rows (TILE*nx*8
//        contiguous bytes per trajectory).
// Lane <-> trajectory, all state in VGPRs, exactly as rollout_kernel.
// ---------------------------------------------------------------------------------------
constexpr int BTU_TILE = 2;

template <int NU> __device__ __forceinline__ int in_swizzle(int d, int j) {
    if constexpr (NU == 8) return d ^ ((j >> 1) & 7);   // conflict-free for ds_read_b128 (checked exhaustively)
    else return d;                                       // nu = 6: 2-way at worst
}

template <int MODEL, int INTEG, int LAGMODE, bool TRACK, bool GENERIC>
__global__ void __launch_bounds__(256) rollout_btu_lds_kernel(const FastParams* __restrict__ pg, int64_t B, int64_t T, double dt,
                                                              const double* __restrict__ X0, const double* __restrict__ U,
                                                              double* __restrict__ lag_io, double* __restrict__ traj,
                                                              double* __restrict__ XT) {
    constexpr int NX = Dims<MODEL>::NX, NU = Dims<MODEL>::NU;
    constexpr int CH = BTU_TILE * NU / 2;                 // 16-byte chunks per trajectory per input tile
    constexpr int IN_SLOTS = 64 * CH;                     // per wave per buffer
    constexpr int ORS = BTU_TILE * NX + 2;                // padded output row (doubles)
    constexpr int OCB = (NX % 2 == 0) ? 2 : 1;            // doubles per output chunk (16 B when rows stay 16-B aligned)
    constexpr int OCPR = BTU_TILE * NX / OCB;             // output chunks per trajectory per tile
    __shared__ __attribute__((aligned(16))) double lds_in[4][2][IN_SLOTS * 2];
    __shared__ __attribute__((aligned(16))) double lds_out[4][64 * ORS];
    __shared__ double2 qt[4];
    init_quadrant_table(qt);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t b0w = (int64_t)blockIdx.x * 256 + wave * 64;     // first trajectory of this wave
    const int64_t b_raw = b0w + lane;
    const bool live = b_raw < B;
    const int64_t b = live ? b_raw : B - 1;                        // dead lanes shadow the last trajectory, never store
    const CFP p = as_constant(pg);
    HotConsts h;
    load_hot(p, h);
    double x[NX];
    load_row<NX>(X0 + b * NX, x);
    LagZ lz;
    double Xl[8][3];
    if constexpr (MODEL == MODEL_THRUSTER_EULER) {
        if constexpr (TRACK) { load_row<24>(lag_io + b * 24, &Xl[0][0]); lz.from_thrusters(p, Xl); }
        else lz.zero();
        if constexpr (!GENERIC) lz.to_observer(p);
    }
    if (traj && live) store_row<NX>(traj + b * (T + 1) * NX, x);

    // issue the LDS-DMA of input tile `k` into buffer k&1
    auto issue_tile = [&](int64_t k) {
        const int64_t t0 = k * BTU_TILE;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int g = i * 64 + lane;
            const int j = g / CH, c = g % CH;
            const int d = in_swizzle<NU>(c, j);                    // data chunk this lane fetches
            int64_t bj = b0w + j;
            if (bj >= B) bj = B - 1;
            int64_t t = t0 + (2 * d) / NU;
            if (t >= T) t = T - 1;                                 // stay inside the buffer; value is never used
            const double* src = U + (bj * T + t) * NU + (2 * d) % NU;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(&lds_in[wave][k & 1][i * 128]), 16, 0, 0);
        }
    };

    const int64_t ntiles = (T + BTU_TILE - 1) / BTU_TILE;
    Trig tcarry;
    if (ntiles > 0) issue_tile(0);
    for (int64_t k = 0; k < ntiles; ++k) {
        __syncthreads();                     // tile k has landed (vmcnt(0)); last tile's LDS reads are finished
        if (k + 1 < ntiles) issue_tile(k + 1);
        const double* ib = &lds_in[wave][k & 1][0];
        double* ob = &lds_out[wave][lane * ORS];
        const int64_t t0 = k * BTU_TILE;
#pragma unroll 1
        for (int s = 0; s < BTU_TILE; ++s) {
            if (t0 + s < T) {
                double u[NU];
#pragma unroll
                for (int cc = 0; cc < NU / 2; ++cc) {
                    const int d = s * (NU / 2) + cc;
                    const double2 v = *reinterpret_cast<const double2*>(ib + (lane * CH + in_swizzle<NU>(d, lane)) * 2);
                    u[2 * cc] = v.x; u[2 * cc + 1] = v.y;
                }
                step_fast<MODEL, INTEG, LAGMODE, TRACK, GENERIC>(h, p, dt, x, u, lz, Xl, qt, &tcarry, ((t0 + s) & (TRIG_REFRESH - 1)) == 0);
                if (traj) store_row<NX>(ob + s * NX, x);
            }
        }
        if (traj) {
            __syncthreads();                 // the wave's output tile is complete
            const double* ot = &lds_out[wave][0];
#pragma unroll
            for (int i = 0; i < (64 * OCPR + 63) / 64; ++i) {
                const int g = i * 64 + lane;
                const int j = g / OCPR, c = g % OCPR;
                const int64_t bj = b0w + j;
                const int64_t row = t0 + 1 + (c * OCB) / NX;       // trajectory row of this chunk
                if (j < 64 && bj < B && row <= T) {
                    double* dst = traj + (bj * (T + 1) + t0 + 1) * NX + c * OCB;
                    const double* sp = ot + j * ORS + c * OCB;
                    if constexpr (OCB == 2) *reinterpret_cast<double2*>(dst) = *reinterpret_cast<const double2*>(sp);
                    else *dst = *sp;
                }
            }
        }
    }
    if (XT && live) store_row<NX>(XT + b * NX, x);
    if constexpr (MODEL == MODEL_THRUSTER_EULER && TRACK) { if (live) store_row<24>(lag_io + b * 24, &Xl[0][0]); }
}

// ---------------------------------------------------------------------------------------
// K3: sliding-window endpoint error (multistep_rmse_endpoint_physics,
// training/train_tank_brov2_full_comparison.py:469-487).  Quirk Q2: the reference uses ONE
// vehicle object for all windows, so window k starts from the lag state window k-1 left.
// The lag bank is LTI and driven by the commands only, so:
//   (A) zero-state response b_k of each window (parallel over windows),
//   (B) x_{k+1} = Phi x_k + b_k, Phi = Ad^(samples per window)  (sequential, 8 lanes, 9 FMA/iter),
//   (C) every window is an independent lane starting from its own x_k.
// ---------------------------------------------------------------------------------------
template <int NSUB>
__global__ void __launch_bounds__(256) window_lag_response_kernel(const FastParams* __restrict__ pg, int64_t nwin, int64_t H,
                                                                  const double* __restrict__ U, double* __restrict__ resp) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nwin) return;
    LagZ lz;
    lz.zero();
    for (int64_t t = 0; t < H; ++t) {
        const CFP pp = relaunder(as_constant(pg));
        double u[8], fcmd[8], acmd[6];
        load_row<8>(U + (k + t) * 8, u);
        command_accel<MODEL_THRUSTER_EULER, false>(pp, u, fcmd, acmd);
        lz.advance(NSUB == 4 ? pp->A4 : pp->A1, NSUB == 4 ? pp->b4 : pp->b1, acmd);
    }
    store_row<18>(resp + k * 18, &lz.z[0][0]);
}

// start[k] = acceleration-space lag state at the beginning of window k; start[0] = 0 (fresh vehicle object):
//     x_{k+1} = Phi x_k + b_k,  Phi = Ad^(samples per window), the same matrix for every window.
// Blocked scan over chunks of WSCAN_CHUNK windows, one lane per (chunk, wrench component):
//   (1) chunk_end[c]   = the recurrence over chunk c from a zero state               (parallel over chunks)
//   (2) chunk_start[c] : S_{c+1} = Phi^CHUNK S_c + chunk_end[c]                      (sequential over nwin/CHUNK chunks)
//   (3) start[k]       = the recurrence over chunk c from chunk_start[c]             (parallel over chunks)
// A single sequential pass over all windows (the first version) paid one exposed global-memory latency per window:
// 11 ms for the reference's 45 723 windows, more than everything else in the evaluator together.
constexpr int WSCAN_CHUNK = 64;

__device__ __forceinline__ void wscan_step(const double P[9], double& x0, double& x1, double& x2, double b0, double b1, double b2) {
    const double n0 = fma(P[2], x2, fma(P[1], x1, fma(P[0], x0, b0)));
    const double n1 = fma(P[5], x2, fma(P[4], x1, fma(P[3], x0, b1)));
    const double n2 = fma(P[8], x2, fma(P[7], x1, fma(P[6], x0, b2)));
    x0 = n0; x1 = n1; x2 = n2;
}

// phase 1 (store_start = 0): chunk_io[c] <- end state of chunk c from zero;  phase 3 (store_start = 1): start[k] for the
// chunk's windows, beginning from chunk_io[c]
__global__ void __launch_bounds__(256) window_lag_chunk_kernel(int64_t nwin, const double* __restrict__ Phi9, const double* __restrict__ resp,
                                                              double* __restrict__ chunk_io, double* __restrict__ start, int store_start) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t c = g / 6;
    const int i = (int)(g - c * 6);
    const int64_t k0 = c * WSCAN_CHUNK;
    if (k0 >= nwin) return;
    double P[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) P[j] = Phi9[j];
    double x0 = 0.0, x1 = 0.0, x2 = 0.0;
    if (store_start) { x0 = chunk_io[c * 18 + i * 3]; x1 = chunk_io[c * 18 + i * 3 + 1]; x2 = chunk_io[c * 18 + i * 3 + 2]; }
    const int64_t k1 = k0 + WSCAN_CHUNK < nwin ? k0 + WSCAN_CHUNK : nwin;
    for (int64_t k = k0; k < k1; ++k) {
        const double* r = resp + k * 18 + i * 3;
        if (store_start) { double* s = start + k * 18 + i * 3; s[0] = x0; s[1] = x1; s[2] = x2; }
        wscan_step(P, x0, x1, x2, r[0], r[1], r[2]);
    }
    if (!store_start) { chunk_io[c * 18 + i * 3] = x0; chunk_io[c * 18 + i * 3 + 1] = x1; chunk_io[c * 18 + i * 3 + 2] = x2; }
}

// phase 2: in place, chunk_io[c] (end-from-zero) -> state at the beginning of chunk c.  PhiC9 = Phi^WSCAN_CHUNK.
__global__ void __launch_bounds__(64) window_lag_scan_kernel(int64_t nchunks, const double* __restrict__ PhiC9, double* __restrict__ chunk_io) {
    const int i = threadIdx.x;  // wrench component
    if (i >= 6) return;
    double P[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) P[j] = PhiC9[j];
    double x0 = 0.0, x1 = 0.0, x2 = 0.0;
    double* e = chunk_io + i * 3;
    double b0 = 0, b1 = 0, b2 = 0;
    if (nchunks > 0) { b0 = e[0]; b1 = e[1]; b2 = e[2]; }
    for (int64_t c = 0; c < nchunks; ++c) {
        const double c0 = b0, c1 = b1, c2 = b2;
        if (c + 1 < nchunks) { b0 = e[(c + 1) * 18 + 0]; b1 = e[(c + 1) * 18 + 1]; b2 = e[(c + 1) * 18 + 2]; }
        e[c * 18 + 0] = x0; e[c * 18 + 1] = x1; e[c * 18 + 2] = x2;
        wscan_step(P, x0, x1, x2, c0, c1, c2);
    }
}

template <int MODEL, int INTEG, bool GENERIC>
__global__ void __launch_bounds__(256) window_endpoint_kernel(const FastParams* __restrict__ pg, int64_t nwin, int64_t H, double dt,
                                                              const double* __restrict__ X, const double* __restrict__ U,
                                                              const double* __restrict__ lag_start, double* __restrict__ se) {
    constexpr int NX = Dims<MODEL>::NX, NU = Dims<MODEL>::NU;
    __shared__ double2 qt[4];
    init_quadrant_table(qt);
    __syncthreads();
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nwin) return;
    const CFP p = as_constant(pg);
    HotConsts h;
    load_hot(p, h);
    double x[NX];
    load_row<NX>(X + k * NX, x);
    LagZ lz;
    double Xl[8][3];
    if constexpr (MODEL == MODEL_THRUSTER_EULER) {
        if (lag_start) load_row<18>(lag_start + k * 18, &lz.z[0][0]);
        else lz.zero();
        if constexpr (!GENERIC) lz.to_observer(p);
    }
    for (int64_t t = 0; t < H; ++t) {
        double u[NU];
        load_row<NU>(U + (k + t) * NU, u);   // lane k reads row k+t: coalesced across the wave
        // full sin/cos at every step (no carry): the evaluator's windows are short, and recorded wrench sequences drive the
        // open-loop models far off the data (RMSE ~ 20), where every ulp is amplified
        step_fast<MODEL, INTEG, 0, false, GENERIC>(h, p, dt, x, u, lz, Xl, qt);
    }
    double ref[NX], e = 0.0;
    load_row<NX>(X + (k + H) * NX, ref);
#pragma unroll
    for (int i = 0; i < NX; ++i) { const double d = x[i] - ref[i]; e = fma(d, d, e); }
    se[k] = e;
}

// Deterministic sum of n doubles: fixed-shape tree, one block.  out[0] = sum.
__global__ void __launch_bounds__(1024) sum_kernel(int64_t n, const double* __restrict__ v, double* __restrict__ out) {
    __shared__ double sh[1024];
    double a = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) a += v[i];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
hipError_t launch_sum(hipStream_t st, int64_t n, const double* v, double* out) {
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, st, n, v, out);
    return hipGetLastError();
}
static inline unsigned nblk(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

#define BROV_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

template <int MODEL>
static hipError_t launch_rhs_m(hipStream_t st, const DevParams& p, int64_t B, const double* x, const double* u, double* lag, double* xd,
                               unsigned long long* done, unsigned long long seq) {
    hipLaunchKernelGGL(rhs_kernel<MODEL>, dim3(nblk(B, 256)), dim3(256), 0, st, p, B, x, u, lag, xd, done, seq);
    return hipGetLastError();
}
hipError_t launch_rhs(hipStream_t st, const DevParams& p, int model, int64_t B, const double* x, const double* u, double* lag, double* xd,
                      unsigned long long* done, unsigned long long seq) {
    if (B <= 0) return hipSuccess;
    switch (model) {
        case MODEL_THRUSTER_EULER: return launch_rhs_m<MODEL_THRUSTER_EULER>(st, p, B, x, u, lag, xd, done, seq);
        case MODEL_WRENCH_EULER: return launch_rhs_m<MODEL_WRENCH_EULER>(st, p, B, x, u, lag, xd, done, seq);
        default: return launch_rhs_m<MODEL_WRENCH_QUAT>(st, p, B, x, u, lag, xd, done, seq);
    }
}
hipError_t launch_thruster_forces(hipStream_t st, const DevParams& p, int64_t B, const double* u, double* lag, double* tau,
                                  unsigned long long* done, unsigned long long seq) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(thruster_forces_kernel, dim3(nblk(B, 256)), dim3(256), 0, st, p, B, u, lag, tau, done, seq);
    return hipGetLastError();
}

template <int MODEL, int INTEG, int LAYOUT, int LAGMODE, bool TRACK, bool GENERIC>
static hipError_t launch_rollout_g(hipStream_t st, const FastParams* p, int64_t B, int64_t T, double dt, const double* x0,
                                   const double* U, double* lag, double* traj, int64_t stride, double* xT, bool want_lds, bool want_pair) {
    if constexpr (MODEL == MODEL_THRUSTER_EULER) {
        // the caller layout BTU takes the two-wave kernel where the one-lane kernel would use lane-per-row accesses anyway
        // (RK4: instruction-bound); the memory-bound Euler case keeps its LDS-staged kernel
        const bool staged_btu = LAYOUT == LAYOUT_BTU && (!traj || stride == 1) && T > 0 && want_lds;
        if (want_pair && !staged_btu) {
            hipLaunchKernelGGL((rollout_pair_kernel<INTEG, LAYOUT, LAGMODE, TRACK, GENERIC>), dim3(nblk(B, 256)), dim3(512), 0, st,
                               p, B, T, dt, x0, U, lag, traj, stride, xT);
            return hipGetLastError();
        }
    }
    if (LAYOUT == LAYOUT_BTU && (!traj || stride == 1) && T > 0 && want_lds)
        hipLaunchKernelGGL((rollout_btu_lds_kernel<MODEL, INTEG, LAGMODE, TRACK, GENERIC>), dim3(nblk(B, 256)), dim3(256), 0, st,
                           p, B, T, dt, x0, U, lag, traj, xT);
    else
        hipLaunchKernelGGL((rollout_kernel<MODEL, INTEG, LAYOUT, LAGMODE, TRACK, GENERIC>), dim3(nblk(B, 256)), dim3(256), 0, st,
                           p, B, T, dt, x0, U, lag, traj, stride, xT);
    return hipGetLastError();
}
template <int MODEL, int INTEG, int LAYOUT, int LAGMODE>
static hipError_t launch_rollout_t(hipStream_t st, const FastParams* p, int64_t B, int64_t T, double dt, const double* x0,
                                   const double* U, double* lag, double* traj, int64_t stride, double* xT, int btu_staging) {
    // LDS staging pays where the kernel is memory-bound (Euler, wrench models); the thruster RK4 kernel is
    // instruction-issue bound at one wave per SIMD and loses 14 % to the staging instructions
    const bool want_lds = (btu_staging & 3) == 1 || ((btu_staging & 3) == 0 && (INTEG == INTEG_EULER || MODEL != MODEL_THRUSTER_EULER));
    const bool generic = (btu_staging & 4) != 0;      // bit 2: vehicle has a current or xb/yb != 0 (set by the C ABI layer)
    const bool want_pair = (btu_staging & 8) == 0;    // bit 3: keep every step in one lane (A/B runs; set by the C ABI layer)
    constexpr bool DI = model_is_di(MODEL);
    if (MODEL == MODEL_THRUSTER_EULER && lag) {
        if constexpr (MODEL == MODEL_THRUSTER_EULER) {
            return generic ? launch_rollout_g<MODEL, INTEG, LAYOUT, LAGMODE, true, true>(st, p, B, T, dt, x0, U, lag, traj, stride, xT, want_lds, want_pair)
                           : launch_rollout_g<MODEL, INTEG, LAYOUT, LAGMODE, true, false>(st, p, B, T, dt, x0, U, lag, traj, stride, xT, want_lds, want_pair);
        }
    }
    if (generic && !DI) return launch_rollout_g<MODEL, INTEG, LAYOUT, LAGMODE, false, true>(st, p, B, T, dt, x0, U, lag, traj, stride, xT, want_lds, want_pair);
    return launch_rollout_g<MODEL, INTEG, LAYOUT, LAGMODE, false, false>(st, p, B, T, dt, x0, U, lag, traj, stride, xT, want_lds, want_pair);
}
template <int MODEL, int INTEG, int LAYOUT>
static hipError_t launch_rollout_l(hipStream_t st, const FastParams* p, int lag_mode, int64_t B, int64_t T, double dt,
                                   const double* x0, const double* U, double* lag, double* traj, int64_t stride, double* xT, int btu_staging) {
    if constexpr (MODEL == MODEL_THRUSTER_EULER && INTEG == INTEG_RK4) {
        if (lag_mode == 1) return launch_rollout_t<MODEL, INTEG, LAYOUT, 1>(st, p, B, T, dt, x0, U, lag, traj, stride, xT, btu_staging);
    }
    return launch_rollout_t<MODEL, INTEG, LAYOUT, 0>(st, p, B, T, dt, x0, U, lag, traj, stride, xT, btu_staging);
}
template <int MODEL>
static hipError_t launch_rollout_m(hipStream_t st, const FastParams* p, int integ, int lag_mode, int layout, int64_t B, int64_t T,
                                   double dt, const double* x0, const double* U, double* lag, double* traj, int64_t stride, double* xT, int btu_staging) {
    if (integ == INTEG_EULER) {
        if (layout == LAYOUT_BTU) return launch_rollout_l<MODEL, INTEG_EULER, LAYOUT_BTU>(st, p, lag_mode, B, T, dt, x0, U, lag, traj, stride, xT, btu_staging);
        if (layout == LAYOUT_TUB) return launch_rollout_l<MODEL, INTEG_EULER, LAYOUT_TUB>(st, p, lag_mode, B, T, dt, x0, U, lag, traj, stride, xT, btu_staging);
        return launch_rollout_l<MODEL, INTEG_EULER, LAYOUT_TPB>(st, p, lag_mode, B, T, dt, x0, U, lag, traj, stride, xT, btu_staging);
    }
    if (layout == LAYOUT_BTU) return launch_rollout_l<MODEL, INTEG_RK4, LAYOUT_BTU>(st, p, lag_mode, B, T, dt, x0, U, lag, traj, stride, xT, btu_staging);
    if (layout == LAYOUT_TUB) return launch_rollout_l<MODEL, INTEG_RK4, LAYOUT_TUB>(st, p, lag_mode, B, T, dt, x0, U, lag, traj, stride, xT, btu_staging);
    return launch_rollout_l<MODEL, INTEG_RK4, LAYOUT_TPB>(st, p, lag_mode, B, T, dt, x0, U, lag, traj, stride, xT, btu_staging);
}
hipError_t launch_rollout(hipStream_t st, const FastParams* p, int model, int integ, int lag_mode, int layout, int64_t B, int64_t T,
                          double dt, const double* x0, const double* U, double* lag, double* traj, int64_t stride, double* xT, int btu_staging) {
    if (B <= 0) return hipSuccess;
    switch (model) {
        case MODEL_THRUSTER_EULER: return launch_rollout_m<MODEL_THRUSTER_EULER>(st, p, integ, lag_mode, layout, B, T, dt, x0, U, lag, traj, stride, xT, btu_staging);
        case MODEL_WRENCH_EULER: return launch_rollout_m<MODEL_WRENCH_EULER>(st, p, integ, lag_mode, layout, B, T, dt, x0, U, lag, traj, stride, xT, btu_staging);
        case MODEL_WRENCH_QUAT: return launch_rollout_m<MODEL_WRENCH_QUAT>(st, p, integ, lag_mode, layout, B, T, dt, x0, U, lag, traj, stride, xT, btu_staging);
        case MODEL_DI_THRUSTER_EULER: return launch_rollout_m<MODEL_DI_THRUSTER_EULER>(st, p, integ, lag_mode, layout, B, T, dt, x0, U, nullptr, traj, stride, xT, btu_staging);
        case MODEL_DI_WRENCH_EULER: return launch_rollout_m<MODEL_DI_WRENCH_EULER>(st, p, integ, lag_mode, layout, B, T, dt, x0, U, nullptr, traj, stride, xT, btu_staging);
        default: return launch_rollout_m<MODEL_DI_WRENCH_QUAT>(st, p, integ, lag_mode, layout, B, T, dt, x0, U, nullptr, traj, stride, xT, btu_staging);
    }
}

template <int MODEL, int INTEG>
static hipError_t launch_window_t(hipStream_t st, const FastParams* p, int64_t nwin, int64_t H, double dt, const double* X,
                                  const double* U, const double* lag_start, double* se) {
    // the evaluator always uses the generic form (current / xb, yb branches): it is short-lived and not the benchmark path
    hipLaunchKernelGGL((window_endpoint_kernel<MODEL, INTEG, true>), dim3(nblk(nwin, 256)), dim3(256), 0, st, p, nwin, H, dt, X, U, lag_start, se);
    return hipGetLastError();
}
// scratch: resp [nwin][18], start [nwin][18], phi [9] (device) -- only used for the thruster model with carry_lag
int window_scan_chunk() { return WSCAN_CHUNK; }

hipError_t launch_window_endpoint(hipStream_t st, const FastParams* p, int model, int integ, int64_t N, int64_t H, double dt,
                                  const double* X, const double* U, int carry_lag, const double* d_phi9,
                                  double* d_resp, double* d_start, double* d_se, double* d_total) {
    const int64_t nwin = N - H;
    if (nwin <= 0) return hipSuccess;
    const double* lag_start = nullptr;
    if (model == MODEL_THRUSTER_EULER && carry_lag) {
        if (integ == INTEG_RK4)
            hipLaunchKernelGGL(window_lag_response_kernel<4>, dim3(nblk(nwin, 256)), dim3(256), 0, st, p, nwin, H, U, d_resp);
        else
            hipLaunchKernelGGL(window_lag_response_kernel<1>, dim3(nblk(nwin, 256)), dim3(256), 0, st, p, nwin, H, U, d_resp);
        BROV_LAUNCH_CHECK();
        // d_phi9: [Phi (9) | Phi^WSCAN_CHUNK (9)]; the chunk states live behind the nwin start states in d_start
        const int64_t nchunks = (nwin + WSCAN_CHUNK - 1) / WSCAN_CHUNK;
        double* d_chunk = d_start + nwin * 18;
        hipLaunchKernelGGL(window_lag_chunk_kernel, dim3(nblk(nchunks * 6, 256)), dim3(256), 0, st, nwin, d_phi9, d_resp, d_chunk, d_start, 0);
        BROV_LAUNCH_CHECK();
        hipLaunchKernelGGL(window_lag_scan_kernel, dim3(1), dim3(64), 0, st, nchunks, d_phi9 + 9, d_chunk);
        BROV_LAUNCH_CHECK();
        hipLaunchKernelGGL(window_lag_chunk_kernel, dim3(nblk(nchunks * 6, 256)), dim3(256), 0, st, nwin, d_phi9, d_resp, d_chunk, d_start, 1);
        BROV_LAUNCH_CHECK();
        lag_start = d_start;
    }
    hipError_t e;
    if (model == MODEL_THRUSTER_EULER)
        e = integ == INTEG_RK4 ? launch_window_t<MODEL_THRUSTER_EULER, INTEG_RK4>(st, p, nwin, H, dt, X, U, lag_start, d_se)
                               : launch_window_t<MODEL_THRUSTER_EULER, INTEG_EULER>(st, p, nwin, H, dt, X, U, lag_start, d_se);
    else if (model == MODEL_WRENCH_EULER)
        e = integ == INTEG_RK4 ? launch_window_t<MODEL_WRENCH_EULER, INTEG_RK4>(st, p, nwin, H, dt, X, U, lag_start, d_se)
                               : launch_window_t<MODEL_WRENCH_EULER, INTEG_EULER>(st, p, nwin, H, dt, X, U, lag_start, d_se);
    else if (model == MODEL_WRENCH_QUAT)
        e = integ == INTEG_RK4 ? launch_window_t<MODEL_WRENCH_QUAT, INTEG_RK4>(st, p, nwin, H, dt, X, U, lag_start, d_se)
                               : launch_window_t<MODEL_WRENCH_QUAT, INTEG_EULER>(st, p, nwin, H, dt, X, U, lag_start, d_se);
    else if (model == MODEL_DI_THRUSTER_EULER)
        e = integ == INTEG_RK4 ? launch_window_t<MODEL_DI_THRUSTER_EULER, INTEG_RK4>(st, p, nwin, H, dt, X, U, lag_start, d_se)
                               : launch_window_t<MODEL_DI_THRUSTER_EULER, INTEG_EULER>(st, p, nwin, H, dt, X, U, lag_start, d_se);
    else if (model == MODEL_DI_WRENCH_EULER)
        e = integ == INTEG_RK4 ? launch_window_t<MODEL_DI_WRENCH_EULER, INTEG_RK4>(st, p, nwin, H, dt, X, U, lag_start, d_se)
                               : launch_window_t<MODEL_DI_WRENCH_EULER, INTEG_EULER>(st, p, nwin, H, dt, X, U, lag_start, d_se);
    else
        e = integ == INTEG_RK4 ? launch_window_t<MODEL_DI_WRENCH_QUAT, INTEG_RK4>(st, p, nwin, H, dt, X, U, lag_start, d_se)
                               : launch_window_t<MODEL_DI_WRENCH_QUAT, INTEG_EULER>(st, p, nwin, H, dt, X, U, lag_start, d_se);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, st, nwin, d_se, d_total);
    return hipGetLastError();
}

}  // namespace brov
