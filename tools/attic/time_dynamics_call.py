import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bluerov2_dynamics_amd.fossen.BlueROV2 import BlueROV2
from bluerov2_dynamics_amd.fossen import BlueROV2_thrust, BlueROV2_wrench
rov = BlueROV2()
x = np.zeros(12); x[2] = 5.0
u = np.array([0.1, 0.1, 0.1, 0, 0.5, 0.5, 0.5, 0.5])
for _ in range(200): rov.dynamics(x, u, 0.02)
t0 = time.perf_counter()
n = 5000
for _ in range(n): xd = rov.dynamics(x, u, 0.02)
print("thruster dynamics(): %.1f us per call" % ((time.perf_counter() - t0) / n * 1e6))
t0 = time.perf_counter()
for _ in range(n): tau = rov.compute_thruster_forces(u, 0.02)
print("compute_thruster_forces(): %.1f us per call" % ((time.perf_counter() - t0) / n * 1e6))
w = BlueROV2_thrust.BlueROV2()
tau = np.array([10, -5, 3, 0.5, -0.4, 0.8.__float__()])
for _ in range(200): w.dynamics(x, tau)
t0 = time.perf_counter()
for _ in range(n): w.dynamics(x, tau)
print("wrench dynamics(): %.1f us per call" % ((time.perf_counter() - t0) / n * 1e6))
# Euler loop like the reference's simulate_physics
t0 = time.perf_counter()
xx = x.copy()
for k in range(2000):
    xx = xx + 0.02 * rov.dynamics(xx, u, 0.02)
print("python Euler loop: %.0f steps/s" % (2000 / (time.perf_counter() - t0)))

# long run: 300 000 calls without an explicit synchronisation in between (the per-call path spins on completion flags)
t0 = time.perf_counter()
xx = x.copy()
for k in range(300000):
    xd = rov.dynamics(xx, u, 0.02)
print("300k calls: %.2f s, last xdot finite: %s" % (time.perf_counter() - t0, bool(np.isfinite(xd).all())))
