"""Constants of the 4-DOF (surge, sway, heave, yaw) BlueROV model behind the PINc physics loss
(reference: fossen/parameters.py, values as published there; consumed by bluerov_torch.bluerov_compute)."""

m = 11.4                         # mass [kg]
g = 9.82                         # gravity [m/s^2]
F_bouy = 1026 * 0.0115 * g       # buoyancy [N]

# added mass
X_ud, Y_vd, Z_wd = -2.6, -18.5, -13.3
K_pd, M_qd, N_rd = -0.054, -0.0173, -0.28
# inertia
I_xx, I_yy, I_zz = 0.21, 0.245, 0.245
# linear damping
X_u, Y_v, Z_w = -0.09, -0.26, -0.19
K_p, M_q, N_r = -0.895, -0.287, -4.64
# quadratic damping
X_uc, Y_vc, Z_wc = -34.96, -103.25, -74.23
K_pc, M_qc, N_rc = -0.084, -0.028, -0.43

z_b = -0.1                       # CB above CG [m]
