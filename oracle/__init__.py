"""CPU oracle for the BlueROV2 rollout + EDMDc hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``bluerov2_dynamics_amd/`` may import,
link or execute anything in this directory: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and there
only as the checker / the CPU baseline, never as the thing measured or shipped.

Parity is PINNED: every function here is checked (tests/test_oracle_golden.py)
against fixtures in ``tests/golden/`` that ``tools/gen_golden.py`` produced by
importing the unmodified reference in the build container.
"""
