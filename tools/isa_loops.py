#!/usr/bin/env python3
"""Instruction histogram of the (up to three) largest top-level loops of one kernel in a hipcc device listing.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -DBROV2_BUILDING=1 --offload-device-only -S -o /tmp/r.s csrc/rollout.hip
    python tools/isa_loops.py /tmp/r.s _ZN4brov19rollout_pair_kernelILi1ELi2ELi0ELb0ELb0E     # RK4, TPB: body loop, thrust loop

Prints one line per loop with a Python dict: f64 (fp64 VALU arithmetic, of which fmac), vmov, acc (v_accvgpr_*), lane
(v_readlane / v_writelane), valu_other, salu, s_mov, smem, lds, vmem, wait, barrier.  The counts are static: blocks that
run rarely (the full sin/cos refresh every 64 steps, the range-extension loops of trig_delta) are included."""
import re, sys, collections
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
i0 = next(i for i, l in enumerate(lines) if l.startswith(pat) and ":" in l)
i1 = next(i for i in range(i0, len(lines)) if "s_endpgm" in lines[i])
body = lines[i0:i1]
lab = {}
for j, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: lab[m.group(1)] = j
loops = []
for j, l in enumerate(body):
    m = re.match(r"\s+s_cbranch_\w+ (\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch (\.LBB\d+_\d+)", l)
    if m and m.group(1) in lab and lab[m.group(1)] < j:
        loops.append((lab[m.group(1)], j))
loops.sort(key=lambda s: s[0] - s[1])
def hist(a, b):
    c = collections.Counter()
    for l in body[a:b + 1]:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."): continue
        op = t.split()[0]
        if re.search(r"_f64(_e32|_e64)?$", op) and not op.startswith("v_cmp"): c["f64"] += 1; c["fmac"] += op.startswith("v_fmac")
        elif op.startswith("v_accvgpr"): c["acc"] += 1
        elif op.startswith("v_readlane") or op.startswith("v_writelane"): c["lane"] += 1
        elif op.startswith("v_mov"): c["vmov"] += 1
        elif op.startswith("v_"): c["valu_other"] += 1
        elif op.startswith("s_load"): c["smem"] += 1
        elif op.startswith("s_waitcnt"): c["wait"] += 1
        elif op.startswith("s_mov"): c["s_mov"] += 1
        elif op.startswith("s_barrier"): c["barrier"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith("global_"): c["vmem"] += 1
        else: c["other"] += 1
    return dict(c)
# top-level loops only (not nested in a bigger one already printed)
shown = []
for a, b in loops:
    if any(a >= A and b <= Bq for A, Bq in shown): continue
    shown.append((a, b))
    print(f"loop lines {a}-{b} ({b-a+1} lines):", hist(a, b))
    if len(shown) >= 3: break
