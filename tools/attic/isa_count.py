#!/usr/bin/env python3
"""Per-kernel ISA summary of a hipcc -S --offload-device-only listing: VGPRs, scratch, and the fp64 / other VALU /
SALU / SMEM / VMEM instruction counts of the hottest loop (the longest basic-block run that ends in a backward branch).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude --offload-device-only -S -o /tmp/k.s csrc/rollout.hip
    python tools/isa_count.py /tmp/k.s rollout_kernelILi0ELi1ELi2
"""
import re
import sys

path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
lines = open(path).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
for n, i0 in enumerate(starts):
    name = lines[i0].split(":")[0]
    if pat not in name:
        continue
    i1 = starts[n + 1] if n + 1 < len(starts) else len(lines)
    body = lines[i0:i1]
    meta = {}
    for l in body:
        m = re.match(r"; (NumVgprs|ScratchSize|NumSgprs|Occupancy): (\d+)", l)
        if m:
            meta[m.group(1)] = int(m.group(2))
    # labels -> index; loops = backward branches
    lab = {}
    for j, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            lab[m.group(1)] = j
    best = None
    for j, l in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+ (\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch (\.LBB\d+_\d+)", l)
        if m and m.group(1) in lab and lab[m.group(1)] < j:
            span = (lab[m.group(1)], j)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    cnt = dict(f64=0, valu=0, salu=0, smem=0, vmem=0, lds=0, wait=0, other=0)
    if best:
        for l in body[best[0]:best[1] + 1]:
            m = re.match(r"\s+([a-z_0-9]+)", l)
            if not m or l.strip().startswith(";") or l.strip().startswith("."):
                continue
            op = m.group(1)
            if op.startswith("v_") and "_f64" in op:
                cnt["f64"] += 1
            elif op.startswith("v_"):
                cnt["valu"] += 1
            elif op.startswith("s_load") or op.startswith("s_buffer_load"):
                cnt["smem"] += 1
            elif op.startswith("s_waitcnt"):
                cnt["wait"] += 1
            elif op.startswith("s_"):
                cnt["salu"] += 1
            elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
                cnt["vmem"] += 1
            elif op.startswith("ds_"):
                cnt["lds"] += 1
            else:
                cnt["other"] += 1
    print(name[:100])
    print("   ", meta, "hot loop:", cnt)
    if len(sys.argv) > 3 and best:     # any third argument: opcode histogram of the hot loop
        import collections
        c = collections.Counter()
        for l in body[best[0]:best[1] + 1]:
            m = re.match(r"\s+([a-z_0-9]+)", l)
            if m and not l.strip().startswith(";") and not l.strip().startswith("."):
                c[m.group(1)] += 1
        print("    " + ", ".join("%s %d" % kv for kv in c.most_common(40)))
