#!/bin/bash
# tools/isa_variant.sh NAME [-DFLAG ...]  -- device listing of csrc/rollout.hip with the given flags, instruction histogram of
# the time loop of the benchmark kernel (thruster model, RK4, TPB layout, reference vehicle) and its register usage.
name=$1; shift
mkdir -p /tmp/isa
cd "$(dirname "$0")/../bluerov2_dynamics_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DBROV2_BUILDING=1 "$@" --offload-device-only -S -o /tmp/isa/$name.s rollout.hip 2>/dev/null
python3 - "$name" <<'PY'
import re, sys, collections
name = sys.argv[1]
lines = open(f"/tmp/isa/{name}.s").read().split("\n")
pat = "rollout_kernelILi0ELi1ELi2ELi0ELb0ELb0E"
i0 = next(i for i, l in enumerate(lines) if l.startswith("_ZN4brov14" + pat) and l.rstrip().endswith(":") or (l.startswith("_ZN4brov14" + pat) and ":" in l))
i1 = next(i for i in range(i0, len(lines)) if "s_endpgm" in lines[i])
body = lines[i0:i1]
meta = {}
for l in lines[i1:i1 + 80]:
    m = re.match(r"; (NumVgprs|NumAgprs|TotalNumVgprs|NumSgprs|ScratchSize|Occupancy|SGPRSpill|sgpr_spill_count|vgpr_spill_count)\S*: (\d+)", l.strip())
    if m: meta[m.group(1)] = int(m.group(2))
lab = {}
for j, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: lab[m.group(1)] = j
best = None
for j, l in enumerate(body):
    m = re.match(r"\s+s_cbranch_\w+ (\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch (\.LBB\d+_\d+)", l)
    if m and m.group(1) in lab and lab[m.group(1)] < j:
        sp = (lab[m.group(1)], j)
        if best is None or sp[1] - sp[0] > best[1] - best[0]: best = sp
c = collections.Counter()
for l in body[best[0]:best[1] + 1]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."): continue
    op = t.split()[0]
    if op.endswith("_f64") or op.endswith("_f64_e32") or op.endswith("_f64_e64"):
        if op.startswith("v_cmp") : c["valu_other"] += 1
        else: c["f64"] += 1
    elif op.startswith("v_accvgpr"): c["accvgpr"] += 1
    elif op.startswith("v_readlane") or op.startswith("v_writelane"): c["lane_spill"] += 1
    elif op.startswith("v_mov_b64") or op.startswith("v_mov_b32"): c["v_mov"] += 1
    elif op.startswith("v_"): c["valu_other"] += 1
    elif op.startswith("s_load"): c["smem"] += 1
    elif op.startswith("s_waitcnt"):
        c["wait"] += 1
        if "lgkmcnt" in t: c["wait_lgkm"] += 1
    elif op.startswith("s_mov"): c["s_mov"] += 1
    elif op.startswith("s_"): c["salu_other"] += 1
    elif op.startswith("ds_"): c["lds"] += 1
    elif op.startswith("global_") or op.startswith("buffer_"): c["vmem"] += 1
    else: c["other"] += 1
valu = c["f64"] + c["accvgpr"] + c["lane_spill"] + c["v_mov"] + c["valu_other"]
slots = valu + c["smem"] + c["s_mov"] + c["salu_other"] + c["lds"] + c["vmem"]
print(f"{name:28s} f64={c['f64']} valu={valu} (mov {c['v_mov']} acc {c['accvgpr']} lane {c['lane_spill']} other {c['valu_other']}) "
      f"salu={c['s_mov']+c['salu_other']} (s_mov {c['s_mov']}) smem={c['smem']} lds={c['lds']} vmem={c['vmem']} waits={c['wait']} (lgkm {c['wait_lgkm']}) "
      f"issue={slots}  regs={meta}")
PY
