#!/bin/bash
# GPU box: SQ / GRBM counters of the headline rollout launch, two-wave kernel and one-lane kernel -> gpurun_out/pmc_rollout/
set -o pipefail
root=$PWD
cmd="python3 $root/bench.py --no-edmdc --no-cfg4 --no-cpu --no-ar1 --steps 2 --warmup 1"
for m in 0 1; do
  export BROV2_ROLLOUT_SINGLE_LANE=$m
  tools/pmc_pass.sh gpurun_out/pmc_rollout/m${m}_sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM" -- $cmd
  tools/pmc_pass.sh gpurun_out/pmc_rollout/m${m}_sq2 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_FMA_F64 SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" -- $cmd
  tools/pmc_pass.sh gpurun_out/pmc_rollout/m${m}_grbm "GRBM_GUI_ACTIVE" -- $cmd
done
python3 - <<'PY'
import csv, glob, collections
for m in (0, 1):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_rollout/m{m}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "rollout" in r["Kernel_Name"] and "fill" not in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    print("single_lane =", m)
    for (k, c), v in sorted(acc.items()):
        if len(v) >= 3 or True:
            print(f"   {k:60s} {c:26s} n={len(v):2d} mean={sum(v)/len(v):.6g}")
PY
