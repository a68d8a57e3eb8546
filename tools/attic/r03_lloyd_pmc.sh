#!/bin/bash
# GPU box: SQ counters of the E-step kernel over the shipped Lloyd loop (300 iterations, sorted order) on config-3 data (tools/time_lloyd.py)
set -e -o pipefail
out=gpurun_out/${1:-r03_lloyd}; mkdir -p $out
root=$(pwd)
cmd="python3 $root/tools/time_lloyd.py 10000000 ${2:-300} default"
tools/pmc_pass.sh $out/pmc/sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE" -- $cmd
tools/pmc_pass.sh $out/pmc/sq2 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_INSTS_VALU_FMA_F64 SQ_ACTIVE_INST_LDS" -- $cmd
tools/pmc_pass.sh $out/pmc/sq3 "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT SQ_WAVES_EQ_64" -- $cmd || echo "sq3 failed"
python3 tools/pmc_summary.py $out/pmc "kmeans_assign_lds_kernel<12>" > $out/pmc_summary.json
rm -rf $out/pmc/*/
python3 - $out/pmc_summary.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if isinstance(v, dict) and v:
        print(k)
        for kk in sorted(v):
            if not kk.startswith("launches_"): print("   %-28s %.4g" % (kk, v[kk]))
PY
