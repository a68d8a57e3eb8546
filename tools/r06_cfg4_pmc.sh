#!/bin/bash
# GPU box, repo root: tools/r06_cfg4_pmc.sh [TAG]  -- counters of config 4's caller-layout kernels: rollout_pair_kernel<RK4, BTU>
# (stored rollout of the 2^20-trajectory ensemble) and fill_ar1_btu_kernel<8> (its AR(1) command stream), one rocprofv3 --pmc pass per
# counter group with the kernel trace only, summary -> gpurun_out/prof_TAG/cfg4_pmc_summary.json
set -e -o pipefail
tag=${1:-r06}
out=gpurun_out/prof_$tag/cfg4
mkdir -p $out
root=$(pwd)
tools/pmc_pass.sh $out/fetch "FETCH_SIZE" -- python3 $root/tools/cfg4_kernels.py
tools/pmc_pass.sh $out/write "WRITE_SIZE" -- python3 $root/tools/cfg4_kernels.py
tools/pmc_pass.sh $out/sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" -- python3 $root/tools/cfg4_kernels.py
tools/pmc_pass.sh $out/sq2 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU_FMA_F64" -- python3 $root/tools/cfg4_kernels.py
tools/pmc_pass.sh $out/sq3 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" -- python3 $root/tools/cfg4_kernels.py || echo "sq3 group not available"
tools/pmc_pass.sh $out/grbm "GRBM_GUI_ACTIVE" -- python3 $root/tools/cfg4_kernels.py
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/trace -o run -- python3 $root/tools/cfg4_kernels.py > $root/$out/trace.log 2>&1 )
python3 tools/kernel_times.py $(find $out/trace -name "*kernel_trace.csv" | head -1) > $out/kernel_times.json
python3 tools/pmc_summary.py $out "rollout_pair_kernel<1, 0," "fill_ar1_btu_kernel" > gpurun_out/prof_$tag/cfg4_pmc_summary.json
cat gpurun_out/prof_$tag/cfg4_pmc_summary.json
