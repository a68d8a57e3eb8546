#!/bin/bash
# GPU box, repo root: tools/r03_fit_pmc.sh TAG [args of tools/time_fit.py] -- kernel trace + PMC passes of the fit() path
set -e -o pipefail
tag=${1:-r03_fit}; shift || true
out=gpurun_out/$tag
mkdir -p $out
root=$(pwd)
cmd="python3 $root/tools/time_fit.py ${@:-10000000 2}"
$cmd > $out/time_fit.txt 2>&1; cat $out/time_fit.txt
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/trace -o run -- $cmd > $root/$out/trace.log 2>&1 )
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv; rm -rf $out/trace
tools/pmc_pass.sh $out/pmc/fetch "FETCH_SIZE" -- $cmd
tools/pmc_pass.sh $out/pmc/write "WRITE_SIZE" -- $cmd
tools/pmc_pass.sh $out/pmc/sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" -- $cmd
tools/pmc_pass.sh $out/pmc/sq2 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU_FMA_F64" -- $cmd
tools/pmc_pass.sh $out/pmc/grbm "GRBM_GUI_ACTIVE" -- $cmd
tools/pmc_pass.sh $out/pmc/tcp "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" -- $cmd || echo "tcp/tcc pass failed (counter names)"
python3 tools/pmc_summary.py $out/pmc "gram_kernel<false>" "gram_kernel<true>" wrows_kernel rows_times_pt_simple_kernel lift_rows_kernel lift_tail_kernel > $out/pmc_summary.json
rm -rf $out/pmc/*/  # keep the logs only
python3 tools/kstats.py $out/kernel_stats.csv
python3 - $out/pmc_summary.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if isinstance(v, dict) and v:
        print(k, {kk: v[kk] for kk in ("mfma_busy_fraction", "cycles_per_mfma", "hbm_read_GB_per_launch_corrected_x2", "hbm_write_GB_per_launch", "SQ_INSTS_VMEM", "SQ_INSTS_MFMA", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCC_HIT_sum", "TCC_MISS_sum") if kk in v})
PY
