#!/bin/bash
export BROV2_LIBRARY=$PWD/build_variants/blk/libbrov2.so
tools/kstats_run.sh gpurun_out/ks_blk -- python3 $PWD/tools/lloyd_balance.py | grep -E "assign_lds_kernel<12, true"
grep "Lloyd" gpurun_out/ks_blk/trace.log
