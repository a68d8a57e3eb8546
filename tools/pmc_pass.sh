#!/bin/bash
# usage: tools/pmc_pass.sh OUTDIR "COUNTER1 COUNTER2 ..." -- python3 bench.py ...
# One rocprofv3 counter pass (kernel trace only, as gpurun requires); run from the repo root on the GPU box.
set -e
out=$1; shift
ctrs=$1; shift
shift   # --
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out" -o run -- "$@" > "$GRAFT_REPO_ROOT/$out.log" 2>&1
