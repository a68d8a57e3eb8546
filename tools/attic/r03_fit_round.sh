#!/bin/bash
# GPU box, repo root: tools/r03_fit_round.sh TAG -- parity of the apply path, bench (tuned and plain W-rows kernel), kernel trace.
set -e -o pipefail
tag=${1:-r03a}
out=gpurun_out/$tag
mkdir -p $out
root=$(pwd)
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "apply_kernels or fit_keeps or bench_prints or edmdc" > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
timeout -k 10 600 python3 bench.py > $out/bench.json 2> $out/bench.err || { tail -30 $out/bench.err; exit 1; }
echo "bench done"
BROV2_APPLY_SIMPLE=1 timeout -k 10 600 python3 bench.py --no-cfg4 --no-cpu --no-variants --no-ar1 --steps 2 --warmup 1 > $out/bench_apply_simple.json 2> $out/bench_simple.err
echo "bench (plain W-rows kernel) done"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/trace -o run -- python3 $root/bench.py --no-cpu --no-cfg4 --steps 3 --warmup 1 > $root/$out/trace.log 2>&1 )
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/trace
python3 tools/kstats.py $out/kernel_stats.csv
