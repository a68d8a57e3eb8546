#!/bin/bash
# Run on the GPU box from the repo root:  tools/profile_round.sh TAG
# Produces under gpurun_out/prof_TAG/: bench.json (default bench.py run), kernel_stats.csv (rocprofv3 --kernel-trace
# --stats of the same command without the host baselines), pmc/<pass>/ (one rocprofv3 --pmc pass per counter group, kernel
# trace only) and pmc_summary.json, fetch_probe.txt (known-bytes calibration of FETCH_SIZE / WRITE_SIZE).
# Copy what should be judged into profiles/.
set -e -o pipefail
tag=${1:-r04}
out=gpurun_out/prof_$tag
mkdir -p $out
root=$(pwd)
python3 bench.py --details $root/$out/bench_details.json > $out/bench.json 2> $out/bench.err
echo "bench done"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/trace -o run -- python3 $root/bench.py --no-cpu --no-variants > $root/$out/trace.log 2>&1 )
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
# per-kernel launch times WITHOUT the first launch of each kernel in the profiled process (code-object load, cold caches): median /
# min / mean -- the figures comparable with bench.py's own ms_per_step (the --stats average includes that first launch)
python3 tools/kernel_times.py $(find $out/trace -name "*kernel_trace.csv" | head -1) > $out/kernel_times.json
echo "trace done"
short="--steps 1 --warmup 1 --edmdc-steps 1 --no-cpu --no-cfg4 --no-ar1 --no-variants --no-fit --kmeans-iters 10"
tools/pmc_pass.sh $out/pmc/fetch "FETCH_SIZE" -- python3 $root/bench.py $short
tools/pmc_pass.sh $out/pmc/write "WRITE_SIZE" -- python3 $root/bench.py $short
tools/pmc_pass.sh $out/pmc/sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" -- python3 $root/bench.py $short
tools/pmc_pass.sh $out/pmc/sq2 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU_FMA_F64" -- python3 $root/bench.py $short
tools/pmc_pass.sh $out/pmc/grbm "GRBM_GUI_ACTIVE" -- python3 $root/bench.py $short
python3 tools/pmc_summary.py $out/pmc > $out/pmc_summary.json
echo "pmc done"
hipcc --offload-arch=gfx950 -O3 tools/fetch_probe.hip -o /tmp/fetch_probe
tools/pmc_pass.sh $out/probe/fetch "FETCH_SIZE" -- /tmp/fetch_probe
tools/pmc_pass.sh $out/probe/write "WRITE_SIZE" -- /tmp/fetch_probe
python3 - $out <<'PY' > $out/fetch_probe.txt
import csv, glob, sys, collections
out = sys.argv[1]
known = {"read16_kernel": 4 << 30, "read8_kernel": 4 << 30, "read8row_kernel": (4 << 30) // (544 * 8) // 4 * 4 * 544 * 8, "write16_kernel": 4 << 30}
acc = collections.defaultdict(list)
for f in glob.glob(out + "/probe/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k in known:
            if k in r["Kernel_Name"]:
                acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
print("kernel            counter      counter_KiB     known_bytes   counter_bytes/known   (-> multiply the counter by 1/ratio)")
for (k, c), v in sorted(acc.items()):
    m = sum(v) / len(v)
    print(f"{k:17s} {c:11s} {m:14.1f} {known[k]:14d}   {m * 1024 / known[k]:.4f}")
PY
cat $out/fetch_probe.txt
