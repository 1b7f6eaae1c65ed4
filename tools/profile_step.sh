#!/bin/bash
# Round profile of the default bench command on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_step.sh <tag>     -> gpurun_out/<tag>_kernel_stats.csv, <tag>_pmc_traffic.json, <tag>_gaps.txt
# Kernel trace and PMC counters are separate rocprofv3 runs (counters with --kernel-trace only), as the pool requires.
tag=${1:-prof}
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
run="python3 $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$tag.kt -o p -- $run > /tmp/$tag.kt.log 2>&1 < /dev/null
cp /tmp/$tag.kt/p_kernel_stats.csv $root/gpurun_out/${tag}_kernel_stats.csv
timeout 120 python3 $root/tools/gap_analysis.py /tmp/$tag.kt/p_kernel_trace.csv > $root/gpurun_out/${tag}_gaps.txt 2>&1 < /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/$tag.$c -o p -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/$tag.$c.log 2>&1 < /dev/null
done
timeout 120 python3 $root/tools/pmc_traffic.py /tmp/$tag.FETCH_SIZE /tmp/$tag.WRITE_SIZE 3 $root/gpurun_out/${tag}_pmc_traffic.json < /dev/null | cut -c1-600
ls -la $root/gpurun_out/${tag}_*
