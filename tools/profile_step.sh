#!/bin/bash
# Round profile of the default bench command on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_step.sh <tag>     -> gpurun_out/<tag>_kernel_stats.csv, <tag>_pmc_traffic.json, <tag>_gaps.txt
# Kernel trace and PMC counters are separate rocprofv3 runs (counters with --kernel-trace only), as the pool requires.
tag=${1:-prof}
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
run="python3 $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-native-leg --no-collective-path"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$tag.kt -o p -- $run > /tmp/$tag.kt.log 2>&1 < /dev/null
cp /tmp/$tag.kt/p_kernel_stats.csv $root/gpurun_out/${tag}_kernel_stats.csv
timeout 120 python3 $root/tools/gap_analysis.py /tmp/$tag.kt/p_kernel_trace.csv > $root/gpurun_out/${tag}_gaps.txt 2>&1 < /dev/null
timeout 120 python3 $root/tools/stream_timeline.py /tmp/$tag.kt/p_kernel_trace.csv > $root/gpurun_out/${tag}_streams.txt 2>&1 < /dev/null
# exclusive durations: side streams off
PDFNET_SIDE_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$tag.kx -o p -- $run > /tmp/$tag.kx.log 2>&1 < /dev/null
cp /tmp/$tag.kx/p_kernel_stats.csv $root/gpurun_out/${tag}_kernel_stats_exclusive.csv
for c in FETCH_SIZE WRITE_SIZE; do
  PDFNET_SIDE_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/$tag.$c -o p -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs --no-native-leg --no-collective-path > /tmp/$tag.$c.log 2>&1 < /dev/null
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/$tag.fps.$c -o p -- python3 $root/tools/hbm_bench.py fps > /tmp/$tag.fps.$c.log 2>&1 < /dev/null
done
timeout 120 python3 $root/tools/pmc_traffic.py /tmp/$tag.FETCH_SIZE /tmp/$tag.WRITE_SIZE 3 $root/gpurun_out/${tag}_pmc_traffic.json $tag < /dev/null | cut -c1-400
# the bf16 kernels (BASELINE configs[4] per rank: --dtype bf16 --batch 64): their own two PMC passes, merged into the same file
for c in FETCH_SIZE WRITE_SIZE; do
  PDFNET_SIDE_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/$tag.bf16.$c -o p -- python3 $root/bench.py --dtype bf16 --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-mpjpe --no-collective-path > /tmp/$tag.bf16.$c.log 2>&1 < /dev/null
done
timeout 120 python3 $root/tools/pmc_traffic.py --bf16 /tmp/$tag.bf16.FETCH_SIZE /tmp/$tag.bf16.WRITE_SIZE 3 $root/gpurun_out/${tag}_pmc_traffic.json < /dev/null | cut -c1-300
# ... and the B=32 per-rank step (configs[3]) with its own two passes (the B32 leg's traffic must come from its own run)
for c in FETCH_SIZE WRITE_SIZE; do
  PDFNET_SIDE_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/$tag.bf16b32.$c -o p -- python3 $root/bench.py --dtype bf16 --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-mpjpe --no-collective-path > /tmp/$tag.bf16b32.$c.log 2>&1 < /dev/null
done
timeout 120 python3 $root/tools/pmc_traffic.py --bf16 /tmp/$tag.bf16b32.FETCH_SIZE /tmp/$tag.bf16b32.WRITE_SIZE 3 $root/gpurun_out/${tag}_pmc_traffic.json 32 < /dev/null | cut -c1-300
PDFNET_SIDE_STREAMS=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$tag.kxb -o p -- python3 $root/bench.py --dtype bf16 --batch 64 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-mpjpe --no-collective-path > /tmp/$tag.kxb.log 2>&1 < /dev/null
cp /tmp/$tag.kxb/p_kernel_stats.csv $root/gpurun_out/${tag}_kernel_stats_exclusive_bf16_B64.csv
timeout 120 python3 $root/tools/pmc_traffic.py /tmp/$tag.fps.FETCH_SIZE /tmp/$tag.fps.WRITE_SIZE 1 $root/gpurun_out/${tag}_pmc_fps.json $tag < /dev/null | grep fps_kernel
ls -la $root/gpurun_out/${tag}_*
