# kernel trace of the default bench step + stream timeline (critical-path view) -> gpurun_out/<tag>_streams.txt
tag=${1:-trace}
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/$tag.kt -o p -- python3 $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-mpjpe > /tmp/$tag.kt.log 2>&1 < /dev/null
timeout 120 python3 $root/tools/stream_timeline.py /tmp/$tag.kt/p_kernel_trace.csv > $root/gpurun_out/${tag}_streams.txt 2>&1 < /dev/null
