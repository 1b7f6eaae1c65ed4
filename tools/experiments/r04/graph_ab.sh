#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { env $1 timeout 600 python bench.py --no-cpu-baseline --no-bf16-legs --no-mpjpe --no-roofline --steps 12 --warmup 4 $2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-44s %-8s %.1f img/s  %.2f ms/step' % (sys.argv[1], sys.argv[2], d['value'], d['ms_per_step']))" "$1" "$2"; }
run "A=0" ""
run "A=0" "--graph"
run "PDF_GRAPH_WGRAD_STREAM=1" "--graph"
run "A=0" ""
python tools/host_time.py 2>&1 | tail -3 | head -1
