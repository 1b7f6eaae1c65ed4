# what the driver runs at round end, on a fresh box: smoke(), the default bench line, the torchrun launch of bench.py (one rank)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r06_driver_like.txt
: > $out
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2 ) >> $out
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('direct:', d['value'], d['ms_per_step'], d['n_gpus'], d['steps'], d['warmup'], d['roofline']['frac'], d['roofline']['traffic'])" ) >> $out 2>&1
( timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('torchrun:', d['value'], d['ms_per_step'], d['n_gpus'], d['config'].get('rccl_ranks'))" ) >> $out 2>&1
cat $out
