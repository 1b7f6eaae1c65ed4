cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep "passed\|failed\|error" | tail -3 > gpurun_out/r06_final_pytest.txt
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -1 ) >> gpurun_out/r06_final_pytest.txt
( timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['step_level']['executed_frac'])" ) >> gpurun_out/r06_final_pytest.txt 2>&1
cat gpurun_out/r06_final_pytest.txt
