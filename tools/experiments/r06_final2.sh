cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_x3_gpu.py -q -s 2>&1 | grep -v "^W\|amdgpu.ids" | tail -40 > gpurun_out/r06_x3_tests.txt
bash tools/run_final.sh r06 > gpurun_out/r06_final_run.log 2>&1
cat gpurun_out/r06_x3_tests.txt; tail -12 gpurun_out/r06_final_run.log
