# the round's final set on the final tree: full GPU suite, then tools/run_final.sh (profiles + bench lines)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r06_final_pytest.txt
bash tools/run_final.sh r06 > gpurun_out/r06_final_run.log 2>&1
cat gpurun_out/r06_final_pytest.txt; tail -8 gpurun_out/r06_final_run.log
