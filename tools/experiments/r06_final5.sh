cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/run_final.sh r06 > gpurun_out/r06_final_run.log 2>&1
tail -8 gpurun_out/r06_final_run.log
