cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3k
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_full_gradient_gpu.py tests/test_trainer_gpu.py tests/test_configs_gpu.py tests/test_modules_gpu.py tests/test_loss_gpu.py -q 2>&1 | grep -E "passed|failed|Error" | tail -5 > gpurun_out/r3k/t_model.log
cat gpurun_out/r3k/t_model.log
bash tools/profile_step.sh r03 > gpurun_out/r3k/profile.log 2>&1
tail -5 gpurun_out/r3k/profile.log
python tools/probe/phase_events.py 32 > gpurun_out/r03_phase_events.txt 2>&1
tail -14 gpurun_out/r03_phase_events.txt
