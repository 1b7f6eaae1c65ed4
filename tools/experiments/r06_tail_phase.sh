cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/probe/phase_events.py 32 2>&1 | grep -v "^W\|amdgpu.ids" | tail -20 > gpurun_out/r06_tail_phase.txt
cat gpurun_out/r06_tail_phase.txt
