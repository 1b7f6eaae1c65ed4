cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_x3_gpu.py -q -s -rA 2>&1 | grep -v "^W\|amdgpu.ids" > gpurun_out/r06_x3_tests.txt
grep -n "NT \|TN \|rms\|AssertionError\|passed\|failed" gpurun_out/r06_x3_tests.txt | cut -c1-250
