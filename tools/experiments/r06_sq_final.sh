# SQ counters of the final kernels: the x3 NT kernel (split accumulators) on the `feat` forward product, the native 64x64 / 128x128 kernels on their layer shapes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( bash tools/probe/sq_x3.sh feat.fwd "--variants=0,1 --nprod=6" ) 2>&1 | grep -v "^W\|amdgpu.ids" > gpurun_out/r06_x3_sq_final.txt
( SQ_SHAPES="l3.conv2 l1.conv3 l3.conv3" bash tools/probe/sq_fp32.sh ) 2>&1 | grep -v "^W\|amdgpu.ids" > gpurun_out/r06_sq_fp32.txt
head -40 gpurun_out/r06_x3_sq_final.txt | cut -c1-250; head -40 gpurun_out/r06_sq_fp32.txt | cut -c1-250
