# round 4: SQ counters of the kernels the fp32 step now runs -- Winograd path on feat / head 3x3, direct kernels on the rest
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( export PDF_BENCH_WINOGRAD=1 SQ_SHAPES="feat_3x3 head_3x3"; bash tools/probe/sq_fp32.sh ) > gpurun_out/r04_sq_fp32.txt 2>&1
( export SQ_SHAPES="l3.conv2 l1.conv3 netR1.3"; bash tools/probe/sq_fp32.sh ) >> gpurun_out/r04_sq_fp32.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/op_census.py > gpurun_out/r04_op_census.txt 2>&1
tail -30 gpurun_out/r04_sq_fp32.txt
