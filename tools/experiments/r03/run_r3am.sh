cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3am
PDFNET_LAZY_SA_BN=0 timeout 400 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-mpjpe --no-bf16-legs --gemm-shapes gpurun_out/r3am/shapes0.txt > gpurun_out/r3am/b0.json 2>/dev/null
PDFNET_LAZY_SA_BN=1 timeout 400 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-mpjpe --no-bf16-legs --gemm-shapes gpurun_out/r3am/shapes1.txt > gpurun_out/r3am/b1.json 2>/dev/null
