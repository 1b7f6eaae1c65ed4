#!/bin/bash
# r04 A/B: LDS-DMA 64x64 implicit GEMM (PDF_IG_DMA = variant + 1) vs the register-staged igemm_nt<64,64>: correctness, then per-layer TFLOP/s
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for v in 2 1; do
  echo "=== correctness PDF_IG_DMA=$v"
  PDF_IG_DMA=$v timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "conv2d or linear or deconv" 2>&1 | tail -5
done
for v in 0 1 2 3 4; do
  echo "=== gemm_bench PDF_IG_DMA=$v"
  PDF_IG_DMA=$v timeout 600 python tools/gemm_bench.py 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/gemm_bench_dma$v.txt | cut -c1-200
done
