#!/bin/bash
# r04: LDS-DMA bf16 implicit GEMM (PDF_BF16_DMA = variant + 1) vs the register-staged igemm_bf16_kernel: correctness, then TFLOP/s per layer
cd "$GRAFT_REPO_ROOT" || exit 1
for v in 1 2; do
  echo "=== correctness PDF_BF16_DMA=$v"
  PDF_BF16_DMA=$v PDF_BF16_DMA_TILES=3 timeout 900 python -m pytest tests/test_bf16_gpu.py -x -q 2>&1 | tail -4
done
for v in 0 1 2 3 4; do
  echo "=== gemm_bench bf16 PDF_BF16_DMA=$v"
  PDF_BENCH_BF16=1 PDF_BF16_DMA=$v PDF_BF16_DMA_TILES=3 timeout 600 python tools/gemm_bench.py 3x3 2>&1 | grep -v "amdgpu.ids" | cut -c1-120
done
