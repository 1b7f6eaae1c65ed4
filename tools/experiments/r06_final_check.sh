# full GPU suite on the final tree + the repaired PDF_IG_DMA switch (it faulted on batched launches before the dispatch guard)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_final_pytest.txt
for d in 0 1 2; do
  echo "PDF_IG_DMA=$d" >> gpurun_out/r06_ig_dma_fixed.txt
  PDF_IG_DMA=$d timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mpjpe --no-roofline --no-bf16-legs --no-native-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/r06_ig_dma_fixed.txt 2>&1
done
cat gpurun_out/r06_final_pytest.txt gpurun_out/r06_ig_dma_fixed.txt
