cd $GRAFT_REPO_ROOT
timeout 300 python tools/x3_bench.py --nprod=6 --variants=0,1,2 > gpurun_out/x3_nt_5.txt 2>&1
timeout 900 python -m pytest tests/test_headline_gpu.py tests/test_ops_gpu.py tests/test_dualgraph_golden_gpu.py tests/test_trainer_gpu.py -x -q -m gpu -k "heaviest or winograd or transformed or share or dualgraph or taped" > gpurun_out/t5_tests.txt 2>&1
for x in 0 1; do PDF_X3=$x timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mpjpe --no-bf16-legs --no-roofline --no-collective-path 2>/dev/null | tail -1 | cut -c1-300 > gpurun_out/t5_bench_x3_$x.txt; done
