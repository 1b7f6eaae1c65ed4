cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3n
bash tools/profile_step.sh r03 > gpurun_out/r3n/profile.log 2>&1
tail -3 gpurun_out/r3n/profile.log
python tools/probe/phase_events.py 32 > gpurun_out/r03_phase_events.txt 2>&1
cp gpurun_out/r03_pmc_traffic.json profiles/r03_pmc_traffic.json
python bench.py > gpurun_out/r03_bench_B32_1gpu.json 2> gpurun_out/r3n/bench.err
python bench.py --gemm-shapes gpurun_out/r03_gemm_shapes.txt --no-cpu-baseline --no-bf16-legs --no-mpjpe > /dev/null 2>&1
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_B32_1gpu.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['frac'], r['launches_per_step'], r['ms_per_step'], r['traffic'], r['traffic_source'])
print(d['roofline_hbm']['traffic'], d['roofline_hbm']['traffic_source'])
print(d['bf16_per_gpu']['B32']['images_per_s'], d['bf16_per_gpu']['B64']['images_per_s'])"
for i in 1 2; do timeout 1700 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2; done
