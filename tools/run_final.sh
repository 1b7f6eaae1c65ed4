# Round-end measurement set (run through gpurun from the repo root): tests, default bench line, per-shape table, phase events, profile
cd $GRAFT_REPO_ROOT
tag=${1:-r03}
mkdir -p gpurun_out
bash tools/profile_step.sh $tag > gpurun_out/${tag}_profile.log 2>&1
cd $GRAFT_REPO_ROOT
# the bench line looks its `traffic` up in the newest profiles/*_pmc_traffic.json taken from the SAME kernel sources: the one just made
cp gpurun_out/${tag}_pmc_traffic.json profiles/${tag}_pmc_traffic.json
timeout 900 python bench.py --gemm-shapes gpurun_out/${tag}_gemm_shapes.txt --hbm-shapes gpurun_out/${tag}_hbm_shapes.txt > gpurun_out/${tag}_bench_B32_1gpu.json 2> gpurun_out/${tag}_bench_err.txt
timeout 300 python tools/probe/phase_events.py 32 > gpurun_out/${tag}_phase_events.txt 2>&1
timeout 300 python tools/mesh_bench.py 32 2>&1 | grep -v "^W\|amdgpu.ids" > gpurun_out/${tag}_mesh_bench.txt
for a in "f32 32" "bf16 32" "bf16 64"; do timeout 250 python3 tools/host_time.py $a 2>&1 | grep -v "^W\|amdgpu.ids" | tail -4; done > gpurun_out/${tag}_host_time.txt 2>&1
timeout 300 python3 tools/op_census.py 2>&1 | grep -v "^W\|amdgpu.ids" > gpurun_out/${tag}_op_census.txt
timeout 300 python bench.py --config rgb-encoder --no-cpu-baseline > gpurun_out/${tag}_bench_rgb_encoder_B8.json 2>/dev/null
timeout 300 python bench.py --batch 8 --no-cpu-baseline --no-mpjpe --no-collective-path > gpurun_out/${tag}_bench_B8_1gpu.json 2>/dev/null
timeout 300 python bench.py --dtype bf16 --batch 32 --steps 30 --warmup 10 --no-cpu-baseline --no-mpjpe > gpurun_out/${tag}_bench_bf16_B32_1gpu.json 2>/dev/null
timeout 300 python bench.py --dtype bf16 --batch 64 --steps 30 --warmup 10 --no-cpu-baseline --no-mpjpe > gpurun_out/${tag}_bench_bf16_B64_1gpu.json 2>/dev/null
tail -c 1500 gpurun_out/${tag}_bench_B32_1gpu.json | head -c 600; echo
python - <<PY
import json
d = json.loads(open('gpurun_out/${tag}_bench_B32_1gpu.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['traffic'])
print(d['roofline']['all_gemm_kernels']['achieved'], d['roofline']['step_level'])
print({k: (v['images_per_s'], v['ms_per_step']) for k, v in d['bf16_per_gpu'].items() if isinstance(v, dict)})
print(d['cpu_baseline'])
PY
