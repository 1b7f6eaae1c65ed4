cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3j
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "gather_sub or group_then or statistics or batchnorm" 2>&1 | tail -4 > gpurun_out/r3j/t_ops.log
B="--no-cpu-baseline --no-bf16-legs --no-mpjpe --no-roofline --steps 20 --warmup 6"
python bench.py $B > gpurun_out/r3j/b1.json 2>/dev/null
python bench.py $B > gpurun_out/r3j/b2.json 2>/dev/null
PDFNET_GATHER_SORTED=0 python bench.py $B > gpurun_out/r3j/b_atomic.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
for f in feat_3x3; do
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d /tmp/sqa_$f -o p -- python3 $root/tools/gemm_bench.py $f > /tmp/sqa_$f.log 2>&1 < /dev/null
python3 $root/tools/pmc_sq.py /tmp/sqa_$f/p_counter_collection.csv
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/sqb_$f -o p -- python3 $root/tools/gemm_bench.py $f > /tmp/sqb_$f.log 2>&1 < /dev/null
python3 - <<PY
import collections, csv
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open('/tmp/sqb_$f/p_counter_collection.csv')):
    agg[r['Kernel_Name'][:60]][r['Counter_Name']] += float(r['Counter_Value'])
for k, c in sorted(agg.items()):
    if 'gemm' not in k: continue
    wc = c['SQ_WAVE_CYCLES'] or 1
    print("%-60s VALU %.3g LDS %.3g VMEM %.3g | per wave-cycle: LDS_IDX_ACTIVE %.3f BANK_CONFLICT %.3f ACTIVE_LDS %.3f ACTIVE_VALU %.3f" % (k, c['SQ_INSTS_VALU'], c['SQ_INSTS_LDS'], c['SQ_INSTS_VMEM'], c['SQ_LDS_IDX_ACTIVE']/wc, c['SQ_LDS_BANK_CONFLICT']/wc, c['SQ_ACTIVE_INST_LDS']/wc, c['SQ_ACTIVE_INST_VALU']/wc))
PY
done > $root/gpurun_out/r3j/sq_fp32.txt 2>&1
cd $root
cat gpurun_out/r3j/t_ops.log gpurun_out/r3j/sq_fp32.txt
for f in gpurun_out/r3j/b*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'])"; done
