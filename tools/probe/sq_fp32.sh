# SQ counters of the fp32 GEMM kernels on four layer shapes (two passes of 8 counters).
# PDF_BENCH_WINOGRAD=1 in the environment: the stride-1 3x3 layers take the Winograd path the train step takes (transform kernels are listed too).
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
# fp32 kernels
for f in ${SQ_SHAPES:-feat_3x3 l3.conv2 l1.conv3 netR1.3}; do
python3 $root/tools/gemm_bench.py $f 2>&1 | grep fwd | cut -c1-160
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d /tmp/sqa_$f -o p -- python3 $root/tools/gemm_bench.py $f > /tmp/sqa_$f.log 2>&1 < /dev/null
python3 $root/tools/pmc_sq.py /tmp/sqa_$f/p_counter_collection.csv
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/sqb_$f -o p -- python3 $root/tools/gemm_bench.py $f > /tmp/sqb_$f.log 2>&1 < /dev/null
python3 - <<PY
import collections, csv
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open('/tmp/sqb_$f/p_counter_collection.csv')):
    k = r['Kernel_Name'][:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, c in sorted(agg.items()):
    if 'gemm' not in k and 'wino' not in k: continue
    wc = c['SQ_WAVE_CYCLES'] or 1
    print("%-60s VALU %.3g LDS %.3g VMEM %.3g | per wave-cycle: LDS_IDX_ACTIVE %.3f BANK_CONFLICT %.3f ACTIVE_LDS %.3f ACTIVE_VALU %.3f" % (k, c['SQ_INSTS_VALU'], c['SQ_INSTS_LDS'], c['SQ_INSTS_VMEM'], c['SQ_LDS_IDX_ACTIVE']/wc, c['SQ_LDS_BANK_CONFLICT']/wc, c['SQ_ACTIVE_INST_LDS']/wc, c['SQ_ACTIVE_INST_VALU']/wc))
PY
done
