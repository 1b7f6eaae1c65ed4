# SQ / LDS / L2 counters of the x3 GEMM kernels (gemm_x3.hip) on one product shape: three rocprofv3 --pmc passes (8 SQ slots; TCC: FETCH_SIZE alone)
#   bash tools/probe/sq_x3.sh feat.fwd "--variants=0,1 --nprod=6"
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
f=${1:-feat.fwd}; opts=${2:---variants=0,1 --nprod=6}
run() { timeout 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d /tmp/x3_$1 -o p -- python3 $root/tools/x3_bench.py $f $opts > /tmp/x3_$1.log 2>&1 < /dev/null; }
run a "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES"
python3 $root/tools/pmc_sq.py /tmp/x3_a/p_counter_collection.csv
run b "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
run c "FETCH_SIZE"
run d "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
run e "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr"
python3 - <<PY
import collections, csv
for tag in "bcde":
    try:
        rows = list(csv.DictReader(open('/tmp/x3_%s/p_counter_collection.csv' % tag)))
    except Exception as e:
        print(tag, "no data:", e); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in rows:
        k = r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        n[(k, r['Counter_Name'])] += 1
    for k, c in sorted(agg.items()):
        if 'gemm' not in k: continue
        print(tag, "%-60s" % k, "  ".join("%s %.4g/launch" % (cn, v / n[(k, cn)]) for cn, v in sorted(c.items())))
PY
