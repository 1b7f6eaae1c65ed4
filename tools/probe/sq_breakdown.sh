cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
for f in feat_3x3 l1.conv3 l3.conv2 l2.conv2_3x3; do
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d /tmp/sq_$f -o p -- python3 $root/tools/gemm_bench.py $f > /tmp/sq_$f.log 2>&1 < /dev/null
echo "=== $f"; grep -E "fwd" /tmp/sq_$f.log | cut -c1-200
python3 $root/tools/pmc_sq.py /tmp/sq_$f/p_counter_collection.csv
done
