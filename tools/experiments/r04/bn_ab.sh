# BatchNorm block-count sweep + the unit tests of what changed (invert_index, upsample backward)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "gather_sub or pool_upsample or batchnorm or bn" 2>&1 | tail -3
o=gpurun_out/r04_bn_ab.txt
: > $o
for cfg in "1024 256 4096" "2048 256 4096" "4096 1024 4096" "1024 256 8192" "1024 256 2048" "2048 512 8192"; do
  set -- $cfg
  echo "== PDF_BN_PARTIAL_BLOCKS=$1 PDF_BN_PARTIAL_CAP=$2 PDF_BN_APPLY_BLOCKS=$3" >> $o
  PDF_BN_PARTIAL_BLOCKS=$1 PDF_BN_PARTIAL_CAP=$2 PDF_BN_APPLY_BLOCKS=$3 timeout 300 python tools/experiments/r04/bn_bench.py >> $o 2>&1
done
grep -E "==|all:" $o
