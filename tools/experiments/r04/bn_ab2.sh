cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r04_bn_ab2.txt
: > $o
for il in 0 1 2 3; do
  echo "== PDF_BN_INTERLEAVE=$il" >> $o
  PDF_BN_INTERLEAVE=$il timeout 300 python tools/experiments/r04/bn_bench.py >> $o 2>&1
done
echo "== PDF_BN_INTERLEAVE=3 PDF_BN_PARTIAL_BLOCKS=2048 PDF_BN_PARTIAL_CAP=1024 PDF_BN_APPLY_BLOCKS=2048" >> $o
PDF_BN_INTERLEAVE=3 PDF_BN_PARTIAL_BLOCKS=2048 PDF_BN_PARTIAL_CAP=1024 PDF_BN_APPLY_BLOCKS=2048 timeout 300 python tools/experiments/r04/bn_bench.py >> $o 2>&1
PDF_BN_INTERLEAVE=3 timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "batchnorm or bn" 2>&1 | tail -3
grep -v amdgpu.ids $o | grep -E "==|all:|R=1048576|R=524288|R=262144"
