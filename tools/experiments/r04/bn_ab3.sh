cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r04_bn_ab3.txt
: > $o
for cfg in "1024 256" "1024 1024" "2048 1024" "2048 2048" "4096 4096" "512 512"; do
  set -- $cfg
  echo "== PDF_BN_PARTIAL_BLOCKS=$1 PDF_BN_PARTIAL_CAP=$2" >> $o
  PDF_BN_PARTIAL_BLOCKS=$1 PDF_BN_PARTIAL_CAP=$2 timeout 300 python tools/experiments/r04/bn_bench.py "C=64 " >> $o 2>&1
  PDF_BN_PARTIAL_BLOCKS=$1 PDF_BN_PARTIAL_CAP=$2 timeout 300 python tools/experiments/r04/bn_bench.py "C=128 " >> $o 2>&1
done
grep -v amdgpu.ids $o
