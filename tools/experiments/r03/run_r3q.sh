cd $GRAFT_REPO_ROOT
for cfg in "128 320" "256 512" "256 768" "512 768" "512 1024" "1024 1024"; do set -- $cfg
 for f in l3.conv2 l4.conv2 l4.conv3 l3.conv3 l3.conv1 dec_1x1 l4.down; do
  echo "== maxt $1 target $2 $f"; PDF_IG_SPLITK_MAXT=$1 PDF_IG_SPLITK_TARGET=$2 timeout 120 python tools/gemm_bench.py $f 2>&1 | grep "fwd" | sed -e 's/ GF |/ |/' | cut -c1-125
 done; done
