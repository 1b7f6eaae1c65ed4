cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r06_loss_curve.txt
: > $out
for m in 7 7 0 0 3; do timeout 300 python tools/probe/loss_curve.py 10 $m 2>&1 | grep "x3 mode" >> $out; done
cat $out
