cd $GRAFT_REPO_ROOT
for e in "X=1" "PDFNET_LAZY_TRUNK_BN=0" "PDFNET_WINOGRAD_KEEP_V=0" "PDFNET_LAZY_TRUNK_BN=0 PDFNET_WINOGRAD_KEEP_V=0"; do
echo "== $e"
for r in 1 2; do env $e timeout 300 python bench.py --config rgb-encoder --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  rgb-encoder B=8', d['value'], d['ms_per_step'])"; done
env $e timeout 300 python bench.py --batch 8 --no-cpu-baseline --no-mpjpe --no-collective-path --no-roofline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  full step B=8', d['value'], d['ms_per_step'])"
done
