cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 300 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-roofline --no-mpjpe --no-bf16-legs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
run X=0
run PDF_IG_T128=300
run PDF_IG_T128=1000
run PDF_IG_SHORTK=0
run PDF_IG_HALO_MINC=128
run PDF_IG_SPLITK_MAXT=128
run PDF_IG_SPLITK_MAXT=512 PDF_IG_SPLITK_TARGET=768
run PDF_WG_MINROWS=256
done
