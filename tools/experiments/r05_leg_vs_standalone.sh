cd $GRAFT_REPO_ROOT
for f in "--no-collective-path" ""; do
echo "== legs with: $f"
timeout 600 python bench.py --no-cpu-baseline --no-mpjpe $f 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('fp32', d['value']); print({k: (v['images_per_s'], v['launch'], v['auto_choice_ms']) for k, v in d['bf16_per_gpu'].items() if isinstance(v, dict)})"
done
