# weight-gradient side streams at normal (0) / low (1) / high (-1) HIP priority, alternating within one call
cd ${GRAFT_REPO_ROOT:-$PWD}
python - <<'PY'
import ctypes
from pdfnet_amd import hip
L = hip.lib()
lo, hi = ctypes.c_int(), ctypes.c_int()
L.pdf_stream_create(None, 0, ctypes.byref(lo), ctypes.byref(hi))
print("hipDeviceGetStreamPriorityRange: least %d, greatest %d" % (lo.value, hi.value))
PY
b="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-mpjpe --no-bf16-legs --no-roofline --no-native-leg --no-collective-path"
for r in 1 2; do
  for pr in 0 1 -1; do
    v=$(PDFNET_WGRAD_STREAM_PRIORITY=$pr timeout 300 $b 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['median_step_ms'])")
    echo "round $r  PDFNET_WGRAD_STREAM_PRIORITY=$pr  img/s, ms/step, median ms: $v"
  done
done
