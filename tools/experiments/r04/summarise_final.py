"""Key figures of a round's measurement set (gpurun_out/<tag>_*) for DESIGN.md section 5 / profiles/README.md.  python summarise_final.py r04"""
import csv, json, sys, os
tag = sys.argv[1] if len(sys.argv) > 1 else 'r04'
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', '..', 'gpurun_out')
def line(name):
    p = os.path.join(root, '%s_%s.json' % (tag, name))
    if not os.path.exists(p):
        return None
    t = open(p).read().strip().splitlines()
    return json.loads(t[-1]) if t else None
d = line('bench_B32_1gpu')
if d:
    r = d['roofline']
    print("headline: %.1f img/s, %.2f ms/step" % (d['value'], d['ms_per_step']))
    print("dominant:", {k: r[k] for k in ('kernel', 'achieved', 'frac', 'launches_per_step', 'ms_per_step', 'algorithmic_gflop_per_launch', 'algorithmic_bytes_per_launch', 'traffic', 'traffic_source')})
    print("all_gemm:", {k: v for k, v in r['all_gemm_kernels'].items() if k != 'per_entry_point'})
    print("step_level:", r['step_level'])
    ps = r['per_symbol']
    for row in (list(ps.items()) if isinstance(ps, dict) else ps)[:10]:
        print("   ", row)
    h = d['roofline_hbm']
    print("hbm pointnet:", h['achieved'], h['frac'], h['ms_per_step'], h['traffic'])
    print("hbm all:", h['all_hbm_bound_entry_points'])
    print("   ", {k: (v['ms'], v['GBs']) for k, v in h['per_entry_point'].items()})
    print("fps:", h['fps']['us_per_pick'], h['fps_single_wave']['us_per_pick'])
    for k, v in d.get('bf16_per_gpu', {}).items():
        if isinstance(v, dict):
            print("bf16 leg", k, v['images_per_s'], v['ms_per_step'], v.get('launch'), v.get('dominant_kernel'))
    print("cpu_baseline:", d.get('cpu_baseline'))
    print("mpjpe:", {k: v for k, v in (d.get('mpjpe') or {}).items() if not isinstance(v, dict)})
for n in ('bench_rgb_encoder_B8', 'bench_B8_1gpu', 'bench_bf16_B32_1gpu', 'bench_bf16_B64_1gpu'):
    x = line(n)
    if x:
        print(n, x['value'], x['ms_per_step'], x.get('launch'))
for name in ('kernel_stats_exclusive', 'kernel_stats_exclusive_bf16_B64'):
    p = os.path.join(root, '%s_%s.csv' % (tag, name))
    if not os.path.exists(p):
        continue
    rows = list(csv.DictReader(open(p)))
    steps = max(int(r['Calls']) for r in rows if 'adam' in r['Name'])
    fam = {}
    for r in rows:
        nm = r['Name']
        k = ('batchnorm' if ('bn_' in nm or 'affine_apply' in nm) else 'implicit GEMM' if 'igemm' in nm else 'weight gradients' if 'wgemm' in nm or 'wgrad' in nm else
             'reductions' if ('reduce_slab' in nm or 'splitk' in nm or 'slab_sum' in nm) else 'winograd transforms' if 'wino' in nm else 'aten / copies' if ('at::' in nm or 'rocclr' in nm) else 'other HIP')
        f = fam.setdefault(k, [0, 0.0]); f[0] += int(r['Calls']) / steps; f[1] += float(r['TotalDurationNs']) / steps / 1e6
    print(name, "steps", steps, "total %.1f ms, %d launches" % (sum(v[1] for v in fam.values()), sum(v[0] for v in fam.values())))
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print("   %-20s %6.0f launches %7.2f ms" % (k, v[0], v[1]))
    for r in rows[:8]:
        print("   %-80s %5.0f %8.3f ms" % (r['Name'][:80], int(r['Calls']) / steps, float(r['TotalDurationNs']) / steps / 1e6))
p = os.path.join(root, '%s_pmc_traffic.json' % tag)
if os.path.exists(p):
    t = json.load(open(p))
    print("gemm family GB/step", t['gemm_family'])
    for k, v in t['symbols'].items():
        if v['bytes_per_launch'] > 100e6:
            print("   ", k, {a: round(b, 1) for a, b in v.items() if isinstance(b, float)})
    for k, v in t.get('bf16_symbols', {}).items():
        if v['bytes_per_launch'] > 100e6:
            print("   bf16", k, {a: round(b, 1) for a, b in v.items() if isinstance(b, float)})
