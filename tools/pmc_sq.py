"""Wave-cycle breakdown of the GEMM kernels from a rocprofv3 --pmc pass (SQ block, 8 slots):
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \
              SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d DIR -o p -- python3 tools/gemm_bench.py feat
    python tools/pmc_sq.py DIR/p_counter_collection.csv
WAIT_ANY = parked at s_waitcnt / barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing (MI355X_MICROARCH.md, PMC slots)."""
import collections, csv, sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
dur = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'][:56]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVES':
        n[k] += 1
        dur[k] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for k, c in sorted(agg.items()):
    if 'gemm' not in k and 'wino' not in k:
        continue
    wc = c['SQ_WAVE_CYCLES'] or 1
    print("%-56s x%-3d WAIT_ANY %.2f  WAIT_INST_ANY %.2f  ACTIVE_INST %.2f  WAIT_INST_LDS %.2f | MFMA_BUSY/BUSY %.3f" % (
        k, n[k], c['SQ_WAIT_ANY'] / wc, c['SQ_WAIT_INST_ANY'] / wc, c['SQ_ACTIVE_INST_ANY'] / wc, c['SQ_WAIT_INST_LDS'] / wc,
        c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['SQ_BUSY_CYCLES'] or 1)))
    # MFMA pipe utilisation if the counter is per-SIMD cycles summed over the 1024 SIMDs, at the time-averaged clock the
    # wave cycles imply (SQ_WAVE_CYCLES is in quad-cycles per wave): reported raw so the reader can redo the arithmetic
    print("    raw: MFMA_BUSY %.4g  BUSY %.4g  WAVE_CYCLES %.4g  WAVES %.4g  duration %.3f ms/launch" % (
        c['SQ_VALU_MFMA_BUSY_CYCLES'], c['SQ_BUSY_CYCLES'], c['SQ_WAVE_CYCLES'], c['SQ_WAVES'], dur[k] / max(n[k], 1) / 1e6))
