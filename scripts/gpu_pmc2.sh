# usage (through gpurun): bash scripts/gpu_pmc2.sh <tag> [batch] [lib]
# one rocprofv3 --pmc pass (8 SQ counters) of the serial-mode bench: where each kernel's wave time goes
cd $GRAFT_REPO_ROOT
TAG=$1; BATCH=${2:-8}; LIB=${3:-$GRAFT_REPO_ROOT/hesaff_amd/libhesaff_amd_tuning.so}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc2_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export HESAFF_AMD_LIB=$LIB HESAFF_OVERLAP=0
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --batch $BATCH --no-cpu-baseline --no-host-path $BENCH_EXTRA > $OUT/bench.json 2> $OUT/log.txt
cd $GRAFT_REPO_ROOT
python3 - $OUT <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float); calls = collections.Counter()
f = glob.glob(out + '/**/p_counter_collection.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]
    agg[n][r['Counter_Name']] += float(r['Counter_Value'])
f = glob.glob(out + '/**/p_kernel_trace.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]
    dur[n] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6; calls[n] += 1
print('| %-40s | %5s | %8s | %9s | %6s | %7s | %7s | %7s | %6s | %5s |' % ('kernel', 'calls', 'ms', 'VALU inst', 'VALU %', 'parked%', 'stall%', 'active%', 'waves/SIMD', 'GHz'))
print('|---|---|---|---|---|---|---|---|---|---|')
for n in sorted(dur, key=lambda k: -dur[k])[:int(__import__("os").environ.get("PMC_ROWS", "16"))]:
    if not n.startswith('k_'): continue
    a = agg[n]; ms = dur[n]
    wc = max(a['SQ_WAVE_CYCLES'], 1.0)
    # GRBM_GUI_ACTIVE: shader-clock cycles the GPU was busy during the kernel (summed over its launches) -> effective clock
    # (the counter is kept per XCD and rocprofv3 sums the eight of them)
    cyc = a['GRBM_GUI_ACTIVE'] / 8.0 if a['GRBM_GUI_ACTIVE'] > 0 else 2.4e6 * ms
    ghz = cyc / (ms * 1e6) if ms else 0
    valu_pct = 100.0 * a['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * cyc) if cyc else 0
    # average resident waves per SIMD = wave quad-cycles * 4 / (SIMD cycles of the kernel)
    occ = a['SQ_WAVE_CYCLES'] * 4 / (1024 * cyc) if cyc else 0
    print('| %-40s | %5d | %8.2f | %9.3g | %6.1f | %7.1f | %7.1f | %7.1f | %6.2f | %5.2f |' % (n, calls[n], ms, a['SQ_INSTS_VALU'], valu_pct,
          100 * a['SQ_WAIT_ANY'] / wc, 100 * a['SQ_WAIT_INST_ANY'] / wc, 100 * a['SQ_ACTIVE_INST_ANY'] / wc, occ, ghz))
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
