# usage (through gpurun): bash scripts/gpu_pmc_mem.sh <tag> [batch] [lib]
# vector-memory pipe counters of the serial-mode bench: is a kernel paced by the texture addresser / L1 (divergent gathers)?
cd $GRAFT_REPO_ROOT
TAG=$1; BATCH=${2:-8}; LIB=${3:-$GRAFT_REPO_ROOT/hesaff_amd/libhesaff_amd_tuning.so}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcm_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export HESAFF_AMD_LIB=$LIB HESAFF_OVERLAP=0
# two TA and two TCP counters fit one pass (more: "exceeds the capabilities of the hardware", and the aborted profiler hangs)
timeout -k 5 240 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --batch $BATCH --no-cpu-baseline --no-host-path $BENCH_EXTRA > $OUT/bench.json 2> $OUT/log.txt
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
print('| %-40s | %5s | %8s | %8s | %10s | %12s | %12s |' % ('kernel', 'calls', 'ms', 'TA busy%', 'TA waves', 'L1 accesses', 'L1->L2 reads'))
print('|---|---|---|---|---|---|---|')
for n in sorted(dur, key=lambda k: -dur[k])[:int(__import__("os").environ.get("PMC_ROWS", "16"))]:
    if not n.startswith('k_'): continue
    a = agg[n]; ms = dur[n]
    cyc = a['GRBM_GUI_ACTIVE'] / 8.0 if a['GRBM_GUI_ACTIVE'] > 0 else 2.4e6 * ms
    # *_sum counters add up the 256 CUs' texture addressers: busy % = / (256 x kernel cycles)
    d = 256.0 * cyc
    print('| %-40s | %5d | %8.2f | %8.1f | %10.3g | %12.3g | %12.3g |' % (n, calls[n], ms, 100 * a['TA_TA_BUSY_sum'] / d, a['TA_TOTAL_WAVEFRONTS_sum'],
          a['TCP_TOTAL_CACHE_ACCESSES_sum'], a['TCP_TCC_READ_REQ_sum']))
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
