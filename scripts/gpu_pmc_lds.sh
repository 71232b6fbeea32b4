# usage (through gpurun): bash scripts/gpu_pmc_lds.sh <tag> [batch] [lib]
# one rocprofv3 --pmc pass with the LDS counters of the serial-mode bench: is a kernel paced by the LDS pipe or by the VALU?
cd $GRAFT_REPO_ROOT
TAG=$1; BATCH=${2:-8}; LIB=${3:-$GRAFT_REPO_ROOT/hesaff_amd/libhesaff_amd_tuning.so}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcl_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export HESAFF_AMD_LIB=$LIB HESAFF_OVERLAP=0
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --batch $BATCH --no-cpu-baseline --no-host-path $BENCH_EXTRA > $OUT/bench.json 2> $OUT/log.txt
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
print('| %-40s | %5s | %8s | %9s | %9s | %8s | %8s | %8s | %6s |' % ('kernel', 'calls', 'ms', 'VALU inst', 'LDS inst', 'LDS busy%', 'conflict%', 'LDSwait', 'VALU %'))
print('|---|---|---|---|---|---|---|---|---|')
for n in sorted(dur, key=lambda k: -dur[k])[:int(__import__("os").environ.get("PMC_ROWS", "16"))]:
    if not n.startswith('k_'): continue
    a = agg[n]; ms = dur[n]
    cyc = a['GRBM_GUI_ACTIVE'] / 8.0 if a['GRBM_GUI_ACTIVE'] > 0 else 2.4e6 * ms
    # SQ_LDS_IDX_ACTIVE: LDS-array cycles summed over the 256 CUs; busy % = / (256 x kernel cycles)
    lds_busy = 100.0 * a['SQ_LDS_IDX_ACTIVE'] / (256 * cyc) if cyc else 0
    conf = 100.0 * a['SQ_LDS_BANK_CONFLICT'] / max(a['SQ_LDS_IDX_ACTIVE'], 1.0)
    valu_pct = 100.0 * a['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * cyc) if cyc else 0
    print('| %-40s | %5d | %8.2f | %9.3g | %9.3g | %8.1f | %8.1f | %8.3g | %6.1f |' % (n, calls[n], ms, a['SQ_INSTS_VALU'], a['SQ_INSTS_LDS'], lds_busy, conf, a['SQ_WAIT_INST_LDS'], valu_pct))
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
